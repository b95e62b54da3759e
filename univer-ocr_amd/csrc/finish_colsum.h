// Shared finish kernel of the weight-gradient kernels: float64 column sums of the block partials, then an epilogue
// that writes dw / db -- ONE launch, coalesced.
//
// The partial sums of a weight-gradient kernel are a matrix partial[row = block][column = accumulator] (float32).  The
// finish kernels of rounds 1-2 gave every output its own block whose threads walked DOWN a column: each 4-byte load a
// different 64-byte sector, re-read by the neighbouring columns' blocks (upsample+conv: 404 blocks x 1024 rows x 4
// scattered loads for a 2.4 MB matrix, 10 us).  Here block (cb, slice) owns 32 consecutive columns and a slice of the
// rows: 1024 threads = 32 columns x 32 row segments, 128-byte coalesced reads, 8 loads in flight per thread, segments
// and slices added in a fixed order.  Hand-offs inside the launch (uocr_common.h: sc1 stores, arrival ticket, sc1
// loads): the last slice of a column block to arrive adds the slices; the last column block to arrive runs the
// epilogue over the complete sums.  Counters come from ctx->sync and are left at zero.
#pragma once
#include <algorithm>

#include "uocr_common.h"

struct ColsumLayout {
    const float* partial;
    int nrows;              // blocks of the producing kernel
    int ncols;              // accumulators per block, all groups
    int group_cols;         // columns per group (= ncols when the matrix is one piece)
    size_t group_stride;    // floats between the groups' matrices (partial[group][row][group_cols])
    size_t row_stride;      // floats between rows
};

constexpr int COLSUM_MAX_SLICES = 16;

// Epi: struct with `int noutputs` and `__device__ void store(int e, const double* sums) const` reading sums[...] through
// colsum_get (the sums were published by other blocks)
__device__ __forceinline__ double colsum_get(const double* sums, int col) { return pub_load(sums + col); }

template <typename Epi>
__global__ __launch_bounds__(1024) void colsum_finish_kernel(ColsumLayout L, double* stage, double* sums,
                                                           unsigned* counters, Epi epi) {
    constexpr int NSEG = 32;
    __shared__ double seg[NSEG][32];
    __shared__ int flag;
    const int o = threadIdx.x & 31, sg = threadIdx.x >> 5, col = blockIdx.x * 32 + o;
    const int colblocks = gridDim.x, nslices = gridDim.y, rg = blockIdx.y;
    const int per_slice = (L.nrows + nslices - 1) / nslices;
    const int s0 = rg * per_slice, s1 = min(L.nrows, s0 + per_slice);
    double s = 0.0;
    if (col < L.ncols) {
        const float* src = L.partial + (size_t)(col / L.group_cols) * L.group_stride + (col % L.group_cols);
        const int per = (max(s1 - s0, 0) + NSEG - 1) / NSEG, b0 = s0 + sg * per, b1 = min(s1, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(b + k) * L.row_stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
        for (; b < b1; ++b) s += (double)src[(size_t)b * L.row_stride];
    }
    seg[sg][o] = s;
    __syncthreads();
    if (sg == 0) {
#pragma unroll
        for (int k = 1; k < NSEG; ++k) s += seg[k][o];
    }
    if (nslices > 1) {                                     // (block-uniform) the last slice adds the slices, in order
        if (sg == 0) pub_store(stage + (size_t)rg * (colblocks * 32) + col, s);
        if (threadIdx.x == 0) flag = sync_arrive(counters + blockIdx.x) == (unsigned)nslices - 1;   // (threads 0-31: one wave)
        __syncthreads();
        if (!flag) return;
        if (sg == 0) {
            s = 0.0;
            for (int r = 0; r < nslices; ++r) s += pub_load(stage + (size_t)r * (colblocks * 32) + col);
        }
        if (threadIdx.x == 0) sync_clear(counters + blockIdx.x);
        __syncthreads();
    }
    // the column block's sums are complete: publish; the last column block runs the epilogue
    if (sg == 0) pub_store(sums + col, s);
    if (threadIdx.x == 0) flag = sync_arrive(counters + colblocks) == (unsigned)colblocks - 1;
    __syncthreads();
    if (!flag) return;
    for (int e = threadIdx.x; e < epi.noutputs; e += blockDim.x) epi.store(e, sums);
    if (threadIdx.x == 0) sync_clear(counters + colblocks);
}

// workspace: the partial matrix is at ctx->workspace [0, partial_bytes); stage and sums go behind it.  `counter_base`:
// this call site's words in ctx->sync (it uses colblocks + 1 of them; sites on one stream never overlap in time, but
// distinct bases keep a fault in one from poisoning the others)
template <typename Epi>
int launch_colsum_finish(uocr_ctx* ctx, const ColsumLayout& L, size_t partial_bytes, int counter_base, const Epi& epi) {
    const int colblocks = (L.ncols + 31) / 32;
    const int nslices = std::min(COLSUM_MAX_SLICES, std::max(1, L.nrows / 256));
    const size_t off = (partial_bytes + 15) & ~(size_t)15;
    const size_t need = off + ((size_t)nslices + 1) * colblocks * 32 * sizeof(double);
    int rc = uocr_need_workspace(ctx, need);
    if (rc) return rc;
    UOCR_REQUIRE(ctx, (const char*)L.partial >= (const char*)ctx->workspace &&
                          (const char*)L.partial < (const char*)ctx->workspace + off);
    UOCR_REQUIRE(ctx, counter_base + colblocks + 1 <= UOCR_SYNC_WORDS);
    double* stage = (double*)((char*)ctx->workspace + off);
    double* sums = stage + (size_t)nslices * colblocks * 32;
    hipLaunchKernelGGL((colsum_finish_kernel<Epi>), dim3(colblocks, nslices), dim3(1024), 0, ctx->stream, L, stage, sums,
                       ctx->sync + counter_base, epi);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

// `s` (float64) -> *dst, scaled and accumulated as every weight-gradient finish does
__device__ __forceinline__ void colsum_out(float* dst, double s, bool live, float unscale, int accumulate) {
    s = live ? s * (double)unscale : 0.0;                  // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}
