// Weight gradients of the Line net's 4-channel convolutions for binary16 storage (UOCR_F16) on
// v_mfma_f32_16x16x16_f16 (reference: nn/layers/convolutional.py:101-145; upsample.py:21-39 for the decoder).
//
// dw contracts over POSITIONS, so the K index of an MFMA (4 consecutive binary16 values per lane) must run along
// image columns for a fixed channel -- the transpose of channels-last storage.  The staging step therefore writes
// channel PLANES into LDS (a 4x4 binary16 transpose in registers per 4 pixels, 64-bit LDS writes); an operand
// that is read at arbitrary column shifts is stored as PAIR WORDS (word[c] = (v[c], v[c+1])) so that any 4
// consecutive values are one ds_read2_b32.
//
// (1) 5x5, 4 -> 2, stride 1, padding 2 (the Line output conv):  with col = (dy column) - sx
//         dw[ty][4 - sx][ci][co] = sum_{row, col} xpad[row + ty - 2][col + 2][ci] * dy[row][col + sx][co]
//     i.e. the column shift of the tap lives on the dy operand (N = (co, sx): 10 of 16 columns), x is read at a
//     fixed, 8-byte aligned offset, M = (ty, ci) = 20 rows -> 2 MFMAs per 16 positions (a plain im2col GEMM needs
//     7 with 2 of 16 columns used).  Row 4 of the second M tile is all ones: its results are db.
// (2) Upsample2D(2) + 5x5 4 -> 4 on the low-res grid (conv_up.hip): dWeff[(my, mx, ci), (phase, co)]
//         = sum_pos xl[pos + m - 1][ci] * dy[2 pos + phase][co]:  N = (phase, co) = 16 columns exactly,
//     M = (mx, ci) per source row my -> 3 MFMAs per 16 low-res positions = 64 dy pixels; the phase sums are folded
//     back into the 5x5 kernel by conv_up.hip's finish kernel.
// Blocks are persistent over a flat tile index; one reduction per block, float64 finish.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "finish_group.h"
#include "up_phase.h"
#include "conv_dims.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ f32x4 mfma16(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t lo_pair(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }
__device__ __forceinline__ uint32_t hi_pair(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xFFFF0000u); }
__device__ __forceinline__ f16x4 read8(const _Float16* p) { return *reinterpret_cast<const f16x4*>(p); }
__device__ __forceinline__ f16x4 read_words(const uint32_t* p) {          // values c .. c+3 of a pair-word row
    const u32x2 v = {p[0], p[2]};
    return __builtin_bit_cast(f16x4, v);
}
constexpr uint32_t ONES = 0x3C003C00u;                                     // (1.0, 1.0) binary16

// ------------------------------------------------------------------------------------------------------------
// (1) 5x5 4 -> 2
// ------------------------------------------------------------------------------------------------------------
namespace e42 {
constexpr int BR = 16, BC = 64;                  // positions per tile
constexpr int XR = BR + 4;                       // x rows of a tile
constexpr int XRS = 72;                          // x plane row stride (halves): 36 dwords, with XP = 16 (mod 64) dwords
constexpr int XP = XR * XRS;                     //   the 16 (ty, ci) rows of an A read fall on disjoint banks
constexpr int DRS = 72;                          // dy plane row stride (pair words): 64 + 4 shifts + 2, 16-byte rows
constexpr int DP = BR * DRS + 32;                // dy plane stride (words): = 32 (mod 64)
constexpr int NV = 202;                          // 200 dw + 2 db
static_assert((XP / 2) % 64 == 16 && DP % 64 == 32, "bank layout");
}  // namespace e42

__global__ __launch_bounds__(256) void wgrad_h16_e42_kernel(const _Float16* __restrict__ x,
                                                            const _Float16* __restrict__ dy,
                                                            float* __restrict__ partial, int h, int wd, int tiles_x,
                                                            int tiles_y, int ntiles, float pad) {
    using namespace e42;
    __shared__ __attribute__((aligned(16))) _Float16 xs[4 * XP + 8];      // [ci][row][col] (+ the ones)
    __shared__ __attribute__((aligned(16))) uint32_t ds[2 * DP];          // [co][row][pair word]
    __shared__ float red[4][NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) reinterpret_cast<uint32_t*>(xs + 4 * XP)[tid] = ONES;
    // operand addresses of this lane inside a tile (group (r, c0) adds r * stride + c0)
    const int m_ty = n >> 2, m_ci = n & 3;
    const int a0 = m_ci * XP + m_ty * XRS + 4 * kq;                        // tile 0: ty = m / 4
    const int a1 = m_ci * XP + 4 * XRS + 4 * kq;                           // tile 1: ty = 4 (rows 0..3), row 4 = ones
    const bool ones_row = n == 4;
    const int sx = min(n & 7, 4), co = n >> 3;
    const int b0 = co * DP + 4 * kq + sx;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const uint32_t padw = __builtin_bit_cast(uint32_t, f16x2{(_Float16)pad, (_Float16)pad});
    // staging roles: x unit = 4 columns x 4 channels (32 bytes), 16 units per row; dy unit = 4 pixels (16 bytes) +
    // the next one, 18 units per row
    const int xu = tid & 15, xr0 = tid >> 4;
    const int du = tid % 18, dr0 = tid / 18;

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC - 4, R0 = trow * BR;
        const _Float16* xb = x + (size_t)img * h * wd * 4;
        const _Float16* gb = dy + (size_t)img * h * wd * 2;
        __syncthreads();                                 // the previous tile's reads are over
        {   // ---- x: image (R0 - 2 + r, C0 + 2 + 4 xu + p) -> planes
            const int gx0 = C0 + 2 + 4 * xu;
            const bool all_in = gx0 >= 0 && gx0 + 3 < wd, any_in = gx0 + 3 >= 0 && gx0 < wd;
            uint4 v[3][2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = xr0 + 16 * k, gy = R0 - 2 + r;
                const bool row_ok = r < XR && (unsigned)gy < (unsigned)h;
                const _Float16* src = xb + ((size_t)min(max(gy, 0), h - 1) * wd) * 4;
                v[k][0] = v[k][1] = uint4{padw, padw, padw, padw};
                if (row_ok && all_in) {
                    v[k][0] = *reinterpret_cast<const uint4*>(src + (size_t)gx0 * 4);
                    v[k][1] = *reinterpret_cast<const uint4*>(src + (size_t)gx0 * 4 + 8);
                } else if (row_ok && any_in) {
                    uint32_t* vw = reinterpret_cast<uint32_t*>(&v[k][0]);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wd) {
                            const uint32_t* sp = reinterpret_cast<const uint32_t*>(src + (size_t)(gx0 + p) * 4);
                            vw[2 * p] = sp[0];
                            vw[2 * p + 1] = sp[1];
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = xr0 + 16 * k;
                if (r >= XR) continue;
                // pixel p = dwords (2p, 2p+1) = (c0 | c1 << 16, c2 | c3 << 16); plane ci gets (P0.ci, P1.ci, P2.ci, P3.ci)
                const uint32_t* d = reinterpret_cast<const uint32_t*>(&v[k][0]);
                _Float16* row = xs + r * XRS + 4 * xu;
                *reinterpret_cast<u32x2*>(row + 0 * XP) = u32x2{lo_pair(d[0], d[2]), lo_pair(d[4], d[6])};
                *reinterpret_cast<u32x2*>(row + 1 * XP) = u32x2{hi_pair(d[0], d[2]), hi_pair(d[4], d[6])};
                *reinterpret_cast<u32x2*>(row + 2 * XP) = u32x2{lo_pair(d[1], d[3]), lo_pair(d[5], d[7])};
                *reinterpret_cast<u32x2*>(row + 3 * XP) = u32x2{hi_pair(d[1], d[3]), hi_pair(d[5], d[7])};
            }
        }
        {   // ---- dy: image (R0 + r, C0 + 4 du + p), p = 0..4 -> pair-word planes, zero outside the image
            const int gx0 = C0 + 4 * du;
            const bool all_in = gx0 >= 0 && gx0 + 4 < wd, any_in = gx0 + 4 >= 0 && gx0 < wd;
            uint32_t g[3][5];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = dr0 + 14 * k, gy = R0 + r;
                const bool row_ok = r < BR && dr0 < 14 && gy < h;
                const uint32_t* src = reinterpret_cast<const uint32_t*>(gb + (size_t)min(gy, h - 1) * wd * 2);
#pragma unroll
                for (int p = 0; p < 5; ++p) g[k][p] = 0u;
                if (row_ok && all_in) {
                    const uint4 q = *reinterpret_cast<const uint4*>(src + gx0);
                    g[k][0] = q.x, g[k][1] = q.y, g[k][2] = q.z, g[k][3] = q.w;
                    g[k][4] = src[gx0 + 4];
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wd) g[k][p] = src[gx0 + p];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = dr0 + 14 * k;
                if (r >= BR || dr0 >= 14) continue;
                uint32_t* row = ds + r * DRS + 4 * du;
                *reinterpret_cast<uint4*>(row) = uint4{lo_pair(g[k][0], g[k][1]), lo_pair(g[k][1], g[k][2]),
                                                       lo_pair(g[k][2], g[k][3]), lo_pair(g[k][3], g[k][4])};
                *reinterpret_cast<uint4*>(row + DP) = uint4{hi_pair(g[k][0], g[k][1]), hi_pair(g[k][1], g[k][2]),
                                                            hi_pair(g[k][2], g[k][3]), hi_pair(g[k][3], g[k][4])};
            }
        }
        __syncthreads();
        // ---- wave wv: rows wv, wv + 4, ...; 4 groups of 16 positions per row
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const f16x4 b = read_words(ds + b0 + r * DRS + c0);
                const f16x4 x0 = read8(xs + a0 + r * XRS + c0);
                const f16x4 x1 = read8(xs + (ones_row ? 4 * XP : a1 + r * XRS + c0));
                acc0 = mfma16(x0, b, acc0);
                acc1 = mfma16(x1, b, acc1);
            }
        }
    }
    // ---- block reduction: lane (n = (co, sx), kq) holds dw[ty = kq][tx = 4 - sx][ci = i][co] (acc0),
    // dw[4][4 - sx][i][co] (acc1, kq = 0) and db[co] (acc1[0] at kq = 1, sx = 0)
    const bool col_ok = (n & 7) <= 4;
    for (int i = tid; i < 4 * NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    if (col_ok) {
        const int tx = 4 - (n & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[wv][((kq * 5 + tx) * 4 + i) * 2 + co] = acc0[i];
            if (kq == 0) red[wv][((4 * 5 + tx) * 4 + i) * 2 + co] = acc1[i];
        }
        if (kq == 1 && (n & 7) == 0) red[wv][200 + co] = acc1[0];
    }
    __syncthreads();
    if (tid < NV) partial[(size_t)blockIdx.x * NV + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------------------------------
// (1b) 5x5 1 -> 1 (the Paragraph output conv): the same shift trick with M = ty (5 rows + the ones row for db),
// N = sx: one MFMA per 16 positions; one channel is its own plane, so x is staged as it lies in memory
// ------------------------------------------------------------------------------------------------------------
namespace e11 {
constexpr int BR = 32, BC = 64;
constexpr int XR = BR + 4, XRS = 72;             // x rows / row stride (halves)
constexpr int DRS = 72;                          // dy row stride (pair words)
constexpr int NV = 26;                           // 25 dw + db
}  // namespace e11

__global__ __launch_bounds__(256) void wgrad_h16_e11_kernel(const _Float16* __restrict__ x,
                                                            const _Float16* __restrict__ dy,
                                                            float* __restrict__ partial, int h, int wd, int tiles_x,
                                                            int tiles_y, int ntiles, float pad) {
    using namespace e11;
    __shared__ __attribute__((aligned(16))) _Float16 xs[XR * XRS + 8];    // [row][col] (+ the ones)
    __shared__ __attribute__((aligned(16))) uint32_t ds[BR * DRS];        // [row][pair word]
    __shared__ float red[4][NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) reinterpret_cast<uint32_t*>(xs + XR * XRS)[tid] = ONES;
    const int a0 = min(n, 4) * XRS + 4 * kq;             // A row m = ty (0..4); row 5 = ones; the rest is not read back
    const bool ones_row = n == 5;
    const int b0 = 4 * kq + min(n, 4);                   // B column n = sx
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint32_t padw = __builtin_bit_cast(uint32_t, f16x2{(_Float16)pad, (_Float16)pad});
    const int xu = tid & 7, xr0 = tid >> 3;              // x unit = 8 pixels (16 bytes), 8 units per row
    const int du = tid % 18, dr0 = tid / 18;             // dy unit = 4 pixels + the next one, 18 units per row

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC - 4, R0 = trow * BR;
        const _Float16* xb = x + (size_t)img * h * wd;
        const _Float16* gb = dy + (size_t)img * h * wd;
        __syncthreads();                                 // the previous tile's reads are over
        {   // ---- x: image (R0 - 2 + r, C0 + 2 + 8 xu + p)
            const int gx0 = C0 + 2 + 8 * xu;
            const bool all_in = gx0 >= 0 && gx0 + 7 < wd, any_in = gx0 + 7 >= 0 && gx0 < wd;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = xr0 + 32 * k, gy = R0 - 2 + r;
                if (r >= XR) continue;
                const bool row_ok = (unsigned)gy < (unsigned)h;
                const _Float16* src = xb + (size_t)min(max(gy, 0), h - 1) * wd;
                uint4 v = uint4{padw, padw, padw, padw};
                if (row_ok && all_in) {
                    v = *reinterpret_cast<const uint4*>(src + gx0);                   // (2-byte aligned at least)
                } else if (row_ok && any_in) {
                    unsigned short* vh = reinterpret_cast<unsigned short*>(&v);
#pragma unroll
                    for (int p = 0; p < 8; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wd) vh[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                }
                *reinterpret_cast<uint4*>(xs + r * XRS + 8 * xu) = v;
            }
        }
        {   // ---- dy: image (R0 + r, C0 + 4 du + p), p = 0..4 -> pair words, zero outside the image
            const int gx0 = C0 + 4 * du;
            const bool all_in = gx0 >= 0 && gx0 + 4 < wd, any_in = gx0 + 4 >= 0 && gx0 < wd;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int r = dr0 + 14 * k, gy = R0 + r;
                if (r >= BR || dr0 >= 14) continue;
                const bool row_ok = gy < h;
                const _Float16* src = gb + (size_t)min(gy, h - 1) * wd;
                uint32_t g[5] = {0u, 0u, 0u, 0u, 0u};
                if (row_ok && all_in) {
                    const uint2 q = *reinterpret_cast<const uint2*>(src + gx0);
                    g[0] = q.x & 0xFFFFu, g[1] = q.x >> 16, g[2] = q.y & 0xFFFFu, g[3] = q.y >> 16;
                    g[4] = __builtin_bit_cast(unsigned short, src[gx0 + 4]);
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wd) g[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                }
                *reinterpret_cast<uint4*>(ds + r * DRS + 4 * du) =
                    uint4{g[0] | (g[1] << 16), g[1] | (g[2] << 16), g[2] | (g[3] << 16), g[3] | (g[4] << 16)};
            }
        }
        __syncthreads();
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const f16x4 b = read_words(ds + b0 + r * DRS + c0);
                const f16x4 a = read8(xs + (ones_row ? XR * XRS : a0 + r * XRS + c0));
                acc = mfma16(a, b, acc);
            }
        }
    }
    // ---- lane (n = sx, kq) holds rows m = 4kq + i: dw[ty = m][tx = 4 - sx] for m < 5, db at m = 5 (sx = 0)
    for (int i = tid; i < 4 * NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    if (n <= 4 && kq < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * kq + i;
            if (m < 5) red[wv][m * 5 + 4 - n] = acc[i];
            if (m == 5 && n == 0) red[wv][25] = acc[i];
        }
    }
    __syncthreads();
    if (tid < NV) partial[(size_t)blockIdx.x * NV + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// block a: out[a] (+)= unscale * sum over blocks of partial[blk][a]; a < ndw -> dw, else db
__global__ __launch_bounds__(256) void wgrad_h16_finish(const float* __restrict__ partial, int nv, int ndw,
                                                        float* __restrict__ dw, float* __restrict__ db, int nblocks,
                                                        int use_bias, int accumulate, float unscale) {
    __shared__ double smem[16];
    const int a = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += (double)partial[(size_t)i * nv + a];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst = a < ndw ? dw + a : db + (a - ndw);
    if (a >= ndw && !use_bias) s = 0.0;
    s *= (double)unscale;                                  // UOCR_F16_SCALED(k): 2^-k
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// the same finish as a record of an open deferred group (finish_group.h); false: launch wgrad_h16_finish
static bool h16_finish_deferred(uocr_ctx* ctx, const float* partial, int nv, int ndw, float* dw, float* db, int nblocks,
                                int use_bias, int accumulate, float unscale) {
    FinishDesc fd{};
    fd.kind = FIN_COLS;
    fd.partial = partial;
    fd.nblocks = nblocks;
    fd.ncols = fd.group_cols = nv;
    fd.row_stride = nv;
    fd.dw = dw, fd.db = db;
    fd.use_bias = use_bias, fd.accumulate = accumulate;
    fd.unscale = unscale;
    fd.p[0] = ndw;
    return uocr_finish_defer(ctx, fd);
}

// ------------------------------------------------------------------------------------------------------------
// (2) upsample2x + 5x5 4 -> 4 on the low-res grid
// ------------------------------------------------------------------------------------------------------------
namespace up {
constexpr int BR = 16, BC = 64;                  // low-res positions per tile
constexpr int XR = BR + 2, XRS = 68;             // xl pair-word planes: rows, row stride (words): 64 + 2 + lookahead
constexpr int XPW = XR * XRS + 8;                // plane stride (words) = 16 (mod 64)
constexpr int GRS = 64;                          // dy plane row stride (halves)
constexpr int GP = BR * GRS + 8;                 // dy plane stride (halves): 516 dwords = 4 (mod 64)
constexpr int NV = 36 * 16 + 4;
static_assert(XPW % 64 == 16 && (GP / 2) % 64 == 4, "bank layout");
}  // namespace up

__global__ __launch_bounds__(256) void wgrad_h16_up_kernel(const _Float16* __restrict__ xl,
                                                           const _Float16* __restrict__ dy,
                                                           float* __restrict__ partial, int hl, int wl, int tiles_x,
                                                           int tiles_y, int ntiles) {
    using namespace up;
    __shared__ __attribute__((aligned(16))) uint32_t xs[4 * XPW + 4];     // [ci][row][pair word] (+ the ones)
    __shared__ __attribute__((aligned(16))) _Float16 gs[16 * GP];         // [(py, px, co)][row][col]
    __shared__ float red[4][NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) xs[4 * XPW + tid] = ONES;
    // A rows m = mx * 4 + ci (12 of 16; row 12 of the my = 0 tile is all ones -> db); B column n = (py, px, co)
    const int mx = min(n >> 2, 2), ci = n & 3;
    const int a0 = ci * XPW + 4 * kq + mx;
    const bool ones_row = n == 12;
    const int b0 = n * GP + 4 * kq;
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int H = 2 * hl, W = 2 * wl;
    // staging roles: xl unit = 4 pixels + the next one (40 bytes), 17 units per row; dy unit = 4 low-res columns of
    // one high-res row = 8 pixels (64 bytes), 16 units per row
    const int xu = tid % 17, xr0 = tid / 17;
    const int gu = tid & 15, gr0 = tid >> 4;

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC, R0 = trow * BR;
        const _Float16* xb = xl + (size_t)img * hl * wl * 4;
        const _Float16* gb = dy + (size_t)img * H * W * 4;
        __syncthreads();                                 // the previous tile's reads are over
        {   // ---- xl: (R0 - 1 + r, C0 - 1 + 4 xu + p), p = 0..4 -> pair-word planes, zero outside
            const int gx0 = C0 - 1 + 4 * xu;
            const bool all_in = gx0 >= 0 && gx0 + 4 < wl, any_in = gx0 + 4 >= 0 && gx0 < wl;
            uint32_t v[2][10];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = xr0 + 15 * k, gy = R0 - 1 + r;
                const bool row_ok = r < XR && xr0 < 15 && (unsigned)gy < (unsigned)hl;
                const uint32_t* src = reinterpret_cast<const uint32_t*>(xb + (size_t)min(max(gy, 0), hl - 1) * wl * 4);
#pragma unroll
                for (int p = 0; p < 10; ++p) v[k][p] = 0u;
                if (row_ok && all_in) {
                    const uint4 q0 = *reinterpret_cast<const uint4*>(src + 2 * gx0);
                    const uint4 q1 = *reinterpret_cast<const uint4*>(src + 2 * gx0 + 4);
                    const uint2 q2 = *reinterpret_cast<const uint2*>(src + 2 * gx0 + 8);
                    v[k][0] = q0.x, v[k][1] = q0.y, v[k][2] = q0.z, v[k][3] = q0.w;
                    v[k][4] = q1.x, v[k][5] = q1.y, v[k][6] = q1.z, v[k][7] = q1.w;
                    v[k][8] = q2.x, v[k][9] = q2.y;
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wl) {
                            v[k][2 * p] = src[2 * (gx0 + p)];
                            v[k][2 * p + 1] = src[2 * (gx0 + p) + 1];
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = xr0 + 15 * k;
                if (r >= XR || xr0 >= 15) continue;
                const uint32_t* d = v[k];                // pixel p = dwords (2p, 2p+1)
                uint32_t* row = xs + r * XRS + 4 * xu;
                *reinterpret_cast<uint4*>(row + 0 * XPW) = uint4{lo_pair(d[0], d[2]), lo_pair(d[2], d[4]), lo_pair(d[4], d[6]), lo_pair(d[6], d[8])};
                *reinterpret_cast<uint4*>(row + 1 * XPW) = uint4{hi_pair(d[0], d[2]), hi_pair(d[2], d[4]), hi_pair(d[4], d[6]), hi_pair(d[6], d[8])};
                *reinterpret_cast<uint4*>(row + 2 * XPW) = uint4{lo_pair(d[1], d[3]), lo_pair(d[3], d[5]), lo_pair(d[5], d[7]), lo_pair(d[7], d[9])};
                *reinterpret_cast<uint4*>(row + 3 * XPW) = uint4{hi_pair(d[1], d[3]), hi_pair(d[3], d[5]), hi_pair(d[5], d[7]), hi_pair(d[7], d[9])};
            }
        }
        {   // ---- dy: high-res row 2 (R0 + r) + py, low-res columns C0 + 4 gu + j -> planes (py, px, co)
            const int gx0 = C0 + 4 * gu;
            const bool all_in = gx0 + 3 < wl, any_in = gx0 < wl;
            uint4 q[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rr = gr0 + 16 * k, r = rr >> 1, py = rr & 1, gy = 2 * (R0 + r) + py;
                const bool row_ok = gy < H;
                const _Float16* src = gb + ((size_t)min(gy, H - 1) * W + 2 * gx0) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q[k][j] = uint4{0u, 0u, 0u, 0u};
                    if (row_ok && (all_in || (any_in && gx0 + j < wl))) q[k][j] = *reinterpret_cast<const uint4*>(src + 8 * j);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rr = gr0 + 16 * k, r = rr >> 1, py = rr & 1;
                // q[j] = low-res column j: (px0: c0|c1, c2|c3), (px1: c0|c1, c2|c3); plane (py, px, co) gets columns 0..3
                _Float16* row = gs + (py * 8) * GP + r * GRS + 4 * gu;
                const uint4* c = q[k];
                *reinterpret_cast<u32x2*>(row + 0 * GP) = u32x2{lo_pair(c[0].x, c[1].x), lo_pair(c[2].x, c[3].x)};
                *reinterpret_cast<u32x2*>(row + 1 * GP) = u32x2{hi_pair(c[0].x, c[1].x), hi_pair(c[2].x, c[3].x)};
                *reinterpret_cast<u32x2*>(row + 2 * GP) = u32x2{lo_pair(c[0].y, c[1].y), lo_pair(c[2].y, c[3].y)};
                *reinterpret_cast<u32x2*>(row + 3 * GP) = u32x2{hi_pair(c[0].y, c[1].y), hi_pair(c[2].y, c[3].y)};
                *reinterpret_cast<u32x2*>(row + 4 * GP) = u32x2{lo_pair(c[0].z, c[1].z), lo_pair(c[2].z, c[3].z)};
                *reinterpret_cast<u32x2*>(row + 5 * GP) = u32x2{hi_pair(c[0].z, c[1].z), hi_pair(c[2].z, c[3].z)};
                *reinterpret_cast<u32x2*>(row + 6 * GP) = u32x2{lo_pair(c[0].w, c[1].w), lo_pair(c[2].w, c[3].w)};
                *reinterpret_cast<u32x2*>(row + 7 * GP) = u32x2{hi_pair(c[0].w, c[1].w), hi_pair(c[2].w, c[3].w)};
            }
        }
        __syncthreads();
        // ---- wave wv: rows wv, wv + 4, ...; xl tile row of position row r and source offset my is r + my
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const f16x4 b = read8(gs + b0 + r * GRS + c0);
                const f16x4 x0 = read_words(xs + (ones_row ? 4 * XPW : a0 + r * XRS + c0));
                const f16x4 x1 = read_words(xs + a0 + (r + 1) * XRS + c0);
                const f16x4 x2 = read_words(xs + a0 + (r + 2) * XRS + c0);
                acc[0] = mfma16(x0, b, acc[0]);
                acc[1] = mfma16(x1, b, acc[1]);
                acc[2] = mfma16(x2, b, acc[2]);
            }
        }
    }
    // ---- block reduction: lane (n = (phase, co), kq = mx) holds dWeff[(my * 3 + mx) * 4 + ci = i][n]; db from the
    // ones row (my = 0, kq = 3, i = 0), summed over the phases
    for (int i = tid; i < 4 * NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int my = 0; my < 3; ++my) {
        if (kq < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wv][(((my * 3 + kq) * 4 + i) * 16) + n] = acc[my][i];
        }
    }
    float dbv = kq == 3 ? acc[0][0] : 0.f;               // lanes 48..63: n = (phase, co)
    dbv += __shfl_xor(dbv, 4, 64);
    dbv += __shfl_xor(dbv, 8, 64);
    if (kq == 3 && n < 4) red[wv][36 * 16 + n] = dbv;
    __syncthreads();
    // the block's partial row: dw and db themselves (phase entries added here: up_phase.h)
    up4_write_row(partial + (size_t)blockIdx.x * UP4_NOUT, [&](int i) { return red[0][i] + red[1][i] + red[2][i] + red[3][i]; },
                  tid, 256);
}

// ------------------------------------------------------------------------------------------------------------
// (2b) upsample2x + 5x5 1 -> 1 (the Paragraph decoder blocks): dWeff[(my, mx)][phase] = sum_pos xl[pos + m - 1] *
// dy[2 pos + phase].  One xl pair-word plane; dy as four PHASE planes (row parity from the row, column parity
// de-interleaved while staging); per source row my one MFMA with M = mx (3 rows + the ones row), N = phase.
// Partial rows as conv_up.hip's 1-channel producer writes them: dw (5 x 5) and db (up_phase.h).
// ------------------------------------------------------------------------------------------------------------
namespace up1 {
constexpr int BR = 16, BC = 64;                  // low-res positions per tile
constexpr int XR = BR + 2, XRS = 68;             // xl pair words: rows, row stride
constexpr int GRS = 72;                          // dy phase plane row stride (halves)
constexpr int GP = BR * GRS + 8;                 // phase plane stride (halves)
constexpr int NV = 37;
}  // namespace up1

__global__ __launch_bounds__(256) void wgrad_h16_up1_kernel(const _Float16* __restrict__ xl,
                                                            const _Float16* __restrict__ dy,
                                                            float* __restrict__ partial, int hl, int wl, int tiles_x,
                                                            int tiles_y, int ntiles) {
    using namespace up1;
    __shared__ __attribute__((aligned(16))) uint32_t xs[XR * XRS + 4];    // [row][pair word] (+ the ones)
    __shared__ __attribute__((aligned(16))) _Float16 gs[4 * GP];          // [(py, px)][row][col]
    __shared__ float red[4][NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) xs[XR * XRS + tid] = ONES;
    // A rows m = mx (0..2); row 3 of the my = 0 tile is all ones -> db.  B column n = phase (0..3)
    const int a0 = 4 * kq + min(n, 2);
    const bool ones_row = n == 3;
    const int b0 = min(n, 3) * GP + 4 * kq;
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int H = 2 * hl, W = 2 * wl;
    const int xu = tid % 17, xr0 = tid / 17;             // xl unit = 4 pixels + the next one, 17 units per row
    const int gu = tid & 15, gr0 = tid >> 4;             // dy unit = 8 high-res pixels (16 bytes) of one row

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC, R0 = trow * BR;
        const _Float16* xb = xl + (size_t)img * hl * wl;
        const _Float16* gb = dy + (size_t)img * H * W;
        __syncthreads();
        {   // ---- xl: (R0 - 1 + r, C0 - 1 + 4 xu + p), p = 0..4 -> pair words, zero outside
            const int gx0 = C0 - 1 + 4 * xu;
            const bool all_in = gx0 >= 0 && gx0 + 4 < wl, any_in = gx0 + 4 >= 0 && gx0 < wl;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = xr0 + 15 * k, gy = R0 - 1 + r;
                if (r >= XR || xr0 >= 15) continue;
                const bool row_ok = (unsigned)gy < (unsigned)hl;
                const _Float16* src = xb + (size_t)min(max(gy, 0), hl - 1) * wl;
                uint32_t g[5] = {0u, 0u, 0u, 0u, 0u};
                if (row_ok && all_in) {
                    const uint2 q = *reinterpret_cast<const uint2*>(src + gx0);
                    g[0] = q.x & 0xFFFFu, g[1] = q.x >> 16, g[2] = q.y & 0xFFFFu, g[3] = q.y >> 16;
                    g[4] = __builtin_bit_cast(unsigned short, src[gx0 + 4]);
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)wl) g[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                }
                *reinterpret_cast<uint4*>(xs + r * XRS + 4 * xu) =
                    uint4{g[0] | (g[1] << 16), g[1] | (g[2] << 16), g[2] | (g[3] << 16), g[3] | (g[4] << 16)};
            }
        }
        {   // ---- dy: high-res row 2 (R0 + r) + py, pixels 2 C0 + 8 gu + p (p = 0..7) -> planes (py, px) at column 4 gu + p / 2
            const int gx0 = 2 * C0 + 8 * gu;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int rr = gr0 + 16 * k, r = rr >> 1, py = rr & 1, gy = 2 * (R0 + r) + py;
                const _Float16* src = gb + (size_t)min(gy, H - 1) * W;
                uint32_t d[4] = {0u, 0u, 0u, 0u};
                if (gy < H) {
                    if (gx0 + 8 <= W) {
                        const uint4 v = *reinterpret_cast<const uint4*>(src + gx0);
                        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
                    } else if (gx0 < W) {
                        unsigned short* dh = reinterpret_cast<unsigned short*>(d);
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if (gx0 + p < W) dh[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                    }
                }
                _Float16* row = gs + (py * 2) * GP + r * GRS + 4 * gu;
                *reinterpret_cast<u32x2*>(row) = u32x2{lo_pair(d[0], d[1]), lo_pair(d[2], d[3])};           // px = 0
                *reinterpret_cast<u32x2*>(row + GP) = u32x2{hi_pair(d[0], d[1]), hi_pair(d[2], d[3])};      // px = 1
            }
        }
        __syncthreads();
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const f16x4 b = read8(gs + b0 + r * GRS + c0);
                const f16x4 x0 = read_words(xs + (ones_row ? XR * XRS : a0 + r * XRS + c0));
                const f16x4 x1 = read_words(xs + a0 + (r + 1) * XRS + c0);
                const f16x4 x2 = read_words(xs + a0 + (r + 2) * XRS + c0);
                acc[0] = mfma16(x0, b, acc[0]);
                acc[1] = mfma16(x1, b, acc[1]);
                acc[2] = mfma16(x2, b, acc[2]);
            }
        }
    }
    // ---- lane (n = phase, kq = 0) holds rows m = i: dWeff[(my * 3 + mx = i) * 4 + phase]; row 3 of my = 0: db per phase
    for (int i = tid; i < 4 * NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    float dbv = (kq == 0 && n < 4) ? acc[0][3] : 0.f;
    dbv += __shfl_xor(dbv, 1, 64);
    dbv += __shfl_xor(dbv, 2, 64);
    if (kq == 0 && n < 4) {
#pragma unroll
        for (int my = 0; my < 3; ++my)
#pragma unroll
            for (int i = 0; i < 3; ++i) red[wv][(my * 3 + i) * 4 + n] = acc[my][i];
        if (n == 0) red[wv][36] = dbv;
    }
    __syncthreads();
    up1_write_row(partial + (size_t)blockIdx.x * UP1_NOUT, [&](int i) { return red[0][i] + red[1][i] + red[2][i] + red[3][i]; },
                  tid);
}

// ------------------------------------------------------------------------------------------------------------
// (3) 5x5 / stride 2 / padding 2 (the encoder convs: 4 -> 4, 1 -> 4, 1 -> 1)
//     dw[ty][tx][ci][co] = sum_{Y, X} xpad[2Y + ty - 2][2X + tx - 2][ci] * dy[Y][X][co]
// The stride makes the x operand every second column: x is staged as EVEN / ODD column planes per channel
// (E[j] = x[2j], O[j] = x[2j+1]); tap column tx = 2e reads E[X + e - 1], tx = 2o + 1 reads O[X + o - 1].  As in (1)
// the shift s in {-1, 0, +1} moves to the dy operand: with Q = X + s, N = (co, sx = 1 - s) reads dy[Q + sx - 1]
// (pair words) and M = (parity, ty, ci) reads its plane at Q (aligned): 10 * CI rows -> 3 MFMAs (CI = 4) or 1 (CI = 1)
// per 16 positions; (odd plane, sx = 0) would be tx = 5 and is dropped; a row of ones yields db.
// ------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
struct S2 {
    static constexpr int BR = 8, BC = 64;            // positions (Y, Q) per tile
    static constexpr int XR = 2 * BR + 3;            // x rows of a tile
    static constexpr int XRS = 72;                   // plane row stride (halves): 68 used
    static constexpr int XP = XR * XRS + 8;          // plane stride (halves)
    static constexpr int DRS = 72;                   // dy plane row stride (pair words)
    static constexpr int DP = BR * DRS + 8;          // dy plane stride (words)
    static constexpr int NT = (10 * CI + 1 + 15) / 16;   // M tiles (the ones row included)
    static constexpr int NV = 25 * CI * CO + CO;
    static constexpr int XU = 17;                    // x staging units (8 pixels) per row: 136 >= 132 pixels
};

template <int CI, int CO>
__global__ __launch_bounds__(256) void wgrad_h16_s2_kernel(const _Float16* __restrict__ x,
                                                           const _Float16* __restrict__ dy,
                                                           float* __restrict__ partial, int h, int wd, int oh, int ow,
                                                           int tiles_x, int tiles_y, int ntiles, float pad) {
    using G = S2<CI, CO>;
    constexpr int BR = G::BR, BC = G::BC, XR = G::XR, XRS = G::XRS, XP = G::XP, DRS = G::DRS, DP = G::DP, NT = G::NT;
    __shared__ __attribute__((aligned(16))) _Float16 xs[2 * CI * XP + 8];   // [(ci, parity)][row][j] (+ the ones)
    __shared__ __attribute__((aligned(16))) uint32_t ds[CO * DP];           // [co][row][pair word]
    __shared__ float red[4][G::NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) reinterpret_cast<uint32_t*>(xs + 2 * CI * XP)[tid] = ONES;
    // A rows m = (parity * 5 + ty) * CI + ci (10 * CI of them), then the ones row; per M tile the lane's plane offset
    int a_off[NT];
    bool a_ones[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = 16 * t + n, mm = min(m, 10 * CI - 1);
        const int ci = mm % CI, pt = mm / CI, ty = pt % 5, par = pt / 5;
        a_off[t] = (ci * 2 + par) * XP + ty * XRS + 4 * kq;      // + 2 r * XRS + c0
        a_ones[t] = m == 10 * CI;
    }
    // B column n = co * 4 + sx (CO = 4) / sx (CO = 1); sx = 3 (and columns >= 3 for CO = 1) are not read back
    const int b_sx = min(CO == 4 ? (n & 3) : n, 2), b_co = CO == 4 ? (n >> 2) : 0;
    const int b0 = b_co * DP + 4 * kq + b_sx;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t padw = __builtin_bit_cast(uint32_t, f16x2{(_Float16)pad, (_Float16)pad});
    const int xu = tid % G::XU, xr0 = tid / G::XU;       // x unit = 8 pixels, 17 units per row, 15 rows per pass
    const int du = tid % 18, dr0 = tid / 18;             // dy unit = 4 pixels + the next one, 18 units per row

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC - 1, R0 = trow * BR;   // plane index Q of tile column 0; first position row
        const _Float16* xb = x + (size_t)img * h * wd * CI;
        const _Float16* gb = dy + (size_t)img * oh * ow * CO;
        __syncthreads();                                 // the previous tile's reads are over
        {   // ---- x: image row 2 R0 - 2 + r, pixels 2 C0 + 8 xu + p (p = 0..7) -> even / odd planes at j = 4 xu + p / 2
            const int gx0 = 2 * C0 + 8 * xu;
            const bool all_in = gx0 >= 0 && gx0 + 7 < wd, any_in = gx0 + 7 >= 0 && gx0 < wd;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = xr0 + 15 * k, gy = 2 * R0 - 2 + r;
                if (r >= XR || xr0 >= 15) continue;
                const bool row_ok = (unsigned)gy < (unsigned)h;
                const _Float16* src = xb + (size_t)min(max(gy, 0), h - 1) * wd * CI;
                _Float16* row = xs + r * XRS + 4 * xu;
                if constexpr (CI == 4) {
                    uint32_t d[16];                      // pixel p = dwords (2p, 2p+1) = (c0 | c1 << 16, c2 | c3 << 16)
#pragma unroll
                    for (int q = 0; q < 16; ++q) d[q] = padw;
                    if (row_ok && all_in) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)gx0 * 4 + 8 * q);
                            d[4 * q] = v.x, d[4 * q + 1] = v.y, d[4 * q + 2] = v.z, d[4 * q + 3] = v.w;
                        }
                    } else if (row_ok && any_in) {
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if ((unsigned)(gx0 + p) < (unsigned)wd) {
                                const uint32_t* sp = reinterpret_cast<const uint32_t*>(src + (size_t)(gx0 + p) * 4);
                                d[2 * p] = sp[0], d[2 * p + 1] = sp[1];
                            }
                    }
                    // plane (ci, parity) <- that channel of the pixels parity, parity + 2, parity + 4, parity + 6
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        const int o = 2 * par;
                        *reinterpret_cast<u32x2*>(row + (0 * 2 + par) * XP) = u32x2{lo_pair(d[o], d[o + 4]), lo_pair(d[o + 8], d[o + 12])};
                        *reinterpret_cast<u32x2*>(row + (1 * 2 + par) * XP) = u32x2{hi_pair(d[o], d[o + 4]), hi_pair(d[o + 8], d[o + 12])};
                        *reinterpret_cast<u32x2*>(row + (2 * 2 + par) * XP) = u32x2{lo_pair(d[o + 1], d[o + 5]), lo_pair(d[o + 9], d[o + 13])};
                        *reinterpret_cast<u32x2*>(row + (3 * 2 + par) * XP) = u32x2{hi_pair(d[o + 1], d[o + 5]), hi_pair(d[o + 9], d[o + 13])};
                    }
                } else {
                    uint32_t d[4] = {padw, padw, padw, padw};         // 8 one-channel pixels
                    if (row_ok && all_in) {
                        const uint4 v = *reinterpret_cast<const uint4*>(src + gx0);
                        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
                    } else if (row_ok && any_in) {
                        unsigned short* dh = reinterpret_cast<unsigned short*>(d);
#pragma unroll
                        for (int p = 0; p < 8; ++p)
                            if ((unsigned)(gx0 + p) < (unsigned)wd) dh[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                    }
                    *reinterpret_cast<u32x2*>(row + 0 * XP) = u32x2{lo_pair(d[0], d[1]), lo_pair(d[2], d[3])};
                    *reinterpret_cast<u32x2*>(row + 1 * XP) = u32x2{hi_pair(d[0], d[1]), hi_pair(d[2], d[3])};
                }
            }
        }
        if (dr0 < BR) {   // ---- dy: (R0 + dr0, C0 - 1 + 4 du + p), p = 0..4 -> pair-word planes, zero outside
            const int gx0 = C0 - 1 + 4 * du, gy = R0 + dr0;
            const bool all_in = gx0 >= 0 && gx0 + 4 < ow, any_in = gx0 + 4 >= 0 && gx0 < ow;
            const bool row_ok = gy < oh;
            const _Float16* src = gb + (size_t)min(gy, oh - 1) * ow * CO;
            uint32_t* row = ds + dr0 * DRS + 4 * du;
            if constexpr (CO == 4) {
                uint32_t v[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) v[q] = 0u;
                const uint32_t* sw = reinterpret_cast<const uint32_t*>(src);
                if (row_ok && all_in) {
                    const uint4 q0 = *reinterpret_cast<const uint4*>(sw + 2 * gx0);
                    const uint4 q1 = *reinterpret_cast<const uint4*>(sw + 2 * gx0 + 4);
                    const uint2 q2 = *reinterpret_cast<const uint2*>(sw + 2 * gx0 + 8);
                    v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
                    v[8] = q2.x, v[9] = q2.y;
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)ow) v[2 * p] = sw[2 * (gx0 + p)], v[2 * p + 1] = sw[2 * (gx0 + p) + 1];
                }
                *reinterpret_cast<uint4*>(row + 0 * DP) = uint4{lo_pair(v[0], v[2]), lo_pair(v[2], v[4]), lo_pair(v[4], v[6]), lo_pair(v[6], v[8])};
                *reinterpret_cast<uint4*>(row + 1 * DP) = uint4{hi_pair(v[0], v[2]), hi_pair(v[2], v[4]), hi_pair(v[4], v[6]), hi_pair(v[6], v[8])};
                *reinterpret_cast<uint4*>(row + 2 * DP) = uint4{lo_pair(v[1], v[3]), lo_pair(v[3], v[5]), lo_pair(v[5], v[7]), lo_pair(v[7], v[9])};
                *reinterpret_cast<uint4*>(row + 3 * DP) = uint4{hi_pair(v[1], v[3]), hi_pair(v[3], v[5]), hi_pair(v[5], v[7]), hi_pair(v[7], v[9])};
            } else {
                uint32_t g[5] = {0u, 0u, 0u, 0u, 0u};
                if (row_ok && all_in) {
                    const uint2 q = *reinterpret_cast<const uint2*>(src + gx0);
                    g[0] = q.x & 0xFFFFu, g[1] = q.x >> 16, g[2] = q.y & 0xFFFFu, g[3] = q.y >> 16;
                    g[4] = __builtin_bit_cast(unsigned short, src[gx0 + 4]);
                } else if (row_ok && any_in) {
#pragma unroll
                    for (int p = 0; p < 5; ++p)
                        if ((unsigned)(gx0 + p) < (unsigned)ow) g[p] = __builtin_bit_cast(unsigned short, src[gx0 + p]);
                }
                *reinterpret_cast<uint4*>(row) = uint4{g[0] | (g[1] << 16), g[1] | (g[2] << 16), g[2] | (g[3] << 16), g[3] | (g[4] << 16)};
            }
        }
        __syncthreads();
        // ---- wave wv: position rows wv, wv + 4; x rows of position row r and tap row ty: 2 r + ty
#pragma unroll
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const f16x4 b = read_words(ds + b0 + r * DRS + c0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const f16x4 a = read8(xs + (a_ones[tt] ? 2 * CI * XP : a_off[tt] + 2 * r * XRS + c0));
                    acc[tt] = mfma16(a, b, acc[tt]);
                }
            }
        }
    }
    // ---- block reduction: lane (n = (co, sx), kq) holds rows m = 16 t + 4 kq + i = (parity, ty, ci) | ones
    for (int i = tid; i < 4 * G::NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    {
        const int sx = CO == 4 ? (n & 3) : n, co = CO == 4 ? (n >> 2) : 0;
        const bool col_ok = CO == 4 ? sx < 3 : n < 3;
        if (col_ok) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * tt + 4 * kq + i;
                    if (m < 10 * CI) {
                        const int ci = m % CI, pt = m / CI, ty = pt % 5, par = pt / 5;
                        const int tx = 2 * (2 - sx) + par;           // s = 1 - sx; even plane: 2 (s + 1), odd: 2 (s + 1) + 1
                        if (tx < 5) red[wv][((ty * 5 + tx) * CI + ci) * CO + co] = acc[tt][i];
                    } else if (m == 10 * CI && sx == 0) {
                        red[wv][25 * CI * CO + co] = acc[tt][i];
                    }
                }
        }
    }
    __syncthreads();
    for (int i = tid; i < G::NV; i += 256)
        partial[(size_t)blockIdx.x * G::NV + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

template <typename K>
int resident_blocks(uocr_ctx* ctx, K kernel, int* cache) {
    if (*cache == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess || nb < 1) nb = 1;
        *cache = nb;
    }
    return *cache;
}

}  // namespace

bool uocr_conv_wgrad_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d) {
    return UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_fast && ctx->opt_h16 && d.kh == 5 && d.kw == 5 && d.sh == 1 &&
           d.sw == 1 && d.ph == 2 && d.pw == 2 && d.oh == d.h && d.ow == d.w &&
           ((d.cin == 4 && d.cout == 2) || (d.cin == 1 && d.cout == 1)) && (long)d.h * d.w * 4 < (1l << 31);
}

int uocr_conv_wgrad_h16(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                        double pad_value, int use_bias, int accumulate) {
    static int cache42 = 0, cache11 = 0;
    const bool one = d.cin == 1;
    const int nv = one ? e11::NV : e42::NV;
    const int br = one ? e11::BR : e42::BR;
    const int tiles_x = (d.w + 4 + e42::BC - 1) / e42::BC, tiles_y = (d.h + br - 1) / br;   // (both: 64 columns)
    const long ntiles = (long)d.n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31));
    const long cap = (long)ctx->cu_count * (one ? resident_blocks(ctx, wgrad_h16_e11_kernel, &cache11)
                                                : resident_blocks(ctx, wgrad_h16_e42_kernel, &cache42));
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, (size_t)grid * nv * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    hipLaunchKernelGGL(one ? wgrad_h16_e11_kernel : wgrad_h16_e42_kernel, dim3(grid), dim3(256), 0, ctx->stream,
                       (const _Float16*)x, (const _Float16*)dy, partial, d.h, d.w, tiles_x, tiles_y, (int)ntiles,
                       (float)pad_value);
    UOCR_LAUNCH_CHECK(ctx);
    if (h16_finish_deferred(ctx, partial, nv, one ? 25 : 200, (float*)dw, (float*)db, grid, use_bias, accumulate,
                            (float)uocr_grad_unscale(dtype)))
        return UOCR_OK;
    hipLaunchKernelGGL(wgrad_h16_finish, dim3(nv), dim3(256), 0, ctx->stream, (const float*)partial, nv,
                       one ? 25 : 200, (float*)dw, (float*)db, grid, use_bias, accumulate,
                       (float)uocr_grad_unscale(dtype));
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

// partial rows as conv_up.hip's producers write them: dw and db themselves (up_phase.h: UP4_NOUT / UP1_NOUT floats)
int uocr_upconv_wgrad_h16(uocr_ctx* ctx, const void* x_low, const void* dy, float* partial, size_t partial_floats,
                          int n, int hl, int wl, int ch, int* nblocks) {
    if (ch == 1) {
        static int cache1 = 0;
        const int tiles_x = (wl + up1::BC - 1) / up1::BC, tiles_y = (hl + up1::BR - 1) / up1::BR;
        const long ntiles = (long)n * tiles_y * tiles_x;
        UOCR_REQUIRE(ctx, ntiles < (1l << 31) && (long)hl * wl * 4 < (1l << 31));
        const long cap = (long)ctx->cu_count * resident_blocks(ctx, wgrad_h16_up1_kernel, &cache1);
        const int grid = (int)(ntiles < cap ? ntiles : cap);
        UOCR_REQUIRE(ctx, (size_t)grid * UP1_NOUT <= partial_floats);
        hipLaunchKernelGGL(wgrad_h16_up1_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const _Float16*)x_low,
                           (const _Float16*)dy, partial, hl, wl, tiles_x, tiles_y, (int)ntiles);
        UOCR_LAUNCH_CHECK(ctx);
        *nblocks = grid;
        return UOCR_OK;
    }
    static int cache = 0;
    const int tiles_x = (wl + up::BC - 1) / up::BC, tiles_y = (hl + up::BR - 1) / up::BR;
    const long ntiles = (long)n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31) && (long)hl * wl * 16 < (1l << 31));
    const long cap = (long)ctx->cu_count * resident_blocks(ctx, wgrad_h16_up_kernel, &cache);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    UOCR_REQUIRE(ctx, (size_t)grid * UP4_NOUT <= partial_floats);
    hipLaunchKernelGGL(wgrad_h16_up_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const _Float16*)x_low,
                       (const _Float16*)dy, partial, hl, wl, tiles_x, tiles_y, (int)ntiles);
    UOCR_LAUNCH_CHECK(ctx);
    *nblocks = grid;
    return UOCR_OK;
}

// dw / db of the 5x5 / stride 2 / padding 2 encoder convs (4 -> 4, 1 -> 4, 1 -> 1)
bool uocr_conv_wgrad_s2_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d) {
    return UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_fast && ctx->opt_h16 && d.kh == 5 && d.kw == 5 && d.sh == 2 &&
           d.sw == 2 && d.ph == 2 && d.pw == 2 && d.oh == (d.h + 1) / 2 && d.ow == (d.w + 1) / 2 &&
           ((d.cin == 4 && d.cout == 4) || (d.cin == 1 && (d.cout == 4 || d.cout == 1))) &&
           (long)d.h * d.w * d.cin < (1l << 31);
}

namespace {
template <int CI, int CO>
int launch_s2(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
              double pad_value, int use_bias, int accumulate) {
    using G = S2<CI, CO>;
    static int cache = 0;
    const int tiles_x = (d.ow + 2 + G::BC - 1) / G::BC, tiles_y = (d.oh + G::BR - 1) / G::BR;   // Q runs over [-1, ow]
    const long ntiles = (long)d.n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31));
    const long cap = (long)ctx->cu_count * resident_blocks(ctx, wgrad_h16_s2_kernel<CI, CO>, &cache);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, (size_t)grid * G::NV * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    hipLaunchKernelGGL((wgrad_h16_s2_kernel<CI, CO>), dim3(grid), dim3(256), 0, ctx->stream, (const _Float16*)x,
                       (const _Float16*)dy, partial, d.h, d.w, d.oh, d.ow, tiles_x, tiles_y, (int)ntiles, (float)pad_value);
    UOCR_LAUNCH_CHECK(ctx);
    if (h16_finish_deferred(ctx, partial, G::NV, 25 * CI * CO, (float*)dw, (float*)db, grid, use_bias, accumulate,
                            (float)uocr_grad_unscale(dtype)))
        return UOCR_OK;
    hipLaunchKernelGGL(wgrad_h16_finish, dim3(G::NV), dim3(256), 0, ctx->stream, (const float*)partial, G::NV,
                       25 * CI * CO, (float*)dw, (float*)db, grid, use_bias, accumulate, (float)uocr_grad_unscale(dtype));
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}
}  // namespace

int uocr_conv_wgrad_s2_h16(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                           double pad_value, int use_bias, int accumulate) {
    if (d.cin == 4) return launch_s2<4, 4>(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (d.cout == 4) return launch_s2<1, 4>(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    return launch_s2<1, 1>(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
}
