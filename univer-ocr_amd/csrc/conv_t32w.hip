// Weight gradients of the page nets' small-channel 5x5 convolutions in float32 on v_mfma_f32_16x16x4_f32: the
// float32 form of conv_h16w.hip (reference: nn/layers/convolutional.py:101-145).
//
// dw contracts over POSITIONS: the K index of the MFMA runs along image columns for a fixed channel, so both
// operands are staged as channel PLANES in LDS (channels-last pixels are de-interleaved while they are written:
// four pixels per thread, the four values of a plane are four registers of four different loads -- no ALU work).
// The tap's column shift lives on the dy operand (N = (co, sx)); the x operand is read at a fixed, 16-byte aligned
// offset.  The 16 positions of an MFMA group are assigned to (k-step g, k-group kq) as 4*kq + g, so a lane's four
// k-steps are four CONSECUTIVE floats: one ds_read_b128 for x, two ds_read2_b32 for the shifted dy.
//   stride 1 (5x5, padding 2):  dw[ty][4 - sx][ci][co] = sum xpad[row + ty - 2][col + 2][ci] * dy[row][col + sx][co]
//        M = (ty, ci) (+ a row of ones -> db): 4 -> 2: 2 M tiles x 4 k-steps = 8 MFMAs per 16 positions, 1 -> 1: 4
//   stride 2 (5x5, padding 2):  x as EVEN / ODD column planes, E[j] = x[2j], O[j] = x[2j+1]; tap column 2e reads
//        E[X + e - 1], 2o + 1 reads O[X + o - 1]; with Q = X + s: M = (parity, ty, ci) reads its plane at Q, N = (co,
//        sx = 1 - s) reads dy[Q + sx - 1]: 4 -> 4: 12 MFMAs per 16 positions, 1 -> 4 / 1 -> 1: 4
// Products and sums are exact float32 FMA chains (float64 across blocks).  Measured at 32 x 256 x 512 inside their
// nets: stride 2 4 -> 4 17.6 us (vector kernel 23.0), 1 -> 4 16.1 (18.5), 1 -> 1 13.2 (14.7); stride 1 4 -> 2 53.7 (49.8),
// 1 -> 1 29.3 (32.7) -- and in the four-net step the stride-2 set is neutral (37.0 k images/s either way), both sets
// together cost 2.5 % (36.1 k): float32 MFMAs do not overlap the vector work of the kernels sharing the CU (DESIGN.md
// section 5a).  Both sets are therefore OFF by default ("t32" option bits 64 / 128) and kept selectable and tested.
// The vector kernels they would replace (conv_fast.hip) give every thread all taps of one pixel (25-400 FMAs and as
// many shuffles / LDS reads per pixel) or make five passes over x and dy.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "conv_dims.h"

// Off by default and slower in the page step: built only with UOCR_BUILD_EXPERIMENTS=1 ./build.sh
// (-DUOCR_EXPERIMENTS); the default library answers "not eligible".
#ifdef UOCR_EXPERIMENTS

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// four consecutive floats at a 4-byte aligned LDS address
__device__ __forceinline__ float4 read4_unaligned(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }

// NP pixels of C channels from gx0 on of an image row: zero / pad outside [0, wd); v[p * C + c]
template <int C, int NP>
__device__ __forceinline__ void load_pixels(float (&v)[NP * C], const float* __restrict__ src, int gx0, int wd, bool row_ok,
                                            float fill) {
#pragma unroll
    for (int i = 0; i < NP * C; ++i) v[i] = fill;
    if (!row_ok) return;
    if (gx0 >= 0 && gx0 + NP <= wd) {
#pragma unroll
        for (int q = 0; q < NP * C / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(src + (size_t)gx0 * C + 4 * q);      // (4-byte aligned at least)
            v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
        }
    } else if (gx0 + NP > 0 && gx0 < wd) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if ((unsigned)(gx0 + p) < (unsigned)wd) {
#pragma unroll
                for (int c = 0; c < C; ++c) v[p * C + c] = src[(size_t)(gx0 + p) * C + c];
            }
    }
}

// block a: out[a] (+)= sum over blocks of partial[blk][a]; a < ndw -> dw, else db
__global__ __launch_bounds__(256) void wgrad_t32_finish(const float* __restrict__ partial, int nv, int ndw,
                                                        float* __restrict__ dw, float* __restrict__ db, int nblocks,
                                                        int use_bias, int accumulate) {
    __shared__ double smem[16];
    const int a = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += (double)partial[(size_t)i * nv + a];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst = a < ndw ? dw + a : db + (a - ndw);
    if (a >= ndw && !use_bias) s = 0.0;
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// ------------------------------------------------------------------------------------------------------------
// stride 2
// ------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
struct S2 {
    static constexpr int BR = 8, BC = 64;            // positions (Y, Q) per tile
    static constexpr int XR = 2 * BR + 3;            // x rows of a tile
    static constexpr int XRS = 72;                   // plane row stride (floats): 68 used
    static constexpr int XP = XR * XRS + 8;          // plane stride
    static constexpr int DRS = 72;                   // dy plane row stride
    static constexpr int DP = BR * DRS + 8;
    static constexpr int NT = (10 * CI + 1 + 15) / 16;   // M tiles (the ones row included)
    static constexpr int NV = 25 * CI * CO + CO;
    static constexpr int XU = 17;                    // x staging units (8 pixels) per row: 136 >= 132 pixels
};

template <int CI, int CO>
__global__ __launch_bounds__(256) void wgrad_t32_s2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ partial, int h, int wd, int oh, int ow,
                                                           int tiles_x, int tiles_y, int ntiles, float pad) {
    using G = S2<CI, CO>;
    constexpr int BR = G::BR, BC = G::BC, XR = G::XR, XRS = G::XRS, XP = G::XP, DRS = G::DRS, DP = G::DP, NT = G::NT;
    __shared__ __attribute__((aligned(16))) float xs[2 * CI * XP + 4];     // [(ci, parity)][row][j] (+ the ones)
    __shared__ __attribute__((aligned(16))) float ds[CO * DP];             // [co][row][column]
    __shared__ float red[4][G::NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) xs[2 * CI * XP + tid] = 1.f;
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
    const int xu = tid % G::XU, xr0 = tid / G::XU;       // x unit = 8 pixels, 17 units per row, 15 rows per pass
    const int du = tid % 18, dr0 = tid / 18;             // dy unit = 4 pixels, 18 units per row

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC - 1, R0 = trow * BR;   // plane index Q of tile column 0; first position row
        const float* xb = x + (size_t)img * h * wd * CI;
        const float* gb = dy + (size_t)img * oh * ow * CO;
        __syncthreads();                                 // the previous tile's reads are over
        // ---- x: image row 2 R0 - 2 + r, pixels 2 C0 + 8 xu + p (p = 0..7) -> even / odd planes at j = 4 xu + p / 2
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = xr0 + 15 * k, gy = 2 * R0 - 2 + r;
            if (r >= XR || xr0 >= 15) continue;
            float v[8 * CI];
            load_pixels<CI, 8>(v, xb + (size_t)min(max(gy, 0), h - 1) * wd * CI, 2 * C0 + 8 * xu, wd,
                               (unsigned)gy < (unsigned)h, pad);
            float* row = xs + r * XRS + 4 * xu;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int par = 0; par < 2; ++par)
                    *reinterpret_cast<float4*>(row + (ci * 2 + par) * XP) =
                        make_float4(v[par * CI + ci], v[(par + 2) * CI + ci], v[(par + 4) * CI + ci], v[(par + 6) * CI + ci]);
        }
        if (dr0 < BR) {   // ---- dy: (R0 + dr0, C0 - 1 + 4 du + p), p = 0..3 -> planes, zero outside
            const int gy = R0 + dr0;
            float v[4 * CO];
            load_pixels<CO, 4>(v, gb + (size_t)min(gy, oh - 1) * ow * CO, C0 - 1 + 4 * du, ow, gy < oh, 0.f);
            float* row = ds + dr0 * DRS + 4 * du;
#pragma unroll
            for (int co = 0; co < CO; ++co)
                *reinterpret_cast<float4*>(row + co * DP) = make_float4(v[co], v[CO + co], v[2 * CO + co], v[3 * CO + co]);
        }
        __syncthreads();
        // ---- wave wv: position rows wv, wv + 4; x rows of position row r and tap row ty: 2 r + ty
#pragma unroll
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const float4 b = read4_unaligned(ds + b0 + r * DRS + c0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const float4 a = *reinterpret_cast<const float4*>(xs + (a_ones[tt] ? 2 * CI * XP : a_off[tt] + 2 * r * XRS + c0));
                    acc[tt] = mfma4(a.x, b.x, acc[tt]);
                    acc[tt] = mfma4(a.y, b.y, acc[tt]);
                    acc[tt] = mfma4(a.z, b.z, acc[tt]);
                    acc[tt] = mfma4(a.w, b.w, acc[tt]);
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

// ------------------------------------------------------------------------------------------------------------
// stride 1:  M = (ty, ci) (5 * CI rows + the ones row), N = (co, sx) with sx = 0..4 (8 slots per co)
// ------------------------------------------------------------------------------------------------------------
template <int CI, int CO>
struct E1 {
    static constexpr int BR = 16, BC = 64;
    static constexpr int XR = BR + 4, XRS = 72, XP = XR * XRS + 8;
    static constexpr int DRS = 72, DP = BR * DRS + 8;
    static constexpr int NT = (5 * CI + 1 + 15) / 16;
    static constexpr int NV = 25 * CI * CO + CO;
    static_assert(CO * 8 <= 16, "N = (co, sx) with 8 slots per channel");
};

template <int CI, int CO>
__global__ __launch_bounds__(256) void wgrad_t32_e_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ partial, int h, int wd, int tiles_x,
                                                          int tiles_y, int ntiles, float pad) {
    using G = E1<CI, CO>;
    constexpr int BR = G::BR, BC = G::BC, XR = G::XR, XRS = G::XRS, XP = G::XP, DRS = G::DRS, DP = G::DP, NT = G::NT;
    __shared__ __attribute__((aligned(16))) float xs[CI * XP + 4];          // [ci][row][col] (+ the ones)
    __shared__ __attribute__((aligned(16))) float ds[CO * DP];              // [co][row][col]
    __shared__ float red[4][G::NV];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 4) xs[CI * XP + tid] = 1.f;
    int a_off[NT];
    bool a_ones[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = 16 * t + n, mm = min(m, 5 * CI - 1);
        a_off[t] = (mm % CI) * XP + (mm / CI) * XRS + 4 * kq;   // row m = ty * CI + ci
        a_ones[t] = m == 5 * CI;
    }
    const int b_sx = min(n & 7, 4), b_co = min(n >> 3, CO - 1);
    const int b0 = b_co * DP + 4 * kq + b_sx;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xu = tid & 15, xr0 = tid >> 4;             // x unit = 4 pixels, 16 units per row, 16 rows per pass
    const int du = tid % 18, dr0 = tid / 18;             // dy unit = 4 pixels, 18 units per row, 14 rows per pass

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int C0 = strip * BC - 4, R0 = trow * BR;   // position col = (dy column) - sx starts at -4
        const float* xb = x + (size_t)img * h * wd * CI;
        const float* gb = dy + (size_t)img * h * wd * CO;
        __syncthreads();
        // ---- x: image (R0 - 2 + r, C0 + 2 + 4 xu + p) -> planes
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = xr0 + 16 * k, gy = R0 - 2 + r;
            if (r >= XR) continue;
            float v[4 * CI];
            load_pixels<CI, 4>(v, xb + (size_t)min(max(gy, 0), h - 1) * wd * CI, C0 + 2 + 4 * xu, wd,
                               (unsigned)gy < (unsigned)h, pad);
            float* row = xs + r * XRS + 4 * xu;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
                *reinterpret_cast<float4*>(row + ci * XP) = make_float4(v[ci], v[CI + ci], v[2 * CI + ci], v[3 * CI + ci]);
        }
        // ---- dy: image (R0 + r, C0 + 4 du + p) -> planes, zero outside
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = dr0 + 14 * k, gy = R0 + r;
            if (r >= BR || dr0 >= 14) continue;
            float v[4 * CO];
            load_pixels<CO, 4>(v, gb + (size_t)min(gy, h - 1) * wd * CO, C0 + 4 * du, wd, gy < h, 0.f);
            float* row = ds + r * DRS + 4 * du;
#pragma unroll
            for (int co = 0; co < CO; ++co)
                *reinterpret_cast<float4*>(row + co * DP) = make_float4(v[co], v[CO + co], v[2 * CO + co], v[3 * CO + co]);
        }
        __syncthreads();
        for (int r = wv; r < BR; r += 4) {
#pragma unroll
            for (int c0 = 0; c0 < BC; c0 += 16) {
                const float4 b = read4_unaligned(ds + b0 + r * DRS + c0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const float4 a = *reinterpret_cast<const float4*>(xs + (a_ones[tt] ? CI * XP : a_off[tt] + r * XRS + c0));
                    acc[tt] = mfma4(a.x, b.x, acc[tt]);
                    acc[tt] = mfma4(a.y, b.y, acc[tt]);
                    acc[tt] = mfma4(a.z, b.z, acc[tt]);
                    acc[tt] = mfma4(a.w, b.w, acc[tt]);
                }
            }
        }
    }
    // ---- lane (n = (co, sx), kq) holds rows m = 16 t + 4 kq + i = (ty, ci) | ones: dw[ty][4 - sx][ci][co]
    for (int i = tid; i < 4 * G::NV; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    if ((n & 7) <= 4 && (n >> 3) < CO) {
        const int sx = n & 7, co = n >> 3;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * tt + 4 * kq + i;
                if (m < 5 * CI) red[wv][(((m / CI) * 5 + 4 - sx) * CI + m % CI) * CO + co] = acc[tt][i];
                else if (m == 5 * CI && sx == 0) red[wv][25 * CI * CO + co] = acc[tt][i];
            }
    }
    __syncthreads();
    for (int i = tid; i < G::NV; i += 256)
        partial[(size_t)blockIdx.x * G::NV + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

template <typename K>
int resident_blocks(K kernel, int* cache) {
    if (*cache == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0) != hipSuccess || nb < 1) nb = 1;
        *cache = nb;
    }
    return *cache;
}

template <int CI, int CO>
int launch_s2(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d, double pad_value,
              int use_bias, int accumulate) {
    using G = S2<CI, CO>;
    static int cache = 0;
    const int tiles_x = (d.ow + 2 + G::BC - 1) / G::BC, tiles_y = (d.oh + G::BR - 1) / G::BR;   // Q runs over [-1, ow]
    const long ntiles = (long)d.n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31));
    const long cap = (long)ctx->cu_count * resident_blocks(wgrad_t32_s2_kernel<CI, CO>, &cache);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    int rc = uocr_need_workspace(ctx, (size_t)grid * G::NV * sizeof(float));
    if (rc != UOCR_OK) return rc;
    float* partial = (float*)ctx->workspace;
    hipLaunchKernelGGL((wgrad_t32_s2_kernel<CI, CO>), dim3(grid), dim3(256), 0, ctx->stream, (const float*)x,
                       (const float*)dy, partial, d.h, d.w, d.oh, d.ow, tiles_x, tiles_y, (int)ntiles, (float)pad_value);
    UOCR_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(wgrad_t32_finish, dim3(G::NV), dim3(256), 0, ctx->stream, (const float*)partial, G::NV,
                       25 * CI * CO, (float*)dw, (float*)db, grid, use_bias, accumulate);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

template <int CI, int CO>
int launch_e(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d, double pad_value,
             int use_bias, int accumulate) {
    using G = E1<CI, CO>;
    static int cache = 0;
    const int tiles_x = (d.w + 4 + G::BC - 1) / G::BC, tiles_y = (d.h + G::BR - 1) / G::BR;     // col runs over [-4, w)
    const long ntiles = (long)d.n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31));
    const long cap = (long)ctx->cu_count * resident_blocks(wgrad_t32_e_kernel<CI, CO>, &cache);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    int rc = uocr_need_workspace(ctx, (size_t)grid * G::NV * sizeof(float));
    if (rc != UOCR_OK) return rc;
    float* partial = (float*)ctx->workspace;
    hipLaunchKernelGGL((wgrad_t32_e_kernel<CI, CO>), dim3(grid), dim3(256), 0, ctx->stream, (const float*)x,
                       (const float*)dy, partial, d.h, d.w, tiles_x, tiles_y, (int)ntiles, (float)pad_value);
    UOCR_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(wgrad_t32_finish, dim3(G::NV), dim3(256), 0, ctx->stream, (const float*)partial, G::NV,
                       25 * CI * CO, (float*)dw, (float*)db, grid, use_bias, accumulate);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

inline bool is5x5(const ConvDims& d, int s) {
    return d.kh == 5 && d.kw == 5 && d.sh == s && d.sw == s && d.ph == 2 && d.pw == 2 &&
           d.oh == (s == 1 ? d.h : (d.h + 1) / 2) && d.ow == (s == 1 ? d.w : (d.w + 1) / 2);
}

}  // namespace

// "t32" option bits: 64 = stride-1 weight gradients (4 -> 2, 1 -> 1), 128 = stride-2 (4 -> 4, 1 -> 4, 1 -> 1)
bool uocr_conv_wgrad_t32_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d) {
    if (dtype != UOCR_F32 || !ctx->opt_fast || (long)d.h * d.w * d.cin >= (1l << 31)) return false;
    if ((ctx->opt_t32 & 64) && is5x5(d, 1)) return (d.cin == 4 && d.cout == 2) || (d.cin == 1 && d.cout == 1);
    if ((ctx->opt_t32 & 128) && is5x5(d, 2))
        return (d.cin == 4 && d.cout == 4) || (d.cin == 1 && (d.cout == 4 || d.cout == 1));
    return false;
}

int uocr_conv_wgrad_t32(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                        double pad_value, int use_bias, int accumulate) {
    if (d.sh == 1) {
        if (d.cin == 4) return launch_e<4, 2>(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
        return launch_e<1, 1>(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    }
    if (d.cin == 4) return launch_s2<4, 4>(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (d.cout == 4) return launch_s2<1, 4>(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    return launch_s2<1, 1>(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
}

#else
bool uocr_conv_wgrad_t32_eligible(uocr_ctx*, int, const ConvDims&) { return false; }
int uocr_conv_wgrad_t32(uocr_ctx* ctx, const void*, const void*, void*, void*, const ConvDims&, double, int, int) {
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_t32w: library built without UOCR_BUILD_EXPERIMENTS");
}
#endif
