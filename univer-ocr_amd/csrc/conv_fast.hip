// Shape-specialised direct convolution kernels (float32) for the skinny convs of my_model
// (reference shapes: my_model/model.py:108-304; semantics: nn/layers/convolutional.py:62-145).
//
// Arithmetic intensity of these layers is 2.6-25 flop/byte: they are HBM-bound, so
// MFMA is the wrong tool (N = 1..4 output channels would waste >90 % of a 16x16 tile).  Design:
//   * kernel size / channels / stride are compile-time: taps fully unrolled, weights come from the
//     scalar cache (uniform s_load of a const __restrict__ pointer) straight into FMA operands;
//   * one thread per pixel, lanes along x: every load/store of a wave is contiguous in NHWC;
//     the 3x3 / 5x5 re-reads of neighbouring pixels are served by L1/L2, HBM sees x once;
//   * dw/db: per-thread register accumulators over a band of output rows, then a reduce-scatter
//     butterfly across the 64 lanes (NP shuffles instead of 6*NP for a plain all-reduce), LDS
//     across the 4 waves, one float32 partial per block; a second kernel sums the block partials
//     in float64 in a fixed order (deterministic, no atomics).
#include "conv_dims.h"

namespace {

template <int N>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<2> {
    using type = float2;
};
template <>
struct VecT<4> {
    using type = float4;
};

// load / store C consecutive floats (C = 1, 2, 4 or a multiple of 4) with the widest vectors
template <int C>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&v)[C]) {
    if constexpr (C == 1) {
        v[0] = p[0];
    } else if constexpr (C == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x;
        v[1] = t.y;
    } else {
        static_assert(C % 4 == 0, "channel count must be 1, 2 or a multiple of 4");
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const float4 t = reinterpret_cast<const float4*>(p)[q];
            v[4 * q] = t.x;
            v[4 * q + 1] = t.y;
            v[4 * q + 2] = t.z;
            v[4 * q + 3] = t.w;
        }
    }
}

template <int C>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&v)[C]) {
    if constexpr (C == 1) {
        p[0] = v[0];
    } else if constexpr (C == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
            reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= 0.f ? 1.f : 0.f);
        case UOCR_ACT_LEAKY: return v * ((v >= 0.f ? 1.f : 0.f) + alpha * (v < 0.f ? 1.f : 0.f));
        case UOCR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

struct FastDims {
    int n, h, w, oh, ow, ph, pw;
};

// ---------------------------------------------------------------------------------------------
// forward: block (64, 4); a thread computes COB output channels of one pixel and NQ = COUT / COB
// adjacent lanes share a pixel, so one wave stores 64 * COB contiguous floats per instruction
// (with one lane per pixel and COUT = 16 every store instruction would touch 64 B per lane at a
// 64 B stride: 4 partial-line store instructions per wave instead of 4 full ones)
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int COUT, int SH, int SW, int COB>
__global__ __launch_bounds__(256) void conv_fwd_fast(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     FastDims d, float pad, int use_bias, int act, float alpha) {
    constexpr int NQ = COUT / COB;     // lanes per pixel
    constexpr int PXW = 64 / NQ;       // pixels per wave row
    const int ox = blockIdx.x * PXW + threadIdx.x / NQ;
    const int oy = blockIdx.y * 4 + threadIdx.y;
    const int oc0 = (threadIdx.x % NQ) * COB;
    const int b = blockIdx.z;
    if (ox >= d.ow || oy >= d.oh) return;
    float acc[COB];
#pragma unroll
    for (int o = 0; o < COB; ++o) acc[o] = 0.f;
    const int iy0 = oy * SH - d.ph, ix0 = ox * SW - d.pw;
    const float* xb = x + (size_t)b * d.h * d.w * CIN;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int iy = iy0 + ky;
        const bool row_ok = iy >= 0 && iy < d.h;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int ix = ix0 + kx;
            float xv[CIN];
            if (row_ok && ix >= 0 && ix < d.w) {
                load_vec<CIN>(xb + ((size_t)iy * d.w + ix) * CIN, xv);
            } else {
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[c] = pad;
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int o = 0; o < COB; ++o) acc[o] += xv[c] * w[((ky * KW + kx) * CIN + c) * COUT + oc0 + o];
        }
    }
    float out[COB];
#pragma unroll
    for (int o = 0; o < COB; ++o) {
        float v = acc[o];
        if (use_bias) v += bias[oc0 + o];
        out[o] = act_apply(v, act, alpha);
    }
    store_vec<COB>(y + (((size_t)b * d.oh + oy) * d.ow + ox) * COUT + oc0, out);
}

// ---------------------------------------------------------------------------------------------
// backward data: block (64, 4) = 64 x 4 INPUT pixels; all CIN channels per thread
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int COUT, int SH, int SW>
__global__ __launch_bounds__(256) void conv_dgrad_fast(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, FastDims d) {
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (ix >= d.w || iy >= d.h) return;
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
    const float* gb = dy + (size_t)b * d.oh * d.ow * COUT;
#pragma unroll
    for (int ky = KH - 1; ky >= 0; --ky) {
        const int ty = iy + d.ph - ky;
        const int gy = SH == 1 ? ty : ty / SH;
        const bool row_ok = ty >= 0 && gy < d.oh && (SH == 1 || gy * SH == ty);
#pragma unroll
        for (int kx = KW - 1; kx >= 0; --kx) {
            const int tx = ix + d.pw - kx;
            const int gx = SW == 1 ? tx : tx / SW;
            if (row_ok && tx >= 0 && gx < d.ow && (SW == 1 || gx * SW == tx)) {
                float g[COUT];
                load_vec<COUT>(gb + ((size_t)gy * d.ow + gx) * COUT, g);
#pragma unroll
                for (int c = 0; c < CIN; ++c)
#pragma unroll
                    for (int o = 0; o < COUT; ++o) acc[c] += g[o] * w[((ky * KW + kx) * CIN + c) * COUT + o];
            }
        }
    }
    store_vec<CIN>(dx + (((size_t)b * d.h + iy) * d.w + ix) * CIN, acc);
}

// ---------------------------------------------------------------------------------------------
// backward weights: block (64, 4) owns output rows [row0, row0 + rows) of image b, tap rows
// [kyg*KYR, (kyg+1)*KYR) and output channels [ocg*COB, (ocg+1)*COB).
// accumulator a = ((kyl*KW + kx)*CIN + c)*COB + o ; db accumulators (tap group 0 only) follow.
// ---------------------------------------------------------------------------------------------
// One step per template instance: with the step as a runtime-looking loop variable hipcc (ROCm 7.2)
// leaves acc[] in scratch memory for NP = 64 / 128 (528 B/lane of scratch traffic per FMA).
template <int NP, int S>
__device__ __forceinline__ void reduce_scatter_step(float (&acc)[NP], int lane) {
    constexpr int bit = 32 >> S;
    constexpr int half = NP >> (S + 1);
    const bool up = lane & bit;
#pragma unroll
    for (int i = 0; i < half; ++i) {
        const float send = up ? acc[i] : acc[i + half];
        const float keep = up ? acc[i + half] : acc[i];
        acc[i] = keep + __shfl_xor(send, bit, 64);
    }
}

template <int NP>
__device__ __forceinline__ void lane_reduce_scatter(float (&acc)[NP], int lane) {
    // after the 6 steps lane L holds, in acc[0 .. NP/64), the full 64-lane sums of the original
    // indices base(L) + r with base(L) = sum_s bit_{5-s}(L) * NP / 2^(s+1)
    reduce_scatter_step<NP, 0>(acc, lane);
    reduce_scatter_step<NP, 1>(acc, lane);
    reduce_scatter_step<NP, 2>(acc, lane);
    reduce_scatter_step<NP, 3>(acc, lane);
    reduce_scatter_step<NP, 4>(acc, lane);
    reduce_scatter_step<NP, 5>(acc, lane);
}

template <int KH, int KW, int CIN, int COUT, int SH, int SW, int KYR, int COB>
struct WgradCfg {
    static constexpr int NW = KYR * KW * CIN * COB;     // weight accumulators per thread
    static constexpr int NACC = NW + COB;               // + db
    static constexpr int NP = ((NACC + 63) / 64) * 64;  // padded to a multiple of the wave size
    static constexpr int KYG = KH / KYR;                // tap-row groups
    static constexpr int OCG = COUT / COB;              // output-channel groups
};

template <int KH, int KW, int CIN, int COUT, int SH, int SW, int KYR, int COB>
__global__ __launch_bounds__(256) void conv_wgrad_fast(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ partial, FastDims d, float pad,
                                                       int rows_per_block, int nbands) {
    using C = WgradCfg<KH, KW, CIN, COUT, SH, SW, KYR, COB>;
    __shared__ float red[4][C::NP];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int band = blockIdx.x % nbands, b = blockIdx.x / nbands;
    const int kyg = blockIdx.y / C::OCG, ocg = blockIdx.y % C::OCG;
    const int ky0 = kyg * KYR, oc0 = ocg * COB;
    const int row0 = band * rows_per_block;
    const int row1 = min(d.oh, row0 + rows_per_block);
    float acc[C::NP];
#pragma unroll
    for (int a = 0; a < C::NP; ++a) acc[a] = 0.f;
    const float* xb = x + (size_t)b * d.h * d.w * CIN;
    const float* gb = dy + (size_t)b * d.oh * d.ow * COUT + oc0;
    for (int oy = row0 + wv; oy < row1; oy += 4) {
        for (int ox = lane; ox < d.ow; ox += 64) {
            float g[COB];
            load_vec<COB>(gb + ((size_t)oy * d.ow + ox) * COUT, g);
            if (kyg == 0) {
#pragma unroll
                for (int o = 0; o < COB; ++o) acc[C::NW + o] += g[o];
            }
            const int iy0 = oy * SH - d.ph + ky0, ix0 = ox * SW - d.pw;
#pragma unroll
            for (int kyl = 0; kyl < KYR; ++kyl) {
                const int iy = iy0 + kyl;
                const bool row_ok = iy >= 0 && iy < d.h;
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const int ix = ix0 + kx;
                    float xv[CIN];
                    if (row_ok && ix >= 0 && ix < d.w) {
                        load_vec<CIN>(xb + ((size_t)iy * d.w + ix) * CIN, xv);
                    } else {
#pragma unroll
                        for (int c = 0; c < CIN; ++c) xv[c] = pad;
                    }
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int o = 0; o < COB; ++o) acc[((kyl * KW + kx) * CIN + c) * COB + o] += xv[c] * g[o];
                }
            }
        }
    }
    lane_reduce_scatter<C::NP>(acc, lane);
    int base = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s)
        if (lane & (32 >> s)) base += C::NP >> (s + 1);
#pragma unroll
    for (int r = 0; r < C::NP / 64; ++r) red[wv][base + r] = acc[r];
    __syncthreads();
    const int tid = wv * 64 + lane;
    float* out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C::NP;
    for (int a = tid; a < C::NP; a += 256) out[a] = red[0][a] + red[1][a] + red[2][a] + red[3][a];
}

// block per accumulator index: sums the block partials (float64, fixed order) into dw / db
template <int KH, int KW, int CIN, int COUT, int KYR, int COB, int NW, int NP>
__global__ __launch_bounds__(256) void conv_wgrad_fast_finish(const float* __restrict__ partial, float* __restrict__ dw,
                                                              float* __restrict__ db, int nblocks, int use_bias,
                                                              int accumulate) {
    __shared__ double smem[16];
    const int a = blockIdx.x, grp = blockIdx.y;
    constexpr int OCG = COUT / COB;
    const int kyg = grp / OCG, ocg = grp % OCG;
    const float* src = partial + (size_t)grp * nblocks * NP + a;
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += (double)src[(size_t)i * NP];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst;
    if (a < NW) {
        const int o = a % COB, t = a / COB;
        const int c = t % CIN, tap = t / CIN;
        const int kx = tap % KW, kyl = tap / KW;
        dst = dw + (((size_t)(kyg * KYR + kyl) * KW + kx) * CIN + c) * COUT + ocg * COB + o;
    } else {
        if (kyg != 0) return;
        dst = db + ocg * COB + (a - NW);
        if (!use_bias) s = 0.0;
    }
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int KH, int KW, int CIN, int COUT, int SH, int SW, int COB, int KYR, int WCOB>
struct FastConv {
    static bool match(const ConvDims& d) {
        return d.kh == KH && d.kw == KW && d.cin == CIN && d.cout == COUT && d.sh == SH && d.sw == SW;
    }
    static FastDims dims(const ConvDims& d) { return FastDims{d.n, d.h, d.w, d.oh, d.ow, d.ph, d.pw}; }

    static int fwd(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                   double pad, int use_bias, int act, double alpha) {
        constexpr int PXW = 64 / (COUT / COB);
        const dim3 grid((d.ow + PXW - 1) / PXW, (d.oh + 3) / 4, d.n), block(64, 4);
        hipLaunchKernelGGL((conv_fwd_fast<KH, KW, CIN, COUT, SH, SW, COB>), grid, block, 0, ctx->stream,
                           (const float*)x, (const float*)w, (const float*)b, (float*)y, dims(d), (float)pad, use_bias,
                           act, (float)alpha);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }

    static int dgrad(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d) {
        const dim3 grid((d.w + 63) / 64, (d.h + 3) / 4, d.n), block(64, 4);
        hipLaunchKernelGGL((conv_dgrad_fast<KH, KW, CIN, COUT, SH, SW>), grid, block, 0, ctx->stream,
                           (const float*)dy, (const float*)w, (float*)dx, dims(d));
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }

    static int wgrad(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                     double pad, int use_bias, int accumulate) {
        using C = WgradCfg<KH, KW, CIN, COUT, SH, SW, KYR, WCOB>;
        // bands of output rows: ~512 blocks per (tap group, channel group), at least 4 rows each
        int rows = (d.n * d.oh + 511) / 512;
        rows = ((rows + 3) / 4) * 4;
        if (rows > d.oh) rows = ((d.oh + 3) / 4) * 4;
        const int nbands = (d.oh + rows - 1) / rows;
        const int nblocks = nbands * d.n, ngroups = C::KYG * C::OCG;
        const size_t bytes = (size_t)nblocks * ngroups * C::NP * sizeof(float);
        int rc = uocr_need_workspace(ctx, bytes);
        if (rc) return rc;
        float* partial = (float*)ctx->workspace;
        hipLaunchKernelGGL((conv_wgrad_fast<KH, KW, CIN, COUT, SH, SW, KYR, WCOB>), dim3(nblocks, ngroups),
                           dim3(64, 4), 0, ctx->stream, (const float*)x, (const float*)dy, partial, dims(d),
                           (float)pad, rows, nbands);
        UOCR_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((conv_wgrad_fast_finish<KH, KW, CIN, COUT, KYR, WCOB, C::NW, C::NP>),
                           dim3(C::NACC, ngroups), dim3(256), 0, ctx->stream, (const float*)partial, (float*)dw,
                           (float*)db, nblocks, use_bias, accumulate);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
};

// the my_model shapes:        KH KW CIN COUT SH SW  fwd-COB  wgrad-KYR  wgrad-COB
#define UOCR_FAST_CONVS(X)                                                        \
    X(3, 3, 1, 16, 1, 1, 4, 3, 16)  /* Monochrome conv_1 */                       \
    X(3, 3, 16, 1, 1, 1, 1, 3, 1)   /* Monochrome conv_2 */                       \
    X(5, 5, 1, 1, 2, 2, 1, 5, 1)    /* Paragraph down_1/2 */                      \
    X(5, 5, 1, 1, 1, 1, 1, 5, 1)    /* Paragraph up_2, up_1, end */               \
    X(5, 5, 1, 4, 2, 2, 4, 5, 4)    /* Line down_1 */                             \
    X(5, 5, 4, 4, 2, 2, 4, 1, 4)    /* Line down_2 */                             \
    X(5, 5, 4, 4, 1, 1, 4, 1, 4)    /* Line up_2, up_1 */                         \
    X(5, 5, 4, 2, 1, 1, 2, 1, 2)    /* Line end */                                \
    X(5, 3, 1, 64, 2, 1, 4, 5, 8)   /* Char conv_1 */

}  // namespace

bool uocr_conv_fast_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, const void* p0, const void* p1,
                             const void* p2) {
    if (dtype != UOCR_F32 || !ctx->opt_fast) return false;
    if (!aligned16(p0) || !aligned16(p1) || !aligned16(p2)) return false;
#define X(KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB) \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::match(d)) return true;
    UOCR_FAST_CONVS(X)
#undef X
    return false;
}

int uocr_conv_fwd_fast(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                       double pad_value, int use_bias, int act, double act_alpha) {
#define X(KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB)                        \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::match(d))      \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::fwd(ctx, x, w, b, y, d, pad_value, use_bias, act, \
                                                                        act_alpha);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}

int uocr_conv_dgrad_fast(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d) {
#define X(KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB)                   \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::match(d)) \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::dgrad(ctx, dy, w, dx, d);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}

int uocr_conv_wgrad_fast(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                         double pad_value, int use_bias, int accumulate) {
#define X(KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB)                   \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::match(d)) \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, KYR, WCOB>::wgrad(ctx, x, dy, dw, db, d, pad_value, use_bias, \
                                                                          accumulate);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}
