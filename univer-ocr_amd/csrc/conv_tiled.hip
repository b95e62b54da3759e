// LDS-tiled direct convolution (float32) for the 4-channel, many-tap, stride-1 convs of the Line net
// (5x5, 4 -> 4 and 4 -> 2 at 128x256 and 256x512; reference shapes my_model/model.py:194-247).
//
// One pixel per thread with 25 taps re-reads every input vector 25 times through L1
// (64 B/clk/CU): the register-tiled kernels of conv_fast.hip were bound by that, at 1-2 TB/s of
// algorithmic bandwidth.  Here a block of 8 x 32 output pixels stages its (8+KH-1) x (32+KW-1)
// input window ONCE into LDS as float4 pixels (coalesced 16-B loads, 1.7x halo overhead) and every
// tap is a ds_read_b128 with consecutive lanes on consecutive 16-B slots (conflict-free,
// 256 B/clk/CU).  Weights: one tap ROW at a time in SGPRs (the tap-row loop is not unrolled, see
// conv_fast.hip).  Measured (line.up_1, 32x256x512x4): forward 72 -> 62 us.  The same structure was
// tried for dx (69 -> 77 us) and dw (117 -> 226 us: a barrier per tile and 5 tap-row blocks re-staging
// the same rows) and lost to the register-tiled kernels, so only the forward uses it.
#include "conv_dims.h"

namespace {

constexpr int TH = 8, TW = 32;     // output tile = 256 threads, one pixel each

struct TileDims {
    int n, h, w, ph, pw;           // same-size stride-1 conv: output (h, w) == input (h, w)
};

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= 0.f ? 1.f : 0.f);
        case UOCR_ACT_LEAKY: return v * ((v >= 0.f ? 1.f : 0.f) + alpha * (v < 0.f ? 1.f : 0.f));
        case UOCR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

template <int C, typename TA>
__device__ __forceinline__ void store_c(TA* p, const float (&v)[C]) {
    if constexpr (C == 4) st4(p, make_float4(v[0], v[1], v[2], v[3]));
    else if constexpr (C == 2) st2(p, make_float2(v[0], v[1]));
    else st1(p, v[0]);
}

// stage rows [y0, y0+ROWS) x cols [x0, x0+COLS) of a C-channel image (C = 2 or 4) into LDS as float4
// pixels (C = 2: .z/.w unused); outside the image -> fill
template <int C, int ROWS, int COLS, typename TA>
__device__ __forceinline__ void stage_tile(float4* __restrict__ tile, const TA* __restrict__ img, int h, int w,
                                           int y0, int x0, float fill, int tid) {
    for (int i = tid; i < ROWS * COLS; i += TH * TW) {
        const int r = i / COLS, c = i - r * COLS;
        const int y = y0 + r, x = x0 + c;
        float4 v = make_float4(fill, fill, fill, fill);
        if (y >= 0 && y < h && x >= 0 && x < w) {
            const TA* p = img + ((size_t)y * w + x) * C;
            if constexpr (C == 4) v = ld4(p);
            else { const float2 t = ld2(p); v.x = t.x; v.y = t.y; }
        }
        tile[i] = v;
    }
}

// y[p,o] = sum_{ky,kx,c} x[p + (ky,kx) - pad, c] w[ky,kx,c,o] (+ b, activation)
template <int KH, int KW, int CIN, int COUT, typename TA>
__global__ __launch_bounds__(256) void conv_tiled_kernel(const TA* __restrict__ src, const float* __restrict__ w,
                                                         const float* __restrict__ bias, TA* __restrict__ dst,
                                                         TileDims d, float pad, int use_bias, int act, float alpha) {
    constexpr int SRC = CIN, DST = COUT;
    constexpr int ROWS = TH + KH - 1, COLS = TW + KW - 1;
    __shared__ float4 tile[ROWS * COLS];
    const int tx = threadIdx.x & (TW - 1), ty = threadIdx.x / TW;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;
    const int PH = d.ph, PW = d.pw;
    stage_tile<SRC, ROWS, COLS>(tile, src + (size_t)b * d.h * d.w * SRC, d.h, d.w, y0 - PH, x0 - PW, pad, threadIdx.x);
    __syncthreads();
    float acc[DST];
#pragma unroll
    for (int i = 0; i < DST; ++i) acc[i] = 0.f;
#pragma unroll 1
    for (int r = 0; r < KH; ++r) {
        const int ky = r;
        const float* wr = w + ky * (KW * CIN * COUT);
#pragma unroll
        for (int q = 0; q < KW; ++q) {
            const int kx = q;
            const float4 v4 = tile[(ty + r) * COLS + tx + q];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const float wv = wr[(kx * CIN + c) * COUT + o];
                    acc[o] += v[c] * wv;
                }
        }
    }
    const int ox = x0 + tx, oy = y0 + ty;
    if (ox >= d.w || oy >= d.h) return;
    const size_t off = (((size_t)b * d.h + oy) * d.w + ox) * DST;
    float out[DST];
#pragma unroll
    for (int i = 0; i < DST; ++i) {
        float t = acc[i];
        if (use_bias) t += bias[i];
        out[i] = act_apply(t, act, alpha);
    }
    store_c<DST>(dst + off, out);
}

// ---------------------------------------------------------------------------------------------
// 5x5, 4 -> 2 (the Line output conv) with FOUR adjacent pixels per thread: a 16 x 64 output tile, its
// 20 x 68 window staged once (halo overhead 1.33x instead of 1.69x), 8 window vectors per tap row feed
// 160 FMAs (10 LDS reads per pixel instead of 25).  Window columns are stored 4-way interleaved (column c at
// (c % 4) * 17 + c / 4) so that the 16 lanes of a row, which read columns 4 apart, touch consecutive 16-byte
// LDS words.  One tap row of weights (40 values) in SGPRs per rolled iteration.
// ---------------------------------------------------------------------------------------------
namespace wide {
constexpr int WTH = 16, WTW = 64, WH = WTH + 4, WW = WTW + 4;
__device__ __forceinline__ int swz(int row, int c) { return row * WW + (c & 3) * (WW / 4) + (c >> 2); }
}  // namespace wide

template <typename TA>
__global__ __launch_bounds__(256) void conv_fwd_t542(const TA* __restrict__ src, const float* __restrict__ w,
                                                     const float* __restrict__ bias, TA* __restrict__ dst, int h,
                                                     int wd, float pad, int use_bias, int act, float alpha) {
    using namespace wide;
    __shared__ float4 xs[WH * WW];
    const int tid = threadIdx.x, cg = tid & 15, r = tid >> 4;
    const int x0 = blockIdx.x * WTW, y0 = blockIdx.y * WTH, b = blockIdx.z;
    const TA* xb = src + (size_t)b * h * wd * 4;
    {
        // all of a thread's window loads are issued before the first one is waited for (a rolled load -> LDS-store loop is
        // a chain of global round trips, six per block: 35.7 us for the kernel at 32 x 256 x 512, 30.2 with the loads batched)
        constexpr int NST = (WH * WW + 255) / 256;
        float4 v[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = min(tid + 256 * k, WH * WW - 1), rr = e / WW, c = e - rr * WW;
            const int gy = y0 - 2 + rr, gx = x0 - 2 + c;
            v[k] = ld4(xb + ((size_t)min(max(gy, 0), h - 1) * wd + min(max(gx, 0), wd - 1)) * 4);
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = tid + 256 * k, rr = e / WW, c = e - rr * WW;
            const int gy = y0 - 2 + rr, gx = x0 - 2 + c;
            if (gy < 0 || gy >= h || gx < 0 || gx >= wd) v[k] = make_float4(pad, pad, pad, pad);
            if (e < WH * WW) xs[swz(rr, c)] = v[k];
        }
    }
    __syncthreads();
    // the two output channels of a pixel are one packed accumulator: v_pk_fma_f32 (both halves of the vector ALU's 64-bit
    // lanes, the input value broadcast, the weight pair from SGPRs) -- 400 instead of 800 FMA instructions per thread,
    // same products, same order, same rounding
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[p] = v2f{0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 5; ++ky) {
        const float* wr = w + ky * 40;                    // [kx][ci][co] of this tap row
        float xv[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = xs[swz(r + ky, 4 * cg + j)];
            xv[j][0] = v.x, xv[j][1] = v.y, xv[j][2] = v.z, xv[j][3] = v.w;
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const v2f wv = v2f{wr[(kx * 4 + ci) * 2], wr[(kx * 4 + ci) * 2 + 1]};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xs1 = xv[p + kx][ci];
                    acc[p] = __builtin_elementwise_fma(v2f{xs1, xs1}, wv, acc[p]);
                }
            }
    }
    const int oy = y0 + r, ox = x0 + 4 * cg;
    if (oy >= h || ox >= wd) return;
    const float b0 = use_bias ? bias[0] : 0.f, b1 = use_bias ? bias[1] : 0.f;
    float out[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        out[2 * p] = act_apply(acc[p].x + b0, act, alpha);
        out[2 * p + 1] = act_apply(acc[p].y + b1, act, alpha);
    }
    TA* o = dst + (((size_t)b * h + oy) * wd + ox) * 2;
    if (ox + 3 < wd && (wd & 1) == 0) {                   // 8 contiguous elements (row starts are aligned for 4-element
        st4(o, make_float4(out[0], out[1], out[2], out[3]));                              // accesses: ox % 4 == 0, wd even)
        st4(o + 4, make_float4(out[4], out[5], out[6], out[7]));
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (ox + p < wd) st2(o + 2 * p, make_float2(out[2 * p], out[2 * p + 1]));
    }
}

// (The same structure for dx -- 16 accumulators per thread over a float2 window of dy -- ran at the speed of the
// register-tiled conv_dgrad_px, 40 us, and was not kept.)
template <int COUT>
bool shape_ok(const ConvDims& d) {
    return d.kh == 5 && d.kw == 5 && d.cin == 4 && d.cout == COUT && d.sh == 1 && d.sw == 1 && d.oh == d.h &&
           d.ow == d.w;
}

}  // namespace

bool uocr_conv_tiled_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d) {
    const int base = UOCR_DTYPE_BASE(dtype);
    if ((base != UOCR_F32 && base != UOCR_F16) || !ctx->opt_fast || ctx->opt_tiled == 0) return false;
    return shape_ok<4>(d) || shape_ok<2>(d);
}

int uocr_conv_fwd_tiled(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                        const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha) {
    const TileDims td{d.n, d.h, d.w, d.ph, d.pw};
    const dim3 grid((d.w + TW - 1) / TW, (d.h + TH - 1) / TH, d.n), block(256);
    UOCR_DISPATCH_TA(ctx, dtype, {
        if (d.cout == 2 && d.ph == 2 && d.pw == 2 && ctx->opt_tiled != 2) {
            const dim3 wgrid((d.w + wide::WTW - 1) / wide::WTW, (d.h + wide::WTH - 1) / wide::WTH, d.n);
            hipLaunchKernelGGL((conv_fwd_t542<TA>), wgrid, block, 0, ctx->stream, (const TA*)x, (const float*)w,
                               (const float*)b, (TA*)y, d.h, d.w, (float)pad_value, use_bias, act, (float)act_alpha);
        } else if (d.cout == 4) {
            hipLaunchKernelGGL((conv_tiled_kernel<5, 5, 4, 4, TA>), grid, block, 0, ctx->stream, (const TA*)x,
                               (const float*)w, (const float*)b, (TA*)y, td, (float)pad_value, use_bias, act,
                               (float)act_alpha);
        } else {
            hipLaunchKernelGGL((conv_tiled_kernel<5, 5, 4, 2, TA>), grid, block, 0, ctx->stream, (const TA*)x,
                               (const float*)w, (const float*)b, (TA*)y, td, (float)pad_value, use_bias, act,
                               (float)act_alpha);
        }
    });
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

