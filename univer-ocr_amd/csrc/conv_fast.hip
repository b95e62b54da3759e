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
#include "finish_group.h"
#include "conv_dims.h"

namespace {

// load / store C consecutive activation elements (C = 1, 2, 4 or a multiple of 4) with the widest accesses;
// TA = float or _Float16 (uocr_common.h: ld1 / ld2 / ld4), values always arrive as float
template <int C, typename TA>
__device__ __forceinline__ void load_vec(const TA* __restrict__ p, float (&v)[C]) {
    if constexpr (C == 1) {
        v[0] = ld1(p);
    } else if constexpr (C == 2) {
        const float2 t = ld2(p);
        v[0] = t.x;
        v[1] = t.y;
    } else {
        static_assert(C % 4 == 0, "channel count must be 1, 2 or a multiple of 4");
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
            const float4 t = ld4(p + 4 * q);
            v[4 * q] = t.x;
            v[4 * q + 1] = t.y;
            v[4 * q + 2] = t.z;
            v[4 * q + 3] = t.w;
        }
    }
}

template <int C, typename TA>
__device__ __forceinline__ void store_vec(TA* __restrict__ p, const float (&v)[C]) {
    if constexpr (C == 1) {
        st1(p, v[0]);
    } else if constexpr (C == 2) {
        st2(p, make_float2(v[0], v[1]));
    } else {
#pragma unroll
        for (int q = 0; q < C / 4; ++q) st4(p + 4 * q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
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

// dx epilogue: v[c] *= act'(y[off + c]) from the activation output y (ActMask, conv_dims.h)
template <int C, typename TA>
__device__ __forceinline__ void apply_mask(float (&v)[C], const TA* __restrict__ mask_y, size_t off, int act,
                                           float alpha) {
    if (act == UOCR_ACT_NONE) return;
    float m[C];
    load_vec<C>(mask_y + off, m);
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] *= act_grad_from_output<float>(m[c], act, alpha);
}

// ---------------------------------------------------------------------------------------------
// forward: block (64, 4); a thread computes COB output channels of PY vertically adjacent pixels
// (register tiling: the (PY-1)*SH + KH input rows are loaded once and feed all PY outputs, which
// cuts the L1 re-read traffic of a KHxKW window from KH*KW to ((PY-1)*SH+KH)*KW/PY loads per pixel;
// these kernels were L1-bandwidth bound, not HBM bound, with one pixel per thread).
// NQ = COUT / COB adjacent lanes share a pixel, so one wave stores 64 * COB contiguous floats.
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int COUT, int SH, int SW, int COB, int PY, typename TA>
__global__ __launch_bounds__(256) void conv_fwd_fast(const TA* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, TA* __restrict__ y,
                                                     FastDims d, float pad, int use_bias, int act, float alpha) {
    constexpr int NQ = COUT / COB;     // lanes per pixel
    constexpr int PXW = 64 / NQ;       // pixels per wave row
    constexpr int ROWS = (PY - 1) * SH + KH;
    const int ox = blockIdx.x * PXW + threadIdx.x / NQ;
    const int oy0 = (blockIdx.y * 4 + threadIdx.y) * PY;
    const int oc0 = (threadIdx.x % NQ) * COB;
    const int b = blockIdx.z;
    const float* __restrict__ wl = w;   // scalar-cache operands (fine for the shapes left on this kernel)
    const bool valid = ox < d.ow && oy0 < d.oh;
    if (!valid) return;
    float acc[PY][COB];
#pragma unroll
    for (int p = 0; p < PY; ++p)
#pragma unroll
        for (int o = 0; o < COB; ++o) acc[p][o] = 0.f;
    const int iy0 = oy0 * SH - d.ph, ix0 = ox * SW - d.pw;
    const TA* xb = x + (size_t)b * d.h * d.w * CIN;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = iy0 + r;
        const bool row_ok = iy >= 0 && iy < d.h;
        const int iyc = min(max(iy, 0), d.h - 1);
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int ix = ix0 + kx;
            // unconditional load from a clamped address + select: a per-tap branch would put every
            // load in its own divergent region (s_waitcnt vmcnt(0) per tap, no loads in flight)
            float xv[CIN];
            load_vec<CIN>(xb + ((size_t)iyc * d.w + min(max(ix, 0), d.w - 1)) * CIN, xv);
            if (!(row_ok && ix >= 0 && ix < d.w)) {
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[c] = pad;
            }
#pragma unroll
            for (int p = 0; p < PY; ++p) {
                const int ky = r - p * SH;          // compile-time after unrolling
                if (ky >= 0 && ky < KH) {
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int o = 0; o < COB; ++o)
                            acc[p][o] += xv[c] * wl[((ky * KW + kx) * CIN + c) * COUT + oc0 + o];
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PY; ++p) {
        if (!valid || oy0 + p >= d.oh) break;
        float out[COB];
#pragma unroll
        for (int o = 0; o < COB; ++o) {
            float v = acc[p][o];
            if (use_bias) v += bias[oc0 + o];
            out[o] = act_apply(v, act, alpha);
        }
        store_vec<COB>(y + (((size_t)b * d.oh + oy0 + p) * d.ow + ox) * COUT + oc0, out);
    }
}

// ---------------------------------------------------------------------------------------------
// backward data: block (64, 4); a thread computes all CIN channels of PY vertically adjacent INPUT
// pixels.  Stride 1: the PY + KH - 1 rows of dy that the PY pixels see are loaded once.
// Stride > 1: PY must be 1 (taps are predicated on the stride phase of each lane).
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int COUT, int SH, int SW, int PY, typename TA>
__global__ __launch_bounds__(256) void conv_dgrad_fast(const TA* __restrict__ dy, const float* __restrict__ w,
                                                       TA* __restrict__ dx, FastDims d,
                                                       const TA* __restrict__ mask_y, int mask_act, float mask_alpha) {
    static_assert(PY == 1 || (SH == 1 && SW == 1), "row tiling of dgrad needs stride 1");
    const int ix = blockIdx.x * 64 + threadIdx.x;
    const int iy0 = (blockIdx.y * 4 + threadIdx.y) * PY;
    const int b = blockIdx.z;
    const float* __restrict__ wl = w;
    const bool valid = ix < d.w && iy0 < d.h;
    if (!valid) return;
    float acc[PY][CIN];
#pragma unroll
    for (int p = 0; p < PY; ++p)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[p][c] = 0.f;
    const TA* gb = dy + (size_t)b * d.oh * d.ow * COUT;
    if constexpr (SH == 1 && SW == 1) {
        constexpr int ROWS = PY + KH - 1;
        const int gy0 = iy0 + d.ph - (KH - 1);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int gy = gy0 + r;
            const bool row_ok = gy >= 0 && gy < d.oh;
            const int gyc = min(max(gy, 0), d.oh - 1);
#pragma unroll
            for (int kx = KW - 1; kx >= 0; --kx) {
                const int gx = ix + d.pw - kx;
                float g[COUT];   // clamped unconditional load + select (see conv_fwd_fast)
                load_vec<COUT>(gb + ((size_t)gyc * d.ow + min(max(gx, 0), d.ow - 1)) * COUT, g);
                if (!(row_ok && gx >= 0 && gx < d.ow)) {
#pragma unroll
                    for (int o = 0; o < COUT; ++o) g[o] = 0.f;
                }
#pragma unroll
                for (int p = 0; p < PY; ++p) {
                    const int ky = p + (KH - 1) - r;   // compile-time after unrolling
                    if (ky >= 0 && ky < KH) {
#pragma unroll
                        for (int c = 0; c < CIN; ++c)
#pragma unroll
                            for (int o = 0; o < COUT; ++o)
                                acc[p][c] += g[o] * wl[((ky * KW + kx) * CIN + c) * COUT + o];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int ky = KH - 1; ky >= 0; --ky) {
            const int ty = iy0 + d.ph - ky;
            const int gy = ty / SH;
            const bool row_ok = ty >= 0 && gy < d.oh && gy * SH == ty;
#pragma unroll
            for (int kx = KW - 1; kx >= 0; --kx) {
                const int tx = ix + d.pw - kx;
                const int gx = tx / SW;
                // strided: most taps miss the stride phase of a lane, so a branch (no load at all)
                // beats the clamped unconditional load here (Char conv_1 dx: 16 us vs 94 us)
                if (row_ok && tx >= 0 && gx < d.ow && gx * SW == tx) {
                    float g[COUT];
                    load_vec<COUT>(gb + ((size_t)gy * d.ow + gx) * COUT, g);
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int o = 0; o < COUT; ++o) acc[0][c] += g[o] * wl[((ky * KW + kx) * CIN + c) * COUT + o];
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PY; ++p) {
        if (!valid || iy0 + p >= d.h) break;
        const size_t off = (((size_t)b * d.h + iy0 + p) * d.w + ix) * CIN;
        apply_mask<CIN>(acc[p], mask_y, off, mask_act, mask_alpha);
        store_vec<CIN>(dx + off, acc[p]);
    }
}

// ---------------------------------------------------------------------------------------------
// backward data of the 5x5 / stride 2 / pad 2 encoder convs (Paragraph down_1/2, Line down_1/2).  A thread
// owns the 2x2 block of dx pixels (2j + py, 2i + px): pixel parity selects the taps (ky = py, py + 2, ..)
// and all four pixels read the same 3x3 neighbourhood of dy, so the 25 taps are spread over the four pixels
// without a single predicated tap (conv_dgrad_fast tests every tap against the stride phase of its lane:
// three quarters of the lanes idle in each of the 25 iterations).
//   y = 2j:     ky = 0, 2, 4 -> dy rows j+1, j, j-1          y = 2j+1:  ky = 1, 3 -> dy rows j+1, j
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, typename TA>
__global__ __launch_bounds__(256) void conv_dgrad_s2(const TA* __restrict__ dy, const float* __restrict__ w,
                                                     TA* __restrict__ dx, FastDims d,
                                                     const TA* __restrict__ mask_y, int mask_act,
                                                     float mask_alpha) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (2 * i >= d.w || 2 * j >= d.h) return;
    const TA* gb = dy + (size_t)b * d.oh * d.ow * COUT;
    float g[3][3][COUT];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int gy = j - 1 + r;
        const bool row_ok = gy >= 0 && gy < d.oh;
        const int gyc = min(max(gy, 0), d.oh - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int gx = i - 1 + c;
            load_vec<COUT>(gb + ((size_t)gyc * d.ow + min(max(gx, 0), d.ow - 1)) * COUT, g[r][c]);
            if (!(row_ok && gx >= 0 && gx < d.ow)) {
#pragma unroll
                for (int o = 0; o < COUT; ++o) g[r][c][o] = 0.f;
            }
        }
    }
    float out[2][2][CIN];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) out[q >> 1][q & 1][ci] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const int py = ky & 1, px = kx & 1;
            const int r = (py ? (3 - ky) / 2 : 1 - ky / 2) + 1, c = (px ? (3 - kx) / 2 : 1 - kx / 2) + 1;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int o = 0; o < COUT; ++o)
                    out[py][px][ci] += g[r][c][o] * w[((ky * 5 + kx) * CIN + ci) * COUT + o];
        }
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        const int y = 2 * j + py;
        if (y >= d.h) break;
        const size_t off = (((size_t)b * d.h + y) * d.w + 2 * i) * CIN;
        if (CIN == 1 && (d.w & 1) == 0) {                // even width: the pixel pair is one aligned 8-byte store
            float v[2] = {out[py][0][0], out[py][1][0]};
            apply_mask<2>(v, mask_y, off, mask_act, mask_alpha);
            store_vec<2>(dx + off, v);
        } else {
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                if (2 * i + px >= d.w) break;
                apply_mask<CIN>(out[py][px], mask_y, off + px * CIN, mask_act, mask_alpha);
                store_vec<CIN>(dx + off + px * CIN, out[py][px]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward data of the Char net's first conv (5x3, stride (2, 1), pad (0, 1), 1 <- 64 channels): the page-input
// gradient of that net.  conv_dgrad_fast gives a thread a whole pixel -- 15 predicated taps x 16 float4 loads one
// after the other on 65 k threads (one wave per SIMD): latency.  Here SIXTEEN lanes share a pixel, each owns 4
// of the 64 channels (a pixel's dy is one contiguous 256-B read), only the tap rows of the pixel's stride phase
// are visited, and the 16 partial sums are added with four DPP row shifts.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_dgrad_c64s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                        float* __restrict__ dx, int n, int h, int wd, int oh, int ow,
                                                        const float* __restrict__ mask_y, int mask_act,
                                                        float mask_alpha) {
    constexpr int KH = 5, KW = 3, CO = 64;
    // grid = (16-pixel groups of a row, rows, images): the pixel's coordinates without an integer division (three 64-bit
    // ones per thread used to come before the first load)
    const int q = threadIdx.x & 15;                        // channels 4q .. 4q + 3
    const int b = blockIdx.z, y = blockIdx.y;
    const bool live = (int)(blockIdx.x * 16 + (threadIdx.x >> 4)) < wd;
    const int x = live ? blockIdx.x * 16 + (threadIdx.x >> 4) : wd - 1;
    const size_t pix = ((size_t)b * h + y) * wd + x;
    const float* gb = dy + (size_t)b * oh * ow * CO + 4 * q;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {                          // tap rows ky = (y & 1) + 2t: y - ky even
        const int ky = (y & 1) + 2 * t, gy = (y - ky) / 2;
        if (ky >= KH || y - ky < 0 || gy >= oh) continue;  // (uniform over the 16 lanes of a pixel)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int gx = x + 1 - kx;
            if (gx < 0 || gx >= ow) continue;
            const float4 g = *reinterpret_cast<const float4*>(gb + ((size_t)gy * ow + gx) * CO);
            const float4 wv = *reinterpret_cast<const float4*>(w + (ky * KW + kx) * CO + 4 * q);
            acc += g.x * wv.x + g.y * wv.y + g.z * wv.z + g.w * wv.w;
        }
    }
    // sum over the 16 lanes of the pixel (one DPP row = 16 lanes)
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (!live || q != 0) return;
    if (mask_act != UOCR_ACT_NONE) acc *= act_grad_from_output<float>(mask_y[pix], mask_act, mask_alpha);
    dx[pix] = acc;
}

// One-channel row segment seg[j] = row[first + j], j < N, as aligned float4 loads: `first` = 4*a - OFFS
// with OFFS compile-time, the row start 16-byte aligned and width % 4 == 0, so every float4 is either
// fully inside or fully outside the row (outside -> fill).  8-11 dword loads at a 16 B lane stride
// (25 % of each 128-B line per instruction, TA-bound) become 3-4 fully contiguous 16-B loads.
template <int N, int OFFS, typename TA>
__device__ __forceinline__ void load_row_c1(const TA* __restrict__ row, int a4, int width, bool row_ok,
                                            float fill, float (&seg)[N][1]) {
    constexpr int NV4 = (OFFS + N + 3) / 4;
    float buf[NV4 * 4];
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
        const int col = a4 + 4 * k;
        float4 v = ld4(row + min(max(col, 0), width - 4));
        if (!(row_ok && col >= 0 && col < width)) v = make_float4(fill, fill, fill, fill);
        buf[4 * k] = v.x;
        buf[4 * k + 1] = v.y;
        buf[4 * k + 2] = v.z;
        buf[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) seg[j][0] = buf[OFFS + j];
}

// ---------------------------------------------------------------------------------------------
// Row-loop kernels for the few-channel (COUT <= 4 / CIN <= 16) convs.  Structure chosen from the ISA:
//   * the tap-ROW loop (ky) is a real loop (not unrolled), so only one row of weights
//     (KW*CIN*COUT <= 80 values) is live: they stay in SGPRs (s_load from the uniform pointer
//     w + ky*row) and feed v_fmac directly -- fully unrolled, the 400 weights of a 5x5x4x4 kernel
//     overflow the ~100 SGPRs and every FMA pays 3.4 v_readlane; through LDS/VGPRs they cost an
//     LDS read per 4 FMAs and 256 VGPRs (occupancy 1);
//   * a thread owns PX horizontally adjacent pixels: the (PX-1)*SW + KW input vectors of a row are
//     loaded once and reused by all PX*KW taps, so L1 traffic per pixel drops from KW to
//     ((PX-1)*SW + KW)/PX loads per row, and every weight is reused PX times from its SGPR.
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int CIN, int COUT, int SH, int SW, int PX, typename TA>
__global__ __launch_bounds__(256) void conv_fwd_px(const TA* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, TA* __restrict__ y,
                                                   FastDims d, float pad, int use_bias, int act, float alpha) {
    constexpr int NXV = (PX - 1) * SW + KW;
    const int ox0 = (blockIdx.x * 64 + threadIdx.x) * PX;
    const int oy = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (ox0 >= d.ow || oy >= d.oh) return;
    float acc[PX][COUT];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[p][o] = 0.f;
    const int ix0 = ox0 * SW - d.pw;
    const TA* xb = x + (size_t)b * d.h * d.w * CIN;
#pragma unroll 1
    for (int ky = 0; ky < KH; ++ky) {
        const int iy = oy * SH - d.ph + ky;
        const bool row_ok = iy >= 0 && iy < d.h;
        const TA* xr = xb + (size_t)min(max(iy, 0), d.h - 1) * d.w * CIN;
        const float* wr = w + ky * (KW * CIN * COUT);
        float xv[NXV][CIN];
        bool vec_done = false;
        if constexpr (CIN == 1 && PX % 4 == 0) {
            constexpr int PWC = KW / 2, OFFS = (4 - PWC % 4) % 4;      // "same" padding: ix0 = 4a*SW - PWC
            if (d.pw == PWC && d.w % 4 == 0) {
                load_row_c1<NXV, OFFS>(xr, ix0 - OFFS, d.w, row_ok, pad, xv);
                vec_done = true;
            }
        }
        if (!vec_done) {
#pragma unroll
            for (int j = 0; j < NXV; ++j) {
                const int ix = ix0 + j;
                load_vec<CIN>(xr + (size_t)min(max(ix, 0), d.w - 1) * CIN, xv[j]);
                if (!(row_ok && ix >= 0 && ix < d.w)) {
#pragma unroll
                    for (int c = 0; c < CIN; ++c) xv[j][c] = pad;
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const float wv = wr[(kx * CIN + c) * COUT + o];
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[p][o] += xv[p * SW + kx][c] * wv;
                }
    }
    if constexpr (COUT == 1 && PX % 4 == 0) {
        if (d.ow % PX == 0) {         // the PX pixels of the thread are PX / 4 aligned float4s
            float out[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) out[p] = act_apply(acc[p][0] + (use_bias ? bias[0] : 0.f), act, alpha);
            store_vec<PX>(y + ((size_t)b * d.oh + oy) * d.ow + ox0, out);
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (ox0 + p >= d.ow) break;
        float out[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float v = acc[p][o];
            if (use_bias) v += bias[o];
            out[o] = act_apply(v, act, alpha);
        }
        store_vec<COUT>(y + (((size_t)b * d.oh + oy) * d.ow + ox0 + p) * COUT, out);
    }
}

// stride-1 backward data with the same structure: dx[y, x0+p, c] = sum_{ky,kx,o} dy[y+ph-ky, x0+p+pw-kx, o] w[ky,kx,c,o]
template <int KH, int KW, int CIN, int COUT, int PX, typename TA>
__global__ __launch_bounds__(256) void conv_dgrad_px(const TA* __restrict__ dy, const float* __restrict__ w,
                                                     TA* __restrict__ dx, FastDims d,
                                                     const TA* __restrict__ mask_y, int mask_act, float mask_alpha) {
    constexpr int NG = PX + KW - 1;
    const int ix0 = (blockIdx.x * 64 + threadIdx.x) * PX;
    const int iy = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    if (ix0 >= d.w || iy >= d.h) return;
    float acc[PX][CIN];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[p][c] = 0.f;
    const TA* gb = dy + (size_t)b * d.oh * d.ow * COUT;
    const int gx0 = ix0 + d.pw - (KW - 1);
#pragma unroll 1
    for (int ky = KH - 1; ky >= 0; --ky) {
        const int gy = iy + d.ph - ky;
        const bool row_ok = gy >= 0 && gy < d.oh;
        const TA* gr = gb + (size_t)min(max(gy, 0), d.oh - 1) * d.ow * COUT;
        const float* wr = w + ky * (KW * CIN * COUT);
        float g[NG][COUT];
        bool vec_done = false;
        if constexpr (COUT == 1 && PX % 4 == 0) {
            constexpr int PWC = KW / 2, OFFS = (((KW - 1 - PWC) % 4) + 4) % 4;   // gx0 = 4a - (KW-1-PWC)
            if (d.pw == PWC && d.ow % 4 == 0) {
                load_row_c1<NG, OFFS>(gr, gx0 - OFFS, d.ow, row_ok, 0.f, g);
                vec_done = true;
            }
        }
        if (!vec_done) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int gx = gx0 + j;
                load_vec<COUT>(gr + (size_t)min(max(gx, 0), d.ow - 1) * COUT, g[j]);
                if (!(row_ok && gx >= 0 && gx < d.ow)) {
#pragma unroll
                    for (int o = 0; o < COUT; ++o) g[j][o] = 0.f;
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const float wv = wr[(kx * CIN + c) * COUT + o];
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[p][c] += g[p + (KW - 1) - kx][o] * wv;
                }
    }
    if constexpr (CIN == 1 && PX % 4 == 0) {
        if (d.w % PX == 0) {
            float out[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) out[p] = acc[p][0];
            const size_t off = ((size_t)b * d.h + iy) * d.w + ix0;
            apply_mask<PX>(out, mask_y, off, mask_act, mask_alpha);
            store_vec<PX>(dx + off, out);
            return;
        }
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (ix0 + p >= d.w) break;
        const size_t off = (((size_t)b * d.h + iy) * d.w + ix0 + p) * CIN;
        apply_mask<CIN>(acc[p], mask_y, off, mask_act, mask_alpha);
        store_vec<CIN>(dx + off, acc[p]);
    }
}

// ---------------------------------------------------------------------------------------------
// 16-channel <-> 1-channel 3x3 convs of the Monochrome net (conv_2 forward / dw, conv_1 dx): the
// 16-channel tensor (268 MB at 32x256x512) is read through a KHxKW window.  Four adjacent lanes
// share a pixel, each owns one channel quad: every load instruction of a wave is 64 x 16 B fully
// contiguous (one lane per pixel reads 16 B at a 64 B stride -> 4x the cache-line accesses), the
// lane keeps its KH*KW*4 weights in VGPRs for the whole thread (no weight traffic at all), PY
// vertically adjacent outputs share the loaded rows, and the 4 partial sums of a pixel are
// combined with two DPP-class shuffles.
//   FLIP = false: y[p]  = sum_{ky,kx,c} x[p + (ky,kx) - pad, c] * w[ky,kx,c]        (forward, COUT = 1)
//   FLIP = true : dx[p] = sum_{ky,kx,o} dy[p - (ky,kx) + pad, o] * w[ky,kx,o]       (dx, CIN = 1)
// (both weight tensors are the flat array w[(ky*KW + kx)*16 + ch])
// ---------------------------------------------------------------------------------------------
template <int KH, int KW, int PY, bool FLIP>
__global__ __launch_bounds__(256) void conv_c16_reduce(const float* __restrict__ src, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ dst,
                                                       int n, int h, int wd, int ph, int pw, float pad, int use_bias,
                                                       int act, float alpha, const float* __restrict__ mask_y,
                                                       int mask_act, float mask_alpha) {
    constexpr int C = 16, ROWS = PY + KH - 1;
    const int q = threadIdx.x & 3;
    const int ox = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int oy0 = (blockIdx.y * 4 + threadIdx.y) * PY;
    const int b = blockIdx.z;
    float wreg[KH][KW][4];
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int sky = FLIP ? KH - 1 - ky : ky, skx = FLIP ? KW - 1 - kx : kx;
            const float4 t = *reinterpret_cast<const float4*>(w + (sky * KW + skx) * C + q * 4);
            wreg[ky][kx][0] = t.x;
            wreg[ky][kx][1] = t.y;
            wreg[ky][kx][2] = t.z;
            wreg[ky][kx][3] = t.w;
        }
    // same-size stride-1 window: source row = oy - PH + r, PH = pad (forward) or KH-1-pad (flipped)
    const int PH = FLIP ? KH - 1 - ph : ph, PW = FLIP ? KW - 1 - pw : pw;
    float acc[PY];
#pragma unroll
    for (int p = 0; p < PY; ++p) acc[p] = 0.f;
    const float* sb = src + (size_t)b * h * wd * C + q * 4;
    const int oxc = min(ox, wd - 1);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = oy0 - PH + r;
        const bool row_ok = iy >= 0 && iy < h;
        const float* sr = sb + (size_t)min(max(iy, 0), h - 1) * wd * C;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int ix = oxc - PW + kx;
            float4 v = *reinterpret_cast<const float4*>(sr + (size_t)min(max(ix, 0), wd - 1) * C);
            if (!(row_ok && ix >= 0 && ix < wd)) v = make_float4(pad, pad, pad, pad);
#pragma unroll
            for (int p = 0; p < PY; ++p) {
                const int ky = r - p;      // compile-time after unrolling
                if (ky >= 0 && ky < KH)
                    acc[p] += v.x * wreg[ky][kx][0] + v.y * wreg[ky][kx][1] + v.z * wreg[ky][kx][2] +
                              v.w * wreg[ky][kx][3];
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PY; ++p) {
        float v = acc[p];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (q == 0 && ox < wd && oy0 + p < h) {
            if (use_bias) v += bias[0];
            const size_t off = ((size_t)b * h + oy0 + p) * wd + ox;
            v = act_apply(v, act, alpha);
            if (mask_act != UOCR_ACT_NONE) v *= act_grad_from_output<float>(mask_y[off], mask_act, mask_alpha);
            dst[off] = v;
        }
    }
}

// The opposite direction, 1 channel -> 16 channels through a KHxKW window (Monochrome conv_1 forward,
// conv_2 dx): lane (pixel, quad) produces 4 of the 16 channels, so every store instruction of a wave
// is 64 x 16 B contiguous.  With one lane per pixel(s) the 64 B per pixel were written as four 16 B
// pieces at a 64-128 B lane stride and rocprofv3 WRITE_SIZE showed 1.37x the algorithmic bytes.
//   FLIP = false: y[p,o]  = sum_k x[p + k - pad] * w[k,o] (+ b[o], activation)
//   FLIP = true : dx[p,c] = sum_k dy[p - k + pad] * w[k,c]  (* act'(mask_y[p,c]))
template <int KH, int KW, int PY, bool FLIP>
__global__ __launch_bounds__(256) void conv_c16_expand(const float* __restrict__ src, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ dst,
                                                       int n, int h, int wd, int ph, int pw, float pad, int use_bias,
                                                       int act, float alpha, const float* __restrict__ mask_y,
                                                       int mask_act, float mask_alpha) {
    constexpr int C = 16, ROWS = PY + KH - 1;
    const int q = threadIdx.x & 3;
    const int ox = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int oy0 = (blockIdx.y * 4 + threadIdx.y) * PY;
    const int b = blockIdx.z;
    float wreg[KH][KW][4];
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int sky = FLIP ? KH - 1 - ky : ky, skx = FLIP ? KW - 1 - kx : kx;
            const float4 t = *reinterpret_cast<const float4*>(w + (sky * KW + skx) * C + q * 4);
            wreg[ky][kx][0] = t.x;
            wreg[ky][kx][1] = t.y;
            wreg[ky][kx][2] = t.z;
            wreg[ky][kx][3] = t.w;
        }
    const int PH = FLIP ? KH - 1 - ph : ph, PW = FLIP ? KW - 1 - pw : pw;
    float acc[PY][4];
#pragma unroll
    for (int p = 0; p < PY; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = 0.f;
    const float* sb = src + (size_t)b * h * wd;
    const int oxc = min(ox, wd - 1);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int iy = oy0 - PH + r;
        const bool row_ok = iy >= 0 && iy < h;
        const float* sr = sb + (size_t)min(max(iy, 0), h - 1) * wd;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int ix = oxc - PW + kx;
            float v = sr[min(max(ix, 0), wd - 1)];
            if (!(row_ok && ix >= 0 && ix < wd)) v = pad;
#pragma unroll
            for (int p = 0; p < PY; ++p) {
                const int ky = r - p;      // compile-time after unrolling
                if (ky >= 0 && ky < KH) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[p][j] += v * wreg[ky][kx][j];
                }
            }
        }
    }
    if (ox >= wd) return;
#pragma unroll
    for (int p = 0; p < PY; ++p) {
        if (oy0 + p >= h) break;
        const size_t off = (((size_t)b * h + oy0 + p) * wd + ox) * C + q * 4;
        float out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[p][j];
            if (use_bias) v += bias[q * 4 + j];
            out[j] = act_apply(v, act, alpha);
        }
        apply_mask<4>(out, mask_y, off, mask_act, mask_alpha);
        store_vec<4>(dst + off, out);
    }
}

// dw[ky,kx,c] (+ db) of the 16 -> 1 conv, x-stationary: dw[ky,kx,c] = sum_q x[q,c] * dy[q - (ky,kx) + pad].
// A lane (input pixel q, channel quad) loads its 16 B of x exactly ONCE (64 lanes = 1 KiB contiguous)
// and the 1-channel dy at the KH*KW shifted positions (4 B each, L1 hits), so the 268 MB tensor
// crosses L1 once instead of KH*KW times.  Needs padding_value == 0 (the padded border then
// contributes nothing) and a same-size output.  36 + 1 accumulators per lane; all-reduce over the
// 16 pixels of a wave by xor-shuffles 4..32, LDS over the 4 waves, partial[blk][quad][KH*KW*4 + 1].
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_c16_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ partial, int h, int wd, int ph, int pw,
                                                      int rows_per_block, int nbands) {
    constexpr int C = 16, NA = KH * KW * 4 + 1;
    __shared__ float red[4][4][NA];
    const int q = threadIdx.x & 3, pl = threadIdx.x >> 2, wv = threadIdx.y;
    const int band = blockIdx.x % nbands, b = blockIdx.x / nbands;
    const int row0 = band * rows_per_block, row1 = min(h, row0 + rows_per_block);
    float acc[KH][KW][4], accb = 0.f;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ky][kx][j] = 0.f;
    const float* xb = x + (size_t)b * h * wd * C + q * 4;
    const float* gb = dy + (size_t)b * h * wd;
    for (int iy = row0 + wv; iy < row1; iy += 4) {
        for (int ix = pl; ix < wd; ix += 16) {
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)iy * wd + ix) * C);
            accb += gb[(size_t)iy * wd + ix];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const int oy = iy + ph - ky;
                const bool row_ok = oy >= 0 && oy < h;
                const float* gr = gb + (size_t)min(max(oy, 0), h - 1) * wd;
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const int ox = ix + pw - kx;
                    float g = gr[min(max(ox, 0), wd - 1)];
                    if (!(row_ok && ox >= 0 && ox < wd)) g = 0.f;
                    acc[ky][kx][0] += v.x * g;
                    acc[ky][kx][1] += v.y * g;
                    acc[ky][kx][2] += v.z * g;
                    acc[ky][kx][3] += v.w * g;
                }
            }
        }
    }
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[ky][kx][j];
#pragma unroll
                for (int m = 4; m < 64; m <<= 1) t += __shfl_xor(t, m, 64);
                acc[ky][kx][j] = t;
            }
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) accb += __shfl_xor(accb, m, 64);
    if (pl == 0) {
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wv][q][(ky * KW + kx) * 4 + j] = acc[ky][kx][j];
        red[wv][q][NA - 1] = accb;
    }
    __syncthreads();
    const int tid = wv * 64 + threadIdx.x;
    float* out = partial + (size_t)blockIdx.x * 4 * NA;
    for (int a = tid; a < 4 * NA; a += 256) {
        const int qq = a / NA, i = a % NA;
        out[a] = red[0][qq][i] + red[1][qq][i] + red[2][qq][i] + red[3][qq][i];
    }
}

// block per (accumulator, quad): float64 sum of the block partials into dw[tap*16 + q*4 + j] / db
template <int KH, int KW>
__global__ __launch_bounds__(256) void conv_c16_wgrad_finish(const float* __restrict__ partial, float* __restrict__ dw,
                                                             float* __restrict__ db, int nblocks, int use_bias,
                                                             int accumulate) {
    constexpr int NA = KH * KW * 4 + 1;
    __shared__ double smem[16];
    const int i = blockIdx.x, q = blockIdx.y;
    double s = 0.0;
    for (int blk = threadIdx.x; blk < nblocks; blk += blockDim.x) s += (double)partial[((size_t)blk * 4 + q) * NA + i];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst;
    if (i < NA - 1) {
        dst = dw + (i / 4) * 16 + q * 4 + (i % 4);
    } else {
        if (q != 0) return;      // every quad lane summed dy: keep quad 0's copy
        dst = db;
        if (!use_bias) s = 0.0;
    }
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
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

template <int KH, int KW, int CIN, int COUT, int SH, int SW, int KYR, int COB, int PX, bool CLAMP, typename TA>
__global__ __launch_bounds__(256) void conv_wgrad_fast(const TA* __restrict__ x, const TA* __restrict__ dy,
                                                       float* __restrict__ partial, FastDims d, float pad,
                                                       int rows_per_block, int nbands) {
    // a thread visits PX horizontally adjacent output pixels per step: the (PX-1)*SW + KW input vectors
    // of a tap row are loaded once for all PX*KW (pixel, tap) pairs
    using C = WgradCfg<KH, KW, CIN, COUT, SH, SW, KYR, COB>;
    constexpr int NXV = (PX - 1) * SW + KW;
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
    const TA* xb = x + (size_t)b * d.h * d.w * CIN;
    const TA* gb = dy + (size_t)b * d.oh * d.ow * COUT + oc0;
    for (int oy = row0 + wv; oy < row1; oy += 4) {
        for (int ox0 = lane * PX; ox0 < d.ow; ox0 += 64 * PX) {
            float g[PX][COB];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (ox0 + p < d.ow) {
                    load_vec<COB>(gb + ((size_t)oy * d.ow + ox0 + p) * COUT, g[p]);
                } else {
#pragma unroll
                    for (int o = 0; o < COB; ++o) g[p][o] = 0.f;
                }
                if (kyg == 0) {
#pragma unroll
                    for (int o = 0; o < COB; ++o) acc[C::NW + o] += g[p][o];
                }
            }
            const int iy0 = oy * SH - d.ph + ky0, ix0 = ox0 * SW - d.pw;
#pragma unroll
            for (int kyl = 0; kyl < KYR; ++kyl) {
                const int iy = iy0 + kyl;
                const bool row_ok = iy >= 0 && iy < d.h;
                const int iyc = min(max(iy, 0), d.h - 1);
                float xv[NXV][CIN];
                bool vec_done = false;
                if constexpr (CIN == 1 && PX % 4 == 0) {
                    constexpr int PWC = KW / 2, OFFS = (4 - PWC % 4) % 4;
                    if (d.pw == PWC && d.w % 4 == 0) {
                        load_row_c1<NXV, OFFS>(xb + (size_t)iyc * d.w, ix0 - OFFS, d.w, row_ok, pad, xv);
                        vec_done = true;
                    }
                }
                if (!vec_done) {
#pragma unroll
                    for (int j = 0; j < NXV; ++j) {
                        const int ix = ix0 + j;
                        // measured per shape: the clamped unconditional load is 1.3-2x faster for the
                        // CIN = 4 kernels and 3.5x SLOWER for the CIN = 1 5x5 kernel than the branch
                        if constexpr (CLAMP) {
                            load_vec<CIN>(xb + ((size_t)iyc * d.w + min(max(ix, 0), d.w - 1)) * CIN, xv[j]);
                            if (!(row_ok && ix >= 0 && ix < d.w)) {
#pragma unroll
                                for (int c = 0; c < CIN; ++c) xv[j][c] = pad;
                            }
                        } else {
                            if (row_ok && ix >= 0 && ix < d.w) {
                                load_vec<CIN>(xb + ((size_t)iy * d.w + ix) * CIN, xv[j]);
                            } else {
#pragma unroll
                                for (int c = 0; c < CIN; ++c) xv[j][c] = pad;
                            }
                        }
                    }
                }
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int o = 0; o < COB; ++o)
#pragma unroll
                            for (int p = 0; p < PX; ++p)
                                acc[((kyl * KW + kx) * CIN + c) * COB + o] += xv[p * SW + kx][c] * g[p][o];
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
                                                              int accumulate, float unscale) {
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
    s *= (double)unscale;                                  // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// ---------------------------------------------------------------------------------------------
// dw / db of the Line output conv (5x5, 4 -> 2, stride 1, pad 2) from an LDS tile.  conv_wgrad_fast gives
// each of the 5 tap rows its own block, so x and dy cross the L2 five times (0.5 GB at 256x512x32: the
// kernel ran at L2 speed); here a block of FIVE waves stages a 16x64 tile of dy and its 20x68 window of x
// once and wave k owns tap row k: 40 accumulators per lane, 4 adjacent pixels per lane per trip (8 window
// vectors feed 160 FMAs).  Window columns are stored 4-way interleaved (column c at (c % 4) * 17 + c / 4)
// so that the 16 lanes of a row, which read columns 4 apart, touch consecutive 16-byte LDS words.
// ---------------------------------------------------------------------------------------------
namespace t542 {
constexpr int TH = 16, TW = 64, WH = TH + 4, WW = TW + 4, NACC = 42, NP = 64;   // 40 dw of a tap row + db
__device__ __forceinline__ int swz(int row, int c) { return row * WW + (c & 3) * (WW / 4) + (c >> 2); }
}  // namespace t542

template <typename TA>
__global__ __launch_bounds__(320) void conv_wgrad_t542(const TA* __restrict__ x, const TA* __restrict__ dy,
                                                       float* __restrict__ partial, int h, int wd, float pad,
                                                       int tiles_per_block) {
    using namespace t542;
    __shared__ float4 xs[WH * WW];
    __shared__ float2 gs[TH * TW];
    const int tid = threadIdx.x, lane = tid & 63, ky = tid >> 6;
    const int x0 = blockIdx.x * TW, b = blockIdx.z;
    const int tile0 = blockIdx.y * tiles_per_block, tiles_y = (h + TH - 1) / TH;
    const TA* xb = x + (size_t)b * h * wd * 4;
    const TA* gb = dy + (size_t)b * h * wd * 2;
    float acc[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) acc[a] = 0.f;
    const int cg = lane & 15, rsub = lane >> 4;
    const int t_end = min(tiles_y, tile0 + tiles_per_block);
    for (int t = tile0; t < t_end; ++t) {
        const int y0 = t * TH;
        __syncthreads();                                   // the previous tile's reads are over
        // (rolled staging loops on purpose: with the loads batched in registers, or prefetched across tiles,
        // the kernel needs 160+ VGPRs and loses more in occupancy than the overlap gains: 57 -> 60-64 us; pairs of
        // loads through stage_batched: 56 -> 81 us)
        for (int e = tid; e < WH * WW; e += 320) {
            const int r = e / WW, c = e - r * WW;
            const int gy = y0 - 2 + r, gx = x0 - 2 + c;
            float4 v = ld4(xb + ((size_t)min(max(gy, 0), h - 1) * wd + min(max(gx, 0), wd - 1)) * 4);
            if (gy < 0 || gy >= h || gx < 0 || gx >= wd) v = make_float4(pad, pad, pad, pad);
            xs[swz(r, c)] = v;
        }
        for (int e = tid; e < TH * TW; e += 320) {
            const int r = e / TW, c = e - r * TW;
            const int gy = y0 + r, gx = x0 + c;
            float2 v = ld2(gb + ((size_t)min(gy, h - 1) * wd + min(gx, wd - 1)) * 2);
            if (gy >= h || gx >= wd) v = make_float2(0.f, 0.f);
            gs[e] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < TH / 4; ++it) {
            const int r = it * 4 + rsub;
            float g[4][2];
            {
                const float4* gp = reinterpret_cast<const float4*>(gs + r * TW + 4 * cg);
                const float4 g01 = gp[0], g23 = gp[1];
                g[0][0] = g01.x, g[0][1] = g01.y, g[1][0] = g01.z, g[1][1] = g01.w;
                g[2][0] = g23.x, g[2][1] = g23.y, g[3][0] = g23.z, g[3][1] = g23.w;
            }
            float xv[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = xs[swz(r + ky, 4 * cg + j)];
                xv[j][0] = v.x, xv[j][1] = v.y, xv[j][2] = v.z, xv[j][3] = v.w;
            }
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int co = 0; co < 2; ++co)
#pragma unroll
                        for (int px = 0; px < 4; ++px) acc[(kx * 4 + ci) * 2 + co] += xv[px + kx][ci] * g[px][co];
            if (ky == 0) {
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    acc[40] += g[px][0];
                    acc[41] += g[px][1];
                }
            }
        }
    }
    lane_reduce_scatter<NP>(acc, lane);                    // lane L now holds the wave's sum of accumulator L
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (lane < NACC) partial[(blk * 5 + ky) * NACC + lane] = acc[0];
}

// block a < 200: dw[a] (= tap row a / 40, accumulator a % 40); a = 200, 201: db (from tap row 0's waves)
__global__ __launch_bounds__(256) void conv_wgrad_t542_finish(const float* __restrict__ partial, float* __restrict__ dw,
                                                              float* __restrict__ db, int nblocks, int use_bias,
                                                              int accumulate, float unscale) {
    using namespace t542;
    __shared__ double smem[16];
    const int a = blockIdx.x;
    const int ky = a < 200 ? a / 40 : 0, idx = a < 200 ? a % 40 : 40 + (a - 200);
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += (double)partial[((size_t)i * 5 + ky) * NACC + idx];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst = a < 200 ? dw + a : db + (a - 200);
    if (a >= 200 && !use_bias) s = 0.0;
    s *= (double)unscale;                                  // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// ---------------------------------------------------------------------------------------------
// dw / db of the 5x5 / stride 2 / pad 2 encoder convs (Paragraph down_1/2: 1 -> 1, Line down_1: 1 -> 4,
// down_2: 4 -> 4) with the structure of conv_wgrad_t542: a block of five waves stages an 8 x 32 tile of dy
// and its 19 x 67 window of x once, wave k owns tap row k (5 * CIN * COUT accumulators per lane), one output
// pixel per lane per trip.  conv_wgrad_fast gives a thread ALL its taps from global memory (1 -> COUT:
// 25 strided 4-byte loads per pixel) or one block per tap row (4 -> 4: five passes over x and dy).
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct S2Cfg {
    static constexpr int TH = 8, TW = 32, WH = 2 * TH + 3, WW = 2 * TW + 3;
    static constexpr int NW = 5 * CIN * COUT, NACC = NW + COUT;      // one tap row of dw + db
    static constexpr int NP = ((NACC + 63) / 64) * 64;
};

template <int CIN, int COUT, typename TA>
__global__ __launch_bounds__(320) void conv_wgrad_s2_tiled(const TA* __restrict__ x, const TA* __restrict__ dy,
                                                           float* __restrict__ partial, int h, int wd, int oh, int ow,
                                                           float pad, int tiles_per_block) {
    using C = S2Cfg<CIN, COUT>;
    __shared__ __attribute__((aligned(16))) float xs[C::WH * C::WW * CIN];
    __shared__ float gs[C::TH * C::TW * COUT];
    const int tid = threadIdx.x, lane = tid & 63, ky = tid >> 6;
    const int ox0 = blockIdx.x * C::TW, b = blockIdx.z;
    const int tile0 = blockIdx.y * tiles_per_block, tiles_y = (oh + C::TH - 1) / C::TH;
    const TA* xb = x + (size_t)b * h * wd * CIN;
    const TA* gb = dy + (size_t)b * oh * ow * COUT;
    float acc[C::NP];
#pragma unroll
    for (int a = 0; a < C::NP; ++a) acc[a] = 0.f;
    const int t_end = min(tiles_y, tile0 + tiles_per_block);
    for (int t = tile0; t < t_end; ++t) {
        const int oy0 = t * C::TH;
        __syncthreads();                                   // the previous tile's reads are over
        if constexpr (sizeof(TA) == 4 && (CIN == 1 || CIN == 4)) {
            using VX = std::conditional_t<CIN == 1, float, float4>;        // (the window's loads in flight together)
            stage_batched<C::WH * C::WW, 320, 4, VX>(
                tid,
                [&](int e, bool& inside) {
                    const int r = e / C::WW, c = e - r * C::WW;
                    const int gy = 2 * oy0 - 2 + r, gx = 2 * ox0 - 2 + c;
                    inside = gy >= 0 && gy < h && gx >= 0 && gx < wd;
                    return reinterpret_cast<const VX*>(xb + ((size_t)min(max(gy, 0), h - 1) * wd + min(max(gx, 0), wd - 1)) * CIN);
                },
                [&](int e, VX v, bool inside) {
                    if constexpr (CIN == 1) xs[e] = inside ? v : pad;
                    else *reinterpret_cast<float4*>(xs + e * 4) = inside ? v : make_float4(pad, pad, pad, pad);
                });
        } else {
        for (int e = tid; e < C::WH * C::WW; e += 320) {
            const int r = e / C::WW, c = e - r * C::WW;
            const int gy = 2 * oy0 - 2 + r, gx = 2 * ox0 - 2 + c;
            float v[CIN];
            load_vec<CIN>(xb + ((size_t)min(max(gy, 0), h - 1) * wd + min(max(gx, 0), wd - 1)) * CIN, v);
            const bool in = gy >= 0 && gy < h && gx >= 0 && gx < wd;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) xs[e * CIN + ci] = in ? v[ci] : pad;
        }
        }
        for (int e = tid; e < C::TH * C::TW; e += 320) {
            const int r = e / C::TW, c = e - r * C::TW;
            const int gy = oy0 + r, gx = ox0 + c;
            float v[COUT];
            load_vec<COUT>(gb + ((size_t)min(gy, oh - 1) * ow + min(gx, ow - 1)) * COUT, v);
            const bool in = gy < oh && gx < ow;
#pragma unroll
            for (int co = 0; co < COUT; ++co) gs[e * COUT + co] = in ? v[co] : 0.f;
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < C::TH * C::TW / 64; ++it) {
            const int e = it * 64 + lane, r = e / C::TW, c = e - r * C::TW;
            float g[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) g[co] = gs[e * COUT + co];
            const float* xr = xs + ((2 * r + ky) * C::WW + 2 * c) * CIN;       // window row 2r + ky, columns 2c + kx
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float xv = xr[kx * CIN + ci];
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[(kx * CIN + ci) * COUT + co] += xv * g[co];
                }
            if (ky == 0) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[C::NW + co] += g[co];
            }
        }
    }
    lane_reduce_scatter<C::NP>(acc, lane);                 // lane L holds the wave's sums of accumulators base(L) + r
    int base = 0;
#pragma unroll
    for (int sft = 0; sft < 6; ++sft)
        if (lane & (32 >> sft)) base += C::NP >> (sft + 1);
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* out = partial + (blk * 5 + ky) * C::NP;
#pragma unroll
    for (int q = 0; q < C::NP / 64; ++q) out[base + q] = acc[q];
}

// block a < 25 * CIN * COUT: dw[a] (tap row a / NW, accumulator a % NW); then COUT blocks for db
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_s2_finish(const float* __restrict__ partial, float* __restrict__ dw,
                                                            float* __restrict__ db, int nblocks, int use_bias,
                                                            int accumulate, float unscale) {
    using C = S2Cfg<CIN, COUT>;
    __shared__ double smem[16];
    const int a = blockIdx.x, ndw = 5 * C::NW;
    const int ky = a < ndw ? a / C::NW : 0, idx = a < ndw ? a % C::NW : C::NW + (a - ndw);
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += (double)partial[((size_t)i * 5 + ky) * C::NP + idx];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst = a < ndw ? dw + a : db + (a - ndw);
    if (a >= ndw && !use_bias) s = 0.0;
    s *= (double)unscale;                                  // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

template <int CIN, int COUT>
int launch_wgrad_s2(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                    double pad_value, int use_bias, int accumulate) {
    using C = S2Cfg<CIN, COUT>;
    const int tiles_x = (d.ow + C::TW - 1) / C::TW, tiles_y = (d.oh + C::TH - 1) / C::TH;
    int per_block = 1;                                      // ~1024 blocks: a few tiles of one column strip each
    while (per_block < tiles_y && (size_t)tiles_x * ((tiles_y + per_block - 1) / per_block) * d.n > 1024) ++per_block;
    const dim3 grid(tiles_x, (tiles_y + per_block - 1) / per_block, d.n);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, (size_t)nblocks * 5 * C::NP * sizeof(float), &rc);
    if (rc) return rc;
    UOCR_DISPATCH_TA(ctx, dtype, {
        hipLaunchKernelGGL((conv_wgrad_s2_tiled<CIN, COUT, TA>), grid, dim3(320), 0, ctx->stream, (const TA*)x,
                           (const TA*)dy, partial, d.h, d.w, d.oh, d.ow, (float)pad_value, per_block);
    });
    UOCR_LAUNCH_CHECK(ctx);
    FinishDesc fd{};                                     // recorded when a deferred group is open (finish_group.h)
    fd.kind = FIN_TAPROWS;
    fd.partial = partial;
    fd.nblocks = nblocks;
    fd.ncols = fd.group_cols = 5 * C::NP;
    fd.row_stride = (size_t)5 * C::NP;
    fd.dw = (float*)dw, fd.db = (float*)db;
    fd.use_bias = use_bias, fd.accumulate = accumulate;
    fd.unscale = (float)uocr_grad_unscale(dtype);
    fd.p[0] = C::NP, fd.p[1] = C::NW, fd.p[2] = COUT;
    if (uocr_finish_defer(ctx, fd)) return UOCR_OK;
    hipLaunchKernelGGL((conv_wgrad_s2_finish<CIN, COUT>), dim3(5 * C::NW + COUT), dim3(256), 0, ctx->stream,
                       (const float*)partial, (float*)dw, (float*)db, nblocks, use_bias, accumulate,
                       (float)uocr_grad_unscale(dtype));
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

template <int KH, int KW, int CIN, int COUT, int SH, int SW, int COB, int PY, int DPY, int KYR, int WCOB, int WPY, int FPX,
          int DPX>
struct FastConv {
    static bool match(const ConvDims& d) {
        return d.kh == KH && d.kw == KW && d.cin == CIN && d.cout == COUT && d.sh == SH && d.sw == SW;
    }
    static FastDims dims(const ConvDims& d) { return FastDims{d.n, d.h, d.w, d.oh, d.ow, d.ph, d.pw}; }

    static int fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                   double pad, int use_bias, int act, double alpha) {
        UOCR_DISPATCH_TA(ctx, dtype, {
            if constexpr (FPX > 0) {
                const int groups = (d.ow + FPX - 1) / FPX;
                const dim3 grid((groups + 63) / 64, (d.oh + 3) / 4, d.n), block(64, 4);
                hipLaunchKernelGGL((conv_fwd_px<KH, KW, CIN, COUT, SH, SW, FPX, TA>), grid, block, 0, ctx->stream,
                                   (const TA*)x, (const float*)w, (const float*)b, (TA*)y, dims(d), (float)pad,
                                   use_bias, act, (float)alpha);
            } else {
                constexpr int PXW = 64 / (COUT / COB);
                const dim3 grid((d.ow + PXW - 1) / PXW, (d.oh + 4 * PY - 1) / (4 * PY), d.n), block(64, 4);
                hipLaunchKernelGGL((conv_fwd_fast<KH, KW, CIN, COUT, SH, SW, COB, PY, TA>), grid, block, 0, ctx->stream,
                                   (const TA*)x, (const float*)w, (const float*)b, (TA*)y, dims(d), (float)pad,
                                   use_bias, act, (float)alpha);
            }
        });
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }

    static int dgrad(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d,
                     const ActMask& mask) {
        UOCR_DISPATCH_TA(ctx, dtype, {
            if constexpr (DPX > 0) {
                static_assert(DPX == 0 || (SH == 1 && SW == 1), "conv_dgrad_px is stride 1 only");
                const int groups = (d.w + DPX - 1) / DPX;
                const dim3 grid((groups + 63) / 64, (d.h + 3) / 4, d.n), block(64, 4);
                hipLaunchKernelGGL((conv_dgrad_px<KH, KW, CIN, COUT, DPX, TA>), grid, block, 0, ctx->stream,
                                   (const TA*)dy, (const float*)w, (TA*)dx, dims(d), (const TA*)mask.y, mask.act,
                                   (float)mask.alpha);
            } else {
                const dim3 grid((d.w + 63) / 64, (d.h + 4 * DPY - 1) / (4 * DPY), d.n), block(64, 4);
                hipLaunchKernelGGL((conv_dgrad_fast<KH, KW, CIN, COUT, SH, SW, DPY, TA>), grid, block, 0, ctx->stream,
                                   (const TA*)dy, (const float*)w, (TA*)dx, dims(d), (const TA*)mask.y, mask.act,
                                   (float)mask.alpha);
            }
        });
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }

    static int wgrad(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                     double pad, int use_bias, int accumulate) {
        using C = WgradCfg<KH, KW, CIN, COUT, SH, SW, KYR, WCOB>;
        // bands of output rows: ~512 (or 64) blocks per (tap group, channel group), at least 4 rows each
        constexpr int RQ = 4;             // rows consumed per block iteration (one per wave)
        // (a wave ends in a reduce-scatter of its NP accumulators over the 64 lanes, ~3 instructions per accumulator: with 128
        // of them a band of 4 rows -- one row per wave -- spent four fifths of its time there.  Measured per band target,
        // 32 x 64 x 128 x 4 -> 4 stride 2: 27.8 us at 512, 17.4 at 128, 17.3 at 64; the 64-accumulator one-channel kernels
        // are fastest at 512: 13.1 us against 18.5 at 256 and 15.5 at 1024)
        const int bands = ctx->opt_wgrad_bands > 0 ? ctx->opt_wgrad_bands : (C::NP >= 128 ? 64 : 512);
        int rows = (d.n * d.oh + bands - 1) / bands;
        rows = ((rows + RQ - 1) / RQ) * RQ;
        if (rows > d.oh) rows = ((d.oh + RQ - 1) / RQ) * RQ;
        const int nbands = (d.oh + rows - 1) / rows;
        const int nblocks = nbands * d.n, ngroups = C::KYG * C::OCG;
        const size_t bytes = (size_t)nblocks * ngroups * C::NP * sizeof(float);
        int rc = UOCR_OK;
        float* partial = uocr_partial_buffer(ctx, bytes, &rc);
        if (rc) return rc;
        UOCR_DISPATCH_TA(ctx, dtype, {
            hipLaunchKernelGGL((conv_wgrad_fast<KH, KW, CIN, COUT, SH, SW, KYR, WCOB, WPY, (CIN >= 4), TA>),
                               dim3(nblocks, ngroups), dim3(64, 4), 0, ctx->stream, (const TA*)x, (const TA*)dy,
                               partial, dims(d), (float)pad, rows, nbands);
        });
        UOCR_LAUNCH_CHECK(ctx);
        FinishDesc fd{};                                 // recorded when a deferred group is open (finish_group.h)
        fd.kind = FIN_FAST;
        fd.partial = partial;
        fd.nblocks = nblocks;
        fd.ncols = ngroups * C::NP;
        fd.group_cols = C::NP;
        fd.group_stride = (size_t)nblocks * C::NP;
        fd.row_stride = C::NP;
        fd.dw = (float*)dw, fd.db = (float*)db;
        fd.use_bias = use_bias, fd.accumulate = accumulate;
        fd.unscale = (float)uocr_grad_unscale(dtype);
        fd.p[0] = C::NP, fd.p[1] = C::NW, fd.p[2] = KW, fd.p[3] = CIN, fd.p[4] = COUT, fd.p[5] = KYR, fd.p[6] = WCOB;
        if (uocr_finish_defer(ctx, fd)) return UOCR_OK;
        hipLaunchKernelGGL((conv_wgrad_fast_finish<KH, KW, CIN, COUT, KYR, WCOB, C::NW, C::NP>),
                           dim3(C::NACC, ngroups), dim3(256), 0, ctx->stream, (const float*)partial, (float*)dw,
                           (float*)db, nblocks, use_bias, accumulate, (float)uocr_grad_unscale(dtype));
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
};

// the my_model shapes:  KH KW CIN COUT SH SW | legacy fwd: COB PY | legacy dgrad: PY | wgrad: KYR COB PX |
//                        row-loop kernels: fwd PX, dgrad PX (0 = use the legacy kernel)
#define UOCR_FAST_CONVS(X)                                                                         \
    X(3, 3, 1, 16, 1, 1, 4, 2, 4, 3, 16, 1, 0, 4) /* Monochrome conv_1: dgrad re-reads 16-ch dy */ \
    X(3, 3, 16, 1, 1, 1, 1, 4, 1, 1, 1, 1, 4, 2)  /* Monochrome conv_2: fwd/wgrad re-read 16-ch x */ \
    X(5, 5, 1, 1, 2, 2, 1, 1, 1, 5, 1, 4, 4, 0)   /* Paragraph down_1/2 */                         \
    X(5, 5, 1, 1, 1, 1, 1, 4, 4, 5, 1, 8, 4, 8)   /* Paragraph up_2, up_1, end (PX 8: fwd 22 -> 24, dx 22 -> 17, dw 35 -> 28 us) */                \
    X(5, 5, 1, 4, 2, 2, 4, 1, 1, 5, 4, 1, 4, 0)   /* Line down_1 */                                \
    X(5, 5, 4, 4, 2, 2, 4, 2, 1, 1, 4, 2, 4, 0)   /* Line down_2 */                                \
    X(5, 5, 4, 4, 1, 1, 4, 4, 4, 1, 4, 4, 4, 4)   /* Line up_2, up_1 */                            \
    X(5, 5, 4, 2, 1, 1, 2, 4, 4, 1, 2, 4, 4, 4)   /* Line end */                                   \
    X(5, 3, 1, 64, 2, 1, 4, 1, 1, 5, 8, 1, 0, 0)  /* Char conv_1 */

}  // namespace

bool uocr_conv_fast_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, const void* p0, const void* p1,
                             const void* p2) {
    const int base = UOCR_DTYPE_BASE(dtype);
    if ((base != UOCR_F32 && base != UOCR_F16) || !ctx->opt_fast) return false;
    // p0, p1: the two activation tensors of the call (4-element accesses); p2: a float32 parameter tensor
    if (!uocr_aligned_act(p0, dtype) || !uocr_aligned_act(p1, dtype) || !uocr_aligned_act(p2, UOCR_F32)) return false;
#define X(KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX) \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::match(d)) return true;
    UOCR_FAST_CONVS(X)
#undef X
    return false;
}

namespace {
inline bool is_c16_same(const ConvDims& d, int cin, int cout) {
    return d.kh == 3 && d.kw == 3 && d.cin == cin && d.cout == cout && d.sh == 1 && d.sw == 1 && d.oh == d.h &&
           d.ow == d.w;
}
}  // namespace

// (the 16-channel quad-lane kernels and the Char conv_1 dx kernel exist in float32 only: binary16 runs those
// shapes -- none of them is on the page nets' fused path -- through the FastConv kernels of the table)
int uocr_conv_fwd_fast(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                       const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha) {
    const bool f32 = UOCR_DTYPE_BASE(dtype) == UOCR_F32;
    if (f32 && is_c16_same(d, 1, 16)) {
        constexpr int PY = 2;
        const dim3 grid((d.w + 15) / 16, (d.h + 4 * PY - 1) / (4 * PY), d.n), block(64, 4);
        hipLaunchKernelGGL((conv_c16_expand<3, 3, PY, false>), grid, block, 0, ctx->stream, (const float*)x,
                           (const float*)w, (const float*)b, (float*)y, d.n, d.h, d.w, d.ph, d.pw, (float)pad_value,
                           use_bias, act, (float)act_alpha, (const float*)nullptr, (int)UOCR_ACT_NONE, 0.f);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    if (f32 && is_c16_same(d, 16, 1)) {
        constexpr int PY = 4;
        const dim3 grid((d.w + 15) / 16, (d.h + 4 * PY - 1) / (4 * PY), d.n), block(64, 4);
        hipLaunchKernelGGL((conv_c16_reduce<3, 3, PY, false>), grid, block, 0, ctx->stream, (const float*)x,
                           (const float*)w, (const float*)b, (float*)y, d.n, d.h, d.w, d.ph, d.pw, (float)pad_value,
                           use_bias, act, (float)act_alpha, (const float*)nullptr, (int)UOCR_ACT_NONE, 0.f);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
#define X(KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX)                        \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::match(d))      \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::fwd(ctx, dtype, x, w, b, y, d, pad_value, use_bias, act, \
                                                                        act_alpha);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}

int uocr_conv_dgrad_fast(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d,
                         const ActMask& mask) {
    const bool f32 = UOCR_DTYPE_BASE(dtype) == UOCR_F32;
    if (f32 && is_c16_same(d, 16, 1)) {
        constexpr int PY = 2;
        const dim3 grid((d.w + 15) / 16, (d.h + 4 * PY - 1) / (4 * PY), d.n), block(64, 4);
        hipLaunchKernelGGL((conv_c16_expand<3, 3, PY, true>), grid, block, 0, ctx->stream, (const float*)dy,
                           (const float*)w, (const float*)nullptr, (float*)dx, d.n, d.h, d.w, d.ph, d.pw, 0.f, 0,
                           (int)UOCR_ACT_NONE, 0.f, (const float*)mask.y, mask.act, (float)mask.alpha);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    if (f32 && is_c16_same(d, 1, 16)) {
        constexpr int PY = 4;
        const dim3 grid((d.w + 15) / 16, (d.h + 4 * PY - 1) / (4 * PY), d.n), block(64, 4);
        hipLaunchKernelGGL((conv_c16_reduce<3, 3, PY, true>), grid, block, 0, ctx->stream, (const float*)dy,
                           (const float*)w, (const float*)nullptr, (float*)dx, d.n, d.h, d.w, d.ph, d.pw, 0.f, 0,
                           (int)UOCR_ACT_NONE, 0.f, (const float*)mask.y, mask.act, (float)mask.alpha);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    if (f32 && d.kh == 5 && d.kw == 3 && d.sh == 2 && d.sw == 1 && d.ph == 0 && d.pw == 1 && d.cin == 1 && d.cout == 64 &&
        d.h <= 65535 && d.n <= 65535) {                   // (rows and images are grid dimensions y and z)
        hipLaunchKernelGGL(conv_dgrad_c64s2, dim3((unsigned)((d.w + 15) / 16), d.h, d.n), dim3(256), 0, ctx->stream,
                           (const float*)dy, (const float*)w, (float*)dx, d.n, d.h, d.w, d.oh, d.ow,
                           (const float*)mask.y, mask.act, (float)mask.alpha);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    if (d.kh == 5 && d.kw == 5 && d.sh == 2 && d.sw == 2 && d.ph == 2 && d.pw == 2 &&
        ((d.cin == 1 && (d.cout == 1 || d.cout == 4)) || (d.cin == 4 && d.cout == 4))) {
        const dim3 grid(((d.w + 1) / 2 + 63) / 64, ((d.h + 1) / 2 + 3) / 4, d.n), block(64, 4);
        const FastDims fd{d.n, d.h, d.w, d.oh, d.ow, d.ph, d.pw};
        UOCR_DISPATCH_TA(ctx, dtype, {
            auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, (const TA*)dy, (const float*)w, (TA*)dx, fd,
                                   (const TA*)mask.y, mask.act, (float)mask.alpha);
            };
            if (d.cin == 4) launch(conv_dgrad_s2<4, 4, TA>);
            else if (d.cout == 1) launch(conv_dgrad_s2<1, 1, TA>);
            else launch(conv_dgrad_s2<1, 4, TA>);
        });
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
#define X(KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX)                   \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::match(d)) \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::dgrad(ctx, dtype, dy, w, dx, d, mask);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}

int uocr_conv_wgrad_fast(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                         const ConvDims& d, double pad_value, int use_bias, int accumulate) {
    const bool f32 = UOCR_DTYPE_BASE(dtype) == UOCR_F32;
    if (f32 && is_c16_same(d, 16, 1) && pad_value == 0.0) {
        constexpr int NA = 3 * 3 * 4 + 1;
        int rows = (d.n * d.h + 2047) / 2048;
        rows = ((rows + 3) / 4) * 4;
        if (rows > d.h) rows = ((d.h + 3) / 4) * 4;
        const int nbands = (d.h + rows - 1) / rows, nblocks = nbands * d.n;
        int rc = uocr_need_workspace(ctx, (size_t)nblocks * 4 * NA * sizeof(float));
        if (rc) return rc;
        float* partial = (float*)ctx->workspace;
        hipLaunchKernelGGL((conv_c16_wgrad<3, 3>), dim3(nblocks), dim3(64, 4), 0, ctx->stream, (const float*)x,
                           (const float*)dy, partial, d.h, d.w, d.ph, d.pw, rows, nbands);
        UOCR_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((conv_c16_wgrad_finish<3, 3>), dim3(NA, 4), dim3(256), 0, ctx->stream,
                           (const float*)partial, (float*)dw, (float*)db, nblocks, use_bias, accumulate);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    if (d.kh == 5 && d.kw == 5 && d.sh == 2 && d.sw == 2 && d.ph == 2 && d.pw == 2 && ctx->opt_tiled == 1) {
        // (measured against conv_wgrad_fast on one box: 1 -> 4 20 vs 25 us; 1 -> 1 17 vs 15 and 4 -> 4 34 vs 29 us
        // lose -- too little work per staged tile -- and stay on the register kernels)
        if (d.cin == 1 && d.cout == 4)
            return launch_wgrad_s2<1, 4>(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    }
    if (d.kh == 5 && d.kw == 5 && d.cin == 4 && d.cout == 2 && d.sh == 1 && d.sw == 1 && d.ph == 2 && d.pw == 2) {
        const int tiles_x = (d.w + t542::TW - 1) / t542::TW, tiles_y = (d.h + t542::TH - 1) / t542::TH;
        int per_block = 1;                                  // ~1024 blocks: a few tiles of one column strip each
        // (measured at 32 x 256 x 512: 512 blocks 58 us, 1024 57, 2048 65, 4096 74)
        while (per_block < tiles_y && (size_t)tiles_x * ((tiles_y + per_block - 1) / per_block) * d.n > 1024) ++per_block;
        const dim3 grid(tiles_x, (tiles_y + per_block - 1) / per_block, d.n);
        const int nblocks = (int)(grid.x * grid.y * grid.z);
        int rc = UOCR_OK;
        float* partial = uocr_partial_buffer(ctx, (size_t)nblocks * 5 * t542::NACC * sizeof(float), &rc);
        if (rc) return rc;
        UOCR_DISPATCH_TA(ctx, dtype, {
            hipLaunchKernelGGL((conv_wgrad_t542<TA>), grid, dim3(320), 0, ctx->stream, (const TA*)x, (const TA*)dy,
                               partial, d.h, d.w, (float)pad_value, per_block);
        });
        UOCR_LAUNCH_CHECK(ctx);
        FinishDesc fd{};                                 // recorded when a deferred group is open (finish_group.h)
        fd.kind = FIN_TAPROWS;
        fd.partial = partial;
        fd.nblocks = nblocks;
        fd.ncols = fd.group_cols = 5 * t542::NACC;
        fd.row_stride = (size_t)5 * t542::NACC;
        fd.dw = (float*)dw, fd.db = (float*)db;
        fd.use_bias = use_bias, fd.accumulate = accumulate;
        fd.unscale = (float)uocr_grad_unscale(dtype);
        fd.p[0] = t542::NACC, fd.p[1] = 40, fd.p[2] = 2;
        if (uocr_finish_defer(ctx, fd)) return UOCR_OK;
        hipLaunchKernelGGL(conv_wgrad_t542_finish, dim3(202), dim3(256), 0, ctx->stream, (const float*)partial,
                           (float*)dw, (float*)db, nblocks, use_bias, accumulate, (float)uocr_grad_unscale(dtype));
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
#define X(KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX)                   \
    if (FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::match(d)) \
        return FastConv<KH, KW, CIN, COUT, SH, SW, COB, PY, DPY, KYR, WCOB, WPY, FPX, DPX>::wgrad(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, \
                                                                          accumulate);
    UOCR_FAST_CONVS(X)
#undef X
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "no fast conv kernel for this shape");
}
