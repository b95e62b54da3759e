// Small-channel convolutions of the page nets in float32 on v_mfma_f32_16x16x4_f32: the vertical-Toeplitz
// formulation of conv_h16.hip with one float per lane and MFMA (K = 4 window rows x one element of the row).
// (reference layers: nn/layers/convolutional.py:62-145, nn/layers/upsample.py:21-39, my_model/model.py:194-247)
//
// The vector-ALU kernels of these layers (conv_fast.hip, conv_tiled.hip, conv_up.hip) run 100-400 FMAs per pixel
// at 30-45 % of the packed-FMA rate.  As an im2col GEMM the layers would use 1-4 of the 16 result columns; with
// result rows = (vertical shift dy, output channel) every lane of the MFMA result is a real output:
//       D[(dy, co), col] = sum_{ty', e} Wt[(dy, co), (ty', e)] * X[row0*S + ty'][col*S*C + e]
// e = (tap column, input channel) runs over the KW*C floats of a window row (channels-last: contiguous),
// ty' over the (DY-1)*S + KH window rows of the DY output rows.  MFMA (ib, e): k-group kq = window row 4*ib + kq.
// Executed / useful multiply-adds: 2.4x (5x5 4->2), 1.6x (its backward-data), so 25.6 / 17 us of matrix-core time
// at batch 32 x 256 x 512 where the vector kernels take 37 / 51 us.  Measured (profiles/r02_t32_*): 58 / 39 us,
// 36.7 us against 24.1 for the upsample+conv backward-data, 20 / 20 / 15 us against 28 / 17 / 10 for the 1-channel
// layers: float32 MFMAs and vector instructions do not overlap on a SIMD, so staging,
// epilogue and address arithmetic (4 vector instructions per MFMA in the backward-data kernel) add to the MFMA
// time, and with 3-6 blocks per CU the barrier-separated load / compute phases leave the matrix pipe 40 % busy.
// Hence the default ("t32" option = 2) uses this file for the 4-channel backward-data only; the other
// instantiations stay selectable (bits 1 / 4 / 8 / 16 / 32) and tested.  Results are exact float32 FMA chains
// (the summation order differs from the reference's: tests at 1e-5 normalised, as for the other fused kernels).
// A zero weight still multiplies what the tile holds there: inputs must be finite.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include "conv_toeplitz.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= 0.f ? 1.f : 0.f);
        case UOCR_ACT_LEAKY: return v * ((v >= 0.f ? 1.f : 0.f) + alpha * (v < 0.f ? 1.f : 0.f));
        case UOCR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

template <int C_, int COUT_, int KH_, int KW_, int S_, int MODE_, int DY_ = 0>
struct Geo {
    static constexpr int C = C_, COUT = COUT_, KH = KH_, KW = KW_, S = S_, MODE = MODE_;
    static constexpr int U = (MODE == M_UPFWD || MODE == M_S2DGRAD) ? 2 : 1;   // output pixels per position and axis
    static constexpr int NCO = U * U * COUT;                   // result rows per position: (phase, co)
    static constexpr int DY = DY_ ? DY_ : 16 / NCO;            // position rows of one MFMA chain
    static constexpr int ROWS = (DY - 1) * S + KH;             // window rows of one chain
    static constexpr int IB = (ROWS + 3) / 4;                  // row quads
    static constexpr int Q = KW * C;                           // floats per window row
    static constexpr int NM = IB * Q;                          // MFMAs per chain
    static constexpr int BC = S == 1 ? 64 : 32;                // position columns of a block tile
    static constexpr int NCG = BC / 16;                        // chains per row band
    static constexpr int BRT = S == 1 ? 32 : 16;               // target tile height
    static constexpr int RPW = BRT / (4 * DY) > 0 ? BRT / (4 * DY) : 1;   // row bands per wave
    static constexpr int BR = 4 * RPW * DY;                    // position rows of a block tile
    static constexpr int IH = (BR - 1) * S + KH;               // staged input rows
    static constexpr int IHA = (BR - DY) * S + 4 * IB;         // rows a chain may address (the rest stays zero)
    static constexpr int PXU = 4 / C;                          // pixels per 16-byte staging unit
    static constexpr int UW = ((BC - 1) * S * C + Q + 3) / 4;  // staging units per tile row
    // row stride in floats (a multiple of 4: 16-byte staging writes).  The 16 lanes of a k-group read 16 / 8 / 4
    // bytes each (C = 4 / 2 / 1): 4 channels: a k-group covers all banks by itself; 2 channels: the two k-groups of
    // a half wave 32 banks apart; 1 channel: the four k-groups 16 banks apart
    static constexpr int RS = C == 1 ? round_up(UW * 4 - 16, 64) + 16 : C == 2 ? round_up(UW * 4 - 32, 64) + 32 : UW * 4;
    static constexpr int RPP = 256 / UW;                       // tile rows staged per pass of the block
    static constexpr int NPASS = (IH + RPP - 1) / RPP;
    static constexpr int NW = (MODE == M_UPDGRAD || MODE == M_UPFWD || MODE == M_S2DGRAD ? 25 : KH * KW) * C * COUT;
    static_assert(16 % NCO == 0 && DY * NCO <= 16 && 4 % C == 0 && UW <= 256 && RS >= UW * 4, "unsupported geometry");
};

// the Q floats of one window row segment, with the widest aligned LDS reads the channel count allows
template <class G>
__device__ __forceinline__ void load_row(float (&bv)[G::Q], const float* p) {
    if constexpr (G::C == 4) {
#pragma unroll
        for (int j = 0; j < G::Q / 4; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(p + 4 * j);
            bv[4 * j] = t.x, bv[4 * j + 1] = t.y, bv[4 * j + 2] = t.z, bv[4 * j + 3] = t.w;
        }
    } else if constexpr (G::C == 2) {
#pragma unroll
        for (int j = 0; j < G::Q / 2; ++j) {
            const float2 t = *reinterpret_cast<const float2*>(p + 2 * j);
            bv[2 * j] = t.x, bv[2 * j + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < G::Q; ++e) bv[e] = p[e];
    }
}

// Persistent blocks over a flat tile index (image, tile row, column strip): block b takes tiles b, b + grid, ...
// A tile is BR x BC positions; wave w owns its row bands w*RPW .. w*RPW + RPW - 1 (DY rows each), NCG chains of 16
// columns per band.
//   in      [n][h_in][w_in][C]      float32
//   out     [n][h_out][w_out][COUT] float32;  mask_y (backward-data: the activation OUTPUT of the layer below, same
//           shape as out) multiplies the result by act'(y)
template <class G>
__global__ __launch_bounds__(256) void conv_t32_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const float* __restrict__ mask_y, int h_in, int w_in, int h_out,
                                                       int w_out, int ph, int pw, int tiles_x, int tiles_y, int ntiles,
                                                       float pad, int use_bias, int act, float alpha, int mask_act,
                                                       float mask_alpha) {
    constexpr int C = G::C, COUT = G::COUT, S = G::S, RS = G::RS;
    __shared__ __attribute__((aligned(16))) float tile[G::IHA * RS];
    __shared__ float wl[G::NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;

    // the rows / floats no load ever writes must read as zero (their weights are zero, 0 * garbage is not)
    for (int i = tid; i < G::IHA * RS; i += 256) tile[i] = 0.f;
    for (int i = tid; i < G::NW; i += 256) wl[i] = w[i];
    __syncthreads();

    // weight operand: row m = lane % 16 = (dy, (phase,) co), k-group kq = window row 4*ib + kq, element e of the row
    float wa[G::NM];
    {
        const int m = n, dyi = m / G::NCO, co = m % G::NCO;
#pragma unroll
        for (int ib = 0; ib < G::IB; ++ib)
#pragma unroll
            for (int e = 0; e < G::Q; ++e) {
                const int ty = 4 * ib + kq - dyi * S, tx = e / C, ci = e % C;
                const bool live = dyi < G::DY && ty >= 0 && ty < G::KH;
                wa[ib * G::Q + e] = live ? weight_of<G>(wl, min(max(ty, 0), G::KH - 1), tx, ci, co) : 0.f;
            }
    }
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = use_bias ? bias[(4 * kq + i) % COUT] : 0.f;
    // staging role of this thread: 16-byte unit su of tile rows sr, sr + RPP, ...
    const int sr = tid / G::UW, su = tid - sr * G::UW;
    const bool stager = sr < G::RPP;

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int c_begin = strip * G::BC, r0 = trow * G::BR;
        const float* inb = in + (size_t)img * h_in * w_in * C;
        const size_t out_img = (size_t)img * h_out * w_out * COUT;
        __syncthreads();                                 // the previous tile's reads are over (and the zero fill)
        // ---- stage the input tile: pixel (iy0 + r, ix0 + c) -> tile[r][c*C ..], padding outside the image.
        // The column part of a thread's address and its in-image bits are the same for all its rows.
        if (stager) {
            const int iy0 = r0 * S - ph, gx0 = c_begin * S - pw + su * G::PXU;
            bool in_px[G::PXU];
            bool all_in = true, any_in = false;
#pragma unroll
            for (int k = 0; k < G::PXU; ++k) {
                in_px[k] = (unsigned)(gx0 + k) < (unsigned)w_in;
                all_in = all_in && in_px[k];
                any_in = any_in || in_px[k];
            }
            float4 v[G::NPASS];
#pragma unroll
            for (int k = 0; k < G::NPASS; ++k) {
                const int gy = iy0 + sr + k * G::RPP;
                const bool row_ok = (unsigned)gy < (unsigned)h_in;
                const float* src = inb + (size_t)min(max(gy, 0), h_in - 1) * w_in * C;
                v[k] = make_float4(pad, pad, pad, pad);
                if (row_ok && all_in) {
                    v[k] = *reinterpret_cast<const float4*>(src + (size_t)gx0 * C);      // (4-byte aligned at least)
                } else if (row_ok && any_in) {                                          // the image edge cuts the unit
                    float* vf = reinterpret_cast<float*>(&v[k]);
#pragma unroll
                    for (int px = 0; px < G::PXU; ++px)
                        if (in_px[px]) {
#pragma unroll
                            for (int d = 0; d < C; ++d) vf[px * C + d] = src[(size_t)(gx0 + px) * C + d];
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < G::NPASS; ++k) {
                const int r = sr + k * G::RPP;
                if (r < G::IH) {
                    if constexpr (RS % 4 == 0) {
                        *reinterpret_cast<float4*>(tile + r * RS + su * 4) = v[k];
                    } else {
                        float* dst = tile + r * RS + su * 4;
                        dst[0] = v[k].x, dst[1] = v[k].y, dst[2] = v[k].z, dst[3] = v[k].w;
                    }
                }
            }
        }
        __syncthreads();
        // ---- chains
#pragma unroll
        for (int s = 0; s < G::RPW; ++s) {
            const int rb = (wv * G::RPW + s) * G::DY;    // first position row of the band, relative to the tile
            // steps (chain cg, row quad ib): the B values of step s + 1 are read from LDS before the MFMAs of step s
            // are issued (the compiler orders a read right before its use otherwise, and every batch of MFMAs then
            // waits out the LDS latency)
            const float* band = tile + (rb * S + kq) * RS + n * S * C;
            float bcur[G::Q], bnxt[G::Q];
            load_row<G>(bcur, band);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int step = 0; step < G::NCG * G::IB; ++step) {
                const int cg = step / G::IB, ib = step % G::IB;
                if (step + 1 < G::NCG * G::IB) {
                    const int cg1 = (step + 1) / G::IB, ib1 = (step + 1) % G::IB;
                    load_row<G>(bnxt, band + cg1 * 16 * S * C + ib1 * 4 * RS);
                }
#pragma unroll
                for (int e = 0; e < G::Q; ++e) acc = mfma4(wa[ib * G::Q + e], bcur[e], acc);
#pragma unroll
                for (int e = 0; e < G::Q; ++e) bcur[e] = bnxt[e];
                if (ib != G::IB - 1) continue;
                const f32x4 res = acc;
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                // lane (column n, kq): rows m = 4kq + i of the result = (dy, (phase,) co)
                const int col = c_begin + cg * 16 + n;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_apply(res[i] + bias4[i], act, alpha);
                if constexpr (COUT == 4) {               // the 4 channels of one pixel; depth to space: phase = kq
                    const int row = G::U == 2 ? 2 * (r0 + rb) + (kq >> 1) : r0 + rb + kq;
                    const int ocol = G::U == 2 ? 2 * col + (kq & 1) : col;
                    if (row >= h_out || ocol >= w_out || (G::U == 1 && kq >= G::DY)) continue;
                    const size_t off = out_img + ((size_t)row * w_out + ocol) * 4;
                    if (mask_act != UOCR_ACT_NONE) {
                        const float4 my = *reinterpret_cast<const float4*>(mask_y + off);
                        v[0] *= act_grad_from_output<float>(my.x, mask_act, mask_alpha);
                        v[1] *= act_grad_from_output<float>(my.y, mask_act, mask_alpha);
                        v[2] *= act_grad_from_output<float>(my.z, mask_act, mask_alpha);
                        v[3] *= act_grad_from_output<float>(my.w, mask_act, mask_alpha);
                    }
                    *reinterpret_cast<float4*>(out + off) = make_float4(v[0], v[1], v[2], v[3]);
                } else if constexpr (COUT == 2) {        // rows 2kq, 2kq + 1, both channels each
                    static_assert(COUT != 2 || G::U == 1, "2 channels: plain store only");
                    if (col >= w_out) continue;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = r0 + rb + 2 * kq + hh;
                        if (row >= h_out || 2 * kq + hh >= G::DY) continue;
                        const size_t off = out_img + ((size_t)row * w_out + col) * 2;
                        float v0 = v[2 * hh], v1 = v[2 * hh + 1];
                        if (mask_act != UOCR_ACT_NONE) {
                            const float2 my = *reinterpret_cast<const float2*>(mask_y + off);
                            v0 *= act_grad_from_output<float>(my.x, mask_act, mask_alpha);
                            v1 *= act_grad_from_output<float>(my.y, mask_act, mask_alpha);
                        }
                        *reinterpret_cast<float2*>(out + off) = make_float2(v0, v1);
                    }
                } else {                                 // 1 channel: rows 4kq + i, or (depth to space) row kq, phase i
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int dyi = G::U == 2 ? kq : 4 * kq + i;
                        const int row = G::U == 2 ? 2 * (r0 + rb + dyi) + (i >> 1) : r0 + rb + dyi;
                        const int ocol = G::U == 2 ? 2 * col + (i & 1) : col;
                        if (row >= h_out || ocol >= w_out || dyi >= G::DY) continue;
                        const size_t off = out_img + (size_t)row * w_out + ocol;
                        float vi = v[i];
                        if (mask_act != UOCR_ACT_NONE) vi *= act_grad_from_output<float>(mask_y[off], mask_act, mask_alpha);
                        out[off] = vi;
                    }
                }
            }
        }
    }
}

template <class G>
int launch_t32(uocr_ctx* ctx, const void* in, const void* w, const void* bias, void* out, const void* mask_y, int n,
               int h_in, int w_in, int h_out, int w_out, int ph, int pw, float pad, int use_bias, int act, float alpha,
               int mask_act, float mask_alpha) {
    static int resident = 0;                             // blocks of this kernel one CU holds
    if (resident == 0) {
        int nb = 0;
        UOCR_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_t32_kernel<G>, 256, 0));
        resident = nb > 0 ? nb : 1;
    }
    const int hp = (h_out + G::U - 1) / G::U, wp = (w_out + G::U - 1) / G::U;      // the position grid
    const int tiles_x = (wp + G::BC - 1) / G::BC, tiles_y = (hp + G::BR - 1) / G::BR;
    const long ntiles = (long)n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31) && (long)h_in * w_in * G::C < (1l << 31));
    const long cap = (long)ctx->cu_count * resident * 4;      // (4 x: conv_h16.hip, the tail when lanes share the CUs)
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    hipLaunchKernelGGL(conv_t32_kernel<G>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)in, (const float*)w,
                       (const float*)bias, (float*)out, (const float*)mask_y, h_in, w_in, h_out, w_out, ph, pw, tiles_x,
                       tiles_y, (int)ntiles, pad, use_bias, act, alpha, mask_act, mask_alpha);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

inline bool same5x5(const ConvDims& d) {
    return d.kh == 5 && d.kw == 5 && d.sh == 1 && d.sw == 1 && d.ph == 2 && d.pw == 2 && d.oh == d.h && d.ow == d.w;
}
inline bool half5x5(const ConvDims& d) {                 // the encoder convs: 5x5 / stride 2 / padding 2
    return d.kh == 5 && d.kw == 5 && d.sh == 2 && d.sw == 2 && d.ph == 2 && d.pw == 2 && d.oh == (d.h + 1) / 2 &&
           d.ow == (d.w + 1) / 2;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// which: 0 forward, 1 backward-data.  Bits of the "t32" option: 1 forward / 2 backward-data / 4 upsample+conv
// backward-data of the 4-channel layers, 8 / 16 / 32 the same for the 1-channel layers.
// The default library holds only what is on by default (bit 2: backward-data of the 4-channel layers); the other forms --
// measured slower in the page step, DESIGN.md section 5 -- are built with UOCR_BUILD_EXPERIMENTS=1 ./build.sh.
bool uocr_conv_t32_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which) {
    if (dtype != UOCR_F32 || !ctx->opt_fast || d.n > 65535 || !same5x5(d)) return false;
    const int built = ctx->opt_t32 & UOCR_T32_BUILT;
    if (d.cin == 4 && (d.cout == 2 || d.cout == 4)) return built & (1 << which);
    if (d.cin == 1 && d.cout == 1) return built & (8 << which);
    return false;
}

int uocr_conv_fwd_t32(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                      double pad_value, int use_bias, int act, double act_alpha) {
    auto run = [&](auto geo) {
        using G = decltype(geo);
        return launch_t32<G>(ctx, x, w, b, y, nullptr, d.n, d.h, d.w, d.oh, d.ow, d.ph, d.pw, (float)pad_value, use_bias,
                             act, (float)act_alpha, UOCR_ACT_NONE, 0.f);
    };
#ifdef UOCR_EXPERIMENTS
    if (d.cin == 1) return run(Geo<1, 1, 5, 5, 1, M_FWD>{});
    if (d.cout == 2) return run(Geo<4, 2, 5, 5, 1, M_FWD>{});
    return run(Geo<4, 4, 5, 5, 1, M_FWD>{});
#else
    (void)run;
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_t32 forward: library built without UOCR_BUILD_EXPERIMENTS");
#endif
}

int uocr_conv_dgrad_t32(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d,
                        const ActMask& mask) {
    const int mact = mask.y ? mask.act : UOCR_ACT_NONE;
    auto run = [&](auto geo) {                           // a forward conv over dy with flipped taps: padding kh - 1 - ph
        using G = decltype(geo);
        return launch_t32<G>(ctx, dy, w, nullptr, dx, mask.y, d.n, d.oh, d.ow, d.h, d.w, d.kh - 1 - d.ph,
                             d.kw - 1 - d.pw, 0.f, 0, UOCR_ACT_NONE, 0.f, mact, (float)mask.alpha);
    };
#ifdef UOCR_EXPERIMENTS
    if (d.cin == 1) return run(Geo<1, 1, 5, 5, 1, M_DGRAD>{});
#endif
    if (d.cout == 2) return run(Geo<2, 4, 5, 5, 1, M_DGRAD>{});
    return run(Geo<4, 4, 5, 5, 1, M_DGRAD>{});
}

// backward-data of Upsample2D(2) + conv 5x5 / padding 2 (4 -> 4 or 1 -> 1 channels) on the low-res grid
bool uocr_upconv_t32_eligible(uocr_ctx* ctx, int dtype, int cin, int cout) {
    const int built = ctx->opt_t32 & UOCR_T32_BUILT;
    return dtype == UOCR_F32 && ctx->opt_fast &&
           ((cin == 4 && cout == 4 && (built & 4)) || (cin == 1 && cout == 1 && (built & 32)));
}

int uocr_upconv_dgrad_t32(uocr_ctx* ctx, const void* dy, const void* w, void* dx_low, int n, int hl, int wl, int ch,
                          const void* mask_y, int mask_act, double mask_alpha) {
#ifndef UOCR_EXPERIMENTS
    UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "upconv_t32: library built without UOCR_BUILD_EXPERIMENTS");
#else
    const int mact = mask_y ? mask_act : UOCR_ACT_NONE;
    if (ch == 1)
        return launch_t32<Geo<1, 1, 6, 6, 2, M_UPDGRAD, 4>>(ctx, dy, w, nullptr, dx_low, mask_y, n, 2 * hl, 2 * wl, hl, wl,
                                                             2, 2, 0.f, 0, UOCR_ACT_NONE, 0.f, mact, (float)mask_alpha);
    return launch_t32<Geo<4, 4, 6, 6, 2, M_UPDGRAD>>(ctx, dy, w, nullptr, dx_low, mask_y, n, 2 * hl, 2 * wl, hl, wl, 2, 2,
                                                     0.f, 0, UOCR_ACT_NONE, 0.f, mact, (float)mask_alpha);
#endif
}
