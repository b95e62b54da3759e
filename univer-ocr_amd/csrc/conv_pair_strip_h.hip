// The Monochrome block on column strips, binary16 storage (UOCR_F16): see conv_pair_strip.hip for the float32 kernels
// and the idea (reference: my_model/model.py:108-135, nn/layers/convolutional.py:62-145, nn/layers/layers.py:377-418).
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize
#include <algorithm>
#include <type_traits>

#include "finish_group.h"
#include "conv_pair.h"
#include "conv_pair_strip.h"
#include "uocr_common.h"

namespace {

using namespace pair_strip;

// ---------------------------------------------------------------------------------------------------------------
// binary16 storage (UOCR_F16): the same strips on v_mfma_f32_16x16x16_f16 -- 16 K values per MFMA, so a whole 3x3
// window, all 16 positions of a group or all 16 channels are ONE instruction: 5 MFMAs per group of 16 positions in
// the backward pass (float32: 18), 2 in the forward pass (7).  Operands are binary16 (x / dy / y are stored that way;
// the float32 master weights, a1 and d_a1 are rounded to binary16 as operands -- the same rounding points as the tile
// kernels of conv_pair.hip, what a layer-by-layer run in this mode stores in HBM for a1 / d_a1 anyway); every sum
// is float32.  K slot (kq, j) of a window operand = (column shift tx = kq, row j): lane (kq, n) packs ITS three
// window rows into two registers, (row t-1, row t) and (row t+1, 0); a step shifts them by one half (v_alignbit)
// and the freshly loaded row is the new second register.  The LDS rings hold the shifted copies as binary16 (the
// weight-gradient operand = 4 consecutive positions = one ds_read_b64), the transpose scratch holds d_a1 as
// [position][channel] binary16 with 48-byte rows (ds_read_b64 conflict-free).
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

constexpr int XROWH = 80;      // halves per (ring slot, group) of the x ring: copies tx = 0, 1, 2, ones, dump
constexpr int GROWH = 64;
constexpr int TSTH = 24;       // halves per position row of the transpose scratch (48 B)
constexpr int TRSZH = 16 * TSTH;

template <int G>
struct StripH {
    // ring slot strides (halves) = 24 dwords mod 64: the three tap rows of a ds_read_b64 lane group on different banks
    static constexpr int XSLOT = ((G * XROWH / 2 + 63) / 64 * 64 + 24) * 2;
    static constexpr int GSLOT = ((G * GROWH / 2 + 63) / 64 * 64 + 24) * 2;
    static constexpr int XS = 3 * XSLOT, GS = 3 * GSLOT, TR = G * TRSZH;
    static constexpr int WAVE = XS + GS + TR;        // halves of wave-private LDS (a multiple of 4)
    static constexpr int COLS = 16 * G;
};

__device__ __forceinline__ f32x4 mfma16(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x4 halves4(uint32_t lo, uint32_t hi) {
    const u32x2 v = {lo, hi};
    return __builtin_bit_cast(f16x4, v);
}
__device__ __forceinline__ f16x4 pack4h(float a, float b, float c, float d) {
    const f32x4 v = {a, b, c, d};
    return __builtin_convertvector(v, f16x4);
}
__device__ __forceinline__ uint32_t half_bits(float v) {
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v);
}

// MODE 0: no position is ever masked; MODE 2: a column block between other blocks, all of its computed columns inside
// the image: only its first and last computed column (owned by the neighbours) are taken out of the weight gradients
// (two registers per row step); MODE 1: anything else, per-position selects.  bx0: first column block of this launch.
template <int G, bool DX, bool SIG, int MODE>
__global__ __launch_bounds__(512) void pair_strip_bwd_h_kernel(const _Float16* __restrict__ x,
                                                               const _Float16* __restrict__ yout,
                                                               const _Float16* __restrict__ dy,
                                                               const float* __restrict__ w1,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ w2,
                                                               float* __restrict__ partial, _Float16* __restrict__ dx,
                                                               int h, int wd, int band_h, float pad1, int use_b1,
                                                               float alpha, int bx0, int nbx_total) {
    using L = StripH<G>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int bwc = nw * L::COLS;
    const int outw = bwc + 16;
    _Float16* const xring = reinterpret_cast<_Float16*>(lds) + wv * L::WAVE;
    _Float16* const gring = xring + L::XS;
    _Float16* const trs = gring + L::GS;
    float* const outr = lds + (nw * L::WAVE + 1) / 2;          // [NSLOT][NPLANE][outw] floats, shared by the block

    const int bx = bx0 + blockIdx.x;
    const int cstart = bx * (bwc - 2);
    const int own_lo = cstart + (bx > 0 ? 1 : 0);
    const bool last_block = cstart + bwc >= wd;
    const int own_hi = last_block ? wd : cstart + bwc - 1;
    const int wc0 = cstart + wv * L::COLS;
    const bool active = wc0 < wd;
    const int r0 = blockIdx.y * band_h, r1 = min(h, r0 + band_h);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const _Float16 *xb = x + img, *gb = dy + img, *yb = yout + img;

    // ---- constant MFMA operands (binary16): K slot (kq, j) ------------------------------------------------
    // Z^T / S^T [pos, ch]: A = window of x / g (rows t-1, t, t+1 in j = 0, 1, 2), B = W1[tap (j, kq)] / W2[tap (2-j, kq)]
    f16x4 w1b, w2b, w1u;
    {
        const int txm = n >> 2, tym = n & 3;
        const bool live = txm < 3 && tym < 3;
        const int tapu = live ? tym * 3 + txm : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool lv = kq < 3 && j < 3;
            w1b[j] = lv ? (_Float16)w1[(min(j, 2) * 3 + min(kq, 2)) * CH + n] : (_Float16)0.f;
            w2b[j] = lv ? (_Float16)w2[((2 - min(j, 2)) * 3 + min(kq, 2)) * CH + n] : (_Float16)0.f;
            w1u[j] = live ? (_Float16)w1[tapu * CH + 4 * kq + j] : (_Float16)0.f;     // U^T: A[m = n][k = ch 4kq + j]
        }
    }
    const float bias = use_b1 ? b1[n] : 0.f;

    // ---- per-lane addresses (halves) ----------------------------------------------------------------------
    const int xw_addr = (kq < 3 ? kq : 4) * 16 + n;
    const int gw_addr = kq * 16 + n;
    int xr_addr[3], gr_addr[3];
    {
        const int tapc = n <= 8 ? n : n >= 12 ? 0 : n == 9 ? 8 : 4, ty = tapc / 3, tx = tapc - 3 * ty;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            xr_addr[p] = ((p + ty) % 3) * L::XSLOT + (n == 9 ? 3 : tx) * 16 + 4 * kq;       // row t-1+ty
            gr_addr[p] = ((p + 2 - ty) % 3) * L::GSLOT + tx * 16 + 4 * kq;                  // row t+1-ty
        }
    }
    const int tw_addr = 4 * kq * TSTH + n, tr_addr = n * TSTH + 4 * kq;
    const int ow_addr = kq * outw + 1 + wv * L::COLS + n;

    // ---- global loads (one 2-byte element per lane, row descriptors as in the float32 kernel) -------------------
    const int sh = min(kq, 2) - 1;
    int xoff[G], goff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        xoff[g] = max(wc0 + 16 * g + n + sh, 0) * 2;
        goff[g] = max(wc0 + 16 * g + n - sh, 0) * 2;
    }
    const uint32_t xkeep = wc0 + n + sh >= 0 ? 0xFFFFu : 0u, gkeep = wc0 + n - sh >= 0 ? 0xFFFFu : 0u;
    const unsigned row_bytes = (unsigned)wd * 2u;
    auto row_rsrc = [&](const _Float16* base, int row) {
        const bool in = row >= 0 && row < h;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + (size_t)min(max(row, 0), h - 1) * wd), 0,
                                                 in ? row_bytes : 0u, 0x00020000);
    };
    // the two halo positions of MODE 2 (first computed column of the first wave, last of the last wave)
    const bool halo_l = MODE == 2 && wv == 0 && bx > 0 && kq == 0;
    const bool halo_r = MODE == 2 && wv == nw - 1 && !last_block && kq == 3;

    // ---- state --------------------------------------------------------------------------------------------
    uint32_t xp0[G], xp1[G], gp0[G], gp1[G];           // windows: (row t-1, row t), (row t+1, 0)
    float rr[G][3];
    f32x4 acc1[2], acc2[2];
    float db2acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        db2acc[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[g][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[j] = acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_row = [&](int row, uint32_t (&xn)[G], uint32_t (&dyn)[G], uint32_t (&yn)[G]) {
        const auto rx = row_rsrc(xb, row), rg = row_rsrc(gb, row), ry = row_rsrc(yb, row);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xn[g] = __builtin_amdgcn_raw_buffer_load_b16(rx, xoff[g], 0, 0);
            dyn[g] = __builtin_amdgcn_raw_buffer_load_b16(rg, goff[g], 0, 0);
            if constexpr (SIG) yn[g] = __builtin_amdgcn_raw_buffer_load_b16(ry, goff[g], 0, 0);
        }
    };
    // the new row (bits of one binary16 value per lane) enters the windows and the LDS rings
    auto finish_row = [&](int row, int slot, uint32_t (&xn)[G], uint32_t (&dyn)[G], uint32_t (&yn)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t gv = dyn[g];
            if constexpr (SIG) {                       // g = dy * y (1 - y) in float32, one rounding back to binary16
                const float yf = (float)__builtin_bit_cast(_Float16, (unsigned short)yn[g]);
                const float gf = (float)__builtin_bit_cast(_Float16, (unsigned short)dyn[g]);
                gv = half_bits(gf * (yf * (1.f - yf)));
            }
            uint32_t xv = xn[g];
            if (g == 0) {
                xv &= xkeep;
                gv &= gkeep;
            }
            if constexpr (MODE == 1) {
                const int cx = wc0 + 16 * g + n + sh;
                xv = (row >= 0 && row < h && cx >= 0 && cx < wd) ? xv : half_bits(pad1);
            }
            xp0[g] = __builtin_amdgcn_alignbit(xp1[g], xp0[g], 16);
            xp1[g] = xv;
            gp0[g] = __builtin_amdgcn_alignbit(gp1[g], gp0[g], 16);
            gp1[g] = gv;
            xring[slot * L::XSLOT + g * XROWH + xw_addr] = __builtin_bit_cast(_Float16, (unsigned short)xv);
            gring[slot * L::GSLOT + g * GROWH + gw_addr] = __builtin_bit_cast(_Float16, (unsigned short)gv);
        }
    };

    // ---- prologue --------------------------------------------------------------------------------------------
    for (int i = lane; i < 3 * G * 16; i += 64)
        xring[(i / (16 * G)) * L::XSLOT + ((i >> 4) % G) * XROWH + 3 * 16 + (i & 15)] = (_Float16)1.f;
    if constexpr (DX)
        for (int i = tid; i < NSLOT * NPLANE * outw; i += blockDim.x) outr[i] = 0.f;
    uint32_t xn[G], dyn[G], yn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xp0[g] = xp1[g] = gp0[g] = gp1[g] = 0u;
    if (active) {
        uint32_t xq[3][G], dq[3][G], yq[3][G];
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(r0 - 2 + j, xq[j], dq[j], yq[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) finish_row(r0 - 2 + j, j, xq[j], dq[j], yq[j]);
        load_row(r0 + 1, xn, dyn, yn);
    }
    __syncthreads();

    auto compute = [&](auto ptag, auto otag, int t, float (&cdone)[G]) {
        constexpr int P = decltype(ptag)::value;
        constexpr bool OWN = decltype(otag)::value;
        constexpr int S0 = P, S1 = (P + 1) % 3, S2 = (P + 2) % 3;
        f32x4 z[G], s[G];
        f16x4 xa[G], ga[G];
        if constexpr (OWN) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xa[g] = *reinterpret_cast<const f16x4*>(xring + xr_addr[P] + g * XROWH);
                ga[g] = *reinterpret_cast<const f16x4*>(gring + gr_addr[P] + g * GROWH);
            }
        }
        const f32x4 zinit = {bias, bias, bias, bias}, zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            z[g] = mfma16(halves4(xp0[g], xp1[g]), w1b, zinit);
            s[g] = mfma16(halves4(gp0[g], gp1[g]), w2b, zero);
        }
        mfma_round();
        // results: channel n at positions (t, wc0 + 16g + 4kq + i)
        f16x4 a4[G], d4[G], dn4[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a[4], d[4], dn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float slope = z[g][i] >= 0.f ? 1.f : alpha;
                a[i] = z[g][i] * slope;
                d[i] = dn[i] = s[g][i] * slope;
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = wc0 + 16 * g + 4 * kq + i;
                    const bool owned = c >= own_lo && c < own_hi;
                    d[i] = c < wd ? d[i] : 0.f;
                    a[i] = owned ? a[i] : 0.f;
                    dn[i] = owned ? d[i] : 0.f;
                }
            }
            if constexpr (MODE == 2) {
                if (g == 0) {
                    a[0] = halo_l ? 0.f : a[0];
                    dn[0] = halo_l ? 0.f : dn[0];
                }
                if (g == G - 1) {
                    a[3] = halo_r ? 0.f : a[3];
                    dn[3] = halo_r ? 0.f : dn[3];
                }
            }
            a4[g] = pack4h(a[0], a[1], a[2], a[3]);
            dn4[g] = pack4h(dn[0], dn[1], dn[2], dn[3]);
            d4[g] = (MODE == 1 || (MODE == 2 && (g == 0 || g == G - 1))) ? pack4h(d[0], d[1], d[2], d[3]) : dn4[g];
            if constexpr (DX) {
                _Float16* tsc = trs + g * TRSZH;
#pragma unroll
                for (int i = 0; i < 4; ++i) tsc[tw_addr + i * TSTH] = d4[g][i];
            }
        }
        if constexpr (OWN) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc2[g & 1] = mfma16(ga[g], a4[g], acc2[g & 1]);
                acc1[g & 1] = mfma16(xa[g], dn4[g], acc1[g & 1]);
                if (g & 1) mfma_round();
            }
            const f16x2 pick = {(_Float16)0.f, (_Float16)1.f};            // row t = the high half of (row t-1, row t)
#pragma unroll
            for (int g = 0; g < G; ++g) db2acc[g] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, gp0[g]), pick, db2acc[g], false);
        }
        if constexpr (DX) {
            __builtin_amdgcn_wave_barrier();
            f16x4 dt[G];
#pragma unroll
            for (int g = 0; g < G; ++g) dt[g] = *reinterpret_cast<const f16x4*>(trs + g * TRSZH + tr_addr);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 u = mfma16(w1u, dt[g], zero);
                cdone[g] = rr[g][S0] + u[0];
                rr[g][S1] += u[1];
                rr[g][S2] = u[2];
            }
        }
    };
    auto step = [&](auto ptag, int t, int oslot) {
        constexpr int P = decltype(ptag)::value;
        constexpr int S0 = P, S2 = (P + 2) % 3;
        float cdone[G];
        if (active && t >= 0 && t < h && t <= r1) {
            if (t >= r0 && t < r1) compute(ptag, std::true_type{}, t, cdone);
            else compute(ptag, std::false_type{}, t, cdone);
        } else if constexpr (DX) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cdone[g] = rr[g][S0];
                rr[g][S2] = 0.f;
            }
        }
        if constexpr (DX) {
            if (active && kq < 3) {
#pragma unroll
                for (int g = 0; g < G; ++g) outr[oslot * NPLANE * outw + ow_addr + 16 * g] = cdone[g];
            }
        }
        if (active) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                asm volatile("" : "+v"(xn[g]), "+v"(dyn[g]));
                if constexpr (SIG) asm volatile("" : "+v"(yn[g]));
            }
            finish_row(t + 2, S0, xn, dyn, yn);
            load_row(t + 3, xn, dyn, yn);
        }
    };

    const int nsteps = r1 - r0 + 2;
    for (int ss = 0, t0 = r0 - 1; ss * 3 < nsteps; ++ss, t0 += 3) {
        const int ob = (ss & 1) * 3;
        step(phase_t<0>{}, t0, ob);
        step(phase_t<1>{}, t0 + 1, ob + 1);
        step(phase_t<2>{}, t0 + 2, ob + 2);
        if constexpr (DX) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int row = t0 - 1 + j;
                if (row < r0 || row >= r1) continue;
                const float* o = outr + (ob + j) * NPLANE * outw + 1 + wv * L::COLS;
                for (int cc = lane; cc < L::COLS; cc += 64) {
                    const int c = wc0 + cc;
                    const float v = o[outw + cc] + o[cc + 1] + o[2 * outw + cc - 1];
                    if (c >= own_lo && c < own_hi) dx[img + (size_t)row * wd + c] = (_Float16)v;
                }
            }
        }
    }

    __syncthreads();
    float* red = lds;
    {
        const f32x4 s1 = acc1[0] + acc1[1], s2 = acc2[0] + acc2[1];
        float* rw = red + wv * PAIR_NPART;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rw[(4 * kq + i) * 16 + n] = s1[i];
            rw[256 + (4 * kq + i) * 16 + n] = s2[i];
        }
        float b2 = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = wc0 + 16 * g + n;
            if (kq == 1 && c >= own_lo && c < own_hi) b2 += db2acc[g];
        }
        b2 = wave_reduce_sum(b2);
        if (lane == 0) rw[512] = b2;
    }
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nbx_total + bx;
    for (int i = tid; i < PAIR_NPART; i += blockDim.x) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * PAIR_NPART + i];
        partial[blk * PAIR_NPART + i] = v;
    }
}

// Images wider than one block of waves: INDEPENDENT waves.  Every wave is its own work item (strip s of 16*G computed
// columns starting at s * (16*G - 2), band of rows): it recomputes the one column on each side that belongs to its
// neighbours (2 of 64: 3 %) instead of exchanging edge sums with them, so there is no barrier in the loop, its output
// rows ring is private, and any number of strips balances over the chip (cooperative blocks of 8 waves left a ragged
// fifth column block at 2048 columns that ran as a serial tail of 160 us).  Per wave at run time: a strip whose
// computed columns all lie inside the image masks only its two halo positions (front<2>), the last strip of a ragged
// width or a padding value takes the per-position selects (front<1>).
// d_a1 changes hands through LDS with no sub-dword stores and no wait: a lane's four positions of one channel are one
// ds_write_b64 into a [channel][position] image (40-byte rows), ds_read_b64_tr_b16 hands lane (kq, position) the
// channels 4kq..4kq+3 of its position (the hardware transpose), and that read belongs to the NEXT step: U^T of row
// t-1 is computed in step t, so the round trip through LDS is a whole step old when its data are needed.
constexpr int TWST = 20;                   // halves per channel row of the [channel][position] image (40 B)
constexpr int TWSZ = 16 * TWST;
using fp16x4_lds = __fp16 __attribute__((__vector_size__(4 * sizeof(__fp16))));
using f32x2 = __attribute__((ext_vector_type(2))) float;
// two float32 -> one register of two binary16 (round to nearest even: v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t cvt2h(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// 0xFFFF in every half whose binary16 value has its sign bit set (v_pk_ashrrev_i16 by 15; hipcc expands the vector
// shift into compares and selects per element)
__device__ __forceinline__ uint32_t neg_halves(uint32_t packed) {
    uint32_t r;
    // (op_sel_hi [0, 1]: the HIGH lane too takes the low 16 bits of the constant 15 as its shift -- an inline constant has
    // zeros in its upper half, which would leave the high half unshifted)
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(packed));
    return r;
}
__device__ __forceinline__ uint32_t mul2h(uint32_t a, f16x2 b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) * b);
}
__device__ __forceinline__ uint32_t max2h(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)));
}

template <int G, bool DX, bool SIG>
__global__ __launch_bounds__(256) void pair_wave_bwd_h_kernel(const _Float16* __restrict__ x,
                                                              const _Float16* __restrict__ yout,
                                                              const _Float16* __restrict__ dy,
                                                              const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ w2,
                                                              float* __restrict__ partial, _Float16* __restrict__ dx,
                                                              int h, int wd, int band_h, float pad1, int use_b1,
                                                              float alpha, int nstrips) {
    using L = StripH<G>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    constexpr int OUTW = L::COLS + 2;                 // private output rows: columns -1 .. COLS of the strip
    constexpr int RINGH = L::XS + L::GS + 3 * G * TWSZ;               // halves: x ring, g ring, three d_a1 images per group
    constexpr int WAVEF = (RINGH + 3) / 4 * 2 + 3 * NPLANE * OUTW;    // floats of wave-private LDS (8-byte granules)
    _Float16* const xring = reinterpret_cast<_Float16*>(lds + wv * WAVEF);
    _Float16* const gring = xring + L::XS;
    _Float16* const trs = gring + L::GS;
    float* const outr = lds + wv * WAVEF + (RINGH + 3) / 4 * 2;     // [3 rows][NPLANE][OUTW]
    const _Float16* const zeros = reinterpret_cast<const _Float16*>(lds + nw * WAVEF);   // G * XROWH halves of 0, shared

    const int sidx = blockIdx.x * nw + wv;            // this wave's strip
    const bool active = sidx < nstrips;
    const int wc0 = sidx * (L::COLS - 2);             // first computed column
    const bool last_strip = wc0 + L::COLS >= wd;
    const int own_lo = wc0 + (sidx > 0 ? 1 : 0);
    const int own_hi = last_strip ? wd : wc0 + L::COLS - 1;
    const bool ragged = wc0 + L::COLS > wd || pad1 != 0.f;     // (wave-uniform) needs the per-position selects
    const int r0 = blockIdx.y * band_h, r1 = min(h, r0 + band_h);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const _Float16 *xb = x + img, *gb = dy + img, *yb = yout + img;

    // ---- constant MFMA operands (binary16): K slot (kq, j) ------------------------------------------------
    f16x4 w1b, w2b, w1u;
    {
        const int txm = n >> 2, tym = n & 3;
        const bool live = txm < 3 && tym < 3;
        const int tapu = live ? tym * 3 + txm : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool lv = kq < 3 && j < 3;
            w1b[j] = lv ? (_Float16)w1[(min(j, 2) * 3 + min(kq, 2)) * CH + n] : (_Float16)0.f;
            w2b[j] = lv ? (_Float16)w2[((2 - min(j, 2)) * 3 + min(kq, 2)) * CH + n] : (_Float16)0.f;
            w1u[j] = live ? (_Float16)w1[tapu * CH + 4 * kq + j] : (_Float16)0.f;     // U^T: A[m = n][k = ch 4kq + j]
        }
    }
    const float bias = use_b1 ? b1[n] : 0.f;

    // ---- per-lane addresses (halves) ----------------------------------------------------------------------
    const int xw_addr = (kq < 3 ? kq : 4) * 16 + n;
    const int gw_addr = kq * 16 + n;
    int xr_addr[3], gr_addr[3];
    {
        const int tapc = n <= 8 ? n : n >= 12 ? 0 : n == 9 ? 8 : 4, ty = tapc / 3, tx = tapc - 3 * ty;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            xr_addr[p] = ((p + ty) % 3) * L::XSLOT + (n == 9 ? 3 : tx) * 16 + 4 * kq;       // row t-1+ty
            gr_addr[p] = ((p + 2 - ty) % 3) * L::GSLOT + tx * 16 + 4 * kq;                  // row t+1-ty
        }
    }
    // d_a1 image [channel][position]: lane (kq, ch n) stores positions 4kq..4kq+3; the transposed read of lane
    // 4q+p of quarter kq names row (channel) 4kq+q, positions 4p..4p+3, and receives its own position's four channels
    const int tw_addr = n * TWST + 4 * kq;
    const int tr_addr = (4 * kq + ((n >> 2) & 3)) * TWST + 4 * (n & 3);
    const int ow_addr = kq * OUTW + 1 + n;

    // ---- global memory: one 2-byte element per lane, one descriptor per page row -------------------------------
    const int sh = min(kq, 2) - 1;
    int xoff[G], goff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        xoff[g] = max(wc0 + 16 * g + n + sh, 0) * 2;
        goff[g] = max(wc0 + 16 * g + n - sh, 0) * 2;
    }
    const uint32_t xkeep = wc0 + n + sh >= 0 ? 0xFFFFu : 0u, gkeep = wc0 + n - sh >= 0 ? 0xFFFFu : 0u;
    const unsigned row_bytes = (unsigned)wd * 2u;
    auto row_rsrc = [&](const _Float16* base, int row) {
        const bool in = row >= 0 && row < h;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base + (size_t)min(max(row, 0), h - 1) * wd), 0,
                                                 in ? row_bytes : 0u, 0x00020000);
    };
    // the two halo positions (first / last computed column: owned by the neighbouring strips) as masks on the packed halves
    const uint32_t keep_l = (sidx > 0 && kq == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;
    const uint32_t keep_r = (!last_strip && kq == 3) ? 0x0000FFFFu : 0xFFFFFFFFu;
    const f16x2 alpha2 = {(_Float16)alpha, (_Float16)alpha};
    const uint32_t alpha_bits = __builtin_bit_cast(uint32_t, alpha2);
    // the dx columns this lane stores (lane = column of the strip, COLS <= 64): byte offset in the row, or past the row
    const unsigned dx_off = (lane < L::COLS && wc0 + lane >= own_lo && wc0 + lane < own_hi) ? (unsigned)(wc0 + lane) * 2u : 0x7FFFFFFFu;

    // ---- state --------------------------------------------------------------------------------------------
    u32x2 xp[G], gp[G];                                // windows: {(row t-1, row t), (row t+1, 0)}
    float rr[G][3];
    f32x4 acc1[2], acc2[2];
    float db2acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        db2acc[g] = 0.f;
        xp[g] = gp[g] = u32x2{0u, 0u};
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[g][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[j] = acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_row = [&](int row, uint32_t (&xn)[G], uint32_t (&dyn)[G], uint32_t (&yn)[G]) {
        const auto rx = row_rsrc(xb, row), rg = row_rsrc(gb, row), ry = row_rsrc(yb, row);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xn[g] = __builtin_amdgcn_raw_buffer_load_b16(rx, xoff[g], 0, 0);
            dyn[g] = __builtin_amdgcn_raw_buffer_load_b16(rg, goff[g], 0, 0);
            if constexpr (SIG) yn[g] = __builtin_amdgcn_raw_buffer_load_b16(ry, goff[g], 0, 0);
        }
    };
    auto finish_row = [&](int row, int slot, uint32_t (&xn)[G], uint32_t (&dyn)[G], uint32_t (&yn)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t gv = dyn[g];
            if constexpr (SIG) {                       // g = dy * y (1 - y) in float32, one rounding back to binary16
                const float yf = (float)__builtin_bit_cast(_Float16, (unsigned short)yn[g]);
                const float gf = (float)__builtin_bit_cast(_Float16, (unsigned short)dyn[g]);
                gv = half_bits(gf * (yf * (1.f - yf)));
            }
            uint32_t xv = xn[g];
            if (g == 0) {
                xv &= xkeep;
                gv &= gkeep;
            }
            if (ragged) {
                const int cx = wc0 + 16 * g + n + sh;
                xv = (row >= 0 && row < h && cx >= 0 && cx < wd) ? xv : half_bits(pad1);
            }
            xp[g][0] = __builtin_amdgcn_alignbit(xp[g][1], xp[g][0], 16);
            xp[g][1] = xv;
            gp[g][0] = __builtin_amdgcn_alignbit(gp[g][1], gp[g][0], 16);
            gp[g][1] = gv;
            xring[slot * L::XSLOT + g * XROWH + xw_addr] = __builtin_bit_cast(_Float16, (unsigned short)xv);
            gring[slot * L::GSLOT + g * GROWH + gw_addr] = __builtin_bit_cast(_Float16, (unsigned short)gv);
        }
    };

    // ---- prologue --------------------------------------------------------------------------------------------
    for (int i = lane; i < 3 * G * 16; i += 64)
        xring[(i / (16 * G)) * L::XSLOT + ((i >> 4) % G) * XROWH + 3 * 16 + (i & 15)] = (_Float16)1.f;
    if constexpr (DX)
        for (int i = lane; i < 3 * NPLANE * OUTW; i += 64) outr[i] = 0.f;
    for (int i = tid; i < G * XROWH / 2; i += blockDim.x) lds[nw * WAVEF + i] = 0.f;
    __syncthreads();
    uint32_t xn[G], dyn[G], yn[G];
    if (active) {
        uint32_t xq[3][G], dq[3][G], yq[3][G];
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(r0 - 2 + j, xq[j], dq[j], yq[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) finish_row(r0 - 2 + j, j, xq[j], dq[j], yq[j]);
        load_row(r0 + 1, xn, dyn, yn);
    }

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // front half of a row step: Z^T, S^T, activation, weight gradients, d_a1 into image P.  ONE variant for owned and
    // overlap rows (two variants made the compiler copy all accumulators where they meet): a row outside the band reads
    // its weight-gradient operands from a block of zeros and adds (0, 0) . (g, g) to db2.
    // LeakyReLU and its derivative on PACKED binary16 (7 instead of 10 vector instructions per pair of elements; the
    // kernel is bound by vector issue): z and s are rounded to binary16 first, the mask is the sign of z, the negative
    // branch is one binary16 product (alpha16 * z16, alpha16 * s16) -- on values 1 / alpha times smaller than the
    // positive branch, which is exact (a = z16, d = s16).
    auto front = [&](auto ptag, auto mtag, int t, bool own) {
        constexpr int P = decltype(ptag)::value;
        constexpr int MODE = decltype(mtag)::value;
        f32x4 z[G], s[G];
        f16x4 xa[G], ga[G];
        const _Float16* xsrc = own ? xring + xr_addr[P] : zeros;
        const _Float16* gsrc = own ? gring + gr_addr[P] : zeros;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xa[g] = *reinterpret_cast<const f16x4*>(xsrc + g * XROWH);
            ga[g] = *reinterpret_cast<const f16x4*>(gsrc + g * GROWH);
        }
        const f32x4 zinit = {bias, bias, bias, bias};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            z[g] = mfma16(__builtin_bit_cast(f16x4, xp[g]), w1b, zinit);
            s[g] = mfma16(__builtin_bit_cast(f16x4, gp[g]), w2b, zero);
        }
        mfma_round();
        u32x2 a4[G], dn4[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x2 d4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t zb = cvt2h(z[g][2 * r], z[g][2 * r + 1]), sb = cvt2h(s[g][2 * r], s[g][2 * r + 1]);
                // slope = z < 0 ? alpha : 1 on the packed halves (one v_bfi_b32), then two packed products: x * 1 is exact,
                // so these are the values of the two selects they replace, for one instruction less
                const uint32_t neg = neg_halves(zb);
                const f16x2 slope = __builtin_bit_cast(f16x2, (neg & alpha_bits) | (~neg & 0x3C003C00u));
                a4[g][r] = mul2h(zb, slope);
                d4[r] = mul2h(sb, slope);
            }
            dn4[g] = d4;
            if constexpr (MODE == 1) {                 // per-position selects on the packed halves
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint32_t keep_d = 0u, keep_o = 0u;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = wc0 + 16 * g + 4 * kq + 2 * r + e;
                        keep_d |= c < wd ? (0xFFFFu << (16 * e)) : 0u;
                        keep_o |= (c >= own_lo && c < own_hi) ? (0xFFFFu << (16 * e)) : 0u;
                    }
                    d4[r] &= keep_d;
                    a4[g][r] &= keep_o;
                    dn4[g][r] = d4[r] & keep_o;
                }
            } else {
                if (g == 0) {
                    a4[g][0] &= keep_l;
                    dn4[g][0] &= keep_l;
                }
                if (g == G - 1) {
                    a4[g][1] &= keep_r;
                    dn4[g][1] &= keep_r;
                }
            }
            if constexpr (DX) *reinterpret_cast<u32x2*>(trs + (P * G + g) * TWSZ + tw_addr) = d4;
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc2[g & 1] = mfma16(ga[g], __builtin_bit_cast(f16x4, a4[g]), acc2[g & 1]);
            acc1[g & 1] = mfma16(xa[g], __builtin_bit_cast(f16x4, dn4[g]), acc1[g & 1]);
            if (g & 1) mfma_round();
        }
        const f16x2 pick = {(_Float16)0.f, own ? (_Float16)1.f : (_Float16)0.f};      // row t = the high half of (row t-1, row t)
#pragma unroll
        for (int g = 0; g < G; ++g) db2acc[g] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, gp[g][0]), pick, db2acc[g], false);
    };
    auto computed_row = [&](int q) { return q >= 0 && q < h && q >= r0 - 1 && q <= r1; };
    // dx rows leave through a row descriptor: columns this lane does not own carry an offset past the row
    auto flush_row = [&](int row, int slot) {
        if (row < r0 || row >= r1) return;
        const float* o = outr + slot * NPLANE * OUTW + 1 + min(lane, L::COLS - 1);
        const float v = o[OUTW] + o[1] + o[2 * OUTW - 1];
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(dx + img + (size_t)row * wd, 0, row_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rd, dx_off, 0, 0);
    };
    auto step = [&](auto ptag, int t) {
        constexpr int P = decltype(ptag)::value;
        constexpr int S0 = P, S1 = (P + 1) % 3, S2 = (P + 2) % 3;
        // back half of the PREVIOUS row: its d_a1 image was written a step ago
        const bool prev = DX && computed_row(t - 1);
        f16x4 dt[G];
        if (prev) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                dt[g] = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (fp16x4_lds __attribute__((address_space(3)))*)(trs + (S2 * G + g) * TWSZ + tr_addr)));
        }
        if (computed_row(t)) {
            const bool own = t >= r0 && t < r1;
            if (ragged) front(ptag, std::integral_constant<int, 1>{}, t, own);
            else front(ptag, std::integral_constant<int, 2>{}, t, own);
        }
        if constexpr (DX) {
            // U^T of row t-1 completes dx row t-2 (window row j of the accumulators: t-1+j <-> slot (P+j) mod 3)
            float cdone[G];
            if (prev) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32x4 u = mfma16(w1u, dt[g], zero);
                    cdone[g] = rr[g][S2] + u[0];
                    rr[g][S0] += u[1];
                    rr[g][S1] = u[2];
                }
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    cdone[g] = rr[g][S2];
                    rr[g][S1] = 0.f;
                }
            }
            if (kq < 3) {
#pragma unroll
                for (int g = 0; g < G; ++g) outr[S2 * NPLANE * OUTW + ow_addr + 16 * g] = cdone[g];     // row t-2
            }
            flush_row(t - 3, S1);                      // written one step ago (same wave: the LDS keeps the order)
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            asm volatile("" : "+v"(xn[g]), "+v"(dyn[g]));
            if constexpr (SIG) asm volatile("" : "+v"(yn[g]));
        }
        finish_row(t + 2, S0, xn, dyn, yn);
        load_row(t + 3, xn, dyn, yn);
    };

    if (active) {
        const int nsteps = r1 - r0 + 3;                // t = r0-1 .. r1+1 (the last one only finishes row r1's U^T)
        int t0 = r0 - 1;
        for (int ss = 0; ss * 3 < nsteps; ++ss, t0 += 3) {
            step(phase_t<0>{}, t0);
            step(phase_t<1>{}, t0 + 1);
            step(phase_t<2>{}, t0 + 2);
        }
        if constexpr (DX) flush_row(t0 - 3, 1);        // the row the last step (phase 2) wrote into slot (2+2) mod 3
    }

    __syncthreads();
    float* red = lds;
    {
        const f32x4 s1 = acc1[0] + acc1[1], s2 = acc2[0] + acc2[1];
        float* rw = red + wv * PAIR_NPART;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rw[(4 * kq + i) * 16 + n] = s1[i];
            rw[256 + (4 * kq + i) * 16 + n] = s2[i];
        }
        float b2 = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int c = wc0 + 16 * g + n;
            if (kq == 1 && c >= own_lo && c < own_hi) b2 += db2acc[g];
        }
        b2 = wave_reduce_sum(b2);
        if (lane == 0) rw[512] = b2;
    }
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = tid; i < PAIR_NPART; i += blockDim.x) {
        float v = 0.f;
        for (int w = 0; w < nw; ++w) v += red[w * PAIR_NPART + i];
        partial[blk * PAIR_NPART + i] = v;
    }
}

}  // namespace

namespace {

// Forward, binary16 storage, independent waves (any width): Z[ch, pos] = W1^T . Xcol with the window registers as the B
// operand (one MFMA), LeakyReLU on packed binary16, P^T[tap, pos] = W2 . A1^T (one MFMA), the three row shifts of
// y[p] = b2 + sum_tap P[p + tap - 1, tap] in rolling registers, the three column shifts through the wave's private
// output rows.  No LDS ring for x (nothing needs 4 consecutive positions here), no barrier, ~12 vector instructions
// and 2 MFMAs per group of 16 positions (tile kernel: 2 MFMAs + a 9-tap LDS gather per output).
template <int G, int PF>
__global__ __launch_bounds__(256) void pair_wave_fwd_h_kernel(const _Float16* __restrict__ x, const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ w2,
                                                              const float* __restrict__ b2, _Float16* __restrict__ y,
                                                              int h, int wd, int band_h, float pad1, int use_b1,
                                                              int use_b2, float alpha, int act2, int nstrips) {
    constexpr int COLS = 16 * G, OUTW = COLS + 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    float* const outr = lds + wv * 3 * NPLANE * OUTW;               // [3 rows][NPLANE][OUTW], private
    const int sidx = blockIdx.x * nw + wv;
    if (sidx >= nstrips) return;                                    // (no barrier anywhere in this kernel)
    const int wc0 = sidx * (COLS - 2);
    const bool last_strip = wc0 + COLS >= wd;
    const int own_lo = wc0 + (sidx > 0 ? 1 : 0);
    const int own_hi = last_strip ? wd : wc0 + COLS - 1;
    const bool ragged = wc0 + COLS > wd || pad1 != 0.f;
    const int r0 = blockIdx.y * band_h, r1 = min(h, r0 + band_h);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const _Float16* xb = x + img;

    // Z: A[m = ch n][k = (kq, j)] = W1[tap (j, kq)][n];  P^T: A[m = n][k = ch 4kq + j] = W2[tap(m)][4kq + j], m = 4 tx + ty
    f16x4 w1a, w2a;
    f32x4 bias4;
    {
        const int txm = n >> 2, tym = n & 3;
        const bool live = txm < 3 && tym < 3;
        const int tapu = live ? tym * 3 + txm : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w1a[j] = (kq < 3 && j < 3) ? (_Float16)w1[(min(j, 2) * 3 + min(kq, 2)) * CH + n] : (_Float16)0.f;
            w2a[j] = live ? (_Float16)w2[tapu * CH + 4 * kq + j] : (_Float16)0.f;
            bias4[j] = use_b1 ? b1[4 * kq + j] : 0.f;
        }
    }
    const float bias2 = use_b2 ? b2[0] : 0.f;
    const f16x2 alpha2 = {(_Float16)alpha, (_Float16)alpha};
    const int ow_addr = kq * OUTW + 1 + n;
    const int sh = min(kq, 2) - 1;
    int xoff[G];
    uint32_t keep_a[G];                                // a1 is zero outside the image (conv_2's padding)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        xoff[g] = max(wc0 + 16 * g + n + sh, 0) * 2;
        keep_a[g] = wc0 + 16 * g + n < wd ? 0xFFFFFFFFu : 0u;
    }
    const uint32_t xkeep = wc0 + n + sh >= 0 ? 0xFFFFu : 0u;
    const unsigned row_bytes = (unsigned)wd * 2u;
    const unsigned y_off = (lane < COLS && wc0 + lane >= own_lo && wc0 + lane < own_hi) ? (unsigned)(wc0 + lane) * 2u : 0x7FFFFFFFu;

    u32x2 xp[G];
    float rr[G][3];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        xp[g] = u32x2{0u, 0u};
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[g][j] = 0.f;
    }
    auto load_row = [&](int row, uint32_t (&xn)[G]) {
        const bool in = row >= 0 && row < h;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(xb + (size_t)min(max(row, 0), h - 1) * wd), 0,
                                                          in ? row_bytes : 0u, 0x00020000);
#pragma unroll
        for (int g = 0; g < G; ++g) xn[g] = __builtin_amdgcn_raw_buffer_load_b16(rx, xoff[g], 0, 0);
    };
    auto finish_row = [&](int row, uint32_t (&xn)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint32_t xv = xn[g];
            if (g == 0) xv &= xkeep;
            if (ragged) {
                const int cx = wc0 + 16 * g + n + sh;
                xv = (row >= 0 && row < h && cx >= 0 && cx < wd) ? xv : half_bits(pad1);
            }
            xp[g][0] = __builtin_amdgcn_alignbit(xp[g][1], xp[g][0], 16);
            xp[g][1] = xv;
        }
    };
    for (int i = lane; i < 3 * NPLANE * OUTW; i += 64) outr[i] = 0.f;
    // rows are loaded THREE steps ahead of their use (one register set per phase of the unrolled loop): a step of this
    // kernel is 8 MFMAs of 16 cycles, far shorter than a load round trip
    uint32_t xq[3][G];
#pragma unroll
    for (int j = 0; j < 3; ++j) load_row(r0 - 2 + j, xq[j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) finish_row(r0 - 2 + j, xq[j]);
    if constexpr (PF == 2) {
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(r0 + 1 + j, xq[j]);
    } else {
        load_row(r0 + 1, xq[0]);
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // y[row][c] = act2(b2 + plane1[c] + plane0[c-1] + plane2[c+1]) of the wave's own output rows
    auto flush_row = [&](int row, int slot) {
        if (row < r0 || row >= r1) return;
        const float* o = outr + slot * NPLANE * OUTW + 1 + min(lane, COLS - 1);
        float v = bias2 + (o[OUTW] + o[-1] + o[2 * OUTW + 1]);
        if (act2 == UOCR_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));         // (binary16 result)
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(y + img + (size_t)row * wd, 0, row_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rd, y_off, 0, 0);
    };
    auto step = [&](auto ptag, int t) {
        constexpr int P = decltype(ptag)::value;
        constexpr int S0 = P, S1 = (P + 1) % 3, S2 = (P + 2) % 3;
        float cdone[G];
        if (t >= 0 && t < h && t <= r1) {
            f32x4 z[G], u[G];
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = mfma16(w1a, __builtin_bit_cast(f16x4, xp[g]), bias4);
            mfma_round();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                u32x2 a4;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t zb = cvt2h(z[g][2 * r], z[g][2 * r + 1]);
                    a4[r] = max2h(zb, mul2h(zb, alpha2));            // LeakyReLU for 0 <= alpha <= 1: v_pk_mul + v_pk_max
                    if (ragged) a4[r] &= keep_a[g];
                }
                u[g] = mfma16(w2a, __builtin_bit_cast(f16x4, a4), zero);
            }
            mfma_round();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cdone[g] = rr[g][S0] + u[g][2];          // y row t-1 is complete (tap row 2 of a1 row t)
                rr[g][S1] += u[g][1];
                rr[g][S2] = u[g][0];
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cdone[g] = rr[g][S0];
                rr[g][S2] = 0.f;
            }
        }
        if (kq < 3) {
#pragma unroll
            for (int g = 0; g < G; ++g) outr[S0 * NPLANE * OUTW + ow_addr + 16 * g] = cdone[g];     // row t-1
        }
        flush_row(t - 2, S2);
        if constexpr (PF == 2) {
            finish_row(t + 2, xq[P]);
            load_row(t + 5, xq[P]);
        } else {
            if constexpr (PF == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) asm volatile("" : "+v"(xq[0][g]));
            }
            finish_row(t + 2, xq[0]);
            load_row(t + 3, xq[0]);
        }
    };
    const int nsteps = r1 - r0 + 2;                    // t = r0-1 .. r1
    int t0 = r0 - 1;
    for (int ss = 0; ss * 3 < nsteps; ++ss, t0 += 3) {
        step(phase_t<0>{}, t0);
        step(phase_t<1>{}, t0 + 1);
        step(phase_t<2>{}, t0 + 2);
    }
    flush_row(t0 - 2, 2);
}

}  // namespace

// binary16 forward of the pair block: independent-wave strips for every width
int uocr_pair_strip_fwd_f16(uocr_ctx* ctx, const void* x, const float* w1, const float* b1, const float* w2,
                            const float* b2, void* y, int n, int h, int w, float pad1, int use_b1, int use_b2,
                            float alpha, int act2) {
    constexpr int G = 4, COLS = 16 * G;
    const int nstrips = w <= COLS ? 1 : 1 + (w - COLS + (COLS - 2) - 1) / (COLS - 2);
    const int nw = std::min(4, nstrips);
    const int blocks_x = (nstrips + nw - 1) / nw;
    int bands = 1;
    {
        const long resident = 4L * ctx->cu_count;
        double best = 1e30;
        for (int b = 1; b <= std::max(1, h / 8); ++b) {
            const int bh = (h + b - 1) / b;
            const long blocks = (long)blocks_x * ((h + bh - 1) / bh) * n;
            const double cost = (double)((blocks + resident - 1) / resident) * (bh + 2);
            if (cost < best * 0.999) { best = cost; bands = (h + bh - 1) / bh; }
        }
        if (ctx->opt_pair_band > 0) bands = (h + ctx->opt_pair_band - 1) / ctx->opt_pair_band;
    }
    const int band_h = (h + bands - 1) / bands;
    bands = (h + band_h - 1) / band_h;
    UOCR_REQUIRE(ctx, bands <= 65535 && n <= 65535);
    const size_t lds = sizeof(float) * nw * 3 * NPLANE * (COLS + 2);
    auto go = [&](auto pf) {
        hipLaunchKernelGGL((pair_wave_fwd_h_kernel<G, decltype(pf)::value>), dim3(blocks_x, bands, n), dim3(nw * 64), lds,
                           ctx->stream, (const _Float16*)x, w1, b1, w2, b2, (_Float16*)y, h, w, band_h, pad1, use_b1,
                           use_b2, alpha, act2, nstrips);
    };
    if (ctx->opt_pair_pf == 0) go(pair_strip::phase_t<0>{});
    else if (ctx->opt_pair_pf == 1) go(pair_strip::phase_t<1>{});
    else go(pair_strip::phase_t<2>{});
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

namespace {

// independent-wave launch (images wider than one cooperative block): strips of COLS computed columns every COLS - 2,
// 4 strips per block, bands so that the blocks make whole rounds of what is resident at once -- a last round that is a
// tenth full costs a full one.  G = 4: 187 registers, two blocks per CU; G = 2: four.
template <int G>
int wave_bwd_launch(uocr_ctx* ctx, const void* x, const void* y, const void* dy, const float* w1, const float* b1,
                    const float* w2, float* dw1, float* db1, float* dw2, float* db2, void* dx, int n, int h, int w,
                    float pad1, int use_b1, int use_b2, float alpha, bool sig, int accumulate, float unscale) {
    using L = StripH<G>;
    const int nstrips = w <= L::COLS ? 1 : 1 + (w - L::COLS + (L::COLS - 2) - 1) / (L::COLS - 2);
    const int nw = std::min(4, nstrips);
    const int blocks_x = (nstrips + nw - 1) / nw;
    int bands = 1;
    {
        const long resident = (G == 4 ? 2L : 4L) * ctx->cu_count;
        double best = 1e30;
        for (int b = 1; b <= std::max(1, h / 8); ++b) {
            const int bh = (h + b - 1) / b;
            const long blocks = (long)blocks_x * ((h + bh - 1) / bh) * n;
            const double cost = (double)((blocks + resident - 1) / resident) * (bh + 2);
            if (cost < best * 0.999) { best = cost; bands = (h + bh - 1) / bh; }
        }
        if (ctx->opt_pair_band > 0) bands = (h + ctx->opt_pair_band - 1) / ctx->opt_pair_band;
    }
    const int band_h = (h + bands - 1) / bands;
    bands = (h + band_h - 1) / band_h;
    const size_t lds = std::max(sizeof(float) * (nw * ((L::XS + L::GS + 3 * G * TWSZ + 3) / 4 * 2 + 3 * NPLANE * (L::COLS + 2)) + G * XROWH / 2),
                                sizeof(float) * nw * PAIR_NPART);
    const size_t nblocks = (size_t)blocks_x * bands * n;
    UOCR_REQUIRE(ctx, bands <= 65535 && n <= 65535);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, nblocks * PAIR_NPART * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    auto run = [&](auto dxtag, auto sigtag) -> int {
        constexpr bool D = decltype(dxtag)::value, S = decltype(sigtag)::value;
        static bool attr_set = false;
        if (!attr_set) {
            UOCR_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(pair_wave_bwd_h_kernel<G, D, S>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL((pair_wave_bwd_h_kernel<G, D, S>), dim3(blocks_x, bands, n), dim3(nw * 64), lds, ctx->stream,
                           (const _Float16*)x, (const _Float16*)y, (const _Float16*)dy, w1, b1, w2, partial, (_Float16*)dx,
                           h, w, band_h, pad1, use_b1, alpha, nstrips);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    };
    if (dx) rc = sig ? run(std::true_type{}, std::true_type{}) : run(std::true_type{}, std::false_type{});
    else rc = sig ? run(std::false_type{}, std::true_type{}) : run(std::false_type{}, std::false_type{});
    if (rc != UOCR_OK) return rc;
    return uocr_pair_strip_finish(ctx, partial, dw1, db1, dw2, db2, (int)nblocks, use_b1, use_b2, accumulate, unscale);
}

}  // namespace

// binary16 backward of the pair block on the strip kernels (float32 parameters and gradients; unscale = 2^-k of
// UOCR_F16_SCALED(k)): one block of cooperating waves where it spans the image (no halo columns at all), independent
// waves for wider pages
int uocr_pair_strip_bwd_f16(uocr_ctx* ctx, const void* x, const void* y, const void* dy, const float* w1,
                            const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                            void* dx, int n, int h, int w, float pad1, int use_b1, int use_b2, float alpha,
                            bool sig, int accumulate, float unscale) {
    constexpr int G = 4;
    using L = StripH<G>;
    const int nw = std::min(8, (w + L::COLS - 1) / L::COLS);
    const int bwc = nw * L::COLS;
    if (w > bwc) {
        if (ctx->opt_pair_g == 2)
            return wave_bwd_launch<2>(ctx, x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, dx, n, h, w, pad1, use_b1, use_b2, alpha,
                                      sig, accumulate, unscale);
        return wave_bwd_launch<4>(ctx, x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, dx, n, h, w, pad1, use_b1, use_b2, alpha, sig,
                                  accumulate, unscale);
    }
    int bands = std::max(1, (ctx->cu_count + n - 1) / n);
    if (ctx->opt_pair_band > 0) bands = (h + ctx->opt_pair_band - 1) / ctx->opt_pair_band;
    int band_h = std::min(h, std::max(4, (h + bands - 1) / bands));
    bands = (h + band_h - 1) / band_h;
    const size_t nblocks = (size_t)bands * n;
    UOCR_REQUIRE(ctx, bands <= 65535 && n <= 65535);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, nblocks * PAIR_NPART * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    const size_t ring = dx ? sizeof(float) * NSLOT * NPLANE * (bwc + 16) : 0;
    const size_t lds = std::max(sizeof(float) * ((nw * L::WAVE + 1) / 2) + ring, sizeof(float) * nw * PAIR_NPART);
    auto launch = [&](auto kernel) -> int {
        static bool attr_set = false;
        if (!attr_set) {
            UOCR_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kernel, dim3(1, bands, n), dim3(nw * 64), lds, ctx->stream, (const _Float16*)x,
                           (const _Float16*)y, (const _Float16*)dy, w1, b1, w2, partial, (_Float16*)dx, h, w, band_h, pad1,
                           use_b1, alpha, 0, 1);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    };
    const bool plain = w == bwc && pad1 == 0.f;
    auto run = [&](auto dxtag, auto sigtag) -> int {
        constexpr bool D = decltype(dxtag)::value, S = decltype(sigtag)::value;
        return plain ? launch(pair_strip_bwd_h_kernel<G, D, S, 0>) : launch(pair_strip_bwd_h_kernel<G, D, S, 1>);
    };
    if (dx) rc = sig ? run(std::true_type{}, std::true_type{}) : run(std::true_type{}, std::false_type{});
    else rc = sig ? run(std::false_type{}, std::true_type{}) : run(std::false_type{}, std::false_type{});
    if (rc != UOCR_OK) return rc;
    return uocr_pair_strip_finish(ctx, partial, dw1, db1, dw2, db2, (int)nblocks, use_b1, use_b2, accumulate, unscale);
}
