// The Monochrome block, float32, as COLUMN-STRIP kernels (round 3; replaces the 16 x 32 tile kernels of
// conv_pair.hip for float32):
//     x (1 ch) -> conv3x3 (1 -> 16, pad 1) -> LeakyReLU -> conv3x3 (16 -> 1, pad 1) -> [Sigmoid] -> y
// (reference: my_model/model.py:108-135 built from nn/layers/convolutional.py:62-145 and
// nn/layers/layers.py:377-418).
//
// A wave owns a strip of 16*G columns and walks DOWN the rows of its band, one row (G groups of 16 positions) per
// step; the lane index of every MFMA is the COLUMN, so
//   * the 3x3 window operands are registers: lane (kq, n) of group g holds x[row][col0 + n + kq - 1] (the three
//     column shifts of a row live in the three lane quarters), rows t-1, t, t+1 are three registers of a rolling
//     window -- one global load per row and group instead of an LDS tile with halo;
//   * the same registers written to a wave-private LDS ring ARE the three shifted copies of the row, so the
//     operand of the weight-gradient MFMAs (4 consecutive positions seen from tap (ty, tx)) is ONE aligned
//     ds_read_b128 (copy tx, row t+ty-1); a copy of ones gives db1 as row 9 of the same MFMA result;
//   * backward-data: U^T[tap, pos] = W1 . D^T with the column on the lane, tap (ty, tx) in result register ty of
//     lane quarter tx: the three row shifts of dx[p] = sum_tap U[p - tap + 1, tap] are THREE ROLLING REGISTERS
//     (row t-1 completes at step t), the three column shifts one LDS row per tx read at c-1, c, c+1.  The old
//     kernels scattered the 9 partial sums of every position through LDS (41 of 127 us) and needed a second kernel
//     for the tile borders.
// Waves of a block sit side by side (block = up to 8 waves = 512 columns) and only share the output rows ring
// (the column c+-1 of a wave's edge belongs to its neighbour): one __syncthreads per three rows.  Bands overlap
// by one row of d_a1 on each side (recomputed, 10 of 18 MFMAs), column blocks by one column on each side.
// Per group of 16 positions: 18 MFMAs (Z^T 3, S^T 3, dW1^T 4, dW2^T 4, U^T 4), ~27 vector and ~11 LDS
// instructions (tile kernels: 58 / 19 + the 9-tap gather).  Deterministic: no atomics, fixed summation order.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize
#include <algorithm>
#include <type_traits>

#include "conv_pair.h"
#include "conv_pair_strip.h"
#include "finish_group.h"
#include "uocr_common.h"

namespace {

using namespace pair_strip;

// MODE 0: one column block whose computed columns are exactly the image's (w = 64 * waves) and a zero padding
// value: no position is ever masked.  MODE 1: anything else (halo columns between column blocks, computed columns
// beyond the image, padding value) with per-position selects.
template <int G, bool DX, bool SIG, int MODE>
__global__ __launch_bounds__(G <= 2 ? 1024 : 512) void pair_strip_bwd_kernel(const float* __restrict__ x, const float* __restrict__ yout,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ w1,
                                                             const float* __restrict__ b1,
                                                             const float* __restrict__ w2,
                                                             float* __restrict__ partial, float* __restrict__ dx,
                                                             int h, int wd, int band_h, float pad1, int use_b1,
                                                             float alpha) {
    using L = Strip<G>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int bwc = nw * L::COLS;                     // computed columns of the block
    const int outw = bwc + 16;                        // row stride of an output plane (= 16 mod 32 for G = 4 / 2)
    float* const xring = lds + wv * L::WAVE;
    float* const gring = xring + L::XS;
    float* const trs = gring + L::GS;
    float* const outr = lds + nw * L::WAVE;           // [NSLOT][NPLANE][outw], shared by the block

    // columns: block bx computes [cstart, cstart + bwc) and owns [own_lo, own_hi) of them
    const int cstart = blockIdx.x * (bwc - 2);
    const int own_lo = cstart + (blockIdx.x > 0 ? 1 : 0);
    const bool last_block = cstart + bwc >= wd;
    const int own_hi = last_block ? wd : cstart + bwc - 1;
    const int wc0 = cstart + wv * L::COLS;            // first computed column of this wave
    const bool active = wc0 < wd;                     // (wave-uniform) some computed column lies inside the image
    const int r0 = blockIdx.y * band_h, r1 = min(h, r0 + band_h);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const float *xb = x + img, *gb = dy + img, *yb = yout + img;

    // ---- constant MFMA operands ------------------------------------------------------------------------------
    // Z^T / S^T [pos, ch]: A = x / g of row (t-1+kc) / (t+1-kc) shifted by tx = kq, B = W1 / W2 [tap (kc, kq)][ch n]
    float w1b[3], w2b[3], w1u[4];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        const int tap = kc * 3 + min(kq, 2);
        w1b[kc] = kq < 3 ? w1[tap * CH + n] : 0.f;
        w2b[kc] = kq < 3 ? w2[tap * CH + n] : 0.f;
    }
    // U^T [m, pos]: result row m = 4 tx + ty (register ty of lane quarter tx); A[m = n][k = ch 4kq + kc]
    {
        const int txm = n >> 2, tym = n & 3;
        const bool live = txm < 3 && tym < 3;
        const int tap = live ? tym * 3 + txm : 0;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) w1u[kc] = live ? w1[tap * CH + 4 * kq + kc] : 0.f;
    }
    const float bias = use_b1 ? b1[n] : 0.f;

    // ---- per-lane addresses (floats) --------------------------------------------------------------------------
    const int xw_addr = (kq < 3 ? kq : 4) * 16 + n;   // ring write: copy kq (lane quarter 3 -> dump copy 4)
    const int gw_addr = kq * 16 + n;                  // (copy 3 of the g ring is its dump)
    // dW operand rows: lane n = tap row of A (n = 9: the ones copy -> db1; n > 9: unused result rows, tap 8)
    int xr_addr[3], gr_addr[3];
    {
        // (unused rows 10..15 read what a used lane of their ds_read_b128 lane group reads: a broadcast, no conflict)
        const int tapc = n <= 8 ? n : n >= 12 ? 0 : n == 9 ? 8 : 4, ty = tapc / 3, tx = tapc - 3 * ty;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            xr_addr[p] = ((p + ty) % 3) * L::XSLOT + (n == 9 ? 3 : tx) * 16 + 4 * kq;       // row t-1+ty
            gr_addr[p] = ((p + 2 - ty) % 3) * L::GSLOT + tx * 16 + 4 * kq;                  // row t+1-ty
        }
    }
    // transpose scratch: lane (kq, ch n) writes position 4kq + i (slot f(4kq) + 2i), lane (kq, pos n) reads quad kq
    const int tw_addr = (n >> 2) * TPL + 4 * tslot(4 * kq) + (n & 3), tr_addr = kq * TPL + 4 * tslot(n);
    const int ow_addr = kq * outw + 1 + wv * L::COLS + n;

    // ---- global loads: one buffer descriptor per page row (base = the row, size = the row or nothing), so that the
    // column one past the image and whole rows outside it read as 0 without a vector instruction.
    // lane (kq, n) of group g reads x at column wc0 + 16g + n + kq - 1 and g at wc0 + 16g + n - kq + 1;
    // the one column LEFT of the image (wave 0, group 0) is read at column 0 and zeroed by a per-lane factor
    const int sh = min(kq, 2) - 1;
    int xoff[G], goff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        xoff[g] = max(wc0 + 16 * g + n + sh, 0) * 4;
        goff[g] = max(wc0 + 16 * g + n - sh, 0) * 4;
    }
    const float xkeep = wc0 + n + sh >= 0 ? 1.f : 0.f, gkeep = wc0 + n - sh >= 0 ? 1.f : 0.f;
    const unsigned row_bytes = (unsigned)wd * 4u;
    auto row_rsrc = [&](const float* base, int row) {
        const bool in = row >= 0 && row < h;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)min(max(row, 0), h - 1) * wd), 0,
                                                 in ? row_bytes : 0u, 0x00020000);
    };

    // ---- state ------------------------------------------------------------------------------------------------
    float xw[G][3], gw[G][3];                          // rolling row windows, index = ring slot
    float rr[G][3];                                    // dx row accumulators of lane (tx = kq, column n)
    f32x4 acc1[2], acc2[2];                            // dW1^T / dW2^T [tap 4kq + i][ch n], two chains each
    float db2acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        db2acc[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[g][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[j] = acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one row of the page: x and g = dy * act2'(y) of row `row` in the lane layout above
    auto load_row = [&](int row, float (&xn)[G], float (&dyn)[G], float (&yn)[G]) {
        const auto rx = row_rsrc(xb, row), rg = row_rsrc(gb, row), ry = row_rsrc(yb, row);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xn[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[g], 0, 0));
            dyn[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, goff[g], 0, 0));
            if constexpr (SIG) yn[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, goff[g], 0, 0));
        }
    };
    auto finish_row = [&](int row, int slot, float (&xn)[G], float (&dyn)[G], float (&yn)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float gv = dyn[g];
            if constexpr (SIG) gv *= yn[g] * (1.f - yn[g]);
            float xv = xn[g];
            if (g == 0) {
                xv *= xkeep;
                gv *= gkeep;
            }
            if constexpr (MODE == 1) {                 // padding value of the first conv
                const int cx = wc0 + 16 * g + n + sh;
                xv = (row >= 0 && row < h && cx >= 0 && cx < wd) ? xv : pad1;
            }
            xw[g][slot] = xv;
            gw[g][slot] = gv;
            xring[slot * L::XSLOT + g * XROW + xw_addr] = xv;
            gring[slot * L::GSLOT + g * GROW + gw_addr] = gv;
        }
    };

    // ---- prologue: ones copy, output ring, rows r0-2 .. r0 --------------------------------------------------
    for (int i = lane; i < 3 * G * 16; i += 64) xring[(i / (16 * G)) * L::XSLOT + ((i >> 4) % G) * XROW + 3 * 16 + (i & 15)] = 1.f;
    if constexpr (DX)
        for (int i = tid; i < NSLOT * NPLANE * outw; i += blockDim.x) outr[i] = 0.f;
    float xn[G], dyn[G], yn[G];                        // the row in flight: issued one whole step before it is used
    if (active) {
        float xp[3][G], dyp[3][G], yp[3][G];
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(r0 - 2 + j, xp[j], dyp[j], yp[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) finish_row(r0 - 2 + j, j, xp[j], dyp[j], yp[j]);
        load_row(r0 + 1, xn, dyn, yn);
    }
    __syncthreads();

    // ---- one row step: phase P = (t - (r0 - 1)) mod 3; row t-1+j lives in window / ring slot (P + j) mod 3.
    // The G groups of the row go through every stage together (one basic block, G independent MFMA chains per stage).
    // OWN: row t belongs to the band (weight gradients); otherwise it only feeds dx of the neighbouring rows.
    auto compute = [&](auto ptag, auto otag, int t, float (&cdone)[G]) {
        constexpr int P = decltype(ptag)::value;
        constexpr bool OWN = decltype(otag)::value;
        constexpr int S0 = P, S1 = (P + 1) % 3, S2 = (P + 2) % 3;
        f32x4 z[G], s[G], xa[G], ga[G];
        if constexpr (OWN) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                xa[g] = *reinterpret_cast<const f32x4*>(xring + xr_addr[P] + g * XROW);
                ga[g] = *reinterpret_cast<const f32x4*>(gring + gr_addr[P] + g * GROW);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            z[g] = f32x4{bias, bias, bias, bias};
            s[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int g = 0; g < G; ++g) { z[g] = mfma4(xw[g][S0], w1b[0], z[g]); s[g] = mfma4(gw[g][S2], w2b[0], s[g]); }
        mfma_round();
#pragma unroll
        for (int g = 0; g < G; ++g) { z[g] = mfma4(xw[g][S1], w1b[1], z[g]); s[g] = mfma4(gw[g][S1], w2b[1], s[g]); }
        mfma_round();
#pragma unroll
        for (int g = 0; g < G; ++g) { z[g] = mfma4(xw[g][S2], w1b[2], z[g]); s[g] = mfma4(gw[g][S0], w2b[2], s[g]); }
        mfma_round();
        // results: channel n at positions (t, wc0 + 16g + 4kq + i)
        float a[G][4], d[G][4], dn[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float slope = z[g][i] >= 0.f ? 1.f : alpha;
                a[g][i] = z[g][i] * slope;
                d[g][i] = dn[g][i] = s[g][i] * slope;
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = wc0 + 16 * g + 4 * kq + i;
                    const bool owned = c >= own_lo && c < own_hi;
                    d[g][i] = c < wd ? d[g][i] : 0.f;
                    a[g][i] = owned ? a[g][i] : 0.f;
                    dn[g][i] = owned ? d[g][i] : 0.f;
                }
            }
            if constexpr (DX) {
                float* tsc = trs + g * TRSZ;
#pragma unroll
                for (int i = 0; i < 4; ++i) tsc[tw_addr + 8 * i] = d[g][i];
            }
        }
        if constexpr (OWN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    acc2[g & 1] = mfma4(ga[g][i], a[g][i], acc2[g & 1]);
                    acc1[g & 1] = mfma4(xa[g][i], dn[g][i], acc1[g & 1]);
                    if (g & 1) mfma_round();
                }
#pragma unroll
            for (int g = 0; g < G; ++g) db2acc[g] += gw[g][S1];
        }
        if constexpr (DX) {
            __builtin_amdgcn_wave_barrier();             // same wave: the LDS executes its instructions in order
            f32x4 dt[G], u[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                dt[g] = *reinterpret_cast<const f32x4*>(trs + g * TRSZ + tr_addr);
                u[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
                for (int g = 0; g < G; ++g) u[g] = mfma4(w1u[kc], dt[g][kc], u[g]);
                mfma_round();
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                cdone[g] = rr[g][S0] + u[g][0];          // row t-1 is complete
                rr[g][S1] += u[g][1];
                rr[g][S2] = u[g][2];
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
            // row t+2 was requested a whole step ago; keep its first use down here (the scheduler would hoist the
            // arithmetic on it into the MFMA block above and wait for memory there), then request row t+3
#pragma unroll
            for (int g = 0; g < G; ++g) {
                asm volatile("" : "+v"(xn[g]), "+v"(dyn[g]));
                if constexpr (SIG) asm volatile("" : "+v"(yn[g]));
            }
            finish_row(t + 2, S0, xn, dyn, yn);
            load_row(t + 3, xn, dyn, yn);
        }
    };

    const int nsteps = r1 - r0 + 2;                    // t = r0-1 .. r1
    for (int ss = 0, t0 = r0 - 1; ss * 3 < nsteps; ++ss, t0 += 3) {
        const int ob = (ss & 1) * 3;
        step(phase_t<0>{}, t0, ob);
        step(phase_t<1>{}, t0 + 1, ob + 1);
        step(phase_t<2>{}, t0 + 2, ob + 2);
        if constexpr (DX) {
            __syncthreads();
            // rows t0-1 .. t0+1 are complete in the ring: dx[row][c] = plane1[c] + plane0[c+1] + plane2[c-1]
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int row = t0 - 1 + j;
                if (row < r0 || row >= r1) continue;
                const float* o = outr + (ob + j) * NPLANE * outw + 1 + wv * L::COLS;
                for (int cc = lane; cc < L::COLS; cc += 64) {
                    const int c = wc0 + cc;
                    const float v = o[outw + cc] + o[cc + 1] + o[2 * outw + cc - 1];
                    if (c >= own_lo && c < own_hi) dx[img + (size_t)row * wd + c] = v;
                }
            }
        }
    }

    // ---- block reduction of the weight gradients -> partial[block][PAIR_NPART] ----------------------------------
    __syncthreads();
    float* red = lds;                                  // [nw][PAIR_NPART]
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

// ---------------------------------------------------------------------------------------------------------------
// Forward on the same strips.  Here the first product is taken the OTHER way round, Z[ch, pos] = W1^T . Xcol (the
// window registers are its B operand), so that its result -- channels 4kq..4kq+3 of column n in lane (kq, n) -- is,
// after LeakyReLU, directly the B operand of P^T[tap, pos] = W2 . A1^T: no transpose, no LDS but the output ring.
// y[p] = b2 + sum_tap P[p + tap - 1, tap]: row q feeds rows q+1, q, q-1 (ty = 0, 1, 2: three rolling registers),
// column c of lane quarter tx feeds column c - tx + 1 (one LDS row per tx).  7 MFMAs per group of 16 positions
// (tile kernel: 7 + a 9-tap LDS gather per output), 3 window registers per group instead of an LDS tile with halo.
// Band rows overlap by one row of a1 on each side.  MODE as in the backward kernel.
// PF: how the rows are prefetched.  0: one step ahead, rows pinned inside the loop; 1: one step ahead, only the rows
// loaded in front of the loop pinned; 2: three steps ahead (one register set per phase).
template <int G, int MODE, int PF>
__global__ __launch_bounds__(512) void pair_strip_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                             const float* __restrict__ b1,
                                                             const float* __restrict__ w2,
                                                             const float* __restrict__ b2, float* __restrict__ y,
                                                             int h, int wd, int band_h, float pad1, int use_b1,
                                                             int use_b2, float alpha, int act2) {
    using L = Strip<G>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int bwc = nw * L::COLS;
    const int outw = bwc + 16;
    float* const outr = lds;                           // [NSLOT][NPLANE][outw]
    const int cstart = blockIdx.x * (bwc - 2);
    const int own_lo = cstart + (blockIdx.x > 0 ? 1 : 0);
    const bool last_block = cstart + bwc >= wd;
    const int own_hi = last_block ? wd : cstart + bwc - 1;
    const int wc0 = cstart + wv * L::COLS;
    const bool active = wc0 < wd;
    const int r0 = blockIdx.y * band_h, r1 = min(h, r0 + band_h);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const float* xb = x + img;

    // Z[ch, pos]: A = W1^T[ch n][tap (kc, kq)], B = x of row t-1+kc shifted by tx = kq
    // P^T[m, pos]: m = 4 tx + ty; A[m = n][k = ch 4kq + i] = W2[tap(m)][4kq + i], B = a1 (this lane's four Z results)
    float w1a[3], w2a[4], bias4[4];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) w1a[kc] = kq < 3 ? w1[(kc * 3 + min(kq, 2)) * CH + n] : 0.f;
    {
        const int txm = n >> 2, tym = n & 3;
        const bool live = txm < 3 && tym < 3;
        const int tap = live ? tym * 3 + txm : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w2a[i] = live ? w2[tap * CH + 4 * kq + i] : 0.f;
            bias4[i] = use_b1 ? b1[4 * kq + i] : 0.f;
        }
    }
    const float bias2 = use_b2 ? b2[0] : 0.f;
    const int ow_addr = kq * outw + 1 + wv * L::COLS + n;

    const int sh = min(kq, 2) - 1;
    int xoff[G];
#pragma unroll
    for (int g = 0; g < G; ++g) xoff[g] = max(wc0 + 16 * g + n + sh, 0) * 4;
    const float xkeep = wc0 + n + sh >= 0 ? 1.f : 0.f;
    const unsigned row_bytes = (unsigned)wd * 4u;
    auto load_row = [&](int row, float (&xn)[G]) {
        const bool in = row >= 0 && row < h;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb + (size_t)min(max(row, 0), h - 1) * wd), 0,
                                                          in ? row_bytes : 0u, 0x00020000);
#pragma unroll
        for (int g = 0; g < G; ++g)
            xn[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff[g], 0, 0));
    };
    float xw[G][3], rr[G][3];
    auto finish_row = [&](int row, int slot, float (&xn)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float xv = xn[g];
            if (g == 0) xv *= xkeep;
            if constexpr (MODE == 1) {
                const int cx = wc0 + 16 * g + n + sh;
                xv = (row >= 0 && row < h && cx >= 0 && cx < wd) ? xv : pad1;
            }
            xw[g][slot] = xv;
        }
    };
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[g][j] = 0.f;
    for (int i = tid; i < NSLOT * NPLANE * outw; i += blockDim.x) outr[i] = 0.f;
    float xq[3][G];                                    // rows in flight, loaded three steps ahead (one set per phase)
    if (active) {
        float xp[3][G];
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(r0 - 2 + j, xp[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) finish_row(r0 - 2 + j, j, xp[j]);
        // The window registers ARE the load destinations (no instruction in between): make these loads complete here.
        // Left pending into the loop, the wait-count pass has to assume them in flight on the paths that skip a step's
        // tail and puts an s_waitcnt vmcnt(0) -- i.e. a wait for the NEXT row's loads, issued a moment ago -- in
        // front of the MFMAs of every step.  (Pinning the rows inside the loop instead does the same harm: the
        // loads then keep fixed registers that the skip paths see as in flight.  Rows are loaded three steps ahead.)
        if constexpr (PF >= 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int g = 0; g < G; ++g) asm volatile("" : "+v"(xw[g][j]));
        }
        if constexpr (PF == 2) {
#pragma unroll
            for (int j = 0; j < 3; ++j) load_row(r0 + 1 + j, xq[j]);
        } else {
            load_row(r0 + 1, xq[0]);
        }
    }
    __syncthreads();

    auto step = [&](auto ptag, int t, int oslot) {
        constexpr int P = decltype(ptag)::value;
        constexpr int S0 = P, S1 = (P + 1) % 3, S2 = (P + 2) % 3;
        float cdone[G];
        if (active && t >= 0 && t < h && t <= r1) {
            f32x4 z[G], u[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                z[g] = f32x4{bias4[0], bias4[1], bias4[2], bias4[3]};
                u[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = mfma4(w1a[0], xw[g][S0], z[g]);
            mfma_round();
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = mfma4(w1a[1], xw[g][S1], z[g]);
            mfma_round();
#pragma unroll
            for (int g = 0; g < G; ++g) z[g] = mfma4(w1a[2], xw[g][S2], z[g]);
            mfma_round();
            // results: channels 4kq + i of position (t, wc0 + 16g + n)
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = fmaxf(z[g][i], alpha * z[g][i]);     // LeakyReLU for 0 <= alpha <= 1 (checked by the entry point)
                    if constexpr (MODE == 1) a = wc0 + 16 * g + n < wd ? a : 0.f;     // conv_2 pads a1 with zeros
                    z[g][i] = a;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int g = 0; g < G; ++g) u[g] = mfma4(w2a[i], z[g][i], u[g]);
                mfma_round();
            }
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
        if (active && kq < 3) {
#pragma unroll
            for (int g = 0; g < G; ++g) outr[oslot * NPLANE * outw + ow_addr + 16 * g] = cdone[g];
        }
        if (active) {
            if constexpr (PF == 2) {
                finish_row(t + 2, S0, xq[P]);
                load_row(t + 5, xq[P]);
            } else {
                if constexpr (PF == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) asm volatile("" : "+v"(xq[0][g]));
                }
                finish_row(t + 2, S0, xq[0]);
                load_row(t + 3, xq[0]);
            }
        }
    };

    const int nsteps = r1 - r0 + 2;
    for (int ss = 0, t0 = r0 - 1; ss * 3 < nsteps; ++ss, t0 += 3) {
        const int ob = (ss & 1) * 3;
        step(phase_t<0>{}, t0, ob);
        step(phase_t<1>{}, t0 + 1, ob + 1);
        step(phase_t<2>{}, t0 + 2, ob + 2);
        __syncthreads();
        // y[row][c] = act2(b2 + plane1[c] + plane0[c-1] + plane2[c+1])
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = t0 - 1 + j;
            if (row < r0 || row >= r1) continue;
            const float* o = outr + (ob + j) * NPLANE * outw + 1 + wv * L::COLS;
            for (int cc = lane; cc < L::COLS; cc += 64) {
                const int c = wc0 + cc;
                float v = bias2 + (o[outw + cc] + o[cc - 1] + o[2 * outw + cc + 1]);
                // v_exp_f32 + v_rcp_f32 (1 ulp each; the argument's scaling by log2 e adds |v| * 6e-8 relative to e^-v:
                // <= 3e-7 absolute on y for |v| <= 20) instead of the ~25 instructions of expf and an IEEE division
                if (act2 == UOCR_ACT_SIGMOID) v = __builtin_amdgcn_rcpf(1.f + __expf(-v));
                if (c >= own_lo && c < own_hi) y[img + (size_t)row * wd + c] = v;
            }
        }
    }
}

// Sum of the block partials in float64, fixed order.  Block = FC consecutive partial columns x 32 segments of the
// blocks (8 loads in flight per thread), segments added in order; then the columns that are outputs (dw1: 9 x 16,
// db1 = row 9 of dW1^T, dw2, db2) are stored.  FC = 8: 256-thread blocks -- with 32 columns (1024 threads) a block
// needs sixteen free wave slots on one CU at once, which the other lanes' kernels rarely leave inside the page step:
// rocprofv3 averaged 23 us there for the 1 000 partials of the 8 x 1024 x 2048 step (4.9 us at 32 x 256 x 512).
constexpr int FC = 8;
__global__ __launch_bounds__(FC * 32) void pair_strip_finish(const float* __restrict__ partial, float* __restrict__ dw1,
                                                          float* __restrict__ db1, float* __restrict__ dw2,
                                                          float* __restrict__ db2, int nblocks, int use_b1, int use_b2,
                                                          int accumulate, float unscale) {
    constexpr int NSEG = 32;
    __shared__ double seg[NSEG][FC];
    const int o = threadIdx.x % FC, sg = threadIdx.x / FC, j = blockIdx.x * FC + o;
    double s = 0.0;
    if (j < PAIR_NPART) {
        const int per = (nblocks + NSEG - 1) / NSEG, b0 = sg * per, b1 = min(nblocks, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(b + k) * PAIR_NPART + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
        for (; b < b1; ++b) s += (double)partial[(size_t)b * PAIR_NPART + j];
    }
    seg[sg][o] = s;
    __syncthreads();
    if (sg != 0 || j >= PAIR_NPART) return;
#pragma unroll
    for (int k = 1; k < NSEG; ++k) s += seg[k][o];
    float* dst = nullptr;
    bool live = true;
    if (j < 144) dst = dw1 + j;                                           // dW1^T[tap][ch] = dw1[tap * 16 + ch]
    else if (j < 160) { dst = db1 + (j - 144); live = use_b1; }           // row 9: the ones copy
    else if (j >= 256 && j < 256 + 144) dst = dw2 + (j - 256);
    else if (j == 512) { dst = db2; live = use_b2; }
    if (!dst) return;
    s = live ? s * (double)unscale : 0.0;
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

template <int G>
size_t strip_lds_bytes(int nw, bool dx) {
    const size_t wave = (size_t)nw * Strip<G>::WAVE;
    const size_t ring = dx ? (size_t)NSLOT * NPLANE * (nw * Strip<G>::COLS + 16) : 0;
    const size_t red = (size_t)nw * PAIR_NPART;
    return sizeof(float) * (wave + ring > red ? wave + ring : red);
}

// G groups of 16 columns per wave, up to MAXW waves per block
template <int G, int MAXW>
int strip_bwd_launch(uocr_ctx* ctx, const float* x, const float* y, const float* dy, const float* w1,
                     const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                     float* dx, int n, int h, int w, float pad1, int use_b1, int use_b2, float alpha,
                     bool sig, int accumulate, float unscale) {
    const int nw = std::min(MAXW, (w + Strip<G>::COLS - 1) / Strip<G>::COLS);
    const int bwc = nw * Strip<G>::COLS;
    const int nbx = w <= bwc ? 1 : 1 + (w - bwc + (bwc - 2) - 1) / (bwc - 2);
    // bands: about one block per CU (a block of 8 waves holds 126 KB of LDS), rows per band a multiple of 3 + 1
    // so that the t = r0-1 .. r1 steps fill whole batches of three
    int bands = std::max(1, (ctx->cu_count + n * nbx - 1) / (n * nbx));
    if (ctx->opt_pair_band > 0) bands = (h + ctx->opt_pair_band - 1) / ctx->opt_pair_band;
    int band_h = std::max(4, (h + bands - 1) / bands);
    band_h = std::min(h, band_h);
    bands = (h + band_h - 1) / band_h;
    const size_t nblocks = (size_t)nbx * bands * n;
    UOCR_REQUIRE(ctx, bands <= 65535 && n <= 65535);
    int rc = UOCR_OK;
    float* partial = uocr_partial_buffer(ctx, nblocks * PAIR_NPART * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    const size_t lds = strip_lds_bytes<G>(nw, dx != nullptr);
    auto launch = [&](auto kernel) -> int {
        static bool attr_set = false;      // (per instantiation) allow more than 64 KB of dynamic LDS
        if (!attr_set) {
            UOCR_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kernel, dim3(nbx, bands, n), dim3(nw * 64), lds, ctx->stream, x, y, dy, w1, b1, w2, partial,
                           dx, h, w, band_h, pad1, use_b1, alpha);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    };
    const bool plain = nbx == 1 && w == bwc && pad1 == 0.f;      // MODE 0: no position is ever masked
    auto pick = [&](auto dxtag, auto sigtag) -> int {
        constexpr bool D = decltype(dxtag)::value, S = decltype(sigtag)::value;
        return plain ? launch(pair_strip_bwd_kernel<G, D, S, 0>) : launch(pair_strip_bwd_kernel<G, D, S, 1>);
    };
    if (dx) rc = sig ? pick(std::true_type{}, std::true_type{}) : pick(std::true_type{}, std::false_type{});
    else rc = sig ? pick(std::false_type{}, std::true_type{}) : pick(std::false_type{}, std::false_type{});
    if (rc != UOCR_OK) return rc;
    return uocr_pair_strip_finish(ctx, partial, dw1, db1, dw2, db2, (int)nblocks, use_b1, use_b2, accumulate, unscale);
}
}  // namespace

// float32 backward of the pair block on the strip kernels.  Returns UOCR_OK or an error code (ctx->err set).
int uocr_pair_strip_bwd_f32(uocr_ctx* ctx, const float* x, const float* y, const float* dy, const float* w1,
                            const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                            float* dx, int n, int h, int w, float pad1, int use_b1, int use_b2, float alpha,
                            bool sig, int accumulate, float unscale) {
    if (ctx->opt_pair_g == 2)
        return strip_bwd_launch<2, 16>(ctx, x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, dx, n, h, w, pad1, use_b1, use_b2,
                                       alpha, sig, accumulate, unscale);
    return strip_bwd_launch<4, 8>(ctx, x, y, dy, w1, b1, w2, dw1, db1, dw2, db2, dx, n, h, w, pad1, use_b1, use_b2, alpha,
                                  sig, accumulate, unscale);
}

// float32 forward of the pair block on the strip kernels
int uocr_pair_strip_fwd_f32(uocr_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* y, int n, int h, int w, float pad1, int use_b1, int use_b2,
                            float alpha, int act2) {
    constexpr int G = 4;
    const int nw = std::min(8, (w + Strip<G>::COLS - 1) / Strip<G>::COLS);
    const int bwc = nw * Strip<G>::COLS;
    const int nbx = w <= bwc ? 1 : 1 + (w - bwc + (bwc - 2) - 1) / (bwc - 2);
    // the forward block holds 38 KB of LDS and ~100 registers: two blocks per CU
    int bands = std::max(1, (2 * ctx->cu_count + n * nbx - 1) / (n * nbx));
    if (ctx->opt_pair_band > 0) bands = (h + ctx->opt_pair_band - 1) / ctx->opt_pair_band;
    int band_h = std::min(h, std::max(4, (h + bands - 1) / bands));
    bands = (h + band_h - 1) / band_h;
    UOCR_REQUIRE(ctx, bands <= 65535 && n <= 65535);
    const size_t lds = sizeof(float) * NSLOT * NPLANE * (bwc + 16);
    const bool plain = nbx == 1 && w == bwc && pad1 == 0.f;
    auto go = [&](auto mode, auto pf) {
        hipLaunchKernelGGL((pair_strip_fwd_kernel<G, decltype(mode)::value, decltype(pf)::value>), dim3(nbx, bands, n),
                           dim3(nw * 64), lds, ctx->stream, x, w1, b1, w2, b2, y, h, w, band_h, pad1, use_b1, use_b2, alpha,
                           act2);
    };
    auto by_pf = [&](auto mode) {
        if (ctx->opt_pair_pf == 0) go(mode, phase_t<0>{});
        else if (ctx->opt_pair_pf == 2) go(mode, phase_t<2>{});
        else go(mode, phase_t<1>{});
    };
    if (plain) by_pf(phase_t<0>{});
    else by_pf(phase_t<1>{});
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

int uocr_pair_strip_finish(uocr_ctx* ctx, const float* partial, float* dw1, float* db1, float* dw2, float* db2,
                           int nblocks, int use_b1, int use_b2, int accumulate, float unscale) {
    FinishDesc fd{};                                     // recorded when a deferred group is open (finish_group.h)
    fd.kind = FIN_PAIR;
    fd.partial = partial;
    fd.nblocks = nblocks;
    fd.ncols = fd.group_cols = PAIR_NPART;
    fd.row_stride = PAIR_NPART;
    fd.dw = dw1, fd.db = db1, fd.dw2 = dw2, fd.db2 = db2;
    fd.use_bias = use_b1, fd.use_bias2 = use_b2, fd.accumulate = accumulate;
    fd.unscale = unscale;
    if (uocr_finish_defer(ctx, fd)) return UOCR_OK;
    hipLaunchKernelGGL(pair_strip_finish, dim3((PAIR_NPART + FC - 1) / FC), dim3(FC * 32), 0, ctx->stream, partial, dw1, db1, dw2,
                       db2, nblocks, use_b1, use_b2, accumulate, unscale);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}
