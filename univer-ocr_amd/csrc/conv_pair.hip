// The Monochrome block as ONE forward and ONE backward kernel (float32):
//     x (1 ch) -> conv3x3 (1 -> 16, pad 1) -> LeakyReLU -> conv3x3 (16 -> 1, pad 1) -> [Sigmoid] -> y
// (reference: my_model/model.py:108-135 built from nn/layers/convolutional.py:62-145 and
// nn/layers/layers.py:377-418).
//
// Run layer by layer the 16-channel activation a1 (268 MB at 32x256x512) is written once and read three
// times per train step and its gradient (another 268 MB) is written once and read once: 1.6 GB of HBM
// traffic for 34 MB of real input and output.  Here neither tensor leaves the chip: a1 is recomputed from
// the 1-channel input where it is needed, and every 16-channel contraction runs on the matrix cores as a
// v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) over groups of 16 positions (16 consecutive columns of a
// row of the block's 16 x 32 region):
//   forward   Z[ch,pos]   = W1^T[ch,tap] Xcol[tap,pos]        3 MFMAs (9 taps padded to 12)
//             P[pos,tap]  = A1[pos,ch] W2^T[ch,tap]           4 MFMAs; y[p] = b2 + sum_t P[p+t-1, t] (LDS gather)
//   backward  Z^T[pos,ch], S^T[pos,ch] = Gcol^T[pos,tap] W2[tap,ch]            3 + 3   (g = dy * act2'(y))
//             d = S * lrelu'(Z)  (d_a1),  a = lrelu(Z)                          elementwise on the 4+4 results
//             dW2^T[tap,ch] += Gshift[tap,pos] A[pos,ch],  dW1^T[tap,ch] += Xshift[tap,pos] D[pos,ch]   4 + 4
//             U[pos,tap] = D[pos,ch] W1^T[ch,tap]  (dx only; D transposed through LDS)                  4
//             dx[p] = sum_s U[p-s+1, s]  (LDS gather)
// Operand trick: the K index of an MFMA may be permuted freely, so the 4 result registers of one MFMA
// (rows 4*(lane/16)+i) are fed straight back as the A or B operand of chunk i of the next one.
// The f32 MFMA rate equals the f32 vector rate (157 TF), but one MFMA replaces 16 v_fma plus their
// operand moves, and it runs beside the VALU work (masks, LDS addressing): the quad-lane VALU version of
// these kernels was issue-bound at 4 cycles per vector instruction (97 / 241 us; this one: see DESIGN.md).
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
// (MFMA accumulators in VGPRs: no v_accvgpr copies between the MFMAs and the VALU code that consumes them)
#include <type_traits>

#include "conv_pair.h"
#include "uocr_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int RH = 16, RW = 32;            // region of positions one tile iteration computes (a1 / d_a1)
constexpr int XH = RH + 2, XW = RW + 2;    // x / g tiles in LDS: region + halo 1
constexpr int C = 16;
constexpr int NA = 36 + 36 + 4 + 1;        // layout of a partial row: dw1[tap*4+j], dw2, db1[j], db2 per channel quad
constexpr int NPF = (XH * XW + 255) / 256; // x / g tile elements staged per thread
constexpr int TS = 20;                     // row stride of the per-wave transpose scratch

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float out_act(float v, int act) {
    return act == UOCR_ACT_SIGMOID ? 1.f / (1.f + expf(-v)) : v;
}

// the elements of the XH x XW tile with origin (ys, xs0) this thread stages: clamped offsets + in-image mask
struct Stage {
    size_t off[NPF];
    bool in[NPF];
    // what does not change while a block walks down its column strip: tile row, clamped image column, column in image
    int row[NPF], col[NPF];
    bool cin[NPF];
    __device__ __forceinline__ void prepare(int tid, int xs0, int wd) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 256;
            const int r = i / XW, c = i - r * XW, gx = xs0 + c;
            row[k] = r;
            col[k] = min(max(gx, 0), wd - 1);
            cin[k] = i < XH * XW && gx >= 0 && gx < wd;
        }
    }
    __device__ __forceinline__ void locate(int ys, int h, int wd) {       // tile origin row ys
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int gy = ys + row[k];
            in[k] = cin[k] && (unsigned)gy < (unsigned)h;
            off[k] = (size_t)min(max(gy, 0), h - 1) * wd + col[k];
        }
    }
};

// Block = column strip of TW outputs x rows [band*rows_per_block, +rows_per_block) of image blockIdx.z, walked
// tile by tile (TH x TW = 14 x 30 outputs need a1 on the 16 x 32 region); the next tile's x is in flight
// (registers) while the current one is computed.  Wave w owns region rows 4w..4w+3 (8 groups of 16 positions).
template <typename TA>
__global__ __launch_bounds__(256) void conv_pair_fwd_kernel(const TA* __restrict__ x, const float* __restrict__ w1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b2, TA* __restrict__ y,
                                                            int h, int wd, int rows_per_block, float pad1,
                                                            int use_b1, int use_b2, float alpha, int act2) {
    constexpr int TH = RH - 2, TW = RW - 2;
    __shared__ float xs[XH * XW];
    __shared__ float ps[RH * RW * 9];                    // P[pos][tap]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const TA* xb = x + (size_t)blockIdx.z * h * wd;
    TA* yb = y + (size_t)blockIdx.z * h * wd;
    // constant MFMA operands.  Z: A = W1^T (m = ch = n, k -> tap 4kc+kq), B = x at (pos n) + tap.
    // P: A = a1 register j (m = pos n, k -> ch 4kq+j), B = W2^T (k -> ch 4kq+j, n = tap)
    float w1a[3], w2b[4], bias4[4];
    int xoff[3];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        const int tap = 4 * kc + kq, t = tap < 9 ? tap : 8;
        w1a[kc] = tap < 9 ? w1[t * C + n] : 0.f;
        xoff[kc] = (t / 3) * XW + t % 3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w2b[j] = n < 9 ? w2[(n < 9 ? n : 0) * C + 4 * kq + j] : 0.f;
        bias4[j] = use_b1 ? b1[4 * kq + j] : 0.f;
    }
    const float bias2 = use_b2 ? b2[0] : 0.f;

    Stage st;
    float px[NPF];
    st.prepare(tid, x0 - 2, wd);
    st.locate(row_begin - 2, h, wd);
#pragma unroll
    for (int k = 0; k < NPF; ++k) px[k] = ld1(xb + st.off[k]);
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) xs[tid + k * 256] = st.in[k] ? px[k] : pad1;
        __syncthreads();
        if (y0 + TH < row_end) {                         // next tile: loads overlap this tile's math
            st.locate(y0 + TH - 2, h, wd);
#pragma unroll
            for (int k = 0; k < NPF; ++k) px[k] = ld1(xb + st.off[k]);
        }
        // region origin = (y0 - 1, x0 - 1), xs origin one further out: tap (ty,tx) of region (r,c) = xs[r+ty][c+tx]
#pragma unroll 2
        for (int k = 0; k < 8; ++k) {
            const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
            const float* xr = xs + r * XW + c0 + n;
            f32x4 z = {bias4[0], bias4[1], bias4[2], bias4[3]};
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) z = mfma(w1a[kc], xr[xoff[kc]], z);
            const int ay = y0 - 1 + r, ax = x0 - 1 + c0 + n;
            const bool inside = ay >= 0 && ay < h && ax >= 0 && ax < wd;   // outside: conv_2's zero padding
            f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = inside ? (z[j] >= 0.f ? z[j] : alpha * z[j]) : 0.f;
                p = mfma(a, w2b[j], p);
            }
            if (n < 9) {
#pragma unroll
                for (int v = 0; v < 4; ++v) ps[(r * RW + c0 + 4 * kq + v) * 9 + n] = p[v];
            }
        }
        __syncthreads();
        for (int p = tid; p < TH * TW; p += 256) {
            const int pr = p / TW, pc = p - pr * TW;
            const int gy = y0 + pr, gx = x0 + pc;
            float v = bias2;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) v += ps[((pr + ty) * RW + pc + tx) * 9 + ty * 3 + tx];
            if (gy < row_end && gx < wd) st1(yb + (size_t)gy * wd + gx, out_act(v, act2));
        }
    }
}

// Same walk for the backward; a tile owns all 16 x 32 positions of its region.  Per block one reduction of the
// two 16 x 16 accumulator tiles (4 VGPRs each) and of db1 / db2; partial[blk][q][NA] in the layout
// conv_pair_bwd_finish sums.
// DX: dx[p] = sum_s U[p - s + 1, s] needs U of the 8 neighbours of p, one ring beyond the tile.  Instead of
// recomputing d_a1 on a halo (x1.22 work, 14 x 30 tiles), the tile scatters its own U over its 18 x 34
// extended area: the inside goes to dx (edge pixels still incomplete), the ring -- what this tile contributes
// to pixels of its 8 neighbours -- to border[tile][100]; conv_pair_dx_border then adds, per edge pixel and in
// a fixed order, the ring entries of the neighbouring tiles.  No atomics: every float is written once.
constexpr int RING = 2 * XW + 2 * RH;      // top row, bottom row (XW each, corners included), left, right column

__device__ __forceinline__ int ring_index(int er, int ec) {      // (er, ec) in tile coordinates, on the ring
    if (er == -1) return ec + 1;
    if (er == RH) return XW + ec + 1;
    if (ec == -1) return 2 * XW + er;
    return 2 * XW + RH + er;
}
// dx[e] = sum_s U[e - s + 1, s] from the tile's U in LDS (us[pos][tap]): the inside of the tile goes to dx, the
// ring around it to border[tile_id] (see above).  dxb = dx of this image.
template <typename TA>
__device__ __forceinline__ void pair_dx_scatter(const float* us, TA* dxb, float* border, int tid, int y0, int x0,
                                                int h, int wd, size_t tile_id) {
    // Inner pixels (rows 1..RH-2, columns 1..RW-2): all nine sources lie in the tile -- nine LDS reads at
    // compile-time offsets, no tests
    constexpr int IH = RH - 2, IW = RW - 2;
    for (int e = tid; e < IH * IW; e += 256) {
        const int er = e / IW + 1, ec = e - (er - 1) * IW + 1;
        const float* u0 = us + ((er + 1) * RW + ec + 1) * 9;       // source of s = (0, 0)
        float v = 0.f;
#pragma unroll
        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) v += u0[-(sy * RW + sx) * 9 + sy * 3 + sx];
        const int gy = y0 + er, gx = x0 + ec;
        if (gy < h && gx < wd) st1(dxb + (size_t)gy * wd + gx, v);
    }
    // the two outer pixel frames: the tile's edge pixels (to dx, still lacking the neighbours' rings) and
    // the ring around the tile (to the border buffer): 4 rows of XW + (RH - 2) rows of 4 = 192 elements
    constexpr int NFRAME = 4 * XW + 4 * (RH - 2);
    for (int k = tid; k < NFRAME; k += 256) {
        int er, ec;
        if (k < 4 * XW) {
            const int rr = k / XW;
            er = rr == 0 ? -1 : rr == 1 ? 0 : rr == 2 ? RH - 1 : RH;
            ec = k - rr * XW - 1;
        } else {
            const int kk = k - 4 * XW, j = kk & 3;
            er = (kk >> 2) + 1;
            ec = j == 0 ? -1 : j == 1 ? 0 : j == 2 ? RW - 1 : RW;
        }
        float v = 0.f;
#pragma unroll
        for (int sy = 0; sy < 3; ++sy)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int qr = er + 1 - sy, qc = ec + 1 - sx;      // q = e - s + 1
                if (qr >= 0 && qr < RH && qc >= 0 && qc < RW) v += us[(qr * RW + qc) * 9 + sy * 3 + sx];
            }
        const int gy = y0 + er, gx = x0 + ec;
        if (gy < 0 || gy >= h || gx < 0 || gx >= wd) continue;
        if (er >= 0 && er < RH && ec >= 0 && ec < RW) st1(dxb + (size_t)gy * wd + gx, v);
        else border[tile_id * RING + ring_index(er, ec)] = v;
    }
}

// block reduction through LDS of the two dW^T accumulator tiles ([tap 4kq+v][ch n] per wave), db1 (channel n per
// lane) and db2 -> partial[(q * NA + k) * nblk + blk], the layout conv_pair_bwd_finish sums
__device__ __forceinline__ void pair_block_reduce(float (*red)[2][16][16], float (*reddb)[4][16], float* redb2,
                                                  f32x4 acc1, f32x4 acc2, float db1acc, float db2acc,
                                                  float* __restrict__ partial) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        red[wv][0][4 * kq + v][n] = acc1[v];
        red[wv][1][4 * kq + v][n] = acc2[v];
    }
    reddb[wv][kq][n] = db1acc;
    db2acc = wave_reduce_sum(db2acc);
    if (lane == 0) redb2[wv] = db2acc;
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int nblk = gridDim.x * gridDim.y * gridDim.z;
    float* out = partial + blk;                          // partial[(q * NA + k) * nblk + blk]: the finish kernel
    for (int i = tid; i < 4 * NA; i += 256) {            // then reads every sum's block partials contiguously
        const int q = i / NA, k = i - q * NA;
        float v = 0.f;
        if (k < 72) {
            const int which = k / 36, kk = k - which * 36, tap = kk >> 2, ch = q * 4 + (kk & 3);
            v = red[0][which][tap][ch] + red[1][which][tap][ch] + red[2][which][tap][ch] + red[3][which][tap][ch];
        } else if (k < 76) {
            const int ch = q * 4 + (k - 72);
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) v += reddb[w][g4][ch];
        } else {
            v = redb2[0] + redb2[1] + redb2[2] + redb2[3];
        }
        out[(size_t)i * nblk] = v;
    }
}

template <bool DX, bool SIG, typename TA>
__global__ __launch_bounds__(256) void conv_pair_bwd_kernel(const TA* __restrict__ x, const TA* __restrict__ yout,
                                                            const TA* __restrict__ dy,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ w2,
                                                            float* __restrict__ partial, TA* __restrict__ dx,
                                                            float* __restrict__ border, int h, int wd,
                                                            int rows_per_block, float pad1, int use_b1, float alpha) {
    constexpr int OFF = 0;
    constexpr int TH = RH, TW = RW;
    __shared__ float xs[XH * XW];
    __shared__ float gs[XH * XW];
    __shared__ float us[DX ? RH * RW * 9 : 1];           // U[pos][tap]
    __shared__ float tr[DX ? 4 * 16 * TS : 1];           // per wave: d_a1[pos][ch] of the current group
    __shared__ float red[4][2][16][16];
    __shared__ float reddb[4][4][16];
    __shared__ float redb2[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const TA *xb = x + img, *gb = dy + img, *yb = yout + img;
    // constant operands.  Z^T / S^T: A = x or g at (pos n) shifted by tap 4kc+kq, B = W1 / W2 [tap][ch = n]
    float w1b[3], w2b[3], w1u[4];
    int xoff[3], goff[3];
#pragma unroll
    for (int kc = 0; kc < 3; ++kc) {
        const int tap = 4 * kc + kq, t = tap < 9 ? tap : 8;
        w1b[kc] = tap < 9 ? w1[t * C + n] : 0.f;
        w2b[kc] = tap < 9 ? w2[t * C + n] : 0.f;
        xoff[kc] = (t / 3) * XW + t % 3;
        goff[kc] = (2 - t / 3) * XW + 2 - t % 3;         // g[pos - tap + 1]
    }
    // dW^T: A = g / x around pos 4kq+i seen from tap n (rows 9..15 of the result are unused)
    const bool tap_ok = n < 9;
    const int tn = tap_ok ? n : 8;
    const int xA = (tn / 3) * XW + tn % 3 + 4 * kq, gA = (2 - tn / 3) * XW + 2 - tn % 3 + 4 * kq;
    // U: A = d_a1[pos n][ch 4kc+kq] (transposed through LDS), B = W1[tap n][ch 4kc+kq]
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) w1u[kc] = tap_ok ? w1[tn * C + 4 * kc + kq] : 0.f;
    const float bias = use_b1 ? b1[n] : 0.f;
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // dW1^T, dW2^T [tap 4kq+v][ch n]
    float db1acc = 0.f, db2acc = 0.f;

    Stage st;
    st.prepare(tid, x0 - OFF - 1, wd);
    // the loads of the next tile are only issued here; their values are first touched at the LDS write after
    // the MFMAs (the sigmoid derivative included), so nothing in between waits for global memory
    float px[NPF], pg[NPF], py[SIG ? NPF : 1];
    auto prefetch = [&](int y0) {
        st.locate(y0 - OFF - 1, h, wd);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            px[k] = ld1(xb + st.off[k]);
            pg[k] = ld1(gb + st.off[k]);
            if constexpr (SIG) py[k] = ld1(yb + st.off[k]);
        }
    };
    prefetch(row_begin);
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) {
                float g = pg[k];
                if constexpr (SIG) g *= py[k] * (1.f - py[k]);
                xs[tid + k * 256] = st.in[k] ? px[k] : pad1;
                gs[tid + k * 256] = st.in[k] ? g : 0.f;
            }
        __syncthreads();
        if (y0 + TH < row_end) prefetch(y0 + TH);
        const int ry = y0, rx = x0;                      // region origin; xs / gs origin one further out
        // FULL: every position of the region lies inside the image and inside this block's band (true for all
        // tiles but the last partial one of a band / row): the per-element masks vanish from the loop body --
        // 12 v_cndmask and as many scalar ANDs per group of 16 positions
        auto groups = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll 2
            for (int k = 0; k < 8; ++k) {
                const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
                const float* xr = xs + r * XW + c0;
                const float* gr = gs + r * XW + c0;
                f32x4 z = {bias, bias, bias, bias}, s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < 3; ++kc) {
                    z = mfma(xr[n + xoff[kc]], w1b[kc], z);
                    s = mfma(gr[n + goff[kc]], w2b[kc], s);
                }
                // results: channel n at positions (r, c0 + 4kq + i)
                const int ay = ry + r;
                const bool row_in = ay >= 0 && ay < h, row_own = row_in && ay < row_end;
                float a[4], d[4], dn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float slope = z[i] >= 0.f ? 1.f : alpha;
                    if constexpr (FULL) {
                        a[i] = z[i] * slope;
                        d[i] = dn[i] = s[i] * slope;
                    } else {
                        const int c = c0 + 4 * kq + i, ax = rx + c;
                        const bool inside = row_in && ax >= 0 && ax < wd;
                        const bool owned = row_own && ax >= 0 && ax < wd;
                        a[i] = owned ? z[i] * slope : 0.f;
                        d[i] = inside ? s[i] * slope : 0.f;
                        dn[i] = owned ? d[i] : 0.f;
                    }
                    db1acc += dn[i];
                }
                // (lanes n >= 9 feed rows 9..15 of the two dW^T tiles, which nobody reads: their operand is whatever
                // the clamped tap address holds -- no select needed)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc2 = mfma(gr[i + gA], a[i], acc2);
                    acc1 = mfma(xr[i + xA], dn[i], acc1);
                }
                if constexpr (DX) {
                    float* t = tr + wv * 16 * TS;
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[(4 * kq + i) * TS + n] = d[i];
                    __builtin_amdgcn_wave_barrier();     // same wave: LDS executes its instructions in order
                    f32x4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) u = mfma(t[n * TS + 4 * kc + kq], w1u[kc], u);
                    __builtin_amdgcn_wave_barrier();
                    if (tap_ok) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) us[(r * RW + c0 + 4 * kq + v) * 9 + n] = u[v];
                    }
                }
            }
        };
        if (y0 + RH <= row_end && x0 + RW <= wd) groups(std::true_type{});
        else groups(std::false_type{});
        for (int p = tid; p < TH * TW; p += 256) {
            const int pr = p / TW, pc = p - pr * TW;
            if (y0 + pr < row_end && x0 + pc < wd) db2acc += gs[(pr + 1) * XW + pc + 1];
        }
        if constexpr (DX) {
            __syncthreads();
            pair_dx_scatter(us, dx + img, border, tid, y0, x0, h, wd,
                            ((size_t)blockIdx.z * ((h + RH - 1) / RH) + y0 / RH) * gridDim.x + blockIdx.x);
        }
    }
    pair_block_reduce(red, reddb, redb2, acc1, acc2, db1acc, db2acc, partial);
}

// ---------------------------------------------------------------------------------------------------------
// binary16 storage (UOCR_F16): the same two kernels on v_mfma_f32_16x16x16_f16 -- 16 K-values per MFMA
// instead of 4, so a whole 3x3 window (or all 16 channels) is ONE instruction: forward 2 MFMAs per group of
// 16 positions instead of 7, backward 5 instead of 18.  Operands are binary16 (x / dy / y are stored that way;
// the float32 master weights, a1 and d_a1 are rounded to binary16 as operands -- what a layer-by-layer run
// in this mode stores in HBM for a1 / d_a1 anyway); accumulation stays float32.
// LDS tiles hold PAIR WORDS: word[r][c] = (x[r][c], x[r][c+1]) as two halves, so the 4-half operand
// x[r][c..c+3] of any column c is one ds_read2_b32 (words c and c+2), no packing VALU.
// K permutation of a 3x3 window: k = 4*kq + j  <->  tap row kq (kq = 3: zero weights, row clamped),
// tap column j (j = 3: zero weight; the value read there lies outside the window and must only be finite).
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
constexpr int TSH = 24;                    // row stride (halves) of the per-wave d_a1 transpose scratch

__device__ __forceinline__ f32x4 mfma16(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pair_word(float lo, float hi) {
    const f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ f16x4 pack4(float a, float b, float c, float d) {
    const f32x4 v = {a, b, c, d};
    return __builtin_convertvector(v, f16x4);
}
__device__ __forceinline__ f16x4 window(const uint32_t* p) {      // halves c .. c+3 of a pair-word row
    const u32x2 v = {p[0], p[2]};
    return __builtin_bit_cast(f16x4, v);
}

// pair-word staging of the XH x XW tile with origin (ys, xs0): two clamped offsets per word + the in-image
// bits of its two halves (outside the image the LDS write substitutes the padding value)
struct StageH {
    int off[NPF], off2[NPF];
    uint32_t mask[NPF];
    // what does not change while a block walks down its column strip: the tile row / column of the thread's
    // elements, their clamped image columns and the in-image bits of the two columns of a pair word
    int row[NPF], col0[NPF], col1[NPF];
    uint32_t cmask[NPF];
    __device__ __forceinline__ void prepare(int tid, int xs0, int wd) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 256;
            const int r = i / XW, c = i - r * XW, gx = xs0 + c;
            row[k] = r;
            col0[k] = min(max(gx, 0), wd - 1);
            col1[k] = min(max(gx + 1, 0), wd - 1);
            cmask[k] = i < XH * XW ? ((gx >= 0 && gx < wd ? 0xFFFFu : 0u) | (gx + 1 >= 0 && gx + 1 < wd ? 0xFFFF0000u : 0u)) : 0u;
        }
    }
    __device__ __forceinline__ void locate(int ys, int h, int wd) {       // tile origin row ys
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int gy = ys + row[k];
            const int base = min(max(gy, 0), h - 1) * wd;
            off[k] = base + col0[k];
            off2[k] = base + col1[k];
            mask[k] = (unsigned)gy < (unsigned)h ? cmask[k] : 0u;
        }
    }
};
__device__ __forceinline__ uint32_t load_pair(const _Float16* p, int o0, int o1) {
    const f16x2 v = {p[o0], p[o1]};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t select_bits(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

__global__ __launch_bounds__(256) void conv_pair_fwd_h_kernel(const _Float16* __restrict__ x, const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ w2,
                                                              const float* __restrict__ b2, _Float16* __restrict__ y,
                                                              int h, int wd, int rows_per_block, float pad1,
                                                              int use_b1, int use_b2, float alpha, int act2) {
    constexpr int TH = RH - 2, TW = RW - 2;
    __shared__ uint32_t xs2[XH * XW];
    __shared__ float ps[RH * RW * 9];                    // P[pos][tap]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const _Float16* xb = x + (size_t)blockIdx.z * h * wd;
    _Float16* yb = y + (size_t)blockIdx.z * h * wd;
    // Z[ch, pos]: A = W1^T (m = ch = n, k = (tap row kq, tap column j)), B = window of x at pos n.
    // P[pos, tap]: A = a1 (m = pos = n, k = ch 4kq+j: this lane's four Z results), B = W2^T (k = ch, n = tap)
    f16x4 w1a, w2b;
    f32x4 bias4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w1a[j] = kq < 3 && j < 3 ? (_Float16)w1[(min(kq, 2) * 3 + min(j, 2)) * C + n] : (_Float16)0.f;
        w2b[j] = n < 9 ? (_Float16)w2[min(n, 8) * C + 4 * kq + j] : (_Float16)0.f;
        bias4[j] = use_b1 ? b1[4 * kq + j] : 0.f;
    }
    const int xoff = min(kq, 2) * XW + n;
    const float bias2 = use_b2 ? b2[0] : 0.f;
    const uint32_t padword = pair_word(pad1, pad1);

    StageH st;
    uint32_t px[NPF];
    st.prepare(tid, x0 - 2, wd);
    st.locate(row_begin - 2, h, wd);
#pragma unroll
    for (int k = 0; k < NPF; ++k) px[k] = load_pair(xb, st.off[k], st.off2[k]);
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) xs2[tid + k * 256] = select_bits(st.mask[k], px[k], padword);
        __syncthreads();
        if (y0 + TH < row_end) {                         // next tile: loads overlap this tile's math
            st.locate(y0 + TH - 2, h, wd);
#pragma unroll
            for (int k = 0; k < NPF; ++k) px[k] = load_pair(xb, st.off[k], st.off2[k]);
        }
        // region origin = (y0 - 1, x0 - 1), tile origin one further out.  FULL: the whole region lies inside the
        // image, conv_2's zero padding of a1 never applies
        auto groups = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll 2
            for (int k = 0; k < 8; ++k) {
                const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
                const f32x4 z = mfma16(w1a, window(xs2 + r * XW + c0 + xoff), bias4);
                bool inside = true;
                if constexpr (!FULL) {
                    const int ay = y0 - 1 + r, ax = x0 - 1 + c0 + n;
                    inside = ay >= 0 && ay < h && ax >= 0 && ax < wd;
                }
                float a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = inside ? (z[j] >= 0.f ? z[j] : alpha * z[j]) : 0.f;
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 p = mfma16(pack4(a[0], a[1], a[2], a[3]), w2b, zero);
                if (n < 9) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) ps[(r * RW + c0 + 4 * kq + v) * 9 + n] = p[v];
                }
            }
        };
        if (y0 >= 1 && y0 - 1 + RH <= h && x0 >= 1 && x0 - 1 + RW <= wd) groups(std::true_type{});
        else groups(std::false_type{});
        __syncthreads();
        for (int p = tid; p < TH * TW; p += 256) {
            const int pr = p / TW, pc = p - pr * TW;
            const int gy = y0 + pr, gx = x0 + pc;
            float v = bias2;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) v += ps[((pr + ty) * RW + pc + tx) * 9 + ty * 3 + tx];
            if (gy < row_end && gx < wd) st1(yb + (size_t)gy * wd + gx, act2 == UOCR_ACT_SIGMOID ? __builtin_amdgcn_rcpf(1.f + __expf(-v)) : v);   // (binary16 result: v_exp / v_rcp suffice)
        }
    }
}

// backward, binary16: Z^T / S^T [pos, ch] one MFMA each (A = window of x / of g mirrored, B = W1 / W2 rows),
// dW2^T / dW1^T [tap, ch] one each (A = g / x at the group's 4kq..4kq+3 positions seen from tap n, B = this
// lane's a / d_a1), U [pos, tap] one (A = d_a1 transposed through a binary16 LDS scratch, one ds_read_b64).
template <bool DX, bool SIG>
__global__ __launch_bounds__(256) void conv_pair_bwd_h_kernel(const _Float16* __restrict__ x,
                                                              const _Float16* __restrict__ yout,
                                                              const _Float16* __restrict__ dy,
                                                              const float* __restrict__ w1,
                                                              const float* __restrict__ b1,
                                                              const float* __restrict__ w2,
                                                              float* __restrict__ partial, _Float16* __restrict__ dx,
                                                              float* __restrict__ border, int h, int wd,
                                                              int rows_per_block, float pad1, int use_b1, float alpha) {
    constexpr int TH = RH, TW = RW;
    __shared__ uint32_t xs2[XH * XW];
    __shared__ uint32_t gs2[XH * XW];
    __shared__ float us[DX ? RH * RW * 9 : 1];           // U[pos][tap]
    __shared__ __attribute__((aligned(8))) _Float16 tr[DX ? 4 * 16 * TSH : 4];   // per wave: d_a1[pos][ch]
    __shared__ float red[4][2][16][16];
    __shared__ float reddb[4][4][16];
    __shared__ float redb2[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const _Float16 *xb = x + img, *gb = dy + img, *yb = yout + img;
    // constant operands.  S: g[pos - tap + 1] read left to right is tap column 2, 1, 0
    f16x4 w1b, w2b, w1u;
    const bool tap_ok = n < 9;
    const int tn = tap_ok ? n : 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool live = kq < 3 && j < 3;
        w1b[j] = live ? (_Float16)w1[(min(kq, 2) * 3 + min(j, 2)) * C + n] : (_Float16)0.f;
        w2b[j] = live ? (_Float16)w2[(min(kq, 2) * 3 + 2 - min(j, 2)) * C + n] : (_Float16)0.f;
        w1u[j] = tap_ok ? (_Float16)w1[tn * C + 4 * kq + j] : (_Float16)0.f;
    }
    const int xoff = min(kq, 2) * XW + n, goff = (2 - min(kq, 2)) * XW + n;
    const int xA = (tn / 3) * XW + tn % 3 + 4 * kq, gA = (2 - tn / 3) * XW + 2 - tn % 3 + 4 * kq;
    const float bias = use_b1 ? b1[n] : 0.f;
    const uint32_t padword = pair_word(pad1, pad1);
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // dW1^T, dW2^T [tap 4kq+v][ch n]
    float db1acc = 0.f, db2acc = 0.f;

    StageH st;
    st.prepare(tid, x0 - 1, wd);
    uint32_t px[NPF], pg[NPF], py[SIG ? NPF : 1];
    auto prefetch = [&](int y0) {
        st.locate(y0 - 1, h, wd);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            px[k] = load_pair(xb, st.off[k], st.off2[k]);
            pg[k] = load_pair(gb, st.off[k], st.off2[k]);
            if constexpr (SIG) py[k] = load_pair(yb, st.off[k], st.off2[k]);
        }
    };
    prefetch(row_begin);
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) {
                uint32_t g = pg[k];
                if constexpr (SIG) {                     // g *= y (1 - y) in float32, one rounding back to binary16
                    const f16x2 gh = __builtin_bit_cast(f16x2, pg[k]), yh = __builtin_bit_cast(f16x2, py[k]);
                    const float y0f = (float)yh[0], y1f = (float)yh[1];
                    g = pair_word((float)gh[0] * (y0f * (1.f - y0f)), (float)gh[1] * (y1f * (1.f - y1f)));
                }
                xs2[tid + k * 256] = select_bits(st.mask[k], px[k], padword);
                gs2[tid + k * 256] = g & st.mask[k];
            }
        __syncthreads();
        if (y0 + TH < row_end) prefetch(y0 + TH);
        const int ry = y0, rx = x0;                      // region origin; tile origin one further out
        auto groups = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll 2
            for (int k = 0; k < 8; ++k) {
                const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
                const uint32_t* xr = xs2 + r * XW + c0;
                const uint32_t* gr = gs2 + r * XW + c0;
                const f32x4 zinit = {bias, bias, bias, bias}, zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 z = mfma16(window(xr + xoff), w1b, zinit);
                const f32x4 s = mfma16(window(gr + goff), w2b, zero);
                // results: channel n at positions (r, c0 + 4kq + i)
                const int ay = ry + r;
                const bool row_in = ay >= 0 && ay < h, row_own = row_in && ay < row_end;
                float a[4], d[4], dn[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float slope = z[i] >= 0.f ? 1.f : alpha;
                    if constexpr (FULL) {
                        a[i] = z[i] * slope;
                        d[i] = dn[i] = s[i] * slope;
                    } else {
                        const int c = c0 + 4 * kq + i, ax = rx + c;
                        const bool inside = row_in && ax >= 0 && ax < wd;
                        const bool owned = row_own && ax >= 0 && ax < wd;
                        a[i] = owned ? z[i] * slope : 0.f;
                        d[i] = inside ? s[i] * slope : 0.f;
                        dn[i] = owned ? d[i] : 0.f;
                    }
                    db1acc += dn[i];
                }
                const f16x4 dn4 = pack4(dn[0], dn[1], dn[2], dn[3]);
                acc2 = mfma16(window(gr + gA), pack4(a[0], a[1], a[2], a[3]), acc2);
                acc1 = mfma16(window(xr + xA), dn4, acc1);
                if constexpr (DX) {
                    _Float16* t = tr + wv * 16 * TSH;
                    const f16x4 d4 = FULL ? dn4 : pack4(d[0], d[1], d[2], d[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[(4 * kq + i) * TSH + n] = d4[i];
                    __builtin_amdgcn_wave_barrier();     // same wave: LDS executes its instructions in order
                    const f16x4 dt = *reinterpret_cast<const f16x4*>(t + n * TSH + 4 * kq);
                    const f32x4 u = mfma16(dt, w1u, zero);
                    __builtin_amdgcn_wave_barrier();
                    if (tap_ok) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) us[(r * RW + c0 + 4 * kq + v) * 9 + n] = u[v];
                    }
                }
            }
        };
        if (y0 + RH <= row_end && x0 + RW <= wd) groups(std::true_type{});
        else groups(std::false_type{});
        for (int p = tid; p < TH * TW; p += 256) {
            const int pr = p / TW, pc = p - pr * TW;
            if (y0 + pr < row_end && x0 + pc < wd)
                db2acc += (float)__builtin_bit_cast(f16x2, gs2[(pr + 1) * XW + pc + 1])[0];
        }
        if constexpr (DX) {
            __syncthreads();
            pair_dx_scatter(us, dx + img, border, tid, y0, x0, h, wd,
                            ((size_t)blockIdx.z * ((h + RH - 1) / RH) + y0 / RH) * gridDim.x + blockIdx.x);
        }
    }
    pair_block_reduce(red, reddb, redb2, acc1, acc2, db1acc, db2acc, partial);
}

// block (k, q): float64 sum of the block partials -> dw1 / dw2 [tap*16 + q*4 + j], db1[q*4 + j], db2
__global__ __launch_bounds__(256) void conv_pair_bwd_finish(const float* __restrict__ partial, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2, int nblocks, int use_b1,
                                                            int use_b2, int accumulate, float unscale) {
    __shared__ double smem[16];
    const int k = blockIdx.x, q = blockIdx.y;
    double s = 0.0;
    const float* src = partial + (size_t)(q * NA + k) * nblocks;
    for (int blk = threadIdx.x; blk < nblocks; blk += blockDim.x) s += (double)src[blk];
    s = block_reduce_sum(s, smem);
    if (threadIdx.x != 0) return;
    float* dst;
    if (k < 36) {
        dst = dw1 + (k / 4) * C + q * 4 + (k % 4);
    } else if (k < 72) {
        dst = dw2 + ((k - 36) / 4) * C + q * 4 + (k % 4);
    } else if (k < 76) {
        dst = db1 + q * 4 + (k - 72);
        if (!use_b1) s = 0.0;
    } else {
        if (q != 0) return;                              // all four lanes of a quad summed the same g
        dst = db2;
        if (!use_b2) s = 0.0;
    }
    s *= (double)unscale;                                // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// dx of the tile-edge pixels += what the 8 neighbouring tiles wrote on their rings (fixed order: N, S, W, E,
// NW, NE, SW, SE).  One block of 128 threads per tile (grid = tiles_x, tiles_y, images), thread k < 92 = edge
// pixel k: 2 * RW + 2 * (RH - 2) per tile; tile coordinates come from the block index (the first version
// decoded a flat index with five integer divisions per pixel and ran 19 us for 15 MB of traffic).
template <typename TA>
__global__ __launch_bounds__(128) void conv_pair_dx_border(const float* __restrict__ border, TA* __restrict__ dx,
                                                           int n, int h, int wd, int tiles_y, int tiles_x) {
    constexpr int EDGE = 2 * RW + 2 * (RH - 2);
    const int k = threadIdx.x;
    if (k >= EDGE) return;
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    int pr, pc;
    if (k < RW) { pr = 0; pc = k; }
    else if (k < 2 * RW) { pr = RH - 1; pc = k - RW; }
    else if (k < 2 * RW + RH - 2) { pr = k - 2 * RW + 1; pc = 0; }
    else { pr = k - 2 * RW - (RH - 2) + 1; pc = RW - 1; }
    const int gy = ty * RH + pr, gx = tx * RW + pc;
    if (gy >= h || gx >= wd) return;
    const int dys[8] = {-1, 1, 0, 0, -1, -1, 1, 1}, dxs[8] = {0, 0, -1, 1, -1, 1, -1, 1};
    float add = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ny = ty + dys[j], nx = tx + dxs[j];
        const int er = pr - RH * dys[j], ec = pc - RW * dxs[j];              // this pixel seen from tile (ny, nx)
        const bool touches = er >= -1 && er <= RH && ec >= -1 && ec <= RW && ny >= 0 && ny < tiles_y && nx >= 0 &&
                             nx < tiles_x;
        if (touches) add += border[(((size_t)b * tiles_y + ny) * tiles_x + nx) * RING + ring_index(er, ec)];
    }
    TA* p = dx + ((size_t)b * h + gy) * wd + gx;
    st1(p, ld1(p) + add);
}

// rows per block = a multiple of the tile height th giving at most max_blocks blocks (measured at
// 32 x 256 x 512: forward 8192 blocks (2048: 74, 4096: 71, 8192: 69 us), backward with dx 1024, without dx 2048)
int pair_rows_per_block(int strips, int h, int n, int th, unsigned max_blocks) {
    int rows = th;
    while (rows < h && (size_t)strips * ((h + rows - 1) / rows) * n > max_blocks) rows += th;
    return rows;
}

// the backward kernel of a storage type: float32 MFMAs for float, binary16 MFMAs for _Float16
template <typename TA>
auto pair_bwd_kernel_for(bool dx, bool sig) {
    if constexpr (std::is_same<TA, _Float16>::value) {
        return dx ? (sig ? conv_pair_bwd_h_kernel<true, true> : conv_pair_bwd_h_kernel<true, false>)
                  : (sig ? conv_pair_bwd_h_kernel<false, true> : conv_pair_bwd_h_kernel<false, false>);
    } else {
        return dx ? (sig ? conv_pair_bwd_kernel<true, true, TA> : conv_pair_bwd_kernel<true, false, TA>)
                  : (sig ? conv_pair_bwd_kernel<false, true, TA> : conv_pair_bwd_kernel<false, false, TA>);
    }
}

int check_pair(uocr_ctx* ctx, int dtype, int n, int h, int w, int cmid, int act2) {
    if (UOCR_DTYPE_BASE(dtype) != UOCR_F32 && UOCR_DTYPE_BASE(dtype) != UOCR_F16)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: float32 / float16 only");
    if (cmid != C) UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: 16 middle channels only (got %d)", cmid);
    if (act2 != UOCR_ACT_NONE && act2 != UOCR_ACT_SIGMOID)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: output activation must be none or sigmoid");
    UOCR_REQUIRE(ctx, n > 0 && h > 0 && w > 0 && n <= 65535);
    return UOCR_OK;
}

}  // namespace

extern "C" int uocr_conv_pair_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w1, const void* b1,
                                  const void* w2, const void* b2, void* y, int n, int h, int w, int cmid,
                                  double pad_value1, int use_bias1, int use_bias2, double alpha1, int act2) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && w1 && b1 && w2 && b2 && y);
    int rc = check_pair(ctx, dtype, n, h, w, cmid, act2);
    if (rc != UOCR_OK) return rc;
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F32 && ctx->opt_pair)
        return uocr_pair_strip_fwd_f32(ctx, (const float*)x, (const float*)w1, (const float*)b1, (const float*)w2,
                                       (const float*)b2, (float*)y, n, h, w, (float)pad_value1, use_bias1, use_bias2,
                                       (float)alpha1, act2);
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_pair)
        return uocr_pair_strip_fwd_f16(ctx, x, (const float*)w1, (const float*)b1, (const float*)w2, (const float*)b2, y, n,
                                       h, w, (float)pad_value1, use_bias1, use_bias2, (float)alpha1, act2);
    const int strips = (w + RW - 3) / (RW - 2);
    const int rows_per_block = pair_rows_per_block(strips, h, n, RH - 2, 8192u);
    const dim3 grid(strips, (h + rows_per_block - 1) / rows_per_block, n);
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F16)
        hipLaunchKernelGGL(conv_pair_fwd_h_kernel, grid, dim3(256), 0, ctx->stream, (const _Float16*)x, (const float*)w1,
                           (const float*)b1, (const float*)w2, (const float*)b2, (_Float16*)y, h, w, rows_per_block,
                           (float)pad_value1, use_bias1, use_bias2, (float)alpha1, act2);
    else
        hipLaunchKernelGGL(conv_pair_fwd_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)x,
                           (const float*)w1, (const float*)b1, (const float*)w2, (const float*)b2, (float*)y, h, w,
                           rows_per_block, (float)pad_value1, use_bias1, use_bias2, (float)alpha1, act2);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

extern "C" int uocr_conv_pair_bwd(uocr_ctx* ctx, int dtype, const void* x, const void* y, const void* dy,
                                  const void* w1, const void* b1, const void* w2, void* dw1, void* db1, void* dw2,
                                  void* db2, void* dx, int n, int h, int w, int cmid, double pad_value1,
                                  int use_bias1, int use_bias2, double alpha1, int act2, int accumulate) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && y && dy && w1 && b1 && w2 && dw1 && db1 && dw2 && db2);
    int rc = check_pair(ctx, dtype, n, h, w, cmid, act2);
    if (rc != UOCR_OK) return rc;
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F32 && ctx->opt_pair)
        return uocr_pair_strip_bwd_f32(ctx, (const float*)x, (const float*)y, (const float*)dy, (const float*)w1,
                                       (const float*)b1, (const float*)w2, (float*)dw1, (float*)db1, (float*)dw2,
                                       (float*)db2, (float*)dx, n, h, w, (float)pad_value1, use_bias1, use_bias2,
                                       (float)alpha1, act2 == UOCR_ACT_SIGMOID, accumulate, 1.f);
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_pair)
        return uocr_pair_strip_bwd_f16(ctx, x, y, dy, (const float*)w1, (const float*)b1, (const float*)w2, (float*)dw1,
                                       (float*)db1, (float*)dw2, (float*)db2, dx, n, h, w, (float)pad_value1, use_bias1,
                                       use_bias2, (float)alpha1, act2 == UOCR_ACT_SIGMOID, accumulate,
                                       (float)uocr_grad_unscale(dtype));
    const int strips = (w + RW - 1) / RW, tiles_y = (h + RH - 1) / RH;
    const int rows_per_block = pair_rows_per_block(strips, h, n, RH, dx ? 1024u : 2048u);
    const int bands = (h + rows_per_block - 1) / rows_per_block;
    const int nblocks = strips * bands * n;
    const size_t partial_bytes = (size_t)nblocks * 4 * NA * sizeof(float);
    const size_t border_bytes = dx ? (size_t)n * tiles_y * strips * RING * sizeof(float) : 0;
    rc = uocr_need_workspace(ctx, partial_bytes + border_bytes);
    if (rc != UOCR_OK) return rc;
    float* partial = (float*)ctx->workspace;
    float* border = (float*)((char*)ctx->workspace + partial_bytes);
    const dim3 grid(strips, bands, n);
    const bool sig = act2 == UOCR_ACT_SIGMOID;
    UOCR_DISPATCH_TA(ctx, dtype, {
        hipLaunchKernelGGL(pair_bwd_kernel_for<TA>(dx != nullptr, sig), grid, dim3(256), 0, ctx->stream, (const TA*)x,
                           (const TA*)y, (const TA*)dy, (const float*)w1, (const float*)b1, (const float*)w2, partial,
                           (TA*)dx, dx ? border : (float*)nullptr, h, w, rows_per_block, (float)pad_value1, use_bias1,
                           (float)alpha1);
        UOCR_LAUNCH_CHECK(ctx);
        if (dx) {
            hipLaunchKernelGGL((conv_pair_dx_border<TA>), dim3(strips, tiles_y, n), dim3(128), 0, ctx->stream,
                               (const float*)border, (TA*)dx, n, h, w, tiles_y, strips);
            UOCR_LAUNCH_CHECK(ctx);
        }
    });
    hipLaunchKernelGGL(conv_pair_bwd_finish, dim3(NA, 4), dim3(256), 0, ctx->stream, (const float*)partial,
                       (float*)dw1, (float*)db1, (float*)dw2, (float*)db2, nblocks, use_bias1, use_bias2, accumulate,
                       (float)uocr_grad_unscale(dtype));
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}
