// Upsample2D(2) followed by Convolutional2D(5x5, stride 1, padding 2, zero padding value), 4 -> 4 channels,
// as ONE op on the LOW-RESOLUTION tensor (float32): the decoder blocks of the Line net
// (reference: my_model/model.py:194-247 `up_i` = nn/layers/upsample.py:21-39 + nn/layers/convolutional.py:62-145).
//
// Nearest-neighbour upsampling makes 4 high-res pixels share one source pixel, so the 5x5 window of an
// output pixel only sees a 3x3 block of source pixels, with the 25 taps summed in groups that depend on
// the output's parity phase (py, px).  Per axis, source offset m = floor((phase + k - 2) / 2):
//        phase 0: k {0,1} -> -1, {2,3} -> 0, {4} -> +1          phase 1: k {0} -> -1, {1,2} -> 0, {3,4} -> +1
//   Weff[tap m][c][(phase, o)] = sum of w[ky][kx][c][o] over the groups  (9 x 4 x 16, made by a tiny kernel)
//   y[2P + phase][o] = b[o] + sum_{m,c} Weff[m][c][(phase,o)] xl[P + m][c]
// i.e. a 3x3 convolution 4 -> 16 on the low-res tensor followed by depth-to-space: 2.8x fewer multiply-adds
// than the 5x5 on the upsampled tensor, no 4x larger intermediate (written once, read twice per step before),
// and -- 16 output columns, K = 36 -- a shape the matrix cores take without padding:
//   forward   Y^T[(phase,o), pos] = Weff^T[(phase,o), (m,c)] Xcol[(m,c), pos]      9 MFMAs (16x16x4) per 16 positions;
//             the 4 result registers of a lane are the 4 channels of ONE high-res pixel -> one 16-B store
//   dw        dWeff[(m,c), (phase,o)] += Xcol^T[(m,c), pos] dY[pos, (phase,o)]      12 MFMAs per 16 positions,
//             then dw[ky][kx] (+)= sum over the phases of the group it belongs to (finish kernel)
//   dx        dxl[Q][c] = sum_{m,(phase,o)} Weff[m][c][(phase,o)] dy[2(Q - m) + phase][o]: only 4 output columns,
//             so it stays on the vector ALU: one tap (64 weights in SGPRs) per iteration of a rolled loop,
//             dy window from an LDS tile, optional LeakyReLU' epilogue (ActMask as in conv_dims.h)
// Sums of weights first / different summation order: results agree with the layer-by-layer path to
// float32 rounding (tests: 1e-5 normalised), not bit for bit.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
// (MFMA accumulators in VGPRs: no v_accvgpr copies between the MFMAs and the VALU code that consumes them)
#include "up_phase.h"
#include "finish_group.h"
#include "conv_dims.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CH = 4;
constexpr int RH = 16, RW = 32;                 // low-res positions per tile
constexpr int XH = RH + 2, XW = RW + 2;         // xl tile: halo 1
constexpr int XPLANE = 624;                     // >= XH*XW, = 16 mod 32: the 4 channel planes of a half wave on disjoint banks
constexpr int NWEFF = 9 * CH * 16;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
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

// taps k of one axis that land on source offset m (index mi = m + 1) for output parity `phase`: [lo, hi)
__host__ __device__ __forceinline__ void tap_group(int phase, int mi, int& lo, int& hi) {
    if (phase == 0) {
        lo = 2 * mi;
        hi = mi == 2 ? 5 : 2 * mi + 2;
    } else {
        lo = mi == 0 ? 0 : 2 * mi - 1;
        hi = mi == 0 ? 1 : 2 * mi + 1;
    }
}

// weff[(m*4 + c)*16 + phase*4 + o], m = my*3 + mx, phase = py*2 + px
__global__ __launch_bounds__(256) void upconv_weff_kernel(const float* __restrict__ w, float* __restrict__ weff) {
    for (int i = threadIdx.x; i < NWEFF; i += blockDim.x) {
        const int o = i & 3, phase = (i >> 2) & 3, c = (i >> 4) & 3, m = i >> 6;
        int ylo, yhi, xlo, xhi;
        tap_group(phase >> 1, m / 3, ylo, yhi);
        tap_group(phase & 1, m % 3, xlo, xhi);
        float s = 0.f;
        for (int ky = ylo; ky < yhi; ++ky)
            for (int kx = xlo; kx < xhi; ++kx) s += w[((ky * 5 + kx) * CH + c) * CH + o];
        weff[i] = s;
    }
}

// xl tile (halo 1) of the region at low-res origin (ry, rx), channel-planar in LDS; zero outside the image
template <typename TA>
__device__ __forceinline__ void stage_planar(float* __restrict__ xs, const TA* __restrict__ xb, int ry, int rx,
                                             int hl, int wl, int tid) {
    if constexpr (sizeof(TA) == 4) {
        stage_batched<XH * XW, 256, 3, float4>(
            tid,
            [&](int i, bool& inside) {
                const int r = i / XW, c = i - r * XW;
                const int gy = ry - 1 + r, gx = rx - 1 + c;
                inside = gy >= 0 && gy < hl && gx >= 0 && gx < wl;
                return reinterpret_cast<const float4*>(xb + ((size_t)min(max(gy, 0), hl - 1) * wl + min(max(gx, 0), wl - 1)) * CH);
            },
            [&](int i, float4 v, bool inside) {
                if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
                xs[i] = v.x;
                xs[XPLANE + i] = v.y;
                xs[2 * XPLANE + i] = v.z;
                xs[3 * XPLANE + i] = v.w;
            });
        return;
    }
    for (int i = tid; i < XH * XW; i += 256) {
        const int r = i / XW, c = i - r * XW;
        const int gy = ry - 1 + r, gx = rx - 1 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < hl && gx >= 0 && gx < wl) v = ld4(xb + ((size_t)gy * wl + gx) * CH);
        xs[i] = v.x;
        xs[XPLANE + i] = v.y;
        xs[2 * XPLANE + i] = v.z;
        xs[3 * XPLANE + i] = v.w;
    }
}

// forward.  Block = 16 x 32 low-res positions per tile iteration over a band of rows; wave w owns rows
// 4w..4w+3 (8 groups of 16 consecutive positions).
template <typename TA>
__global__ __launch_bounds__(256) void upconv_fwd_kernel(const TA* __restrict__ xl, const float* __restrict__ w,
                                                         const float* __restrict__ bias, TA* __restrict__ y,
                                                         int hl, int wl, int rows_per_block, int use_bias, int act,
                                                         float alpha, float* __restrict__ weff_out) {
    __shared__ float xs[CH * XPLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    // the per-phase 3 x 3 weights the backward-data kernel of this layer wants (upconv_weff_kernel's table), written by
    // block 0 while it is here anyway: the caller hands the buffer back to uocr_upconv2x_bwd_data, which then needs no
    // launch of its own for them (a launch costs the lane 8-10 us inside the page step)
    if (weff_out && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        for (int i = tid; i < NWEFF; i += 256) {
            const int o = i & 3, phase = (i >> 2) & 3, c = (i >> 4) & 3, m = i >> 6;
            int ylo, yhi, xlo, xhi;
            tap_group(phase >> 1, m / 3, ylo, yhi);
            tap_group(phase & 1, m % 3, xlo, xhi);
            float s = 0.f;
            for (int ky = ylo; ky < yhi; ++ky)
                for (int kx = xlo; kx < xhi; ++kx) s += w[((ky * 5 + kx) * CH + c) * CH + o];
            weff_out[i] = s;
        }
    }
    const int rx = blockIdx.x * RW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(hl, row_begin + rows_per_block);
    const TA* xb = xl + (size_t)blockIdx.z * hl * wl * CH;
    const int W = 2 * wl;
    TA* yb = y + (size_t)blockIdx.z * (2 * hl) * W * CH;
    // A = Weff^T: lane (m = (phase,o) = n, k = channel kq), one register per tap, summed here from the 25
    // taps of w (each belongs to exactly one source offset for this lane's phase)
    float wa[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int ylo, yhi, xlo, xhi;
        tap_group(n >> 3, t / 3, ylo, yhi);
        tap_group((n >> 2) & 1, t % 3, xlo, xhi);
        float s = 0.f;
        for (int ky = ylo; ky < yhi; ++ky)
            for (int kx = xlo; kx < xhi; ++kx) s += w[((ky * 5 + kx) * CH + kq) * CH + (n & 3)];
        wa[t] = s;
    }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (use_bias) b4 = f32x4{bias[0], bias[1], bias[2], bias[3]};
    for (int ry = row_begin; ry < row_end; ry += RH) {
        __syncthreads();
        stage_planar(xs, xb, ry, rx, hl, wl, tid);
        __syncthreads();
#pragma unroll 2
        for (int k = 0; k < 8; ++k) {
            const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
            // B = Xcol: lane (k = channel kq, n = position): x at position + tap
            const float* xr = xs + kq * XPLANE + r * XW + c0 + n;
            f32x4 acc = b4;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = mfma(wa[t], xr[(t / 3) * XW + t % 3], acc);
            // lane: phase = kq, registers = the 4 channels of high-res pixel (2P + phase)
            const int py = ry + r, pxl = rx + c0 + n;
            if (py < row_end && pxl < wl) {
                float4 out;
                out.x = act_apply(acc[0], act, alpha);
                out.y = act_apply(acc[1], act, alpha);
                out.z = act_apply(acc[2], act, alpha);
                out.w = act_apply(acc[3], act, alpha);
                st4(yb + ((size_t)(2 * py + (kq >> 1)) * W + 2 * pxl + (kq & 1)) * CH, out);
            }
        }
    }
}

// dw: partial[blk][48][16] (rows 36..47 unused) + partial_db[blk][4]
template <typename TA>
__global__ __launch_bounds__(256) void upconv_wgrad_kernel(const TA* __restrict__ xl, const TA* __restrict__ dy,
                                                           float* __restrict__ partial, int hl, int wl,
                                                           int rows_per_block) {
    __shared__ float xs[CH * XPLANE];
    __shared__ float red[4][48][16];
    __shared__ float reddb[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int rx = blockIdx.x * RW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(hl, row_begin + rows_per_block);
    const TA* xb = xl + (size_t)blockIdx.z * hl * wl * CH;
    const int W = 2 * wl;
    const TA* gb = dy + (size_t)blockIdx.z * (2 * hl) * W * CH;
    // A = Xcol^T: lane (m = K index 16j + n -> tap K/4, channel K%4; k = position 4i + kq)
    int aoff[3];
    bool aok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int K = 16 * j + n;
        aok[j] = K < 36;
        const int Kc = aok[j] ? K : 35, t = Kc >> 2;
        aoff[j] = (Kc & 3) * XPLANE + (t / 3) * XW + t % 3 + kq;
    }
    // B = dY: lane (k = position 4i + kq, n = (phase, o))
    const int boff = ((n >> 3) * W + ((n >> 2) & 1)) * CH + (n & 3);
    f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float dbacc = 0.f;
    for (int ry = row_begin; ry < row_end; ry += RH) {
        __syncthreads();
        stage_planar(xs, xb, ry, rx, hl, wl, tid);
        __syncthreads();
#pragma unroll 2
        for (int k = 0; k < 8; ++k) {
            const int gi = wv * 8 + k, r = gi >> 1, c0 = (gi & 1) * 16;
            const int py = ry + r;
            const bool row_ok = py < row_end;
            const float* xr = xs + r * XW + c0;
            const TA* gr = gb + ((size_t)(2 * min(py, hl - 1)) * W + 2 * rx) * CH + boff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pl = c0 + 4 * i + kq;          // position of this lane's K slot
                const bool ok = row_ok && rx + pl < wl;
                float g = ld1(gr + (size_t)2 * min(pl, wl - 1 - rx) * CH);
                g = ok ? g : 0.f;
                dbacc += g;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float xv = xr[4 * i + aoff[j]];
                    acc[j] = mfma(aok[j] ? xv : 0.f, g, acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wv][16 * j + 4 * kq + v][n] = acc[j][v];
    reddb[wv][lane] = dbacc;
    __syncthreads();
    // The block's partial row: dw and db themselves (the four phase entries of dWeff added HERE, up_phase.h), so that the
    // finish is a plain, coalesced column sum of a 404-column matrix (it used to gather four scattered entries of a
    // 580-column row per output and block: 10-13 us inside the step)
    if (tid < 4) {                                       // db[o]: lanes with (n & 3) == o
        float s = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int l = tid; l < 64; l += 4) s += reddb[w][l];
        reddb[0][tid] = s;                               // (read back below as entry 576 + o)
    }
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    up4_write_row(partial + (size_t)blk * UP4_NOUT, [&](int i) {
        if (i >= 576) return reddb[0][i - 576];
        const int K = i >> 4, col = i & 15;
        return red[0][K][col] + red[1][K][col] + red[2][K][col] + red[3][K][col];
    }, tid, 256);
}

// out[a] (+)= unscale * (float64 column sum a of the block partials [nblocks][ncols]), a < ndw -> dw, else db.  Block = 8
// columns x 32 segments of the rows (pair_strip_finish's shape: 256 threads are schedulable inside the page step, eight
// loads in flight per thread, 32-byte coalesced segments), segments added in order.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nblocks, int ncols,
                                                            int ndw, float* __restrict__ dw, float* __restrict__ db,
                                                            int use_bias, int accumulate, float unscale) {
    constexpr int FC = 8, NSEG = 32;
    __shared__ double seg[NSEG][FC];
    const int o = threadIdx.x % FC, sg = threadIdx.x / FC, j = blockIdx.x * FC + o;
    double s = 0.0;
    if (j < ncols) {
        const int per = (nblocks + NSEG - 1) / NSEG, b0 = sg * per, b1 = min(nblocks, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(b + k) * ncols + j];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
        for (; b < b1; ++b) s += (double)partial[(size_t)b * ncols + j];
    }
    seg[sg][o] = s;
    __syncthreads();
    if (sg != 0 || j >= ncols) return;
#pragma unroll
    for (int k = 1; k < NSEG; ++k) s += seg[k][o];
    float* dst = j < ndw ? dw + j : db + (j - ndw);
    if (j >= ndw && !use_bias) s = 0.0;
    s *= (double)unscale;                                // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// dx on the vector ALU.  Block = 16 x 32 low-res positions, 2 per thread (rows r and r + 8); dy tile of the
// (16 + 2) x (32 + 2) source blocks = 36 x 68 high-res pixels in LDS.
constexpr int GH = 2 * (RH + 2), GW = 2 * (RW + 2);
template <typename TA>
__global__ __launch_bounds__(256) void upconv_dgrad_kernel(const TA* __restrict__ dy, const float* __restrict__ weff,
                                                           TA* __restrict__ dxl, int hl, int wl,
                                                           const TA* __restrict__ mask_y, int mask_act,
                                                           float mask_alpha) {
    __shared__ float4 gs[GH * GW];
    const int tid = threadIdx.x;
    const int rx = blockIdx.x * RW, ry = blockIdx.y * RH;
    const int H = 2 * hl, W = 2 * wl;
    const TA* gb = dy + (size_t)blockIdx.z * H * W * CH;
    if constexpr (sizeof(TA) == 4) {
        stage_batched<GH * GW, 256, 5, float4>(
            tid,
            [&](int i, bool& inside) {
                const int r = i / GW, c = i - r * GW;
                const int gy = 2 * (ry - 1) + r, gx = 2 * (rx - 1) + c;
                inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                return reinterpret_cast<const float4*>(gb + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * CH);
            },
            [&](int i, float4 v, bool inside) { gs[i] = inside ? v : make_float4(0.f, 0.f, 0.f, 0.f); });
    } else {
        for (int i = tid; i < GH * GW; i += 256) {
            const int r = i / GW, c = i - r * GW;
            const int gy = 2 * (ry - 1) + r, gx = 2 * (rx - 1) + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = ld4(gb + ((size_t)gy * W + gx) * CH);
            gs[i] = v;
        }
    }
    __syncthreads();
    const int c = tid & 31, r0 = tid >> 5;               // positions (r0, c) and (r0 + 8, c)
    float acc[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = 0.f;
#pragma unroll 1
    for (int m = 0; m < 9; ++m) {
        const float* wt = weff + m * 64;                 // [c][phase][o]: uniform -> scalar loads
        const int my = m / 3 - 1, mx = m % 3 - 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = r0 + 8 * p;
            // source block Q - m: tile rows 2(r - my + 1) + py, cols 2(c - mx + 1) + px
            const float4* g0 = gs + (2 * (r - my + 1)) * GW + 2 * (c - mx + 1);
            const float4 g[4] = {g0[0], g0[1], g0[GW], g0[GW + 1]};
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int phase = 0; phase < 4; ++phase) {
                    const float* wv = wt + ch * 16 + phase * 4;
                    acc[p][ch] += wv[0] * g[phase].x + wv[1] * g[phase].y + wv[2] * g[phase].z + wv[3] * g[phase].w;
                }
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int qy = ry + r0 + 8 * p, qx = rx + c;
        if (qy >= hl || qx >= wl) continue;
        const size_t off = (((size_t)blockIdx.z * hl + qy) * wl + qx) * CH;
        float4 out = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        if (mask_act != UOCR_ACT_NONE) {
            const float4 yv = ld4(mask_y + off);
            out.x *= act_grad_from_output<float>(yv.x, mask_act, mask_alpha);
            out.y *= act_grad_from_output<float>(yv.y, mask_act, mask_alpha);
            out.z *= act_grad_from_output<float>(yv.z, mask_act, mask_alpha);
            out.w *= act_grad_from_output<float>(yv.w, mask_act, mask_alpha);
        }
        st4(dxl + off, out);
    }
}

// ---------------------------------------------------------------------------------------------
// 1 -> 1 channel (Paragraph decoder): 36 multiply-adds per low-res position, everything on the vector ALU;
// the kernels are bound by their HBM bytes (low-res tensor + high-res tensor) and launch latency.
// weff1[m*4 + phase], built per thread from the 25 taps (uniform loads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void weff1_build(float (&we)[36], const float* __restrict__ w) {
#pragma unroll
    for (int m = 0; m < 9; ++m)
#pragma unroll
        for (int phase = 0; phase < 4; ++phase) {
            int ylo, yhi, xlo, xhi;
            tap_group(phase >> 1, m / 3, ylo, yhi);
            tap_group(phase & 1, m % 3, xlo, xhi);
            float s = 0.f;
            for (int ky = ylo; ky < yhi; ++ky)
                for (int kx = xlo; kx < xhi; ++kx) s += w[ky * 5 + kx];
            we[m * 4 + phase] = s;
        }
}

// block = 16 x 32 low-res positions, 2 per thread (rows r0 and r0 + 8)
template <typename TA>
__global__ __launch_bounds__(256) void up1_fwd_kernel(const TA* __restrict__ xl, const float* __restrict__ w,
                                                      const float* __restrict__ bias, TA* __restrict__ y, int hl,
                                                      int wl, int use_bias, int act, float alpha) {
    __shared__ float xs[XH * XW];
    const int tid = threadIdx.x;
    const int rx = blockIdx.x * RW, ry = blockIdx.y * RH;
    const TA* xb = xl + (size_t)blockIdx.z * hl * wl;
    stage_batched<XH * XW, 256, 3, TA>(
        tid,
        [&](int i, bool& inside) {
            const int r = i / XW, c = i - r * XW;
            const int gy = ry - 1 + r, gx = rx - 1 + c;
            inside = gy >= 0 && gy < hl && gx >= 0 && gx < wl;
            return xb + (size_t)min(max(gy, 0), hl - 1) * wl + min(max(gx, 0), wl - 1);
        },
        [&](int i, TA v, bool inside) { xs[i] = inside ? (float)v : 0.f; });
    float we[36];
    weff1_build(we, w);
    const float b0 = use_bias ? bias[0] : 0.f;
    __syncthreads();
    const int c = tid & 31, r0 = tid >> 5;
    const int W = 2 * wl;
    TA* yb = y + (size_t)blockIdx.z * (2 * hl) * W;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = r0 + 8 * p, py = ry + r, pxl = rx + c;
        float acc[4] = {b0, b0, b0, b0};
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const float xv = xs[(r + m / 3) * XW + c + m % 3];
#pragma unroll
            for (int phase = 0; phase < 4; ++phase) acc[phase] += xv * we[m * 4 + phase];
        }
        if (py < hl && pxl < wl) {
            TA* o = yb + (size_t)(2 * py) * W + 2 * pxl;
            st2(o, make_float2(act_apply(acc[0], act, alpha), act_apply(acc[1], act, alpha)));
            st2(o + W, make_float2(act_apply(acc[2], act, alpha), act_apply(acc[3], act, alpha)));
        }
    }
}

template <typename TA>
__global__ __launch_bounds__(256) void up1_dgrad_kernel(const TA* __restrict__ dy, const float* __restrict__ w,
                                                        TA* __restrict__ dxl, int hl, int wl,
                                                        const TA* __restrict__ mask_y, int mask_act,
                                                        float mask_alpha) {
    __shared__ float gs[GH * GW];
    const int tid = threadIdx.x;
    const int rx = blockIdx.x * RW, ry = blockIdx.y * RH;
    const int H = 2 * hl, W = 2 * wl;
    const TA* gb = dy + (size_t)blockIdx.z * H * W;
    stage_batched<GH * GW, 256, 10, TA>(
        tid,
        [&](int i, bool& inside) {
            const int r = i / GW, c = i - r * GW;
            const int gy = 2 * (ry - 1) + r, gx = 2 * (rx - 1) + c;
            inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            return gb + (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1);
        },
        [&](int i, TA v, bool inside) { gs[i] = inside ? (float)v : 0.f; });
    float we[36];
    weff1_build(we, w);
    __syncthreads();
    const int c = tid & 31, r0 = tid >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = r0 + 8 * p;
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < 9; ++m) {
            const int my = m / 3 - 1, mx = m % 3 - 1;
            const float* g0 = gs + (2 * (r - my + 1)) * GW + 2 * (c - mx + 1);   // source block Q - m
            acc += we[m * 4] * g0[0] + we[m * 4 + 1] * g0[1] + we[m * 4 + 2] * g0[GW] + we[m * 4 + 3] * g0[GW + 1];
        }
        const int qy = ry + r, qx = rx + c;
        if (qy >= hl || qx >= wl) continue;
        const size_t off = ((size_t)blockIdx.z * hl + qy) * wl + qx;
        if (mask_act != UOCR_ACT_NONE) acc *= act_grad_from_output<float>(ld1(mask_y + off), mask_act, mask_alpha);
        st1(dxl + off, acc);
    }
}

// partial[blk][37]: dWeff[m*4 + phase] and db
template <typename TA>
__global__ __launch_bounds__(256) void up1_wgrad_kernel(const TA* __restrict__ xl, const TA* __restrict__ dy,
                                                        float* __restrict__ partial, int hl, int wl,
                                                        int rows_per_block) {
    __shared__ float xs[XH * XW];
    __shared__ float red[4][37];
    const int tid = threadIdx.x;
    const int rx = blockIdx.x * RW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(hl, row_begin + rows_per_block);
    const TA* xb = xl + (size_t)blockIdx.z * hl * wl;
    const int W = 2 * wl;
    const TA* gb = dy + (size_t)blockIdx.z * (2 * hl) * W;
    const int c = tid & 31, r0 = tid >> 5;
    float acc[36], dbacc = 0.f;
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.f;
    for (int ry = row_begin; ry < row_end; ry += RH) {
        __syncthreads();
        stage_batched<XH * XW, 256, 3, TA>(
            tid,
            [&](int i, bool& inside) {
                const int r = i / XW, cc = i - r * XW;
                const int gy = ry - 1 + r, gx = rx - 1 + cc;
                inside = gy >= 0 && gy < hl && gx >= 0 && gx < wl;
                return xb + (size_t)min(max(gy, 0), hl - 1) * wl + min(max(gx, 0), wl - 1);
            },
            [&](int i, TA v, bool inside) { xs[i] = inside ? (float)v : 0.f; });
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = r0 + 8 * p, py = ry + r, pxl = rx + c;
            if (py >= row_end || pxl >= wl) continue;
            const TA* g0 = gb + (size_t)(2 * py) * W + 2 * pxl;
            const float2 ga = ld2(g0), gbv = ld2(g0 + W);
            const float g[4] = {ga.x, ga.y, gbv.x, gbv.y};
            dbacc += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                const float xv = xs[(r + m / 3) * XW + c + m % 3];
#pragma unroll
                for (int phase = 0; phase < 4; ++phase) acc[m * 4 + phase] += xv * g[phase];
            }
        }
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const float v = wave_reduce_sum(acc[i]);
        if (lane == 0) red[wv][i] = v;
    }
    dbacc = wave_reduce_sum(dbacc);
    if (lane == 0) red[wv][36] = dbacc;
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    up1_write_row(partial + (size_t)blk * UP1_NOUT, [&](int i) { return red[0][i] + red[1][i] + red[2][i] + red[3][i]; }, tid);
}


int up_rows_per_block(int strips, int hl, int n, unsigned max_blocks = 2048u) {
    int rows = RH;
    while (rows < hl && (size_t)strips * ((hl + rows - 1) / rows) * n > max_blocks) rows += RH;
    return rows;
}

int check_up(uocr_ctx* ctx, int dtype, int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw) {
    if (UOCR_DTYPE_BASE(dtype) != UOCR_F32 && UOCR_DTYPE_BASE(dtype) != UOCR_F16)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "upconv2x: float32 / float16 only");
    if (!((cin == CH && cout == CH) || (cin == 1 && cout == 1)) || kh != 5 || kw != 5 || ph != 2 || pw != 2)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED,
                  "upconv2x: 5x5 / padding 2 / 4 -> 4 or 1 -> 1 channels only (got %dx%d pad %d,%d %d -> %d)", kh, kw, ph,
                  pw, cin, cout);
    UOCR_REQUIRE(ctx, n > 0 && hl > 0 && wl > 0 && n <= 65535);
    return UOCR_OK;
}

}  // namespace

extern "C" int uocr_upconv2x_fwd(uocr_ctx* ctx, int dtype, const void* x_low, const void* w, const void* b, void* y,
                                 int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                                 int use_bias, int act, double act_alpha, void* weff) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x_low && w && b && y);
    int rc = check_up(ctx, dtype, n, hl, wl, cin, cout, kh, kw, ph, pw);
    if (rc != UOCR_OK) return rc;
    if (uocr_upconv_h16_eligible(ctx, dtype, cin, cout) && uocr_aligned_act(x_low, dtype) && uocr_aligned_act(y, dtype))
        return uocr_upconv_fwd_h16(ctx, x_low, w, b, y, n, hl, wl, use_bias, act, act_alpha);
    const int strips = (wl + RW - 1) / RW, rows = up_rows_per_block(strips, hl, n);
    UOCR_DISPATCH_TA(ctx, dtype, {
        if (cin == 1)
            hipLaunchKernelGGL((up1_fwd_kernel<TA>), dim3((wl + RW - 1) / RW, (hl + RH - 1) / RH, n), dim3(256), 0,
                               ctx->stream, (const TA*)x_low, (const float*)w, (const float*)b, (TA*)y, hl, wl, use_bias,
                               act, (float)act_alpha);
        else
            hipLaunchKernelGGL((upconv_fwd_kernel<TA>), dim3(strips, (hl + rows - 1) / rows, n), dim3(256), 0,
                               ctx->stream, (const TA*)x_low, (const float*)w, (const float*)b, (TA*)y, hl, wl, rows,
                               use_bias, act, (float)act_alpha, (float*)weff);
    });
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

extern "C" int uocr_upconv2x_bwd_data(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx_low, int n,
                                      int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                                      const void* x_act, int act, double act_alpha, const void* weff_in) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, dy && w && dx_low);
    UOCR_REQUIRE(ctx, act == UOCR_ACT_NONE || x_act != nullptr);
    int rc = check_up(ctx, dtype, n, hl, wl, cin, cout, kh, kw, ph, pw);
    if (rc != UOCR_OK) return rc;
    if (uocr_upconv_h16_eligible(ctx, dtype, cin, cout) && uocr_aligned_act(dy, dtype) &&
        uocr_aligned_act(dx_low, dtype) && (act == UOCR_ACT_NONE || uocr_aligned_act(x_act, dtype)))
        return uocr_upconv_dgrad_h16(ctx, dy, w, dx_low, n, hl, wl, act == UOCR_ACT_NONE ? nullptr : x_act, act, act_alpha);
    if (uocr_upconv_t32_eligible(ctx, dtype, cin, cout) && (reinterpret_cast<uintptr_t>(dy) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(dx_low) & 15u) == 0 && (act == UOCR_ACT_NONE || (reinterpret_cast<uintptr_t>(x_act) & 15u) == 0))
        return uocr_upconv_dgrad_t32(ctx, dy, w, dx_low, n, hl, wl, cin, act == UOCR_ACT_NONE ? nullptr : x_act, act, act_alpha);
    const float* weff = (const float*)weff_in;             // from this layer's forward call with the same w, or null
    if (cin != 1 && !weff) {
        rc = uocr_need_workspace(ctx, NWEFF * sizeof(float));
        if (rc != UOCR_OK) return rc;
        hipLaunchKernelGGL(upconv_weff_kernel, dim3(1), dim3(256), 0, ctx->stream, (const float*)w, (float*)ctx->workspace);
        weff = (const float*)ctx->workspace;
    }
    const dim3 grid((wl + RW - 1) / RW, (hl + RH - 1) / RH, n);
    UOCR_DISPATCH_TA(ctx, dtype, {
        if (cin == 1)
            hipLaunchKernelGGL((up1_dgrad_kernel<TA>), grid, dim3(256), 0, ctx->stream, (const TA*)dy, (const float*)w,
                               (TA*)dx_low, hl, wl, (const TA*)x_act, act, (float)act_alpha);
        else
            hipLaunchKernelGGL((upconv_dgrad_kernel<TA>), grid, dim3(256), 0, ctx->stream, (const TA*)dy,
                               weff, (TA*)dx_low, hl, wl, (const TA*)x_act, act, (float)act_alpha);
    });
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

extern "C" int uocr_upconv2x_bwd_weight(uocr_ctx* ctx, int dtype, const void* x_low, const void* dy, void* dw, void* db,
                                        int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                                        int use_bias, int accumulate) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x_low && dy && dw && db);
    int rc = check_up(ctx, dtype, n, hl, wl, cin, cout, kh, kw, ph, pw);
    if (rc != UOCR_OK) return rc;
    const float unscale = (float)uocr_grad_unscale(dtype);
    const int ncols = cin == 1 ? UP1_NOUT : UP4_NOUT, ndw = ncols - cin;
    // the block partials are rows of dw and db themselves (up_phase.h); their float64 column sums: recorded when a deferred
    // group is open (finish_group.h), else one coalesced launch
    auto finish = [&](const float* partial, int nblocks) -> int {
        FinishDesc fd{};
        fd.kind = FIN_COLS;
        fd.partial = partial;
        fd.nblocks = nblocks;
        fd.ncols = fd.group_cols = ncols;
        fd.row_stride = ncols;
        fd.dw = (float*)dw, fd.db = (float*)db;
        fd.use_bias = use_bias, fd.accumulate = accumulate;
        fd.unscale = unscale;
        fd.p[0] = ndw;
        if (uocr_finish_defer(ctx, fd)) return UOCR_OK;
        hipLaunchKernelGGL(colsum_finish_kernel, dim3((ncols + 7) / 8), dim3(256), 0, ctx->stream, partial, nblocks, ncols,
                           ndw, (float*)dw, (float*)db, use_bias, accumulate, unscale);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    };
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_fast && ctx->opt_h16 && uocr_aligned_act(x_low, dtype) &&
        uocr_aligned_act(dy, dtype)) {
        // binary16 MFMAs over channel / phase planes (conv_h16w.hip), same partial rows
        const size_t floats = (size_t)ctx->cu_count * 8 * ncols;
        float* partial = uocr_partial_buffer(ctx, floats * sizeof(float), &rc);
        if (rc != UOCR_OK) return rc;
        int nblocks = 0;
        rc = uocr_upconv_wgrad_h16(ctx, x_low, dy, partial, floats, n, hl, wl, cin, &nblocks);
        if (rc != UOCR_OK) return rc;
        return finish(partial, nblocks);
    }
    // fewer, longer blocks than the forward: every block ends with a reduction and a partial row for the finish kernel
    const int strips = (wl + RW - 1) / RW, rows = up_rows_per_block(strips, hl, n, 1024u);
    const int bands = (hl + rows - 1) / rows, nblocks = strips * bands * n;
    float* partial = uocr_partial_buffer(ctx, (size_t)nblocks * ncols * sizeof(float), &rc);
    if (rc != UOCR_OK) return rc;
    UOCR_DISPATCH_TA(ctx, dtype, {
        if (cin == 1)
            hipLaunchKernelGGL((up1_wgrad_kernel<TA>), dim3(strips, bands, n), dim3(256), 0, ctx->stream,
                               (const TA*)x_low, (const TA*)dy, partial, hl, wl, rows);
        else
            hipLaunchKernelGGL((upconv_wgrad_kernel<TA>), dim3(strips, bands, n), dim3(256), 0, ctx->stream,
                               (const TA*)x_low, (const TA*)dy, partial, hl, wl, rows);
    });
    UOCR_LAUNCH_CHECK(ctx);
    return finish(partial, nblocks);
}
