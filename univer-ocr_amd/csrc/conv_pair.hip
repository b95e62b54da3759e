// The Monochrome block as ONE forward and ONE backward kernel (float32):
//     x (1 ch) -> conv3x3 (1 -> 16, pad 1) -> LeakyReLU -> conv3x3 (16 -> 1, pad 1) -> [Sigmoid] -> y
// (reference: my_model/model.py:108-135 built from nn/layers/convolutional.py:62-145 and
// nn/layers/layers.py:377-418).
//
// Run layer by layer the 16-channel activation a1 (268 MB at 32x256x512) is written once and read three
// times per train step (conv_2 forward, conv_2 dx + LeakyReLU', conv_2 dw) and its gradient (another
// 268 MB) is written once and read once: 1.6 GB of the step's HBM traffic for 34 MB of real input and
// output.  a1 costs 9 FMAs per channel to recompute from the 1-channel input, so neither tensor is
// materialised here:
//   forward : a block recomputes a1 on its 16x32 tile (+1 halo) into LDS and applies conv_2 from LDS.
//   backward: per position q the lane recomputes a1[q], gathers g = dy * act2'(y) through the 3x3 window
//             ONCE for both  dw2[t,c] += a1[q,c] g[q-t+1]  and  d_a1[q,c] = lrelu'(a1) sum_t w2[t,c] g[q-t+1],
//             then  dw1[s,c] += x[q+s-1] d_a1[q,c],  db1 += d_a1,  db2 += g[q];
//             dx (optional) in scatter form: u[q,s] = sum_c w1[s,c] d_a1[q,c] goes to LDS (9 floats per
//             position instead of 16 channels x 9 reads), dx[p] = sum_s u[p-s+1, s].
// Lane layout as in the c16 kernels of conv_fast.hip: 4 adjacent lanes share a position, each owns 4 of
// the 16 channels (weights in VGPRs, partial sums combined with two quad shuffles).
// HBM traffic per image pixel: forward 8 B, backward 12-16 B; the kernels are FMA-bound
// (forward ~320, backward ~900 lane-FMAs per pixel).
//
// hipcc-flags: -fno-slp-vectorize
// (gfx950 SIMDs are 32 lanes wide: v_fma_f32 issues in 2 cycles and v_pk_fma_f32 is no faster than the two
// FMAs it replaces, while the SLP vectoriser's packing adds v_pk_mov / v_mov traffic to feed it)
#include "uocr_common.h"

namespace {

constexpr int TH = 16, TW = 30;            // tile of positions owned by a block iteration
constexpr int XH = TH + 4, XW = TW + 4;    // x / g tiles in LDS: halo 2
constexpr int AH = TH + 2, AW = TW + 2;    // a1 / u region: halo 1 -- 18 x 32: one column per quad of a half block
constexpr int C = 16;
constexpr int NA = 36 + 36 + 4 + 1;        // per-lane accumulators of the backward: dw1, dw2, db1, db2
constexpr int NPF = (XH * XW + 255) / 256; // x / g tile elements staged per thread

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over the 4 lanes of a quad with two DPP quad_perm moves (VALU only; __shfl_xor would be two
// dependent ds_bpermute round trips through LDS)
__device__ __forceinline__ float quad_sum(float v) {
    v += dpp_move<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);      // quad_perm [2,3,0,1]
    return v;
}

// sum over the 4 lanes {l, l+4, l+8, l+12} of a 16-lane row (same channel quad q = l & 3)
__device__ __forceinline__ float row_sum_q(float v) {
    v += dpp_move<0x124>(v);     // row_ror:4
    v += dpp_move<0x128>(v);     // row_ror:8
    return v;
}

__device__ __forceinline__ float out_act(float v, int act) {
    return act == UOCR_ACT_SIGMOID ? 1.f / (1.f + expf(-v)) : v;
}

// 4 channels [q*4, q*4+4) of every tap of a (3,3,1,16) or (3,3,16,1) weight tensor: flat [tap*16 + ch]
__device__ __forceinline__ void load_taps(float (&dst)[9][4], const float* __restrict__ w, int q) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(w + t * C + q * 4);
        dst[t][0] = v.x;
        dst[t][1] = v.y;
        dst[t][2] = v.z;
        dst[t][3] = v.w;
    }
}

// the x (or dy, y) values of the XH x XW tile at (y0-2, x0-2) this thread stages: clamped loads, `in` mask
struct Stage {
    size_t off[NPF];
    bool in[NPF];
    __device__ __forceinline__ void locate(int tid, int y0, int x0, int h, int wd) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + k * 256;
            const int r = i / XW, c = i - r * XW;
            const int gy = y0 - 2 + r, gx = x0 - 2 + c;
            in[k] = i < XH * XW && gy >= 0 && gy < h && gx >= 0 && gx < wd;
            off[k] = (size_t)min(max(gy, 0), h - 1) * wd + min(max(gx, 0), wd - 1);
        }
    }
};

// Block = column strip of TW outputs x rows [band*rows_per_block, +rows_per_block) of image blockIdx.z,
// walked tile by tile; the next tile's x is in flight (registers) while the current one is computed.
__global__ __launch_bounds__(256) void conv_pair_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ y,
                                                            int h, int wd, int rows_per_block, float pad1,
                                                            int use_b1, int use_b2, float alpha, int act2) {
    __shared__ float xs[XH * XW];
    __shared__ float4 a1s[AH * AW * 4];
    __shared__ float ws[2][9 * C];                       // both weight tensors: a phase keeps only its own in VGPRs
    const int tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
    const int col = quad & 31, half = quad >> 5;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const float* xb = x + (size_t)blockIdx.z * h * wd;
    float* yb = y + (size_t)blockIdx.z * h * wd;
    for (int i = tid; i < 2 * 9 * C; i += 256) ws[i / (9 * C)][i % (9 * C)] = (i < 9 * C ? w1 : w2 - 9 * C)[i];
    float wr[9][4], br[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) br[j] = use_b1 ? b1[q * 4 + j] : 0.f;
    const float bias2 = use_b2 ? b2[0] : 0.f;

    Stage st;
    float px[NPF];
    st.locate(tid, row_begin, x0, h, wd);
#pragma unroll
    for (int k = 0; k < NPF; ++k) px[k] = xb[st.off[k]];
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) xs[tid + k * 256] = st.in[k] ? px[k] : pad1;
        __syncthreads();
        if (y0 + TH < row_end) {                         // next tile: loads overlap this tile's math
            st.locate(tid, y0 + TH, x0, h, wd);
#pragma unroll
            for (int k = 0; k < NPF; ++k) px[k] = xb[st.off[k]];
        }
        // a1 = LeakyReLU(conv_1(x)) on the 18 x 32 region (tile + halo 1): a quad walks 9 rows of one
        // column with a sliding 3x3 window; outside the image a1 is conv_2's zero padding
        {
            load_taps(wr, ws[0], q);
            const int r0 = half * 9;
            float xw[3][3], nx[3];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                xw[1][sx] = xs[r0 * XW + col + sx];
                xw[2][sx] = xs[(r0 + 1) * XW + col + sx];
                nx[sx] = xs[(r0 + 2) * XW + col + sx];
            }
#pragma unroll 1
            for (int k = 0; k < 9; ++k) {
                const int r = r0 + k;
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    xw[0][sx] = xw[1][sx];
                    xw[1][sx] = xw[2][sx];
                    xw[2][sx] = nx[sx];
                    nx[sx] = xs[min(r + 3, XH - 1) * XW + col + sx];   // next step's row, in flight during the math
                }
                float v[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += xw[t / 3][t % 3] * wr[t][j];
                const int ay = y0 - 1 + r, ax = x0 - 1 + col;
                const bool inside = ay >= 0 && ay < h && ax >= 0 && ax < wd;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = inside ? (v[j] >= 0.f ? v[j] : alpha * v[j]) : 0.f;
                a1s[(r * AW + col) * 4 + q] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        // conv_2: a quad owns 8 vertically adjacent outputs of one column and walks the 10 a1 rows they
        // need once (3 float4 per row from LDS); an a1 row feeds tap row 0 of the output starting there,
        // tap row 1 of the one above and completes the one two above (rolled loop: few live registers)
        {
            load_taps(wr, ws[1], q);
            const int pc = col < TW ? col : 0, pr0 = half * 8;
            float above2 = 0.f, above1 = 0.f;           // partial sums of outputs rr - 2 and rr - 1
#pragma unroll 1
            for (int rr = 0; rr < 10; ++rr) {
                float t[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const float4 v = a1s[((pr0 + rr) * AW + pc + tx) * 4 + q];
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
                        t[ty] += v.x * wr[ty * 3 + tx][0] + v.y * wr[ty * 3 + tx][1] + v.z * wr[ty * 3 + tx][2] +
                                 v.w * wr[ty * 3 + tx][3];
                }
                const float done = quad_sum(above2 + t[2]);
                above2 = above1 + t[1];
                above1 = t[0];
                const int gy = y0 + pr0 + rr - 2, gx = x0 + col;
                if (rr >= 2 && q == 0 && col < TW && gy < h && gx < wd)
                    yb[(size_t)gy * wd + gx] = out_act(done + bias2, act2);
            }
        }
    }
}

// Same walk for the backward; the 77 accumulators stay in registers over all tiles of the block and are
// reduced once: DPP over the 4 positions of a 16-lane row, LDS over the 16 rows.  partial[blk][q][NA]
template <bool DX>
__global__ __launch_bounds__(256) void conv_pair_bwd_kernel(const float* __restrict__ x, const float* __restrict__ yout,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ w2,
                                                            float* __restrict__ partial, float* __restrict__ dx,
                                                            int h, int wd, int rows_per_block, float pad1,
                                                            int use_b1, float alpha, int act2) {
    constexpr int OFF = DX ? 0 : 1;                      // region origin - (xs origin + 1)
    constexpr int RW = DX ? AW : TW;                     // region width: halo 1 only when dx is wanted
    constexpr int RPQ = DX ? 9 : 8;                      // positions (rows) per quad: 2 halves x RPQ rows
    __shared__ float xs[XH * XW];
    __shared__ float gs[XH * XW];
    __shared__ float us[DX ? AH * AW * 9 : 1];
    __shared__ float red[16][4][NA];
    __shared__ float w2s[9 * C];                         // conv_2 weights: read per tap (frees 36 VGPRs)
    const int tid = threadIdx.x, q = tid & 3, quad = tid >> 2;
    const int col = quad & 31, half = quad >> 5;
    const bool lane_on = col < RW;
    const int c = lane_on ? col : 0;
    const int x0 = blockIdx.x * TW;
    const int row_begin = blockIdx.y * rows_per_block, row_end = min(h, row_begin + rows_per_block);
    const size_t img = (size_t)blockIdx.z * h * wd;
    const float *xb = x + img, *gb = dy + img, *yb = yout + img;

    float w1r[9][4], b1r[4];
    load_taps(w1r, w1, q);
    if (tid < 9 * C) w2s[tid] = w2[tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) b1r[j] = use_b1 ? b1[q * 4 + j] : 0.f;
    float dw1[9][4], dw2[9][4], db1[4], db2 = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) dw1[t][j] = dw2[t][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) db1[j] = 0.f;

    Stage st;
    float px[NPF], pg[NPF];
    auto prefetch = [&](int y0) {
        st.locate(tid, y0, x0, h, wd);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            px[k] = xb[st.off[k]];
            pg[k] = gb[st.off[k]];
            if (act2 == UOCR_ACT_SIGMOID) {
                const float yv = yb[st.off[k]];
                pg[k] *= yv * (1.f - yv);
            }
        }
    };
    prefetch(row_begin);
    for (int y0 = row_begin; y0 < row_end; y0 += TH) {
        __syncthreads();                                 // the previous tile's LDS reads are over
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (tid + k * 256 < XH * XW) {
                xs[tid + k * 256] = st.in[k] ? px[k] : pad1;
                gs[tid + k * 256] = st.in[k] ? pg[k] : 0.f;
            }
        __syncthreads();
        if (y0 + TH < row_end) prefetch(y0 + TH);
        // a quad walks RPQ rows of one region column; 3x3 windows of x and g slide down in registers
        const int r0 = half * RPQ;
        float xw[3][3], gw[3][3], nx[3], ng[3];
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
            const int o = c + OFF + sx;
            xw[1][sx] = xs[(r0 + OFF) * XW + o];
            gw[1][sx] = gs[(r0 + OFF) * XW + o];
            xw[2][sx] = xs[(r0 + OFF + 1) * XW + o];
            gw[2][sx] = gs[(r0 + OFF + 1) * XW + o];
            nx[sx] = xs[(r0 + OFF + 2) * XW + o];
            ng[sx] = gs[(r0 + OFF + 2) * XW + o];
        }
#pragma unroll 1
        for (int k = 0; k < RPQ; ++k) {
            const int r = r0 + k;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const int o = min(r + OFF + 3, XH - 1) * XW + c + OFF + sx;
                xw[0][sx] = xw[1][sx];
                xw[1][sx] = xw[2][sx];
                xw[2][sx] = nx[sx];
                gw[0][sx] = gw[1][sx];
                gw[1][sx] = gw[2][sx];
                gw[2][sx] = ng[sx];
                nx[sx] = xs[o];                          // next step's row, in flight during the math
                ng[sx] = gs[o];
            }
            const int ay = y0 - (DX ? 1 : 0) + r, ax = x0 - (DX ? 1 : 0) + c;
            const bool inside = lane_on && ay >= 0 && ay < h && ax >= 0 && ax < wd;
            const bool owned = inside && (!DX || (r >= 1 && r <= TH && c >= 1 && c <= TW));
            // a1 of this position (pre-activation z), recomputed
            float z[4] = {b1r[0], b1r[1], b1r[2], b1r[3]};
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] += xw[t / 3][t % 3] * w1r[t][j];
            float a[4], slope[4], s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                slope[j] = z[j] >= 0.f ? 1.f : alpha;
                a[j] = owned ? z[j] * slope[j] : 0.f;
                s[j] = 0.f;
            }
            // one pass over the 3x3 window of g for dw2 and for conv_2's dx: tap t pairs with g[q - t + 1]
            int wofs = q * 4;
            asm volatile("" : "+v"(wofs));               // keeps the w2 reads inside the loop (36 VGPRs otherwise)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float g = gw[2 - t / 3][2 - t % 3];
                const float4 wt = *reinterpret_cast<const float4*>(w2s + t * C + wofs);
                const float w2t[4] = {wt.x, wt.y, wt.z, wt.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dw2[t][j] += a[j] * g;
                    s[j] += w2t[j] * g;
                }
            }
            float d[4], dn[4];                           // d_a1 (also on the halo, for dx); dn: owned only
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[j] = inside ? s[j] * slope[j] : 0.f;
                dn[j] = owned ? d[j] : 0.f;
                db1[j] += dn[j];
            }
            db2 += owned ? gw[1][1] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) dw1[t][j] += xw[t / 3][t % 3] * dn[j];
            if constexpr (DX) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float u = w1r[t][0] * d[0] + w1r[t][1] * d[1] + w1r[t][2] * d[2] + w1r[t][3] * d[3];
                    u = quad_sum(u);
                    if ((t & 3) == q) us[(r * AW + col) * 9 + t] = u;
                }
            }
        }
        if constexpr (DX) {
            __syncthreads();
            for (int p = tid; p < TH * TW; p += 256) {
                const int pr = p / TW, pc = p - pr * TW;
                const int gy = y0 + pr, gx = x0 + pc;
                float v = 0.f;
#pragma unroll
                for (int sy = 0; sy < 3; ++sy)
#pragma unroll
                    for (int sx = 0; sx < 3; ++sx)
                        v += us[((pr + 2 - sy) * AW + pc + 2 - sx) * 9 + sy * 3 + sx];
                if (gy < h && gx < wd) dx[img + (size_t)gy * wd + gx] = v;
            }
        }
    }
    // reduction: 4 positions of a 16-lane row by DPP, the 16 rows of the block through LDS
    const int row = tid >> 4;
    const bool writer = (tid & 15) < 4;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v1 = row_sum_q(dw1[t][j]), v2 = row_sum_q(dw2[t][j]);
            if (writer) {
                red[row][q][t * 4 + j] = v1;
                red[row][q][36 + t * 4 + j] = v2;
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = row_sum_q(db1[j]);
        if (writer) red[row][q][72 + j] = v;
    }
    db2 = row_sum_q(db2);
    if (writer) red[row][q][76] = db2;
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* out = partial + (size_t)blk * 4 * NA;
    for (int i = tid; i < 4 * NA; i += 256) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][0][i];
        out[i] = v;
    }
}

// block (k, q): float64 sum of the block partials -> dw1 / dw2 [tap*16 + q*4 + j], db1[q*4 + j], db2
__global__ __launch_bounds__(256) void conv_pair_bwd_finish(const float* __restrict__ partial, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2, int nblocks, int use_b1,
                                                            int use_b2, int accumulate) {
    __shared__ double smem[16];
    const int k = blockIdx.x, q = blockIdx.y;
    double s = 0.0;
    for (int blk = threadIdx.x; blk < nblocks; blk += blockDim.x) s += (double)partial[((size_t)blk * 4 + q) * NA + k];
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
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// rows per block = a multiple of the tile height giving at most ~8 blocks per CU
int pair_rows_per_block(int strips, int h, int n) {
    int rows = TH;
    while (rows < h && (size_t)strips * ((h + rows - 1) / rows) * n > 2048u) rows += TH;
    return rows;
}

int check_pair(uocr_ctx* ctx, int dtype, int n, int h, int w, int cmid, int act2) {
    if (dtype != UOCR_F32) UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: float32 only");
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
    const int strips = (w + TW - 1) / TW;
    const int rows_per_block = pair_rows_per_block(strips, h, n);
    const dim3 grid(strips, (h + rows_per_block - 1) / rows_per_block, n);
    hipLaunchKernelGGL(conv_pair_fwd_kernel, grid, dim3(256), 0, ctx->stream, (const float*)x, (const float*)w1,
                       (const float*)b1, (const float*)w2, (const float*)b2, (float*)y, h, w, rows_per_block,
                       (float)pad_value1, use_bias1, use_bias2, (float)alpha1, act2);
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
    const int strips = (w + TW - 1) / TW;
    const int rows_per_block = pair_rows_per_block(strips, h, n);
    const int bands = (h + rows_per_block - 1) / rows_per_block;
    const int nblocks = strips * bands * n;
    rc = uocr_need_workspace(ctx, (size_t)nblocks * 4 * NA * sizeof(float));
    if (rc != UOCR_OK) return rc;
    float* partial = (float*)ctx->workspace;
    const dim3 grid(strips, bands, n);
    if (dx)
        hipLaunchKernelGGL((conv_pair_bwd_kernel<true>), grid, dim3(256), 0, ctx->stream, (const float*)x,
                           (const float*)y, (const float*)dy, (const float*)w1, (const float*)b1, (const float*)w2,
                           partial, (float*)dx, h, w, rows_per_block, (float)pad_value1, use_bias1, (float)alpha1,
                           act2);
    else
        hipLaunchKernelGGL((conv_pair_bwd_kernel<false>), grid, dim3(256), 0, ctx->stream, (const float*)x,
                           (const float*)y, (const float*)dy, (const float*)w1, (const float*)b1, (const float*)w2,
                           partial, (float*)nullptr, h, w, rows_per_block, (float)pad_value1, use_bias1,
                           (float)alpha1, act2);
    UOCR_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(conv_pair_bwd_finish, dim3(NA, 4), dim3(256), 0, ctx->stream, (const float*)partial,
                       (float*)dw1, (float*)db1, (float*)dw2, (float*)db2, nblocks, use_bias1, use_bias2, accumulate);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}
