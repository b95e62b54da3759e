// Small-channel convolutions of the page nets for binary16 storage (UOCR_F16) on v_mfma_f32_16x16x16_f16
// (reference layers: nn/layers/convolutional.py:62-145, nn/layers/upsample.py:21-39 as chained by
// my_model/model.py:194-247).
//
// With 1-4 channels a 5x5 convolution is vector-ALU bound (200-400 FMAs per pixel for 12-16 bytes) and as an
// im2col GEMM it wastes the matrix cores: 2 or 4 result columns out of 16.  Here the 16 result rows of an MFMA
// are (vertical shift dy, output channel): a VERTICAL TOEPLITZ operand.  For an output block of 16 columns x DY
// rows (DY = 16 / channels) the window is the (DY-1)*S + KH input rows the block touches; row m = (dy, co) of
// the weight operand holds w[ty' - dy*S][tx][ci][co] (zero outside the kernel), so ONE chain of MFMAs over the
// window produces all DY rows and every lane of the result is a real output:
//       D[(dy, co), col] = sum_{ty', tx, ci} Wt[(dy, co), (ty', tx, ci)] * X[row0*S + ty', col*S + tx, ci]
// The K index of an MFMA is (k-group kq = lane / 16, j = 0..3) = 4 consecutive binary16 values of one window
// row: with channels-last storage that is one pixel (C = 4), two pixels (C = 2) -- a single 8-byte LDS read of
// the staged input tile, no packing instructions.  k-group kq of MFMA (ib, q) is window row 4*ib + kq, halves
// 4q..4q+3 of the row segment that starts at the output column: the LDS address is lane base + immediate.
// Result layout D[m = 4*kq + i][n = column]: lane (column, kq) holds 4 values that are contiguous in memory
// (the channels of a pixel, or two 2-channel pixels rows apart): 8- / 4-byte stores, 16 lanes = 128 / 64 bytes.
//
// MFMAs per output pixel (executed / 16 cycles each): 5x5 4->2: 15 per 128 px; 5x5 2->4 (its backward-data): 6
// per 64; upsample+5x5 backward-data as a stride-2 6x6 window over dy (4->4): 18 per 64 low-res = 256 high-res
// pixels.  The float32 master weights are rounded to binary16 as operands; accumulation is float32.
// A zero weight still multiplies what the tile holds there: inputs must be finite.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
#include <type_traits>

#include "conv_toeplitz.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ f32x4 mfma16(f16x4 a, f16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= 0.f ? 1.f : 0.f);
        case UOCR_ACT_LEAKY: return v * ((v >= 0.f ? 1.f : 0.f) + alpha * (v < 0.f ? 1.f : 0.f));
        case UOCR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// smallest row stride (halves, a multiple of 4) >= need whose dword count is == target (mod mod)
constexpr int pick_stride(int need, int mod, int target) {
    int rs = round_up(need, 4);
    while ((rs / 2) % mod != target) rs += 4;
    return rs;
}

template <int C_, int COUT_, int KH_, int KW_, int S_, int MODE_, int DY_ = 0>
struct Geo {
    static constexpr int C = C_, COUT = COUT_, KH = KH_, KW = KW_, S = S_, MODE = MODE_;
    static constexpr int U = (MODE == M_UPFWD || MODE == M_S2DGRAD) ? 2 : 1;   // output pixels per position and axis
    static constexpr int NCO = U * U * COUT;                   // result rows per position: (phase, co)
    static constexpr int DY = DY_ ? DY_ : 16 / NCO;            // position rows of one MFMA chain (fewer: zero rows)
    static constexpr int ROWS = (DY - 1) * S + KH;             // window rows of one chain
    static constexpr int IB = (ROWS + 3) / 4;                  // row quads
    static constexpr int Q = (KW * C + 3) / 4;                 // 4-half chunks per window row
    static constexpr int NM = IB * Q;                          // MFMAs per chain
    static constexpr int BC = 64;                              // output columns of a block tile (4 chains)
    static constexpr int BRT = S == 1 ? 32 : 16;
    static constexpr int RPW = BRT / (4 * DY) > 0 ? BRT / (4 * DY) : 1;  // chains (row bands) per wave
    static constexpr int BR = 4 * RPW * DY;                    // output rows of a block tile
    static constexpr int IH = (BR - 1) * S + KH;               // staged input rows
    static constexpr int IHA = (BR - DY) * S + 4 * IB;         // rows a chain may address (the rest stays zero)
    // one channel, stride 1: a chunk starts at any pixel, i.e. at any 2-byte offset -- the tile holds PAIR WORDS
    // (word[c] = (x[c], x[c+1])), a chunk is the words c and c + 2 (one ds_read2_b32)
    static constexpr bool PAIRW = C == 1 && S == 1;
    static constexpr int XS = PAIRW ? 2 : C;                   // halves per pixel in the tile
    static constexpr int QS = PAIRW ? 8 : 4;                   // halves from one chunk to the next
    static constexpr int PXU = 8 / XS;                         // pixels per 16-byte staging unit
    static constexpr int UW = ((BC - 1) * S * XS + Q * QS + 7) / 8;      // staging units per tile row
    // row stride in halves.  A chain's 8-byte read takes 2 LDS passes at best (64 lanes x 8 bytes); the rows kq and
    // kq + 1 of a half wave must then fall on complementary banks: lanes are 2 dwords apart (C = 4), 1 (C = 2 and
    // pair words), 4 (stride 2)
    static constexpr int RS = (S == 2 && C > 1) ? pick_stride(UW * 8, 4, 2) : pick_stride(UW * 8, 64, C == 4 ? 32 : 16);
    static constexpr int RPP = 256 / UW;                       // tile rows staged per pass of the block
    static constexpr int NPASS = (IH + RPP - 1) / RPP;
    static constexpr int NW = (MODE == M_UPDGRAD || MODE == M_UPFWD || MODE == M_S2DGRAD ? 25 : KH * KW) * C * COUT;
    static_assert(RPW >= 1 && 16 % NCO == 0 && DY * NCO <= 16 && (C == 1 || C == 2 || C == 4) && UW <= 256, "unsupported geometry");
};

template <class G>
__device__ __forceinline__ f16x4 read_chunk(const _Float16* p) {
    if constexpr (G::C == 4) {
        return *reinterpret_cast<const f16x4*>(p);                       // 8-byte aligned: one pixel
    } else {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);        // 4-byte aligned
        const u32x2 v = {q[0], q[G::PAIRW ? 2 : 1]};                     // pair words c, c + 2 / two adjacent dwords
        return __builtin_bit_cast(f16x4, v);
    }
}

__device__ __forceinline__ float fast_act(float v, int act, float alpha) {
    // binary16 results: v_exp_f32 / v_rcp_f32 (1 ulp of float32) are exact enough by a factor of 2^12
    if (act == UOCR_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if (act == UOCR_ACT_LEAKY) return v >= 0.f ? v : alpha * v;
    if (act == UOCR_ACT_RELU) return v >= 0.f ? v : 0.f;
    return v;
}

// Persistent blocks over a flat tile index (image, tile row, column strip): block b takes tiles b, b + grid, ...
// (all tiles cost the same, so the chip is balanced to within one tile).  A tile is BR x BC outputs; wave w owns
// its row bands w*RPW .. w*RPW + RPW - 1 (DY rows each), four chains of 16 columns per band.
//   in      [n][h_in][w_in][C]     binary16
//   out     [n][h_out][w_out][COUT] binary16;  mask_y (backward-data: the activation OUTPUT of the layer below,
//           same shape as out) multiplies the result by act'(y)
template <class G>
__global__ __launch_bounds__(256) void conv_h16_kernel(const _Float16* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, _Float16* __restrict__ out,
                                                       const _Float16* __restrict__ mask_y, int h_in, int w_in,
                                                       int h_out, int w_out, int ph, int pw, int tiles_x, int tiles_y,
                                                       int ntiles, float pad, int use_bias, int act, float alpha,
                                                       int mask_act, float mask_alpha) {
    constexpr int C = G::C, COUT = G::COUT, S = G::S, RS = G::RS;
    __shared__ __attribute__((aligned(16))) _Float16 tile[G::IHA * RS];
    __shared__ float wl[G::NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;

    for (int i = tid; i < G::NW; i += 256) wl[i] = w[i];
    __syncthreads();

    // weight operand: row m = lane % 16 = (dy, co), k-group kq = window row 4*ib + kq, j = half 4q + j of the row.
    // The table [NM][64 lanes] is the same for the four waves: each wave builds a quarter of it (in the tile
    // buffer, not yet in use), then every lane picks up its NM entries -- the prologue is what a block pays once,
    // and the grid is 4 x the resident blocks
    static_assert(G::NM * 64 * 4 <= G::IHA * RS, "the weight table is built in the tile buffer");
    f16x4* wtab = reinterpret_cast<f16x4*>(tile);
    {
        const int m = n, dyi = m / G::NCO, co = m % G::NCO;   // co = (phase, channel) in the depth-to-space modes
#pragma unroll
        for (int i0 = 0; i0 < G::NM; i0 += 4) {
            const int i = i0 + wv;
            if (i < G::NM) {
                const int ib = i / G::Q, q = i - ib * G::Q;
                f16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ty = 4 * ib + kq - dyi * S, hh = 4 * q + j, tx = hh / C, ci = hh % C;
                    const bool live = dyi < G::DY && ty >= 0 && ty < G::KH && tx < G::KW;
                    v[j] = live ? (_Float16)weight_of<G>(wl, min(max(ty, 0), G::KH - 1), min(tx, G::KW - 1), ci, co)
                                : (_Float16)0.f;
                }
                wtab[i * 64 + lane] = v;
            }
        }
    }
    __syncthreads();
    f16x4 wa[G::NM];
#pragma unroll
    for (int i = 0; i < G::NM; ++i) wa[i] = wtab[i * 64 + lane];
    __syncthreads();
    // the rows / halves no load ever writes must read as zero (their weights are zero, 0 * garbage is not): the
    // columns behind the staged units of every row, and the rows behind the staged ones
    {
        constexpr int PADC = (RS - G::UW * 8) / 4;                      // 8-byte words per row behind the units
        for (int i = tid; i < G::IHA * PADC; i += 256)
            reinterpret_cast<uint64_t*>(tile + (i / (PADC > 0 ? PADC : 1)) * RS + G::UW * 8)[i % (PADC > 0 ? PADC : 1)] = 0;
        for (int i = tid; i < (G::IHA - G::IH) * (G::UW * 2); i += 256)
            reinterpret_cast<uint64_t*>(tile + (G::IH + i / (G::UW * 2)) * RS)[i % (G::UW * 2)] = 0;
    }
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = use_bias ? bias[(4 * kq + i) % COUT] : 0.f;   // (row % NCO) % COUT
    const uint32_t padw = __builtin_bit_cast(uint32_t, f16x2{(_Float16)pad, (_Float16)pad});
    // staging role of this thread: 16-byte unit su of tile rows sr, sr + RPP, ...
    const int sr = tid / G::UW, su = tid - sr * G::UW;
    const bool stager = sr < G::RPP;

    TileWalk walk(blockIdx.x, gridDim.x, tiles_x, tiles_y);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, walk.next(tiles_x, tiles_y)) {
        const int strip = walk.strip, trow = walk.trow, img = walk.img;
        const int c_begin = strip * G::BC, r0 = trow * G::BR;
        const _Float16* inb = in + (size_t)img * h_in * w_in * C;
        const size_t out_img = (size_t)img * h_out * w_out * COUT;
        __syncthreads();                                 // the previous tile's reads are over (and the zero fill)
        // ---- stage the input tile: pixel (iy0 + r, ix0 + c) -> tile[r][c*C ..], padding outside the image.
        // The column part of a thread's address and its in-image bits are the same for all its rows.
        if (stager) {
            const int iy0 = r0 * S - ph, gx0 = c_begin * S - pw + su * G::PXU;
            constexpr int NPX = G::PXU + (G::PAIRW ? 1 : 0);          // pair words need the pixel after the unit
            bool in_px[NPX];
            bool all_in = true, any_in = false;
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
                in_px[k] = (unsigned)(gx0 + k) < (unsigned)w_in;
                all_in = all_in && in_px[k];
                any_in = any_in || in_px[k];
            }
            const int col_off = min(max(gx0, 0), w_in - 1) * C;
            uint4 v[G::NPASS];
#pragma unroll
            for (int k = 0; k < G::NPASS; ++k) {
                const int gy = iy0 + sr + k * G::RPP;
                const bool row_ok = (unsigned)gy < (unsigned)h_in;
                const _Float16* src = inb + (size_t)min(max(gy, 0), h_in - 1) * w_in * C;
                v[k] = uint4{padw, padw, padw, padw};
                if constexpr (G::PAIRW) {
                    // pixels gx0 .. gx0 + 4 -> the words (p0,p1) (p1,p2) (p2,p3) (p3,p4)
                    uint32_t px[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) px[q] = padw & 0xFFFFu;
                    if (row_ok && all_in) {
                        const uint2 d = *reinterpret_cast<const uint2*>(src + col_off);   // (2-byte aligned at least)
                        px[0] = d.x & 0xFFFFu, px[1] = d.x >> 16, px[2] = d.y & 0xFFFFu, px[3] = d.y >> 16;
                        px[4] = __builtin_bit_cast(unsigned short, src[col_off + 4]);
                    } else if (row_ok && any_in) {
#pragma unroll
                        for (int q = 0; q < 5; ++q)
                            if (in_px[q]) px[q] = __builtin_bit_cast(unsigned short, src[gx0 + q]);
                    }
                    v[k] = uint4{px[0] | (px[1] << 16), px[1] | (px[2] << 16), px[2] | (px[3] << 16), px[3] | (px[4] << 16)};
                } else if (row_ok && all_in) {
                    v[k] = *reinterpret_cast<const uint4*>(src + col_off);      // (4-byte aligned at least; C = 1: 2)
                } else if (row_ok && any_in) {                                   // the image edge cuts the unit
                    if constexpr (C == 1) {
                        unsigned short* vh = reinterpret_cast<unsigned short*>(&v[k]);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (in_px[q]) vh[q] = __builtin_bit_cast(unsigned short, src[gx0 + q]);
                    } else {
                        uint32_t* vw = reinterpret_cast<uint32_t*>(&v[k]);
#pragma unroll
                        for (int px = 0; px < G::PXU; ++px)
                            if (in_px[px]) {
                                const uint32_t* sp = reinterpret_cast<const uint32_t*>(src + (size_t)(gx0 + px) * C);
#pragma unroll
                                for (int d = 0; d < C / 2; ++d) vw[px * (C / 2) + d] = sp[d];
                            }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < G::NPASS; ++k) {
                const int r = sr + k * G::RPP;
                if (r < G::IH) {
                    uint64_t* dst = reinterpret_cast<uint64_t*>(tile + r * RS + su * 8);
                    dst[0] = (uint64_t)v[k].x | ((uint64_t)v[k].y << 32);
                    dst[1] = (uint64_t)v[k].z | ((uint64_t)v[k].w << 32);
                }
            }
        }
        __syncthreads();
        // ---- chains
#pragma unroll
        for (int s = 0; s < G::RPW; ++s) {
            const int rb = (wv * G::RPW + s) * G::DY;    // first output row of the band, relative to the tile
            // backward-data (4 channels, plain store): the mask of the band's four chains is requested before their
            // MFMAs (the epilogue otherwise waits out one global load per chain)
            constexpr bool PREMASK = COUT == 4 && G::U == 1;
            uocr_h4 mraw[PREMASK ? 4 : 1];
            if constexpr (PREMASK) {
                if (mask_act != UOCR_ACT_NONE) {
                    const int row = min(r0 + rb + kq, h_out - 1);
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg) {
                        const int col = min(c_begin + cg * 16 + n, w_out - 1);
                        mraw[cg] = *reinterpret_cast<const uocr_h4*>(mask_y + out_img + ((size_t)row * w_out + col) * 4);
                    }
                }
            }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                const _Float16* base = tile + (rb * S + kq) * RS + (cg * 16 + n) * S * G::XS;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ib = 0; ib < G::IB; ++ib)
#pragma unroll
                    for (int q = 0; q < G::Q; ++q)
                    {
                        // one output channel, plain store: operands swapped, D[column, row shift] -- a lane then holds 4
                        // consecutive columns of one row (an 8-byte store) instead of 4 rows of one column (2-byte stores)
                        const f16x4 chunk = read_chunk<G>(base + ib * 4 * RS + G::QS * q);
                        acc = (COUT == 1 && G::U == 1) ? mfma16(chunk, wa[ib * G::Q + q], acc) : mfma16(wa[ib * G::Q + q], chunk, acc);
                    }
                // lane (column n, kq): rows m = 4kq + i of the result = (dy, co)
                const int col = c_begin + cg * 16 + n;
                if (G::U == 1 && COUT != 1 && col >= w_out) continue;
                if constexpr (COUT == 1 && G::U == 1) {   // D[column 4kq + i, row shift n]: 4 columns of row rb + n
                    const int row = r0 + rb + n, col4 = c_begin + cg * 16 + 4 * kq;
                    if (row >= h_out || n >= G::DY || col4 >= w_out) continue;
                    const size_t off = out_img + (size_t)row * w_out + col4;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fast_act(acc[i] + bias4[0], act, alpha);
                    if (col4 + 4 <= w_out && (off & 3) == 0) {
                        if (mask_act != UOCR_ACT_NONE) {
                            const float4 my = ld4(mask_y + off);
                            v[0] *= act_grad_from_output<float>(my.x, mask_act, mask_alpha);
                            v[1] *= act_grad_from_output<float>(my.y, mask_act, mask_alpha);
                            v[2] *= act_grad_from_output<float>(my.z, mask_act, mask_alpha);
                            v[3] *= act_grad_from_output<float>(my.w, mask_act, mask_alpha);
                        }
                        st4(out + off, make_float4(v[0], v[1], v[2], v[3]));
                    } else {                             // image edge, or a row that does not start on 8 bytes (odd width)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (col4 + i < w_out) {
                                float vi = v[i];
                                if (mask_act != UOCR_ACT_NONE)
                                    vi *= act_grad_from_output<float>(ld1(mask_y + off + i), mask_act, mask_alpha);
                                st1(out + off + i, vi);
                            }
                    }
                } else if constexpr (COUT == 1) {         // (depth to space) row kq, phase i
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int dyi = G::U == 2 ? kq : 4 * kq + i;
                        const int row = G::U == 2 ? 2 * (r0 + rb + dyi) + (i >> 1) : r0 + rb + dyi;
                        const int ocol = G::U == 2 ? 2 * col + (i & 1) : col;
                        if (row >= h_out || ocol >= w_out || dyi >= G::DY) continue;
                        const size_t off = out_img + (size_t)row * w_out + ocol;
                        float vi = fast_act(acc[i] + bias4[i], act, alpha);
                        if (mask_act != UOCR_ACT_NONE) vi *= act_grad_from_output<float>(ld1(mask_y + off), mask_act, mask_alpha);
                        st1(out + off, vi);
                    }
                } else if constexpr (G::U == 2) {         // rows m = (phase = kq, channel i): output pixel 2 P + phase
                    const int row = 2 * (r0 + rb) + (kq >> 1), ocol = 2 * col + (kq & 1);
                    if (row >= h_out || ocol >= w_out) continue;
                    const size_t off = out_img + ((size_t)row * w_out + ocol) * 4;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fast_act(acc[i] + bias4[i], act, alpha);
                    if (mask_act != UOCR_ACT_NONE) {
                        const float4 my = ld4(mask_y + off);
                        v[0] *= act_grad_from_output<float>(my.x, mask_act, mask_alpha);
                        v[1] *= act_grad_from_output<float>(my.y, mask_act, mask_alpha);
                        v[2] *= act_grad_from_output<float>(my.z, mask_act, mask_alpha);
                        v[3] *= act_grad_from_output<float>(my.w, mask_act, mask_alpha);
                    }
                    st4(out + off, make_float4(v[0], v[1], v[2], v[3]));
                } else if constexpr (COUT == 4) {
                    const int row = r0 + rb + kq;
                    if (row >= h_out) continue;
                    const size_t off = out_img + ((size_t)row * w_out + col) * 4;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fast_act(acc[i] + bias4[i], act, alpha);
                    if (mask_act != UOCR_ACT_NONE) {
                        const uocr_h4 my = mraw[cg];
                        v[0] *= act_grad_from_output<float>((float)my.x, mask_act, mask_alpha);
                        v[1] *= act_grad_from_output<float>((float)my.y, mask_act, mask_alpha);
                        v[2] *= act_grad_from_output<float>((float)my.z, mask_act, mask_alpha);
                        v[3] *= act_grad_from_output<float>((float)my.w, mask_act, mask_alpha);
                    }
                    st4(out + off, make_float4(v[0], v[1], v[2], v[3]));
                } else {                                 // COUT == 2: rows 2kq, 2kq + 1, both channels each
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int row = r0 + rb + 2 * kq + hh;
                        if (row >= h_out) continue;
                        const size_t off = out_img + ((size_t)row * w_out + col) * 2;
                        float v0 = fast_act(acc[2 * hh] + bias4[2 * hh], act, alpha);
                        float v1 = fast_act(acc[2 * hh + 1] + bias4[2 * hh + 1], act, alpha);
                        if (mask_act != UOCR_ACT_NONE) {
                            const float2 my = ld2(mask_y + off);
                            v0 *= act_grad_from_output<float>(my.x, mask_act, mask_alpha);
                            v1 *= act_grad_from_output<float>(my.y, mask_act, mask_alpha);
                        }
                        st2(out + off, make_float2(v0, v1));
                    }
                }
            }
        }
    }
}

template <class G>
int launch_h16(uocr_ctx* ctx, const void* in, const void* w, const void* bias, void* out, const void* mask_y, int n,
               int h_in, int w_in, int h_out, int w_out, int ph, int pw, float pad, int use_bias, int act, float alpha,
               int mask_act, float mask_alpha) {
    static int resident = 0;                             // blocks of this kernel one CU holds
    if (resident == 0) {
        int nb = 0;
        UOCR_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_h16_kernel<G>, 256, 0));
        resident = nb > 0 ? nb : 1;
    }
    const int hp = (h_out + G::U - 1) / G::U, wp = (w_out + G::U - 1) / G::U;      // the position grid
    const int tiles_x = (wp + G::BC - 1) / G::BC, tiles_y = (hp + G::BR - 1) / G::BR;
    const long ntiles = (long)n * tiles_y * tiles_x;
    UOCR_REQUIRE(ctx, ntiles < (1l << 31) && (long)h_in * w_in * G::C < (1l << 31));
    // 4 x the resident blocks: when other lanes hold part of the CUs only some blocks of a launch are resident and
    // the rest start late -- with exactly one block per slot the late ones still own 1/grid of the tiles each (measured
    // in the three-lane step: 6.66 -> 6.75 k pages/s; alone the kernels do not care)
    const long cap = (long)ctx->cu_count * resident * 4;
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    hipLaunchKernelGGL(conv_h16_kernel<G>, dim3(grid), dim3(256), 0, ctx->stream, (const _Float16*)in, (const float*)w,
                       (const float*)bias, (_Float16*)out, (const _Float16*)mask_y, h_in, w_in, h_out, w_out, ph, pw,
                       tiles_x, tiles_y, (int)ntiles, pad, use_bias, act, alpha, mask_act, mask_alpha);
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

}  // namespace

// which: 0 forward, 1 backward-data.  Measured against the vector kernels at 8 x 1024 x 2048 (tools/bench_h16.py):
// every 4-channel layer wins (1.2 - 2.8x); of the 1-channel layers only the stride-1 forward (49 -> 38 us) and the
// stride-2 1 -> 4 backward-data (30 -> 23 us) do -- 2-byte result stores and 25 FMAs per pixel leave the matrix
// cores nothing to win -- so the others stay on conv_fast.hip / conv_up.hip.
bool uocr_conv_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which) {
    if (UOCR_DTYPE_BASE(dtype) != UOCR_F16 || !ctx->opt_fast || !ctx->opt_h16 || d.n > 65535) return false;
    if (same5x5(d)) return (d.cin == 4 && (d.cout == 2 || d.cout == 4)) || (d.cin == 1 && d.cout == 1 && which == 0);
    if (half5x5(d)) return (d.cin == 4 && d.cout == 4) || (d.cin == 1 && d.cout == 4 && which == 1);
    return false;
}

int uocr_conv_fwd_h16(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                      double pad_value, int use_bias, int act, double act_alpha) {
    auto run = [&](auto geo) {
        using G = decltype(geo);
        return launch_h16<G>(ctx, x, w, b, y, nullptr, d.n, d.h, d.w, d.oh, d.ow, d.ph, d.pw, (float)pad_value, use_bias,
                             act, (float)act_alpha, UOCR_ACT_NONE, 0.f);
    };
    if (d.sh == 2) return run(Geo<4, 4, 5, 5, 2, M_FWD>{});
    if (d.cin == 1) return run(Geo<1, 1, 5, 5, 1, M_FWD>{});
    if (d.cout == 2) return run(Geo<4, 2, 5, 5, 1, M_FWD>{});
    return run(Geo<4, 4, 5, 5, 1, M_FWD>{});
}

int uocr_conv_dgrad_h16(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d,
                        const ActMask& mask) {
    const int mact = mask.y ? mask.act : UOCR_ACT_NONE;
    // stride 2: a 3x3 window over dy per 2x2 block of dx;  stride 1: a forward conv over dy with flipped taps
    // (padding kh - 1 - ph)
    auto run = [&](auto geo, int ph, int pw) {
        using G = decltype(geo);
        return launch_h16<G>(ctx, dy, w, nullptr, dx, mask.y, d.n, d.oh, d.ow, d.h, d.w, ph, pw, 0.f, 0, UOCR_ACT_NONE, 0.f,
                             mact, (float)mask.alpha);
    };
    if (d.sh == 2) {
        if (d.cin == 4) return run(Geo<4, 4, 3, 3, 1, M_S2DGRAD>{}, 1, 1);
        return run(Geo<4, 1, 3, 3, 1, M_S2DGRAD>{}, 1, 1);
    }
    const int ph = d.kh - 1 - d.ph, pw = d.kw - 1 - d.pw;
    if (d.cout == 2) return run(Geo<2, 4, 5, 5, 1, M_DGRAD>{}, ph, pw);
    return run(Geo<4, 4, 5, 5, 1, M_DGRAD>{}, ph, pw);
}

// Upsample2D(2) + conv 5x5 / padding 2, 4 -> 4 channels, on the low-res grid (conv_up.hip)
bool uocr_upconv_h16_eligible(uocr_ctx* ctx, int dtype, int cin, int cout) {
    return UOCR_DTYPE_BASE(dtype) == UOCR_F16 && ctx->opt_fast && ctx->opt_h16 && cin == 4 && cout == 4;
}

int uocr_upconv_fwd_h16(uocr_ctx* ctx, const void* x_low, const void* w, const void* b, void* y, int n, int hl, int wl,
                        int use_bias, int act, double act_alpha) {
    return launch_h16<Geo<4, 4, 3, 3, 1, M_UPFWD>>(ctx, x_low, w, b, y, nullptr, n, hl, wl, 2 * hl, 2 * wl, 1, 1, 0.f,
                                                   use_bias, act, (float)act_alpha, UOCR_ACT_NONE, 0.f);
}

int uocr_upconv_dgrad_h16(uocr_ctx* ctx, const void* dy, const void* w, void* dx_low, int n, int hl, int wl,
                          const void* mask_y, int mask_act, double mask_alpha) {
    return launch_h16<Geo<4, 4, 6, 6, 2, M_UPDGRAD>>(ctx, dy, w, nullptr, dx_low, mask_y, n, 2 * hl, 2 * wl, hl, wl, 2, 2,
                                                     0.f, 0, UOCR_ACT_NONE, 0.f, mask_y ? mask_act : UOCR_ACT_NONE,
                                                     (float)mask_alpha);
}
