// Shared by the strip kernels of the fused Monochrome block (conv_pair_strip.hip: float32, conv_pair_strip_h.hip:
// binary16 storage): LDS layouts, MFMA helpers, the partial-sum layout and the finish launch.  Not part of the C ABI.
#pragma once
#include <type_traits>

#include "uocr_common.h"

namespace pair_strip {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CH = 16;
constexpr int XROW = 80;      // floats per (ring slot, group) of the x ring: copies tx = 0, 1, 2, ones, dump
constexpr int GROW = 64;      // ... of the g ring: copies tx = 0, 1, 2, dump
// d_a1 transpose scratch of a group: [channel quad q][slot f(pos)][4 channels], plane stride TPL = 64 + 8 floats.
// f sends the positions {0-3, 12-15} to the even and {4-11} to the odd slots: a ds_read_b128 lane group holds exactly
// those two position sets of two neighbouring quads (MI355X LDS lane groups), so its 16 lanes hit 16 different 16-B
// slots, and the 32 lanes of a ds_write_b32 group hit 32 different banks (row stride 20 floats: 2- to 3-way conflicts)
constexpr int TPL = 72;
constexpr int TRSZ = 4 * TPL;
__device__ __forceinline__ int tslot(int pos) { return pos < 4 ? 2 * pos : pos >= 12 ? 2 * (pos - 12) + 8 : 2 * (pos - 4) + 1; }
constexpr int NSLOT = 6;      // rows of the output ring: two batches of three
constexpr int NPLANE = 3;     // tx = 0, 1, 2 (lane quarter 3 does not store)

template <int G>
struct Strip {
    // ring slot strides = 48 mod 64 floats: the three window rows a ds_read_b128 lane group reads (tap rows ty) then
    // fall on different 16-float bank units ((3 ty + tx) mod 4) instead of on the same one
    static constexpr int XSLOT = (G * XROW + 63) / 64 * 64 + 48;
    static constexpr int GSLOT = (G * GROW + 63) / 64 * 64 + 48;
    static constexpr int XS = 3 * XSLOT;             // x ring [slot][group][XROW]
    static constexpr int GS = 3 * GSLOT;
    static constexpr int TR = G * TRSZ;              // one transpose scratch per group
    static constexpr int WAVE = XS + GS + TR;        // floats of wave-private LDS
    static constexpr int COLS = 16 * G;              // computed columns per wave
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int P>
using phase_t = std::integral_constant<int, P>;

// Keeps the MFMAs on either side in program order (everything else may still move across): hipcc otherwise
// clusters the MFMAs of one accumulator back to back, and a dependent v_mfma_f32_16x16x4_f32 issues every 40
// cycles instead of every 32 (measured here: 39.6 cycles per MFMA before, rounds of independent accumulators after)
__device__ __forceinline__ void mfma_round() { __builtin_amdgcn_sched_barrier(0x7F6); }

// partial[block][PAIR_NPART]: dW1^T (16 rows: taps 0..8, row 9 = db1) x 16 channels, dW2^T likewise, db2
constexpr int PAIR_NPART = 2 * 256 + 1;


}  // namespace pair_strip

// float64 sum of partial[nblocks][PAIR_NPART] -> dw1 / db1 / dw2 / db2 (x unscale, accumulate or overwrite)
int uocr_pair_strip_finish(uocr_ctx* ctx, const float* partial, float* dw1, float* db1, float* dw2, float* db2,
                           int nblocks, int use_b1, int use_b2, int accumulate, float unscale);
