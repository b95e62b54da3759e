// Shared between the vertical-Toeplitz convolution kernels (conv_h16.hip: binary16 MFMAs, conv_t32.hip: float32
// MFMAs): what a result row of the weight operand means in each mode and where its value comes from.
#pragma once
#include "conv_dims.h"

namespace {

enum Mode {
    M_FWD = 0,       // forward conv, w[ky][kx][ci][co]
    M_DGRAD = 1,     // backward-data of a stride-1 conv: input dy (C = cout), flipped taps, w[ky][kx][co_out][ci_in]
    M_UPDGRAD = 2,   // backward-data of upsample2x + 5x5: stride-2 6x6 window over dy, taps summed per parity phase
    M_UPFWD = 3,     // upsample2x + 5x5 forward: 3x3 window on the low-res input, 4 phases x COUT result rows, the
                     // phases are the 2x2 output pixels of the source pixel (depth to space in the store)
    M_S2DGRAD = 4,   // backward-data of a 5x5 / stride 2 / padding 2 conv: 3x3 window over dy, result rows =
                     // (phase, ci) = the 2x2 input pixels around the source -- the same store
};

// taps k of one axis of the 5x5 kernel that land on source offset mi - 1 for output parity `phase`: [lo, hi)
// (upsample2x + conv, see conv_up.hip)
__device__ __forceinline__ void tap_group(int phase, int mi, int& lo, int& hi) {
    if (phase == 0) {
        lo = 2 * mi;
        hi = mi == 2 ? 5 : 2 * mi + 2;
    } else {
        lo = mi == 0 ? 0 : 2 * mi - 1;
        hi = mi == 0 ? 1 : 2 * mi + 1;
    }
}

constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }

// weight of window position (ty, tx), input channel ci, output channel co, from the layer's float32 weights
// (w = their copy in LDS: every lane builds its NM x 4 operand values from it once per block)
template <class G>
__device__ __forceinline__ float weight_of(const float* w, int ty, int tx, int ci, int co) {
    if constexpr (G::MODE == M_FWD) {
        return w[((ty * G::KW + tx) * G::C + ci) * G::COUT + co];
    } else if constexpr (G::MODE == M_DGRAD) {
        // dx[p][co] = sum dy[p + t - pad'][ci] w[K-1-t][co][ci]
        return w[(((G::KH - 1 - ty) * G::KW + (G::KW - 1 - tx)) * G::COUT + co) * G::C + ci];
    } else if constexpr (G::MODE == M_UPDGRAD) {
        // dxl[Q][co] = sum_{a,b in 0..5} dy[2Q - 2 + (a,b)][ci] Weff[phase (a%2, b%2)][o = (4 - a + py) / 2, ..][co][ci],
        // Weff = the 5x5 taps of the phase that share a source pixel (at most 2 x 2 of them), summed
        const int py = ty & 1, px = tx & 1;
        int ylo, yhi, xlo, xhi;
        tap_group(py, (4 - ty + py) >> 1, ylo, yhi);
        tap_group(px, (4 - tx + px) >> 1, xlo, xhi);
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ky = min(ylo + a, 4), kx = min(xlo + b, 4);
                const float v = w[((ky * 5 + kx) * G::COUT + co) * G::C + ci];
                s += (ylo + a < yhi && xlo + b < xhi) ? v : 0.f;
            }
        return s;
    } else if constexpr (G::MODE == M_UPFWD) {
        // y[2P + phase][o] = sum_{m in 3x3, ci} Weff[phase][m][ci][o] xl[P + m - 1][ci]; row co = phase * COUT + o
        const int phase = co / G::COUT, o = co % G::COUT;
        int ylo, yhi, xlo, xhi;
        tap_group(phase >> 1, ty, ylo, yhi);
        tap_group(phase & 1, tx, xlo, xhi);
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ky = min(ylo + a, 4), kx = min(xlo + b, 4);
                const float v = w[((ky * 5 + kx) * G::C + ci) * G::COUT + o];
                s += (ylo + a < yhi && xlo + b < xhi) ? v : 0.f;
            }
        return s;
    } else {
        // dx[2P + phase][c] = sum_{m in 3x3, o} dy[P + m - 1][o] w[4 - 2 m + phase][c][o] (taps beyond 4: none);
        // row co = phase * COUT + c, input channel ci = o
        const int phase = co / G::COUT, c = co % G::COUT;
        const int ky = 4 - 2 * ty + (phase >> 1), kx = 4 - 2 * tx + (phase & 1);
        const float v = w[((min(ky, 4) * 5 + min(kx, 4)) * G::COUT + c) * G::C + ci];
        return (ky <= 4 && kx <= 4) ? v : 0.f;
    }
}

}  // namespace
