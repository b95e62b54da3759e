// Partial rows of the upsample(2x) + conv5x5 weight-gradient producers (conv_up.hip, conv_h16w.hip).
// The producers accumulate dWeff[(my, mx), c, (phase, o)] -- the gradient of the 3 x 3 per-phase weights -- and every dw[ky][kx]
// is the sum of its four phase entries.  That sum is taken by the producing block, so that a block's partial row holds dw and
// db themselves (UP4_NOUT = 5*5*4*4 + 4, UP1_NOUT = 5*5 + 1 floats) and the finish is a plain column sum (finish_group.h).
#pragma once
#include <hip/hip_runtime.h>

constexpr int UP4_NOUT = 404, UP1_NOUT = 26;

// source offset index (0..2) of tap k for output parity p (conv_up.hip: tap_group, in closed form)
__device__ __forceinline__ int up_tap_m(int k, int p) { return (k + p) >> 1; }

// value(i): the block's sum of entry i of the 4-channel dWeff row [((my*3 + mx)*4 + c)*16 + phase*4 + o], db at 576 + o
template <typename V>
__device__ __forceinline__ void up4_write_row(float* __restrict__ out, V value, int tid, int nthreads) {
    for (int e = tid; e < UP4_NOUT; e += nthreads) {
        if (e >= 400) {
            out[e] = value(576 + (e - 400));
            continue;
        }
        const int o = e & 3, c = (e >> 2) & 3, kk = e >> 4, ky = kk / 5, kx = kk % 5;
        float s = 0.f;
#pragma unroll
        for (int phase = 0; phase < 4; ++phase)
            s += value(((up_tap_m(ky, phase >> 1) * 3 + up_tap_m(kx, phase & 1)) * 4 + c) * 16 + phase * 4 + o);
        out[e] = s;
    }
}

// value(i): the block's sum of entry i of the 1-channel dWeff row [(my*3 + mx)*4 + phase], db at 36
template <typename V>
__device__ __forceinline__ void up1_write_row(float* __restrict__ out, V value, int tid) {
    if (tid >= UP1_NOUT) return;
    if (tid == 25) {
        out[25] = value(36);
        return;
    }
    const int ky = tid / 5, kx = tid % 5;
    float s = 0.f;
#pragma unroll
    for (int phase = 0; phase < 4; ++phase) s += value((up_tap_m(ky, phase >> 1) * 3 + up_tap_m(kx, phase & 1)) * 4 + phase);
    out[tid] = s;
}
