// Do f32 MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2) and f32 VALU instructions of OTHER waves on the same SIMD
// overlap on gfx950, or do they share the FP32 ALUs?  One block of 512 threads per CU = 2 waves per SIMD:
//   mode 0: waves 0-3 MFMA only, waves 4-7 idle     mode 1: waves 0-3 idle, waves 4-7 VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (co-resident on every SIMD)
//   mode 3: every wave alternates 1 MFMA : VPM VALU in its own instruction stream
//   mode 4/5: as 0/2 with the f16 MFMA (v_mfma_f32_16x16x16_f16)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_coexec.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

template <int VPM>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    const f16x4 ah = {(_Float16)a, (_Float16)a, (_Float16)a, (_Float16)a}, bh = {(_Float16)b, (_Float16)b, (_Float16)b, (_Float16)b};
    const bool mf = (mode == 0 || mode == 2 || mode == 4 || mode == 5) && wave < 4;
    const bool va = (mode == 1 || mode == 2 || mode == 5) && wave >= 4;
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < VPM; ++q) v[(j * VPM + q) & 7] = fmaf(v[(j * VPM + q) & 7], b, a);
            }
        }
    } else if (mf) {
        if (mode >= 4) {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[j], 0, 0, 0);
        } else {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    } else if (va) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4 * VPM; ++q) v[q & 7] = fmaf(v[q & 7], b, a);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](auto kern, int mode, const char* what) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.3f ms  = %7.1f ns per iteration (4 MFMA and/or 4*VPM VALU per wave)\n", what, ms, ms * 1e6 / iters);
    };
    printf("VPM = 6 (24 v_fma per 4 MFMAs)\n");
    run(k<6>, 0, "f32 MFMA 16x16x4 only (waves 0-3)");
    run(k<6>, 1, "VALU only (waves 4-7)");
    run(k<6>, 2, "f32 MFMA (waves 0-3) beside VALU (waves 4-7)");
    run(k<6>, 3, "every wave: 1 f32 MFMA : 6 v_fma interleaved");
    run(k<6>, 4, "f16 MFMA 16x16x16 only (waves 0-3)");
    run(k<6>, 5, "f16 MFMA (waves 0-3) beside VALU (waves 4-7)");
    printf("VPM = 2 (8 v_fma per 4 MFMAs)\n");
    run(k<2>, 1, "VALU only (waves 4-7)");
    run(k<2>, 2, "f32 MFMA (waves 0-3) beside VALU (waves 4-7)");
    run(k<2>, 3, "every wave: 1 f32 MFMA : 2 v_fma interleaved");
    return 0;
}
