// What does it cost ONE wave to alternate f32 MFMAs with other instruction classes on gfx950?
// Every wave runs NM MFMAs (v_mfma_f32_16x16x4_f32, 4 independent accumulators) then NV instructions of
// class X, repeated; time per (NM MFMA + NV X) group vs. the two parts alone.  1 wave per SIMD (256 threads/CU).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_switch.hip -o /tmp/sw && /tmp/sw
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum { X_NONE = 0, X_VALU = 1, X_LDS = 2, X_SALU = 3 };

template <int NM, int NV, int X, bool MF>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    lds[threadIdx.x + 256] = threadIdx.x;
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    int sacc = iters;
    const float* lp = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        if (MF) {
#pragma unroll
            for (int j = 0; j < NM; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
        }
        if (X == X_VALU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], b, a);
        } else if (X == X_LDS) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] += lp[(q & 7) * 64];     // ds_read + the add that consumes it
        } else if (X == X_SALU) {
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
        }
    }
    float s = sacc;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

float* out;
hipEvent_t e0, e1;
template <typename K>
float run(K kern, int iters) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, 50, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;       // ns per group
}

template <int NM, int NV, int X>
void row(const char* name) {
    const int iters = 20000;
    const float both = run(k<NM, NV, X, true>, iters), mf = run(k<NM, 0, X_NONE, true>, iters), other = run(k<NM, NV, X, false>, iters);
    printf("%2d MFMA + %2d %-5s : %7.1f ns   (MFMA alone %6.1f, %s alone %6.1f, sum %6.1f, extra %+6.1f)\n", NM, NV, name, both,
           mf, name, other, mf + other, both - mf - other);
}

int main() {
    hipMalloc(&out, 4);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    row<1, 2, X_VALU>("VALU");
    row<1, 6, X_VALU>("VALU");
    row<4, 8, X_VALU>("VALU");
    row<4, 24, X_VALU>("VALU");
    row<16, 32, X_VALU>("VALU");
    row<16, 96, X_VALU>("VALU");
    row<1, 2, X_LDS>("LDS");
    row<4, 8, X_LDS>("LDS");
    row<16, 32, X_LDS>("LDS");
    row<1, 4, X_SALU>("SALU");
    row<4, 16, X_SALU>("SALU");
    return 0;
}
