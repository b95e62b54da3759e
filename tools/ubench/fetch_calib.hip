// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the strip kernels: every lane reads ONE dword
// (buffer_load_dword / global_load_dword, 256 contiguous bytes per wave instruction), against the 16-byte-per-lane
// pattern for which the gfx950 x2 correction is documented (MI355X_MICROARCH.md, HBM).  Reads 512 MiB once (twice the
// Infinity Cache) with each width and writes 4 bytes per block:
//     hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib
// FETCH_SIZE (KB) of read_dword / read_dwordx4 against 524288 KB read gives the factor per width.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_dword(const float* __restrict__ p, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;       // (never true for the fill value: keeps the loads alive)
}

__global__ void read_dwordx4(const float4* __restrict__ p, float* out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

__global__ void read_ushort(const unsigned short* __restrict__ p, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += (float)p[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = 512ull << 20, n = bytes / 4;
    float *p, *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4096 * 4) != hipSuccess) return 1;
    hipMemset(p, 0x3c, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_dword, dim3(2048), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(read_dwordx4, dim3(2048), dim3(256), 0, 0, (const float4*)p, out, n / 4);
        hipLaunchKernelGGL(read_ushort, dim3(2048), dim3(256), 0, 0, (const unsigned short*)p, out, n * 2);
    }
    hipDeviceSynchronize();
    printf("read %zu KB per launch with each kernel\n", bytes >> 10);
    return 0;
}
