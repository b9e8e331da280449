// Micro-benchmark: LDS atomic throughput per CU — ds_add_f32 against ds_add_u32 (no return / with return), random addresses
// over a 128 KB slice (the bucketed scatter's owner pass) and a 1 KB range (coarse levels: many lanes per address).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomics.hip -o tools/micro/lds_atomics_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(1024, 1) k(float *out, int iters, unsigned mask) {
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < 32768; e += 1024) lds[e] = 0.0f;
    __syncthreads();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        const unsigned a = hash32(threadIdx.x * 7919u + i * 104729u + blockIdx.x) & mask;
        if (MODE == 0) __hip_atomic_fetch_add(lds + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(lds) + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) acc += __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(lds) + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) { lds[a] += 1.0f; }  // plain read-modify-write (loses colliding updates: rate only)
    }
    __syncthreads();
    float s = 0;
    for (int e = threadIdx.x; e < 32768; e += 1024) s += lds[e];
    out[blockIdx.x * 1024 + threadIdx.x] = s + (float)acc;
}

template <int MODE>
void run(const char *name, float *out, unsigned mask) {
    const int iters = 2000, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(1024), 131072, 0, out, iters, mask);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(1024), 131072, 0, out, iters, mask);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double ops = (double)blocks * 1024 * iters;
    printf("%-34s addresses %6u  %8.3f ms  %8.2f G lane-ops/s  (%.2f lane-ops per clock and CU at 2.4 GHz)\n", name, mask + 1, ms,
           ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
}

int main() {
    float *out;
    CK(hipMalloc(&out, 256 * 1024 * 4));
    for (unsigned mask : {32767u, 255u}) {
        run<0>("ds_add_f32", out, mask);
        run<1>("ds_add_u32", out, mask);
        run<2>("ds_add_rtn_u32", out, mask);
        run<3>("ds_read + add + ds_write", out, mask);
    }
    return 0;
}
