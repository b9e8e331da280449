// Micro-benchmark: does the scattered-atomic rate (tools/micro/atomics.hip: ~21 G line transactions/s for fp32 adds) depend on
// the operand type (u32 / u64 / f32 / f64), on the number of CUs issuing, or on the lanes-per-line grouping?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomics_types.hip -o tools/micro/atomics_types_bin
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every lane adds to its own random 64-byte line (64 lines per instruction)
template <typename T>
__global__ void __launch_bounds__(256) k(T *table, unsigned lines_log2, int per_thread) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned mask = (1u << lines_log2) - 1u;
    for (int i = 0; i < per_thread; ++i) {
        const unsigned line = hash32(tid * 7919u + (unsigned)i * 104729u) & mask;
        T *p = reinterpret_cast<T *>(reinterpret_cast<char *>(table) + (size_t)line * 64);
        if constexpr (sizeof(T) == 4 && !__is_same(T, float)) atomicAdd(p, (T)1);
        else if constexpr (__is_same(T, float)) unsafeAtomicAdd(p, 1.0f);
        else if constexpr (__is_same(T, double)) unsafeAtomicAdd(p, 1.0);
        else atomicAdd(p, (T)1);
    }
}

template <typename T>
void run(const char *name, void *table, unsigned lines_log2, int blocks) {
    const long long total = 2048LL * 256 * 48;
    const int per_thread = (int)(total / ((long long)blocks * 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(256), 0, 0, (T *)table, lines_log2, per_thread);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<T>), dim3(blocks), dim3(256), 0, 0, (T *)table, lines_log2, per_thread);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-8s blocks %5d  lines 2^%2u  %8.3f ms  %7.2f G line-atomics/s\n", name, blocks, lines_log2, ms,
           (double)blocks * 256 * per_thread / ms / 1e6);
}

int main() {
    void *table;
    CK(hipMalloc(&table, (size_t)64 << 20));
    CK(hipMemset(table, 0, (size_t)64 << 20));
    const unsigned lg = 20;  // 1 M lines = 64 MB
    for (int blocks : {2048}) {
        run<float>("f32", table, lg, blocks);
        run<unsigned>("u32", table, lg, blocks);
        run<unsigned long long>("u64", table, lg, blocks);
        run<double>("f64", table, lg, blocks);
    }
    for (int blocks : {32, 64, 128, 256, 512, 1024, 4096}) run<float>("f32", table, lg, blocks);
    for (unsigned l : {10u, 12u, 14u, 16u}) run<float>("f32", table, l, 2048);
    for (unsigned l : {10u, 14u}) run<unsigned>("u32", table, l, 2048);
    return 0;
}
