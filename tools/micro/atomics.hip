// Micro-benchmark: rate of scattered fp32 atomic adds (pairs on neighbouring lanes, the pattern of hash_encode_bwd) into a 64 MB
// table, by memory scope and by whether an XCD only touches its own slice of the table.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int SCOPE, bool PARTITION>
__global__ void __launch_bounds__(256) k(float *table, unsigned entries_log2, long long per_thread, unsigned *census) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0) atomicOr(census, 1u << xcc);
    const unsigned mask = (1u << entries_log2) - 1u;
    for (long long i = 0; i < per_thread; ++i) {
        // lane pairs add the two floats of one 8-byte entry
        unsigned e = hash32((tid >> 1) * 7919u + (unsigned)i * 104729u) & mask;
        if (PARTITION) e = (e & ~(7u << (entries_log2 - 3))) | (xcc << (entries_log2 - 3));  // top 3 bits = this XCD's slice
        float *p = table + (size_t)e * 2 + (lane & 1);
        if (SCOPE == 0) unsafeAtomicAdd(p, 1.0f);
        else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

template <int SCOPE, bool PARTITION>
void run(const char *name, float *table, unsigned log2e, unsigned *census) {
    const int blocks = 2048, threads = 256;
    const long long per_thread = 96;  // 2048*256*96 = 50.3 M float atomics = 25.2 M pairs
    CK(hipMemset(table, 0, (size_t)8 << log2e));
    CK(hipMemset(census, 0, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<SCOPE, PARTITION>), dim3(blocks), dim3(threads), 0, 0, table, log2e, per_thread, census);
    CK(hipDeviceSynchronize());
    CK(hipMemset(table, 0, (size_t)8 << log2e));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<SCOPE, PARTITION>), dim3(blocks), dim3(threads), 0, 0, table, log2e, per_thread, census);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // checksum: every add is 1.0 -> the table must sum to the number of adds if no update was lost
    size_t n = (size_t)2 << log2e;
    float *h = (float *)malloc(n * 4);
    CK(hipMemcpy(h, table, n * 4, hipMemcpyDeviceToHost));
    double sum = 0; for (size_t i = 0; i < n; ++i) sum += h[i];
    free(h);
    unsigned c; CK(hipMemcpy(&c, census, 4, hipMemcpyDeviceToHost));
    const double adds = (double)blocks * threads * per_thread;
    printf("%-34s %8.3f ms  %7.2f G float-adds/s  sum/adds = %.6f  xcc mask 0x%x\n", name, ms, adds / ms / 1e6, sum / adds, c);
}

// GROUP consecutive lanes add to GROUP consecutive floats (GROUP * 4 contiguous bytes): does the rate follow lanes or segments?
template <int GROUP>
__global__ void __launch_bounds__(256) kg(float *table, unsigned floats_log2, long long per_thread) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned mask = (1u << floats_log2) - 1u;
    for (long long i = 0; i < per_thread; ++i) {
        unsigned f = (hash32((tid / GROUP) * 7919u + (unsigned)i * 104729u) * GROUP) & mask;
        unsafeAtomicAdd(table + f + (tid % GROUP), 1.0f);
    }
}
template <int GROUP>
void rung(float *table, unsigned log2e) {
    const int blocks = 2048, threads = 256;
    const long long per_thread = 96;
    CK(hipMemset(table, 0, (size_t)8 << log2e));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((kg<GROUP>), dim3(blocks), dim3(threads), 0, 0, table, log2e + 1, per_thread);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((kg<GROUP>), dim3(blocks), dim3(threads), 0, 0, table, log2e + 1, per_thread);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double adds = (double)blocks * threads * per_thread;
    printf("contiguous group of %2d lanes (%3d B)   %8.3f ms  %7.2f G float-adds/s  %7.2f G segments/s\n", GROUP, GROUP * 4, ms,
           adds / ms / 1e6, adds / GROUP / ms / 1e6);
}

int main(int argc, char **argv) {
    float *table; unsigned *census;
    CK(hipMalloc(&table, (size_t)8 << 23));
    CK(hipMalloc(&census, 4));
    if (argc > 1) {
        // L2-residency sweep: table of 2^log2e entries x 8 B, whole or sliced by XCD (slice = table / 8)
        for (int i = 1; i < argc; ++i) {
            const unsigned log2e = (unsigned)atoi(argv[i]);
            printf("-- table 2^%u entries = %.2f MB (%.2f MB per XCD slice)\n", log2e, (double)(8u << log2e) / 1048576.0, (double)(1u << log2e) / 1048576.0);
            run<1, false>("agent scope, whole table", table, log2e, census);
            run<2, false>("workgroup scope, whole table", table, log2e, census);
            run<1, true>("agent scope, XCD-partitioned", table, log2e, census);
            run<2, true>("workgroup scope, XCD-partitioned", table, log2e, census);
        }
        return 0;
    }
    const unsigned log2e = 23;  // 8 M entries x 8 B = 64 MB (16 levels x 2^19)
    run<0, false>("unsafeAtomicAdd, whole table", table, log2e, census);
    run<1, false>("agent scope, whole table", table, log2e, census);
    run<2, false>("workgroup scope, whole table", table, log2e, census);
    run<3, false>("wavefront scope, whole table", table, log2e, census);
    run<1, true>("agent scope, XCD-partitioned", table, log2e, census);
    run<2, true>("workgroup scope, XCD-partitioned", table, log2e, census);
    run<3, true>("wavefront scope, XCD-partitioned", table, log2e, census);
    rung<1>(table, log2e); rung<2>(table, log2e); rung<4>(table, log2e); rung<8>(table, log2e); rung<16>(table, log2e);
    rung<32>(table, log2e); rung<64>(table, log2e);
    return 0;
}
