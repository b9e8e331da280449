// Micro-benchmark: cost of a 64-lane gather instruction by the number of distinct cache lines its lanes touch and by width
// (4 / 8 / 16 bytes per lane), table resident in L2 (4 MB per XCD) or in the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather.hip -o /tmp/gather && /tmp/gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every instruction: lanes are split into `lines` groups; a group reads consecutive elements of one random 128-byte-aligned
// line (group size 64/lines lanes, wrapping inside the line)
template <typename T>
__global__ void __launch_bounds__(256) k(const T *table, unsigned bytes_log2, int lines, int iters, float *out) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned per = 64 / lines, grp = lane / per, within = lane % per;
    const unsigned line_mask = (1u << (bytes_log2 - 7)) - 1u;
    const unsigned elems_per_line = 128 / sizeof(T);
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 8 independent gathers in flight
            const unsigned line = hash32((wave * 131u + grp) * 2654435761u + (unsigned)(i * 8 + u) * 40503u) & line_mask;
            v[u] = table[(size_t)line * elems_per_line + (within % elems_per_line)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += reinterpret_cast<const float *>(&v[u])[0];
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <typename T>
void run(const char *name, const void *table, unsigned log2b, float *out) {
    for (int lines : {1, 2, 4, 8, 16, 32, 64}) {
        const int blocks = 256 * 8, iters = 64;  // 8 blocks (32 waves) per CU
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, (const T *)table, log2b, lines, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, (const T *)table, log2b, lines, iters, out);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)blocks * 4 * iters * 8;  // wave-level gather instructions
        const double per_cu_ns = ms * 1e6 / (instr / 256.0);
        printf("%-22s table %4u MB  %2d lines/instr: %7.3f ms  %6.2f ns per gather instr per CU (%5.1f cycles @2.3 GHz)\n", name,
               (1u << log2b) >> 20, lines, ms, per_cu_ns, per_cu_ns * 2.3);
    }
}

// cost of a gather with only the first `active` lanes enabled (all in one line): does the texture-address pipe charge per
// instruction or per active quad?
__global__ void __launch_bounds__(256) kact(const float2 *table, int active, int iters, float *out) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            v[u] = make_float2(0.0f, 0.0f);
            const unsigned line = hash32(wave * 2654435761u + (unsigned)(i * 8 + u) * 40503u) & 16383u;
            if ((int)lane < active) v[u] = table[(size_t)line * 16 + (lane & 15)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    if (acc == 1234.5f) out[0] = acc;
}
void runact(const void *table, float *out) {
    for (int active : {1, 4, 8, 16, 32, 64}) {
        const int blocks = 256 * 8, iters = 64;
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(kact, dim3(blocks), dim3(256), 0, 0, (const float2 *)table, active, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kact, dim3(blocks), dim3(256), 0, 0, (const float2 *)table, active, iters, out);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)blocks * 4 * iters * 8;
        const double per_cu_ns = ms * 1e6 / (instr / 256.0);
        printf("dwordx2, %2d active lanes, one line: %7.3f ms  %6.2f ns per gather instr per CU (%5.1f cycles @2.3 GHz)\n", active, ms,
               per_cu_ns, per_cu_ns * 2.3);
    }
}

int main() {
    void *table; float *out;
    CK(hipMalloc(&table, 128u << 20));
    CK(hipMemset(table, 0, 128u << 20));
    CK(hipMalloc(&out, 4));
    runact(table, out);
    for (unsigned log2b : {21u}) {  // 2 MB (L2-resident) and 64 MB (Infinity Cache)
        run<float>("dword  (4 B/lane)", table, log2b, out);
        run<float2>("dwordx2 (8 B/lane)", table, log2b, out);
        run<float4>("dwordx4 (16 B/lane)", table, log2b, out);
    }
    return 0;
}
