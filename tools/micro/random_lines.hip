// Micro-benchmark: the chip's rate of RANDOM 64-byte-line reads (one 8-byte gather per lane, every lane of an instruction on a
// different random line — what the fine hash levels of a training batch look like: incoherent rays, 64 distinct lines per
// gather instruction) by table size: 2 MB (TCP/L2), 64 MB = the field's table (L2 misses, Infinity Cache hits), 256 MB
// (the Infinity Cache's size) and 2 GB (HBM).  Gives the line-granular bound of the training step's gathers
// (bench.py `line_granular_view`).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/random_lines.hip -o /tmp/random_lines && /tmp/random_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int INFLIGHT>
__global__ void __launch_bounds__(256) k(const float2 *table, unsigned long long line_mask, int iters, float *out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float2 v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            const unsigned long long h = ((unsigned long long)hash32(tid * 2654435761u + (unsigned)(i * INFLIGHT + u) * 40503u) << 7) ^
                                         hash32(tid + 77u * (unsigned)(i * INFLIGHT + u));
            v[u] = table[(h & line_mask) * 8 + (tid & 7)];  // 8 float2 per 64-byte line
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) acc += v[u].x;
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <int INFLIGHT>
void run(const void *table, unsigned long long bytes, int blocks_per_cu, float *out) {
    const int blocks = 256 * blocks_per_cu, iters = 256 / INFLIGHT * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned long long mask = bytes / 64 - 1;
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, (const float2 *)table, mask, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, (const float2 *)table, mask, iters, out);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double lines = (double)blocks * 256 * iters * INFLIGHT;
    printf("table %5llu MB  %2d waves/SIMD  %2d gathers in flight per lane: %8.3f ms  %7.2f G random lines/s  (%6.1f GB/s of 64-byte lines)\n",
           bytes >> 20, blocks_per_cu, INFLIGHT, ms, lines / ms / 1e6, lines * 64 / ms / 1e6);
}

int main() {
    void *table; float *out;
    const unsigned long long cap = 2ull << 30;
    CK(hipMalloc(&table, cap));
    CK(hipMemset(table, 0, cap));
    CK(hipMalloc(&out, 4));
    for (unsigned long long bytes : {2ull << 20, 64ull << 20, 256ull << 20, 2ull << 30})
        for (int bpc : {4, 8}) {
            run<8>(table, bytes, bpc, out);
            run<16>(table, bytes, bpc, out);
        }
    return 0;
}
