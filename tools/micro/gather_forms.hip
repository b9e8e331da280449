// Micro-benchmark: does the ADDRESS FORM of a 64-lane gather change its per-instruction cost on gfx950?
//   form 0  global_load  v, v[lo:hi], off          (64-bit per-lane address)
//   form 1  global_load  v, v_off, s[base:base+1]  (scalar base + 32-bit per-lane offset: what the render kernels issue)
//   form 2  buffer_load  v, v_off, s[rsrc:rsrc+3], 0 offen   (raw buffer, 32-bit per-lane offset)
// by width (4 / 8 / 16 B per lane), by distinct 128-byte lines per instruction and by table size (16 KB: TCP-resident,
// 2 MB: L2-resident).  tools/micro/gather.hip measured 12.5 / 17.5 / 17.5 cycles per CU and instruction for form 0/1 whatever
// the width — the floor under proposal_rays_kernel (20 dwordx4 gathers per sample).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_forms.hip -o /tmp/gather_forms && /tmp/gather_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <typename T> struct Raw;
template <> struct Raw<float> {
    static __device__ __forceinline__ float load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    }
};
template <> struct Raw<float2> {
    static __device__ __forceinline__ float2 load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    }
};
template <> struct Raw<float4> {
    static __device__ __forceinline__ float4 load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    }
};

template <typename T> __device__ __forceinline__ float first(const T &v) {  // every component: keeps the full-width load
    float s = 0.0f;
    for (unsigned c = 0; c < sizeof(T) / 4; ++c) s += reinterpret_cast<const float *>(&v)[c];
    return s;
}

template <typename T, int FORM>
__global__ void __launch_bounds__(256) k(const T *table, unsigned long long zero, unsigned bytes_log2, int lines, int iters,
                                         float *out) {
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned per = 64 / lines, grp = lane / per, within = lane % per;
    const unsigned line_mask = (1u << (bytes_log2 - 7)) - 1u;
    const unsigned elems_per_line = 128 / sizeof(T);
    const T *mine = table + zero * lane;  // form 0: a 64-bit per-lane address the compiler cannot fold (zero = 0 at run time)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 1u << bytes_log2, 0x00020000);
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 8 independent gathers in flight
            const unsigned line = hash32((wave * 131u + grp) * 2654435761u + (unsigned)(i * 8 + u) * 40503u) & line_mask;
            const unsigned e = line * elems_per_line + (within % elems_per_line);
            if (FORM == 2) v[u] = Raw<T>::load(rsrc, e * (unsigned)sizeof(T));
            else if (FORM == 1) v[u] = *reinterpret_cast<const T *>(reinterpret_cast<const char *>(table) + e * (unsigned)sizeof(T));  // 32-bit byte offset
            else v[u] = mine[e];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += first(v[u]);
    }
    if (acc == 1234.5f) out[0] = acc;
}

template <typename T, int FORM>
void run(const char *name, const T *table, unsigned long long zero, unsigned log2b, float *out) {
    for (int lines : {1, 4, 16, 64}) {
        const int blocks = 256 * 8, iters = 64;  // 8 blocks (32 waves) per CU
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL((k<T, FORM>), dim3(blocks), dim3(256), 0, 0, table, zero, log2b, lines, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<T, FORM>), dim3(blocks), dim3(256), 0, 0, table, zero, log2b, lines, iters, out);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)blocks * 4 * iters * 8;  // wave-level gather instructions
        const double per_cu_ns = ms * 1e6 / (instr / 256.0);
        printf("%-34s table %5u KB  %2d lines/instr: %7.3f ms  %6.2f ns per instr per CU (%5.1f cycles @2.3 GHz)\n", name,
               (1u << log2b) >> 10, lines, ms, per_cu_ns, per_cu_ns * 2.3);
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    }
}

// the same gathers from LDS (a 32 KB table per block, staged once): ds_read_b32 / b64 / b128 with per-lane addresses
template <typename T>
__global__ void __launch_bounds__(256) klds(const T *table, int lines, int iters, float *out) {
    __shared__ T tab[32768 / sizeof(T)];
    for (unsigned e = threadIdx.x; e < 32768 / sizeof(T); e += 256) tab[e] = table[e];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned per = 64 / lines, grp = lane / per, within = lane % per;
    const unsigned elems_per_line = 128 / sizeof(T);
    float acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned line = hash32((wave * 131u + grp) * 2654435761u + (unsigned)(i * 8 + u) * 40503u) & 255u;
            v[u] = tab[line * elems_per_line + (within % elems_per_line)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += first(v[u]);
    }
    if (acc == 1234.5f) out[0] = acc;
}
template <typename T>
void runlds(const char *name, const T *table, float *out) {
    for (int lines : {1, 4, 16, 64}) {
        const int blocks = 256 * 4, iters = 128;  // 4 blocks (16 waves) per CU: 128 KB of LDS
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL((klds<T>), dim3(blocks), dim3(256), 0, 0, table, lines, iters, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((klds<T>), dim3(blocks), dim3(256), 0, 0, table, lines, iters, out);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)blocks * 4 * iters * 8;
        const double per_cu_ns = ms * 1e6 / (instr / 256.0);
        printf("%-34s table    32 KB  %2d lines/instr: %7.3f ms  %6.2f ns per instr per CU (%5.1f cycles @2.3 GHz)\n", name, lines, ms,
               per_cu_ns, per_cu_ns * 2.3);
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    }
}

int main() {
    void *table; float *out;
    CK(hipMalloc(&table, 128u << 20));
    CK(hipMemset(table, 0, 128u << 20));
    CK(hipMalloc(&out, 4));
    const unsigned long long zero = getenv("GATHER_FORMS_NONZERO") ? 1ull : 0ull;
    for (unsigned log2b : {14u, 21u}) {
#define ROW(T, W)                                                                                            \
        run<T, 0>("global, 64-bit vaddr,   " W, (const T *)table, zero, log2b, out);          \
        run<T, 1>("global, saddr + voffset, " W, (const T *)table, zero, log2b, out);         \
        run<T, 2>("buffer, offen,           " W, (const T *)table, zero, log2b, out);
        ROW(float, "4 B")
        ROW(float2, "8 B")
        ROW(float4, "16 B")
    }
    runlds<float>("LDS ds_read_b32,          4 B", (const float *)table, out);
    runlds<float2>("LDS ds_read_b64,          8 B", (const float2 *)table, out);
    runlds<float4>("LDS ds_read_b128,        16 B", (const float4 *)table, out);
    return 0;
}
