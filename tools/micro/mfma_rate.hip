// Micro-benchmark: back-to-back issue cost of the MFMA forms the training backward chooses between, one and two waves per SIMD
// (four independent accumulators per wave).  Prints shader cycles per wave-instruction per SIMD (s_memtime ticks).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, long long *cycles, int iters) {
    f32x4 acc[4] = {};
    f32x16 big[2] = {};
    const float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
    v8bf a8, b8;
    v4s a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(fa + i); b8[i] = (__bf16)(fb + i); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + i); b4[i] = (short)(0x3f00 + i); }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i], 0, 0, 0);
                if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
                if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
                if (MODE == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, big[i & 1], 0, 0, 0);
                if (MODE == 4) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, big[i & 1], 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    s += big[0][0] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
void run(const char *name, float *out, long long *cyc) {
    const int iters = 2000;
    for (int threads : {256, 512}) {  // one / two waves per SIMD
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        CK(hipDeviceSynchronize());
        long long c;
        CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
        // wall clock over a long launch: MFMAs per second per SIMD, in cycles of a 2.4 GHz clock
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int long_iters = 200000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, long_iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_wave = (double)c / ((double)iters * 16);
        const double ns_per_mfma_simd = ms * 1e6 / ((double)long_iters * 16 * (threads / 256));
        printf("%-34s %d wave(s)/SIMD: %6.2f counter ticks per MFMA of one wave; wall: %6.2f ns per MFMA per SIMD = %6.2f cycles at 2.4 GHz\n", name,
               threads / 256, per_wave, ns_per_mfma_simd, ns_per_mfma_simd * 2.4);
    }
}

int main() {
    float *out;
    long long *cyc;
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&cyc, sizeof(long long)));
    run<0>("v_mfma_f32_16x16x4_f32", out, cyc);
    run<1>("v_mfma_f32_16x16x32_bf16", out, cyc);
    run<2>("v_mfma_f32_16x16x16_bf16 (1k)", out, cyc);
    run<3>("v_mfma_f32_32x32x16_bf16", out, cyc);
    run<4>("v_mfma_f32_32x32x2_f32", out, cyc);
    return 0;
}
