// Micro-benchmark: issue rate of fp32 VALU forms on gfx950 — v_fma_f32 against v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32
// (VGPR sources and an SGPR-pair source), at 1, 2 and 4 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, long long *cycles, int iters, float sa, float sb) {
    f32x2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f32x2{(float)threadIdx.x + i, (float)i};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    const f32x2 sm = {sa, sb};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) {  // 2 x v_fma_f32
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
                } else if (MODE == 1) {  // 1 x v_pk_fma_f32
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c));
                } else if (MODE == 2) {  // v_pk_add_f32
                    asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                } else if (MODE == 3) {  // v_pk_mul_f32
                    asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
                } else if (MODE == 4) {  // v_pk_fma_f32 with SGPR pair source
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sm), "v"(c));
                } else if (MODE == 5) {  // v_pk_fma_f32 broadcast low half of src1 (op_sel_hi:[1,0,1])
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "v"(c));
                } else if (MODE == 6) {  // 2 x v_add_f32
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
                    asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].y) : "v"(m.y));
                } else if (MODE == 7) {  // 2 x v_fmac with SGPR
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "s"(sa), "v"(c.x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].y) : "s"(sb), "v"(c.y));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_pair, float *out, long long *cyc, int threads) {
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.9999f);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0001f, 0.9999f);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double pairs = (double)iters * 32;           // fp32 pairs (2 lanes' worth of work) per wave
    const int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    printf("%-44s threads %4d  %7.3f ms  wave clock/pair %6.2f (x%d instr)  -> per SIMD per pair %6.2f\n", name, threads, ms,
           (double)c / pairs, per_pair, (double)c / pairs / waves_per_simd);
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc(&out, 256 * 1024 * 4));
    CK(hipMalloc(&cyc, 8));
    for (int threads : {256, 512, 1024}) {
        run<0>("2 x v_fma_f32", 2, out, cyc, threads);
        run<7>("2 x v_fmac_f32 (SGPR src)", 2, out, cyc, threads);
        run<1>("1 x v_pk_fma_f32", 1, out, cyc, threads);
        run<4>("1 x v_pk_fma_f32 (SGPR-pair src)", 1, out, cyc, threads);
        run<5>("1 x v_pk_fma_f32 op_sel_hi:[1,0,1]", 1, out, cyc, threads);
        run<6>("2 x v_add_f32", 2, out, cyc, threads);
        run<2>("1 x v_pk_add_f32", 1, out, cyc, threads);
        run<3>("1 x v_pk_mul_f32", 1, out, cyc, threads);
    }
    return 0;
}
