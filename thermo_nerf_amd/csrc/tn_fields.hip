// Per-sample field kernels — the Field plugin surface (arbitrary sample sets, one lane per sample).
//
//   tn_density_fwd        NS HashMLPDensityField.density_fn                
//   tn_field_density_fwd  NS NerfactoField.get_density                     (a8)  [REF thermal_field.py:186-190]
//   tn_field_heads_fwd    ThermalNerfactoTField.get_outputs                (a9)  [REF thermal_field.py:108-181]
//
// Straightforward forms: hash-grid gathers per lane, MLP weights staged once per block in LDS (transposed so
// that broadcast ds_reads feed the FMAs) and hidden activations in registers.  The fused ray kernel
// (tn_render.hip) is the throughput path; these serve Field.get_density / get_outputs / density_fn callers and
// are the on-GPU cross-check for the fused kernel.
#include "tn_field_eval.h"

using namespace tn;

namespace {

constexpr int kBlock = 256;

template <int H>
__global__ void __launch_bounds__(kBlock)
density_kernel(Grid g, tn_space space, const float *w0, const float *b0, const float *w1, const float *b1,
               float avg_density, const float *__restrict__ positions, long long n, float *__restrict__ density) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TwoLayerLds w = stage_two_layer<H>(smem, w0, b0, w1, b1, 2 * g.num_levels, 1);
    __syncthreads();
    const Space sp = make_space(space);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float px, py, pz;
        const float sel = normalize_position(sp, positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], px, py, pz);
        density[i] = proposal_density_eval<H>(g, w, avg_density, px, py, pz, sel);
    }
}

__global__ void __launch_bounds__(kBlock)
field_density_kernel(Grid g, tn_space space, const float *w0, const float *b0, const float *w1, const float *b1,
                     int geo_dim, float avg_density, const float *__restrict__ positions, long long n,
                     float *__restrict__ density, float *__restrict__ geo) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nout = 1 + geo_dim;
    const TwoLayerLds w = stage_two_layer<HW>(smem, w0, b0, w1, b1, 2 * g.num_levels, nout);
    __syncthreads();
    const Space sp = make_space(space);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float px, py, pz;
        const float sel = normalize_position(sp, positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], px, py, pz);
        float hid[HW];
        hidden_from_grid<HW>(g, w, px, py, pz, hid);
        for (int o = 0; o < nout; ++o) {
            float acc = w.B1[o];
            const float *wr = w.W1 + o * HW;
#pragma unroll
            for (int h = 0; h < HW; ++h) acc = fmaf(wr[h], hid[h], acc);
            if (o == 0)
                density[i] = mul_rn(mul_rn(avg_density, expf(acc)), sel);
            else
                geo[i * geo_dim + (o - 1)] = acc;
        }
    }
}

__global__ void __launch_bounds__(kBlock)
field_heads_kernel(HeadsArgs a, const float *__restrict__ dirs, const float *__restrict__ geo,
                   const int *__restrict__ cam, long long n, int training, float *__restrict__ rgb,
                   float *__restrict__ thermal) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const HeadsLds w = stage_heads(smem, a, training);
    __syncthreads();
    const int G = a.geo_dim, A = a.app_dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float *app = training ? (a.appearance + (long long)cam[i] * A) : w.APP;
        float c[3], t;
        heads_eval<0>(w, G, A, a.sh_shifted, dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2], geo + i * G, app, c, t);
        rgb[i * 3 + 0] = c[0];
        rgb[i * 3 + 1] = c[1];
        rgb[i * 3 + 2] = c[2];
        thermal[i] = t;
    }
}

inline unsigned grid_for(long long n) {
    const long long b = (n + kBlock - 1) / kBlock;
    return (unsigned)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

}  // namespace

extern "C" {

int tn_density_fwd(const tn_density_field *f, const float *positions, int64_t n, float *density, void *stream) {
    if (n == 0) return TN_OK;
    if (!f || !positions || !density) return TN_ERR_NULL;
    TN_TRY(tn_check_density_field(f));
    if (n < 0) return TN_ERR_SHAPE;
    if (n == 0) return TN_OK;
    const int H = f->l0.out_dim;
    const Grid g = tn_make_grid(f->grid);
    const size_t smem = (size_t)two_layer_floats(2 * f->grid.num_levels, H, 1) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (H == 16)
        hipLaunchKernelGGL(density_kernel<16>, dim3(grid_for(n)), dim3(kBlock), smem, s, g, f->space, f->l0.weight,
                           f->l0.bias, f->l1.weight, f->l1.bias, f->average_init_density, positions, (long long)n, density);
    else
        hipLaunchKernelGGL(density_kernel<64>, dim3(grid_for(n)), dim3(kBlock), smem, s, g, f->space, f->l0.weight,
                           f->l0.bias, f->l1.weight, f->l1.bias, f->average_init_density, positions, (long long)n, density);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_field_density_fwd(const tn_thermal_field *f, const float *positions, int64_t n, float *density, float *geo,
                         void *stream) {
    if (n == 0) return TN_OK;
    if (!f || !positions || !density || !geo) return TN_ERR_NULL;
    TN_TRY(tn_check_thermal_field(f));
    if (n < 0) return TN_ERR_SHAPE;
    if (n == 0) return TN_OK;
    const Grid g = tn_make_grid(f->grid);
    const size_t smem = (size_t)two_layer_floats(2 * f->grid.num_levels, HW, 1 + f->geo_feat_dim) * sizeof(float);
    hipLaunchKernelGGL(field_density_kernel, dim3(grid_for(n)), dim3(kBlock), smem, (hipStream_t)stream, g, f->space,
                       f->base0.weight, f->base0.bias, f->base1.weight, f->base1.bias, f->geo_feat_dim,
                       f->average_init_density, positions, (long long)n, density, geo);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_field_heads_fwd(const tn_thermal_field *f, const float *directions, const float *geo,
                       const int32_t *camera_indices, int64_t n, int32_t training, float *rgb, float *thermal,
                       void *stream) {
    if (n == 0) return TN_OK;
    if (!f || !directions || !geo || !rgb || !thermal) return TN_ERR_NULL;
    if (training && !camera_indices) return TN_ERR_NULL;
    TN_TRY(tn_check_thermal_field(f));
    if (n < 0) return TN_ERR_SHAPE;
    if (n == 0) return TN_OK;
    const HeadsArgs a = make_heads_args(f);
    const size_t smem = (size_t)heads_floats(f->geo_feat_dim, f->app_dim) * sizeof(float);
    if (!tn_ensure_dynamic_lds<field_heads_kernel>(smem)) return TN_ERR_LAUNCH;
    hipLaunchKernelGGL(field_heads_kernel, dim3(grid_for(n)), dim3(kBlock), smem, (hipStream_t)stream, a, directions, geo,
                       camera_indices, (long long)n, training, rgb, thermal);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
