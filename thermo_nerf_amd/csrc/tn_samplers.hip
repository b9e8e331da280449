// Samplers, weights and renderers of the plugin surface (one wave64 per ray; rays are independent).
//
//   tn_frustum_positions  NS Frustums.get_positions
//   tn_sample_initial     NS UniformLinDispPiecewiseSampler | UniformSampler
//   tn_weights_fwd        NS RaySamples.get_weights                   (a6)  [REF thermal_nerf_model.py:233]
//   tn_sample_pdf         NS PDFSampler.generate_ray_samples          (a11)
//   tn_composite_fwd      ThermalRenderer / RGBRenderer(last_sample)  (a12,a13) [REF thermal_renderer.py:27-80,113-149]
//   tn_depth_fwd          Accumulation + Depth(median|expected)       (a13) [REF thermal_nerf_model.py:238-243,267-270]
//
// All of these are HBM-streaming kernels: every input element is read once, coalesced along the sample axis.
#include "tn_device.h"

using namespace tn;

namespace {

constexpr int kBlock = 256;  // 4 waves / block, one ray per wave
constexpr int kWavesPerBlock = kBlock / TN_WAVE;
constexpr int kMaxPdfIn = 1024;

__global__ void frustum_positions_kernel(const float *__restrict__ o, const float *__restrict__ d,
                                         const float *__restrict__ starts, const float *__restrict__ ends,
                                         long long total, int n, float *__restrict__ pos) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long r = i / n;
    const float s = starts[i], e = ends[i];
    pos[i * 3 + 0] = frustum_pos(o[r * 3 + 0], d[r * 3 + 0], s, e);
    pos[i * 3 + 1] = frustum_pos(o[r * 3 + 1], d[r * 3 + 1], s, e);
    pos[i * 3 + 2] = frustum_pos(o[r * 3 + 2], d[r * 3 + 2], s, e);
}

// the training step's form: both edge columns of eucl [R,n+1] are split into contiguous starts / ends / deltas here (the
// adjoint kernels read them per sample), so the level costs one launch instead of two strided copies + a subtraction + this
__global__ void frustum_from_edges_kernel(const float *__restrict__ o, const float *__restrict__ d,
                                          const float *__restrict__ eucl, long long total, int n,
                                          float *__restrict__ pos, float *__restrict__ starts,
                                          float *__restrict__ ends, float *__restrict__ deltas) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long r = i / n;
    const float s = eucl[i + r], e = eucl[i + r + 1];
    starts[i] = s;
    ends[i] = e;
    deltas[i] = e - s;
    pos[i * 3 + 0] = frustum_pos(o[r * 3 + 0], d[r * 3 + 0], s, e);
    pos[i * 3 + 1] = frustum_pos(o[r * 3 + 1], d[r * 3 + 1], s, e);
    pos[i * 3 + 2] = frustum_pos(o[r * 3 + 2], d[r * 3 + 2], s, e);
}

__global__ void sample_initial_kernel(const float *__restrict__ lin_bins, const float *__restrict__ t_rand,
                                      const float *__restrict__ nears, const float *__restrict__ fars,
                                      long long num_rays, int n, bool lin, bool per_sample, float *__restrict__ spacing,
                                      float *__restrict__ eucl) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = n + 1;
    if (i >= num_rays * nb) return;
    const long long r = i / nb;
    const int j = (int)(i - r * nb);
    float b = lin_bins[j];
    if (t_rand) {
        // bins = lower + (upper - lower) * t ; centers between neighbouring linspace points
        const float lo = (j == 0) ? lin_bins[0] : add_rn(lin_bins[j], lin_bins[j - 1]) / 2.0f;
        const float hi = (j == n) ? lin_bins[n] : add_rn(lin_bins[j + 1], lin_bins[j]) / 2.0f;
        b = add_rn(lo, mul_rn(sub_rn(hi, lo), t_rand[per_sample ? i : r]));  // single_jitter: one draw per ray; else [R, n+1]
    }
    const float sn = spacing_fn(nears[r], lin), sf = spacing_fn(fars[r], lin);
    spacing[i] = b;
    eucl[i] = spacing_to_eucl(b, sn, sf, lin);
}

// weights = nan_to_num((1 - exp(-d*sigma)) * exp(-exclusive_cumsum(d*sigma)))
__global__ void weights_kernel(const float *__restrict__ deltas, const float *__restrict__ dens, long long num_rays,
                               int n, float *__restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= num_rays) return;
    const float *dl = deltas + r * n;
    const float *dn = dens + r * n;
    float *w = weights + r * n;
    float carry = 0.0f;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const float dd = (i < n) ? mul_rn(dl[i], dn[i]) : 0.0f;
        const float incl = wave_incl_scan(dd, lane);
        const float excl = carry + wave_excl_from_incl(incl, lane);
        if (i < n) {
            const float alpha = sub_rn(1.0f, expf(-dd));
            w[i] = nan_to_num(mul_rn(alpha, expf(-excl)));
        }
        carry += lane_value<63>(incl);
    }
}

__global__ void sample_pdf_kernel(const float *__restrict__ weights, const float *__restrict__ existing,
                                  const float *__restrict__ u, const float *__restrict__ u_rand,
                                  const float *__restrict__ nears, const float *__restrict__ fars,
                                  long long num_rays, int n_in, int n_out, bool lin, bool per_sample, float *__restrict__ spacing,
                                  float *__restrict__ eucl) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * kWavesPerBlock + wave;
    float *cdf = smem + (size_t)wave * 2 * (n_in + 1);
    float *bins = cdf + (n_in + 1);
    if (r >= num_rays) return;
    const float *w = weights + r * n_in;
    const float *ex = existing + r * (long long)(n_in + 1);
    // pass 1: sum of padded weights
    float part = 0.0f;
    for (int i = lane; i < n_in; i += 64) part += add_rn(w[i], 0.01f);
    float ws = wave_sum(part);
    const float padding = fmaxf(sub_rn(1e-5f, ws), 0.0f);
    const float pad_each = padding / (float)n_in;
    ws = add_rn(ws, padding);
    // pass 2: cdf
    float carry = 0.0f;
    if (lane == 0) cdf[0] = 0.0f;
    for (int base = 0; base < n_in; base += 64) {
        const int i = base + lane;
        const float pdf = (i < n_in) ? add_rn(add_rn(w[i], 0.01f), pad_each) / ws : 0.0f;
        const float incl = wave_incl_scan(pdf, lane) + carry;
        if (i < n_in) cdf[i + 1] = fminf(1.0f, incl);
        carry = lane_value<63>(incl);
    }
    for (int i = lane; i <= n_in; i += 64) bins[i] = ex[i];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int nb = n_out + 1;
    const float sn = spacing_fn(nears[r], lin), sf = spacing_fn(fars[r], lin);
    const float jit = (u_rand && !per_sample) ? u_rand[r] / (float)nb : 0.0f;
    for (int j = lane; j < nb; j += 64) {
        // single_jitter: one draw per ray; else u_rand [R, n_out+1] (NS PDFSampler: u + rand / num_bins)
        const float uu = u_rand ? add_rn(u[j], per_sample ? u_rand[r * nb + j] / (float)nb : jit) : u[j];
        // searchsorted(cdf, uu, side="right"): first index with cdf[idx] > uu, over n_in+1 entries
        int lo = 0, hi = n_in + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = min(max(lo - 1, 0), n_in);
        const int above = min(max(lo, 0), n_in);
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float t = sub_rn(uu, c0) / sub_rn(c1, c0);
        t = nan_to_num(t);
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        const float b = add_rn(b0, mul_rn(t, sub_rn(b1, b0)));
        spacing[r * nb + j] = b;
        eucl[r * nb + j] = spacing_to_eucl(b, sn, sf, lin);
    }
}

template <int C>
__global__ void composite_kernel(const float *__restrict__ values, const float *__restrict__ weights,
                                 long long num_rays, int n, int training, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (r >= num_rays) return;
    const float *v = values + r * (long long)n * C;
    const float *w = weights + r * n;
    float acc = 0.0f;
    float comp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) comp[c] = 0.0f;
    for (int i = lane; i < n; i += 64) {
        const float wi = w[i];
        acc += wi;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float x = v[(long long)i * C + c];
            if (!training) x = nan_to_num(x);
            comp[c] += mul_rn(wi, x);
        }
    }
    acc = wave_sum(acc);
#pragma unroll
    for (int c = 0; c < C; ++c) comp[c] = wave_sum(comp[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float last = v[(long long)(n - 1) * C + c];
            if (!training) last = nan_to_num(last);
            float o = add_rn(comp[c], mul_rn(last, sub_rn(1.0f, acc)));
            if (!training) o = fminf(fmaxf(o, 0.0f), 1.0f);
            out[r * C + c] = o;
        }
    }
}

// monotone float <-> uint key, so unsigned atomicMin/Max order floats
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void minmax_init_kernel(unsigned *mm) {
    mm[0] = 0xffffffffu;  // running min key
    mm[1] = 0u;           // running max key
}

__global__ void depth_kernel(const float *__restrict__ weights, const float *__restrict__ starts,
                             const float *__restrict__ ends, long long num_rays, int n, float *__restrict__ accum,
                             float *__restrict__ median, float *__restrict__ expected, unsigned *mm) {
    const int lane = threadIdx.x & 63;
    float smin = INFINITY, smax = -INFINITY;  // running over every ray this wave handles
    const long long stride = (long long)gridDim.x * kWavesPerBlock;
    for (long long r = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); r < num_rays; r += stride) {
    const float *w = weights + r * n;
    const float *st = starts + r * n;
    const float *en = ends + r * n;
    float carry = 0.0f, wsum = 0.0f, wsteps = 0.0f;
    int med_idx = n;  // first index with cumsum >= 0.5 (searchsorted side="left")
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool ok = i < n;
        const float wi = ok ? w[i] : 0.0f;
        const float step = ok ? add_rn(st[i], en[i]) / 2.0f : 0.0f;
        const float incl = wave_incl_scan(wi, lane) + carry;
        const unsigned long long hit = __ballot(ok && (incl >= 0.5f));
        if (hit && med_idx == n) med_idx = base + __ffsll((long long)hit) - 1;
        carry = lane_value<63>(incl);
        wsum += wi;
        wsteps += mul_rn(wi, step);
        if (ok) {
            smin = fminf(smin, step);
            smax = fmaxf(smax, step);
        }
    }
    wsum = wave_sum(wsum);
    wsteps = wave_sum(wsteps);
    if (lane == 0) {
        if (accum) accum[r] = wsum;
        if (median) {
            const int idx = min(med_idx, n - 1);
            median[r] = add_rn(st[idx], en[idx]) / 2.0f;
        }
        if (expected) expected[r] = wsteps / add_rn(wsum, 1e-10f);
    }
    }  // ray loop
    if (expected) {
        // ONE atomic pair per BLOCK: atomics on a single address serialise in L2 (~12 ns each) — one pair per ray, or per
        // wave with a ray per wave (a 4096-ray training batch), is 50 us of serialisation in a 10 us kernel
        __shared__ float red[2][kWavesPerBlock];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            smin = fminf(smin, __shfl_xor(smin, o, 64));
            smax = fmaxf(smax, __shfl_xor(smax, o, 64));
        }
        if (lane == 0) {
            red[0][threadIdx.x >> 6] = smin;
            red[1][threadIdx.x >> 6] = smax;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int wv = 1; wv < kWavesPerBlock; ++wv) {
                smin = fminf(smin, red[0][wv]);
                smax = fmaxf(smax, red[1][wv]);
            }
            if (smin <= smax) {
                atomicMin(&mm[0], f2key(smin));
                atomicMax(&mm[1], f2key(smax));
            }
        }
    }
}

__global__ void depth_clip_kernel(float *__restrict__ expected, long long num_rays, const unsigned *mm) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    const float lo = key2f(mm[0]), hi = key2f(mm[1]);
    expected[r] = fminf(fmaxf(expected[r], lo), hi);
}

// NS Cameras.generate_rays for a perspective camera: pixel centres at +0.5,
// d = ((x-cx)/fx, -(y-cy)/fy, -1) rotated by c2w[:3,:3] and normalised, origin = c2w[:3,3],
// pixel_area = |d - d(x+1)| * |d - d(y+1)| (the two neighbouring directions, each normalised).
struct CamArgs {
    float c2w[12];
    float fx, fy, cx, cy;
    int H, W;
};
__device__ __forceinline__ void cam_dir(const CamArgs &c, float u, float v, float &dx, float &dy, float &dz) {
    // sum_j dir_j * R[i][j] in torch's left-to-right order, dir = (u, v, -1)
    const float x = add_rn(add_rn(mul_rn(u, c.c2w[0]), mul_rn(v, c.c2w[1])), mul_rn(-1.0f, c.c2w[2]));
    const float y = add_rn(add_rn(mul_rn(u, c.c2w[4]), mul_rn(v, c.c2w[5])), mul_rn(-1.0f, c.c2w[6]));
    const float z = add_rn(add_rn(mul_rn(u, c.c2w[8]), mul_rn(v, c.c2w[9])), mul_rn(-1.0f, c.c2w[10]));
    const float n = fmaxf(sqrtf(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z))), 1.1920928955078125e-07f);
    dx = x / n;
    dy = y / n;
    dz = z / n;
}
// NS camera_utils.radial_and_tangential_undistort: 10 Newton steps on the OPENCV model (k1..k4, p1, p2), started at the
// distorted point; a step is skipped where the Jacobian determinant is below eps = 1e-3.
struct Distortion {
    float k1, k2, k3, k4, p1, p2;
    int on;
};
__device__ __forceinline__ void undistort(const Distortion &k, float xd, float yd, float &xo, float &yo) {
    float x = xd, y = yd;
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
        const float r = x * x + y * y;
        const float d = 1.0f + r * (k.k1 + r * (k.k2 + r * (k.k3 + r * k.k4)));
        const float fx = d * x + 2.0f * k.p1 * x * y + k.p2 * (r + 2.0f * x * x) - xd;
        const float fy = d * y + 2.0f * k.p2 * x * y + k.p1 * (r + 2.0f * y * y) - yd;
        const float d_r = k.k1 + r * (2.0f * k.k2 + r * (3.0f * k.k3 + r * 4.0f * k.k4));
        const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
        const float fx_x = d + d_x * x + 2.0f * k.p1 * y + 6.0f * k.p2 * x;
        const float fx_y = d_y * x + 2.0f * k.p1 * x + 2.0f * k.p2 * y;
        const float fy_x = d_x * y + 2.0f * k.p2 * y + 2.0f * k.p1 * x;
        const float fy_y = d + d_y * y + 2.0f * k.p2 * x + 6.0f * k.p1 * y;
        const float den = fy_x * fx_y - fx_x * fy_y;
        const bool ok = fabsf(den) > 1e-3f;
        x += ok ? (fx * fy_y - fy * fx_y) / den : 0.0f;
        y += ok ? (fy * fx_x - fx * fy_x) / den : 0.0f;
    }
    xo = x;
    yo = y;
}

__global__ void generate_rays_kernel(CamArgs c, Distortion k, long long first, long long count,
                                     float *__restrict__ origins, float *__restrict__ dirs, float *__restrict__ area) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const long long p = first + i;
    const float px = (float)(p % c.W) + 0.5f, py = (float)(p / c.W) + 0.5f;
    float u = sub_rn(px, c.cx) / c.fx, v = -(sub_rn(py, c.cy) / c.fy);
    float ux = add_rn(sub_rn(px, c.cx), 1.0f) / c.fx;  // (x - cx + 1) / fx, torch's left-to-right order
    float vy = -(add_rn(sub_rn(py, c.cy), 1.0f) / c.fy);
    float d0, d1, d2, a0, a1, a2, b0, b1, b2;
    if (k.on) {  // each of the three coordinate pairs is undistorted on its own, in NS's y-flipped frame
        float ax, ay, bx, by;
        undistort(k, ux, v, ax, ay);
        undistort(k, u, vy, bx, by);
        undistort(k, u, v, u, v);
        cam_dir(c, u, v, d0, d1, d2);
        cam_dir(c, ax, ay, a0, a1, a2);
        cam_dir(c, bx, by, b0, b1, b2);
    } else {
        cam_dir(c, u, v, d0, d1, d2);
        cam_dir(c, ux, v, a0, a1, a2);
        cam_dir(c, u, vy, b0, b1, b2);
    }
    origins[i * 3 + 0] = c.c2w[3];
    origins[i * 3 + 1] = c.c2w[7];
    origins[i * 3 + 2] = c.c2w[11];
    dirs[i * 3 + 0] = d0;
    dirs[i * 3 + 1] = d1;
    dirs[i * 3 + 2] = d2;
    if (area) {
        const float ex = sqrtf((d0 - a0) * (d0 - a0) + (d1 - a1) * (d1 - a1) + (d2 - a2) * (d2 - a2));
        const float ey = sqrtf((d0 - b0) * (d0 - b0) + (d1 - b1) * (d1 - b1) + (d2 - b2) * (d2 - b2));
        area[i] = ex * ey;
    }
}

inline unsigned blocks_for(long long items, int per_block) { return (unsigned)((items + per_block - 1) / per_block); }

}  // namespace

extern "C" {

int tn_generate_rays(const float *c2w_host, float fx, float fy, float cx, float cy, int32_t height, int32_t width,
                     const float *distortion_host, int64_t first_pixel, int64_t num_pixels, float *origins,
                     float *directions, float *pixel_area, void *stream) {
    if (num_pixels == 0) return TN_OK;
    if (!c2w_host || !origins || !directions) return TN_ERR_NULL;
    if (height < 1 || width < 1 || first_pixel < 0 || num_pixels < 0 ||
        first_pixel + num_pixels > (int64_t)height * width || !(fx > 0.0f) || !(fy > 0.0f))
        return TN_ERR_SHAPE;
    CamArgs c;
    for (int i = 0; i < 12; ++i) c.c2w[i] = c2w_host[i];
    c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.H = height; c.W = width;
    Distortion k = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0};
    if (distortion_host) {
        k.k1 = distortion_host[0]; k.k2 = distortion_host[1]; k.k3 = distortion_host[2];
        k.k4 = distortion_host[3]; k.p1 = distortion_host[4]; k.p2 = distortion_host[5];
        for (int i = 0; i < 6; ++i) k.on |= distortion_host[i] != 0.0f;  // NS skips cameras whose parameters are all zero
    }
    hipLaunchKernelGGL(generate_rays_kernel, dim3(blocks_for(num_pixels, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, c, k,
                       (long long)first_pixel, (long long)num_pixels, origins, directions, pixel_area);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_frustum_positions(const float *origins, const float *directions, const float *starts, const float *ends,
                         int64_t num_rays, int32_t n, float *positions, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!origins || !directions || !starts || !ends || !positions) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (num_rays == 0) return TN_OK;
    const long long total = (long long)num_rays * n;
    hipLaunchKernelGGL(frustum_positions_kernel, dim3(blocks_for(total, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, origins, directions, starts, ends, total, n, positions);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_frustum_from_edges(const float *origins, const float *directions, const float *eucl_bins, int64_t num_rays,
                          int32_t n, float *positions, float *starts, float *ends, float *deltas, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!origins || !directions || !eucl_bins || !positions || !starts || !ends || !deltas) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    const long long total = (long long)num_rays * n;
    hipLaunchKernelGGL(frustum_from_edges_kernel, dim3(blocks_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       origins, directions, eucl_bins, total, n, positions, starts, ends, deltas);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_sample_initial(const float *lin_bins, const float *t_rand, const float *nears, const float *fars,
                      int64_t num_rays, int32_t n, int32_t uniform_spacing, float *spacing_bins, float *eucl_bins,
                      void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!lin_bins || !nears || !fars || !spacing_bins || !eucl_bins) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (num_rays == 0) return TN_OK;
    const long long total = (long long)num_rays * (n + 1);
    hipLaunchKernelGGL(sample_initial_kernel, dim3(blocks_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       lin_bins, t_rand, nears, fars, (long long)num_rays, n, (uniform_spacing & 1) != 0, (uniform_spacing & 2) != 0,
                       spacing_bins, eucl_bins);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_weights_fwd(const float *deltas, const float *densities, int64_t num_rays, int32_t n, float *weights,
                   void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!deltas || !densities || !weights) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (num_rays == 0) return TN_OK;
    hipLaunchKernelGGL(weights_kernel, dim3(blocks_for(num_rays, kWavesPerBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, deltas, densities, (long long)num_rays, n, weights);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_sample_pdf(const float *weights, const float *existing_bins, const float *u, const float *u_rand,
                  const float *nears, const float *fars, int64_t num_rays, int32_t n_in, int32_t n_out,
                  int32_t uniform_spacing, float *spacing_bins, float *eucl_bins, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!weights || !existing_bins || !u || !nears || !fars || !spacing_bins || !eucl_bins) return TN_ERR_NULL;
    if (num_rays < 0 || n_in < 1 || n_in > kMaxPdfIn || n_out < 1) return TN_ERR_SHAPE;
    if (num_rays == 0) return TN_OK;
    const size_t smem = (size_t)kWavesPerBlock * 2 * (n_in + 1) * sizeof(float);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(blocks_for(num_rays, kWavesPerBlock)), dim3(kBlock), smem,
                       (hipStream_t)stream, weights, existing_bins, u, u_rand, nears, fars, (long long)num_rays, n_in,
                       n_out, (uniform_spacing & 1) != 0, (uniform_spacing & 2) != 0, spacing_bins, eucl_bins);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_composite_fwd(const float *values, const float *weights, int64_t num_rays, int32_t n, int32_t channels,
                     int32_t training, float *out, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!values || !weights || !out) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (channels != 1 && channels != 3 && channels != 4) return TN_ERR_UNSUPPORTED;
    if (num_rays == 0) return TN_OK;
    const dim3 grid(blocks_for(num_rays, kWavesPerBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (channels == 1)
        hipLaunchKernelGGL(composite_kernel<1>, grid, block, 0, s, values, weights, (long long)num_rays, n, training, out);
    else if (channels == 3)
        hipLaunchKernelGGL(composite_kernel<3>, grid, block, 0, s, values, weights, (long long)num_rays, n, training, out);
    else
        hipLaunchKernelGGL(composite_kernel<4>, grid, block, 0, s, values, weights, (long long)num_rays, n, training, out);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_depth_fwd(const float *weights, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                 float *accumulation, float *median, float *expected, float *minmax_scratch, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!weights || !starts || !ends) return TN_ERR_NULL;
    if (expected && !minmax_scratch) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    if (num_rays == 0) return TN_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned *mm = reinterpret_cast<unsigned *>(minmax_scratch);
    if (expected) hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
    const unsigned depth_blocks = blocks_for(num_rays, kWavesPerBlock);
    hipLaunchKernelGGL(depth_kernel, dim3(depth_blocks < 512u ? depth_blocks : 512u), dim3(kBlock), 0, s, weights, starts,
                       ends, (long long)num_rays, n, accumulation, median, expected, mm);
    if (expected)
        hipLaunchKernelGGL(depth_clip_kernel, dim3(blocks_for(num_rays, kBlock)), dim3(kBlock), 0, s, expected,
                           (long long)num_rays, mm);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
