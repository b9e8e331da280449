// Image metrics of the eval harness: what ThermalNerfModel.get_image_metrics_and_images asks of torchmetrics
// on a rendered frame [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:362-363; NS NerfactoModel.get_image_metrics_and_images].
//
//   tn_ssim_fwd   torchmetrics.functional.structural_similarity_index_measure (1.7.2, uv.lock:5468-5469) with its defaults:
//                 gaussian window 11 x 11, sigma 1.5, k1 0.01, k2 0.03; variances clamped at 0; the image is reflect-padded by
//                 5 and the result cropped by 5 again, i.e. the mean runs over the windows that lie INSIDE the image.
//
// One block = a 32 x 8 tile of window centres of one channel: the (32+10) x (8+10) input patch of both images goes to LDS, a
// horizontal 11-tap pass leaves the five filtered rows (p, t, p^2, t^2, p t) in LDS, the vertical pass finishes the window
// sums, forms the index and the block adds its 256 values; a second single-block kernel sums the blocks' partials in a fixed
// order (deterministic) and divides by the count.  Images are [H, W, C] as the renderers produce them (no transposes).
#include "tn_device.h"

using namespace tn;

namespace {

constexpr int kBlock = 256;
constexpr int kTW = 32, kTH = 8, kTaps = 11, kPad = kTaps / 2;
constexpr int kPW = kTW + 2 * kPad, kPH = kTH + 2 * kPad;  // 42 x 18 input patch

struct Taps {
    float g[kTaps];
};

__global__ void __launch_bounds__(kBlock)
ssim_tile_kernel(const float *__restrict__ pred, const float *__restrict__ target, int H, int W, int C, Taps taps, float c1,
                 float c2, float *__restrict__ partials) {
    __shared__ float ps[kPH][kPW], ts[kPH][kPW];
    __shared__ float hs[5][kPH][kTW];
    __shared__ float red[kBlock / 64];
    const int c = blockIdx.z;
    const int ox = blockIdx.x * kTW, oy = blockIdx.y * kTH;  // window-centre coordinates minus kPad = patch origin in the image
    const int vw = W - 2 * kPad, vh = H - 2 * kPad;          // valid centres: [0, vw) x [0, vh) in these coordinates
    for (int e = threadIdx.x; e < kPH * kPW; e += kBlock) {
        const int py = e / kPW, px = e - py * kPW;
        const int y = oy + py, x = ox + px;
        float p = 0.0f, t = 0.0f;
        if (y < H && x < W) {
            const size_t k = ((size_t)y * W + x) * C + c;
            p = pred[k];
            t = target[k];
        }
        ps[py][px] = p;
        ts[py][px] = t;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kPH * kTW; e += kBlock) {
        const int py = e / kTW, x = e - py * kTW;
        float sp = 0.0f, st = 0.0f, spp = 0.0f, stt = 0.0f, spt = 0.0f;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            const float w = taps.g[k], p = ps[py][x + k], t = ts[py][x + k];
            sp = fmaf(w, p, sp);
            st = fmaf(w, t, st);
            spp = fmaf(w, p * p, spp);
            stt = fmaf(w, t * t, stt);
            spt = fmaf(w, p * t, spt);
        }
        hs[0][py][x] = sp; hs[1][py][x] = st; hs[2][py][x] = spp; hs[3][py][x] = stt; hs[4][py][x] = spt;
    }
    __syncthreads();
    const int x = threadIdx.x & (kTW - 1), y = threadIdx.x / kTW;
    float v[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
        const float w = taps.g[k];
#pragma unroll
        for (int m = 0; m < 5; ++m) v[m] = fmaf(w, hs[m][y + k][x], v[m]);
    }
    float s = 0.0f;
    if (ox + x < vw && oy + y < vh) {
        const float mu_pp = v[0] * v[0], mu_tt = v[1] * v[1], mu_pt = v[0] * v[1];
        const float sig_p = fmaxf(v[2] - mu_pp, 0.0f), sig_t = fmaxf(v[3] - mu_tt, 0.0f), sig_pt = v[4] - mu_pt;
        const float upper = 2.0f * sig_pt + c2, lower = sig_p + sig_t + c2;
        s = ((2.0f * mu_pt + c1) * upper) / ((mu_pp + mu_tt + c1) * lower);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partials[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kBlock) ssim_reduce_kernel(const float *__restrict__ partials, long long n, double inv_count,
                                                             float *__restrict__ out) {
    __shared__ double red[kBlock];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += kBlock) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_count);
}

inline long long tiles_of(int H, int W, int C) {
    const int vw = W - 2 * kPad, vh = H - 2 * kPad;
    return (long long)((vw + kTW - 1) / kTW) * ((vh + kTH - 1) / kTH) * C;
}

}  // namespace

extern "C" {

size_t tn_ssim_workspace_bytes(int32_t height, int32_t width, int32_t channels) {
    if (height < kTaps || width < kTaps || channels < 1) return 0;
    return (size_t)tiles_of(height, width, channels) * sizeof(float);
}

int tn_ssim_fwd(const float *pred, const float *target, int32_t height, int32_t width, int32_t channels, float data_range,
                void *workspace, size_t workspace_bytes, float *out, void *stream) {
    if (!pred || !target || !out || !workspace) return TN_ERR_NULL;
    // torchmetrics reflect-pads by 5, which torch refuses for images narrower than 6; a window must fit: 11 x 11 at least
    if (height < kTaps || width < kTaps || channels < 1 || channels > 65535) return TN_ERR_SHAPE;
    if (workspace_bytes < tn_ssim_workspace_bytes(height, width, channels)) return TN_ERR_WORKSPACE;
    // torchmetrics _gaussian: dist = arange((1 - k) / 2, (1 + k) / 2), exp(-(dist / sigma)^2 / 2), normalised — in float32
    Taps taps;
    float sum = 0.0f;
    for (int k = 0; k < kTaps; ++k) {
        const float d = (float)(k - kPad) / 1.5f;
        taps.g[k] = expf(-(d * d) / 2.0f);
        sum += taps.g[k];
    }
    for (int k = 0; k < kTaps; ++k) taps.g[k] /= sum;
    const float c1 = (0.01f * data_range) * (0.01f * data_range), c2 = (0.03f * data_range) * (0.03f * data_range);
    const int vw = width - 2 * kPad, vh = height - 2 * kPad;
    const dim3 grid((vw + kTW - 1) / kTW, (vh + kTH - 1) / kTH, channels);
    if (grid.y > 65535) return TN_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    float *partials = reinterpret_cast<float *>(workspace);
    hipLaunchKernelGGL(ssim_tile_kernel, grid, dim3(kBlock), 0, s, pred, target, height, width, channels, taps, c1, c2, partials);
    TN_LAUNCH_CHECK();
    const long long n = (long long)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(ssim_reduce_kernel, dim3(1), dim3(kBlock), 0, s, partials, n, 1.0 / ((double)vw * vh * channels), out);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
