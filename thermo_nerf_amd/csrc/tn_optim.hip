// Adam for the training step: the optimizer the reference's method config names for every parameter group —
// AdamOptimizerConfig(lr 1e-2, eps 1e-15) [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:31-44] = torch.optim.Adam:
//
//     g  = grad + weight_decay * p                     (L2 form, not AdamW)
//     m  = m + (1 - beta1) (g - m)                     (torch: exp_avg.lerp_(grad, 1 - beta1))
//     v  = beta2 v + (1 - beta2) g g
//     p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// HBM streaming: 16 B read + 12 B written per element, no reuse.  One launch covers a LIST of tensors (up to TN_ADAM_MAX_TENSORS
// descriptors passed by value: no descriptor upload, no per-step host-to-device copy): the ~26 small tensors of a step (MLP layers,
// embedding, pose adjustments: 20 k floats) are one launch instead of torch's per-group multi-tensor pair, and the field's 64 MB
// table is its own call so that the caller can queue it on the stream its gradient's scatter ran on (thermo_nerf_amd/optim.py).
// Every tensor carries its own scalars: groups differ in lr / eps / weight decay, tensors in their step count (the proposal
// networks receive gradient on one step in six).
#include "tn_device.h"

using namespace tn;

namespace {

constexpr int kBlock = 256;
constexpr int kPerThread = 4;                      // one float4 per array and thread
constexpr int kPerBlock = kBlock * kPerThread;     // 1024 elements
constexpr int kMaxBlocksPerTensor = 256 * 16;      // grid-stride beyond that: 16 blocks per CU keep 8 TB/s busy

struct AdamList {
    tn_adam_tensor t[TN_ADAM_MAX_TENSORS];
    int32_t first_block[TN_ADAM_MAX_TENSORS + 1];  // tensor k owns blocks [first_block[k], first_block[k+1])
    int32_t count;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const tn_adam_tensor &t) {
    if (t.weight_decay != 0.0f) g = __fadd_rn(g, __fmul_rn(t.weight_decay, p));
    m = __fadd_rn(m, __fmul_rn(__fsub_rn(g, m), t.one_minus_beta1));
    v = __fadd_rn(__fmul_rn(t.beta2, v), __fmul_rn(__fmul_rn(t.one_minus_beta2, g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), t.bias_correction2_sqrt), t.eps);
    p = __fsub_rn(p, __fmul_rn(t.step_size, __fdiv_rn(m, denom)));
}

__global__ void __launch_bounds__(kBlock) adam_kernel(AdamList L) {
    // which tensor: a linear walk over <= 48 block offsets held in SGPRs (uniform per block)
    int k = 0;
    while (k + 1 < L.count && (int)blockIdx.x >= L.first_block[k + 1]) ++k;
    const tn_adam_tensor t = L.t[k];
    const int nb = L.first_block[k + 1] - L.first_block[k];
    const int b = blockIdx.x - L.first_block[k];
    const int64_t n = t.n;
    float *__restrict__ p = t.param;
    const float *__restrict__ g = t.grad;
    float *__restrict__ m = t.exp_avg;
    float *__restrict__ v = t.exp_avg_sq;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    for (int64_t base = (int64_t)b * kPerBlock; base < n; base += (int64_t)nb * kPerBlock) {
        const int64_t i = base + (int64_t)threadIdx.x * kPerThread;
        if (vec && i + kPerThread <= n) {
            float4 pp = *reinterpret_cast<const float4 *>(p + i);
            const float4 gg = *reinterpret_cast<const float4 *>(g + i);
            float4 mm = *reinterpret_cast<const float4 *>(m + i);
            float4 vv = *reinterpret_cast<const float4 *>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, t);
            adam_one(pp.y, gg.y, mm.y, vv.y, t);
            adam_one(pp.z, gg.z, mm.z, vv.z, t);
            adam_one(pp.w, gg.w, mm.w, vv.w, t);
            *reinterpret_cast<float4 *>(p + i) = pp;
            *reinterpret_cast<float4 *>(m + i) = mm;
            *reinterpret_cast<float4 *>(v + i) = vv;
        } else {
            for (int e = 0; e < kPerThread; ++e) {
                const int64_t j = i + e;
                if (j < n) {
                    float pj = p[j], mj = m[j], vj = v[j];
                    adam_one(pj, g[j], mj, vj, t);
                    p[j] = pj;
                    m[j] = mj;
                    v[j] = vj;
                }
            }
        }
    }
}

}  // namespace

extern "C" int tn_adam_step(const tn_adam_tensor *tensors, int32_t count, void *stream) {
    if (count == 0) return TN_OK;
    if (!tensors) return TN_ERR_NULL;
    if (count < 0 || count > TN_ADAM_MAX_TENSORS) return TN_ERR_SHAPE;
    AdamList L;
    int blocks = 0, used = 0;
    for (int k = 0; k < count; ++k) {
        const tn_adam_tensor &t = tensors[k];
        if (t.n < 0) return TN_ERR_SHAPE;
        if (t.n == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return TN_ERR_NULL;
        L.t[used] = t;
        L.first_block[used] = blocks;
        const int64_t nb = (t.n + kPerBlock - 1) / kPerBlock;
        blocks += (int)(nb < kMaxBlocksPerTensor ? nb : kMaxBlocksPerTensor);
        ++used;
    }
    if (used == 0) return TN_OK;
    L.first_block[used] = blocks;
    L.count = used;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, L);
    TN_LAUNCH_CHECK();
    return TN_OK;
}
