// Fused forward of Model.forward (collider output in) + ThermalNerfModel.get_outputs
// [REF thermo_nerf/thermal_nerf/thermal_nerf_model.py:210-275], one wave64 per ray.
//
//   (the fused kernels use the FAST arithmetic flavour of tn_device.h: v_rcp/v_exp reciprocal, quotient, exp and the
//    fma-form interpolation, <= 2 ulp from the torch op order the plugin-surface kernels keep)
//   proposal_kernel   NS ProposalNetworkSampler.generate_ray_samples, both proposal levels fused:
//                     piecewise bins -> prop net 0 -> weights -> PDF -> prop net 1 -> weights -> PDF -> S+1 bins
//                     (SURVEY §8a a4,a5,a6,a11) + the two prop_depth_i medians [REF :267-270].  Per-ray state
//                     (bins, weights, cdf) lives in LDS; nothing but the final S+1 bin edges goes back to HBM.
//   main_valu_kernel  ThermalNerfactoTField.forward [REF thermal_field.py:183-201] on the S final samples +
//                     get_weights + RGB/thermal/accumulation/depth renderers [REF :233-243,271-273], lane per
//                     sample, MLPs on the VALU with weights in LDS (reference form of the fused kernel).
//   main_mfma_kernel  same contract, MLPs on the f32 MFMA pipe (tn_render_mfma.hip).
#include "tn_field_eval.h"

#include <type_traits>

#ifndef TN_PROP_WAVES
#define TN_PROP_WAVES 4   // waves per SIMD of proposal_rays_kernel (= workgroups per CU)
#endif
#ifndef TN_RESAMPLE_BLOCK
#define TN_RESAMPLE_BLOCK 16  // bins per round trip in proposal_resample_kernel's scan and PDF walk (8 / 16 / 32: 0.444 / 0.435 / 0.435 ms at 80 000 rays)
#endif
#ifndef TN_PROP_SPLIT
#define TN_PROP_SPLIT 1  // calls under 3 072 tiles: the lane = ray proposal pass as density segments + per-tile resampling (same bits)
#endif
// s_setprio of a proposal wave while it computes a level group's indices and issues its gathers (0 = none): four waves share a
// SIMD, and the one whose memory requests can go out goes ahead of the others' interpolation / MLP arithmetic —
// proposal_rays_kernel 2.93 -> 2.89 ms per 640 k rays at S=192, 2.82 -> 2.76 at S=64 (round 4, A/B both orders; same bits).
// the same for the PDF walks' block loads of weights and edges: 2.88 -> 2.865 ms at S=192, 2.76 -> 2.735 at S=64 (gather priority 2
// beside it: worse)
#ifndef TN_PDF_LOAD_PRIO
#define TN_PDF_LOAD_PRIO 1
#endif
#ifndef TN_PROP_GATHER_PRIO
#define TN_PROP_GATHER_PRIO 1
#endif
using namespace tn;

namespace tn {
// implemented in tn_render_mfma.hip
int launch_main_mfma(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                     const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                     hipStream_t stream, int split, float *seg_scratch);
// implemented in tn_render_h3.hip (the two split-precision policies of one kernel)
int launch_main_b6(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                   const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                   hipStream_t stream);
int launch_main_h3(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                   const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                   hipStream_t stream);
}

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / TN_WAVE;
constexpr int PH = 16;  // proposal-net hidden width (proposal_net_args_list: hidden_dim 16)
constexpr size_t kWsMid = 2048;      // workspace bytes between the final edges and the proposal scratch
constexpr int kPropWFloats = 208;    // per net: W0 k-major [10][16] | b0 [16] | w1 [16] | b1 [1] (+ pad)

struct PropNet {
    Grid g;
    tn_space space;
    const float *w0, *b0, *w1, *b1;
    float avg;
};

struct PropArgs {
    PropNet net[2];
    const float *origins, *dirs, *nears, *fars;
    const float *lin0, *u1, *u2, *jitter;
    int jper;  // 0: jitter [3,R] (single_jitter); 1: [R,P0+1] | [R,P1+1] | [R,S+1] (NS single_jitter=False)
    const float *wk;  // [2][kPropWFloats] k-major copies of the two proposal MLPs (workspace), scalar-operand evaluation
    long long R;
    int P0, P1, S, training;
    int lin;                 // initial sampler: 0 piecewise (UniformLinDispPiecewise), 1 UniformSampler
    float anneal;
    float *ws_spacing;       // final spacing bins, ray-tiled (tn_ws_bin), always written
    float *out_spacing[3];   // optional
    float *out_eucl[3];      // optional
    float *out_w[2];         // optional [R,P0], [R,P1]
    float *prop_depth[2];    // optional
};

// one proposal level for one ray (one wave): density -> weights (left in wts[]) -> median depth
__device__ __forceinline__ float prop_level(const PropNet &net, const TwoLayerLds &w, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float s_near, float s_far, bool lin,
                                            const float *bins, int n, float *wts, int lane) {
    const Space sp = make_space(net.space);
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) {
            const float st = spacing_to_eucl<true>(bins[i], s_near, s_far, lin);
            const float en = spacing_to_eucl<true>(bins[i + 1], s_near, s_far, lin);
            float px, py, pz;
            const float sel = normalize_position<true>(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                 frustum_pos(oz, dz, st, en), px, py, pz);
            // proposal_net_args_list uses num_levels 5: unrolled form keeps all 40 gathers of a sample in flight
            const float dens = (net.g.num_levels == 5)
                                   ? proposal_density_scalar<PH, 5, true>(net.g, as_scalar(net.w0), as_scalar(net.b0),
                                                                          as_scalar(net.w1), as_scalar(net.b1), net.avg, px, py, pz, sel)
                                   : proposal_density_eval<PH, 0, true>(net.g, w, net.avg, px, py, pz, sel);
            wts[i] = mul_rn(sub_rn(en, st), dens);
        }
    }
    float carry = 0.0f, carry_w = 0.0f;
    int med_idx = n;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool ok = i < n;
        const float dd = ok ? wts[i] : 0.0f;
        const float incl = wave_incl_scan(dd, lane);
        const float excl = carry + wave_excl_from_incl(incl, lane);
        const float wi = ok ? nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-excl))) : 0.0f;
        carry += lane_value<63>(incl);
        const float incl_w = wave_incl_scan(wi, lane) + carry_w;
        const unsigned long long hit = __ballot(ok && (incl_w >= 0.5f));
        if (hit && med_idx == n) med_idx = base + __ffsll((long long)hit) - 1;
        carry_w = lane_value<63>(incl_w);
        if (ok) wts[i] = wi;
    }
    const int idx = min(med_idx, n - 1);
    const float st = spacing_to_eucl<true>(bins[idx], s_near, s_far, lin);
    const float en = spacing_to_eucl<true>(bins[idx + 1], s_near, s_far, lin);
    return add_rn(st, en) / 2.0f;
}

// NS PDFSampler: wts[n_in] (weights), bins[n_in+1] (existing spacing bins) -> new_bins[n_out+1]
__device__ __forceinline__ void pdf_resample(const float *wts, const float *bins, int n_in, float *cdf,
                                             const float *u, bool jittered, const float *jrow, bool jper, float anneal, int n_out,
                                             float *new_bins, int lane) {
    float part = 0.0f;
    for (int i = lane; i < n_in; i += 64) {
        const float wa = (anneal == 1.0f) ? wts[i] : powf(wts[i], anneal);
        part += add_rn(wa, 0.01f);
    }
    float ws = wave_sum(part);
    const float padding = fmaxf(sub_rn(1e-5f, ws), 0.0f);
    const float pad_each = padding / (float)n_in;
    ws = add_rn(ws, padding);
    float carry = 0.0f;
    if (lane == 0) cdf[0] = 0.0f;
    for (int base = 0; base < n_in; base += 64) {
        const int i = base + lane;
        float pdf = 0.0f;
        if (i < n_in) {
            const float wa = (anneal == 1.0f) ? wts[i] : powf(wts[i], anneal);
            pdf = t_div<true>(add_rn(add_rn(wa, 0.01f), pad_each), ws);
        }
        const float incl = wave_incl_scan(pdf, lane) + carry;
        if (i < n_in) cdf[i + 1] = fminf(1.0f, incl);
        carry = lane_value<63>(incl);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nb = n_out + 1;
    const float jit = (jittered && !jper) ? jrow[0] / (float)nb : 0.0f;
    for (int j = lane; j < nb; j += 64) {
        const float uu = jittered ? add_rn(u[j], jper ? jrow[j] / (float)nb : jit) : u[j];
        int lo = 0, hi = n_in + 1;
        while (lo < hi) {  // searchsorted(cdf, uu, side="right")
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int below = min(max(lo - 1, 0), n_in);
        const int above = min(max(lo, 0), n_in);
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = bins[below], b1 = bins[above];
        float t = nan_to_num(t_div<true>(sub_rn(uu, c0), sub_rn(c1, c0)));
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        new_bins[j] = add_rn(b0, mul_rn(t, sub_rn(b1, b0)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void store_bins(const float *bins, int nb, float s_near, float s_far, bool lin, float *spacing,
                                           float *eucl, long long r, int lane) {
    if (spacing)
        for (int j = lane; j < nb; j += 64) spacing[r * nb + j] = bins[j];
    if (eucl)
        for (int j = lane; j < nb; j += 64) eucl[r * nb + j] = spacing_to_eucl<true>(bins[j], s_near, s_far, lin);
}

// One ray per wave is a chain of dependent round trips (4 + 2 chunks of gathers, scans, two binary-search resamples): the
// kernel is latency-bound and wants every ray of a training batch resident at once.  At the 176 VGPRs hipcc takes when left
// alone, 2 waves fit a SIMD and a 4096-ray batch runs as two rounds (101 us); capped at 128 (42 spilled dwords, L1-resident
// scratch) all 4096 waves are resident: 85 us.
__global__ void __launch_bounds__(kBlock, 4) proposal_kernel(PropArgs a, int nmax, int nbmax) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int in0 = 2 * a.net[0].g.num_levels, in1 = 2 * a.net[1].g.num_levels;
    float *wbase0 = smem;
    float *wbase1 = wbase0 + two_layer_floats(in0, PH, 1);
    float *per_wave = wbase1 + two_layer_floats(in1, PH, 1);
    const TwoLayerLds w0 = stage_two_layer<PH>(wbase0, a.net[0].w0, a.net[0].b0, a.net[0].w1, a.net[0].b1, in0, 1);
    const TwoLayerLds w1 = stage_two_layer<PH>(wbase1, a.net[1].w0, a.net[1].b0, a.net[1].w1, a.net[1].b1, in1, 1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool lin = a.lin != 0;
    float *binsA = per_wave + (size_t)wave * (3 * nbmax + nmax);
    float *binsB = binsA + nbmax;
    float *cdf = binsB + nbmax;
    float *wts = cdf + nbmax;
    const long long stride = (long long)gridDim.x * kWaves;
    for (long long r = (long long)blockIdx.x * kWaves + wave; r < a.R; r += stride) {
        const float ox = a.origins[r * 3], oy = a.origins[r * 3 + 1], oz = a.origins[r * 3 + 2];
        const float dx = a.dirs[r * 3], dy = a.dirs[r * 3 + 1], dz = a.dirs[r * 3 + 2];
        const float s_near = spacing_fn(a.nears[r], lin), s_far = spacing_fn(a.fars[r], lin);
        // level 0 bins: linspace (+ single stratified jitter in training), SURVEY A.7
        const int P0 = a.P0, P1 = a.P1, S = a.S;
        for (int j = lane; j <= P0; j += 64) {
            float b = a.lin0[j];
            if (a.jitter) {
                const float t = a.jper ? a.jitter[r * (P0 + 1) + j] : a.jitter[r];
                const float lo = (j == 0) ? a.lin0[0] : add_rn(a.lin0[j], a.lin0[j - 1]) / 2.0f;
                const float hi = (j == P0) ? a.lin0[P0] : add_rn(a.lin0[j + 1], a.lin0[j]) / 2.0f;
                b = add_rn(lo, mul_rn(sub_rn(hi, lo), t));
            }
            binsA[j] = b;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        store_bins(binsA, P0 + 1, s_near, s_far, lin, a.out_spacing[0], a.out_eucl[0], r, lane);
        const float med0 = prop_level(a.net[0], w0, ox, oy, oz, dx, dy, dz, s_near, s_far, lin, binsA, P0, wts, lane);
        if (a.out_w[0]) for (int i = lane; i < P0; i += 64) a.out_w[0][r * P0 + i] = wts[i];
        const float *j1 = !a.jitter ? nullptr : (a.jper ? a.jitter + a.R * (P0 + 1) + r * (P1 + 1) : a.jitter + a.R + r);
        const float *j2 = !a.jitter ? nullptr : (a.jper ? a.jitter + a.R * (P0 + 1 + P1 + 1) + r * (S + 1) : a.jitter + 2 * a.R + r);
        pdf_resample(wts, binsA, P0, cdf, a.u1, a.jitter != nullptr, j1, a.jper != 0, a.anneal, P1, binsB, lane);
        store_bins(binsB, P1 + 1, s_near, s_far, lin, a.out_spacing[1], a.out_eucl[1], r, lane);
        const float med1 = prop_level(a.net[1], w1, ox, oy, oz, dx, dy, dz, s_near, s_far, lin, binsB, P1, wts, lane);
        if (a.out_w[1]) for (int i = lane; i < P1; i += 64) a.out_w[1][r * P1 + i] = wts[i];
        pdf_resample(wts, binsB, P1, cdf, a.u2, a.jitter != nullptr, j2, a.jper != 0, a.anneal, S, binsA, lane);
        for (int j = lane; j <= S; j += 64) a.ws_spacing[tn_ws_bin(r, j, S)] = binsA[j];
        store_bins(binsA, S + 1, s_near, s_far, lin, nullptr, a.out_eucl[2], r, lane);
        if (a.out_spacing[2]) for (int j = lane; j <= S; j += 64) a.out_spacing[2][r * (S + 1) + j] = binsA[j];
        if (lane == 0) {
            if (a.prop_depth[0]) a.prop_depth[0][r] = med0;
            if (a.prop_depth[1]) a.prop_depth[1][r] = med1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}


// ------------------------------------------------------------------------------------------------------
// k-major copy of a 10 -> 16 -> 1 proposal MLP for the scalar-operand evaluation below: W0t[k][h] = W0[h][k], so that the
// two weights of a packed FMA (hidden units 2j, 2j+1 of one input feature) are adjacent dwords of one s_load.
// Block 0 also resets the call's depth min/max words (ordered keys): the field kernel of the same call accumulates into them.
__global__ void prop_weights_kmajor_kernel(PropNet n0, PropNet n1, float *__restrict__ dst, unsigned *__restrict__ mm) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        mm[0] = 0xffffffffu;
        mm[1] = 0u;
    }
    const PropNet &n = blockIdx.x == 0 ? n0 : n1;
    float *d = dst + blockIdx.x * kPropWFloats;
    const int in = 2 * n.g.num_levels;
    for (int e = threadIdx.x; e < kPropWFloats; e += blockDim.x) {
        float v = 0.0f;
        if (in == 10) {
            if (e < 160) v = n.w0[(e & 15) * 10 + (e >> 4)];
            else if (e < 176) v = n.b0[e - 160];
            else if (e < 192) v = n.w1[e - 176];
            else if (e == 192) v = n.b1[0];
        }
        d[e] = v;
    }
}

// density of the 5-level / 16-hidden proposal net with the weights as SCALAR operands in k-major order: per input feature
// eight packed FMAs (v_pk_fma_f32, SGPR-pair source) update the 16 hidden units.  Per hidden unit the accumulation order is
// bias, then the features in level order — the order of hidden_from_grid: bit-identical results.
// ND = levels read from the dense re-layout: a compile-time split (-1: by the grid's run-time count, level by level), so that
// all of a sample's gathers are issued together and waited for with exact counts.
template <bool FAST, int ND = -1>
__device__ __forceinline__ float proposal_density_kmajor(const Grid &g, const tn_cfloat *wk, float avg, float px, float py,
                                                         float pz, float sel) {
    float2 f[5];
    if (ND >= 0) {
        if (FAST) {
            // Explicit stages per group of levels (index arithmetic | gathers | interpolation), 3 + 2 levels: left alone hipcc
            // issues one level's gathers, waits and interpolates before it touches the next level — four memory round
            // trips per sample where two do (3.44 -> 3.03 ms per 640 k rays; 2 + 3: 3.04, 4 + 1: 3.14; the second group's
            // gathers issued ahead of the first group's interpolation: 2.94 with 20 spilled registers — not kept).  112 VGPRs,
            // no spills.
            auto group = [&](auto L0, auto L1) {
                constexpr int l0 = decltype(L0)::value, l1 = decltype(L1)::value;
                HashTaps t[l1 - l0];
                float2 fv[l1 - l0][8];
#if TN_PROP_GATHER_PRIO
                __builtin_amdgcn_s_setprio(TN_PROP_GATHER_PRIO);
#endif
#pragma unroll
                for (int l = l0; l < l1; ++l) {
                    if (l < ND) dense_taps(g, l, px, py, pz, t[l - l0]); else hash_taps(g, l, px, py, pz, t[l - l0]);
                }
                TN_STAGE_FENCE();
#pragma unroll
                for (int l = l0; l < l1; ++l) {
                    if (l < ND) dense_gather(g, l, t[l - l0], fv[l - l0]); else hash_gather(g, l, t[l - l0], fv[l - l0]);
                }
#pragma unroll
                for (int l = l0; l < l1; ++l) hash_hold(t[l - l0]);
#if TN_PROP_GATHER_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                TN_STAGE_FENCE();
#pragma unroll
                for (int l = l0; l < l1; ++l) f[l] = hash_blend(t[l - l0], fv[l - l0]);
                TN_STAGE_FENCE();
            };
            group(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 3>{}, std::integral_constant<int, 5>{});
        } else {
#pragma unroll
            for (int l = 0; l < 5; ++l) f[l] = (l < ND) ? encode_level<true, FAST>(g, l, px, py, pz) : encode_level<false, FAST>(g, l, px, py, pz);
        }
    } else if (g.num_dense == 0) {
#pragma unroll
        for (int l = 0; l < 5; ++l) f[l] = encode_level<false, FAST>(g, l, px, py, pz);
    } else {
#pragma unroll
        for (int l = 0; l < 5; ++l) f[l] = encode_level_any<FAST>(g, l, px, py, pz);
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(4))) const f32x2 cf32x2;
    const cf32x2 *wk2 = (const cf32x2 *)wk;  // 8-byte aligned: the copy starts 256 B into the workspace's mid region
    f32x2 hid[PH / 2];
#pragma unroll
    for (int j = 0; j < PH / 2; ++j) hid[j] = wk2[80 + j];
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const f32x2 fx = {f[l].x, f[l].x}, fy = {f[l].y, f[l].y};
#pragma unroll
        for (int j = 0; j < PH / 2; ++j) hid[j] = __builtin_elementwise_fma(wk2[(2 * l) * (PH / 2) + j], fx, hid[j]);
#pragma unroll
        for (int j = 0; j < PH / 2; ++j) hid[j] = __builtin_elementwise_fma(wk2[(2 * l + 1) * (PH / 2) + j], fy, hid[j]);
    }
    float o = wk[192];
#pragma unroll
    for (int j = 0; j < PH / 2; ++j) {
        o = fmaf(wk[176 + 2 * j], fmaxf(hid[j].x, 0.0f), o);
        o = fmaf(wk[176 + 2 * j + 1], fmaxf(hid[j].y, 0.0f), o);
    }
    return mul_rn(mul_rn(avg, t_exp<FAST>(o)), sel);
}

// proposal_rays_kernel: the same two fused proposal levels with lane = RAY (one wave64 owns 64 consecutive rays and
// walks their samples in lock-step), like the field kernel:
//   * adjacent rays at one sample index share hash-grid cells -> coherent gathers;
//   * get_weights / median / sums become per-lane running values: no wave scans, no LDS;
//   * PDFSampler's searchsorted + gathers become a per-lane MERGE WALK over the (sorted) cdf and the (sorted)
//     sample positions u: no cdf array, no binary search;
//   * exactly P0 + P1 density evaluations per ray (the wave-per-ray form pads 96 samples to 128 lanes).
// Per-ray arrays that must survive between the passes (the level's weights, the 97 resampled edges) live in a
// ray-tiled global scratch [tile][index][64 lanes] (coalesced, L2-resident), part of the caller's workspace.
// ------------------------------------------------------------------------------------------------------
struct PropRaysArgs {
    PropArgs p;
    float *w_scratch;   // [tiles][nmax][64]
    float *b1_scratch;  // [tiles][P1+1][64]
    int nmax;
};

// PDFSampler (histogram_padding 0.01, eps 1e-5) for one lane: inputs w[i] (scratch), existing edges via `edge(i)`,
// total = sum(w^anneal + 0.01) over the level; emits n_out+1 new edges through `emit(j, value)`.
#ifndef TN_PDF_WALK_BLOCK
#define TN_PDF_WALK_BLOCK 8
#endif
template <int WB = TN_PDF_WALK_BLOCK, typename EdgeFn, typename EmitFn>
__device__ __forceinline__ void pdf_walk(const float *w, int n_in, float total, float anneal, const float *u, bool jittered,
                                         float jit, int n_out, EdgeFn edge, EmitFn emit, const float *jrow = nullptr) {
    // jrow (NS single_jitter=False): this ray's n_out + 1 draws; jit is then unused
    const float padding = fmaxf(sub_rn(1e-5f, total), 0.0f);
    const float pad_each = padding / (float)n_in;
    const float ws = add_rn(total, padding);
    const float rws = __builtin_amdgcn_rcpf(ws);
    const int nb = n_out + 1;
    int j = 0;
    float run = 0.0f, c0 = 0.0f;          // running cumsum(pdf); cdf[i] = min(1, run) before adding bin i
    float b0 = edge(0);
    float uj = jittered ? add_rn(u[0], jrow ? jrow[0] / (float)(n_out + 1) : jit) : u[0];
    // WB (TN_PDF_WALK_BLOCK) bins at a time: their weights (scratch) and right edges are loaded up front, so a lane has that many
    // loads in flight instead of one round trip per bin (the emit stores inside the walk keep hipcc from hoisting them itself)
    for (int i0 = 0; i0 < n_in; i0 += WB) {
        float wv[WB], ev[WB];
#if TN_PDF_LOAD_PRIO
        __builtin_amdgcn_s_setprio(TN_PDF_LOAD_PRIO);
#endif
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            const int i = i0 + k < n_in ? i0 + k : n_in - 1;
            wv[k] = w[(size_t)i * 64];
            ev[k] = edge(i + 1);
        }
#if TN_PDF_LOAD_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            if (i0 + k >= n_in) break;
            const float wi = wv[k];
            const float wa = (anneal == 1.0f) ? wi : powf(wi, anneal);
            run += add_rn(add_rn(wa, 0.01f), pad_each) * rws;
            const float c1 = fminf(1.0f, run);
            const float b1 = ev[k];
            // searchsorted(cdf, u, side="right") lands in bin i  <=>  cdf[i] <= u < cdf[i+1]
            while (j < nb && uj < c1) {
                float t = nan_to_num(t_div<true>(sub_rn(uj, c0), sub_rn(c1, c0)));
                t = fminf(fmaxf(t, 0.0f), 1.0f);
                emit(j, add_rn(b0, mul_rn(t, sub_rn(b1, b0))));
                ++j;
                if (j < nb) uj = jittered ? add_rn(u[j], jrow ? jrow[j] / (float)nb : jit) : u[j];
            }
            c0 = c1;
            b0 = b1;
        }
    }
    // u >= cdf[n_in]: below == above == n_in -> t * 0 -> the last edge
    for (; j < nb; ++j) emit(j, b0);
}

// ND0 / ND1: dense levels of the two networks as compile-time constants (5, 4 = what the default 64 MB budget holds of the
// reference's proposal grids), or -1 / -1 = whatever the grids carry, decided per level at run time.
// LEAN = the eval default as compile-time facts (no jitter, piecewise spacing, anneal 1, scene contraction, both nets the
// 5-level / 16-hidden shape, no per-level outputs): the run-time switches of the general form cost scalar registers
// (spilled to lanes and read back per sample) and branches inside the sample loops.
template <int ND0, int ND1, bool LEAN>
__global__ void __launch_bounds__(kBlock, TN_PROP_WAVES) proposal_rays_kernel(PropRaysArgs ra) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PropArgs &a = ra.p;
    const int in0 = 2 * a.net[0].g.num_levels, in1 = 2 * a.net[1].g.num_levels;
    const int P0 = a.P0, P1 = a.P1, S = a.S;
    float *wbase0 = smem;
    float *wbase1 = wbase0 + two_layer_floats(in0, PH, 1);
    float *lin0 = wbase1 + two_layer_floats(in1, PH, 1);  // [P0+1]
    float *u1 = lin0 + (P0 + 1);                           // [P1+1]
    float *u2 = u1 + (P1 + 1);                             // [S+1]
    const TwoLayerLds w0 = stage_two_layer<PH>(wbase0, a.net[0].w0, a.net[0].b0, a.net[0].w1, a.net[0].b1, in0, 1);
    const TwoLayerLds w1 = stage_two_layer<PH>(wbase1, a.net[1].w0, a.net[1].b0, a.net[1].w1, a.net[1].b1, in1, 1);
    for (int i = threadIdx.x; i <= P0; i += blockDim.x) lin0[i] = a.lin0[i];
    for (int i = threadIdx.x; i <= P1; i += blockDim.x) u1[i] = a.u1[i];
    for (int i = threadIdx.x; i <= S; i += blockDim.x) u2[i] = a.u2[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Space sp0 = make_space(a.net[0].space), sp1 = make_space(a.net[1].space);
    if (LEAN) sp0.contraction = sp1.contraction = 1;
    const bool fast0 = LEAN || a.net[0].g.num_levels == 5, fast1 = LEAN || a.net[1].g.num_levels == 5;
    const bool jittered = !LEAN && a.jitter != nullptr;
    const bool lin = !LEAN && a.lin != 0;
    const float anneal = LEAN ? 1.0f : a.anneal;
    float *const osp0 = LEAN ? nullptr : a.out_spacing[0], *const osp1 = LEAN ? nullptr : a.out_spacing[1];
    float *const osp2 = LEAN ? nullptr : a.out_spacing[2];
    float *const oeu0 = LEAN ? nullptr : a.out_eucl[0], *const oeu1 = LEAN ? nullptr : a.out_eucl[1];
    float *const oeu2 = LEAN ? nullptr : a.out_eucl[2];
    float *const ow0 = LEAN ? nullptr : a.out_w[0], *const ow1 = LEAN ? nullptr : a.out_w[1];
    const long long tiles = (a.R + 63) >> 6;
    const long long stride = (long long)gridDim.x * kWaves;
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += stride) {
        const long long r = tile * 64 + lane;
        const bool live = r < a.R;
        const long long rc = live ? r : a.R - 1;
        const float ox = a.origins[rc * 3], oy = a.origins[rc * 3 + 1], oz = a.origins[rc * 3 + 2];
        const float dx = a.dirs[rc * 3], dy = a.dirs[rc * 3 + 1], dz = a.dirs[rc * 3 + 2];
        const float s_near = spacing_fn(a.nears[rc], lin), s_far = spacing_fn(a.fars[rc], lin);
        float *wsc = ra.w_scratch + (size_t)tile * ra.nmax * 64 + lane;       // w[i] at wsc[i*64]
        float *b1sc = ra.b1_scratch + (size_t)tile * (P1 + 1) * 64 + lane;   // edge j at b1sc[j*64]
        float *fin = a.ws_spacing + tn_ws_bin(tile * 64, 0, S) + lane;       // final edge j at fin[j*64]
        const bool jper = jittered && a.jper != 0;
        const float t0 = (jittered && !jper) ? a.jitter[rc] : 0.0f;
        const float *jr0 = jper ? a.jitter + rc * (P0 + 1) : nullptr;
        const float *jr1 = jper ? a.jitter + a.R * (P0 + 1) + rc * (P1 + 1) : nullptr;
        const float *jr2 = jper ? a.jitter + a.R * (P0 + 1 + P1 + 1) + rc * (S + 1) : nullptr;
        // level-0 spacing edge j: linspace, or its stratified jitter
        auto edge0 = [&](int j) -> float {
            float b = lin0[j];
            if (jittered) {
                const float lo = (j == 0) ? lin0[0] : add_rn(lin0[j], lin0[j - 1]) / 2.0f;
                const float hi = (j == P0) ? lin0[P0] : add_rn(lin0[j + 1], lin0[j]) / 2.0f;
                b = add_rn(lo, mul_rn(sub_rn(hi, lo), jper ? jr0[j] : t0));
            }
            return b;
        };
        // ================= level 0: P0 samples through proposal net 0 =====================================
        float total = 0.0f, med0 = 0.0f;
        {
            float accum = 0.0f, cum_w = 0.0f, step = 0.0f;
            bool found = false;
            float sb = edge0(0);
            float en = spacing_to_eucl<true>(sb, s_near, s_far, lin);
            if (live && osp0) osp0[r * (P0 + 1)] = sb;
            if (live && oeu0) oeu0[r * (P0 + 1)] = en;
            for (int i = 0; i < P0; ++i) {
                const float st = en;
                sb = edge0(i + 1);
                en = spacing_to_eucl<true>(sb, s_near, s_far, lin);
                step = add_rn(st, en) / 2.0f;
                float px, py, pz;
                const float sel = normalize_position<true>(sp0, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                           frustum_pos(oz, dz, st, en), px, py, pz);
                const float dens = fast0 ? proposal_density_kmajor<true, ND0>(a.net[0].g, as_scalar(a.wk), a.net[0].avg, px, py, pz, sel)
                                         : proposal_density_eval<PH, 0, true>(a.net[0].g, w0, a.net[0].avg, px, py, pz, sel);
                const float dd = mul_rn(sub_rn(en, st), dens);
                const float wi = nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-accum)));
                accum += dd;
                cum_w += wi;
                if (!found && cum_w >= 0.5f) {
                    found = true;
                    med0 = step;
                }
                wsc[(size_t)i * 64] = wi;
                total += add_rn((anneal == 1.0f) ? wi : powf(wi, anneal), 0.01f);
                if (live && ow0) ow0[r * P0 + i] = wi;
                if (live && osp0) osp0[r * (P0 + 1) + i + 1] = sb;
                if (live && oeu0) oeu0[r * (P0 + 1) + i + 1] = en;
            }
            if (!found) med0 = step;
        }
        // ================= PDF resample 0 -> P1 + 1 edges ===================================================
        pdf_walk(wsc, P0, total, anneal, u1, jittered, (jittered && !jper) ? a.jitter[a.R + rc] / (float)(P1 + 1) : 0.0f, P1,
                 edge0, [&](int j, float v) { b1sc[(size_t)j * 64] = v; }, jr1);
        // ================= level 1: P1 samples through proposal net 1 =====================================
        float med1 = 0.0f;
        total = 0.0f;
        {
            float accum = 0.0f, cum_w = 0.0f, step = 0.0f;
            bool found = false;
            float sb = b1sc[0];
            float en = spacing_to_eucl<true>(sb, s_near, s_far, lin);
            if (live && osp1) osp1[r * (P1 + 1)] = sb;
            if (live && oeu1) oeu1[r * (P1 + 1)] = en;
            for (int i = 0; i < P1; ++i) {
                const float st = en;
                sb = b1sc[(size_t)(i + 1) * 64];
                en = spacing_to_eucl<true>(sb, s_near, s_far, lin);
                step = add_rn(st, en) / 2.0f;
                float px, py, pz;
                const float sel = normalize_position<true>(sp1, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                           frustum_pos(oz, dz, st, en), px, py, pz);
                const float dens = fast1 ? proposal_density_kmajor<true, ND1>(a.net[1].g, as_scalar(a.wk + kPropWFloats), a.net[1].avg, px, py,
                                                                         pz, sel)
                                         : proposal_density_eval<PH, 0, true>(a.net[1].g, w1, a.net[1].avg, px, py, pz, sel);
                const float dd = mul_rn(sub_rn(en, st), dens);
                const float wi = nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-accum)));
                accum += dd;
                cum_w += wi;
                if (!found && cum_w >= 0.5f) {
                    found = true;
                    med1 = step;
                }
                wsc[(size_t)i * 64] = wi;
                total += add_rn((anneal == 1.0f) ? wi : powf(wi, anneal), 0.01f);
                if (live && ow1) ow1[r * P1 + i] = wi;
                if (live && osp1) osp1[r * (P1 + 1) + i + 1] = sb;
                if (live && oeu1) oeu1[r * (P1 + 1) + i + 1] = en;
            }
            if (!found) med1 = step;
        }
        // ================= PDF resample 1 -> S + 1 final edges (ray-tiled workspace) ========================
        pdf_walk(wsc, P1, total, anneal, u2, jittered, (jittered && !jper) ? a.jitter[2 * a.R + rc] / (float)(S + 1) : 0.0f, S,
                 [&](int j) -> float { return b1sc[(size_t)j * 64]; },
                 [&](int j, float v) {
                     fin[(size_t)j * 64] = v;
                     if (live && osp2) osp2[r * (S + 1) + j] = v;
                     if (live && oeu2) oeu2[r * (S + 1) + j] = spacing_to_eucl<true>(v, s_near, s_far, lin);
                 }, jr2);
        if (live) {
            if (a.prop_depth[0]) a.prop_depth[0][r] = med0;
            if (a.prop_depth[1]) a.prop_depth[1][r] = med1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The lane = ray proposal pass for calls that do not fill the chip (under 3 072 tiles; round 5): the SAME arithmetic in four launches.
// proposal_rays_kernel marches a tile's 256 + 96 samples on one wave, so a call lasts one whole march (0.6 ms) however few tiles it
// has: an 80 000-ray shard (1 250 tiles on 4 096 wave slots) pays 0.75 ms where its share of a frame is 0.36.  What is serial in a
// level is only the transmittance scan and the PDF walk; the density evaluations — the gathers and the MLP — are independent per
// sample.  So:  (1) proposal_density_segments_kernel<., 0>: (tile, segment) virtual tiles evaluate delta x density of level 0 into the
// scratch; (2) proposal_resample_kernel<0>: one wave per tile turns them into weights IN SAMPLE ORDER (the running optical depth is
// the same sum in the same order: weights, median, total and therefore the resampled edges are those of proposal_rays_kernel bit for
// bit), walks the PDF and leaves the 97 edges; (3) + (4) the same for level 1 -> the final S + 1 edges.  LEAN configuration only
// (the eval default: no jitter, piecewise spacing, anneal 1, contraction, 5-level nets, no per-level outputs).
// ------------------------------------------------------------------------------------------------------
struct PropSplitArgs {
    PropRaysArgs ra;
    int k[2], len[2];  // segments per tile and samples per segment, per level
};

template <int ND, int LEVEL>
__global__ void __launch_bounds__(kBlock, TN_PROP_WAVES) proposal_density_segments_kernel(PropSplitArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PropRaysArgs &ra = sa.ra;
    const PropArgs &a = ra.p;
    const int P0 = a.P0, P1 = a.P1;
    const int n = LEVEL == 0 ? P0 : P1, K = sa.k[LEVEL], len = sa.len[LEVEL];
    float *lin0 = smem;  // [P0+1] (level 0)
    if (LEVEL == 0) {
        for (int i = threadIdx.x; i <= P0; i += blockDim.x) lin0[i] = a.lin0[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Space sp = make_space(a.net[LEVEL].space);
    sp.contraction = 1;
    const long long tiles = (a.R + 63) >> 6, vtiles = tiles * K;
    const long long stride = (long long)gridDim.x * kWaves;
    for (long long vt = (long long)blockIdx.x * kWaves + wave; vt < vtiles; vt += stride) {
        const long long tile = vt / K;
        const int s0 = (int)(vt - tile * K) * len, s1 = s0 + len < n ? s0 + len : n;
        const long long r = tile * 64 + lane;
        const long long rc = r < a.R ? r : a.R - 1;
        const float ox = a.origins[rc * 3], oy = a.origins[rc * 3 + 1], oz = a.origins[rc * 3 + 2];
        const float dx = a.dirs[rc * 3], dy = a.dirs[rc * 3 + 1], dz = a.dirs[rc * 3 + 2];
        const float s_near = spacing_fn(a.nears[rc], false), s_far = spacing_fn(a.fars[rc], false);
        float *wsc = ra.w_scratch + (size_t)tile * ra.nmax * 64 + lane;
        const float *b1sc = ra.b1_scratch + (size_t)tile * (P1 + 1) * 64 + lane;
        auto edge = [&](int j) -> float { return LEVEL == 0 ? lin0[j] : b1sc[(size_t)j * 64]; };
        float en = spacing_to_eucl<true>(edge(s0), s_near, s_far, false);
        for (int i = s0; i < s1; ++i) {
            const float st = en;
            en = spacing_to_eucl<true>(edge(i + 1), s_near, s_far, false);
            float px, py, pz;
            const float sel = normalize_position<true>(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                       frustum_pos(oz, dz, st, en), px, py, pz);
            const float dens = proposal_density_kmajor<true, ND>(a.net[LEVEL].g, as_scalar(a.wk + LEVEL * kPropWFloats), a.net[LEVEL].avg,
                                                                 px, py, pz, sel);
            wsc[(size_t)i * 64] = mul_rn(sub_rn(en, st), dens);
        }
    }
}

template <int LEVEL>
__global__ void __launch_bounds__(kBlock, TN_PROP_WAVES) proposal_resample_kernel(PropSplitArgs sa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PropRaysArgs &ra = sa.ra;
    const PropArgs &a = ra.p;
    const int P0 = a.P0, P1 = a.P1, S = a.S;
    const int n = LEVEL == 0 ? P0 : P1, n_out = LEVEL == 0 ? P1 : S;
    float *lin0 = smem;          // [P0+1] (level 0)
    float *u = smem + (P0 + 1);  // [n_out+1]
    if (LEVEL == 0)
        for (int i = threadIdx.x; i <= P0; i += blockDim.x) lin0[i] = a.lin0[i];
    for (int i = threadIdx.x; i <= n_out; i += blockDim.x) u[i] = (LEVEL == 0 ? a.u1 : a.u2)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tiles = (a.R + 63) >> 6;
    const long long stride = (long long)gridDim.x * kWaves;
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += stride) {
        const long long r = tile * 64 + lane;
        const bool live = r < a.R;
        const long long rc = live ? r : a.R - 1;
        float *wsc = ra.w_scratch + (size_t)tile * ra.nmax * 64 + lane;
        float *b1sc = ra.b1_scratch + (size_t)tile * (P1 + 1) * 64 + lane;
        float *fin = a.ws_spacing + tn_ws_bin(tile * 64, 0, S) + lane;
        auto edge = [&](int j) -> float { return LEVEL == 0 ? lin0[j] : b1sc[(size_t)j * 64]; };
        // get_weights in sample order: the sums of proposal_rays_kernel, term for term
        float accum = 0.0f, cum_w = 0.0f, total = 0.0f;
        int med_idx = n - 1;
        bool found = false;
        constexpr int WB = TN_RESAMPLE_BLOCK;  // (a light kernel, one wave per tile: as many loads in flight as registers allow)
        for (int i0 = 0; i0 < n; i0 += WB) {
            float dv[WB];
#pragma unroll
            for (int k = 0; k < WB; ++k) dv[k] = wsc[(size_t)(i0 + k < n ? i0 + k : n - 1) * 64];
#pragma unroll
            for (int k = 0; k < WB; ++k) {
                if (i0 + k >= n) break;
                const float dd = dv[k];
                const float wi = nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-accum)));
                accum += dd;
                cum_w += wi;
                if (!found && cum_w >= 0.5f) {
                    found = true;
                    med_idx = i0 + k;
                }
                wsc[(size_t)(i0 + k) * 64] = wi;
                total += add_rn(wi, 0.01f);
            }
        }
        if (LEVEL == 0)
            pdf_walk<TN_RESAMPLE_BLOCK>(wsc, n, total, 1.0f, u, false, 0.0f, n_out, edge, [&](int j, float v) { b1sc[(size_t)j * 64] = v; });
        else
            pdf_walk<TN_RESAMPLE_BLOCK>(wsc, n, total, 1.0f, u, false, 0.0f, n_out, edge, [&](int j, float v) { fin[(size_t)j * 64] = v; });
        if (live && a.prop_depth[LEVEL]) {  // the median's mid-point, from the level's own edges (read before level 1 overwrites none of them)
            const float s_near = spacing_fn(a.nears[rc], false), s_far = spacing_fn(a.fars[rc], false);
            const float st = spacing_to_eucl<true>(edge(med_idx), s_near, s_far, false);
            const float en = spacing_to_eucl<true>(edge(med_idx + 1), s_near, s_far, false);
            a.prop_depth[LEVEL][r] = add_rn(st, en) / 2.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// main field + composite, lane per sample (reference form)
// ------------------------------------------------------------------------------------------------------
struct MainArgs {
    Grid g;
    tn_space space;
    const float *b0w, *b0b, *b1w, *b1b;
    HeadsArgs heads;
    float avg;
    const float *origins, *dirs, *nears, *fars;
    const int *cam;
    const float *spacing;  // [R,S+1]
    long long R;
    int S, training, lin;
    float *rgb, *acc, *depth, *expected, *thermal;
    float *out_w;  // optional [R,S]
    DepthSlots minmax;  // expected-depth clip bounds: one key pair per call, or per reference chunk of the frame
};


constexpr int GF = 15;  // geo_feat_dim supported by the fused path

__global__ void __launch_bounds__(kBlock) main_valu_kernel(MainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int in_dim = 2 * a.g.num_levels;
    const TwoLayerLds wb = stage_two_layer<HW>(smem, a.b0w, a.b0b, a.b1w, a.b1b, in_dim, 1 + GF);
    const HeadsLds wh = stage_heads(smem + two_layer_floats(in_dim, HW, 1 + GF), a.heads, a.training);
    __syncthreads();
    const Space sp = make_space(a.space);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool lin = a.lin != 0;
    const int S = a.S, A = a.heads.app_dim;
    const long long stride = (long long)gridDim.x * kWaves;
    float smin = INFINITY, smax = -INFINITY;  // running over every ray this wave renders
    long long mm_slot = 0;
    for (long long r = (long long)blockIdx.x * kWaves + wave; r < a.R; r += stride) {
        if (a.minmax.chunk_rays > 0 && a.minmax.slot(r) != mm_slot) {  // (a wave's rays ascend: at most one change per chunk)
            depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
            mm_slot = a.minmax.slot(r);
        }
        const float ox = a.origins[r * 3], oy = a.origins[r * 3 + 1], oz = a.origins[r * 3 + 2];
        const float dx = a.dirs[r * 3], dy = a.dirs[r * 3 + 1], dz = a.dirs[r * 3 + 2];
        const float s_near = spacing_fn(a.nears[r], lin), s_far = spacing_fn(a.fars[r], lin);
        const float *app = a.training ? (a.heads.appearance + (long long)a.cam[r] * A) : wh.APP;
        const WsBins sb{a.spacing + tn_ws_bin(r, 0, S)};  // ray-tiled workspace layout
        float carry = 0.0f, carry_w = 0.0f;
        float wsum = 0.0f, wr = 0.0f, wg = 0.0f, wbl = 0.0f, wth = 0.0f, wsteps = 0.0f;
        float last_r = 0.0f, last_g = 0.0f, last_b = 0.0f, last_t = 0.0f;
        int med_idx = S;
        for (int base = 0; base < S; base += 64) {
            const int i = base + lane;
            const bool ok = i < S;
            float dd = 0.0f, step = 0.0f, c[3] = {0.0f, 0.0f, 0.0f}, th = 0.0f;
            if (ok) {
                const float st = spacing_to_eucl(sb[i], s_near, s_far, lin);
                const float en = spacing_to_eucl(sb[i + 1], s_near, s_far, lin);
                step = add_rn(st, en) / 2.0f;
                float px, py, pz;
                const float sel = normalize_position(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                     frustum_pos(oz, dz, st, en), px, py, pz);
                float hid[HW];
                hidden_from_grid<HW>(a.g, wb, px, py, pz, hid);
                float o[1 + GF];
#pragma unroll
                for (int q = 0; q < 1 + GF; ++q) {
                    float acc = wb.B1[q];
                    const float *wrow = wb.W1 + q * HW;
#pragma unroll
                    for (int h = 0; h < HW; ++h) acc = fmaf(wrow[h], hid[h], acc);
                    o[q] = acc;
                }
                const float dens = mul_rn(mul_rn(a.avg, expf(o[0])), sel);
                heads_eval<GF>(wh, GF, A, a.heads.sh_shifted, dx, dy, dz, o + 1, app, c, th);
                dd = mul_rn(sub_rn(en, st), dens);
                if (!a.training) {
                    c[0] = nan_to_num(c[0]); c[1] = nan_to_num(c[1]); c[2] = nan_to_num(c[2]);
                    th = nan_to_num(th);
                }
                smin = fminf(smin, step);
                smax = fmaxf(smax, step);
            }
            const float incl = wave_incl_scan(dd, lane);
            const float excl = carry + wave_excl_from_incl(incl, lane);
            const float wi = ok ? nan_to_num(mul_rn(sub_rn(1.0f, expf(-dd)), expf(-excl))) : 0.0f;
            carry += lane_value<63>(incl);
            const float incl_w = wave_incl_scan(wi, lane) + carry_w;
            const unsigned long long hit = __ballot(ok && (incl_w >= 0.5f));
            if (hit && med_idx == S) med_idx = base + __ffsll((long long)hit) - 1;
            carry_w = lane_value<63>(incl_w);
            wsum += wi;
            wr += mul_rn(wi, c[0]);
            wg += mul_rn(wi, c[1]);
            wbl += mul_rn(wi, c[2]);
            wth += mul_rn(wi, th);
            wsteps += mul_rn(wi, step);
            if (a.out_w && ok) a.out_w[r * S + i] = wi;
            if (base + 64 >= S) {  // chunk holding the last sample
                const int src = (S - 1) - base;
                last_r = __shfl(c[0], src, 64);
                last_g = __shfl(c[1], src, 64);
                last_b = __shfl(c[2], src, 64);
                last_t = __shfl(th, src, 64);
            }
        }
        wsum = wave_sum(wsum);
        wr = wave_sum(wr); wg = wave_sum(wg); wbl = wave_sum(wbl); wth = wave_sum(wth); wsteps = wave_sum(wsteps);
        const int idx = min(med_idx, S - 1);
        if (lane == 0) {
            const float bg = sub_rn(1.0f, wsum);
            float cr = add_rn(wr, mul_rn(last_r, bg)), cg = add_rn(wg, mul_rn(last_g, bg)), cb = add_rn(wbl, mul_rn(last_b, bg));
            float ct = add_rn(wth, mul_rn(last_t, bg));
            if (!a.training) {
                cr = fminf(fmaxf(cr, 0.0f), 1.0f); cg = fminf(fmaxf(cg, 0.0f), 1.0f); cb = fminf(fmaxf(cb, 0.0f), 1.0f);
                ct = fminf(fmaxf(ct, 0.0f), 1.0f);
            }
            a.rgb[r * 3 + 0] = cr; a.rgb[r * 3 + 1] = cg; a.rgb[r * 3 + 2] = cb;
            a.thermal[r] = ct;
            a.acc[r] = wsum;
            const float st = spacing_to_eucl(sb[idx], s_near, s_far, lin), en = spacing_to_eucl(sb[idx + 1], s_near, s_far, lin);
            a.depth[r] = add_rn(st, en) / 2.0f;
            a.expected[r] = wsteps / add_rn(wsum, 1e-10f);
        }
    }
    // one atomic pair per wave (see main_mfma_kernel): per-ray returned atomics on one address serialise in L2
    depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
}

__global__ void depth_clip_kernel(float *__restrict__ expected, long long num_rays, const unsigned *mm) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    const float lo = key2f(mm[0]), hi = key2f(mm[1]);
    expected[r] = fminf(fmaxf(expected[r], lo), hi);
}

// per-chunk key pairs of a chunked call (tn_field_render_chunked_fwd): reset | keys -> floats in place | clip by the ray's chunk
__global__ void depth_slots_init_kernel(unsigned *keys, int slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < slots) {
        keys[2 * i] = 0xffffffffu;
        keys[2 * i + 1] = 0u;
    }
}

__global__ void depth_slots_export_kernel(unsigned *keys, int slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * slots) keys[i] = __float_as_uint(key2f(keys[i]));  // (every slot holds at least one ray of the call)
}

__global__ void depth_clip_chunked_kernel(float *__restrict__ expected, long long num_rays, long long first_ray, long long chunk_rays,
                                          const float *__restrict__ bounds) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rays) return;
    const long long slot = (first_ray + r) / chunk_rays - first_ray / chunk_rays;
    expected[r] = fminf(fmaxf(expected[r], bounds[2 * slot]), bounds[2 * slot + 1]);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline PropNet make_prop(const tn_density_field *f) {
    PropNet p;
    p.g = tn_make_grid(f->grid);
    p.space = f->space;
    p.w0 = f->l0.weight; p.b0 = f->l0.bias; p.w1 = f->l1.weight; p.b1 = f->l1.bias;
    p.avg = f->average_init_density;
    return p;
}

inline unsigned ray_grid(long long R, int blocks_per_cu) {
    const long long need = (R + kWaves - 1) / kWaves;
    const long long cap = 256LL * blocks_per_cu;
    return (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
}

}  // namespace

extern "C" {

size_t tn_render_workspace_bytes(const tn_render_config *cfg, int64_t num_rays) {
    if (!cfg || num_rays < 0) return 0;
    // [final edges, ray-tiled] [kWsMid B: depth min/max (8 B) | at +256: the proposal MLPs' k-major weight copies]
    // [proposal scratch: level weights + level-1 edges, ray-tiled]
    const size_t tiles = (size_t)((num_rays + 63) >> 6);
    const int P0 = cfg->num_proposal_samples[0], P1 = cfg->num_proposal_samples[1];
    const size_t nmax = (size_t)(P0 > P1 ? P0 : P1);
    return align_up(tn_ws_bin_floats(num_rays, cfg->num_nerf_samples) * sizeof(float), 256) + kWsMid +
           tiles * 64 * (nmax + (size_t)P1 + 1) * sizeof(float);
}

static int check_render_common(const tn_render_config *cfg, int64_t num_rays, void *workspace, size_t workspace_bytes) {
    if (!cfg || !workspace) return TN_ERR_NULL;
    const int P0 = cfg->num_proposal_samples[0], P1 = cfg->num_proposal_samples[1], S = cfg->num_nerf_samples;
    if (P0 < 1 || P1 < 1 || S < 1 || P0 > 1024 || P1 > 1024 || S > 1024 || num_rays < 0) return TN_ERR_SHAPE;
    if (cfg->initial_sampler != 0 && cfg->initial_sampler != 1) return TN_ERR_UNSUPPORTED;
    if (workspace_bytes < tn_render_workspace_bytes(cfg, num_rays)) return TN_ERR_WORKSPACE;
    return TN_OK;
}

static inline unsigned *ws_minmax(void *workspace, int64_t num_rays, int S) {
    return reinterpret_cast<unsigned *>(reinterpret_cast<char *>(workspace) +
                                        align_up(tn_ws_bin_floats(num_rays, S) * sizeof(float), 256));
}

int tn_proposal_sample_fwd(const tn_density_field *prop0, const tn_density_field *prop1, const tn_render_config *cfg,
                           const tn_render_inputs *in, const tn_render_outputs *out, int64_t num_rays, void *workspace,
                           size_t workspace_bytes, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!prop0 || !prop1 || !in || !out) return TN_ERR_NULL;
    TN_TRY(check_render_common(cfg, num_rays, workspace, workspace_bytes));
    if (!in->origins || !in->directions || !in->nears || !in->fars || !in->lin_bins0 || !in->u1 || !in->u2) return TN_ERR_NULL;
    if (cfg->training && !in->jitter) return TN_ERR_NULL;
    TN_TRY(tn_check_density_field(prop0));
    TN_TRY(tn_check_density_field(prop1));
    if (prop0->l0.out_dim != PH || prop1->l0.out_dim != PH) return TN_ERR_UNSUPPORTED;
    const int P0 = cfg->num_proposal_samples[0], P1 = cfg->num_proposal_samples[1], S = cfg->num_nerf_samples;
    hipStream_t s = (hipStream_t)stream;
    PropArgs pa;
    pa.net[0] = make_prop(prop0);
    pa.net[1] = make_prop(prop1);
    pa.origins = in->origins; pa.dirs = in->directions; pa.nears = in->nears; pa.fars = in->fars;
    pa.lin0 = in->lin_bins0; pa.u1 = in->u1; pa.u2 = in->u2;
    pa.jitter = cfg->training ? in->jitter : nullptr;
    pa.jper = cfg->per_sample_jitter != 0;
    pa.R = num_rays; pa.P0 = P0; pa.P1 = P1; pa.S = S; pa.training = cfg->training; pa.anneal = cfg->pdf_anneal;
    pa.lin = cfg->initial_sampler == 1;
    pa.ws_spacing = reinterpret_cast<float *>(workspace);
    for (int i = 0; i < 3; ++i) { pa.out_spacing[i] = out->spacing_bins[i]; pa.out_eucl[i] = out->eucl_bins[i]; }
    pa.out_w[0] = out->weights[0]; pa.out_w[1] = out->weights[1];
    pa.prop_depth[0] = out->prop_depth_0; pa.prop_depth[1] = out->prop_depth_1;
    const int nmax = P0 > P1 ? P0 : P1;
    int nbmax = nmax > S ? nmax : S;
    nbmax += 1;
    const size_t prop_smem = (size_t)(two_layer_floats(2 * prop0->grid.num_levels, PH, 1) +
                                      two_layer_floats(2 * prop1->grid.num_levels, PH, 1) +
                                      kWaves * (3 * nbmax + nmax)) * sizeof(float);

    // lane = ray needs >= ~1250 tiles to fill the chip (a tile marches its 64 rays serially: 1.1 ms whatever the count);
    // below ~80 k rays one wave per ray finishes sooner (4096 rays: 0.08 vs 1.1 ms).  cfg->kernel_family forces either form.
    {   // k-major weight copies for the scalar-operand MLP of both kernel forms (two tiny blocks, same stream)
        float *wk = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                              align_up(tn_ws_bin_floats(num_rays, S) * sizeof(float), 256) + 256);
        hipLaunchKernelGGL(prop_weights_kmajor_kernel, dim3(2), dim3(256), 0, s, pa.net[0], pa.net[1], wk,
                           ws_minmax(workspace, num_rays, S));
        pa.wk = wk;
    }
    const bool small_call = tn_render_kernel_form(nullptr, cfg, num_rays, 0) == 2;
    if (!small_call) {
        // lane = ray
        PropRaysArgs ra;
        ra.p = pa;
        ra.nmax = nmax;
        char *scratch = reinterpret_cast<char *>(workspace) + align_up(tn_ws_bin_floats(num_rays, S) * sizeof(float), 256) + kWsMid;
        const size_t tiles = (size_t)((num_rays + 63) >> 6);
        ra.w_scratch = reinterpret_cast<float *>(scratch);
        ra.b1_scratch = ra.w_scratch + tiles * 64 * (size_t)nmax;
        const size_t rsmem = (size_t)(two_layer_floats(2 * prop0->grid.num_levels, PH, 1) +
                                      two_layer_floats(2 * prop1->grid.num_levels, PH, 1) + (P0 + 1) + (P1 + 1) + (S + 1)) *
                             sizeof(float);
        const long long need = ((long long)tiles + kWaves - 1) / kWaves;
        constexpr long long kMaxGrid = 256LL * TN_PROP_WAVES;  // TN_PROP_WAVES workgroups (x 4 waves) per CU
        const unsigned grid = (unsigned)(need < kMaxGrid ? (need < 1 ? 1 : need) : kMaxGrid);
        const int nd0 = pa.net[0].g.num_dense, nd1 = pa.net[1].g.num_dense;
        const bool five = pa.net[0].g.num_levels == 5 && pa.net[1].g.num_levels == 5;
        bool lean = five && !pa.jitter && !pa.lin && pa.anneal == 1.0f && pa.net[0].space.contraction && pa.net[1].space.contraction &&
                    !pa.out_w[0] && !pa.out_w[1];
        for (int i = 0; i < 3; ++i) lean = lean && !pa.out_spacing[i] && !pa.out_eucl[i];
        // calls that leave most wave slots idle: density evaluations as (tile, segment) virtual tiles, scans + PDF walks per tile — the
        // same edges bit for bit (see proposal_density_segments_kernel).  Measured (tools/ab_prop_split.sh, S = 48, one-launch form ->
        // segments): 65 536 rays 0.62 -> 0.41 ms, 80 000: 0.73 -> 0.44, 160 000: 0.92 -> 0.80, 259 200: 1.16 -> 1.27: up to 3 072 tiles
        constexpr long long kSlots = 256LL * TN_PROP_WAVES * kWaves;
        if (TN_PROP_SPLIT && lean && five && 4 * (long long)tiles < 3 * kSlots && ((nd0 == 5 && nd1 == 4) || (nd0 == 0 && nd1 == 0)) && P0 >= 64 &&
            P1 >= 32) {
            PropSplitArgs sa;
            sa.ra = ra;
            const int counts[2] = {P0, P1}, kcap[2] = {8, 4};
            for (int l = 0; l < 2; ++l) {
                long long k = (3 * kSlots + (long long)tiles - 1) / (long long)tiles;  // ~three rounds of virtual tiles
                if (k > kcap[l]) k = kcap[l];
                if (k < 1) k = 1;
                sa.len[l] = (int)((counts[l] + k - 1) / k);
                sa.k[l] = (counts[l] + sa.len[l] - 1) / sa.len[l];
            }
            auto blocks = [&](long long units) {
                const long long nb = (units + kWaves - 1) / kWaves;
                return dim3((unsigned)(nb < kMaxGrid ? (nb < 1 ? 1 : nb) : kMaxGrid));
            };
            const size_t lds0 = (size_t)(P0 + 1) * sizeof(float);
            const size_t ldsr0 = (size_t)(P0 + 1 + P1 + 1) * sizeof(float), ldsr1 = (size_t)(P0 + 1 + S + 1) * sizeof(float);
            if (nd0 == 5) {
                hipLaunchKernelGGL((proposal_density_segments_kernel<5, 0>), blocks((long long)tiles * sa.k[0]), dim3(kBlock), lds0, s, sa);
                hipLaunchKernelGGL(proposal_resample_kernel<0>, blocks((long long)tiles), dim3(kBlock), ldsr0, s, sa);
                hipLaunchKernelGGL((proposal_density_segments_kernel<4, 1>), blocks((long long)tiles * sa.k[1]), dim3(kBlock), 0, s, sa);
            } else {
                hipLaunchKernelGGL((proposal_density_segments_kernel<0, 0>), blocks((long long)tiles * sa.k[0]), dim3(kBlock), lds0, s, sa);
                hipLaunchKernelGGL(proposal_resample_kernel<0>, blocks((long long)tiles), dim3(kBlock), ldsr0, s, sa);
                hipLaunchKernelGGL((proposal_density_segments_kernel<0, 1>), blocks((long long)tiles * sa.k[1]), dim3(kBlock), 0, s, sa);
            }
            hipLaunchKernelGGL(proposal_resample_kernel<1>, blocks((long long)tiles), dim3(kBlock), ldsr1, s, sa);
            TN_LAUNCH_CHECK();
            return TN_OK;
        }
        if (five && nd0 == 5 && nd1 == 4 && lean)
            hipLaunchKernelGGL((proposal_rays_kernel<5, 4, true>), dim3(grid), dim3(kBlock), rsmem, s, ra);
        else if (five && nd0 == 5 && nd1 == 4)
            hipLaunchKernelGGL((proposal_rays_kernel<5, 4, false>), dim3(grid), dim3(kBlock), rsmem, s, ra);
        else if (nd0 == 0 && nd1 == 0 && lean)
            hipLaunchKernelGGL((proposal_rays_kernel<0, 0, true>), dim3(grid), dim3(kBlock), rsmem, s, ra);
        else if (nd0 == 0 && nd1 == 0)
            hipLaunchKernelGGL((proposal_rays_kernel<0, 0, false>), dim3(grid), dim3(kBlock), rsmem, s, ra);
        else
            hipLaunchKernelGGL((proposal_rays_kernel<-1, -1, false>), dim3(grid), dim3(kBlock), rsmem, s, ra);
        TN_LAUNCH_CHECK();
        return TN_OK;
    }
    if (prop_smem > 64 * 1024 && !tn_ensure_dynamic_lds<proposal_kernel>(prop_smem)) return TN_ERR_LAUNCH;
    hipLaunchKernelGGL(proposal_kernel, dim3(ray_grid(num_rays, 8)), dim3(kBlock), prop_smem, s, pa, nmax, nbmax);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

// depth_bounds == nullptr: the call is one chunk (bounds in the workspace, clip applied).  Otherwise [slots, 2] device floats
// that receive, per reference chunk the call touches, the [min, max] of this call's sample mid-points in it; clip: apply them.
static int field_render_fwd(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                            const tn_render_outputs *out, int64_t num_rays, void *workspace, size_t workspace_bytes,
                            int64_t first_ray, int64_t chunk_rays, float *depth_bounds, int clip, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!field || !in || !out) return TN_ERR_NULL;
    TN_TRY(check_render_common(cfg, num_rays, workspace, workspace_bytes));
    if (!in->origins || !in->directions || !in->nears || !in->fars) return TN_ERR_NULL;
    if (!out->rgb || !out->accumulation || !out->depth || !out->expected_depth || !out->thermal) return TN_ERR_NULL;
    if (cfg->training && !in->camera_indices) return TN_ERR_NULL;
    TN_TRY(tn_check_thermal_field(field));
    if (field->geo_feat_dim != GF) return TN_ERR_UNSUPPORTED;
    const int S = cfg->num_nerf_samples;
    hipStream_t s = (hipStream_t)stream;
    const float *ws_spacing = reinterpret_cast<const float *>(workspace);
    DepthSlots minmax{ws_minmax(workspace, num_rays, S), 0, 0};
    int slots = 1;
    if (depth_bounds) {
        if (chunk_rays < 64 || chunk_rays % 64 != 0 || first_ray < 0 || first_ray % 64 != 0) return TN_ERR_UNSUPPORTED;
        const int64_t n_slots = (first_ray + num_rays - 1) / chunk_rays - first_ray / chunk_rays + 1;
        if (n_slots > (1 << 20)) return TN_ERR_SHAPE;
        slots = (int)n_slots;
        minmax = DepthSlots{reinterpret_cast<unsigned *>(depth_bounds), (long long)first_ray, (long long)chunk_rays};
        hipLaunchKernelGGL(depth_slots_init_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, minmax.keys, slots);
    }
    // minmax was reset by tn_proposal_sample_fwd, which filled this workspace; the bounds are a function of the edges in
    // it alone, so re-running the field kernel on the same workspace re-derives the same two values (atomic min/max)
    // the split-precision kernel only exists in the lane = ray form: small calls take the exact-fp32 ray-per-wave kernel
    const bool split_ok = !out->weights[2] && tn_render_kernel_form(field, cfg, num_rays, 1) == 1 &&
                          (field->prepared_bf16x6 || field->prepared_f16x3);
    if (field->prepared_bf16x6 && split_ok) {
        TN_TRY(launch_main_b6(field, cfg, in, out, (long long)num_rays, ws_spacing, minmax, s));
    } else if (field->prepared_f16x3 && split_ok) {
        TN_TRY(launch_main_h3(field, cfg, in, out, (long long)num_rays, ws_spacing, minmax, s));
    } else if (field->prepared) {
        // sample-split tiles (tn_render_sample_split): records + per-sample cumulative weights live in the proposal pass's scratch
        // region of this workspace (dead once the bin edges exist; 12 k + S floats per ray <= its 256 + 97 by the policy's cap)
        const int split = out->weights[2] ? 1 : tn_render_sample_split(field, cfg, num_rays);
        float *seg_scratch = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) +
                                                       align_up(tn_ws_bin_floats(num_rays, S) * sizeof(float), 256) + kWsMid);
        TN_TRY(launch_main_mfma(field, cfg, in, out, (long long)num_rays, ws_spacing, minmax, s, split, seg_scratch));
    } else {
        MainArgs ma;
        ma.g = tn_make_grid(field->grid);
        ma.space = field->space;
        ma.b0w = field->base0.weight; ma.b0b = field->base0.bias; ma.b1w = field->base1.weight; ma.b1b = field->base1.bias;
        ma.heads = make_heads_args(field);
        ma.avg = field->average_init_density;
        ma.origins = in->origins; ma.dirs = in->directions; ma.nears = in->nears; ma.fars = in->fars;
        ma.cam = in->camera_indices;
        ma.spacing = ws_spacing;
        ma.R = num_rays; ma.S = S; ma.training = cfg->training; ma.lin = cfg->initial_sampler == 1;
        ma.rgb = out->rgb; ma.acc = out->accumulation; ma.depth = out->depth; ma.expected = out->expected_depth;
        ma.thermal = out->thermal; ma.out_w = out->weights[2]; ma.minmax = minmax;
        const size_t main_smem = (size_t)(two_layer_floats(2 * field->grid.num_levels, HW, 1 + GF) +
                                          heads_floats(GF, field->app_dim)) * sizeof(float);
        if (!tn_ensure_dynamic_lds<main_valu_kernel>(main_smem)) return TN_ERR_LAUNCH;
        hipLaunchKernelGGL(main_valu_kernel, dim3(ray_grid(num_rays, 2)), dim3(kBlock), main_smem, s, ma);
        TN_LAUNCH_CHECK();
    }
    if (depth_bounds) {
        hipLaunchKernelGGL(depth_slots_export_kernel, dim3((2 * slots + 255) / 256), dim3(256), 0, s, minmax.keys, slots);
        if (clip)  // the call covers its chunks wholly; otherwise the clip waits for the other parts' bounds
            hipLaunchKernelGGL(depth_clip_chunked_kernel, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, s,
                               out->expected_depth, (long long)num_rays, (long long)first_ray, (long long)chunk_rays, depth_bounds);
    } else {
        hipLaunchKernelGGL(depth_clip_kernel, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, s, out->expected_depth,
                           (long long)num_rays, minmax.keys);
    }
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_field_render_fwd(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                        const tn_render_outputs *out, int64_t num_rays, void *workspace, size_t workspace_bytes,
                        void *stream) {
    return field_render_fwd(field, cfg, in, out, num_rays, workspace, workspace_bytes, 0, 0, nullptr, 1, stream);
}

// the most sample segments per tile the exact-fp32 lane = ray field kernel can use on this configuration (1 = it cannot split): the
// records (12 floats per segment) and the per-sample cumulative weights (S floats) of a ray must fit the proposal scratch of the
// workspace (max(P0, P1) + P1 + 1 floats per ray); a segment is at least 8 samples
static int sample_split_kmax(const tn_thermal_field *field, const tn_render_config *cfg) {
    if (!cfg || !field || cfg->sample_split == 1) return 1;
    if (cfg->training || cfg->early_stop_transmittance > 0.0f || !field->prepared || field->prepared_bf16x6 || field->prepared_f16x3) return 1;
    const int S = cfg->num_nerf_samples, P0 = cfg->num_proposal_samples[0], P1 = cfg->num_proposal_samples[1];
    const int scratch = (P0 > P1 ? P0 : P1) + P1 + 1;
    int kmax = (scratch - S) / 12;
    if (kmax > S / 8) kmax = S / 8;
    if (kmax > 8) kmax = 8;
    return kmax < 2 ? 1 : kmax;
}

int32_t tn_render_kernel_form(const tn_thermal_field *field, const tn_render_config *cfg, int64_t num_rays, int32_t pass) {
    if (!cfg) return 0;
    if (cfg->kernel_family == 1 || cfg->kernel_family == 2) return cfg->kernel_family;
    // measured, tools/small_call_forms.py (profiles/micro/round5_small_call_forms.txt): the proposal pass's lane = ray form — density
    // segments + per-tile resampling for calls this small — has a floor of 0.21 ms and meets the ray-per-wave form at 24 576 rays
    // (0.26 against 0.27 ms; 65 536 rays: 0.41 against 0.62); a training call's one-launch form (0.6 ms floor) at 65 536
    if (pass == 0) return num_rays < (cfg->training ? 65536 : 24576) ? 2 : 1;
    // field pass: the split-precision kernels exist in the lane = ray form only and pay from 640 tiles; the exact-fp32 kernel marches
    // a small call's tiles in segments (tn_render_sample_split) and then beats one ray per wave from 8 192 rays (0.50 against
    // 0.57 ms at S = 192, 0.19 against 0.24 at S = 48; 32 768 rays: 1.61 against 2.15 and 0.43 against 0.76); without segments
    // (early termination, sample_split = 1) its whole tiles pay from 896
    const bool split_precision = field && !cfg->training && (field->prepared_bf16x6 || field->prepared_f16x3);
    const int64_t from = split_precision ? 40960 : (sample_split_kmax(field, cfg) >= 2 ? 8192 : 57344);
    return num_rays < from ? 2 : 1;
}

int32_t tn_render_sample_split(const tn_thermal_field *field, const tn_render_config *cfg, int64_t num_rays) {
    if (!cfg || !field || num_rays <= 0) return 1;
    const int kmax = sample_split_kmax(field, cfg);  // only the exact-fp32 lane = ray eval kernel has the segmented form
    if (kmax < 2 || tn_render_kernel_form(field, cfg, num_rays, 1) != 1) return 1;
    const int S = cfg->num_nerf_samples;
    if (cfg->sample_split > 1) return cfg->sample_split < kmax ? cfg->sample_split : kmax;
    // by call size: a call of T tiles lasts about ceil(k T / 2048) marches of ceil(S / k) samples (+ half a sample's worth of
    // per-segment set-up).  The waves do not run in lock step, so finer pieces also balance better than the round count says
    // (measured, tools/ab_split.py: 4 050 tiles at S = 48 are two exact rounds, and 4.83 ms whole against 4.53 in 6 segments):
    // the LARGEST k within 7 % of the cheapest.  From four full rounds on (the 800 x 800 frame: 10 000 tiles) the serial march stays.
    const long long slots = 2048, tiles = (num_rays + 63) / 64;
    if (tiles >= 4 * slots) return 1;
    long long cost[9], best_cost = 0;
    for (int k = 1; k <= kmax; ++k) {
        cost[k] = ((tiles * k + slots - 1) / slots) * (2 * ((S + k - 1) / k) + 1);
        if (k == 1 || cost[k] < best_cost) best_cost = cost[k];
    }
    int best = 1;
    for (int k = 2; k <= kmax; ++k)
        if (100 * cost[k] <= 107 * best_cost) best = k;
    return best;
}

int64_t tn_depth_bound_slots(int64_t first_ray, int64_t num_rays, int64_t chunk_rays) {
    if (num_rays <= 0 || chunk_rays <= 0 || first_ray < 0) return 0;
    return (first_ray + num_rays - 1) / chunk_rays - first_ray / chunk_rays + 1;
}

int tn_field_render_chunked_fwd(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                                const tn_render_outputs *out, int64_t num_rays, void *workspace, size_t workspace_bytes,
                                int64_t first_ray, int64_t chunk_rays, float *depth_bounds, int32_t clip, void *stream) {
    if (!depth_bounds) return TN_ERR_NULL;
    return field_render_fwd(field, cfg, in, out, num_rays, workspace, workspace_bytes, first_ray, chunk_rays, depth_bounds,
                            clip, stream);
}

int tn_expected_depth_clip_chunked(float *expected_depth, int64_t num_rays, int64_t first_ray, int64_t chunk_rays,
                                   const float *depth_bounds, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!expected_depth || !depth_bounds) return TN_ERR_NULL;
    if (num_rays < 0 || chunk_rays < 1 || first_ray < 0) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(depth_clip_chunked_kernel, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       expected_depth, (long long)num_rays, (long long)first_ray, (long long)chunk_rays, depth_bounds);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_render_rays_fwd(const tn_density_field *prop0, const tn_density_field *prop1, const tn_thermal_field *field,
                       const tn_render_config *cfg, const tn_render_inputs *in, const tn_render_outputs *out,
                       int64_t num_rays, void *workspace, size_t workspace_bytes, void *stream) {
    TN_TRY(tn_proposal_sample_fwd(prop0, prop1, cfg, in, out, num_rays, workspace, workspace_bytes, stream));
    return tn_field_render_fwd(field, cfg, in, out, num_rays, workspace, workspace_bytes, stream);
}

const char *tn_version(void) { return "thermonerf_hip 0.1 gfx950"; }

}  // extern "C"
