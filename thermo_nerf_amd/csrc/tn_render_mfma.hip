// main_mfma_kernel — ThermalNerfactoTField.forward + get_weights + renderers for the S final samples of a ray
// [REF thermo_nerf/thermal_nerf/thermal_field.py:108-201; thermal_nerf_model.py:225-243,271-273] with the
// five 64-wide MLP layers on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact f32 = an fmaf chain).
//
// One wave64 owns one ray; a pass handles 64 samples (lane = sample for the hash-grid phase and for compositing,
// two 32-sample N-tiles for the MFMA phase).  Every layer is computed TRANSPOSED, H_out^T = W . H_in^T:
//   A operand = weights      A[i = out feature][k]   lane l holds (i = l&31, k = ks(l>>5))
//   B operand = activations  B[k][j = sample]        lane l holds (j = l&31, k = ks(l>>5))
//   C/D                      lane l, reg r holds out feature (r&3) + 8(r>>2) + 4(l>>5), sample l&31
// so a layer's accumulator registers ARE the next layer's B operands (lane<->sample is preserved; the k order a
// register carries is baked into the pre-permuted A fragments by tn_field_prepare).  No activation ever goes
// through LDS; the hash-grid features enter the first layer through 16 v_permlane32_swap.
//
// Work per 64 samples: 384 32x32x2 MFMAs (base 32->64: 64, geo->colour/thermal hidden: 32+32, 64->64: 128+128) + 64
// 16x16x4 MFMAs (base 64->16: its 16 output rows fill a 16x16 tile exactly; as a 32-row tile half the matrix work would be
// zero rows) = 26.6 k MFMA cycles per SIMD; SH + appearance fold into a per-RAY bias of the colour layer; the 64->3 and
// 64->1 output layers are VALU dot products on the accumulator registers.
//
// Layout changes between the two tile shapes cost one v_permlane16_swap per two registers:
//   32x32 C/D -> 16x16x4 B:  swap16(R, R') of two accumulator registers = [R.row0 R'.row0 R.row2 R'.row2] and
//                            [R.row1 R'.row1 R.row3 R'.row3] (row = 16 lanes): samples 0-15 resp. 16-31 of the 32-sample
//                            tile, each with four features in the four 16-lane groups = one k-step of 16x16x4 for each of
//                            the two 16-sample tiles
//   16x16 C/D -> 32x32x2 B:  swap16(G_2n[q], G_2n+1[q]) = rows (q | 8+q) and (4+q | 12+q) of the 32 samples of tile n in the
//                            lower | upper 32 lanes = two k-steps of the geo -> hidden layers
#include "tn_field_eval.h"
// s_setprio of a wave while it is in its matrix-rich MLP block (0 = none).  Two waves share a SIMD: with both at priority 0 the
// issue arbitration goes by age, and the partner's hash block (index arithmetic, gathers) holds up the first MFMAs of a sample's
// chain; at priority 1 the MLP block goes first and the partner's vector work fills in behind it.  main_mfma_rays_kernel: 31.9 ->
// 30.4 ms per 640 k-ray launch at S=192, 10.7 -> 10.1 at S=64 (A/B on one box, both orders; outputs bit-identical).
#ifndef TN_MFMA_MLP_PRIO
#define TN_MFMA_MLP_PRIO 1
#endif
// priority of a wave while it computes a level group's hash indices and issues its gathers (hash_encode_pipelined's GP).  Eval
// frame (coherent rays, the kernel waits for the matrix pipe): 1 beside MLP priority 1 costs 3 % (31.4 against 30.6 ms), beside
// MLP priority 2 nothing: off.  Training forward (incoherent rays, the kernel waits for its gathers): the step 2.49-2.51 ->
// 2.43-2.46 ms at S=192 (tools/ab_train.sh, two interleaved repetitions): on.
#ifndef TN_EVAL_GATHER_PRIO
#define TN_EVAL_GATHER_PRIO 0
#endif
#ifndef TN_TRAIN_GATHER_PRIO
#define TN_TRAIN_GATHER_PRIO 1
#endif

using namespace tn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / TN_WAVE;
constexpr int GF = 15, APP = 32, L16 = 16, IN0 = 16 + GF + APP;
#ifndef TN_LEVEL_GROUP
#define TN_LEVEL_GROUP 4
#endif
constexpr int LG = TN_LEVEL_GROUP;  // hash levels whose gathers are in flight together

// ---- prepared blob / LDS layout (floats) ---------------------------------------------------------------
constexpr int A_BASE1 = 0, A_BASE2 = 32, A_C1 = 48, A_T1 = 64, A_C2 = 80, A_T2 = 144, A_SH = 208, A_COMBOS = 224;
constexpr int OFF_A = 0;
constexpr int OFF_B_BASE1 = A_COMBOS * 64;        // [64]
constexpr int OFF_B_BASE2 = OFF_B_BASE1 + 64;     // [32] rows 0..15 valid
constexpr int OFF_B_C1_EVAL = OFF_B_BASE2 + 32;   // [64] bias + W_app . mean(appearance)   (eval)
constexpr int OFF_B_C1_RAW = OFF_B_C1_EVAL + 64;  // [64] bias                              (training)
constexpr int OFF_B_T1 = OFF_B_C1_RAW + 64;       // [64]
constexpr int OFF_B_C2 = OFF_B_T1 + 64;           // [64]
constexpr int OFF_B_T2 = OFF_B_C2 + 64;           // [64]
constexpr int OFF_W_SH = OFF_B_T2 + 64;           // [16][64]
constexpr int OFF_W_APP = OFF_W_SH + 16 * 64;     // [32][64]
constexpr int OFF_W3 = OFF_W_APP + 32 * 64;       // [3][64] + [4] bias
constexpr int OFF_WTH = OFF_W3 + 3 * 64 + 4;      // [64] + [4] bias
constexpr int BLOB_FLOATS = OFF_WTH + 64 + 4;     // 18024 floats = 72 096 B
constexpr int OFF_SCRATCH = BLOB_FLOATS;          // per-wave [64] colour-layer bias of the current ray
constexpr int LDS_FLOATS = OFF_SCRATCH + kWaves * 64;
static_assert(BLOB_FLOATS % 4 == 0, "blob must be float4-copyable");

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// geo row (0 = raw density, 1..15 = geo features) fed by k-step j of the geo -> hidden layers in lane half h (see swap16 above)
__device__ __forceinline__ int grow(int j, int h) { return (j >> 1) + ((j & 1) ? 4 : 0) + 8 * h; }

struct RawField {
    const float *b0w, *b0b, *b1w, *b1b, *h0w, *h0b, *h1w, *h1b, *h2w, *h2b, *t0w, *t0b, *t1w, *t1b, *thw, *thb;
    const float *appearance;
    int num_images, use_avg;
};

__global__ void field_prepare_kernel(RawField w, float *__restrict__ blob) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BLOB_FLOATS) return;
    float v = 0.0f;
    if (idx < A_COMBOS * 64) {
        const int combo = idx >> 6, lane = idx & 63, i = lane & 31, h = lane >> 5;
        if (combo < A_BASE2) {  // mlp_base layer 0: [64,32]; k-step s feeds features (2s, 2s+1)
            const int mt = combo >> 4, s = combo & 15;
            v = w.b0w[(i + 32 * mt) * 32 + 2 * s + h];
        } else if (combo < A_C1) {  // mlp_base layer 1: [16,64] as 16x16x4 A fragments: lane = (out row l&15, k slot l>>4)
            const int c = combo - A_BASE2, mi = c >> 3, rp = c & 7, slot = lane >> 4;
            v = w.b1w[(lane & 15) * 64 + 32 * mi + crow(2 * rp + (slot & 1), slot >> 1)];
        } else if (combo < A_T1) {  // mlp_head layer 0, geo columns [16, 16+GF); row 0 of the input is the raw density
            const int c = combo - A_C1, mt = c >> 3, j = c & 7, row = grow(j, h);
            v = (row >= 1) ? w.h0w[(i + 32 * mt) * IN0 + 16 + (row - 1)] : 0.0f;
        } else if (combo < A_C2) {  // mlp_thermal layer 0: [64,15]
            const int c = combo - A_T1, mt = c >> 3, j = c & 7, row = grow(j, h);
            v = (row >= 1) ? w.t0w[(i + 32 * mt) * GF + (row - 1)] : 0.0f;
        } else if (combo >= A_SH) {  // mlp_head layer 0, SH columns [0,16): k-step s feeds SH comps (2s, 2s+1)
            const int c = combo - A_SH, mt = c >> 3, s = c & 7;
            v = w.h0w[(i + 32 * mt) * IN0 + 2 * s + h];
        } else {  // the two 64->64 layers
            const bool thermal = combo >= A_T2;
            const int c = combo - (thermal ? A_T2 : A_C2), mt = c >> 5, mi = (c >> 4) & 1, s = c & 15;
            const float *m = thermal ? w.t1w : w.h1w;
            v = m[(i + 32 * mt) * 64 + 32 * mi + crow(s, h)];
        }
    } else if (idx < OFF_B_BASE2) {
        v = w.b0b[idx - OFF_B_BASE1];
    } else if (idx < OFF_B_C1_EVAL) {
        const int f = idx - OFF_B_BASE2;
        v = f < 1 + GF ? w.b1b[f] : 0.0f;
    } else if (idx < OFF_B_C1_RAW) {
        const int f = idx - OFF_B_C1_EVAL;
        v = w.h0b[f];
        if (w.use_avg) {  // REF thermal_field.py:128-132: ones * mean(embedding)
            for (int k = 0; k < APP; ++k) {
                float m = 0.0f;
                for (int im = 0; im < w.num_images; ++im) m += w.appearance[im * APP + k];
                v = fmaf(w.h0w[f * IN0 + 16 + GF + k], m / (float)w.num_images, v);
            }
        }
    } else if (idx < OFF_B_T1) {
        v = w.h0b[idx - OFF_B_C1_RAW];
    } else if (idx < OFF_B_C2) {
        v = w.t0b[idx - OFF_B_T1];
    } else if (idx < OFF_B_T2) {
        v = w.h1b[idx - OFF_B_C2];
    } else if (idx < OFF_W_SH) {
        v = w.t1b[idx - OFF_B_T2];
    } else if (idx < OFF_W_APP) {
        const int e = idx - OFF_W_SH, k = e >> 6, f = e & 63;
        v = w.h0w[f * IN0 + k];
    } else if (idx < OFF_W3) {
        const int e = idx - OFF_W_APP, k = e >> 6, f = e & 63;
        v = w.h0w[f * IN0 + 16 + GF + k];
    } else if (idx < OFF_WTH) {
        const int e = idx - OFF_W3;
        v = e < 192 ? w.h2w[e] : (e < 195 ? w.h2b[e - 192] : 0.0f);
    } else {
        const int e = idx - OFF_WTH;
        v = e < 64 ? w.thw[e] : (e == 64 ? w.thb[0] : 0.0f);
    }
    blob[idx] = v;
}

// ---- kernel --------------------------------------------------------------------------------------------
struct MfmaArgs {
    Grid g;
    tn_space space;
    const float *blob;
    const float *appearance;
    float avg;
    int sh_shifted;
    const float *origins, *dirs, *nears, *fars;
    const int *cam;
    const float *spacing;  // [R,S+1]
    long long R;
    int S, training, lin;
    float *rgb, *acc, *depth, *expected, *thermal;
    float *out_w;
    DepthSlots minmax;  // expected-depth clip bounds: one key pair per call, or per reference chunk of the frame
    float early_eps;  // eval only; 0 = never stop early
    // sample-split tiles (main_mfma_rays_kernel<., true>): a tile's S samples as `split` contiguous segments of seg_len, one wave
    // each; per (tile, segment) a 12-row record [64 lanes] in seg_rec, per (tile, sample) the segment-local cumulative weight in
    // seg_cum [tile][S][64] — both inside the proposal pass's scratch region of the workspace, which is dead by now
    int split, seg_len;
    float *seg_rec, *seg_cum;
};
constexpr int SEG_ROWS = 12;  // optical depth | sum w | sum w r,g,b | sum w th | sum w step | last sample's r,g,b,th | last mid-point

#define MFMA32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)
#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

// accumulator init from a natural-order bias vector in LDS: reg r <- bias[32*mt + crow(r,h)]
__device__ __forceinline__ f32x16 bias_frag(const float *bias, int mt, int h) {
    f32x16 v;
    const float4 q0 = *reinterpret_cast<const float4 *>(bias + 32 * mt + 4 * h);
    const float4 q1 = *reinterpret_cast<const float4 *>(bias + 32 * mt + 8 + 4 * h);
    const float4 q2 = *reinterpret_cast<const float4 *>(bias + 32 * mt + 16 + 4 * h);
    const float4 q3 = *reinterpret_cast<const float4 *>(bias + 32 * mt + 24 + 4 * h);
    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
    v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    v[8] = q2.x; v[9] = q2.y; v[10] = q2.z; v[11] = q2.w;
    v[12] = q3.x; v[13] = q3.y; v[14] = q3.z; v[15] = q3.w;
    return v;
}

__device__ __forceinline__ void swap32(float a, float b, float &lo, float &hi) {
    // lo = [a.lanes0-31 | b.lanes0-31 moved up], hi = [a.lanes32-63 moved down | b.lanes32-63]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

__device__ __forceinline__ void swap16(float a, float b, float &even, float &odd) {
    // even = [a.row0 b.row0 a.row2 b.row2], odd = [a.row1 b.row1 a.row3 b.row3]   (row = 16 lanes)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    even = __uint_as_float(r[0]);
    odd = __uint_as_float(r[1]);
}

// mlp_base layer 1 (64 -> 1 + geo = 16 rows, no activation) on relu(h1), then its outputs re-laid as the B operands of the
// eight k-steps of the geo -> hidden layers: gb[nt][j] (k rows grow(j, 0) | grow(j, 1)).  gb[nt][0] lanes 0-31 = raw density.
// G[T][q]: row 4 (l >> 4) + q of mlp_base's 16 outputs for sample 16 T + (l & 15)
__device__ __forceinline__ void base2_tiles(const float *A, const float *bias16, int lane, const f32x16 (&h1)[2][2],
                                            f32x4 (&G)[4]) {
    const float4 bq = *reinterpret_cast<const float4 *>(bias16 + 4 * (lane >> 4));  // C rows 4 (l >> 4) + q
    const f32x4 b4 = {bq.x, bq.y, bq.z, bq.w};
    G[0] = b4; G[1] = b4; G[2] = b4; G[3] = b4;  // 16-sample tiles 0..3
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const float aw = A[(A_BASE2 + mi * 8 + rp) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float lo, hi;
                swap16(relu_bits(h1[mi][nt][2 * rp]), relu_bits(h1[mi][nt][2 * rp + 1]), lo, hi);
                MFMA16(G[2 * nt], aw, lo);
                MFMA16(G[2 * nt + 1], aw, hi);
            }
        }
    }
}
__device__ __forceinline__ void geo_relayout(const f32x4 (&G)[4], float (&gb)[2][8]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) swap16(G[2 * nt][q], G[2 * nt + 1][q], gb[nt][2 * q], gb[nt][2 * q + 1]);
    }
}
__device__ __forceinline__ void base2_geo(const float *A, const float *bias16, int lane, const f32x16 (&h1)[2][2],
                                          float (&gb)[2][8]) {
    f32x4 G[4];
    base2_tiles(A, bias16, lane, h1, G);
    geo_relayout(G, gb);
}

// one 64 -> 64 layer: out[mt][nt] = bias + sum over (mi, s) A[mt][mi][s] x relu(in[mi][nt][s])
__device__ __forceinline__ void layer64(const float *A, int combo0, const float *bias, int lane, int h,
                                        const f32x16 (&in)[2][2], f32x16 (&out)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        out[mt][0] = bias_frag(bias, mt, h);
        out[mt][1] = out[mt][0];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b0 = relu_bits(in[mi][0][s]), b1 = relu_bits(in[mi][1][s]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float a = A[(combo0 + mt * 32 + mi * 16 + s) * 64 + lane];
                MFMA32(out[mt][0], a, b0);
                MFMA32(out[mt][1], a, b1);
            }
        }
    }
}

// geo (rows 1..15 of g) -> 64 hidden: out[mt][nt] = bias + sum_s A[mt][s] x g[nt][s], s = 0..7
__device__ __forceinline__ void layer_geo(const float *A, int combo0, const float *bias, int lane, int h,
                                          const float (&g)[2][8], f32x16 (&out)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        out[mt][0] = bias_frag(bias, mt, h);
        out[mt][1] = out[mt][0];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float a = A[(combo0 + mt * 8 + s) * 64 + lane];
            MFMA32(out[mt][0], a, g[0][s]);
            MFMA32(out[mt][1], a, g[1][s]);
        }
    }
}

// VALU dot of the lane's 32 hidden features (both M tiles) with one output row, for both N tiles; ACT applied first
template <int ACT>  // 0 relu, 1 sigmoid
__device__ __forceinline__ float2 out_dot(const float *wrow, int h, const f32x16 (&x)[2][2]) {
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4 *>(wrow + 32 * mt + 8 * q + 4 * h);
            const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = x[mt][0][4 * q + e], a1 = x[mt][1][4 * q + e];
                const float v0 = ACT ? sigmoidf(a0) : relu_bits(a0);
                const float v1 = ACT ? sigmoidf(a1) : relu_bits(a1);
                p0 = fmaf(ww[e], v0, p0);
                p1 = fmaf(ww[e], v1, p1);
            }
        }
    }
    return make_float2(p0, p1);
}

// lanes 0-31 <- total of N-tile 0, lanes 32-63 <- total of N-tile 1 (partial sums live in both halves)
__device__ __forceinline__ float combine_halves(float2 p) {
    float lo, hi;
    swap32(p.x, p.y, lo, hi);
    return lo + hi;
}


__global__ void __launch_bounds__(kBlock, 2) main_mfma_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {   // stage the prepared blob (float4, coalesced)
        const float4 *src = reinterpret_cast<const float4 *>(a.blob);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();
    const float *A = lds + OFF_A;
    const Space sp = make_space(a.space);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const bool lin = a.lin != 0;
    float *scratch = lds + OFF_SCRATCH + wave * 64;
    const int S = a.S;
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
        const WsBins sb{a.spacing + tn_ws_bin(r, 0, S)};  // ray-tiled workspace layout
        {   // per-ray colour-layer bias: b + W_sh . SH(dir) (+ W_app . embedding[cam] in training); lane = feature
            float sx = dx, sy = dy, sz = dz;
            if (a.sh_shifted) {
                sx = add_rn(sx, 1.0f) / 2.0f; sy = add_rn(sy, 1.0f) / 2.0f; sz = add_rn(sz, 1.0f) / 2.0f;
            }
            float c[16];
            sh16(sx, sy, sz, c);
            float v = lds[(a.training ? OFF_B_C1_RAW : OFF_B_C1_EVAL) + lane];
#pragma unroll
            for (int k = 0; k < 16; ++k) v = fmaf(lds[OFF_W_SH + k * 64 + lane], c[k], v);
            if (a.training) {
                const float *emb = a.appearance + (long long)a.cam[r] * APP;
                for (int k = 0; k < APP; ++k) v = fmaf(lds[OFF_W_APP + k * 64 + lane], emb[k], v);
            }
            scratch[lane] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        float carry = 0.0f, carry_w = 0.0f;
        float wsum = 0.0f, wr = 0.0f, wg = 0.0f, wbl = 0.0f, wth = 0.0f, wsteps = 0.0f;
        float last_r = 0.0f, last_g = 0.0f, last_b = 0.0f, last_t = 0.0f;
        int med_idx = S;
        for (int base = 0; base < S; base += 64) {
            const int i = base + lane;
            const bool ok = i < S;
            const int ic = ok ? i : S - 1;  // idle lanes re-evaluate the last sample (masked out below)
            const float st = spacing_to_eucl<true>(sb[ic], s_near, s_far, lin);
            const float en = spacing_to_eucl<true>(sb[ic + 1], s_near, s_far, lin);
            const float step = add_rn(st, en) / 2.0f;
            float px, py, pz;
            const float sel = normalize_position<true>(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                 frustum_pos(oz, dz, st, en), px, py, pz);
            // ---- hash grid: 32 features of this lane's sample -> B operands of the two N tiles -------------
            float bt0[16], bt1[16];
            if (a.g.num_dense == 0) {
                hash_encode_pipelined<L16, 2>(a.g, px, py, pz, [&](int l, float2 f) { swap32(f.x, f.y, bt0[l], bt1[l]); });
            } else {
#pragma unroll
                for (int l0 = 0; l0 < L16; l0 += LG) {
                    float2 f[LG];
#pragma unroll
                    for (int q = 0; q < LG; ++q) f[q] = encode_level_any<true>(a.g, l0 + q, px, py, pz);
#pragma unroll
                    for (int q = 0; q < LG; ++q) swap32(f[q].x, f[q].y, bt0[l0 + q], bt1[l0 + q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- mlp_base layer 0: 32 -> 64 ---------------------------------------------------------------
            f32x16 h1[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                h1[mt][0] = bias_frag(lds + OFF_B_BASE1, mt, h);
                h1[mt][1] = h1[mt][0];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float aw = A[(A_BASE1 + mt * 16 + s) * 64 + lane];
                    MFMA32(h1[mt][0], aw, bt0[s]);
                    MFMA32(h1[mt][1], aw, bt1[s]);
                }
            }
            // ---- mlp_base layer 1: 64 -> 16 on 16x16x4 tiles, outputs re-laid for the geo -> hidden layers ----
            float g[2][8];
            base2_geo(A, lds + OFF_B_BASE2, lane, h1, g);
            // row 0 (lanes 0-31 of k-step 0) is the raw density of sample l&31 of each tile -> lane = sample
            float raw, unused;
            swap32(g[0][0], g[1][0], raw, unused);
            const float dens = mul_rn(mul_rn(a.avg, __expf(raw)), sel);
            // ---- colour branch: geo -> 64 (per-ray bias) -> 64 -> 3 sigmoid --------------------------------
            float cr, cg, cb, th;
            {
                f32x16 x1[2][2], x2[2][2];
                layer_geo(A, A_C1, scratch, lane, h, g, x1);
                layer64(A, A_C2, lds + OFF_B_C2, lane, h, x1, x2);
                const float *w3 = lds + OFF_W3;
                cr = sigmoidf(combine_halves(out_dot<0>(w3, h, x2)) + w3[192]);
                cg = sigmoidf(combine_halves(out_dot<0>(w3 + 64, h, x2)) + w3[193]);
                cb = sigmoidf(combine_halves(out_dot<0>(w3 + 128, h, x2)) + w3[194]);
            }
            // ---- thermal branch: geo -> 64 -> 64 sigmoid -> 1 -----------------------------------------------
            {
                f32x16 x1[2][2], x2[2][2];
                layer_geo(A, A_T1, lds + OFF_B_T1, lane, h, g, x1);
                layer64(A, A_T2, lds + OFF_B_T2, lane, h, x1, x2);
                const float *wt = lds + OFF_WTH;
                th = combine_halves(out_dot<1>(wt, h, x2)) + wt[64];
            }
            // ---- compositing (lane = sample) ---------------------------------------------------------------
            if (!a.training) {
                cr = nan_to_num(cr); cg = nan_to_num(cg); cb = nan_to_num(cb); th = nan_to_num(th);
            }
            const float dd = ok ? mul_rn(sub_rn(en, st), dens) : 0.0f;
            if (ok) {
                smin = fminf(smin, step);
                smax = fmaxf(smax, step);
            }
            const float incl = wave_incl_scan(dd, lane);
            const float excl = carry + wave_excl_from_incl(incl, lane);
            const float wi = ok ? nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-excl))) : 0.0f;
            carry += lane_value<63>(incl);
            const float incl_w = wave_incl_scan(wi, lane) + carry_w;
            const unsigned long long hit = __ballot(ok && (incl_w >= 0.5f));
            if (hit && med_idx == S) med_idx = base + __ffsll((long long)hit) - 1;
            carry_w = lane_value<63>(incl_w);
            wsum += wi;
            wr += mul_rn(wi, cr);
            wg += mul_rn(wi, cg);
            wbl += mul_rn(wi, cb);
            wth += mul_rn(wi, th);
            wsteps += mul_rn(wi, step);
            if (a.out_w && ok) a.out_w[r * S + i] = wi;
            if (base + 64 >= S) {
                const int src = (S - 1) - base;
                last_r = __shfl(cr, src, 64);
                last_g = __shfl(cg, src, 64);
                last_b = __shfl(cb, src, 64);
                last_t = __shfl(th, src, 64);
            }
        }
        wsum = wave_sum(wsum);
        wr = wave_sum(wr); wg = wave_sum(wg); wbl = wave_sum(wbl); wth = wave_sum(wth); wsteps = wave_sum(wsteps);
        const int idx = min(med_idx, S - 1);
        if (lane == 0) {
            const float bg = sub_rn(1.0f, wsum);
            float c0 = add_rn(wr, mul_rn(last_r, bg)), c1 = add_rn(wg, mul_rn(last_g, bg)), c2 = add_rn(wbl, mul_rn(last_b, bg));
            float ct = add_rn(wth, mul_rn(last_t, bg));
            if (!a.training) {
                c0 = fminf(fmaxf(c0, 0.0f), 1.0f); c1 = fminf(fmaxf(c1, 0.0f), 1.0f); c2 = fminf(fmaxf(c2, 0.0f), 1.0f);
                ct = fminf(fmaxf(ct, 0.0f), 1.0f);
            }
            a.rgb[r * 3 + 0] = c0; a.rgb[r * 3 + 1] = c1; a.rgb[r * 3 + 2] = c2;
            a.thermal[r] = ct;
            a.acc[r] = wsum;
            const float st = spacing_to_eucl<true>(sb[idx], s_near, s_far, lin), en = spacing_to_eucl<true>(sb[idx + 1], s_near, s_far, lin);
            a.depth[r] = add_rn(st, en) / 2.0f;
            a.expected[r] = wsteps / add_rn(wsum, 1e-10f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // call-global [min, max] of the sample mid-points (DepthRenderer "expected" clip): ONE atomic pair per wave for
    // all its rays.  (A returned atomic per ray on one address serialises at ~12 ns each in L2: 1.5 ms / 64k rays.)
    depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
}

// ------------------------------------------------------------------------------------------------------
// main_mfma_rays_kernel (eval): one wave64 owns 64 CONSECUTIVE RAYS and walks their samples in lock-step
// (lane = ray, loop index = sample).  Adjacent rays at the same sample index land in the same / neighbouring
// hash-grid cells, so a gather instruction touches a handful of cache lines instead of 64 (the ray-per-wave
// form spreads its 64 lanes along one ray: every fine-level gather is 64 distinct lines and the kernel was bound
// by the random-access rate of TCP/L2).  Compositing becomes a per-lane running sum — no cross-lane scan at all.
// The per-ray SH(dir) contribution of the colour layer rides along as 8 extra k-steps (B operands built once per
// ray group), so a pass is 416 32x32x2 + 64 16x16x4 MFMAs; the appearance term is folded into the bias by tn_field_prepare (eval).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x) {
    // v_exp_f32 / v_rcp_f32 (1 ulp each): |error| < 3e-7 absolute on a value in (0,1)
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

template <int ACT>  // 0 relu, 1 fast sigmoid
__device__ __forceinline__ float2 out_dot_fast(const float *wrow, int h, const f32x16 (&x)[2][2]) {
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4 *>(wrow + 32 * mt + 8 * q + 4 * h);
            const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = x[mt][0][4 * q + e], a1 = x[mt][1][4 * q + e];
                const float v0 = ACT ? fast_sigmoid(a0) : relu_bits(a0);
                const float v1 = ACT ? fast_sigmoid(a1) : relu_bits(a1);
                p0 = fmaf(ww[e], v0, p0);
                p1 = fmaf(ww[e], v1, p1);
            }
        }
    }
    return make_float2(p0, p1);
}

// SPLIT (round 5): the unit of work is a SEGMENT of a tile's sample march (virtual tile = tile * split + segment, neighbours in the
// grid: a tile's segments run side by side and share its table lines).  A lane = ray tile marches its samples serially, so a call
// of T tiles lasts ceil(T / 2048 wave slots) full marches whatever T is: 1 250 tiles (an 80 000-ray shard of the metric's frame on
// 8 GPUs) cost as much as 2 048.  With k segments per tile the same call is ceil(k T / 2048) marches of S / k samples.  A segment
// composites with its OWN transmittance (starting at 1) and leaves a record; segments_combine_kernel chains the records:
// w_i = T_j w_i^local with T_j = exp(-(optical depth of the segments before j)) — the reference's weights in another
// association of the same products (not bit-equal to the unsplit march: same tolerance, and the split count is a property of the
// CALL, tn_render_sample_split, so parts of a call agree with the whole bit for bit).  The median needs the crossing of 0.5 by
// the GLOBAL cumulative weight: every sample stores its segment-local one, the combine pass searches the segment that crosses.
template <bool DENSE, bool SPLIT = false>
__global__ void __launch_bounds__(kBlock, 2) main_mfma_rays_kernel(MfmaArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.blob);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();
    const float *A = lds + OFF_A;
    const Space sp = make_space(a.space);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const bool lin = a.lin != 0;
    const int S = a.S;
    const long long groups = (a.R + 63) >> 6;
    const long long stride = (long long)gridDim.x * kWaves;
    const int K = SPLIT ? a.split : 1;
    const long long vgroups = groups * K;
    float smin = INFINITY, smax = -INFINITY;
    long long mm_slot = 0;
    for (long long vg = (long long)blockIdx.x * kWaves + wave; vg < vgroups; vg += stride) {
        const long long grp = SPLIT ? vg / K : vg;
        const int s0 = SPLIT ? (int)(vg - grp * K) * a.seg_len : 0;
        const int s1 = SPLIT ? (s0 + a.seg_len < S ? s0 + a.seg_len : S) : S;
        // (a tile never straddles two chunks: first_ray and chunk_rays are multiples of 64; a wave's tiles ascend)
        if (a.minmax.chunk_rays > 0 && a.minmax.slot(grp * 64) != mm_slot) {
            depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
            mm_slot = a.minmax.slot(grp * 64);
        }
        const long long r = grp * 64 + lane;
        const bool live = r < a.R;
        const long long rc = live ? r : a.R - 1;  // idle lanes shadow the last ray (stores masked)
        const float ox = a.origins[rc * 3], oy = a.origins[rc * 3 + 1], oz = a.origins[rc * 3 + 2];
        const float dx = a.dirs[rc * 3], dy = a.dirs[rc * 3 + 1], dz = a.dirs[rc * 3 + 2];
        const float s_near = spacing_fn(a.nears[rc], lin), s_far = spacing_fn(a.fars[rc], lin);
        const float *tb = a.spacing + tn_ws_bin(grp * 64, 0, S) + (rc - grp * 64);  // edge j at tb[j*64]
        // SH(dir) of this lane's ray -> B operands of the 8 SH k-steps (constant over the sample loop)
        float bs0[8], bs1[8];
        {
            float sx = dx, sy = dy, sz = dz;
            if (a.sh_shifted) {
                sx = add_rn(sx, 1.0f) / 2.0f; sy = add_rn(sy, 1.0f) / 2.0f; sz = add_rn(sz, 1.0f) / 2.0f;
            }
            float c[16];
            sh16(sx, sy, sz, c);
#pragma unroll
            for (int s = 0; s < 8; ++s) swap32(c[2 * s], c[2 * s + 1], bs0[s], bs1[s]);
        }
        float en = spacing_to_eucl<true>(tb[(size_t)s0 * 64], s_near, s_far, lin);
        float accum = 0.0f, cum_w = 0.0f;  // sum of delta*sigma before this sample; running sum of weights
        float wsum = 0.0f, wr = 0.0f, wg = 0.0f, wbl = 0.0f, wth = 0.0f, wsteps = 0.0f;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, th = 0.0f, med = 0.0f, step = 0.0f;
        bool med_found = false;
        // the bin edges come from the workspace (HBM / Infinity Cache): the next one is requested a sample ahead, so that its
        // round trip is not the first thing a sample waits for (field kernel 32.0 -> 31.75 ms per 640 k rays at S=192)
        float sb_next = tb[(size_t)(s0 + 1) * 64];
        for (int i = s0; i < s1; ++i) {
            const float st = en;
            en = spacing_to_eucl<true>(sb_next, s_near, s_far, lin);
            sb_next = tb[(size_t)(i + 2 <= S ? i + 2 : S) * 64];
            step = add_rn(st, en) / 2.0f;
            float px, py, pz;
            const float sel = normalize_position<true>(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                 frustum_pos(oz, dz, st, en), px, py, pz);
            float bt0[16], bt1[16];
            // index arithmetic | gathers | interpolation in explicit stages, two groups of LG levels in flight; DENSE: the first
            // kFieldDense levels come from the dense re-layout (4 aligned 16-byte gathers per level instead of 8 8-byte ones)
            hash_encode_pipelined<L16, LG, DENSE ? kFieldDense : 0, TN_EVAL_GATHER_PRIO>(a.g, px, py, pz,
                                                                    [&](int l, float2 f) { swap32(f.x, f.y, bt0[l], bt1[l]); });
#if TN_MFMA_MLP_PRIO
            __builtin_amdgcn_s_setprio(TN_MFMA_MLP_PRIO);
#endif
            f32x16 h1[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                h1[mt][0] = bias_frag(lds + OFF_B_BASE1, mt, h);
                h1[mt][1] = h1[mt][0];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float aw = A[(A_BASE1 + mt * 16 + s) * 64 + lane];
                    MFMA32(h1[mt][0], aw, bt0[s]);
                    MFMA32(h1[mt][1], aw, bt1[s]);
                }
            }
            float g[2][8];
            base2_geo(A, lds + OFF_B_BASE2, lane, h1, g);
            float raw, unused;
            swap32(g[0][0], g[1][0], raw, unused);
            const float dens = mul_rn(mul_rn(a.avg, __expf(raw)), sel);
            {   // colour: [geo | SH] -> 64 -> 64 -> 3
                f32x16 x1[2][2], x2[2][2];
                layer_geo(A, A_C1, lds + OFF_B_C1_EVAL, lane, h, g, x1);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const float aw = A[(A_SH + mt * 8 + s) * 64 + lane];
                        MFMA32(x1[mt][0], aw, bs0[s]);
                        MFMA32(x1[mt][1], aw, bs1[s]);
                    }
                }
                layer64(A, A_C2, lds + OFF_B_C2, lane, h, x1, x2);
                const float *w3 = lds + OFF_W3;
                cr = fast_sigmoid(combine_halves(out_dot_fast<0>(w3, h, x2)) + w3[192]);
                cg = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 64, h, x2)) + w3[193]);
                cb = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 128, h, x2)) + w3[194]);
            }
            {   // thermal: geo -> 64 -> 64 sigmoid -> 1
                f32x16 x1[2][2], x2[2][2];
                layer_geo(A, A_T1, lds + OFF_B_T1, lane, h, g, x1);
                layer64(A, A_T2, lds + OFF_B_T2, lane, h, x1, x2);
                const float *wt = lds + OFF_WTH;
                th = combine_halves(out_dot_fast<1>(wt, h, x2)) + wt[64];
            }
#if TN_MFMA_MLP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            cr = nan_to_num(cr); cg = nan_to_num(cg); cb = nan_to_num(cb); th = nan_to_num(th);  // eval renderers
            // ---- per-lane compositing: NS get_weights + renderers, sequential along the ray ----------------
            const float dd = mul_rn(sub_rn(en, st), dens);
            const float wi = nan_to_num(mul_rn(sub_rn(1.0f, __expf(-dd)), __expf(-accum)));
            accum += dd;
            cum_w += wi;
            if (!med_found && cum_w >= 0.5f) {
                med_found = true;
                med = step;
            }
            wsum += wi;
            wr += mul_rn(wi, cr);
            wg += mul_rn(wi, cg);
            wbl += mul_rn(wi, cb);
            wth += mul_rn(wi, th);
            wsteps += mul_rn(wi, step);
            smin = fminf(smin, step);
            smax = fmaxf(smax, step);
            if (SPLIT) {  // (the launch guarantees no weights output and no early termination in this form)
                a.seg_cum[((size_t)grp * S + i) * 64 + lane] = cum_w;
                continue;
            }
            if (a.out_w && live) a.out_w[r * S + i] = wi;
            // early ray termination (eval, opt-in): wave-wide vote on the transmittance left after this sample
            if (a.early_eps > 0.0f && i + 1 < S && __all(__expf(-accum) < a.early_eps)) {
                // keep the call-global depth bounds exact: they only miss the last mid-point
                const float e0 = spacing_to_eucl<true>(tb[(size_t)(S - 1) * 64], s_near, s_far, lin);
                const float e1 = spacing_to_eucl<true>(tb[(size_t)S * 64], s_near, s_far, lin);
                smax = fmaxf(smax, add_rn(e0, e1) / 2.0f);
                break;
            }
        }
        if (SPLIT) {
            float *rec = a.seg_rec + (size_t)vg * SEG_ROWS * 64 + lane;
            const float row[SEG_ROWS] = {accum, wsum, wr, wg, wbl, wth, wsteps, cr, cg, cb, th, step};
#pragma unroll
            for (int k = 0; k < SEG_ROWS; ++k) rec[k * 64] = row[k];
            continue;
        }
        if (live) {  // cr..th / step now hold the LAST sample: the "last_sample" background
            const float bg = sub_rn(1.0f, wsum);
            const float c0 = add_rn(wr, mul_rn(cr, bg)), c1 = add_rn(wg, mul_rn(cg, bg)), c2 = add_rn(wbl, mul_rn(cb, bg));
            const float ct = add_rn(wth, mul_rn(th, bg));
            a.rgb[r * 3 + 0] = fminf(fmaxf(c0, 0.0f), 1.0f);
            a.rgb[r * 3 + 1] = fminf(fmaxf(c1, 0.0f), 1.0f);
            a.rgb[r * 3 + 2] = fminf(fmaxf(c2, 0.0f), 1.0f);
            a.thermal[r] = fminf(fmaxf(ct, 0.0f), 1.0f);
            a.acc[r] = wsum;
            a.depth[r] = med_found ? med : step;  // searchsorted index clamped to the last sample
            a.expected[r] = wsteps / add_rn(wsum, 1e-10f);
        }
    }
    depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
}

// The records of a ray's segments chained into the renderers' outputs (one lane per ray; see main_mfma_rays_kernel<., true>).
// Segment j's samples carry the weights T_j w^local with T_j = exp(-(optical depth before j)); NS get_weights' nan_to_num applies per
// sample, so a NaN prefix (a NaN density earlier on the ray) zeroes everything behind it, as it does in the serial march.
__global__ void segments_combine_kernel(MfmaArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const long long grp = r >> 6;
    const int lane = (int)(r & 63), S = a.S, K = a.split;
    const bool lin = a.lin != 0;
    float depth_so_far = 0.0f;
    float wsum = 0.0f, wr = 0.0f, wg = 0.0f, wbl = 0.0f, wth = 0.0f, wsteps = 0.0f, med = 0.0f;
    bool med_found = false;
    const float *rec = a.seg_rec + (size_t)grp * K * SEG_ROWS * 64 + lane;
    for (int j = 0; j < K; ++j, rec += SEG_ROWS * 64) {
        float T = __expf(-depth_so_far);
        T = T == T ? T : 0.0f;
        const float total = add_rn(wsum, mul_rn(T, rec[1 * 64]));
        if (!med_found && total >= 0.5f) {  // the crossing lies in this segment: first sample whose global cumulative weight reaches 0.5
            const int s0 = j * a.seg_len, s1 = s0 + a.seg_len < S ? s0 + a.seg_len : S;
            const float *cum = a.seg_cum + ((size_t)grp * S + s0) * 64 + lane;
            int i = s0;
            for (; i < s1 - 1; ++i, cum += 64)
                if (add_rn(wsum, mul_rn(T, cum[0])) >= 0.5f) break;
            const float s_near = spacing_fn(a.nears[r], lin), s_far = spacing_fn(a.fars[r], lin);
            const float *tb = a.spacing + tn_ws_bin(grp * 64, 0, S) + lane;
            med = add_rn(spacing_to_eucl<true>(tb[(size_t)i * 64], s_near, s_far, lin),
                         spacing_to_eucl<true>(tb[(size_t)(i + 1) * 64], s_near, s_far, lin)) / 2.0f;
            med_found = true;
        }
        wsum = total;
        wr = add_rn(wr, mul_rn(T, rec[2 * 64]));
        wg = add_rn(wg, mul_rn(T, rec[3 * 64]));
        wbl = add_rn(wbl, mul_rn(T, rec[4 * 64]));
        wth = add_rn(wth, mul_rn(T, rec[5 * 64]));
        wsteps = add_rn(wsteps, mul_rn(T, rec[6 * 64]));
        depth_so_far += rec[0];
    }
    rec -= SEG_ROWS * 64;  // the last segment's last sample: the "last_sample" background and the clamped median
    const float cr = rec[7 * 64], cg = rec[8 * 64], cb = rec[9 * 64], th = rec[10 * 64], step = rec[11 * 64];
    const float bg = sub_rn(1.0f, wsum);
    const float c0 = add_rn(wr, mul_rn(cr, bg)), c1 = add_rn(wg, mul_rn(cg, bg)), c2 = add_rn(wbl, mul_rn(cb, bg));
    const float ct = add_rn(wth, mul_rn(th, bg));
    a.rgb[r * 3 + 0] = fminf(fmaxf(c0, 0.0f), 1.0f);
    a.rgb[r * 3 + 1] = fminf(fmaxf(c1, 0.0f), 1.0f);
    a.rgb[r * 3 + 2] = fminf(fmaxf(c2, 0.0f), 1.0f);
    a.thermal[r] = fminf(fmaxf(ct, 0.0f), 1.0f);
    a.acc[r] = wsum;
    a.depth[r] = med_found ? med : step;
    a.expected[r] = wsteps / add_rn(wsum, 1e-10f);
}


// ------------------------------------------------------------------------------------------------------
// field_fwd_taped_kernel — the final level's field forward of a TRAINING step in one launch: hash
// encoding -> mlp_base -> density | [SH, geo, appearance] -> mlp_head | geo -> mlp_thermal -> head, for N = rays x samples
// flat samples, with every activation the backward needs written to the tape in the row-major [N, width] layout of the
// stage-by-stage entry points (tn_hash_encode_fwd, tn_linear_fwd ...), which it replaces: nine Linear launches + encode +
// activation, each streaming [N, 64] matrices through HBM, become one MFMA chain whose activations leave the registers
// only as tape stores.  Differences to the eval kernels: per-SAMPLE camera (the appearance embedding enters as 16 k-steps
// of the colour layer, B operands gathered from the embedding table) and per-sample direction (SH as 8 k-steps).
// A C/D register quad r = 4q..4q+3 holds four CONSECUTIVE features (32 mt + 8 q + 4 h + 0..3) of one sample: one 16-byte
// store per quad.
// ------------------------------------------------------------------------------------------------------
struct TapedArgs {
    Grid g;
    tn_space space;
    const float *blob;
    const float *appearance;  // [num_images, 32]
    float avg;
    int sh_shifted;
    const float *positions;   // [N,3]
    const float *dirs;        // [R,3]
    const int *cam;           // [R]
    long long N;
    int n;                    // samples per ray
    float *enc, *sel, *h1, *bo, *density, *c1, *c2, *rgb, *t1, *t2, *thermal;
    const float *ray_bias;    // [R,64]  (TAPE == false) mlp_head.0's bias + SH + appearance part of each ray
    float *jac;               // (TAPE == false, optional) d hash features / d normalised position in pass tiles
                              // [pass][16 levels][3 axes][64 samples][2 features]
};

template <int ACT>  // 1 relu, 2 sigmoid (exact flavour: the tape is what the backward differentiates)
__device__ __forceinline__ void tape_store(float *dst, long long row0, long long N, int lane, int h, f32x16 (&x)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long long row = row0 + 32 * nt + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;
                float *pv = &v.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = x[mt][nt][4 * q + e];
                    const float y = ACT == 1 ? relu_bits(a) : sigmoidf(a);
                    x[mt][nt][4 * q + e] = y;  // the next layer consumes the ACTIVATED value (relu_bits there is idempotent)
                    pv[e] = y;
                }
                if (row < N) *reinterpret_cast<float4 *>(dst + row * 64 + 32 * mt + 8 * q + 4 * h) = v;
            }
        }
    }
}

// TAPE == false (tn_field_fwd_train): nothing but enc / selector / density / rgb / thermal leaves the kernel, and the per-ray
// constant part of the colour layer (SH(direction), appearance embedding: 24 of its 32 k-steps) arrives as a per-ray bias.
#ifndef TN_TRAIN_JAC_LG
// hash levels in flight per group in the Jacobian forward: with its 16 stores per group of four levels on top of two groups of 32
// gathers a wave has more memory operations outstanding than the 6-bit counter holds; two levels per group: 429 -> 413 us at S=192
#define TN_TRAIN_JAC_LG 2
#endif
template <bool TAPE, bool JAC = false>
__global__ void __launch_bounds__(kBlock, 2) field_fwd_taped_kernel(TapedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.blob);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < BLOB_FLOATS / 4; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();
    const float *A = lds + OFF_A;
    const Space sp = make_space(a.space);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const long long passes = (a.N + 63) >> 6;
    const long long stride = (long long)gridDim.x * kWaves;
    for (long long ps = (long long)blockIdx.x * kWaves + wave; ps < passes; ps += stride) {
        const long long row0 = ps * 64;
        const long long i = row0 + lane;
        const bool live = i < a.N;
        const long long ic = live ? i : a.N - 1;
        const long long ray = ic / a.n;
        float px, py, pz;
        const float sel = normalize_position(sp, a.positions[ic * 3], a.positions[ic * 3 + 1], a.positions[ic * 3 + 2], px, py, pz);
        // ---- hash encoding (lane = sample): features to the tape and, through 16 permlane swaps, to the B operands ----
        float bt0[16], bt1[16];
        if (!TAPE) {
            // the hash features in pass tiles [pass][level][64 samples][2]: every store instruction writes 512 contiguous bytes
            // (row-major [N,32] rows put 8 bytes of each of 64 lines on a store); tn_field_bwd_fused reads the same tiling
            float2 *et = reinterpret_cast<float2 *>(a.enc) + ps * (16 * 64) + lane;
            if (JAC) {
                // camera-pose optimisation (round 5): the corner values are in registers here — the backward's position gradient
                // re-read all of them (100 M table reads per S=192 step); 18 more vector instructions per level and 24 more
                // bytes per (sample, level) give it d features / d position instead
                float2 *jt = reinterpret_cast<float2 *>(a.jac) + ps * (16 * 3 * 64) + lane;
                hash_encode_pipelined_raw<L16, TN_TRAIN_JAC_LG, 0, TN_TRAIN_GATHER_PRIO>(a.g, px, py, pz, [&](int l, const HashTaps &t, const float2 (&fc)[8]) {
                    float2 jc[3];
                    const float2 f = hash_blend_jac(t, fc, jc);
                    const float s = a.g.scal[l];
                    et[l * 64] = f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) jt[(l * 3 + c) * 64] = make_float2(jc[c].x * s, jc[c].y * s);
                    swap32(f.x, f.y, bt0[l], bt1[l]);
                });
            } else {
                hash_encode_pipelined<L16, LG, 0, TN_TRAIN_GATHER_PRIO>(a.g, px, py, pz, [&](int l, float2 f) {
                    et[l * 64] = f;
                    swap32(f.x, f.y, bt0[l], bt1[l]);
                });
            }
        } else if (a.g.num_dense == 0) {
            hash_encode_pipelined<L16, 2>(a.g, px, py, pz, [&](int l, float2 f) {
                if (live) *reinterpret_cast<float2 *>(a.enc + ic * 32 + 2 * l) = f;
                swap32(f.x, f.y, bt0[l], bt1[l]);
            });
        } else {
#pragma unroll
            for (int l = 0; l < L16; ++l) {
                const float2 f = encode_level_any<true>(a.g, l, px, py, pz);
                if (live) *reinterpret_cast<float2 *>(a.enc + ic * 32 + 2 * l) = f;
                swap32(f.x, f.y, bt0[l], bt1[l]);
            }
        }
        if (live) a.sel[ic] = sel;
        // ---- mlp_base layer 0 -------------------------------------------------------------------------------------------
        f32x16 h1[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            h1[mt][0] = bias_frag(lds + OFF_B_BASE1, mt, h);
            h1[mt][1] = h1[mt][0];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float aw = A[(A_BASE1 + mt * 16 + s) * 64 + lane];
                MFMA32(h1[mt][0], aw, bt0[s]);
                MFMA32(h1[mt][1], aw, bt1[s]);
            }
        }
        if (TAPE) tape_store<1>(a.h1, row0, a.N, lane, h, h1);
        // ---- mlp_base layer 1: [N,16] = raw density | geo ------------------------------------------------------------
        f32x4 G[4];
        base2_tiles(A, lds + OFF_B_BASE2, lane, h1, G);
        if (TAPE || a.bo) {  // (tn_field_fwd_train: optional — the head launches of tn_field_bwd_fused then skip mlp_base's recomputation)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const long long row = row0 + 16 * T + (lane & 15);
                if (row < a.N) *reinterpret_cast<float4 *>(a.bo + row * 16 + 4 * (lane >> 4)) = float4{G[T][0], G[T][1], G[T][2], G[T][3]};
            }
        }
        float g[2][8];
        geo_relayout(G, g);
        float raw, unused;
        swap32(g[0][0], g[1][0], raw, unused);
        if (live) a.density[ic] = mul_rn(mul_rn(a.avg, TAPE ? expf(raw) : __expf(raw)), sel);
        // ---- colour branch: [geo | SH(dir) | appearance[cam]] -> 64 -> 64 -> 3 ----------------------------------------
        {
            f32x16 x1[2][2], x2[2][2];
            if (TAPE) {
                layer_geo(A, A_C1, lds + OFF_B_C1_RAW, lane, h, g, x1);
                {   // SH of this lane's ray direction: 8 k-steps
                    float sx = a.dirs[ray * 3], sy = a.dirs[ray * 3 + 1], sz = a.dirs[ray * 3 + 2];
                    if (a.sh_shifted) {
                        sx = add_rn(sx, 1.0f) / 2.0f; sy = add_rn(sy, 1.0f) / 2.0f; sz = add_rn(sz, 1.0f) / 2.0f;
                    }
                    float c[16];
                    sh16(sx, sy, sz, c);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        float b0, b1;
                        swap32(c[2 * s], c[2 * s + 1], b0, b1);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const float aw = A[(A_SH + mt * 8 + s) * 64 + lane];
                            MFMA32(x1[mt][0], aw, b0);
                            MFMA32(x1[mt][1], aw, b1);
                        }
                    }
                }
                {   // appearance embedding of this lane's camera: 16 k-steps; A read from the natural [k][f] layout of W_app
                    const float4 *emb = reinterpret_cast<const float4 *>(a.appearance + (long long)a.cam[ray] * APP);
#pragma unroll
                    for (int s4 = 0; s4 < APP / 4; ++s4) {
                        const float4 e = emb[s4];
                        float b0, b1, b2, b3;
                        swap32(e.x, e.y, b0, b1);
                        swap32(e.z, e.w, b2, b3);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            const float a0 = lds[OFF_W_APP + (4 * s4 + h) * 64 + 32 * mt + (lane & 31)];
                            const float a1 = lds[OFF_W_APP + (4 * s4 + 2 + h) * 64 + 32 * mt + (lane & 31)];
                            MFMA32(x1[mt][0], a0, b0);
                            MFMA32(x1[mt][1], a0, b1);
                            MFMA32(x1[mt][0], a1, b2);
                            MFMA32(x1[mt][1], a1, b3);
                        }
                    }
                }
            } else {
                // accumulators start from the ray's bias (both N tiles of this lane belong to samples row0 + (l & 31) and + 32)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    long long smp = row0 + 32 * nt + (lane & 31);
                    if (smp >= a.N) smp = a.N - 1;
                    const float *rb = a.ray_bias + (smp / a.n) * 64;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) x1[mt][nt] = bias_frag(rb, mt, h);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const float aw = A[(A_C1 + mt * 8 + s) * 64 + lane];
                        MFMA32(x1[mt][0], aw, g[0][s]);
                        MFMA32(x1[mt][1], aw, g[1][s]);
                    }
                }
            }
            if (TAPE) tape_store<1>(a.c1, row0, a.N, lane, h, x1);
            layer64(A, A_C2, lds + OFF_B_C2, lane, h, x1, x2);
            if (TAPE) tape_store<1>(a.c2, row0, a.N, lane, h, x2);
            const float *w3 = lds + OFF_W3;
            const float o0 = combine_halves(out_dot<0>(w3, h, x2)) + w3[192], o1 = combine_halves(out_dot<0>(w3 + 64, h, x2)) + w3[193],
                        o2 = combine_halves(out_dot<0>(w3 + 128, h, x2)) + w3[194];
            // (untaped: v_exp_f32 / v_rcp_f32 sigmoids, |error| < 3e-7; the backward differentiates its own recomputation)
            const float cr = TAPE ? sigmoidf(o0) : fast_sigmoid(o0), cg = TAPE ? sigmoidf(o1) : fast_sigmoid(o1),
                        cb = TAPE ? sigmoidf(o2) : fast_sigmoid(o2);
            if (live) {
                a.rgb[ic * 3 + 0] = cr;
                a.rgb[ic * 3 + 1] = cg;
                a.rgb[ic * 3 + 2] = cb;
            }
        }
        // ---- thermal branch: geo -> 64 -> 64 sigmoid -> 1 --------------------------------------------------------------
        {
            f32x16 x1[2][2], x2[2][2];
            layer_geo(A, A_T1, lds + OFF_B_T1, lane, h, g, x1);
            if (TAPE) tape_store<1>(a.t1, row0, a.N, lane, h, x1);
            layer64(A, A_T2, lds + OFF_B_T2, lane, h, x1, x2);
            if (TAPE) {
                tape_store<2>(a.t2, row0, a.N, lane, h, x2);
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) x2[mt][nt][r] = fast_sigmoid(x2[mt][nt][r]);
                    }
                }
            }
            // x2 now holds sigmoid(t2): the head is a plain dot product on it
            const float *wt = lds + OFF_WTH;
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4 *>(wt + 32 * mt + 8 * q + 4 * h);
                    const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p0 = fmaf(ww[e], x2[mt][0][4 * q + e], p0);
                        p1 = fmaf(ww[e], x2[mt][1][4 * q + e], p1);
                    }
                }
            }
            const float th = combine_halves(make_float2(p0, p1)) + wt[64];
            if (live) a.thermal[ic] = th;
        }
    }
}

inline bool mfma_supported(const tn_thermal_field *f) {
    return f && f->geo_feat_dim == GF && f->app_dim == APP && f->grid.num_levels == L16;
}

}  // namespace

namespace tn {

int launch_main_mfma(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                     const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                     hipStream_t stream, int split, float *seg_scratch) {
    if (!mfma_supported(field) || !field->prepared) return TN_ERR_UNSUPPORTED;
    MfmaArgs a;
    a.g = tn_make_grid(field->grid);
    a.space = field->space;
    a.blob = field->prepared;
    a.appearance = field->appearance;
    a.avg = field->average_init_density;
    a.sh_shifted = field->sh_shifted;
    a.origins = in->origins; a.dirs = in->directions; a.nears = in->nears; a.fars = in->fars;
    a.cam = in->camera_indices;
    a.spacing = spacing_ws;
    a.R = num_rays; a.S = cfg->num_nerf_samples; a.training = cfg->training; a.lin = cfg->initial_sampler == 1;
    a.rgb = out->rgb; a.acc = out->accumulation; a.depth = out->depth; a.expected = out->expected_depth;
    a.thermal = out->thermal; a.out_w = out->weights[2]; a.minmax = minmax;
    a.early_eps = cfg->training ? 0.0f : fminf(fmaxf(cfg->early_stop_transmittance, 0.0f), 0.25f);
    a.split = 1; a.seg_len = a.S; a.seg_rec = nullptr; a.seg_cum = nullptr;
    const size_t smem = (size_t)LDS_FLOATS * sizeof(float);
    const long long cap = 256LL * 2;  // 2 resident blocks per CU (LDS 77 KB each)
    // 1.5 ms floor of the tile march vs 0.16 ms at 4096 rays
    const bool small_call = tn_render_kernel_form(field, cfg, num_rays, 1) == 2;
    if (!cfg->training && !out->weights[2] && !small_call) {
        // eval: lane = ray (64 consecutive rays per wave), coherent gathers
        const bool dense = a.g.num_dense >= kFieldDense;  // the dense variant reads exactly kFieldDense levels densely
        if (split > 1 && seg_scratch && a.early_eps == 0.0f) {
            const long long groups = (num_rays + 63) / 64;
            a.seg_len = (a.S + split - 1) / split;
            a.split = (a.S + a.seg_len - 1) / a.seg_len;  // (no empty segment)
            a.seg_rec = seg_scratch;
            a.seg_cum = seg_scratch + (size_t)groups * a.split * SEG_ROWS * 64;
            if (!(dense ? tn_ensure_dynamic_lds<main_mfma_rays_kernel<true, true>>(smem) : tn_ensure_dynamic_lds<main_mfma_rays_kernel<false, true>>(smem)))
                return TN_ERR_LAUNCH;
            const long long need = (groups * a.split + kWaves - 1) / kWaves;
            const unsigned grid = (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
            if (dense)
                hipLaunchKernelGGL((main_mfma_rays_kernel<true, true>), dim3(grid), dim3(kBlock), smem, stream, a);
            else
                hipLaunchKernelGGL((main_mfma_rays_kernel<false, true>), dim3(grid), dim3(kBlock), smem, stream, a);
            hipLaunchKernelGGL(segments_combine_kernel, dim3((unsigned)((num_rays + 255) / 256)), dim3(256), 0, stream, a);
            if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
            return TN_OK;
        }
        a.split = 1;
        if (!(dense ? tn_ensure_dynamic_lds<main_mfma_rays_kernel<true>>(smem) : tn_ensure_dynamic_lds<main_mfma_rays_kernel<false>>(smem)))
            return TN_ERR_LAUNCH;
        const long long groups = (num_rays + 63) / 64;
        const long long need = (groups + kWaves - 1) / kWaves;
        const unsigned grid = (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
        if (dense)
            hipLaunchKernelGGL(main_mfma_rays_kernel<true>, dim3(grid), dim3(kBlock), smem, stream, a);
        else
            hipLaunchKernelGGL(main_mfma_rays_kernel<false>, dim3(grid), dim3(kBlock), smem, stream, a);
        if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
        return TN_OK;
    }
    if (!tn_ensure_dynamic_lds<main_mfma_kernel>(smem)) return TN_ERR_LAUNCH;
    const long long need = (num_rays + kWaves - 1) / kWaves;
    const unsigned grid = (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
    hipLaunchKernelGGL(main_mfma_kernel, dim3(grid), dim3(kBlock), smem, stream, a);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

}  // namespace tn

extern "C" {

int tn_field_fwd_taped(const tn_thermal_field *f, const float *positions, const float *directions,
                       const int32_t *camera_indices, int64_t num_rays, int32_t n, float *enc, float *selector, float *h1,
                       float *bo, float *density, float *c1, float *c2, float *rgb, float *t1, float *t2, float *thermal,
                       void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !positions || !directions || !camera_indices) return TN_ERR_NULL;
    if (!enc || !selector || !h1 || !bo || !density || !c1 || !c2 || !rgb || !t1 || !t2 || !thermal) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (!mfma_supported(f) || !f->prepared) return TN_ERR_UNSUPPORTED;
    TapedArgs a;
    a.g = tn_make_grid(f->grid);
    a.space = f->space;
    a.blob = f->prepared;
    a.appearance = f->appearance;
    a.avg = f->average_init_density;
    a.sh_shifted = f->sh_shifted;
    a.positions = positions; a.dirs = directions; a.cam = camera_indices;
    a.N = (long long)num_rays * n; a.n = n;
    a.enc = enc; a.sel = selector; a.h1 = h1; a.bo = bo; a.density = density; a.c1 = c1; a.c2 = c2; a.rgb = rgb;
    a.t1 = t1; a.t2 = t2; a.thermal = thermal; a.ray_bias = nullptr; a.jac = nullptr;
    const size_t smem = (size_t)LDS_FLOATS * sizeof(float);
    if (!tn_ensure_dynamic_lds<field_fwd_taped_kernel<true>>(smem)) return TN_ERR_LAUNCH;
    const long long passes = (a.N + 63) / 64;
    const long long need = (passes + kWaves - 1) / kWaves;
    const unsigned grid = (unsigned)(need < 512 ? (need < 1 ? 1 : need) : 512);
    hipLaunchKernelGGL(field_fwd_taped_kernel<true>, dim3(grid), dim3(kBlock), smem, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

int tn_field_fwd_train(const tn_thermal_field *f, const float *positions, const float *ray_bias, int64_t num_rays, int32_t n,
                       float *enc, float *selector, float *density, float *rgb, float *thermal, float *base_out, float *position_jacobian,
                       void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !positions || !ray_bias || !enc || !selector || !density || !rgb || !thermal) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (!mfma_supported(f) || !f->prepared) return TN_ERR_UNSUPPORTED;
    TapedArgs a;
    a.g = tn_make_grid(f->grid);
    a.space = f->space;
    a.blob = f->prepared;
    a.appearance = f->appearance;
    a.avg = f->average_init_density;
    a.sh_shifted = f->sh_shifted;
    a.positions = positions; a.dirs = nullptr; a.cam = nullptr;
    a.N = (long long)num_rays * n; a.n = n;
    a.enc = enc; a.sel = selector; a.h1 = nullptr; a.bo = base_out; a.density = density; a.c1 = nullptr; a.c2 = nullptr; a.rgb = rgb;
    a.t1 = nullptr; a.t2 = nullptr; a.thermal = thermal; a.ray_bias = ray_bias; a.jac = position_jacobian;
    const size_t smem = (size_t)LDS_FLOATS * sizeof(float);
    if (!tn_ensure_dynamic_lds<field_fwd_taped_kernel<false>>(smem) || !tn_ensure_dynamic_lds<field_fwd_taped_kernel<false, true>>(smem))
        return TN_ERR_LAUNCH;
    const long long passes = (a.N + 63) / 64;
    const long long need = (passes + kWaves - 1) / kWaves;
    const unsigned grid = (unsigned)(need < 512 ? (need < 1 ? 1 : need) : 512);
    if (a.jac) hipLaunchKernelGGL((field_fwd_taped_kernel<false, true>), dim3(grid), dim3(kBlock), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(field_fwd_taped_kernel<false>, dim3(grid), dim3(kBlock), smem, (hipStream_t)stream, a);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

size_t tn_field_prepare_bytes(const tn_thermal_field *field) {
    if (!mfma_supported(field) || tn_check_thermal_field(field) != TN_OK) return 0;
    return (size_t)BLOB_FLOATS * sizeof(float);
}

int tn_field_prepare(const tn_thermal_field *f, void *prepared_dev, size_t bytes, void *stream) {
    if (!f || !prepared_dev) return TN_ERR_NULL;
    TN_TRY(tn_check_thermal_field(f));
    if (!mfma_supported(f)) return TN_ERR_UNSUPPORTED;
    if (bytes < (size_t)BLOB_FLOATS * sizeof(float)) return TN_ERR_WORKSPACE;
    RawField w;
    w.b0w = f->base0.weight; w.b0b = f->base0.bias; w.b1w = f->base1.weight; w.b1b = f->base1.bias;
    w.h0w = f->head0.weight; w.h0b = f->head0.bias; w.h1w = f->head1.weight; w.h1b = f->head1.bias;
    w.h2w = f->head2.weight; w.h2b = f->head2.bias; w.t0w = f->th0.weight; w.t0b = f->th0.bias;
    w.t1w = f->th1.weight; w.t1b = f->th1.bias; w.thw = f->thead.weight; w.thb = f->thead.bias;
    w.appearance = f->appearance; w.num_images = f->num_images; w.use_avg = f->use_average_appearance;
    hipLaunchKernelGGL(field_prepare_kernel, dim3((BLOB_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float *>(prepared_dev));
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

}  // extern "C"
