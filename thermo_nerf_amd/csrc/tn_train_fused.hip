// field_bwd_fused_kernel — the final level's field BACKWARD of a training step without a tape (SURVEY §8f row 2;
// [REF thermo_nerf/thermal_nerf/thermal_field.py:108-201] differentiated).
//
// The taped form (tn_field_fwd_taped + three tn_linear_chain_bwd + tn_color_input_* + tn_density_act_bwd) streams ~1.5 KB of
// activations per sample out of the forward and back into the backward: the chained backward ran at 0.2-0.3 of the matrix
// pipe because it stages [N,64] tiles, not because of its math.  Here the forward keeps only what compositing needs
// (enc, selector, density, rgb, thermal: 38 floats per sample) and the backward RECOMPUTES the five hidden layers from the
// stored hash features, in registers, next to their adjoints:
//
//   * one wave64 owns a tile of 32 consecutive samples, from enc to d_enc, with no block barrier in the loop;
//   * forward and dx chains run TRANSPOSED on the fp32 MFMAs exactly like the eval kernel (tn_render_mfma.hip): weights are
//     the A operand, activations / adjoints the B operand (lane = sample), so a layer's C/D registers are the next layer's B
//     operands; W and W^T fragments are both read from ONE natural-layout copy of each matrix in LDS (row stride odd: a
//     fragment of W walks rows, a fragment of W^T walks columns, both conflict-free);
//   * dW_l = d_l^T x_l has the samples as its K dimension, so both operands need lane = feature: the two [32 samples, 64]
//     matrices go through a wave-private LDS transpose (16-byte row writes, 4-byte column reads), 17 KB per wave; the LDS
//     round trip hides behind the dx MFMAs of the same layer, which are issued in between;
//   * every dW / db of the eight layers accumulates in registers across the wave's tiles (~230 accumulators: one wave per
//     SIMD, 512 registers) and leaves as one slab per block (field_bwd_reduce_kernel sums the slabs);
//   * the per-ray constant part of mlp_head's first layer (SH(direction), appearance embedding: 48 of its 63 inputs) is a
//     per-ray bias [R,64] in the forward; its adjoint is the per-ray sum of the layer's pre-activation gradient (gsum
//     [R,64]), from which the ray-level Linear backward gives dW (SH and appearance columns), db, the embedding gradient
//     and the direction gradient — 4096 rows instead of 786 k.
//
// MFMA conventions (see tn_render_mfma.hip): lane l = (j = l & 31, h = l >> 5).
//   v_mfma_f32_32x32x2_f32  A: lane holds A[i = j][k = h]   B: lane holds B[k = h][n = j]   C/D reg r: [i = crow(r,h)][n = j]
//   v_mfma_f32_16x16x4_f32  A: lane holds A[i = l & 15][k = l >> 4]   B: B[k = l >> 4][n = l & 15]   C/D reg q: [i = 4 (l >> 4) + q][n = l & 15]
#include "tn_field_eval.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

using namespace tn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / TN_WAVE;
constexpr int GF = 15, APP = 32, IN0 = 16 + GF + APP;
constexpr int TS = 32;   // samples per tile
constexpr int LDT = 68;  // row stride of the transpose buffers: 16-byte row writes of 8 lanes cover 32 distinct banks

// ---- LDS: natural-layout weights (floats) --------------------------------------------------------------
constexpr int LD_B0 = 33, LD_B1 = 66, LD_G = 17, LD_GT = 66, LD_64 = 65;
constexpr int O_WB0 = 0;                        // mlp_base.0      [64][33]
constexpr int O_WB1 = O_WB0 + 64 * LD_B0;       // mlp_base.1      [16][66]
constexpr int O_WC0G = O_WB1 + 16 * LD_B1;      // mlp_head.0, geo columns, as [64][17]: column 0 (raw density row) zero, 1.. = geo
constexpr int O_WC0T = O_WC0G + 64 * LD_G;      // the same transposed [16][66]
constexpr int O_WC1 = O_WC0T + 16 * LD_GT;      // mlp_head.1      [64][65]
constexpr int O_WC2 = O_WC1 + 64 * LD_64;       // mlp_head.2      [3][64]
constexpr int O_WT0G = O_WC2 + 3 * 64;          // mlp_thermal.0   [64][17]
constexpr int O_WT0T = O_WT0G + 64 * LD_G;      //                 [16][66]
constexpr int O_WT1 = O_WT0T + 16 * LD_GT;      // mlp_thermal.1   [64][65]
constexpr int O_WTH = O_WT1 + 64 * LD_64;       // thermal head    [64]
constexpr int O_BB0 = O_WTH + 64;               // biases: base.0 [64]
constexpr int O_BB1 = O_BB0 + 64;               // base.1 [16]
constexpr int O_BC1 = O_BB1 + 16;               // head.1 [64]
constexpr int O_BT0 = O_BC1 + 64;               // thermal.0 [64]
constexpr int O_BT1 = O_BT0 + 64;               // thermal.1 [64]
constexpr int W_FLOATS = O_BT1 + 64;
constexpr int O_SCRATCH = (W_FLOATS + 3) & ~3;  // per wave: X [32][68] | D [32][68]
constexpr int SCRATCH_PER_WAVE = 2 * TS * LDT;
constexpr int LDS_FLOATS = O_SCRATCH + kWaves * SCRATCH_PER_WAVE;

// ---- slab (one per block): the parameter gradients in their natural layouts ----------------------------
constexpr int S_WB0 = 0;                   // [64][32]
constexpr int S_BB0 = S_WB0 + 64 * 32;     // [64]
constexpr int S_WB1 = S_BB0 + 64;          // [16][64]
constexpr int S_BB1 = S_WB1 + 16 * 64;     // [16]
constexpr int S_WC0 = S_BB1 + 16;          // [64][16]  column 0 = the raw-density row (dropped), 1.. = geo columns of mlp_head.0
constexpr int S_WC1 = S_WC0 + 64 * 16;     // [64][64]
constexpr int S_BC1 = S_WC1 + 64 * 64;     // [64]
constexpr int S_WC2 = S_BC1 + 64;          // [3][64]
constexpr int S_BC2 = S_WC2 + 3 * 64;      // [4]
constexpr int S_WT0 = S_BC2 + 4;           // [64][16]
constexpr int S_BT0 = S_WT0 + 64 * 16;     // [64]
constexpr int S_WT1 = S_BT0 + 64;          // [64][64]
constexpr int S_BT1 = S_WT1 + 64 * 64;     // [64]
constexpr int S_WTH = S_BT1 + 64;          // [64]
constexpr int S_BTH = S_WTH + 64;          // [4]
constexpr int SLAB_FLOATS = S_BTH + 4;
static_assert(SLAB_FLOATS <= kWaves * SCRATCH_PER_WAVE, "the block's slab image reuses the transpose buffers");
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

constexpr int kFusedBlocks = 256;  // one persistent block per CU (LDS 135 KB, one wave per SIMD)

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int grow(int j, int h) { return (j >> 1) + ((j & 1) ? 4 : 0) + 8 * h; }
// the four samples of k-step s of a 16x16x4 product over a 32-sample tile: slots 0 / 1 (one ds_read lane group) sit 4 rows
// apart = 16 banks apart at a row stride of 68
__device__ __forceinline__ int samp16(int s, int slot) { return (s & 3) + 16 * (s >> 2) + 4 * slot; }

#define MFMA32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (acc), 0, 0, 0)
#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

__device__ __forceinline__ void swap16(float a, float b, float &even, float &odd) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    even = __uint_as_float(r[0]);
    odd = __uint_as_float(r[1]);
}
__device__ __forceinline__ float both_halves_lo(float v) {  // [v.lanes0-31 | v.lanes0-31]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]);
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// accumulator init from a natural-order vector: reg r <- v[32 mt + crow(r, h)]  (LDS or global, 16-byte aligned)
__device__ __forceinline__ f32x16 frag_from(const float *v, int mt, int h) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(v + 32 * mt + 8 * q + 4 * h);
        o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    return o;
}

// a [64 features x 32 samples] matrix in the C/D layout -> rows of a transpose buffer: buf[sample][feature]
template <int ACT>  // 0 none, 1 relu
__device__ __forceinline__ void rows_store(float *buf, int j, int h, const f32x16 (&x)[2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v;
            v.x = ACT ? relu_bits(x[mt][4 * q]) : x[mt][4 * q];
            v.y = ACT ? relu_bits(x[mt][4 * q + 1]) : x[mt][4 * q + 1];
            v.z = ACT ? relu_bits(x[mt][4 * q + 2]) : x[mt][4 * q + 2];
            v.w = ACT ? relu_bits(x[mt][4 * q + 3]) : x[mt][4 * q + 3];
            *reinterpret_cast<float4 *>(buf + j * LDT + 32 * mt + 8 * q + 4 * h) = v;
        }
    }
}
// the 16-row matrices (bo = raw | geo, and its adjoint) in the 16x16 C/D layout: G[T][q] = row 4 (l >> 4) + q, sample 16 T + (l & 15)
__device__ __forceinline__ void rows_store16(float *buf, int lane, const f32x4 (&G)[2]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
        *reinterpret_cast<float4 *>(buf + (16 * T + (lane & 15)) * LDT + 4 * (lane >> 4)) = float4{G[T][0], G[T][1], G[T][2], G[T][3]};
}

// ---- the layers, one 32-sample tile -----------------------------------------------------------------------------------
// out[mt] += W[64][64] . relu?(in): k-step (mi, s) feeds features 32 mi + crow(s, h)
template <bool RELU_IN>
__device__ __forceinline__ void layer64(const float *W, int j, int h, const f32x16 (&in)[2], f32x16 (&out)[2]) {
    const float *wl = W + j * LD_64 + 4 * h;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b = RELU_IN ? relu_bits(in[mi][s]) : in[mi][s];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float a = wl[(32 * mt) * LD_64 + 32 * mi + (s & 3) + 8 * (s >> 2)];
                MFMA32(out[mt], a, b);
            }
        }
    }
}
// out[mi] = W^T . d: k-step (mo, s) feeds OUTPUT features 32 mo + crow(s, h) of the layer
__device__ __forceinline__ void layer64_t(const float *W, int j, int h, const f32x16 (&d)[2], f32x16 (&out)[2]) {
    const float *wl = W + (4 * h) * LD_64 + j;
#pragma unroll
    for (int mo = 0; mo < 2; ++mo) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b = d[mo][s];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float a = wl[(32 * mo + (s & 3) + 8 * (s >> 2)) * LD_64 + 32 * mi];
                MFMA32(out[mi], a, b);
            }
        }
    }
}
// 16 rows (raw | geo, B operands gb[8] of geo_relayout) -> 64: out[mt] += Wg[64][17] . g
__device__ __forceinline__ void layer_geo(const float *Wg, int j, int h, const float (&gb)[8], f32x16 (&out)[2]) {
    const float *wl = Wg + j * LD_G + 8 * h;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float a = wl[(32 * mt) * LD_G + (s >> 1) + ((s & 1) ? 4 : 0)];
            MFMA32(out[mt], a, gb[s]);
        }
    }
}
// 64 -> 16 rows on 16x16x4 tiles: G[T] += Wt[16][66] . relu?(in), Wt[row][feature]
template <bool RELU_IN>
__device__ __forceinline__ void layer_to16(const float *Wt, int lane, const f32x16 (&in)[2], f32x4 (&G)[2]) {
    const int slot = lane >> 4;
    const float *wl = Wt + (lane & 15) * LD_GT + (slot & 1) + 4 * (slot >> 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const float a = wl[32 * mi + 2 * (rp & 1) + 8 * (rp >> 1)];
            float lo, hi;
            if (RELU_IN)
                swap16(relu_bits(in[mi][2 * rp]), relu_bits(in[mi][2 * rp + 1]), lo, hi);
            else
                swap16(in[mi][2 * rp], in[mi][2 * rp + 1], lo, hi);
            MFMA16(G[0], a, lo);
            MFMA16(G[1], a, hi);
        }
    }
}
// the two 16-sample C/D tiles of a 16-row matrix -> its 8 B operands for a 16 -> 64 product (k rows grow(s, h))
__device__ __forceinline__ void geo_relayout(const f32x4 (&G)[2], float (&gb)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) swap16(G[0][q], G[1][q], gb[2 * q], gb[2 * q + 1]);
}
// out[mt] += Wb1^T . d_bo: A[i = h1 feature][k = bo row] = WB1[row][feature]
__device__ __forceinline__ void layer_from16_t(const float *Wb1, int j, int h, const float (&dgb)[8], f32x16 (&out)[2]) {
    const float *wl = Wb1 + (8 * h) * LD_B1 + j;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const float a = wl[((s >> 1) + ((s & 1) ? 4 : 0)) * LD_B1 + 32 * mt];
            MFMA32(out[mt], a, dgb[s]);
        }
    }
}

// ---- weight-gradient products from the transpose buffers (K = the tile's 32 samples) -----------------------------------
// acc[mo][mi] += d^T x for a 64 x 64 layer (D, X: [32][LDT] rows); dbp[mo] += the A operands (lane = output feature)
__device__ __forceinline__ void dw_64x64(const float *D, const float *X, int j, int h, f32x16 (&acc)[2][2], float (&dbp)[2]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int row = (2 * s) * LDT;
        const float a0 = D[row + h * LDT + j], a1 = D[row + h * LDT + 32 + j];
        const float b0 = X[row + h * LDT + j], b1 = X[row + h * LDT + 32 + j];
        MFMA32(acc[0][0], a0, b0);
        MFMA32(acc[0][1], a0, b1);
        MFMA32(acc[1][0], a1, b0);
        MFMA32(acc[1][1], a1, b1);
        dbp[0] += a0;
        dbp[1] += a1;
    }
}
// acc[mo] += d^T x for a 64 x 32 layer (mlp_base.0: x = the hash features)
__device__ __forceinline__ void dw_64x32(const float *D, const float *X, int j, int h, f32x16 (&acc)[2], float (&dbp)[2]) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int row = (2 * s + h) * LDT;
        const float a0 = D[row + j], a1 = D[row + 32 + j];
        const float b0 = X[row + j];
        MFMA32(acc[0], a0, b0);
        MFMA32(acc[1], a1, b0);
        dbp[0] += a0;
        dbp[1] += a1;
    }
}
// 64 outputs x 16 inputs (the geo -> hidden layers): acc[ot] += d[:, 16 ot ..]^T x[:, 0..15]; dbp[ot] += A operands
template <bool BIAS>
__device__ __forceinline__ void dw_64x16(const float *D, const float *X, int lane, f32x4 (&acc)[4], float (&dbp)[4]) {
    const int i = lane & 15, slot = lane >> 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int row = ((s & 3) + 16 * (s >> 2) + 4 * slot) * LDT;
        const float b = X[row + i];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            const float a = D[row + 16 * ot + i];
            MFMA16(acc[ot], a, b);
            if (BIAS) dbp[ot] += a;
        }
    }
}
// 16 outputs x 64 inputs (mlp_base.1): acc[it] += d[:, 0..15]^T x[:, 16 it ..]
__device__ __forceinline__ void dw_16x64(const float *D, const float *X, int lane, f32x4 (&acc)[4], float &dbp) {
    const int i = lane & 15, slot = lane >> 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int row = ((s & 3) + 16 * (s >> 2) + 4 * slot) * LDT;
        const float a = D[row + i];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float b = X[row + 16 * it + i];
            MFMA16(acc[it], a, b);
        }
        dbp += a;
    }
}

struct FusedBwdArgs {
    const float *b0w, *b0b, *b1w, *b1b, *h0w, *h1w, *h1b, *h2w, *t0w, *t0b, *t1w, *t1b, *thw;
    float avg, exp_clamp_min;
    long long N;
    int S, pass_thermal;
    const float *enc;       // [N,32]  hash features of the forward
    const float *sel;       // [N]
    const float *ray_bias;  // [R,64]  mlp_head.0 bias + its SH and appearance columns applied to the ray's constants
    const float *rgb;       // [N,3]   forward output (sigmoid)
    const float *g_rgb;     // [N,3] or nullptr
    const float *g_th;      // [N]   or nullptr
    const float *g_dens;    // [N]
    float *g_enc;           // [N,32]
    float *gsum;            // [R,64]  += per-ray sums of mlp_head.0's pre-activation gradient
    float *slabs;           // [gridDim.x][SLAB_FLOATS]
};

struct TileIn {
    float4 e[4];
    float rgb[3], g_rgb[3], g_th, g_dens, sel;
    unsigned ray;
};

__device__ __forceinline__ void tile_load(TileIn &t, const FusedBwdArgs &a, long long tile, int j, int h) {
    const long long i = tile * TS + j;
    const bool live = i < a.N;
    const long long ic = live ? i : a.N - 1;
    const float4 *ep = reinterpret_cast<const float4 *>(a.enc + ic * 32 + 16 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) t.e[q] = ep[q];
    t.ray = (unsigned)ic / (unsigned)a.S;
    t.sel = a.sel[ic];
    t.g_dens = live ? a.g_dens[ic] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t.rgb[c] = a.g_rgb ? a.rgb[ic * 3 + c] : 0.0f;
        t.g_rgb[c] = (a.g_rgb && live) ? a.g_rgb[ic * 3 + c] : 0.0f;
    }
    t.g_th = (a.g_th && live) ? a.g_th[ic] : 0.0f;
}

// mlp_base.0 on the tile's hash features: h1[mt] = b + W . enc  (lane (j, h) holds features 16 h .. 16 h + 15 of sample j)
__device__ __forceinline__ void base0(const float *lds, int j, int h, const float4 (&e)[4], f32x16 (&h1)[2]) {
    h1[0] = frag_from(lds + O_BB0, 0, h);
    h1[1] = frag_from(lds + O_BB0, 1, h);
    const float *wl = lds + O_WB0 + j * LD_B0 + 16 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float ev[4] = {e[q].x, e[q].y, e[q].z, e[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) MFMA32(h1[mt], wl[(32 * mt) * LD_B0 + 4 * q + u], ev[u]);
        }
    }
}

__global__ void __launch_bounds__(kBlock, 1) field_bwd_fused_kernel(FusedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- stage the weights in their natural layouts (padded row strides) -------------------------------------------------
    for (int e = threadIdx.x; e < 64 * 32; e += kBlock) lds[O_WB0 + (e >> 5) * LD_B0 + (e & 31)] = a.b0w[e];
    for (int e = threadIdx.x; e < 16 * 64; e += kBlock) lds[O_WB1 + (e >> 6) * LD_B1 + (e & 63)] = a.b1w[e];
    for (int e = threadIdx.x; e < 64 * 16; e += kBlock) {
        const int f = e >> 4, row = e & 15;
        const float vc = row >= 1 ? a.h0w[f * IN0 + 16 + row - 1] : 0.0f;
        const float vt = row >= 1 ? a.t0w[f * GF + row - 1] : 0.0f;
        lds[O_WC0G + f * LD_G + row] = vc;
        lds[O_WC0T + row * LD_GT + f] = vc;
        lds[O_WT0G + f * LD_G + row] = vt;
        lds[O_WT0T + row * LD_GT + f] = vt;
    }
    for (int e = threadIdx.x; e < 64 * 64; e += kBlock) {
        lds[O_WC1 + (e >> 6) * LD_64 + (e & 63)] = a.h1w[e];
        lds[O_WT1 + (e >> 6) * LD_64 + (e & 63)] = a.t1w[e];
    }
    for (int e = threadIdx.x; e < 3 * 64; e += kBlock) lds[O_WC2 + e] = a.h2w[e];
    for (int e = threadIdx.x; e < 64; e += kBlock) {
        lds[O_WTH + e] = a.thw[e];
        lds[O_BB0 + e] = a.b0b[e];
        lds[O_BC1 + e] = a.h1b[e];
        lds[O_BT0 + e] = a.t0b[e];
        lds[O_BT1 + e] = a.t1b[e];
        if (e < 16) lds[O_BB1 + e] = a.b1b[e];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    float *X = lds + O_SCRATCH + wave * SCRATCH_PER_WAVE;
    float *D = X + TS * LDT;
    const bool has_rgb = a.g_rgb != nullptr, has_th = a.g_th != nullptr;

    // ---- the step's parameter gradients, accumulated over this wave's tiles ------------------------------------------------
    f32x16 aw_c1[2][2], aw_t1[2][2], aw_b0[2];
    f32x4 aw_c0[4], aw_t0[4], aw_b1[4];
    float aw_c2[3] = {0.0f, 0.0f, 0.0f}, aw_th = 0.0f;                          // lane = input feature
    float ab_c1[2] = {0.0f, 0.0f}, ab_t1[2] = {0.0f, 0.0f}, ab_b0[2] = {0.0f, 0.0f};
    float ab_t0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ab_unused[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ab_b1 = 0.0f;
    float ab_c2[3] = {0.0f, 0.0f, 0.0f}, ab_th = 0.0f;                          // lane = sample partials (h == 0 lanes)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        aw_b0[p] = zero16();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            aw_c1[p][q] = zero16();
            aw_t1[p][q] = zero16();
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        aw_c0[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        aw_t0[p] = aw_c0[p];
        aw_b1[p] = aw_c0[p];
    }

    const long long tiles = (a.N + TS - 1) / TS;
    const long long stride = (long long)gridDim.x * kWaves;
    long long tile = (long long)blockIdx.x * kWaves + wave;
    TileIn cur, nxt;
    if (tile < tiles) tile_load(cur, a, tile, j, h);
    for (; tile < tiles; tile += stride) {
        const long long i0 = tile * TS;
        // ---- mlp_head.0's per-ray part: issued now, consumed after mlp_base ---------------------------------------------
        f32x16 c1[2];
        if (has_rgb) {
            c1[0] = frag_from(a.ray_bias + (size_t)cur.ray * 64, 0, h);
            c1[1] = frag_from(a.ray_bias + (size_t)cur.ray * 64, 1, h);
        }
        // ---- recompute: mlp_base.  h1 is NOT kept across the two heads (32 registers): mlp_base's own adjoint recomputes it.
        f32x4 G[2];
        float gb[8];
        {
            f32x16 h1[2];
            base0(lds, j, h, cur.e, h1);
            const float4 bq = *reinterpret_cast<const float4 *>(lds + O_BB1 + 4 * (lane >> 4));
            G[0] = f32x4{bq.x, bq.y, bq.z, bq.w};
            G[1] = G[0];
            layer_to16<true>(lds + O_WB1, lane, h1, G);
            geo_relayout(G, gb);
        }
        const float raw = both_halves_lo(gb[0]);  // bo row 0 of sample j, in both lane halves
        // adjoint of the 16 rows of bo (16x16 C/D layout), filled branch by branch
        f32x4 dG[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};

        // =================================== colour branch ===============================================================
        if (has_rgb) {
            layer_geo(lds + O_WC0G, j, h, gb, c1);  // + W_geo . geo (row 0's column is zero)
            f32x16 c2[2];
            c2[0] = frag_from(lds + O_BC1, 0, h);
            c2[1] = frag_from(lds + O_BC1, 1, h);
            layer64<true>(lds + O_WC1, j, h, c1, c2);
            // d(rgb pre-activation) = g_rgb . rgb (1 - rgb)
            float d3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d3[c] = cur.g_rgb[c] * (cur.rgb[c] * (1.0f - cur.rgb[c]));
            // ---- mlp_head.2: dW / db on the vector unit (3 output rows), x = relu(c2) through the transpose buffer ---------
            rows_store<1>(X, j, h, c2);
            f32x16 d2[2];  // d(c2 pre-activation) = (W2^T d3) . [c2 > 0]
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(lds + O_WC2 + 32 * mt + 8 * q + 4 * h);
                    const float4 w1 = *reinterpret_cast<const float4 *>(lds + O_WC2 + 64 + 32 * mt + 8 * q + 4 * h);
                    const float4 w2 = *reinterpret_cast<const float4 *>(lds + O_WC2 + 128 + 32 * mt + 8 * q + 4 * h);
                    const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w}, ww2[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = fmaf(ww2[e], d3[2], fmaf(ww1[e], d3[1], ww0[e] * d3[0]));
                        d2[mt][4 * q + e] = c2[mt][4 * q + e] > 0.0f ? g : 0.0f;
                    }
                }
            }
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) ab_c2[c] += d3[c];
            }
            wave_sync();
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                const float x = X[s * LDT + lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) aw_c2[c] = fmaf(read_lane(d3[c], s), x, aw_c2[c]);
            }
            wave_sync();
            // ---- mlp_head.1: stage (d2, relu(c1)), dx chain while the LDS round trip completes, then dW -------------------
            rows_store<0>(D, j, h, d2);
            rows_store<1>(X, j, h, c1);
            f32x16 d1[2] = {zero16(), zero16()};
            layer64_t(lds + O_WC1, j, h, d2, d1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[mt][r] = c1[mt][r] > 0.0f ? d1[mt][r] : 0.0f;
            }
            wave_sync();
            dw_64x64(D, X, j, h, aw_c1, ab_c1);
            wave_sync();
            // ---- mlp_head.0: geo columns here, the per-ray columns through gsum ----------------------------------------------
            rows_store<0>(D, j, h, d1);
            rows_store16(X, lane, G);
            layer_to16<false>(lds + O_WC0T, lane, d1, dG);
            wave_sync();
            dw_64x16<false>(D, X, lane, aw_c0, ab_unused);
            {   // per-ray sums of d1 (lane = feature): one atomic per (ray, feature) and tile
                const unsigned S = (unsigned)a.S;
                unsigned ray = __builtin_amdgcn_readfirstlane((unsigned)(i0 / a.S));
                long long seg_end = (long long)(ray + 1) * S - i0;  // first sample of the tile that belongs to the next ray
                const long long live_n = a.N - i0;                   // samples of the tile inside the batch
                const unsigned R = (unsigned)(a.N / a.S);
                float acc = 0.0f;
#pragma unroll
                for (int s = 0; s < TS; ++s) {
                    if (s == seg_end) {
                        if (ray < R) unsafeAtomicAdd(a.gsum + (size_t)ray * 64 + lane, acc);
                        acc = 0.0f;
                        ++ray;
                        seg_end += S;
                    }
                    if (s < live_n) acc += D[s * LDT + lane];
                }
                if (ray < R) unsafeAtomicAdd(a.gsum + (size_t)ray * 64 + lane, acc);
            }
            wave_sync();
        }

        // =================================== thermal branch ==============================================================
        if (has_th) {
            f32x16 t1[2];
            t1[0] = frag_from(lds + O_BT0, 0, h);
            t1[1] = frag_from(lds + O_BT0, 1, h);
            layer_geo(lds + O_WT0G, j, h, gb, t1);
            f32x16 t2[2];
            t2[0] = frag_from(lds + O_BT1, 0, h);
            t2[1] = frag_from(lds + O_BT1, 1, h);
            layer64<true>(lds + O_WT1, j, h, t1, t2);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) t2[mt][r] = sigmoid_exact(t2[mt][r]);
            }
            const float dth = cur.g_th;
            // ---- thermal head (1 output row): dW on the vector unit, x = sigmoid(t2) -----------------------------------------
            rows_store<0>(X, j, h, t2);
            f32x16 d2[2];  // d(t2 pre-activation) = w_head . dth . s (1 - s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *reinterpret_cast<const float4 *>(lds + O_WTH + 32 * mt + 8 * q + 4 * h);
                    const float ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sg = t2[mt][4 * q + e];
                        d2[mt][4 * q + e] = (ww[e] * dth) * (sg * (1.0f - sg));
                    }
                }
            }
            if (h == 0) ab_th += dth;
            wave_sync();
#pragma unroll
            for (int s = 0; s < TS; ++s) aw_th = fmaf(read_lane(dth, s), X[s * LDT + lane], aw_th);
            wave_sync();
            // ---- mlp_thermal.1 ---------------------------------------------------------------------------------------------
            rows_store<0>(D, j, h, d2);
            rows_store<1>(X, j, h, t1);
            f32x16 d1[2] = {zero16(), zero16()};
            layer64_t(lds + O_WT1, j, h, d2, d1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[mt][r] = t1[mt][r] > 0.0f ? d1[mt][r] : 0.0f;
            }
            wave_sync();
            dw_64x64(D, X, j, h, aw_t1, ab_t1);
            wave_sync();
            // ---- mlp_thermal.0 ---------------------------------------------------------------------------------------------
            rows_store<0>(D, j, h, d1);
            rows_store16(X, lane, G);
            if (a.pass_thermal) layer_to16<false>(lds + O_WT0T, lane, d1, dG);  // REF thermal_field.py:171-172: .detach() otherwise
            wave_sync();
            dw_64x16<true>(D, X, lane, aw_t0, ab_t0);
            wave_sync();
        }

        // =================================== mlp_base ===================================================================
        {
            // row 0 of d_bo: trunc_exp backward, g . exp(clamp(raw)) with the selector and the average density folded in.
            // dG row 0 lives in lanes 0-15 (l >> 4 == 0), q == 0, of both 16-sample tiles.
            const float dr = cur.g_dens * cur.sel * a.avg * expf(fminf(fmaxf(raw, a.exp_clamp_min), 15.0f));  // lane (j, h): sample j
            // sample 16 T + (l & 15) of tile T: lanes 0-15 of `dr` hold samples 0-15, lanes 16-31 samples 16-31
            const float dr_hi = __shfl(dr, (lane & 15) + 16, 64);
            if ((lane >> 4) == 0) {
                dG[0][0] = dr;     // the geo -> hidden weights have a zero column for row 0: nothing was added there
                dG[1][0] = dr_hi;
            }
            // the next tile's inputs: in flight during this section, whose own register needs are small
            const bool more = tile + stride < tiles;
            if (more) tile_load(nxt, a, tile + stride, j, h);
            float4 e4[4];  // this tile's hash features again (L2): x of mlp_base.0's weight gradient, input of the h1 recompute
            {
                const long long ic = (i0 + j < a.N) ? i0 + j : a.N - 1;
                const float4 *ep = reinterpret_cast<const float4 *>(a.enc + ic * 32 + 16 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) e4[q] = ep[q];
            }
            rows_store16(D, lane, dG);
            float dgb[8];
            geo_relayout(dG, dgb);
            f32x16 dh[2] = {zero16(), zero16()};
            layer_from16_t(lds + O_WB1, j, h, dgb, dh);
            {
                f32x16 h1[2];
                base0(lds, j, h, e4, h1);
                rows_store<1>(X, j, h, h1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dh[mt][r] = h1[mt][r] > 0.0f ? dh[mt][r] : 0.0f;
                }
            }
            wave_sync();
            dw_16x64(D, X, lane, aw_b1, ab_b1);
            wave_sync();
            rows_store<0>(D, j, h, dh);
            {   // x = the hash features: lane (j, h) holds features 16 h .. 16 h + 15 of sample j
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(X + j * LDT + 16 * h + 4 * q) = e4[q];
            }
            // d_enc^T [32 features][32 samples] = W_b0^T . dh
            f32x16 de = zero16();
            {
                const float *wl = lds + O_WB0 + (4 * h) * LD_B0 + j;
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) MFMA32(de, wl[(32 * mo + (s & 3) + 8 * (s >> 2)) * LD_B0], dh[mo][s]);
                }
            }
            if (i0 + j < a.N) {
                float *gp = a.g_enc + (i0 + j) * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(gp + 8 * q) = float4{de[4 * q], de[4 * q + 1], de[4 * q + 2], de[4 * q + 3]};
            }
            wave_sync();
            dw_64x32(D, X, j, h, aw_b0, ab_b0);
            wave_sync();
        }
        cur = nxt;
    }

    // ---- block slab: the four waves' accumulators summed through an LDS image of the slab ------------------------------------
    __syncthreads();
    float *img = lds + O_SCRATCH;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += kBlock) img[e] = 0.0f;
    __syncthreads();
    for (int w = 0; w < kWaves; ++w) {
        if (wave == w) {
            auto put = [&](int addr, float v) { img[addr] += v; };
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * mo + crow(r, h);
                    put(S_WB0 + o * 32 + j, aw_b0[mo][r]);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        put(S_WC1 + o * 64 + 32 * mi + j, aw_c1[mo][mi][r]);
                        put(S_WT1 + o * 64 + 32 * mi + j, aw_t1[mo][mi][r]);
                    }
                }
                const float s0 = ab_b0[mo] + __shfl_xor(ab_b0[mo], 32, 64);
                const float s1 = ab_c1[mo] + __shfl_xor(ab_c1[mo], 32, 64);
                const float s2 = ab_t1[mo] + __shfl_xor(ab_t1[mo], 32, 64);
                if (h == 0) {
                    put(S_BB0 + 32 * mo + j, s0);
                    put(S_BC1 + 32 * mo + j, s1);
                    put(S_BT1 + 32 * mo + j, s2);
                }
            }
            const int i16 = lane & 15, rg = lane >> 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    put(S_WC0 + (16 * t + 4 * rg + q) * 16 + i16, aw_c0[t][q]);
                    put(S_WT0 + (16 * t + 4 * rg + q) * 16 + i16, aw_t0[t][q]);
                    put(S_WB1 + (4 * rg + q) * 64 + 16 * t + i16, aw_b1[t][q]);
                }
                float sb = ab_t0[t];
                sb += __shfl_xor(sb, 16, 64);
                sb += __shfl_xor(sb, 32, 64);
                if (rg == 0) put(S_BT0 + 16 * t + i16, sb);
            }
            {
                float sb = ab_b1;
                sb += __shfl_xor(sb, 16, 64);
                sb += __shfl_xor(sb, 32, 64);
                if (rg == 0) put(S_BB1 + i16, sb);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                put(S_WC2 + c * 64 + lane, aw_c2[c]);
                float sb = ab_c2[c];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
                if (lane == 0) put(S_BC2 + c, sb);
            }
            put(S_WTH + lane, aw_th);
            {
                float sb = ab_th;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o, 64);
                if (lane == 0) put(S_BTH, sb);
            }
        }
        __syncthreads();
    }
    float *slab = a.slabs + (size_t)blockIdx.x * SLAB_FLOATS;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += kBlock) slab[e] = img[e];
}

// ---- slab reduction: dst[...] += sum over blocks --------------------------------------------------------------------------------
struct RedSeg {
    int off, rows, cols, src_ld, src_col0;  // slab segment [rows][src_ld], columns src_col0 .. src_col0 + cols - 1 used
    int dst_ld, dst_col0;
    float *dst;
};
constexpr int kRedSegs = 15;
struct RedArgs {
    RedSeg seg[kRedSegs];
    const float *slabs;
    int blocks;
};
constexpr int kRedSplit = 4;

__global__ void __launch_bounds__(kBlock) field_bwd_reduce_kernel(RedArgs a) {
    const RedSeg &sg = a.seg[blockIdx.y];
    if (!sg.dst) return;
    const int total = sg.rows * sg.cols;
    const int per = (a.blocks + kRedSplit - 1) / kRedSplit;
    const int b0 = blockIdx.z * per, b1 = min(a.blocks, b0 + per);
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        const int r = e / sg.cols, c = e - r * sg.cols;
        const float *p = a.slabs + sg.off + r * sg.src_ld + sg.src_col0 + c;
        float s0 = 0.0f, s1 = 0.0f;
        int b = b0;
        for (; b + 1 < b1; b += 2) {
            s0 += p[(size_t)b * SLAB_FLOATS];
            s1 += p[(size_t)(b + 1) * SLAB_FLOATS];
        }
        if (b < b1) s0 += p[(size_t)b * SLAB_FLOATS];
        unsafeAtomicAdd(sg.dst + r * sg.dst_ld + sg.dst_col0 + c, s0 + s1);
    }
}

}  // namespace

extern "C" {

size_t tn_field_bwd_fused_workspace_bytes(void) { return (size_t)kFusedBlocks * SLAB_FLOATS * sizeof(float); }

int tn_field_bwd_fused(const tn_thermal_field *f, int64_t num_rays, int32_t n, const float *enc, const float *selector,
                       const float *ray_bias, const float *rgb, const float *d_rgb, const float *d_thermal,
                       const float *d_density, int32_t pass_thermal_gradients, float trunc_exp_min, float *d_enc, float *d_ray_sum,
                       const tn_field_grads *grads, void *workspace, size_t workspace_bytes, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !enc || !selector || !d_density || !d_enc || !grads || !workspace) return TN_ERR_NULL;
    if (d_rgb && (!rgb || !ray_bias || !d_ray_sum)) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || (long long)num_rays * n > 0x7fffffffLL) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (f->geo_feat_dim != GF || f->app_dim != APP || f->grid.num_levels != 16) return TN_ERR_UNSUPPORTED;
    if (workspace_bytes < tn_field_bwd_fused_workspace_bytes()) return TN_ERR_WORKSPACE;
    FusedBwdArgs a;
    a.b0w = f->base0.weight; a.b0b = f->base0.bias; a.b1w = f->base1.weight; a.b1b = f->base1.bias;
    a.h0w = f->head0.weight; a.h1w = f->head1.weight; a.h1b = f->head1.bias; a.h2w = f->head2.weight;
    a.t0w = f->th0.weight; a.t0b = f->th0.bias; a.t1w = f->th1.weight; a.t1b = f->th1.bias; a.thw = f->thead.weight;
    a.avg = f->average_init_density;
    a.exp_clamp_min = trunc_exp_min;
    a.N = (long long)num_rays * n;
    a.S = n;
    a.pass_thermal = pass_thermal_gradients;
    a.enc = enc; a.sel = selector; a.ray_bias = ray_bias; a.rgb = rgb; a.g_rgb = d_rgb; a.g_th = d_thermal; a.g_dens = d_density;
    a.g_enc = d_enc; a.gsum = d_ray_sum;
    a.slabs = reinterpret_cast<float *>(workspace);
    const size_t smem = (size_t)LDS_FLOATS * sizeof(float);
    if (!tn_ensure_dynamic_lds<field_bwd_fused_kernel>(smem)) return TN_ERR_LAUNCH;
    const long long tiles = (a.N + TS - 1) / TS;
    const long long need = (tiles + kWaves - 1) / kWaves;
    const int blocks = (int)(need < kFusedBlocks ? need : kFusedBlocks);
    hipLaunchKernelGGL(field_bwd_fused_kernel, dim3(blocks), dim3(kBlock), smem, (hipStream_t)stream, a);
    TN_LAUNCH_CHECK();

    RedArgs r;
    r.slabs = a.slabs;
    r.blocks = blocks;
    auto seg = [&](int k, int off, int rows, int cols, int src_ld, int src_col0, float *dst, int dst_ld, int dst_col0) {
        r.seg[k] = RedSeg{off, rows, cols, src_ld, src_col0, dst_ld, dst_col0, dst};
    };
    const bool c = d_rgb != nullptr, t = d_thermal != nullptr;
    seg(0, S_WB0, 64, 32, 32, 0, grads->base0_w, 32, 0);
    seg(1, S_BB0, 1, 64, 64, 0, grads->base0_b, 64, 0);
    seg(2, S_WB1, 16, 64, 64, 0, grads->base1_w, 64, 0);
    seg(3, S_BB1, 1, 16, 16, 0, grads->base1_b, 16, 0);
    seg(4, S_WC0, 64, GF, 16, 1, c ? grads->head0_w : nullptr, IN0, 16);
    seg(5, S_WC1, 64, 64, 64, 0, c ? grads->head1_w : nullptr, 64, 0);
    seg(6, S_BC1, 1, 64, 64, 0, c ? grads->head1_b : nullptr, 64, 0);
    seg(7, S_WC2, 3, 64, 64, 0, c ? grads->head2_w : nullptr, 64, 0);
    seg(8, S_BC2, 1, 3, 4, 0, c ? grads->head2_b : nullptr, 3, 0);
    seg(9, S_WT0, 64, GF, 16, 1, t ? grads->th0_w : nullptr, GF, 0);
    seg(10, S_BT0, 1, 64, 64, 0, t ? grads->th0_b : nullptr, 64, 0);
    seg(11, S_WT1, 64, 64, 64, 0, t ? grads->th1_w : nullptr, 64, 0);
    seg(12, S_BT1, 1, 64, 64, 0, t ? grads->th1_b : nullptr, 64, 0);
    seg(13, S_WTH, 1, 64, 64, 0, t ? grads->thead_w : nullptr, 64, 0);
    seg(14, S_BTH, 1, 1, 4, 0, t ? grads->thead_b : nullptr, 1, 0);
    hipLaunchKernelGGL(field_bwd_reduce_kernel, dim3(4, kRedSegs, kRedSplit), dim3(kBlock), 0, (hipStream_t)stream, r);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"
