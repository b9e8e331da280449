// field_bwd_fused_kernel — the final level's field BACKWARD of a training step without a tape (SURVEY §8f row 2;
// [REF thermo_nerf/thermal_nerf/thermal_field.py:108-201] differentiated).
//
// The taped form (tn_field_fwd_taped + three tn_linear_chain_bwd + tn_color_input_* + tn_density_act_bwd) streams ~1.5 KB of
// activations per sample out of the forward and back into the backward: the chained backward ran at 0.2-0.3 of the matrix
// pipe because it stages [N,64] tiles, not because of its math.  Here the forward keeps only what compositing needs
// (enc, selector, density, rgb, thermal: 38 floats per sample) and the backward RECOMPUTES the five hidden layers from the
// stored hash features, in registers, next to their adjoints:
//
//   * one wave64 owns a tile of 16 consecutive samples, from enc to d_enc, with no block barrier in the loop;
//   * every product runs on v_mfma_f32_16x16x4_f32 (exact fp32), TRANSPOSED as in the eval kernel (tn_render_mfma.hip):
//     weights are the A operand, activations / adjoints the B operand (lane = sample).  In the 16x16 C/D layout register q
//     of a lane holds row 4 (l >> 4) + q — exactly the k index that lane supplies as a B operand — so the four registers of an
//     output block ARE the B operands of four k-steps of the next layer: no re-layout of any kind between layers, forward
//     or backward; the matching A fragments are four consecutive floats of a weight row (one ds_read_b128);
//   * dW_l = d_l^T x_l has the samples as its K dimension, so both operands need lane = feature: the two [16 samples, 64]
//     matrices go through a wave-private LDS transpose (16-byte row writes, 4-byte column reads), 8.7 KB per wave; the LDS
//     round trip hides behind the dx MFMAs of the same layer, which are issued in between;
//   * every dW / db accumulates in registers across the wave's tiles (208 MFMA accumulators for the whole field) and leaves as
//     one slab per block (field_bwd_reduce_kernel sums the slabs);
//   * the per-ray constant part of mlp_head's first layer (SH(direction), appearance embedding: 48 of its 63 inputs) is a
//     per-ray bias [R,64] in the forward; its adjoint is the per-ray sum of the layer's pre-activation gradient (gsum
//     [R,64]), from which the ray-level Linear backward gives dW (SH and appearance columns), db, the embedding gradient
//     and the direction gradient — 4096 rows instead of 786 k.
//
// MODE (bit mask: 1 colour head, 2 thermal head, 4 mlp_base's adjoint): 7 = the whole field in one launch — 208 accumulators
// plus the working set need more than 256 registers: one wave per SIMD, every MFMA result in AGPRs.  Split (the default): three
// launches 1 | 2 | 4 of two waves per SIMD each (80 / 80 / 48 accumulators), the heads' adjoints of mlp_base's 16 outputs
// passing through HBM (64 B per sample and head); each launch recomputes mlp_base's forward (+15 % MFMAs in total).
//
// MFMA convention: lane l = (n = l & 15, sl = l >> 4).
//   A: lane holds A[i = n][k = sl]   B: lane holds B[k = sl][col = n]   C/D reg q: [row = 4 sl + q][col = n]
#include "tn_field_eval.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

using namespace tn;


namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GF = 15, APP = 32, IN0 = 16 + GF + APP;
constexpr int TS = 16;   // samples per tile
constexpr int LDT = 68;  // row stride of the transpose buffers (16-byte row writes: 8 lanes x 4 banks; column reads 4 rows apart: 16 banks)

// ---- LDS weights (floats).  R form [out][in], T form [in][out]; row strides 8 x odd: the 16-byte fragment reads of a
// 16-lane group (rows n, columns 4 sl ..) fall on 64 distinct banks.  The narrow layers' W^T fragments are read 4 bytes at a
// time from the R form (2-way conflicts on 80 of a tile's ~700 reads).
constexpr int LD_B0 = 40, LD_64 = 72, LD_G = 24;
// B6 (round 5): the 64 x 64 layers as MFMA fragments of their three bf16 pieces, [ob 4][k pair 2][piece 3][lane 64] x 16 bytes
constexpr int FRAG_64 = 4 * 2 * 3 * 64 * 4;
constexpr int FRAG_B0 = 4 * 1 * 3 * 64 * 4;  // mlp_base.0 [64][32]: 4 output blocks x 1 k pair (and its transpose: 2 x 2)
template <int MODE, bool B6 = false>
struct Lay {  // a launch stages only the matrices its branches read
    static constexpr bool C = (MODE & 1) != 0, T = (MODE & 2) != 0, B = (MODE & 4) != 0;
    static constexpr int M64 = B6 ? FRAG_64 : 64 * LD_64;
    static constexpr int B0R = 0;                              // mlp_base.0      [64][40]; B6 mlp_base launch: fragments of W | of W^T
    static constexpr int B0T = B0R + FRAG_B0;                  //   (B6 mlp_base launch only)
    static constexpr int B1R = B0R + (B6 && B ? 2 * FRAG_B0 : 64 * LD_B0);  // mlp_base.1      [16][72]
    static constexpr int C0R = B1R + 16 * LD_64;               // mlp_head.0, columns of bo's rows: [64][24], column 0 (raw density) zero
    static constexpr int C1R = C0R + (C ? 64 * LD_G : 0);      // mlp_head.1      [64][72]
    static constexpr int C1T = C1R + (C ? M64 : 0);            //   transposed    [64][72]
    static constexpr int C2 = C1T + (C ? M64 : 0);             // mlp_head.2      [3][64]
    static constexpr int T0R = C2 + (C ? 3 * 64 : 0);          // mlp_thermal.0   [64][24]
    static constexpr int T1R = T0R + (T ? 64 * LD_G : 0);      // mlp_thermal.1   [64][72]
    static constexpr int T1T = T1R + (T ? M64 : 0);            //   transposed    [64][72]
    static constexpr int TH = T1T + (T ? M64 : 0);             // thermal head    [64]
    static constexpr int BB0 = TH + (T ? 64 : 0);              // biases: base.0 [64] | base.1 [16] | head.1 [64] | thermal.0 [64] | thermal.1 [64]
    static constexpr int BB1 = BB0 + 64;
    static constexpr int BC1 = BB1 + 16;
    static constexpr int BT0 = BC1 + (C ? 64 : 0);
    static constexpr int BT1 = BT0 + (T ? 64 : 0);
    static constexpr int W_FLOATS = BT1 + (T ? 64 : 0);
    static constexpr int SCRATCH = (W_FLOATS + 3) & ~3;        // per wave: X [16][68] | D [16][68]
};
constexpr int SCRATCH_PER_WAVE = 2 * TS * LDT;

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
constexpr int IMG_FLOATS = SLAB_FLOATS;    // the block's slab image lives over the weights + transpose buffers once the tiles are done
static_assert(IMG_FLOATS <= Lay<4>::SCRATCH + 8 * SCRATCH_PER_WAVE, "slab image must fit the block's LDS");
static_assert((Lay<7>::SCRATCH + 4 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024, "LDS budget (whole field, 4 waves)");
static_assert((Lay<1>::SCRATCH + 8 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024 && (Lay<2>::SCRATCH + 8 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024,
              "LDS budget (split, 8 waves)");
static_assert((Lay<1, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024 && (Lay<2, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024,
              "LDS budget (split, bf16 pieces, 8 waves)");
static_assert(Lay<1, true>::C1R % 4 == 0 && Lay<1, true>::C1T % 4 == 0 && Lay<2, true>::T1R % 4 == 0 && Lay<2, true>::T1T % 4 == 0 &&
                  Lay<4, true>::B0R % 4 == 0 && Lay<4, true>::B0T % 4 == 0,
              "fragment arrays are read 16 bytes at a time");
static_assert((Lay<4, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * 4 <= 160 * 1024 && IMG_FLOATS <= Lay<4, true>::SCRATCH + 8 * SCRATCH_PER_WAVE,
              "LDS budget (mlp_base launch, bf16 pieces)");

constexpr int kFusedBlocks = 256;  // one persistent block per CU

#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (acc), 0, 0, 0)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ f32x4 ld4(const float *p) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void st4(float *p, const f32x4 &v) { *reinterpret_cast<float4 *>(p) = float4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
__device__ __forceinline__ f32x4 relu4(const f32x4 &v) { return f32x4{relu_bits(v[0]), relu_bits(v[1]), relu_bits(v[2]), relu_bits(v[3])}; }

// ---- products on the C/D layout (x[b][q] = feature 16 b + 4 sl + q of sample n) ---------------------------------------------
// out[ob] += W . act(in): W in R form [16 NOB][LD], NKB input blocks
template <int NOB, int NKB, int LD, bool RELU_IN>
__device__ __forceinline__ void mm(const float *W, int n, int sl, const f32x4 *in, f32x4 *out) {
    // the fragments of k-block kb + 1 are read while the MFMAs of kb issue, and no further ahead: left alone, hipcc hoists all
    // NOB x NKB 16-byte reads of a layer (64 registers) to the top
    const float *wl = W + n * LD + 4 * sl;
    f32x4 a[NOB], an[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) a[ob] = ld4(wl + (16 * ob) * LD);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) an[ob] = ld4(wl + (16 * ob) * LD + 16 * (kb + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 b = RELU_IN ? relu4(in[kb]) : in[kb];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) MFMA16(out[ob], a[ob][q], b[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < NKB) {
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) a[ob] = an[ob];
        }
    }
}
// out[ib] += W^T . d with the fragments of W^T read 4 bytes at a time from the R form W[16 NKB][LD] (narrow layers)
template <int NIB, int NKB, int LD>
__device__ __forceinline__ void mm_t_from_r(const float *W, int n, int sl, const f32x4 *d, f32x4 *out) {
    const float *wl = W + (4 * sl) * LD + n;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) {
#pragma unroll
            for (int q = 0; q < 4; ++q) MFMA16(out[ib], wl[(16 * kb + q) * LD + 16 * ib], d[kb][q]);
        }
    }
}

// ---- the 64 x 64 products on the bf16 matrix cores, fp32-exact (round 5) ------------------------------------------------------------
// Every operand as three bf16 pieces (a = p1 + p2 + p3 exactly, tn_render_h3.hip's BF16x6), the product as the six piece products
// above 2^-24, on v_mfma_f32_16x16x32_bf16: 16 384 flop in 16 cycles where v_mfma_f32_16x16x4_f32 does 2 048 in 32.  K = 32 spans
// two 16-feature blocks; the k order is free as long as A and B agree, so lane (n, sl) supplies k = 8 sl + j as feature
// 16 (2 kp) + 4 sl + j (j < 4) | 16 (2 kp + 1) + 4 sl + j - 4: its C/D registers of the two blocks, as they are.  The weights are
// split once per block into MFMA fragments [ob][kp][piece][lane] of 8 bf16 (one ds_read_b128 each).
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_pair_bf16(float x0, float x1, unsigned (&pk)[3]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {x0, x1};
    pk[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));  // RNE, one v_cvt_pk_bf16_f32
    const f2 r1 = {x0 - __uint_as_float(pk[0] << 16), x1 - __uint_as_float(pk[0] & 0xffff0000u)};  // exact
    pk[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, b2));
    const f2 r2 = {r1[0] - __uint_as_float(pk[1] << 16), r1[1] - __uint_as_float(pk[1] & 0xffff0000u)};  // exact, 8 bits left
    pk[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, b2));
}
// stage a matrix as fragments at `dst`: NOB output blocks x NKP k pairs; element (row, k) = w[row * ld + k], TRANSPOSED: w[k * ld + row]
template <bool TRANSPOSED, int NOB, int NKP>
__device__ __forceinline__ void stage_frag(float *dst, const float *__restrict__ w, int ld, int tid, int threads) {
    u32x4 *d = reinterpret_cast<u32x4 *>(dst);
    for (int e = tid; e < NOB * NKP * 64; e += threads) {  // (ob, kp, lane)
        const int lane = e & 63, kp = (e >> 6) % NKP, ob = (e >> 6) / NKP, n = lane & 15, sl = lane >> 4;
        const int row = 16 * ob + n;
        unsigned pk[4][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int k0 = 16 * (2 * kp + h) + 4 * sl + 2 * pr;
                const float x0 = TRANSPOSED ? w[k0 * ld + row] : w[row * ld + k0];
                const float x1 = TRANSPOSED ? w[(k0 + 1) * ld + row] : w[row * ld + k0 + 1];
                split_pair_bf16(x0, x1, pk[2 * h + pr]);
            }
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) d[((ob * NKP + kp) * 3 + pc) * 64 + lane] = u32x4{pk[0][pc], pk[1][pc], pk[2][pc], pk[3][pc]};
    }
}
#define MFMA_B6(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), (acc), 0, 0, 0)
// out[ob] += W . act(in): 32 NKP inputs (in[2 kp], in[2 kp + 1] = the k pair's two 16-feature blocks), 16 NOB outputs
template <int NOB, int NKP, bool RELU_IN>
__device__ __forceinline__ void mm_b6(const float *W, int lane, const f32x4 *in, f32x4 *out) {
    const u32x4 *wf = reinterpret_cast<const u32x4 *>(W) + lane;
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        const f32x4 x0 = RELU_IN ? relu4(in[2 * kp]) : in[2 * kp], x1 = RELU_IN ? relu4(in[2 * kp + 1]) : in[2 * kp + 1];
        unsigned pk[4][3];
        split_pair_bf16(x0[0], x0[1], pk[0]);
        split_pair_bf16(x0[2], x0[3], pk[1]);
        split_pair_bf16(x1[0], x1[1], pk[2]);
        split_pair_bf16(x1[2], x1[3], pk[3]);
        u32x4 b[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) b[pc] = u32x4{pk[0][pc], pk[1][pc], pk[2][pc], pk[3][pc]};
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const u32x4 a0 = wf[((ob * NKP + kp) * 3 + 0) * 64], a1 = wf[((ob * NKP + kp) * 3 + 1) * 64], a2 = wf[((ob * NKP + kp) * 3 + 2) * 64];
            // small terms first
            MFMA_B6(out[ob], a0, b[2]);
            MFMA_B6(out[ob], a2, b[0]);
            MFMA_B6(out[ob], a1, b[1]);
            MFMA_B6(out[ob], a0, b[1]);
            MFMA_B6(out[ob], a1, b[0]);
            MFMA_B6(out[ob], a0, b[0]);
        }
    }
}
template <bool RELU_IN>
__device__ __forceinline__ void mm64_b6(const float *W, int lane, const f32x4 *in, f32x4 *out) { mm_b6<4, 2, RELU_IN>(W, lane, in, out); }

// a [16 NB features x 16 samples] matrix in the C/D layout -> rows of a transpose buffer: buf[sample][feature]
template <int NB, bool RELU>
__device__ __forceinline__ void rows_store(float *buf, int n, int sl, const f32x4 *x) {
#pragma unroll
    for (int b = 0; b < NB; ++b) st4(buf + n * LDT + 16 * b + 4 * sl, RELU ? relu4(x[b]) : x[b]);
}
// acc[ob][ib] += d^T x over the tile's 16 samples; dbp[ob] += the A operands (lane = output feature, summed over sl at the end)
template <int NOB, int NIB, bool BIAS>
__device__ __forceinline__ void dw(const float *D, const float *X, int n, int sl, f32x4 *acc, float *dbp) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = (s + 4 * sl) * LDT + n;
        float a[NOB], b[NIB];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) a[ob] = D[row + 16 * ob];
#pragma unroll
        for (int ib = 0; ib < NIB; ++ib) b[ib] = X[row + 16 * ib];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) MFMA16(acc[ob * NIB + ib], a[ob], b[ib]);
            if (BIAS) dbp[ob] += a[ob];
        }
    }
}

struct FusedBwdArgs {
    const float *b0w, *b0b, *b1w, *b1b, *h0w, *h1w, *h1b, *h2w, *t0w, *t0b, *t1w, *t1b, *thw;
    float avg, exp_clamp_min;
    long long N;
    int S, pass_thermal;
    const float *enc;       // hash features of the forward in its pass tiles [ceil(N/64)][16 levels][64][2]
    const float *bo;        // [N,16] mlp_base's output rows as the forward computed them (head launches: no recomputation), or nullptr
    const float *sel;       // [N]
    const float *ray_bias;  // [R,64]  mlp_head.0 bias + its SH and appearance columns applied to the ray's constants
    const float *rgb;       // [N,3]   forward output (sigmoid)
    const float *g_rgb;     // [N,3] or nullptr
    const float *g_th;      // [N]   or nullptr
    const float *g_dens;    // [N]
    float *g_enc;           // [N,32]
    float *gsum;            // [R,64]  += per-ray sums of mlp_head.0's pre-activation gradient
    Grid g;                 // the field's hash grid and position normalisation: only for d_pos
    tn_space space;
    const float *jac;       // d hash features / d normalised position of the forward (tn_field_fwd_train), pass tiles
                            // [pass][16][3][64][2], or nullptr: the pose gradient re-reads the table
    const float *positions; // [N,3]   sample positions (d_pos != nullptr)
    float *d_pos;           // [N,3]   d loss / d position through the hash encoding (camera-pose optimisation), or nullptr
    float *g_bo_c, *g_bo_t; // [N,16]  split launches: the colour / thermal head's adjoint of bo's rows (nullptr: that head did not run)
    float *slabs;           // [gridDim.x][SLAB_FLOATS]
};

struct TileIn {
    f32x4 e[2];  // hash features 4 sl .. + 3 and 16 + 4 sl .. + 3 of sample n
    float rgb[3], g_rgb[3], g_th, g_dens, sel;
    unsigned ray;
};

// the tile's hash features (what the recomputed forward starts from) ...
__device__ __forceinline__ void tile_load_enc(f32x4 (&e)[2], const FusedBwdArgs &a, long long tile, int n, int sl) {
    const long long i = tile * TS + n;
    const long long ic = i < a.N ? i : a.N - 1;
    // pass tiles of tn_field_fwd_train: [pass of 64 samples][level][sample][2]; features 4 sl .. = levels 2 sl, 2 sl + 1
    const float2 *et = reinterpret_cast<const float2 *>(a.enc) + (ic >> 6) * (16 * 64) + (ic & 63);
    const float2 f0 = et[(2 * sl) * 64], f1 = et[(2 * sl + 1) * 64], f2 = et[(8 + 2 * sl) * 64], f3 = et[(9 + 2 * sl) * 64];
    e[0] = f32x4{f0.x, f0.y, f1.x, f1.y};
    e[1] = f32x4{f2.x, f2.y, f3.x, f3.y};
}
// ... and the rest of its inputs (output gradients, the forward's rgb, the selector)
template <int MODE>
__device__ __forceinline__ void tile_load_rest(TileIn &t, const FusedBwdArgs &a, long long tile, int n, int sl) {
    const long long i = tile * TS + n;
    const bool live = i < a.N;
    const long long ic = live ? i : a.N - 1;
    t.ray = (unsigned)ic / (unsigned)a.S;
    if (MODE & 4) {
        t.sel = a.sel[ic];
        t.g_dens = live ? a.g_dens[ic] : 0.0f;
    }
    if (MODE & 2) t.g_th = (a.g_th && live) ? a.g_th[ic] : 0.0f;
    if (MODE & 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t.rgb[c] = a.g_rgb ? a.rgb[ic * 3 + c] : 0.0f;
            t.g_rgb[c] = (a.g_rgb && live) ? a.g_rgb[ic * 3 + c] : 0.0f;
        }
    }
}
template <int MODE>
__device__ __forceinline__ void tile_load(TileIn &t, const FusedBwdArgs &a, long long tile, int n, int sl) {
    tile_load_enc(t.e, a, tile, n, sl);
    tile_load_rest<MODE>(t, a, tile, n, sl);
}

// STORED (head launches of the split form, round 5): mlp_base's 16 output rows come from the forward's copy (a.bo, 64 B per
// sample) instead of being recomputed from the hash features — 48 of a head tile's ~330 MFMAs, the 128 B feature read and the
// staging of mlp_base's weights go away.
// B6 (round 5): the products whose K is a multiple of 32 features — a head's two 64 x 64 ones (forward and dx of its second layer),
// mlp_base.0's forward (K = 32 hash features) and dx (K = 64) — on the bf16 matrix cores as fp32-exact six-product splits (mm_b6);
// the weight-gradient products (K = the tile's 16 samples) and the 16-wide layers stay on the fp32 MFMA.
template <int MODE, int WAVES, bool STORED = false, bool B6 = false>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) field_bwd_fused_kernel(FusedBwdArgs a) {
    constexpr bool COLOUR = (MODE & 1) != 0, THERMAL = (MODE & 2) != 0, BASE = (MODE & 4) != 0;
    static_assert(!STORED || !BASE, "the mlp_base launch needs its hidden layer: it recomputes");
    constexpr int kThreads = WAVES * 64;
    using L = Lay<MODE, B6>;
    constexpr int O_B0R = L::B0R, O_B1R = L::B1R, O_C0R = L::C0R, O_C1R = L::C1R, O_C1T = L::C1T, O_C2 = L::C2, O_T0R = L::T0R,
                  O_T1R = L::T1R, O_T1T = L::T1T, O_TH = L::TH, O_BB0 = L::BB0, O_BB1 = L::BB1, O_BC1 = L::BC1, O_BT0 = L::BT0,
                  O_BT1 = L::BT1, O_SCRATCH = L::SCRATCH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- stage the weights -----------------------------------------------------------------------------------------------
    if (!STORED) {
        if (B6 && BASE) {
            stage_frag<false, 4, 1>(lds + O_B0R, a.b0w, 32, threadIdx.x, kThreads);
            stage_frag<true, 2, 2>(lds + L::B0T, a.b0w, 32, threadIdx.x, kThreads);
        } else {
            for (int e = threadIdx.x; e < 64 * 32; e += kThreads) lds[O_B0R + (e >> 5) * LD_B0 + (e & 31)] = a.b0w[e];
        }
        for (int e = threadIdx.x; e < 16 * 64; e += kThreads) lds[O_B1R + (e >> 6) * LD_64 + (e & 63)] = a.b1w[e];
    }
    for (int e = threadIdx.x; e < 64 * 16; e += kThreads) {
        const int f = e >> 4, row = e & 15;
        if (COLOUR) lds[O_C0R + f * LD_G + row] = row >= 1 ? a.h0w[f * IN0 + 16 + row - 1] : 0.0f;
        if (THERMAL) lds[O_T0R + f * LD_G + row] = row >= 1 ? a.t0w[f * GF + row - 1] : 0.0f;
    }
    if (B6) {
        if (COLOUR) {
            stage_frag<false, 4, 2>(lds + O_C1R, a.h1w, 64, threadIdx.x, kThreads);
            stage_frag<true, 4, 2>(lds + O_C1T, a.h1w, 64, threadIdx.x, kThreads);
        }
        if (THERMAL) {
            stage_frag<false, 4, 2>(lds + O_T1R, a.t1w, 64, threadIdx.x, kThreads);
            stage_frag<true, 4, 2>(lds + O_T1T, a.t1w, 64, threadIdx.x, kThreads);
        }
    } else {
        for (int e = threadIdx.x; e < 64 * 64; e += kThreads) {
            const int o = e >> 6, i = e & 63;
            if (COLOUR) {
                const float v = a.h1w[e];
                lds[O_C1R + o * LD_64 + i] = v;
                lds[O_C1T + i * LD_64 + o] = v;
            }
            if (THERMAL) {
                const float v = a.t1w[e];
                lds[O_T1R + o * LD_64 + i] = v;
                lds[O_T1T + i * LD_64 + o] = v;
            }
        }
    }
    if (COLOUR)
        for (int e = threadIdx.x; e < 3 * 64; e += kThreads) lds[O_C2 + e] = a.h2w[e];
    for (int e = threadIdx.x; e < 64; e += kThreads) {
        lds[O_BB0 + e] = a.b0b[e];
        if (e < 16) lds[O_BB1 + e] = a.b1b[e];
        if (COLOUR) lds[O_BC1 + e] = a.h1b[e];
        if (THERMAL) {
            lds[O_TH + e] = a.thw[e];
            lds[O_BT0 + e] = a.t0b[e];
            lds[O_BT1 + e] = a.t1b[e];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, sl = lane >> 4;
    float *X = lds + O_SCRATCH + wave * SCRATCH_PER_WAVE;
    float *D = X + TS * LDT;
    const bool has_rgb = COLOUR && a.g_rgb != nullptr, has_th = THERMAL && a.g_th != nullptr;

    // ---- the step's parameter gradients, accumulated over this wave's tiles (acc[ob * NIB + ib][q] = dW[16 ob + 4 sl + q][16 ib + n])
    f32x4 aw_c1[16], aw_t1[16], aw_b0[8], aw_c0[4], aw_t0[4], aw_b1[4];
    float aw_c2[3] = {0.0f, 0.0f, 0.0f}, aw_th = 0.0f;  // lane = input feature
    float ab_c1[4], ab_t1[4], ab_t0[4], ab_b0[4], ab_b1[1] = {0.0f}, ab_none[4];
    float ab_c2[3] = {0.0f, 0.0f, 0.0f}, ab_th = 0.0f;  // lane = sample partials (sl == 0 lanes)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        aw_c1[p] = zero4();
        aw_t1[p] = zero4();
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) aw_b0[p] = zero4();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        aw_c0[p] = zero4(); aw_t0[p] = zero4(); aw_b1[p] = zero4();
        ab_c1[p] = 0.0f; ab_t1[p] = 0.0f; ab_t0[p] = 0.0f; ab_b0[p] = 0.0f; ab_none[p] = 0.0f;
    }

    const long long tiles = (a.N + TS - 1) / TS;
    const long long stride = (long long)gridDim.x * WAVES;
    long long tile = (long long)blockIdx.x * WAVES + wave;
    const unsigned R = (unsigned)(a.N / a.S);
    TileIn cur, nxt;
    // BASE launch: the next tile's inputs are requested at the start of the mlp_base section and are in flight under its products.
    // Head launches (80 accumulators + their working set in 256 registers: no room for a second TileIn): the hash features are
    // only read by the first product of a tile, so the NEXT tile's features are requested into the same registers right after
    // it; the tile's remaining inputs (output gradients, rgb) are requested at the top of the tile, a hundred MFMAs before
    // they are read.  No load is issued right in front of its use.
    f32x4 bo_next = zero4();  // STORED: the next tile's rows of bo, requested a tile ahead
    auto load_bo = [&](long long t) {
        const long long i = t * TS + n;
        return ld4(a.bo + (i < a.N ? i : a.N - 1) * 16 + 4 * sl);
    };
    if (tile < tiles) {
        if (BASE) tile_load<MODE>(cur, a, tile, n, sl);
        else if (STORED) bo_next = load_bo(tile);
        else tile_load_enc(cur.e, a, tile, n, sl);
    }
    for (; tile < tiles; tile += stride) {
        const long long i0 = tile * TS;
        const bool live = i0 + n < a.N;
        if (!BASE) tile_load_rest<MODE>(cur, a, tile, n, sl);
        // ---- mlp_head.0's per-ray part: issued now, consumed after mlp_base ---------------------------------------------
        f32x4 c1[4];
        if (has_rgb) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) c1[ob] = ld4(a.ray_bias + (size_t)cur.ray * 64 + 16 * ob + 4 * sl);
        }
        // ---- recompute: mlp_base -------------------------------------------------------------------------------------------
        f32x4 h1[4];
        f32x4 G[1];   // G[0][q] = row 4 sl + q of bo (0 = raw density, 1.. = geo), sample n
        if (STORED) {
            G[0] = bo_next;
            if (tile + stride < tiles) bo_next = load_bo(tile + stride);
        } else {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) h1[ob] = ld4(lds + O_BB0 + 16 * ob + 4 * sl);
            if (B6 && BASE) mm_b6<4, 1, false>(lds + O_B0R, lane, cur.e, h1);
            else mm<4, 2, LD_B0, false>(lds + O_B0R, n, sl, cur.e, h1);
            if (!BASE && tile + stride < tiles) tile_load_enc(cur.e, a, tile + stride, n, sl);
            G[0] = ld4(lds + O_BB1 + 4 * sl);
            mm<1, 4, LD_64, true>(lds + O_B1R, n, sl, h1, G);
        }
        // adjoint of bo's 16 rows, same layout; the heads add their parts
        f32x4 dG[1] = {zero4()};
        if (MODE == 4) {  // split launches: the heads ran before
            const long long ic = live ? i0 + n : a.N - 1;
            if (a.g_bo_c) dG[0] = ld4(a.g_bo_c + ic * 16 + 4 * sl);
            if (a.g_bo_t) dG[0] += ld4(a.g_bo_t + ic * 16 + 4 * sl);
            if (!live) dG[0] = zero4();  // samples beyond the batch contribute nothing
        }

        // =================================== colour head =================================================================
        if (has_rgb) {
            mm<4, 1, LD_G, false>(lds + O_C0R, n, sl, G, c1);  // + W_geo . geo (the raw-density row's column is zero)
            f32x4 c2[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) c2[ob] = ld4(lds + O_BC1 + 16 * ob + 4 * sl);
            if (B6) mm64_b6<true>(lds + O_C1R, lane, c1, c2);
            else mm<4, 4, LD_64, true>(lds + O_C1R, n, sl, c1, c2);
            float d3[3];  // d(rgb pre-activation) = g_rgb . rgb (1 - rgb)
#pragma unroll
            for (int c = 0; c < 3; ++c) d3[c] = cur.g_rgb[c] * (cur.rgb[c] * (1.0f - cur.rgb[c]));
            // ---- mlp_head.2 (3 output rows): dW / db on the vector unit, x = relu(c2) through the transpose buffer ---------
            rows_store<4, true>(X, n, sl, c2);
            f32x4 d2[4];  // d(c2 pre-activation) = (W2^T d3) . [c2 > 0]
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const f32x4 w0 = ld4(lds + O_C2 + 16 * ob + 4 * sl), w1 = ld4(lds + O_C2 + 64 + 16 * ob + 4 * sl),
                            w2 = ld4(lds + O_C2 + 128 + 16 * ob + 4 * sl);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float g = fmaf(w2[q], d3[2], fmaf(w1[q], d3[1], w0[q] * d3[0]));
                    d2[ob][q] = c2[ob][q] > 0.0f ? g : 0.0f;
                }
            }
            if (sl == 0) {
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
            // ---- mlp_head.1: stage (d2, relu(c1)), dx while the LDS round trip completes, then dW --------------------------
            rows_store<4, false>(D, n, sl, d2);
            rows_store<4, true>(X, n, sl, c1);
            f32x4 d1[4] = {zero4(), zero4(), zero4(), zero4()};
            if (B6) mm64_b6<false>(lds + O_C1T, lane, d2, d1);
            else mm<4, 4, LD_64, false>(lds + O_C1T, n, sl, d2, d1);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int q = 0; q < 4; ++q) d1[ob][q] = c1[ob][q] > 0.0f ? d1[ob][q] : 0.0f;
            }
            wave_sync();
            dw<4, 4, true>(D, X, n, sl, aw_c1, ab_c1);
            wave_sync();
            // ---- mlp_head.0: columns of bo's rows here, the per-ray columns through gsum ----------------------------------------
            rows_store<4, false>(D, n, sl, d1);
            rows_store<1, false>(X, n, sl, G);
            mm_t_from_r<1, 4, LD_G>(lds + O_C0R, n, sl, d1, dG);
            wave_sync();
            dw<4, 1, false>(D, X, n, sl, aw_c0, ab_none);
            {   // per-ray sums of d1 (lane = feature): one atomic per (ray, feature) and tile
                const int S = a.S;
                const int i0s = __builtin_amdgcn_readfirstlane((int)i0);
                unsigned ray = (unsigned)i0s / (unsigned)S;
                const int live_n = (int)min((long long)TS, a.N - i0);
                int first = (int)((ray + 1) * (unsigned)S) - i0s;  // samples of the tile that belong to the first ray
                if (first >= live_n) {  // the usual case: one ray
                    float acc = 0.0f;
#pragma unroll
                    for (int s = 0; s < TS; ++s) acc += D[s * LDT + lane];  // rows beyond the batch hold zeros
                    if (ray < R) unsafeAtomicAdd(a.gsum + (size_t)ray * 64 + lane, acc);
                } else {
                    int s = 0;
                    while (s < live_n) {
                        const int end = min(first, live_n);
                        float acc = 0.0f;
                        for (; s < end; ++s) acc += D[s * LDT + lane];
                        if (ray < R) unsafeAtomicAdd(a.gsum + (size_t)ray * 64 + lane, acc);
                        ++ray;
                        first += S;
                    }
                }
            }
            wave_sync();
            if (MODE == 1 && live) st4(a.g_bo_c + (i0 + n) * 16 + 4 * sl, dG[0]);
        }

        // =================================== thermal head ================================================================
        if (has_th) {
            f32x4 t1[4], t2[4];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                t1[ob] = ld4(lds + O_BT0 + 16 * ob + 4 * sl);
                t2[ob] = ld4(lds + O_BT1 + 16 * ob + 4 * sl);
            }
            mm<4, 1, LD_G, false>(lds + O_T0R, n, sl, G, t1);
            if (B6) mm64_b6<true>(lds + O_T1R, lane, t1, t2);
            else mm<4, 4, LD_64, true>(lds + O_T1R, n, sl, t1, t2);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t2[ob][q] = __builtin_amdgcn_rcpf(1.0f + __expf(-t2[ob][q]));  // |error| < 3e-7
            }
            const float dth = cur.g_th;
            // ---- thermal head (1 output row): dW on the vector unit, x = sigmoid(t2) -----------------------------------------
            rows_store<4, false>(X, n, sl, t2);
            f32x4 d2[4];  // d(t2 pre-activation) = w_head . dth . s (1 - s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const f32x4 w = ld4(lds + O_TH + 16 * ob + 4 * sl);
#pragma unroll
                for (int q = 0; q < 4; ++q) d2[ob][q] = (w[q] * dth) * (t2[ob][q] * (1.0f - t2[ob][q]));
            }
            if (sl == 0) ab_th += dth;
            wave_sync();
#pragma unroll
            for (int s = 0; s < TS; ++s) aw_th = fmaf(read_lane(dth, s), X[s * LDT + lane], aw_th);
            wave_sync();
            // ---- mlp_thermal.1 ---------------------------------------------------------------------------------------------
            rows_store<4, false>(D, n, sl, d2);
            rows_store<4, true>(X, n, sl, t1);
            f32x4 d1[4] = {zero4(), zero4(), zero4(), zero4()};
            if (B6) mm64_b6<false>(lds + O_T1T, lane, d2, d1);
            else mm<4, 4, LD_64, false>(lds + O_T1T, n, sl, d2, d1);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int q = 0; q < 4; ++q) d1[ob][q] = t1[ob][q] > 0.0f ? d1[ob][q] : 0.0f;
            }
            wave_sync();
            dw<4, 4, true>(D, X, n, sl, aw_t1, ab_t1);
            wave_sync();
            // ---- mlp_thermal.0 ---------------------------------------------------------------------------------------------
            rows_store<4, false>(D, n, sl, d1);
            rows_store<1, false>(X, n, sl, G);
            if (a.pass_thermal) mm_t_from_r<1, 4, LD_G>(lds + O_T0R, n, sl, d1, dG);  // REF thermal_field.py:171-172: .detach() otherwise
            wave_sync();
            dw<4, 1, true>(D, X, n, sl, aw_t0, ab_t0);
            wave_sync();
            if (MODE == 2 && a.pass_thermal && live) st4(a.g_bo_t + (i0 + n) * 16 + 4 * sl, dG[0]);
        }

        // =================================== mlp_base ===================================================================
        float2 jv[12];  // (BASE, pose gradient from the forward's Jacobian) levels 2 sl, 2 sl + 1, 8 + 2 sl, 9 + 2 sl x 3 axes of sample n
        float pos_x = 0.0f, pos_y = 0.0f, pos_z = 0.0f;
        if (BASE) {
            // the next tile's inputs: in flight during this section
            if (tile + stride < tiles) tile_load<MODE>(nxt, a, tile + stride, n, sl);
            if (a.d_pos) {  // the sample's position: requested here, a hundred MFMAs before position_grad_finish reads it
                const long long ic = live ? i0 + n : a.N - 1;
                pos_x = a.positions[ic * 3]; pos_y = a.positions[ic * 3 + 1]; pos_z = a.positions[ic * 3 + 2];
            }
            if (a.d_pos && a.jac) {
                const long long ic = live ? i0 + n : a.N - 1;
                const float2 *jt = reinterpret_cast<const float2 *>(a.jac) + (ic >> 6) * (16 * 3 * 64) + (ic & 63);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) jv[(2 * b + jj) * 3 + c] = jt[((8 * b + 2 * sl + jj) * 3 + c) * 64];
                    }
                }
            }
            // row 0 of d_bo: trunc_exp backward, g . exp(clamp(raw)) with the selector and the average density folded in
            const float raw = __shfl(G[0][0], n, 64);  // lane n (sl == 0), q == 0: bo row 0 of sample n
            const float dr = cur.g_dens * cur.sel * a.avg * expf(fminf(fmaxf(raw, a.exp_clamp_min), 15.0f));
            if (sl == 0) dG[0][0] = dr;  // the heads' weights have a zero column for row 0: nothing was added there
            rows_store<1, false>(D, n, sl, dG);
            rows_store<4, true>(X, n, sl, h1);
            f32x4 dh[4] = {zero4(), zero4(), zero4(), zero4()};
            mm_t_from_r<4, 1, LD_64>(lds + O_B1R, n, sl, dG, dh);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dh[ob][q] = h1[ob][q] > 0.0f ? dh[ob][q] : 0.0f;
            }
            wave_sync();
            dw<1, 4, true>(D, X, n, sl, aw_b1, ab_b1);
            wave_sync();
            rows_store<4, false>(D, n, sl, dh);
            rows_store<2, false>(X, n, sl, cur.e);
            f32x4 de[2] = {zero4(), zero4()};  // d_enc^T [32 features][16 samples] = W_b0^T . dh
            if (B6) mm_b6<2, 2, false>(lds + L::B0T, lane, dh, de);
            else mm_t_from_r<2, 4, LD_B0>(lds + O_B0R, n, sl, dh, de);
            if (live) {
                st4(a.g_enc + (i0 + n) * 32 + 4 * sl, de[0]);
                st4(a.g_enc + (i0 + n) * 32 + 16 + 4 * sl, de[1]);
            }
            if (a.d_pos) {
                // camera-pose optimisation: d loss / d position of sample n through the encoding.  The lane holds d_enc of four of
                // the sample's 16 levels (2 sl, 2 sl + 1, 8 + 2 sl, 9 + 2 sl): 32 table reads per lane, in flight under the
                // weight-gradient MFMAs below, summed over the sample's four lanes.  (As a kernel of its own — one lane per
                // (sample, level), tn_hash_encode_bwd_input — this was 310 us of pure gather time per step at S=192.)
                const Space sp = make_space(a.space);
                const float x = pos_x, y = pos_y, z = pos_z;
                float px, py, pz;
                const float selp = normalize_position(sp, x, y, z, px, py, pz);
                float gx = 0.0f, gy = 0.0f, gz = 0.0f;
                if (a.jac) {
                    // (round 5) the forward kept d features / d position: 12 streamed 8-byte reads per lane instead of 32 table reads
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const float gex = de[b][2 * jj], gey = de[b][2 * jj + 1];
                            gx = fmaf(gex, jv[(2 * b + jj) * 3 + 0].x, fmaf(gey, jv[(2 * b + jj) * 3 + 0].y, gx));
                            gy = fmaf(gex, jv[(2 * b + jj) * 3 + 1].x, fmaf(gey, jv[(2 * b + jj) * 3 + 1].y, gy));
                            gz = fmaf(gex, jv[(2 * b + jj) * 3 + 2].x, fmaf(gey, jv[(2 * b + jj) * 3 + 2].y, gz));
                        }
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            encode_level_grad(a.g, 8 * b + 2 * sl + jj, px, py, pz, make_float2(de[b][2 * jj], de[b][2 * jj + 1]), gx, gy, gz);
                    }
                }
                gx += __shfl_xor(gx, 16, 64); gy += __shfl_xor(gy, 16, 64); gz += __shfl_xor(gz, 16, 64);
                gx += __shfl_xor(gx, 32, 64); gy += __shfl_xor(gy, 32, 64); gz += __shfl_xor(gz, 32, 64);
                if (sl == 0 && live) {
                    float rx, ry, rz;
                    position_grad_finish(sp, x, y, z, selp, gx, gy, gz, rx, ry, rz);
                    a.d_pos[(i0 + n) * 3] = rx; a.d_pos[(i0 + n) * 3 + 1] = ry; a.d_pos[(i0 + n) * 3 + 2] = rz;
                }
            }
            wave_sync();
            dw<4, 2, true>(D, X, n, sl, aw_b0, ab_b0);
            wave_sync();
            cur = nxt;
        }
    }

    // ---- block slab: the waves' accumulators summed through an LDS image of the slab ---------------------------------------
    __syncthreads();
    float *img = lds;
    for (int e = threadIdx.x; e < IMG_FLOATS; e += kThreads) img[e] = 0.0f;
    __syncthreads();
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
            auto put = [&](int addr, float v) { img[addr] += v; };
            auto put_bias = [&](int addr, float v) {  // lane (n, sl) partials of feature n -> sum over sl
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (sl == 0) put(addr + n, v);
            };
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = 16 * ob + 4 * sl + q;
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        if (COLOUR) put(S_WC1 + o * 64 + 16 * ib + n, aw_c1[ob * 4 + ib][q]);
                        if (THERMAL) put(S_WT1 + o * 64 + 16 * ib + n, aw_t1[ob * 4 + ib][q]);
                    }
                    if (BASE) {
                        put(S_WB0 + o * 32 + n, aw_b0[ob * 2][q]);
                        put(S_WB0 + o * 32 + 16 + n, aw_b0[ob * 2 + 1][q]);
                        put(S_WB1 + (4 * sl + q) * 64 + 16 * ob + n, aw_b1[ob][q]);
                    }
                    if (THERMAL) put(S_WT0 + o * 16 + n, aw_t0[ob][q]);
                    if (COLOUR) put(S_WC0 + o * 16 + n, aw_c0[ob][q]);
                }
                if (COLOUR) put_bias(S_BC1 + 16 * ob, ab_c1[ob]);
                if (THERMAL) {
                    put_bias(S_BT1 + 16 * ob, ab_t1[ob]);
                    put_bias(S_BT0 + 16 * ob, ab_t0[ob]);
                }
                if (BASE) put_bias(S_BB0 + 16 * ob, ab_b0[ob]);
            }
            if (BASE) put_bias(S_BB1, ab_b1[0]);
            auto put_sum = [&](int addr, float v) {  // partials in the sl == 0 lanes, one per sample
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) put(addr, v);
            };
            if (COLOUR) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    put(S_WC2 + c * 64 + lane, aw_c2[c]);
                    put_sum(S_BC2 + c, ab_c2[c]);
                }
            }
            if (THERMAL) {
                put(S_WTH + lane, aw_th);
                put_sum(S_BTH, ab_th);
            }
        }
        __syncthreads();
    }
    float *slab = a.slabs + (size_t)blockIdx.x * SLAB_FLOATS;
    for (int e = threadIdx.x; e < SLAB_FLOATS; e += kThreads) slab[e] = img[e];
}

// ---- slab reduction: dst[...] += sum over blocks --------------------------------------------------------------------------------
struct RedSeg {
    int off, rows, cols, src_ld, src_col0;  // slab segment [rows][src_ld], columns src_col0 .. src_col0 + cols - 1 used
    int dst_ld, dst_col0;
    float *dst;
    const float *slabs;  // the slabs of the launch that accumulated this segment (round 5: ONE reduction behind the split form's
    int blocks;          // three launches instead of one behind each — two 6 us launches less on the step's critical path)
};
constexpr int kRedSegs = 15;
struct RedArgs {
    RedSeg seg[kRedSegs];
};
constexpr int kRedSplit = 8;   // slices of the block range (one atomic per entry and slice)
constexpr int kRedUnroll = 8;  // independent slab reads in flight per thread: the kernel is bound by load latency, not bytes
constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock) field_bwd_reduce_kernel(RedArgs a) {
    const RedSeg &sg = a.seg[blockIdx.y];
    if (!sg.dst) return;
    const int total = sg.rows * sg.cols;
    const int per = (sg.blocks + kRedSplit - 1) / kRedSplit;
    const int b0 = blockIdx.z * per, b1 = min(sg.blocks, b0 + per);
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < total; e += gridDim.x * kBlock) {
        const int r = e / sg.cols, c = e - r * sg.cols;
        const float *p = sg.slabs + sg.off + r * sg.src_ld + sg.src_col0 + c;
        float s = 0.0f;
        for (int b = b0; b < b1; b += kRedUnroll) {
            float v[kRedUnroll];
#pragma unroll
            for (int u = 0; u < kRedUnroll; ++u) v[u] = b + u < b1 ? p[(size_t)(b + u) * SLAB_FLOATS] : 0.0f;
#pragma unroll
            for (int u = 0; u < kRedUnroll; ++u) s += v[u];
        }
        unsafeAtomicAdd(sg.dst + r * sg.dst_ld + sg.dst_col0 + c, s);
    }
}


// ---- mlp_head.0's per-ray part ---------------------------------------------------------------------------------------------------
// 48 of the layer's 63 inputs — SH(direction) [REF thermal_field.py:117-119] and the appearance embedding of the ray's camera
// [REF :121-126] — are constant along a ray: the forward folds them into a per-ray bias, the backward works on per-ray sums.
constexpr int RH_IN = 16 + APP;  // the ray-constant inputs: SH 0..15, appearance 16..47
__device__ __forceinline__ int rh_col(int k) { return k < 16 ? k : GF + k; }  // column of mlp_head.0's weight

__global__ void __launch_bounds__(256) ray_head_fwd_kernel(const float *__restrict__ w, const float *__restrict__ b,
                                                           const float *__restrict__ emb, const float *__restrict__ dirs,
                                                           const int *__restrict__ cam, int num_images, long long R, int sh_shifted,
                                                           float *__restrict__ ray_bias) {
    __shared__ float Wt[RH_IN][64];  // [k][f]
    for (int e = threadIdx.x; e < RH_IN * 64; e += 256) Wt[e >> 6][e & 63] = w[(e & 63) * IN0 + rh_col(e >> 6)];
    __syncthreads();
    const int f = threadIdx.x & 63;
    const float bf = b[f];
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (long long)gridDim.x * 4) {
        float x = dirs[r * 3], y = dirs[r * 3 + 1], z = dirs[r * 3 + 2];
        if (sh_shifted) {
            x = add_rn(x, 1.0f) / 2.0f; y = add_rn(y, 1.0f) / 2.0f; z = add_rn(z, 1.0f) / 2.0f;
        }
        float c[16];
        sh16(x, y, z, c);
        float v = bf;
#pragma unroll
        for (int k = 0; k < 16; ++k) v = fmaf(Wt[k][f], c[k], v);
        const int ci = cam[r];
        if (ci < 0 || ci >= num_images) {  // no embedding row: poison the ray (nn.Embedding would raise) instead of reading elsewhere
            ray_bias[r * 64 + f] = __int_as_float(0x7fc00000);
            continue;
        }
        const float4 *e4 = reinterpret_cast<const float4 *>(emb + (size_t)ci * APP);
#pragma unroll
        for (int k4 = 0; k4 < APP / 4; ++k4) {
            const float4 e = e4[k4];
            v = fmaf(Wt[16 + 4 * k4][f], e.x, v);
            v = fmaf(Wt[17 + 4 * k4][f], e.y, v);
            v = fmaf(Wt[18 + 4 * k4][f], e.z, v);
            v = fmaf(Wt[19 + 4 * k4][f], e.w, v);
        }
        ray_bias[r * 64 + f] = v;
    }
}

// backward on the per-ray sums g[r][f] of the layer's pre-activation gradient (tn_field_bwd_fused's d_ray_sum):
//   d_w[f][col k] += sum_r g[r][f] x[r][k],  d_b[f] += sum_r g[r][f],  d_emb[cam_r][k] += (W^T g[r])[16 + k],
//   d_x [R,64] (optional; columns 0..15 = the SH part, what tn_color_input_bwd turns into the direction gradient)
constexpr int RH_BATCH = 64;
__global__ void __launch_bounds__(256) ray_head_bwd_kernel(const float *__restrict__ w, const float *__restrict__ emb,
                                                           const float *__restrict__ dirs, const int *__restrict__ cam, int num_images,
                                                           long long R, int sh_shifted, const float *__restrict__ g, float *__restrict__ d_w,
                                                           float *__restrict__ d_b, float *__restrict__ d_emb,
                                                           float *__restrict__ d_x) {
    __shared__ float Wt[RH_IN][64];        // [k][f]
    __shared__ float G[RH_BATCH][65];      // [ray][f]
    __shared__ float X[RH_BATCH][RH_IN];   // [ray][k]
    for (int e = threadIdx.x; e < RH_IN * 64; e += 256) Wt[e >> 6][e & 63] = w[(e & 63) * IN0 + rh_col(e >> 6)];
    const int lo = threadIdx.x & 63, part = threadIdx.x >> 6;
    float aw[RH_IN / 4], ab = 0.0f;  // thread (f = lo, part): d_w[f][col(part + 4 j)]
#pragma unroll
    for (int j = 0; j < RH_IN / 4; ++j) aw[j] = 0.0f;
    for (long long r0 = (long long)blockIdx.x * RH_BATCH; r0 < R; r0 += (long long)gridDim.x * RH_BATCH) {
        __syncthreads();
        const long long r = r0 + lo;  // staging: thread (ray lo, part)
        const bool live = r < R && cam[r < R ? r : 0] >= 0 && cam[r < R ? r : 0] < num_images;  // (rays without an embedding row: skipped)
#pragma unroll
        for (int q = 0; q < 16; ++q) G[lo][part * 16 + q] = live ? g[r * 64 + part * 16 + q] : 0.0f;
        if (part == 0) {
            float x = live ? dirs[r * 3] : 0.0f, y = live ? dirs[r * 3 + 1] : 0.0f, z = live ? dirs[r * 3 + 2] : 1.0f;
            if (sh_shifted) {
                x = add_rn(x, 1.0f) / 2.0f; y = add_rn(y, 1.0f) / 2.0f; z = add_rn(z, 1.0f) / 2.0f;
            }
            float c[16];
            sh16(x, y, z, c);
#pragma unroll
            for (int k = 0; k < 16; ++k) X[lo][k] = c[k];
        } else if (part <= 2) {
            const float *e = emb + (size_t)(live ? cam[r] : 0) * APP + (part - 1) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) X[lo][16 + (part - 1) * 16 + k] = e[k];
        }
        __syncthreads();
        // weight / bias gradient partials: thread (f = lo, part)
        for (int rr = 0; rr < RH_BATCH; ++rr) {
            const float gv = G[rr][lo];
#pragma unroll
            for (int j = 0; j < RH_IN / 4; ++j) aw[j] = fmaf(gv, X[rr][part + 4 * j], aw[j]);
            if (part == 0) ab += gv;
        }
        // d_x of ray lo, inputs part + 4 j
        float dx[RH_IN / 4];
#pragma unroll
        for (int j = 0; j < RH_IN / 4; ++j) dx[j] = 0.0f;
        for (int f = 0; f < 64; ++f) {
            const float gv = G[lo][f];
#pragma unroll
            for (int j = 0; j < RH_IN / 4; ++j) dx[j] = fmaf(Wt[part + 4 * j][f], gv, dx[j]);
        }
        __syncthreads();  // every reader of X is done: it now carries d_x to the lanes that issue the atomics
#pragma unroll
        for (int j = 0; j < RH_IN / 4; ++j) X[lo][part + 4 * j] = dx[j];
        __syncthreads();
        // embedding gradient: a wave covers two rays x 32 entries per instruction — up to 64 distinct addresses in flight (with
        // lane = ray and one entry per instruction a wave hits only as many addresses as it has cameras, and same-address
        // atomics retire one at a time: 240 us instead of 15 for 4096 rays on 8 cameras)
        if (d_emb) {
            const int k = threadIdx.x & 31;
            for (int rr = threadIdx.x >> 5; rr < RH_BATCH; rr += 8) {
                if (r0 + rr < R) {
                    const int ci = cam[r0 + rr];
                    if (ci >= 0 && ci < num_images) unsafeAtomicAdd(d_emb + (size_t)ci * APP + k, X[rr][16 + k]);
                }
            }
        }
        if (d_x && live && part == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) d_x[r * 64 + k] = X[lo][k];
        }
    }
#pragma unroll
    for (int j = 0; j < RH_IN / 4; ++j) unsafeAtomicAdd(d_w + lo * IN0 + rh_col(part + 4 * j), aw[j]);
    if (part == 0 && d_b) unsafeAtomicAdd(d_b + lo, ab);
}

// ---- a proposal level of an update step (one step in six after warm-up: the proposal networks take gradient) -------------------
// HashMLPDensityField [REF thermal_nerf_model.py:127-149]: 5-level grid -> Linear(10,16) + ReLU -> Linear(16,1) -> trunc_exp.
// The taped stage chain was tn_hash_encode_fwd + 2 x tn_linear_fwd + tn_density_act_fwd forward and tn_density_act_bwd +
// tn_linear_chain_bwd<2> backward — the latter 225 us per level for 176 MACs per row, because it tiles [N,16] matrices for
// the matrix pipe.  Here: lane = sample, the network's 193 weights as SCALAR operands (constant address space -> s_load), the
// hidden layer recomputed in the backward from the stored hash features, and every weight gradient accumulated per lane in
// registers (193 accumulators), reduced per wave and per block at the end: one atomic per block and entry.
constexpr int PE = 10, PH = 16;  // encoding width (5 levels x 2), hidden width

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// every lane <- the sum over its row of 16 lanes: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return v;
}

__global__ void __launch_bounds__(256)
density_fwd_train_kernel(Grid g, tn_space space, const tn_cfloat *w0, const tn_cfloat *b0, const tn_cfloat *w1, const tn_cfloat *b1,
                         float avg, const float *__restrict__ positions, long long n, float *__restrict__ enc,
                         float *__restrict__ sel_out, float *__restrict__ raw_out, float *__restrict__ density) {
    const Space sp = make_space(space);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float px, py, pz;
        const float sel = normalize_position(sp, positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2], px, py, pz);
        float2 f[PE / 2];
#pragma unroll
        for (int l = 0; l < PE / 2; ++l) f[l] = encode_level<false, false>(g, l, px, py, pz);
#pragma unroll
        for (int l = 0; l < PE / 2; ++l) reinterpret_cast<float2 *>(enc + i * PE)[l] = f[l];
        float o = b1[0];
#pragma unroll
        for (int h = 0; h < PH; ++h) {
            float a = b0[h];
#pragma unroll
            for (int l = 0; l < PE / 2; ++l) {
                a = fmaf(w0[h * PE + 2 * l], f[l].x, a);
                a = fmaf(w0[h * PE + 2 * l + 1], f[l].y, a);
            }
            o = fmaf(w1[h], fmaxf(a, 0.0f), o);
        }
        sel_out[i] = sel;
        raw_out[i] = o;
        density[i] = mul_rn(mul_rn(avg, expf(o)), sel);
    }
}

__global__ void __launch_bounds__(256)
density_bwd_train_kernel(const tn_cfloat *w0, const tn_cfloat *b0, const tn_cfloat *w1, float avg, float clamp_min,
                         const float *__restrict__ enc, const float *__restrict__ raw, const float *__restrict__ sel,
                         const float *__restrict__ g_density, long long n, float *__restrict__ g_enc, float *__restrict__ d_w0,
                         float *__restrict__ d_b0, float *__restrict__ d_w1, float *__restrict__ d_b1) {
    float aw0[PH * PE], ab0[PH], aw1[PH], ab1 = 0.0f;
#pragma unroll
    for (int k = 0; k < PH * PE; ++k) aw0[k] = 0.0f;
#pragma unroll
    for (int h = 0; h < PH; ++h) ab0[h] = aw1[h] = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float e[PE];
#pragma unroll
        for (int l = 0; l < PE / 2; ++l) {
            const float2 t = reinterpret_cast<const float2 *>(enc + i * PE)[l];
            e[2 * l] = t.x;
            e[2 * l + 1] = t.y;
        }
        // NS trunc_exp backward with the selector and the average density folded in
        const float gr = g_density[i] * sel[i] * avg * expf(fminf(fmaxf(raw[i], clamp_min), 15.0f));
        float ge[PE];
#pragma unroll
        for (int k = 0; k < PE; ++k) ge[k] = 0.0f;
        ab1 += gr;
#pragma unroll
        for (int h = 0; h < PH; ++h) {
            float a = b0[h];
#pragma unroll
            for (int k = 0; k < PE; ++k) a = fmaf(w0[h * PE + k], e[k], a);
            const float hid = fmaxf(a, 0.0f);
            aw1[h] = fmaf(gr, hid, aw1[h]);
            const float gh = a > 0.0f ? w1[h] * gr : 0.0f;
            ab0[h] += gh;
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                aw0[h * PE + k] = fmaf(gh, e[k], aw0[h * PE + k]);
                ge[k] = fmaf(w0[h * PE + k], gh, ge[k]);
            }
        }
#pragma unroll
        for (int l = 0; l < PE / 2; ++l) reinterpret_cast<float2 *>(g_enc + i * PE)[l] = make_float2(ge[2 * l], ge[2 * l + 1]);
    }
    // row (16-lane) sums on the DPP path — no LDS round trip: as 193 x 6 ds_bpermute steps with a wait each the epilogue cost
    // more than the loop (50 us) — then the 16 rows of the block through LDS, then one atomic per block and entry
    constexpr int NV = PH * PE + 2 * PH + 1;
    __shared__ float red[16][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = wave * 4 + (lane >> 4);
    const bool leader = (lane & 15) == 0;
#pragma unroll
    for (int k = 0; k < PH * PE; ++k) {
        const float v = row_sum16(aw0[k]);
        if (leader) red[row][k] = v;
    }
#pragma unroll
    for (int h = 0; h < PH; ++h) {
        const float v0 = row_sum16(ab0[h]), v1 = row_sum16(aw1[h]);
        if (leader) {
            red[row][PH * PE + h] = v0;
            red[row][PH * PE + PH + h] = v1;
        }
    }
    {
        const float v = row_sum16(ab1);
        if (leader) red[row][PH * PE + 2 * PH] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += 256) {
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v += red[r][k];
        float *dst = k < PH * PE ? d_w0 + k : (k < PH * PE + PH ? d_b0 + (k - PH * PE) : (k < PH * PE + 2 * PH ? d_w1 + (k - PH * PE - PH) : d_b1));
        if (v != 0.0f) unsafeAtomicAdd(dst, v);
    }
}

}  // namespace

extern "C" {

size_t tn_field_bwd_fused_workspace_bytes(int64_t num_rays, int32_t n) {
    // one slab per persistent block and launch (three in split mode) + the two heads' adjoints of bo's rows, [N,16] each
    return (size_t)3 * kFusedBlocks * SLAB_FLOATS * sizeof(float) + (size_t)2 * num_rays * n * 16 * sizeof(float);
}

int tn_field_bwd_fused(const tn_thermal_field *f, int64_t num_rays, int32_t n, const float *enc, const float *selector,
                       const float *base_out, const float *ray_bias, const float *rgb, const float *d_rgb, const float *d_thermal,
                       const float *d_density, int32_t pass_thermal_gradients, float trunc_exp_min, int32_t split, float *d_enc,
                       float *d_ray_sum, const float *positions, const float *position_jacobian, float *d_positions,
                       const tn_field_grads *grads, void *workspace, size_t workspace_bytes, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !enc || !selector || !d_density || !d_enc || !grads || !workspace) return TN_ERR_NULL;
    if (d_rgb && (!rgb || !ray_bias || !d_ray_sum)) return TN_ERR_NULL;
    if (d_positions && !positions) return TN_ERR_NULL;
    if (num_rays < 0 || n < 1 || (long long)num_rays * n > 0x7fffffffLL) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (f->geo_feat_dim != GF || f->app_dim != APP || f->grid.num_levels != 16) return TN_ERR_UNSUPPORTED;
    if (workspace_bytes < tn_field_bwd_fused_workspace_bytes(num_rays, n)) return TN_ERR_WORKSPACE;
    FusedBwdArgs a;
    a.b0w = f->base0.weight; a.b0b = f->base0.bias; a.b1w = f->base1.weight; a.b1b = f->base1.bias;
    a.h0w = f->head0.weight; a.h1w = f->head1.weight; a.h1b = f->head1.bias; a.h2w = f->head2.weight;
    a.t0w = f->th0.weight; a.t0b = f->th0.bias; a.t1w = f->th1.weight; a.t1b = f->th1.bias; a.thw = f->thead.weight;
    a.avg = f->average_init_density;
    a.exp_clamp_min = trunc_exp_min;
    a.N = (long long)num_rays * n;
    a.S = n;
    a.pass_thermal = pass_thermal_gradients;
    a.enc = enc; a.bo = base_out; a.sel = selector; a.ray_bias = ray_bias; a.rgb = rgb; a.g_rgb = d_rgb; a.g_th = d_thermal; a.g_dens = d_density;
    a.g_enc = d_enc; a.gsum = d_ray_sum;
    a.g = tn_make_grid(f->grid); a.space = f->space; a.positions = positions; a.d_pos = d_positions;
    a.jac = split ? position_jacobian : nullptr;  // (the one-launch form has no registers to spare for it)
    float *slabs = reinterpret_cast<float *>(workspace);
    float *g_bo = slabs + (size_t)3 * kFusedBlocks * SLAB_FLOATS;
    const long long tiles = (a.N + TS - 1) / TS;
    const hipStream_t st = (hipStream_t)stream;
    const bool c = d_rgb != nullptr, t = d_thermal != nullptr;
    int which = 0;
    RedArgs red;
    for (int i = 0; i < kRedSegs; ++i) red.seg[i] = RedSeg{0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, 0};
    // one launch of the branches in `mode`; the segments of the slab it accumulates are noted for the reduction at the end
    auto launch = [&](auto kernel, int mode, int waves, size_t smem) -> int {
        const long long need = (tiles + waves - 1) / waves;
        const int blocks = (int)(need < kFusedBlocks ? need : kFusedBlocks);
        a.slabs = slabs + (size_t)which * kFusedBlocks * SLAB_FLOATS;
        ++which;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(waves * 64), smem, st, a);
        if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
        auto seg = [&](bool mine, int idx, int off, int rows, int cols, int src_ld, int src_col0, float *dst, int dst_ld, int dst_col0) {
            if (mine) red.seg[idx] = RedSeg{off, rows, cols, src_ld, src_col0, dst_ld, dst_col0, dst, a.slabs, blocks};
        };
        const bool cc = c && (mode & 1), tt = t && (mode & 2), bb = (mode & 4) != 0;
        seg(bb, 0, S_WB0, 64, 32, 32, 0, grads->base0_w, 32, 0);
        seg(bb, 1, S_BB0, 1, 64, 64, 0, grads->base0_b, 64, 0);
        seg(bb, 2, S_WB1, 16, 64, 64, 0, grads->base1_w, 64, 0);
        seg(bb, 3, S_BB1, 1, 16, 16, 0, grads->base1_b, 16, 0);
        seg(cc, 4, S_WC0, 64, GF, 16, 1, grads->head0_w, IN0, 16);
        seg(cc, 5, S_WC1, 64, 64, 64, 0, grads->head1_w, 64, 0);
        seg(cc, 6, S_BC1, 1, 64, 64, 0, grads->head1_b, 64, 0);
        seg(cc, 7, S_WC2, 3, 64, 64, 0, grads->head2_w, 64, 0);
        seg(cc, 8, S_BC2, 1, 3, 4, 0, grads->head2_b, 3, 0);
        seg(tt, 9, S_WT0, 64, GF, 16, 1, grads->th0_w, GF, 0);
        seg(tt, 10, S_BT0, 1, 64, 64, 0, grads->th0_b, 64, 0);
        seg(tt, 11, S_WT1, 64, 64, 64, 0, grads->th1_w, 64, 0);
        seg(tt, 12, S_BT1, 1, 64, 64, 0, grads->th1_b, 64, 0);
        seg(tt, 13, S_WTH, 1, 64, 64, 0, grads->thead_w, 64, 0);
        seg(tt, 14, S_BTH, 1, 1, 4, 0, grads->thead_b, 1, 0);
        return TN_OK;
    };
    auto reduce = [&]() -> int {
        hipLaunchKernelGGL(field_bwd_reduce_kernel, dim3(16, kRedSegs, kRedSplit), dim3(kBlock), 0, st, red);
        return hipGetLastError() == hipSuccess ? TN_OK : TN_ERR_LAUNCH;
    };
    a.g_bo_c = a.g_bo_t = nullptr;
    if (!split) {
        constexpr size_t smem = (size_t)(Lay<7>::SCRATCH + 4 * SCRATCH_PER_WAVE) * sizeof(float);
        if (!tn_ensure_dynamic_lds<field_bwd_fused_kernel<7, 4>>(smem)) return TN_ERR_LAUNCH;
        TN_TRY(launch(field_bwd_fused_kernel<7, 4>, 7, 4, smem));
        return reduce();
    }
    constexpr size_t smem1 = (size_t)(Lay<1>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    constexpr size_t smem2 = (size_t)(Lay<2>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    constexpr size_t smem4 = (size_t)(Lay<4>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    if (!tn_ensure_dynamic_lds<field_bwd_fused_kernel<1, 8>>(smem1) || !tn_ensure_dynamic_lds<field_bwd_fused_kernel<2, 8>>(smem2) ||
        !tn_ensure_dynamic_lds<field_bwd_fused_kernel<4, 8>>(smem4))
        return TN_ERR_LAUNCH;
    constexpr size_t smem1b = (size_t)(Lay<1, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    constexpr size_t smem2b = (size_t)(Lay<2, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    const bool b6 = split == 2 && base_out;  // (the bf16-piece products come with the stored-base head launches only)
    if (base_out && (!tn_ensure_dynamic_lds<field_bwd_fused_kernel<1, 8, true>>(smem1) ||
                     !tn_ensure_dynamic_lds<field_bwd_fused_kernel<2, 8, true>>(smem2) ||
                     !tn_ensure_dynamic_lds<field_bwd_fused_kernel<1, 8, true, true>>(smem1b) ||
                     !tn_ensure_dynamic_lds<field_bwd_fused_kernel<2, 8, true, true>>(smem2b)))
        return TN_ERR_LAUNCH;
    if (c) {
        a.g_bo_c = g_bo;
        if (b6) TN_TRY(launch(field_bwd_fused_kernel<1, 8, true, true>, 1, 8, smem1b));
        else if (base_out) TN_TRY(launch(field_bwd_fused_kernel<1, 8, true>, 1, 8, smem1));
        else TN_TRY(launch(field_bwd_fused_kernel<1, 8>, 1, 8, smem1));
    }
    if (t) {
        if (pass_thermal_gradients) a.g_bo_t = g_bo + (size_t)a.N * 16;
        if (b6) TN_TRY(launch(field_bwd_fused_kernel<2, 8, true, true>, 2, 8, smem2b));
        else if (base_out) TN_TRY(launch(field_bwd_fused_kernel<2, 8, true>, 2, 8, smem2));
        else TN_TRY(launch(field_bwd_fused_kernel<2, 8>, 2, 8, smem2));
    }
    constexpr size_t smem4b = (size_t)(Lay<4, true>::SCRATCH + 8 * SCRATCH_PER_WAVE) * sizeof(float);
    if (split == 2) {  // mlp_base.0's two K >= 32 products as bf16 pieces (this launch recomputes mlp_base either way)
        if (!tn_ensure_dynamic_lds<field_bwd_fused_kernel<4, 8, false, true>>(smem4b)) return TN_ERR_LAUNCH;
        TN_TRY(launch(field_bwd_fused_kernel<4, 8, false, true>, 4, 8, smem4b));
    } else {
        TN_TRY(launch(field_bwd_fused_kernel<4, 8>, 4, 8, smem4));
    }
    return reduce();
}

int tn_ray_head_fwd(const tn_thermal_field *f, const float *directions, const int32_t *camera_indices, int64_t num_rays,
                    float *ray_bias, void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !directions || !camera_indices || !ray_bias) return TN_ERR_NULL;
    if (num_rays < 0) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (f->geo_feat_dim != GF || f->app_dim != APP) return TN_ERR_UNSUPPORTED;
    const long long blocks = (num_rays + 3) / 4;
    hipLaunchKernelGGL(ray_head_fwd_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, (hipStream_t)stream,
                       f->head0.weight, f->head0.bias, f->appearance, directions, camera_indices, f->num_images, (long long)num_rays,
                       f->sh_shifted, ray_bias);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_ray_head_bwd(const tn_thermal_field *f, const float *directions, const int32_t *camera_indices, int64_t num_rays,
                    const float *d_ray_sum, float *d_head0_weight, float *d_head0_bias, float *d_appearance, float *d_ray_inputs,
                    void *stream) {
    if (num_rays == 0) return TN_OK;
    if (!f || !directions || !camera_indices || !d_ray_sum || !d_head0_weight) return TN_ERR_NULL;
    if (num_rays < 0) return TN_ERR_SHAPE;
    TN_TRY(tn_check_thermal_field(f));
    if (f->geo_feat_dim != GF || f->app_dim != APP) return TN_ERR_UNSUPPORTED;
    const long long blocks = (num_rays + RH_BATCH - 1) / RH_BATCH;
    hipLaunchKernelGGL(ray_head_bwd_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, (hipStream_t)stream,
                       f->head0.weight, f->appearance, directions, camera_indices, f->num_images, (long long)num_rays, f->sh_shifted, d_ray_sum,
                       d_head0_weight, d_head0_bias, d_appearance, d_ray_inputs);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

static inline const tn_cfloat *cs(const float *p) { return (const tn_cfloat *)p; }  // the kernels read these as scalars

static int density_train_supported(const tn_density_field *f) {
    if (!f) return TN_ERR_NULL;
    TN_TRY(tn_check_density_field(f));
    if (2 * f->grid.num_levels != PE || f->l0.out_dim != PH) return TN_ERR_UNSUPPORTED;
    return TN_OK;
}

int tn_density_fwd_train(const tn_density_field *f, const float *positions, int64_t n, float *enc, float *selector, float *raw,
                         float *density, void *stream) {
    if (n == 0) return TN_OK;
    TN_TRY(density_train_supported(f));
    if (!positions || !enc || !selector || !raw || !density) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(density_fwd_train_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       tn_make_grid(f->grid), f->space, cs(f->l0.weight), cs(f->l0.bias), cs(f->l1.weight),
                       cs(f->l1.bias), f->average_init_density, positions, (long long)n, enc, selector, raw, density);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

int tn_density_bwd_train(const tn_density_field *f, const float *enc, const float *raw, const float *selector,
                         const float *d_density, int64_t n, float trunc_exp_min, float *d_enc, float *d_w0, float *d_b0, float *d_w1,
                         float *d_b1, void *stream) {
    if (n == 0) return TN_OK;
    TN_TRY(density_train_supported(f));
    if (!enc || !raw || !selector || !d_density || !d_enc || !d_w0 || !d_b0 || !d_w1 || !d_b1) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(density_bwd_train_kernel, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, (hipStream_t)stream,
                       cs(f->l0.weight), cs(f->l0.bias), cs(f->l1.weight), f->average_init_density, trunc_exp_min,
                       enc, raw, selector, d_density, (long long)n, d_enc, d_w0, d_b0, d_w1, d_b1);
    TN_LAUNCH_CHECK();
    return TN_OK;
}

}  // extern "C"


