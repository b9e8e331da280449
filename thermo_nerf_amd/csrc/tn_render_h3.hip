// main_split_rays_kernel<P> — the eval field kernel (lane = ray, see tn_render_mfma.hip) with every fp32 product of the five MLP
// layers evaluated as a few low-precision MFMA products of operand PIECES, accumulated in fp32.  Two policies:
//
//   BF16x6 (mlp_precision = "bf16x6", round 4): a = p1 + p2 + p3 EXACTLY (three bf16 pieces = 24 significand bits, fp32's exponent
//     range: p1 = bf16(a), p2 = bf16(a - p1), p3 = bf16(a - p1 - p2); every residual is exact in fp32), and likewise w; of the
//     nine piece products the six of order <= 2 are kept: p1q1 + p1q2 + p2q1 + p2q2 + p1q3 + p3q1.  Each piece product is exact
//     in fp32 (8 x 8 bits); the dropped p2q3, p3q2, p3q3 are <= (2 + 2^-8) 2^-24 |a w|: a per-product error of 2^-23 relative, the
//     size of fp32's own rounding of that product — an fp32 dot product in another summation order, not a reduced-precision net.
//   F16x3 (mlp_precision = "f16x3", round 1): THREE f16 MFMA products of two f16 pieces per operand (22 of the 24 bits):
//
//     a = a_h + a_l,  w = w_h + w_l   (a_h = f16(a), a_l = f16(a - a_h); weights split once in tn_field_prepare_f16x3)
//     a.w ~= a_h.w_h + a_l.w_h + a_h.w_l            (the dropped a_l.w_l term is <= 2^-22 |a.w|)
//
// Each f16 x f16 product is exact in fp32 and v_mfma_f32_32x32x16_f16 accumulates in fp32, so the result carries a
// ~2^-22 relative product error (fp32 MFMA: 2^-24) — "f32 via 3 x f16 split", NOT an f16 network.  It needs
// |activation| < 65504 (f16 range).  Matrix-pipe time per 64 samples drops from 480 x 64 = 30.7 k cycles
// (v_mfma_f32_32x32x2_f32) to 180 x 32 = 5.8 k cycles; the price is ~3 VALU ops per activation for the split.
//
// Same transposed chain as the fp32 kernel: weights are the A operand, activations the B operand (lane <-> ray), a
// layer's accumulator registers feed the next layer's B operand directly (8 consecutive registers = one K=16 step:
// lane (j,h) register 8q+e holds feature 32*mi + 16q + (e&3) + 8(e>>2) + 4h, the order baked into the A fragments).
#include "tn_field_eval.h"

using namespace tn;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int GF = 15, APP = 32, L16 = 16, IN0 = 16 + GF + APP;
constexpr int LG = 4;
#ifndef TN_SPLIT_MLP_PRIO
#define TN_SPLIT_MLP_PRIO 1
#endif

// ---- the two splits ------------------------------------------------------------------------------------------------------------
// piece 0 is the leading one.  A block's LDS holds the A fragments of all 30 (layer, tile, k-step) combos: 16 B per lane, combo
// and piece.  F16x3: 64 KB, two 256-thread blocks per CU; BF16x6: 96 KB, ONE 512-thread block per CU (both: two waves per SIMD).
struct F16x3 {
    static constexpr int NP = 2, kBlock = 256, kBlocksPerCU = 2;
    static constexpr bool kShInLds = false;  // SH(dir) operand of a tile's rays: 16 VGPRs held over the sample loop
    typedef _Float16 elem;
    typedef v8h vec;
    static __device__ __forceinline__ void split(float v, elem (&p)[NP]) {
        p[0] = (elem)v;
        p[1] = (elem)(v - (float)p[0]);
    }
    // two values -> per piece one dword holding the pair (x0 in the low half): the unit the MFMA operands are assembled from
    static __device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&pk)[NP]) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
        pk[0] = __builtin_bit_cast(unsigned, h2{h0, h1});
        pk[1] = __builtin_bit_cast(unsigned, h2{(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1)});
    }
    // the products of one fp32 product as a list (a piece, b piece), in issue order (mma() below = the whole list)
    static constexpr int NPROD = 3;
    static __host__ __device__ constexpr int ka(int j) { return j == 0 ? 1 : 0; }
    static __host__ __device__ constexpr int kb(int j) { return j == 1 ? 1 : 0; }
    static __device__ __forceinline__ f32x16 mfma(const vec &a, const vec &b, const f32x16 &acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma(f32x16 &acc, const vec (&a)[NP], const vec (&b)[NP]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
};
struct BF16x6 {
    static constexpr int NP = 3, kBlock = 512, kBlocksPerCU = 1;
    // SH(dir) operand of a tile's rays (24 VGPRs, constant over the sample loop) parked in LDS: 6 KB per wave behind the blob
    static constexpr bool kShInLds = true;
    typedef __bf16 elem;
    typedef v8bf vec;
    static __device__ __forceinline__ void split(float v, elem (&p)[NP]) {
        p[0] = (elem)v;
        const float r1 = v - (float)p[0];  // exact: |r1| <= half a bf16 ulp of v, 16 bits
        p[1] = (elem)r1;
        p[2] = (elem)(r1 - (float)p[1]);   // exact, and exactly representable: 8 bits are left
    }
    // as pairs: one v_cvt_pk_bf16_f32 rounds two values (RNE — truncation would leave |p2| < 2^-7 |a| and the dropped products at
    // 2^-20), the pair's floats come back as a shift and a mask, the residuals are exact subtractions: 11 instructions per pair
    static __device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&pk)[NP]) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const f2 v = {x0, x1};
        const b2 p1 = __builtin_convertvector(v, b2);
        pk[0] = __builtin_bit_cast(unsigned, p1);
        // (scalar subtractions: as one v_pk_add_f32 per pair the kernel took 23.2 instead of 21.5 ms in round 4 — aligned register
        // pairs, more spills — and 21.9 against 21.0 with round 5's pipelined layers; that variant is in git history, HISTORY.md)
        const f2 r1 = {x0 - __uint_as_float(pk[0] << 16), x1 - __uint_as_float(pk[0] & 0xffff0000u)};
        const b2 p2 = __builtin_convertvector(r1, b2);
        pk[1] = __builtin_bit_cast(unsigned, p2);
        const f2 r2 = {r1[0] - __uint_as_float(pk[1] << 16), r1[1] - __uint_as_float(pk[1] & 0xffff0000u)};
        pk[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, b2));
    }
    static constexpr int NPROD = 6;  // (0,2) (2,0) (1,1) (0,1) (1,0) (0,0): small terms first
    static __host__ __device__ constexpr int ka(int j) { return j == 1 ? 2 : (j == 2 || j == 4) ? 1 : 0; }
    static __host__ __device__ constexpr int kb(int j) { return j == 0 ? 2 : (j == 2 || j == 3) ? 1 : 0; }
    static __device__ __forceinline__ f32x16 mfma(const vec &a, const vec &b, const f32x16 &acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma(f32x16 &acc, const vec (&a)[NP], const vec (&b)[NP]) {
        // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
};

// ---- blob / LDS layout ------------------------------------------------------------------------------------
// A fragments: [30 combos][64 lanes][NP pieces][8 elements] = 16 NP bytes per lane and combo; then the fp32 biases / output rows
constexpr int C_BASE1 = 0, C_BASE2 = 4, C_C1 = 8, C_T1 = 12, C_C2 = 14, C_T2 = 22;
constexpr int N_COMBOS = 30;
template <int NP>
struct Lay {
    static constexpr int A_FLOATS = N_COMBOS * 64 * 4 * NP;
    static constexpr int B_BASE1 = A_FLOATS;      // [64]
    static constexpr int B_BASE2 = B_BASE1 + 64;  // [32]
    static constexpr int B_C1 = B_BASE2 + 32;     // [64] eval bias incl. the folded appearance term
    static constexpr int B_T1 = B_C1 + 64;
    static constexpr int B_C2 = B_T1 + 64;
    static constexpr int B_T2 = B_C2 + 64;
    static constexpr int W3 = B_T2 + 64;          // [3][64] + [4]
    static constexpr int WTH = W3 + 196;          // [64] + [4]
    static constexpr int BLOB_FLOATS = WTH + 68;  // (30 combos) NP = 2: 15 976 floats = 63 904 B; NP = 3: 23 656 floats = 94 624 B
    static_assert(BLOB_FLOATS % 4 == 0, "blob must be float4-copyable");
};

__host__ __device__ inline int krow(int q, int e, int h) { return 16 * q + (e & 3) + 8 * (e >> 2) + 4 * h; }

struct RawField {
    const float *b0w, *b0b, *b1w, *b1b, *h0w, *h0b, *h1w, *h1b, *h2w, *h2b, *t0w, *t0b, *t1w, *t1b, *thw, *thb;
    const float *appearance;
    int num_images, use_avg;
};

__device__ __forceinline__ float frag_weight(const RawField &w, int combo, int i, int h, int e) {
    if (combo < C_BASE2) {  // mlp_base layer 0 [64,32]: natural feature order 16ks + 8h + e
        const int mt = combo >> 1, ks = combo & 1;
        return w.b0w[(i + 32 * mt) * 32 + 16 * ks + 8 * h + e];
    }
    if (combo < C_C1) {  // mlp_base layer 1 [16,64]
        const int ks = combo - C_BASE2;
        return i < 1 + GF ? w.b1w[i * 64 + 32 * (ks >> 1) + krow(ks & 1, e, h)] : 0.0f;
    }
    if (combo < C_T1) {  // mlp_head layer 0: ks 0 = geo (accumulator rows, row 0 = raw density -> 0), ks 1 = SH comps
        const int c = combo - C_C1, mt = c >> 1, ks = c & 1;
        if (ks == 0) {
            const int row = krow(0, e, h);
            return row >= 1 ? w.h0w[(i + 32 * mt) * IN0 + 16 + row - 1] : 0.0f;
        }
        return w.h0w[(i + 32 * mt) * IN0 + 8 * h + e];
    }
    if (combo < C_C2) {  // mlp_thermal layer 0 [64,15]
        const int mt = combo - C_T1, row = krow(0, e, h);
        return row >= 1 ? w.t0w[(i + 32 * mt) * GF + row - 1] : 0.0f;
    }
    const bool thermal = combo >= C_T2;
    const int c = combo - (thermal ? C_T2 : C_C2), mt = c >> 2, ks = c & 3;
    const float *m = thermal ? w.t1w : w.h1w;
    return m[(i + 32 * mt) * 64 + 32 * (ks >> 1) + krow(ks & 1, e, h)];
}

template <class P>
__global__ void field_prepare_split_kernel(RawField w, float *__restrict__ blob) {
    typedef Lay<P::NP> LY;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < N_COMBOS * 64 * 8) {  // one thread per (combo, lane, e): writes that element of every piece
        const int e = idx & 7, lane = (idx >> 3) & 63, combo = idx >> 9;
        typename P::elem pc[P::NP];
        P::split(frag_weight(w, combo, lane & 31, lane >> 5, e), pc);
        typename P::elem *dst = reinterpret_cast<typename P::elem *>(blob) + ((size_t)(combo * 64 + lane)) * 8 * P::NP;
#pragma unroll
        for (int k = 0; k < P::NP; ++k) dst[8 * k + e] = pc[k];
        return;
    }
    const int j = idx - N_COMBOS * 64 * 8 + LY::A_FLOATS;
    if (j >= LY::BLOB_FLOATS) return;
    float v = 0.0f;
    if (j < LY::B_BASE2) v = w.b0b[j - LY::B_BASE1];
    else if (j < LY::B_C1) { const int f = j - LY::B_BASE2; v = f < 1 + GF ? w.b1b[f] : 0.0f; }
    else if (j < LY::B_T1) {
        const int f = j - LY::B_C1;
        v = w.h0b[f];
        if (w.use_avg) {  // REF thermal_field.py:128-132
            for (int k = 0; k < APP; ++k) {
                float m = 0.0f;
                for (int im = 0; im < w.num_images; ++im) m += w.appearance[im * APP + k];
                v = fmaf(w.h0w[f * IN0 + 16 + GF + k], m / (float)w.num_images, v);
            }
        }
    }
    else if (j < LY::B_C2) v = w.t0b[j - LY::B_T1];
    else if (j < LY::B_T2) v = w.h1b[j - LY::B_C2];
    else if (j < LY::W3) v = w.t1b[j - LY::B_T2];
    else if (j < LY::WTH) { const int e = j - LY::W3; v = e < 192 ? w.h2w[e] : (e < 195 ? w.h2b[e - 192] : 0.0f); }
    else { const int e = j - LY::WTH; v = e < 64 ? w.thw[e] : (e == 64 ? w.thb[0] : 0.0f); }
    blob[j] = v;
}

// ---- device helpers ------------------------------------------------------------------------------------------
template <class P>
struct Pieces {  // one MFMA operand (A: 8 k-values of a weight row, B: of a ray's activations) as NP vectors of 8 elements
    typename P::vec p[P::NP];
};

template <class P>
__device__ __forceinline__ void mma(f32x16 &acc, const Pieces<P> &a, const Pieces<P> &b) {
    P::mma(acc, a.p, b.p);
}

template <class P>
__device__ __forceinline__ Pieces<P> load_a(const float *lds, int combo, int lane) {
    const uint4 *q = reinterpret_cast<const uint4 *>(lds) + (size_t)(combo * 64 + lane) * P::NP;
    Pieces<P> a;
#pragma unroll
    for (int k = 0; k < P::NP; ++k) a.p[k] = __builtin_bit_cast(typename P::vec, q[k]);
    return a;
}

// one value pair (2 m, 2 m + 1 of the k-step's 8 values) of split8, into dword m of every piece
template <class P, bool RELU>
__device__ __forceinline__ void split_pair_of(const f32x16 &x, int q, int m, unsigned (&pk)[P::NP][4]) {
    float v0 = x[8 * q + 2 * m], v1 = x[8 * q + 2 * m + 1];
    if (RELU) {
        v0 = relu_bits(v0);
        v1 = relu_bits(v1);
    }
    unsigned pr[P::NP];
    P::split_pair(v0, v1, pr);
#pragma unroll
    for (int k = 0; k < P::NP; ++k) pk[k][m] = pr[k];
}
template <class P>
__device__ __forceinline__ Pieces<P> pieces_of(const unsigned (&pk)[P::NP][4]) {
    Pieces<P> r;
#pragma unroll
    for (int k = 0; k < P::NP; ++k) r.p[k] = __builtin_bit_cast(typename P::vec, uint4{pk[k][0], pk[k][1], pk[k][2], pk[k][3]});
    return r;
}

template <class P, bool RELU>
__device__ __forceinline__ Pieces<P> split8(const f32x16 &x, int q) {
    unsigned pk[P::NP][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float v0 = x[8 * q + 2 * m], v1 = x[8 * q + 2 * m + 1];
        if (RELU) {
            v0 = relu_bits(v0);
            v1 = relu_bits(v1);
        }
        unsigned pr[P::NP];
        P::split_pair(v0, v1, pr);
#pragma unroll
        for (int k = 0; k < P::NP; ++k) pk[k][m] = pr[k];
    }
    Pieces<P> r;
#pragma unroll
    for (int k = 0; k < P::NP; ++k) r.p[k] = __builtin_bit_cast(typename P::vec, uint4{pk[k][0], pk[k][1], pk[k][2], pk[k][3]});
    return r;
}

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

__device__ __forceinline__ void swap32u(unsigned a, unsigned b, unsigned &lo, unsigned &hi) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    lo = r[0];
    hi = r[1];
}
__device__ __forceinline__ void swap32(float a, float b, float &lo, float &hi) {
    unsigned l, hh;
    swap32u(__float_as_uint(a), __float_as_uint(b), l, hh);
    lo = __uint_as_float(l);
    hi = __uint_as_float(hh);
}

// 16 per-lane values (this lane's ray) -> B operands of one K=16 step for both N tiles.
// in: natural order v[0..15]; lane (j,h) of tile t ends up holding v[8h + e] of ray (32t + j).
template <class P>
__device__ __forceinline__ void pack_step(const float (&v)[16], Pieces<P> &t0, Pieces<P> &t1) {
    unsigned pk[P::NP][8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        unsigned pr[P::NP];
        P::split_pair(v[2 * m], v[2 * m + 1], pr);
#pragma unroll
        for (int k = 0; k < P::NP; ++k) pk[k][m] = pr[k];
    }
#pragma unroll
    for (int k = 0; k < P::NP; ++k) {
        unsigned a0[4], a1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)  // lower half keeps packs 0..3 (features 0..7), upper half gets packs 4..7 (8..15)
            swap32u(pk[k][q], pk[k][4 + q], a0[q], a1[q]);
        t0.p[k] = __builtin_bit_cast(typename P::vec, uint4{a0[0], a0[1], a0[2], a0[3]});
        t1.p[k] = __builtin_bit_cast(typename P::vec, uint4{a1[0], a1[1], a1[2], a1[3]});
    }
}

// Software pipelining of the MLP block (round 5).  Left alone, hipcc emits a layer as bursts of 10-14 back-to-back MFMAs followed by
// runs of 50-95 split instructions: during a burst the wave is stuck behind its own MFMAs (in-order issue, 32 cycles each), during a
// run it feeds the matrix pipe nothing, and the pipe is busy 48 % of the time although the SIMD's two waves offer it 11.5 k cycles
// of work per 24 k-cycle pass.  Here the split of k-step ks + 1 is placed UNDER the MFMAs of k-step ks: one scheduling region per
// k-step (sched_barrier on both sides) with an explicit issue pattern — one MFMA, then a few vector instructions — so a wave's own
// vector work rides in its own MFMA shadow and only the first split of a layer is exposed.
// A [64 x 64] x relu(in) product, NMT output tiles of 32 features (layer64: 2, mlp_base's second layer: 1), as a software pipeline
// over (k-step, ray tile): the 2 NMT NPROD ... MFMAs of one (ks, nt) in four chunks, each followed by the split of ONE value pair of the
// NEXT (ks, nt)'s B operand, a scheduling fence after every chunk — the order below is the order in the binary.  Only the very
// first B operand (4 pairs) is split in the open; B operands live for one (ks, nt) only (12 + 12 registers instead of 48).
// A fragments: mt = 0's of the next k-step are requested once this k-step's are dead (tile 1, chunk NMT), mt = 1's at tile 0, chunk 0.
template <class P, int NMT>
__device__ __forceinline__ void layer_pipelined(const float *lds, int combo0, int lane, const f32x16 (&in)[2][2], f32x16 (&out)[NMT][2]) {
    constexpr int NM = NMT * P::NPROD;  // MFMAs per (ks, nt)
    Pieces<P> bc = split8<P, true>(in[0][0], 0);
    Pieces<P> a[2];
    a[0] = load_a<P>(lds, combo0, lane);
    a[1] = a[0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int step = 0; step < 8; ++step) {
        const int ks = step >> 1, nt = step & 1;
        const int nks = (step + 1) >> 1, nnt = (step + 1) & 1;
        const bool has_next = step < 7;
        unsigned nx[P::NP][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (NMT == 2 && nt == 0 && c == 0) a[1] = load_a<P>(lds, combo0 + 4 + ks, lane);
#pragma unroll
            for (int idx = c * NM / 4; idx < (c + 1) * NM / 4; ++idx) {
                const int mt = idx / P::NPROD, j = idx % P::NPROD;
                out[mt][nt] = P::mfma(a[mt].p[P::ka(j)], bc.p[P::kb(j)], out[mt][nt]);
            }
            // (the last MFMA that reads a[0] of this k-step sits in chunk 4 / NMT - 1 of tile 1)
            if (nt == 1 && ks < 3 && c == 4 / NMT - 1) a[0] = load_a<P>(lds, combo0 + ks + 1, lane);
            if (has_next) split_pair_of<P, true>(in[nks >> 1][nnt], nks & 1, c, nx);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) bc = pieces_of<P>(nx);
    }
}

// one 64 -> 64 layer from relu(in): out[mt][nt] = bias + sum_ks A[mt][ks] x B[nt][ks]
template <class P>
__device__ __forceinline__ void layer64(const float *lds, int combo0, const float *bias, int lane, int h,
                                        const f32x16 (&in)[2][2], f32x16 (&out)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        out[mt][0] = bias_frag(bias, mt, h);
        out[mt][1] = out[mt][0];
    }
    layer_pipelined<P, 2>(lds, combo0, lane, in, out);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int ACT>
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

__device__ __forceinline__ float combine_halves(float2 p) {
    float lo, hi;
    swap32(p.x, p.y, lo, hi);
    return lo + hi;
}


struct H3Args {
    Grid g;
    tn_space space;
    const float *blob;
    float avg;
    int sh_shifted;
    const float *origins, *dirs, *nears, *fars;
    const float *spacing;
    long long R;
    int S, lin;
    float *rgb, *acc, *depth, *expected, *thermal;
    DepthSlots minmax;  // expected-depth clip bounds: one key pair per call, or per reference chunk of the frame
    float early_eps;  // 0 = never stop early
};

template <class P>
__global__ void __launch_bounds__(P::kBlock, P::kBlocksPerCU) main_split_rays_kernel(H3Args a) {
    typedef Lay<P::NP> LY;
    typedef Pieces<P> HL;
    constexpr int kBlock = P::kBlock, kWaves = P::kBlock / TN_WAVE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.blob);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < LY::BLOB_FLOATS / 4; i += kBlock) dst[i] = src[i];
    }
    __syncthreads();
    const Space sp = make_space(a.space);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const bool lin = a.lin != 0;
    const int S = a.S;
    const long long groups = (a.R + 63) >> 6;
    const long long stride = (long long)gridDim.x * kWaves;
    float smin = INFINITY, smax = -INFINITY;
    long long mm_slot = 0;
    for (long long grp = (long long)blockIdx.x * kWaves + wave; grp < groups; grp += stride) {
        // (a tile never straddles two chunks: first_ray and chunk_rays are multiples of 64; a wave's tiles ascend)
        if (a.minmax.chunk_rays > 0 && a.minmax.slot(grp * 64) != mm_slot) {
            depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
            mm_slot = a.minmax.slot(grp * 64);
        }
        const long long r = grp * 64 + lane;
        const bool live = r < a.R;
        const long long rc = live ? r : a.R - 1;
        const float ox = a.origins[rc * 3], oy = a.origins[rc * 3 + 1], oz = a.origins[rc * 3 + 2];
        const float dx = a.dirs[rc * 3], dy = a.dirs[rc * 3 + 1], dz = a.dirs[rc * 3 + 2];
        const float s_near = spacing_fn(a.nears[rc], lin), s_far = spacing_fn(a.fars[rc], lin);
        const float *tb = a.spacing + tn_ws_bin(grp * 64, 0, S) + (rc - grp * 64);
        HL sh0, sh1;  // SH(dir) of this lane's ray as the K=16 step of the colour layer (constant over samples)
        uint4 *sh_lds = reinterpret_cast<uint4 *>(lds + LY::BLOB_FLOATS) + (size_t)wave * 64 * 2 * P::NP + lane;  // [2 tiles][NP][64 lanes]
        {
            float sx = dx, sy = dy, sz = dz;
            if (a.sh_shifted) {
                sx = add_rn(sx, 1.0f) / 2.0f; sy = add_rn(sy, 1.0f) / 2.0f; sz = add_rn(sz, 1.0f) / 2.0f;
            }
            float c[16];
            sh16(sx, sy, sz, c);
            pack_step<P>(c, sh0, sh1);
            if (P::kShInLds) {  // wave-private, written and read by the same lane: no barrier
#pragma unroll
                for (int k = 0; k < P::NP; ++k) {
                    sh_lds[(size_t)k * 64] = __builtin_bit_cast(uint4, sh0.p[k]);
                    sh_lds[(size_t)(P::NP + k) * 64] = __builtin_bit_cast(uint4, sh1.p[k]);
                }
            }
        }
        float en = spacing_to_eucl<true>(tb[0], s_near, s_far, lin);
        float accum = 0.0f, cum_w = 0.0f;
        float wsum = 0.0f, wr = 0.0f, wg = 0.0f, wbl = 0.0f, wth = 0.0f, wsteps = 0.0f;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, th = 0.0f, med = 0.0f, step = 0.0f;
        bool med_found = false;
        float sb_next = tb[64];
        for (int i = 0; i < S; ++i) {
            const float st = en;
            en = spacing_to_eucl<true>(sb_next, s_near, s_far, lin);
            sb_next = tb[(size_t)(i + 2 <= S ? i + 2 : S) * 64];  // the next bin edge, requested a sample ahead (tn_render_mfma.hip)
            step = add_rn(st, en) / 2.0f;
            float px, py, pz;
            const float sel = normalize_position<true>(sp, frustum_pos(ox, dx, st, en), frustum_pos(oy, dy, st, en),
                                                 frustum_pos(oz, dz, st, en), px, py, pz);
            // ---- hash grid -> two K=16 steps of B operands per N tile --------------------------------------
            HL e0[2], e1[2];  // [ks] for tile 0 / tile 1
            if (a.g.num_dense >= kFieldDense) {
                // the first kFieldDense (<= 8) levels from the dense re-layout: 4 aligned 16-byte gathers per level
                static_assert(kFieldDense <= 8, "the dense levels sit in the first half");
                float v[16];
                hash_encode_pipelined<8, 2, kFieldDense>(a.g, px, py, pz, [&](int l, float2 f) {
                    v[2 * l] = f.x;
                    v[2 * l + 1] = f.y;
                }, 0);
                pack_step<P>(v, e0[0], e1[0]);
                hash_encode_pipelined<8, 2>(a.g, px, py, pz, [&](int l, float2 f) {
                    v[2 * l] = f.x;
                    v[2 * l + 1] = f.y;
                }, 8);
                pack_step<P>(v, e0[1], e1[1]);
            } else if (a.g.num_dense == 0) {
                // hashed levels: index arithmetic | gathers | interpolation in explicit stages; two groups of 2 levels (2 x 16
                // gathers) in flight — this kernel has ~30 fewer free VGPRs than the fp32 one
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[16];
                    hash_encode_pipelined<8, 2>(a.g, px, py, pz, [&](int l, float2 f) {
                        v[2 * l] = f.x;
                        v[2 * l + 1] = f.y;
                    }, 8 * ks);
                    pack_step<P>(v, e0[ks], e1[ks]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float v[16];
#pragma unroll
                    for (int l0 = 0; l0 < 8; l0 += LG) {
                        float2 f[LG];
#pragma unroll
                        for (int q = 0; q < LG; ++q) f[q] = encode_level_any<true>(a.g, 8 * ks + l0 + q, px, py, pz);
#pragma unroll
                        for (int q = 0; q < LG; ++q) {
                            v[2 * (l0 + q)] = f[q].x;
                            v[2 * (l0 + q) + 1] = f[q].y;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    pack_step<P>(v, e0[ks], e1[ks]);
                }
            }
            // ---- mlp_base layer 0: 32 -> 64 ---------------------------------------------------------------
#if TN_SPLIT_MLP_PRIO
            // the wave in its matrix-rich block goes ahead of its partner's hash / compositing block in the SIMD's issue arbitration,
            // so the matrix pipe is fed as soon as operands exist: bf16x6 23.3 -> 22.5 ms, f16x3 16.5 -> 15.7 (priority 1 = 3; round 4)
            __builtin_amdgcn_s_setprio(TN_SPLIT_MLP_PRIO);
#endif
            f32x16 h1[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                h1[mt][0] = bias_frag(lds + LY::B_BASE1, mt, h);
                h1[mt][1] = h1[mt][0];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const HL aw = load_a<P>(lds, C_BASE1 + mt * 2 + ks, lane);
                    mma<P>(h1[mt][0], aw, e0[ks]);
                    mma<P>(h1[mt][1], aw, e1[ks]);
                }
            }
            // ---- mlp_base layer 1: 64 -> 16 ---------------------------------------------------------------
            f32x16 g[1][2];
            g[0][0] = bias_frag(lds + LY::B_BASE2, 0, h);
            g[0][1] = g[0][0];
            layer_pipelined<P, 1>(lds, C_BASE2, lane, h1, g);
            float raw, unused;
            swap32(g[0][0][0], g[0][1][0], raw, unused);
            const float dens = mul_rn(mul_rn(a.avg, __expf(raw)), sel);
            const HL g0 = split8<P, false>(g[0][0], 0), g1 = split8<P, false>(g[0][1], 0);  // geo rows (row 0 has zero weight)
            {   // colour: [geo | SH] -> 64 -> 64 -> 3
                f32x16 x1[2][2], x2[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    x1[mt][0] = bias_frag(lds + LY::B_C1, mt, h);
                    x1[mt][1] = x1[mt][0];
                    const HL ag = load_a<P>(lds, C_C1 + mt * 2, lane), as = load_a<P>(lds, C_C1 + mt * 2 + 1, lane);
                    mma<P>(x1[mt][0], ag, g0);
                    mma<P>(x1[mt][1], ag, g1);
                    if (P::kShInLds) {
                        HL s0, s1;
#pragma unroll
                        for (int k = 0; k < P::NP; ++k) {
                            s0.p[k] = __builtin_bit_cast(typename P::vec, sh_lds[(size_t)k * 64]);
                            s1.p[k] = __builtin_bit_cast(typename P::vec, sh_lds[(size_t)(P::NP + k) * 64]);
                        }
                        mma<P>(x1[mt][0], as, s0);
                        mma<P>(x1[mt][1], as, s1);
                    } else {
                        mma<P>(x1[mt][0], as, sh0);
                        mma<P>(x1[mt][1], as, sh1);
                    }
                }
                layer64<P>(lds, C_C2, lds + LY::B_C2, lane, h, x1, x2);
                const float *w3 = lds + LY::W3;
                cr = fast_sigmoid(combine_halves(out_dot_fast<0>(w3, h, x2)) + w3[192]);
                cg = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 64, h, x2)) + w3[193]);
                cb = fast_sigmoid(combine_halves(out_dot_fast<0>(w3 + 128, h, x2)) + w3[194]);
            }
            {   // thermal: geo -> 64 -> 64 sigmoid -> 1
                f32x16 x1[2][2], x2[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    x1[mt][0] = bias_frag(lds + LY::B_T1, mt, h);
                    x1[mt][1] = x1[mt][0];
                    const HL ag = load_a<P>(lds, C_T1 + mt, lane);
                    mma<P>(x1[mt][0], ag, g0);
                    mma<P>(x1[mt][1], ag, g1);
                }
                layer64<P>(lds, C_T2, lds + LY::B_T2, lane, h, x1, x2);
                const float *wt = lds + LY::WTH;
                th = combine_halves(out_dot_fast<1>(wt, h, x2)) + wt[64];
            }
#if TN_SPLIT_MLP_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            cr = nan_to_num(cr); cg = nan_to_num(cg); cb = nan_to_num(cb); th = nan_to_num(th);
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
            // early ray termination (eval, opt-in): wave-wide vote on the transmittance left after this sample
            if (a.early_eps > 0.0f && i + 1 < S && __all(__expf(-accum) < a.early_eps)) {
                // keep the call-global depth bounds exact: they only miss the last mid-point
                const float e0 = spacing_to_eucl<true>(tb[(size_t)(S - 1) * 64], s_near, s_far, lin);
                const float e1 = spacing_to_eucl<true>(tb[(size_t)S * 64], s_near, s_far, lin);
                smax = fmaxf(smax, add_rn(e0, e1) / 2.0f);
                break;
            }
        }
        if (live) {
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
    }
    depth_bounds_flush(a.minmax, mm_slot, smin, smax, lane);
}


// ---- the split itself, exposed for its unit test (tn_bf16x6_split_product) ---------------------------------------------------------
// per element: the three bf16 pieces of a and of b (as floats) and the six-product sum p1q3 + p3q1 + p2q2 + p1q2 + p2q1 + p1q1
// accumulated in fp32 in the kernel's order (every product is exact in fp32: 8 x 8 significand bits)
__global__ void bf16x6_split_product_kernel(const float *a, const float *b, long long n, float *pieces_a, float *pieces_b, float *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __bf16 pa[3], pb[3];
    BF16x6::split(a[i], pa);
    BF16x6::split(b[i], pb);
    float fa[3], fb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        fa[k] = (float)pa[k];
        fb[k] = (float)pb[k];
        pieces_a[3 * i + k] = fa[k];
        pieces_b[3 * i + k] = fb[k];
    }
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < BF16x6::NPROD; ++j) acc = add_rn(acc, mul_rn(fa[BF16x6::ka(j)], fb[BF16x6::kb(j)]));
    out[i] = acc;
}
// one v_mfma_f32_32x32x16_bf16 whose A operand is the constant `av` and whose B operand is the constant `bv` (both rounded to
// bf16): out[0] = 16 av bv if the matrix core takes sub-normal bf16 inputs as they are, 0 if it flushes them
__global__ void bf16_mfma_value_probe_kernel(float av, float bv, float *out) {
    const __bf16 x = (__bf16)av, y = (__bf16)bv;
    const v8bf a = {x, x, x, x, x, x, x, x}, b = {y, y, y, y, y, y, y, y};
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

inline bool h3_supported(const tn_thermal_field *f) {
    return f && f->geo_feat_dim == GF && f->app_dim == APP && f->grid.num_levels == L16;
}

}  // namespace

namespace tn {

template <class P>
static int launch_main_split(const tn_thermal_field *field, const float *blob, const tn_render_config *cfg, const tn_render_inputs *in,
                             const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                             hipStream_t stream) {
    if (!h3_supported(field) || !blob || cfg->training) return TN_ERR_UNSUPPORTED;
    H3Args a;
    a.g = tn_make_grid(field->grid);
    a.space = field->space;
    a.blob = blob;
    a.avg = field->average_init_density;
    a.sh_shifted = field->sh_shifted;
    a.origins = in->origins; a.dirs = in->directions; a.nears = in->nears; a.fars = in->fars;
    a.spacing = spacing_ws;
    a.R = num_rays; a.S = cfg->num_nerf_samples; a.lin = cfg->initial_sampler == 1;
    a.rgb = out->rgb; a.acc = out->accumulation; a.depth = out->depth; a.expected = out->expected_depth;
    a.thermal = out->thermal; a.minmax = minmax;
    a.early_eps = fminf(fmaxf(cfg->early_stop_transmittance, 0.0f), 0.25f);
    constexpr int kWaves = P::kBlock / TN_WAVE;
    const size_t smem = (size_t)Lay<P::NP>::BLOB_FLOATS * sizeof(float) + (P::kShInLds ? (size_t)kWaves * 64 * 2 * P::NP * 16 : 0);
    if (!tn_ensure_dynamic_lds<main_split_rays_kernel<P>>(smem)) return TN_ERR_LAUNCH;
    const long long groups = (num_rays + 63) / 64;
    const long long need = (groups + kWaves - 1) / kWaves;
    const long long cap = 256LL * P::kBlocksPerCU;
    const unsigned grid = (unsigned)(need < cap ? (need < 1 ? 1 : need) : cap);
    hipLaunchKernelGGL(main_split_rays_kernel<P>, dim3(grid), dim3(P::kBlock), smem, stream, a);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

int launch_main_h3(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                   const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                   hipStream_t stream) {
    return launch_main_split<F16x3>(field, field ? field->prepared_f16x3 : nullptr, cfg, in, out, num_rays, spacing_ws, minmax, stream);
}

int launch_main_b6(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                   const tn_render_outputs *out, long long num_rays, const float *spacing_ws, DepthSlots minmax,
                   hipStream_t stream) {
    return launch_main_split<BF16x6>(field, field ? field->prepared_bf16x6 : nullptr, cfg, in, out, num_rays, spacing_ws, minmax, stream);
}

template <class P>
static int field_prepare_split(const tn_thermal_field *f, void *prepared_dev, size_t bytes, void *stream) {
    typedef Lay<P::NP> LY;
    if (!f || !prepared_dev) return TN_ERR_NULL;
    TN_TRY(tn_check_thermal_field(f));
    if (!h3_supported(f)) return TN_ERR_UNSUPPORTED;
    if (bytes < (size_t)LY::BLOB_FLOATS * sizeof(float)) return TN_ERR_WORKSPACE;
    RawField w;
    w.b0w = f->base0.weight; w.b0b = f->base0.bias; w.b1w = f->base1.weight; w.b1b = f->base1.bias;
    w.h0w = f->head0.weight; w.h0b = f->head0.bias; w.h1w = f->head1.weight; w.h1b = f->head1.bias;
    w.h2w = f->head2.weight; w.h2b = f->head2.bias; w.t0w = f->th0.weight; w.t0b = f->th0.bias;
    w.t1w = f->th1.weight; w.t1b = f->th1.bias; w.thw = f->thead.weight; w.thb = f->thead.bias;
    w.appearance = f->appearance; w.num_images = f->num_images; w.use_avg = f->use_average_appearance;
    const int threads = N_COMBOS * 64 * 8 + (LY::BLOB_FLOATS - LY::A_FLOATS);
    hipLaunchKernelGGL(field_prepare_split_kernel<P>, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float *>(prepared_dev));
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

}  // namespace tn

extern "C" {

size_t tn_field_prepare_f16x3_bytes(const tn_thermal_field *field) {
    if (!h3_supported(field) || tn_check_thermal_field(field) != TN_OK) return 0;
    return (size_t)Lay<F16x3::NP>::BLOB_FLOATS * sizeof(float);
}

int tn_field_prepare_f16x3(const tn_thermal_field *f, void *prepared_dev, size_t bytes, void *stream) {
    return tn::field_prepare_split<F16x3>(f, prepared_dev, bytes, stream);
}

size_t tn_field_prepare_bf16x6_bytes(const tn_thermal_field *field) {
    if (!h3_supported(field) || tn_check_thermal_field(field) != TN_OK) return 0;
    return (size_t)Lay<BF16x6::NP>::BLOB_FLOATS * sizeof(float);
}

int tn_field_prepare_bf16x6(const tn_thermal_field *f, void *prepared_dev, size_t bytes, void *stream) {
    return tn::field_prepare_split<BF16x6>(f, prepared_dev, bytes, stream);
}

int tn_bf16x6_split_product(const float *a, const float *b, int64_t n, float *pieces_a, float *pieces_b, float *out, void *stream) {
    if (n == 0) return TN_OK;
    if (!a || !b || !pieces_a || !pieces_b || !out) return TN_ERR_NULL;
    if (n < 0) return TN_ERR_SHAPE;
    hipLaunchKernelGGL(bf16x6_split_product_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                       (long long)n, pieces_a, pieces_b, out);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

int tn_bf16_mfma_value_probe(float a_value, float b_value, float *out, void *stream) {
    if (!out) return TN_ERR_NULL;
    hipLaunchKernelGGL(bf16_mfma_value_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a_value, b_value, out);
    if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH;
    return TN_OK;
}

}  // extern "C"
