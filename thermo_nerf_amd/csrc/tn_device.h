// Device-side building blocks shared by every kernel of libthermonerf_hip (gfx950 / wave64 only).
//
// Arithmetic follows the nerfstudio 1.1.5 torch-fallback op order with explicit
// round-to-nearest mul/add intrinsics wherever the reference rounds twice, so position / hash-grid
// arithmetic is reproducible against the fp32 CPU oracle irrespective of -ffp-contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/thermonerf_hip.h"

#define TN_WAVE 64

namespace tn {

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

// torch.nan_to_num defaults: nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
__device__ __forceinline__ float nan_to_num(float x) {
    if (x != x) return 0.0f;
    if (x == INFINITY) return 3.4028234663852886e38f;
    if (x == -INFINITY) return -3.4028234663852886e38f;
    return x;
}

// ReLU of a value that comes out of an MFMA accumulator.  fmaxf(x, 0) lowers to v_max_f32 x, x (canonicalise) + v_max_f32 x, 0:
// hipcc cannot see that matrix-core results are already canonical.  The signed-integer max of the bit pattern is ONE v_max_i32:
// negative floats (sign bit set, -0 included) are negative integers -> +0, everything else passes through unchanged (a NaN
// with a clear sign bit stays NaN, like torch.relu).  Same value as fmaxf for every non-NaN input.
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// ---- arithmetic flavour -------------------------------------------------------------------------------
// FAST = false (default; every plugin-surface kernel): IEEE division, full-precision expf, torch's three-rounding
// interpolation — the op order of the torch reference.
// FAST = true (the fused throughput kernels): v_rcp_f32 / v_exp_f32 based reciprocal, quotient and exp (<= 2 ulp)
// and the two-instruction fma form of the interpolation.  Same formulas, ulp-level different rounding.
template <bool FAST>
__device__ __forceinline__ float t_rcp(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(x);
    return 1.0f / x;
}
template <bool FAST>
__device__ __forceinline__ float t_div(float a, float b) {
    if (FAST) return a * __builtin_amdgcn_rcpf(b);
    return a / b;
}
template <bool FAST>
__device__ __forceinline__ float t_exp(float x) {
    if (FAST) return __expf(x);
    return expf(x);
}

// ---- NS SpacedSampler spacing functions ---------------------------------------------------
// lin == false: UniformLinDispPiecewiseSampler (the "piecewise" proposal_initial_sampler, the reference default);
// lin == true:  UniformSampler (proposal_initial_sampler="uniform", REF thermal_nerf_model.py:164-170): both functions are
//               the identity, so edges map linearly between near and far.  The choice is made once per call and carried by
//               every level (PDFSampler reuses the initial sampler's spacing_to_euclidean_fn).
__device__ __forceinline__ float spacing_fn(float x, bool lin = false) {
    return lin ? x : (x < 1.0f ? x / 2.0f : sub_rn(1.0f, 1.0f / mul_rn(2.0f, x)));
}
template <bool FAST = false>
__device__ __forceinline__ float spacing_fn_inv(float x) {
    return x < 0.5f ? mul_rn(2.0f, x) : t_rcp<FAST>(sub_rn(2.0f, mul_rn(2.0f, x)));
}
// spacing_to_euclidean_fn(x) = s_inv(x * s_far + (1 - x) * s_near)
template <bool FAST = false>
__device__ __forceinline__ float spacing_to_eucl(float x, float s_near, float s_far, bool lin = false) {
    const float u = add_rn(mul_rn(x, s_far), mul_rn(sub_rn(1.0f, x), s_near));
    return lin ? u : spacing_fn_inv<FAST>(u);
}

// ---- NS Frustums.get_positions: o + d * (s + e) / 2 ---------------------------------------------------
__device__ __forceinline__ float frustum_pos(float o, float d, float s, float e) {
    return add_rn(o, mul_rn(d, add_rn(s, e)) / 2.0f);
}

// ---- position normalisation + selector ----------------------------------------------------
struct Space {
    int contraction;
    float mn[3];
    float mx[3];
};
__device__ __forceinline__ Space make_space(const tn_space &s) {
    Space r;
    r.contraction = s.contraction;
    for (int i = 0; i < 3; ++i) {
        r.mn[i] = s.aabb_min[i];
        r.mx[i] = s.aabb_max[i];
    }
    return r;
}

// returns selector (0/1) and writes p (already multiplied by the selector) in [0,1]
template <bool FAST = false>
__device__ __forceinline__ float normalize_position(const Space &sp, float x, float y, float z, float &px, float &py,
                                                    float &pz) {
    if (sp.contraction) {
        const float mag = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
        if (!(mag < 1.0f)) {
            const float rinv = t_rcp<FAST>(mag);
            const float k = sub_rn(2.0f, rinv);
            x = mul_rn(k, FAST ? x * rinv : x / mag);
            y = mul_rn(k, FAST ? y * rinv : y / mag);
            z = mul_rn(k, FAST ? z * rinv : z / mag);
        }
        px = add_rn(x, 2.0f) / 4.0f;
        py = add_rn(y, 2.0f) / 4.0f;
        pz = add_rn(z, 2.0f) / 4.0f;
    } else {
        px = sub_rn(x, sp.mn[0]) / sub_rn(sp.mx[0], sp.mn[0]);
        py = sub_rn(y, sp.mn[1]) / sub_rn(sp.mx[1], sp.mn[1]);
        pz = sub_rn(z, sp.mn[2]) / sub_rn(sp.mx[2], sp.mn[2]);
    }
    const bool in = (px > 0.0f) && (px < 1.0f) && (py > 0.0f) && (py < 1.0f) && (pz > 0.0f) && (pz < 1.0f);
    const float sel = in ? 1.0f : 0.0f;
    px = mul_rn(px, sel);
    py = mul_rn(py, sel);
    pz = mul_rn(pz, sel);
    return sel;
}

// ---- NS HashEncoding.pytorch_fwd, one level ----------------------------------------------
// by-value device view of tn_hashgrid
struct Grid {
    const float2 *table;
    const float2 *dense;
    float scal[TN_MAX_LEVELS];
    long long dense_off[TN_MAX_LEVELS];
    int dense_res[TN_MAX_LEVELS];
    int num_levels;
    int num_dense;
    unsigned mask;   // T - 1
    unsigned tsize;  // T
};

#define TN_P1 2654435761u
#define TN_P2 805459861u

template <bool FAST = false>
__device__ __forceinline__ float lerp_t(float a, float b, float o) {
    if (FAST) return fmaf(o, a - b, b);  // b + o (a - b): two instructions
    // torch order: a*o + b*(1-o), three roundings
    return add_rn(mul_rn(a, o), mul_rn(b, sub_rn(1.0f, o)));
}

template <bool DENSE, bool FAST = false>
__device__ __forceinline__ float2 encode_level(const Grid &g, int l, float px, float py, float pz) {
    const float s = g.scal[l];
    const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
    // coordinates are >= 0 (p in [0,1] times the selector), so floor is the truncating convert and the offset is
    // v_fract_f32 = s - floor(s) (exact for these values): one instruction each instead of floor + subtract + convert
    const float fxf = (float)(int)sx, fyf = (float)(int)sy, fzf = (float)(int)sz;
    const float ox = __builtin_amdgcn_fractf(sx), oy = __builtin_amdgcn_fractf(sy), oz = __builtin_amdgcn_fractf(sz);
    float2 f0, f1, f2, f3, f4, f5, f6, f7;
    if (DENSE) {
        // dense[x][y][z] = table[hash(x,y,z)]; ceil corner == floor+1 whenever its weight is non-zero
        const int res = g.dense_res[l];
        const float4 *d = reinterpret_cast<const float4 *>(g.dense) + g.dense_off[l];
        // every dense element is the pair (entry(z), entry(z+1)): the (.,.,f) and (.,.,c) corners are ONE aligned 16-byte load
        float4 v00, v01, v10, v11;
        if (FAST) {
            // 32-bit byte offsets from a wave-uniform level base (scalar-base global_load), 24-bit multiplies (side <= 1024)
            const unsigned ures = (unsigned)res;
            const unsigned i00 = __umul24(__umul24((unsigned)(int)fxf, ures) + (unsigned)(int)fyf, ures) + (unsigned)(int)fzf;
            const unsigned b00 = i00 << 4, sy = ures << 4, sx = __umul24(ures, ures) << 4;
            const char *db = reinterpret_cast<const char *>(d);
            v00 = *reinterpret_cast<const float4 *>(db + b00);
            v01 = *reinterpret_cast<const float4 *>(db + (b00 + sy));
            v10 = *reinterpret_cast<const float4 *>(db + (b00 + sx));
            v11 = *reinterpret_cast<const float4 *>(db + (b00 + sx + sy));
        } else {
            const int fx = (int)fxf, fy = (int)fyf, fz = (int)fzf;
            const int i00 = (fx * res + fy) * res + fz;
            const int i10 = i00 + res * res;  // x+1
            const int i01 = i00 + res;        // y+1
            const int i11 = i10 + res;
            v00 = d[i00]; v01 = d[i01]; v10 = d[i10]; v11 = d[i11];
        }
        f6 = make_float2(v00.x, v00.y);  // (f,f,f)
        f2 = make_float2(v00.z, v00.w);  // (f,f,c)
        f7 = make_float2(v01.x, v01.y);  // (f,c,f)
        f3 = make_float2(v01.z, v01.w);  // (f,c,c)
        f5 = make_float2(v10.x, v10.y);  // (c,f,f)
        f1 = make_float2(v10.z, v10.w);  // (c,f,c)
        f4 = make_float2(v11.x, v11.y);  // (c,c,f)
        f0 = make_float2(v11.z, v11.w);  // (c,c,c)
    } else {
        // coordinates are >= 0: ceil = floor + (offset > 0), so the ceil corner's hash product is the floor corner's plus
        // 0 or the prime (mod 2^32) — two quarter-rate integer multiplies instead of four, no v_ceil / second convert
        const unsigned fx = (unsigned)(int)fxf, fy = (unsigned)(int)fyf, fz = (unsigned)(int)fzf;
        // FAST: the reference's ceil corner is floor + 1 unless the coordinate sits exactly on a grid plane — and there its
        // interpolation weight (the offset) is exactly 0, so whichever finite entry is read the result is the same bit for
        // bit: always read floor + 1 (no compare / select per axis).  The torch-order kernels keep the reference's index.
        const unsigned cx = fx + ((FAST || ox > 0.0f) ? 1u : 0u);
        const unsigned hfy = fy * TN_P1, hfz = fz * TN_P2;
        const unsigned hcy = hfy + ((FAST || oy > 0.0f) ? TN_P1 : 0u), hcz = hfz + ((FAST || oz > 0.0f) ? TN_P2 : 0u);
        if (FAST) {
            // byte offsets straight from the hash: ((x ^ y P1 ^ z P2) & (T-1)) * 8 == (8x ^ y (8 P1) ^ z (8 P2)) & (8 (T-1))
            // (mod 2^32; log2 T <= 24 keeps the masked bits below 2^27).  A wave-uniform level base + a 32-bit lane offset is
            // the scalar-base form of global_load (no 64-bit address arithmetic per corner).
            const unsigned x0 = fx << 3, x1 = x0 + 8u;
            const unsigned y0 = fy * (TN_P1 << 3), y1 = y0 + (TN_P1 << 3);
            const unsigned z0 = fz * (TN_P2 << 3), z1 = z0 + (TN_P2 << 3);
            const unsigned m8 = g.mask << 3;
            const char *tb = reinterpret_cast<const char *>(g.table + (size_t)l * g.tsize);
            f0 = *reinterpret_cast<const float2 *>(tb + ((x1 ^ y1 ^ z1) & m8));
            f1 = *reinterpret_cast<const float2 *>(tb + ((x1 ^ y0 ^ z1) & m8));
            f2 = *reinterpret_cast<const float2 *>(tb + ((x0 ^ y0 ^ z1) & m8));
            f3 = *reinterpret_cast<const float2 *>(tb + ((x0 ^ y1 ^ z1) & m8));
            f4 = *reinterpret_cast<const float2 *>(tb + ((x1 ^ y1 ^ z0) & m8));
            f5 = *reinterpret_cast<const float2 *>(tb + ((x1 ^ y0 ^ z0) & m8));
            f6 = *reinterpret_cast<const float2 *>(tb + ((x0 ^ y0 ^ z0) & m8));
            f7 = *reinterpret_cast<const float2 *>(tb + ((x0 ^ y1 ^ z0) & m8));
        } else {
            const float2 *t = g.table + (size_t)l * g.tsize;
            const unsigned m = g.mask;
            f0 = t[(cx ^ hcy ^ hcz) & m];
            f1 = t[(cx ^ hfy ^ hcz) & m];
            f2 = t[(fx ^ hfy ^ hcz) & m];
            f3 = t[(fx ^ hcy ^ hcz) & m];
            f4 = t[(cx ^ hcy ^ hfz) & m];
            f5 = t[(cx ^ hfy ^ hfz) & m];
            f6 = t[(fx ^ hfy ^ hfz) & m];
            f7 = t[(fx ^ hcy ^ hfz) & m];
        }
    }
    float2 r;
    {
        const float f03 = lerp_t<FAST>(f0.x, f3.x, ox), f12 = lerp_t<FAST>(f1.x, f2.x, ox);
        const float f56 = lerp_t<FAST>(f5.x, f6.x, ox), f47 = lerp_t<FAST>(f4.x, f7.x, ox);
        const float f0312 = lerp_t<FAST>(f03, f12, oy), f4756 = lerp_t<FAST>(f47, f56, oy);
        r.x = lerp_t<FAST>(f0312, f4756, oz);
    }
    {
        const float f03 = lerp_t<FAST>(f0.y, f3.y, ox), f12 = lerp_t<FAST>(f1.y, f2.y, ox);
        const float f56 = lerp_t<FAST>(f5.y, f6.y, ox), f47 = lerp_t<FAST>(f4.y, f7.y, ox);
        const float f0312 = lerp_t<FAST>(f03, f12, oy), f4756 = lerp_t<FAST>(f47, f56, oy);
        r.y = lerp_t<FAST>(f0312, f4756, oz);
    }
    return r;
}

// ---- the FAST hashed level in three stages (index arithmetic | the 8 gathers | interpolation) -----------------------------
// Same arithmetic as encode_level<false, true>; split so that a kernel can put a scheduling barrier between the stages and
// keep the gathers of SEVERAL levels in flight together: left to itself hipcc drains the load queue after every level
// (8 gathers, s_waitcnt vmcnt(0), interpolate), which exposes one memory round trip per level.
// Stage fence: the scheduling barrier stops the machine scheduler; the empty asm with a memory clobber keeps instruction
// selection from placing the (side-effect-free) gathers after it — sched_barrier alone is IntrNoMem and does not hold loads.
#define TN_STAGE_FENCE()                        \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

struct HashTaps {
    unsigned off[8];  // byte offsets of the 8 corners inside the level's table, in the f0..f7 order of encode_level
    float ox, oy, oz;
};
template <typename G>
__device__ __forceinline__ void hash_taps(const G &g, int l, float px, float py, float pz, HashTaps &t) {
    const float s = g.scal[l];
    const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
    t.ox = __builtin_amdgcn_fractf(sx);
    t.oy = __builtin_amdgcn_fractf(sy);
    t.oz = __builtin_amdgcn_fractf(sz);
    const unsigned fx = (unsigned)(int)sx, fy = (unsigned)(int)sy, fz = (unsigned)(int)sz;
    const unsigned x0 = fx << 3, x1 = x0 + 8u;
    const unsigned y0 = fy * (TN_P1 << 3), y1 = y0 + (TN_P1 << 3);
    const unsigned z0 = fz * (TN_P2 << 3), z1 = z0 + (TN_P2 << 3);
    const unsigned m8 = g.mask << 3;
    t.off[0] = (x1 ^ y1 ^ z1) & m8;
    t.off[1] = (x1 ^ y0 ^ z1) & m8;
    t.off[2] = (x0 ^ y0 ^ z1) & m8;
    t.off[3] = (x0 ^ y1 ^ z1) & m8;
    t.off[4] = (x1 ^ y1 ^ z0) & m8;
    t.off[5] = (x1 ^ y0 ^ z0) & m8;
    t.off[6] = (x0 ^ y0 ^ z0) & m8;
    t.off[7] = (x0 ^ y1 ^ z0) & m8;
}
template <typename G>
__device__ __forceinline__ void hash_gather(const G &g, int l, const HashTaps &t, float2 (&f)[8]) {
    const char *tb = reinterpret_cast<const char *>(g.table + (size_t)l * g.tsize);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = *reinterpret_cast<const float2 *>(tb + t.off[k]);
}
// How many leading levels the lane = ray field kernels read from the dense re-layout when the grid carries at least that
// many (6 levels of the reference's 16-level grid = 14.5 MB; 4 / 8 / 10 levels measured within 0.3 % of it).
#ifndef TN_FIELD_DENSE_LEVELS
#define TN_FIELD_DENSE_LEVELS 6
#endif
constexpr int kFieldDense = TN_FIELD_DENSE_LEVELS;

// The same two stages for a level of the DENSE re-layout (elements = aligned (entry(z), entry(z+1)) pairs): four byte offsets,
// four 16-byte gathers, landing in the f0..f7 order of the hashed form.
template <typename G>
__device__ __forceinline__ void dense_taps(const G &g, int l, float px, float py, float pz, HashTaps &t) {
    const float s = g.scal[l];
    const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
    t.ox = __builtin_amdgcn_fractf(sx);
    t.oy = __builtin_amdgcn_fractf(sy);
    t.oz = __builtin_amdgcn_fractf(sz);
    const unsigned ures = (unsigned)g.dense_res[l];
    const unsigned i00 = __umul24(__umul24((unsigned)(int)sx, ures) + (unsigned)(int)sy, ures) + (unsigned)(int)sz;
    const unsigned b00 = i00 << 4, dy = ures << 4, dx = __umul24(ures, ures) << 4;
    t.off[0] = b00;            // (x0, y0): f6 | f2
    t.off[1] = b00 + dy;       // (x0, y1): f7 | f3
    t.off[2] = b00 + dx;       // (x1, y0): f5 | f1
    t.off[3] = b00 + dx + dy;  // (x1, y1): f4 | f0
}
template <typename G>
__device__ __forceinline__ void dense_gather(const G &g, int l, const HashTaps &t, float2 (&f)[8]) {
    const char *db = reinterpret_cast<const char *>(reinterpret_cast<const float4 *>(g.dense) + g.dense_off[l]);
    const float4 v00 = *reinterpret_cast<const float4 *>(db + t.off[0]), v01 = *reinterpret_cast<const float4 *>(db + t.off[1]);
    const float4 v10 = *reinterpret_cast<const float4 *>(db + t.off[2]), v11 = *reinterpret_cast<const float4 *>(db + t.off[3]);
    f[6] = make_float2(v00.x, v00.y); f[2] = make_float2(v00.z, v00.w);
    f[7] = make_float2(v01.x, v01.y); f[3] = make_float2(v01.z, v01.w);
    f[5] = make_float2(v10.x, v10.y); f[1] = make_float2(v10.z, v10.w);
    f[4] = make_float2(v11.x, v11.y); f[0] = make_float2(v11.z, v11.w);
}
// Pins the interpolation of a level BELOW the point where this is called: the interpolation weights pass through an opaque
// asm, so the (pure) arithmetic that reads them cannot be placed earlier — without it instruction selection emits each
// level's interpolation right behind its own eight gathers and the load queue drains once per level.
__device__ __forceinline__ void hash_hold(HashTaps &t) {
    asm volatile("" : "+v"(t.ox), "+v"(t.oy), "+v"(t.oz)::"memory");
}
__device__ __forceinline__ float2 hash_blend(const HashTaps &t, const float2 (&f)[8]) {
    float2 r;
    {
        const float f03 = lerp_t<true>(f[0].x, f[3].x, t.ox), f12 = lerp_t<true>(f[1].x, f[2].x, t.ox);
        const float f56 = lerp_t<true>(f[5].x, f[6].x, t.ox), f47 = lerp_t<true>(f[4].x, f[7].x, t.ox);
        const float f0312 = lerp_t<true>(f03, f12, t.oy), f4756 = lerp_t<true>(f47, f56, t.oy);
        r.x = lerp_t<true>(f0312, f4756, t.oz);
    }
    {
        const float f03 = lerp_t<true>(f[0].y, f[3].y, t.ox), f12 = lerp_t<true>(f[1].y, f[2].y, t.ox);
        const float f56 = lerp_t<true>(f[5].y, f[6].y, t.ox), f47 = lerp_t<true>(f[4].y, f[7].y, t.ox);
        const float f0312 = lerp_t<true>(f03, f12, t.oy), f4756 = lerp_t<true>(f47, f56, t.oy);
        r.y = lerp_t<true>(f0312, f4756, t.oz);
    }
    return r;
}

// NL hashed levels of one position (FAST flavour), software-pipelined over groups of LG levels: the 8*LG gathers of group
// g+1 are issued before group g is interpolated, so two groups (2 x 8*LG gathers, <= 64 = the vmcnt range) are in flight and
// the memory round trip is paid about once per position instead of once per level.  Levels base .. base+NL-1;
// emit(level - base, features) is called in level order.  Values are those of encode_level<false, true>, bit for bit.
// ND > 0 (with base == 0): levels 0 .. ND-1 are read from the dense re-layout (a compile-time split: a run-time branch between
// the stages would cost the exact wait counts the pipelining lives on).
// GP > 0: s_setprio(GP) while a group's indices are computed and its gathers issued, s_setprio(0) for the interpolation (a caller's
// A/B switch: the wave whose memory requests can go out ahead of its SIMD partners' arithmetic).
// (hash_encode_pipelined_raw hands the level's taps and corner values to emit_raw(level - base, taps, corners): a caller that
// wants more than the blended features — the training forward's position Jacobian — interpolates itself.)
template <int NL, int LG, int ND = 0, int GP = 0, typename G, typename EmitRaw>
__device__ __forceinline__ void hash_encode_pipelined_raw(const G &g, float px, float py, float pz, EmitRaw emit_raw, int base = 0) {
    static_assert(NL % LG == 0 && 16 * LG <= 64, "two groups of 8*LG gathers must fit the 6-bit vmcnt counter");
    constexpr int NG = NL / LG;
    HashTaps taps[2][LG];
    float2 fv[2][LG][8];
    auto taps_of = [&](int lvl, HashTaps &t) {  // lvl is a constant after unrolling
        if (lvl < ND) dense_taps(g, lvl, px, py, pz, t); else hash_taps(g, base + lvl, px, py, pz, t);
    };
    auto gather_of = [&](int lvl, const HashTaps &t, float2 (&f)[8]) {
        if (lvl < ND) dense_gather(g, lvl, t, f); else hash_gather(g, base + lvl, t, f);
    };
    if (GP > 0) __builtin_amdgcn_s_setprio(GP);
#pragma unroll
    for (int q = 0; q < LG; ++q) taps_of(q, taps[0][q]);
    TN_STAGE_FENCE();
#pragma unroll
    for (int q = 0; q < LG; ++q) gather_of(q, taps[0][q], fv[0][q]);
    TN_STAGE_FENCE();
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const int cur = gi & 1, nxt = cur ^ 1;
        if (gi + 1 < NG) {
#pragma unroll
            for (int q = 0; q < LG; ++q) taps_of((gi + 1) * LG + q, taps[nxt][q]);
            TN_STAGE_FENCE();
#pragma unroll
            for (int q = 0; q < LG; ++q) gather_of((gi + 1) * LG + q, taps[nxt][q], fv[nxt][q]);
        }
#pragma unroll
        for (int q = 0; q < LG; ++q) hash_hold(taps[cur][q]);  // group gi is interpolated below group gi+1's gathers
        if (GP > 0) __builtin_amdgcn_s_setprio(0);
        TN_STAGE_FENCE();
#pragma unroll
        for (int q = 0; q < LG; ++q) emit_raw(gi * LG + q, taps[cur][q], fv[cur][q]);
        if (GP > 0 && gi + 1 < NG) __builtin_amdgcn_s_setprio(GP);
        TN_STAGE_FENCE();
    }
}
template <int NL, int LG, int ND = 0, int GP = 0, typename G, typename Emit>
__device__ __forceinline__ void hash_encode_pipelined(const G &g, float px, float py, float pz, Emit emit, int base = 0) {
    hash_encode_pipelined_raw<NL, LG, ND, GP>(g, px, py, pz, [&](int l, const HashTaps &t, const float2 (&f)[8]) { emit(l, hash_blend(t, f)); }, base);
}
// hash_blend's features AND their derivatives with respect to the three interpolation offsets (jac[c] = d features / d offset c;
// times the level's scale = d features / d normalised position).  The differences are the ones the lerps form anyway.
__device__ __forceinline__ float2 hash_blend_jac(const HashTaps &t, const float2 (&f)[8], float2 (&jac)[3]) {
    float2 r;
#define TN_BLEND_JAC(c)                                                                                              \
    {                                                                                                                \
        const float d03 = f[0].c - f[3].c, d12 = f[1].c - f[2].c, d56 = f[5].c - f[6].c, d47 = f[4].c - f[7].c;      \
        const float f03 = fmaf(t.ox, d03, f[3].c), f12 = fmaf(t.ox, d12, f[2].c);                                    \
        const float f56 = fmaf(t.ox, d56, f[6].c), f47 = fmaf(t.ox, d47, f[7].c);                                    \
        const float e0312 = f03 - f12, e4756 = f47 - f56;                                                            \
        const float f0312 = fmaf(t.oy, e0312, f12), f4756 = fmaf(t.oy, e4756, f56);                                  \
        const float ez = f0312 - f4756;                                                                              \
        r.c = fmaf(t.oz, ez, f4756);                                                                                 \
        const float dx_hi = fmaf(t.oy, d03 - d12, d12), dx_lo = fmaf(t.oy, d47 - d56, d56);                          \
        jac[0].c = fmaf(t.oz, dx_hi - dx_lo, dx_lo);                                                                 \
        jac[1].c = fmaf(t.oz, e0312 - e4756, e4756);                                                                 \
        jac[2].c = ez;                                                                                               \
    }
    TN_BLEND_JAC(x)
    TN_BLEND_JAC(y)
#undef TN_BLEND_JAC
    // torch's corners are ceil / floor of the scaled coordinate: on a grid plane (offset exactly 0 — one (sample, axis, level) in
    // ~1e3 samples at fp32, the fine levels' coordinates have 13 fraction bits) the two coincide and the feature has no slope along
    // that axis, while the floor / floor + 1 corners read here would give the right-hand one
    if (t.ox == 0.0f) jac[0] = make_float2(0.0f, 0.0f);
    if (t.oy == 0.0f) jac[1] = make_float2(0.0f, 0.0f);
    if (t.oz == 0.0f) jac[2] = make_float2(0.0f, 0.0f);
    return r;
}

template <bool FAST = false>
__device__ __forceinline__ float2 encode_level_any(const Grid &g, int l, float px, float py, float pz) {
    if (l < g.num_dense) return encode_level<true, FAST>(g, l, px, py, pz);
    return encode_level<false, FAST>(g, l, px, py, pz);
}

// ---- gradient of the encoding w.r.t. the position (camera-pose optimisation) ---------------------------
// (gx, gy, gz) += d (ge . enc_l) / d p for one level: d enc / d offset from the 8 corners (table reads, torch's ceil / floor
// corner convention), times the level scale; p = the normalised position with the selector applied.
__device__ __forceinline__ void encode_level_grad(const Grid &g, int l, float px, float py, float pz, float2 ge, float &gx,
                                                  float &gy, float &gz) {
    const float s = g.scal[l];
    const float sx = mul_rn(px, s), sy = mul_rn(py, s), sz = mul_rn(pz, s);
    const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
    const float ox = sub_rn(sx, fxf), oy = sub_rn(sy, fyf), oz = sub_rn(sz, fzf);
    const float qx = 1.0f - ox, qy = 1.0f - oy, qz = 1.0f - oz;
    const unsigned cx = (unsigned)(int)ceilf(sx), cy = (unsigned)(int)ceilf(sy), cz = (unsigned)(int)ceilf(sz);
    const unsigned fx = (unsigned)(int)fxf, fy = (unsigned)(int)fyf, fz = (unsigned)(int)fzf;
    const unsigned hcy = cy * TN_P1, hfy = fy * TN_P1, hcz = cz * TN_P2, hfz = fz * TN_P2;
    const float2 *t = g.table + (size_t)l * g.tsize;
    const unsigned m = g.mask;
    const float2 f0 = t[(cx ^ hcy ^ hcz) & m], f1 = t[(cx ^ hfy ^ hcz) & m], f2 = t[(fx ^ hfy ^ hcz) & m];
    const float2 f3 = t[(fx ^ hcy ^ hcz) & m], f4 = t[(cx ^ hcy ^ hfz) & m], f5 = t[(cx ^ hfy ^ hfz) & m];
    const float2 f6 = t[(fx ^ hfy ^ hfz) & m], f7 = t[(fx ^ hcy ^ hfz) & m];
    // enc = ((f0 ox + f3 qx) oy + (f1 ox + f2 qx) qy) oz + ((f4 ox + f7 qx) oy + (f5 ox + f6 qx) qy) qz
#define TN_DENC(c)                                                                                                   \
    {                                                                                                                \
        const float f03 = f0.c * ox + f3.c * qx, f12 = f1.c * ox + f2.c * qx;                                        \
        const float f47 = f4.c * ox + f7.c * qx, f56 = f5.c * ox + f6.c * qx;                                        \
        const float dox = ((f0.c - f3.c) * oy + (f1.c - f2.c) * qy) * oz + ((f4.c - f7.c) * oy + (f5.c - f6.c) * qy) * qz; \
        const float doy = (f03 - f12) * oz + (f47 - f56) * qz;                                                       \
        const float doz = (f03 * oy + f12 * qy) - (f47 * oy + f56 * qy);                                             \
        gx += ge.c * dox * s;                                                                                        \
        gy += ge.c * doy * s;                                                                                        \
        gz += ge.c * doz * s;                                                                                        \
    }
    TN_DENC(x)
    TN_DENC(y)
#undef TN_DENC
}

// ... and back from p to the world position (x, y, z): through `p * selector`, the (c + 2) / 4 shift and the L-inf contraction
// (torch's inf-norm backward splits the subgradient evenly among tied maxima), or the AABB normalisation
__device__ __forceinline__ void position_grad_finish(const Space &sp, float x, float y, float z, float sel, float gx, float gy,
                                                     float gz, float &rx, float &ry, float &rz) {
    gx *= sel; gy *= sel; gz *= sel;  // p = p * selector
    if (sp.contraction) {
        gx *= 0.25f; gy *= 0.25f; gz *= 0.25f;  // (c + 2) / 4
        const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
        const float mag = fmaxf(fmaxf(ax, ay), az);
        if (mag < 1.0f) {
            rx = gx; ry = gy; rz = gz;
        } else {
            // c = s(m) x,  s = 2/m - 1/m^2,  m = |x_k| (k = arg max)
            const float sm = 2.0f / mag - 1.0f / (mag * mag);
            const float dsm = -2.0f / (mag * mag) + 2.0f / (mag * mag * mag);
            const float dot = gx * x + gy * y + gz * z;
            rx = sm * gx; ry = sm * gy; rz = sm * gz;
            const int ties = (ax == mag) + (ay == mag) + (az == mag);
            const float share = dsm * dot / (float)ties;
            if (ax == mag) rx += share * (x > 0.0f ? 1.0f : -1.0f);
            if (ay == mag) ry += share * (y > 0.0f ? 1.0f : -1.0f);
            if (az == mag) rz += share * (z > 0.0f ? 1.0f : -1.0f);
        }
    } else {
        rx = gx / (sp.mx[0] - sp.mn[0]); ry = gy / (sp.mx[1] - sp.mn[1]); rz = gz / (sp.mx[2] - sp.mn[2]);
    }
}

// ---- the expected-depth clip's running [min, max] of the sample mid-points ------------------------------------------
// monotone float <-> uint key, so unsigned atomicMin / atomicMax order floats
__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// Which [min, max] key pair a ray of a call belongs to.  DepthRenderer("expected") clips to the bounds of its CALL, which in the
// reference is one eval_num_rays_per_chunk chunk of the frame; a launch that covers several such chunks (or parts of them:
// tn_field_render_chunked_fwd) keeps one pair per chunk: chunk = (first_ray + ray) / chunk_rays, slot 0 = the call's first chunk.
// chunk_rays == 0: one pair for the whole call.
struct DepthSlots {
    unsigned *keys;
    long long first_ray, chunk_rays;
    __device__ __forceinline__ long long slot(long long ray) const {
        return chunk_rays > 0 ? (first_ray + ray) / chunk_rays - first_ray / chunk_rays : 0;
    }
};
// wave-reduce the lanes' running bounds into the pair of `slot` (ONE atomic pair per wave and slot: per-ray atomics on one
// address serialise in L2 at ~10 ns each) and restart them.  Every lane of the wave must be active.
__device__ __forceinline__ void depth_bounds_flush(const DepthSlots &m, long long slot, float &smin, float &smax, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        smin = fminf(smin, __shfl_xor(smin, o, 64));
        smax = fmaxf(smax, __shfl_xor(smax, o, 64));
    }
    if (lane == 0 && smin <= smax) {
        atomicMin(&m.keys[2 * slot], f2key(smin));
        atomicMax(&m.keys[2 * slot + 1], f2key(smax));
    }
    smin = INFINITY;
    smax = -INFINITY;
}

// ---- wave64 collectives ------------------------------------------------------------------------------
// The wave scans on the vector unit's data-parallel primitives instead of ds_bpermute steps (six dependent LDS round trips per
// scan in kernels that are one latency chain per ray): row_shr 1 / 2 / 4 / 8 inside the rows of 16, row_bcast:15 into rows 1
// and 3, row_bcast:31 into rows 2 and 3 — the GCN wave64 scan; a lane without a source keeps its value.  All lanes must be active.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}
// PRECONDITION (DPP form): every lane of the wave is active — lane 63 in particular, whose scan value is read back; a
// caller inside divergent control flow must hoist the call out of it (all ~40 call sites sit at wave-uniform points: the
// per-ray kernels run one ray per FULL wave and mask lanes by value, not by branch).
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_incl_scan(v, 0)), 63));
}

// one lane's value in every lane (v_readlane_b32: a scalar move, no LDS round trip; the lane must be active)
template <int LANE>
__device__ __forceinline__ float lane_value(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
}

// exclusive prefix from an inclusive one (robust to inf entries: no inf - inf)
__device__ __forceinline__ float wave_excl_from_incl(float incl, int lane) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(incl), 0x138, 0xf, 0xf, false));  // wave_shr:1, lane 0 <- 0
}

}  // namespace tn

// Layout of the final S+1 bin edges handed from the proposal kernel to the field kernel through the workspace:
// ray-tiled [ceil(R/64)][S+1][64], so a wave that owns 64 consecutive rays reads 64 consecutive floats per edge.
__host__ __device__ static inline size_t tn_ws_bin(long long r, int j, int S) {
    return ((size_t)(r >> 6) * (size_t)(S + 1) + (size_t)j) * 64 + (size_t)(r & 63);
}
__host__ __device__ static inline size_t tn_ws_bin_floats(long long num_rays, int S) {
    return (size_t)((num_rays + 63) >> 6) * 64 * (size_t)(S + 1);
}

#ifdef __HIPCC__
struct WsBins {  // one ray's bin edges inside the ray-tiled workspace: edge j lives 64 floats after edge j-1
    const float *base;
    __device__ __forceinline__ float operator[](int j) const { return base[(size_t)j * 64]; }
};
#endif

// host-side conversion of the C-ABI grid struct into the by-value kernel argument
static inline tn::Grid tn_make_grid(const tn_hashgrid &h) {
    tn::Grid g;
    g.table = reinterpret_cast<const float2 *>(h.table);
    g.dense = reinterpret_cast<const float2 *>(h.dense);
    for (int i = 0; i < TN_MAX_LEVELS; ++i) {
        g.scal[i] = h.scalings[i];
        g.dense_off[i] = h.dense_offset[i];
        g.dense_res[i] = h.dense_res[i];
    }
    g.num_levels = h.num_levels;
    g.num_dense = h.dense ? h.num_dense_levels : 0;
    g.tsize = 1u << h.log2_hashmap_size;
    g.mask = g.tsize - 1u;
    return g;
}

static inline int tn_check_grid(const tn_hashgrid &h) {
    if (!h.table) return TN_ERR_NULL;
    if (h.num_levels < 1 || h.num_levels > TN_MAX_LEVELS) return TN_ERR_SHAPE;
    if (h.log2_hashmap_size < 1 || h.log2_hashmap_size > 24) return TN_ERR_SHAPE;
    return TN_OK;
}

// Opt a kernel in to `bytes` of dynamic LDS (> 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize).  The attribute is a
// property of the loaded code object per device, so it is set ONCE per process, device and kernel (the largest size granted
// so far is remembered) instead of on every launch: an immutable, idempotent initialisation, not mutable library state.
#ifdef __HIPCC__
#include <atomic>
template <auto Kernel>
static inline bool tn_ensure_dynamic_lds(size_t bytes) {
    constexpr int kMaxDev = 64;
    static std::atomic<size_t> granted[kMaxDev];
    int dev = 0;
    const bool cached = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev;
    if (cached && granted[dev].load(std::memory_order_acquire) >= bytes) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess)
        return false;
    if (cached) {
        size_t cur = granted[dev].load(std::memory_order_relaxed);
        while (cur < bytes && !granted[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {
        }
    }
    return true;
}
#endif

#define TN_LAUNCH_CHECK()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return TN_ERR_LAUNCH; \
    } while (0)
