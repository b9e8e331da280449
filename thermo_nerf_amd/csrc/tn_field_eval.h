// Per-sample evaluation of the two field types with MLP weights staged in LDS (lane-per-sample form).
// Shared by the per-sample plugin-surface kernels (tn_fields.hip) and the reference-form fused ray kernel
// (tn_render.hip, TN_MAIN_VALU path).  Follows SURVEY.md §8a rows a5, a8, a9.
#pragma once
#include "tn_device.h"

namespace tn {

constexpr int HW = 64;  // hidden width of mlp_base / mlp_head / mlp_thermal

// ---- two-layer "hash grid -> Linear(2L,H)+ReLU -> Linear(H,NOUT)" block ------------------------------
// LDS layout: W0t [2L][H] | b0 [H] | W1 [NOUT][H] | b1 [NOUT]
struct TwoLayerLds {
    const float *W0t, *B0, *W1, *B1;
};
__host__ __device__ inline int two_layer_floats(int in_dim, int H, int nout) { return in_dim * H + H + nout * H + nout; }

template <int H>
__device__ __forceinline__ TwoLayerLds stage_two_layer(float *smem, const float *w0, const float *b0,
                                                       const float *w1, const float *b1, int in_dim, int nout) {
    float *W0t = smem;
    float *B0 = W0t + in_dim * H;
    float *W1 = B0 + H;
    float *B1 = W1 + nout * H;
    for (int i = threadIdx.x; i < in_dim * H; i += blockDim.x) {
        const int k = i / H, h = i - k * H;
        W0t[i] = w0[h * in_dim + k];
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) B0[i] = b0[i];
    for (int i = threadIdx.x; i < nout * H; i += blockDim.x) W1[i] = w1[i];
    for (int i = threadIdx.x; i < nout; i += blockDim.x) B1[i] = b1[i];
    TwoLayerLds r{W0t, B0, W1, B1};
    return r;
}

// NL > 0: level count known at compile time -> the level loop unrolls and all 8*NL gathers of a sample are
// issued before the first one is consumed (latency paid once per sample instead of once per level).
template <int H, int NL = 0, bool FAST = false>
__device__ __forceinline__ void hidden_from_grid(const Grid &g, const TwoLayerLds &w, float px, float py, float pz,
                                                 float (&hid)[H]) {
#pragma unroll
    for (int h = 0; h < H; ++h) hid[h] = w.B0[h];
    if (NL > 0) {
        float2 f[NL > 0 ? NL : 1];
        if (g.num_dense == 0) {  // uniform branch hoisted out so the unrolled gathers are straight-line code
#pragma unroll
            for (int l = 0; l < NL; ++l) f[l] = encode_level<false, FAST>(g, l, px, py, pz);
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) f[l] = encode_level_any<FAST>(g, l, px, py, pz);
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float *wa = w.W0t + (2 * l) * H;
            const float *wb = wa + H;
#pragma unroll
            for (int h = 0; h < H; ++h) hid[h] = fmaf(wa[h], f[l].x, hid[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) hid[h] = fmaf(wb[h], f[l].y, hid[h]);
        }
    } else {
        for (int l = 0; l < g.num_levels; ++l) {
            const float2 f = encode_level_any<FAST>(g, l, px, py, pz);
            const float *wa = w.W0t + (2 * l) * H;
            const float *wb = wa + H;
#pragma unroll
            for (int h = 0; h < H; ++h) hid[h] = fmaf(wa[h], f.x, hid[h]);
#pragma unroll
            for (int h = 0; h < H; ++h) hid[h] = fmaf(wb[h], f.y, hid[h]);
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) hid[h] = fmaxf(hid[h], 0.0f);
}

// HashMLPDensityField: density = avg * exp(mlp(enc(p))) * selector
template <int H, int NL = 0, bool FAST = false>
__device__ __forceinline__ float proposal_density_eval(const Grid &g, const TwoLayerLds &w, float avg, float px,
                                                       float py, float pz, float sel) {
    float hid[H];
    hidden_from_grid<H, NL, FAST>(g, w, px, py, pz, hid);
    float o = w.B1[0];
#pragma unroll
    for (int h = 0; h < H; ++h) o = fmaf(w.W1[h], hid[h], o);
    return mul_rn(mul_rn(avg, t_exp<FAST>(o)), sel);
}

// The same density with the MLP weights read as SCALAR operands (constant address space -> s_load, SGPR sources): a
// wave-uniform weight broadcast from LDS still costs LDS->VGPR bandwidth for 64 lanes (8 clk per ds_read_b128), and the
// ~50 broadcast reads per sample of the LDS form keep the LDS pipe ~75 % busy at 12 waves per CU.  Same accumulation
// order as hidden_from_grid (per hidden unit: bias, then the features in level order): bit-identical results.
typedef __attribute__((address_space(4))) const float tn_cfloat;
__device__ __forceinline__ const tn_cfloat *as_scalar(const float *p) { return (const tn_cfloat *)p; }

template <int H, int NL, bool FAST>
__device__ __forceinline__ float proposal_density_scalar(const Grid &g, const tn_cfloat *w0, const tn_cfloat *b0,
                                                         const tn_cfloat *w1, const tn_cfloat *b1, float avg, float px, float py,
                                                         float pz, float sel) {
    float2 f[NL];
    if (g.num_dense == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) f[l] = encode_level<false, FAST>(g, l, px, py, pz);
    } else {
#pragma unroll
        for (int l = 0; l < NL; ++l) f[l] = encode_level_any<FAST>(g, l, px, py, pz);
    }
    float o = b1[0];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        float a = b0[h];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            a = fmaf(w0[h * 2 * NL + 2 * l], f[l].x, a);
            a = fmaf(w0[h * 2 * NL + 2 * l + 1], f[l].y, a);
        }
        o = fmaf(w1[h], fmaxf(a, 0.0f), o);
    }
    return mul_rn(mul_rn(avg, t_exp<FAST>(o)), sel);
}

// ---- SH degree-3 basis (16 comps), NS components_from_spherical_harmonics -----------------------------
__device__ __forceinline__ void sh16(float x, float y, float z, float (&c)[16]) {
    const float xx = x * x, yy = y * y, zz = z * z;
    c[0] = 0.28209479177387814f;
    c[1] = 0.4886025119029199f * y;
    c[2] = 0.4886025119029199f * z;
    c[3] = 0.4886025119029199f * x;
    c[4] = 1.0925484305920792f * x * y;
    c[5] = 1.0925484305920792f * y * z;
    c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
    c[7] = 1.0925484305920792f * x * z;
    c[8] = 0.5462742152960396f * (xx - yy);
    c[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
    c[10] = 2.890611442640554f * x * y * z;
    c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
    c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
    c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
    c[14] = 1.445305721320277f * z * (xx - yy);
    c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- colour + thermal heads ---------------------------------------------------------------------------
struct HeadsArgs {
    const float *h0w, *h0b, *h1w, *h1b, *h2w, *h2b;
    const float *t0w, *t0b, *t1w, *t1b, *thw, *thb;
    const float *appearance;
    int num_images, app_dim, geo_dim, use_avg, sh_shifted;
};

// LDS: H0t [in0][64] | H0b [64] | H1t [64][64] | H1b [64] | H2 [3][64] | H2b [4] | T0t [geo][64] | T0b [64] |
//      T1t [64][64] | T1b [64] | TH [64] | THb [4] | APP [app_dim]
struct HeadsLds {
    const float *H0t, *H0b, *H1t, *H1b, *H2, *H2b, *T0t, *T0b, *T1t, *T1b, *TH, *THb, *APP;
};
__host__ __device__ inline int heads_floats(int G, int A) {
    return (16 + G + A) * HW + HW + HW * HW + HW + 3 * HW + 4 + G * HW + HW + HW * HW + HW + HW + 4 + A;
}

__device__ __forceinline__ HeadsLds stage_heads(float *smem, const HeadsArgs &a, int training) {
    const int G = a.geo_dim, A = a.app_dim;
    const int in0 = 16 + G + A;
    float *H0t = smem;
    float *H0b = H0t + in0 * HW;
    float *H1t = H0b + HW;
    float *H1b = H1t + HW * HW;
    float *H2 = H1b + HW;
    float *H2b = H2 + 3 * HW;
    float *T0t = H2b + 4;
    float *T0b = T0t + G * HW;
    float *T1t = T0b + HW;
    float *T1b = T1t + HW * HW;
    float *TH = T1b + HW;
    float *THb = TH + HW;
    float *APP = THb + 4;
    for (int i = threadIdx.x; i < in0 * HW; i += blockDim.x) {
        const int k = i / HW, h = i - k * HW;
        H0t[i] = a.h0w[h * in0 + k];
    }
    for (int i = threadIdx.x; i < HW * HW; i += blockDim.x) {
        const int k = i / HW, h = i - k * HW;
        H1t[i] = a.h1w[h * HW + k];
        T1t[i] = a.t1w[h * HW + k];
    }
    for (int i = threadIdx.x; i < G * HW; i += blockDim.x) {
        const int k = i / HW, h = i - k * HW;
        T0t[i] = a.t0w[h * G + k];
    }
    for (int i = threadIdx.x; i < 3 * HW; i += blockDim.x) H2[i] = a.h2w[i];
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        H0b[i] = a.h0b[i];
        H1b[i] = a.h1b[i];
        T0b[i] = a.t0b[i];
        T1b[i] = a.t1b[i];
        TH[i] = a.thw[i];
    }
    if (threadIdx.x < 3) H2b[threadIdx.x] = a.h2b[threadIdx.x];
    if (threadIdx.x == 0) THb[0] = a.thb[0];
    // eval-mode appearance vector: mean over images (REF thermal_field.py:128-132) or zeros (:133-137)
    for (int k = threadIdx.x; k < A; k += blockDim.x) {
        float m = 0.0f;
        if (!training && a.use_avg) {
            for (int im = 0; im < a.num_images; ++im) m += a.appearance[(long long)im * A + k];
            m = m / (float)a.num_images;
        }
        APP[k] = m;
    }
    HeadsLds r{H0t, H0b, H1t, H1b, H2, H2b, T0t, T0b, T1t, T1b, TH, THb, APP};
    return r;
}

// geo: pointer to this sample's geo features (global or private memory); app: appearance vector to use.
// GC > 0 fixes geo_dim at compile time (register-resident geo), GC == 0 uses the runtime G.
template <int GC>
__device__ __forceinline__ void heads_eval(const HeadsLds &w, int G, int A, int sh_shifted, float dx, float dy,
                                           float dz, const float *geo, const float *app, float (&rgb)[3],
                                           float &thermal) {
    float h1[HW];
#pragma unroll
    for (int h = 0; h < HW; ++h) h1[h] = w.H0b[h];
    {   // SH of the direction, REF thermal_field.py:117-119
        if (sh_shifted) {
            dx = add_rn(dx, 1.0f) / 2.0f;
            dy = add_rn(dy, 1.0f) / 2.0f;
            dz = add_rn(dz, 1.0f) / 2.0f;
        }
        float c[16];
        sh16(dx, dy, dz, c);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float *wr = w.H0t + k * HW;
#pragma unroll
            for (int h = 0; h < HW; ++h) h1[h] = fmaf(wr[h], c[k], h1[h]);
        }
    }
    float t1[HW];
#pragma unroll
    for (int h = 0; h < HW; ++h) t1[h] = w.T0b[h];
    const int Gn = GC > 0 ? GC : G;
#pragma unroll
    for (int k = 0; k < Gn; ++k) {  // geo feeds both branches, REF :160-167, :171
        const float x = geo[k];
        const float *wr = w.H0t + (16 + k) * HW;
        const float *wt = w.T0t + k * HW;
#pragma unroll
        for (int h = 0; h < HW; ++h) h1[h] = fmaf(wr[h], x, h1[h]);
#pragma unroll
        for (int h = 0; h < HW; ++h) t1[h] = fmaf(wt[h], x, t1[h]);
    }
    for (int k = 0; k < A; ++k) {
        const float x = app[k];
        const float *wr = w.H0t + (16 + Gn + k) * HW;
#pragma unroll
        for (int h = 0; h < HW; ++h) h1[h] = fmaf(wr[h], x, h1[h]);
    }
    // colour branch: REF :168 (mlp_head 3 layers, Sigmoid)
    float h2[HW];
#pragma unroll
    for (int h = 0; h < HW; ++h) h2[h] = w.H1b[h];
#pragma unroll 4
    for (int k = 0; k < HW; ++k) {
        const float x = fmaxf(h1[k], 0.0f);
        const float *wr = w.H1t + k * HW;
#pragma unroll
        for (int h = 0; h < HW; ++h) h2[h] = fmaf(wr[h], x, h2[h]);
    }
    float o0 = w.H2b[0], o1 = w.H2b[1], o2 = w.H2b[2];
#pragma unroll
    for (int h = 0; h < HW; ++h) {
        const float x = fmaxf(h2[h], 0.0f);
        o0 = fmaf(w.H2[h], x, o0);
        o1 = fmaf(w.H2[HW + h], x, o1);
        o2 = fmaf(w.H2[2 * HW + h], x, o2);
    }
    rgb[0] = sigmoidf(o0);
    rgb[1] = sigmoidf(o1);
    rgb[2] = sigmoidf(o2);
    // thermal branch: REF :171-179 (mlp_thermal 2 layers, Sigmoid; Linear head, no activation)
#pragma unroll
    for (int h = 0; h < HW; ++h) h2[h] = w.T1b[h];
#pragma unroll 4
    for (int k = 0; k < HW; ++k) {
        const float x = fmaxf(t1[k], 0.0f);
        const float *wr = w.T1t + k * HW;
#pragma unroll
        for (int h = 0; h < HW; ++h) h2[h] = fmaf(wr[h], x, h2[h]);
    }
    float t = w.THb[0];
#pragma unroll
    for (int h = 0; h < HW; ++h) t = fmaf(w.TH[h], sigmoidf(h2[h]), t);
    thermal = t;
}

inline HeadsArgs make_heads_args(const tn_thermal_field *f) {
    HeadsArgs a;
    a.h0w = f->head0.weight; a.h0b = f->head0.bias; a.h1w = f->head1.weight; a.h1b = f->head1.bias;
    a.h2w = f->head2.weight; a.h2b = f->head2.bias; a.t0w = f->th0.weight; a.t0b = f->th0.bias;
    a.t1w = f->th1.weight; a.t1b = f->th1.bias; a.thw = f->thead.weight; a.thb = f->thead.bias;
    a.appearance = f->appearance; a.num_images = f->num_images; a.app_dim = f->app_dim; a.geo_dim = f->geo_feat_dim;
    a.use_avg = f->use_average_appearance; a.sh_shifted = f->sh_shifted;
    return a;
}

}  // namespace tn

static inline int tn_check_linear(const tn_linear &l, int in_dim, int out_dim) {
    if (!l.weight || !l.bias) return TN_ERR_NULL;
    if (l.in_dim != in_dim || l.out_dim != out_dim) return TN_ERR_SHAPE;
    return TN_OK;
}

static inline int tn_check_density_field(const tn_density_field *f) {
    if (!f) return TN_ERR_NULL;
    int e = tn_check_grid(f->grid);
    if (e) return e;
    const int H = f->l0.out_dim;
    if ((e = tn_check_linear(f->l0, 2 * f->grid.num_levels, H))) return e;
    if ((e = tn_check_linear(f->l1, H, 1))) return e;
    if (H != 16 && H != 64) return TN_ERR_UNSUPPORTED;
    return TN_OK;
}

static inline int tn_check_thermal_field(const tn_thermal_field *f) {
    if (!f || !f->appearance) return TN_ERR_NULL;
    int e = tn_check_grid(f->grid);
    if (e) return e;
    const int G = f->geo_feat_dim, A = f->app_dim;
    if (G < 1 || G > 31 || A < 0 || A > 64 || f->num_images < 1) return TN_ERR_SHAPE;
    if ((e = tn_check_linear(f->base0, 2 * f->grid.num_levels, 64))) return e;
    if ((e = tn_check_linear(f->base1, 64, 1 + G))) return e;
    if ((e = tn_check_linear(f->head0, 16 + G + A, 64))) return e;
    if ((e = tn_check_linear(f->head1, 64, 64))) return e;
    if ((e = tn_check_linear(f->head2, 64, 3))) return e;
    if ((e = tn_check_linear(f->th0, G, 64))) return e;
    if ((e = tn_check_linear(f->th1, 64, 64))) return e;
    if ((e = tn_check_linear(f->thead, 64, 1))) return e;
    return TN_OK;
}

#define TN_TRY(expr)                \
    do {                            \
        const int _e = (expr);      \
        if (_e != TN_OK) return _e; \
    } while (0)
