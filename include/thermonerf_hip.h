/*
 * thermonerf_hip.h — C-ABI of libthermonerf_hip.so: the MI355X (gfx950) drop-in for ThermoNeRF's
 * volumetric-rendering hot path.
 *
 * The reference is 100 % Python (a nerfstudio plugin) and has no FFI of its own; the entry points below are
 * what a ctypes binding placed behind the reference's plugin surface binds (INTEGRATION.md shows the stub).
 * Each entry point cites the reference interface it replaces:
 *   REF = /root/reference/thermo_nerf/...        NS = nerfstudio 1.1.5 symbol invoked from that REF line.
 *
 * Conventions
 *   - every pointer named  *_dev / inside the structs  is a DEVICE pointer to contiguous row-major fp32
 *     (int32 where stated); the caller (PyTorch-ROCm) owns every buffer; nothing is allocated or freed here.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the device is never synchronised.
 *   - return value: TN_OK or a negative TN_ERR_* code; no exceptions cross the boundary.
 *   - re-entrant across devices/streams: the library keeps no mutable global state.
 *   - `training` != 0 selects nerfstudio's train-mode semantics (no nan_to_num / clamp in the renderers).
 */
#ifndef THERMONERF_HIP_H
#define THERMONERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TN_OK 0
#define TN_ERR_NULL (-1)        /* a required pointer is NULL                           */
#define TN_ERR_SHAPE (-2)       /* a dimension is out of the supported range            */
#define TN_ERR_UNSUPPORTED (-3) /* configuration not implemented by the kernels         */
#define TN_ERR_WORKSPACE (-4)   /* workspace too small (see tn_render_workspace_bytes)  */
#define TN_ERR_LAUNCH (-5)      /* hipGetLastError() != hipSuccess after a launch       */

#define TN_MAX_LEVELS 16

/* NS HashEncoding (torch path): table [L*T, 2], per-level scalings taken from the module's `scalings`
 * buffer (never recomputed here — SURVEY.md §8c). Built at REF thermal_nerf/thermal_field.py:62-88 (main
 * field, via NerfactoField) and REF thermal_nerf/thermal_nerf_model.py:136-149 (proposal nets). */
typedef struct tn_hashgrid {
    const float *table;            /* [num_levels << log2_hashmap_size, 2]                         */
    float scalings[TN_MAX_LEVELS]; /* HashEncoding.scalings                                        */
    int32_t num_levels;            /* 1..16                                                        */
    int32_t log2_hashmap_size;     /* 1..24                                                        */
    /* optional re-laid ("dense") copy of the coarse levels produced by tn_hashgrid_prepare: for level
     * l < num_dense_levels, dense + dense_offset[l] holds a [(res+2)^3, 2] x-major / z-fastest grid with
     * dense[x][y][z] = table[hash(x,y,z)] — a pure layout change, results are bit-identical.        */
    const float *dense;
    int64_t dense_offset[TN_MAX_LEVELS]; /* in float2 elements                                      */
    int32_t dense_res[TN_MAX_LEVELS];    /* grid side (= scalings[l] + 2)                           */
    int32_t num_dense_levels;
    int32_t _pad;
} tn_hashgrid;

/* torch.nn.Linear: weight [out_dim, in_dim], bias [out_dim]. */
typedef struct tn_linear {
    const float *weight;
    const float *bias;
    int32_t in_dim;
    int32_t out_dim;
} tn_linear;

/* Position normalisation shared by both density fields (NS NerfactoField.get_density /
 * HashMLPDensityField.get_density): contraction != 0 -> SceneContraction(order=inf) then (x+2)/4
 * [REF thermal_nerf_model.py:91-94]; else (x - aabb_min) / (aabb_max - aabb_min). */
typedef struct tn_space {
    int32_t contraction;
    float aabb_min[3];
    float aabb_max[3];
    int32_t _pad;
} tn_space;

/* NS HashMLPDensityField(num_layers=2, hidden_dim=16, use_linear=False), REF thermal_nerf_model.py:140-149. */
typedef struct tn_density_field {
    tn_hashgrid grid;
    tn_linear l0; /* [hidden, 2*L] + ReLU */
    tn_linear l1; /* [1, hidden]          */
    tn_space space;
    float average_init_density; /* 1.0 */
    int32_t _pad;
} tn_density_field;

/* ThermalNerfactoTField, REF thermal_nerf/thermal_field.py:33-201 (NerfactoField base + thermal branch). */
typedef struct tn_thermal_field {
    tn_hashgrid grid;
    tn_linear base0;  /* mlp_base.mlp.layers.0  [64, 32]  ReLU                    */
    tn_linear base1;  /* mlp_base.mlp.layers.1  [1+geo, 64]                        */
    tn_linear head0;  /* mlp_head.layers.0      [64, 16+geo+app] ReLU              */
    tn_linear head1;  /* mlp_head.layers.1      [64, 64] ReLU                      */
    tn_linear head2;  /* mlp_head.layers.2      [3, 64]  Sigmoid                   */
    tn_linear th0;    /* mlp_thermal.layers.0   [64, geo] ReLU          REF :90-98 */
    tn_linear th1;    /* mlp_thermal.layers.1   [64, 64] Sigmoid        REF :90-98 */
    tn_linear thead;  /* field_head_thermal.net [1, 64] no activation   REF :100-102, thermal_field_head.py:50-51 */
    const float *appearance; /* embedding_appearance.embedding.weight [num_images, app_dim]        */
    int32_t num_images;
    int32_t app_dim;       /* 32 */
    int32_t geo_feat_dim;  /* 15 */
    int32_t use_average_appearance; /* eval: mean(embedding) if != 0 else zeros   REF :128-137 */
    int32_t sh_shifted;    /* 1: SH evaluated on (d+1)/2 (torch fallback), 0: on d */
    tn_space space;
    float average_init_density; /* 1.0, REF thermal_field.py:86 */
    /* optional blob produced by tn_field_prepare (MFMA-fragment-ordered MLP weights); NULL = raw path */
    const float *prepared;
    /* optional blob produced by tn_field_prepare_f16x3: every fp32 weight split into two f16 halves so that the eval
     * field kernel can evaluate each fp32 product as three f16 MFMA products accumulated in fp32 (~2^-22 relative
     * product error, needs |activation| < 65504).  Takes precedence over `prepared` for eval calls when non-NULL. */
    const float *prepared_f16x3;
    /* optional blob produced by tn_field_prepare_bf16x6: every fp32 weight as THREE bf16 pieces (24 significand bits: the split is
     * exact) so that the eval field kernel can evaluate each fp32 product as the six piece products of order <= 2, accumulated in
     * fp32: a per-product error of 2^-23 relative — fp32's own rounding size — at the matrix cores' bf16 rate.  Takes precedence
     * over `prepared_f16x3` and `prepared` for eval calls when non-NULL.
     * ABI NOTE (round 4 appended this member): ZERO-INITIALISE the struct (`tn_thermal_field f = {0};` / memset) before filling it
     * — the three `prepared*` pointers are optional and a non-NULL one is dereferenced; a caller compiled against the older,
     * shorter struct must be recompiled.  RANGE NOTE (tests/test_gpu_parity.py::test_bf16x6_split_is_exact_...): x = p1 + p2 + p3
     * exactly for 2^-110 <= |x| <= (2 - 2^-8) 2^127 = 3.396e38.  Below, the third piece falls under bf16's smallest sub-normal
     * (2^-133) and the sum is off by < 2^-133 absolute; above, bf16(x) rounds to infinity (the field's activations are O(1)).
     * gfx950's bf16 MFMA takes sub-normal bf16 INPUTS as they are (test_bf16_mfma_takes_subnormal_inputs_as_they_are), so the small
     * pieces of small operands still contribute.  Measured on 1e6 random pairs over 60 binades: the six-product sum is within
     * 0.77 x 2^-23 of the fp64 product. */
    const float *prepared_bf16x6;
} tn_thermal_field;

/* ------------------------------------------------------------------------------------------------------
 * Per-sample entry points (the Field plugin surface)
 * ---------------------------------------------------------------------------------------------------- */

/* NS Cameras.generate_rays for one perspective camera, as called by the reference's harnesses
 * [REF thermo_nerf/render/renderer.py:182-184; thermo_nerf/evaluator/evaluator.py:68-70]: pixels
 * [first_pixel, first_pixel + num_pixels) of the row-major H x W image (pixel centres at +0.5) ->
 * origins [n,3], unit directions [n,3], pixel_area [n] (may be NULL).  c2w_host = 12 HOST floats, row-major [3,4].
 * distortion_host = 6 HOST floats (k1, k2, k3, k4, p1, p2: NS get_distortion_params order) or NULL; when any is non-zero
 * the normalised pixel coordinates go through NS camera_utils.radial_and_tangential_undistort (10 Newton steps). */
int tn_generate_rays(const float *c2w_host, float fx, float fy, float cx, float cy, int32_t height, int32_t width,
                     const float *distortion_host, int64_t first_pixel, int64_t num_pixels, float *origins,
                     float *directions, float *pixel_area, void *stream);

/* NS Frustums.get_positions: pos = origins + directions * (starts + ends) / 2.
 * origins/directions [R,3]; starts/ends [R,n]; positions out [R,n,3]. */
int tn_frustum_positions(const float *origins, const float *directions, const float *starts, const float *ends,
                         int64_t num_rays, int32_t n, float *positions, void *stream);

/* The same positions straight from a level's bin edges eucl_bins [R,n+1] (RaySamples.frustums.starts / .ends are its two
 * edge columns, NS Sampler.generate_ray_samples via get_ray_samples), also writing the contiguous starts / ends / deltas
 * [R,n] the training step's adjoint kernels read: one launch for the level's geometry. */
int tn_frustum_from_edges(const float *origins, const float *directions, const float *eucl_bins, int64_t num_rays,
                          int32_t n, float *positions, float *starts, float *ends, float *deltas, void *stream);

/* HashMLPDensityField.density_fn(positions) [REF thermal_nerf_model.py:146-149,222-224]:
 * positions [N,3] -> density [N]. */
int tn_density_fwd(const tn_density_field *field, const float *positions, int64_t n, float *density, void *stream);

/* NerfactoField.get_density as used by ThermalNerfactoTField.forward [REF thermal_field.py:186-190]:
 * positions [N,3] -> density [N], geo embedding [N, geo_feat_dim]. */
int tn_field_density_fwd(const tn_thermal_field *field, const float *positions, int64_t n, float *density,
                         float *geo, void *stream);

/* ThermalNerfactoTField.get_outputs [REF thermal_field.py:108-181]:
 * directions [N,3] (un-normalised to [0,1]; the kernel applies (d+1)/2), geo [N,geo], camera_indices [N] int32
 * (only read when training != 0) -> rgb [N,3], thermal [N]. */
int tn_field_heads_fwd(const tn_thermal_field *field, const float *directions, const float *geo,
                       const int32_t *camera_indices, int64_t n, int32_t training, float *rgb, float *thermal,
                       void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Samplers and weights (NS ray_samplers.py / RaySamples.get_weights, invoked at
 * REF thermal_nerf_model.py:172-179,222-224,233)
 * ---------------------------------------------------------------------------------------------------- */

/* The proposal sampler's initial sampler [REF thermal_nerf_model.py:164-170]: UniformLinDispPiecewiseSampler
 * (uniform_spacing == 0, the reference default) or UniformSampler (uniform_spacing == 1; spacing_fn and its inverse are the
 * identity).  spacing bins [n+1] (= torch.linspace(0,1,n+1), host-computed, device resident) (+ optional per-ray
 * stratified jitter t_rand [R], NULL in eval) -> spacing_bins [R,n+1], euclidean bins [R,n+1]. nears/fars [R].
 * Bit 1 of uniform_spacing (value 2, round 5) = NS single_jitter=False: t_rand is [R,n+1], one draw per bin edge. */
int tn_sample_initial(const float *lin_bins, const float *t_rand, const float *nears, const float *fars,
                      int64_t num_rays, int32_t n, int32_t uniform_spacing, float *spacing_bins, float *eucl_bins,
                      void *stream);

/* RaySamples.get_weights: deltas [R,n], densities [R,n] -> weights [R,n]. */
int tn_weights_fwd(const float *deltas, const float *densities, int64_t num_rays, int32_t n, float *weights,
                   void *stream);

/* PDFSampler.generate_ray_samples (histogram_padding 0.01, eps 1e-5): weights [R,n_in] (already annealed),
 * existing spacing bins [R,n_in+1], u [n_out+1] (host-computed eval positions) or u_rand [R] jitter (training;
 * NULL in eval), nears/fars [R] -> spacing_bins [R,n_out+1], eucl_bins [R,n_out+1].  uniform_spacing = the initial
 * sampler's spacing function, which PDFSampler reuses for the new bins (see tn_sample_initial); its bit 1 (value 2) =
 * single_jitter=False: u_rand is [R,n_out+1], one draw per new bin edge. */
int tn_sample_pdf(const float *weights, const float *existing_bins, const float *u, const float *u_rand,
                  const float *nears, const float *fars, int64_t num_rays, int32_t n_in, int32_t n_out,
                  int32_t uniform_spacing, float *spacing_bins, float *eucl_bins, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Renderers
 * ---------------------------------------------------------------------------------------------------- */

/* ThermalRenderer.forward [REF thermal_renderer.py:113-149] and NS RGBRenderer(background "last_sample")
 * [witness REF rgb_concat/rgbt_renderer.py:62-81,159-174]: values [R,n,C], weights [R,n] -> out [R,C]
 * = sum_s w*v + v[last]*(1 - sum_s w); eval (training == 0): nan_to_num(values) first, clamp [0,1] last. */
int tn_composite_fwd(const float *values, const float *weights, int64_t num_rays, int32_t n, int32_t channels,
                     int32_t training, float *out, void *stream);

/* NS AccumulationRenderer + DepthRenderer("median") + DepthRenderer("expected") in one pass
 * [REF thermal_nerf_model.py:238-243,267-270]. weights/starts/ends [R,n]; any output may be NULL.
 * expected depth is clipped to the call-global [min(steps), max(steps)] exactly as the reference does;
 * `minmax_scratch` = 2 floats of device scratch (required when expected != NULL). */
int tn_depth_fwd(const float *weights, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                 float *accumulation, float *median, float *expected, float *minmax_scratch, void *stream);

/* torchmetrics.functional.structural_similarity_index_measure with its defaults (gaussian 11 x 11, sigma 1.5, k1 0.01,
 * k2 0.03; torchmetrics 1.7.2, uv.lock:5468-5469) as the reference's models call it on a rendered frame
 * [REF thermal_nerf_model.py:195,363; NS NerfactoModel.get_image_metrics_and_images]: pred / target [H,W,C] device floats
 * (the renderers' layout), data_range = what torchmetrics derives when none is given, max(pred.max() - pred.min(),
 * target.max() - target.min()); out[0] = mean index over the windows inside the image and the channels.
 * height, width >= 11.  workspace: tn_ssim_workspace_bytes(height, width, channels) bytes of device scratch. */
size_t tn_ssim_workspace_bytes(int32_t height, int32_t width, int32_t channels);
int tn_ssim_fwd(const float *pred, const float *target, int32_t height, int32_t width, int32_t channels, float data_range,
                void *workspace, size_t workspace_bytes, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Fused forward: Model.forward (collider) + ThermalNerfModel.get_outputs
 * [REF thermal_nerf_model.py:210-275]
 * ---------------------------------------------------------------------------------------------------- */

typedef struct tn_render_config {
    int32_t num_proposal_samples[2]; /* (256, 96); 1..1024 each       */
    int32_t num_nerf_samples;        /* 48 | 64 | 192 ...; 1..1024    */
    int32_t training;                /* eval == 0                     */
    float pdf_anneal;                /* ProposalNetworkSampler._anneal (1.0 at inference) */
    /* eval only, 0 = off (exact): a wave of 64 rays stops marching once EVERY ray's transmittance is below this value
     * (wave-wide vote).  Skipped samples carry weights < eps each; rgb/thermal/accumulation move by <= eps, the
     * expected depth by <= eps * far.  The reference has no such switch: 0 reproduces it. */
    float early_stop_transmittance;
    /* 0 = choose by call size (tn_render_kernel_form: lane = ray — a wavefront marches 64 consecutive rays — from 24 576 rays in
     * the proposal pass and from 8 192 in the field pass, whose tiles a small call marches in segments (sample_split); one ray
     * per wave below), 1 = lane = ray, 2 = one ray per wave. */
    int32_t kernel_family;
    /* ProposalNetworkSampler's initial sampler [REF thermal_nerf_model.py:164-170]: 0 = UniformLinDispPiecewiseSampler
     * ("piecewise", the default), 1 = UniformSampler ("uniform").  Every level's spacing -> euclidean map follows it. */
    int32_t initial_sampler;
    /* Sample-split tiles of the exact-fp32 lane = ray field kernel (eval; round 5 — APPENDED: zero-initialise the struct, 0 keeps
     * every older caller's meaning of "the library decides").  A 64-ray tile marches its S samples serially, so a call of T tiles
     * lasts ceil(T / 2048) marches however few tiles there are; with k segments per tile it lasts ceil(k T / 2048) marches of S / k
     * samples, each segment composited with its own transmittance and chained by a second pass (w_i = T_j w_i_local: the
     * reference's weights [NS RaySamples.get_weights, called at REF thermal_nerf_model.py:233] in another association — same
     * tolerance, other last bits than the serial march).  0 = by call size (tn_render_sample_split; 1 from ~400 k rays up),
     * 1 = never, k > 1 = k segments (capped so that the records fit the workspace: tn_render_sample_split reports the value used).
     * Ignored (1) by the training, early-termination, split-precision and one-ray-per-wave kernels. */
    int32_t sample_split;
    /* Training only (round 5, appended): 0 = NS single_jitter=True (the reference's default, REF thermal_nerf_model.py:176): ONE
     * stratified draw per ray and level, tn_render_inputs.jitter = [3,R]; 1 = single_jitter=False: one draw per bin edge,
     * jitter = [R,P0+1] | [R,P1+1] | [R,S+1] back to back [NS SpacedSampler / PDFSampler.generate_ray_samples]. */
    int32_t per_sample_jitter;
} tn_render_config;

typedef struct tn_render_inputs {
    const float *origins;            /* [R,3] */
    const float *directions;         /* [R,3] */
    const float *nears;              /* [R]   (collider output) */
    const float *fars;               /* [R]   */
    const int32_t *camera_indices;   /* [R] or NULL (eval) */
    const float *lin_bins0;          /* [P0+1] torch.linspace(0,1,P0+1) */
    const float *u1;                 /* [P1+1] PDFSampler eval positions for level 1 */
    const float *u2;                 /* [S+1]  PDFSampler eval positions for the final level */
    const float *jitter;             /* training: [3,R] per-level single-jitter draws, or the per-edge draws of
                                      * tn_render_config.per_sample_jitter; NULL in eval */
} tn_render_inputs;

typedef struct tn_render_outputs {
    float *rgb;            /* [R,3] */
    float *accumulation;   /* [R]   */
    float *depth;          /* [R]   median */
    float *expected_depth; /* [R]   */
    float *prop_depth_0;   /* [R]   */
    float *prop_depth_1;   /* [R]   */
    float *thermal;        /* [R]   */
    /* optional (training / debugging): per-level weights and spacing/euclidean bins; NULL to skip */
    float *weights[3];       /* [R,P0], [R,P1], [R,S]       */
    float *spacing_bins[3];  /* [R,P0+1], [R,P1+1], [R,S+1] */
    float *eucl_bins[3];     /* same shapes                  */
} tn_render_outputs;

size_t tn_render_workspace_bytes(const tn_render_config *cfg, int64_t num_rays);

int tn_render_rays_fwd(const tn_density_field *prop0, const tn_density_field *prop1, const tn_thermal_field *field,
                       const tn_render_config *cfg, const tn_render_inputs *in, const tn_render_outputs *out,
                       int64_t num_rays, void *workspace, size_t workspace_bytes, void *stream);

/* The two halves of tn_render_rays_fwd, callable separately (tn_render_rays_fwd == both, in this order, sharing
 * `workspace`): NS ProposalNetworkSampler.generate_ray_samples [REF thermal_nerf_model.py:222-224] (+ prop_depth_i
 * [REF :267-270]) leaves the final S+1 bin edges in `workspace`; tn_field_render_fwd evaluates
 * ThermalNerfactoTField.forward on them and composites [REF :225-243,271-273].
 * PRECONDITION of tn_field_render_fwd: `workspace` was produced by tn_proposal_sample_fwd of the SAME call shape on the same
 * stream.  Besides the bin edges, that call resets the two words the expected-depth clip accumulates its call-global
 * [min, max] in (DepthRenderer "expected" clips to them): on a workspace that did not come from it, expected_depth is clipped
 * to stale bounds.  Every other output is independent of them. */
int tn_proposal_sample_fwd(const tn_density_field *prop0, const tn_density_field *prop1, const tn_render_config *cfg,
                           const tn_render_inputs *in, const tn_render_outputs *out, int64_t num_rays, void *workspace,
                           size_t workspace_bytes, void *stream);
int tn_field_render_fwd(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                        const tn_render_outputs *out, int64_t num_rays, void *workspace, size_t workspace_bytes,
                        void *stream);

/* One launch for SEVERAL of the reference's chunks, or for parts of them.  The one cross-ray quantity of the path is
 * DepthRenderer("expected")'s clip to the [min, max] of the sample mid-points of its call = of one eval_num_rays_per_chunk chunk of
 * the row-major frame [REF render/renderer.py:182-187 -> NS Model.get_outputs_for_camera_ray_bundle; REF config_thermal_nerf.py:30].
 * tn_field_render_chunked_fwd is tn_field_render_fwd for rays [first_ray, first_ray + num_rays) of such a frame: it keeps one
 * [min, max] pair per chunk the call touches (chunk = (first_ray + ray) / chunk_rays) and writes them to depth_bounds
 * [tn_depth_bound_slots(...), 2] (device floats; slot 0 = the call's first chunk).
 *   clip != 0: every ray's expected depth is clipped to its chunk's pair — a call that covers its chunks wholly (a whole frame in
 *              one launch pair) then equals the reference's chunk-by-chunk loop bit for bit at the throughput of one launch;
 *   clip == 0: expected_depth is left unclipped — the caller (ray sharding finer than the chunks, thermo_nerf_amd/distributed.py)
 *              reduces the pairs with the other parts' (min / max over the ranks sharing a chunk) and applies
 *              tn_expected_depth_clip_chunked with the reduced array.
 * first_ray and chunk_rays must be multiples of 64 (a 64-ray wavefront tile never straddles two chunks): TN_ERR_UNSUPPORTED
 * otherwise.  Same workspace precondition as tn_field_render_fwd. */
int64_t tn_depth_bound_slots(int64_t first_ray, int64_t num_rays, int64_t chunk_rays);

/* Which kernel form a call of num_rays rays takes under cfg->kernel_family (0 = by call size): 1 = lane = ray (a wavefront marches
 * 64 consecutive rays), 2 = one ray per wavefront.  pass 0: tn_proposal_sample_fwd (lane = ray from 24 576 rays; training calls: 65 536); pass 1:
 * tn_field_render_fwd / tn_field_render_chunked_fwd on `field`: lane = ray from 8 192 rays where the exact-fp32 kernel can march
 * its tiles in segments (sample_split != 1, no early termination, eval), from 40 960 with a split-precision blob (those kernels only
 * exist in form 1; a form-2 call with such a blob runs the exact-fp32 ray-per-wave kernel), from 57 344 otherwise or with field ==
 * NULL.  The two forms sum in different orders, so a caller that renders PART of a launch (ray shards of one frame,
 * thermo_nerf_amd/engine.py::render_shard) and wants the whole launch's bits asks for the whole launch's form here and passes it
 * as cfg->kernel_family.  No reference counterpart (the reference has one code path, REF render/renderer.py:182-187); 0 on a
 * NULL cfg. */
int32_t tn_render_kernel_form(const tn_thermal_field *field, const tn_render_config *cfg, int64_t num_rays, int32_t pass);
/* The number of sample segments per tile tn_field_render_fwd / _chunked_fwd use for a call of num_rays rays under
 * cfg->sample_split (see there).  Like the kernel form it is a property of the CALL: a caller that renders part of a launch and
 * wants the whole launch's bits passes the whole launch's value as cfg->sample_split. */
int32_t tn_render_sample_split(const tn_thermal_field *field, const tn_render_config *cfg, int64_t num_rays);
int tn_field_render_chunked_fwd(const tn_thermal_field *field, const tn_render_config *cfg, const tn_render_inputs *in,
                                const tn_render_outputs *out, int64_t num_rays, void *workspace, size_t workspace_bytes,
                                int64_t first_ray, int64_t chunk_rays, float *depth_bounds, int32_t clip, void *stream);
int tn_expected_depth_clip_chunked(float *expected_depth, int64_t num_rays, int64_t first_ray, int64_t chunk_rays,
                                   const float *depth_bounds, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Weight preparation (layout-only transforms, run once per weight update)
 * ---------------------------------------------------------------------------------------------------- */

/* dense re-layout of the coarse hash levels; fills grid_out->dense* fields (grid_out may alias grid_in). */
size_t tn_hashgrid_prepare_bytes(const tn_hashgrid *grid, int64_t max_bytes);
int tn_hashgrid_prepare(const tn_hashgrid *grid_in, tn_hashgrid *grid_out, void *dense_dev, size_t dense_bytes,
                        void *stream);

/* MFMA-fragment-ordered copy of the main field's MLP weights. */
size_t tn_field_prepare_bytes(const tn_thermal_field *field);
int tn_field_prepare(const tn_thermal_field *field, void *prepared_dev, size_t bytes, void *stream);

/* split-precision variants of the above (see tn_thermal_field.prepared_f16x3 / prepared_bf16x6). */
size_t tn_field_prepare_f16x3_bytes(const tn_thermal_field *field);
int tn_field_prepare_f16x3(const tn_thermal_field *field, void *prepared_dev, size_t bytes, void *stream);
size_t tn_field_prepare_bf16x6_bytes(const tn_thermal_field *field);
int tn_field_prepare_bf16x6(const tn_thermal_field *field, void *prepared_dev, size_t bytes, void *stream);

/* The operand split of mlp_precision = "bf16x6" on its own (test surface; no reference counterpart): per element i the three bf16
 * pieces of a[i] and b[i] — p1 = bf16(x), p2 = bf16(x - p1), p3 = bf16(x - p1 - p2), round to nearest even — written as floats to
 * pieces_a / pieces_b [n,3], and out[i] = the six piece products of order <= 2 accumulated in fp32 in the field kernel's order.
 * x = p1 + p2 + p3 exactly while p3 is representable (|x| >= 2^-110: its exponent is >= 2^-133, bf16's smallest sub-normal). */
int tn_bf16x6_split_product(const float *a, const float *b, int64_t n, float *pieces_a, float *pieces_b, float *out, void *stream);
/* One v_mfma_f32_32x32x16_bf16 with constant operands bf16(a_value), bf16(b_value): out[0] = 16 a b (fp32) — what the matrix core
 * does with sub-normal bf16 inputs (test surface). */
int tn_bf16_mfma_value_probe(float a_value, float b_value, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------
 * Training step: forward with a tape of per-sample activations, losses, backward.
 * The reference gets all of this from torch autograd over nerfstudio's modules
 * [REF thermal_nerf_model.py:210-326; nerfstudio model_components/losses.py]; here each differentiable stage
 * is one entry point on [N, width] fp32 row-major matrices (N = rays * samples of the level), the binding chains
 * them inside a torch.autograd.Function.  Gradient outputs marked (+=) are accumulated with atomics: zero them first.
 * ---------------------------------------------------------------------------------------------------- */

#define TN_ACT_NONE 0
#define TN_ACT_RELU 1
#define TN_ACT_SIGMOID 2

/* NS get_density position normalisation + HashEncoding.pytorch_fwd: positions [N,3] (world) ->
 * enc [N, 2*num_levels], selector [N] (0/1; the encoding is evaluated at p * selector like the reference). */
int tn_hash_encode_fwd(const tn_hashgrid *grid, const tn_space *space, const float *positions, int64_t n, float *enc,
                       float *selector, void *stream);
/* its backward w.r.t. the table: d_enc [N, 2*num_levels] -> d_table [num_levels << log2_hashmap_size, 2] (+=). */
int tn_hash_encode_bwd(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                       int64_t n, float *d_table, void *stream);

/* tn_hash_encode_bwd restricted to the levels [level_begin, level_end) (d_enc keeps all num_levels columns). */
int tn_hash_encode_bwd_levels(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, int32_t level_end, void *stream);
/* tn_hash_encode_bwd_levels with the COARSE levels (dense vertex grid of at most 48^3: scalings up to 46, the first four
 * levels of the reference grids) accumulated in 16 private dense copies of their vertex grids and summed into d_table by
 * a second launch: at those levels the samples of a trained scene hit the same few thousand entries, and same-address atomics
 * retire one at a time in the memory-side atomic unit.  Same sums (other order).  The workspace is cleared by the call;
 * levels outside the coarse set, or a call that does not start at level 0, run exactly as tn_hash_encode_bwd_levels. */
size_t tn_hash_encode_bwd_spread_workspace_bytes(const tn_hashgrid *grid);
int tn_hash_encode_bwd_spread(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, int32_t level_end, void *workspace,
                              size_t workspace_bytes, void *stream);

/* The same adjoint without global atomics for the levels [level_begin, num_levels): contributions are written out as records
 * bucketed by the table slice (2^14 entries) that owns them and summed per slice in LDS, then added to d_table (+=) with plain
 * loads / stores.  Needs tn_hash_encode_bwd_sorted_workspace_bytes(grid, n, level_begin) bytes of 16-byte aligned device
 * scratch (a fixed region per (level, slice) bin sized 1.25 x an even spread: ~25 B per (sample, level, corner pair); a
 * record that does not fit its bin's region is added with global atomics instead); that function returns 0 — and tn_hash_encode_bwd_sorted
 * TN_ERR_UNSUPPORTED — for a geometry the bucketing does not cover (finest scaling + 2 >= 2^14 with more than one slice, or
 * >= 2^32 records).  tn_hash_encode_bwd_sorted_first_level: the level from which this form is the faster one on this part
 * (levels with a scaling >= 200, at least 128 (level, slice) bins), or -1: the caller runs tn_hash_encode_bwd_levels on the
 * levels below it and this on the rest. */
size_t tn_hash_encode_bwd_sorted_workspace_bytes(const tn_hashgrid *grid, int64_t n, int32_t level_begin);
int tn_hash_encode_bwd_sorted_first_level(const tn_hashgrid *grid, int64_t n);
int tn_hash_encode_bwd_sorted(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                              int64_t n, float *d_table, int32_t level_begin, void *workspace, size_t workspace_bytes,
                              void *stream);

/* and w.r.t. the WORLD positions (camera-pose optimisation, NS CameraOptimizer applied at REF thermal_nerf_model.py:
 * 218-219): through the trilinear offsets, `p * selector`, (x + 2) / 4 and the L-inf contraction (or the AABB
 * normalisation): d_enc [N, 2*num_levels] -> d_positions [N,3] (=). */
int tn_hash_encode_bwd_input(const tn_hashgrid *grid, const tn_space *space, const float *positions, const float *d_enc,
                             int64_t n, float *d_positions, void *stream);
/* backward of tn_frustum_positions: d_positions [R,n,3] -> d_origins [R,3] (+=), d_directions [R,3] (+=). */
int tn_frustum_positions_bwd(const float *d_positions, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                             float *d_origins, float *d_directions, void *stream);

/* torch.nn.Linear (+ activation): y[n, :out] = act(x[n, :in] W^T + b); x rows are ldx floats apart, y rows ldy.
 * in_dim, out_dim <= 256 (TN_ERR_SHAPE beyond).  Layers up to 64 wide — the reference's configs — keep a row's inputs in
 * registers; wider ones (config.hidden_dim / hidden_dim_color / hidden_dim_transient [REF thermal_nerf_model.py:96-114] above 64)
 * stage 64-row tiles of x in LDS, and their backward runs the 64-wide kernel once per 64 x 64 block of the weight matrix (it
 * needs the workspace: TN_ERR_WORKSPACE without one when weight gradients are asked for). */
int tn_linear_fwd(const float *x, int32_t ldx, const tn_linear *lin, int32_t act, int64_t n, float *y, int32_t ldy,
                  void *stream);
/* backward: g = dy * act'(y) (y = the forward OUTPUT, rows ldy apart like dy);  dx[n,:in] (= or += when
 * accumulate_dx) = g W;  d_weight [out,in] (+=) = g^T x;  d_bias [out] (+=) = sum_n g.  dx / d_weight / d_bias
 * may be NULL.  `workspace` (tn_linear_bwd_workspace_bytes(), reusable across calls on one stream) holds per-block
 * partial weight gradients summed by a second kernel; NULL falls back to one atomic per block and weight entry. */
size_t tn_linear_bwd_workspace_bytes(void);
int tn_linear_bwd(const float *x, int32_t ldx, const float *y, const float *dy, int32_t ldy, const tn_linear *lin,
                  int32_t act, int64_t n, float *dx, int32_t lddx, int32_t accumulate_dx, float *d_weight,
                  float *d_bias, void *workspace, size_t workspace_bytes, void *stream);

/* density = average_init_density * trunc_exp(raw) * selector (NS get_density); raw rows ld_raw floats apart.
 * backward: d_raw = d_density * average_init_density * exp(clamp(raw, trunc_exp_min, 15)) * selector (NS trunc_exp.backward:
 * trunc_exp_min = -15 is nerfstudio's / torch-ngp's two-sided clamp, -INFINITY clamps from above only), written to
 * column 0 of rows ld_d_raw apart; columns 1 .. clear_cols-1 of each row are set to zero (clear_cols <= 1: untouched) so that
 * the row can be the += target of the stages behind it without a separate fill. */
int tn_density_act_fwd(const float *raw, int32_t ld_raw, const float *selector, float average_init_density, int64_t n,
                       float *density, void *stream);
int tn_density_act_bwd(const float *raw, int32_t ld_raw, const float *selector, float average_init_density,
                       float trunc_exp_min, const float *d_density, int64_t n, float *d_raw, int32_t ld_d_raw,
                       int32_t clear_cols, void *stream);

/* backward of tn_weights_fwd: d_weights [R,n] -> d_densities [R,n]. */
int tn_weights_bwd(const float *deltas, const float *densities, const float *d_weights, int64_t num_rays, int32_t n,
                   float *d_densities, void *stream);

/* Backward of up to three CONSECUTIVE Linear(+activation) layers in one launch (an MLP's chain: mlp_head, mlp_thermal + head,
 * mlp_base).  layers[0] is the layer nearest the loss; layers[j].x = that layer's input rows (the ACTIVATED output of the
 * layer below, produced by activation act_x), so layers[j].lin.out_dim == layers[j-1].lin.in_dim.  dy [n, out_0] (rows lddy
 * apart) is the gradient w.r.t. layer 0's ACTIVATED output y_top (needed when act_top != none).  d_weight / d_bias (+=) per
 * layer, may be NULL; dx [n, in_last] (= or += when accumulate_dx) may be NULL.  Same arithmetic as num_layers calls of
 * tn_linear_bwd, with one [n, width] read per layer instead of three reads and a write. */
typedef struct tn_chain_layer {
    tn_linear lin;
    const float *x;
    int32_t ldx;
    int32_t act_x;   /* TN_ACT_* of the layer that produced x (unused for the last layer) */
    float *d_weight;
    float *d_bias;
} tn_chain_layer;
size_t tn_linear_chain_bwd_workspace_bytes(void);
int tn_linear_chain_bwd(const tn_chain_layer *layers, int32_t num_layers, const float *y_top, int32_t act_top, const float *dy,
                        int32_t lddy, int64_t n, float *dx, int32_t lddx, int32_t accumulate_dx, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The final level's field forward of a training step in ONE launch [REF thermal_field.py:183-201 in train mode]: what the
 * chain tn_hash_encode_fwd -> tn_linear_fwd (mlp_base x2) -> tn_density_act_fwd -> tn_color_input_fwd -> tn_linear_fwd
 * (mlp_head x3, mlp_thermal x2, head) computes, with every activation the backward differentiates written to its tape
 * buffer in the same row-major layout: positions [N,3] (N = num_rays * n, ray-major), directions [R,3], camera_indices [R];
 * enc [N,32], selector [N], h1 [N,64] (ReLU applied), bo [N,16] (raw density | geo), density [N], c1, c2 [N,64] (ReLU),
 * rgb [N,3] (sigmoid), t1 [N,64] (ReLU), t2 [N,64] (sigmoid), thermal [N].  Needs field->prepared (tn_field_prepare on the
 * CURRENT weights); the reference geometry only (16 levels, geo 15, appearance 32). */
int tn_field_fwd_taped(const tn_thermal_field *field, const float *positions, const float *directions,
                       const int32_t *camera_indices, int64_t num_rays, int32_t n, float *enc, float *selector, float *h1,
                       float *bo, float *density, float *c1, float *c2, float *rgb, float *t1, float *t2, float *thermal,
                       void *stream);

/* The tape-free pair of the final level's training pass [REF thermal_field.py:108-201 differentiated; the modules
 * constructed at thermal_field.py:62-102].  tn_field_fwd_train is tn_field_fwd_taped keeping only what compositing, the
 * losses and the backward read: selector [N], density [N], rgb [N,3], thermal [N] and the hash features `enc`, the latter
 * in tiles of 64 consecutive samples, [ceil(N/64)][16 levels][64 samples][2] floats (the layout tn_field_bwd_fused reads;
 * allocate 32 * 64 * ceil(N/64) floats).  `ray_bias` [R,64] = mlp_head.0's
 * bias plus its SH(direction) and appearance-embedding columns applied to each ray's constants (tn_ray_head_fwd): the colour
 * layer sees them as a per-ray bias.  `base_out` (round 5; NULL to skip): mlp_base's 16 output rows [N,16] (raw density | geo
 * features) as well — 64 B per sample more — which tn_field_bwd_fused's split form then reads in its two head launches
 * instead of recomputing mlp_base there (pass the same pointer, or NULL to recompute).  `position_jacobian` (round 5; NULL to
 * skip; 96 * 64 * ceil(N/64) floats): d hash features / d normalised position, [ceil(N/64)][16 levels][3 axes][64 samples][2] —
 * the corner values are in registers in the forward; handed to tn_field_bwd_fused, its d_positions needs no table read.
 *
 * tn_field_bwd_fused recomputes the five hidden layers from `enc` in registers and runs their adjoints next to them (no
 * [N,64] activation ever touches HBM).  Inputs: enc / selector / rgb of the forward, ray_bias, the per-sample output
 * gradients d_rgb [N,3], d_thermal [N] (either may be NULL: that branch is skipped), d_density [N].  Outputs: d_enc [N,32]
 * (=), d_ray_sum [R,64] (+=, zero it first) = per-ray sums of mlp_head.0's pre-activation gradient — tn_ray_head_bwd then
 * yields mlp_head.0's bias gradient, its SH and appearance weight columns and the embedding / direction gradients — and the
 * gradients of every other Linear of the field (+=; NULL entries skipped; head0_w receives its geo columns 16..30 only).
 * d_positions [N,3] (=; NULL to skip; needs `positions` [N,3]): d loss / d sample position through the hash encoding — what
 * tn_hash_encode_bwd_input computes from d_enc — for camera-pose optimisation, produced by the mlp_base launch.
 * trunc_exp_min: lower clamp of trunc_exp's backward (g * exp(clamp(x, min, 15)); -15 = torch-ngp / nerfstudio's
 * two-sided clamp, -INFINITY = upper clamp only).  pass_thermal_gradients = 0 keeps the thermal branch from the geo
 * features [REF thermal_field.py:171-172].  split = 0: one launch for the whole field (one wave per SIMD: its ~210 gradient
 * accumulators); 1: the colour head and (thermal head + mlp_base) as two launches of two waves per SIMD each, the colour
 * head's adjoint of mlp_base's outputs passing through the workspace; 2 (round 5): the split form with the products whose K
 * is a multiple of 32 features — the 64 x 64 ones of the two head launches (second layer: recomputed forward and dx; with
 * base_out only) and mlp_base.0's forward and dx in the mlp_base launch — on v_mfma_f32_16x16x32_bf16 as six products of three
 * exact bf16 pieces per operand: fp32's rounding size per product (see tn_bf16x6_split_product).  The parameter gradients of
 * all launches leave through ONE slab reduction at the end.  The reference geometry only (16 levels, geo 15, appearance 32). */
/* One proposal level of a step on which the proposal networks take gradient, forward and backward in one launch each
 * [HashMLPDensityField built at REF thermal_nerf_model.py:127-149: 5 levels -> Linear(10,16)+ReLU -> Linear(16,1) -> trunc_exp]:
 * forward writes what tn_hash_encode_fwd + 2 x tn_linear_fwd + tn_density_act_fwd write (enc [n,10], selector, raw, density
 * [n]); backward takes d_density [n] to d_enc [n,10] (=) and the four parameter gradients (+=), the hidden layer recomputed
 * from enc.  The reference geometry only (5 levels, hidden 16): TN_ERR_UNSUPPORTED otherwise (use the stage entry points). */
int tn_density_fwd_train(const tn_density_field *f, const float *positions, int64_t n, float *enc, float *selector, float *raw,
                         float *density, void *stream);
int tn_density_bwd_train(const tn_density_field *f, const float *enc, const float *raw, const float *selector,
                         const float *d_density, int64_t n, float trunc_exp_min, float *d_enc, float *d_w0, float *d_b0,
                         float *d_w1, float *d_b1, void *stream);

/* mlp_head.0's ray-constant part [REF thermal_field.py:117-126,160-168]: ray_bias [R,64] = bias + W[:, 0:16] . SH(direction) +
 * W[:, 31:63] . embedding[camera] (training-mode appearance), and its adjoint on the per-ray sums d_ray_sum [R,64] of the layer's
 * pre-activation gradient: d_head0_weight (+=, the SH and appearance columns only), d_head0_bias (+=), d_appearance
 * [num_images, 32] (+=), and optionally d_ray_inputs [R,64] (=; columns 0..15 = gradient w.r.t. the SH basis values, which
 * tn_color_input_bwd with n = 1 carries on to the directions). */
int tn_ray_head_fwd(const tn_thermal_field *field, const float *directions, const int32_t *camera_indices, int64_t num_rays,
                    float *ray_bias, void *stream);
int tn_ray_head_bwd(const tn_thermal_field *field, const float *directions, const int32_t *camera_indices, int64_t num_rays,
                    const float *d_ray_sum, float *d_head0_weight, float *d_head0_bias, float *d_appearance, float *d_ray_inputs,
                    void *stream);
typedef struct tn_field_grads {
    float *base0_w, *base0_b, *base1_w, *base1_b;
    float *head0_w, *head1_w, *head1_b, *head2_w, *head2_b;
    float *th0_w, *th0_b, *th1_w, *th1_b, *thead_w, *thead_b;
} tn_field_grads;
int tn_field_fwd_train(const tn_thermal_field *field, const float *positions, const float *ray_bias, int64_t num_rays,
                       int32_t n, float *enc, float *selector, float *density, float *rgb, float *thermal, float *base_out,
                       float *position_jacobian, void *stream);
size_t tn_field_bwd_fused_workspace_bytes(int64_t num_rays, int32_t n);
int tn_field_bwd_fused(const tn_thermal_field *field, int64_t num_rays, int32_t n, const float *enc, const float *selector,
                       const float *base_out, const float *ray_bias, const float *rgb, const float *d_rgb, const float *d_thermal,
                       const float *d_density, int32_t pass_thermal_gradients, float trunc_exp_min, int32_t split,
                       float *d_enc, float *d_ray_sum, const float *positions, const float *position_jacobian,
                       float *d_positions, const tn_field_grads *grads, void *workspace, size_t workspace_bytes, void *stream);

/* NS scale_gradients_by_distance_squared [REF thermal_nerf_model.py:228-231, use_gradient_scaling]: the forward is the
 * identity; in the backward the gradient of EVERY field output of a sample (density [n], rgb [n,3], thermal [n]; any may
 * be NULL) is multiplied in place by clamp(((start + end) / 2)^2, 0, 1).  starts/ends [n]. */
int tn_gradient_scale_bwd(const float *starts, const float *ends, int64_t n, float *d_density, float *d_rgb,
                          float *d_thermal, void *stream);

/* NS CameraOptimizer(mode="SO3xR3").apply_to_raybundle — the first statement of the training forward
 * [REF thermal_nerf_model.py:218-219]: with M_c = exp_map_SO3xR3(pose_adjustment[c]) ([t | w] -> [R(w) | t], Rodrigues with
 * theta = sqrt(max(|w|^2, 1e-4))), out_origins = origins + t_c, out_directions = R(w_c) directions, c = camera_indices[ray]
 * (int64, as nerfstudio's ray bundles carry them; 0 <= c < num_cameras is the caller's contract).  pose_adjustment
 * [num_cameras, 6], origins / directions / outputs [R,3]; outputs may alias the inputs. */
int tn_camera_opt_fwd(const float *pose_adjustment, const int64_t *camera_indices, const float *origins,
                      const float *directions, int64_t num_rays, int32_t num_cameras, float *out_origins,
                      float *out_directions, void *stream);

/* its backward: d_out_origins / d_out_directions [R,3] (either may be NULL = zero), directions = the INPUT directions of the
 * forward -> d_pose_adjustment [num_cameras, 6] (+=, caller clears it; torch.clamp's sub-gradient: theta is a constant where
 * |w|^2 < 1e-4), optional d_directions [R,3] (=) = R^T d_out_directions.  d origins = d_out_origins (identity, not written). */
int tn_camera_opt_bwd(const float *pose_adjustment, const int64_t *camera_indices, const float *directions,
                      const float *d_out_origins, const float *d_out_directions, int64_t num_rays, int32_t num_cameras,
                      float *d_pose_adjustment, float *d_directions, void *stream);

/* backward of tn_composite_fwd in training mode (no nan_to_num / clamp): d_out [R,C], accumulation [R] ->
 * d_values [R,n,C] (=), d_weights [R,n] (+=). */
int tn_composite_bwd(const float *values, const float *weights, const float *accumulation, const float *d_out,
                     int64_t num_rays, int32_t n, int32_t channels, float *d_values, float *d_weights, void *stream);

/* input of mlp_head [REF thermal_field.py:160-167]: cin [R*n, 64] = [SH16(dir) | geo | appearance | 0];
 * geo rows ld_geo floats apart; training != 0: embedding[camera_indices[r]], else mean / zeros [REF :124-137]. */
int tn_color_input_fwd(const tn_thermal_field *field, const float *directions, const float *geo, int32_t ld_geo,
                       const int32_t *camera_indices, int32_t training, int64_t num_rays, int32_t n, float *cin,
                       void *stream);
/* backward: d_cin [R*n,64] -> d_geo (+=, rows ld_d_geo apart), d_appearance [num_images, app_dim] (+=; training),
 * d_directions [R,3] (+=; through the SH basis; NULL to skip, needs `directions`). */
int tn_color_input_bwd(const tn_thermal_field *field, const float *d_cin, const int32_t *camera_indices,
                       int32_t training, int64_t num_rays, int32_t n, float *d_geo, int32_t ld_d_geo,
                       float *d_appearance, const float *directions, float *d_directions, void *stream);

/* The final level's per-ray renderers of a training step in one launch, and their adjoints in one launch: weights =
 * RaySamples.get_weights(density) [REF thermal_nerf_model.py:233] [R,n], rgb [R,3] / thermal [R] = "last_sample" compositing
 * [REF :237, :271-273; thermal_renderer.py:55-79], accumulation [R] (what tn_weights_fwd + 2 x tn_composite_fwd compute).
 * Backward: d_rgb [R,3], d_thermal [R], d_accumulation [R], d_weights [R,n] (the regularisers' gradient) — each may be NULL
 * — to d_rgb_samples [R n,3], d_thermal_samples [R n], d_densities [R n] (=); starts / ends [R n] non-NULL apply
 * scale_gradients_by_distance_squared [REF :228-231] to the three.  n <= 1024. */
int tn_ray_render_fwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                      int64_t num_rays, int32_t n, float *weights, float *rgb, float *thermal, float *accumulation, void *stream);
/* tn_ray_render_fwd + tn_depth_fwd(median, expected) of the same level in one pass [REF thermal_nerf_model.py:233-243]: the weights
 * are in registers when the depth renderers need them.  Bit-equal outputs to the two calls; the expected depth's call-global clip
 * [NS DepthRenderer("expected")] reduces per-block bounds instead of two atomically updated words: bounds_scratch holds
 * 2 * ceil(num_rays / 4) floats and needs no initialisation. */
int tn_ray_render_depth_fwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                            const float *starts, const float *ends, int64_t num_rays, int32_t n, float *weights, float *rgb,
                            float *thermal, float *accumulation, float *median, float *expected, float *bounds_scratch, void *stream);
int tn_ray_render_bwd(const float *deltas, const float *densities, const float *rgb_samples, const float *thermal_samples,
                      const float *accumulation, const float *d_rgb, const float *d_thermal, const float *d_accumulation,
                      const float *d_weights, const float *starts, const float *ends, int64_t num_rays, int32_t n,
                      float *d_rgb_samples, float *d_thermal_samples, float *d_densities, void *stream);

/* The two image losses of get_loss_dict [REF thermal_nerf_model.py:294-295, 319-323: MSELoss means] and the PSNR of NS
 * get_metrics_dict in one launch: rgb / gt_rgb [R,3], thermal / gt_thermal [R] (thermal may be NULL) -> out[0] = rgb MSE,
 * out[1] = thermal MSE, out[2] = 10 log10(1 / out[0]); d_rgb [R,3] (=) and d_thermal [R] (=) = the gradients of the two means. */
int tn_image_losses(const float *rgb, const float *gt_rgb, const float *thermal, const float *gt_thermal, int64_t num_rays,
                    float *out, float *d_rgb, float *d_thermal, void *stream);

/* The step's total loss as nerfstudio's Trainer forms it — functools.reduce(torch.add, loss_dict.values()) [NS Trainer.train_iteration,
 * driven by REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:17-30] — in one launch: out[0] = ((t0 + t1) + t2) + ...;
 * `terms` is a HOST array of `count` <= TN_SUM_MAX_TERMS device pointers to one float each. */
#define TN_SUM_MAX_TERMS 16
int tn_sum_scalars(const float *const *terms, int32_t count, float *out, void *stream);

/* NS losses.distortion_loss on one level: spacing bins [R,n+1], weights [R,n] -> loss_sum[0] (+=) = scale * sum over rays
 * of lossfun_distortion (scale = 1/R for nerfstudio's mean), d_weights [R,n] (=) = scale * d(sum)/dw.  O(n) per ray. */
int tn_distortion_loss(const float *spacing_bins, const float *weights, int64_t num_rays, int32_t n, float scale,
                       float *loss_sum, float *d_weights, void *stream);
/* the metric and the loss term NerfactoModel makes of it (metrics_dict["distortion"], then distortion_loss_mult * that
 * [REF thermal_nerf_model.py:301-304]) from the same launch: loss_pair[0] (+=) = scale * sum, loss_pair[1] (+=) = mult * scale *
 * sum, d_weights [R,n] (=) = the gradient of loss_pair[1]. */
int tn_distortion_loss_term(const float *spacing_bins, const float *weights, int64_t num_rays, int32_t n, float scale,
                            float mult, float *loss_pair, float *d_weights, void *stream);
/* NS losses.interlevel_loss, one proposal level: final bins c [R,n+1] / weights w [R,n] (constants), proposal bins
 * cp [R,p+1] / weights wp [R,p] -> loss_sum[0] (+=) = scale * sum over rays and samples of lossfun_outer (scale = 1/(R*n)),
 * d_wp [R,p] (=) = scale * d(sum)/dwp.  p <= 1024.  Several levels may add into the same loss_sum. */
int tn_interlevel_loss(const float *c, const float *w, const float *cp, const float *wp, int64_t num_rays, int32_t n,
                       int32_t p, float scale, float *loss_sum, float *d_wp, void *stream);
/* every proposal level of NS interlevel_loss in one launch (the levels are independent): cp / wp / d_wp are HOST arrays of
 * num_levels (<= 4) device pointers, p the levels' sample counts; same sums as num_levels calls of tn_interlevel_loss. */
int tn_interlevel_loss_levels(const float *c, const float *w, int64_t num_rays, int32_t n, int32_t num_levels,
                              const float *const *cp, const float *const *wp, const int32_t *p, float scale, float *loss_sum,
                              float *const *d_wp, void *stream);

/* ---- the training step's launch chains as two calls -------------------------------------------------------------------------
 * What ThermalNerfModel.get_outputs [REF thermal_nerf_model.py:210-275] queues in training mode on a step whose proposal networks
 * take no gradient (nerfstudio's ProposalNetworkSampler: 5 steps of 6 after warm-up), and the adjoint chain autograd runs for it —
 * the SAME entry points as above, in the same order and on the same streams as the per-call host path queues them (bit-identical
 * results), issued from C++: a step is ~20 launches whose Python cost (one ctypes call and ~2 output allocations each) decided the
 * step at the reference's default S = 48.  Every buffer is the caller's; the cross-stream order is made with events
 * created and destroyed inside the call (no state is kept).
 *
 * tn_train_step_fwd:  tn_field_prepare(field_raw -> field->prepared) | tn_proposal_sample_fwd (train-mode, untaped) |
 *   tn_frustum_from_edges | tn_ray_head_fwd | [wait for `wait_events`: a deferred table update of the previous step] |
 *   tn_field_fwd_train | tn_ray_render_depth_fwd | the two regularisers on `second` / `third` behind the level's weights (when their
 *   loss pointers are set: tn_distortion_loss_term, tn_interlevel_loss_levels).
 * tn_train_step_bwd:  tn_ray_render_bwd | tn_field_bwd_fused | the table scatter — tn_hash_encode_bwd_sorted on `second` beside
 *   tn_hash_encode_bwd_spread / _levels for the levels below `first_sorted_level` on `stream` (defer == 0: `stream` waits for
 *   `second` at the end) or on `third` (defer != 0: nobody waits — the caller joins, thermo_nerf_amd/_hip.py) | the ray-level
 *   adjoints on `stream`: tn_ray_head_bwd (+ tn_color_input_bwd with sh_direction_gradient), tn_frustum_positions_bwd. */
typedef struct tn_train_step {
    const tn_density_field *prop0, *prop1;
    const tn_thermal_field *field_raw;   /* the field as stored (what tn_field_prepare reads)                          */
    const tn_thermal_field *field;       /* the same with `prepared` -> a buffer of prepared_bytes this call (re)fills  */
    size_t prepared_bytes;
    const tn_render_config *cfg;         /* training = 1; sample counts, anneal, initial sampler, jitter form          */
    const tn_render_inputs *in;          /* rays, planes, camera indices (int32), jitter, bin tables                   */
    int64_t num_rays;
    /* kept for the backward (caller-allocated) */
    float *spacing[3], *eucl[3];         /* [R, n+1] per level                                                          */
    float *weights[3];                   /* [R, n]                                                                      */
    float *prop_depth[2];                /* [R]                                                                         */
    float *positions;                    /* [R S, 3]                                                                    */
    float *starts, *ends, *deltas;       /* [R, S]                                                                      */
    float *ray_bias;                     /* [R, 64]                                                                     */
    float *enc, *selector, *density, *rgb_samples, *thermal_samples, *base_out, *jacobian; /* tn_field_fwd_train's      */
    float *rgb, *thermal, *accumulation, *depth, *expected_depth, *depth_scratch; /* [R,3] [R] [R] [R] [R] [2 ceil(R/4)] */
    void *workspace;                     /* tn_render_workspace_bytes(cfg, num_rays)                                    */
    size_t workspace_bytes;
    void *zero_buffer;                   /* optional: cleared (hipMemsetAsync) on `second` at the start of the call — the  */
    size_t zero_bytes;                   /* backward's gradient arena, off the calling stream (tn_train_step_bwd waits);   */
                                         /* `second` first waits for everything queued on `stream` (the buffer's memory may */
                                         /* have been an earlier step's gradients, still read by kernels queued there)      */
    /* regularisers (optional: loss pointers NULL = not launched); losses are += accumulators the caller zeroed         */
    float distortion_mult, interlevel_mult;
    float *distortion_loss_pair, *distortion_grad;   /* [2], [R,S]                                                      */
    float *interlevel_loss, *interlevel_grad[2];     /* [1], [R,P0], [R,P1]                                             */
    /* streams; wait_events: hipEvent_t handles `stream` waits for right before the field's first table read            */
    void *stream, *second, *third;
    void *const *wait_events;
    int32_t num_wait_events;
} tn_train_step;
int tn_train_step_fwd(const tn_train_step *s);

typedef struct tn_train_step_bwd_args {
    const tn_thermal_field *field;       /* raw struct (no prepared blob needed)                                        */
    int64_t num_rays;
    int32_t n;                           /* samples per ray of the final level                                          */
    /* the forward's tensors */
    const float *positions, *starts, *ends, *deltas, *ray_bias, *enc, *selector, *density, *rgb_samples, *thermal_samples;
    const float *base_out, *jacobian, *accumulation, *directions;
    const int32_t *camera_indices;
    /* incoming gradients (any may be NULL) */
    const float *d_rgb, *d_thermal, *d_accumulation, *d_weights;
    int32_t use_gradient_scaling, pass_thermal_gradients, split_form, sh_direction_gradient;
    float trunc_exp_min;
    /* scratch the call fills: [R S,3] [R S] [R S] [R S,32] [R S,3 or NULL] [R,64 zeroed or NULL] [R,64 or NULL]        */
    float *d_rgb_samples, *d_thermal_samples, *d_density, *d_enc, *d_positions, *d_ray_sum, *d_ray_inputs;
    const tn_field_grads *grads;         /* zero-initialised parameter gradients (+=)                                   */
    float *d_table, *d_appearance, *d_head0_bias; /* zeroed (+=); mlp_head.0's bias gradient is the ray-level adjoint's         */
    float *d_origins, *d_directions;     /* [R,3] zeroed (+=) or NULL: no ray gradients                                 */
    void *fused_workspace; size_t fused_workspace_bytes;      /* tn_field_bwd_fused_workspace_bytes                     */
    int32_t first_sorted_level;          /* tn_hash_encode_bwd_sorted_first_level; < 0: every level with atomics        */
    void *sorted_workspace; size_t sorted_workspace_bytes;
    int32_t spread;                      /* config.spread_coarse_scatter                                                */
    void *spread_workspace; size_t spread_workspace_bytes;    /* of the stream the atomic levels run on                 */
    int32_t overlap, defer;              /* config.overlap_table_scatter / deferred_table_update                        */
    int32_t wait_second_first;           /* the gradients were cleared on `second` by tn_train_step_fwd (zero_buffer)   */
    void *stream, *second, *third;
} tn_train_step_bwd_args;
int tn_train_step_bwd(const tn_train_step_bwd_args *a);

/* ---- optimizer ------------------------------------------------------------------------------------------------------------
 * torch.optim.Adam as the reference's method config sets it for every parameter group — AdamOptimizerConfig(lr 1e-2,
 * eps 1e-15), nerfstudio's Optimizers.optimizer_step_all [REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:31-44;
 * thermo_nerf/nerfacto_config/config_nerfacto.py:28-34 for camera_opt's weight decay] — over a LIST of tensors in ONE launch:
 *     g = grad + weight_decay * param ;  m += (1 - beta1) (g - m) ;  v = beta2 v + (1 - beta2) g g ;
 *     param -= step_size * m / (sqrt(v) / bias_correction2_sqrt + eps)
 * with the step-dependent scalars formed by the caller (step_size = lr / (1 - beta1^t), bias_correction2_sqrt =
 * sqrt(1 - beta2^t), t = the tensor's own step count): tensors of different groups and step counts share a launch.
 * `tensors` is a HOST array of `count` <= TN_ADAM_MAX_TENSORS descriptors (device pointers inside); param / exp_avg /
 * exp_avg_sq are updated in place, grad is read only; tensors with n = 0 are skipped. */
#define TN_ADAM_MAX_TENSORS 32
typedef struct tn_adam_tensor {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t n;
    float step_size;
    float bias_correction2_sqrt;
    float one_minus_beta1;
    float beta2;
    float one_minus_beta2;
    float eps;
    float weight_decay;
} tn_adam_tensor;
int tn_adam_step(const tn_adam_tensor *tensors, int32_t count, void *stream);

/* library identification: returns a static string "thermonerf_hip <version> gfx950". */
const char *tn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* THERMONERF_HIP_H */
