// The training step's launch chains issued from C++ (include/thermonerf_hip.h: tn_train_step_fwd / tn_train_step_bwd).
// No kernel lives here: both functions call the library's own entry points in the order — and on the streams — in which
// thermo_nerf_amd/training.py queues them one ctypes call at a time, so the two host paths give bit-identical results
// (tests/test_gpu_training.py compares them).  What changes is the host: one call and one output slab per direction instead of
// ~20 calls and ~45 allocations per step, which at the reference's default S = 48 is the difference between a device-bound and a
// host-bound step.
#include <hip/hip_runtime.h>

#include "../../include/thermonerf_hip.h"

namespace {

#define STEP_TRY(expr)              \
    do {                            \
        const int _e = (expr);      \
        if (_e != TN_OK) return _e; \
    } while (0)

// `waiter` waits for everything queued on `signal` so far (an event that lives only as long as the wait needs it)
int stream_after(hipStream_t waiter, hipStream_t signal) {
    if (waiter == signal) return TN_OK;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return TN_ERR_LAUNCH;
    const bool ok = hipEventRecord(e, signal) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess;
    (void)hipEventDestroy(e);  // (released once the wait has been satisfied)
    return ok ? TN_OK : TN_ERR_LAUNCH;
}

}  // namespace

extern "C" int tn_train_step_fwd(const tn_train_step *s) {
    if (!s || !s->prop0 || !s->prop1 || !s->field_raw || !s->field || !s->cfg || !s->in) return TN_ERR_NULL;
    if (!s->field->prepared || !s->positions || !s->starts || !s->ends || !s->deltas || !s->ray_bias) return TN_ERR_NULL;
    if (!s->cfg->training) return TN_ERR_UNSUPPORTED;
    const int64_t R = s->num_rays;
    const int S = s->cfg->num_nerf_samples;
    if (R <= 0) return R == 0 ? TN_OK : TN_ERR_SHAPE;
    hipStream_t main = (hipStream_t)s->stream, second = (hipStream_t)s->second, third = (hipStream_t)s->third;
    if (s->zero_buffer && s->zero_bytes) {
        // the backward's gradient arena (75 MB with the table's): cleared beside this forward instead of in front of the backward.
        // The buffer is the CALLING stream's allocation: its memory may have served an earlier step's gradients, which kernels
        // still queued on the calling stream (that step's backward, its optimizer) read — the fill on the second stream must not
        // overtake them.  (Found by the config-1 run: with no bucketed scatter queued on the second stream — small tables — and
        // the host two steps ahead of the device, 4 runs in 240 lost part of a step's gradients to this fill.)
        STEP_TRY(stream_after(second, main));
        if (hipMemsetAsync(s->zero_buffer, 0, s->zero_bytes, second) != hipSuccess) return TN_ERR_LAUNCH;
    }
    STEP_TRY(tn_field_prepare(s->field_raw, const_cast<float *>(s->field->prepared), s->prepared_bytes, main));
    tn_render_outputs o = {};
    o.prop_depth_0 = s->prop_depth[0];
    o.prop_depth_1 = s->prop_depth[1];
    for (int i = 0; i < 3; ++i) {
        o.spacing_bins[i] = s->spacing[i];
        o.eucl_bins[i] = s->eucl[i];
    }
    o.weights[0] = s->weights[0];
    o.weights[1] = s->weights[1];
    STEP_TRY(tn_proposal_sample_fwd(s->prop0, s->prop1, s->cfg, s->in, &o, R, s->workspace, s->workspace_bytes, main));
    STEP_TRY(tn_frustum_from_edges(s->in->origins, s->in->directions, s->eucl[2], R, S, s->positions, s->starts, s->ends, s->deltas, main));
    STEP_TRY(tn_ray_head_fwd(s->field_raw, s->in->directions, s->in->camera_indices, R, s->ray_bias, main));
    for (int k = 0; k < s->num_wait_events; ++k)  // a deferred table update of the previous step ends here
        if (hipStreamWaitEvent(main, (hipEvent_t)s->wait_events[k], 0) != hipSuccess) return TN_ERR_LAUNCH;
    STEP_TRY(tn_field_fwd_train(s->field, s->positions, s->ray_bias, R, S, s->enc, s->selector, s->density, s->rgb_samples,
                                s->thermal_samples, s->base_out, s->jacobian, main));
    STEP_TRY(tn_ray_render_depth_fwd(s->deltas, s->density, s->rgb_samples, s->thermal_samples, s->starts, s->ends, R, S, s->weights[2],
                                     s->rgb, s->thermal, s->accumulation, s->depth, s->expected_depth, s->depth_scratch, main));
    if (s->distortion_loss_pair) {
        STEP_TRY(stream_after(second, main));
        STEP_TRY(tn_distortion_loss_term(s->spacing[2], s->weights[2], R, S, 1.0f / (float)R, s->distortion_mult,
                                         s->distortion_loss_pair, s->distortion_grad, second));
    }
    if (s->interlevel_loss) {
        STEP_TRY(stream_after(third, main));
        const float *cp[2] = {s->spacing[0], s->spacing[1]};
        const float *wp[2] = {s->weights[0], s->weights[1]};
        const int32_t p[2] = {s->cfg->num_proposal_samples[0], s->cfg->num_proposal_samples[1]};
        float *g[2] = {s->interlevel_grad[0], s->interlevel_grad[1]};
        STEP_TRY(tn_interlevel_loss_levels(s->spacing[2], s->weights[2], R, S, 2, cp, wp, p, s->interlevel_mult / ((float)R * (float)S),
                                           s->interlevel_loss, g, third));
    }
    return TN_OK;
}

extern "C" int tn_train_step_bwd(const tn_train_step_bwd_args *a) {
    if (!a || !a->field || !a->grads || !a->d_table || !a->d_enc || !a->d_density) return TN_ERR_NULL;
    const int64_t R = a->num_rays;
    const int S = a->n;
    if (R <= 0 || S <= 0) return (R == 0) ? TN_OK : TN_ERR_SHAPE;
    const int64_t N = R * S;
    hipStream_t main = (hipStream_t)a->stream, second = (hipStream_t)a->second, third = (hipStream_t)a->third;
    const tn_hashgrid *grid = &a->field->grid;
    const tn_space *space = &a->field->space;
    if (a->wait_second_first) STEP_TRY(stream_after(main, second));
    STEP_TRY(tn_ray_render_bwd(a->deltas, a->density, a->rgb_samples, a->thermal_samples, a->accumulation, a->d_rgb, a->d_thermal,
                               a->d_accumulation, a->d_weights, a->use_gradient_scaling ? a->starts : nullptr,
                               a->use_gradient_scaling ? a->ends : nullptr, R, S, a->d_rgb_samples, a->d_thermal_samples, a->d_density, main));
    const bool rays = a->d_origins != nullptr;
    STEP_TRY(tn_field_bwd_fused(a->field, R, S, a->enc, a->selector, a->base_out, a->ray_bias, a->rgb_samples, a->d_rgb_samples,
                                a->d_thermal_samples, a->d_density, a->pass_thermal_gradients, a->trunc_exp_min, a->split_form, a->d_enc,
                                a->d_ray_sum, rays ? a->positions : nullptr, a->jacobian, rays ? a->d_positions : nullptr, a->grads,
                                a->fused_workspace, a->fused_workspace_bytes, main));
    // ---- the table scatter ------------------------------------------------------------------------------------------------------
    const int L = grid->num_levels;
    const int first = (a->first_sorted_level > 0 && a->sorted_workspace && a->sorted_workspace_bytes) ? a->first_sorted_level : -1;
    auto atomic_levels = [&](int lo, int hi, hipStream_t st) -> int {
        if (hi <= lo) return TN_OK;
        if (a->spread && lo == 0 && a->spread_workspace && a->spread_workspace_bytes)
            return tn_hash_encode_bwd_spread(grid, space, a->positions, a->d_enc, N, a->d_table, lo, hi, a->spread_workspace,
                                             a->spread_workspace_bytes, st);
        return tn_hash_encode_bwd_levels(grid, space, a->positions, a->d_enc, N, a->d_table, lo, hi, st);
    };
    bool join_second = false;
    if (first < 0 || !a->overlap) {  // one stream: atomic levels, then the bucketed ones
        STEP_TRY(atomic_levels(0, first < 0 ? L : first, main));
        if (first >= 0)
            STEP_TRY(tn_hash_encode_bwd_sorted(grid, space, a->positions, a->d_enc, N, a->d_table, first, a->sorted_workspace,
                                               a->sorted_workspace_bytes, main));
    } else {
        STEP_TRY(stream_after(second, main));
        STEP_TRY(tn_hash_encode_bwd_sorted(grid, space, a->positions, a->d_enc, N, a->d_table, first, a->sorted_workspace,
                                           a->sorted_workspace_bytes, second));
        if (a->defer) {
            STEP_TRY(stream_after(third, main));
            STEP_TRY(atomic_levels(0, first, third));
        } else {
            STEP_TRY(atomic_levels(0, first, main));
            join_second = true;
        }
    }
    // ---- ray-level adjoints: nothing here touches the table gradient ---------------------------------------------------------------
    if (a->d_ray_sum) {
        STEP_TRY(tn_ray_head_bwd(a->field, a->directions, a->camera_indices, R, a->d_ray_sum, a->grads->head0_w, a->d_head0_bias,
                                 a->d_appearance, (a->sh_direction_gradient && rays) ? a->d_ray_inputs : nullptr, main));
        if (a->sh_direction_gradient && rays)
            STEP_TRY(tn_color_input_bwd(a->field, a->d_ray_inputs, a->camera_indices, 1, R, 1, nullptr, 0, nullptr, a->directions,
                                        a->d_directions, main));
    }
    if (rays) STEP_TRY(tn_frustum_positions_bwd(a->d_positions, a->starts, a->ends, R, S, a->d_origins, a->d_directions, main));
    if (join_second) STEP_TRY(stream_after(main, second));
    return TN_OK;
}
