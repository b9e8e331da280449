// MFMA form of the main-field kernel (placeholder until the fp32-MFMA kernel lands in this file).
#include "tn_field_eval.h"

namespace tn {
int launch_main_mfma(const tn_thermal_field *, const tn_render_config *, const tn_render_inputs *,
                     const tn_render_outputs *, long long, const float *, unsigned *, hipStream_t) {
    return TN_ERR_UNSUPPORTED;
}
}  // namespace tn

extern "C" {
size_t tn_field_prepare_bytes(const tn_thermal_field *) { return 0; }
int tn_field_prepare(const tn_thermal_field *, void *, size_t, void *) { return TN_ERR_UNSUPPORTED; }
}
