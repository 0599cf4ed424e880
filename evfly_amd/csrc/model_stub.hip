// TEMPORARY: model entry points not built yet (replaced by model.hip).
#include "common.h"
using namespace evfly;
#define NI(name) return fail(-3, name ": not implemented yet")
extern "C" {
int evfly_model_create(const evfly_model_config *, evfly_model **) { NI("evfly_model_create"); }
int evfly_model_load_tensor(evfly_model *, const char *, const float *, const int64_t *, int) { NI("evfly_model_load_tensor"); }
int evfly_model_finalize(evfly_model *) { NI("evfly_model_finalize"); }
void evfly_model_destroy(evfly_model *) {}
int evfly_unet_forward(evfly_model *, const float *, int, int, float *, float *, float *, float *, void *) { NI("evfly_unet_forward"); }
int evfly_vit_forward(evfly_model *, const float *, int, int, int, const float *, const float *, int, int, float *, float *, float *, void *) { NI("evfly_vit_forward"); }
int evfly_vit_stage_forward(evfly_model *, int, const float *, int, int, int, float *, void *) { NI("evfly_vit_stage_forward"); }
int evfly_e2v_forward(evfly_model *, const float *, const float *, int, int, float *, float *, float *, float *, float *, float *, float *, void *) { NI("evfly_e2v_forward"); }
int64_t evfly_model_tap(evfly_model *, const char *, float *, int64_t, int64_t *, void *) { NI("evfly_model_tap"); }
int evfly_model_set_profiling(evfly_model *, int) { NI("evfly_model_set_profiling"); }
int evfly_model_profile_count(evfly_model *) { return 0; }
int evfly_model_profile_get(evfly_model *, int, char *, int, double *, double *, double *, int *) { NI("evfly_model_profile_get"); }
int evfly_model_profile_reset(evfly_model *) { return 0; }
int evfly_op_conv2d_nhwc(const float *, int, int, int, int, const float *, const float *, int, int, int, int, int, int, const float *, float *, int, void *) { NI("evfly_op_conv2d_nhwc"); }
}
