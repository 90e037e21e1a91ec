// NeO-360 (NeRF_TP) entry points of the C ABI.
#include "../../include/neo360_hip.h"

#include <hip/hip_runtime.h>

extern "C" {

int neo_tp_upload_mlp(neo_ctx*, int, int, const float* const*, const float* const*, void*) { return NEO_ERR_STATE; }

int neo_tp_set_scene(neo_ctx*, const float*, const float*, const float*, int, int, int, int, const float*, int, int,
                     int, float, float, void*) { return NEO_ERR_STATE; }

int neo_tp_render(neo_ctx*, const float*, const float*, const float*, int, int, const float*, int, float, float,
                  float, int, int, int, const neo_tp_level_out*, const neo_tp_level_out*, void*) { return NEO_ERR_STATE; }

}  // extern "C"
