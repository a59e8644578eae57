// TEST INFRASTRUCTURE.  The f32 / f16 MFMA export path (lbs_forward.hip) is not emulated: its entry points report "unsupported".
#include <hip/hip_runtime.h>
#include "../../include/moshii.h"
struct ModelDev;
extern "C" void moshii_lbs32_free(void*) {}
extern "C" int moshii_lbs32_prepare(moshii_model_t) { return MOSHII_ERR_UNSUPPORTED; }
extern "C" hipError_t moshii_launch_lbs_f32(hipStream_t, const ModelDev*, int, const float*, const float*, float*, void*) { return hipErrorUnknown; }
