// SamsungV2Decompressor plans (rsx_samsung_v2.hip), used by rsx_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "rsx.h"

namespace rsx {

struct Sv2Plan;
struct KernelTimer;

int samsung_v2_validate(const rsx_samsung_v2_desc& d, const rsx_image& img);
int samsung_v2_plan_create(rsx_ctx* ctx, int n_jobs, const rsx_samsung_v2_job* jobs,
                           Sv2Plan** out);
void samsung_v2_plan_destroy(Sv2Plan* p);
int samsung_v2_plan_run(Sv2Plan* p, const void* in_dev, void* out_dev, hipStream_t s,
                        KernelTimer* timer);
int samsung_v2_plan_results(Sv2Plan* p, hipStream_t s, bool ran, int32_t* job_status);

} // namespace rsx
