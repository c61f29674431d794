// rsx_ljpeg.h -- host interface of the lossless-JPEG decode pipeline
// (implementation + kernels: rsx_ljpeg.hip).
#pragma once

#include "rsx_internal.h"

#include <vector>

namespace rsx {

struct LJpegJobIn {
  int status = RSX_OK;    // validation result; failed jobs are skipped
  StreamGeom geom;        // flattened geometry (rsx_host.cpp)
  const rsx_huff_table* tables = nullptr;
  int n_tables = 0;
  int rows_per_restart_interval = 0; // LJPEG only; 0 = no restart markers
  int frame_h = 0;
};

struct LJpegPlan;

int ljpeg_plan_create(rsx_ctx* ctx, const std::vector<LJpegJobIn>& jobs,
                      LJpegPlan** out);
int ljpeg_plan_run(LJpegPlan* plan, const void* in_dev, void* out_dev,
                   hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);
int ljpeg_plan_results(LJpegPlan* plan, hipStream_t stream, bool ran,
                       int32_t* job_status, uint32_t* job_consumed);
void ljpeg_plan_destroy(LJpegPlan* plan);
const char* ljpeg_dominant_kernel_name();

struct LJpegPlanDeleter {
  void operator()(LJpegPlan* p) const { ljpeg_plan_destroy(p); }
};

} // namespace rsx
