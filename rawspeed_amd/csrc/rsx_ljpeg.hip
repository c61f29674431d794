// placeholder until the LJPEG pipeline lands
#include "rsx_ljpeg.h"
namespace rsx {
struct LJpegPlan { int n = 0; };
int ljpeg_plan_create(rsx_ctx*, const std::vector<LJpegJobIn>& jobs, LJpegPlan** out) {
  *out = new LJpegPlan{int(jobs.size())};
  return RSX_OK;
}
int ljpeg_plan_run(LJpegPlan*, const void*, void*, hipStream_t, hipEvent_t, hipEvent_t) {
  return RSX_ERR_UNSUPPORTED;
}
int ljpeg_plan_results(LJpegPlan* p, hipStream_t, bool, int32_t* st, uint32_t* c) {
  for (int i = 0; i < p->n; ++i) { if (st) st[i] = RSX_ERR_UNSUPPORTED; if (c) c[i] = 0; }
  return RSX_ERR_UNSUPPORTED;
}
void ljpeg_plan_destroy(LJpegPlan* p) { delete p; }
const char* ljpeg_dominant_kernel_name() { return "ljpeg_decode_kernel"; }
}
