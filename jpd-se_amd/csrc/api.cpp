// Library-level entry points: version, error string, architecture check.
#include "common.h"

#include <string.h>

#include <vector>

namespace jpdse {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return JPDSE_OK;
}

// ---- in-library timer for the HBM-bound calls (bench.py "roofline_hbm"): hipEvent pairs around whole InstanceNorm /
// Adam calls, recorded on the stream the kernels run on; each region carries the call's ALGORITHMIC bytes.
struct HbmProf {
  bool on = false;
  int used = 0;
  int max_regions = 0;          // of the CURRENT selection: `ev` only ever grows, `bytes` / `cls` are sized to this
  std::vector<hipEvent_t> ev;   // 2 per region
  std::vector<double> bytes;
  std::vector<int> cls;
};
static HbmProf g_hbm;

int hbm_prof_begin(hipStream_t s) {
  if (!g_hbm.on || g_hbm.used >= g_hbm.max_regions || (size_t)(2 * g_hbm.used + 2) > g_hbm.ev.size()) return -1;
  (void)hipEventRecord(g_hbm.ev[2 * g_hbm.used], s);
  return g_hbm.used++;
}

void hbm_prof_end(int slot, int cls, double bytes, hipStream_t s) {
  if (slot < 0) return;
  (void)hipEventRecord(g_hbm.ev[2 * slot + 1], s);
  g_hbm.bytes[slot] = bytes;
  g_hbm.cls[slot] = cls;
}

}  // namespace jpdse

extern "C" {

int jpdse_prof_hbm_select(int32_t enable, int32_t max_regions) {
  using jpdse::g_hbm;
  g_hbm.on = false;
  g_hbm.used = 0;
  g_hbm.max_regions = 0;
  if (!enable) return JPDSE_OK;
  JPDSE_REQUIRE(max_regions > 0, "prof_hbm_select: max_regions must be positive");
  while (g_hbm.ev.size() < (size_t)2 * max_regions) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return jpdse::set_error(JPDSE_ELAUNCH, "prof_hbm_select: hipEventCreate failed");
    g_hbm.ev.push_back(e);
  }
  g_hbm.bytes.assign(max_regions, 0.0);
  g_hbm.cls.assign(max_regions, -1);
  g_hbm.max_regions = max_regions;
  g_hbm.on = true;
  return JPDSE_OK;
}

int jpdse_prof_hbm_collect(int32_t cls, double* total_ms, double* total_bytes, int64_t* regions) {
  using jpdse::g_hbm;
  JPDSE_REQUIRE(total_ms && total_bytes && regions && cls >= 0 && cls <= 2, "prof_hbm_collect: bad argument");
  double ms = 0.0, by = 0.0;
  int64_t n = 0;
  for (int i = 0; i < g_hbm.used; ++i) {
    if (g_hbm.cls[i] != cls) continue;
    if (hipEventSynchronize(g_hbm.ev[2 * i + 1]) != hipSuccess)
      return jpdse::set_error(JPDSE_ELAUNCH, "prof_hbm_collect: hipEventSynchronize failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_hbm.ev[2 * i], g_hbm.ev[2 * i + 1]) != hipSuccess)
      return jpdse::set_error(JPDSE_ELAUNCH, "prof_hbm_collect: hipEventElapsedTime failed");
    ms += t;
    by += g_hbm.bytes[i];
    ++n;
  }
  *total_ms = ms;
  *total_bytes = by;
  *regions = n;
  return JPDSE_OK;
}

int jpdse_version(void) { return JPDSE_ABI_VERSION; }

const char* jpdse_last_error(void) { return jpdse::g_err; }

int jpdse_arch_check(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return jpdse::set_error(JPDSE_EARCH, "hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return jpdse::set_error(JPDSE_EARCH, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
  return JPDSE_OK;
}

}  // extern "C"
