// Library-level entry points: version, error string, architecture check.
#include "common.h"

#include <string.h>

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

}  // namespace jpdse

extern "C" {

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
