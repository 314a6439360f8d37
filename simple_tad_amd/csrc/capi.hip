// Error reporting, version and device-info entry points of libtad_mi355x.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace tad {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TAD_ELAUNCH;
  }
  return TAD_OK;
}

}  // namespace tad

extern "C" {

int tad_abi_version(void) { return TAD_ABI_VERSION; }

const char* tad_last_error_string(void) { return tad::g_err; }

int tad_device_info(int* cu_count, int* clock_khz, int* lds_bytes_per_cu, char* name, int name_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { tad::set_error("device_info: no HIP device"); return TAD_ELAUNCH; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { tad::set_error("device_info: hipGetDeviceProperties failed"); return TAD_ELAUNCH; }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  return TAD_OK;
}

}  // extern "C"
