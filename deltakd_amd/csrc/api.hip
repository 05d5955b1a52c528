// Error channel, version and device queries of the C ABI (include/dkd.h).
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dkd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dkd_version(void) { return 100; }
extern "C" const char* dkd_last_error(void) { return g_err; }

extern "C" int dkd_device_info(int device, int* cu_count, char* name, int name_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    dkd_set_error("device_info: %s", hipGetErrorString(e));
    return DKD_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return DKD_OK;
}

