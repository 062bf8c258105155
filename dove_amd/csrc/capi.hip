// Error channel, ABI version and device query of libdove_hip.so (see include/dove_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"
#include "../../include/dove_hip.h"

static thread_local char g_err[512] = "";

extern "C" void dove_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dove_last_error(void) { return g_err; }

extern "C" int dove_abi_version(void) { return DOVE_ABI_VERSION; }

extern "C" int dove_device_info(int dev, char* name, int name_len, int* cu_count, long long* total_mem) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    dove_set_error("device_info: hipGetDeviceProperties(%d) failed", dev);
    return DOVE_EINVAL;
  }
  if (name && name_len > 0) {
    strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (total_mem) *total_mem = (long long)p.totalGlobalMem;
  return DOVE_OK;
}
