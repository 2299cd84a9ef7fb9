// Error reporting + device info for libjatts_hip.so.
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

int jatts_set_error(hipError_t e, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  return JATTS_ERR_HIP;
}
int jatts_set_error_msg(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

extern "C" int jatts_abi_version(void) { return JATTS_ABI_VERSION; }
extern "C" const char* jatts_last_error(void) { return g_err; }
extern "C" int jatts_device_info(char* buf, int buflen) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) return jatts_set_error(e, __FILE__, __LINE__);
  snprintf(buf, buflen, "%s|%s|cus=%d|lds=%zu", p.name, p.gcnArchName, p.multiProcessorCount,
           (size_t)p.sharedMemPerBlock);
  return JATTS_OK;
}
