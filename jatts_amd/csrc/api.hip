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

// ---- scratch of the deterministic reductions (det_reduce.h)
#include "det_reduce.h"
jatts_ws_t jatts_g_ws = {nullptr, nullptr, 0};

extern "C" int jatts_set_workspace(void* buf, int64_t bytes) {
  if (!buf) {
    jatts_g_ws = {nullptr, nullptr, 0};
    return JATTS_OK;
  }
  const int64_t head = (int64_t)JATTS_WS_TICKETS * 4;
  if (bytes < head + 4096 || ((uintptr_t)buf & 15)) return jatts_set_error_msg(JATTS_ERR_ARG, "set_workspace: >= 20 KiB, 16-byte aligned, ZERO-filled");
  jatts_g_ws.tickets = (unsigned*)buf;
  jatts_g_ws.slabs = (float*)((char*)buf + head);
  jatts_g_ws.slab_floats = (bytes - head) / 4;
  return JATTS_OK;
}

int jatts_ws_need(int64_t groups, int64_t slab_floats) {
  if (groups <= JATTS_WS_TICKETS && slab_floats <= jatts_g_ws.slab_floats) return JATTS_OK;
  char msg[256];
  snprintf(msg, sizeof(msg), "deterministic reduction needs %lld tickets (max %d) and %lld bytes of scratch: call jatts_set_workspace with a "
           "zero-filled buffer of at least that size (+ %d bytes of tickets)", (long long)groups, JATTS_WS_TICKETS, (long long)slab_floats * 4,
           JATTS_WS_TICKETS * 4);
  return jatts_set_error_msg(JATTS_ERR_ARG, msg);
}
