// C-ABI housekeeping for libvitlens_hip.so: error string, version, device probe.
#include <hip/hip_runtime.h>
#include <string.h>
#include "vitlens_hip.h"

static thread_local char g_err[512] = "";

extern "C" int vl_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return 1;
}
extern "C" const char* vl_last_error(void) { return g_err; }
extern "C" int vl_version(void) { return VL_ABI_VERSION; }

extern "C" int vl_device_info(int device, char* arch, int arch_len, int* cus, int* clock_khz, long* hbm_bytes) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return vl_set_error(hipGetErrorString(e));
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  if (cus) *cus = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = (long)prop.totalGlobalMem;
  return 0;
}
