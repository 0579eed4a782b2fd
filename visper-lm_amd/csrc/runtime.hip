// Error reporting + version for the C-ABI library (no exceptions cross the ABI; every entry point
// returns 0 or a negative VP_ERR_* code and leaves a thread-local message for vp_last_error_string()).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void vp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int vp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    vp_set_error("%s: HIP launch error: %s", what, hipGetErrorString(e));
    return VP_ERR_HIP;
  }
  return VP_OK;
}

extern "C" {
const char* vp_last_error_string(void) { return g_err; }
int vp_version(void) { return 100; }   // 0.1.0
// Zero `bytes` bytes of device memory on `stream` (gradient buffers, scatter targets): hipMemsetAsync, no kernel of ours and none of the
// caller's tensor library on the hot path.
int vp_memset_zero(void* ptr, long bytes, hipStream_t stream) {
  VP_REQUIRE(ptr && bytes >= 0, VP_ERR_BAD_ARG, "vp_memset_zero: bad args");
  if (bytes == 0) return VP_OK;
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, stream);
  if (e != hipSuccess) {
    vp_set_error("vp_memset_zero: %s", hipGetErrorString(e));
    return VP_ERR_HIP;
  }
  return VP_OK;
}
// Device facts the host side sizes grids / workspaces with.
int vp_device_info(int* cu_count, int* wave_size, long* lds_bytes_per_cu) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    vp_set_error("vp_device_info: no HIP device");
    return VP_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (wave_size) *wave_size = prop.warpSize;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (long)prop.maxSharedMemoryPerMultiProcessor;
  return VP_OK;
}
}
