// api.hip -- error state and version of libfrcnn_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace frcnn {

static thread_local char g_last_error[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

int check_hip(hipError_t e, const char *what) {
  if (e == hipSuccess) return FRCNN_OK;
  return set_error(FRCNN_EHIP, "%s: %s", what, hipGetErrorString(e));
}

int check_launch(const char *what) { return check_hip(hipGetLastError(), what); }

}  // namespace frcnn

#ifdef FRCNN_STAMPS
namespace frcnn {
__device__ long long g_stamps[256];
__device__ int g_nstamps;
}
extern "C" int frcnn_debug_stamps(long long *host_out, int reset) {
  int n = 0;
  hipMemcpyFromSymbol(&n, HIP_SYMBOL(frcnn::g_nstamps), sizeof(int));
  hipMemcpyFromSymbol(host_out, HIP_SYMBOL(frcnn::g_stamps), sizeof(long long) * 256);
  if (reset) {
    int z = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(frcnn::g_nstamps), &z, sizeof(int));
  }
  return n;
}
#endif

extern "C" {

int frcnn_version(void) { return 100; /* 0.1.0 */ }

const char *frcnn_last_error(void) { return frcnn::g_last_error; }

}  // extern "C"
