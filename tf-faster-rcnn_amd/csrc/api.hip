// api.hip -- error state and version of libfrcnn_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

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

namespace {
struct LdsLimit {
  const void *kernel;
  int device;
  size_t bytes;
};
std::mutex g_lds_mutex;
std::vector<LdsLimit> g_lds_limits;
}  // namespace

int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what) {
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), what);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  LdsLimit *hit = nullptr;
  for (auto &e : g_lds_limits)
    if (e.kernel == kernel && e.device == dev) hit = &e;
  if (hit && hit->bytes >= bytes) return FRCNN_OK;
  rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), what);
  if (rc) return rc;
  if (hit)
    hit->bytes = bytes;
  else
    g_lds_limits.push_back({kernel, dev, bytes});
  return FRCNN_OK;
}

static const char *const kKnobNames[kKnobCount] = {
    "FRCNN_CROP_SLICE",       "FRCNN_CROP_ROWS",       "FRCNN_DECODER_THREADS",
    "FRCNN_ROI_BBOX_UNFUSED", "FRCNN_DECODER_UNFUSED", "FRCNN_ROI_BBOX_BITMASK_MIN_R",
    "FRCNN_PROPOSAL_CHUNK",   "FRCNN_PROPOSAL_THREADS", "FRCNN_IOU_TILE", "FRCNN_IOU_MAP_FLAT", "FRCNN_CROP_PREFETCH", "FRCNN_CROP_SPLIT", "FRCNN_CROP_SORT", "FRCNN_IOU_REDUCE_SPLIT",
    "FRCNN_PROPOSAL_EXACT_SLICES", "FRCNN_SORT_RADIX_ONLY", "FRCNN_BITMASK_THREADS", "FRCNN_CROP_STAGE"};

struct KnobTable {
  std::atomic<int> value[kKnobCount];
  std::atomic<bool> set[kKnobCount];
  KnobTable() {  // the one place the environment is read: library load
    for (int k = 0; k < kKnobCount; ++k) {
      const char *e = getenv(kKnobNames[k]);
      set[k].store(e != nullptr, std::memory_order_relaxed);
      value[k].store(e ? atoi(e) : 0, std::memory_order_relaxed);
    }
  }
};
static KnobTable g_knobs;

bool knob_set(Knob k) { return g_knobs.set[k].load(std::memory_order_relaxed); }
int knob(Knob k, int unset_value) { return knob_set(k) ? g_knobs.value[k].load(std::memory_order_relaxed) : unset_value; }

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

int frcnn_debug_set_schedule(const char *name, int is_set, int value) {
  using namespace frcnn;
  FRCNN_REQUIRE(name, "debug_set_schedule: null name");
  for (int k = 0; k < kKnobCount; ++k) {
    if (strcmp(name, kKnobNames[k]) == 0) {
      g_knobs.value[k].store(value, std::memory_order_relaxed);
      g_knobs.set[k].store(is_set != 0, std::memory_order_relaxed);
      return FRCNN_OK;
    }
  }
  return set_error(FRCNN_EINVAL, "debug_set_schedule: unknown knob '%s'", name);
}

}  // extern "C"
