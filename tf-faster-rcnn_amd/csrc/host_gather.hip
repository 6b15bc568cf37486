// host_gather.hip -- the native runtime behind the "final host gather" of a batch that is sharded over the GPUs of one node
// (the reference is single-device and returns host arrays from predict(), faster_rcnn_predictor.py:56).
//
// Every rank's small per-image outputs travel in two hops: a copy kernel on the rank's stream writes them into the rank's
// pinned staging slot, and -- once the step's event has completed -- the host moves the staged bytes into their place in the
// buffer all ranks share (POSIX shared memory, mapped by the Python side).  The second hop runs on a worker thread owned by
// this object, so the thread that enqueues GPU work pays one call per step: no event wait, no copy, no interpreter work.
// (hipHostRegister of a shared mapping, which would let the GPU write the shared buffer directly, is refused to ordinary
// users on this platform.)
//
// One mutex + two condition variables; jobs are few (one per step) and small, a lock-free queue would buy nothing here.
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "common.hpp"

static const int kHostGatherSegments = 8;  // == the copy kernel's segments per launch (codec.hip)

namespace {

struct HostGatherJob {
  int slot;
  int n;
  const void *stage[kHostGatherSegments];
  void *dst[kHostGatherSegments];
  size_t bytes[kHostGatherSegments];
};

}  // namespace

struct frcnn_host_gather {
  int device = 0;
  int slots = 0;
  std::vector<hipEvent_t> events;
  std::vector<char> staged;  // slot holds a job that is not yet in the shared buffer
  std::deque<HostGatherJob> queue;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::thread worker;
  bool stop = false;
  int failed = 0;  // first HIP error seen by the worker (hipError_t), reported by the next submit / flush
  // diagnostics
  double publish_seconds = 0.0, submit_wait_seconds = 0.0;
  long long jobs = 0;
};

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void worker_main(frcnn_host_gather *h) {
  (void)hipSetDevice(h->device);
  for (;;) {
    HostGatherJob job;
    {
      std::unique_lock<std::mutex> lock(h->mu);
      h->cv_work.wait(lock, [&] { return h->stop || !h->queue.empty(); });
      if (h->queue.empty()) return;  // stop requested and nothing left to publish
      job = h->queue.front();
      h->queue.pop_front();
    }
    const hipError_t e = hipEventSynchronize(h->events[job.slot]);
    const double t0 = now_s();
    if (e == hipSuccess)
      for (int i = 0; i < job.n; ++i) memcpy(job.dst[i], job.stage[i], job.bytes[i]);
    const double dt = now_s() - t0;
    {
      std::lock_guard<std::mutex> lock(h->mu);
      if (e != hipSuccess && !h->failed) h->failed = (int)e;
      h->staged[job.slot] = 0;
      h->publish_seconds += dt;
      h->jobs += 1;
    }
    h->cv_done.notify_all();
  }
}

}  // namespace

using namespace frcnn;

extern "C" {

int frcnn_host_gather_create(int slots, frcnn_host_gather_t **out) {
  FRCNN_REQUIRE(out, "host_gather_create: null out pointer");
  FRCNN_REQUIRE(slots >= 1 && slots <= 64, "host_gather_create: slots must be 1..64 (got %d)", slots);
  *out = nullptr;
  frcnn_host_gather *h = new (std::nothrow) frcnn_host_gather();
  FRCNN_REQUIRE(h, "host_gather_create: out of host memory");
  int rc = check_hip(hipGetDevice(&h->device), "host_gather_create: hipGetDevice");
  h->slots = slots;
  h->staged.assign((size_t)slots, 0);
  for (int i = 0; i < slots && !rc; ++i) {
    hipEvent_t ev = nullptr;
    // host-visible completion (system-scope release: the staged bytes are in host memory when it fires); the worker blocks
    // on it instead of spinning, it has nothing else to do and must not steal a core from the enqueuing thread
    rc = check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventBlockingSync), "host_gather_create: hipEventCreateWithFlags");
    if (!rc) h->events.push_back(ev);
  }
  if (rc) {
    for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
    delete h;
    return rc;
  }
  h->worker = std::thread(worker_main, h);
  *out = h;
  return FRCNN_OK;
}

int frcnn_host_gather_submit(frcnn_host_gather_t *h, int slot, const void *const *srcs, void *const *stage,
                             void *const *dsts, const size_t *bytes, int n, frcnn_stream_t stream) {
  FRCNN_REQUIRE(h, "host_gather_submit: null handle");
  FRCNN_REQUIRE(slot >= 0 && slot < h->slots, "host_gather_submit: slot %d outside 0..%d", slot, h->slots - 1);
  FRCNN_REQUIRE(n >= 0 && n <= kHostGatherSegments, "host_gather_submit: 0..%d segments (got %d)", kHostGatherSegments, n);
  FRCNN_REQUIRE(n == 0 || (srcs && stage && dsts && bytes), "host_gather_submit: null table");
  HostGatherJob job = {};
  job.slot = slot;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] == 0) continue;
    FRCNN_REQUIRE(srcs[i] && stage[i] && dsts[i], "host_gather_submit: null pointer in segment %d", i);
    job.stage[job.n] = stage[i];
    job.dst[job.n] = dsts[i];
    job.bytes[job.n] = bytes[i];
    ++job.n;
  }
  {
    // back-pressure: the staging slot is free again once the step that used it `slots` steps ago is in the shared buffer
    std::unique_lock<std::mutex> lock(h->mu);
    if (h->failed) return set_error(FRCNN_EHIP, "host_gather: the worker's event wait failed: %s", hipGetErrorString((hipError_t)h->failed));
    if (h->staged[slot]) {
      const double t0 = now_s();
      h->cv_done.wait(lock, [&] { return !h->staged[slot]; });
      h->submit_wait_seconds += now_s() - t0;
    }
  }
  int rc = frcnn_gather_copy_multi(srcs, stage, bytes, n, stream);  // device -> pinned staging, a kernel on `stream`
  if (rc) return rc;
  rc = check_hip(hipEventRecord(h->events[slot], (hipStream_t)stream), "host_gather_submit: hipEventRecord");
  if (rc) return rc;
  {
    std::lock_guard<std::mutex> lock(h->mu);
    h->staged[slot] = 1;
    h->queue.push_back(job);
  }
  h->cv_work.notify_one();
  return FRCNN_OK;
}

int frcnn_host_gather_flush(frcnn_host_gather_t *h) {
  FRCNN_REQUIRE(h, "host_gather_flush: null handle");
  std::unique_lock<std::mutex> lock(h->mu);
  h->cv_done.wait(lock, [&] {
    for (char s : h->staged)
      if (s) return false;
    return true;
  });
  if (h->failed) return set_error(FRCNN_EHIP, "host_gather: the worker's event wait failed: %s", hipGetErrorString((hipError_t)h->failed));
  return FRCNN_OK;
}

int frcnn_host_gather_stats(frcnn_host_gather_t *h, double *publish_seconds, double *submit_wait_seconds, long long *jobs) {
  FRCNN_REQUIRE(h, "host_gather_stats: null handle");
  std::lock_guard<std::mutex> lock(h->mu);
  if (publish_seconds) *publish_seconds = h->publish_seconds;
  if (submit_wait_seconds) *submit_wait_seconds = h->submit_wait_seconds;
  if (jobs) *jobs = h->jobs;
  return FRCNN_OK;
}

int frcnn_host_gather_destroy(frcnn_host_gather_t *h) {
  if (!h) return FRCNN_OK;
  {
    std::lock_guard<std::mutex> lock(h->mu);
    h->stop = true;  // the worker publishes what is queued, then leaves
  }
  h->cv_work.notify_all();
  if (h->worker.joinable()) h->worker.join();
  for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
  delete h;
  return FRCNN_OK;
}

}  // extern "C"
