// coresidency_probe.hip -- does a small-footprint workgroup (256 threads, <= 128 VGPRs) become resident beside the crop
// kernel's workgroups, where a 1024-thread workgroup has to wait for a crop launch's tail?  (diagnostic; not part of the
// library).  One or two "lanes" (streams) launch the library crop kernel back to back; a third stream runs a chain of
// 16-workgroup spin kernels (T threads, LDS bytes as the proposal kernel would take) that each stay busy for SPIN_US.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/coresidency_probe.hip -o tools/diag/coresidency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#include "../tf-faster-rcnn_amd/csrc/crop_resize.hip"
#include "../tf-faster-rcnn_amd/csrc/api.hip"

template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(4))) void spin_kernel(float *sink, long long ticks, int prio) {
  extern __shared__ float sm[];
  if (prio) __builtin_amdgcn_s_setprio(3);
  const long long t0 = wall_clock64();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  sm[threadIdx.x] = a;
  __syncthreads();
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 64; ++i) a = a * b + sm[(threadIdx.x + i) & (T - 1)];
    __syncthreads();
  }
  if (a == 12345.f) sink[0] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv) {
  const int B = 16, H = 32, W = 32, C = 576, R = 300, n = B * R;
  const long long out_floats = (long long)n * 49 * C;
  float *img, *boxes, *out[2], *sink;
  CK(hipMalloc(&img, (size_t)B * H * W * C * 4));
  CK(hipMalloc(&boxes, (size_t)n * 16));
  CK(hipMalloc(&out[0], (size_t)out_floats * 4));
  CK(hipMalloc(&out[1], (size_t)out_floats * 4));
  CK(hipMalloc(&sink, 64));
  std::vector<float> hb(n * 4), hi((size_t)B * H * W * C);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float cy = rand() / (float)RAND_MAX, cx = rand() / (float)RAND_MAX;
    float h = 0.15f + 0.55f * (rand() / (float)RAND_MAX), w = 0.15f + 0.55f * (rand() / (float)RAND_MAX);
    hb[i * 4 + 0] = std::max(0.f, cy - h / 2); hb[i * 4 + 2] = std::min(1.f, cy + h / 2);
    hb[i * 4 + 1] = std::max(0.f, cx - w / 2); hb[i * 4 + 3] = std::min(1.f, cx + w / 2);
  }
  for (auto &v : hi) v = rand() / (float)RAND_MAX;
  CK(hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));

  int lo, hiP;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hiP));  // lo = least priority (numerically greatest)
  hipStream_t lane[2], side;
  CK(hipStreamCreateWithPriority(&lane[0], hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&lane[1], hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hiP));
  const long long ticks_per_us = 100;  // wall_clock64: 100 MHz
  const int ncrop = 80;

  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));

  auto run = [&](int lanes, int T, int spin_us, int lds_kb, int prio, const char *name) {
    hipEvent_t a, b, sa, sb;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&sa)); CK(hipEventCreate(&sb));
    CK(hipDeviceSynchronize());
    // enough spin launches to outlast the crops; count those finished by the time the crops end via a second event pair
    const int nspin = T ? 400 : 0;
    for (int i = 0; i < 4; ++i) frcnn_crop_and_resize(img, B, H, W, C, boxes, nullptr, n, R, 7, 7, 0.f, out[i & 1], lane[i % lanes]);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, lane[0]));
    if (T) CK(hipEventRecord(sa, side));
    // interleave the enqueues so that the queues fill concurrently
    int si = 0;
    for (int i = 0; i < ncrop; ++i) {
      frcnn_crop_and_resize(img, B, H, W, C, boxes, nullptr, n, R, 7, 7, 0.f, out[i & 1], lane[i % lanes]);
      for (int k = 0; k < 2 && si < 160 && T; ++k, ++si) {
        if (T == 1024) hipLaunchKernelGGL(spin_kernel<1024>, dim3(16), dim3(1024), lds_kb * 1024, side, sink, spin_us * ticks_per_us, prio);
        else hipLaunchKernelGGL(spin_kernel<256>, dim3(16), dim3(256), lds_kb * 1024, side, sink, spin_us * ticks_per_us, prio);
      }
    }
    (void)nspin;
    if (T) CK(hipEventRecord(sb, side));
    if (lanes == 2) { hipEvent_t j; CK(hipEventCreate(&j)); CK(hipEventRecord(j, lane[1])); CK(hipStreamWaitEvent(lane[0], j, 0)); }
    CK(hipEventRecord(b, lane[0]));
    CK(hipDeviceSynchronize());
    float ms = 0, sms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (T) CK(hipEventElapsedTime(&sms, sa, sb));
    printf("%-46s crop period %7.1f us", name, ms * 1e3 / ncrop);
    if (T) printf("   spin kernel period %7.1f us (busy %d us each, %d launched)", sms * 1e3 / si, spin_us, si);
    printf("\n");
  };
  run(1, 0, 0, 0, 0, "1 lane, crops only");
  run(2, 0, 0, 0, 0, "2 lanes, crops only");
  run(1, 1024, 60, 100, 1, "1 lane + 1024-thread spin (100 KB LDS)");
  run(2, 1024, 60, 100, 1, "2 lanes + 1024-thread spin (100 KB LDS)");
  run(1, 256, 60, 70, 1, "1 lane + 256-thread spin (70 KB LDS)");
  run(2, 256, 60, 70, 1, "2 lanes + 256-thread spin (70 KB LDS)");
  run(2, 256, 120, 70, 1, "2 lanes + 256-thread spin 120 us (70 KB LDS)");
  run(2, 256, 120, 70, 0, "2 lanes + 256-thread spin 120 us, no setprio");
  run(2, 256, 120, 24, 1, "2 lanes + 256-thread spin 120 us (24 KB LDS)");
  return 0;
}
