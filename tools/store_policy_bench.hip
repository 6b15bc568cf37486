// store_policy_bench.hip -- sustained rate of the crop kernel's store pattern (one 110 KB region per 256-thread workgroup,
// 4 stores of 16 B per lane per trip) under the cache-policy bits of global_store on gfx950 (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_policy_bench.hip -o tools/diag/store_policy_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEF_STORE(NAME, BITS)                                                                     \
  __device__ __forceinline__ void NAME(f32x4 *p, f32x4 v) {                                      \
    asm volatile("global_store_dwordx4 %0, %1, off " BITS : : "v"(p), "v"(v) : "memory");       \
  }
DEF_STORE(st_plain, "")
DEF_STORE(st_nt, "nt")
DEF_STORE(st_sc0, "sc0")
DEF_STORE(st_sc1, "sc1")
DEF_STORE(st_sc0sc1, "sc0 sc1")
DEF_STORE(st_sc0nt, "sc0 nt")
DEF_STORE(st_sc1nt, "sc1 nt")
DEF_STORE(st_all, "sc0 sc1 nt")

template <int POLICY>
__global__ __launch_bounds__(256) void store_kernel(f32x4 *out, float v) {
  const int per = 7056;
  f32x4 *dst = out + (long long)blockIdx.x * per;
  const f32x4 x = (f32x4)(v);
  for (int e = threadIdx.x; e < per; e += 256) {
    if (POLICY == 0) st_plain(dst + e, x);
    if (POLICY == 1) st_nt(dst + e, x);
    if (POLICY == 2) st_sc0(dst + e, x);
    if (POLICY == 3) st_sc1(dst + e, x);
    if (POLICY == 4) st_sc0sc1(dst + e, x);
    if (POLICY == 5) st_sc0nt(dst + e, x);
    if (POLICY == 6) st_sc1nt(dst + e, x);
    if (POLICY == 7) st_all(dst + e, x);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <class F>
static float time_ms(F f, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main() {
  const int n = 4800;
  const long long n4 = (long long)n * 7056;
  f32x4 *out;
  CK(hipMalloc(&out, (size_t)n4 * 16));
  const double bytes = (double)n4 * 16;
  const char *names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  float ms;
#define RUN(P) ms = time_ms([&] { hipLaunchKernelGGL(store_kernel<P>, dim3(n), dim3(256), 0, 0, out, 1.f); }, 40); \
  printf("store, block per crop, %-12s %8.2f us  %7.1f GB/s\n", names[P], ms * 1e3, bytes / ms / 1e6);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  RUN(0) RUN(1)
  return 0;
}
