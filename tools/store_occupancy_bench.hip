// store_occupancy_bench.hip -- sustained rate of the crop kernel's store pattern (one 110 KB region per workgroup) against the
// number of workgroups resident per CU (dynamic LDS as the occupancy limiter) and the workgroup size (diagnostic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_occupancy_bench.hip -o tools/diag/store_occupancy_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int T, int U>
__global__ __launch_bounds__(T) void store_crop_kernel(f32x4 *out, float v) {
  extern __shared__ float pad[];
  const int per = 7056;
  f32x4 *dst = out + (long long)blockIdx.x * per;
  f32x4 x = (f32x4)(v);
  if (v == 12345.f) pad[threadIdx.x] = v;
  int e = threadIdx.x;
  for (; e + (U - 1) * T < per; e += U * T) {
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x, &dst[e + u * T]);
  }
  for (; e < per; e += T) __builtin_nontemporal_store(x, &dst[e]);
}
// the same pattern, but a wave waits for the acknowledgement of its stores every U of them (at most U KB in flight per wave)
template <int T, int U>
__global__ __launch_bounds__(T) void store_crop_paced_kernel(f32x4 *out, float v) {
  const int per = 7056;
  f32x4 *dst = out + (long long)blockIdx.x * per;
  f32x4 x = (f32x4)(v);
  int e = threadIdx.x;
  for (; e + (U - 1) * T < per; e += U * T) {
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x, &dst[e + u * T]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (; e < per; e += T) __builtin_nontemporal_store(x, &dst[e]);
}
// persistent grid-stride over 4 KB chunks in address order, paced the same way
template <int U>
__global__ __launch_bounds__(256) void store_gs_paced_kernel(f32x4 *out, long long n4, float v) {
  f32x4 x = (f32x4)(v);
  const long long stride = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (e + u * stride < n4) __builtin_nontemporal_store(x, &out[e + u * stride]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// 4 KB chunks in address order, one per workgroup; MAP decides which XCD (= blockIdx % 8) writes which chunk:
//  0: chunk c by XCD c % 8;  1: by XCD (c + 1) % 8;  2: by XCD (c / 8) % 8 (eight consecutive chunks per XCD);
//  3: by XCD (c / 64) % 8;   4: chunk c by XCD (c % 8) ^ ((c / 8) % 8) (every residue class visits every XCD)
template <int MAP>
__global__ __launch_bounds__(256) void store_chunk_map_kernel(f32x4 *out, long long n4, float v) {
  const unsigned b = blockIdx.x;
  unsigned c = b;
  if (MAP == 1) c = (b & ~7u) | ((b + 7u) & 7u);
  if (MAP == 2) c = (b & ~63u) | ((b & 7u) << 3) | ((b >> 3) & 7u);
  if (MAP == 3) c = (b & ~511u) | ((b & 7u) << 6) | ((b >> 3) & 63u);
  if (MAP == 4) c = (b & ~7u) | ((b & 7u) ^ ((b >> 3) & 7u));
  const long long e = (long long)c * 256 + threadIdx.x;
  if (e < n4) __builtin_nontemporal_store((f32x4)(v), &out[e]);
}
// the box pattern with the library kernel's XCD-aware remap: XCD g (= blockIdx % 8) owns the contiguous box range g
template <int T, int U>
__global__ __launch_bounds__(T) void store_crop_xcd_kernel(f32x4 *out, int n, float v) {
  const int per = 7056;
  const int grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = n >> 3, r = n & 7;
  const int box = ((grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q) + slot;
  f32x4 *dst = out + (long long)box * per;
  f32x4 x = (f32x4)(v);
  int e = threadIdx.x;
  for (; e + (U - 1) * T < per; e += U * T) {
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x, &dst[e + u * T]);
  }
  for (; e < per; e += T) __builtin_nontemporal_store(x, &dst[e]);
}
// consecutive pieces of PER float4 (16 KB row = 1008, 4 KB tile = 256), each XCD owning one contiguous eighth of them
template <int PER>
__global__ __launch_bounds__(256) void store_piece_xcd_kernel(f32x4 *out, unsigned npieces, float v) {
  const unsigned grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned q = npieces >> 3, r = npieces & 7;
  const unsigned piece = ((grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q) + slot;
  f32x4 *dst = out + (long long)piece * PER;
  f32x4 x = (f32x4)(v);
  for (int e = threadIdx.x; e < PER; e += 256) __builtin_nontemporal_store(x, &dst[e]);
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
  float ms;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(store_crop_kernel<256, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(store_crop_kernel<1024, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(store_crop_kernel<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int per_cu : {1, 2, 3, 4, 6, 8}) {
    const int lds = 160 * 1024 / per_cu - 256;
    ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<256, 4>), dim3(n), dim3(256), lds, 0, out, 1.f); }, 30);
    printf("T256  U4, <=%d workgroups per CU   %8.2f us  %7.1f GB/s\n", per_cu, ms * 1e3, bytes / ms / 1e6);
  }
  for (int per_cu : {1, 2}) {
    const int lds = 160 * 1024 / per_cu - 256;
    ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<1024, 4>), dim3(n), dim3(1024), lds, 0, out, 1.f); }, 30);
    printf("T1024 U4, <=%d workgroups per CU   %8.2f us  %7.1f GB/s\n", per_cu, ms * 1e3, bytes / ms / 1e6);
  }
  for (int per_cu : {1, 2, 4, 8, 16}) {
    const int lds = 160 * 1024 / per_cu - 256;
    ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<64, 8>), dim3(n), dim3(64), lds, 0, out, 1.f); }, 30);
    printf("T64   U8, <=%d workgroups per CU   %8.2f us  %7.1f GB/s\n", per_cu, ms * 1e3, bytes / ms / 1e6);
  }
#define PACED(T, U) ms = time_ms([&] { hipLaunchKernelGGL((store_crop_paced_kernel<T, U>), dim3(n), dim3(T), 0, 0, out, 1.f); }, 30); \
  printf("T%-4d paced every %d stores            %8.2f us  %7.1f GB/s\n", T, U, ms * 1e3, bytes / ms / 1e6);
  PACED(256, 1) PACED(256, 2) PACED(256, 4) PACED(256, 8) PACED(1024, 1) PACED(1024, 2) PACED(64, 1) PACED(64, 4)
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_xcd_kernel<256, 4>), dim3(n), dim3(256), 0, 0, out, n, 1.f); }, 30);
  printf("box pattern, XCD-chunked box ranges     %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<256, 4>), dim3(n), dim3(256), 0, 0, out, 1.f); }, 30);
  printf("box pattern, box = blockIdx             %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_piece_xcd_kernel<1008>), dim3(n * 7), dim3(256), 0, 0, out, (unsigned)(n * 7), 1.f); }, 30);
  printf("16 KB rows, XCD-chunked ranges          %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_piece_xcd_kernel<256>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, out, (unsigned)(n4 / 256), 1.f); }, 30);
  printf("4 KB tiles, XCD-chunked ranges          %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_piece_xcd_kernel<512>), dim3((unsigned)(n4 / 512)), dim3(256), 0, 0, out, (unsigned)(n4 / 512), 1.f); }, 30);
  printf("8 KB tiles, XCD-chunked ranges          %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
#define CMAP(M) ms = time_ms([&] { hipLaunchKernelGGL((store_chunk_map_kernel<M>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, out, n4, 1.f); }, 30); \
  printf("4 KB chunks, XCD map %d                 %8.2f us  %7.1f GB/s\n", M, ms * 1e3, bytes / ms / 1e6);
  CMAP(0) CMAP(1) CMAP(2) CMAP(3) CMAP(4) CMAP(0)
#define GSP(G, U) ms = time_ms([&] { hipLaunchKernelGGL((store_gs_paced_kernel<U>), dim3(G), dim3(256), 0, 0, out, n4, 1.f); }, 30); \
  printf("grid-stride %5d blocks, paced every %d  %8.2f us  %7.1f GB/s\n", G, U, ms * 1e3, bytes / ms / 1e6);
  GSP(2048, 1) GSP(2048, 2) GSP(4096, 1) GSP(8192, 1) GSP(1024, 1) GSP(1024, 4)
  return 0;
}
