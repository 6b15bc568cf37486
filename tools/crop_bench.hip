// crop_bench.hip -- standalone A/B harness for the crop_and_resize kernel against pure-store ceilings
// measured on the same device (diagnostic; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/crop_bench.hip -o tools/diag/crop_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#include "../tf-faster-rcnn_amd/csrc/crop_resize.hip"
#include "../tf-faster-rcnn_amd/csrc/api.hip"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void store_kernel(f32x4 *out, long long n4, float v) {
  // same shape as the crop kernel: one block per 7056-float4 crop
  const long long per = 7056;
  f32x4 *dst = out + (long long)blockIdx.x * per;
  f32x4 x = (f32x4)(v);
  for (int e = threadIdx.x; e < per; e += 256) {
    if (NT) __builtin_nontemporal_store(x, &dst[e]); else dst[e] = x;
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void store_gs_kernel(f32x4 *out, long long n4, float v) {
  f32x4 x = (f32x4)(v);
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, &out[e]); else out[e] = x;
  }
}

// variant: block per crop, T threads, U stores in flight per thread
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void store_crop_kernel(f32x4 *out, float v) {
  const int per = 7056;
  f32x4 *dst = out + (long long)blockIdx.x * per;
  f32x4 x = (f32x4)(v);
  int e = threadIdx.x;
  for (; e + (U - 1) * T < per; e += U * T) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(x, &dst[e + u * T]); else dst[e + u * T] = x;
    }
  }
  for (; e < per; e += T) {
    if (NT) __builtin_nontemporal_store(x, &dst[e]); else dst[e] = x;
  }
}

// variant: linear sweep like a memset: block b writes [b*chunk, (b+1)*chunk) float4, chunk small, blocks in order
template <int T, int PER_THREAD>
__global__ __launch_bounds__(T) void store_linear_kernel(f32x4 *out, long long n4, float v) {
  f32x4 x = (f32x4)(v);
  long long base = (long long)blockIdx.x * T * PER_THREAD;
#pragma unroll
  for (int u = 0; u < PER_THREAD; ++u) {
    long long e = base + (long long)u * T + threadIdx.x;
    if (e < n4) out[e] = x;
  }
}

// variant: block = (crop, channel chunk of QC float4): 49 segments of QC*16 B strided by 2304 B; chunks of one crop are
// consecutive blocks (optionally XCD-aware: consecutive crops stay on one XCD)
template <int QC, bool XCD>
__global__ __launch_bounds__(256) void store_chunk_kernel(f32x4 *out, int n, float v) {
  const int nchunk = 144 / QC;
  int bid = blockIdx.x;
  int box, chunk;
  if (XCD) {
    const int grp = bid & 7, slot = bid >> 3;
    const int q = n >> 3, r = n & 7;
    const int start = (grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q;
    box = start + slot / nchunk;
    chunk = slot % nchunk;
  } else {
    box = bid / nchunk;
    chunk = bid % nchunk;
  }
  f32x4 x = (f32x4)(v);
  const int q = threadIdx.x % QC;
  for (int p = threadIdx.x / QC; p < 49; p += 256 / QC) {
    __builtin_nontemporal_store(x, &out[((long long)box * 49 + p) * 144 + chunk * QC + q]);
  }
}

// variant: block = (crop, output row): 1008 contiguous float4
template <bool NT>
__global__ __launch_bounds__(256) void store_row_kernel(f32x4 *out, float v) {
  f32x4 x = (f32x4)(v);
  f32x4 *dst = out + (long long)blockIdx.x * 1008;
  for (int e = threadIdx.x; e < 1008; e += 256) {
    if (NT) __builtin_nontemporal_store(x, &dst[e]); else dst[e] = x;
  }
}

// variant: tile of T*U float4 per block; each WAVE writes U consecutive KB (wave-contiguous), vs store_linear_kernel where
// a wave's stores are T*16 B apart
template <int T, int U, bool NT>
__global__ __launch_bounds__(T) void store_wavecontig_kernel(f32x4 *out, long long n4, float v) {
  f32x4 x = (f32x4)(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long base = (long long)blockIdx.x * T * U + (long long)wave * 64 * U + lane;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    long long e = base + u * 64;
    if (e < n4) { if (NT) __builtin_nontemporal_store(x, &out[e]); else out[e] = x; }
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

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

int main(int argc, char **argv) {
  const int B = 16, H = 32, W = 32, C = 576, R = 300, n = B * R;
  const long long out_floats = (long long)n * 49 * C;
  float *img, *boxes, *out;
  CK(hipMalloc(&img, (size_t)B * H * W * C * 4));
  CK(hipMalloc(&boxes, (size_t)n * 16));
  CK(hipMalloc(&out, (size_t)out_floats * 4));
  std::vector<float> hb(n * 4), hi((size_t)B * H * W * C);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float a = rand() / (float)RAND_MAX, b = rand() / (float)RAND_MAX, c = rand() / (float)RAND_MAX, d = rand() / (float)RAND_MAX;
    hb[i * 4 + 0] = std::min(a, c); hb[i * 4 + 2] = std::max(a, c);
    hb[i * 4 + 1] = std::min(b, d); hb[i * 4 + 3] = std::max(b, d);
  }
  for (auto &v : hi) v = rand() / (float)RAND_MAX;
  CK(hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  const double bytes = (double)out_floats * 4;
  const double alg = bytes + (double)B * H * W * C * 4 + n * 20.0;
  const int iters = 30;
  float ms;
  ms = time_ms([&] { frcnn_crop_and_resize(img, B, H, W, C, boxes, nullptr, n, R, 7, 7, 0.f, out, nullptr); }, iters);
  printf("crop_and_resize (library)       %8.2f us  %7.1f GB/s algorithmic\n", ms * 1e3, alg / ms / 1e6);
  if (argc > 1) return 0;  // quick mode: the library kernel only (PMC passes)
  ms = time_ms([&] { hipLaunchKernelGGL(store_kernel<true>, dim3(n), dim3(256), 0, 0, (f32x4 *)out, out_floats / 4, 1.f); }, iters);
  printf("store, block per crop, nt       %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL(store_kernel<false>, dim3(n), dim3(256), 0, 0, (f32x4 *)out, out_floats / 4, 1.f); }, iters);
  printf("store, block per crop, plain    %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  for (int g : {1024, 2048, 4096, 8192}) {
    ms = time_ms([&] { hipLaunchKernelGGL(store_gs_kernel<true>, dim3(g), dim3(256), 0, 0, (f32x4 *)out, out_floats / 4, 1.f); }, iters);
    printf("store, grid-stride %5d, nt     %8.2f us  %7.1f GB/s\n", g, ms * 1e3, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(store_gs_kernel<false>, dim3(g), dim3(256), 0, 0, (f32x4 *)out, out_floats / 4, 1.f); }, iters);
    printf("store, grid-stride %5d, plain  %8.2f us  %7.1f GB/s\n", g, ms * 1e3, bytes / ms / 1e6);
  }
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<256, 4, false>), dim3(n), dim3(256), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store crop T256 U4 plain        %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<512, 4, false>), dim3(n), dim3(512), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store crop T512 U4 plain        %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<1024, 4, false>), dim3(n), dim3(1024), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store crop T1024 U4 plain       %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_crop_kernel<64, 8, false>), dim3(n), dim3(64), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store crop T64 U8 plain         %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  {
    const long long n4 = out_floats / 4;
    ms = time_ms([&] { hipLaunchKernelGGL((store_linear_kernel<256, 1>), dim3((n4 + 255) / 256), dim3(256), 0, 0, (f32x4 *)out, n4, 1.f); }, iters);
    printf("store linear T256 x1            %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((store_linear_kernel<256, 4>), dim3((n4 + 1023) / 1024), dim3(256), 0, 0, (f32x4 *)out, n4, 1.f); }, iters);
    printf("store linear T256 x4            %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((store_linear_kernel<256, 16>), dim3((n4 + 4095) / 4096), dim3(256), 0, 0, (f32x4 *)out, n4, 1.f); }, iters);
    printf("store linear T256 x16           %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((store_linear_kernel<1024, 4>), dim3((n4 + 4095) / 4096), dim3(1024), 0, 0, (f32x4 *)out, n4, 1.f); }, iters);
    printf("store linear T1024 x4           %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  }
  ms = time_ms([&] { hipLaunchKernelGGL((store_chunk_kernel<16, false>), dim3(n * 9), dim3(256), 0, 0, (f32x4 *)out, n, 1.f); }, iters);
  printf("store chunk 16q linear          %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_chunk_kernel<16, true>), dim3(n * 9), dim3(256), 0, 0, (f32x4 *)out, n, 1.f); }, iters);
  printf("store chunk 16q xcd             %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_chunk_kernel<8, false>), dim3(n * 18), dim3(256), 0, 0, (f32x4 *)out, n, 1.f); }, iters);
  printf("store chunk 8q linear           %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_chunk_kernel<48, false>), dim3(n * 3), dim3(256), 0, 0, (f32x4 *)out, n, 1.f); }, iters);
  printf("store chunk 48q linear          %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_row_kernel<true>), dim3(n * 7), dim3(256), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store row (16 KB) nt            %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  ms = time_ms([&] { hipLaunchKernelGGL((store_row_kernel<false>), dim3(n * 7), dim3(256), 0, 0, (f32x4 *)out, 1.f); }, iters);
  printf("store row (16 KB) plain         %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  {
    const long long n4 = out_floats / 4;
#define WC(T, U, NT) ms = time_ms([&] { hipLaunchKernelGGL((store_wavecontig_kernel<T, U, NT>), dim3((n4 + T * U - 1) / (T * U)), dim3(T), 0, 0, (f32x4 *)out, n4, 1.f); }, iters); \
    printf("store wave-contig T%d x%d nt=%d      %8.2f us  %7.1f GB/s\n", T, U, (int)NT, ms * 1e3, bytes / ms / 1e6);
    WC(256, 1, false) WC(256, 1, true) WC(64, 1, false) WC(128, 1, false) WC(512, 1, false) WC(1024, 1, false)
    WC(256, 2, false) WC(256, 4, false) WC(256, 4, true) WC(256, 8, false) WC(64, 4, false) WC(64, 16, false) WC(1024, 4, false)
  }
  ms = time_ms([&] { CK(hipMemsetAsync(out, 0, (size_t)bytes, 0)); }, iters);
  printf("hipMemsetAsync                  %8.2f us  %7.1f GB/s\n", ms * 1e3, bytes / ms / 1e6);
  return 0;
}
