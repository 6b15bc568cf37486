// crop_bench2.hip -- A/B of crop_and_resize work decompositions whose WRITE FRONT is compact (diagnostic; not part of the
// library).  crop_bench.hip showed that the store pattern alone moves the sustained ceiling from 5.3 TB/s (one 110 KB
// region per workgroup) to 6.9 TB/s (4 KB per workgroup, workgroups in address order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/crop_bench2.hip -o tools/diag/crop_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "../tf-faster-rcnn_amd/csrc/crop_resize.hip"
#include "../tf-faster-rcnn_amd/csrc/api.hip"

// ---- pre-pass: sampling parameters of every output pixel, offsets relative to the whole feature tensor ----------------
__global__ __launch_bounds__(256) void params_kernel(int B, int H, int W, int C, const float *__restrict__ boxes, int n,
                                                     int boxes_per_image, int ch, int cw, PixParam *__restrict__ params) {
  const int gp = blockIdx.x * 256 + threadIdx.x;
  const int P = ch * cw;
  if (gp >= n * P) return;
  const int box = gp / P, p = gp - box * P;
  const int bimg = box / boxes_per_image;
  const bool img_ok = bimg >= 0 && bimg < B;
  const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + (long long)box * 4);
  PixParam prm = pix_param(p, H, W, (unsigned)C * 4u, ch, cw, bx.x, bx.y, bx.z, bx.w, img_ok);
  const unsigned ib = (unsigned)(img_ok ? bimg : 0) * (unsigned)(H * W * C) * 4u;
  prm.off_tl += ib;
  prm.off_tr += ib;
  prm.off_bl += ib;
  prm.off_br += ib;
  params[gp] = prm;
}

// ---- tile kernel: workgroup = 256*U consecutive float4 of the output, workgroups in address order ----------------------
// MAP 0: tile = blockIdx; MAP 1: each XCD (blockIdx % 8) owns one contiguous eighth of the tiles.
template <int CV, int U, int MAP, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void crop_tile_kernel(const char *__restrict__ image, const PixParam *__restrict__ params, f32x4 *__restrict__ out,
                      unsigned n4, unsigned ntiles, float extrap) {
  unsigned tile = blockIdx.x;
  if (MAP == 1) {
    const unsigned grp = tile & 7, slot = tile >> 3;
    const unsigned q = ntiles >> 3, r = ntiles & 7;
    tile = ((grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q) + slot;
  }
  const unsigned e0 = tile * (256u * U) + threadIdx.x;
  f32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    unsigned e = e0 + u * 256u;
    e = e < n4 ? e : n4 - 1;
    const unsigned gp = e / CV, c = e - gp * CV;
    const unsigned gpw = __builtin_amdgcn_readfirstlane(gp);
    const PixParam *pa = params + gpw;  // wave-uniform: scalar loads
    const PixParam a = pa[0], b = pa[1];
    const bool second = gp != gpw;      // CV >= 64: a wave spans at most two pixels
    PixParam prm;
    prm.off_tl = second ? b.off_tl : a.off_tl;
    prm.off_tr = second ? b.off_tr : a.off_tr;
    prm.off_bl = second ? b.off_bl : a.off_bl;
    prm.off_br = second ? b.off_br : a.off_br;
    prm.lx = second ? b.lx : a.lx;
    prm.ly = second ? b.ly : a.ly;
    prm.valid = second ? b.valid : a.valid;
    v[u] = bilerp<f32x4>(image, prm, c * 16u, extrap);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned e = e0 + u * 256u;
    if (e < n4) __builtin_nontemporal_store(v[u], &out[e]);
  }
}

// ---- persistent tile kernel: 8 * J workgroups; XCD x (blockIdx % 8) owns a contiguous eighth of the output; its J
// workgroups walk it together: in round i workgroup j handles the U tiles ((i*U + u)*J + j) of 256 float4 (4 KB), so the
// XCD's write front is U*J*4 KB wide and moves in address order.  Parameters are prefetched one round ahead.
template <int CV, int U, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void crop_ptile_kernel(const char *__restrict__ image, const PixParam *__restrict__ params, f32x4 *__restrict__ out,
                       unsigned n4, float extrap) {
  const unsigned x = blockIdx.x & 7, j = blockIdx.x >> 3, J = gridDim.x >> 3;
  const unsigned tiles_all = (n4 + 255u) / 256u;
  const unsigned q = tiles_all >> 3, r = tiles_all & 7;
  const unsigned t0 = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;  // first tile of this XCD
  const unsigned nt = q + (x < r ? 1u : 0u);
  const unsigned rounds = (nt + U * J - 1) / (U * J);
  auto elem = [&](unsigned i, int u) -> unsigned {
    const unsigned t = (i * U + u) * J + j;
    return t < nt ? (t0 + t) * 256u + threadIdx.x : 0xFFFFFFFFu;
  };
  // wave-uniform parameters (scalar loads): a wave's 64 float4 span at most two pixels (CV >= 64)
  PixParam na[U], nb[U];
  auto load_params = [&](unsigned i, PixParam *da, PixParam *db) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned e = elem(i, u);
      e = e < n4 ? e : 0u;
      const unsigned gpw = __builtin_amdgcn_readfirstlane(e / CV);
      da[u] = params[gpw];
      db[u] = params[gpw + 1];
    }
  };
  load_params(0, na, nb);
  for (unsigned i = 0; i < rounds; ++i) {
    PixParam ca[U], cb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ca[u] = na[u];
      cb[u] = nb[u];
    }
    if (i + 1 < rounds) load_params(i + 1, na, nb);
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned e = elem(i, u);
      e = e < n4 ? e : 0u;
      const unsigned gp = e / CV, c = e - gp * CV;
      const bool second = gp != (unsigned)__builtin_amdgcn_readfirstlane(gp);
      PixParam prm;
      prm.off_tl = second ? cb[u].off_tl : ca[u].off_tl;
      prm.off_tr = second ? cb[u].off_tr : ca[u].off_tr;
      prm.off_bl = second ? cb[u].off_bl : ca[u].off_bl;
      prm.off_br = second ? cb[u].off_br : ca[u].off_br;
      prm.lx = second ? cb[u].lx : ca[u].lx;
      prm.ly = second ? cb[u].ly : ca[u].ly;
      prm.valid = second ? cb[u].valid : ca[u].valid;
      v[u] = bilerp<f32x4>(image, prm, c * 16u, extrap);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned e = elem(i, u);
      if (e < n4) __builtin_nontemporal_store(v[u], &out[e]);
    }
  }
}

// ---- row kernel: workgroup = (box, output row): cw*CV contiguous float4 (16 KB at 7 x 576) ----------------------------
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void crop_row_kernel(const float *__restrict__ image, int B, int H, int W, int C, const float *__restrict__ boxes, int n,
                     int boxes_per_image, int ch, int cw, float extrap, float *__restrict__ out) {
  __shared__ PixParam pp[64];
  const int tid = threadIdx.x;
  const unsigned nrows = (unsigned)n * ch;
  const unsigned grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned q = nrows >> 3, r = nrows & 7;
  const unsigned row = ((grp < r) ? grp * (q + 1) : r * (q + 1) + (grp - r) * q) + slot;
  const int box = row / ch, y = row - box * ch;
  const int bimg = box / boxes_per_image;
  const bool img_ok = bimg >= 0 && bimg < B;
  const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + (long long)box * 4);
  if (tid < cw) pp[tid] = pix_param(y * cw + tid, H, W, (unsigned)C * 4u, ch, cw, bx.x, bx.y, bx.z, bx.w, img_ok);
  __syncthreads();
  const int CV = C >> 2;
  const int total = cw * CV;
  const char *__restrict__ base = reinterpret_cast<const char *>(image + (long long)(img_ok ? bimg : 0) * H * W * C);
  char *__restrict__ dstb = reinterpret_cast<char *>(out + (long long)row * cw * C);
  int p = tid / CV;
  int c = tid - p * CV;
  const int dp = 256 / CV, dc = 256 - dp * CV;
  f32x4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool in = tid + u * 256 < total;
    const PixParam prm = pp[in ? p : 0];
    v[u] = bilerp<f32x4>(base, prm, (unsigned)(in ? c : 0) * 16u, extrap);
    c += dc;
    p += dp;
    if (c >= CV) {
      c -= CV;
      ++p;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + u * 256;
    if (e < total) __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4 *>(dstb + (size_t)e * 16));
  }
}


// ---- LDS-deduplicated chunk kernel: workgroup = (box, QC float4 channels) --------------------------------------------
// The 4*ch*cw taps of a box are the cross product rows_used x cols_used (<= 2ch x 2cw, about 128 of 196 distinct on
// RoI-like boxes).  Each distinct tap goes through the vector-memory path once (global -> LDS), the four taps of an
// output pixel are ds_read_b128 (256 B/clk/CU against the 64 B/clk/CU of the texture-addresser path).
template <int QC>
__global__ __launch_bounds__(256) void crop_lds_kernel(const float *__restrict__ image, int B, int H, int W, int C,
                                                       const float *__restrict__ boxes, int n, int boxes_per_image, int ch,
                                                       int cw, float extrap, float *__restrict__ out, int nchunk) {
  __shared__ f32x4 tile[14 * 14 * QC];
  __shared__ int t_lo[32], t_hi[32], t_ok[32];  // [0,16): rows by y, [16,32): cols by x -- slots into the tile
  __shared__ float t_fr[32];
  __shared__ int t_list[32];                    // slot -> source row / col
  __shared__ int t_cnt[2];
  const int tid = threadIdx.x;
  const unsigned nwg = (unsigned)n * nchunk;
  const unsigned grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned q8 = nwg >> 3, r8 = nwg & 7;
  const unsigned wg = ((grp < r8) ? grp * (q8 + 1) : r8 * (q8 + 1) + (grp - r8) * q8) + slot;
  const int box = wg / nchunk, chunk = wg - box * nchunk;
  const int bimg = box / boxes_per_image;
  const bool img_ok = bimg >= 0 && bimg < B;
  if (tid < 64) {
    const f32x4 bx = *reinterpret_cast<const f32x4 *>(boxes + (long long)box * 4);
    const bool isx = (tid & 16) != 0;
    const int i = tid & 15;
    const int cn = isx ? cw : ch;
    const float a1 = isx ? bx.y : bx.x, a2 = isx ? bx.w : bx.z;
    const float lim = (float)((isx ? W : H) - 1);
    float in;
    if (cn > 1) {
      float t = a2 - a1;
      t = t * lim;
      const float sc = t / (float)(cn - 1);
      const float t0 = a1 * lim;
      const float t1 = (float)i * sc;
      in = t0 + t1;
    } else {
      float t0 = a1 + a2;
      t0 = 0.5f * t0;
      in = t0 * lim;
    }
    const bool ok = (tid < 32) && (i < cn) && img_ok && (in >= 0.0f) && (in <= lim);
    const int lo = ok ? (int)floorf(in) : 0, hi = ok ? (int)ceilf(in) : 0;
    const float fr = ok ? in - (float)lo : 0.0f;
    unsigned long long m = ok ? ((1ull << lo) | (1ull << hi)) : 0ull;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const unsigned lo32 = __shfl_xor((unsigned)m, off, 64), hi32 = __shfl_xor((unsigned)(m >> 32), off, 64);
      m |= ((unsigned long long)hi32 << 32) | lo32;
    }
    if (tid < 32) {
      t_lo[tid] = __popcll(m & ((1ull << lo) - 1ull));
      t_hi[tid] = __popcll(m & ((1ull << hi) - 1ull));
      t_ok[tid] = ok;
      t_fr[tid] = fr;
    }
    const unsigned long long rmask = ((unsigned long long)__shfl((unsigned)(m >> 32), 0, 64) << 32) | __shfl((unsigned)m, 0, 64);
    const unsigned long long cmask = ((unsigned long long)__shfl((unsigned)(m >> 32), 16, 64) << 32) | __shfl((unsigned)m, 16, 64);
    if ((rmask >> tid) & 1ull) t_list[__popcll(rmask & ((1ull << tid) - 1ull))] = tid;
    if ((cmask >> tid) & 1ull) t_list[16 + __popcll(cmask & ((1ull << tid) - 1ull))] = tid;
    if (tid == 0) {
      t_cnt[0] = __popcll(rmask);
      t_cnt[1] = __popcll(cmask);
    }
  }
  __syncthreads();
  const int nr = t_cnt[0], nc = t_cnt[1];
  const int CV = C >> 2;
  const int c0 = chunk * QC;
  const int cq = tid & (QC - 1);
  const bool c_in = c0 + cq < CV;
  const f32x4 *__restrict__ img = reinterpret_cast<const f32x4 *>(image + (long long)(img_ok ? bimg : 0) * H * W * C) + c0 + cq;
  const int ntap = nr * nc;
  for (int t = tid / QC; t < ntap; t += 256 / QC) {
    const int ri = t / nc, ci = t - ri * nc;
    const int row = t_list[ri], col = t_list[16 + ci];
    if (c_in) tile[t * QC + cq] = img[(row * W + col) * CV];
  }
  __syncthreads();
  const int P = ch * cw;
  f32x4 *__restrict__ dst = reinterpret_cast<f32x4 *>(out + (long long)box * P * C) + c0 + cq;
  for (int p = tid / QC; p < P; p += 256 / QC) {
    const int y = p / cw, x = p - y * cw;
    f32x4 v = (f32x4)(extrap);
    if (t_ok[y] && t_ok[16 + x]) {
      const int rt = t_lo[y] * nc, rb = t_hi[y] * nc, cl = t_lo[16 + x], cr = t_hi[16 + x];
      const float ly = t_fr[y], lx = t_fr[16 + x];
      const f32x4 tl = tile[(rt + cl) * QC + cq], tr = tile[(rt + cr) * QC + cq];
      const f32x4 bl = tile[(rb + cl) * QC + cq], br = tile[(rb + cr) * QC + cq];
      f32x4 t = vsub(tr, tl);
      t = t * lx;
      const f32x4 tp = tl + t;
      t = vsub(br, bl);
      t = t * lx;
      const f32x4 bt = bl + t;
      t = vsub(bt, tp);
      t = t * ly;
      v = tp + t;
    }
    if (c_in) __builtin_nontemporal_store(v, &dst[p * CV]);
  }
}

// ---- x-deduplicated kernel: lane = one float4 of channels for a whole output row, taps shared along x in registers -----
// The 2*cw column taps of a source row are left_0, right_0, left_1, ... ; consecutive entries are often the same column
// (right_x == left_x+1's left, or an integral in_x).  Which ones is workgroup-uniform, so a lane loads a tap once and
// keeps it for the next output pixel: scalar branches, no LDS, no barrier.
template <int CW, int XB>
__global__ __launch_bounds__(256) void crop_xdedupe_kernel(const float *__restrict__ image, int B, int H, int W, int C,
                                                           const float *__restrict__ boxes, int n, int boxes_per_image,
                                                           int ch, float extrap, float *__restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const unsigned q8 = (unsigned)n >> 3, r8 = (unsigned)n & 7;
  const int box = (int)(((grp < r8) ? grp * (q8 + 1) : r8 * (q8 + 1) + (grp - r8) * q8) + slot);
  const int bimg = box / boxes_per_image;
  const bool img_ok = bimg >= 0 && bimg < B;
  const float y1 = boxes[(long long)box * 4 + 0], x1 = boxes[(long long)box * 4 + 1];
  const float y2 = boxes[(long long)box * 4 + 2], x2 = boxes[(long long)box * 4 + 3];
  const float hm1 = (float)(H - 1), wm1 = (float)(W - 1);
  // x parameters: lane i computes output column i, then everything is broadcast into SGPRs
  float in_x;
  {
    if (CW > 1) {
      float t = x2 - x1;
      t = t * wm1;
      const float ws = t / (float)(CW - 1);
      const float t0 = x1 * wm1;
      const float t1 = (float)lane * ws;
      in_x = t0 + t1;
    } else {
      float t0 = x1 + x2;
      t0 = 0.5f * t0;
      in_x = t0 * wm1;
    }
  }
  const bool vx_l = img_ok && (in_x >= 0.0f) && (in_x <= wm1);
  const int left_l = vx_l ? (int)floorf(in_x) : 0, right_l = vx_l ? (int)ceilf(in_x) : 0;
  const float lx_l = vx_l ? in_x - (float)left_l : 0.0f;
  int col[2 * CW];
  float lx[CW];
  bool vx[CW];
  bool ld[2 * CW];
#pragma unroll
  for (int x = 0; x < CW; ++x) {
    col[2 * x] = __builtin_amdgcn_readlane(left_l, x);
    col[2 * x + 1] = __builtin_amdgcn_readlane(right_l, x);
    lx[x] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lx_l), x));
    vx[x] = __builtin_amdgcn_readlane((int)vx_l, x) != 0;
  }
#pragma unroll
  for (int k = 0; k < 2 * CW; ++k) {
    const bool valid = vx[k >> 1];
    const bool prev_valid = k > 0 && vx[(k - 1) >> 1];
    ld[k] = valid && !(prev_valid && col[k] == col[k - 1]);
  }
  const int CV = C >> 2;
  const unsigned pixb = (unsigned)C * 4u;
  const char *__restrict__ img = reinterpret_cast<const char *>(image + (long long)(img_ok ? bimg : 0) * H * W * C);
  f32x4 *__restrict__ dst = reinterpret_cast<f32x4 *>(out + (long long)box * ch * CW * C);
  for (int c = tid; c < CV; c += blockDim.x) {
    for (int y = 0; y < ch; ++y) {
      float in_y;
      if (ch > 1) {
        float t = y2 - y1;
        t = t * hm1;
        const float hs = t / (float)(ch - 1);
        const float t0 = y1 * hm1;
        const float t1 = (float)y * hs;
        in_y = t0 + t1;
      } else {
        float t0 = y1 + y2;
        t0 = 0.5f * t0;
        in_y = t0 * hm1;
      }
      const bool vy = __builtin_amdgcn_readfirstlane((int)(img_ok && (in_y >= 0.0f) && (in_y <= hm1))) != 0;
      f32x4 *__restrict__ drow = dst + (long long)y * CW * CV + c;
      if (!vy) {
#pragma unroll
        for (int x = 0; x < CW; ++x) __builtin_nontemporal_store((f32x4)(extrap), drow + x * CV);
        continue;
      }
      const int top = __builtin_amdgcn_readfirstlane((int)floorf(in_y));
      const int bot = __builtin_amdgcn_readfirstlane((int)ceilf(in_y));
      const float ly = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(in_y - (float)top)));
      const char *__restrict__ trow = img + (size_t)((unsigned)(top * W) * pixb);
      const char *__restrict__ brow = img + (size_t)((unsigned)(bot * W) * pixb);
      const bool two_rows = bot != top;
      f32x4 carryT = (f32x4)(0.0f), carryB = (f32x4)(0.0f);  // last tap of the previous block of pixels
#pragma unroll
      for (int x0 = 0; x0 < CW; x0 += XB) {
        constexpr int NE = 2 * XB;
        f32x4 T[NE], Bt[NE];
        // phase 1: every distinct tap of the block is issued (no use of a loaded value in between)
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int k = 2 * x0 + j;
          T[j] = (f32x4)(0.0f);
          Bt[j] = (f32x4)(0.0f);
          if (k < 2 * CW) {
            if (ld[k]) T[j] = *reinterpret_cast<const f32x4 *>(trow + (size_t)((unsigned)col[k] * pixb) + (size_t)c * 16);
            if (ld[k] && two_rows) Bt[j] = *reinterpret_cast<const f32x4 *>(brow + (size_t)((unsigned)col[k] * pixb) + (size_t)c * 16);
          }
        }
        // phase 2: a repeated column takes its neighbour's registers
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int k = 2 * x0 + j;
          if (k < 2 * CW && !ld[k]) {
            T[j] = j > 0 ? T[j - 1] : carryT;
            Bt[j] = j > 0 ? Bt[j - 1] : carryB;
          }
        }
        if (!two_rows) {
#pragma unroll
          for (int j = 0; j < NE; ++j) Bt[j] = T[j];
        }
        carryT = T[NE - 1];
        carryB = Bt[NE - 1];
#pragma unroll
        for (int xx = 0; xx < XB; ++xx) {
          const int x = x0 + xx;
          if (x < CW) {
            const f32x4 tl = T[2 * xx], tr = T[2 * xx + 1], bl = Bt[2 * xx], br = Bt[2 * xx + 1];
            f32x4 t = vsub(tr, tl);
            t = t * lx[x];
            const f32x4 tp = tl + t;
            t = vsub(br, bl);
            t = t * lx[x];
            const f32x4 bt = bl + t;
            t = vsub(bt, tp);
            t = t * ly;
            f32x4 v = tp + t;
            if (!vx[x]) v = (f32x4)(extrap);
            __builtin_nontemporal_store(v, drow + x * CV);
          }
        }
      }
    }
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

int main(int argc, char **argv) {
  const int B = 16, H = 32, W = 32, C = 576, R = 300, n = B * R;
  const long long out_floats = (long long)n * 49 * C;
  float *img, *boxes, *out, *ref;
  PixParam *params;
  CK(hipMalloc(&img, (size_t)B * H * W * C * 4));
  CK(hipMalloc(&boxes, (size_t)n * 16));
  CK(hipMalloc(&out, (size_t)out_floats * 4));
  CK(hipMalloc(&ref, (size_t)out_floats * 4));
  CK(hipMalloc(&params, (size_t)(n * 49 + 1) * sizeof(PixParam)));
  CK(hipMemset(params, 0, (size_t)(n * 49 + 1) * sizeof(PixParam)));
  std::vector<float> hb(n * 4), hi((size_t)B * H * W * C);
  srand(1);
  // RoI-like boxes: sizes 0.15..0.7 of the image, some touching the border, a few zero-padded rows
  for (int i = 0; i < n; ++i) {
    float cy = rand() / (float)RAND_MAX, cx = rand() / (float)RAND_MAX;
    float h = 0.15f + 0.55f * (rand() / (float)RAND_MAX), w = 0.15f + 0.55f * (rand() / (float)RAND_MAX);
    hb[i * 4 + 0] = std::max(0.f, cy - h / 2); hb[i * 4 + 2] = std::min(1.f, cy + h / 2);
    hb[i * 4 + 1] = std::max(0.f, cx - w / 2); hb[i * 4 + 3] = std::min(1.f, cx + w / 2);
    if (i % 97 == 0) hb[i * 4 + 0] = hb[i * 4 + 1] = hb[i * 4 + 2] = hb[i * 4 + 3] = 0.f;
  }
  for (auto &v : hi) v = rand() / (float)RAND_MAX;
  CK(hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(img, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
  const double bytes = (double)out_floats * 4;
  const double alg = bytes + (double)B * H * W * C * 4 + n * 20.0;
  const int iters = 30;
  const unsigned n4 = (unsigned)(out_floats / 4);
  std::vector<float> h_ref((size_t)out_floats), h_out((size_t)out_floats);

  frcnn_crop_and_resize(img, B, H, W, C, boxes, nullptr, n, R, 7, 7, 0.f, ref, nullptr);
  CK(hipMemcpy(h_ref.data(), ref, (size_t)out_floats * 4, hipMemcpyDeviceToHost));
  auto check = [&](const char *name) {
    CK(hipMemcpy(h_out.data(), out, (size_t)out_floats * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(h_out.data(), h_ref.data(), (size_t)out_floats * 4) == 0;
    printf("   %s: %s\n", name, same ? "bit-identical to the library kernel" : "MISMATCH");
    CK(hipMemset(out, 0xff, (size_t)out_floats * 4));
  };
  float ms;
  ms = time_ms([&] { frcnn_crop_and_resize(img, B, H, W, C, boxes, nullptr, n, R, 7, 7, 0.f, out, nullptr); }, iters);
  printf("library box kernel                 %8.2f us  %7.1f GB/s\n", ms * 1e3, alg / ms / 1e6);
  check("library");

  auto prepass = [&] {
    hipLaunchKernelGGL(params_kernel, dim3((n * 49 + 255) / 256), dim3(256), 0, 0, B, H, W, C, boxes, n, R, 7, 7, params);
  };
  ms = time_ms(prepass, iters);
  printf("params pre-pass                    %8.2f us\n", ms * 1e3);

#define PT(U, WPE, J) { \
    ms = time_ms([&] { prepass(); hipLaunchKernelGGL((crop_ptile_kernel<144, U, WPE>), dim3(8 * J), dim3(256), 0, 0, (const char *)img, params, (f32x4 *)out, n4, 0.f); }, iters); \
    printf("persistent tiles U=%d wpe=%d J=%d (pre-pass included)   %8.2f us  %7.1f GB/s\n", U, WPE, J, ms * 1e3, alg / ms / 1e6); \
    check("persistent tiles"); }
  PT(4, 4, 128) PT(2, 4, 128) PT(4, 4, 64) PT(2, 8, 256) PT(4, 6, 192) PT(1, 8, 256) PT(4, 8, 256) PT(2, 6, 192)
#define TILE(U, MAP, WPE) { \
    ms = time_ms([&] { prepass(); hipLaunchKernelGGL((crop_tile_kernel<144, U, MAP, WPE>), dim3((n4 + 256 * U - 1) / (256 * U)), dim3(256), 0, 0, (const char *)img, params, (f32x4 *)out, n4, (n4 + 256 * U - 1) / (256 * U), 0.f); }, iters); \
    printf("one tile per workgroup U=%d map=%d wpe=%d (pre-pass included) %8.2f us  %7.1f GB/s\n", U, MAP, WPE, ms * 1e3, alg / ms / 1e6); \
    check("tile"); }
  TILE(4, 1, 4) TILE(1, 0, 8) TILE(4, 1, 6) TILE(4, 1, 8) TILE(2, 1, 8) TILE(4, 0, 8)
#define XDD(NT, XB) { \
    ms = time_ms([&] { hipLaunchKernelGGL((crop_xdedupe_kernel<7, XB>), dim3(n), dim3(NT), 0, 0, img, B, H, W, C, boxes, n, R, 7, 0.f, out); }, iters); \
    printf("x-dedupe kernel NT=%d XB=%d             %8.2f us  %7.1f GB/s\n", NT, XB, ms * 1e3, alg / ms / 1e6); \
    check("x-dedupe"); }
  XDD(192, 4)
#define ROW(WPE) { \
    ms = time_ms([&] { hipLaunchKernelGGL((crop_row_kernel<WPE>), dim3(n * 7), dim3(256), 0, 0, img, B, H, W, C, boxes, n, R, 7, 7, 0.f, out); }, iters); \
    printf("row kernel wpe=%d                   %8.2f us  %7.1f GB/s\n", WPE, ms * 1e3, alg / ms / 1e6); \
    check("row"); }
  ROW(6)
  return 0;
}
