// block_primitives.hpp -- workgroup-cooperative building blocks (NT threads = NT/64 waves of 64; 1024 unless the caller
// names another block size as the first template argument):
// exclusive scan, wave-aggregated LDS histogram add, 4x8-bit radix SELECT and the stable LSD radix
// SORT of 16-bit slot numbers.  Used by sort_select.hip, proposal.hip and rpn_targets.hip.
//
// All functions must be called by every thread of the block (they contain __syncthreads()).
#pragma once

#include "common.hpp"

namespace frcnn {

static const int kNT = 1024;
static const int kNW = 16;

// Exclusive scan of one int per thread; `total` is returned to every thread.  wsum: kNW ints of LDS.
// LEAD_BARRIER = false: the caller vouches that every thread has passed a barrier since wsum was last read.
template <int NT = kNT, bool LEAD_BARRIER = true>
__device__ __forceinline__ int block_excl_scan(int v, int *wsum, int &total) {
  constexpr int kNW = NT / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
  inc = wave_incl_scan(inc);
  if (LEAD_BARRIER) __syncthreads();  // wsum may still be read from a previous call
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kNW; ++i) {
    int s = wsum[i];
    if (i < w) base += s;
    tot += s;
  }
  total = tot;
  return base + inc - v;
}

// hist[bin] += 1 for every active lane with `valid`.  The two most frequent... rather: the bins of the first
// two distinct pending lanes are added wave-aggregated (one LDS add of the lane count each); what is left
// adds individually.  For the high-order digits of fp32 scores a wave sees one or two distinct bins, and 64
// same-address LDS atomics (which the LDS serialises) collapse into one or two.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool valid) {
  bool pend = valid;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const unsigned long long act = ballot64(pend);
    if (act == 0ull) return;  // wave-uniform
    const int first = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(act));
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
    const unsigned long long same = ballot64(pend && (bin == b0));
    if (lane_id() == first) atomicAdd(&hist[b0], (uint32_t)__popcll(same));
    pend = pend && (bin != b0);
  }
  if (pend) atomicAdd(&hist[bin], 1u);
}

// Radix select over n elements.  keyf(i, key) -> bool valid (key: smaller = earlier).  Finds the key T
// that the element of 1-based ascending rank `need` carries, and need_eq = how many elements equal to T
// are inside the first `need` (to be taken in index order).  all = true when fewer than `need` valid
// elements exist (then every valid element is inside).  hist: kNW*256 uint32 of LDS, misc: >= 4 ints.
template <int NT = kNT, class KeyF>
__device__ __forceinline__ void radix_select(KeyF keyf, int n, int need, uint32_t *hist, int *misc, uint32_t &T,
                                             int &need_eq, bool &all) {
  constexpr int kNT = NT, kNW = NT / 64;
  static_assert(NT >= 256 && NT % 64 == 0, "radix_select: the 256 bins are summed by the first 256 threads");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  __syncthreads();
  if (tid == 0) misc[2] = 0;
  uint32_t prefix = 0;
  all = false;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    // wave-private histograms (no cross-wave contention on the LDS atomics), summed below
    for (int i = tid; i < kNW * 256; i += kNT) hist[i] = 0;
    __syncthreads();
    uint32_t *myhist = hist + w * 256;
    if (pass == 3) {
      // top digit of fp32 keys: a wave sees one or two distinct bins -> aggregate (uniform trip count for the ballots)
      const int n_up = (n + kNT - 1) / kNT * kNT;
      for (int i = tid; i < n_up; i += kNT) {
        uint32_t k = 0;
        const bool v = (i < n) && keyf(i, k);
        hist_add(myhist, (k >> shift) & 255u, v);
      }
    } else {
      // lower digits are spread over the 256 bins (and only keys with the right prefix count): plain LDS adds
      for (int i = tid; i < n; i += kNT) {
        uint32_t k = 0;
        if (keyf(i, k) && (k >> (shift + 8)) == prefix) atomicAdd(&myhist[(k >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t sum = 0;
#pragma unroll
      for (int ww = 0; ww < kNW; ++ww) sum += hist[ww * 256 + tid];
      hist[tid] = sum;  // (row 0, column tid: read above by this thread only)
    }
    __syncthreads();
    if (w == 0) {
      const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const uint32_t s4 = h0 + h1 + h2 + h3;
      uint32_t inc = s4;
      inc = wave_incl_scan(inc);
      const uint32_t exc = inc - s4;
      const bool here = (exc < (uint32_t)need) && (inc >= (uint32_t)need);
      const unsigned long long bal = ballot64(here);
      if (bal == 0ull) {
        if (lane == 0) misc[2] = 1;
      } else if (here) {
        uint32_t cum = exc;
        int d = 4 * lane;
        if (cum + h0 < (uint32_t)need) {
          cum += h0;
          ++d;
          if (cum + h1 < (uint32_t)need) {
            cum += h1;
            ++d;
            if (cum + h2 < (uint32_t)need) {
              cum += h2;
              ++d;
            }
          }
        }
        misc[0] = (int)((prefix << 8) | (uint32_t)d);
        misc[1] = need - (int)cum;
      }
    }
    __syncthreads();
    if (misc[2]) {
      all = true;
      break;
    }
    prefix = (uint32_t)misc[0];
    need = misc[1];
    __syncthreads();  // misc is rewritten by the next pass
  }
  T = prefix;
  need_eq = need;
  __syncthreads();
}

// The two upper passes of a radix selection, for SMALL n (a sample of the keys): P = the smallest 16-bit prefix such
// that at least `need` valid keys have (key >> 16) <= P, and cnt_le = how many valid keys that is (>= need; the surplus
// is what the prefix's bin holds beyond the rank asked for).  all = true when fewer than `need` valid keys exist (then
// P = 0xFFFF and cnt_le = their number).  low_byte = the smallest top byte any valid key has (0 when there is none):
// low_byte << 24 bounds every valid key from below.  One 256-bin histogram shared by the waves (the top byte through
// wave-aggregated adds), three barriers per pass.  hist: 256 uint32 of LDS, misc: >= 5 ints.
// have_pass1 (workgroup-uniform): the caller has already put the top-byte histogram of the valid keys into hist[0..255]
// (complete before the call: the function starts with a barrier), so the first pass only has to read it.
template <int NT = kNT, class KeyF>
__device__ __forceinline__ void prefix16_select(KeyF keyf, int n, int need, uint32_t *hist, int *misc, uint32_t &P,
                                                int &cnt_le, bool &all, uint32_t &low_byte, bool have_pass1 = false) {
  constexpr int kNT = NT;
  static_assert(NT >= 256 && NT % 64 == 0, "prefix16_select: the 256 bins are cleared by the first 256 threads");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  __syncthreads();  // hist / misc may still be read by what came before
  if (tid == 0) misc[2] = 0;
  uint32_t prefix = 0;
  int below = 0;  // valid keys whose prefix is smaller than the bin chosen so far
  all = false;
  cnt_le = 0;
  for (int pass = 1; pass >= 0; --pass) {
    const int shift = 16 + pass * 8;
    if (!(pass == 1 && have_pass1)) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      if (pass == 1) {
        const int n_up = (n + kNT - 1) / kNT * kNT;  // uniform trip count for the ballots
        for (int i = tid; i < n_up; i += kNT) {
          uint32_t k = 0;
          const bool v = (i < n) && keyf(i, k);
          hist_add(hist, (k >> shift) & 255u, v);
        }
      } else {
        for (int i = tid; i < n; i += kNT) {
          uint32_t k = 0;
          if (keyf(i, k) && (k >> 24) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
    }
    if (w == 0) {
      const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const uint32_t s4 = h0 + h1 + h2 + h3;
      uint32_t inc = s4;
      inc = wave_incl_scan(inc);
      const uint32_t exc = inc - s4;
      const bool here = (exc < (uint32_t)need) && (inc >= (uint32_t)need);
      const unsigned long long bal = ballot64(here);
      if (pass == 1) {
        const unsigned long long occ = ballot64(s4 != 0u);
        if (occ == 0ull) {
          if (lane == 0) misc[4] = 0;
        } else if (lane == (int)__builtin_ctzll(occ)) {
          misc[4] = 4 * lane + (h0 ? 0 : h1 ? 1 : h2 ? 2 : 3);
        }
      }
      if (bal == 0ull) {
        if (lane == 63) {
          misc[2] = 1;
          misc[3] = (int)inc;  // every valid key
        }
      } else if (here) {
        uint32_t cum = exc, hd = h0;
        int d = 4 * lane;
        if (cum + h0 < (uint32_t)need) {
          cum += h0;
          hd = h1;
          ++d;
          if (cum + h1 < (uint32_t)need) {
            cum += h1;
            hd = h2;
            ++d;
            if (cum + h2 < (uint32_t)need) {
              cum += h2;
              hd = h3;
              ++d;
            }
          }
        }
        misc[0] = (int)((prefix << 8) | (uint32_t)d);
        misc[1] = (int)cum;         // keys of this pass below the chosen bin
        misc[3] = (int)(cum + hd);  // ... and up to its end
      }
    }
    __syncthreads();
    // (misc is next written two barriers from here, hist cleared by threads that are past this one)
    if (misc[2]) {  // only possible in the first pass (the second looks inside a bin that holds the rank)
      all = true;
      prefix = 0xFFFFu;
      cnt_le = below + misc[3];
      break;
    }
    prefix = (uint32_t)misc[0];
    cnt_le = below + misc[3];
    below += misc[1];
    need -= misc[1];
  }
  P = prefix;
  low_byte = (uint32_t)misc[4];
}

// Stable least-significant-digit radix sort of V slot numbers: `npass` passes of 8 bits, digit(slot, pass)
// gives the pass's digit of a slot (pass 0 = least significant).  perm[0..V) holds the initial order
// (normally 0..V-1); perm[stride..stride+V) is the ping-pong half.  Returns the offset (0 or stride) of the
// half that holds the sorted order.  (The two halves are addressed through one base pointer and an integer
// offset: selecting between two LDS pointers makes hipcc 7.2 emit an illegal null-check on
// src_shared_base.)  hist: kNW*256 uint32 of LDS; wsum: kNW ints.  V <= 65536.
template <int NT = kNT, class DigitF>
__device__ __forceinline__ int block_sort_digits(DigitF digit, int npass, uint16_t *perm, int stride, int V,
                                                 uint32_t *hist, int *wsum) {
  constexpr int kNT = NT, kNW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nchunks = (V + 63) >> 6;
  const int cpw = (nchunks + kNW - 1) / kNW;
  const int c0 = w * cpw;
  const int c1 = (c0 + cpw < nchunks) ? (c0 + cpw) : nchunks;
  for (int pass = 0; pass < npass; ++pass) {
    const int src = (pass & 1) ? stride : 0;
    const int dst = stride - src;
    for (int i = tid; i < kNW * 256; i += kNT) hist[i] = 0;
    __syncthreads();  // also publishes the caller's keys / perm on the first pass
    for (int ch = c0; ch < c1; ++ch) {
      const int p = (ch << 6) + lane;
      const bool valid = p < V;
      const uint32_t d = valid ? (digit((int)perm[src + p], pass) & 255u) : 0u;
      hist_add(hist, d * kNW + w, valid);
    }
    __syncthreads();
    {
      const uint32_t a0 = hist[4 * tid], a1 = hist[4 * tid + 1], a2 = hist[4 * tid + 2], a3 = hist[4 * tid + 3];
      int total;
      const int ex = block_excl_scan<NT>((int)(a0 + a1 + a2 + a3), wsum, total);
      hist[4 * tid] = (uint32_t)ex;
      hist[4 * tid + 1] = (uint32_t)ex + a0;
      hist[4 * tid + 2] = (uint32_t)ex + a0 + a1;
      hist[4 * tid + 3] = (uint32_t)ex + a0 + a1 + a2;
    }
    __syncthreads();
    for (int ch = c0; ch < c1; ++ch) {
      const int p = (ch << 6) + lane;
      const bool valid = p < V;
      const uint32_t id = valid ? (uint32_t)perm[src + p] : 0u;
      const uint32_t d = digit((int)id, pass) & 255u;
      unsigned long long peers = ballot64(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool on = (d >> bit) & 1u;
        const unsigned long long bm = ballot64(on);
        peers &= on ? bm : ~bm;
      }
      if (valid) {
        // relaxed atomics (not volatile: a volatile LDS access is lowered through a generic pointer by hipcc 7.2)
        const uint32_t base = __hip_atomic_load(&hist[d * kNW + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const int r = __popcll(peers & lanemask_lt());
        perm[dst + (int)base + r] = (uint16_t)id;
        if (r == 0)
          __hip_atomic_store(&hist[d * kNW + w], base + (uint32_t)__popcll(peers), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  return (npass & 1) ? stride : 0;
}

// Bucket sort of V slot numbers by (keys[slot], slot) ascending: 8 * NT buckets linear in the key over [lowest, highest
// key] (16-bit counters, two to an LDS word), one LDS add per slot that also hands back its place inside the bucket, a
// scan, the placement, and an insertion sort inside every bucket (the slot number as second key is what a stable sort of
// index-ordered slots would give).  Seven barriers for any V, against four LSD passes of six.  fp32 score keys are
// log-spaced across binades and linear inside one, so uniform scores put half the slots into 1/13 of the buckets: hence
// the many buckets.  Pays when the keys spread, not when they cluster: returns false -- perm undefined -- when a bucket
// holds more than 48 slots, and the caller sorts by radix instead.
// perm: [V] out; place: [V] scratch; hist: 4 * NT uint32; wsum: NT / 64 ints; misc: >= 10 ints (uses [6], [7], [9]).  V < 65536.
// by_value: the keys are ~f_order_key(score) and the buckets are linear in the SCORE instead of in the key's bit pattern
// (fp32 bit patterns are log-spaced across binades: uniformly distributed scores put half the slots into the top
// binade's share of key-linear buckets); refused (false) when the lowest or highest key is not a finite score.
template <int NT = kNT>
__device__ __forceinline__ bool block_bucket_sort_perm(const uint32_t *keys, uint16_t *perm, uint16_t *place, int V,
                                                       uint32_t *hist, int *wsum, int *misc, bool by_value = false) {
  constexpr int NBK = 8 * NT;
  uint16_t *start = reinterpret_cast<uint16_t *>(hist);  // [NBK] counters, then bucket starts
  const int tid = threadIdx.x;
  __syncthreads();  // the caller's keys; hist / misc may still be read by what came before
  if (tid == 0) {
    misc[6] = -1;  // lowest key (unsigned min)
    misc[7] = 0;   // highest key
    misc[9] = 0;   // overflow
  }
  for (int i = tid; i < NBK / 2; i += NT) hist[i] = 0u;
  __syncthreads();
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
  for (int j = tid; j < V; j += NT) {
    const uint32_t k = keys[j];
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t ol = (uint32_t)__shfl_xor((int)lo, off, 64), oh = (uint32_t)__shfl_xor((int)hi, off, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  if ((tid & 63) == 0) {
    atomicMin(reinterpret_cast<uint32_t *>(&misc[6]), lo);
    atomicMax(reinterpret_cast<uint32_t *>(&misc[7]), hi);
  }
  __syncthreads();
  const uint32_t kmin = (uint32_t)misc[6], kmax = (uint32_t)misc[7];
  // score of a key: the inverse of ~f_order_key (non-negative floats carry the top bit in the order key)
  auto value_of = [](uint32_t k) -> float {
    const uint32_t o = ~k;
    return u2f((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
  };
  const float fmax = value_of(kmin), fmin = value_of(kmax);  // ascending key = descending score
  if (by_value && !(fabsf(fmax) <= 3.402823466e+38f && fabsf(fmin) <= 3.402823466e+38f)) return false;  // block-uniform
  const float scale = by_value ? (float)NBK / (fmax - fmin) : (float)NBK / ((float)(kmax - kmin) + 1.0f);
  auto bucket_of = [&](uint32_t k) -> int {
    // monotone in k either way (a subtraction from a constant, a product with a positive constant, a truncation);
    // scale = inf (one distinct score) makes 0 * inf = NaN -> bucket 0 for the slot(s) at fmax, the last bucket otherwise
    const float x = by_value ? (fmax - value_of(k)) * scale : (float)(k - kmin) * scale;
    const int bkt = (x >= 0.0f) ? (int)fminf(x, (float)(NBK - 1)) : 0;
    return bkt < NBK - 1 ? bkt : NBK - 1;
  };
  for (int j = tid; j < V; j += NT) {
    const int bkt = bucket_of(keys[j]);
    const uint32_t old = atomicAdd(&hist[bkt >> 1], (bkt & 1) ? 0x10000u : 1u);  // (V < 65536: the low half cannot carry)
    place[j] = (uint16_t)((bkt & 1) ? (old >> 16) : (old & 0xFFFFu));
  }
  __syncthreads();
  {
    uint32_t wds[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wds[u] = hist[4 * tid + u];
    uint32_t sum = 0, big = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t c0 = wds[u] & 0xFFFFu, c1 = wds[u] >> 16;
      sum += c0 + c1;
      big = (c0 > big) ? c0 : big;
      big = (c1 > big) ? c1 : big;
    }
    int total;
    uint32_t run = (uint32_t)block_excl_scan<NT>((int)sum, wsum, total);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t c0 = wds[u] & 0xFFFFu, c1 = wds[u] >> 16;
      hist[4 * tid + u] = run | ((run + c0) << 16);  // the two buckets' starts (< 65536)
      run += c0 + c1;
    }
    if (big > 48u) misc[9] = 1;
  }
  __syncthreads();
  if (misc[9]) return false;  // block-uniform
  for (int j = tid; j < V; j += NT) perm[start[bucket_of(keys[j])] + place[j]] = (uint16_t)j;
  __syncthreads();
  {
    uint32_t begin = start[8 * tid];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t end = (8 * tid + u + 1 < NBK) ? start[8 * tid + u + 1] : (uint32_t)V;
      for (uint32_t a = begin + 1; a < end; ++a) {
        const uint16_t sj = perm[a];
        const uint32_t kj = keys[sj];
        uint32_t c = a;
        while (c > begin) {
          const uint16_t sp = perm[c - 1];
          const uint32_t kp = keys[sp];
          if (!(kp > kj || (kp == kj && sp > sj))) break;
          perm[c] = sp;
          --c;
        }
        perm[c] = sj;
      }
      begin = end;
    }
  }
  __syncthreads();
  return true;
}

// The common case: 4 passes over 32-bit keys[slot]; the sorted order ends up in perm[0..V).
__device__ __forceinline__ void block_sort_perm(const uint32_t *keys, uint16_t *perm, int stride, int V,
                                                uint32_t *hist, int *wsum) {
  block_sort_digits([&](int slot, int pass) -> uint32_t { return keys[slot] >> (8 * pass); }, 4, perm, stride, V, hist,
                    wsum);
}

}  // namespace frcnn
