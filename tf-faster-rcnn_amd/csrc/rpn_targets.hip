// rpn_targets.hip -- RPN target assignment: random mask sub-sampling and the target tail.
//
// Replaces utils/train_utils.py:72-88 (randomly_select_xyz_mask) and :180-202 (the part of
// calculate_rpn_actual_outputs after the IoU reductions, which live in iou_map.hip).
//
// randomly_select_xyz_mask keeps, per image, the masked elements whose position in the descending
// argsort of key = mask * random_int is below `select` (ties: lower index first).  Two full-row
// argsorts are replaced by a SELECTION: a 4x8-bit radix select over the N keys (LDS histogram) finds
// the key T of rank `select`; elements with key > T are in, elements with key == T are in while their
// index-ordered rank among equals is below the remainder.  The result is identical to the argsort
// formulation for every input, including ties and random integers <= 0.
//
// One 1024-thread workgroup per image; the per-anchor inputs (max IoU, random ints) are re-read from
// global memory in each radix pass (36 KB per image, L2 resident) instead of being held in LDS, which
// only keeps three N-bit maps.  Latency bound: ~0.3 MB of traffic per image.
#include <mutex>

#include "block_primitives.hpp"

namespace frcnn {

static const int kTNT = 1024;
static const int kTNW = 16;
static const int kMaxPosList = 1024;  // rpn_targets_kernel: selected positives listed in LDS (beyond that: encoded in place)

// ascending-order key: smaller = better (larger signed int first)
__device__ __forceinline__ uint32_t inv_key(int key) { return ~(((uint32_t)key) ^ 0x80000000u); }

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// In-kernel stand-in for tf.random.uniform(minval=1, maxval=maxval, dtype=int32) (train_utils.py:83): a counter-based hash
// of (seed, sampling call, image, element).  The 64-bit mixing of (seed, call, image) happens once per workgroup (DrawKey,
// block-uniform); per element it is one 32-bit multiply-xorshift round ("lowbias32") and a multiply-high range reduction --
// the first version mixed 64 bits twice per draw and took a modulo, and those ~160 instructions per anchor were 7 of the
// target kernel's 32 us.
struct DrawKey {
  uint32_t k0, k1;
};
__device__ __forceinline__ DrawKey make_draw_key(uint64_t seed, uint32_t stream, int b) {
  const uint64_t h = mix64(mix64(seed ^ ((uint64_t)stream << 56)) + (uint64_t)(uint32_t)b);
  DrawKey k;
  k.k0 = (uint32_t)h;
  k.k1 = (uint32_t)(h >> 32);
  return k;
}
__device__ __forceinline__ int draw_int(const DrawKey &key, int n, int maxval) {
  uint32_t x = ((uint32_t)n ^ key.k0) * 0x9E3779B1u + key.k1;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  const uint32_t span = (maxval > 2) ? (uint32_t)(maxval - 1) : 1u;
  return 1 + (int)__umulhi(x, span);  // U[1, maxval)
}

// Workgroup-cooperative selection.  keyf(n) -> int key (= mask * random); emit(n, in) is called once
// for every n < N with in == (rank of n in the descending key order, ties by index) < S.
// hist: 256 uint32 in LDS; misc: 8 ints in LDS; wsum: kTNW ints in LDS; kcache: N uint32 of LDS or NULL -- with it
// keyf is evaluated once per element (its inputs usually sit in global memory) and the passes read LDS.
struct NoMask {};

// maskf (optional, with kcache): maskf(n) = the element's mask bit, for the shortcut below.  The reference keeps the
// masked elements whose rank in the descending order of key = mask * random is below S.  If no more than S keys are
// positive, every positive key has rank < S; and if besides no MASKED element has a key <= 0 (random integers are >= 1 by
// contract, but out-of-contract ones are honoured), the elements with key <= 0 are all unmasked and dropped by the caller
// whatever their rank: the radix selection is skipped.  This is the usual case for the positives of target assignment
// (a few dozen anchors above the IoU threshold against S = 128).
// hist4k (optional, with kcache): 4096 uint32 of LDS.  When every key lies in [0, 4096) -- mask * (a random integer below
// 10 * the sample count: the reference's own range, utils/train_utils.py:82-83) -- the rank-S key comes out of ONE counting
// pass (a histogram over the key values and a block scan from the top) instead of the radix passes.
template <class KeyF, class EmitF, class MaskF = NoMask>
__device__ __forceinline__ void select_largest(KeyF keyf, int N, int S, uint32_t *hist, int *misc, int *wsum,
                                               EmitF emit, uint32_t *kcache = nullptr, MaskF maskf = MaskF(),
                                               uint32_t *hist4k = nullptr) {
  constexpr bool kHasMask = !__is_same(MaskF, NoMask);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (S <= 0 || S >= N) {  // uniform: nobody / everybody has rank < S
    const bool in = (S >= N) && (S > 0);
    for (int n = tid; n < N; n += kTNT) emit(n, in);
    return;
  }
  uint32_t vary = 0xFFFFFFFFu, common = 0u;  // key bits that differ between elements / the value of those that do not
  if (kcache) {
    if (tid == 0) {
      misc[4] = 0;
      misc[5] = -1;
    }
    __syncthreads();  // the cache may still be read by a previous selection
    int npos = 0, bad = 0;
    uint32_t orv = 0u, andv = 0xFFFFFFFFu;
    for (int n0 = tid; n0 < N; n0 += 4 * kTNT) {  // four elements' loads in flight (keyf usually reads global memory)
      int k4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k4[u] = (n0 + u * kTNT < N) ? keyf(n0 + u * kTNT) : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + u * kTNT;
        if (n < N) {
          const int k = k4[u];
          const uint32_t ik = inv_key(k);
          kcache[n] = ik;
          orv |= ik;
          andv &= ik;
          npos += (k > 0);
          if constexpr (kHasMask) bad |= (k <= 0) && maskf(n);
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      orv |= (uint32_t)__shfl_xor((int)orv, off, 64);
      andv &= (uint32_t)__shfl_xor((int)andv, off, 64);
    }
    if (lane == 0) {
      atomicOr(reinterpret_cast<unsigned int *>(&misc[4]), orv);
      atomicAnd(reinterpret_cast<unsigned int *>(&misc[5]), andv);
    }
    if constexpr (kHasMask) {
      int total;
      (void)block_excl_scan(npos + (bad << 20), wsum, total);  // N < 2^20 (kMaxTargetsN), 1024 flags < 2^11; has the barriers
      if ((total >> 20) == 0 && (total & 0xFFFFF) <= S) {      // block-uniform
        for (int n = tid; n < N; n += kTNT) emit(n, kcache[n] < inv_key(0));
        return;
      }
    } else {
      __syncthreads();
    }
    // Keys are mask * (random integer below a few thousand): their upper bytes are the same for every element, and a
    // radix pass over a byte in which no two keys differ selects nothing -- it is skipped (2 passes instead of 4).
    common = (uint32_t)misc[4];
    vary = common ^ (uint32_t)misc[5];
  }
  FRCNN_STAMP(35);
  auto key_at = [&](int n) -> uint32_t { return kcache ? kcache[n] : inv_key(keyf(n)); };
  uint32_t prefix = 0;
  int need = S;
  const bool counting = kcache && hist4k && (common >> 12) == 0x7FFFFu && (vary >> 12) == 0u;  // block-uniform
  if (counting) {
    // inv_key(k) = 0x7FFFF000 + (4095 - k) for k in [0, 4096): bin d = 4095 - k ascends as the key descends
    for (int i = tid; i < 4096; i += kTNT) hist4k[i] = 0u;
    __syncthreads();
    for (int n = tid; n < N; n += kTNT) {
      const uint32_t d = kcache[n] - 0x7FFFF000u;
      if (d != 4095u) atomicAdd(&hist4k[d], 1u);  // key 0 (masked out, most of the row) is what is left over
    }
    __syncthreads();
    static_assert(kTNT * 4 == 4096, "four bins per thread");
    const uint32_t a0 = hist4k[4 * tid], a1 = hist4k[4 * tid + 1], a2 = hist4k[4 * tid + 2], a3 = hist4k[4 * tid + 3];
    int total;
    const int ex = block_excl_scan((int)(a0 + a1 + a2 + a3), wsum, total);  // keys above this thread's bins / positive keys
    if (ex < S && S <= ex + (int)(a0 + a1 + a2 + a3)) {  // exactly one thread when S <= total
      int cum = ex, d = 4 * tid;
      if (cum + (int)a0 < S) {
        cum += (int)a0;
        ++d;
        if (cum + (int)a1 < S) {
          cum += (int)a1;
          ++d;
          if (cum + (int)a2 < S) {
            cum += (int)a2;
            ++d;
          }
        }
      }
      misc[0] = d;
      misc[1] = S - cum;
    }
    if (tid == 0 && S > total) {  // the quota reaches into the zeros: they fill it in index order
      misc[0] = 4095;
      misc[1] = S - total;
    }
    __syncthreads();
    prefix = 0x7FFFF000u + (uint32_t)misc[0];
    need = misc[1];
  } else
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    if (((vary >> shift) & 255u) == 0u) {  // block-uniform: every key carries the same digit here
      prefix = (prefix << 8) | ((common >> shift) & 255u);
      continue;
    }
    __syncthreads();
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    {
      // Keys are mask * (small random integer): the masked-out elements (key 0, usually most of the row) share every digit
      // and the rest share the upper ones, so in every pass most lanes of a wave hit one or two bins -- wave-aggregated adds
      // (hist_add) instead of thousands of serialised same-address LDS atomics.  Uniform trip count for its ballots.
      const int n_up = (N + kTNT - 1) / kTNT * kTNT;
      for (int n = tid; n < n_up; n += kTNT) {
        const uint32_t k = (n < N) ? key_at(n) : 0u;
        const bool v = (n < N) && (pass == 3 || (k >> (shift + 8)) == prefix);
        hist_add(hist, (k >> shift) & 255u, v);
      }
    }
    __syncthreads();
    if (w == 0) {
      const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const uint32_t s4 = h0 + h1 + h2 + h3;
      uint32_t inc = s4;
      inc = (uint32_t)wave_incl_scan((int)inc);
      const uint32_t exc = inc - s4;
      if ((exc < (uint32_t)need) && (inc >= (uint32_t)need)) {  // exactly one lane (S < N)
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
    prefix = (uint32_t)misc[0];
    need = misc[1];
  }
  const uint32_t T = prefix;
  const int need_eq = need;
  FRCNN_STAMP(36);
  // index-ordered rank among the elements with key == T: every thread owns a contiguous run of elements, one block scan
  // of the per-run counts gives the rank of its first equal element
  const int E = (N + kTNT - 1) / kTNT;
  const int n0 = tid * E;
  const int n1 = (n0 + E < N) ? (n0 + E) : N;
  int eq = 0;
  for (int n = n0; n < n1; ++n) eq += (key_at(n) == T);
  int total_eq;
  int rank = block_excl_scan(eq, wsum, total_eq);
  FRCNN_STAMP(37);
  for (int n = n0; n < n1; ++n) {
    const uint32_t k = key_at(n);
    const bool is_eq = (k == T);
    emit(n, (k < T) || (is_eq && rank < need_eq));
    rank += is_eq;
  }
}

// utils/train_utils.py:72-88, one image per workgroup.
// rnd == NULL: the random integers are drawn in-kernel from `seed`, U[1, 10 * max(select)) as :82-83 -- the maximum over
// the batch is taken here, on the device (every workgroup reads the n_select <= B entries), so the host never has to
// look at `select`.
__global__ __launch_bounds__(1024) void select_mask_kernel(const uint8_t *__restrict__ mask,
                                                           const int *__restrict__ rnd, unsigned long long seed,
                                                           const int *__restrict__ select, int n_select, int N,
                                                           uint8_t *__restrict__ out) {
  __shared__ uint32_t hist[256];
  __shared__ int misc[8];
  __shared__ int wsum[kTNW];
  __shared__ int s_max;
  const int b = blockIdx.x;
  const uint8_t *m = mask + (long long)b * N;
  uint8_t *o = out + (long long)b * N;
  const int S = select[n_select == 1 ? 0 : b];
  if (rnd) {
    const int *r = rnd + (long long)b * N;
    select_largest([&](int n) { return (m[n] ? 1 : 0) * r[n]; }, N, S, hist, misc, wsum,
                   [&](int n, bool in) { o[n] = (uint8_t)((m[n] != 0) && in); });
    return;
  }
  if (threadIdx.x == 0) s_max = select[0];
  __syncthreads();
  int mx = select[0];
  for (int i = threadIdx.x; i < n_select; i += kTNT) mx = (select[i] > mx) ? select[i] : mx;
  if (mx > select[0]) atomicMax(&s_max, mx);
  __syncthreads();
  const long long mv = 10ll * s_max;
  const int maxval = mv > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)mv;  // (<= 1: tf.random.uniform raises; here every draw is 1 and S <= 0 selects nothing)
  const DrawKey dk = make_draw_key(seed, 4u, b);
  select_largest([&](int n) { return (m[n] ? 1 : 0) * draw_int(dk, n, maxval); }, N, S, hist, misc, wsum,
                 [&](int n, bool in) { o[n] = (uint8_t)((m[n] != 0) && in); });
}

// utils/train_utils.py:180-202, one image per workgroup.  LDS: five N-bit maps (forced / positive mask / "max IoU below
// 0.3" / the two selections), optionally the selection keys (N words) and the 4096 counting bins.
__global__ __launch_bounds__(1024) void rpn_targets_kernel(
    const float *__restrict__ anchors, const float *__restrict__ gt_boxes, const int *__restrict__ gt_labels,
    const int *__restrict__ rnd_pos, const int *__restrict__ rnd_neg, unsigned long long seed, int N, int M,
    int total_pos, int total_neg, float v0, float v1, float v2, float v3, const float *__restrict__ max_iou,
    const int *__restrict__ argmax_row, const unsigned long long *__restrict__ col_keys, float *__restrict__ out_deltas,
    float *__restrict__ out_labels, int cache_keys) {
  extern __shared__ __attribute__((aligned(16))) uint32_t maps[];
  __shared__ uint32_t hist[256];
  __shared__ int misc[8];
  __shared__ int wsum[kTNW];
  __shared__ int s_pos_count;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int words = ((N + 63) >> 6) << 1;  // whole 64-bit words: a wave's ballot is stored as one
  uint32_t *forced = maps;
  uint32_t *pos_sel = maps + words;
  uint32_t *neg_sel = maps + 2 * words;
  uint32_t *pos_msk = maps + 3 * words;  // max IoU > 0.7 or forced (:180-188)
  uint32_t *low_msk = maps + 4 * words;  // max IoU < 0.3 (:193)
  uint32_t *kcache = cache_keys ? maps + 5 * words : nullptr;  // [N]
  uint32_t *hist4k = cache_keys ? kcache + N : nullptr;        // [4096]
  int *krn = cache_keys ? reinterpret_cast<int *>(hist4k + 4096) : nullptr;  // [N] the negatives' random integers
  int *plist = cache_keys ? krn + N : nullptr;                 // [kMaxPosList] the selected positives
  const float *miou = max_iou + (long long)b * N;
  const int *rp = rnd_pos ? rnd_pos + (long long)b * N : nullptr;
  const int *rn = rnd_neg ? rnd_neg + (long long)b * N : nullptr;
  auto bit = [](const uint32_t *map, int n) -> bool { return (map[n >> 5] >> (n & 31)) & 1u; };
  const DrawKey dk_pos = make_draw_key(seed, 0u, b), dk_neg = make_draw_key(seed, 1u, b);

  FRCNN_STAMP(30);
  for (int i = tid; i < 3 * words; i += kTNT) maps[i] = 0u;
  if (tid == 0) s_pos_count = 0;
  __syncthreads();
  // :181-188 every valid gt forces its best anchor positive
  for (int m = tid; m < M; m += kTNT) {
    if (gt_labels[(long long)b * M + m] != -1) {
      const int a = col_key_index(col_keys[(long long)b * M + m]);  // the per-gt argmax, still in its merge-key form
      if (a >= 0 && a < N) atomicOr(&forced[a >> 5], 1u << (a & 31));
    }
  }
  __syncthreads();
  // The two masks as bit maps, the per-anchor inputs read ONCE (coalesced, twelve loads per thread in flight): with the
  // key cache the positives' raw keys (mask * random integer) and the negatives' random integers go to LDS here, and the
  // selections' key and emit functions only touch LDS.
  {
    const int n_up = (N + 63) & ~63;  // whole waves (uniform per wave) for the ballots
    for (int n0 = tid; n0 < n_up; n0 += 4 * kTNT) {
      float mi[4];
      int rpv[4], rnv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + u * kTNT;
        mi[u] = (n < N) ? miou[n] : 0.5f;
        rpv[u] = rnv[u] = 0;
        if (kcache && n < N) {
          rpv[u] = rp ? rp[n] : draw_int(dk_pos, n, 10 * total_pos);
          rnv[u] = rn ? rn[n] : draw_int(dk_neg, n, 10 * (total_pos + total_neg));
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + u * kTNT;
        if (n - lane < n_up) {  // wave-uniform
          const bool live = n < N;
          const bool pmb = live && ((mi[u] > 0.7f) || bit(forced, n));
          const unsigned long long pm = ballot64(pmb);
          const unsigned long long lm = ballot64(live && (mi[u] < 0.3f));
          if (kcache && live) {
            kcache[n] = (uint32_t)((pmb ? 1 : 0) * rpv[u]);  // the raw key; select_largest turns it into its sort key
            krn[n] = rnv[u];
          }
          if (lane == 0) {
            pos_msk[n >> 5] = (uint32_t)pm;
            pos_msk[(n >> 5) + 1] = (uint32_t)(pm >> 32);
            low_msk[n >> 5] = (uint32_t)lm;
            low_msk[(n >> 5) + 1] = (uint32_t)(lm >> 32);
          }
        }
      }
    }
  }
  __syncthreads();

  FRCNN_STAMP(31);
  // :180,:188-189 positives, sub-sampled to total_pos
  auto pos_key = [&](int n) -> int {
    if (kcache) return (int)kcache[n];  // (each thread reads the slot it then overwrites)
    const int r = rp ? rp[n] : draw_int(dk_pos, n, 10 * total_pos);
    return (bit(pos_msk, n) ? 1 : 0) * r;
  };
  select_largest(pos_key, N, total_pos, hist, misc, wsum, [&](int n, bool in) {
    if (in && bit(pos_msk, n)) {
      atomicOr(&pos_sel[n >> 5], 1u << (n & 31));
      const int at = atomicAdd(&s_pos_count, 1);
      if (plist && at < kMaxPosList) plist[at] = n;
    }
  }, kcache, [&](int n) -> bool { return bit(pos_msk, n); }, hist4k);
  __syncthreads();
  FRCNN_STAMP(32);
  // :190-191
  const int neg_count = (total_pos + total_neg) - s_pos_count;

  // :193-194 negatives, sub-sampled to neg_count
  auto neg_key = [&](int n) -> int {
    const bool msk = bit(low_msk, n) && !bit(pos_sel, n);
    const int r = krn ? krn[n] : (rn ? rn[n] : draw_int(dk_neg, n, 10 * (total_pos + total_neg)));
    return (msk ? 1 : 0) * r;
  };
  select_largest(neg_key, N, neg_count, hist, misc, wsum, [&](int n, bool in) {
    if (in && bit(low_msk, n) && !bit(pos_sel, n)) atomicOr(&neg_sel[n >> 5], 1u << (n & 31));
  }, kcache, [&](int n) -> bool { return bit(low_msk, n) && !bit(pos_sel, n); }, hist4k);
  __syncthreads();

  FRCNN_STAMP(33);
  // :195-200 labels and regression targets.  An anchor that is not positive is paired with the all-zero gt row, for which
  // get_deltas_from_bboxes yields exactly (0, 0, 0, 0) whatever the anchor (bbox_utils.py:121-135: gt width and height
  // are 0, so every tf.where takes its zeros branch): only the positives read their anchor and gt box and go through
  // the encoder; the rest of the (N,4) rows is a streamed constant, 0 / variance.
  const float4 zero_d = make_float4(0.0f / v0, 0.0f / v1, 0.0f / v2, 0.0f / v3);
  auto encode_row = [&](int n) {
    const int m = argmax_row[(long long)b * N + n];
    const float4 g = *reinterpret_cast<const float4 *>(gt_boxes + ((long long)b * M + m) * 4);
    const float4 a = *reinterpret_cast<const float4 *>(anchors + (long long)n * 4);
    float4 d = encode_box(a, g);
    d.x = d.x / v0;
    d.y = d.y / v1;
    d.z = d.z / v2;
    d.w = d.w / v3;
    *reinterpret_cast<float4 *>(out_deltas + ((long long)b * N + n) * 4) = d;
  };
  // with the list of positives the streamed pass has no loads and no divergent work; the positives (at most total_pos)
  // are encoded by one thread each afterwards
  const int npos_sel = s_pos_count;
  const bool listed = plist && npos_sel <= kMaxPosList;  // block-uniform
  for (int n = tid; n < N; n += kTNT) {
    const bool pos = bit(pos_sel, n);
    const bool neg = bit(neg_sel, n);
    const float pl = pos ? 1.0f : -1.0f;
    const float nl = neg ? 1.0f : 0.0f;
    out_labels[(long long)b * N + n] = pl + nl;
    if (pos && M > 0) {
      if (!listed) encode_row(n);
    } else {
      *reinterpret_cast<float4 *>(out_deltas + ((long long)b * N + n) * 4) = zero_d;
    }
  }
  if (listed && M > 0)
    for (int i = tid; i < npos_sel; i += kTNT) encode_row(plist[i]);
  FRCNN_STAMP(34);
}

// models/faster_rcnn.py:218-234 (RoIDelta.call after the IoU reductions), one image per workgroup.
// LDS: two R-bit maps, then the per-RoI encoded delta (float4), label (int), selection key (uint32) and the 4096 counting bins.
__global__ __launch_bounds__(1024) void roi_delta_kernel(
    const float *__restrict__ rois, const float *__restrict__ gt_boxes, const int *__restrict__ gt_labels,
    const int *__restrict__ rnd_pos, const int *__restrict__ rnd_neg, unsigned long long seed, int R, int M, int L,
    int total_pos, int total_neg, float v0, float v1, float v2, float v3, const float *__restrict__ max_iou,
    const int *__restrict__ argmax_row, float *__restrict__ out_deltas, float *__restrict__ out_labels) {
  extern __shared__ __attribute__((aligned(16))) uint32_t maps[];
  __shared__ uint32_t hist[256];
  __shared__ int misc[8];
  __shared__ int wsum[kTNW];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int words = (R + 31) >> 5;
  const int words4 = (2 * words + 3) & ~3;  // keep the float4 array 16-byte aligned
  uint32_t *pos_sel = maps;
  uint32_t *neg_sel = maps + words;
  float4 *s_delta = reinterpret_cast<float4 *>(maps + words4);  // [R]
  int *s_label = reinterpret_cast<int *>(s_delta + R);           // [R]
  uint32_t *kcache = reinterpret_cast<uint32_t *>(s_label + R);  // [R] selection keys
  uint32_t *hist4k = kcache + R;                                 // [4096] counting bins
  const float *miou = max_iou + (long long)b * R;
  const int *rp = rnd_pos ? rnd_pos + (long long)b * R : nullptr;
  const int *rn = rnd_neg ? rnd_neg + (long long)b * R : nullptr;

  const DrawKey dk_pos = make_draw_key(seed, 2u, b), dk_neg = make_draw_key(seed, 3u, b);
  for (int i = tid; i < 2 * words; i += kTNT) maps[i] = 0u;
  __syncthreads();
  // :218-219 positives: max IoU > 0.5, sub-sampled to total_pos
  select_largest(
      [&](int n) -> int {
        const int r = rp ? rp[n] : draw_int(dk_pos, n, 10 * total_pos);
        return ((miou[n] > 0.5f) ? 1 : 0) * r;
      },
      R, total_pos, hist, misc, wsum,
      [&](int n, bool in) {
        if ((miou[n] > 0.5f) && in) atomicOr(&pos_sel[n >> 5], 1u << (n & 31));
      },
      kcache, NoMask(), hist4k);
  // :220-221 negatives: 0.1 < max IoU < 0.5, sub-sampled to total_neg
  select_largest(
      [&](int n) -> int {
        const int r = rn ? rn[n] : draw_int(dk_neg, n, 10 * total_neg);
        return (((miou[n] < 0.5f) && (miou[n] > 0.1f)) ? 1 : 0) * r;
      },
      R, total_neg, hist, misc, wsum,
      [&](int n, bool in) {
        if ((miou[n] < 0.5f) && (miou[n] > 0.1f) && in) atomicOr(&neg_sel[n >> 5], 1u << (n & 31));
      },
      kcache, NoMask(), hist4k);
  __syncthreads();

  // :222-230 per RoI: gathered gt box / label, encoded delta / variances
  for (int n = tid; n < R; n += kTNT) {
    const bool pos = (pos_sel[n >> 5] >> (n & 31)) & 1u;
    const bool neg = (neg_sel[n >> 5] >> (n & 31)) & 1u;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    int lab = -1;
    if (pos) {
      const int m = argmax_row[(long long)b * R + n];
      g = *reinterpret_cast<const float4 *>(gt_boxes + ((long long)b * M + m) * 4);
      lab = gt_labels[(long long)b * M + m];
    }
    lab += neg ? 1 : 0;
    const float4 a = *reinterpret_cast<const float4 *>(rois + ((long long)b * R + n) * 4);
    float4 d = encode_box(a, g);
    d.x = d.x / v0;
    d.y = d.y / v1;
    d.z = d.z / v2;
    d.w = d.w / v3;
    s_delta[n] = d;
    s_label[n] = lab;
  }
  __syncthreads();

  // :231-234 one-hot labels (B,R,L) and the "scatter" = one-hot * delta (a multiply: 0 * delta keeps sign / NaN), (B,R*L,4)
  const long long total = (long long)R * L;
  float *ol = out_labels + (long long)b * total;
  float4 *od = reinterpret_cast<float4 *>(out_deltas) + (long long)b * total;
  int n = tid / L, l = tid - n * L;
  const int dn = kTNT / L, dl = kTNT - dn * L;
  for (long long e = tid; e < total; e += kTNT) {
    const float oh = (s_label[n] == l) ? 1.0f : 0.0f;
    const float4 d = s_delta[n];
    float4 o;
    o.x = oh * d.x;
    o.y = oh * d.y;
    o.z = oh * d.z;
    o.w = oh * d.w;
    ol[e] = oh;
    od[e] = o;
    n += dn;
    l += dl;
    if (l >= L) {
      l -= L;
      ++n;
    }
  }
}

static const int kMaxTargetsN = 262144;
static const int kMaxRoiDeltaR = 4096;

// ---- frcnn_rpn_targets_static: the scratch of frcnn_rpn_targets in a file-scope device array --------------------------
// For callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc).  Every GPU the
// library runs on holds its own copy of the array until the process ends.  `scratch` is a host-side stand-in in host code,
// NOT a device address: its per-device address is looked up once per device with hipGetSymbolAddress (static_scratch)
// and only that pointer is ever handed to kernels / hipMemsetAsync.
static const size_t kStaticScratchBytes = FRCNN_RPN_TARGETS_STATIC_MAX_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

static const int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[kMaxDevices];  // per device: the address of that device's copy of `scratch`

// The current device's copy of `scratch` (looked up on the first call on that device, with that device current).
static int static_scratch(void **out) {
  int dev = -1;
  int rc = check_hip(hipGetDevice(&dev), "rpn_targets_static: hipGetDevice");
  if (rc) return rc;
  if (dev < 0 || dev >= kMaxDevices) return set_error(FRCNN_EUNSUPPORTED, "rpn_targets_static: device %d beyond %d", dev, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void *ptr = nullptr;
    rc = check_hip(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "rpn_targets_static: hipGetSymbolAddress(scratch)");
    if (rc) return rc;
    if (!ptr || !aligned_to(ptr, 256)) return set_error(FRCNN_EHIP, "rpn_targets_static: bad scratch address %p", ptr);
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return FRCNN_OK;
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

static int select_mask_impl(const uint8_t *mask, const int32_t *random_mask, uint64_t seed, const int32_t *select,
                            int n_select, int B, int N, uint8_t *out, frcnn_stream_t stream) {
  FRCNN_REQUIRE(B >= 0 && N >= 0, "select_mask: negative shape");
  FRCNN_REQUIRE(n_select == 1 || n_select == B, "select_mask: select_xyz must have 1 or batch_size (%d) entries, got %d", B, n_select);
  if (B == 0 || N == 0) return FRCNN_OK;
  hipLaunchKernelGGL(select_mask_kernel, dim3((unsigned)B), dim3(kTNT), 0, (hipStream_t)stream, mask, random_mask,
                     (unsigned long long)seed, select, n_select, N, out);
  return check_launch("select_mask");
}

int frcnn_select_mask(const uint8_t *mask, const int32_t *random_mask, const int32_t *select,
                      int n_select, int B, int N, uint8_t *out, frcnn_stream_t stream) {
  FRCNN_REQUIRE(mask && random_mask && select && out, "select_mask: null pointer");
  return select_mask_impl(mask, random_mask, 0, select, n_select, B, N, out, stream);
}

int frcnn_select_mask_seeded(const uint8_t *mask, const int32_t *select, int n_select, uint64_t seed, int B, int N,
                             uint8_t *out, frcnn_stream_t stream) {
  FRCNN_REQUIRE(mask && select && out, "select_mask_seeded: null pointer");
  return select_mask_impl(mask, nullptr, seed, select, n_select, B, N, out, stream);
}

size_t frcnn_rpn_targets_workspace_bytes(int B, int N, int M) {
  if (B < 0 || N < 0 || M < 0) return 0;
  size_t off = 0;
  off += round_up((size_t)B * N * 4, 256);                      // max_iou
  off += round_up((size_t)B * N * 4, 256);                      // argmax_row
  off += round_up((size_t)B * (M > 0 ? M : 1) * 4, 256);        // argmax_col
  off += frcnn_iou_reduce_workspace_bytes(B, N, M > 0 ? M : 1); // column keys
  return off + 256;
}

int frcnn_rpn_targets(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                      const int32_t *rnd_pos, const int32_t *rnd_neg, uint64_t seed, int B, int N,
                      int M, int total_pos, int total_neg, const float *variances,
                      float *out_deltas, float *out_labels, void *workspace,
                      size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(anchors && gt_boxes && gt_labels && variances && out_deltas && out_labels, "rpn_targets: null pointer");
  FRCNN_REQUIRE((rnd_pos == nullptr) == (rnd_neg == nullptr), "rpn_targets: give both random tensors or neither");
  FRCNN_REQUIRE(B >= 0 && N > 0 && M > 0, "rpn_targets: bad shape B=%d N=%d M=%d", B, N, M);
  FRCNN_REQUIRE(N <= kMaxTargetsN, "rpn_targets: N=%d anchors exceeds the limit %d", N, kMaxTargetsN);
  FRCNN_REQUIRE(total_pos >= 0 && total_neg >= 0, "rpn_targets: negative sample counts");
  FRCNN_REQUIRE(aligned_to(anchors, 16) && aligned_to(gt_boxes, 16) && aligned_to(out_deltas, 16),
                "rpn_targets: box pointers must be 16-byte aligned");
  FRCNN_REQUIRE(workspace && aligned_to(workspace, 256) && workspace_bytes >= frcnn_rpn_targets_workspace_bytes(B, N, M),
                "rpn_targets: workspace must be non-null, 256-byte aligned and >= %zu bytes", frcnn_rpn_targets_workspace_bytes(B, N, M));
  if (B == 0) return FRCNN_OK;
  unsigned char *p = reinterpret_cast<unsigned char *>(workspace);
  size_t off = 0;
  float *max_iou = reinterpret_cast<float *>(p + off);
  off += round_up((size_t)B * N * 4, 256);
  int *argmax_row = reinterpret_cast<int *>(p + off);
  off += round_up((size_t)B * N * 4, 256);
  int *argmax_col = reinterpret_cast<int *>(p + off);
  off += round_up((size_t)B * M * 4, 256);
  void *keys = p + off;
  const size_t keys_bytes = frcnn_iou_reduce_workspace_bytes(B, N, M);
  int rc = iou_map_reduce_impl(anchors, 0, gt_boxes, B, N, M, max_iou, argmax_row, argmax_col, keys, keys_bytes,
                               (hipStream_t)stream, false);
  if (rc) return rc;
  const int words = ((N + 63) >> 6) << 1;
  size_t lds = (size_t)5 * words * 4;
  // the selections' keys, the negatives' random integers, the counting bins and the list of positives in LDS when they fit
  const size_t cache_bytes = (size_t)N * 8 + 4096 * 4 + (size_t)kMaxPosList * 4;
  const int cache_keys = (lds + cache_bytes <= 128 * 1024) ? 1 : 0;
  if (cache_keys) lds += cache_bytes;
  auto kern = rpn_targets_kernel;
  if (lds > 48 * 1024) {
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "rpn_targets: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(kTNT), lds, (hipStream_t)stream, anchors, gt_boxes, gt_labels, rnd_pos,
                     rnd_neg, (unsigned long long)seed, N, M, total_pos, total_neg, variances[0], variances[1],
                     variances[2], variances[3], max_iou, argmax_row, reinterpret_cast<const unsigned long long *>(keys),
                     out_deltas, out_labels, cache_keys);
  return check_launch("rpn_targets");
}

int frcnn_rpn_targets_static(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                             const int32_t *rnd_pos, const int32_t *rnd_neg, uint64_t seed, int B, int N,
                             int M, int total_pos, int total_neg, const float *variances,
                             float *out_deltas, float *out_labels, frcnn_stream_t stream) {
  FRCNN_REQUIRE(B >= 0 && N > 0 && M > 0, "rpn_targets_static: bad shape B=%d N=%d M=%d", B, N, M);
  const size_t need = frcnn_rpn_targets_workspace_bytes(B, N, M);
  if (need > kStaticScratchBytes)
    return set_error(FRCNN_EUNSUPPORTED,
                     "rpn_targets_static: B=%d N=%d M=%d needs %zu bytes of scratch, the static buffer holds %zu "
                     "(use frcnn_rpn_targets with a caller workspace)", B, N, M, need, kStaticScratchBytes);
  void *ptr = nullptr;
  int rc = static_scratch(&ptr);
  if (rc) return rc;
  // the shared kernels take the scratch as an argument, exactly as in frcnn_rpn_targets; what must start at zero
  // (the column keys) is cleared by frcnn_iou_map_reduce on `stream` in every call
  return frcnn_rpn_targets(anchors, gt_boxes, gt_labels, rnd_pos, rnd_neg, seed, B, N, M, total_pos, total_neg, variances,
                           out_deltas, out_labels, ptr, kStaticScratchBytes, stream);
}

size_t frcnn_roi_delta_workspace_bytes(int B, int R, int M) {
  (void)M;
  if (B < 0 || R < 0) return 0;
  return 2 * round_up((size_t)B * (R > 0 ? R : 1) * 4, 256) + 256;  // max_iou, argmax_row
}

int frcnn_roi_delta(const float *rois, const float *gt_boxes, const int32_t *gt_labels, const int32_t *rnd_pos,
                    const int32_t *rnd_neg, uint64_t seed, int B, int R, int M, int total_labels, int total_pos,
                    int total_neg, const float *variances, float *out_deltas, float *out_labels, void *workspace,
                    size_t workspace_bytes, frcnn_stream_t stream) {
  FRCNN_REQUIRE(rois && gt_boxes && gt_labels && variances && out_deltas && out_labels, "roi_delta: null pointer");
  FRCNN_REQUIRE((rnd_pos == nullptr) == (rnd_neg == nullptr), "roi_delta: give both random tensors or neither");
  FRCNN_REQUIRE(B >= 0 && R >= 0 && M > 0 && total_labels > 0, "roi_delta: bad shape B=%d R=%d M=%d L=%d", B, R, M, total_labels);
  FRCNN_REQUIRE(total_labels <= 1024, "roi_delta: total_labels=%d exceeds 1024", total_labels);
  FRCNN_REQUIRE(total_pos >= 0 && total_neg >= 0, "roi_delta: negative sample counts");
  FRCNN_REQUIRE(aligned_to(rois, 16) && aligned_to(gt_boxes, 16) && aligned_to(out_deltas, 16),
                "roi_delta: box pointers must be 16-byte aligned");
  if (R > kMaxRoiDeltaR)
    return set_error(FRCNN_EUNSUPPORTED, "roi_delta: R=%d RoIs per image exceeds the LDS limit %d", R, kMaxRoiDeltaR);
  FRCNN_REQUIRE(workspace && aligned_to(workspace, 256) && workspace_bytes >= frcnn_roi_delta_workspace_bytes(B, R, M),
                "roi_delta: workspace must be non-null, 256-byte aligned and >= %zu bytes", frcnn_roi_delta_workspace_bytes(B, R, M));
  if (B == 0 || R == 0) return FRCNN_OK;
  unsigned char *p = reinterpret_cast<unsigned char *>(workspace);
  float *max_iou = reinterpret_cast<float *>(p);
  int *argmax_row = reinterpret_cast<int *>(p + round_up((size_t)B * R * 4, 256));
  // :215-217 RoI x gt IoU, row max and first argmax (RoIs are per image: batched boxes); no map is written
  int rc = frcnn_iou_map_reduce(rois, 1, gt_boxes, B, R, M, max_iou, argmax_row, nullptr, nullptr, 0, stream);
  if (rc) return rc;
  const int words = (R + 31) >> 5;
  const size_t lds = (size_t)((2 * words + 3) & ~3) * 4 + (size_t)R * 24 + 4096 * 4;
  auto kern = roi_delta_kernel;
  if (lds > 48 * 1024) {
    rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, "roi_delta: hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(kTNT), lds, (hipStream_t)stream, rois, gt_boxes, gt_labels, rnd_pos, rnd_neg,
                     (unsigned long long)seed, R, M, total_labels, total_pos, total_neg, variances[0], variances[1],
                     variances[2], variances[3], max_iou, argmax_row, out_deltas, out_labels);
  return check_launch("roi_delta");
}

}  // extern "C"
