// Stable least-significant-digit radix sort of (key, 32-bit value) pairs for gfx950: one kernel per 8-bit digit, each
// workgroup ranking a tile of 4096 pairs and finding its global write positions by a chained scan over the tiles before
// it ("Onesweep", Adinets & Merrill 2022).  Written for the two sorts of the binning -- about a million depth keys and
// a few million (camera | tile) keys -- which are latency-bound chains of small launches:
//   * ONE histogram kernel counts every digit of every pass and clears the chained-scan state of all passes, so a pass
//     is a single launch with no fill in front of it (a library sort clears its state before each pass: two 5 us fills
//     per pass, nine in a four-pass sort of a million keys);
//   * the first pass reads its keys and values through functors (depth bits straight from the projection's output, the
//     index sequence from the position), so no kernel materialises them; the last pass can skip writing keys;
//   * the number of valid items may live in device memory (the caller sized the buffers before the count was known).
// The chained scan spins on words other workgroups publish with device-scope atomics.  A workgroup takes its tile
// number from a ticket counter, so every tile it waits for belongs to a workgroup that started earlier and publishes
// its count before it waits for anything itself: the wait always ends.
#pragma once
#include "common.h"

namespace pr {
namespace rs {

// 16 pairs per thread: measured against 12 and 8 (more workgroups per CU, more tiles in the chain) at 6.4 M tile keys
// and 1 M depth keys: 0.272 / 0.284 / 0.303 ms for the tile sort's entry point, 0.115 / 0.125 / 0.132 for the depth order
constexpr int THREADS = 256, WAVES = THREADS / WAVE, BINS = 256, ITEMS = 16, TILE = THREADS * ITEMS;
constexpr int MAX_PASSES = 4, MAX_HIST_BLOCKS = 256, MAX_GROUPS = 4;   // MAX_HIST_BLOCKS / ROWS_PER_GROUP
constexpr int64_t HIST_CHUNK = TILE;   // keys a histogram workgroup requests together
constexpr uint32_t HAS_AGGREGATE = 1u << 30, HAS_PREFIX = 2u << 30, COUNT_MASK = (1u << 30) - 1u;
constexpr int64_t MAX_ITEMS = (int64_t)COUNT_MASK;   // counts travel in 30 bits next to the two flags

template <class T>
struct FromArray {
  const T *p;
  __device__ __forceinline__ T operator()(int64_t i) const { return p[i]; }
};
struct Iota {
  __device__ __forceinline__ uint32_t operator()(int64_t i) const { return (uint32_t)i; }
};

struct Layout {
  size_t partials, offsets, tickets, state, state_words, total;
  int hist_blocks;
  int64_t n_tiles;
};

inline Layout layout(int64_t cap, int passes) {
  Layout L;
  L.n_tiles = (cap + TILE - 1) / TILE;
  int64_t hb = (cap + HIST_CHUNK - 1) / HIST_CHUNK;
  L.hist_blocks = (int)(hb < 1 ? 1 : hb > MAX_HIST_BLOCKS ? MAX_HIST_BLOCKS : hb);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  L.partials = take((size_t)L.hist_blocks * passes * BINS * 4);
  L.offsets = take((size_t)passes * MAX_GROUPS * BINS * 4);
  L.tickets = take(256);
  L.state_words = (size_t)passes * (size_t)L.n_tiles * BINS;
  L.state = take(L.state_words * 4);
  L.total = off;
  return L;
}

__device__ __forceinline__ int64_t count_of(int64_t cap, const int64_t *__restrict__ n_dev) {
  if (!n_dev) return cap;
  const int64_t n = *n_dev;
  return n < cap ? n : cap;
}

// Digit counts of every pass over a contiguous share of the keys -> this workgroup's row of `partials`; also clears
// the chained-scan state words of all passes.
template <class KeyFn>
__global__ void __launch_bounds__(THREADS) hist_kernel(KeyFn keys, int64_t cap, const int64_t *__restrict__ n_dev,
                                                       int passes, uint32_t *__restrict__ partials,
                                                       uint32_t *__restrict__ state, int64_t state_words) {
  __shared__ uint32_t h[MAX_PASSES * BINS];
  const int tid = threadIdx.x;
  const int64_t n = count_of(cap, n_dev);
  for (int i = tid; i < passes * BINS; i += THREADS) h[i] = 0;
  uint4 *s4 = (uint4 *)state;
  for (int64_t i = (int64_t)blockIdx.x * THREADS + tid; i < state_words / 4; i += (int64_t)gridDim.x * THREADS)
    s4[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  int64_t per = (n + gridDim.x - 1) / gridDim.x;
  per = (per + HIST_CHUNK - 1) / HIST_CHUNK * HIST_CHUNK;
  const int64_t begin = (int64_t)blockIdx.x * per;
  const int64_t end = begin + per < n ? begin + per : n;
  // ITEMS keys per thread are requested before any is counted: the loop costs one memory round trip per chunk, not
  // one per key
  for (int64_t chunk = begin; chunk < end; chunk += HIST_CHUNK) {
    uint32_t k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int64_t i = chunk + r * THREADS + tid;
      k[r] = i < end ? (uint32_t)keys(i) : 0u;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      if (chunk + r * THREADS + tid < end) {
#pragma unroll
        for (int p = 0; p < MAX_PASSES; ++p)
          if (p < passes) atomicAdd(&h[p * BINS + ((k[r] >> (8 * p)) & 255u)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < passes * BINS; i += THREADS) partials[(size_t)blockIdx.x * passes * BINS + i] = h[i];
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)v, d);
    if (lane >= d) v += up;
  }
  return v;
}

// Exclusive scan over the 256 digits of `total` (one per thread of a 256-thread group); `wave_tot` is 4 words of LDS.
__device__ __forceinline__ uint32_t digits_exclusive_scan(uint32_t total, int tid, uint32_t *wave_tot) {
  const int lane = tid & 63, w = tid >> 6;
  const uint32_t incl = wave_inclusive_scan(total, lane);
  if (lane == 63) wave_tot[w] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int j = 0; j < w; ++j) base += wave_tot[j];
  return base + incl - total;
}

// Sums the histogram rows, 64 per workgroup (16 per thread, requested together): `reduced` [pass][group][digit].  The
// pass kernels add the few groups up and scan the digits themselves.  Also resets the passes' ticket counters.
constexpr int ROWS_PER_GROUP = 64;
__global__ void __launch_bounds__(1024) reduce_kernel(int passes, int hist_blocks, const uint32_t *__restrict__ partials,
                                                      uint32_t *__restrict__ reduced, uint32_t *__restrict__ tickets) {
  __shared__ uint32_t red[4][BINS];
  const int tid = threadIdx.x, p = blockIdx.x, q = blockIdx.y, d = tid & 255, slice = tid >> 8;
  uint32_t v[ROWS_PER_GROUP / 4];
#pragma unroll
  for (int i = 0; i < ROWS_PER_GROUP / 4; ++i) {
    const int b = q * ROWS_PER_GROUP + slice + 4 * i;
    v[i] = b < hist_blocks ? partials[((size_t)b * passes + p) * BINS + d] : 0u;
  }
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < ROWS_PER_GROUP / 4; ++i) sum += v[i];
  red[slice][d] = sum;
  __syncthreads();
  if (tid < BINS) reduced[((size_t)p * gridDim.y + q) * BINS + d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
  if (q == 0 && tid == 0) tickets[p] = 0;
}

// Lanes of the wavefront whose 8-bit digit equals this lane's.
__device__ __forceinline__ unsigned long long same_digit_lanes(uint32_t d) {
  unsigned long long m = ~0ull;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

__device__ __forceinline__ uint32_t load_state(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_state(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One pass: pairs ordered by the 8-bit digit at `shift`, equal digits keeping their input order.
template <class KeyT, class KeyFn, class ValFn, bool WRITE_KEYS>
__global__ void __launch_bounds__(THREADS) pass_kernel(KeyFn keys_in, ValFn vals_in, int64_t cap,
                                                       const int64_t *__restrict__ n_dev, int shift,
                                                       const uint32_t *__restrict__ reduced, int groups,
                                                       uint32_t *__restrict__ ticket, uint32_t *__restrict__ state,
                                                       KeyT *__restrict__ keys_out,
                                                       uint32_t *__restrict__ vals_out) {
  __shared__ uint32_t wave_cnt[WAVES][BINS];   // per wavefront and digit: running count, then the wavefront's offset
  __shared__ uint32_t local_start[BINS];       // first slot of the digit in the tile's sorted order
  __shared__ uint32_t out_base[BINS];          // global position of the digit's first pair of this tile - local_start
  __shared__ KeyT lkeys[TILE];
  __shared__ uint32_t lvals[TILE];
  __shared__ uint32_t wave_tot[WAVES], wave_tot_all[WAVES];
  __shared__ uint32_t tile_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t n = count_of(cap, n_dev);
  uint32_t all = 0;   // pairs with digit `tid` in the whole input
  for (int q = 0; q < groups; ++q) all += reduced[q * BINS + tid];
  if (tid == 0) tile_s = atomicAdd(ticket, 1u);
#pragma unroll
  for (int j = 0; j < WAVES; ++j) wave_cnt[j][tid] = 0;
  __syncthreads();
  const int64_t tile = tile_s;
  const int64_t base = tile * TILE;
  if (base >= n) return;   // the whole workgroup: tiles past the true count have nothing to do and nobody waits for them
  const int n_here = (int)(n - base < TILE ? n - base : TILE);
  const uint32_t first_of_digit = digits_exclusive_scan(all, tid, wave_tot_all);

  // the tile in input order: wavefront w holds pairs [w * 64 * ITEMS, (w + 1) * 64 * ITEMS), round r the 64 at r * 64
  KeyT key[ITEMS];
  uint32_t val[ITEMS], rank[ITEMS];
  const int64_t first = base + (int64_t)w * (WAVE * ITEMS) + lane;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t i = first + r * WAVE;
    const bool ok = i < n;
    key[r] = ok ? (KeyT)keys_in(i) : (KeyT) ~(KeyT)0;   // padding sorts behind every real pair of the (last) tile
    val[r] = ok ? (uint32_t)vals_in(i) : 0u;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t d = ((uint32_t)key[r] >> shift) & 255u;
    const unsigned long long m = same_digit_lanes(d);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t prev = wave_cnt[w][d];
    if (lane == 63 - __builtin_clzll(m)) wave_cnt[w][d] = prev + (uint32_t)__builtin_popcountll(m);
    rank[r] = prev + below;
    // the next round reads what this round's leader lanes stored: keep the LDS accesses of the unrolled rounds in
    // program order (a wavefront's LDS operations complete in issue order; this stops the compiler moving them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();

  // thread t <-> digit t
  const uint32_t c0 = wave_cnt[0][tid], c1 = wave_cnt[1][tid], c2 = wave_cnt[2][tid], c3 = wave_cnt[3][tid];
  const uint32_t total = c0 + c1 + c2 + c3;
  wave_cnt[0][tid] = 0;
  wave_cnt[1][tid] = c0;
  wave_cnt[2][tid] = c0 + c1;
  wave_cnt[3][tid] = c0 + c1 + c2;
  uint32_t *mine = state + (size_t)tile * BINS + tid;
  store_state(mine, total | (tile == 0 ? HAS_PREFIX : HAS_AGGREGATE));
  const uint32_t lstart = digits_exclusive_scan(total, tid, wave_tot);
  local_start[tid] = lstart;
  // pairs with this digit in the tiles before this one: walk back until a tile that knows its own prefix; the words
  // of LOOK tiles are requested together, so a long walk costs one memory round trip per LOOK tiles
  uint32_t before = 0;
  constexpr int LOOK = 8;
  bool done = tile == 0;
  for (int64_t j = tile - 1; !done; j -= LOOK) {
    uint32_t s[LOOK];
#pragma unroll
    for (int i = 0; i < LOOK; ++i) s[i] = j - i >= 0 ? load_state(state + (size_t)(j - i) * BINS + tid) : HAS_PREFIX;
#pragma unroll
    for (int i = 0; i < LOOK; ++i) {
      if (done) break;
      while ((s[i] >> 30) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        s[i] = load_state(state + (size_t)(j - i) * BINS + tid);
      }
      before += s[i] & COUNT_MASK;
      done = (s[i] & HAS_PREFIX) != 0u;
    }
  }
  if (tile > 0) store_state(mine, (before + total) | HAS_PREFIX);
  out_base[tid] = first_of_digit + before - lstart;
  __syncthreads();

#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t d = ((uint32_t)key[r] >> shift) & 255u;
    const uint32_t pos = local_start[d] + wave_cnt[w][d] + rank[r];
    lkeys[pos] = key[r];
    lvals[pos] = val[r];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = k * THREADS + tid;
    if (i < n_here) {
      const KeyT kk = lkeys[i];
      const uint32_t dst = out_base[((uint32_t)kk >> shift) & 255u] + (uint32_t)i;
      if (WRITE_KEYS) keys_out[dst] = kk;
      vals_out[dst] = lvals[i];
    }
  }
}

// Sorts `cap` pairs (or *n_dev of them, if given and smaller) by the low `bits` bits of the key.  The first pass reads
// keys0(i), vals0(i); later passes ping-pong between (kbuf[0], vbuf[0]) and (kbuf[1], vbuf[1]); the last pass writes its
// values to `vals_final` and, if `keys_final` is given, its keys there.  `temp`: layout(cap, passes).total bytes.
template <class KeyT, class KeyFn0, class ValFn0>
int sort_pairs(KeyFn0 keys0, ValFn0 vals0, int64_t cap, const int64_t *n_dev, int bits, KeyT *const kbuf[2],
               uint32_t *const vbuf[2], KeyT *keys_final, uint32_t *vals_final, char *temp, hipStream_t st) {
  if (cap <= 0) return PR_OK;
  const int passes = bits <= 0 ? 1 : (bits + 7) / 8;
  if (passes > MAX_PASSES || passes * 8 > (int)sizeof(KeyT) * 8 || cap > MAX_ITEMS) return PR_EUNSUPPORTED;
  const Layout L = layout(cap, passes);
  uint32_t *partials = (uint32_t *)(temp + L.partials), *offsets = (uint32_t *)(temp + L.offsets);
  uint32_t *tickets = (uint32_t *)(temp + L.tickets), *state = (uint32_t *)(temp + L.state);
  hipLaunchKernelGGL(hist_kernel<KeyFn0>, dim3(L.hist_blocks), dim3(THREADS), 0, st, keys0, cap, n_dev, passes, partials,
                     state, (int64_t)L.state_words);
  const int groups = (L.hist_blocks + ROWS_PER_GROUP - 1) / ROWS_PER_GROUP;
  hipLaunchKernelGGL(reduce_kernel, dim3(passes, groups), dim3(1024), 0, st, passes, L.hist_blocks,
                     (const uint32_t *)partials, offsets, tickets);
  const dim3 grid((unsigned)L.n_tiles);
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    KeyT *ko = last ? keys_final : kbuf[p & 1];
    uint32_t *vo = last ? vals_final : vbuf[p & 1];
    const uint32_t *off_p = offsets + (size_t)p * groups * BINS;
    uint32_t *state_p = state + (size_t)p * L.n_tiles * BINS;
    if (p == 0) {
      if (ko)
        hipLaunchKernelGGL((pass_kernel<KeyT, KeyFn0, ValFn0, true>), grid, dim3(THREADS), 0, st, keys0, vals0, cap, n_dev,
                           0, off_p, groups, tickets + p, state_p, ko, vo);
      else
        hipLaunchKernelGGL((pass_kernel<KeyT, KeyFn0, ValFn0, false>), grid, dim3(THREADS), 0, st, keys0, vals0, cap,
                           n_dev, 0, off_p, groups, tickets + p, state_p, ko, vo);
    } else {
      const FromArray<KeyT> ki{kbuf[(p - 1) & 1]};
      const FromArray<uint32_t> vi{vbuf[(p - 1) & 1]};
      if (ko)
        hipLaunchKernelGGL((pass_kernel<KeyT, FromArray<KeyT>, FromArray<uint32_t>, true>), grid, dim3(THREADS), 0, st, ki,
                           vi, cap, n_dev, 8 * p, off_p, groups, tickets + p, state_p, ko, vo);
      else
        hipLaunchKernelGGL((pass_kernel<KeyT, FromArray<KeyT>, FromArray<uint32_t>, false>), grid, dim3(THREADS), 0, st,
                           ki, vi, cap, n_dev, 8 * p, off_p, groups, tickets + p, state_p, ko, vo);
    }
  }
  return check_launch("rs::sort_pairs");
}

}  // namespace rs
}  // namespace pr
