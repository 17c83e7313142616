// Tile-intersection binning for gfx950: per-Gaussian tile counts + prefix sum, depth-ordered key/value
// emission, stable device radix sorts (rocPRIM, restricted to the significant key bits), per-tile offsets.
//
// Integer work; results must be bit-identical to the CPU restatement, so the tile rectangle is
// computed with true float32 divisions (no reciprocal multiply that could be contracted into an FMA).
#include <cstring>
#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <mutex>

#include "common.h"
#include "onesweep.h"

namespace pr {

struct TileRect {
  int x0, y0, x1, y1;
};

__device__ __forceinline__ TileRect tile_rect(float mx, float my, int radius, float ts, int tile_w, int tile_h) {
  float tr = (float)radius / ts;
  float tx = mx / ts, ty = my / ts;
  TileRect r;
  r.x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tile_w);
  r.y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)tile_h);
  r.x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tile_w);
  r.y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)tile_h);
  return r;
}

__global__ void __launch_bounds__(256) isect_count_kernel(int64_t total, const float *__restrict__ means2d,
                                                          const int32_t *__restrict__ radii, float ts, int tile_w,
                                                          int tile_h, int32_t *__restrict__ tiles_per_gauss) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int r = radii[idx];
  int cnt = 0;
  if (r > 0) {
    TileRect t = tile_rect(means2d[2 * idx], means2d[2 * idx + 1], r, ts, tile_w, tile_h);
    cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
  }
  tiles_per_gauss[idx] = cnt;
}

__global__ void __launch_bounds__(256) isect_offsets_kernel(int64_t n_isects, const int64_t *__restrict__ isect_ids,
                                                            int n_tiles_total, int tile_bits, int n_tiles,
                                                            int32_t *__restrict__ offsets) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_isects) return;
  int64_t hi = isect_ids[k] >> 32;
  int cur = (int)(hi >> tile_bits) * n_tiles + (int)(hi & ((1ll << tile_bits) - 1));
  if (k == 0) {
    for (int t = 0; t <= cur; ++t) offsets[t] = 0;
  } else {
    int64_t hp = isect_ids[k - 1] >> 32;
    int prev = (int)(hp >> tile_bits) * n_tiles + (int)(hp & ((1ll << tile_bits) - 1));
    for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)k;
  }
  if (k == n_isects - 1)
    for (int t = cur + 1; t < n_tiles_total; ++t) offsets[t] = (int32_t)n_isects;
}

// ---- footprint-tested ("tight") binning --------------------------------------------------------------
// The reference bins a Gaussian into every tile its square 3-sigma bounding box touches.  About half of
// those (Gaussian, tile) pairs cannot reach alpha >= 1/255 at any pixel of the tile (elongated or faint
// Gaussians, box corners); the compositor skips them pixel by pixel, so dropping them at binning time leaves
// every rendered value unchanged while the sort, the lists and the staging loads shrink.  The test is the
// one the compositor applies per wave region, here against the whole tile; count and emission call the SAME
// out-of-line function so that they cannot disagree by a rounding.
__device__ __noinline__ bool touches_tile(float gx, float gy, float o, float a, float b, float c, float rx0,
                                          float ry0, float rx1, float ry1) {
  return touches_region(gx, gy, o, a, b, c, rx0, ry0, rx1, ry1);
}

// Candidate tiles of one Gaussian: gsplat's bounding-box rectangle, clipped to the axis-aligned box of the
// alpha = 1/255 level set (half extents sqrt(2 t c / det), sqrt(2 t a / det), t = ln(255 o), slightly
// enlarged so that it never excludes a tile the exact test would keep).  Out of line for the same reason as
// touches_tile: count and emission must see identical rectangles.
struct Candidates {
  int x0, y0, w, cnt;
};
__device__ __noinline__ Candidates candidate_rect(float gx, float gy, int rad, float o, float a, float b, float c,
                                                  float ts, int tile_w, int tile_h) {
  Candidates r = {0, 0, 1, 0};
  TileRect t = tile_rect(gx, gy, rad, ts, tile_w, tile_h);
  if (a > 0.f && c > 0.f && a * c - b * b > 0.f) {
    const float lvl = __logf(255.f * o) + 4e-3f;
    if (!(lvl > 0.f)) return r;  // cannot reach 1/255 anywhere
    const float det = a * c - b * b;
    const float ex = sqrtf(2.f * lvl * c / det) * 1.0001f + 1e-3f, ey = sqrtf(2.f * lvl * a / det) * 1.0001f + 1e-3f;
    // tile x holds pixel centres [x ts + 0.5, x ts + ts - 0.5]
    const float xl = ceilf((gx - ex + 0.5f) / ts - 1.f), xh = floorf((gx + ex - 0.5f) / ts);
    const float yl = ceilf((gy - ey + 0.5f) / ts - 1.f), yh = floorf((gy + ey - 0.5f) / ts);
    t.x0 = max(t.x0, (int)fmaxf(xl, 0.f));
    t.y0 = max(t.y0, (int)fmaxf(yl, 0.f));
    t.x1 = min(t.x1, (int)fminf(xh, (float)tile_w) + 1);
    t.y1 = min(t.y1, (int)fminf(yh, (float)tile_h) + 1);
  }
  if (t.x1 > t.x0 && t.y1 > t.y0) {
    r.x0 = t.x0;
    r.y0 = t.y0;
    r.w = t.x1 - t.x0;
    r.cnt = r.w * (t.y1 - t.y0);
  }
  return r;
}

// Lanes of one wavefront exchange data through LDS without a workgroup barrier (LDS operations of a wave
// complete in order); this only stops the COMPILER from keeping a lane's own store in a register or moving
// accesses across the exchange point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-wave LDS records for the cooperative enumeration: the 64 Gaussians of a wave, with candidates.
struct alignas(16) WaveRecords {
  float4 ga[WAVE];     // gx, gy, opacity, conic a
  float4 gb[WAVE];     // conic b, conic c, 1/w, bit_cast(x0 | y0 << 16)
  int2 gc[WAVE];       // first candidate number, w
  uint2 gd[WAVE];      // camera bits of the key (or the camera's first flat tile index), flat Gaussian index
  int owner[WAVE];     // scratch for the owner search of one chunk of 64 candidates
};

// Lane's Gaussian `idx` (nothing when !valid) -> its LDS record; returns the wave's candidate total.
__device__ __forceinline__ int load_records(WaveRecords &R, bool valid, int64_t idx, uint32_t key_hi,
                                            const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                                            const float *__restrict__ conics, const float *__restrict__ opacities,
                                            float ts, int tile_w, int tile_h, int lane, int &cnt, int &cstart) {
  float gx = 0.f, gy = 0.f, o = 0.f, a = 0.f, b = 0.f, c = 0.f;
  Candidates r = {0, 0, 1, 0};
  if (valid) {
    const int rad = radii[idx];
    if (rad > 0) {
      gx = means2d[2 * idx];
      gy = means2d[2 * idx + 1];
      a = conics[3 * idx];
      b = conics[3 * idx + 1];
      c = conics[3 * idx + 2];
      o = opacities[idx];
      r = candidate_rect(gx, gy, rad, o, a, b, c, ts, tile_w, tile_h);
    }
  }
  cnt = r.cnt;
  int incl = cnt;  // inclusive scan over the 64 lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  cstart = incl - cnt;
  R.ga[lane] = make_float4(gx, gy, o, a);
  R.gb[lane] = make_float4(b, c, 1.f / (float)r.w, __int_as_float(r.x0 | (r.y0 << 16)));
  R.gc[lane] = make_int2(cstart, r.w);
  R.gd[lane] = make_uint2(key_hi, (uint32_t)idx);
  wave_lds_sync();
  return __shfl(incl, 63);
}

// One chunk of 64 candidates starting at c0: which Gaussian (record j) and which tile each lane's candidate
// is, and whether it passes the footprint test.  `carry` is the owner of the candidate just before c0.
// Owners are found by having every Gaussian whose first candidate lies in the chunk mark that position and
// taking a running maximum over the lanes -- two LDS round trips instead of a six-step search.
__device__ __forceinline__ bool test_chunk(WaveRecords &R, int c0, int total, int lane, int cnt, int cstart, float ts,
                                           int &carry, int &j, int &tile_x, int &tile_y) {
  R.owner[lane] = 0;
  wave_lds_sync();
  const int rel = cstart - c0;
  if (cnt > 0 && rel >= 0 && rel < WAVE) R.owner[rel] = lane + 1;
  wave_lds_sync();
  int own = R.owner[lane];
  wave_lds_sync();
  // inclusive running maximum over lanes 0..lane (markers increase with the lane that wrote them)
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x111, 0xf, 0xf, true));  // row_shr:1
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x112, 0xf, 0xf, true));  // row_shr:2
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x114, 0xf, 0xf, true));  // row_shr:4
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x118, 0xf, 0xf, true));  // row_shr:8
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x142, 0xa, 0xf, true));  // row_bcast:15 -> rows 1,3
  own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x143, 0xc, 0xf, true));  // row_bcast:31 -> rows 2,3
  j = own > 0 ? own - 1 : carry;
  carry = __builtin_amdgcn_readlane(j, 63);
  const int c = c0 + lane;
  const float4 ga = R.ga[j], gb = R.gb[j];
  const int2 gc = R.gc[j];
  const int t = c - gc.x;
  const int ty = (int)(((float)t + 0.5f) * gb.z);  // t / w: (t + 0.5) / w is never within 0.004 of an integer
  const int tx = t - ty * gc.y;
  const int xy = __float_as_int(gb.w);
  tile_x = (xy & 0xffff) + tx;
  tile_y = (xy >> 16) + ty;
  if (c >= total) return false;
  const float rx0 = (float)tile_x * ts + 0.5f, ry0 = (float)tile_y * ts + 0.5f;
  return touches_tile(ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, rx0, ry0, rx0 + ts - 1.f, ry0 + ts - 1.f);
}

__global__ void __launch_bounds__(256) isect_count_tight_kernel(
    int64_t total, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ opacities, float ts, int tile_w, int tile_h,
    int32_t *__restrict__ tiles_per_gauss) {
  __shared__ WaveRecords recs[4];
  __shared__ int counts[256];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  WaveRecords &R = recs[threadIdx.x >> 6];
  int *mine = counts + (threadIdx.x & ~63);
  mine[lane] = 0;
  int cnt, cstart;
  const int n_cand = load_records(R, idx < total, idx, 0u, means2d, radii, conics, opacities, ts, tile_w, tile_h, lane,
                                  cnt, cstart);
  int carry = 0;
  for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
    int j, tx, ty;
    if (test_chunk(R, c0, n_cand, lane, cnt, cstart, ts, carry, j, tx, ty)) atomicAdd(&mine[j], 1);
  }
  if (idx < total) tiles_per_gauss[idx] = mine[lane];
}

// Footprint-tested emission: like isect_emit_sorted_kernel, but the wave enumerates the candidate tiles of its
// 64 Gaussians and writes only those that pass, compacted in candidate order.
template <class KeyT>
__global__ void __launch_bounds__(256) isect_emit_tight_kernel(
    int64_t total, int N, const uint32_t *__restrict__ order, const uint32_t *__restrict__ starts,
    const float *__restrict__ means2d, const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ opacities, const int32_t *__restrict__ tiles_per_gauss, float ts, int tile_w,
    int tile_h, int tile_bits, int64_t cap, const int64_t *__restrict__ n_dev, KeyT *__restrict__ keys,
    uint32_t *__restrict__ vals) {
  __shared__ WaveRecords recs[4];
  const int64_t n_isects = n_dev ? (*n_dev < cap ? *n_dev : cap) : cap;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  WaveRecords &R = recs[threadIdx.x >> 6];
  uint32_t idx = 0, start = 0;
  bool valid = false;
  if (r < total) {
    idx = order[r];
    start = starts[r];
    valid = tiles_per_gauss[idx] > 0;
  }
  int cnt, cstart;
  const int n_cand = load_records(R, valid, (int64_t)idx, (idx / (uint32_t)N) << tile_bits, means2d, radii, conics,
                                  opacities, ts, tile_w, tile_h, lane, cnt, cstart);
  int64_t pos = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)start, 0);
  int carry = 0;
  for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
    int j, tx, ty;
    const bool pass = test_chunk(R, c0, n_cand, lane, cnt, cstart, ts, carry, j, tx, ty);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
    if (pass) {
      const int64_t k = pos + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (k < n_isects) {
        const uint2 gd = R.gd[j];
        keys[k] = (KeyT)(gd.x | (uint32_t)(ty * tile_w + tx));
        vals[k] = gd.y;
      }
    }
    pos += __builtin_popcountll(m);
  }
}

// ---- binning straight into per-tile lists (the render path) ---------------------------------------------------
// What the compositor needs is, per tile, the flat indices of the Gaussians that can contribute, ordered by (depth
// bits, flat index).  The two-level sort above reaches that order with four radix passes over the C*N depth keys and
// two over the intersections, every one of them a small latency-bound launch.  Here instead:
//   1. tile_lists_count_kernel: the footprint-tested enumeration of isect_count_tight_kernel, which also bumps a
//      counter per (camera, tile);
//   2. tile_lists_scan_kernel (one workgroup): exclusive scan of the counters -> isect_offsets and the total; counters
//      reset to zero; launch order of the tiles (heaviest first);
//   3. tile_lists_emit_kernel: the same enumeration again; each kept pair takes the next free slot of its tile's list
//      (returning atomic on the tile's counter) and stores the 64-bit key depth_bits << 32 | flat_index there --
//      unordered within the tile;
//   4. tile_lists_sort_kernel: one workgroup per tile sorts its list in LDS (bitonic network on the 64-bit keys, which
//      are unique, so the result does not depend on the order the slots were handed out) and writes the indices.
// The order is the one a stable sort by (camera, tile, depth bits) of index-ordered intersections gives: gsplat's.
__global__ void __launch_bounds__(256) tile_lists_count_kernel(
    int64_t total, int N, int n_tiles, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ conics, const float *__restrict__ opacities, float ts, int tile_w, int tile_h,
    int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ tile_counts) {
  __shared__ WaveRecords recs[4];
  __shared__ int counts[256];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  WaveRecords &R = recs[threadIdx.x >> 6];
  int *mine = counts + (threadIdx.x & ~63);
  mine[lane] = 0;
  int cnt, cstart;
  const uint32_t tile_base = idx < total ? (uint32_t)(idx / N) * (uint32_t)n_tiles : 0u;
  const int n_cand = load_records(R, idx < total, idx, tile_base, means2d, radii, conics, opacities, ts, tile_w, tile_h,
                                  lane, cnt, cstart);
  int carry = 0;
  for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
    int j, tx, ty;
    if (test_chunk(R, c0, n_cand, lane, cnt, cstart, ts, carry, j, tx, ty)) {
      if (tiles_per_gauss) atomicAdd(&mine[j], 1);
      atomicAdd(&tile_counts[R.gd[j].x + (uint32_t)(ty * tile_w + tx)], 1);
    }
  }
  if (tiles_per_gauss && idx < total) tiles_per_gauss[idx] = mine[lane];
}

__device__ void tile_order_body(int n_tiles_total, const int32_t *__restrict__ offsets, int64_t n_isects,
                                int32_t *__restrict__ order);

// counts[t] -> offsets[t] (exclusive), *n_isects_dev = total, counts[t] = 0 (they become the emission's cursors),
// and, when asked for, the launch order of the tiles.  One workgroup of 1024 threads.
__global__ void __launch_bounds__(1024) tile_lists_scan_kernel(int n_tiles_total, int32_t *__restrict__ counts,
                                                               int32_t *__restrict__ offsets,
                                                               int64_t *__restrict__ n_isects_dev,
                                                               int32_t *__restrict__ order) {
  __shared__ long long part[1024];
  const int tid = threadIdx.x;
  const int chunk = (n_tiles_total + 1023) / 1024;
  const int lo = min(tid * chunk, n_tiles_total), hi = min(lo + chunk, n_tiles_total);
  long long sum = 0;
  for (int t = lo; t < hi; ++t) sum += counts[t];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {   // inclusive scan of the 1024 partial sums
    const long long up = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += up;
    __syncthreads();
  }
  const long long total = part[1023];
  long long run = part[tid] - sum;
  for (int t = lo; t < hi; ++t) {
    const int c = counts[t];
    offsets[t] = (int32_t)(run < 2147483647ll ? run : 2147483647ll);
    counts[t] = 0;
    run += c;
  }
  if (tid == 0) *n_isects_dev = total;
  if (order) {
    __threadfence_block();
    __syncthreads();
    tile_order_body(n_tiles_total, offsets, total, order);
  }
}

__global__ void __launch_bounds__(256) tile_lists_emit_kernel(
    int64_t total, int N, int n_tiles, const float *__restrict__ means2d, const int32_t *__restrict__ radii,
    const float *__restrict__ depths, const float *__restrict__ conics, const float *__restrict__ opacities, float ts,
    int tile_w, int tile_h, const int32_t *__restrict__ offsets, int32_t *__restrict__ cursors, int64_t cap,
    unsigned long long *__restrict__ keys) {
  __shared__ WaveRecords recs[4];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  WaveRecords &R = recs[threadIdx.x >> 6];
  int cnt, cstart;
  const uint32_t tile_base = idx < total ? (uint32_t)(idx / N) * (uint32_t)n_tiles : 0u;
  const int n_cand = load_records(R, idx < total, idx, tile_base, means2d, radii, conics, opacities, ts, tile_w, tile_h,
                                  lane, cnt, cstart);
  int carry = 0;
  for (int c0 = 0; c0 < n_cand; c0 += WAVE) {
    int j, tx, ty;
    if (test_chunk(R, c0, n_cand, lane, cnt, cstart, ts, carry, j, tx, ty)) {
      const uint2 gd = R.gd[j];
      const uint32_t tile = gd.x + (uint32_t)(ty * tile_w + tx);
      const int64_t pos = (int64_t)offsets[tile] + atomicAdd(&cursors[tile], 1);
      if (pos < cap)
        keys[pos] = ((unsigned long long)(uint32_t)__float_as_int(depths[gd.y]) << 32) | (unsigned long long)gd.y;
    }
  }
}

// Ascending bitonic network with every comparator pointing the same way (the first step of each merge pairs i with
// its mirror image in the block), so positions >= n behave as +infinity without being stored: a comparator whose
// upper end lies beyond n never swaps.
template <class Keys>
__device__ __forceinline__ void bitonic_sort_block(Keys keys, int n, int tid, int n_threads) {
  int lp = 0;
  while ((1 << lp) < n) ++lp;
  const int pairs = (1 << lp) >> 1;
  for (int lk = 1; lk <= lp; ++lk) {   // merge blocks of k = 2^lk; all strides are powers of two: shifts, no division
    const int k = 1 << lk, half = k >> 1;
    for (int i = tid; i < pairs; i += n_threads) {
      const int base = (i >> (lk - 1)) << lk, r = i & (half - 1);
      const int lo = base + r, hi = base + k - 1 - r;
      if (hi < n) {
        const unsigned long long a = keys[lo], b = keys[hi];
        if (a > b) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
    }
    __syncthreads();
    for (int lj = lk - 2; lj >= 0; --lj) {
      const int j = 1 << lj;
      for (int i = tid; i < pairs; i += n_threads) {
        const int lo = ((i >> lj) << (lj + 1)) + (i & (j - 1)), hi = lo + j;
        if (hi < n) {
          const unsigned long long a = keys[lo], b = keys[hi];
          if (a > b) {
            keys[lo] = b;
            keys[hi] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int TILE_SORT_LDS = 4096;   // list lengths sorted in LDS (32 KB); longer lists are sorted in global memory
__global__ void __launch_bounds__(256) tile_lists_sort_kernel(int n_tiles_total, const int32_t *__restrict__ offsets,
                                                              int64_t cap, const int64_t *__restrict__ n_isects_dev,
                                                              unsigned long long *__restrict__ keys,
                                                              int32_t *__restrict__ flatten_ids) {
  __shared__ unsigned long long buf[TILE_SORT_LDS];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int64_t n_all = *n_isects_dev < cap ? *n_isects_dev : cap;
  const int64_t begin = offsets[t] < n_all ? offsets[t] : n_all;
  int64_t end = t + 1 < n_tiles_total ? (int64_t)offsets[t + 1] : n_all;
  if (end > n_all) end = n_all;
  const int n = (int)(end - begin);
  if (n <= 0) return;
  unsigned long long *list = keys + begin;
  if (n <= TILE_SORT_LDS) {
    for (int i = tid; i < n; i += 256) buf[i] = list[i];
    __syncthreads();
    bitonic_sort_block(buf, n, tid, 256);
    for (int i = tid; i < n; i += 256) flatten_ids[begin + i] = (int32_t)(uint32_t)buf[i];
  } else {
    // rare: a list too long for LDS is sorted where it lies (the workgroup's own stores, made visible to its other
    // wavefronts by the barrier's fence; volatile keeps the compiler from caching elements in registers)
    volatile unsigned long long *g = list;
    __syncthreads();
    bitonic_sort_block(g, n, tid, 256);
    __threadfence_block();
    for (int i = tid; i < n; i += 256) flatten_ids[begin + i] = (int32_t)(uint32_t)g[i];
  }
}

// ---- depth order by sample sort (32 K .. 1.25 M Gaussians; opt-in, POGS_RAST_DEPTH_SORT=2) ------------------------
// Four radix passes over a million keys are latency-bound (each pass a chain of a few hundred look-back workgroups);
// a sample sort needs two passes over the data and no look-back.  Total order = (depth bits, flat index), i.e. exactly
// the order a stable sort by depth bits gives.  Measured at 1 M keys it is SLOWER than the radix sort (229 against
// 142 us): counting and scattering take 17 + 14 us as planned, but the three steps built on LDS bitonic networks are
// bound by the latency of their barrier steps (sample sort 82 us in one workgroup, offsets 59 us, bucket sorts 67 us).
// Kept as a cross-check of the radix order and as the starting point for a barrier-free bucket sort.
//   1. one workgroup sorts 8192 evenly spaced (key, index) samples in LDS; every fourth is a splitter (2047 of them);
//   2. every element finds its bucket (binary search over the splitters in LDS) and is counted per (workgroup, bucket);
//   3. one workgroup turns the counts into offsets;
//   4. the elements are scattered to their buckets (any order inside a bucket);
//   5. one workgroup per bucket sorts its ~N/2048 elements in LDS and writes the indices out.
// Four samples per bucket put the largest bucket at a few times the mean whatever the depth distribution is (equal depths
// included: the index is part of the key); a bucket over SS_CAP elements is still sorted correctly, in global memory.
constexpr int SS_BUCKETS = 2048, SS_SAMPLES = 8192, SS_CAP = 4096, SS_THREADS = 1024;
constexpr int64_t SS_MIN_TOTAL = 32768, SS_MAX_TOTAL = 1250000;

__device__ __forceinline__ unsigned long long depth_pair(const float *__restrict__ depths, int64_t i) {
  return ((unsigned long long)(uint32_t)__float_as_int(depths[i]) << 32) | (unsigned long long)(uint32_t)i;
}

__global__ void __launch_bounds__(SS_THREADS) ss_sample_kernel(int64_t total, const float *__restrict__ depths,
                                                               unsigned long long *__restrict__ splitters) {
  extern __shared__ unsigned long long ss_buf[];
  const int tid = threadIdx.x;
  for (int i = tid; i < SS_SAMPLES; i += SS_THREADS) ss_buf[i] = depth_pair(depths, (int64_t)i * total / SS_SAMPLES);
  __syncthreads();
  bitonic_sort_block(ss_buf, SS_SAMPLES, tid, SS_THREADS);
  for (int j = tid; j < SS_BUCKETS - 1; j += SS_THREADS) splitters[j] = ss_buf[(j + 1) * (SS_SAMPLES / SS_BUCKETS)];
}

// number of splitters <= x
__device__ __forceinline__ int ss_bucket_of(const unsigned long long *spl, unsigned long long x) {
  int lo = 0, hi = SS_BUCKETS - 1;  // answer in [lo, hi]
#pragma unroll
  for (int step = 0; step < 11; ++step) {
    const int mid = (lo + hi) >> 1;   // splitter index mid: is spl[mid] <= x ?
    const bool le = mid < SS_BUCKETS - 1 && spl[mid] <= x;
    lo = le ? mid + 1 : lo;
    hi = le ? hi : mid;
  }
  return lo;
}

__global__ void __launch_bounds__(SS_THREADS) ss_count_kernel(int64_t total, int64_t per_block,
                                                              const float *__restrict__ depths,
                                                              const unsigned long long *__restrict__ splitters,
                                                              uint16_t *__restrict__ bucket_ids,
                                                              uint32_t *__restrict__ counts) {
  __shared__ unsigned long long spl[SS_BUCKETS];
  __shared__ uint32_t hist[SS_BUCKETS];
  const int tid = threadIdx.x;
  for (int j = tid; j < SS_BUCKETS; j += SS_THREADS) {
    spl[j] = j < SS_BUCKETS - 1 ? splitters[j] : ~0ull;
    hist[j] = 0u;
  }
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * per_block, end = min(begin + per_block, total);
  for (int64_t i = begin + tid; i < end; i += SS_THREADS) {
    const int b = ss_bucket_of(spl, depth_pair(depths, i));
    bucket_ids[i] = (uint16_t)b;
    atomicAdd(&hist[b], 1u);
  }
  __syncthreads();
  for (int j = tid; j < SS_BUCKETS; j += SS_THREADS) counts[(size_t)blockIdx.x * SS_BUCKETS + j] = hist[j];
}

// counts[block][bucket] -> offset of that block's share inside the bucket; bucket_start[0 .. SS_BUCKETS]
__global__ void __launch_bounds__(SS_THREADS) ss_scan_kernel(int n_blocks, uint32_t *__restrict__ counts,
                                                             uint32_t *__restrict__ bucket_start) {
  __shared__ uint32_t wave_tot[SS_THREADS / 64];
  const int tid = threadIdx.x;
  uint32_t mine[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int bk = 2 * tid + k;
    uint32_t run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 16) {
      uint32_t c[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u] = b0 + u < n_blocks ? counts[(size_t)(b0 + u) * SS_BUCKETS + bk] : 0u;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (b0 + u < n_blocks) counts[(size_t)(b0 + u) * SS_BUCKETS + bk] = run;
        run += c[u];
      }
    }
    mine[k] = run;
  }
  // exclusive scan of the 2048 bucket totals: two per thread, wavefront scan, then the sixteen wavefront totals
  const uint32_t pair = mine[0] + mine[1];
  uint32_t inc = pair;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(inc, d);
    if (lane >= d) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wave_tot[w];
  const uint32_t excl = before + inc - pair;
  bucket_start[2 * tid] = excl;
  bucket_start[2 * tid + 1] = excl + mine[0];
  if (tid == SS_THREADS - 1) bucket_start[SS_BUCKETS] = excl + pair;
}

__global__ void __launch_bounds__(SS_THREADS) ss_scatter_kernel(int64_t total, int64_t per_block,
                                                                const float *__restrict__ depths,
                                                                const uint16_t *__restrict__ bucket_ids,
                                                                const uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ bucket_start,
                                                                unsigned long long *__restrict__ pairs) {
  __shared__ uint32_t cursor[SS_BUCKETS];
  const int tid = threadIdx.x;
  for (int j = tid; j < SS_BUCKETS; j += SS_THREADS)
    cursor[j] = bucket_start[j] + counts[(size_t)blockIdx.x * SS_BUCKETS + j];
  __syncthreads();
  const int64_t begin = (int64_t)blockIdx.x * per_block, end = min(begin + per_block, total);
  for (int64_t i = begin + tid; i < end; i += SS_THREADS) {
    const uint32_t pos = atomicAdd(&cursor[bucket_ids[i]], 1u);
    pairs[pos] = depth_pair(depths, i);
  }
}

__global__ void __launch_bounds__(256) ss_bucket_sort_kernel(const uint32_t *__restrict__ bucket_start,
                                                             unsigned long long *__restrict__ pairs,
                                                             int32_t *__restrict__ depth_order) {
  __shared__ unsigned long long buf[SS_CAP];
  const int tid = threadIdx.x;
  const uint32_t begin = bucket_start[blockIdx.x], end = bucket_start[blockIdx.x + 1];
  const int n = (int)(end - begin);
  if (n <= 0) return;
  unsigned long long *list = pairs + begin;
  if (n <= SS_CAP) {
    for (int i = tid; i < n; i += 256) buf[i] = list[i];
    __syncthreads();
    bitonic_sort_block(buf, n, tid, 256);
    for (int i = tid; i < n; i += 256) depth_order[begin + i] = (int32_t)(uint32_t)buf[i];
  } else {
    volatile unsigned long long *g = list;   // see tile_lists_sort_kernel
    __syncthreads();
    bitonic_sort_block(g, n, tid, 256);
    __threadfence_block();
    for (int i = tid; i < n; i += 256) depth_order[begin + i] = (int32_t)(uint32_t)g[i];
  }
}

struct SampleSortLayout {
  size_t splitters, bucket_ids, counts, bucket_start, pairs, total;
  int n_blocks;
  int64_t per_block;
};

struct ToI64 {
  __device__ __host__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static SampleSortLayout sample_sort_layout(int64_t total) {
  SampleSortLayout L;
  L.n_blocks = (int)std::min<int64_t>(256, std::max<int64_t>(1, (total + 8191) / 8192));
  L.per_block = (total + L.n_blocks - 1) / L.n_blocks;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align256(bytes);
    return o;
  };
  L.splitters = take((size_t)SS_BUCKETS * 8);
  L.bucket_ids = take((size_t)total * 2);
  L.counts = take((size_t)L.n_blocks * SS_BUCKETS * 4);
  L.bucket_start = take((size_t)(SS_BUCKETS + 1) * 4);
  L.pairs = take((size_t)total * 8);
  L.total = off + 256;
  return L;
}

static inline int bit_length(int64_t v) {
  int b = 0;
  while (v > 0) {
    ++b;
    v >>= 1;
  }
  return b;
}

}  // namespace pr

using namespace pr;

extern "C" size_t pr_isect_count_scratch_bytes(int C, int N) {
  size_t bytes = 0;
  int64_t total = (int64_t)C * N;
  if (total <= 0) return 256;
  auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, ToI64());
  (void)rocprim::inclusive_scan(nullptr, bytes, in, (int64_t *)nullptr, (size_t)total, rocprim::plus<int64_t>());
  return align256(bytes) + 256;
}

extern "C" int pr_isect_count(int C, int N, const float *means2d, const int32_t *radii, const float *conics,
                              const float *opacities, int tile_size, int tile_w, int tile_h,
                              int32_t *tiles_per_gauss, int64_t *cum_tiles, void *scratch, size_t scratch_bytes,
                              void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return PR_EINVAL;
  int64_t total = (int64_t)C * N;
  if (total == 0) return PR_OK;
  if (!means2d || !radii || !tiles_per_gauss) return PR_EINVAL;
  if ((conics == nullptr) != (opacities == nullptr)) return PR_EINVAL;
  if (cum_tiles && !scratch) return PR_EINVAL;
  unsigned grid = (unsigned)((total + 255) / 256);
  int rc;
  if (conics) {
    hipLaunchKernelGGL(isect_count_tight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, means2d,
                       radii, conics, opacities, (float)tile_size, tile_w, tile_h, tiles_per_gauss);
    rc = check_launch("isect_count_tight_kernel");
  } else {
    hipLaunchKernelGGL(isect_count_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, means2d, radii,
                       (float)tile_size, tile_w, tile_h, tiles_per_gauss);
    rc = check_launch("isect_count_kernel");
  }
  if (rc || !cum_tiles) return rc;
  size_t need = 0;
  auto in = rocprim::make_transform_iterator((const int32_t *)tiles_per_gauss, ToI64());
  PR_HIP(rocprim::inclusive_scan(nullptr, need, in, cum_tiles, (size_t)total, rocprim::plus<int64_t>(),
                                 (hipStream_t)stream));
  if (need > scratch_bytes) return PR_ESCRATCH;
  PR_HIP(rocprim::inclusive_scan(scratch, need, in, cum_tiles, (size_t)total, rocprim::plus<int64_t>(),
                                 (hipStream_t)stream));
  return PR_OK;
}

// ---- phase 2: two-level stable sort -----------------------------------------------------------------
// The reference sorts 64-bit keys (camera | tile | depth bits) over all intersections: ceil((32+tb+cb)/8)
// radix passes over I pairs.  The same order is reached with far less traffic by sorting the C*N
// GAUSSIANS by depth bits first (stable, so ties keep ascending flat index), emitting intersections in
// that order with 32-bit (camera | tile) keys, and finishing with a stable sort over only the tb+cb tile
// bits (2 passes of 8-byte pairs at 1080p instead of 6 passes of 12-byte pairs).
// rocPRIM's default switches from block-sort + merge passes to Onesweep above 1M items; measured on gfx950
// for (u32,u32) pairs the merge path wins at 200k items (60 vs 120 us) and loses from about 1M on
// (159 vs 138 us), so the switch point is moved to 512k.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 512 * 1024>;

struct Phase2Layout {
  size_t dkeys_in, dkeys_out, dvals_in, dvals_out, starts, tkeys_in, tkeys_out, tvals_in, tvals_tmp, temp, temp_bytes,
      total;
};

struct TilesOf {
  const int32_t *tpg;
  __device__ __host__ uint32_t operator()(uint32_t flat) const { return (uint32_t)tpg[flat]; }
};

static Phase2Layout phase2_layout(int64_t cn, int64_t n_isects) {
  Phase2Layout L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align256(bytes);
    return o;
  };
  L.dkeys_in = take((size_t)cn * 4);
  L.dkeys_out = take((size_t)cn * 4);
  L.dvals_in = take((size_t)cn * 4);
  L.dvals_out = take((size_t)cn * 4);
  L.starts = take((size_t)cn * 4);
  L.tkeys_in = take((size_t)n_isects * 4);
  L.tkeys_out = take((size_t)n_isects * 4);
  L.tvals_in = take((size_t)n_isects * 4);
  L.tvals_tmp = take((size_t)n_isects * 4);
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, t1, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)cn, 0u, 32u);
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, t2, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n_isects, 0u, 32u);
  size_t t4 = 0;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, t4, (const uint16_t *)nullptr, (uint16_t *)nullptr,
                                  (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n_isects, 0u, 16u);
  t2 = t2 > t4 ? t2 : t4;
  auto it = rocprim::make_transform_iterator((const uint32_t *)nullptr, TilesOf{nullptr});
  (void)rocprim::exclusive_scan(nullptr, t3, it, (uint32_t *)nullptr, 0u, (size_t)cn, rocprim::plus<uint32_t>());
  L.temp_bytes = t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
  // the chained-scan sort's counters: the depth keys' second value buffer (4 * cn bytes) sits in front of them
  L.temp_bytes = std::max(L.temp_bytes, align256((size_t)cn * 4) + std::max(rs::layout(cn, 4).total,
                                                                            rs::layout(n_isects, 4).total));
  L.temp = take(L.temp_bytes);
  L.total = off + 256;
  return L;
}

__global__ void __launch_bounds__(256) depth_keys_kernel(int64_t total, const float *__restrict__ depths,
                                                         const int32_t *__restrict__ tiles_per_gauss,
                                                         uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  // Gaussians without tiles emit nothing wherever they land; pushing them to the end when the counts are
  // known just keeps the emitting waves dense
  keys[idx] = (!tiles_per_gauss || tiles_per_gauss[idx] > 0) ? (uint32_t)__float_as_int(depths[idx]) : 0xFFFFFFFFu;
  vals[idx] = (uint32_t)idx;
}

// The same keys as a functor: the chained-scan sort reads them where they are instead of from a key array.
struct DepthBits {
  const float *depths;
  const int32_t *tiles_per_gauss;
  __device__ __forceinline__ uint32_t operator()(int64_t i) const {
    return (!tiles_per_gauss || tiles_per_gauss[i] > 0) ? (uint32_t)__float_as_int(depths[i]) : 0xFFFFFFFFu;
  }
};

// Which device sort orders the depth keys and the (camera | tile) keys: the chained-scan sort of onesweep.h unless
// POGS_RAST_DEPTH_SORT=1 asks for rocPRIM's (the cross-check) or the item count exceeds its 30-bit counters.
static inline bool own_radix_sort(int64_t items) { return tuning().depth_sort != 1 && items <= rs::MAX_ITEMS; }

// Number of valid intersections: the buffers hold `cap`; when the caller sized them before the true count
// reached the host, the count is read from device memory (and clamped, so an undersized guess stays
// memory-safe -- the host detects it and repeats the call).
__device__ __forceinline__ int64_t true_count(int64_t cap, const int64_t *__restrict__ n_dev) {
  if (!n_dev) return cap;
  const int64_t n = *n_dev;
  return n < cap ? n : cap;
}

// Emits the (camera|tile) key and the flat Gaussian index of every intersection, Gaussians in depth order.
// Load-balanced expansion per wavefront: the 64 Gaussians of a wave own a contiguous output range
// [start of lane 0, end of lane 63); the wave walks that range 64 outputs at a time, every lane finding the
// Gaussian its output belongs to by a 6-step search over the lanes' start offsets -- so the 4-byte writes
// are coalesced however uneven the tile counts are.
template <class KeyT>
__global__ void __launch_bounds__(256) isect_emit_sorted_kernel(
    int64_t total, int N, const uint32_t *__restrict__ order, const uint32_t *__restrict__ starts,
    const float *__restrict__ means2d, const int32_t *__restrict__ radii, float ts, int tile_w, int tile_h,
    int tile_bits, int64_t cap, const int64_t *__restrict__ n_dev, KeyT *__restrict__ keys,
    uint32_t *__restrict__ vals) {
  const int64_t n_isects = true_count(cap, n_dev);
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t idx = 0, start = (uint32_t)n_isects;
  int x0 = 0, y0 = 0, w = 1, cnt = 0;
  if (r < total) {
    idx = order[r];
    start = starts[r];
    const int rad = radii[idx];
    if (rad > 0) {
      TileRect t = tile_rect(means2d[2 * (int64_t)idx], means2d[2 * (int64_t)idx + 1], rad, ts, tile_w, tile_h);
      const int c = (t.x1 - t.x0) * (t.y1 - t.y0);
      if (c > 0) {
        x0 = t.x0;
        y0 = t.y0;
        w = t.x1 - t.x0;
        cnt = c;
      }
    }
  }
  const uint32_t hi = (idx / (uint32_t)N) << tile_bits;
  const uint32_t begin = (uint32_t)__builtin_amdgcn_readlane((int)start, 0);
  uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)(start + (uint32_t)cnt), 63);
  if ((int64_t)end > n_isects) end = (uint32_t)n_isects;
  for (uint32_t k0 = begin; k0 < end; k0 += 64) {
    const uint32_t k = k0 + (uint32_t)lane;
    int j = 0;  // last lane whose start is <= k
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const uint32_t sc = (uint32_t)__shfl((int)start, (j + step) & 63);
      if (j + step < 64 && sc <= k) j += step;
    }
    const uint32_t sj = (uint32_t)__shfl((int)start, j);
    const int xj = __shfl(x0, j), yj = __shfl(y0, j), wj = __shfl(w, j);
    const uint32_t hj = (uint32_t)__shfl((int)hi, j), ij = (uint32_t)__shfl((int)idx, j);
    if (k < end) {
      const int t = (int)(k - sj);
      const int ty = t / wj, tx = t - ty * wj;
      keys[k] = (KeyT)(hj | (uint32_t)((yj + ty) * tile_w + (xj + tx)));
      vals[k] = ij;
    }
  }
}

// Start of every tile's range in the sorted list, from the sorted (camera|tile) keys.
constexpr int TILE_OFFSETS_ITEMS = 8;   // keys per thread, requested together
template <class KeyT>
__global__ void __launch_bounds__(256) tile_offsets_kernel(int64_t cap, const int64_t *__restrict__ n_dev,
                                                           const KeyT *__restrict__ tkeys,
                                                           int n_tiles_total, int tile_bits, int n_tiles,
                                                           int32_t *__restrict__ offsets) {
  const int64_t n_isects = true_count(cap, n_dev);
  const uint32_t mask = (1u << tile_bits) - 1u;
  const int64_t k0 = (int64_t)blockIdx.x * (256 * TILE_OFFSETS_ITEMS) + threadIdx.x;
  uint32_t hc[TILE_OFFSETS_ITEMS], hp[TILE_OFFSETS_ITEMS];
#pragma unroll
  for (int i = 0; i < TILE_OFFSETS_ITEMS; ++i) {
    const int64_t k = k0 + i * 256;
    hc[i] = k < n_isects ? (uint32_t)tkeys[k] : 0u;
    hp[i] = k < n_isects && k > 0 ? (uint32_t)tkeys[k - 1] : 0u;
  }
#pragma unroll
  for (int i = 0; i < TILE_OFFSETS_ITEMS; ++i) {
    const int64_t k = k0 + i * 256;
    if (k >= n_isects) break;
    const int cur = (int)(hc[i] >> tile_bits) * n_tiles + (int)(hc[i] & mask);
    if (k == 0) {
      for (int t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
      const int prev = (int)(hp[i] >> tile_bits) * n_tiles + (int)(hp[i] & mask);
      for (int t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)k;
    }
    if (k == n_isects - 1)
      for (int t = cur + 1; t < n_tiles_total; ++t) offsets[t] = (int32_t)n_isects;
  }
}

// Launch order of the compositing kernels: tiles by list length, longest first.  Only the rough order
// matters (it shortens the tail of the launch), so one workgroup buckets the tiles by length -- 1024 linear
// buckets between 0 and the longest list -- instead of a full device sort (six small kernels for ~8k keys).
namespace pr {
__device__ void tile_order_body(int n_tiles_total, const int32_t *__restrict__ offsets, int64_t n_isects,
                                int32_t *__restrict__ order) {
  constexpr int NB = 1024;
  __shared__ int bucket[NB];
  __shared__ int wave_part[16];
  __shared__ int longest;
  const int tid = threadIdx.x;
  auto length = [&](int t) {
    const int32_t hi = t + 1 < n_tiles_total ? offsets[t + 1] : (int32_t)n_isects;
    return hi - offsets[t];
  };
  // every pass over the tiles requests eight lengths per thread before it uses any: a pass costs one memory round
  // trip per 8192 tiles instead of one per tile and thread
  auto for_each_tile = [&](auto &&use) {
    for (int t0 = tid; t0 < n_tiles_total; t0 += 8 * NB) {
      int len[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) len[i] = t0 + i * NB < n_tiles_total ? length(t0 + i * NB) : 0;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (t0 + i * NB < n_tiles_total) use(t0 + i * NB, len[i]);
    }
  };
  bucket[tid] = 0;
  if (tid == 0) longest = 0;
  __syncthreads();
  int mx = 0;
  for_each_tile([&](int, int len) { mx = max(mx, len); });
  atomicMax(&longest, mx);
  __syncthreads();
  const float scale = longest > 0 ? (float)(NB - 1) / (float)longest : 0.f;
  for_each_tile([&](int, int len) { atomicAdd(&bucket[(int)((float)(longest - len) * scale)], 1); });
  __syncthreads();
  // exclusive scan of the 1024 bucket sizes: one value per thread
  const int lane = tid & 63, wave = tid >> 6;
  const int mine = bucket[tid];
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_part[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wave_part[w];
  __syncthreads();
  bucket[tid] = base + incl - mine;  // from here on: the bucket's write cursor
  __syncthreads();
  for_each_tile([&](int t, int len) { order[atomicAdd(&bucket[(int)((float)(longest - len) * scale)], 1)] = t; });
}
}  // namespace pr

__global__ void __launch_bounds__(1024) tile_order_kernel(int n_tiles_total, const int32_t *__restrict__ offsets,
                                                          int64_t cap, const int64_t *__restrict__ n_dev,
                                                          int32_t *__restrict__ order) {
  tile_order_body(n_tiles_total, offsets, true_count(cap, n_dev), order);
}

template <class KeyT>
__global__ void __launch_bounds__(256) isect_ids_kernel(int64_t cap, const int64_t *__restrict__ n_dev,
                                                        const KeyT *__restrict__ tkeys,
                                                        const int32_t *__restrict__ flatten_ids,
                                                        const float *__restrict__ depths,
                                                        int64_t *__restrict__ isect_ids) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= true_count(cap, n_dev)) return;
  const uint32_t d = (uint32_t)__float_as_int(depths[flatten_ids[k]]);
  isect_ids[k] = ((int64_t)tkeys[k] << 32) | (int64_t)d;
}

// The intersection count for the host without a copy command: a one-thread kernel stores it into page-locked host
// memory, then a sequence number next to it; the host polls the sequence word.  A copy plus an event costs two command
// barriers on the stream (about 10 us of idle GPU each on this hardware); this costs one 2 us launch.
__global__ void publish_count_kernel(const int64_t *__restrict__ count_dev, int64_t *__restrict__ host_words,
                                     int64_t sequence) {
  __hip_atomic_store(host_words, *count_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_words + 1, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int pr_publish_count(const int64_t *count_dev, int64_t *host_words, int64_t sequence, void *stream) {
  if (!count_dev || !host_words) return PR_EINVAL;
  hipLaunchKernelGGL(publish_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, count_dev, host_words, sequence);
  return check_launch("publish_count_kernel");
}

extern "C" size_t pr_isect_sort_scratch_bytes(int C, int N, int64_t n_isects, int64_t n_tiles_total) {
  if (n_isects <= 0 || (int64_t)C * N <= 0) return 256;
  (void)n_tiles_total;
  return phase2_layout((int64_t)C * N, n_isects).total;
}

// Depth order of all C*N Gaussians on its own: depends on the projection's depths only, so a caller can run
// it on a second stream next to the colour evaluation and the tile counting.
extern "C" size_t pr_depth_order_scratch_bytes(int C, int N) {
  const int64_t cn = (int64_t)C * N;
  if (cn <= 0) return 256;
  size_t t = 0;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, t, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                              (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)cn, 0u, 32u);
  const size_t radix = align256((size_t)cn * 4) * 3 + align256(t) + 256;
  const size_t chained = align256((size_t)cn * 4) * 4 + rs::layout(cn, 4).total + 256;
  return std::max(std::max(radix, chained), sample_sort_layout(cn).total);
}

static int depth_order_sample_sort(int64_t total, const float *depths, int32_t *depth_order, char *base,
                                   hipStream_t st) {
  const SampleSortLayout L = sample_sort_layout(total);
  unsigned long long *splitters = (unsigned long long *)(base + L.splitters);
  uint16_t *bucket_ids = (uint16_t *)(base + L.bucket_ids);
  uint32_t *counts = (uint32_t *)(base + L.counts), *bucket_start = (uint32_t *)(base + L.bucket_start);
  unsigned long long *pairs = (unsigned long long *)(base + L.pairs);
  const size_t sample_lds = (size_t)SS_SAMPLES * 8;
  PR_HIP(hipFuncSetAttribute((const void *)ss_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sample_lds));
  hipLaunchKernelGGL(ss_sample_kernel, dim3(1), dim3(SS_THREADS), sample_lds, st, total, depths, splitters);
  hipLaunchKernelGGL(ss_count_kernel, dim3(L.n_blocks), dim3(SS_THREADS), 0, st, total, L.per_block, depths, splitters,
                     bucket_ids, counts);
  hipLaunchKernelGGL(ss_scan_kernel, dim3(1), dim3(SS_THREADS), 0, st, L.n_blocks, counts, bucket_start);
  hipLaunchKernelGGL(ss_scatter_kernel, dim3(L.n_blocks), dim3(SS_THREADS), 0, st, total, L.per_block, depths, bucket_ids,
                     counts, bucket_start, pairs);
  hipLaunchKernelGGL(ss_bucket_sort_kernel, dim3(SS_BUCKETS), dim3(256), 0, st, bucket_start, pairs, depth_order);
  return check_launch("depth_order_sample_sort");
}

extern "C" int pr_depth_order(int C, int N, const float *depths, int32_t *depth_order, void *scratch,
                              size_t scratch_bytes, void *stream) {
  if (C < 0 || N < 0) return PR_EINVAL;
  const int64_t total = (int64_t)C * N;
  if (total == 0) return PR_OK;
  if (total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  if (!depths || !depth_order || !scratch) return PR_EINVAL;
  if (scratch_bytes < pr_depth_order_scratch_bytes(C, N)) return PR_ESCRATCH;
  char *base = (char *)scratch;
  if (total >= SS_MIN_TOTAL && total <= SS_MAX_TOTAL && tuning().depth_sort == 2)
    return depth_order_sample_sort(total, depths, depth_order, base, (hipStream_t)stream);
  const size_t seg = align256((size_t)total * 4);
  if (own_radix_sort(total)) {
    uint32_t *const kbuf[2] = {(uint32_t *)base, (uint32_t *)(base + seg)};
    uint32_t *const vbuf[2] = {(uint32_t *)(base + 2 * seg), (uint32_t *)(base + 3 * seg)};
    return rs::sort_pairs<uint32_t>(DepthBits{depths, nullptr}, rs::Iota{}, total, nullptr, 32, kbuf, vbuf,
                                    (uint32_t *)nullptr, (uint32_t *)depth_order, base + 4 * seg, (hipStream_t)stream);
  }
  uint32_t *k_in = (uint32_t *)base, *k_out = (uint32_t *)(base + seg), *v_in = (uint32_t *)(base + 2 * seg);
  void *temp = base + 3 * seg;
  size_t tb = scratch_bytes - 3 * seg;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(depth_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, depths,
                     (const int32_t *)nullptr, k_in, v_in);
  int rc = check_launch("depth_keys_kernel");
  if (rc) return rc;
  PR_HIP(rocprim::radix_sort_pairs<SortConfig>(temp, tb, (const uint32_t *)k_in, k_out, (const uint32_t *)v_in,
                                               (uint32_t *)depth_order, (size_t)total, 0u, 32u, st));
  return PR_OK;
}

// ---- the same depth order for callers that cannot allocate ---------------------------------------------------
// pr_depth_order's kernels work in a caller-supplied scratch buffer (sort keys in and out, the index sequence, the
// radix sort's own temporary storage).  This variant keeps that buffer in a file-scope device array instead, so a
// caller under HIP graph capture, in a tight serving loop or in a harness that forbids allocation hands over
// nothing but its input and its output.  Each GPU the library runs on holds the array until the process ends.
namespace {
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)24 << 20;   // enough for C*N of about 1.25 million on MI355X (see the size check)
constexpr int MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
// In host code `scratch` is only a handle for the symbol; its device address differs per GPU and is looked up
// once per device (with that device current) and used from then on wherever the memory is meant.
static void *g_static_scratch[MAX_DEVICES];
static std::mutex g_static_scratch_lock;

static int static_scratch_ptr(void **out) {
  int dev = -1;
  PR_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return PR_EUNSUPPORTED;
  std::lock_guard<std::mutex> hold(g_static_scratch_lock);
  if (!g_static_scratch[dev]) {
    void *p = nullptr;
    PR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return PR_EHIP;
    g_static_scratch[dev] = p;
  }
  *out = g_static_scratch[dev];
  return PR_OK;
}

extern "C" int64_t pr_depth_order_static_max(void) {
  // the largest C*N whose scratch (three 4-byte arrays + the sort's temporary storage) fits the static array
  int64_t lo = 0, hi = (int64_t)1 << 24;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) / 2;
    if (pr_depth_order_scratch_bytes(1, (int)mid) <= STATIC_SCRATCH_BYTES) lo = mid; else hi = mid - 1;
  }
  return lo;
}

extern "C" int pr_depth_order_static(int C, int N, const float *depths, int32_t *depth_order, void *stream) {
  if (C < 0 || N < 0) return PR_EINVAL;
  const int64_t total = (int64_t)C * N;
  if (total == 0) return PR_OK;
  if (!depths || !depth_order) return PR_EINVAL;
  if (total > (int64_t)2147483647 || pr_depth_order_scratch_bytes(C, N) > STATIC_SCRATCH_BYTES) {
    set_error_text("pr_depth_order_static: C*N exceeds the static scratch (pr_depth_order_static_max()); use "
                   "pr_depth_order with a caller-supplied buffer");
    return PR_EUNSUPPORTED;
  }
  void *ptr = nullptr;
  const int rc = static_scratch_ptr(&ptr);
  if (rc) return rc;
  // every byte the kernels read from the buffer is written earlier in the same call (keys and indices by
  // depth_keys_kernel, the sort's counters by its own initialisation), so nothing needs clearing between calls
  return pr_depth_order(C, N, depths, depth_order, ptr, STATIC_SCRATCH_BYTES, stream);
}

template <class KeyT>
static int emit_and_tile_sort(int C, int N, int64_t total, int64_t n_isects, const int64_t *n_isects_dev,
                              int64_t n_tiles_total, int tile_bits, int cam_bits, const uint32_t *dv_out,
                              const uint32_t *starts, const float *means2d, const int32_t *radii, const float *conics,
                              const float *opacities, const int32_t *tiles_per_gauss, const float *depths, int tile_size,
                              int tile_w, int tile_h, KeyT *tk_in, KeyT *tk_out, uint32_t *tv_in, uint32_t *tv_tmp, int32_t *flatten_ids,
                              int32_t *isect_offsets, int32_t *tile_order, int64_t *isect_ids, void *temp,
                              size_t temp_bytes, hipStream_t st) {
  const unsigned grid_n = (unsigned)((total + 255) / 256);
  const bool own = own_radix_sort(n_isects);
  if (n_isects_dev) {
    // capacity mode: slots past the true count keep an all-ones key and sort to the end (the chained-scan sort reads
    // the true count itself and never looks at them)
    if (!own) PR_HIP(hipMemsetAsync(tk_in, 0xFF, (size_t)n_isects * sizeof(KeyT), st));
    if (isect_offsets) PR_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)n_tiles_total * 4, st));
  }
  if (conics)
    hipLaunchKernelGGL(isect_emit_tight_kernel<KeyT>, dim3(grid_n), dim3(256), 0, st, total, N, dv_out, starts, means2d,
                       radii, conics, opacities, tiles_per_gauss, (float)tile_size, tile_w, tile_h, tile_bits,
                       n_isects, n_isects_dev, tk_in, tv_in);
  else
    hipLaunchKernelGGL(isect_emit_sorted_kernel<KeyT>, dim3(grid_n), dim3(256), 0, st, total, N, dv_out, starts, means2d,
                       radii, (float)tile_size, tile_w, tile_h, tile_bits, n_isects, n_isects_dev, tk_in, tv_in);
  int rc = check_launch("isect_emit_kernel");
  if (rc) return rc;
  if (own) {
    const int bits = tile_bits + cam_bits, passes = bits <= 0 ? 1 : (bits + 7) / 8;
    KeyT *const kbuf[2] = {tk_out, tk_in};
    uint32_t *const vbuf[2] = {tv_tmp, tv_in};
    KeyT *sorted = kbuf[(passes - 1) & 1];
    rc = rs::sort_pairs<KeyT>(rs::FromArray<KeyT>{tk_in}, rs::FromArray<uint32_t>{tv_in}, n_isects, n_isects_dev, bits, kbuf,
                              vbuf, sorted, (uint32_t *)flatten_ids, (char *)temp, st);
    if (rc) return rc;
    tk_out = sorted;
  } else {
    size_t tb = temp_bytes;
    PR_HIP(rocprim::radix_sort_pairs<SortConfig>(temp, tb, (const KeyT *)tk_in, tk_out, (const uint32_t *)tv_in,
                                                 (uint32_t *)flatten_ids, (size_t)n_isects, 0u,
                                                 (unsigned)(tile_bits + cam_bits), st));
  }
  const unsigned grid_i = (unsigned)((n_isects + 255) / 256);
  if (isect_offsets) {
    const unsigned grid_o = (unsigned)((n_isects + 256 * TILE_OFFSETS_ITEMS - 1) / (256 * TILE_OFFSETS_ITEMS));
    hipLaunchKernelGGL(tile_offsets_kernel<KeyT>, dim3(grid_o), dim3(256), 0, st, n_isects, n_isects_dev,
                       (const KeyT *)tk_out, (int)n_tiles_total, tile_bits, tile_w * tile_h, isect_offsets);
    rc = check_launch("tile_offsets_kernel");
    if (rc) return rc;
  }
  if (tile_order) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, (int)n_tiles_total, isect_offsets, n_isects,
                       n_isects_dev, tile_order);
    rc = check_launch("tile_order_kernel");
    if (rc) return rc;
  }
  if (isect_ids) {
    hipLaunchKernelGGL(isect_ids_kernel<KeyT>, dim3(grid_i), dim3(256), 0, st, n_isects, n_isects_dev,
                       (const KeyT *)tk_out, flatten_ids, depths, isect_ids);
    rc = check_launch("isect_ids_kernel");
  }
  return rc;
}

extern "C" int pr_isect_emit_sort(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                                  const float *conics, const float *opacities, const int32_t *tiles_per_gauss, int tile_size, int tile_w, int tile_h,
                                  int64_t n_isects, int64_t *isect_ids, int32_t *flatten_ids,
                                  int32_t *isect_offsets, int32_t *tile_order, const int64_t *n_isects_dev,
                                  const int32_t *depth_order, void *scratch, size_t scratch_bytes, void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || n_isects < 0) return PR_EINVAL;
  const int64_t total = (int64_t)C * N;
  if (tile_order && !isect_offsets) return PR_EINVAL;  // the order is derived from the offsets table
  if ((conics == nullptr) != (opacities == nullptr)) return PR_EINVAL;
  const int64_t n_tiles_total = (int64_t)C * tile_w * tile_h;
  if (n_tiles_total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  if (total == 0 || n_isects == 0) {
    if (isect_offsets && n_tiles_total > 0) {
      PR_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)n_tiles_total * 4, (hipStream_t)stream));
      if (tile_order) {
        hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)n_tiles_total,
                           isect_offsets, (int64_t)0, (const int64_t *)nullptr, tile_order);
        return check_launch("tile_order_kernel");
      }
    }
    return PR_OK;
  }
  if (n_isects > (int64_t)2147483647 || total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  if (!means2d || !radii || !depths || !tiles_per_gauss || !flatten_ids || !scratch) return PR_EINVAL;
  const int tile_bits = bit_length((int64_t)tile_w * tile_h);
  const int cam_bits = bit_length((int64_t)C - 1);
  if (tile_bits + cam_bits > 32) return PR_EUNSUPPORTED;
  const Phase2Layout L = phase2_layout(total, n_isects);
  if (L.total > scratch_bytes) return PR_ESCRATCH;
  char *base = (char *)scratch;
  uint32_t *dk_in = (uint32_t *)(base + L.dkeys_in), *dk_out = (uint32_t *)(base + L.dkeys_out);
  uint32_t *dv_in = (uint32_t *)(base + L.dvals_in), *dv_out = (uint32_t *)(base + L.dvals_out);
  uint32_t *starts = (uint32_t *)(base + L.starts);
  uint32_t *tk_in = (uint32_t *)(base + L.tkeys_in), *tk_out = (uint32_t *)(base + L.tkeys_out);
  uint32_t *tv_in = (uint32_t *)(base + L.tvals_in), *tv_tmp = (uint32_t *)(base + L.tvals_tmp);
  void *temp = base + L.temp;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid_n = (unsigned)((total + 255) / 256);
  int rc = PR_OK;
  size_t tb = L.temp_bytes;
  if (depth_order) {
    dv_out = (uint32_t *)depth_order;  // computed by pr_depth_order
  } else if (own_radix_sort(total)) {
    uint32_t *const kbuf[2] = {dk_in, dk_out};
    uint32_t *const vbuf[2] = {dv_in, (uint32_t *)temp};
    rc = rs::sort_pairs<uint32_t>(DepthBits{depths, tiles_per_gauss}, rs::Iota{}, total, nullptr, 32, kbuf, vbuf,
                                  (uint32_t *)nullptr, dv_out, (char *)temp + align256((size_t)total * 4), st);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(depth_keys_kernel, dim3(grid_n), dim3(256), 0, st, total, depths, tiles_per_gauss, dk_in,
                       dv_in);
    rc = check_launch("depth_keys_kernel");
    if (rc) return rc;
    PR_HIP(rocprim::radix_sort_pairs<SortConfig>(temp, tb, (const uint32_t *)dk_in, dk_out, (const uint32_t *)dv_in,
                                                 dv_out, (size_t)total, 0u, 32u, st));
  }
  auto it = rocprim::make_transform_iterator((const uint32_t *)dv_out, TilesOf{tiles_per_gauss});
  tb = L.temp_bytes;
  PR_HIP(rocprim::exclusive_scan(temp, tb, it, starts, 0u, (size_t)total, rocprim::plus<uint32_t>(), st));
  // (camera | tile) keys of at most 16 bits travel as 16-bit words: the tile sort then moves 6 instead of 8 bytes per pair
  // and pass.  0xFFFF must stay free for the padding of capacity mode.
  if (tile_bits + cam_bits <= 16 && n_tiles_total < 65535)
    return emit_and_tile_sort<uint16_t>(C, N, total, n_isects, n_isects_dev, n_tiles_total, tile_bits, cam_bits, dv_out,
                                        starts, means2d, radii, conics, opacities, tiles_per_gauss, depths, tile_size, tile_w,
                                        tile_h, (uint16_t *)tk_in, (uint16_t *)tk_out, tv_in, tv_tmp, flatten_ids, isect_offsets,
                                        tile_order, isect_ids, temp, L.temp_bytes, st);
  return emit_and_tile_sort<uint32_t>(C, N, total, n_isects, n_isects_dev, n_tiles_total, tile_bits, cam_bits, dv_out,
                                      starts, means2d, radii, conics, opacities, tiles_per_gauss, depths, tile_size, tile_w,
                                      tile_h, tk_in, tk_out, tv_in, tv_tmp, flatten_ids, isect_offsets, tile_order, isect_ids, temp,
                                      L.temp_bytes, st);
}

extern "C" int pr_tile_lists_count(int C, int N, const float *means2d, const int32_t *radii, const float *conics,
                                   const float *opacities, int tile_size, int tile_w, int tile_h,
                                   int32_t *tiles_per_gauss, int32_t *tile_counters, int32_t *isect_offsets,
                                   int32_t *tile_order, int64_t *n_isects_dev, void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return PR_EINVAL;
  const int64_t total = (int64_t)C * N, n_tiles_total = (int64_t)C * tile_w * tile_h;
  if (total > (int64_t)2147483647 || n_tiles_total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  if (n_tiles_total == 0) return PR_OK;
  if (!tile_counters || !isect_offsets || !n_isects_dev) return PR_EINVAL;
  if (total > 0 && (!means2d || !radii || !conics || !opacities)) return PR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PR_HIP(hipMemsetAsync(tile_counters, 0, (size_t)n_tiles_total * 4, st));
  if (total > 0) {
    hipLaunchKernelGGL(tile_lists_count_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, N,
                       tile_w * tile_h, means2d, radii, conics, opacities, (float)tile_size, tile_w, tile_h,
                       tiles_per_gauss, tile_counters);
    const int rc = check_launch("tile_lists_count_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(tile_lists_scan_kernel, dim3(1), dim3(1024), 0, st, (int)n_tiles_total, tile_counters,
                     isect_offsets, n_isects_dev, tile_order);
  return check_launch("tile_lists_scan_kernel");
}

extern "C" int pr_tile_lists_fill(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                                  const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                                  const int32_t *isect_offsets, int32_t *tile_counters, int64_t capacity,
                                  const int64_t *n_isects_dev, uint64_t *keys_scratch, int32_t *flatten_ids,
                                  void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || capacity < 0) return PR_EINVAL;
  const int64_t total = (int64_t)C * N, n_tiles_total = (int64_t)C * tile_w * tile_h;
  if (total > (int64_t)2147483647 || n_tiles_total > (int64_t)2147483647 || capacity > (int64_t)2147483647)
    return PR_EUNSUPPORTED;
  if (total == 0 || capacity == 0 || n_tiles_total == 0) return PR_OK;
  if (!means2d || !radii || !depths || !conics || !opacities || !isect_offsets || !tile_counters || !n_isects_dev ||
      !keys_scratch || !flatten_ids)
    return PR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tile_lists_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, N,
                     tile_w * tile_h, means2d, radii, depths, conics, opacities, (float)tile_size, tile_w, tile_h,
                     isect_offsets, tile_counters, capacity, (unsigned long long *)keys_scratch);
  int rc = check_launch("tile_lists_emit_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(tile_lists_sort_kernel, dim3((unsigned)n_tiles_total), dim3(256), 0, st, (int)n_tiles_total,
                     isect_offsets, capacity, n_isects_dev, (unsigned long long *)keys_scratch, flatten_ids);
  return check_launch("tile_lists_sort_kernel");
}

extern "C" int pr_isect_offsets(int64_t n_isects, const int64_t *isect_ids, int C, int tile_w, int tile_h,
                                int32_t *isect_offsets, void *stream) {
  if (C < 0 || tile_w <= 0 || tile_h <= 0 || n_isects < 0) return PR_EINVAL;
  int64_t n_tiles_total = (int64_t)C * tile_w * tile_h;
  if (n_tiles_total == 0) return PR_OK;
  if (!isect_offsets) return PR_EINVAL;
  if (n_isects == 0) {
    PR_HIP(hipMemsetAsync(isect_offsets, 0, (size_t)n_tiles_total * 4, (hipStream_t)stream));
    return PR_OK;
  }
  if (!isect_ids) return PR_EINVAL;
  int tile_bits = bit_length((int64_t)tile_w * tile_h);
  unsigned grid = (unsigned)((n_isects + 255) / 256);
  hipLaunchKernelGGL(isect_offsets_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_isects, isect_ids,
                     (int)n_tiles_total, tile_bits, tile_w * tile_h, isect_offsets);
  return check_launch("isect_offsets_kernel");
}
