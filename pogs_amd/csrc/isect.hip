// Tile-intersection binning for gfx950: per-Gaussian tile counts + prefix sum, key/value emission,
// stable device radix sort (rocPRIM, restricted to the significant key bits), per-tile offsets.
//
// Integer work; results must be bit-identical to the CPU restatement, so the tile rectangle is
// computed with true float32 divisions (no reciprocal multiply that could be contracted into an FMA).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

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

__global__ void __launch_bounds__(256) isect_emit_kernel(int64_t total, int N, const float *__restrict__ means2d,
                                                         const int32_t *__restrict__ radii,
                                                         const float *__restrict__ depths,
                                                         const int64_t *__restrict__ cum_tiles, float ts, int tile_w,
                                                         int tile_h, int tile_bits, int64_t n_isects,
                                                         int64_t *__restrict__ keys, int32_t *__restrict__ vals) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int r = radii[idx];
  if (r <= 0) return;
  TileRect t = tile_rect(means2d[2 * idx], means2d[2 * idx + 1], r, ts, tile_w, tile_h);
  int cnt = (t.x1 - t.x0) * (t.y1 - t.y0);
  if (cnt <= 0) return;
  int64_t k = cum_tiles[idx] - cnt;
  int64_t cam = idx / N;
  int64_t dbits = (int64_t)(uint32_t)__float_as_int(depths[idx]);
  int64_t hi = cam << tile_bits;
  for (int y = t.y0; y < t.y1; ++y)
    for (int x = t.x0; x < t.x1; ++x) {
      if (k < n_isects) {
        keys[k] = ((hi | (int64_t)(y * tile_w + x)) << 32) | dbits;
        vals[k] = (int32_t)idx;
      }
      ++k;
    }
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

struct ToI64 {
  __device__ __host__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

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

extern "C" int pr_isect_count(int C, int N, const float *means2d, const int32_t *radii, int tile_size, int tile_w,
                              int tile_h, int32_t *tiles_per_gauss, int64_t *cum_tiles, void *scratch,
                              size_t scratch_bytes, void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return PR_EINVAL;
  int64_t total = (int64_t)C * N;
  if (total == 0) return PR_OK;
  if (!means2d || !radii || !tiles_per_gauss || !cum_tiles || !scratch) return PR_EINVAL;
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(isect_count_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, means2d, radii,
                     (float)tile_size, tile_w, tile_h, tiles_per_gauss);
  int rc = check_launch("isect_count_kernel");
  if (rc) return rc;
  size_t need = 0;
  auto in = rocprim::make_transform_iterator((const int32_t *)tiles_per_gauss, ToI64());
  PR_HIP(rocprim::inclusive_scan(nullptr, need, in, cum_tiles, (size_t)total, rocprim::plus<int64_t>(),
                                 (hipStream_t)stream));
  if (need > scratch_bytes) return PR_ESCRATCH;
  PR_HIP(rocprim::inclusive_scan(scratch, need, in, cum_tiles, (size_t)total, rocprim::plus<int64_t>(),
                                 (hipStream_t)stream));
  return PR_OK;
}

static size_t sort_temp_bytes(int64_t n_isects) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int64_t *)nullptr, (int64_t *)nullptr, (const int32_t *)nullptr,
                            (int32_t *)nullptr, (size_t)n_isects, 0u, 64u);
  return bytes;
}

extern "C" size_t pr_isect_sort_scratch_bytes(int C, int N, int64_t n_isects) {
  (void)C;
  (void)N;
  if (n_isects <= 0) return 256;
  return align256((size_t)n_isects * 8) + align256((size_t)n_isects * 4) + align256(sort_temp_bytes(n_isects)) + 256;
}

extern "C" int pr_isect_emit_sort(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                                  const int64_t *cum_tiles, int tile_size, int tile_w, int tile_h, int64_t n_isects,
                                  int64_t *isect_ids, int32_t *flatten_ids, void *scratch, size_t scratch_bytes,
                                  void *stream) {
  if (C < 0 || N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || n_isects < 0) return PR_EINVAL;
  int64_t total = (int64_t)C * N;
  if (total == 0 || n_isects == 0) return PR_OK;
  if (n_isects > (int64_t)2147483647) return PR_EUNSUPPORTED;  // offsets and last_ids are int32
  if (!means2d || !radii || !depths || !cum_tiles || !isect_ids || !flatten_ids || !scratch) return PR_EINVAL;
  size_t kb = align256((size_t)n_isects * 8), vb = align256((size_t)n_isects * 4);
  size_t tb = sort_temp_bytes(n_isects);
  if (kb + vb + tb > scratch_bytes) return PR_ESCRATCH;
  int64_t *keys = (int64_t *)scratch;
  int32_t *vals = (int32_t *)((char *)scratch + kb);
  void *temp = (char *)scratch + kb + vb;
  int tile_bits = bit_length((int64_t)tile_w * tile_h);
  int cam_bits = bit_length((int64_t)C - 1);
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(isect_emit_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, N, means2d, radii,
                     depths, cum_tiles, (float)tile_size, tile_w, tile_h, tile_bits, n_isects, keys, vals);
  int rc = check_launch("isect_emit_kernel");
  if (rc) return rc;
  unsigned end_bit = (unsigned)(32 + tile_bits + cam_bits);
  if (end_bit > 64u) return PR_EUNSUPPORTED;
  PR_HIP(rocprim::radix_sort_pairs(temp, tb, (const int64_t *)keys, isect_ids, (const int32_t *)vals, flatten_ids,
                                   (size_t)n_isects, 0u, end_bit, (hipStream_t)stream));
  return PR_OK;
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
