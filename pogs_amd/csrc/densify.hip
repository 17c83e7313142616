// Densification (grow / prune) of the Gaussian set after the backward pass: gsplat 1.4.0's DefaultStrategy
// (third-party gsplat/strategy/default.py and ops.py) as the reference configures and drives it
// (nerfstudio/models/splatfacto.py:264-281 construction, :365-375 step_post_backward; pogs/pogs.py:373-379
// step_pre_backward), restated as four streaming passes:
//
//   pr_densify_accumulate   every training step: per Gaussian, over the cameras that saw it, add the norm of the
//                           screen-normalised 2D-mean gradient and a visit count, track the largest normalised
//                           radius.  One kernel instead of clone / scale / mask / norm / where / 2 x index_add /
//                           gather-max-scatter.
//   pr_densify_plan         every refine step: classify every Gaussian (duplicate, split, prune -- the prune
//                           test also for the children a split would create) and scan the classes.
//   pr_densify_emit         turn the scan into ONE gather map in the published order: survivors that are not split,
//                           copies made by duplication, first children, second children -- each filtered by the
//                           prune test, i.e. exactly what duplicate -> split -> remove leave behind.
//   pr_densify_gather /     build every parameter and optimiser-state tensor of the new set with one gather each
//   pr_densify_split_children   (the published sequence materialises each of them three times).
//
// All of it is HBM-bound integer / gather work: coalesced streaming reads, row gathers with consecutive lanes on
// consecutive floats of a row.
#include <math.h>

#include "common.h"

namespace pr {
namespace {

constexpr int BLOCK = 256;
constexpr int ITEMS = 8;                   // consecutive Gaussians per thread in the scan passes
constexpr int PER_BLOCK = BLOCK * ITEMS;   // 2048
constexpr int MID_THREADS = 1024;

enum : unsigned char { F_DUP = 1, F_SPLIT = 2, F_PRUNE_SELF = 4, F_PRUNE_CHILD = 8 };
enum : int { KIND_ORIGINAL = 0, KIND_COPY = 1, KIND_CHILD0 = 2, KIND_CHILD1 = 3 };

// counters carried by the scan: originals kept, copies kept, children kept (per sample), splits (rank of a split
// source among all split sources, pruned or not: it indexes the noise samples), duplications
struct Cnt {
  int a, b, c, rank, dup;
};
__device__ __host__ __forceinline__ Cnt operator+(const Cnt &x, const Cnt &y) {
  return Cnt{x.a + y.a, x.b + y.b, x.c + y.c, x.rank + y.rank, x.dup + y.dup};
}
__device__ __forceinline__ Cnt cnt_of(unsigned char f) {
  const bool split = f & F_SPLIT, dup = f & F_DUP;
  return Cnt{(!split && !(f & F_PRUNE_SELF)) ? 1 : 0, (dup && !(f & F_PRUNE_SELF)) ? 1 : 0,
             (split && !(f & F_PRUNE_CHILD)) ? 1 : 0, split ? 1 : 0, dup ? 1 : 0};
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float logitf(float x) { return logf(x / (1.f - x)); }
// scale parameter (log space) of a split child, and the opacity parameter under the "revised opacity" rule
__device__ __forceinline__ float child_scale(float s) { return logf(expf(s) / 1.6f); }
__device__ __forceinline__ float child_opacity(float o) { return logitf(1.f - sqrtf(1.f - sigmoidf(o))); }

__global__ void __launch_bounds__(BLOCK) accumulate_kernel(int C, int N, const float2 *__restrict__ grads,
                                                           const int32_t *__restrict__ radii, float sx, float sy,
                                                           float extent, float *__restrict__ grad2d,
                                                           float *__restrict__ count, float *__restrict__ radii_max) {
  const int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= N) return;
  float g = 0.f, n = 0.f, r = radii_max ? radii_max[i] : 0.f;
  for (int c = 0; c < C; ++c) {
    const int rad = radii[(int64_t)c * N + i];
    if (rad > 0) {
      const float2 v = grads[(int64_t)c * N + i];
      const float gx = v.x * sx, gy = v.y * sy;
      g += sqrtf(gx * gx + gy * gy);
      n += 1.f;
      r = fmaxf(r, (float)rad / extent);
    }
  }
  if (n > 0.f) {
    grad2d[i] += g;
    count[i] += n;
    if (radii_max) radii_max[i] = r;
  }
}

__device__ __forceinline__ unsigned char classify(int i, const float *__restrict__ grad2d,
                                                  const float *__restrict__ count, const float *__restrict__ scales,
                                                  const float *__restrict__ opacities,
                                                  const float *__restrict__ radii_max, const pr_densify_rule &R) {
  const float mean = grad2d[i] / fmaxf(count[i], 1.f);
  const bool high = mean > R.grow_grad2d;
  const float s0 = scales[3 * (int64_t)i], s1 = scales[3 * (int64_t)i + 1], s2 = scales[3 * (int64_t)i + 2];
  const float extent = fmaxf(fmaxf(expf(s0), expf(s1)), expf(s2));
  const bool small = extent <= R.grow_scale3d;
  const float rad = radii_max ? radii_max[i] : 0.f;
  const bool dup = high && small;
  bool split = high && !small;
  if (R.grow_by_radius) split = split || rad > R.grow_scale2d;
  const float o = opacities[i];
  bool gone_self = sigmoidf(o) < R.prune_opa;
  bool gone_child = R.revised_opacity ? sigmoidf(child_opacity(o)) < R.prune_opa : gone_self;
  if (R.prune_big) {
    bool big_self = extent > R.prune_scale3d;
    const float child_extent = fmaxf(fmaxf(expf(child_scale(s0)), expf(child_scale(s1))), expf(child_scale(s2)));
    bool big_child = child_extent > R.prune_scale3d;
    if (R.prune_by_radius) {
      big_self = big_self || rad > R.prune_scale2d;
      big_child = big_child || rad > R.prune_scale2d;
    }
    gone_self = gone_self || big_self;
    gone_child = gone_child || big_child;
  }
  return (unsigned char)((dup ? F_DUP : 0) | (split ? F_SPLIT : 0) | (gone_self ? F_PRUNE_SELF : 0) |
                         (gone_child ? F_PRUNE_CHILD : 0));
}

// Block-wide sum of one Cnt per thread; valid in every thread.  `sh` holds BLOCK / 64 entries.
__device__ __forceinline__ int wave_sum_int(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__global__ void __launch_bounds__(BLOCK) classify_kernel(int N, const float *__restrict__ grad2d,
                                                         const float *__restrict__ count,
                                                         const float *__restrict__ scales,
                                                         const float *__restrict__ opacities,
                                                         const float *__restrict__ radii_max, const pr_densify_rule R,
                                                         unsigned char *__restrict__ flags, Cnt *__restrict__ block_cnt) {
  __shared__ Cnt sh[BLOCK / 64];
  Cnt mine{0, 0, 0, 0, 0};
  // coalesced: thread t takes Gaussians base + k * BLOCK + t (the order inside a block does not matter for its sum)
  const int64_t base = (int64_t)blockIdx.x * PER_BLOCK;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t i = base + k * BLOCK + threadIdx.x;
    if (i < N) {
      const unsigned char f = classify((int)i, grad2d, count, scales, opacities, radii_max, R);
      flags[i] = f;
      mine = mine + cnt_of(f);
    }
  }
  Cnt w{wave_sum_int(mine.a), wave_sum_int(mine.b), wave_sum_int(mine.c), wave_sum_int(mine.rank),
        wave_sum_int(mine.dup)};
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    Cnt t = sh[0];
    for (int k = 1; k < BLOCK / 64; ++k) t = t + sh[k];
    block_cnt[blockIdx.x] = t;
  }
}

// One workgroup: exclusive prefix of the per-block counters (in place) and the grand totals.
// counts_out = {n_out, n_duplicated, n_split, n_pruned}; totals (a, b, c) go to `totals` for the emission.
__global__ void __launch_bounds__(MID_THREADS) block_prefix_kernel(int n_blocks, int N, Cnt *__restrict__ block_cnt,
                                                                   Cnt *__restrict__ totals,
                                                                   int32_t *__restrict__ counts_out) {
  __shared__ Cnt sh[MID_THREADS];
  const int t = threadIdx.x;
  const int per = (n_blocks + MID_THREADS - 1) / MID_THREADS;
  const int lo = min(t * per, n_blocks), hi = min(lo + per, n_blocks);
  Cnt sum{0, 0, 0, 0, 0};
  for (int k = lo; k < hi; ++k) sum = sum + block_cnt[k];
  sh[t] = sum;
  __syncthreads();
  for (int d = 1; d < MID_THREADS; d <<= 1) {  // inclusive Hillis-Steele scan of the thread sums
    Cnt add{0, 0, 0, 0, 0};
    if (t >= d) add = sh[t - d];
    __syncthreads();
    sh[t] = sh[t] + add;
    __syncthreads();
  }
  Cnt run = t == 0 ? Cnt{0, 0, 0, 0, 0} : sh[t - 1];
  for (int k = lo; k < hi; ++k) {
    const Cnt c = block_cnt[k];
    block_cnt[k] = run;
    run = run + c;
  }
  if (t == MID_THREADS - 1) {
    const Cnt all = sh[MID_THREADS - 1];
    *totals = all;
    const int n_out = all.a + all.b + 2 * all.c;
    counts_out[0] = n_out;
    counts_out[1] = all.dup;
    counts_out[2] = all.rank;
    counts_out[3] = N + all.dup + all.rank - n_out;
  }
}

__global__ void __launch_bounds__(BLOCK) emit_kernel(int N, const unsigned char *__restrict__ flags,
                                                     const Cnt *__restrict__ block_cnt, const Cnt *__restrict__ totals,
                                                     int64_t M, int32_t *__restrict__ map_src,
                                                     int32_t *__restrict__ map_tag) {
  __shared__ Cnt sh[BLOCK];
  // thread t takes the ITEMS consecutive Gaussians base + t * ITEMS .. : the scan order is the index order
  const int64_t first = (int64_t)blockIdx.x * PER_BLOCK + (int64_t)threadIdx.x * ITEMS;
  unsigned char f[ITEMS];
  Cnt mine{0, 0, 0, 0, 0};
  if (first + ITEMS <= N) {
    const uint2 raw = *reinterpret_cast<const uint2 *>(flags + first);  // 8 flags, 8-byte aligned
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) f[k] = (unsigned char)(((k < 4 ? raw.x : raw.y) >> (8 * (k & 3))) & 0xffu);
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) f[k] = first + k < N ? flags[first + k] : (unsigned char)F_PRUNE_SELF;
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (first + k < N) mine = mine + cnt_of(f[k]);
  sh[threadIdx.x] = mine;
  __syncthreads();
  for (int d = 1; d < BLOCK; d <<= 1) {
    Cnt add{0, 0, 0, 0, 0};
    if ((int)threadIdx.x >= d) add = sh[threadIdx.x - d];
    __syncthreads();
    sh[threadIdx.x] = sh[threadIdx.x] + add;
    __syncthreads();
  }
  Cnt run = block_cnt[blockIdx.x];
  if (threadIdx.x > 0) run = run + sh[threadIdx.x - 1];
  const Cnt all = *totals;
  const int64_t base_b = all.a, base_c = (int64_t)all.a + all.b, base_d = base_c + all.c;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int64_t i = first + k;
    if (i >= N) break;
    const Cnt c = cnt_of(f[k]);
    if (c.a && run.a < M) {
      map_src[run.a] = (int32_t)i;
      map_tag[run.a] = KIND_ORIGINAL << 30;
    }
    if (c.b && base_b + run.b < M) {
      map_src[base_b + run.b] = (int32_t)i;
      map_tag[base_b + run.b] = KIND_COPY << 30;
    }
    if (c.c && base_d + run.c < M) {
      map_src[base_c + run.c] = (int32_t)i;
      map_tag[base_c + run.c] = (int32_t)(((unsigned)KIND_CHILD0 << 30) | (unsigned)run.rank);
      map_src[base_d + run.c] = (int32_t)i;
      map_tag[base_d + run.c] = (int32_t)(((unsigned)KIND_CHILD1 << 30) | (unsigned)run.rank);
    }
    run = run + c;
  }
}

__global__ void __launch_bounds__(BLOCK) gather_kernel(int64_t total, int width, const float *__restrict__ src,
                                                       const int32_t *__restrict__ map_src,
                                                       const int32_t *__restrict__ map_tag, int zero_new,
                                                       float *__restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= total) return;
  const int64_t m = e / width;
  const int k = (int)(e - m * width);
  float v = 0.f;
  if (!zero_new || ((unsigned)map_tag[m] >> 30) == KIND_ORIGINAL) v = src[(int64_t)map_src[m] * width + k];
  dst[e] = v;
}

__global__ void __launch_bounds__(BLOCK) split_children_kernel(
    int64_t M, const int32_t *__restrict__ map_src, const int32_t *__restrict__ map_tag, int64_t n_split,
    const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ quats,
    const float *__restrict__ opacities, const float *__restrict__ noise, float *__restrict__ means_out,
    float *__restrict__ scales_out, float *__restrict__ opacities_out) {
  const int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (m >= M) return;
  const unsigned tag = (unsigned)map_tag[m];
  const int kind = (int)(tag >> 30);
  if (kind < KIND_CHILD0) return;
  const int64_t rank = tag & 0x3fffffffu, i = map_src[m];
  if (rank >= n_split) return;
  const float *z = noise + ((int64_t)(kind - KIND_CHILD0) * n_split + rank) * 3;
  float w = quats[4 * i], x = quats[4 * i + 1], y = quats[4 * i + 2], q3 = quats[4 * i + 3];
  const float inv = 1.f / fmaxf(sqrtf(w * w + x * x + y * y + q3 * q3), 1e-12f);
  w *= inv, x *= inv, y *= inv, q3 *= inv;
  const float e0 = expf(scales[3 * i]), e1 = expf(scales[3 * i + 1]), e2 = expf(scales[3 * i + 2]);
  const float a0 = e0 * z[0], a1 = e1 * z[1], a2 = e2 * z[2];
  const float r00 = 1.f - 2.f * (y * y + q3 * q3), r01 = 2.f * (x * y - w * q3), r02 = 2.f * (x * q3 + w * y);
  const float r10 = 2.f * (x * y + w * q3), r11 = 1.f - 2.f * (x * x + q3 * q3), r12 = 2.f * (y * q3 - w * x);
  const float r20 = 2.f * (x * q3 - w * y), r21 = 2.f * (y * q3 + w * x), r22 = 1.f - 2.f * (x * x + y * y);
  means_out[3 * m] = means[3 * i] + (r00 * a0 + r01 * a1 + r02 * a2);
  means_out[3 * m + 1] = means[3 * i + 1] + (r10 * a0 + r11 * a1 + r12 * a2);
  means_out[3 * m + 2] = means[3 * i + 2] + (r20 * a0 + r21 * a1 + r22 * a2);
  scales_out[3 * m] = logf(e0 / 1.6f);
  scales_out[3 * m + 1] = logf(e1 / 1.6f);
  scales_out[3 * m + 2] = logf(e2 / 1.6f);
  if (opacities_out) opacities_out[m] = child_opacity(opacities[i]);
}

int n_scan_blocks(int N) { return (N + PER_BLOCK - 1) / PER_BLOCK; }
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// scratch layout: flags [N] | per-block counters [blocks] | totals [1]
size_t off_blocks(int N) { return align256((size_t)N + 8); }
size_t off_totals(int N) { return off_blocks(N) + align256((size_t)n_scan_blocks(N) * sizeof(Cnt)); }

}  // namespace
}  // namespace pr

using namespace pr;

extern "C" int pr_densify_accumulate(int C, int N, const float *grads2d, const int32_t *radii, int width, int height,
                                     int n_cameras, float *grad2d, float *count, float *radii_max, void *stream) {
  if (C < 0 || N < 0 || width <= 0 || height <= 0 || n_cameras <= 0) return PR_EINVAL;
  if (C == 0 || N == 0) return PR_OK;
  if (!grads2d || !radii || !grad2d || !count) return PR_EINVAL;
  if (((uintptr_t)grads2d & 7) != 0) return PR_EINVAL;
  // the scale factors are formed in double and rounded once, as a Python-float times tensor product is
  const float sx = (float)((double)width / 2.0 * (double)n_cameras), sy = (float)((double)height / 2.0 * (double)n_cameras);
  const float extent = (float)(width > height ? width : height);
  hipLaunchKernelGGL(accumulate_kernel, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, C, N,
                     (const float2 *)grads2d, radii, sx, sy, extent, grad2d, count, radii_max);
  return check_launch("pr_densify_accumulate");
}

extern "C" size_t pr_densify_plan_scratch_bytes(int N) {
  if (N <= 0) return 0;
  return off_totals(N) + 256;
}

extern "C" int pr_densify_plan(int N, const float *grad2d, const float *count, const float *scales,
                               const float *opacities, const float *radii_max, const pr_densify_rule *rule,
                               int32_t *counts_dev, void *scratch, size_t scratch_bytes, void *stream) {
  if (N < 0 || !rule || !counts_dev) return PR_EINVAL;
  if (N == 0) {
    PR_HIP(hipMemsetAsync(counts_dev, 0, 4 * sizeof(int32_t), (hipStream_t)stream));
    return PR_OK;
  }
  if (!grad2d || !count || !scales || !opacities || !scratch) return PR_EINVAL;
  if ((rule->grow_by_radius || rule->prune_by_radius) && !radii_max) return PR_EINVAL;
  if (scratch_bytes < pr_densify_plan_scratch_bytes(N) || ((uintptr_t)scratch & 255) != 0) return PR_EINVAL;
  unsigned char *flags = (unsigned char *)scratch;
  Cnt *block_cnt = (Cnt *)((char *)scratch + off_blocks(N));
  Cnt *totals = (Cnt *)((char *)scratch + off_totals(N));
  const int nb = n_scan_blocks(N);
  hipLaunchKernelGGL(classify_kernel, dim3(nb), dim3(BLOCK), 0, (hipStream_t)stream, N, grad2d, count, scales,
                     opacities, radii_max, *rule, flags, block_cnt);
  hipLaunchKernelGGL(block_prefix_kernel, dim3(1), dim3(MID_THREADS), 0, (hipStream_t)stream, nb, N, block_cnt, totals,
                     counts_dev);
  return check_launch("pr_densify_plan");
}

extern "C" int pr_densify_emit(int N, const void *scratch, size_t scratch_bytes, int64_t n_out, int32_t *map_src,
                               int32_t *map_tag, void *stream) {
  if (N < 0 || n_out < 0 || n_out > 3ll * N) return PR_EINVAL;
  if (N == 0 || n_out == 0) return PR_OK;
  if (!scratch || !map_src || !map_tag) return PR_EINVAL;
  if (scratch_bytes < pr_densify_plan_scratch_bytes(N) || ((uintptr_t)scratch & 255) != 0) return PR_EINVAL;
  const unsigned char *flags = (const unsigned char *)scratch;
  const Cnt *block_cnt = (const Cnt *)((const char *)scratch + off_blocks(N));
  const Cnt *totals = (const Cnt *)((const char *)scratch + off_totals(N));
  hipLaunchKernelGGL(emit_kernel, dim3(n_scan_blocks(N)), dim3(BLOCK), 0, (hipStream_t)stream, N, flags, block_cnt,
                     totals, n_out, map_src, map_tag);
  return check_launch("pr_densify_emit");
}

extern "C" int pr_densify_gather(int64_t n_out, int width, const float *src, const int32_t *map_src,
                                 const int32_t *map_tag, int zero_new, float *dst, void *stream) {
  if (n_out < 0 || width <= 0) return PR_EINVAL;
  if (n_out == 0) return PR_OK;
  if (!src || !map_src || !map_tag || !dst) return PR_EINVAL;
  const int64_t total = n_out * width;
  const int64_t blocks = (total + BLOCK - 1) / BLOCK;
  if (blocks > 0x7fffffffll) return PR_EUNSUPPORTED;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)stream, total, width, src,
                     map_src, map_tag, zero_new, dst);
  return check_launch("pr_densify_gather");
}

extern "C" int pr_densify_split_children(int64_t n_out, const int32_t *map_src, const int32_t *map_tag,
                                         int64_t n_split, const float *means, const float *scales, const float *quats,
                                         const float *opacities, const float *noise, float *means_out,
                                         float *scales_out, float *opacities_out, void *stream) {
  if (n_out < 0 || n_split < 0) return PR_EINVAL;
  if (n_out == 0 || n_split == 0) return PR_OK;
  if (!map_src || !map_tag || !means || !scales || !quats || !noise || !means_out || !scales_out) return PR_EINVAL;
  if (opacities_out && !opacities) return PR_EINVAL;
  const int64_t blocks = (n_out + BLOCK - 1) / BLOCK;
  if (blocks > 0x7fffffffll) return PR_EUNSUPPORTED;
  hipLaunchKernelGGL(split_children_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)stream, n_out, map_src,
                     map_tag, n_split, means, scales, quats, opacities, noise, means_out, scales_out, opacities_out);
  return check_launch("pr_densify_split_children");
}
