// The tracker's per-iteration loss as two kernels (value, gradients) instead of the ~30 element-wise and reduction
// launches the same expression costs in PyTorch:
//   loss = nanmean_p || target_p - rendered_p ||_2
//        + mult * mean over kept pixels of (clamp(depth_p * inv_scale, 1e-8, 1) - clamp(target_depth_p, 1e-8, 1))^2
// (pogs/tracking/rigid_group_optimizer.py:364, 373-386: a pixel's depth term is kept when `valid` says so and its
// squared error is below the cut).  Sixteen lanes share a pixel's feature row (float4 each per 64 channels).
#include "common.h"

namespace pr {
namespace {

constexpr float DEPTH_LO = 1e-8f, DEPTH_HI = 1.0f;

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// sum of squared differences of one pixel's rows over the 16 lanes of a DPP row; every lane of the row gets the total
__device__ __forceinline__ float row_sumsq(const float *a, const float *b, int D, int l16) {
  float ss = 0.f;
  for (int c = 4 * l16; c < D; c += 64) {
    const float4 x = *reinterpret_cast<const float4 *>(a + c), y = *reinterpret_cast<const float4 *>(b + c);
    const float d0 = y.x - x.x, d1 = y.y - x.y, d2 = y.z - x.z, d3 = y.w - x.w;
    ss += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  ss = row_sum_to_lane15(ss);
  return __shfl(ss, (threadIdx.x & 63) | 15, 64);
}

// sums[0] += sum of the finite feature norms, [1] += their number, [2] += sum of the kept squared depth errors,
// [3] += their number
__global__ void __launch_bounds__(256) track_loss_sums_kernel(int64_t P, int D, const float *__restrict__ rendered,
                                                              const float *__restrict__ target,
                                                              const float *__restrict__ depth,
                                                              const float *__restrict__ target_depth,
                                                              const uint8_t *__restrict__ valid, float inv_scale,
                                                              float cut2, float *__restrict__ sums) {
  const int l16 = threadIdx.x & 15;
  float s_norm = 0.f, n_norm = 0.f, s_depth = 0.f, n_depth = 0.f;
  for (int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; p < P; p += (int64_t)gridDim.x * 16) {
    const float ss = row_sumsq(rendered + p * D, target + p * D, D, l16);
    if (l16 == 0) {
      const float n = sqrtf(ss);
      if (n == n) {
        s_norm += n;
        n_norm += 1.f;
      }
      if (depth && (!valid || valid[p])) {
        const float e = clampf(depth[p] * inv_scale, DEPTH_LO, DEPTH_HI) - clampf(target_depth[p], DEPTH_LO, DEPTH_HI);
        if (e * e < cut2) {
          s_depth += e * e;
          n_depth += 1.f;
        }
      }
    }
  }
  const float t = quad_reduce(s_norm, n_norm, s_depth, n_depth);  // lane 16r+15: wave total of value {0, 2, 1, 3}[r]
  __shared__ float part[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 15) == 15) part[wave][lane >> 4] = t;
  __syncthreads();
  if (threadIdx.x < 4) {
    const int slot = threadIdx.x == 1 ? 2 : threadIdx.x == 2 ? 1 : (int)threadIdx.x;
    atomicAdd(sums + slot, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
  }
}

__global__ void track_loss_value_kernel(const float *__restrict__ sums, float mult, float *__restrict__ loss) {
  float v = sums[0] / sums[1];                    // no finite pixel: nan, as nanmean of nothing
  if (sums[3] > 0.f) v += mult * sums[2] / sums[3];
  loss[0] = v;
}

// v_rendered[P,D], v_depth[P] (may be NULL) from the cotangent of the loss (a device scalar)
__global__ void __launch_bounds__(256) track_loss_bwd_kernel(int64_t P, int D, const float *__restrict__ rendered,
                                                             const float *__restrict__ target,
                                                             const float *__restrict__ depth,
                                                             const float *__restrict__ target_depth,
                                                             const uint8_t *__restrict__ valid, float inv_scale,
                                                             float cut2, float mult, const float *__restrict__ sums,
                                                             const float *__restrict__ v_loss,
                                                             float *__restrict__ v_rendered, float *__restrict__ v_depth) {
  const int l16 = threadIdx.x & 15;
  const float v = v_loss[0];
  const float w_norm = v / sums[1];
  const float w_depth = sums[3] > 0.f ? v * mult * 2.f * inv_scale / sums[3] : 0.f;
  for (int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; p < P; p += (int64_t)gridDim.x * 16) {
    const float *a = rendered + p * D, *b = target + p * D;
    const float ss = row_sumsq(a, b, D, l16);
    const float n = sqrtf(ss);
    // d||t - r|| / dr = -(t - r) / ||t - r||; nothing from a pixel nanmean skipped, or at distance zero
    const bool live = n == n && n > 0.f;
    const float k = live ? -w_norm / n : 0.f;
    for (int c = 4 * l16; c < D; c += 64) {
      const float4 x = *reinterpret_cast<const float4 *>(a + c), y = *reinterpret_cast<const float4 *>(b + c);
      *reinterpret_cast<float4 *>(v_rendered + p * D + c) =
          live ? make_float4(k * (y.x - x.x), k * (y.y - x.y), k * (y.z - x.z), k * (y.w - x.w))
               : make_float4(0.f, 0.f, 0.f, 0.f);   // (0 * nan would not be 0)
    }
    if (v_depth && l16 == 0) {
      float g = 0.f;
      if (!valid || valid[p]) {
        const float ds = depth[p] * inv_scale;
        const float e = clampf(ds, DEPTH_LO, DEPTH_HI) - clampf(target_depth[p], DEPTH_LO, DEPTH_HI);
        if (e * e < cut2 && ds >= DEPTH_LO && ds <= DEPTH_HI) g = w_depth * e;
      }
      v_depth[p] = g;
    }
  }
}


// ---- depth masks of the per-object crops (pogs/tracking/rigid_group_optimizer.py:316-337) ---------------------------
// Per crop g of size (h_g, w_g) inside a padded [G, Hm, Wm] batch:
//   valid   = accumulation > acc_min and the observed depth is not NaN                       (:316-318)
//   vd      = 5x5 erosion of valid                                                           (:323)
//   median  = lower median of the observed depths over vd; vd &= !(depth > factor * median)  (:324-329)
//   keep    = 9x9 erosion of (vd and depth != 0); masked = keep ? depth : 0                  (:331-337)
// An erosion ignores what lies beyond the crop (kornia's geodesic border).  In PyTorch this is about forty small
// launches and a sort-based median per iteration; here four: vd, the median by radix selection on the float bits (one
// workgroup per crop, four 8-bit passes over an LDS histogram), vd after rejection, the second erosion.
__device__ __forceinline__ bool roi_valid(const float *acc, const float *obs, int64_t i, float acc_min) {
  const float d = obs[i];
  return acc[i] > acc_min && d == d;
}

// every neighbour is loaded whatever the others said (a short-circuit AND would chain 25 or 81 dependent loads)
template <int R, class F>
__device__ __forceinline__ bool eroded(int y, int x, int h, int w, F &&at) {
  int all = 1;
#pragma unroll
  for (int dy = -R; dy <= R; ++dy)
#pragma unroll
    for (int dx = -R; dx <= R; ++dx) {
      const int yy = y + dy, xx = x + dx;
      const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
      const int v = at(min(max(yy, 0), h - 1), min(max(xx, 0), w - 1)) ? 1 : 0;
      all &= v | (inside ? 0 : 1);
    }
  return all != 0;
}

// order-preserving map of float bits to unsigned (negative floats reversed below the positive ones)
__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// grid (pixels of a padded image / 256, G).  Also appends the depth keys of the pixels it keeps to the crop's list
// (any order: only the median is taken), one counter bump per wavefront.
__global__ void __launch_bounds__(256) roi_erode_valid_kernel(int Hm, int Wm, const int32_t *__restrict__ sizes,
                                                              const float *__restrict__ acc,
                                                              const float *__restrict__ obs, float acc_min,
                                                              uint8_t *__restrict__ vd, uint32_t *__restrict__ keys,
                                                              uint32_t *__restrict__ counters) {
  const int g = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  const int64_t base = (int64_t)g * Hm * Wm;
  const int h = sizes[2 * g], w = sizes[2 * g + 1];
  bool v = false;
  const int y = r / Wm, x = r % Wm;
  if (r < Hm * Wm && y < h && x < w)
    v = eroded<2>(y, x, h, w, [&](int yy, int xx) { return roi_valid(acc, obs, base + (int64_t)yy * Wm + xx, acc_min); });
  if (r < Hm * Wm) vd[base + r] = v ? 1 : 0;
  // one counter bump per workgroup (the crops' four counters would otherwise take one per wavefront, in turn)
  __shared__ uint32_t wave_n[4], block_at;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_n[wave] = (uint32_t)__builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t n = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3];
    block_at = n ? atomicAdd(&counters[g], n) : 0u;
  }
  __syncthreads();
  if (v) {
    uint32_t at = block_at;
    for (int j = 0; j < wave; ++j) at += wave_n[j];
    keys[base + at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = float_key(obs[base + r]);
  }
}

__global__ void __launch_bounds__(1024) roi_median_kernel(int Hm, int Wm, const uint32_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ counters,
                                                          float *__restrict__ medians) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t sel_prefix, sel_rank;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const uint32_t *__restrict__ k = keys + (int64_t)g * Hm * Wm;
  const int n = (int)counters[g];
  if (n == 0) {   // (uniform) nothing to take a median of: rejects nothing
    if (tid == 0) medians[g] = __uint_as_float(0x7fc00000u);
    return;
  }
  if (tid == 0) sel_prefix = 0, sel_rank = (uint32_t)(n - 1) / 2;   // torch.median: the lower of the two middle values
  // up to 40 K keys (a 224 x 168 crop has 37.6 K pixels) stay in registers for the four passes: one memory round trip
  constexpr int HELD = 40;
  const bool in_registers = n <= HELD * 1024;
  uint32_t held[HELD];
  if (in_registers) {
#pragma unroll
    for (int j = 0; j < HELD; ++j) held[j] = tid + j * 1024 < n ? k[tid + j * 1024] : 0u;
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sel_prefix, mask_hi = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    // depths of one object share their leading bits: a thread counts runs of equal digits itself and adds a run at a
    // time (one LDS atomic per thread in the first passes instead of one per key on the same word); eight keys are
    // requested before any is used
    uint32_t run_digit = 0, run = 0;
    auto count = [&](uint32_t key) {
      if ((key & mask_hi) != prefix) return;
      const uint32_t d = (key >> shift) & 255u;
      if (d != run_digit && run) {
        atomicAdd(&hist[run_digit], run);
        run = 0;
      }
      run_digit = d;
      ++run;
    };
    if (in_registers) {
#pragma unroll
      for (int j = 0; j < HELD; ++j)
        if (tid + j * 1024 < n) count(held[j]);
    } else {
      for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
        uint32_t kk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kk[j] = i0 + j * 1024 < n ? k[i0 + j * 1024] : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (i0 + j * 1024 < n) count(kk[j]);
      }
    }
    if (run) atomicAdd(&hist[run_digit], run);
    __syncthreads();
    // thread t < 256 <-> digit t: inclusive scan, then the digit whose range holds the wanted rank reports itself
    uint32_t mine = 0, incl = 0;
    if (tid < 256) {
      mine = hist[tid];
      incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if (lane >= d) incl += up;
      }
      if (lane == 63) wave_tot[tid >> 6] = incl;
    }
    __syncthreads();
    if (tid < 256)
      for (int j = 0; j < (tid >> 6); ++j) incl += wave_tot[j];
    const uint32_t rank = sel_rank;
    __syncthreads();
    if (tid < 256 && mine > 0 && rank >= incl - mine && rank < incl) {
      sel_rank = rank - (incl - mine);
      sel_prefix = prefix | ((uint32_t)tid << shift);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t key = sel_prefix;
    medians[g] = __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
  }
}

__global__ void __launch_bounds__(256) roi_reject_kernel(int G, int Hm, int Wm, const float *__restrict__ obs,
                                                         const uint8_t *__restrict__ vd,
                                                         const float *__restrict__ medians, float factor,
                                                         uint8_t *__restrict__ vd2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)G * Hm * Wm) return;
  const int g = (int)(i / ((int64_t)Hm * Wm));
  bool v = vd[i] != 0;
  if (v) {
    const float d = obs[i];
    v = !(d > medians[g] * factor) && d != 0.f;   // (a NaN median compares false: rejects nothing)
  }
  vd2[i] = v ? 1 : 0;
}

__global__ void __launch_bounds__(256) roi_erode_keep_kernel(int G, int Hm, int Wm, const int32_t *__restrict__ sizes,
                                                             const float *__restrict__ obs,
                                                             const uint8_t *__restrict__ vd2,
                                                             uint8_t *__restrict__ keep, float *__restrict__ masked) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)G * Hm * Wm) return;
  const int g = (int)(i / ((int64_t)Hm * Wm)), r = (int)(i % ((int64_t)Hm * Wm)), y = r / Wm, x = r % Wm;
  const int h = sizes[2 * g], w = sizes[2 * g + 1];
  const int64_t base = (int64_t)g * Hm * Wm;
  bool v = false;
  if (y < h && x < w) v = eroded<4>(y, x, h, w, [&](int yy, int xx) { return vd2[base + (int64_t)yy * Wm + xx] != 0; });
  keep[i] = v ? 1 : 0;
  masked[i] = v ? obs[i] : 0.f;
}

}  // namespace
}  // namespace pr

using namespace pr;

extern "C" int pr_track_loss_fwd(int64_t P, int D, const float *rendered, const float *target, const float *depth,
                                 const float *target_depth, const uint8_t *valid, float inv_scale, float cut2,
                                 float mult, float *sums, float *loss, void *stream) {
  if (P <= 0 || D <= 0 || (D & 3) || !rendered || !target || !sums || !loss) return PR_EINVAL;
  if ((depth == nullptr) != (target_depth == nullptr)) return PR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  PR_HIP(hipMemsetAsync(sums, 0, 4 * sizeof(float), s));
  const int64_t groups = (P + 15) / 16;
  // few blocks: every block ends in four atomic adds to the same four floats
  const unsigned grid = (unsigned)(groups < 16 * 1024 ? (groups + 15) / 16 : 1024);
  hipLaunchKernelGGL(track_loss_sums_kernel, dim3(grid), dim3(256), 0, s, P, D, rendered, target, depth, target_depth,
                     valid, inv_scale, cut2, sums);
  hipLaunchKernelGGL(track_loss_value_kernel, dim3(1), dim3(1), 0, s, sums, mult, loss);
  return check_launch("track_loss_fwd");
}

extern "C" int pr_track_loss_bwd(int64_t P, int D, const float *rendered, const float *target, const float *depth,
                                 const float *target_depth, const uint8_t *valid, float inv_scale, float cut2,
                                 float mult, const float *sums, const float *v_loss, float *v_rendered, float *v_depth,
                                 void *stream) {
  if (P <= 0 || D <= 0 || (D & 3) || !rendered || !target || !sums || !v_loss || !v_rendered) return PR_EINVAL;
  if (v_depth && (!depth || !target_depth)) return PR_EINVAL;
  const int64_t groups = (P + 15) / 16;
  const unsigned grid = (unsigned)(groups < 4096 ? groups : 4096);
  hipLaunchKernelGGL(track_loss_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, D, rendered, target, depth,
                     target_depth, valid, inv_scale, cut2, mult, sums, v_loss, v_rendered, v_depth);
  return check_launch("track_loss_bwd");
}

extern "C" size_t pr_roi_depth_masks_scratch_bytes(int G, int Hm, int Wm) {
  if (G <= 0 || Hm <= 0 || Wm <= 0) return 0;
  const size_t total = (size_t)G * Hm * Wm;
  return ((2 * total + 15) & ~(size_t)15) + 4 * total + 4 * (size_t)G;
}

extern "C" int pr_roi_depth_masks(int G, int Hm, int Wm, const int32_t *sizes, const float *accumulation,
                                  const float *observed_depth, float acc_min, float reject_factor, uint8_t *scratch,
                                  float *medians, uint8_t *keep, float *masked_depth, void *stream) {
  if (G < 0 || Hm <= 0 || Wm <= 0) return PR_EINVAL;
  if (G == 0) return PR_OK;
  if (!sizes || !accumulation || !observed_depth || !scratch || !medians || !keep || !masked_depth) return PR_EINVAL;
  if ((((uintptr_t)scratch) & 15) != 0) return PR_EINVAL;
  const int64_t image = (int64_t)Hm * Wm, total = (int64_t)G * image;
  if (image > 0x7fffffffll || G > 65535) return PR_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((total + 255) / 256), grid_image = (unsigned)((image + 255) / 256);
  uint8_t *vd = scratch, *vd2 = scratch + total;
  uint32_t *keys = (uint32_t *)(scratch + ((2 * (size_t)total + 15) & ~(size_t)15)), *counters = keys + total;
  PR_HIP(hipMemsetAsync(counters, 0, 4 * (size_t)G, st));
  hipLaunchKernelGGL(pr::roi_erode_valid_kernel, dim3(grid_image, (unsigned)G), dim3(256), 0, st, Hm, Wm, sizes,
                     accumulation, observed_depth, acc_min, vd, keys, counters);
  hipLaunchKernelGGL(pr::roi_median_kernel, dim3((unsigned)G), dim3(1024), 0, st, Hm, Wm, (const uint32_t *)keys,
                     (const uint32_t *)counters, medians);
  hipLaunchKernelGGL(pr::roi_reject_kernel, dim3(grid), dim3(256), 0, st, G, Hm, Wm, observed_depth, (const uint8_t *)vd,
                     (const float *)medians, reject_factor, vd2);
  hipLaunchKernelGGL(pr::roi_erode_keep_kernel, dim3(grid), dim3(256), 0, st, G, Hm, Wm, sizes, observed_depth,
                     (const uint8_t *)vd2, keep, masked_depth);
  return pr::check_launch("pr_roi_depth_masks");
}
