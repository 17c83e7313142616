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
