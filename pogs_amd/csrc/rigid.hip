// Rigid-group transform of a Gaussian model and its adjoint (pr_rigid_fwd / pr_rigid_bwd).
//
// Replaces the Warp kernel `apply_to_model` (pogs/tracking/utils.py:53-93) and the `wp.Tape` backward that
// the tracker runs after `loss.backward()` (pogs/tracking/rigid_group_optimizer.py:413-430): every Gaussian
// n with group label g is moved by  new = p2w[g] * delta[g] * inverse(p2w[g]) * (mean[n], quat[n]),
// 7-vectors are (x y z qw qx qy qz), model quaternions are (w x y z).  The arithmetic follows Warp's
// transform algebra: quat_rotate(q, v) = v(2w^2-1) + 2w (q x v) + 2 q (q.v), inverse = conjugate, and the
// product is evaluated in the reference's order, (p2w * delta) * (inverse(p2w) * g2w).
//
// HBM-bound streaming kernels: 32 B read + 28 B written per Gaussian forward, 32 B (+12/28 B of cotangent)
// read backward; the [G,7] gradient is reduced in LDS per block and leaves as at most 7*G float atomics
// per block.
#include "rigid_math.h"

namespace pr {
namespace {
using namespace rg;

constexpr int BLOCK = 256;

__global__ void __launch_bounds__(BLOCK)
rigid_fwd_kernel(int N, int G, const float *__restrict__ init_p2ws, const float *__restrict__ p_deltas,
                 const int *__restrict__ labels, const float *__restrict__ means, const float *__restrict__ quats,
                 float *__restrict__ means_out, float *__restrict__ quats_out) {
  int n = blockIdx.x * BLOCK + threadIdx.x;
  if (n >= N) return;
  int g = labels[n];
  V3 m = {means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2]};
  float4 qm = reinterpret_cast<const float4 *>(quats)[n];  // (w x y z)
  if (g >= 0 && g < G) transform(load_pose(init_p2ws, g), load_pose(p_deltas, g), m, qm);
  means_out[3 * (size_t)n] = m.x;
  means_out[3 * (size_t)n + 1] = m.y;
  means_out[3 * (size_t)n + 2] = m.z;
  reinterpret_cast<float4 *>(quats_out)[n] = qm;
}

template <bool QUATS>
__global__ void __launch_bounds__(BLOCK)
rigid_bwd_kernel(int N, int G, const float *__restrict__ init_p2ws, const float *__restrict__ p_deltas,
                 const int *__restrict__ labels, const float *__restrict__ means, const float *__restrict__ quats,
                 const float *__restrict__ v_means_out, const float *__restrict__ v_quats_out,
                 float *__restrict__ v_p_deltas) {
  __shared__ float acc[LDS_GROUPS * 7];
  zero_groups<BLOCK>(acc, G);
  int n = blockIdx.x * BLOCK + threadIdx.x;
  int g = n < N ? labels[n] : -1;
  float out[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool valid = g >= 0 && g < G;
  if (valid) {
    V3 m = {means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2]};
    V3 gm = {v_means_out[3 * (size_t)n], v_means_out[3 * (size_t)n + 1], v_means_out[3 * (size_t)n + 2]};
    float4 qm = make_float4(0.f, 0.f, 0.f, 0.f), hq = qm;
    if (QUATS) {
      qm = reinterpret_cast<const float4 *>(quats)[n];
      hq = reinterpret_cast<const float4 *>(v_quats_out)[n];  // cotangent in (w x y z) order
    }
    adjoint(load_pose(init_p2ws, g), load_pose(p_deltas, g), m, gm, QUATS ? &qm : nullptr, QUATS ? &hq : nullptr, out);
  }
  reduce_to_groups(valid, g, out, acc, v_p_deltas);
  flush_groups<BLOCK>(acc, G, v_p_deltas);
}

}  // namespace
}  // namespace pr

extern "C" int pr_rigid_fwd(int64_t N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                            const float *means, const float *quats, float *means_out, float *quats_out,
                            void *stream) {
  if (N < 0 || G < 0 || N > (int64_t)INT32_MAX) return PR_EINVAL;
  if (N == 0) return PR_OK;
  if (!labels || !means || !quats || !means_out || !quats_out) return PR_EINVAL;
  if (G > 0 && (!init_p2ws || !p_deltas)) return PR_EINVAL;
  unsigned blocks = (unsigned)((N + pr::BLOCK - 1) / pr::BLOCK);
  hipLaunchKernelGGL(pr::rigid_fwd_kernel, dim3(blocks), dim3(pr::BLOCK), 0, (hipStream_t)stream, (int)N, G,
                     init_p2ws, p_deltas, labels, means, quats, means_out, quats_out);
  return pr::check_launch("pr_rigid_fwd");
}

extern "C" int pr_rigid_bwd(int64_t N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                            const float *means, const float *quats, const float *v_means_out,
                            const float *v_quats_out, float *v_p_deltas, void *stream) {
  if (N < 0 || G < 0 || N > (int64_t)INT32_MAX) return PR_EINVAL;
  if (N == 0 || G == 0) return PR_OK;
  if (!labels || !means || !quats || !v_means_out || !v_p_deltas || !init_p2ws || !p_deltas) return PR_EINVAL;
  unsigned blocks = (unsigned)((N + pr::BLOCK - 1) / pr::BLOCK);
  if (v_quats_out)
    hipLaunchKernelGGL(pr::rigid_bwd_kernel<true>, dim3(blocks), dim3(pr::BLOCK), 0, (hipStream_t)stream, (int)N, G,
                       init_p2ws, p_deltas, labels, means, quats, v_means_out, v_quats_out, v_p_deltas);
  else
    hipLaunchKernelGGL(pr::rigid_bwd_kernel<false>, dim3(blocks), dim3(pr::BLOCK), 0, (hipStream_t)stream, (int)N, G,
                       init_p2ws, p_deltas, labels, means, quats, v_means_out, v_quats_out, v_p_deltas);
  return pr::check_launch("pr_rigid_bwd");
}
