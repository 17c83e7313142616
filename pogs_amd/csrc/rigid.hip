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

// One iteration of the tracker's pose optimiser on the [G,7] deltas: torch.optim.Adam's update (the arithmetic of
// pr_adam_step, same order of operations), then the division of the quaternion part by its norm with which the
// reference opens the NEXT iteration (rigid_group_optimizer.py:411-412).  One lane per group.
__global__ void __launch_bounds__(64)
pose_step_kernel(int G, float *__restrict__ deltas, const float *__restrict__ grad, float *__restrict__ exp_avg,
                 float *__restrict__ exp_avg_sq, float step_size, float omb1, float beta2, float omb2, float bias2_sqrt,
                 float eps, int renormalise) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= G) return;
  float p[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int i = 7 * g + k;
    const float gr = grad[i];
    float m = exp_avg[i], v = exp_avg_sq[i];
    m = m + (gr - m) * omb1;
    v = v * beta2 + omb2 * gr * gr;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    p[k] = deltas[i] - step_size * (m / denom);
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
  }
  if (renormalise) {
    const float inv = 1.f / sqrtf(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
#pragma unroll
    for (int k = 3; k < 7; ++k) p[k] = p[k] * inv;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) deltas[7 * g + k] = p[k];
}

}  // namespace
}  // namespace pr

extern "C" int pr_pose_step(int G, float *deltas, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t step,
                            double lr, double beta1, double beta2, double eps, int renormalise, void *stream) {
  if (G < 0 || step < 1 || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0)) return PR_EINVAL;
  if (G == 0) return PR_OK;
  if (!deltas || !grad || !exp_avg || !exp_avg_sq) return PR_EINVAL;
  // scalars are formed in double and rounded once, as torch does with its Python-float hyperparameters
  const double bias1 = 1.0 - pow(beta1, (double)step), bias2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(pr::pose_step_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, (hipStream_t)stream, G, deltas,
                     grad, exp_avg, exp_avg_sq, (float)(lr / bias1), (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)sqrt(bias2), (float)eps, renormalise);
  return pr::check_launch("pr_pose_step");
}

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
