// Fused multi-group Adam step (pr_adam_step): every parameter group that is due this iteration is updated
// in ONE launch.
//
// Replaces the per-group `torch.optim.Adam(...).step()` calls the trainer issues through
// nerfstudio's Optimizers.optimizer_scaler_step_some (nerfstudio/engine/optimizers.py:159-172, driven by
// nerfstudio/engine/trainer.py:494-510) with the groups, learning rates and eps of pogs/pogs_config.py:38-85.
// The arithmetic is torch.optim.Adam's single-tensor update (betas, eps, no weight decay, no amsgrad,
// maximize=False), in its order of operations:
//   m  = m + (g - m)(1 - beta1)                       exp_avg.lerp_(grad, 1 - beta1)
//   v  = v beta2 + (1 - beta2) g g                     exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The host passes lr / (1 - beta1^t) and sqrt(1 - beta2^t) per group (computed in double, as torch does).
//
// HBM-bound streaming kernel: 16 B read + 12 B written per element.  Workgroups are assigned to
// (group, chunk) pairs through a prefix table in the kernel argument, so small and large groups share one
// grid; each lane handles four consecutive elements with 16-byte accesses.
#include <math.h>

#include "common.h"

namespace pr {
namespace {

constexpr int MAX_GROUPS = 16;
constexpr int BLOCK = 256;
constexpr int PER_BLOCK = BLOCK * 4 * 4;  // elements per workgroup: 4 float4 per lane

struct AdamTable {
  float *param[MAX_GROUPS];
  const float *grad[MAX_GROUPS];
  float *exp_avg[MAX_GROUPS];
  float *exp_avg_sq[MAX_GROUPS];
  int64_t numel[MAX_GROUPS];
  float step_size[MAX_GROUPS], one_minus_beta1[MAX_GROUPS], beta2[MAX_GROUPS], one_minus_beta2[MAX_GROUPS],
      bias2_sqrt[MAX_GROUPS], eps[MAX_GROUPS];
  unsigned first_block[MAX_GROUPS + 1];
  int n;
};

__device__ __forceinline__ void adam_update(float &p, float g, float &m, float &v, float step_size, float omb1,
                                            float beta2, float omb2, float bias2_sqrt, float eps) {
  m = m + (g - m) * omb1;
  v = v * beta2 + omb2 * g * g;
  const float denom = sqrtf(v) / bias2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(BLOCK) adam_kernel(const AdamTable t) {
  int gi = 0;
#pragma unroll 1
  while (gi + 1 < t.n && blockIdx.x >= t.first_block[gi + 1]) ++gi;
  const int64_t n = t.numel[gi];
  float *__restrict__ p = t.param[gi];
  const float *__restrict__ g = t.grad[gi];
  float *__restrict__ m = t.exp_avg[gi];
  float *__restrict__ v = t.exp_avg_sq[gi];
  const float ss = t.step_size[gi], omb1 = t.one_minus_beta1[gi], b2 = t.beta2[gi], omb2 = t.one_minus_beta2[gi],
              bc = t.bias2_sqrt[gi], eps = t.eps[gi];
  const int64_t base = (int64_t)(blockIdx.x - t.first_block[gi]) * PER_BLOCK;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + ((int64_t)r * BLOCK + threadIdx.x) * 4;
    if (i + 3 < n) {
      float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i);
      float4 vv = *reinterpret_cast<float4 *>(v + i);
      const float4 gg = *reinterpret_cast<const float4 *>(g + i);
      adam_update(pp.x, gg.x, mm.x, vv.x, ss, omb1, b2, omb2, bc, eps);
      adam_update(pp.y, gg.y, mm.y, vv.y, ss, omb1, b2, omb2, bc, eps);
      adam_update(pp.z, gg.z, mm.z, vv.z, ss, omb1, b2, omb2, bc, eps);
      adam_update(pp.w, gg.w, mm.w, vv.w, ss, omb1, b2, omb2, bc, eps);
      *reinterpret_cast<float4 *>(p + i) = pp;
      *reinterpret_cast<float4 *>(m + i) = mm;
      *reinterpret_cast<float4 *>(v + i) = vv;
    } else {
      for (int64_t k = i; k < n; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_update(pk, g[k], mk, vk, ss, omb1, b2, omb2, bc, eps);
        p[k] = pk;
        m[k] = mk;
        v[k] = vk;
      }
    }
  }
}

}  // namespace
}  // namespace pr

extern "C" int pr_adam_max_groups(void) { return pr::MAX_GROUPS; }

extern "C" int pr_adam_step(const pr_adam_group *groups, int n_groups, void *stream) {
  if (n_groups < 0 || n_groups > pr::MAX_GROUPS) return PR_EINVAL;
  if (n_groups == 0) return PR_OK;
  if (!groups) return PR_EINVAL;
  pr::AdamTable t;
  t.n = 0;
  unsigned blocks = 0;
  for (int i = 0; i < n_groups; ++i) {
    const pr_adam_group &g = groups[i];
    if (g.numel < 0 || g.step < 1 || !(g.beta1 >= 0.0 && g.beta1 < 1.0) || !(g.beta2 >= 0.0 && g.beta2 < 1.0))
      return PR_EINVAL;
    if (g.numel == 0) continue;
    if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq) return PR_EINVAL;
    // 16-byte accesses: torch allocations are; a sliced view might not be
    if ((((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15) != 0)
      return PR_EINVAL;
    const int k = t.n++;
    t.param[k] = g.param;
    t.grad[k] = g.grad;
    t.exp_avg[k] = g.exp_avg;
    t.exp_avg_sq[k] = g.exp_avg_sq;
    t.numel[k] = g.numel;
    // scalars are formed in double and rounded once, as torch does with its Python-float hyperparameters
    const double bias1 = 1.0 - pow(g.beta1, (double)g.step);
    const double bias2 = 1.0 - pow(g.beta2, (double)g.step);
    t.step_size[k] = (float)(g.lr / bias1);
    t.one_minus_beta1[k] = (float)(1.0 - g.beta1);
    t.beta2[k] = (float)g.beta2;
    t.one_minus_beta2[k] = (float)(1.0 - g.beta2);
    t.bias2_sqrt[k] = (float)sqrt(bias2);
    t.eps[k] = (float)g.eps;
    t.first_block[k] = blocks;
    const int64_t nb = (g.numel + pr::PER_BLOCK - 1) / pr::PER_BLOCK;
    if (nb + blocks > 0x7fffffffll) return PR_EUNSUPPORTED;
    blocks += (unsigned)nb;
  }
  if (t.n == 0) return PR_OK;
  t.first_block[t.n] = blocks;
  hipLaunchKernelGGL(pr::adam_kernel, dim3(blocks), dim3(pr::BLOCK), 0, (hipStream_t)stream, t);
  return pr::check_launch("pr_adam_step");
}
