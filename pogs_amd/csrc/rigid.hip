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
#include "common.h"

namespace pr {
namespace {

struct V3 {
  float x, y, z;
};
struct Q4 {  // Hamilton quaternion, vector part first (Warp's storage order)
  float x, y, z, w;
};

__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 vec(Q4 q) { return {q.x, q.y, q.z}; }
__device__ __forceinline__ Q4 conj(Q4 q) { return {-q.x, -q.y, -q.z, q.w}; }

__device__ __forceinline__ V3 rotate(Q4 q, V3 v) {
  V3 qv = vec(q);
  return v * (2.f * q.w * q.w - 1.f) + cross(qv, v) * (q.w * 2.f) + qv * (dot(qv, v) * 2.f);
}
__device__ __forceinline__ Q4 mul(Q4 a, Q4 b) {
  return {a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y, a.w * b.y + b.w * a.y + a.z * b.x - a.x * b.z,
          a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

struct Pose {
  V3 p;
  Q4 q;
};
__device__ __forceinline__ Pose load_pose(const float *__restrict__ poses, int g) {
  const float *r = poses + 7 * (size_t)g;
  return {{r[0], r[1], r[2]}, {r[4], r[5], r[6], r[3]}};
}

constexpr int BLOCK = 256;
constexpr int LDS_GROUPS = 512;  // groups whose gradient rows are reduced in LDS; beyond that: global atomics

__global__ void __launch_bounds__(BLOCK)
rigid_fwd_kernel(int N, int G, const float *__restrict__ init_p2ws, const float *__restrict__ p_deltas,
                 const int *__restrict__ labels, const float *__restrict__ means, const float *__restrict__ quats,
                 float *__restrict__ means_out, float *__restrict__ quats_out) {
  int n = blockIdx.x * BLOCK + threadIdx.x;
  if (n >= N) return;
  int g = labels[n];
  V3 m = {means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2]};
  float4 qm = reinterpret_cast<const float4 *>(quats)[n];  // (w x y z)
  if (g >= 0 && g < G) {
    Pose P = load_pose(init_p2ws, g), D = load_pose(p_deltas, g);
    Q4 q = {qm.y, qm.z, qm.w, qm.x};
    Q4 qpi = conj(P.q);
    V3 u = rotate(qpi, m) + (-rotate(qpi, P.p));  // inverse(p2w) * g2w
    Q4 r = mul(qpi, q);
    V3 t = rotate(P.q, D.p) + P.p;  // p2w * delta
    Q4 s = mul(P.q, D.q);
    m = rotate(s, u) + t;
    Q4 o = mul(s, r);
    qm = make_float4(o.w, o.x, o.y, o.z);
  }
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
  const int rows = G < LDS_GROUPS ? G : LDS_GROUPS;
  for (int i = threadIdx.x; i < rows * 7; i += BLOCK) acc[i] = 0.f;
  __syncthreads();
  int n = blockIdx.x * BLOCK + threadIdx.x;
  int g = n < N ? labels[n] : -1;
  float out[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool valid = g >= 0 && g < G;
  if (valid) {
    Pose P = load_pose(init_p2ws, g), D = load_pose(p_deltas, g);
    V3 m = {means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2]};
    V3 gm = {v_means_out[3 * (size_t)n], v_means_out[3 * (size_t)n + 1], v_means_out[3 * (size_t)n + 2]};
    Q4 qpi = conj(P.q);
    V3 u = rotate(qpi, m) + (-rotate(qpi, P.p));
    Q4 s = mul(P.q, D.q);
    // new_mean = rotate(s, u) + t  ->  cotangents of s and t
    V3 sv = vec(s);
    float vs_w = dot(gm, u * (4.f * s.w) + cross(sv, u) * 2.f);
    V3 vs_v = cross(u, gm) * (2.f * s.w) + gm * (2.f * dot(sv, u)) + u * (2.f * dot(sv, gm));
    if (QUATS) {
      // new_quat = s * r,  r = conj(qp) * q
      float4 qm = reinterpret_cast<const float4 *>(quats)[n];
      float4 hq = reinterpret_cast<const float4 *>(v_quats_out)[n];  // cotangent in (w x y z) order
      Q4 r = mul(qpi, Q4{qm.y, qm.z, qm.w, qm.x});
      V3 hv = {hq.y, hq.z, hq.w}, rv = vec(r);
      vs_w += hq.x * r.w + dot(hv, rv);
      vs_v = vs_v + rv * (-hq.x) + hv * r.w + cross(rv, hv);
    }
    // t = rotate(qp, pd) + pp  ->  v_pd = rotate(conj(qp), v_t) (the rotation matrix transposed)
    V3 v_pd = rotate(qpi, gm);
    // s = qp * qd  ->  cotangent of the right factor
    V3 pv = vec(P.q);
    float v_qd_w = vs_w * P.q.w + dot(vs_v, pv);
    V3 v_qd_v = pv * (-vs_w) + vs_v * P.q.w + cross(vs_v, pv);
    out[0] = v_pd.x, out[1] = v_pd.y, out[2] = v_pd.z, out[3] = v_qd_w;
    out[4] = v_qd_v.x, out[5] = v_qd_v.y, out[6] = v_qd_v.z;
  }
  // Wavefront reduction before any atomic: the tracker has a handful of groups, so the 64 lanes of a wavefront
  // would otherwise hit the same seven addresses.  One pass per group present in the wavefront (wave-uniform
  // loop): lanes of that group contribute, the others add zeros, and one lane issues the seven adds.
  unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int g0 = __builtin_amdgcn_readlane(g, leader);
    const bool mine = valid && g == g0;
    todo &= ~__builtin_amdgcn_ballot_w64(mine);
    float tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) tot[k] = wave_sum_to_lane63(mine ? out[k] : 0.f);
    if ((threadIdx.x & 63) == 63) {
      if (g0 < LDS_GROUPS) {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomicAdd(&acc[g0 * 7 + k], tot[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) atomic_add_noret(v_p_deltas + 7 * (size_t)g0 + k, tot[k]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * 7; i += BLOCK) {
    float v = acc[i];
    if (v != 0.f) atomic_add_noret(v_p_deltas + i, v);
  }
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
