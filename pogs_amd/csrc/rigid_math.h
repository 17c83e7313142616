// Rigid-group transform algebra shared by the standalone kernels (rigid.hip) and the projection kernels that apply
// the transform on the fly (project.hip).  Warp's conventions (pogs/tracking/utils.py:53-93): 7-vectors are
// (x y z qw qx qy qz), quat_rotate(q, v) = v(2w^2-1) + 2w (q x v) + 2 q (q.v), inverse = conjugate, and the product is
// evaluated as (p2w * delta) * (inverse(p2w) * g2w).
#pragma once
#include "common.h"

namespace pr {
namespace rg {

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

constexpr int LDS_GROUPS = 512;  // groups whose gradient rows are reduced in LDS; beyond that: global atomics

// (m, qm) <- p2w * delta * inverse(p2w) * (m, qm); qm in the model's (w x y z) order.
__device__ __forceinline__ void transform(const Pose &P, const Pose &D, V3 &m, float4 &qm) {
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

// Cotangent of delta (x y z qw qx qy qz) from the cotangent gm of the moved mean and, when hq is not null, the
// cotangent *hq (w x y z) of the moved quaternion; m, qm: the Gaussian BEFORE the move.
__device__ __forceinline__ void adjoint(const Pose &P, const Pose &D, V3 m, V3 gm, const float4 *qm, const float4 *hq,
                                        float (&out)[7]) {
  Q4 qpi = conj(P.q);
  V3 u = rotate(qpi, m) + (-rotate(qpi, P.p));
  Q4 s = mul(P.q, D.q);
  // new_mean = rotate(s, u) + t  ->  cotangents of s and t
  V3 sv = vec(s);
  float vs_w = dot(gm, u * (4.f * s.w) + cross(sv, u) * 2.f);
  V3 vs_v = cross(u, gm) * (2.f * s.w) + gm * (2.f * dot(sv, u)) + u * (2.f * dot(sv, gm));
  if (hq) {
    // new_quat = s * r,  r = conj(qp) * q
    Q4 r = mul(qpi, Q4{qm->y, qm->z, qm->w, qm->x});
    V3 hv = {hq->y, hq->z, hq->w}, rv = vec(r);
    vs_w += hq->x * r.w + dot(hv, rv);
    vs_v = vs_v + rv * (-hq->x) + hv * r.w + cross(rv, hv);
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

// Wavefront reduction before any atomic: the tracker has a handful of groups, so the 64 lanes of a wavefront would
// otherwise hit the same seven addresses.  One pass per group present in the wavefront (wave-uniform loop): lanes of
// that group contribute, the others add zeros, and one lane issues the seven adds -- into the workgroup's LDS rows
// `acc` (zeroed by the caller, flushed by flush_groups) for groups below LDS_GROUPS, else to global memory.
__device__ __forceinline__ void reduce_to_groups(bool valid, int g, const float (&out)[7], float *acc,
                                                 float *__restrict__ v_p_deltas) {
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
}
template <int BLOCK>
__device__ __forceinline__ void zero_groups(float *acc, int G) {
  const int rows = G < LDS_GROUPS ? G : LDS_GROUPS;
  for (int i = threadIdx.x; i < rows * 7; i += BLOCK) acc[i] = 0.f;
  __syncthreads();
}
template <int BLOCK>
__device__ __forceinline__ void flush_groups(const float *acc, int G, float *__restrict__ v_p_deltas) {
  const int rows = G < LDS_GROUPS ? G : LDS_GROUPS;
  __syncthreads();
  for (int i = threadIdx.x; i < rows * 7; i += BLOCK) {
    float v = acc[i];
    if (v != 0.f) atomic_add_noret(v_p_deltas + i, v);
  }
}

}  // namespace rg
}  // namespace pr
