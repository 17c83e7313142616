// Projection (EWA 3D->2D) and spherical-harmonics kernels, forward and backward, for gfx950.
//
// Streaming, HBM-bound kernels: one lane per (camera, Gaussian) forward, one lane per Gaussian
// (looping over cameras) backward so that per-Gaussian gradients need no atomics; the only
// reductions over N (camera-matrix and camera-centre gradients) go wave-DPP -> LDS -> one atomic
// per workgroup and component.
#include "rigid_math.h"

namespace pr {

// Rigid groups applied on the fly (pr_project_fwd_rigid / pr_project_bwd_rigid): Gaussian n is moved by its group's
// p2w * delta * inverse(p2w) before it is projected, so the tracker's per-iteration transform
// (pogs/tracking/rigid_group_optimizer.py:455-479) and its adjoint need no pass over the model of their own.
struct RigidGroups {
  int G;
  const float *init_p2ws, *p_deltas;
  const int32_t *labels;
  float *means_world;          // forward: the moved means [N,3] (the view-dependent colours need them), may be null
  const float *v_means_world;  // backward: a further cotangent of the moved means [N,3], may be null
  float *v_p_deltas;           // backward: [G,7], accumulated
};

struct Cam {
  float R[9];
  float t[3];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ viewmats, const float *__restrict__ Ks, int c) {
  Cam cam;
  const float *V = viewmats + 16 * c;
  const float *K = Ks + 9 * c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) cam.R[3 * i + j] = V[4 * i + j];
    cam.t[i] = V[4 * i + 3];
  }
  cam.fx = K[0];
  cam.fy = K[4];
  cam.cx = K[2];
  cam.cy = K[5];
  return cam;
}

// Camera-space depth of a point, with the order of operations pinned: the projection and pr_view_depths must produce
// the same bits (the depth order built from one is used with the other's visibility).
__device__ __forceinline__ float view_depth(float r20, float r21, float r22, float t2, float x, float y, float z) {
  return __builtin_fmaf(r22, z, __builtin_fmaf(r21, y, r20 * x)) + t2;
}

// Everything the forward computes per (camera, Gaussian); shared by forward and backward.
struct Proj {
  float tx, ty, tz;          // camera-space mean
  float Rq[9];               // rotation of the normalised quaternion
  float qn[4];               // normalised quaternion (w,x,y,z)
  float inv_norm;            // 1/|q|
  float S[6];                // world covariance, upper triangle 00 01 02 11 12 22
  float Sc[6];               // camera-space covariance, same packing
  float J00, J02, J11, J12;  // pinhole Jacobian
  bool x_in, y_in;           // frustum clamp inactive
  float a0, b0, d0;          // 2-D covariance before blur
  float a, d, det;           // after blur; det after the 1e-10 clamp
  float det_orig;
};

__device__ __forceinline__ void sym3_congruence(const float *A /*3x3 row-major*/, const float *S /*sym6*/,
                                                float *out /*sym6*/) {
  // out = A S A^T
  float s[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
  float AS[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) AS[3 * i + j] = A[3 * i] * s[j] + A[3 * i + 1] * s[3 + j] + A[3 * i + 2] * s[6 + j];
  int k = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) out[k++] = AS[3 * i] * A[3 * j] + AS[3 * i + 1] * A[3 * j + 1] + AS[3 * i + 2] * A[3 * j + 2];
}

__device__ __forceinline__ void quat_scale_to_covar(const float *q, const float *s, Proj &P) {
  float n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
  P.inv_norm = inv;
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  P.qn[0] = w; P.qn[1] = x; P.qn[2] = y; P.qn[3] = z;
  P.Rq[0] = 1.f - 2.f * (y * y + z * z); P.Rq[1] = 2.f * (x * y - w * z); P.Rq[2] = 2.f * (x * z + w * y);
  P.Rq[3] = 2.f * (x * y + w * z); P.Rq[4] = 1.f - 2.f * (x * x + z * z); P.Rq[5] = 2.f * (y * z - w * x);
  P.Rq[6] = 2.f * (x * z - w * y); P.Rq[7] = 2.f * (y * z + w * x); P.Rq[8] = 1.f - 2.f * (x * x + y * y);
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[3 * i + j] = P.Rq[3 * i + j] * s[j];
  int k = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) P.S[k++] = M[3 * i] * M[3 * j] + M[3 * i + 1] * M[3 * j + 1] + M[3 * i + 2] * M[3 * j + 2];
}

// Camera-dependent part.  Returns false when the Gaussian is culled by the depth range.
__device__ __forceinline__ bool project_one(const Cam &cam, const float *p, int width, int height, float eps2d,
                                            float near_plane, float far_plane, Proj &P) {
  P.tx = cam.R[0] * p[0] + cam.R[1] * p[1] + cam.R[2] * p[2] + cam.t[0];
  P.ty = cam.R[3] * p[0] + cam.R[4] * p[1] + cam.R[5] * p[2] + cam.t[1];
  P.tz = view_depth(cam.R[6], cam.R[7], cam.R[8], cam.t[2], p[0], p[1], p[2]);
  if (!(P.tz > near_plane && P.tz < far_plane)) return false;
  sym3_congruence(cam.R, P.S, P.Sc);
  float rz = 1.f / P.tz, rz2 = rz * rz;
  float tan_fovx = 0.5f * width / cam.fx, tan_fovy = 0.5f * height / cam.fy;
  float lim_x_pos = (width - cam.cx) / cam.fx + 0.3f * tan_fovx;
  float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
  float lim_y_pos = (height - cam.cy) / cam.fy + 0.3f * tan_fovy;
  float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
  float xr = P.tx * rz, yr = P.ty * rz;
  P.x_in = (xr <= lim_x_pos) && (xr >= -lim_x_neg);
  P.y_in = (yr <= lim_y_pos) && (yr >= -lim_y_neg);
  float txc = P.tz * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
  float tyc = P.tz * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
  P.J00 = cam.fx * rz; P.J02 = -cam.fx * txc * rz2;
  P.J11 = cam.fy * rz; P.J12 = -cam.fy * tyc * rz2;
  // cov2d = J Sc J^T, J = [[J00,0,J02],[0,J11,J12]]
  const float *s = P.Sc;  // 00 01 02 11 12 22
  float r0x = P.J00 * s[0] + P.J02 * s[2];  // (J Sc) row 0
  float r0y = P.J00 * s[1] + P.J02 * s[4];
  float r0z = P.J00 * s[2] + P.J02 * s[5];
  float r1y = P.J11 * s[3] + P.J12 * s[4];
  float r1z = P.J11 * s[4] + P.J12 * s[5];
  P.a0 = r0x * P.J00 + r0z * P.J02;
  P.b0 = r0y * P.J11 + r0z * P.J12;
  P.d0 = r1y * P.J11 + r1z * P.J12;
  P.det_orig = P.a0 * P.d0 - P.b0 * P.b0;
  P.a = P.a0 + eps2d;
  P.d = P.d0 + eps2d;
  P.det = fmaxf(P.a * P.d - P.b0 * P.b0, 1e-10f);
  return true;
}

template <bool RIGID>
__global__ void __launch_bounds__(256) project_fwd_kernel(
    const RigidGroups rigid, int C, int N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats, const float *__restrict__ Ks, int width,
    int height, float eps2d, float near_plane, float far_plane, float radius_clip, int32_t *__restrict__ radii,
    float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
    float *__restrict__ comps) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)C * N) return;
  int c = (int)(idx / N), n = (int)(idx % N);
  Cam cam = load_cam(viewmats, Ks, c);
  float p[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
  float q[4] = {quats[4 * n], quats[4 * n + 1], quats[4 * n + 2], quats[4 * n + 3]};
  float s[3] = {scales[3 * n], scales[3 * n + 1], scales[3 * n + 2]};
  if (RIGID) {
    const int g = rigid.labels[n];
    if (g >= 0 && g < rigid.G) {
      rg::V3 m = {p[0], p[1], p[2]};
      float4 qm = make_float4(q[0], q[1], q[2], q[3]);
      rg::transform(rg::load_pose(rigid.init_p2ws, g), rg::load_pose(rigid.p_deltas, g), m, qm);
      p[0] = m.x, p[1] = m.y, p[2] = m.z;
      q[0] = qm.x, q[1] = qm.y, q[2] = qm.z, q[3] = qm.w;
    }
    if (c == 0 && rigid.means_world) {
      rigid.means_world[3 * n] = p[0];
      rigid.means_world[3 * n + 1] = p[1];
      rigid.means_world[3 * n + 2] = p[2];
    }
  }
  Proj P;
  quat_scale_to_covar(q, s, P);
  int radius = 0;
  float mx = 0.f, my = 0.f, depth = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, comp = 0.f;
  if (project_one(cam, p, width, height, eps2d, near_plane, far_plane, P)) {
    float rz = 1.f / P.tz;
    mx = cam.fx * P.tx * rz + cam.cx;
    my = cam.fy * P.ty * rz + cam.cy;
    float b = 0.5f * (P.a + P.d);
    float v1 = b + sqrtf(fmaxf(0.01f, b * b - P.det));
    float r = ceilf(3.f * sqrtf(v1));
    bool ok = (r > radius_clip) && (mx + r > 0.f) && (mx - r < (float)width) && (my + r > 0.f) &&
              (my - r < (float)height) && (r < 2.0e9f);
    if (ok) {
      radius = (int)r;
      depth = P.tz;
      float inv_det = 1.f / P.det;
      ca = P.d * inv_det;
      cb = -P.b0 * inv_det;
      cc = P.a * inv_det;
      comp = sqrtf(fmaxf(0.f, P.det_orig * inv_det));
    } else {
      mx = 0.f;
      my = 0.f;
    }
  }
  radii[idx] = radius;
  means2d[2 * idx] = mx;
  means2d[2 * idx + 1] = my;
  depths[idx] = depth;
  conics[3 * idx] = ca;
  conics[3 * idx + 1] = cb;
  conics[3 * idx + 2] = cc;
  if (comps) comps[idx] = comp;
}

// Sum `v` over the workgroup (256 threads) and add it to *dst with one atomic.
__device__ __forceinline__ void block_atomic_add(float v, float *dst, float *lds /*4 floats*/) {
  v = wave_sum_to_lane63(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 63) lds[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = lds[0] + lds[1] + lds[2] + lds[3];
    if (s != 0.f) atomicAdd(dst, s);
  }
}

template <bool RIGID>
__global__ void __launch_bounds__(256) project_bwd_kernel(
    const RigidGroups rigid, int C, int N, const float *__restrict__ means, const float *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats, const float *__restrict__ Ks, int width,
    int height, float eps2d, const int32_t *__restrict__ radii, const float *__restrict__ v_means2d, int ld_m,
    const float *__restrict__ v_depths, const float *__restrict__ v_conics, int ld_c,
    const float *__restrict__ v_comps, float *__restrict__ v_means, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_viewmats) {
  __shared__ float red[4];
  __shared__ float acc[RIGID ? rg::LDS_GROUPS * 7 : 1];
  if (RIGID) rg::zero_groups<256>(acc, rigid.G);
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = n < N;
  int nn = live ? n : 0;
  float p[3] = {means[3 * nn], means[3 * nn + 1], means[3 * nn + 2]};
  float q[4] = {quats[4 * nn], quats[4 * nn + 1], quats[4 * nn + 2], quats[4 * nn + 3]};
  float s[3] = {scales[3 * nn], scales[3 * nn + 1], scales[3 * nn + 2]};
  const rg::V3 m_model = {p[0], p[1], p[2]};
  int group = -1;
  if (RIGID) {
    group = live ? rigid.labels[nn] : -1;
    if (group >= 0 && group < rigid.G) {
      rg::V3 m = m_model;
      float4 qm = make_float4(q[0], q[1], q[2], q[3]);
      rg::transform(rg::load_pose(rigid.init_p2ws, group), rg::load_pose(rigid.p_deltas, group), m, qm);
      p[0] = m.x, p[1] = m.y, p[2] = m.z;
      q[0] = qm.x, q[1] = qm.y, q[2] = qm.z, q[3] = qm.w;
    } else {
      group = -1;
    }
  }
  Proj P;
  quat_scale_to_covar(q, s, P);
  float vp[3] = {0.f, 0.f, 0.f};
  float vS[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // gradient w.r.t. the FULL symmetric world covariance,
                                                // packed 00 01 02 11 12 22 (off-diagonals = one of the two equal entries)
  for (int c = 0; c < C; ++c) {
    Cam cam = load_cam(viewmats, Ks, c);
    int64_t idx = (int64_t)c * N + nn;
    float vR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float vt[3] = {0.f, 0.f, 0.f};
    bool act = live && radii[idx] > 0;
    if (act) act = project_one(cam, p, width, height, eps2d, 0.f, 3.0e38f, P);
    if (act) {
      float vmx = v_means2d[idx * ld_m], vmy = v_means2d[idx * ld_m + 1];
      float vca = v_conics[idx * ld_c], vcb = v_conics[idx * ld_c + 1], vcc = v_conics[idx * ld_c + 2];
      float vdepth = v_depths ? v_depths[idx] : 0.f;
      float inv_det = 1.f / P.det;
      float ca = P.d * inv_det, cb = -P.b0 * inv_det, cc = P.a * inv_det;
      // conic = inverse(cov2d'):  G = -inv V inv, V = [[vca, vcb/2],[vcb/2, vcc]]
      float h = 0.5f * vcb;
      float X00 = ca * vca + cb * h, X01 = ca * h + cb * vcc;
      float X10 = cb * vca + cc * h, X11 = cb * h + cc * vcc;
      float G00 = -(X00 * ca + X01 * cb);
      float G01 = -(X00 * cb + X01 * cc);
      float G11 = -(X10 * cb + X11 * cc);
      if (v_comps) {
        float comp = sqrtf(fmaxf(0.f, P.det_orig * inv_det));
        float vcomp = v_comps[idx];
        if (comp > 0.f) {
          float k = vcomp * 0.5f / comp;
          float v_do = k * inv_det;                          // d comp / d det_orig
          float v_d = -k * P.det_orig * inv_det * inv_det;   // d comp / d det
          G00 += v_do * P.d0 + v_d * P.d;
          G11 += v_do * P.a0 + v_d * P.a;
          G01 += -(v_do + v_d) * P.b0;
        }
      }
      // cov2d = J Sc J^T
      const float *sc = P.Sc;
      float Sc[9] = {sc[0], sc[1], sc[2], sc[1], sc[3], sc[4], sc[2], sc[4], sc[5]};
      float J[6] = {P.J00, 0.f, P.J02, 0.f, P.J11, P.J12};
      float GJ[6];  // G J (2x3)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        GJ[j] = G00 * J[j] + G01 * J[3 + j];
        GJ[3 + j] = G01 * J[j] + G11 * J[3 + j];
      }
      // v_Sc = J^T G J  (3x3, symmetric)
      float vSc[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) vSc[3 * i + j] = J[i] * GJ[j] + J[3 + i] * GJ[3 + j];
      // v_J = 2 G J Sc
      float vJ[6];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          vJ[3 * i + j] = 2.f * (GJ[3 * i] * Sc[j] + GJ[3 * i + 1] * Sc[3 + j] + GJ[3 * i + 2] * Sc[6 + j]);
      float rz = 1.f / P.tz, rz2 = rz * rz, rz3 = rz2 * rz;
      float fx = cam.fx, fy = cam.fy;
      float txc = -P.J02 / (fx * rz2), tyc = -P.J12 / (fy * rz2);  // clamped camera-space x,y
      float vtx = 0.f, vty = 0.f, vtz = 0.f;
      vtz += -fx * rz2 * vJ[0] - fy * rz2 * vJ[4];
      if (P.x_in) {
        vtx += -fx * rz2 * vJ[2];
        vtz += 2.f * fx * txc * rz3 * vJ[2];
      } else {
        vtz += fx * txc * rz3 * vJ[2];
      }
      if (P.y_in) {
        vty += -fy * rz2 * vJ[5];
        vtz += 2.f * fy * tyc * rz3 * vJ[5];
      } else {
        vtz += fy * tyc * rz3 * vJ[5];
      }
      // means2d and depth
      vtx += fx * rz * vmx;
      vty += fy * rz * vmy;
      vtz += -(fx * P.tx * vmx + fy * P.ty * vmy) * rz2 + vdepth;
      float vpc[3] = {vtx, vty, vtz};
      // pc = R p + t
#pragma unroll
      for (int j = 0; j < 3; ++j) vp[j] += cam.R[j] * vpc[0] + cam.R[3 + j] * vpc[1] + cam.R[6 + j] * vpc[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        vt[i] = vpc[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) vR[3 * i + j] = vpc[i] * p[j];
      }
      // Sc = R S R^T :  v_S = R^T vSc R ; v_R += 2 vSc R S
      float Sw[9] = {P.S[0], P.S[1], P.S[2], P.S[1], P.S[3], P.S[4], P.S[2], P.S[4], P.S[5]};
      float VR[9];  // vSc R
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          VR[3 * i + j] = vSc[3 * i] * cam.R[j] + vSc[3 * i + 1] * cam.R[3 + j] + vSc[3 * i + 2] * cam.R[6 + j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          vR[3 * i + j] += 2.f * (VR[3 * i] * Sw[j] + VR[3 * i + 1] * Sw[3 + j] + VR[3 * i + 2] * Sw[6 + j]);
      int k = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j)
          vS[k++] += cam.R[i] * VR[j] + cam.R[3 + i] * VR[3 + j] + cam.R[6 + i] * VR[6 + j];
    }
    if (v_viewmats) {
      float *dst = v_viewmats + 16 * c;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) block_atomic_add(vR[3 * i + j], dst + 4 * i + j, red);
        block_atomic_add(vt[i], dst + 4 * i + 3, red);
      }
    }
  }
  if (RIGID) {
    // The model is frozen on this path: the cotangent of the moved mean goes straight on to the group's pose delta
    // (the moved quaternion passes no gradient on, as in the reference, whose tape holds the moved means only).
    float out[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool valid = group >= 0;
    if (valid) {
      rg::V3 gm = {vp[0], vp[1], vp[2]};
      if (rigid.v_means_world) {
        gm.x += rigid.v_means_world[3 * nn];
        gm.y += rigid.v_means_world[3 * nn + 1];
        gm.z += rigid.v_means_world[3 * nn + 2];
      }
      rg::adjoint(rg::load_pose(rigid.init_p2ws, group), rg::load_pose(rigid.p_deltas, group), m_model, gm, nullptr,
                  nullptr, out);
    }
    rg::reduce_to_groups(valid, group, out, acc, rigid.v_p_deltas);
    rg::flush_groups<256>(acc, rigid.G, rigid.v_p_deltas);
    return;
  }
  if (!live) return;
  // S = M M^T, M = Rq diag(s):  v_M = 2 v_S M
  float VS[9] = {vS[0], vS[1], vS[2], vS[1], vS[3], vS[4], vS[2], vS[4], vS[5]};
  float M[9], vM[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[3 * i + j] = P.Rq[3 * i + j] * s[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      vM[3 * i + j] = 2.f * (VS[3 * i] * M[j] + VS[3 * i + 1] * M[3 + j] + VS[3 * i + 2] * M[6 + j]);
  float vs[3], vRq[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    vs[j] = P.Rq[j] * vM[j] + P.Rq[3 + j] * vM[3 + j] + P.Rq[6 + j] * vM[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) vRq[3 * i + j] = vM[3 * i + j] * s[j];
  }
  float w = P.qn[0], x = P.qn[1], y = P.qn[2], z = P.qn[3];
  const float *v = vRq;
  float vq[4];
  vq[0] = 2.f * (z * (v[3] - v[1]) + y * (v[2] - v[6]) + x * (v[7] - v[5]));
  vq[1] = 2.f * (y * (v[1] + v[3]) + z * (v[2] + v[6]) + w * (v[7] - v[5])) - 4.f * x * (v[4] + v[8]);
  vq[2] = 2.f * (x * (v[1] + v[3]) + w * (v[2] - v[6]) + z * (v[5] + v[7])) - 4.f * y * (v[0] + v[8]);
  vq[3] = 2.f * (w * (v[3] - v[1]) + x * (v[2] + v[6]) + y * (v[5] + v[7])) - 4.f * z * (v[0] + v[4]);
  float dot = vq[0] * w + vq[1] * x + vq[2] * y + vq[3] * z;
#pragma unroll
  for (int i = 0; i < 4; ++i) v_quats[4 * n + i] = (vq[i] - dot * P.qn[i]) * P.inv_norm;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v_means[3 * n + i] = vp[i];
    v_scales[3 * n + i] = vs[i];
  }
}

// ------------------------------------------------------------------------------------------------
// Spherical harmonics, degree <= 3 (16 bases), Sloan's fast polynomial basis.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_bases(int nb, float x, float y, float z, float *b) {
  b[0] = 0.2820947917738781f;
  if (nb > 1) {
    b[1] = -0.48860251190292f * y;
    b[2] = 0.48860251190292f * z;
    b[3] = -0.48860251190292f * x;
  }
  if (nb > 4) {
    float z2 = z * z;
    float fTmpB = -1.092548430592079f * z;
    float fC1 = x * x - y * y, fS1 = 2.f * x * y;
    b[4] = 0.5462742152960395f * fS1;
    b[5] = fTmpB * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmpB * x;
    b[8] = 0.5462742152960395f * fC1;
    if (nb > 9) {
      float fTmpC = -2.285228997322329f * z2 + 0.4570457994644658f;
      float fTmpB3 = 1.445305721320277f * z;
      float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
      b[9] = -0.5900435899266435f * fS2;
      b[10] = fTmpB3 * fS1;
      b[11] = fTmpC * y;
      b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
      b[13] = fTmpC * x;
      b[14] = fTmpB3 * fC1;
      b[15] = -0.5900435899266435f * fC2;
    }
  }
}

// d(basis_i)/d(x,y,z) for the same polynomials.
__device__ __forceinline__ void sh_bases_grad(int nb, float x, float y, float z, float *bx, float *by, float *bz) {
#pragma unroll
  for (int i = 0; i < 16; ++i) bx[i] = by[i] = bz[i] = 0.f;
  if (nb > 1) {
    by[1] = -0.48860251190292f;
    bz[2] = 0.48860251190292f;
    bx[3] = -0.48860251190292f;
  }
  if (nb > 4) {
    const float a2 = 0.5462742152960395f, b2 = 1.092548430592079f;
    bx[4] = a2 * 2.f * y; by[4] = a2 * 2.f * x;
    by[5] = -b2 * z; bz[5] = -b2 * y;
    bz[6] = 2.f * 0.9461746957575601f * z;
    bx[7] = -b2 * z; bz[7] = -b2 * x;
    bx[8] = 2.f * a2 * x; by[8] = -2.f * a2 * y;
    if (nb > 9) {
      const float a3 = 0.5900435899266435f, b3 = 1.445305721320277f, c3 = 2.285228997322329f;
      float z2 = z * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
      float fTmpC = -c3 * z2 + 0.4570457994644658f;
      bx[9] = -a3 * 6.f * x * y; by[9] = -a3 * 3.f * fC1;
      bx[10] = b3 * 2.f * y * z; by[10] = b3 * 2.f * x * z; bz[10] = b3 * fS1;
      by[11] = fTmpC; bz[11] = -2.f * c3 * z * y;
      bz[12] = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
      bx[13] = fTmpC; bz[13] = -2.f * c3 * z * x;
      bx[14] = b3 * z * 2.f * x; by[14] = -b3 * z * 2.f * y; bz[14] = b3 * fC1;
      bx[15] = -a3 * 3.f * fC1; by[15] = a3 * 6.f * x * y;
    }
  }
}

// Coefficient rows are 3K floats per Gaussian; a lane reading its own row touches 64 different cache
// lines per load instruction.  For K = 16 (the only case POGS uses) a wavefront instead moves the 64 rows
// it owns as ONE contiguous 12 KiB block of 16-byte accesses and transposes through its LDS slice
// (52-float pitch: 16-byte aligned rows whose 16-byte reads by 16 consecutive lanes hit distinct banks).
constexpr int SH_ROW = 48, SH_PITCH = 52;

__device__ __forceinline__ void wave_rows48_load(const float *__restrict__ src, int64_t row0, int64_t n_rows,
                                                 float *lds, int lane, float *out) {
  const float4 *s4 = reinterpret_cast<const float4 *>(src + row0 * SH_ROW);
  const int64_t avail4 = (n_rows - row0) * (SH_ROW / 4);
#pragma unroll
  for (int i = 0; i < SH_ROW / 4; ++i) {
    const int idx4 = i * 64 + lane;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx4 < avail4) v = s4[idx4];
    const int row = idx4 / (SH_ROW / 4), col4 = idx4 % (SH_ROW / 4);
    *reinterpret_cast<float4 *>(lds + row * SH_PITCH + 4 * col4) = v;
  }
#pragma unroll
  for (int i = 0; i < SH_ROW / 4; ++i) {
    const float4 v = *reinterpret_cast<const float4 *>(lds + lane * SH_PITCH + 4 * i);
    out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
  }
}

__device__ __forceinline__ void wave_rows48_store(float *__restrict__ dst, int64_t row0, int64_t n_rows, float *lds,
                                                  int lane, const float *in) {
#pragma unroll
  for (int i = 0; i < SH_ROW / 4; ++i)
    *reinterpret_cast<float4 *>(lds + lane * SH_PITCH + 4 * i) =
        make_float4(in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]);
  float4 *d4 = reinterpret_cast<float4 *>(dst + row0 * SH_ROW);
  const int64_t avail4 = (n_rows - row0) * (SH_ROW / 4);
#pragma unroll
  for (int i = 0; i < SH_ROW / 4; ++i) {
    const int idx4 = i * 64 + lane;
    const int row = idx4 / (SH_ROW / 4), col4 = idx4 % (SH_ROW / 4);
    const float4 v = *reinterpret_cast<const float4 *>(lds + row * SH_PITCH + 4 * col4);
    if (idx4 < avail4) d4[idx4] = v;
  }
}

// grid = (ceil(N/256), C): block (b, c) handles Gaussians [256 b, 256 b + 256) of camera c.
template <bool K16>
__global__ void __launch_bounds__(256) sh_fwd_kernel(int C, int N, int K, int nb, const float *__restrict__ means,
                                                     const float *__restrict__ campos,
                                                     const float *__restrict__ coeffs,
                                                     const int32_t *__restrict__ radii, int post,
                                                     float *__restrict__ colors, int ld) {
  __shared__ float lds[K16 ? 4 * 64 * SH_PITCH : 4];
  const int c = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = n < N;
  const int nn = live ? n : 0;
  float cf[48];
  if (K16) {
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
    if (row0 < N) wave_rows48_load(coeffs, row0, N, lds + wave * 64 * SH_PITCH, lane, cf);
  } else {
    const float *src = coeffs + (int64_t)nn * K * 3;
#pragma unroll
    for (int i = 0; i < 48; ++i) cf[i] = (i < 3 * nb) ? src[i] : 0.f;
  }
  if (!live) return;
  const int64_t idx = (int64_t)c * N + n;
  float r = 0.f, g = 0.f, bl = 0.f;
  if (radii[idx] > 0) {
    float dx = means[3 * n] - campos[3 * c], dy = means[3 * n + 1] - campos[3 * c + 1],
          dz = means[3 * n + 2] - campos[3 * c + 2];
    float inv = 1.f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    float b[16];
    sh_bases(nb, dx * inv, dy * inv, dz * inv, b);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < nb) {
        r += b[i] * cf[3 * i];
        g += b[i] * cf[3 * i + 1];
        bl += b[i] * cf[3 * i + 2];
      }
  }
  if (post) {
    r = fmaxf(r + 0.5f, 0.f);
    g = fmaxf(g + 0.5f, 0.f);
    bl = fmaxf(bl + 0.5f, 0.f);
  }
  float *o = colors + idx * ld;
  o[0] = r; o[1] = g; o[2] = bl;
}

template <bool K16>
__global__ void __launch_bounds__(256) sh_bwd_kernel(int C, int N, int K, int nb, const float *__restrict__ means,
                                                     const float *__restrict__ campos,
                                                     const float *__restrict__ coeffs,
                                                     const int32_t *__restrict__ radii, int post,
                                                     const float *__restrict__ v_colors, int ld,
                                                     float *__restrict__ v_coeffs, float *__restrict__ v_means,
                                                     float *__restrict__ v_campos) {
  __shared__ float red[4];
  __shared__ float lds[K16 ? 4 * 64 * SH_PITCH : 4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = n < N;
  const int nn = live ? n : 0;
  const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
  float cf[48];
  if (K16) {
    if (row0 < N) wave_rows48_load(coeffs, row0, N, lds + wave * 64 * SH_PITCH, lane, cf);
  } else {
    const float *src = coeffs + (int64_t)nn * K * 3;
#pragma unroll
    for (int i = 0; i < 48; ++i) cf[i] = (i < 3 * nb) ? src[i] : 0.f;
  }
  float vcf[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) vcf[i] = 0.f;
  float vm[3] = {0.f, 0.f, 0.f};
  float p[3] = {means[3 * nn], means[3 * nn + 1], means[3 * nn + 2]};
  for (int c = 0; c < C; ++c) {
    int64_t idx = (int64_t)c * N + nn;
    float vdir[3] = {0.f, 0.f, 0.f};
    if (live && radii[idx] > 0) {
      float dx = p[0] - campos[3 * c], dy = p[1] - campos[3 * c + 1], dz = p[2] - campos[3 * c + 2];
      float inv = 1.f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
      float x = dx * inv, y = dy * inv, z = dz * inv;
      float b[16];
      sh_bases(nb, x, y, z, b);
      float vc[3] = {v_colors[idx * ld], v_colors[idx * ld + 1], v_colors[idx * ld + 2]};
      if (post) {
        float col[3] = {0.5f, 0.5f, 0.5f};
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < nb) {
            col[0] += b[i] * cf[3 * i];
            col[1] += b[i] * cf[3 * i + 1];
            col[2] += b[i] * cf[3 * i + 2];
          }
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (col[k] < 0.f) vc[k] = 0.f;
      }
      float bx[16], by[16], bz[16];
      if (v_means || v_campos) sh_bases_grad(nb, x, y, z, bx, by, bz);
      float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < nb) {
          vcf[3 * i] += b[i] * vc[0];
          vcf[3 * i + 1] += b[i] * vc[1];
          vcf[3 * i + 2] += b[i] * vc[2];
          if (v_means || v_campos) {
            float w = cf[3 * i] * vc[0] + cf[3 * i + 1] * vc[1] + cf[3 * i + 2] * vc[2];
            vx += bx[i] * w;
            vy += by[i] * w;
            vz += bz[i] * w;
          }
        }
      }
      // through d = dir/|dir|
      float dot = vx * x + vy * y + vz * z;
      vdir[0] = (vx - dot * x) * inv;
      vdir[1] = (vy - dot * y) * inv;
      vdir[2] = (vz - dot * z) * inv;
#pragma unroll
      for (int k = 0; k < 3; ++k) vm[k] += vdir[k];
    }
    if (v_campos) {
#pragma unroll
      for (int k = 0; k < 3; ++k) block_atomic_add(-vdir[k], v_campos + 3 * c + k, red);
    }
  }
  if (K16) {
    if (row0 < N) wave_rows48_store(v_coeffs, row0, N, lds + wave * 64 * SH_PITCH, lane, vcf);
  } else if (live) {
    float *o = v_coeffs + (int64_t)n * K * 3;
#pragma unroll
    for (int i = 0; i < 48; ++i)
      if (i < 3 * K) o[i] = vcf[i];
  }
  if (live && v_means) {
    v_means[3 * n] = vm[0];
    v_means[3 * n + 1] = vm[1];
    v_means[3 * n + 2] = vm[2];
  }
}

}  // namespace pr

using namespace pr;

extern "C" int pr_project_fwd(int C, int N, const float *means, const float *quats, const float *scales,
                              const float *viewmats, const float *Ks, int width, int height, float eps2d,
                              float near_plane, float far_plane, float radius_clip, int32_t *radii,
                              float *means2d, float *depths, float *conics, float *compensations, void *stream) {
  if (C < 0 || N < 0 || width <= 0 || height <= 0) return PR_EINVAL;
  if ((int64_t)C * N == 0) return PR_OK;
  if (!means || !quats || !scales || !viewmats || !Ks || !radii || !means2d || !depths || !conics) return PR_EINVAL;
  int64_t total = (int64_t)C * N;
  if (total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(project_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, RigidGroups{}, C, N,
                     means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
                     radii, means2d, depths, conics, compensations);
  return check_launch("project_fwd_kernel");
}

extern "C" int pr_project_fwd_rigid(int C, int N, int G, const float *init_p2ws, const float *p_deltas,
                                    const int32_t *labels, const float *means, const float *quats,
                                    const float *scales, const float *viewmats, const float *Ks, int width, int height,
                                    float eps2d, float near_plane, float far_plane, float radius_clip, int32_t *radii,
                                    float *means2d, float *depths, float *conics, float *compensations,
                                    float *means_world, void *stream) {
  if (C < 0 || N < 0 || G < 0 || width <= 0 || height <= 0) return PR_EINVAL;
  if ((int64_t)C * N == 0) return PR_OK;
  if (!means || !quats || !scales || !viewmats || !Ks || !radii || !means2d || !depths || !conics || !labels)
    return PR_EINVAL;
  if (G > 0 && (!init_p2ws || !p_deltas)) return PR_EINVAL;
  int64_t total = (int64_t)C * N;
  if (total > (int64_t)2147483647) return PR_EUNSUPPORTED;
  const RigidGroups rigid = {G, init_p2ws, p_deltas, labels, means_world, nullptr, nullptr};
  unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(project_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, rigid, C, N, means, quats,
                     scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip, radii, means2d,
                     depths, conics, compensations);
  return check_launch("pr_project_fwd_rigid");
}

extern "C" int pr_project_bwd(int C, int N, const float *means, const float *quats, const float *scales,
                              const float *viewmats, const float *Ks, int width, int height, float eps2d,
                              const int32_t *radii, const float *v_means2d, int ld_means2d, const float *v_depths,
                              const float *v_conics, int ld_conics, const float *v_compensations, float *v_means,
                              float *v_quats, float *v_scales, float *v_viewmats, void *stream) {
  if (C < 0 || N < 0 || width <= 0 || height <= 0 || ld_means2d < 2 || ld_conics < 3) return PR_EINVAL;
  if (N == 0) return PR_OK;
  if (!means || !quats || !scales || !viewmats || !Ks || !radii || !v_means2d || !v_conics || !v_means ||
      !v_quats || !v_scales)
    return PR_EINVAL;
  unsigned grid = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, RigidGroups{}, C, N,
                     means, quats, scales, viewmats, Ks, width, height, eps2d, radii, v_means2d, ld_means2d, v_depths,
                     v_conics, ld_conics, v_compensations, v_means, v_quats, v_scales, v_viewmats);
  return check_launch("project_bwd_kernel");
}

extern "C" int pr_project_bwd_rigid(int C, int N, int G, const float *init_p2ws, const float *p_deltas,
                                    const int32_t *labels, const float *means, const float *quats,
                                    const float *scales, const float *viewmats, const float *Ks, int width, int height,
                                    float eps2d, const int32_t *radii, const float *v_means2d, int ld_means2d,
                                    const float *v_depths, const float *v_conics, int ld_conics,
                                    const float *v_compensations, const float *v_means_world, float *v_p_deltas,
                                    float *v_viewmats, void *stream) {
  if (C < 0 || N < 0 || G < 0 || width <= 0 || height <= 0 || ld_means2d < 2 || ld_conics < 3) return PR_EINVAL;
  if (N == 0 || G == 0) return PR_OK;
  if (!means || !quats || !scales || !viewmats || !Ks || !radii || !v_means2d || !v_conics || !labels || !init_p2ws ||
      !p_deltas || !v_p_deltas)
    return PR_EINVAL;
  const RigidGroups rigid = {G, init_p2ws, p_deltas, labels, nullptr, v_means_world, v_p_deltas};
  unsigned grid = (unsigned)((N + 255) / 256);
  hipLaunchKernelGGL(project_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, rigid, C, N, means, quats,
                     scales, viewmats, Ks, width, height, eps2d, radii, v_means2d, ld_means2d, v_depths, v_conics,
                     ld_conics, v_compensations, nullptr, nullptr, nullptr, v_viewmats);
  return check_launch("pr_project_bwd_rigid");
}

extern "C" int pr_sh_fwd(int C, int N, int K, int degree, const float *means, const float *campos,
                         const float *coeffs, const int32_t *radii, int post, float *colors, int ld_colors,
                         void *stream) {
  if (C < 0 || N < 0 || degree < 0 || ld_colors < 3) return PR_EINVAL;
  if (degree > 3 || K > 16) return PR_EUNSUPPORTED;
  int nb = (degree + 1) * (degree + 1);
  if (nb > K) return PR_EINVAL;
  if ((int64_t)C * N == 0) return PR_OK;
  if (!means || !campos || !coeffs || !radii || !colors) return PR_EINVAL;
  if (C > 65535) return PR_EUNSUPPORTED;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)C);
  if (K == 16)
    hipLaunchKernelGGL(sh_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, C, N, K, nb, means, campos,
                       coeffs, radii, post, colors, ld_colors);
  else
    hipLaunchKernelGGL(sh_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, C, N, K, nb, means, campos,
                       coeffs, radii, post, colors, ld_colors);
  return check_launch("sh_fwd_kernel");
}

extern "C" int pr_sh_bwd(int C, int N, int K, int degree, const float *means, const float *campos,
                         const float *coeffs, const int32_t *radii, int post, const float *v_colors, int ld_colors,
                         float *v_coeffs, float *v_means, float *v_campos, void *stream) {
  if (C < 0 || N < 0 || degree < 0 || ld_colors < 3) return PR_EINVAL;
  if (degree > 3 || K > 16) return PR_EUNSUPPORTED;
  int nb = (degree + 1) * (degree + 1);
  if (nb > K) return PR_EINVAL;
  if (N == 0) return PR_OK;
  if (!means || !campos || !coeffs || !radii || !v_colors || !v_coeffs) return PR_EINVAL;
  unsigned grid = (unsigned)((N + 255) / 256);
  if (K == 16)
    hipLaunchKernelGGL(sh_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, K, nb, means, campos,
                       coeffs, radii, post, v_colors, ld_colors, v_coeffs, v_means, v_campos);
  else
    hipLaunchKernelGGL(sh_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, C, N, K, nb, means,
                       campos, coeffs, radii, post, v_colors, ld_colors, v_coeffs, v_means, v_campos);
  return check_launch("sh_bwd_kernel");
}


// ---- camera centres -------------------------------------------------------------------------------------------
// c = inverse(V)[:3, 3] for rigid world->camera matrices V = [R t; 0 1]: c = -R^T t.  The backward is that of the
// GENERAL matrix inverse, dM = -M dV M with M = inverse(V) = [R^T c; 0 1], so the gradient equals what autograd gives
// for torch.inverse(viewmats)[:, :3, 3] (the expression gsplat's rasterization() differentiates):
//   v_V[i][j] = -u[i] * M[j][3],  u[i] = sum_a M[a][i] v_c[a]  ->  u = (R v_c, <c, v_c>),  M[.][3] = (c, 1).
namespace pr {
namespace {
__global__ void campos_fwd_kernel(int C, const float *__restrict__ viewmats, float *__restrict__ campos) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float *V = viewmats + 16 * (int64_t)c;
#pragma unroll
  for (int j = 0; j < 3; ++j) campos[3 * c + j] = -(V[0 + j] * V[3] + V[4 + j] * V[7] + V[8 + j] * V[11]);
}
__global__ void campos_bwd_kernel(int C, const float *__restrict__ viewmats, const float *__restrict__ v_campos,
                                  float *__restrict__ v_viewmats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float *V = viewmats + 16 * (int64_t)c;
  const float g[3] = {v_campos[3 * c], v_campos[3 * c + 1], v_campos[3 * c + 2]};
  float cc[4], u[4];
#pragma unroll
  for (int j = 0; j < 3; ++j) cc[j] = -(V[0 + j] * V[3] + V[4 + j] * V[7] + V[8 + j] * V[11]);
  cc[3] = 1.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = V[4 * i] * g[0] + V[4 * i + 1] * g[1] + V[4 * i + 2] * g[2];
  u[3] = cc[0] * g[0] + cc[1] * g[1] + cc[2] * g[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) v_viewmats[16 * (int64_t)c + 4 * i + j] = -u[i] * cc[j];
}
}  // namespace
}  // namespace pr

namespace pr {
__global__ void __launch_bounds__(256) view_depths_kernel(int C, int N, const float *__restrict__ means,
                                                          const float *__restrict__ viewmats, float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)C * N) return;
  const int c = (int)(idx / N), n = (int)(idx % N);
  const float *V = viewmats + 16 * c;
  out[idx] = view_depth(V[8], V[9], V[10], V[11], means[3 * n], means[3 * n + 1], means[3 * n + 2]);
}
}  // namespace pr

extern "C" int pr_view_depths(int C, int N, const float *means, const float *viewmats, float *depths, void *stream) {
  if (C < 0 || N < 0) return PR_EINVAL;
  if (C == 0 || N == 0) return PR_OK;
  if (!means || !viewmats || !depths) return PR_EINVAL;
  const int64_t total = (int64_t)C * N;
  hipLaunchKernelGGL(pr::view_depths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, C, N,
                     means, viewmats, depths);
  return pr::check_launch("pr_view_depths");
}

extern "C" int pr_campos_fwd(int C, const float *viewmats, float *campos, void *stream) {
  if (C < 0) return PR_EINVAL;
  if (C == 0) return PR_OK;
  if (!viewmats || !campos) return PR_EINVAL;
  hipLaunchKernelGGL(pr::campos_fwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, viewmats, campos);
  return pr::check_launch("pr_campos_fwd");
}

extern "C" int pr_campos_bwd(int C, const float *viewmats, const float *v_campos, float *v_viewmats, void *stream) {
  if (C < 0) return PR_EINVAL;
  if (C == 0) return PR_OK;
  if (!viewmats || !v_campos || !v_viewmats) return PR_EINVAL;
  hipLaunchKernelGGL(pr::campos_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, viewmats,
                     v_campos, v_viewmats);
  return pr::check_launch("pr_campos_bwd");
}
