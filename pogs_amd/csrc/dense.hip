// Bias-free dense layers on the matrix cores, for the CLIP and instance heads that sit on the rendered 96-channel
// feature image (pr_dense_pack / pr_dense_fwd / pr_dense_wgrad, pr_l2norm_fwd / pr_l2norm_bwd).
//
// Replaces the two tiny-cuda-nn CutlassMLP networks of pogs/fields/gaussian_field.py:105-127 (97 -> 256 x 3 -> 512
// and 96 -> 256 x 4 -> 128, ReLU between, no bias) and the normalisation after them (:184-190, :194-204) on the
// paths get_outputs_from_feature / get_instance_outputs_from_feature / get_clip_outputs_from_feature.  The
// reference computes these layers in float16; here inputs, outputs and accumulators are float32 and every product
// is three bf16 matrix instructions (operands split into bf16 head + residual, 2^-16 per product).
//
// One layer Y = act(X A^T) per launch.  A is packed once per call into matrix-operand order, already split:
// [K chunk of <= 256][out tile of 32][k step of 16]{heads, residuals}[lane]{8 values}; a workgroup of four wavefronts
// stages one out tile (<= 32 KB) at a time in LDS, double buffered, and each wavefront multiplies it with ITS 32
// rows of X, which stay in registers, split, for the whole launch.  The result tile has the row of X on the lane
// and four consecutive output columns per register quad, so it leaves as 16-byte stores.  The same kernel does the
// backward data product G_{l-1} = (G_l W_l) * [H_{l-1} > 0] with W packed transposed and the stored activation as
// the mask.  The weight gradient dW[n][k] = sum over rows of G[p][n] H[p][k] contracts over the ROW index, which
// sits on the lanes in memory order; both factors go through LDS images [column][row] (bf16 head and residual,
// written as 16-byte stores by a thread that owns one column and eight rows), from which a lane reads its eight
// consecutive rows as one matrix operand.
//
// Bounds: matrix cores for 256-wide layers (3 x 2 x 256 x 256 flop per row and layer), HBM for the activations
// that travel between layers (2 x 4 x width bytes per row and layer).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace pr {
namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, bf2v));
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a - ha, b - hb}, bf2v));
}
__device__ __forceinline__ void split8(const float (&v)[8], u4v &hi, u4v &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h, l;
    split2(v[2 * i], v[2 * i + 1], h, l);
    hi[i] = h;
    lo[i] = l;
  }
}

constexpr int KCHUNK = 256;  // widest contraction one launch keeps in registers
__host__ __device__ constexpr int padded_chunk(int k) { return k <= 64 ? 64 : k <= 128 ? 128 : 256; }
__host__ __device__ constexpr int64_t chunk_words(int m_tiles, int kc) {  // in 16-byte units
  return (int64_t)m_tiles * (kc / 16) * 64 * 2;
}

// A[m][k] = src[m * stride_m + k * stride_k] for m < M, k < K, else 0.
__global__ void __launch_bounds__(256) dense_pack_kernel(int M, int K, const float *__restrict__ src,
                                                         int64_t stride_m, int64_t stride_k, u4v *__restrict__ packed) {
  const int m_tiles = (M + 31) / 32;
  int64_t unit = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one (chunk, out tile, k step, lane) per thread
  int k0 = 0;
  u4v *dst = packed;
  for (;; k0 += KCHUNK) {
    if (k0 >= K) return;
    const int kc = padded_chunk(min(KCHUNK, K - k0));
    const int64_t units = (int64_t)m_tiles * (kc / 16) * 64;
    if (unit < units) {
      const int lane = (int)(unit & 63), ks = (int)((unit >> 6) % (kc / 16)), ot = (int)((unit >> 6) / (kc / 16));
      const int m = 32 * ot + (lane & 31), kb = k0 + 16 * ks + 8 * (lane >> 5);
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = (m < M && kb + i < K && kb + i < k0 + KCHUNK) ? src[m * stride_m + (kb + i) * stride_k] : 0.f;
      u4v hi, lo;
      split8(v, hi, lo);
      // per (out tile, k step): the 64 head operands, then the 64 residual operands (a wavefront's 16-byte LDS reads of
      // either plane are consecutive: no bank conflicts)
      dst[(unit >> 6) * 128 + lane] = hi;
      dst[(unit >> 6) * 128 + 64 + lane] = lo;
      return;
    }
    unit -= units;
    dst += units * 2;
  }
}

// Y[p][m] (+)= sum_k X[p][k] A[m][k] over one K chunk; KS = k steps of 16 the chunk is padded to.
// (Eight wavefronts per workgroup would halve the L2 -> LDS traffic of the weight tiles: measured slower, 0.375 against
// 0.358 ms per layer.)
constexpr int FWD_THREADS = 256, FWD_ROWS = FWD_THREADS / 2;
template <int KS>
__global__ void __launch_bounds__(FWD_THREADS, 2) dense_fwd_kernel(int64_t P, int M, const float *__restrict__ X, int ldx, int Kx,
                                                        const u4v *__restrict__ packed, int relu,
                                                        const float *__restrict__ mask, int ldm, int accumulate,
                                                        float *__restrict__ Y, int ldy) {
  extern __shared__ u4v wl[];  // two buffers of KS * 128 operands
  constexpr int TILE = KS * 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 31, h = lane >> 5;
  const int64_t pix = ((int64_t)blockIdx.x * (FWD_THREADS / 64) + wave) * 32 + p;
  const int m_tiles = (M + 31) / 32;
  bf8v xh[KS], xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k = 16 * ks + 8 * h;
    float v[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pix < P && k + 4 * q < Kx) t = *reinterpret_cast<const float4 *>(X + pix * ldx + k + 4 * q);
      v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
    u4v hi, lo;
    split8(v, hi, lo);
    xh[ks] = __builtin_bit_cast(bf8v, hi);
    xl[ks] = __builtin_bit_cast(bf8v, lo);
  }
  for (int i = threadIdx.x; i < TILE; i += FWD_THREADS) wl[i] = packed[i];
  __syncthreads();
  // The next out tile travels from memory into registers while the matrix cores work on the current one and goes to
  // LDS after them: a wavefront that copied it straight through would sit out the load latency in front of every tile
  // with only one other wavefront on its SIMD to cover for it.
  constexpr int STAGE = TILE / FWD_THREADS;
  u4v stage[STAGE];
  if (1 < m_tiles) {
#pragma unroll
    for (int i = 0; i < STAGE; ++i) stage[i] = packed[(int64_t)TILE + i * FWD_THREADS + threadIdx.x];
  }
  for (int ot = 0; ot < m_tiles; ++ot) {
    const u4v *cur = wl + (ot & 1) * TILE;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf8v ah = __builtin_bit_cast(bf8v, cur[ks * 128 + lane]);
      const bf8v al = __builtin_bit_cast(bf8v, cur[ks * 128 + 64 + lane]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[ks], acc, 0, 0, 0);
    }
    // (the transposed product -- out column on the lane, 128 consecutive bytes of a row per half wavefront as sixteen
    // 4-byte accesses -- measured slower, 0.47 against 0.36 ms per layer: the 16-byte accesses stay)
    if (pix < P) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m0 = 32 * ot + 8 * q + 4 * h;
        if (m0 >= M) continue;
        float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        float *dst = Y + pix * ldy + m0;
        if (accumulate) {
          const float4 o = *reinterpret_cast<const float4 *>(dst);
          v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
        }
        if (mask) {
          const float4 g = *reinterpret_cast<const float4 *>(mask + pix * ldm + m0);
          v.x = g.x > 0.f ? v.x : 0.f, v.y = g.y > 0.f ? v.y : 0.f;
          v.z = g.z > 0.f ? v.z : 0.f, v.w = g.w > 0.f ? v.w : 0.f;
        }
        if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<float4 *>(dst) = v;
      }
    }
    if (ot + 1 < m_tiles) {
      u4v *nxt = wl + ((ot + 1) & 1) * TILE;
#pragma unroll
      for (int i = 0; i < STAGE; ++i) nxt[i * FWD_THREADS + threadIdx.x] = stage[i];
    }
    __syncthreads();
    if (ot + 2 < m_tiles) {
      const u4v *src = packed + (int64_t)(ot + 2) * TILE;
#pragma unroll
      for (int i = 0; i < STAGE; ++i) stage[i] = src[i * FWD_THREADS + threadIdx.x];
    }
  }
}

// dW[n][k] += sum over rows p of G[p][n] H[p][k].  Workgroup tile: 128 n x 64 k, rows in steps of 32, the steps of a
// tile dealt round-robin over gridDim.z workgroups; totals leave as float atomics.
constexpr int IMG_PITCH = 40;  // 16-bit units per image row: 32 rows of the operand + 8 of padding (80 bytes)
constexpr int WG_N = 128, WG_K = 64;
__global__ void __launch_bounds__(256) dense_wgrad_kernel(int64_t P, int N, int K, const float *__restrict__ G, int ldg,
                                                          const float *__restrict__ H, int ldh,
                                                          float *__restrict__ dW, int ldw) {
  __shared__ __attribute__((aligned(16))) unsigned short img[2][(WG_N + WG_K) * IMG_PITCH];  // [head | residual]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * WG_N, k0 = blockIdx.y * WG_K;
  // loader role: one column of the (G | H) tile and eight consecutive rows per unit; units 0..511 are G, 512..767 H
  const int col = threadIdx.x & 127, oct_g = threadIdx.x >> 7;  // G: columns 0..127, row octets {oct_g, oct_g + 2}
  const int col_h = threadIdx.x & 63, oct_h = threadIdx.x >> 6;  // H: columns 0..63, row octet oct_h
  const int64_t steps = (P + 31) / 32;
  float vg[2][8], vh[8];
  auto fetch = [&](int64_t step) {
    const int64_t base = step * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = base + 8 * (oct_g + 2 * u) + i;
        vg[u][i] = (row < P && n0 + col < N) ? G[row * ldg + n0 + col] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t row = base + 8 * oct_h + i;
      vh[i] = (row < P && k0 + col_h < K) ? H[row * ldh + k0 + col_h] : 0.f;
    }
  };
  f16v acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int m = lane & 31, kh = lane >> 5;
  int64_t step = blockIdx.z;
  if (step < steps) fetch(step);
  for (; step < steps; step += gridDim.z) {
    __syncthreads();  // the previous step's operands have been read
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u4v hi, lo;
      split8(vg[u], hi, lo);
      const int at = col * IMG_PITCH + 8 * (oct_g + 2 * u);
      *reinterpret_cast<u4v *>(&img[0][at]) = hi;
      *reinterpret_cast<u4v *>(&img[1][at]) = lo;
    }
    {
      u4v hi, lo;
      split8(vh, hi, lo);
      const int at = (WG_N + col_h) * IMG_PITCH + 8 * oct_h;
      *reinterpret_cast<u4v *>(&img[0][at]) = hi;
      *reinterpret_cast<u4v *>(&img[1][at]) = lo;
    }
    __syncthreads();
    if (step + gridDim.z < steps) fetch(step + gridDim.z);  // travels while the matrix cores work
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ga = (32 * wave + m) * IMG_PITCH + 16 * s + 8 * kh;
      const bf8v ah = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img[0][ga]));
      const bf8v al = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img[1][ga]));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ha = (WG_N + 32 * t + m) * IMG_PITCH + 16 * s + 8 * kh;
        const bf8v bh = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img[0][ha]));
        const bf8v bl = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img[1][ha]));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
      }
    }
  }
  // result tile t: row = n0 + 32 wave + (r & 3) + 8 (r >> 2) + 4 kh, column = k0 + 32 t + m
  settle_mfma_results();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh, k = k0 + 32 * t + m;
      if (n < N && k < K) atomic_add_noret(dW + (int64_t)n * ldw + k, acc[t][r]);
    }
}

// The same contraction with a dW block of NC (128 or 256) rows n and up to 256 columns k per workgroup of NC / 32
// wavefronts, each owning 64 rows x 16 KT columns (KT x 16 accumulator registers).  The 128 x 64 tiles above re-read G
// four times and H twice per layer from L2 (3.2 GB at 960x540 rows and 256 x 256: the launch is bound by that traffic);
// here G is read once and H once per NC rows of dW.  NC = 128: 60 KB of LDS and four wavefronts, so two workgroups
// share a CU and one loads while the other multiplies.
template <int NC, int KT>
__global__ void __launch_bounds__(NC * 2) dense_wgrad_wide_kernel(int64_t P, int N, int K, const float *__restrict__ G,
                                                                  int ldg, const float *__restrict__ H, int ldh,
                                                                  float *__restrict__ dW, int ldw) {
  extern __shared__ __attribute__((aligned(16))) unsigned short wide_img[];  // [head | residual][(NC + 32 KT) cols]
  constexpr int THREADS = NC * 2, KC = 32 * KT, COLS = NC + KC, PLANE = COLS * IMG_PITCH;
  unsigned short *const img0 = wide_img, *const img1 = wide_img + PLANE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * NC;
  // loader roles: one column and eight consecutive rows per unit.  G: NC columns x 4 row octets = 2 units per thread;
  // H: 32 KT columns x 4 octets = HU units per thread
  const int col = threadIdx.x % NC, oct = threadIdx.x / NC;              // octets {oct, oct + 2}
  constexpr int HU = 4 * KC / THREADS, HS = 4 / HU;                      // units per thread, octet stride
  const int col_h = threadIdx.x % KC, oct_h = threadIdx.x / KC;
  const int64_t steps = (P + 31) / 32;
  float vg[2][8], vh[HU][8];
  auto fetch = [&](int64_t step) {
    const int64_t base = step * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = base + 8 * (oct + 2 * u) + i;
        vg[u][i] = (row < P && n0 + col < N) ? G[row * ldg + n0 + col] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < HU; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = base + 8 * (oct_h + HS * u) + i;
        vh[u][i] = (row < P && col_h < K) ? H[row * ldh + col_h] : 0.f;
      }
  };
  // wavefront (wn, wk): row tiles {2 wn, 2 wn + 1} x column tiles [wk KT/2, (wk + 1) KT/2) -- with a row tile per
  // wavefront and all columns, every wavefront would read the whole H image
  constexpr int KH = KT / 2, WN = NC / 64;
  const int wn = wave % WN, wk = wave / WN;
  f16v acc[2][KH];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
  const int m = lane & 31, kh = lane >> 5;
  int64_t step = blockIdx.z;
  if (step < steps) fetch(step);
  for (; step < steps; step += gridDim.z) {
    __syncthreads();  // the previous step's operands have been read
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u4v hi, lo;
      split8(vg[u], hi, lo);
      const int at = col * IMG_PITCH + 8 * (oct + 2 * u);
      *reinterpret_cast<u4v *>(&img0[at]) = hi;
      *reinterpret_cast<u4v *>(&img1[at]) = lo;
    }
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      u4v hi, lo;
      split8(vh[u], hi, lo);
      const int at = (NC + col_h) * IMG_PITCH + 8 * (oct_h + HS * u);
      *reinterpret_cast<u4v *>(&img0[at]) = hi;
      *reinterpret_cast<u4v *>(&img1[at]) = lo;
    }
    __syncthreads();
    if (step + gridDim.z < steps) fetch(step + gridDim.z);  // travels while the matrix cores work
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf8v ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ga = (64 * wn + 32 * i + m) * IMG_PITCH + 16 * s + 8 * kh;
        ah[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img0[ga]));
        al[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img1[ga]));
      }
#pragma unroll
      for (int t = 0; t < KH; ++t) {
        const int ha = (NC + 32 * (wk * KH + t) + m) * IMG_PITCH + 16 * s + 8 * kh;
        const bf8v bh = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img0[ha]));
        const bf8v bl = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&img1[ha]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][t], 0, 0, 0);
        }
      }
    }
  }
  // result tile (i, t): row = n0 + 64 wn + 32 i + (r & 3) + 8 (r >> 2) + 4 kh, column = 32 (wk KH + t) + m
  settle_mfma_results();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh, k = 32 * (wk * KH + t) + m;
        if (n < N && k < K) atomic_add_noret(dW + (int64_t)n * ldw + k, acc[i][t][r]);
      }
}

template <int NC, int KT>
int launch_wgrad_wide(int64_t P, int N, int K, const float *G, int ldg, const float *H, int ldh, float *dW, int ldw,
                      hipStream_t st) {
  const int64_t steps = (P + 31) / 32;
  const unsigned gx = (unsigned)(N / NC);
  // about one (NC = 256) or two (NC = 128) workgroups per CU, at least four steps each
  int64_t gz = std::max<int64_t>(1, (NC == 256 ? 256 : 512) / gx);
  gz = std::min(gz, std::max<int64_t>(1, steps / 4));
  const size_t lds = (size_t)2 * (NC + 32 * KT) * IMG_PITCH * sizeof(unsigned short);
  PR_HIP(hipFuncSetAttribute((const void *)dense_wgrad_wide_kernel<NC, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((dense_wgrad_wide_kernel<NC, KT>), dim3(gx, 1, (unsigned)gz), dim3(NC * 2), lds, st, P, N, K, G, ldg,
                     H, ldh, dW, ldw);
  return check_launch("pr_dense_wgrad");
}

// One wavefront per row: out = y / (|y| + eps), norms = |y|.
__global__ void __launch_bounds__(256) l2norm_fwd_kernel(int64_t P, int N, const float *__restrict__ y, float eps,
                                                         float *__restrict__ out, float *__restrict__ norms) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  const float4 *src = reinterpret_cast<const float4 *>(y + row * N);
  float ss = 0.f;
  for (int i = lane; i < N / 4; i += 64) {
    const float4 v = src[i];
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  const float n = sqrtf(wave_sum_all(ss)), inv = 1.f / (n + eps);
  float4 *dst = reinterpret_cast<float4 *>(out + row * N);
  for (int i = lane; i < N / 4; i += 64) {
    const float4 v = src[i];
    dst[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  }
  if (lane == 0 && norms) norms[row] = n;
}

// dy = g / (n + eps) - y <g, y> / (n (n + eps)^2)
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(int64_t P, int N, const float *__restrict__ y,
                                                         const float *__restrict__ norms, float eps,
                                                         const float *__restrict__ g, float *__restrict__ dy) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P) return;
  const float4 *ys = reinterpret_cast<const float4 *>(y + row * N), *gs = reinterpret_cast<const float4 *>(g + row * N);
  float dot = 0.f;
  for (int i = lane; i < N / 4; i += 64) {
    const float4 a = ys[i], b = gs[i];
    dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  dot = wave_sum_all(dot);
  // at the zero vector the norm's subgradient is taken as 0 (as torch does): dy = g / eps, finite for eps > 0
  const float n = norms[row], inv = 1.f / (n + eps), c = n > 0.f ? dot * inv * inv / n : 0.f;
  float4 *dst = reinterpret_cast<float4 *>(dy + row * N);
  for (int i = lane; i < N / 4; i += 64) {
    const float4 a = ys[i], b = gs[i];
    dst[i] = make_float4(b.x * inv - a.x * c, b.y * inv - a.y * c, b.z * inv - a.z * c, b.w * inv - a.w * c);
  }
}

bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

template <int KS>
int launch_fwd(int64_t P, int M, const float *X, int ldx, int Kx, const u4v *packed, int relu, const float *mask,
               int ldm, int accumulate, float *Y, int ldy, hipStream_t st) {
  const size_t lds = (size_t)2 * KS * 128 * sizeof(u4v);
  PR_HIP(hipFuncSetAttribute((const void *)dense_fwd_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(dense_fwd_kernel<KS>, dim3((unsigned)((P + FWD_ROWS - 1) / FWD_ROWS)), dim3(FWD_THREADS), lds, st, P, M, X, ldx, Kx,
                     packed, relu, mask, ldm, accumulate, Y, ldy);
  return check_launch("pr_dense_fwd");
}

}  // namespace

static int tuning_wgrad_wide() {
  static const int v = [] {
    const char *e = getenv("POGS_DENSE_WGRAD_WIDE");
    return e ? atoi(e) : 1;
  }();
  return v;
}
}  // namespace pr

extern "C" size_t pr_dense_packed_bytes(int M, int K) {
  if (M <= 0 || K <= 0) return 0;
  const int m_tiles = (M + 31) / 32;
  int64_t words = 0;
  for (int k0 = 0; k0 < K; k0 += pr::KCHUNK) words += pr::chunk_words(m_tiles, pr::padded_chunk(std::min(pr::KCHUNK, K - k0)));
  return (size_t)words * 16;
}

extern "C" int pr_dense_pack(int M, int K, const float *src, int64_t stride_m, int64_t stride_k, void *packed,
                             void *stream) {
  if (M <= 0 || K <= 0 || !src || !packed || !pr::aligned16(packed)) return PR_EINVAL;
  const int64_t units = (int64_t)(pr_dense_packed_bytes(M, K) / 32);
  hipLaunchKernelGGL(pr::dense_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, K,
                     src, stride_m, stride_k, (pr::u4v *)packed);
  return pr::check_launch("pr_dense_pack");
}

extern "C" int pr_dense_fwd(int64_t P, int M, int K, const float *X, int ldx, const void *packed, int relu,
                            const float *mask, int ldm, float *Y, int ldy, void *stream) {
  if (P < 0 || M <= 0 || K <= 0) return PR_EINVAL;
  if (P == 0) return PR_OK;
  if (!X || !packed || !Y || (M & 3) || (K & 3) || (ldx & 3) || (ldy & 3) || ldx < K || ldy < M) return PR_EINVAL;
  if (mask && ((ldm & 3) || ldm < M)) return PR_EINVAL;
  if (!pr::aligned16(X) || !pr::aligned16(Y) || !pr::aligned16(packed) || !pr::aligned16(mask)) return PR_EINVAL;
  const int m_tiles = (M + 31) / 32;
  const pr::u4v *pk = (const pr::u4v *)packed;
  hipStream_t st = (hipStream_t)stream;
  for (int k0 = 0; k0 < K; k0 += pr::KCHUNK) {
    const int kx = std::min(pr::KCHUNK, K - k0), kc = pr::padded_chunk(kx);
    const bool last = k0 + pr::KCHUNK >= K;
    const int rl = last ? relu : 0, acc = k0 > 0;
    const float *mk = last ? mask : nullptr;
    int rc;
    if (kc == 64) rc = pr::launch_fwd<4>(P, M, X + k0, ldx, kx, pk, rl, mk, ldm, acc, Y, ldy, st);
    else if (kc == 128) rc = pr::launch_fwd<8>(P, M, X + k0, ldx, kx, pk, rl, mk, ldm, acc, Y, ldy, st);
    else rc = pr::launch_fwd<16>(P, M, X + k0, ldx, kx, pk, rl, mk, ldm, acc, Y, ldy, st);
    if (rc != PR_OK) return rc;
    pk += pr::chunk_words(m_tiles, kc);
  }
  return PR_OK;
}

extern "C" int pr_dense_wgrad(int64_t P, int N, int K, const float *G, int ldg, const float *H, int ldh, float *dW,
                              int ldw, void *stream) {
  if (P < 0 || N <= 0 || K <= 0) return PR_EINVAL;
  if (P == 0) return PR_OK;
  if (!G || !H || !dW || ldg < N || ldh < K || ldw < K) return PR_EINVAL;
  const int64_t steps = (P + 31) / 32;
  // POGS_DENSE_WGRAD_WIDE=0: the 128 x 64 tiles everywhere (A/B runs).  (128-row blocks, two workgroups per CU, need
  // 48 registers of prefetch next to 128 of accumulators: they spill, 23.8 ms against 12 for the heads' backward.)
  if (pr::tuning_wgrad_wide() && N % 256 == 0 && K <= 256) {
    hipStream_t st = (hipStream_t)stream;
    return K <= 128 ? pr::launch_wgrad_wide<256, 4>(P, N, K, G, ldg, H, ldh, dW, ldw, st)
                    : pr::launch_wgrad_wide<256, 8>(P, N, K, G, ldg, H, ldh, dW, ldw, st);
  }
  const unsigned gx = (unsigned)((N + pr::WG_N - 1) / pr::WG_N), gy = (unsigned)((K + pr::WG_K - 1) / pr::WG_K);
  int64_t gz = std::max<int64_t>(1, 2048 / ((int64_t)gx * gy));  // about eight workgroups per CU in flight
  gz = std::min(gz, std::max<int64_t>(1, steps / 4));           // at least four steps per workgroup
  hipLaunchKernelGGL(pr::dense_wgrad_kernel, dim3(gx, gy, (unsigned)gz), dim3(256), 0, (hipStream_t)stream, P, N, K, G,
                     ldg, H, ldh, dW, ldw);
  return pr::check_launch("pr_dense_wgrad");
}

extern "C" int pr_l2norm_fwd(int64_t P, int N, const float *y, float eps, float *out, float *norms, void *stream) {
  if (P < 0 || N <= 0 || (N & 3)) return PR_EINVAL;
  if (P == 0) return PR_OK;
  if (!y || !out || !pr::aligned16(y) || !pr::aligned16(out)) return PR_EINVAL;
  hipLaunchKernelGGL(pr::l2norm_fwd_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, N, y, eps,
                     out, norms);
  return pr::check_launch("pr_l2norm_fwd");
}

extern "C" int pr_l2norm_bwd(int64_t P, int N, const float *y, const float *norms, float eps, const float *g,
                             float *dy, void *stream) {
  if (P < 0 || N <= 0 || (N & 3)) return PR_EINVAL;
  if (P == 0) return PR_OK;
  if (!y || !norms || !g || !dy || !pr::aligned16(y) || !pr::aligned16(g) || !pr::aligned16(dy)) return PR_EINVAL;
  hipLaunchKernelGGL(pr::l2norm_bwd_kernel, dim3((unsigned)((P + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, N, y,
                     norms, eps, g, dy);
  return pr::check_launch("pr_l2norm_bwd");
}
