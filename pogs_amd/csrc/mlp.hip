// Alpha-normalise + 4 x (64 -> 64, no bias, ReLU between) channel MLP on the rendered feature image, fused,
// on the matrix cores (pr_mlp_fwd / pr_mlp_bwd_input).
//
// Replaces pogs/pogs.py:516-522 (normalise by the clamped alpha, `self.nn(...)`, optional zeroing where
// alpha < 0.8) with the module defined at pogs/pogs.py:153-161, and -- for callers that optimise something
// upstream of the render with the weights frozen, as the tracker does -- its gradient with respect to the
// rendered features.
//
// Arithmetic, two variants.  Exact float32: v_mfma_f32_32x32x2_f32 is bit for bit an fmaf chain in k order, so the
// result differs from torch's sgemm only by summation order.  Split bf16 (default, see below): 8x the matrix rate.  The activations stay in registers from the first layer
// to the last, kept TRANSPOSED: a 32 (features) x 32 (pixels) MFMA result has the pixel on the lane and the
// feature in the register index, which is exactly the B-operand layout of the next layer's product
// H_out^T = W . H_in^T (sum over H_in^T's row index): no lane movement, no LDS round trip between layers.
// Register r of lane half h holds feature  8 (r >> 2) + 4 h + (r & 3)  of its 32-feature tile; the weights
// are staged in LDS as [layer][k / 4][m][k % 4] so that a lane fetches the four A operands of registers
// 4q .. 4q+3 with one 16-byte read.  The forward kernel records the ReLU signs (3 layers x 64 bits per
// pixel) so that the backward kernel does not recompute the forward pass.
//
// Bound: matrix cores.  4 layers x 64 x 64 x 2 = 32.8 kFLOP per pixel at the f32 MFMA rate (64 FLOP/clk/SIMD).
// Workgroups walk pixel groups persistently so the 64 KB of weights are staged once per workgroup; two
// workgroups of eight wavefronts per CU (4 wavefronts per SIMD) keep the matrix pipe fed while other
// wavefronts load, normalise and store.  Measured for a 1080p image: float32 0.69 ms (99 TFLOP/s, matrix-bound),
// split bf16 0.345 ms (516 B/pixel -> 3.1 TB/s: conversions, ReLU bookkeeping and HBM, no longer the matrix pipe).
#include <stdlib.h>

#include "common.h"

namespace pr {
namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int DIM = 64, LAYERS = 4;

// Stage the four weight matrices into LDS in A-operand order.  TRANSPOSE = false: forward, A[m = out][k = in] =
// W[m][k]; true: backward, A[m = in][k = out] = W[k][m].  Layout: [layer][k >> 2][m][k & 3].
struct Weights {
  const float *w0, *w1, *w2, *w3;
};
template <bool TRANSPOSE, int BLOCK>
__device__ __forceinline__ void stage_weights(float *lds, const Weights &W) {
#pragma unroll
  for (int l = 0; l < LAYERS; ++l) {
    const float *__restrict__ w = l == 0 ? W.w0 : l == 1 ? W.w1 : l == 2 ? W.w2 : W.w3;
    for (int i = threadIdx.x; i < DIM * DIM; i += BLOCK) {
      const int row = i / DIM, col = i % DIM;  // W[row = out][col = in]
      const int m = TRANSPOSE ? col : row, k = TRANSPOSE ? row : col;
      lds[((l * (DIM / 4) + (k >> 2)) * DIM + m) * 4 + (k & 3)] = w[i];
    }
  }
}

// H_out^T (PT pixel tiles x 2 feature tiles) = A(layer) . H_in^T for one wavefront.
template <int PT>
__device__ __forceinline__ void layer_product(const float *lds_layer, int lane, const f16v (&in)[PT][2],
                                              f16v (&out)[PT][2]) {
  const int m = lane & 31, h = lane >> 5;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[pt][mt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kg = kt * 8 + 2 * q + h;  // group of four k = 32 kt + 8 q + 4 h + (0..3)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float4 a = *reinterpret_cast<const float4 *>(lds_layer + (kg * DIM + mt * 32 + m) * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            out[pt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], in[pt][kt][4 * q + j], out[pt][mt], 0, 0, 0);
      }
    }
}

// ---- split-bf16 variant --------------------------------------------------------------------------------------
// Each product is done as hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_bf16 (operands = bf16 head + bf16 residual:
// 16 mantissa bits, error <= 2^-16 per product) at 8x the f32 matrix rate.  The register layout of the activations
// is unchanged: a k step of the next product takes registers 8s .. 8s+7 of a feature tile as they are, because
// the order of the summation index is free -- the weights are staged in LDS in the matching order, already split,
// as [layer][out tile][k step][lane]{hi, lo} (16 bytes each): element i of lane (m, h) at k step (kt, s) is
// W[32 mo + m][32 kt + feature(8 s + i, h)],  feature(r, h) = 8 (r >> 2) + 4 h + (r & 3).
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a, b}, bf2v));
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{a - ha, b - hb}, bf2v));
}

template <bool TRANSPOSE, int BLOCK>
__device__ __forceinline__ void stage_weights_bf16(float *lds_f, const Weights &W) {
  unsigned *lds = reinterpret_cast<unsigned *>(lds_f);
  // one thread per (layer, out tile, k step, lane, element pair): 4 * 2 * 4 * 64 * 4 = 8192 pairs
  for (int idx = threadIdx.x; idx < LAYERS * 2 * 4 * 64 * 4; idx += BLOCK) {
    const int pair = idx & 3, lane = (idx >> 2) & 63, kk = (idx >> 8) & 3, mo = (idx >> 10) & 1, l = idx >> 11;
    const float *__restrict__ w = l == 0 ? W.w0 : l == 1 ? W.w1 : l == 2 ? W.w2 : W.w3;
    const int m = 32 * mo + (lane & 31), h = lane >> 5;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 8 * (kk & 1) + 2 * pair + e;
      const int k = 32 * (kk >> 1) + 8 * (r >> 2) + 4 * h + (r & 3);
      v[e] = TRANSPOSE ? w[k * DIM + m] : w[m * DIM + k];
    }
    unsigned hi, lo;
    split_pair(v[0], v[1], hi, lo);
    const int base = ((((l * 2 + mo) * 4 + kk) * 64 + lane) * 2) * 4;  // in 32-bit words: {hi[4], lo[4]}
    lds[base + pair] = hi;
    lds[base + 4 + pair] = lo;
  }
}

template <int PT>
__device__ __forceinline__ void layer_product_bf16(const float *lds_f, int layer, int lane, const f16v (&in)[PT][2],
                                                   f16v (&out)[PT][2]) {
  const u4v *lds = reinterpret_cast<const u4v *>(lds_f);
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[pt][mt][r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf8v ah[2], al[2];
#pragma unroll
    for (int mo = 0; mo < 2; ++mo) {
      const int at = (((layer * 2 + mo) * 4 + kk) * 64 + lane) * 2;
      ah[mo] = __builtin_bit_cast(bf8v, lds[at]);
      al[mo] = __builtin_bit_cast(bf8v, lds[at + 1]);
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      unsigned bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        split_pair(in[pt][kk >> 1][8 * (kk & 1) + 2 * i], in[pt][kk >> 1][8 * (kk & 1) + 2 * i + 1], bh[i], bl[i]);
      const bf8v vh = __builtin_bit_cast(bf8v, u4v{bh[0], bh[1], bh[2], bh[3]});
      const bf8v vl = __builtin_bit_cast(bf8v, u4v{bl[0], bl[1], bl[2], bl[3]});
#pragma unroll
      for (int mo = 0; mo < 2; ++mo) {
        out[pt][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mo], vh, out[pt][mo], 0, 0, 0);
        out[pt][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mo], vh, out[pt][mo], 0, 0, 0);
        out[pt][mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mo], vl, out[pt][mo], 0, 0, 0);
      }
    }
  }
}

// Rows of a [P, 64] image <-> the transposed register layout; pixels past P read as zero and are not written.
template <int PT>
__device__ __forceinline__ void load_rows(const float *__restrict__ src, int64_t first_pixel, int64_t P, int lane,
                                          f16v (&t)[PT][2]) {
  const int p = lane & 31, h = lane >> 5;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int64_t pix = first_pixel + pt * 32 + p;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix < P) v = *reinterpret_cast<const float4 *>(src + pix * DIM + kt * 32 + 8 * q + 4 * h);
        t[pt][kt][4 * q] = v.x;
        t[pt][kt][4 * q + 1] = v.y;
        t[pt][kt][4 * q + 2] = v.z;
        t[pt][kt][4 * q + 3] = v.w;
      }
  }
}
template <int PT>
__device__ __forceinline__ void store_rows(float *__restrict__ dst, int64_t first_pixel, int64_t P, int lane,
                                           const f16v (&t)[PT][2]) {
  const int p = lane & 31, h = lane >> 5;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int64_t pix = first_pixel + pt * 32 + p;
    if (pix >= P) continue;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(dst + pix * DIM + kt * 32 + 8 * q + 4 * h) =
            make_float4(t[pt][kt][4 * q], t[pt][kt][4 * q + 1], t[pt][kt][4 * q + 2], t[pt][kt][4 * q + 3]);
  }
}

// ReLU sign bits: each lane stores the 32 bits (bit mt*16 + r) of ITS registers per 32-pixel tile and layer,
// masks[(tile * 64 + lane) * 3 + layer] as 32-bit words; the backward kernel reads them back in the same layout.
// PT: 32-pixel tiles per wavefront; WPE: wavefronts per SIMD the register budget is sized for; HOIST: let the
// compiler keep all four weight matrices in registers across the persistent loop (256 registers, WPE = 1).
template <int PT, int WPE, int BLOCK, bool HOIST, bool BF16>
__global__ void __launch_bounds__(BLOCK, WPE) mlp_fwd_kernel(int64_t P, const float *__restrict__ feats,
                                                        const float *__restrict__ alphas, const Weights W,
                                                        float zero_below, float *__restrict__ out,
                                                        unsigned *__restrict__ masks, float *__restrict__ acts) {
  extern __shared__ float lds[];
  if (BF16) stage_weights_bf16<false, BLOCK>(lds, W);
  else stage_weights<false, BLOCK>(lds, W);
  constexpr int PIX_PER_WAVE = 32 * PT;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_groups = (P + PIX_PER_WAVE - 1) / PIX_PER_WAVE;
  for (int64_t grp = (int64_t)blockIdx.x * (BLOCK / 64) + wave; grp < n_groups;
       grp += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int64_t first = grp * PIX_PER_WAVE;
    if (!HOIST) asm volatile("" ::: "memory");  // re-read the weights from LDS per group
    f16v a[PT][2], b[PT][2];
    load_rows<PT>(feats, first, P, lane, a);
    float al[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) al[pt] = 1.f;
    if (alphas) {
      // x = alpha > 0 ? x / max(alpha, 1e-6) : 0   (true division, as torch)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int64_t pix = first + pt * 32 + (lane & 31);
        al[pt] = pix < P ? alphas[pix] : 0.f;
        const float d = fmaxf(al[pt], 1e-6f), inv = al[pt] > 0.f ? 1.f / d : 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)  // float32: true division, as torch; split-bf16: one reciprocal (1 ulp)
            a[pt][kt][r] = BF16 ? a[pt][kt][r] * inv : (al[pt] > 0.f ? a[pt][kt][r] / d : 0.f);
      }
    }
#pragma unroll
    for (int l = 0; l < LAYERS; ++l) {
      f16v(&src)[PT][2] = (l & 1) ? b : a;
      f16v(&dst)[PT][2] = (l & 1) ? a : b;
      if (BF16) layer_product_bf16<PT>(lds, l, lane, src, dst);
      else layer_product<PT>(lds + l * DIM * DIM, lane, src, dst);
      if (l < LAYERS - 1) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          unsigned bits = 0u;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool on = dst[pt][mt][r] > 0.f;
              bits |= on ? (1u << (mt * 16 + r)) : 0u;
              dst[pt][mt][r] = on ? dst[pt][mt][r] : 0.f;
            }
          if (masks) masks[(((grp * PT + pt) * 64 + lane) * 3 + l)] = bits;
        }
        // training: the hidden activations are kept for the weight gradients (pr_mlp_bwd_full), [3][P][64]
        if (acts) store_rows<PT>(acts + (int64_t)l * P * DIM, first, P, lane, dst);
      }
    }
    // LAYERS is even: the result is back in `a`
    if (zero_below > 0.f) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        if (al[pt] < zero_below)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[pt][mt][r] = 0.f;
    }
    store_rows<PT>(out, first, P, lane, a);
  }
}

template <int PT, int WPE, int BLOCK, bool HOIST, bool BF16>
__global__ void __launch_bounds__(BLOCK, WPE) mlp_bwd_input_kernel(int64_t P, const float *__restrict__ v_out,
                                                              const float *__restrict__ alphas, const Weights W,
                                                              float zero_below,
                                                              const unsigned *__restrict__ masks,
                                                              float *__restrict__ v_feats) {
  extern __shared__ float lds[];
  if (BF16) stage_weights_bf16<true, BLOCK>(lds, W);
  else stage_weights<true, BLOCK>(lds, W);
  constexpr int PIX_PER_WAVE = 32 * PT;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_groups = (P + PIX_PER_WAVE - 1) / PIX_PER_WAVE;
  for (int64_t grp = (int64_t)blockIdx.x * (BLOCK / 64) + wave; grp < n_groups;
       grp += (int64_t)gridDim.x * (BLOCK / 64)) {
    const int64_t first = grp * PIX_PER_WAVE;
    if (!HOIST) asm volatile("" ::: "memory");  // re-read the weights from LDS per group
    f16v a[PT][2], b[PT][2];
    load_rows<PT>(v_out, first, P, lane, a);
    float al[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) al[pt] = 1.f;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int64_t pix = first + pt * 32 + (lane & 31);
      if (alphas) al[pt] = pix < P ? alphas[pix] : 0.f;
      if (zero_below > 0.f && al[pt] < zero_below)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) a[pt][mt][r] = 0.f;
    }
#pragma unroll
    for (int l = LAYERS - 1; l >= 0; --l) {
      f16v(&src)[PT][2] = ((LAYERS - 1 - l) & 1) ? b : a;
      f16v(&dst)[PT][2] = ((LAYERS - 1 - l) & 1) ? a : b;
      if (BF16) layer_product_bf16<PT>(lds, l, lane, src, dst);
      else layer_product<PT>(lds + l * DIM * DIM, lane, src, dst);
      if (l > 0) {  // through the ReLU of layer l - 1
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const unsigned bits = masks[((grp * PT + pt) * 64 + lane) * 3 + (l - 1)];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[pt][mt][r] = ((bits >> (mt * 16 + r)) & 1u) ? dst[pt][mt][r] : 0.f;
        }
      }
    }
    // LAYERS is even: the result is back in `a`; through the normalisation (alpha itself gets no gradient)
    if (alphas) {
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const float d = fmaxf(al[pt], 1e-6f), inv = al[pt] > 0.f ? 1.f / d : 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)  // float32: true division, as torch; split-bf16: one reciprocal (1 ulp)
            a[pt][kt][r] = BF16 ? a[pt][kt][r] * inv : (al[pt] > 0.f ? a[pt][kt][r] / d : 0.f);
      }
    }
    store_rows<PT>(v_feats, first, P, lane, a);
  }
}

// ---- backward with weight gradients (training) ----------------------------------------------------------------
// dW_l[out][in] = sum over pixels of G_l[out][p] * H_l[in][p]  (H_l: input of layer l, G_l: gradient at its output)
// reduces over the PIXEL index, which sits on the lanes in the transposed register layout above.  Both factors go
// through a per-wavefront LDS image [feature][pixel] (bf16 head and residual), from which a lane (feature m, pixel
// half h) reads its eight consecutive pixels as one 16-byte matrix operand: A = G_l rows, B = H_l rows, 2 x 2 result
// tiles x 2 k steps x 3 split products per layer and 32-pixel tile.  The four 64 x 64 gradients stay in 256
// accumulator registers for the whole persistent loop (one wavefront per SIMD) and leave as float atomics at the
// end.  The chain G_{l-1} = (W_l^T G_l) * [H_l > 0] is the frozen-weight kernel's; H_l comes from the activations the
// training forward pass stored (H_0 = the normalised input).  Split bf16 throughout (2^-16 per product, gradients are
// held to 1e-3).
constexpr int IMG_PITCH = 40;   // 16-bit units per feature row: 32 pixels + 8 of padding (rows 80 bytes apart)
struct WaveImages {
  unsigned short gh[DIM * IMG_PITCH], gl[DIM * IMG_PITCH], hh[DIM * IMG_PITCH], hl[DIM * IMG_PITCH];
};

__device__ __forceinline__ void write_image(unsigned short *hi, unsigned short *lo, const f16v (&t)[1][2], int lane) {
  const int p = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      unsigned xh, xl;
      split_pair(t[0][mt][r], t[0][mt][r + 1], xh, xl);
      const int f = 32 * mt + 8 * (r >> 2) + 4 * h + (r & 3);
      hi[f * IMG_PITCH + p] = (unsigned short)xh;
      hi[(f + 1) * IMG_PITCH + p] = (unsigned short)(xh >> 16);
      lo[f * IMG_PITCH + p] = (unsigned short)xl;
      lo[(f + 1) * IMG_PITCH + p] = (unsigned short)(xl >> 16);
    }
}

__global__ void __launch_bounds__(256, 1) mlp_bwd_full_kernel(int64_t P, const float *__restrict__ v_out,
                                                             const float *__restrict__ feats,
                                                             const float *__restrict__ alphas,
                                                             const float *__restrict__ acts, const Weights W,
                                                             float zero_below, float *__restrict__ v_feats,
                                                             float *__restrict__ v_w) {
  extern __shared__ float lds[];
  stage_weights_bf16<true, 256>(lds, W);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveImages &I = reinterpret_cast<WaveImages *>(lds + LAYERS * DIM * DIM)[wave];
  __syncthreads();
  const int m = lane & 31, kh = lane >> 5;
  f16v dw[LAYERS][2][2];
#pragma unroll
  for (int l = 0; l < LAYERS; ++l)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[l][a][b][r] = 0.f;
  const int64_t n_groups = (P + 31) / 32;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + wave; grp < n_groups; grp += (int64_t)gridDim.x * 4) {
    const int64_t first = grp * 32;
    asm volatile("" ::: "memory");
    f16v g[1][2], h[1][2], t[1][2];
    load_rows<1>(v_out, first, P, lane, g);
    const int64_t pix = first + m;
    float al = 1.f;
    if (alphas) al = pix < P ? alphas[pix] : 0.f;
    const float inv = alphas ? (al > 0.f ? 1.f / fmaxf(al, 1e-6f) : 0.f) : 1.f;
    if (zero_below > 0.f && al < zero_below)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[0][mt][r] = 0.f;
#pragma unroll
    for (int l = LAYERS - 1; l >= 0; --l) {
      if (l > 0) {
        load_rows<1>(acts + (int64_t)(l - 1) * P * DIM, first, P, lane, h);
      } else {
        load_rows<1>(feats, first, P, lane, h);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[0][mt][r] *= inv;
      }
      write_image(I.gh, I.gl, g, lane);
      write_image(I.hh, I.hl, h, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf8v ah[2], al2[2], bh[2], bl[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int at = (32 * x + m) * IMG_PITCH + 16 * ks + 8 * kh;
          ah[x] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&I.gh[at]));
          al2[x] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&I.gl[at]));
          bh[x] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&I.hh[at]));
          bl[x] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v *>(&I.hl[at]));
        }
#pragma unroll
        for (int mo = 0; mo < 2; ++mo)
#pragma unroll
          for (int no = 0; no < 2; ++no) {
            dw[l][mo][no] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mo], bh[no], dw[l][mo][no], 0, 0, 0);
            dw[l][mo][no] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al2[mo], bh[no], dw[l][mo][no], 0, 0, 0);
            dw[l][mo][no] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mo], bl[no], dw[l][mo][no], 0, 0, 0);
          }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the images are rewritten by the next layer
      __builtin_amdgcn_wave_barrier();
      if (l > 0 || v_feats) {
        layer_product_bf16<1>(lds, l, lane, g, t);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[0][mt][r] = (l == 0 || h[0][mt][r] > 0.f) ? t[0][mt][r] : 0.f;
      }
    }
    if (v_feats) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[0][mt][r] *= inv;
      store_rows<1>(v_feats, first, P, lane, g);
    }
  }
  // result tile (mo, no): row = out feature 32 mo + (r&3) + 8 (r>>2) + 4 kh, column = in feature 32 no + m
  settle_mfma_results();
#pragma unroll
  for (int l = 0; l < LAYERS; ++l)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int no = 0; no < 2; ++no)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int out = 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * kh;
          atomic_add_noret(v_w + (l * DIM + out) * DIM + 32 * no + m, dw[l][mo][no][r]);
        }
}

struct Variant {
  int pt, wpe, block;
  bool hoist, bf16;
};
// POGS_MLP_VARIANT selects a kernel shape and arithmetic (read once, see Tuning in common.h; for measurements and tests):
// 0-3 exact float32 (v_mfma_f32_32x32x2_f32), 4-6 split bf16 (3 x v_mfma_f32_32x32x16_bf16 per product).
constexpr int N_VARIANTS = 7, DEFAULT_VARIANT = 4;
Variant variant() {
  // measured on a 1080p image, forward: 0.725 / 0.688 / 0.701 / 0.802 ms (float32)
  static const Variant table[N_VARIANTS] = {{1, 2, 256, false, false}, {1, 4, 512, false, false},
                                            {2, 2, 256, false, false}, {2, 1, 256, true, false},
                                            {1, 4, 512, false, true},  {2, 2, 256, false, true},
                                            {1, 2, 256, false, true}};
  const int v = tuning().mlp_variant;
  return table[v >= 0 && v < N_VARIANTS ? v : DEFAULT_VARIANT];
}

int grid_for(int64_t P, const Variant &v) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  const int64_t per_block = (int64_t)(v.block / 64) * 32 * v.pt;
  const int64_t blocks_needed = (P + per_block - 1) / per_block;
  const int64_t cap = (int64_t)cus * 2;  // 64 KB of LDS each: two workgroups per CU
  return (int)(blocks_needed < cap ? blocks_needed : cap);
}

constexpr size_t LDS_BYTES = (size_t)LAYERS * DIM * DIM * sizeof(float);

template <typename K, typename... Args>
int launch(K kernel, const char *what, int64_t P, const Variant &v, hipStream_t stream, Args... args) {
  PR_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  hipLaunchKernelGGL(kernel, dim3(grid_for(P, v)), dim3(v.block), LDS_BYTES, stream, P, args...);
  return check_launch(what);
}

}  // namespace
}  // namespace pr

extern "C" size_t pr_mlp_mask_words(int64_t n_pixels) {
  if (n_pixels <= 0) return 0;
  return (size_t)((n_pixels + 31) / 32) * 64 * 3 / 2;  // 32 bits per (tile, lane, layer)
}

static int mlp_forward(int64_t n_pixels, const float *feats, const float *alphas, const float *w0, const float *w1,
                       const float *w2, const float *w3, float zero_below, float *out, uint64_t *masks, float *acts,
                       void *stream) {
  if (n_pixels < 0) return PR_EINVAL;
  if (n_pixels == 0) return PR_OK;
  if (!feats || !w0 || !w1 || !w2 || !w3 || !out) return PR_EINVAL;
  if ((((uintptr_t)feats | (uintptr_t)out | (uintptr_t)acts) & 15) != 0) return PR_EINVAL;
  const pr::Weights W = {w0, w1, w2, w3};
  const pr::Variant v = pr::variant();
  hipStream_t st = (hipStream_t)stream;
  unsigned *m = (unsigned *)masks;
#define PR_MLP_CASE(PT, WPE, BLK, HOIST, BF)                                                                    \
  if (v.pt == PT && v.wpe == WPE && v.block == BLK && v.hoist == HOIST && v.bf16 == BF)                         \
    return pr::launch(pr::mlp_fwd_kernel<PT, WPE, BLK, HOIST, BF>, "pr_mlp_fwd", n_pixels, v, st, feats, alphas, W, \
                      zero_below, out, m, acts);
  PR_MLP_CASE(1, 2, 256, false, false)
  PR_MLP_CASE(1, 4, 512, false, false)
  PR_MLP_CASE(2, 2, 256, false, false)
  PR_MLP_CASE(2, 1, 256, true, false)
  PR_MLP_CASE(1, 4, 512, false, true)
  PR_MLP_CASE(2, 2, 256, false, true)
  PR_MLP_CASE(1, 2, 256, false, true)
#undef PR_MLP_CASE
  return PR_EUNSUPPORTED;
}

extern "C" int pr_mlp_fwd(int64_t n_pixels, const float *feats, const float *alphas, const float *w0, const float *w1,
                          const float *w2, const float *w3, float zero_below, float *out, uint64_t *masks,
                          void *stream) {
  return mlp_forward(n_pixels, feats, alphas, w0, w1, w2, w3, zero_below, out, masks, nullptr, stream);
}

extern "C" int pr_mlp_fwd_train(int64_t n_pixels, const float *feats, const float *alphas, const float *w0,
                                const float *w1, const float *w2, const float *w3, float zero_below, float *out,
                                float *acts, void *stream) {
  if (n_pixels > 0 && !acts) return PR_EINVAL;
  return mlp_forward(n_pixels, feats, alphas, w0, w1, w2, w3, zero_below, out, nullptr, acts, stream);
}

extern "C" int pr_mlp_bwd_full(int64_t n_pixels, const float *v_out, const float *feats, const float *alphas,
                               const float *acts, const float *w0, const float *w1, const float *w2, const float *w3,
                               float zero_below, float *v_feats, float *v_weights, void *stream) {
  if (n_pixels < 0) return PR_EINVAL;
  if (n_pixels == 0) return PR_OK;
  if (!v_out || !feats || !acts || !w0 || !w1 || !w2 || !w3 || !v_weights) return PR_EINVAL;
  if ((((uintptr_t)v_out | (uintptr_t)feats | (uintptr_t)acts | (uintptr_t)v_feats) & 15) != 0) return PR_EINVAL;
  const pr::Weights W = {w0, w1, w2, w3};
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  const int64_t groups = (n_pixels + 31) / 32, blocks = (groups + 3) / 4;
  const size_t lds = pr::LDS_BYTES + 4 * sizeof(pr::WaveImages);
  PR_HIP(hipFuncSetAttribute((const void *)pr::mlp_bwd_full_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds));
  hipLaunchKernelGGL(pr::mlp_bwd_full_kernel, dim3((unsigned)(blocks < cus ? blocks : cus)), dim3(256), lds,
                     (hipStream_t)stream, n_pixels, v_out, feats, alphas, acts, W, zero_below, v_feats, v_weights);
  return pr::check_launch("pr_mlp_bwd_full");
}

extern "C" int pr_mlp_bwd_input(int64_t n_pixels, const float *v_out, const float *alphas, const float *w0,
                                const float *w1, const float *w2, const float *w3, float zero_below,
                                const uint64_t *masks, float *v_feats, void *stream) {
  if (n_pixels < 0) return PR_EINVAL;
  if (n_pixels == 0) return PR_OK;
  if (!v_out || !w0 || !w1 || !w2 || !w3 || !masks || !v_feats) return PR_EINVAL;
  if ((((uintptr_t)v_out | (uintptr_t)v_feats) & 15) != 0) return PR_EINVAL;
  const pr::Weights W = {w0, w1, w2, w3};
  const pr::Variant v = pr::variant();
  hipStream_t st = (hipStream_t)stream;
  const unsigned *m = (const unsigned *)masks;
#define PR_MLP_CASE(PT, WPE, BLK, HOIST, BF)                                                                   \
  if (v.pt == PT && v.wpe == WPE && v.block == BLK && v.hoist == HOIST && v.bf16 == BF)                        \
    return pr::launch(pr::mlp_bwd_input_kernel<PT, WPE, BLK, HOIST, BF>, "pr_mlp_bwd_input", n_pixels, v, st, v_out, \
                      alphas, W, zero_below, m, v_feats);
  PR_MLP_CASE(1, 2, 256, false, false)
  PR_MLP_CASE(1, 4, 512, false, false)
  PR_MLP_CASE(2, 2, 256, false, false)
  PR_MLP_CASE(2, 1, 256, true, false)
  PR_MLP_CASE(1, 4, 512, false, true)
  PR_MLP_CASE(2, 2, 256, false, true)
  PR_MLP_CASE(1, 2, 256, false, true)
#undef PR_MLP_CASE
  return PR_EUNSUPPORTED;
}
