// Alpha-normalise + 4 x (64 -> 64, no bias, ReLU between) channel MLP on the rendered feature image, fused,
// on the matrix cores (pr_mlp_fwd / pr_mlp_bwd_input).
//
// Replaces pogs/pogs.py:516-522 (normalise by the clamped alpha, `self.nn(...)`, optional zeroing where
// alpha < 0.8) with the module defined at pogs/pogs.py:153-161, and -- for callers that optimise something
// upstream of the render with the weights frozen, as the tracker does -- its gradient with respect to the
// rendered features.
//
// Arithmetic: exact float32.  v_mfma_f32_32x32x2_f32 is bit for bit an fmaf chain in k order, so the result
// differs from torch's sgemm only by summation order.  The activations stay in registers from the first layer
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
// wavefronts load, normalise and store.  Measured: 0.69 ms for a 1080p image, 99 TFLOP/s.
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
template <int PT, int WPE, int BLOCK, bool HOIST>
__global__ void __launch_bounds__(BLOCK, WPE) mlp_fwd_kernel(int64_t P, const float *__restrict__ feats,
                                                        const float *__restrict__ alphas, const Weights W,
                                                        float zero_below, float *__restrict__ out,
                                                        unsigned *__restrict__ masks) {
  extern __shared__ float lds[];
  stage_weights<false, BLOCK>(lds, W);
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
        const float d = fmaxf(al[pt], 1e-6f);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) a[pt][kt][r] = al[pt] > 0.f ? a[pt][kt][r] / d : 0.f;
      }
    }
#pragma unroll
    for (int l = 0; l < LAYERS; ++l) {
      f16v(&src)[PT][2] = (l & 1) ? b : a;
      f16v(&dst)[PT][2] = (l & 1) ? a : b;
      layer_product<PT>(lds + l * DIM * DIM, lane, src, dst);
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

template <int PT, int WPE, int BLOCK, bool HOIST>
__global__ void __launch_bounds__(BLOCK, WPE) mlp_bwd_input_kernel(int64_t P, const float *__restrict__ v_out,
                                                              const float *__restrict__ alphas, const Weights W,
                                                              float zero_below,
                                                              const unsigned *__restrict__ masks,
                                                              float *__restrict__ v_feats) {
  extern __shared__ float lds[];
  stage_weights<true, BLOCK>(lds, W);
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
      layer_product<PT>(lds + l * DIM * DIM, lane, src, dst);
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
        const float d = fmaxf(al[pt], 1e-6f);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) a[pt][kt][r] = al[pt] > 0.f ? a[pt][kt][r] / d : 0.f;
      }
    }
    store_rows<PT>(v_feats, first, P, lane, a);
  }
}

struct Variant {
  int pt, wpe, block;
  bool hoist;
};
// POGS_MLP_VARIANT = 0..3 selects a kernel shape (for measurements); the default is the fastest measured.
Variant variant() {
  // measured on a 1080p image, forward: 0.725 / 0.688 / 0.701 / 0.802 ms
  static const Variant table[] = {{1, 2, 256, false}, {1, 4, 512, false}, {2, 2, 256, false}, {2, 1, 256, true}};
  static const int chosen = [] {
    const char *e = getenv("POGS_MLP_VARIANT");
    const int v = e ? atoi(e) : 1;
    return v >= 0 && v < 4 ? v : 1;
  }();
  return table[chosen];
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

extern "C" int pr_mlp_fwd(int64_t n_pixels, const float *feats, const float *alphas, const float *w0, const float *w1,
                          const float *w2, const float *w3, float zero_below, float *out, uint64_t *masks,
                          void *stream) {
  if (n_pixels < 0) return PR_EINVAL;
  if (n_pixels == 0) return PR_OK;
  if (!feats || !w0 || !w1 || !w2 || !w3 || !out) return PR_EINVAL;
  if ((((uintptr_t)feats | (uintptr_t)out) & 15) != 0) return PR_EINVAL;
  const pr::Weights W = {w0, w1, w2, w3};
  const pr::Variant v = pr::variant();
  hipStream_t st = (hipStream_t)stream;
  unsigned *m = (unsigned *)masks;
#define PR_MLP_CASE(PT, WPE, BLK, HOIST)                                                                        \
  if (v.pt == PT && v.wpe == WPE && v.block == BLK && v.hoist == HOIST)                                         \
    return pr::launch(pr::mlp_fwd_kernel<PT, WPE, BLK, HOIST>, "pr_mlp_fwd", n_pixels, v, st, feats, alphas, W, \
                      zero_below, out, m);
  PR_MLP_CASE(1, 2, 256, false)
  PR_MLP_CASE(1, 4, 512, false)
  PR_MLP_CASE(2, 2, 256, false)
  PR_MLP_CASE(2, 1, 256, true)
#undef PR_MLP_CASE
  return PR_EUNSUPPORTED;
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
#define PR_MLP_CASE(PT, WPE, BLK, HOIST)                                                                       \
  if (v.pt == PT && v.wpe == WPE && v.block == BLK && v.hoist == HOIST)                                        \
    return pr::launch(pr::mlp_bwd_input_kernel<PT, WPE, BLK, HOIST>, "pr_mlp_bwd_input", n_pixels, v, st, v_out, \
                      alphas, W, zero_below, m, v_feats);
  PR_MLP_CASE(1, 2, 256, false)
  PR_MLP_CASE(1, 4, 512, false)
  PR_MLP_CASE(2, 2, 256, false)
  PR_MLP_CASE(2, 1, 256, true)
#undef PR_MLP_CASE
  return PR_EUNSUPPORTED;
}
