// Backward compositor for gfx950, geometry + colour gradients at 2 pixels per lane with the colour-gradient
// contraction deferred to the matrix cores (see the comment above the kernel).  Selected by pr_raster_bwd.
#include "raster_common.h"

namespace pr {

// ---- geometry + colour gradients at 2 pixels per lane, colour-gradient contraction on the matrix cores --------------
// The vector kernel spends most of a contributing (wavefront, Gaussian) iteration on the colour gradient
//   vC[c] = sum over the region's 128 pixels of fac[p] * v[p][c]:
// D/2 packed FMAs, D/4 lane-swap folds, an LDS round trip and a read-back per Gaussian.  Here that contraction is
// deferred: an iteration only stores its two fac values (bf16 head + residual, one 32-bit store each) into the next
// free ROW of a 16 x 128 LDS image; every 16 contributing iterations -- whichever list slots they came from, so the
// rows are always full -- the image is multiplied with the region's cotangents on v_mfma_f32_16x16x32_bf16:
//   vC[16 rows][32 channels] = fac[16 x 128 pixels] . v[128 pixels x 32 channels]   (hi.hi + lo.hi + hi.lo)
// v is constant for the region: it is transposed through LDS once, in the prologue, into B-operand registers (lane =
// channel, registers = pixels; pixel index k = 2 * lane + slot, so that a lane's two fac values are neighbours in a
// row).  The product's result layout (lane & 15 = channel, 4 rows per lane) leaves as float atomics on 64-byte row
// segments.  The dot product S = <colour_g, v_p> and the geometry stay on the vector unit as in raster_bwd_kernel;
// channels past the first 32 (the expected-depth channel among them) keep the fold path next to the geometry values.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v mfma16_bf16(u4v a, const unsigned (&b)[4], f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a),
                                                 __builtin_bit_cast(bf8v, u4v{b[0], b[1], b[2], b[3]}), c, 0, 0, 0);
}

template <int CD>
__global__ void __launch_bounds__(64, 2) raster_bwd_cmfma_kernel(
    int n_regions, int N, int D, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int width,
    int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
    const float *__restrict__ render_colors, int ed_channel, const float *__restrict__ v_render_colors,
    const float *__restrict__ v_render_alphas, int absgrad, float *__restrict__ v_geom, const GradSegs gs) {
  constexpr int PPL = 2;
  using Rg = Region<16, PPL>;
  constexpr int ROW = color_row(CD), CDP = round4(CD), NQ = CDP / 4, CH2 = CDP / 2;
  constexpr int CC = 32, CR = CDP - CC, NQR = CR / 4;  // channels on the matrix cores / on the fold path
  constexpr int NV = CR + 8;                           // values folded per Gaussian: rest colours, then geometry
  // list entries staged per batch: 1.15 ms (32) -> 1.085 (48) -> 1.067 (56) at the 1M-Gaussian 1080p 36-channel shape;
  // 64 would leave room for seven wavefronts per CU only
  constexpr int BS = 56;
  constexpr int RW = 16;                               // rows (contributing iterations) of the fac image
  constexpr int NPX = 64 * PPL;                        // pixels of the region = columns of the image
  constexpr int FP = NPX + 8;                          // its row pitch in 16-bit units
  constexpr int KS = NPX / 32;                         // k steps of the deferred product
  static_assert(CDP >= CC && CR < 32 && NV <= 64, "pass width");
  __shared__ WaveLds<CD, false, BS> L;
  __shared__ float red[(2 + NQR) * 68];
  union FacImage {
    struct {
      unsigned short hi[RW * FP], lo[RW * FP];  // pixel k = PPL * lane + slot: a lane's entries are neighbours
    } f;
    float tr[NPX * 17];  // prologue: 16 channels of the region's cotangents, for the transpose
  };
  __shared__ __attribute__((aligned(16))) FacImage F;
  static_assert(sizeof(F.f) >= sizeof(F.tr), "transpose area");
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * 16 + (lane % Rg::LX);
  const int y0 = ty * 16 + sub * Rg::RH + (lane / Rg::LX);
  const int range_lo = isect_offsets[tile_flat];
  const int nch = min(CD, D);
  const int l15 = lane & 15, lg = lane >> 4;

  float T[PPL], Bsum[PPL], W[PPL];
  f2 vout[PPL][CH2];
  int last[PPL];
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;
  int max_last = range_lo;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int x = x0, y = y0 + j * Rg::SY;
    const bool inside = x < width && y < height;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    T[j] = 1.f;
    Bsum[j] = 0.f;
    W[j] = 0.f;
    last[j] = range_lo;
#pragma unroll
    for (int k = 0; k < CH2; ++k) vout[j][k] = splat(0.f);
    if (inside) {
      T[j] = 1.f - render_alphas[pix];
      last[j] = last_ids[pix];
      const float *src = v_render_colors + pix * D;
      if ((D & 3) == 0) {
#pragma unroll
        for (int k = 0; k + 3 < CDP; k += 4)
          if (k < nch) {
            const float4 v = *reinterpret_cast<const float4 *>(src + k);
            vout[j][k / 2] = f2{v.x, v.y};
            vout[j][k / 2 + 1] = f2{v.z, v.w};
          }
      } else {
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) vout[j][k / 2][k % 2] = src[k];
      }
      float w_ed = 0.f;
      if (ed_channel >= 0) {  // expected depth out_d = X / max(a, 1e-10): see raster_bwd_kernel
        const float a = 1.f - T[j], inv = 1.f / fmaxf(a, 1e-10f);
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k == ed_channel) {
            if (a > 1e-10f) w_ed = -vout[j][k / 2][k % 2] * render_colors[pix * D + ed_channel] * inv;
            vout[j][k / 2][k % 2] *= inv;
          }
      }
      float w = v_render_alphas[pix] + w_ed;
      if (backgrounds) {
        const float *bg = backgrounds + (int64_t)cam * D;
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) w -= bg[k] * vout[j][k / 2][k % 2];
      }
      W[j] = T[j] * w;
    }
    max_last = max(max_last, last[j]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));

  // B operand of the deferred product: N tile n = channels 16n + (lane & 15); k step s, register i = pixels
  // k = 32s + 8(lane >> 4) + 2i and k + 1, i.e. both pixel slots of lane 16s + 4(lane >> 4) + i
  unsigned vbh[2][KS][4], vbl[2][KS][4];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int j = 0; j < PPL; ++j)
#pragma unroll
      for (int c = 0; c < 16; ++c) F.tr[(PPL * lane + j) * 17 + c] = vout[j][(16 * n + c) / 2][c & 1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 32 * s + 8 * lg + 2 * i;
        split_pair(F.tr[k * 17 + l15], F.tr[(k + 1) * 17 + l15], vbh[n][s][i], vbl[n][s][i]);
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  // fold path: lane v < NV owns value v (rest colours first, then the 8 geometry values)
  const int own_q = lane >> 2, own_row = ((lane & 1) << 1) | ((lane >> 1) & 1);
  const float4 *own = reinterpret_cast<const float4 *>(red + own_q * 68 + own_row * 16);
  const int gi = lane - CR;
  const float gscale = (gi == 0 || gi == 1) ? -LN2 : (gi == 2 || gi == 4) ? 0.5f : (gi == 6 || gi == 7) ? LN2 : 1.f;
  const int n_geo = absgrad ? 8 : 6;
  float *rcol = nullptr;
  int rld = 0;
  bool rshared = false;
  if (CR > 0 && lane < CR) {
    const int ch = CC + lane;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    rcol = (p && ch < nch) ? p + (ch - gs.begin[sgm]) : nullptr;
    rld = gs.ld[sgm];
    rshared = gs.shared[sgm] != 0;
  }
  // matrix path: this lane's column of N tile n
  float *mcol[2];
  int mld[2];
  bool mshared[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int ch = 16 * n + l15;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    mcol[n] = (p && ch < nch) ? p + (ch - gs.begin[sgm]) : nullptr;
    mld[n] = gs.ld[sgm];
    mshared[n] = gs.shared[sgm] != 0;
  }
  const int cam_first = cam * N;

  int nrows = 0;    // rows of the fac image in use (wave-uniform)
  int row_gid = 0;  // lane r < 16: flat Gaussian index of row r
  // vC of the image's first `count` rows -> atomics
  auto flush = [&](int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    f4v acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[n] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u4v ah = *reinterpret_cast<const u4v *>(&F.f.hi[l15 * FP + 32 * s + 8 * lg]);
      const u4v al = *reinterpret_cast<const u4v *>(&F.f.lo[l15 * FP + 32 * s + 8 * lg]);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        acc[n] = mfma16_bf16(ah, vbh[n][s], acc[n]);
        acc[n] = mfma16_bf16(al, vbh[n][s], acc[n]);
        acc[n] = mfma16_bf16(ah, vbl[n][s], acc[n]);
      }
    }
    settle_mfma_results();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = __shfl(row_gid, 4 * lg + r);
      if (4 * lg + r < count) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
          if (mcol[n]) atomic_add_noret(mcol[n] + (int64_t)(mshared[n] ? g - cam_first : g) * mld[n], acc[n][r]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  const float rx0 = (float)(tx * 16) + 0.5f, ry0 = (float)(ty * 16 + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * 16 + 16, width) - 0.5f;
  const float ry1 = (float)min(ty * 16 + sub * Rg::RH + Rg::RH, height) - 0.5f;
  BatchGeometry pre = prefetch_batch(lane, max(range_lo, max_last - BS), max_last - max(range_lo, max_last - BS),
                                     flatten_ids, means2d, conics, opacities);
  for (int bend = max_last; bend > range_lo; bend -= BS) {
    const int first = max(range_lo, bend - BS);
    unsigned long long todo = stage_batch<CD, false, BS>(L, lane, pre, cs, N, D, 0, rx0, ry0, rx1, ry1);
    {  // the next batch's geometry travels while this one is composited
      const int nend = bend - BS, nfirst = max(range_lo, nend - BS);
      pre = prefetch_batch(lane, nfirst, nend - nfirst, flatten_ids, means2d, conics, opacities);
    }
    while (todo) {
      const int t = 63 - __builtin_clzll(todo);
      todo &= ~(1ull << t);
      const int idx = first + t;
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      const float dx = g.x - px0;
      const float adx2 = g.w * dx * dx, bdx = h.x * dx;
      float dy[PPL], e[PPL], ar[PPL], alpha[PPL];
      unsigned long long any = 0ull;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        dy[j] = g.y - (py0 + (float)(j * Rg::SY));
        const float power = adx2 + h.y * dy[j] * dy[j] + bdx * dy[j];
        e[j] = __builtin_amdgcn_exp2f(power);
        ar[j] = g.z * e[j];
        const float a = fminf(ALPHA_MAX, ar[j]);
        const bool active = (power <= 0.f) && (a >= ALPHA_MIN) && (idx < last[j]);
        alpha[j] = active ? a : 0.f;
        any |= __builtin_amdgcn_ballot_w64(active);
      }
      if (any == 0ull) continue;

      const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW);
      const int gid = L.ids[t];
      float fac[PPL], ra[PPL];
      f2 S2[PPL];
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        ra[j] = __builtin_amdgcn_rcpf(1.f - alpha[j]);
        T[j] *= ra[j];
        fac[j] = alpha[j] * T[j];
        S2[j] = splat(0.f);
      }
      {  // this lane's two entries of the image row
        unsigned fh, fl;
        split_pair(fac[0], fac[1], fh, fl);
        *reinterpret_cast<unsigned *>(&F.f.hi[nrows * FP + 2 * lane]) = fh;
        *reinterpret_cast<unsigned *>(&F.f.lo[nrows * FP + 2 * lane]) = fl;
        row_gid = lane == nrows ? gid : row_gid;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 c4 = crow[q];
        const f2 c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w};
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          S2[j] = fma2(c01, vout[j][2 * q], S2[j]);
          S2[j] = fma2(c23, vout[j][2 * q + 1], S2[j]);
        }
      }
#pragma unroll
      for (int q = 0; q < NQR; ++q) {  // rest colours: fac * v, folded like the geometry
        f2 a01 = splat(0.f), a23 = splat(0.f);
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          const f2 f = splat(fac[j]);
          a01 = fma2(f, vout[j][CC / 2 + 2 * q], a01);
          a23 = fma2(f, vout[j][CC / 2 + 2 * q + 1], a23);
        }
        red[q * 68 + lane] = quad_fold_rows(a01[0], a01[1], a23[0], a23[1]);
      }
      {
        float geo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // x y a b c o |x| |y|
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          const float S = S2[j][0] + S2[j][1];
          const float v_alpha = S * T[j] - (Bsum[j] - W[j]) * ra[j];
          Bsum[j] = fmaf(fac[j], S, Bsum[j]);
          const float ev = (alpha[j] == ar[j] ? e[j] : 0.f) * v_alpha;
          geo[5] += ev;
          const float vs = -g.z * ev;
          const float t1 = vs * dx, t2 = vs * dy[j];
          geo[2] = fmaf(t1, dx, geo[2]);
          geo[3] = fmaf(t1, dy[j], geo[3]);
          geo[4] = fmaf(t2, dy[j], geo[4]);
          const float gx = 2.f * g.w * dx + h.x * dy[j];
          const float gy = h.x * dx + 2.f * h.y * dy[j];
          const float vx = vs * gx, vy = vs * gy;
          geo[0] += vx;
          geo[1] += vy;
          geo[6] += fabsf(vx);
          geo[7] += fabsf(vy);
        }
        red[NQR * 68 + lane] = quad_fold_rows(geo[0], geo[1], geo[2], geo[3]);
        red[(NQR + 1) * 68 + lane] = quad_fold_rows(geo[4], geo[5], geo[6], geo[7]);
      }
      if (lane < NV) {
        const float4 p0 = own[0], p1 = own[1], p2 = own[2], p3 = own[3];
        const f2 s0 = f2{p0.x, p0.y} + f2{p0.z, p0.w}, s1 = f2{p1.x, p1.y} + f2{p1.z, p1.w};
        const f2 s2 = f2{p2.x, p2.y} + f2{p2.z, p2.w}, s3 = f2{p3.x, p3.y} + f2{p3.z, p3.w};
        const f2 u = (s0 + s1) + (s2 + s3);
        const float v = u[0] + u[1];
        if (lane < CR) {
          if (rcol) atomic_add_noret(rcol + (int64_t)(rshared ? gid - cam_first : gid) * rld, v);
        } else if (gi < n_geo) {
          atomic_add_noret(v_geom + (int64_t)gid * gs.geom_ld + gi, v * gscale);
        }
      }
      if (++nrows == RW) {
        flush(RW);
        nrows = 0;
      }
    }
  }
  if (nrows) flush(nrows);
}

template <int CD>
static int launch_bwd_cmfma_t(int C, int N, int D, const float *means2d, const float *conics, const ColorSegs &cs,
                            const float *opacities, const float *backgrounds, int width, int height, int tile_w,
                            int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                            const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas,
                            const int32_t *last_ids, const float *render_colors, int ed_channel,
                            const float *v_render_colors, const float *v_render_alphas, int absgrad, float *v_geom,
                            const GradSegs &gs, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<16, 2>::WPT;
  const unsigned grid = region_grid(n_regions, Region<16, 2>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_bwd_cmfma_kernel<CD>), dim3(grid), dim3(64), 0, stream, (int)n_regions, N, D, means2d,
                     conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order,
                     flatten_ids, (int)n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors,
                     v_render_alphas, absgrad, v_geom, gs);
  return check_launch("raster_bwd_cmfma_kernel");
}

// ---- colour gradients alone, contraction on the matrix cores ----------------------------------------------------------
// vC[g][c] = sum_p w[p][g] * v[p][c] with w = alpha * T the compositing weight needs neither the colours nor the dot
// product S: the kernel replays the FORWARD walk (front to back, T from 1, stopping where the forward stopped, which
// last_ids records) on the 32-byte records alone -- no colour rows are staged -- and defers the contraction exactly
// as raster_bwd_cmfma_kernel does, every channel on the matrix cores (N tiles of 16 channels).  pr_raster_bwd pairs
// it with the geometry-only vector kernel: two launches whose wavefronts each fit more often per SIMD than the one
// kernel that keeps cotangents, operand and geometry state all at once.
template <int NT, int BS>
struct alignas(16) ColourLds {
  float4 g4[BS];
  float2 g2[BS];
  int32_t ids[BS];
};

template <int NT>
__global__ void __launch_bounds__(64, 3) raster_bwd_colour_mfma_kernel(
    int n_regions, int N, int D, const float *__restrict__ means2d, const float *__restrict__ conics,
    const float *__restrict__ opacities, int width, int height, int tile_w, int tile_h,
    const int32_t *__restrict__ isect_offsets, const int32_t *__restrict__ tile_order,
    const int32_t *__restrict__ flatten_ids, const float *__restrict__ render_alphas,
    const int32_t *__restrict__ last_ids, int ed_channel, const float *__restrict__ v_render_colors,
    const GradSegs gs) {
  constexpr int PPL = 2;
  using Rg = Region<16, PPL>;
  constexpr int BS = 64;
  constexpr int RW = 16, NPX = 64 * PPL, FP = NPX + 8, KS = NPX / 32;
  __shared__ ColourLds<NT, BS> L;
  union FacImage {
    struct {
      unsigned short hi[RW * FP], lo[RW * FP];
    } f;
    float tr[NPX * 17];
  };
  __shared__ __attribute__((aligned(16))) FacImage F;
  static_assert(sizeof(F.f) >= sizeof(F.tr), "transpose area");
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * 16 + (lane % Rg::LX);
  const int y0 = ty * 16 + sub * Rg::RH + (lane / Rg::LX);
  const int range_lo = isect_offsets[tile_flat];
  const int l15 = lane & 15, lg = lane >> 4;
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;

  float T[PPL];
  int last[PPL];
  int64_t pixv[PPL];
  float vscale[PPL];  // expected depth out_d = X / max(a, 1e-10): the accumulated depth's cotangent is v_out_d / a
  int max_last = range_lo;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int x = x0, y = y0 + j * Rg::SY;
    const bool inside = x < width && y < height;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    T[j] = 1.f;
    last[j] = range_lo;
    pixv[j] = -1;
    vscale[j] = 1.f;
    if (inside) {
      last[j] = last_ids[pix];
      pixv[j] = pix;
      if (ed_channel >= 0) vscale[j] = 1.f / fmaxf(render_alphas[pix], 1e-10f);
    }
    max_last = max(max_last, last[j]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));
  if (max_last <= range_lo) return;

  // B operand: N tile n = channels 16n + (lane & 15); k step s, register i = pixels k = 32s + 8(lane >> 4) + 2i, k + 1
  // (pixel k = 2 * lane' + slot).  Built 16 channels at a time through LDS from the cotangent rows in memory.
  unsigned vbh[NT][KS][4], vbl[NT][KS][4];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
      float v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = 0.f;
      if (pixv[j] >= 0) {
        const float *src = v_render_colors + pixv[j] * D + 16 * n;
        if ((D & 3) == 0) {
#pragma unroll
          for (int c = 0; c < 16; c += 4)
            if (16 * n + c < D) {
              const float4 q = *reinterpret_cast<const float4 *>(src + c);
              v[c] = q.x, v[c + 1] = q.y, v[c + 2] = q.z, v[c + 3] = q.w;
            }
        } else {
#pragma unroll
          for (int c = 0; c < 16; ++c)
            if (16 * n + c < D) v[c] = src[c];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (16 * n + c == ed_channel) v[c] *= vscale[j];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) F.tr[(PPL * lane + j) * 17 + c] = v[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 32 * s + 8 * lg + 2 * i;
        split_pair(F.tr[k * 17 + l15], F.tr[(k + 1) * 17 + l15], vbh[n][s][i], vbl[n][s][i]);
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  float *mcol[NT];
  int mld[NT];
  bool mshared[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int ch = 16 * n + l15;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    mcol[n] = (p && ch < D) ? p + (ch - gs.begin[sgm]) : nullptr;
    mld[n] = gs.ld[sgm];
    mshared[n] = gs.shared[sgm] != 0;
  }
  const int cam_first = cam * N;

  int nrows = 0;    // rows of the image in use (wave-uniform)
  int row_gid = 0;  // lane r < 16: flat Gaussian index of row r
  auto flush = [&](int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    f4v acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {  // one k step's A operand at a time: few registers live next to the B operands
      const u4v ah = *reinterpret_cast<const u4v *>(&F.f.hi[l15 * FP + 32 * s + 8 * lg]);
      const u4v al = *reinterpret_cast<const u4v *>(&F.f.lo[l15 * FP + 32 * s + 8 * lg]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n] = mfma16_bf16(ah, vbh[n][s], acc[n]);
        acc[n] = mfma16_bf16(al, vbh[n][s], acc[n]);
        acc[n] = mfma16_bf16(ah, vbl[n][s], acc[n]);
      }
    }
    settle_mfma_results();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = __shfl(row_gid, 4 * lg + r);
      if (4 * lg + r < count) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (mcol[n]) atomic_add_noret(mcol[n] + (int64_t)(mshared[n] ? g - cam_first : g) * mld[n], acc[n][r]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  const float rx0 = (float)(tx * 16) + 0.5f, ry0 = (float)(ty * 16 + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * 16 + 16, width) - 0.5f;
  const float ry1 = (float)min(ty * 16 + sub * Rg::RH + Rg::RH, height) - 0.5f;
  BatchGeometry pre = prefetch_batch(lane, range_lo, min(BS, max_last - range_lo), flatten_ids, means2d, conics,
                                     opacities);
  for (int first = range_lo; first < max_last; first += BS) {
    bool keep = false;
    if (pre.g >= 0) {
      keep = touches_region(pre.xy.x, pre.xy.y, pre.o, pre.ca, pre.cb, pre.cc, rx0, ry0, rx1, ry1);
      if (keep) {
        L.g4[lane] = make_float4(pre.xy.x, pre.xy.y, pre.o, -0.5f * LOG2E * pre.ca);
        L.g2[lane] = make_float2(-LOG2E * pre.cb, -0.5f * LOG2E * pre.cc);
        L.ids[lane] = pre.g;
      }
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(keep);
    pre = prefetch_batch(lane, first + BS, min(BS, max_last - first - BS), flatten_ids, means2d, conics, opacities);
    while (todo) {
      const int t = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int idx = first + t;
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      const float dx = g.x - px0;
      const float adx2 = g.w * dx * dx, bdx = h.x * dx;
      float w[PPL];
      unsigned long long any = 0ull;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const float dy = g.y - (py0 + (float)(j * Rg::SY));
        const float power = adx2 + h.y * dy * dy + bdx * dy;
        const float a = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
        const bool active = (power <= 0.f) && (a >= ALPHA_MIN) && (idx < last[j]);
        w[j] = active ? a * T[j] : 0.f;
        T[j] -= w[j];
        any |= __builtin_amdgcn_ballot_w64(active);
      }
      if (any == 0ull) continue;
      unsigned fh, fl;
      split_pair(w[0], w[1], fh, fl);
      *reinterpret_cast<unsigned *>(&F.f.hi[nrows * FP + 2 * lane]) = fh;
      *reinterpret_cast<unsigned *>(&F.f.lo[nrows * FP + 2 * lane]) = fl;
      row_gid = lane == nrows ? L.ids[t] : row_gid;
      if (++nrows == RW) {
        flush(RW);
        nrows = 0;
      }
    }
  }
  if (nrows) flush(nrows);
}

int launch_bwd_colour_mfma(int C, int N, int D, const float *means2d, const float *conics, const float *opacities,
                           int width, int height, int tile_w, int tile_h, const int32_t *isect_offsets,
                           const int32_t *tile_order, const int32_t *flatten_ids, const float *render_alphas,
                           const int32_t *last_ids, int ed_channel, const float *v_render_colors, const GradSegs &gs,
                           hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<16, 2>::WPT;
  const unsigned grid = region_grid(n_regions, Region<16, 2>::WPT, tile_order != nullptr);
#define COLOUR_ARGS                                                                                              \
  dim3(grid), dim3(64), 0, stream, (int)n_regions, N, D, means2d, conics, opacities, width, height, tile_w, tile_h, \
      isect_offsets, tile_order, flatten_ids, render_alphas, last_ids, ed_channel, v_render_colors, gs
  switch ((D + 15) / 16) {
    case 1: hipLaunchKernelGGL((raster_bwd_colour_mfma_kernel<1>), COLOUR_ARGS); break;
    case 2: hipLaunchKernelGGL((raster_bwd_colour_mfma_kernel<2>), COLOUR_ARGS); break;
    case 3: hipLaunchKernelGGL((raster_bwd_colour_mfma_kernel<3>), COLOUR_ARGS); break;
    default: return PR_EUNSUPPORTED;
  }
#undef COLOUR_ARGS
  return check_launch("raster_bwd_colour_mfma_kernel");
}

int launch_bwd_cmfma(int cd, int C, int N, int D, const float *means2d, const float *conics, const ColorSegs &cs,
                     const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                     const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                     int64_t n_isects, const float *render_alphas, const int32_t *last_ids,
                     const float *render_colors, int ed_channel, const float *v_render_colors,
                     const float *v_render_alphas, int absgrad, float *v_geom, const GradSegs &gs,
                     hipStream_t stream) {
#define CMFMA_ARGS                                                                                                 \
  C, N, D, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order, \
      flatten_ids, n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas, \
      absgrad, v_geom, gs, stream
  switch (cd) {
    case 32: return launch_bwd_cmfma_t<32>(CMFMA_ARGS);
    case 36: return launch_bwd_cmfma_t<36>(CMFMA_ARGS);
    default: return PR_EUNSUPPORTED;
  }
#undef CMFMA_ARGS
}

}  // namespace pr
