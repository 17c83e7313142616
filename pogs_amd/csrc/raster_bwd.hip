// Backward compositing kernel for gfx950: back-to-front replay of each tile's list, per-Gaussian
// totals reduced over the wavefront's pixels before ONE contiguous float-atomic per Gaussian.
//
// Algebra (per pixel, walking the list backwards; T = transmittance in front of the Gaussian):
//   ra = 1/(1-alpha);  T *= ra;  fac = alpha*T
//   S  = <colour_g, v_out_pixel>                      (one dot product per pair)
//   v_alpha = S*T - (B - W)*ra,   B = sum over Gaussians BEHIND g of fac*S,
//             W = T_final*(v_alpha_out - <background, v_out_pixel>)
//   v_colour_g += fac * v_out_pixel
// Keeping the scalar B instead of a per-channel "colour behind" buffer halves the per-pair channel
// work and removes D registers per pixel.
//
// Reduction over the wavefront's pixels: values are produced four at a time and folded across the four
// DPP rows by the lane-swap "transposed" reduction (quad_fold_rows in common.h): one register per four
// values, every lane of row r holding a 4-lane partial of value 4q+slot(r).  Each quad's register goes to
// the wave's LDS slice as one unpredicated store; lane v then reads back the 16 partials of value v as
// four 16-byte loads, adds them, and issues the atomic -- contiguous addresses per Gaussian.
#include <stdlib.h>

#include <type_traits>

#include "raster_common.h"

namespace pr {
#ifdef PR_DIAG
__device__ unsigned long long g_diag[8];
// PR_DIAG == 3: start / end time stamp, hardware id and walked list length of every wavefront (launch timeline)
__device__ unsigned long long g_wave_trace[4 * 65536];
#endif

// GEOM: gradients of the 2D means, conics and opacities are wanted (needs the dot product S);  CGRAD: gradients of
// the colours are wanted.  A caller that optimises poses through frozen appearance (the tracker) asks for GEOM only:
// the per-channel accumulation, its folds and its read-back -- most of the per-Gaussian work -- are compiled out.
template <int TS, int PPL, int CD, bool GEOM, bool CGRAD>
__global__ void __launch_bounds__(64 * WPB) raster_bwd_kernel(
    int n_regions, int N, int D, int ch_off, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds,
    int width, int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects, const float *__restrict__ render_alphas,
    const int32_t *__restrict__ last_ids, const float *__restrict__ render_colors, int ed_channel,
    const float *__restrict__ v_render_colors, const float *__restrict__ v_render_alphas, int absgrad,
    float *__restrict__ v_geom, const GradSegs gs) {
#ifdef PR_DIAG
  const unsigned long long tk_start = __builtin_amdgcn_s_memtime();
  unsigned long long dg_alpha = 0, dg_body = 0, dg_tail = 0, dg_iters = 0;
#endif
  using Rg = Region<TS, PPL>;
  constexpr int ROW = color_row(CD);
  constexpr int CDP = round4(CD);
  constexpr int NQ = CDP / 4;                 // colour quads
  constexpr int CDR = CGRAD ? CDP : 0;        // colour values reduced per Gaussian
  constexpr int NQR = CDR / 4;                // ... as quads in the reduction area
  constexpr int NV = CDR + (GEOM ? 8 : 0);    // values reduced per Gaussian
  static_assert(GEOM || CGRAD, "nothing to compute");
  // 1 pixel per lane: 16-entry batches keep the slice (colours + reduction area) under 10 KB = 4 wavefronts/SIMD
  // 2 pixels per lane, up to 36 channels: 56-entry batches -- the largest whose slice (13 KB) lets twelve wavefronts (the
  // register limit) share a CU's LDS: 1.165 (32 entries) -> 1.148 (64, eleven wavefronts) -> 1.12 ms; wider passes: 32
  constexpr int BS = PPL == 1 ? 16 : (PPL == 2 && CD <= 36) ? 56 : batch_size(PPL);
  __shared__ WaveLds<CD, true, BS> lds[WPB];
  static_assert(WPB == 1, "one wavefront per workgroup");
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  WaveLds<CD, true, BS> &L = lds[0];
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * TS + (lane % Rg::LX) * Rg::PX;
  const int y0 = ty * TS + sub * Rg::RH + (lane / Rg::LX) * Rg::PY;
  const int range_lo = isect_offsets[tile_flat];
  const int nch = min(CD, D - ch_off);

  constexpr int CH2 = CDP / 2;  // channel pairs
  float T[PPL], Bsum[PPL], W[PPL];
  f2 vout[PPL][CH2];
  int last[PPL];
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;
  int max_last = range_lo;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int x = x0 + (j % Rg::BW) * Rg::SX, y = y0 + (j / Rg::BW) * Rg::SY;
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
      const float *src = v_render_colors + pix * D + ch_off;
      if (((D | ch_off) & 3) == 0) {
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
      if (ed_channel >= 0) {
        // out_d = X / max(a, 1e-10): the cotangent of the accumulated depth X is v_out_d / a, and the
        // direct dependence on a adds -v_out_d * out_d / a to the alpha cotangent (none when clamped).
        const float a = 1.f - T[j], inv = 1.f / fmaxf(a, 1e-10f);
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k + ch_off == ed_channel) {
            if (a > 1e-10f) w_ed = -vout[j][k / 2][k % 2] * render_colors[pix * D + ed_channel] * inv;
            vout[j][k / 2][k % 2] *= inv;
          }
      }
      if (GEOM) {
        float w = v_render_alphas[pix] + w_ed;
        if (backgrounds) {
          const float *bg = backgrounds + (int64_t)cam * D + ch_off;
#pragma unroll
          for (int k = 0; k < CD; ++k)
            if (k < nch) w -= bg[k] * vout[j][k / 2][k % 2];
        }
        W[j] = T[j] * w;
      }
    }
    max_last = max(max_last, last[j]);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));

  // Which reduced value this lane hands to the atomics, and where it goes (set per Gaussian below).
  // Row r's last lane ends up with value 4*q + slot(r) of quad q, slot = {0,2,1,3}.
  // lane v < NV owns value v = 4*q + s: its 16 partials sit in row r(s) of quad q's 64-float line
  const int own_q = lane >> 2, own_row = ((lane & 1) << 1) | ((lane >> 1) & 1);  // slot s -> row {0,2,1,3}
  const float4 *own = reinterpret_cast<const float4 *>(L.red + own_q * 68 + own_row * 16);
  // post-scale of the geometry totals: x y (-ln2), conic a c (1/2), b, opacity (1), |x| |y| (ln2)
  float gscale = 1.f;
  if (GEOM) {
    const int gi = lane - CDR;
    gscale = (gi == 0 || gi == 1) ? -LN2 : (gi == 2 || gi == 4) ? 0.5f : (gi == 6 || gi == 7) ? LN2 : 1.f;
  }

  // Where this lane's colour values go: value k = k0 + lane is channel ch_off + k of some gradient segment;
  // its address for Gaussian gid is gcol + row * gld, row = gid (per camera) or gid - cam*N (shared).
  constexpr int NK = (NV + WAVE - 1) / WAVE;
  float *gcol[NK];
  int gld[NK];
  bool gshared[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int ch = ch_off + i * WAVE + lane;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    gcol[i] = (p && i * WAVE + lane < nch) ? p + (ch - gs.begin[sgm]) : nullptr;
    gld[i] = gs.ld[sgm];
    gshared[i] = gs.shared[sgm] != 0;
  }
  const int cam_first = cam * N;

  const float rx0 = (float)(tx * TS) + 0.5f, ry0 = (float)(ty * TS + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * TS + TS, width) - 0.5f;
  const float ry1 = (float)min(ty * TS + sub * Rg::RH + Rg::RH, height) - 0.5f;
  BatchGeometry pre = prefetch_batch(lane, max(range_lo, max_last - BS), max_last - max(range_lo, max_last - BS),
                                     flatten_ids, means2d, conics, opacities);
  for (int bend = max_last; bend > range_lo; bend -= BS) {
    const int first = max(range_lo, bend - BS);
    unsigned long long todo = stage_batch<CD, true, BS>(L, lane, pre, cs, N, D, ch_off, rx0, ry0, rx1, ry1);
#ifdef PR_DIAG
    if (PR_DIAG == 1 && lane == 0) {
      atomicAdd(&g_diag[6], 1ull);                                         // batches staged
      atomicAdd(&g_diag[7], (unsigned long long)(bend - first));           // list entries looked at
    }
#endif
    {  // the next batch's geometry travels while this one is composited
      const int nend = bend - BS, nfirst = max(range_lo, nend - BS);
      pre = prefetch_batch(lane, nfirst, nend - nfirst, flatten_ids, means2d, conics, opacities);
    }
    while (todo) {
      const int t = 63 - __builtin_clzll(todo);
      todo &= ~(1ull << t);
      const int idx = first + t;
#ifdef PR_DIAG
      const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      float dx[Rg::BW], dy[Rg::BH], adx2[Rg::BW], cdy2[Rg::BH], bdx[Rg::BW];
#pragma unroll
      for (int i = 0; i < Rg::BW; ++i) {
        dx[i] = g.x - (px0 + (float)(i * Rg::SX));
        adx2[i] = g.w * dx[i] * dx[i];
        bdx[i] = h.x * dx[i];
      }
#pragma unroll
      for (int i = 0; i < Rg::BH; ++i) {
        dy[i] = g.y - (py0 + (float)(i * Rg::SY));
        cdy2[i] = h.y * dy[i] * dy[i];
      }
      // e = exp(-sigma), ar = opacity * e, alpha = the alpha this pixel composited with, or 0 when the Gaussian
      // did not contribute to it.  Everything downstream is arithmetic on alpha (an alpha of 0 gives ra = 1,
      // fac = 0 and leaves T untouched exactly), so no per-pixel flags are carried through the loop body.
      float e[PPL], ar[PPL], alpha[PPL];
      unsigned long long slot[PPL], any = 0ull;  // per pixel slot: the lanes with an active pixel
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int jx = j % Rg::BW, jy = j / Rg::BW;
        const float power = adx2[jx] + cdy2[jy] + bdx[jx] * dy[jy];
        e[j] = __builtin_amdgcn_exp2f(power);
        ar[j] = g.z * e[j];
        const float a = fminf(ALPHA_MAX, ar[j]);
        const bool active = (power <= 0.f) && (a >= ALPHA_MIN) && (idx < last[j]);
        alpha[j] = active ? a : 0.f;
        slot[j] = __builtin_amdgcn_ballot_w64(active);
        any |= slot[j];
      }
#ifdef PR_DIAG
      if (PR_DIAG == 1 && lane == 0) atomicAdd(&g_diag[4], 1ull);   // iterations entered (staged entries walked)
#endif
      if (any == 0ull) continue;
#ifdef PR_DIAG
      const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
      if (PR_DIAG == 1) {
        int npx = 0, nslots = 0;
        for (int j = 0; j < PPL; ++j) {
          npx += __builtin_popcountll(slot[j]);
          nslots += slot[j] != 0;
        }
        if (lane == 0) {
          atomicAdd(&g_diag[0], 1ull);
          atomicAdd(&g_diag[1], (unsigned long long)npx);
          atomicAdd(&g_diag[2], (unsigned long long)nslots);
          atomicAdd(&g_diag[5], (unsigned long long)__builtin_popcountll(any));
          if (PPL == 4 && HALF_SKIP && ((slot[2 % PPL] | slot[3 % PPL]) == 0ull || (slot[0] | slot[1 % PPL]) == 0ull))
            atomicAdd(&g_diag[3], 1ull);   // iterations composited over half the tile only
        }
      }
#endif

      const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW);
      const int gid = L.ids[t];
      float *red = L.red;
      // Pixel slots J0..J1-1 of every lane; slots outside the range are known to have no active pixel in
      // any lane (an inactive pixel contributes exact zeros to every sum and leaves its state untouched).
      auto composite = [&](auto j0c, auto j1c) {
        constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
        float fac[PPL], ra[PPL];
        f2 S2[PPL];
#pragma unroll
        for (int j = J0; j < J1; ++j) {
          ra[j] = __builtin_amdgcn_rcpf(1.f - alpha[j]);
          T[j] *= ra[j];
          fac[j] = alpha[j] * T[j];
          S2[j] = splat(0.f);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (GEOM) c4 = crow[q];
          const f2 c01 = {c4.x, c4.y}, c23 = {c4.z, c4.w};
          f2 a01 = splat(0.f), a23 = splat(0.f);
#pragma unroll
          for (int j = J0; j < J1; ++j) {
            if (GEOM) {
              S2[j] = fma2(c01, vout[j][2 * q], S2[j]);
              S2[j] = fma2(c23, vout[j][2 * q + 1], S2[j]);
            }
            if (CGRAD) {
              const f2 f = splat(fac[j]);
              a01 = fma2(f, vout[j][2 * q], a01);
              a23 = fma2(f, vout[j][2 * q + 1], a23);
            }
          }
          if (CGRAD) red[q * 68 + lane] = quad_fold_rows(a01[0], a01[1], a23[0], a23[1]);
        }
        if (GEOM) {
          float geo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // x y a b c o |x| |y|
#pragma unroll
          for (int j = J0; j < J1; ++j) {
            const int jx = j % Rg::BW, jy = j / Rg::BW;
            const float S = S2[j][0] + S2[j][1];
            const float v_alpha = S * T[j] - (Bsum[j] - W[j]) * ra[j];
            Bsum[j] = fmaf(fac[j], S, Bsum[j]);
            // the exponential passes a gradient on when the pixel took this Gaussian at its unclamped alpha
            const float ev = (alpha[j] == ar[j] ? e[j] : 0.f) * v_alpha;
            geo[5] += ev;                    // d loss / d opacity
            const float vs = -g.z * ev;      // d loss / d sigma
            const float t1 = vs * dx[jx], t2 = vs * dy[jy];
            geo[2] = fmaf(t1, dx[jx], geo[2]);
            geo[3] = fmaf(t1, dy[jy], geo[3]);
            geo[4] = fmaf(t2, dy[jy], geo[4]);
            const float gx = 2.f * g.w * dx[jx] + h.x * dy[jy];  // = -log2e * (a dx + b dy)
            const float gy = h.x * dx[jx] + 2.f * h.y * dy[jy];
            const float vx = vs * gx, vy = vs * gy;
            geo[0] += vx;
            geo[1] += vy;
            geo[6] += fabsf(vx);
            geo[7] += fabsf(vy);
          }
          red[NQR * 68 + lane] = quad_fold_rows(geo[0], geo[1], geo[2], geo[3]);
          red[(NQR + 1) * 68 + lane] = quad_fold_rows(geo[4], geo[5], geo[6], geo[7]);
        }
      };
      // A Gaussian that only reaches the upper or the lower half of the tile (38 % of the contributing pairs
      // at the benchmark shape) is composited over that half's two pixel slots only: one wave-uniform branch
      // per Gaussian.  (Finer, per-slot branches were measured slower than doing the arithmetic.)
      using std::integral_constant;
      if (PPL == 4 && HALF_SKIP && (slot[2 % PPL] | slot[3 % PPL]) == 0ull)
        composite(integral_constant<int, 0>{}, integral_constant<int, PPL / 2>{});
      else if (PPL == 4 && HALF_SKIP && (slot[0] | slot[1 % PPL]) == 0ull)
        composite(integral_constant<int, PPL / 2>{}, integral_constant<int, PPL>{});
      // two pixels per lane: 24 % of the contributing iterations touch one of the two 16x4 strips only (-1.4 % kernel time)
      else if (PPL == 2 && HALF_SKIP && slot[1 % PPL] == 0ull)
        composite(integral_constant<int, 0>{}, integral_constant<int, (PPL + 1) / 2>{});
      else if (PPL == 2 && HALF_SKIP && slot[0] == 0ull)
        composite(integral_constant<int, PPL / 2>{}, integral_constant<int, PPL>{});
      else
        composite(integral_constant<int, 0>{}, integral_constant<int, PPL>{});
#ifdef PR_DIAG
      const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
#endif
      // one lane per value: colours go to v_colors[gid, ch_off + k], the 8 geometry values to v_geom[gid, :]
#pragma unroll
      for (int k0 = 0; k0 < NV; k0 += WAVE) {
        const int k = k0 + lane;
        if (k < NV) {
          const float4 *src = own + k0 / 4 * 17;  // 64 values further = 16 quads = 16*68 floats
          const float4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
          const f2 s0 = f2{p0.x, p0.y} + f2{p0.z, p0.w}, s1 = f2{p1.x, p1.y} + f2{p1.z, p1.w};
          const f2 s2 = f2{p2.x, p2.y} + f2{p2.z, p2.w}, s3 = f2{p3.x, p3.y} + f2{p3.z, p3.w};
          const f2 u = (s0 + s1) + (s2 + s3);
          const float v = u[0] + u[1];
          if (k < CDR) {
            float *dst = gcol[k0 / WAVE];
            if (dst) atomic_add_noret(dst + (int64_t)(gshared[k0 / WAVE] ? gid - cam_first : gid) * gld[k0 / WAVE], v);
          } else if (GEOM) {
            const int gi = k - CDR;
            const float sc = (k0 == 0) ? gscale
                                       : ((gi == 0 || gi == 1) ? -LN2
                                          : (gi == 2 || gi == 4) ? 0.5f
                                          : (gi == 6 || gi == 7) ? LN2 : 1.f);
            if (gi < (absgrad ? 8 : 6)) atomic_add_noret(v_geom + (int64_t)gid * gs.geom_ld + gi, v * sc);
          }
        }
      }
#ifdef PR_DIAG
      {
        const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
        dg_alpha += tk1 - tk0;   // record read + alpha + ballots
        dg_body += tk2 - tk1;    // contractions, folds, geometry
        dg_tail += tk3 - tk2;    // read-back + atomics
        dg_iters += 1;
      }
#endif
    }
  }
#ifdef PR_DIAG
  if (lane == 0 && PR_DIAG == 3 && blockIdx.x < 65536) {
    unsigned long long *t = g_wave_trace + 4 * (size_t)blockIdx.x;
    t[0] = tk_start;
    t[1] = __builtin_amdgcn_s_memtime();
    t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // XCC_ID
           (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                       // HW_ID
    t[3] = (unsigned long long)(max_last - range_lo);
  }
  if (lane == 0 && PR_DIAG == 2) {
    atomicAdd(&g_diag[0], dg_iters);
    atomicAdd(&g_diag[3], dg_alpha);
    atomicAdd(&g_diag[4], dg_body);
    atomicAdd(&g_diag[6], dg_tail);
    atomicAdd(&g_diag[7], __builtin_amdgcn_s_memtime() - tk_start);   // whole wave
  }
#endif
}

template <int TS, int PPL, int CD, bool GEOM, bool CGRAD>
static int launch_bwd(int C, int N, int D, int ch_off, const float *means2d, const float *conics, const ColorSegs &cs,
                      const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                      const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                      int64_t n_isects,
                      const float *render_alphas, const int32_t *last_ids, const float *render_colors,
                      int ed_channel, const float *v_render_colors, const float *v_render_alphas, int absgrad,
                      float *v_geom, const GradSegs &gs, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<TS, PPL>::WPT;
  const unsigned grid = region_grid(n_regions, Region<TS, PPL>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_bwd_kernel<TS, PPL, CD, GEOM, CGRAD>), dim3(grid), dim3(64 * WPB), 0, stream, (int)n_regions, N, D,
                     ch_off, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h,
                     isect_offsets, tile_order, flatten_ids, (int)n_isects, render_alphas, last_ids, render_colors, ed_channel,
                     v_render_colors, v_render_alphas, absgrad, v_geom, gs);
  return check_launch("raster_bwd_kernel");
}

// ---- geometry-only backward with the dot products on the matrix cores (16x16 tiles, 1 pixel per lane) -------
// For callers that ask for no colour gradients (the tracker) what is left per (pixel, Gaussian) is the dot product
// S = <colour_g, v_pixel> and a handful of scalars.  S for a whole staged batch is one matrix product,
// S^T (32 staged slots x 32 pixels) = C (32 x channels) . V^T (channels x 32 pixels) per half of the wavefront, on
// v_mfma_f32_32x32x16_bf16 with both operands split into bf16 head + residual (hi.hi + lo.hi + hi.lo).  V^T is
// constant for the tile: it becomes the B operand once, in the prologue (the lanes of a pixel's own registers are
// the operand's register index; one v_permlane32_swap per register pair sorts out the two 32-pixel halves), and
// the float32 cotangents are then dropped from the registers.  The A operand is the staged colour rows as they lie
// in LDS (lane = slot).  16 more swaps bring S[pixel][slot] home to the pixel's lane; the back-to-front walk over
// the slots is unrolled so that every slot's S is a fixed register.  Channels past the last full 16 stay on the
// vector ALU.
template <int CD>
__global__ void __launch_bounds__(64) raster_bwd_geom_mfma_kernel(
    int n_regions, int N, int D, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int width,
    int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
    const float *__restrict__ render_colors, int ed_channel, const float *__restrict__ v_render_colors,
    const float *__restrict__ v_render_alphas, int absgrad, float *__restrict__ v_geom, int geom_ld) {
  using Rg = Region<16, 1>;
  constexpr int ROW = color_row(CD), CDP = round4(CD), BS = 32;
  constexpr int KS = CDP / 16, CM = 16 * KS, CV = CDP - CM;  // k steps / channels on the matrix cores / on the VALU
  static_assert(KS >= 1 && CV >= 0 && CV < 16, "pass width");
  __shared__ WaveLds<CD, false, BS> L;
  __shared__ float red[2 * 68];
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x = tx * 16 + (lane % Rg::LX), y = ty * 16 + sub * Rg::RH + (lane / Rg::LX);
  const int range_lo = isect_offsets[tile_flat];
  const int nch = min(CD, D);
  const float px0 = (float)x + 0.5f, py0 = (float)y + 0.5f;

  float T = 1.f, Bsum = 0.f, W = 0.f;
  int last = range_lo;
  unsigned b0h[KS][4], b1h[KS][4], b0l[KS][4], b1l[KS][4];  // V^T as B operand: pixels 0-31 / 32-63, head / residual
  float vrest[CV > 0 ? CV : 1];
  {
    float v[CDP];
#pragma unroll
    for (int k = 0; k < CDP; ++k) v[k] = 0.f;
    if (x < width && y < height) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      T = 1.f - render_alphas[pix];
      last = last_ids[pix];
      const float *src = v_render_colors + pix * D;
      if ((D & 3) == 0) {
#pragma unroll
        for (int k = 0; k + 3 < CDP; k += 4)
          if (k < nch) {
            const float4 q = *reinterpret_cast<const float4 *>(src + k);
            v[k] = q.x, v[k + 1] = q.y, v[k + 2] = q.z, v[k + 3] = q.w;
          }
      } else {
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) v[k] = src[k];
      }
      float w_ed = 0.f;
      if (ed_channel >= 0) {  // expected depth out_d = X / max(a, 1e-10): see raster_bwd_kernel
        const float a = 1.f - T, inv = 1.f / fmaxf(a, 1e-10f);
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k == ed_channel) {
            if (a > 1e-10f) w_ed = -v[k] * render_colors[pix * D + ed_channel] * inv;
            v[k] *= inv;
          }
      }
      float w = v_render_alphas[pix] + w_ed;
      if (backgrounds) {
        const float *bg = backgrounds + (int64_t)cam * D;
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) w -= bg[k] * v[k];
      }
      W = T * w;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned xh, xl, yh, yl;
        split_pair(v[16 * ks + 2 * i], v[16 * ks + 2 * i + 1], xh, xl);
        split_pair(v[16 * ks + 8 + 2 * i], v[16 * ks + 9 + 2 * i], yh, yl);
        swap32(xh, yh, b0h[ks][i], b1h[ks][i]);
        swap32(xl, yl, b0l[ks][i], b1l[ks][i]);
      }
#pragma unroll
    for (int k = 0; k < CV; ++k) vrest[k] = v[CM + k];
  }
  int max_last = last;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));

  const int own_q = lane >> 2, own_row = ((lane & 1) << 1) | ((lane >> 1) & 1);
  const float4 *own = reinterpret_cast<const float4 *>(red + own_q * 68 + own_row * 16);
  const float gscale = (lane == 0 || lane == 1) ? -LN2 : (lane == 2 || lane == 4) ? 0.5f
                       : (lane == 6 || lane == 7) ? LN2 : 1.f;
  const int n_geo = absgrad ? 8 : 6;
  const int mrow = lane & 31, mh = lane >> 5;

  const float rx0 = (float)(tx * 16) + 0.5f, ry0 = (float)(ty * 16 + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * 16 + 16, width) - 0.5f;
  const float ry1 = (float)min(ty * 16 + sub * Rg::RH + Rg::RH, height) - 0.5f;
  BatchGeometry pre = prefetch_batch(lane, max(range_lo, max_last - BS), max_last - max(range_lo, max_last - BS),
                                     flatten_ids, means2d, conics, opacities);
  for (int bend = max_last; bend > range_lo; bend -= BS) {
    const int first = max(range_lo, bend - BS);
    const unsigned long long todo = stage_batch<CD, false, BS>(L, lane, pre, cs, N, D, 0, rx0, ry0, rx1, ry1);
    {
      const int nend = bend - BS, nfirst = max(range_lo, nend - BS);
      pre = prefetch_batch(lane, nfirst, nend - nfirst, flatten_ids, means2d, conics, opacities);
    }
    if (todo == 0ull) continue;
    // S for every staged slot and every pixel of the region
    f16v s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 *row = reinterpret_cast<const float4 *>(L.col + mrow * ROW + 16 * ks + 8 * mh);
      const float4 c0 = row[0], c1 = row[1];
      unsigned ah[4], al[4];
      split_pair(c0.x, c0.y, ah[0], al[0]);
      split_pair(c0.z, c0.w, ah[1], al[1]);
      split_pair(c1.x, c1.y, ah[2], al[2]);
      split_pair(c1.z, c1.w, ah[3], al[3]);
      s0 = mfma_bf16(ah, b0h[ks], s0);
      s1 = mfma_bf16(ah, b1h[ks], s1);
      s0 = mfma_bf16(al, b0h[ks], s0);
      s1 = mfma_bf16(al, b1h[ks], s1);
      s0 = mfma_bf16(ah, b0l[ks], s0);
      s1 = mfma_bf16(ah, b1l[ks], s1);
    }
    // s0 = pixels 0-31, s1 = pixels 32-63, rows (slots) of lane half mh -> every pixel's own lane, all 32 slots:
    // sa[r] = slot (r&3) + 8 (r>>2), sb[r] = that + 4
    float sa[16], sb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // Written as inline assembly on purpose: hipcc (ROCm 7.2) merges the sixteen permlane32_swap builtin calls on
      // the elements of an accumulator vector into ONE (every slot then sees slot 0's dot product).
      float fa = s0[r], fb = s1[r];
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(fa), "+v"(fb));
      sa[r] = fa;
      sb[r] = fb;
    }
#pragma unroll
    for (int t = BS - 1; t >= 0; --t) {
      if (!((todo >> t) & 1ull)) continue;
      const int idx = first + t;
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      const float dx = g.x - px0, dy = g.y - py0;
      const float power = g.w * dx * dx + h.y * dy * dy + h.x * dx * dy;
      const float e = __builtin_amdgcn_exp2f(power), ar = g.z * e;
      const float a = fminf(ALPHA_MAX, ar);
      const bool active = (power <= 0.f) && (a >= ALPHA_MIN) && (idx < last);
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) continue;
      const float alpha = active ? a : 0.f;
      float S = ((t >> 2) & 1) ? sb[(t & 3) + 4 * (t >> 3)] : sa[(t & 3) + 4 * (t >> 3)];
      if (CV > 0) {
        const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW + CM);
#pragma unroll
        for (int q = 0; q < CV / 4; ++q) {
          const float4 c = crow[q];
          S = fmaf(c.x, vrest[4 * q], S);
          S = fmaf(c.y, vrest[4 * q + 1], S);
          S = fmaf(c.z, vrest[4 * q + 2], S);
          S = fmaf(c.w, vrest[4 * q + 3], S);
        }
      }
      const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
      T *= ra;
      const float fac = alpha * T;
      const float v_alpha = S * T - (Bsum - W) * ra;
      Bsum = fmaf(fac, S, Bsum);
      const float ev = (alpha == ar ? e : 0.f) * v_alpha;
      const float vs = -g.z * ev;
      const float t1 = vs * dx, t2 = vs * dy;
      const float gx = 2.f * g.w * dx + h.x * dy, gy = h.x * dx + 2.f * h.y * dy;
      const float vx = vs * gx, vy = vs * gy;
      red[lane] = quad_fold_rows(vx, vy, t1 * dx, t1 * dy);
      red[68 + lane] = quad_fold_rows(t2 * dy, ev, fabsf(vx), fabsf(vy));
      const int gid = L.ids[t];
      if (lane < 8) {
        const float4 p0 = own[0], p1 = own[1], p2 = own[2], p3 = own[3];
        const float tot = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w)) +
                          ((p2.x + p2.y) + (p2.z + p2.w)) + ((p3.x + p3.y) + (p3.z + p3.w));
        if (lane < n_geo) atomic_add_noret(v_geom + (int64_t)gid * geom_ld + lane, tot * gscale);
      }
    }
  }
}

template <int CD>
static int launch_bwd_geom_mfma(int C, int N, int D, const float *means2d, const float *conics, const ColorSegs &cs,
                                const float *opacities, const float *backgrounds, int width, int height, int tile_w,
                                int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                                const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas,
                                const int32_t *last_ids, const float *render_colors, int ed_channel,
                                const float *v_render_colors, const float *v_render_alphas, int absgrad, float *v_geom,
                                int geom_ld, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<16, 1>::WPT;
  const unsigned grid = region_grid(n_regions, Region<16, 1>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_bwd_geom_mfma_kernel<CD>), dim3(grid), dim3(64), 0, stream, (int)n_regions,
                     N, D, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,
                     tile_order, flatten_ids, (int)n_isects, render_alphas, last_ids, render_colors, ed_channel,
                     v_render_colors, v_render_alphas, absgrad, v_geom, geom_ld);
  return check_launch("raster_bwd_geom_mfma_kernel");
}

// ---- full backward (geometry + colour gradients) with BOTH channel contractions on the matrix cores ----------
// 16x16 tiles, 1 pixel per lane (a 16x4 strip per wavefront), passes of 32..72 channels.  Per batch of 32 staged
// Gaussians:
//   (1) S[g][p] = <colour_g, v_p>: as in the geometry-only kernel above -- A = staged colour rows from LDS, B = v^T
//       built once per tile from the lanes' own registers.
//   (2) vC[g][c] = sum_p fac[p][g] * v[p][c] reduces over the PIXEL index, which lives on the lanes.  Its B operand
//       v[p][c] (k = pixel, column = channel) is constant for the tile: transposed through LDS once, in the prologue,
//       and kept as split-bf16 registers.  Its A operand fac^T[g][p] is produced by the back-to-front walk: every lane
//       stores the bf16 head and residual of its pixel's fac into row g of a [32 slots][64 pixels] LDS image, which is
//       read back row-wise (one 16-byte read per operand) after the walk.  The product replaces, per contributing
//       (wavefront, Gaussian), the D packed FMAs, D/4 lane-swap folds and the per-Gaussian LDS read-back of the vector
//       kernel.  Rows of slots that did not contribute hold stale data and are never written out (rows of a matrix
//       product are independent).
//   Result layout: lane = channel (l & 31), register r = slot (r&3) + 8(r>>2) + 4(l>>5): one float atomic instruction
//   covers two 128-byte row segments, the shape global float atomics run at full rate for.
// Channels past the last full 32 of the pass (the expected-depth channel among them) keep the vector path: fac * v is
// folded with the geometry values.  Both contractions use hi.hi + lo.hi + hi.lo (about 16 mantissa bits per product);
// gradients are held to 1e-3 relative, three orders above that.
template <int CD>
__global__ void __launch_bounds__(64) raster_bwd_full_mfma_kernel(
    int n_regions, int N, int D, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int width,
    int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
    const float *__restrict__ render_colors, int ed_channel, const float *__restrict__ v_render_colors,
    const float *__restrict__ v_render_alphas, int absgrad, float *__restrict__ v_geom, const GradSegs gs) {
  using Rg = Region<16, 1>;
  constexpr int ROW = color_row(CD), CDP = round4(CD), BS = 32;
  constexpr int KS = CDP / 16, CM = 16 * KS, CV = CDP - CM;  // S: k steps / channels on the matrix cores / on the VALU
  constexpr int NT = CDP / 32, CC = 32 * NT, CR = CDP - CC;  // vC: 32-channel tiles / their channels / the rest
  constexpr int NQR = CR / 4;                                // rest channels as quads in the reduction area
  constexpr int FP = 72;                                     // fac image row pitch in 16-bit units: 64 pixels + pad
  static_assert(KS >= 2 && NT >= 1 && CV < 16 && CR < 32 && CR % 4 == 0, "pass width");
  __shared__ WaveLds<CD, false, BS> L;
  __shared__ float red[(2 + NQR) * 68];
  union FacImage {
    struct {
      unsigned short hi[BS * FP], lo[BS * FP];
    } f;
    float tr[64 * 33];  // prologue: v of the 64 pixels, one 32-channel block at a time, for the transpose
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
  const int x = tx * 16 + (lane % Rg::LX), y = ty * 16 + sub * Rg::RH + (lane / Rg::LX);
  const int range_lo = isect_offsets[tile_flat];
  const int nch = min(CD, D);
  const float px0 = (float)x + 0.5f, py0 = (float)y + 0.5f;
  const int mrow = lane & 31, mh = lane >> 5;

  float T = 1.f, Bsum = 0.f, W = 0.f;
  int last = range_lo;
  unsigned b0h[KS][4], b1h[KS][4], b0l[KS][4], b1l[KS][4];  // v^T as B operand of S: pixels 0-31 / 32-63, head / residual
  unsigned vbh[NT][4][4], vbl[NT][4][4];                    // v as B operand of vC: [channel tile][k step][register]
  float vrest[CV > 0 ? CV : 1];                             // channels CM.. for S on the VALU
  float crest[CR > 0 ? CR : 1];                             // channels CC.. for vC on the VALU
  {
    float v[CDP];
#pragma unroll
    for (int k = 0; k < CDP; ++k) v[k] = 0.f;
    if (x < width && y < height) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      T = 1.f - render_alphas[pix];
      last = last_ids[pix];
      const float *src = v_render_colors + pix * D;
      if ((D & 3) == 0) {
#pragma unroll
        for (int k = 0; k + 3 < CDP; k += 4)
          if (k < nch) {
            const float4 q = *reinterpret_cast<const float4 *>(src + k);
            v[k] = q.x, v[k + 1] = q.y, v[k + 2] = q.z, v[k + 3] = q.w;
          }
      } else {
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) v[k] = src[k];
      }
      float w_ed = 0.f;
      if (ed_channel >= 0) {  // expected depth out_d = X / max(a, 1e-10): see raster_bwd_kernel
        const float a = 1.f - T, inv = 1.f / fmaxf(a, 1e-10f);
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k == ed_channel) {
            if (a > 1e-10f) w_ed = -v[k] * render_colors[pix * D + ed_channel] * inv;
            v[k] *= inv;
          }
      }
      float w = v_render_alphas[pix] + w_ed;
      if (backgrounds) {
        const float *bg = backgrounds + (int64_t)cam * D;
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k < nch) w -= bg[k] * v[k];
      }
      W = T * w;
    }
    // vC operand: transpose each 32-channel block through LDS (pixel-major rows of 33 floats), read back by channel
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int k = 0; k < 32; ++k) F.tr[lane * 33 + k] = v[32 * nt + k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p0 = 16 * s4 + 8 * mh + 2 * i;
          split_pair(F.tr[p0 * 33 + mrow], F.tr[(p0 + 1) * 33 + mrow], vbh[nt][s4][i], vbl[nt][s4][i]);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned xh, xl, yh, yl;
        split_pair(v[16 * ks + 2 * i], v[16 * ks + 2 * i + 1], xh, xl);
        split_pair(v[16 * ks + 8 + 2 * i], v[16 * ks + 9 + 2 * i], yh, yl);
        swap32(xh, yh, b0h[ks][i], b1h[ks][i]);
        swap32(xl, yl, b0l[ks][i], b1l[ks][i]);
      }
#pragma unroll
    for (int k = 0; k < CV; ++k) vrest[k] = v[CM + k];
#pragma unroll
    for (int k = 0; k < CR; ++k) crest[k] = v[CC + k];
  }
  int max_last = last;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));

  // lane v < 8 + CR owns reduced value v (0-7 geometry, 8.. rest colours): its 16 partials sit in one red row
  const int own_q = lane >> 2, own_row = ((lane & 1) << 1) | ((lane >> 1) & 1);
  const float4 *own = reinterpret_cast<const float4 *>(red + own_q * 68 + own_row * 16);
  const float gscale = (lane == 0 || lane == 1) ? -LN2 : (lane == 2 || lane == 4) ? 0.5f
                       : (lane == 6 || lane == 7) ? LN2 : 1.f;
  const int n_geo = absgrad ? 8 : 6;
  // gradient destinations: channel tile nt, column mrow (matrix part); channel CC + lane - 8 (vector part)
  float *mcol[NT];
  int mld[NT];
  bool mshared[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ch = 32 * nt + mrow;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    mcol[nt] = (p && ch < nch) ? p + (ch - gs.begin[sgm]) : nullptr;
    mld[nt] = gs.ld[sgm];
    mshared[nt] = gs.shared[sgm] != 0;
  }
  float *rcol = nullptr;
  int rld = 0;
  bool rshared = false;
  if (CR > 0 && lane >= 8 && lane < 8 + CR) {
    const int ch = CC + lane - 8;
    const int sgm = ch >= gs.begin[2] ? 2 : ch >= gs.begin[1] ? 1 : 0;
    float *p = gs.ptr[sgm];
    rcol = (p && ch < nch) ? p + (ch - gs.begin[sgm]) : nullptr;
    rld = gs.ld[sgm];
    rshared = gs.shared[sgm] != 0;
  }
  const int cam_first = cam * N;

  const float rx0 = (float)(tx * 16) + 0.5f, ry0 = (float)(ty * 16 + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * 16 + 16, width) - 0.5f;
  const float ry1 = (float)min(ty * 16 + sub * Rg::RH + Rg::RH, height) - 0.5f;
  BatchGeometry pre = prefetch_batch(lane, max(range_lo, max_last - BS), max_last - max(range_lo, max_last - BS),
                                     flatten_ids, means2d, conics, opacities);
  for (int bend = max_last; bend > range_lo; bend -= BS) {
    const int first = max(range_lo, bend - BS);
    const unsigned long long todo = stage_batch<CD, false, BS>(L, lane, pre, cs, N, D, 0, rx0, ry0, rx1, ry1);
    {
      const int nend = bend - BS, nfirst = max(range_lo, nend - BS);
      pre = prefetch_batch(lane, nfirst, nend - nfirst, flatten_ids, means2d, conics, opacities);
    }
    if (todo == 0ull) continue;
    // (1) S for every staged slot and every pixel of the region
    f16v s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 *row = reinterpret_cast<const float4 *>(L.col + mrow * ROW + 16 * ks + 8 * mh);
      const float4 c0 = row[0], c1 = row[1];
      unsigned ah[4], al[4];
      split_pair(c0.x, c0.y, ah[0], al[0]);
      split_pair(c0.z, c0.w, ah[1], al[1]);
      split_pair(c1.x, c1.y, ah[2], al[2]);
      split_pair(c1.z, c1.w, ah[3], al[3]);
      s0 = mfma_bf16(ah, b0h[ks], s0);
      s1 = mfma_bf16(ah, b1h[ks], s1);
      s0 = mfma_bf16(al, b0h[ks], s0);
      s1 = mfma_bf16(al, b1h[ks], s1);
      s0 = mfma_bf16(ah, b0l[ks], s0);
      s1 = mfma_bf16(ah, b1l[ks], s1);
    }
    float sa[16], sb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // inline assembly on purpose: see raster_bwd_geom_mfma_kernel
      float fa = s0[r], fb = s1[r];
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(fa), "+v"(fb));
      sa[r] = fa;
      sb[r] = fb;
    }
    unsigned contributed = 0u;  // slots of this batch that some pixel of the region took
#pragma unroll
    for (int t = BS - 1; t >= 0; --t) {
      if (!((todo >> t) & 1ull)) continue;
      const int idx = first + t;
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      const float dx = g.x - px0, dy = g.y - py0;
      const float power = g.w * dx * dx + h.y * dy * dy + h.x * dx * dy;
      const float e = __builtin_amdgcn_exp2f(power), ar = g.z * e;
      const float a = fminf(ALPHA_MAX, ar);
      const bool active = (power <= 0.f) && (a >= ALPHA_MIN) && (idx < last);
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) continue;
      contributed |= 1u << t;
      const float alpha = active ? a : 0.f;
      float S = ((t >> 2) & 1) ? sb[(t & 3) + 4 * (t >> 3)] : sa[(t & 3) + 4 * (t >> 3)];
      if (CV > 0) {
        const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW + CM);
#pragma unroll
        for (int q = 0; q < CV / 4; ++q) {
          const float4 c = crow[q];
          S = fmaf(c.x, vrest[4 * q], S);
          S = fmaf(c.y, vrest[4 * q + 1], S);
          S = fmaf(c.z, vrest[4 * q + 2], S);
          S = fmaf(c.w, vrest[4 * q + 3], S);
        }
      }
      const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
      T *= ra;
      const float fac = alpha * T;
      {  // this pixel's entry of row t of the fac image
        unsigned fh, fl;
        split_pair(fac, 0.f, fh, fl);
        F.f.hi[t * FP + lane] = (unsigned short)fh;
        F.f.lo[t * FP + lane] = (unsigned short)fl;
      }
      const float v_alpha = S * T - (Bsum - W) * ra;
      Bsum = fmaf(fac, S, Bsum);
      const float ev = (alpha == ar ? e : 0.f) * v_alpha;
      const float vs = -g.z * ev;
      const float t1 = vs * dx, t2 = vs * dy;
      const float gx = 2.f * g.w * dx + h.x * dy, gy = h.x * dx + 2.f * h.y * dy;
      const float vx = vs * gx, vy = vs * gy;
      red[lane] = quad_fold_rows(vx, vy, t1 * dx, t1 * dy);
      red[68 + lane] = quad_fold_rows(t2 * dy, ev, fabsf(vx), fabsf(vy));
#pragma unroll
      for (int q = 0; q < NQR; ++q)
        red[(2 + q) * 68 + lane] = quad_fold_rows(fac * crest[4 * q], fac * crest[4 * q + 1], fac * crest[4 * q + 2],
                                                  fac * crest[4 * q + 3]);
      const int gid = L.ids[t];
      if (lane < 8 + CR) {
        const float4 p0 = own[0], p1 = own[1], p2 = own[2], p3 = own[3];
        const float tot = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w)) +
                          ((p2.x + p2.y) + (p2.z + p2.w)) + ((p3.x + p3.y) + (p3.z + p3.w));
        if (lane < n_geo) atomic_add_noret(v_geom + (int64_t)gid * gs.geom_ld + lane, tot * gscale);
        if (CR > 0 && lane >= 8 && rcol) atomic_add_noret(rcol + (int64_t)(rshared ? gid - cam_first : gid) * rld, tot);
      }
    }
    if (contributed == 0u) continue;
    // (2) vC[slot][channel] = sum over the region's 64 pixels of fac * v, rows = slots
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const u4v ah4 = *reinterpret_cast<const u4v *>(&F.f.hi[mrow * FP + 16 * s4 + 8 * mh]);
        const u4v al4 = *reinterpret_cast<const u4v *>(&F.f.lo[mrow * FP + 16 * s4 + 8 * mh]);
        const unsigned ah[4] = {ah4[0], ah4[1], ah4[2], ah4[3]}, al[4] = {al4[0], al4[1], al4[2], al4[3]};
        acc = mfma_bf16(ah, vbh[nt][s4], acc);
        acc = mfma_bf16(al, vbh[nt][s4], acc);
        acc = mfma_bf16(ah, vbl[nt][s4], acc);
      }
      float *dst = mcol[nt];
      settle_mfma_results();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int slot = (r & 3) + 8 * (r >> 2) + 4 * mh;
        if (dst && ((contributed >> slot) & 1u)) {
          const int gid = L.ids[slot];
          atomic_add_noret(dst + (int64_t)(mshared[nt] ? gid - cam_first : gid) * mld[nt], acc[r]);
        }
      }
    }
  }
}

template <int CD>
static int launch_bwd_full_mfma(int C, int N, int D, const float *means2d, const float *conics, const ColorSegs &cs,
                                const float *opacities, const float *backgrounds, int width, int height, int tile_w,
                                int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                                const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas,
                                const int32_t *last_ids, const float *render_colors, int ed_channel,
                                const float *v_render_colors, const float *v_render_alphas, int absgrad, float *v_geom,
                                const GradSegs &gs, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<16, 1>::WPT;
  const unsigned grid = region_grid(n_regions, Region<16, 1>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_bwd_full_mfma_kernel<CD>), dim3(grid), dim3(64), 0, stream, (int)n_regions, N, D, means2d,
                     conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order,
                     flatten_ids, (int)n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors,
                     v_render_alphas, absgrad, v_geom, gs);
  return check_launch("raster_bwd_full_mfma_kernel");
}

#ifdef PR_DIAG
extern "C" int pr_diag_trace_read(unsigned long long *out, int n_waves) {
  if (n_waves > 65536) n_waves = 65536;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pr::g_wave_trace), sizeof(unsigned long long) * 4 * n_waves) != hipSuccess)
    return -4;
  return 0;
}
extern "C" int pr_diag_read(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pr::g_diag), sizeof(pr::g_diag)) != hipSuccess) return -4;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pr::g_diag), z, sizeof(z)) != hipSuccess) return -4;
  }
  return 0;
}
#endif

static const int kChunks[] = {4, 8, 16, 24, 32, 36, 48, 64, 72, 100};
static int pick_chunk(int want) {
  for (int c : kChunks)
    if (c >= want) return c;
  return -1;
}

struct BwdPlan {
  int ppl, per;
};

// Does the call go to raster_bwd_mm_kernel (8x8 pixel regions, channel contractions as float32 matrix products)?
// POGS_RAST_BWD_MFMA=5 forces it wherever it is built; unset, it takes every call except narrow passes (up to 8
// channels) on 16x16 tiles with geometry gradients -- with so few channels the contractions are a small part of an
// iteration and the vector kernel's larger regions (half tiles) win: measured at 1M Gaussians, 1080p, 4 channels
// 0.51 ms against 0.62; at 36 channels 1.19 against 1.02, at 64 channels on a 450x250 image 0.98 against 0.33.  Any
// other value of the variable, or a forced pixels-per-lane count, keeps the older kernels (tests, A/B runs).
static bool plan_mm(int D, int tile_size, int geom_grad, int64_t n_tiles_total) {
  const int cd = pick_chunk(D);
  if (cd < 0 || !bwd_mm_supports(cd, tile_size)) return false;
  const int mode = tuning().bwd_mfma;
  if (mode == 5 || mode == 6) return true;   // 6: as 5 with geometry and colour gradients as separate launches
  if (mode != -1 || tuning().bwd_ppl != 0) return false;
  // (under 3072 tiles the half-tile kernel is a few long walks per SIMD and the 8x8 regions win at any width: the
  // tracker's RGB+ED backward at 640x360, 920 tiles, 0.196 ms against 0.176)
  return !(tile_size == 16 && cd <= 8 && geom_grad && n_tiles_total >= 3072);
}

// Geometry gradients need the dot product over ALL channels in one pass; colour-only gradients can be
// chunked freely (narrow chunks keep 4 pixels per lane).  POGS_RAST_BWD_PPL overrides (tuning aid).
static BwdPlan plan_bwd(int D, int geom_grad, bool color_grad, int tile_size, int64_t n_tiles_total) {
  BwdPlan p;
  p.per = D;
  if (!geom_grad && D > 36) {
    const int n_chunks = (D + 35) / 36;
    p.per = ((D + n_chunks - 1) / n_chunks + 3) & ~3;
  }
  if (tile_size == 8) {
    p.ppl = 1;
    return p;
  }
  const int cd = pick_chunk(p.per);
  // One wavefront per 16x16 tile (4 pixels per lane) amortises the per-Gaussian work best, but its jobs are long: with
  // ~8k tiles a launch is four rounds of jobs per wavefront slot and the last one runs half empty.  Measured at 36
  // channels, 1M-3M Gaussians: 1080p (8160 tiles) 1.16-1.53 ms at 2 pixels per lane against 1.19-1.60 at 4; 4K (32400
  // tiles) 3.89 against 3.60.
  p.ppl = cd <= 36 ? (n_tiles_total >= 16384 ? 4 : 2) : cd <= 72 ? 2 : 1;
  // without colour gradients the per-Gaussian folds are few and the registers of two pixels per lane allow three
  // wavefronts per SIMD: measured 0.74 ms against 0.81 ms at 36 channels, 1M Gaussians, 1080p
  if (!color_grad && p.ppl == 4) p.ppl = 2;
  while (p.ppl > 1 && n_tiles_total * (4 / p.ppl) < 3072) p.ppl >>= 1;
  const int forced = tuning().bwd_ppl;
  if (forced == 1 || (forced == 2 && cd <= 72) || (forced == 4 && cd <= 36)) p.ppl = forced;
  return p;
}

}  // namespace pr

using namespace pr;

extern "C" int pr_max_channels(void) { return 100; }

extern "C" int pr_raster_bwd_plan(int D, int tile_size, int64_t n_tiles_total, int geom_grad, int color_grad,
                                  int *pixels_per_lane, int *channels_per_pass, int *matrix_cores) {
  if (D <= 0 || D > pr_max_channels() || (tile_size != 16 && tile_size != 8) || n_tiles_total < 0) return PR_EINVAL;
  if (!geom_grad && !color_grad) return PR_EINVAL;
  const BwdPlan plan = plan_bwd(D, geom_grad, color_grad != 0, tile_size, n_tiles_total);
  if (pixels_per_lane) *pixels_per_lane = plan.ppl;
  if (channels_per_pass) *channels_per_pass = plan.per;
  if (plan_mm(D, tile_size, geom_grad, n_tiles_total)) {
    if (pixels_per_lane) *pixels_per_lane = 1;
    if (channels_per_pass) *channels_per_pass = geom_grad ? D : min(D, 48);
    if (matrix_cores) *matrix_cores = 1;
    return PR_OK;
  }
  if (matrix_cores) {
    const int cd = pick_chunk(D);
    const int mode = tuning().bwd_mfma == -1 ? 1 : tuning().bwd_mfma;
    *matrix_cores = (geom_grad && !color_grad && tile_size == 16 && plan.ppl == 1 && mode != 0 && cd >= 16 && cd <= 72) ||
                    (geom_grad && color_grad && tile_size == 16 && mode == 2 && plan.per == D && (cd == 32 || cd == 36 || cd == 64 || cd == 72)) ||
                    (geom_grad && color_grad && tile_size == 16 && mode == 3 && plan.per == D && plan.ppl == 2 && (cd == 32 || cd == 36)) ||
                    (color_grad && tile_size == 16 && mode == 4 && D <= 48);
  }
  return PR_OK;
}

#define BWD_ARGS                                                                                                     \
  C, N, D, off, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,           \
      tile_order, flatten_ids, n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors,        \
      v_render_alphas, absgrad, v_geom, gs, (hipStream_t)stream
#define BWD_CASE(TS, PPL, CDV)                                      \
  case CDV:                                                         \
    rc = !geom_grad       ? launch_bwd<TS, PPL, CDV, false, true>(BWD_ARGS) \
         : any_color_grad ? launch_bwd<TS, PPL, CDV, true, true>(BWD_ARGS)  \
                          : launch_bwd<TS, PPL, CDV, true, false>(BWD_ARGS); \
    break;

extern "C" int pr_raster_bwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                             const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                             int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                             const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas,
                             const int32_t *last_ids, const float *render_colors, int ed_channel,
                             const float *v_render_colors, const float *v_render_alphas, int absgrad, int geom_grad,
                             float *v_geom, int v_geom_ld, float *v_colors, const pr_color_segment *segments,
                             const pr_grad_segment *v_segments, int n_segments, void *stream) {
  if (C < 0 || N < 0 || D <= 0 || width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || n_isects < 0)
    return PR_EINVAL;
  if (tile_size != 16 && tile_size != 8) return PR_EUNSUPPORTED;
  if (D > pr_max_channels() || n_isects > 2147483647ll - 4096 /* list positions plus a batch of look-ahead stay inside int32 */) return PR_EUNSUPPORTED;
  if ((int64_t)tile_w * tile_size < width || (int64_t)tile_h * tile_size < height) return PR_EINVAL;
  if (C == 0 || n_isects == 0) return PR_OK;
  if (!means2d || !conics || !opacities || !isect_offsets || !flatten_ids || !render_alphas || !last_ids ||
      !v_render_colors || !v_render_alphas)
    return PR_EINVAL;
  if (n_segments > 0 && !v_segments) return PR_EINVAL;
  ColorSegs cs;
  GradSegs gs;
  int rcs = make_color_segs(colors, D, segments, n_segments, true, cs);
  if (!rcs) rcs = make_grad_segs(v_colors, D, v_segments, n_segments, gs);
  if (rcs) return rcs;
  if (v_geom_ld != 0 && v_geom_ld < 8) return PR_EINVAL;
  gs.geom_ld = v_geom_ld ? v_geom_ld : 8;
  for (int i = 0; i < gs.n; ++i)
    if (n_segments > 0 && (gs.begin[i] != cs.begin[i])) return PR_EINVAL;  // widths must match the colours'
  bool any_color_grad = false;
  for (int i = 0; i < gs.n; ++i) any_color_grad = any_color_grad || gs.ptr[i] != nullptr;
  if (geom_grad && !v_geom) return PR_EINVAL;
  if (ed_channel >= D || (ed_channel >= 0 && !render_colors)) return PR_EINVAL;
  if (!geom_grad && !any_color_grad) return PR_OK;
  if ((int64_t)C * tile_w * tile_h * 4 > 2147483647ll) return PR_EUNSUPPORTED;
  // Colour gradients alone, up to 48 channels: the forward walk replayed on the records, contraction on the matrix cores
  if (!geom_grad && any_color_grad && tile_size == 16 && tuning().bwd_mfma == 4 && D <= 48)
    return launch_bwd_colour_mfma(C, N, D, means2d, conics, opacities, width, height, tile_w, tile_h, isect_offsets,
                                  tile_order, flatten_ids, render_alphas, last_ids, ed_channel, v_render_colors, gs,
                                  (hipStream_t)stream);
  // Both kinds of gradients, up to 48 channels: the geometry-only kernel, then the colour gradients by themselves with
  // their contraction on the matrix cores (raster_bwd_cmfma.hip)
  if (geom_grad && any_color_grad && tile_size == 16 && tuning().bwd_mfma == 4 && D <= 48) {
    pr_grad_segment none[3];
    for (int i = 0; i < 3; ++i) {
      none[i].ptr = nullptr;
      none[i].width = i < n_segments ? v_segments[i].width : 0;
      none[i].ld = i < n_segments ? v_segments[i].ld : 0;
      none[i].shared = i < n_segments ? v_segments[i].shared : 0;
    }
    const int rc = pr_raster_bwd(C, N, D, means2d, conics, colors, opacities, backgrounds, width, height, tile_size,
                                 tile_w, tile_h, isect_offsets, tile_order, flatten_ids, n_isects, render_alphas,
                                 last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas, absgrad, 1, v_geom,
                                 v_geom_ld, nullptr, segments, none, n_segments, stream);
    if (rc) return rc;
    return launch_bwd_colour_mfma(C, N, D, means2d, conics, opacities, width, height, tile_w, tile_h, isect_offsets,
                                  tile_order, flatten_ids, render_alphas, last_ids, ed_channel, v_render_colors, gs,
                                  (hipStream_t)stream);
  }
  // geometry (+ colour) gradients with both contractions on the float32 matrix cores, 8x8 pixel regions
  // (it addresses rows with 32-bit byte offsets from each tensor's base)
  bool small_tensors = (int64_t)C * N * gs.geom_ld * 4 < (1ll << 32);
  for (int i = 0; i < cs.n; ++i) {
    small_tensors = small_tensors && (int64_t)C * N * cs.ld[i] * 4 < (1ll << 32);
    small_tensors = small_tensors && (int64_t)C * N * gs.ld[i] * 4 < (1ll << 32);
  }
  if (small_tensors && plan_mm(D, tile_size, geom_grad, (int64_t)C * tile_w * tile_h))
    return launch_bwd_mm(pick_chunk(D), tile_size, geom_grad != 0, any_color_grad, tuning().bwd_mfma == 6, C, N, D, means2d, conics, cs, opacities, backgrounds,
                         width, height, tile_w, tile_h, isect_offsets, tile_order, flatten_ids, n_isects,
                         render_alphas, last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas, absgrad,
                         v_geom, gs, (hipStream_t)stream);
  const BwdPlan plan = plan_bwd(D, geom_grad, any_color_grad, tile_size, (int64_t)C * tile_w * tile_h);
  // geometry gradients only, 16x16 tiles at 1 pixel per lane: the dot products go to the matrix cores
  if (geom_grad && !any_color_grad && tile_size == 16 && plan.ppl == 1 && tuning().bwd_mfma != 0) {  // (-1 included)
    const int cd = pick_chunk(D);
#define GEOM_MFMA_ARGS                                                                                             \
  C, N, D, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order, \
      flatten_ids, n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas, \
      absgrad, v_geom, gs.geom_ld, (hipStream_t)stream
    switch (cd) {
      case 16: return launch_bwd_geom_mfma<16>(GEOM_MFMA_ARGS);
      case 24: return launch_bwd_geom_mfma<24>(GEOM_MFMA_ARGS);
      case 32: return launch_bwd_geom_mfma<32>(GEOM_MFMA_ARGS);
      case 36: return launch_bwd_geom_mfma<36>(GEOM_MFMA_ARGS);
      case 48: return launch_bwd_geom_mfma<48>(GEOM_MFMA_ARGS);
      case 64: return launch_bwd_geom_mfma<64>(GEOM_MFMA_ARGS);
      case 72: return launch_bwd_geom_mfma<72>(GEOM_MFMA_ARGS);
      default: break;  // narrower or wider passes: the vector kernels below
    }
#undef GEOM_MFMA_ARGS
  }
  // geometry + colour gradients with both contractions on the matrix cores (1 pixel per lane)
  if (geom_grad && any_color_grad && tile_size == 16 && tuning().bwd_mfma == 2 && plan.per == D) {
    const int cd = pick_chunk(D);
#define FULL_MFMA_ARGS                                                                                             \
  C, N, D, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order, \
      flatten_ids, n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas, \
      absgrad, v_geom, gs, (hipStream_t)stream
    switch (cd) {
      case 32: return launch_bwd_full_mfma<32>(FULL_MFMA_ARGS);
      case 36: return launch_bwd_full_mfma<36>(FULL_MFMA_ARGS);
      case 64: return launch_bwd_full_mfma<64>(FULL_MFMA_ARGS);
      case 72: return launch_bwd_full_mfma<72>(FULL_MFMA_ARGS);
      default: break;
    }
#undef FULL_MFMA_ARGS
  }
  // geometry + colour gradients at 2 pixels per lane with the colour-gradient contraction on the matrix cores
  if (geom_grad && any_color_grad && tile_size == 16 && tuning().bwd_mfma == 3 && plan.per == D && plan.ppl == 2) {
    const int cd = pick_chunk(D);
    if (cd == 32 || cd == 36)
      return launch_bwd_cmfma(cd, C, N, D, means2d, conics, cs, opacities, backgrounds, width, height, tile_w,
                              tile_h, isect_offsets, tile_order, flatten_ids, n_isects, render_alphas, last_ids,
                              render_colors, ed_channel, v_render_colors, v_render_alphas, absgrad, v_geom, gs,
                              (hipStream_t)stream);
  }
  for (int off = 0; off < D; off += plan.per) {
    const int cd = pick_chunk(min(plan.per, D - off));
    int rc = PR_EUNSUPPORTED;
    if (tile_size == 8) {
      switch (cd) {
        BWD_CASE(8, 1, 4) BWD_CASE(8, 1, 8) BWD_CASE(8, 1, 16) BWD_CASE(8, 1, 24) BWD_CASE(8, 1, 32)
        BWD_CASE(8, 1, 36) BWD_CASE(8, 1, 48) BWD_CASE(8, 1, 64) BWD_CASE(8, 1, 72) BWD_CASE(8, 1, 100)
      }
    } else if (plan.ppl == 4) {
      switch (cd) {
        BWD_CASE(16, 4, 4) BWD_CASE(16, 4, 8) BWD_CASE(16, 4, 16) BWD_CASE(16, 4, 24) BWD_CASE(16, 4, 32)
        BWD_CASE(16, 4, 36)
      }
    } else if (plan.ppl == 2) {
      switch (cd) {
        BWD_CASE(16, 2, 4) BWD_CASE(16, 2, 8) BWD_CASE(16, 2, 16) BWD_CASE(16, 2, 24) BWD_CASE(16, 2, 32)
        BWD_CASE(16, 2, 36) BWD_CASE(16, 2, 48) BWD_CASE(16, 2, 64) BWD_CASE(16, 2, 72)
      }
    } else {
      switch (cd) {
        BWD_CASE(16, 1, 4) BWD_CASE(16, 1, 8) BWD_CASE(16, 1, 16) BWD_CASE(16, 1, 24) BWD_CASE(16, 1, 32)
        BWD_CASE(16, 1, 36) BWD_CASE(16, 1, 48) BWD_CASE(16, 1, 64) BWD_CASE(16, 1, 72) BWD_CASE(16, 1, 100)
      }
    }
    if (rc) return rc;
  }
  return PR_OK;
}
