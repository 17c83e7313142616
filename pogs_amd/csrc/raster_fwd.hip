// Forward compositing kernel for gfx950: per-tile front-to-back alpha blending.
#include <stdlib.h>

#include "raster_common.h"

namespace pr {

// Wavefronts per SIMD the register allocation is held to: the 2-pixel-per-lane kernel at 33..36 channels lands one
// register above the 128 that four wavefronts per SIMD allow unless it is told to stay below.
constexpr int fwd_waves_per_simd(int ppl, int cd) { return (ppl == 2 && cd > 32 && cd <= 36) ? 4 : 1; }

template <int TS, int PPL, int CD>
__global__ void __launch_bounds__(64 * WPB, fwd_waves_per_simd(PPL, CD)) raster_fwd_kernel(
    int n_regions, int N, int D, int ch_off0, int per, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds,
    int width, int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects, int ed_channel, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  using Rg = Region<TS, PPL>;
  constexpr int ROW = color_row(CD);
  constexpr int CDP = round4(CD);
  // 2 pixels per lane, narrow passes: 56-entry batches -- the largest whose slice (9.6 KB) still lets sixteen wavefronts
  // share a CU's LDS: 0.475 (32 entries) -> 0.465 (64, fourteen wavefronts) -> 0.454 ms
  constexpr int BS = (PPL == 2 && CD <= 36) ? 56 : batch_size(PPL);
  __shared__ WaveLds<CD, false, BS> lds[WPB];
  static_assert(WPB == 1, "one wavefront per workgroup");
  const int lane = threadIdx.x & 63;
  // the channel passes of a call are the rows of the grid: on a small image one pass alone leaves SIMDs idle
  const int ch_off = ch_off0 + (int)blockIdx.y * per;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  WaveLds<CD, false, BS> &L = lds[0];
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * TS + (lane % Rg::LX) * Rg::PX;
  const int y0 = ty * TS + sub * Rg::RH + (lane / Rg::LX) * Rg::PY;
  const int range_lo = isect_offsets[tile_flat];
  const int range_hi = (tile_flat + 1 < n_regions / Rg::WPT) ? isect_offsets[tile_flat + 1] : n_isects;

  constexpr int CH2 = CDP / 2;  // channel pairs
  // T: transmittance the pixel has reached; Tl: the same while the pixel still accepts Gaussians, 0 once it is
  // saturated or if it lies outside the image -- "is this pixel finished" is then arithmetic (a weight of
  // alpha * Tl = 0) rather than a flag carried per pixel through the loop.
  float T[PPL], Tl[PPL];
  f2 acc[PPL][CH2];
  int last[PPL];
  bool inside[PPL];
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int x = x0 + (j % Rg::BW) * Rg::SX, y = y0 + (j / Rg::BW) * Rg::SY;
    inside[j] = x < width && y < height;
    T[j] = 1.f;
    Tl[j] = inside[j] ? 1.f : 0.f;
    last[j] = range_lo;
#pragma unroll
    for (int k = 0; k < CH2; ++k) acc[j][k] = splat(0.f);
  }

  // pixel-centre rectangle of this wave's region, clipped to the image
  const float rx0 = (float)(tx * TS) + 0.5f, ry0 = (float)(ty * TS + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * TS + TS, width) - 0.5f;
  const float ry1 = (float)min(ty * TS + sub * Rg::RH + Rg::RH, height) - 0.5f;
  bool all_done = false;
  BatchGeometry pre = prefetch_batch(lane, range_lo, min(BS, range_hi - range_lo), flatten_ids, means2d, conics,
                                     opacities);
  for (int first = range_lo; first < range_hi && !all_done; first += BS) {
    unsigned long long todo = stage_batch<CD, false, BS>(L, lane, pre, cs, N, D, ch_off, rx0, ry0, rx1, ry1);
    // the next batch's geometry travels while this one is composited
    pre = prefetch_batch(lane, first + BS, min(BS, range_hi - first - BS), flatten_ids, means2d, conics,
                         opacities);
    while (todo) {
      const int t = __builtin_ctzll(todo);
      todo &= todo - 1;
      const float4 g = L.g4[t];
      const float2 h = L.g2[t];
      float vis[PPL];
      bool any = false;
      // shared partial products of the quadratic form over the lane's pixel block
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
      bool live = false;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int jx = j % Rg::BW, jy = j / Rg::BW;
        const float power = adx2[jx] + cdy2[jy] + bdx[jx] * dy[jy];  // = -sigma * log2(e)
        const float alpha = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
        const bool valid = (power <= 0.f) && (alpha >= ALPHA_MIN);
        const float next_T = Tl[j] - alpha * Tl[j];
        const bool contrib = valid && (next_T > T_STOP);   // never for a finished pixel: next_T = 0
        vis[j] = contrib ? alpha * Tl[j] : 0.f;
        Tl[j] = valid ? (contrib ? next_T : 0.f) : Tl[j];  // valid but not composited: the pixel saturates here
        T[j] = contrib ? next_T : T[j];
        last[j] = contrib ? first + t + 1 : last[j];
        any = any || contrib;
        live = live || Tl[j] > 0.f;
      }
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
        const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW);
#pragma unroll
        for (int q = 0; q < CDP / 4; ++q) {
          const float4 c = crow[q];
          const f2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
#pragma unroll
          for (int j = 0; j < PPL; ++j) {
            const f2 v = splat(vis[j]);
            acc[j][2 * q] = fma2(c01, v, acc[j][2 * q]);
            acc[j][2 * q + 1] = fma2(c23, v, acc[j][2 * q + 1]);
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
        all_done = true;
        break;
      }
    }
  }

  const int nch = min(CD, D - ch_off);
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    if (!inside[j]) continue;
    const int x = x0 + (j % Rg::BW) * Rg::SX, y = y0 + (j / Rg::BW) * Rg::SY;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    float *out = render_colors + pix * D + ch_off;
    if (backgrounds) {
      const float *bg = backgrounds + (int64_t)cam * D + ch_off;
#pragma unroll
      for (int k = 0; k < CD; ++k)
        if (k < nch) acc[j][k / 2][k % 2] = fmaf(T[j], bg[k], acc[j][k / 2][k % 2]);
    }
    if (ed_channel >= 0) {  // expected depth: accumulated depth / max(alpha, 1e-10)
      const float inv = 1.f / fmaxf(1.f - T[j], 1e-10f);
#pragma unroll
      for (int k = 0; k < CD; ++k)
        if (k + ch_off == ed_channel) acc[j][k / 2][k % 2] *= inv;
    }
    if (((D | ch_off) & 3) == 0) {
#pragma unroll
      for (int k = 0; k + 3 < CD; k += 4)
        if (k < nch)
          *reinterpret_cast<float4 *>(out + k) =
              make_float4(acc[j][k / 2][0], acc[j][k / 2][1], acc[j][k / 2 + 1][0], acc[j][k / 2 + 1][1]);
    } else {
#pragma unroll
      for (int k = 0; k < CD; ++k)
        if (k < nch) out[k] = acc[j][k / 2][k % 2];
    }
    if (ch_off == 0) {
      render_alphas[pix] = 1.f - T[j];
      last_ids[pix] = last[j];
    }
  }
}

// ---- matrix-core variant (16x16 tiles, 2 pixels per lane, 25..48 channels per pass) ------------------------
// The colour accumulation out[p][c] += vis[p][g] * colour[g][c] of 16 Gaussians at a time is one matrix
// product per 32-pixel group, out^T (32 channels x 32 pixels) += C^T (32 x 16) . V^T (16 x 32), on
// v_mfma_f32_32x32x16_bf16 with both operands split into a bf16 head and a bf16 residual
// (hi.hi + lo.hi + hi.lo: 16 mantissa bits, error <= 2^-16 per product, well inside the 1e-4 image tolerance).
// The weights need no data movement: a lane computes vis for ITS pixel and 16 Gaussians in a row, which is
// the B operand's register index; since lanes l and l+32 are different pixels here but the same matrix column
// in the operand layout (rows 8h .. 8h+7 on lane half h), one v_permlane32_swap per register pair turns
// "slots 0-7 / slots 8-15 of my pixel" into "operand of pixels 0-31 / operand of pixels 32-63".  The result has
// the pixel on the lane and the channels in the accumulator registers for the whole tile.  Channels past the
// first 32 of the pass (the expected-depth channel among them in the usual [rgb | features | depth] order)
// stay on the vector ALU in float32.
// PPL: pixels per lane (1: quarter tile per wavefront, 2: half tile); NM: 32-channel blocks on the matrix cores.
template <int PPL, int CD, int NM>
__global__ void __launch_bounds__(64) raster_fwd_mfma_kernel(
    int n_regions, int N, int D, int ch_off, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds,
    int width, int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects, int ed_channel,
    float *__restrict__ render_colors, float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  constexpr int GS = 16;                 // Gaussians per matrix product
  using Rg = Region<16, PPL>;
  constexpr int ROW = color_row(CD), CDP = round4(CD);
  // staged entries per batch: a full wavefront's worth at 2 pixels per lane (11-14 KB of LDS still leave the three
  // wavefronts per SIMD the registers allow; half as many staging round trips: 0.451 -> 0.436 ms), 32 otherwise
  constexpr int BS = PPL == 2 ? WAVE : batch_size(PPL);
  constexpr int CM = 32 * NM;            // channels on the matrix cores
  constexpr int CV = CDP - CM, CV2 = CV / 2;  // channels on the vector ALU
  static_assert(CD >= CM && CV >= 0 && CV <= 16, "pass width");
  __shared__ WaveLds<CD, false, BS> L;
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<Rg::WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * 16 + (lane % Rg::LX) * Rg::PX;
  const int y0 = ty * 16 + sub * Rg::RH + (lane / Rg::LX) * Rg::PY;
  const int range_lo = isect_offsets[tile_flat];
  const int range_hi = (tile_flat + 1 < n_regions / Rg::WPT) ? isect_offsets[tile_flat + 1] : n_isects;

  // T: transmittance the pixel has reached (what alpha and the background see); Tl: the same while the pixel
  // still accepts Gaussians, 0 once it is saturated or if it lies outside the image -- every test on "is this
  // pixel finished" is then arithmetic on Tl (a weight alpha * Tl of 0) instead of a flag carried per pixel.
  float T[PPL], Tl[PPL];
  f16v acc[PPL][2][NM];                  // [pixel slot][lane half of the pixel][channel block]: 32 ch x 32 px each
  f2 accv[PPL][CV2 > 0 ? CV2 : 1];
  int last[PPL];
  bool inside[PPL];
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    const int x = x0, y = y0 + j * Rg::SY;
    inside[j] = x < width && y < height;
    T[j] = 1.f;
    Tl[j] = inside[j] ? 1.f : 0.f;
    last[j] = range_lo;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][hh][mt][r] = 0.f;
#pragma unroll
    for (int k = 0; k < (CV2 > 0 ? CV2 : 1); ++k) accv[j][k] = splat(0.f);
  }

  const float rx0 = (float)(tx * 16) + 0.5f, ry0 = (float)(ty * 16 + sub * Rg::RH) + 0.5f;
  const float rx1 = (float)min(tx * 16 + 16, width) - 0.5f;
  const float ry1 = (float)min(ty * 16 + sub * Rg::RH + Rg::RH, height) - 0.5f;
  const int mc = lane & 31, mh = lane >> 5;   // this lane's row (channel) and k half in the A operand
  bool all_done = false;
  BatchGeometry pre = prefetch_batch(lane, range_lo, min(BS, range_hi - range_lo), flatten_ids, means2d, conics,
                                     opacities);
  for (int first = range_lo; first < range_hi && !all_done; first += BS) {
    unsigned long long todo = stage_batch<CD, false, BS>(L, lane, pre, cs, N, D, ch_off, rx0, ry0, rx1, ry1);
    pre = prefetch_batch(lane, first + BS, min(BS, range_hi - first - BS), flatten_ids, means2d, conics, opacities);
    while (todo) {
      unsigned wh[PPL][GS / 2], wl[PPL][GS / 2];  // packed weights: [slot][i] = Gaussians 2i, 2i+1 of the group
      float ca[NM][GS / 2];                       // A operands, this lane's k half: colour 32 mt + mc of members 8 mh + i
      bool any = false;
#pragma unroll
      for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int i = 0; i < GS / 2; ++i) ca[mt][i] = 0.f;
#pragma unroll
      for (int s = 0; s < GS; s += 2) {
        float w[PPL][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int j = 0; j < PPL; ++j) w[j][u] = 0.f;
          int t = -1;
          if (todo) {
            t = __builtin_ctzll(todo);
            todo &= todo - 1;
          }
          if (t < 0) continue;
#pragma unroll
          for (int mt = 0; mt < NM; ++mt) {  // lanes of half mh carry members 8 mh .. 8 mh + 7 of the group in the A operand
            const float c = L.col[t * ROW + 32 * mt + mc];
            ca[mt][(s + u) & 7] = (mh == ((s + u) >> 3)) ? c : ca[mt][(s + u) & 7];
          }
          const float4 g = L.g4[t];
          const float2 h = L.g2[t];
          const float dx = g.x - px0, adx2 = g.w * dx * dx, bdx = h.x * dx;
#pragma unroll
          for (int j = 0; j < PPL; ++j) {
            const float dy = g.y - (py0 + (float)(j * Rg::SY));
            const float power = adx2 + h.y * dy * dy + bdx * dy;  // = -sigma * log2(e)
            const float alpha = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
            const bool valid = (power <= 0.f) && (alpha >= ALPHA_MIN);
            const float next_T = Tl[j] - alpha * Tl[j];
            const bool contrib = valid && (next_T > T_STOP);   // never for a finished pixel: next_T = 0
            w[j][u] = contrib ? alpha * Tl[j] : 0.f;
            Tl[j] = valid ? (contrib ? next_T : 0.f) : Tl[j];  // valid but not composited = the pixel saturates here
            T[j] = contrib ? next_T : T[j];
            last[j] = contrib ? first + t + 1 : last[j];
            any = any || contrib;
          }
          if (CV > 0) {
            const float4 *crow = reinterpret_cast<const float4 *>(L.col + t * ROW + CM);
#pragma unroll
            for (int q = 0; q < CV / 4; ++q) {
              const float4 c = crow[q];
              const f2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
#pragma unroll
              for (int j = 0; j < PPL; ++j) {
                const f2 v = splat(w[j][u]);
                accv[j][2 * q] = fma2(c01, v, accv[j][2 * q]);
                accv[j][2 * q + 1] = fma2(c23, v, accv[j][2 * q + 1]);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          split_pair(w[j][0], w[j][1], wh[j][s / 2], wl[j][s / 2]);
          // convert here, not lazily at the matrix product: the float weights die and 16 registers are saved
          asm volatile("" : "+v"(wh[j][s / 2]), "+v"(wl[j][s / 2]));
        }
        // every pixel of the region is saturated: the rest of the group (and of the list) is not needed
        if ((s & 3) == 2) {
          bool live = false;
#pragma unroll
          for (int j = 0; j < PPL; ++j) live = live || Tl[j] > 0.f;
          if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
            todo = 0ull;
            all_done = true;
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
        // A operands: row = channel 32 mt + mc, k = group members 8 mh .. 8 mh + 7
        unsigned ah[NM][4], al[NM][4];
#pragma unroll
        for (int mt = 0; mt < NM; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) split_pair(ca[mt][2 * i], ca[mt][2 * i + 1], ah[mt][i], al[mt][i]);
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          unsigned b0h[4], b1h[4], b0l[4], b1l[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            swap32(wh[j][i], wh[j][4 + i], b0h[i], b1h[i]);
            swap32(wl[j][i], wl[j][4 + i], b0l[i], b1l[i]);
          }
#pragma unroll
          for (int mt = 0; mt < NM; ++mt) {
            acc[j][0][mt] = mfma_bf16(ah[mt], b0h, acc[j][0][mt]);
            acc[j][1][mt] = mfma_bf16(ah[mt], b1h, acc[j][1][mt]);
            acc[j][0][mt] = mfma_bf16(al[mt], b0h, acc[j][0][mt]);
            acc[j][1][mt] = mfma_bf16(al[mt], b1h, acc[j][1][mt]);
            acc[j][0][mt] = mfma_bf16(ah[mt], b0l, acc[j][0][mt]);
            acc[j][1][mt] = mfma_bf16(ah[mt], b1l, acc[j][1][mt]);
          }
        }
      }
      if (all_done) break;
    }
  }

  const int nch = min(CD, D - ch_off);
  const bool aligned = ((D | ch_off) & 3) == 0;
  // matrix part: accumulator [j][hh] of this lane belongs to the pixel of lane pl = 32 hh + (lane & 31)
#pragma unroll
  for (int j = 0; j < PPL; ++j)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int pl = 32 * hh + mc;
      const float Tp = __shfl(T[j], pl);
      const int x = tx * 16 + (pl % Rg::LX), y = ty * 16 + sub * Rg::RH + pl / Rg::LX + j * Rg::SY;
      if (x >= width || y >= height) continue;
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      float *out = render_colors + pix * D + ch_off;
      const float inv = 1.f / fmaxf(1.f - Tp, 1e-10f);
#pragma unroll
      for (int mq = 0; mq < 4 * NM; ++mq) {
        const int mt = mq / 4, q = mq % 4;
        const int c0 = 32 * mt + 8 * q + 4 * mh;  // channels c0 .. c0+3 of the pass sit in registers 4q .. 4q+3
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[j][hh][mt][4 * q + i];
          if (backgrounds && c0 + i < nch) v[i] = fmaf(Tp, backgrounds[(int64_t)cam * D + ch_off + c0 + i], v[i]);
          if (c0 + i + ch_off == ed_channel) v[i] *= inv;
        }
        if (aligned) {
          if (c0 < nch) *reinterpret_cast<float4 *>(out + c0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c0 + i < nch) out[c0 + i] = v[i];
        }
      }
    }
  // vector part and the per-pixel scalars: the pixel's own lane
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    if (!inside[j]) continue;
    const int x = x0, y = y0 + j * Rg::SY;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    float *out = render_colors + pix * D + ch_off;
    if (CV > 0) {
      const float inv = 1.f / fmaxf(1.f - T[j], 1e-10f);
#pragma unroll
      for (int k = 0; k < CV; ++k) {
        float v = accv[j][k / 2][k % 2];
        if (backgrounds && CM + k < nch) v = fmaf(T[j], backgrounds[(int64_t)cam * D + ch_off + CM + k], v);
        if (CM + k + ch_off == ed_channel) v *= inv;
        accv[j][k / 2][k % 2] = v;
      }
      if (aligned) {
#pragma unroll
        for (int k = 0; k < CV; k += 4)
          if (CM + k < nch)
            *reinterpret_cast<float4 *>(out + CM + k) =
                make_float4(accv[j][k / 2][0], accv[j][k / 2][1], accv[j][k / 2 + 1][0], accv[j][k / 2 + 1][1]);
      } else {
#pragma unroll
        for (int k = 0; k < CV; ++k)
          if (CM + k < nch) out[CM + k] = accv[j][k / 2][k % 2];
      }
    }
    if (ch_off == 0) {
      render_alphas[pix] = 1.f - T[j];
      last_ids[pix] = last[j];
    }
  }
}

template <int PPL, int CD, int NM>
static int launch_fwd_mfma(int C, int N, int D, int ch_off, const float *means2d, const float *conics,
                           const ColorSegs &cs, const float *opacities, const float *backgrounds, int width, int height,
                           int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                           const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors,
                           float *render_alphas, int32_t *last_ids, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<16, PPL>::WPT;
  const unsigned grid = region_grid(n_regions, Region<16, PPL>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_fwd_mfma_kernel<PPL, CD, NM>), dim3(grid), dim3(64), 0, stream, (int)n_regions, N, D,
                     ch_off, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,
                     tile_order, flatten_ids, (int)n_isects, ed_channel, render_colors, render_alphas, last_ids);
  return check_launch("raster_fwd_mfma_kernel");
}

template <int TS, int PPL, int CD>
static int launch_fwd(int C, int N, int D, int ch_off, int per, int n_pass, const float *means2d, const float *conics, const ColorSegs &cs,
                      const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                      const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                      int64_t n_isects, int ed_channel,
                      float *render_colors, float *render_alphas, int32_t *last_ids, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<TS, PPL>::WPT;
  const unsigned grid = region_grid(n_regions, Region<TS, PPL>::WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_fwd_kernel<TS, PPL, CD>), dim3(grid, n_pass), dim3(64 * WPB), 0, stream, (int)n_regions, N, D,
                     ch_off, per, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,
                     tile_order, flatten_ids, (int)n_isects, ed_channel, render_colors, render_alphas, last_ids);
  return check_launch("raster_fwd_kernel");
}

// Compiled (pixels-per-lane, channel-chunk) combinations for 16x16 tiles; 8x8 tiles always use one
// wavefront per tile (1 pixel per lane).  More pixels per lane amortise the per-Gaussian overhead better
// but need PPL * CD accumulator registers and give fewer wavefronts, so narrow renders of small images
// drop to fewer pixels per lane to keep the chip filled.
static const int kChunks[] = {4, 8, 16, 24, 32, 36, 48, 64, 72, 100};

static int pick_chunk(int want) {
  for (int c : kChunks)
    if (c >= want) return c;
  return -1;
}

struct FwdPlan {
  int ppl, per;
  bool blocks;  // the per-block-walk kernel (raster_fwd_blocks.hip): 2 pixels per lane, all passes in one launch
  bool ring;    // ... its form with independently advancing walks (else batch by batch)
};

// The per-block walks, batch by batch, are the planner's choice for small images (under 3072 tiles of 16x16) with 16
// channels or more (with 4 channels the vector kernel stays ahead: the tracker's RGB+ED render at 640x360, 0.090 against
// 0.099 ms): there the lists are long, the Gaussians small against a half tile, and a half tile's wavefront walks
// a third to a half fewer iterations.  Measured against the vector kernel's best pass width (kernel alone): 200k Gaussians
// at 640x360 with 64 channels 0.21 -> 0.18 ms; 1M at 450x250 with 64 channels 0.27 -> 0.21; 1M at 960x540 with 96
// channels 0.51 -> 0.50; at 1080p with 36 channels (8160 tiles) 0.46 against 0.46: no gain, the vector kernel stays.
// POGS_RAST_FWD_BLOCKS: unset = this rule, 0 = never, 1 = the ring form everywhere, 2 = batch by batch everywhere.
static bool plan_blocks(int D, int tile_size, int64_t n_tiles_total) {
  if (tile_size != 16 || tuning().fwd_ppl != 0) return false;
  if (tuning().fwd_blocks >= 1) return true;
  return tuning().fwd_blocks < 0 && tuning().fwd_mfma == 0 && n_tiles_total < 3072 && D >= 16;
}

// D channels -> chunk width and pixels per lane.  POGS_RAST_FWD_PPL / POGS_RAST_FWD_CHUNK override (tuning aid).
static FwdPlan plan_fwd(int D, int tile_size, int64_t n_tiles_total) {
  FwdPlan p;
  p.blocks = false;
  p.ring = tuning().fwd_blocks == 1;
  if (plan_blocks(D, tile_size, n_tiles_total)) {
    const int widest = n_tiles_total < 3072 ? 32 : 48;
    const int max_chunk = tuning().fwd_chunk > 0 ? min(tuning().fwd_chunk, 48) : widest;
    const int n_chunks = (D + max_chunk - 1) / max_chunk;
    p.per = ((D + n_chunks - 1) / n_chunks + 3) & ~3;
    p.ppl = 2;
    p.blocks = true;
    return p;
  }
  // Large images have tiles to spare: passes of <= 48 channels keep the accumulators of two pixels per lane within four
  // wavefronts per SIMD (alpha is recomputed per pass, which is cheaper than losing that).  A small image has too few
  // tiles to fill the chip with one wavefront per region, and every wavefront is one long dependent chain down its
  // tile's list: there the passes, which run side by side as rows of one launch, are made narrower until the launch
  // has about six thousand wavefronts -- <= 32 channels, or <= 16 on the smallest images.  Measured (forward kernel
  // alone, one pass of all channels -> this rule): 1M Gaussians at 960x540 with 96 channels 0.73 -> 0.51 ms; at 450x250
  // with 64 channels 0.37 -> 0.27 ms; 200k Gaussians at 640x360 with 64 channels 0.23 -> 0.23 ms.
  const bool many_tiles = tile_size == 16 && n_tiles_total >= 3072;
  int max_chunk = many_tiles ? 48 : 100;
  if (tile_size == 16 && !many_tiles && D > 32) {
    max_chunk = 32;
    const int64_t regions = n_tiles_total * (n_tiles_total < 1536 ? 4 : 2);  // wavefronts of one pass (see ppl below)
    if (regions * ((D + 31) / 32) < 6000) max_chunk = 16;
  }
  if (tuning().fwd_chunk > 0) max_chunk = tuning().fwd_chunk;
  const int n_chunks = (D + max_chunk - 1) / max_chunk;
  p.per = ((D + n_chunks - 1) / n_chunks + 3) & ~3;
  if (tile_size == 8) {
    p.ppl = 1;
    return p;
  }
  const int cd = pick_chunk(p.per);
  // 2 pixels per lane up to 72 channels: half the accumulators of 4 pixels per lane (<= 128 registers up to 36 channels:
  // four wavefronts per SIMD) and twice as many, shorter jobs, which matters more than the per-Gaussian work a second
  // wavefront per tile repeats.  Measured at 1M Gaussians, 1080p: 0.19 / 0.30 / 0.47 ms at 4 / 16 / 36 channels against
  // 0.24 / 0.34 / 0.53 at 4 pixels per lane and 0.21 / 0.34 / 0.54 at 1; the same order at 200k / 720p and 3M / 4K.
  p.ppl = cd <= 72 ? 2 : 1;
  while (p.ppl > 1 && n_tiles_total * (4 / p.ppl) < 3072) p.ppl >>= 1;
  const int forced = tuning().fwd_ppl;
  if (forced == 1 || (forced == 2 && cd <= 72) || (forced == 4 && cd <= 48)) p.ppl = forced;
  return p;
}

}  // namespace pr

using namespace pr;

extern "C" int pr_raster_fwd_plan(int D, int tile_size, int64_t n_tiles_total, int *pixels_per_lane,
                                  int *channels_per_pass, int *matrix_cores) {
  if (D <= 0 || D > pr_max_channels() || (tile_size != 16 && tile_size != 8) || n_tiles_total < 0) return PR_EINVAL;
  const FwdPlan plan = plan_fwd(D, tile_size, n_tiles_total);
  const int cd = pick_chunk(min(plan.per, D));
  if (pixels_per_lane) *pixels_per_lane = plan.ppl;
  if (channels_per_pass) *channels_per_pass = plan.per;
  if (matrix_cores)
    *matrix_cores = !plan.blocks && tuning().fwd_mfma != 0 && tile_size == 16 && cd >= 32 &&
                    ((plan.ppl == 2 && cd <= 48) || (plan.ppl == 1 && cd <= 72));
  return PR_OK;
}

#define FWD_ARGS                                                                                                    \
  C, N, D, off, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,          \
      tile_order, flatten_ids, n_isects, ed_channel, render_colors, render_alphas, last_ids, (hipStream_t)stream
#define FWD_ARGS_V                                                                                                  \
  C, N, D, off, plan.per, n_pass, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h,        \
      isect_offsets, tile_order, flatten_ids, n_isects, ed_channel, render_colors, render_alphas, last_ids,          \
      (hipStream_t)stream
#define FWD_CASE(TS, PPL, CDV) \
  case CDV:                    \
    rc = launch_fwd<TS, PPL, CDV>(FWD_ARGS_V); \
    break;

extern "C" int pr_raster_fwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                             const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                             int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                             const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors,
                             float *render_alphas, int32_t *last_ids, const pr_color_segment *segments,
                             int n_segments, void *stream) {
  if (C < 0 || N < 0 || D <= 0 || width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || n_isects < 0)
    return PR_EINVAL;
  if (tile_size != 16 && tile_size != 8) return PR_EUNSUPPORTED;
  if (D > pr_max_channels() || n_isects > 2147483647ll - 4096 /* list positions plus a batch of look-ahead stay inside int32 */) return PR_EUNSUPPORTED;
  if ((int64_t)tile_w * tile_size < width || (int64_t)tile_h * tile_size < height) return PR_EINVAL;
  if (ed_channel >= D) return PR_EINVAL;
  if (C == 0) return PR_OK;
  if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return PR_EINVAL;
  if (n_isects > 0 && (!means2d || !conics || !opacities || !flatten_ids)) return PR_EINVAL;
  if ((int64_t)C * tile_w * tile_h * 4 > 2147483647ll) return PR_EUNSUPPORTED;
  ColorSegs cs;
  int rcs = make_color_segs(colors, D, segments, n_segments, n_isects > 0, cs);
  if (rcs) return rcs;
  const FwdPlan plan = plan_fwd(D, tile_size, (int64_t)C * tile_w * tile_h);
  // Float32 accumulation (what gsplat's kernel does) unless the caller opts into the split-bf16 matrix-core
  // variant, whose error grows with the colour magnitude (about 1.2e-5 * max|colour|).
  const bool use_mfma = tuning().fwd_mfma != 0;
  if (plan.blocks)
    return launch_fwd_blocks(pick_chunk(min(plan.per, D)), plan.ring, C, N, D, 0, plan.per, (D + plan.per - 1) / plan.per, means2d,
                             conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order,
                             flatten_ids, n_isects, ed_channel, render_colors, render_alphas, last_ids,
                             (hipStream_t)stream);
  const int cd_all = pick_chunk(min(plan.per, D));
  const bool pass_mfma = use_mfma && tile_size == 16 && plan.ppl != 4 && cd_all >= 32 &&
                         ((plan.ppl == 2 && cd_all <= 48) || (plan.ppl == 1 && cd_all <= 72));
  // The vector kernels take all passes of the call in one launch (the passes are rows of the grid, all compiled for
  // the width of the first); the matrix-core variant goes pass by pass.
  for (int off = 0; off < D; off += pass_mfma ? plan.per : D) {
    const int cd = pass_mfma ? pick_chunk(min(plan.per, D - off)) : cd_all;
    const int n_pass = (D + plan.per - 1) / plan.per;
    int rc = PR_EUNSUPPORTED;
    if (tile_size == 8) {
      switch (cd) {
        FWD_CASE(8, 1, 4) FWD_CASE(8, 1, 8) FWD_CASE(8, 1, 16) FWD_CASE(8, 1, 24) FWD_CASE(8, 1, 32)
        FWD_CASE(8, 1, 36) FWD_CASE(8, 1, 48) FWD_CASE(8, 1, 64) FWD_CASE(8, 1, 72) FWD_CASE(8, 1, 100)
      }
    } else if (plan.ppl == 4) {
      switch (cd) {
        FWD_CASE(16, 4, 4) FWD_CASE(16, 4, 8) FWD_CASE(16, 4, 16) FWD_CASE(16, 4, 24) FWD_CASE(16, 4, 32)
        FWD_CASE(16, 4, 36) FWD_CASE(16, 4, 48)
      }
    } else if (use_mfma && cd >= 32 && ((plan.ppl == 2 && cd <= 48) || (plan.ppl == 1 && cd <= 72))) {
      // 32 or 64 channels of the pass on the matrix cores, the remaining 0..16 on the vector ALU
      if (plan.ppl == 2) {
        switch (cd) {
          case 32: rc = launch_fwd_mfma<2, 32, 1>(FWD_ARGS); break;
          case 36: rc = launch_fwd_mfma<2, 36, 1>(FWD_ARGS); break;
          case 48: rc = launch_fwd_mfma<2, 48, 1>(FWD_ARGS); break;
        }  // 64 channel blocks at 2 pixels per lane would need 128 accumulators + 160 registers: one wave per SIMD
      } else {
        switch (cd) {
          case 32: rc = launch_fwd_mfma<1, 32, 1>(FWD_ARGS); break;
          case 36: rc = launch_fwd_mfma<1, 36, 1>(FWD_ARGS); break;
          case 48: rc = launch_fwd_mfma<1, 48, 1>(FWD_ARGS); break;
          case 64: rc = launch_fwd_mfma<1, 64, 2>(FWD_ARGS); break;
          case 72: rc = launch_fwd_mfma<1, 72, 2>(FWD_ARGS); break;
        }
      }
    } else if (plan.ppl == 2) {
      switch (cd) {
        FWD_CASE(16, 2, 4) FWD_CASE(16, 2, 8) FWD_CASE(16, 2, 16) FWD_CASE(16, 2, 24) FWD_CASE(16, 2, 32)
        FWD_CASE(16, 2, 36) FWD_CASE(16, 2, 48) FWD_CASE(16, 2, 64) FWD_CASE(16, 2, 72)
      }
    } else {
      switch (cd) {
        FWD_CASE(16, 1, 4) FWD_CASE(16, 1, 8) FWD_CASE(16, 1, 16) FWD_CASE(16, 1, 24) FWD_CASE(16, 1, 32)
        FWD_CASE(16, 1, 36) FWD_CASE(16, 1, 48) FWD_CASE(16, 1, 64) FWD_CASE(16, 1, 72) FWD_CASE(16, 1, 100)
      }
    }
    if (rc) return rc;
  }
  return PR_OK;
}
