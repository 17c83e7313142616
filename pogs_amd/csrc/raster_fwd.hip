// Forward compositing kernel for gfx950: per-tile front-to-back alpha blending.
#include <stdlib.h>

#include "raster_common.h"

namespace pr {

template <int TS, int PPL, int CD>
__global__ void __launch_bounds__(64 * WPB) raster_fwd_kernel(
    int n_regions, int N, int D, int ch_off, const float *__restrict__ means2d, const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds,
    int width, int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects, int ed_channel, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  using Rg = Region<TS, PPL>;
  constexpr int ROW = color_row(CD);
  constexpr int CDP = round4(CD);
  constexpr int BS = batch_size(PPL);
  __shared__ WaveLds<CD, false, BS> lds[WPB];
  const int wave = WPB == 1 ? 0 : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // Without an explicit order, blocks that share an XCD take a contiguous range of tiles (neighbouring
  // tiles share Gaussians -> one L2).  With one, slot s processes tile tile_order[s]: the caller lists the
  // tiles heaviest first so the last waves of the launch are the cheapest ones.
  const unsigned bid = tile_order ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const int region = (int)bid * WPB + wave;
  if (region >= n_regions) return;
  WaveLds<CD, false, BS> &L = lds[wave];

  const int tile_flat = tile_order ? tile_order[region / Rg::WPT] : region / Rg::WPT, sub = region % Rg::WPT;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int x0 = tx * TS + (lane % Rg::LX) * Rg::PX;
  const int y0 = ty * TS + sub * Rg::RH + (lane / Rg::LX) * Rg::PY;
  const int range_lo = isect_offsets[tile_flat];
  const int range_hi = (tile_flat + 1 < n_regions / Rg::WPT) ? isect_offsets[tile_flat + 1] : n_isects;

  constexpr int CH2 = CDP / 2;  // channel pairs
  float T[PPL];
  f2 acc[PPL][CH2];
  int last[PPL];
  bool done[PPL], inside[PPL];
  const float px0 = (float)x0 + 0.5f, py0 = (float)y0 + 0.5f;
#pragma unroll
  for (int j = 0; j < PPL; ++j) {
    int x = x0 + (j % Rg::BW) * Rg::SX, y = y0 + (j / Rg::BW) * Rg::SY;
    inside[j] = x < width && y < height;
    done[j] = !inside[j];
    T[j] = 1.f;
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
      bool all = true;
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int jx = j % Rg::BW, jy = j / Rg::BW;
        const float power = adx2[jx] + cdy2[jy] + bdx[jx] * dy[jy];  // = -sigma * log2(e)
        const float alpha = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
        const bool valid = (power <= 0.f) && (alpha >= ALPHA_MIN) && !done[j];
        const float next_T = T[j] - alpha * T[j];
        const bool stop = valid && (next_T <= T_STOP);
        const bool contrib = valid && !stop;
        done[j] = done[j] || stop;
        vis[j] = contrib ? alpha * T[j] : 0.f;
        T[j] = contrib ? next_T : T[j];
        last[j] = contrib ? first + t + 1 : last[j];
        any = any || contrib;
        all = all && done[j];
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
      if (__builtin_amdgcn_ballot_w64(!all) == 0ull) {
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

template <int TS, int PPL, int CD>
static int launch_fwd(int C, int N, int D, int ch_off, const float *means2d, const float *conics, const ColorSegs &cs,
                      const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                      const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                      int64_t n_isects, int ed_channel,
                      float *render_colors, float *render_alphas, int32_t *last_ids, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * Region<TS, PPL>::WPT;
  const unsigned grid = (unsigned)((n_regions + WPB - 1) / WPB);
  hipLaunchKernelGGL((raster_fwd_kernel<TS, PPL, CD>), dim3(grid), dim3(64 * WPB), 0, stream, (int)n_regions, N, D, ch_off,
                     means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,
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

static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

struct FwdPlan {
  int ppl, per;
};

// D channels -> chunk width and pixels per lane.  POGS_RAST_FWD_PPL / POGS_RAST_FWD_CHUNK override (tuning aid).
static FwdPlan plan_fwd(int D, int tile_size, int64_t n_tiles_total) {
  FwdPlan p;
  // Large images have tiles to spare: channel chunks of <= 48 keep 4 pixels per lane (alpha is recomputed
  // per chunk, which is cheaper than losing the per-Gaussian amortisation).  Small images take all channels
  // in one pass with fewer pixels per lane, which also doubles or quadruples the number of wavefronts.
  const bool many_tiles = tile_size == 16 && n_tiles_total >= 3072;
  int max_chunk = env_int("POGS_RAST_FWD_CHUNK", many_tiles ? 48 : 100);
  const int n_chunks = (D + max_chunk - 1) / max_chunk;
  p.per = ((D + n_chunks - 1) / n_chunks + 3) & ~3;
  if (tile_size == 8) {
    p.ppl = 1;
    return p;
  }
  const int cd = pick_chunk(p.per);
  // Narrow chunks: 4 pixels per lane (few registers either way).  From 25 channels on, 2 pixels per lane: half
  // the accumulators -> <= 128 registers -> three to four wavefronts per SIMD instead of two, which matters
  // more than the per-Gaussian work a second wavefront per tile repeats (a wavefront issues at most one
  // vector instruction every four cycles, the SIMD takes one every two): 0.52 vs 0.57 ms at 36 channels.
  p.ppl = cd <= 24 ? 4 : cd <= 72 ? 2 : 1;
  while (p.ppl > 1 && n_tiles_total * (4 / p.ppl) < 3072) p.ppl >>= 1;
  const int forced = env_int("POGS_RAST_FWD_PPL", 0);
  if (forced == 1 || (forced == 2 && cd <= 72) || (forced == 4 && cd <= 48)) p.ppl = forced;
  return p;
}

}  // namespace pr

using namespace pr;

#define FWD_ARGS                                                                                                    \
  C, N, D, off, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,          \
      tile_order, flatten_ids, n_isects, ed_channel, render_colors, render_alphas, last_ids, (hipStream_t)stream
#define FWD_CASE(TS, PPL, CDV) \
  case CDV:                    \
    rc = launch_fwd<TS, PPL, CDV>(FWD_ARGS); \
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
  if (D > pr_max_channels() || n_isects > 2147483647ll) return PR_EUNSUPPORTED;
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
  for (int off = 0; off < D; off += plan.per) {
    const int cd = pick_chunk(min(plan.per, D - off));
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
