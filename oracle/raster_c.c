/*
 * C restatement (float32, scalar loops) of gsplat 1.4.0's tile compositor, forward and backward.
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see oracle/__init__.py): gsplat's source is not part of the
 * reference tree; this file restates the published per-pixel algorithm of its CUDA kernels
 * `rasterize_to_pixels_fwd` / `rasterize_to_pixels_bwd` (gsplat/cuda/csrc) that the reference reaches
 * through `rasterization()` at pogs/pogs.py:352-371,419-437,493-510,930-948.
 *
 * It is written the way the CUDA kernels are (explicit front-to-back loop with the skip and stop rules,
 * back-to-front replay with a per-channel "colour behind" buffer), i.e. differently from both the
 * vectorised PyTorch oracle (cumprod formulation) and the HIP kernels (scalar suffix-sum formulation), so
 * the three check one another.  It doubles as the CPU baseline of bench.py (OpenMP over tiles).
 *
 * Layouts: means2d[C*N,2] conics[C*N,3] colors[C*N,D] opacities[C*N] backgrounds[C,D] or NULL,
 * isect_offsets[C*th*tw] int32, flatten_ids[n_isects] int32, images [C,H,W,*].
 * `tiles` (may be NULL = all) lists the flat (camera*th*tw + tile) indices to process.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.999f
#define T_STOP 1e-4f

static void tile_bounds(int t, int tw, int th, int ts, int W, int H, int *cam, int *x0, int *y0, int *x1, int *y1) {
  int per = tw * th;
  *cam = t / per;
  int tile = t % per;
  *x0 = (tile % tw) * ts;
  *y0 = (tile / tw) * ts;
  *x1 = *x0 + ts < W ? *x0 + ts : W;
  *y1 = *y0 + ts < H ? *y0 + ts : H;
}

void oracle_raster_fwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                       const float *opacities, const float *backgrounds, int W, int H, int ts, int tw, int th,
                       const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                       const int32_t *tiles, int n_tiles_sel, float *render_colors, float *render_alphas,
                       int32_t *last_ids) {
  (void)N;
  int total = C * tw * th;
  int count = tiles ? n_tiles_sel : total;
#pragma omp parallel for schedule(dynamic, 4)
  for (int ti = 0; ti < count; ++ti) {
    int t = tiles ? tiles[ti] : ti;
    int cam, x0, y0, x1, y1;
    tile_bounds(t, tw, th, ts, W, H, &cam, &x0, &y0, &x1, &y1);
    int lo = isect_offsets[t];
    int hi = (t + 1 < total) ? isect_offsets[t + 1] : (int)n_isects;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        float px = (float)x + 0.5f, py = (float)y + 0.5f;
        int64_t pix = ((int64_t)cam * H + y) * W + x;
        float *out = render_colors + pix * D;
        for (int k = 0; k < D; ++k) out[k] = 0.f;
        float T = 1.f;
        int cur = lo; /* one past the last contributing intersection; `lo` when none */
        for (int i = lo; i < hi; ++i) {
          int g = flatten_ids[i];
          float dx = means2d[2 * (int64_t)g] - px, dy = means2d[2 * (int64_t)g + 1] - py;
          const float *cn = conics + 3 * (int64_t)g;
          float sigma = 0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
          float alpha = fminf(ALPHA_MAX, opacities[g] * expf(-sigma));
          if (sigma < 0.f || alpha < ALPHA_MIN) continue;
          float next_T = T * (1.f - alpha);
          if (next_T <= T_STOP) break;
          float vis = alpha * T;
          const float *c = colors + (int64_t)g * D;
          for (int k = 0; k < D; ++k) out[k] += c[k] * vis;
          cur = i + 1;
          T = next_T;
        }
        if (backgrounds)
          for (int k = 0; k < D; ++k) out[k] += T * backgrounds[(int64_t)cam * D + k];
        render_alphas[pix] = 1.f - T;
        last_ids[pix] = cur;
      }
  }
}

/* Gradients are ACCUMULATED into v_means2d[C*N,2], v_means2d_abs[C*N,2] (may be NULL), v_conics[C*N,3],
 * v_colors[C*N,D], v_opacities[C*N]; the caller zero-fills them. */
void oracle_raster_bwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                       const float *opacities, const float *backgrounds, int W, int H, int ts, int tw, int th,
                       const int32_t *isect_offsets, const int32_t *flatten_ids, int64_t n_isects,
                       const int32_t *tiles, int n_tiles_sel, const float *render_alphas, const int32_t *last_ids,
                       const float *v_render_colors, const float *v_render_alphas, float *v_means2d,
                       float *v_means2d_abs, float *v_conics, float *v_colors, float *v_opacities) {
  (void)N;
  int total = C * tw * th;
  int count = tiles ? n_tiles_sel : total;
#pragma omp parallel
  {
    float *buffer = (float *)malloc(sizeof(float) * (size_t)D);
    float *loc = NULL; /* per-tile accumulators: L x (D + 8) */
    size_t loc_cap = 0;
#pragma omp for schedule(dynamic, 4)
    for (int ti = 0; ti < count; ++ti) {
      int t = tiles ? tiles[ti] : ti;
      int cam, x0, y0, x1, y1;
      tile_bounds(t, tw, th, ts, W, H, &cam, &x0, &y0, &x1, &y1);
      int lo = isect_offsets[t];
      int hi = (t + 1 < total) ? isect_offsets[t + 1] : (int)n_isects;
      int L = hi - lo;
      if (L <= 0) continue;
      int S = D + 8;
      if ((size_t)L * S > loc_cap) {
        loc_cap = (size_t)L * S;
        free(loc);
        loc = (float *)malloc(sizeof(float) * loc_cap);
      }
      memset(loc, 0, sizeof(float) * (size_t)L * S);
      for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
          float px = (float)x + 0.5f, py = (float)y + 0.5f;
          int64_t pix = ((int64_t)cam * H + y) * W + x;
          const float *vo = v_render_colors + pix * D;
          float v_a_out = v_render_alphas[pix];
          float T_final = 1.f - render_alphas[pix];
          float T = T_final;
          int bin_final = last_ids[pix];
          for (int k = 0; k < D; ++k) buffer[k] = 0.f;
          float bg_dot = 0.f;
          if (backgrounds)
            for (int k = 0; k < D; ++k) bg_dot += backgrounds[(int64_t)cam * D + k] * vo[k];
          for (int i = bin_final - 1; i >= lo; --i) {
            int g = flatten_ids[i];
            float dx = means2d[2 * (int64_t)g] - px, dy = means2d[2 * (int64_t)g + 1] - py;
            const float *cn = conics + 3 * (int64_t)g;
            float sigma = 0.5f * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
            float vis = expf(-sigma);
            float opac = opacities[g];
            float alpha = fminf(ALPHA_MAX, opac * vis);
            if (sigma < 0.f || alpha < ALPHA_MIN) continue;
            float ra = 1.f / (1.f - alpha);
            T *= ra;
            float fac = alpha * T;
            const float *c = colors + (int64_t)g * D;
            float *acc = loc + (size_t)(i - lo) * S;
            float v_alpha = 0.f;
            for (int k = 0; k < D; ++k) {
              acc[k] += fac * vo[k];
              v_alpha += (c[k] * T - buffer[k] * ra) * vo[k];
            }
            v_alpha += T_final * ra * v_a_out;
            if (backgrounds) v_alpha += -T_final * ra * bg_dot;
            if (opac * vis <= ALPHA_MAX) {
              float v_sigma = -opac * vis * v_alpha;
              acc[D + 0] += 0.5f * v_sigma * dx * dx;
              acc[D + 1] += v_sigma * dx * dy;
              acc[D + 2] += 0.5f * v_sigma * dy * dy;
              float vx = v_sigma * (cn[0] * dx + cn[1] * dy);
              float vy = v_sigma * (cn[1] * dx + cn[2] * dy);
              acc[D + 3] += vx;
              acc[D + 4] += vy;
              acc[D + 5] += fabsf(vx);
              acc[D + 6] += fabsf(vy);
              acc[D + 7] += vis * v_alpha;
            }
            for (int k = 0; k < D; ++k) buffer[k] += c[k] * fac;
          }
        }
      for (int j = 0; j < L; ++j) {
        int g = flatten_ids[lo + j];
        const float *acc = loc + (size_t)j * S;
        for (int k = 0; k < D; ++k) {
#pragma omp atomic
          v_colors[(int64_t)g * D + k] += acc[k];
        }
        for (int k = 0; k < 3; ++k) {
#pragma omp atomic
          v_conics[3 * (int64_t)g + k] += acc[D + k];
        }
        for (int k = 0; k < 2; ++k) {
#pragma omp atomic
          v_means2d[2 * (int64_t)g + k] += acc[D + 3 + k];
        }
        if (v_means2d_abs)
          for (int k = 0; k < 2; ++k) {
#pragma omp atomic
            v_means2d_abs[2 * (int64_t)g + k] += acc[D + 5 + k];
          }
#pragma omp atomic
        v_opacities[g] += acc[D + 7];
      }
    }
    free(buffer);
    free(loc);
  }
}
