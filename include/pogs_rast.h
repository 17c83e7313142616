/*
 * pogs_rast.h -- C ABI of the MI355X (gfx950) Gaussian-splat rasterizer library, libpogs_rast.so.
 *
 * The library is the device half of the drop-in for `gsplat.rendering.rasterization()`, the call the
 * reference makes at pogs/pogs.py:352-371, :419-437, :493-510, :930-948 and (through get_outputs) from
 * pogs/tracking/rigid_group_optimizer.py:299,314,545.  gsplat itself is a third-party dependency that is
 * not part of the reference tree (pinned gsplat==1.4.0 at
 * src/pogs/dependencies/nerfstudio/pyproject.toml:67); each entry point below names the gsplat 1.4.0
 * routine (`gsplat/cuda/_wrapper.py`) whose CUDA extension call it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - every array is dense, row-major, float32 unless typed otherwise;
 *   - the caller owns every buffer (inputs, outputs, scratch); the library allocates nothing persistent;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises the device
 *     except where documented;
 *   - return value: 0 on success, a negative PR_E* code otherwise (never throws, never aborts);
 *     pr_error_string() maps a code to text;
 *   - N = Gaussians, C = cameras, flat index = c*N + n, D = composited channels.
 */
#ifndef POGS_RAST_H
#define POGS_RAST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PR_OK 0
#define PR_EINVAL (-1)      /* bad argument (null pointer, negative size, unsupported value) */
#define PR_EUNSUPPORTED (-2)/* valid request the library has no kernel for (e.g. channel count too large) */
#define PR_ESCRATCH (-3)    /* scratch buffer too small */
#define PR_EHIP (-4)        /* a HIP runtime call or launch failed; hipGetLastError text via pr_last_hip_error */

/* Library/ABI version, (major<<16)|minor. */
int pr_version(void);
const char *pr_error_string(int code);
/* Text of the last HIP error seen by the calling thread (empty string if none). Host pointer. */
const char *pr_last_hip_error(void);
/* Largest channel count pr_raster_fwd / pr_raster_bwd accept in one call. */
int pr_max_channels(void);
/* Kernel-variant overrides (environment variables POGS_RAST_FWD_PPL, POGS_RAST_FWD_CHUNK, POGS_RAST_FWD_MFMA,
 * POGS_RAST_BWD_PPL, POGS_RAST_BWD_MFMA, POGS_MLP_VARIANT) are read once, at the first call that needs them;
 * this re-reads them.  For tests and A/B measurements that switch variants inside one process; not to be called
 * while another thread is inside the library.  No gsplat counterpart. */
void pr_reload_tuning(void);

/* ---- projection: stands in for gsplat `fully_fused_projection` forward (non-packed) --------------------
 * means[N,3] quats[N,4] (wxyz, unnormalised) scales[N,3] viewmats[C,4,4] (world->camera) Ks[C,3,3].
 * Out: radii[C,N] int32 (0 = culled), means2d[C,N,2], depths[C,N], conics[C,N,3],
 *      compensations[C,N] (may be NULL: not computed).  Culled entries are written as zeros. */
int pr_project_fwd(int C, int N, const float *means, const float *quats, const float *scales,
                   const float *viewmats, const float *Ks, int width, int height, float eps2d,
                   float near_plane, float far_plane, float radius_clip, int32_t *radii, float *means2d,
                   float *depths, float *conics, float *compensations, void *stream);

/* ---- projection backward: stands in for gsplat `fully_fused_projection` backward ------------------------
 * v_means2d[C,N,*] with row stride ld_means2d floats, v_conics[C,N,*] with row stride ld_conics,
 * v_depths[C,N] (may be NULL), v_compensations[C,N] (may be NULL).
 * Out (overwritten): v_means[N,3], v_quats[N,4], v_scales[N,3] -- summed over cameras.
 * v_viewmats[C,4,4] may be NULL; otherwise it must be zero-filled by the caller and rows 0..2 are
 * accumulated into with float atomics. */
int pr_project_bwd(int C, int N, const float *means, const float *quats, const float *scales,
                   const float *viewmats, const float *Ks, int width, int height, float eps2d,
                   const int32_t *radii, const float *v_means2d, int ld_means2d, const float *v_depths,
                   const float *v_conics, int ld_conics, const float *v_compensations, float *v_means,
                   float *v_quats, float *v_scales, float *v_viewmats, void *stream);

/* ---- spherical harmonics: stands in for gsplat `spherical_harmonics` forward/backward -------------------
 * View directions are means[n] - campos[c] (campos[C,3] = camera centres in world coordinates).
 * coeffs[N,K,3]; degree in 0..3 with (degree+1)^2 <= K.  Entries with radii[c,n] <= 0 are masked.
 * post != 0 fuses the caller's `clamp_min(x + 0.5, 0)` (pogs colour convention) into the kernel.
 * colors out: [C,N,3] with row stride ld_colors floats. */
int pr_sh_fwd(int C, int N, int K, int degree, const float *means, const float *campos,
              const float *coeffs, const int32_t *radii, int post, float *colors, int ld_colors,
              void *stream);
/* Out (overwritten): v_coeffs[N,K,3] (summed over cameras; zero for inactive bands),
 * v_means[N,3] (gradient through the view direction, summed over cameras; may be NULL),
 * v_campos[C,3] (may be NULL; otherwise zero-filled by the caller, accumulated with atomics). */
int pr_sh_bwd(int C, int N, int K, int degree, const float *means, const float *campos,
              const float *coeffs, const int32_t *radii, int post, const float *v_colors, int ld_colors,
              float *v_coeffs, float *v_means, float *v_campos, void *stream);

/* Camera centres in world coordinates, campos[C,3] = inverse(viewmats)[:, :3, 3], for rigid world->camera matrices
 * (closed form -R^T t), and the cotangent of viewmats [C,4,4] (overwritten) as the derivative of the general matrix
 * inverse gives it -- what gsplat's `rasterization` obtains from `torch.inverse(viewmats)` when it forms the SH view
 * directions. */
/* Camera-space depth of EVERY Gaussian, culled or not: depths[c,n] = (viewmats[c] * means[n]).z, bit-identical to what
 * pr_project_fwd writes for the Gaussians it keeps (it writes 0 for the ones it culls, as gsplat reports them).  A
 * depth order built from these (pr_depth_order) depends on positions and view matrices alone, so one order serves
 * every render of the same model through the same view matrix, whatever its intrinsics, image size or channels (POGS
 * renders three per training iteration, pogs/pogs.py:352,419,493).  No counterpart in gsplat. */
int pr_view_depths(int C, int N, const float *means, const float *viewmats, float *depths, void *stream);
int pr_campos_fwd(int C, const float *viewmats, float *campos, void *stream);
int pr_campos_bwd(int C, const float *viewmats, const float *v_campos, float *v_viewmats, void *stream);

/* ---- tile binning: stands in for gsplat `isect_tiles` + `isect_offset_encode` --------------------------
 * Phase 1: tiles_per_gauss[C,N] int32 and its inclusive prefix sum cum_tiles[C*N] int64 (may be NULL with
 * scratch NULL: counts only).  The caller reads cum_tiles[C*N-1] back to the host to learn n_isects (the only
 * D2H of the path), then calls phase 2 with buffers of that size.
 * conics[C,N,3] / opacities[C,N] (both NULL or both given) select which pairs are binned:
 *   NULL  -- gsplat's rule: every tile the square 3-sigma bounding box of the Gaussian touches;
 *   given -- of those, only the tiles in which the Gaussian can reach alpha >= 1/255 at some pixel centre.
 *            The compositor ignores the other pairs pixel by pixel, so rendered values and gradients are the
 *            same either way; the lists (and gsplat's `tiles_per_gauss`, `isect_ids`, `flatten_ids`,
 *            `isect_offsets`) are shorter.  Phase 2 must be given the same conics/opacities. */
size_t pr_isect_count_scratch_bytes(int C, int N);
int pr_isect_count(int C, int N, const float *means2d, const int32_t *radii, const float *conics,
                   const float *opacities, int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss,
                   int64_t *cum_tiles, void *scratch, size_t scratch_bytes, void *stream);
/* Phase 2: key = cam<<(32+tile_bits) | tile<<32 | float_bits(depth), value = flat index, in stable
 * ascending key order (ties by ascending flat index).  tiles_per_gauss is phase 1's output.
 * Out: flatten_ids[n_isects] int32; isect_ids[n_isects] int64 (may be NULL: the 64-bit keys are only
 * materialised for callers that want gsplat's `isect_ids`); isect_offsets[C,tile_h,tile_w] int32 (may be
 * NULL), the same table pr_isect_offsets derives from isect_ids, written here straight from the sorted
 * tile keys; tile_order[C*tile_h*tile_w] int32 (may be NULL; needs isect_offsets), the flat tile indices
 * ordered by list length, longest first up to a bucket of 1/1024 of the longest list -- the launch order
 * pr_raster_fwd/bwd accept.  n_tiles_total = C*tile_h*tile_w when tile_order is wanted, else 0 (reserved:
 * the order needs no scratch at present).
 * n_isects_dev (may be NULL): device pointer to the true intersection count (cum_tiles + C*N - 1).  With
 * it, n_isects is only the CAPACITY of the buffers -- a caller that does not want to wait for the count
 * to reach the host sizes them from a guess, launches, and compares afterwards: if the true count is
 * larger than the capacity the outputs are incomplete (never out of bounds) and the call must be repeated.
 * Entries of flatten_ids / isect_ids past the true count are unspecified.
 * depth_order (may be NULL): the output of pr_depth_order; without it the depth sort is done here. */
size_t pr_isect_sort_scratch_bytes(int C, int N, int64_t n_isects, int64_t n_tiles_total);
int pr_isect_emit_sort(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                       const float *conics, const float *opacities, const int32_t *tiles_per_gauss, int tile_size, int tile_w, int tile_h, int64_t n_isects,
                       int64_t *isect_ids, int32_t *flatten_ids, int32_t *isect_offsets, int32_t *tile_order,
                       const int64_t *n_isects_dev, const int32_t *depth_order, void *scratch,
                       size_t scratch_bytes, void *stream);
/* The count of phase 1 for a host that does not want to wait for it: a one-thread kernel stores *count_dev into
 * host_words[0] and then `sequence` into host_words[1] (page-locked, device-visible host memory; system-scope
 * stores, the second one a release).  The caller launches phase 2 in capacity mode, then polls host_words[1] for
 * its sequence number and reads host_words[0].  Takes the place of gsplat's blocking `cum_tiles[-1].item()`
 * (gsplat/cuda/_wrapper.py `isect_tiles`, csrc isect_tiles) without a copy command or an event on the stream. */
int pr_publish_count(const int64_t *count_dev, int64_t *host_words, int64_t sequence, void *stream);
/* The compositor's lists in two calls, without a device-wide sort: what `isect_tiles` + `isect_offset_encode` produce
 * for the pairs that can contribute (conics/opacities required: the footprint-tested rule of pr_isect_count), in the
 * same order (camera, tile, depth bits, flat index), but WITHOUT gsplat's 64-bit keys.
 * pr_tile_lists_count: tile_counters[C*tile_h*tile_w] int32 is working memory carried to pr_tile_lists_fill;
 *   out: isect_offsets[C,tile_h,tile_w], *n_isects_dev (device int64: the total, to be read back by the caller to size
 *   the lists), tile_order (may be NULL; as in pr_isect_emit_sort), tiles_per_gauss[C,N] (may be NULL).
 * pr_tile_lists_fill: capacity = entries flatten_ids and keys_scratch (uint64 each) hold; a caller that sized them
 *   from a guess compares *n_isects_dev with it afterwards -- when the total is larger the lists are incomplete
 *   (never out of bounds) and both calls must be repeated.  The same means2d/radii/conics/opacities must be given
 *   to both calls.  No gsplat counterpart as separate routines. */
int pr_tile_lists_count(int C, int N, const float *means2d, const int32_t *radii, const float *conics,
                        const float *opacities, int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss,
                        int32_t *tile_counters, int32_t *isect_offsets, int32_t *tile_order, int64_t *n_isects_dev,
                        void *stream);
int pr_tile_lists_fill(int C, int N, const float *means2d, const int32_t *radii, const float *depths,
                       const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                       const int32_t *isect_offsets, int32_t *tile_counters, int64_t capacity,
                       const int64_t *n_isects_dev, uint64_t *keys_scratch, int32_t *flatten_ids, void *stream);
/* depth_order[C*N] int32: the flat indices c*N+n in stable ascending order of the depths' float bits (ties by
 * ascending index) -- the first level of phase 2's two-level sort, split out because it depends on the
 * projection only: a caller may run it on another stream while colours are evaluated and tiles counted. */
size_t pr_depth_order_scratch_bytes(int C, int N);
int pr_depth_order(int C, int N, const float *depths, int32_t *depth_order, void *scratch, size_t scratch_bytes,
                   void *stream);
/* pr_depth_order for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference (gsplat allocates its sort buffers through torch).
 * Same result as pr_depth_order, but the kernels' scratch (sort keys, index sequence, the radix sort's temporary
 * storage) lives in a 24 MiB device array inside the library instead of a caller-supplied buffer; the call
 * allocates and frees nothing.  Limit: C*N <= pr_depth_order_static_max() (about 1.25 million on MI355X: the 1M-Gaussian training shape fits); larger inputs
 * return PR_EUNSUPPORTED with the reason in pr_last_hip_error().  Every call on one GPU uses that GPU's one array:
 * calls that overlap on one GPU (two streams, two threads) race -- serialise them or use pr_depth_order.
 * The FIRST call on each device looks the array's address up (hipGetSymbolAddress, under a lock): make that call
 * outside stream capture, e.g. once at start-up with N = 1; later calls on that device only launch kernels.  With
 * POGS_RAST_DEPTH_SORT=2 (the sample sort, a tuning aid) the call also sets a kernel attribute and must not be
 * captured at all. */
int64_t pr_depth_order_static_max(void);
int pr_depth_order_static(int C, int N, const float *depths, int32_t *depth_order, void *stream);
/* isect_offsets[C,tile_h,tile_w] int32: index of the first intersection of every (camera, tile). */
int pr_isect_offsets(int64_t n_isects, const int64_t *isect_ids, int C, int tile_w, int tile_h,
                     int32_t *isect_offsets, void *stream);

/* The D composited channels may be supplied as up to three tensors laid side by side in channel order instead
 * of one [C,N,D] tensor (e.g. SH colours [C,N,3] | per-Gaussian features [N,F] | depths [C,N]): segment i holds
 * `width` channels in rows of pitch `ld` floats; shared != 0 means one row per Gaussian for all cameras (row
 * index n), otherwise the row index is c*N + n.  Widths add up to D.  No gsplat counterpart (gsplat
 * concatenates into one tensor, pogs/pogs.py passes the result). */
typedef struct {
  const float *ptr;
  int width, ld, shared;
} pr_color_segment;
/* Gradient destinations in the same layout; ptr NULL = no gradient wanted for that segment.  A shared segment
 * accumulates the contributions of all cameras.  Zero-filled by the caller. */
typedef struct {
  float *ptr;
  int width, ld, shared;
} pr_grad_segment;

/* ---- compositing: stands in for gsplat `rasterize_to_pixels` forward/backward --------------------------
 * means2d[C,N,2] conics[C,N,3] colors[C,N,D] opacities[C,N] backgrounds[C,D] (may be NULL).
 * ed_channel >= 0 names the channel that holds depth and is to be output as EXPECTED depth, i.e. divided by
 * max(alpha, 1e-10) (what gsplat's `rasterization` does after the kernel for render modes ED / RGB+ED);
 * -1 = none.
 * segments/n_segments (NULL/0 = use `colors`): the channels gathered from up to three tensors, see
 * pr_color_segment; `colors` may then be NULL.  pr_raster_bwd scatters the colour gradients into v_segments
 * (same count and widths) instead of v_colors.
 * tile_order (may be NULL) is a permutation of the C*tile_h*tile_w flat tile indices: the order in which
 * wavefronts pick tiles up.  Results do not depend on it; listing the longest tiles first shortens the tail
 * of the launch.
 * Out: render_colors[C,H,W,D], render_alphas[C,H,W], last_ids[C,H,W] int32 (one past the index of the
 * last intersection that contributed to the pixel; the tile's first index when none did). */
int pr_raster_fwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                  const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                  int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                  const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors, float *render_alphas,
                  int32_t *last_ids, const pr_color_segment *segments, int n_segments, void *stream);
/* v_render_colors[C,H,W,D], v_render_alphas[C,H,W]; render_colors (the forward output) is read only when
 * ed_channel >= 0 and may be NULL otherwise.
 * Out, all zero-filled by the caller and accumulated into with float atomics:
 *   v_geom[C*N,v_geom_ld], columns 0-7 = (v_means2d.x, v_means2d.y, v_conic.a, v_conic.b, v_conic.c, v_opacity,
 *                    |v_means2d|.x, |v_means2d|.y)  -- the last two only when absgrad != 0; v_geom_ld (0 = 8) is the
 *                    row pitch in floats: a caller that interleaves the geometry gradients with narrow colour gradients
 *                    in rows of 16 floats (v_segments pointing into the same rows) gets all of a Gaussian's narrow
 *                    values delivered as ONE 64-byte atomic request instead of one per tensor;
 *   v_colors[C*N,D] (may be NULL: colour gradients not needed).
 * geom_grad == 0 skips every geometry gradient (v_geom may then be NULL). */
int pr_raster_bwd(int C, int N, int D, const float *means2d, const float *conics, const float *colors,
                  const float *opacities, const float *backgrounds, int width, int height, int tile_size,
                  int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                  const int32_t *flatten_ids, int64_t n_isects, const float *render_alphas, const int32_t *last_ids,
                  const float *render_colors, int ed_channel, const float *v_render_colors,
                  const float *v_render_alphas, int absgrad, int geom_grad, float *v_geom, int v_geom_ld,
                  float *v_colors, const pr_color_segment *segments, const pr_grad_segment *v_segments,
                  int n_segments, void *stream);

/* Which kernel variant pr_raster_fwd / pr_raster_bwd pick for a D-channel call over n_tiles_total = C*tile_h*tile_w
 * tiles: pixels per lane (4: one wavefront per 16x16 tile, 2: two, 1: four; always 1 for 8x8 tiles), channels per
 * pass (wider renders take ceil(D / channels_per_pass) passes), and whether the pass runs its channel contraction on
 * the matrix cores.  Any out pointer may be NULL.  For tests, benchmarks and profiles that must name the kernel
 * they exercised.  No gsplat counterpart. */
int pr_raster_fwd_plan(int D, int tile_size, int64_t n_tiles_total, int *pixels_per_lane, int *channels_per_pass,
                       int *matrix_cores);
int pr_raster_bwd_plan(int D, int tile_size, int64_t n_tiles_total, int geom_grad, int color_grad,
                       int *pixels_per_lane, int *channels_per_pass, int *matrix_cores);

/* ---- the tracker's per-iteration loss (value and gradients in one kernel each) ----
 * Replaces the element-wise / reduction chain of get_optim_loss (pogs/tracking/rigid_group_optimizer.py:364, 373-386):
 *   loss = nanmean_p ||target[p,:] - rendered[p,:]||_2
 *        + mult * mean over kept p of (clamp(depth[p] * inv_scale, 1e-8, 1) - clamp(target_depth[p], 1e-8, 1))^2
 * rendered, target [P,D] (D a multiple of 4, rows 16-byte aligned); depth, target_depth [P] or both NULL (no depth
 * term); valid [P] bytes or NULL (all): pixels that may enter the depth term; kept = valid and squared error < cut2.
 * sums[4] (device, written here): sum of finite norms, their number, sum of kept squared errors, their number;
 * loss[1] (device).  Without a kept pixel the depth term is left out, as the reference leaves out a NaN mean. */
int pr_track_loss_fwd(int64_t P, int D, const float *rendered, const float *target, const float *depth,
                      const float *target_depth, const uint8_t *valid, float inv_scale, float cut2, float mult,
                      float *sums, float *loss, void *stream);
/* Gradients for the cotangent v_loss[1] (device) of the loss: v_rendered[P,D] and, unless NULL, v_depth[P], both
 * overwritten.  sums as pr_track_loss_fwd left it. */
int pr_track_loss_bwd(int64_t P, int D, const float *rendered, const float *target, const float *depth,
                      const float *target_depth, const uint8_t *valid, float inv_scale, float cut2, float mult,
                      const float *sums, const float *v_loss, float *v_rendered, float *v_depth, void *stream);

/* Depth masks of the tracker's per-object crops (pogs/tracking/rigid_group_optimizer.py:316-337), all crops of a pass at
 * once: crop g occupies the top-left sizes[g] = (h, w) window of image g of a padded [G, Hm, Wm] batch.
 *   valid = accumulation > acc_min (0.85) and observed_depth is not NaN; eroded 5x5;
 *   pixels whose observed depth exceeds reject_factor (1.3) x the crop's median over that mask are dropped
 *   (torch.median: the lower middle value; a crop without a valid pixel drops nothing);
 *   keep = 9x9 erosion of (mask and observed_depth != 0); masked_depth = keep ? observed_depth : 0.
 * Erosions ignore what lies beyond a crop.  scratch: pr_roi_depth_masks_scratch_bytes(G, Hm, Wm) bytes, 16-byte aligned;
 * medians[G] receives the medians (NaN: none). */
size_t pr_roi_depth_masks_scratch_bytes(int G, int Hm, int Wm);
int pr_roi_depth_masks(int G, int Hm, int Wm, const int32_t *sizes, const float *accumulation,
                       const float *observed_depth, float acc_min, float reject_factor, uint8_t *scratch,
                       float *medians, uint8_t *keep, float *masked_depth, void *stream);

/* ---- rigid-group transform of the Gaussian model (the tracker's per-iteration pose update) ----
 * Replaces the Warp kernel apply_to_model (pogs/tracking/utils.py:53-93, launched at
 * pogs/tracking/rigid_group_optimizer.py:466-477) and its wp.Tape backward (:413-430).
 * init_p2ws[G,7], p_deltas[G,7]: poses as (x y z qw qx qy qz); labels[N] int32 in [0,G) (a label outside
 * that range leaves the Gaussian where it is and contributes no gradient); means[N,3]; quats[N,4] (w x y z).
 *   (means_out[n], quats_out[n]) = p2w[g] * delta[g] * inverse(p2w[g]) * (means[n], quats[n])
 * with Warp's transform algebra (quaternions are taken as given, not renormalised). */
int pr_rigid_fwd(int64_t N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                 const float *means, const float *quats, float *means_out, float *quats_out, void *stream);
/* Adjoint with respect to p_deltas only (the tracker's one trainable tensor). v_means_out[N,3];
 * v_quats_out[N,4] in (w x y z) order, or NULL when the cotangent of quats_out is zero -- the reference
 * allocates quats_out with requires_grad=False (rigid_group_optimizer.py:460-462), so its tape only ever
 * sees the means' cotangent. v_p_deltas[G,7] is zero-filled by the caller and accumulated into. */
int pr_rigid_bwd(int64_t N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                 const float *means, const float *quats, const float *v_means_out, const float *v_quats_out,
                 float *v_p_deltas, void *stream);

/* One iteration of the tracker's pose optimiser (pogs/tracking/rigid_group_optimizer.py:432 part_optimizer.step(),
 * a torch.optim.Adam on the [G,7] deltas, and :411-412, the renormalisation the next iteration opens with):
 * deltas, grad, exp_avg, exp_avg_sq: [G,7] (x y z qw qx qy qz); step >= 1 is Adam's step count after this update; lr the
 * scheduled rate of this iteration.  Update as pr_adam_step (no weight decay, no amsgrad); renormalise != 0 then divides
 * each group's quaternion part by its norm. */
int pr_pose_step(int G, float *deltas, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t step, double lr,
                 double beta1, double beta2, double eps, int renormalise, void *stream);

/* The same transform applied INSIDE the projection, for the tracker's loop (frozen model, pose deltas trainable;
 * pogs/tracking/rigid_group_optimizer.py:402-479): pr_project_fwd_rigid is pr_project_fwd on the moved Gaussians
 * without writing them out, except means_world[N,3] (may be NULL) which the view-dependent colours need;
 * pr_project_bwd_rigid is pr_project_bwd whose mean cotangent -- plus v_means_world[N,3] (may be NULL), the cotangent
 * that reached means_world from elsewhere -- goes straight on to v_p_deltas[G,7] (zero-filled by the caller,
 * accumulated).  As in the reference, whose tape holds the moved means only, the moved quaternions pass no gradient
 * on; the model's own means/quats/scales receive none. */
int pr_project_fwd_rigid(int C, int N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                         const float *means, const float *quats, const float *scales, const float *viewmats,
                         const float *Ks, int width, int height, float eps2d, float near_plane, float far_plane,
                         float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics,
                         float *compensations, float *means_world, void *stream);
int pr_project_bwd_rigid(int C, int N, int G, const float *init_p2ws, const float *p_deltas, const int32_t *labels,
                         const float *means, const float *quats, const float *scales, const float *viewmats,
                         const float *Ks, int width, int height, float eps2d, const int32_t *radii,
                         const float *v_means2d, int ld_means2d, const float *v_depths, const float *v_conics,
                         int ld_conics, const float *v_compensations, const float *v_means_world, float *v_p_deltas,
                         float *v_viewmats, void *stream);

/* ---- fused Adam over parameter groups (the step between the gradient exchange and the next forward) ------
 * Replaces the per-group torch.optim.Adam(lr, eps=1e-15).step() calls issued through nerfstudio's
 * Optimizers.optimizer_scaler_step_some (nerfstudio/engine/optimizers.py:159-172; trainer.py:494-510 decides
 * which groups are due under their gradient-accumulation periods; groups and rates: pogs/pogs_config.py:38-85).
 * Arithmetic of torch.optim.Adam without weight decay / amsgrad / maximize.  All groups passed are updated in
 * one launch.  step = the group's update count INCLUDING this one (>= 1); lr = the rate after its scheduler.
 * Pointers must be 16-byte aligned; param/exp_avg/exp_avg_sq are updated in place. */
typedef struct {
  float *param;
  const float *grad;
  float *exp_avg, *exp_avg_sq;
  int64_t numel, step;
  double lr, beta1, beta2, eps;
} pr_adam_group;
int pr_adam_max_groups(void);
int pr_adam_step(const pr_adam_group *groups, int n_groups, void *stream);

/* ---- fused channel MLP on the rendered feature image ------------------------------------------------------
 * Replaces pogs/pogs.py:516-522 with the module of pogs/pogs.py:153-161: for every pixel
 *   x = alpha > 0 ? feats / max(alpha, 1e-6) : 0            (skipped when alphas is NULL)
 *   y = W3 relu(W2 relu(W1 relu(W0 x)))                      (Wl: [64 out, 64 in] row-major, torch.nn.Linear.weight)
 *   y = 0 where alpha < zero_below                           (the reference's eval-mode cut at 0.8; <= 0: none)
 * in exact float32 on the matrix cores.  feats/out: [n_pixels, 64], 16-byte aligned; alphas: [n_pixels].
 * masks (may be NULL in pr_mlp_fwd): pr_mlp_mask_words(n_pixels) 64-bit words that receive the ReLU signs the
 * backward needs.  pr_mlp_bwd_input: gradient with respect to feats for FROZEN weights (what the tracker
 * needs; alpha receives no gradient, as in the reference where it is detached). */
size_t pr_mlp_mask_words(int64_t n_pixels);
int pr_mlp_fwd(int64_t n_pixels, const float *feats, const float *alphas, const float *w0, const float *w1,
               const float *w2, const float *w3, float zero_below, float *out, uint64_t *masks, void *stream);
int pr_mlp_bwd_input(int64_t n_pixels, const float *v_out, const float *alphas, const float *w0, const float *w1,
                     const float *w2, const float *w3, float zero_below, const uint64_t *masks, float *v_feats,
                     void *stream);
/* Training (the weights need gradients, pogs/pogs.py:153-161 with `nn_projection` in the optimisers,
 * pogs/pogs_config.py:78-85).  pr_mlp_fwd_train is pr_mlp_fwd that also stores the three hidden activations,
 * acts[3, n_pixels, 64].  pr_mlp_bwd_full: from v_out and those activations, the gradients of the four weight
 * matrices, v_weights[4, 64, 64] (torch.nn.Linear.weight layout; zero-filled by the caller, accumulated with float
 * atomics) and, when v_feats is not NULL, the gradient with respect to feats (overwritten).  Split-bf16 arithmetic. */
int pr_mlp_fwd_train(int64_t n_pixels, const float *feats, const float *alphas, const float *w0, const float *w1,
                     const float *w2, const float *w3, float zero_below, float *out, float *acts, void *stream);
int pr_mlp_bwd_full(int64_t n_pixels, const float *v_out, const float *feats, const float *alphas, const float *acts,
                    const float *w0, const float *w1, const float *w2, const float *w3, float zero_below,
                    float *v_feats, float *v_weights, void *stream);

/* ---- CLIP / instance heads on the rendered 96-channel feature image --------------------------------------------
 * Replace the two tiny-cuda-nn CutlassMLP networks of pogs/fields/gaussian_field.py:105-127 (bias-free, ReLU between
 * layers: 97 -> 256 x 3 -> 512 and 96 -> 256 x 4 -> 128) and the normalisation applied to their outputs at
 * pogs/fields/gaussian_field.py:184-190,194-204, one dense layer per call.  float32 in and out, split-bf16 products.
 *
 * pr_dense_pack: A[m][k] = src[m * stride_m + k * stride_k] (m < M, k < K) -> matrix-operand order, split, into
 *   `packed` (pr_dense_packed_bytes(M, K) bytes, 16-byte aligned).  Strides (K, 1) pack a Linear.weight as it is
 *   (forward), strides (1, K_in) with M = K_in, K = N_out pack its transpose (backward data product).
 * pr_dense_fwd: Y[p][m] = act(sum_k X[p][k] A[m][k]), p < P, m < M; relu != 0: act = max(0, .); mask != NULL: the
 *   result is zeroed where mask[p][m] <= 0 (backward through the ReLU whose output was `mask`).  X: [P, ldx >= K],
 *   Y: [P, ldy >= M], mask: [P, ldm >= M]; M, K and the leading dimensions multiples of 4, pointers 16-byte aligned.
 * pr_dense_wgrad: dW[n][k] += sum_p G[p][n] H[p][k] (float atomics: the caller zero-fills), G: [P, ldg >= N],
 *   H: [P, ldh >= K], dW: [N, ldw >= K].
 * pr_l2norm_fwd: out[p] = y[p] / (|y[p]| + eps), norms[p] = |y[p]| (may be NULL); pr_l2norm_bwd its adjoint.
 *   N a multiple of 4. */
size_t pr_dense_packed_bytes(int M, int K);
int pr_dense_pack(int M, int K, const float *src, int64_t stride_m, int64_t stride_k, void *packed, void *stream);
int pr_dense_fwd(int64_t P, int M, int K, const float *X, int ldx, const void *packed, int relu, const float *mask,
                 int ldm, float *Y, int ldy, void *stream);
int pr_dense_wgrad(int64_t P, int N, int K, const float *G, int ldg, const float *H, int ldh, float *dW, int ldw,
                   void *stream);
int pr_l2norm_fwd(int64_t P, int N, const float *y, float eps, float *out, float *norms, void *stream);
int pr_l2norm_bwd(int64_t P, int N, const float *y, const float *norms, float eps, const float *g, float *dy,
                  void *stream);

/* ---- densification: grow / prune the Gaussian set after the backward pass ---------------------------------
 * Replaces gsplat 1.4.0's DefaultStrategy (third-party gsplat/strategy/default.py, ops.py) as constructed at
 * nerfstudio/models/splatfacto.py:264-281 and driven at pogs/pogs.py:373-379 (step_pre_backward) and
 * nerfstudio/models/splatfacto.py:365-375 (step_post_backward). */

/* Per training step (DefaultStrategy._update_state, unpacked): for every Gaussian i and every camera c with
 * radii[c,i] > 0:  grad2d[i] += |(gx * width/2 * n_cameras, gy * height/2 * n_cameras)|,  count[i] += 1,
 * radii_max[i] = max(radii_max[i], radii[c,i] / max(width, height)) (radii_max may be NULL).
 * grads2d: [C,N,2] float (means2d.grad or .absgrad), radii: [C,N] int32. */
int pr_densify_accumulate(int C, int N, const float *grads2d, const int32_t *radii, int width, int height,
                          int n_cameras, float *grad2d, float *count, float *radii_max, void *stream);

/* Thresholds of one refinement, already multiplied by the scene scale where the strategy does so. */
typedef struct pr_densify_rule {
  float grow_grad2d;    /* duplicate/split when grad2d/max(count,1) exceeds it */
  float grow_scale3d;   /* ... duplicate when max(exp(scales)) <= it, else split */
  float grow_scale2d;   /* also split when radii_max exceeds it (only if grow_by_radius) */
  float prune_opa;      /* prune when sigmoid(opacity) is below it */
  float prune_scale3d;  /* prune when max(exp(scales)) exceeds it (only if prune_big) */
  float prune_scale2d;  /* ... or radii_max exceeds it (only if prune_by_radius) */
  int grow_by_radius;   /* step < refine_scale2d_stop_iter */
  int prune_big;        /* step > reset_every */
  int prune_by_radius;  /* prune_big and step < refine_scale2d_stop_iter */
  int revised_opacity;  /* children get opacity 1 - sqrt(1 - o) */
} pr_densify_rule;

/* Refinement, step 1 (DefaultStrategy._grow_gs + _prune_gs decisions): classify all N Gaussians and scan.
 * counts_dev[4] (device) = {n_out, n_duplicated, n_split, n_pruned}; scratch (256-byte aligned,
 * pr_densify_plan_scratch_bytes(N)) carries the plan to pr_densify_emit.  scales [N,3] and opacities [N] are
 * the raw (log / logit) parameters. */
size_t pr_densify_plan_scratch_bytes(int N);
int pr_densify_plan(int N, const float *grad2d, const float *count, const float *scales, const float *opacities,
                    const float *radii_max, const pr_densify_rule *rule, int32_t *counts_dev, void *scratch,
                    size_t scratch_bytes, void *stream);
/* Step 2: the gather map of the new set, n_out = counts_dev[0] entries, in the order duplicate -> split ->
 * remove (ops.py) leave: kept originals that were not split, kept copies, kept first children, kept second
 * children.  map_src[m] = source Gaussian; map_tag[m] = kind << 30 | r, kind 0 original, 1 copy, 2/3 first /
 * second child, r = rank of the source among all split sources (row of the noise samples). */
int pr_densify_emit(int N, const void *scratch, size_t scratch_bytes, int64_t n_out, int32_t *map_src,
                    int32_t *map_tag, void *stream);
/* Step 3: dst[m,:] = src[map_src[m],:] for rows of `width` floats; zero_new != 0 writes zeros for every entry
 * that is not an original (optimiser moments of new Gaussians, ops.py duplicate/split optimizer_fn). */
int pr_densify_gather(int64_t n_out, int width, const float *src, const int32_t *map_src, const int32_t *map_tag,
                      int zero_new, float *dst, void *stream);
/* Step 4 (ops.py split): overwrite the children's rows: means_out = mean + R(q/|q|) (exp(scale) * noise[b,r,:]),
 * scales_out = log(exp(scale) / 1.6), and, when opacities_out is not NULL, the revised opacity.
 * noise: [2, n_split, 3] standard-normal samples. */
int pr_densify_split_children(int64_t n_out, const int32_t *map_src, const int32_t *map_tag, int64_t n_split,
                              const float *means, const float *scales, const float *quats, const float *opacities,
                              const float *noise, float *means_out, float *scales_out, float *opacities_out,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif /* POGS_RAST_H */
