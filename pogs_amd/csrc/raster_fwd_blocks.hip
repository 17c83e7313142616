// Forward compositing for gfx950 with the wavefront split into eight lane groups that walk their own lists.
//
// One wavefront owns a 16x8 half tile.  Lanes 8q..8q+7 own the 4x4 pixel block q of it (two pixels per lane, two rows
// apart).  A Gaussian of the tile's list usually reaches only some of the eight blocks -- at 1M Gaussians and 1080p
// a list entry that passes the footprint test of the half tile touches 4.9 of them on average, and a pixel block is
// finished (saturated) long before its half tile is -- so the walk is decoupled: every entry that can touch the half
// tile is staged once (record + colour row, in a ring of LDS slots), an exact footprint test per (entry, block) then
// appends it to the lists of the blocks it touches, and in iteration i block q composites the i-th entry of ITS list.
// Different blocks are different pixels, so the per-pixel order of the Gaussians -- the only order alpha compositing
// depends on -- is the list order, and every pixel sees exactly the arithmetic of raster_fwd_kernel: an entry left out
// of a block's list is one whose alpha is below 1/255 on all of its pixels.  The eight walks advance at their own
// pace inside the window of the ring; a slot is reused once every block that listed its entry has composited it.
//
// Two forms.  The RING form above (POGS_RAST_FWD_BLOCKS=1) does not pay: at 1M Gaussians, 1080p, 36 channels it needs a
// third fewer iterations than raster_fwd_kernel<16,2,36>, but its bookkeeping (oldest-entry search, entries staged a few
// at a time as slots free up, a round of footprint tests per insertion) costs more than that: 214 M vector instructions
// per launch against 201 M, 0.56 against 0.46 ms.  The BATCH form (POGS_RAST_FWD_BLOCKS=2, and the planner's choice on
// images of fewer than 3072 tiles with 16 channels or more) drops the ring: 64 list entries are tested and staged at
// once, every block lists its entries of the batch, the blocks walk their lists side by side and meet again at the end
// of the batch.  It ties with the vector kernel at 1080p (0.46 ms) and wins where Gaussians are small against a half
// tile and lists long: 1M Gaussians at 450x250 with 64 channels 0.27 -> 0.21 ms, 200k at 640x360 with 64 channels
// 0.21 -> 0.18, 1M at 960x540 with 96 channels 0.51 -> 0.50.  Both forms are held bit-identical to the 2-pixel-per-lane
// vector kernel by the tests.
#include <stdlib.h>

#include "raster_common.h"

namespace pr {

// Ring slots per wavefront.  A slot is 32 bytes of record + a colour row; 64 slots at up to 36 channels are 12 KB
// (thirteen wavefronts per CU), 32 slots above.
constexpr int fb_ring(int cd) { return cd <= 36 ? 64 : 32; }
// List entries tested and staged per batch in the batch form.  (48 or 56 entries, which would let a fourth wavefront
// per SIMD fit the LDS, were measured 1-3 % slower at 36 channels, with or without the registers held to 128.)
constexpr int fb_batch(int) { return 64; }

template <int CD, bool RING>
struct alignas(16) BlockLds {
  static constexpr int R = RING ? fb_ring(CD) : fb_batch(CD);   // batch form: list entries per batch
  float4 ra[R];                    // x, y, opacity, A' (= -0.5 a log2e)
  float4 rb[R];                    // B' (= -b log2e), C' (= -0.5 c log2e), list position + 1 (bits), unused
  float col[R * color_row(CD)];
  unsigned short idx[8][64];       // per block: entry numbers (mod 2^16) of the staged entries that touch it
};

// Maximum over the wavefront of a non-negative value that is the same in the eight lanes of a group.
__device__ __forceinline__ int group_wave_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));  // row_shr:8  : lane 15 of a row = max of its groups
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xf, 0xf, true));  // row_bcast:15
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xf, 0xf, true));  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// RING: the eight walks advance independently inside a ring of staged entries (above).  !RING: batch by batch -- 64 list
// entries are tested and staged, every block lists its entries of the batch, the blocks walk their lists side by side
// and wait for each other at the end of the batch: no ring bookkeeping, at the price of the idle iterations of the
// blocks with the shorter lists.
template <int CD, bool RING>
__global__ void __launch_bounds__(64, CD <= 36 ? 3 : 2) raster_fwd_blocks_kernel(
    int n_regions, int N, int D, int ch_off0, int per, const float *__restrict__ means2d,
    const float *__restrict__ conics, const ColorSegs cs, const float *__restrict__ opacities,
    const float *__restrict__ backgrounds, int width, int height, int tile_w, int tile_h,
    const int32_t *__restrict__ isect_offsets, const int32_t *__restrict__ tile_order,
    const int32_t *__restrict__ flatten_ids, int n_isects, int ed_channel, float *__restrict__ render_colors,
    float *__restrict__ render_alphas, int32_t *__restrict__ last_ids) {
  constexpr int ROW = color_row(CD);
  constexpr int CDP = round4(CD);
  constexpr int CH2 = CDP / 2;
  constexpr int R = BlockLds<CD, RING>::R;
  __shared__ BlockLds<CD, RING> L;
  const int lane = threadIdx.x & 63;
  const int ch_off = ch_off0 + (int)blockIdx.y * per;   // channel passes of one call are rows of the grid
  const RegionId rid = region_of<2>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int ox = tx * 16, oy = ty * 16 + sub * 8;  // the half tile's first pixel
  if (ox >= width || oy >= height) return;         // nothing of it lies inside the image
  const int q = lane >> 3, li = lane & 7;
  const int bx = ox + (q & 3) * 4, by = oy + (q >> 2) * 4;  // the block's first pixel
  const int x = bx + (li & 3), y0 = by + (li >> 2);          // the lane's pixels: (x, y0) and (x, y0 + 2)
  const int range_lo = __builtin_amdgcn_readfirstlane(isect_offsets[tile_flat]);
  const int range_hi =
      __builtin_amdgcn_readfirstlane((tile_flat + 1 < n_regions / 2) ? isect_offsets[tile_flat + 1] : n_isects);

  float T[2], Tl[2];
  f2 acc[2][CH2];
  int last[2];
  bool inside[2];
  const float px = (float)x + 0.5f, py0 = (float)y0 + 0.5f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    inside[j] = x < width && y0 + 2 * j < height;
    T[j] = 1.f;
    Tl[j] = inside[j] ? 1.f : 0.f;
    last[j] = range_lo;
#pragma unroll
    for (int k = 0; k < CH2; ++k) acc[j][k] = splat(0.f);
  }
  // pixel-centre rectangles, clipped to the image: the half tile's, and the lane's block's (empty if outside)
  const float rx0 = (float)ox + 0.5f, ry0 = (float)oy + 0.5f;
  const float rx1 = (float)min(ox + 16, width) - 0.5f, ry1 = (float)min(oy + 8, height) - 0.5f;
  const float cx0 = (float)bx + 0.5f, cy0 = (float)by + 0.5f;
  const float cx1 = (float)min(bx + 4, width) - 0.5f, cy1 = (float)min(by + 4, height) - 0.5f;
  bool alive = bx < width && by < height;  // the block still has a pixel that accepts Gaussians

  // ---- the list, read two batches ahead (indices of the batch after next; records of the next through the indices
  //      that arrived a batch ago)
  constexpr int BS = RING ? 64 : fb_batch(CD);   // list entries per batch of candidates
  auto load_ids = [&](int first) { return lane < BS && first + lane < range_hi ? flatten_ids[first + lane] : -1; };
  auto load_records = [&](int g) {
    BatchGeometry b;
    b.g = g;
    b.xy = make_float2(0.f, 0.f);
    b.ca = b.cb = b.cc = b.o = 0.f;
    if (g >= 0) {
      b.xy = *reinterpret_cast<const float2 *>(means2d + 2 * (int64_t)g);
      b.ca = conics[3 * (int64_t)g];
      b.cb = conics[3 * (int64_t)g + 1];
      b.cc = conics[3 * (int64_t)g + 2];
      b.o = opacities[g];
    }
    return b;
  };
  int next = range_lo;  // first list entry of the batch held in `pre`
  BatchGeometry pre = load_records(load_ids(next));
  int ids_next = load_ids(next + BS);
  BatchGeometry cur = pre;  // the batch being inserted: lane e holds list entry cur_first + e
  int cur_first = range_lo;
  unsigned long long cand = 0ull;  // its entries that passed the half tile's test and are not staged yet

  if constexpr (RING) {
    int head = 0;       // entries staged so far; entry number s lives in slot s % R
    int k = 0, w = 0;   // this lane's block: entries composited / listed (the same in the 8 lanes of a group)
    int e_cur = 0;      // L.idx[q][k % R] while k < w; else the block's last entry (slot 0 before its first)
    // Every pass of the loop below stages an entry or composites one for some block, so it ends by itself; the bound
    // (every entry staged once and composited by at most eight blocks) only keeps a logic error from hanging the GPU.
    int guard = 9 * (range_hi - range_lo) + 64;

    for (; guard > 0; --guard) {
      // ---- stage entries while the ring has room
      bool staged = false;
      for (;;) {
        if (cand == 0ull) {
          if (next >= range_hi) break;
          cur = pre;
          cur_first = next;
          next += 64;
          pre = load_records(ids_next);
          ids_next = load_ids(next + 64);
          bool keep = false;
          if (cur.g >= 0) keep = touches_region(cur.xy.x, cur.xy.y, cur.o, cur.ca, cur.cb, cur.cc, rx0, ry0, rx1, ry1);
          cand = __builtin_amdgcn_ballot_w64(keep);
          if (cand == 0ull) continue;
        }
        // slots still referenced: from the oldest entry some block has not composited yet up to head
        const int age = k < w ? ((head - e_cur) & 0xffff) : 0;
        const int room = R - group_wave_max(age);
        if (room <= 0) break;
        const bool mine = (cand >> lane) & 1ull;
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cand >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cand, 0u));
        const bool ins = mine && pos < room;
        if (ins) {
          const int slot = (head + pos) & (R - 1);
          L.ra[slot] = make_float4(cur.xy.x, cur.xy.y, cur.o, -0.5f * LOG2E * cur.ca);
          L.rb[slot] = make_float4(-LOG2E * cur.cb, -0.5f * LOG2E * cur.cc,
                                   __builtin_bit_cast(float, cur_first + lane + 1), 0.f);
          stage_color_row<CD>(L.col + slot * ROW, cur.g, cs, N, D, ch_off);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the colour rows have landed (and nothing stays pending into the walk)
        const unsigned long long insmask = __builtin_amdgcn_ballot_w64(ins);
        const int n_ins = __builtin_popcountll(insmask);
        cand &= ~insmask;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // exact footprint test per (new entry, block): lane (q, li) tests entry head + base + li against block q
        for (int base = 0; base < n_ins; base += 8) {
          const int s = head + base + li;
          bool t = false;
          if (base + li < n_ins && alive) {
            const float4 a = L.ra[s & (R - 1)];
            const float4 b = L.rb[s & (R - 1)];
            t = touches_region_log2(a.x, a.y, a.z, -2.f * a.w, -b.x, -2.f * b.y, cx0, cy0, cx1, cy1);
          }
          const unsigned long long m = __builtin_amdgcn_ballot_w64(t);
          const unsigned byte = (unsigned)(m >> (8 * q)) & 0xffu;  // this block's eight results
          if (t) L.idx[q][(w + __builtin_popcount(byte & ((1u << li) - 1u))) & (R - 1)] = (unsigned short)s;
          w += __builtin_popcount(byte);
        }
        head += n_ins;
        staged = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k < w) e_cur = L.idx[q][k & (R - 1)];  // a block that had run out may have entries again
        if (cand != 0ull) break;        // the ring is full
      }

      // ---- how many iterations until a block runs out of listed entries (then more can be staged), or, with the
      //      list exhausted or the ring full, until the fullest block is done / has made room
      const int pending = w - k;
      const int most = group_wave_max(pending);
      const bool exhausted = next >= range_hi && cand == 0ull;
      if (most == 0) {
        if (exhausted) break;
        continue;
      }
      // (a fixed stride of a dozen iterations between two looks at the ring, with entries staged a dozen at a time, was
      // measured slower: 0.63 against 0.56 ms at 1M Gaussians, 1080p, 36 channels)
      int n_iter = most;
      if (!exhausted) {
        const int fewest = 0x7fff - group_wave_max(pending > 0 ? 0x7fff - pending : 0);
        n_iter = staged ? fewest : max(fewest, min(most, 8));
      }

      for (int it = 0; it < n_iter; ++it) {
        const bool act = k < w;
        const int slot = e_cur & (R - 1);
        const float4 g = L.ra[slot];
        const float4 h = L.rb[slot];
        const int e_next = L.idx[q][(k + 1) & (R - 1)];
        const int pos1 = __builtin_bit_cast(int, h.z);
        float vis[2];
        bool any = false;
        const float dx = g.x - px;
        const float adx2 = g.w * dx * dx;
        const float bdx = h.x * dx;
  #pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float dy = g.y - (py0 + (float)(2 * j));
          const float cdy2 = h.y * dy * dy;
          const float power = adx2 + cdy2 + bdx * dy;  // = -sigma * log2(e)
          const float alpha = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
          const bool valid = act && (power <= 0.f) && (alpha >= ALPHA_MIN);
          const float next_T = Tl[j] - alpha * Tl[j];
          const bool contrib = valid && (next_T > T_STOP);   // never for a finished pixel: next_T = 0
          vis[j] = contrib ? alpha * Tl[j] : 0.f;
          Tl[j] = valid ? (contrib ? next_T : 0.f) : Tl[j];  // valid but not composited: the pixel saturates here
          T[j] = contrib ? next_T : T[j];
          last[j] = contrib ? pos1 : last[j];
          any = any || contrib;
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          const float4 *crow = reinterpret_cast<const float4 *>(L.col + slot * ROW);
  #pragma unroll
          for (int c4 = 0; c4 < CDP / 4; ++c4) {
            const float4 c = crow[c4];
            const f2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
  #pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f2 v = splat(vis[j]);
              acc[j][2 * c4] = fma2(c01, v, acc[j][2 * c4]);
              acc[j][2 * c4 + 1] = fma2(c23, v, acc[j][2 * c4 + 1]);
            }
          }
        }
        // (a block without a further entry keeps pointing at its last one: its lanes go through the motions with zero
        // weights, and must do so on a slot that holds staged numbers)
        if (act) {
          ++k;
          if (k < w) e_cur = e_next;
        }
      }

      // ---- blocks whose pixels are all finished drop out: they list nothing further and discard what they had listed
      const unsigned long long livemask = __builtin_amdgcn_ballot_w64(Tl[0] > 0.f || Tl[1] > 0.f);
      if (livemask == 0ull) break;
      alive = ((unsigned)(livemask >> (8 * q)) & 0xffu) != 0u;
      if (!alive) k = w;
    }
  } else {
    for (int first = range_lo; first < range_hi; first += BS) {
      cur = pre;
      pre = load_records(ids_next);
      ids_next = load_ids(first + 2 * BS);
      bool keep = false;
      if (cur.g >= 0) keep = touches_region(cur.xy.x, cur.xy.y, cur.o, cur.ca, cur.cb, cur.cc, rx0, ry0, rx1, ry1);
      cand = __builtin_amdgcn_ballot_w64(keep);
      if (cand == 0ull) continue;
      if (keep) {   // survivors compacted in list order into slots 0..n-1
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cand >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cand, 0u));
        L.ra[slot] = make_float4(cur.xy.x, cur.xy.y, cur.o, -0.5f * LOG2E * cur.ca);
        L.rb[slot] = make_float4(-LOG2E * cur.cb, -0.5f * LOG2E * cur.cc, __builtin_bit_cast(float, first + lane + 1),
                                 0.f);
        stage_color_row<CD>(L.col + slot * ROW, cur.g, cs, N, D, ch_off);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the colour rows have landed
      const int n = __builtin_popcountll(cand);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // exact footprint test per (entry, block): lane (q, li) tests entry base + li against block q
      int cnt = 0;  // entries of this batch on the lane's block's list
      for (int base = 0; base < n; base += 8) {
        const int e = base + li;
        bool t = false;
        if (e < n && alive) {
          const float4 a = L.ra[e];
          const float4 b = L.rb[e];
          t = touches_region_log2(a.x, a.y, a.z, -2.f * a.w, -b.x, -2.f * b.y, cx0, cy0, cx1, cy1);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(t);
        const unsigned byte = (unsigned)(m >> (8 * q)) & 0xffu;  // this block's eight results
        if (t) L.idx[q][cnt + __builtin_popcount(byte & ((1u << li) - 1u))] = (unsigned short)e;
        cnt += __builtin_popcount(byte);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n_iter = group_wave_max(cnt);
      // (a block with fewer entries goes through the remaining iterations with zero weights on its last entry, or on
      // slot 0, which this batch has staged)
      int e_now = cnt > 0 ? (int)L.idx[q][0] : 0;
      for (int it = 0; it < n_iter; ++it) {
        const bool act = it < cnt;
        const int slot = e_now;
        const float4 g = L.ra[slot];
        const float4 h = L.rb[slot];
        const int e_next = L.idx[q][(it + 1) & 63];
        const int pos1 = __builtin_bit_cast(int, h.z);
        float vis[2];
        bool any = false;
        const float dx = g.x - px;
        const float adx2 = g.w * dx * dx;
        const float bdx = h.x * dx;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float dy = g.y - (py0 + (float)(2 * j));
          const float cdy2 = h.y * dy * dy;
          const float power = adx2 + cdy2 + bdx * dy;  // = -sigma * log2(e)
          const float alpha = fminf(ALPHA_MAX, g.z * __builtin_amdgcn_exp2f(power));
          const bool valid = act && (power <= 0.f) && (alpha >= ALPHA_MIN);
          const float next_T = Tl[j] - alpha * Tl[j];
          const bool contrib = valid && (next_T > T_STOP);   // never for a finished pixel: next_T = 0
          vis[j] = contrib ? alpha * Tl[j] : 0.f;
          Tl[j] = valid ? (contrib ? next_T : 0.f) : Tl[j];  // valid but not composited: the pixel saturates here
          T[j] = contrib ? next_T : T[j];
          last[j] = contrib ? pos1 : last[j];
          any = any || contrib;
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          const float4 *crow = reinterpret_cast<const float4 *>(L.col + slot * ROW);
#pragma unroll
          for (int c4 = 0; c4 < CDP / 4; ++c4) {
            const float4 c = crow[c4];
            const f2 c01 = {c.x, c.y}, c23 = {c.z, c.w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f2 v = splat(vis[j]);
              acc[j][2 * c4] = fma2(c01, v, acc[j][2 * c4]);
              acc[j][2 * c4 + 1] = fma2(c23, v, acc[j][2 * c4 + 1]);
            }
          }
        }
        if (it + 1 < cnt) e_now = e_next;
      }
      const unsigned long long livemask = __builtin_amdgcn_ballot_w64(Tl[0] > 0.f || Tl[1] > 0.f);
      if (livemask == 0ull) break;
      alive = ((unsigned)(livemask >> (8 * q)) & 0xffu) != 0u;
    }
  }

  const int nch = min(CD, D - ch_off);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (!inside[j]) continue;
    const int y = y0 + 2 * j;
    const int64_t pix = ((int64_t)cam * height + y) * width + x;
    float *out = render_colors + pix * D + ch_off;
    if (backgrounds) {
      const float *bg = backgrounds + (int64_t)cam * D + ch_off;
#pragma unroll
      for (int c = 0; c < CD; ++c)
        if (c < nch) acc[j][c / 2][c % 2] = fmaf(T[j], bg[c], acc[j][c / 2][c % 2]);
    }
    if (ed_channel >= 0) {  // expected depth: accumulated depth / max(alpha, 1e-10)
      const float inv = 1.f / fmaxf(1.f - T[j], 1e-10f);
#pragma unroll
      for (int c = 0; c < CD; ++c)
        if (c + ch_off == ed_channel) acc[j][c / 2][c % 2] *= inv;
    }
    if (((D | ch_off) & 3) == 0) {
#pragma unroll
      for (int c = 0; c + 3 < CD; c += 4)
        if (c < nch)
          *reinterpret_cast<float4 *>(out + c) =
              make_float4(acc[j][c / 2][0], acc[j][c / 2][1], acc[j][c / 2 + 1][0], acc[j][c / 2 + 1][1]);
    } else {
#pragma unroll
      for (int c = 0; c < CD; ++c)
        if (c < nch) out[c] = acc[j][c / 2][c % 2];
    }
    if (ch_off == 0) {
      render_alphas[pix] = 1.f - T[j];
      last_ids[pix] = last[j];
    }
  }
}

template <int CD, bool RING>
static int launch_blocks(int C, int N, int D, int ch_off, int per, int n_pass, const float *means2d, const float *conics,
                         const ColorSegs &cs, const float *opacities, const float *backgrounds, int width, int height,
                         int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                         const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors,
                         float *render_alphas, int32_t *last_ids, hipStream_t stream) {
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * 2;
  const unsigned grid = region_grid(n_regions, 2, tile_order != nullptr);
  hipLaunchKernelGGL((raster_fwd_blocks_kernel<CD, RING>), dim3(grid, n_pass), dim3(64), 0, stream, (int)n_regions, N, D, ch_off,
                     per, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets,
                     tile_order, flatten_ids, (int)n_isects, ed_channel, render_colors, render_alphas, last_ids);
  return check_launch("raster_fwd_blocks_kernel");
}

bool fwd_blocks_supports(int cd, int tile_size) { return tile_size == 16 && cd >= 4 && cd <= 48; }

// `n_pass` passes of `per` channels each from ch_off on (the last may be narrower: cd covers `per`), as one launch.
int launch_fwd_blocks(int cd, bool ring, int C, int N, int D, int ch_off, int per, int n_pass, const float *means2d,
                      const float *conics, const ColorSegs &cs, const float *opacities, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                      const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors,
                      float *render_alphas, int32_t *last_ids, hipStream_t stream) {
#define BLOCKS_CASE(CDV)                                                                                               \
  case CDV:                                                                                                            \
    return ring ? launch_blocks<CDV, true>(C, N, D, ch_off, per, n_pass, means2d, conics, cs, opacities, backgrounds,   \
                                           width, height, tile_w, tile_h, isect_offsets, tile_order, flatten_ids,     \
                                           n_isects, ed_channel, render_colors, render_alphas, last_ids, stream)     \
                : launch_blocks<CDV, false>(C, N, D, ch_off, per, n_pass, means2d, conics, cs, opacities, backgrounds,  \
                                            width, height, tile_w, tile_h, isect_offsets, tile_order, flatten_ids,    \
                                            n_isects, ed_channel, render_colors, render_alphas, last_ids, stream);
  switch (cd) {
    BLOCKS_CASE(4) BLOCKS_CASE(8) BLOCKS_CASE(16) BLOCKS_CASE(24) BLOCKS_CASE(32) BLOCKS_CASE(36) BLOCKS_CASE(48)
  }
#undef BLOCKS_CASE
  return PR_EUNSUPPORTED;
}

}  // namespace pr
