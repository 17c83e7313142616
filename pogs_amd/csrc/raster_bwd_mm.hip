// Backward compositor for gfx950 with BOTH channel contractions on the float32 matrix cores and nothing unrolled
// over list slots.  Selected by pr_raster_bwd (raster_bwd.hip).
//
// One wavefront owns an 8x8 pixel region (a quadrant of a 16x16 tile, or a whole 8x8 tile), one pixel per lane.
// Per (pixel p, Gaussian g) the backward needs S[p][g] = <colour_g, v_p> (v = the pixel's colour cotangents) and
// per Gaussian the colour gradient vC[g][c] = sum_p fac[p][g] v[p][c].  Both are matrix products in which one factor,
// V[64 pixels][channels], is constant for the region.  It is brought into matrix-operand registers ONCE, in both
// layouts the two contraction indices ask for (channels across the lane groups for S, pixels across them for vC),
// and the float32 copy of the cotangents is dropped.  v_mfma_f32_16x16x4_f32 multiplies float32 by float32 exactly
// and accumulates in float32: the arithmetic is the vector kernel's, only the summation order differs.
//
//  * List entries that survive the region's footprint test are compacted into a ring of records in LDS; the walk
//    takes them 16 at a time.  For a chunk, S^T[16 Gaussians][64 pixels] = C[16][channels] . V^T: the A operand (the
//    chunk's colour rows) is gathered straight from global memory into registers -- colour rows never pass through
//    LDS -- and the result goes to an LDS table S[slot][pixel] from which each iteration reads its pixel's value.
//  * A contributing iteration stores its pixel's fac into the next free row of a 16 x 64 LDS image; every 16
//    contributing iterations vC[16 rows][channels] = fac[16 x 64] . V leaves as float atomics on 64-byte row segments
//    (rows of a matrix product are independent, so rows not yet filled may hold anything).
//  * What stays on the vector unit per iteration: alpha, the transmittance recurrence, the geometry terms and
//    their two lane-swap folds.
#include <algorithm>

#include "raster_common.h"

namespace pr {

typedef float f4v __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A[row l&15][k = l>>4], B[k = l>>4][col l&15], D: col l&15, rows 4(l>>4) + reg
__device__ __forceinline__ f4v mfma_f32(float a, float b, f4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int MM_QN = 64;   // ring of compacted records
constexpr int MM_BS = 32;   // list entries tested per refill (<= QN - 31: a chunk and its successor stay in the ring)
constexpr int MM_GP = 4;    // contributing iterations whose geometry totals are read back together
constexpr int MM_RW = 16;   // rows of a chunk / of the fac image
constexpr int MM_SP = 68;   // row pitch (floats) of the S table and the fac image

template <bool GEOM, bool CGRAD>
struct alignas(16) MmLds {
  float4 ra[MM_QN];  // x, y, opacity, A' (= -0.5 a log2e)
  float4 rb[MM_QN];  // B' (= -b log2e), C' (= -0.5 c log2e), flat Gaussian index, list position (both as bits)
  union {
    struct {
      float sbuf[GEOM ? MM_RW * MM_SP : 4];     // S[slot][pixel]
      float fimg[CGRAD ? MM_RW * MM_SP : 4];    // fac[row][pixel]
    } w;
    float tr[64 * 20];  // prologue: 16 channels of the region's cotangents, pixel-major, for the two operand layouts
  } u;
  float red[GEOM ? MM_GP * 2 * 68 : 4];  // lane-swap folds of the geometry values of up to GP iterations
  int32_t rowgid[MM_RW];      // flat Gaussian index of every row of the fac image
  int32_t geogid[MM_GP];      // ... and of every pending geometry fold
};

// MODE: what the launch produces -- the geometry gradients alone (all D channels enter the dot product S), both kinds of
// gradients, or the colour gradients of the channel window [ch_off, ch_off + CD) alone (no S, no geometry: the walk only
// rebuilds fac).  Passes wider than 72 channels run as one geometry launch plus colour launches over windows: the two
// operand copies of a wider cotangent block do not fit the register file together.
enum { MM_GEOM = 0, MM_BOTH = 1, MM_COLOUR = 2 };
// registers: the operands (CD for S plus its A-operand prefetch, 16 per column block for vC) and about 70 of state
// (the LDS slice allows four wavefronts per SIMD for the one-sided modes, three when both are produced)
constexpr int mm_regs(int cd, int mode) {
  return (mode != MM_COLOUR ? cd + (cd + 3) / 4 : 0) + (mode != MM_GEOM ? 16 * ((cd + 15) / 16) : 0) + 70;
}
constexpr int mm_waves(int cd, int mode) {
  return mm_regs(cd, mode) <= 128 && mode != MM_BOTH ? 4 : mm_regs(cd, mode) <= 168 ? 3 : 2;
}

template <int TS, int CD, int MODE>
__global__ void __launch_bounds__(64, mm_waves(CD, MODE)) raster_bwd_mm_kernel(
    int n_regions, int N, int D, int ch_off0, int per, const float *__restrict__ means2d,
    const float *__restrict__ conics,
    const ColorSegs cs, const float *__restrict__ opacities, const float *__restrict__ backgrounds, int width,
    int height, int tile_w, int tile_h, const int32_t *__restrict__ isect_offsets,
    const int32_t *__restrict__ tile_order, const int32_t *__restrict__ flatten_ids, int n_isects,
    const float *__restrict__ render_alphas, const int32_t *__restrict__ last_ids,
    const float *__restrict__ render_colors, int ed_channel, const float *__restrict__ v_render_colors,
    const float *__restrict__ v_render_alphas, int absgrad, float *__restrict__ v_geom, const GradSegs gs) {
  constexpr bool GEOM = MODE != MM_COLOUR, CGRAD = MODE != MM_GEOM;
  constexpr int WPT = TS == 16 ? 4 : 1;
  constexpr int NS = (CD + 3) / 4;     // k steps of S: channel 4 s + (lane >> 4)
  constexpr int NCB = (CD + 15) / 16;  // 16-channel column blocks of vC (and of the prologue's transposes)
  constexpr int CDP = 16 * NCB;
  constexpr int QN = MM_QN, BS = MM_BS, RW = MM_RW, SP = MM_SP, GP = MM_GP;
  __shared__ MmLds<GEOM, CGRAD> L;
  const int lane = threadIdx.x & 63;
  const RegionId rid = region_of<WPT>(tile_order, n_regions);
  if (!rid.valid) return;
  const int tile_flat = rid.tile_flat, sub = rid.sub;
  const int n_tiles = tile_w * tile_h;
  const int cam = tile_flat / n_tiles, tile = tile_flat % n_tiles;
  const int ty = tile / tile_w, tx = tile % tile_w;
  const int ox = tx * TS + (sub & 1) * 8, oy = ty * TS + (sub >> 1) * 8;
  const int x = ox + (lane & 7), y = oy + (lane >> 3);
  const int range_lo = __builtin_amdgcn_readfirstlane(isect_offsets[tile_flat]);
  // the colour windows of a call are rows of one launch
  const int ch_off = ch_off0 + (MODE == MM_COLOUR ? (int)blockIdx.y * per : 0);
  const int nch = min(CD, D - ch_off);
  const int l15 = lane & 15, kk = lane >> 4;
  const float px = (float)x + 0.5f, py = (float)y + 0.5f;

  float T = 1.f, Bsum = 0.f, W = 0.f;
  int last = range_lo;
  float Bs[4][NS];     // S operand: pixel 16 b + (lane & 15), channel 4 s + (lane >> 4)
  float Bc[16][NCB];   // vC operand: pixel 16 (lane >> 4) + s, channel vch[n]
  // Column (lane & 15) of block n of the colour-gradient product is channel (16 n + (lane & 15) + rot) mod CDP, rot chosen
  // so that the widest gradient tensor starts on a block boundary: a block's 16 floats of a row then are one aligned
  // 64-byte segment of that tensor, which is the unit the memory system delivers float atomics in.
  int vch[NCB];   // channel ch_off + vch[n]
  {
    // widths of the (up to three) gradient tensors; unused segments begin at INT_MAX
    const int e1 = min(gs.begin[1], D), e2 = min(gs.begin[2], D);
    const int w0 = e1, w1 = e2 - e1, w2 = D - e2;
    const int rot = ((w2 > w1 && w2 > w0 ? e2 : w1 > w0 ? e1 : 0) - ch_off) & 15;
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
      const int c = 16 * n + l15 + rot;
      vch[n] = c >= CDP ? c - CDP : c;
    }
  }
  {
    float v[CDP];
#pragma unroll
    for (int k = 0; k < CDP; ++k) v[k] = 0.f;
    if (x < width && y < height) {
      const int64_t pix = ((int64_t)cam * height + y) * width + x;
      T = 1.f - render_alphas[pix];
      last = last_ids[pix];
      const float *src = v_render_colors + pix * D + ch_off;
      if (((D | ch_off) & 3) == 0) {
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
      if (ed_channel >= 0) {
        // out_d = X / max(a, 1e-10): the cotangent of the accumulated depth X is v_out_d / a, and the
        // direct dependence on a adds -v_out_d * out_d / a to the alpha cotangent (none when clamped).
        const float a = 1.f - T, inv = 1.f / fmaxf(a, 1e-10f);
#pragma unroll
        for (int k = 0; k < CD; ++k)
          if (k + ch_off == ed_channel) {
            if (a > 1e-10f) w_ed = -v[k] * render_colors[pix * D + ed_channel] * inv;
            v[k] *= inv;
          }
      }
      if (GEOM) {   // (launched with ch_off = 0: every channel is here)
        float w = v_render_alphas[pix] + w_ed;
        if (backgrounds) {
          const float *bg = backgrounds + (int64_t)cam * D;
#pragma unroll
          for (int k = 0; k < CD; ++k)
            if (k < nch) w -= bg[k] * v[k];
        }
        W = T * w;
      }
    }
    // both operand layouts, 16 channels at a time through LDS (pixel-major rows of 20 floats)
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        *reinterpret_cast<float4 *>(&L.u.tr[lane * 20 + 4 * c4]) =
            make_float4(v[16 * n + 4 * c4], v[16 * n + 4 * c4 + 1], v[16 * n + 4 * c4 + 2], v[16 * n + 4 * c4 + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int m = 0; m < (CGRAD ? NCB : 0); ++m)   // block m's column of this lane, if it lies in this chunk of channels
        if ((vch[m] >> 4) == n) {
#pragma unroll
          for (int s = 0; s < 16; ++s) Bc[s][m] = L.u.tr[(16 * kk + s) * 20 + (vch[m] & 15)];
        }
#pragma unroll
      for (int s = 4 * n; s < 4 * n + 4; ++s)
        if (GEOM && s < NS) {
#pragma unroll
          for (int b = 0; b < 4; ++b) Bs[b][s] = L.u.tr[(16 * b + l15) * 20 + 4 * (s & 3) + kk];
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  int max_last = last;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) max_last = max(max_last, __shfl_xor(max_last, m));
  max_last = __builtin_amdgcn_readfirstlane(max_last);  // the same in every lane: keep the loop control scalar
  if (max_last <= range_lo) return;

  // Geometry totals of up to GP contributing iterations are read back together: lane 8 i + v owns value v of
  // pending iteration i; its 16 partials sit in one 16-float row segment of that iteration's fold area.
  const int own_it = lane >> 3, own_v = lane & 7;
  const int own_q = own_v >> 2, own_row = ((own_v & 1) << 1) | ((own_v >> 1) & 1);
  const float4 *own = reinterpret_cast<const float4 *>(L.red + own_it * (2 * 68) + own_q * 68 + own_row * 16);
  // post-scale: x y (-ln2), conic a c (1/2), b, opacity (1), |x| |y| (ln2)
  const float gscale = (own_v == 0 || own_v == 1) ? -LN2 : (own_v == 2 || own_v == 4) ? 0.5f
                       : (own_v == 6 || own_v == 7) ? LN2 : 1.f;
  const int n_geo = absgrad ? 8 : 6;
  const int cam_first = cam * N;
  int npend = 0;  // iterations whose geometry folds wait in LDS (wave-uniform)
  auto geo_flush = [&](int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane < 8 * count) {
      const float4 p0 = own[0], p1 = own[1], p2 = own[2], p3 = own[3];
      const int gid = L.geogid[own_it];
      const float tot = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w)) +
                        ((p2.x + p2.y) + (p2.z + p2.w)) + ((p3.x + p3.y) + (p3.z + p3.w));
      if (own_v < n_geo) atomic_add_noret_s(v_geom, 4u * ((unsigned)gid * (unsigned)gs.geom_ld + own_v), tot * gscale);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  int nrows = 0;  // rows of the fac image in use (wave-uniform)
  // vC of the image's first `count` rows -> atomics
  auto flush = [&](int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    f4v acc[NCB];
#pragma unroll
    for (int n = 0; n < NCB; ++n) acc[n] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a4 = *reinterpret_cast<const float4 *>(&L.u.w.fimg[l15 * SP + 16 * kk + 4 * i]);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NCB; ++n) acc[n] = mfma_f32(a[j], Bc[4 * i + j][n], acc[n]);
    }
    settle_mfma_results();
    const int4 g4 = *reinterpret_cast<const int4 *>(&L.rowgid[4 * kk]);
    const int gr[4] = {g4.x, g4.y, g4.z, g4.w};
    // A block's lanes may belong to different gradient tensors: per (block, tensor) the lanes of that tensor issue
    // with the tensor's scalar base and a 32-bit byte offset; combinations without a lane are skipped as a whole.
    auto scatter = [&](int n, float *base, int begin, int end, int ld, int shared) {
      const int ch = ch_off + vch[n];
      const bool mine = base != nullptr && ch >= begin && ch < end && vch[n] < nch;
      if (__builtin_amdgcn_ballot_w64(mine) == 0ull) return;
      if (mine) {
        const int sub = shared ? cam_first : 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kk + r < count) {
            const unsigned off = 4u * ((unsigned)(gr[r] - sub) * (unsigned)ld + (unsigned)(ch - begin));
            atomic_add_noret_s(base, off, acc[n][r]);
          }
      }
    };
    // The gradient tensors' table, field by field into scalar registers at the point of use.  (Left in the argument
    // block, a per-lane choice between tensors turns into a per-lane LOAD from it, and the wait for that load also
    // waits out every atomic in flight; held in registers for the whole kernel it costs thirty of them.)
    float *gp0 = gs.ptr[0], *gp1 = gs.ptr[1], *gp2 = gs.ptr[2];
    int gb1 = gs.begin[1], gb2 = gs.begin[2], gl0 = gs.ld[0], gl1 = gs.ld[1], gl2 = gs.ld[2];
    int gsh0 = gs.shared[0], gsh1 = gs.shared[1], gsh2 = gs.shared[2];
    asm volatile("" : "+s"(gp0), "+s"(gp1), "+s"(gp2));
    asm volatile("" : "+s"(gb1), "+s"(gb2), "+s"(gl0), "+s"(gl1), "+s"(gl2), "+s"(gsh0), "+s"(gsh1), "+s"(gsh2));
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
      scatter(n, gp0, 0, gb1, gl0, gsh0);
      scatter(n, gp1, gb1, gb2, gl1, gsh1);
      scatter(n, gp2, gb2, 0x7fffffff, gl2, gsh2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  const int ox1 = min(ox + 8, width), oy1 = min(oy + 8, height);

  int qhead = 0, qcount = 0;  // ring state (wave-uniform)
  int bend = max_last;        // list entries [range_lo, bend) not yet tested
  // The list is read two batches ahead: the flat indices of the batch after next, and through the indices that
  // arrived a batch ago the records of the next one -- neither request waits for the other.
  auto load_ids = [&](int end) {
    const int first = max(range_lo, end - BS);
    return lane < end - first ? flatten_ids[first + lane] : -1;
  };
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
  BatchGeometry pre = load_records(load_ids(bend));
  int ids_next = load_ids(max(range_lo, bend - BS));
  // Test list entries, back to front, until `want` records wait in the ring or the list is exhausted.  The next
  // batch's geometry is requested as soon as one has been consumed, a whole walk ahead of its use.
  auto refill = [&](int want) {
    while (qcount < want && bend > range_lo) {
      const int first = max(range_lo, bend - BS);
      // The region's pixel-centre rectangle is rebuilt from scalar registers at every use: as four loop-invariant
      // vector registers it would be spilled (the empty asm keeps the conversion inside the loop).
      int sx0 = ox, sy0 = oy, sx1 = ox1, sy1 = oy1;
      asm volatile("" : "+s"(sx0), "+s"(sy0), "+s"(sx1), "+s"(sy1));
      const float rx0 = (float)sx0 + 0.5f, ry0 = (float)sy0 + 0.5f, rx1 = (float)sx1 - 0.5f, ry1 = (float)sy1 - 0.5f;
      bool keep = false;
      if (pre.g >= 0) keep = touches_region(pre.xy.x, pre.xy.y, pre.o, pre.ca, pre.cb, pre.cc, rx0, ry0, rx1, ry1);
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
      if (keep) {
        // walk order is descending list position: this entry's place = survivors above it
        const int pos = __builtin_popcountll(mask >> 1 >> lane);
        const int qi = (qhead + qcount + pos) & (QN - 1);
        L.ra[qi] = make_float4(pre.xy.x, pre.xy.y, pre.o, -0.5f * LOG2E * pre.ca);
        L.rb[qi] = make_float4(-LOG2E * pre.cb, -0.5f * LOG2E * pre.cc, __builtin_bit_cast(float, pre.g),
                               __builtin_bit_cast(float, first + lane));
      }
      qcount += __builtin_popcountll(mask);
      bend = first;
      pre = load_records(ids_next);
      ids_next = load_ids(max(range_lo, bend - BS));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // A operand of S for the chunk whose first record sits at ring position `base`: lane (row, k group) takes the
  // channels 4 s + (lane >> 4) of the row's Gaussian straight from the colour tensors; rows past `rows` are zero.
  float a[NS];
  auto gather = [&](int base, int rows) {
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = 0.f;
    // the colour tensors' table into scalar registers at the point of use (see flush)
    const float *cp0 = cs.ptr[0], *cp1 = cs.ptr[1], *cp2 = cs.ptr[2];
    int cb1 = cs.begin[1], cb2 = cs.begin[2], cl0 = cs.ld[0], cl1 = cs.ld[1], cl2 = cs.ld[2];
    int csh0 = cs.shared[0], csh1 = cs.shared[1], csh2 = cs.shared[2];
    asm volatile("" : "+s"(cp0), "+s"(cp1), "+s"(cp2));
    asm volatile("" : "+s"(cb1), "+s"(cb2), "+s"(cl0), "+s"(cl1), "+s"(cl2), "+s"(csh0), "+s"(csh1), "+s"(csh2));
    if (l15 < rows) {
      const int gid = __builtin_bit_cast(int, L.rb[(base + l15) & (QN - 1)].z);
      const int gn = gid - cam_first;
      // byte offset of the row inside each tensor (32 bits: the entry point sends larger tensors elsewhere)
      const unsigned r0 = 4u * (unsigned)(csh0 ? gn : gid) * (unsigned)cl0;
      const unsigned r1 = 4u * (unsigned)(csh1 ? gn : gid) * (unsigned)cl1;
      const unsigned r2 = 4u * (unsigned)(csh2 ? gn : gid) * (unsigned)cl2;
      // (explicitly a GLOBAL load: behind the empty asm the pointers have lost their address space, and a generic
      // load would also count as an LDS access in the wait counters)
      auto at = [](const float *p, unsigned byte_off) {
        typedef __attribute__((address_space(1))) const char gchar;
        typedef __attribute__((address_space(1))) const float gfloat;
        return *(gfloat *)((gchar *)p + byte_off);
      };
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c0 = 4 * s, ch = c0 + kk;
        // the step's four channels usually lie in one tensor: one load with a scalar base; else chosen per lane
        if (c0 + 3 < cb1) {
          if (ch < nch) a[s] = at(cp0, r0 + 4u * ch);
        } else if (c0 >= cb1 && c0 + 3 < cb2) {
          if (ch < nch) a[s] = at(cp1, r1 + 4u * (ch - cb1));
        } else if (c0 >= cb2) {
          if (ch < nch) a[s] = at(cp2, r2 + 4u * (ch - cb2));
        } else if (ch < nch) {
          if (ch >= cb2) a[s] = at(cp2, r2 + 4u * (ch - cb2));
          else if (ch >= cb1) a[s] = at(cp1, r1 + 4u * (ch - cb1));
          else a[s] = at(cp0, r0 + 4u * ch);
        }
      }
    }
  };

  refill(RW);
  if (qcount == 0) return;
  if (GEOM) gather(qhead, min(RW, qcount));
  for (;;) {
    const int n = min(RW, qcount);
    if (GEOM) {  // S^T[slot][pixel] for the chunk
      f4v acc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = mfma_f32(a[s], Bs[b][s], acc[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.u.w.sbuf[(4 * kk + r) * SP + 16 * b + l15] = acc[b][r];
    }
    // the records of the chunk after this one, then its colour rows: both travel during the walk
    refill(n + RW);
    if (GEOM) gather(qhead + n, min(RW, qcount - n));

    float4 g_n = L.ra[qhead], h_n = L.rb[qhead];
    float s_n = GEOM ? L.u.w.sbuf[lane] : 0.f;
    for (int t = 0; t < n; ++t) {
      const float4 g = g_n, h = h_n;
      const float S = s_n;
      {  // the next slot's record and S are requested before this one is worked on
        const int qn = (qhead + t + 1) & (QN - 1);
        g_n = L.ra[qn];
        h_n = L.rb[qn];
        if (GEOM) s_n = L.u.w.sbuf[((t + 1) & (RW - 1)) * SP + lane];
      }
      const int gid = __builtin_bit_cast(int, h.z), idx = __builtin_bit_cast(int, h.w);
      const float dx = g.x - px, dy = g.y - py;
      const float power = g.w * dx * dx + h.y * dy * dy + h.x * dx * dy;
      const float e = __builtin_amdgcn_exp2f(power), ar = g.z * e;
      const float al = fminf(ALPHA_MAX, ar);
      const bool active = (power <= 0.f) && (al >= ALPHA_MIN) && (idx < last);
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) continue;
      const float alpha = active ? al : 0.f;
      const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
      T *= ra;
      const float fac = alpha * T;
      if (CGRAD) {
        L.u.w.fimg[nrows * SP + lane] = fac;
        L.rowgid[nrows] = gid;
      }
      if (!GEOM) {
        if (++nrows == RW) {
          flush(RW);
          nrows = 0;
        }
        continue;
      }
      const float v_alpha = S * T - (Bsum - W) * ra;
      Bsum = fmaf(fac, S, Bsum);
      // the exponential passes a gradient on when the pixel took this Gaussian at its unclamped alpha
      const float ev = (alpha == ar ? e : 0.f) * v_alpha;
      const float vs = -g.z * ev;
      const float t1 = vs * dx, t2 = vs * dy;
      const float gx = 2.f * g.w * dx + h.x * dy, gy = h.x * dx + 2.f * h.y * dy;
      const float vx = vs * gx, vy = vs * gy;
      L.red[npend * (2 * 68) + lane] = quad_fold_rows(vx, vy, t1 * dx, t1 * dy);
      L.red[npend * (2 * 68) + 68 + lane] = quad_fold_rows(t2 * dy, ev, fabsf(vx), fabsf(vy));
      L.geogid[npend] = gid;
      if (++npend == GP) {
        geo_flush(GP);
        npend = 0;
      }
      if (CGRAD && ++nrows == RW) {
        flush(RW);
        nrows = 0;
      }
    }
    qhead = (qhead + n) & (QN - 1);
    qcount -= n;
    if (qcount == 0) break;
  }
  if (GEOM && npend) geo_flush(npend);
  if (CGRAD && nrows) flush(nrows);
}

template <int TS, int CD, int MODE>
static int launch_mm(int C, int N, int D, int ch_off, int per, int n_pass, const float *means2d, const float *conics, const ColorSegs &cs,
                     const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                     const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                     int64_t n_isects, const float *render_alphas, const int32_t *last_ids,
                     const float *render_colors, int ed_channel, const float *v_render_colors,
                     const float *v_render_alphas, int absgrad, float *v_geom, const GradSegs &gs,
                     hipStream_t stream) {
  constexpr int WPT = TS == 16 ? 4 : 1;
  const int64_t n_regions = (int64_t)C * tile_w * tile_h * WPT;
  const unsigned grid = region_grid(n_regions, WPT, tile_order != nullptr);
  hipLaunchKernelGGL((raster_bwd_mm_kernel<TS, CD, MODE>), dim3(grid, n_pass), dim3(64), 0, stream, (int)n_regions, N, D,
                     ch_off, per, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h,
                     isect_offsets, tile_order, flatten_ids, (int)n_isects, render_alphas, last_ids, render_colors,
                     ed_channel, v_render_colors, v_render_alphas, absgrad, v_geom, gs);
  return check_launch("raster_bwd_mm_kernel");
}

// Channel counts the kernel is built for (pick_chunk's list in raster_bwd.hip); 100 = geometry launch only.
bool bwd_mm_supports(int cd, int tile_size) {
  return (tile_size == 16 || tile_size == 8) && cd >= 4 && cd <= 100;
}

int launch_bwd_mm(int cd, int tile_size, bool geom_grad, bool color_grad, bool split, int C, int N, int D,
                  const float *means2d, const float *conics, const ColorSegs &cs, const float *opacities,
                  const float *backgrounds, int width, int height, int tile_w, int tile_h,
                  const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                  int64_t n_isects, const float *render_alphas, const int32_t *last_ids, const float *render_colors,
                  int ed_channel, const float *v_render_colors, const float *v_render_alphas, int absgrad,
                  float *v_geom, const GradSegs &gs, hipStream_t stream) {
#define MM_ARGS(off)                                                                                                   \
  C, N, D, off, per, n_pass, means2d, conics, cs, opacities, backgrounds, width, height, tile_w, tile_h, isect_offsets, tile_order, \
      flatten_ids, n_isects, render_alphas, last_ids, render_colors, ed_channel, v_render_colors, v_render_alphas,    \
      absgrad, v_geom, gs, stream
#define MM_CASE(CDV, MODE, off)                                                         \
  case CDV:                                                                             \
    rc = tile_size == 16 ? launch_mm<16, CDV, MODE>(MM_ARGS(off)) : launch_mm<8, CDV, MODE>(MM_ARGS(off)); \
    break;
  int rc = PR_EUNSUPPORTED;
  int per = 0, n_pass = 1;   // (only the colour-window launch has more than one row)
  if (geom_grad && color_grad && cd <= 64 && !split) {   // one launch
    switch (cd) {
      MM_CASE(4, MM_BOTH, 0) MM_CASE(8, MM_BOTH, 0) MM_CASE(16, MM_BOTH, 0) MM_CASE(24, MM_BOTH, 0)
      MM_CASE(32, MM_BOTH, 0) MM_CASE(36, MM_BOTH, 0) MM_CASE(48, MM_BOTH, 0) MM_CASE(64, MM_BOTH, 0)
    }
    return rc;
  }
  if (geom_grad) {
    switch (cd) {
      MM_CASE(4, MM_GEOM, 0) MM_CASE(8, MM_GEOM, 0) MM_CASE(16, MM_GEOM, 0) MM_CASE(24, MM_GEOM, 0)
      MM_CASE(32, MM_GEOM, 0) MM_CASE(36, MM_GEOM, 0) MM_CASE(48, MM_GEOM, 0) MM_CASE(64, MM_GEOM, 0)
      MM_CASE(72, MM_GEOM, 0) MM_CASE(100, MM_GEOM, 0)
    }
    if (rc || !color_grad) return rc;
  }
  // Colour gradients in windows of up to 48 channels (three 16-channel column blocks), all windows of the call as rows
  // of one launch: 64 channels at 450x250 (two windows of 32 over 1856 regions) 0.25 -> 0.17 ms against one launch per
  // window, 96 channels at 960x540 after the geometry launch 1.01 -> 0.94.  Narrower windows do not pay, on small images
  // either (every window repeats the walk): 0.22 ms with four windows of 16 in the first case.
  const int widest = tuning().bwd_window > 0 ? tuning().bwd_window : 48;
  // Few channels want gradients (a wide image of frozen features with a depth or colour channel at either end): only
  // the 16-channel blocks that hold a wanted channel are walked, one launch per run of such blocks.
  {
    int lo[3], hi[3], n_runs = 0, wanted = 0;
    for (int i = 0; i < gs.n; ++i) {
      if (!gs.ptr[i]) continue;
      const int b = gs.begin[i], e = i + 1 < gs.n ? gs.begin[i + 1] : D;
      const int a0 = b & ~15, a1 = std::min(D, (e + 15) & ~15);
      if (n_runs && a0 <= hi[n_runs - 1]) hi[n_runs - 1] = std::max(hi[n_runs - 1], a1);   // (segments are in channel order)
      else lo[n_runs] = a0, hi[n_runs] = a1, ++n_runs;
    }
    for (int r = 0; r < n_runs; ++r) wanted += hi[r] - lo[r];
    if (n_runs > 0 && 2 * wanted <= D) {
      n_pass = 1;
      for (int r = 0; r < n_runs; ++r)
        for (int off = lo[r]; off < hi[r]; off += 48) {
          per = std::min(48, (hi[r] - off + 15) & ~15);
          rc = PR_EUNSUPPORTED;
          switch (per <= 16 ? 16 : per <= 32 ? 32 : 48) {
            MM_CASE(16, MM_COLOUR, off) MM_CASE(32, MM_COLOUR, off) MM_CASE(48, MM_COLOUR, off)
          }
          if (rc) return rc;
        }
      return rc;
    }
  }
  const int n_win = (D + widest - 1) / widest;
  per = ((D + n_win - 1) / n_win + 15) & ~15;
  n_pass = (D + per - 1) / per;
  rc = PR_EUNSUPPORTED;
  switch (per <= 16 ? 16 : per <= 32 ? 32 : 48) {
    MM_CASE(16, MM_COLOUR, 0) MM_CASE(32, MM_COLOUR, 0) MM_CASE(48, MM_COLOUR, 0)
  }
  return rc;
#undef MM_CASE
#undef MM_ARGS
}

}  // namespace pr
