// Shared pieces of the compositing kernels (forward and backward) for gfx950.
//
// Work decomposition: ONE WAVEFRONT owns one rectangular pixel region of one tile and walks that tile's
// depth-sorted intersection list on its own -- no workgroup barrier anywhere.  A 16x16 tile is covered
// by 1, 2 or 4 wavefronts (4, 2 or 1 pixels per lane); an 8x8 tile by one wavefront (1 pixel per lane).
// Several pixels per lane amortise the per-Gaussian broadcast reads from LDS (the same 16 bytes
// delivered to all lanes costs a full LDS cycle group) over 4x the FMAs, which is what keeps the loop
// VALU-bound rather than LDS-bound.  Gaussians are staged 64 at a time (one per lane) into the
// wavefront's private LDS slice; termination and "does anyone need this Gaussian" are wave ballots.
#pragma once
#include "common.h"

namespace pr {

// Two-float vector: arithmetic on it compiles to packed-FP32 instructions (v_pk_fma_f32 ...), i.e. one
// issue slot for two lanes of work.  A wavefront issues at most one VALU instruction every four cycles,
// so halving the instruction count of the channel loops is worth more than it looks.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

constexpr float LN2 = 0.6931471805599453f;
constexpr float ALPHA_MAX = 0.999f;
constexpr float T_STOP = 1e-4f;

// Pixel region owned by one wavefront, and the lane -> pixel map inside it.  Lane l sits at column
// l % LX, row l / LX of the region's lane grid (pitch PX, PY pixels) and owns BW x BH pixels spaced SX, SY
// apart.  With several pixels per lane the lane grid is 16 x 4 and pixel slot j of every lane lies in the
// j-th 16x4 strip of the region, so one dx serves all of a lane's pixels.  A Gaussian that only reaches part
// of the tile leaves whole slots without a single active lane (2.8 of 4 slots are live on average at the
// benchmark shape).  Skipping slot by slot with wave-uniform branches was measured SLOWER than doing the
// arithmetic (~20 extra branches per Gaussian cost more issue slots than the skipped FMAs); what pays is ONE
// branch per Gaussian between "all four slots", "upper two" and "lower two" (HALF_SKIP) in the backward
// kernel (-2 %); the same three-way split pushed the forward kernel to 256 registers and one wave per SIMD.
constexpr bool HALF_SKIP = true;
template <int TS, int PPL>
struct Region;
template <>
struct Region<16, 4> {  // whole tile, lane = one pixel in each of the four 16x4 strips
  static constexpr int WPT = 1, BW = 1, BH = 4, LX = 16, RH = 16, PX = 1, PY = 1, SX = 1, SY = 4;
};
template <>
struct Region<16, 2> {  // half tile (16x8), lane = one pixel in each of its two strips
  static constexpr int WPT = 2, BW = 1, BH = 2, LX = 16, RH = 8, PX = 1, PY = 1, SX = 1, SY = 4;
};
template <>
struct Region<16, 1> {  // quarter tile (16x4), lane = 1 pixel
  static constexpr int WPT = 4, BW = 1, BH = 1, LX = 16, RH = 4, PX = 1, PY = 1, SX = 1, SY = 1;
};
template <>
struct Region<8, 1> {  // whole 8x8 tile, lane = 1 pixel
  static constexpr int WPT = 1, BW = 1, BH = 1, LX = 8, RH = 8, PX = 1, PY = 1, SX = 1, SY = 1;
};

// Bijective block-id remap: blocks that share an XCD (equal blockIdx % 8 under round-robin dispatch)
// get a contiguous range of regions, so neighbouring tiles -- which share Gaussians -- hit one L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nb) {
  unsigned q = nb >> 3, r = nb & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Which region a workgroup (= one wavefront) processes.  Without an explicit tile order, blocks that share an
// XCD take a contiguous range of regions (neighbouring tiles share Gaussians -> one L2).  With one, launch slot
// s processes tile tile_order[s] -- the caller lists the tiles heaviest first so that the last wavefronts of the
// launch are the cheapest ones -- and the WPT wavefronts that share a tile are blocks b, b + 8, ..: the same XCD
// under round-robin dispatch, a few blocks apart in time, so the tile's records and colour rows leave HBM once
// and the other wavefronts' gathers hit that XCD's L2 (placement is for speed only; nothing depends on it).
struct RegionId {
  int tile_flat, sub;
  bool valid;
};
template <int WPT>
__device__ __forceinline__ RegionId region_of(const int32_t *__restrict__ tile_order, int n_regions) {
  RegionId r;
  if (tile_order) {
    const unsigned b = blockIdx.x, x = b & 7u, i = b >> 3;
    const int slot = (int)((i / WPT) * 8u + x);
    r.sub = (int)(i % WPT);
    r.valid = slot < n_regions / WPT;
    r.tile_flat = r.valid ? tile_order[slot] : 0;
  } else {
    const int region = (int)xcd_remap(blockIdx.x, gridDim.x);
    r.valid = region < n_regions;
    r.tile_flat = region / WPT;
    r.sub = region % WPT;
  }
  return r;
}
// Grid size that goes with region_of: with a tile order the tile slots are dealt to the 8 XCDs in whole groups.
inline unsigned region_grid(int64_t n_regions, int wpt, bool ordered) {
  if (!ordered) return (unsigned)n_regions;
  const int64_t n_tiles = n_regions / wpt;
  return (unsigned)((n_tiles + 7) / 8 * 8 * wpt);
}

// Wavefronts per workgroup.  One: a workgroup's LDS and wave slots are only handed to the next workgroup
// when ALL of its waves have finished, so with tiles of unequal length multi-wave workgroups leave slots
// idle behind their slowest tile; nothing here needs a workgroup (no barrier, LDS slices are per wave).
constexpr int WPB = 1;

constexpr int round4(int x) { return (x + 3) & ~3; }
// LDS row stride (floats) for staged colours; rows that are a multiple of 32 dwords would put every
// lane's 16-byte store on the same banks.
constexpr int color_row(int cd) { return round4(cd) % 32 == 0 ? round4(cd) + 4 : round4(cd); }

// BWD adds the backward kernel's reduction area; the forward kernel leaves it out (LDS per wavefront decides
// how many wavefronts a CU holds).
// Intersections staged per batch.  A full wavefront's worth (one per lane) with 4 pixels per lane; half of
// that with fewer pixels per lane, where the smaller LDS slice lets the CU hold the extra wavefronts the
// smaller register footprint allows.
constexpr int batch_size(int ppl) { return ppl >= 4 ? WAVE : WAVE / 2; }

template <int CD, bool BWD, int BS>
struct alignas(16) WaveLds {
  float4 g4[BS];                // x, y, opacity, A' (= -0.5 a log2e)
  float2 g2[BS];                // B' (= -b log2e), C' (= -0.5 c log2e)
  int32_t ids[BS];              // flat Gaussian index
  float col[BS * color_row(CD)];
  // backward: row-folded partial sums of the per-Gaussian values, [quad][lane] with a 68-float pitch so
  // that the 16-float row segments read back by consecutive lanes fall on distinct banks
  float red[BWD ? (round4(CD) / 4 + 2) * 68 : 4];
};

// The D composited channels of a Gaussian may live in up to three tensors laid side by side in channel order
// (e.g. [rgb from SH | per-Gaussian features | depth]): segment i supplies channels [begin[i], begin[i+1]) from
// rows of pitch ld[i]; a shared segment is indexed by n alone (one row per Gaussian for all cameras), the
// others by the flat index c*N + n.  Unused trailing segments have begin = INT_MAX.  The compositor gathers
// from (and the backward scatters into) the tensors where they are, so no [C,N,D] copy is ever assembled.
struct ColorSegs {
  const float *ptr[3];
  int begin[3], ld[3], shared[3], n;
};
struct GradSegs {  // same layout for the colour gradients; ptr NULL = this segment needs no gradient
  float *ptr[3];
  int begin[3], ld[3], shared[3], n;
  int geom_ld;  // row pitch (floats) of the packed geometry gradients v_geom[C*N, geom_ld]: 8, or more when the caller
                // interleaves them with narrow colour gradients so that one Gaussian's values share a 64-byte segment
};

inline void set_geom_ld(ColorSegs &) {}
inline void set_geom_ld(GradSegs &g) { g.geom_ld = 8; }
// Host side: the kernels' segment tables from the C-ABI arguments.  Without segments the single [C,N,D]
// tensor is segment 0.  Widths must be positive and add up to D.
template <class Segs, class Ptr, class AbiSeg>
inline int make_segs(Ptr whole, int D, const AbiSeg *segments, int n_segments, bool need_ptrs, bool null_ok,
                     Segs &out) {
  for (int i = 0; i < 3; ++i) {
    out.ptr[i] = nullptr;
    out.begin[i] = 0x7fffffff;
    out.ld[i] = 0;
    out.shared[i] = 0;
  }
  set_geom_ld(out);
  if (n_segments < 0 || n_segments > 3) return PR_EINVAL;
  if (n_segments == 0) {
    if (need_ptrs && !whole && !null_ok) return PR_EINVAL;
    out.ptr[0] = whole;
    out.begin[0] = 0;
    out.ld[0] = D;
    out.n = 1;
    return PR_OK;
  }
  if (!segments) return PR_EINVAL;
  int at = 0;
  for (int i = 0; i < n_segments; ++i) {
    if (segments[i].width <= 0 || segments[i].ld < segments[i].width) return PR_EINVAL;
    if (need_ptrs && !segments[i].ptr && !null_ok) return PR_EINVAL;
    out.ptr[i] = segments[i].ptr;
    out.begin[i] = at;
    out.ld[i] = segments[i].ld;
    out.shared[i] = segments[i].shared != 0;
    at += segments[i].width;
  }
  if (at != D) return PR_EINVAL;
  out.n = n_segments;
  return PR_OK;
}
inline int make_color_segs(const float *colors, int D, const pr_color_segment *segments, int n_segments,
                           bool need_ptrs, ColorSegs &out) {
  return make_segs(colors, D, segments, n_segments, need_ptrs, false, out);
}
inline int make_grad_segs(float *v_colors, int D, const pr_grad_segment *segments, int n_segments, GradSegs &out) {
  return make_segs(v_colors, D, segments, n_segments, true, true, out);
}

// raster_bwd_cmfma.hip: backward at 2 pixels per lane with the colour-gradient contraction on the matrix cores
// (passes of 32 or 36 channels, geometry and colour gradients both wanted)
int launch_bwd_cmfma(int cd, int C, int N, int D, const float *means2d, const float *conics, const ColorSegs &cs,
                     const float *opacities, const float *backgrounds, int width, int height, int tile_w, int tile_h,
                     const int32_t *isect_offsets, const int32_t *tile_order, const int32_t *flatten_ids,
                     int64_t n_isects, const float *render_alphas, const int32_t *last_ids,
                     const float *render_colors, int ed_channel, const float *v_render_colors,
                     const float *v_render_alphas, int absgrad, float *v_geom, const GradSegs &gs,
                     hipStream_t stream);

// colour gradients alone (any pass of up to 48 channels), contraction on the matrix cores
int launch_bwd_colour_mfma(int C, int N, int D, const float *means2d, const float *conics, const float *opacities,
                           int width, int height, int tile_w, int tile_h, const int32_t *isect_offsets,
                           const int32_t *tile_order, const int32_t *flatten_ids, const float *render_alphas,
                           const int32_t *last_ids, int ed_channel, const float *v_render_colors, const GradSegs &gs,
                           hipStream_t stream);

// raster_bwd_mm.hip: backward with both channel contractions on the float32 matrix cores, one wavefront per 8x8
// pixel region: geometry gradients, colour gradients or both (wide passes as several launches)
bool bwd_mm_supports(int cd, int tile_size);
int launch_bwd_mm(int cd, int tile_size, bool geom_grad, bool color_grad, bool split, int C, int N, int D,
                  const float *means2d, const float *conics, const ColorSegs &cs, const float *opacities, const float *backgrounds,
                  int width, int height, int tile_w, int tile_h, const int32_t *isect_offsets,
                  const int32_t *tile_order, const int32_t *flatten_ids, int64_t n_isects,
                  const float *render_alphas, const int32_t *last_ids, const float *render_colors, int ed_channel,
                  const float *v_render_colors, const float *v_render_alphas, int absgrad, float *v_geom,
                  const GradSegs &gs, hipStream_t stream);

// raster_fwd_blocks.hip: forward compositor on 16x16 tiles whose wavefronts (one per half tile) walk a list per 4x4
// pixel block; passes of up to 48 channels, all passes of a call in one launch
bool fwd_blocks_supports(int cd, int tile_size);
int launch_fwd_blocks(int cd, bool ring, int C, int N, int D, int ch_off, int per, int n_pass, const float *means2d,
                      const float *conics, const ColorSegs &cs, const float *opacities, const float *backgrounds, int width,
                      int height, int tile_w, int tile_h, const int32_t *isect_offsets, const int32_t *tile_order,
                      const int32_t *flatten_ids, int64_t n_isects, int ed_channel, float *render_colors,
                      float *render_alphas, int32_t *last_ids, hipStream_t stream);

// The geometry of one batch of up to 64 intersections, one per lane, loaded ahead of time: the ids and the
// records they point to are two dependent global round trips, which the caller issues a whole batch early
// (right after staging the previous one) so that only the colour rows remain to be fetched at staging time.
struct BatchGeometry {
  int g;  // flat Gaussian index, -1 beyond the batch
  float2 xy;
  float ca, cb, cc, o;
};
__device__ __forceinline__ BatchGeometry prefetch_batch(int lane, int first, int count,
                                                        const int32_t *__restrict__ flatten_ids,
                                                        const float *__restrict__ means2d,
                                                        const float *__restrict__ conics,
                                                        const float *__restrict__ opacities) {
  BatchGeometry b;
  b.g = -1;
  b.xy = make_float2(0.f, 0.f);
  b.ca = b.cb = b.cc = b.o = 0.f;
  if (lane < count) {
    b.g = flatten_ids[first + lane];
    b.xy = *reinterpret_cast<const float2 *>(means2d + 2 * (int64_t)b.g);
    b.ca = conics[3 * (int64_t)b.g];
    b.cb = conics[3 * (int64_t)b.g + 1];
    b.cc = conics[3 * (int64_t)b.g + 2];
    b.o = opacities[b.g];
  }
  return b;
}

// Copy the CD channels from ch_off on of Gaussian g (flat index) from the colour tensors to `dst` (16-byte aligned,
// round4(CD) floats); channels past D are written as zeros.
template <int CD>
__device__ __forceinline__ void stage_color_row(float *dst, int g, const ColorSegs &cs, int N, int D, int ch_off) {
  int nch = D - ch_off;  // channels available from ch_off on
  const float *src = cs.ptr[0] + (int64_t)g * cs.ld[0] + ch_off;
  if (cs.n > 1 || cs.shared[0]) {
    // channels gathered from several tensors: one pointer per segment, pre-offset so that [channel] indexes it
    const int gn = g % N;
    const float *b0 = cs.ptr[0] + (int64_t)(cs.shared[0] ? gn : g) * cs.ld[0] - cs.begin[0];
    const float *b1 = cs.n > 1 ? cs.ptr[1] + (int64_t)(cs.shared[1] ? gn : g) * cs.ld[1] - cs.begin[1] : b0;
    const float *b2 = cs.n > 2 ? cs.ptr[2] + (int64_t)(cs.shared[2] ? gn : g) * cs.ld[2] - cs.begin[2] : b0;
#pragma unroll
    for (int k = 0; k < round4(CD); k += 4) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = ch_off + k + i;
        const float *b = ch >= cs.begin[2] ? b2 : ch >= cs.begin[1] ? b1 : b0;
        v[i] = (k + i < nch) ? b[ch] : 0.f;
      }
      *reinterpret_cast<float4 *>(dst + k) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else if (((cs.ld[0] | ch_off) & 3) == 0) {
#pragma unroll
    for (int k = 0; k < round4(CD); k += 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < nch) v = *reinterpret_cast<const float4 *>(src + k);
      *reinterpret_cast<float4 *>(dst + k) = v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < round4(CD); k += 4) {
      float4 v;
      v.x = (k + 0 < nch) ? src[k + 0] : 0.f;
      v.y = (k + 1 < nch) ? src[k + 1] : 0.f;
      v.z = (k + 2 < nch) ? src[k + 2] : 0.f;
      v.w = (k + 3 < nch) ? src[k + 3] : 0.f;
      *reinterpret_cast<float4 *>(dst + k) = v;
    }
  }
}

// Stage a prefetched batch: lane j tests Gaussian j against the region and, if it can contribute, writes its
// record and colour row to the wave's LDS slice.  Returns the wave-wide mask of staged Gaussians.
template <int CD, bool BWD, int BS>
__device__ __forceinline__ unsigned long long stage_batch(
    WaveLds<CD, BWD, BS> &L, int lane, const BatchGeometry &pre, const ColorSegs &cs, int N, int D, int ch_off, float rx0,
    float ry0, float rx1, float ry1) {
  constexpr int ROW = color_row(CD);
  bool keep = false;
  if (pre.g >= 0) {
    const int g = pre.g;
    const float2 xy = pre.xy;
    const float ca = pre.ca, cb = pre.cb, cc = pre.cc, o = pre.o;
    keep = touches_region(xy.x, xy.y, o, ca, cb, cc, rx0, ry0, rx1, ry1);
    if (keep) {
      L.g4[lane] = make_float4(xy.x, xy.y, o, -0.5f * LOG2E * ca);
      L.g2[lane] = make_float2(-LOG2E * cb, -0.5f * LOG2E * cc);
      L.ids[lane] = g;
      stage_color_row<CD>(L.col + lane * ROW, g, cs, N, D, ch_off);
    }
  }
  // Retire every staging load here.  Without this the compiler leaves some of them pending into the
  // per-Gaussian loop, and must then put a conservative `s_waitcnt vmcnt(0)` at that loop's head -- which
  // also waits out the previous iteration's gradient atomics (1-3 us each under load).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), other counters untouched
  return __builtin_amdgcn_ballot_w64(keep);
}

// ---- split-bf16 matrix-core helpers (v_mfma_f32_32x32x16_bf16: A[row l&31][k = 8(l>>5)+j], B[k = 8(l>>5)+j][col l&31],
// C: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5)) ----------------------------------------------------------------
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// (a, b) -> packed bf16 heads and packed bf16 residuals
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2v));
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a - ha, b - hb}, bf2v));
}
__device__ __forceinline__ void swap32(unsigned a, unsigned b, unsigned &lo_tile, unsigned &hi_tile) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  lo_tile = r[0];
  hi_tile = r[1];
  asm("" : "+v"(lo_tile), "+v"(hi_tile));
}
__device__ __forceinline__ f16v mfma_bf16(const unsigned (&a)[4], const unsigned (&b)[4], f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, u4v{a[0], a[1], a[2], a[3]}),
                                                 __builtin_bit_cast(bf8v, u4v{b[0], b[1], b[2], b[3]}), c, 0, 0, 0);
}

}  // namespace pr
