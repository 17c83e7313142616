// Shared helpers for libpogs_rast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pogs_rast.h"

namespace pr {

// Records the text of a failed HIP call for pr_last_hip_error().
void set_hip_error(hipError_t e, const char *where);
// Detail for a PR_EUNSUPPORTED / PR_EINVAL return, read through the same accessor (pr_last_hip_error).
void set_error_text(const char *text);

inline int check_launch(const char *where) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_hip_error(e, where);
    return PR_EHIP;
  }
  return PR_OK;
}

#define PR_HIP(call)                          \
  do {                                        \
    hipError_t _e = (call);                   \
    if (_e != hipSuccess) {                   \
      ::pr::set_hip_error(_e, #call);         \
      return PR_EHIP;                         \
    }                                         \
  } while (0)

constexpr int WAVE = 64;

// Tuning aids (POGS_RAST_* / POGS_MLP_VARIANT environment variables): read ONCE, when the library first needs
// them, never on the call path.  pr_reload_tuning() re-reads the environment (tests and A/B scripts that switch
// kernel variants inside one process call it after changing a variable).  Fields are plain ints written before
// any reader exists or under the caller's own serialisation of pr_reload_tuning against render calls.
struct Tuning {
  int fwd_ppl;      // POGS_RAST_FWD_PPL: force 1/2/4 pixels per lane in the forward compositor (0 = planner)
  int fwd_chunk;    // POGS_RAST_FWD_CHUNK: widest forward channel pass (0 = planner)
  int fwd_blocks;   // POGS_RAST_FWD_BLOCKS: forward compositor with per-4x4-block walks (16x16 tiles): -1 (unset) = planner (small
                    // images), 0 = never, 1 = ring form everywhere, 2 = batch-by-batch form everywhere
  int fwd_mfma;     // POGS_RAST_FWD_MFMA: 1 = split-bf16 matrix-core colour accumulation for 32..72-channel passes
  int bwd_ppl;      // POGS_RAST_BWD_PPL: as fwd_ppl for the backward compositor
  int bwd_mfma;     // POGS_RAST_BWD_MFMA: -1 (unset) = planner: the float32 matrix-core kernel on 8x8 regions wherever it wins,
                    // the bf16 matrix-core dot products for what is left of the geometry-only backward at 1 pixel per lane;
                    // 0 = vector kernels only; 1 = as -1 without the float32 matrix-core kernel; 2 = both contractions as
                    // split-bf16 matrix products at 1 pixel per lane; 3 = deferred split-bf16 colour gradient at 2 pixels
                    // per lane (32/36-channel passes); 4 = colour gradients in a split-bf16 kernel of their own (up to 48
                    // channels); 5 = the float32 matrix-core kernel for every call it is built for
  int bwd_window;   // POGS_RAST_BWD_WINDOW: widest channel window (16/32/48) of the colour-gradient launches (0 = planner)
  int mlp_variant;  // POGS_MLP_VARIANT: -1 = default
  int depth_sort;   // POGS_RAST_DEPTH_SORT: 0 = chained-scan radix sort (onesweep.h), 1 = rocPRIM's radix sort for both binning
                    // sorts, 2 = sample sort of the depth keys between 32 K and 1.25 M Gaussians
};
const Tuning &tuning();

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// DPP helpers: v + (v moved across lanes by `ctrl`), lanes without a source add 0.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}

// Sum over the 64 lanes of a wavefront; the total is valid in lane 63 (only).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111>(v);  // row_shr:1
  v = dpp_add<0x112>(v);  // row_shr:2
  v = dpp_add<0x114>(v);  // row_shr:4
  v = dpp_add<0x118>(v);  // row_shr:8   -> lane 15 of every row holds the row total
  v = dpp_add<0x142>(v);  // row_bcast:15 -> rows 1,3 += previous row's lane 15
  v = dpp_add<0x143>(v);  // row_bcast:31 -> rows 2,3 += lane 31
  return v;
}

// Sum over the 16 lanes of each DPP row; the row total is valid in the row's last lane (15, 31, 47, 63).
__device__ __forceinline__ float row_sum_to_lane15(float v) {
  v = dpp_add<0x111>(v);
  v = dpp_add<0x112>(v);
  v = dpp_add<0x114>(v);
  v = dpp_add<0x118>(v);
  return v;
}

// "Transposed" pair reductions built on gfx950's lane-swap instructions: two values go in, ONE register
// comes out in which half of the lanes carry a partial sum of `a` and the other half one of `b` -- the
// number of live values halves at every level instead of every lane carrying every sum.
//   swap32_add: lanes 0..31 <- a[l] + a[l+32];   lanes 32..63 <- b[l-32] + b[l]
__device__ __forceinline__ float swap32_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false,
                                            false);
  unsigned r0 = r[0], r1 = r[1];
  // Keep the two results as distinct values: without this barrier hipcc (ROCm 7.2) folds the float add of
  // the intrinsic's two results into r0 + r0.
  asm("" : "+v"(r0), "+v"(r1));
  return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}
//   swap16_add: rows 0,2 <- a.row(r) + a.row(r+1);   rows 1,3 <- b.row(r-1) + b.row(r)
__device__ __forceinline__ float swap16_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false,
                                            false);
  unsigned r0 = r[0], r1 = r[1];
  asm("" : "+v"(r0), "+v"(r1));
  return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
}

// Fold four per-lane values across the four DPP rows.  On return every lane of row r holds the sum over
// the 4 lanes {c, c+16, c+32, c+48} (c = its column) of value slot(r), slot = {0, 2, 1, 3}; what is left
// is the sum over the 16 columns of a row.
__device__ __forceinline__ float quad_fold_rows(float v0, float v1, float v2, float v3) {
  return swap16_add(swap32_add(v0, v1), swap32_add(v2, v3));
}

// Full reduction of four values: the LAST lane of DPP row r (lane 16r+15) holds the wave total of
// value slot(r).
__device__ __forceinline__ float quad_reduce(float v0, float v1, float v2, float v3) {
  return row_sum_to_lane15(quad_fold_rows(v0, v1, v2, v3));
}

// Fire-and-forget float atomic add to global memory.  Issued through inline asm on purpose: hipcc's
// wait-count pass otherwise puts an `s_waitcnt vmcnt(0)` at the head of a loop whose body ends in an atomic,
// so every iteration would wait out the previous iteration's atomic round trip (1-3 us under load) although
// nothing ever reads its result.  The instruction still counts in vmcnt in issue order, so the compiler's
// own (now conservative) waits for later loads stay correct.
__device__ __forceinline__ void atomic_add_noret(float *addr, float v) {
  asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
}

// The same with a scalar (wave-uniform) base address and a 32-bit byte offset per lane: no 64-bit address arithmetic,
// and the base costs no vector registers.  The `s_nop 4` is part of the instruction's contract here: a vector-memory
// instruction must not read a scalar register within five wait states of a VECTOR instruction writing it
// (v_readlane_b32 restoring a spilled base, v_readfirstlane_b32), and the compiler's hazard pass, which inserts those
// wait states for its own instructions, does not look inside inline assembly -- without it the access can go out
// with a stale half of the base address.
__device__ __forceinline__ void atomic_add_noret_s(float *base, unsigned byte_off, float v) {
  asm volatile("s_nop 4\n\tglobal_atomic_add_f32 %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(base) : "memory");
}
// For values that come straight out of matrix instructions: the hazard pass also does not know that the inline
// assembly above reads them, and a vector-memory read of a matrix result needs up to 18 wait states after the
// instruction that produces it.
__device__ __forceinline__ void settle_mfma_results() {
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;  // smallest alpha the compositor lets contribute

// Exact footprint test of one Gaussian against a pixel-centre rectangle [rx0,rx1] x [ry0,ry1]: the
// minimum of the (convex) quadratic form sigma over the rectangle bounds the largest alpha any pixel of
// the region can see.  If even that is below 1/255 the Gaussian cannot contribute to this region (the
// square 3-sigma bounding box that built the tile lists is far looser for elongated or faint Gaussians).
// Conservative by a margin of 1e-3 in the exponent; non-positive-definite conics are always kept.
// Smallest value of sigma = (a dx^2 + c dy^2)/2 + b dx dy over the rectangle, for a positive definite conic.
__device__ __forceinline__ float min_sigma_in_rect(float gx, float gy, float a, float b, float c, float rx0, float ry0,
                                                   float rx1, float ry1) {
  const float xl = rx0 - gx, xr = rx1 - gx, yl = ry0 - gy, yr = ry1 - gy;
  const float X = fminf(fmaxf(0.f, xl), xr), Y = fminf(fmaxf(0.f, yl), yr);  // nearest rectangle point per axis
  float smin = 0.f;
  if (X != 0.f || Y != 0.f) {
    float s1 = 3.0e38f, s2 = 3.0e38f;
    if (X != 0.f) {
      const float uy = fminf(fmaxf(-b * X / c, yl), yr);
      s1 = 0.5f * (a * X * X + c * uy * uy) + b * X * uy;
    }
    if (Y != 0.f) {
      const float ux = fminf(fmaxf(-b * Y / a, xl), xr);
      s2 = 0.5f * (a * ux * ux + c * Y * Y) + b * ux * Y;
    }
    smin = fmaxf(fminf(s1, s2), 0.f);
  }
  return smin;
}
__device__ __forceinline__ bool touches_region(float gx, float gy, float o, float a, float b, float c, float rx0,
                                               float ry0, float rx1, float ry1) {
  if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) return true;
  const float smin = min_sigma_in_rect(gx, gy, a, b, c, rx0, ry0, rx1, ry1);
  return o * __builtin_amdgcn_exp2f(1e-3f - smin * LOG2E) >= ALPHA_MIN;
}
// The same test for a conic that is already multiplied by log2(e) (al = a log2e, ...), as the compositors' staged
// records hold it.
__device__ __forceinline__ bool touches_region_log2(float gx, float gy, float o, float al, float bl, float cl, float rx0,
                                                    float ry0, float rx1, float ry1) {
  if (!(al > 0.f && cl > 0.f && al * cl - bl * bl > 0.f)) return true;
  const float smin = min_sigma_in_rect(gx, gy, al, bl, cl, rx0, ry0, rx1, ry1);
  return o * __builtin_amdgcn_exp2f(1e-3f - smin) >= ALPHA_MIN;
}

__device__ __forceinline__ float wave_sum_all(float v) {
  v = wave_sum_to_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace pr
