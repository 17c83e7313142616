// Shared helpers for libpogs_rast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pogs_rast.h"

namespace pr {

// Records the text of a failed HIP call for pr_last_hip_error().
void set_hip_error(hipError_t e, const char *where);

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

__device__ __forceinline__ float wave_sum_all(float v) {
  v = wave_sum_to_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace pr
