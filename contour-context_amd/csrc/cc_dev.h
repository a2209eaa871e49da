// Shared device-side definitions of the MI355X contour-context kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cont2_amd.h"

// Launch-time constants derived on the host from cc_manager_cfg_t exactly as the reference's
// ContourManager constructor derives them (contour_mng.h:478-498, :448-472).
struct cc_dev_cfg {
  float x_lo, x_hi, y_lo, y_hi;  // x_min_+padding, x_max_-padding, ... evaluated in f32 on the host
  float blind_sq;
  float reso_row, reso_col;
  float lidar_height;
  int n_row, n_col, half_row, half_col, n_cell;
  float lv_grads[CC_NLEV];
  int min_cont_key_cnt, min_cont_cell_cnt, piv_firsts, dist_firsts;
  float roi_radius;
  int min_cell_cov;
  float point_sigma, com_bias_thres;
  float inv_row, inv_col;  // 1 / reso, exact when reso_pow2
  int reso_pow2;           // both resolutions are powers of two (the shipped 1.0 and the paper's 2.0 are)
};

#define CC_BEV_EMPTY (-1000.0f)  // VAL_ABS_INF_, contour_mng.h:418,488

// order-preserving map f32 -> u32 (total order of finite floats), used for LDS atomicMax on heights
__device__ __forceinline__ unsigned cc_fkey(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);  // negative: ~b, else b | sign bit
}
__device__ __forceinline__ float cc_funkey(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// hashPointToImage (contour_mng.h:448-463).  Returns cell index or -1 (rejected or row 0:
// makeBEV only uses points with rc.first > 0, contour_mng.h:515).
// POW2: the resolutions are powers of two, so x * (1 / reso) and x / reso are the correctly rounded value of the same real
// number -- bit-identical, and an IEEE f32 division is ~11 instructions against one.
template <bool POW2 = false>
__device__ __forceinline__ int cc_point_cell(const cc_dev_cfg &c, float x, float y) {
  // written so that a NaN coordinate is rejected (the reference's int(floor(NaN)) is undefined behaviour); identical to
  // `x < lo || x > hi || ...` for every other value.  Selects, no early return: K1 runs this per point and the branches an
  // early return compiles to cost more than the few instructions they skip (round 6).  A rejected point's row / col are
  // computed from whatever it holds and dropped (the float -> int conversion saturates).
  const bool ok = (x >= c.x_lo) & (x <= c.x_hi) & (y >= c.y_lo) & (y <= c.y_hi) & !((y * y + x * x) < c.blind_sq);  // `&`: no short-circuit branches
  const int row = (int)floorf(POW2 ? x * c.inv_row : x / c.reso_row) + c.half_row;
  const int col = (int)floorf(POW2 ? y * c.inv_col : y / c.reso_col) + c.half_col;
  const int cell = __mul24(row, c.n_col) + col;  // |row| <= 75 + 150 for an accepted point, n_col <= 150
  return (ok & (row > 0)) ? cell : -1;
}

// value of `v` in lane `src_lane` (wave-uniform index), e.g. the lane found by ffs of a ballot mask
__device__ __forceinline__ float cc_lane_bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
