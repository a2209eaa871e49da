// Per-contour descriptor maths on the device: ContourView::calcStatVals (contour.h:142-255),
// including a restatement of Eigen 3.3 SelfAdjointEigenSolver<Matrix2f>::compute (scale, trivial
// 2x2 tridiagonalisation, implicit symmetric QR with Wilkinson shift, ascending sort).
// Compiled with -ffp-contract=off: every operation below is an individually rounded IEEE f32/f64
// operation in the order the reference performs it.
#pragma once
#include "cc_dev.h"

// evals ascending; evecs column-major [v00 v10 v01 v11], column c <-> evals[c]
__device__ __forceinline__ void cc_eigen2f(float m00, float m10, float m11, float evals[2], float evecs[4]) {
  float d0 = m00, d1 = m11, e = m10;
  float scale = fabsf(d0);
  if (fabsf(e) > scale) scale = fabsf(e);
  if (fabsf(d1) > scale) scale = fabsf(d1);
  if (scale == 0.f) scale = 1.f;
  d0 /= scale;
  e /= scale;
  d1 /= scale;
  float q00 = 1.f, q01 = 0.f, q10 = 0.f, q11 = 1.f;
  const float considerAsZero = 1.17549435e-38f;
  const float precision = 2.f * 1.1920929e-07f;
  int iter = 0;
  while (true) {
    if (fabsf(e) <= (fabsf(d0) + fabsf(d1)) * precision || fabsf(e) <= considerAsZero) e = 0.f;
    if (e == 0.f) break;
    iter++;
    if (iter > 60) break;
    float td = (d0 - d1) * 0.5f;
    float mu = d1;
    if (td == 0.f) {
      mu -= fabsf(e);
    } else {
      float e2 = e * e;
      float ax = fabsf(td), ay = fabsf(e), p, qp;
      if (ax > ay) {
        p = ax;
        qp = ay / p;
      } else {
        p = ay;
        qp = ax / p;
      }
      float h = (p == 0.f) ? 0.f : p * sqrtf(1.f + qp * qp);
      if (e2 == 0.f)
        mu -= (e / (td + (td > 0.f ? 1.f : -1.f))) * (e / h);
      else
        mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = d0 - mu;
    float z = e;
    float c, s;
    if (z == 0.f) {
      c = x < 0.f ? -1.f : 1.f;
      s = 0.f;
    } else if (x == 0.f) {
      c = 0.f;
      s = z < 0.f ? 1.f : -1.f;
    } else if (fabsf(x) > fabsf(z)) {
      float t = z / x;
      float u = sqrtf(1.f + t * t);
      if (x < 0.f) u = -u;
      c = 1.f / u;
      s = -t * c;
    } else {
      float t = x / z;
      float u = sqrtf(1.f + t * t);
      if (z < 0.f) u = -u;
      s = -1.f / u;
      c = -t * s;
    }
    float sdk = s * d0 + c * e;
    float dkp1 = s * e + c * d1;
    d0 = c * (c * d0 - s * e) - s * (c * e - s * d1);
    d1 = s * sdk + c * dkp1;
    e = c * sdk - s * dkp1;
    float x0 = q00, y0 = q01, x1 = q10, y1 = q11;
    q00 = c * x0 - s * y0;
    q01 = s * x0 + c * y0;
    q10 = c * x1 - s * y1;
    q11 = s * x1 + c * y1;
  }
  if (d1 < d0) {
    float t = d0;
    d0 = d1;
    d1 = t;
    t = q00;
    q00 = q01;
    q01 = t;
    t = q10;
    q10 = q11;
    q11 = t;
  }
  evals[0] = d0 * scale;
  evals[1] = d1 * scale;
  evecs[0] = q00;
  evecs[1] = q10;
  evecs[2] = q01;
  evecs[3] = q11;
}

// RunningStatRecorder (contour.h:48-95) accumulated by the caller in raster order
struct cc_running_stat {
  int cnt;
  double ps_x, ps_y;
  double t_xx, t_xy, t_yy;
  float vol3;
  double tq_x, tq_y;
};

__device__ __forceinline__ bool cc_diff_perc(float a, float b, float perc) {
  return fabsf((a - b) / (a < b ? b : a)) > perc;  // tools/algos.h:13-15 (std::max(a,b) returns b if a<b)
}
__device__ __forceinline__ bool cc_diff_delt(float a, float b, float delta) { return fabsf(a - b) > delta; }

__device__ __forceinline__ void cc_calc_stat_vals(const cc_dev_cfg &cfg, const cc_running_stat &rec, int level, int poi_r,
                                                  int poi_c, cc_contour_t *o) {
  o->level = (int16_t)level;
  o->poi[0] = (int16_t)poi_r;
  o->poi[1] = (int16_t)poi_c;
  o->cell_cnt = (int16_t)rec.cnt;
  const float cntf = (float)rec.cnt;
  const float pm0 = (float)rec.ps_x / cntf, pm1 = (float)rec.ps_y / cntf;
  o->pos_mean[0] = pm0;
  o->pos_mean[1] = pm1;
  o->vol3_mean = rec.vol3 / cntf;
  const float com0 = (float)rec.tq_x / rec.vol3, com1 = (float)rec.tq_y / rec.vol3;
  o->com[0] = com0;
  o->com[1] = com1;
  o->pad_[0] = o->pad_[1] = 0;
  if (rec.cnt < cfg.min_cell_cov) {
    const float ss = 1.f * cfg.point_sigma * cfg.point_sigma;
    const float zz = 0.f * cfg.point_sigma * cfg.point_sigma;
    o->pos_cov[0] = ss;
    o->pos_cov[1] = zz;
    o->pos_cov[2] = zz;
    o->pos_cov[3] = ss;
    o->eig_vals[0] = cfg.point_sigma;
    o->eig_vals[1] = cfg.point_sigma;
    o->eig_vecs[0] = 1.f;
    o->eig_vecs[1] = 0.f;
    o->eig_vecs[2] = 0.f;
    o->eig_vecs[3] = 1.f;
    o->eccen = 0.f;
    o->ecc_feat = 0;
    o->com_feat = 0;
  } else {
    const float denom = (float)(rec.cnt - 1);
    const float c00 = ((float)rec.t_xx - (pm0 * pm0) * cntf) / denom;
    const float c01 = ((float)rec.t_xy - (pm0 * pm1) * cntf) / denom;
    const float c10 = ((float)rec.t_xy - (pm1 * pm0) * cntf) / denom;
    const float c11 = ((float)rec.t_yy - (pm1 * pm1) * cntf) / denom;
    o->pos_cov[0] = c00;
    o->pos_cov[1] = c10;
    o->pos_cov[2] = c01;
    o->pos_cov[3] = c11;
    float ev[2], vec[4];
    cc_eigen2f(c00, c01, c11, ev, vec);  // selfadjointView<Upper>: the (0,1) entry is used for both off-diagonals
    if (ev[0] < cfg.point_sigma) ev[0] = cfg.point_sigma;
    if (ev[1] < cfg.point_sigma) ev[1] = cfg.point_sigma;
    o->eig_vals[0] = ev[0];
    o->eig_vals[1] = ev[1];
    o->eig_vecs[0] = vec[0];
    o->eig_vecs[1] = vec[1];
    o->eig_vecs[2] = vec[2];
    o->eig_vecs[3] = vec[3];
    o->eccen = sqrtf(ev[1] * ev[1] - ev[0] * ev[0]) / ev[1];
    o->ecc_feat = (rec.cnt > 5 && cc_diff_perc(ev[0], ev[1], 0.2f) && ev[1] > 2.5f) ? 1 : 0;
    const float dx = com0 - pm0, dy = com1 - pm1;
    o->com_feat = (sqrtf(dx * dx + dy * dy) > cfg.com_bias_thres) ? 1 : 0;
  }
}

// std::atan2(float, float) of the BCI build (contour_mng.h:860: RelativePoint::theta) -- glibc's atan2f, i.e. fdlibm's
// e_atan2f.c / s_atanf.c (argument reduction to four intervals, odd/even degree-11 polynomial, hi/lo table), restated
// operation for operation in f32: the device library's atan2f is as accurate but not the same function, and one ulp of
// theta can move a check across the pi/16 window of BCI::checkConstellSim.  Checked bit for bit against glibc 2.35's
// atan2f on 2e8 arguments (1e8 of them differences of BEV coordinates), tests/test_atan2f_replica.py.
__device__ __forceinline__ float cc_atanf_fdlibm(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f,  -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                        6.6610731184e-02f,  -5.8335702866e-02f, 4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  int id;
  if (ix < 0x3ee00000) {  // |x| < 0.4375
    if (ix < 0x31000000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {  // |x| < 1.1875
      if (ix < 0x3f300000) {
        id = 0;
        x = (2.0f * x - 1.0f) / (2.0f + x);
      } else {
        id = 1;
        x = (x - 1.0f) / (x + 1.0f);
      }
    } else if (ix < 0x401c0000) {  // |x| < 2.4375
      id = 2;
      x = (x - 1.5f) / (1.0f + 1.5f * x);
    } else {
      id = 3;
      x = -1.0f / x;
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];
  const float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
// glibc's acosf (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm routine; glibc 2.35), operation for operation: the orientation
// filter of checkConstellCorrespSim compares two such angles with pi / 6 (contour_mng.h:1195-1210), and the device
// library's acosf differs from glibc's in the last bit now and then -- once in ~10^7 comparisons a pair is kept on one
// side and dropped on the other (round 6: drive 131409 of tests/fuzz_gpu_query.py, one check of 100 000 queries).  Bit-identical
// to this libm on every third float of [-1, 1] (profiles/r6/acosf_replica_check.c) and in tests/test_atan2f_replica.py.
__device__ __forceinline__ float cc_acosf_fdlibm(float x) {
  const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
              pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
              pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;  // |x| == 1
  if (ix > 0x3f800000) return (x - x) / (x - x);                     // |x| > 1: NaN
  if (ix < 0x3f000000) {                                             // |x| < 0.5
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) {  // x < -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  // x > 0.5
  const float z = (one - x) * 0.5f;
  const float s = sqrtf(z);
  const float df = __int_as_float(__float_as_int(s) & (int)0xfffff000);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}
__device__ __forceinline__ float cc_atan2f_fdlibm(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return cc_atanf_fdlibm(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000 || iy == 0x7f800000) {  // an infinite coordinate difference: not reachable from BEV cells; fdlibm's table
    if (ix == 0x7f800000 && iy == 0x7f800000) return m == 0 ? 0.78539818525f + tiny : m == 1 ? -0.78539818525f - tiny : m == 2 ? 3.0f * 0.78539818525f + tiny : -3.0f * 0.78539818525f - tiny;
    if (ix == 0x7f800000) return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  }
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60)
    z = 0.0f;
  else
    z = cc_atanf_fdlibm(fabsf(y / x));
  if (m == 0) return z;
  if (m == 1) return __int_as_float(__float_as_int(z) ^ (int)0x80000000);
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}
