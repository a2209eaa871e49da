// K5 / K6 -- GMM-L2 correlation (GMMPair / ConstellCorrelation, correlation.h:42-238), the candidates fineOptimize refines
// and the final selection (contour_db.h:560-648).
//
//   cc_k_gmm_prep    per scan, once: the ellipses GMMPair's ctor selects and the scan's auto-correlation term
//   cc_k_gmm_init    every (query, candidate) problem: pair pre-selection and the initial correlation in one sweep over
//                    the ellipse grid; the selected pairs' (level, src, tgt) codes are filed in the chunk's code pool
//   cc_k_select      the <= max_fine_opt_ candidates per query the reference refines, split by pair count (the long ones by
//                    length class); hands out the pair pool
//   cc_k_gmm_refine  those problems: Ceres' LineSearchMinimizer restated (L-BFGS + Wolfe / cubic).  The first evaluation
//                    reads the problem's codes and files a 56-byte record per pair (the first CC_GMM_NL in LDS, the rest in
//                    the pair pool, unit by unit), every further evaluation streams those records.
//                    Two instances: 16 lanes per problem (4 problems per wave) and 64 lanes for the long pair lists.
//   cc_k_final       tidyUp compaction, fineOptimize ordering, result record
// The ellipse tables are read where they lie; the two pools are sized per query lane (cc_db_api.inc) and running out of
// either is reported (CC_ECAPACITY), never a shorter pair list.
#pragma once
#include "cc_dev.h"
#include "cc_group.h"
#include "cc_sort.h"
#include "cc_fmath.h"
#include "k_merge.h"

#define CC_GMM_ECAP_L CC_MAXC  // ellipses per level kept in a scan's correlation inputs (cc_gmm_feat): as many as the descriptor
                               // stores contours, so the correlation has no capacity of its own (round 3: 128 -- a street scene
                               // with ~100 contours on a level keeps up to ~150 ellipses, the KITTI-shaped world showed it)
// Refined by the 16-lane instance (four problems per wave) up to this many pairs, by the 64-lane instance above CC_GMM_MID_MAX_PAIRS.
// Round 5: once the term had lost half of its instructions, an ablation showed the pair arithmetic at 5-10 % of the refinement --
// what a problem costs is the SERIAL part every lane repeats (L-BFGS recursion, Wolfe search: IEEE divisions, square roots), and
// four problems share it on a 16-lane wave.  Measured per 1 024 headline queries with everything up to N pairs on the 16-lane
// instance: 96: 0.456 ms, 192: 0.419, 256: 0.354, 384: 0.358, 768: 0.354; KITTI-shaped (long lists): 0.64 up to 384, 1.06 at 512, 1.16 at 768 (a wave lasts as long as its longest member).
#ifndef CC_GMM_G16_MAX_PAIRS
#define CC_GMM_G16_MAX_PAIRS 96
#endif
// ... and the problems in between (CC_GMM_G16_MAX_PAIRS < pairs <= CC_GMM_MID_MAX_PAIRS) go where the chunk's problem COUNT says:
// with few problems to refine (an online sub-batch against a young database: fewer than the chip has wave slots) one wave
// each spreads them over the SIMDs; with many (the headline: ~8 000 per chunk) four to a wave share the serial code.
// Measured: everything up to 256 pairs on the 16-lane instance made the online replay 3.5 % slower (297 k against 307-310 k scans/s).
#ifndef CC_GMM_MID_MAX_PAIRS
#define CC_GMM_MID_MAX_PAIRS 256
#endif
#ifndef CC_GMM_PACK_MIN_PROBLEMS
#define CC_GMM_PACK_MIN_PROBLEMS 3072   // selected problems of a chunk from which the in-between ones are packed four to a wave
#endif
// (A 256-lane instance -- a workgroup per problem -- existed until round 6; measured on KITTI-shaped input it LOST twice: four
// waves repeat the serial line-search code, and a 256-thread workgroup at 250 registers leaves room for 512 problems on the
// chip instead of 2 048: cc_k_gmm_refine<64> 503 us -> <64> 175 + <256> 672 us per chunk in round 5, K5 0.65 -> 0.90 ms in round 6.)

// The long problems are listed by length class, longest first: the 64-lane refinement starts them in that order, so the
// waves that are still running when the launch runs dry hold the SHORT lists.  (In list order, a chunk's ~3 000 one-wave
// problems ran as one full round of 2 048 and a second, under-filled one: the launch lasted two problem lengths, 383 us,
// where its instructions fill the SIMDs for 225.)
#define CC_GMM_NCLS 8
#ifndef CC_GMM_WPE
#define CC_GMM_WPE 2  // waves per SIMD the refinement is compiled for
#endif
__device__ __forceinline__ int cc_gmm_len_class(int np) {
  return np > 1400 ? 0 : (np > 1100 ? 1 : (np > 950 ? 2 : (np > 850 ? 3 : (np > 750 ? 4 : (np > 650 ? 5 : (np > 500 ? 6 : 7))))));
}

struct cc_gmm_result {
  double corr_init;
  double corr_opt;
  double tf_opt[3];
  int optimized;   // 0: init correlation below the bar (no refinement)
  int iterations;
  int termination;
  int flags;       // bit0: a scan kept only its first CC_GMM_ECAP_L ellipses of a level (cannot happen while CC_GMM_ECAP_L == CC_MAXC), bit1: the pair pool was full,
                   // bit2: contour table truncated (CC_MAXC)
  int n_pairs;     // selected (src, tgt) ellipse pairs
  int code_seg;    // first segment of the problem's pair-code list in the chunk's code pool (cc_k_gmm_init), -1: none
};

struct cc_ell {  // values are f32 in the reference too (getManualCov, pos_mean_, cell_cnt_), widened to f64 at use
  float c00, c01, c10, c11, mx, my, w, maj;
};

// Per-scan inputs of the correlation, computed once per scan instead of once per (query, candidate) pair: the ellipses
// GMMPair's ctor selects (correlation.h:49-82) and the scan's auto-correlation term (correlation.h:102-119).
struct cc_gmm_feat {
  int n_ell[CC_GMM_LEVELS];
  int flags;  // bit0: more than CC_GMM_ECAP_L ellipses on a level, bit2: a needed contour was not stored in the descriptor
  int pad[3];
  double ac;  // sum over levels and ordered ellipse pairs (i, j) of the self term
  cc_ell ell[CC_GMM_LEVELS][CC_GMM_ECAP_L];
};

// One term of the auto-correlation sum (correlation.h:102-119): ellipses a, b of one level.
__device__ __forceinline__ double cc_gmm_self_term(const cc_ell &a, const cc_ell &b) {
  const double n00 = 2.0 * ((double)a.c00 + (double)b.c00), n01 = 2.0 * ((double)a.c01 + (double)b.c01);
  const double n10 = 2.0 * ((double)a.c10 + (double)b.c10), n11 = 2.0 * ((double)a.c11 + (double)b.c11);
  const double mx = (double)a.mx - (double)b.mx, my = (double)a.my - (double)b.my;
  const double det = n00 * n11 - n10 * n01, invdet = 1.0 / det;
  const double i00 = n11 * invdet, i10 = -n10 * invdet, i01 = -n01 * invdet, i11 = n00 * invdet;
  const double h0 = -0.5 * mx, h1 = -0.5 * my;
  const double r0 = h0 * i00 + h1 * i10, r1 = h0 * i01 + h1 * i11;
  return (double)a.w * (double)b.w / sqrt(det) * exp(r0 * mx + r1 * my);
}

// grid = n scans, block = CC_GMM_PREP_BLOCK (four waves per scan).
// The auto-correlation term sums n^2 ordered pairs per level; a street scene has n ~ 140 on the low levels (80 000 f64
// exp / sqrt / divisions per scan: on one wave that was a third of K2's own time on the KITTI-shaped workload).  Three facts
// take most of it away without touching the value beyond the rounding of the sum:
//   * term(i, j) == term(j, i) bit for bit (the sums of the covariances commute, the centre difference changes sign and
//     enters twice): the upper triangle is evaluated, off-diagonal terms count twice (an exact doubling);
//   * a term is at most exp(x) times the sum's own diagonal terms (prefactor <= geometric mean of the two self terms, by
//     Minkowski's determinant inequality), x <= -|m|^2 / (2 tr N) for a positive definite N: pairs with |m|^2 > 200 tr N
//     (x < -100, e^-100 = 4e-44) cannot reach the last bit of an f64 sum and are dropped by an f32 test -- only where N is
//     safely positive definite in f32 (det > 1e-4 tr^2); degenerate ellipses go the exact way and keep whatever the reference
//     makes of them;
//   * the pairs that stay (a fifth on street scenes) are queued per wave in LDS and evaluated 64 at a time, so the f64
//     code runs on full waves.
#define CC_GMM_PREP_BLOCK 256
__global__ void __launch_bounds__(CC_GMM_PREP_BLOCK)
cc_k_gmm_prep(const cc_scan_desc_t *__restrict__ desc, int n, cc_gmm_feat *__restrict__ feat) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = CC_GMM_PREP_BLOCK / 64;
  if ((int)blockIdx.x >= n) return;
  const cc_scan_desc_t *d = desc + blockIdx.x;
  cc_gmm_feat *F = feat + blockIdx.x;
  __shared__ cc_ell E[CC_GMM_ECAP_L];
  __shared__ unsigned s_q[NW][128];  // per wave: pairs waiting for the exact evaluation, (i << 16) | j, a ring
  __shared__ double s_part[NW];
  int flags = 0;
  double acc = 0;
  for (int li = 0; li < CC_GMM_LEVELS; li++) {
    const int lev = li + 1;  // GMMOptConfig::levels_ = {1,2,3,4}
    const int full = d->layer_cell_cnt[lev];
    const int ncont = d->n_cont[lev], nst = d->n_stored[lev];
    // contours in sorted order until >= 95 % of the level's cells: contour j is used iff the cells before it are < 95 %
    // (every wave works this out for itself: the same few loads, no hand-over)
    int n_use = 0, run = 0;
    for (int j0 = 0; j0 < ncont; j0 += 64) {
      const int j = j0 + lane;
      const int cnt = (j < ncont && j < nst) ? d->cont[lev][j].cell_cnt : 0;
      int incl = cnt;
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      const int before = run + incl - cnt;
      const bool use = j < ncont && !((double)before * 1.0 / (double)full >= 0.95);
      const unsigned long long m = __ballot(use);
      if (__ballot(use && j >= nst)) flags |= 4;
      n_use += __popcll(m);
      run += __shfl(incl, 63);
      if (m != ~0ull) break;  // the used contours are a prefix
    }
    if (n_use > nst) n_use = nst;
    if (n_use > CC_GMM_ECAP_L) {
      n_use = CC_GMM_ECAP_L;
      flags |= 1;
    }
    __syncthreads();  // the previous level's reads of E are done
    for (int j = tid; j < n_use; j += CC_GMM_PREP_BLOCK) {
      const cc_contour_t &cv = d->cont[lev][j];
      // getManualCov (contour.h:376-378) in f32; the reference then casts to double
      const float v00 = cv.eig_vecs[0], v10 = cv.eig_vecs[1], v01 = cv.eig_vecs[2], v11 = cv.eig_vecs[3];
      const float e0 = cv.eig_vals[0], e1 = cv.eig_vals[1];
      const float a00 = v00 * e0, a01 = v01 * e1, a10 = v10 * e0, a11 = v11 * e1;
      cc_ell e;
      e.c00 = a00 * v00 + a01 * v01;
      e.c01 = a00 * v10 + a01 * v11;
      e.c10 = a10 * v00 + a11 * v01;
      e.c11 = a10 * v10 + a11 * v11;
      e.mx = cv.pos_mean[0];
      e.my = cv.pos_mean[1];
      e.w = (float)cv.cell_cnt;
      e.maj = sqrtf(e1);
      E[j] = e;
      F->ell[li][j] = e;
    }
    __syncthreads();
    // upper triangle as a rectangle: rows r and n - 1 - r together hold n + 1 entries (j >= i)
    const int nrow = (n_use + 1) >> 1, wid = n_use + 1, tot = nrow * wid;
    int qh = 0, qn = 0;  // this wave's ring: head, entries (wave-uniform)
    for (int i0 = wave * 64; i0 < tot; i0 += CC_GMM_PREP_BLOCK) {
      const int idx = i0 + lane;
      bool go = false;
      int i = 0, j = 0;
      if (idx < tot) {
        const int r = idx / wid, c = idx - r * wid;
        if (c < n_use - r) {
          i = r;
          j = r + c;
          go = true;
        } else if (n_use - 1 - r != r) {  // the middle row of an odd n has no partner
          i = n_use - 1 - r;
          j = i + (c - (n_use - r));
          go = true;
        }
      }
      if (go) {
        const cc_ell a = E[i], b = E[j];
        const float dx = a.mx - b.mx, dy = a.my - b.my;
        const float t00 = a.c00 + b.c00, t11 = a.c11 + b.c11, t01 = a.c01 + b.c01, t10 = a.c10 + b.c10;
        const float tr = t00 + t11, det = t00 * t11 - t01 * t10;  // of N / 2
        // |m|^2 > 200 tr N = 400 tr(N / 2); NaN anywhere fails the comparisons and keeps the pair
        if (dx * dx + dy * dy > 400.f * tr && det > 1e-4f * tr * tr) go = false;
      }
      const unsigned long long m = __ballot(go);
      if (go) s_q[wave][(qh + qn + __popcll(m & ((1ull << lane) - 1ull))) & 127] = ((unsigned)i << 16) | (unsigned)j;
      qn += __popcll(m);
      cc_wave_sync();
      if (qn >= 64) {
        const unsigned e = s_q[wave][(qh + lane) & 127];
        const int ei = (int)(e >> 16), ej = (int)(e & 0xFFFFu);
        const double t = cc_gmm_self_term(E[ei], E[ej]);
        acc += ei == ej ? t : 2.0 * t;
        qh = (qh + 64) & 127;
        qn -= 64;
        cc_wave_sync();
      }
    }
    if (lane < qn) {
      const unsigned e = s_q[wave][(qh + lane) & 127];
      const int ei = (int)(e >> 16), ej = (int)(e & 0xFFFFu);
      const double t = cc_gmm_self_term(E[ei], E[ej]);
      acc += ei == ej ? t : 2.0 * t;
    }
    cc_wave_sync();
    if (tid == 0) F->n_ell[li] = n_use;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int w = 0; w < NW; w++) t += s_part[w];
    F->ac = t;
    F->flags = flags;
  }
}

// LDS view of one problem (pointers into the dynamic LDS block, sized by the kernel instance).  A wave handles 64/G
// problems at once, G lanes each (G = 16 for the common instance, 64 for the large-cap instance); every cross-lane
// operation below is G-wide, so problems in the same wave may diverge freely.

// One selected (src, tgt) ellipse pair as the evaluations read it: what does NOT depend on the pose, combined once
// when the pair is filed (round 4 kept the sixteen f32 values and every evaluation -- ~40 per refined problem -- widened and
// combined them again).  With C_s = m I + [d b; b -d] + a J (m = (c00+c11)/2, d = (c00-c11)/2, b = (c01+c10)/2, a = (c10-c01)/2)
// the matrix N = 2 (R C_s R^T + C_t) of a term is
//   N00 = a00 + 2p, N11 = a11 - 2p, N01 = a01 + 2q, N10 = a10 + 2q      p = d cos2t - b sin2t, q = d sin2t + b cos2t
// and only N01 + N10 = as + 4q and N01 N10 = (2q)^2 + 2q as + ap are needed.
struct cc_gpair {
  double sd, sb;    // d, b of the src covariance
  double a00, a11;  // 2 (m + t00), 2 (m + t11)
  double as, ap;    // a01 + a10 and a01 * a10 with a01 = 2 (t01 - a), a10 = 2 (t10 + a)
  double w;         // w_s * w_t
  double pad_;
  float smx, smy, tmx, tmy;
};  // 80 B: five 16-byte units (sd sb | a00 a11 | as ap | w - | means)
static_assert(sizeof(cc_gpair) == 80, "the pose-independent half of a term, in registers");
// What the refinement keeps of a pair between its evaluations: the two ellipses' f32 values as they are (56 B instead of
// the 80 B of the combined record above; cc_gmm_make_pair is repeated by every evaluation).  Round 6: with ~2 000 problems
// of ~800 pairs in flight, each streaming its records once per evaluation, the 64-lane refinement ran at the HBM's
// bandwidth (1.8 GB per chunk in 0.38 ms), not at the f64 rate -- bytes per pair are what it costs.
// Unit by unit -- pair i's k-th unit at unit index k * n + i -- so that the 64 lanes of a load or store touch ONE
// contiguous kilobyte (records side by side cost a 16-byte piece of 40 different cache lines per instruction).
struct cc_graw {
  float4 s;  // src covariance c00 c01 c10 c11
  float4 t;  // tgt covariance
  float4 m;  // src mean, tgt mean
  float2 w;  // src weight, tgt weight
};
#define CC_GRAW_BYTES 56
// the first pairs of a problem stay in LDS (CC_GMM_NL of a 64-lane problem, a quarter of that for each of the four
// problems of a 16-lane wave), the rest goes through the global pool
#ifndef CC_GMM_NL
#define CC_GMM_NL 128
#endif
struct cc_gsrc {       // where a problem's records are
  float4 *lds;         // [3][nl_cap] float4 | [nl_cap] float2 (this problem's part of the workgroup's block)
  char *glb;           // units of the pairs >= nl: [3][ng_alloc] float4 | [ng_alloc] float2, 16-byte aligned
  int nl_cap, nl, ng_alloc;
};
__device__ __forceinline__ cc_graw cc_gsrc_load(const cc_gsrc &Q, int i) {
  cc_graw r;
  if (i < Q.nl) {
    r.s = Q.lds[i];
    r.t = Q.lds[Q.nl_cap + i];
    r.m = Q.lds[2 * Q.nl_cap + i];
    r.w = ((const float2 *)(Q.lds + 3 * Q.nl_cap))[i];
  } else {
    const int j = i - Q.nl;
    const float4 *g = (const float4 *)Q.glb + j;
    r.s = g[0];
    r.t = g[Q.ng_alloc];
    r.m = g[2 * (size_t)Q.ng_alloc];
    r.w = ((const float2 *)((const float4 *)Q.glb + 3 * (size_t)Q.ng_alloc))[j];
  }
  return r;
}
__device__ __forceinline__ void cc_gsrc_store(const cc_gsrc &Q, int i, const cc_graw &r) {
  if (i < Q.nl) {
    Q.lds[i] = r.s;
    Q.lds[Q.nl_cap + i] = r.t;
    Q.lds[2 * Q.nl_cap + i] = r.m;
    ((float2 *)(Q.lds + 3 * Q.nl_cap))[i] = r.w;
  } else {
    const int j = i - Q.nl;
    float4 *g = (float4 *)Q.glb + j;
    g[0] = r.s;
    g[Q.ng_alloc] = r.t;
    g[2 * (size_t)Q.ng_alloc] = r.m;
    ((float2 *)((float4 *)Q.glb + 3 * (size_t)Q.ng_alloc))[j] = r.w;
  }
}

// sum over the lanes of a problem, result in every lane
template <int G>
__device__ __forceinline__ double cc_gsum(double v) {
  if (G >= 64) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
  }
  return cc_group_sum_d(v);
}
// the four sums of an evaluation (cost, gradient)
template <int G>
__device__ __forceinline__ void cc_gsum4(double &a, double &b, double &c, double &d) {
  a = cc_gsum<G>(a);
  b = cc_gsum<G>(b);
  c = cc_gsum<G>(c);
  d = cc_gsum<G>(d);
}

// One term of GMMPair::operator() (correlation.h:123-160) and its gradient, in closed form.  The reference builds the
// same function from ceres::Jet arithmetic; here the rotation is applied analytically:
//   R C R^T = m I + [p q; q -p] + a J     m = (c00+c11)/2, d = (c00-c11)/2, b = (c01+c10)/2, a = (c10-c01)/2,
//                                         p = d cos2t - b sin2t, q = d sin2t + b cos2t,  dp/dt = -2q, dq/dt = 2p
//   N = 2 (R C_s R^T + C_t),  mu = R m_s + t - m_t,  E = mu^T adj(N) mu,  Q = -E / (2 det N)
//   term = -w_s w_t / sqrt(det N) * exp(Q)
// Same value and derivatives as the Jet evaluation up to f64 rounding of the individual terms.
struct cc_gterm {
  double v, gx, gy, gt;
};
// Round 5: ~95 f64 instructions per term instead of ~190.  The pose-independent half comes combined (cc_gpair), the
// products are fused (the translation unit is built with -ffp-contract=off, so every fma below is written out), 1 / det
// and 1 / sqrt(det) are one reciprocal square root and its square, the exponential is the table routine of cc_fmath.h.
// Each of these moves a term by a few ulp -- the correlation by ~1e-15 -- inside the 1e-4 the contract gives the scores
// (BASELINE.json) and far below what changes a gate, a candidate order or an outcome (tests + tests/fuzz_gpu_query.py).
__device__ __forceinline__ cc_gterm cc_gmm_term(const cc_gpair &P, double px, double py, double c, double s, double c2, double s2,
                                                const double *exp_tab) {
  const double p = fma(P.sd, c2, -(P.sb * s2)), q = fma(P.sd, s2, P.sb * c2);
  const double p2 = p + p, q2 = q + q;
  const double n00 = P.a00 + p2, n11 = P.a11 - p2;
  const double nx = fma(4.0, q, P.as);
  const double det = fma(n00, n11, -fma(q2, q2 + P.as, P.ap));
  const double ddet = 4.0 * fma(q, n00 - n11, -(p * nx));
  const double smx = (double)P.smx, smy = (double)P.smy;
  const double g0 = -fma(s, smx, c * smy), g1 = fma(c, smx, -(s * smy));  // d mu / d theta
  const double m0 = g1 + (px - (double)P.tmx), m1 = (py - (double)P.tmy) - g0;
  const double m00 = m0 * m0, m11 = m1 * m1, m01 = m0 * m1;
  const double E = fma(m00, n11, fma(m11, n00, -(m01 * nx)));
  const double dE = fma(2.0, fma(m0 * g0, n11, m1 * g1 * n00), fma(4.0 * q, m00 - m11, -fma(fma(g0, m1, m0 * g1), nx, 8.0 * p * m01)));
  const double r = cc_rsqrt(det);
  const double idet = r * r;
  const double hi = -0.5 * idet;
  const double Q = hi * E;
  const double v = -(P.w * r) * cc_exp_nonpos(Q, exp_tab);
  cc_gterm o;
  o.v = v;
  o.gx = v * (hi * fma(2.0 * m0, n11, -(m1 * nx)));
  o.gy = v * (hi * fma(2.0 * m1, n00, -(m0 * nx)));
  o.gt = v * (idet * fma(-0.5, dE, -(ddet * (Q + 0.5))));
  return o;
}

// the pair pre-selection test of GMMPair's ctor (correlation.h:85-96) on dx, dy = transformed src mean - tgt mean:
//   sqrt(dx^2 + dy^2) < 3 (maj_s + maj_t)
// decided without the f64 square root whenever x = dx^2 + dy^2 is not within 1e-12 (relative) of y^2: the correctly
// rounded sqrt(x) is within 1.2e-16 of the real root and y * y, y^2 (1 +- 1e-12) carry three roundings, so outside that
// band the comparison of the squares and the reference's comparison agree; inside it the reference's expression decides.
__device__ __forceinline__ bool cc_gmm_pair_near(double dx, double dy, float smaj, float tmaj) {
  const double x = dx * dx + dy * dy;
  const double y = 3.0 * (double)(smaj + tmaj);
  const double y2 = y * y;
  if (x < y2 * (1.0 - 1e-12)) return true;
  if (x > y2 * (1.0 + 1e-12)) return false;
  return sqrt(x) < y;
}
__device__ __forceinline__ cc_gpair cc_gmm_make_pair(const cc_graw &r) {
  const double s00 = (double)r.s.x, s01 = (double)r.s.y, s10 = (double)r.s.z, s11 = (double)r.s.w;
  const double sm = 0.5 * (s00 + s11), sa = 0.5 * (s10 - s01);
  const double a01 = 2.0 * ((double)r.t.y - sa), a10 = 2.0 * ((double)r.t.z + sa);
  cc_gpair P;
  P.sd = 0.5 * (s00 - s11);
  P.sb = 0.5 * (s01 + s10);
  P.a00 = 2.0 * (sm + (double)r.t.x);
  P.a11 = 2.0 * (sm + (double)r.t.w);
  P.as = a01 + a10;
  P.ap = a01 * a10;
  P.w = (double)r.w.x * (double)r.w.y;
  P.smx = r.m.x;
  P.smy = r.m.y;
  P.tmx = r.m.z;
  P.tmy = r.m.w;
  P.pad_ = 0.0;
  return P;
}
__device__ __forceinline__ cc_graw cc_gmm_raw_of(const cc_ell &es, const cc_ell &et) {
  cc_graw r;
  r.s = make_float4(es.c00, es.c01, es.c10, es.c11);
  r.t = make_float4(et.c00, et.c01, et.c10, et.c11);
  r.m = make_float4(es.mx, es.my, et.mx, et.my);
  r.w = make_float2(es.w, et.w);
  return r;
}
__device__ __forceinline__ cc_gpair cc_gmm_make_pair(const cc_ell &es, const cc_ell &et) { return cc_gmm_make_pair(cc_gmm_raw_of(es, et)); }

// hand-off between the G lanes of a problem through memory (lockstep on the GPU: a compiler fence; the CPU test harness
// needs its threads to meet)
template <int G>
__device__ __forceinline__ void cc_gsync() {
  if (G == 64) {
    cc_wave_sync();
  } else {
    cc_group_sync();
  }
}

// ---- the selected (src, tgt) ellipse pairs of one problem -------------------------------------------------------------
// G lanes own G src ellipses of a level at a time and walk the level's tgt ellipses, whose (mean, major axis) sit in LDS
// (64 at a time, one broadcast ds_read per test).  ~1 % of the (src, tgt) grid passes GMMPair's test, so the grid is
// swept with a CONSERVATIVE f32 test first -- |d_f32| <= 3 (maj_s + maj_t) + 0.01, four f32 operations per pair; the f32
// image of the f64 expression is off by < 1e-4 for the coordinates the tidyUp gates let through, and a lane whose
// transformed mean is beyond 4096 skips the shortcut -- and only its survivors take the exact f64 test (the reference's
// expression decides, cc_gmm_pair_near).  A lane collects its hits of a 64-tgt chunk in a bit mask: no cross-lane traffic
// in the sweep; per chunk one prefix sum over the group files the (level, src, tgt) codes in an LDS list in (src, tgt)
// order, which is handed to `flush` whenever it is nearly full and at the end -- so the expensive per-pair work (term
// evaluation, pool record) runs on full lanes.  Round 3 tested every pair in f64 and balloted once per tgt: K5's largest
// part on contour-rich scans (0.66 ms of cc_k_gmm_init per 1 024 KITTI-shaped queries).
#define CC_GMM_TCHUNK 64
#ifdef CC_TUNE_GMM_CLK
__device__ unsigned long long cc_gmm_scan_clk[8];  // tuning aid: cycles of the scan's parts, summed over the 64-lane problems (fill, f32 sweep, f64 tests, filing, flush, count)
#define CC_CLK_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define CC_CLK_ADD(i, a, b) do { clk_acc[i] += (b) - (a); } while (0)
#else
#define CC_CLK_T(v)
#define CC_CLK_ADD(i, a, b)
#endif
#define CC_GMM_LIST_CAP 256
#define CC_GMM_PRE_MARGIN 0.01f
struct alignas(16) cc_gmm_scan_lds {
  float Tx[CC_GMM_TCHUNK], Ty[CC_GMM_TCHUNK], Tw[CC_GMM_TCHUNK], Tm[CC_GMM_TCHUNK];  // mean, 3 maj + margin, maj of the current tgt chunk
                                                                                     // (slots beyond the chunk: a mean no src comes near)
  unsigned code[CC_GMM_LIST_CAP];
};
static_assert(CC_GMM_ECAP_L <= 512 && CC_GMM_LEVELS <= 4, "pair codes are level:2 | src:9 | tgt:9 bits");

template <int G>
__device__ __forceinline__ unsigned long long cc_gballot(bool pred) {
  if (G == 64) return __ballot(pred);
  return (unsigned long long)cc_group_ballot(pred);
}
template <int G>
__device__ __forceinline__ int cc_gscan_incl(int v, int sl) {
  if (G == 64) {
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o);
      if (sl >= o) v += t;
    }
    return v;
  }
  return cc_group_scan_incl(v);
}
template <int G>
__device__ __forceinline__ int cc_gbcast_i(int v, int src) {
  return G == 64 ? __shfl(v, src) : cc_group_bcast(v, src);
}

// returns the number of selected pairs; flush(n, last) consumes L.code[0..n) (last: the sweep's final call)
template <int G, typename Flush>
__device__ __forceinline__ int cc_gmm_scan_pairs(const cc_gmm_feat *__restrict__ fsrc, const cc_gmm_feat *__restrict__ ftgt, double tx, double ty,
                                                 double ct0, double st0, cc_gmm_scan_lds &L, int sl, Flush flush) {
  int cnt = 0, total = 0;
#ifdef CC_TUNE_GMM_CLK
  unsigned long long clk_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  // The tgt chunks of all levels in one sequence, each chunk's (mean, major axis) values requested a chunk ahead: what a
  // chunk boundary costs is then the LDS hand-over, not a memory round trip.  A lane holds TPL tgts of the next chunk.
  constexpr int TPL = CC_GMM_TCHUNK / G;
  float nx_[TPL], ny_[TPL], nm_[TPL];
  float fsx_ = 0.f, fsy_ = 0.f, fsm_ = 0.f;  // ... and the first src block of the next chunk's level
  // the levels' ellipse counts, read once (the stores to the code pool in between keep the compiler from holding on to them)
  int ns4[CC_GMM_LEVELS], nt4[CC_GMM_LEVELS];
#pragma unroll
  for (int l = 0; l < CC_GMM_LEVELS; l++) {
    ns4[l] = fsrc->n_ell[l];
    nt4[l] = ftgt->n_ell[l];
  }
  auto cnt_of = [&](const int (&a)[CC_GMM_LEVELS], int l) {  // a[l], l in 0 .. CC_GMM_LEVELS (register array: no dynamic index)
    int v = 0;
#pragma unroll
    for (int k = 0; k < CC_GMM_LEVELS; k++) v = l == k ? a[k] : v;
    return v;
  };
  auto request = [&](int li, int t0) {
    const int ntl = cnt_of(nt4, li), nsl = cnt_of(ns4, li);  // 0 for li == CC_GMM_LEVELS
#pragma unroll
    for (int u = 0; u < TPL; u++) {
      const int j = t0 + sl + u * G;
      nx_[u] = 3.0e38f;  // beyond the chunk: dx^2 overflows to +inf, the test fails
      ny_[u] = 0.f;
      nm_[u] = 0.f;
      if (j < ntl) {
        const cc_ell *pt = &ftgt->ell[li][j];
        nx_[u] = pt->mx;
        ny_[u] = pt->my;
        nm_[u] = pt->maj;
      }
    }
    if (sl < nsl) {
      const cc_ell *ps = &fsrc->ell[li][sl];
      fsx_ = ps->mx, fsy_ = ps->my, fsm_ = ps->maj;
    }
  };
  auto next_chunk = [&](int &li, int &t0) {  // the chunk after (li, t0) that has src and tgt ellipses, or li = CC_GMM_LEVELS
    t0 += CC_GMM_TCHUNK;
    while (li < CC_GMM_LEVELS && (cnt_of(ns4, li) <= 0 || t0 >= cnt_of(nt4, li))) {
      li++;
      t0 = 0;
    }
  };
  int li = 0, t0 = -CC_GMM_TCHUNK;
  next_chunk(li, t0);
  request(li, t0);
  while (li < CC_GMM_LEVELS) {
    const int ns = cnt_of(ns4, li), ntg = cnt_of(nt4, li);
    const int tn = ntg - t0 < CC_GMM_TCHUNK ? ntg - t0 : CC_GMM_TCHUNK;
    CC_CLK_T(c_f0);
    cc_gsync<G>();  // the previous chunk is no longer read
#pragma unroll
    for (int u = 0; u < TPL; u++) {
      const int j = sl + u * G;
      L.Tx[j] = nx_[u];
      L.Ty[j] = ny_[u];
      L.Tm[j] = nm_[u];
      L.Tw[j] = 3.f * nm_[u] + CC_GMM_PRE_MARGIN;
    }
    cc_gsync<G>();
    // this chunk's first src block came with the chunk; the block after it is requested while this one is swept
    float pmx = fsx_, pmy = fsy_, pmaj = fsm_;
    int li_n = li, t0_n = t0;
    next_chunk(li_n, t0_n);
    request(li_n, t0_n);
    CC_CLK_T(c_f1);
    CC_CLK_ADD(0, c_f0, c_f1);
    for (int s0 = 0; s0 < ns; s0 += G) {
      const int si = s0 + sl;
      unsigned long long mask = 0ull;
      CC_CLK_T(c_p0);
      unsigned long long c_p1v = 0;
      const float cmx = pmx, cmy = pmy, smaj = pmaj;
      if (si + G < ns) {
        const cc_ell *ps = &fsrc->ell[li][si + G];
        pmx = ps->mx, pmy = ps->my, pmaj = ps->maj;
      }
      if (si < ns) {
        const double mx = (double)cmx, my = (double)cmy;
        const double sx = ct0 * mx + (-st0) * my + tx;  // T_init applied to the src mean, as written at correlation.h:87-90
        const double sy = st0 * mx + ct0 * my + ty;
        const float sxf = (float)sx, syf = (float)sy, s3 = 3.f * smaj;
        const bool pre_ok = fabsf(sxf) < 4096.f && fabsf(syf) < 4096.f;
        // pass 1, branch-free: the f32 test of the chunk's tgts, two per instruction (packed f32), eight per step (LDS reads
        // in flight together), the outcome pushed into a bit string by its sign (r^2 - d^2 < 0: no candidate).  A branch
        // to the f64 test inside this loop would be taken by the WAVE whenever any of its 64 lanes has a candidate among
        // the step's tgts -- nearly always -- although ~1 % of the pairs are candidates.
        unsigned long long cand = 0ull;
        const cc_f2 sx2 = {sxf, sxf}, sy2 = {syf, syf}, s32 = {s3, s3};
        for (int tb = 0; tb < tn; tb += 8) {
          unsigned m8 = 0u;
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            const cc_f2 x2 = *(const cc_f2 *)&L.Tx[tb + u], y2 = *(const cc_f2 *)&L.Ty[tb + u], w2 = *(const cc_f2 *)&L.Tw[tb + u];
            const cc_f2 d = sx2 - x2, e = sy2 - y2, r = s32 + w2;
            const cc_f2 q = cc_pk_fma(e, e, d * d);
            const cc_f2 z = cc_pk_fma(r, r, -q);  // >= 0: candidate (the margin of the conservative test covers the fused rounding)
            m8 = cc_push_sign(m8, z.x);
            m8 = cc_push_sign(m8, z.y);
          }
          // the string holds the step's eight outcomes first tgt highest, 1 = no candidate
          cand |= (unsigned long long)((cc_brev(~m8) >> 24) & 0xffu) << tb;
        }
        if (!pre_ok) cand = tn >= 64 ? ~0ull : (1ull << tn) - 1ull;
#ifdef CC_TUNE_GMM_CLK
        c_p1v = __builtin_readcyclecounter();
#endif
        // pass 2: the reference's f64 expression on the candidates (a handful per lane)
        while (cand) {
          const int tj = __ffsll((unsigned long long)cand) - 1;
          cand &= cand - 1;
          if (tj < tn && cc_gmm_pair_near(sx - (double)L.Tx[tj], sy - (double)L.Ty[tj], smaj, L.Tm[tj])) mask |= 1ull << tj;
        }
      }
      CC_CLK_T(c_p2);
#ifdef CC_TUNE_GMM_CLK
      if (c_p1v) { CC_CLK_ADD(1, c_p0, c_p1v); CC_CLK_ADD(2, c_p1v, c_p2); }
#endif
      const int c = __popcll(mask);
      const int incl = cc_gscan_incl<G>(c, sl);
      const int tot = cc_gbcast_i<G>(incl, G - 1);
      if (tot == 0) continue;
      if (cnt + tot > CC_GMM_LIST_CAP) {
        CC_CLK_T(c_q0);
        flush(cnt, false);
        CC_CLK_T(c_q1);
        CC_CLK_ADD(4, c_q0, c_q1);
        total += cnt;
        cnt = 0;
      }
      CC_CLK_T(c_p3);
      if (tot <= CC_GMM_LIST_CAP) {
        int pos = cnt + incl - c;
        while (mask) {
          const int tj = __ffsll((unsigned long long)mask) - 1;
          mask &= mask - 1;
          L.code[pos++] = (unsigned)((li << 18) | (si << 9) | (t0 + tj));
        }
        cnt += tot;
        CC_CLK_T(c_p4);
        CC_CLK_ADD(3, c_p3, c_p4);
      } else {  // one (src chunk, tgt chunk) block with more hits than the list holds: lane by lane (a lane has <= 64)
        for (int l = 0; l < G; l++) {
          const int cl = cc_gbcast_i<G>(c, l);
          if (cl == 0) continue;
          if (cnt + cl > CC_GMM_LIST_CAP) {
            flush(cnt, false);
            total += cnt;
            cnt = 0;
          }
          if (sl == l) {
            int pos = cnt;
            while (mask) {
              const int tj = __ffsll((unsigned long long)mask) - 1;
              mask &= mask - 1;
              L.code[pos++] = (unsigned)((li << 18) | (si << 9) | (t0 + tj));
            }
          }
          cnt += cl;
        }
      }
    }
    li = li_n;
    t0 = t0_n;
  }
  CC_CLK_T(c_q2);
  if (cnt > 0) flush(cnt, true);
  CC_CLK_T(c_q3);
  CC_CLK_ADD(4, c_q2, c_q3);
  CC_CLK_ADD(5, 0ull, 1ull);
#ifdef CC_TUNE_GMM_CLK
  if (G == 64 && sl == 0)
    for (int i = 0; i < 6; i++) atomicAdd(&cc_gmm_scan_clk[i], clk_acc[i]);
#endif
  return total + cnt;
}

// K5a: initial correlation of every problem (tryProblem, correlation.h:196-202).  The selected pairs' terms are evaluated
// from the compacted list, G pairs at a time.  16 lanes per problem (four problems per wave) when the chunk has more
// problems than the launch has waves; a whole wave per problem otherwise (contour-rich scans with few candidates: ~1 100
// problems of ~30 000 grid cells each per 1 024 KITTI-shaped queries -- with 16 lanes each they occupied 290 waves of a GPU
// that holds thousands, and the kernel lasted as long as the largest grid).
// The (level, src, tgt) codes of the selected pairs are kept, so that the refinement of a problem reads its pair list
// instead of sweeping the ellipse grid a second time (round 6: the second sweep was a third of cc_k_gmm_refine's time on
// contour-rich scans).  A problem's codes go to BLOCKS of the chunk's code pool -- {n, next block, n codes}: the first
// holds CC_GMM_BLK0 codes (most problems of a sparse scene need no more), the following ones CC_GMM_BLK, so a long list is
// a few long contiguous runs.  A block is taken with one atomic, issued BEFORE it is needed (the first at the problem's
// start, the next when the current one is half full): what these phases cost is memory round trips, not instructions.
// A full code pool marks the pair pool full as well: every refinement of the chunk is then skipped and the host reports
// CC_ECAPACITY (chunk_status).
#define CC_GMM_BLK0 CC_GMM_LIST_CAP
#define CC_GMM_BLK 1024
struct cc_gmm_code_pool {
  unsigned *codes;
  int cap;
  int *head;       // next free entry of codes[]
  int *pair_head;  // the refinement's pair pool head (see above)
  int pair_cap;
};
__device__ __forceinline__ cc_ell cc_gmm_ell_of(const cc_gmm_feat *f, int li, int i) {
  return f->ell[li][i];
}
template <int G>
__device__ __forceinline__ void cc_gmm_init_one(const cc_gmm_problem &pb, int pidx, const cc_gmm_feat *__restrict__ qfeat,
                                                const cc_gmm_feat *__restrict__ db_feat, cc_gmm_result *__restrict__ results, cc_gmm_scan_lds &L, int sl,
                                                const double *exp_tab, const cc_gmm_code_pool &CPL, int blk0 /*the problem's first block, or -1: from the head*/, int dyn0 /*where the blocks taken with the atomic start*/) {
  const cc_gmm_feat *fsrc = db_feat + pb.gidx;
  const cc_gmm_feat *ftgt = qfeat + pb.q;
  // the first block is requested now and bound at the first flush
  int raw_blk = sl != 0 ? 0 : (blk0 >= 0 ? blk0 : dyn0 + atomicAdd(CPL.head, CC_GMM_BLK0 + 2));  // lane 0: the requested block
  int raw_cap = CC_GMM_BLK0;
  bool have_raw = true;
  const double ct0 = cos(pb.tf[2]), st0 = sin(pb.tf[2]);
  const double c2 = ct0 * ct0 - st0 * st0, s2 = 2.0 * st0 * ct0;
  double acc = 0.0;
  int first_blk = -1, blk = -1, blk_n = 0, blk_cap = 0;
  bool pool_full = false;
  const int np = cc_gmm_scan_pairs<G>(fsrc, ftgt, pb.tf[0], pb.tf[1], ct0, st0, L, sl, [&](int n, bool last) {
    cc_gsync<G>();
    // the step's first pairs are on their way while the block bookkeeping runs
    int e = sl;
    unsigned code = 0u;
    cc_ell es = {}, et = {};
    if (e < n) {
      code = L.code[e];
      es = cc_gmm_ell_of(fsrc, (int)code >> 18, ((int)code >> 9) & 511);
      et = cc_gmm_ell_of(ftgt, (int)code >> 18, (int)code & 511);
    }
    if (blk < 0 || blk_n + n > blk_cap) {  // (uniform over the problem's lanes) bind the requested block, or take one now
      if (!have_raw) {
        raw_blk = sl == 0 ? dyn0 + atomicAdd(CPL.head, CC_GMM_BLK + 2) : 0;
        raw_cap = CC_GMM_BLK;
      }
      const int nb = cc_gbcast_i<G>(raw_blk, 0);
      have_raw = false;
      if (nb + raw_cap + 2 > CPL.cap) pool_full = true;
      if (!pool_full) {
        if (blk >= 0) {
          if (sl == 0) {
            CPL.codes[blk] = (unsigned)blk_n;
            CPL.codes[blk + 1] = (unsigned)nb;
          }
        } else {
          first_blk = nb;
        }
      }
      blk = nb;
      blk_n = 0;
      blk_cap = raw_cap;
    }
    // (the codes go out in a loop of their own: a loop that mixes loads and stores waits for ALL of them at every use of a
    // loaded value -- one counter serves both on gfx9 and the two kinds complete out of order)
    if (!pool_full) {
      unsigned *dst = CPL.codes + blk + 2 + blk_n;
      for (int k = sl; k < n; k += G) dst[k] = L.code[k];
    }
    while (e < n) {
      const int e2 = e + G;
      unsigned code2 = 0u;
      cc_ell es2 = es, et2 = et;
      if (e2 < n) {
        code2 = L.code[e2];
        es2 = cc_gmm_ell_of(fsrc, (int)code2 >> 18, ((int)code2 >> 9) & 511);
        et2 = cc_gmm_ell_of(ftgt, (int)code2 >> 18, (int)code2 & 511);
      }
      const cc_gpair P = cc_gmm_make_pair(es, et);
      acc += cc_gmm_term(P, pb.tf[0], pb.tf[1], ct0, st0, c2, s2, exp_tab).v;
      e = e2;
      code = code2;
      es = es2;
      et = et2;
    }
    blk_n += n;
    if (!last && !have_raw && 2 * blk_n >= blk_cap) {  // the next block, ahead of its use
      raw_blk = sl == 0 ? dyn0 + atomicAdd(CPL.head, CC_GMM_BLK + 2) : 0;
      raw_cap = CC_GMM_BLK;
      have_raw = true;
    }
    cc_gsync<G>();
  });
  const double cost = cc_gsum<G>(acc);
  if (sl == 0) {
    if (pool_full) {
      atomicMax(CPL.pair_head, CPL.pair_cap + 1);
    } else if (blk >= 0) {
      CPL.codes[blk] = (unsigned)blk_n;
      CPL.codes[blk + 1] = 0xffffffffu;
    }
    cc_gmm_result R;
    R.corr_init = -cost / sqrt(fsrc->ac * ftgt->ac);
    R.corr_opt = R.corr_init;
    R.tf_opt[0] = pb.tf[0];
    R.tf_opt[1] = pb.tf[1];
    R.tf_opt[2] = pb.tf[2];
    R.optimized = 0;
    R.iterations = 0;
    R.termination = 0;
    R.flags = (fsrc->flags | ftgt->flags) & 5;
    R.n_pairs = np;
    R.code_seg = pool_full ? -1 : first_blk;
    results[pidx] = R;
  }
}
// grid = any (grid-stride over the device-side problem count), block = 64
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
cc_k_gmm_init(const cc_gmm_problem *__restrict__ probs, const int *__restrict__ prob_list, const int *__restrict__ n_prob_p,
              const cc_gmm_feat *__restrict__ qfeat, const cc_gmm_feat *__restrict__ db_feat, cc_gmm_result *__restrict__ results,
              cc_gmm_code_pool CPL) {
  __shared__ cc_gmm_scan_lds lds[64 / CC_G];
  __shared__ double exp_tab[64];
  exp_tab[threadIdx.x] = __longlong_as_double((long long)cc_exp2_tab64[threadIdx.x]);
  cc_wave_sync();
  const int n_prob = *n_prob_p;
  // The FIRST code block of the i-th problem of the chunk's list is block i of the pool -- no atomic (a chunk of a sparse
  // scene has ~30 000 problems; one atomic each on the pool's head, all waves asking at once, cost 36-54 us per chunk: a
  // third of this kernel).  Half of the pool is set aside for such blocks; problems beyond that many (a 50 000-scan DB
  // brings ~160 000 per chunk) take their first block from the head like every further block, and the head counts from
  // behind the static ones.
  const int n_static = n_prob < CPL.cap / 2 / (CC_GMM_BLK0 + 2) ? n_prob : CPL.cap / 2 / (CC_GMM_BLK0 + 2);
  const int dyn0 = n_static * (CC_GMM_BLK0 + 2);
  if (n_prob <= (int)gridDim.x) {  // a wave per problem (uniform over the launch)
    for (int pi = blockIdx.x; pi < n_prob; pi += gridDim.x) {
      const int pidx = prob_list[pi];
      const cc_gmm_problem pb = probs[pidx];
      cc_gmm_init_one<64>(pb, pidx, qfeat, db_feat, results, lds[0], (int)threadIdx.x, exp_tab, CPL, pi < n_static ? pi * (CC_GMM_BLK0 + 2) : -1, dyn0);
    }
    return;
  }
  const int sub = threadIdx.x / CC_G, sl = threadIdx.x % CC_G;
  // (the list entry two problems ahead and the record one ahead, held in registers across a problem: measured, 73 -> 80 us per
  // chunk of a sparse scene -- the kernel is at its register limit and spills what it is asked to keep)
  for (int pi = blockIdx.x * (64 / CC_G) + sub; pi < n_prob; pi += gridDim.x * (64 / CC_G)) {
    const int pidx = prob_list[pi];
    const cc_gmm_problem pb = probs[pidx];
    cc_gmm_init_one<CC_G>(pb, pidx, qfeat, db_feat, results, lds[sub], sl, exp_tab, CPL, pi < n_static ? pi * (CC_GMM_BLK0 + 2) : -1, dyn0);
  }
}

// cost and gradient at p over a problem's pair list, summed over its G lanes
template <int G>
__device__ __forceinline__ void cc_gmm_eval(const cc_gsrc &Q, int np, int sl, const double p[3], double *cost, double grad[3],
                                            const double *exp_tab) {
  double c, s;
  sincos(p[2], &s, &c);
  const double c2 = c * c - s * s, s2 = 2.0 * s * c;
  double a = 0.0, ax = 0.0, ay = 0.0, at = 0.0;
  auto add_term = [&](const cc_graw &w) {
    const cc_gpair P = cc_gmm_make_pair(w);
    const cc_gterm t = cc_gmm_term(P, p[0], p[1], c, s, c2, s2, exp_tab);
#ifdef CC_TUNE_GMM_TWICE  // tuning aid: the pair arithmetic twice (what it costs = this build's K5 minus the product's)
    {
      const cc_gterm t2 = cc_gmm_term(P, p[0] + 1e-9, p[1], c * 1.000000001, s, c2 * 1.000000001, s2, exp_tab);  // nothing in common with the first but the loads
      a += 1e-300 * (t2.v + t2.gx + t2.gy + t2.gt);
    }
#endif
    a += t.v;
    ax += t.gx;
    ay += t.gy;
    at += t.gt;
  };
  // A lane's pairs in ascending position: those in LDS, then those in the pool -- two loops (one loop that picks the
  // source per step carried a branch, sixteen register moves and seven lane reads per pair)
  for (int i = sl; i < Q.nl; i += G) {
    cc_graw w;
    w.s = Q.lds[i];
    w.t = Q.lds[Q.nl_cap + i];
    w.m = Q.lds[2 * Q.nl_cap + i];
    w.w = ((const float2 *)(Q.lds + 3 * Q.nl_cap))[i];
    add_term(w);
  }
  // (Q.nl is np or a multiple of G: the lane's first pool position is its lane index)
  const int ng = np - Q.nl;
  const float4 *gb = (const float4 *)Q.glb;
  const float2 *gw = (const float2 *)(gb + 3 * (size_t)Q.ng_alloc);
  auto load = [&](int j) {  // (scalar bases + a 32-bit lane offset: the kernel has no scalar registers left, they came back through six lane reads per step)
    cc_graw r;
    r.s = gb[j];
    r.t = gb[Q.ng_alloc + j];
    r.m = gb[2 * (size_t)Q.ng_alloc + j];
    r.w = gw[j];
    return r;
  };
  // the next pair's record is requested before this pair's ~150 f64 instructions are issued
  cc_graw w = {};
  if (sl < ng) w = load(sl);
  for (int j = sl; j < ng; j += G) {
    cc_graw n = w;
#ifdef CC_TUNE_GMM_NOLOAD
    if (j + G < ng && p[0] == 1.2345e-300) n = load(j + G);
#else
    if (j + G < ng) n = load(j + G);
#endif
    add_term(w);
    w = n;
  }
  cc_gsum4<G>(a, ax, ay, at);
  *cost = a;
  grad[0] = ax;
  grad[1] = ay;
  grad[2] = at;
}

// The FIRST evaluation of a refined problem (at the initial pose) also files the problem's pair records: the codes
// cc_k_gmm_init left are read block by block, the two ellipses of a pair gathered, the record stored for the evaluations
// that follow and the term taken from it right away.  Codes are requested a sub-batch ahead, the next block's header a
// block ahead.  A lane takes the positions it takes in cc_gmm_eval (position = lane, mod G): the sums are those of a
// plain evaluation over the filed records, bit for bit.
template <int G>
__device__ __forceinline__ void cc_gmm_eval_first(const unsigned *__restrict__ codes, int blk, const cc_gmm_feat *__restrict__ fsrc,
                                                  const cc_gmm_feat *__restrict__ ftgt, const cc_gsrc &Q, int sl, const double p[3],
                                                  double *cost, double grad[3], const double *exp_tab) {
  double c, s;
  sincos(p[2], &s, &c);
  const double c2 = c * c - s * s, s2 = 2.0 * s * c;
  double a = 0.0, ax = 0.0, ay = 0.0, at = 0.0;
  int done = 0, n = 0, nxt = -1;
  if (blk >= 0) {
    n = (int)codes[blk];
    nxt = (int)codes[blk + 1];
  }
  while (blk >= 0) {
    int n2 = 0, nxt2 = -1;
    if (nxt >= 0) {
      n2 = (int)codes[nxt];
      nxt2 = (int)codes[nxt + 1];
    }
    const unsigned *cb = codes + blk + 2;
    int e0 = sl - done % G;
    e0 = e0 < 0 ? e0 + G : e0;
    // U steps at a time: their codes were requested a sub-batch ago, their 2 U ellipses are requested together, then the
    // U records are stored and evaluated.  (Loads and stores share one counter and complete out of order: a step-by-step
    // pipeline waits for its own stores at every step -- measured: 4 700 cycles per step instead of ~1 100.)
    constexpr int U = 2;
    unsigned cc_[U];
#pragma unroll
    for (int u = 0; u < U; u++) cc_[u] = e0 + u * G < n ? cb[e0 + u * G] : 0u;
    for (; e0 < n; e0 += U * G) {
      cc_ell es[U], et[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        es[u] = cc_ell{};
        et[u] = cc_ell{};
        if (e0 + u * G < n) {
          es[u] = cc_gmm_ell_of(fsrc, (int)cc_[u] >> 18, ((int)cc_[u] >> 9) & 511);
          et[u] = cc_gmm_ell_of(ftgt, (int)cc_[u] >> 18, (int)cc_[u] & 511);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int en = e0 + (U + u) * G;
        cc_[u] = en < n ? cb[en] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (e0 + u * G < n) {
          const cc_graw r = cc_gmm_raw_of(es[u], et[u]);
          cc_gsrc_store(Q, done + e0 + u * G, r);
          const cc_gpair P = cc_gmm_make_pair(r);
          const cc_gterm t = cc_gmm_term(P, p[0], p[1], c, s, c2, s2, exp_tab);
          a += t.v;
          ax += t.gx;
          ay += t.gy;
          at += t.gt;
        }
      }
    }
    done += n;
    blk = nxt;
    n = n2;
    nxt = nxt2;
  }
  cc_gsum4<G>(a, ax, ay, at);
  *cost = a;
  grad[0] = ax;
  grad[1] = ay;
  grad[2] = at;
}

struct cc_gmm_ctx {  // what a line-search evaluation needs
  cc_gsrc Q;
  const double *exp_tab;  // cc_exp2_tab64 in LDS
  int np, sl;
#ifdef CC_TUNE_GMM_CLK
  unsigned long long *ev_clk;  // tuning aid: cycles spent in evaluations, their count
  int *n_ev;
#endif
};
#ifdef CC_TUNE_GMM_CLK  // tuning aid: per refined problem {np | G << 20 | iterations << 40 | evaluations << 48, cycles, cycles in evaluations, cycles filing pairs}
#define CC_GMM_CLK_CAP 65536
__device__ unsigned long long cc_gmm_clk[CC_GMM_CLK_CAP * 4];
__device__ unsigned long long cc_gmm_clk2[CC_GMM_CLK_CAP * 4];  // wall clock (100 MHz) at start / end, HW_ID, -
__device__ int cc_gmm_clk_n;
#endif
// ---- Ceres 2.x line search pieces (see oracle/orc_gmm.h for the provenance notes) ----
struct cc_fs {  // FunctionSample; vector_x is not kept (it is pos + x * dir, recomputed where needed)
  double x, value, gradient;
  double vg[3];
  bool value_ok, grad_ok;
};
__device__ __forceinline__ cc_fs cc_fs_sel(bool c, const cc_fs &a, const cc_fs &b) {  // c ? a : b, field by field (keeps both in registers)
  cc_fs r;
  r.x = c ? a.x : b.x;
  r.value = c ? a.value : b.value;
  r.gradient = c ? a.gradient : b.gradient;
  r.vg[0] = c ? a.vg[0] : b.vg[0];
  r.vg[1] = c ? a.vg[1] : b.vg[1];
  r.vg[2] = c ? a.vg[2] : b.vg[2];
  r.value_ok = c ? a.value_ok : b.value_ok;
  r.grad_ok = c ? a.grad_ok : b.grad_ok;
  return r;
}

__device__ __forceinline__ double cc_ipow(double x, int k) {  // the Vandermonde entries: k in 0..3
  return k == 0 ? 1.0 : (k == 1 ? x : (k == 2 ? x * x : x * x * x));
}

// FullPivLU solve of the n x n Vandermonde system (what Eigen's fullPivLu().solve() does in
// FindInterpolatingPolynomial), n <= 4.  Every array index is a compile-time constant after unrolling and the
// permutations are conditional register swaps, so nothing spills to scratch.
__device__ __forceinline__ void cc_solve_fullpiv(double (&A)[4][4], double (&b)[4], int n, double (&x)[4]) {
  int cp[4] = {0, 1, 2, 3};
  bool done = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int pr = k, pc = k;
    double best = -1;
#pragma unroll
    for (int i = k; i < 4; i++) {
#pragma unroll
      for (int j = k; j < 4; j++) {
        const double v = fabs(A[i][j]);
        if (i < n && j < n && v > best) {
          best = v;
          pr = i;
          pc = j;
        }
      }
    }
    if (k >= n || best == 0.0) done = true;
    if (!done) {
#pragma unroll
      for (int r = k + 1; r < 4; r++) {
        if (pr == r) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const double t = A[k][j];
            A[k][j] = A[r][j];
            A[r][j] = t;
          }
          const double t = b[k];
          b[k] = b[r];
          b[r] = t;
        }
      }
#pragma unroll
      for (int c = k + 1; c < 4; c++) {
        if (pc == c) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double t = A[i][k];
            A[i][k] = A[i][c];
            A[i][c] = t;
          }
          const int t = cp[k];
          cp[k] = cp[c];
          cp[c] = t;
        }
      }
#pragma unroll
      for (int i = k + 1; i < 4; i++) {
        if (i < n) {
          const double f = A[i][k] / A[k][k];
          A[i][k] = 0;
#pragma unroll
          for (int j = k + 1; j < 4; j++) A[i][j] -= f * A[k][j];
          b[i] -= f * b[k];
        }
      }
    }
  }
  double yv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    if (i < n) {
      double sacc = b[i];
#pragma unroll
      for (int j = i + 1; j < 4; j++)
        if (j < n) sacc -= A[i][j] * yv[j];
      yv[i] = (A[i][i] != 0.0) ? sacc / A[i][i] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (i < n && cp[i] == c) x[c] = yv[i];
}

// polynomial with coefficients p[4 - np .. 3] (highest power first), Horner
__device__ __forceinline__ double cc_polyval(const double (&p)[4], int np, double x) {
  double v = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i >= 4 - np) v = v * x + p[i];
  return v;
}

// InterpolatingPolynomialMinimizingStepSize for CUBIC with samples {lowerbound, current}
__device__ double cc_interp_step(const cc_fs &lo, const cc_fs &cur, double x_min, double x_max) {
  if (!cur.value_ok) {
    double s = cur.x * 0.5;
    s = s < x_min ? x_min : s;
    return s < x_max ? s : x_max;
  }
  const int nc = (lo.value_ok ? 1 : 0) + (lo.grad_ok ? 1 : 0) + (cur.value_ok ? 1 : 0) + (cur.grad_ok ? 1 : 0);
  const int degree = nc - 1;
  // rows in the reference's order (sample 0 value, sample 0 gradient, sample 1 value, sample 1 gradient), absent rows
  // skipped; built as 4 candidates that are compacted with conditional moves
  double A[4][4], b[4], poly[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    b[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) A[i][j] = 0;
  }
  int row = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const bool is_grad = (c & 1) != 0;
    const bool present = c == 0 ? lo.value_ok : (c == 1 ? lo.grad_ok : (c == 2 ? cur.value_ok : cur.grad_ok));
    const double sx = c < 2 ? lo.x : cur.x;
    double rv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!is_grad)
        rv[j] = j <= degree ? cc_ipow(sx, degree - j) : 0.0;
      else
        rv[j] = j < degree ? (double)(degree - j) * cc_ipow(sx, degree - j - 1) : 0.0;
    }
    const double rb = c == 0 ? lo.value : (c == 1 ? lo.gradient : (c == 2 ? cur.value : cur.gradient));
    if (present) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (row == r) {
#pragma unroll
          for (int j = 0; j < 4; j++) A[r][j] = rv[j];
          b[r] = rb;
        }
      }
      row++;
    }
  }
  double sol[4] = {0, 0, 0, 0};
  if (nc == 4 && cur.x != lo.x) {
    // Both samples carry value and gradient (the case of every regular line-search step): the interpolating cubic in
    // closed form -- Hermite coefficients about lo.x, expanded to the monomial basis the rest of this function works in --
    // instead of the 4x4 full-pivot LU solve of the Vandermonde system; the same polynomial up to rounding.
    const double x0 = lo.x, h = cur.x - lo.x, ih = 1.0 / h;
    const double df = (cur.value - lo.value) * ih;
    const double Ac = (lo.gradient + cur.gradient - 2.0 * df) * ih * ih;
    const double Bc = (3.0 * df - 2.0 * lo.gradient - cur.gradient) * ih;
    sol[0] = Ac;
    sol[1] = Bc - 3.0 * Ac * x0;
    sol[2] = lo.gradient - 2.0 * Bc * x0 + 3.0 * Ac * x0 * x0;
    sol[3] = lo.value - lo.gradient * x0 + Bc * x0 * x0 - Ac * x0 * x0 * x0;
  } else {
    cc_solve_fullpiv(A, b, nc, sol);
  }
  // poly[] right-aligned: coefficient of x^(np-1-i) at index 4 - np + i
  const int np = nc;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    poly[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (i - (4 - np) == j) poly[i] = sol[j];
  }
  double opt_x = (x_min + x_max) / 2.0;
  double opt_v = cc_polyval(poly, np, opt_x);
  const double vmin = cc_polyval(poly, np, x_min);
  if (vmin < opt_v) {
    opt_v = vmin;
    opt_x = x_min;
  }
  const double vmax = cc_polyval(poly, np, x_max);
  if (vmax < opt_v) {
    opt_v = vmax;
    opt_x = x_max;
  }
  if (np > 2) {
    // derivative coefficients, right-aligned in d[0..2] (d[2] = constant term); leading zeros are skipped
    double d[3];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = (double)(3 - i) * poly[i];  // exponent of poly[i] is 3 - i
    // `while (lead + 1 < deg && d[lead] == 0) lead++`: quadratic, linear or constant derivative (for np == 3, d[0] is 0)
    double a = 0, bb = 0, c = 0;
    int dd = 0;
    if (d[0] != 0.0) {
      dd = 2;
      a = d[0];
      bb = d[1];
      c = d[2];
    } else if (d[1] != 0.0) {
      dd = 1;
      a = d[1];
      bb = d[2];
    }
    double roots[2] = {0, 0};
    int nr = 0;
    if (dd == 1) {
      roots[0] = -bb / a;
      nr = 1;
    } else if (dd == 2) {
      const double D = bb * bb - 4 * a * c;
      const double sq = sqrt(fabs(D));
      nr = 2;
      if (D >= 0) {
        if (bb >= 0) {
          roots[0] = (-bb - sq) / (2.0 * a);
          roots[1] = (2.0 * c) / (-bb - sq);
        } else {
          roots[0] = (2.0 * c) / (-bb + sq);
          roots[1] = (-bb + sq) / (2.0 * a);
        }
      } else {
        roots[0] = -bb / (2.0 * a);
        roots[1] = -bb / (2.0 * a);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const double r = roots[i];
      if (i < nr && !(r < x_min || r > x_max)) {
        const double v = cc_polyval(poly, np, r);
        if (v < opt_v) {
          opt_v = v;
          opt_x = r;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const double sx = i == 0 ? lo.x : cur.x;
    if (!(sx < x_min || sx > x_max)) {
      const double v = cc_polyval(poly, np, sx);
      if (v < opt_v) {
        opt_x = sx;
        opt_v = v;
      }
    }
  }
  return opt_x;
}


template <int G>
__device__ __forceinline__ void cc_ls_eval(const cc_gmm_ctx &S, const double pos[3], const double dir[3], double x, cc_fs *o) {
  o->x = x;
  double vx[3];
  for (int i = 0; i < 3; i++) vx[i] = pos[i] + x * dir[i];
#ifdef CC_TUNE_GMM_CLK
  const unsigned long long t_ev = __builtin_readcyclecounter();
#endif
  cc_gmm_eval<G>(S.Q, S.np, S.sl, vx, &o->value, o->vg, S.exp_tab);
#ifdef CC_TUNE_GMM_CLK
  *S.ev_clk += __builtin_readcyclecounter() - t_ev;
  *S.n_ev += 1;
#endif
  o->value_ok = isfinite(o->value);
  o->grad_ok = o->value_ok && isfinite(o->vg[0]) && isfinite(o->vg[1]) && isfinite(o->vg[2]);
  o->gradient = dir[0] * o->vg[0] + dir[1] * o->vg[1] + dir[2] * o->vg[2];
}

// WolfeLineSearch::DoSearch (bracketing + zoom).  Returns success; *opt is the accepted sample.
template <int G>
__device__ bool cc_wolfe(const cc_gmm_ctx &S, const double pos[3], const double dir[3], double step0, double cost0,
                         double dgrad0, const double g0[3], cc_fs *opt) {
  const double min_step_size = 1e-9, suff_dec = 1e-4, suff_curv = 0.9, max_expand = 10.0;
  const int max_it = 20;
  const double dnorm = fmax(fabs(dir[0]), fmax(fabs(dir[1]), fabs(dir[2])));
  cc_fs init;
  init.x = 0;
  init.value = cost0;
  init.gradient = dgrad0;
  init.value_ok = init.grad_ok = true;
  for (int i = 0; i < 3; i++) init.vg[i] = g0[i];
  int nit = 0;
  cc_fs prev = init, cur, lo = init, hi = init;
  bool zoom = false;
  cc_ls_eval<G>(S, pos, dir, step0, &cur);
  while (true) {
    ++nit;
    if (cur.value_ok && (cur.value > (init.value + suff_dec * init.gradient * cur.x) || (prev.value_ok && cur.value > prev.value))) {
      zoom = true;
      lo = prev;
      hi = cur;
      break;
    }
    if (cur.value_ok && fabs(cur.gradient) <= -suff_curv * init.gradient) {
      lo = cur;
      hi = cur;
      break;
    } else if (cur.value_ok && cur.gradient >= 0) {
      zoom = true;
      lo = cur;
      hi = prev;
      break;
    } else if (nit >= max_it) {
      if (cur.value_ok && cur.value < lo.value) lo = cur;
      break;
    }
    const double mn = cur.value_ok ? cur.x : prev.x;
    const double mx = cur.value_ok ? (cur.x * max_expand) : cur.x;
    const double step = cc_interp_step(prev, cur, mn, mx);
    if (step * dnorm < min_step_size) return false;
    if (cur.value_ok) prev = cur;
    cc_ls_eval<G>(S, pos, dir, step, &cur);
  }
  if (zoom && fabs(hi.x - lo.x) * dnorm < min_step_size) zoom = false;
  if (!zoom) {
    *opt = lo;
    return true;
  }
  // zoom phase
  cc_fs sol;
  sol.value_ok = false;
  bool zoom_ok = true;
  cc_fs blo = lo, bhi = hi;
  if (blo.gradient * (bhi.x - blo.x) >= 0) {
    zoom_ok = false;
  } else {
    while (true) {
      sol = blo;
      if (nit >= max_it) {
        zoom_ok = false;
        break;
      }
      if (fabs(bhi.x - blo.x) * dnorm < min_step_size) {
        zoom_ok = false;
        break;
      }
      ++nit;
      const bool lo_first = blo.x < bhi.x;
      const cc_fs lb = cc_fs_sel(lo_first, blo, bhi);
      const cc_fs ub = cc_fs_sel(lo_first, bhi, blo);
      const double step = cc_interp_step(lb, ub, lb.x, ub.x);
      cc_ls_eval<G>(S, pos, dir, step, &sol);
      if (!sol.value_ok || !sol.grad_ok) {
        zoom_ok = false;
        break;
      }
      if ((sol.value > (init.value + suff_dec * init.gradient * sol.x)) || (sol.value >= blo.value)) {
        bhi = sol;
        continue;
      }
      if (fabs(sol.gradient) <= -suff_curv * init.gradient) break;
      if (sol.gradient * (bhi.x - blo.x) >= 0) bhi = blo;
      blo = sol;
    }
  }
  if (!zoom_ok && !sol.value_ok) return false;
  if (!sol.value_ok || sol.value > lo.value)
    *opt = lo;
  else
    *opt = sol;
  return true;
}


// K5b: calcCorrelation (correlation.h:206-238) for the problems cc_k_select listed: LineSearchMinimizer, LBFGS rank 20,
// Wolfe / cubic interpolation, <= 10 iterations.  G lanes per problem.
// grid = any (grid-stride over the device-side list), block = 64
// G = 16: four problems per wave; 64: one wave per problem.
template <int G>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CC_GMM_WPE)))
cc_k_gmm_refine(const cc_gmm_problem *__restrict__ probs, const int *__restrict__ n_sel_p, const int *__restrict__ sel_list,
                const int *__restrict__ n_mid_p, const int *__restrict__ mid_list, const int *__restrict__ n_other_p,
                const cc_gmm_feat *__restrict__ qfeat, const cc_gmm_feat *__restrict__ db_feat, float corr_lb,
                char *__restrict__ pool /*[pool_cap] records of CC_GRAW_BYTES*/, int pool_cap, const int *__restrict__ pool_off /*[problem slot], cc_k_select*/, cc_gmm_result *__restrict__ results,
                const unsigned *__restrict__ codes, const int *__restrict__ cls_list /*the own list by length class (64-lane instance) or nullptr*/,
                const int *__restrict__ cls_cnt, int sel_stride) {
  constexpr int NP = 64 / G;                // problems per workgroup
  constexpr int NLCAP = CC_GMM_NL / NP;     // records of a problem that stay in LDS
  static_assert(NLCAP % 2 == 0, "a problem's LDS block is 3 NLCAP float4 + NLCAP float2");
  __shared__ double hist_all[NP][80];  // L-BFGS history: dx[10][3] | dg[10][3] | dx.dg[10] | alpha[10]  (group-uniform values)
  __shared__ float4 rec_lds[NP][NLCAP > 0 ? NLCAP * 7 / 2 : 1];
  __shared__ double exp_tab[64];
  exp_tab[threadIdx.x] = __longlong_as_double((long long)cc_exp2_tab64[threadIdx.x]);
  cc_wave_sync();
  const int sub = threadIdx.x / G, sl = threadIdx.x % G;
  double *hist = hist_all[sub];
  // this instance's own list, then the in-between problems if the chunk's problem count sends them here
  const int n_own = *n_sel_p, n_mid = *n_mid_p;
  const bool mid_here = ((n_own + n_mid + *n_other_p) >= CC_GMM_PACK_MIN_PROBLEMS) == (G == 16);
  const int n_sel = n_own + (mid_here ? n_mid : 0);
  for (int k = blockIdx.x * NP + sub; k < n_sel; k += gridDim.x * NP) {
    int pidx;
    if (k >= n_own) {
      pidx = mid_list[k - n_own];
    } else if (cls_list == nullptr) {
      pidx = sel_list[k];
    } else {  // the k-th problem in class order (the class counts add up to n_own: cc_k_select has finished)
      int c = 0, kk = k;
#pragma unroll
      for (int j = 0; j < CC_GMM_NCLS - 1; j++) {
        const int cj = cls_cnt[j];
        if (c == j && kk >= cj) {
          kk -= cj;
          c = j + 1;
        }
      }
      pidx = cls_list[(size_t)c * sel_stride + kk];
    }
    cc_gmm_result R = results[pidx];
    if ((float)R.corr_init < corr_lb) continue;
    const cc_gmm_problem pb = probs[pidx];
    const cc_gmm_feat *fsrc = db_feat + pb.gidx;
    const cc_gmm_feat *ftgt = qfeat + pb.q;
    const int np = G >= 64 ? cc_uniform_i(R.n_pairs) : R.n_pairs;  // one problem per wave: counts and strides are scalars
    // The ellipse tables of the two scans were swept by cc_k_gmm_init a few thousand problems ago and have left the L2 since
    // (a chunk's problems touch ~250 MB of them): their cache lines (four ellipses each) are requested now, all at once,
    // so that the gathers of the first evaluation find them in the L2 instead of paying an HBM round trip per step.
#pragma unroll
    for (int li = 0; li < CC_GMM_LEVELS; li++) {
      for (int i = sl * 4; i < fsrc->n_ell[li]; i += G * 4) cc_touch_global(&fsrc->ell[li][i]);
      for (int i = sl * 4; i < ftgt->n_ell[li]; i += G * 4) cc_touch_global(&ftgt->ell[li][i]);
    }
    // ---- where the problem's records go: the first NLCAP in LDS, the rest in the pool (an even count: 16-byte units)
    cc_gmm_ctx S;
    S.Q.lds = rec_lds[sub];
    S.Q.nl_cap = NLCAP;
    S.Q.nl = np < NLCAP ? np : NLCAP;
    S.Q.ng_alloc = (np - S.Q.nl + 1) & ~1;
    int off = S.Q.ng_alloc > 0 ? pool_off[pidx] : 0;
    if (G == 64) off = cc_uniform_i(off);
    if (off + S.Q.ng_alloc > pool_cap) {
      if (sl == 0) results[pidx].flags = R.flags | 2;
      continue;
    }
    S.Q.glb = pool + (size_t)off * CC_GRAW_BYTES;
#ifdef CC_TUNE_GMM_CLK
    const unsigned long long t_p0 = __builtin_readcyclecounter();
    const unsigned long long t_w0 = wall_clock64();
    unsigned long long ev_clk = 0;
    int n_ev = 0;
#endif
    S.exp_tab = exp_tab;
    S.np = np;
    S.sl = sl;
#ifdef CC_TUNE_GMM_CLK
    S.ev_clk = &ev_clk;
    S.n_ev = &n_ev;
#endif
    cc_gsync<G>();  // the previous problem's records in LDS are no longer read
    double x[3] = {pb.tf[0], pb.tf[1], pb.tf[2]};
    double cost, g[3];
    // the evaluation at the initial pose files the pair records (from the code list cc_k_gmm_init left) on its way
    cc_gmm_eval_first<G>(codes, R.code_seg, fsrc, ftgt, S.Q, sl, x, &cost, g, exp_tab);
    __threadfence_block();  // the records are read back by other lanes of the problem
    cc_gsync<G>();
#ifdef CC_TUNE_GMM_CLK
    const unsigned long long t_p1 = __builtin_readcyclecounter();
#endif
    const double denom = sqrt(fsrc->ac * ftgt->ac);
    {
    // ---- calcCorrelation (correlation.h:206-238): LineSearchMinimizer, LBFGS rank 20, Wolfe/cubic, <= 10 iterations
    R.optimized = 1;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double cur_cost = cost, cur_g[3] = {g[0], g[1], g[2]};
    double prev_cost = 0, prev_g[3] = {0, 0, 0}, prev_dir[3] = {0, 0, 0}, prev_step = 0;
    double *dxh = hist, *dgh = hist + 30, *dxdg = hist + 60, *alpha = hist + 70;  // [k * 3 + i]
    int ncorr = 0;
    int restarts = 0;
    int term = 0, iter = 0;
    double gmax = fmax(fabs(cur_g[0]), fmax(fabs(cur_g[1]), fabs(cur_g[2])));
    double final_cost = cur_cost;
    if (gmax <= gradient_tolerance) {
      term = 1;
    } else {
      while (true) {
        if (iter >= 10) {
          term = 0;
          break;
        }
        iter++;
        double dir[3];
        bool ls_status = true;
        if (iter == 1) {
          for (int i = 0; i < 3; i++) dir[i] = -cur_g[i];
        } else {
          double ddx[3], ddg[3];
          for (int i = 0; i < 3; i++) {
            ddx[i] = prev_dir[i] * prev_step;
            ddg[i] = cur_g[i] - prev_g[i];
          }
          const double dd = ddx[0] * ddg[0] + ddx[1] * ddg[1] + ddx[2] * ddg[2];
          if (dd > 1e-14 && ncorr < 10) {
            for (int i = 0; i < 3; i++) {
              dxh[ncorr * 3 + i] = ddx[i];
              dgh[ncorr * 3 + i] = ddg[i];
            }
            dxdg[ncorr] = dd;
            ncorr++;
          }
          double sd[3] = {cur_g[0], cur_g[1], cur_g[2]};
          for (int k = ncorr - 1; k >= 0; k--) {
            const double al = (dxh[k * 3 + 0] * sd[0] + dxh[k * 3 + 1] * sd[1] + dxh[k * 3 + 2] * sd[2]) / dxdg[k];
            for (int i = 0; i < 3; i++) sd[i] -= al * dgh[k * 3 + i];
            alpha[k] = al;
          }
          for (int k = 0; k < ncorr; k++) {
            const double beta = (dgh[k * 3 + 0] * sd[0] + dgh[k * 3 + 1] * sd[1] + dgh[k * 3 + 2] * sd[2]) / dxdg[k];
            for (int i = 0; i < 3; i++) sd[i] += dxh[k * 3 + i] * (alpha[k] - beta);
          }
          for (int i = 0; i < 3; i++) dir[i] = -1.0 * sd[i];
          if (dir[0] * cur_g[0] + dir[1] * cur_g[1] + dir[2] * cur_g[2] >= 0.0) ls_status = false;
        }
        if (!ls_status && restarts >= 5) {
          term = -1;
          break;
        } else if (!ls_status) {
          restarts++;
          ncorr = 0;
          for (int i = 0; i < 3; i++) dir[i] = -cur_g[i];
        }
        const double dderiv = cur_g[0] * dir[0] + cur_g[1] * dir[1] + cur_g[2] * dir[2];
        const double step0 = (iter == 1 || !ls_status) ? fmin(1.0, 1.0 / gmax) : fmin(1.0, 2.0 * (cur_cost - prev_cost) / dderiv);
        if (step0 < 0.0) {
          term = -1;
          break;
        }
        cc_fs opt;
        if (!cc_wolfe<G>(S, x, dir, step0, cur_cost, dderiv, cur_g, &opt)) {
          term = -1;
          break;
        }
        prev_cost = cur_cost;
        for (int i = 0; i < 3; i++) {
          prev_g[i] = cur_g[i];
          prev_dir[i] = dir[i];
        }
        prev_step = opt.x;
        cur_cost = opt.value;
        for (int i = 0; i < 3; i++) cur_g[i] = opt.vg[i];
        gmax = fmax(fabs(cur_g[0]), fmax(fabs(cur_g[1]), fabs(cur_g[2])));
        const double xnorm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        double sn = 0;
        double nx[3];
        for (int i = 0; i < 3; i++) nx[i] = x[i] + opt.x * dir[i];  // optimal_point.vector_x
        for (int i = 0; i < 3; i++) sn += (nx[i] - x[i]) * (nx[i] - x[i]);
        sn = sqrt(sn);
        for (int i = 0; i < 3; i++) x[i] = nx[i];
        final_cost = cur_cost;
        if (sn <= parameter_tolerance * (xnorm + parameter_tolerance)) {
          term = 3;
          break;
        }
        if (gmax <= gradient_tolerance) {
          term = 1;
          break;
        }
        if (fabs(prev_cost - cur_cost) <= function_tolerance * fabs(prev_cost)) {
          term = 2;
          break;
        }
      }
    }
    R.iterations = iter;
    R.termination = term;
    R.corr_opt = -final_cost / denom;
    R.tf_opt[0] = x[0];
    R.tf_opt[1] = x[1];
    R.tf_opt[2] = x[2];
    }
    if (sl == 0) results[pidx] = R;
#ifdef CC_TUNE_GMM_CLK
    if (sl == 0) {
      const unsigned long long t_p2 = __builtin_readcyclecounter();
      const int e = atomicAdd(&cc_gmm_clk_n, 1);
      if (e < CC_GMM_CLK_CAP) {
        cc_gmm_clk[e * 4 + 0] = (unsigned long long)np | ((unsigned long long)G << 20) | ((unsigned long long)R.iterations << 40) | ((unsigned long long)n_ev << 48);
        cc_gmm_clk[e * 4 + 1] = t_p2 - t_p0;
        cc_gmm_clk[e * 4 + 2] = ev_clk;
        cc_gmm_clk[e * 4 + 3] = t_p1 - t_p0;
        cc_gmm_clk2[e * 4 + 0] = t_w0;
        cc_gmm_clk2[e * 4 + 1] = wall_clock64();
        cc_gmm_clk2[e * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
        cc_gmm_clk2[e * 4 + 3] = (unsigned long long)blockIdx.x;
      }
    }
#endif
  }
}
// tidyUpCandidates' order-changing compaction (contour_db.h:580-592) followed by fineOptimize's std::sort on the
// still-all-zero correlation_ (contour_db.h:604-610), for one wave.
//   compaction: the sequential two-pointer loop swaps the k-th candidate without an estimate, counted from the front,
//     with the k-th candidate with one, counted from the back, until the pointers cross -- so the first n positions
//     (n = candidates with an estimate) end up holding: the candidate itself where it has one, else the matching one
//     from the back part.  Found with ballots, 64 positions per round.
//   sort: a comparator that is always false makes libstdc++'s introsort a fixed permutation of its n inputs (median and
//     partition swaps that never look at the values; the final insertion sort moves nothing): perm_tab holds it for
//     every n (built on the host with std::sort itself, cc_db_create), row n at offset n (n - 1) / 2.
// idx[0..n) = the candidates in the order fineOptimize sees them; returns n.  scr: CC_MAXCAND u16 of LDS scratch.
__device__ __forceinline__ int cc_tidy_order(int nc, const unsigned char *has, unsigned short *idx, unsigned short *scr,
                                             const unsigned short *__restrict__ perm_tab, int lane) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  int n = 0;
  for (int k0 = 0; k0 < nc; k0 += 64) n += __popcll(__ballot(k0 + lane < nc && has[k0 + lane]));
  // front part: positions < n without an estimate, ascending -> scr[0..nf); back part: positions >= n with one,
  // ascending -> scr[CC_MAXCAND - 1 - j] (read back descending)
  int nf = 0, nb = 0;
  for (int k0 = 0; k0 < nc; k0 += 64) {
    const int k = k0 + lane;
    const bool fr = k < n && !has[k], bk = k >= n && k < nc && has[k];
    const unsigned long long mf = __ballot(fr), mb = __ballot(bk);
    if (fr) scr[nf + __popcll(mf & lt)] = (unsigned short)k;
    if (bk) scr[CC_MAXCAND - 1 - (nb + __popcll(mb & lt))] = (unsigned short)k;
    nf += __popcll(mf);
    nb += __popcll(mb);
  }
  __syncthreads();
  // nf == nb; the a-th front gap takes the a-th estimate from the back, i.e. the (nb - 1 - a)-th in ascending order
  unsigned short mine[CC_MAXCAND / 64];
#pragma unroll
  for (int u = 0; u < CC_MAXCAND / 64; u++) {
    const int i = u * 64 + lane;
    mine[u] = 0;
    if (i < n) {
      const int p = (int)perm_tab[(size_t)n * (n - 1) / 2 + i];  // position (after the compaction) that the sort puts at i
      mine[u] = (unsigned short)p;
    }
  }
  __syncthreads();
  for (int a = lane; a < nf; a += 64) idx[scr[a]] = scr[CC_MAXCAND - 1 - (nb - 1 - a)];  // idx was the identity
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CC_MAXCAND / 64; u++) {
    const int i = u * 64 + lane;
    if (i < n) mine[u] = idx[mine[u]];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CC_MAXCAND / 64; u++) {
    const int i = u * 64 + lane;
    if (i < n) idx[i] = mine[u];
  }
  __syncthreads();
  return n;
}

// ------------------------------------------------------------------------------------------------
// K5s: which candidates of a query get refined -- the first max_fine_opt_ of candidates_ after tidyUpCandidates'
// compaction and fineOptimize's sort on the still-all-zero correlation_ (the same replay as in K6; contour_db.h:560-616).
// Their GMM problems are appended to sel_list.  One wave per query.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
cc_k_select(int nq, float corr_lb, int max_fine_opt, const cc_cand_out *__restrict__ cands_all, const cc_qstate *__restrict__ qstate,
            const cc_gmm_result *__restrict__ gres, int *__restrict__ sel_list /*[3][sel_stride]*/, int sel_stride,
            int *__restrict__ n_sel /*[2]*/, int *__restrict__ n_sel_wide, const unsigned short *__restrict__ perm_tab,
            int *__restrict__ cls_list /*[CC_GMM_NCLS][sel_stride]: the long problems by length class*/, int *__restrict__ cls_cnt /*[CC_GMM_NCLS]*/,
            int *__restrict__ pool_head, int *__restrict__ pool_off /*[problem slot]: where a selected problem's records go in the pair pool*/) {
  __shared__ unsigned short idx[CC_MAXCAND];
  __shared__ unsigned short scr[CC_MAXCAND];
  __shared__ unsigned char has[CC_MAXCAND];
  __shared__ int gm[CC_MAXCAND];
  __shared__ int s_off[4];
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= nq) return;
  const cc_cand_out *cands = cands_all + (size_t)q * CC_MAXCAND;
  const int nc = qstate[q].n_cand;
  for (int k = lane; k < nc; k += 64) {
    const int g = cands[k].gmm_idx;
    idx[k] = (unsigned short)k;
    gm[k] = g;
    has[k] = (g >= 0 && !((float)gres[g].corr_init < corr_lb)) ? 1 : 0;
  }
  __syncthreads();
  const int n = cc_tidy_order(nc, has, idx, scr, perm_tab, lane);
  if (n <= 0) return;
  const int pre = max_fine_opt < n ? max_fine_opt : n;
  // three lists by pair count: the 16-lane refinement instance, the 64-lane one, and the problems in between, which go to
  // one or the other by the chunk's problem count (cc_k_gmm_refine)
  // (one atomic per list and query, not one per problem: same-address atomics are served one after the other)
  const unsigned long long lt = (1ull << lane) - 1ull;
  // ... and the pair pool is handed out here as well (the part of a problem's records that does not stay in LDS, an even
  // count): a problem that asked for its share itself, at the start of the refinement, stood in a queue of thousands
  // (same-address atomics are served ~9 ns apart: the 64-lane launch took 30 us to get its 2 048 waves going)
  int n_big = 0, n_wide = 0, need_tot = 0;
  for (int i0 = 0; i0 < pre; i0 += 64) {
    const int i = i0 + lane;
    const int np = i < pre ? gres[gm[idx[i]]].n_pairs : 0;
    const bool big = np > CC_GMM_MID_MAX_PAIRS;
    n_big += __popcll(__ballot(big));
    n_wide += __popcll(__ballot(np > CC_GMM_G16_MAX_PAIRS && np <= CC_GMM_MID_MAX_PAIRS));
    const int nl = big ? CC_GMM_NL : CC_GMM_NL / (64 / CC_G);  // (an in-between problem: the smaller share, whichever instance takes it)
    int need = i < pre && np > nl ? ((np - nl + 1) & ~1) : 0;
    for (int o = 32; o > 0; o >>= 1) need += __shfl_xor(need, o);
    need_tot += need;
  }
  const int n_small = pre - n_big - n_wide;
  {  // the four requests at once, from four lanes
    int *const dst = lane == 0 ? &n_sel[0] : (lane == 1 ? &n_sel[1] : (lane == 2 ? n_sel_wide : pool_head));
    const int amt = lane == 0 ? n_small : (lane == 1 ? n_big : (lane == 2 ? n_wide : need_tot));
    if (lane < 4) s_off[lane] = amt ? atomicAdd(dst, amt) : 0;
  }
  __syncthreads();
  int o_small = s_off[0], o_big = s_off[1], o_wide = s_off[2], o_pool = s_off[3];
  for (int i0 = 0; i0 < pre; i0 += 64) {
    const int i = i0 + lane;
    const int g = i < pre ? gm[idx[i]] : 0;
    const int np = i < pre ? gres[g].n_pairs : 0;
    {
      const int nl = np > CC_GMM_MID_MAX_PAIRS ? CC_GMM_NL : CC_GMM_NL / (64 / CC_G);
      const int need = i < pre && np > nl ? ((np - nl + 1) & ~1) : 0;
      int incl = need;
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (i < pre) pool_off[g] = o_pool + incl - need;
      o_pool += __shfl(incl, 63);
    }
    const bool wide = i < pre && np > CC_GMM_G16_MAX_PAIRS && np <= CC_GMM_MID_MAX_PAIRS /* the in-between list */, big = i < pre && np > CC_GMM_MID_MAX_PAIRS,
               small = i < pre && !wide && !big;
    const unsigned long long mw = __ballot(wide), mbig = __ballot(big), msm = __ballot(small);
    if (wide) sel_list[2 * (size_t)sel_stride + o_wide + __popcll(mw & lt)] = g;
    if (mbig) {  // (uniform) a handful per query: one atomic per class that occurs
      const int c = big ? cc_gmm_len_class(np) : -1;
      for (unsigned long long left = mbig; left;) {
        const int c0 = __shfl(c, __ffsll(left) - 1);
        const unsigned long long mc = __ballot(c == c0);
        int base = 0;
        if (lane == __ffsll(mc) - 1) base = atomicAdd(&cls_cnt[c0], __popcll(mc));
        base = __shfl(base, __ffsll(mc) - 1);
        if (c == c0) cls_list[(size_t)c0 * sel_stride + base + __popcll(mc & lt)] = g;
        left &= ~mc;
      }
    }
    if (small) sel_list[o_small + __popcll(msm & lt)] = g;
    o_wide += __popcll(mw);
    o_big += __popcll(mbig);
    o_small += __popcll(msm);
  }
}

// ------------------------------------------------------------------------------------------------
// K6: per query, the rest of tidyUpCandidates (correlation bar + order-changing compaction, contour_db.h:560-592) and
// fineOptimize (contour_db.h:604-648): std::sort on the still-all-zero correlation_ (replayed), take the first
// max_fine_opt_, adopt their refined score/pose, re-sort those, return the best.  One lane per query.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
cc_k_final(int nq, float corr_lb, int max_fine_opt, const cc_cand_out *__restrict__ cands_all, const cc_qstate *__restrict__ qstate,
           const cc_gmm_result *__restrict__ gres, const int *__restrict__ pass_cnt, const int *__restrict__ hit_cnt,
           const cc_hot_desc_t *__restrict__ qhot, cc_query_result_t *__restrict__ out, const unsigned short *__restrict__ perm_tab,
           const int *__restrict__ nprob /*the chunk's problem counters and pool head*/, int *__restrict__ nprob_host /*or nullptr: copy them there*/) {
  // one wave per query: lanes fetch the per-candidate inputs and order the candidates in parallel (cc_tidy_order), lane 0
  // replays the short order-dependent rest on LDS
  __shared__ unsigned short idx[CC_MAXCAND];
  __shared__ unsigned short scr[CC_MAXCAND];
  __shared__ unsigned char has[CC_MAXCAND];
  __shared__ float corr_o[CC_MAXCAND];
  __shared__ int gm[CC_MAXCAND];
  __shared__ int s_tot;
  __shared__ unsigned stk[CC_SORT_STACK];
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= nq) return;
  if (nprob_host && q == 0 && lane < 4) nprob_host[lane] = nprob[lane];  // small chunks: no copy command behind the chain
  const cc_cand_out *cands = cands_all + (size_t)q * CC_MAXCAND;
  const int nc = qstate[q].n_cand;
  // what lane 0 needs of the query's counters, fetched up front (every load it would issue later is a round trip)
  const int pc0 = pass_cnt[q * 4 + 0], pc1 = pass_cnt[q * 4 + 1], pc2 = pass_cnt[q * 4 + 2], pc3 = pass_cnt[q * 4 + 3];
  const int qflags = qhot[q].flags;
  int gfl = 0;  // capacity flags of the query's correlation problems (every one of them gates or ranks a candidate)
  for (int k = lane; k < nc; k += 64) {
    const int g = cands[k].gmm_idx;
    idx[k] = (unsigned short)k;
    gm[k] = g;
    bool h = false;
    float co = 0.f;
    if (g >= 0) {
      h = !((float)gres[g].corr_init < corr_lb);
      co = (float)gres[g].corr_opt;
      gfl |= gres[g].flags;
    }
    has[k] = h ? 1 : 0;
    corr_o[k] = co;
  }
  for (int o = 32; o > 0; o >>= 1) gfl |= __shfl_xor(gfl, o);
  int tot = 0;
  for (int s2 = lane; s2 < CC_NQLEV * CC_NPIV; s2 += 64) tot += hit_cnt[q * CC_NQLEV * CC_NPIV + s2];
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  if (lane == 0) s_tot = tot;
  __syncthreads();
  // two-pointer compaction of candidates_ (has = corr_est_ != nullptr), contour_db.h:580-592, and the first std::sort:
  // every anch_props_[0].correlation_ is still 0 -> the comparator is always false
  const int n = cc_tidy_order(nc, has, idx, scr, perm_tab, lane);
  if (lane != 0) return;
  cc_query_result_t r;
  r.n_res = 0;
  r.cand_gidx = -1;
  r.correlation = 0;
  r.tf[0] = r.tf[1] = r.tf[2] = 0;
  r.cand_aft_check1 = pc1;
  r.cand_aft_check2 = pc2;
  r.cand_aft_check3 = pc3;
  r.n_cand_pose = nc;
  r.n_knn_hits = s_tot;
  r.flags = (pc0 & CC_QF_CHECK_CAP) | ((gfl & 1) ? CC_QF_GMM_CAP : 0) | ((gfl & 4) ? CC_QF_DESC_CAP : 0) |
            ((qflags & (CC_DESC_INEXACT_COMPONENTS | CC_DESC_INEXACT_KEYS)) ? CC_QF_QUERY_INEXACT : 0);
  r.pad_ = 0;
  r.n_cand_tidy = n;
  if (n > 0) {
    const int pre = max_fine_opt < n ? max_fine_opt : n;
    // candidates beyond `pre` keep correlation_ = 0
    ccsort::std_sort(idx, pre, [&](unsigned short a, unsigned short b) { return corr_o[a] > corr_o[b]; }, stk);
    const int b = idx[0];
    r.n_res = 1;
    r.cand_gidx = cands[b].gidx;
    r.correlation = pre > 0 ? (double)corr_o[b] : 0.0;
    if (pre > 0) {
      const cc_gmm_result *g = &gres[gm[b]];
      // T_best_ = Identity.rotate(theta).pretranslate(x, y); reported as (x, y, atan2(T10, T00))
      r.tf[0] = g->tf_opt[0];
      r.tf[1] = g->tf_opt[1];
      r.tf[2] = atan2(sin(g->tf_opt[2]), cos(g->tf_opt[2]));
    }
  }
  out[q] = r;
}
