// Query-side kernels.  Replace, for a batch of query scans against the device-resident DB,
//   K3  LayerDB::layerKNNSearch / TreeBucket::knnSearch / nanoflann kNN     src/cont2/contour_db.cpp:319-403
//   K4  CandidateManager::checkCandWithHint: ContourView::checkSim,           contour_db.h:374-488
//       BCI::checkConstellSim, checkConstellCorrespSim, getTFFromConstell     contour.h:278-329, contour_mng.h:288-388,1124-1277
//   K5  GMMPair / ConstellCorrelation::initProblem + calcCorrelation          correlation.h:42-238
//   K4b CandidatePoseData::addProposal, tidyUpCandidates (selection part)         contour_db.h:286-338,494-546
//   K6  tidyUpCandidates (compaction), fineOptimize ordering                       contour_db.h:560-648
// A query batch is one launch chain (knn -> check a/b/c -> merge -> gmm prep/small/large -> final) and one D2H copy.
#pragma once
#include "cc_dev.h"
#include "cc_sort.h"

// LDS hand-off between the lanes of one group: the lanes of a wave run in lockstep, so only the compiler has to be kept
// from reordering; the G-wide shuffle doubles as the rendezvous under the CPU test harness.
__device__ __forceinline__ void cc_group_sync(int G) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  (void)__shfl(0, 0, G);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// K3: k nearest retrieval keys with the reference's visibility rules.
//
// The reference keeps one kd-tree per bucket (nanoflann, src/cont2/contour_db.cpp:319-403); what it returns per anchor
// key is the nnk nearest keys (squared L2 over 10 dims, ties by insertion) among the visible ones, within dist_ub.
// Here each layer keeps its keys SORTED BY THE FIRST DIMENSION (the widest one, sqrt(eig_large * cell_cnt)); a search
// starts at the anchor's own position and walks outwards in both directions, 64 keys at a time, and a direction stops
// when (key[0] - q[0])^2 alone reaches the current radius -- the 1-D form of the kd-tree's pruning rule.  The radius
// starts at dist_ub and drops to the nnk-th best distance as soon as nnk candidates are known, so a search typically
// touches a few hundred keys of tens of thousands.
// ------------------------------------------------------------------------------------------------
// LDS candidate buffer per search (entries of 8 B).  Between two tightenings at most 2 * nnk - 1 kept candidates + one
// 64-key step are pending (191 at nnk = CC_KNN_MAX = 64), and the bitonic sort pads that to the next power of two.
#define CC_KNN_CAP 256
static_assert(2 * CC_KNN_MAX - 1 + 64 <= CC_KNN_CAP && (CC_KNN_CAP & (CC_KNN_CAP - 1)) == 0,
              "cc_k_knn: the padded sort width must fit the LDS buffer");

struct cc_knn_params {
  const float *skeys[CC_NQLEV];       // SoA [CC_KEY_DIM][cap_k], sorted by dim 0 (ties: insertion order)
  const int *sid[CC_NQLEV];           // insertion index (key id) of the i-th sorted key
  const int *sact[CC_NQLEV];          // first epoch at which that key sits in a tree
  const int *kgidx[CC_NQLEV];         // by key id: scan index
  const unsigned char *kseq[CC_NQLEV];
  int n_sorted[CC_NQLEV];             // keys in the layer
  int cap_k;
  int nnk;
  int n_q_levels;
  int q_levels[CC_NQLEV];
  int dbg_cut;  // tuning aid (env CC_KNN_CUT)
};

struct cc_query_meta {  // per query scan, host-built
  int epoch;
  int n_keys[CC_NQLEV];            // keys appended to the layer before this epoch
  float ranges[CC_NQLEV][7];       // LayerDB::bucket_ranges_ at this epoch
};

__device__ __forceinline__ void cc_bitonic_sort_u64(unsigned long long *a, int n_pow2, int tid, int nt) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pow2; i += nt) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long x = a[i], y = a[ixj];
          bool up = ((i & k) == 0);
          if ((x > y) == up) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---- maintenance of the sorted view when keys are appended (ids [n_old, n_old + m) of the insertion-ordered arrays).
// A: every new key finds its rank among the new keys (brute force, tiles through LDS) and among the old sorted ones
//    (binary search); B: every old key is shifted by the number of new keys below it; both write into the other buffer.
// grid = ceil(m / 256), block = 256
__global__ void __launch_bounds__(256)
cc_k_ksort_new(const float *__restrict__ keys /*insertion order, SoA*/, int cap_k, int n_old, int m,
               const float *__restrict__ s_old0 /*sorted dim 0, n_old entries*/, int *__restrict__ newpos,
               float *__restrict__ new_sorted0) {
  __shared__ float tile[256];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const float c = j < m ? keys[n_old + j] : 0.f;
  int r_new = 0;
  for (int t0 = 0; t0 < m; t0 += 256) {
    __syncthreads();
    tile[threadIdx.x] = (t0 + (int)threadIdx.x < m) ? keys[n_old + t0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int lim = m - t0 < 256 ? m - t0 : 256;
    for (int i = 0; i < lim; i++) {
      const float ci = tile[i];
      r_new += (ci < c || (ci == c && t0 + i < j)) ? 1 : 0;
    }
  }
  if (j >= m) return;
  int lo = 0, hi = n_old;  // #old keys with c0 <= c (old keys precede new ones among equals)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (s_old0[mid] <= c)
      lo = mid + 1;
    else
      hi = mid;
  }
  newpos[j] = lo + r_new;
  new_sorted0[r_new] = c;
}

// grid = ceil((n_old + m) / 256), block = 256
__global__ void __launch_bounds__(256)
cc_k_ksort_merge(const float *__restrict__ keys, int cap_k, int n_old, int m, const float *__restrict__ s_old, const int *__restrict__ sid_old,
                 const int *__restrict__ newpos, const float *__restrict__ new_sorted0, float *__restrict__ s_new, int *__restrict__ sid_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_old) {
    const float c = s_old[i];
    int lo = 0, hi = m;  // #new keys with c0 < c
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (new_sorted0[mid] < c)
        lo = mid + 1;
      else
        hi = mid;
    }
    const int dst = i + lo;
    for (int d = 0; d < CC_KEY_DIM; d++) s_new[(size_t)d * cap_k + dst] = s_old[(size_t)d * cap_k + i];
    sid_new[dst] = sid_old[i];
  } else if (i < n_old + m) {
    const int j = i - n_old, dst = newpos[j];
    for (int d = 0; d < CC_KEY_DIM; d++) s_new[(size_t)d * cap_k + dst] = keys[(size_t)d * cap_k + n_old + j];
    sid_new[dst] = n_old + j;
  }
}

// activation epochs in sorted order (they change when the host moves keys from a bucket's buffer into its tree)
__global__ void __launch_bounds__(256)
cc_k_ksort_act(const int *__restrict__ act, const int *__restrict__ sid, int n, int *__restrict__ sact) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sact[i] = act[sid[i]];
}

// grid = nq * CC_NQLEV * CC_NPIV, block = 64 (one wave per anchor key)
__global__ void __launch_bounds__(64)
cc_k_knn(cc_knn_params P, const cc_scan_desc_t *__restrict__ qdesc, const cc_query_meta *__restrict__ qmeta,
         cc_knn_hit_t *__restrict__ hits, int *__restrict__ hit_cnt) {
  __shared__ unsigned long long buf[CC_KNN_CAP];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x % (CC_NQLEV * CC_NPIV);
  const int q = blockIdx.x / (CC_NQLEV * CC_NPIV);
  const int ll = slot / CC_NPIV, seq = slot - ll * CC_NPIV;
  cc_knn_hit_t *out = hits + (size_t)blockIdx.x * CC_KNN_MAX;
  if (ll >= P.n_q_levels) {
    if (lane == 0) hit_cnt[blockIdx.x] = 0;
    return;
  }
  const int level = P.q_levels[ll];
  const float *qk = &qdesc[q].keys[level][seq][0];
  float k[CC_KEY_DIM];
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < CC_KEY_DIM; d++) {
    k[d] = qk[d];
    sum += k[d];
  }
  if (!(sum != 0.f)) {  // q_keys[seq].sum() != 0 (contour_db.h:726)
    if (lane == 0) hit_cnt[blockIdx.x] = 0;
    return;
  }
  const cc_query_meta qm = qmeta[q];
  // dist_ub (contour_db.h:733-749), f32 results of f64 products exactly as written there
  const float b00 = (float)((double)k[0] * 0.8), b01 = (float)((double)k[0] / 0.8);
  const float b10 = (float)((double)k[1] * 0.8), b11 = (float)((double)k[1] / 0.8);
  const float b20 = (float)((double)k[2] * 0.8 * 0.75), b21 = (float)((double)k[2] / (0.8 * 0.75));
  const float t0a = (k[0] - b00) * (k[0] - b00), t0b = (k[0] - b01) * (k[0] - b01);
  const float t1a = (k[1] - b10) * (k[1] - b10), t1b = (k[1] - b11) * (k[1] - b11);
  const float t2a = (k[2] - b20) * (k[2] - b20), t2b = (k[2] - b21) * (k[2] - b21);
  float ub = (t0a < t0b ? t0b : t0a) + (t1a < t1b ? t1b : t1a) + (t2a < t2b ? t2b : t2a);
  // mid bucket and the buckets layerKNNSearch actually visits (src/cont2/contour_db.cpp:322-369):
  // {0..mid} and {mid+i : i > mid, mid+i < 6}
  float rg[7];
#pragma unroll
  for (int i = 0; i < 7; i++) rg[i] = qm.ranges[ll][i];
  int mid = 0;
  {
    bool found = false;
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (!found && rg[i] <= k[0] && rg[i + 1] > k[0]) {
        mid = i;
        found = true;
      }
  }
  unsigned vis = 0;
#pragma unroll
  for (int b = 0; b < 6; b++)
    if (b <= mid || b >= 2 * mid + 1) vis |= 1u << b;
  const int n = P.n_sorted[ll];
  const float *K = P.skeys[ll];
  const int *sid = P.sid[ll];
  const int *sact = P.sact[ll];
  const int cap = P.cap_k;
  const int epoch = qm.epoch;
  const int nnk = P.nnk;
  // position of the anchor's first dimension in the sorted layer
  int right;
  {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int md = (lo + hi) >> 1;
      if (K[md] < k[0])
        lo = md + 1;
      else
        hi = md;
    }
    right = lo;
  }
  int left = right - 1;  // next index to visit on the low side
  int cnt = 0;
  bool tightened = false;
  bool open[2] = {right < n && P.dbg_cut != 1, left >= 0 && P.dbg_cut != 1};  // [0]: upwards, [1]: downwards
  // one 64-key step per direction and iteration; the next step's keys are loaded while the current one is scored
  float c[2][CC_KEY_DIM], cn[2][CC_KEY_DIM];
  int act[2], actn[2], kid[2], kidn[2];
  int id[2] = {right + lane, left - lane};
#pragma unroll
  for (int dir = 0; dir < 2; dir++) {
    act[dir] = 0x7fffffff;
    kid[dir] = 0;
#pragma unroll
    for (int d = 0; d < CC_KEY_DIM; d++) c[dir][d] = 0.f;
    if (id[dir] >= 0 && id[dir] < n) {
      act[dir] = sact[id[dir]];
      kid[dir] = sid[id[dir]];
#pragma unroll
      for (int d = 0; d < CC_KEY_DIM; d++) c[dir][d] = K[(size_t)d * cap + id[dir]];
    }
  }
  while (open[0] || open[1]) {
    int idn[2] = {id[0] + 64, id[1] - 64};
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      actn[dir] = 0x7fffffff;
      kidn[dir] = 0;
#pragma unroll
      for (int d = 0; d < CC_KEY_DIM; d++) cn[dir][d] = 0.f;
      if (open[dir] && idn[dir] >= 0 && idn[dir] < n) {
        actn[dir] = sact[idn[dir]];
        kidn[dir] = sid[idn[dir]];
#pragma unroll
        for (int d = 0; d < CC_KEY_DIM; d++) cn[dir][d] = K[(size_t)d * cap + idn[dir]];
      }
    }
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      if (!open[dir]) continue;  // wave-uniform
      bool pass = false;
      float res = 0.f;
      const bool inside = id[dir] >= 0 && id[dir] < n;
      const float c0 = c[dir][0];
      const float e0 = k[0] - c0;
      if (inside && act[dir] <= epoch) {
        int bk = -1;
#pragma unroll
        for (int b = 0; b < 6; b++)
          if (bk < 0 && rg[b] <= c0 && c0 < rg[b + 1]) bk = b;
        if (bk >= 0 && ((vis >> bk) & 1u)) {
          // L2_Adaptor::evalMetric accumulation order (nanoflann.hpp:427-461)
          float d0 = e0, d1 = k[1] - c[dir][1], d2 = k[2] - c[dir][2], d3 = k[3] - c[dir][3];
          res += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          d0 = k[4] - c[dir][4];
          d1 = k[5] - c[dir][5];
          d2 = k[6] - c[dir][6];
          d3 = k[7] - c[dir][7];
          res += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          d0 = k[8] - c[dir][8];
          res += d0 * d0;
          d0 = k[9] - c[dir][9];
          res += d0 * d0;
          // before nnk candidates are known a key must be strictly inside dist_ub; afterwards keys AT the nnk-th best
          // distance still compete, on the key id
          pass = tightened ? (res <= ub) : (res < ub);
        }
      }
      const unsigned long long m = __ballot(pass);
      if (pass) buf[cnt + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)__float_as_uint(res) << 32) | (unsigned)kid[dir];
      cnt += __popcll(m);
      __syncthreads();
      if (cnt >= 2 * nnk || (!tightened && cnt >= nnk)) {  // keep the best nnk (by distance, then key id); the radius follows
        int np2 = 64;
        while (np2 < cnt) np2 <<= 1;
        for (int i = cnt + lane; i < np2; i += 64) buf[i] = ~0ull;
        __syncthreads();
        cc_bitonic_sort_u64(buf, np2, lane, 64);
        ub = __uint_as_float((unsigned)(buf[nnk - 1] >> 32));
        cnt = nnk;
        tightened = true;
        __syncthreads();
      }
      // the step's outermost key decides whether the direction goes on: (key[0] - q[0])^2 is a lower bound of the
      // distance and grows outwards
      const int last_in = __builtin_amdgcn_readlane((int)inside, 63);
      const float e_far = cc_lane_bcast(e0, 63);
      const float far2 = e_far * e_far;
      open[dir] = last_in && (tightened ? (far2 <= ub) : (far2 < ub));
    }
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      id[dir] = idn[dir];
      act[dir] = actn[dir];
      kid[dir] = kidn[dir];
#pragma unroll
      for (int d = 0; d < CC_KEY_DIM; d++) c[dir][d] = cn[dir][d];
    }
  }
  {
    int np2 = 64;
    while (np2 < cnt) np2 <<= 1;
    for (int i = cnt + lane; i < np2; i += 64) buf[i] = ~0ull;
    __syncthreads();
    cc_bitonic_sort_u64(buf, np2, lane, 64);
    const int mm = cnt < nnk ? cnt : nnk;
    for (int i = lane; i < mm; i += 64) {
      const unsigned id = (unsigned)(buf[i] & 0xFFFFFFFFu);
      cc_knn_hit_t h;
      h.gidx = P.kgidx[ll][id];
      h.level = (int16_t)level;
      h.seq = (int16_t)P.kseq[ll][id];
      h.dist_sq = __uint_as_float((unsigned)(buf[i] >> 32));
      out[i] = h;
    }
    if (lane == 0) hit_cnt[blockIdx.x] = mm;
  }
}

// ------------------------------------------------------------------------------------------------
// K4: the four-stage gate per (candidate scan, anchor pair): stage A one lane per KNN hit, stage B 16 lanes per survivor,
// stage C one lane per passing check.
// ------------------------------------------------------------------------------------------------
#define CC_PP_MAX 256      // potential (src,tgt) neighbour pairs per check (large instance of stage B)
#define CC_PP_SMALL 64     // ... handled by the common, high-occupancy instance
#define CC_CSTL_MAX 64     // pairs kept in a constellation

struct cc_pass_rec {
  int q;           // query index within the launch
  int order;       // slot * CC_KNN_MAX + j : position in the reference's candidate iteration order
  int gidx;        // candidate scan
  int n_pairs;     // tmp_pairs2.size() (vote weight)
  int flags;       // bit0: a capacity (CC_PP_MAX / CC_CSTL_MAX) was hit
  int pad;
  double tf[3];    // T_pass = (x, y, theta)
  double cs[3];    // cos(theta), sin(theta), atan2(sin, cos): entries of the Isometry2d built by rotate(theta), hoisted out of
                   // the sequential merge
  unsigned long long bits[7];  // constellation pairs as a set: bit (level-1)*100 + seq_src*10 + seq_tgt
};

struct cc_check_params {
  cc_sim_cfg_t sim;
  cc_score_t lb;
  int n_q_levels;
  int q_levels[CC_NQLEV];
  int dbg_cut;  // tuning aid (env CC_CHKB_CUT): stage B stops after phase dbg_cut, 0 = run everything
};


__device__ __forceinline__ bool cc_check_sim(const cc_contour_t &a, const cc_contour_t &b, const cc_sim_cfg_t &th) {
  const float ca = (float)a.cell_cnt, cb = (float)b.cell_cnt;
  if ((fabsf((ca - cb) / (ca < cb ? cb : ca)) > th.tp_cell_cnt) && (fabsf(ca - cb) > th.ta_cell_cnt)) return false;
  {
    const float ea = a.eig_vals[1], eb = b.eig_vals[1];
    if ((ea < eb ? eb : ea) > 2.0f) {
      const float sa = sqrtf(ea), sb = sqrtf(eb);
      if (fabsf((sa - sb) / (sa < sb ? sb : sa)) > th.tp_eigval) return false;
    }
  }
  {
    const float ea = a.eig_vals[0], eb = b.eig_vals[0];
    if ((ea < eb ? eb : ea) > 2.0f) {
      const float sa = sqrtf(ea), sb = sqrtf(eb);
      if (fabsf((sa - sb) / (sa < sb ? sb : sa)) > th.tp_eigval) return false;
    }
  }
  if ((a.cell_cnt < b.cell_cnt ? b.cell_cnt : a.cell_cnt) > 15 && fabsf(a.vol3_mean - b.vol3_mean) > th.ta_h_bar) return false;
  const float ax = a.com[0] - a.pos_mean[0], ay = a.com[1] - a.pos_mean[1];
  const float bx = b.com[0] - b.pos_mean[0], by = b.com[1] - b.pos_mean[1];
  const float r1 = sqrtf(ax * ax + ay * ay), r2 = sqrtf(bx * bx + by * by);
  if (fabsf(r1 - r2) > th.ta_rcom && fabsf((r1 - r2) / (r1 < r2 ? r2 : r1)) > th.tp_rcom) return false;
  return true;
}

__device__ __forceinline__ float cc_norm2f(float x, float y) { return sqrtf(x * x + y * y); }

#define CC_CHK_STRIDE (CC_NQLEV * CC_NPIV * CC_KNN_MAX)  // dense check slots per query: slot * CC_KNN_MAX + j
#define CC_NSCORE 5  // per-check gate scores (hint flow): ovlp_sum, max_one, in_ang_rng, indiv_sim, orie_sim
// A KNN hit names the candidate's anchor (level, seq); the query's anchor is implied by the slot.  In the hint flow
// (cc_db_check_hints) the checks sit in caller order instead, and the query's anchor rides in the high byte of `level`
// (0 = none: derive it from the slot).
#define CC_HIT_LEVEL(h) ((int)((h).level & 0xFF))
#define CC_HIT_SEQ_TGT(h, slot) (((h).level >> 8) ? (int)((h).level >> 8) - 1 : (slot) % CC_NPIV)
#define CC_HIT_PACK_LEVEL(level, seq_tgt) ((int16_t)((level) | (((seq_tgt) + 1) << 8)))

// Stage A (one lane per check slot): (1/4) anchor ContourView::checkSim and the popcount part of (2/4)
// BCI::checkConstellSim (ovlp_sum / max_one bars).  Survivors are written as an ORDERED list per query; the slot
// index t = slot * CC_KNN_MAX + j IS the reference's candidate iteration order (levels -> anchors -> ascending
// distance, contour_db.h:721-771), so later stages can replay checks in order without sorting.
// grid = nq, block = 256
__global__ void __launch_bounds__(256)
cc_k_check_a(cc_check_params P, const cc_scan_desc_t *__restrict__ qdesc, const cc_scan_desc_t *__restrict__ db_desc,
             const cc_knn_hit_t *__restrict__ hits, const int *__restrict__ hit_cnt, unsigned short *__restrict__ surv,
             cc_knn_hit_t *__restrict__ surv_hit, int *__restrict__ surv_cnt, unsigned char *__restrict__ pass_ok,
             int *__restrict__ pass_cnt /*[nq][4]*/, int *__restrict__ redo_cnt /*[nq]*/,
             int *__restrict__ scores /*[nq][CC_CHK_STRIDE][CC_NSCORE] or nullptr: per-check gate scores (hint flow)*/) {
  __shared__ int wcnt[4];
  __shared__ int s_base, s_chk1;
  const int q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int NS = CC_NQLEV * CC_NPIV;
  const int wave = tid >> 6, lane = tid & 63;
  if (tid == 0) {
    s_base = 0;
    s_chk1 = 0;
  }
  __syncthreads();
  const cc_scan_desc_t *tgt = qdesc + q;
  for (int t0 = 0; t0 < CC_CHK_STRIDE; t0 += nt) {
    const int t = t0 + tid;
    bool anchor_ok = false, keep = false;
    int sc_sum = 0, sc_max = 0;
    cc_knn_hit_t h;
    h.gidx = 0;
    h.level = h.seq = 0;
    h.dist_sq = 0.f;
    if (t < CC_CHK_STRIDE) {
      const int slot = t / CC_KNN_MAX, j = t - slot * CC_KNN_MAX;
      if (j < hit_cnt[q * NS + slot]) {
        h = hits[((size_t)q * NS + slot) * CC_KNN_MAX + j];
        const int seq_tgt = CC_HIT_SEQ_TGT(h, slot), lev = CC_HIT_LEVEL(h);
        const cc_scan_desc_t *src = db_desc + h.gidx;
        anchor_ok = cc_check_sim(src->cont[lev][h.seq], tgt->cont[lev][seq_tgt], P.sim);
        if (anchor_ok) {
          const cc_bci_t *bs = &src->bcis[lev][h.seq];
          const cc_bci_t *bt = &tgt->bcis[lev][seq_tgt];
          unsigned long long S[4], T[4];
          for (int w = 0; w < 4; w++) {
            S[w] = bs->dist_bin[w];
            T[w] = bt->dist_bin[w];
          }
          int ov1 = 0, ov2 = 0, ov3 = 0;
          for (int w = 0; w < 4; w++) {
            const unsigned long long shl = (S[w] << 1) | (w > 0 ? (S[w - 1] >> 63) : 0ull);
            const unsigned long long shr = (S[w] >> 1) | (w < 3 ? (S[w + 1] << 63) : 0ull);
            ov1 += __popcll(S[w] & T[w]);
            ov2 += __popcll(shl & T[w]);
            ov3 += __popcll(shr & T[w]);
          }
          const int ovlp_sum = ov1 + ov2 + ov3;
          int max_one = ov2 < ov3 ? ov3 : ov2;
          max_one = ov1 < max_one ? max_one : ov1;
          keep = (ovlp_sum >= P.lb.i_ovlp_sum && max_one >= P.lb.i_ovlp_max_one);
          sc_sum = ovlp_sum;
          sc_max = max_one;
        }
      }
      pass_ok[(size_t)q * CC_CHK_STRIDE + t] = 0;
      if (scores) {
        int *sc = scores + ((size_t)q * CC_CHK_STRIDE + t) * CC_NSCORE;
        sc[0] = sc_sum;
        sc[1] = sc_max;
        sc[2] = sc[3] = sc[4] = 0;
      }
    }
    const unsigned long long mk = __ballot(keep), ma = __ballot(anchor_ok);
    if (lane == 0) {
      wcnt[wave] = __popcll(mk);
      atomicAdd(&s_chk1, __popcll(ma));
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; w++) off += wcnt[w];
    off += __popcll(mk & ((1ull << lane) - 1ull));
    if (keep) {
      surv[(size_t)q * CC_CHK_STRIDE + off] = (unsigned short)t;
      surv_hit[(size_t)q * CC_CHK_STRIDE + off] = h;  // stage B reads the hit next to the slot instead of chasing it
    }
    __syncthreads();
    if (tid == 0) s_base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  if (tid == 0) {
    surv_cnt[q] = s_base;
    redo_cnt[q] = 0;
    pass_cnt[q * 4 + 0] = 0;
    pass_cnt[q * 4 + 1] = s_chk1;
    pass_cnt[q * 4 + 2] = 0;
    pass_cnt[q * 4 + 3] = 0;
  }
}

// Stage B (16 lanes per surviving check, 4 checks per wave): the rest of BCI::checkConstellSim (neighbour pairing,
// common-rotation window), checkConstellCorrespSim and getTFFromConstell.  Parallel where the reference's result does
// not depend on order, sequential (uniform loops on LDS) where it does.  All control flow is uniform per 16-lane group;
// the groups of a wave diverge freely, so every hand-off goes through cc_group_sync / 16-wide shuffles.
#define CC_CHKB_G 16
#define CC_CHKB_GPW (64 / CC_CHKB_G)
#define CC_CHKB_PER_Q 8  // stage-B workgroups (waves) per query (default; a launch parameter)

template <int PPM>
struct cc_chkb_lds {  // per group; the unions hold data of phases that never overlap in time
  unsigned long long bitsw[8];             // pair bitmap staging (first member: 8-byte aligned for the 64-bit LDS atomics)
  unsigned long long pp[PPM];              // potential pairs in generation order: fkey(orie) << 32 | l | s << 8 | t << 16
  union {
    int hist[PPM > 64 ? 256 : 2];          // sort: orientation bins (count -> start -> end), large instance only
    struct {
      float spm[CC_CSTL_MAX][2], tpm[CC_CSTL_MAX][2];  // pos_mean of the constellation's contours
    } m;
  };
  union {
    struct {                               // pair generation
      cc_relpt_t sp[CC_BCI_MAXPTS], tp[CC_BCI_MAXPTS];
      unsigned short off[CC_BCI_MAXPTS + 2];  // first potential pair of each tgt point
      unsigned char lo[CC_BCI_MAXPTS], hi[CC_BCI_MAXPTS];
    } g;
    float skey[PPM];                       // sort result -> window search: orie in sorted order
    struct {                               // constellation checks
      signed char cs[CC_CSTL_MAX][3];
      unsigned char keepf[CC_CSTL_MAX];
      float cn[48], nn[48];                // shaft candidates: length, length after normalisation
      int misc[4];
    } c;
  };
  unsigned char binidx[PPM];               // pair indices grouped by bin
  unsigned char sidx[PPM];                 // pair indices in sorted order
  short seg[3][20];                        // sort: pending quicksort segments (first, last, depth left)
};
static_assert(CC_PP_MAX <= 256, "pair indices are bytes");
static_assert(CC_BCI_MAXPTS <= 3 * CC_CHKB_G && sizeof(cc_relpt_t) == 12, "three 12-byte point loads per lane cover a BCI");

__device__ __forceinline__ unsigned cc_group_ballot(bool pred, int sl) {  // bit i = pred of group lane i
  int v = pred ? (1 << sl) : 0;
  for (int o = 1; o < CC_CHKB_G; o <<= 1) v |= __shfl_xor(v, o, CC_CHKB_G);
  return (unsigned)v;
}

// potential pairs (contour_mng.h:311-334): for tgt point i (ascending bit_pos) all src points with bit_pos within +-1, in
// src order.
template <int PPM>
__device__ __forceinline__ void cc_chkb_gen_pairs(cc_chkb_lds<PPM> &L, int ntp, int sl) {
  for (int i = sl; i < ntp; i += CC_CHKB_G) {
    const cc_relpt_t r2 = L.g.tp[i];
    int o = L.g.off[i];
    for (int sj = L.g.lo[i]; sj < L.g.hi[i]; sj++, o++) {
      if (o >= PPM) break;
      const cc_relpt_t r1 = L.g.sp[sj];
      float od = r2.theta - r1.theta;
      // clampAng (tools/algos.h:49-51): ang - floor((ang + pi) / (2 pi)) * 2 * pi in double.  |theta| <= float(pi), so the
      // quotient lies in (-0.51, 1.51) and its floor is -1 (x < 0), 1 (x >= 2 pi; the quotient of two doubles cannot round
      // up to 1 from below) or 0 -- two compares instead of an f64 division
      const double xw = (double)od + 3.14159265358979323846;
      const double kw = xw < 0.0 ? -1.0 : (xw >= 2 * 3.14159265358979323846 ? 1.0 : 0.0);
      od = (float)((double)od - kw * 2 * 3.14159265358979323846);
      L.pp[o] = ((unsigned long long)cc_fkey(od) << 32) |
                (unsigned long long)((unsigned)(r1.level & 0xFF) | ((unsigned)(r1.seq & 0xFF) << 8) | ((unsigned)(r2.seq & 0xFF) << 16));
    }
  }
}

// orientation bin of the counting sort: monotone non-decreasing in od (f32 add, multiply by a positive constant and
// truncation all preserve order), so bins partition the sorted sequence
__device__ __forceinline__ int cc_chkb_bin(float od) {
  int b = (int)((od + 3.14159274f) * 40.7436638f);
  return b < 0 ? 0 : (b > 255 ? 255 : b);
}

#define CC_CHKB_KEY(w) cc_funkey((unsigned)((w) >> 32))

// std::sort(potential_pairs, orie_diff <) (contour_mng.h:340).  Equal orie_diff are common (contour centres are means of
// integer cell coordinates, so revisits reproduce them bit for bit) and the reference's order among them is whatever
// libstdc++'s introsort leaves, so the algorithm is replayed -- in parallel, which its structure allows:
//   (1) median-of-3 Hoare partitions until every segment has <= 16 elements.  One partition is data-parallel: with the
//       positions of the elements !(x < pivot) in ascending order (l_k) and of the elements !(pivot < x) in descending
//       order (r_k), the sequential two-pointer loop swaps exactly the pairs (l_k, r_k) with l_k < r_k -- a prefix k < K --
//       and returns min(l_K+1, r_K): neither pointer ever re-reads a swapped position before they cross.
//   (2) the final insertion sort is a STABLE sort of what (1) left: rank = #smaller + #equal-and-earlier.
// The heapsort branch (depth limit 2*floor(log2 n) exhausted) is replayed serially by one lane on regenerated input.
// Result: L.sidx[k] = index into L.pp of the k-th pair, L.skey[k] = its orie_diff.
template <int PPM>
__device__ __noinline__ void cc_chkb_sort(cc_chkb_lds<PPM> &L, int npp, int ntp, int sl) {
  const int G = CC_CHKB_G;
  unsigned char *lpos = L.binidx, *rasc = L.sidx;  // stopper lists (both arrays are free until step 2)
  bool deep = false;
  cc_group_sync(G);  // the pairs are in place
  if (npp > 16) {
    int lg = 0;
    for (int t = npp; t > 1; t >>= 1) lg++;
    int nseg = 1;
    L.seg[0][0] = 0;
    L.seg[1][0] = (short)npp;
    L.seg[2][0] = (short)(lg * 2);
    cc_group_sync(G);
    while (nseg > 0) {
      nseg--;
      const int first = L.seg[0][nseg], last = L.seg[1][nseg];
      int depth = L.seg[2][nseg];
      if (depth == 0) {
        deep = true;
        break;
      }
      depth--;
      const int mid = first + (last - first) / 2;
      const int ia = first + 1, ib = mid, ic = last - 1;
      const float ka = CC_CHKB_KEY(L.pp[ia]), kb = CC_CHKB_KEY(L.pp[ib]), kc = CC_CHKB_KEY(L.pp[ic]);
      int sel;  // __move_median_to_first(first, first+1, mid, last-1)
      if (ka < kb) {
        if (kb < kc)
          sel = ib;
        else if (ka < kc)
          sel = ic;
        else
          sel = ia;
      } else if (ka < kc)
        sel = ia;
      else if (kb < kc)
        sel = ic;
      else
        sel = ib;
      cc_group_sync(G);
      if (sl == 0) {
        const unsigned long long t = L.pp[first];
        L.pp[first] = L.pp[sel];
        L.pp[sel] = t;
      }
      cc_group_sync(G);
      const float piv = CC_CHKB_KEY(L.pp[first]);
      int nL = 0, nR = 0;
      for (int r0 = first + 1; r0 < last; r0 += G) {
        const int i = r0 + sl;
        bool ls = false, rs = false;
        if (i < last) {
          const float k = CC_CHKB_KEY(L.pp[i]);
          ls = !(k < piv);
          rs = !(piv < k);
        }
        const unsigned mL = cc_group_ballot(ls, sl), mR = cc_group_ballot(rs, sl);
        if (ls) lpos[nL + __popc(mL & ((1u << sl) - 1u))] = (unsigned char)i;
        if (rs) rasc[nR + __popc(mR & ((1u << sl) - 1u))] = (unsigned char)i;
        nL += __popc(mL);
        nR += __popc(mR);
      }
      cc_group_sync(G);
      const int nmin = nL < nR ? nL : nR;
      int K = 0;
      for (int k0 = 0; k0 < nmin; k0 += G) {
        const int k = k0 + sl;
        K += __popc(cc_group_ballot(k < nmin && lpos[k] < rasc[nR - 1 - k], sl));
      }
      for (int k = sl; k < K; k += G) {
        const int a = lpos[k], b = rasc[nR - 1 - k];
        const unsigned long long t = L.pp[a];
        L.pp[a] = L.pp[b];
        L.pp[b] = t;
      }
      int cut = 0x7fff;
      if (K < nL) cut = lpos[K];
      if (K > 0 && (int)rasc[nR - K] < cut) cut = rasc[nR - K];
      cc_group_sync(G);
      if (last - cut > 16) {
        L.seg[0][nseg] = (short)cut;
        L.seg[1][nseg] = (short)last;
        L.seg[2][nseg] = (short)depth;
        nseg++;
      }
      if (cut - first > 16) {
        L.seg[0][nseg] = (short)first;
        L.seg[1][nseg] = (short)cut;
        L.seg[2][nseg] = (short)depth;
        nseg++;
      }
      cc_group_sync(G);
    }
  }
  if (deep) {
    cc_group_sync(G);
    cc_chkb_gen_pairs(L, ntp, sl);
    cc_group_sync(G);
    if (sl == 0)
      ccsort::std_sort(L.pp, npp, [](const unsigned long long &x, const unsigned long long &y) { return CC_CHKB_KEY(x) < CC_CHKB_KEY(y); });
    cc_group_sync(G);
    for (int k = sl; k < npp; k += G) {
      L.sidx[k] = (unsigned char)k;
      L.skey[k] = CC_CHKB_KEY(L.pp[k]);
    }
    cc_group_sync(G);
    return;
  }
  if (npp <= G) {  // one pair per lane: stable rank through 16-wide shuffles, no LDS traffic
    const float f = sl < npp ? CC_CHKB_KEY(L.pp[sl]) : 0.f;
    int rank = 0;
    for (int j = 0; j < npp; j++) {
      const float fj = __shfl(f, j, G);
      rank += (fj < f || (fj == f && j < sl)) ? 1 : 0;
    }
    if (sl < npp) {
      L.sidx[rank] = (unsigned char)sl;
      L.skey[rank] = f;
    }
    cc_group_sync(G);
    return;
  }
  if (npp <= 48 || PPM <= 64) {
    for (int p = sl; p < npp; p += G) {
      const float f = CC_CHKB_KEY(L.pp[p]);
      int rank = 0;
      for (int j = 0; j < npp; j++) {
        const float fj = CC_CHKB_KEY(L.pp[j]);
        rank += (fj < f || (fj == f && j < p)) ? 1 : 0;
      }
      L.sidx[rank] = (unsigned char)p;
      L.skey[rank] = f;
    }
    cc_group_sync(G);
    return;
  }
  if constexpr (PPM > 64) {
  // counting sort on orientation bins + exact rank inside a bin
  for (int i = sl; i < 256; i += G) L.hist[i] = 0;
  cc_group_sync(G);
  for (int o = sl; o < npp; o += G) atomicAdd(&L.hist[cc_chkb_bin(CC_CHKB_KEY(L.pp[o]))], 1);
  cc_group_sync(G);
  {
    int loc[256 / CC_CHKB_G];
    int sum = 0;
    for (int u = 0; u < 256 / CC_CHKB_G; u++) {
      loc[u] = L.hist[sl * (256 / CC_CHKB_G) + u];
      sum += loc[u];
    }
    int incl = sum;
    for (int o = 1; o < G; o <<= 1) {
      const int v = __shfl_up(incl, o, G);
      if (sl >= o) incl += v;
    }
    int run = incl - sum;
    for (int u = 0; u < 256 / CC_CHKB_G; u++) {
      L.hist[sl * (256 / CC_CHKB_G) + u] = run;  // start of the bin; used as the scatter cursor next
      run += loc[u];
    }
  }
  cc_group_sync(G);
  for (int o = sl; o < npp; o += G) {
    const int pos = atomicAdd(&L.hist[cc_chkb_bin(CC_CHKB_KEY(L.pp[o]))], 1);
    L.binidx[pos] = (unsigned char)o;
  }
  cc_group_sync(G);  // hist[b] is now the END of bin b
  for (int p = sl; p < npp; p += G) {
    const int o = L.binidx[p];
    const float f = CC_CHKB_KEY(L.pp[o]);
    const int bn = cc_chkb_bin(f);
    const int start = bn ? L.hist[bn - 1] : 0, end = L.hist[bn];
    int rank = 0;
    for (int p2 = start; p2 < end; p2++) {
      const int o2 = L.binidx[p2];
      const float f2 = CC_CHKB_KEY(L.pp[o2]);
      rank += (f2 < f || (f2 == f && o2 < o)) ? 1 : 0;
    }
    L.sidx[start + rank] = (unsigned char)o;
    L.skey[start + rank] = f;
  }
  cc_group_sync(G);
  }
}

// Two instances: <CC_PP_SMALL, false> handles every check with <= 64 potential pairs (12 KB of LDS per workgroup) and
// marks the others (pass_ok = 2); <CC_PP_MAX, true> then runs only those.
// grid = nq * per_q, block = 64
template <int PPM, bool REDO>
__global__ void __launch_bounds__(64)
cc_k_check_b(cc_check_params P, const cc_scan_desc_t *__restrict__ qdesc, const cc_scan_desc_t *__restrict__ db_desc,
             const cc_knn_hit_t *__restrict__ surv_hit, const unsigned short *__restrict__ surv, const int *__restrict__ surv_cnt,
             cc_pass_rec *__restrict__ pass, unsigned char *__restrict__ pass_ok, int *__restrict__ pass_cnt,
             int *__restrict__ redo_cnt, int per_q, int *__restrict__ scores /*see cc_k_check_a; or nullptr*/) {
  __shared__ cc_chkb_lds<PPM> LG[CC_CHKB_GPW];
  if (REDO && redo_cnt[blockIdx.x / per_q] == 0) return;  // nothing was left over for this query
  const int G = CC_CHKB_G;
  const int q = blockIdx.x / per_q, part = blockIdx.x % per_q;
  const int sub = threadIdx.x / CC_CHKB_G, sl = threadIdx.x % CC_CHKB_G;
  cc_chkb_lds<PPM> &L = LG[sub];
  const cc_scan_desc_t *tgt = qdesc + q;
  const int ns = surv_cnt[q];
  // the (slot, hit) of the next check is fetched while the current one is worked on
  const int si0 = part * CC_CHKB_GPW + sub;
  int t_nxt = 0;
  cc_knn_hit_t h_nxt;
  h_nxt.gidx = 0;
  h_nxt.level = h_nxt.seq = 0;
  h_nxt.dist_sq = 0.f;
  if (si0 < ns) {
    t_nxt = surv[(size_t)q * CC_CHK_STRIDE + si0];
    h_nxt = surv_hit[(size_t)q * CC_CHK_STRIDE + si0];
  }
  for (int si = si0; si < ns; si += per_q * CC_CHKB_GPW) {
    const int t = t_nxt;
    const cc_knn_hit_t h = h_nxt;
    {
      const int sn = si + per_q * CC_CHKB_GPW;
      if (sn < ns) {
        t_nxt = surv[(size_t)q * CC_CHK_STRIDE + sn];
        h_nxt = surv_hit[(size_t)q * CC_CHK_STRIDE + sn];
      }
    }
    if (REDO) {
      const bool left_over = pass_ok[(size_t)q * CC_CHK_STRIDE + t] == 2;
      cc_group_sync(G);  // every lane of the group has read the mark before lane 0 clears it
      if (!left_over) continue;
      if (sl == 0) pass_ok[(size_t)q * CC_CHK_STRIDE + t] = 0;
    }
    const int slot = t / CC_KNN_MAX;
    const int level = CC_HIT_LEVEL(h), seq_src = h.seq, seq_tgt = CC_HIT_SEQ_TGT(h, slot);
    int *sc = scores ? scores + ((size_t)q * CC_CHK_STRIDE + t) * CC_NSCORE : nullptr;
    const cc_scan_desc_t *src = db_desc + h.gidx;
    const cc_bci_t *bs = &src->bcis[level][seq_src];
    const cc_bci_t *bt = &tgt->bcis[level][seq_tgt];
    // point tables and their sizes are fetched together (no dependent round trip): CC_BCI_MAXPTS <= 3 * G
    // (moved as raw dwords: a cc_relpt_t is 3 of them)
    unsigned ps[3][3], pt[3][3];
    const unsigned *gs = (const unsigned *)bs->pts, *gt = (const unsigned *)bt->pts;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int i = sl + u * G;
#pragma unroll
      for (int w = 0; w < 3; w++) {
        ps[u][w] = i < CC_BCI_MAXPTS ? gs[i * 3 + w] : 0u;
        pt[u][w] = i < CC_BCI_MAXPTS ? gt[i * 3 + w] : 0u;
      }
    }
    const int nsp = bs->n_pts, ntp = bt->n_pts;
    cc_group_sync(G);
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int i = sl + u * G;
#pragma unroll
      for (int w = 0; w < 3; w++) {
        if (i < nsp) ((unsigned *)L.g.sp)[i * 3 + w] = ps[u][w];
        if (i < ntp) ((unsigned *)L.g.tp)[i * 3 + w] = pt[u][w];
      }
    }
    if (P.dbg_cut == 1) continue;
    cc_group_sync(G);
    // src points are sorted by bit_pos: the partners of a tgt point are the contiguous range [lo, hi)
    int npp_all = 0;
    for (int r0 = 0; r0 < ntp; r0 += G) {
      const int i = r0 + sl;
      int cnt_i = 0;
      if (i < ntp) {
        const int tb = (int)L.g.tp[i].bit_pos;
        int a = 0, b = nsp;
        while (a < b) {  // #(sb < tb - 1)
          const int mid = (a + b) >> 1;
          if ((int)L.g.sp[mid].bit_pos < tb - 1)
            a = mid + 1;
          else
            b = mid;
        }
        const int lo = a;
        b = nsp;
        while (a < b) {  // #(sb <= tb + 1)
          const int mid = (a + b) >> 1;
          if ((int)L.g.sp[mid].bit_pos <= tb + 1)
            a = mid + 1;
          else
            b = mid;
        }
        L.g.lo[i] = (unsigned char)lo;
        L.g.hi[i] = (unsigned char)a;
        cnt_i = a - lo;
      }
      int incl = cnt_i;
      for (int o = 1; o < G; o <<= 1) {
        const int v = __shfl_up(incl, o, G);
        if (sl >= o) incl += v;
      }
      if (i < ntp) L.g.off[i] = (unsigned short)(npp_all + incl - cnt_i);
      npp_all += __shfl(incl, G - 1, G);
    }
    if (P.dbg_cut == 2) continue;
    int flags = 0;
    int npp = npp_all;
    if (npp > PPM) {
      if (!REDO) {  // left to the large instance
        if (sl == 0) {
          pass_ok[(size_t)q * CC_CHK_STRIDE + t] = 2;
          atomicAdd(&redo_cnt[q], 1);
        }
        continue;
      }
      npp = PPM;
      flags |= 1;
    }
    cc_group_sync(G);
    if (npp == 0) {
      if (sc && sl == 0) sc[2] = 1;  // the window search starts from longest_in_range = 1 (contour_mng.h:345)
      continue;
    }
    cc_chkb_gen_pairs(L, ntp, sl);
    if (P.dbg_cut == 3) continue;
    if (P.dbg_cut >= 10) {  // tuning aid: per-query sums reported through cand_aft_check2
      if (sl == 0) atomicAdd(&pass_cnt[q * 4 + 2], P.dbg_cut == 11 ? npp : 1);
      continue;
    }
    cc_chkb_sort(L, npp, ntp, sl);
    if (P.dbg_cut == 4) continue;
    // circular window of width pi/16 (contour_mng.h:344-357): for each start p1 the furthest p2, then the first start
    // that attains the maximum length (what the two-pointer loop records)
    const float angular_range = (float)(3.14159265358979323846 / 16);
    int bestL = 0, bestP = 0x7fffffff;
    for (int p1 = sl; p1 < npp; p1 += G) {
      const float v1 = L.skey[p1];
      int a = p1, b = p1 + npp - 1;  // window [p1, p2], p2 in [p1, p1+npp)
      while (a < b) {                // largest p2 with valid(p2); valid is monotone in p2
        const int mid = (a + b + 1) >> 1;
        const int wr = mid >= npp ? 1 : 0;  // mid < 2 * npp: mid % npp and mid / npp without a division
        const double v = (double)(L.skey[mid - (wr ? npp : 0)] - v1) + 2 * 3.14159265358979323846 * (double)wr;
        if (v > (double)angular_range)
          b = mid - 1;
        else
          a = mid;
      }
      const int len = a - p1 + 1;
      if (len > bestL || (len == bestL && p1 < bestP)) {
        bestL = len;
        bestP = p1;
      }
    }
    for (int o = G >> 1; o > 0; o >>= 1) {
      const int oL = __shfl_xor(bestL, o, G), oP = __shfl_xor(bestP, o, G);
      if (oL > bestL || (oL == bestL && oP < bestP)) {
        bestL = oL;
        bestP = oP;
      }
    }
    if (P.dbg_cut == 5) continue;
    int longest = bestL, beg = bestP;
    if (longest <= 1) {  // the loop starts from longest = 1, beg = 0 and only records strictly longer windows
      longest = 1;
      beg = 0;
    }
    if (sc && sl == 0) sc[2] = longest;
    if (longest < P.lb.i_in_ang_rng) continue;
    if (sl == 0) atomicAdd(&pass_cnt[q * 4 + 2], 1);
    // (3/4) individual similarity of the window pairs + the anchors, in cstl_in order
    int n_in = longest + 1;
    if (n_in > CC_CSTL_MAX) {
      n_in = CC_CSTL_MAX;
      flags |= 1;
    }
    int ncs = 0;
    for (int r0 = 0; r0 < n_in; r0 += G) {
      const int e = r0 + sl;
      int l = 0, s_ = 0, t_ = 0;
      bool sim = false;
      if (e < n_in) {
        if (e < longest && e < n_in - 1) {
          const int pe = beg + e;  // < 2 * npp
          const unsigned w = (unsigned)L.pp[L.sidx[pe >= npp ? pe - npp : pe]];
          l = (int)(signed char)(w & 0xFF);
          s_ = (int)(signed char)((w >> 8) & 0xFF);
          t_ = (int)(signed char)((w >> 16) & 0xFF);
        } else {
          l = level;
          s_ = seq_src;
          t_ = seq_tgt;
        }
        sim = cc_check_sim(src->cont[l][s_], tgt->cont[l][t_], P.sim);
      }
      const unsigned ms = cc_group_ballot(sim, sl);
      if (sim) {
        const int o = ncs + __popc(ms & ((1u << sl) - 1u));
        if (o < CC_CSTL_MAX) {
          L.c.cs[o][0] = (signed char)l;
          L.c.cs[o][1] = (signed char)s_;
          L.c.cs[o][2] = (signed char)t_;
          const cc_contour_t &sc = src->cont[l][s_];
          const cc_contour_t &tc = tgt->cont[l][t_];
          L.m.spm[o][0] = sc.pos_mean[0];
          L.m.spm[o][1] = sc.pos_mean[1];
          L.m.tpm[o][0] = tc.pos_mean[0];
          L.m.tpm[o][1] = tc.pos_mean[1];
        }
      }
      ncs += __popc(ms);
    }
    if (P.dbg_cut == 6) continue;
    if (ncs > CC_CSTL_MAX) {
      ncs = CC_CSTL_MAX;
      flags |= 1;
    }
    if (sc && sl == 0) sc[3] = ncs;
    if (ncs < P.lb.i_indiv_sim) continue;
    cc_group_sync(G);
    // part 2: the "shaft" (contour_mng.h:1173-1184).  The reference scans the (i, j<i) pairs of the first <=10 entries in
    // order, replacing the running (normalised) src vector whenever the candidate is longer than it.  Candidate lengths and
    // the length the running vector would have after the update are computed one pair per lane; the scan itself is a
    // uniform loop over them.
    float shx = 0.f, shy = 0.f, thx = 0.f, thy = 0.f;
    {
      const int lim = ncs < 10 ? ncs : 10;
      const int npair = lim * (lim - 1) / 2;
      for (int pr = sl; pr < npair; pr += G) {
        int i = 1, acc = 0;  // pair index -> (i, jj) in the reference's loop order: i = 1.., jj = 0..i-1
        while (acc + i <= pr) {
          acc += i;
          i++;
        }
        const int jj = pr - acc;
        const float cx = L.m.spm[i][0] - L.m.spm[jj][0], cy = L.m.spm[i][1] - L.m.spm[jj][1];
        float ux, uy;
        const float z = cx * cx + cy * cy;
        if (z > 0.f) {
          const float sq = sqrtf(z);
          ux = cx / sq;
          uy = cy / sq;
        } else {
          ux = cx;
          uy = cy;
        }
        L.c.cn[pr] = cc_norm2f(cx, cy);
        L.c.nn[pr] = cc_norm2f(ux, uy);
      }
      cc_group_sync(G);
      float sn = 0.f;  // norm of the running shaft_src (initially the zero vector)
      int last = -1;
      for (int k = 0; k < npair; k++) {
        if (L.c.cn[k] > sn) {
          sn = L.c.nn[k];
          last = k;
        }
      }
      if (last >= 0) {
        int i = 1, acc = 0;
        while (acc + i <= last) {
          acc += i;
          i++;
        }
        const int jj = last - acc;
        const float cx = L.m.spm[i][0] - L.m.spm[jj][0], cy = L.m.spm[i][1] - L.m.spm[jj][1];
        float z = cx * cx + cy * cy;
        if (z > 0.f) {
          const float sq = sqrtf(z);
          shx = cx / sq;
          shy = cy / sq;
        } else {
          shx = cx;
          shy = cy;
        }
        const float tx = L.m.tpm[i][0] - L.m.tpm[jj][0], ty = L.m.tpm[i][1] - L.m.tpm[jj][1];
        z = tx * tx + ty * ty;
        if (z > 0.f) {
          const float sq = sqrtf(z);
          thx = tx / sq;
          thy = ty / sq;
        } else {
          thx = tx;
          thy = ty;
        }
      }
    }
    if (P.dbg_cut == 7) continue;
    // orientation test per pair (order-independent), then the order-dependent swap-to-back removal (contour_mng.h:1186-1201)
    unsigned long long rmm = 0ull;
    for (int r0 = 0; r0 < ncs; r0 += G) {
      const int e = r0 + sl;
      bool rm = false;
      if (e < ncs) {
        const cc_contour_t &sc = src->cont[L.c.cs[e][0]][L.c.cs[e][1]];
        const cc_contour_t &tc = tgt->cont[L.c.cs[e][0]][L.c.cs[e][2]];
        if (sc.ecc_feat && tc.ecc_feat) {
          const float pi6 = (float)(3.14159265358979323846 / 6);
          const float theta_s = acosf(shx * sc.eig_vecs[2] + shy * sc.eig_vecs[3]);
          const float theta_t = acosf(thx * tc.eig_vecs[2] + thy * tc.eig_vecs[3]);
          const float pms = (float)(3.14159265358979323846 - (double)theta_s);
          rm = fabsf(theta_s - theta_t) > pi6 && fabsf(pms - theta_t) > pi6;
        }
        L.c.keepf[e] = (unsigned char)e;  // position -> original index (identity when nothing is removed)
      }
      rmm |= (unsigned long long)cc_group_ballot(rm, sl) << r0;
    }
    cc_group_sync(G);
    if (rmm) {
      if (sl == 0) {
        int num_sim = ncs;
        for (int i = 0; i < num_sim;) {
          const int o = L.c.keepf[i];
          if ((rmm >> o) & 1ull) {
            L.c.keepf[i] = L.c.keepf[num_sim - 1];  // std::swap(cstl_out[i], cstl_out[num_sim-1]); the tail is erased afterwards
            num_sim--;
            continue;
          }
          i++;
        }
        L.c.misc[0] = num_sim;
      }
      cc_group_sync(G);
      ncs = L.c.misc[0];
    }
    if (P.dbg_cut == 8) continue;
    if (sc && sl == 0) sc[4] = ncs;
    if (ncs < P.lb.i_orie_sim) continue;
    // (4/4) getTFFromConstell: 2-D umeyama without scaling, closed form; sums in list order (uniform loops on LDS)
    if (sl < 8) L.bitsw[sl] = 0ull;
    cc_group_sync(G);
    for (int e = sl; e < ncs; e += G) {
      const int o = L.c.keepf[e];
      const int bit = (L.c.cs[o][0] - 1) * 100 + L.c.cs[o][1] * 10 + L.c.cs[o][2];
      atomicOr(&L.bitsw[bit >> 6], 1ull << (bit & 63));
    }
    const double one_over_n = 1.0 / (double)ncs;
    double smx = 0, smy = 0, dmx = 0, dmy = 0;
    for (int i = 0; i < ncs; i++) {
      const int o = L.c.keepf[i];
      smx += (double)L.m.spm[o][0];
      smy += (double)L.m.spm[o][1];
      dmx += (double)L.m.tpm[o][0];
      dmy += (double)L.m.tpm[o][1];
    }
    smx = smx * one_over_n;
    smy = smy * one_over_n;
    dmx = dmx * one_over_n;
    dmy = dmy * one_over_n;
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    for (int i = 0; i < ncs; i++) {
      const int o = L.c.keepf[i];
      const double ax = (double)L.m.spm[o][0] - smx, ay = (double)L.m.spm[o][1] - smy;
      const double bx = (double)L.m.tpm[o][0] - dmx, by = (double)L.m.tpm[o][1] - dmy;
      s00 += bx * ax;
      s01 += bx * ay;
      s10 += by * ax;
      s11 += by * ay;
    }
    s00 *= one_over_n;
    s01 *= one_over_n;
    s10 *= one_over_n;
    s11 *= one_over_n;
    const double sn2 = s10 - s01, cs_ = s00 + s11;
    const double nrm = sqrt(sn2 * sn2 + cs_ * cs_);
    double r00 = 1, r10 = 0;
    if (nrm > 0) {
      r00 = cs_ / nrm;
      r10 = sn2 / nrm;
    }
    cc_group_sync(G);
    if (sl == 0) {
      atomicAdd(&pass_cnt[q * 4 + 3], 1);
      atomicAdd(&pass_cnt[q * 4 + 0], 1);
      cc_pass_rec *rec = &pass[(size_t)q * CC_CHK_STRIDE + t];
      rec->q = q;
      rec->order = t;
      rec->gidx = h.gidx;
      rec->n_pairs = ncs;
      rec->flags = flags;
      rec->pad = 0;
      rec->tf[0] = dmx - (r00 * smx + (-r10) * smy);
      rec->tf[1] = dmy - (r10 * smx + r00 * smy);
      // the rotation's angle and the entries of Isometry2d::rotate(angle) are filled in by cc_k_check_c, one lane per
      // passing check (here they would be evaluated by a mostly idle wave)
      rec->tf[2] = 0.0;
      rec->cs[0] = r00;
      rec->cs[1] = r10;
      rec->cs[2] = 0.0;
      pass_ok[(size_t)q * CC_CHK_STRIDE + t] = 1;
    }
    if (sl < 7) pass[(size_t)q * CC_CHK_STRIDE + t].bits[sl] = L.bitsw[sl];
  }
}

// Stage C (one lane per check slot): T_pass's angle atan2(R10, R00) and the rotation rebuilt from it, as the reference
// does (getTFFromConstell returns Isometry2d; addProposal and the pose output go through rotate(angle)).
// grid = ceil(nq * CC_CHK_STRIDE / 256), block = 256
__global__ void __launch_bounds__(256)
cc_k_check_c(int n_slots, cc_pass_rec *__restrict__ pass, const unsigned char *__restrict__ pass_ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots || !pass_ok[i]) return;
  cc_pass_rec *rec = &pass[i];
  const double r00 = rec->cs[0], r10 = rec->cs[1];
  const double th = atan2(r10, r00);
  const double c_ = cos(th), s_2 = sin(th);
  rec->tf[2] = th;
  rec->cs[0] = c_;
  rec->cs[1] = s_2;
  rec->cs[2] = atan2(s_2, c_);
}

// ------------------------------------------------------------------------------------------------
// K4b: per query, replay the passing checks in order: CandidatePoseData::addProposal (contour_db.h:286-338) and the
// part of tidyUpCandidates before the correlation (contour_db.h:503-546).  The greedy proposal merge is sequential
// only among checks that name the SAME candidate scan, so the passing checks are threaded into one ordered list per
// candidate and every candidate is replayed by its own lane.  candidates_ keeps first-appearance order.
// ------------------------------------------------------------------------------------------------
#define CC_MAXCAND CC_CHK_STRIDE  // every passing check may name a different scan: no cap to overflow
#define CC_MERGE_BLOCK 128
#define CC_MERGE_PER_T (CC_CHK_STRIDE / CC_MERGE_BLOCK)  // consecutive check slots scanned by one thread

struct cc_gmm_problem {
  int q;          // index into qdesc (tgt)
  int gidx;       // index into db_desc (src)
  double tf[3];   // T_init = (x, y, theta)
};

struct cc_dprop {  // CandidateAnchorProp (contour_db.h:267-274); constell_ kept as a 400-bit set in key order
  unsigned long long bits[7];
  double c, s, tx, ty;  // T_delta_ = [c -s tx; s c ty]
  int vote_cnt;
  float area_perc;
};
struct cc_dcand {  // CandidatePoseData (working state of one lane)
  int gidx, nprops, gmm_idx, pad;
  cc_dprop props[4];
};
struct cc_cand_out {  // what the final-selection kernel needs of a candidate
  int gidx, nprops, gmm_idx, pad;
};
struct cc_qstate {
  int n_cand;  // candidates_.size() before tidyUpCandidates
  int flags;
};
struct cc_merge_lds {
  cc_dcand st[CC_MERGE_BLOCK];             // lane-private candidate state
  int gid[CC_CHK_STRIDE];                  // candidate scan of the i-th passing check
  unsigned short ord[CC_CHK_STRIDE];       // its check slot
  short next[CC_CHK_STRIDE];               // next passing check naming the same scan, -1 = none
  unsigned short firstrec[CC_CHK_STRIDE];  // first passing check of candidate k (candidates in first-appearance order)
  int wsum[CC_MERGE_BLOCK / 64];
  int base;
};

static_assert(CC_CHK_STRIDE % CC_MERGE_BLOCK == 0, "merge scan split");

// grid = nq, block = CC_MERGE_BLOCK
__global__ void __launch_bounds__(CC_MERGE_BLOCK)
cc_k_merge(int nq, cc_score_t lb, int n_row, int n_col, const cc_scan_desc_t *__restrict__ qdesc,
           const cc_scan_desc_t *__restrict__ db_desc, const cc_pass_rec *__restrict__ pass, const unsigned char *__restrict__ pass_ok,
           const int *__restrict__ pass_cnt, cc_cand_out *__restrict__ cands_all, cc_qstate *__restrict__ qstate,
           cc_gmm_problem *__restrict__ probs, int prob_cap, int *__restrict__ n_prob) {
  __shared__ cc_merge_lds L;
  const int q = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (q >= nq) return;
  const unsigned char *okp = pass_ok + (size_t)q * CC_CHK_STRIDE;
  const cc_pass_rec *recs = pass + (size_t)q * CC_CHK_STRIDE;
  // ---- ordered list of the passing checks (slot order = the reference's iteration order)
  int n;
  {
    unsigned okm = 0;
    int cnt = 0;
    for (int u = 0; u < CC_MERGE_PER_T; u++) {
      const int ok = okp[tid * CC_MERGE_PER_T + u] != 0;
      okm |= (unsigned)ok << u;
      cnt += ok;
    }
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) L.wsum[wave] = incl;
    __syncthreads();
    int off = incl - cnt;
    n = 0;
    for (int w = 0; w < CC_MERGE_BLOCK / 64; w++) {
      if (w < wave) off += L.wsum[w];
      n += L.wsum[w];
    }
    for (int u = 0; u < CC_MERGE_PER_T; u++) {
      if ((okm >> u) & 1u) {
        const int t = tid * CC_MERGE_PER_T + u;
        L.ord[off] = (unsigned short)t;
        L.gid[off] = recs[t].gidx;
        L.next[off] = -1;
        off++;
      }
    }
  }
  __syncthreads();
  // ---- thread the checks of one scan together; number the candidates in first-appearance order
  int nc = 0;
  for (int b0 = 0; b0 < n; b0 += CC_MERGE_BLOCK) {
    const int i = b0 + tid;
    bool first = false;
    if (i < n) {
      const int g = L.gid[i];
      int j = i - 1;
      while (j >= 0 && L.gid[j] != g) j--;
      if (j >= 0)
        L.next[j] = (short)i;  // j is the immediately preceding check of this scan: written by exactly one i
      else
        first = true;
    }
    const unsigned long long m = __ballot(first);
    if (lane == 0) L.wsum[wave] = __popcll(m);
    __syncthreads();
    int off = nc + __popcll(m & ((1ull << lane) - 1ull));
    int tot = 0;
    for (int w = 0; w < CC_MERGE_BLOCK / 64; w++) {
      if (w < wave) off += L.wsum[w];
      tot += L.wsum[w];
    }
    if (first) L.firstrec[off] = (unsigned short)i;
    nc += tot;
    __syncthreads();
  }
  if (tid == 0) {
    cc_qstate st;
    st.n_cand = nc;
    st.flags = 0;
    qstate[q] = st;
  }
  // ---- one lane per candidate
  const cc_scan_desc_t *tl = qdesc + q;
  cc_dcand *c = &L.st[tid];
  for (int k = tid; k < nc; k += CC_MERGE_BLOCK) {
    int i = L.firstrec[k];
    c->gidx = L.gid[i];
    c->nprops = 0;
    for (; i >= 0; i = L.next[i]) {
      const cc_pass_rec *rec = &recs[L.ord[i]];
      const int np = rec->n_pairs;
      const double ptx = rec->tf[0], pty = rec->tf[1];
      const double pc = rec->cs[0], ps = rec->cs[1];
      const int nprops = c->nprops;
      // CandidatePoseData::addProposal: first proposal within 2.0 (pixels) and 0.3 rad
      int hit = -1;
      for (int pi = 0; pi < nprops && hit < 0; pi++) {
        const cc_dprop *p = &c->props[pi];
        const double i00 = pc, i01 = ps, i10 = -ps, i11 = pc;
        const double itx = -(i00 * ptx + i01 * pty), ity = -(i10 * ptx + i11 * pty);
        const double d00 = i00 * p->c + i01 * p->s, d10 = i10 * p->c + i11 * p->s;
        const double dtx = i00 * p->tx + i01 * p->ty + itx, dty = i10 * p->tx + i11 * p->ty + ity;
        if (sqrt(dtx * dtx + dty * dty) < 2.0 && fabs(atan2(d10, d00)) < 0.3) hit = pi;
      }
      if (hit >= 0) {
        cc_dprop *p = &c->props[hit];
        for (int w = 0; w < 7; w++) p->bits[w] |= rec->bits[w];
        p->vote_cnt += np;
        const int w1 = p->vote_cnt, w2 = np;
        const double bx = (p->tx * w1 + ptx * w2) / (w1 + w2), by = (p->ty * w1 + pty * w2) / (w1 + w2);
        const double ang1 = atan2(p->s, p->c), ang2 = rec->cs[2];
        double diff = ang2 - ang1;
        if (diff < 0) diff += 2 * 3.14159265358979323846;
        if (diff > 3.14159265358979323846) diff -= 2 * 3.14159265358979323846;
        const double ang_bl = diff * w2 / (w1 + w2) + ang1;
        p->c = cos(ang_bl);
        p->s = sin(ang_bl);
        p->tx = bx;
        p->ty = by;
      } else if (nprops <= 3) {
        cc_dprop *p = &c->props[nprops];
        for (int w = 0; w < 7; w++) p->bits[w] = rec->bits[w];
        p->c = pc;
        p->s = ps;
        p->tx = ptx;
        p->ty = pty;
        p->vote_cnt = np;
        p->area_perc = 0.f;
        c->nprops = nprops + 1;
      }
    }
    // tidyUpCandidates before the correlation (contour_db.h:503-546)
    const cc_scan_desc_t *sl = db_desc + c->gidx;
    int idx_sel = 0;
    for (int pi = 0; pi < c->nprops; pi++) {
      float lev_perc[CC_NLEV] = {0, 0, 0, 0, 0, 0};
      for (int w = 0; w < 7; w++) {
        unsigned long long m = c->props[pi].bits[w];
        while (m) {
          const int b = w * 64 + (__ffsll((unsigned long long)m) - 1);
          m &= m - 1;
          const int l = b / 100 + 1, s_ = (b % 100) / 10, t_ = b % 10;
          const float psrc = (float)sl->cont[l][s_].cell_cnt * 1.0f / (float)sl->layer_cell_cnt[l];
          const float ptgt = (float)tl->cont[l][t_].cell_cnt * 1.0f / (float)tl->layer_cell_cnt[l];
          lev_perc[l] += 0.5f * (psrc + ptgt);
        }
      }
      float perc = 0.f;
      perc += 0.3f * lev_perc[1];
      perc += 0.3f * lev_perc[2];
      perc += 0.3f * lev_perc[3];
      perc += 0.1f * lev_perc[4];
      c->props[pi].area_perc = perc;
      if (c->props[pi].vote_cnt > c->props[idx_sel].vote_cnt) idx_sel = pi;
    }
    const cc_dprop *p0 = &c->props[idx_sel];  // std::swap(anch_props_[0], anch_props_[idx_sel]): only [0] is used afterwards
    int gi = -1;
    if (!(p0->area_perc < lb.area_perc)) {
      // getEstSensTF: T_so^-1 * T_delta * T_so with T_so = translate(n_row/2 - 0.5, n_col/2 - 0.5)
      const double ox = n_row / 2 - 0.5, oy = n_col / 2 - 0.5;
      const double mx = p0->c * ox + (-p0->s) * oy + p0->tx, my = p0->s * ox + p0->c * oy + p0->ty;
      const double ex = 1.0 * mx + 0.0 * my + (-(1.0 * ox + 0.0 * oy)), ey = 0.0 * mx + 1.0 * my + (-(0.0 * ox + 1.0 * oy));
      const double neg = -sqrt(ex * ex + ey * ey);
      if (!(neg < (double)lb.neg_est_dist)) {
        const int pi = atomicAdd(n_prob, 1);
        if (pi < prob_cap) {
          cc_gmm_problem pb;
          pb.q = q;
          pb.gidx = c->gidx;
          pb.tf[0] = p0->tx;
          pb.tf[1] = p0->ty;
          pb.tf[2] = atan2(p0->s, p0->c);
          probs[pi] = pb;
          gi = pi;
        }
      }
    }
    cc_cand_out o;
    o.gidx = c->gidx;
    o.nprops = c->nprops;
    o.gmm_idx = gi;
    o.pad = 0;
    cands_all[(size_t)q * CC_MAXCAND + k] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// K5: GMM-L2 correlation (init) + <= 10 L-BFGS iterations, one wave per (query, candidate) problem.
// ------------------------------------------------------------------------------------------------
// LDS caps of the GMM kernel: the common instance keeps 8 workgroups per CU resident; problems that
// overflow it (flags bit0/bit1) are re-run by the host with the large instance.
#define CC_GMM_ECAP_S 24
#define CC_GMM_PCAP_S 512
#define CC_GMM_PPW_S 4     // problems per wave in the common instance (16 lanes each)
#define CC_GMM_ECAP_L 128  // ellipses per (side, level) held in LDS
#define CC_GMM_PCAP_L 4096 // selected (src,tgt) ellipse pairs

struct cc_gmm_result {
  double corr_init;
  double corr_opt;
  double tf_opt[3];
  int optimized;   // 0: init correlation below the bar (no refinement)
  int iterations;
  int termination;
  int flags;       // bit0: ellipse cap hit, bit1: pair cap hit, bit2: contour table truncated (CC_MAXC)
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

// grid = n scans, block = 64
__global__ void __launch_bounds__(64)
cc_k_gmm_prep(const cc_scan_desc_t *__restrict__ desc, int n, cc_gmm_feat *__restrict__ feat) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n) return;
  const cc_scan_desc_t *d = desc + blockIdx.x;
  cc_gmm_feat *F = feat + blockIdx.x;
  __shared__ cc_ell E[CC_GMM_ECAP_L];
  int flags = 0;
  double acc = 0;
  for (int li = 0; li < CC_GMM_LEVELS; li++) {
    const int lev = li + 1;  // GMMOptConfig::levels_ = {1,2,3,4}
    const int full = d->layer_cell_cnt[lev];
    const int ncont = d->n_cont[lev], nst = d->n_stored[lev];
    // contours in sorted order until >= 95 % of the level's cells: contour j is used iff the cells before it are < 95 %
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
    __syncthreads();
    for (int j = lane; j < n_use; j += 64) {
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
    for (int idx = lane; idx < n_use * n_use; idx += 64) {
      const int i = idx / n_use, j = idx - i * n_use;
      const cc_ell a = E[i], b = E[j];
      const double n00 = 2.0 * ((double)a.c00 + (double)b.c00), n01 = 2.0 * ((double)a.c01 + (double)b.c01);
      const double n10 = 2.0 * ((double)a.c10 + (double)b.c10), n11 = 2.0 * ((double)a.c11 + (double)b.c11);
      const double mx = (double)a.mx - (double)b.mx, my = (double)a.my - (double)b.my;
      const double det = n00 * n11 - n10 * n01, invdet = 1.0 / det;
      const double i00 = n11 * invdet, i10 = -n10 * invdet, i01 = -n01 * invdet, i11 = n00 * invdet;
      const double h0 = -0.5 * mx, h1 = -0.5 * my;
      const double r0 = h0 * i00 + h1 * i10, r1 = h0 * i01 + h1 * i11;
      acc += (double)a.w * (double)b.w / sqrt(det) * exp(r0 * mx + r1 * my);
    }
    if (lane == 0) F->n_ell[li] = n_use;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) {
    F->ac = acc;
    F->flags = flags;
  }
}

// LDS view of one problem (pointers into the dynamic LDS block, sized by the kernel instance).  A wave handles 64/G
// problems at once, G lanes each (G = 16 for the common instance, 64 for the large-cap instance); every cross-lane
// operation below is G-wide, so problems in the same wave may diverge freely.
struct cc_gmm_lds {
  cc_ell *ell;      // [2][CC_GMM_LEVELS][ecap]   side 0 = src, 1 = tgt
  double *hist;     // L-BFGS history: dx[10][3] | dg[10][3] | dx.dg[10] | alpha[10]  (group-uniform values)
  unsigned *pairs;  // [pcap]  (li << 28) | (si << 14) | ti
  int *n_ell;       // [2][CC_GMM_LEVELS]
  int *n_pairs;
  int *flags;
  int ecap, pcap;
  int G, sl;        // lanes per problem, this lane's index within its group
  __device__ __forceinline__ const cc_ell &E(int side, int li, int i) const { return ell[(side * CC_GMM_LEVELS + li) * ecap + i]; }
  __device__ __forceinline__ cc_ell &E(int side, int li, int i) { return ell[(side * CC_GMM_LEVELS + li) * ecap + i]; }
};
#define CC_GMM_LDS_BYTES(ecap, pcap) (2 * CC_GMM_LEVELS * (ecap) * sizeof(cc_ell) + 80 * 8 + (pcap) * 4 + 64)

__device__ __forceinline__ double cc_group_sum_d(double v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}
// cost (+ gradient) of GMMPair::operator() at p, summed over the selected pairs by the lanes of the problem's group
// The reference differentiates GMMPair::operator() with ceres::Jet<double, 3> (correlation.h:123-160).  Every quantity
// built from the rotation alone has d/dx = d/dy = 0 structurally, so only (value, d/dtheta) is carried for those
// (cc_j1); the dropped Jet terms are exact zeros (finite * 0, + 0), which leaves every kept component bit-identical to
// the full Jet arithmetic.  Operation order follows ceres/jet.h: f*g = (fa*ga, fa*gv + fv*ga),
// f/g = (fa/ga =: q via 1/ga, (fv - q*gv)/ga), sqrt f = (r, fv / (2r)), exp f = (e, e*fv).
struct cc_j1 {
  double a, t;
};
__device__ __forceinline__ cc_j1 j1_mulc(const cc_j1 &f, double c) { return cc_j1{f.a * c, f.t * c}; }
__device__ __forceinline__ cc_j1 j1_mul(const cc_j1 &f, const cc_j1 &g) { return cc_j1{f.a * g.a, f.a * g.t + f.t * g.a}; }
__device__ __forceinline__ cc_j1 j1_add(const cc_j1 &f, const cc_j1 &g) { return cc_j1{f.a + g.a, f.t + g.t}; }
__device__ __forceinline__ cc_j1 j1_sub(const cc_j1 &f, const cc_j1 &g) { return cc_j1{f.a - g.a, f.t - g.t}; }
__device__ __forceinline__ cc_j1 j1_neg(const cc_j1 &f) { return cc_j1{-f.a, -f.t}; }

__device__ void cc_gmm_eval(const cc_gmm_lds *S, const double p[3], bool want_grad, double *cost, double grad[3]) {
  const int G = S->G, sl = S->sl;
  const double px = p[0], py = p[1];
  const double ct = cos(p[2]), st = sin(p[2]);
  const cc_j1 R00{ct, -st}, R01{-st, -ct}, R10{st, ct}, R11{ct, -st};
  double acc_a = 0.0, acc_x = 0.0, acc_y = 0.0, acc_t = 0.0;
  const int np = *S->n_pairs;
  for (int i = sl; i < np; i += G) {
    const unsigned pr = S->pairs[i];
    const int li = pr >> 28, si = (pr >> 14) & 0x3FFF, ti = pr & 0x3FFF;
    const cc_ell es = S->E(0, li, si), et = S->E(1, li, ti);
    // new_cov = scale_ * (R cov_s R^T + cov_t), scale_ = 2
    const cc_j1 RC00 = j1_add(j1_mulc(R00, (double)es.c00), j1_mulc(R01, (double)es.c10));
    const cc_j1 RC01 = j1_add(j1_mulc(R00, (double)es.c01), j1_mulc(R01, (double)es.c11));
    const cc_j1 RC10 = j1_add(j1_mulc(R10, (double)es.c00), j1_mulc(R11, (double)es.c10));
    const cc_j1 RC11 = j1_add(j1_mulc(R10, (double)es.c01), j1_mulc(R11, (double)es.c11));
    cc_j1 n00 = j1_add(j1_mul(RC00, R00), j1_mul(RC01, R01));
    cc_j1 n01 = j1_add(j1_mul(RC00, R10), j1_mul(RC01, R11));
    cc_j1 n10 = j1_add(j1_mul(RC10, R00), j1_mul(RC11, R01));
    cc_j1 n11 = j1_add(j1_mul(RC10, R10), j1_mul(RC11, R11));
    n00 = cc_j1{2.0 * (n00.a + (double)et.c00), 2.0 * n00.t};
    n01 = cc_j1{2.0 * (n01.a + (double)et.c01), 2.0 * n01.t};
    n10 = cc_j1{2.0 * (n10.a + (double)et.c10), 2.0 * n10.t};
    n11 = cc_j1{2.0 * (n11.a + (double)et.c11), 2.0 * n11.t};
    // new_mean = R mean_s + t - mean_t: (value, d/dx, d/dy, d/dtheta) = (m0a, 1, 0, m0t), (m1a, 0, 1, m1t)
    const double m0a = R00.a * (double)es.mx + R01.a * (double)es.my + px - (double)et.mx;
    const double m0t = R00.t * (double)es.mx + R01.t * (double)es.my;
    const double m1a = R10.a * (double)es.mx + R11.a * (double)es.my + py - (double)et.my;
    const double m1t = R10.t * (double)es.mx + R11.t * (double)es.my;
    const cc_j1 det = j1_sub(j1_mul(n00, n11), j1_mul(n10, n01));
    const double dgi = 1.0 / det.a, dfg = 1.0 * dgi;
    const cc_j1 invdet{dfg, (0.0 - dfg * det.t) * dgi};
    const cc_j1 i00 = j1_mul(n11, invdet), i01 = j1_mul(j1_neg(n01), invdet);
    const cc_j1 i10 = j1_mul(j1_neg(n10), invdet), i11 = j1_mul(n00, invdet);
    const double h0a = -0.5 * m0a, h0t = -0.5 * m0t, h1a = -0.5 * m1a, h1t = -0.5 * m1t;  // d/dx h0 = d/dy h1 = -0.5
    const double r0a = h0a * i00.a + h1a * i10.a, r0x = -0.5 * i00.a, r0y = -0.5 * i10.a;
    const double r0t = (h0a * i00.t + h0t * i00.a) + (h1a * i10.t + h1t * i10.a);
    const double r1a = h0a * i01.a + h1a * i11.a, r1x = -0.5 * i01.a, r1y = -0.5 * i11.a;
    const double r1t = (h0a * i01.t + h0t * i01.a) + (h1a * i11.t + h1t * i11.a);
    const double qa = r0a * m0a + r1a * m1a;
    const double qx = (r0a + r0x * m0a) + r1x * m1a;
    const double qy = r0y * m0a + (r1a + r1y * m1a);
    const double qt = (r0a * m0t + r0t * m0a) + (r1a * m1t + r1t * m1a);
    const double sq = sqrt(det.a), sqh = 1.0 / (2.0 * sq), sqt = sqh * det.t;
    const double B = (-(double)et.w) * (double)es.w * 1.0;
    const double cgi = 1.0 / sq, Ca = B * cgi, Ct = (0.0 - Ca * sqt) * cgi;
    const double e = exp(qa);
    acc_a = acc_a + Ca * e;
    acc_x = acc_x + Ca * (e * qx);
    acc_y = acc_y + Ca * (e * qy);
    acc_t = acc_t + (Ca * (e * qt) + Ct * e);
  }
  *cost = cc_group_sum_d(acc_a, G);
  if (want_grad) {
    grad[0] = cc_group_sum_d(acc_x, G);
    grad[1] = cc_group_sum_d(acc_y, G);
    grad[2] = cc_group_sum_d(acc_t, G);
  }
}

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
  cc_solve_fullpiv(A, b, nc, sol);
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

__device__ __forceinline__ void cc_ls_eval(const cc_gmm_lds *S, const double pos[3], const double dir[3], double x, cc_fs *o) {
  o->x = x;
  double vx[3];
  for (int i = 0; i < 3; i++) vx[i] = pos[i] + x * dir[i];
  cc_gmm_eval(S, vx, true, &o->value, o->vg);
  o->value_ok = isfinite(o->value);
  o->grad_ok = o->value_ok && isfinite(o->vg[0]) && isfinite(o->vg[1]) && isfinite(o->vg[2]);
  o->gradient = dir[0] * o->vg[0] + dir[1] * o->vg[1] + dir[2] * o->vg[2];
}

// WolfeLineSearch::DoSearch (bracketing + zoom).  Returns success; *opt is the accepted sample.
__device__ bool cc_wolfe(const cc_gmm_lds *S, const double pos[3], const double dir[3], double step0, double cost0,
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
  cc_ls_eval(S, pos, dir, step0, &cur);
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
    cc_ls_eval(S, pos, dir, step, &cur);
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
      cc_ls_eval(S, pos, dir, step, &sol);
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

// grid = any (grid-stride over the device-side problem count), block = 64, `ppw` problems per wave (G = 64/ppw lanes
// each), dynamic LDS = ppw * CC_GMM_LDS_BYTES(ecap, pcap).
// redo_only: process only problems whose previous result overflowed an LDS cap (flags & 3) -- the large-cap instance.
// Two passes over the problems (the reference refines only the first max_fine_opt_ candidates of a query,
// contour_db.h:604-648, but needs the initial correlation of all of them, :548-592):
//   sel_list == nullptr : every problem, initial correlation only (tryProblem)
//   sel_list != nullptr : the problems cc_k_select listed (n_prob_p = their count), initial correlation + L-BFGS
__global__ void __launch_bounds__(64)
cc_k_gmm(const cc_gmm_problem *__restrict__ probs, const int *__restrict__ n_prob_p, int prob_cap, int redo_only,
         const cc_gmm_feat *__restrict__ qfeat, const cc_gmm_feat *__restrict__ db_feat, float corr_lb, int ecap, int pcap,
         int ppw, const int *__restrict__ sel_list, cc_gmm_result *__restrict__ results) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int lane = threadIdx.x;
  const int G = 64 / ppw, sub = lane / G, sl = lane - sub * G;
  char *base_lds = smem + (size_t)sub * CC_GMM_LDS_BYTES(ecap, pcap);
  cc_gmm_lds Sv;
  Sv.ell = (cc_ell *)base_lds;
  Sv.hist = (double *)(base_lds + 2 * CC_GMM_LEVELS * (size_t)ecap * sizeof(cc_ell));
  Sv.pairs = (unsigned *)(Sv.hist + 80);
  Sv.n_ell = (int *)(Sv.pairs + pcap);
  Sv.n_pairs = Sv.n_ell + 2 * CC_GMM_LEVELS;
  Sv.flags = Sv.n_pairs + 1;
  Sv.ecap = ecap;
  Sv.pcap = pcap;
  Sv.G = G;
  Sv.sl = sl;
  cc_gmm_lds *S = &Sv;
  int n_prob = *n_prob_p;
  if (n_prob > prob_cap) n_prob = prob_cap;
  for (int pbase = blockIdx.x * ppw; pbase < n_prob; pbase += gridDim.x * ppw) {
  if (pbase + sub >= n_prob) continue;
  const int pidx = sel_list ? sel_list[pbase + sub] : pbase + sub;
  if ((redo_only & 1) && !(results[pidx].flags & 3)) continue;
  const int dbg = redo_only >> 8;  // tuning aid (env CC_GMM_CUT): stop after a phase, results are then meaningless
  const cc_gmm_problem pb = probs[pidx];
  const cc_gmm_feat *fsrc = db_feat + pb.gidx;
  const cc_gmm_feat *ftgt = qfeat + pb.q;
  cc_group_sync(G);
  if (sl == 0) {
    *S->n_pairs = 0;
    *S->flags = 0;
  }
  cc_group_sync(G);
  // ---- ellipses of both scans (GMMPair ctor, correlation.h:49-82; selected per scan by cc_k_gmm_prep)
  if (sl < 2 * CC_GMM_LEVELS) {
    const int side = sl / CC_GMM_LEVELS, li = sl % CC_GMM_LEVELS;
    const cc_gmm_feat *f = side == 0 ? fsrc : ftgt;
    int n = f->n_ell[li];
    if (n > ecap) {
      n = ecap;
      atomicOr((unsigned *)S->flags, 1u);
    }
    S->n_ell[side * CC_GMM_LEVELS + li] = n;
    if (li == 0 && (f->flags & 5)) atomicOr((unsigned *)S->flags, (unsigned)(f->flags & 5));
  }
  cc_group_sync(G);
  for (int side = 0; side < 2; side++) {
    const cc_gmm_feat *f = side == 0 ? fsrc : ftgt;
    for (int li = 0; li < CC_GMM_LEVELS; li++) {
      const int n = S->n_ell[side * CC_GMM_LEVELS + li];
      for (int i = sl; i < n; i += G) S->E(side, li, i) = f->ell[li][i];
    }
  }
  cc_group_sync(G);
  // ---- pair pre-selection (correlation.h:85-96), ordered compaction by a G-wide prefix sum
  const double ct0 = cos(pb.tf[2]), st0 = sin(pb.tf[2]);
  int np = 0;
  for (int li = 0; li < CC_GMM_LEVELS; li++) {
    const int ns = S->n_ell[li], ntg = S->n_ell[CC_GMM_LEVELS + li];
    const int tot = ns * ntg;
    for (int base = 0; base < tot; base += G) {
      const int idx = base + sl;
      int sel = 0;
      int si = 0, ti = 0;
      if (idx < tot) {
        si = idx / ntg;
        ti = idx - si * ntg;
        const cc_ell &es = S->E(0, li, si), &et = S->E(1, li, ti);
        const double dx = (ct0 * (double)es.mx + (-st0) * (double)es.my + pb.tf[0]) - (double)et.mx;
        const double dy = (st0 * (double)es.mx + ct0 * (double)es.my + pb.tf[1]) - (double)et.my;
        sel = sqrt(dx * dx + dy * dy) < 3.0 * (double)(es.maj + et.maj) ? 1 : 0;
      }
      int incl = sel;
      for (int o = 1; o < G; o <<= 1) {
        const int v = __shfl_up(incl, o, G);
        if (sl >= o) incl += v;
      }
      const int off = np + incl - sel;
      if (sel) {
        if (off < pcap)
          S->pairs[off] = ((unsigned)li << 28) | ((unsigned)si << 14) | (unsigned)ti;
        else
          atomicOr((unsigned *)S->flags, 2u);
      }
      np += __shfl(incl, G - 1, G);
    }
  }
  if (np > pcap) np = pcap;
  if (sl == 0) *S->n_pairs = np;
  if (dbg == 1) {
    if (sl == 0) {
      cc_gmm_result Z = {};
      results[pidx] = Z;
    }
    continue;
  }
  // ---- auto-correlation (correlation.h:102-119): per-scan terms
  const double ac[2] = {fsrc->ac, ftgt->ac};
  cc_group_sync(G);
  if (dbg == 2) {
    if (sl == 0) {
      cc_gmm_result Z = {};
      results[pidx] = Z;
    }
    continue;
  }
  // ---- initial correlation (tryProblem, correlation.h:196-202)
  double x[3] = {pb.tf[0], pb.tf[1], pb.tf[2]};
  double cost, g[3];
  cc_gmm_eval(S, x, true, &cost, g);
  const double denom = sqrt(ac[0] * ac[1]);
  cc_gmm_result R;
  R.corr_init = -cost / denom;
  R.corr_opt = R.corr_init;
  R.tf_opt[0] = x[0];
  R.tf_opt[1] = x[1];
  R.tf_opt[2] = x[2];
  R.optimized = 0;
  R.iterations = 0;
  R.termination = 0;
  if (dbg == 3) {
    if (sl == 0) {
      cc_gmm_result Z = {};
      results[pidx] = Z;
    }
    continue;
  }
  if (sel_list && !((float)R.corr_init < corr_lb)) {
    // ---- calcCorrelation (correlation.h:206-238): LineSearchMinimizer, LBFGS rank 20, Wolfe/cubic, <= 10 iterations
    R.optimized = 1;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double cur_cost = cost, cur_g[3] = {g[0], g[1], g[2]};
    double prev_cost = 0, prev_g[3] = {0, 0, 0}, prev_dir[3] = {0, 0, 0}, prev_step = 0;
    double *dxh = S->hist, *dgh = S->hist + 30, *dxdg = S->hist + 60, *alpha = S->hist + 70;  // [k * 3 + i]
    int ncorr = 0;
    int restarts = 0;
    int term = 0, iter = 0;
    double gmax = fmax(fabs(cur_g[0]), fmax(fabs(cur_g[1]), fabs(cur_g[2])));
    double final_cost = cur_cost;
    if (gmax <= gradient_tolerance) {
      term = 1;
    } else {
      while (true) {
        if (iter >= (dbg >= 4 ? dbg - 3 : 10)) {
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
        if (!cc_wolfe(S, x, dir, step0, cur_cost, dderiv, cur_g, &opt)) {
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
  cc_group_sync(G);
  R.flags = *S->flags;
  if (sl == 0) results[pidx] = R;
  }  // problem loop
}

// ------------------------------------------------------------------------------------------------
// K5s: which candidates of a query get refined -- the first max_fine_opt_ of candidates_ after tidyUpCandidates'
// compaction and fineOptimize's sort on the still-all-zero correlation_ (the same replay as in K6; contour_db.h:560-616).
// Their GMM problems are appended to sel_list.  One wave per query.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
cc_k_select(int nq, float corr_lb, int max_fine_opt, const cc_cand_out *__restrict__ cands_all, const cc_qstate *__restrict__ qstate,
            const cc_gmm_result *__restrict__ gres, int *__restrict__ sel_list, int *__restrict__ n_sel) {
  __shared__ unsigned short idx[CC_MAXCAND];
  __shared__ unsigned char has[CC_MAXCAND];
  __shared__ int gm[CC_MAXCAND];
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
  if (lane != 0) return;
  int p1 = 0, p2 = nc - 1;
  while (p1 <= p2) {
    if (!has[idx[p1]] && has[idx[p2]]) {
      const unsigned short t = idx[p1];
      idx[p1] = idx[p2];
      idx[p2] = t;
      p1++;
      p2--;
    } else {
      if (has[idx[p1]]) p1++;
      if (!has[idx[p2]]) p2--;
    }
  }
  const int n = p2 + 1;
  if (n <= 0) return;
  ccsort::std_sort(idx, n, [](unsigned short, unsigned short) { return false; });
  const int pre = max_fine_opt < n ? max_fine_opt : n;
  const int base = atomicAdd(n_sel, pre);
  for (int i = 0; i < pre; i++) sel_list[base + i] = gm[idx[i]];
}

// ------------------------------------------------------------------------------------------------
// K6: per query, the rest of tidyUpCandidates (correlation bar + order-changing compaction, contour_db.h:560-592) and
// fineOptimize (contour_db.h:604-648): std::sort on the still-all-zero correlation_ (replayed), take the first
// max_fine_opt_, adopt their refined score/pose, re-sort those, return the best.  One lane per query.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
cc_k_final(int nq, float corr_lb, int max_fine_opt, const cc_cand_out *__restrict__ cands_all, const cc_qstate *__restrict__ qstate,
           const cc_gmm_result *__restrict__ gres, const int *__restrict__ pass_cnt, const int *__restrict__ hit_cnt,
           cc_query_result_t *__restrict__ out) {
  // one wave per query: lanes fetch the per-candidate inputs in parallel, lane 0 replays the order-dependent part on LDS
  __shared__ unsigned short idx[CC_MAXCAND];
  __shared__ unsigned char has[CC_MAXCAND];
  __shared__ float corr_o[CC_MAXCAND];
  __shared__ int gm[CC_MAXCAND];
  __shared__ int s_tot;
  const int q = blockIdx.x, lane = threadIdx.x;
  if (q >= nq) return;
  const cc_cand_out *cands = cands_all + (size_t)q * CC_MAXCAND;
  const int nc = qstate[q].n_cand;
  for (int k = lane; k < nc; k += 64) {
    const int g = cands[k].gmm_idx;
    idx[k] = (unsigned short)k;
    gm[k] = g;
    bool h = false;
    float co = 0.f;
    if (g >= 0) {
      h = !((float)gres[g].corr_init < corr_lb);
      co = (float)gres[g].corr_opt;
    }
    has[k] = h ? 1 : 0;
    corr_o[k] = co;
  }
  int tot = 0;
  for (int s2 = lane; s2 < CC_NQLEV * CC_NPIV; s2 += 64) tot += hit_cnt[q * CC_NQLEV * CC_NPIV + s2];
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  if (lane == 0) s_tot = tot;
  __syncthreads();
  if (lane != 0) return;
  cc_query_result_t r;
  r.n_res = 0;
  r.cand_gidx = -1;
  r.correlation = 0;
  r.tf[0] = r.tf[1] = r.tf[2] = 0;
  r.cand_aft_check1 = pass_cnt[q * 4 + 1];
  r.cand_aft_check2 = pass_cnt[q * 4 + 2];
  r.cand_aft_check3 = pass_cnt[q * 4 + 3];
  r.n_cand_pose = nc;
  r.n_knn_hits = s_tot;
  // two-pointer compaction of candidates_ (has = corr_est_ != nullptr), contour_db.h:580-592
  int p1 = 0, p2 = nc - 1;
  while (p1 <= p2) {
    if (!has[idx[p1]] && has[idx[p2]]) {
      const unsigned short t = idx[p1];
      idx[p1] = idx[p2];
      idx[p2] = t;
      p1++;
      p2--;
    } else {
      if (has[idx[p1]]) p1++;
      if (!has[idx[p2]]) p2--;
    }
  }
  const int n = p2 + 1;
  r.n_cand_tidy = n;
  if (n > 0) {
    // first std::sort: every anch_props_[0].correlation_ is still 0 -> comparator is always false
    ccsort::std_sort(idx, n, [](unsigned short, unsigned short) { return false; });
    const int pre = max_fine_opt < n ? max_fine_opt : n;
    // candidates beyond `pre` keep correlation_ = 0
    ccsort::std_sort(idx, pre, [&](unsigned short a, unsigned short b) { return corr_o[a] > corr_o[b]; });
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
