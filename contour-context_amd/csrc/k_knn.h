// K3 -- retrieval: maintenance of the per-layer sorted key view and the k-nearest-key search with the reference's
// visibility rules.  Replaces LayerDB::layerKNNSearch / TreeBucket::knnSearch / nanoflann kNN (src/cont2/contour_db.cpp:319-403).
#pragma once
#include "cc_dev.h"

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
cc_k_knn(cc_knn_params P, const cc_hot_desc_t *__restrict__ qhot, const cc_query_meta *__restrict__ qmeta,
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
  const float *qk = &qhot[q].keys[level - 1][seq][0];
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
  bool open[2] = {right < n, left >= 0};  // [0]: upwards, [1]: downwards
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

