// K3 -- retrieval: maintenance of the per-layer sorted key view and the k-nearest-key search with the reference's
// visibility rules.  Replaces LayerDB::layerKNNSearch / TreeBucket::knnSearch / nanoflann kNN (src/cont2/contour_db.cpp:319-403).
#pragma once
#include "cc_dev.h"

// ------------------------------------------------------------------------------------------------
// K3: k nearest retrieval keys with the reference's visibility rules.
//
// The reference keeps one kd-tree per bucket (nanoflann, src/cont2/contour_db.cpp:319-403); what it returns per anchor
// key is the nnk nearest keys (squared L2 over 10 dims, ties by insertion) among the visible ones, within dist_ub.
// Here each layer keeps its keys SORTED BY THE FIRST DIMENSION (the widest one, sqrt(eig_large * cell_cnt): std 37
// against 13 for the second and ~6 for the rest on Velodyne scans, so further index dimensions would prune next to
// nothing).  One wave per search:
//   * a bucket is an interval of the first dimension, i.e. an INDEX RANGE of the sorted layer: the ranges of the buckets
//     layerKNNSearch visits ({0..mid} and {2 mid + 1..5}) and the anchor's own position are found with five
//     simultaneous 9-ary searches (8 lanes each); keys of the other buckets are never touched;
//   * the search walks outwards from the anchor in both directions over those ranges, 64 keys per direction and step (the
//     next step's keys are requested as soon as the current ones are scored); a direction stops when
//     (key[0] - q[0])^2 alone exceeds the radius -- the 1-D form of the kd-tree's pruning rule;
//   * the radius starts at dist_ub and drops to the nnk-th best distance whenever 2 nnk candidates are pending; the
//     candidates are ordered by (distance, key id) with a bitonic network held in registers (<= 4 entries per lane,
//     cross-lane exchanges through ds_bpermute, no barriers).
// The result is the exact set and order of the reference: any radius between the final nnk-th distance and dist_ub is
// an admissible filter, only the survivors' order (distance, then key id) matters.
// ------------------------------------------------------------------------------------------------
// LDS candidate buffer per search (entries of 8 B): at most 2 * nnk - 1 kept candidates + one 64-key step per direction
// are pending when the buffer is reduced (255 at nnk = CC_KNN_MAX = 64).
#define CC_KNN_CAP 256
static_assert(2 * CC_KNN_MAX - 1 + 128 <= CC_KNN_CAP && CC_KNN_MAX <= 64, "cc_k_knn: pending candidates must fit the LDS buffer");

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

// hand-off of LDS data between the lanes of ONE wave (the searches run one wave per workgroup): LDS operations of a wave
// execute in issue order, only the compiler has to be kept from reordering them.  The CPU harness runs the lanes as OS
// threads and needs a real rendezvous.
__device__ __forceinline__ void cc_wave_sync() {
#ifndef CC_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#else
  __syncthreads();
#endif
}

// Ascending bitonic sort of 64 * R keys held R per lane: position p lives in v[p / 64] of lane p % 64.
template <int R>
__device__ __forceinline__ void cc_wave_bitonic_u64(unsigned long long (&v)[R], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {  // partner in the same lane
#pragma unroll
        for (int a = 0; a < R; a++) {
          const int b = a ^ (j >> 6);
          if (b > a) {
            const bool up = ((a * 64) & k) == 0;
            const unsigned long long x = v[a], y = v[b];
            if ((x > y) == up) {
              v[a] = y;
              v[b] = x;
            }
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < R; a++) {
          const unsigned long long o = __shfl_xor(v[a], j);
          const bool up = (((a * 64 + lane) & k) == 0);
          const bool lower = (lane & j) == 0;
          const unsigned long long mn = o < v[a] ? o : v[a], mx = o < v[a] ? v[a] : o;
          v[a] = (lower == up) ? mn : mx;
        }
      }
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

// Keep the best nnk of the cnt pending candidates (sorted, at buf[0..nnk)); returns the nnk-th best distance.
template <int R>
__device__ __forceinline__ float cc_knn_reduce(unsigned long long *buf, int cnt, int nnk, int lane, unsigned long long &first) {
  unsigned long long v[R];
  cc_wave_sync();
#pragma unroll
  for (int a = 0; a < R; a++) v[a] = (a * 64 + lane < cnt) ? buf[a * 64 + lane] : ~0ull;
  cc_wave_bitonic_u64<R>(v, lane);
  cc_wave_sync();
  if (lane < nnk) buf[lane] = v[0];
  first = v[0];
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(v[0] >> 32), nnk - 1));
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
  const int n = P.n_sorted[ll];
  if (!(sum != 0.f) || n <= 0) {  // q_keys[seq].sum() != 0 (contour_db.h:726)
    if (lane == 0) hit_cnt[blockIdx.x] = 0;
    return;
  }
  const cc_query_meta *qm = qmeta + q;
  // dist_ub (contour_db.h:733-749), f32 results of f64 products exactly as written there
  const float b00 = (float)((double)k[0] * 0.8), b01 = (float)((double)k[0] / 0.8);
  const float b10 = (float)((double)k[1] * 0.8), b11 = (float)((double)k[1] / 0.8);
  const float b20 = (float)((double)k[2] * 0.8 * 0.75), b21 = (float)((double)k[2] / (0.8 * 0.75));
  const float t0a = (k[0] - b00) * (k[0] - b00), t0b = (k[0] - b01) * (k[0] - b01);
  const float t1a = (k[1] - b10) * (k[1] - b10), t1b = (k[1] - b11) * (k[1] - b11);
  const float t2a = (k[2] - b20) * (k[2] - b20), t2b = (k[2] - b21) * (k[2] - b21);
  float ub = (t0a < t0b ? t0b : t0a) + (t1a < t1b ? t1b : t1a) + (t2a < t2b ? t2b : t2a);
  const float *K = P.skeys[ll];
  const int *sid = P.sid[ll];
  const int *sact = P.sact[ll];
  const unsigned cap = (unsigned)P.cap_k;
  const int epoch = qm->epoch;
  const int nnk = P.nnk;

  // ---- index ranges of the visible buckets (src/cont2/contour_db.cpp:322-369: mid = the bucket of the anchor's first
  // dimension, visited are {0..mid} and {mid + i : i > mid, mid + i < 6}) and the anchor's own position.
  // A key sits in bucket b iff rg[b] <= key[0] < rg[b + 1], i.e. iff its sorted index is in [lb(rg[b]), lb(rg[b + 1])),
  // lb(t) = number of keys with key[0] < t.
  int L0, E1, S2, E2, right;
  {
    float rg[7];
#pragma unroll
    for (int i = 0; i < 7; i++) rg[i] = qm->ranges[ll][i];
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
    float t_e1 = rg[6], t_s2 = rg[6];
#pragma unroll
    for (int i = 1; i < 7; i++) {
      if (i == mid + 1) t_e1 = rg[i];
      if (i == 2 * mid + 1) t_s2 = rg[i];
    }
    const int g = lane >> 3, sub = lane & 7;  // search g: 0 -> lb(rg[0]), 1 -> E1, 2 -> S2, 3 -> lb(rg[6]), 4.. -> lb(k[0])
    const float tg = g == 0 ? rg[0] : g == 1 ? t_e1 : g == 2 ? t_s2 : g == 3 ? rg[6] : k[0];
    int lo = 0, hi = n;
    while (__ballot(hi > lo) != 0ull) {
      const int len = hi - lo;
      const int w = (len + 8) / 9 > 0 ? (len + 8) / 9 : 1;
      // probes p_j = min(lo + (j + 1) w - 1, hi - 1), j = 0..7: key[0] < target is true up to some j, false after
      int p = lo + (sub + 1) * w - 1;
      p = p < hi - 1 ? p : hi - 1;
      const bool below = (len > 0) && (K[(unsigned)(p < 0 ? 0 : p)] < tg);
      const int c = __popc((unsigned)(__ballot(below) >> (lane & 56)) & 0xFFu);
      if (len > 0) {
        int pl = lo + c * w - 1;  // p_{c-1}
        pl = pl < hi - 1 ? pl : hi - 1;
        int ph = lo + (c + 1) * w - 1;  // p_c
        ph = ph < hi - 1 ? ph : hi - 1;
        if (c < 8) hi = ph;
        if (c > 0) lo = pl + 1;
      }
    }
    L0 = __builtin_amdgcn_readlane(lo, 0);
    E1 = __builtin_amdgcn_readlane(lo, 8);
    S2 = __builtin_amdgcn_readlane(lo, 16);
    E2 = __builtin_amdgcn_readlane(lo, 24);
    right = __builtin_amdgcn_readlane(lo, 32);
  }
  // upwards: indices >= right of [L0, E1) then of [S2, E2); downwards: indices < right of [S2, E2) then of [L0, E1),
  // each as one run of "virtual" positions 0, 1, 2, ...
  const int ua0 = L0 > right ? L0 : right, ua_len = E1 > ua0 ? E1 - ua0 : 0;
  const int ub0 = S2 > right ? S2 : right, ub_len = E2 > ub0 ? E2 - ub0 : 0;
  const int db_top = (E2 < right ? E2 : right) - 1, db_len = db_top + 1 > S2 ? db_top + 1 - S2 : 0;
  const int da_top = (E1 < right ? E1 : right) - 1, da_len = da_top + 1 > L0 ? da_top + 1 - L0 : 0;
  const int tot[2] = {ua_len + ub_len, db_len + da_len};

  int cnt = 0;
  bool tightened = false;
  bool open[2] = {tot[0] > 0, tot[1] > 0};
  int vpos[2] = {lane, lane};  // this lane's virtual position in the current step of each direction
  float c[2][CC_KEY_DIM];
  int act[2], kid[2];
  bool inside[2];
#define CC_KNN_FETCH(dir)                                                                                        \
  {                                                                                                              \
    const int v_ = vpos[dir];                                                                                    \
    inside[dir] = v_ < tot[dir];                                                                                 \
    int i_ = (dir) == 0 ? (v_ < ua_len ? ua0 + v_ : ub0 + (v_ - ua_len)) : (v_ < db_len ? db_top - v_ : da_top - (v_ - db_len)); \
    i_ = inside[dir] ? i_ : 0;                                                                                   \
    const unsigned u_ = (unsigned)i_;                                                                            \
    act[dir] = sact[u_];                                                                                         \
    kid[dir] = sid[u_];                                                                                          \
    _Pragma("unroll") for (int d = 0; d < CC_KEY_DIM; d++) c[dir][d] = K[(size_t)d * cap + u_];                 \
  }
  if (open[0]) CC_KNN_FETCH(0)
  if (open[1]) CC_KNN_FETCH(1)
  while (open[0] || open[1]) {
    bool pass[2] = {false, false};
    float res[2] = {0.f, 0.f};
    int kcur[2] = {0, 0};
    bool last_in[2] = {false, false};
    float far2[2] = {0.f, 0.f};
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      if (!open[dir]) continue;  // wave-uniform
      const float e0 = k[0] - c[dir][0];
      {
        // L2_Adaptor::evalMetric accumulation order (nanoflann.hpp:427-461)
        float r = 0.f;
        float d0 = e0, d1 = k[1] - c[dir][1], d2 = k[2] - c[dir][2], d3 = k[3] - c[dir][3];
        r += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        d0 = k[4] - c[dir][4];
        d1 = k[5] - c[dir][5];
        d2 = k[6] - c[dir][6];
        d3 = k[7] - c[dir][7];
        r += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        d0 = k[8] - c[dir][8];
        r += d0 * d0;
        d0 = k[9] - c[dir][9];
        r += d0 * d0;
        res[dir] = r;
        // before nnk candidates are known a key must be strictly inside dist_ub; afterwards keys AT the nnk-th best
        // distance still compete, on the key id
        pass[dir] = inside[dir] && act[dir] <= epoch && (tightened ? (r <= ub) : (r < ub));
      }
      kcur[dir] = kid[dir];
      // the step's outermost key decides whether the direction goes on: (key[0] - q[0])^2 is a lower bound of the
      // distance and grows outwards
      last_in[dir] = __builtin_amdgcn_readlane((int)inside[dir], 63) != 0;
      const float e_far = cc_lane_bcast(e0, 63);
      far2[dir] = e_far * e_far;
      // the next step's keys travel while this step's candidates are filed
      vpos[dir] += 64;
      CC_KNN_FETCH(dir)
    }
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
      if (!open[dir]) continue;
      const unsigned long long m = __ballot(pass[dir]);
      if (pass[dir]) buf[cnt + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)__float_as_uint(res[dir]) << 32) | (unsigned)kcur[dir];
      cnt += __popcll(m);
    }
    if (cnt >= 2 * nnk || (!tightened && cnt >= nnk)) {  // keep the best nnk (by distance, then key id); the radius follows
      unsigned long long first;
      ub = cnt <= 128 ? cc_knn_reduce<2>(buf, cnt, nnk, lane, first) : cc_knn_reduce<4>(buf, cnt, nnk, lane, first);
      cnt = nnk;
      tightened = true;
      cc_wave_sync();
    }
#pragma unroll
    for (int dir = 0; dir < 2; dir++)
      if (open[dir]) open[dir] = last_in[dir] && (tightened ? (far2[dir] <= ub) : (far2[dir] < ub));
  }
#undef CC_KNN_FETCH
  {
    unsigned long long first;
    if (cnt <= 128)
      cc_knn_reduce<2>(buf, cnt, nnk, lane, first);
    else
      cc_knn_reduce<4>(buf, cnt, nnk, lane, first);
    const int mm = cnt < nnk ? cnt : nnk;
    if (lane < mm) {
      const unsigned id = (unsigned)(first & 0xFFFFFFFFu);
      cc_knn_hit_t h;
      h.gidx = P.kgidx[ll][id];
      h.level = (int16_t)level;
      h.seq = (int16_t)P.kseq[ll][id];
      h.dist_sq = __uint_as_float((unsigned)(first >> 32));
      out[lane] = h;
    }
    if (lane == 0) hit_cnt[blockIdx.x] = mm;
  }
}
