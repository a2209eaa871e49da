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
  const float *skeys[CC_NQLEV];       // SoA [CC_KEY_DIM + 1][cap_k], sorted by dim 0 (ties: insertion order); last row = |key|^2
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
  // What the buckets' kd-trees INDEX at this epoch (cc_hostdb.h, Bucket::idx_lo / idx_hi): a key in a tree is found by the
  // reference only if its bucket's index has been rebuilt since the key arrived there.  idx_full[l] != 0: every bucket of
  // the layer indexes its whole range (the steady state, no per-key test); else a key of bucket b is visible iff
  // idx[l][2 b] <= key[0] < idx[l][2 b + 1].
  int idx_full[CC_NQLEV];
  float idx[CC_NQLEV][12];
};

// the slow path of the visibility test (a layer with a bucket whose index does not cover its range at the query's epoch)
__device__ __forceinline__ bool cc_knn_key_indexed(const cc_query_meta *qm, int ll, float k0) {
  bool ok = false;
#pragma unroll
  for (int b = 0; b < 6; b++) {
    const bool in_b = qm->ranges[ll][b] <= k0 && k0 < qm->ranges[ll][b + 1];
    ok = ok || (in_b && qm->idx[ll][2 * b] <= k0 && k0 < qm->idx[ll][2 * b + 1]);
  }
  return ok;
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
// A: every new key finds its rank among the new keys and among the old sorted ones (binary search); B: every old key is
//    shifted by the number of new keys below it; both write into the other buffer.
// Rank among the new keys: one workgroup sorts (key[0], arrival index) pairs in LDS when the append is small (an online
// add of a few hundred scans: m <= CC_KSORT_LDS keys); a bulk load of a prebuilt database falls back to counting.

// workgroup-wide bitonic sort of n_pow2 keys in LDS
__device__ __forceinline__ void cc_bitonic_sort_u64(unsigned long long *a, int n_pow2, int tid, int nt) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pow2; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = a[i], y = a[ixj];
          const bool up = ((i & k) == 0);
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

#define CC_KSORT_LDS 4096
// One append touches every query layer: the three kernels below take all layers in ONE launch each (the layer is the
// slow index of the 1-D grid), so an online append is a chain of four short launches, not ten.
struct cc_ksort_params {
  const float *keys[CC_NQLEV];     // insertion order, SoA [CC_KEY_DIM][cap_k]
  const float *s_old[CC_NQLEV];    // sorted view before the append ([CC_KEY_DIM + 1][cap_k]) ...
  const int *sid_old[CC_NQLEV];
  float *s_new[CC_NQLEV];          // ... and after it (the other buffer); == s_old when the layer got no new key
  int *sid_new[CC_NQLEV];
  int *newpos[CC_NQLEV];           // scratch per layer: position of the j-th new key in the new view
  float *newsorted0[CC_NQLEV];     // scratch per layer: key[0] of the new keys, ascending
  const int *act[CC_NQLEV];        // by key id: first epoch at which the key sits in a tree
  int *sact_new[CC_NQLEV];         // the same in the new view's order
  int n_old[CC_NQLEV], m[CC_NQLEV];
  int cap_k, n_layers, blocks_per_layer;
};

// grid = n_layers, block = 1024; m <= CC_KSORT_LDS.  Keys are compared as floats by cc_k_ksort_merge's searches, so the sort key
// must order exactly like `<` on floats: -0 is folded onto +0 first (NaN keys never get here: pushBuffer drops them).
__global__ void __launch_bounds__(1024)
cc_k_ksort_new_lds(cc_ksort_params P) {
  __shared__ unsigned long long a[CC_KSORT_LDS];
  const int l = blockIdx.x, tid = threadIdx.x;
  const int m = P.m[l], n_old = P.n_old[l];
  if (m <= 0 || m > CC_KSORT_LDS) return;  // nothing new / counted by cc_k_ksort_new instead
  const float *keys = P.keys[l], *s_old0 = P.s_old[l];
  int np2 = 64;
  while (np2 < m) np2 <<= 1;
  for (int i = tid; i < np2; i += 1024) {
    unsigned long long v = ~0ull;
    if (i < m) {
      const float c = keys[n_old + i] + 0.f;  // -0 -> +0
      v = ((unsigned long long)cc_fkey(c) << 32) | (unsigned)i;
    }
    a[i] = v;
  }
  __syncthreads();
  cc_bitonic_sort_u64(a, np2, tid, 1024);
  for (int r = tid; r < m; r += 1024) {
    const int j = (int)(a[r] & 0xFFFFFFFFu);
    const float c = keys[n_old + j];
    int lo = 0, hi = n_old;  // #old keys with c0 <= c (old keys precede new ones among equals)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_old0[mid] <= c)
        lo = mid + 1;
      else
        hi = mid;
    }
    P.newpos[l][j] = lo + r;
    P.newsorted0[l][r] = c;
  }
}

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

// The new keys of an append, written from the scans' hot records (already in the DB's array) into the layer's
// insertion-ordered SoA key matrix; `ent` packs (scan index << 3 | anchor) per new key, layer after layer.
// grid = ceil(total / 256), block = 256
struct cc_kappend_params {
  float *keys[CC_NQLEV];
  int *kgidx[CC_NQLEV];
  unsigned char *kseq[CC_NQLEV];
  int first[CC_NQLEV + 1];  // entries of layer l: [first[l], first[l + 1])
  int n_old[CC_NQLEV];
  int q_levels[CC_NQLEV];
  int cap_k;
  // small appends only (else act_first[CC_NQLEV] == 0): the changed tails of the layers' activation arrays ride along --
  // tail l = ent[act_src[l] .. act_src[l] + (act_first[l + 1] - act_first[l])) goes to act_dst[l][..]
  int *act_dst[CC_NQLEV];
  int act_src[CC_NQLEV];
  int act_first[CC_NQLEV + 1];
};
// grid = ceil(max(entries, tail ints) / 256), block = 256.  `ent` is the append's staging buffer: device memory, or (small
// appends) the pinned host buffer itself -- a few hundred bytes are not worth a copy command in front of the kernel.
__global__ void __launch_bounds__(256)
cc_k_keys_append(cc_kappend_params P, const cc_hot_desc_t *__restrict__ hot, const int *__restrict__ ent) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < P.act_first[CC_NQLEV]) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < CC_NQLEV; i++)
      if (t >= P.act_first[i]) l = i;
    const int j = t - P.act_first[l];
    P.act_dst[l][j] = ent[P.act_src[l] + j];
  }
  if (t >= P.first[CC_NQLEV]) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < CC_NQLEV; i++)
    if (t >= P.first[i]) l = i;
  const int j = t - P.first[l];
  const int e = ent[t];
  const int gidx = e >> 3, seq = e & 7;
  const float *k = &hot[gidx].keys[P.q_levels[l] - 1][seq][0];
  const size_t dst = (size_t)(P.n_old[l] + j);
#pragma unroll
  for (int d = 0; d < CC_KEY_DIM; d++) P.keys[l][(size_t)d * P.cap_k + dst] = k[d];
  P.kgidx[l][dst] = gidx;
  P.kseq[l][dst] = (unsigned char)seq;
}

// grid = n_layers * blocks_per_layer (blocks_per_layer >= ceil(max_l(n_old + m) / 256)), block = 256
__global__ void __launch_bounds__(256)
cc_k_ksort_merge(cc_ksort_params P) {
  const int l = blockIdx.x / P.blocks_per_layer;
  const int i = (blockIdx.x - l * P.blocks_per_layer) * blockDim.x + threadIdx.x;
  const int n_old = P.n_old[l], m = P.m[l];
  if (m <= 0) return;  // the layer keeps its buffer
  const size_t cap_k = (size_t)P.cap_k;
  const float *s_old = P.s_old[l], *keys = P.keys[l], *new_sorted0 = P.newsorted0[l];
  float *s_new = P.s_new[l];
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
    for (int d = 0; d <= CC_KEY_DIM; d++) s_new[(size_t)d * cap_k + dst] = s_old[(size_t)d * cap_k + i];  // row CC_KEY_DIM: |key|^2
    P.sid_new[l][dst] = P.sid_old[l][i];
    P.sact_new[l][dst] = P.act[l][P.sid_old[l][i]];
  } else if (i < n_old + m) {
    const int j = i - n_old, dst = P.newpos[l][j];
    float nrm = 0.f;
    for (int d = 0; d < CC_KEY_DIM; d++) {
      const float v = keys[(size_t)d * cap_k + n_old + j];
      s_new[(size_t)d * cap_k + dst] = v;
      nrm += v * v;
    }
    s_new[(size_t)CC_KEY_DIM * cap_k + dst] = nrm;  // input of the tiled search's prefilter only (cc_k_knn_tile)
    P.sid_new[l][dst] = n_old + j;
    P.sact_new[l][dst] = P.act[l][n_old + j];
  }
}

// activation epochs in sorted order for the layers that got no new key (they change when the host moves keys from a
// bucket's buffer into its tree); the layers that did get theirs from the merge.  Same grid as cc_k_ksort_merge.
__global__ void __launch_bounds__(256)
cc_k_ksort_act(cc_ksort_params P) {
  const int l = blockIdx.x / P.blocks_per_layer;
  const int i = (blockIdx.x - l * P.blocks_per_layer) * blockDim.x + threadIdx.x;
  if (P.m[l] > 0 || i >= P.n_old[l]) return;
  P.sact_new[l][i] = P.act[l][P.sid_old[l][i]];
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

// grid = nq * CC_NQLEV * CC_NPIV, block = 64 (one wave per anchor key).  VIS: the chunk has a query at an epoch at which some
// bucket's kd-tree does not index its whole range (cc_query_meta::idx_full): that instance tests every candidate key against
// its bucket's indexed interval; the common instance carries none of it (the test in the shared loop cost the walk 8 %:
// the kernel sits at its scalar-register limit).
template <bool VIS>
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
  const bool idx_full = VIS ? qm->idx_full[ll] != 0 : true;  // wave-uniform

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
        if (VIS && !idx_full && pass[dir]) pass[dir] = cc_knn_key_indexed(qm, ll, c[dir][0]);  // rare: see cc_query_meta
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

// ------------------------------------------------------------------------------------------------
// Ordering of a chunk's searches by key[0] and their grouping, for the tiled search below.  (Round 2's "shared walk" --
// one wave scoring four neighbouring searches against each 64-key step -- lived here; measured at the 50 000-scan DB it
// was written for it gained nothing (148.1 k vs 147.5 k scans/s, profiles/r3/d_knn_shared_walk_50k.txt) and was removed.)
// ------------------------------------------------------------------------------------------------
#define CC_KNN_ORDER_CAP 8192  // searches of one layer in a chunk (QB * CC_NPIV = 6144), padded to a power of two
#define CC_KNN_ORDER_GROUP 16  // searches per group of the tiled search (= CC_KNN_TQ)
#define CC_KNN_ORDER_LDS (12 * CC_KNN_ORDER_CAP)  // dynamic LDS of cc_k_knn_order: three 4-byte arrays per search

// Ascending bitonic sort of 1024 * R 32-bit keys by a workgroup of 1024 threads, R keys per lane in registers: position
// P = (wave * R + a) * 64 + lane lives in v[a].  Partners inside a lane are register moves, inside a wave shuffles; only
// the log2(16) * (log2(16) + 1) / 2 = 10 stages whose partner sits in another wave go through LDS (xch, 1024 * R words) --
// a sort of 8 192 keys with every stage in LDS (91 barriers over 64 KB) took 0.5 ms on the three workgroups this kernel
// has, longer than the search it prepares.
template <int R>
__device__ __forceinline__ void cc_block_bitonic_u32(unsigned (&v)[R], unsigned *xch, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int N = 1024 * R;
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * R) {  // partner in another wave
#pragma unroll
        for (int a = 0; a < R; a++) xch[(wave * R + a) * 64 + lane] = v[a];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < R; a++) {
          const int P = (wave * R + a) * 64 + lane;
          const unsigned o = xch[P ^ j];
          const bool up = (P & k) == 0, lower = (P & j) == 0;
          const unsigned mn = o < v[a] ? o : v[a], mx = o < v[a] ? v[a] : o;
          v[a] = (lower == up) ? mn : mx;
        }
        __syncthreads();
      } else if (j >= 64) {  // partner in the same lane
#pragma unroll
        for (int a = 0; a < R; a++) {
          const int b = a ^ (j >> 6);
          if (b > a) {
            const bool up = (((wave * R + a) * 64) & k) == 0;
            const unsigned x = v[a], y = v[b];
            const bool sw = (x > y) == up;
            v[a] = sw ? y : x;
            v[b] = sw ? x : y;
          }
        }
      } else {
#pragma unroll
        for (int a = 0; a < R; a++) {
          const unsigned o = (unsigned)__shfl_xor((int)v[a], j);
          const bool up = ((((wave * R + a) * 64 + lane) & k) == 0);
          const bool lower = (lane & j) == 0;
          const unsigned mn = o < v[a] ? o : v[a], mx = o < v[a] ? v[a] : o;
          v[a] = (lower == up) ? mn : mx;
        }
      }
    }
  }
}

// sort key of a search: (quantised log of key[0]) << 13 | search index; the top bits are the geometric bucket (ratio 1.04)
#define CC_KNN_ORD_IDX_BITS 13
#define CC_KNN_ORD_SUB_BITS 6
static_assert((1 << CC_KNN_ORD_IDX_BITS) >= CC_KNN_ORDER_CAP, "search index bits of the order key");
__device__ __forceinline__ unsigned cc_knn_order_key(float q0, int i) {
  // floor(log2(q0) * 17.673 * 64) + 177 * 64: >= 0 for q0 >= 1e-3, < 2^19 for any float
  int ql = (int)floorf(log2f(q0 > 1e-3f ? q0 : 1e-3f) * (17.673f * (1 << CC_KNN_ORD_SUB_BITS))) + (177 << CC_KNN_ORD_SUB_BITS);
  ql = ql < 0 ? 0 : (ql > (1 << 18) ? (1 << 18) : ql);
  return ((unsigned)ql << CC_KNN_ORD_IDX_BITS) | (unsigned)i;
}
template <int R>
__device__ __forceinline__ void cc_knn_order_sort(const cc_knn_params &P, const cc_hot_desc_t *__restrict__ qhot, int ns, int ll, int level,
                                                  int *__restrict__ hit_cnt, unsigned *sk, int tid, int *nv) {
  const int lane = tid & 63, wave = tid >> 6;
  unsigned v[R];
  int mine = 0;
#pragma unroll
  for (int a = 0; a < R; a++) {
    const int i = (wave * R + a) * 64 + lane;
    v[a] = 0xFFFFFFFFu;
    if (i < ns) {
      const int q = i / CC_NPIV, seq = i - q * CC_NPIV;
      const float *qk = &qhot[q].keys[level - 1][seq][0];
      float sum = 0.f;
#pragma unroll
      for (int d = 0; d < CC_KEY_DIM; d++) sum += qk[d];
      if (sum != 0.f && P.n_sorted[ll] > 0) {
        v[a] = cc_knn_order_key(qk[0], i);
        mine++;
      } else {
        hit_cnt[q * (CC_NQLEV * CC_NPIV) + ll * CC_NPIV + seq] = 0;
      }
    }
  }
  if (mine) atomicAdd(nv, mine);
  cc_block_bitonic_u32<R>(v, sk, tid);
#pragma unroll
  for (int a = 0; a < R; a++) sk[(wave * R + a) * 64 + lane] = v[a];
  __syncthreads();
}

// block-wide inclusive scan (sum, or running maximum; values >= 0) of n_pow2 ints in LDS, in place: every thread takes a
// stretch of consecutive entries, the stretches' totals are scanned inside the waves and across them.  block = 1024.
template <bool MAX>
__device__ __forceinline__ void cc_block_scan(int *a, int n_pow2, int tid, int *wsum /*[16]*/) {
  const int E = n_pow2 >= 1024 ? n_pow2 / 1024 : 1;
  const int base = tid * E, lane = tid & 63, wave = tid >> 6;
  int run = 0;
  if (base < n_pow2)
    for (int e = 0; e < E; e++) {
      const int v = a[base + e];
      run = MAX ? (v > run ? v : run) : run + v;
      a[base + e] = run;
    }
  int incl = run;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl = MAX ? (v > incl ? v : incl) : incl + v;
  }
  int excl = __shfl_up(incl, 1);  // of the lanes before this one
  if (lane == 0) excl = 0;
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; w++) excl = MAX ? (wsum[w] > excl ? wsum[w] : excl) : excl + wsum[w];
  if (base < n_pow2)
    for (int e = 0; e < E; e++) {
      const int v = a[base + e];
      a[base + e] = MAX ? (v > excl ? v : excl) : v + excl;
    }
  __syncthreads();
}

// Layout of the per-chunk order buffer (ints): per layer the searches in key[0] order, the group starts of the tiled search,
// then the counts.
#define CC_KNN_ORD_ORDER(ll) ((ll) * CC_KNN_ORDER_CAP)
#define CC_KNN_ORD_GSTART(ll) (CC_NQLEV * CC_KNN_ORDER_CAP + (ll) * (CC_KNN_ORDER_CAP + 1))
#define CC_KNN_ORD_NVALID (CC_NQLEV * CC_KNN_ORDER_CAP + CC_NQLEV * (CC_KNN_ORDER_CAP + 1))
#define CC_KNN_ORD_NGROUP (CC_KNN_ORD_NVALID + CC_NQLEV)
#define CC_KNN_ORD_INTS (CC_KNN_ORD_NGROUP + CC_NQLEV)

// grid = n_q_levels, block = 1024.  order[ll][i] = search (q * CC_NPIV + seq) with the i-th smallest key[0] among the
// layer's searches that have a key (q_keys[seq].sum() != 0, contour_db.h:726); the others get their empty result here.
// For the tiled search the ordered searches are also cut into GROUPS of at most 16 whose key[0] lie within ~4 % of each
// other (a group walks the union of its searches' windows: where searches are sparse -- the large keys -- a fixed
// sixteen would span a multiple of a window): gstart[ll][g] .. gstart[ll][g + 1] are group g's positions in `order`.
__global__ void __launch_bounds__(1024)
cc_k_knn_order(cc_knn_params P, const cc_hot_desc_t *__restrict__ qhot, int nq, int *__restrict__ ord, int *__restrict__ hit_cnt) {
  HIP_DYNAMIC_SHARED(char, smem)  // CC_KNN_ORDER_LDS bytes
  unsigned *sk = (unsigned *)smem;                       // [CC_KNN_ORDER_CAP] sort exchange, then the sorted keys
  int *sa = (int *)(smem + 4 * CC_KNN_ORDER_CAP);         // [CC_KNN_ORDER_CAP]
  int *fl = (int *)(smem + 8 * CC_KNN_ORDER_CAP);         // [CC_KNN_ORDER_CAP]
  __shared__ int nv;
  __shared__ int wsum[16];
  const int ll = blockIdx.x, tid = threadIdx.x;
  const int level = P.q_levels[ll];
  const int ns = nq * CC_NPIV;
  int *order = ord + CC_KNN_ORD_ORDER(ll), *gstart = ord + CC_KNN_ORD_GSTART(ll);
  if (tid == 0) nv = 0;
  if (ll == 0)  // layers the configuration does not query: empty results
    for (int i = tid; i < nq * (CC_NQLEV - P.n_q_levels) * CC_NPIV; i += 1024) {
      const int q = i / ((CC_NQLEV - P.n_q_levels) * CC_NPIV), r = i - q * ((CC_NQLEV - P.n_q_levels) * CC_NPIV);
      hit_cnt[q * (CC_NQLEV * CC_NPIV) + P.n_q_levels * CC_NPIV + r] = 0;
    }
  __syncthreads();
  // the chunk's searches of this layer by (key[0], search), padded to the sort's width
  int np2;
  if (ns <= 1024) {
    np2 = 1024;
    cc_knn_order_sort<1>(P, qhot, ns, ll, level, hit_cnt, sk, tid, &nv);
  } else if (ns <= 4096) {
    np2 = 4096;
    cc_knn_order_sort<4>(P, qhot, ns, ll, level, hit_cnt, sk, tid, &nv);
  } else {
    np2 = 8192;
    cc_knn_order_sort<8>(P, qhot, ns, ll, level, hit_cnt, sk, tid, &nv);
  }
  const int nvl = nv;
  for (int i = tid; i < nvl; i += 1024) order[i] = (int)(sk[i] & ((1u << CC_KNN_ORD_IDX_BITS) - 1u));
  // groups: geometric key[0] buckets of ratio 1.04, each cut into runs of 16
  for (int i = tid; i < np2; i += 1024) {
    int head = 0;
    if (i < nvl && i > 0) {
      const unsigned b0 = sk[i] >> (CC_KNN_ORD_IDX_BITS + CC_KNN_ORD_SUB_BITS), bp = sk[i - 1] >> (CC_KNN_ORD_IDX_BITS + CC_KNN_ORD_SUB_BITS);
      head = b0 != bp ? i : 0;
    }
    sa[i] = head;  // index of the bucket's first search where a bucket starts, else 0 (search 0 starts the first bucket)
  }
  __syncthreads();
  cc_block_scan<true>(sa, np2, tid, wsum);  // sa[i] = first search of i's bucket
  for (int i = tid; i < np2; i += 1024) fl[i] = (i < nvl && ((i - sa[i]) & (CC_KNN_ORDER_GROUP - 1)) == 0) ? 1 : 0;
  __syncthreads();
  cc_block_scan<false>(fl, np2, tid, wsum);  // fl[i] = groups started up to and including i
  for (int i = tid; i < nvl; i += 1024)
    if (i == 0 || fl[i] != fl[i - 1]) gstart[fl[i] - 1] = i;
  if (tid == 0) {
    const int ng = nvl > 0 ? fl[nvl - 1] : 0;
    gstart[ng] = nvl;
    ord[CC_KNN_ORD_NVALID + ll] = nvl;
    ord[CC_KNN_ORD_NGROUP + ll] = ng;
  }
}

// ------------------------------------------------------------------------------------------------
// K3, tiled form (cc_db: env CC_KNN_MODE=2): 16 searches per workgroup of four waves, distances on the matrix cores.
//
// The wave-per-search walk evaluates ~15 % of a layer per search whatever the layer's size (the 50-th neighbour in ten
// dimensions is far in every single one), ~100 vector instructions per 64 keys and search: at a 50 000-scan DB that is
// 830 M key evaluations per 1 024 queries and 60 % of the step.  Here a workgroup takes 16 searches that are adjacent in
// the chunk's key[0] order (cc_k_knn_order) -- their windows of the sorted view nearly coincide -- and walks the union
// of their windows once; per round its four waves take the next two 64-key steps upwards and the next two downwards:
//   * PREFILTER on v_mfma_f32_16x16x4_f32: with the keys as rows (k_0..k_9, |k|^2, 1) and the searches as columns
//     (-2 q_0..-2 q_9, 1, |q|^2) three instructions give the 16 x 16 squared distances of a tile, four tiles per step.
//     The result is a k-ordered fmaf chain, NOT nanoflann's sum, so it only FILTERS: a pair goes on iff its value is
//     <= the search's current radius + a bound on what the chain can be off by (cc_knn_tile_slack) and the key's index
//     lies in the search's visible bucket ranges.  Every pair whose reference distance is inside the radius passes;
//     about 1 % of all pairs do.  The sixteen compares of a step each yield a wave mask; most are empty.
//   * the pairs that pass are queued in LDS (one queue per wave) and worked off by all 256 threads once 256 are pending
//     (one pair per thread, every load independent): the squared distance in nanoflann's accumulation order, the epoch
//     mask, then the search's candidate buffer.  A buffer that fills up is cut back to the candidates within its nnk-th
//     smallest distance (found by bisection on the distance bits with wave ballots -- no sort), which becomes the radius.
//     A radius that is tightened late only lets more pairs through.
// Which keys a search ends up with does not depend on the filter (any superset of the final set gives the same result);
// the final order is (distance, key id): hit lists are bit-identical to cc_k_knn's.  The sorted view carries |k|^2 as an
// 11th row (cc_k_ksort_merge).
// ------------------------------------------------------------------------------------------------
#define CC_KNN_TILE_MIN_KEYS 24000  // cc_db picks the tiled search from this many keys in a layer on (~4 000 scans).  Round 6: 60 000 before;
                                    // at the 5 000-scan DB (30 k keys per layer) the tiled search gives the KITTI-shaped step +2.3 % (420-425 k ->
                                    // 431-434 k scans/s: no faster alone, but it leaves more of the chip to the other streams) and costs the
                                    // sparse world 0.5 % (482-486 k -> 481-483 k: every query there has its own keys in the DB bit for bit and the
                                    // walk's radius collapses at once) -- profiles/r6/ab_knn_mode.txt
#define CC_KNN_TQ 16      // searches per workgroup = columns of a 16x16x4 tile
#define CC_KNN_TW 8       // waves per workgroup: half of them walk upwards, half downwards
#define CC_KNN_TSTRIDE (64 * (CC_KNN_TW / 2))  // keys a direction advances by per round
#ifndef CC_KNN_TREP
#define CC_KNN_TREP 2     // 64-key steps a wave takes per round
#endif
#ifndef CC_KNN_TTRIG
#define CC_KNN_TTRIG 128  // a buffer holding this many candidates is cut back after a pass of the queue
#endif
#ifndef CC_KNN_TPASS
#define CC_KNN_TPASS 384  // pairs worked off per pass (threads 0 .. CC_KNN_TPASS - 1), and the queue length that starts one (round 3:
                          // 128 -- a pass every 1.5 rounds, each one a round trip to the keys with the whole workgroup waiting; rounds 4-6a:
                          // 256; 128 / 192 / 256 / 320 / 384 measured in round 6: cc_k_knn 0.49 / 0.44 / 0.42 / 0.40 / 0.39 ms at the 5 000-scan DB)
#endif
#ifndef CC_KNN_TCAP
#define CC_KNN_TCAP 512   // candidate buffer per search: < CC_KNN_TTRIG kept + CC_KNN_TPASS from one pass
#endif
#ifndef CC_KNN_TWL
#define CC_KNN_TWL 448    // queue per wave: a wave stops queueing once fewer than 64 places are left and carries the rest of its
                          // step's pairs over to the next round -- by then a pass has run: a queue that full holds CC_KNN_TPASS pairs
                          // (round 3 sized the queues for a step in which all 1 024 pairs pass: 38 KB that were never used)
#endif
// LDS: 64 KB of buffers + 14 KB of queues + 1.3 KB; with ~115 registers per lane two workgroups (16 waves) fit a CU
typedef float cc_f32x4 __attribute__((__vector_size__(4 * sizeof(float))));
static_assert(CC_KNN_TTRIG >= 2 * CC_KNN_MAX && CC_KNN_TTRIG - 1 + CC_KNN_TPASS <= CC_KNN_TCAP && CC_KNN_TPASS + 64 <= CC_KNN_TWL &&
                  CC_KNN_TPASS <= 64 * CC_KNN_TW && CC_KNN_TCAP <= 512, "cc_k_knn_tile: buffer bounds");

// |value of the fmaf chain - real squared distance| for every key whose real distance is within radius^2 = ub of the
// search: the chain sums 12 products of magnitude <= (|q| + |k|)^2 in total with one rounding each (<= 13 * 2^-24 relative
// to that sum), its two norm inputs carry <= 10 * 2^-24 relative error each: <= 23 * 2^-24 (|q| + |k|)^2 in all, assuming
// the matrix core rounds every product-accumulate once, to nearest; and |k| <= |q| + sqrt(ub) for such a key.
// 2^-16 * (2 |q| + sqrt(ub))^2 = 256 * 2^-24 (...)^2 is eleven times that bound (round 3 used 2^-18: only 2.8 times, the
// advisor's finding); the wider band lets a few more of the ~1 % of pairs through to the exact test and changes no result.
__device__ __forceinline__ float cc_knn_tile_slack(float qnorm2, float ub) {
  const float s = 2.f * sqrtf(qnorm2) + sqrtf(ub);
  return s * s * (1.f / 65536.f);
}

struct cc_knn_tstate {  // per search of a workgroup
  float ub;
  int cnt, tight;
};
struct cc_knn_tlds {
  unsigned long long buf[CC_KNN_TQ][CC_KNN_TCAP];
  unsigned wl[CC_KNN_TW][CC_KNN_TWL];  // (search << 28) | sorted index
  float qk[CC_KNN_TQ][CC_KEY_DIM];     // the searches' keys, for the threads that work off other searches' pairs
  int qb[CC_KNN_TQ][6];                // L0, E1, S2, E2 (visible index ranges), the epoch, the search's own position
  int qq[CC_KNN_TQ];                   // the search's query, or -1 when the layer's buckets index their whole ranges at its epoch (no per-key test)
  cc_knn_tstate st[CC_KNN_TQ];
  // per round parity (a wave may be one round ahead of another between two barriers):
  int wn[2][CC_KNN_TW];                // pairs pending in each wave's queue
  int go[2][CC_KNN_TW];                // the wave's sub-walk (every other step of its direction) still has a search to serve
#ifdef CC_KNN_TPAD
  char pad[CC_KNN_TPAD];               // tuning aid: LDS nobody uses (how many workgroups share a CU, and with whom)
#endif
};

// Cut a search's buffer back: keep the candidates with distance <= x for an x with nnk <= #kept <= nnk + SLACK (bisection
// on the distance bits with wave ballots, no sort); returns x -- an admissible radius: at least the nnk-th smallest distance,
// so every key that can still make the result passes r <= x.  SLACK = 0 gives exactly the nnk-th smallest distance (more
// than nnk are kept only when distances tie there); a few more kept candidates cost nothing and save most of the
// bisection's ~31 rounds.  One wave, cnt <= 64 * R.
// Round 6: the bisection starts from the data's own range -- lo = the smallest distance in the buffer, hi = the search's
// current radius `ub` (every candidate was admitted under it) -- instead of [0, inf): the first ~20 of its rounds only found
// the distances' exponent and leading mantissa bits (cut-backs were a quarter of cc_k_knn_tile at a 5 000-scan DB).
template <int R, int SLACK>
__device__ __forceinline__ float cc_knn_select(unsigned long long *buf, int cnt, int nnk, int lane, int &kept, float ub) {
  unsigned long long v[R];
  unsigned d[R];
  cc_wave_sync();
#pragma unroll
  for (int a = 0; a < R; a++) {
    v[a] = (a * 64 + lane < cnt) ? buf[a * 64 + lane] : ~0ull;
    d[a] = (unsigned)(v[a] >> 32);  // squared distances are >= 0: their bit patterns order like the values
  }
  unsigned lo = 0xFFFFFFFFu, hi = __float_as_uint(ub);  // invariant: #(d <= hi) >= nnk, #(d < lo) < nnk
#pragma unroll
  for (int a = 0; a < R; a++) lo = d[a] < lo ? d[a] : lo;  // (the padding's ~0 does not win)
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned o_ = (unsigned)__shfl_xor((int)lo, o);
    lo = o_ < lo ? o_ : lo;
  }
  if (!(hi <= 0x7F800000u) || lo > hi) {  // not a radius every candidate lies under (cannot happen): the whole range
    lo = 0u;
    hi = 0x7F800000u;
  }
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int c = 0;
#pragma unroll
    for (int a = 0; a < R; a++) c += __popcll(__ballot(d[a] <= mid));
    if (c >= nnk) {
      hi = mid;
      if (c <= nnk + SLACK) break;
    } else {
      lo = mid + 1u;
    }
  }
  cc_wave_sync();
  int off = 0;
#pragma unroll
  for (int a = 0; a < R; a++) {
    const bool keep = d[a] <= hi;
    const unsigned long long m = __ballot(keep);
    if (keep) buf[off + __popcll(m & ((1ull << lane) - 1ull))] = v[a];
    off += __popcll(m);
  }
  kept = off;
  return __uint_as_float(hi);
}

// A search's buffer of cnt <= CC_KNN_TCAP candidates cut back to those within its nnk-th smallest distance (a few more,
// see cc_knn_select); returns that distance, `kept` candidates stay at buf[0..kept), kept <= 64.
__device__ __forceinline__ float cc_knn_cut(unsigned long long *buf, int cnt, int nnk, int lane, int &kept, float ub /*the search's radius so far*/) {
  float nub = cnt <= 128 ? cc_knn_select<2, 12>(buf, cnt, nnk, lane, kept, ub)
                         : (cnt <= 256 ? cc_knn_select<4, 12>(buf, cnt, nnk, lane, kept, ub)
                                        : (cnt <= 384 ? cc_knn_select<6, 12>(buf, cnt, nnk, lane, kept, ub) : cc_knn_select<8, 12>(buf, cnt, nnk, lane, kept, ub)));
  if (kept > 64) {  // a crowd of exactly equal distances at the radius: only the nnk smallest (distance, key id) can end up
                    // in the result -- order them and drop the rest, so that the buffer bound holds
    unsigned long long first;
    nub = kept <= 128 ? cc_knn_reduce<2>(buf, kept, nnk, lane, first)
                      : (kept <= 256 ? cc_knn_reduce<4>(buf, kept, nnk, lane, first) : cc_knn_reduce<8>(buf, kept, nnk, lane, first));
    kept = nnk;
  }
  return nub;
}

// grid = n_q_levels * ceil(nq * CC_NPIV / CC_KNN_TQ), block = 64 * CC_KNN_TW.  PH: the phase timers (tuning aid) are compiled in.
template <bool PH>
__global__ void __launch_bounds__(64 * CC_KNN_TW)
cc_k_knn_tile(cc_knn_params P, const cc_hot_desc_t *__restrict__ qhot, const cc_query_meta *__restrict__ qmeta, int nq,
              const int *__restrict__ ordbuf /*cc_k_knn_order's output*/, cc_knn_hit_t *__restrict__ hits, int *__restrict__ hit_cnt,
              long long *__restrict__ phase_clk /*tuning aid (CC_KNN_PHASES=1), else nullptr: [grid][8] ticks of 10 ns*/) {
  __shared__ cc_knn_tlds L;
  long long pc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // setup | step | barrier | pass | cut-back | results | rounds | passes
  long long pt_ = PH ? wall_clock64() : 0;
#define CC_KNN_TICK(slot)                      \
  if (PH) {                                    \
    const long long now_ = wall_clock64();     \
    pc_[slot] += now_ - pt_;                   \
    pt_ = now_;                                \
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // one workgroup slot per search and layer (at most that many groups; most slots exit at once), layers interleaved
  const int ll = blockIdx.x % P.n_q_levels, w = blockIdx.x / P.n_q_levels;
  (void)nq;
  const int ng_ = ordbuf[CC_KNN_ORD_NGROUP + ll];
  if (w >= ng_) return;
  // the groups with the largest key[0] first: their windows are the widest (dist_ub grows with the key, the keys get
  // sparse), they set the kernel's duration and must not start last
  const int gsel = ng_ - 1 - w;
  const int *order_l = ordbuf + CC_KNN_ORD_ORDER(ll);
  const int base = ordbuf[CC_KNN_ORD_GSTART(ll) + gsel];
  const int ns = ordbuf[CC_KNN_ORD_GSTART(ll) + gsel + 1] - base;  // searches of this workgroup, 1 .. CC_KNN_TQ
  const int level = P.q_levels[ll];
  const int n = P.n_sorted[ll];
  const float *K = P.skeys[ll];
  const int *sid = P.sid[ll];
  const int *sact = P.sact[ll];
  const size_t cap = (size_t)P.cap_k;
  const int nnk = P.nnk;
  const int j = lane & 15, kq = lane >> 4;  // this lane's search (column) and its k-slice of every 4-wide MFMA step
  const int dir = wave / (CC_KNN_TW / 2), sub = wave % (CC_KNN_TW / 2);  // this wave's direction and which of the round's steps it takes

  // ---- this lane's search: key, dist_ub, epoch, bucket thresholds (cc_k_knn); padding columns repeat search 0
  const int srch = order_l[base + (j < ns ? j : 0)];
  const int q = srch / CC_NPIV, seq = srch - q * CC_NPIV;
  float k[CC_KEY_DIM];
  {
    const float *qk = &qhot[q].keys[level - 1][seq][0];
#pragma unroll
    for (int d = 0; d < CC_KEY_DIM; d++) k[d] = qk[d];
  }
  const cc_query_meta *qm = qmeta + q;
  float ub0;
  {
    const float b00 = (float)((double)k[0] * 0.8), b01 = (float)((double)k[0] / 0.8);
    const float b10 = (float)((double)k[1] * 0.8), b11 = (float)((double)k[1] / 0.8);
    const float b20 = (float)((double)k[2] * 0.8 * 0.75), b21 = (float)((double)k[2] / (0.8 * 0.75));
    const float t0a = (k[0] - b00) * (k[0] - b00), t0b = (k[0] - b01) * (k[0] - b01);
    const float t1a = (k[1] - b10) * (k[1] - b10), t1b = (k[1] - b11) * (k[1] - b11);
    const float t2a = (k[2] - b20) * (k[2] - b20), t2b = (k[2] - b21) * (k[2] - b21);
    ub0 = (t0a < t0b ? t0b : t0a) + (t1a < t1b ? t1b : t1a) + (t2a < t2b ? t2b : t2a);
  }
  float qn2 = 0.f;
#pragma unroll
  for (int d = 0; d < CC_KEY_DIM; d++) qn2 += k[d] * k[d];
  // B operand: column j of (-2 q_0 .. -2 q_9, 1, |q|^2), this lane's element of k-step s is 4 s + kq
  float bop[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < CC_KEY_DIM; d++)
      if (d == 4 * s + kq) v = -2.f * k[d];
    if (4 * s + kq == 10) v = 1.f;
    if (4 * s + kq == 11) v = qn2;
    bop[s] = v;
  }
  // index boundaries lb(t) = number of keys with key[0] < t: wave 0, lane (j, kq) -> target kq of search j (rg[0], t_e1,
  // t_s2, rg[6]); wave 1, lanes kq == 0 -> the search's own key[0]
  if (wave < 2) {
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
    const float tg = wave ? k[0] : (kq == 0 ? rg[0] : kq == 1 ? t_e1 : kq == 2 ? t_s2 : rg[6]);
    int lo = 0, hi = (wave && kq != 0) ? 0 : n;
    while (lo < hi) {
      const int m_ = (lo + hi) >> 1;
      if (K[m_] < tg)
        lo = m_ + 1;
      else
        hi = m_;
    }
    if (!wave)
      L.qb[j][kq] = lo;
    else if (kq == 0)
      L.qb[j][5] = lo;
  } else if (wave == 2) {
    if (kq == 0) {
      L.st[j].ub = ub0;
      L.st[j].cnt = 0;
      L.st[j].tight = 0;
      L.qb[j][4] = qm->epoch;
      L.qq[j] = qm->idx_full[ll] ? -1 : q;
#pragma unroll
      for (int d = 0; d < CC_KEY_DIM; d++) L.qk[j][d] = k[d];
    }
  }
  __syncthreads();
  const int L0 = L.qb[j][0], E1 = L.qb[j][1], S2 = L.qb[j][2], E2 = L.qb[j][3];
  const int p0 = __builtin_amdgcn_readfirstlane(L.qb[ns >> 1][5]);  // the middle search's own position splits the walk
  const bool valid = j < ns;
  float ubj = ub0;
  int tightj = 0;
  const float slack = cc_knn_tile_slack(qn2, ub0);
  float thr = ubj + slack;

  // ---- the common walk: upwards over [p0, n), downwards over [0, p0).  Each wave walks every other 64-key step of its
  // direction (steps sub, sub + 2, ...) and stops on its own: once the outermost key of ITS step lies beyond a search's
  // radius (or past its ranges) so does everything further out.  One barrier per round.
  bool open_d = valid && (dir == 0 ? (p0 < E2 && p0 < n) : (p0 > L0 && p0 > 0));  // this lane's search, this wave's direction
  int sb = dir == 0 ? p0 + 64 * sub : p0 - 64 * (sub + 1);  // first index of this wave's current step (ascending inside a step)
  if (sub > 0) open_d = open_d && (dir == 0 ? (sb < E2 && sb < n) : (sb + 64 > L0 && sb + 64 > 0));
  bool mine = __ballot(open_d) != 0ull;  // wave-uniform
  int wn = 0;                            // pairs pending in this wave's queue (wave-uniform)
  unsigned m_left = 0u;                  // this lane's pairs of the wave's last step that found no room in the queue yet
  int sb_left = 0;                       // ... and that step's first index
  bool has_left = false;                 // wave-uniform: some lane has such pairs
  float a[4][3];  // A operand of the fetched step: tile t = keys sb + 16 t + (lane & 15), element 4 s + kq
  // this lane's three rows of the sorted view (0..9 key dims, 10 = |k|^2; row 11 is the constant 1: any readable row, not used)
  const float *Krow[3];
#pragma unroll
  for (int s = 0; s < 3; s++) Krow[s] = K + (size_t)(4 * s + kq < CC_KEY_DIM + 1 ? 4 * s + kq : 0) * cap;
  const bool one_row = kq == 3;  // element 11
#define CC_KNN_TFETCH()                                                  \
  {                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; t++) {                      \
      int i_ = sb + 16 * t + j;                                          \
      i_ = i_ < 0 ? 0 : (i_ >= n ? n - 1 : i_);                          \
      a[t][0] = Krow[0][(unsigned)i_];                                   \
      a[t][1] = Krow[1][(unsigned)i_];                                   \
      const float v2_ = Krow[2][(unsigned)i_];                           \
      a[t][2] = one_row ? 1.f : v2_;                                     \
    }                                                                    \
  }
  if (mine) CC_KNN_TFETCH()
  CC_KNN_TICK(0)
  int n_pass_ = 0;  // passes so far (workgroup-uniform)
  for (int par = 0;; par ^= 1) {
    if (PH) pc_[6]++;
    if (has_left) {  // the rest of the last step's pairs first (the pass in between has emptied the queue)
      unsigned m = m_left;
      while (__ballot(m != 0u) != 0ull && wn + 64 <= CC_KNN_TWL) {
        const bool push = m != 0u;
        const int bit = __ffs(m) - 1;
        m &= m - 1u;
        const unsigned long long mk = __ballot(push);
        if (push) L.wl[wave][wn + __popcll(mk & ((1ull << lane) - 1ull))] = ((unsigned)j << 28) | (unsigned)(sb_left + 16 * (bit >> 2) + 4 * kq + (bit & 3));
        wn += __popcll(mk);
      }
      m_left = m;
      has_left = __ballot(m != 0u) != 0ull;
    } else {
      // a wave takes CC_KNN_TREP steps per round: the barrier, the look at the other waves' queues and the pass decision are
      // paid once per 128 keys
      for (int rep_ = 0; rep_ < CC_KNN_TREP && mine && !has_left; rep_++) {
      const int sb_cur = sb;
      cc_f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = (cc_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], bop[s], acc[t], 0, 0, 0);
      // the step's outermost key[0] (row 0 lives in the lanes kq == 0, element s = 0): upwards the last key, downwards the first
      const float far0 = dir == 0 ? cc_lane_bcast(a[3][0], 15) : cc_lane_bcast(a[0][0], 0);
      // the wave's next step travels while this one's pairs are filtered
      sb += dir == 0 ? CC_KNN_TSTRIDE : -CC_KNN_TSTRIDE;
      CC_KNN_TFETCH()
      // filter: D[row = 4 kq + r of tile t][column j] <= radius + slack -> one bit per pair, this lane's sixteen
      unsigned m = 0u;
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) m |= (acc[t][r] <= thr) ? (1u << (4 * t + r)) : 0u;
      if (!valid) m = 0u;
      // ... and the key inside the search's visible index ranges.  Usually the whole step lies inside one range of every
      // search (wave-uniform test); at a range's end the lane's bits are checked one by one
      {
        const int s0 = sb_cur, s1 = sb_cur + 64;  // the step's indices [s0, s1)
        const bool whole = s0 >= 0 && s1 <= n && ((s0 >= L0 && s1 <= E1) || (s0 >= S2 && s1 <= E2));
        if (__ballot(valid && !whole) != 0ull) {
          unsigned vis = 0u;
#pragma unroll
          for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int idx = sb_cur + 16 * t + 4 * kq + r;
              vis |= (idx >= 0 && idx < n && ((idx >= L0 && idx < E1) || (idx >= S2 && idx < E2))) ? (1u << (4 * t + r)) : 0u;
            }
          m &= vis;
        }
      }
      // queue the pairs, one per lane and turn; what finds no room waits for the next round
      while (__ballot(m != 0u) != 0ull && wn + 64 <= CC_KNN_TWL) {
        const bool push = m != 0u;
        const int bit = __ffs(m) - 1;  // -1 when m == 0 (unused)
        m &= m - 1u;
        const unsigned long long mk = __ballot(push);
        if (push) L.wl[wave][wn + __popcll(mk & ((1ull << lane) - 1ull))] = ((unsigned)j << 28) | (unsigned)(sb_cur + 16 * (bit >> 2) + 4 * kq + (bit & 3));
        wn += __popcll(mk);
      }
      m_left = m;
      sb_left = sb_cur;
      has_left = __ballot(m != 0u) != 0ull;
      // who goes on with this wave: a search leaves when the step's outermost key lies beyond its own key[0] on that side by
      // more than its radius (the radius may be a pass old: then it only leaves later), or when the wave's next step is past
      // its visible ranges
      {
        const float e = k[0] - far0;
        const bool beyond = dir == 0 ? (e < 0.f) : (e > 0.f);
        const bool out = beyond && (tightj ? (e * e > ubj) : (e * e >= ubj));
        const bool more = dir == 0 ? (sb < E2 && sb < n) : (sb + 64 > L0 && sb + 64 > 0);
        open_d = open_d && !out && more;
        mine = __ballot(open_d) != 0ull;
        // Buckets the searches do not visit (contour_db.cpp:341-369: the right neighbours mid + 1 .. 2 mid) are whole index
        // ranges: when the wave's next step touches no open search's visible range, jump to the first of its steps that does
        if (mine) {
          const int s0 = sb, s1 = sb + 64;
          const bool touch = open_d && ((s0 < E1 && s1 > L0) || (s0 < E2 && s1 > S2));
          if (__ballot(touch) == 0ull) {
            int tgt;  // nearest visible index in walking direction, as a distance >= 0 from the step's near end
            if (dir == 0)
              tgt = !open_d ? 0x7fffffff : (s0 < E1 ? (L0 > s0 ? L0 - s0 : 0) : (s0 < E2 ? (S2 > s0 ? S2 - s0 : 0) : 0x7fffffff));
            else
              tgt = !open_d ? 0x7fffffff : (s1 > S2 ? (s1 > E2 ? s1 - E2 : 0) : (s1 > L0 ? (s1 > E1 ? s1 - E1 : 0) : 0x7fffffff));
            for (int o = 32; o > 0; o >>= 1) {
              const int v = __shfl_xor(tgt, o);
              tgt = v < tgt ? v : tgt;
            }
            tgt = __builtin_amdgcn_readfirstlane(tgt);
            if (tgt != 0x7fffffff && tgt >= 64) {
              const int kk = (tgt - 63 + CC_KNN_TSTRIDE - 1) / CC_KNN_TSTRIDE;  // of the wave's own steps until one reaches that index
              sb += dir == 0 ? CC_KNN_TSTRIDE * kk : -CC_KNN_TSTRIDE * kk;
              CC_KNN_TFETCH()
            }
          }
        }
      }
      }
    }
    if (lane == 0) {
      L.wn[par][wave] = wn;
      L.go[par][wave] = (mine || has_left) ? 1 : 0;
    }
    CC_KNN_TICK(1)
    __syncthreads();
    CC_KNN_TICK(2)
    // ---- the queues: worked off once CC_KNN_TPASS pairs are pending, or when the walk is over
    int cum[CC_KNN_TW + 1];  // queue w holds the pairs cum[w] .. cum[w + 1] of the round's list
    cum[0] = 0;
    int going = 0;
#pragma unroll
    for (int w_ = 0; w_ < CC_KNN_TW; w_++) {
      cum[w_ + 1] = cum[w_] + L.wn[par][w_];
      going |= L.go[par][w_];
    }
    const int tot = cum[CC_KNN_TW];
    const bool walking = going != 0;
    // the first pass comes early (64 pairs): until it has run every search filters with dist_ub, the widest radius it will ever have
    if (tot >= (n_pass_ == 0 ? 64 : CC_KNN_TPASS) || (!walking && tot > 0)) {
      n_pass_++;
      for (int e0 = 0; e0 < tot; e0 += CC_KNN_TPASS) {
        const int e = e0 + tid;
        if (tid < CC_KNN_TPASS && e < tot) {
          int qw = 0;
#pragma unroll
          for (int w_ = 1; w_ < CC_KNN_TW; w_++) qw += e >= cum[w_] ? 1 : 0;
          int qoff = cum[0];
#pragma unroll
          for (int w_ = 1; w_ < CC_KNN_TW; w_++) qoff = qw == w_ ? cum[w_] : qoff;
          const unsigned ent = L.wl[qw][e - qoff];
          const int js = (int)(ent >> 28);
          const unsigned u_ = ent & 0x0FFFFFFFu;
          const int act = sact[u_];
          const int kid = sid[u_];
          float c[CC_KEY_DIM];
#pragma unroll
          for (int d = 0; d < CC_KEY_DIM; d++) c[d] = K[(size_t)d * cap + u_];
          const float *kk = L.qk[js];
          // L2_Adaptor::evalMetric accumulation order (nanoflann.hpp:427-461)
          float r_ = 0.f;
          float d0 = kk[0] - c[0], d1 = kk[1] - c[1], d2 = kk[2] - c[2], d3 = kk[3] - c[3];
          r_ += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          d0 = kk[4] - c[4];
          d1 = kk[5] - c[5];
          d2 = kk[6] - c[6];
          d3 = kk[7] - c[7];
          r_ += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
          d0 = kk[8] - c[8];
          r_ += d0 * d0;
          d0 = kk[9] - c[9];
          r_ += d0 * d0;
          const float ub = L.st[js].ub;
          // before nnk candidates are known a key must be strictly inside dist_ub; afterwards keys AT the nnk-th best
          // distance still compete, on the key id
          const int qq_ = L.qq[js];
          if (act <= L.qb[js][4] && (L.st[js].tight ? (r_ <= ub) : (r_ < ub)) && (qq_ < 0 || cc_knn_key_indexed(qmeta + qq_, ll, c[0]))) {
            const int slot = atomicAdd(&L.st[js].cnt, 1);
            L.buf[js][slot] = ((unsigned long long)__float_as_uint(r_) << 32) | (unsigned)kid;
          }
        }
        __syncthreads();
        if (PH) pc_[7]++;
        CC_KNN_TICK(3)
        // cut back the buffers that filled up: wave w looks after the searches w, w + 4, ...
        const int cnt_l = L.st[j].cnt, tight_l = L.st[j].tight;  // lane (j, kq): search j's
        unsigned long long due = __ballot(kq == 0 && j < ns && (j & (CC_KNN_TW - 1)) == wave && (cnt_l >= CC_KNN_TTRIG || (!tight_l && cnt_l >= nnk)));
        while (due) {
          const int jj = __ffsll((unsigned long long)due) - 1;
          due &= due - 1ull;
          const int cnt = __builtin_amdgcn_readlane(cnt_l, jj);
          int kept;
          const float nub = cc_knn_cut(L.buf[jj], cnt, nnk, lane, kept, L.st[jj].ub);
          if (lane == 0) {
            L.st[jj].ub = nub;
            L.st[jj].cnt = kept;
            L.st[jj].tight = 1;
          }
        }
        __syncthreads();
        CC_KNN_TICK(4)
      }
      wn = 0;
      ubj = L.st[j].ub;
      tightj = L.st[j].tight;
      thr = ubj + slack;
    }
    if (!walking) break;
  }
#undef CC_KNN_TFETCH
  // ---- results: the nnk best by (distance, key id); wave w writes the searches w, w + 4, ...
  for (int jj = wave; jj < ns; jj += CC_KNN_TW) {
    int cnt = __builtin_amdgcn_readfirstlane(L.st[jj].cnt);
    if (cnt > 128) {  // what is within the nnk-th distance fits the smaller sorting network
      int kept;
      cc_knn_cut(L.buf[jj], cnt, nnk, lane, kept, L.st[jj].ub);
      cnt = kept;
    }
    unsigned long long first;
    cc_knn_reduce<2>(L.buf[jj], cnt, nnk, lane, first);
    const int s_ = order_l[base + jj];
    const int q_ = s_ / CC_NPIV, seq_ = s_ - q_ * CC_NPIV;
    const int slot = q_ * (CC_NQLEV * CC_NPIV) + ll * CC_NPIV + seq_;
    const int mm = cnt < nnk ? cnt : nnk;
    if (lane < mm) {
      const unsigned id = (unsigned)(first & 0xFFFFFFFFu);
      cc_knn_hit_t h;
      h.gidx = P.kgidx[ll][id];
      h.level = (int16_t)level;
      h.seq = (int16_t)P.kseq[ll][id];
      h.dist_sq = __uint_as_float((unsigned)(first >> 32));
      hits[(size_t)slot * CC_KNN_MAX + lane] = h;
    }
    if (lane == 0) hit_cnt[slot] = mm;
  }
  CC_KNN_TICK(5)
#undef CC_KNN_TICK
  if (PH && phase_clk && tid == 0)
    for (int i = 0; i < 8; i++) phase_clk[(size_t)blockIdx.x * 8 + i] = pc_[i];
}
