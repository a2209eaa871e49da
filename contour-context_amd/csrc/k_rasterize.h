// K1 -- BEV rasterisation.  Replaces ContourManager::makeBEV (contour_mng.h:505-556) for a batch
// of scans: one workgroup per scan, the 150x150 max-height grid lives in LDS.
//
//   sweep : stream the scan's (x,y,z,i) records ONCE with coalesced 16-B loads, a register-held chunk at a time:
//           LDS atomicMax of the order-preserving height key per cell               (90 KB LDS), then, among the
//           chunk's points whose height equals the cell maximum, keep the smallest point index -- the reference
//           updates a cell only on `bev < height` (strict), so the FIRST point in file order wins ties
//           (contour_mng.h:517).  Indices are 21-bit fields, three per 64-bit LDS word, erased with an atomicOr when
//           the cell's maximum rises and min-updated with a CAS loop                 (60 KB LDS)
//   out   : dense bev image + continuous (row_f,col_f) of the winning point per occupied cell
//           (pointToContRowCol, contour_mng.h:468-472), max/min accepted height, #occupied cells.
//
// Roofline: HBM.  Algorithmic bytes = 16 B x points = what the sweep reads.
#pragma once
#include "cc_dev.h"
#include "cc_group.h"

#define CC_K1_IDX_BITS 21
#define CC_K1_IDX_MASK 0x1FFFFFull
#ifndef CC_K1_U_DEFAULT
#define CC_K1_U_DEFAULT 4  // points per lane and chunk (8 measured equal: the sweep is bound by instruction issue, not by loads in flight)
#endif

// The first-index fields: CC_K1_IDX_BITS bits per cell, three cells per 64-bit LDS word; word w holds the cells w, w + n_w3,
// w + 2 n_w3 (round 6: w held 3 w .. 3 w + 2 -- neighbouring cells are what neighbouring lanes bring, and their CAS
// attempts on one word failed each other).
__device__ __forceinline__ void cc_k1_field(int cell, int n_w3, int &w, int &sh) {
  const int f = (cell >= n_w3 ? 1 : 0) + (cell >= 2 * n_w3 ? 1 : 0);
  w = cell - f * n_w3;
  sh = f * CC_K1_IDX_BITS;
}

#ifdef CC_TUNE_K1_CLK  // tuning aid: where a workgroup's time goes (10-ns ticks summed over the workgroups; printed by cc_destroy)
__device__ unsigned long long cc_k1_clk[8];
#define CC_K1_STAMP(slot)                                                  \
  {                                                                        \
    const long long now_ = (long long)wall_clock64();                      \
    k1_acc_[slot] += now_ - k1_t_;                                         \
    k1_t_ = now_;                                                          \
  }
#else
#define CC_K1_STAMP(slot)
#endif

struct cc_k1_scan_out {
  float max_bin_val, min_bin_val;
  int n_pix;
  int pad;
};

__device__ __forceinline__ float cc_wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o);
    v = v < t ? t : v;
  }
  return v;
}
__device__ __forceinline__ float cc_wave_min(float v) {
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o);
    v = v > t ? t : v;
  }
  return v;
}
__device__ __forceinline__ int cc_wave_sum(int v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Partial results of one point range of a scan (CC_K1_SPLIT ranges per scan when a call brings only a few scans: the
// per-scan loop of the class mirror brings one, and one workgroup sweeping 120 000 points alone lasts ~100 us):
// per cell the height key and the scan-relative index of the first point that reaches it, plus the range's max / min keys.
#define CC_K1_SPLIT 8
#define CC_K1_SPLIT_MAX_SCANS 8   // calls with up to this many scans take the split path
struct cc_k1_part {
  unsigned *key;   // [n_scans * CC_K1_SPLIT][n_cell]
  int *idx;        // same
  unsigned *red;   // [n_scans * CC_K1_SPLIT][2]: max key, min key
};

// The scan's ACTIVE cells (above the lowest level) as a raster-ordered list -- what K2's list kernel (k_contours_list.h) starts
// from: K1 has every cell's height in LDS when it writes the dense image, so it lists the active ones on the way out
// instead of K2 re-reading 90 KB of image to find the ~2 500 cells it wants (round 6).  Per scan: header (entries, (cell,
// level) slots), and for the first CC_LIST_CAP entries (row << 8 | col), level count, height, continuous position.
#define CC_LIST_CAP 3072
#define CC_K1_NCHUNK ((CC_MAX_CELLS + 63) / 64)
struct cc_k1_list_out {
  int4 *hdr;            // [n_scans]: x = entries (all of them, also beyond the capacity), y = slots = sum of the level counts, z = the dense image / positions were written
  uint16_t *rc;         // [n_scans][CC_LIST_CAP]
  unsigned char *lev;   // same
  float *h;             // same
  float2 *pix;          // same
};
#define CC_K1_EB 8  // cells per thread whose records cc_k1_emit has in flight together
#define CC_K1_LB 4  // list entries per thread whose records it has in flight together
#define CC_K1_EMIT_TAB_BYTES (CC_K1_NCHUNK * 2 * 3 + 16)  // u16 entries per chunk | u16 entries before the chunk | u16 slots per chunk | totals
#define CC_K1_EMIT_LDS_BYTES (CC_K1_EMIT_TAB_BYTES + CC_LIST_CAP * 2)  // ... | u16 cell of every list entry

// The output pass shared by the one-sweep kernel and the merge kernel.  keyfn(c) / idxfn(c): the cell's height key and the
// scan-relative index of the point that owns it (asked for occupied cells only).  Two sweeps over the cells, a wave
// on 64 consecutive cells at a time: (1) active cells per chunk (one ballot), prefix by wave 0; (2) the
// dense image, the continuous position of every occupied cell, and the list entries at their raster-order positions.
// Returns this thread's count of occupied cells.  tab: CC_K1_EMIT_LDS_BYTES of LDS.
template <typename KeyFn, typename IdxFn>
__device__ __forceinline__ int cc_k1_emit(const cc_dev_cfg &cfg, KeyFn keyfn, IdxFn idxfn, const float4 *__restrict__ P, float *__restrict__ bev,
                                          float2 *__restrict__ pix, const cc_k1_list_out &L, int scan, char *tab, int n_pts, int want_dense, unsigned *kmax_out /*LDS: the largest cell key is max-ed into it (nullptr: not wanted)*/) {
  const int n_cell = cfg.n_cell, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const unsigned KEY_EMPTY = cc_fkey(CC_BEV_EMPTY);
  uint16_t *ccnt = (uint16_t *)tab, *cbase = ccnt + CC_K1_NCHUNK;
  int *tot = (int *)(tab + CC_K1_NCHUNK * 6);
  const int n_chunk = (n_cell + 63) >> 6;
  // (1) active cells per 64-cell chunk: one compare and one ballot per cell (a wave is on 64 consecutive cells)
  unsigned kmx = KEY_EMPTY;
  for (int c0 = 0; c0 < n_cell; c0 += nt) {  // block-uniform trip count: the ballots see whole waves
    const int c = c0 + tid;
    const unsigned k = c < n_cell ? keyfn(c) : KEY_EMPTY;
    kmx = k > kmx ? k : kmx;
    const float h = cc_funkey(k);
    const unsigned long long m0 = __ballot(h > cfg.lv_grads[0]);
    if (lane == 0 && c < n_cell) ccnt[c >> 6] = (uint16_t)__popcll(m0);
  }
  if (kmax_out) {
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned a = (unsigned)__shfl_xor((int)kmx, o);
      kmx = a > kmx ? a : kmx;
    }
    if (lane == 0) atomicMax(kmax_out, kmx);
  }
  if (tid == 0) tot[1] = 0;
  __syncthreads();
  if (tid < 64) {  // prefix over the chunks, one wave
    int n_act = 0;
    for (int q = 0; q < n_chunk; q += 64) {
      const int b = q + lane;
      const int v1 = b < n_chunk ? (int)ccnt[b] : 0;
      const int i1 = cc_wave_scan_incl(v1);
      if (b < n_chunk) cbase[b] = (uint16_t)(n_act + i1 - v1);
      n_act += cc_wave_scan_total(i1);
    }
    if (lane == 0) tot[0] = n_act;
  }
  __syncthreads();
  // The dense image and the dense position array are written when somebody reads them: the caller asked for them
  // (want_dense: debug outputs, cc_scan_bev, a configuration K2's list kernel does not take) or the list cannot hold the
  // scan's active cells.  Otherwise the list is all K2 needs -- 270 KB per scan that nobody read were a quarter of this
  // kernel's time (round 6) -- and cc_k_contours_mid rebuilds the two arrays from the list for a scan the list kernel hands
  // on (hdr.z says which it is).
  const bool dense = want_dense != 0 || tot[0] > CC_LIST_CAP;
  uint16_t *l_rc = L.rc + (size_t)scan * CC_LIST_CAP;
  unsigned char *l_lev = L.lev + (size_t)scan * CC_LIST_CAP;
  float *l_h = L.h + (size_t)scan * CC_LIST_CAP;
  float2 *l_pix = L.pix + (size_t)scan * CC_LIST_CAP;
  int npix = 0, nslot = 0;
  if (!dense) {
    // (2') the usual case -- only the list is wanted: the active cells' indices go to LDS in raster order (one more sweep over
    // the keys), then every thread takes list entries i, i + nt, ...: all lanes busy, the owners' records requested together,
    // the entries stored side by side (round 6: the dense sweep below kept 22 cells per thread for ~2.4 of them)
    uint16_t *acell = (uint16_t *)(tab + CC_K1_EMIT_TAB_BYTES);  // [CC_LIST_CAP]
    for (int c0 = 0; c0 < n_cell; c0 += nt) {  // block-uniform trip count
      const int c = c0 + tid;
      const unsigned k = c < n_cell ? keyfn(c) : KEY_EMPTY;
      const bool act = cc_funkey(k) > cfg.lv_grads[0];  // cv::threshold BINARY is strict `>` (contour_mng.cpp:283)
      const unsigned long long m0 = __ballot(act);
      npix += k != KEY_EMPTY ? 1 : 0;
      if (act) acell[(int)cbase[c >> 6] + cc_mbcnt(m0)] = (uint16_t)c;  // < CC_LIST_CAP: the list holds the scan
    }
    __syncthreads();
    const int n_act = tot[0];
    for (int i0 = 0; i0 < n_act; i0 += nt * CC_K1_LB) {
      int cc[CC_K1_LB];
      unsigned key[CC_K1_LB];
      float2 xy[CC_K1_LB];
#pragma unroll
      for (int e = 0; e < CC_K1_LB; e++) {
        const int i = i0 + e * nt + tid;
        cc[e] = i < n_act ? (int)acell[i] : -1;
        key[e] = cc[e] >= 0 ? keyfn(cc[e]) : KEY_EMPTY;
      }
#pragma unroll
#ifdef CC_TUNE_K1_NOB
      for (int e = 0; e < CC_K1_LB; e++) xy[e] = *(const float2 *)(P + (cc[e] >= 0 ? idxfn(cc[e]) % n_pts : 0));
#else
      for (int e = 0; e < CC_K1_LB; e++) xy[e] = *(const float2 *)(P + (cc[e] >= 0 ? idxfn(cc[e]) : 0));  // (an active cell has an owner: n_pts > 0)
#endif
#pragma unroll
      for (int e = 0; e < CC_K1_LB; e++) {
        const int i = i0 + e * nt + tid, c = cc[e];
        if (c >= 0) {
          const float h = cc_funkey(key[e]);
          float2 rcf;  // pointToContRowCol, as below
          rcf.x = (cfg.reso_pow2 ? xy[e].x * cfg.inv_row : xy[e].x / cfg.reso_row) + (float)cfg.half_row - 0.5f;
          rcf.y = (cfg.reso_pow2 ? xy[e].y * cfg.inv_col : xy[e].y / cfg.reso_col) + (float)cfg.half_col - 0.5f;
          int lv = 1;
          for (int l = 1; l < CC_NLEV; l++) lv += (h > cfg.lv_grads[l]) ? 1 : 0;
          nslot += lv;
          const int r = c / cfg.n_col;
          l_rc[i] = (uint16_t)((r << 8) | (c - r * cfg.n_col));
          l_lev[i] = (unsigned char)lv;
          l_h[i] = h;
          l_pix[i] = rcf;
        }
      }
    }
  } else {
  // (2) the dense image, the continuous position of every occupied cell, the list entries at their raster-order positions.
  // CC_K1_EB cells per thread at a time: their keys and owners first, the owners' records requested TOGETHER, then the
  // outputs -- one cell at a time every thread waited for its record 22 times in a row, a quarter of the kernel (round 6,
  // -DCC_TUNE_K1_CLK).  A cell without an owner asks for the scan's first record and drops it.
  for (int c0 = 0; c0 < n_cell; c0 += nt * CC_K1_EB) {  // block-uniform trip counts: the ballots see whole waves
    unsigned key[CC_K1_EB];
    float2 xy[CC_K1_EB];
#pragma unroll
    for (int e = 0; e < CC_K1_EB; e++) {
      const int c = c0 + e * nt + tid;
      key[e] = c < n_cell ? keyfn(c) : KEY_EMPTY;
    }
#pragma unroll
    for (int e = 0; e < CC_K1_EB; e++) {
      const int c = c0 + e * nt + tid;
      const bool need = dense ? key[e] != KEY_EMPTY : cc_funkey(key[e]) > cfg.lv_grads[0];
      const int own = need ? idxfn(c) : 0;
      xy[e] = n_pts > 0 ? *(const float2 *)(P + own) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < CC_K1_EB; e++) {
      const int cb = c0 + e * nt;  // block-uniform
      if (cb >= n_cell) break;
      const int c = cb + tid;
      const float h = cc_funkey(key[e]);
      const bool act = h > cfg.lv_grads[0];  // cv::threshold BINARY is strict `>` (contour_mng.cpp:283)
      const unsigned long long m0 = __ballot(act);
      if (c < n_cell) {
        if (dense) bev[c] = h;
        if (key[e] != KEY_EMPTY) {
          // pointToContRowCol: x / reso + n_row/2 - 0.5f, left to right in f32 (a power-of-two resolution: the product with
          // its reciprocal is the same correctly rounded value)
          float2 rcf;
          rcf.x = (cfg.reso_pow2 ? xy[e].x * cfg.inv_row : xy[e].x / cfg.reso_row) + (float)cfg.half_row - 0.5f;
          rcf.y = (cfg.reso_pow2 ? xy[e].y * cfg.inv_col : xy[e].y / cfg.reso_col) + (float)cfg.half_col - 0.5f;
          if (dense) pix[c] = rcf;
          npix++;
          if (act) {
            int lv = 1;
            for (int l = 1; l < CC_NLEV; l++) lv += (h > cfg.lv_grads[l]) ? 1 : 0;
            nslot += lv;
            const int i = (int)cbase[c >> 6] + cc_mbcnt(m0);
            if (i < CC_LIST_CAP) {
              const int r = c / cfg.n_col;
              l_rc[i] = (uint16_t)((r << 8) | (c - r * cfg.n_col));
              l_lev[i] = (unsigned char)lv;
              l_h[i] = h;
              l_pix[i] = rcf;
            }
          }
        }
      }
    }
  }
  }
  nslot = cc_wave_sum(nslot);
  if (lane == 0 && nslot) atomicAdd(&tot[1], nslot);
  __syncthreads();
  if (tid == 0) L.hdr[scan] = make_int4(tot[0], tot[1], dense ? 1 : 0, 0);
  return npix;
}

// grid = n_scans (PART: n_scans * CC_K1_SPLIT), block = multiple of 64.  dynamic LDS: n_cell*4 + ((n_cell+2)/3)*8 + 16 + CC_K1_EMIT_LDS_BYTES bytes.
template <int CC_K1_U, bool CC_K1_POW2, bool PART = false>
__global__ void __launch_bounds__(1024)
cc_k_rasterize(cc_dev_cfg cfg, const float4 *__restrict__ pts, const long long *__restrict__ offsets,
               float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out, cc_k1_part part, cc_k1_list_out list_out,
               int want_dense, int n_units /*scans (PART: scans * CC_K1_SPLIT); workgroup b takes units b, b + gridDim.x, ...*/) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int n_cell = cfg.n_cell;
  unsigned *hmax = (unsigned *)smem;
  const int n_w3 = (n_cell + 2) / 3;
  unsigned long long *idx3 = (unsigned long long *)(smem + (((size_t)n_cell * 4 + 15) & ~(size_t)15));
  unsigned *red = (unsigned *)(idx3 + n_w3);  // [0]=max key [1]=min key [2]=n_pix
  char *emit_tab = (char *)(red + 4);          // CC_K1_EMIT_LDS_BYTES: the output pass' chunk tables
  unsigned *idle = (unsigned *)(emit_tab + CC_K1_EMIT_TAB_BYTES);  // [blockDim]: where a lane with nothing to send aims its atomicMax (the sweep's; the output pass has its cell list there)
  static_assert(CC_LIST_CAP * 2 >= 4 * 1024 && CC_K1_EMIT_TAB_BYTES % 4 == 0, "cc_k_rasterize: the idle words fit the list's cells");

  const int tid = threadIdx.x, nt = blockDim.x;
#ifdef CC_TUNE_K1_CLK
  long long k1_t_ = 0, k1_acc_[6] = {0, 0, 0, 0, 0, 0};
#endif
  // A workgroup takes units b, b + gridDim.x, ...: the host normally launches one per unit; CC_K1_WGS brings fewer, each
  // keeping its CU (a K1 workgroup needs one to itself) for several scans -- measured: K1's own in-step time falls, K2's rises
  // by as much (profiles/r6/notes_negative_results.md).
  for (int unit = (int)blockIdx.x; unit < n_units; unit += (int)gridDim.x) {
  if (unit != (int)blockIdx.x) __syncthreads();  // the previous scan's output pass has read the grid
  const int scan = PART ? unit / CC_K1_SPLIT : unit;
  long long p0 = offsets[scan];
  int n_pts = (int)(offsets[scan + 1] - p0);
  int idx_base = 0;  // scan-relative index of this workgroup's first point
  if (PART) {
    const int per = (n_pts + CC_K1_SPLIT - 1) / CC_K1_SPLIT, pi = unit % CC_K1_SPLIT;
    idx_base = pi * per < n_pts ? pi * per : n_pts;
    n_pts = n_pts - idx_base < per ? n_pts - idx_base : per;
    p0 += idx_base;
  }
  const float4 *P = pts + p0;

#ifdef CC_TUNE_K1_CLK
  k1_t_ = (long long)wall_clock64();
#endif
  const unsigned KEY_EMPTY = cc_fkey(CC_BEV_EMPTY);
  for (int i = tid; i < n_cell; i += nt) hmax[i] = KEY_EMPTY;
  for (int i = tid; i < n_w3; i += nt) idx3[i] = ~0ull;
  idle[tid] = 0u;
  if (tid == 0) {
    red[0] = cc_fkey(CC_BEV_EMPTY);   // max_bin_val_ starts at -VAL_ABS_INF_ (contour_mng.h:436)
    red[1] = cc_fkey(-CC_BEV_EMPTY);  // min_bin_val_ starts at +VAL_ABS_INF_
    red[2] = 0;
  }
  __syncthreads();
  CC_K1_STAMP(0)

  // ---- one sweep over the stream, in chunks of CC_K1_U * blockDim points held in registers ----
  // step A (all lanes): atomicMax of the chunk's heights; a point that RAISES a cell's maximum erases the cell's
  //                     index field (the index recorded so far belongs to a lower height)
  // step B (after a barrier): the chunk's points that equal the cell maximum min-reduce their index into the field.
  // A cell whose maximum dates from an earlier chunk keeps that (smaller) index: later equal heights never replace
  // it, which is the strict `bev < height` update of contour_mng.h:517.  The barrier after step B keeps the next
  // chunk's erasures behind this chunk's index updates.
  // min accepted height as a key (the map keeps the order); the max accepted height is the largest cell maximum: taken from
  // the grid on the way out
  unsigned kmin = cc_fkey(-CC_BEV_EMPTY);
  const int chunk = CC_K1_U * nt;
  // The records are loaded UNCONDITIONALLY from an index clamped to the scan's last point and a lane past the end drops
  // its point when it uses it: a load under a branch is waited for where the branch ends (round 6: the prefetch below was
  // no prefetch for two of the four loads).
  const int last = n_pts > 0 ? n_pts - 1 : 0;
  float4 q[CC_K1_U];
#pragma unroll
  for (int u = 0; u < CC_K1_U; u++) q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n_pts > 0) {
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      const int j = tid + u * nt;
      q[u] = P[j < last ? j : last];
    }
  }
  for (int base = 0; base < n_pts; base += chunk) {
    int cell[CC_K1_U];
    unsigned key[CC_K1_U];
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      const int c = cc_point_cell<CC_K1_POW2>(cfg, q[u].x, q[u].y);
      const float h = cfg.lidar_height + q[u].z;
      key[u] = cc_fkey(h);
      // a NaN height never updates a cell or the max/min in the reference (`bev < NaN`, `max < NaN`, `min > NaN` are
      // all false, contour_mng.h:517-524): such a point is dropped here
      cell[u] = ((h == h) & (tid + u * nt < n_pts - base)) ? c : -1;
      const unsigned kb = cell[u] >= 0 ? key[u] : 0xFFFFFFFFu;
      kmin = kb < kmin ? kb : kmin;
    }
    // the next chunk's records travel while this chunk is resolved in LDS
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      const int j = base + chunk + tid + u * nt;
      q[u] = P[j < last ? j : last];
    }
    // Consecutive records are neighbouring azimuth steps of one laser: close to the sensor dozens of them fall into
    // the same cell, and same-address LDS atomics of a wave are served one after the other.  So the lanes of a 16-lane
    // row first combine their heights per run of equal cells (segmented max over DPP row shifts), and only the last
    // lane of a run goes to the LDS, with the run's maximum.  Which lanes continue their left neighbour's run is ONE wave
    // mask; the masks of the wider steps ("the 2, 4, 8 lanes to my left are in my run") and the senders' come from it with
    // scalar shifts (round 6: a compare of shifted cells per step before) -- a step is a DPP max and a select.
    unsigned kr[CC_K1_U], was[CC_K1_U];
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      const int c1 = cell[u] + 1;  // 0 = rejected point (and what a row shift reads beyond the row's end: a run ends at its row's end)
      unsigned k = cell[u] >= 0 ? key[u] : 0u;  // (rejected lanes form runs of their own, of zeros)
      const unsigned long long m1 = __ballot(cc_row_shr<1>(c1) == c1);
      const unsigned long long m2 = m1 & (m1 << 1), m4 = m2 & (m2 << 2), m8 = m4 & (m4 << 4);
      {
        const unsigned ok = (unsigned)cc_row_shr<1>((int)k);
        k = cc_mask_lane(m1) ? (ok > k ? ok : k) : k;
      }
      {
        const unsigned ok = (unsigned)cc_row_shr<2>((int)k);
        k = cc_mask_lane(m2) ? (ok > k ? ok : k) : k;
      }
      {
        const unsigned ok = (unsigned)cc_row_shr<4>((int)k);
        k = cc_mask_lane(m4) ? (ok > k ? ok : k) : k;
      }
      {
        const unsigned ok = (unsigned)cc_row_shr<8>((int)k);
        k = cc_mask_lane(m8) ? (ok > k ? ok : k) : k;
      }
      const unsigned long long last_of_run = ~(m1 >> 1) | 0x8000800080008000ull;
      kr[u] = cc_mask_lane(last_of_run) ? k : 0u;  // 0: this lane sends nothing (no height maps to key 0; a rejected lane holds 0)
    }
    // the chunk's atomics leave together (round 6: one after the other, each waited for, they were four LDS round trips)
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) was[u] = atomicMax(kr[u] ? &hmax[cell[u]] : &idle[tid], kr[u]);
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++)
      if (was[u] < kr[u]) {
        int w, sh;
        cc_k1_field(cell[u], n_w3, w, sh);
        atomicOr(&idx3[w], CC_K1_IDX_MASK << sh);
      }
    CC_K1_STAMP(1)
    __syncthreads();
    CC_K1_STAMP(2)
    // step B: which of this lane's points hold their cell's maximum (four reads in flight), then ONE loop in which a lane
    // works off its winners one CAS attempt per turn -- a retry and the next winner's first attempt share a turn
    unsigned pend = 0u;
#ifndef CC_TUNE_K1_NOB  // (tuning aid: the sweep without its index pass -- wrong owners, for timing only)
    {
      unsigned hm[CC_K1_U];
#pragma unroll
      for (int u = 0; u < CC_K1_U; u++) hm[u] = hmax[cell[u] >= 0 ? cell[u] : 0];
#pragma unroll
      for (int u = 0; u < CC_K1_U; u++) pend |= (cell[u] >= 0 && key[u] == hm[u] && key[u] != KEY_EMPTY) ? (1u << u) : 0u;
    }
#endif
    {
      bool busy = false;
      int w = 0, sh = 0;
      unsigned long long j = 0ull, old = 0ull;
      while (pend || busy) {
        if (!busy) {
          const int u = __ffs(pend) - 1;
          pend &= pend - 1u;
          int c = cell[0];
#pragma unroll
          for (int v = 1; v < CC_K1_U; v++) c = u == v ? cell[v] : c;
          cc_k1_field(c, n_w3, w, sh);
          j = (unsigned long long)(idx_base + base + tid + u * nt);
          old = idx3[w];
          busy = true;
        }
        const unsigned long long cur = (old >> sh) & CC_K1_IDX_MASK;
        if (j >= cur) {
          busy = false;
        } else {
          const unsigned long long nw = (old & ~(CC_K1_IDX_MASK << sh)) | (j << sh);
          const unsigned long long got = atomicCAS(&idx3[w], old, nw);
          busy = got != old;
          old = got;
        }
      }
    }
    CC_K1_STAMP(3)
    __syncthreads();
    CC_K1_STAMP(4)
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned b2 = (unsigned)__shfl_xor((int)kmin, o);
    kmin = b2 < kmin ? b2 : kmin;
  }
  if ((tid & 63) == 0) atomicMin(&red[1], kmin);
  __syncthreads();

  if (PART) {  // this range's grid to the scratch; cc_k_rasterize_merge combines the ranges
    unsigned *pk = part.key + (size_t)unit * n_cell;
    int *pj = part.idx + (size_t)unit * n_cell;
    unsigned kmx = KEY_EMPTY;
    for (int c = tid; c < n_cell; c += nt) {
      const unsigned k = hmax[c];
      kmx = k > kmx ? k : kmx;
      pk[c] = k;
      int w, sh;
      cc_k1_field(c, n_w3, w, sh);
      pj[c] = k != KEY_EMPTY ? (int)((idx3[w] >> sh) & CC_K1_IDX_MASK) : -1;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned a = (unsigned)__shfl_xor((int)kmx, o);
      kmx = a > kmx ? a : kmx;
    }
    if ((tid & 63) == 0) atomicMax(&red[0], kmx);
    __syncthreads();
    if (tid == 0) {
      part.red[(size_t)unit * 2] = red[0];
      part.red[(size_t)unit * 2 + 1] = red[1];
    }
    continue;
  }
  // ---- outputs ----
  float *bev = bev_out + (size_t)scan * n_cell;
  float2 *pix = pix_out + (size_t)scan * n_cell;
  int npix = cc_k1_emit(
      cfg,
      [&](int c) { return hmax[c]; },
      [&](int c) {
        int w, sh;
        cc_k1_field(c, n_w3, w, sh);
        return (int)((idx3[w] >> sh) & CC_K1_IDX_MASK);
      },
      P, bev, pix, list_out, scan, emit_tab, n_pts, want_dense, &red[0]);
  npix = cc_wave_sum(npix);
  if ((tid & 63) == 0) atomicAdd(&red[2], (unsigned)npix);
  __syncthreads();
  if (tid == 0) {
    cc_k1_scan_out o;
    o.max_bin_val = cc_funkey(red[0]);
    o.min_bin_val = cc_funkey(red[1]);
    o.n_pix = (int)red[2];
    o.pad = 0;
    scan_out[scan] = o;
  }
  CC_K1_STAMP(5)
  }  // units of this workgroup
#ifdef CC_TUNE_K1_CLK
  if (tid == 0)
    for (int i = 0; i < 6; i++) atomicAdd(&cc_k1_clk[i], (unsigned long long)k1_acc_[i]);
#endif
}

// The ranges of a scan combined: a cell's height is the largest of the ranges' keys and its point the one of the FIRST
// range that reaches it -- ranges are in file order and each holds the first of its own points, so this is the first point
// of the scan at that height: the reference's strict `bev < height` update (contour_mng.h:517) as in the one-sweep kernel.
// grid = n_scans, block = multiple of 64
__global__ void __launch_bounds__(1024)
cc_k_rasterize_merge(cc_dev_cfg cfg, const float4 *__restrict__ pts, const long long *__restrict__ offsets, cc_k1_part part,
                     float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out, cc_k1_list_out list_out, int want_dense) {
  __shared__ unsigned red[3];
  __shared__ __attribute__((aligned(16))) char emit_tab[CC_K1_EMIT_LDS_BYTES];
  const int scan = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, n_cell = cfg.n_cell;
  const unsigned KEY_EMPTY = cc_fkey(CC_BEV_EMPTY);
  const float4 *P = pts + offsets[scan];
  if (tid == 0) {
    unsigned mx = cc_fkey(CC_BEV_EMPTY), mn = cc_fkey(-CC_BEV_EMPTY);
    for (int p = 0; p < CC_K1_SPLIT; p++) {
      const unsigned a = part.red[((size_t)scan * CC_K1_SPLIT + p) * 2], b = part.red[((size_t)scan * CC_K1_SPLIT + p) * 2 + 1];
      mx = a > mx ? a : mx;
      mn = b < mn ? b : mn;
    }
    red[0] = mx;
    red[1] = mn;
    red[2] = 0;
  }
  __syncthreads();
  float *bev = bev_out + (size_t)scan * n_cell;
  float2 *pix = pix_out + (size_t)scan * n_cell;
  int npix = cc_k1_emit(
      cfg,
      [&](int c) {  // the largest of the ranges' keys
        unsigned best = KEY_EMPTY;
#pragma unroll
        for (int p = 0; p < CC_K1_SPLIT; p++) {
          const unsigned k = part.key[((size_t)scan * CC_K1_SPLIT + p) * n_cell + c];
          best = (k != KEY_EMPTY && (best == KEY_EMPTY || k > best)) ? k : best;
        }
        return best;
      },
      [&](int c) {  // ... and among equals the FIRST range's point (ranges are in file order)
        unsigned best = KEY_EMPTY;
        int bp = 0;
#pragma unroll
        for (int p = 0; p < CC_K1_SPLIT; p++) {
          const unsigned k = part.key[((size_t)scan * CC_K1_SPLIT + p) * n_cell + c];
          if (k != KEY_EMPTY && (best == KEY_EMPTY || k > best)) {
            best = k;
            bp = p;
          }
        }
        return part.idx[((size_t)scan * CC_K1_SPLIT + bp) * n_cell + c];
      },
      P, bev, pix, list_out, scan, emit_tab, (int)(offsets[scan + 1] - offsets[scan]), want_dense, nullptr);
  npix = cc_wave_sum(npix);
  if ((tid & 63) == 0) atomicAdd(&red[2], (unsigned)npix);
  __syncthreads();
  if (tid == 0) {
    cc_k1_scan_out o;
    o.max_bin_val = cc_funkey(red[0]);
    o.min_bin_val = cc_funkey(red[1]);
    o.n_pix = (int)red[2];
    o.pad = 0;
    scan_out[scan] = o;
  }
}
