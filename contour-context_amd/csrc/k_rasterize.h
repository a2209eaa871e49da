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
#define CC_K1_U_DEFAULT 4  // points per lane and chunk (8 measured equal: the sweep is bound by instruction issue, not by loads in flight)

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
  int4 *hdr;            // [n_scans]: x = entries (all of them, also beyond the capacity), y = slots = sum of the level counts
  uint16_t *rc;         // [n_scans][CC_LIST_CAP]
  unsigned char *lev;   // same
  float *h;             // same
  float2 *pix;          // same
};
#define CC_K1_EMIT_LDS_BYTES (CC_K1_NCHUNK * 2 * 3 + 16)  // u16 entries per chunk | u16 entries before the chunk | u16 slots per chunk | totals

// The output pass shared by the one-sweep kernel and the merge kernel.  keyfn(c) / idxfn(c): the cell's height key and the
// scan-relative index of the point that owns it (asked for occupied cells only).  Two sweeps over the cells, a wave
// on 64 consecutive cells at a time: (1) active cells per chunk (one ballot), prefix by wave 0; (2) the
// dense image, the continuous position of every occupied cell, and the list entries at their raster-order positions.
// Returns this thread's count of occupied cells.  tab: CC_K1_EMIT_LDS_BYTES of LDS.
template <typename KeyFn, typename IdxFn>
__device__ __forceinline__ int cc_k1_emit(const cc_dev_cfg &cfg, KeyFn keyfn, IdxFn idxfn, const float4 *__restrict__ P, float *__restrict__ bev,
                                          float2 *__restrict__ pix, const cc_k1_list_out &L, int scan, char *tab) {
  const int n_cell = cfg.n_cell, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const unsigned KEY_EMPTY = cc_fkey(CC_BEV_EMPTY);
  uint16_t *ccnt = (uint16_t *)tab, *cbase = ccnt + CC_K1_NCHUNK;
  int *tot = (int *)(tab + CC_K1_NCHUNK * 6);
  const int n_chunk = (n_cell + 63) >> 6;
  // (1) active cells per 64-cell chunk: one compare and one ballot per cell (a wave is on 64 consecutive cells)
  for (int c0 = 0; c0 < n_cell; c0 += nt) {  // block-uniform trip count: the ballots see whole waves
    const int c = c0 + tid;
    const float h = c < n_cell ? cc_funkey(keyfn(c)) : CC_BEV_EMPTY;
    const unsigned long long m0 = __ballot(h > cfg.lv_grads[0]);
    if (lane == 0 && c < n_cell) ccnt[c >> 6] = (uint16_t)__popcll(m0);
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
  uint16_t *l_rc = L.rc + (size_t)scan * CC_LIST_CAP;
  unsigned char *l_lev = L.lev + (size_t)scan * CC_LIST_CAP;
  float *l_h = L.h + (size_t)scan * CC_LIST_CAP;
  float2 *l_pix = L.pix + (size_t)scan * CC_LIST_CAP;
  int npix = 0, nslot = 0;
  // (2) the dense image, the continuous position of every occupied cell, the list entries at their raster-order positions
  for (int c0 = 0; c0 < n_cell; c0 += nt) {
    const int c = c0 + tid;
    const unsigned key = c < n_cell ? keyfn(c) : KEY_EMPTY;
    const float h = cc_funkey(key);
    const bool act = h > cfg.lv_grads[0];  // cv::threshold BINARY is strict `>` (contour_mng.cpp:283)
    const unsigned long long m0 = __ballot(act);
    if (c < n_cell) {
      bev[c] = h;
      if (key != KEY_EMPTY) {
        const float4 q = P[idxfn(c)];
        // pointToContRowCol: x / reso + n_row/2 - 0.5f, left to right in f32
        float2 rcf;
        rcf.x = q.x / cfg.reso_row + (float)cfg.half_row - 0.5f;
        rcf.y = q.y / cfg.reso_col + (float)cfg.half_col - 0.5f;
        pix[c] = rcf;
        npix++;
        if (act) {
          int lv = 1;
          for (int e = 1; e < CC_NLEV; e++) lv += (h > cfg.lv_grads[e]) ? 1 : 0;
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
  nslot = cc_wave_sum(nslot);
  if (lane == 0 && nslot) atomicAdd(&tot[1], nslot);
  __syncthreads();
  if (tid == 0) L.hdr[scan] = make_int4(tot[0], tot[1], 0, 0);
  return npix;
}

// grid = n_scans (PART: n_scans * CC_K1_SPLIT), block = multiple of 64.  dynamic LDS: n_cell*4 + ((n_cell+2)/3)*8 + 16 + CC_K1_EMIT_LDS_BYTES bytes.
template <int CC_K1_U, bool CC_K1_POW2, bool PART = false>
__global__ void __launch_bounds__(1024)
cc_k_rasterize(cc_dev_cfg cfg, const float4 *__restrict__ pts, const long long *__restrict__ offsets,
               float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out, cc_k1_part part, cc_k1_list_out list_out) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int n_cell = cfg.n_cell;
  unsigned *hmax = (unsigned *)smem;
  const int n_w3 = (n_cell + 2) / 3;
  unsigned long long *idx3 = (unsigned long long *)(smem + (((size_t)n_cell * 4 + 15) & ~(size_t)15));
  unsigned *red = (unsigned *)(idx3 + n_w3);  // [0]=max key [1]=min key [2]=n_pix
  char *emit_tab = (char *)(red + 4);          // CC_K1_EMIT_LDS_BYTES: the output pass' chunk tables

  const int scan = PART ? (int)blockIdx.x / CC_K1_SPLIT : (int)blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  long long p0 = offsets[scan];
  int n_pts = (int)(offsets[scan + 1] - p0);
  int idx_base = 0;  // scan-relative index of this workgroup's first point
  if (PART) {
    const int per = (n_pts + CC_K1_SPLIT - 1) / CC_K1_SPLIT, pi = (int)blockIdx.x % CC_K1_SPLIT;
    idx_base = pi * per < n_pts ? pi * per : n_pts;
    n_pts = n_pts - idx_base < per ? n_pts - idx_base : per;
    p0 += idx_base;
  }
  const float4 *P = pts + p0;

  const unsigned KEY_EMPTY = cc_fkey(CC_BEV_EMPTY);
  for (int i = tid; i < n_cell; i += nt) hmax[i] = KEY_EMPTY;
  for (int i = tid; i < n_w3; i += nt) idx3[i] = ~0ull;
  if (tid == 0) {
    red[0] = cc_fkey(CC_BEV_EMPTY);   // max_bin_val_ starts at -VAL_ABS_INF_ (contour_mng.h:436)
    red[1] = cc_fkey(-CC_BEV_EMPTY);  // min_bin_val_ starts at +VAL_ABS_INF_
    red[2] = 0;
  }
  __syncthreads();

  // ---- one sweep over the stream, in chunks of CC_K1_U * blockDim points held in registers ----
  // step A (all lanes): atomicMax of the chunk's heights; a point that RAISES a cell's maximum erases the cell's
  //                     index field (the index recorded so far belongs to a lower height)
  // step B (after a barrier): the chunk's points that equal the cell maximum min-reduce their index into the field.
  // A cell whose maximum dates from an earlier chunk keeps that (smaller) index: later equal heights never replace
  // it, which is the strict `bev < height` update of contour_mng.h:517.  The barrier after step B keeps the next
  // chunk's erasures behind this chunk's index updates.
  float vmax = CC_BEV_EMPTY, vmin = -CC_BEV_EMPTY;
  const int chunk = CC_K1_U * nt;
  float4 q[CC_K1_U];
#pragma unroll
  for (int u = 0; u < CC_K1_U; u++) {
    int j = tid + u * nt;
    q[u] = (j < n_pts) ? P[j] : make_float4(1e9f, 1e9f, 0.f, 0.f);
  }
  for (int base = 0; base < n_pts; base += chunk) {
    int cell[CC_K1_U];
    unsigned key[CC_K1_U];
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      cell[u] = cc_point_cell<CC_K1_POW2>(cfg, q[u].x, q[u].y);
      float h = cfg.lidar_height + q[u].z;
      key[u] = cc_fkey(h);
      // a NaN height never updates a cell or the max/min in the reference (`bev < NaN`, `max < NaN`, `min > NaN` are
      // all false, contour_mng.h:517-524): such a point is dropped here
      if (!(h == h)) cell[u] = -1;
      if (cell[u] >= 0) {
        vmax = vmax < h ? h : vmax;
        vmin = vmin > h ? h : vmin;
      }
    }
    // the next chunk's records travel while this chunk is resolved in LDS
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      int j = base + chunk + tid + u * nt;
      q[u] = (j < n_pts) ? P[j] : make_float4(1e9f, 1e9f, 0.f, 0.f);
    }
    // Consecutive records are neighbouring azimuth steps of one laser: close to the sensor dozens of them fall into
    // the same cell, and same-address LDS atomics of a wave are served one after the other.  So the lanes of a 16-lane
    // row first combine their heights per run of equal cells (segmented max over DPP row shifts), and only the last
    // lane of a run goes to the LDS, with the run's maximum.
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      const int c1 = cell[u] + 1;  // 0 = rejected point (and what a row shift reads beyond the row's end)
      unsigned k = cell[u] >= 0 ? key[u] : 0u;
      {
        const int oc = cc_row_shr<1>(c1);
        const unsigned ok = (unsigned)cc_row_shr<1>((int)k);
        if (oc == c1 && ok > k) k = ok;
      }
      {
        const int oc = cc_row_shr<2>(c1);
        const unsigned ok = (unsigned)cc_row_shr<2>((int)k);
        if (oc == c1 && ok > k) k = ok;
      }
      {
        const int oc = cc_row_shr<4>(c1);
        const unsigned ok = (unsigned)cc_row_shr<4>((int)k);
        if (oc == c1 && ok > k) k = ok;
      }
      {
        const int oc = cc_row_shr<8>(c1);
        const unsigned ok = (unsigned)cc_row_shr<8>((int)k);
        if (oc == c1 && ok > k) k = ok;
      }
      const bool last_of_run = cc_row_shl1(c1) != c1;
      if (cell[u] >= 0 && last_of_run) {
        unsigned old = atomicMax(&hmax[cell[u]], k);
        if (old < k) {
          const int w = cell[u] / 3, sh = (cell[u] - 3 * w) * CC_K1_IDX_BITS;
          atomicOr(&idx3[w], CC_K1_IDX_MASK << sh);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CC_K1_U; u++) {
      if (cell[u] >= 0 && key[u] == hmax[cell[u]] && key[u] != KEY_EMPTY) {
        const unsigned long long j = (unsigned long long)(idx_base + base + tid + u * nt);
        const int w = cell[u] / 3, sh = (cell[u] - 3 * w) * CC_K1_IDX_BITS;
        unsigned long long old = idx3[w];
        while (true) {
          unsigned long long cur = (old >> sh) & CC_K1_IDX_MASK;
          if (j >= cur) break;
          unsigned long long nw = (old & ~(CC_K1_IDX_MASK << sh)) | (j << sh);
          unsigned long long got = atomicCAS(&idx3[w], old, nw);
          if (got == old) break;
          old = got;
        }
      }
    }
    __syncthreads();
  }
  vmax = cc_wave_max(vmax);
  vmin = cc_wave_min(vmin);
  if ((tid & 63) == 0) {
    atomicMax(&red[0], cc_fkey(vmax));
    atomicMin(&red[1], cc_fkey(vmin));
  }
  __syncthreads();

  if (PART) {  // this range's grid to the scratch; cc_k_rasterize_merge combines the ranges
    unsigned *pk = part.key + (size_t)blockIdx.x * n_cell;
    int *pj = part.idx + (size_t)blockIdx.x * n_cell;
    for (int c = tid; c < n_cell; c += nt) {
      const unsigned k = hmax[c];
      pk[c] = k;
      const int w = c / 3, sh = (c - 3 * w) * CC_K1_IDX_BITS;
      pj[c] = k != KEY_EMPTY ? (int)((idx3[w] >> sh) & CC_K1_IDX_MASK) : -1;
    }
    if (tid == 0) {
      part.red[(size_t)blockIdx.x * 2] = red[0];
      part.red[(size_t)blockIdx.x * 2 + 1] = red[1];
    }
    return;
  }
  // ---- outputs ----
  float *bev = bev_out + (size_t)scan * n_cell;
  float2 *pix = pix_out + (size_t)scan * n_cell;
  int npix = cc_k1_emit(
      cfg,
      [&](int c) { return hmax[c]; },
      [&](int c) {
        const int w = c / 3, sh = (c - 3 * w) * CC_K1_IDX_BITS;
        return (int)((idx3[w] >> sh) & CC_K1_IDX_MASK);
      },
      P, bev, pix, list_out, scan, emit_tab);
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

// The ranges of a scan combined: a cell's height is the largest of the ranges' keys and its point the one of the FIRST
// range that reaches it -- ranges are in file order and each holds the first of its own points, so this is the first point
// of the scan at that height: the reference's strict `bev < height` update (contour_mng.h:517) as in the one-sweep kernel.
// grid = n_scans, block = multiple of 64
__global__ void __launch_bounds__(1024)
cc_k_rasterize_merge(cc_dev_cfg cfg, const float4 *__restrict__ pts, const long long *__restrict__ offsets, cc_k1_part part,
                     float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out, cc_k1_list_out list_out) {
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
      P, bev, pix, list_out, scan, emit_tab);
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
