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

// grid = n_scans (PART: n_scans * CC_K1_SPLIT), block = multiple of 64.  dynamic LDS: n_cell*4 + ((n_cell+2)/3)*8 + 16 bytes.
template <int CC_K1_U, bool CC_K1_POW2, bool PART = false>
__global__ void __launch_bounds__(1024)
cc_k_rasterize(cc_dev_cfg cfg, const float4 *__restrict__ pts, const long long *__restrict__ offsets,
               float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out, cc_k1_part part = cc_k1_part()) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int n_cell = cfg.n_cell;
  unsigned *hmax = (unsigned *)smem;
  const int n_w3 = (n_cell + 2) / 3;
  unsigned long long *idx3 = (unsigned long long *)(smem + (((size_t)n_cell * 4 + 15) & ~(size_t)15));
  unsigned *red = (unsigned *)(idx3 + n_w3);  // [0]=max key [1]=min key [2]=n_pix

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
  int npix = 0;
  for (int c = tid; c < n_cell; c += nt) {
    unsigned k = hmax[c];
    bev[c] = cc_funkey(k);
    if (k != KEY_EMPTY) {
      const int w = c / 3, sh = (c - 3 * w) * CC_K1_IDX_BITS;
      int j = (int)((idx3[w] >> sh) & CC_K1_IDX_MASK);
      float4 q = P[j];
      // pointToContRowCol: x / reso + n_row/2 - 0.5f, left to right in f32
      float2 rc;
      rc.x = q.x / cfg.reso_row + (float)cfg.half_row - 0.5f;
      rc.y = q.y / cfg.reso_col + (float)cfg.half_col - 0.5f;
      pix[c] = rc;
      npix++;
    }
  }
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
                     float *__restrict__ bev_out, float2 *__restrict__ pix_out, cc_k1_scan_out *__restrict__ scan_out) {
  __shared__ unsigned red[3];
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
  int npix = 0;
  for (int c = tid; c < n_cell; c += nt) {
    unsigned k[CC_K1_SPLIT];
#pragma unroll
    for (int p = 0; p < CC_K1_SPLIT; p++) k[p] = part.key[((size_t)scan * CC_K1_SPLIT + p) * n_cell + c];
    unsigned best = KEY_EMPTY;
    int bp = -1;
#pragma unroll
    for (int p = 0; p < CC_K1_SPLIT; p++)
      if (k[p] != KEY_EMPTY && (bp < 0 || k[p] > best)) {
        best = k[p];
        bp = p;
      }
    bev[c] = cc_funkey(best);
    if (bp >= 0) {
      const int j = part.idx[((size_t)scan * CC_K1_SPLIT + bp) * n_cell + c];
      const float4 q = P[j];
      float2 rc;
      rc.x = q.x / cfg.reso_row + (float)cfg.half_row - 0.5f;
      rc.y = q.y / cfg.reso_col + (float)cfg.half_col - 0.5f;
      pix[c] = rc;
      npix++;
    }
  }
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
