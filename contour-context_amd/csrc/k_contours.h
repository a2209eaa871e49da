// K2 -- multi-level contour extraction + descriptors + retrieval keys + BCIs, one workgroup per
// scan, BEV grid and label image resident in LDS.  Replaces
//   ContourManager::makeContourRecursiveHelper   src/cont2/contour_mng.cpp:274-353
//   RunningStatRecorder / ContourView::calcStatVals   contour.h:48-95,142-255
//   ContourManager::makeContoursRecurs           contour_mng.h:588-895 (sort, keys, BCI)
//
// How the recursion is flattened: the level sets bev > lv_grads_[l] are nested,
// so the recursive "threshold inside the parent's mask + CCL" equals, per level, a global 8-connected
// labelling of bev > lv_grads_[l] keeping components of >= min_cont_cell_cnt_ cells.  What the
// recursion adds is the INSERTION ORDER into cont_views_[l] (depth-first, children in OpenCV label
// order = order of the first 2x2 block in block-raster order relative to the parent's bounding
// box).  Levels are processed top-down so that a child's parent is a plain label lookup; the
// insertion rank is then rebuilt bottom-up from (parent rank, first-block key), and the reference's
// unstable size sort is replayed with the libstdc++ introsort replica (cc_sort.h).
//
// Exactness: the reference accumulates cell_vol3_ (f32) and the ring-bin divisions (f32) in raster
// order; f32 addition is not associative, so those sums are evaluated by ONE lane per contour /
// per (anchor, division) in the same order.  Parallelism comes from contours x levels x anchors.
#pragma once
#include "cc_dev.h"
#include "cc_group.h"
#include "cc_sort.h"
#include "cc_stats.h"
#include "cc_fmath.h"

// the value becomes opaque to the optimiser at this point (no instruction is emitted)
#ifndef CC_OPAQUE_I
#define CC_OPAQUE_I(x) asm volatile("" : "+v"(x))
#endif

#define CC_NC CC_MAXC          // kept components per level handled exactly
#define CC_LAB_NONE 0xFFFFu
#define CC_LAB_PENDING 0x7FFEu  // a kept root of the list's tail between the kept test and its numbering (never a cell index: n_cell <= 22500)

struct cc_comp_t {  // per kept component, spilled to global scratch between levels
  uint16_t root, area, parent, rank;
  uint8_t r0, r1, c0, c1, cA, cB, pad[2];
};  // 16 B

template <int NC>
struct cc_k2_scratch_t {  // per scan of a launch
  cc_comp_t comp[CC_NLEV][NC];
  cc_contour_t cont[CC_NLEV][NC];
  alignas(16) uint16_t memb[CC_NLEV][((CC_MAX_CELLS + 7) & ~7) + 8 * NC];  // per level: the components' member lists (positions in `act`, raster
                                                              // order), every list starts on a 16-byte boundary
  uint16_t act[CC_MAX_CELLS];                               // active cells (above the lowest level), raster order
  uint16_t compidx[CC_NLEV][CC_MAX_CELLS];                  // per level: component index of list entry i, 0x7FFF = none
};
typedef cc_k2_scratch_t<CC_NC> cc_k2_scratch;
// The exact slow path for scans with more than CC_NC components on a level (the reference has no such limit,
// contour_mng.h:92-110): the same kernel body with room for CC_NC_BIG components per level, its per-component tables in a
// global block instead of LDS.  CC_NC_BIG < 4096 (the sort replay's range); a 150 x 150 image cannot hold that many
// 8-separated components of three cells (each needs one of the 5 625 aligned 2 x 2 blocks to itself, and its neighbours'
// blocks cannot all be used).
#define CC_NC_BIG 3840
struct cc_k2_big_tables {  // what the fast instance keeps in LDS, sized for CC_NC_BIG
  unsigned W[7 * CC_NC_BIG];
  uint16_t roots[CC_NC_BIG], cand[CC_NC_BIG], prev_root[CC_NC_BIG];
  uint16_t moff[CC_NLEV * CC_NC_BIG], mcnt[CC_NLEV * CC_NC_BIG];
  uint16_t big[CC_NLEV * CC_NC_BIG];
  alignas(16) cc_comp_t T[CC_NLEV * CC_NC_BIG];
  unsigned skey[CC_NLEV * CC_NC_BIG], arr[CC_NLEV * CC_NC_BIG];
};
struct cc_k2_big_slot {  // one workgroup of the slow path
  cc_k2_scratch_t<CC_NC_BIG> scr;
  cc_k2_big_tables tab;
};
struct cc_k2_big_queue {  // filled by the fast launch, drained by the slow one (whose last workgroup leaves it empty again)
  int n_flagged, next, exited, total;  // total: scans ever queued (statistics, never reset)
  int why[4];   // ... by reason (list kernel: configuration, cells / slots, components per level, list padding)
  int scan[1];  // [max_batch] follows
};
#define CC_K2_NCLK 32    // phase-clock slots per scan (tuning aid)
#define CC_K2_OWN 6       // list entries a thread keeps in registers (beyond: read from the scratch block; a street scene has
                          // 3-4 k active cells: with four per thread half of them were re-read from the scratch block -- an L2 round
                          // trip -- in every pass of every level)
#define CC_K2_CACHE 3072  // active cells whose height / position are staged in LDS for the walk
#define CC_K2_BIG 128     // components with more cells than this are walked by eight lanes, one running sum each

struct cc_anchor_lds {  // top contours of each level needed by keys / BCI
  float pm[2];
  float ev[2];
  int cnt;
};

#define CC_K2_R_BYTES 57344
// dynamic LDS: LV u8[n_cell] | R  -- 78 KB for the 150x150 grid, so that two workgroups (scans) share a CU and the
// lane-serial stretches of one (sort replays, raster-order sums) overlap with the parallel phases of the other
#define CC_K2_LV_BYTES(nc) (((size_t)(nc) + 15) & ~(size_t)15)
#define CC_K2_LDS_BYTES(nc) (CC_K2_LV_BYTES(nc) + CC_K2_R_BYTES)
#define CC_K2_BLOCK 512
#define CC_KEYS_GRP 18   // anchors whose RoI cell lists are in LDS at a time (a street scene's ~18 valid anchors: one group)
#define CC_KEYS_KL 4     // lanes per (anchor, division): a quad shares the list's cells, its first lane keeps the sum
#define CC_KEYS_CAP 416  // cells per list: a disc of radius 10 touches < 400 unit cells

// Label image conventions (u16 per cell): 0xFFFF = not in the level set; a non-root cell holds its root's cell index
// (< 0x8000); a root holds its own index while the labelling runs and, once the kept components are numbered,
// 0x8000 | component index.  cc_lab_comp: component index of a cell, CC_COMP_NONE if it has none.
#define CC_COMP_NONE 0x7FFFu
__device__ __forceinline__ unsigned cc_lab_comp(const uint16_t *LAB, int c) {
  unsigned v = LAB[c];
  if (!(v & 0x8000u)) v = LAB[v];            // cell -> its root
  return (v & 0x8000u) ? (v & 0x7FFFu) : CC_COMP_NONE;  // unmarked root: component not kept
}

// ---- union-find on the u16 label image (labels = cell indices, a root points to itself, parents always point to a
// smaller index so the root of a component is its smallest cell: the label min-propagation would converge to).
// Links are 32-bit CAS on the word holding the u16; no plain stores happen while unions run.
__device__ __forceinline__ unsigned cc_uf_find(const uint16_t *LAB, unsigned x) {
  unsigned p;
  while ((p = cc_lds_vread16(LAB + x)) != x) x = p;
  return x;
}
__device__ __forceinline__ void cc_uf_union(uint16_t *LAB, unsigned a, unsigned b) {
  a = cc_uf_find(LAB, a);
  b = cc_uf_find(LAB, b);
  while (a != b) {
    if (a < b) {
      const unsigned t = a;
      a = b;
      b = t;
    }
    // a > b: hang root a under b, provided a is still a root
    unsigned *w = (unsigned *)LAB + (a >> 1);
    const int shf = (a & 1) * 16;
    unsigned old = cc_lds_vread32(w);
    unsigned cur;
    while (true) {
      cur = (old >> shf) & 0xFFFFu;
      if (cur != a) break;
      const unsigned got = atomicCAS(w, old, (old & ~(0xFFFFu << shf)) | (b << shf));
      if (got == old) {
        cur = b;
        break;
      }
      old = got;
    }
    if (cur == b) break;
    a = cc_uf_find(LAB, cur);
    b = cc_uf_find(LAB, b);
  }
}


__device__ __forceinline__ unsigned cc_umin(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ int cc_cnt2_get(const unsigned *cnt2, int r) { return (cnt2[r >> 4] >> ((r & 15) * 2)) & 3; }

// Rare configuration (min_cont_cell_cnt_ > 3), kept out of line so that its registers do not count against the kernel:
// compaction of the kept-component tables W[7][CC_NC] / roots to the components with area >= min_cnt; dropped roots
// become unmarked again.  Entries only move to lower indices, so chunks of blockDim components go front to back.
// wsum: 8 ints of LDS.  Returns the new count (uniform).
__device__ __noinline__ int cc_k2_drop_small(int min_cnt, int n_kept, unsigned *W, uint16_t *roots, uint16_t *LAB, int *wsum,
                                            uint16_t *remap /*[old index] = new index or 0x7FFF*/, int nc /* row stride of W */) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave_id = tid >> 6, n_waves = nt >> 6;
  int n_new = 0;
  for (int k0 = 0; k0 < n_kept; k0 += nt) {
    const int k = k0 + tid;
    unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0, rt = 0;
    bool keep = false;
    if (k < n_kept) {
      v0 = W[k];
      v1 = W[nc + k];
      v2 = W[2 * nc + k];
      v3 = W[3 * nc + k];
      v4 = W[4 * nc + k];
      v5 = W[5 * nc + k];
      v6 = W[6 * nc + k];
      rt = roots[k];
      keep = (int)v3 >= min_cnt;  // W[3] = area
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wave_id] = __popcll(m);
    __syncthreads();
    int off = n_new + __popcll(m & ((1ull << lane) - 1ull));
    int tot = 0;
    for (int w = 0; w < n_waves; w++) {
      if (w < wave_id) off += wsum[w];
      tot += wsum[w];
    }
    if (k < n_kept) {
      if (keep) {
        W[off] = v0;
        W[nc + off] = v1;
        W[2 * nc + off] = v2;
        W[3 * nc + off] = v3;
        W[4 * nc + off] = v4;
        W[5 * nc + off] = v5;
        W[6 * nc + off] = v6;
        roots[off] = (uint16_t)rt;
        LAB[rt] = (uint16_t)(0x8000u | (unsigned)off);
        remap[k] = (uint16_t)off;
      } else {
        LAB[rt] = (uint16_t)rt;  // unmarked root: component not kept
        remap[k] = (uint16_t)0x7FFFu;
      }
    }
    n_new += tot;
    __syncthreads();
  }
  return n_new;
}

// optional phase timestamps (tuning aid): phase_clk[scan*CC_K2_NCLK + i], written by thread 0; the macros below expect
// `phase_clk`, `scan`, `tid`, `tmark`, `tsub` in scope
#define CC_K2_STAMP(i)                                                                     \
  do {                                                                                     \
    if (phase_clk && threadIdx.x == 0) phase_clk[(size_t)scan * CC_K2_NCLK + (i)] = (long long)wall_clock64(); \
  } while (0)
#define CC_K2_SUBLAP(j)                                                                   \
  do {                                                                                    \
    if (phase_clk) {                                                                      \
      const long long now_ = (long long)wall_clock64();                                   \
      if (tid == 0) phase_clk[(size_t)scan * CC_K2_NCLK + 16 + (j)] += now_ - tsub; \
      tsub = now_;                                                                        \
    }                                                                                     \
  } while (0)
#define CC_K2_LAP(acc)                                     \
  do {                                                     \
    if (phase_clk) {                                       \
      const long long now_ = (long long)wall_clock64();    \
      acc += now_ - tmark;                                 \
      tmark = now_;                                        \
    }                                                      \
  } while (0)

// How the back half (ordering, emit, keys, BCI) looks up the level count of a cell: the cell-indexed level image of the
// original front half, or the list front half's occupancy bit map + per-entry level bytes.
struct cc_k2_levmap {
  const unsigned char *LV;
  const unsigned long long *bitmap;  // bit c & 63 of word c >> 6: cell c is active
  const uint16_t *cbase;             // active cells before chunk c >> 6
  const unsigned char *lev;          // level count per list entry
  const uint16_t *g_rc;              // the scan's entry list in global memory (k_rasterize.h): (row << 8) | col ...
  const float2 *g_pix;               // ... and continuous position per entry
  int n_act;                         // entries
};
template <bool LISTED>
__device__ __forceinline__ int cc_k2_lev_at(const cc_k2_levmap &M, int cell) {
  if (!LISTED) return (int)M.LV[cell];
  const unsigned long long m = M.bitmap[cell >> 6];
  const int bit = cell & 63;
  if (!((m >> bit) & 1ull)) return 0;
  return (int)M.lev[(int)M.cbase[cell >> 6] + __popcll(m & ((1ull << bit) - 1ull))];
}

// The back half of K2: insertion order, size sort, emit, retrieval keys, BCIs -- from the per-level component records
// (scr->comp) and contour rows (scr->cont) a front half has left in the scratch block.  R: 57 344 bytes of LDS to carve.
template <int NC, bool BIG, bool LISTED>
__device__ __forceinline__ void cc_k2_back(const cc_dev_cfg &cfg, const float2 *__restrict__ pix, const cc_k1_scan_out *__restrict__ k1_out,
                                           cc_k2_scratch_t<NC> *__restrict__ scr, cc_k2_big_tables *__restrict__ bigtab, int scan,
                                           cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg, long long *__restrict__ phase_clk,
                                           char *R, const int *n_lev_in, int flags_in, const cc_k2_levmap &lm) {
  const int n_cell = cfg.n_cell, n_col = cfg.n_col, n_row = cfg.n_row;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave_id = tid >> 6, lane = tid & 63, n_waves = nt >> 6;
  cc_scan_desc_t *desc = desc_out + scan;
  long long tmark = 0;
  // =========================== phase "order": region R re-carved ===========================
  int n_lev[CC_NLEV];
  for (int l = 0; l < CC_NLEV; l++) n_lev[l] = n_lev_in[l];
  // lane-dependent level index: a select chain over the six registers (a lane-indexed array would live in scratch memory)
#define CC_NLEV_AT(i) ((i) == 0 ? n_lev[0] : (i) == 1 ? n_lev[1] : (i) == 2 ? n_lev[2] : (i) == 3 ? n_lev[3] : (i) == 4 ? n_lev[4] : n_lev[5])
  const int flags0 = flags_in;
  cc_comp_t *T = BIG ? bigtab->T : (cc_comp_t *)R;                          // [6][NC] 30720 B
  unsigned *skey = BIG ? bigtab->skey : (unsigned *)(R + 30720);            // [6][NC] u32 7680 B
  unsigned *arr = BIG ? bigtab->arr : (unsigned *)(R + 30720 + 7680);       // [6][NC] u32 7680 B
  cc_anchor_lds *top = (cc_anchor_lds *)(R + 30720 + 2 * 7680);             // [6][10] 1200 B
  int *sh2 = (int *)(R + 30720 + 2 * 7680 + 1280);                          // scalars (64 ints)
  char *R2 = R + 30720 + 2 * 7680 + 1280 + 256;                             // free for keys / BCI (~17.8 KB) -- see below
  for (int l = 0; l < CC_NLEV; l++)
    for (int k = tid; k < n_lev[l]; k += nt) T[l * NC + k] = scr->comp[l][k];
  __syncthreads();
  // insertion rank, bottom-up
  for (int l = 0; l < CC_NLEV; l++) {
    const int n = n_lev[l];
    for (int k = tid; k < n; k += nt) {
      const cc_comp_t cp = T[l * NC + k];
      int py0 = 0, px0 = 0, prank = 0;
      if (l > 0) {
        unsigned p = cp.parent;
        if (p < (unsigned)n_lev[l - 1]) {
          const cc_comp_t pp = T[(l - 1) * NC + p];
          py0 = pp.r0;
          px0 = pp.c0;
          prank = pp.rank;
        }
      }
      const int rmin = cp.r0;
      const int brow = (rmin - py0) >> 1;
      const int ra = py0 + 2 * brow;
      int cm = cp.cA;
      if (ra == rmin && cp.cB < cm) cm = cp.cB;
      const int bcol = (cm - px0) >> 1;
      skey[l * NC + k] = ((unsigned)prank << 14) | ((unsigned)brow << 7) | (unsigned)bcol;
    }
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += nt >> 2) {  // four threads per component, every fourth key each (block-uniform trip count)
      const int k = k0 + (tid >> 2), q = tid & 3;
      const unsigned me = k < n ? skey[l * NC + k] : 0u;
      int rk = 0;
      for (int j = q; j < n; j += 4) rk += (skey[l * NC + j] < me) ? 1 : 0;
      rk += __shfl_xor(rk, 1);
      rk += __shfl_xor(rk, 2);
      if (k < n && q == 0) {
        T[l * NC + k].rank = (uint16_t)rk;
        // pre-sort sequence: element at insertion position rk is component k
        arr[l * NC + rk] = ((unsigned)T[l * NC + k].area << 16) | (unsigned)k;
      }
    }
    __syncthreads();
  }
  // size sort, bigger first (contour_mng.h:596-599): libstdc++'s std::sort replayed by one WAVE per level (round 3: one
  // lane per level, ~40 of a KITTI-shaped scan's 440 us): parallel partitions + stable rank, cc_sort.h.  The rank keys of
  // the insertion order (skey) are dead by now: their rows hold the partitions' stopper lists, then the ranked copy.
  for (int l = wave_id; l < CC_NLEV; l += n_waves) {
    unsigned *a = arr + l * NC;
    const int n = CC_NLEV_AT(l);
    unsigned short *lpos = (unsigned short *)(skey + l * NC), *rasc = lpos + NC;
    ccsort::std_sort_wave(
        a, n, [](unsigned x) { return 0xFFFFu - (x >> 16); },
        [&]() {
          for (int k = lane; k < n; k += 64) a[T[l * NC + k].rank] = ((unsigned)T[l * NC + k].area << 16) | (unsigned)k;
        },
        lane, lpos, rasc, skey + l * NC, (unsigned *)R2 + l * CC_SORT_STACK);
    int tot = 0;
    for (int i = lane; i < n; i += 64) tot += (int)(a[i] >> 16);
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) sh2[l] = tot;
  }
  __syncthreads();
  CC_K2_STAMP(5);
  // emit sorted contour tables + header
  for (int l = 0; l < CC_NLEV; l++) {
    const int n = n_lev[l] < CC_MAXC ? n_lev[l] : CC_MAXC;
    const int n_words = n * (int)(sizeof(cc_contour_t) / 4);
    const unsigned *src = (const unsigned *)&scr->cont[l][0];
    unsigned *dst = (unsigned *)&desc->cont[l][0];
    for (int w = tid; w < n_words; w += nt) {
      const int seq = w / 19, off = w - seq * 19;
      const int k = (int)(arr[l * NC + seq] & 0xFFFFu);
      dst[w] = src[k * 19 + off];
    }
    for (int seq = tid; seq < CC_NDIST && seq < n; seq += nt) {
      const int k = (int)(arr[l * NC + seq] & 0xFFFFu);
      const cc_contour_t *cv = &scr->cont[l][k];
      cc_anchor_lds a;
      a.pm[0] = cv->pos_mean[0];
      a.pm[1] = cv->pos_mean[1];
      a.ev[0] = cv->eig_vals[0];
      a.ev[1] = cv->eig_vals[1];
      a.cnt = cv->cell_cnt;
      top[l * CC_NDIST + seq] = a;
    }
  }
  if (tid < CC_NLEV) {
    desc->n_cont[tid] = CC_NLEV_AT(tid);
    desc->n_stored[tid] = CC_NLEV_AT(tid) < CC_MAXC ? CC_NLEV_AT(tid) : CC_MAXC;
    desc->layer_cell_cnt[tid] = sh2[tid];
  }
  if (tid == 0) {
    const cc_k1_scan_out k1 = k1_out[scan];
    desc->max_bin_val = k1.max_bin_val;
    desc->min_bin_val = k1.min_bin_val;
    desc->n_pix = k1.n_pix;
    int fl = flags0;
    for (int l = 0; l < CC_NLEV; l++) fl |= n_lev[l] > CC_MAXC ? CC_DESC_TRUNCATED : 0;  // exact, the CC_MAXC largest are stored
    desc->flags = fl;
  }
  if (labels_dbg) {
    // component index -> seq (position after the size sort); skey is free now
    for (int l = 0; l < CC_NLEV; l++)
      for (int seq = tid; seq < n_lev[l]; seq += nt) skey[l * NC + (arr[l * NC + seq] & 0xFFFFu)] = (unsigned)seq;
    __syncthreads();
    for (int l = 0; l < CC_NLEV; l++) {
      int16_t *ld = labels_dbg + ((size_t)scan * CC_NLEV + l) * n_cell;
      for (int c = tid; c < n_cell; c += nt) {
        int16_t v = ld[c];
        if (v >= 0) ld[c] = (int16_t)skey[l * NC + v];
      }
    }
  }
  __syncthreads();

  CC_K2_STAMP(6);
  // =========================== phase "keys" (contour_mng.h:693-830) ===========================
  // R2 layout: divs f32 [36][35] (5040) | cntp int[36] | valid int[36] | acc int[36] | bci tmp | bci pts
  float *divs = (float *)R2;
  int *cntp = (int *)(R2 + 5056);
  int *valid = cntp + 36;
  int *accum = valid + 36;
  const int NA = CC_NLEV * CC_NPIV;
  if (tid < NA) {
    const int ll = tid / CC_NPIV, seq = tid - ll * CC_NPIV;
    int ok = 0, acc = 0;
    if (seq < cfg.piv_firsts) {
      for (int s = 0; s <= seq; s++)
        if (s < CC_NLEV_AT(ll)) acc += top[ll * CC_NDIST + s].cnt;  // accumulate_cell_cnt (contour_mng.h:705-706)
      ok = (seq < CC_NLEV_AT(ll) && top[ll * CC_NDIST + seq].cnt >= cfg.min_cont_key_cnt) ? 1 : 0;
    }
    valid[tid] = ok;
    accum[tid] = acc;
    cntp[tid] = 0;
  }
  // the BCI phase's per-wave scratch (R2 + 5504 ...) is idle until then: the exp table and the list of valid anchors
  double *exp_tab = (double *)(R2 + 5504);                    // 512 B
  unsigned char *vlist = (unsigned char *)(R2 + 5504 + 512);  // [36] anchors with a key, ascending
  if (tid >= 64 && tid < 128) exp_tab[tid - 64] = __longlong_as_double((long long)cc_exp2_tab64[tid - 64]);
  __syncthreads();
  int NV = 0;
  for (int a = 0; a < CC_NLEV * CC_NPIV; a++) {  // uniform; 36 broadcast reads
    const int ok = valid[a];
    if (ok && tid == 0) vlist[NV] = (unsigned char)a;
    NV += ok;
  }
  __syncthreads();
  const int roi_pad = (int)ceilf(cfg.roi_radius + 1.f);
  const float div_len = cfg.roi_radius / (float)(7 * 5);
  const float bin_len = cfg.roi_radius / (float)7;
  const double r_lim = (double)cfg.roi_radius - 1e-2;
  // The 35 divisions of an anchor accumulate over the same cells (RoI cells above level 1 within the radius, in raster
  // order; contour_mng.h:735-770): the cell list -- distance to the anchor, number of levels above -- is built once per
  // anchor by a wave (ballot-ordered, so raster order is kept), then one lane per (anchor, division) walks it in LDS with
  // the reference's f32 accumulation order.  Anchors are handled CC_KEYS_GRP at a time; the lists live where the
  // ordering tables were.
  {
    const int CAP = CC_KEYS_CAP;
    static_assert(CC_KEYS_GRP * CC_KEYS_CAP * 5 + 4 * CC_KEYS_GRP <= 30720 + 2 * 7680 && CC_KEYS_KL == 4, "the RoI lists end below the anchor table; a quad per (anchor, division)");
    float *ldist = (float *)R;
    unsigned char *lhi = (unsigned char *)(R + (size_t)CC_KEYS_GRP * CC_KEYS_CAP * 4);
    int *lcnt = (int *)(R + (size_t)CC_KEYS_GRP * CC_KEYS_CAP * 5);
    long long acc_klist = 0, acc_kexp = 0;
    tmark = phase_clk ? (long long)wall_clock64() : 0;
    // groups of CC_KEYS_GRP VALID anchors (a street scene has ~18 of the 36: one group)
    for (int g0 = 0; g0 < NV; g0 += CC_KEYS_GRP) {
      for (int av = g0 + wave_id; av < g0 + CC_KEYS_GRP && av < NV; av += n_waves) {
        const int a = (int)vlist[av];
        int n = 0;
        {
          const int ll = a / CC_NPIV, seq = a - ll * CC_NPIV;
          const float vcx = top[ll * CC_NDIST + seq].pm[0], vcy = top[ll * CC_NDIST + seq].pm[1];
          const int r_cen = (int)vcx, c_cen = (int)vcy;
          const int r_min = r_cen - roi_pad > 0 ? r_cen - roi_pad : 0;
          const int r_max = r_cen + roi_pad < n_row - 1 ? r_cen + roi_pad : n_row - 1;
          const int c_min = c_cen - roi_pad > 0 ? c_cen - roi_pad : 0;
          const int c_max = c_cen + roi_pad < n_col - 1 ? c_cen + roi_pad : n_col - 1;
          if (LISTED) {
            // The list front half's scans: the window's rows are ONE stretch of the raster-ordered entry list (first entry at or
            // after the first row's first cell .. last entry of the last row), so the wave reads entries -- (row, col), level
            // count, position, side by side in memory -- instead of probing 23 x 23 cells through the occupancy words; the
            // column test keeps the window's cells, in the same raster order.
            auto before = [&](int cell) {  // entries before `cell` (cell <= n_cell)
              if (cell >= n_cell) return lm.n_act;
              return (int)lm.cbase[cell >> 6] + (int)__popcll(lm.bitmap[cell >> 6] & ((1ull << (cell & 63)) - 1ull));
            };
            const int i_lo = before(r_min * n_col), i_hi = before((r_max + 1) * n_col);
            const int KB = 9;
            for (int base0 = i_lo; base0 < i_hi; base0 += 64 * KB) {
              int lvv[KB];
              float2 rcv[KB];
#pragma unroll
              for (int u = 0; u < KB; u++) {
                const int idx = base0 + 64 * u + lane;
                lvv[u] = 0;
                rcv[u] = make_float2(0.f, 0.f);
                if (idx < i_hi) {
                  // (column and position requested together -- both coalesced; a position fetched only for the cells that pass
                  // the column and level tests would be a second global round trip behind the first)
                  const int col = (int)(lm.g_rc[idx] & 255u);
                  rcv[u] = lm.g_pix[idx];
                  // `h < g1 -> skip`, then `h > g1` (contour_mng.h:742-748): together h > lv_grads[1] <=> level count >= 2;
                  // "higher" = #{e >= 1 : h > lv_grads[e]} = level count - 1
                  if (col >= c_min && col <= c_max) lvv[u] = (int)lm.lev[idx];
                }
              }
#pragma unroll
              for (int u = 0; u < KB; u++) {
                if (base0 + 64 * u >= i_hi) break;  // uniform
                bool q = false;
                float dist = 0.f;
                if (lvv[u] >= 2) {
                  const float dx = rcv[u].x - vcx, dy = rcv[u].y - vcy;
                  dist = sqrtf(dx * dx + dy * dy);
                  q = (double)dist < r_lim;
                }
                const unsigned long long m = __ballot(q);
                const int pos = n + cc_mbcnt(m);
                if (q && pos < CAP) {
                  ldist[(av - g0) * CAP + pos] = dist;
                  lhi[(av - g0) * CAP + pos] = (unsigned char)(lvv[u] - 1);
                }
                n += __popcll(m);
              }
            }
          } else {
            const int W = c_max - c_min + 1, tot = W * (r_max - r_min + 1);
            // nine 64-cell stretches (a 23 x 23 window) at a time: their level bytes and positions are fetched before any
            // of them is compacted, so the L2 round trips of the positions overlap
            const int KB = 9;
            for (int base0 = 0; base0 < tot; base0 += 64 * KB) {
              int lvv[KB];
              float2 rcv[KB];
#pragma unroll
              for (int u = 0; u < KB; u++) {
                const int idx = base0 + 64 * u + lane;
                lvv[u] = 0;
                rcv[u] = make_float2(0.f, 0.f);
                if (idx < tot) {
                  const int ro = idx / W;
                  const int cell = (r_min + ro) * n_col + c_min + (idx - ro * W);
                  // `h < g1 -> skip`, then `h > g1` (contour_mng.h:742-748): together h > lv_grads[1] <=> LV >= 2;
                  // "higher" = #{e >= 1 : h > lv_grads[e]} = LV - 1
                  lvv[u] = cc_k2_lev_at<LISTED>(lm, cell);
                  if (lvv[u] >= 2) rcv[u] = pix[cell];
                }
              }
#pragma unroll
              for (int u = 0; u < KB; u++) {
                if (base0 + 64 * u >= tot) break;  // uniform
                bool q = false;
                float dist = 0.f;
                if (lvv[u] >= 2) {
                  const float dx = rcv[u].x - vcx, dy = rcv[u].y - vcy;
                  dist = sqrtf(dx * dx + dy * dy);
                  q = (double)dist < r_lim;
                }
                const unsigned long long m = __ballot(q);
                const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
                if (q && pos < CAP) {
                  ldist[(av - g0) * CAP + pos] = dist;
                  lhi[(av - g0) * CAP + pos] = (unsigned char)(lvv[u] - 1);
                }
                n += __popcll(m);
              }
            }
          }
          if (n > CAP && lane == 0) atomicOr((unsigned *)&desc->flags, 4u);  // more RoI cells than the list holds: keys not exact
        }
        if (lane == 0) lcnt[av - g0] = n;
      }
      __syncthreads();
      CC_K2_LAP(acc_klist);
      // Four lanes per (anchor, division) (round 6; one before): 36 x 35 sums of up to 400 f64 exp each kept 420 of the 512
      // lanes busy for two rounds of anchors.  The quad's lanes take the list's cells i, i + 1, i + 2, i + 3, every lane adds
      // the four products in list order (the first lane's sum is the one kept): the same f32 additions in the same order --
      // a lane past the list's end contributes +0.0, which changes no sum.
      const int n_grp = NV - g0 < CC_KEYS_GRP ? NV - g0 : CC_KEYS_GRP;
      for (int t = tid; t < n_grp * 35 * CC_KEYS_KL; t += nt) {
        const int task = t / CC_KEYS_KL, r = t - task * CC_KEYS_KL;
        const int al = task / 35, d = task - al * 35;
        const int a = (int)vlist[g0 + al];
        float acc = 0.f;
        const int n = lcnt[al] < CAP ? lcnt[al] : CAP;
        {
          // gaussPDF<float>(div_idx*div_len + 0.5*div_len, dist, 1.0)  (tools/algos.h:54-56): exp(-u^2/2) / sqrt(2 pi), here
          // exp(...) * (1 / sqrt(2 pi)) -- one f64 rounding away from the quotient, gone in the conversion to f32
          const float xg = (float)((double)((float)d * div_len) + 0.5 * (double)div_len);
          for (int i0 = 0; i0 < n; i0 += CC_KEYS_KL) {
            const int i = i0 + r;
            float v = 0.f;
            if (i < n) {
              const float dist = ldist[al * CAP + i];
              const int higher = lhi[al * CAP + i];
              const float u = (xg - dist) / 1.0f;
              const float pdf = (float)(cc_exp_nonpos(-0.5 * (double)u * (double)u, exp_tab) * 0.3989422804014327);
              v = (float)higher * pdf;
            }
            acc += cc_quad_bcast<0>(v);
            acc += cc_quad_bcast<1>(v);
            acc += cc_quad_bcast<2>(v);
            acc += cc_quad_bcast<3>(v);
          }
        }
        if (r == 0) {
          divs[a * 35 + d] = acc;
          if (d == 0) cntp[a] = lcnt[al];
        }
      }
      __syncthreads();
      CC_K2_LAP(acc_kexp);
    }
    if (phase_clk && tid == 0) {
      phase_clk[(size_t)scan * CC_K2_NCLK + 14] = acc_klist;
      phase_clk[(size_t)scan * CC_K2_NCLK + 15] = acc_kexp;
    }
  }
  for (int t = tid; t < NA * CC_KEY_DIM; t += nt) {
    const int a = t / CC_KEY_DIM, kd = t - a * CC_KEY_DIM;
    const int ll = a / CC_NPIV, seq = a - ll * CC_NPIV;
    float v = 0.f;
    if (valid[a]) {
      const cc_anchor_lds an = top[ll * CC_NDIST + seq];
      if (kd == 0)
        v = sqrtf(an.ev[1] * (float)an.cnt);
      else if (kd == 1)
        v = sqrtf(an.ev[0] * (float)an.cnt);
      else if (kd == 2)
        v = (float)sqrt((double)accum[a]);
      else {
        const int b = kd - 3;
        float ring = 0.f;
        for (int d = 0; d < 5; d++) ring += divs[a * 35 + b * 5 + d];
        ring = (float)((double)ring * ((double)bin_len / sqrt((double)cntp[a])));
        v = ring;
      }
    }
    desc->keys[ll][seq][kd] = v;
  }
  __syncthreads();

  CC_K2_STAMP(7);
  // =========================== phase "BCI" (contour_mng.h:848-883) ===========================
  struct bci_tmp {
    int ok;
    int bit;
    float r, theta;
  };
  bci_tmp *btmp = (bci_tmp *)(R + 17280);                          // [36][40] 23040 B  (T/skey/arr dead; ends < top)
  for (int t = tid; t < NA * CC_BCI_MAXPTS; t += nt) {
    const int a = t / CC_BCI_MAXPTS, q = t - a * CC_BCI_MAXPTS;
    const int bl = q / CC_NDIST, j = q - bl * CC_NDIST;
    const int ll = a / CC_NPIV, seq = a - ll * CC_NPIV;
    bci_tmp o;
    o.ok = 0;
    o.bit = 0;
    o.r = 0.f;
    o.theta = 0.f;
    const int lev = bl + 1;  // DIST_BIN_LAYERS = {1,2,3,4}
    const int lim = cfg.dist_firsts < CC_NLEV_AT(lev) ? cfg.dist_firsts : CC_NLEV_AT(lev);
    if (valid[a] && j < lim && !(ll == lev && j == seq)) {
      const float vx = top[lev * CC_NDIST + j].pm[0] - top[ll * CC_NDIST + seq].pm[0];
      const float vy = top[lev * CC_NDIST + j].pm[1] - top[ll * CC_NDIST + seq].pm[1];
      const float dist = sqrtf(vx * vx + vy * vy);
      const double dd = (double)dist;
      if (!(dd > (CC_BITS_PER_LAYER - 1) * 1.01 + 5.43 - 1e-3 || dd <= 5.43)) {
        const float orie = cc_atan2f_fdlibm(vy, vx);  // glibc's atan2f, operation for operation (cc_stats.h)
        double fl = floor((dd - 5.43) / 1.01);
        if (fl > CC_BITS_PER_LAYER - 1.0) fl = CC_BITS_PER_LAYER - 1.0;
        o.ok = 1;
        o.bit = (int)(fl + (double)(bl * CC_BITS_PER_LAYER));
        o.r = dist;
        o.theta = orie;
      }
    }
    btmp[t] = o;
  }
  __syncthreads();
  // The neighbour points are sorted through 32-bit proxies (bit_pos << 8 | slot, compared on bit_pos only): the replay of
  // std::sort depends on the comparison outcomes alone, so the proxies end up in the order the reference's
  // RelativePoint records would.
  unsigned *bkey = (unsigned *)(R + 0);                            // [36][40] (T is dead now)
  int *bcnt = (int *)(R + 5760);                                   // [36] points per anchor
  // One WAVE per anchor (round 3: one lane per anchor, a lane-serial gather + sort replay + segment scan of dependent LDS
  // reads, 26 us per scan whatever the scan): the <= 40 candidate neighbours sit one per lane, the valid ones are compacted
  // with a ballot (slot order, as the serial loop appended them), sorted by the wave-parallel std::sort replay (cc_sort.h),
  // segment starts found with a ballot.  Per-wave scratch behind the key tables in R2 (divs 5040 B + 3 x 36 ints end at 5488).
  {
    char *ws = R2 + 5504 + wave_id * 448;
    unsigned short *lpos = (unsigned short *)ws, *rasc = lpos + CC_BCI_MAXPTS;   // 2 x 80 B
    unsigned *tmp = (unsigned *)(ws + 160);                                        // 160 B
    unsigned *seg = (unsigned *)(ws + 320);                                        // CC_SORT_STACK words = 104 B
    static_assert(320 + CC_SORT_STACK * 4 <= 448 && CC_BCI_MAXPTS <= 64, "per-wave BCI scratch");
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int a = wave_id; a < NA; a += n_waves) {
      const int ll = a / CC_NPIV, seq = a - ll * CC_NPIV;
      unsigned *p = bkey + a * CC_BCI_MAXPTS;
      bci_tmp o;
      o.ok = 0;
      o.bit = 0;
      if (lane < CC_BCI_MAXPTS) o = btmp[a * CC_BCI_MAXPTS + lane];
      const unsigned long long mok = __ballot(o.ok != 0);
      const int n = __popcll(mok);
      auto gather = [&]() {
        if (o.ok) p[__popcll(mok & lt)] = ((unsigned)o.bit << 8) | (unsigned)lane;
      };
      gather();
      // the 256-bit ring: word w = OR of the valid lanes' bits of that word
      unsigned long long bw[4];
#pragma unroll
      for (int w = 0; w < 4; w++) {
        unsigned long long v = (o.ok && (o.bit >> 6) == w) ? (1ull << (o.bit & 63)) : 0ull;
        for (int sh_ = 32; sh_ > 0; sh_ >>= 1) v |= __shfl_xor(v, sh_);
        bw[w] = v;
      }
      ccsort::std_sort_wave(p, n, [](unsigned x) { return x >> 8; }, gather, lane, lpos, rasc, tmp, seg);
      cc_bci_t *ob = &desc->bcis[ll][seq];
      // segment starts: positions where bit_pos changes, then n (contour_mng.h:871-883)
      const bool head = lane < n && (lane == 0 || (p[lane] >> 8) != (p[lane - 1] >> 8));
      const unsigned long long mh = __ballot(head);
      const int nh = __popcll(mh), ns = n > 0 ? nh + 1 : 0;
      if (head) ob->segs[__popcll(mh & lt)] = (uint16_t)lane;
      if (lane == 0 && n > 0) ob->segs[nh] = (uint16_t)n;
      if (lane >= ns && lane < CC_BCI_MAXPTS + 2) ob->segs[lane] = 0;
      if (lane == 0) {
        bcnt[a] = n;
        ob->dist_bin[0] = bw[0];
        ob->dist_bin[1] = bw[1];
        ob->dist_bin[2] = bw[2];
        ob->dist_bin[3] = bw[3];
        // layer_key_bcis_ has piv_firsts_ entries per level (contour_mng.h:835-838): the record's slots beyond that stay all-zero
        ob->piv_seq = seq < cfg.piv_firsts ? (int8_t)seq : (int8_t)0;
        ob->level = seq < cfg.piv_firsts ? (int8_t)ll : (int8_t)0;
        ob->n_pts = (uint8_t)n;
        ob->n_segs = (uint8_t)ns;
      }
    }
  }
  __syncthreads();
  // the point records, all threads
  for (int t = tid; t < NA * CC_BCI_MAXPTS; t += nt) {
    const int a = t / CC_BCI_MAXPTS, i = t - a * CC_BCI_MAXPTS;
    const int ll = a / CC_NPIV, seq = a - ll * CC_NPIV;
    unsigned w0 = 0u;
    float r = 0.f, th = 0.f;
    if (i < bcnt[a]) {
      const unsigned k = bkey[a * CC_BCI_MAXPTS + i];
      const int q = (int)(k & 0xFFu);
      const bci_tmp o = btmp[a * CC_BCI_MAXPTS + q];
      // level (int8) | seq (int8) << 8 | bit_pos (int16) << 16
      w0 = (unsigned)(q / CC_NDIST + 1) | ((unsigned)(q % CC_NDIST) << 8) | ((k >> 8) << 16);
      r = o.r;
      th = o.theta;
    }
    unsigned *dst = (unsigned *)&desc->bcis[ll][seq].pts[i];
    dst[0] = w0;
    dst[1] = __float_as_uint(r);
    dst[2] = __float_as_uint(th);
  }
  CC_K2_STAMP(8);
}

// The kernel body.  NC = components per level it handles exactly; BIG = the per-component tables live in `bigtab` (global)
// instead of LDS.  `scan` indexes the launch's inputs and outputs, `scr` is the scratch block to use.
template <int NC, bool BIG>
__device__ __forceinline__ void cc_k2_body(const cc_dev_cfg &cfg, const float *__restrict__ bev_in, const float2 *__restrict__ pix_in,
                                           const cc_k1_scan_out *__restrict__ k1_out, cc_k2_scratch_t<NC> *__restrict__ scr,
                                           cc_k2_big_tables *__restrict__ bigtab, cc_k2_big_queue *__restrict__ queue, int scan,
                                           cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg, long long *__restrict__ phase_clk,
                                           char *smem) {
  CC_K2_STAMP(0);
  const int n_cell = cfg.n_cell, n_col = cfg.n_col, n_row = cfg.n_row;
  const int tid = threadIdx.x, nt = blockDim.x;

  unsigned char *LV = (unsigned char *)smem;                       // #levels the cell's height exceeds: bev > lv_grads[l] <=> LV > l
  char *R = smem + CC_K2_LV_BYTES(n_cell);
  // ---- region R, phase "levels" ----
  // (the LDS offsets below are the fast instance's layout, NC components per level; the big instance keeps the same
  // offsets for what does not grow with the component count and takes the rest from `bigtab`)
  uint16_t *LAB = (uint16_t *)R;                                   // n_cell u16 (45000)
  unsigned *W = BIG ? bigtab->W : (unsigned *)(R + 45056);         // 7 * NC u32 working arrays (8960 B of LDS, over the bit maps)
  uint16_t *roots = BIG ? bigtab->roots : (uint16_t *)(R + 45056 + 8960);                // NC u16
  uint16_t *cand = BIG ? bigtab->cand : roots + NC;                                      // NC u16
  uint16_t *prev_root = BIG ? bigtab->prev_root : cand + NC;                             // NC u16
  int *sh = (int *)(R + 45056 + 8960 + 3 * CC_NC * 2 + 64);        // small scalars
  // sh[2]=flags  sh[3]=#large components  sh[8+l]=n_kept[l]  sh[24+w]=per-wave counts

  const float *bev = bev_in + (size_t)scan * n_cell;
  const float2 *pix = pix_in + (size_t)scan * n_cell;
  cc_scan_desc_t *desc = desc_out + scan;

  // ---- level index of every cell; the ACTIVE cells (above the lowest level) as a raster-ordered list ----
  // Everything below works on that list (a few hundred to a few thousand cells of the 22 500): thread t owns entries
  // t, t + nt, ... and keeps its first CC_K2_OWN of them in registers; the list itself lives in the scan's scratch block.
  // The BEV is read once with coalesced 16-byte loads (n_row, n_col even: n_cell is a multiple of 4 and every scan's
  // image starts on a 16-byte boundary), four cells per lane: level index as one 32-bit LDS store, initial labels as one
  // 64-bit store.  The raster-ordered list then comes from the level image in LDS: every wave owns a contiguous range of
  // cells, counts its active cells with ballots, and after one barrier writes them at its offset -- consecutive lanes
  // write consecutive entries.
  {
    const float4 *bev4 = (const float4 *)bev;
    const int n_quad = n_cell >> 2;
#pragma unroll 6
    for (int v = tid; v < n_quad; v += nt) {  // 11 loads per thread for the 150 x 150 grid: two batches in flight
      const float4 h4 = bev4[v];
      const float hh[4] = {h4.x, h4.y, h4.z, h4.w};
      unsigned lv4 = 0;
      unsigned lab[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int lv = 0;
        for (int e = 0; e < CC_NLEV; e++) lv += (hh[u] > cfg.lv_grads[e]) ? 1 : 0;  // cv::threshold BINARY is strict `>` (contour_mng.cpp:283)
        lv4 |= (unsigned)lv << (8 * u);
        lab[u] = lv ? (unsigned)(4 * v + u) : (unsigned)CC_LAB_NONE;  // an active cell starts as its own root
      }
      ((unsigned *)LV)[v] = lv4;
      ((uint2 *)LAB)[v] = make_uint2(lab[0] | (lab[1] << 16), lab[2] | (lab[3] << 16));
    }
  }
  if (tid < 40) sh[tid] = 0;
  __syncthreads();
  const int wave_id = cc_wave_id(), lane = tid & 63, n_waves = nt >> 6;
  int n_act;
  {
    const int per_wave = (((n_cell + n_waves - 1) / n_waves) + 63) & ~63;
    const int w_lo = wave_id * per_wave < n_cell ? wave_id * per_wave : n_cell;
    const int w_hi = w_lo + per_wave < n_cell ? w_lo + per_wave : n_cell;
    int cnt = 0;
    for (int b = w_lo; b < w_hi; b += 64) {
      const int c = b + lane;
      cnt += __popcll(__ballot(c < w_hi && LV[c] != 0));
    }
    if (lane == 0) sh[24 + wave_id] = cnt;
    __syncthreads();
    int off = 0;
    n_act = 0;
    for (int w = 0; w < n_waves; w++) {
      if (w < wave_id) off += sh[24 + w];
      n_act += sh[24 + w];
    }
    for (int b = w_lo; b < w_hi; b += 64) {
      const int c = b + lane;
      const bool on = c < w_hi && LV[c] != 0;
      const unsigned long long m = __ballot(on);
      if (on) scr->act[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)c;
      off += __popcll(m);
    }
  }
  if (labels_dbg)
    for (int i = tid; i < CC_NLEV * n_cell; i += nt) labels_dbg[(size_t)scan * CC_NLEV * n_cell + i] = (int16_t)-1;
  __threadfence_block();
  __syncthreads();
  int mc[CC_K2_OWN];
#pragma unroll
  for (int u = 0; u < CC_K2_OWN; u++) mc[u] = tid + u * nt < n_act ? (int)scr->act[tid + u * nt] : -1;
  // `body(c, i)` for every active cell c = act[i] this thread owns
#define CC_K2_FOR_ACTIVE(...)                                                    \
  {                                                                              \
    _Pragma("unroll") for (int u_ = 0; u_ < CC_K2_OWN; u_++) {                   \
      if (mc[u_] >= 0) {                                                         \
        const int c = mc[u_], i = tid + u_ * nt;                                 \
        __VA_ARGS__                                                               \
      }                                                                          \
    }                                                                            \
    for (int i = tid + CC_K2_OWN * nt; i < n_act; i += nt) {                     \
      const int c = (int)scr->act[i];                                            \
      __VA_ARGS__                                                                 \
    }                                                                            \
  }

  CC_K2_STAMP(9);
  int prev_n = 0;
  long long acc_ccl = 0, acc_enum = 0, acc_walk = 0, tmark = phase_clk ? (long long)wall_clock64() : 0;
  // per pass of the level loop, summed over the levels: accumulated in the clock block itself (thread 0), so that the
  // production launch (phase_clk == nullptr) carries no accumulator registers for it
  long long tsub = tmark;
  if (phase_clk && tid == 0)
    for (int j = 0; j < 6; j++) phase_clk[(size_t)scan * CC_K2_NCLK + 16 + j] = 0;
  unsigned *w_minc = W + NC, *w_maxc = W + 2 * NC, *w_area = W + 3 * NC, *w_cB = W + 4 * NC;
  // "has a second cell" / "has a third cell" bit per root (the kept test of a level), over the working arrays
  unsigned *bitA = (unsigned *)(R + 45056), *bitB = bitA + ((n_cell + 31) >> 5);
  const int n_bw = 2 * ((n_cell + 31) >> 5);
  unsigned char *scnt = (unsigned char *)(R + 45056 + 8960 + CC_NC * 2);  // kept roots per 64-entry stretch of the active list (<= 352 stretches; the fast instance's `cand`)
  uint16_t *sbase = (uint16_t *)(sh + 64);       // their exclusive prefix sum
  const bool has_tail = n_act > CC_K2_OWN * nt;  // block-uniform: more active cells than the threads keep in registers
  const unsigned long long lane_lt = (1ull << lane) - 1ull;
  // Per owned cell, once: its level count and, for each of its four backward neighbours (W, NW, N, NE), the number of level
  // sets the two cells share -- the pair is linked at exactly one level, the highest one that holds both (two cells that
  // were together in the level above already share a root).  3 bits each: 12 bits per cell, two cells per register.
  unsigned lvo = 0, ew[CC_K2_OWN / 2];
  unsigned keptm = 0;  // owned cells that sat in a numbered component of the level above: their component has its three cells
#pragma unroll
  for (int u = 0; u < CC_K2_OWN / 2; u++) ew[u] = 0;
#pragma unroll
  for (int u = 0; u < CC_K2_OWN; u++) {
    if (mc[u] >= 0) {
      const int c = mc[u];
      const unsigned lvc = LV[c];
      const int r = c / n_col, cc = c - r * n_col;
      unsigned f = 0;
      if (cc > 0) f |= cc_umin(lvc, (unsigned)LV[c - 1]);
      if (r > 0) {
        if (cc > 0) f |= cc_umin(lvc, (unsigned)LV[c - n_col - 1]) << 3;
        f |= cc_umin(lvc, (unsigned)LV[c - n_col]) << 6;
        if (cc < n_col - 1) f |= cc_umin(lvc, (unsigned)LV[c - n_col + 1]) << 9;
      }
      lvo |= lvc << (3 * u);
      ew[u >> 1] |= f << ((u & 1) * 12);
    }
  }
  for (int l = CC_NLEV - 1; l >= 0; --l) {
    // (a) 8-connected labelling of the level set LV > l.  LAB still holds the forest of level l+1 (a subset of this level's
    //     cells, lv_grads ascending): those components stay merged; cells new at this level (LV == l + 1) are still their
    //     own roots.  One union per adjacent pair whose shared level count is l + 1.  A thread's pairs of this level are a
    //     bit mask over (owned cell, direction); the loop runs once per PAIR -- round 4 walked all 32 (cell, direction)
    //     slots and a wave entered the union code of a slot as soon as one of its lanes had a pair there: 75 of a street
    //     scene's 385 us.  The result does not depend on the order of the unions (a root is its component's smallest cell).
#pragma unroll
    for (int u = 0; u < CC_K2_OWN; u++) CC_OPAQUE_I(mc[u]);  // keeps the neighbour offsets of the owned cells from being hoisted (and then spilled)
    for (int i = tid; i < n_bw; i += nt) bitA[i] = 0;
    {
      const unsigned t = (unsigned)l + 1u;
      unsigned em = 0;
#pragma unroll
      for (int u = 0; u < CC_K2_OWN; u++) {
        const unsigned f = ew[u >> 1] >> ((u & 1) * 12);
#pragma unroll
        for (int d = 0; d < 4; d++) em |= (((f >> (3 * d)) & 7u) == t ? 1u : 0u) << (u * 4 + d);
      }
      while (em) {
        const int b = __ffs((int)em) - 1;
        em &= em - 1u;
        const int u = b >> 2, d = b & 3;
        int c = mc[0];
#pragma unroll
        for (int k = 1; k < CC_K2_OWN; k++) c = u == k ? mc[k] : c;
        const unsigned nbc = (unsigned)(c - (d == 0 ? 1 : n_col + 2 - d));
        // two cells with the same parent are in one tree already (most pairs of a cell lead to the component its first pair
        // joined): two independent reads instead of two finds one after the other
        if (cc_lds_vread16(LAB + c) != cc_lds_vread16(LAB + nbc)) cc_uf_union(LAB, (unsigned)c, nbc);
      }
    }
    if (has_tail)
      for (int i = tid + CC_K2_OWN * nt; i < n_act; i += nt) {
        const int c = (int)scr->act[i];
        const int lvc = LV[c];
        if (lvc > l) {
          const bool newc = lvc == l + 1;
          const int r = c / n_col, cc = c - r * n_col;
          if (cc > 0) {
            const int lvn = LV[c - 1];
            if (lvn > l && (newc || lvn == l + 1)) cc_uf_union(LAB, c, c - 1);
          }
          if (r > 0) {
            if (cc > 0) {
              const int lvn = LV[c - n_col - 1];
              if (lvn > l && (newc || lvn == l + 1)) cc_uf_union(LAB, c, c - n_col - 1);
            }
            {
              const int lvn = LV[c - n_col];
              if (lvn > l && (newc || lvn == l + 1)) cc_uf_union(LAB, c, c - n_col);
            }
            if (cc < n_col - 1) {
              const int lvn = LV[c - n_col + 1];
              if (lvn > l && (newc || lvn == l + 1)) cc_uf_union(LAB, c, c - n_col + 1);
            }
          }
        }
      }
    __syncthreads();
    CC_K2_SUBLAP(0);
    // (b) every cell is pointed at its root (a concurrent find that passes through the cell meets either its old parent or
    //     the root: both lead to the root); the owned cells' finds advance hop by hop TOGETHER (eight LDS reads in flight,
    //     not eight chains one after the other).  Which roots own >= min_cont_cell_cnt_ (3) cells: a member that is not the
    //     root sets the root's bit in A, and in B if A was set already -- A: a second cell, B: a third one.
    const int need = cfg.min_cont_cell_cnt < 3 ? cfg.min_cont_cell_cnt : 3;
    unsigned inm = 0, rootm = 0;  // owned cells in the level set / that are roots
    {
      unsigned x[CC_K2_OWN];
#pragma unroll
      for (int u = 0; u < CC_K2_OWN; u++) {
        inm |= (((lvo >> (3 * u)) & 7u) > (unsigned)l ? 1u : 0u) << u;
        x[u] = (unsigned)mc[u];
      }
      bool more = inm != 0;
      while (more) {
        more = false;
#pragma unroll
        for (int u = 0; u < CC_K2_OWN; u++) {
          const unsigned pq = (inm >> u) & 1u ? cc_lds_vread16(LAB + x[u]) : x[u];
          more |= pq != x[u];
          x[u] = pq;
        }
      }
#pragma unroll
      for (int u = 0; u < CC_K2_OWN; u++) {
        if ((inm >> u) & 1u) {
          const unsigned rt = x[u];
          if (rt != (unsigned)mc[u]) {
            LAB[mc[u]] = (uint16_t)rt;
            if (!((keptm >> u) & 1u)) {  // else: the component's old root reports it below
              const unsigned bit = 1u << (rt & 31u);
              if (atomicOr(&bitA[rt >> 5], bit) & bit) atomicOr(&bitB[rt >> 5], bit);
            }
          } else {
            rootm |= 1u << u;
          }
        }
      }
    }
    // a component numbered at the level above (>= 3 cells there) is part of one component here: its old root marks the new
    // one, its other cells (keptm) skip the bit protocol -- a large component's members no longer queue on one LDS word
    for (int k = tid; k < prev_n; k += nt) {
      const unsigned rt = cc_uf_find(LAB, prev_root[k]);
      const unsigned bit = 1u << (rt & 31u);
      atomicOr(&bitA[rt >> 5], bit);
      atomicOr(&bitB[rt >> 5], bit);
    }
    if (has_tail)
      for (int i = tid + CC_K2_OWN * nt; i < n_act; i += nt) {
        const int c = (int)scr->act[i];
        if (LV[c] > l) {
          const unsigned rt = cc_uf_find(LAB, c);
          if (rt != (unsigned)c) {
            LAB[c] = (uint16_t)rt;
            const unsigned bit = 1u << (rt & 31u);
            if (atomicOr(&bitA[rt >> 5], bit) & bit) atomicOr(&bitB[rt >> 5], bit);
          }
        }
      }
    __syncthreads();
    CC_K2_SUBLAP(1);
    CC_K2_LAP(acc_ccl);
    // (c) kept roots, numbered by cell index = raster order of their first cells.  The active list IS in raster order: a
    //     kept root's number is the count of kept roots before it in the list -- ballots per 64-entry stretch (thread t's
    //     u-th cell is entry u * nt + t: a wave's u-th cells are one stretch) and a prefix sum over the stretches.  (Round 4:
    //     an atomic counter, then every root ranked against every other: 24 us of a street scene's scan.)
#define CC_K2_ROOT_KEPT(c_) (need <= 1 || (((need == 2 ? bitA : bitB)[(c_) >> 5] >> ((c_) & 31)) & 1u))
    unsigned kpm = 0;
#pragma unroll
    for (int u = 0; u < CC_K2_OWN; u++) {
      const bool kp = ((rootm >> u) & 1u) && CC_K2_ROOT_KEPT(mc[u]);
      kpm |= (kp ? 1u : 0u) << u;
      const unsigned long long m = __ballot(kp);
      if (lane == 0) scnt[u * n_waves + wave_id] = (unsigned char)__popcll(m);
    }
    if (has_tail)
      for (int ib = CC_K2_OWN * nt; ib < n_act; ib += nt) {  // block-uniform trip count
        const int i = ib + tid;
        bool kp = false;
        if (i < n_act) {
          const int c = (int)scr->act[i];
          kp = LV[c] > l && LAB[c] == (unsigned)c && CC_K2_ROOT_KEPT(c);
          if (kp) LAB[c] = (uint16_t)CC_LAB_PENDING;  // remembered for the numbering pass (the bit maps are gone by then); only this thread looks at LAB[c] in between
        }
        const unsigned long long m = __ballot(kp);
        if (lane == 0) scnt[(ib >> 6) + wave_id] = (unsigned char)__popcll(m);
      }
    __syncthreads();
    CC_K2_SUBLAP(2);
    // every wave makes the whole prefix array (identical values from all of them) and then reads its own entries
    int n_kept = 0;
    {
      const int n_str = (((n_act + nt - 1) / nt) * nt) >> 6;  // stretches written above
      for (int q = 0; q < n_str; q += 64) {
        const int v = q + lane < n_str ? (int)scnt[q + lane] : 0;
        const int incl = cc_wave_scan_incl(v);
        if (q + lane < n_str) sbase[q + lane] = (uint16_t)(n_kept + incl - v);
        n_kept += cc_wave_scan_total(incl);
      }
    }
    cc_wave_sync();
    if (n_kept > NC && tid == 0) sh[2] |= 2;  // capacity exceeded: this scan's descriptor is not exact (the NC first roots are kept)
#pragma unroll
    for (int u = 0; u < CC_K2_OWN; u++) {
      const bool kp = (kpm >> u) & 1u;
      const unsigned long long m = __ballot(kp);
      if (kp) {
        const int rk = (int)sbase[u * n_waves + wave_id] + __popcll(m & lane_lt);
        if (rk < NC) {
          roots[rk] = (uint16_t)mc[u];
          LAB[mc[u]] = (uint16_t)(0x8000u | (unsigned)rk);
          // working arrays of the kept components (they lie over the bit maps: every kept test is behind the barrier above)
          w_minc[rk] = 0xFFFFu;
          w_maxc[rk] = 0;
          w_area[rk] = 0;
          w_cB[rk] = 255;
        }
      }
    }
    if (has_tail)
      for (int ib = CC_K2_OWN * nt; ib < n_act; ib += nt) {
        const int i = ib + tid;
        bool kp = false;
        int c = 0;
        if (i < n_act) {
          c = (int)scr->act[i];
          kp = LAB[c] == (unsigned)CC_LAB_PENDING;
        }
        const unsigned long long m = __ballot(kp);
        if (kp) {
          const int rk = (int)sbase[(ib >> 6) + wave_id] + __popcll(m & lane_lt);
          if (rk < NC) {
            roots[rk] = (uint16_t)c;
            LAB[c] = (uint16_t)(0x8000u | (unsigned)rk);
            w_minc[rk] = 0xFFFFu;
            w_maxc[rk] = 0;
            w_area[rk] = 0;
            w_cB[rk] = 255;
          } else {
            LAB[c] = (uint16_t)c;  // beyond the capacity: an unmarked root again
          }
        }
      }
    if (n_kept > NC) n_kept = NC;
    __syncthreads();
    CC_K2_SUBLAP(3);
    // (d) per component: area, column range, last cell of the raster order, first member column of the second row
    //     (the root IS the first cell: first row and its first member column need no search), and for the walk the
    //     component index of every member cell (list position -> index, in the scratch block)
    {
      uint16_t *cidx = scr->compidx[l];
      int16_t *ld = labels_dbg ? labels_dbg + ((size_t)scan * CC_NLEV + l) * n_cell : nullptr;
      unsigned keptm_next = 0;
      CC_K2_FOR_ACTIVE({
        unsigned j = CC_COMP_NONE;
        if (LV[c] > l) {
          unsigned v = LAB[c];
          unsigned root_cell = (unsigned)c;
          if (!(v & 0x8000u)) {  // a non-root cell holds its root's cell index
            root_cell = v;
            v = LAB[v];
          }
          if (v & 0x8000u) {  // else: unmarked root, component with < 3 cells (or beyond the capacity)
            j = v & 0x7FFFu;
            const int rr = c / n_col, cc = c - rr * n_col;
            atomicMin(&w_minc[j], (unsigned)cc);
            atomicMax(&w_maxc[j], (unsigned)cc);
            atomicAdd(&w_area[j], 1u);
            if (rr == (int)(root_cell / (unsigned)n_col) + 1) atomicMin(&w_cB[j], (unsigned)cc);
            if (ld) ld[c] = (int16_t)j;
          }
        }
        cidx[i] = (uint16_t)j;
        if (i < CC_K2_OWN * nt) keptm_next |= (j != CC_COMP_NONE ? 1u : 0u) << (i / nt);  // i = tid + u * nt: bit u (folds to a constant shift in the unrolled part)
      })
      keptm = cfg.min_cont_cell_cnt > 3 ? 0u : keptm_next;  // (components dropped below are not in prev_root: no shortcut then)
    }
    __syncthreads();
    // min_cont_cell_cnt_ > 3: the saturating counters only prove ">= 3 cells"; with the exact areas known, drop the
    // components below the bar (stats(n,4) < cfg_.min_cont_cell_cnt_, contour_mng.cpp:303) and renumber the rest
    if (cfg.min_cont_cell_cnt > 3) {
      const int n_before = n_kept;
      n_kept = cc_k2_drop_small(cfg.min_cont_cell_cnt, n_kept, W, roots, LAB, sh + 24, cand, NC);
      if (n_kept != n_before) {  // uniform: the walk's index image follows the renumbering (cand[old] = new or 0x7FFF)
        uint16_t *cidx = scr->compidx[l];
        int16_t *ld = labels_dbg ? labels_dbg + ((size_t)scan * CC_NLEV + l) * n_cell : nullptr;
        CC_K2_FOR_ACTIVE({
          const unsigned j = cidx[i];
          if (j != CC_COMP_NONE) {
            const unsigned jn = cand[j];
            cidx[i] = (uint16_t)jn;
            if (ld) ld[c] = jn == CC_COMP_NONE ? (int16_t)-1 : (int16_t)jn;
          }
        })
        __syncthreads();
      }
    }
    CC_K2_SUBLAP(4);
    CC_K2_LAP(acc_enum);
    // (e) component records; parents of the level above (processed in the previous iteration): index of the root that
    //     owns the child's root cell
    for (int k = tid; k < prev_n; k += nt) {
      unsigned j = cc_lab_comp(LAB, prev_root[k]);
      if (j == CC_COMP_NONE) j = 0xFFFF;
      scr->comp[l + 1][k].parent = (uint16_t)j;
    }
    for (int k = tid; k < n_kept; k += nt) {
      const unsigned root = roots[k];
      cc_comp_t cp;
      cp.root = (uint16_t)root;
      cp.area = (uint16_t)w_area[k];
      cp.parent = 0xFFFF;
      cp.rank = 0;
      cp.r0 = (uint8_t)(root / (unsigned)n_col);
      cp.r1 = 0;
      cp.c0 = (uint8_t)w_minc[k];
      cp.c1 = (uint8_t)w_maxc[k];
      cp.cA = (uint8_t)(root % (unsigned)n_col);
      cp.cB = (uint8_t)w_cB[k];
      cp.pad[0] = cp.pad[1] = 0;
      scr->comp[l][k] = cp;
    }
    __syncthreads();
    for (int k = tid; k < n_kept; k += nt) {
      prev_root[k] = roots[k];
      LAB[roots[k]] = roots[k];  // roots point at themselves again: the next level's labelling continues from here
    }
    if (tid == 0) sh[8 + l] = n_kept;
    prev_n = n_kept;
    __syncthreads();
    CC_K2_SUBLAP(5);
  }
  // More components on a level than this instance numbers: the scan goes to the slow path (cc_k_contours_big), which
  // redoes it with room for CC_NC_BIG per level; nothing of this workgroup's output is kept (block-uniform exit).
  if (!BIG && queue != nullptr && (sh[2] & 2)) {
    if (tid == 0) {
      queue->scan[atomicAdd(&queue->n_flagged, 1)] = scan;
      desc_out[scan].flags = CC_DESC_INEXACT_COMPONENTS;  // stands until the slow path has rewritten the descriptor
    }
    return;
  }
  // ---- (f) raster-order running statistics of every kept component of every level (contour_mng.cpp:317-331): ONE LANE
  //      per component.  The reference adds a component's cells one after the other (f32 cell_vol3_, f64 sums): that chain
  //      is serial, but the ~100-600 components of a scan are independent, so each gets a lane and a wave works on 64 of
  //      them at once.  (Round 3 gave a component a whole wave that found its members with ballots and then accumulated
  //      them on all 64 lanes redundantly: the SIMDs were busy repeating one lane's arithmetic -- 190 of a KITTI-shaped
  //      scan's 580 us.)  First the member lists: per level one wave sweeps the level's index image in list = raster order,
  //      64 entries at a time, and gives every entry its rank inside its component (entries of one component meet through
  //      ballots; a running count per component in LDS) -- a stable counting sort, so every list is in raster order.  Then
  //      every lane walks its component's list, eight positions per 16-byte load (the next load in flight), heights and
  //      continuous positions of the first CC_K2_CACHE active cells from LDS, and finishes with calcStatVals.
  __threadfence_block();
  __syncthreads();
  {
    float *cbev = (float *)R;                                      // [CC_K2_CACHE]
    float2 *cpix = (float2 *)(R + CC_K2_CACHE * 4);                // [CC_K2_CACHE]
    uint16_t *moff = BIG ? bigtab->moff : (uint16_t *)(R + CC_K2_CACHE * 12);  // [6][NC] start of a component's list in memb[l], in units of 8
    uint16_t *mcnt = BIG ? bigtab->mcnt : moff + CC_NLEV * NC;                 // [6][NC] members filed so far
    uint16_t *big = BIG ? bigtab->big : (uint16_t *)(R + 45056);               // [<= n_tot] components left to the eight-lane pass (the levels' working arrays are dead)
    const int n_cache = n_act < CC_K2_CACHE ? n_act : CC_K2_CACHE;
    const bool all_cached = n_act <= CC_K2_CACHE;  // block-uniform: every active cell's height and position sit in LDS
    if (tid == 0) sh[3] = 0;
    for (int i = tid; i < n_cache; i += nt) {
      const int c = (int)scr->act[i];
      cbev[i] = bev[c];
      cpix[i] = pix[c];
    }
    for (int i = tid; i < CC_NLEV * NC; i += nt) mcnt[i] = 0;
    int n_tot = 0, lev_base[CC_NLEV + 1];
    for (int l = 0; l < CC_NLEV; l++) {
      lev_base[l] = n_tot;
      n_tot += sh[8 + l];
    }
    lev_base[CC_NLEV] = n_tot;
    CC_K2_STAMP(10);
    for (int l = wave_id; l < CC_NLEV; l += n_waves) {  // list starts: exclusive prefix sum of the areas, each rounded up to a multiple of 8
      const int n = sh[8 + l];
      int run = 0;
      for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const int a8 = k < n ? ((int)scr->comp[l][k].area + 7) >> 3 : 0;
        const int incl = cc_wave_scan_incl(a8);
        if (k < n) moff[l * NC + k] = (uint16_t)(run + incl - a8);
        run += cc_wave_scan_total(incl);
      }
    }
    __syncthreads();
    for (int l = wave_id; l < CC_NLEV; l += n_waves) {  // member lists, a wave per level
      const uint16_t *cidx = scr->compidx[l];
      uint16_t *memb = scr->memb[l];
      uint16_t *cnt_l = mcnt + l * NC;
      const uint16_t *off_l = moff + l * NC;
      const unsigned long long lt = (1ull << lane) - 1ull;
      unsigned jn = lane < n_act ? (unsigned)cidx[lane] : CC_COMP_NONE;
      for (int b0 = 0; b0 < n_act; b0 += 64) {
        const unsigned j = jn;
        const int i = b0 + lane;
        jn = i + 64 < n_act ? (unsigned)cidx[i + 64] : CC_COMP_NONE;  // the next stretch travels while this one is filed
        // rank of every entry among the stretch's entries of its component, with ballots only (one turn per distinct
        // component, no memory access in the loop); then ONE gather of the components' running counts, the list writes, and
        // the last entry of every component moves its count on -- two LDS round trips per stretch, not per component
        unsigned long long todo = __ballot(j != CC_COMP_NONE);
        int rank = 0, total = 0;
        bool last = false;
        while (todo) {
          const int src = __ffsll(todo) - 1;
          const unsigned j0 = (unsigned)__builtin_amdgcn_readlane((int)j, src);  // src is wave-uniform
          const unsigned long long m = __ballot(j == j0);
          if (j == j0) {
            rank = __popcll(m & lt);
            total = __popcll(m);
            last = (m >> lane) == 1ull;
          }
          todo &= ~m;
        }
        int base = 0;
        if (j != CC_COMP_NONE) {
          base = (int)cnt_l[j];
          memb[(int)off_l[j] * 8 + base + rank] = (uint16_t)i;
        }
        cc_wave_sync();  // every lane has read its count
        if (j != CC_COMP_NONE && last) cnt_l[j] = (uint16_t)(base + total);
        cc_wave_sync();
      }
    }
    __threadfence_block();
    __syncthreads();
    CC_K2_STAMP(11);
    for (int w = tid; w < n_tot; w += nt) {
      int l = 0;
      for (int e = 1; e < CC_NLEV; e++) l += (w >= lev_base[e]) ? 1 : 0;
      int kbase = 0;
      for (int e = 0; e < CC_NLEV; e++) kbase = (e == l) ? lev_base[e] : kbase;
      const int k = w - kbase;
      const int area = (int)mcnt[l * NC + k];  // == comp[l][k].area
      if (area > CC_K2_BIG) {  // left to the eight-lane pass below
        big[atomicAdd(&sh[3], 1)] = (uint16_t)w;
        continue;
      }
      const uint4 *ml = (const uint4 *)(scr->memb[l] + (int)moff[l * NC + k] * 8);
      cc_running_stat rec;
      rec.cnt = 0;
      rec.ps_x = rec.ps_y = rec.t_xx = rec.t_xy = rec.t_yy = rec.tq_x = rec.tq_y = 0.0;
      rec.vol3 = 0.f;
      int poi_i = -1;
      uint4 nx = make_uint4(0u, 0u, 0u, 0u);
      if (area > 0) nx = ml[0];
      if (all_cached) {
        // Straight-line code per eight cells: the eight heights / positions are fetched before any of them is added (eight
        // LDS reads in flight instead of read -> nine additions -> next read: 170 cycles per cell, measured), and the tail of
        // the last stretch is masked to +0.0 instead of branched around -- every sum starts at +0.0 and never becomes -0.0
        // (a sum of round-to-nearest additions is -0.0 only if every term was), so x + 0.0 == x bit for bit.
        for (int m0 = 0; m0 < area; m0 += 8) {
          const uint4 cur = nx;
          if (m0 + 8 < area) nx = ml[(m0 >> 3) + 1];
          const unsigned wds[4] = {cur.x, cur.y, cur.z, cur.w};
          const int nv = area - m0;  // >= 8: all eight are members
          unsigned mk[8];
          float hv[8];
          float2 rv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            mk[u] = (unsigned)-(int)(u < nv);  // all ones for a member, 0 for the list's padding (not initialised)
            const unsigned iu = ((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu) & mk[u];
            hv[u] = cbev[iu];
            rv[u] = cpix[iu];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const float h = __uint_as_float(__float_as_uint(hv[u]) & mk[u]);
            const double vr = (double)__uint_as_float(__float_as_uint(rv[u].x) & mk[u]);
            const double vc = (double)__uint_as_float(__float_as_uint(rv[u].y) & mk[u]);
            rec.ps_x += vr;
            rec.ps_y += vc;
            rec.t_xx += vr * vr;
            rec.t_xy += vr * vc;
            rec.t_yy += vc * vc;
            rec.vol3 += h;
            rec.tq_x += (double)h * vr;
            rec.tq_y += (double)h * vc;
          }
        }
        rec.cnt = area;
        if (area > 0) poi_i = (int)((const uint16_t *)ml)[area - 1];  // the last member in raster order
      } else {
        for (int m0 = 0; m0 < area; m0 += 8) {
          const uint4 cur = nx;
          if (m0 + 8 < area) nx = ml[(m0 >> 3) + 1];
          const unsigned wds[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
          for (int u = 0; u < 8; u++) {
            if (m0 + u < area) {
              const int i = (int)((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu);
              float h;
              float2 rc;
              if (i < n_cache) {
                h = cbev[i];
                rc = cpix[i];
              } else {
                const int c = (int)scr->act[i];
                h = bev[c];
                rc = pix[c];
              }
              const double vr = (double)rc.x, vc = (double)rc.y;
              rec.cnt += 1;
              rec.ps_x += vr;
              rec.ps_y += vc;
              rec.t_xx += vr * vr;
              rec.t_xy += vr * vc;
              rec.t_yy += vc * vc;
              rec.vol3 += h;
              rec.tq_x += (double)h * vr;
              rec.tq_y += (double)h * vc;
              poi_i = i;
            }
          }
        }
      }
      const int pc = poi_i >= 0 ? (int)scr->act[poi_i] : 0;
      cc_contour_t cvw;
      cc_calc_stat_vals(cfg, rec, l, poi_i >= 0 ? pc / n_col : -1, poi_i >= 0 ? pc % n_col : -1, &cvw);
      scr->cont[l][k] = cvw;
    }
    __syncthreads();
    CC_K2_STAMP(12);
    // The large components (a street scene's ground-connected blob holds a few thousand cells): one lane adding nine running
    // values per cell is ~75 cycles per cell whatever the other 63 lanes do, and the phase lasted as long as the largest
    // component.  Every running sum is a sequential chain of its own, so EIGHT LANES share a component, one sum each
    // (a product a * b with (a, b) picked per lane, 1.0 for the plain sums: the same values added in the same order), and
    // the chain per cell shrinks to one f64 multiply and one add; lane 0 of the eight collects the sums and finishes.
    {
      const int n_big = sh[3];
      const int role = tid & 7;
      // role: 0 ps_x  1 ps_y  2 t_xx  3 t_xy  4 t_yy  5 tq_x  6 tq_y  (7: nothing of its own; the f32 height sum is kept by every lane).
      // sum += fa * fb with fa in {h, row, col}, fb in {1, row, col}; the factors are f32 and widened afterwards (exact)
      const unsigned fa_h = role >= 5 ? ~0u : 0u, fa_y = (role == 1 || role == 4) ? ~0u : 0u, fa_x = ~(fa_h | fa_y);
      const unsigned fb_1 = role < 2 ? ~0u : 0u, fb_x = (role == 2 || role == 5) ? ~0u : 0u, fb_y = ~(fb_1 | fb_x);
      for (int g0 = 0; g0 < n_big; g0 += nt >> 3) {  // block-uniform trip count
        const int g = g0 + (tid >> 3);
        const bool on = g < n_big;
        double acc = 0.0;
        float vol3 = 0.f;
        int poi_i = -1, cnt = 0, l = 0, k = 0;
        if (on) {
          const int w = (int)big[g];
          for (int e = 1; e < CC_NLEV; e++) l += (w >= lev_base[e]) ? 1 : 0;
          int kbase = 0;
          for (int e = 0; e < CC_NLEV; e++) kbase = (e == l) ? lev_base[e] : kbase;
          k = w - kbase;
          const int area = (int)mcnt[l * NC + k];
          const uint4 *ml = (const uint4 *)(scr->memb[l] + (int)moff[l * NC + k] * 8);
          uint4 nx = ml[0];
          if (all_cached) {
            // as the lane walk: straight-line, masked tail.  The lane's two factors are picked with bit masks fixed per role (a
            // `role == ...` select in the loop body became a nest of divergent branches with a full wait at every join)
            for (int m0 = 0; m0 < area; m0 += 8) {
              const uint4 cur = nx;
              if (m0 + 8 < area) nx = ml[(m0 >> 3) + 1];
              const unsigned wds[4] = {cur.x, cur.y, cur.z, cur.w};
              const int nv = area - m0;
              unsigned mk[8];
              float hv[8];
              float2 rv[8];
#pragma unroll
              for (int u = 0; u < 8; u++) {
                mk[u] = (unsigned)-(int)(u < nv);
                const unsigned iu = ((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu) & mk[u];
                hv[u] = cbev[iu];
                rv[u] = cpix[iu];
              }
#pragma unroll
              for (int u = 0; u < 8; u++) {
                const unsigned hb = __float_as_uint(hv[u]) & mk[u], xb = __float_as_uint(rv[u].x) & mk[u], yb = __float_as_uint(rv[u].y) & mk[u];
                const float fa = __uint_as_float((hb & fa_h) | (xb & fa_x) | (yb & fa_y));
                const float fb = __uint_as_float((0x3F800000u & fb_1) | (xb & fb_x) | (yb & fb_y));
                acc += (double)fa * (double)fb;  // a padding slot adds (+0.0) * fb = +0.0
                vol3 += __uint_as_float(hb);
              }
            }
            cnt = area;
            poi_i = (int)((const uint16_t *)ml)[area - 1];
          } else {
            for (int m0 = 0; m0 < area; m0 += 8) {
              const uint4 cur = nx;
              if (m0 + 8 < area) nx = ml[(m0 >> 3) + 1];
              const unsigned wds[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
              for (int u = 0; u < 8; u++) {
                if (m0 + u < area) {
                  const int i = (int)((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu);
                  float h;
                  float2 rc;
                  if (i < n_cache) {
                    h = cbev[i];
                    rc = cpix[i];
                  } else {
                    const int c = (int)scr->act[i];
                    h = bev[c];
                    rc = pix[c];
                  }
                  const unsigned hb = __float_as_uint(h), xb = __float_as_uint(rc.x), yb = __float_as_uint(rc.y);
                  const float fa = __uint_as_float((hb & fa_h) | (xb & fa_x) | (yb & fa_y));
                  const float fb = __uint_as_float((0x3F800000u & fb_1) | (xb & fb_x) | (yb & fb_y));
                  acc += (double)fa * (double)fb;
                  vol3 += h;
                  cnt += 1;
                  poi_i = i;
                }
              }
            }
          }
        }
        cc_running_stat rec;
        const int b8 = (tid & 63) & ~7;
        rec.ps_x = __shfl(acc, b8 + 0);
        rec.ps_y = __shfl(acc, b8 + 1);
        rec.t_xx = __shfl(acc, b8 + 2);
        rec.t_xy = __shfl(acc, b8 + 3);
        rec.t_yy = __shfl(acc, b8 + 4);
        rec.tq_x = __shfl(acc, b8 + 5);
        rec.tq_y = __shfl(acc, b8 + 6);
        rec.vol3 = vol3;
        rec.cnt = cnt;
        if (on && role == 0) {
          const int pc = poi_i >= 0 ? (int)scr->act[poi_i] : 0;
          cc_contour_t cvw;
          cc_calc_stat_vals(cfg, rec, l, poi_i >= 0 ? pc / n_col : -1, poi_i >= 0 ? pc % n_col : -1, &cvw);
          scr->cont[l][k] = cvw;
        }
      }
    }
    __syncthreads();
  }
  CC_K2_LAP(acc_walk);
  if (phase_clk && tid == 0) {
    phase_clk[(size_t)scan * CC_K2_NCLK + 1] = acc_ccl;
    phase_clk[(size_t)scan * CC_K2_NCLK + 2] = acc_enum;
    phase_clk[(size_t)scan * CC_K2_NCLK + 3] = acc_walk;
  }
  CC_K2_STAMP(4);
  __threadfence_block();
  __syncthreads();
  int n_lev_f[CC_NLEV];
  for (int l = 0; l < CC_NLEV; l++) n_lev_f[l] = sh[8 + l];
  const int flags_f = sh[2];
  cc_k2_levmap lm;
  lm.LV = LV;
  lm.bitmap = nullptr;
  lm.cbase = nullptr;
  lm.lev = nullptr;
  lm.g_rc = nullptr;
  lm.g_pix = nullptr;
  lm.n_act = 0;
  cc_k2_back<NC, BIG, false>(cfg, pix, k1_out, scr, bigtab, scan, desc_out, labels_dbg, phase_clk, R, n_lev_f, flags_f, lm);
}
