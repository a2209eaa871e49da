// K2, list front half (round 6): labelling of ALL six level sets at once on the raster-ordered list of the scan's active
// cells, everything the passes touch resident in LDS.  Replaces, for scans of up to CC_K2L_NCAP active cells, the front
// half of cc_k2_body (k_contours.h) -- the level loop, the member lists and the statistics walks; the back half
// (cc_k2_back: insertion order, size sort, emit, keys, BCIs) is shared.  Same reference code as k_contours.h:
//   ContourManager::makeContourRecursiveHelper   src/cont2/contour_mng.cpp:274-353
//   RunningStatRecorder / ContourView::calcStatVals   contour.h:48-95,142-255
//
// Why: the original front half walks the levels top-down through one cell-indexed label image -- 7 barrier intervals per
// level, 42 per scan, each bounded by LDS latency and not by work (a level of a nearly empty scan costs ~9 us) -- and keeps
// six cells per thread in registers across that loop (128 VGPR + 160 B of scratch per lane), its per-level index images and
// member lists in a global scratch block (2.5 x the kernel's algorithmic HBM traffic).  Here:
//   * an ENTRY is an active cell (above the lowest level), entries are in raster order; entry i with level count L_i owns
//     the SLOTS off[i] .. off[i] + L_i - 1, one per level set it belongs to (off = prefix sum of the level counts);
//   * the label forest lives in slot space (u16 per slot): level l's components are labelled independently of the other
//     levels' -- one pass over the adjacent pairs links a pair at every level both cells share -- so the whole labelling is
//     ONE union pass, ONE flatten pass, ONE numbering pass (a dozen barrier intervals per scan instead of 42+);
//   * a root is its component's smallest slot = its first cell in raster order, exactly as before, so components are
//     numbered in the same order (OpenCV's label order restated) and every later stage sees the same input;
//   * member lists (stable counting sort per level, one wave per level) and the heights / continuous positions the
//     statistics walks read are in LDS; the only global scratch left are the component records and contour rows the back
//     half reads.
// A scan that does not fit (more than CC_K2L_NCAP active cells, CC_K2L_SCAP slots, CC_NC components on a level, or
// min_cont_cell_cnt_ > 3) is queued for cc_k_contours_mid -- the original body, unchanged -- which in turn queues what IT
// cannot number for cc_k_contours_big.
#pragma once
#include "k_contours.h"

#define CC_K2L_NCAP CC_LIST_CAP                   // entries (active cells) per scan: what K1 lists (k_rasterize.h)
#define CC_K2L_SCAP 12800                         // (entry, level) slots per scan (< 0x8000: labels carry a mark bit)
#ifndef CC_K2L_MEMB
#define CC_K2L_MEMB 11264
#endif
//                        // member-list entries incl. the lists' alignment padding
#define CC_K2L_BIG 64                             // components of this many cells and more are walked by eight lanes
#define CC_K2L_WL_CAP (CC_K2L_MEMB / 2)            // link items the work list holds (the member lists' block, 4 bytes per item)
#define CC_K2L_NSTR (CC_K2L_NCAP / 64)            // 64-entry stretches of the list (<= 64: one lane per stretch in the prefix)
#define CC_K2L_NCHUNK ((CC_MAX_CELLS + 63) / 64)  // 64-cell chunks of the grid
// LDS map (bytes).  Persistent through the back half: bit map, chunk bases, level bytes (the keys' RoI lookups).
#define CC_K2L_O_BITMAP 0      // u64[352]: active cells of chunk b
#define CC_K2L_O_CBASE 2816    // u16[352]: entries before chunk b
#define CC_K2L_O_LEV 3584      // u8[NCAP]: level count of entry i
#define CC_K2L_O_REST 6656     // what follows is the back half's region R (>= CC_K2_R_BYTES) once the walks are done
#define CC_K2L_R_RC 0          // u16[NCAP]: (row << 8) | col of entry i
#define CC_K2L_R_X 6144        // 36 864 B: labels + offsets + bit maps while the labelling runs, then heights + positions
#define CC_K2L_R_MEMB 43008    // u16[MEMB] member lists (stage A: the level bytes per cell, u8[n_cell])
#define CC_K2L_R_AREA (CC_K2L_R_MEMB + 2 * CC_K2L_MEMB)  // u16[6][NC] members per component
#define CC_K2L_R_PTR (CC_K2L_R_AREA + 3840)              // u16[6][NC] list write pointers (level-relative)
#define CC_K2L_R_SH (CC_K2L_R_PTR + 3840)                // int[128] scalars | u8[6][48] kept roots per stretch | u16[96] large components
#define CC_K2L_LDS_BYTES (CC_K2L_O_REST + CC_K2L_R_SH + 1024)
#define CC_K2L_X_OFF (2 * CC_K2L_SCAP)                          // u16[NCAP + 1] slot offsets (behind the labels u16[SCAP])
#define CC_K2L_X_BITA (CC_K2L_X_OFF + 2 * CC_K2L_NCAP + 16)     // u32[SCAP / 32]
#define CC_K2L_X_BITB (CC_K2L_X_BITA + CC_K2L_SCAP / 8)
static_assert(CC_K2L_SCAP / CC_K2L_BIG <= 256 && CC_K2L_BIG == 64 && CC_K2L_SCAP % 32 == 0 && CC_K2L_X_BITA % 4 == 0 && CC_K2L_SCAP < 0x8000 && CC_K2L_NSTR <= 64 && CC_NC * CC_NLEV * 2 == 3840, "list front half: table sizes");
static_assert(CC_K2L_X_BITB + CC_K2L_SCAP / 8 <= 36864 && CC_K2L_NCAP * 12 <= 36864, "list front half: region X / stage A overlays");
static_assert(CC_K2L_R_SH >= CC_K2_R_BYTES, "the back half's region");

#ifdef CC_EMU  // CPU test harness only: say which scans leave the list kernel (tests assert on the path taken)
#define CC_K2L_TRACE_BAIL() do { if (getenv("CC_EMU_TRACE_K2")) fprintf(stderr, "[k2 list] scan %d handed to the mid path (line %d)\n", scan, __LINE__); } while (0)
#else
#define CC_K2L_TRACE_BAIL() do { } while (0)
#endif

__device__ __forceinline__ int cc_k2l_idx_of(const unsigned long long *bitmap, const uint16_t *cbase, int cell) {
  const unsigned long long m = bitmap[cell >> 6];
  const int bit = cell & 63;
  return ((m >> bit) & 1ull) ? (int)cbase[cell >> 6] + __popcll(m & ((1ull << bit) - 1ull)) : -1;
}

// find with path halving: every second cell of the path is pointed at its grandparent (a 16-bit store to a cell that is not a
// root: the CAS of a concurrent link compares the whole 32-bit word and retries).  The list kernel labels every level set from
// single runs, in whatever order the threads come by -- without it a component's tree grows as deep as the component is tall.
__device__ __forceinline__ unsigned cc_uf_find_h(uint16_t *LAB, unsigned x) {
  for (;;) {
    const unsigned p = cc_lds_vread16(LAB + x);
    if (p == x) return x;
    const unsigned gp = cc_lds_vread16(LAB + p);
    if (gp == p) return p;
    cc_lds_vwrite16(LAB + x, gp);
    x = gp;
  }
}
__device__ __forceinline__ void cc_uf_union_h(uint16_t *LAB, unsigned a, unsigned b) {
  a = cc_uf_find_h(LAB, a);
  b = cc_uf_find_h(LAB, b);
  while (a != b) {
    if (a < b) {
      const unsigned t = a;
      a = b;
      b = t;
    }
    unsigned *w = (unsigned *)LAB + (a >> 1);  // a > b: hang root a under b, provided a is still a root
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
    a = cc_uf_find_h(LAB, cur);
    b = cc_uf_find_h(LAB, b);
  }
}

// Returns false when the scan was handed to the mid path (block-uniform); on true n_lev_out[] holds the levels' component
// counts and scr->comp / scr->cont are written (visible after the caller's barrier).
__device__ __forceinline__ bool cc_k2_front_list(const cc_dev_cfg &cfg, const float *__restrict__ bev_in, const float2 *__restrict__ pix_in,
                                                 cc_k2_scratch *__restrict__ scr, cc_k2_big_queue *__restrict__ midq, int scan,
                                                 cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg,
                                                 long long *__restrict__ phase_clk, char *smem, int *n_lev_out, const cc_k1_list_out &list) {
  CC_K2_STAMP(0);
  const int n_cell = cfg.n_cell, n_col = cfg.n_col;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave_id = cc_wave_id(), lane = tid & 63, n_waves = nt >> 6;
  const unsigned long long lane_lt = (1ull << lane) - 1ull;
  const int NC = CC_NC;
#define CC_K2L_BAIL(why_)                                                \
  do {                                                                 \
    if (tid == 0) {                                                    \
      CC_K2L_TRACE_BAIL();                                             \
      midq->scan[atomicAdd(&midq->n_flagged, 1)] = scan;               \
      atomicAdd(&midq->total, 1);                                      \
      atomicAdd(&midq->why[why_], 1);                                  \
      desc_out[scan].flags = CC_DESC_INEXACT_COMPONENTS;               \
    }                                                                  \
    return false;                                                      \
  } while (0)
  if (cfg.min_cont_cell_cnt > 3) CC_K2L_BAIL(0);  // the exact-area renumbering of that configuration lives in the original body only

  unsigned long long *bitmap = (unsigned long long *)(smem + CC_K2L_O_BITMAP);
  uint16_t *cbase = (uint16_t *)(smem + CC_K2L_O_CBASE);
  unsigned char *lev = (unsigned char *)(smem + CC_K2L_O_LEV);
  char *R = smem + CC_K2L_O_REST;
  uint16_t *rc = (uint16_t *)(R + CC_K2L_R_RC);
  char *X = R + CC_K2L_R_X;
  uint16_t *LAB = (uint16_t *)X;
  uint16_t *off = (uint16_t *)(X + CC_K2L_X_OFF);
  unsigned *bitA = (unsigned *)(X + CC_K2L_X_BITA), *bitB = (unsigned *)(X + CC_K2L_X_BITB);
  uint16_t *memb = (uint16_t *)(R + CC_K2L_R_MEMB);
  unsigned *wl = (unsigned *)(R + CC_K2L_R_MEMB);  // stage B: the work list of links (slot a | slot b << 16)
  uint16_t *area = (uint16_t *)(R + CC_K2L_R_AREA);
  uint16_t *ptr = (uint16_t *)(R + CC_K2L_R_PTR);
  int *sh = (int *)(R + CC_K2L_R_SH);
  unsigned char *scnt = (unsigned char *)(R + CC_K2L_R_SH + 512);
  uint16_t *big = (uint16_t *)(R + CC_K2L_R_SH + 512);  // (over scnt: the stretch counts are dead by then) 256 entries

  const float *bev = bev_in + (size_t)scan * n_cell;
  const float2 *pix = pix_in + (size_t)scan * n_cell;
  long long acc_ccl = 0, acc_enum = 0, acc_walk = 0, tmark = phase_clk ? (long long)wall_clock64() : 0, tsub = tmark;
  if (phase_clk && tid == 0)
    for (int j = 0; j < 6; j++) phase_clk[(size_t)scan * CC_K2_NCLK + 16 + j] = 0;

  // ---- (A) the scan's active cells come as a raster-ordered list from K1 (k_rasterize.h: cc_k1_emit -- it has every cell's
  //      height in LDS when it writes the image): (row, col) and level count per entry.  Here: the occupancy bit map and
  //      chunk bases (cell -> entry, for the neighbour and RoI lookups), the slot offsets (prefix of the level counts) and
  //      the initial labels -- every slot starts pointing at the first cell of its horizontal RUN at its level (entries side by
  //      side in one row, all in the level set): per 64-entry stretch the run starts are bit operations on ballots, the
  //      start's slot comes by a shuffle.  The union pass below then only links runs of adjacent rows (and runs that
  //      continue across a stretch border).
  const int4 hd = list.hdr[scan];
  const int n_act = hd.x, n_slot = hd.y;
#ifdef CC_EMU
  if (tid == 0 && getenv("CC_EMU_TRACE_K2")) fprintf(stderr, "[k2 list] scan %d: %d active cells, %d slots\n", scan, n_act, n_slot);
#endif
  if (n_act > CC_K2L_NCAP || n_slot > CC_K2L_SCAP) CC_K2L_BAIL(1);
  const int n_chunk = (n_cell + 63) >> 6;
  const int n_str = (n_act + 63) >> 6;
  uint16_t *ssum = (uint16_t *)scnt;  // slots per stretch (stage A; the kept-root counts of stage D come later)
  {
    const uint16_t *g_rc = list.rc + (size_t)scan * CC_LIST_CAP;
    const unsigned char *g_lev = list.lev + (size_t)scan * CC_LIST_CAP;
    for (int i = tid; i < n_act; i += nt) {
      rc[i] = g_rc[i];
      lev[i] = g_lev[i];
    }
    for (int i = tid; i < n_chunk; i += nt) bitmap[i] = 0ull;
    if (tid < 128) sh[tid] = 0;
    if (labels_dbg)
      for (int i = tid; i < CC_NLEV * n_cell; i += nt) labels_dbg[(size_t)scan * CC_NLEV * n_cell + i] = (int16_t)-1;
  }
  __syncthreads();
  CC_K2_STAMP(22);
  for (int i = tid; i < n_act; i += nt) {
    const unsigned rcv = rc[i];
    const int c = (int)(rcv >> 8) * n_col + (int)(rcv & 255u);
    atomicOr(&bitmap[c >> 6], 1ull << (c & 63));
  }
  for (int q = wave_id; q < n_str; q += n_waves) {
    const int i = q * 64 + lane;
    int v = i < n_act ? (int)lev[i] : 0;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) ssum[q] = (uint16_t)v;
  }
  __syncthreads();
  CC_K2_STAMP(23);
  {
    const unsigned long long lane_le = lane_lt | (1ull << lane);
    int spre;  // slots before stretch `lane` (every wave makes the prefix for itself: <= 48 stretches, one lane each)
    {
      const int v = lane < n_str ? (int)ssum[lane] : 0;
      const int incl = cc_wave_scan_incl(v);
      spre = incl - v;
    }
    // entries before every 64-cell chunk (every wave makes all of it and writes the same values: no hand-over)
    for (int q = 0, run = 0; q < n_chunk; q += 64) {
      const int b = q + lane;
      const int v = b < n_chunk ? __popcll(bitmap[b]) : 0;
      const int incl = cc_wave_scan_incl(v);
      if (b < n_chunk) cbase[b] = (uint16_t)(run + incl - v);
      run += cc_wave_scan_total(incl);
    }
    CC_K2_STAMP(24);
    for (int q = wave_id; q < n_str; q += n_waves) {
      const int i = q * 64 + lane;
      const bool valid = i < n_act;
      const int lvc = valid ? (int)lev[i] : 0;
      const unsigned rcv = valid ? (unsigned)rc[i] : 0xFFFFu;
      const int incl = cc_wave_scan_incl(lvc);
      const int so = __builtin_amdgcn_readlane(spre, q) + incl - lvc;
      if (valid) off[i] = (uint16_t)so;
      // entry x continues entry x - 1's run: same row, next column (lane 0 never does: a run that crosses the stretch border is
      // linked in the union pass)
      const unsigned rcp = (unsigned)__shfl_up((int)rcv, 1);
      const unsigned long long adj = __ballot(valid && lane > 0 && rcv == rcp + 1u);
      int so_start[CC_NLEV];
#pragma unroll
      for (int k = 0; k < CC_NLEV; k++) {  // (the six levels without a branch between them: their shuffles are in flight together)
        const unsigned long long mk = __ballot(lvc > k);
        const unsigned long long starts = mk & ~((mk << 1) & adj);
        const int ys = 63 - __clzll((long long)(starts & lane_le));  // (lanes outside the level set: any lane, value unused)
        so_start[k] = __shfl(so, ys & 63);
      }
#pragma unroll
      for (int k = 0; k < CC_NLEV; k++)
        if (lvc > k) LAB[so + k] = (uint16_t)(so_start[k] + k);
    }
  }
  if (tid == 0) off[n_act] = (uint16_t)n_slot;
  __syncthreads();
  CC_K2_STAMP(9);
  tmark = phase_clk ? (long long)wall_clock64() : 0;
  tsub = tmark;

  // ---- (B) 8-connected labelling of all level sets: one union per (adjacent pair, level both cells are in).  Backward
  //      neighbours only (W, NW, N, NE): W is the previous entry if its cell is; the row above through the bit map.  Labels
  //      are slot indices, parents point to smaller slots, so a root is its component's first cell in raster order whatever
  //      the order of the unions.
  for (int i = tid; i < CC_NLEV * NC / 2; i += nt) ((unsigned *)area)[i] = 0u;  // (stage A's level bytes reached into this table)
  for (int i = tid; i < 2 * (CC_K2L_SCAP / 32); i += nt) bitA[i] = 0u;           // bitA and bitB are adjacent (stage A's chunk tables lay here)
  for (int i = tid; i < n_act; i += nt) {
    const unsigned rcv = rc[i];
    const int r = (int)(rcv >> 8), cc = (int)(rcv & 255u);
    const int c = r * n_col + cc;
    const int Li = (int)lev[i], oi = (int)off[i];
    // Which links this cell has to make (the others are somebody else's or already there):
    //  * W: inside a 64-entry stretch the run labels have it; a run that continues across a stretch border is linked here;
    //  * row above: if N is in the level set, NW and NE (when they are) belong to N's run -- one link; else NW and NE each;
    //  * of the cells of my run that touch the same run above only the leftmost links: with W in my run and NW in the level
    //    set, W's own link (its N is my NW) has joined the two runs.
    // NW, N, NE are three consecutive cells: one look at the occupancy words (the field may straddle two of them) and one
    // chunk base give all three entry indices -- the list is in raster order, so they are consecutive among the active ones.
    int sh_w = 0, o_w = 0, sh_nw = 0, o_nw = 0, sh_n = 0, o_n = 0, sh_ne = 0, o_ne = 0;
    if (cc > 0 && i > 0 && (unsigned)rc[i - 1] == rcv - 1u) {
      const int Lj = (int)lev[i - 1];
      sh_w = Lj < Li ? Lj : Li;
      o_w = (int)off[i - 1];
    }
    if (r > 0) {
      const int q0 = c - n_col - 1;        // the NW cell; -1 for the first cell of row 1 (then the field starts at N)
      const int qb = q0 < 0 ? 0 : q0;
      const int wb = qb >> 6, bit = qb & 63;
      const unsigned long long w0 = bitmap[wb], w1 = bitmap[wb + 1 < CC_K2L_NCHUNK ? wb + 1 : wb];
      unsigned raw = (unsigned)(w0 >> bit);
      if (bit > 61) raw |= (unsigned)(w1 << (64 - bit));
      raw = q0 < 0 ? (raw << 1) & 6u : raw & 7u;  // bit 0 NW, 1 N, 2 NE
      unsigned f3 = raw;
      if (cc == 0) f3 &= 6u;          // no NW (the cell there is the previous row's last)
      if (cc == n_col - 1) f3 &= 3u;  // no NE
      if (f3) {
        const int j_nw = (int)cbase[wb] + __popcll(w0 & ((1ull << bit) - 1ull));  // entries before the field's first cell
        const int j_n = j_nw + (int)(raw & 1u), j_ne = j_n + (int)((raw >> 1) & 1u);
        if (f3 & 1u) {
          const int Lj = (int)lev[j_nw];
          sh_nw = Lj < Li ? Lj : Li;
          o_nw = (int)off[j_nw];
        }
        if (f3 & 2u) {
          const int Lj = (int)lev[j_n];
          sh_n = Lj < Li ? Lj : Li;
          o_n = (int)off[j_n];
        }
        if (f3 & 4u) {
          const int Lj = (int)lev[j_ne];
          sh_ne = Lj < Li ? Lj : Li;
          o_ne = (int)off[j_ne];
        }
      }
    }
    // the levels at which each link is this cell's to make, as bit masks (bit l = level l)
    const unsigned b_w = (1u << sh_w) - 1u, b_nw = (1u << sh_nw) - 1u, b_n = (1u << sh_n) - 1u, b_ne = (1u << sh_ne) - 1u;
    unsigned m_w = ((i & 63) == 0) ? b_w : 0u;
    unsigned m_n = b_n & ~(b_w & b_nw);
    unsigned m_nw = b_nw & ~b_n & ~b_w;
    unsigned m_ne = b_ne & ~b_n;
    // The links go on a work list (slot pair per item) that the whole workgroup works off below: linked here, a wave would run
    // the union code whenever ONE of its lanes has a link to make, at every turn of every lane's level loop.
    const int cnt = __popc(m_w) + __popc(m_n) + __popc(m_nw) + __popc(m_ne);
    if (cnt) {
      int pos = atomicAdd(&sh[5], cnt);
      const bool fits = pos + cnt <= CC_K2L_WL_CAP;
#define CC_K2L_LINKS(mask_, o_)                                                                       \
  while (mask_) {                                                                                     \
    const int l_ = __ffs((int)mask_) - 1;                                                             \
    mask_ &= mask_ - 1u;                                                                              \
    const unsigned a_ = (unsigned)(oi + l_), b_ = (unsigned)((o_) + l_);                              \
    if (fits)                                                                                         \
      wl[pos++] = a_ | (b_ << 16);                                                                    \
    else {                                                                                            \
      if (pos < CC_K2L_WL_CAP) wl[pos] = 0u; /* a no-op item where the list still had room */         \
      pos++;                                                                                          \
      if (cc_lds_vread16(LAB + a_) != cc_lds_vread16(LAB + b_)) cc_uf_union_h(LAB, a_, b_);           \
    }                                                                                                 \
  }
      CC_K2L_LINKS(m_w, o_w)
      CC_K2L_LINKS(m_n, o_n)
      CC_K2L_LINKS(m_nw, o_nw)
      CC_K2L_LINKS(m_ne, o_ne)
#undef CC_K2L_LINKS
    }
  }
  __syncthreads();
  {
    const int n_items = sh[5] < CC_K2L_WL_CAP ? sh[5] : CC_K2L_WL_CAP;
    for (int j = tid; j < n_items; j += nt) {
      const unsigned it = wl[j];
      const unsigned a_ = it & 0xFFFFu, b_ = it >> 16;
      // two slots with the same parent are in one tree already: two independent reads instead of two finds
      if (cc_lds_vread16(LAB + a_) != cc_lds_vread16(LAB + b_)) cc_uf_union_h(LAB, a_, b_);
    }
  }
  __syncthreads();
  CC_K2_SUBLAP(0);
  // ---- (C) every slot is pointed at its root (an entry's finds advance hop by hop together); which roots own >= 3 cells:
  //      a member that is not the root sets the root's bit in A, and in B if A was set already
  const int need = cfg.min_cont_cell_cnt < 3 ? cfg.min_cont_cell_cnt : 3;
  for (int i = tid; i < n_act; i += nt) {
    const int Li = (int)lev[i], oi = (int)off[i];
    unsigned x[CC_NLEV];
#pragma unroll
    for (int l = 0; l < CC_NLEV; l++) x[l] = (unsigned)(oi + l);
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int l = 0; l < CC_NLEV; l++) {
        const unsigned pq = l < Li ? cc_lds_vread16(LAB + x[l]) : x[l];
        more |= pq != x[l];
        x[l] = pq;
      }
    }
#pragma unroll
    for (int l = 0; l < CC_NLEV; l++) {
      if (l < Li && x[l] != (unsigned)(oi + l)) {
        const unsigned rt = x[l];
        LAB[oi + l] = (uint16_t)rt;
        const unsigned bit = 1u << (rt & 31u);
        if (atomicOr(&bitA[rt >> 5], bit) & bit) atomicOr(&bitB[rt >> 5], bit);
      }
    }
  }
  __syncthreads();
  CC_K2_SUBLAP(1);
  CC_K2_LAP(acc_ccl);
  // ---- (D) kept roots, numbered per level in raster order of their first cells: ballots per 64-entry stretch (a wave
  //      takes every n_waves-th stretch), a prefix over the <= 48 stretches per level in registers (one lane per stretch,
  //      every wave makes it for itself), then the same ballots again give the numbers.
#define CC_K2L_KEPT(s_) (need <= 1 || (((need == 2 ? bitA : bitB)[(s_) >> 5] >> ((s_) & 31)) & 1u))
  for (int q = wave_id; q < n_str; q += n_waves) {
    const int i = q * 64 + lane;
    const int Li = i < n_act ? (int)lev[i] : 0, oi = i < n_act ? (int)off[i] : 0;
#pragma unroll
    for (int l = 0; l < CC_NLEV; l++) {
      const bool kp = l < Li && (unsigned)LAB[oi + l] == (unsigned)(oi + l) && CC_K2L_KEPT(oi + l);
      const unsigned long long m = __ballot(kp);
      if (lane == 0) scnt[l * CC_K2L_NSTR + q] = (unsigned char)__popcll(m);
    }
  }
  __syncthreads();
  CC_K2_SUBLAP(2);
  int nk[CC_NLEV], pre[CC_NLEV];
  bool too_many = false;
#pragma unroll
  for (int l = 0; l < CC_NLEV; l++) {
    const int v = lane < n_str ? (int)scnt[l * CC_K2L_NSTR + lane] : 0;
    const int incl = cc_wave_scan_incl(v);
    pre[l] = incl - v;
    nk[l] = cc_wave_scan_total(incl);
    too_many = too_many || nk[l] > NC;
  }
  if (too_many) CC_K2L_BAIL(2);  // more components on a level than the tables hold: the mid path decides (and queues for the big one)
  for (int q = wave_id; q < n_str; q += n_waves) {
    const int i = q * 64 + lane;
    const int Li = i < n_act ? (int)lev[i] : 0, oi = i < n_act ? (int)off[i] : 0;
#pragma unroll
    for (int l = 0; l < CC_NLEV; l++) {
      const bool kp = l < Li && (unsigned)LAB[oi + l] == (unsigned)(oi + l) && CC_K2L_KEPT(oi + l);
      const unsigned long long m = __ballot(kp);
      const int base = __builtin_amdgcn_readlane(pre[l], q);
      if (kp) LAB[oi + l] = (uint16_t)(0x8000u | (unsigned)(base + cc_mbcnt(m)));
    }
  }
  __syncthreads();
  CC_K2_SUBLAP(3);
  // ---- (E) component index of every slot, in place (a kept root carries 0x8000 | index; a member of a kept component
  //      reads it from its root and carries it from here on); members counted per component; a root's parent is the
  //      component its own cell belongs to one level down (the same entry's previous slot)
  for (int i = tid; i < n_act; i += nt) {
    const int Li = (int)lev[i], oi = (int)off[i];
    int16_t *ld = nullptr;
    int cell = 0;
    if (labels_dbg) {
      const unsigned rcv = rc[i];
      cell = (int)(rcv >> 8) * n_col + (int)(rcv & 255u);
      ld = labels_dbg + (size_t)scan * CC_NLEV * n_cell;
    }
    unsigned jprev = 0xFFFFu;
    for (int l = 0; l < Li; l++) {
      const unsigned v = LAB[oi + l];
      unsigned j = CC_COMP_NONE;
      if (v & 0x8000u) {  // a kept root (only its owner -- this thread -- ever writes the slot)
        j = v & 0x7FFFu;
        if (l > 0) scr->comp[l][j].parent = (uint16_t)jprev;
      } else if (v != (unsigned)(oi + l)) {
        const unsigned rv = LAB[v];  // roots are not rewritten in this pass
        if (rv & 0x8000u) {
          j = rv & 0x7FFFu;
          LAB[oi + l] = (uint16_t)rv;
        }
      }
      if (j != CC_COMP_NONE) {
        const unsigned t = (unsigned)(l * NC) + j;
        atomicAdd((unsigned *)area + (t >> 1), 1u << ((t & 1u) * 16));
        if (ld) ld[(size_t)l * n_cell + cell] = (int16_t)j;
      }
      jprev = j == CC_COMP_NONE ? 0xFFFFu : j;
    }
  }
  __syncthreads();
  CC_K2_SUBLAP(4);
  CC_K2_LAP(acc_enum);
  // ---- (F) member lists: list starts (prefix of the areas, each rounded up to two entries: 4-byte aligned lists), then a
  //      wave per level sweeps the list, 64 entries at a time, and gives every member its rank inside its component
  //      (entries of one component meet through ballots; a running write pointer per component) -- a stable counting sort,
  //      so every list is in raster order
#define CC_K2L_NK(l_) ((l_) == 0 ? nk[0] : (l_) == 1 ? nk[1] : (l_) == 2 ? nk[2] : (l_) == 3 ? nk[3] : (l_) == 4 ? nk[4] : nk[5])
  CC_K2_STAMP(10);
  for (int l = wave_id; l < CC_NLEV; l += n_waves) {
    const int n = CC_K2L_NK(l);
    int run = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
      const int k = k0 + lane;
      const int a = k < n ? (int)area[l * NC + k] : 0;
      const int a4 = (a + 1) & ~1;  // lists start on a 4-byte boundary
      const int incl = cc_wave_scan_incl(a4);
      if (k < n) {
        ptr[l * NC + k] = (uint16_t)(run + incl - a4);
        // size class = floor(log2 area), 7 = CC_K2L_BIG cells and more (eight-lane walk): counted here, placed below -- the
        // lane walk takes the components largest first, so the 64 lanes of a wave walk lists of similar length
        const int cls = a >= CC_K2L_BIG ? 7 : 31 - __clz(a);
        atomicAdd(&sh[32 + cls], 1);
      }
      run += cc_wave_scan_total(incl);
    }
    if (lane == 0) sh[16 + l] = run;
  }
  __syncthreads();
  int lbase[CC_NLEV + 1];
  lbase[0] = 0;
#pragma unroll
  for (int l = 0; l < CC_NLEV; l++) lbase[l + 1] = lbase[l] + sh[16 + l];
  if (lbase[CC_NLEV] > CC_K2L_MEMB) CC_K2L_BAIL(3);  // (the padding of very many tiny components)
#define CC_K2L_LBASE(l_) ((l_) == 0 ? lbase[0] : (l_) == 1 ? lbase[1] : (l_) == 2 ? lbase[2] : (l_) == 3 ? lbase[3] : (l_) == 4 ? lbase[4] : lbase[5])
  // the heights and continuous positions the walks read: requested now (a thread's entries), stored behind the sweep's
  // barrier -- they go where the labels are
  float ch[CC_K2L_NCAP / CC_K2_BLOCK];
  float2 cp[CC_K2L_NCAP / CC_K2_BLOCK];
  static_assert(CC_K2L_NCAP % CC_K2_BLOCK == 0, "entries per thread");
#pragma unroll
  for (int u = 0; u < CC_K2L_NCAP / CC_K2_BLOCK; u++) {
    const int i = tid + u * CC_K2_BLOCK;
    ch[u] = 0.f;
    cp[u] = make_float2(0.f, 0.f);
    if (i < n_act) {  // (coalesced: K1 wrote them in list order)
      ch[u] = list.h[(size_t)scan * CC_LIST_CAP + i];
      cp[u] = list.pix[(size_t)scan * CC_LIST_CAP + i];
    }
  }
  // walk order: position of every component in the size-ordered sequence (big ones first in their own list), in the scan's
  // scratch block (2 bytes per component; read back behind two barriers)
  {
    int cbase_[7];
    cbase_[6] = 0;  // (classes 6 and 7 are the same list now: CC_K2L_BIG = 2^6)
#pragma unroll
    for (int cI = 5; cI >= 0; cI--) cbase_[cI] = cbase_[cI + 1] + (cI == 5 ? 0 : sh[32 + cI + 1]);  // classes 5 .. 0 behind each other
    // (by the waves the sweep below leaves idle -- there are six levels -- or by everybody if the workgroup has no such waves)
    const int t0 = n_waves > CC_NLEV ? tid - CC_NLEV * 64 : tid, tn = n_waves > CC_NLEV ? nt - CC_NLEV * 64 : nt;
    if (t0 >= 0)
      for (int l = 0; l < CC_NLEV; l++) {
        for (int k = t0; k < CC_K2L_NK(l); k += tn) {
          const int a = (int)area[l * NC + k];
          const int cls = a >= CC_K2L_BIG ? 7 : 31 - __clz(a);
          const int p = atomicAdd(&sh[40 + cls], 1);
          if (cls == 7)
            big[p] = (uint16_t)(l * NC + k);  // <= SCAP / CC_K2L_BIG = 200 of them
          else
            scr->act[(cls == 0 ? cbase_[0] : cls == 1 ? cbase_[1] : cls == 2 ? cbase_[2] : cls == 3 ? cbase_[3] : cls == 4 ? cbase_[4] : cbase_[5]) + p] = (uint16_t)(l * NC + k);
        }
      }
  }
  for (int l = wave_id; l < CC_NLEV; l += n_waves) {
    uint16_t *ml = memb + CC_K2L_LBASE(l);
    uint16_t *ptr_l = ptr + l * NC;
    const int n_bits = 32 - __clz(CC_K2L_NK(l) > 1 ? CC_K2L_NK(l) - 1 : 1);  // bits of this level's component indices (wave-uniform)
    auto comp_of = [&](int i) -> unsigned {
      if (i >= n_act || (int)lev[i] <= l) return CC_COMP_NONE;
      const unsigned v = LAB[(int)off[i] + l];
      return (v & 0x8000u) ? (v & 0x7FFFu) : CC_COMP_NONE;
    };
    unsigned jn = comp_of(lane);
    for (int b0 = 0; b0 < n_act; b0 += 64) {
      const unsigned j = jn;
      const int i = b0 + lane;
      jn = comp_of(i + 64);  // the next stretch travels while this one is filed
      // the stretch's entries of one component find each other without a loop over the components: M = lanes whose index
      // agrees with mine in every bit (a ballot per index bit; a loop over the distinct components was 15 turns of dependent
      // scalar work per stretch on a street scene, 30 us of the scan)
      const bool valid = j != CC_COMP_NONE;
      unsigned long long M = __ballot(valid);
      if (M == 0ull) continue;  // wave-uniform: nothing of this level in the stretch (the upper levels are sparse)
      static_assert(CC_NC <= 512, "nine index bits");
      for (int bq = 0; bq < n_bits; bq++) {
        const bool bit = (j >> bq) & 1u;
        const unsigned long long B = __ballot(valid && bit);
        M &= bit ? B : ~B;
      }
      const int rank = cc_mbcnt(M), total = __popcll(M);
      const bool last = (M >> lane) == 1ull;
      int base = 0;
      if (j != CC_COMP_NONE) {
        base = (int)ptr_l[j];
        ml[base + rank] = (uint16_t)i;
      }
      cc_wave_sync();  // every lane has read its pointer
      if (j != CC_COMP_NONE && last) ptr_l[j] = (uint16_t)(base + total);
      cc_wave_sync();
    }
  }
  __threadfence_block();
  __syncthreads();
  float *cbev = (float *)X;                              // [NCAP]
  float2 *cpix = (float2 *)(X + CC_K2L_NCAP * 4);        // [NCAP]
#pragma unroll
  for (int u = 0; u < CC_K2L_NCAP / CC_K2_BLOCK; u++) {
    const int i = tid + u * CC_K2_BLOCK;
    if (i < n_act) {
      cbev[i] = ch[u];
      cpix[i] = cp[u];
    }
  }
  __syncthreads();
  CC_K2_STAMP(11);
  // ---- (G) raster-order running statistics (contour_mng.cpp:317-331): ONE LANE per component walks its member list, eight
  //      members per step -- their heights / positions fetched before any is added, the tail of the last step masked to +0.0
  //      (every sum starts at +0.0 and never becomes -0.0, so x + 0.0 == x bit for bit) -- and finishes with calcStatVals;
  //      the lane also finds what the insertion order needs of the component's shape: first row and its first column (the
  //      first member), smallest column, first member column of the second row.
  int lev_base[CC_NLEV + 1];
  lev_base[0] = 0;
#pragma unroll
  for (int l = 0; l < CC_NLEV; l++) lev_base[l + 1] = lev_base[l] + nk[l];
  const int n_tot = lev_base[CC_NLEV];
  auto finish_stats = [&](int l, int k, int a, const uint16_t *ml, const cc_running_stat &rec) {
    const unsigned rcl = rc[ml[a - 1]];  // last member in raster order = poi_
    cc_contour_t cvw;
    cc_calc_stat_vals(cfg, rec, l, (int)(rcl >> 8), (int)(rcl & 255u), &cvw);
    scr->cont[l][k] = cvw;
  };
  auto finish_shape = [&](int l, int k, int a, const uint16_t *ml, int c0, int c1, int cB) {
    const unsigned rc0 = rc[ml[0]];     // first member = the root
    cc_comp_t *cpo = &scr->comp[l][k];  // (.parent was written in (E))
    const unsigned root_cell = (rc0 >> 8) * (unsigned)n_col + (rc0 & 255u);
    *(unsigned *)cpo = root_cell | ((unsigned)a << 16);
    cpo->rank = 0;
    cpo->r0 = (uint8_t)(rc0 >> 8);
    cpo->r1 = 0;
    cpo->c0 = (uint8_t)c0;
    cpo->c1 = (uint8_t)c1;
    cpo->cA = (uint8_t)(rc0 & 255u);
    cpo->cB = (uint8_t)cB;
    cpo->pad[0] = cpo->pad[1] = 0;
  };
  // The two kinds of walk run SIDE BY SIDE (round 6: one after the other before, 16 + 11 us per street scene): the lower half of the
  // workgroup takes the lane walks -- the ~500 components largest first, so its first round holds everything of any length --
  // and the large components' shapes, the upper half the eight-lane walks of the large components (a few dozen).
  const int n_big = sh[40 + 7];
  const int half = nt >> 1;
  const bool lane_walker = tid < half || nt < 128;
  const int wt = nt < 128 ? tid : (tid < half ? tid : tid - half), wn = nt < 128 ? nt : half;  // index / count inside the thread's half
  if (lane_walker)
  for (int g = wt; g < n_big; g += wn) {  // the large components' shape (their sums: the eight-lane pass)
    const int t_ = (int)big[g];
    const int l = t_ / NC, k = t_ - l * NC;
    const int a = (int)area[t_];
    const uint16_t *ml = memb + CC_K2L_LBASE(l) + (int)ptr[t_] - a;
    const unsigned *mlw = (const unsigned *)ml;
    const int row1 = (int)(rc[ml[0]] >> 8) + 1;
    int c0 = 255, c1 = 0, cB = 255;
    for (int m0 = 0; m0 < a; m0 += 8) {
      const unsigned wds[4] = {mlw[m0 >> 1], mlw[(m0 >> 1) + 1], mlw[(m0 >> 1) + 2], mlw[(m0 >> 1) + 3]};  // (reads past the list's end stay inside the LDS block)
      const int nv = a - m0;
      unsigned rcu[8];
#pragma unroll
      for (int u = 0; u < 8; u++) rcu[u] = rc[((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu) & (unsigned)-(int)(u < nv)];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int col = (int)(rcu[u] & 255u);
        if (u < nv) {
          c0 = col < c0 ? col : c0;
          c1 = col > c1 ? col : c1;
          cB = ((int)(rcu[u] >> 8) == row1 && col < cB) ? col : cB;
        }
      }
    }
    finish_shape(l, k, a, ml, c0, c1, cB);
  }
  const int n_small = n_tot - n_big;
  if (lane_walker)
  for (int w = wt; w < n_small; w += wn) {
    const int t_ = (int)scr->act[w];
    const int l = t_ / NC, k = t_ - l * NC;
    const int a = (int)area[t_];
    const uint16_t *ml = memb + CC_K2L_LBASE(l) + (int)ptr[t_] - a;
    const unsigned *mlw = (const unsigned *)ml;
    cc_running_stat rec;
    rec.cnt = a;
    rec.ps_x = rec.ps_y = rec.t_xx = rec.t_xy = rec.t_yy = rec.tq_x = rec.tq_y = 0.0;
    rec.vol3 = 0.f;
    const int row1 = (int)(rc[ml[0]] >> 8) + 1;
    int c0 = 255, c1 = 0, cB = 255;
    unsigned nx[4] = {mlw[0], mlw[1], mlw[2], mlw[3]};  // (reads past the list's end stay inside the LDS block)
    for (int m0 = 0; m0 < a; m0 += 8) {
      const unsigned wds[4] = {nx[0], nx[1], nx[2], nx[3]};
      if (m0 + 8 < a) {
#pragma unroll
        for (int u = 0; u < 4; u++) nx[u] = mlw[(m0 >> 1) + 4 + u];
      }
      const int nv = a - m0;
      unsigned mk[8], rcu[8];
      float hv[8];
      float2 rv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        mk[u] = (unsigned)-(int)(u < nv);  // all ones for a member, 0 for the list's padding (not initialised)
        const unsigned iu = ((wds[u >> 1] >> ((u & 1) * 16)) & 0xFFFFu) & mk[u];
        hv[u] = cbev[iu];
        rv[u] = cpix[iu];
        rcu[u] = rc[iu];
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
        const int col = (int)(rcu[u] & 255u);
        const int colm = mk[u] ? col : 255;
        c0 = colm < c0 ? colm : c0;
        c1 = (mk[u] && col > c1) ? col : c1;
        cB = (mk[u] && (int)(rcu[u] >> 8) == row1 && cB == 255) ? col : cB;
      }
    }
    finish_stats(l, k, a, ml, rec);
    finish_shape(l, k, a, ml, c0, c1, cB);
  }
  CC_K2_STAMP(12);
  // The large components (a street scene's ground-connected blob): EIGHT LANES share one, one running sum each (a product
  // a * b with (a, b) picked per lane, 1.0 for the plain sums: the same values added in the same order), the f32 height sum
  // by every lane; lane 0 of the eight collects the sums and finishes.
  {
    const int role = tid & 7;
    // role: 0 ps_x  1 ps_y  2 t_xx  3 t_xy  4 t_yy  5 tq_x  6 tq_y  (7: nothing of its own)
    const unsigned fa_h = role >= 5 ? ~0u : 0u, fa_y = (role == 1 || role == 4) ? ~0u : 0u, fa_x = ~(fa_h | fa_y);
    const unsigned fb_1 = role < 2 ? ~0u : 0u, fb_x = (role == 2 || role == 5) ? ~0u : 0u, fb_y = ~(fb_1 | fb_x);
    if (!lane_walker || nt < 128)
    for (int g0 = 0; g0 < n_big; g0 += wn >> 3) {  // uniform trip count over the waves that are here
      const int g = g0 + (wt >> 3);
      const bool on = g < n_big;
      double acc = 0.0;
      float vol3 = 0.f;
      int l = 0, k = 0, a = 0;
      const uint16_t *ml = memb;
      if (on) {
        const int t = (int)big[g];
        l = t / NC;
        k = t - l * NC;
        a = (int)area[t];
        ml = memb + CC_K2L_LBASE(l) + (int)ptr[t] - a;
        const unsigned *mlw = (const unsigned *)ml;
        unsigned nx[4] = {mlw[0], mlw[1], mlw[2], mlw[3]};
        for (int m0 = 0; m0 < a; m0 += 8) {
          const unsigned wds[4] = {nx[0], nx[1], nx[2], nx[3]};
          if (m0 + 8 < a) {
#pragma unroll
            for (int u = 0; u < 4; u++) nx[u] = mlw[(m0 >> 1) + 4 + u];
          }
          const int nv = a - m0;
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
      rec.cnt = a;
      if (on && role == 0) finish_stats(l, k, a, ml, rec);
    }
  }
  CC_K2_LAP(acc_walk);
  if (phase_clk && tid == 0) {
    phase_clk[(size_t)scan * CC_K2_NCLK + 1] = acc_ccl;
    phase_clk[(size_t)scan * CC_K2_NCLK + 2] = acc_enum;
    phase_clk[(size_t)scan * CC_K2_NCLK + 3] = acc_walk;
  }
  CC_K2_STAMP(4);
#pragma unroll
  for (int l = 0; l < CC_NLEV; l++) n_lev_out[l] = nk[l];
  return true;
#undef CC_K2L_BAIL
#undef CC_K2L_KEPT
#undef CC_K2L_NK
#undef CC_K2L_LBASE
}

// 4 waves per SIMD = two 512-thread workgroups (scans) per CU: at most 128 VGPRs
__global__ void __launch_bounds__(CC_K2_BLOCK, 4)
cc_k_contours(cc_dev_cfg cfg, const float *__restrict__ bev_in, const float2 *__restrict__ pix_in,
              const cc_k1_scan_out *__restrict__ k1_out, cc_k2_scratch *__restrict__ scratch_all,
              cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg, long long *__restrict__ phase_clk,
              cc_k2_big_queue *__restrict__ midq, cc_k1_list_out list) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int scan = (int)blockIdx.x;
  cc_k2_scratch *scr = scratch_all + scan;
  int n_lev[CC_NLEV];
  if (!cc_k2_front_list(cfg, bev_in, pix_in, scr, midq, scan, desc_out, labels_dbg, phase_clk, smem, n_lev, list)) return;
  __threadfence_block();
  __syncthreads();
  cc_k2_levmap lm;
  lm.LV = nullptr;
  lm.bitmap = (const unsigned long long *)(smem + CC_K2L_O_BITMAP);
  lm.cbase = (const uint16_t *)(smem + CC_K2L_O_CBASE);
  lm.lev = (const unsigned char *)(smem + CC_K2L_O_LEV);
  lm.g_rc = list.rc + (size_t)scan * CC_LIST_CAP;
  lm.g_pix = list.pix + (size_t)scan * CC_LIST_CAP;
  lm.n_act = list.hdr[scan].x;
  cc_k2_back<CC_NC, false, true>(cfg, pix_in + (size_t)scan * cfg.n_cell, k1_out, scr, nullptr, scan, desc_out, labels_dbg, phase_clk,
                                 smem + CC_K2L_O_REST, n_lev, 0, lm);
}

// The scans the list kernel handed on (more active cells / slots / components than its LDS tables hold, or
// min_cont_cell_cnt_ > 3): the original body with its cell-indexed label image, a scan at a time per workgroup.  Launched
// behind every list launch; with an empty queue it ends at once.  What IT cannot number goes on to cc_k_contours_big.
__global__ void __launch_bounds__(CC_K2_BLOCK, 2)
cc_k_contours_mid(cc_dev_cfg cfg, float *bev_io, float2 *pix_io /*written here for a scan K1 left without them, then read*/,
                  const cc_k1_scan_out *__restrict__ k1_out, cc_k2_scratch *__restrict__ scratch_all, cc_k2_big_queue *__restrict__ midq,
                  cc_k2_big_queue *__restrict__ bigq, cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg,
                  int *__restrict__ seen /*pinned host memory: how many scans this launch found queued (the host sizes the NEXT launch by it)*/,
                  cc_k1_list_out list) {
  HIP_DYNAMIC_SHARED(char, smem)
  __shared__ int s_next;
  // An empty queue (the usual case) is left alone: its counters are zero already, and the two atomics every workgroup would
  // spend on finding that out are served one after the other (512 workgroups: 12 us behind every ingest launch).  A queue
  // that holds scans is reset only after EVERY workgroup has counted itself out, so none can see the zero of the reset here.
  if (midq->n_flagged == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *seen = 0;
    return;
  }
  for (;;) {
    __syncthreads();  // the previous scan's LDS is no longer read
    if (threadIdx.x == 0) s_next = atomicAdd(&midq->next, 1);
    __syncthreads();
    const int k = s_next;
    if (k >= midq->n_flagged) {
      if (threadIdx.x == 0 && atomicAdd(&midq->exited, 1) == (int)gridDim.x - 1) {
        *seen = midq->n_flagged;
        midq->n_flagged = 0;
        midq->next = 0;
        midq->exited = 0;
      }
      return;
    }
    const int scan = midq->scan[k];
    // K1 writes the dense image and positions only on request or when its list overflows (k_rasterize.h: cc_k1_emit); a scan
    // that comes here without them gets them from its (complete) list: every cell the body looks at beyond a threshold test
    // is an active one, the others only have to stay below the lowest level
    const int4 hd = list.hdr[scan];
    if (hd.z == 0) {
#ifdef CC_EMU
      if (threadIdx.x == 0 && getenv("CC_EMU_TRACE_K2")) fprintf(stderr, "[k2 mid] scan %d: dense image rebuilt from the list (%d entries)\n", scan, hd.x);
#endif
      float *bev = bev_io + (size_t)scan * cfg.n_cell;
      float2 *pix = pix_io + (size_t)scan * cfg.n_cell;
      for (int c = threadIdx.x; c < cfg.n_cell; c += blockDim.x) bev[c] = CC_BEV_EMPTY;
      __threadfence_block();
      __syncthreads();
      const uint16_t *l_rc = list.rc + (size_t)scan * CC_LIST_CAP;
      const float *l_h = list.h + (size_t)scan * CC_LIST_CAP;
      const float2 *l_pix = list.pix + (size_t)scan * CC_LIST_CAP;
      for (int i = threadIdx.x; i < hd.x; i += blockDim.x) {
        const int rc = (int)l_rc[i], c = (rc >> 8) * cfg.n_col + (rc & 255);
        bev[c] = l_h[i];
        pix[c] = l_pix[i];
      }
      __threadfence_block();
      __syncthreads();
      if (threadIdx.x == 0) list.hdr[scan].z = 1;
    }
    cc_k2_body<CC_NC, false>(cfg, bev_io, pix_io, k1_out, scratch_all + scan, nullptr, bigq, scan, desc_out, labels_dbg, nullptr, smem);
  }
}

// The slow path: a few workgroups take the scans the fast launch has queued, one after the other, each with its own block
// of global scratch.  Launched behind every fast launch (the host cannot know whether anything was queued without waiting
// for it); with an empty queue it ends at once.
__global__ void __launch_bounds__(CC_K2_BLOCK, 2)
cc_k_contours_big(cc_dev_cfg cfg, const float *__restrict__ bev_in, const float2 *__restrict__ pix_in,
                  const cc_k1_scan_out *__restrict__ k1_out, cc_k2_big_slot *__restrict__ slots, cc_k2_big_queue *__restrict__ queue,
                  cc_scan_desc_t *__restrict__ desc_out, int16_t *__restrict__ labels_dbg) {
  HIP_DYNAMIC_SHARED(char, smem)
  __shared__ int s_next;
  if (queue->n_flagged == 0) return;  // as in cc_k_contours_mid
  for (;;) {
    __syncthreads();  // the previous scan's LDS is no longer read
    if (threadIdx.x == 0) s_next = atomicAdd(&queue->next, 1);
    __syncthreads();
    const int k = s_next;
    if (k >= queue->n_flagged) {
      // the last workgroup to leave resets the queue for the next call (no memset launch per ingest call)
      if (threadIdx.x == 0 && atomicAdd(&queue->exited, 1) == (int)gridDim.x - 1) {
        queue->n_flagged = 0;
        queue->next = 0;
        queue->exited = 0;
      }
      return;
    }
    cc_k2_body<CC_NC_BIG, true>(cfg, bev_in, pix_in, k1_out, &slots[blockIdx.x].scr, &slots[blockIdx.x].tab, nullptr, queue->scan[k], desc_out,
                                labels_dbg, nullptr, smem);
  }
}

#undef CC_K2_STAMP
