// K4 -- the four-stage gate of CandidateManager::checkCandWithHint (contour_db.h:374-488) for every KNN hit of a chunk of
// queries:  ContourView::checkSim (contour.h:278-329), BCI::checkConstellSim (contour_mng.h:288-388),
// ContourManager::checkConstellCorrespSim + getTFFromConstell (contour_mng.h:1124-1277).
//
// Shape of the work: ~900 hits per query, of which ~60 % pass the anchor test, ~50 % the rotation-window test and ~25 % all
// gates.  The stages therefore run as separate launches over COMPACTED work lists that span the whole chunk, so that the
// lanes of a wave stay on the same stage and the load is balanced over the chip no matter how the survivors are spread
// over the queries:
//   A   one lane per hit slot: anchor checkSim + the popcount bars of checkConstellSim          -> list of checks
//   B1  16 lanes per check: neighbour pairing, exact replay of std::sort, rotation window,
//       pairwise checkSim of the window's pairs                                                  -> list of constellations
//       (+ a large-capacity instance for the rare checks with more than 64 potential pairs)
//   B2  16 lanes per constellation: shaft / orientation filter, 2-D umeyama                      -> pass records
//   C   one lane per pass record: the pose's angle and rotation entries
// Results land in slot-indexed arrays (slot = the reference's candidate iteration order), so the order in which the lists
// were filled never shows.  Both scans of a check are read through their 18 KB "hot" records (cc_hot_desc_t).
#pragma once
#include <cstddef>

#include "cc_dev.h"
#include "cc_group.h"
#include "cc_sort.h"

#define CC_PP_MAX 256      // potential (src,tgt) neighbour pairs per check (large instance of stage B1)
#define CC_PP_SMALL 64     // ... handled by the common, high-occupancy instance
#define CC_CSTL_MAX 64     // pairs kept in a constellation
#define CC_CHK_STRIDE (CC_NQLEV * CC_NPIV * CC_KNN_MAX)  // dense check slots per query: slot * CC_KNN_MAX + j
#define CC_NSCORE 5  // per-check gate scores (hint flow): ovlp_sum, max_one, in_ang_rng, indiv_sim, orie_sim

// A KNN hit names the candidate's anchor (level, seq); the query's anchor is implied by the slot.  In the hint flow
// (cc_db_check_hints) the checks sit in caller order instead, and the query's anchor rides in the high byte of `level`
// (0 = none: derive it from the slot).
#define CC_HIT_LEVEL(h) ((int)((h).level & 0xFF))
#define CC_HIT_SEQ_TGT(h, slot) (((h).level >> 8) ? (int)((h).level >> 8) - 1 : (slot) % CC_NPIV)
#define CC_HIT_PACK_LEVEL(level, seq_tgt) ((int16_t)((level) | (((seq_tgt) + 1) << 8)))

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
  int size_class[3];  // stage A: overlap counts that separate the four size classes of the check list (CC_A_CLASSES)
  int cstl_class[3];  // compaction: constellation lengths that separate the four size classes of stage B2's list (CC_B2_CLASSES)
#ifdef CC_TUNE
  int ablate;  // tuning aid (CC_ABLATE): stage B1 stops after its n-th part (1..5), stage B2 after part n - 10 (11..13)
#endif
};
#ifdef CC_TUNE
#define CC_ABLATE_AT(n) if (P.ablate == (n)) continue
#else
#define CC_ABLATE_AT(n)
#endif

struct cc_chk_item {  // a check that passed stage A
  int q, t;  // t: slot * CC_KNN_MAX + j in the low 16 bits, the two point tables' sizes above (see cc_k_check_a)
  cc_knn_hit_t h;
};
struct cc_cstl_item {  // a check that passed the rotation-window test and kept enough pairs in the individual similarity:
                       // those pairs in cstl_in order (n_in of them; 0 = the check did not get this far)
  int q, t, gidx;
  unsigned char level, seq_src, seq_tgt, n_in;
  int flags;
  unsigned short cs[CC_CSTL_MAX + 2];  // (level << 8) | (seq_src << 4) | seq_tgt
};
// device-side list heads of a chunk: [0] checks, [1] checks left to the large B1 instance, [2] constellations
#define CC_CNT_CHK 0
#define CC_CNT_REDO 1
#define CC_CNT_CSTL 2

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

// Stage A (one lane per check slot of the chunk; a wave's 64 slots are the 64 hit positions of one anchor key): (1/4)
// anchor ContourView::checkSim and the popcount part of (2/4) BCI::checkConstellSim (ovlp_sum / max_one bars).
// grid = ceil(nq * CC_CHK_STRIDE / CC_CHKA_BLOCK), block = CC_CHKA_BLOCK
// (1 024 threads: the list append below costs one returning atomic per workgroup on ONE address, and those are served
// ~11.5 ns apart whatever else the chip does -- profiles/r6/micro/atomic_convoy.hip; with 256-thread workgroups the 4 608
// atomics of a chunk WERE this kernel: 55 of its 69 us)
#define CC_CHKA_BLOCK 1024
__global__ void __launch_bounds__(CC_CHKA_BLOCK)
cc_k_check_a(cc_check_params P, const cc_hot_desc_t *__restrict__ qhot, const cc_hot_desc_t *__restrict__ db_hot,
             int nq, const cc_knn_hit_t *__restrict__ hits, const int *__restrict__ hit_cnt, cc_chk_item *__restrict__ items,
             int *__restrict__ cnt, unsigned char *__restrict__ pass_ok, int *__restrict__ pass_cnt /*[nq][4]*/,
             int *__restrict__ scores /*[nq][CC_CHK_STRIDE][CC_NSCORE] or nullptr: per-check gate scores (hint flow)*/) {
  const int NS = CC_NQLEV * CC_NPIV;
  const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(gt / CC_CHK_STRIDE), t = (int)(gt - (size_t)q * CC_CHK_STRIDE);
  const bool in_range = q < nq;  // whole waves: CC_CHK_STRIDE is a multiple of 64 (no early return: barriers below)
  const int lane = threadIdx.x & 63;
  const int slot = t / CC_KNN_MAX, j = t - slot * CC_KNN_MAX;
  bool anchor_ok = false, keep = false;
  int sc_sum = 0, sc_max = 0, npts = 0;
  cc_knn_hit_t h;
  h.gidx = 0;
  h.level = h.seq = 0;
  h.dist_sq = 0.f;
  if (in_range && j < hit_cnt[q * NS + slot]) {
    h = hits[((size_t)q * NS + slot) * CC_KNN_MAX + j];
    const int seq_tgt = CC_HIT_SEQ_TGT(h, slot), li = CC_HIT_LEVEL(h) - 1;
    const cc_hot_desc_t *src = db_hot + h.gidx, *tgt = qhot + q;
    // the rings are fetched next to the contour rows (both addresses follow from the hit alone): one dependent round trip
    // instead of two, at the price of 64 unused bytes for the ~20 % of the hits that fail the anchor test
    const cc_bci_t *bs = &src->bcis[li][h.seq];
    const cc_bci_t *bt = &tgt->bcis[li][seq_tgt];
    unsigned long long S[4], T[4];
    for (int w = 0; w < 4; w++) {
      S[w] = bs->dist_bin[w];
      T[w] = bt->dist_bin[w];
    }
    npts = (int)bs->n_pts | ((int)bt->n_pts << 8);  // same 64 bytes as the rings: stage B1 sizes its table loads with them
    anchor_ok = cc_check_sim(src->cont[li][h.seq], tgt->cont[li][seq_tgt], P.sim);
    if (anchor_ok) {
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
  if (in_range) pass_ok[gt] = 0;
  if (scores && in_range) {
    int *sc = scores + gt * CC_NSCORE;
    sc[0] = sc_sum;
    sc[1] = sc_max;
    sc[2] = sc[3] = sc[4] = 0;
  }
  // list append: one global atomic per workgroup (a single counter would otherwise see an atomic per wave, and
  // same-address atomics are served one after the other).  Inside the workgroup's stretch of the list the checks are
  // grouped by expected work: stage B1 gives four consecutive checks to the four 16-lane groups of a wave, which advance
  // together, so a wave takes as long as its largest check -- the potential pairs grow with the overlap count, and
  // checks of one size class sit side by side (the order of the list never shows: results are slot-indexed).
  constexpr int NW = CC_CHKA_BLOCK / 64;
  __shared__ int s_bcnt[4][NW], s_base;  // [size class][wave]; after the hand-over: exclusive prefix in (class, wave) order
  const int bk = !keep ? -1 : (sc_sum < P.size_class[0] ? 0 : (sc_sum < P.size_class[1] ? 1 : (sc_sum < P.size_class[2] ? 2 : 3)));
  const unsigned long long ma = __ballot(anchor_ok);
  const unsigned long long mb0 = __ballot(bk == 0), mb1 = __ballot(bk == 1), mb2 = __ballot(bk == 2), mb3 = __ballot(bk == 3);
  const int wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_bcnt[0][wave] = __popcll(mb0);
    s_bcnt[1][wave] = __popcll(mb1);
    s_bcnt[2][wave] = __popcll(mb2);
    s_bcnt[3][wave] = __popcll(mb3);
    if (ma) atomicAdd(&pass_cnt[q * 4 + 1], __popcll(ma));
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // 4 NW <= 64 counts: prefix by one wave, the total to the list head
    static_assert(4 * NW <= 64, "one wave scans the (class, wave) counts");
    const int v = lane < 4 * NW ? (&s_bcnt[0][0])[lane] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    const int tot = __shfl(incl, 63);
    if (lane < 4 * NW) (&s_bcnt[0][0])[lane] = incl - v;
    if (lane == 0) s_base = tot ? atomicAdd(&cnt[CC_CNT_CHK], tot) : 0;
  }
  __syncthreads();
  if (keep) {
    const int pos = s_base + s_bcnt[bk][wave];
    const unsigned long long mine = bk == 0 ? mb0 : (bk == 1 ? mb1 : (bk == 2 ? mb2 : mb3));
    cc_chk_item it;
    it.q = q;
    it.t = t | (npts << 16);  // slot position (11 bits) | points of the src table << 16 | of the tgt table << 24
    it.h = h;
    items[pos + __popcll(mine & ((1ull << lane) - 1ull))] = it;
  }
}

// ---- stage B1 -------------------------------------------------------------------------------------------------------
#define CC_CHKB_GPW (64 / CC_G)  // groups (checks) per wave

template <int PPM>
struct cc_b1_lds {  // per group; the unions hold data of phases that never overlap in time
  unsigned long long pp[PPM];              // potential pairs in generation order: fkey(orie) << 32 | l | s << 8 | t << 16
  union {
    struct {                               // pair generation
      cc_relpt_t sp[CC_BCI_MAXPTS], tp[CC_BCI_MAXPTS];
      unsigned short off[CC_BCI_MAXPTS + 2];  // first potential pair of each tgt point
      unsigned char lo[CC_BCI_MAXPTS], hi[CC_BCI_MAXPTS];
    } g;
    float skey[PPM];                       // sort result -> window search: orie in sorted order
  };
  int hist[PPM > 64 ? 256 : 2];            // sort: orientation bins (count -> start -> end), large instance only
  unsigned char binidx[PPM];               // pair indices grouped by bin
  unsigned char sidx[PPM];                 // pair indices in sorted order
  short seg[3][20];                        // sort: pending quicksort segments (first, last, depth left); serial-sort stack
};
static_assert(CC_PP_MAX <= 256, "pair indices are bytes");
static_assert(CC_BCI_MAXPTS * sizeof(cc_relpt_t) == 60 * 8 && offsetof(cc_bci_t, pts) % 8 == 0, "a point table is 60 aligned 8-byte words");

// potential pairs (contour_mng.h:311-334): for tgt point i (ascending bit_pos) all src points with bit_pos within +-1, in
// src order.
template <int PPM, typename LT>
__device__ __forceinline__ void cc_b1_gen_pairs(LT &L, int ntp, int sl) {
  for (int i = sl; i < ntp; i += CC_G) {
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
      od += 0.0f;  // -0 -> +0: the order-preserving integer key below must not tell them apart (the float compare does not)
      L.pp[o] = ((unsigned long long)cc_fkey(od) << 32) |
                (unsigned long long)((unsigned)(r1.level & 0xFF) | ((unsigned)(r1.seq & 0xFF) << 8) | ((unsigned)(r2.seq & 0xFF) << 16));
    }
  }
}

// orientation bin of the counting sort: monotone non-decreasing in od (f32 add, multiply by a positive constant and
// truncation all preserve order), so bins partition the sorted sequence
__device__ __forceinline__ int cc_b1_bin(float od) {
  int b = (int)((od + 3.14159274f) * 40.7436638f);
  return b < 0 ? 0 : (b > 255 ? 255 : b);
}

// smallest f32 difference d with (double)d + 2 * M_PI > (double)(float)(M_PI / 16): the wrapped window test in f32
#define CC_B1_WRAP_T (-0x1.858eb6p+2f)
#define CC_B1_KEY(w) cc_funkey((unsigned)((w) >> 32))
#define CC_B1_UKEY(w) ((unsigned)((w) >> 32))

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
__device__ __forceinline__ void cc_b1_sort(cc_b1_lds<PPM> &L, int npp, int ntp, int sl) {
  const int G = CC_G;
  unsigned char *lpos = L.binidx, *rasc = L.sidx;  // stopper lists (both arrays are free until step 2)
  bool deep = false;
  cc_group_sync();  // the pairs are in place
  // No "all keys distinct" shortcut: the same pair of contours shows up on several levels whenever an object's walls are
  // vertical (identical cell sets, identical centres), so almost every check has equal keys (measured on the bench world:
  // 95 % of the checks) and the order among them is what the replay below is for.
  if (npp > 16) {
    int lg = 0;
    for (int t = npp; t > 1; t >>= 1) lg++;
    int nseg = 1;
    L.seg[0][0] = 0;
    L.seg[1][0] = (short)npp;
    L.seg[2][0] = (short)(lg * 2);
    cc_group_sync();
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
      // comparisons on the order-preserving integer keys (same outcome as on the floats: no NaN, no -0)
      const unsigned ka = CC_B1_UKEY(L.pp[ia]), kb = CC_B1_UKEY(L.pp[ib]), kc = CC_B1_UKEY(L.pp[ic]);
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
      cc_group_sync();
      if (sl == 0) {
        const unsigned long long t = L.pp[first];
        L.pp[first] = L.pp[sel];
        L.pp[sel] = t;
      }
      cc_group_sync();
      const unsigned piv = CC_B1_UKEY(L.pp[first]);
      int nL = 0, nR = 0;
      for (int r0 = first + 1; r0 < last; r0 += G) {
        const int i = r0 + sl;
        bool ls = false, rs = false;
        if (i < last) {
          const unsigned k = CC_B1_UKEY(L.pp[i]);
          ls = k >= piv;
          rs = k <= piv;
        }
        const unsigned mL = cc_group_ballot(ls), mR = cc_group_ballot(rs);
        if (ls) lpos[nL + __popc(mL & ((1u << sl) - 1u))] = (unsigned char)i;
        if (rs) rasc[nR + __popc(mR & ((1u << sl) - 1u))] = (unsigned char)i;
        nL += __popc(mL);
        nR += __popc(mR);
      }
      cc_group_sync();
      const int nmin = nL < nR ? nL : nR;
      int K = 0;
      for (int k0 = 0; k0 < nmin; k0 += G) {
        const int k = k0 + sl;
        K += __popc(cc_group_ballot(k < nmin && lpos[k] < rasc[nR - 1 - k]));
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
      cc_group_sync();
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
      cc_group_sync();
    }
  }
  if (deep) {
    cc_group_sync();
    cc_b1_gen_pairs<PPM>(L, ntp, sl);
    cc_group_sync();
    if (sl == 0)
      ccsort::std_sort(L.pp, npp,
                       [](const unsigned long long &x, const unsigned long long &y) { return CC_B1_KEY(x) < CC_B1_KEY(y); },
                       (unsigned *)&L.seg[0][0]);  // rare path: generic pointers are fine here
    cc_group_sync();
    for (int k = sl; k < npp; k += G) {
      L.sidx[k] = (unsigned char)k;
      L.skey[k] = CC_B1_KEY(L.pp[k]);
    }
    cc_group_sync();
    return;
  }
  // Stable rank = #(smaller keys) + #(equal keys at earlier positions): one 64-bit unsigned comparison of
  // (order-preserving key << 32 | position) per pair of elements.
  if (npp <= G) {  // one pair per lane: through group broadcasts, no LDS traffic
    const unsigned kme = sl < npp ? CC_B1_UKEY(L.pp[sl]) : 0xFFFFFFFFu;
    const unsigned long long me = ((unsigned long long)kme << 32) | (unsigned)sl;
    int rank = 0;
    for (int j = 0; j < npp; j++) {
      const unsigned long long oj = ((unsigned long long)cc_group_bcast(kme, j) << 32) | (unsigned)j;
      rank += oj < me ? 1 : 0;
    }
    cc_group_sync();  // every lane has read its pp before skey (same storage as the point tables, not as pp) is written
    if (sl < npp) {
      L.sidx[rank] = (unsigned char)sl;
      L.skey[rank] = cc_funkey(kme);
    }
    cc_group_sync();
    return;
  }
  if (npp <= 48 || PPM <= 64) {
    constexpr int NU = (PPM <= 64 ? 64 : 48) / CC_G;
    unsigned kv[NU];
    unsigned long long me[NU];
    int rk[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int p = sl + u * G;
      kv[u] = p < npp ? CC_B1_UKEY(L.pp[p]) : 0xFFFFFFFFu;
      me[u] = ((unsigned long long)kv[u] << 32) | (unsigned)p;
      rk[u] = 0;
    }
    if (npp <= 2 * G) {
      for (int j = 0; j < npp; j++) {
        const unsigned long long oj = ((unsigned long long)CC_B1_UKEY(L.pp[j]) << 32) | (unsigned)j;
        rk[0] += oj < me[0] ? 1 : 0;
        rk[1] += oj < me[1] ? 1 : 0;
      }
    } else {
      for (int j = 0; j < npp; j++) {
        const unsigned long long oj = ((unsigned long long)CC_B1_UKEY(L.pp[j]) << 32) | (unsigned)j;
#pragma unroll
        for (int u = 0; u < NU; u++) rk[u] += oj < me[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int p = sl + u * G;
      if (p < npp) {
        L.sidx[rk[u]] = (unsigned char)p;
        L.skey[rk[u]] = cc_funkey(kv[u]);
      }
    }
    cc_group_sync();
    return;
  }
  if constexpr (PPM > 64) {
    // counting sort on orientation bins + exact rank inside a bin
    for (int i = sl; i < 256; i += G) L.hist[i] = 0;
    cc_group_sync();
    for (int o = sl; o < npp; o += G) atomicAdd(&L.hist[cc_b1_bin(CC_B1_KEY(L.pp[o]))], 1);
    cc_group_sync();
    {
      int loc[256 / CC_G];
      int sum = 0;
      for (int u = 0; u < 256 / CC_G; u++) {
        loc[u] = L.hist[sl * (256 / CC_G) + u];
        sum += loc[u];
      }
      const int incl = cc_group_scan_incl(sum);
      int run = incl - sum;
      for (int u = 0; u < 256 / CC_G; u++) {
        L.hist[sl * (256 / CC_G) + u] = run;  // start of the bin; used as the scatter cursor next
        run += loc[u];
      }
    }
    cc_group_sync();
    for (int o = sl; o < npp; o += G) {
      const int pos = atomicAdd(&L.hist[cc_b1_bin(CC_B1_KEY(L.pp[o]))], 1);
      L.binidx[pos] = (unsigned char)o;
    }
    cc_group_sync();  // hist[b] is now the END of bin b
    for (int p = sl; p < npp; p += G) {
      const int o = L.binidx[p];
      const float f = CC_B1_KEY(L.pp[o]);
      const int bn = cc_b1_bin(f);
      const int start = bn ? L.hist[bn - 1] : 0, end = L.hist[bn];
      int rank = 0;
      for (int p2 = start; p2 < end; p2++) {
        const int o2 = L.binidx[p2];
        const float f2 = CC_B1_KEY(L.pp[o2]);
        rank += (f2 < f || (f2 == f && o2 < o)) ? 1 : 0;
      }
      L.sidx[start + rank] = (unsigned char)o;
      L.skey[start + rank] = f;
    }
    cc_group_sync();
  }
}

// Two instances: <CC_PP_SMALL, false> handles every check with <= 64 potential pairs and lists the others;
// <CC_PP_MAX, true> then runs only those.
// grid = any (grid-stride over the device-side list), block = 64
template <int PPM, bool REDO, int WPE = 4>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
cc_k_check_b1(cc_check_params P, const cc_hot_desc_t *__restrict__ qhot, const cc_hot_desc_t *__restrict__ db_hot,
              const cc_chk_item *__restrict__ items, int *__restrict__ redo_idx, int *__restrict__ cnt, cc_cstl_item *__restrict__ cstl,
              int *__restrict__ pass_cnt, int *__restrict__ scores /*see cc_k_check_a; or nullptr*/) {
  __shared__ cc_b1_lds<PPM> LG[CC_CHKB_GPW];
  const int G = CC_G;
  const int sub = threadIdx.x / CC_G, sl = threadIdx.x % CC_G;
  cc_b1_lds<PPM> &L = LG[sub];
  const int n_items = REDO ? cnt[CC_CNT_REDO] : cnt[CC_CNT_CHK];
  const int stride = gridDim.x * CC_CHKB_GPW;
  // the next check's item (and, in the redo instance, its index) is fetched while the current one is worked on
  int i0 = blockIdx.x * CC_CHKB_GPW + sub;
  cc_chk_item it_nxt;
  it_nxt.q = it_nxt.t = 0;
  it_nxt.h.gidx = 0;
  it_nxt.h.level = it_nxt.h.seq = 0;
  it_nxt.h.dist_sq = 0.f;
  if (i0 < n_items) it_nxt = items[REDO ? redo_idx[i0] : i0];
  for (int i = i0; i < n_items; i += stride) {
    const cc_chk_item it = it_nxt;
    if (i + stride < n_items) it_nxt = items[REDO ? redo_idx[i + stride] : i + stride];
    const int q = it.q, t = it.t & 0xFFFF;
    const int nsp = (it.t >> 16) & 0xFF, ntp = (it.t >> 24) & 0xFF;
    const cc_knn_hit_t h = it.h;
    // the check's constellation record sits at the check's own list position; n_in = 0 until (unless) it passes
    cc_cstl_item *out = cstl + (REDO ? redo_idx[i] : i);
    if (sl == 0) out->n_in = 0;
    const int slot = t / CC_KNN_MAX;
    const int level = CC_HIT_LEVEL(h), seq_src = h.seq, seq_tgt = CC_HIT_SEQ_TGT(h, slot);
    int *sc = scores ? scores + ((size_t)q * CC_CHK_STRIDE + t) * CC_NSCORE : nullptr;
    const cc_bci_t *bs = &db_hot[h.gidx].bcis[level - 1][seq_src];
    const cc_bci_t *bt = &qhot[q].bcis[level - 1][seq_tgt];
    // point tables as 8-byte words: a table is 40 x 12 B = 60 words at an 8-byte aligned offset, four words per lane; only
    // the words that hold points are fetched (the sizes came with the item: ~22 of the 40 entries are in use)
    uint2 ps[4], pt[4];
    const uint2 *gs = (const uint2 *)bs->pts, *gt = (const uint2 *)bt->pts;
    const int nws = (nsp * 12 + 7) >> 3, nwt = (ntp * 12 + 7) >> 3;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = sl + u * G;
      ps[u] = k < nws ? gs[k] : make_uint2(0u, 0u);
      pt[u] = k < nwt ? gt[k] : make_uint2(0u, 0u);
    }
    cc_group_sync();  // the previous check's reads of the shared storage are done
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = sl + u * G;
      if (k < 60) {
        ((uint2 *)L.g.sp)[k] = ps[u];
        ((uint2 *)L.g.tp)[k] = pt[u];
      }
    }
    cc_group_sync();
    CC_ABLATE_AT(1);
    // src points are sorted by bit_pos: the partners of a tgt point are the contiguous range [lo, hi)
    int npp_all = 0;
    for (int r0 = 0; r0 < ntp; r0 += G) {
      const int k = r0 + sl;
      int cnt_k = 0;
      if (k < ntp) {
        const int tb = (int)L.g.tp[k].bit_pos;
        // two lower-bound searches over <= 40 sorted entries with a FIXED trip count and predicated updates: the lanes of a
        // wave (four checks in different states) stay on one instruction stream instead of peeling off through exec masks
        int a = 0, b = nsp;
#pragma unroll
        for (int it = 0; it < 6; it++) {  // #(sb < tb - 1); 2^6 > CC_BCI_MAXPTS
          const int mid = (a + b) >> 1;   // < nsp while a < b; == a == b <= nsp afterwards (clamped for the read)
          const bool go = a < b, up = (int)L.g.sp[mid < CC_BCI_MAXPTS ? mid : CC_BCI_MAXPTS - 1].bit_pos < tb - 1;
          a = (go && up) ? mid + 1 : a;
          b = (go && !up) ? mid : b;
        }
        const int lo = a;
        b = nsp;
#pragma unroll
        for (int it = 0; it < 6; it++) {  // #(sb <= tb + 1)
          const int mid = (a + b) >> 1;
          const bool go = a < b, up = (int)L.g.sp[mid < CC_BCI_MAXPTS ? mid : CC_BCI_MAXPTS - 1].bit_pos <= tb + 1;
          a = (go && up) ? mid + 1 : a;
          b = (go && !up) ? mid : b;
        }
        L.g.lo[k] = (unsigned char)lo;
        L.g.hi[k] = (unsigned char)a;
        cnt_k = a - lo;
      }
      const int incl = cc_group_scan_incl(cnt_k);
      if (k < ntp) L.g.off[k] = (unsigned short)(npp_all + incl - cnt_k);
      npp_all += cc_group_sum_i(cnt_k);
    }
    CC_ABLATE_AT(2);
    int flags = 0;
    int npp = npp_all;
    if (npp > PPM) {
      if (!REDO) {  // left to the large instance
        if (sl == 0) redo_idx[atomicAdd(&cnt[CC_CNT_REDO], 1)] = i;  // rare: no contention to speak of
        continue;
      }
      npp = PPM;
      flags |= 1;
      if (sl == 0) atomicOr((unsigned *)&pass_cnt[q * 4 + 0], (unsigned)CC_QF_CHECK_CAP);  // whether or not the check goes on to pass
    }
    cc_group_sync();
    if (npp == 0) {
      if (sc && sl == 0) sc[2] = 1;  // the window search starts from longest_in_range = 1 (contour_mng.h:345)
      continue;
    }
    cc_b1_gen_pairs<PPM>(L, ntp, sl);
    CC_ABLATE_AT(3);
    cc_b1_sort<PPM>(L, npp, ntp, sl);
    CC_ABLATE_AT(4);
    // circular window of width pi/16 (contour_mng.h:344-357): for each start p1 the furthest p2, then the first start
    // that attains the maximum length (what the two-pointer loop records)
    const float angular_range = (float)(3.14159265358979323846 / 16);
    int bestL = 0, bestP = 0x7fffffff;
    // The reference's test is `(double)(f32 difference) + 2 pi * (p2 / n) > (double)angular_range`.  Without the wrap
    // term that is the f32 comparison itself; with it, the f64 sum is monotone in the difference and first exceeds the
    // range at the f32 value CC_B1_WRAP_T (tests/test_b1_window_threshold.py walks the floats around it), so the search
    // needs no f64 arithmetic.
    for (int p1 = sl; p1 < npp; p1 += G) {
      const float v1 = L.skey[p1];
      int a = p1, b = p1 + npp - 1;  // window [p1, p2], p2 in [p1, p1+npp)
      // largest p2 with valid(p2); valid is monotone in p2.  Fixed trip count (npp <= PPM), predicated updates: see above
#pragma unroll
      for (int it = 0; it < (PPM <= 64 ? 6 : 8); it++) {
        const int mid = (a + b + 1) >> 1;  // in (a, b] while a < b; == a == b afterwards: a valid index either way
        const bool wr = mid >= npp;  // mid < 2 * npp: mid % npp and mid / npp without a division
        const float d = L.skey[mid - (wr ? npp : 0)] - v1;
        const bool go = a < b, out = wr ? d >= CC_B1_WRAP_T : d > angular_range;
        b = (go && out) ? mid - 1 : b;
        a = (go && !out) ? mid : a;
      }
      const int len = a - p1 + 1;
      if (len > bestL || (len == bestL && p1 < bestP)) {
        bestL = len;
        bestP = p1;
      }
    }
    cc_group_best(bestL, bestP);
    int longest = bestL, beg = bestP;
    if (longest <= 1) {  // the loop starts from longest = 1, beg = 0 and only records strictly longer windows
      longest = 1;
      beg = 0;
    }
    if (sc && sl == 0) sc[2] = longest;
    CC_ABLATE_AT(5);
    if (longest < P.lb.i_in_ang_rng) continue;
    // hand the constellation over to stage B2: the window pairs in sorted order, then the anchors (cstl_in order)
    int n_in = longest + 1;
    if (n_in > CC_CSTL_MAX) {
      n_in = CC_CSTL_MAX;
      flags |= 1;
      if (sl == 0) atomicOr((unsigned *)&pass_cnt[q * 4 + 0], (unsigned)CC_QF_CHECK_CAP);
    }
    if (sl == 0) atomicAdd(&pass_cnt[q * 4 + 2], 1);
    // (3/4, first part) the individual similarity of the window pairs and the anchors (contour_mng.h:1138-1160) is
    // decided here, where the pairs are at hand: a check that keeps too few pairs ends without a record (a third of those
    // that reach this point), and stage B2 is handed the pairs that passed, in cstl_in order, instead of reading the
    // window back and gathering both contour rows of every pair only to drop it
    const cc_hot_desc_t *src_d = db_hot + h.gidx, *tgt_d = qhot + q;
    int ncs = 0;
    for (int r0 = 0; r0 < n_in; r0 += G) {
      const int e = r0 + sl;
      unsigned v = 0;
      bool sim = false;
      if (e < n_in) {
        if (e < longest && e < n_in - 1) {
          const int pe = beg + e;  // < 2 * npp
          const unsigned w = (unsigned)L.pp[L.sidx[pe >= npp ? pe - npp : pe]];
          v = ((w & 0xFF) << 8) | (((w >> 8) & 0xF) << 4) | ((w >> 16) & 0xF);
        } else {
          v = (unsigned)((level << 8) | (seq_src << 4) | seq_tgt);
        }
        sim = cc_check_sim(src_d->cont[(v >> 8) - 1][(v >> 4) & 0xF], tgt_d->cont[(v >> 8) - 1][v & 0xF], P.sim);
      }
      const unsigned ms = cc_group_ballot(sim);
      if (sim) out->cs[ncs + __popc(ms & ((1u << sl) - 1u))] = (unsigned short)v;  // <= n_in <= CC_CSTL_MAX entries
      ncs += __popc(ms);
    }
    if (sc && sl == 0) sc[3] = ncs;
    CC_ABLATE_AT(6);
    if (ncs < P.lb.i_indiv_sim) continue;
    if (sl == 0) {
      out->q = q;
      out->t = t;
      out->gidx = h.gidx;
      out->level = (unsigned char)level;
      out->seq_src = (unsigned char)seq_src;
      out->seq_tgt = (unsigned char)seq_tgt;
      out->n_in = (unsigned char)ncs;  // the pairs that passed the individual similarity (>= i_indiv_sim >= 1: never 0 here)
      out->flags = flags;
    }
  }
}

// The constellations that passed, as a dense index list for stage B2 (order irrelevant: results are slot-indexed).
// One global atomic per workgroup.  grid = ceil(n_chk_max / CC_CHKA_BLOCK) (device-side bound check), block = CC_CHKA_BLOCK
__global__ void __launch_bounds__(CC_CHKA_BLOCK)
cc_k_compact_cstl(cc_check_params P, const cc_cstl_item *__restrict__ cstl, int *__restrict__ cnt, int *__restrict__ cstl_idx) {
  constexpr int NW = CC_CHKA_BLOCK / 64;
  __shared__ int s_bcnt[4][NW], s_base;  // [size class][wave]: stage B2's groups advance four to a wave, like stage B1's
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_in = i < cnt[CC_CNT_CHK] ? (int)cstl[i].n_in : 0;
  const bool ok = n_in != 0;
  const int bk = !ok ? -1 : (n_in < P.cstl_class[0] ? 0 : (n_in < P.cstl_class[1] ? 1 : (n_in < P.cstl_class[2] ? 2 : 3)));
  const unsigned long long mb0 = __ballot(bk == 0), mb1 = __ballot(bk == 1), mb2 = __ballot(bk == 2), mb3 = __ballot(bk == 3);
  if (lane == 0) {
    s_bcnt[0][wave] = __popcll(mb0);
    s_bcnt[1][wave] = __popcll(mb1);
    s_bcnt[2][wave] = __popcll(mb2);
    s_bcnt[3][wave] = __popcll(mb3);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int v = lane < 4 * NW ? (&s_bcnt[0][0])[lane] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    const int tot = __shfl(incl, 63);
    if (lane < 4 * NW) (&s_bcnt[0][0])[lane] = incl - v;
    if (lane == 0) s_base = tot ? atomicAdd(&cnt[CC_CNT_CSTL], tot) : 0;
  }
  __syncthreads();
  if (ok) {
    const int pos = s_base + s_bcnt[bk][wave];
    const unsigned long long mine = bk == 0 ? mb0 : (bk == 1 ? mb1 : (bk == 2 ? mb2 : mb3));
    cstl_idx[pos + __popcll(mine & ((1ull << lane) - 1ull))] = i;
  }
}

// ---- stage B2 -------------------------------------------------------------------------------------------------------
struct cc_b2_lds {  // per group
  unsigned long long bitsw[8];             // pair bitmap staging (first member: 8-byte aligned for the 64-bit LDS atomics)
  float spm[CC_CSTL_MAX][2], tpm[CC_CSTL_MAX][2];  // pos_mean of the constellation's contours
  unsigned short cs[CC_CSTL_MAX];          // entries that passed the individual similarity, packed as in cc_cstl_item
  unsigned char keepf[CC_CSTL_MAX];
  float cn[48], nn[48];                    // shaft candidates: length, length after normalisation
  int misc[4];
};

// pair index 0..44 of the shaft scan -> i (the reference's loop order: i = 1.., j = 0..i-1): 4 bits per entry
__device__ __forceinline__ int cc_shaft_i(int pr) {
  // i = 1 for pr 0; 2 for 1-2; 3 for 3-5; 4 for 6-9; 5 for 10-14; 6 for 15-20; 7 for 21-27; 8 for 28-35; 9 for 36-44
  const unsigned long long t0 = 0x6555554444333221ull;  // pr 0..15
  const unsigned long long t1 = 0x8888777777766666ull;  // pr 16..31
  const unsigned long long t2 = 0x0009999999998888ull;  // pr 32..47
  const unsigned long long w = pr < 16 ? t0 : (pr < 32 ? t1 : t2);
  return (int)((w >> ((pr & 15) * 4)) & 0xF);
}

// grid = any (grid-stride over the device-side list), block = 64
__global__ void __launch_bounds__(64)
cc_k_check_b2(cc_check_params P, const cc_hot_desc_t *__restrict__ qhot, const cc_hot_desc_t *__restrict__ db_hot,
              const cc_cstl_item *__restrict__ cstl, const int *__restrict__ cstl_idx, const int *__restrict__ cnt,
              cc_pass_rec *__restrict__ pass, unsigned char *__restrict__ pass_ok, int *__restrict__ pass_cnt,
              int *__restrict__ scores) {
  __shared__ cc_b2_lds LG[CC_CHKB_GPW];
  const int G = CC_G;
  const int sub = threadIdx.x / CC_G, sl = threadIdx.x % CC_G;
  cc_b2_lds &L = LG[sub];
  const int n_items = cnt[CC_CNT_CSTL];
  for (int i = blockIdx.x * CC_CHKB_GPW + sub; i < n_items; i += gridDim.x * CC_CHKB_GPW) {
    const cc_cstl_item *it = cstl + cstl_idx[i];
    const int q = it->q, t = it->t, gidx = it->gidx, n_in = it->n_in;
    int flags = it->flags;
    int *sc = scores ? scores + ((size_t)q * CC_CHK_STRIDE + t) * CC_NSCORE : nullptr;
    const cc_hot_desc_t *src = db_hot + gidx, *tgt = qhot + q;
    cc_group_sync();  // the previous constellation's reads of the group's LDS are done
    // (3/4) the pairs that passed the individual similarity (stage B1's tail), in cstl_in order: their centres into LDS
    // What the orientation test below needs of the same two rows (the major axes, the eccentricity flags) comes along in
    // this gather and waits in registers -- entry e = sl + 16 u is this lane's in both loops -- instead of a second dependent
    // round trip to the rows.
    const int ncs_in = n_in;
    float ax_s[CC_CSTL_MAX / CC_G][2], ax_t[CC_CSTL_MAX / CC_G][2];
    unsigned ecc_both = 0u;  // bit u: both contours of entry sl + 16 u have ecc_feat
#pragma unroll
    for (int u = 0; u < CC_CSTL_MAX / CC_G; u++) {
      const int e = sl + u * G;
      ax_s[u][0] = ax_s[u][1] = ax_t[u][0] = ax_t[u][1] = 0.f;
      if (e < ncs_in) {
        const unsigned v = it->cs[e];
        const int l = (int)(v >> 8), s_ = (int)((v >> 4) & 0xF), t_ = (int)(v & 0xF);
        const cc_contour_t &scv = src->cont[l - 1][s_];
        const cc_contour_t &tcv = tgt->cont[l - 1][t_];
        L.cs[e] = (unsigned short)v;
        L.spm[e][0] = scv.pos_mean[0];
        L.spm[e][1] = scv.pos_mean[1];
        L.tpm[e][0] = tcv.pos_mean[0];
        L.tpm[e][1] = tcv.pos_mean[1];
        ax_s[u][0] = scv.eig_vecs[2];
        ax_s[u][1] = scv.eig_vecs[3];
        ax_t[u][0] = tcv.eig_vecs[2];
        ax_t[u][1] = tcv.eig_vecs[3];
        ecc_both |= (scv.ecc_feat && tcv.ecc_feat) ? (1u << u) : 0u;
      }
    }
    int ncs = ncs_in;
    CC_ABLATE_AT(11);
    cc_group_sync();
    // part 2: the "shaft" (contour_mng.h:1173-1184).  The reference scans the (i, j<i) pairs of the first <=10 entries in
    // order, replacing the running (normalised) src vector whenever the candidate is LONGER THAN THE RUNNING VECTOR'S NORM
    // -- which, once a vector has been taken, is 1 up to rounding.  So the result is the last candidate in scan order that
    // is longer than ~1 pixel: every candidate clearly longer than 1 replaces whatever came before, and only the
    // candidates after the last such one (rare: centres closer than a pixel) need the exact sequential comparison.
    float shx = 0.f, shy = 0.f, thx = 0.f, thy = 0.f;
    {
      const int lim = ncs < 10 ? ncs : 10;
      const int npair = lim * (lim - 1) / 2;
      int sure = -1;   // last pair index of this lane with a length clearly above any normalised length
      bool murky = false;
      for (int pr = sl; pr < npair; pr += G) {
        const int ii = cc_shaft_i(pr), jj = pr - ii * (ii - 1) / 2;
        const float cx = L.spm[ii][0] - L.spm[jj][0], cy = L.spm[ii][1] - L.spm[jj][1];
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
        const float cn = cc_norm2f(cx, cy);
        L.cn[pr] = cn;
        L.nn[pr] = cc_norm2f(ux, uy);
        if (cn > 1.001f) sure = pr;
        if (cn > 0.f && !(cn > 1.001f)) murky = true;
      }
      int sure_all = sure, dummy = 0;
      cc_group_best(sure_all, dummy);  // the largest index
      const unsigned any_murky = cc_group_ballot(murky);
      cc_group_sync();
      int last = sure_all;
      if (any_murky) {  // exact replay of the part of the scan that is not decided by length alone
        float sn = 0.f;
        last = -1;
        for (int k = 0; k < npair; k++) {
          if (L.cn[k] > sn) {
            sn = L.nn[k];
            last = k;
          }
        }
      }
      if (last >= 0) {
        const int ii = cc_shaft_i(last), jj = last - ii * (ii - 1) / 2;
        const float cx = L.spm[ii][0] - L.spm[jj][0], cy = L.spm[ii][1] - L.spm[jj][1];
        float z = cx * cx + cy * cy;
        if (z > 0.f) {
          const float sq = sqrtf(z);
          shx = cx / sq;
          shy = cy / sq;
        } else {
          shx = cx;
          shy = cy;
        }
        const float tx = L.tpm[ii][0] - L.tpm[jj][0], ty = L.tpm[ii][1] - L.tpm[jj][1];
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
    CC_ABLATE_AT(12);
    // orientation test per pair (order-independent), then the order-dependent swap-to-back removal (contour_mng.h:1186-1201)
    unsigned long long rmm = 0ull;
#pragma unroll
    for (int u = 0; u < CC_CSTL_MAX / CC_G; u++) {
      const int r0 = u * G, e = r0 + sl;
      if (r0 < ncs) {  // group-uniform
        bool rm = false;
        if (e < ncs) {
          if ((ecc_both >> u) & 1u) {
            const float pi6 = (float)(3.14159265358979323846 / 6);
            // The reference compares glibc's acosf values with pi / 6; the device library's acosf is within an ulp or two of
            // them (< 1e-6 absolute).  So it decides every pair that is not within 1e-5 of the threshold, and only those that are
            // go through the restated glibc routine (cc_stats.h: two IEEE divisions and a square root -- on every pair it made
            // this kernel 27 % slower on the sparse world).
            const float ds = shx * ax_s[u][0] + shy * ax_s[u][1], dt = thx * ax_t[u][0] + thy * ax_t[u][1];
            float theta_s = acosf(ds), theta_t = acosf(dt);
            float pms = (float)(3.14159265358979323846 - (double)theta_s);
            float da = fabsf(theta_s - theta_t), db = fabsf(pms - theta_t);
            if (fabsf(da - pi6) < 1e-5f || fabsf(db - pi6) < 1e-5f) {
              theta_s = cc_acosf_fdlibm(ds);
              theta_t = cc_acosf_fdlibm(dt);
              pms = (float)(3.14159265358979323846 - (double)theta_s);
              da = fabsf(theta_s - theta_t);
              db = fabsf(pms - theta_t);
            }
            rm = da > pi6 && db > pi6;
          }
          L.keepf[e] = (unsigned char)e;  // position -> original index (identity when nothing is removed)
        }
        rmm |= (unsigned long long)cc_group_ballot(rm) << r0;
      }
    }
    cc_group_sync();
    if (rmm) {
      if (sl == 0) {
        int num_sim = ncs;
        for (int k = 0; k < num_sim;) {
          const int o = L.keepf[k];
          if ((rmm >> o) & 1ull) {
            L.keepf[k] = L.keepf[num_sim - 1];  // std::swap(cstl_out[i], cstl_out[num_sim-1]); the tail is erased afterwards
            num_sim--;
            continue;
          }
          k++;
        }
        L.misc[0] = num_sim;
      }
      cc_group_sync();
      ncs = L.misc[0];
    }
    if (sc && sl == 0) sc[4] = ncs;
    CC_ABLATE_AT(13);
    if (ncs < P.lb.i_orie_sim) continue;
    // (4/4) getTFFromConstell: 2-D umeyama without scaling, closed form.  The sums run over the list in parallel
    // (partial sums per lane, then a fixed butterfly): the same terms as the reference's sequential sums in another
    // association, i.e. equal up to f64 rounding of the sum -- far inside the pose tolerance.
    if (sl < 8) L.bitsw[sl] = 0ull;
    cc_group_sync();
    double smx = 0, smy = 0, dmx = 0, dmy = 0;
    for (int e = sl; e < ncs; e += G) {
      const int o = L.keepf[e];
      const unsigned v = L.cs[o];
      const int bit = ((int)(v >> 8) - 1) * 100 + (int)((v >> 4) & 0xF) * 10 + (int)(v & 0xF);
      atomicOr(&L.bitsw[bit >> 6], 1ull << (bit & 63));
      smx += (double)L.spm[o][0];
      smy += (double)L.spm[o][1];
      dmx += (double)L.tpm[o][0];
      dmy += (double)L.tpm[o][1];
    }
    const double one_over_n = 1.0 / (double)ncs;
    smx = cc_group_sum_d(smx) * one_over_n;
    smy = cc_group_sum_d(smy) * one_over_n;
    dmx = cc_group_sum_d(dmx) * one_over_n;
    dmy = cc_group_sum_d(dmy) * one_over_n;
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    for (int e = sl; e < ncs; e += G) {
      const int o = L.keepf[e];
      const double ax = (double)L.spm[o][0] - smx, ay = (double)L.spm[o][1] - smy;
      const double bx = (double)L.tpm[o][0] - dmx, by = (double)L.tpm[o][1] - dmy;
      s00 += bx * ax;
      s01 += bx * ay;
      s10 += by * ax;
      s11 += by * ay;
    }
    s00 = cc_group_sum_d(s00) * one_over_n;
    s01 = cc_group_sum_d(s01) * one_over_n;
    s10 = cc_group_sum_d(s10) * one_over_n;
    s11 = cc_group_sum_d(s11) * one_over_n;
    double sn2 = s10 - s01, cs_ = s00 + s11;
    double nrm = sqrt(sn2 * sn2 + cs_ * cs_);
    if (nrm < 1e-6) {  // (group-uniform)
      // A constellation whose cross-covariance cancels (every src contour of a pair set paired with every tgt contour: the
      // sums are products of sums of deviations, i.e. zero) leaves rounding noise in sn2 and cs_, and the rotation is THAT
      // noise's angle: it depends on the association of the sums.  Such a case is summed again in the reference's
      // sequential order (contour_mng.h:1252-1277 as the oracle restates it), by every lane for itself, so that the same
      // noise comes out (round 6: fuzz drive 201965 -- angle exactly 0 here, 0.09 rad there, a candidate kept on one side only).
      smx = smy = dmx = dmy = 0;
      for (int e = 0; e < ncs; e++) {
        const int o = L.keepf[e];
        smx += (double)L.spm[o][0];
        smy += (double)L.spm[o][1];
        dmx += (double)L.tpm[o][0];
        dmy += (double)L.tpm[o][1];
      }
      smx = smx * one_over_n;
      smy = smy * one_over_n;
      dmx = dmx * one_over_n;
      dmy = dmy * one_over_n;
      s00 = s01 = s10 = s11 = 0;
      for (int e = 0; e < ncs; e++) {
        const int o = L.keepf[e];
        const double ax = (double)L.spm[o][0] - smx, ay = (double)L.spm[o][1] - smy;
        const double bx = (double)L.tpm[o][0] - dmx, by = (double)L.tpm[o][1] - dmy;
        s00 += bx * ax;
        s01 += bx * ay;
        s10 += by * ax;
        s11 += by * ay;
      }
      s00 *= one_over_n;
      s01 *= one_over_n;
      s10 *= one_over_n;
      s11 *= one_over_n;
      sn2 = s10 - s01;
      cs_ = s00 + s11;
      nrm = sqrt(sn2 * sn2 + cs_ * cs_);
    }
    double r00 = 1, r10 = 0;
    if (nrm > 0) {
      r00 = cs_ / nrm;
      r10 = sn2 / nrm;
    }
    cc_group_sync();
    cc_pass_rec *rec = &pass[(size_t)q * CC_CHK_STRIDE + t];
    if (sl == 0) {
      atomicAdd(&pass_cnt[q * 4 + 3], 1);
      rec->q = q;
      rec->order = t;
      rec->gidx = gidx;
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
    if (sl < 7) rec->bits[sl] = L.bitsw[sl];
  }
}

// Stage C (one lane per constellation): T_pass's angle atan2(R10, R00) and the rotation rebuilt from it, as the reference
// does (getTFFromConstell returns Isometry2d; addProposal and the pose output go through rotate(angle)).
// grid = any (grid-stride), block = 256
__global__ void __launch_bounds__(256)
cc_k_check_c(const cc_cstl_item *__restrict__ cstl, const int *__restrict__ cstl_idx, const int *__restrict__ cnt,
             cc_pass_rec *__restrict__ pass, const unsigned char *__restrict__ pass_ok) {
  const int n = cnt[CC_CNT_CSTL];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const cc_cstl_item *it = cstl + cstl_idx[i];
    const size_t slot = (size_t)it->q * CC_CHK_STRIDE + it->t;
    if (!pass_ok[slot]) continue;
    cc_pass_rec *rec = &pass[slot];
    const double r00 = rec->cs[0], r10 = rec->cs[1];
    const double th = atan2(r10, r00);
    const double c_ = cos(th), s_2 = sin(th);
    rec->tf[2] = th;
    rec->cs[0] = c_;
    rec->cs[1] = s_2;
    rec->cs[2] = atan2(s_2, c_);
  }
}
