// K4b -- CandidatePoseData::addProposal and the selection part of tidyUpCandidates (contour_db.h:286-338, 494-546).
#pragma once
#include "cc_dev.h"
#include "k_check.h"

// ------------------------------------------------------------------------------------------------
// K4b: per query, replay the passing checks in order: CandidatePoseData::addProposal (contour_db.h:286-338) and the
// part of tidyUpCandidates before the correlation (contour_db.h:503-546).  The greedy proposal merge is sequential
// only among checks that name the SAME candidate scan, so the passing checks are threaded into one ordered list per
// candidate and every candidate is replayed by its own lane.  candidates_ keeps first-appearance order.
// ------------------------------------------------------------------------------------------------
#define CC_MAXCAND CC_CHK_STRIDE  // every passing check may name a different scan: no cap to overflow
#define CC_MERGE_BLOCK 64   // one wave per query: no cross-wave hand-offs, 46 KB of LDS (three queries per CU)

struct cc_gmm_problem {
  int q;          // index into qdesc (tgt)
  int gidx;       // index into db_desc (src)
  double tf[3];   // T_init = (x, y, theta)
};

struct cc_dprop {  // CandidateAnchorProp (contour_db.h:267-274); constell_ kept as a 400-bit set in key order
  unsigned long long bits[7];
  double c, s, tx, ty;  // T_delta_ = [c -s tx; s c ty]
  double ang;           // atan2(s, c), carried along instead of being re-derived from the matrix at every merge
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
struct alignas(16) cc_merge_lds {
  cc_dcand st[CC_MERGE_BLOCK];             // lane-private candidate state
  alignas(16) int gid[CC_CHK_STRIDE];      // candidate scan of the i-th passing check (read four at a time)
  unsigned short ord[CC_CHK_STRIDE];       // its check slot
  short next[CC_CHK_STRIDE];               // next passing check naming the same scan, -1 = none
  unsigned short firstrec[CC_CHK_STRIDE];  // first passing check of candidate k (candidates in first-appearance order)
  alignas(16) int cg[CC_CHK_STRIDE + 4];   // candidate k's scan (+ four sentinels behind the last one)
  unsigned short clast[CC_CHK_STRIDE];     // candidate k's last check so far
  int base;
  unsigned char want[CC_CHK_STRIDE];       // candidate k goes on to the correlation
  float tperc[CC_HOT_LEVELS][CC_NDIST];    // cont_perc_ of the query's top contours: cell_cnt * 1.0f / layer_cell_cnt
};

static_assert(CC_CHK_STRIDE % CC_MERGE_BLOCK == 0, "merge scan split");
static_assert(CC_MERGE_BLOCK == 64, "the list building below uses wave ballots");

// grid = nq, block = CC_MERGE_BLOCK
__global__ void __launch_bounds__(CC_MERGE_BLOCK)
cc_k_merge(int nq, cc_score_t lb, int n_row, int n_col, const cc_hot_desc_t *__restrict__ qdesc,
           const cc_hot_desc_t *__restrict__ db_desc, const cc_pass_rec *__restrict__ pass, const unsigned char *__restrict__ pass_ok,
           const int *__restrict__ pass_cnt, cc_cand_out *__restrict__ cands_all, cc_qstate *__restrict__ qstate,
           cc_gmm_problem *__restrict__ probs /*[nq][CC_MAXCAND]: problem of candidate k of query q*/,
           int *__restrict__ prob_list /*dense list of the problems that exist*/, int *__restrict__ n_prob,
           long long *__restrict__ phase /*tuning aid (CC_MERGE_PHASES=1): [nq][8] ticks per stage, else nullptr*/) {
  __shared__ cc_merge_lds L;
#define CC_MERGE_STAMP(i)                                                                                   \
  do {                                                                                                      \
    if (phase && threadIdx.x == 0) phase[(size_t)blockIdx.x * 8 + (i)] = (long long)wall_clock64();          \
  } while (0)
  CC_MERGE_STAMP(0);
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (q >= nq) return;
  const unsigned char *okp = pass_ok + (size_t)q * CC_CHK_STRIDE;
  const cc_pass_rec *recs = pass + (size_t)q * CC_CHK_STRIDE;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  if (tid < CC_HOT_LEVELS * CC_NDIST) {  // visible after the barrier of the list building below
    const int l = tid / CC_NDIST, t_ = tid - l * CC_NDIST;
    const cc_hot_desc_t *tq = qdesc + q;
    L.tperc[l][t_] = (float)tq->cont[l][t_].cell_cnt * 1.0f / (float)tq->layer_cell_cnt[l];
  }
  // ---- ordered list of the passing checks (slot order = the reference's iteration order): 64 slots per round, the
  //      flags of all rounds fetched up front (coalesced byte loads), positions from ballots
  int n = 0;
  {
    unsigned char okv[CC_CHK_STRIDE / 64];
#pragma unroll
    for (int u = 0; u < CC_CHK_STRIDE / 64; u++) okv[u] = okp[u * 64 + lane];
#pragma unroll
    for (int u = 0; u < CC_CHK_STRIDE / 64; u++) {
      const bool ok = okv[u] != 0;
      const unsigned long long m = __ballot(ok);
      if (ok) L.ord[n + __popcll(m & lt_mask)] = (unsigned short)(u * 64 + lane);
      n += __popcll(m);
    }
  }
  __syncthreads();
  CC_MERGE_STAMP(1);
  // the candidate scans of the listed checks: gathers from the pass records, four per lane in flight
  for (int i0 = 0; i0 < n; i0 += 4 * CC_MERGE_BLOCK) {
    int g[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * CC_MERGE_BLOCK + lane;
      g[u] = i < n ? recs[L.ord[i]].gidx : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = i0 + u * CC_MERGE_BLOCK + lane;
      if (i < n) {
        L.gid[i] = g[u];
        L.next[i] = -1;
      }
    }
  }
  __syncthreads();
  CC_MERGE_STAMP(2);
  // ---- thread the checks of one scan together; number the candidates in first-appearance order.  64 checks per round:
  //      (1) each lane looks its scan up among the candidates of the earlier rounds (four per LDS read; ~60 candidates per
  //      query, against a backwards scan over up to all the ~270 earlier checks), (2) the round's checks of the same scan
  //      find each other through ballots, one scan per step, (3) scans seen for the first time are numbered in lane order.
  int nc = 0;
  if (lane < 4) L.cg[lane] = -1;  // behind the last candidate: a value no scan has
  __syncthreads();
  for (int b0 = 0; b0 < n; b0 += CC_MERGE_BLOCK) {
    const int i = b0 + lane;
    const bool valid = i < n;
    const int g = valid ? L.gid[i] : 0;
    int k = -1;
    for (int j0 = 0; j0 < nc; j0 += 4) {
      const int4 v = *(const int4 *)&L.cg[j0];
      k = v.x == g ? j0 : (v.y == g ? j0 + 1 : (v.z == g ? j0 + 2 : (v.w == g ? j0 + 3 : k)));
    }
    if (!valid) k = -1;
    const int before = k >= 0 ? (int)L.clast[k] : -1;  // the scan's last check of the earlier rounds (read before this round moves it)
    unsigned long long rem = __ballot(valid);
    int prev_lane = -1, lead_lane = lane;
    bool last_in_round = false;
    while (rem) {
      const int lead = __ffsll(rem) - 1;
      const int g_lead = __builtin_amdgcn_readlane(g, lead);
      const bool mine = valid && g == g_lead;
      const unsigned long long same = __ballot(mine);
      if (mine) {
        const unsigned long long lower = same & lt_mask;
        prev_lane = lower ? 63 - __builtin_clzll(lower) : -1;
        lead_lane = lead;
        last_in_round = (same >> lane) == 1ull;
      }
      rem &= ~same;
    }
    const bool opens = valid && k < 0 && prev_lane < 0;
    const unsigned long long mo = __ballot(opens);
    int kk = k;
    if (opens) {
      kk = nc + __popcll(mo & lt_mask);
      L.cg[kk] = g;
      L.firstrec[kk] = (unsigned short)i;
    }
    const int k_lead = __shfl(kk, lead_lane);  // a scan opened in this round: its later checks take the number from the first
    if (valid && kk < 0) kk = k_lead;
    if (valid) {
      if (prev_lane >= 0)
        L.next[b0 + prev_lane] = (short)i;
      else if (before >= 0)
        L.next[before] = (short)i;
      if (last_in_round) L.clast[kk] = (unsigned short)i;
    }
    nc += __popcll(mo);
    if (lane < 4) L.cg[nc + lane] = -1;
    __syncthreads();  // the next round reads this round's candidates
  }
  __syncthreads();
  if (tid == 0) {
    cc_qstate st;
    st.n_cand = nc;
    st.flags = 0;
    qstate[q] = st;
  }
  CC_MERGE_STAMP(3);
  // ---- one lane per candidate
  cc_dcand *c = &L.st[tid];
  for (int k = tid; k < nc; k += CC_MERGE_BLOCK) {
    int i = L.firstrec[k];
    c->gidx = L.gid[i];
    c->nprops = 0;
    // the replay is a serial chain per candidate (the hottest candidate of a query sets the kernel's latency): the next
    // record's pose is fetched while the current one is worked on
    const cc_pass_rec *rec = &recs[L.ord[i]];
    int np_n = rec->n_pairs;
    double ptx_n = rec->tf[0], pty_n = rec->tf[1], pc_n = rec->cs[0], ps_n = rec->cs[1], ang_n = rec->cs[2];
    for (; i >= 0;) {
      rec = &recs[L.ord[i]];
      const int np = np_n;
      const double ptx = ptx_n, pty = pty_n;
      const double pc = pc_n, ps = ps_n, ang2_cur = ang_n;
      i = L.next[i];
      if (i >= 0) {
        const cc_pass_rec *rn = &recs[L.ord[i]];
        np_n = rn->n_pairs;
        ptx_n = rn->tf[0];
        pty_n = rn->tf[1];
        pc_n = rn->cs[0];
        ps_n = rn->cs[1];
        ang_n = rn->cs[2];
      }
      const int nprops = c->nprops;
      // CandidatePoseData::addProposal: first proposal within 2.0 (pixels) and 0.3 rad
      int hit = -1;
      for (int pi = 0; pi < nprops && hit < 0; pi++) {
        const cc_dprop *p = &c->props[pi];
        const double i00 = pc, i01 = ps, i10 = -ps, i11 = pc;
        const double itx = -(i00 * ptx + i01 * pty), ity = -(i10 * ptx + i11 * pty);
        const double d00 = i00 * p->c + i01 * p->s, d10 = i10 * p->c + i11 * p->s;
        const double dtx = i00 * p->tx + i01 * p->ty + itx, dty = i10 * p->tx + i11 * p->ty + ity;
        // sqrt(n2) < 2.0 && |atan2(d10, d00)| < 0.3, decided without the f64 square root and arc tangent unless a value lies
        // within 1e-9 (relative) of its bar -- then the reference's expressions decide (sqrt and atan2 are monotone, their
        // rounding moves a result by an ulp, eight orders of magnitude inside that margin)
        const double n2 = dtx * dtx + dty * dty;
        bool near_t = n2 < 4.0 * (1.0 - 1e-9);
        if (!near_t && n2 <= 4.0 * (1.0 + 1e-9)) near_t = sqrt(n2) < 2.0;
        if (near_t) {
          const double t03 = 0.30933624960962325;  // tan(0.3)
          const double ay = fabs(d10), lim = d00 * t03;
          bool near_r = d00 > 0.0 && ay < lim * (1.0 - 1e-9);
          if (!near_r && d00 > 0.0 && ay <= lim * (1.0 + 1e-9)) near_r = fabs(atan2(d10, d00)) < 0.3;
          if (near_r) hit = pi;
        }
      }
      if (hit >= 0) {
        cc_dprop *p = &c->props[hit];
        for (int w = 0; w < 7; w++) p->bits[w] |= rec->bits[w];
        p->vote_cnt += np;
        const int w1 = p->vote_cnt, w2 = np;
        const double bx = (p->tx * w1 + ptx * w2) / (w1 + w2), by = (p->ty * w1 + pty * w2) / (w1 + w2);
        // ang1 = atan2(T_delta(1,0), T_delta(0,0)) (contour_db.h:310): the matrix holds (cos, sin) of an angle this lane
        // produced itself, so that angle -- brought back into (-pi, pi] -- is carried along instead of an f64 atan2 per
        // merge; equal to the recomputed value up to the rounding of sin/cos/atan2 (1e-16, as the device's own
        // transcendental functions differ from glibc's anyway; the pose tolerance is 1e-4)
        const double ang1 = p->ang, ang2 = ang2_cur;
        double diff = ang2 - ang1;
        if (diff < 0) diff += 2 * 3.14159265358979323846;
        if (diff > 3.14159265358979323846) diff -= 2 * 3.14159265358979323846;
        const double ang_bl = diff * w2 / (w1 + w2) + ang1;
        double sn, cs;
        sincos(ang_bl, &sn, &cs);
        p->c = cs;
        p->s = sn;
        double aw = ang_bl;
        if (aw > 3.14159265358979323846) aw -= 2 * 3.14159265358979323846;
        if (aw <= -3.14159265358979323846) aw += 2 * 3.14159265358979323846;
        p->ang = aw;
        p->tx = bx;
        p->ty = by;
      } else if (nprops <= 3) {
        cc_dprop *p = &c->props[nprops];
        for (int w = 0; w < 7; w++) p->bits[w] = rec->bits[w];
        p->c = pc;
        p->s = ps;
        p->ang = ang2_cur;
        p->tx = ptx;
        p->ty = pty;
        p->vote_cnt = np;
        p->area_perc = 0.f;
        c->nprops = nprops + 1;
      }
    }
    // tidyUpCandidates before the correlation (contour_db.h:503-546)
    const cc_hot_desc_t *sl = db_desc + c->gidx;
    int idx_sel = 0;
    for (int pi = 0; pi < c->nprops; pi++) {
      float lev_perc[CC_NLEV] = {0, 0, 0, 0, 0, 0};
      for (int w = 0; w < 7; w++) {
        unsigned long long m = c->props[pi].bits[w];
        while (m) {
          const int b = w * 64 + (__ffsll((unsigned long long)m) - 1);
          m &= m - 1;
          const int l = b / 100 + 1, s_ = (b % 100) / 10, t_ = b % 10;
          const float psrc = (float)sl->cont[l - 1][s_].cell_cnt * 1.0f / (float)sl->layer_cell_cnt[l - 1];
          const float ptgt = L.tperc[l - 1][t_];
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
        // the candidate's correlation problem has a fixed place (no shared counter on this path: a single-address
        // atomic per candidate would serialise the chunk); the dense list is built once per query below
        gi = q * CC_MAXCAND + k;
        cc_gmm_problem pb;
        pb.q = q;
        pb.gidx = c->gidx;
        pb.tf[0] = p0->tx;
        pb.tf[1] = p0->ty;
        pb.tf[2] = p0->ang;
        probs[gi] = pb;
      }
    }
    L.want[k] = gi >= 0 ? 1 : 0;
    cc_cand_out o;
    o.gidx = c->gidx;
    o.nprops = c->nprops;
    o.gmm_idx = gi;
    o.pad = 0;
    cands_all[(size_t)q * CC_MAXCAND + k] = o;
  }
  // ---- dense problem list: ordered ranks within the query, one global atomic per query
  __syncthreads();
  CC_MERGE_STAMP(4);
  int n_want = 0;
  for (int b0 = 0; b0 < nc; b0 += CC_MERGE_BLOCK) {
    const int k = b0 + tid;
    const bool w = k < nc && L.want[k];
    const unsigned long long m = __ballot(w);
    if (w) L.ord[n_want + __popcll(m & lt_mask)] = (unsigned short)k;  // ord (the check slots) is dead by now
    n_want += __popcll(m);
  }
  if (tid == 0) L.base = n_want ? atomicAdd(n_prob, n_want) : 0;
  __syncthreads();
  for (int i = tid; i < n_want; i += CC_MERGE_BLOCK) prob_list[L.base + i] = q * CC_MAXCAND + (int)L.ord[i];
  CC_MERGE_STAMP(5);
  if (phase && tid == 0) {
    phase[(size_t)blockIdx.x * 8 + 6] = n;
    phase[(size_t)blockIdx.x * 8 + 7] = nc;
  }
#undef CC_MERGE_STAMP
}

