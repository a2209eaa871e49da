// Test program (CPU harness): the class mirror's ContourDB works ahead of the driver when the scan source has published the
// next scans (hostcpp/cont2/contour_db.h "read-ahead of the database").  Whatever the driver then does -- the reference's
// loop, the same scan queried again with other thresholds, scans that are never added, a jump in the scan list -- every
// answer must be the one the strictly sequential database (CC_DB_READ_AHEAD=0) gives.
// usage: db_read_ahead_check <poses.txt> <scans.txt> <mode>;  prints one line per query: "<label> <matched id or -1> <correlation>"
//   mode 0: the reference driver's loop (test/batch_bin_test.cpp:131-237)
//   mode 1: every scan queried twice, the second time with tighter thresholds
//   mode 2: every fifth scan is queried but never added
//   mode 3: the list is walked 0..n/2, then again from n/4 (a jump: the scans in between are read a second time and added again)
//   mode >= 100: a RANDOM driver seeded by the mode: per scan a second query with other thresholds (15 %), the scan not added (10 %), added with
//           another seed than the one the driver's count predicts (5 %) or a later time stamp (5 %), a jump back in the list (4 %, at most six); odd modes: all of it at 0.3 of these rates
//   mode 4: TWO evaluators over the same list feed TWO databases in turn (scan i of A, scan i of B, scan i + 1 of A ...): each
//           database must follow its own source's sequence (lines "q" from A's database, "p" from B's)
#include <cstdio>
#include <cstdlib>

#include "eval/evaluator.h"

SequentialTimeProfiler stp;

static void thresholds(CandidateScoreEnsemble &lb, CandidateScoreEnsemble &ub, bool tight) {
  lb.sim_constell.i_ovlp_sum = lb.sim_constell.i_ovlp_max_one = lb.sim_constell.i_in_ang_rng = tight ? 4 : 3;
  lb.sim_pair.i_indiv_sim = tight ? 4 : 3;
  lb.sim_pair.i_orie_sim = 4;
  lb.sim_post.correlation = tight ? 0.5f : 0.3f;
  lb.sim_post.area_perc = 0.03f;
  lb.sim_post.neg_est_dist = -5.01f;
  ub.sim_constell.i_ovlp_sum = ub.sim_constell.i_ovlp_max_one = ub.sim_constell.i_in_ang_rng = 6;
  ub.sim_pair.i_indiv_sim = ub.sim_pair.i_orie_sim = 6;
  ub.sim_post.correlation = 0.75f;
  ub.sim_post.area_perc = 0.15f;
  ub.sim_post.neg_est_dist = -5.0f;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const int mode = atoi(argv[3]);
  ContourManagerConfig cfg;
  cfg.lv_grads_ = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};
  ContourDBConfig dcfg;
  dcfg.q_levels_ = {1, 2, 3};
  dcfg.tb_cfg_.max_elapse_ = 10.0;
  dcfg.tb_cfg_.min_elapse_ = 6.0;
  ContourDB db(dcfg);
  ContLCDEvaluator ev(argv[1], argv[2], 0.5);
  CandidateScoreEnsemble lb, ub, lb2, ub2;
  thresholds(lb, ub, false);
  thresholds(lb2, ub2, true);
  std::vector<std::shared_ptr<const ContourManager>> cands;
  std::vector<double> corr;
  std::vector<Eigen::Isometry2d> tfs;
  int seq = 0, walked = 0, n_total = 0;
  bool jumped = false;
  auto ask = [&](const char *label, const std::shared_ptr<ContourManager> &cm, const CandidateScoreEnsemble &l, const CandidateScoreEnsemble &u) {
    db.queryRangedKNN(cm, l, u, cands, corr, tfs);
    printf("%s%d %d %.6f\n", label, cm->getIntID(), cands.empty() ? -1 : cands[0]->getIntID(), cands.empty() ? 0.0 : corr[0]);
  };
  if (mode == 4) {
    ContourDB db2(dcfg);
    ContLCDEvaluator ev2(argv[1], argv[2], 0.5);
    int s2 = 0;
    while (ev.loadNewScan() && ev2.loadNewScan()) {
      auto cm = ev.getCurrContourManager(cfg);
      ask("q", cm, lb, ub);
      db.addScan(cm, ev.getCurrScanInfo().ts);
      db.pushAndBalance(seq++, ev.getCurrScanInfo().ts);
      auto cm2 = ev2.getCurrContourManager(cfg);
      db2.queryRangedKNN(cm2, lb, ub, cands, corr, tfs);
      printf("p%d %d %.6f\n", cm2->getIntID(), cands.empty() ? -1 : cands[0]->getIntID(), cands.empty() ? 0.0 : corr[0]);
      db2.addScan(cm2, ev2.getCurrScanInfo().ts);
      db2.pushAndBalance(s2++, ev2.getCurrScanInfo().ts);
      n_total++;
    }
    printf("done %d\n", n_total);
    return 0;
  }
  if (mode >= 100) {
    unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)mode + 12345;
    auto rnd = [&]() {  // uniform in [0, 1)
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      return (double)((st >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
    };
    const double k = (mode & 1) ? 0.3 : 1.0;  // odd modes: a driver that deviates rarely (long stretches answered from the queued work)
    double last_ts = -1e9, t_off = 0.0;
    int jumps = 0, addr = 0;
    while (ev.loadNewScan()) {
      const auto info = ev.getCurrScanInfo();
      auto cm = ev.getCurrContourManager(cfg);
      ask("q", cm, lb, ub);
      if (rnd() < 0.15 * k) ask("t", cm, lb2, ub2);
      if (rnd() >= 0.10 * k) {
        if (rnd() < 0.05 * k) t_off += 0.5;
        double ts = info.ts + t_off;
        if (ts <= last_ts) ts = last_ts + 0.01;  // time stamps keep increasing, whatever the walk does
        last_ts = ts;
        const int sd = rnd() < 0.05 * k ? seq + 1000 : seq;
        seq++;
        db.addScan(cm, ts);
        db.pushAndBalance(sd, ts);
      }
      n_total++;
      addr++;
      if (addr > 8 && jumps < 6 && rnd() < 0.04 * k) {
        jumps++;
        addr = addr - 1 - (int)(rnd() * 6.0);
        t_off += 1000.0;
        ev.jumpTo(addr);
      }
    }
    printf("done %d\n", n_total);
    return 0;
  }
  while (ev.loadNewScan()) {
    const auto info = ev.getCurrScanInfo();
    auto cm = ev.getCurrContourManager(cfg);
    ask("q", cm, lb, ub);
    if (mode == 1) ask("t", cm, lb2, ub2);
    const bool skip_add = mode == 2 && walked % 5 == 4;
    if (!skip_add) {
      db.addScan(cm, info.ts + (jumped ? 1000.0 : 0.0));  // time stamps keep increasing after the jump
      db.pushAndBalance(seq++, info.ts + (jumped ? 1000.0 : 0.0));
    }
    walked++;
    n_total++;
    if (mode == 3 && !jumped && walked == 24) {
      jumped = true;
      ev.jumpTo(12);  // mirror-only helper of the evaluator: the next loadNewScan() loads address 12
    }
  }
  printf("done %d\n", n_total);
  return 0;
}
