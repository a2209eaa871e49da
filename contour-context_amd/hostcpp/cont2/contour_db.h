// Host mirror of include/cont2/contour_db.h (ContourDB, configs, CandidateScoreEnsemble) and of the two
// ConstellCorrelation statics the drivers call (include/cont2/correlation.h:241-296), on top of the C-ABI.
#pragma once
#include <algorithm>
#include <deque>
#include <memory>

#include "../tools/bm_util.h"
#include "contour_mng.h"

// The reference's DYNAMIC_THRES=1 build (CMakeLists.txt:13-21; contour_db.h:439-466, 566-574) raises the lower bars of a
// query's checks from candidate to candidate inside CandidateManager -- a sequential dependence between the checks of one
// query.  The device path evaluates a query's checks side by side with CONSTANT bars (the shipped DYNAMIC_THRES=0; the upper
// ensemble is validated -- lb.strictSmaller(ub) -- and otherwise unused, as in that build).  A driver compiled for the
// dynamic variant must not silently get the constant one:
#if defined(DYNAMIC_THRES) && DYNAMIC_THRES
#error "cont2_amd: DYNAMIC_THRES=1 is not supported (checks of a query run in parallel with constant thresholds); build with -DDYNAMIC_THRES=0"
#endif

// As in the reference (contour_db.h:23): the library records its stage timers ("KNN search", "Constell", "L2 opt") into a
// profiler object that the EXECUTABLE defines.
extern SequentialTimeProfiler stp;

// contour_db.h:54-57
struct TreeBucketConfig {
  double max_elapse_ = 25.0;
  double min_elapse_ = 15.0;
};
// contour_db.h:244-250
struct CandidateScoreEnsemble {
  ScoreConstellSim sim_constell;
  ScorePairwiseSim sim_pair;
  ScorePostProc sim_post;
};
// contour_db.h:658-669
struct ContourDBConfig {
  int nnk_ = 50;
  int max_fine_opt_ = 10;
  std::vector<int> q_levels_;
  ContourSimThresConfig cont_sim_cfg_;
  TreeBucketConfig tb_cfg_;
};

class ContourDB {
  const ContourDBConfig cfg_;
  mutable cc_db *db_ = nullptr;  // created lazily by the first call that sees a ContourManager (queryRangedKNN is const)
  int capacity_;
  std::vector<std::shared_ptr<const ContourManager>> all_bevs_;
  std::vector<double> all_ts_;   // addScan's time stamp and pushAndBalance's seed of every added scan: what a rebuild needs
  std::vector<int> all_seed_;

  // ---- read-ahead of the database (mirror-only; the reference's loop is strictly sequential) ----
  // The driver's loop is  query(i) -> add(i) -> query(i+1) -> ...  and every call waits for a chain of ~20 small kernels.
  // When the scan source has published the scans that come next (cc_host::lookahead(): the evaluator mirror's read-ahead)
  // the database works ahead on the device: it appends scan k and queues scan k+1's query at epoch k+1 -- exactly the
  // state the sequential loop queries it in -- for up to SPEC_DEPTH scans, several chains in flight.  The driver's later
  // calls are then VALIDATED against that work instead of launching it: queryRangedKNN(k) hands out the queued answer if it
  // is the same scan with the same thresholds, addScan / pushAndBalance(k) is a no-op if it is the same scan, time stamp
  // and seed.  Any other call sequence stays correct: a different query is answered at the official epoch (an epoch hides
  // the scans appended after it), a different add REBUILDS the device database from the scans the driver really added.
  struct Spec {
    cc_scan *scan = nullptr;
    double ts = 0;
    int seed = 0;
    cc_score_t lb, ub;
    std::shared_ptr<std::vector<cc_query_result_t>> block;  // the answers of the batch the scan was queued with (epoch = position of the scan)
    int idx = 0;                                             // ... this scan's among them; filled by cc_db_query_collect / _wait
    bool collected = false;
    cc_query_result_t *result() const { return block->data() + idx; }
  };
  static constexpr int SPEC_LOW = 2;  // fewer answers than this queued ahead of the driver: a step goes out with whatever has been published
  mutable std::deque<Spec> spec_;   // scans appended ahead of the driver, oldest first (spec_[j] sits at DB index n_official + j)
  mutable bool have_thres_ = false;
  mutable cc_score_t last_lb_, last_ub_;
  mutable long n_spec_hit_ = 0, n_spec_miss_ = 0, n_rebuild_ = 0;
  mutable double t_ra_[2] = {0, 0};  // host seconds in the two calls of a read-ahead step (CC_EVAL_TIMERS prints them)
  mutable long n_ra_ = 0, n_ra_scans_ = 0;
  mutable int ra_pause_ = 0, ra_backoff_ = 16;  // driver steps the read-ahead still sits out after a rebuild | the next pause (see rebuild())
  mutable const void *source_ = nullptr;  // the scan source of the driver's last scan (ContourManager::scanSource): the one sequence this database predicts
  mutable bool need_rebuild_ = false;  // the device database holds scans the driver has not added (and will not): rebuilt at the next call
  int hub_token_ = -1;
  static int specDepth() {
    static const int d = [] {
      // Scans worked ahead (CC_DB_READ_AHEAD=n, 0: off).  Round 5, first version (one scan per step, one chain of launches per
      // scan): the answers were there when the driver asked and the loop got SLOWER (2 500 -> 1 950 scans/s: every step needed
      // the newest published scan, whose single-scan ingest became the wait).  Batched (a step appends and queries
      // specBatch() scans with ONE chain each; the evaluator ingests its files eight at a time, cc_scan_ingest_batch):
      // 3 080 -> 8 200-8 900 scans/s on MI355X (profiles/r5/dropin_batched_read_ahead.txt).  On by default: 32 scans deep.
      const char *e = getenv("CC_DB_READ_AHEAD");
      return e ? std::max(0, atoi(e)) : 32;
    }();
    return d;
  }
  // scans per read-ahead step: as many of the published scans as have ARRIVED, at most specBatch() (CC_DB_READ_AHEAD_BATCH, default
  // CC_SCAN_BATCH_MAX) and -- unless the queue is about to run dry -- at least specMin() (half of that, at most 8): the step follows
  // the source's own batches (the evaluator publishes eight or sixteen scans at a time)
  static int specBatch() {
    static const int b = [] {
      const char *e = getenv("CC_DB_READ_AHEAD_BATCH");
      const int v = e ? atoi(e) : CC_SCAN_BATCH_MAX;
      return std::max(1, std::min({v, CC_SCAN_BATCH_MAX, std::max(1, specDepth())}));
    }();
    return b;
  }
  static int specMin() { return std::max(1, std::min(8, specBatch() / 2 + specBatch() % 2)); }
  static bool same(const cc_score_t &a, const cc_score_t &b) { return memcmp(&a, &b, sizeof(cc_score_t)) == 0; }
  static void die_cc() {
    fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
    abort();
  }
  void collectSpec() const {  // every queued answer is on the host afterwards
    bool pending = false;
    for (auto &sp : spec_) pending = pending || !sp.collected;
    if (!pending) return;
    if (cc_db_query_wait(db_) != CC_OK) die_cc();
    for (auto &sp : spec_) sp.collected = true;
  }
  // the device database back to the scans the driver has added (all_bevs_): after the driver left the predicted sequence
  // A rebuild costs one append per scan the driver has added so far (batched, sixteen handles per call): a driver that keeps
  // leaving the predicted sequence (every fifth scan never added, two databases fed alternately) would pay that again and
  // again.  So every rebuild PAUSES the read-ahead for ra_backoff_ of the driver's steps and doubles that number: a driver that
  // deviates at a steady rate meets O(log n) rebuilds, one that deviated once is back to full speed sixteen scans later.
  void rebuild() const {
    n_rebuild_++;
    need_rebuild_ = false;
    ra_pause_ = ra_backoff_;
    ra_backoff_ = std::min(ra_backoff_ * 2, 1 << 24);
    if (db_) {
      cc_db_query_wait(db_);
      cc_db_destroy(db_);
      db_ = nullptr;
    }
    spec_.clear();
    if (all_bevs_.empty()) return;
    ensure(*all_bevs_[0]);
    for (size_t i = 0; i < all_bevs_.size();) {
      cc_scan *hs[CC_SCAN_BATCH_MAX];
      int n = 0;
      while (i + n < all_bevs_.size() && n < CC_SCAN_BATCH_MAX) {
        cc_scan *h = all_bevs_[i + n]->scanHandle();
        if (!h || !cc_scan_on_device(h)) break;
        hs[n++] = h;
      }
      int rc;
      if (n > 0) {
        rc = cc_db_add_scan_batch(db_, hs, n, &all_ts_[i], &all_seed_[i]);
        i += (size_t)n;
      } else {  // a scan whose descriptor was moved to the host goes by that copy
        rc = cc_db_add_scan_host(db_, &all_bevs_[i]->desc(), all_ts_[i], all_seed_[i]);
        i++;
      }
      if (rc != CC_OK) die_cc();
    }
  }
  void dropReadAhead() {  // the scan source is about to release the scans spec_ refers to: nothing of them may stay in flight
    if (spec_.empty()) return;
    if (db_ && cc_db_query_wait(db_) != CC_OK) die_cc();
    spec_.clear();
    need_rebuild_ = true;  // their records are still in the device database: rebuilt when (if) the database is used again
  }
  // append + queue for the scans the source has published, behind what is there already: up to specBatch() scans per step
  // (ONE append and ONE chain of query launches for all of them: the loop is bound by the number of launches, host and
  // device side, not by their work).  A step waits until a whole batch has been published unless the queued work is
  // about to run out.
  void readAhead() const {
    if (specDepth() <= 0 || !have_thres_ || !db_) return;
    if (ra_pause_ > 0) {  // called once per pushAndBalance: the driver's steps
      ra_pause_--;
      return;
    }
    if (!source_) return;  // the last scan came from a source that publishes nothing
    const auto up = cc_host::lookahead().snapshot(source_);
    // the published scans that are not in spec_ yet must continue it: spec_ = a prefix of (scan in the driver's hands?, upcoming...)
    size_t pos = 0;
    if (!spec_.empty()) {  // what follows spec_.back() among the published scans is new
      bool found = false;
      for (size_t j = 0; j < up.size(); j++)
        if (up[j].scan == spec_.back().scan) {
          pos = j + 1;
          found = true;
        }
      if (!found) return;
    }
    int seed_next = (all_seed_.empty() ? 0 : all_seed_.back() + 1) + (int)spec_.size();  // the driver counts its scans (batch_bin_test.cpp:237)
    for (;;) {
      const int room = specDepth() - (int)spec_.size();
      int n = (int)std::min<size_t>(up.size() - pos, (size_t)std::max(0, std::min(room, specBatch())));
      const int epoch0 = cc_db_size(db_);
      n = std::min(n, capacity_ - 1 - epoch0);
      for (int j = 0; j < n; j++)
        if (!cc_scan_on_device(up[pos + j].scan)) n = j;
      if (n <= 0) return;
      if ((int)spec_.size() > SPEC_LOW) {  // the queue is not about to run dry: only scans that have ARRIVED, and enough of them
        if (n < specMin()) return;         // (a published scan may still be on its way through K1 / K2: the append would wait for it)
        int n_ready = 0;
        while (n_ready < n && cc_scan_ready(up[pos + n_ready].scan)) n_ready++;
        n = n_ready;
        if (n < specMin()) return;
      }
      cc_scan *scans[CC_SCAN_BATCH_MAX];
      double ts[CC_SCAN_BATCH_MAX];
      int32_t seed[CC_SCAN_BATCH_MAX], epoch[CC_SCAN_BATCH_MAX];
      auto block = std::make_shared<std::vector<cc_query_result_t>>((size_t)n);
      for (int j = 0; j < n; j++) {
        scans[j] = up[pos + j].scan;
        ts[j] = up[pos + j].ts;
        seed[j] = seed_next++;
        epoch[j] = epoch0 + j;  // scan j's query sees the database as it is after the scans before it (an epoch hides the later ones)
      }
      TicToc t0;
      if (cc_db_add_scan_batch(db_, scans, n, ts, seed) != CC_OK) die_cc();
      t_ra_[0] += t0.toc();
      TicToc t1;
      if (cc_db_query_scan_batch_submit(db_, scans, n, epoch, &last_lb_, &last_ub_, block->data()) != CC_OK) die_cc();
      t_ra_[1] += t1.toc();
      n_ra_++;
      n_ra_scans_ += n;
      for (int j = 0; j < n; j++) {
        Spec sp;
        sp.scan = scans[j];
        sp.ts = ts[j];
        sp.seed = seed[j];
        sp.lb = last_lb_;
        sp.ub = last_ub_;
        sp.block = block;
        sp.idx = j;
        spec_.push_back(std::move(sp));
      }
      pos += (size_t)n;
    }
  }
  // the answer of spec_[i] on the host (its whole batch's chain is waited for; batches queued behind it stay in flight)
  void collectOne(size_t i) const {
    Spec &sp = spec_[i];
    if (sp.collected) return;
    if (cc_db_query_collect(db_, sp.result(), 1) != CC_OK) die_cc();
    for (auto &o : spec_)
      if (o.block == sp.block) o.collected = true;
  }

  static cc_score_t to_c(const CandidateScoreEnsemble &e) {
    cc_score_t s;
    s.i_ovlp_sum = e.sim_constell.i_ovlp_sum;
    s.i_ovlp_max_one = e.sim_constell.i_ovlp_max_one;
    s.i_in_ang_rng = e.sim_constell.i_in_ang_rng;
    s.i_indiv_sim = e.sim_pair.i_indiv_sim;
    s.i_orie_sim = e.sim_pair.i_orie_sim;
    s.correlation = e.sim_post.correlation;
    s.area_perc = e.sim_post.area_perc;
    s.neg_est_dist = e.sim_post.neg_est_dist;
    return s;
  }
  void ensure(const ContourManager &cm) const {
    if (db_) return;
    cc_db_cfg_t d;
    cc_default_db_cfg(&d);
    d.nnk = cfg_.nnk_;
    d.max_fine_opt = cfg_.max_fine_opt_;
    d.n_q_levels = (int)cfg_.q_levels_.size();
    for (int i = 0; i < d.n_q_levels && i < CC_NQLEV; i++) d.q_levels[i] = cfg_.q_levels_[i];
    d.cont_sim.ta_cell_cnt = cfg_.cont_sim_cfg_.ta_cell_cnt;
    d.cont_sim.tp_cell_cnt = cfg_.cont_sim_cfg_.tp_cell_cnt;
    d.cont_sim.tp_eigval = cfg_.cont_sim_cfg_.tp_eigval;
    d.cont_sim.ta_h_bar = cfg_.cont_sim_cfg_.ta_h_bar;
    d.cont_sim.ta_rcom = cfg_.cont_sim_cfg_.ta_rcom;
    d.cont_sim.tp_rcom = cfg_.cont_sim_cfg_.tp_rcom;
    d.max_elapse = cfg_.tb_cfg_.max_elapse_;
    d.min_elapse = cfg_.tb_cfg_.min_elapse_;
    if (cc_db_create(cc_host::context(cm.ccfg()), &d, capacity_, &db_) != CC_OK) {
      fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
      abort();
    }
    // per-stage DEVICE times of every query under the reference's stage names (six events per query: they cost a few
    // per cent, so only on request); the wall time of the call is recorded either way
    if (getenv("CC_STP_DEVICE_TIMERS")) cc_db_profile_enable(db_, 1);
    if (specDepth() > 0) cc_db_set_lanes(db_, 4);  // one chain per queued query
  }

 public:
  explicit ContourDB(const ContourDBConfig &config, int capacity_scans = 65536) : cfg_(config), capacity_(capacity_scans) {
    CC_CHECK(!cfg_.q_levels_.empty());
    cc_host::runtime_warm();
    if (specDepth() > 0) hub_token_ = cc_host::lookahead().subscribe([this](const void *src) {
      if (src == source_) dropReadAhead();  // (another source's scans are not among the ones this database has worked ahead on)
    });
    if (hub_token_ >= 0) cc_host::lookahead().subscribeRelease(hub_token_, [this](cc_scan *h) {
      for (const Spec &sp : spec_)
        if (sp.scan == h) {  // a scan this database has worked ahead on dies before the driver added it: its handle's address may come back
          dropReadAhead();
          break;
        }
    });
  }
  ContourDB(const ContourDB &) = delete;
  ContourDB &operator=(const ContourDB &) = delete;
  ~ContourDB() {
    if (hub_token_ >= 0) cc_host::lookahead().unsubscribe(hub_token_);
    if (getenv("CC_EVAL_TIMERS"))
    {
      if (n_ra_ > 0)
        fprintf(stderr, "[ContourDB read-ahead, mean host us per step over %ld steps of %.1f scans] append %.1f  query submit %.1f\n", n_ra_,
                (double)n_ra_scans_ / n_ra_, 1e6 * t_ra_[0] / n_ra_, 1e6 * t_ra_[1] / n_ra_);
      fprintf(stderr, "[ContourDB read-ahead] answers handed out from queued queries %ld, launched on the spot %ld, rebuilds %ld\n", n_spec_hit_, n_spec_miss_, n_rebuild_);
    }
    cc_db_destroy(db_);
  }

  // contour_db.h:698-703
  void queryRangedKNN(const std::shared_ptr<const ContourManager> &q_ptr, const CandidateScoreEnsemble &thres_lb,
                      const CandidateScoreEnsemble &thres_ub, std::vector<std::shared_ptr<const ContourManager>> &cand_ptrs,
                      std::vector<double> &cand_corr, std::vector<Eigen::Isometry2d> &cand_tf) const {
    cand_ptrs.clear();
    cand_corr.clear();
    cand_tf.clear();
    ensure(*q_ptr);
    const cc_score_t lb = to_c(thres_lb), ub = to_c(thres_ub);
    cc_query_result_t r;
    TicToc wall;
    if (need_rebuild_) rebuild();
    last_lb_ = lb;
    last_ub_ = ub;
    have_thres_ = true;
    cc_scan *qh = q_ptr->scanHandle();
    if (!spec_.empty() && qh && spec_.front().scan == qh && same(spec_.front().lb, lb) && same(spec_.front().ub, ub)) {
      // the answer was queued when the scan was published (at the epoch the database is officially in now); only ITS chain is
      // waited for, the queries queued behind it stay in flight
      collectOne(0);
      r = *spec_.front().result();
      n_spec_hit_++;
    } else {
      n_spec_miss_++;
      const int32_t epoch = (int32_t)all_bevs_.size();  // the scans appended ahead of the driver are hidden by the epoch
      int rc;
      if (qh && cc_scan_on_device(qh)) {
        rc = cc_db_query_scan_submit(db_, qh, epoch, &lb, &ub, &r);
        const int r2 = cc_db_query_wait(db_);
        for (auto &sp : spec_) sp.collected = true;
        if (rc == CC_OK) rc = r2;
      } else {  // a scan that was offloaded goes by its host copy
        collectSpec();
        rc = cc_db_query_batch_host(db_, &q_ptr->desc(), 1, &epoch, &lb, &ub, &r);
      }
      if (rc != CC_OK) die_cc();  // CHECK(sim_lb.strictSmaller(sim_ub)) etc.; also CC_ECAPACITY (the reference has no capacities)
    }
    stp.addSample("queryRangedKNN (wall)", wall.toc());
    {  // the reference's stage names (contour_db.h:755,772,787), with the device times of this query's kernels:
       // retrieval | checks + proposal merge | correlation + selection
      double ms[5];
      int nq = 0;
      if (getenv("CC_STP_DEVICE_TIMERS") && cc_db_profile_read(db_, ms, &nq) == CC_OK && nq > 0) {
        stp.addSample("KNN search", ms[0] * 1e-3);
        stp.addSample("Constell", (ms[1] + ms[2]) * 1e-3);
        stp.addSample("L2 opt", (ms[3] + ms[4]) * 1e-3);
      }
    }
    if (r.n_res > 0) {
      cand_ptrs.push_back(all_bevs_[r.cand_gidx]);
      cand_corr.push_back(r.correlation);
      Eigen::Isometry2d T;
      T.rotate(r.tf[2]);
      T.pretranslate(r.tf[0], r.tf[1]);
      cand_tf.push_back(T);
    }
  }
  // contour_db.h:814 and :827
  void addScan(const std::shared_ptr<ContourManager> &added, double curr_timestamp) {
    ensure(*added);
    pending_ = added;
    pending_ts_ = curr_timestamp;
  }
  void pushAndBalance(int seed, double curr_timestamp) {
    // the C-ABI couples addScan + pushAndBalance (they are always called back to back, batch_bin_test.cpp:234-237)
    CC_CHECK(pending_);
    (void)curr_timestamp;
    if (need_rebuild_) rebuild();
    cc_scan *h = pending_->scanHandle();
    if (!spec_.empty() && h && spec_.front().scan == h && spec_.front().ts == pending_ts_ && spec_.front().seed == seed) {
      // appended ahead of time, with exactly these arguments.  (Its own query has been handed out, or is no longer wanted:
      // its answer buffer must outlive the chain that writes it.)
      collectOne(0);
      spec_.pop_front();
    } else {
      if (!spec_.empty()) rebuild();  // the driver left the predicted sequence: back to the scans it really added
      ensure(*pending_);
      const int rc = h && cc_scan_on_device(h) ? cc_db_add_scan(db_, h, pending_ts_, seed) : cc_db_add_scan_host(db_, &pending_->desc(), pending_ts_, seed);
      if (rc != CC_OK) die_cc();
    }
    all_ts_.push_back(pending_ts_);
    all_seed_.push_back(seed);
    // The DB keeps its own compact records of the scan.  The full descriptor (169 KB) stays in its device slot for now: its
    // host copy is fetched when a getter first asks for it, and moving every scan to the host right here cost the loop a
    // 169 KB copy + a stream synchronisation per scan.  Only the most recent residentScans() descriptors stay (1.4 GB at
    // the default 8 192 -- a KITTI sequence fits); beyond that the oldest one moves to the host and its slot is reused.
    if (h) on_device_.push_back(pending_);
    while (on_device_.size() > residentScans()) {
      cc_scan *old = on_device_.front()->scanHandle();
      if (old && cc_scan_offload(old) != CC_OK) {
        fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
        abort();
      }
      on_device_.pop_front();
    }
    all_bevs_.push_back(pending_);
    source_ = pending_->scanSource();
    pending_.reset();
    readAhead();
  }
  // how many added scans keep their full descriptor on the device (env CC_SCANS_ON_DEVICE; 0: none, as before round 4)
  static size_t residentScans() {
    static const size_t n = [] {
      const char *e = getenv("CC_SCANS_ON_DEVICE");
      return e ? (size_t)std::max(0L, atol(e)) : (size_t)8192;
    }();
    return n;
  }

 private:
  std::shared_ptr<ContourManager> pending_;
  double pending_ts_ = 0;
  std::deque<std::shared_ptr<ContourManager>> on_device_;  // added scans whose descriptor still sits in a device slot, oldest first
};

// contour_db.h:264-656, the hint-driven use of CandidateManager (the single-pair flow of test/kitti_read_bin_test.cpp:226-291):
//   CandidateManager m(cm_query, lb, ub);  m.checkCandWithHint(cm_cand, {level, seq_src, seq_tgt}, cont_sim); ...
//   m.tidyUpCandidates();  m.fineOptimize(max_fine_opt, cands, corr, tfs);
// All three stages run on the device (cc_db_check_hints): checkCandWithHint evaluates its hint at once for the returned
// scores and records it; fineOptimize replays the recorded hints in call order through checks, proposal merge, tidy-up
// and refinement.  Candidate scans live in a process-wide device store (one per cont_sim setting).  Hint levels: 1..4.
class CandidateManager {
  // Candidate scans live in a device store shared by the managers of one (grid, cont_sim) setting.  It is a cache: when it
  // is full it is emptied and refilled with what the manager at hand needs, so no number of distinct candidates exhausts it.
  struct Store {
    cc_db *db = nullptr;
    cc_db_cfg_t cfg;
    cc_ctx *ctx = nullptr;
    int cap = 1024;
    std::map<const ContourManager *, int> pos;
    std::vector<std::shared_ptr<const ContourManager>> keep;  // position -> scan (kept alive while it is in the store)
    void reset() {
      cc_db_destroy(db);
      db = nullptr;
      pos.clear();
      keep.clear();
      if (cc_db_create(ctx, &cfg, cap, &db) != CC_OK) die();
    }
    bool has(const ContourManager *p) const { return pos.count(p) != 0; }
    int add(const std::shared_ptr<const ContourManager> &cm) {
      const int g = cc_db_size(db);
      if (cc_db_add_scan_host(db, &cm->desc(), (double)g, g) != CC_OK) die();
      pos[cm.get()] = g;
      keep.push_back(cm);
      return g;
    }
  };
  static Store &store(const ContourManager &cm, const ContourSimThresConfig &cs) {
    static std::map<std::string, Store> pool;
    std::string key((const char *)&cm.ccfg(), sizeof(cc_manager_cfg_t));
    key.append((const char *)&cs, sizeof(cs));
    Store &st = pool[key];
    if (!st.db) {
      cc_default_db_cfg(&st.cfg);
      st.cfg.cont_sim.ta_cell_cnt = cs.ta_cell_cnt;
      st.cfg.cont_sim.tp_cell_cnt = cs.tp_cell_cnt;
      st.cfg.cont_sim.tp_eigval = cs.tp_eigval;
      st.cfg.cont_sim.ta_h_bar = cs.ta_h_bar;
      st.cfg.cont_sim.ta_rcom = cs.ta_rcom;
      st.cfg.cont_sim.tp_rcom = cs.tp_rcom;
      st.ctx = cc_host::context(cm.ccfg());
      if (const char *e = getenv("CC_CAND_STORE_CAP")) st.cap = atoi(e) > 0 ? atoi(e) : st.cap;
      if (cc_db_create(st.ctx, &st.cfg, st.cap, &st.db) != CC_OK) die();
    }
    return st;
  }
  // every candidate of THIS manager present in the store (positions may change when the store is recycled)
  void ensureMine() {
    Store &st = *st_;
    bool all = true;
    for (const auto &c : my_cands_) all = all && st.has(c.get());
    if (all) return;
    int missing = 0;
    for (const auto &c : my_cands_) missing += st.has(c.get()) ? 0 : 1;
    CC_CHECK((int)my_cands_.size() <= st.cap);  // one manager's candidates must fit (the reference has no bound; raise CC_CAND_STORE_CAP)
    if (cc_db_size(st.db) + missing > st.cap) st.reset();
    for (const auto &c : my_cands_)
      if (!st.has(c.get())) st.add(c);
  }
  static void die() {
    fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
    abort();
  }
  static cc_score_t to_c(const CandidateScoreEnsemble &e) {
    cc_score_t s;
    s.i_ovlp_sum = e.sim_constell.i_ovlp_sum;
    s.i_ovlp_max_one = e.sim_constell.i_ovlp_max_one;
    s.i_in_ang_rng = e.sim_constell.i_in_ang_rng;
    s.i_indiv_sim = e.sim_pair.i_indiv_sim;
    s.i_orie_sim = e.sim_pair.i_orie_sim;
    s.correlation = e.sim_post.correlation;
    s.area_perc = e.sim_post.area_perc;
    s.neg_est_dist = e.sim_post.neg_est_dist;
    return s;
  }

  std::shared_ptr<const ContourManager> cm_tgt_;
  const CandidateScoreEnsemble sim_ub_;
  CandidateScoreEnsemble sim_var_;
  Store *st_ = nullptr;
  std::vector<cc_hint_t> hints_;                                // cand_gidx = index into my_cands_ until the call is made
  std::vector<std::shared_ptr<const ContourManager>> my_cands_;  // this manager's candidate scans, first-appearance order
  std::map<int, int> id2local_;                                 // the reference keys candidates_ by getIntID() (contour_db.h:468-476)
  int flow_valve = 0;

 public:
  int cand_aft_check1 = 0, cand_aft_check2 = 0, cand_aft_check3 = 0;

  CandidateManager(std::shared_ptr<const ContourManager> cm_q, const CandidateScoreEnsemble sim_lb, const CandidateScoreEnsemble sim_ub)
      : cm_tgt_(std::move(cm_q)), sim_ub_(sim_ub), sim_var_(sim_lb) {
    CC_CHECK(sim_lb.sim_constell.strictSmaller(sim_ub.sim_constell));
    CC_CHECK(sim_lb.sim_pair.strictSmaller(sim_ub.sim_pair));
    CC_CHECK(sim_lb.sim_post.strictSmaller(sim_ub.sim_post));
  }

  // contour_db.h:374-488
  CandidateScoreEnsemble checkCandWithHint(const std::shared_ptr<const ContourManager> &cm_cand, const ConstellationPair &anchor_pair,
                                           const ContourSimThresConfig &cont_sim = ContourSimThresConfig()) {
    CC_CHECK(flow_valve == 0);
    Store &st = store(*cm_tgt_, cont_sim);
    CC_CHECK(st_ == nullptr || st_ == &st);  // one cont_sim setting per manager
    st_ = &st;
    auto it = id2local_.find(cm_cand->getIntID());
    if (it == id2local_.end()) {
      it = id2local_.insert({cm_cand->getIntID(), (int)my_cands_.size()}).first;
      my_cands_.push_back(cm_cand);
    }
    ensureMine();
    cc_hint_t h;
    h.cand_gidx = st.pos[my_cands_[it->second].get()];
    h.level = anchor_pair.level;
    h.seq_src = anchor_pair.seq_src;
    h.seq_tgt = anchor_pair.seq_tgt;
    h.pad = 0;
    const cc_score_t lb = to_c(sim_var_), ub = to_c(sim_ub_);
    cc_query_result_t r;
    cc_hint_score_t sc;
    if (cc_db_check_hints_host(st.db, &cm_tgt_->desc(), &h, 1, &lb, &ub, 1, &r, &sc) != CC_OK) die();
    h.cand_gidx = it->second;  // recorded by the manager's own numbering
    hints_.push_back(h);
    cand_aft_check1 += r.cand_aft_check1;
    cand_aft_check2 += r.cand_aft_check2;
    cand_aft_check3 += r.cand_aft_check3;
    CandidateScoreEnsemble ret;
    ret.sim_constell.i_ovlp_sum = sc.i_ovlp_sum;
    ret.sim_constell.i_ovlp_max_one = sc.i_ovlp_max_one;
    ret.sim_constell.i_in_ang_rng = sc.i_in_ang_rng;
    ret.sim_pair.i_indiv_sim = sc.i_indiv_sim;
    ret.sim_pair.i_orie_sim = sc.i_orie_sim;
    return ret;
  }

  // contour_db.h:494-596.  The work happens in fineOptimize (one device pass over the recorded hints).
  void tidyUpCandidates() {
    CC_CHECK(flow_valve < 1);
    flow_valve++;
  }

  // contour_db.h:604-648: returns the number of results (0 or 1)
  int fineOptimize(int max_fine_opt, std::vector<std::shared_ptr<const ContourManager>> &res_cand, std::vector<double> &res_corr,
                   std::vector<Eigen::Isometry2d> &res_T) {
    CC_CHECK(flow_valve == 1);
    flow_valve++;
    res_cand.clear();
    res_corr.clear();
    res_T.clear();
    if (hints_.empty() || !st_) return 0;
    const cc_score_t lb = to_c(sim_var_), ub = to_c(sim_ub_);
    cc_query_result_t r;
    ensureMine();
    std::vector<cc_hint_t> hs(hints_);
    for (auto &h : hs) h.cand_gidx = st_->pos[my_cands_[h.cand_gidx].get()];
    if (cc_db_check_hints_host(st_->db, &cm_tgt_->desc(), hs.data(), (int)hs.size(), &lb, &ub, max_fine_opt, &r, nullptr) != CC_OK)
      die();
    if (r.n_res > 0) {
      res_cand.push_back(st_->keep[r.cand_gidx]);
      res_corr.push_back(r.correlation);
      Eigen::Isometry2d T;
      T.rotate(r.tf[2]);
      T.pretranslate(r.tf[0], r.tf[1]);
      res_T.push_back(T);
    }
    return r.n_res;
  }
};

// correlation.h:287-296
struct ConstellCorrelation {
  static Eigen::Isometry2d getEstSensTF(const Eigen::Isometry2d &T_delta, const ContourManagerConfig &bev_config) {
    CC_CHECK(bev_config.reso_row_ == bev_config.reso_col_);
    const double in[3] = {T_delta(0, 2), T_delta(1, 2), std::atan2(T_delta(1, 0), T_delta(0, 0))};
    double out[3];
    cc_est_sens_tf(in, bev_config.n_row_, bev_config.n_col_, out);
    Eigen::Isometry2d T;
    T.rotate(out[2]);
    T.pretranslate(out[0], out[1]);
    return T;
  }

  // correlation.h:241-280: error of the estimated sensor-to-sensor transform (metres, radians) against the 3-D ground
  // truth projected to 2-D (xy offset; yaw after rotating the relative z axis back onto z).  Returns T_gt^-1 * T_est.
  static Eigen::Isometry2d evalMetricEst(const Eigen::Isometry2d &T_delta, const Eigen::Isometry3d &gt_src_3d,
                                         const Eigen::Isometry3d &gt_tgt_3d, const ContourManagerConfig &bev_config) {
    CC_CHECK(bev_config.reso_row_ == bev_config.reso_col_);
    Eigen::Isometry2d T_est = getEstSensTF(T_delta, bev_config);
    T_est.m[0][2] *= bev_config.reso_row_;
    T_est.m[1][2] *= bev_config.reso_row_;
    const Eigen::Isometry3d rel = gt_tgt_3d.inverse() * gt_src_3d;  // src sensor in the tgt sensor frame
    // axis-angle that takes the relative z axis back to (0,0,1): axis = z0 x z1 normalised, angle = -acos(z0 . z1)
    const double z1[3] = {rel.R[0][2], rel.R[1][2], rel.R[2][2]};
    double ax[3] = {-z1[1], z1[0], 0.0};
    const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1]);
    if (an > 0) {
      ax[0] /= an;
      ax[1] /= an;
    }
    const double ang = -std::acos(z1[2]);
    const double c = std::cos(ang), s = std::sin(ang), C = 1 - c;
    const double D[3][3] = {{c + ax[0] * ax[0] * C, ax[0] * ax[1] * C - ax[2] * s, ax[0] * ax[2] * C + ax[1] * s},
                            {ax[1] * ax[0] * C + ax[2] * s, c + ax[1] * ax[1] * C, ax[1] * ax[2] * C - ax[0] * s},
                            {ax[2] * ax[0] * C - ax[1] * s, ax[2] * ax[1] * C + ax[0] * s, c + ax[2] * ax[2] * C}};
    double Rr[2][2];  // top-left 2x2 of D * rel.R
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) Rr[i][j] = D[i][0] * rel.R[0][j] + D[i][1] * rel.R[1][j] + D[i][2] * rel.R[2][j];
    Eigen::Isometry2d T_gt;
    T_gt.rotate(std::atan2(Rr[1][0], Rr[0][0]));
    T_gt.pretranslate(rel.t[0], rel.t[1]);
    return T_gt.inverse() * T_est;
  }
};
