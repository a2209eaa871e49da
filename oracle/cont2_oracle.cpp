// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// C-ABI (ctypes-friendly) around the CPU restatement of the reference hot path
// (orc_contour.h / orc_gmm.h / orc_db.h).  Used by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg, only as the checker / the timed CPU baseline.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "orc_contour.h"
#include "orc_db.h"
#include "orc_gmm.h"

using namespace orc;

namespace {
ContourManagerConfig toMngCfg(const cc_manager_cfg_t *c) {
  ContourManagerConfig m;
  m.lv_grads_.assign(c->lv_grads, c->lv_grads + CC_NLEV);
  m.reso_row_ = c->reso_row;
  m.reso_col_ = c->reso_col;
  m.n_row_ = c->n_row;
  m.n_col_ = c->n_col;
  m.lidar_height_ = c->lidar_height;
  m.blind_sq_ = c->blind_sq;
  m.min_cont_key_cnt_ = c->min_cont_key_cnt;
  m.min_cont_cell_cnt_ = c->min_cont_cell_cnt;
  m.piv_firsts_ = c->piv_firsts;
  m.dist_firsts_ = c->dist_firsts;
  m.roi_radius_ = c->roi_radius;
  return m;
}
ContourDBConfig toDbCfg(const cc_db_cfg_t *c) {
  ContourDBConfig d;
  d.nnk_ = c->nnk;
  d.max_fine_opt_ = c->max_fine_opt;
  d.q_levels_.assign(c->q_levels, c->q_levels + c->n_q_levels);
  d.cont_sim_cfg_.ta_cell_cnt = c->cont_sim.ta_cell_cnt;
  d.cont_sim_cfg_.tp_cell_cnt = c->cont_sim.tp_cell_cnt;
  d.cont_sim_cfg_.tp_eigval = c->cont_sim.tp_eigval;
  d.cont_sim_cfg_.ta_h_bar = c->cont_sim.ta_h_bar;
  d.cont_sim_cfg_.ta_rcom = c->cont_sim.ta_rcom;
  d.cont_sim_cfg_.tp_rcom = c->cont_sim.tp_rcom;
  d.tb_cfg_.max_elapse_ = c->max_elapse;
  d.tb_cfg_.min_elapse_ = c->min_elapse;
  return d;
}
CandidateScoreEnsemble toScore(const cc_score_t *s) {
  CandidateScoreEnsemble e;
  e.sim_constell.i_ovlp_sum = s->i_ovlp_sum;
  e.sim_constell.i_ovlp_max_one = s->i_ovlp_max_one;
  e.sim_constell.i_in_ang_rng = s->i_in_ang_rng;
  e.sim_pair.i_indiv_sim = s->i_indiv_sim;
  e.sim_pair.i_orie_sim = s->i_orie_sim;
  e.sim_post.correlation = s->correlation;
  e.sim_post.area_perc = s->area_perc;
  e.sim_post.neg_est_dist = s->neg_est_dist;
  return e;
}
struct ScanH {
  std::shared_ptr<ContourManager> cm;
};
struct DbH {
  std::unique_ptr<ContourDB> db;
  StageTimers timers;
};
}  // namespace

extern "C" {

// ---- ingest: ContourManager ctor + makeBEV + makeContoursRecurs (evaluator.h:285-302) ----
void *orc_scan_create(const float *xyzi, int64_t n_pts, const cc_manager_cfg_t *cfg, int int_id, int keep_cells) {
  auto *h = new ScanH();
  h->cm = std::make_shared<ContourManager>(toMngCfg(cfg), int_id);
  h->cm->keep_cells_ = keep_cells != 0;
  if (!h->cm->makeBEV(xyzi, n_pts)) {
    delete h;
    return nullptr;
  }
  h->cm->makeContoursRecurs();
  return h;
}
// Rebuild a ContourManager from a descriptor (contour tables, keys, BCIs): lets the oracle's DB / query code run on
// descriptors produced elsewhere (e.g. hand-made keys for bookkeeping tests, or the device's output).
void *orc_scan_from_desc(const cc_scan_desc_t *d, const cc_manager_cfg_t *cfg, int int_id) {
  auto *h = new ScanH();
  h->cm = std::make_shared<ContourManager>(toMngCfg(cfg), int_id);
  ContourManager &cm = *h->cm;
  cm.clearImage();
  cm.max_bin_val_ = d->max_bin_val;
  cm.min_bin_val_ = d->min_bin_val;
  for (int l = 0; l < CC_NLEV; l++) {
    cm.layer_cell_cnt_[l] = d->layer_cell_cnt[l];
    for (int j = 0; j < d->n_stored[l]; j++) {
      const cc_contour_t &c = d->cont[l][j];
      auto v = std::make_shared<ContourView>(c.level, c.poi[0], c.poi[1]);
      v->cell_cnt_ = c.cell_cnt;
      v->pos_mean_ = V2F(c.pos_mean[0], c.pos_mean[1]);
      v->pos_cov_.a[0][0] = c.pos_cov[0];
      v->pos_cov_.a[1][0] = c.pos_cov[1];
      v->pos_cov_.a[0][1] = c.pos_cov[2];
      v->pos_cov_.a[1][1] = c.pos_cov[3];
      v->eig_vals_ = V2F(c.eig_vals[0], c.eig_vals[1]);
      v->eig_vecs_.a[0][0] = c.eig_vecs[0];
      v->eig_vecs_.a[1][0] = c.eig_vecs[1];
      v->eig_vecs_.a[0][1] = c.eig_vecs[2];
      v->eig_vecs_.a[1][1] = c.eig_vecs[3];
      v->eccen_ = c.eccen;
      v->vol3_mean_ = c.vol3_mean;
      v->com_ = V2F(c.com[0], c.com[1]);
      v->ecc_feat_ = c.ecc_feat != 0;
      v->com_feat_ = c.com_feat != 0;
      cm.cont_views_[l].push_back(v);
      cm.cont_perc_[l].push_back(c.cell_cnt * 1.0f / d->layer_cell_cnt[l]);
    }
    for (int s = 0; s < cm.cfg_.piv_firsts_; s++) {
      RetrievalKey k;
      for (int q = 0; q < CC_KEY_DIM; q++) k.array[q] = d->keys[l][s][q];
      cm.layer_keys_[l].push_back(k);
      const cc_bci_t &b = d->bcis[l][s];
      BCI bci(b.piv_seq, b.level);
      for (int w = 0; w < CC_BCI_LAYERS; w++)
        for (int bit = 0; bit < 64; bit++)
          if ((b.dist_bin[w] >> bit) & 1ull) bci.dist_bin_.set(w * 64 + bit, true);
      for (int q = 0; q < b.n_pts; q++) bci.nei_pts_.emplace_back(b.pts[q].level, b.pts[q].seq, b.pts[q].bit_pos, b.pts[q].r, b.pts[q].theta);
      for (int q = 0; q < b.n_segs; q++) bci.nei_idx_segs_.push_back(b.segs[q]);
      cm.layer_key_bcis_[l].push_back(bci);
    }
  }
  return h;
}
void orc_scan_free(void *h) { delete (ScanH *)h; }
void orc_scan_export(void *h, cc_scan_desc_t *out) { ((ScanH *)h)->cm->exportDesc(out); }
int orc_scan_ncont(void *h, int level) { return (int)((ScanH *)h)->cm->cont_views_[level].size(); }
// bev: [n_row*n_col] ; pix_rc: [n_row*n_col][2] (-1 where no pixel)
void orc_scan_bev(void *h, float *bev, float *pix_rc) {
  ContourManager &cm = *((ScanH *)h)->cm;
  size_t n = (size_t)cm.cfg_.n_row_ * cm.cfg_.n_col_;
  if (bev && !cm.bev_.empty()) std::memcpy(bev, cm.bev_.data(), n * sizeof(float));
  if (pix_rc) {
    for (size_t i = 0; i < 2 * n; i++) pix_rc[i] = -1.f;
    for (auto &p : cm.bev_pixfs_) {
      pix_rc[2 * p.first] = p.second.row_f;
      pix_rc[2 * p.first + 1] = p.second.col_f;
    }
  }
}
void orc_scan_labels(void *h, int16_t *labels) { ((ScanH *)h)->cm->exportLabels(labels); }
void orc_scan_clear_image(void *h) { ((ScanH *)h)->cm->clearImage(); }

// ---- database ----
void *orc_db_create(const cc_db_cfg_t *cfg) {
  auto *d = new DbH();
  d->db.reset(new ContourDB(toDbCfg(cfg)));
  d->db->timers = &d->timers;
  return d;
}
// sensitivity knobs of the restated third-party pieces (orc_math.h: orc::Variant); 0 / 10 / 1e-4 / 0.9 = the restatement
void orc_set_variant(unsigned label_shuffle_seed, int lbfgs_max_iterations, double wolfe_c1, double wolfe_c2) {
  orc::Variant &v = orc::variant();
  v.label_shuffle_seed = label_shuffle_seed;
  v.lbfgs_max_iterations = lbfgs_max_iterations;
  v.wolfe_sufficient_decrease = wolfe_c1;
  v.wolfe_curvature = wolfe_c2;
}
void orc_db_free(void *d) { delete (DbH *)d; }
// the query-side stage timers accumulated so far {KNN search, Constell, L2 opt} (seconds); reset != 0 clears them
void orc_db_timers(void *d, double *out3, int reset) {
  StageTimers &t = ((DbH *)d)->timers;
  out3[0] = t.knn_search;
  out3[1] = t.constell;
  out3[2] = t.l2_opt;
  if (reset) t = StageTimers();
}
void orc_db_add_scan(void *d, void *scan, double ts) { ((DbH *)d)->db->addScan(((ScanH *)scan)->cm, ts); }
void orc_db_push_and_balance(void *d, int seed, double ts) { ((DbH *)d)->db->pushAndBalance(seed, ts); }
void orc_db_bucket_state(void *d, int32_t *tree_sizes, float *ranges) {
  ContourDB &db = *((DbH *)d)->db;
  for (size_t l = 0; l < db.layer_db_.size(); l++) {
    for (int b = 0; b < 6; b++) tree_sizes[l * 6 + b] = (int32_t)db.layer_db_[l].buckets_[b].getTreeSize();
    for (int b = 0; b < 7; b++) ranges[l * 7 + b] = db.layer_db_[l].bucket_ranges_[b];
  }
}
// gidx of the returned candidate = position in all_bevs_ (found by pointer identity)
void orc_db_query(void *d, void *scan, const cc_score_t *lb, const cc_score_t *ub, cc_query_result_t *res,
                  cc_knn_hit_t *knn /*[3][6][CC_KNN_MAX] or NULL*/, int32_t *knn_cnt /*[3][6] or NULL*/) {
  ContourDB &db = *((DbH *)d)->db;
  std::vector<std::shared_ptr<const ContourManager>> cands;
  std::vector<double> corr;
  std::vector<Iso2d> tfs;
  ContourDB::QueryDebug dbg;
  db.queryRangedKNN(((ScanH *)scan)->cm, toScore(lb), toScore(ub), cands, corr, tfs, &dbg);
  std::memset(res, 0, sizeof(*res));
  res->cand_gidx = -1;
  res->n_res = (int)cands.size();
  if (!cands.empty()) {
    for (size_t i = 0; i < db.all_bevs_.size(); i++)
      if (db.all_bevs_[i].get() == cands[0].get()) {
        res->cand_gidx = (int)i;
        break;
      }
    res->correlation = corr[0];
    res->tf[0] = tfs[0](0, 2);
    res->tf[1] = tfs[0](1, 2);
    res->tf[2] = std::atan2(tfs[0](1, 0), tfs[0](0, 0));
  }
  res->cand_aft_check1 = dbg.chk1;
  res->cand_aft_check2 = dbg.chk2;
  res->cand_aft_check3 = dbg.chk3;
  res->n_cand_pose = dbg.n_cand_pose;
  res->n_cand_tidy = dbg.n_cand_tidy;
  int piv = ((ScanH *)scan)->cm->getConfig().piv_firsts_;
  int total = 0;
  for (size_t k = 0; k < dbg.knn.size(); k++) {
    total += (int)dbg.knn[k].size();
    int ll = (int)k / piv, seq = (int)k % piv;
    if (knn_cnt) knn_cnt[ll * CC_NPIV + seq] = (int)dbg.knn[k].size();
    if (knn)
      for (size_t j = 0; j < dbg.knn[k].size() && j < CC_KNN_MAX; j++) {
        cc_knn_hit_t &h = knn[(ll * CC_NPIV + seq) * CC_KNN_MAX + j];
        h.gidx = (int32_t)dbg.knn[k][j].first.gidx;
        h.level = (int16_t)dbg.knn[k][j].first.level;
        h.seq = (int16_t)dbg.knn[k][j].first.seq;
        h.dist_sq = dbg.knn[k][j].second;
      }
  }
  res->n_knn_hits = total;
}

// ---- the reference driver loop, timed with the five reference stage names
//      (test/batch_bin_test.cpp:105-247): per scan: make bev -> clearImage -> query -> addScan ->
//      pushAndBalance.  timers_out[5] = {make bev, KNN search, Constell, L2 opt, Update database}
//      in seconds (totals over the run).  desc_out optional [n].
int orc_run_sequence(const float *xyzi, const int64_t *offsets, int n_scans, const double *ts, const int32_t *seeds,
                     const cc_manager_cfg_t *mcfg, const cc_db_cfg_t *dcfg, const cc_score_t *lb, const cc_score_t *ub,
                     cc_query_result_t *results, double *timers_out, cc_scan_desc_t *desc_out) {
  ContourManagerConfig mc = toMngCfg(mcfg);
  ContourDB db(toDbCfg(dcfg));
  StageTimers tm;
  db.timers = &tm;
  CandidateScoreEnsemble elb = toScore(lb), eub = toScore(ub);
  for (int i = 0; i < n_scans; i++) {
    double t0 = StageTimers::now();
    auto cm = std::make_shared<ContourManager>(mc, seeds[i]);
    if (!cm->makeBEV(xyzi + 4 * offsets[i], offsets[i + 1] - offsets[i])) return -1;
    cm->makeContoursRecurs();
    tm.make_bev += StageTimers::now() - t0;
    if (desc_out) cm->exportDesc(&desc_out[i]);
    cm->clearImage();
    std::vector<std::shared_ptr<const ContourManager>> cands;
    std::vector<double> corr;
    std::vector<Iso2d> tfs;
    ContourDB::QueryDebug dbg;
    db.queryRangedKNN(cm, elb, eub, cands, corr, tfs, &dbg);
    cc_query_result_t &r = results[i];
    std::memset(&r, 0, sizeof(r));
    r.cand_gidx = -1;
    r.n_res = (int)cands.size();
    if (!cands.empty()) {
      for (size_t k = 0; k < db.all_bevs_.size(); k++)
        if (db.all_bevs_[k].get() == cands[0].get()) {
          r.cand_gidx = (int)k;
          break;
        }
      r.correlation = corr[0];
      r.tf[0] = tfs[0](0, 2);
      r.tf[1] = tfs[0](1, 2);
      r.tf[2] = std::atan2(tfs[0](1, 0), tfs[0](0, 0));
    }
    r.cand_aft_check1 = dbg.chk1;
    r.cand_aft_check2 = dbg.chk2;
    r.cand_aft_check3 = dbg.chk3;
    r.n_cand_pose = dbg.n_cand_pose;
    r.n_cand_tidy = dbg.n_cand_tidy;
    int total = 0;
    for (auto &k : dbg.knn) total += (int)k.size();
    r.n_knn_hits = total;
    double t1 = StageTimers::now();
    db.addScan(cm, ts[i]);
    db.pushAndBalance(seeds[i], ts[i]);
    tm.update_db += StageTimers::now() - t1;
  }
  if (timers_out) {
    timers_out[0] = tm.make_bev;
    timers_out[1] = tm.knn_search;
    timers_out[2] = tm.constell;
    timers_out[3] = tm.l2_opt;
    timers_out[4] = tm.update_db;
  }
  return 0;
}

// ingest only (CPU baseline of the ingest stage; also used to build fixtures)
int orc_ingest_batch(const float *xyzi, const int64_t *offsets, int n_scans, const cc_manager_cfg_t *mcfg,
                     cc_scan_desc_t *desc_out) {
  ContourManagerConfig mc = toMngCfg(mcfg);
  for (int i = 0; i < n_scans; i++) {
    ContourManager cm(mc, i);
    if (!cm.makeBEV(xyzi + 4 * offsets[i], offsets[i + 1] - offsets[i])) return -1;
    cm.makeContoursRecurs();
    if (desc_out) cm.exportDesc(&desc_out[i]);
  }
  return 0;
}

// ---- unit hooks for the parity tests ----
// std::atan2(float, float) as the reference's BCI build calls it (contour_mng.h:860), on arrays: the yardstick of the
// device's atan2f replica (tests/test_atan2f_replica.py)
void orc_atan2f(const float *y, const float *x, float *out, long n) {
  for (long i = 0; i < n; i++) out[i] = std::atan2(y[i], x[i]);
}
// device's acosf replica (tests/test_atan2f_replica.py): this libm's acosf on an array
void orc_acosf(const float *x, float *out, long n) {
  for (long i = 0; i < n; i++) out[i] = std::acos(x[i]);
}

void orc_eigen2f(const float m[4] /*a00 a01 a10 a11*/, float evals[2], float evecs[4] /*row-major*/) {
  M2F mm, ev;
  mm.a[0][0] = m[0];
  mm.a[0][1] = m[1];
  mm.a[1][0] = m[2];
  mm.a[1][1] = m[3];
  selfAdjointEigen2f(mm, evals, ev);
  evecs[0] = ev.a[0][0];
  evecs[1] = ev.a[0][1];
  evecs[2] = ev.a[1][0];
  evecs[3] = ev.a[1][1];
}
// the oracle's restatement of cv::connectedComponentsWithStats(img, labels, stats, centroids, 8, CV_32S) on one patch
// (orc_contour.h:connectedComponentsWithStats8): label image out, returns the number of labels incl. background.  For
// tests/test_oracle_ccl_second_restatement.py, which checks the numbering against an independent two-pass restatement.
int orc_ccl8(const uint8_t *img, int rows, int cols, int32_t *labels_out, int32_t *stats_out /*[n][5] or null*/) {
  std::vector<uint8_t> im(img, img + (size_t)rows * cols);
  std::vector<int> labels;
  std::vector<std::array<int, 5>> stats;
  const int n = connectedComponentsWithStats8(im, rows, cols, labels, stats);
  for (size_t i = 0; i < labels.size(); i++) labels_out[i] = labels[i];
  if (stats_out)
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 5; k++) stats_out[i * 5 + k] = stats[i][k];
  return n;
}
// permutation produced by std::sort with comparator key[a] > key[b] (contour_mng.h:596-599)
void orc_sort_desc_perm(const int32_t *keys, int n, int32_t *perm) {
  std::vector<std::pair<int32_t, int32_t>> v(n);
  for (int i = 0; i < n; i++) v[i] = {keys[i], i};
  std::sort(v.begin(), v.end(),
            [](const std::pair<int32_t, int32_t> &a, const std::pair<int32_t, int32_t> &b) { return a.first > b.first; });
  for (int i = 0; i < n; i++) perm[i] = v[i].second;
}
// permutation produced by std::sort with comparator key[a] < key[b] on float keys (contour_mng.h:340, :871)
void orc_sort_asc_perm_f(const float *keys, int n, int32_t *perm) {
  std::vector<std::pair<float, int32_t>> v(n);
  for (int i = 0; i < n; i++) v[i] = {keys[i], i};
  std::sort(v.begin(), v.end(), [](const std::pair<float, int32_t> &a, const std::pair<float, int32_t> &b) { return a.first < b.first; });
  for (int i = 0; i < n; i++) perm[i] = v[i].second;
}
// checkCandWithHint on a fresh CandidateManager: scores + proposal
// out_i[0..4] = ovlp_sum, max_one, in_ang_rng, indiv_sim, orie_sim ; out_i[5] = passed(0/1) ; out_i[6] = n pairs
// out_tf[3] = x,y,theta of T_pass ; pairs[3*k..] = level,seq_src,seq_tgt
void orc_check_pair(void *cand, void *tgt, int level, int seq_src, int seq_tgt, const cc_sim_cfg_t *sim, const cc_score_t *lb,
                    const cc_score_t *ub, int32_t *out_i, double *out_tf, int8_t *pairs) {
  ContourSimThresConfig cs;
  cs.ta_cell_cnt = sim->ta_cell_cnt;
  cs.tp_cell_cnt = sim->tp_cell_cnt;
  cs.tp_eigval = sim->tp_eigval;
  cs.ta_h_bar = sim->ta_h_bar;
  cs.ta_rcom = sim->ta_rcom;
  cs.tp_rcom = sim->tp_rcom;
  CandidateManager mng(((ScanH *)tgt)->cm, toScore(lb), toScore(ub));
  CandidateScoreEnsemble s = mng.checkCandWithHint(((ScanH *)cand)->cm, ConstellationPair(level, seq_src, seq_tgt), cs);
  out_i[0] = s.sim_constell.i_ovlp_sum;
  out_i[1] = s.sim_constell.i_ovlp_max_one;
  out_i[2] = s.sim_constell.i_in_ang_rng;
  out_i[3] = s.sim_pair.i_indiv_sim;
  out_i[4] = s.sim_pair.i_orie_sim;
  out_i[5] = mng.candidates_.empty() ? 0 : 1;
  out_i[6] = 0;
  if (!mng.candidates_.empty()) {
    const auto &p = mng.candidates_[0].anch_props_[0];
    out_tf[0] = p.T_delta_(0, 2);
    out_tf[1] = p.T_delta_(1, 2);
    out_tf[2] = std::atan2(p.T_delta_(1, 0), p.T_delta_(0, 0));
    int k = 0;
    for (auto &kv : p.constell_) {
      pairs[3 * k] = kv.first.level;
      pairs[3 * k + 1] = kv.first.seq_src;
      pairs[3 * k + 2] = kv.first.seq_tgt;
      k++;
    }
    out_i[6] = k;
  }
}
// The single-pair flow of test/kitti_read_bin_test.cpp:226-291: one CandidateManager for the query `tgt`,
// checkCandWithHint for every hint in the given order, tidyUpCandidates, fineOptimize(max_fine_opt).
// hints[i] = {index into cands[], level, seq_src, seq_tgt}; scores[i] = {ovlp_sum, max_one, in_ang_rng, indiv_sim,
// orie_sim, passed}; res->cand_gidx = index into cands[].
void orc_check_hints(void *tgt, void **cands, int n_cands, const int32_t *hints /*[n][4]*/, int n_hints, const cc_sim_cfg_t *sim,
                     const cc_score_t *lb, const cc_score_t *ub, int max_fine_opt, cc_query_result_t *res, int32_t *scores /*[n][6]*/) {
  ContourSimThresConfig cs;
  cs.ta_cell_cnt = sim->ta_cell_cnt;
  cs.tp_cell_cnt = sim->tp_cell_cnt;
  cs.tp_eigval = sim->tp_eigval;
  cs.ta_h_bar = sim->ta_h_bar;
  cs.ta_rcom = sim->ta_rcom;
  cs.tp_rcom = sim->tp_rcom;
  CandidateManager mng(((ScanH *)tgt)->cm, toScore(lb), toScore(ub));
  for (int i = 0; i < n_hints; i++) {
    const int32_t *h = hints + 4 * i;
    const int before = mng.cand_aft_check3;
    CandidateScoreEnsemble s = mng.checkCandWithHint(((ScanH *)cands[h[0]])->cm, ConstellationPair(h[1], h[2], h[3]), cs);
    if (scores) {
      int32_t *o = scores + 6 * i;
      o[0] = s.sim_constell.i_ovlp_sum;
      o[1] = s.sim_constell.i_ovlp_max_one;
      o[2] = s.sim_constell.i_in_ang_rng;
      o[3] = s.sim_pair.i_indiv_sim;
      o[4] = s.sim_pair.i_orie_sim;
      o[5] = mng.cand_aft_check3 > before ? 1 : 0;
    }
  }
  std::memset(res, 0, sizeof(*res));
  res->cand_gidx = -1;
  res->cand_aft_check1 = mng.cand_aft_check1;
  res->cand_aft_check2 = mng.cand_aft_check2;
  res->cand_aft_check3 = mng.cand_aft_check3;
  mng.tidyUpCandidates();
  res->n_cand_pose = mng.n_cand_pose;
  res->n_cand_tidy = (int)mng.candidates_.size();
  res->n_knn_hits = n_hints;
  std::vector<std::shared_ptr<const ContourManager>> rc;
  std::vector<double> corr;
  std::vector<Iso2d> tfs;
  res->n_res = mng.fineOptimize(max_fine_opt, rc, corr, tfs);
  if (res->n_res) {
    for (int i = 0; i < n_cands; i++)
      if (((ScanH *)cands[i])->cm.get() == rc[0].get()) {
        res->cand_gidx = i;
        break;
      }
    res->correlation = corr[0];
    res->tf[0] = tfs[0](0, 2);
    res->tf[1] = tfs[0](1, 2);
    res->tf[2] = std::atan2(tfs[0](1, 0), tfs[0](0, 0));
  }
}
// GMM: init correlation at T_init and the refined (correlation, x, y, theta) after <=10 L-BFGS iterations
void orc_gmm(void *src, void *tgt, const double tf_init[3], double *corr_init, double *corr_opt, double tf_opt[3],
             int32_t *iters) {
  ConstellCorrelation cc((GMMOptConfig()));
  Iso2d T = Iso2d::fromAngTrans(tf_init[2], V2D(tf_init[0], tf_init[1]));
  *corr_init = cc.initProblem(*((ScanH *)src)->cm, *((ScanH *)tgt)->cm, T);
  ceres_like::SolveSummary sum;
  auto r = cc.calcCorrelation(&sum);
  *corr_opt = r.first;
  tf_opt[0] = r.second(0, 2);
  tf_opt[1] = r.second(1, 2);
  tf_opt[2] = std::atan2(r.second(1, 0), r.second(0, 0));
  if (iters) {
    iters[0] = sum.iterations;
    iters[1] = sum.termination;
    iters[2] = sum.n_eval;
  }
}
// the accepted points of the refinement (x_0 .. x_n, n = iterations), for tests/test_oracle_linesearch_properties.py
int orc_gmm_trace(void *src, void *tgt, const double tf_init[3], double *xs /*[11][3]*/, int32_t *termination) {
  ConstellCorrelation cc((GMMOptConfig()));
  Iso2d T = Iso2d::fromAngTrans(tf_init[2], V2D(tf_init[0], tf_init[1]));
  cc.initProblem(*((ScanH *)src)->cm, *((ScanH *)tgt)->cm, T);
  ceres_like::SolveSummary sum;
  cc.calcCorrelation(&sum);
  int n = 0;
  for (const auto &x : sum.iterates) {
    if (n >= 11) break;
    xs[n * 3] = x[0];
    xs[n * 3 + 1] = x[1];
    xs[n * 3 + 2] = x[2];
    n++;
  }
  if (termination) *termination = sum.termination;
  return n;
}
// cost + gradient of the GMM functor at p (autodiff restatement), for cross-checks against scipy
void orc_gmm_eval(void *src, void *tgt, const double tf_init[3], const double p[3], double *cost, double grad[3],
                  double autocorr[2]) {
  Iso2d T = Iso2d::fromAngTrans(tf_init[2], V2D(tf_init[0], tf_init[1]));
  GMMPair g(*((ScanH *)src)->cm, *((ScanH *)tgt)->cm, GMMOptConfig(), T);
  g.evaluate(p, cost, grad);
  autocorr[0] = g.auto_corr_src_;
  autocorr[1] = g.auto_corr_tgt_;
}
void orc_umeyama(void *src, void *tgt, const int8_t *pairs, int n, double tf[3]) {
  std::vector<ConstellationPair> c;
  for (int i = 0; i < n; i++) c.emplace_back(pairs[3 * i], pairs[3 * i + 1], pairs[3 * i + 2]);
  Iso2d T = ContourManager::getTFFromConstell(*((ScanH *)src)->cm, *((ScanH *)tgt)->cm, c);
  tf[0] = T(0, 2);
  tf[1] = T(1, 2);
  tf[2] = std::atan2(T(1, 0), T(0, 0));
}
// exact-scan KNN of the restatement on a raw key matrix (validated against real nanoflann in oracle/_ref)
int orc_knn_scan(const float *keys, int n, const float *q, int k, float max_dist_sq, int32_t *idx_out, float *dist_out) {
  TreeBucket tb(TreeBucketConfig(), -1000.f, 1000.f);
  for (int i = 0; i < n; i++) {
    RetrievalKey rk;
    std::memcpy(rk.array, keys + 10 * i, 40);
    tb.data_tree_.push_back(rk);
    tb.gkidx_tree_.emplace_back((size_t)i, 0, 0);
  }
  tb.tree_built = true;
  RetrievalKey qk;
  std::memcpy(qk.array, q, 40);
  std::vector<IndexOfKey> ri;
  std::vector<KeyFloatType> rd;
  tb.knnSearch(k, ri, rd, qk, max_dist_sq);
  int cnt = 0;
  for (int j = 0; j < k; j++) {
    if (rd[j] < max_dist_sq) {
      idx_out[cnt] = (int32_t)ri[j].gidx;
      dist_out[cnt] = rd[j];
      cnt++;
    } else
      break;
  }
  return cnt;
}
size_t orc_sizeof_desc(void) { return sizeof(cc_scan_desc_t); }
// install / remove the real-nanoflann kd-tree backend (function pointers from oracle/_ref/libref_knn.so)
void orc_set_knn_backend(void *create, void *destroy, void *build, void *query) {
  KnnBackend &b = knn_backend();
  b.create = (void *(*)())create;
  b.destroy = (void (*)(void *))destroy;
  b.build = (void (*)(void *, const float *, int))build;
  b.query = (void (*)(void *, const float *, int, float, size_t *, float *))query;
}
int orc_knn_backend_active(void) { return knn_backend().create != nullptr; }

}  // extern "C"
