// ROS-free counterpart of the reference's offline driver (test/batch_bin_test.cpp:30-330): reads the same YAML keys,
// feeds the scans of the list file through ContourManager / ContourDB (device path) in order -- query first, then
// insert -- judges every prediction with ContLCDEvaluator and writes the outcome file scripts/pr_mpe.py consumes.
//   batch_bin_test <config.yaml>
#include "eval/evaluator.h"
#include "tools/config_handler.h"

SequentialTimeProfiler stp;  // the library's stage timers land here (contour_db.h: extern)

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s config.yaml\n", argv[0]);
    return 2;
  }
  ContourManagerConfig cm_config;
  ContourDBConfig db_config;
  CandidateScoreEnsemble thres_lb_, thres_ub_;
  std::string fpath_sens_gt_pose, fpath_lidar_bins, sav_path;
  double corr_thres = 0;
  printf("Loading parameters...\n");
  {
    yamlLoader yl(argv[1]);
    if (!yl.opened) {
      fprintf(stderr, "cannot open %s\n", argv[1]);
      return 2;
    }
    yl.loadOneConfig({"fpath_sens_gt_pose"}, fpath_sens_gt_pose);
    yl.loadOneConfig({"fpath_lidar_bins"}, fpath_lidar_bins);
    yl.loadOneConfig({"correlation_thres"}, corr_thres);
    yl.loadOneConfig({"ContourDBConfig", "nnk_"}, db_config.nnk_);
    yl.loadOneConfig({"ContourDBConfig", "max_fine_opt_"}, db_config.max_fine_opt_);
    yl.loadSeqConfig({"ContourDBConfig", "q_levels_"}, db_config.q_levels_);
    yl.loadOneConfig({"ContourDBConfig", "TreeBucketConfig", "max_elapse_"}, db_config.tb_cfg_.max_elapse_);
    yl.loadOneConfig({"ContourDBConfig", "TreeBucketConfig", "min_elapse_"}, db_config.tb_cfg_.min_elapse_);
    for (auto kv : {std::make_pair("ta_cell_cnt", &db_config.cont_sim_cfg_.ta_cell_cnt), std::make_pair("tp_cell_cnt", &db_config.cont_sim_cfg_.tp_cell_cnt),
                    std::make_pair("tp_eigval", &db_config.cont_sim_cfg_.tp_eigval), std::make_pair("ta_h_bar", &db_config.cont_sim_cfg_.ta_h_bar),
                    std::make_pair("ta_rcom", &db_config.cont_sim_cfg_.ta_rcom), std::make_pair("tp_rcom", &db_config.cont_sim_cfg_.tp_rcom)})
      yl.loadOneConfig({"ContourDBConfig", "ContourSimThresConfig", kv.first}, *kv.second);
    for (auto side : {std::make_pair("thres_lb_", &thres_lb_), std::make_pair("thres_ub_", &thres_ub_)}) {
      CandidateScoreEnsemble &e = *side.second;
      yl.loadOneConfig({side.first, "i_ovlp_sum"}, e.sim_constell.i_ovlp_sum);
      yl.loadOneConfig({side.first, "i_ovlp_max_one"}, e.sim_constell.i_ovlp_max_one);
      yl.loadOneConfig({side.first, "i_in_ang_rng"}, e.sim_constell.i_in_ang_rng);
      yl.loadOneConfig({side.first, "i_indiv_sim"}, e.sim_pair.i_indiv_sim);
      yl.loadOneConfig({side.first, "i_orie_sim"}, e.sim_pair.i_orie_sim);
      yl.loadOneConfig({side.first, "correlation"}, e.sim_post.correlation);
      yl.loadOneConfig({side.first, "area_perc"}, e.sim_post.area_perc);
      yl.loadOneConfig({side.first, "neg_est_dist"}, e.sim_post.neg_est_dist);
    }
    yl.loadSeqConfig({"ContourManagerConfig", "lv_grads_"}, cm_config.lv_grads_);
    yl.loadOneConfig({"ContourManagerConfig", "reso_row_"}, cm_config.reso_row_);
    yl.loadOneConfig({"ContourManagerConfig", "reso_col_"}, cm_config.reso_col_);
    yl.loadOneConfig({"ContourManagerConfig", "n_row_"}, cm_config.n_row_);
    yl.loadOneConfig({"ContourManagerConfig", "n_col_"}, cm_config.n_col_);
    yl.loadOneConfig({"ContourManagerConfig", "lidar_height_"}, cm_config.lidar_height_);
    yl.loadOneConfig({"ContourManagerConfig", "blind_sq_"}, cm_config.blind_sq_);
    yl.loadOneConfig({"ContourManagerConfig", "min_cont_key_cnt_"}, cm_config.min_cont_key_cnt_);
    yl.loadOneConfig({"ContourManagerConfig", "min_cont_cell_cnt_"}, cm_config.min_cont_cell_cnt_);
    yl.loadOneConfig({"ContourManagerConfig", "piv_firsts_"}, cm_config.piv_firsts_);
    yl.loadOneConfig({"ContourManagerConfig", "dist_firsts_"}, cm_config.dist_firsts_);
    yl.loadOneConfig({"ContourManagerConfig", "roi_radius_"}, cm_config.roi_radius_);
    yl.loadOneConfig({"fpath_outcome_sav"}, sav_path);
    yl.close();
  }
  TicToc init_clk;
  ContLCDEvaluator evaluator(fpath_sens_gt_pose, fpath_lidar_bins, corr_thres);
  ContourDB contour_db(db_config, 65536);
  const double init_s = init_clk.toc();  // the pose / scan lists, and the device runtime + stream pool (cc_runtime_init)
  int cnt_tp = 0, cnt_fn = 0, cnt_fp = 0, n_loops = 0;
  stp = SequentialTimeProfiler(sav_path);
  TicToc loop_clk;
  while (evaluator.loadNewScan()) {
    // stage timers as in the reference's spinner (test/batch_bin_test.cpp:131-134, 231-238)
    stp.lap();
    stp.start();
    std::shared_ptr<ContourManager> cm_tgt = evaluator.getCurrContourManager(cm_config);
    stp.record("make bev");
    const auto info = evaluator.getCurrScanInfo();
    cm_tgt->clearImage();
    std::vector<std::shared_ptr<const ContourManager>> cands;
    std::vector<double> cand_corr;
    std::vector<Eigen::Isometry2d> bev_tfs;
    contour_db.queryRangedKNN(cm_tgt, thres_lb_, thres_ub_, cands, cand_corr, bev_tfs);
    CC_CHECK(cands.size() < 2);
    const PredictionOutcome pred = cands.empty() ? evaluator.addPrediction(cm_tgt, 0.0)
                                                 : evaluator.addPrediction(cm_tgt, cand_corr[0], cands[0], bev_tfs[0]);
    cnt_tp += pred.tfpn == PredictionOutcome::TP;
    cnt_fp += pred.tfpn == PredictionOutcome::FP;
    cnt_fn += pred.tfpn == PredictionOutcome::FN;
    stp.start();
    contour_db.addScan(cm_tgt, info.ts);
    contour_db.pushAndBalance(info.seq, info.ts);
    stp.record("Update database");
    n_loops++;
  }
  const double loop_s = loop_clk.toc();
  stp.printScreen(true);
  printf("Loop wall time: %.6f s for %d scans (%.1f scans/s, file reading included)\n", loop_s, n_loops, n_loops / loop_s);
  printf("Construction time: %.6f s (evaluator + ContourDB: lists, device runtime, streams); with it %.1f scans/s\n", init_s, n_loops / (loop_s + init_s));
  printf("Accumulated tp poses: %d\nAccumulated fn poses: %d\nAccumulated fp poses: %d\n", cnt_tp, cnt_fn, cnt_fp);
  printf("TP Error mean: t:%7.4f m, r:%7.4f rad\n", evaluator.getTPMeanTrans(), evaluator.getTPMeanRot());
  printf("TP Error rmse: t:%7.4f m, r:%7.4f rad\n", evaluator.getTPRMSETrans(), evaluator.getTPRMSERot());
  evaluator.savePredictionResults(sav_path);
  return 0;
}
