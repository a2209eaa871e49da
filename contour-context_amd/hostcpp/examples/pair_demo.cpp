// The single-pair flow (BASELINE config 0; test/kitti_read_bin_test.cpp:140-300 without ROS/OpenCV output) written against
// the class mirror: two KITTI .bin scans -> two ContourManagers -> every plausible anchor pair of the two scans goes
// through CandidateManager::checkCandWithHint -> tidyUpCandidates -> fineOptimize -> score + 3-DoF pose (BEV frame and
// sensor frame).  Output: one line per hint `H level seq_src seq_tgt  ovlp_sum max_one in_ang_rng  indiv_sim orie_sim`,
// then `R n_res correlation x y theta  sens_x sens_y sens_theta`.
//   g++ -O2 -std=c++17 pair_demo.cpp -I.. -I../../../include -L../.. -lcont2_amd -Wl,-rpath,$PWD/../.. -L/opt/rocm/lib -lamdhip64
//   ./pair_demo old.bin new.bin [max_fine_opt [image_prefix]]
// With image_prefix the SAVE_MID_FILE artefacts of the reference drivers are written too: <prefix>_pair.png
// (ContourManager::saveMatchedPairImg, kitti_read_bin_test.cpp:301) and <prefix>_lv2.png (saveContourImage of level 2).
#include <cmath>

#include "cont2/contour_db.h"

SequentialTimeProfiler stp;  // the library's stage timers land here (contour_db.h: extern)

template <typename PointType>
typename pcl::PointCloud<PointType>::ConstPtr readKITTIPointCloudBin(const std::string &path) {  // tools/pointcloud_util.h:9-47
  auto out = std::make_shared<pcl::PointCloud<PointType>>();
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) {
    printf("Lidar bin file %s does not exist.\n", path.c_str());
    exit(-1);
  }
  std::vector<float> buf(1000000);
  const size_t n = fread(buf.data(), sizeof(float), buf.size(), f) / 4;
  fclose(f);
  out->reserve(n);
  for (size_t i = 0; i < n; i++) out->push_back(PointType{buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], 0.f});
  return out;
}

static std::shared_ptr<ContourManager> load(const ContourManagerConfig &cfg, const std::string &path, int id) {
  std::shared_ptr<ContourManager> cm(new ContourManager(cfg, id));
  auto cloud = readKITTIPointCloudBin<pcl::PointXYZ>(path);
  cm->makeBEV<pcl::PointXYZ>(cloud, std::to_string(id));
  cm->makeContoursRecurs();
  cm->clearImage();
  return cm;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <old.bin> <new.bin> [max_fine_opt=5]\n", argv[0]);
    return 2;
  }
  const int max_fine_opt = argc >= 4 ? atoi(argv[3]) : 5;  // kitti_read_bin_test.cpp:278
  if (argc >= 5) ContourManager::keepImages() = true;      // before the scans are ingested
  ContourManagerConfig config;
  config.lv_grads_ = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};
  auto cm_old = load(config, argv[1], 0), cm_new = load(config, argv[2], 1);
  if (argc >= 5) {
    ContourManager::saveMatchedPairImg(std::string(argv[4]) + "_pair.png", *cm_old, *cm_new);
    cm_new->saveContourImage(std::string(argv[4]) + "_lv2.png", 2);
  }

  CandidateScoreEnsemble lb, ub;  // shipped thresholds (config/batch_bin_test_config.yaml:69-87)
  lb.sim_constell.i_ovlp_sum = lb.sim_constell.i_ovlp_max_one = lb.sim_constell.i_in_ang_rng = 3;
  lb.sim_pair.i_indiv_sim = 3;
  lb.sim_pair.i_orie_sim = 4;
  lb.sim_post.correlation = 0.3f;
  lb.sim_post.area_perc = 0.03f;
  lb.sim_post.neg_est_dist = -5.01f;
  ub.sim_constell.i_ovlp_sum = ub.sim_constell.i_ovlp_max_one = ub.sim_constell.i_in_ang_rng = 6;
  ub.sim_pair.i_indiv_sim = ub.sim_pair.i_orie_sim = 6;
  ub.sim_post.correlation = 0.75f;
  ub.sim_post.area_perc = 0.15f;
  ub.sim_post.neg_est_dist = -5.0f;
  const ContourSimThresConfig cont_sim;

  CandidateManager cand_mng(cm_new, lb, ub);
  // as if the key search had returned every key pair that is not absurdly far apart (kitti_read_bin_test.cpp:229-262);
  // anchors come from the levels that carry distance bits (1..4)
  for (int ll = 1; ll <= CC_BCI_LAYERS; ll++) {
    const auto keys1 = cm_old->getLevRetrievalKey(ll), keys2 = cm_new->getLevRetrievalKey(ll);
    for (int i1 = 0; i1 < (int)keys1.size(); i1++)
      for (int i2 = 0; i2 < (int)keys2.size(); i2++) {
        if (keys1[i1].sum() == 0 || keys2[i2].sum() == 0) continue;
        KeyFloatType d2 = 0;
        for (int k = 0; k < RET_KEY_DIM; k++) d2 += (keys1[i1][k] - keys2[i2][k]) * (keys1[i1][k] - keys2[i2][k]);
        if (d2 > 1000.0f) continue;
        const CandidateScoreEnsemble s = cand_mng.checkCandWithHint(cm_old, ConstellationPair(ll, i1, i2), cont_sim);
        printf("H %d %d %d  %d %d %d  %d %d\n", ll, i1, i2, s.sim_constell.i_ovlp_sum, s.sim_constell.i_ovlp_max_one,
               s.sim_constell.i_in_ang_rng, s.sim_pair.i_indiv_sim, s.sim_pair.i_orie_sim);
      }
  }
  cand_mng.tidyUpCandidates();
  std::vector<std::shared_ptr<const ContourManager>> res_cand;
  std::vector<double> res_corr;
  std::vector<Eigen::Isometry2d> res_T;
  const int n = cand_mng.fineOptimize(max_fine_opt, res_cand, res_corr, res_T);
  if (n == 0) {
    printf("R 0 0 0 0 0  0 0 0\n");
    return 0;
  }
  const Eigen::Isometry2d T_sens = ConstellCorrelation::getEstSensTF(res_T[0], config);
  printf("R %d %.9g %.9g %.9g %.9g  %.9g %.9g %.9g\n", n, res_corr[0], res_T[0](0, 2), res_T[0](1, 2),
         std::atan2(res_T[0](1, 0), res_T[0](0, 0)), T_sens(0, 2), T_sens(1, 2), std::atan2(T_sens(1, 0), T_sens(0, 0)));
  return 0;
}
