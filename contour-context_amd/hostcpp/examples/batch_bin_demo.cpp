// A batch_bin_test-shaped driver (test/batch_bin_test.cpp:105-247) written against the class mirror:
// per scan  ContourManager ctor -> makeBEV -> makeContoursRecurs -> clearImage -> queryRangedKNN -> addScan ->
// pushAndBalance.  Input: a list file `ts seq path` of KITTI .bin files (the reference's lidar-bins list format,
// scripts/gen_batch_bin_configs.py:101-159).  Output: one line per scan `seq cand_seq correlation x y theta`.
//   g++ -O2 -std=c++17 batch_bin_demo.cpp -I.. -L../.. -lcont2_amd -Wl,-rpath,$PWD/../.. -o batch_bin_demo
#include <fstream>
#include <iostream>
#include <sstream>

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

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <ts-lidar_bins list> [min_elapse max_elapse [dump_seq dump_path]]\n", argv[0]);
    return 2;
  }
  ContourManagerConfig cm_config;
  cm_config.lv_grads_ = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};
  ContourDBConfig db_config;
  db_config.q_levels_ = {1, 2, 3};
  if (argc >= 4) {
    db_config.tb_cfg_.min_elapse_ = atof(argv[2]);
    db_config.tb_cfg_.max_elapse_ = atof(argv[3]);
  }
  CandidateScoreEnsemble lb, ub;
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
  ContourDB db(db_config, 8192);
  std::ifstream lst(argv[1]);
  std::string line;
  while (std::getline(lst, line)) {
    std::istringstream iss(line);
    double ts;
    int seq;
    std::string path;
    if (!(iss >> ts >> seq >> path)) continue;
    std::shared_ptr<ContourManager> cm(new ContourManager(cm_config, seq));
    auto cloud = readKITTIPointCloudBin<pcl::PointXYZ>(path);
    cm->makeBEV<pcl::PointXYZ>(cloud, std::to_string(seq));
    cm->makeContoursRecurs();
    cm->clearImage();
    if (argc >= 6 && seq == atoi(argv[4])) cm->saveContours(argv[5]);  // contour dump of one scan (parity / plotting)
    std::vector<std::shared_ptr<const ContourManager>> cands;
    std::vector<double> corr;
    std::vector<Eigen::Isometry2d> tfs;
    db.queryRangedKNN(cm, lb, ub, cands, corr, tfs);
    CC_CHECK(cands.size() < 2);
    if (cands.empty())
      printf("%d -1 0 0 0 0\n", seq);
    else
      printf("%d %d %.9g %.9g %.9g %.9g\n", seq, cands[0]->getIntID(), corr[0], tfs[0](0, 2), tfs[0](1, 2),
             std::atan2(tfs[0](1, 0), tfs[0](0, 0)));
    db.addScan(cm, ts);
    db.pushAndBalance(seq, ts);
  }
  return 0;
}
