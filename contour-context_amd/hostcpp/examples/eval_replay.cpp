// Replays recorded predictions through ContLCDEvaluator: judges them against the ground truth and writes the outcome
// file.  No device work: a ContourManager only lends its id and config here.
//   eval_replay <pose file> <scan list> <sim threshold> <predictions: `tgt src|-1 corr tx ty theta`> <outcome out>
#include "eval/evaluator.h"

SequentialTimeProfiler stp;  // the library's stage timers land here (contour_db.h: extern)

int main(int argc, char **argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s poses.txt scans.txt thres predictions.txt outcome.txt\n", argv[0]);
    return 2;
  }
  ContLCDEvaluator ev(argv[1], argv[2], atof(argv[3]));
  ContourManagerConfig cfg;
  cfg.lv_grads_ = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};  // config/batch_bin_test_config.yaml
  std::ifstream pred(argv[4]);
  std::string line;
  while (std::getline(pred, line)) {
    std::istringstream iss(line);
    int tgt, src;
    double corr, tx, ty, th;
    if (!(iss >> tgt >> src >> corr >> tx >> ty >> th)) continue;
    std::shared_ptr<const ContourManager> q(new ContourManager(cfg, tgt)), c;
    Eigen::Isometry2d T;
    T.rotate(th);
    T.pretranslate(tx, ty);
    if (src >= 0) c.reset(new ContourManager(cfg, src));
    ev.addPrediction(q, corr, c, T);
  }
  ev.savePredictionResults(argv[5]);
  printf("TP mean trans %.6f rot %.6f, rmse trans %.6f rot %.6f\n", ev.getTPMeanTrans(), ev.getTPMeanRot(), ev.getTPRMSETrans(),
         ev.getTPRMSERot());
  return 0;
}
