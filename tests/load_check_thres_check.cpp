// Test program: ContLCDEvaluator::loadCheckThres of the class mirror (hostcpp/eval/evaluator.h) on a `name lower upper` file;
// prints the sixteen values it read.  usage: load_check_thres_check <file>
#include "eval/evaluator.h"
int main(int argc, char **argv) {
  CandidateScoreEnsemble lb, ub;
  ContLCDEvaluator::loadCheckThres(argv[1], lb, ub);
  printf("RES %d %d %d %d %d %.4f %.4f %.4f | %d %d %d %d %d %.4f %.4f %.4f\n", lb.sim_constell.i_ovlp_sum, lb.sim_constell.i_ovlp_max_one,
         lb.sim_constell.i_in_ang_rng, lb.sim_pair.i_indiv_sim, lb.sim_pair.i_orie_sim, lb.sim_post.correlation, lb.sim_post.area_perc,
         lb.sim_post.neg_est_dist, ub.sim_constell.i_ovlp_sum, ub.sim_constell.i_ovlp_max_one, ub.sim_constell.i_in_ang_rng, ub.sim_pair.i_indiv_sim,
         ub.sim_pair.i_orie_sim, ub.sim_post.correlation, ub.sim_post.area_perc, ub.sim_post.neg_est_dist);
}
