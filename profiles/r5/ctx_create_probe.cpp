// Tuning aid: where do the ~60 ms of the drop-in loop's first scan go?   hipcc -O2 ctx_create_probe.cpp -I include -L pkg -lcont2_amd
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include "cont2_amd.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double t = now();
  auto lap = [&](const char *what) {
    const double n = now();
    printf("%-60s %8.2f ms\n", what, 1e3 * (n - t));
    t = n;
  };
  hipSetDevice(0);
  hipFree(nullptr);
  lap("first HIP call (hipSetDevice + hipFree(0))");
  cc_manager_cfg_t m;
  cc_default_manager_cfg(&m);
  cc_ctx *ctx = nullptr;
  cc_create(0, &m, 8, &ctx);
  lap("cc_create(max_batch_scans = 8)");
  float *p = cc_stage_points(ctx, 250000);
  lap("first cc_stage_points (streams, channel scratch, one pinned slot)");
  for (int i = 0; i < 120000; i++) {
    p[4 * i] = (float)((i * 7919) % 8000) / 100.f - 40.f;
    p[4 * i + 1] = (float)((i * 104729) % 8000) / 100.f - 40.f;
    p[4 * i + 2] = (float)(i % 50) / 10.f;
    p[4 * i + 3] = 0;
  }
  lap("fill 120 000 points");
  cc_scan *sc = nullptr;
  cc_scan_ingest(ctx, p, 120000, 0, &sc);
  lap("first cc_scan_ingest (slot block, first launches: code object load)");
  const cc_scan_desc_t *d = nullptr;
  cc_scan_desc(sc, &d);
  lap("cc_scan_desc (wait + copy)");
  cc_db_cfg_t dc;
  cc_default_db_cfg(&dc);
  cc_db *db = nullptr;
  cc_db_create(ctx, &dc, 65536, &db);
  lap("cc_db_create(capacity 65 536)");
  cc_score_t lb, ub;
  int *li = (int *)&lb, *ui = (int *)&ub;
  lb.i_ovlp_sum = lb.i_ovlp_max_one = lb.i_in_ang_rng = lb.i_indiv_sim = 3; lb.i_orie_sim = 4; lb.correlation = 0.3f; lb.area_perc = 0.03f; lb.neg_est_dist = -5.01f;
  ub.i_ovlp_sum = ub.i_ovlp_max_one = ub.i_in_ang_rng = ub.i_indiv_sim = ub.i_orie_sim = 6; ub.correlation = 0.75f; ub.area_perc = 0.15f; ub.neg_est_dist = -5.0f;
  (void)li; (void)ui;
  cc_query_result_t r;
  cc_db_query_scan(db, sc, &lb, &ub, &r);
  lap("first cc_db_query_scan (lane scratch, first launches)");
  cc_db_add_scan(db, sc, 0.0, 0);
  lap("first cc_db_add_scan");
  cc_db_query_scan(db, sc, &lb, &ub, &r);
  lap("second cc_db_query_scan");
  cc_db_add_scan(db, sc, 0.1, 1);
  lap("second cc_db_add_scan");
  return 0;
}
