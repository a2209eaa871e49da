// Test program (CPU harness or GPU): the batched forms of the per-scan calls give the answers of the calls made one by one.
//   cc_scan_ingest_batch            vs cc_scan_ingest                        : descriptors, byte for byte
//   cc_db_add_scan_batch + cc_db_query_scan_batch_submit (scan k at epoch k) vs the loop query(k), add(k) : results, byte for byte
// usage: scan_batch_check <ts step> <file.bin>...      prints "ok <scans> <loops closed>" or the first difference
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cont2_amd.h"

#define CHK(x)                                                        \
  do {                                                                \
    if ((x) != CC_OK) {                                               \
      fprintf(stderr, "%s: %s\n", #x, cc_last_error());               \
      return 1;                                                       \
    }                                                                 \
  } while (0)

// everything a descriptor defines (entries behind n_stored / n_pts / n_segs are never written by the kernels)
static const char *desc_diff(const cc_scan_desc_t &x, const cc_scan_desc_t &y) {
  if (memcmp(&x, &y, offsetof(cc_scan_desc_t, bcis)) != 0) return "counts / keys";
  for (int l = 0; l < CC_NLEV; l++) {
    for (int s = 0; s < CC_NPIV; s++) {
      const cc_bci_t &p = x.bcis[l][s], &q = y.bcis[l][s];
      if (memcmp(p.dist_bin, q.dist_bin, sizeof(p.dist_bin)) != 0 || p.piv_seq != q.piv_seq || p.level != q.level || p.n_pts != q.n_pts ||
          p.n_segs != q.n_segs)
        return "bci header";
      if (memcmp(p.segs, q.segs, sizeof(uint16_t) * p.n_segs) != 0) return "bci segments";
      if (memcmp(p.pts, q.pts, sizeof(cc_relpt_t) * p.n_pts) != 0) return "bci points";
    }
    if (memcmp(x.cont[l], y.cont[l], sizeof(cc_contour_t) * (size_t)x.n_stored[l]) != 0) return "contours";
  }
  return nullptr;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const double dt = atof(argv[1]);
  const int n = argc - 2;
  cc_manager_cfg_t m;
  cc_default_manager_cfg(&m);
  cc_ctx *ctx = nullptr;
  CHK(cc_create(0, &m, 8, &ctx));
  const int64_t cap = 250000;
  std::vector<std::vector<float>> pts(n);
  for (int i = 0; i < n; i++) {
    FILE *f = fopen(argv[2 + i], "rb");
    if (!f) return 3;
    pts[i].resize(4 * cap);
    const size_t got = fread(pts[i].data(), 16, cap, f);
    fclose(f);
    pts[i].resize(4 * got);
  }
  // 1. one by one
  std::vector<cc_scan *> a(n), b(n);
  for (int i = 0; i < n; i++) {
    float *dst = cc_stage_points_slot(ctx, cap, i % 2);
    if (!dst) return 4;
    memcpy(dst, pts[i].data(), pts[i].size() * 4);
    CHK(cc_scan_ingest(ctx, dst, (int64_t)pts[i].size() / 4, 0, &a[i]));
  }
  // 2. in batches of 1..16 scans, slots walking through the ring
  int slot = 0;
  for (int i0 = 0, round = 0; i0 < n; round++) {
    static const int sizes[7] = {1, 2, 5, 8, 9, CC_SCAN_BATCH_MAX, 3};  // <= 8 scans: K1 split over eight workgroups per scan; more: one each
    const int nb = std::min(n - i0, sizes[round % 7]);
    const float *src[CC_SCAN_BATCH_MAX];
    int64_t np[CC_SCAN_BATCH_MAX];
    for (int j = 0; j < nb; j++) {
      float *dst = cc_stage_points_slot(ctx, cap, slot);
      slot = (slot + 1) % (2 * CC_SCAN_BATCH_MAX);
      if (!dst) return 4;
      memcpy(dst, pts[i0 + j].data(), pts[i0 + j].size() * 4);
      src[j] = dst;
      np[j] = (int64_t)pts[i0 + j].size() / 4;
    }
    CHK(cc_scan_ingest_batch(ctx, src, np, nb, &b[i0]));
    i0 += nb;
  }
  for (int i = 0; i < n; i++) {
    const cc_scan_desc_t *da = nullptr, *db_ = nullptr;
    CHK(cc_scan_desc(a[i], &da));
    CHK(cc_scan_desc(b[i], &db_));
    const char *why = desc_diff(*da, *db_);
    if (why) {
      printf("descriptor of scan %d differs between cc_scan_ingest and cc_scan_ingest_batch: %s\n", i, why);
      return 10;
    }
    if (!cc_scan_ready(b[i])) {
      printf("scan %d not ready after its descriptor was fetched\n", i);
      return 11;
    }
  }
  // 3. the database: sequential loop on the first set of handles, batched steps on the second
  cc_db_cfg_t dc;
  cc_default_db_cfg(&dc);
  dc.max_elapse = 10.0;
  dc.min_elapse = 6.0;
  cc_score_t lb, ub;
  cc_default_thresholds(&lb, &ub);
  cc_db *d1 = nullptr, *d2 = nullptr;
  CHK(cc_db_create(ctx, &dc, 4096, &d1));
  CHK(cc_db_create(ctx, &dc, 4096, &d2));
  CHK(cc_db_set_lanes(d2, 2));
  std::vector<cc_query_result_t> r1(n), r2(n);
  memset(r1.data(), 0, sizeof(cc_query_result_t) * n);
  memset(r2.data(), 0, sizeof(cc_query_result_t) * n);
  for (int i = 0; i < n; i++) {
    CHK(cc_db_query_scan(d1, a[i], &lb, &ub, &r1[i]));
    CHK(cc_db_add_scan(d1, a[i], dt * i, i));
  }
  for (int i0 = 0, round = 0; i0 < n; round++) {
    static const int steps[6] = {1, 4, 7, CC_SCAN_BATCH_MAX, 2, 8};
    const int nb = std::min(n - i0, steps[round % 6]);
    double ts[CC_SCAN_BATCH_MAX];
    int32_t seed[CC_SCAN_BATCH_MAX], epoch[CC_SCAN_BATCH_MAX];
    for (int j = 0; j < nb; j++) {
      ts[j] = dt * (i0 + j);
      seed[j] = i0 + j;
      epoch[j] = i0 + j;
    }
    CHK(cc_db_add_scan_batch(d2, &b[i0], nb, ts, seed));
    CHK(cc_db_query_scan_batch_submit(d2, &b[i0], nb, epoch, &lb, &ub, &r2[i0]));
    if (round % 2) CHK(cc_db_query_collect(d2, &r2[i0], nb));  // every other batch stays in flight behind the next append
    i0 += nb;
  }
  CHK(cc_db_query_wait(d2));
  int hits = 0;
  for (int i = 0; i < n; i++) {
    if (memcmp(&r1[i], &r2[i], sizeof(cc_query_result_t)) != 0) {
      printf("result of scan %d differs: sequential n_res %d gidx %d corr %.9g | batched n_res %d gidx %d corr %.9g\n", i, r1[i].n_res,
             r1[i].cand_gidx, r1[i].correlation, r2[i].n_res, r2[i].cand_gidx, r2[i].correlation);
      return 12;
    }
    hits += r1[i].n_res > 0;
  }
  // argument checks
  if (cc_scan_ingest_batch(ctx, nullptr, nullptr, 0, nullptr) == CC_OK) return 20;
  if (cc_db_add_scan_batch(d2, b.data(), CC_SCAN_BATCH_MAX + 1, nullptr, nullptr) == CC_OK) return 21;
  cc_db_destroy(d1);
  cc_db_destroy(d2);
  for (int i = 0; i < n; i++) {
    cc_scan_release(a[i]);
    cc_scan_release(b[i]);
  }
  cc_destroy(ctx);
  printf("ok %d %d\n", n, hits);
  return 0;
}
