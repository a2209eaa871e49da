// Test program (CPU harness): the evaluator mirror's read-ahead helper (hostcpp/eval/evaluator.h) must hand out the same
// descriptors whichever way a scan reaches getCurrContourManager -- fetched ahead in order, asked for twice (the helper's
// queue is dropped and the scan is read again on the spot), or with the image switch flipped between two calls.
// usage: evaluator_prefetch_check <poses.txt> <scans.txt>;  prints "OK <n>" or the first difference.
#include <cstddef>
#include <cstdio>
#include <cstring>

#include "eval/evaluator.h"

// Everything a descriptor defines: the rows and points beyond the stored counts are whatever the device slot held before.
static bool same(const ContourManager &ma, const ContourManager &mb) {
  const cc_scan_desc_t &a = ma.desc(), &b = mb.desc();
  if (std::memcmp(&a, &b, offsetof(cc_scan_desc_t, bcis)) != 0) return false;  // counts, extrema, flags, keys
  for (int l = 0; l < CC_NLEV; l++) {
    if (std::memcmp(a.cont[l], b.cont[l], sizeof(cc_contour_t) * (size_t)a.n_stored[l]) != 0) return false;
    for (int s = 0; s < CC_NPIV; s++) {
      const cc_bci_t &x = a.bcis[l][s], &y = b.bcis[l][s];
      if (std::memcmp(&x, &y, offsetof(cc_bci_t, segs)) != 0) return false;
      if (std::memcmp(x.segs, y.segs, sizeof(uint16_t) * x.n_segs) != 0) return false;
      if (std::memcmp(x.pts, y.pts, sizeof(cc_relpt_t) * x.n_pts) != 0) return false;
    }
  }
  return true;
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  ContourManagerConfig cfg;
  cfg.lv_grads_ = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};
  std::vector<std::shared_ptr<ContourManager>> first;
  {  // pass 1: plain sequential use, everything but the first scan comes from the helper
    ContLCDEvaluator ev(argv[1], argv[2], 0.5);
    while (ev.loadNewScan()) first.push_back(ev.getCurrContourManager(cfg));
  }
  int n = 0;
  {  // pass 2: every scan asked for twice, the image switch flipped on every third scan
    ContLCDEvaluator ev(argv[1], argv[2], 0.5);
    while (ev.loadNewScan()) {
      if (n % 3 == 2) ContourManager::keepImages() = !ContourManager::keepImages();
      auto a = ev.getCurrContourManager(cfg);
      auto b = ev.getCurrContourManager(cfg);
      if (!same(*a, *first[n]) || !same(*b, *first[n])) {
        printf("scan %d differs\n", n);
        return 1;
      }
      if (ContourManager::keepImages()) {
        const auto img = a->getBevImage();
        bool any = false;
        for (float v : img.data) any = any || v > -999.f;
        if (!any) {
          printf("scan %d: empty image\n", n);
          return 1;
        }
      }
      n++;
    }
  }
  if (n != (int)first.size() || n < 4) {
    printf("only %d scans\n", n);
    return 1;
  }
  printf("OK %d\n", n);
  return 0;
}
