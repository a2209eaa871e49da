// TEST INFRASTRUCTURE: runs the product's HIP kernels on the CPU through tests/emu/hip/hip_runtime.h
// so kernel logic can be checked against the oracle without a GPU.  Built by tests/emu/build.sh into
// tests/emu/libcc_emu.so; never shipped as a product path.
#include <hip/hip_runtime.h>

#include "../../contour-context_amd/csrc/k_rasterize.h"
#include "../../contour-context_amd/csrc/k_contours.h"
#include "../../contour-context_amd/csrc/cc_hostcfg.h"

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx *t_block;
thread_local WaveCtx *t_wave;
thread_local int t_lane;
}  // namespace emu

extern "C" {

int emu_ingest(const float *xyzi, const long long *offsets, int n_scans, const cc_manager_cfg_t *mcfg, int block,
               cc_scan_desc_t *desc, float *bev, float *pix_rc, int16_t *labels) {
  cc_dev_cfg cfg;
  if (cc_make_dev_cfg(mcfg, &cfg) != 0) return -1;
  const int n_cell = cfg.n_cell;
  std::vector<cc_k1_scan_out> k1(n_scans);
  std::vector<float> bev_tmp, pix_tmp;
  if (!bev) {
    bev_tmp.resize((size_t)n_scans * n_cell);
    bev = bev_tmp.data();
  }
  if (!pix_rc) {
    pix_tmp.resize((size_t)n_scans * n_cell * 2);
    pix_rc = pix_tmp.data();
  }
  for (size_t i = 0; i < (size_t)n_scans * n_cell * 2; i++) pix_rc[i] = -1.f;
  size_t lds1 = (((size_t)n_cell * 4 + 15) & ~(size_t)15) + (size_t)((n_cell + 2) / 3) * 8 + 64;
  emu::launch(cc_k_rasterize, n_scans, block, lds1, cfg, (const float4 *)xyzi, offsets, bev, (float2 *)pix_rc, k1.data());
  std::vector<cc_k2_scratch> scr(n_scans);
  size_t lds2 = (((size_t)n_cell * 4 + 15) & ~(size_t)15) + CC_K2_R_BYTES;
  emu::launch(cc_k_contours, n_scans, block, lds2, cfg, (const float *)bev, (const float2 *)pix_rc,
              (const cc_k1_scan_out *)k1.data(), scr.data(), desc, labels);
  return 0;
}

// std::sort replica on raw keys (descending on the high 16 bits like the contour size sort)
void emu_sort_desc(unsigned *arr, int n) {
  ccsort::std_sort(arr, n, [](unsigned x, unsigned y) { return (x >> 16) > (y >> 16); });
}
struct fkey { float k; int idx; };
void emu_sort_asc_f(fkey *arr, int n) {
  ccsort::std_sort(arr, n, [](const fkey &x, const fkey &y) { return x.k < y.k; });
}
void emu_eigen2f(const float m[3], float ev[2], float vec[4]) { cc_eigen2f(m[0], m[1], m[2], ev, vec); }
}
