// libcont2_amd.so -- C-ABI of the MI355X-native contour-context hot path (include/cont2_amd.h).
// Host code: device memory ownership, launches, the LayerDB bookkeeping timeline.  Kernels live in
// the k_*.h headers next to this file.  Built for gfx950 only:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC cont2_amd.hip -o libcont2_amd.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/cont2_amd.h"
#include "cc_hostcfg.h"
#include "k_rasterize.h"
#include "k_contours.h"
#include "k_contours_list.h"
#include "k_knn.h"
#include "k_check.h"
#include "k_merge.h"
#include "k_gmm.h"
#include "cc_hostdb.h"

#ifndef CC_INGEST_BLOCK
#define CC_INGEST_BLOCK 1024  // threads per workgroup of the per-scan ingest kernels
#endif

static thread_local std::string g_err;
static int set_err(int code, const char *what, hipError_t e = hipSuccess) {
  g_err = what;
  if (e != hipSuccess) {
    g_err += ": ";
    g_err += hipGetErrorString(e);
  }
  return code;
}
#define HIPCHK(call)                                          \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return set_err(CC_EHIP, #call, e_); \
  } while (0)

struct cc_ctx {
  int device = 0;
  cc_manager_cfg_t mcfg;
  cc_dev_cfg dcfg;
  int max_batch = 0;
  // What ONE ingest launch chain works in.  Calls on the same set are ordered (ev_last: the next call waits, on the device,
  // for the previous one's last kernel when it comes in on another stream); different sets may be in flight together.  The
  // context has the set of the batched calls (`main`, max_batch scans) and, for the per-scan loop, CC_NCHAN one-scan sets
  // (`chan[i]`, made at first use): consecutive scans of the loop are ingested on alternating channels, so the ~0.25 ms one
  // scan's K1 + K2 take overlap with the next scan's instead of queueing behind them.
  struct Scratch {
    int cap = 0;  // scans
    float *d_bev = nullptr;
    float2 *d_pix = nullptr;
    cc_k1_scan_out *d_k1 = nullptr;
    cc_k1_part k1_part;  // scratch of the split rasterisation (calls of <= CC_K1_SPLIT_MAX_SCANS scans), allocated at first use
    cc_k2_scratch *d_scr = nullptr;
    long long *d_offsets = nullptr;
    // the slow path of K2 (scans with more than CC_MAXC components on a level): queue filled by the fast launch, a few
    // workgroups with CC_NC_BIG-sized tables in global memory
    int n_bigslots = 0;
    cc_k2_big_queue *d_bigq = nullptr;
    cc_k2_big_queue *d_midq = nullptr;  // scans the list kernel hands to the original body (cc_k_contours_mid)
    int *h_mid_seen = nullptr;          // pinned: the queue length the last cc_k_contours_mid launch found (-1: none has run yet)
    cc_k1_list_out list;                // K1 -> K2: the scans' active cells as raster-ordered lists (k_rasterize.h)
    cc_k2_big_slot *d_bigslots = nullptr;
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
  };
  static const int N_BIG_SLOTS = 8;
  Scratch main;
  // pinned staging ring for the per-chunk point offsets: a slot is reused only after the copy that read it has finished
  static const int NSLOT = 4;
  long long *h_off[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t off_ev[NSLOT] = {nullptr, nullptr, nullptr, nullptr};
  bool off_busy[NSLOT] = {false, false, false, false};
  int off_next = 0;
  long long *d_phase_clk = nullptr;  // tuning aid: per-scan phase timestamps of cc_k_contours (CC_K2_PHASES=1)
  // the per-scan loop (cc_scan_*): own stream, pinned + device point staging, a pool of device descriptor slots
  hipStream_t s_loop = nullptr;       // per-scan loop: descriptor fetches, cc_db_query_scan / cc_db_add_scan
  // per-scan loop: cc_scan_ingest (copy of the points, K1, K2) goes to the next of CC_NCHAN channels -- own stream, device point
  // buffer, one-scan scratch set; a scan's `ready` event is recorded on its channel's stream
  static const int NCHAN = 2;
  struct Channel {
    hipStream_t s = nullptr;
    float *d_pts = nullptr;
    int64_t d_pts_cap = 0;             // points d_pts holds (one staging buffer's worth; CC_SCAN_BATCH_MAX of them once a batch came by)
    cc_scan_desc_t *d_desc_tmp = nullptr;  // [CC_SCAN_BATCH_MAX] where a batch's descriptors are written before they go to their slots
    float *d_bev_copy = nullptr;  // the max-height image of a scan that asked for it (want_bev), until its copy to the host has passed
    Scratch scr;
  };
  Channel chan[NCHAN];
  int chan_next = 0;
  std::mutex slot_mu;                 // slot_free: cc_scan_ingest may run on a helper thread next to cc_scan_offload / cc_scan_release
  // per-scan loop: two pinned staging buffers (the caller may fill the second one -- e.g. read the next scan's file from
  // another thread -- while the first one's scan is in flight), one device point buffer (the stream orders its reuse)
  // Slots 0 and 1 are the caller's to name (cc_stage_points_slot), slot 2 is cc_stage_points' own -- a thread that stages
  // without naming a slot (ContourManager::makeBEV) never gets a buffer a read-ahead helper writes.
  // Round 5: 2 * CC_SCAN_BATCH_MAX caller slots (a read-ahead thread fills one batch of files while the batch before it is on
  // its way: cc_scan_ingest_batch), each allocated when it is first asked for; the last slot is cc_stage_points' own.
  static const int NPTS = 2 * CC_SCAN_BATCH_MAX + 1;
  static const int OWN_SLOT = NPTS - 1;
  float *h_pts[NPTS] = {};
  hipEvent_t pts_ev[NPTS] = {};  // recorded behind a slot's H2D copy: the slot may be rewritten once it has passed
  bool pts_busy[NPTS] = {};
  // Ingest state (the staging slots, d_pts, the offsets ring, the K1/K2 scratch, ev_last) is shared by every call of the
  // context: ing_mu is held inside cc_ingest_batch / cc_stage_points* / cc_scan_ingest.  A slot handed out by
  // cc_stage_points* belongs to the calling thread until that thread's cc_scan_ingest has queued its copy (or the thread
  // stages the slot again); another thread asking for it WAITS (pts_cv) instead of being handed memory that is being filled.
  std::recursive_mutex ing_mu;
  std::condition_variable_any pts_cv;
  bool pts_handed[NPTS] = {};
  std::thread::id pts_owner[NPTS];
  int64_t pts_cap = 0;  // points
  std::vector<cc_scan_desc_t *> slot_free, slot_blocks;
  std::vector<int> slot_block_n;  // slots per block
  size_t lds1 = 0, lds2 = 0;
  int k1_div = 0;  // CC_K1_DIV=1: keep the IEEE divisions even for power-of-two resolutions (A/B aid)
  int k1_nosplit = 0;  // CC_K1_NOSPLIT=1: one workgroup per scan also for calls of a few scans (A/B aid)
  int k1_wgs = 0;      // CC_K1_WGS: workgroups of a many-scan K1 launch, each taking scans b, b + grid, ... (0 = one per scan, the default:
                       // 256 persistent workgroups make K1 0.55 -> 0.49 ms inside the pipelined step and K2 1.15 -> 1.22, the step 1.79 -> 1.82)
  int k1_dense = 0;    // CC_K1_DENSE=1: K1 writes the dense image / positions of every scan (A/B aid; round 5's behaviour)
  // optional per-kernel timing (cc_profile_*)
  bool prof = false;
  std::vector<hipEvent_t> ev;  // triplets (before K1, between, after K2)
  size_t ev_used = 0;
  double ms_acc[2] = {0, 0};
  int launches = 0;
};

static int prof_flush(cc_ctx *c) {
  for (size_t i = 0; i + 3 <= c->ev_used; i += 3) {
    float a = 0, b = 0;
    if (hipEventSynchronize(c->ev[i + 2]) != hipSuccess) return CC_EHIP;
    hipEventElapsedTime(&a, c->ev[i], c->ev[i + 1]);
    hipEventElapsedTime(&b, c->ev[i + 1], c->ev[i + 2]);
    c->ms_acc[0] += a;
    c->ms_acc[1] += b;
    c->launches++;
  }
  c->ev_used = 0;
  return CC_OK;
}


static int scratch_alloc(cc_ctx *c, cc_ctx::Scratch &S, int cap, int n_bigslots) {
  const size_t nc = (size_t)c->dcfg.n_cell;
  S.cap = cap;
  HIPCHK(hipMalloc(&S.d_bev, sizeof(float) * nc * cap));
  HIPCHK(hipMalloc(&S.d_pix, sizeof(float2) * nc * cap));
  HIPCHK(hipMalloc(&S.d_k1, sizeof(cc_k1_scan_out) * cap));
  HIPCHK(hipMalloc(&S.d_scr, sizeof(cc_k2_scratch) * cap));
  HIPCHK(hipMalloc(&S.d_offsets, sizeof(long long) * (cap + 1)));
  HIPCHK(hipMalloc(&S.d_bigq, sizeof(cc_k2_big_queue) + sizeof(int) * (size_t)cap));
  HIPCHK(hipMemset(S.d_bigq, 0, sizeof(cc_k2_big_queue)));  // the slow launch leaves it empty again
  HIPCHK(hipMalloc(&S.d_midq, sizeof(cc_k2_big_queue) + sizeof(int) * (size_t)cap));
  HIPCHK(hipMalloc(&S.list.hdr, sizeof(int4) * (size_t)cap));
  HIPCHK(hipMalloc(&S.list.rc, sizeof(uint16_t) * (size_t)CC_LIST_CAP * cap));
  HIPCHK(hipMalloc(&S.list.lev, (size_t)CC_LIST_CAP * cap));
  HIPCHK(hipMalloc(&S.list.h, sizeof(float) * (size_t)CC_LIST_CAP * cap));
  HIPCHK(hipMalloc(&S.list.pix, sizeof(float2) * (size_t)CC_LIST_CAP * cap));
  HIPCHK(hipMemset(S.d_midq, 0, sizeof(cc_k2_big_queue)));
  HIPCHK(hipHostMalloc((void **)&S.h_mid_seen, sizeof(int), hipHostMallocDefault));
  *S.h_mid_seen = -1;
  S.n_bigslots = n_bigslots < cap ? n_bigslots : cap;
  HIPCHK(hipMalloc(&S.d_bigslots, sizeof(cc_k2_big_slot) * S.n_bigslots));
  HIPCHK(hipEventCreateWithFlags(&S.ev_last, hipEventDisableTiming));
  return CC_OK;
}
static void scratch_free(cc_ctx::Scratch &S) {
  hipFree(S.d_bev);
  hipFree(S.d_pix);
  hipFree(S.d_k1);
  hipFree(S.k1_part.key);
  hipFree(S.k1_part.idx);
  hipFree(S.k1_part.red);
  hipFree(S.d_scr);
  hipFree(S.d_offsets);
  hipFree(S.d_bigq);
  hipFree(S.d_midq);
  if (S.h_mid_seen) hipHostFree(S.h_mid_seen);
  hipFree(S.list.hdr);
  hipFree(S.list.rc);
  hipFree(S.list.lev);
  hipFree(S.list.h);
  hipFree(S.list.pix);
  hipFree(S.d_bigslots);
  if (S.ev_last) hipEventDestroy(S.ev_last);
  S = cc_ctx::Scratch();
}

extern "C" {

const char *cc_last_error(void) { return g_err.c_str(); }
int cc_version(void) { return 100; }

void cc_default_manager_cfg(cc_manager_cfg_t *c) {
  const float g[CC_NLEV] = {1.5f, 2.f, 2.5f, 3.f, 3.5f, 4.f};
  for (int i = 0; i < CC_NLEV; i++) c->lv_grads[i] = g[i];
  c->reso_row = c->reso_col = 1.0f;
  c->n_row = c->n_col = 150;
  c->lidar_height = 2.0f;
  c->blind_sq = 9.0f;
  c->min_cont_key_cnt = 9;
  c->min_cont_cell_cnt = 3;
  c->piv_firsts = 6;
  c->dist_firsts = 10;
  c->roi_radius = 10.0f;
  c->min_cell_cov = 4;
  c->point_sigma = 1.0f;
  c->com_bias_thres = 0.5f;
}
void cc_default_db_cfg(cc_db_cfg_t *d) {
  d->nnk = 50;
  d->max_fine_opt = 10;
  d->n_q_levels = 3;
  d->q_levels[0] = 1;
  d->q_levels[1] = 2;
  d->q_levels[2] = 3;
  d->cont_sim.ta_cell_cnt = 6.0f;
  d->cont_sim.tp_cell_cnt = 0.2f;
  d->cont_sim.tp_eigval = 0.2f;
  d->cont_sim.ta_h_bar = 0.3f;
  d->cont_sim.ta_rcom = 0.4f;
  d->cont_sim.tp_rcom = 0.25f;
  d->max_elapse = 25.0;
  d->min_elapse = 15.0;
}
void cc_default_thresholds(cc_score_t *lb, cc_score_t *ub) {
  lb->i_ovlp_sum = lb->i_ovlp_max_one = lb->i_in_ang_rng = lb->i_indiv_sim = 3;
  lb->i_orie_sim = 4;
  lb->correlation = 0.3f;
  lb->area_perc = 0.03f;
  lb->neg_est_dist = -5.01f;
  ub->i_ovlp_sum = ub->i_ovlp_max_one = ub->i_in_ang_rng = ub->i_indiv_sim = ub->i_orie_sim = 6;
  ub->correlation = 0.75f;
  ub->area_perc = 0.15f;
  ub->neg_est_dist = -5.0f;
}

// ---- start of the device runtime, and a per-device pool of streams ----
// Measured on MI355X / ROCm 7.2: the first HIP call of a process takes ~54 ms, loading the code object ~5-20 ms, and
// hipStreamCreateWithFlags 16 / 8.5 / 8.5 / 8.5 ms for the first four streams of a process and 3.4 ms for every further one
// (profiles/r5/stream_probe.cpp) -- a per-scan driver's context + database use 4-7 streams.  cc_runtime_init pays all of that
// in one call a host can make when it starts (the class mirror: the ContourDB / evaluator constructors); contexts and
// databases take their streams from the pool and give them back when they are destroyed.
#define CC_RT_MAX_DEV 64
static std::mutex g_rt_mu;
static std::vector<hipStream_t> g_stream_pool[CC_RT_MAX_DEV];
static hipError_t stream_take(int device, hipStream_t *out) {
  if (device >= 0 && device < CC_RT_MAX_DEV) {
    std::lock_guard<std::mutex> lk(g_rt_mu);
    if (!g_stream_pool[device].empty()) {
      *out = g_stream_pool[device].back();
      g_stream_pool[device].pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
// A query lane's stream.  CC_QUERY_CUS = "a/b" (a of every b CUs of each XCD) or "xa/b" (a of every b XCDs) keeps the
// lanes' kernels off the other CUs (hipExtStreamCreateWithCUMask; bit i of the mask is CU i / 8 of XCD i % 8: the driver
// deals the bits out to the XCDs in turn): in a pipelined loop the ingest stream is the critical path and its big
// workgroups (79-158 KB of LDS) wait for CUs that many small query workgroups keep occupied.  Unset: the pool's stream.
static std::vector<hipStream_t> g_masked_streams;  // (never pooled: their mask is part of them)
static hipError_t lane_stream_take(int device, hipStream_t *out) {
  const char *e = getenv("CC_QUERY_CUS");
  if (e && *e && *e != '-') {
    const bool by_xcd = e[0] == 'x';
    int a = 0, b = 0;
    if (sscanf(by_xcd ? e + 1 : e, "%d/%d", &a, &b) == 2 && a > 0 && b >= a) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) {
        const int n_cu = pr.multiProcessorCount, n_xcd = 8;
        std::vector<uint32_t> words((size_t)(n_cu + 31) / 32, 0u);
        for (int i = 0; i < n_cu; i++) {
          const int unit = by_xcd ? i % n_xcd : i / n_xcd;
          if (unit % b < a) words[(size_t)i / 32] |= 1u << (i % 32);
        }
        const hipError_t r = hipExtStreamCreateWithCUMask(out, (uint32_t)words.size(), words.data());
        if (r == hipSuccess) {
          std::lock_guard<std::mutex> lk(g_rt_mu);
          g_masked_streams.push_back(*out);
        }
        return r;
      }
    }
  }
  return stream_take(device, out);
}
static void stream_give(int device, hipStream_t s) {
  hipStreamSynchronize(s);
  if (device >= 0 && device < CC_RT_MAX_DEV) {
    std::lock_guard<std::mutex> lk(g_rt_mu);
    for (size_t i = 0; i < g_masked_streams.size(); i++)
      if (g_masked_streams[i] == s) {
        g_masked_streams.erase(g_masked_streams.begin() + (long)i);
        hipStreamDestroy(s);
        return;
      }
    if (g_stream_pool[device].size() < 32) {
      g_stream_pool[device].push_back(s);
      return;
    }
  }
  hipStreamDestroy(s);
}
int cc_runtime_init(int device, int n_streams) {
  if (device < 0 || n_streams < 0 || n_streams > 32) return set_err(CC_EINVAL, "cc_runtime_init: bad argument (0..32 streams)");
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipFree(nullptr));
  hipFuncAttributes fa;
  HIPCHK(hipFuncGetAttributes(&fa, (const void *)cc_k_contours));  // loads the code object
  for (;;) {
    {
      std::lock_guard<std::mutex> lk(g_rt_mu);
      if (device >= CC_RT_MAX_DEV || (int)g_stream_pool[device].size() >= n_streams) break;
    }
    hipStream_t s = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::lock_guard<std::mutex> lk(g_rt_mu);
    g_stream_pool[device].push_back(s);
  }
  return CC_OK;
}

int cc_create(int device, const cc_manager_cfg_t *cfg, int max_batch_scans, cc_ctx **out) {
  if (!cfg || !out || max_batch_scans < 1) return set_err(CC_EINVAL, "cc_create: bad argument");
  cc_dev_cfg dc;
  if (cc_make_dev_cfg(cfg, &dc) != 0)
    return set_err(CC_EINVAL, "cc_create: unsupported ContourManagerConfig (need even n_row/n_col <= 150x150, 6 increasing lv_grads_)");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0 || device < 0 || device >= ndev) return set_err(CC_EHIP, "cc_create: no such HIP device (this library has no CPU path)");
  HIPCHK(hipSetDevice(device));
  cc_ctx *c = new cc_ctx();
#define CREATE_CHK(call)                      \
  do {                                        \
    hipError_t e_ = (call);                   \
    if (e_ != hipSuccess) {                   \
      cc_destroy(c);                          \
      return set_err(CC_EHIP, #call, e_);     \
    }                                         \
  } while (0)
  c->device = device;
  c->mcfg = *cfg;
  c->dcfg = dc;
  c->max_batch = max_batch_scans;
  const size_t nc = (size_t)dc.n_cell;
  if (scratch_alloc(c, c->main, max_batch_scans, cc_ctx::N_BIG_SLOTS) != CC_OK) {
    cc_destroy(c);
    return CC_EHIP;  // (the message is set)
  }
  for (int i = 0; i < cc_ctx::NSLOT; i++) {
    CREATE_CHK(hipHostMalloc((void **)&c->h_off[i], sizeof(long long) * ((max_batch_scans > CC_SCAN_BATCH_MAX ? max_batch_scans : CC_SCAN_BATCH_MAX) + 1), hipHostMallocDefault));
    CREATE_CHK(hipEventCreateWithFlags(&c->off_ev[i], hipEventDisableTiming));
  }
  if (getenv("CC_K2_PHASES"))  // (a channel launch brings up to CC_SCAN_BATCH_MAX scans whatever max_batch_scans is)
    CREATE_CHK(hipMalloc(&c->d_phase_clk, sizeof(long long) * CC_K2_NCLK * (size_t)(max_batch_scans > CC_SCAN_BATCH_MAX ? max_batch_scans : CC_SCAN_BATCH_MAX)));
  c->lds1 = ((nc * 4 + 15) & ~(size_t)15) + ((nc + 2) / 3) * 8 + 64 + ((CC_K1_EMIT_LDS_BYTES + 15) & ~15);
  c->lds2 = CC_K2_LDS_BYTES(nc);
  {
    const char *e = getenv("CC_K2_LDS_PAD");  // tuning aid, read once: extra bytes asked for (above 80 KB one scan per CU instead of two)
    if (e && atoi(e) > 0 && atoi(e) <= 65536) c->lds2 += (size_t)atoi(e);
  }
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_rasterize<CC_K1_U_DEFAULT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds1));
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_rasterize<CC_K1_U_DEFAULT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds1));
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_rasterize<CC_K1_U_DEFAULT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds1));
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_rasterize<CC_K1_U_DEFAULT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds1));
  {
    const char *e = getenv("CC_K1_DIV");  // tuning aid, read once
    c->k1_div = (e && atoi(e) == 1) ? 1 : 0;
    const char *e2 = getenv("CC_K1_NOSPLIT");
    c->k1_nosplit = (e2 && atoi(e2) == 1) ? 1 : 0;
    const char *e4 = getenv("CC_K1_WGS");
    c->k1_wgs = (e4 && atoi(e4) > 0) ? atoi(e4) : 0;
    const char *e3 = getenv("CC_K1_DENSE");
    c->k1_dense = (e3 && atoi(e3) == 1) ? 1 : 0;
  }
  if (nc > (size_t)CC_MAX_CELLS) {
    cc_destroy(c);
    return set_err(CC_EINVAL, "cc_create: grid larger than 150 x 150 cells");
  }
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_contours, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC_K2L_LDS_BYTES));
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_contours_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds2));
  CREATE_CHK(hipFuncSetAttribute((const void *)cc_k_contours_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds2));
#undef CREATE_CHK
  *out = c;
  return CC_OK;
}

int cc_profile_enable(cc_ctx *c, int on) {
  if (!c) return set_err(CC_EINVAL, "cc_profile_enable: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (on && c->ev.empty()) {
    c->ev.resize(3 * 64);
    for (auto &e : c->ev) HIPCHK(hipEventCreate(&e));
  }
  c->prof = on != 0;
  return CC_OK;
}
int cc_profile_read(cc_ctx *c, double ms_out[2], int *n_launches) {
  if (!c || !ms_out) return set_err(CC_EINVAL, "cc_profile_read: bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (prof_flush(c) != CC_OK) return set_err(CC_EHIP, "cc_profile_read: event sync failed");
  if (c->d_phase_clk) {  // tuning aid: mean phase durations of the last launch, in microseconds (100 MHz wall clock)
    std::vector<long long> h(CC_K2_NCLK * (size_t)c->max_batch);
    HIPCHK(hipMemcpy(h.data(), c->d_phase_clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    const int n = c->max_batch < 256 ? c->max_batch : 256;
    double ph[9] = {0}, sub[8] = {0}, lv[6] = {0}, sa[4] = {0};
    for (int i = 0; i < n; i++) {
      const long long *p = &h[(size_t)i * CC_K2_NCLK];
      ph[0] += p[1] * 0.01;
      ph[1] += p[2] * 0.01;
      ph[2] += p[3] * 0.01;
      ph[3] += (p[5] - p[4]) * 0.01;
      ph[4] += (p[6] - p[5]) * 0.01;
      ph[5] += (p[7] - p[6]) * 0.01;
      ph[6] += (p[8] - p[7]) * 0.01;
      ph[7] += (p[8] - p[0]) * 0.01;
      sub[0] += (p[9] - p[0]) * 0.01;    // fill + active list
      sub[1] += (p[11] - p[10]) * 0.01;  // list starts + member lists
      sub[2] += (p[12] - p[11]) * 0.01;  // lane walk
      sub[3] += (p[4] - p[12]) * 0.01;   // eight-lane walk
      sub[4] += p[14] * 0.01;            // keys: RoI lists
      sub[5] += p[15] * 0.01;            // keys: division sums
      for (int j = 0; j < 6; j++) lv[j] += p[16 + j] * 0.01;
      sa[0] += (p[22] - p[0]) * 0.01;
      sa[1] += (p[23] - p[22]) * 0.01;
      sa[2] += (p[24] - p[23]) * 0.01;
      sa[3] += (p[9] - p[24]) * 0.01;
    }
    fprintf(stderr, "[cc_k_contours phases, mean us over %d scans] ccl %.1f  enum+bbox %.1f  walk %.1f  order+sort %.1f  emit %.1f  keys %.1f  bci %.1f  | total %.1f\n",
            n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n, ph[6] / n, ph[7] / n);
    fprintf(stderr, "[cc_k_contours sub-phases] fill+list %.1f | walk: member lists %.1f  lane walk %.1f  eight-lane walk %.1f | keys: RoI lists %.1f  division sums %.1f\n",
            sub[0] / n, sub[1] / n, sub[2] / n, sub[3] / n, sub[4] / n, sub[5] / n);
    {
      cc_k2_big_queue hq;
      HIPCHK(hipMemcpy(&hq, c->main.d_midq, sizeof(hq), hipMemcpyDeviceToHost));
      fprintf(stderr, "[cc_k_contours] scans handed to the mid path so far: %d (configuration %d, cells / slots %d, components %d, list padding %d)\n", hq.total, hq.why[0], hq.why[1], hq.why[2], hq.why[3]);
    }
    fprintf(stderr, "[cc_k_contours list stage A] levels %.1f  chunk ballots %.1f  prefix %.1f  entries + run labels %.1f\n", sa[0] / n, sa[1] / n, sa[2] / n, sa[3] / n);
    fprintf(stderr, "[cc_k_contours level loop, summed over the levels] unions %.1f  flatten+count %.1f  kept roots %.1f  rank %.1f  bbox/area %.1f  records %.1f\n",
            lv[0] / n, lv[1] / n, lv[2] / n, lv[3] / n, lv[4] / n, lv[5] / n);
  }
  ms_out[0] = c->ms_acc[0];
  ms_out[1] = c->ms_acc[1];
  if (n_launches) *n_launches = c->launches;
  c->ms_acc[0] = c->ms_acc[1] = 0;
  c->launches = 0;
  return CC_OK;
}

int cc_destroy(cc_ctx *c) {
#ifdef CC_TUNE_K1_CLK
  {
    unsigned long long h[8] = {0};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(cc_k1_clk), sizeof h) == hipSuccess)
      fprintf(stderr, "[cc_k_rasterize clocks, ticks of 10 ns summed over workgroups (thread 0)] init %llu | A %llu | barrier %llu | B %llu | barrier %llu | emit %llu\n", h[0], h[1], h[2],
              h[3], h[4], h[5]);
  }
#endif
  if (!c) return CC_OK;
  hipSetDevice(c->device);
  for (auto &e : c->ev) hipEventDestroy(e);
  for (auto &ch : c->chan) {
    if (ch.s) {
      stream_give(c->device, ch.s);
    }
    hipFree(ch.d_pts);
    hipFree(ch.d_desc_tmp);
    hipFree(ch.d_bev_copy);
    scratch_free(ch.scr);
  }
  scratch_free(c->main);
  for (int i = 0; i < cc_ctx::NSLOT; i++) {
    if (c->h_off[i]) hipHostFree(c->h_off[i]);
    if (c->off_ev[i]) hipEventDestroy(c->off_ev[i]);
  }
  hipFree(c->d_phase_clk);
  if (c->s_loop) {
    stream_give(c->device, c->s_loop);
  }
  for (int i = 0; i < cc_ctx::NPTS; i++) {
    if (c->h_pts[i]) hipHostFree(c->h_pts[i]);
    if (c->pts_ev[i]) hipEventDestroy(c->pts_ev[i]);
  }
  for (auto *b : c->slot_blocks) hipFree(b);
  delete c;
  return CC_OK;
}

__global__ void cc_k_fill_f32(float *p, float v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// cc_ingest_batch on the scratch set S (c->ing_mu held by the caller)
static int ingest_on(cc_ctx *c, cc_ctx::Scratch &S, const float *d_xyzi, const int64_t *h_offsets, int n_scans, cc_scan_desc_t *d_out,
                     const cc_ingest_debug_t *dbg, hipStream_t stream) {
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < n_scans; i++) {
    const int64_t n = h_offsets[i + 1] - h_offsets[i];
    if (!(n > 10)) return set_err(CC_EINVAL, "cc_ingest_batch: scan with <= 10 points (CHECK_GT(size, 10), contour_mng.h:507)");
    if (n >= (1 << CC_K1_IDX_BITS)) return set_err(CC_EINVAL, "cc_ingest_batch: scan with >= 2^21 points");
  }
  const size_t nc = (size_t)c->dcfg.n_cell;
  if (S.has_last && S.last_stream != stream) HIPCHK(hipStreamWaitEvent(stream, S.ev_last, 0));
  for (int b0 = 0; b0 < n_scans; b0 += S.cap) {
    const int nb = (n_scans - b0 < S.cap) ? n_scans - b0 : S.cap;
    // offsets relative to the chunk's first point, staged in pinned memory: the call only queues work
    const int slot = c->off_next;
    c->off_next = (slot + 1) % cc_ctx::NSLOT;
    if (c->off_busy[slot]) HIPCHK(hipEventSynchronize(c->off_ev[slot]));
    long long *off = c->h_off[slot];
    for (int i = 0; i <= nb; i++) off[i] = (long long)(h_offsets[b0 + i] - h_offsets[b0]);
    HIPCHK(hipMemcpyAsync(S.d_offsets, off, sizeof(long long) * (nb + 1), hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(c->off_ev[slot], stream));
    c->off_busy[slot] = true;
    const float4 *pts = (const float4 *)d_xyzi + h_offsets[b0];
    // K1's dense image / positions: for the debug outputs, for a configuration K2's list kernel hands on as a whole
    // (min_cont_cell_cnt_ > 3), CC_K1_DENSE=1 (tuning aid); otherwise only for scans whose active cells overflow the list
    // CC_K1_WGS: fewer K1 workgroups, each keeping its CU for scans b, b + grid, ... (k_rasterize.h; tuning aid)
    const int k1_grid = (c->k1_wgs > 0 && nb > c->k1_wgs) ? c->k1_wgs : nb;
    const int want_dense = ((dbg && (dbg->d_bev || dbg->d_pix_rc)) || c->dcfg.min_cont_cell_cnt > 3 || c->k1_dense) ? 1 : 0;
    if (dbg && dbg->d_pix_rc)
      hipLaunchKernelGGL(cc_k_fill_f32, dim3(512), dim3(256), 0, stream, (float *)S.d_pix, -1.f, nc * 2 * nb);
    hipEvent_t *pe = nullptr;
    if (c->prof) {
      if (c->ev_used + 3 > c->ev.size() && prof_flush(c) != CC_OK) return set_err(CC_EHIP, "profiling event sync failed");
      pe = &c->ev[c->ev_used];
      c->ev_used += 3;
      HIPCHK(hipEventRecord(pe[0], stream));
    }
    if (nb <= CC_K1_SPLIT_MAX_SCANS && !c->k1_nosplit) {
      // a handful of scans (the per-scan loop brings one): CC_K1_SPLIT workgroups per scan sweep a range of its points each,
      // a second small kernel combines the ranges (first range wins ties: file order)
      if (!S.k1_part.key) {
        const size_t np = (size_t)CC_K1_SPLIT_MAX_SCANS * CC_K1_SPLIT;
        HIPCHK(hipMalloc(&S.k1_part.key, sizeof(unsigned) * np * nc));
        HIPCHK(hipMalloc(&S.k1_part.idx, sizeof(int) * np * nc));
        HIPCHK(hipMalloc(&S.k1_part.red, sizeof(unsigned) * np * 2));
      }
      if (c->dcfg.reso_pow2 && !c->k1_div)
        hipLaunchKernelGGL((cc_k_rasterize<CC_K1_U_DEFAULT, true, true>), dim3(nb * CC_K1_SPLIT), dim3(CC_INGEST_BLOCK), c->lds1, stream, c->dcfg, pts,
                           (const long long *)S.d_offsets, S.d_bev, S.d_pix, S.d_k1, S.k1_part, S.list, want_dense, nb * CC_K1_SPLIT);
      else
        hipLaunchKernelGGL((cc_k_rasterize<CC_K1_U_DEFAULT, false, true>), dim3(nb * CC_K1_SPLIT), dim3(CC_INGEST_BLOCK), c->lds1, stream, c->dcfg, pts,
                           (const long long *)S.d_offsets, S.d_bev, S.d_pix, S.d_k1, S.k1_part, S.list, want_dense, nb * CC_K1_SPLIT);
      hipLaunchKernelGGL(cc_k_rasterize_merge, dim3(nb), dim3(1024), 0, stream, c->dcfg, pts, (const long long *)S.d_offsets, S.k1_part, S.d_bev,
                         S.d_pix, S.d_k1, S.list, want_dense);
    } else if (c->dcfg.reso_pow2 && !c->k1_div)
      hipLaunchKernelGGL((cc_k_rasterize<CC_K1_U_DEFAULT, true>), dim3(k1_grid), dim3(CC_INGEST_BLOCK), c->lds1, stream, c->dcfg, pts, (const long long *)S.d_offsets,
                         S.d_bev, S.d_pix, S.d_k1, cc_k1_part(), S.list, want_dense, nb);
    else
      hipLaunchKernelGGL((cc_k_rasterize<CC_K1_U_DEFAULT, false>), dim3(k1_grid), dim3(CC_INGEST_BLOCK), c->lds1, stream, c->dcfg, pts, (const long long *)S.d_offsets,
                         S.d_bev, S.d_pix, S.d_k1, cc_k1_part(), S.list, want_dense, nb);
    if (pe) HIPCHK(hipEventRecord(pe[1], stream));
    int16_t *lab = (dbg && dbg->d_labels) ? dbg->d_labels + (size_t)b0 * CC_NLEV * nc : nullptr;
    hipLaunchKernelGGL(cc_k_contours, dim3(nb), dim3(CC_K2_BLOCK), (size_t)CC_K2L_LDS_BYTES, stream, c->dcfg, (const float *)S.d_bev,
                       (const float2 *)S.d_pix, (const cc_k1_scan_out *)S.d_k1, S.d_scr, d_out + b0, lab, c->d_phase_clk, S.d_midq, S.list);
    // the scans the list kernel handed on (more active cells / components than its LDS tables hold): the original body.
    // Its workgroups need 78 KB of LDS each to START, even those that find the queue empty and leave at once: behind a
    // pipelined ingest 512 of them waited 0.13 ms for their turns (the other streams' kernels hold the LDS).  So the
    // launch is sized by what the previous one found: 16 workgroups while the queue stays empty (they take whatever shows
    // up, one scan after the other, and the next launch is a full one again), 512 otherwise and at first.
    const int mid_seen = *(volatile int *)S.h_mid_seen;
    const int mid_wgs = mid_seen == 0 ? 16 : 512;
    hipLaunchKernelGGL(cc_k_contours_mid, dim3(nb < mid_wgs ? nb : mid_wgs), dim3(CC_K2_BLOCK), c->lds2, stream, c->dcfg, S.d_bev, S.d_pix,
                       (const cc_k1_scan_out *)S.d_k1, S.d_scr, S.d_midq, S.d_bigq, d_out + b0, lab, S.h_mid_seen, S.list);
    // the scans the launch above could not number (more than CC_MAXC components on a level): exact, slow, usually none
    hipLaunchKernelGGL(cc_k_contours_big, dim3(nb < S.n_bigslots ? nb : S.n_bigslots), dim3(CC_K2_BLOCK), c->lds2, stream, c->dcfg,
                       (const float *)S.d_bev, (const float2 *)S.d_pix, (const cc_k1_scan_out *)S.d_k1, S.d_bigslots, S.d_bigq, d_out + b0, lab);
    if (pe) HIPCHK(hipEventRecord(pe[2], stream));
    HIPCHK(hipGetLastError());
    if (dbg && dbg->d_bev)
      HIPCHK(hipMemcpyAsync(dbg->d_bev + (size_t)b0 * nc, S.d_bev, sizeof(float) * nc * nb, hipMemcpyDeviceToDevice, stream));
    if (dbg && dbg->d_pix_rc)
      HIPCHK(hipMemcpyAsync(dbg->d_pix_rc + (size_t)b0 * nc * 2, S.d_pix, sizeof(float2) * nc * nb, hipMemcpyDeviceToDevice, stream));
  }
  if (n_scans > 0) {
    HIPCHK(hipEventRecord(S.ev_last, stream));
    S.last_stream = stream;
    S.has_last = true;
  }
  return CC_OK;
}


int cc_ingest_batch(cc_ctx *c, const float *d_xyzi, const int64_t *h_offsets, int n_scans, cc_scan_desc_t *d_out,
                    const cc_ingest_debug_t *dbg, void *stream_) {
  if (!c || !d_xyzi || !h_offsets || !d_out || n_scans < 0) return set_err(CC_EINVAL, "cc_ingest_batch: bad argument");
  std::lock_guard<std::recursive_mutex> ing_lk(c->ing_mu);  // offsets ring, K1/K2 scratch, ev_last: one call at a time
  return ingest_on(c, c->main, d_xyzi, h_offsets, n_scans, d_out, dbg, (hipStream_t)stream_);
}

int cc_ingest_host(cc_ctx *c, const float *h_xyzi, const int64_t *h_offsets, int n_scans, cc_scan_desc_t *h_out) {
  return cc_ingest_host_bev(c, h_xyzi, h_offsets, n_scans, h_out, nullptr);
}

int cc_ingest_host_bev(cc_ctx *c, const float *h_xyzi, const int64_t *h_offsets, int n_scans, cc_scan_desc_t *h_out, float *h_bev) {
  if (!c || !h_xyzi || !h_offsets || !h_out || n_scans < 1) return set_err(CC_EINVAL, "cc_ingest_host: bad argument");
  HIPCHK(hipSetDevice(c->device));
  const int64_t base = h_offsets[0], total = h_offsets[n_scans] - base;
  const size_t bev_bytes = sizeof(float) * (size_t)c->dcfg.n_cell * (size_t)n_scans;
  float *d_x = nullptr, *d_b = nullptr;
  cc_scan_desc_t *d_o = nullptr;
  HIPCHK(hipMalloc(&d_x, sizeof(float) * 4 * (size_t)total));
  hipError_t e = hipMalloc(&d_o, sizeof(cc_scan_desc_t) * (size_t)n_scans);
  if (e == hipSuccess && h_bev) e = hipMalloc(&d_b, bev_bytes);
  if (e != hipSuccess) {
    hipFree(d_x);
    hipFree(d_o);
    return set_err(CC_EHIP, "cc_ingest_host: hipMalloc", e);
  }
  cc_ingest_debug_t dbg;
  dbg.d_bev = d_b;
  dbg.d_pix_rc = nullptr;
  dbg.d_labels = nullptr;
  int rc = CC_OK;
  std::vector<int64_t> off(n_scans + 1);
  for (int i = 0; i <= n_scans; i++) off[i] = h_offsets[i] - base;
  e = hipMemcpy(d_x, h_xyzi + 4 * base, sizeof(float) * 4 * (size_t)total, hipMemcpyHostToDevice);
  if (e != hipSuccess) rc = set_err(CC_EHIP, "cc_ingest_host: H2D", e);
  if (rc == CC_OK) rc = cc_ingest_batch(c, d_x, off.data(), n_scans, d_o, h_bev ? &dbg : nullptr, nullptr);
  if (rc == CC_OK) {
    e = hipMemcpy(h_out, d_o, sizeof(cc_scan_desc_t) * (size_t)n_scans, hipMemcpyDeviceToHost);
    if (e == hipSuccess && h_bev) e = hipMemcpy(h_bev, d_b, bev_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = set_err(CC_EHIP, "cc_ingest_host: D2H", e);
  }
  if (rc == CC_OK)
    for (int i = 0; i < n_scans; i++)
      if (h_out[i].flags & (CC_DESC_INEXACT_COMPONENTS | CC_DESC_INEXACT_KEYS)) {
        rc = set_err(CC_ECAPACITY, "cc_ingest_host: a scan exceeds a fixed capacity of the contour kernel (more than CC_MAXC components on "
                                   "a level, or an over-full key RoI): its descriptor is not exact");
        break;
      }
  hipFree(d_x);
  hipFree(d_o);
  hipFree(d_b);
  return rc;
}

void cc_est_sens_tf(const double tf_bev[3], int n_row, int n_col, double tf_sens[3]) {
  // T_to_tsen^-1 * T_delta * T_so_ssen with T_so_ssen = translate(n_row/2 - 0.5, n_col/2 - 0.5)
  const double ox = n_row / 2 - 0.5, oy = n_col / 2 - 0.5;
  const double c = cos(tf_bev[2]), s = sin(tf_bev[2]);
  tf_sens[0] = c * ox - s * oy + tf_bev[0] - ox;
  tf_sens[1] = s * ox + c * oy + tf_bev[1] - oy;
  tf_sens[2] = tf_bev[2];
}

// ------------------------------------------------------------------------------------------ per-scan loop
struct cc_scan {
  cc_ctx *ctx = nullptr;
  cc_scan_desc_t *d_desc = nullptr;  // device slot (nullptr once offloaded)
  cc_scan_desc_t *h_desc = nullptr;  // host copy (malloc), fetched on demand
  hipEvent_t ready = nullptr;        // recorded on the ingest stream behind the scan's last kernel / copy
  float *h_bev = nullptr;            // host copy of the max-height image, if it was asked for
  bool bev_pending = false;
};

static int loop_reserve_points(cc_ctx *c, int64_t n_points) {  // ing_mu held
  if (!c->s_loop) HIPCHK(stream_take(c->device, &c->s_loop));
  for (auto &ch : c->chan) {
    if (!ch.s) HIPCHK(stream_take(c->device, &ch.s));
    if (ch.scr.cap == 0) {
      const int rc = scratch_alloc(c, ch.scr, 1, 1);
      if (rc != CC_OK) return rc;
    }
  }
  if (n_points <= c->pts_cap) return CC_OK;
  // growing re-allocates every slot: none may be in another thread's hands (being filled) at that moment
  const std::thread::id me = std::this_thread::get_id();
  for (int i = 0; i < cc_ctx::NPTS; i++)
    if (c->pts_handed[i] && c->pts_owner[i] != me)
      return set_err(CC_EINVAL, "cc_stage_points: the staging buffers must grow while another thread fills one of them (stage the largest "
                                "scan first, or give every thread its own context)");
  for (auto &ch : c->chan) HIPCHK(hipStreamSynchronize(ch.s));
  for (int i = 0; i < cc_ctx::NPTS; i++) {
    if (c->h_pts[i]) hipHostFree(c->h_pts[i]);
    c->h_pts[i] = nullptr;
    c->pts_busy[i] = false;
    c->pts_handed[i] = false;
  }
  for (auto &ch : c->chan) {
    hipFree(ch.d_pts);
    ch.d_pts = nullptr;
    ch.d_pts_cap = 0;
  }
  // a quarter more than was asked for (a sequence's scans differ by a few per cent: the buffers should not grow twice), at least
  // 64 K points; pinned memory costs ~0.2-0.6 ms per MB to allocate, so not readKITTIPointCloudBin's 1 M floats up front
  c->pts_cap = n_points + n_points / 4;
  if (c->pts_cap < 65536) c->pts_cap = 65536;
  c->pts_cap = (c->pts_cap + 4095) / 4096 * 4096;
  for (auto &ch : c->chan) {
    HIPCHK(hipMalloc(&ch.d_pts, sizeof(float) * 4 * (size_t)c->pts_cap));
    ch.d_pts_cap = c->pts_cap;
  }
  return CC_OK;
}
// a slot's pinned buffer (pts_cap points) and copy event exist from its first use on
static int loop_slot_alloc(cc_ctx *c, int slot) {  // ing_mu held
  if (!c->pts_ev[slot]) HIPCHK(hipEventCreateWithFlags(&c->pts_ev[slot], hipEventDisableTiming));
  if (!c->h_pts[slot]) HIPCHK(hipHostMalloc((void **)&c->h_pts[slot], sizeof(float) * 4 * (size_t)c->pts_cap, hipHostMallocDefault));
  return CC_OK;
}

// ing_mu held by `lk`.  Hands slot `slot` to the calling thread: waits while another thread holds it, then for the slot's
// last H2D copy (that copy, not the stream).
static float *stage_slot_locked(cc_ctx *c, int64_t n_points, int slot, std::unique_lock<std::recursive_mutex> &lk) {
  const std::thread::id me = std::this_thread::get_id();
  c->pts_cv.wait(lk, [&] { return !c->pts_handed[slot] || c->pts_owner[slot] == me; });
  if (hipSetDevice(c->device) != hipSuccess) return nullptr;
  if (loop_reserve_points(c, n_points) != CC_OK) return nullptr;
  if (loop_slot_alloc(c, slot) != CC_OK) return nullptr;
  if (c->pts_busy[slot]) {
    if (hipEventSynchronize(c->pts_ev[slot]) != hipSuccess) return nullptr;
    c->pts_busy[slot] = false;
  }
  c->pts_handed[slot] = true;
  c->pts_owner[slot] = me;
  return c->h_pts[slot];
}

float *cc_stage_points_slot(cc_ctx *c, int64_t n_points, int slot) {
  if (!c || n_points < 1 || slot < 0 || slot >= cc_ctx::OWN_SLOT) return nullptr;
  std::unique_lock<std::recursive_mutex> lk(c->ing_mu);
  return stage_slot_locked(c, n_points, slot, lk);
}
float *cc_stage_points(cc_ctx *c, int64_t n_points) {
  if (!c || n_points < 1) return nullptr;
  std::unique_lock<std::recursive_mutex> lk(c->ing_mu);
  return stage_slot_locked(c, n_points, cc_ctx::OWN_SLOT, lk);
}

int cc_stage_points_cancel(cc_ctx *c, const float *staged) {
  if (!c || !staged) return set_err(CC_EINVAL, "cc_stage_points_cancel: bad argument");
  std::unique_lock<std::recursive_mutex> lk(c->ing_mu);
  for (int i = 0; i < cc_ctx::NPTS; i++)
    if (c->h_pts[i] == staged) {
      if (!c->pts_handed[i] || c->pts_owner[i] != std::this_thread::get_id())
        return set_err(CC_EINVAL, "cc_stage_points_cancel: the buffer is not in this thread's hands");
      c->pts_handed[i] = false;
      c->pts_cv.notify_all();
      return CC_OK;
    }
  return set_err(CC_EINVAL, "cc_stage_points_cancel: not a staging buffer of this context");
}

int cc_scan_ingest(cc_ctx *c, const float *h_xyzi, int64_t n_points, int want_bev, cc_scan **out) {
  if (!c || !h_xyzi || !out || n_points < 1) return set_err(CC_EINVAL, "cc_scan_ingest: bad argument");
  std::unique_lock<std::recursive_mutex> lk(c->ing_mu);  // d_pts, the slots, the scratch behind cc_ingest_batch
  HIPCHK(hipSetDevice(c->device));
  int slot = -1;
  for (int i = 0; i < cc_ctx::NPTS; i++)
    if (c->h_pts[i] && h_xyzi == c->h_pts[i]) slot = i;
  if (slot < 0) {
    float *dst = stage_slot_locked(c, n_points, cc_ctx::OWN_SLOT, lk);  // waits for the slot's holder and its previous copy, grows the buffers if need be
    if (!dst) return set_err(CC_EHIP, "cc_scan_ingest: staging buffer");
    memcpy(dst, h_xyzi, sizeof(float) * 4 * (size_t)n_points);
    slot = cc_ctx::OWN_SLOT;
  } else if (!c->pts_handed[slot] || c->pts_owner[slot] != std::this_thread::get_id()) {
    return set_err(CC_EINVAL, "cc_scan_ingest: the staging buffer was not handed to this thread by cc_stage_points* (or was ingested already)");
  } else if (n_points > c->pts_cap) {
    return set_err(CC_EINVAL, "cc_scan_ingest: more points than were staged");
  }
  // whatever happens below, the buffer is no longer the caller's: the next thread waiting for the slot may have it once this
  // call has queued (or given up on) the copy
  struct hand_back {
    cc_ctx *c;
    int slot;
    ~hand_back() {
      c->pts_handed[slot] = false;
      c->pts_cv.notify_all();
    }
  } hb{c, slot};
  // The scan goes to the next channel: its own stream, device point buffer and one-scan scratch set, so that it can run next to
  // the scan before it (a caller may ingest scans i + 1, i + 2 -- from a helper thread, as the evaluator mirror does -- while scan i
  // is queried and added on the loop stream); whoever reads the descriptor waits for `ready`.
  cc_ctx::Channel &ch = c->chan[c->chan_next];
  c->chan_next = (c->chan_next + 1) % cc_ctx::NCHAN;
  if (want_bev && !ch.d_bev_copy) HIPCHK(hipMalloc(&ch.d_bev_copy, sizeof(float) * (size_t)c->dcfg.n_cell));
  HIPCHK(hipMemcpyAsync(ch.d_pts, c->h_pts[slot], sizeof(float) * 4 * (size_t)n_points, hipMemcpyHostToDevice, ch.s));
  HIPCHK(hipEventRecord(c->pts_ev[slot], ch.s));
  c->pts_busy[slot] = true;
  cc_scan *sc = new cc_scan();  // from here on every failure path gives the handle (and, once taken, the descriptor slot) back
  sc->ctx = c;
  {
    std::lock_guard<std::mutex> lk(c->slot_mu);
    if (c->slot_free.empty()) {
      // the pool grows geometrically (64, 64, 128, 256, ... up to 1 024 slots = 169 MB per block): a hipMalloc synchronises the
      // device, and a driver that keeps thousands of scans resident should not pay that every 64 scans of its loop
      size_t have = 0;
      for (size_t b = 0; b < c->slot_blocks.size(); b++) have += c->slot_block_n[b];
      const int nblk = (int)(have < 64 ? 64 : (have > 1024 ? 1024 : have));
      cc_scan_desc_t *blk = nullptr;
      const hipError_t e_ = hipMalloc(&blk, sizeof(cc_scan_desc_t) * nblk);
      if (e_ != hipSuccess) {
        delete sc;
        return set_err(CC_EHIP, "cc_scan_ingest: descriptor slots", e_);
      }
      c->slot_blocks.push_back(blk);
      c->slot_block_n.push_back(nblk);
      for (int i = nblk - 1; i >= 0; i--) c->slot_free.push_back(blk + i);
    }
    sc->d_desc = c->slot_free.back();
    c->slot_free.pop_back();
  }
  auto give_back = [&](void) {
    {
      std::lock_guard<std::mutex> lk(c->slot_mu);
      c->slot_free.push_back(sc->d_desc);
    }
    if (sc->ready) hipEventDestroy(sc->ready);
    free(sc->h_bev);
    delete sc;
  };
  const int64_t off[2] = {0, n_points};
  cc_ingest_debug_t dbg;
  dbg.d_bev = want_bev ? ch.d_bev_copy : nullptr;
  dbg.d_pix_rc = nullptr;
  dbg.d_labels = nullptr;
  const int rc = ingest_on(c, ch.scr, ch.d_pts, off, 1, sc->d_desc, want_bev ? &dbg : nullptr, ch.s);
  if (rc != CC_OK) {
    give_back();
    return rc;
  }
  if (want_bev) {  // the image scratch is shared: bring it over now (asynchronously, into the handle's own buffer)
    sc->h_bev = (float *)malloc(sizeof(float) * (size_t)c->dcfg.n_cell);
    if (!sc->h_bev) {
      give_back();
      return set_err(CC_ENOMEM, "cc_scan_ingest: out of host memory");
    }
    const hipError_t e_ = hipMemcpyAsync(sc->h_bev, ch.d_bev_copy, sizeof(float) * (size_t)c->dcfg.n_cell, hipMemcpyDeviceToHost, ch.s);
    if (e_ != hipSuccess) {
      give_back();
      return set_err(CC_EHIP, "cc_scan_ingest: copy of the BEV image", e_);
    }
    sc->bev_pending = true;
  }
  hipError_t e_ = hipEventCreateWithFlags(&sc->ready, hipEventDisableTiming);
  if (e_ == hipSuccess) e_ = hipEventRecord(sc->ready, ch.s);
  if (e_ != hipSuccess) {
    hipStreamSynchronize(ch.s);  // the queued kernels write the slot
    give_back();
    return set_err(CC_EHIP, "cc_scan_ingest: ready event", e_);
  }
  *out = sc;
  return CC_OK;
}

// cc_scan_ingest for 1..CC_SCAN_BATCH_MAX staged scans at once: ONE K1/K2 launch chain for all of them on the next channel (a
// single scan's chain takes ~0.2 ms of launch latencies whatever it holds; a loop that reads its files ahead pays that per batch).
// The batch's descriptors are written side by side and then moved to their own slots of the pool by one small kernel, so every
// handle is an ordinary cc_scan afterwards.  All-or-nothing: on an error no handle is returned and every buffer is given back.
struct cc_desc_out_tab {
  cc_scan_desc_t *p[CC_SCAN_BATCH_MAX];
};
#define CC_SCATTER_BLOCKS 16  // workgroups per descriptor
__global__ void __launch_bounds__(256)
cc_k_scatter_desc(cc_desc_out_tab tab, int n, const cc_scan_desc_t *__restrict__ src) {
  const int s = (int)blockIdx.x / CC_SCATTER_BLOCKS, part = (int)blockIdx.x % CC_SCATTER_BLOCKS;
  if (s >= n) return;
  const unsigned long long *__restrict__ in = (const unsigned long long *)(src + s);
  unsigned long long *__restrict__ out = (unsigned long long *)tab.p[s];
  const int nv = (int)(sizeof(cc_scan_desc_t) / 8);
  for (int i = part * 256 + (int)threadIdx.x; i < nv; i += CC_SCATTER_BLOCKS * 256) out[i] = in[i];
}

int cc_scan_ingest_batch(cc_ctx *c, const float *const *h_xyzi, const int64_t *n_points, int n, cc_scan **out) {
  if (!c || !h_xyzi || !n_points || !out || n < 1 || n > CC_SCAN_BATCH_MAX)
    return set_err(CC_EINVAL, "cc_scan_ingest_batch: bad argument (1..CC_SCAN_BATCH_MAX scans)");
  static_assert(sizeof(cc_scan_desc_t) % 8 == 0, "cc_k_scatter_desc copies 8 bytes per lane");
  std::unique_lock<std::recursive_mutex> lk(c->ing_mu);
  HIPCHK(hipSetDevice(c->device));
  const std::thread::id me = std::this_thread::get_id();
  int slot[CC_SCAN_BATCH_MAX];
  int64_t off[CC_SCAN_BATCH_MAX + 1];
  off[0] = 0;
  for (int i = 0; i < n; i++) {
    slot[i] = -1;
    for (int k = 0; k < cc_ctx::NPTS; k++)
      if (c->h_pts[k] && h_xyzi[i] == c->h_pts[k]) slot[i] = k;
    if (slot[i] < 0 || !c->pts_handed[slot[i]] || c->pts_owner[slot[i]] != me)
      return set_err(CC_EINVAL, "cc_scan_ingest_batch: every buffer must be a staging buffer handed to this thread by cc_stage_points*");
    for (int k = 0; k < i; k++)
      if (slot[k] == slot[i]) return set_err(CC_EINVAL, "cc_scan_ingest_batch: the same staging buffer twice");
    if (n_points[i] < 1 || n_points[i] > c->pts_cap) return set_err(CC_EINVAL, "cc_scan_ingest_batch: more points than were staged");
    if (!(n_points[i] > 10)) return set_err(CC_EINVAL, "cc_scan_ingest_batch: scan with <= 10 points (CHECK_GT(size, 10), contour_mng.h:507)");
    off[i + 1] = off[i] + n_points[i];
  }
  // from here on the buffers are no longer the caller's, whatever happens
  struct hand_back {
    cc_ctx *c;
    const int *slot;
    int n;
    ~hand_back() {
      for (int i = 0; i < n; i++) c->pts_handed[slot[i]] = false;
      c->pts_cv.notify_all();
    }
  } hb{c, slot, n};
  cc_ctx::Channel &ch = c->chan[c->chan_next];
  c->chan_next = (c->chan_next + 1) % cc_ctx::NCHAN;
  // the channel's point buffer, scratch set and descriptor row grow to a batch's size the first time a batch comes by
  if (ch.d_pts_cap < off[n] || ch.scr.cap < n || !ch.d_desc_tmp) {
    HIPCHK(hipStreamSynchronize(ch.s));
    if (ch.d_pts_cap < (int64_t)CC_SCAN_BATCH_MAX * c->pts_cap) {
      hipFree(ch.d_pts);
      ch.d_pts = nullptr;
      ch.d_pts_cap = 0;
      HIPCHK(hipMalloc(&ch.d_pts, sizeof(float) * 4 * (size_t)c->pts_cap * CC_SCAN_BATCH_MAX));
      ch.d_pts_cap = (int64_t)CC_SCAN_BATCH_MAX * c->pts_cap;
    }
    if (ch.scr.cap < CC_SCAN_BATCH_MAX) {
      scratch_free(ch.scr);
      const int rc = scratch_alloc(c, ch.scr, CC_SCAN_BATCH_MAX, CC_SCAN_BATCH_MAX);
      if (rc != CC_OK) return rc;
    }
    if (!ch.d_desc_tmp) HIPCHK(hipMalloc(&ch.d_desc_tmp, sizeof(cc_scan_desc_t) * CC_SCAN_BATCH_MAX));
  }
  for (int i = 0; i < n; i++) {
    HIPCHK(hipMemcpyAsync(ch.d_pts + 4 * (size_t)off[i], c->h_pts[slot[i]], sizeof(float) * 4 * (size_t)n_points[i], hipMemcpyHostToDevice, ch.s));
    HIPCHK(hipEventRecord(c->pts_ev[slot[i]], ch.s));
    c->pts_busy[slot[i]] = true;
  }
  cc_scan *sc[CC_SCAN_BATCH_MAX] = {};
  cc_desc_out_tab tab;
  for (int i = 0; i < CC_SCAN_BATCH_MAX; i++) tab.p[i] = nullptr;
  int n_have = 0;
  auto give_back = [&](void) {
    {
      std::lock_guard<std::mutex> slk(c->slot_mu);
      for (int i = 0; i < n_have; i++)
        if (sc[i] && sc[i]->d_desc) c->slot_free.push_back(sc[i]->d_desc);
    }
    for (int i = 0; i < n_have; i++) {
      if (!sc[i]) continue;
      if (sc[i]->ready) hipEventDestroy(sc[i]->ready);
      delete sc[i];
    }
  };
  {
    std::lock_guard<std::mutex> slk(c->slot_mu);
    for (int i = 0; i < n; i++) {
      if (c->slot_free.empty()) {
        size_t have = 0;
        for (size_t b = 0; b < c->slot_blocks.size(); b++) have += c->slot_block_n[b];
        const int nblk = (int)(have < 64 ? 64 : (have > 1024 ? 1024 : have));
        cc_scan_desc_t *blk = nullptr;
        const hipError_t e_ = hipMalloc(&blk, sizeof(cc_scan_desc_t) * nblk);
        if (e_ != hipSuccess) {
          for (int k = 0; k < n_have; k++) {
            c->slot_free.push_back(sc[k]->d_desc);
            delete sc[k];
          }
          return set_err(CC_EHIP, "cc_scan_ingest_batch: descriptor slots", e_);
        }
        c->slot_blocks.push_back(blk);
        c->slot_block_n.push_back(nblk);
        for (int k = nblk - 1; k >= 0; k--) c->slot_free.push_back(blk + k);
      }
      sc[i] = new cc_scan();
      sc[i]->ctx = c;
      sc[i]->d_desc = c->slot_free.back();
      c->slot_free.pop_back();
      tab.p[i] = sc[i]->d_desc;
      n_have = i + 1;
    }
  }
  const int rc = ingest_on(c, ch.scr, ch.d_pts, off, n, ch.d_desc_tmp, nullptr, ch.s);
  if (rc != CC_OK) {
    give_back();
    return rc;
  }
  hipLaunchKernelGGL(cc_k_scatter_desc, dim3(n * CC_SCATTER_BLOCKS), dim3(256), 0, ch.s, tab, n, (const cc_scan_desc_t *)ch.d_desc_tmp);
  hipError_t e_ = hipGetLastError();
  for (int i = 0; i < n && e_ == hipSuccess; i++) {
    e_ = hipEventCreateWithFlags(&sc[i]->ready, hipEventDisableTiming);
    if (e_ == hipSuccess) e_ = hipEventRecord(sc[i]->ready, ch.s);
  }
  if (e_ != hipSuccess) {
    hipStreamSynchronize(ch.s);  // the queued kernels write the slots
    give_back();
    return set_err(CC_EHIP, "cc_scan_ingest_batch: launch / ready events", e_);
  }
  for (int i = 0; i < n; i++) out[i] = sc[i];
  return CC_OK;
}

// 1: the scan's ingest has finished on the device (its descriptor can be read without waiting), 0: still in flight
int cc_scan_ready(const cc_scan *sc) {
  if (!sc) return 0;
  if (!sc->ready) return 1;
  hipSetDevice(sc->ctx->device);
  const hipError_t e_ = hipEventQuery(sc->ready);
  if (e_ != hipSuccess) (void)hipGetLastError();
  return e_ == hipSuccess ? 1 : 0;
}

// the loop stream (or the host) behind the scan's ingest
static int scan_wait_ready(cc_scan *sc, bool host) {
  if (!sc->ready) return CC_OK;
  if (host)
    HIPCHK(hipEventSynchronize(sc->ready));
  else
    HIPCHK(hipStreamWaitEvent(sc->ctx->s_loop, sc->ready, 0));
  return CC_OK;
}

static int scan_fetch(cc_scan *sc) {
  if (sc->h_desc) return CC_OK;
  if (!sc->d_desc) return set_err(CC_EINVAL, "cc_scan: the descriptor is neither on the device nor on the host");
  sc->h_desc = (cc_scan_desc_t *)malloc(sizeof(cc_scan_desc_t));
  if (!sc->h_desc) return set_err(CC_ENOMEM, "cc_scan: out of host memory");
  HIPCHK(hipSetDevice(sc->ctx->device));
  const int rcw = scan_wait_ready(sc, false);
  if (rcw != CC_OK) return rcw;
  HIPCHK(hipMemcpyAsync(sc->h_desc, sc->d_desc, sizeof(cc_scan_desc_t), hipMemcpyDeviceToHost, sc->ctx->s_loop));
  HIPCHK(hipStreamSynchronize(sc->ctx->s_loop));
  sc->bev_pending = false;  // the image copy was queued before `ready`
  return CC_OK;
}

int cc_scan_desc(cc_scan *sc, const cc_scan_desc_t **h_desc) {
  if (!sc || !h_desc) return set_err(CC_EINVAL, "cc_scan_desc: bad argument");
  const int rc = scan_fetch(sc);
  if (rc != CC_OK) return rc;
  *h_desc = sc->h_desc;
  if (sc->h_desc->flags & (CC_DESC_INEXACT_COMPONENTS | CC_DESC_INEXACT_KEYS))
    return set_err(CC_ECAPACITY, "cc_scan_desc: the scan exceeds a fixed capacity of the contour kernel (more than CC_MAXC components on a "
                                 "level, or an over-full key RoI): its descriptor is not exact");
  return CC_OK;
}

int cc_scan_bev(cc_scan *sc, const float **h_bev) {
  if (!sc || !h_bev) return set_err(CC_EINVAL, "cc_scan_bev: bad argument");
  if (!sc->h_bev) return set_err(CC_EINVAL, "cc_scan_bev: the image was not asked for at cc_scan_ingest");
  if (sc->bev_pending) {
    HIPCHK(hipSetDevice(sc->ctx->device));
    const int rcw = scan_wait_ready(sc, true);
    if (rcw != CC_OK) return rcw;
    sc->bev_pending = false;
  }
  *h_bev = sc->h_bev;
  return CC_OK;
}

int cc_scan_offload(cc_scan *sc) {
  if (!sc) return set_err(CC_EINVAL, "cc_scan_offload: bad argument");
  if (!sc->d_desc) return CC_OK;
  const int rc = scan_fetch(sc);
  if (rc != CC_OK) return rc;
  {  // the fetch above synchronised the loop stream: nothing queued reads the slot any more
    std::lock_guard<std::mutex> lk(sc->ctx->slot_mu);
    sc->ctx->slot_free.push_back(sc->d_desc);
  }
  sc->d_desc = nullptr;
  return CC_OK;
}

int cc_scan_on_device(const cc_scan *sc) { return sc && sc->d_desc ? 1 : 0; }

int cc_scan_release(cc_scan *sc) {
  if (!sc) return CC_OK;
  if (sc->d_desc || sc->bev_pending) {
    hipSetDevice(sc->ctx->device);
    if (sc->ready) hipEventSynchronize(sc->ready);               // the ingest may still write the slot / the image
    if (sc->ctx->s_loop) hipStreamSynchronize(sc->ctx->s_loop);  // queued work may still read the slot
    if (sc->d_desc) {
      std::lock_guard<std::mutex> lk(sc->ctx->slot_mu);
      sc->ctx->slot_free.push_back(sc->d_desc);
    }
  }
  if (sc->ready) hipEventDestroy(sc->ready);
  free(sc->h_desc);
  free(sc->h_bev);
  delete sc;
  return CC_OK;
}

#include "cc_db_api.inc"
#include "cc_comm.inc"

}  // extern "C"
