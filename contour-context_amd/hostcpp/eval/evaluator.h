// Mirror of the reference's loop-closure evaluator (include/eval/evaluator.h:13-431): same class, method names and
// file formats, no Eigen/glog.  It feeds scans to the detector in order, judges every prediction against the ground
// truth (TP/FP/TN/FN), accumulates pose errors and writes the 8-column outcome file scripts/pr_mpe.py consumes.
//
//   pose file : `ts r00 r01 r02 tx r10 r11 r12 ty r20 r21 r22 tz` per line (sensor pose, z up), any order
//   scan list : `ts seq path` per line, ordered by ts and seq
//   outcome   : `tfpn \t tgt-src \t correlation \t err_x \t err_y \t err_theta \t tgt_path \t src_path` (`x` = no candidate)
#pragma once
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <type_traits>

#include "../cont2/contour_db.h"

// nearest element of a sorted vector within `tol`, or -1 (tools/algos.h:77-90)
template <typename T>
int lookupNN(const T &q_val, const std::vector<T> &sorted_vec, const T &tol) {
  if (sorted_vec.empty()) return -1;
  auto lo = std::lower_bound(sorted_vec.begin(), sorted_vec.end(), q_val);
  auto it = lo;
  if (lo == sorted_vec.end())
    it = lo - 1;
  else if (lo != sorted_vec.begin())
    it = std::abs(q_val - *lo) < std::abs(q_val - *(lo - 1)) ? lo : lo - 1;
  if (std::abs(*it - q_val) > tol) return -1;
  return (int)(it - sorted_vec.begin());
}

template <int dim>
struct SimpleRMSE {  // evaluator.h:13-33
  double sum_sqs = 0, sum_abs = 0;
  int cnt_sqs = 0;
  void addOneErr(const double *d) {
    double sq = 0;
    for (int i = 0; i < dim; i++) sq += d[i] * d[i];
    cnt_sqs++;
    sum_sqs += sq;
    sum_abs += std::sqrt(sq);
  }
  double getRMSE() const { return cnt_sqs ? std::sqrt(sum_sqs / cnt_sqs) : -1; }
  double getMean() const { return cnt_sqs ? sum_abs / cnt_sqs : -1; }
};

struct PredictionOutcome {  // evaluator.h:35-45
  enum Res { TP, FP, TN, FN };
  int id_src = -1;
  int id_tgt = -1;
  Res tfpn = Res::TN;
  double est_err[3]{};  // error on SE(2) when a candidate was proposed, else zero
  double correlation{};
};

class ContLCDEvaluator {
 public:
  struct LaserScanInfo {
    bool has_gt_positive_lc = false;
    Eigen::Isometry3d sens_pose;
    int seq = 0;
    double ts = 0;
    std::string fpath;
  };

 private:
  std::vector<LaserScanInfo> laser_info_;
  std::vector<int> assigned_seqs_;
  const double ts_diff_tol = 10e-3;   // a scan is used only if a gt pose lies within 10 ms
  const double min_time_excl = 15.0;  // revisits younger than 15 s are not loops
  const double sim_thres;             // similarity at or above which a prediction counts as positive
  int p_lidar_curr = -1;
  // The scans ahead of the driver's position on their way: file -> pinned staging buffer -> device (getCurrContourManager).
  // One helper thread for the evaluator's lifetime runs up to ahead() scans ahead of the scan the driver holds; what it has
  // finished waits in `ready`, in address order.
  struct Prefetch {
    // Scans in flight ahead of the driver (env CC_EVAL_AHEAD, 1..64; default four ingest batches).  The helper works in
    // BATCHES of up to ingestBatch() scans (env CC_EVAL_INGEST_BATCH, 1..CC_SCAN_BATCH_MAX, default 8, 16 for a long list): their files are read in
    // parallel by readers() threads (env CC_EVAL_READERS, default 4; a 1.9 MB KITTI file takes ~0.15 ms to read into pinned
    // memory, four threads bring a batch of eight over in ~0.4 ms) and go to the device as ONE launch chain (cc_scan_ingest_batch) -- a scan's own chain takes ~0.2 ms of launch
    // latencies whatever it holds.  Staging buffers: scan `addr` goes through slot addr % (2 * ingestBatch()), reused
    // when its copy has passed.
    static int envInt(const char *name, int lo, int hi, int dflt) {
      const char *e = getenv(name);
      return e ? std::min(hi, std::max(lo, atoi(e))) : dflt;
    }
    // set when the helper starts (the list's length is known then): batches of sixteen for a long list (>= 2 048 scans: 13.6 k
    // against 11.2 k scans/s on a 4 096-scan drive), of eight for a short one (even on 1 024 scans, and half the pinned memory)
    int ib_ = 8, ahead_ = 32;
    void configure(int n_scans) {
      ib_ = envInt("CC_EVAL_INGEST_BATCH", 1, CC_SCAN_BATCH_MAX, n_scans >= 2048 ? 16 : 8);
      ahead_ = envInt("CC_EVAL_AHEAD", 1, 64, std::max(4, 4 * ib_));
    }
    int ingestBatch() const { return ib_; }
    int ahead() const { return ahead_; }
    static int readers() {
      static const int r = envInt("CC_EVAL_READERS", 1, 8, 4);
      return r;
    }
    // records a scan file holds, at most what readKITTIPointCloudBin reads (1 000 000 floats); 0 if it cannot be examined.
    // Staging buffers are pinned memory (~0.2-0.6 ms per MB to allocate on the MI355X host): they are asked for at the
    // files' size, not at the reader's maximum.
    static size_t filePoints(const std::string &path, size_t cap) {
      struct stat st;
      if (stat(path.c_str(), &st) != 0 || st.st_size <= 0) return 0;
      return std::min(cap, (size_t)st.st_size / (4 * sizeof(float)));
    }
    struct ReadJob {
      const std::string *path = nullptr;
      float *dst = nullptr;
      size_t cap = 0, n = 0;
      bool opened = false;
    };
    std::vector<std::thread> rd;  // readers() - 1 threads next to the helper, which reads too
    std::mutex rmu;
    std::condition_variable rcv;
    std::deque<ReadJob *> rq;  // under rmu
    int r_open = 0;            // jobs of the current batch not finished yet (under rmu)
    bool r_quit = false;
    std::atomic<long> job_ns{0}, job_n{0};  // tuning aid: time inside the reads themselves
    void readOne(ReadJob &j) {
      const auto t0 = std::chrono::steady_clock::now();
      FILE *f = fopen(j.path->c_str(), "rb");
      j.opened = f != nullptr;
      j.n = f ? fread(j.dst, 4 * sizeof(float), j.cap, f) : 0;
      if (f) fclose(f);
      job_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      job_n++;
    }
    void readerLoop() {
      for (;;) {
        std::unique_lock<std::mutex> lk(rmu);
        rcv.wait(lk, [this] { return r_quit || !rq.empty(); });
        if (r_quit) return;
        ReadJob *j = rq.front();
        rq.pop_front();
        lk.unlock();
        readOne(*j);
        lk.lock();
        if (--r_open == 0) rcv.notify_all();
      }
    }
    // all jobs read, by the pool and the calling thread
    void readAll(std::vector<ReadJob> &jobs) {
      while ((int)rd.size() < readers() - 1) rd.emplace_back([this] { readerLoop(); });
      {
        std::lock_guard<std::mutex> lk(rmu);
        for (auto &j : jobs) rq.push_back(&j);
        r_open = (int)jobs.size();
      }
      rcv.notify_all();
      for (;;) {
        std::unique_lock<std::mutex> lk(rmu);
        if (rq.empty()) {
          rcv.wait(lk, [this] { return r_open == 0; });
          return;
        }
        ReadJob *j = rq.front();
        rq.pop_front();
        lk.unlock();
        readOne(*j);
        lk.lock();
        if (--r_open == 0) {
          rcv.notify_all();
          return;
        }
      }
    }
    void stopReaders() {
      {
        std::lock_guard<std::mutex> lk(rmu);
        r_quit = true;
      }
      rcv.notify_all();
      for (auto &t : rd) t.join();
      rd.clear();
    }
    enum class Status { OK, MISSING_FILE, TOO_FEW_POINTS, STAGING_FAILED, INGEST_FAILED };
    struct Item {
      int addr = -1;
      size_t n = 0;
      Status status = Status::OK;
      cc_scan *scan = nullptr;  // ingested ahead of time (non-null iff status == OK)
      std::string err;          // the library's message for STAGING_FAILED / INGEST_FAILED
    };
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    // all below: under `mu`
    bool busy = false, quit = false;
    int next = -1, limit = -1;  // the helper fetches address `next` while next < limit
    cc_ctx *ctx = nullptr;
    bool with_images = false;
    std::deque<Item> ready;
    double t_stage = 0, t_read = 0, t_ingest = 0;  // helper seconds (CC_EVAL_TIMERS=1 prints them when the evaluator goes)
    long n_done = 0, n_batches = 0;
    double t_wait = 0, t_call = 0;  // driver thread: waiting for the helper's item / the whole getCurrContourManager call
    long n_call = 0;
  };
  mutable Prefetch pf_;
  SimpleRMSE<2> tp_trans_rmse, all_trans_rmse;
  SimpleRMSE<1> tp_rot_rmse, all_rot_rmse;
  std::vector<PredictionOutcome> pred_records;

 public:
  ContLCDEvaluator(const std::string &fpath_pose, const std::string &fpath_laser, const double &bar) : sim_thres(bar) {
    cc_host::runtime_warm();
    // 1. stamped ground-truth poses, sorted by time
    std::ifstream f_pose(fpath_pose);
    if (!f_pose.good()) {
      std::cerr << "Error opening gt pose file: " << fpath_pose << std::endl;
      return;
    }
    std::vector<double> gt_tss;
    std::vector<Eigen::Isometry3d> gt_poses;
    std::string line;
    while (std::getline(f_pose, line)) {
      std::istringstream iss(line);
      double v[13];
      for (double &x : v) CC_CHECK(iss >> x);
      double M[3][3];
      Eigen::Isometry3d T;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) M[r][c] = v[1 + 4 * r + c];
        T.t[r] = v[1 + 4 * r + 3];
      }
      T.setRotationViaQuaternion(M);
      gt_tss.push_back(v[0]);
      gt_poses.push_back(T);
    }
    printf("Added %lu stamped gt poses.\n", gt_poses.size());
    std::vector<int> order(gt_poses.size());
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return gt_tss[a] < gt_tss[b]; });
    {
      std::vector<double> ts2(order.size());
      std::vector<Eigen::Isometry3d> p2(order.size());
      for (size_t i = 0; i < order.size(); i++) {
        ts2[i] = gt_tss[order[i]];
        p2[i] = gt_poses[order[i]];
      }
      gt_tss.swap(ts2);
      gt_poses.swap(p2);
    }
    // 2. scans that have a ground-truth pose
    std::ifstream f_laser(fpath_laser);
    if (!f_laser.good()) {
      std::cerr << "Error opening laser info file: " << fpath_laser << std::endl;
      return;
    }
    size_t n_listed = 0;
    while (std::getline(f_laser, line)) {
      std::istringstream iss(line);
      LaserScanInfo info;
      if (!(iss >> info.ts)) continue;
      iss >> info.seq >> info.fpath;
      n_listed++;
      const int gi = lookupNN<double>(info.ts, gt_tss, ts_diff_tol);
      if (gi < 0) continue;
      info.sens_pose = gt_poses[gi];
      laser_info_.push_back(info);
      assigned_seqs_.push_back(info.seq);
    }
    printf("Added %lu laser bin paths.\n", n_listed);
    printf("Found %d laser scans with gt poses.\n", (int)laser_info_.size());
    for (size_t i = 1; i < laser_info_.size(); i++) {
      CC_CHECK(laser_info_[i - 1].seq < laser_info_[i].seq);
      CC_CHECK(laser_info_[i - 1].ts < laser_info_[i].ts);
    }
    printf("Ordering check passed\n");
    // 3. ground-truth loops: an earlier scan (by at least min_time_excl) within 5 m
    int cnt_gt_lc_p = 0, cnt_gt_lc = 0;
    for (auto &fast : laser_info_) {
      for (const auto &slow : laser_info_) {
        if (fast.ts < slow.ts + min_time_excl) break;
        if ((fast.sens_pose.translation() - slow.sens_pose.translation()).norm() < 5.0) {
          if (!fast.has_gt_positive_lc) {
            fast.has_gt_positive_lc = true;
            cnt_gt_lc_p++;
          }
          cnt_gt_lc++;
        }
      }
    }
    printf("Found %d poses with %d gt loops.\n", cnt_gt_lc_p, cnt_gt_lc);
  }

  bool loadNewScan() {
    p_lidar_curr++;
    CC_CHECK(p_lidar_curr >= 0);
    if (p_lidar_curr >= (int)laser_info_.size()) {
      printf("\n===\ncurrent addr %d exceeds boundary\n", p_lidar_curr);
      return false;
    }
    printf("\n===\nloaded scan addr %d, seq: %d, fpath: %s\n", p_lidar_curr, laser_info_[p_lidar_curr].seq,
           laser_info_[p_lidar_curr].fpath.c_str());
    return true;
  }

  // Mirror-only (tests): the next loadNewScan() loads address `addr` -- a driver that does not walk its list in order
  void jumpTo(int addr) { p_lidar_curr = addr - 1; }

  const LaserScanInfo &getCurrScanInfo() const {
    CC_CHECK(p_lidar_curr >= 0 && p_lidar_curr < (int)laser_info_.size());
    return laser_info_[p_lidar_curr];
  }

  // read the current scan's .bin (x,y,z,i f32; tools/pointcloud_util.h:9-47) and build its descriptor on the device.
  // readKITTIPointCloudBin + makeBEV in one step: the records go straight to one of the context's two pinned staging
  // buffers.  While this scan is queried and added, a helper thread reads the NEXT scans' files (plain fopen / fread) and
  // queues their ingest on the context's ingest stream (cc_scan_ingest: copy of the points, rasterisation, contours --
  // nothing of it is observable before the call that hands the scan out), so that the next call finds its descriptor on
  // the way or done -- the driver's loop (test/batch_bin_test.cpp:131-237) is unchanged.
  std::shared_ptr<ContourManager> getCurrContourManager(const ContourManagerConfig &config) const {
    const auto tc0 = std::chrono::steady_clock::now();
    const LaserScanInfo &info = getCurrScanInfo();
    std::shared_ptr<ContourManager> cm(new ContourManager(config, info.seq));
    std::string str_id = std::to_string(info.seq);
    str_id = "assigned_id_" + std::string(8 - str_id.length(), '0') + str_id;
    cc_ctx *ctx = ContourManager::contextOf(config);  // the first call creates the context (~60 ms, once)
    const size_t cap = 1000000 / 4;  // readKITTIPointCloudBin reads at most 1 000 000 floats
    const bool with_images = ContourManager::keepImages();
    const int n_scans = (int)laser_info_.size();
    bool adopted = false;
    {
      std::unique_lock<std::mutex> lk(pf_.mu);
      // is this scan among the ones the helper has fetched or is fetching (in order, same context, same image switch)?
      const bool coming = pf_.ctx == ctx && pf_.with_images == with_images && pf_.next >= 0 &&
                          ((!pf_.ready.empty() && pf_.ready.front().addr == p_lidar_curr) ||
                           (pf_.ready.empty() && pf_.next == p_lidar_curr && pf_.next < pf_.limit));
      if (coming) {
        const auto tw0 = std::chrono::steady_clock::now();
        pf_.cv.wait(lk, [this] { return !pf_.ready.empty(); });
        pf_.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
        Prefetch::Item it = std::move(pf_.ready.front());
        pf_.ready.pop_front();
        if (it.scan) cc_host::lookahead().popFront(it.scan);  // the driver's from here on
        pf_.limit = std::min(n_scans, p_lidar_curr + 1 + pf_.ahead());
        lk.unlock();
        pf_.cv.notify_all();
        switch (it.status) {  // what the reference does at the same points (evaluator.h:285-302, contour_mng.h:507)
          case Prefetch::Status::MISSING_FILE:
            printf("Lidar bin file %s does not exist.\n", info.fpath.c_str());
            exit(-1);
          case Prefetch::Status::TOO_FEW_POINTS:
            fprintf(stderr, "%s: %zu points\n", info.fpath.c_str(), it.n);
            CC_CHECK(it.n > 10);
            abort();
          case Prefetch::Status::STAGING_FAILED:
          case Prefetch::Status::INGEST_FAILED:
            fprintf(stderr, "cont2_amd: %s of %s: %s\n", it.status == Prefetch::Status::STAGING_FAILED ? "staging buffer" : "ingest",
                    info.fpath.c_str(), it.err.c_str());
            abort();
          case Prefetch::Status::OK:
            break;
        }
        CC_CHECK(it.scan);
        cm->adoptIngested(it.scan, with_images, str_id);
        adopted = true;
      } else {  // first scan, a jump, another configuration: the helper is parked and what it fetched is dropped
        pf_.next = -1;
        pf_.cv.wait(lk, [this] { return !pf_.busy; });
        cc_host::lookahead().invalidate(this);  // databases that worked ahead on these scans drop that work before the scans go
        for (auto &it : pf_.ready)
          if (it.scan) cc_scan_release(it.scan);
        pf_.ready.clear();
      }
    }
    if (!adopted) {
      FILE *f = fopen(info.fpath.c_str(), "rb");
      if (!f) {
        printf("Lidar bin file %s does not exist.\n", info.fpath.c_str());
        exit(-1);
      }
      const size_t cap_f = std::max<size_t>(16, Prefetch::filePoints(info.fpath, cap));
      float *dst = cc_stage_points(ctx, (int64_t)cap_f);  // the driver thread's own slot, never one the helper fills
      CC_CHECK(dst);
      const size_t n = fread(dst, 4 * sizeof(float), cap_f, f);
      fclose(f);
      cm->makeBEVFromStaged(dst, n, str_id);
      static const bool read_ahead = [] {  // CC_EVAL_READ_AHEAD=0: every scan is read and ingested by the call that asks for it
        const char *e = getenv("CC_EVAL_READ_AHEAD");
        return !(e && atoi(e) == 0);
      }();
      if (read_ahead) {
        {
          std::lock_guard<std::mutex> lk(pf_.mu);
          pf_.ctx = ctx;
          pf_.with_images = with_images;
          pf_.configure(n_scans);
          pf_.next = p_lidar_curr + 1;
          pf_.limit = std::min(n_scans, p_lidar_curr + 1 + pf_.ahead());
        }
        if (!pf_.th.joinable()) pf_.th = std::thread([this, cap] { prefetchLoop(cap); });
        pf_.cv.notify_all();
      }
    }
    cm->setScanSource(this);
    cm->makeContoursRecurs();
    pf_.t_call += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
    pf_.n_call++;
    return cm;
  }
  ~ContLCDEvaluator() {
    if (pf_.th.joinable()) {
      {
        std::lock_guard<std::mutex> lk(pf_.mu);
        pf_.quit = true;
      }
      pf_.cv.notify_all();
      pf_.th.join();
    }
    pf_.stopReaders();
    cc_host::lookahead().invalidate(this);
    for (auto &it : pf_.ready)
      if (it.scan) cc_scan_release(it.scan);
    if (getenv("CC_EVAL_TIMERS") && pf_.n_done > 0)
      fprintf(stderr, "[evaluator helper, mean us PER SCAN over %ld scans in %ld batches] staging buffer %.1f  file read %.1f  cc_scan_ingest %.1f | driver thread: wait for the helper %.1f of %.1f per getCurrContourManager (the first call creates the context); one file's read took %.1f us\n",
              pf_.n_done, pf_.n_batches, 1e6 * pf_.t_stage / pf_.n_done, 1e6 * pf_.t_read / pf_.n_done, 1e6 * pf_.t_ingest / pf_.n_done,
              1e6 * pf_.t_wait / std::max(1L, pf_.n_call), 1e6 * pf_.t_call / std::max(1L, pf_.n_call), 1e-3 * pf_.job_ns / std::max(1L, pf_.job_n.load()));
  }

 private:
  void prefetchLoop(size_t cap) const {
    Prefetch &pf = pf_;
    const int n_scans = (int)laser_info_.size();
    for (;;) {
      std::unique_lock<std::mutex> lk(pf.mu);
      int nb = 0;
      // a batch goes out when a whole one fits ahead of the driver -- or with whatever fits when little is queued (the driver
      // is about to wait) or the list ends
      pf.cv.wait(lk, [&] {
        if (pf.quit) return true;
        if (pf.next < 0 || pf.next >= pf.limit) return false;
        const int want = std::min(pf.limit - pf.next, pf.ahead() - (int)pf.ready.size());
        const int ib = pf.with_images ? 1 : pf.ingestBatch();  // a scan that keeps its image goes alone (cc_scan_ingest)
        if (want <= 0) return false;
        if (want < ib && (int)pf.ready.size() >= 2 && pf.limit < n_scans) return false;
        nb = std::min(want, ib);
        return true;
      });
      if (pf.quit) return;
      const int first = pf.next;
      cc_ctx *ctx = pf.ctx;
      const bool with_images = pf.with_images;
      pf.busy = true;
      lk.unlock();
      // a slot's previous occupant is scan addr - n_slots: its copy to the device was queued a batch ago and is waited for here
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<Prefetch::Item> items((size_t)nb);
      std::vector<Prefetch::ReadJob> jobs;
      std::vector<float *> dst((size_t)nb, nullptr);
      std::vector<size_t> caps((size_t)nb, 0);
      const int n_slots = 2 * (with_images ? 1 : pf.ingestBatch());  // one batch is read while the one before it is copied
      int n_staged = 0;
      // every buffer of the batch is asked for at the batch's largest file: if the context's buffers have to grow (they are all
      // re-allocated then), that happens at the first request, before a pointer of this batch is held
      size_t cap_batch = 16;
      for (int j = 0; j < nb; j++) {
        caps[j] = std::max<size_t>(16, Prefetch::filePoints(laser_info_[first + j].fpath, cap));  // (a missing file is found by its reader)
        cap_batch = std::max(cap_batch, caps[j]);
      }
      for (int j = 0; j < nb; j++) {
        items[j].addr = first + j;
        dst[j] = cc_stage_points_slot(ctx, (int64_t)cap_batch, (first + j) % n_slots);
        if (!dst[j]) {
          items[j].status = Prefetch::Status::STAGING_FAILED;
          items[j].err = cc_last_error();  // the message is per thread
          break;
        }
        n_staged = j + 1;
      }
      const auto t1 = std::chrono::steady_clock::now();
      jobs.resize((size_t)n_staged);
      for (int j = 0; j < n_staged; j++) {
        jobs[j].path = &laser_info_[first + j].fpath;
        jobs[j].dst = dst[j];
        jobs[j].cap = caps[j];
      }
      if (n_staged > 0) pf.readAll(jobs);
      const auto t2 = std::chrono::steady_clock::now();
      // the scans up to the first one that cannot be ingested go to the device together; that one is reported when the driver
      // gets there (the reference stops there: evaluator.h:285-302, contour_mng.h:507); what was staged behind it is given back
      int n_good = 0;
      for (int j = 0; j < n_staged; j++) {
        items[j].n = jobs[j].n;
        if (!jobs[j].opened)
          items[j].status = Prefetch::Status::MISSING_FILE;
        else if (jobs[j].n <= 10)
          items[j].status = Prefetch::Status::TOO_FEW_POINTS;
        if (items[j].status != Prefetch::Status::OK) break;
        n_good = j + 1;
      }
      int n_items = n_good;  // items that will be queued
      if (n_good < nb) n_items = n_good + 1;
      if (n_good > 0) {
        cc_scan *sc[CC_SCAN_BATCH_MAX] = {};
        int rc;
        if (n_good == 1) {
          rc = cc_scan_ingest(ctx, dst[0], (int64_t)items[0].n, with_images ? 1 : 0, &sc[0]);
        } else {
          int64_t np[CC_SCAN_BATCH_MAX];
          for (int j = 0; j < n_good; j++) np[j] = (int64_t)items[j].n;
          rc = cc_scan_ingest_batch(ctx, dst.data(), np, n_good, sc);
        }
        if (rc != CC_OK) {  // nothing of the batch was ingested: its first scan carries the error
          items[0].status = Prefetch::Status::INGEST_FAILED;
          items[0].err = cc_last_error();
          n_items = 1;
          n_good = 0;
        } else {
          for (int j = 0; j < n_good; j++) items[j].scan = sc[j];
        }
      }
      for (int j = n_good; j < n_staged; j++) cc_stage_points_cancel(ctx, dst[j]);  // (a no-op for buffers an ingest call has taken)
      const auto t3 = std::chrono::steady_clock::now();
      lk.lock();
      pf.t_stage += std::chrono::duration<double>(t1 - t0).count();
      pf.t_read += std::chrono::duration<double>(t2 - t1).count();
      pf.t_ingest += std::chrono::duration<double>(t3 - t2).count();
      pf.n_done += n_items;
      pf.n_batches++;
      pf.busy = false;
      if (pf.next == first) {  // still wanted (the driver did not jump meanwhile)
        for (int j = 0; j < n_items; j++) {
          if (items[j].scan) cc_host::lookahead().push(items[j].scan, laser_info_[first + j].ts, this);
          pf.ready.push_back(std::move(items[j]));
        }
        pf.next += n_items;
      } else {
        lk.unlock();
        for (int j = 0; j < n_items; j++)
          if (items[j].scan) cc_scan_release(items[j].scan);
        lk.lock();
      }
      lk.unlock();
      pf.cv.notify_all();
    }
  }

 public:
  // judge one query: `cand_mng` is the proposed loop candidate (nullptr: none), T_est_delta_2d its BEV-frame transform
  PredictionOutcome addPrediction(const std::shared_ptr<const ContourManager> &q_mng, double est_corr,
                                  const std::shared_ptr<const ContourManager> &cand_mng = nullptr,
                                  const Eigen::Isometry2d &T_est_delta_2d = Eigen::Isometry2d::Identity()) {
    PredictionOutcome res;
    res.id_tgt = q_mng->getIntID();
    res.correlation = est_corr;
    const int addr_tgt = lookupNN<int>(res.id_tgt, assigned_seqs_, 0);
    CC_CHECK(addr_tgt >= 0);
    const bool gt_pos = laser_info_[addr_tgt].has_gt_positive_lc;
    if (cand_mng) {
      res.id_src = cand_mng->getIntID();
      const int addr_src = lookupNN<int>(res.id_src, assigned_seqs_, 0);
      CC_CHECK(addr_src >= 0);
      const ContourManagerConfig bev_cfg = q_mng->getConfig();
      const Eigen::Isometry2d tf_err = ConstellCorrelation::evalMetricEst(T_est_delta_2d, laser_info_[addr_src].sens_pose,
                                                                          laser_info_[addr_tgt].sens_pose, bev_cfg);
      const double est_trans_norm2d = ConstellCorrelation::getEstSensTF(T_est_delta_2d, bev_cfg).translation().norm();
      const double gt_trans_norm3d = (laser_info_[addr_src].sens_pose.translation() - laser_info_[addr_tgt].sens_pose.translation()).norm();
      printf(" Dist: Est2d: %.2f; GT3d: %.2f\n", est_trans_norm2d, gt_trans_norm3d);
      const double err_vec[3] = {tf_err.translation().x(), tf_err.translation().y(), std::atan2(tf_err(1, 0), tf_err(0, 0))};
      printf(" Error: dx=%f, dy=%f, dtheta=%f\n", err_vec[0], err_vec[1], err_vec[2]);
      std::memcpy(res.est_err, err_vec, sizeof(err_vec));
      if (est_corr >= sim_thres) {
        if (gt_pos && gt_trans_norm3d < 5.0) {
          res.tfpn = PredictionOutcome::TP;
          tp_trans_rmse.addOneErr(err_vec);
          tp_rot_rmse.addOneErr(err_vec + 2);
        } else {
          res.tfpn = PredictionOutcome::FP;
        }
      } else {
        res.tfpn = gt_pos ? PredictionOutcome::FN : PredictionOutcome::TN;
      }
      all_trans_rmse.addOneErr(err_vec);
      all_rot_rmse.addOneErr(err_vec + 2);
    } else {
      res.tfpn = gt_pos ? PredictionOutcome::FN : PredictionOutcome::TN;
    }
    pred_records.push_back(res);
    return res;
  }

  void savePredictionResults(const std::string &sav_path) const {
    std::fstream out(sav_path, std::ios::out);
    if (!out.good()) {
      std::cerr << "Error opening " << sav_path << std::endl;
      return;
    }
    auto tail32 = [](const std::string &s) { return s.substr(s.length() > 32 ? s.length() - 32 : 0); };
    for (const auto &rec : pred_records) {
      const int addr_tgt = lookupNN<int>(rec.id_tgt, assigned_seqs_, 0);
      CC_CHECK(addr_tgt >= 0);
      std::string rep_src = "x";
      out << rec.tfpn << "\t" << rec.id_tgt << "-";
      if (rec.id_src < 0) {
        out << "x";
      } else {
        const int addr_src = lookupNN<int>(rec.id_src, assigned_seqs_, 0);
        CC_CHECK(addr_src >= 0);
        out << rec.id_src;
        rep_src = laser_info_[addr_src].fpath;
      }
      out << "\t" << rec.correlation << "\t" << rec.est_err[0] << "\t" << rec.est_err[1] << "\t" << rec.est_err[2] << "\t"
          << tail32(laser_info_[addr_tgt].fpath) << "\t" << tail32(rep_src) << "\n";
    }
    printf("In outcome file:\nTP is %d\nFP is %d\nTN is %d\nFN is %d\n", (int)PredictionOutcome::TP, (int)PredictionOutcome::FP,
           (int)PredictionOutcome::TN, (int)PredictionOutcome::FN);
    out.close();
    printf("Outcome saved successfully.\n");
  }

  double getTPMeanTrans() const { return tp_trans_rmse.getMean(); }
  double getTPMeanRot() const { return tp_rot_rmse.getMean(); }
  double getTPRMSETrans() const { return tp_trans_rmse.getRMSE(); }
  double getTPRMSERot() const { return tp_rot_rmse.getRMSE(); }

  // evaluator.h:436, src/eval/evaluator.cpp:7-64: the check thresholds from a `name lower upper` text file
  // (config/score_thres_*.cfg): the five integer gates and correlation / area_perc / neg_est_dist; `#` starts a comment,
  // unknown names are skipped, a name is echoed as it is read
  static void loadCheckThres(const std::string &fpath, CandidateScoreEnsemble &thres_lb, CandidateScoreEnsemble &thres_ub) {
    std::ifstream in(fpath);
    if (!in.good()) {
      std::cerr << "Error opening thres config file: " << fpath << std::endl;
      return;
    }
    std::string line, name;
    while (std::getline(in, line)) {
      std::istringstream iss(line);
      if (!(iss >> name)) continue;
      std::cout << name << std::endl;
      if (name[0] == '#') continue;
      if (name == "i_ovlp_sum")
        iss >> thres_lb.sim_constell.i_ovlp_sum >> thres_ub.sim_constell.i_ovlp_sum;
      else if (name == "i_ovlp_max_one")
        iss >> thres_lb.sim_constell.i_ovlp_max_one >> thres_ub.sim_constell.i_ovlp_max_one;
      else if (name == "i_in_ang_rng")
        iss >> thres_lb.sim_constell.i_in_ang_rng >> thres_ub.sim_constell.i_in_ang_rng;
      else if (name == "i_indiv_sim")
        iss >> thres_lb.sim_pair.i_indiv_sim >> thres_ub.sim_pair.i_indiv_sim;
      else if (name == "i_orie_sim")
        iss >> thres_lb.sim_pair.i_orie_sim >> thres_ub.sim_pair.i_orie_sim;
      else if (name == "correlation")
        iss >> thres_lb.sim_post.correlation >> thres_ub.sim_post.correlation;
      else if (name == "area_perc")
        iss >> thres_lb.sim_post.area_perc >> thres_ub.sim_post.area_perc;
      else if (name == "neg_est_dist")
        iss >> thres_lb.sim_post.neg_est_dist >> thres_ub.sim_post.neg_est_dist;
    }
  }
};
