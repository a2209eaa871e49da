// Host mirror of the reference's include/cont2/contour.h + contour_mng.h public surface, on top of the C-ABI
// (include/cont2_amd.h).  Same names, argument meaning and error behaviour (CHECK -> abort) as used by
// test/batch_bin_test.cpp and include/eval/evaluator.h; the work itself runs in the HIP kernels.
#pragma once
#include <fstream>
#include <iostream>
#include <array>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/cont2_amd.h"
#include "../compat/compat.h"

typedef float KeyFloatType;
const int RET_KEY_DIM = CC_KEY_DIM;

// contour_mng.h:39-90
template <size_t sz>
struct ArrayAsKey {
  enum { SizeAtCompileTime = sz };
  KeyFloatType array[sz]{};
  KeyFloatType *data() { return array; }
  KeyFloatType &operator()(size_t i) { return array[i]; }
  const KeyFloatType &operator()(size_t i) const { return array[i]; }
  KeyFloatType &operator[](size_t i) { return array[i]; }
  const KeyFloatType &operator[](size_t i) const { return array[i]; }
  size_t size() const { return sz; }
  KeyFloatType sum() const {
    KeyFloatType ret(0);
    for (const auto &dat : array) ret += dat;
    return ret;
  }
};
using RetrievalKey = ArrayAsKey<RET_KEY_DIM>;

// contour.h:40-45
struct ContourSimThresConfig {
  float ta_cell_cnt = 6, tp_cell_cnt = 0.2;
  float tp_eigval = 0.2;
  float ta_h_bar = 0.3;
  float ta_rcom = 0.4, tp_rcom = 0.25;
};

// contour_mng.h:92-110
struct ContourManagerConfig {
  std::vector<float> lv_grads_;
  float reso_row_ = 1.0f, reso_col_ = 1.0f;
  int n_row_ = 150, n_col_ = 150;
  float lidar_height_ = 2.0f;
  float blind_sq_ = 9.0f;
  int min_cont_key_cnt_ = 9;
  int min_cont_cell_cnt_ = 3;
  int piv_firsts_ = 6;
  int dist_firsts_ = 10;
  float roi_radius_ = 10.0f;
};

// contour_mng.h:121-219 (the three score unions; same member names)
union ScoreConstellSim {
  enum { SizeAtCompileTime = 3 };
  int data[SizeAtCompileTime]{};
  struct {
    int i_ovlp_sum, i_ovlp_max_one, i_in_ang_rng;
  };
  const int &overall() const { return i_in_ang_rng; }  // the value the stage is judged by
  int cnt() const { return i_in_ang_rng; }
  void print() const { printf("%d, %d, %d;", i_ovlp_sum, i_ovlp_max_one, i_in_ang_rng); }
  bool strictSmaller(const ScoreConstellSim &b) const {
    for (int i = 0; i < SizeAtCompileTime; i++)
      if (!(data[i] < b.data[i])) return false;
    return true;
  }
};
union ScorePairwiseSim {
  enum { SizeAtCompileTime = 2 };
  int data[SizeAtCompileTime]{};
  struct {
    int i_indiv_sim, i_orie_sim;
  };
  const int &overall() const { return i_orie_sim; }
  int cnt() const { return i_orie_sim; }
  void print() const { printf("%d, %d;", i_indiv_sim, i_orie_sim); }
  bool strictSmaller(const ScorePairwiseSim &b) const {
    for (int i = 0; i < SizeAtCompileTime; i++)
      if (!(data[i] < b.data[i])) return false;
    return true;
  }
};
union ScorePostProc {
  enum { SizeAtCompileTime = 3 };
  float data[SizeAtCompileTime]{};
  struct {
    float correlation, area_perc, neg_est_dist;
  };
  const float &overall() const { return correlation; }
  void print() const { printf("%6f, %6f%%, %6fm;", correlation, 100 * area_perc, neg_est_dist); }
  bool strictSmaller(const ScorePostProc &b) const {
    for (int i = 0; i < SizeAtCompileTime; i++)
      if (!(data[i] < b.data[i])) return false;
    return true;
  }
};

// contour_mng.h:221-240: one matched contour pair (the "hint" of CandidateManager::checkCandWithHint when it is an anchor)
struct ConstellationPair {
  int8_t level, seq_src, seq_tgt;
  ConstellationPair(int8_t l, int8_t s, int8_t t) : level(l), seq_src(s), seq_tgt(t) {}
  bool operator<(const ConstellationPair &a) const {
    if (level != a.level) return level < a.level;
    if (seq_src != a.seq_src) return seq_src < a.seq_src;
    return seq_tgt < a.seq_tgt;
  }
  bool operator==(const ConstellationPair &a) const { return level == a.level && seq_src == a.seq_src && seq_tgt == a.seq_tgt; }
};

// contour.h:97-119, read-only view over one cc_contour_t
struct ContourView {
  int16_t level_, poi_[2], cell_cnt_;
  float pos_mean_[2], pos_cov_[4], eig_vals_[2], eig_vecs_[4], eccen_, vol3_mean_, com_[2];
  bool ecc_feat_, com_feat_;
};

namespace cc_host {
inline cc_manager_cfg_t to_c(const ContourManagerConfig &c) {
  cc_manager_cfg_t m;
  cc_default_manager_cfg(&m);
  CC_CHECK(c.lv_grads_.size() == CC_NLEV);  // this build handles the shipped 6-level configs
  for (int i = 0; i < CC_NLEV; i++) m.lv_grads[i] = c.lv_grads_[i];
  m.reso_row = c.reso_row_;
  m.reso_col = c.reso_col_;
  m.n_row = c.n_row_;
  m.n_col = c.n_col_;
  m.lidar_height = c.lidar_height_;
  m.blind_sq = c.blind_sq_;
  m.min_cont_key_cnt = c.min_cont_key_cnt_;
  m.min_cont_cell_cnt = c.min_cont_cell_cnt_;
  m.piv_firsts = c.piv_firsts_;
  m.dist_firsts = c.dist_firsts_;
  m.roi_radius = c.roi_radius_;
  return m;
}
// one device context per process and config (the reference has no such object: created lazily).  The device is HIP
// device 0 unless the environment names another one (CC_DEVICE=<n>: one process per GPU in a multi-GPU deployment).
inline int device_id() {
  const char *e = getenv("CC_DEVICE");
  return e ? atoi(e) : 0;
}
// once per process: the device runtime, the code object and the streams a per-scan driver's context and database will use
// (cc_runtime_init) -- called by the constructors of ContourDB and of the evaluator mirror, i.e. before a driver's loop
inline void runtime_warm() {
  static std::once_flag once;
  std::call_once(once, [] {
    if (getenv("CC_NO_RUNTIME_INIT")) return;
    (void)cc_runtime_init(device_id(), 8);  // no device (the evaluator's bookkeeping alone needs none): whoever uses one reports it
  });
}
inline cc_ctx *context(const cc_manager_cfg_t &m) {
  static std::map<std::string, cc_ctx *> pool;
  static std::mutex pool_mu;  // drivers with several worker threads build ContourManagers of one configuration concurrently
  std::lock_guard<std::mutex> lk(pool_mu);
  std::string key((const char *)&m, sizeof(m));
  auto it = pool.find(key);
  if (it != pool.end()) return it->second;
  cc_ctx *c = nullptr;
  if (cc_create(device_id(), &m, 8, &c) != CC_OK) {
    fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
    abort();
  }
  pool[key] = c;
  return c;
}
// What a driver's scan source knows about the scans AFTER the one the driver holds: their device handles (ingested ahead of
// time) and time stamps, in the order the driver will ask for them.  The evaluator mirror publishes its read-ahead here;
// ContourDB reads it to append those scans and queue their queries early (contour_db.h: "read-ahead of the database").
// A source that changes its mind (a jump in the scan list) calls invalidate(): every database that has worked ahead drops
// that work first (the scans it refers to are about to be released).
struct ScanLookahead {
  struct Entry {
    cc_scan *scan;
    double ts;
    const void *source;  // who published it (an evaluator): a database follows ONE source, the one its last scan came from
  };
  std::mutex mu;
  std::deque<Entry> upcoming;                        // under mu; the entries of one source are in that source's order
  std::map<int, std::function<void(const void *)>> on_invalidate;  // registered by the databases (driver thread only); argument: the source
  int next_token = 0;
  int subscribe(std::function<void(const void *)> f) {
    on_invalidate[next_token] = std::move(f);
    return next_token++;
  }
  void unsubscribe(int token) {
    on_invalidate.erase(token);
    on_release.erase(token);
  }
  // A scan handle is about to be released (its ContourManager dies): a database whose queued work refers to the handle drops
  // that work -- the address may be handed out again for a later scan, and a queued answer must never be matched by address
  // against a different scan (driver thread only, like invalidate()).
  std::map<int, std::function<void(cc_scan *)>> on_release;
  void subscribeRelease(int token, std::function<void(cc_scan *)> f) { on_release[token] = std::move(f); }
  void released(cc_scan *scan) {
    if (!scan) return;
    for (auto &f : on_release) f.second(scan);
    popFront(scan);
  }
  void push(cc_scan *scan, double ts, const void *source) {
    std::lock_guard<std::mutex> lk(mu);
    upcoming.push_back({scan, ts, source});
  }
  void popFront(cc_scan *scan) {  // the driver has taken the scan (the oldest one of its source)
    std::lock_guard<std::mutex> lk(mu);
    for (auto it = upcoming.begin(); it != upcoming.end(); ++it)
      if (it->scan == scan) {
        upcoming.erase(it);
        return;
      }
  }
  std::vector<Entry> snapshot(const void *source) {
    std::lock_guard<std::mutex> lk(mu);
    std::vector<Entry> v;
    for (const Entry &e : upcoming)
      if (e.source == source) v.push_back(e);
    return v;
  }
  void invalidate(const void *source) {  // driver thread: `source` is about to release what it has published
    for (auto &f : on_invalidate) f.second(source);
    std::lock_guard<std::mutex> lk(mu);
    for (auto it = upcoming.begin(); it != upcoming.end();) it = it->source == source ? upcoming.erase(it) : it + 1;
  }
};
inline ScanLookahead &lookahead() {
  static ScanLookahead h;
  return h;
}
}  // namespace cc_host

// contour_mng.h:414-1314 (public surface used by the drivers)
class ContourManager {
  static constexpr float VAL_ABS_INF_ = 1e3f;  // contour_mng.h:418: the "no point" height is -VAL_ABS_INF_
  const ContourManagerConfig cfg_;
  cc_manager_cfg_t ccfg_;
  int int_id_;
  std::string str_id_;
  // The scan on the device (cc_scan, include/cont2_amd.h): makeBEV writes the points straight into the context's pinned
  // staging buffer and queues rasterisation + contours + keys + BCIs; the descriptor stays on the device for
  // ContourDB::queryRangedKNN / addScan, its host copy (what the getters below read) is fetched on first use.
  cc_scan *scan_ = nullptr;
  bool want_images_ = false;
  const void *source_ = nullptr;  // mirror-only: the scan source that handed the scan out (ContourDB's read-ahead follows it)

 public:
  const void *scanSource() const { return source_; }
  void setScanSource(const void *src) { source_ = src; }
  explicit ContourManager(const ContourManagerConfig &config, int int_id) : cfg_(config), int_id_(int_id) {
    CC_CHECK(cfg_.n_col_ % 2 == 0);
    CC_CHECK(cfg_.n_row_ % 2 == 0);
    ccfg_ = cc_host::to_c(cfg_);
  }

  ~ContourManager() {
    cc_host::lookahead().released(scan_);  // (nobody may keep matching queued work against this address)
    cc_scan_release(scan_);
  }
  ContourManager(const ContourManager &) = delete;
  ContourManager &operator=(const ContourManager &) = delete;

  // Whether a scan's max-height image is brought back from the device (90 KB extra per scan).  On when the reference
  // would write its SAVE_MID_FILE artefacts (CMakeLists.txt:17), or switched on by the caller before makeBEV().
  static bool &keepImages() {
#if defined(SAVE_MID_FILE) && SAVE_MID_FILE
    static bool keep = true;
#else
    static bool keep = false;
#endif
    return keep;
  }

  // contour_mng.h:505: x,y,z of every point (KITTI layout, intensity unused) -> the device; the kernels are queued here
  // already (nothing of them is observable before makeContoursRecurs() in the reference either)
  template <typename PointType>
  void makeBEV(typename pcl::PointCloud<PointType>::ConstPtr &ptr_gapc, std::string str_id = "") {
    CC_CHECK(ptr_gapc);
    CC_CHECK(ptr_gapc->size() > 10);
    CC_CHECK(!scan_);
    cc_ctx *ctx = cc_host::context(ccfg_);
    const size_t n = ptr_gapc->size();
    float *dst = cc_stage_points(ctx, (int64_t)n);
    if (!dst) die();
    for (size_t i = 0; i < n; i++) {
      dst[4 * i] = ptr_gapc->points[i].x;
      dst[4 * i + 1] = ptr_gapc->points[i].y;
      dst[4 * i + 2] = ptr_gapc->points[i].z;
      dst[4 * i + 3] = 0.f;
    }
    want_images_ = keepImages();
    if (cc_scan_ingest(ctx, dst, (int64_t)n, want_images_ ? 1 : 0, &scan_) != CC_OK) die();
    str_id_ = !str_id.empty() ? std::move(str_id) : std::to_string(ptr_gapc->header.stamp);
  }

  // Mirror-only: the evaluator's .bin reader (tools/pointcloud_util.h:9-47: at most 1 000 000 floats, x y z i records,
  // intensity dropped) without the intermediate cloud.  A KITTI record IS a staging record (the kernels never read the
  // fourth float), so the file is read straight into the context's pinned buffer.  Returns the number of points.
  size_t makeBEVFromKittiBin(FILE *f, std::string str_id) {
    CC_CHECK(f);
    CC_CHECK(!scan_);
    cc_ctx *ctx = cc_host::context(ccfg_);
    const size_t cap = 1000000 / 4;
    float *dst = cc_stage_points(ctx, (int64_t)cap);
    if (!dst) die();
    const size_t n = fread(dst, 4 * sizeof(float), cap, f);
    CC_CHECK(n > 10);
    want_images_ = keepImages();
    if (cc_scan_ingest(ctx, dst, (int64_t)n, want_images_ ? 1 : 0, &scan_) != CC_OK) die();
    str_id_ = std::move(str_id);
    return n;
  }

  // Mirror-only: n KITTI records already sitting in one of the context's staging buffers (cc_stage_points_slot), e.g. read
  // there ahead of time by the evaluator's prefetch thread.
  void makeBEVFromStaged(const float *staged, size_t n, std::string str_id) {
    CC_CHECK(staged);
    CC_CHECK(n > 10);
    CC_CHECK(!scan_);
    cc_ctx *ctx = cc_host::context(ccfg_);
    want_images_ = keepImages();
    if (cc_scan_ingest(ctx, staged, (int64_t)n, want_images_ ? 1 : 0, &scan_) != CC_OK) die();
    str_id_ = std::move(str_id);
  }
  // Mirror-only: a scan that was ingested ahead of time (cc_scan_ingest on the evaluator's helper thread, with
  // want_bev = keepImages() at that moment); the handle is owned from here on.
  void adoptIngested(cc_scan *scan, bool with_images, std::string str_id) {
    CC_CHECK(scan);
    CC_CHECK(!scan_);
    scan_ = scan;
    want_images_ = with_images;
    str_id_ = std::move(str_id);
  }
  // the context the scans of this configuration are ingested on (one per ContourManagerConfig)
  static cc_ctx *contextOf(const ContourManagerConfig &config) { return cc_host::context(cc_host::to_c(config)); }

  // contour_mng.h:588: the work was queued by makeBEV; results are waited for where they are read
  void makeContoursRecurs() { CC_CHECK(scan_); }
  void clearImage() {}  // the dense image is never kept here (see bev_cells_)
  // contour_mng.h:562-571 rebuilds bev_ from the pillar list after clearImage(); here getBevImage() reads the copy that came
  // back from the device with the scan (keepImages() at makeBEV time), which clearImage() does not drop: nothing to rebuild
  void resumeImage() {}

  // contour_mng.h:573-586: the dense max-height image, -VAL_ABS_INF_ where no point fell
  cc_host::Image<float> getBevImage() const {
    cc_host::Image<float> img(cfg_.n_row_, cfg_.n_col_, -VAL_ABS_INF_);
    if (want_images_) {
      const float *bev = nullptr;
      if (cc_scan_bev(scan_, &bev) != CC_OK) die();
      std::memcpy(img.data.data(), bev, sizeof(float) * img.data.size());
    }
    return img;
  }
  // contour_mng.h:1041-1049: cv::threshold(bev, lv_grads_[level], THRESH_TOZERO) then cv::normalize(0, 255, NORM_MINMAX,
  // CV_8U).  Restated from OpenCV's documented semantics (OpenCV is absent here): TOZERO keeps src where src > thresh;
  // MINMAX maps [min, max] of that image to [0, 255] with scale = 255 / (max - min) (0 when max - min <= DBL_EPSILON),
  // shift = -min * scale, evaluated per pixel in f32 and rounded half-to-even with saturation (convertTo to CV_8U).
  cc_host::Image<unsigned char> getContourImage(int level) const {
    CC_CHECK(level >= 0 && level < (int)cfg_.lv_grads_.size());
    cc_host::Image<float> m = getBevImage();
    const float thr = cfg_.lv_grads_[level];
    double mn = 0, mx = 0;
    bool first = true;
    for (float &v : m.data) {
      v = v > thr ? v : 0.f;
      if (first || v < mn) mn = v;
      if (first || v > mx) mx = v;
      first = false;
    }
    const double scale = 255.0 * ((mx - mn) > 2.220446049250313e-16 ? 1.0 / (mx - mn) : 0.0), shift = 0.0 - mn * scale;
    const float a = (float)scale, b = (float)shift;
    cc_host::Image<unsigned char> out(m.rows, m.cols, 0);
    for (size_t i = 0; i < m.data.size(); i++) {
      const long r = lrintf(m.data[i] * a + b);
      out.data[i] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
    return out;
  }
  // contour_mng.cpp (saveContourImage): cv::imwrite(fpath, getContourImage(level))
  void saveContourImage(const std::string &fpath, int level) const {
    if (!cc_host::write_png_gray8(fpath, getContourImage(level))) std::cerr << "Error opening " << fpath << std::endl;
  }
  // contour_mng.h:1286-1311: the level images of two scans, cm1's on the top row and cm2's below, one column per level,
  // separated by one white pixel
  static void saveMatchedPairImg(const std::string &fpath, const ContourManager &cm1, const ContourManager &cm2) {
    const ContourManagerConfig config = cm2.getConfig();
    const int n_lev = (int)config.lv_grads_.size();
    cc_host::Image<unsigned char> output(config.n_row_ * 2 + 1, (config.n_col_ + 1) * n_lev, 255);
    for (int i = 0; i < n_lev; i++) {
      cm1.getContourImage(i).copyTo(output, i * config.n_col_ + i, 0);
      cm2.getContourImage(i).copyTo(output, i * config.n_col_ + i, config.n_row_ + 1);
    }
    if (!cc_host::write_png_gray8(fpath, output)) std::cerr << "Error opening " << fpath << std::endl;
  }

  static void die() {
    fprintf(stderr, "cont2_amd: %s\n", cc_last_error());
    abort();
  }
  // the host copy of everything the reference's ContourManager keeps after makeContoursRecurs() + clearImage()
  const cc_scan_desc_t &desc() const {
    CC_CHECK(scan_);
    const cc_scan_desc_t *d = nullptr;
    if (cc_scan_desc(scan_, &d) != CC_OK) die();  // includes CC_ECAPACITY: the reference has no capacities to exceed
    return *d;
  }
  cc_scan *scanHandle() const { return scan_; }
  const cc_manager_cfg_t &ccfg() const { return ccfg_; }
  std::vector<RetrievalKey> getLevRetrievalKey(int level) const {
    std::vector<RetrievalKey> r(cfg_.piv_firsts_);
    for (int s = 0; s < cfg_.piv_firsts_; s++) std::memcpy(r[s].array, desc().keys[level][s], sizeof(float) * RET_KEY_DIM);
    return r;
  }
  RetrievalKey getRetrievalKey(int level, int seq) const { return getLevRetrievalKey(level)[seq]; }
  // contour_mng.h:1066 (there: a const reference to the stored vector; here the views are rebuilt from the descriptor)
  std::vector<std::shared_ptr<ContourView>> getLevContours(int level) const {
    const cc_scan_desc_t &d = desc();
    std::vector<std::shared_ptr<ContourView>> v(d.n_stored[level]);
    for (int j = 0; j < d.n_stored[level]; j++) {
      const cc_contour_t &c = d.cont[level][j];
      v[j] = std::make_shared<ContourView>();
      ContourView &o = *v[j];
      o.level_ = c.level;
      o.poi_[0] = c.poi[0];
      o.poi_[1] = c.poi[1];
      o.cell_cnt_ = c.cell_cnt;
      std::memcpy(o.pos_mean_, c.pos_mean, 8);
      std::memcpy(o.pos_cov_, c.pos_cov, 16);
      std::memcpy(o.eig_vals_, c.eig_vals, 8);
      std::memcpy(o.eig_vecs_, c.eig_vecs, 16);
      o.eccen_ = c.eccen;
      o.vol3_mean_ = c.vol3_mean;
      std::memcpy(o.com_, c.com, 8);
      o.ecc_feat_ = c.ecc_feat;
      o.com_feat_ = c.com_feat;
    }
    return v;
  }
  int getLevTotalPix(int level) const { return desc().layer_cell_cnt[level]; }
  const cc_bci_t &getBCI(int level, int seq) const { return desc().bcis[level][seq]; }
  // contour_mng.h:1079: the level's BCIs, one per anchor key that exists (valid key <=> a non-zero retrieval key)
  std::vector<cc_bci_t> getLevBCI(int level) const {
    std::vector<cc_bci_t> v;
    const cc_scan_desc_t &d = desc();
    for (int s = 0; s < CC_NPIV; s++) {
      float sum = 0.f;
      for (int k = 0; k < CC_KEY_DIM; k++) sum += d.keys[level][s][k];
      if (sum != 0.f) v.push_back(d.bcis[level][s]);
    }
    return v;
  }
  float getAreaPerc(const int8_t &lev, const int8_t &seq) const {
    return desc().cont[lev][seq].cell_cnt * 1.0f / desc().layer_cell_cnt[lev];
  }
  // contour_mng.cpp:7-47 (saveContours) / contour_mng.h:915-918: the 20-column text dump scripts/plot_contours.py reads --
  // level, cell_cnt, pos_mean(2), pos_cov(4, column-major), eig_vals(2), eig_vecs(4), eccen, vol3_mean, com(2), ecc_feat,
  // com_feat; one row per stored contour, levels in order, between "DATA_START" and "DATA_END"
  void saveContours(const std::string &fpath) const {
    std::fstream out(fpath, std::ios::out);
    if (!out.good()) {
      std::cerr << "Error opening " << fpath << std::endl;
      return;
    }
    printf("Writing results to file \"%s\" ...", fpath.c_str());
    const cc_scan_desc_t &d = desc();
    out << "\nDATA_START\n";
    for (int l = 0; l < CC_NLEV; l++) {
      for (int j = 0; j < d.n_stored[l]; j++) {
        const cc_contour_t &c = d.cont[l][j];
        out << c.level << '\t' << c.cell_cnt << '\t' << c.pos_mean[0] << '\t' << c.pos_mean[1] << '\t';
        for (int i = 0; i < 4; i++) out << c.pos_cov[i] << '\t';
        out << c.eig_vals[0] << '\t' << c.eig_vals[1] << '\t';
        for (int i = 0; i < 4; i++) out << c.eig_vecs[i] << '\t';
        out << c.eccen << '\t' << c.vol3_mean << '\t' << c.com[0] << '\t' << c.com[1] << '\t' << int(c.ecc_feat) << '\t'
            << int(c.com_feat) << '\t' << '\n';
      }
    }
    out << "DATA_END\n";
    out.close();
    printf("Writing results finished.\n");
  }
  std::string getStrID() const { return str_id_; }
  int getIntID() const { return int_id_; }
  const ContourManagerConfig &getConfig() const { return cfg_; }
};
