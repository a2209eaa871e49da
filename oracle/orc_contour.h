// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// CPU restatement of include/cont2/contour.h, include/cont2/contour_mng.h and
// src/cont2/contour_mng.cpp:274-353 of the reference (file:line cited per function).
// Single-threaded, same data structures (std::map pillars, recursive threshold + CCL,
// binary-searched pixel list), f32/f64 types and operation order as the reference.
#pragma once
#include <algorithm>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "../include/cont2_amd.h"
#include "orc_math.h"

namespace orc {

// tools/algos.h:13-20
template <typename T>
inline bool diff_perc(const T &num1, const T &num2, const T &perc) {
  return std::abs((num1 - num2) / std::max(num1, num2)) > perc;
}
template <typename T>
inline bool diff_delt(const T &num1, const T &num2, const T &delta) {
  return std::abs(num1 - num2) > delta;
}
// tools/algos.h:49-51
template <typename T>
inline void clampAng(T &ang) {
  ang = ang - std::floor((ang + M_PI) / (2 * M_PI)) * 2 * M_PI;
}
// tools/algos.h:54-56
template <typename T>
inline T gaussPDF(const T &x, const T &mean, const T &sd) {
  return std::exp(-0.5 * ((x - mean) / sd) * ((x - mean) / sd)) / std::sqrt(2 * M_PI * sd * sd);
}

// contour.h:32-45
struct ContourViewStatConfig {
  int16_t min_cell_cov = 4;
  float point_sigma = 1.0;
  float com_bias_thres = 0.5;
};
struct ContourSimThresConfig {
  float ta_cell_cnt = 6, tp_cell_cnt = 0.2;
  float tp_eigval = 0.2;
  float ta_h_bar = 0.3;
  float ta_rcom = 0.4, tp_rcom = 0.25;
};

// contour.h:48-95
struct RunningStatRecorder {
  int16_t cell_cnt_{};
  V2D cell_pos_sum_;
  M2D cell_pos_tss_;
  float cell_vol3_{};
  V2D cell_vol3_torq_;
  void runningStatsF(float curr_row, float curr_col, float height) {
    cell_cnt_ += 1;
    V2D v_rc(curr_row, curr_col);
    cell_pos_sum_.x += v_rc.x;
    cell_pos_sum_.y += v_rc.y;
    cell_pos_tss_.a[0][0] += v_rc.x * v_rc.x;
    cell_pos_tss_.a[0][1] += v_rc.x * v_rc.y;
    cell_pos_tss_.a[1][0] += v_rc.y * v_rc.x;
    cell_pos_tss_.a[1][1] += v_rc.y * v_rc.y;
    cell_vol3_ += height;
    cell_vol3_torq_.x += height * v_rc.x;  // float * double -> double
    cell_vol3_torq_.y += height * v_rc.y;
  }
};

// contour.h:97-380
struct ContourView {
  int16_t level_;
  int16_t poi_[2];
  int16_t cell_cnt_{};
  V2F pos_mean_;
  M2F pos_cov_;
  V2F eig_vals_;
  M2F eig_vecs_;
  float eccen_{};
  float vol3_mean_{};
  V2F com_;
  bool ecc_feat_ = false;
  bool com_feat_ = false;

  ContourView(int16_t level, int16_t poi_r, int16_t poi_c) : level_(level) {
    poi_[0] = poi_r;
    poi_[1] = poi_c;
  }

  // contour.h:142-255
  void calcStatVals(const RunningStatRecorder &rec, const ContourViewStatConfig &cfg) {
    cell_cnt_ = rec.cell_cnt_;
    float cntf = (float)cell_cnt_;
    pos_mean_ = V2F((float)rec.cell_pos_sum_.x / cntf, (float)rec.cell_pos_sum_.y / cntf);
    vol3_mean_ = rec.cell_vol3_ / cell_cnt_;
    com_ = V2F((float)rec.cell_vol3_torq_.x / rec.cell_vol3_, (float)rec.cell_vol3_torq_.y / rec.cell_vol3_);
    if (cell_cnt_ < cfg.min_cell_cov) {
      pos_cov_ = M2F::Identity();
      pos_cov_.a[0][0] = 1.f * cfg.point_sigma * cfg.point_sigma;
      pos_cov_.a[1][1] = 1.f * cfg.point_sigma * cfg.point_sigma;
      pos_cov_.a[0][1] = 0.f * cfg.point_sigma * cfg.point_sigma;
      pos_cov_.a[1][0] = 0.f * cfg.point_sigma * cfg.point_sigma;
      eig_vals_ = V2F(cfg.point_sigma, cfg.point_sigma);
      eig_vecs_ = M2F::Identity();
      ecc_feat_ = false;
      com_feat_ = false;
    } else {
      float pm[2] = {pos_mean_.x, pos_mean_.y};
      float denom = (float)(cell_cnt_ - 1);
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
          pos_cov_.a[i][j] = ((float)rec.cell_pos_tss_.a[i][j] - (pm[i] * pm[j]) * cntf) / denom;
      // selfadjointView<Upper>() -> dense symmetric from the upper triangle
      M2F sym = pos_cov_;
      sym.a[1][0] = sym.a[0][1];
      float ev[2];
      selfAdjointEigen2f(sym, ev, eig_vecs_);
      eig_vals_ = V2F(ev[0], ev[1]);
      if (eig_vals_.x < cfg.point_sigma) eig_vals_.x = cfg.point_sigma;
      if (eig_vals_.y < cfg.point_sigma) eig_vals_.y = cfg.point_sigma;
      eccen_ = std::sqrt(eig_vals_.y * eig_vals_.y - eig_vals_.x * eig_vals_.x) / eig_vals_.y;
      ecc_feat_ = cell_cnt_ > 5 && diff_perc<float>(eig_vals_.x, eig_vals_.y, 0.2f) && eig_vals_.y > 2.5f;
      com_feat_ = (com_ - pos_mean_).norm() > cfg.com_bias_thres;
    }
  }

  // contour.h:278-329
  static bool checkSim(const ContourView &cont_src, const ContourView &cont_tgt, const ContourSimThresConfig &simthres) {
    if (diff_perc<float>(cont_src.cell_cnt_, cont_tgt.cell_cnt_, simthres.tp_cell_cnt) &&
        diff_delt<float>(cont_src.cell_cnt_, cont_tgt.cell_cnt_, simthres.ta_cell_cnt))
      return false;
    if (std::max(cont_src.eig_vals_.y, cont_tgt.eig_vals_.y) > 2.0 &&
        diff_perc<float>(std::sqrt(cont_src.eig_vals_.y), std::sqrt(cont_tgt.eig_vals_.y), simthres.tp_eigval))
      return false;
    if (std::max(cont_src.eig_vals_.x, cont_tgt.eig_vals_.x) > 2.0 &&
        diff_perc<float>(std::sqrt(cont_src.eig_vals_.x), std::sqrt(cont_tgt.eig_vals_.x), simthres.tp_eigval))
      return false;
    if (std::max(cont_src.cell_cnt_, cont_tgt.cell_cnt_) > 15 &&
        diff_delt<float>(cont_src.vol3_mean_, cont_tgt.vol3_mean_, simthres.ta_h_bar))
      return false;
    const float com_r1 = (cont_src.com_ - cont_src.pos_mean_).norm();
    const float com_r2 = (cont_tgt.com_ - cont_tgt.pos_mean_).norm();
    if (diff_delt<float>(com_r1, com_r2, simthres.ta_rcom) && diff_perc<float>(com_r1, com_r2, simthres.tp_rcom))
      return false;
    return true;
  }

  // contour.h:376-378: eig_vecs_ * eig_vals_.asDiagonal() * eig_vecs_.transpose()
  M2F getManualCov() const {
    M2F vd, r;
    float ev[2] = {eig_vals_.x, eig_vals_.y};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) vd.a[i][j] = eig_vecs_.a[i][j] * ev[j];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) r.a[i][j] = vd.a[i][0] * eig_vecs_.a[j][0] + vd.a[i][1] * eig_vecs_.a[j][1];
    return r;
  }
};

// contour_mng.h:36-90
using KeyFloatType = float;
const int RET_KEY_DIM = 10;
struct RetrievalKey {
  KeyFloatType array[RET_KEY_DIM]{};
  KeyFloatType &operator()(size_t i) { return array[i]; }
  const KeyFloatType &operator()(size_t i) const { return array[i]; }
  KeyFloatType &operator[](size_t i) { return array[i]; }
  const KeyFloatType &operator[](size_t i) const { return array[i]; }
  void setZero() { std::fill(array, array + RET_KEY_DIM, KeyFloatType(0)); }
  KeyFloatType sum() const {
    KeyFloatType ret(0);
    for (const auto &dat : array) ret += dat;
    return ret;
  }
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

// contour_mng.h:112-116
const int16_t BITS_PER_LAYER = 64;
const int8_t DIST_BIN_LAYERS[] = {1, 2, 3, 4};
const float LAYER_AREA_WEIGHTS[] = {0.3, 0.3, 0.3, 0.1};
const int16_t NUM_BIN_KEY_LAYER = sizeof(DIST_BIN_LAYERS) / sizeof(int8_t);

// contour_mng.h:121-219 (unions flattened to structs; only the fields are used)
struct ScoreConstellSim {
  int i_ovlp_sum = 0, i_ovlp_max_one = 0, i_in_ang_rng = 0;
  int overall() const { return i_in_ang_rng; }
  int cnt() const { return i_in_ang_rng; }
};
struct ScorePairwiseSim {
  int i_indiv_sim = 0, i_orie_sim = 0;
  int overall() const { return i_orie_sim; }
  int cnt() const { return i_orie_sim; }
};
struct ScorePostProc {
  float correlation = 0, area_perc = 0, neg_est_dist = 0;
};

// contour_mng.h:221-240
struct ConstellationPair {
  int8_t level, seq_src, seq_tgt;
  ConstellationPair(int8_t l, int8_t s, int8_t t) : level(l), seq_src(s), seq_tgt(t) {}
  bool operator<(const ConstellationPair &a) const {
    return level < a.level || (level == a.level && seq_src < a.seq_src) ||
           (level == a.level && seq_src == a.seq_src && seq_tgt < a.seq_tgt);
  }
};

// contour_mng.h:243-389
struct BCI {
  struct RelativePoint {
    int8_t level, seq;
    int16_t bit_pos;
    float r, theta;
    RelativePoint(int8_t l, int8_t a, int16_t b, float f1, float f2) : level(l), seq(a), bit_pos(b), r(f1), theta(f2) {}
  };
  struct DistSimPair {
    float orie_diff;
    int8_t seq_src, seq_tgt, level;
    DistSimPair(int8_t l, int8_t s, int8_t t, float o) : orie_diff(o), seq_src(s), seq_tgt(t), level(l) {}
  };
  std::bitset<BITS_PER_LAYER * NUM_BIN_KEY_LAYER> dist_bin_;
  std::vector<RelativePoint> nei_pts_;
  std::vector<uint16_t> nei_idx_segs_;
  int8_t piv_seq_, level_;
  explicit BCI(int8_t seq, int8_t lev) : dist_bin_(0), piv_seq_(seq), level_(lev) {}

  // contour_mng.h:288-388
  static ScoreConstellSim checkConstellSim(const BCI &src, const BCI &tgt, const ScoreConstellSim &lb,
                                           std::vector<ConstellationPair> &constell_res) {
    std::bitset<BITS_PER_LAYER * NUM_BIN_KEY_LAYER> and1, and2, and3;
    and1 = src.dist_bin_ & tgt.dist_bin_;
    and2 = (src.dist_bin_ << 1) & tgt.dist_bin_;
    and3 = (src.dist_bin_ >> 1) & tgt.dist_bin_;
    int ovlp1 = and1.count(), ovlp2 = and2.count(), ovlp3 = and3.count();
    int ovlp_sum = ovlp1 + ovlp2 + ovlp3;
    int max_one = std::max(ovlp1, std::max(ovlp2, ovlp3));
    ScoreConstellSim ret;
    ret.i_ovlp_sum = ovlp_sum;
    ret.i_ovlp_max_one = max_one;
    if (ovlp_sum >= lb.i_ovlp_sum && max_one >= lb.i_ovlp_max_one) {
      std::vector<DistSimPair> potential_pairs;
      int16_t p11 = 0, p12;
      for (int16_t p2 = 0; p2 < (int)tgt.nei_idx_segs_.size() - 1; p2++) {
        while (p11 < (int)src.nei_idx_segs_.size() - 1 &&
               src.nei_pts_[src.nei_idx_segs_[p11]].bit_pos < tgt.nei_pts_[tgt.nei_idx_segs_[p2]].bit_pos - 1) {
          p11++;
        }
        p12 = p11;
        while (p12 < (int)src.nei_idx_segs_.size() - 1 &&
               src.nei_pts_[src.nei_idx_segs_[p12]].bit_pos <= tgt.nei_pts_[tgt.nei_idx_segs_[p2]].bit_pos + 1) {
          p12++;
        }
        for (int i = tgt.nei_idx_segs_[p2]; i < tgt.nei_idx_segs_[p2 + 1]; i++) {
          for (int j = src.nei_idx_segs_[p11]; j < src.nei_idx_segs_[p12]; j++) {
            const BCI::RelativePoint &rp1 = src.nei_pts_[j], &rp2 = tgt.nei_pts_[i];
            potential_pairs.emplace_back(rp1.level, rp1.seq, rp2.seq, rp2.theta - rp1.theta);
          }
        }
      }
      for (auto &x : potential_pairs) clampAng<float>(x.orie_diff);
      std::sort(potential_pairs.begin(), potential_pairs.end(),
                [&](const DistSimPair &a, const DistSimPair &b) { return a.orie_diff < b.orie_diff; });
      const float angular_range = M_PI / 16;
      int longest_in_range_beg = 0, longest_in_range = 1, pot_sz = potential_pairs.size(), p1 = 0, p2 = 0;
      while (p1 < pot_sz) {
        if (potential_pairs[p2 % pot_sz].orie_diff - potential_pairs[p1].orie_diff + 2 * M_PI * int(p2 / pot_sz) >
            angular_range)
          p1++;
        else {
          if (p2 - p1 + 1 > longest_in_range) {
            longest_in_range = p2 - p1 + 1;
            longest_in_range_beg = p1;
          }
          p2++;
        }
      }
      ret.i_in_ang_rng = longest_in_range;
      if (longest_in_range < lb.i_in_ang_rng) return ret;
      constell_res.clear();
      constell_res.reserve(longest_in_range + 1);
      for (int i = longest_in_range_beg; i < longest_in_range + longest_in_range_beg; i++) {
        constell_res.emplace_back(potential_pairs[i % pot_sz].level, potential_pairs[i % pot_sz].seq_src,
                                  potential_pairs[i % pot_sz].seq_tgt);
      }
      constell_res.emplace_back(src.level_, src.piv_seq_, tgt.piv_seq_);
      return ret;
    } else {
      return ret;
    }
  }
};

// contour_mng.h:392-411
struct Pixelf {
  float row_f, col_f, elev;
  Pixelf(float r, float c, float e) : row_f(r), col_f(c), elev(e) {}
  Pixelf() : row_f(-1), col_f(-1), elev(-1) {}
};

// tools/algos.h:59-68
template <typename T>
std::pair<int, T> search_vec(const std::vector<std::pair<int, T>> &arr, int p1, int p2, const int &tgt) {
  if (p2 < p1) return {-1, T()};
  int mid = (p1 + p2) / 2;
  if (arr[mid].first == tgt)
    return arr[mid];
  else if (arr[mid].first < tgt)
    return search_vec<T>(arr, mid + 1, p2, tgt);
  return search_vec<T>(arr, p1, mid - 1, tgt);
}

struct Rect {
  int x, y, width, height;
};

// cv::connectedComponentsWithStats(img, labels, stats, centroids, 8, CV_32S) on a patch
// (src/cont2/contour_mng.cpp:298).  OpenCV is not available; this restates the documented
// result: 8-connected components, background label 0, stats = {left, top, width, height, area}.
// Label numbering follows the block-based (BBDT / Spaghetti) two-pass algorithms OpenCV uses
// for 8-connectivity: provisional labels are created per 2x2 block in block-raster order,
// unions keep the smaller root, and the flatten pass renumbers roots in increasing order --
// i.e. components are numbered by their first 2x2 block in block-raster order relative to the
// patch origin.  (Unpinned against OpenCV itself: see DESIGN.md.)
inline int connectedComponentsWithStats8(const std::vector<uint8_t> &img, int rows, int cols, std::vector<int> &labels,
                                         std::vector<std::array<int, 5>> &stats) {
  labels.assign((size_t)rows * cols, 0);
  std::vector<int> stack;
  struct Comp {
    int key, l, t, r, b, area;
  };
  std::vector<Comp> comps;
  const int nbc = (cols + 1) / 2;
  int n_tmp = 0;
  for (int r0 = 0; r0 < rows; r0++)
    for (int c0 = 0; c0 < cols; c0++) {
      if (!img[(size_t)r0 * cols + c0] || labels[(size_t)r0 * cols + c0]) continue;
      n_tmp++;
      Comp cp{(r0 / 2) * nbc + (c0 / 2), c0, r0, c0, r0, 0};
      stack.clear();
      stack.push_back(r0 * cols + c0);
      labels[(size_t)r0 * cols + c0] = n_tmp;
      while (!stack.empty()) {
        int p = stack.back();
        stack.pop_back();
        int r = p / cols, c = p % cols;
        cp.area++;
        cp.l = std::min(cp.l, c);
        cp.r = std::max(cp.r, c);
        cp.t = std::min(cp.t, r);
        cp.b = std::max(cp.b, r);
        cp.key = std::min(cp.key, (r / 2) * nbc + (c / 2));
        for (int dr = -1; dr <= 1; dr++)
          for (int dc = -1; dc <= 1; dc++) {
            int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= rows || cc < 0 || cc >= cols) continue;
            size_t q = (size_t)rr * cols + cc;
            if (img[q] && !labels[q]) {
              labels[q] = n_tmp;
              stack.push_back((int)q);
            }
          }
      }
      comps.push_back(cp);
    }
  // renumber by first-block key
  std::vector<int> order(comps.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return comps[a].key < comps[b].key; });
  if (orc::variant().label_shuffle_seed) {  // sensitivity knob (orc_math.h): any other numbering
    unsigned st = orc::variant().label_shuffle_seed * 2654435761u + (unsigned)comps.size();
    for (size_t i = order.size(); i > 1; i--) {
      st = st * 1664525u + 1013904223u;
      std::swap(order[i - 1], order[(st >> 8) % i]);
    }
  }
  std::vector<int> remap(comps.size() + 1, 0);
  stats.assign(comps.size() + 1, {0, 0, 0, 0, 0});
  for (size_t i = 0; i < order.size(); i++) {
    remap[order[i] + 1] = (int)i + 1;
    const Comp &cp = comps[order[i]];
    stats[i + 1] = {cp.l, cp.t, cp.r - cp.l + 1, cp.b - cp.t + 1, cp.area};
  }
  for (auto &l : labels) l = remap[l];
  return (int)comps.size() + 1;
}

// contour_mng.h:414-1314
class ContourManager {
 public:
  const ContourManagerConfig cfg_;
  const ContourViewStatConfig view_stat_cfg_;
  const float VAL_ABS_INF_ = 1e3;
  float x_max_, x_min_, y_max_, y_min_;
  int int_id_;
  std::vector<std::vector<std::shared_ptr<ContourView>>> cont_views_;
  std::vector<std::vector<float>> cont_perc_;
  std::vector<int> layer_cell_cnt_;
  std::vector<std::vector<RetrievalKey>> layer_keys_;
  std::vector<std::vector<BCI>> layer_key_bcis_;
  std::vector<float> bev_;  // n_row x n_col, row-major (cv::Mat1f)
  std::vector<std::pair<int, Pixelf>> bev_pixfs_;
  float max_bin_val_ = -VAL_ABS_INF_, min_bin_val_ = VAL_ABS_INF_;
  // parity extra: pre-sort (insertion order) id of each contour, to build canonical label images
  std::vector<std::vector<std::vector<int>>> cont_cells_;  // [level][insertion idx] -> cell hashes
  std::vector<std::vector<int>> sort_perm_;               // [level][seq] -> insertion idx
  bool keep_cells_ = false;

  float &bev(int r, int c) { return bev_[(size_t)r * cfg_.n_col_ + c]; }
  const float &bev(int r, int c) const { return bev_[(size_t)r * cfg_.n_col_ + c]; }

  // contour_mng.h:448-463
  std::pair<int, int> hashPointToImage(float ptx, float pty) const {
    std::pair<int, int> res{-1, -1};
    float padding = 1e-2;
    if (ptx < x_min_ + padding || ptx > x_max_ - padding || pty < y_min_ + padding || pty > y_max_ - padding ||
        (pty * pty + ptx * ptx) < cfg_.blind_sq_) {
      return res;
    }
    res.first = int(std::floor(ptx / cfg_.reso_row_)) + cfg_.n_row_ / 2;
    res.second = int(std::floor(pty / cfg_.reso_col_)) + cfg_.n_col_ / 2;
    return res;
  }
  // contour_mng.h:468-472
  V2F pointToContRowCol(const V2F &p_in_l) const {
    V2F continuous_rc(p_in_l.x / cfg_.reso_row_ + cfg_.n_row_ / 2 - 0.5f,
                      p_in_l.y / cfg_.reso_col_ + cfg_.n_col_ / 2 - 0.5f);
    return continuous_rc;
  }

  // contour_mng.h:478-498
  explicit ContourManager(const ContourManagerConfig &config, int int_id) : cfg_(config), int_id_(int_id) {
    x_min_ = -(cfg_.n_row_ / 2) * cfg_.reso_row_;
    x_max_ = -x_min_;
    y_min_ = -(cfg_.n_col_ / 2) * cfg_.reso_col_;
    y_max_ = -y_min_;
    bev_.assign((size_t)cfg_.n_row_ * cfg_.n_col_, -VAL_ABS_INF_);
    cont_views_.resize(cfg_.lv_grads_.size());
    cont_perc_.resize(cfg_.lv_grads_.size());
    layer_cell_cnt_.resize(cfg_.lv_grads_.size());
    layer_keys_.resize(cfg_.lv_grads_.size());
    layer_key_bcis_.resize(cfg_.lv_grads_.size());
    cont_cells_.resize(cfg_.lv_grads_.size());
    sort_perm_.resize(cfg_.lv_grads_.size());
  }

  // contour_mng.h:505-556.  xyzi: KITTI .bin layout, n points x (x,y,z,intensity)
  // (tools/pointcloud_util.h:24-38 keeps x,y,z of each 4-float record).
  bool makeBEV(const float *xyzi, int64_t n_pts) {
    if (!(n_pts > 10)) return false;  // CHECK_GT(ptr_gapc->size(), 10)
    std::map<int, Pixelf> tmp_pillars;
    for (int64_t i = 0; i < n_pts; i++) {
      float px = xyzi[4 * i + 0], py = xyzi[4 * i + 1], pz = xyzi[4 * i + 2];
      std::pair<int, int> rc = hashPointToImage(px, py);
      if (rc.first > 0) {
        float height = cfg_.lidar_height_ + pz;
        if (bev(rc.first, rc.second) < height) {
          bev(rc.first, rc.second) = height;
          V2F coor_f = pointToContRowCol(V2F(px, py));
          tmp_pillars[rc.first * cfg_.n_col_ + rc.second] = Pixelf(coor_f.x, coor_f.y, height);
        }
        max_bin_val_ = max_bin_val_ < height ? height : max_bin_val_;
        min_bin_val_ = min_bin_val_ > height ? height : min_bin_val_;
      }
    }
    bev_pixfs_.clear();
    bev_pixfs_.insert(bev_pixfs_.begin(), tmp_pillars.begin(), tmp_pillars.end());
    return true;
  }

  void clearImage() {
    bev_.clear();
    bev_.shrink_to_fit();
  }

  // src/cont2/contour_mng.cpp:274-353
  void makeContourRecursiveHelper(const Rect &cc_roi, const std::vector<uint8_t> &cc_mask, int level) {
    if (level >= (int)cfg_.lv_grads_.size()) return;
    float h_min = cfg_.lv_grads_[level];
    // cv::threshold(bev_roi, thres_roi, h_min, 255, THRESH_BINARY): 255 where src > thresh; convertTo CV_8U
    std::vector<uint8_t> bin_bev_roi((size_t)cc_roi.width * cc_roi.height);
    for (int i = 0; i < cc_roi.height; i++)
      for (int j = 0; j < cc_roi.width; j++)
        bin_bev_roi[(size_t)i * cc_roi.width + j] = bev(i + cc_roi.y, j + cc_roi.x) > h_min ? 255 : 0;
    if (level)
      for (size_t k = 0; k < bin_bev_roi.size(); k++) bin_bev_roi[k] = bin_bev_roi[k] & cc_mask[k];

    std::vector<int> labels;
    std::vector<std::array<int, 5>> stats;
    int n_lab = connectedComponentsWithStats8(bin_bev_roi, cc_roi.height, cc_roi.width, labels, stats);

    for (int n = 1; n < n_lab; n++) {
      if (stats[n][4] < cfg_.min_cont_cell_cnt_) continue;
      Rect rect_g{stats[n][0] + cc_roi.x, stats[n][1] + cc_roi.y, stats[n][2], stats[n][3]};
      Rect rect_l{stats[n][0], stats[n][1], stats[n][2], stats[n][3]};
      std::vector<uint8_t> mask_n((size_t)rect_l.width * rect_l.height);
      for (int i = 0; i < rect_l.height; i++)
        for (int j = 0; j < rect_l.width; j++)
          mask_n[(size_t)i * rect_l.width + j] =
              labels[(size_t)(i + rect_l.y) * cc_roi.width + (j + rect_l.x)] == n ? 255 : 0;

      RunningStatRecorder tmp_rec;
      int poi_r = -1, poi_c = -1;
      std::vector<int> cells;
      for (int i = 0; i < rect_l.height; i++)
        for (int j = 0; j < rect_l.width; j++)
          if (mask_n[(size_t)i * rect_l.width + j]) {
            poi_r = i + rect_g.y;
            poi_c = j + rect_g.x;
            int q_hash = poi_r * cfg_.n_col_ + poi_c;
            std::pair<int, Pixelf> sear_res = search_vec<Pixelf>(bev_pixfs_, 0, (int)bev_pixfs_.size() - 1, q_hash);
            tmp_rec.runningStatsF(sear_res.second.row_f, sear_res.second.col_f, bev(poi_r, poi_c));
            if (keep_cells_) cells.push_back(q_hash);
          }
      std::shared_ptr<ContourView> ptr_tmp_cv(new ContourView(level, poi_r, poi_c));
      ptr_tmp_cv->calcStatVals(tmp_rec, view_stat_cfg_);
      cont_views_[level].emplace_back(ptr_tmp_cv);
      if (keep_cells_) cont_cells_[level].emplace_back(std::move(cells));
      makeContourRecursiveHelper(rect_g, mask_n, level + 1);
    }
  }

  // contour_mng.h:588-960
  void makeContoursRecurs() {
    Rect full_bev_roi{0, 0, cfg_.n_col_, cfg_.n_row_};
    makeContourRecursiveHelper(full_bev_roi, std::vector<uint8_t>(1, 0), 0);

    for (int ll = 0; ll < (int)cont_views_.size(); ll++) {
      // contour_mng.h:596-599 sorts the shared_ptr vector; sorting (ptr, insertion idx) pairs with the
      // same comparator performs the identical sequence of comparisons and moves.
      std::vector<std::pair<std::shared_ptr<ContourView>, int>> tmp;
      tmp.reserve(cont_views_[ll].size());
      for (int j = 0; j < (int)cont_views_[ll].size(); j++) tmp.emplace_back(cont_views_[ll][j], j);
      std::sort(tmp.begin(), tmp.end(),
                [&](const std::pair<std::shared_ptr<ContourView>, int> &p1,
                    const std::pair<std::shared_ptr<ContourView>, int> &p2) -> bool {
                  return p1.first->cell_cnt_ > p2.first->cell_cnt_;
                });
      sort_perm_[ll].resize(tmp.size());
      for (int j = 0; j < (int)tmp.size(); j++) {
        cont_views_[ll][j] = tmp[j].first;
        sort_perm_[ll][j] = tmp[j].second;
      }
      layer_cell_cnt_[ll] = 0;
      for (int j = 0; j < (int)cont_views_[ll].size(); j++) layer_cell_cnt_[ll] += cont_views_[ll][j]->cell_cnt_;
      cont_perc_[ll].reserve(cont_views_[ll].size());
      for (int j = 0; j < (int)cont_views_[ll].size(); j++)
        cont_perc_[ll].push_back(cont_views_[ll][j]->cell_cnt_ * 1.0f / layer_cell_cnt_[ll]);
    }

    // contour_mng.h:693-895
    const int roi_radius_padded = std::ceil(cfg_.roi_radius_ + 1);
    for (int ll = 0; ll < (int)cfg_.lv_grads_.size(); ll++) {
      int accumulate_cell_cnt = 0;
      for (int seq = 0; seq < cfg_.piv_firsts_; seq++) {
        RetrievalKey key;
        key.setZero();
        BCI bci(seq, ll);
        if ((int)cont_views_[ll].size() > seq) accumulate_cell_cnt += cont_views_[ll][seq]->cell_cnt_;
        if ((int)cont_views_[ll].size() > seq && cont_views_[ll][seq]->cell_cnt_ >= cfg_.min_cont_key_cnt_) {
          V2F v_cen = cont_views_[ll][seq]->pos_mean_;
          int r_cen = int(v_cen.x), c_cen = int(v_cen.y);
          int r_min = std::max(0, r_cen - roi_radius_padded), r_max = std::min(cfg_.n_row_ - 1, r_cen + roi_radius_padded);
          int c_min = std::max(0, c_cen - roi_radius_padded), c_max = std::min(cfg_.n_col_ - 1, c_cen + roi_radius_padded);
          int num_bins = RET_KEY_DIM - 3;
          KeyFloatType bin_len = cfg_.roi_radius_ / num_bins;
          std::vector<KeyFloatType> ring_bins(num_bins, 0);
          int div_per_bin = 5;
          std::vector<KeyFloatType> discrete_divs(div_per_bin * num_bins, 0);
          KeyFloatType div_len = cfg_.roi_radius_ / (num_bins * div_per_bin);
          int cnt_point = 0;
          for (int rr = r_min; rr <= r_max; rr++) {
            for (int cc = c_min; cc <= c_max; cc++) {
              if (bev(rr, cc) < cfg_.lv_grads_[DIST_BIN_LAYERS[0]]) continue;
              int q_hash = rr * cfg_.n_col_ + cc;
              std::pair<int, Pixelf> sear_res = search_vec<Pixelf>(bev_pixfs_, 0, (int)bev_pixfs_.size() - 1, q_hash);
              KeyFloatType dist = (V2F(sear_res.second.row_f, sear_res.second.col_f) - v_cen).norm();
              if (dist < cfg_.roi_radius_ - 1e-2 && bev(rr, cc) > cfg_.lv_grads_[DIST_BIN_LAYERS[0]]) {
                int higher_cnt = 0;
                for (int ele = DIST_BIN_LAYERS[0]; ele < (int)cfg_.lv_grads_.size(); ele++)
                  if (bev(rr, cc) > cfg_.lv_grads_[ele]) higher_cnt++;
                cnt_point++;
                for (int div_idx = 0; div_idx < num_bins * div_per_bin; div_idx++)
                  discrete_divs[div_idx] += higher_cnt * gaussPDF<KeyFloatType>(div_idx * div_len + 0.5 * div_len, dist, 1.0);
              }
            }
          }
          for (int b = 0; b < num_bins; b++) {
            for (int d = 0; d < div_per_bin; d++) ring_bins[b] += discrete_divs[b * div_per_bin + d];
            ring_bins[b] *= bin_len / std::sqrt(cnt_point);
          }
          key(0) = std::sqrt(cont_views_[ll][seq]->eig_vals_.y * cont_views_[ll][seq]->cell_cnt_);
          key(1) = std::sqrt(cont_views_[ll][seq]->eig_vals_.x * cont_views_[ll][seq]->cell_cnt_);
          key(2) = std::sqrt(accumulate_cell_cnt);
          for (int nb = 0; nb < num_bins; nb++) key(3 + nb) = ring_bins[nb];

          // contour_mng.h:848-883
          for (int bl = 0; bl < NUM_BIN_KEY_LAYER; bl++) {
            int bit_offset = bl * BITS_PER_LAYER;
            for (int j = 0; j < std::min(cfg_.dist_firsts_, (int)cont_views_[DIST_BIN_LAYERS[bl]].size()); j++) {
              if (ll != DIST_BIN_LAYERS[bl] || j != seq) {
                V2F vec_cc = cont_views_[DIST_BIN_LAYERS[bl]][j]->pos_mean_ - cont_views_[ll][seq]->pos_mean_;
                float tmp_dist = vec_cc.norm();
                if (tmp_dist > (BITS_PER_LAYER - 1) * 1.01 + 5.43 - 1e-3 || tmp_dist <= 5.43) continue;
                float tmp_orie = std::atan2(vec_cc.y, vec_cc.x);
                int dist_idx = std::min(std::floor((tmp_dist - 5.43) / 1.01), BITS_PER_LAYER - 1.0) + bit_offset;
                bci.dist_bin_.set(dist_idx, true);
                bci.nei_pts_.emplace_back(DIST_BIN_LAYERS[bl], j, dist_idx, tmp_dist, tmp_orie);
              }
            }
          }
          if (!bci.nei_pts_.empty()) {
            std::sort(bci.nei_pts_.begin(), bci.nei_pts_.end(),
                      [&](const BCI::RelativePoint &p1, const BCI::RelativePoint &p2) { return p1.bit_pos < p2.bit_pos; });
            bci.nei_idx_segs_.emplace_back(0);
            for (int p1 = 0; p1 < (int)bci.nei_pts_.size(); p1++) {
              if (bci.nei_pts_[bci.nei_idx_segs_.back()].bit_pos != bci.nei_pts_[p1].bit_pos)
                bci.nei_idx_segs_.emplace_back(p1);
            }
            bci.nei_idx_segs_.emplace_back(bci.nei_pts_.size());
          }
        }
        layer_key_bcis_[ll].emplace_back(bci);
        layer_keys_[ll].emplace_back(key);
      }
    }
  }

  const std::vector<RetrievalKey> &getLevRetrievalKey(int level) const { return layer_keys_[level]; }
  const std::vector<std::shared_ptr<ContourView>> &getLevContours(int level) const { return cont_views_[level]; }
  int getLevTotalPix(int level) const { return layer_cell_cnt_[level]; }
  const std::vector<BCI> &getLevBCI(int level) const { return layer_key_bcis_[level]; }
  const BCI &getBCI(int level, int seq) const { return layer_key_bcis_[level][seq]; }
  int getIntID() const { return int_id_; }
  const ContourManagerConfig &getConfig() const { return cfg_; }

  // contour_mng.h:1279-1284
  static bool checkContPairSim(const ContourManager &src, const ContourManager &tgt, const ConstellationPair &cstl,
                               const ContourSimThresConfig &cont_sim) {
    return ContourView::checkSim(*src.cont_views_[cstl.level][cstl.seq_src], *tgt.cont_views_[cstl.level][cstl.seq_tgt],
                                 cont_sim);
  }

  // contour_mng.h:1124-1242
  static ScorePairwiseSim checkConstellCorrespSim(const ContourManager &src, const ContourManager &tgt,
                                                  const std::vector<ConstellationPair> &cstl_in, const ScorePairwiseSim &lb,
                                                  const ContourSimThresConfig &cont_sim,
                                                  std::vector<ConstellationPair> &cstl_out, std::vector<float> &area_perc) {
    ScorePairwiseSim ret;
    cstl_out.clear();
    area_perc.clear();
    for (auto pr : cstl_in) {
      if (checkContPairSim(src, tgt, pr, cont_sim)) cstl_out.push_back(pr);
    }
    ret.i_indiv_sim = cstl_out.size();
    if (ret.i_indiv_sim < lb.i_indiv_sim) return ret;

    V2F shaft_src(0, 0), shaft_tgt(0, 0);
    for (int i = 1; i < std::min((int)cstl_out.size(), 10); i++) {
      for (int j = 0; j < i; j++) {
        V2F curr_shaft = src.cont_views_[cstl_out[i].level][cstl_out[i].seq_src]->pos_mean_ -
                         src.cont_views_[cstl_out[j].level][cstl_out[j].seq_src]->pos_mean_;
        if (curr_shaft.norm() > shaft_src.norm()) {
          shaft_src = curr_shaft.normalized();
          shaft_tgt = (tgt.cont_views_[cstl_out[i].level][cstl_out[i].seq_tgt]->pos_mean_ -
                       tgt.cont_views_[cstl_out[j].level][cstl_out[j].seq_tgt]->pos_mean_)
                          .normalized();
        }
      }
    }
    int num_sim = cstl_out.size();
    for (int i = 0; i < num_sim;) {
      const auto &sc1 = src.cont_views_[cstl_out[i].level][cstl_out[i].seq_src],
                 &tc1 = tgt.cont_views_[cstl_out[i].level][cstl_out[i].seq_tgt];
      if (sc1->ecc_feat_ && tc1->ecc_feat_) {
        float theta_s = std::acos(shaft_src.dot(sc1->eig_vecs_.col(1)));
        float theta_t = std::acos(shaft_tgt.dot(tc1->eig_vecs_.col(1)));
        if (diff_delt<float>(theta_s, theta_t, M_PI / 6) && diff_delt<float>(M_PI - theta_s, theta_t, M_PI / 6)) {
          std::swap(cstl_out[i], cstl_out[num_sim - 1]);
          num_sim--;
          continue;
        }
      }
      i++;
    }
    cstl_out.erase(cstl_out.begin() + num_sim, cstl_out.end());
    ret.i_orie_sim = cstl_out.size();
    if (ret.i_orie_sim < lb.i_orie_sim) return ret;
    area_perc.reserve(cstl_out.size());
    for (const auto &i : cstl_out)
      area_perc.push_back(0.5f * (src.cont_perc_[i.level][i.seq_src] + tgt.cont_perc_[i.level][i.seq_tgt]));
    return ret;
  }

  // contour_mng.h:1252-1277.  Eigen::umeyama(src, tgt, false) restated for 2-D: the optimal
  // rotation R = U S V^T of sigma = (1/n) * dst_demean * src_demean^T equals the closed form
  // angle atan2(sigma10 - sigma01, sigma00 + sigma11) (both maximise trace(R^T sigma) over SO(2));
  // translation = dst_mean - R * src_mean.  The reference then keeps atan2(T10, T00) and t.
  static Iso2d getTFFromConstell(const ContourManager &src, const ContourManager &tgt,
                                 const std::vector<ConstellationPair> &cstl) {
    int num_elem = cstl.size();
    std::vector<V2D> p1(num_elem), p2(num_elem);
    for (int i = 0; i < num_elem; i++) {
      const V2F &a = src.cont_views_[cstl[i].level][cstl[i].seq_src]->pos_mean_;
      const V2F &b = tgt.cont_views_[cstl[i].level][cstl[i].seq_tgt]->pos_mean_;
      p1[i] = V2D(a.x, a.y);
      p2[i] = V2D(b.x, b.y);
    }
    const double one_over_n = 1.0 / (double)num_elem;
    V2D sm(0, 0), dm(0, 0);
    for (int i = 0; i < num_elem; i++) {
      sm.x += p1[i].x;
      sm.y += p1[i].y;
      dm.x += p2[i].x;
      dm.y += p2[i].y;
    }
    sm = sm * one_over_n;
    dm = dm * one_over_n;
    double s00 = 0, s01 = 0, s10 = 0, s11 = 0;
    for (int i = 0; i < num_elem; i++) {
      V2D a = p1[i] - sm, b = p2[i] - dm;
      s00 += b.x * a.x;
      s01 += b.x * a.y;
      s10 += b.y * a.x;
      s11 += b.y * a.y;
    }
    s00 *= one_over_n;
    s01 *= one_over_n;
    s10 *= one_over_n;
    s11 *= one_over_n;
    double sn = s10 - s01, cs = s00 + s11;
    double nrm = std::sqrt(sn * sn + cs * cs);
    double r00 = 1, r10 = 0;
    if (nrm > 0) {
      r00 = cs / nrm;
      r10 = sn / nrm;
    }
    // Rt.col(2) = dst_mean - R * src_mean
    double tx = dm.x - (r00 * sm.x + (-r10) * sm.y);
    double ty = dm.y - (r10 * sm.x + r00 * sm.y);
    return Iso2d::fromAngTrans(std::atan2(r10, r00), V2D(tx, ty));
  }

  // ---- export to the shared POD layout (comparison with the device output) ----
  void exportDesc(cc_scan_desc_t *out) const {
    std::memset(out, 0, sizeof(*out));
    out->max_bin_val = max_bin_val_;
    out->min_bin_val = min_bin_val_;
    out->n_pix = (int32_t)bev_pixfs_.size();
    for (int l = 0; l < CC_NLEV; l++) {
      out->n_cont[l] = (int32_t)cont_views_[l].size();
      out->n_stored[l] = std::min<int>(out->n_cont[l], CC_MAXC);
      if (out->n_cont[l] > CC_MAXC) out->flags |= 1;
      out->layer_cell_cnt[l] = layer_cell_cnt_[l];
      for (int j = 0; j < out->n_stored[l]; j++) {
        const ContourView &v = *cont_views_[l][j];
        cc_contour_t &c = out->cont[l][j];
        c.level = v.level_;
        c.poi[0] = v.poi_[0];
        c.poi[1] = v.poi_[1];
        c.cell_cnt = v.cell_cnt_;
        c.pos_mean[0] = v.pos_mean_.x;
        c.pos_mean[1] = v.pos_mean_.y;
        c.pos_cov[0] = v.pos_cov_.a[0][0];
        c.pos_cov[1] = v.pos_cov_.a[1][0];
        c.pos_cov[2] = v.pos_cov_.a[0][1];
        c.pos_cov[3] = v.pos_cov_.a[1][1];
        c.eig_vals[0] = v.eig_vals_.x;
        c.eig_vals[1] = v.eig_vals_.y;
        c.eig_vecs[0] = v.eig_vecs_.a[0][0];
        c.eig_vecs[1] = v.eig_vecs_.a[1][0];
        c.eig_vecs[2] = v.eig_vecs_.a[0][1];
        c.eig_vecs[3] = v.eig_vecs_.a[1][1];
        c.eccen = v.eccen_;
        c.vol3_mean = v.vol3_mean_;
        c.com[0] = v.com_.x;
        c.com[1] = v.com_.y;
        c.ecc_feat = v.ecc_feat_;
        c.com_feat = v.com_feat_;
      }
      for (int s = 0; s < cfg_.piv_firsts_ && s < CC_NPIV; s++) {
        for (int k = 0; k < CC_KEY_DIM; k++) out->keys[l][s][k] = layer_keys_[l][s].array[k];
        const BCI &b = layer_key_bcis_[l][s];
        cc_bci_t &o = out->bcis[l][s];
        for (int w = 0; w < CC_BCI_LAYERS; w++) {
          uint64_t word = 0;
          for (int bit = 0; bit < 64; bit++)
            if (b.dist_bin_[w * 64 + bit]) word |= (1ull << bit);
          o.dist_bin[w] = word;
        }
        o.piv_seq = b.piv_seq_;
        o.level = b.level_;
        o.n_pts = (uint8_t)b.nei_pts_.size();
        o.n_segs = (uint8_t)b.nei_idx_segs_.size();
        for (size_t k = 0; k < b.nei_idx_segs_.size(); k++) o.segs[k] = b.nei_idx_segs_[k];
        for (size_t k = 0; k < b.nei_pts_.size(); k++) {
          o.pts[k].level = b.nei_pts_[k].level;
          o.pts[k].seq = b.nei_pts_[k].seq;
          o.pts[k].bit_pos = b.nei_pts_[k].bit_pos;
          o.pts[k].r = b.nei_pts_[k].r;
          o.pts[k].theta = b.nei_pts_[k].theta;
        }
      }
    }
  }

  // canonical label image (SURVEY.md 8(a) parity definition): L_l(r,c) = seq after the size sort, -1 none.
  void exportLabels(int16_t *labels) const {
    const size_t ncell = (size_t)cfg_.n_row_ * cfg_.n_col_;
    for (size_t i = 0; i < CC_NLEV * ncell; i++) labels[i] = -1;
    for (int l = 0; l < CC_NLEV; l++)
      for (int seq = 0; seq < (int)sort_perm_[l].size(); seq++)
        for (int h : cont_cells_[l][sort_perm_[l][seq]]) labels[l * ncell + h] = (int16_t)seq;
  }
};

}  // namespace orc
