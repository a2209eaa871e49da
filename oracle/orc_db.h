// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// CPU restatement of include/cont2/contour_db.h and src/cont2/contour_db.cpp of the reference:
// TreeBucket / LayerDB / ContourDB (bucketed, time-delayed key store), MyKNNResSet semantics,
// CandidateManager (checkCandWithHint / tidyUpCandidates / fineOptimize).
// The kd-tree itself (thirdparty/nanoflann.hpp) is replaced by an exact scan with the same
// result-set rule (top-k, strict `dist < worst`, first-found wins ties, nanoflann.hpp:157-230 and
// L2 accumulation order of nanoflann.hpp:427-461); oracle/_ref links the real nanoflann to
// validate that (tests/test_oracle_knn_ref.py).
#pragma once
#include <algorithm>
#include <chrono>
#include <map>
#include <memory>
#include <numeric>
#include <vector>

#include "orc_contour.h"
#include "orc_gmm.h"

namespace orc {

const KeyFloatType MAX_BUCKET_VAL = 1000.0f;
const KeyFloatType MAX_DIST_SQ = 1e6;

// contour_db.h:54-65
struct TreeBucketConfig {
  double max_elapse_ = 25.0;
  double min_elapse_ = 15.0;
};
struct IndexOfKey {
  size_t gidx{};
  int level{};
  int seq{};
  IndexOfKey(size_t g, int l, int s) : gidx(g), level(l), seq(s) {}
};

// nanoflann.hpp:427-461 L2_Adaptor::evalMetric for dim = 10, worst_dist = -1 (no early exit)
inline KeyFloatType l2_nanoflann(const KeyFloatType *a, const KeyFloatType *b) {
  KeyFloatType result = KeyFloatType();
  int d = 0;
  for (int g = 0; g < 2; g++) {
    const KeyFloatType diff0 = a[d] - b[d];
    const KeyFloatType diff1 = a[d + 1] - b[d + 1];
    const KeyFloatType diff2 = a[d + 2] - b[d + 2];
    const KeyFloatType diff3 = a[d + 3] - b[d + 3];
    result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
    d += 4;
  }
  for (; d < RET_KEY_DIM; d++) {
    const KeyFloatType diff0 = a[d] - b[d];
    result += diff0 * diff0;
  }
  return result;
}

// Optional kd-tree backend: the REAL vendored nanoflann of the reference, compiled into
// oracle/_ref/libref_knn.so (see oracle/ref_knn.cpp).  When installed (orc_set_knn_backend) the
// buckets build and query real kd-trees exactly like TreeBucket::rebuildTree / knnSearch; without it
// the exact scan below is used (same result set; tie order may differ).
struct KnnBackend {
  void *(*create)() = nullptr;
  void (*destroy)(void *) = nullptr;
  void (*build)(void *, const float *, int) = nullptr;
  void (*query)(void *, const float *, int, float, size_t *, float *) = nullptr;
};
inline KnnBackend &knn_backend() {
  static KnnBackend b;
  return b;
}

// contour_db.h:68-156
struct TreeBucket {
  struct RetrTriplet {
    RetrievalKey pt;
    double ts{};
    IndexOfKey iok;
    RetrTriplet(const RetrievalKey &_a, double _b, IndexOfKey i) : pt(_a), ts(_b), iok(i) {}
  };
  TreeBucketConfig cfg_;
  KeyFloatType buc_beg_{}, buc_end_{};
  std::vector<RetrievalKey> data_tree_;
  bool tree_built = false;  // tree_ptr != nullptr
  std::vector<RetrTriplet> buffer_;
  std::vector<IndexOfKey> gkidx_tree_;
  std::shared_ptr<void> kd_;  // backend tree handle

  TreeBucket(const TreeBucketConfig &config, KeyFloatType beg, KeyFloatType end) : cfg_(config), buc_beg_(beg), buc_end_(end) {}
  size_t getTreeSize() const { return data_tree_.size(); }
  void pushBuffer(const RetrievalKey &tree_key, double ts, IndexOfKey iok) { buffer_.emplace_back(tree_key, ts, iok); }
  bool needPopBuffer(double curr_ts) const {
    double ts_overflow = curr_ts - cfg_.max_elapse_;
    if (buffer_.empty() || buffer_[0].ts > ts_overflow) return false;
    return true;
  }
  void rebuildTree() {
    tree_built = true;
    KnnBackend &be = knn_backend();
    if (be.create) {
      if (!kd_) kd_ = std::shared_ptr<void>(be.create(), be.destroy);
      static_assert(sizeof(RetrievalKey) == 40, "key layout");
      be.build(kd_.get(), data_tree_.empty() ? nullptr : data_tree_[0].array, (int)data_tree_.size());
    }
  }
  // contour_db.h:121-143
  void popBufferMax(double curr_ts) {
    double ts_cutoff = curr_ts - cfg_.min_elapse_;
    int gap = 0;
    for (; gap < (int)buffer_.size(); gap++)
      if (buffer_[gap].ts >= ts_cutoff) break;
    if (gap > 0) {
      for (int i = 0; i < gap; i++) {
        data_tree_.emplace_back(buffer_[i].pt);
        gkidx_tree_.emplace_back(buffer_[i].iok);
      }
      buffer_.erase(buffer_.begin(), buffer_.begin() + gap);
      rebuildTree();
    }
  }
  // src/cont2/contour_db.cpp:381-403 with MyKNNResSet (contour_db.h:32-52) over an exact scan
  void knnSearch(const int num_res, std::vector<IndexOfKey> &ret_idx, std::vector<KeyFloatType> &out_dist_sq,
                 RetrievalKey q_key, const KeyFloatType max_dist_sq) const {
    ret_idx.clear();
    out_dist_sq.resize(num_res);
    std::fill(out_dist_sq.begin(), out_dist_sq.end(), MAX_DIST_SQ);
    if (!tree_built) return;
    ret_idx.reserve(num_res);
    std::vector<size_t> idx(num_res, 0);
    if (knn_backend().create && kd_) {
      if (!data_tree_.empty()) knn_backend().query(kd_.get(), q_key.array, num_res, max_dist_sq, idx.data(), out_dist_sq.data());
      else out_dist_sq[num_res - 1] = max_dist_sq;
      for (int i = 0; i < num_res; i++) ret_idx.emplace_back(gkidx_tree_.empty() ? IndexOfKey(0, 0, 0) : gkidx_tree_[idx[i]]);
      return;
    }
    // MyKNNResSet::init
    size_t count = 0;
    const size_t capacity = num_res;
    if (capacity) out_dist_sq[capacity - 1] = max_dist_sq;
    if (!data_tree_.empty()) {
      for (size_t p = 0; p < data_tree_.size(); p++) {
        KeyFloatType dist = l2_nanoflann(q_key.array, data_tree_[p].array);
        if (dist < out_dist_sq[capacity - 1]) {
          // KNNResultSet::addPoint
          size_t i;
          for (i = count; i > 0; --i) {
            if (out_dist_sq[i - 1] > dist) {
              if (i < capacity) {
                out_dist_sq[i] = out_dist_sq[i - 1];
                idx[i] = idx[i - 1];
              }
            } else
              break;
          }
          if (i < capacity) {
            out_dist_sq[i] = dist;
            idx[i] = p;
          }
          if (count < capacity) count++;
        }
      }
    }
    for (int i = 0; i < num_res; i++) ret_idx.emplace_back(gkidx_tree_.empty() ? IndexOfKey(0, 0, 0) : gkidx_tree_[idx[i]]);
  }
};

// contour_db.h:159-217, src/cont2/contour_db.cpp:63-379
struct LayerDB {
  static const int min_elem_split_ = 100;
  static constexpr double imba_diff_ratio_ = 0.2;
  static const int max_num_backets_ = 6;
  static const int bucket_chann_ = 0;
  std::vector<TreeBucket> buckets_;
  std::vector<KeyFloatType> bucket_ranges_;

  explicit LayerDB(const TreeBucketConfig &tb_cfg) {
    bucket_ranges_.resize(max_num_backets_ + 1);
    bucket_ranges_.front() = -MAX_BUCKET_VAL;
    bucket_ranges_.back() = MAX_BUCKET_VAL;
    buckets_.emplace_back(tb_cfg, -MAX_BUCKET_VAL, MAX_BUCKET_VAL);
    for (int i = 1; i < max_num_backets_; i++) {
      bucket_ranges_[i] = MAX_BUCKET_VAL;
      buckets_.emplace_back(tb_cfg, MAX_BUCKET_VAL, MAX_BUCKET_VAL);
    }
  }
  void pushBuffer(const RetrievalKey &layer_key, double ts, IndexOfKey scan_key_gidx) {
    for (int i = 0; i < max_num_backets_; i++) {
      if (bucket_ranges_[i] <= layer_key(bucket_chann_) && layer_key(bucket_chann_) < bucket_ranges_[i + 1]) {
        if (layer_key.sum() != 0) buckets_[i].pushBuffer(layer_key, ts, scan_key_gidx);
        return;
      }
    }
  }

  // src/cont2/contour_db.cpp:63-317
  void rebuild(int idx_t1, double curr_ts) {
    TreeBucket &tr1 = buckets_[idx_t1], &tr2 = buckets_[idx_t1 + 1];
    bool pb1 = tr1.needPopBuffer(curr_ts), pb2 = tr2.needPopBuffer(curr_ts);
    if (!pb1 && !pb2) return;
    int sz1 = tr1.getTreeSize(), sz2 = tr2.getTreeSize();
    double diff_ratio = 1.0 * std::abs(sz1 - sz2) / std::max(sz1, sz2);
    if (pb1 && !pb2 && (diff_ratio < imba_diff_ratio_ || std::max(sz1, sz2) < min_elem_split_)) {
      tr1.popBufferMax(curr_ts);
      return;
    }
    if (!pb1 && pb2 && (diff_ratio < imba_diff_ratio_ || std::max(sz1, sz2) < min_elem_split_)) {
      tr2.popBufferMax(curr_ts);
      return;
    }
    if (diff_ratio < 0.5 * imba_diff_ratio_) {
      if (pb1) tr1.popBufferMax(curr_ts);
      if (pb2) tr2.popBufferMax(curr_ts);
      return;
    }
    if (sz1 > sz2) {
      int to_move_max = int((sz1 - sz2 + imba_diff_ratio_ * sz2) / (2 - imba_diff_ratio_));
      int to_move_mid = int((sz1 - sz2) / 2.0);
      int to_move_min = std::max(0, int((sz1 - sz2 - imba_diff_ratio_ * sz1) / (2 - imba_diff_ratio_)));
      std::vector<int> sort_permu(sz1);
      std::iota(sort_permu.begin(), sort_permu.end(), 0);
      std::sort(sort_permu.begin(), sort_permu.end(), [&](const int &a, const int &b) {
        return tr1.data_tree_[a](bucket_chann_) < tr1.data_tree_[b](bucket_chann_);
      });
      int num_to_move = 0;
      KeyFloatType split_val = tr1.buc_end_;
      if (to_move_mid <= 0 || to_move_mid >= sz1) {
        // reference reads one past the end here (UB); treated as "cannot split"
      } else if (tr1.data_tree_[sort_permu[sz1 - to_move_mid]](bucket_chann_) !=
          tr1.data_tree_[sort_permu[sz1 - to_move_mid - 1]](bucket_chann_)) {
        num_to_move = to_move_mid;
        split_val = tr1.data_tree_[sort_permu[sz1 - to_move_mid]](bucket_chann_);
      } else {
        KeyFloatType contagious_val = tr1.data_tree_[sort_permu[sz1 - to_move_mid]](bucket_chann_);
        int i = to_move_mid - 1;
        for (; i > to_move_min; i--) {
          if (tr1.data_tree_[sort_permu[sz1 - i]](bucket_chann_) != contagious_val) {
            num_to_move = i;
            split_val = tr1.data_tree_[sort_permu[sz1 - i]](bucket_chann_);
            break;
          }
        }
        if (num_to_move == 0) {
          i = to_move_mid + 1;
          for (; i < to_move_max; i++) {
            if (tr1.data_tree_[sort_permu[sz1 - i]](bucket_chann_) != contagious_val) {
              num_to_move = i - 1;
              split_val = contagious_val;
              break;
            }
          }
        }
      }
      if (num_to_move == 0) {
        tr1.popBufferMax(curr_ts);
        if (pb2) tr2.popBufferMax(curr_ts);
        return;
      }
      for (int i = 0; i < num_to_move; i++) {
        tr2.data_tree_.emplace_back(tr1.data_tree_[sort_permu[sz1 - i - 1]]);
        tr2.gkidx_tree_.emplace_back(tr1.gkidx_tree_[sort_permu[sz1 - i - 1]]);
      }
      int p_dat = sz1 - 1, p_perm = sz1 - 1;
      for (; p_perm >= sz1 - num_to_move; p_perm--) {
        while (tr1.data_tree_[p_dat](bucket_chann_) >= split_val) p_dat--;
        if (sort_permu[p_perm] < p_dat) {
          std::swap(tr1.data_tree_[p_dat], tr1.data_tree_[sort_permu[p_perm]]);
          std::swap(tr1.gkidx_tree_[p_dat], tr1.gkidx_tree_[sort_permu[p_perm]]);
          p_dat--;
        }
      }
      tr1.data_tree_.resize(p_dat + 1);
      tr1.gkidx_tree_.resize(p_dat + 1, tr1.gkidx_tree_[0]);
      int p1 = 0, p2 = tr1.buffer_.size() - 1, sz_rem;
      while (p1 <= p2) {
        if (tr1.buffer_[p1].pt(bucket_chann_) >= split_val && tr1.buffer_[p2].pt(bucket_chann_) < split_val) {
          std::swap(tr1.buffer_[p1], tr1.buffer_[p2]);
          p1++;
          p2--;
        } else {
          if (tr1.buffer_[p2].pt(bucket_chann_) >= split_val) p2--;
          if (tr1.buffer_[p1].pt(bucket_chann_) < split_val) p1++;
        }
      }
      sz_rem = p2 + 1;
      tr2.buffer_.insert(tr2.buffer_.end(), tr1.buffer_.begin() + sz_rem, tr1.buffer_.end());
      tr1.buffer_.erase(tr1.buffer_.begin() + sz_rem, tr1.buffer_.end());
      tr1.buc_end_ = tr2.buc_beg_ = split_val;
      bucket_ranges_[idx_t1 + 1] = split_val;
    } else {
      int to_move_max = int((sz2 - sz1 + imba_diff_ratio_ * sz1) / (2 - imba_diff_ratio_));
      int to_move_mid = int((sz2 - sz1) / 2.0);
      int to_move_min = std::max(0, int((sz2 - sz1 - imba_diff_ratio_ * sz2) / (2 - imba_diff_ratio_)));
      std::vector<int> sort_permu(sz2);
      std::iota(sort_permu.begin(), sort_permu.end(), 0);
      std::sort(sort_permu.begin(), sort_permu.end(), [&](const int &a, const int &b) {
        return tr2.data_tree_[a](bucket_chann_) > tr2.data_tree_[b](bucket_chann_);
      });
      int num_to_move = 0;
      KeyFloatType split_val = tr1.buc_end_;
      if (to_move_mid <= 0 || to_move_mid >= sz2) {
        // reference reads one past the end here (UB); treated as "cannot split"
      } else if (tr2.data_tree_[sort_permu[sz2 - to_move_mid]](bucket_chann_) !=
          tr2.data_tree_[sort_permu[sz2 - to_move_mid - 1]](bucket_chann_)) {
        num_to_move = to_move_mid;
        split_val = tr2.data_tree_[sort_permu[sz2 - to_move_mid - 1]](bucket_chann_);
      } else {
        KeyFloatType contagious_val = tr2.data_tree_[sort_permu[sz2 - to_move_mid]](bucket_chann_);
        int i = to_move_mid - 1;
        for (; i > to_move_min; i--) {
          if (tr2.data_tree_[sort_permu[sz2 - i]](bucket_chann_) != contagious_val) {
            num_to_move = i;
            split_val = contagious_val;
            break;
          }
        }
        if (num_to_move == 0) {
          i = to_move_mid + 1;
          for (; i < to_move_max; i++) {
            if (tr2.data_tree_[sort_permu[sz2 - i]](bucket_chann_) != contagious_val) {
              num_to_move = i - 1;
              split_val = tr2.data_tree_[sort_permu[sz2 - i]](bucket_chann_);
              break;
            }
          }
        }
      }
      if (num_to_move == 0) {
        if (pb1) tr1.popBufferMax(curr_ts);
        tr2.popBufferMax(curr_ts);
        return;
      }
      for (int i = 0; i < num_to_move; i++) {
        tr1.data_tree_.emplace_back(tr2.data_tree_[sort_permu[sz2 - i - 1]]);
        tr1.gkidx_tree_.emplace_back(tr2.gkidx_tree_[sort_permu[sz2 - i - 1]]);
      }
      int p_dat = sz2 - 1, p_perm = sz2 - 1;
      for (; p_perm >= sz2 - num_to_move; p_perm--) {
        while (tr2.data_tree_[p_dat](bucket_chann_) < split_val) p_dat--;
        if (sort_permu[p_perm] < p_dat) {
          std::swap(tr2.data_tree_[p_dat], tr2.data_tree_[sort_permu[p_perm]]);
          std::swap(tr2.gkidx_tree_[p_dat], tr2.gkidx_tree_[sort_permu[p_perm]]);
          p_dat--;
        }
      }
      tr2.data_tree_.resize(p_dat + 1);
      tr2.gkidx_tree_.resize(p_dat + 1, tr2.gkidx_tree_[0]);
      int p1 = 0, p2 = tr2.buffer_.size() - 1, sz_rem;
      while (p1 <= p2) {
        if (tr2.buffer_[p1].pt(bucket_chann_) < split_val && tr2.buffer_[p2].pt(bucket_chann_) >= split_val) {
          std::swap(tr2.buffer_[p1], tr2.buffer_[p2]);
          p1++;
          p2--;
        } else {
          if (tr2.buffer_[p2].pt(bucket_chann_) < split_val) p2--;
          if (tr2.buffer_[p1].pt(bucket_chann_) >= split_val) p1++;
        }
      }
      sz_rem = p2 + 1;
      tr1.buffer_.insert(tr1.buffer_.end(), tr2.buffer_.begin() + sz_rem, tr2.buffer_.end());
      tr2.buffer_.erase(tr2.buffer_.begin() + sz_rem, tr2.buffer_.end());
      tr1.buc_end_ = tr2.buc_beg_ = split_val;
      bucket_ranges_[idx_t1 + 1] = split_val;
    }
    std::sort(tr1.buffer_.begin(), tr1.buffer_.end(), [&](const auto &a, const auto &b) { return a.ts < b.ts; });
    std::sort(tr2.buffer_.begin(), tr2.buffer_.end(), [&](const auto &a, const auto &b) { return a.ts < b.ts; });
    tr1.popBufferMax(curr_ts);
    tr2.popBufferMax(curr_ts);
  }

  // src/cont2/contour_db.cpp:319-379
  void layerKNNSearch(const RetrievalKey &q_key, const int k_top, const KeyFloatType max_dist_sq,
                      std::vector<std::pair<IndexOfKey, KeyFloatType>> &res_pairs) const {
    int mid_bucket = 0;
    for (int i = 0; i < max_num_backets_; i++) {
      if (bucket_ranges_[i] <= q_key(bucket_chann_) && bucket_ranges_[i + 1] > q_key(bucket_chann_)) {
        mid_bucket = i;
        break;
      }
    }
    KeyFloatType max_dist_sq_run = max_dist_sq;
    res_pairs.clear();
    for (int i = 0; i < max_num_backets_; i++) {
      std::vector<IndexOfKey> tmp_gidx;
      std::vector<KeyFloatType> tmp_dists_sq;
      if (i == 0) {
        buckets_[mid_bucket].knnSearch(k_top, tmp_gidx, tmp_dists_sq, q_key, max_dist_sq_run);
        for (int j = 0; j < k_top; j++)
          if (tmp_dists_sq[j] < max_dist_sq_run)
            res_pairs.emplace_back(tmp_gidx[j], tmp_dists_sq[j]);
          else
            break;
      } else if (mid_bucket - i >= 0) {
        if ((q_key(bucket_chann_) - bucket_ranges_[mid_bucket - i + 1]) * (q_key(bucket_chann_) - bucket_ranges_[mid_bucket - i + 1]) >
            max_dist_sq_run) {
          continue;
        }
        buckets_[mid_bucket - i].knnSearch(k_top, tmp_gidx, tmp_dists_sq, q_key, max_dist_sq_run);
        for (int j = 0; j < k_top; j++)
          if (tmp_dists_sq[j] < max_dist_sq_run)
            res_pairs.emplace_back(tmp_gidx[j], tmp_dists_sq[j]);
          else
            break;
      } else if (mid_bucket + i < max_num_backets_) {
        if ((q_key(bucket_chann_) - bucket_ranges_[mid_bucket + i]) * (q_key(bucket_chann_) - bucket_ranges_[mid_bucket + i]) >
            max_dist_sq_run) {
          continue;
        }
        buckets_[mid_bucket + i].knnSearch(k_top, tmp_gidx, tmp_dists_sq, q_key, max_dist_sq_run);
        for (int j = 0; j < k_top; j++)
          if (tmp_dists_sq[j] < max_dist_sq_run)
            res_pairs.emplace_back(tmp_gidx[j], tmp_dists_sq[j]);
          else
            break;
      }
      std::sort(res_pairs.begin(), res_pairs.end(),
                [&](const std::pair<IndexOfKey, KeyFloatType> &a, const std::pair<IndexOfKey, KeyFloatType> &b) {
                  return a.second < b.second;
                });
      if ((int)res_pairs.size() >= k_top) {
        res_pairs.resize(k_top, res_pairs[0]);
        max_dist_sq_run = res_pairs.back().second;
      }
    }
  }
};

// contour_db.h:244-250
struct CandidateScoreEnsemble {
  ScoreConstellSim sim_constell;
  ScorePairwiseSim sim_pair;
  ScorePostProc sim_post;
};

// contour_db.h:264-656
struct CandidateManager {
  struct CandidateAnchorProp {
    std::map<ConstellationPair, float> constell_;
    Iso2d T_delta_;
    float correlation_ = 0;
    int vote_cnt_ = 0;
    float area_perc_ = 0;
  };
  struct CandidatePoseData {
    std::shared_ptr<const ContourManager> cm_cand_;
    std::unique_ptr<ConstellCorrelation> corr_est_;
    std::vector<CandidateAnchorProp> anch_props_;
    // contour_db.h:286-338
    void addProposal(const Iso2d &T_prop, const std::vector<ConstellationPair> &sim_pairs, const std::vector<float> &sim_area_perc) {
      for (size_t i = 0; i < anch_props_.size(); i++) {
        const Iso2d delta_T = T_prop.inverse() * anch_props_[i].T_delta_;
        if (delta_T.translation().norm() < 2.0 && std::abs(std::atan2(delta_T(1, 0), delta_T(0, 0))) < 0.3) {
          for (size_t j = 0; j < sim_pairs.size(); j++) anch_props_[i].constell_.insert({sim_pairs[j], sim_area_perc[j]});
          anch_props_[i].vote_cnt_ += sim_pairs.size();
          int w1 = anch_props_[i].vote_cnt_, w2 = sim_pairs.size();
          V2D trans_bl = (anch_props_[i].T_delta_.translation() * w1 + T_prop.translation() * w2) / (w1 + w2);
          double ang1 = std::atan2(anch_props_[i].T_delta_(1, 0), anch_props_[i].T_delta_(0, 0));
          double ang2 = std::atan2(T_prop(1, 0), T_prop(0, 0));
          double diff = ang2 - ang1;
          if (diff < 0) diff += 2 * M_PI;
          if (diff > M_PI) diff -= 2 * M_PI;
          double ang_bl = diff * w2 / (w1 + w2) + ang1;
          anch_props_[i].T_delta_ = Iso2d::fromAngTrans(ang_bl, trans_bl);
          return;
        }
      }
      if (anch_props_.size() > 3) return;
      anch_props_.emplace_back();
      anch_props_.back().T_delta_ = T_prop;
      for (size_t j = 0; j < sim_pairs.size(); j++) anch_props_.back().constell_.insert({sim_pairs[j], sim_area_perc[j]});
      anch_props_.back().vote_cnt_ = sim_pairs.size();
    }
  };

  std::shared_ptr<const ContourManager> cm_tgt_;
  const CandidateScoreEnsemble sim_ub_;
  CandidateScoreEnsemble sim_var_;
  std::map<int, int> cand_id_pos_pair_;
  std::vector<CandidatePoseData> candidates_;
  int cand_aft_check1 = 0, cand_aft_check2 = 0, cand_aft_check3 = 0;
  int n_cand_pose = 0;

  CandidateManager(std::shared_ptr<const ContourManager> cm_q, const CandidateScoreEnsemble sim_lb, const CandidateScoreEnsemble sim_ub)
      : cm_tgt_(std::move(cm_q)), sim_ub_(sim_ub), sim_var_(sim_lb) {}

  // contour_db.h:374-488 (DYNAMIC_THRES = 0, CMakeLists.txt:13-21)
  // cand_key: identity of the candidate scan in cand_id_pos_pair_ (the reference uses getIntID()).
  CandidateScoreEnsemble checkCandWithHint(const std::shared_ptr<const ContourManager> &cm_cand, const ConstellationPair &anchor_pair,
                                           const ContourSimThresConfig &cont_sim) {
    int cand_id = cm_cand->getIntID();
    CandidateScoreEnsemble ret_score;
    bool anchor_sim = ContourManager::checkContPairSim(*cm_cand, *cm_tgt_, anchor_pair, cont_sim);
    if (!anchor_sim) return ret_score;
    cand_aft_check1++;
    std::vector<ConstellationPair> tmp_pairs1;
    ScoreConstellSim ret_constell_sim =
        BCI::checkConstellSim(cm_cand->getBCI(anchor_pair.level, anchor_pair.seq_src),
                              cm_tgt_->getBCI(anchor_pair.level, anchor_pair.seq_tgt), sim_var_.sim_constell, tmp_pairs1);
    ret_score.sim_constell = ret_constell_sim;
    if (ret_constell_sim.overall() < sim_var_.sim_constell.overall()) return ret_score;
    cand_aft_check2++;
    std::vector<ConstellationPair> tmp_pairs2;
    std::vector<float> tmp_area_perc;
    ScorePairwiseSim ret_pairwise_sim =
        ContourManager::checkConstellCorrespSim(*cm_cand, *cm_tgt_, tmp_pairs1, sim_var_.sim_pair, cont_sim, tmp_pairs2, tmp_area_perc);
    ret_score.sim_pair = ret_pairwise_sim;
    if (ret_pairwise_sim.overall() < sim_var_.sim_pair.overall()) return ret_score;
    cand_aft_check3++;
    Iso2d T_pass = ContourManager::getTFFromConstell(*cm_cand, *cm_tgt_, tmp_pairs2);
    auto cand_it = cand_id_pos_pair_.find(cand_id);
    if (cand_it != cand_id_pos_pair_.end()) {
      candidates_[cand_it->second].addProposal(T_pass, tmp_pairs2, tmp_area_perc);
    } else {
      CandidatePoseData new_cand;
      new_cand.cm_cand_ = cm_cand;
      new_cand.addProposal(T_pass, tmp_pairs2, tmp_area_perc);
      cand_id_pos_pair_.insert({cand_id, (int)candidates_.size()});
      candidates_.emplace_back(std::move(new_cand));
    }
    return ret_score;
  }

  // contour_db.h:494-596
  void tidyUpCandidates() {
    GMMOptConfig gmm_config;
    n_cand_pose = (int)candidates_.size();
    int cnt_to_rm = 0;
    for (auto &candidate : candidates_) {
      int idx_sel = 0;
      for (size_t i = 0; i < candidate.anch_props_.size(); i++) {
        std::vector<float> lev_perc(cm_tgt_->getConfig().lv_grads_.size(), 0);
        for (const auto &pr : candidate.anch_props_[i].constell_) lev_perc[pr.first.level] += pr.second;
        float perc = 0;
        for (int j = 0; j < NUM_BIN_KEY_LAYER; j++) perc += LAYER_AREA_WEIGHTS[j] * lev_perc[DIST_BIN_LAYERS[j]];
        candidate.anch_props_[i].area_perc_ = perc;
        if (candidate.anch_props_[i].vote_cnt_ > candidate.anch_props_[idx_sel].vote_cnt_) idx_sel = i;
      }
      std::swap(candidate.anch_props_[0], candidate.anch_props_[idx_sel]);
      if (candidate.anch_props_[0].area_perc_ < sim_var_.sim_post.area_perc) {
        cnt_to_rm++;
        continue;
      }
      double neg_est_trans_norm2d =
          -ConstellCorrelation::getEstSensTF(candidate.anch_props_[0].T_delta_, cm_tgt_->getConfig()).translation().norm();
      if (neg_est_trans_norm2d < sim_var_.sim_post.neg_est_dist) {
        cnt_to_rm++;
        continue;
      }
      std::unique_ptr<ConstellCorrelation> corr_est(new ConstellCorrelation(gmm_config));
      auto corr_score_init = (float)corr_est->initProblem(*(candidate.cm_cand_), *cm_tgt_, candidate.anch_props_[0].T_delta_);
      if (corr_score_init < sim_var_.sim_post.correlation) {
        cnt_to_rm++;
        continue;
      }
      candidate.corr_est_ = std::move(corr_est);
    }
    int p1 = 0, p2 = candidates_.size() - 1;
    while (p1 <= p2) {
      if (!candidates_[p1].corr_est_ && candidates_[p2].corr_est_) {
        std::swap(candidates_[p1], candidates_[p2]);
        p1++;
        p2--;
      } else {
        if (candidates_[p1].corr_est_) p1++;
        if (!candidates_[p2].corr_est_) p2--;
      }
    }
    candidates_.erase(candidates_.begin() + p2 + 1, candidates_.end());
  }

  // contour_db.h:604-648
  int fineOptimize(int max_fine_opt, std::vector<std::shared_ptr<const ContourManager>> &res_cand, std::vector<double> &res_corr,
                   std::vector<Iso2d> &res_T) {
    res_cand.clear();
    res_corr.clear();
    res_T.clear();
    if (candidates_.empty()) return 0;
    std::sort(candidates_.begin(), candidates_.end(), [&](const CandidatePoseData &d1, const CandidatePoseData &d2) {
      return d1.anch_props_[0].correlation_ > d2.anch_props_[0].correlation_;
    });
    int pre_sel_size = std::min(max_fine_opt, (int)candidates_.size());
    for (int i = 0; i < pre_sel_size; i++) {
      auto tmp_res = candidates_[i].corr_est_->calcCorrelation();
      candidates_[i].anch_props_[0].correlation_ = tmp_res.first;
      candidates_[i].anch_props_[0].T_delta_ = tmp_res.second;
    }
    std::sort(candidates_.begin(), candidates_.begin() + pre_sel_size, [&](const CandidatePoseData &d1, const CandidatePoseData &d2) {
      return d1.anch_props_[0].correlation_ > d2.anch_props_[0].correlation_;
    });
    int ret_size = 1;
    for (int i = 0; i < ret_size; i++) {
      res_cand.emplace_back(candidates_[i].cm_cand_);
      res_corr.emplace_back(candidates_[i].anch_props_[0].correlation_);
      res_T.emplace_back(candidates_[i].anch_props_[0].T_delta_);
    }
    return ret_size;
  }
};

// contour_db.h:658-669
struct ContourDBConfig {
  int nnk_ = 50;
  int max_fine_opt_ = 10;
  std::vector<int> q_levels_;
  ContourSimThresConfig cont_sim_cfg_;
  TreeBucketConfig tb_cfg_;
};

// the five stage timers of tools/bm_util.h SequentialTimeProfiler as used by the reference
// (batch_bin_test.cpp:131-134,236; contour_db.h:729,755,763,772,784,787)
struct StageTimers {
  double make_bev = 0, knn_search = 0, constell = 0, l2_opt = 0, update_db = 0;
  static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
};

// contour_db.h:673-845
class ContourDB {
 public:
  const ContourDBConfig cfg_;
  std::vector<LayerDB> layer_db_;
  std::vector<std::shared_ptr<const ContourManager>> all_bevs_;
  StageTimers *timers = nullptr;

  ContourDB(const ContourDBConfig &config) : cfg_(config) {
    for (size_t i = 0; i < cfg_.q_levels_.size(); i++) layer_db_.emplace_back(TreeBucketConfig(cfg_.tb_cfg_));
  }

  struct QueryDebug {
    std::vector<std::vector<std::pair<IndexOfKey, KeyFloatType>>> knn;  // [q_level * piv + seq]
    int chk1 = 0, chk2 = 0, chk3 = 0, n_cand_pose = 0, n_cand_tidy = 0;
  };

  // contour_db.h:698-811
  void queryRangedKNN(const std::shared_ptr<const ContourManager> &q_ptr, const CandidateScoreEnsemble &thres_lb,
                      const CandidateScoreEnsemble &thres_ub, std::vector<std::shared_ptr<const ContourManager>> &cand_ptrs,
                      std::vector<double> &cand_corr, std::vector<Iso2d> &cand_tf, QueryDebug *dbg = nullptr) const {
    cand_ptrs.clear();
    cand_corr.clear();
    cand_tf.clear();
    CandidateManager cand_mng(q_ptr, thres_lb, thres_ub);
    if (dbg) dbg->knn.assign(cfg_.q_levels_.size() * q_ptr->getConfig().piv_firsts_, {});
    for (size_t ll = 0; ll < cfg_.q_levels_.size(); ll++) {
      const std::vector<BCI> &q_bcis = q_ptr->getLevBCI(cfg_.q_levels_[ll]);
      std::vector<RetrievalKey> q_keys = q_ptr->getLevRetrievalKey(cfg_.q_levels_[ll]);
      for (size_t seq = 0; seq < q_bcis.size(); seq++) {
        if (q_keys[seq].sum() != 0) {
          double t0 = StageTimers::now();
          std::vector<std::pair<IndexOfKey, KeyFloatType>> tmp_res;
          KeyFloatType key_bounds[3][2];
          key_bounds[0][0] = q_keys[seq][0] * 0.8;
          key_bounds[0][1] = q_keys[seq][0] / 0.8;
          key_bounds[1][0] = q_keys[seq][1] * 0.8;
          key_bounds[1][1] = q_keys[seq][1] / 0.8;
          key_bounds[2][0] = q_keys[seq][2] * 0.8 * 0.75;
          key_bounds[2][1] = q_keys[seq][2] / (0.8 * 0.75);
          KeyFloatType dist_ub = 1e6;
          dist_ub = std::max((q_keys[seq][0] - key_bounds[0][0]) * (q_keys[seq][0] - key_bounds[0][0]),
                             (q_keys[seq][0] - key_bounds[0][1]) * (q_keys[seq][0] - key_bounds[0][1])) +
                    std::max((q_keys[seq][1] - key_bounds[1][0]) * (q_keys[seq][1] - key_bounds[1][0]),
                             (q_keys[seq][1] - key_bounds[1][1]) * (q_keys[seq][1] - key_bounds[1][1])) +
                    std::max((q_keys[seq][2] - key_bounds[2][0]) * (q_keys[seq][2] - key_bounds[2][0]),
                             (q_keys[seq][2] - key_bounds[2][1]) * (q_keys[seq][2] - key_bounds[2][1]));
          layer_db_[ll].layerKNNSearch(q_keys[seq], cfg_.nnk_, dist_ub, tmp_res);
          double t1 = StageTimers::now();
          if (timers) timers->knn_search += t1 - t0;
          if (dbg) dbg->knn[ll * q_ptr->getConfig().piv_firsts_ + seq] = tmp_res;
          for (const auto &sear_res : tmp_res) {
            cand_mng.checkCandWithHint(all_bevs_[sear_res.first.gidx],
                                       ConstellationPair(cfg_.q_levels_[ll], sear_res.first.seq, seq), cfg_.cont_sim_cfg_);
          }
          if (timers) timers->constell += StageTimers::now() - t1;
        }
      }
    }
    std::vector<std::shared_ptr<const ContourManager>> res_cand_ptr;
    std::vector<double> res_corr;
    std::vector<Iso2d> res_T;
    double t2 = StageTimers::now();
    cand_mng.tidyUpCandidates();
    if (dbg) {
      dbg->chk1 = cand_mng.cand_aft_check1;
      dbg->chk2 = cand_mng.cand_aft_check2;
      dbg->chk3 = cand_mng.cand_aft_check3;
      dbg->n_cand_pose = cand_mng.n_cand_pose;
      dbg->n_cand_tidy = (int)cand_mng.candidates_.size();
    }
    int num_best_cands = cand_mng.fineOptimize(cfg_.max_fine_opt_, res_cand_ptr, res_corr, res_T);
    if (timers) timers->l2_opt += StageTimers::now() - t2;
    for (int i = 0; i < num_best_cands; i++) {
      cand_ptrs.emplace_back(res_cand_ptr[i]);
      cand_corr.emplace_back(res_corr[i]);
      cand_tf.emplace_back(res_T[i]);
    }
  }

  // contour_db.h:814-824
  void addScan(const std::shared_ptr<ContourManager> &added, double curr_timestamp) {
    for (size_t ll = 0; ll < cfg_.q_levels_.size(); ll++) {
      int seq = 0;
      for (const auto &permu_key : added->getLevRetrievalKey(cfg_.q_levels_[ll])) {
        if (permu_key.sum() != 0)
          layer_db_[ll].pushBuffer(permu_key, curr_timestamp, IndexOfKey(all_bevs_.size(), cfg_.q_levels_[ll], seq));
        seq++;
      }
    }
    all_bevs_.emplace_back(added);
  }

  // contour_db.h:827-843
  void pushAndBalance(int seed, double curr_timestamp) {
    int idx_t1 = std::abs(seed) % (2 * (layer_db_[0].max_num_backets_ - 2));
    if (idx_t1 > (layer_db_[0].max_num_backets_ - 2)) idx_t1 = 2 * (layer_db_[0].max_num_backets_ - 2) - idx_t1;
    for (size_t ll = 0; ll < cfg_.q_levels_.size(); ll++) layer_db_[ll].rebuild(idx_t1, curr_timestamp);
  }
};

}  // namespace orc
