// Host-side bookkeeping of the retrieval database: which key is searchable from which epoch on,
// and the bucket ranges at every epoch.  Replaces, for the device path, the data-structure side of
//   TreeBucket (contour_db.h:68-156), LayerDB::pushBuffer/rebuild (contour_db.h:184-192,
//   src/cont2/contour_db.cpp:63-317), ContourDB::addScan/pushAndBalance (contour_db.h:814-843).
// No kd-tree is built: the device searches the key matrix exhaustively with the reference's bucket
// visiting rule (k_query.h), so all that has to be reproduced here is the MEMBERSHIP timeline:
//   * a key waits in its bucket's time-ordered buffer and enters the bucket's tree when popped
//     (ts < now - min_elapse_, triggered when the oldest entry is older than max_elapse_ or by a
//     re-balance);
//   * re-balancing moves the top / bottom slice (by key dimension 0) between two adjacent buckets and
//     moves the boundary bucket_ranges_[i+1].
// Elements are key ids (index into the layer's device key matrix); key0[id] is dimension 0.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <vector>

namespace cchost {

const float MAX_BUCKET_VAL = 1000.0f;  // contour_db.h:29
const int NBUCKET = 6;                 // LayerDB::max_num_backets_, contour_db.h:162

struct Bucket {
  float buc_beg, buc_end;
  // What the bucket's kd-tree INDEXES, as an interval of key dimension 0.  The reference (re)builds a bucket's index only
  // when popBufferMax moves something out of the buffer (contour_db.h:121-143, rebuildTree :109-117); a re-balance that
  // appends the neighbour's slice to data_tree_ without such a pop leaves those keys outside the index -- findNeighbors
  // never reaches them -- until the bucket's next rebuild, and a bucket that has never popped has no tree at all
  // (TreeBucket::knnSearch returns at once, contour_db.cpp:387).  The slices are contiguous in dimension 0, so the indexed
  // part of a bucket stays an interval: [idx_lo, idx_hi), empty until the first rebuild, the whole range after every rebuild.
  float idx_lo, idx_hi;
  std::vector<int> tree;                       // data_tree_/gkidx_tree_ (ids)
  std::vector<std::pair<double, int>> buffer;  // (ts, id), ascending ts
  size_t getTreeSize() const { return tree.size(); }
};

struct LayerBook {
  static constexpr double imba_diff_ratio_ = 0.2;  // contour_db.h:161
  static const int min_elem_split_ = 100;          // contour_db.h:160
  double max_elapse, min_elapse;
  Bucket b[NBUCKET];
  float ranges[NBUCKET + 1];
  std::vector<float> key0;        // per key id
  std::vector<int> active_from;   // per key id: first epoch at which the key is in a tree (INT_MAX: not yet)
  std::vector<int> newly_active;  // ids popped during the current pushAndBalance call

  void init(double max_e, double min_e) {
    max_elapse = max_e;
    min_elapse = min_e;
    ranges[0] = -MAX_BUCKET_VAL;
    ranges[NBUCKET] = MAX_BUCKET_VAL;
    b[0].buc_beg = -MAX_BUCKET_VAL;
    b[0].buc_end = MAX_BUCKET_VAL;
    b[0].idx_lo = b[0].idx_hi = -MAX_BUCKET_VAL;  // no tree yet
    for (int i = 1; i < NBUCKET; i++) {
      ranges[i] = MAX_BUCKET_VAL;
      b[i].buc_beg = b[i].buc_end = MAX_BUCKET_VAL;
      b[i].idx_lo = b[i].idx_hi = MAX_BUCKET_VAL;
    }
  }

  // LayerDB::pushBuffer (contour_db.h:184-192); the caller has already checked key.sum() != 0.
  // Returns false when no bucket accepts the key (|key0| >= 1000 or NaN): the key is dropped.
  bool pushBuffer(int id, float k0, double ts) {
    for (int i = 0; i < NBUCKET; i++) {
      if (ranges[i] <= k0 && k0 < ranges[i + 1]) {
        b[i].buffer.emplace_back(ts, id);
        return true;
      }
    }
    return false;
  }

  bool needPopBuffer(const Bucket &t, double curr_ts) const {  // contour_db.h:102-107
    double ts_overflow = curr_ts - max_elapse;
    if (t.buffer.empty() || t.buffer[0].first > ts_overflow) return false;
    return true;
  }

  void popBufferMax(Bucket &t, double curr_ts) {  // contour_db.h:121-143
    double ts_cutoff = curr_ts - min_elapse;
    int gap = 0;
    for (; gap < (int)t.buffer.size(); gap++)
      if (t.buffer[gap].first >= ts_cutoff) break;
    if (gap > 0) {
      for (int i = 0; i < gap; i++) {
        t.tree.push_back(t.buffer[i].second);
        newly_active.push_back(t.buffer[i].second);
      }
      t.buffer.erase(t.buffer.begin(), t.buffer.begin() + gap);
      t.idx_lo = t.buc_beg;  // rebuildTree(): everything in data_tree_ is indexed from now on
      t.idx_hi = t.buc_end;
    }
  }
  // every bucket's index covers its whole range (or the range is empty): the steady state, in which the searches need no
  // per-key test
  bool fullyIndexed() const {
    for (int i = 0; i < NBUCKET; i++)
      if (ranges[i] < ranges[i + 1] && !(b[i].idx_lo <= ranges[i] && b[i].idx_hi >= ranges[i + 1]) && !b[i].tree.empty()) return false;
    return true;
  }

  float k0(int id) const { return key0[id]; }

  // LayerDB::rebuild (src/cont2/contour_db.cpp:63-317), same control flow on ids.
  void rebuild(int idx_t1, double curr_ts) {
    Bucket &tr1 = b[idx_t1], &tr2 = b[idx_t1 + 1];
    bool pb1 = needPopBuffer(tr1, curr_ts), pb2 = needPopBuffer(tr2, curr_ts);
    if (!pb1 && !pb2) return;
    int sz1 = (int)tr1.getTreeSize(), sz2 = (int)tr2.getTreeSize();
    double diff_ratio = 1.0 * std::abs(sz1 - sz2) / std::max(sz1, sz2);
    if (pb1 && !pb2 && (diff_ratio < imba_diff_ratio_ || std::max(sz1, sz2) < min_elem_split_)) {
      popBufferMax(tr1, curr_ts);
      return;
    }
    if (!pb1 && pb2 && (diff_ratio < imba_diff_ratio_ || std::max(sz1, sz2) < min_elem_split_)) {
      popBufferMax(tr2, curr_ts);
      return;
    }
    if (diff_ratio < 0.5 * imba_diff_ratio_) {
      if (pb1) popBufferMax(tr1, curr_ts);
      if (pb2) popBufferMax(tr2, curr_ts);
      return;
    }
    const bool from1 = sz1 > sz2;
    Bucket &big = from1 ? tr1 : tr2;
    Bucket &small = from1 ? tr2 : tr1;
    const int szb = from1 ? sz1 : sz2, szs = from1 ? sz2 : sz1;
    int to_move_max = int((szb - szs + imba_diff_ratio_ * szs) / (2 - imba_diff_ratio_));
    int to_move_mid = int((szb - szs) / 2.0);
    int to_move_min = std::max(0, int((szb - szs - imba_diff_ratio_ * szb) / (2 - imba_diff_ratio_)));
    // The reference sorts the whole big tree by key dimension 0 (std::sort of an index permutation, ascending when the
    // first bucket gives, descending otherwise).  Everything that follows only reads the sorted values at positions
    // >= szb - to_move_max - 1, and which elements move is decided by their VALUES (the cut is placed at a value change),
    // so only that tail is put in order here: a selection plus a sort of the tail, on the values themselves.  The order
    // of the ids inside a tree never shows (membership, sizes and ranges do).
    const int m_tail = std::min(szb, std::max(to_move_max, to_move_mid + 1) + 1);
    std::vector<float> vals(szb);
    for (int i = 0; i < szb; i++) vals[i] = k0(big.tree[i]);
    if (from1) {
      std::nth_element(vals.begin(), vals.begin() + (szb - m_tail), vals.end());
      std::sort(vals.begin() + (szb - m_tail), vals.end());
    } else {
      std::nth_element(vals.begin(), vals.begin() + (szb - m_tail), vals.end(), std::greater<float>());
      std::sort(vals.begin() + (szb - m_tail), vals.end(), std::greater<float>());
    }
    auto val = [&](int pos) { return vals[pos]; };  // pos >= szb - m_tail
    int num_to_move = 0;
    float split_val = tr1.buc_end;
    if (to_move_mid <= 0 || to_move_mid >= szb) {
      // the reference indexes one past the end here (undefined behaviour, only reachable with two
      // tiny trees that both need a pop); fall through to the "cannot split" branch
    } else if (val(szb - to_move_mid) != val(szb - to_move_mid - 1)) {
      num_to_move = to_move_mid;
      split_val = from1 ? val(szb - to_move_mid) : val(szb - to_move_mid - 1);
    } else {
      float contagious_val = val(szb - to_move_mid);
      int i = to_move_mid - 1;
      for (; i > to_move_min; i--) {
        if (val(szb - i) != contagious_val) {
          num_to_move = i;
          split_val = from1 ? val(szb - i) : contagious_val;
          break;
        }
      }
      if (num_to_move == 0) {
        i = to_move_mid + 1;
        for (; i < to_move_max; i++) {
          if (val(szb - i) != contagious_val) {
            num_to_move = i - 1;
            split_val = from1 ? contagious_val : val(szb - i);
            break;
          }
        }
      }
    }
    if (num_to_move == 0) {
      if (from1) {
        popBufferMax(tr1, curr_ts);
        if (pb2) popBufferMax(tr2, curr_ts);
      } else {
        if (pb1) popBufferMax(tr1, curr_ts);
        popBufferMax(tr2, curr_ts);
      }
      return;
    }
    // the num_to_move extreme elements change trees: exactly the ones on the far side of split_val
    {
      auto moved = [&](int id) { return from1 ? (k0(id) >= split_val) : (k0(id) < split_val); };
      size_t keep = 0;
      for (size_t i = 0; i < big.tree.size(); i++) {
        const int id = big.tree[i];
        if (moved(id))
          small.tree.push_back(id);
        else
          big.tree[keep++] = id;
      }
      big.tree.resize(keep);
    }
    {
      auto moved = [&](int id) { return from1 ? (k0(id) >= split_val) : (k0(id) < split_val); };
      int p1 = 0, p2 = (int)big.buffer.size() - 1;
      while (p1 <= p2) {
        if (moved(big.buffer[p1].second) && !moved(big.buffer[p2].second)) {
          std::swap(big.buffer[p1], big.buffer[p2]);
          p1++;
          p2--;
        } else {
          if (moved(big.buffer[p2].second)) p2--;
          if (!moved(big.buffer[p1].second)) p1++;
        }
      }
      int sz_rem = p2 + 1;
      small.buffer.insert(small.buffer.end(), big.buffer.begin() + sz_rem, big.buffer.end());
      big.buffer.erase(big.buffer.begin() + sz_rem, big.buffer.end());
    }
    tr1.buc_end = tr2.buc_beg = split_val;
    ranges[idx_t1 + 1] = split_val;
    // the giving bucket's index loses the slice (what the reference's stale index does with it until the bucket's next
    // rebuild is undefined behaviour: it holds positions of a vector that has been compacted); the receiving bucket's index
    // does not gain it before ITS next rebuild
    if (from1)
      tr1.idx_hi = std::min(tr1.idx_hi, split_val);
    else
      tr2.idx_lo = std::max(tr2.idx_lo, split_val);
    auto by_ts = [](const std::pair<double, int> &x, const std::pair<double, int> &y) { return x.first < y.first; };
    std::sort(tr1.buffer.begin(), tr1.buffer.end(), by_ts);
    std::sort(tr2.buffer.begin(), tr2.buffer.end(), by_ts);
    popBufferMax(tr1, curr_ts);
    popBufferMax(tr2, curr_ts);
  }
};

// ContourDB::pushAndBalance bucket choice (contour_db.h:828-830)
inline int balance_index(int seed) {
  int idx_t1 = std::abs(seed) % (2 * (NBUCKET - 2));
  if (idx_t1 > (NBUCKET - 2)) idx_t1 = 2 * (NBUCKET - 2) - idx_t1;
  return idx_t1;
}

}  // namespace cchost
