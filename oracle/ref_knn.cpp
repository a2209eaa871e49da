// ORACLE/_ref -- TEST INFRASTRUCTURE ONLY.  Thin C-ABI over the REAL vendored nanoflann of the
// reference (compiled from /root/reference/thirdparty where it lies; nothing is copied).
// Mirrors TreeBucket::knnSearch + MyKNNResSet (include/cont2/contour_db.h:32-52,
// src/cont2/contour_db.cpp:381-403): kd-tree, leaf 10, kNN with a max squared distance.
#include <cstring>
#include <vector>
#include <nanoflann.hpp>
#include "KDTreeVectorOfVectorsAdaptor.h"

struct Key10 {
  enum { SizeAtCompileTime = 10 };
  float array[10];
  float *data() { return array; }
  size_t size() const { return 10; }
  const float &operator[](size_t i) const { return array[i]; }
  float &operator[](size_t i) { return array[i]; }
};
typedef std::vector<Key10> vov_t;
typedef KDTreeVectorOfVectorsAdaptor<vov_t, float> kd_t;

template <typename D, typename I = size_t, typename C = size_t>
class MyKNNResSet : public nanoflann::KNNResultSet<D, I, C> {
 public:
  explicit MyKNNResSet(C capacity_) : nanoflann::KNNResultSet<D, I, C>(capacity_) {}
  void init(I *indices_, D *dists_, D max_dist_metric) {
    this->indices = indices_;
    this->dists = dists_;
    this->count = 0;
    if (this->capacity) this->dists[this->capacity - 1] = max_dist_metric;
  }
};

extern "C" int ref_knn(const float *keys, int n, const float *q, int k, float max_dist_sq, int *idx_out, float *dist_out) {
  vov_t data(n);
  for (int i = 0; i < n; i++) std::memcpy(data[i].array, keys + 10 * i, 40);
  kd_t tree(10, data, 10);
  std::vector<size_t> idx(k, 0);
  std::vector<float> d(k, 1e6f);
  MyKNNResSet<float> rs(k);
  rs.init(&idx[0], &d[0], max_dist_sq);
  tree.index->findNeighbors(rs, q, nanoflann::SearchParams(10));
  int cnt = 0;
  for (int j = 0; j < k; j++) {
    if (d[j] < max_dist_sq) {
      idx_out[cnt] = (int)idx[j];
      dist_out[cnt] = d[j];
      cnt++;
    } else
      break;
  }
  return cnt;
}

// ---- persistent tree: what TreeBucket keeps (data_tree_ + tree_ptr), rebuilt like rebuildTree()
struct RefTree {
  vov_t data;
  kd_t *tree = nullptr;
  ~RefTree() { delete tree; }
};
extern "C" void *refkd_create() { return new RefTree(); }
extern "C" void refkd_free(void *h) { delete (RefTree *)h; }
// TreeBucket::rebuildTree (contour_db.h:109-117): first call constructs the adaptor (which builds
// the index), later calls buildIndex() on the changed data.
extern "C" void refkd_build(void *h, const float *keys, int n) {
  RefTree *t = (RefTree *)h;
  t->data.resize(n);
  for (int i = 0; i < n; i++) std::memcpy(t->data[i].array, keys + 10 * i, 40);
  if (t->tree)
    t->tree->index->buildIndex();
  else
    t->tree = new kd_t(10, t->data, 10);
}
extern "C" void refkd_query(void *h, const float *q, int k, float max_dist_sq, size_t *idx, float *dist) {
  RefTree *t = (RefTree *)h;
  MyKNNResSet<float> rs(k);
  rs.init(idx, dist, max_dist_sq);
  t->tree->index->findNeighbors(rs, q, nanoflann::SearchParams(10));
}
