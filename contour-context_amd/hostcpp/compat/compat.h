// Minimal stand-ins for the third-party types the reference's public API mentions, so a driver written
// against include/cont2/*.h of the reference compiles against this mirror without Eigen / PCL / glog:
//   pcl::PointXYZ, pcl::PointCloud<PointT>(::Ptr/::ConstPtr), Eigen::Isometry2d (the subset the driver uses),
//   CHECK / CHECK_GT (abort like glog).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#define CC_CHECK(cond)                                                              \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      fprintf(stderr, "Check failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__);     \
      abort();                                                                      \
    }                                                                               \
  } while (0)

namespace pcl {
struct PointXYZ {
  float x, y, z;
  float pad_;
};
struct PCLHeader {
  uint64_t stamp = 0;
};
template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  PCLHeader header;
  size_t size() const { return points.size(); }
  void reserve(size_t n) { points.reserve(n); }
  void push_back(const PointT &p) { points.push_back(p); }
};
}  // namespace pcl

namespace Eigen {
// 2-D rigid transform with the accessors the reference driver / evaluator use
struct Isometry2d {
  double m[2][3] = {{1, 0, 0}, {0, 1, 0}};
  static Isometry2d Identity() { return Isometry2d(); }
  void setIdentity() { *this = Isometry2d(); }
  double operator()(int r, int c) const { return m[r][c]; }
  void rotate(double a) {  // linear = linear * R(a)
    const double c = std::cos(a), s = std::sin(a);
    const double l00 = m[0][0], l01 = m[0][1], l10 = m[1][0], l11 = m[1][1];
    m[0][0] = l00 * c + l01 * s;
    m[0][1] = -l00 * s + l01 * c;
    m[1][0] = l10 * c + l11 * s;
    m[1][1] = -l10 * s + l11 * c;
  }
  void pretranslate(double x, double y) {
    m[0][2] += x;
    m[1][2] += y;
  }
  struct Vec2 {
    double x_, y_;
    double x() const { return x_; }
    double y() const { return y_; }
    double norm() const { return std::sqrt(x_ * x_ + y_ * y_); }
  };
  Vec2 translation() const { return {m[0][2], m[1][2]}; }
};
}  // namespace Eigen
