// Minimal stand-ins for the third-party types the reference's public API mentions, so a driver written
// against include/cont2/*.h of the reference compiles against this mirror without Eigen / PCL / glog:
//   pcl::PointXYZ, pcl::PointCloud<PointT>(::Ptr/::ConstPtr), Eigen::Isometry2d / Isometry3d (the subset the driver and the
//   evaluator use),
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

// glog names the reference's headers and drivers use (abort on failure like glog; no logging back end)
#ifndef CHECK
#define CHECK(cond) CC_CHECK(cond)
#define CHECK_GT(a, b) CC_CHECK((a) > (b))
#define CHECK_GE(a, b) CC_CHECK((a) >= (b))
#define CHECK_LT(a, b) CC_CHECK((a) < (b))
#define CHECK_LE(a, b) CC_CHECK((a) <= (b))
#define CHECK_EQ(a, b) CC_CHECK((a) == (b))
#define CHECK_NE(a, b) CC_CHECK((a) != (b))
#define DCHECK(cond) ((void)0)
#endif
static bool FLAGS_alsologtostderr __attribute__((unused)) = false;
namespace google {
inline void InitGoogleLogging(const char *) {}
}  // namespace google

// ---- images without OpenCV: what ContourManager::getBevImage / getContourImage return and what cv::imwrite(".png")
// of a CV_8U single-channel matrix stores (an 8-bit grey PNG; written here with stored deflate blocks) ----
namespace cc_host {
template <typename T>
struct Image {
  int rows = 0, cols = 0;
  std::vector<T> data;  // row-major
  Image() {}
  Image(int r, int c, T fill) : rows(r), cols(c), data((size_t)r * c, fill) {}
  T &at(int r, int c) { return data[(size_t)r * cols + c]; }
  const T &at(int r, int c) const { return data[(size_t)r * cols + c]; }
  bool empty() const { return data.empty(); }
  // cv::Mat::copyTo(dst(cv::Rect(x, y, cols, rows)))
  void copyTo(Image<T> &dst, int x, int y) const {
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) dst.at(y + r, x + c) = at(r, c);
  }
};
inline uint32_t png_crc(const unsigned char *p, size_t n, uint32_t crc = 0) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  crc ^= 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  return crc ^ 0xFFFFFFFFu;
}
inline bool write_png_gray8(const std::string &path, const Image<unsigned char> &img) {
  auto be32 = [](std::vector<unsigned char> &v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s));
  };
  auto chunk = [&](std::vector<unsigned char> &out, const char *type, const std::vector<unsigned char> &body) {
    be32(out, (uint32_t)body.size());
    std::vector<unsigned char> td(type, type + 4);
    td.insert(td.end(), body.begin(), body.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, png_crc(td.data(), td.size()));
  };
  std::vector<unsigned char> raw;  // scanlines, filter type 0
  raw.reserve((size_t)img.rows * (img.cols + 1));
  for (int r = 0; r < img.rows; r++) {
    raw.push_back(0);
    raw.insert(raw.end(), img.data.begin() + (size_t)r * img.cols, img.data.begin() + (size_t)(r + 1) * img.cols);
  }
  std::vector<unsigned char> z = {0x78, 0x01};  // zlib header, then stored (uncompressed) deflate blocks
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < raw.size(); i++) {
    a = (a + raw[i]) % 65521u;
    b = (b + a) % 65521u;
  }
  size_t pos = 0;
  do {
    const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((unsigned char)(n & 0xFF));
    z.push_back((unsigned char)(n >> 8));
    z.push_back((unsigned char)(~n & 0xFF));
    z.push_back((unsigned char)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  be32(z, (b << 16) | a);
  std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<unsigned char> ihdr;
  be32(ihdr, (uint32_t)img.cols);
  be32(ihdr, (uint32_t)img.rows);
  const unsigned char tail[5] = {8, 0, 0, 0, 0};  // 8 bit, grey, deflate, no filter method, no interlace
  ihdr.insert(ihdr.end(), tail, tail + 5);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}
}  // namespace cc_host

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
  Isometry2d inverse() const {
    Isometry2d o;
    o.m[0][0] = m[0][0];
    o.m[0][1] = m[1][0];
    o.m[1][0] = m[0][1];
    o.m[1][1] = m[1][1];
    o.m[0][2] = -(o.m[0][0] * m[0][2] + o.m[0][1] * m[1][2]);
    o.m[1][2] = -(o.m[1][0] * m[0][2] + o.m[1][1] * m[1][2]);
    return o;
  }
  Isometry2d operator*(const Isometry2d &b) const {
    Isometry2d o;
    for (int i = 0; i < 2; i++) {
      o.m[i][0] = m[i][0] * b.m[0][0] + m[i][1] * b.m[1][0];
      o.m[i][1] = m[i][0] * b.m[0][1] + m[i][1] * b.m[1][1];
      o.m[i][2] = m[i][0] * b.m[0][2] + m[i][1] * b.m[1][2] + m[i][2];
    }
    return o;
  }
};

// 3-vector with the few operations the offline driver uses (batch_bin_test.cpp:143-145)
struct Vector3d {
  double v[3] = {0, 0, 0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
  Vector3d operator*(double s) const { return Vector3d(v[0] * s, v[1] * s, v[2] * s); }
  Vector3d operator/(double s) const { return Vector3d(v[0] / s, v[1] / s, v[2] / s); }
};

// 3-D rigid transform: what the evaluator keeps per scan (ground-truth sensor pose) and evalMetricEst consumes
struct Isometry3d {
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  double t[3] = {0, 0, 0};
  struct Vec3 {
    double v[3];
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
    Vec3 operator-(const Vec3 &o) const { return {{v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}}; }
    double operator()(int i) const { return v[i]; }
  };
  static Isometry3d Identity() { return Isometry3d(); }
  Vec3 translation() const { return {{t[0], t[1], t[2]}}; }
  double operator()(int r, int c) const { return c < 3 ? R[r][c] : t[r]; }
  Isometry3d inverse() const {  // [R^T, -R^T t]
    Isometry3d o;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) o.R[i][j] = R[j][i];
    for (int i = 0; i < 3; i++) o.t[i] = -(o.R[i][0] * t[0] + o.R[i][1] * t[1] + o.R[i][2] * t[2]);
    return o;
  }
  Isometry3d operator*(const Isometry3d &b) const {
    Isometry3d o;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) o.R[i][j] = R[i][0] * b.R[0][j] + R[i][1] * b.R[1][j] + R[i][2] * b.R[2][j];
      o.t[i] = R[i][0] * b.t[0] + R[i][1] * b.t[1] + R[i][2] * b.t[2] + t[i];
    }
    return o;
  }
  // Quaterniond(rotation matrix) followed by rotate(q) on an identity transform: the matrix is re-orthonormalised
  // through a unit quaternion, which is what the reference's pose loader does (eval/evaluator.h:118-121)
  void setRotationViaQuaternion(const double M[3][3]) {
    double w, x, y, z;
    const double tr = M[0][0] + M[1][1] + M[2][2];
    if (tr > 0) {
      double s = std::sqrt(tr + 1.0);
      w = 0.5 * s;
      s = 0.5 / s;
      x = (M[2][1] - M[1][2]) * s;
      y = (M[0][2] - M[2][0]) * s;
      z = (M[1][0] - M[0][1]) * s;
    } else {
      int i = 0;
      if (M[1][1] > M[0][0]) i = 1;
      if (M[2][2] > M[i][i]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      double s = std::sqrt(M[i][i] - M[j][j] - M[k][k] + 1.0);
      double q[3];
      q[i] = 0.5 * s;
      s = 0.5 / s;
      w = (M[k][j] - M[j][k]) * s;
      q[j] = (M[j][i] + M[i][j]) * s;
      q[k] = (M[k][i] + M[i][k]) * s;
      x = q[0];
      y = q[1];
      z = q[2];
    }
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                 tzz = tz * z;
    R[0][0] = 1 - (tyy + tzz);
    R[0][1] = txy - twz;
    R[0][2] = txz + twy;
    R[1][0] = txy + twz;
    R[1][1] = 1 - (txx + tzz);
    R[1][2] = tyz - twx;
    R[2][0] = txz - twy;
    R[2][1] = tyz + twx;
    R[2][2] = 1 - (txx + tyy);
  }
};
}  // namespace Eigen
