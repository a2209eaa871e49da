// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// Small fixed-size algebra standing in for the Eigen types the reference uses
// (include/cont2/contour.h:26-29: V2F, M2F, V2D, M2D; Eigen::Isometry2d).
// Operation order follows Eigen 3.3 coefficient-wise evaluation; see DESIGN.md "oracle pinning".
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

// SENSITIVITY KNOBS (tests only, all off by default).  The pieces of the reference that live in third-party code absent
// from the tree are restated here from their published algorithms (see the header of cont2_oracle.cpp): the order in
// which OpenCV numbers connected components, and Ceres' line-search minimiser.  These knobs perturb exactly those
// pieces so that a test can show how much -- how little -- the end result depends on them.
struct Variant {
  unsigned label_shuffle_seed = 0;  // != 0: the components of a level are numbered in a seeded random order instead of the
                                    // first-2x2-block order (changes which of two equal-size contours sorts first)
  int lbfgs_max_iterations = 10;    // correlation.h:215 uses 10
  double wolfe_sufficient_decrease = 1e-4, wolfe_curvature = 0.9;  // Ceres defaults
};
inline Variant &variant() {
  static Variant v;
  return v;
}


struct V2F {
  float x = 0, y = 0;
  V2F() = default;
  V2F(float a, float b) : x(a), y(b) {}
  V2F operator-(const V2F &o) const { return {x - o.x, y - o.y}; }
  V2F operator+(const V2F &o) const { return {x + o.x, y + o.y}; }
  float squaredNorm() const { return x * x + y * y; }
  float norm() const { return std::sqrt(squaredNorm()); }
  // Eigen MatrixBase::normalized(): z = squaredNorm(); z>0 ? v / sqrt(z) : v
  V2F normalized() const {
    float z = squaredNorm();
    if (z > 0.f) {
      float s = std::sqrt(z);
      return {x / s, y / s};
    }
    return *this;
  }
  float dot(const V2F &o) const { return x * o.x + y * o.y; }
};

// 2x2 float matrix, a[r][c]
struct M2F {
  float a[2][2] = {{0, 0}, {0, 0}};
  static M2F Identity() {
    M2F m;
    m.a[0][0] = m.a[1][1] = 1.f;
    return m;
  }
  V2F col(int c) const { return {a[0][c], a[1][c]}; }
};

struct V2D {
  double x = 0, y = 0;
  V2D() = default;
  V2D(double a, double b) : x(a), y(b) {}
  V2D operator-(const V2D &o) const { return {x - o.x, y - o.y}; }
  V2D operator+(const V2D &o) const { return {x + o.x, y + o.y}; }
  V2D operator*(double s) const { return {x * s, y * s}; }
  V2D operator/(double s) const { return {x / s, y / s}; }
  double norm() const { return std::sqrt(x * x + y * y); }
};

struct M2D {
  double a[2][2] = {{0, 0}, {0, 0}};
};

// Eigen::Isometry2d: linear part L (2x2) and translation t.
struct Iso2d {
  double l[2][2] = {{1, 0}, {0, 1}};
  double t[2] = {0, 0};
  static Iso2d Identity() { return Iso2d(); }
  // setIdentity(); rotate(ang); pretranslate(tr)   (contour_mng.h:1271-1274)
  static Iso2d fromAngTrans(double ang, const V2D &tr) {
    Iso2d r;
    double c = std::cos(ang), s = std::sin(ang);
    r.l[0][0] = c;
    r.l[0][1] = -s;
    r.l[1][0] = s;
    r.l[1][1] = c;
    r.t[0] = tr.x;
    r.t[1] = tr.y;
    return r;
  }
  double operator()(int r, int c) const { return c < 2 ? l[r][c] : t[r]; }
  V2D translation() const { return {t[0], t[1]}; }
  double angle() const { return std::atan2(l[1][0], l[0][0]); }
  V2D apply(const V2D &p) const {
    return {l[0][0] * p.x + l[0][1] * p.y + t[0], l[1][0] * p.x + l[1][1] * p.y + t[1]};
  }
  // Transform<Isometry>::inverse(): L^T, -L^T t
  Iso2d inverse() const {
    Iso2d r;
    r.l[0][0] = l[0][0];
    r.l[0][1] = l[1][0];
    r.l[1][0] = l[0][1];
    r.l[1][1] = l[1][1];
    r.t[0] = -(r.l[0][0] * t[0] + r.l[0][1] * t[1]);
    r.t[1] = -(r.l[1][0] * t[0] + r.l[1][1] * t[1]);
    return r;
  }
  Iso2d operator*(const Iso2d &o) const {
    Iso2d r;
    for (int i = 0; i < 2; i++) {
      for (int j = 0; j < 2; j++) r.l[i][j] = l[i][0] * o.l[0][j] + l[i][1] * o.l[1][j];
      r.t[i] = l[i][0] * o.t[0] + l[i][1] * o.t[1] + t[i];
    }
    return r;
  }
};

// ---- Eigen::SelfAdjointEigenSolver<Matrix2f>::compute (Eigen 3.3.x), restated. --------------
// contour.h:165-172 constructs the solver from a dense symmetric 2x2 float matrix.
// Steps: scale by max|coeff| of the lower triangle, (trivial) tridiagonalisation with Q = I,
// implicit symmetric QR steps with Wilkinson shift, ascending sort.
// evals[0] <= evals[1]; evecs.a[r][c], column c belongs to evals[c].
inline void selfAdjointEigen2f(const M2F &m, float evals[2], M2F &evecs) {
  float d0 = m.a[0][0], d1 = m.a[1][1], e = m.a[1][0];
  float scale = std::fabs(d0);
  if (std::fabs(e) > scale) scale = std::fabs(e);
  if (std::fabs(d1) > scale) scale = std::fabs(d1);
  if (scale == 0.f) scale = 1.f;
  d0 /= scale;
  e /= scale;
  d1 /= scale;
  float q[2][2] = {{1.f, 0.f}, {0.f, 1.f}};
  const float considerAsZero = 1.17549435e-38f;     // numeric_limits<float>::min()
  const float precision = 2.f * 1.1920929e-07f;     // 2 * epsilon
  int iter = 0;
  const int maxIter = 30 * 2;
  bool ok = true;
  while (true) {
    if (std::fabs(e) <= (std::fabs(d0) + std::fabs(d1)) * precision || std::fabs(e) <= considerAsZero)
      e = 0.f;
    if (e == 0.f) break;
    iter++;
    if (iter > maxIter) {
      ok = false;
      break;
    }
    // tridiagonal_qr_step, start = 0, end = 1
    float td = (d0 - d1) * 0.5f;
    float mu = d1;
    if (td == 0.f) {
      mu -= std::fabs(e);
    } else {
      float e2 = e * e;
      // numext::hypot(td, e)
      float ax = std::fabs(td), ay = std::fabs(e), p, qp;
      if (ax > ay) {
        p = ax;
        qp = ay / p;
      } else {
        p = ay;
        qp = ax / p;
      }
      float h = (p == 0.f) ? 0.f : p * std::sqrt(1.f + qp * qp);
      if (e2 == 0.f)
        mu -= (e / (td + (td > 0.f ? 1.f : -1.f))) * (e / h);
      else
        mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = d0 - mu;
    float z = e;
    // JacobiRotation::makeGivens(x, z)
    float c, s;
    if (z == 0.f) {
      c = x < 0.f ? -1.f : 1.f;
      s = 0.f;
    } else if (x == 0.f) {
      c = 0.f;
      s = z < 0.f ? 1.f : -1.f;
    } else if (std::fabs(x) > std::fabs(z)) {
      float t = z / x;
      float u = std::sqrt(1.f + t * t);
      if (x < 0.f) u = -u;
      c = 1.f / u;
      s = -t * c;
    } else {
      float t = x / z;
      float u = std::sqrt(1.f + t * t);
      if (z < 0.f) u = -u;
      s = -1.f / u;
      c = -t * s;
    }
    // T = G' T G
    float sdk = s * d0 + c * e;
    float dkp1 = s * e + c * d1;
    d0 = c * (c * d0 - s * e) - s * (c * e - s * d1);
    d1 = s * sdk + c * dkp1;
    e = c * sdk - s * dkp1;
    // Q = Q * G  (applyOnTheRight(0,1,rot)): x' = c x - s y ; y' = s x + c y
    for (int i = 0; i < 2; i++) {
      float xi = q[i][0], yi = q[i][1];
      q[i][0] = c * xi - s * yi;
      q[i][1] = s * xi + c * yi;
    }
  }
  (void)ok;
  // ascending sort (selection sort with column swap)
  if (d1 < d0) {
    float t = d0;
    d0 = d1;
    d1 = t;
    for (int i = 0; i < 2; i++) {
      float tt = q[i][0];
      q[i][0] = q[i][1];
      q[i][1] = tt;
    }
  }
  evals[0] = d0 * scale;
  evals[1] = d1 * scale;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) evecs.a[i][j] = q[i][j];
}

}  // namespace orc
