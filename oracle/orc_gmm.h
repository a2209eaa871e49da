// ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// CPU restatement of include/cont2/correlation.h (GMMPair, ConstellCorrelation) and of the
// Ceres 2.x pieces it calls: AutoDiffFirstOrderFunction<GMMPair,3> (forward-mode Jet<double,3>),
// GradientProblemSolver with default options + max_num_iterations = 10 (correlation.h:213-217),
// i.e. LineSearchMinimizer + LBFGS direction (rank 20, no eigenvalue scaling) + WolfeLineSearch
// with CUBIC interpolation.  Ceres is not available in this image and not vendored by the
// reference: the restatement follows Ceres' published algorithm and defaults from memory of the
// 2.x sources -- PARITY UNPINNED for this third-party part.
#pragma once
#include <array>
#include <cmath>
#include <deque>
#include <vector>

#include "orc_contour.h"

namespace orc {

// ---------------------------------------------------------------- Jet<double,3> ------------
struct Jet3 {
  double a;
  double v[3];
  Jet3() : a(0), v{0, 0, 0} {}
  Jet3(double s) : a(s), v{0, 0, 0} {}
  Jet3(double s, int k) : a(s), v{0, 0, 0} { v[k] = 1.0; }
};
inline Jet3 operator+(const Jet3 &f, const Jet3 &g) {
  Jet3 h;
  h.a = f.a + g.a;
  for (int i = 0; i < 3; i++) h.v[i] = f.v[i] + g.v[i];
  return h;
}
inline Jet3 operator-(const Jet3 &f, const Jet3 &g) {
  Jet3 h;
  h.a = f.a - g.a;
  for (int i = 0; i < 3; i++) h.v[i] = f.v[i] - g.v[i];
  return h;
}
inline Jet3 operator-(const Jet3 &f) {
  Jet3 h;
  h.a = -f.a;
  for (int i = 0; i < 3; i++) h.v[i] = -f.v[i];
  return h;
}
inline Jet3 operator*(const Jet3 &f, const Jet3 &g) {
  Jet3 h;
  h.a = f.a * g.a;
  for (int i = 0; i < 3; i++) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return h;
}
inline Jet3 operator/(const Jet3 &f, const Jet3 &g) {
  // ceres/jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; (f.v - f_a_by_g_a * g.v) * g_a_inverse
  Jet3 h;
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  h.a = f_a_by_g_a;
  for (int i = 0; i < 3; i++) h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return h;
}
inline Jet3 jsqrt(const Jet3 &f) {
  Jet3 h;
  const double tmp = std::sqrt(f.a);
  const double two_a_inverse = 1.0 / (2.0 * tmp);
  h.a = tmp;
  for (int i = 0; i < 3; i++) h.v[i] = two_a_inverse * f.v[i];
  return h;
}
inline Jet3 jexp(const Jet3 &f) {
  Jet3 h;
  const double tmp = std::exp(f.a);
  h.a = tmp;
  for (int i = 0; i < 3; i++) h.v[i] = tmp * f.v[i];
  return h;
}
inline Jet3 jsin(const Jet3 &f) {
  Jet3 h;
  h.a = std::sin(f.a);
  const double c = std::cos(f.a);
  for (int i = 0; i < 3; i++) h.v[i] = c * f.v[i];
  return h;
}
inline Jet3 jcos(const Jet3 &f) {
  Jet3 h;
  h.a = std::cos(f.a);
  const double s = -std::sin(f.a);
  for (int i = 0; i < 3; i++) h.v[i] = s * f.v[i];
  return h;
}
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jexp(double x) { return std::exp(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jcos(double x) { return std::cos(x); }

// correlation.h:15-20
struct GMMOptConfig {
  double min_area_perc_ = 0.95;
  std::vector<int> levels_ = {1, 2, 3, 4};
  double cov_dilate_scale_ = 2.0;
};

// correlation.h:23-154
struct GMMPair {
  struct GMMEllipse {
    M2D cov_;
    V2D mu_;
    double w_;
  };
  std::vector<std::vector<GMMEllipse>> ellipses_src, ellipses_tgt;
  std::vector<std::vector<std::pair<int, int>>> selected_pair_idx_;
  double auto_corr_src_{}, auto_corr_tgt_{};
  const double scale_;

  static double det2(const M2D &m) { return m.a[0][0] * m.a[1][1] - m.a[1][0] * m.a[0][1]; }
  // exp(-0.5 * mu^T * cov^-1 * mu) with Eigen's evaluation order ((-0.5*mu^T) * inv) * mu
  static double quadExp(const M2D &cov, const V2D &mu) {
    double d = det2(cov), invdet = 1.0 / d;
    double i00 = cov.a[1][1] * invdet, i10 = -cov.a[1][0] * invdet, i01 = -cov.a[0][1] * invdet, i11 = cov.a[0][0] * invdet;
    double m0 = -0.5 * mu.x, m1 = -0.5 * mu.y;
    double r0 = m0 * i00 + m1 * i10, r1 = m0 * i01 + m1 * i11;
    return std::exp(r0 * mu.x + r1 * mu.y);
  }

  GMMPair(const ContourManager &cm_src, const ContourManager &cm_tgt, const GMMOptConfig &config, const Iso2d &T_init)
      : scale_(config.cov_dilate_scale_) {
    std::vector<std::vector<float>> max_majax_src, max_majax_tgt;
    for (const auto lev : config.levels_) {
      int cnt_src_run = 0, cnt_src_full = cm_src.getLevTotalPix(lev);
      int cnt_tgt_run = 0, cnt_tgt_full = cm_tgt.getLevTotalPix(lev);
      ellipses_src.emplace_back();
      ellipses_tgt.emplace_back();
      max_majax_src.emplace_back();
      max_majax_tgt.emplace_back();
      selected_pair_idx_.emplace_back();
      auto push = [](std::vector<GMMEllipse> &dst, std::vector<float> &maj, const ContourView &v) {
        GMMEllipse e;
        M2F c = v.getManualCov();
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) e.cov_.a[i][j] = (double)c.a[i][j];
        e.mu_ = V2D(v.pos_mean_.x, v.pos_mean_.y);
        e.w_ = double(v.cell_cnt_);
        dst.push_back(e);
        maj.push_back(std::sqrt(v.eig_vals_.y));
      };
      for (const auto &view_ptr : cm_src.getLevContours(lev)) {
        if (cnt_src_run * 1.0 / cnt_src_full >= config.min_area_perc_) break;
        push(ellipses_src.back(), max_majax_src.back(), *view_ptr);
        cnt_src_run += view_ptr->cell_cnt_;
      }
      for (const auto &view_ptr : cm_tgt.getLevContours(lev)) {
        if (cnt_tgt_run * 1.0 / cnt_tgt_full >= config.min_area_perc_) break;
        push(ellipses_tgt.back(), max_majax_tgt.back(), *view_ptr);
        cnt_tgt_run += view_ptr->cell_cnt_;
      }
    }
    for (size_t li = 0; li < ellipses_src.size(); li++)
      for (size_t si = 0; si < ellipses_src[li].size(); si++)
        for (size_t ti = 0; ti < ellipses_tgt[li].size(); ti++) {
          V2D delta_mu = T_init.apply(ellipses_src[li][si].mu_) - ellipses_tgt[li][ti].mu_;
          if (delta_mu.norm() < 3.0 * (max_majax_src[li][si] + max_majax_tgt[li][ti]))
            selected_pair_idx_[li].emplace_back((int)si, (int)ti);
        }
    auto autocorr = [&](const std::vector<GMMEllipse> &es) {
      double acc = 0;
      for (size_t i = 0; i < es.size(); i++)
        for (size_t j = 0; j < es.size(); j++) {
          M2D new_cov;
          for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) new_cov.a[r][c] = scale_ * (es[i].cov_.a[r][c] + es[j].cov_.a[r][c]);
          V2D new_mu = es[i].mu_ - es[j].mu_;
          acc += es[i].w_ * es[j].w_ / std::sqrt(det2(new_cov)) * quadExp(new_cov, new_mu);
        }
      return acc;
    };
    for (size_t li = 0; li < config.levels_.size(); li++) {
      auto_corr_src_ += autocorr(ellipses_src[li]);
      auto_corr_tgt_ += autocorr(ellipses_tgt[li]);
    }
  }

  // correlation.h:125-152
  template <typename T>
  bool operator()(const T *parameters, T *cost) const {
    const T x = parameters[0];
    const T y = parameters[1];
    const T theta = parameters[2];
    T R[2][2] = {{jcos(theta), -jsin(theta)}, {jsin(theta), jcos(theta)}};
    cost[0] = T(0);
    for (size_t li = 0; li < selected_pair_idx_.size(); li++) {
      for (const auto &pr : selected_pair_idx_[li]) {
        const GMMEllipse &es = ellipses_src[li][pr.first], &et = ellipses_tgt[li][pr.second];
        // new_cov = scale_ * (R * cov_s * R^T + cov_t)
        T RC[2][2], RCRt[2][2], new_cov[2][2];
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) RC[i][j] = R[i][0] * T(es.cov_.a[0][j]) + R[i][1] * T(es.cov_.a[1][j]);
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) RCRt[i][j] = RC[i][0] * R[j][0] + RC[i][1] * R[j][1];
        for (int i = 0; i < 2; i++)
          for (int j = 0; j < 2; j++) new_cov[i][j] = T(scale_) * (RCRt[i][j] + T(et.cov_.a[i][j]));
        // new_mu = R * mu_s + t - mu_t
        T new_mu[2] = {R[0][0] * T(es.mu_.x) + R[0][1] * T(es.mu_.y) + x - T(et.mu_.x),
                       R[1][0] * T(es.mu_.x) + R[1][1] * T(es.mu_.y) + y - T(et.mu_.y)};
        T det = new_cov[0][0] * new_cov[1][1] - new_cov[1][0] * new_cov[0][1];
        T invdet = T(1.0) / det;
        T inv[2][2] = {{new_cov[1][1] * invdet, -new_cov[0][1] * invdet}, {-new_cov[1][0] * invdet, new_cov[0][0] * invdet}};
        T m0 = T(-0.5) * new_mu[0], m1 = T(-0.5) * new_mu[1];
        T r0 = m0 * inv[0][0] + m1 * inv[1][0], r1 = m0 * inv[0][1] + m1 * inv[1][1];
        T qua = r0 * new_mu[0] + r1 * new_mu[1];
        cost[0] = cost[0] + T(-et.w_) * T(es.w_) * T(1.0) / jsqrt(det) * jexp(qua);
      }
    }
    return true;
  }

  // AutoDiffFirstOrderFunction<GMMPair,3>::Evaluate
  void evaluate(const double *p, double *cost, double *grad) const {
    if (!grad) {
      (*this)(p, cost);
      return;
    }
    Jet3 jp[3] = {Jet3(p[0], 0), Jet3(p[1], 1), Jet3(p[2], 2)};
    Jet3 jc;
    (*this)(jp, &jc);
    *cost = jc.a;
    for (int i = 0; i < 3; i++) grad[i] = jc.v[i];
  }
};

// ------------------------------------------------------------------ Ceres restatement ------
namespace ceres_like {

struct FunctionSample {
  double x = 0;
  double vector_x[3] = {0, 0, 0};
  bool vector_x_is_valid = false;
  double value = 0;
  bool value_is_valid = false;
  double vector_gradient[3] = {0, 0, 0};
  bool vector_gradient_is_valid = false;
  double gradient = 0;
  bool gradient_is_valid = false;
};

struct LineSearchFunction {
  const GMMPair *f;
  double position[3], direction[3];
  int n_eval = 0;
  void Init(const double *pos, const double *dir) {
    for (int i = 0; i < 3; i++) {
      position[i] = pos[i];
      direction[i] = dir[i];
    }
  }
  void Evaluate(double x, bool /*evaluate_gradient*/, FunctionSample *out) {
    out->x = x;
    for (int i = 0; i < 3; i++) out->vector_x[i] = position[i] + x * direction[i];
    out->vector_x_is_valid = true;
    f->evaluate(out->vector_x, &out->value, out->vector_gradient);
    n_eval++;
    out->value_is_valid = std::isfinite(out->value);
    bool gfin = std::isfinite(out->vector_gradient[0]) && std::isfinite(out->vector_gradient[1]) &&
                std::isfinite(out->vector_gradient[2]);
    out->vector_gradient_is_valid = out->value_is_valid && gfin;
    out->gradient = direction[0] * out->vector_gradient[0] + direction[1] * out->vector_gradient[1] +
                    direction[2] * out->vector_gradient[2];
    out->gradient_is_valid = out->vector_gradient_is_valid;
  }
  double DirectionInfinityNorm() const {
    return std::max(std::fabs(direction[0]), std::max(std::fabs(direction[1]), std::fabs(direction[2])));
  }
};

// polynomial.cc: coefficients highest degree first
inline double EvaluatePolynomial(const std::vector<double> &p, double x) {
  double v = 0.0;
  for (size_t i = 0; i < p.size(); ++i) v = v * x + p[i];
  return v;
}

// Solve the (n x n, n <= 4) interpolation system.  Ceres uses Eigen::FullPivLU; this is
// Gaussian elimination with full pivoting (same pivoting strategy).
inline std::vector<double> solveFullPiv(std::vector<std::vector<double>> A, std::vector<double> b) {
  const int n = (int)b.size();
  std::vector<int> colperm(n);
  for (int i = 0; i < n; i++) colperm[i] = i;
  for (int k = 0; k < n; k++) {
    int pr = k, pc = k;
    double best = -1;
    for (int i = k; i < n; i++)
      for (int j = k; j < n; j++)
        if (std::fabs(A[i][j]) > best) {
          best = std::fabs(A[i][j]);
          pr = i;
          pc = j;
        }
    if (best == 0.0) break;
    std::swap(A[k], A[pr]);
    std::swap(b[k], b[pr]);
    if (pc != k) {
      for (int i = 0; i < n; i++) std::swap(A[i][k], A[i][pc]);
      std::swap(colperm[k], colperm[pc]);
    }
    for (int i = k + 1; i < n; i++) {
      double f = A[i][k] / A[k][k];
      A[i][k] = 0;
      for (int j = k + 1; j < n; j++) A[i][j] -= f * A[k][j];
      b[i] -= f * b[k];
    }
  }
  std::vector<double> y(n, 0.0), x(n, 0.0);
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int j = i + 1; j < n; j++) s -= A[i][j] * y[j];
    y[i] = (A[i][i] != 0.0) ? s / A[i][i] : 0.0;
  }
  for (int i = 0; i < n; i++) x[colperm[i]] = y[i];
  return x;
}

// polynomial.cc FindInterpolatingPolynomial
inline std::vector<double> FindInterpolatingPolynomial(const std::vector<FunctionSample> &samples) {
  int num_constraints = 0;
  for (const auto &s : samples) {
    if (s.value_is_valid) ++num_constraints;
    if (s.gradient_is_valid) ++num_constraints;
  }
  const int degree = num_constraints - 1;
  std::vector<std::vector<double>> lhs(num_constraints, std::vector<double>(num_constraints, 0.0));
  std::vector<double> rhs(num_constraints, 0.0);
  int row = 0;
  for (const auto &sample : samples) {
    if (sample.value_is_valid) {
      for (int j = 0; j <= degree; ++j) lhs[row][j] = std::pow(sample.x, degree - j);
      rhs[row] = sample.value;
      ++row;
    }
    if (sample.gradient_is_valid) {
      for (int j = 0; j < degree; ++j) lhs[row][j] = (degree - j) * std::pow(sample.x, degree - j - 1);
      rhs[row] = sample.gradient;
      ++row;
    }
  }
  return solveFullPiv(lhs, rhs);
}

// polynomial.cc MinimizePolynomial (+ analytic roots of the <= quadratic derivative)
inline void MinimizePolynomial(const std::vector<double> &polynomial, double x_min, double x_max, double *optimal_x,
                               double *optimal_value) {
  *optimal_x = (x_min + x_max) / 2.0;
  *optimal_value = EvaluatePolynomial(polynomial, *optimal_x);
  const double x_min_value = EvaluatePolynomial(polynomial, x_min);
  if (x_min_value < *optimal_value) {
    *optimal_value = x_min_value;
    *optimal_x = x_min;
  }
  const double x_max_value = EvaluatePolynomial(polynomial, x_max);
  if (x_max_value < *optimal_value) {
    *optimal_value = x_max_value;
    *optimal_x = x_max;
  }
  if (polynomial.size() <= 2) return;
  // DifferentiatePolynomial
  const int degree = (int)polynomial.size() - 1;
  std::vector<double> derivative(degree);
  for (int i = 0; i < degree; ++i) derivative[i] = (degree - i) * polynomial[i];
  // RemoveLeadingZeros
  size_t lead = 0;
  while (lead + 1 < derivative.size() && derivative[lead] == 0.0) ++lead;
  std::vector<double> d(derivative.begin() + lead, derivative.end());
  std::vector<double> roots;
  const int dd = (int)d.size() - 1;
  if (dd == 0) {
    // constant: no roots
  } else if (dd == 1) {
    roots.push_back(-d[1] / d[0]);
  } else if (dd == 2) {
    const double a = d[0], b = d[1], c = d[2];
    const double D = b * b - 4 * a * c;
    const double sqrt_D = std::sqrt(std::fabs(D));
    if (D >= 0) {
      if (b >= 0) {
        roots.push_back((-b - sqrt_D) / (2.0 * a));
        roots.push_back((2.0 * c) / (-b - sqrt_D));
      } else {
        roots.push_back((2.0 * c) / (-b + sqrt_D));
        roots.push_back((-b + sqrt_D) / (2.0 * a));
      }
    } else {
      roots.push_back(-b / (2.0 * a));
      roots.push_back(-b / (2.0 * a));
    }
  } else {
    return;  // not reachable with <= 2 samples (cubic at most)
  }
  for (double root : roots) {
    if ((root < x_min) || (root > x_max)) continue;
    const double value = EvaluatePolynomial(polynomial, root);
    if (value < *optimal_value) {
      *optimal_value = value;
      *optimal_x = root;
    }
  }
}

inline void MinimizeInterpolatingPolynomial(const std::vector<FunctionSample> &samples, double x_min, double x_max,
                                            double *optimal_x, double *optimal_value) {
  const std::vector<double> polynomial = FindInterpolatingPolynomial(samples);
  MinimizePolynomial(polynomial, x_min, x_max, optimal_x, optimal_value);
  for (const auto &sample : samples) {
    if ((sample.x < x_min) || (sample.x > x_max)) continue;
    const double value = EvaluatePolynomial(polynomial, sample.x);
    if (value < *optimal_value) {
      *optimal_x = sample.x;
      *optimal_value = value;
    }
  }
}

struct LSOptions {
  double min_step_size = 1e-9;
  double sufficient_decrease = 1e-4;
  double max_step_contraction = 1e-3;
  double min_step_contraction = 0.6;
  int max_num_iterations = 20;
  double sufficient_curvature_decrease = 0.9;
  double max_step_expansion = 10.0;
};

// line_search.cc LineSearch::InterpolatingPolynomialMinimizingStepSize, CUBIC
inline double InterpolatingStep(const FunctionSample &lowerbound, const FunctionSample &previous,
                                const FunctionSample &current, double min_step_size, double max_step_size) {
  if (!current.value_is_valid) return std::min(std::max(current.x * 0.5, min_step_size), max_step_size);
  std::vector<FunctionSample> samples;
  samples.push_back(lowerbound);
  samples.push_back(current);
  if (previous.value_is_valid) samples.push_back(previous);
  double step_size = 0.0, unused_min_value = 0.0;
  MinimizeInterpolatingPolynomial(samples, min_step_size, max_step_size, &step_size, &unused_min_value);
  return step_size;
}

struct LSSummary {
  bool success = false;
  FunctionSample optimal_point;
  int num_iterations = 0;
};

// line_search.cc WolfeLineSearch::BracketingPhase
inline bool BracketingPhase(const LSOptions &opt, LineSearchFunction *function, const FunctionSample &initial_position,
                            double step_size_estimate, FunctionSample *bracket_low, FunctionSample *bracket_high,
                            bool *do_zoom_search, LSSummary *summary) {
  FunctionSample previous = initial_position;
  FunctionSample current;
  const double descent_direction_max_norm = function->DirectionInfinityNorm();
  *do_zoom_search = false;
  *bracket_low = initial_position;
  function->Evaluate(step_size_estimate, true, &current);
  while (true) {
    ++summary->num_iterations;
    if (current.value_is_valid &&
        (current.value > (initial_position.value + opt.sufficient_decrease * initial_position.gradient * current.x) ||
         (previous.value_is_valid && current.value > previous.value))) {
      *do_zoom_search = true;
      *bracket_low = previous;
      *bracket_high = current;
      break;
    }
    if (current.value_is_valid &&
        std::fabs(current.gradient) <= -opt.sufficient_curvature_decrease * initial_position.gradient) {
      *bracket_low = current;
      *bracket_high = current;
      break;
    } else if (current.value_is_valid && current.gradient >= 0) {
      *do_zoom_search = true;
      *bracket_low = current;
      *bracket_high = previous;
      break;
    } else if (summary->num_iterations >= opt.max_num_iterations) {
      *bracket_low = current.value_is_valid && current.value < bracket_low->value ? current : *bracket_low;
      break;
    }
    const double min_step_size = current.value_is_valid ? current.x : previous.x;
    const double max_step_size = current.value_is_valid ? (current.x * opt.max_step_expansion) : current.x;
    const FunctionSample unused_previous;
    const double step_size = InterpolatingStep(previous, unused_previous, current, min_step_size, max_step_size);
    if (step_size * descent_direction_max_norm < opt.min_step_size) return false;
    previous = current.value_is_valid ? current : previous;
    function->Evaluate(step_size, true, &current);
  }
  if (*do_zoom_search && std::fabs(bracket_high->x - bracket_low->x) * descent_direction_max_norm < opt.min_step_size)
    *do_zoom_search = false;
  return true;
}

// line_search.cc WolfeLineSearch::ZoomPhase
inline bool ZoomPhase(const LSOptions &opt, LineSearchFunction *function, const FunctionSample &initial_position,
                      FunctionSample bracket_low, FunctionSample bracket_high, FunctionSample *solution,
                      LSSummary *summary) {
  if (bracket_low.gradient * (bracket_high.x - bracket_low.x) >= 0) {
    solution->value_is_valid = false;
    return false;
  }
  const double descent_direction_max_norm = function->DirectionInfinityNorm();
  while (true) {
    *solution = bracket_low;
    if (summary->num_iterations >= opt.max_num_iterations) return false;
    if (std::fabs(bracket_high.x - bracket_low.x) * descent_direction_max_norm < opt.min_step_size) return false;
    ++summary->num_iterations;
    const FunctionSample &lower_bound_step = bracket_low.x < bracket_high.x ? bracket_low : bracket_high;
    const FunctionSample &upper_bound_step = bracket_low.x < bracket_high.x ? bracket_high : bracket_low;
    const FunctionSample unused_previous;
    const double step_size =
        InterpolatingStep(lower_bound_step, unused_previous, upper_bound_step, lower_bound_step.x, upper_bound_step.x);
    function->Evaluate(step_size, true, solution);
    if (!solution->value_is_valid || !solution->gradient_is_valid) return false;
    if ((solution->value > (initial_position.value + opt.sufficient_decrease * initial_position.gradient * solution->x)) ||
        (solution->value >= bracket_low.value)) {
      bracket_high = *solution;
      continue;
    }
    if (std::fabs(solution->gradient) <= -opt.sufficient_curvature_decrease * initial_position.gradient) {
      break;
    } else if (solution->gradient * (bracket_high.x - bracket_low.x) >= 0) {
      bracket_high = bracket_low;
    }
    bracket_low = *solution;
  }
  return true;
}

// line_search.cc WolfeLineSearch::DoSearch
inline void WolfeSearch(const LSOptions &opt, LineSearchFunction *function, double step_size_estimate,
                        double initial_cost, double initial_gradient, const double *x0, const double *g0,
                        LSSummary *summary) {
  *summary = LSSummary();
  FunctionSample initial_position;
  initial_position.x = 0.0;
  initial_position.value = initial_cost;
  initial_position.value_is_valid = true;
  initial_position.gradient = initial_gradient;
  initial_position.gradient_is_valid = true;
  for (int i = 0; i < 3; i++) {
    initial_position.vector_x[i] = x0[i];
    initial_position.vector_gradient[i] = g0[i];
  }
  initial_position.vector_x_is_valid = true;
  initial_position.vector_gradient_is_valid = true;
  FunctionSample bracket_low, bracket_high;
  bool do_zoom_search = false;
  if (!BracketingPhase(opt, function, initial_position, step_size_estimate, &bracket_low, &bracket_high, &do_zoom_search,
                       summary))
    return;
  if (!do_zoom_search) {
    summary->optimal_point = bracket_low;
    summary->success = true;
    return;
  }
  FunctionSample solution;
  if (!ZoomPhase(opt, function, initial_position, bracket_low, bracket_high, &solution, summary) &&
      !solution.value_is_valid)
    return;
  if (!solution.value_is_valid || solution.value > bracket_low.value)
    summary->optimal_point = bracket_low;
  else
    summary->optimal_point = solution;
  summary->success = true;
}

// low_rank_inverse_hessian.cc + line_search_direction.cc (LBFGS, rank 20, H0 = I)
struct LBFGS {
  static const int kMax = 20;
  double dx[kMax][3], dg[kMax][3], dxdg[kMax];
  std::deque<int> indices;
  int num_corrections = 0;
  bool Update(const double *delta_x, const double *delta_gradient) {
    const double d = delta_x[0] * delta_gradient[0] + delta_x[1] * delta_gradient[1] + delta_x[2] * delta_gradient[2];
    if (d <= 1e-14) return false;
    if (num_corrections == kMax) {
      indices.push_back(indices.front());
      indices.pop_front();
    } else {
      indices.push_back(num_corrections);
      ++num_corrections;
    }
    const int next = indices.back();
    for (int i = 0; i < 3; i++) {
      dx[next][i] = delta_x[i];
      dg[next][i] = delta_gradient[i];
    }
    dxdg[next] = d;
    return true;
  }
  void RightMultiply(const double *gradient, double *sd) const {
    for (int i = 0; i < 3; i++) sd[i] = gradient[i];
    double alpha[kMax];
    for (auto it = indices.rbegin(); it != indices.rend(); ++it) {
      const int k = *it;
      const double alpha_i = (dx[k][0] * sd[0] + dx[k][1] * sd[1] + dx[k][2] * sd[2]) / dxdg[k];
      for (int i = 0; i < 3; i++) sd[i] -= alpha_i * dg[k][i];
      alpha[k] = alpha_i;
    }
    for (auto it = indices.begin(); it != indices.end(); ++it) {
      const int k = *it;
      const double beta = (dg[k][0] * sd[0] + dg[k][1] * sd[1] + dg[k][2] * sd[2]) / dxdg[k];
      for (int i = 0; i < 3; i++) sd[i] += dx[k][i] * (alpha[k] - beta);
    }
  }
};

struct SolveSummary {
  double initial_cost = 0, final_cost = 0;
  int iterations = 0;
  int termination = 0;  // 0 no-convergence(max iter) 1 gradient tol 2 function tol 3 parameter tol -1 failure
  int n_eval = 0;
  std::vector<std::array<double, 3>> iterates;  // x_0 and the accepted point of every iteration (tests: Wolfe / L-BFGS properties)
};

// line_search_minimizer.cc LineSearchMinimizer::Minimize, options of correlation.h:213-217
inline void Solve(const GMMPair &prob, double *parameters, SolveSummary *summary, int max_num_iterations = 10) {
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const int max_num_line_search_direction_restarts = 5;
  LSOptions lsopt;
  lsopt.sufficient_decrease = orc::variant().wolfe_sufficient_decrease;        // defaults unless a sensitivity test says otherwise
  lsopt.sufficient_curvature_decrease = orc::variant().wolfe_curvature;
  if (max_num_iterations == 10) max_num_iterations = orc::variant().lbfgs_max_iterations;
  struct State {
    double cost = 0, gradient[3] = {0, 0, 0}, gradient_squared_norm = 0, gradient_max_norm = 0;
    double search_direction[3] = {0, 0, 0}, directional_derivative = 0, step_size = 0;
  };
  double x[3] = {parameters[0], parameters[1], parameters[2]};
  State current_state, previous_state;
  prob.evaluate(x, &current_state.cost, current_state.gradient);
  summary->n_eval = 1;
  auto norms = [](State &s) {
    s.gradient_squared_norm = s.gradient[0] * s.gradient[0] + s.gradient[1] * s.gradient[1] + s.gradient[2] * s.gradient[2];
    s.gradient_max_norm = std::max(std::fabs(s.gradient[0]), std::max(std::fabs(s.gradient[1]), std::fabs(s.gradient[2])));
  };
  norms(current_state);
  summary->initial_cost = current_state.cost;
  summary->final_cost = current_state.cost;
  summary->iterations = 0;
  summary->iterates.clear();
  summary->iterates.push_back({x[0], x[1], x[2]});
  if (current_state.gradient_max_norm <= gradient_tolerance) {
    summary->termination = 1;
    return;
  }
  LBFGS lbfgs;
  LineSearchFunction lsf;
  lsf.f = &prob;
  int num_line_search_direction_restarts = 0;
  int iteration = 0;
  while (true) {
    if (iteration >= max_num_iterations) {
      summary->termination = 0;
      break;
    }
    iteration++;
    bool line_search_status = true;
    if (iteration == 1) {
      for (int i = 0; i < 3; i++) current_state.search_direction[i] = -current_state.gradient[i];
    } else {
      // LBFGS::NextDirection
      double delta_x[3], delta_g[3];
      for (int i = 0; i < 3; i++) {
        delta_x[i] = previous_state.search_direction[i] * previous_state.step_size;
        delta_g[i] = current_state.gradient[i] - previous_state.gradient[i];
      }
      lbfgs.Update(delta_x, delta_g);
      double sd[3];
      lbfgs.RightMultiply(current_state.gradient, sd);
      for (int i = 0; i < 3; i++) current_state.search_direction[i] = -1.0 * sd[i];
      double dot = current_state.search_direction[0] * current_state.gradient[0] +
                   current_state.search_direction[1] * current_state.gradient[1] +
                   current_state.search_direction[2] * current_state.gradient[2];
      if (dot >= 0.0) line_search_status = false;
    }
    if (!line_search_status && num_line_search_direction_restarts >= max_num_line_search_direction_restarts) {
      summary->termination = -1;
      break;
    } else if (!line_search_status) {
      num_line_search_direction_restarts++;
      lbfgs = LBFGS();
      for (int i = 0; i < 3; i++) current_state.search_direction[i] = -current_state.gradient[i];
    }
    lsf.Init(x, current_state.search_direction);
    current_state.directional_derivative = current_state.gradient[0] * current_state.search_direction[0] +
                                           current_state.gradient[1] * current_state.search_direction[1] +
                                           current_state.gradient[2] * current_state.search_direction[2];
    const double initial_step_size =
        (iteration == 1 || !line_search_status)
            ? std::min(1.0, 1.0 / current_state.gradient_max_norm)
            : std::min(1.0, 2.0 * (current_state.cost - previous_state.cost) / current_state.directional_derivative);
    if (initial_step_size < 0.0) {
      summary->termination = -1;
      break;
    }
    LSSummary ls;
    WolfeSearch(lsopt, &lsf, initial_step_size, current_state.cost, current_state.directional_derivative, x,
                current_state.gradient, &ls);
    if (!ls.success) {
      summary->termination = -1;
      break;
    }
    const FunctionSample &optimal_point = ls.optimal_point;
    current_state.step_size = optimal_point.x;
    previous_state = current_state;
    double x_plus_delta[3];
    for (int i = 0; i < 3; i++) x_plus_delta[i] = optimal_point.vector_x[i];
    current_state.cost = optimal_point.value;
    for (int i = 0; i < 3; i++) current_state.gradient[i] = optimal_point.vector_gradient[i];
    norms(current_state);
    const double x_norm = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    double step_norm = 0;
    for (int i = 0; i < 3; i++) step_norm += (x_plus_delta[i] - x[i]) * (x_plus_delta[i] - x[i]);
    step_norm = std::sqrt(step_norm);
    const double step_size_tolerance = parameter_tolerance * (x_norm + parameter_tolerance);
    const double cost_change = previous_state.cost - current_state.cost;
    for (int i = 0; i < 3; i++) x[i] = x_plus_delta[i];
    summary->iterates.push_back({x[0], x[1], x[2]});
    summary->iterations = iteration;
    summary->final_cost = current_state.cost;
    if (step_norm <= step_size_tolerance) {
      summary->termination = 3;
      break;
    }
    if (current_state.gradient_max_norm <= gradient_tolerance) {
      summary->termination = 1;
      break;
    }
    const double absolute_function_tolerance = function_tolerance * std::fabs(previous_state.cost);
    if (std::fabs(cost_change) <= absolute_function_tolerance) {
      summary->termination = 2;
      break;
    }
  }
  summary->n_eval += lsf.n_eval;
  for (int i = 0; i < 3; i++) parameters[i] = x[i];
}

}  // namespace ceres_like

// correlation.h:157-298
class ConstellCorrelation {
  GMMOptConfig cfg_;
  std::unique_ptr<GMMPair> problem_ptr;
  double auto_corr_src{}, auto_corr_tgt{};
  Iso2d T_best_;

 public:
  ConstellCorrelation() = default;
  explicit ConstellCorrelation(GMMOptConfig cfg) : cfg_(std::move(cfg)) {}

  // correlation.h:175-191
  double initProblem(const ContourManager &cm_src, const ContourManager &cm_tgt, const Iso2d &T_init) {
    T_best_ = T_init;
    problem_ptr.reset(new GMMPair(cm_src, cm_tgt, cfg_, T_init));
    auto_corr_src = problem_ptr->auto_corr_src_;
    auto_corr_tgt = problem_ptr->auto_corr_tgt_;
    return tryProblem(T_init);
  }
  // correlation.h:196-202
  double tryProblem(const Iso2d &T_try) const {
    double parameters[3] = {T_try(0, 2), T_try(1, 2), std::atan2(T_try(1, 0), T_try(0, 0))};
    double cost[1] = {0};
    problem_ptr->evaluate(parameters, cost, nullptr);
    return -cost[0] / std::sqrt(auto_corr_src * auto_corr_tgt);
  }
  // correlation.h:206-238
  std::pair<double, Iso2d> calcCorrelation(ceres_like::SolveSummary *sum_out = nullptr) {
    double parameters[3] = {T_best_(0, 2), T_best_(1, 2), std::atan2(T_best_(1, 0), T_best_(0, 0))};
    ceres_like::SolveSummary summary;
    ceres_like::Solve(*problem_ptr, parameters, &summary, 10);
    if (sum_out) *sum_out = summary;
    T_best_ = Iso2d::fromAngTrans(parameters[2], V2D(parameters[0], parameters[1]));
    double correlation = -summary.final_cost / std::sqrt(auto_corr_src * auto_corr_tgt);
    return {correlation, T_best_};
  }
  // correlation.h:287-296
  static Iso2d getEstSensTF(const Iso2d &T_delta, const ContourManagerConfig &bev_config) {
    Iso2d T_so_ssen = Iso2d::Identity();
    T_so_ssen.t[0] = bev_config.n_row_ / 2 - 0.5;
    T_so_ssen.t[1] = bev_config.n_col_ / 2 - 0.5;
    Iso2d T_to_tsen = T_so_ssen;
    return T_to_tsen.inverse() * T_delta * T_so_ssen;
  }
};

}  // namespace orc
