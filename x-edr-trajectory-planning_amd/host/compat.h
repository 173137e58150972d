// compat.h -- the small vocabulary the host API needs from Eigen / abseil / eigenmath.
//
// The reference's public API is written against Eigen::ArrayX<double>,
// eigenmath::VectorXd, absl::Status, absl::Span, absl::Time and absl::Duration. None of
// those libraries exist in this image, so the host mirror is written against the
// minimal value types below. They are the ONLY place where the mirror departs from the
// reference's spellings: INTEGRATION.md shows the aliases a maintainer flips to build the
// mirror inside the reference's Bazel workspace (real Eigen/absl types instead of these).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstddef>
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

namespace tpamd {
namespace compat {

// ---- status ----------------------------------------------------------------
enum class StatusCode : int {
  kOk = 0, kCancelled = 1, kUnknown = 2, kInvalidArgument = 3, kDeadlineExceeded = 4,
  kNotFound = 5, kAlreadyExists = 6, kPermissionDenied = 7, kResourceExhausted = 8,
  kFailedPrecondition = 9, kAborted = 10, kOutOfRange = 11, kUnimplemented = 12,
  kInternal = 13, kUnavailable = 14, kDataLoss = 15, kUnauthenticated = 16
};

class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string message) : code_(code), message_(std::move(message)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string &message() const { return message_; }
  std::string ToString() const {
    return ok() ? std::string("OK") : "code " + std::to_string((int)code_) + ": " + message_;
  }

 private:
  StatusCode code_ = StatusCode::kOk;
  std::string message_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(std::string m) { return Status(StatusCode::kInvalidArgument, std::move(m)); }
inline Status FailedPreconditionError(std::string m) { return Status(StatusCode::kFailedPrecondition, std::move(m)); }
inline Status NotFoundError(std::string m) { return Status(StatusCode::kNotFound, std::move(m)); }
inline Status InternalError(std::string m) { return Status(StatusCode::kInternal, std::move(m)); }
inline Status OutOfRangeError(std::string m) { return Status(StatusCode::kOutOfRange, std::move(m)); }
inline Status UnimplementedError(std::string m) { return Status(StatusCode::kUnimplemented, std::move(m)); }
inline Status DeadlineExceededError(std::string m) { return Status(StatusCode::kDeadlineExceeded, std::move(m)); }
inline Status ResourceExhaustedError(std::string m) { return Status(StatusCode::kResourceExhausted, std::move(m)); }

template <typename T>
class StatusOr {
 public:
  StatusOr(const Status &s) : status_(s) {}
  StatusOr(const T &v) : value_(v) {}
  bool ok() const { return status_.ok(); }
  const Status &status() const { return status_; }
  const T &operator*() const { return value_; }
  const T &value() const { return value_; }

 private:
  Status status_;
  T value_{};
};

// ---- span --------------------------------------------------------------------
template <typename T>
class Span {
 public:
  Span() = default;
  Span(T *data, size_t size) : data_(data), size_(size) {}
  template <typename V, typename = decltype(std::declval<V &>().data())>
  Span(V &v) : data_(v.data()), size_(v.size()) {}
  Span(std::initializer_list<typename std::remove_const<T>::type> l) : data_(l.begin()), size_(l.size()) {}
  T *data() const { return data_; }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  T &operator[](size_t i) const { return data_[i]; }
  T *begin() const { return data_; }
  T *end() const { return data_ + size_; }

 private:
  T *data_ = nullptr;
  size_t size_ = 0;
};

// ---- dense arrays --------------------------------------------------------------
// A dynamically sized column of doubles: stands where the reference uses
// Eigen::ArrayX<double> (solver arrays) and eigenmath::VectorXd (joint vectors).
class ArrayXd {
 public:
  ArrayXd() = default;
  explicit ArrayXd(size_t n, double v = 0.0) : v_(n, v) {}
  ArrayXd(std::initializer_list<double> l) : v_(l) {}
  ArrayXd(const double *p, size_t n) : v_(p, p + n) {}
  void resize(size_t n) { v_.resize(n); }
  void setZero() { for (auto &x : v_) x = 0.0; }
  void setConstant(double c) { for (auto &x : v_) x = c; }
  size_t size() const { return v_.size(); }
  size_t rows() const { return v_.size(); }
  double &operator[](size_t i) { return v_[i]; }
  const double &operator[](size_t i) const { return v_[i]; }
  double &operator()(size_t i) { return v_[i]; }
  const double &operator()(size_t i) const { return v_[i]; }
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }
  std::vector<double>::const_iterator begin() const { return v_.begin(); }
  std::vector<double>::const_iterator end() const { return v_.end(); }
  const double &back() const { return v_.back(); }
  const double &front() const { return v_.front(); }
  double squaredNorm() const { double s = 0; for (double x : v_) s += x * x; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  double dot(const ArrayXd &o) const { double s = 0; for (size_t i = 0; i < v_.size(); i++) s += v_[i] * o.v_[i]; return s; }
  double maxAbs() const { double m = 0; for (double x : v_) m = std::fabs(x) > m ? std::fabs(x) : m; return m; }
  bool operator==(const ArrayXd &o) const { return v_ == o.v_; }

 private:
  std::vector<double> v_;
};
using VectorXd = ArrayXd;

// ---- poses ---------------------------------------------------------------------
// What TimeableCartesianSplinePath needs from eigenmath::Pose3d / Eigen::Quaterniond /
// Eigen::AngleAxisd / eigenmath::Matrix6Xd. eigenmath and Eigen are absent from this image; the
// operations below restate Eigen 3.4's published scalar algorithms (quaternion product and
// rotation of a vector, AngleAxis <-> quaternion), so results agree with an Eigen build to
// rounding, not bit for bit (a vectorised Eigen sums in another order): parity at that level is
// unpinned. They are used only in O(waypoints) host pre-processing.
struct Vector3d {
  double v[3] = {0.0, 0.0, 0.0};
  Vector3d() = default;
  Vector3d(double x, double y, double z) : v{x, y, z} {}
  double &operator[](size_t i) { return v[i]; }
  const double &operator[](size_t i) const { return v[i]; }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
  Vector3d operator+(const Vector3d &o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
  Vector3d operator-(const Vector3d &o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
  Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
  Vector3d cross(const Vector3d &o) const {
    return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
  }
};

struct Quaterniond {
  double w = 1.0, x = 0.0, y = 0.0, z = 0.0;
  Quaterniond() = default;
  Quaterniond(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
  static Quaterniond Identity() { return Quaterniond(); }
  Vector3d vec() const { return {x, y, z}; }
  double squaredNorm() const { return w * w + x * x + y * y + z * z; }
  Quaterniond normalized() const {
    const double n = std::sqrt(squaredNorm());
    return {w / n, x / n, y / n, z / n};
  }
  Quaterniond conjugate() const { return {w, -x, -y, -z}; }
  Quaterniond inverse() const {           // Eigen: conjugate / squaredNorm (zero stays zero)
    const double n2 = squaredNorm();
    if (!(n2 > 0.0)) return {0.0, 0.0, 0.0, 0.0};
    return {w / n2, -x / n2, -y / n2, -z / n2};
  }
  Quaterniond operator*(const Quaterniond &b) const {   // Hamilton product
    return {w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
            w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x};
  }
  Vector3d operator*(const Vector3d &p) const {         // Eigen _transformVector
    const Vector3d u = vec();
    const Vector3d uv = u.cross(p) * 2.0;
    return p + uv * w + u.cross(uv);
  }
};

// Eigen::AngleAxisd(Quaterniond) and back (Eigen/src/Geometry/AngleAxis.h)
struct AngleAxisd {
  double angle = 0.0;
  Vector3d axis{1.0, 0.0, 0.0};
  AngleAxisd() = default;
  explicit AngleAxisd(const Quaterniond &q) {
    double n = q.vec().norm();
    if (n != 0.0) {
      angle = 2.0 * std::atan2(n, std::fabs(q.w));
      if (q.w < 0) n = -n;
      axis = Vector3d(q.x / n, q.y / n, q.z / n);
    }
  }
  Quaterniond toQuaternion() const {
    const double ha = 0.5 * angle;
    const double s = std::sin(ha);
    return {std::cos(ha), s * axis[0], s * axis[1], s * axis[2]};
  }
};

// eigenmath::Pose3d: a unit quaternion and a translation; a * b composes, inverse() inverts.
class Pose3d {
 public:
  Pose3d() = default;
  Pose3d(const Quaterniond &q, const Vector3d &t) : q_(q), t_(t) {}
  const Quaterniond &quaternion() const { return q_; }
  void setQuaternion(const Quaterniond &q) { q_ = q; }
  Vector3d &translation() { return t_; }
  const Vector3d &translation() const { return t_; }
  Pose3d inverse() const {
    const Quaterniond qi = q_.inverse();
    return Pose3d(qi, (qi * t_) * -1.0);
  }
  Pose3d operator*(const Pose3d &b) const { return Pose3d(q_ * b.q_, t_ + q_ * b.t_); }

 private:
  Quaterniond q_;
  Vector3d t_;
};

// eigenmath::Matrix6Xd as the Jacobian callback fills it: 6 x cols, (row, col) access; stored
// row-major, which is what the engine's Cartesian entry points take ([6][D] per sample).
class Matrix6Xd {
 public:
  Matrix6Xd() = default;
  Matrix6Xd(size_t rows, size_t cols) : cols_(cols), v_(6 * cols, 0.0) { (void)rows; }
  void resize(size_t rows, size_t cols) { (void)rows; cols_ = cols; v_.assign(6 * cols, 0.0); }
  void setZero() { for (auto &x : v_) x = 0.0; }
  size_t rows() const { return 6; }
  size_t cols() const { return cols_; }
  double &operator()(size_t r, size_t c) { return v_[r * cols_ + c]; }
  const double &operator()(size_t r, size_t c) const { return v_[r * cols_ + c]; }
  const double *data() const { return v_.data(); }
  double *data() { return v_.data(); }

 private:
  size_t cols_ = 0;
  std::vector<double> v_;
};

// ---- time ------------------------------------------------------------------------
// Nanosecond-resolution time point / duration with the handful of operations
// PathTimingTrajectory uses on absl::Time / absl::Duration.
class Duration {
 public:
  constexpr Duration() = default;
  static constexpr Duration FromNanos(int64_t ns) { return Duration(ns); }
  constexpr int64_t nanos() const { return ns_; }
  constexpr Duration operator+(Duration o) const { return Duration(ns_ + o.ns_); }
  constexpr Duration operator-(Duration o) const { return Duration(ns_ - o.ns_); }
  constexpr double operator/(Duration o) const { return (double)ns_ / (double)o.ns_; }
  constexpr bool operator<(Duration o) const { return ns_ < o.ns_; }
  constexpr bool operator<=(Duration o) const { return ns_ <= o.ns_; }
  constexpr bool operator>(Duration o) const { return ns_ > o.ns_; }
  constexpr bool operator>=(Duration o) const { return ns_ >= o.ns_; }
  constexpr bool operator==(Duration o) const { return ns_ == o.ns_; }

 private:
  constexpr explicit Duration(int64_t ns) : ns_(ns) {}
  int64_t ns_ = 0;
};
inline Duration Nanoseconds(int64_t n) { return Duration::FromNanos(n); }
inline Duration Milliseconds(double ms) { return Duration::FromNanos((int64_t)std::llround(ms * 1e6)); }
inline Duration Seconds(double s) { return Duration::FromNanos((int64_t)std::llround(s * 1e9)); }

class Time {
 public:
  constexpr Time() = default;
  static constexpr Time FromUnixNanosRaw(int64_t ns) { return Time(ns); }
  constexpr int64_t unix_nanos() const { return ns_; }
  constexpr Time operator+(Duration d) const { return Time(ns_ + d.nanos()); }
  constexpr Time operator-(Duration d) const { return Time(ns_ - d.nanos()); }
  constexpr Duration operator-(Time o) const { return Duration::FromNanos(ns_ - o.ns_); }
  constexpr bool operator<(Time o) const { return ns_ < o.ns_; }
  constexpr bool operator<=(Time o) const { return ns_ <= o.ns_; }
  constexpr bool operator>(Time o) const { return ns_ > o.ns_; }
  constexpr bool operator>=(Time o) const { return ns_ >= o.ns_; }
  constexpr bool operator==(Time o) const { return ns_ == o.ns_; }

 private:
  constexpr explicit Time(int64_t ns) : ns_(ns) {}
  int64_t ns_ = 0;
};
inline Time FromUnixNanos(int64_t ns) { return Time::FromUnixNanosRaw(ns); }
inline Time FromUnixSeconds(double s) { return Time::FromUnixNanosRaw((int64_t)(s * 1e9)); }
inline int64_t ToUnixNanos(Time t) { return t.unix_nanos(); }

}  // namespace compat
}  // namespace tpamd
