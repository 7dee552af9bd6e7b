// Tiny fixed-size vectors with the handful of Eigen accessors the BA boundary
// uses (x(), y(), operator[], data(), head<3>() is not needed).  The reference's
// data model stores Eigen::Vector2d / Vector4d (feature.h, track.h:84-88); this
// Eigen-free shim keeps the same spelling so host code ports verbatim.
#ifndef THEIA_MI355_EIGEN_LITE_H_
#define THEIA_MI355_EIGEN_LITE_H_
#include <cstddef>
#include <cstdint>
namespace Eigen {
template <typename T, int N>
struct LiteVector {
  T v[N];
  LiteVector() { for (int i = 0; i < N; ++i) v[i] = T(0); }
  LiteVector(T a, T b) { static_assert(N == 2, "size"); v[0] = a; v[1] = b; }
  LiteVector(T a, T b, T c) { static_assert(N == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
  LiteVector(T a, T b, T c, T d) { static_assert(N == 4, "size"); v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T& x() { return v[0]; }
  T& y() { return v[1]; }
  const T& x() const { return v[0]; }
  const T& y() const { return v[1]; }
  T* data() { return v; }
  const T* data() const { return v; }
  static constexpr int size() { return N; }
  static LiteVector Zero() { return LiteVector(); }
};
typedef LiteVector<double, 2> Vector2d;
typedef LiteVector<double, 3> Vector3d;
typedef LiteVector<double, 4> Vector4d;
}  // namespace Eigen
#endif
