// types.h — the two value types of the host API.  The reference uses Eigen
// (drake::VectorX<T>, drake::MatrixX<T>); Eigen is not part of this build, and the
// C-ABI below the host layer takes plain column-major arrays, so these are thin
// std::vector wrappers with the same element order as Eigen's defaults.
#pragma once

#include <cstddef>
#include <initializer_list>
#include <vector>

namespace idto {
namespace optimizer {

using VectorXd = std::vector<double>;

// Dense column-major matrix (Eigen::MatrixXd storage order).
class MatrixXd {
 public:
  MatrixXd() = default;
  MatrixXd(int rows, int cols) : rows_(rows), cols_(cols), data_((std::size_t)rows * cols, 0.0) {}
  static MatrixXd Identity(int n) {
    MatrixXd m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1.0;
    return m;
  }
  // diag(d): what `d.asDiagonal()` gives in the reference's examples (examples/example_base.cc:387-391)
  static MatrixXd Diagonal(const VectorXd& d) {
    MatrixXd m((int)d.size(), (int)d.size());
    for (std::size_t i = 0; i < d.size(); ++i) m((int)i, (int)i) = d[i];
    return m;
  }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  double& operator()(int r, int c) { return data_[(std::size_t)c * rows_ + r]; }
  double operator()(int r, int c) const { return data_[(std::size_t)c * rows_ + r]; }
  const double* data() const { return data_.data(); }
  double* data() { return data_.data(); }

 private:
  int rows_ = 0, cols_ = 0;
  std::vector<double> data_;
};

}  // namespace optimizer
}  // namespace idto
