// penta_diagonal_matrix.h — host container for the symmetric block penta-diagonal Hessian,
// mirroring the parts of reference optimizer/penta_diagonal_matrix.h:36-189 the optimizer's
// public API exposes (block_rows, block_size, MultiplyBy, ExtractDiagonal, ScaleByDiagonal,
// MakeDense).  Only the lower bands A (two below), B (one below) and C (diagonal, both
// triangles) are stored; D_i = B_{i+1}^T and E_i = A_{i+2}^T are implied (MakeSymmetric,
// penta_diagonal_matrix.cc:64-105).  Blocks are column-major and contiguous: exactly the
// arrays IDTO_ARR_H_A/B/C of the C-ABI, so filling one is three copies.
#pragma once

#include <vector>

#include "idto/optimizer/types.h"

namespace idto {
namespace optimizer {

template <typename T>
class PentaDiagonalMatrix {
 public:
  PentaDiagonalMatrix() = default;
  PentaDiagonalMatrix(int block_rows, int block_size)
      : n_(block_rows), k_(block_size), A_((std::size_t)n_ * k_ * k_, 0), B_(A_), C_(A_) {}
  int block_rows() const { return n_; }
  int block_size() const { return k_; }
  int rows() const { return n_ * k_; }
  std::vector<T>& mutable_A() { return A_; }
  std::vector<T>& mutable_B() { return B_; }
  std::vector<T>& mutable_C() { return C_; }
  const std::vector<T>& A() const { return A_; }
  const std::vector<T>& B() const { return B_; }
  const std::vector<T>& C() const { return C_; }
  const T* block(const std::vector<T>& band, int i) const { return band.data() + (std::size_t)i * k_ * k_; }

  // y = H x   (penta_diagonal_matrix.cc:181-207)
  void MultiplyBy(const std::vector<T>& x, std::vector<T>* y) const {
    y->assign((std::size_t)rows(), T(0));
    for (int i = 0; i < n_; ++i) {
      T* yi = y->data() + (std::size_t)i * k_;
      auto add = [&](const T* M, bool transpose, const T* xj) {
        for (int c = 0; c < k_; ++c)
          for (int r = 0; r < k_; ++r) yi[r] += (transpose ? M[(std::size_t)r * k_ + c] : M[(std::size_t)c * k_ + r]) * xj[c];
      };
      if (i >= 2) add(block(A_, i), false, x.data() + (std::size_t)(i - 2) * k_);
      if (i >= 1) add(block(B_, i), false, x.data() + (std::size_t)(i - 1) * k_);
      add(block(C_, i), false, x.data() + (std::size_t)i * k_);
      if (i + 1 < n_) add(block(B_, i + 1), true, x.data() + (std::size_t)(i + 1) * k_);
      if (i + 2 < n_) add(block(A_, i + 2), true, x.data() + (std::size_t)(i + 2) * k_);
    }
  }
  // diag(H)   (penta_diagonal_matrix.cc:210-218)
  void ExtractDiagonal(std::vector<T>* d) const {
    d->resize((std::size_t)rows());
    for (int i = 0; i < n_; ++i)
      for (int r = 0; r < k_; ++r) (*d)[(std::size_t)i * k_ + r] = block(C_, i)[(std::size_t)r * k_ + r];
  }
  // H <- diag(s) H diag(s)   (penta_diagonal_matrix.cc:221-257)
  void ScaleByDiagonal(const std::vector<T>& s) {
    auto scale = [&](std::vector<T>& band, int shift) {
      for (int i = shift; i < n_; ++i)
        for (int c = 0; c < k_; ++c)
          for (int r = 0; r < k_; ++r)
            band[((std::size_t)i * k_ + c) * k_ + r] *= s[(std::size_t)i * k_ + r] * s[(std::size_t)(i - shift) * k_ + c];
    };
    scale(A_, 2); scale(B_, 1); scale(C_, 0);
  }
  MatrixXd MakeDense() const {
    MatrixXd M(rows(), rows());
    for (int i = 0; i < n_; ++i)
      for (int c = 0; c < k_; ++c)
        for (int r = 0; r < k_; ++r) {
          M(i * k_ + r, i * k_ + c) = block(C_, i)[(std::size_t)c * k_ + r];
          if (i >= 1) M(i * k_ + r, (i - 1) * k_ + c) = M((i - 1) * k_ + c, i * k_ + r) = block(B_, i)[(std::size_t)c * k_ + r];
          if (i >= 2) M(i * k_ + r, (i - 2) * k_ + c) = M((i - 2) * k_ + c, i * k_ + r) = block(A_, i)[(std::size_t)c * k_ + r];
        }
    return M;
  }

 private:
  int n_ = 0, k_ = 0;
  std::vector<T> A_, B_, C_;
};

}  // namespace optimizer
}  // namespace idto
