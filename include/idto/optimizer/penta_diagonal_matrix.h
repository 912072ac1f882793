// penta_diagonal_matrix.h — host container for a block penta-diagonal matrix, mirroring reference
// optimizer/penta_diagonal_matrix.h:36-189 (constructors, the five bands and their accessors, the
// `is_symmetric` flag and its rules, MakeSymmetric, MakeIdentity, MakeSymmetricFromLowerDense,
// MakeDense, MultiplyBy, ExtractDiagonal, ScaleByDiagonal).
//
// Storage differs from the reference (std::vector<Eigen::MatrixXd> per band): every band is ONE flat
// array of num_blocks column-major k x k blocks - exactly the arrays IDTO_ARR_H_A/B/C of the C-ABI, so
// that filling a Hessian from the device is three copies.  Block row i is [A_i B_i C_i D_i E_i] with
// A_0, B_0, A_1, E_{n-2}, D_{n-1}, E_{n-1} stored as zeros (reference :30-34).
//
// The symmetric flag follows the reference to the letter (:131-176, penta_diagonal_matrix.cc:64-105,
// :211, :222): a matrix built by the (num_blocks, block_size) or the three-band constructor is
// symmetric; taking a mutable lower band (mutable_A/B/C) clears the flag, MakeSymmetric() mirrors the
// lower bands into D, E and the lower triangle of C into its upper one and sets it again; mutable_D/E
// throw on a symmetric matrix; ExtractDiagonal and ScaleByDiagonal demand a symmetric matrix.
#pragma once

#include <stdexcept>
#include <utility>
#include <vector>

#include "idto/optimizer/types.h"

namespace idto {
namespace optimizer {

template <typename T>
class PentaDiagonalMatrix {
 public:
  PentaDiagonalMatrix() = default;

  // num_blocks x num_blocks blocks of block_size x block_size, all zero (reference :50-53)
  PentaDiagonalMatrix(int num_blocks, int block_size, bool is_symmetric = true)
      : n_(num_blocks), k_(block_size), symmetric_(is_symmetric), A_(band_size(), T(0)), B_(A_), C_(A_), D_(A_), E_(A_) {}

  // general matrix from its five bands (reference :67-69); every band holds num_blocks blocks
  PentaDiagonalMatrix(int num_blocks, int block_size, std::vector<T> A, std::vector<T> B, std::vector<T> C,
                      std::vector<T> D, std::vector<T> E)
      : n_(num_blocks), k_(block_size), symmetric_(false), A_(std::move(A)), B_(std::move(B)), C_(std::move(C)),
        D_(std::move(D)), E_(std::move(E)) {
    check_sizes();
  }

  // symmetric matrix from its lower bands: E_i = A_{i+2}^T, D_i = B_{i+1}^T, only the lower triangle of
  // C_i is used (reference :84-85)
  PentaDiagonalMatrix(int num_blocks, int block_size, std::vector<T> A, std::vector<T> B, std::vector<T> C)
      : n_(num_blocks), k_(block_size), symmetric_(false), A_(std::move(A)), B_(std::move(B)), C_(std::move(C)) {
    D_.assign(band_size(), T(0));
    E_.assign(band_size(), T(0));
    check_sizes();
    MakeSymmetric();
  }

  static PentaDiagonalMatrix<T> MakeIdentity(int num_blocks, int block_size) {   // reference :97
    PentaDiagonalMatrix<T> M(num_blocks, block_size);
    for (int i = 0; i < num_blocks; ++i)
      for (int r = 0; r < block_size; ++r) M.C_[M.at(i, r, r)] = T(1);
    return M;
  }

  // the block penta-diagonal part of the lower triangle of a dense matrix, mirrored (reference :99-101)
  static PentaDiagonalMatrix<T> MakeSymmetricFromLowerDense(const MatrixXd& M, int num_blocks, int block_size) {
    if (M.rows() != num_blocks * block_size || M.cols() != M.rows())
      throw std::invalid_argument("MakeSymmetricFromLowerDense: size mismatch");
    PentaDiagonalMatrix<T> P(num_blocks, block_size);
    const int k = block_size;
    for (int i = 0; i < num_blocks; ++i)
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) {
          P.C_[P.at(i, r, c)] = M(i * k + r, i * k + c);
          if (i >= 1) P.B_[P.at(i, r, c)] = M(i * k + r, (i - 1) * k + c);
          if (i >= 2) P.A_[P.at(i, r, c)] = M(i * k + r, (i - 2) * k + c);
        }
    P.MakeSymmetric();
    return P;
  }

  // E_i = A_{i+2}^T, D_i = B_{i+1}^T, upper triangle of C_i from its lower one (penta_diagonal_matrix.cc:64-105)
  void MakeSymmetric() {
    check_sizes();
    for (int i = 0; i < n_; ++i)
      for (int c = 0; c < k_; ++c)
        for (int r = 0; r < k_; ++r) {
          D_[at(i, r, c)] = (i + 1 < n_) ? B_[at(i + 1, c, r)] : T(0);
          E_[at(i, r, c)] = (i + 2 < n_) ? A_[at(i + 2, c, r)] : T(0);
          if (r < c) C_[at(i, r, c)] = C_[at(i, c, r)];
        }
    symmetric_ = true;
  }

  int block_rows() const { return n_; }
  int block_cols() const { return n_; }
  int block_size() const { return k_; }
  int rows() const { return n_ * k_; }
  int cols() const { return n_ * k_; }
  bool is_symmetric() const { return symmetric_; }

  // (reference :131-176) a caller who takes a mutable lower band may break the symmetry: the flag is cleared
  // until MakeSymmetric(); the upper bands of a symmetric matrix cannot be taken at all
  std::vector<T>& mutable_A() { symmetric_ = false; return A_; }
  std::vector<T>& mutable_B() { symmetric_ = false; return B_; }
  std::vector<T>& mutable_C() { symmetric_ = false; return C_; }
  std::vector<T>& mutable_D() {
    if (symmetric_) throw std::logic_error("PentaDiagonalMatrix::mutable_D: the matrix is symmetric (mutate B and MakeSymmetric)");
    return D_;
  }
  std::vector<T>& mutable_E() {
    if (symmetric_) throw std::logic_error("PentaDiagonalMatrix::mutable_E: the matrix is symmetric (mutate A and MakeSymmetric)");
    return E_;
  }
  const std::vector<T>& A() const { return A_; }
  const std::vector<T>& B() const { return B_; }
  const std::vector<T>& C() const { return C_; }
  const std::vector<T>& D() const { return D_; }
  const std::vector<T>& E() const { return E_; }
  const T* block(const std::vector<T>& band, int i) const { return band.data() + (std::size_t)i * k_ * k_; }

  // y = M x   (penta_diagonal_matrix.cc:181-207; any matrix, symmetric or not)
  void MultiplyBy(const std::vector<T>& x, std::vector<T>* y) const {
    if ((int)x.size() != cols()) throw std::invalid_argument("PentaDiagonalMatrix::MultiplyBy: size mismatch");
    y->assign((std::size_t)rows(), T(0));
    for (int i = 0; i < n_; ++i) {
      T* yi = y->data() + (std::size_t)i * k_;
      auto add = [&](const T* M, const T* xj) {
        for (int c = 0; c < k_; ++c)
          for (int r = 0; r < k_; ++r) yi[r] += M[(std::size_t)c * k_ + r] * xj[c];
      };
      if (i >= 2) add(block(A_, i), x.data() + (std::size_t)(i - 2) * k_);
      if (i >= 1) add(block(B_, i), x.data() + (std::size_t)(i - 1) * k_);
      add(block(C_, i), x.data() + (std::size_t)i * k_);
      if (i + 1 < n_) add(block(D_, i), x.data() + (std::size_t)(i + 1) * k_);
      if (i + 2 < n_) add(block(E_, i), x.data() + (std::size_t)(i + 2) * k_);
    }
  }
  // diag(M)   (penta_diagonal_matrix.cc:210-218)
  void ExtractDiagonal(std::vector<T>* d) const {
    if (!symmetric_) throw std::logic_error("PentaDiagonalMatrix::ExtractDiagonal: the matrix is not symmetric");
    d->resize((std::size_t)rows());
    for (int i = 0; i < n_; ++i)
      for (int r = 0; r < k_; ++r) (*d)[(std::size_t)i * k_ + r] = C_[at(i, r, r)];
  }
  // M <- diag(s) M diag(s)   (penta_diagonal_matrix.cc:221-257)
  void ScaleByDiagonal(const std::vector<T>& s) {
    if (!symmetric_) throw std::logic_error("PentaDiagonalMatrix::ScaleByDiagonal: the matrix is not symmetric");
    if ((int)s.size() != rows()) throw std::invalid_argument("PentaDiagonalMatrix::ScaleByDiagonal: size mismatch");
    auto scale = [&](std::vector<T>& band, int shift) {   // block (i, i - shift)
      for (int i = (shift > 0 ? shift : 0); i < n_ + (shift < 0 ? shift : 0); ++i)
        for (int c = 0; c < k_; ++c)
          for (int r = 0; r < k_; ++r) band[at(i, r, c)] *= s[(std::size_t)i * k_ + r] * s[(std::size_t)(i - shift) * k_ + c];
    };
    scale(A_, 2); scale(B_, 1); scale(C_, 0); scale(D_, -1); scale(E_, -2);
  }
  MatrixXd MakeDense() const {
    MatrixXd M(rows(), cols());
    for (int i = 0; i < n_; ++i)
      for (int c = 0; c < k_; ++c)
        for (int r = 0; r < k_; ++r) {
          M(i * k_ + r, i * k_ + c) = C_[at(i, r, c)];
          if (i >= 1) M(i * k_ + r, (i - 1) * k_ + c) = B_[at(i, r, c)];
          if (i >= 2) M(i * k_ + r, (i - 2) * k_ + c) = A_[at(i, r, c)];
          if (i + 1 < n_) M(i * k_ + r, (i + 1) * k_ + c) = D_[at(i, r, c)];
          if (i + 2 < n_) M(i * k_ + r, (i + 2) * k_ + c) = E_[at(i, r, c)];
        }
    return M;
  }

 private:
  std::size_t band_size() const { return (std::size_t)n_ * k_ * k_; }
  std::size_t at(int i, int r, int c) const { return ((std::size_t)i * k_ + c) * k_ + r; }
  void check_sizes() const {
    if (A_.size() != band_size() || B_.size() != band_size() || C_.size() != band_size() || D_.size() != band_size() ||
        E_.size() != band_size())
      throw std::invalid_argument("PentaDiagonalMatrix: every band must hold num_blocks blocks of block_size x block_size");
  }
  int n_ = 0, k_ = 0;
  bool symmetric_ = false;
  std::vector<T> A_, B_, C_, D_, E_;
};

}  // namespace optimizer
}  // namespace idto
