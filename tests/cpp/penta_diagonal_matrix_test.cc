// The reference's PentaDiagonalMatrix tests (optimizer/test/penta_diagonal_solver_test.cc:42-107:
// SymmetricMatrixEmpty, MutateMatrix, SymmetricMatrix) and the products its optimizer relies on
// (MultiplyBy :109-149 of the same file's style, ExtractDiagonal / ScaleByDiagonal against MakeDense),
// against include/idto/optimizer/penta_diagonal_matrix.h.  Plain asserts; exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "idto/optimizer/penta_diagonal_matrix.h"

using idto::optimizer::MatrixXd;
using idto::optimizer::PentaDiagonalMatrix;

#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

template <class F>
static bool Throws(F f) {
  try { f(); } catch (const std::exception&) { return true; }
  return false;
}

static std::vector<double> Band(int n, int k, std::initializer_list<double> per_block) {
  std::vector<double> b;
  for (double v : per_block) b.insert(b.end(), (std::size_t)k * k, v);
  (void)n;
  return b;
}

int main() {
  {  // SymmetricMatrixEmpty
    PentaDiagonalMatrix<double> M(0, 3, {}, {}, {});
    CHECK(M.rows() == 0);
  }
  {  // MutateMatrix
    const int k = 3;
    PentaDiagonalMatrix<double> M(5, k);
    CHECK(M.is_symmetric());
    CHECK(M.block_rows() == 5 && M.block_cols() == 5 && M.block_size() == 3 && M.rows() == 15 && M.cols() == 15);
    const std::vector<double> some = Band(5, k, {1.5, 2.1, -12.8, -12.8, 15.3});
    // a symmetric matrix only lets its lower bands be mutated
    CHECK(Throws([&] { M.mutable_D(); }));
    CHECK(Throws([&] { M.mutable_E(); }));
    CHECK(M.A() != some);
    M.mutable_A() = some;
    CHECK(M.A() == some);
    CHECK(M.B() != some);
    M.mutable_B() = some;
    CHECK(M.B() == some);
    CHECK(M.C() != some);
    M.mutable_C() = some;
    CHECK(M.C() == some);
    // some terms changed: symmetry can no longer be assumed
    CHECK(!M.is_symmetric());
    CHECK(!Throws([&] { M.mutable_D(); }));
    std::vector<double> d;
    CHECK(Throws([&] { M.ExtractDiagonal(&d); }));
    CHECK(Throws([&] { M.ScaleByDiagonal(std::vector<double>(15, 1.0)); }));
    M.MakeSymmetric();
    CHECK(M.is_symmetric());
    CHECK(!Throws([&] { M.ExtractDiagonal(&d); }));
  }
  {  // SymmetricMatrix: zero padding and the mirrored bands
    const int k = 5;
    PentaDiagonalMatrix<double> M(3, k, Band(3, k, {0, 0, 1.5}), Band(3, k, {0, 2.1, -12.8}), Band(3, k, {1.8, 15.3, 7.1}));
    CHECK(M.rows() == k * 3 && M.block_rows() == 3 && M.is_symmetric());
    auto blk = [&](const std::vector<double>& b, int i) { return std::vector<double>(b.begin() + (std::size_t)i * k * k, b.begin() + (std::size_t)(i + 1) * k * k); };
    const std::vector<double> Z((std::size_t)k * k, 0.0);
    CHECK(blk(M.D(), 0) == blk(M.B(), 1));   // (constant blocks: equal to their transposes)
    CHECK(blk(M.D(), 1) == blk(M.B(), 2));
    CHECK(blk(M.D(), 2) == Z);
    CHECK(blk(M.E(), 0) == blk(M.A(), 2));
    CHECK(blk(M.E(), 1) == Z && blk(M.E(), 2) == Z);
  }
  {  // products, diagonal and scaling against the dense matrix; a non-symmetric matrix multiplies too
    const int n = 6, k = 3, N = n * k;
    MatrixXd Dn(N, N);
    unsigned long long s = 12345;
    auto rnd = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (int c = 0; c < N; ++c)
      for (int r = c; r < N; ++r) {
        const double v = (std::abs(r / k - c / k) <= 2) ? rnd() : 0.0;
        Dn(r, c) = v; Dn(c, r) = v;
      }
    PentaDiagonalMatrix<double> M = PentaDiagonalMatrix<double>::MakeSymmetricFromLowerDense(Dn, n, k);
    CHECK(M.is_symmetric());
    const MatrixXd back = M.MakeDense();
    for (int c = 0; c < N; ++c)
      for (int r = 0; r < N; ++r) CHECK(back(r, c) == Dn(r, c));
    std::vector<double> x(N), y, d, sc(N);
    for (int i = 0; i < N; ++i) { x[i] = rnd(); sc[i] = 0.5 + i * 0.1; }
    M.MultiplyBy(x, &y);
    for (int r = 0; r < N; ++r) {
      double acc = 0.0;
      for (int c = 0; c < N; ++c) acc += Dn(r, c) * x[c];
      CHECK(std::fabs(acc - y[r]) <= 1e-14);
    }
    M.ExtractDiagonal(&d);
    for (int i = 0; i < N; ++i) CHECK(d[i] == Dn(i, i));
    M.ScaleByDiagonal(sc);
    const MatrixXd scaled = M.MakeDense();
    for (int c = 0; c < N; ++c)
      for (int r = 0; r < N; ++r) CHECK(std::fabs(scaled(r, c) - sc[r] * Dn(r, c) * sc[c]) <= 1e-15);
    CHECK(M.is_symmetric());
    // general (non-symmetric) matrix from five bands
    std::vector<double> A((std::size_t)n * k * k), B(A), C(A), D(A), E(A);
    for (auto* b : {&A, &B, &C, &D, &E}) for (double& v : *b) v = rnd();
    PentaDiagonalMatrix<double> G(n, k, A, B, C, D, E);
    CHECK(!G.is_symmetric());
    const MatrixXd Gd = G.MakeDense();
    G.MultiplyBy(x, &y);
    for (int r = 0; r < N; ++r) {
      double acc = 0.0;
      for (int c = 0; c < N; ++c) acc += Gd(r, c) * x[c];
      CHECK(std::fabs(acc - y[r]) <= 1e-13);
    }
    const PentaDiagonalMatrix<double> I = PentaDiagonalMatrix<double>::MakeIdentity(n, k);
    I.MultiplyBy(x, &y);
    CHECK(y == x && I.is_symmetric());
  }
  std::printf("penta_diagonal_matrix_test: all passed\n");
  return 0;
}
