// oracle/penta.h — TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Block penta-diagonal matrix and the block-Thomas factorisation, restating
//   reference optimizer/penta_diagonal_matrix.cc:64-105 (MakeSymmetric),
//   :149-169 (MakeDense), :181-207 (MultiplyBy), :210-218 (ExtractDiagonal),
//   :221-257 (ScaleByDiagonal);
//   reference optimizer/penta_diagonal_solver.h:124-197 (Factorize),
//   :199-248 (SolveInPlace).
// The per-block solver is LU with partial pivoting, the restatement of
// Eigen::PartialPivLU (penta_diagonal_solver.h:40; Eigen itself is not in
// /root/reference — unblocked right-looking elimination, first-maximum pivot).
// Blocks are column-major bs x bs, bands are block-major.
#pragma once

#include <cmath>
#include <cstring>
#include <vector>

namespace oracle {

struct PentaMatrix {
  int n = 0, bs = 0;  // block rows, block size
  std::vector<double> A, B, C, D, E;
  PentaMatrix() {}
  PentaMatrix(int n_, int bs_) { Resize(n_, bs_); }
  void Resize(int n_, int bs_) {
    n = n_; bs = bs_;
    const size_t sz = (size_t)n * bs * bs;
    A.assign(sz, 0.0); B.assign(sz, 0.0); C.assign(sz, 0.0); D.assign(sz, 0.0); E.assign(sz, 0.0);
  }
  int size() const { return n * bs; }
  double* blk(std::vector<double>& X, int i) { return X.data() + (size_t)i * bs * bs; }
  const double* blk(const std::vector<double>& X, int i) const { return X.data() + (size_t)i * bs * bs; }

  // penta_diagonal_matrix.cc:64-105
  void MakeSymmetric() {
    const int k = bs;
    for (int i = 0; i < n; ++i) {
      double* Ci = blk(C, i);
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < c; ++r) Ci[(size_t)c * k + r] = Ci[(size_t)r * k + c];  // upper <- lower^T
    }
    auto transpose_into = [&](const double* src, double* dst) {
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) dst[(size_t)c * k + r] = src[(size_t)r * k + c];
    };
    if (n >= 2) {
      for (int i = 0; i < n - 1; ++i) transpose_into(blk(B, i + 1), blk(D, i));
      std::memset(blk(D, n - 1), 0, sizeof(double) * k * k);
    }
    if (n >= 3) {
      for (int i = 0; i < n - 2; ++i) transpose_into(blk(A, i + 2), blk(E, i));
      std::memset(blk(E, n - 1), 0, sizeof(double) * k * k);
      std::memset(blk(E, n - 2), 0, sizeof(double) * k * k);
    }
  }

  // penta_diagonal_matrix.cc:149-169 ; dense is column-major size x size
  std::vector<double> MakeDense() const {
    const int k = bs, N = size();
    std::vector<double> M((size_t)N * N, 0.0);
    auto put = [&](const double* b, int bi, int bj) {
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) M[(size_t)(bj * k + c) * N + bi * k + r] = b[(size_t)c * k + r];
    };
    for (int i = 0; i < n; ++i) {
      if (i >= 2) put(blk(A, i), i, i - 2);
      if (i >= 1) put(blk(B, i), i, i - 1);
      put(blk(C, i), i, i);
      if (i < n - 1) put(blk(D, i), i, i + 1);
      if (i < n - 2) put(blk(E, i), i, i + 2);
    }
    return M;
  }

  static void GemvAcc(const double* Mb, const double* x, double* y, int k, bool init) {
    // y (+)= Mb * x with the accumulation order acc = sum_c Mb[r][c]*x[c], c ascending
    for (int r = 0; r < k; ++r) {
      double acc = Mb[r] * x[0];
      for (int c = 1; c < k; ++c) acc += Mb[(size_t)c * k + r] * x[c];
      y[r] = init ? acc : y[r] + acc;
    }
  }

  // penta_diagonal_matrix.cc:181-207
  void MultiplyBy(const double* v, double* result) const {
    const int k = bs;
    for (int i = 0; i < n; ++i) {
      double* res = result + (size_t)i * k;
      GemvAcc(blk(C, i), v + (size_t)i * k, res, k, true);
      if (i >= 1) GemvAcc(blk(B, i), v + (size_t)(i - 1) * k, res, k, false);
      if (i >= 2) GemvAcc(blk(A, i), v + (size_t)(i - 2) * k, res, k, false);
      if (i < n - 1) GemvAcc(blk(D, i), v + (size_t)(i + 1) * k, res, k, false);
      if (i < n - 2) GemvAcc(blk(E, i), v + (size_t)(i + 2) * k, res, k, false);
    }
  }

  // penta_diagonal_matrix.cc:210-218
  void ExtractDiagonal(double* d) const {
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < bs; ++r) d[(size_t)i * bs + r] = blk(C, i)[(size_t)r * bs + r];
  }

  // penta_diagonal_matrix.cc:221-257
  void ScaleByDiagonal(const double* s) {
    const int k = bs;
    auto scale = [&](double* b, const double* sl, const double* sr) {
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) b[(size_t)c * k + r] = (sl[r] * b[(size_t)c * k + r]) * sr[c];
    };
    for (int i = 0; i < n; ++i) {
      scale(blk(C, i), s + (size_t)i * k, s + (size_t)i * k);
      if (i >= 1) scale(blk(B, i), s + (size_t)i * k, s + (size_t)(i - 1) * k);
      if (i >= 2) scale(blk(A, i), s + (size_t)i * k, s + (size_t)(i - 2) * k);
    }
    auto transpose_into = [&](const double* src, double* dst) {
      for (int c = 0; c < k; ++c)
        for (int r = 0; r < k; ++r) dst[(size_t)c * k + r] = src[(size_t)r * k + c];
    };
    if (n >= 2)
      for (int i = 0; i < n - 1; ++i) transpose_into(blk(B, i + 1), blk(D, i));
    if (n >= 3)
      for (int i = 0; i < n - 2; ++i) transpose_into(blk(A, i + 2), blk(E, i));
  }
};

// LU with partial pivoting of a k x k column-major block (in place), the
// restatement of Eigen::PartialPivLU.
struct BlockLU {
  int k = 0;
  std::vector<double> lu;
  std::vector<int> piv;  // row swapped with row i at step i
  void Compute(const double* G, int k_) {
    k = k_;
    lu.assign(G, G + (size_t)k * k);
    piv.assign(k, 0);
    for (int j = 0; j < k; ++j) {
      int p = j;
      double best = std::fabs(lu[(size_t)j * k + j]);
      for (int r = j + 1; r < k; ++r) {
        const double a = std::fabs(lu[(size_t)j * k + r]);
        if (a > best) { best = a; p = r; }
      }
      piv[j] = p;
      if (p != j)
        for (int c = 0; c < k; ++c) std::swap(lu[(size_t)c * k + j], lu[(size_t)c * k + p]);
      const double d = lu[(size_t)j * k + j];
      for (int r = j + 1; r < k; ++r) lu[(size_t)j * k + r] = lu[(size_t)j * k + r] / d;
      for (int c = j + 1; c < k; ++c) {
        const double u = lu[(size_t)c * k + j];
        for (int r = j + 1; r < k; ++r) lu[(size_t)c * k + r] = lu[(size_t)c * k + r] - lu[(size_t)j * k + r] * u;
      }
    }
  }
  // X <- G^{-1} X for nrhs column-major columns of length k
  void Solve(double* X, int nrhs) const {
    for (int c = 0; c < nrhs; ++c) {
      double* x = X + (size_t)c * k;
      for (int j = 0; j < k; ++j)
        if (piv[j] != j) std::swap(x[j], x[piv[j]]);
      for (int j = 0; j < k; ++j) {      // L y = P x, unit lower
        const double xj = x[j];
        for (int r = j + 1; r < k; ++r) x[r] = x[r] - lu[(size_t)j * k + r] * xj;
      }
      for (int j = k - 1; j >= 0; --j) {  // U x = y
        x[j] = x[j] / lu[(size_t)j * k + j];
        const double xj = x[j];
        for (int r = 0; r < j; ++r) x[r] = x[r] - lu[(size_t)j * k + r] * xj;
      }
    }
  }
};

// out = X - L * R  (all k x k column-major); acc = sum_j L[r][j] R[j][c], j ascending
inline void BlockSubMul(const double* X, const double* L, const double* R, double* out, int k) {
  for (int c = 0; c < k; ++c)
    for (int r = 0; r < k; ++r) {
      double acc = L[r] * R[(size_t)c * k];
      for (int j = 1; j < k; ++j) acc += L[(size_t)j * k + r] * R[(size_t)c * k + j];
      out[(size_t)c * k + r] = X[(size_t)c * k + r] - acc;
    }
}

// penta_diagonal_solver.h:124-248
struct PentaFactorization {
  int n = 0, k = 0;
  std::vector<double> A, K, Y, Z;  // Y, Z stored with the +2 offset of the reference
  std::vector<BlockLU> Ginv;
  bool ok = false;

  explicit PentaFactorization(const PentaMatrix& M) { Factorize(M); }

  void Factorize(const PentaMatrix& M) {
    n = M.n; k = M.bs;
    const size_t kk = (size_t)k * k;
    A = M.A;
    K.assign((size_t)n * kk, 0.0);
    Y.assign((size_t)(n + 2) * kk, 0.0);
    Z.assign((size_t)(n + 2) * kk, 0.0);
    Ginv.resize(n);
    std::vector<double> G(kk), T(kk);
    for (int i = 0; i < n; ++i) {
      const double* Ai = M.blk(M.A, i);
      const double* Bi = M.blk(M.B, i);
      const double* Ci = M.blk(M.C, i);
      const double* Yim2 = &Y[(size_t)i * kk];
      const double* Zim2 = &Z[(size_t)i * kk];
      const double* Yim1 = &Y[(size_t)(i + 1) * kk];
      const double* Zim1 = &Z[(size_t)(i + 1) * kk];
      double* Ki = &K[(size_t)i * kk];
      double* Yi = &Y[(size_t)(i + 2) * kk];
      double* Zi = &Z[(size_t)(i + 2) * kk];
      BlockSubMul(Bi, Ai, Yim2, Ki, k);            // K = B - A Y_{i-2}   (:167)
      BlockSubMul(Ci, Ai, Zim2, T.data(), k);      // G = C - A Z_{i-2}   (:168)
      BlockSubMul(T.data(), Ki, Yim1, G.data(), k);  // G -= K Y_{i-1}    (:174)
      BlockSubMul(M.blk(M.D, i), Ki, Zim1, Yi, k);   // Y = D - K Z_{i-1} (:178)
      Ginv[i].Compute(G.data(), k);                // (:182)
      Ginv[i].Solve(Yi, k);                        // (:189)
      std::memcpy(Zi, M.blk(M.E, i), sizeof(double) * kk);
      Ginv[i].Solve(Zi, k);                        // (:193-194)
    }
    ok = true;
  }

  // :199-248 ; b has n*k entries, solved in place
  void SolveInPlace(double* b) const {
    const size_t kk = (size_t)k * k;
    std::vector<double> r((size_t)(n + 2) * k, 0.0);
    std::memcpy(r.data() + 2 * k, b, sizeof(double) * n * k);
    std::vector<double> t(k);
    for (int i = 0; i < n; ++i) {
      const double* rim2 = &r[(size_t)i * k];
      const double* rim1 = &r[(size_t)(i + 1) * k];
      double* ri = &r[(size_t)(i + 2) * k];
      PentaMatrix::GemvAcc(&A[(size_t)i * kk], rim2, t.data(), k, true);
      for (int j = 0; j < k; ++j) ri[j] = ri[j] - t[j];
      PentaMatrix::GemvAcc(&K[(size_t)i * kk], rim1, t.data(), k, true);
      for (int j = 0; j < k; ++j) ri[j] = ri[j] - t[j];
      Ginv[i].Solve(ri, 1);
    }
    std::memcpy(b, r.data() + 2 * k, sizeof(double) * n * k);
    if (n >= 2) {
      int i = n - 2;
      PentaMatrix::GemvAcc(&Y[(size_t)(i + 2) * kk], b + (size_t)(n - 1) * k, t.data(), k, true);
      for (int j = 0; j < k; ++j) b[(size_t)i * k + j] -= t[j];
      for (i = n - 3; i >= 0; --i) {
        double* xi = b + (size_t)i * k;
        PentaMatrix::GemvAcc(&Y[(size_t)(i + 2) * kk], b + (size_t)(i + 1) * k, t.data(), k, true);
        for (int j = 0; j < k; ++j) xi[j] = xi[j] - t[j];
        PentaMatrix::GemvAcc(&Z[(size_t)(i + 2) * kk], b + (size_t)(i + 2) * k, t.data(), k, true);
        for (int j = 0; j < k; ++j) xi[j] = xi[j] - t[j];
      }
    }
  }
};

// Dense symmetric solve via LDL^T without pivoting (stand-in for Eigen's
// ldlt(), TO.cc:1395,2090); M is column-major n x n, overwritten.
inline bool DenseLdltSolve(std::vector<double>& M, int n, double* b) {
  // in-place LDL^T (lower)
  for (int j = 0; j < n; ++j) {
    double d = M[(size_t)j * n + j];
    for (int p = 0; p < j; ++p) d -= M[(size_t)p * n + j] * M[(size_t)p * n + j] * M[(size_t)p * n + p];
    M[(size_t)j * n + j] = d;
    if (d == 0.0) return false;
    for (int i = j + 1; i < n; ++i) {
      double s = M[(size_t)j * n + i];
      for (int p = 0; p < j; ++p) s -= M[(size_t)p * n + i] * M[(size_t)p * n + j] * M[(size_t)p * n + p];
      M[(size_t)j * n + i] = s / d;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int p = 0; p < i; ++p) b[i] -= M[(size_t)p * n + i] * b[p];
  for (int i = 0; i < n; ++i) b[i] /= M[(size_t)i * n + i];
  for (int i = n - 1; i >= 0; --i)
    for (int p = i + 1; p < n; ++p) b[i] -= M[(size_t)i * n + p] * b[p];
  return true;
}

}  // namespace oracle
