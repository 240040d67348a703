// ooqpei::QuadraticProblemFormulation::solve on top of the C-ABI: the narrowest seam of the force distribution.
// The reference's ContactForceDistribution assembles A, S, b, W, C, c, D, d, f itself and hands them to
//     ooqpei::QuadraticProblemFormulation::solve(A_, S_, b_, W_, C_, c_, D_, d_, f_, x_)
// (balance_controller/src/contact_force_distribution/ContactForceDistribution.cpp:367 -- the first pass of
// addDesiredLegLoadConstraints -- and :490, solveOptimization; includes ContactForceDistribution.hpp:46-47; the library is
// third-party ooqp_eigen_interface on OOQP + MA27, not in the tree).  With this header on the include path instead,
// that file compiles against qlamd_weighted_lsq_qp_batch and keeps every line of its own assembly.
//
// Same argument order and meaning: finds x that minimises (Ax-b)'S(Ax-b) + x'Wx such that Cx = c and d <= Dx <= f;
// S and W are diagonal (Eigen::DiagonalMatrix in the reference: here the vector of their diagonals); a bound equal to
// std::numeric_limits<double>::max() means "none"; returns false when the solver fails (the reference then logs and
// returns false from solveOptimization, :490-494).  Dense row-major std::vector matrices stand in for
// Eigen::SparseMatrix<double, Eigen::RowMajor> / Eigen::VectorXd, as in qp_solver/quadraticproblemsolver.hpp.
#pragma once

#include <vector>

#include "qp_solver/quadraticproblemsolver.hpp"

namespace ooqpei {

typedef qp_solver::Matrix Matrix;
typedef qp_solver::Vector Vector;

class QuadraticProblemFormulation {
 public:
  // The reference's function is static and OOQP needs no handle; the device context is process-wide state set once
  // (RosBalanceController::init creates it), which keeps the call sites textually unchanged.
  static void setContext(std::shared_ptr<qlamd::Context> ctx) { context() = std::move(ctx); }

  static bool solve(const Matrix &A, const Vector &S, const Vector &b, const Vector &W, const Matrix &C, const Vector &c,
                    const Matrix &D, const Vector &d, const Vector &f, Vector &x) {
    lastStatus() = -1;
    if (!context()) return false;
    const int k = A.rows, n = A.cols, p = C.rows, m = D.rows;
    if (static_cast<int>(S.size()) != k || static_cast<int>(b.size()) != k || static_cast<int>(W.size()) != n ||
        static_cast<int>(c.size()) != p || static_cast<int>(d.size()) != m || static_cast<int>(f.size()) != m ||
        (p && C.cols != n) || (m && D.cols != n))
      return false;
    x.assign(n, 0.0);
    const int rc = qlamd_weighted_lsq_qp_batch(context()->get(), n, k, p, m, A.a.data(), S.data(), b.data(), W.data(),
                                               p ? C.a.data() : nullptr, p ? c.data() : nullptr, m ? D.a.data() : nullptr,
                                               m ? d.data() : nullptr, m ? f.data() : nullptr, 1, x.data(), &lastStatus(),
                                               QLAMD_MEM_HOST, nullptr);
    return rc == QLAMD_OK && lastStatus() == QLAMD_STATUS_OK;
  }
  static int32_t &lastStatus() { static int32_t s = -1; return s; }

 private:
  static std::shared_ptr<qlamd::Context> &context() { static std::shared_ptr<qlamd::Context> c; return c; }
};

} // namespace ooqpei
