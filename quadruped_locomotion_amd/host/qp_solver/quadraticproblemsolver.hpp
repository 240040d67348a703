// qp_solver::QuadraticProblemSolver / QuadraticObjectiveFunction / LinearFunctionConstraints on top of the
// C-ABI (qp_solver/include/qp_solver/quadraticproblemsolver.h:87-89,
//        qp_solver/src/quadraticproblemsolver.cpp:65-97,133-207).
// Dense row-major std::vector matrices stand in for Eigen::MatrixXd.
#pragma once

#include <vector>

#include "qlamd/types.hpp"

namespace qp_solver {

struct Matrix { // rows x cols, row-major
  int rows = 0, cols = 0;
  std::vector<double> a;
  Matrix() = default;
  Matrix(int r, int c) : rows(r), cols(c), a(static_cast<size_t>(r) * c, 0.0) {}
  double &operator()(int i, int j) { return a[static_cast<size_t>(i) * cols + j]; }
  double operator()(int i, int j) const { return a[static_cast<size_t>(i) * cols + j]; }
};
typedef std::vector<double> Vector;

class QuadraticObjectiveFunction {
 public:
  bool setGlobalHessian(const Matrix &hessian) { G_ = hessian; return true; }          // :133-143
  bool setLinearTerm(const Vector &jacobian) { g0_ = jacobian; return true; }          // :145-156
  Matrix G_;
  Vector g0_;
};

class LinearFunctionConstraints {
 public:
  // A x <= b is stored as CI = -A' (one constraint per column), ci0 = b  (:158-169, :189-199)
  bool setGlobalInequalityConstraintJacobian(const Matrix &A) {
    CI_ = Matrix(A.cols, A.rows);
    for (int i = 0; i < A.rows; ++i)
      for (int j = 0; j < A.cols; ++j) CI_(j, i) = -A(i, j);
    return true;
  }
  bool setInequalityConstraintMaxValues(const Vector &b) { ci0_ = b; return true; }
  // equality Jacobian is taken as given, n x p (:170-181)
  bool setGlobalEqualityConstraintJacobian(const Matrix &Aeq) { CE_ = Aeq; return true; }
  bool setEqualityConstraintMaxValues(const Vector &beq) { ce0_ = beq; return true; }
  Matrix CI_, CE_;
  Vector ci0_, ce0_;
};

class QuadraticProblemSolver {
 public:
  explicit QuadraticProblemSolver(std::shared_ptr<qlamd::Context> ctx) : ctx_(std::move(ctx)) {}

  // The reference always returns true and lets QuadProg++ throw on a non-PD Hessian
  // (quadraticproblemsolver.cpp:96, QuadProg++.cc:692-699); here a failed solve returns false and
  // lastStatus() says why.
  bool minimize(const QuadraticObjectiveFunction &function, const LinearFunctionConstraints &constraints,
                Vector &params) {
    const int n = function.G_.rows, p = constraints.CE_.cols, m = constraints.CI_.cols;
    params.assign(n, 0.0);
    double f = 0.0;
    const int rc = qlamd_qp_solve_batch(ctx_->get(), n, p, m, function.G_.a.data(), function.g0_.data(),
                                        p ? constraints.CE_.a.data() : nullptr, p ? constraints.ce0_.data() : nullptr,
                                        m ? constraints.CI_.a.data() : nullptr, m ? constraints.ci0_.data() : nullptr, 1,
                                        params.data(), &f, &status_, QLAMD_MEM_HOST, nullptr);
    objective_ = f;
    return rc == QLAMD_OK && status_ == QLAMD_STATUS_OK;
  }
  int lastStatus() const { return status_; }
  double lastObjective() const { return objective_; }

 private:
  std::shared_ptr<qlamd::Context> ctx_;
  int32_t status_ = -1;
  double objective_ = 0.0;
};

} // namespace qp_solver
