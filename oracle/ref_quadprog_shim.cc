/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * extern "C" entry into the REFERENCE's own solver, quadprogpp::solve_quadprog
 * (qp_solver/src/QuadProg++.cc:52).  This file is ours; the reference sources
 * are compiled where they lie under /root/reference by oracle/Makefile and the
 * result goes to oracle/_ref/ (git-ignored, never committed).  It exists to
 * (1) pin oracle_quadprog.c against the real solver and (2) generate
 * tests/golden/.  Exceptions the reference throws become status codes.
 */
#include <cmath>
#include <stdexcept>

#include "qp_solver/QuadProg++.h"

extern "C" int ref_solve_quadprog(int n, int p, int m, const double *G, const double *g0,
                                  const double *CE, const double *ce0, const double *CI,
                                  const double *ci0, double *x, double *f_out) {
  quadprogpp::Matrix<double> Gm(n, n), CEm(n, p), CIm(n, m);
  quadprogpp::Vector<double> g0v(n), ce0v(p), ci0v(m), xv(n);
  for (int i = 0; i < n; i++) {
    g0v[i] = g0[i];
    for (int j = 0; j < n; j++) Gm[i][j] = G[i * n + j];
    for (int j = 0; j < p; j++) CEm[i][j] = CE[i * p + j];
    for (int j = 0; j < m; j++) CIm[i][j] = CI[i * m + j];
  }
  for (int j = 0; j < p; j++) ce0v[j] = ce0[j];
  for (int j = 0; j < m; j++) ci0v[j] = ci0[j];
  double f;
  try {
    f = quadprogpp::solve_quadprog(Gm, g0v, CEm, ce0v, CIm, ci0v, xv);
  } catch (const std::logic_error &) {
    *f_out = NAN;
    return 2; /* not positive definite / dimension error */
  } catch (const std::invalid_argument &) {
    *f_out = NAN;
    return 3; /* delete of a non-active constraint */
  }
  for (int i = 0; i < n; i++) x[i] = xv[i];
  *f_out = f;
  return std::isinf(f) ? 1 : 0;
}

/* `count` problems of one shape, `reps` passes: timing loop for tests/tools/cpu_qp_calibration.py (no Python in the loop). */
extern "C" int ref_solve_quadprog_batch(int n, int p, int m, int count, int reps, const double *G, const double *g0,
                                        const double *CI, const double *ci0, double *x) {
  int bad = 0;
  double f;
  for (int r = 0; r < reps; r++)
    for (int k = 0; k < count; k++)
      bad += ref_solve_quadprog(n, p, m, G + (size_t)k * n * n, g0 + (size_t)k * n, nullptr, nullptr, CI + (size_t)k * n * m,
                                ci0 + (size_t)k * m, x + (size_t)k * n, &f) != 0;
  return bad;
}
