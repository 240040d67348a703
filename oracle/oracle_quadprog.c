/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_quadprog.h).
 *
 * Goldfarb-Idnani dual active-set QP, restated in plain C from the reference's
 * qp_solver/src/QuadProg++.cc.  Each routine cites the lines it follows.  The
 * arithmetic order of every inner product is kept (ascending index) so that
 * results agree with the reference's compiled solver to the last bits.
 */
#include "oracle_quadprog.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS DBL_EPSILON

typedef struct {
  int n, p, m;
  double *L;      /* n x n: Cholesky factor of G, stored in G's own buffer     */
  double *J;      /* n x n: J = L^-T Q                                          */
  double *R;      /* n x n: upper-triangular R of the active normals            */
  double *d, *z, *np, *xold;
  double *s, *r, *u, *uold;
  int *A, *Aold, *iai;
  unsigned char *iaexcl;
  double Rnorm;
  int iq;
} gi_ws;

#define AT(M, i, j) ((M)[(size_t)(i) * (size_t)w->n + (size_t)(j)])

/* QuadProg++.cc:647-664 -- hypot without overflow */
static double gi_hyp(double a, double b) {
  double a1 = fabs(a), b1 = fabs(b), t;
  if (a1 > b1) { t = b1 / a1; return a1 * sqrt(1.0 + t * t); }
  if (b1 > a1) { t = a1 / b1; return b1 * sqrt(1.0 + t * t); }
  return a1 * sqrt(2.0);
}

static double gi_dot(const double *a, const double *b, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; i++) acc += a[i] * b[i];
  return acc;
}

/* QuadProg++.cc:678-709 -- in-place Cholesky, lower factor mirrored to the
 * upper triangle.  Returns 0 when a pivot is <= 0 (the reference throws). */
static int gi_cholesky(gi_ws *w) {
  int n = w->n;
  for (int i = 0; i < n; i++) {
    for (int j = i; j < n; j++) {
      double acc = AT(w->L, i, j);
      for (int k = i - 1; k >= 0; k--) acc -= AT(w->L, i, k) * AT(w->L, j, k);
      if (i == j) {
        if (acc <= 0.0) return 0;
        AT(w->L, i, i) = sqrt(acc);
      } else {
        AT(w->L, j, i) = acc / AT(w->L, i, i);
      }
    }
    for (int k = i + 1; k < n; k++) AT(w->L, i, k) = AT(w->L, k, i);
  }
  return 1;
}

/* QuadProg++.cc:722-734 */
static void gi_forward(const gi_ws *w, double *y, const double *b) {
  int n = w->n;
  y[0] = b[0] / AT(w->L, 0, 0);
  for (int i = 1; i < n; i++) {
    y[i] = b[i];
    for (int j = 0; j < i; j++) y[i] -= AT(w->L, i, j) * y[j];
    y[i] = y[i] / AT(w->L, i, i);
  }
}

/* QuadProg++.cc:736-748 (uses the mirrored upper triangle) */
static void gi_backward(const gi_ws *w, double *x, const double *y) {
  int n = w->n;
  x[n - 1] = y[n - 1] / AT(w->L, n - 1, n - 1);
  for (int i = n - 2; i >= 0; i--) {
    x[i] = y[i];
    for (int j = i + 1; j < n; j++) x[i] -= AT(w->L, i, j) * x[j];
    x[i] = x[i] / AT(w->L, i, i);
  }
}

/* QuadProg++.cc:448-461: d = J' np */
static void gi_compute_d(gi_ws *w) {
  int n = w->n;
  for (int i = 0; i < n; i++) {
    double acc = 0.0;
    for (int j = 0; j < n; j++) acc += AT(w->J, j, i) * w->np[j];
    w->d[i] = acc;
  }
}

/* QuadProg++.cc:463-474: z = J[:, iq:] d[iq:] */
static void gi_update_z(gi_ws *w) {
  int n = w->n;
  for (int i = 0; i < n; i++) {
    w->z[i] = 0.0;
    for (int j = w->iq; j < n; j++) w->z[i] += AT(w->J, i, j) * w->d[j];
  }
}

/* QuadProg++.cc:476-489: r = R^-1 d (leading iq x iq block) */
static void gi_update_r(gi_ws *w) {
  for (int i = w->iq - 1; i >= 0; i--) {
    double acc = 0.0;
    for (int j = i + 1; j < w->iq; j++) acc += AT(w->R, i, j) * w->r[j];
    w->r[i] = (w->d[i] - acc) / AT(w->R, i, i);
  }
}

/* QuadProg++.cc:491-560.  Returns 0 when the new normal is (numerically)
 * dependent on the active ones. */
static int gi_add_constraint(gi_ws *w) {
  int n = w->n;
  for (int j = n - 1; j >= w->iq + 1; j--) {
    double cc = w->d[j - 1], ss = w->d[j];
    double h = gi_hyp(cc, ss);
    if (fabs(h) < EPS) continue;
    w->d[j] = 0.0;
    ss = ss / h;
    cc = cc / h;
    if (cc < 0.0) {
      cc = -cc;
      ss = -ss;
      w->d[j - 1] = -h;
    } else {
      w->d[j - 1] = h;
    }
    double xny = ss / (1.0 + cc);
    for (int k = 0; k < n; k++) {
      double t1 = AT(w->J, k, j - 1), t2 = AT(w->J, k, j);
      AT(w->J, k, j - 1) = t1 * cc + t2 * ss;
      AT(w->J, k, j) = xny * (t1 + AT(w->J, k, j - 1)) - t2;
    }
  }
  w->iq++;
  for (int i = 0; i < w->iq; i++) AT(w->R, i, w->iq - 1) = w->d[i];
  if (fabs(w->d[w->iq - 1]) <= EPS * w->Rnorm) return 0;
  w->Rnorm = fmax(w->Rnorm, fabs(w->d[w->iq - 1]));
  return 1;
}

/* QuadProg++.cc:562-645.  Returns 0 if constraint l is not active. */
static int gi_delete_constraint(gi_ws *w, int l) {
  int n = w->n, qq = -1;
  for (int i = w->p; i < w->iq; i++)
    if (w->A[i] == l) { qq = i; break; }
  if (qq < 0) return 0;
  for (int i = qq; i < w->iq - 1; i++) {
    w->A[i] = w->A[i + 1];
    w->u[i] = w->u[i + 1];
    for (int j = 0; j < n; j++) AT(w->R, j, i) = AT(w->R, j, i + 1);
  }
  w->A[w->iq - 1] = w->A[w->iq];
  w->u[w->iq - 1] = w->u[w->iq];
  w->A[w->iq] = 0;
  w->u[w->iq] = 0.0;
  for (int j = 0; j < w->iq; j++) AT(w->R, j, w->iq - 1) = 0.0;
  w->iq--;
  if (w->iq == 0) return 1;
  for (int j = qq; j < w->iq; j++) {
    double cc = AT(w->R, j, j), ss = AT(w->R, j + 1, j);
    double h = gi_hyp(cc, ss);
    if (fabs(h) < EPS) continue;
    cc = cc / h;
    ss = ss / h;
    AT(w->R, j + 1, j) = 0.0;
    if (cc < 0.0) {
      AT(w->R, j, j) = -h;
      cc = -cc;
      ss = -ss;
    } else {
      AT(w->R, j, j) = h;
    }
    double xny = ss / (1.0 + cc);
    for (int k = j + 1; k < w->iq; k++) {
      double t1 = AT(w->R, j, k), t2 = AT(w->R, j + 1, k);
      AT(w->R, j, k) = t1 * cc + t2 * ss;
      AT(w->R, j + 1, k) = xny * (t1 + AT(w->R, j, k)) - t2;
    }
    for (int k = 0; k < n; k++) {
      double t1 = AT(w->J, k, j), t2 = AT(w->J, k, j + 1);
      AT(w->J, k, j) = t1 * cc + t2 * ss;
      AT(w->J, k, j + 1) = xny * (AT(w->J, k, j) + t1) - t2;
    }
  }
  return 1;
}

int oracle_solve_quadprog(int n, int p, int m, double *G, const double *g0,
                          const double *CE, const double *ce0, const double *CI,
                          const double *ci0, double *x, double *f_out,
                          int *active, int *n_active, int *iters) {
  gi_ws ws, *w = &ws;
  int mp = m + p + 1; /* +1: the reference writes u[iq]/A[iq] one past m+p-1 only
                         when iq < m+p, but keep a spare slot for safety */
  int status = ORACLE_QP_OK;
  double f_value = 0.0, c1, c2, psi, ss, t, t1, t2;
  int ip = 0, l = 0, iter = 0;
  const double inf = INFINITY;
  long guard = 0;

  memset(w, 0, sizeof(*w));
  w->n = n; w->p = p; w->m = m; w->L = G;
  size_t nn = (size_t)n * (size_t)n;
  double *buf = (double *)calloc(2 * nn + 4 * (size_t)n + 4 * (size_t)mp, sizeof(double));
  int *ibuf = (int *)calloc(3 * (size_t)mp, sizeof(int));
  unsigned char *bbuf = (unsigned char *)calloc((size_t)mp, 1);
  w->J = buf; w->R = buf + nn;
  w->d = buf + 2 * nn; w->z = w->d + n; w->np = w->z + n; w->xold = w->np + n;
  w->s = w->xold + n; w->r = w->s + mp; w->u = w->r + mp; w->uold = w->u + mp;
  w->A = ibuf; w->Aold = ibuf + mp; w->iai = ibuf + 2 * mp;
  w->iaexcl = bbuf;

  /* preprocessing, QuadProg++.cc:117-167 */
  c1 = 0.0;
  for (int i = 0; i < n; i++) c1 += AT(G, i, i);
  if (!gi_cholesky(w)) { status = ORACLE_QP_NOT_PD; f_value = NAN; goto done; }
  w->Rnorm = 1.0;
  c2 = 0.0;
  for (int i = 0; i < n; i++) {
    w->d[i] = 1.0;
    gi_forward(w, w->z, w->d);
    for (int j = 0; j < n; j++) AT(w->J, i, j) = w->z[j];
    c2 += w->z[i];
    w->d[i] = 0.0;
  }
  gi_forward(w, w->z, g0);        /* cholesky_solve, :711-720 */
  gi_backward(w, x, w->z);
  for (int i = 0; i < n; i++) x[i] = -x[i];
  f_value = 0.5 * gi_dot(g0, x, n);

  /* equality constraints, QuadProg++.cc:169-210 (failed adds are ignored there) */
  w->iq = 0;
  for (int i = 0; i < p; i++) {
    for (int j = 0; j < n; j++) w->np[j] = CE[(size_t)j * p + i];
    gi_compute_d(w);
    gi_update_z(w);
    gi_update_r(w);
    t2 = 0.0;
    if (fabs(gi_dot(w->z, w->z, n)) > EPS)
      t2 = (-gi_dot(w->np, x, n) - ce0[i]) / gi_dot(w->z, w->np, n);
    for (int k = 0; k < n; k++) x[k] += t2 * w->z[k];
    w->u[w->iq] = t2;
    for (int k = 0; k < w->iq; k++) w->u[k] -= t2 * w->r[k];
    f_value += 0.5 * (t2 * t2) * gi_dot(w->z, w->np, n);
    w->A[i] = -i - 1;
    (void)gi_add_constraint(w);
  }
  for (int i = 0; i < m; i++) w->iai[i] = i;

  /* main loop, labels l1 / l2 / l2a of QuadProg++.cc:216-445 */
  enum { L1, L2, L2A } at = L1;
  ss = 0.0;
  for (;;) {
    if (++guard > 100000) { status = ORACLE_QP_BAD_DELETE + 1; break; } /* oracle-only guard */
    if (at == L1) {
      iter++;
      for (int i = p; i < w->iq; i++) w->iai[w->A[i]] = -1;
      ss = 0.0; psi = 0.0; ip = 0;
      for (int i = 0; i < m; i++) {
        w->iaexcl[i] = 1;
        double acc = 0.0;
        for (int j = 0; j < n; j++) acc += CI[(size_t)j * m + i] * x[j];
        acc += ci0[i];
        w->s[i] = acc;
        psi += fmin(0.0, acc);
      }
      if (fabs(psi) <= m * EPS * c1 * c2 * 100.0) break; /* :246-250 */
      for (int i = 0; i < w->iq; i++) { w->uold[i] = w->u[i]; w->Aold[i] = w->A[i]; }
      for (int i = 0; i < n; i++) w->xold[i] = x[i];
      at = L2;
    }
    if (at == L2) {
      for (int i = 0; i < m; i++)
        if (w->s[i] < ss && w->iai[i] != -1 && w->iaexcl[i]) { ss = w->s[i]; ip = i; }
      if (ss >= 0.0) break; /* :271-274 */
      for (int i = 0; i < n; i++) w->np[i] = CI[(size_t)i * m + ip];
      w->u[w->iq] = 0.0;
      w->A[w->iq] = ip;
      at = L2A;
    }
    /* l2a */
    gi_compute_d(w);
    gi_update_z(w);
    gi_update_r(w);
    l = 0;
    t1 = inf;
    for (int k = p; k < w->iq; k++)
      if (w->r[k] > 0.0 && w->u[k] / w->r[k] < t1) { t1 = w->u[k] / w->r[k]; l = w->A[k]; }
    if (fabs(gi_dot(w->z, w->z, n)) > EPS) {
      t2 = -w->s[ip] / gi_dot(w->z, w->np, n);
      if (t2 < 0) t2 = inf;
    } else {
      t2 = inf;
    }
    t = fmin(t1, t2);
    if (t >= inf) { status = ORACLE_QP_INFEASIBLE; f_value = inf; break; } /* :339-344 */
    if (t2 >= inf) { /* dual step only, :346-362 */
      for (int k = 0; k < w->iq; k++) w->u[k] -= t * w->r[k];
      w->u[w->iq] += t;
      w->iai[l] = l;
      if (!gi_delete_constraint(w, l)) { status = ORACLE_QP_BAD_DELETE; break; }
      at = L2A;
      continue;
    }
    /* primal + dual step, :364-374 */
    for (int k = 0; k < n; k++) x[k] += t * w->z[k];
    f_value += t * gi_dot(w->z, w->np, n) * (0.5 * t + w->u[w->iq]);
    for (int k = 0; k < w->iq; k++) w->u[k] -= t * w->r[k];
    w->u[w->iq] += t;
    if (fabs(t - t2) < EPS) { /* full step, :384-421 */
      if (!gi_add_constraint(w)) {
        w->iaexcl[ip] = 0;
        if (!gi_delete_constraint(w, ip)) { status = ORACLE_QP_BAD_DELETE; break; }
        for (int i = 0; i < m; i++) w->iai[i] = i;
        for (int i = p; i < w->iq; i++) {
          w->A[i] = w->Aold[i];
          w->u[i] = w->uold[i];
          w->iai[w->A[i]] = -1;
        }
        for (int i = 0; i < n; i++) x[i] = w->xold[i];
        at = L2;
        continue;
      }
      w->iai[ip] = -1;
      at = L1;
      continue;
    }
    /* partial step, :423-445 */
    w->iai[l] = l;
    if (!gi_delete_constraint(w, l)) { status = ORACLE_QP_BAD_DELETE; break; }
    {
      double acc = 0.0;
      for (int k = 0; k < n; k++) acc += CI[(size_t)k * m + ip] * x[k];
      w->s[ip] = acc + ci0[ip];
    }
    at = L2A;
  }

done:
  if (f_out) *f_out = f_value;
  if (n_active) *n_active = w->iq;
  if (active) for (int i = 0; i < w->iq && i < m + p; i++) active[i] = w->A[i];
  if (iters) *iters = iter;
  free(buf); free(ibuf); free(bbuf);
  return status;
}

/* `count` problems of one shape, `reps` passes: timing loop for tests/tools/cpu_qp_calibration.py (no Python in the loop). */
int oracle_solve_quadprog_batch(int n, int p, int m, int count, int reps, const double *G, const double *g0,
                                const double *CI, const double *ci0, double *x) {
  int bad = 0, active[64], nact, iters;
  double f, Gw[12 * 12];
  if (n > 12 || m + p > 63) return -1;
  for (int r = 0; r < reps; r++)
    for (int k = 0; k < count; k++) {
      memcpy(Gw, G + (size_t)k * n * n, sizeof(double) * (size_t)n * (size_t)n); /* the solver factorises G in place */
      bad += oracle_solve_quadprog(n, p, m, Gw, g0 + (size_t)k * n, NULL, NULL, CI + (size_t)k * n * m, ci0 + (size_t)k * m,
                                   x + (size_t)k * n, &f, active, &nact, &iters) != 0;
    }
  return bad;
}
