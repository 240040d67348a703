/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_wholebody.h: PARITY UNPINNED, published algorithms restated). */
#include "oracle_wholebody.h"

#include <math.h>
#include <string.h>

#include "oracle_balance.h"
#include "oracle_model.h"
#include "oracle_quadprog.h"
#include "qlamd_robot_constants.h"

/* Spatial vectors are [angular ; linear] internally (Featherstone's order); the public interface uses
 * [linear ; angular] for the base, like the reference's wrench (F, T) (ContactForceDistribution.cpp:168-206). */

#define NB 13 /* bodies: 0 = base, 1 + 3 leg + k = link k of the leg */

typedef struct { double E[9]; double r[3]; } xform_t; /* parent -> child: E rotates parent coordinates into
                                                         child coordinates, r = child origin in the parent */

static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mv3(const double *A, const double *v, double *o) {
  for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void mtv3(const double *A, const double *v, double *o) {
  for (int i = 0; i < 3; i++) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
static void rpy_matrix(const double rpy[3], double *R) { /* URDF fixed-axis: Rz(yaw) Ry(pitch) Rx(roll) */
  const double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

/* motion vector parent -> child coordinates: [E w ; E (v - r x w)]  (RBDA eq. 2.24) */
static void x_motion(const xform_t *X, const double *m, double *o) {
  double rw[3], t[3];
  cross3(X->r, m, rw);
  for (int i = 0; i < 3; i++) t[i] = m[3 + i] - rw[i];
  mv3(X->E, m, o);
  mv3(X->E, t, o + 3);
}
/* force vector child -> parent coordinates (X^T): [E'n + r x E'f ; E'f]  (RBDA eq. 2.25) */
static void xt_force(const xform_t *X, const double *f, double *o) {
  double n[3], ff[3], rf[3];
  mtv3(X->E, f, n);
  mtv3(X->E, f + 3, ff);
  cross3(X->r, ff, rf);
  for (int i = 0; i < 3; i++) { o[i] = n[i] + rf[i]; o[3 + i] = ff[i]; }
}
/* 6 x 6 matrix of the motion transform, row-major */
static void x_matrix(const xform_t *X, double *M) {
  memset(M, 0, 36 * sizeof(double));
  double rx[9] = {0, -X->r[2], X->r[1], X->r[2], 0, -X->r[0], -X->r[1], X->r[0], 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      M[6 * i + j] = X->E[3 * i + j];
      M[6 * (i + 3) + (j + 3)] = X->E[3 * i + j];
      double acc = 0.0; /* -E rx */
      for (int k = 0; k < 3; k++) acc -= X->E[3 * i + k] * rx[3 * k + j];
      M[6 * (i + 3) + j] = acc;
    }
}
/* spatial inertia of a rigid body: mass m, centre of mass c, rotational inertia Ic about c (RBDA eq. 2.63) */
static void rigid_inertia(double m, const double c[3], const double ic[6], double *I) {
  const double cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
  const double Ic[9] = {ic[0], ic[1], ic[2], ic[1], ic[3], ic[4], ic[2], ic[4], ic[5]};
  memset(I, 0, 36 * sizeof(double));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double cc = 0.0; /* cx cx' */
      for (int k = 0; k < 3; k++) cc += cx[3 * i + k] * cx[3 * j + k];
      I[6 * i + j] = Ic[3 * i + j] + m * cc;
      I[6 * i + (j + 3)] = m * cx[3 * i + j];
      I[6 * (i + 3) + j] = m * cx[3 * j + i];
    }
  for (int i = 0; i < 3; i++) I[6 * (i + 3) + (i + 3)] = m;
}
static void mat6_vec(const double *A, const double *v, double *o) {
  for (int i = 0; i < 6; i++) {
    double acc = 0.0;
    for (int j = 0; j < 6; j++) acc += A[6 * i + j] * v[j];
    o[i] = acc;
  }
}
/* I += X' Ia X */
static void add_transformed_inertia(double *I, const xform_t *X, const double *Ia) {
  double Xm[36], T[36];
  x_matrix(X, Xm);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double acc = 0.0;
      for (int k = 0; k < 6; k++) acc += Ia[6 * i + k] * Xm[6 * k + j];
      T[6 * i + j] = acc;
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double acc = 0.0;
      for (int k = 0; k < 6; k++) acc += Xm[6 * k + i] * T[6 * k + j];
      I[6 * i + j] += acc;
    }
}
/* v x m (motion) and v x* f (force), RBDA eq. 2.31-2.32 */
static void crm(const double *v, const double *m, double *o) {
  double a[3], b[3], c[3];
  cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
static void crf(const double *v, const double *f, double *o) {
  double a[3], b[3], c[3];
  cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}

typedef struct {
  xform_t X[NB];      /* X[i]: parent(i) -> i, i >= 1 */
  double I[NB][36];   /* link inertias in link coordinates (foot link folded into link 3) */
  int parent[NB];
} tree_t;

static void build_tree(const double q[12], tree_t *T) {
  rigid_inertia(QLAMD_BASE_MASS, QLAMD_BASE_COM, QLAMD_BASE_INERTIA, T->I[0]);
  T->parent[0] = -1;
  for (int l = 0; l < 4; l++)
    for (int k = 0; k < 3; k++) {
      const int i = 1 + 3 * l + k;
      T->parent[i] = k == 0 ? 0 : i - 1;
      double R0[9], R[9];
      rpy_matrix(QLAMD_JOINT_RPY[l][k], R0);
      const double c = cos(q[3 * l + k]), s = sin(q[3 * l + k]);
      const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double acc = 0.0;
          for (int m = 0; m < 3; m++) acc += R0[3 * a + m] * Rz[3 * m + b];
          R[3 * a + b] = acc; /* child -> parent rotation */
        }
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) T->X[i].E[3 * a + b] = R[3 * b + a];
      memcpy(T->X[i].r, QLAMD_JOINT_XYZ[l][k], 3 * sizeof(double));
      rigid_inertia(QLAMD_LINK_MASS[l][k], QLAMD_LINK_COM[l][k], QLAMD_LINK_INERTIA[l][k], T->I[i]);
      if (k == 2) { /* fixed foot link */
        xform_t Xf;
        double Rf[9], If[36];
        rpy_matrix(QLAMD_JOINT_RPY[l][3], Rf);
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) Xf.E[3 * a + b] = Rf[3 * b + a];
        memcpy(Xf.r, QLAMD_JOINT_XYZ[l][3], 3 * sizeof(double));
        rigid_inertia(QLAMD_LINK_MASS[l][3], QLAMD_LINK_COM[l][3], QLAMD_LINK_INERTIA[l][3], If);
        add_transformed_inertia(T->I[i], &Xf, If);
      }
    }
}

/* public order [v ; w ; qd]  <->  internal base spatial vector [w ; v] */
static void base_in(const double *pub, double *sp) { for (int i = 0; i < 3; i++) { sp[i] = pub[3 + i]; sp[3 + i] = pub[i]; } }
static void base_out(const double *sp, double *pub) { for (int i = 0; i < 3; i++) { pub[i] = sp[3 + i]; pub[3 + i] = sp[i]; } }

/* Composite-rigid-body algorithm, floating base (RBDA Table 9.6). */
void oracle_wb_mass_matrix(const double q[12], double M[324]) {
  tree_t T;
  build_tree(q, &T);
  double Ic[NB][36];
  memcpy(Ic, T.I, sizeof(Ic));
  for (int i = NB - 1; i >= 1; i--) add_transformed_inertia(Ic[T.parent[i]], &T.X[i], Ic[i]);
  memset(M, 0, 324 * sizeof(double));
  /* base block, internal order -> public order */
  for (int a = 0; a < 6; a++)
    for (int b = 0; b < 6; b++) {
      const int pa = a < 3 ? a + 3 : a - 3, pb = b < 3 ? b + 3 : b - 3;
      M[18 * pa + pb] = Ic[0][6 * a + b];
    }
  for (int i = 1; i < NB; i++) {
    const double S[6] = {0, 0, 1, 0, 0, 0};
    double F[6], Fp[6];
    mat6_vec(Ic[i], S, F);
    M[18 * (5 + i) + (5 + i)] = F[2];
    int j = i;
    while (T.parent[j] > 0) {
      xt_force(&T.X[j], F, Fp);
      memcpy(F, Fp, sizeof(F));
      j = T.parent[j];
      M[18 * (5 + i) + (5 + j)] = M[18 * (5 + j) + (5 + i)] = F[2];
    }
    xt_force(&T.X[j], F, Fp); /* into base coordinates */
    double Fpub[6];
    base_out(Fp, Fpub);
    for (int a = 0; a < 6; a++) M[18 * a + (5 + i)] = M[18 * (5 + i) + a] = Fpub[a];
  }
}

/* Recursive Newton-Euler in link coordinates (RBDA Table 5.1) with a moving base: the base's own spatial
 * acceleration is part of nu' and gravity enters as the fictitious base acceleration -a_g (section 9.4/5.3). */
void oracle_wb_inverse_dynamics(const double q[12], const double base_quat[4], const double nu[18],
                                const double nudot[18], double gravity, double out[18]) {
  tree_t T;
  build_tree(q, &T);
  double Rm[9], gW[3] = {0.0, 0.0, -gravity}, gB[3];
  oracle_quat_to_matrix(base_quat, Rm);
  mtv3(Rm, gW, gB);
  double v[NB][6], a[NB][6], f[NB][6];
  base_in(nu, v[0]);
  base_in(nudot, a[0]);
  for (int i = 0; i < 3; i++) a[0][3 + i] -= gB[i];
  for (int i = 0; i < NB; i++) {
    if (i > 0) {
      const int p = T.parent[i];
      const double qd = nu[5 + i], qdd = nudot[5 + i];
      double vj[6] = {0, 0, qd, 0, 0, 0}, c[6];
      x_motion(&T.X[i], v[p], v[i]);
      v[i][2] += qd;
      x_motion(&T.X[i], a[p], a[i]);
      crm(v[i], vj, c);
      for (int k = 0; k < 6; k++) a[i][k] += c[k];
      a[i][2] += qdd;
    }
    double Ia[6], Iv[6], vIv[6];
    mat6_vec(T.I[i], a[i], Ia);
    mat6_vec(T.I[i], v[i], Iv);
    crf(v[i], Iv, vIv);
    for (int k = 0; k < 6; k++) f[i][k] = Ia[k] + vIv[k];
  }
  for (int i = NB - 1; i >= 1; i--) {
    out[5 + i] = f[i][2];
    double fp[6];
    xt_force(&T.X[i], f[i], fp);
    for (int k = 0; k < 6; k++) f[T.parent[i]][k] += fp[k];
  }
  base_out(f[0], out);
}

void oracle_wb_nonlinear_effects(const double q[12], const double base_quat[4], const double nu[18], double gravity,
                                 double h[18]) {
  double zero[18];
  memset(zero, 0, sizeof(zero));
  oracle_wb_inverse_dynamics(q, base_quat, nu, zero, gravity, h);
}

void oracle_wb_contact_jacobian(const double q[12], double Jc[216]) {
  memset(Jc, 0, 216 * sizeof(double));
  for (int l = 0; l < 4; l++) {
    double r[3], J[9];
    oracle_leg_fk(l, q + 3 * l, r, NULL);
    oracle_leg_jacobian(l, q + 3 * l, J);
    const double mrx[9] = {0, r[2], -r[1], -r[2], 0, r[0], r[1], -r[0], 0}; /* -[r]x */
    for (int a = 0; a < 3; a++) {
      Jc[18 * (3 * l + a) + a] = 1.0;
      for (int b = 0; b < 3; b++) {
        Jc[18 * (3 * l + a) + 3 + b] = mrx[3 * a + b];
        Jc[18 * (3 * l + a) + 6 + 3 * l + b] = J[3 * a + b];
      }
    }
  }
}

/* ---- energies by plain kinematics: every link's frame in the base, its centre-of-mass velocity and angular
 *      velocity from the chain, T = sum 1/2 m |vc|^2 + 1/2 w' R Ic R' w ------------------------------------ */
static void link_frames(int leg, const double q[3], double R[4][9], double p[4][3]) {
  double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};
  for (int k = 0; k < 4; k++) {
    double R0[9], Rs[9], Rn[9], t[3];
    rpy_matrix(QLAMD_JOINT_RPY[leg][k], R0);
    if (k < 3) {
      const double c = cos(q[k]), s = sin(q[k]);
      const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double acc = 0.0;
          for (int m = 0; m < 3; m++) acc += R0[3 * a + m] * Rz[3 * m + b];
          Rs[3 * a + b] = acc;
        }
    } else {
      memcpy(Rs, R0, sizeof(Rs));
    }
    mv3(Rc, QLAMD_JOINT_XYZ[leg][k], t);
    for (int a = 0; a < 3; a++) pc[a] += t[a];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        double acc = 0.0;
        for (int m = 0; m < 3; m++) acc += Rc[3 * a + m] * Rs[3 * m + b];
        Rn[3 * a + b] = acc;
      }
    memcpy(Rc, Rn, sizeof(Rc));
    memcpy(R[k], Rc, sizeof(Rc));
    memcpy(p[k], pc, sizeof(pc));
  }
}

static double body_energy(double m, const double ic[6], const double *R, const double vc[3], const double w[3]) {
  double wl[3];
  if (R) mtv3(R, w, wl); else memcpy(wl, w, sizeof(wl));
  const double Iw[3] = {ic[0] * wl[0] + ic[1] * wl[1] + ic[2] * wl[2], ic[1] * wl[0] + ic[3] * wl[1] + ic[4] * wl[2],
                        ic[2] * wl[0] + ic[4] * wl[1] + ic[5] * wl[2]};
  return 0.5 * m * (vc[0] * vc[0] + vc[1] * vc[1] + vc[2] * vc[2]) + 0.5 * (wl[0] * Iw[0] + wl[1] * Iw[1] + wl[2] * Iw[2]);
}

double oracle_wb_kinetic_energy(const double q[12], const double nu[18]) {
  const double *v = nu, *w = nu + 3;
  double wc[3], vc[3], E;
  cross3(w, QLAMD_BASE_COM, wc);
  for (int i = 0; i < 3; i++) vc[i] = v[i] + wc[i];
  E = body_energy(QLAMD_BASE_MASS, QLAMD_BASE_INERTIA, NULL, vc, w);
  for (int l = 0; l < 4; l++) {
    double R[4][9], p[4][3];
    link_frames(l, q + 3 * l, R, p);
    double wl[3] = {w[0], w[1], w[2]};
    /* velocity of a point x of link k: v + w x x + sum_{j<=k} qd_j z_j x (x - p_j) */
    for (int k = 0; k < 4; k++) {
      double c[3], rc[3];
      mv3(R[k], QLAMD_LINK_COM[l][k], rc);
      for (int i = 0; i < 3; i++) c[i] = p[k][i] + rc[i];
      double wx[3];
      cross3(w, c, wx);
      for (int i = 0; i < 3; i++) vc[i] = v[i] + wx[i];
      if (k < 3) {
        const double z[3] = {R[k][2], R[k][5], R[k][8]};
        for (int i = 0; i < 3; i++) wl[i] += nu[6 + 3 * l + k] * z[i];
      }
      for (int j = 0; j <= (k < 3 ? k : 2); j++) {
        const double z[3] = {R[j][2], R[j][5], R[j][8]};
        const double d[3] = {c[0] - p[j][0], c[1] - p[j][1], c[2] - p[j][2]};
        double zd[3];
        cross3(z, d, zd);
        for (int i = 0; i < 3; i++) vc[i] += nu[6 + 3 * l + j] * zd[i];
      }
      E += body_energy(QLAMD_LINK_MASS[l][k], QLAMD_LINK_INERTIA[l][k], R[k], vc, wl);
    }
  }
  return E;
}

double oracle_wb_potential_energy(const double q[12], const double base_pos[3], const double base_quat[4], double gravity) {
  double Rm[9], cw[3];
  oracle_quat_to_matrix(base_quat, Rm);
  mv3(Rm, QLAMD_BASE_COM, cw);
  double U = QLAMD_BASE_MASS * gravity * (base_pos[2] + cw[2]);
  for (int l = 0; l < 4; l++) {
    double R[4][9], p[4][3];
    link_frames(l, q + 3 * l, R, p);
    for (int k = 0; k < 4; k++) {
      double rc[3], c[3];
      mv3(R[k], QLAMD_LINK_COM[l][k], rc);
      for (int i = 0; i < 3; i++) c[i] = p[k][i] + rc[i];
      mv3(Rm, c, cw);
      U += QLAMD_LINK_MASS[l][k] * gravity * (base_pos[2] + cw[2]);
    }
  }
  return U;
}

/* ---------------------------------------------------------------- whole-body QP ------------------------- */
void oracle_wb_default_params(oracle_wb_params *p) {
  oracle_balance_params b;
  oracle_balance_default_params(&b);
  memcpy(p->force_weights, b.force_weights, sizeof(p->force_weights)); /* controller_gains.yaml:28-38 */
  p->regularizer = b.regularizer;
  p->torque_weight = 1e-3;
  p->friction = b.friction;
  p->min_normal_force = b.min_normal_force;
  p->torque_limit = b.torque_limit; /* +-300, ros_balance_controller.cpp:451-454 */
  p->gravity = 9.81;                /* RBDL's default gravity, the one the swing-leg model uses (model_test_header.cpp:229-244) */
}

int oracle_wb_step(const oracle_wb_params *prm, const double q[12], const double qd[12], const double base_quat[4],
                   const double base_linvel_world[3], const double base_angvel_base[3], const double a_des[6],
                   const double *qdd_des, const uint8_t stance[4], const double *normals_world, double tau[12],
                   double grf[12]) {
  double Rm[9], nu[18], nudot[18], M[324], h[18], gen[18];
  oracle_quat_to_matrix(base_quat, Rm);
  mtv3(Rm, base_linvel_world, nu);
  memcpy(nu + 3, base_angvel_base, 3 * sizeof(double));
  memcpy(nu + 6, qd, 12 * sizeof(double));
  memcpy(nudot, a_des, 6 * sizeof(double));
  for (int i = 0; i < 12; i++) nudot[6 + i] = qdd_des ? qdd_des[i] : 0.0;
  oracle_wb_mass_matrix(q, M);
  oracle_wb_nonlinear_effects(q, base_quat, nu, prm->gravity, h);
  for (int i = 0; i < 18; i++) {
    double acc = h[i];
    for (int j = 0; j < 18; j++) acc += M[18 * i + j] * nudot[j];
    gen[i] = acc; /* [b ; tau0] */
  }
  for (int i = 0; i < 12; i++) { tau[i] = gen[6 + i]; grf[i] = 0.0; }
  int legs[4], nS = 0;
  for (int l = 0; l < 4; l++) if (stance[l]) legs[nS++] = l;
  if (nS == 0) return ORACLE_QP_OK;

  const int nf = 3 * nS, n = 6 * nS, p = 3 * nS, m = 11 * nS;
  double G[24 * 24], g0[24], CE[24 * 12], ce0[12], CI[24 * 44], ci0[44], x[24], fval;
  int active[24 + 44 + 12], nact = 0, iters = 0;
  memset(G, 0, sizeof(G)); memset(g0, 0, sizeof(g0)); memset(CE, 0, sizeof(CE)); memset(CI, 0, sizeof(CI));
  double A[6 * 12], Jl[4][9];
  memset(A, 0, sizeof(A));
  const double ez[3] = {0, 0, 1}, ey[3] = {0, 1, 0};
  double yB[3];
  mtv3(Rm, ey, yB);
  for (int k = 0; k < nS; k++) {
    const int l = legs[k];
    double r[3];
    oracle_leg_fk(l, q + 3 * l, r, NULL);
    oracle_leg_jacobian(l, q + 3 * l, Jl[k]);
    for (int i = 0; i < 3; i++) A[i * nf + 3 * k + i] = 1.0;
    A[3 * nf + 3 * k + 1] = -r[2]; A[3 * nf + 3 * k + 2] = r[1];
    A[4 * nf + 3 * k + 0] = r[2];  A[4 * nf + 3 * k + 2] = -r[0];
    A[5 * nf + 3 * k + 0] = -r[1]; A[5 * nf + 3 * k + 1] = r[0];
  }
  for (int i = 0; i < nf; i++) {
    for (int j = 0; j < nf; j++) {
      double acc = 0.0;
      for (int k = 0; k < 6; k++) acc += A[k * nf + i] * prm->force_weights[k] * A[k * nf + j];
      G[i * n + j] = acc + (i == j ? prm->regularizer : 0.0);
    }
    double acc = 0.0;
    for (int k = 0; k < 6; k++) acc += A[k * nf + i] * prm->force_weights[k] * gen[k];
    g0[i] = -acc;
  }
  for (int i = 0; i < nf; i++) G[(nf + i) * n + (nf + i)] = prm->torque_weight;
  /* equalities  J_leg' f + tau - tau0 = 0 :  column e = 3k + j */
  for (int k = 0; k < nS; k++)
    for (int j = 0; j < 3; j++) {
      const int e = 3 * k + j;
      for (int a = 0; a < 3; a++) CE[(3 * k + a) * p + e] = Jl[k][3 * a + j];
      CE[(nf + e) * p + e] = 1.0;
      ce0[e] = -gen[6 + 3 * legs[k] + j];
    }
  /* inequalities: per stance leg  [min force, 4 friction rows, 3 x (upper, lower) torque rows] */
  for (int k = 0; k < nS; k++) {
    const int l = legs[k];
    double nW[3], nb[3], t1[3], t2[3], nn;
    if (normals_world) memcpy(nW, normals_world + 3 * l, sizeof(nW));
    else mv3(Rm, ez, nW);
    mtv3(Rm, nW, nb);
    cross3(nb, yB, t1); nn = sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]); for (int i = 0; i < 3; i++) t1[i] /= nn;
    cross3(nb, t1, t2); nn = sqrt(t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]); for (int i = 0; i < 3; i++) t2[i] /= nn;
    const int c0 = 11 * k;
    for (int i = 0; i < 3; i++) {
      const int row = 3 * k + i;
      CI[row * m + c0 + 0] = nb[i];
      CI[row * m + c0 + 1] = prm->friction * nb[i] + t1[i];
      CI[row * m + c0 + 2] = prm->friction * nb[i] - t1[i];
      CI[row * m + c0 + 3] = prm->friction * nb[i] + t2[i];
      CI[row * m + c0 + 4] = prm->friction * nb[i] - t2[i];
    }
    ci0[c0] = -prm->min_normal_force;
    for (int t = 1; t < 5; t++) ci0[c0 + t] = 0.0;
    for (int j = 0; j < 3; j++) {
      CI[(nf + 3 * k + j) * m + c0 + 5 + 2 * j] = -1.0; ci0[c0 + 5 + 2 * j] = prm->torque_limit;     /* tau <= tau_max */
      CI[(nf + 3 * k + j) * m + c0 + 6 + 2 * j] = 1.0;  ci0[c0 + 6 + 2 * j] = prm->torque_limit;     /* tau >= -tau_max */
    }
  }
  const int status = oracle_solve_quadprog(n, p, m, G, g0, CE, ce0, CI, ci0, x, &fval, active, &nact, &iters);
  if (status != ORACLE_QP_OK) {
    for (int i = 0; i < 12; i++) { tau[i] = 0.0; grf[i] = 0.0; }
    return status;
  }
  for (int k = 0; k < nS; k++)
    for (int j = 0; j < 3; j++) {
      grf[3 * legs[k] + j] = x[3 * k + j];
      tau[3 * legs[k] + j] = x[nf + 3 * k + j];
    }
  return status;
}

int oracle_wb_step_batch(const oracle_wb_params *prm, int64_t batch, const double *q, const double *qd,
                         const double *base_quat, const double *base_linvel_world, const double *base_angvel_base,
                         const double *a_des, const double *qdd_des, const uint8_t *stance, const double *normals_world,
                         double *tau, double *grf, int32_t *status, int nthreads) {
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
  for (int64_t i = 0; i < batch; i++) {
    double g[12];
    const int st = oracle_wb_step(prm, q + 12 * i, qd + 12 * i, base_quat + 4 * i, base_linvel_world + 3 * i,
                                  base_angvel_base + 3 * i, a_des + 6 * i, qdd_des ? qdd_des + 12 * i : NULL, stance + 4 * i,
                                  normals_world ? normals_world + 12 * i : NULL, tau + 12 * i, grf ? grf + 12 * i : g);
    if (status) status[i] = st;
  }
  (void)nthreads;
  return 0;
}
