/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * CPU restatement of one balance-controller control step of the reference:
 *   state marshalling     balance_controller/src/ros_controller/ros_balance_controller.cpp:198-466
 *   virtual-model wrench  balance_controller/src/motion_control/VirtualModelController.cpp:89-268
 *   force distribution    balance_controller/src/contact_force_distribution/ContactForceDistribution.cpp:99-596
 *   robot state/masses    quadruped_model/src/quadruped_state.cpp:26-45,83-101
 * The QP of ContactForceDistribution.cpp:385-514 is solved in the reference by
 * ooqpei/OOQP/MA27 (third-party, un-vendored, unpinned, absent here):
 * PARITY UNPINNED at that boundary.  The QP is strictly convex, so the target
 * is its unique minimiser, obtained here with the restatement of the
 * reference's own QuadProg++ (oracle_quadprog.c, which IS pinned against the
 * compiled reference solver).  kindr rotation conventions are restated from
 * upstream kindr 1.x, also unpinned.
 */
#ifndef ORACLE_BALANCE_H
#define ORACLE_BALANCE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  /* VirtualModelController gains, controller_gains.yaml:3-26 (x,y,z = heading,lateral,vertical / roll,pitch,yaw) */
  double kp_trans[3], kd_trans[3], kff_trans[3];
  double kp_rot[3], kd_rot[3], kff_rot[3];
  /* ContactForceDistribution weights/constraints, controller_gains.yaml:27-41 */
  double force_weights[6];     /* S = diag(...) */
  double regularizer;          /* W = regularizer * I */
  double friction;             /* mu */
  double min_normal_force;     /* f_min */
  /* ros_balance_controller.cpp:451-454 */
  double torque_limit;
  /* quadruped_state.cpp:28-41,83-97; VirtualModelController.cpp:54,165 */
  double torso_mass;
  double leg_mass[4];
  double gravity;              /* g_W = (0,0,-gravity) */
  double grav_comp_percentage;
  double com_in_base[3];
  double hip_in_base[4][3];
} oracle_balance_params;

void oracle_balance_default_params(oracle_balance_params *p);

/* One robot.  Quaternions are (w,x,y,z) base->world.  stance[leg]!=0 marks a
 * support leg.  normals_world may be NULL: then the reference's update()
 * override n_W = q_m.rotate(z) applies (ros_balance_controller.cpp:378).
 * Outputs: tau[12] clamped joint efforts (0 for non-stance legs), tau_raw[12]
 * before the clamp, grf[12] = QP solution x (ground reaction forces, base
 * frame, 0 for non-stance legs), wrench[6] = (F_B, T_B).  Any output pointer
 * may be NULL.  Returns the QP status (ORACLE_QP_*). */
int oracle_balance_step(const oracle_balance_params *prm,
                        const double q[12],
                        const double base_pos[3], const double base_quat[4],
                        const double base_linvel[3], const double base_angvel[3],
                        const double des_pos[3], const double des_quat[4],
                        const double des_linvel[3], const double des_angvel[3],
                        const uint8_t stance[4], const double *normals_world /*[4][3] or NULL*/,
                        double *tau, double *tau_raw, double *grf, double *wrench,
                        int *qp_iters, int *n_active);

/* Batch over robots, field arrays laid out [B][k].  nthreads<=1: serial. */
void oracle_balance_batch(const oracle_balance_params *prm, int64_t B,
                          const double *q, const double *base_pos, const double *base_quat,
                          const double *base_linvel, const double *base_angvel,
                          const double *des_pos, const double *des_quat,
                          const double *des_linvel, const double *des_angvel,
                          const uint8_t *stance, const double *normals_world,
                          double *tau, double *grf, int32_t *status, int nthreads);

/* Pieces exposed for unit tests. */
void oracle_quat_to_matrix(const double q[4], double R[9]);
void oracle_quat_box_minus(const double a[4], const double b[4], double out[3]);
void oracle_virtual_wrench(const oracle_balance_params *prm,
                           const double base_pos[3], const double base_quat[4],
                           const double base_linvel[3], const double base_angvel[3],
                           const double des_pos[3], const double des_quat[4],
                           const double des_linvel[3], const double des_angvel[3],
                           double wrench[6]);
/* Assemble the force QP in QuadProg++ form for nS stance legs (n = 3 nS,
 * m = 5 nS): G (n x n), g0 (n), CI (n x m), ci0 (m).  r_feet is [nS][3],
 * n_B/t1/t2 are [nS][3] per-leg pyramid directions in the base frame. */
void oracle_force_qp_assemble(const oracle_balance_params *prm, int nS,
                              const double *r_feet, const double wrench[6],
                              const double *n_B, const double *t1, const double *t2,
                              double *G, double *g0, double *CI, double *ci0);

/* The same problem as the reference states it for ooqpei::QuadraticProblemFormulation::solve
 * (ContactForceDistribution.cpp:168-336): A [6][n], S [6] and W [n] (diagonals), b [6], D [5 nS][n], d, f [5 nS] with
 * f = DBL_MAX ("no bound", :246,329). */
void oracle_force_lsq_assemble(const oracle_balance_params *prm, int nS, const double *r_feet, const double wrench[6],
                               const double *n_B, const double *t1, const double *t2, double *A, double *S, double *b,
                               double *W, double *D, double *d, double *f);

/* min (Ax - b)'S(Ax - b) + x'Wx  s.t. Cx = c, d <= Dx <= f: the contract of ooqpei's solve (call sites
 * ContactForceDistribution.cpp:367,490).  Equalities eliminated by a null-space basis, then the pinned Goldfarb-Idnani
 * restatement; +-DBL_MAX / inf bounds dropped.  n, k, p <= 12, m <= 24.  Returns ORACLE_QP_*. */
int oracle_weighted_lsq_qp(int n, int k, int p, int m, const double *A, const double *S, const double *b, const double *W,
                           const double *C, const double *c, const double *D, const double *d, const double *f, double *x);

#ifdef __cplusplus
}
#endif
/* bench.py cpu_baseline: `passes` passes inside one OpenMP region; returns the wall seconds. */
double oracle_balance_batch_repeat(const oracle_balance_params *prm, int64_t B,
                                   const double *q, const double *base_pos, const double *base_quat,
                                   const double *base_linvel, const double *base_angvel,
                                   const double *des_pos, const double *des_quat,
                                   const double *des_linvel, const double *des_angvel,
                                   const uint8_t *stance, double *tau, int32_t *status, int nthreads, int passes);

#endif
