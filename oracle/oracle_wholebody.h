/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * Floating-base rigid-body dynamics of the whole 18-DoF robot (base + 4 x 3 joints) and the whole-body
 * force/torque QP built on them: SURVEY.md section 8 row f4, the part of BASELINE.json's north_star the
 * reference itself does NOT contain (no CompositeRigidBody / NonlinearEffects /
 * CalcPointJacobian call anywhere; RBDL appears only for one swing leg, model_test_header.cpp:229-244,460).
 *
 * PARITY UNPINNED: there is no reference code, test or golden vector for this row.  The algorithms are
 * restated from their published form -- Featherstone, "Rigid Body Dynamics Algorithms" (2008): spatial
 * vectors and Pluecker transforms (ch. 2), recursive Newton-Euler in link coordinates (Table 5.1) with the
 * floating-base treatment of section 9.4, composite-rigid-body algorithm for a floating base (Table 9.6);
 * RBDL (absent third-party dependency, no version pinned by the reference: README.md:5-12) implements the
 * same recursions.  The tests pin this file on what the rest of the oracle already pins or can check
 * independently: contact Jacobians and gravity terms against oracle_model.c (rows a10-a12), M nu' + h against
 * the independent inverse-dynamics recursion below, kinetic energy against plain 3-D kinematics, and
 * conservation of total energy along an integrated free-flight trajectory.
 *
 * Model: quadruped_model/urdf/quadruped_model.urdf (include/qlamd_robot_constants.h): base_link inertial +
 * per leg three revolute-z links and the fixed foot link (rigidly attached to link 3).
 *
 * Generalised velocity  nu = [ v (3) ; w (3) ; qd (12) ]:  v = velocity of the base origin and w = angular
 * velocity of the base, both in BASE coordinates; joints in the order LF, RF, RH, LH (quadruped_state.cpp:340-343).
 * nu' = d(nu)/dt componentwise, i.e. the base part is the spatial acceleration in base coordinates
 * (classical acceleration of the origin in base coordinates = v' + w x v).  Equation of motion:
 *     M(q) nu' + h(q, nu) = [0 ; tau] + Jc' f,
 * f (12) = forces the ground applies on the four feet (base coordinates), Jc (12 x 18) the Jacobian of the four
 * foot-frame origins: rows 3l..3l+2 = [ 1 , -[r_l]x , 0 .. J_leg(l) .. 0 ].  Gravity g_W = (0, 0, -gravity).
 */
#ifndef ORACLE_WHOLEBODY_H
#define ORACLE_WHOLEBODY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* M: 18 x 18 row-major, symmetric positive definite. */
void oracle_wb_mass_matrix(const double q[12], double M[324]);

/* Inverse dynamics: generalised forces  M nu' + h  for the given motion (independent recursion, no M). */
void oracle_wb_inverse_dynamics(const double q[12], const double base_quat[4], const double nu[18],
                                const double nudot[18], double gravity, double out[18]);

/* h = inverse dynamics at nu' = 0 (Coriolis, centrifugal and gravity terms). */
void oracle_wb_nonlinear_effects(const double q[12], const double base_quat[4], const double nu[18], double gravity,
                                 double h[18]);

/* Jc: 12 x 18 row-major (from oracle_model.c's FK and Jacobian, rows a10/a11). */
void oracle_wb_contact_jacobian(const double q[12], double Jc[216]);

/* Energies by plain 3-D kinematics of every link (no spatial algebra): self-check routes only. */
double oracle_wb_kinetic_energy(const double q[12], const double nu[18]);
double oracle_wb_potential_energy(const double q[12], const double base_pos[3], const double base_quat[4], double gravity);

/* Whole-body QP (the ~24-variable problem of the north_star, stated over x = [f ; tau] of the stance legs):
 *     min (A f - b)' S (A f - b) + w_reg |f|^2 + w_tau |tau|^2
 *     s.t. tau + J_leg' f = tau0                    (joint rows of the equation of motion, 3 per stance leg)
 *          n.f >= f_min,  (mu n +- t1).f >= 0,  (mu n +- t2).f >= 0      (ContactForceDistribution.cpp:210-336)
 *          -tau_max <= tau <= tau_max
 *   A = base rows of Jc' on the stance legs (= the reference's A, ContactForceDistribution.cpp:168-206),
 *   b = (M nu'_des + h) base rows,  tau0 = (M nu'_des + h) joint rows,  nu'_des = [a_des ; qdd_des].
 * Solved with oracle_quadprog.c (pinned on the reference's QuadProg++) in this 6 nS-variable form with its
 * 3 nS equalities.  Swing-leg joints get tau0 (pure inverse dynamics).  Returns the ORACLE_QP_* status. */
typedef struct {
  double force_weights[6];   /* S */
  double regularizer;        /* w_reg */
  double torque_weight;      /* w_tau */
  double friction;           /* mu */
  double min_normal_force;   /* f_min */
  double torque_limit;       /* tau_max */
  double gravity;
} oracle_wb_params;

void oracle_wb_default_params(oracle_wb_params *p);

int oracle_wb_step(const oracle_wb_params *prm, const double q[12], const double qd[12], const double base_quat[4],
                   const double base_linvel_world[3], const double base_angvel_base[3], const double a_des[6],
                   const double *qdd_des /*[12] or NULL*/, const uint8_t stance[4],
                   const double *normals_world /*[4][3] or NULL*/, double tau[12], double grf[12]);

int oracle_wb_step_batch(const oracle_wb_params *prm, int64_t batch, const double *q, const double *qd,
                         const double *base_quat, const double *base_linvel_world, const double *base_angvel_base,
                         const double *a_des, const double *qdd_des, const uint8_t *stance, const double *normals_world,
                         double *tau, double *grf, int32_t *status, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
