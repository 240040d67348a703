/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * CPU restatement of the reference's swing-leg torque:
 *   MyRobotSolver::update  single_leg_test/lib/model_test_header.cpp:412-503
 *     qdd   = (qd_newest - qd_oldest of an 11-deep queue) / (10 * period)          :417-431
 *     tau   = RBDL InverseDynamics(leg model, q, qd, 0.5 * qdd)                     :460
 *           + J' (kp o (p_des - p) + kd o (v_des - J qd))                           :485-499
 *   leg models: quadruped_model/urdf/quadruped_model_{lf,rf,rh,lh}_leg.urdf (same constants as
 *   quadruped_model.urdf), fixed base, gravity (0,0,-9.81) in the model's base frame
 *   (model_test_header.cpp:229-244; RBDL urdfreader default).
 * RBDL is third-party, un-vendored, absent: PARITY UNPINNED.  The recursive Newton-Euler algorithm
 * is restated; tests check it against the gravity torque, the symmetry / definiteness of the
 * mass matrix it implies and a power balance along a simulated motion.
 * Quirk (reproduced on request): the reference feeds InverseDynamics the joint POSITIONS of the
 * limb that was stored last (LH), whatever limb is updated (model_test_header.cpp:106-112,460;
 * ros_balance_controller.cpp:239) -- pass that limb's q as q_id to mimic it.
 */
#ifndef ORACLE_SWING_H
#define ORACLE_SWING_H

#ifdef __cplusplus
extern "C" {
#endif

/* tau = M(q) qdd + C(q,qd) qd + G(q) for the 3-joint leg chain, gravity field g (base frame). */
void oracle_leg_rnea(int leg, const double q[3], const double qd[3], const double qdd[3], const double g[3],
                     double tau[3]);

typedef struct {
  double kp[3], kd[3];    /* 300 / 20 each, controller_gains.yaml:42-51 */
  double period;          /* control period (0.0025 s, balance_controller_manager.cpp:48) */
  double accel_window;    /* 10: Time_derta = period * 10, model_test_header.cpp:418 */
  double accel_scale;     /* 0.5, model_test_header.cpp:460 */
  double gravity;         /* 9.81 */
} oracle_swing_params;

void oracle_swing_default_params(oracle_swing_params *p);

/* One leg.  q_id: positions used for the inverse dynamics; q, qd: current joint state (FK, Jacobian,
 * foot velocity, newest queue entry); qd_oldest: oldest queue entry. */
void oracle_swing_leg_torque(const oracle_swing_params *p, int leg, const double q_id[3], const double q[3],
                             const double qd[3], const double qd_oldest[3], const double target_pos[3],
                             const double target_vel[3], double tau[3]);

#ifdef __cplusplus
}
#endif
#endif
