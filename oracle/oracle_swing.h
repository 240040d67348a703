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

/* The same recursion on any 3-joint chain (+ fixed end link), segment k given as the URDF gives it.  This is the
 * entry that PINS the recursion: on the 3-link model of MyRobotSolver::model_initialization
 * (model_test_header.cpp:183-222) it reproduces the 10 001 torques RBDL's InverseDynamics wrote into
 * single_leg_test/DataFloder/TauofInversedynamics.txt (IDynamicsCalculation, :277-301) to their print precision
 * (tests/golden/rbdl_leg_id.npz, tests/test_rbdl_pin.py). */
void oracle_chain_rnea(const double joint_xyz[4][3], const double joint_rpy[4][3], const double link_mass[4],
                       const double link_com[4][3], const double link_inertia[4][6], const double q[3],
                       const double qd[3], const double qdd[3], const double g[3], double tau[3]);

/* rows of (q, qd, qdd) [rows][3] -> tau [rows][3] */
void oracle_chain_rnea_rows(long rows, const double joint_xyz[4][3], const double joint_rpy[4][3],
                            const double link_mass[4], const double link_com[4][3], const double link_inertia[4][6],
                            const double *q, const double *qd, const double *qdd, const double g[3], double *tau);

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

/* ---- the swing branch of RosBalanceController::update (ros_balance_controller.cpp:467-603,720-756) -----------
 * For one NON-support leg:
 *   gravity_in_base = base_orientation.rotate((0,0,-9.8))                            :471   (rotate, as written)
 *   G   = getGravityCompensationForLimb(limb, q_leg, gravity_in_base)                :476-478
 *   pid = computeTorqueFromPositionCommand(command, joint, period)                   :484, 720-756
 *         command clamped to the URDF limits (enforceJointLimits :1152-1166), error from
 *         angles::shortest_angular_distance_with_limits (= command - position while both lie inside limits
 *         narrower than 2 pi), control_toolbox::Pid::computeCommand(error, dt)
 *   effort = swing-leg torque (a18) if the leg mode is "cartesian" or "footstep"     :485-486
 *          = pid + G               if not "leg_mode" (i.e. "joint" or never set)      :487-488
 *          = G                     if "leg_mode"                                      :489-490
 * control_toolbox and angles are third-party, un-vendored: PARITY UNPINNED.  Pid::computeCommand restated from
 * control_toolbox 1.1x: error_dot = (e - e_last)/dt; i_error += dt*e (clamped to i_min/i..i_max/i with
 * antiwindup); i_term = i*i_error (clamped to [i_min,i_max] without antiwindup); cmd = p*e + i_term + d*error_dot;
 * 0 when dt == 0 or the error is not finite.  With the shipped gains {p,i,d} and no i_clamp, i_max = i_min = 0 and
 * the integral term is clamped to zero. */
typedef struct {
  double p[12], i[12], d[12], i_max[12], i_min[12];
  int antiwindup;
  double lower[12], upper[12];   /* joint limits, quadruped_model.urdf (+-3.0) */
} oracle_pid_params;

void oracle_pid_default_params(oracle_pid_params *p);

/* leg_mode: 0 other / never set, 1 "joint", 2 "leg_mode", 3 "cartesian", 4 "footstep".
 * pid_error_last[3], pid_error_integral[3]: controller state of the leg's joints, updated. */
void oracle_swing_branch_leg(const oracle_swing_params *sp, const oracle_pid_params *pid, int leg, int leg_mode,
                             const double base_quat[4], const double q_id[3], const double q[3], const double qd[3],
                             const double qd_oldest[3], const double target_pos[3], const double target_vel[3],
                             const double joint_command[3], double period, double pid_error_last[3],
                             double pid_error_integral[3], double effort[3]);

#ifdef __cplusplus
}
#endif
#endif
