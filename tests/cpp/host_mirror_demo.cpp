// Drives the C++ mirror of the reference interface (quadruped_locomotion_amd/host/) end to end:
//   1. the stance scenario of balance_controller/test/test.cpp:57-130 (yaw 0.5 rad, desired height
//      0.1 above the measured one), through RosBalanceController::{init,update} and through
//      VirtualModelController + ContactForceDistribution::computeForceDistribution(F_B, T_B);
//   2. the SquareUp case of free_gait_core/test/PoseOptimizationSQPTest.cpp:111-150 through
//      PoseOptimizationSQP::optimize;
//   3. the demo QP of qp_solver/src/main.cc:46-101 through QuadraticProblemSolver::minimize.
// Prints "key v0 v1 ..." lines that tests/test_cpp_mirror.py compares with the oracle.
// Exit code 3 when no GPU is present (init() returns false: there is no CPU fallback).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <vector>

#include "balance_controller/RosBalanceController.hpp"
#include "balance_controller/WholeBodyController.hpp"
#include "free_gait_core/PoseConstraintsChecker.hpp"
#include "free_gait_core/PoseOptimizationGeometric.hpp"
#include "free_gait_core/PoseOptimizationQP.hpp"
#include "free_gait_core/PoseOptimizationSQP.hpp"
#include "ooqp_eigen_interface/QuadraticProblemFormulation.hpp"
#include "qp_solver/quadraticproblemsolver.hpp"

int main(int argc, char **argv) {
  qlamd_balance_params params;
  qlamd_balance_default_params(&params);

  // ---- 1. balance controller ------------------------------------------------------------------
  double q[12], effort[12] = {0};
  for (int l = 0; l < 4; ++l) { q[3 * l] = 0.05 * (l - 1.5); q[3 * l + 1] = 0.75; q[3 * l + 2] = -1.5; }
  const double yaw = 0.5;
  const double orientation[4] = {std::cos(yaw / 2), 0.0, 0.0, std::sin(yaw / 2)};
  const double position[3] = {0.0, 0.0, 0.2}, linvel[3] = {0.01, -0.02, 0.0}, angvel[3] = {0.0, 0.01, 0.02};
  bool contact[4] = {true, true, true, true};
  balance_controller::RobotStateHandleData hw;
  hw.orientation = orientation; hw.position = position; hw.linear_velocity = linvel; hw.angular_velocity = angvel;
  hw.joint_position_read = q; hw.joint_effort_write = effort; hw.foot_contact = contact;

  balance_controller::RosBalanceController controller;
  if (!controller.init(hw, params, 0)) { std::printf("init_failed 1\n"); return 3; }
  balance_controller::BaseCommand cmd;
  cmd.position = qlamd::Position(0.0, 0.0, 0.3);                      // 0.1 above the measured height
  cmd.orientation = qlamd::RotationQuaternion(orientation[0], 0, 0, orientation[3]);
  controller.setCommand(cmd);
  if (!controller.update()) { std::printf("update_failed 1\n"); return 4; }
  std::printf("state"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", q[i]); std::printf("\n");
  std::printf("effort"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", effort[i]); std::printf("\n");
  std::printf("grf"); for (double v : controller.vmc().getContactForces()) std::printf(" %.17g", v); std::printf("\n");

  // the inner boundary: computeForceDistribution(F_B, T_B) with a hand-made wrench
  auto ctx = std::make_shared<qlamd::Context>(params, 0);
  auto state = std::make_shared<free_gait::State>();
  std::array<double, 12> qa; for (int i = 0; i < 12; ++i) qa[i] = q[i];
  state->setCurrentLimbJoints(qa);
  state->setPoseBaseToWorld(qlamd::Pose(qlamd::Position(0, 0, 0.2), qlamd::RotationQuaternion(orientation[0], 0, 0, orientation[3])));
  balance_controller::ContactForceDistribution cfd(ctx, state);
  if (cfd.computeForceDistribution(qlamd::Force(0, 0, 500), qlamd::Torque(0, 0, 0))) { std::printf("unloaded_accepted 1\n"); return 5; }
  cfd.loadParameters();
  if (!cfd.computeForceDistribution(qlamd::Force(120.0, -40.0, 520.0), qlamd::Torque(10.0, -20.0, 5.0))) return 6;
  std::printf("cfd_effort"); for (double v : state->getAllJointEfforts()) std::printf(" %.17g", v); std::printf("\n");
  qlamd::Force f_lf; cfd.getForceForLeg(qlamd::LimbEnum::LF_LEG, f_lf);
  std::printf("cfd_force_lf %.17g %.17g %.17g\n", f_lf(0), f_lf(1), f_lf(2));

  // ---- 1c. the OOQP seam: the reference's own assembly (ContactForceDistribution.cpp:168-336) and its literal two-pass
  // sequence (addDesiredLegLoadConstraints :364-381: solve with 3 nS zero equality rows, then C = I, c = x1; :490 solve
  // again) through ooqpei::QuadraticProblemFormulation::solve, on the wrench just given to computeForceDistribution
  {
    ooqpei::QuadraticProblemFormulation::setContext(ctx);
    const int nS = 4, n = 3 * nS;
    double foot[12];
    if (qlamd_leg_kinematics_batch(ctx->get(), q, orientation, 1, foot, nullptr, nullptr, QLAMD_MEM_HOST, nullptr) != QLAMD_OK) return 22;
    ooqpei::Matrix A_(6, n), C_(n, n), D_(5 * nS, n);
    ooqpei::Vector S_(params.force_weights, params.force_weights + 6), b_{120.0, -40.0, 520.0, 10.0, -20.0, 5.0};
    ooqpei::Vector W_(n, params.regularizer), c_(n, 0.0), d_(5 * nS, 0.0), f_(5 * nS, std::numeric_limits<double>::max()), x1, x2;
    for (int l = 0; l < nS; ++l) {
      const double *r = foot + 3 * l;
      for (int i = 0; i < 3; ++i) A_(i, 3 * l + i) = 1.0;
      A_(3, 3 * l + 1) = -r[2]; A_(3, 3 * l + 2) = r[1];               // kindr::getSkewMatrixFromVector(r), :197
      A_(4, 3 * l + 0) = r[2];  A_(4, 3 * l + 2) = -r[0];
      A_(5, 3 * l + 0) = -r[1]; A_(5, 3 * l + 1) = r[0];
      // update() forces the surface normal to q.rotate(z) (ros_balance_controller.cpp:378): n_B = z.  The first tangential is
      // n_B x (world y in the base frame) (:301-305); with the yaw-only attitude of this scenario world y reads
      // (sin yaw, cos yaw, 0) in the base, so t1 = (-cos yaw, sin yaw, 0) and t2 = n_B x t1 = (-sin yaw, -cos yaw, 0)
      const double nb[3] = {0, 0, 1}, t1[3] = {-std::cos(yaw), std::sin(yaw), 0}, t2[3] = {-std::sin(yaw), -std::cos(yaw), 0};
      const double mu = params.friction;
      for (int i = 0; i < 3; ++i) {
        D_(l, 3 * l + i) = nb[i];
        D_(nS + 4 * l + 0, 3 * l + i) = mu * nb[i] + t1[i];
        D_(nS + 4 * l + 1, 3 * l + i) = mu * nb[i] - t1[i];
        D_(nS + 4 * l + 2, 3 * l + i) = mu * nb[i] + t2[i];
        D_(nS + 4 * l + 3, 3 * l + i) = mu * nb[i] - t2[i];
      }
      d_[l] = params.min_normal_force;
    }
    if (!ooqpei::QuadraticProblemFormulation::solve(A_, S_, b_, W_, C_, c_, D_, d_, f_, x1)) return 23;   // C = 0, c = 0
    for (int i = 0; i < n; ++i) { C_(i, i) = 1.0; c_[i] = x1[i]; }
    if (!ooqpei::QuadraticProblemFormulation::solve(A_, S_, b_, W_, C_, c_, D_, d_, f_, x2)) return 24;   // C = I, c = x1
    std::printf("ooqpei_x1"); for (double v : x1) std::printf(" %.17g", v); std::printf("\n");
    std::printf("ooqpei_x2"); for (double v : x2) std::printf(" %.17g", v); std::printf("\n");
    std::printf("cfd_grf");
    for (int l = 0; l < 4; ++l) { qlamd::Force fl; cfd.getForceForLeg(static_cast<qlamd::LimbEnum>(l), fl); for (int i = 0; i < 3; ++i) std::printf(" %.17g", -fl(i)); }
    std::printf("\n");
    // an inconsistent pair of equality rows is refused (OOQP reports failure; solve() returns false)
    ooqpei::Matrix Cbad(2, n); Cbad(0, 0) = 1.0; Cbad(1, 0) = 1.0;
    ooqpei::Vector xb;
    if (ooqpei::QuadraticProblemFormulation::solve(A_, S_, b_, W_, Cbad, {1.0, 2.0}, D_, d_, f_, xb)) return 25;
    std::printf("ooqpei_inconsistent_status %d\n", (int)ooqpei::QuadraticProblemFormulation::lastStatus());
  }

  // ---- 1a. the whole-body controller on the same stance scenario (an extension, no reference counterpart)
  {
    state->setBaseStateFromFeedback(qlamd::LinearVelocity(linvel[0], linvel[1], linvel[2]),
                                    qlamd::LocalAngularVelocity(angvel[0], angvel[1], angvel[2]));
    balance_controller::WholeBodyController wbc(ctx, state);
    if (wbc.compute()) return 17;                                      // parameters not loaded
    wbc.loadParameters();
    std::array<double, 12> wqd; for (int i = 0; i < 12; ++i) wqd[i] = 0.2 * std::cos(0.7 * i);
    wbc.setJointVelocities(wqd);
    wbc.setDesiredBaseAcceleration({0.5, -0.3, 0.8, 0.2, -0.1, 0.4});
    wbc.setTorqueLimit(60.0);
    if (!wbc.compute()) return 18;
    std::printf("wbc_qd"); for (double v : wqd) std::printf(" %.17g", v); std::printf("\n");
    std::printf("wbc_effort"); for (double v : state->getAllJointEfforts()) std::printf(" %.17g", v); std::printf("\n");
    std::printf("wbc_grf"); for (double v : wbc.getContactForces()) std::printf(" %.17g", v); std::printf("\n");
  }

  // ---- 1b. the whole tick from a serialised /desired_robot_state message (argv[1], written by the test) -----
  if (argc > 1) {
    std::vector<uint8_t> msg;
    if (FILE *fp = std::fopen(argv[1], "rb")) {
      uint8_t buf[4096];
      size_t n;
      while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) msg.insert(msg.end(), buf, buf + n);
      std::fclose(fp);
    }
    double qd[12];
    for (int i = 0; i < 12; ++i) qd[i] = 0.1 * std::sin(1.0 + i);
    hw.joint_velocity_read = qd;
    balance_controller::RosBalanceController tick;
    if (!tick.init(hw, params, 0)) return 12;
    if (tick.baseCommandCallback(msg.data(), 17)) return 13;              // a truncated message is refused
    if (!tick.baseCommandCallback(msg.data(), msg.size())) return 14;
    const bool touching[4] = {false, true, true, true};
    tick.footContactsCallback(touching);
    if (!tick.updateFullTick(0.0025)) return 15;
    std::printf("tick_effort"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", effort[i]); std::printf("\n");
    std::printf("tick_leg_state"); for (int l = 0; l < 4; ++l) std::printf(" %d", tick.legStateCodes()[l]); std::printf("\n");
    std::printf("tick_qd"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", qd[i]); std::printf("\n");
    // wall-clock cost of one whole tick (message -> efforts; three host-buffer calls + the balance solve); the
    // reference's loop runs at 400 Hz, i.e. a 2500 us budget (balance_controller_manager.cpp:48)
    std::vector<double> us;
    for (int rep = 0; rep < 300; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      if (!tick.baseCommandCallback(msg.data(), msg.size()) || !tick.updateFullTick(0.0025)) return 16;
      us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(us.begin() + 50, us.end());
    std::printf("tick_latency_us %.1f %.1f\n", us[50 + 125], us[50 + 225]); // median and p90 after 50 warm-up ticks
    // the in-memory log of update() (:606-716): one record per tick, never capped (the reference's guard counts a vector
    // that is never filled); the mirror's own setLogLength() caps, starting() clears (:1142-1153)
    const auto &log = tick.log();
    std::printf("tick_log_size %zu\n", log.size());
    std::printf("tick_log0_contact_force"); for (double v : log.front().desired_contact_force) std::printf(" %.17g", v); std::printf("\n");
    std::printf("tick_log0_effort"); for (double v : log.front().joint_command) std::printf(" %.17g", v); std::printf("\n");
    std::printf("tick_log0_leg_state"); for (int l = 0; l < 4; ++l) std::printf(" %d", log.front().leg_state[l]); std::printf("\n");
    std::printf("tick_log0_phase"); for (double v : log.front().leg_phase) std::printf(" %.17g", v); std::printf("\n");
    tick.setLogLength(log.size() + 2);
    for (int rep = 0; rep < 5; ++rep) tick.updateFullTick(0.0025);
    std::printf("tick_log_capped %zu\n", tick.log().size());
    tick.setLogLength(0);
    for (int rep = 0; rep < 5; ++rep) tick.updateFullTick(0.0025);
    std::printf("tick_log_uncapped %zu\n", tick.log().size());
    tick.starting();
    std::printf("tick_log_after_starting %zu\n", tick.log().size());
    tick.updateFullTick(0.0025);
    tick.stopping();
    std::printf("tick_log_after_stopping %zu\n", tick.log().size());
    // the same first tick through the one-call entry on a fresh controller: identical efforts; then its latency
    double effort1[12] = {0};
    balance_controller::RobotStateHandleData hw1 = hw;
    hw1.joint_effort_write = effort1;
    balance_controller::RosBalanceController one;
    if (!one.init(hw1, params, 0)) return 19;
    one.footContactsCallback(touching);
    if (!one.tick(msg.data(), msg.size(), 0.0025)) return 20;
    std::printf("tick1_effort"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", effort1[i]); std::printf("\n");
    std::vector<double> us1;
    for (int rep = 0; rep < 300; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      if (!one.tick(msg.data(), msg.size(), 0.0025)) return 21;
      us1.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(us1.begin() + 50, us1.end());
    std::printf("tick1_latency_us %.1f %.1f\n", us1[50 + 125], us1[50 + 225]);
    // the same ticks with the working set kept from tick to tick (setWarmStart: the second tick starts from the first one's set)
    double effort2[12] = {0};
    balance_controller::RobotStateHandleData hw2 = hw;
    hw2.joint_effort_write = effort2;
    balance_controller::RosBalanceController warm;
    if (!warm.init(hw2, params, 0)) return 22;
    warm.setWarmStart(true);
    warm.footContactsCallback(touching);
    if (!warm.tick(msg.data(), msg.size(), 0.0025)) return 23;
    if (!warm.tick(msg.data(), msg.size(), 0.0025)) return 24;
    std::printf("tick_warm_effort"); for (int i = 0; i < 12; ++i) std::printf(" %.17g", effort2[i]); std::printf("\n");
    if (!one.tick(msg.data(), msg.size(), 0.0025)) return 25; // (its PID state has moved on: compare the support legs)
  }

  // ---- 2. pose optimisation ---------------------------------------------------------------------
  free_gait::PoseOptimizationSQP optimization(ctx);
  free_gait::Stance nominal, stance;
  const double sx[4] = {0.3, 0.3, -0.3, -0.3}, sy[4] = {0.2, -0.2, 0.2, -0.2};
  const qlamd::LimbEnum order[4] = {qlamd::LimbEnum::LF_LEG, qlamd::LimbEnum::RF_LEG, qlamd::LimbEnum::LH_LEG, qlamd::LimbEnum::RH_LEG};
  for (int k = 0; k < 4; ++k) { nominal[order[k]] = qlamd::Position(sx[k], sy[k], -0.4); stance[order[k]] = qlamd::Position(sx[k], sy[k], -0.1); }
  optimization.setNominalStance(nominal);
  optimization.setStance(stance);
  optimization.setSupportStance(stance);
  optimization.setSupportRegion({{0.3, 0.2}, {-0.3, 0.2}, {-0.3, -0.2}, {0.3, -0.2}}); // counter-clockwise
  optimization.setLimbLengthConstraints({0.2, 0.2, 0.2, 0.2}, {0.565, 0.565, 0.565, 0.565});
  qlamd::Pose result(qlamd::Position(0.0, 0.0, 0.3), qlamd::RotationQuaternion());
  if (!optimization.optimize(result)) return 7;
  std::printf("pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", result.position(0), result.position(1), result.position(2),
              result.rotation.q[0], result.rotation.q[1], result.rotation.q[2], result.rotation.q[3]);

  // BaseAuto::optimizePose order (BaseAuto.cpp:394-400): QP, then the checker, then the SQP.
  // quadrupedSymmetricUnconstrained of PoseOptimizationQpTest.cpp:20-52 with the same stances.
  free_gait::PoseOptimizationQP qp(ctx);
  qp.setNominalStance(nominal); qp.setStance(stance); qp.setSupportStance(stance);
  qp.setSupportRegion({{0.3, 0.2}, {-0.3, 0.2}, {-0.3, -0.2}, {0.3, -0.2}});
  qlamd::Pose qp_result;
  if (!qp.optimize(qp_result)) return 10;
  std::printf("pose_qp %.17g %.17g %.17g\n", qp_result.position(0), qp_result.position(1), qp_result.position(2));
  free_gait::PoseConstraintsChecker checker(ctx);
  checker.setNominalStance(nominal); checker.setStance(stance); checker.setSupportStance(stance);
  checker.setSupportRegion({{0.3, 0.2}, {-0.3, 0.2}, {-0.3, -0.2}, {0.3, -0.2}});
  checker.setLimbLengthConstraints({0.2, 0.2, 0.2, 0.2}, {0.565, 0.565, 0.565, 0.565});
  checker.setTolerances(0.02, 0.0);                                    // BaseAuto.cpp:156
  const bool ok_in = checker.check(qp_result);
  qlamd::Pose far(qlamd::Position(0.5, 0.0, 0.3), qlamd::RotationQuaternion());  // CoM outside the footprint
  const bool ok_out = checker.check(far);
  std::printf("pose_check %d %d\n", ok_in ? 1 : 0, ok_out ? 1 : 0);

  free_gait::PoseOptimizationGeometric geometric(ctx);
  geometric.setNominalStance(nominal); geometric.setStance(stance); geometric.setSupportStance(stance);
  geometric.setSupportRegion({{0.3, 0.2}, {-0.3, 0.2}, {-0.3, -0.2}, {0.3, -0.2}});
  geometric.setStanceForOrientation(stance);
  qlamd::Pose geo;
  if (!geometric.optimize(geo)) return 11;
  std::printf("pose_geometric %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", geo.position(0), geo.position(1), geo.position(2),
              geo.rotation.q[0], geo.rotation.q[1], geo.rotation.q[2], geo.rotation.q[3]);

  // ---- 3. the QuadProg++ demo problem -----------------------------------------------------------
  qp_solver::QuadraticObjectiveFunction cost;
  qp_solver::LinearFunctionConstraints cons;
  qp_solver::Matrix G(2, 2); G(0, 0) = 1; G(0, 1) = -1; G(1, 0) = -1; G(1, 1) = 2;
  cost.setGlobalHessian(G); cost.setLinearTerm({-2.0, -6.0});
  qp_solver::Matrix A(3, 2); // A x <= b  <=>  CI = -A'
  A(0, 0) = 1; A(0, 1) = 1; A(1, 0) = -1; A(1, 1) = 2; A(2, 0) = 2; A(2, 1) = 1;
  cons.setGlobalInequalityConstraintJacobian(A); cons.setInequalityConstraintMaxValues({2.0, 2.0, 3.0});
  qp_solver::QuadraticProblemSolver solver(ctx);
  qp_solver::Vector x;
  if (!solver.minimize(cost, cons, x)) return 8;
  std::printf("qp %.17g %.17g %.17g\n", x[0], x[1], solver.lastObjective());
  qp_solver::Matrix Aeq(2, 1); cons.setGlobalEqualityConstraintJacobian(Aeq); cons.setEqualityConstraintMaxValues({0.0});
  if (!solver.minimize(cost, cons, x)) return 9;
  std::printf("qp_dummy_eq %.17g %.17g %.17g\n", x[0], x[1], solver.lastObjective());
  return 0;
}
