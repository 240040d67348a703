// A motion controller with the interface of balance_controller::MotionControllerBase
// (balance_controller/include/balance_controller/motion_control/MotionControllerBase.hpp:58-110: loadParameters(),
// compute(), results in free_gait::State) that uses the whole-body step of the C-ABI instead of the virtual-model
// wrench: inverse dynamics of the 18-DoF tree for a desired base acceleration -> force/torque QP with friction and
// torque-limit rows -> 12 joint efforts.  The reference has no such controller; in the plugin it
// would be constructed next to virtual_model_controller_ (ros_balance_controller.cpp:88-95) and called at :419.
#pragma once

#include "balance_controller/ContactForceDistribution.hpp"

namespace balance_controller {

class WholeBodyController {
 public:
  WholeBodyController(std::shared_ptr<qlamd::Context> ctx, std::shared_ptr<free_gait::State> robot_state)
      : ctx_(std::move(ctx)), robot_state_(std::move(robot_state)) {
    qlamd_wholebody_default_params(&params_);
  }

  bool loadParameters() { isParametersLoaded_ = (ctx_ != nullptr); return isParametersLoaded_; }
  void setTorqueLimit(double tau_max) { params_.torque_limit = tau_max; }
  void setTorqueWeight(double w) { params_.torque_weight = w; }

  // what free_gait::State does not carry: the measured joint velocities (RobotStateHandle::getJointVelocityRead,
  // robot_state_interface.hpp:28-65) and the acceleration the base should have, [v' ; w'] in base coordinates
  void setJointVelocities(const std::array<double, 12> &qd) { qd_ = qd; }
  void setDesiredBaseAcceleration(const std::array<double, 6> &a) { a_des_ = a; }

  // false when parameters are missing or the QP fails (infeasible torque / friction bounds): efforts are left as they are
  bool compute() {
    if (!isParametersLoaded_) return false;
    const auto &s = *robot_state_;
    qlamd_wholebody_batch in{};
    in.joint_position = s.getJointPositionFeedback().data();
    in.joint_velocity = qd_.data();
    in.base_orientation = s.getPoseBaseToWorld().getRotation().q;
    in.base_linear_velocity = s.getLinearVelocityBaseInWorldFrame().v;
    in.base_angular_velocity = s.getAngularVelocityBaseInBaseFrame().v;
    in.desired_base_acceleration = a_des_.data();
    in.desired_joint_acceleration = nullptr;
    in.support_leg = s.supportLegs();
    in.surface_normal = s.surfaceNormals();
    std::array<double, 12> tau{};
    int32_t status = -1;
    const int rc = qlamd_wholebody_solve_batch(ctx_->get(), &params_, &in, 1, tau.data(), grf_.data(), &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false;
    robot_state_->setAllJointEfforts(tau);
    return true;
  }

  const std::array<double, 12> &getContactForces() const { return grf_; }

 private:
  std::shared_ptr<qlamd::Context> ctx_;
  std::shared_ptr<free_gait::State> robot_state_;
  qlamd_wholebody_params params_;
  std::array<double, 12> qd_{}, grf_{};
  std::array<double, 6> a_des_{};
  bool isParametersLoaded_ = false;
};

} // namespace balance_controller
