// balance_controller::VirtualModelController on top of the C-ABI
// (balance_controller/include/balance_controller/motion_control/VirtualModelController.hpp:100,
//  balance_controller/src/motion_control/VirtualModelController.cpp:89-268).
#pragma once

#include "balance_controller/ContactForceDistribution.hpp"

namespace balance_controller {

class VirtualModelController {
 public:
  VirtualModelController(std::shared_ptr<qlamd::Context> ctx, std::shared_ptr<free_gait::State> robot_state,
                         std::shared_ptr<ContactForceDistribution> contactForceDistribution)
      : ctx_(std::move(ctx)), robot_state_(std::move(robot_state)),
        contactForceDistribution_(std::move(contactForceDistribution)) {}

  bool loadParameters() { isParametersLoaded_ = (ctx_ != nullptr); return isParametersLoaded_; }

  // computeError -> computeGravityCompensation -> computeVirtualForce/Torque -> computeForceDistribution;
  // one fused device step.  false when parameters are missing or the distribution fails (:91,:98-100).
  bool compute() {
    if (!isParametersLoaded_) return false;
    const auto &s = *robot_state_;
    qlamd_state_batch in;
    in.joint_position = s.getJointPositionFeedback().data();
    in.base_position = s.getPoseBaseToWorld().getPosition().v;
    in.base_orientation = s.getPoseBaseToWorld().getRotation().q;
    in.base_linear_velocity = s.getLinearVelocityBaseInWorldFrame().v;
    in.base_angular_velocity = s.getAngularVelocityBaseInBaseFrame().v;
    in.desired_position = s.getTargetPoseBaseToWorld().getPosition().v;
    in.desired_orientation = s.getTargetPoseBaseToWorld().getRotation().q;
    in.desired_linear_velocity = s.getTargetLinearVelocityBaseInWorldFrame().v;
    in.desired_angular_velocity = s.getTargetAngularVelocityBaseInBaseFrame().v;
    in.support_leg = s.supportLegs();
    in.surface_normal = s.surfaceNormals();
    std::array<double, 12> tau{};
    int32_t status = -1;
    const int rc = qlamd_balance_solve_batch(ctx_->get(), &in, 1, tau.data(), grf_.data(), &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false;
    robot_state_->setAllJointEfforts(tau);
    return true;
  }

  const std::array<double, 12> &getContactForces() const { return grf_; } // QP solution x (ground reaction forces)

 private:
  std::shared_ptr<qlamd::Context> ctx_;
  std::shared_ptr<free_gait::State> robot_state_;
  std::shared_ptr<ContactForceDistribution> contactForceDistribution_;
  std::array<double, 12> grf_{};
  bool isParametersLoaded_ = false;
};

} // namespace balance_controller
