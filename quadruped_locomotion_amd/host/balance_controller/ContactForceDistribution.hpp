// balance_controller::ContactForceDistribution on top of the C-ABI.
// Same names, argument meaning and error behaviour as
//   balance_controller/include/balance_controller/contact_force_distribution/ContactForceDistributionBase.hpp:97-98
//   balance_controller/src/contact_force_distribution/ContactForceDistribution.cpp:99-136 (computeForceDistribution),
//   :598-706 (getters), :818-886 (loadParameters).
#pragma once

#include "free_gait_core/State.hpp"

namespace balance_controller {

using qlamd::Force;
using qlamd::LimbEnum;
using qlamd::Torque;

class ContactForceDistribution {
 public:
  ContactForceDistribution(std::shared_ptr<qlamd::Context> ctx, std::shared_ptr<free_gait::State> robot_state)
      : ctx_(std::move(ctx)), robot_state_(std::move(robot_state)) {}

  // The parameters live in the context (they were given to qlamd_context_create); this marks them loaded,
  // as ContactForceDistribution::loadParameters does after reading the ROS parameter server.
  bool loadParameters() { isParametersLoaded_ = (ctx_ != nullptr); return isParametersLoaded_; }

  // false when parameters are not loaded (ContactForceDistribution.cpp:103) or the QP fails (:490-494).
  bool computeForceDistribution(const Force &virtualForceInBaseFrame, const Torque &virtualTorqueInBaseFrame) {
    isForceDistributionComputed_ = false;
    if (!isParametersLoaded_) return false;
    const double wrench[6] = {virtualForceInBaseFrame(0), virtualForceInBaseFrame(1), virtualForceInBaseFrame(2),
                              virtualTorqueInBaseFrame(0), virtualTorqueInBaseFrame(1), virtualTorqueInBaseFrame(2)};
    std::array<double, 12> tau{};
    int32_t status = -1;
    const int rc = qlamd_force_distribution_batch(ctx_->get(), robot_state_->getJointPositionFeedback().data(),
                                                  robot_state_->getPoseBaseToWorld().getRotation().q,
                                                  robot_state_->supportLegs(), robot_state_->surfaceNormals(), wrench, 1,
                                                  tau.data(), grf_.data(), &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false;
    robot_state_->setAllJointEfforts(tau); // computeJointTorques, :516-578 (plus the +-limit clamp)
    isForceDistributionComputed_ = true;
    return true;
  }

  // desiredContactForce_ = -x (ContactForceDistribution.cpp:502-503)
  bool getForceForLeg(LimbEnum leg, Force &force) const {
    if (!isForceDistributionComputed_) return false;
    for (int i = 0; i < 3; ++i) force(i) = -grf_[3 * static_cast<int>(leg) + i];
    return true;
  }
  bool isForceDistributionComputed() const { return isForceDistributionComputed_; }

 private:
  std::shared_ptr<qlamd::Context> ctx_;
  std::shared_ptr<free_gait::State> robot_state_;
  std::array<double, 12> grf_{};
  bool isParametersLoaded_ = false, isForceDistributionComputed_ = false;
};

} // namespace balance_controller
