// free_gait::State -- the data carrier between RosBalanceController, VirtualModelController and
// ContactForceDistribution.  Mirrors the members the hot path touches
//   quadruped_model/include/quadruped_model/quadruped_state.h:28-91 (getters / setters),
//   free_gait_core/src/executor/State.cpp:59-106,218-241 (support legs, surface normals, efforts)
// WITHOUT the reference's process-global statics: every State is its own robot.
#pragma once

#include "qlamd/types.hpp"

namespace free_gait {

using qlamd::Force;
using qlamd::LimbEnum;
using qlamd::LinearVelocity;
using qlamd::LocalAngularVelocity;
using qlamd::Pose;
using qlamd::Position;
using qlamd::RotationQuaternion;
using qlamd::Vector;

class State {
 public:
  // measured (feedback) state, ros_balance_controller.cpp:390-410
  void setCurrentLimbJoints(const std::array<double, 12> &q) { joints_ = q; }
  void setPoseBaseToWorld(const Pose &pose) { pose_ = pose; }
  void setBaseStateFromFeedback(const LinearVelocity &v, const LocalAngularVelocity &w) { lin_ = v; ang_ = w; }
  // target state, ros_balance_controller.cpp:384-387
  void setPositionWorldToBaseInWorldFrame(const Position &p) { target_.position = p; }
  void setOrientationBaseToWorld(const RotationQuaternion &q) { target_.rotation = q; }
  void setLinearVelocityBaseInWorldFrame(const LinearVelocity &v) { target_lin_ = v; }
  void setAngularVelocityBaseInBaseFrame(const LocalAngularVelocity &w) { target_ang_ = w; }
  // contact information, ros_balance_controller.cpp:245-378
  void setSupportLeg(LimbEnum limb, bool support) { support_[static_cast<int>(limb)] = support ? 1 : 0; }
  bool isSupportLeg(LimbEnum limb) const { return support_[static_cast<int>(limb)] != 0; }
  void setSurfaceNormal(LimbEnum limb, const Vector &n) {
    for (int i = 0; i < 3; ++i) normals_[3 * static_cast<int>(limb) + i] = n(i);
    has_normals_ = true;
  }
  void clearSurfaceNormals() { has_normals_ = false; } // == the override of ros_balance_controller.cpp:378

  const std::array<double, 12> &getJointPositionFeedback() const { return joints_; }
  const Pose &getPoseBaseToWorld() const { return pose_; }
  const Pose &getTargetPoseBaseToWorld() const { return target_; }
  const LinearVelocity &getLinearVelocityBaseInWorldFrame() const { return lin_; }
  const LocalAngularVelocity &getAngularVelocityBaseInBaseFrame() const { return ang_; }
  const LinearVelocity &getTargetLinearVelocityBaseInWorldFrame() const { return target_lin_; }
  const LocalAngularVelocity &getTargetAngularVelocityBaseInBaseFrame() const { return target_ang_; }
  const uint8_t *supportLegs() const { return support_.data(); }
  const double *surfaceNormals() const { return has_normals_ ? normals_.data() : nullptr; }

  // results written by the force distribution (State::setJointEffortsForLimb / getAllJointEfforts)
  void setAllJointEfforts(const std::array<double, 12> &tau) { efforts_ = tau; }
  const std::array<double, 12> &getAllJointEfforts() const { return efforts_; }

 private:
  std::array<double, 12> joints_{}, efforts_{}, normals_{};
  std::array<uint8_t, 4> support_{{1, 1, 1, 1}};
  bool has_normals_ = false;
  Pose pose_, target_;
  LinearVelocity lin_, target_lin_;
  LocalAngularVelocity ang_, target_ang_;
};

} // namespace free_gait
