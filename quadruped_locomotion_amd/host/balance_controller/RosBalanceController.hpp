// A ROS-free stand-in for balance_controller::RosBalanceController::{init,update}
// (balance_controller/src/ros_controller/ros_balance_controller.cpp:68-192,198-466), driven by a
// struct shaped like hardware_interface::RobotStateHandle::Data (robot_state_interface.hpp:28-65):
// raw pointers owned by the hardware interface, nothing allocated or freed across the boundary.
#pragma once

#include "balance_controller/VirtualModelController.hpp"

namespace balance_controller {

struct RobotStateHandleData { // robot_state_interface.hpp:28-65
  const double *orientation = nullptr;        // [4] (w,x,y,z)
  const double *position = nullptr;           // [3]
  const double *angular_velocity = nullptr;   // [3]
  const double *linear_velocity = nullptr;    // [3]
  const double *joint_position_read = nullptr;// [12]
  double *joint_effort_write = nullptr;       // [12]
  const bool *foot_contact = nullptr;         // [4]
};

struct BaseCommand { // what baseCommandCallback stores, ros_balance_controller.cpp:761-1083
  qlamd::Position position;
  qlamd::RotationQuaternion orientation;
  qlamd::LinearVelocity linear_velocity;
  qlamd::LocalAngularVelocity angular_velocity;
  bool support[4] = {true, true, true, true};
};

class RosBalanceController {
 public:
  // init: false aborts the controller load (ros_balance_controller.cpp:98-141)
  bool init(const RobotStateHandleData &hw, const qlamd_balance_params &params, int device = 0) {
    hw_ = hw;
    try {
      ctx_ = std::make_shared<qlamd::Context>(params, device);
    } catch (const std::exception &) {
      return false;
    }
    robot_state_ = std::make_shared<free_gait::State>();
    contact_distribution_ = std::make_shared<ContactForceDistribution>(ctx_, robot_state_);
    virtual_model_controller_ = std::make_shared<VirtualModelController>(ctx_, robot_state_, contact_distribution_);
    return contact_distribution_->loadParameters() && virtual_model_controller_->loadParameters();
  }

  void setCommand(const BaseCommand &cmd) { cmd_ = cmd; }

  // update: failures keep the previous efforts, as the reference does (:418-424)
  bool update() {
    std::array<double, 12> q{};
    for (int i = 0; i < 12; ++i) q[i] = hw_.joint_position_read[i];                 // :208-213
    for (int l = 0; l < 4; ++l) robot_state_->setSupportLeg(static_cast<qlamd::LimbEnum>(l), cmd_.support[l]);
    robot_state_->clearSurfaceNormals();                                             // :378
    robot_state_->setPositionWorldToBaseInWorldFrame(cmd_.position);                 // :384-387
    robot_state_->setOrientationBaseToWorld(cmd_.orientation);
    robot_state_->setLinearVelocityBaseInWorldFrame(cmd_.linear_velocity);
    robot_state_->setAngularVelocityBaseInBaseFrame(cmd_.angular_velocity);
    robot_state_->setCurrentLimbJoints(q);                                           // :390-410
    robot_state_->setPoseBaseToWorld(qlamd::Pose(
        qlamd::Position(hw_.position[0], hw_.position[1], hw_.position[2]),
        qlamd::RotationQuaternion(hw_.orientation[0], hw_.orientation[1], hw_.orientation[2], hw_.orientation[3])));
    robot_state_->setBaseStateFromFeedback(
        qlamd::LinearVelocity(hw_.linear_velocity[0], hw_.linear_velocity[1], hw_.linear_velocity[2]),
        qlamd::LocalAngularVelocity(hw_.angular_velocity[0], hw_.angular_velocity[1], hw_.angular_velocity[2]));
    if (!virtual_model_controller_->compute()) return false;                         // :419-424
    for (int i = 0; i < 12; ++i) hw_.joint_effort_write[i] = robot_state_->getAllJointEfforts()[i]; // :441-454
    return true;
  }

  const VirtualModelController &vmc() const { return *virtual_model_controller_; }

 private:
  RobotStateHandleData hw_;
  BaseCommand cmd_;
  std::shared_ptr<qlamd::Context> ctx_;
  std::shared_ptr<free_gait::State> robot_state_;
  std::shared_ptr<ContactForceDistribution> contact_distribution_;
  std::shared_ptr<VirtualModelController> virtual_model_controller_;
};

} // namespace balance_controller
