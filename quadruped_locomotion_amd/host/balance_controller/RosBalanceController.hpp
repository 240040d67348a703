// A ROS-free stand-in for balance_controller::RosBalanceController::{init,update}
// (balance_controller/src/ros_controller/ros_balance_controller.cpp:68-192,198-466), driven by a
// struct shaped like hardware_interface::RobotStateHandle::Data (robot_state_interface.hpp:28-65):
// raw pointers owned by the hardware interface, nothing allocated or freed across the boundary.
#pragma once

#include <vector>

#include "balance_controller/VirtualModelController.hpp"

namespace balance_controller {

struct RobotStateHandleData { // robot_state_interface.hpp:28-65
  const double *orientation = nullptr;        // [4] (w,x,y,z)
  const double *position = nullptr;           // [3]
  const double *angular_velocity = nullptr;   // [3]
  const double *linear_velocity = nullptr;    // [3]
  const double *joint_position_read = nullptr;// [12]
  const double *joint_velocity_read = nullptr;// [12] (only the full tick, updateFullTick, reads it)
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

// One record of the plugin's in-memory log (ros_balance_controller.cpp:606-716: leg_states_, joint_command_, joint_actual_,
// foot_desired_contact_, leg_phases_, up to log_length_ = 10 000 ticks, :17; replayed on /log/* when /capture_log_data is
// called, :1173-1191).  The ROS message types are absent; the fields are the numbers those messages carry.
struct TickLogEntry {
  int8_t leg_state[4];                 // leg_state.data (the state codes of the four limbs)
  double joint_command[12];            // joint_command.effort: what joints[i].setCommand received
  double joint_actual[12];             // joint_actual.position
  double desired_contact_force[12];    // foot_desired_contact_: desiredContactForce_ (= -x) rotated to the WORLD frame (:656-661)
  uint8_t is_contact[4];               // real_contact_ (:670)
  double leg_phase[8];                 // leg_phase.data[2 i] = stance phase, [2 i + 1] = swing phase (:672-673)
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
    // a failed solve keeps the efforts of the tick before, as the reference's update() does (:418-424,441-454)
    if (qlamd_set_option(ctx_->get(), QLAMD_OPT_ON_FAILURE, QLAMD_ON_FAILURE_KEEP) != QLAMD_OK) return false;
    command_.assign(qlamd_tick_command_bytes(1), 0);
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

  // ---- the whole tick (SURVEY.md section 8 rows f1/f2 on top of a1) ----------------------------------------
  // baseCommandCallback on a serialised free_gait_msgs/RobotState (ros_balance_controller.cpp:761-1083)
  bool baseCommandCallback(const uint8_t *msg, size_t len) {
    const int64_t off[2] = {0, static_cast<int64_t>(len)};
    double pos[3], quat[4], lin[3], ang[3];
    uint8_t mode[4];
    qlamd_robot_state_fields f{};
    f.des_pos = pos; f.des_quat = quat; f.des_linvel = lin; f.des_angvel = ang;
    f.joint_command = joint_command_.data(); f.foot_position = foot_target_.data(); f.foot_velocity = foot_velocity_.data();
    f.phase = phase_; f.support_leg = support_leg_; f.leg_mode = mode;
    int32_t status = -1;
    if (qlamd_robot_state_unpack_batch(ctx_->get(), msg, off, 1, &f, &status, QLAMD_MEM_HOST, nullptr) != QLAMD_OK ||
        status != QLAMD_WIRE_OK)
      return false; // the reference would index past the end of the message's arrays here
    cmd_.position = qlamd::Position(pos[0], pos[1], pos[2]);
    cmd_.orientation = qlamd::RotationQuaternion(quat[0], quat[1], quat[2], quat[3]);
    cmd_.linear_velocity = qlamd::LinearVelocity(lin[0], lin[1], lin[2]);
    cmd_.angular_velocity = qlamd::LocalAngularVelocity(ang[0], ang[1], ang[2]);
    for (int l = 0; l < 4; ++l)
      if (mode[l] != QLAMD_LEG_MODE_OTHER) leg_mode_[l] = mode[l]; // an unknown name leaves the flags as they are (:876-964)
    return true;
  }
  // footContactsCallback (:1084-1135) only records the sensors; the transitions run inside updateFullTick
  void footContactsCallback(const bool contact[4]) {
    for (int l = 0; l < 4; ++l) contact_[l] = contact[l] ? 1 : 0;
  }
  // update() with its leg state machine (:234-380), the balance solve (:419-454) and the swing branch (:467-603):
  // all 12 effort commands (what joints[i].setCommand receives) go to joint_effort_write.
  bool updateFullTick(double period) {
    uint8_t is_footstep[4];
    for (int l = 0; l < 4; ++l) is_footstep[l] = leg_mode_[l] == QLAMD_LEG_MODE_FOOTSTEP;
    qlamd_leg_state_batch ls{support_leg_, phase_, is_footstep, contact_, hw_.joint_position_read, limb_state_, store_flag_,
                             stored_joint_position_.data(), joint_command_.data(), foot_target_.data(), support_, leg_state_code_};
    if (qlamd_leg_state_machine_batch(ctx_->get(), &ls, 1, 1, QLAMD_MEM_HOST, nullptr) != QLAMD_OK) return false;
    for (int l = 0; l < 4; ++l) cmd_.support[l] = support_[l] != 0;
    const bool solved = update();                      // stance legs: QP torques, clamped; others 0
    // joint velocity queue of MyRobotSolver (11 deep, model_test_header.cpp:417-431)
    for (int i = 0; i < 12; ++i) {
      for (int k = 10; k > 0; --k) qd_queue_[k][i] = qd_queue_[k - 1][i];
      qd_queue_[0][i] = hw_.joint_velocity_read ? hw_.joint_velocity_read[i] : 0.0;
    }
    qlamd_swing_params sp; qlamd_swing_default_params(&sp);
    sp.period = period;
    qlamd_joint_pid_params pid; qlamd_joint_pid_default_params(&pid);
    const qlamd_swing_batch sw{hw_.joint_position_read, qd_queue_[0].data(), qd_queue_[10].data(), foot_target_.data(),
                               foot_velocity_.data(), support_, nullptr};
    const qlamd_swing_branch_extra ex{hw_.orientation, joint_command_.data(), leg_mode_, pid_error_last_.data(),
                                      pid_error_integral_.data()};
    if (qlamd_swing_branch_batch(ctx_->get(), &sp, &pid, &sw, &ex, period, 1, hw_.joint_effort_write, QLAMD_MEM_HOST, nullptr) !=
        QLAMD_OK)
      return false;
    appendLog();
    return solved;
  }
  // The tail of update() (:606-716): one record per tick.  In the reference the guard in front of it,
  // `base_actual_pose_.size() < log_length_` (:607), counts a vector whose push_back is commented out (:645-646): the
  // guard always holds and leg_states_, joint_command_ ... grow with every tick until starting() clears them.  The mirror
  // does the same by default (no cap).  setLogLength(n) is an addition of the mirror, not reference behaviour: a cap for
  // long-running callers, 0 = none.
  const std::vector<TickLogEntry> &log() const { return log_; }
  void clearLog() { log_.clear(); }
  void setLogLength(size_t n) { log_length_ = n; }
  // controller_interface::Controller::starting / stopping (ros_balance_controller.cpp:1142-1156): starting() clears the
  // tick log (leg_states_, joint_actual_, joint_command_, foot_desired_contact_, leg_phases_, desired_robot_state_,
  // actual_robot_state_: the vectors one TickLogEntry holds), stopping() does nothing.  The ros::Time argument is dropped
  // with the rest of ROS; both are ignored by the reference.
  void starting() { log_.clear(); }
  void stopping() {}
  // The same tick through ONE call of the C-ABI (qlamd_full_tick_batch, batch 1): message bytes in, 12 efforts out.
  // Equivalent to baseCommandCallback(msg) + footContactsCallback + updateFullTick(period); a message that cannot be
  // deserialised leaves the command of the last good one in force, as in the reference where it never reaches the
  // callback.  What stays between ticks (command in force, limb states, stored joints, leg modes, PID errors, velocity
  // queue) lives here.
  bool tick(const uint8_t *msg, size_t len, double period) {
    for (int i = 0; i < 12; ++i) {
      for (int k = 10; k > 0; --k) qd_queue_[k][i] = qd_queue_[k - 1][i];
      qd_queue_[0][i] = hw_.joint_velocity_read ? hw_.joint_velocity_read[i] : 0.0;
    }
    const int64_t off[2] = {0, static_cast<int64_t>(len)};
    qlamd_swing_params sp; qlamd_swing_default_params(&sp);
    qlamd_joint_pid_params pid; qlamd_joint_pid_default_params(&pid);
    int32_t status = -1, message_status = -1;
    qlamd_tick_batch io{msg, off, hw_.joint_position_read, qd_queue_[0].data(), qd_queue_[10].data(), hw_.position,
                        hw_.orientation, hw_.linear_velocity, hw_.angular_velocity, contact_, limb_state_, store_flag_,
                        stored_joint_position_.data(), leg_mode_, support_, pid_error_last_.data(), pid_error_integral_.data(),
                        hw_.joint_effort_write, leg_state_code_, &status, &message_status, command_.data(),
                        warm_start_ ? &working_set_ : nullptr};
    if (qlamd_full_tick_batch(ctx_->get(), &sp, &pid, &io, period, 1, 1, QLAMD_MEM_HOST, nullptr) != QLAMD_OK) return false;
    return message_status == QLAMD_WIRE_OK && status == QLAMD_STATUS_OK;
  }
  // An addition of the mirror: the balance solve of tick() starts from the working set of the tick before (kept here like the
  // plugin's other state; qlamd_tick_batch::working_set) instead of from scratch, as OOQP does in the reference.  Efforts agree
  // to the solver's accuracy; off by default.
  void setWarmStart(bool on) { warm_start_ = on; working_set_ = 0u; }
  const int8_t *legStateCodes() const { return leg_state_code_; }
  const int8_t *limbStates() const { return limb_state_; }

  const VirtualModelController &vmc() const { return *virtual_model_controller_; }

 private:
  void appendLog() {
    if (log_length_ != 0 && log_.size() >= log_length_) return;                      // the mirror's own cap; :607 never caps
    TickLogEntry e{};
    const double w = hw_.orientation[0], x = hw_.orientation[1], y = hw_.orientation[2], z = hw_.orientation[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    const auto &grf = virtual_model_controller_->getContactForces();                 // x: ground reaction, base frame
    for (int l = 0; l < 4; ++l) {
      e.leg_state[l] = leg_state_code_[l];
      e.is_contact[l] = contact_[l];
      e.leg_phase[2 * l] = support_leg_[l] ? phase_[l] : 0.0;
      e.leg_phase[2 * l + 1] = support_leg_[l] ? 0.0 : phase_[l];
      for (int r = 0; r < 3; ++r)                                                    // orientation.rotate(-x_leg)
        e.desired_contact_force[3 * l + r] = -(R[3 * r] * grf[3 * l] + R[3 * r + 1] * grf[3 * l + 1] + R[3 * r + 2] * grf[3 * l + 2]);
    }
    for (int i = 0; i < 12; ++i) { e.joint_command[i] = hw_.joint_effort_write[i]; e.joint_actual[i] = hw_.joint_position_read[i]; }
    log_.push_back(e);
  }
  std::vector<TickLogEntry> log_;
  size_t log_length_ = 0;                                                            // 0 = no cap (the reference's log_length_, :17, caps nothing)
  RobotStateHandleData hw_;
  BaseCommand cmd_;
  std::shared_ptr<qlamd::Context> ctx_;
  std::shared_ptr<free_gait::State> robot_state_;
  std::shared_ptr<ContactForceDistribution> contact_distribution_;
  std::shared_ptr<VirtualModelController> virtual_model_controller_;
  // what the plugin keeps between ticks for the leg state machine and the swing branch
  std::array<double, 12> joint_command_{}, foot_target_{}, foot_velocity_{}, stored_joint_position_{}, pid_error_last_{},
      pid_error_integral_{};
  std::array<std::array<double, 12>, 11> qd_queue_{};
  std::vector<uint8_t> command_; // opaque command block of qlamd_full_tick_batch (batch 1)
  uint32_t working_set_ = 0u;    // final working set of the last tick's force QP (setWarmStart)
  bool warm_start_ = false;
  double phase_[4] = {0, 0, 0, 0};
  uint8_t support_leg_[4] = {1, 1, 1, 1}, contact_[4] = {1, 1, 1, 1}, store_flag_[4] = {0, 0, 0, 0}, support_[4] = {1, 1, 1, 1};
  uint8_t leg_mode_[4] = {0, 0, 0, 0};
  int8_t limb_state_[4] = {0, 0, 0, 0}, leg_state_code_[4] = {0, 0, 0, 0};
};

} // namespace balance_controller
