/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * free_gait_msgs/RobotState in the ROS 1 wire format -> the fields RosBalanceController::baseCommandCallback
 * reads (balance_controller/src/ros_controller/ros_balance_controller.cpp:761-1083).  Message definitions:
 * free_gait_msgs/msg/RobotState.msg, LegMode.msg, EndEffectorTarget.msg; embedded standard messages
 * (std_msgs/Header, sensor_msgs/JointState, nav_msgs/Odometry, geometry_msgs/...) and the serialisation rules
 * are those of ROS 1 (roscpp_serialization: little-endian, unpadded, uint32 length prefixes for strings and
 * variable-length arrays).  ROS is absent here and the reference ships no recorded bag: PARITY UNPINNED; the
 * parser is checked against an independent serialiser written from the same specification
 * (tests/ros1_wire.py).
 */
#ifndef ORACLE_WIRE_H
#define ORACLE_WIRE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double des_pos[3], des_quat[4], des_linvel[3], des_angvel[3]; /* quaternion (w, x, y, z) as :766-769 */
  double joint_command[12];                                     /* :802-812 */
  double foot_position[12], foot_velocity[12], foot_acceleration[12]; /* :816-861 */
  double surface_normal[12], phase[4];
  uint8_t support_leg[4];
  uint8_t leg_mode[4];  /* 1 "joint", 2 "leg_mode", 3 "cartesian", 4 "footstep", 0 anything else (:876-964) */
} oracle_robot_state_fields;

/* Returns 0, 1 (message shorter than its own length fields say) or 2 (an array the callback indexes is too
 * short: joints.position[0..2], target_position/velocity/acceleration[0]). */
int oracle_robot_state_unpack(const uint8_t *msg, size_t len, oracle_robot_state_fields *out);

#ifdef __cplusplus
}
#endif
#endif
