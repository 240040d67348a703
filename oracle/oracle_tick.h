/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * One whole control tick of the balance controller plugin for one robot, chained from the other oracle pieces in
 * the order the reference runs them:
 *   baseCommandCallback   balance_controller/src/ros_controller/ros_balance_controller.cpp:761-1083
 *                         (a message that cannot be deserialised never reaches the callback: the members it
 *                         would have written -- desired base state, joint commands, foot targets, leg modes,
 *                         phases -- keep the values of the last message that did)
 *   footContactsCallback + the switch at the top of update()   :1084-1135, :234-380
 *   update() :384-454     virtual-model wrench -> force distribution -> efforts of the support legs, clamped;
 *                         on a failed solve ("VMC compute failed", :418-424) the efforts held from the previous
 *                         tick are commanded again (keep_on_failure != 0) -- or zeroed (the library's default)
 *   update() :467-603     swing branch for the legs that do not support
 * PARITY UNPINNED like the pieces it chains (ROS / kindr / KDL / OOQP / control_toolbox absent).
 */
#ifndef ORACLE_TICK_H
#define ORACLE_TICK_H

#include <stddef.h>
#include <stdint.h>

#include "oracle_balance.h"
#include "oracle_swing.h"
#include "oracle_wire.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_TICK_NO_COMMAND 4

/* What the plugin keeps between ticks for one robot. */
typedef struct {
  /* the command in force (members baseCommandCallback writes) */
  int has_command;
  oracle_robot_state_fields command;
  /* controller state */
  int8_t limb_state[4];
  uint8_t store_flag[4];
  double stored_joint_position[12];
  uint8_t leg_mode[4];
  uint8_t support[4];
  double pid_error_last[12], pid_error_integral[12];
  double joint_effort[12];
} oracle_tick_state;

/* Returns the status of the balance solve (ORACLE_QP_*), or ORACLE_TICK_NO_COMMAND when no well-formed message has
 * arrived yet (then nothing of `s` changes).  message_status: oracle_robot_state_unpack's code for this tick's message. */
int oracle_full_tick(const oracle_balance_params *bp, const oracle_swing_params *sp, const oracle_pid_params *pid,
                     const uint8_t *msg, size_t len, const double q[12], const double qd[12], const double qd_oldest[12],
                     const double base_pos[3], const double base_quat[4], const double base_linvel[3],
                     const double base_angvel[3], const uint8_t contact[4], double period, int index_quirk,
                     int keep_on_failure, oracle_tick_state *s, int8_t leg_state_code[4], int *message_status);

#ifdef __cplusplus
}
#endif
#endif
