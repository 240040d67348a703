/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into the product library.
 *
 * CPU restatement of the per-leg state machine of the balance controller plugin
 * (SURVEY.md section 8, row f2):
 *   command side   balance_controller/src/ros_controller/ros_balance_controller.cpp:966-1078
 *                  (baseCommandCallback: desired state and phases from RobotState.*_leg_mode)
 *   contact side   ros_balance_controller.cpp:1084-1135 (footContactsCallback), including its
 *                  `continue` that skips `i++`, so that the next contact is applied to the same limb
 *   per-tick side  ros_balance_controller.cpp:234-380 (the switch in update(): support flags,
 *                  leg_state codes, foot-target nudges, hold-position commands)
 * State ids follow balance_controller/include/state_switcher/StateSwitcher.hpp:62-72.
 * Pure integer/flag logic: parity is bit-exact.  The reference has no test for it; the restatement is
 * pinned only by reading (PARITY UNPINNED in the sense of the task statement).
 */
#ifndef ORACLE_LEG_STATE_H
#define ORACLE_LEG_STATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORACLE_LS_INIT = 0, ORACLE_LS_STANCE_NORMAL, ORACLE_LS_STANCE_SLIPPING, ORACLE_LS_STANCE_LOST_CONTACT,
  ORACLE_LS_SWING_NORMAL, ORACLE_LS_SWING_LATE_LIFT_OFF, ORACLE_LS_SWING_EARLY_TOUCH_DOWN,
  ORACLE_LS_SWING_BUMPED_INTO_OBSTACLE, ORACLE_LS_SWING_LATELY_TOUCH_DOWN
};

/* One robot, one tick: command -> contacts -> update().  All arrays by limb id (LF, RF, RH, LH).
 * in:     support_leg[4], phase[4], is_footstep[4], contact[4], joint_position[12]
 * in/out: limb_state[4], store_flag[4], stored_joint_position[12], joint_command[12], foot_target[12],
 *         support[4] (left untouched for states update() has no case for)
 * out:    leg_state_code[4] (std_msgs/Int8MultiArray leg_state, :222-224: 0 unless a case sets it)
 * index_quirk != 0 reproduces the skipped `i++`. */
void oracle_leg_state_machine(const uint8_t *support_leg, const double *phase, const uint8_t *is_footstep,
                              const uint8_t *contact, const double *joint_position, int index_quirk,
                              int8_t *limb_state, uint8_t *store_flag, double *stored_joint_position,
                              double *joint_command, double *foot_target, uint8_t *support, int8_t *leg_state_code);

#ifdef __cplusplus
}
#endif
#endif
