/* ORACLE -- TEST INFRASTRUCTURE ONLY.  See oracle_tick.h. */
#include "oracle_tick.h"

#include <string.h>

#include "oracle_leg_state.h"
#include "oracle_quadprog.h"

int oracle_full_tick(const oracle_balance_params *bp, const oracle_swing_params *sp_in, const oracle_pid_params *pid,
                     const uint8_t *msg, size_t len, const double q[12], const double qd[12], const double qd_oldest[12],
                     const double base_pos[3], const double base_quat[4], const double base_linvel[3],
                     const double base_angvel[3], const uint8_t contact[4], double period, int index_quirk,
                     int keep_on_failure, oracle_tick_state *s, int8_t leg_state_code[4], int *message_status) {
  /* baseCommandCallback, :761-1083: only a message that deserialises replaces the stored command */
  oracle_robot_state_fields f;
  const int mst = oracle_robot_state_unpack(msg, len, &f);
  if (message_status) *message_status = mst;
  if (mst == 0) {
    s->command = f;
    s->has_command = 1;
  }
  if (!s->has_command) return ORACLE_TICK_NO_COMMAND;
  oracle_robot_state_fields *c = &s->command;

  /* leg modes in force: a known name replaces the mode, anything else leaves it (:876-964) */
  uint8_t is_footstep[4];
  for (int l = 0; l < 4; l++) {
    if (c->leg_mode[l] != 0) s->leg_mode[l] = c->leg_mode[l];
    is_footstep[l] = s->leg_mode[l] == 4;
  }

  /* footContactsCallback + the switch of update(): support legs, held joint commands, nudged foot targets; the
   * callback's members (commands, foot_positions) are what the switch modifies, so the nudges stay in the command */
  oracle_leg_state_machine(c->support_leg, c->phase, is_footstep, contact, q, index_quirk, s->limb_state, s->store_flag,
                           s->stored_joint_position, c->joint_command, c->foot_position, s->support, leg_state_code);

  /* update() :384-454: efforts of the support legs */
  double tau[12];
  const int st = oracle_balance_step(bp, q, base_pos, base_quat, base_linvel, base_angvel, c->des_pos, c->des_quat,
                                     c->des_linvel, c->des_angvel, s->support, NULL, tau, NULL, NULL, NULL, NULL, NULL);
  if (st == ORACLE_QP_OK || !keep_on_failure) memcpy(s->joint_effort, tau, sizeof(tau));

  /* update() :467-603: the legs that do not support */
  oracle_swing_params sp = *sp_in;
  sp.period = period;
  for (int l = 0; l < 4; l++) {
    if (s->support[l]) continue;
    oracle_swing_branch_leg(&sp, pid, l, s->leg_mode[l], base_quat, q + 3 * l, q + 3 * l, qd + 3 * l, qd_oldest + 3 * l,
                            c->foot_position + 3 * l, c->foot_velocity + 3 * l, c->joint_command + 3 * l, period,
                            s->pid_error_last + 3 * l, s->pid_error_integral + 3 * l, s->joint_effort + 3 * l);
  }
  return st;
}
