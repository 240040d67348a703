/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_leg_state.h for the reference citations). */
#include "oracle_leg_state.h"

void oracle_leg_state_machine(const uint8_t *support_leg, const double *phase, const uint8_t *is_footstep,
                              const uint8_t *contact, const double *joint_position, int index_quirk,
                              int8_t *limb_state, uint8_t *store_flag, double *stored_joint_position,
                              double *joint_command, double *foot_target, uint8_t *support, int8_t *leg_state_code) {
  /* baseCommandCallback :966-1078 */
  int desired[4];
  double st_phase[4], sw_phase[4];
  for (int l = 0; l < 4; l++) {
    if (support_leg[l]) {
      desired[l] = ORACLE_LS_STANCE_NORMAL;
      st_phase[l] = phase[l];
      sw_phase[l] = 0.0;
    } else {
      desired[l] = ORACLE_LS_SWING_NORMAL;
      sw_phase[l] = phase[l];
      st_phase[l] = 0.0;
    }
  }
  /* footContactsCallback :1089-1135; `i` is only advanced at the end of a full pass (:1134) */
  int i = 0;
  for (int c = 0; c < 4; c++) {
    const int limb = i;
    const int is_contact = contact[c] != 0;
    if (desired[limb] == ORACLE_LS_SWING_NORMAL) {
      limb_state[limb] = ORACLE_LS_SWING_NORMAL;
      if (!is_footstep[limb]) {
        if (!index_quirk) i++;
        continue;
      }
      if (sw_phase[limb] > 0.5) {
        if (is_contact) limb_state[limb] = ORACLE_LS_SWING_EARLY_TOUCH_DOWN;
      } else if (sw_phase[limb] > 0.2) {
        if (is_contact) limb_state[limb] = ORACLE_LS_SWING_BUMPED_INTO_OBSTACLE;
      }
    }
    if (desired[limb] == ORACLE_LS_STANCE_NORMAL) {
      if (!is_footstep[limb]) {
        limb_state[limb] = ORACLE_LS_STANCE_NORMAL;
        if (!index_quirk) i++;
        continue;
      }
      if (is_contact) {
        limb_state[limb] = ORACLE_LS_STANCE_NORMAL;
      } else if (st_phase[limb] < 0.1) {
        limb_state[limb] = ORACLE_LS_SWING_LATELY_TOUCH_DOWN;
      }
      if (st_phase[limb] > 0.5) {
        if (!is_contact) limb_state[limb] = ORACLE_LS_STANCE_LOST_CONTACT;
      }
    }
    i++;
  }
  /* update() :234-380 */
  for (int l = 0; l < 4; l++) {
    leg_state_code[l] = 0;
    switch (limb_state[l]) {
      case ORACLE_LS_SWING_NORMAL:
        support[l] = 0; leg_state_code[l] = 0; store_flag[l] = 0;
        break;
      case ORACLE_LS_STANCE_NORMAL:
        support[l] = 1; leg_state_code[l] = 2; store_flag[l] = 0;
        break;
      case ORACLE_LS_SWING_EARLY_TOUCH_DOWN:
        support[l] = 1; leg_state_code[l] = 1;
        break;
      case ORACLE_LS_SWING_BUMPED_INTO_OBSTACLE:
        support[l] = 0;
        foot_target[3 * l + 0] -= 0.005; /* move back and up, :306-308 */
        foot_target[3 * l + 2] += 0.02;
        break;
      case ORACLE_LS_SWING_LATELY_TOUCH_DOWN:
      case ORACLE_LS_STANCE_LOST_CONTACT:
        support[l] = 0;
        if (limb_state[l] == ORACLE_LS_SWING_LATELY_TOUCH_DOWN) foot_target[3 * l + 2] -= 0.01; /* :321 */
        if (!store_flag[l]) {
          store_flag[l] = 1;
          for (int k = 0; k < 3; k++) stored_joint_position[3 * l + k] = joint_position[3 * l + k];
        } else {
          for (int k = 0; k < 3; k++) joint_command[3 * l + k] = stored_joint_position[3 * l + k];
        }
        leg_state_code[l] = limb_state[l] == ORACLE_LS_SWING_LATELY_TOUCH_DOWN ? 3 : -1;
        break;
      case ORACLE_LS_INIT:
        support[l] = 1;
        break;
      default: /* StanceSlipping, SwingLateLiftOff: "Unspecificed Limb State", nothing changes */
        break;
    }
  }
}
