// Per-leg state machine of the balance controller plugin, one robot per call:
//   command side   ros_balance_controller.cpp:966-1078   (desired state, stance / swing phase)
//   contact side   ros_balance_controller.cpp:1084-1135  (early / late touch-down, bumped, lost contact)
//   per-tick side  ros_balance_controller.cpp:234-380    (support flags, leg_state codes, target nudges,
//                                                         hold-position joint commands)
// State ids: StateSwitcher.hpp:62-72.  Compiles for the device and, for the CPU-only tests, for the host.
#pragma once

#include <stdint.h>

#include "balance_core.hpp"

namespace qlamd {

enum LegState : int {
  kLsInit = 0, kLsStanceNormal, kLsStanceSlipping, kLsStanceLostContact, kLsSwingNormal, kLsSwingLateLiftOff,
  kLsSwingEarlyTouchDown, kLsSwingBumpedIntoObstacle, kLsSwingLatelyTouchDown
};

struct LegStateRobot {
  // inputs
  bool support_leg[4], is_footstep[4], contact[4];
  double phase[4];
  // persistent
  int limb_state[4];
  bool store_flag[4];
  // outputs
  bool support[4], support_written[4];
  int code[4];
  bool nudge_bumped[4], nudge_late[4]; // foot target: (-0.005, 0, +0.02) / (0, 0, -0.01)
  bool capture[4], hold[4];            // stored <- measured joints / command <- stored joints
};

QL_HD void leg_state_machine(LegStateRobot &r, bool index_quirk) {
  // the contact loop advances its limb index only at the end of a full pass (:1134); a `continue`
  // leaves it where it is, so the next contact lands on the same limb
  int i = 0;
  QL_UNROLL for (int c = 0; c < 4; c++) {
    bool sup = false, foot = false;
    double ph = 0.0;
    int cur = 0;
    QL_UNROLL for (int l = 0; l < 4; l++)
      if (l == i) { sup = r.support_leg[l]; foot = r.is_footstep[l]; ph = r.phase[l]; cur = r.limb_state[l]; }
    const bool touching = r.contact[c];
    int next = cur;
    bool advance = true;
    if (!sup) { // desired SwingNormal, swing phase = ph
      next = kLsSwingNormal;
      if (!foot) advance = !index_quirk;
      else if (ph > 0.5) { if (touching) next = kLsSwingEarlyTouchDown; }
      else if (ph > 0.2) { if (touching) next = kLsSwingBumpedIntoObstacle; }
    } else {    // desired StanceNormal, stance phase = ph
      if (!foot) { next = kLsStanceNormal; advance = !index_quirk; }
      else {
        if (touching) next = kLsStanceNormal;
        else if (ph < 0.1) next = kLsSwingLatelyTouchDown;
        if (ph > 0.5 && !touching) next = kLsStanceLostContact;
      }
    }
    QL_UNROLL for (int l = 0; l < 4; l++)
      if (l == i) r.limb_state[l] = next;
    if (advance) i++;
  }
  QL_UNROLL for (int l = 0; l < 4; l++) {
    const int s = r.limb_state[l];
    r.code[l] = 0; r.support_written[l] = true; r.support[l] = false;
    r.nudge_bumped[l] = r.nudge_late[l] = r.capture[l] = r.hold[l] = false;
    if (s == kLsSwingNormal) { r.store_flag[l] = false; }
    else if (s == kLsStanceNormal) { r.support[l] = true; r.code[l] = 2; r.store_flag[l] = false; }
    else if (s == kLsSwingEarlyTouchDown) { r.support[l] = true; r.code[l] = 1; }
    else if (s == kLsSwingBumpedIntoObstacle) { r.nudge_bumped[l] = true; }
    else if (s == kLsSwingLatelyTouchDown || s == kLsStanceLostContact) {
      r.nudge_late[l] = s == kLsSwingLatelyTouchDown;
      r.capture[l] = !r.store_flag[l];
      r.hold[l] = r.store_flag[l];
      r.store_flag[l] = true;
      r.code[l] = s == kLsSwingLatelyTouchDown ? 3 : -1;
    }
    else if (s == kLsInit) { r.support[l] = true; }
    else { r.support_written[l] = false; } // StanceSlipping / SwingLateLiftOff: no case in update()
  }
}

} // namespace qlamd
