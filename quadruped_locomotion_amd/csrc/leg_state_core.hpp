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

// Written as integer arithmetic on 0 / 1 predicates and small lookup words, not as the reference's nest of ifs: one lane runs a
// robot's machine, and on the device a nest of ifs is a chain of ~100 short branches, bools are 64-bit lane masks in scalar
// registers (about 1000 issue slots at the tail of every parser block of the whole tick, 2.3 us; profiles/r6/tick_block_phases.txt).
// The comments name the branch of the reference each expression stands for.
QL_HD void leg_state_machine(LegStateRobot &r, bool index_quirk) {
  // per-leg predicates as bit masks (bit l = leg l): the contact loop picks them by its limb index
  unsigned supm = 0u, footm = 0u, gt05 = 0u, gt02 = 0u, lt01 = 0u, touch = 0u, lsw = 0u;
  QL_UNROLL for (int l = 0; l < 4; l++) {
    supm |= (r.support_leg[l] ? 1u : 0u) << l;
    footm |= (r.is_footstep[l] ? 1u : 0u) << l;
    gt05 |= (r.phase[l] > 0.5 ? 1u : 0u) << l;
    gt02 |= (r.phase[l] > 0.2 ? 1u : 0u) << l;
    lt01 |= (r.phase[l] < 0.1 ? 1u : 0u) << l;
    touch |= (r.contact[l] ? 1u : 0u) << l;
    lsw |= ((unsigned)r.limb_state[l] & 0xFFu) << (8 * l); // limb states travel as int8 (LegStatePtrs): a byte each
  }
  const unsigned stay = index_quirk ? 0u : 1u;
  // the contact loop advances its limb index only at the end of a full pass (:1134); a `continue`
  // leaves it where it is, so the next contact lands on the same limb
  unsigned i = 0u;
  QL_UNROLL for (int c = 0; c < 4; c++) {
    const unsigned sup = (supm >> i) & 1u, foot = (footm >> i) & 1u, g5 = (gt05 >> i) & 1u, g2 = (gt02 >> i) & 1u, l1 = (lt01 >> i) & 1u;
    const unsigned t = (touch >> c) & 1u, cur = (lsw >> (8u * i)) & 0xFFu;
    // desired SwingNormal (4), swing phase = ph: early touch-down (6) past half the swing, bumped into an obstacle (7) past a fifth
    const unsigned ft = foot & t;
    const unsigned nsw = (unsigned)kLsSwingNormal + 2u * (ft & g5) + 3u * (ft & (g5 ^ 1u) & g2);
    // desired StanceNormal (1), stance phase = ph: in contact, or late touch-down (8) in the first tenth, lost contact (3) past half
    const unsigned away = l1 ? (unsigned)kLsSwingLatelyTouchDown : cur;
    const unsigned nfoot = t ? (unsigned)kLsStanceNormal : (g5 ? (unsigned)kLsStanceLostContact : away);
    const unsigned nst = foot ? nfoot : (unsigned)kLsStanceNormal;
    const unsigned next = sup ? nst : nsw;
    lsw = (lsw & ~(0xFFu << (8u * i))) | (next << (8u * i));
    i += foot | stay; // a leg without a footstep `continue`s
  }
  // what update() (:234-380) does in each state, as lookup words indexed by the state (bit s / nibble s); states 2 and 5
  // (StanceSlipping, SwingLateLiftOff) and anything out of range have no case there
  constexpr unsigned kSupport = 1u << kLsInit | 1u << kLsStanceNormal | 1u << kLsSwingEarlyTouchDown;
  constexpr unsigned kWritten = kSupport | 1u << kLsSwingNormal | 1u << kLsSwingBumpedIntoObstacle | 1u << kLsSwingLatelyTouchDown |
                                1u << kLsStanceLostContact;
  constexpr unsigned kLateOrLost = 1u << kLsSwingLatelyTouchDown | 1u << kLsStanceLostContact;
  constexpr unsigned kClears = 1u << kLsSwingNormal | 1u << kLsStanceNormal; // store_flag <- false
  // leg_state code + 1, three bits a state: StanceNormal 2, EarlyTouchDown 1, LatelyTouchDown 3, LostContact -1, else 0
  constexpr unsigned long long kCode = 1ull << (3 * kLsInit) | 3ull << (3 * kLsStanceNormal) | 1ull << (3 * kLsStanceSlipping) |
                                       0ull << (3 * kLsStanceLostContact) | 1ull << (3 * kLsSwingNormal) | 1ull << (3 * kLsSwingLateLiftOff) |
                                       2ull << (3 * kLsSwingEarlyTouchDown) | 1ull << (3 * kLsSwingBumpedIntoObstacle) |
                                       4ull << (3 * kLsSwingLatelyTouchDown);
  QL_UNROLL for (int l = 0; l < 4; l++) {
    const unsigned sb = (lsw >> (8 * l)) & 0xFFu;
    r.limb_state[l] = (int)(signed char)sb;
    const unsigned known = sb < 9u ? 1u : 0u, s = known ? sb : 2u; // out of range: like a state without a case
    const unsigned late_or_lost = (kLateOrLost >> s) & 1u, stored = r.store_flag[l] ? 1u : 0u;
    r.support[l] = (kSupport >> s) & 1u;
    r.support_written[l] = (kWritten >> s) & 1u;
    r.code[l] = (int)((kCode >> (3u * s)) & 7ull) - 1;
    r.nudge_bumped[l] = s == (unsigned)kLsSwingBumpedIntoObstacle;
    r.nudge_late[l] = s == (unsigned)kLsSwingLatelyTouchDown;
    r.capture[l] = late_or_lost & (stored ^ 1u);
    r.hold[l] = late_or_lost & stored;
    r.store_flag[l] = (((kClears >> s) & 1u) ^ 1u) & (late_or_lost | stored);
  }
}

} // namespace qlamd
