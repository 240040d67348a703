// Swing-leg torque of one leg, host/device.
// Replaces MyRobotSolver::update (single_leg_test/lib/model_test_header.cpp:412-503):
// RBDL InverseDynamics on the 3-joint leg chain (fixed base, gravity in the base frame) plus a
// Cartesian PD through the translation Jacobian.  The recursive Newton-Euler pass is written in
// base coordinates; the fixed foot link rides on link 3 (RBDL merges fixed bodies into the parent).
#pragma once

#include "balance_core.hpp"

namespace qlamd {

struct SwingParamsDev {
  double kp[3], kd[3];
  double period, accel_window, accel_scale, gravity;
};

template <class Tab>
QL_HD void leg_rnea(const Tab &tab, const double q[3], const double qd[3], const double qdd[3], const double g[3],
                    double tau[3]) {
  double R[4][9], p[4][3];
  {
    double Rc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pc[3] = {0, 0, 0};
    QL_UNROLL for (int k = 0; k < 4; k++) {
      double R0[9], Rs[9];
      QL_UNROLL for (int i = 0; i < 9; i++) R0[i] = tab[kTabR0 + 9 * k + i];
      if (k < 3) {
        double s, c;
        sincos_reduced(q[k], s, c);
        QL_UNROLL for (int i = 0; i < 3; i++) {
          Rs[i * 3 + 0] = R0[i * 3 + 0] * c + R0[i * 3 + 1] * s;
          Rs[i * 3 + 1] = R0[i * 3 + 1] * c - R0[i * 3 + 0] * s;
          Rs[i * 3 + 2] = R0[i * 3 + 2];
        }
      } else {
        QL_UNROLL for (int i = 0; i < 9; i++) Rs[i] = R0[i];
      }
      QL_UNROLL for (int i = 0; i < 3; i++)
        pc[i] += Rc[i * 3] * tab[kTabXyz + 3 * k] + Rc[i * 3 + 1] * tab[kTabXyz + 3 * k + 1] + Rc[i * 3 + 2] * tab[kTabXyz + 3 * k + 2];
      double Rn[9];
      QL_UNROLL for (int i = 0; i < 3; i++)
        QL_UNROLL for (int j = 0; j < 3; j++)
          Rn[i * 3 + j] = Rc[i * 3] * Rs[j] + Rc[i * 3 + 1] * Rs[3 + j] + Rc[i * 3 + 2] * Rs[6 + j];
      QL_UNROLL for (int i = 0; i < 9; i++) { Rc[i] = Rn[i]; R[k][i] = Rn[i]; }
      QL_UNROLL for (int i = 0; i < 3; i++) p[k][i] = pc[i];
    }
  }
  double w[3][3], al[3][3], a[3][3], z[3][3];
  QL_UNROLL for (int i = 0; i < 3; i++) {
    QL_UNROLL for (int k = 0; k < 3; k++) z[i][k] = R[i][3 * k + 2];
    double wp[3] = {0, 0, 0}, ap[3] = {0, 0, 0}, accp[3] = {-g[0], -g[1], -g[2]}, d[3] = {0, 0, 0};
    if (i > 0) {
      QL_UNROLL for (int k = 0; k < 3; k++) { wp[k] = w[i - 1][k]; ap[k] = al[i - 1][k]; accp[k] = a[i - 1][k]; d[k] = p[i][k] - p[i - 1][k]; }
    }
    const double zq[3] = {qd[i] * z[i][0], qd[i] * z[i][1], qd[i] * z[i][2]};
    double wxzq[3], axd[3], wxd[3], wxwxd[3];
    cross3(wp, zq, wxzq); cross3(ap, d, axd); cross3(wp, d, wxd); cross3(wp, wxd, wxwxd);
    QL_UNROLL for (int k = 0; k < 3; k++) {
      w[i][k] = wp[k] + zq[k];
      al[i][k] = ap[k] + qdd[i] * z[i][k] + wxzq[k];
      a[i][k] = accp[k] + axd[k] + wxwxd[k];
    }
  }
  double F[4][3], N[4][3], c[4][3];
  QL_UNROLL for (int b = 0; b < 4; b++) {
    const int i = b < 3 ? b : 2;
    const double m = tab[kTabMass + b];
    double dc[3], axd[3], wxd[3], wxwxd[3];
    QL_UNROLL for (int k = 0; k < 3; k++) {
      // m * c_b = m * p_b + R_b (m com)
      const double mc = R[b][3 * k] * tab[kTabMcom + 3 * b] + R[b][3 * k + 1] * tab[kTabMcom + 3 * b + 1] + R[b][3 * k + 2] * tab[kTabMcom + 3 * b + 2];
      c[b][k] = p[b][k] + mc / m;
      dc[k] = c[b][k] - p[i][k];
    }
    cross3(al[i], dc, axd); cross3(w[i], dc, wxd); cross3(w[i], wxd, wxwxd);
    QL_UNROLL for (int k = 0; k < 3; k++) F[b][k] = m * (a[i][k] + axd[k] + wxwxd[k]);
    // I_base = R I_link R'
    const double I6[6] = {tab[kTabInertia + 6 * b], tab[kTabInertia + 6 * b + 1], tab[kTabInertia + 6 * b + 2],
                          tab[kTabInertia + 6 * b + 3], tab[kTabInertia + 6 * b + 4], tab[kTabInertia + 6 * b + 5]};
    const double Il[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
    // apply as R (Il (R' v))
    const auto Iapply = [&](const double v[3], double o[3]) {
      double t[3], u[3];
      irot(R[b], v, t);
      QL_UNROLL for (int r = 0; r < 3; r++) u[r] = Il[3 * r] * t[0] + Il[3 * r + 1] * t[1] + Il[3 * r + 2] * t[2];
      rot(R[b], u, o);
    };
    double Ia[3], Iw[3], wIw[3];
    Iapply(al[i], Ia); Iapply(w[i], Iw);
    cross3(w[i], Iw, wIw);
    QL_UNROLL for (int k = 0; k < 3; k++) N[b][k] = Ia[k] + wIw[k];
  }
  QL_UNROLL for (int i = 0; i < 3; i++) {
    double acc = 0.0;
    QL_UNROLL for (int b = 0; b < 4; b++) {
      if (b < i) continue;
      const double d[3] = {c[b][0] - p[i][0], c[b][1] - p[i][1], c[b][2] - p[i][2]};
      double mo[3];
      cross3(d, F[b], mo);
      acc += z[i][0] * (N[b][0] + mo[0]) + z[i][1] * (N[b][1] + mo[1]) + z[i][2] * (N[b][2] + mo[2]);
    }
    tau[i] = acc;
  }
}

template <class Tab>
QL_HD void swing_leg_torque(const Tab &tab, const SwingParamsDev &P, const double q_id[3], const double q[3],
                            const double qd[3], const double qd_oldest[3], const double target_pos[3],
                            const double target_vel[3], double tau[3]) {
  double qdd[3], tid[3], foot[3], J[9], Gq[3];
  const double g[3] = {0.0, 0.0, -P.gravity};
  QL_UNROLL for (int i = 0; i < 3; i++) qdd[i] = P.accel_scale * ((qd[i] - qd_oldest[i]) / (P.period * P.accel_window));
  leg_rnea(tab, q_id, qd, qdd, g, tid);
  leg_kinematics(tab, q, g, foot, J, Gq);
  double f[3];
  QL_UNROLL for (int r = 0; r < 3; r++) {
    const double v = J[3 * r] * qd[0] + J[3 * r + 1] * qd[1] + J[3 * r + 2] * qd[2];
    f[r] = P.kp[r] * (target_pos[r] - foot[r]) + P.kd[r] * (target_vel[r] - v);
  }
  QL_UNROLL for (int j = 0; j < 3; j++) tau[j] = (J[j] * f[0] + J[3 + j] * f[1] + J[6 + j] * f[2]) + tid[j];
}

// ---- the swing branch of RosBalanceController::update (ros_balance_controller.cpp:467-603,720-756) -------
struct PidParamsDev {
  double p[12], i[12], d[12], i_max[12], i_min[12], lower[12], upper[12];
  int antiwindup;
};

QL_HD double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
QL_HD bool finite_d(double v) { return v == v && v - v == 0.0; }

// gains and limits of one leg's three joints (a slice of PidParamsDev; fetched once, up front, by the kernel)
struct PidLeg {
  double p[3], i[3], d[3], i_max[3], i_min[3], lower[3], upper[3];
  int antiwindup;
};
QL_HD void pid_leg_of(const PidParamsDev &g, int leg, PidLeg &o) {
  QL_UNROLL for (int k = 0; k < 3; k++) {
    const int j = 3 * leg + k;
    o.p[k] = g.p[j]; o.i[k] = g.i[j]; o.d[k] = g.d[j]; o.i_max[k] = g.i_max[j]; o.i_min[k] = g.i_min[j];
    o.lower[k] = g.lower[j]; o.upper[k] = g.upper[j];
  }
  o.antiwindup = g.antiwindup;
}

// control_toolbox::Pid::computeCommand(error, dt) (third-party, restated; see oracle_swing.h)
QL_HD double pid_command(const PidLeg &g, int k, double error, double dt, double &e_last, double &e_int) {
  if (dt == 0.0 || !finite_d(error)) return 0.0;
  double error_dot = 0.0;
  if (dt > 0.0) {
    error_dot = (error - e_last) / dt;
    e_last = error;
  }
  if (!finite_d(error_dot)) return 0.0;
  const double p_term = g.p[k] * error;
  e_int += dt * error;
  if (g.antiwindup && g.i[k] != 0.0) {
    const double a = g.i_min[k] / g.i[k], b = g.i_max[k] / g.i[k];
    e_int = clampd(e_int, a < b ? a : b, a < b ? b : a);
  }
  double i_term = g.i[k] * e_int;
  if (!g.antiwindup) i_term = clampd(i_term, g.i_min[k], g.i_max[k]);
  return p_term + i_term + g.d[k] * error_dot;
}

// One non-support leg: joint PID on the position command, gravity compensation, or the swing-leg torque,
// selected by the leg mode (1 "joint", 2 "leg_mode", 3 "cartesian", 4 "footstep", 0 never set = as "joint").
template <class Tab>
QL_HD void swing_branch_leg(const Tab &tab, const SwingParamsDev &SP, const PidLeg &pid, int leg_mode,
                            const double base_quat[4], const double q_id[3], const double q[3], const double qd[3],
                            const double qd_oldest[3], const double target_pos[3], const double target_vel[3],
                            const double joint_command[3], double period, double e_last[3], double e_int[3],
                            double effort[3]) {
  double Rm[9], g[3], foot[3], J[9], G[3], tsw[3];
  quat_to_matrix(base_quat, Rm);
  const double gW[3] = {0.0, 0.0, -9.8};
  rot(Rm, gW, g); // base_orientation.rotate(...), :471 -- as the reference writes it
  leg_kinematics(tab, q, g, foot, J, G);
  swing_leg_torque(tab, SP, q_id, q, qd, qd_oldest, target_pos, target_vel, tsw);
  QL_UNROLL for (int k = 0; k < 3; k++) {
    const double cmd = clampd(joint_command[k], pid.lower[k], pid.upper[k]); // enforceJointLimits
    double e = pid_command(pid, k, cmd - q[k], period, e_last[k], e_int[k]);
    if (leg_mode == 3 || leg_mode == 4) e = tsw[k];
    else if (leg_mode != 2) e += G[k];
    else e = G[k];
    effort[k] = e;
  }
}

template <class Tab>
QL_HD void swing_branch_leg(const Tab &tab, const SwingParamsDev &SP, const PidParamsDev &pid, int leg, int leg_mode,
                            const double base_quat[4], const double q_id[3], const double q[3], const double qd[3],
                            const double qd_oldest[3], const double target_pos[3], const double target_vel[3],
                            const double joint_command[3], double period, double e_last[3], double e_int[3],
                            double effort[3]) {
  PidLeg pl;
  pid_leg_of(pid, leg, pl);
  swing_branch_leg(tab, SP, pl, leg_mode, base_quat, q_id, q, qd, qd_oldest, target_pos, target_vel, joint_command, period,
                   e_last, e_int, effort);
}

} // namespace qlamd
