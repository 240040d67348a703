// TEST INFRASTRUCTURE ONLY -- never part of the product library.
//
// Compiles the per-robot device arithmetic (csrc/balance_core.hpp) for the
// HOST with g++, so that CPU-only tests (-m "not gpu") can check the kernel's
// math against the oracle without a GPU.  The product C-ABI never calls this;
// it fails with QLAMD_ERR_NO_DEVICE when no GPU is present.
#include <cstdint>

#include "balance_core.hpp"
#include "params_build.hpp"

using namespace qlamd;

static void load(RobotIn &in, const double *q, const double *pos, const double *quat, const double *lv,
                 const double *av, const double *dpos, const double *dquat, const double *dlv, const double *dav,
                 const uint8_t *stance) {
  for (int i = 0; i < 12; i++) in.q[i] = q[i];
  for (int i = 0; i < 3; i++) {
    in.pos[i] = pos[i]; in.linvel[i] = lv[i]; in.angvel[i] = av[i];
    in.dpos[i] = dpos[i]; in.dlinvel[i] = dlv[i]; in.dangvel[i] = dav[i];
  }
  for (int i = 0; i < 4; i++) { in.quat[i] = quat[i]; in.dquat[i] = dquat[i]; }
  in.stance = 0;
  for (int l = 0; l < 4; l++) if (stance[l]) in.stance |= 1u << l;
  in.has_wrench = false;
}

extern "C" void mirror_balance_batch(const qlamd_balance_params *prm, int64_t B, const double *q,
                                     const double *pos, const double *quat, const double *lv, const double *av,
                                     const double *dpos, const double *dquat, const double *dlv,
                                     const double *dav, const uint8_t *stance, const double *normals,
                                     double *tau, double *grf, int32_t *status, int32_t *iters,
                                     int32_t *n_active) {
  qlamd_robot_model model;
  default_robot_model(&model);
  DeviceParams P;
  build_device_params(*prm, model, &P);
  for (int64_t i = 0; i < B; i++) {
    RobotIn in;
    load(in, q + 12 * i, pos + 3 * i, quat + 4 * i, lv + 3 * i, av + 3 * i, dpos + 3 * i, dquat + 4 * i,
         dlv + 3 * i, dav + 3 * i, stance + 4 * i);
    HostScratch scr;
    for (int l = 0; l < 4; l++) // phase A
      phase_a_leg(P.legtab + kTabPerLeg * l, l, (in.stance >> l) & 1u, in.q + 3 * l, in.quat, P.grav, scr);
    const QpResult r = normals ? phase_b_robot<true>(P, in, normals + 12 * i, scr)
                               : phase_b_robot<false>(P, in, nullptr, scr);
    for (int l = 0; l < 4; l++) { // phase C
      double t[3], f[3];
      phase_c_leg(l, ((in.stance >> l) & 1u) && r.status == kStatusOk, P.tau_max, scr, t, f);
      for (int a = 0; a < 3; a++) { tau[12 * i + 3 * l + a] = t[a]; if (grf) grf[12 * i + 3 * l + a] = f[a]; }
    }
    status[i] = r.status;
    if (iters) iters[i] = r.iters;
    if (n_active) n_active[i] = r.n_active;
  }
}

extern "C" void mirror_leg_kinematics(const double *q3, int leg, const double *g, double *p, double *J,
                                      double *Gq) {
  qlamd_balance_params prm;
  default_balance_params(&prm);
  qlamd_robot_model model;
  default_robot_model(&model);
  DeviceParams P;
  build_device_params(prm, model, &P);
  leg_kinematics(P.legtab + kTabPerLeg * leg, q3, g, p, J, Gq);
}

extern "C" void mirror_sincos(double x, double *s, double *c) { sincos_reduced(x, *s, *c); }
extern "C" void mirror_default_params(qlamd_balance_params *p) { default_balance_params(p); }

// ---- pose optimisation / dense QP (config 5), host build of gi_core.hpp + pose_core.hpp ----
#include "pose_one_lane.hpp"

template <int N>
struct HostScr {
  double a[N];
  double &at(int e) { return a[e]; }
};

extern "C" int mirror_qp_solve(int n, int p, int m, const double *G, const double *g0, const double *CE,
                               const double *ce0, const double *CI, const double *ci0, double *x, double *f) {
  typedef GiLayout<12, 2, 24> Ly;
  if (n > 12 || p > 2 || m > 24) return -1;
  static HostScr<Ly::kTotal> s;
  for (int i = 0; i < n * n; i++) s.at(Ly::G + i) = G[i];
  for (int i = 0; i < n; i++) s.at(Ly::G0 + i) = g0[i];
  for (int i = 0; i < n * p; i++) s.at(Ly::CE + i) = CE[i];
  for (int i = 0; i < p; i++) s.at(Ly::CE0 + i) = ce0[i];
  for (int i = 0; i < n * m; i++) s.at(Ly::CI + i) = CI[i];
  for (int i = 0; i < m; i++) s.at(Ly::CI0 + i) = ci0[i];
  const int st = gi_solve<12, 2, 24>(s, n, p, m, f, nullptr);
  for (int i = 0; i < n; i++) x[i] = s.at(Ly::X + i);
  return st;
}

extern "C" { int g_mirror_pose_variant = 6; } // 6 = register GI (the kernel's), 0 = LDS-layout GI

extern "C" void mirror_pose_sqp_batch(const PoseParamsDev *P, int64_t B, const double *stance,
                                      const uint8_t *mask, const double *nominal, const double *polygon,
                                      const int32_t *nverts, const double *rcom, const double *maxlen,
                                      const double *pose_in, double *pose_out, int32_t *iters, int32_t *status) {
  for (int64_t i = 0; i < B; i++) {
    PoseProblem pb;
    unsigned limb_mask = 0;
    for (int l = 0; l < 4; l++) if (!mask || mask[4 * i + l]) limb_mask |= 1u << l;
    pose_problem_load_legs(
        *P, pb, [&](int l, int a) { return stance[12 * i + 3 * l + a]; },
        [&](int l, int a) { return nominal[12 * i + 3 * l + a]; }, [&](int l) { return maxlen[4 * i + l]; }, [&]() { return limb_mask; });
    for (int l = 0; l < 4; l++) { pb.polygon[l][0] = polygon[8 * i + 2 * l]; pb.polygon[l][1] = polygon[8 * i + 2 * l + 1]; }
    for (int a = 0; a < 3; a++) pb.r_com[a] = rcom ? rcom[3 * i + a] : 0.0;
    pb.n_vertices = nverts ? nverts[i] : 4;
    double pose[7];
    for (int a = 0; a < 7; a++) pose[a] = pose_in[7 * i + a];
    HostScr<PoseGi::kTotal> s;
    int it = 0;
    status[i] = g_mirror_pose_variant == 6 ? pose_sqp6(*P, pb, s, pose, &it) : pose_sqp(*P, pb, s, pose, &it);
    if (iters) iters[i] = it;
    for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
  }
}

// PoseOptimizationQP / PoseConstraintsChecker / PoseOptimizationGeometric / BaseAuto::optimizePose (row f3):
// mode 1 = QP (pose_out, status), 2 = check (ok), 3 = geometric (pose_out), 4 = the whole sequence
extern "C" void mirror_pose_aux_batch(int mode, const PoseParamsDev *P, int64_t B, const double *stance,
                                      const uint8_t *mask, const double *nominal, const double *polygon,
                                      const int32_t *nverts, const double *rcom, const double *maxlen,
                                      const double *pose_in, const double *min_len, double leg_tol, double *pose_out,
                                      int32_t *status, uint8_t *ok, const double *sfo_in, int32_t *stage,
                                      int32_t *iters) {
  for (int64_t i = 0; i < B; i++) {
    PoseProblem pb;
    unsigned limb_mask = 0;
    for (int l = 0; l < 4; l++) if (!mask || mask[4 * i + l]) limb_mask |= 1u << l;
    pose_problem_load_legs(
        *P, pb, [&](int l, int a) { return stance[12 * i + 3 * l + a]; },
        [&](int l, int a) { return nominal[12 * i + 3 * l + a]; }, [&](int l) { return maxlen[4 * i + l]; }, [&]() { return limb_mask; });
    for (int l = 0; l < 4; l++) { pb.polygon[l][0] = polygon[8 * i + 2 * l]; pb.polygon[l][1] = polygon[8 * i + 2 * l + 1]; }
    for (int a = 0; a < 3; a++) pb.r_com[a] = rcom ? rcom[3 * i + a] : 0.0;
    pb.n_vertices = nverts ? nverts[i] : 4;
    double pose[7], mn[4], sfo[4][3];
    for (int a = 0; a < 7; a++) pose[a] = pose_in ? pose_in[7 * i + a] : (a == 3 ? 1.0 : 0.0);
    for (int k = 0; k < 4; k++) mn[k] = min_len ? min_len[4 * i + P->leg_order[k]] : 0.0;
    for (int l = 0; l < 4; l++)
      for (int a = 0; a < 3; a++) sfo[l][a] = (sfo_in ? sfo_in : stance)[12 * i + 3 * l + a];
    if (mode == 1) {
      HostScr<PoseQpGi::kTotal> s;
      status[i] = pose_qp(*P, pb, s, pose);
    } else if (mode == 2) {
      ok[i] = pose_check(pb, pose, mn, leg_tol) ? 1 : 0;
    } else if (mode == 3) {
      pose_geometric(pb, sfo, pose);
    } else {
      HostScr<(PoseQpGi::kTotal > Gi6Layout::kTotal ? PoseQpGi::kTotal : Gi6Layout::kTotal)> s;
      int stg = 0, it = 0;
      status[i] = base_auto_optimize_pose(*P, pb, sfo, mn, leg_tol, s, pose, &stg, &it);
      stage[i] = stg; iters[i] = it;
    }
    if (mode != 2)
      for (int a = 0; a < 7; a++) pose_out[7 * i + a] = pose[a];
  }
}

// ---- leg state machine (row f2): same unpack / apply as leg_state_kernel ----
#include "leg_state_core.hpp"
extern "C" void mirror_leg_state_batch(int64_t B, const uint8_t *support_leg, const double *phase, const uint8_t *is_footstep,
                                       const uint8_t *contact, const double *joint_position, int index_quirk,
                                       int8_t *limb_state, uint8_t *store_flag, double *stored, double *cmd, double *foot,
                                       uint8_t *support, int8_t *code) {
  for (int64_t i = 0; i < B; i++) {
    LegStateRobot r;
    for (int l = 0; l < 4; l++) {
      r.support_leg[l] = support_leg[4 * i + l] != 0; r.is_footstep[l] = is_footstep[4 * i + l] != 0;
      r.contact[l] = contact[4 * i + l] != 0; r.phase[l] = phase[4 * i + l];
      r.limb_state[l] = limb_state[4 * i + l]; r.store_flag[l] = store_flag[4 * i + l] != 0;
    }
    leg_state_machine(r, index_quirk != 0);
    for (int l = 0; l < 4; l++) {
      limb_state[4 * i + l] = (int8_t)r.limb_state[l]; store_flag[4 * i + l] = r.store_flag[l];
      code[4 * i + l] = (int8_t)r.code[l];
      if (r.support_written[l]) support[4 * i + l] = r.support[l];
      if (r.nudge_bumped[l]) { foot[12 * i + 3 * l] -= 0.005; foot[12 * i + 3 * l + 2] += 0.02; }
      if (r.nudge_late[l]) foot[12 * i + 3 * l + 2] -= 0.01;
      for (int k = 0; k < 3; k++) {
        if (r.capture[l]) stored[12 * i + 3 * l + k] = joint_position[12 * i + 3 * l + k];
        if (r.hold[l]) cmd[12 * i + 3 * l + k] = stored[12 * i + 3 * l + k];
      }
    }
  }
}

// ---- RobotState wire format (row f2) ----
#include "wire_core.hpp"
extern "C" int mirror_robot_state_unpack(const uint8_t *msg, int64_t len, RobotStateFields *out) {
  memset(out, 0, sizeof(*out));
  return robot_state_unpack(PlainBytes{msg}, len, *out);
}

// ---- analytic leg IK (row f4) ----
extern "C" int mirror_leg_ik(int leg, const double *p_base, int config, const double *geom, double *q) {
  qlamd_balance_params prm;
  default_balance_params(&prm);
  qlamd_robot_model model;
  default_robot_model(&model);
  DeviceParams P;
  build_device_params(prm, model, &P);
  return leg_inverse_kinematics(P.legtab + kTabPerLeg * leg, p_base, config, geom, q) ? 1 : 0;
}

// ---- swing-leg torque (row a18) ----
#include "swing_core.hpp"
extern "C" void mirror_swing_leg(int leg, const SwingParamsDev *SP, const double *q_id, const double *q, const double *qd,
                                 const double *qd_old, const double *tp, const double *tv, double *tau) {
  qlamd_balance_params prm;
  default_balance_params(&prm);
  qlamd_robot_model model;
  default_robot_model(&model);
  DeviceParams P;
  build_device_params(prm, model, &P);
  swing_leg_torque(P.legtab + kTabPerLeg * leg, *SP, q_id, q, qd, qd_old, tp, tv, tau);
}

// the same arithmetic on a caller-given robot model (tests/test_rbdl_pin.py: the reference's RBDL test model)
extern "C" void mirror_swing_leg_model(const qlamd_robot_model *model, int leg, const SwingParamsDev *SP, const double *q_id,
                                       const double *q, const double *qd, const double *qd_old, const double *tp,
                                       const double *tv, double *tau) {
  qlamd_balance_params prm;
  default_balance_params(&prm);
  DeviceParams P;
  build_device_params(prm, *model, &P);
  swing_leg_torque(P.legtab + kTabPerLeg * leg, *SP, q_id, q, qd, qd_old, tp, tv, tau);
}

extern "C" void mirror_swing_branch_leg(int leg, int leg_mode, const SwingParamsDev *SP, const PidParamsDev *pid,
                                        const double *quat, const double *q_id, const double *q, const double *qd,
                                        const double *qd_old, const double *tp, const double *tv, const double *cmd,
                                        double period, double *e_last, double *e_int, double *effort) {
  qlamd_balance_params prm;
  default_balance_params(&prm);
  qlamd_robot_model model;
  default_robot_model(&model);
  DeviceParams P;
  build_device_params(prm, model, &P);
  swing_branch_leg(P.legtab + kTabPerLeg * leg, *SP, *pid, leg, leg_mode, quat, q_id, q, qd, qd_old, tp, tv, cmd, period,
                   e_last, e_int, effort);
}

extern "C" int mirror_qp6_solve(int p, int m, const double *G, const double *g0, const double *CE, double ce0,
                                const double *CI, const double *ci0, double *x, double *f) {
  HostScr<Gi6Layout::kTotal> s;
  double Gr[36], g[6], ce[6], xx[6];
  for (int i = 0; i < 36; i++) Gr[i] = G[i];
  for (int i = 0; i < 6; i++) { g[i] = g0[i]; ce[i] = p ? CE[i] : 0.0; }
  for (int i = 0; i < 6 * m; i++) s.at(Gi6Layout::CI + i) = CI[i];
  for (int i = 0; i < m; i++) s.at(Gi6Layout::CI0 + i) = ci0[i];
  const int st = gi6_solve(s, Gr, g, ce, ce0, p, m, xx, f);
  for (int i = 0; i < 6; i++) x[i] = xx[i];
  return st;
}
