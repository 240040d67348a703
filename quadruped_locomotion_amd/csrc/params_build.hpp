// Host side: fold qlamd_balance_params + qlamd_robot_model into the
// batch-invariant DeviceParams block the kernels read.
#pragma once

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "balance_core.hpp"
#include "qlamd.h"
#include "qlamd_robot_constants.h"

namespace qlamd {

inline void default_balance_params(qlamd_balance_params *p) {
  // balance_controller/config/controller_gains.yaml:1-41
  const double kp_t[3] = {5000, 5000, 10000}, kd_t[3] = {5000, 4000, 5000}, kff_t[3] = {10, 10, 100};
  const double kp_r[3] = {10000, 10000, 4000}, kd_r[3] = {1000, 1000, 1000}, kff_r[3] = {0.2, 0.2, 1000};
  const double S[6] = {1, 5, 1, 10, 10, 5};
  // quadruped_state.cpp:83-97, limb order LF, RF, RH, LH
  const double hips[4][3] = {{0.42, 0.075, 0.0}, {0.42, -0.075, 0.0}, {-0.42, -0.075, 0.0}, {-0.42, 0.075, 0.0}};
  memcpy(p->kp_trans, kp_t, sizeof(kp_t)); memcpy(p->kd_trans, kd_t, sizeof(kd_t));
  memcpy(p->kff_trans, kff_t, sizeof(kff_t));
  memcpy(p->kp_rot, kp_r, sizeof(kp_r)); memcpy(p->kd_rot, kd_r, sizeof(kd_r));
  memcpy(p->kff_rot, kff_r, sizeof(kff_r));
  memcpy(p->force_weights, S, sizeof(S));
  p->regularizer = 0.0001; p->friction = 0.6; p->min_normal_force = 10.0;
  p->torque_limit = 300.0;  // ros_balance_controller.cpp:451-454
  p->torso_mass = 27.0;     // quadruped_state.cpp:28
  for (int i = 0; i < 4; i++) p->leg_mass[i] = 6.0; // quadruped_state.cpp:36-41
  p->gravity = 9.8;         // VirtualModelController.cpp:165
  p->grav_comp_percentage = 1.0; // VirtualModelController.cpp:54
  p->com_in_base[0] = p->com_in_base[1] = p->com_in_base[2] = 0.0;
  memcpy(p->hip_in_base, hips, sizeof(hips));
}

inline void default_robot_model(qlamd_robot_model *m) {
  memcpy(m->joint_xyz, QLAMD_JOINT_XYZ, sizeof(m->joint_xyz));
  memcpy(m->joint_rpy, QLAMD_JOINT_RPY, sizeof(m->joint_rpy));
  memcpy(m->link_mass, QLAMD_LINK_MASS, sizeof(m->link_mass));
  memcpy(m->link_com, QLAMD_LINK_COM, sizeof(m->link_com));
  memcpy(m->link_inertia, QLAMD_LINK_INERTIA, sizeof(m->link_inertia));
  m->base_mass = QLAMD_BASE_MASS;
  memcpy(m->base_com, QLAMD_BASE_COM, sizeof(m->base_com));
  memcpy(m->base_inertia, QLAMD_BASE_INERTIA, sizeof(m->base_inertia));
}

// URDF fixed-axis rpy: R = Rz(yaw) Ry(pitch) Rx(roll)  (kdl_parser)
inline void rpy_to_matrix(const double rpy[3], double R[9]) {
  const double cr = cos(rpy[0]), sr = sin(rpy[0]);
  const double cp = cos(rpy[1]), sp = sin(rpy[1]);
  const double cy = cos(rpy[2]), sy = sin(rpy[2]);
  R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
  R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
  R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

inline void build_device_params(const qlamd_balance_params &p, const qlamd_robot_model &m, DeviceParams *d) {
  memset(d, 0, sizeof(*d));
  for (int i = 0; i < 3; i++) {
    d->kp_t[i] = p.kp_trans[i]; d->kd_t[i] = p.kd_trans[i]; d->kff_t[i] = p.kff_trans[i];
    d->kp_r[i] = p.kp_rot[i];   d->kd_r[i] = p.kd_rot[i];   d->kff_r[i] = p.kff_rot[i];
  }
  for (int i = 0; i < 6; i++) d->S[i] = p.force_weights[i];
  d->w_reg = p.regularizer; d->mu = p.friction; d->f_min = p.min_normal_force; d->tau_max = p.torque_limit;
  d->grav = p.gravity;
  d->refine_passes = 1;
  d->keep_on_failure = 0;
  d->warm_fallback = 1;
  double mass = p.torso_mass;
  double arm[3] = {p.torso_mass * p.com_in_base[0], p.torso_mass * p.com_in_base[1], p.torso_mass * p.com_in_base[2]};
  for (int l = 0; l < 4; l++) {
    mass += p.leg_mass[l];
    for (int i = 0; i < 3; i++) arm[i] += p.leg_mass[l] * (p.hip_in_base[l][i] - p.com_in_base[i]);
  }
  d->Fg_scale = p.grav_comp_percentage * mass;
  for (int i = 0; i < 3; i++) d->Tg_arm[i] = p.grav_comp_percentage * arm[i];
  for (int l = 0; l < 4; l++) {
    double *tab = d->legtab + kTabPerLeg * l;
    for (int k = 0; k < 4; k++) {
      rpy_to_matrix(m.joint_rpy[l][k], tab + kTabR0 + 9 * k);
      tab[kTabMass + k] = m.link_mass[l][k];
      for (int i = 0; i < 3; i++) {
        tab[kTabXyz + 3 * k + i] = m.joint_xyz[l][k][i];
        tab[kTabMcom + 3 * k + i] = m.link_mass[l][k] * m.link_com[l][k][i];
      }
      for (int i = 0; i < 6; i++) tab[kTabInertia + 6 * k + i] = m.link_inertia[l][k][i];
    }
  }
}

} // namespace qlamd
