// free_gait::PoseOptimizationGeometric on top of the C-ABI
// (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationGeometric.hpp,
//  free_gait_core/src/pose_optimization/PoseOptimizationGeometric.cpp:29-105).
#pragma once

#include "free_gait_core/PoseOptimizationBase.hpp"

namespace free_gait {

class PoseOptimizationGeometric : public PoseOptimizationBase {
 public:
  using PoseOptimizationBase::PoseOptimizationBase;

  void setStanceForOrientation(const Stance &stance) { stanceForOrientation_ = stance; }

  bool optimize(Pose &pose) {
    Marshalled m;
    if (!marshal(pose, m)) return false;
    double sfo[12];
    for (int l = 0; l < 4; ++l) {
      const LimbEnum limb = static_cast<LimbEnum>(l);
      if (!stanceForOrientation_.contains(limb)) throw std::out_of_range("stanceForOrientation_.at"); // :76-77
      for (int i = 0; i < 3; ++i) sfo[3 * l + i] = stanceForOrientation_.at(limb)(i);
    }
    const qlamd_pose_batch in = m.batch();
    double pose_out[7];
    if (qlamd_pose_geometric_batch(ctx_->get(), &m.prm, &in, sfo, 1, pose_out, QLAMD_MEM_HOST, nullptr) != QLAMD_OK)
      return false;
    pose.position = Position(pose_out[0], pose_out[1], pose_out[2]);
    pose.rotation = qlamd::RotationQuaternion(pose_out[3], pose_out[4], pose_out[5], pose_out[6]);
    return true;
  }

 private:
  Stance stanceForOrientation_;
};

} // namespace free_gait
