// free_gait::PoseOptimizationSQP on top of the C-ABI
// (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationSQP.hpp:36-54,
//  free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111).
#pragma once

#include "free_gait_core/PoseOptimizationBase.hpp"

namespace free_gait {

class PoseOptimizationSQP : public PoseOptimizationBase {
 public:
  using PoseOptimizationBase::PoseOptimizationBase;

  bool optimize(Pose &pose) {
    Marshalled m;
    if (!marshal(pose, m)) return false;
    const qlamd_pose_batch in = m.batch();
    double pose_out[7];
    int32_t status = -1;
    const int rc = qlamd_pose_sqp_batch(ctx_->get(), &m.prm, &in, 1, pose_out, &iterations_, &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false;
    pose.position = Position(pose_out[0], pose_out[1], pose_out[2]);
    pose.rotation = qlamd::RotationQuaternion(pose_out[3], pose_out[4], pose_out[5], pose_out[6]);
    return true;
  }
  size_t getNumberOfIterations() const { return static_cast<size_t>(iterations_); }

 private:
  int32_t iterations_ = 0;
};

} // namespace free_gait
