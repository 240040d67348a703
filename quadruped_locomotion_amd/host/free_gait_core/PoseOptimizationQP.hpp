// free_gait::PoseOptimizationQP on top of the C-ABI
// (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationQP.hpp,
//  free_gait_core/src/pose_optimization/PoseOptimizationQP.cpp:42-140): base position only, the orientation of
// `pose` is an input.
#pragma once

#include "free_gait_core/PoseOptimizationBase.hpp"

namespace free_gait {

class PoseOptimizationQP : public PoseOptimizationBase {
 public:
  using PoseOptimizationBase::PoseOptimizationBase;

  bool optimize(Pose &pose) {
    Marshalled m;
    if (!marshal(pose, m)) return false;
    const qlamd_pose_batch in = m.batch();
    double pose_out[7];
    int32_t status = -1;
    const int rc = qlamd_pose_qp_batch(ctx_->get(), &m.prm, &in, 1, pose_out, &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false; // solver_->minimize failed (:134)
    pose.position = Position(pose_out[0], pose_out[1], pose_out[2]);
    return true;
  }
};

} // namespace free_gait
