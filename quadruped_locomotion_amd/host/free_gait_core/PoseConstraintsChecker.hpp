// free_gait::PoseConstraintsChecker on top of the C-ABI
// (free_gait_core/include/free_gait_core/pose_optimization/PoseConstraintsChecker.hpp:19-33,
//  free_gait_core/src/pose_optimization/PoseConstraintsChecker.cpp:23-64).
#pragma once

#include "free_gait_core/PoseOptimizationBase.hpp"

namespace free_gait {

class PoseConstraintsChecker : public PoseOptimizationBase {
 public:
  using PoseOptimizationBase::PoseOptimizationBase;

  void setTolerances(const double centerOfMassTolerance, const double legLengthTolerance) {
    centerOfMassTolerance_ = centerOfMassTolerance; // stored; the reference tests the un-shrunk region (:44-49)
    legLengthTolerance_ = legLengthTolerance;
  }

  bool check(const Pose &pose) {
    Marshalled m;
    if (!marshal(pose, m)) return false;
    const qlamd_pose_batch in = m.batch();
    uint8_t ok = 0;
    const int rc = qlamd_pose_check_batch(ctx_->get(), &m.prm, &in, m.minlen, legLengthTolerance_, 1, &ok, QLAMD_MEM_HOST, nullptr);
    return rc == QLAMD_OK && ok != 0;
  }

 private:
  double centerOfMassTolerance_ = 0.0;
  double legLengthTolerance_ = 0.0;
};

} // namespace free_gait
