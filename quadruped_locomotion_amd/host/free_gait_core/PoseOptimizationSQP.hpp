// free_gait::PoseOptimizationSQP on top of the C-ABI
// (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationSQP.hpp:36-54,
//  free_gait_core/src/pose_optimization/PoseOptimizationBase.cpp:21-50,
//  free_gait_core/src/pose_optimization/PoseOptimizationSQP.cpp:58-111).
// `Stance` keeps the insertion order the way the reference's unordered_map does under libstdc++
// (iteration = reverse insertion order), which fixes the leg order handed to the device.
#pragma once

#include <utility>
#include <vector>

#include "qlamd/types.hpp"

namespace free_gait {

using qlamd::LimbEnum;
using qlamd::Pose;
using qlamd::Position;

class Stance { // stand-in for std::unordered_map<LimbEnum, Position, EnumClassHash> (TypeDefs.hpp:91)
 public:
  Position &operator[](LimbEnum limb) {
    for (auto &kv : items_)
      if (kv.first == limb) return kv.second;
    items_.emplace_back(limb, Position());
    return items_.back().second;
  }
  size_t size() const { return items_.size(); }
  bool contains(LimbEnum limb) const {
    for (auto &kv : items_)
      if (kv.first == limb) return true;
    return false;
  }
  const Position &at(LimbEnum limb) const {
    for (auto &kv : items_)
      if (kv.first == limb) return kv.second;
    throw std::out_of_range("Stance::at");
  }
  // libstdc++ iteration order: most recently inserted first
  std::vector<LimbEnum> iterationOrder() const {
    std::vector<LimbEnum> o;
    for (auto it = items_.rbegin(); it != items_.rend(); ++it) o.push_back(it->first);
    return o;
  }

 private:
  std::vector<std::pair<LimbEnum, Position>> items_;
};

typedef std::vector<std::array<double, 2>> Polygon; // support region vertices, counter-clockwise

class PoseOptimizationSQP {
 public:
  typedef std::array<double, 4> LimbLengths;

  explicit PoseOptimizationSQP(std::shared_ptr<qlamd::Context> ctx) : ctx_(std::move(ctx)) {
    qlamd_pose_default_params(&params_);
  }
  void setStance(const Stance &stance) { stance_ = stance; }
  void setSupportStance(const Stance &supportStance) { supportStance_ = supportStance; }
  void setNominalStance(const Stance &nominalStanceInBaseFrame) { nominal_ = nominalStanceInBaseFrame; }
  void setSupportRegion(const Polygon &supportRegion) { region_ = supportRegion; }
  void setLimbLengthConstraints(const LimbLengths &minLimbLenghts, const LimbLengths &maxLimbLenghts) {
    (void)minLimbLenghts; // lower bounds are never enforced by the reference's SQP (sequencequadraticproblemsolver.cpp:37,85)
    maxLen_ = maxLimbLenghts;
  }
  void setCenterOfMass(const Position &centerOfMassInBaseFrame) { com_ = centerOfMassInBaseFrame; }
  qlamd_pose_params &parameters() { return params_; }

  bool optimize(Pose &pose) {
    double stance[12] = {0}, nominal[12] = {0}, polygon[8] = {0}, maxlen[4], pose_in[7], pose_out[7];
    uint8_t mask[4] = {0, 0, 0, 0};
    for (int l = 0; l < 4; ++l) {
      const LimbEnum limb = static_cast<LimbEnum>(l);
      maxlen[l] = maxLen_[l];
      if (!stance_.contains(limb)) continue;
      mask[l] = 1;
      for (int i = 0; i < 3; ++i) {
        stance[3 * l + i] = stance_.at(limb)(i);
        nominal[3 * l + i] = nominal_.contains(limb) ? nominal_.at(limb)(i) : 0.0;
      }
    }
    // checkSupportRegion, PoseOptimizationBase.cpp:42-50: default region = support-stance footprints
    Polygon region = region_;
    if (region.empty())
      for (LimbEnum limb : supportStance_.iterationOrder()) region.push_back({supportStance_.at(limb)(0), supportStance_.at(limb)(1)});
    if (region.size() < 3 || region.size() > 4) return false;
    for (size_t k = 0; k < region.size(); ++k) { polygon[2 * k] = region[k][0]; polygon[2 * k + 1] = region[k][1]; }
    const int32_t nv = static_cast<int32_t>(region.size());
    // leg order = iteration order of the stance map, padded with the absent limbs
    qlamd_pose_params prm = params_;
    int k = 0;
    bool used[4] = {false, false, false, false};
    for (LimbEnum limb : stance_.iterationOrder()) { prm.leg_order[k++] = static_cast<int>(limb); used[static_cast<int>(limb)] = true; }
    for (int l = 0; l < 4; ++l)
      if (!used[l]) prm.leg_order[k++] = l;
    for (int i = 0; i < 3; ++i) pose_in[i] = pose.position(i);
    for (int i = 0; i < 4; ++i) pose_in[3 + i] = pose.rotation.q[i];
    const double com[3] = {com_(0), com_(1), com_(2)};
    qlamd_pose_batch in = {stance, mask, nominal, polygon, &nv, com, maxlen, pose_in};
    int32_t status = -1;
    const int rc = qlamd_pose_sqp_batch(ctx_->get(), &prm, &in, 1, pose_out, &iterations_, &status, QLAMD_MEM_HOST, nullptr);
    if (rc != QLAMD_OK || status != QLAMD_STATUS_OK) return false;
    pose.position = Position(pose_out[0], pose_out[1], pose_out[2]);
    pose.rotation = qlamd::RotationQuaternion(pose_out[3], pose_out[4], pose_out[5], pose_out[6]);
    return true;
  }
  size_t getNumberOfIterations() const { return static_cast<size_t>(iterations_); }

 private:
  std::shared_ptr<qlamd::Context> ctx_;
  qlamd_pose_params params_;
  Stance stance_, supportStance_, nominal_;
  Polygon region_;
  LimbLengths maxLen_{{0.565, 0.565, 0.565, 0.565}};
  Position com_;
  int32_t iterations_ = 0;
};

} // namespace free_gait
