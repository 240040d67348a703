// free_gait::PoseOptimizationBase: the setters shared by the SQP, the QP and the constraints checker
// (free_gait_core/include/free_gait_core/pose_optimization/PoseOptimizationBase.hpp,
//  free_gait_core/src/pose_optimization/PoseOptimizationBase.cpp:21-60), plus the marshalling of one problem
// into the [1][k] arrays of qlamd_pose_batch.
// `Stance` keeps the insertion order the way the reference's unordered_map does under libstdc++
// (iteration = reverse insertion order), which fixes the leg order handed to the device.
#pragma once

#include <memory>
#include <stdexcept>
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

class PoseOptimizationBase {
 public:
  typedef std::array<double, 4> LimbLengths;

  explicit PoseOptimizationBase(std::shared_ptr<qlamd::Context> ctx) : ctx_(std::move(ctx)) {
    qlamd_pose_default_params(&params_);
  }
  void setStance(const Stance &stance) { stance_ = stance; }
  void setSupportStance(const Stance &supportStance) { supportStance_ = supportStance; }
  void setNominalStance(const Stance &nominalStanceInBaseFrame) { nominal_ = nominalStanceInBaseFrame; }
  void setSupportRegion(const Polygon &supportRegion) { region_ = supportRegion; }
  void setLimbLengthConstraints(const LimbLengths &minLimbLenghts, const LimbLengths &maxLimbLenghts) {
    minLen_ = minLimbLenghts; // only the constraints checker reads the lower bounds (sequencequadraticproblemsolver.cpp:37,85)
    maxLen_ = maxLimbLenghts;
  }
  void setCenterOfMass(const Position &centerOfMassInBaseFrame) { com_ = centerOfMassInBaseFrame; }
  qlamd_pose_params &parameters() { return params_; }

 protected:
  struct Marshalled {
    double stance[12], nominal[12], polygon[8], maxlen[4], minlen[4], pose[7], com[3];
    uint8_t mask[4];
    int32_t nv;
    qlamd_pose_params prm;
    qlamd_pose_batch batch() const { return {stance, mask, nominal, polygon, &nv, com, maxlen, pose}; }
  };

  bool marshal(const Pose &pose, Marshalled &m) const {
    for (int l = 0; l < 4; ++l) {
      const LimbEnum limb = static_cast<LimbEnum>(l);
      m.maxlen[l] = maxLen_[l];
      m.minlen[l] = minLen_[l];
      m.mask[l] = 0;
      for (int i = 0; i < 3; ++i) m.stance[3 * l + i] = m.nominal[3 * l + i] = 0.0;
      if (!stance_.contains(limb)) continue;
      m.mask[l] = 1;
      for (int i = 0; i < 3; ++i) {
        m.stance[3 * l + i] = stance_.at(limb)(i);
        m.nominal[3 * l + i] = nominal_.contains(limb) ? nominal_.at(limb)(i) : 0.0;
      }
    }
    // checkSupportRegion, PoseOptimizationBase.cpp:52-58: default region = support-stance footprints
    Polygon region = region_;
    if (region.empty())
      for (LimbEnum limb : supportStance_.iterationOrder()) region.push_back({supportStance_.at(limb)(0), supportStance_.at(limb)(1)});
    if (region.size() < 3 || region.size() > 4) return false;
    for (int k = 0; k < 8; ++k) m.polygon[k] = 0.0;
    for (size_t k = 0; k < region.size(); ++k) { m.polygon[2 * k] = region[k][0]; m.polygon[2 * k + 1] = region[k][1]; }
    m.nv = static_cast<int32_t>(region.size());
    // leg order = iteration order of the stance map, padded with the absent limbs
    m.prm = params_;
    int k = 0;
    bool used[4] = {false, false, false, false};
    for (LimbEnum limb : stance_.iterationOrder()) { m.prm.leg_order[k++] = static_cast<int>(limb); used[static_cast<int>(limb)] = true; }
    for (int l = 0; l < 4; ++l)
      if (!used[l]) m.prm.leg_order[k++] = l;
    for (int i = 0; i < 3; ++i) { m.pose[i] = pose.position(i); m.com[i] = com_(i); }
    for (int i = 0; i < 4; ++i) m.pose[3 + i] = pose.rotation.q[i];
    return true;
  }

  std::shared_ptr<qlamd::Context> ctx_;
  qlamd_pose_params params_;
  Stance stance_, supportStance_, nominal_;
  Polygon region_;
  LimbLengths minLen_{{0.2, 0.2, 0.2, 0.2}}, maxLen_{{0.565, 0.565, 0.565, 0.565}};
  Position com_;
};

} // namespace free_gait
