// Minimal value types for the host-side mirror of the reference's C++ interface.
// The reference uses kindr (Position, Force, Torque, RotationQuaternion, Pose) and Eigen;
// neither is available here, and the C-ABI takes plain doubles, so the mirror keeps
// PODs with the same names and the same (w,x,y,z) quaternion convention.
#pragma once

#include <array>
#include <cstddef>
#include <memory>
#include <stdexcept>
#include <string>

#include "qlamd.h"

namespace qlamd {

struct Vector3 {
  double v[3] = {0.0, 0.0, 0.0};
  Vector3() = default;
  Vector3(double x, double y, double z) : v{x, y, z} {}
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double x() const { return v[0]; }
  double y() const { return v[1]; }
  double z() const { return v[2]; }
};
typedef Vector3 Position, Force, Torque, LinearVelocity, LocalAngularVelocity, Vector;

struct RotationQuaternion { // (w, x, y, z), base -> world, as kindr::RotationQuaternion
  double q[4] = {1.0, 0.0, 0.0, 0.0};
  RotationQuaternion() = default;
  RotationQuaternion(double w, double x, double y, double z) : q{w, x, y, z} {}
  double w() const { return q[0]; }
  double x() const { return q[1]; }
  double y() const { return q[2]; }
  double z() const { return q[3]; }
};

struct Pose {
  Position position;
  RotationQuaternion rotation;
  Pose() = default;
  Pose(const Position &p, const RotationQuaternion &r) : position(p), rotation(r) {}
  const Position &getPosition() const { return position; }
  const RotationQuaternion &getRotation() const { return rotation; }
};

// limb order of the reference, quadruped_model/include/quadruped_model/QuadrupedModel.hpp:47-53
enum class LimbEnum { LF_LEG = 0, RF_LEG = 1, RH_LEG = 2, LH_LEG = 3 };

// One opaque device context shared by the mirror classes (RAII over qlamd_context_create).
class Context {
 public:
  explicit Context(const qlamd_balance_params &params, int device = 0) {
    const int rc = qlamd_context_create(&params, nullptr, device, &ctx_);
    if (rc != QLAMD_OK) throw std::runtime_error(std::string("qlamd_context_create: ") + qlamd_strerror(rc));
  }
  ~Context() { qlamd_context_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  qlamd_context *get() const { return ctx_; }

 private:
  qlamd_context *ctx_ = nullptr;
};

} // namespace qlamd
