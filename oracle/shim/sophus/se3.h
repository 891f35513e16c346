// stand-in for the (non-templated, old-style) sophus/se3.h the reference includes: the two accessors utils.hpp calls
// (test infrastructure only)
#pragma once
#include <Eigen/Core>
namespace Sophus {
class SE3 {
  public:
    SE3() { R_.setIdentity(); t_.setZero(); }
    SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : R_(R), t_(t) {}
    Eigen::Matrix3d rotation_matrix() const { return R_; }
    Eigen::Vector3d translation() const { return t_; }

  private:
    Eigen::Matrix3d R_;
    Eigen::Vector3d t_;
};
} // namespace Sophus
