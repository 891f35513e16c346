// stand-in for the (non-templated, old-style) sophus/se3.h the reference includes: construction from (R, t) and (q, t),
// product, inverse and the two accessors (test infrastructure only).  The rotation is kept as a unit quaternion, as that
// Sophus version does.
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>
namespace Sophus {
class SE3 {
  public:
    SE3() { t_.setZero(); }
    SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : q_(R), t_(t) { q_.normalize(); }
    SE3(const Eigen::Quaterniond &q, const Eigen::Vector3d &t) : q_(q), t_(t) { q_.normalize(); }
    Eigen::Matrix3d rotation_matrix() const { return q_.toRotationMatrix(); }
    Eigen::Vector3d translation() const { return t_; }
    const Eigen::Quaterniond &unit_quaternion() const { return q_; }
    SE3 operator*(const SE3 &o) const
    {
        Eigen::Quaterniond q = q_ * o.q_;
        q.normalize();
        return SE3(q, rotation_matrix() * o.t_ + t_);
    }
    SE3 inverse() const
    {
        const Eigen::Quaterniond qi = q_.conjugate();
        return SE3(qi, -(qi.toRotationMatrix() * t_));
    }

  private:
    Eigen::Quaterniond q_;
    Eigen::Vector3d t_;
};
} // namespace Sophus
