// stand-in for <ceres/rotation.h>: ceres::QuaternionRotatePoint as published in Ceres Solver 2.1.0, RESTATED FROM MEMORY
// (the library is neither under /root/reference nor installed): q = [w, x, y, z] is rescaled by 1/|q|, then the point is
// rotated with  p + 2 w (v x p) + 2 v x (v x p).  This one function is therefore NOT the reference's / Ceres' own code.
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include "ceres.h"
namespace ceres {
template <typename T>
inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3])
{
    T uv0 = q[2] * pt[2] - q[3] * pt[1];
    T uv1 = q[3] * pt[0] - q[1] * pt[2];
    T uv2 = q[1] * pt[1] - q[2] * pt[0];
    uv0 += uv0;
    uv1 += uv1;
    uv2 += uv2;
    result[0] = pt[0] + q[0] * uv0;
    result[1] = pt[1] + q[0] * uv1;
    result[2] = pt[2] + q[0] * uv2;
    result[0] += q[2] * uv2 - q[3] * uv1;
    result[1] += q[3] * uv0 - q[1] * uv2;
    result[2] += q[1] * uv1 - q[2] * uv0;
}
template <typename T>
inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3])
{
    using std::sqrt;
    const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
    UnitQuaternionRotatePoint(unit, pt, result);
}
} // namespace ceres
