// TESTS ONLY -- the two Sophus classes the reference's pose-graph builders touch
// (kitti_surf.cpp:693, :793, :1035, :1067), over the mock in ../Eigen.
#pragma once
#include "../Eigen/Geometry"

namespace Sophus {

class SO3d {
 public:
  SO3d() {}
  explicit SO3d(const Eigen::Matrix3d& R) : q_(R) {}
  explicit SO3d(const Eigen::Quaterniond& q) : q_(q) {}
  const Eigen::Quaterniond& unit_quaternion() const { return q_; }
  Eigen::Matrix3d matrix() const { return q_.toRotationMatrix(); }
 private:
  Eigen::Quaterniond q_;
};

class SE3d {
 public:
  SE3d() {}
  SE3d(const Eigen::Quaterniond& q, const Eigen::Vector3d& t) : so3_(q), t_(t) {}
  SE3d(const Eigen::Matrix3d& R, const Eigen::Vector3d& t) : so3_(R), t_(t) {}
  Eigen::Matrix3d rotationMatrix() const { return so3_.matrix(); }
  const Eigen::Vector3d& translation() const { return t_; }
  const SO3d& so3() const { return so3_; }
 private:
  SO3d so3_;
  Eigen::Vector3d t_;
};

}  // namespace Sophus
