// Minimal stand-ins for the Eigen / OpenCV value types the reference's interfaces use, so that
// the adapters below keep the reference's signatures without those libraries.  A build that has
// Eigen and OpenCV maps them one to one:
//   transform::Rigid3f  <->  visual_slam::transform::Rigid3f   (src/transform/rigid_transform.h:31-104)
//   Point2i             <->  cv::Point
//   Matrix3f (row-major)<->  Eigen::Matrix3f
//   DescMat             <->  cv::Mat of type CV_32FC1, N x 256, continuous
#ifndef VSLAM_HOST_TYPES_H_
#define VSLAM_HOST_TYPES_H_

#include <array>
#include <cstddef>

namespace visual_slam {

struct Point2i {
    int x = 0, y = 0;
};

using Matrix3f = std::array<float, 9>;   // row-major

struct DescMat {                         // view of an N x 256 float matrix
    const float *data = nullptr;
    int rows = 0;
};

namespace transform {
struct Rigid3f {
    std::array<float, 3> translation{ { 0.f, 0.f, 0.f } };
    std::array<float, 4> rotation{ { 1.f, 0.f, 0.f, 0.f } };   // quaternion (w, x, y, z)
};
}  // namespace transform

namespace core {
struct TrackedPoint {                    // src/core/tracked_point.h: only Pw is read here
    std::array<float, 3> Pw{ { 0.f, 0.f, 0.f } };
};
}  // namespace core
}  // namespace visual_slam
#endif
