// Drop-in for the reference's core::Optimizer (src/core/optimizer.h:24-36): same public methods,
// same argument meaning.  solve() -- an empty stub in the reference (src/core/optimizer.cc:47-49)
// -- runs the MI355X bundle-adjustment path through the C ABI of include/vslam_hip.h, replacing
// the ceres::Problem / ceres::Solve pair the reference's callers build themselves
// (src/core/tracking.cc:97-110, 139-152).
#ifndef VSLAM_HOST_OPTIMIZER_H_
#define VSLAM_HOST_OPTIMIZER_H_

#include <array>
#include <list>
#include <string>
#include <vector>

#include "vslam_hip.h"
#include "vslam_types.h"

namespace visual_slam {
namespace core {

struct PoseData {                        // src/core/optimizer.h:17-22
    std::array<double, 3> translation;
    std::array<double, 4> rotation;      // only [0..2] = (roll, pitch, yaw) are used
};

// Takes the place of the ceres::Problem argument of addReprojectionEdges: owns the device-side
// problem the edges are added to.
class Problem {
public:
    Problem() = default;
    ~Problem();
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;
    vs_ba_handle handle = nullptr;
    int status = VS_OK;                  // last vs_status; message in error
    bool points_constant = false;        // the edges were added with every point constant: a solve leaves points_ as they are
    std::string error;
};

class Optimizer {
public:
    Optimizer() {}
    void addPose(const transform::Rigid3f pose);
    void addPoints(const std::vector<TrackedPoint *> &points);
    void addKeys(const std::vector<Point2i> &points);
    // Runs LM (reference options: 100 iterations, Ceres defaults) and reads the result back into
    // poses_/points_.  Returns false (poses untouched) when the device path reports an error.
    bool solve();
    void addReprojectionEdges(Problem &problem, Matrix3f K, std::list<std::vector<int>> &tracks,
                              bool fix_camera_or_point);
    transform::Rigid3f getPose(const int id) const;
    std::vector<std::array<double, 3>> &getPoints();
    const vs_lm_summary &summary() const { return summary_; }

private:
    PoseData FromPose(const transform::Rigid3f &pose);
    std::vector<PoseData> poses_;
    std::vector<std::array<double, 3>> points_;
    std::vector<std::vector<std::array<double, 2>>> keys_;
    Problem *problem_ = nullptr;
    vs_lm_summary summary_{};
};

}  // namespace core
}  // namespace visual_slam
#endif
