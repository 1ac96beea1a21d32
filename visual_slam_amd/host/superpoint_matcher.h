// Drop-in for the matcher half of the reference's core::SuperpointFrontend
// (src/core/superpoint_frontend.h:28-35): twoWayMatching and getStereoMatching with the
// reference's argument order and result type, running on the MI355X through include/vslam_hip.h.
// Keypoint extraction (getKeyPoints) is outside this path.
#ifndef VSLAM_HOST_SUPERPOINT_MATCHER_H_
#define VSLAM_HOST_SUPERPOINT_MATCHER_H_

#include <string>
#include <tuple>
#include <vector>

#include "vslam_hip.h"
#include "vslam_types.h"

namespace visual_slam {
namespace core {

class SuperpointFrontend {
public:
    typedef std::vector<std::tuple<int, int, float>> matching_info;   // superpoint_frontend.h:22
    // nn_thresh: config superpoint.nn_thresh (config/config.yaml:4).  ratio >= 1 disables the
    // ratio test (the reference has none).
    explicit SuperpointFrontend(float nn_thresh = 2.f, float ratio = 1.f) : nn_thresh_(nn_thresh), ratio_(ratio) {}
    matching_info getStereoMatching(const std::vector<Point2i> &left_points, const DescMat &left_descs,
                                    const std::vector<Point2i> &right_points, const DescMat &right_descs);
    matching_info twoWayMatching(const std::vector<Point2i> &keys0, const DescMat &desc0,
                                 const std::vector<Point2i> &keys1, const DescMat &desc1);
    int status() const { return status_; }
    const std::string &error() const { return error_; }

private:
    float nn_thresh_, ratio_;
    int status_ = VS_OK;
    std::string error_;
};

}  // namespace core
}  // namespace visual_slam
#endif
