#include "superpoint_matcher.h"

namespace visual_slam {
namespace core {

static SuperpointFrontend::matching_info pack(int n, const std::vector<int32_t> &i, const std::vector<int32_t> &j,
                                              const std::vector<float> &d)
{
    SuperpointFrontend::matching_info out;
    for (int k = 0; k < n; k++) out.emplace_back(i[k], j[k], d[k]);
    return out;
}

// superpoint_frontend.cc:163-211; as there, only the SIZES of keys0/keys1 are used.
SuperpointFrontend::matching_info SuperpointFrontend::twoWayMatching(const std::vector<Point2i> &keys0,
                                                                     const DescMat &desc0,
                                                                     const std::vector<Point2i> &keys1,
                                                                     const DescMat &desc1)
{
    const int n0 = (int)keys0.size(), n1 = (int)keys1.size();
    std::vector<int32_t> oi(n0 > 0 ? n0 : 1), oj(n0 > 0 ? n0 : 1);
    std::vector<float> od(n0 > 0 ? n0 : 1);
    int32_t n = 0;
    status_ = vs_match_two_way(desc0.data, n0, desc1.data, n1, nn_thresh_, ratio_, oi.data(), oj.data(), od.data(), &n);
    if (status_ != VS_OK) { error_ = vs_last_error_string(); return {}; }
    return pack(n, oi, oj, od);
}

// superpoint_frontend.cc:110-161
SuperpointFrontend::matching_info SuperpointFrontend::getStereoMatching(const std::vector<Point2i> &left_points,
                                                                        const DescMat &left_descs,
                                                                        const std::vector<Point2i> &right_points,
                                                                        const DescMat &right_descs)
{
    const int n0 = (int)left_points.size(), n1 = (int)right_points.size();
    std::vector<int32_t> oi(n0 > 0 ? n0 : 1), oj(n0 > 0 ? n0 : 1);
    std::vector<float> od(n0 > 0 ? n0 : 1);
    int32_t n = 0;
    static_assert(sizeof(Point2i) == 2 * sizeof(int32_t), "Point2i is two packed ints");
    status_ = vs_match_stereo(left_descs.data, reinterpret_cast<const int32_t *>(left_points.data()), n0,
                              right_descs.data, reinterpret_cast<const int32_t *>(right_points.data()), n1,
                              nn_thresh_, ratio_, oi.data(), oj.data(), od.data(), &n);
    if (status_ != VS_OK) { error_ = vs_last_error_string(); return {}; }
    return pack(n, oi, oj, od);
}

}  // namespace core
}  // namespace visual_slam
