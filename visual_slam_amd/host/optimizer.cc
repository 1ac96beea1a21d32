#include "optimizer.h"

#include <cmath>

namespace visual_slam {
namespace core {

Problem::~Problem()
{
    if (handle) vs_ba_destroy(handle);
}

namespace {

// Eigen::Quaternionf::toRotationMatrix()
void quat_to_matrix(const std::array<float, 4> &q, float R[9])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float tx = 2.f * x, ty = 2.f * y, tz = 2.f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.f - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.f - (txx + tyy);
}

// Eigen::Matrix3f::eulerAngles(0, 1, 2)
void euler_angles_012(const float R[9], float e[3])
{
    const float pi = 3.14159265358979323846f;
    float r0 = std::atan2(R[5], R[8]);
    const float c2 = std::sqrt(R[0] * R[0] + R[1] * R[1]);
    float r1;
    if (r0 > 0.f) {
        r0 -= pi;
        r1 = std::atan2(-R[2], -c2);
    } else {
        r1 = std::atan2(-R[2], c2);
    }
    const float s1 = std::sin(r0), c1 = std::cos(r0);
    const float r2 = std::atan2(s1 * R[6] - c1 * R[3], c1 * R[4] - s1 * R[7]);
    e[0] = -r0; e[1] = -r1; e[2] = -r2;
}

std::array<float, 4> qmul(const std::array<float, 4> &a, const std::array<float, 4> &b)
{
    return { { a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
               a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
               a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
               a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1] } };
}

}  // namespace

// optimizer.cc:14-17
void Optimizer::addPose(const transform::Rigid3f pose) { poses_.emplace_back(FromPose(pose)); }

// optimizer.cc:19-28
void Optimizer::addPoints(const std::vector<TrackedPoint *> &points)
{
    for (auto p : points) points_.push_back({ { p->Pw[0], p->Pw[1], p->Pw[2] } });
}

// optimizer.cc:30-36
void Optimizer::addKeys(const std::vector<Point2i> &points)
{
    std::vector<std::array<double, 2>> keys;
    for (auto p : points) keys.push_back({ { (double)p.x, (double)p.y } });
    keys_.push_back(keys);
}

// optimizer.cc:38-45
PoseData Optimizer::FromPose(const transform::Rigid3f &pose)
{
    float R[9], rpy[3];
    quat_to_matrix(pose.rotation, R);
    euler_angles_012(R, rpy);
    return PoseData{ { { pose.translation[0], pose.translation[1], pose.translation[2] } },
                     { { rpy[0], rpy[1], rpy[2], 0.0 } } };
}

// optimizer.cc:60-145: the same edge set, constancy and loss placement; the residual blocks go to
// the device problem instead of a ceres::Problem.
void Optimizer::addReprojectionEdges(Problem &problem, Matrix3f K, std::list<std::vector<int>> &tracks,
                                     bool fix_camera_or_point)
{
    if (tracks.size() < 2) return;                       // optimizer.cc:62-63
    problem_ = &problem;
    if (points_.empty()) {                               // size_t underflow in the reference
        problem.status = VS_ERR_INVALID;
        problem.error = "addReprojectionEdges with no points";
        return;
    }
    std::vector<int32_t> cam, pt;
    std::vector<double> uv;
    for (size_t i = 0; i < points_.size() - 1; i++) {    // optimizer.cc:102: last point skipped
        int frame_id = 0;
        for (auto &track : tracks) {                     // optimizer.cc:119
            const int cur_idx = track[i];
            if (cur_idx != -1) {                         // optimizer.cc:124
                const auto &k = keys_[frame_id][cur_idx];
                cam.push_back(frame_id); pt.push_back((int32_t)i);
                uv.push_back(k[0]); uv.push_back(k[1]);
            }
            frame_id++;
        }
    }
    std::vector<double> t(3 * poses_.size()), rpy(3 * poses_.size()), xyz(3 * points_.size());
    for (size_t c = 0; c < poses_.size(); c++)
        for (int k = 0; k < 3; k++) { t[3 * c + k] = poses_[c].translation[k]; rpy[3 * c + k] = poses_[c].rotation[k]; }
    for (size_t j = 0; j < points_.size(); j++)
        for (int k = 0; k < 3; k++) xyz[3 * j + k] = points_[j][k];
    if (problem.handle) { vs_ba_destroy(problem.handle); problem.handle = nullptr; }
    int rc = vs_ba_create(&problem.handle, (int32_t)poses_.size(), (int32_t)points_.size(), (int64_t)cam.size(),
                          (double)K[0], (double)K[4], (double)K[2], (double)K[5]);   // float -> double
    if (rc == VS_OK) rc = vs_ba_set_cameras(problem.handle, t.data(), rpy.data());
    if (rc == VS_OK) rc = vs_ba_set_points(problem.handle, xyz.data());
    if (rc == VS_OK) rc = vs_ba_set_fixed(problem.handle, fix_camera_or_point ? 1 : 0);   // :134-137
    if (rc == VS_OK) rc = vs_ba_set_huber_first(problem.handle, 100.0);                   // :69,128
    if (rc == VS_OK) rc = vs_ba_set_observations(problem.handle, cam.data(), pt.data(), uv.data());
    problem.status = rc;
    problem.points_constant = fix_camera_or_point;
    if (rc != VS_OK) problem.error = vs_last_error_string();
}

// optimizer.cc:47-49 is empty; here: ceres::Solve(options{max_num_iterations = 100}) of
// tracking.cc:101-110, then the parameter blocks are read back.
bool Optimizer::solve()
{
    if (!problem_ || !problem_->handle || problem_->status != VS_OK) return false;
    vs_lm_options o;
    vs_lm_default_options(&o);
    int rc = vs_ba_solve(problem_->handle, &o, &summary_);
    // constant points (tracking.cc:99, solveCamera) are not read back: the solve cannot have moved them
    const bool read_points = !problem_->points_constant;
    std::vector<double> t(3 * poses_.size()), rpy(3 * poses_.size()), xyz(read_points ? 3 * points_.size() : 0);
    if (rc == VS_OK) rc = vs_ba_get_cameras(problem_->handle, t.data(), rpy.data());
    if (rc == VS_OK && read_points) rc = vs_ba_get_points(problem_->handle, xyz.data());
    problem_->status = rc;
    if (rc != VS_OK) { problem_->error = vs_last_error_string(); return false; }
    for (size_t c = 0; c < poses_.size(); c++)
        for (int k = 0; k < 3; k++) { poses_[c].translation[k] = t[3 * c + k]; poses_[c].rotation[k] = rpy[3 * c + k]; }
    if (read_points)
        for (size_t j = 0; j < points_.size(); j++)
            for (int k = 0; k < 3; k++) points_[j][k] = xyz[3 * j + k];
    return true;
}

// optimizer.cc:51-59
transform::Rigid3f Optimizer::getPose(const int id) const
{
    const auto &pose = poses_[id];
    const float r = (float)pose.rotation[0], p = (float)pose.rotation[1], y = (float)pose.rotation[2];
    const std::array<float, 4> qx{ { std::cos(r * 0.5f), std::sin(r * 0.5f), 0.f, 0.f } };
    const std::array<float, 4> qy{ { std::cos(p * 0.5f), 0.f, std::sin(p * 0.5f), 0.f } };
    const std::array<float, 4> qz{ { std::cos(y * 0.5f), 0.f, 0.f, std::sin(y * 0.5f) } };
    transform::Rigid3f out;
    out.rotation = qmul(qmul(qz, qy), qx);
    out.translation = { { (float)pose.translation[0], (float)pose.translation[1], (float)pose.translation[2] } };
    return out;
}

// optimizer.cc:187-190
std::vector<std::array<double, 3>> &Optimizer::getPoints() { return points_; }

}  // namespace core
}  // namespace visual_slam
