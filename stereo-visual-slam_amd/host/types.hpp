// types.hpp -- host-side value types of the C++ mirror.
//
// The reference keeps its state in Feature / Frame / Observation / Landmark records that hold OpenCV and Sophus
// objects (/root/reference/include/stereo_visual_slam_main/types_def.hpp:17-121).  OpenCV, Sophus and Eigen do not
// exist in this build, so the records below carry layout-compatible PODs under the SAME member names
// (keypoint_.pt.x, pt_3d_, T_c_w_, features_, observations_, is_inlier, reliable_depth_, ...): code written against
// the reference's types reads the same.  KeyPoint == cv::KeyPoint == vslam_keypoint (28 B), DMatch == cv::DMatch ==
// vslam_dmatch (16 B), SE3 == Sophus::SE3d memory order (unit quaternion x,y,z,w + translation).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vslam_hip.h"
#include "../csrc/se3_device.h"

namespace vslam {

struct Point2f { float x = 0, y = 0; };
struct Point3f {
    float x = 0, y = 0, z = 0;
    Point3f() = default;
    Point3f(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == sizeof(vslam_keypoint), "KeyPoint must match the C-ABI layout");
struct DMatch { int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 0; };
static_assert(sizeof(DMatch) == sizeof(vslam_dmatch), "DMatch must match the C-ABI layout");

// one 256-bit rBRIEF descriptor (the reference holds a 1x32 cv::Mat row header)
struct Descriptor { std::array<uint8_t, 32> bytes{}; };

// N x 32 descriptor matrix (stands in for the CV_8U cv::Mat); contiguous rows
struct DescriptorMat {
    std::vector<uint8_t> data;
    int rows = 0;
    static constexpr int cols = 32;
    void push_back(const Descriptor& d) { data.insert(data.end(), d.bytes.begin(), d.bytes.end()); ++rows; }
    Descriptor row(int i) const { Descriptor d; std::memcpy(d.bytes.data(), data.data() + (size_t)i * 32, 32); return d; }
    void resize(int n) { data.resize((size_t)n * 32); rows = n; }
    void clear() { data.clear(); rows = 0; }
};

struct Image {
    int cols = 0, rows = 0;
    std::vector<uint8_t> data; // tight rows
    bool empty() const { return data.empty(); }
};

// rigid transform; same operations the reference uses from Sophus::SE3d
class SE3 {
public:
    SE3() : d_{0, 0, 0, 1, 0, 0, 0} {}
    explicit SE3(const double* p) { std::memcpy(d_, p, sizeof(d_)); }
    const double* data() const { return d_; }
    double* data() { return d_; }
    SE3 inverse() const { SE3 r; se3::inverse(d_, r.d_); return r; }
    SE3 operator*(const SE3& o) const { SE3 r; se3::mul(d_, o.d_, r.d_); return r; }
    std::array<double, 3> operator*(const std::array<double, 3>& p) const { std::array<double, 3> o; se3::act(d_, p.data(), o.data()); return o; }
    std::array<double, 6> log() const { std::array<double, 6> x; se3::log(d_, x.data()); return x; }
    static SE3 exp(const std::array<double, 6>& x) { SE3 r; se3::exp(x.data(), r.d_); return r; }
    double angleY() const { return se3::angle_y(d_); }
    std::array<double, 9> rotationMatrix() const { std::array<double, 9> R; se3::rotmat(d_, R.data()); return R; }
    std::array<double, 3> translation() const { return {d_[4], d_[5], d_[6]}; }
private:
    double d_[7];
};
inline double norm6(const std::array<double, 6>& v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); }

struct Feature {
    int feature_id_ = 0;
    int frame_id_ = 0;
    int landmark_id_ = -1;
    KeyPoint keypoint_;
    Descriptor descriptor_;
    bool is_inlier = false;
    Feature() = default;
    Feature(int feature_id, int frame_id, const KeyPoint& kp, const Descriptor& d) : feature_id_(feature_id), frame_id_(frame_id), keypoint_(kp), descriptor_(d) {}
};

struct Frame {
    int frame_id_ = 0;
    Image left_img_, right_img_;
    std::vector<float> disparity_; // h x w f32, -1 = invalid (cv::Mat disparity_, types_def.hpp)
    SE3 T_c_w_;
    bool is_keyframe_ = false;
    int keyframe_id_ = 0;
    std::vector<Feature> features_;
    // KITTI-00 intrinsics (types_def.hpp:53-54)
    double fx_ = 718.856, fy_ = 718.856, cx_ = 607.1928, cy_ = 185.2157;
    double b_ = 0.573;
    void fill_frame(const SE3& T_c_w, bool is_keyframe, int keyframe_id) {
        T_c_w_ = T_c_w; is_keyframe_ = is_keyframe;
        if (is_keyframe) keyframe_id_ = keyframe_id;
    }
};

struct Observation {
    int keyframe_id_, feature_id_;
    bool to_delete = false;
    Observation(int keyframe_id, int feature_id) : keyframe_id_(keyframe_id), feature_id_(feature_id) {}
};

struct Landmark {
    int landmark_id_ = 0;
    Point3f pt_3d_;
    Descriptor descriptor_;
    int observed_times_ = 1;
    std::vector<Observation> observations_;
    bool is_inlier = true;
    bool reliable_depth_ = false;
    Landmark() = default;
    Landmark(int id, const Point3f& p, const Descriptor& d, bool reliable, const Observation& o) : landmark_id_(id), pt_3d_(p), descriptor_(d), reliable_depth_(reliable) { observations_.push_back(o); }
};

} // namespace vslam
