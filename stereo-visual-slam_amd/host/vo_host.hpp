// vo_host.hpp -- C++ mirror of vslam::VO (/root/reference/include/stereo_visual_slam_main/visual_odometry.hpp:27-185)
// on top of the C-ABI.  Same public state and method names; OpenCV/ROS types are replaced by the PODs in types.hpp and
// every arithmetic-heavy step is one C-ABI call into libvslam_hip.so.  Stereo depth has two sources: the reference's
// own (dense SGBM disparity + Frame::find_3d, depth_source_ = DepthSGBM) and the north_star stage (right-image ORB,
// L/R cross-check match, rectified DLT: depth_source_ = DepthStereoMatch, what the headline bench measures).  The defaults are the
// reference's own algorithm (SGBM depth + RANSAC pose).
#pragma once
#include <string>
#include <vector>

#include "map_host.hpp"
#include "types.hpp"

namespace vslam {

enum TrackState { Init, Track, Lost };
enum DepthSource { DepthStereoMatch = 0, DepthSGBM = 1 };
enum PnpMode { PnpMotionOnlyLM = 0, PnpRansac = 1 }; // north_star robust LM, or the reference's RANSAC control flow + LS refinement

// replaces cv::imread on `dataset + "image_0/%06d.png"` (visual_odometry.cpp:37-68): 8-bit gray PNG (zlib), or binary PGM (P5)
struct ImageSource {
    std::string dataset_;
    explicit ImageSource(std::string dataset) : dataset_(std::move(dataset)) {}
    int read(int id, Image& left, Image& right) const;
    static int read_pgm(const std::string& path, Image& img);
    static int read_png(const std::string& path, Image& img);
};

class VO {
public:
    Frame frame_last_;
    Frame frame_current_;
    Map& my_map_;
    ImageSource source_;
    vslam_ctx* ctx_;
    int num_inliers_ = 0;
    SE3 T_c_l_;
    SE3 T_c_w_;
    int seq_ = 1;
    TrackState state_ = Init;
    int num_lost_ = 0;
    int curr_keyframe_id_ = 0;
    int curr_landmark_id_ = 0;
    int pnp_iterations_ = 10;         // motion-only LM stage (PnpMotionOnlyLM)
    int ransac_refine_iterations_ = 0; // PnpRansac: 0 = return the best RANSAC model itself, as OpenCV 3.2.0's solvePnPRansac does (it computes the refined
                                       // pose and then assigns _local_model, solvepnp.cpp); > 0 = return the pose refined on the inliers (3.4.2+ behaviour)
    DepthSource depth_source_ = DepthSGBM;   // the reference's algorithm by default (visual_odometry.cpp:159-217, :277)
    PnpMode pnp_mode_ = PnpRansac;
    // diagnostics of the latest tracking() call, for the CPU-path vs GPU-path trace comparison (not in the reference)
    int last_num_detected_ = 0, last_num_matches_ = 0;
    uint64_t last_match_hash_ = 0; // FNV-1a over (queryIdx, trainIdx, distance) of the gated frame-to-frame matches

    VO(std::string dataset, vslam_ctx* ctx, Map& map);
    ~VO();
    VO(const VO&) = delete;
    VO& operator=(const VO&) = delete;

    int read_img(int id, Image& left_img, Image& right_img) { return source_.read(id, left_img, right_img); }
    int feature_detection(const Image& img, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors);
    void adaptive_non_maximal_suppresion(std::vector<KeyPoint>& keypoints, const int num);
    int feature_matching(const DescriptorMat& descriptors_1, const DescriptorMat& descriptors_2, std::vector<DMatch>& feature_matches);
    int disparity_map(const Frame& frame, std::vector<float>& disparity);
    std::vector<bool> set_ref_3d_position(std::vector<Point3f>& pts_3d, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors, Frame& frame);
    void motion_estimation(Frame& frame);
    bool check_motion_estimation();
    bool insert_key_frame(bool check, std::vector<Point3f>& pts_3d, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors);
    void move_frame() { frame_last_ = frame_current_; }
    bool initialization();
    bool tracking(bool& if_insert_keyframe);
    bool pipeline(bool& if_insert_keyframe);
};

} // namespace vslam
