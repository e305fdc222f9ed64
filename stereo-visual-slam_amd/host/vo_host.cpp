// vo_host.cpp -- tracking state machine, keyframe policy and landmark bookkeeping of the reference's front-end
// (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:348-432, :491-706), re-implemented over the C-ABI.
// Host code stays on the CPU: it is O(N) association and policy; the kernels do the arithmetic.
#include "vo_host.hpp"
#include "ba_host.hpp"

#include <zlib.h>

#include <cstring>
#include <iterator>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <unordered_map>

namespace vslam {

namespace {
void check(int rc, const char* what) {
    if (rc != VSLAM_OK) throw std::runtime_error(std::string(what) + ": " + vslam_last_error());
}
inline uint64_t pixel_key(const Point2f& p) { // exact float equality of (x, y) as a hash key (visual_odometry.cpp:390)
    uint32_t a, b;
    const float x = p.x + 0.0f, y = p.y + 0.0f; // -0 == +0
    std::memcpy(&a, &x, 4); std::memcpy(&b, &y, 4);
    return ((uint64_t)a << 32) | b;
}
} // namespace

// ---------------------------------------------------------------------------------------------- images
int ImageSource::read_pgm(const std::string& path, Image& img) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::string magic;
    f >> magic;
    if (magic != "P5") return -1;
    auto next_int = [&]() {
        int v = -1;
        while (f) {
            f >> std::ws;
            if (f.peek() == '#') { std::string line; std::getline(f, line); continue; }
            f >> v; break;
        }
        return v;
    };
    const int w = next_int(), h = next_int(), maxv = next_int();
    if (w <= 0 || h <= 0 || maxv != 255) return -1;
    f.get(); // single whitespace after maxval
    img.cols = w; img.rows = h; img.data.resize((size_t)w * h);
    f.read(reinterpret_cast<char*>(img.data.data()), (std::streamsize)img.data.size());
    return f.gcount() == (std::streamsize)img.data.size() ? 0 : -1;
}

// 8-bit grayscale, non-interlaced PNG (the KITTI odometry gray sequences): chunk walk, zlib inflate, scanline unfiltering.
// Replaces cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE) for that file type (visual_odometry.cpp:49-50).
int ImageSource::read_png(const std::string& path, Image& img) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return -1;
    auto be32 = [&](size_t o) { return (uint32_t)file[o] << 24 | (uint32_t)file[o + 1] << 16 | (uint32_t)file[o + 2] << 8 | (uint32_t)file[o + 3]; };
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> idat;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o);
        if (o + 12 + (size_t)len > file.size()) return -1;
        const char* type = reinterpret_cast<const char*>(&file[o + 4]);
        const uint8_t* data = &file[o + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return -1;
            w = be32(o + 8); h = be32(o + 12);
            // bit depth 8, colour type 0 (gray), compression 0, filter 0, no interlace
            if (data[8] != 8 || data[9] != 0 || data[10] != 0 || data[11] != 0 || data[12] != 0) return -1;
        } else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 16384 || h > 16384 || idat.empty()) return -1;
    std::vector<uint8_t> raw((size_t)h * (w + 1));
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return -1;
    img.cols = (int)w; img.rows = (int)h; img.data.assign((size_t)w * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* in = &raw[(size_t)y * (w + 1)];
        uint8_t* out = &img.data[(size_t)y * w];
        const uint8_t* up = y ? out - w : nullptr;
        const int ft = in[0];
        for (uint32_t x = 0; x < w; ++x) {
            const int a = x ? out[x - 1] : 0, b = up ? up[x] : 0, c = (x && up) ? up[x - 1] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return -1;
            }
            out[x] = (uint8_t)(in[1 + x] + pred);
        }
    }
    return 0;
}

// dataset_/image_0/%06d.png (KITTI layout, visual_odometry.cpp:44-50); .pgm is accepted as a dependency-free alternative
int ImageSource::read(int id, Image& left, Image& right) const {
    char name[32];
    std::snprintf(name, sizeof(name), "%06d", id);
    const std::string l = dataset_ + "image_0/" + name, r = dataset_ + "image_1/" + name;
    const bool ok = (read_png(l + ".png", left) == 0 && read_png(r + ".png", right) == 0) ||
                    (read_pgm(l + ".pgm", left) == 0 && read_pgm(r + ".pgm", right) == 0);
    if (!ok) {
        std::cout << "Could not open or find the image" << std::endl; // visual_odometry.cpp:53-57
        return -1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- kernels behind methods
// visual_odometry.hpp:69 `VO(std::string dataset, ros::NodeHandle&, Map&)`: the node handle's place is taken by the GPU context, which also
// becomes the backend of the free optimize_map / optimize_pose_only functions (ba_host.hpp).  The LATEST VO binds (a second VO on another
// context must not silently optimise on the first one's), and a VO that goes away unbinds its context: the pointer never outlives its owner.
VO::VO(std::string dataset, vslam_ctx* ctx, Map& map) : my_map_(map), source_(std::move(dataset)), ctx_(ctx) {
    set_optimizer_backend(ctx);
}
VO::~VO() { clear_optimizer_backend(ctx_); }

int VO::feature_detection(const Image& img, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors) {
    if (img.empty()) { std::cout << "Could not open or find the image" << std::endl; return -1; } // :73-77
    const int cap = 4096;
    keypoints.resize(cap); descriptors.resize(cap);
    int n = 0;
    check(vslam_feature_detection(ctx_, img.data.data(), img.cols, img.rows, img.cols, reinterpret_cast<vslam_keypoint*>(keypoints.data()),
                                  descriptors.data.data(), cap, &n), "feature_detection");
    keypoints.resize((size_t)n); descriptors.resize(n);
    return 0; // the GUI calls (:88-91) are not part of the hot path
}

void VO::adaptive_non_maximal_suppresion(std::vector<KeyPoint>& keypoints, const int num) {
    int n = 0;
    check(vslam_anms(ctx_, reinterpret_cast<vslam_keypoint*>(keypoints.data()), (int)keypoints.size(), num, &n), "anms");
    keypoints.resize((size_t)n);
}

int VO::feature_matching(const DescriptorMat& d1, const DescriptorMat& d2, std::vector<DMatch>& feature_matches) {
    feature_matches.assign((size_t)std::max(d1.rows, 1), DMatch());
    int n = 0;
    const double frame_gap = frame_current_.frame_id_ - frame_last_.frame_id_; // :239
    check(vslam_feature_matching(ctx_, d1.data.data(), d1.rows, d2.data.data(), d2.rows, frame_gap, 1, reinterpret_cast<vslam_dmatch*>(feature_matches.data()), &n),
          "feature_matching");
    feature_matches.resize((size_t)n);
    return 0;
}

// VO::disparity_map (:159-174): StereoSGBM(0, 96, 9, 648, 2592, 1, 63, 10, 100, 32) + convertTo(CV_32F, 1/16) on the device
int VO::disparity_map(const Frame& frame, std::vector<float>& disparity) {
    const Image& l = frame.left_img_; const Image& r = frame.right_img_;
    disparity.assign((size_t)l.cols * l.rows, -1.0f);
    check(vslam_disparity_map(ctx_, l.data.data(), r.data.data(), l.cols, l.rows, l.cols, disparity.data(), nullptr, nullptr), "disparity_map");
    return 0;
}

// VO::set_ref_3d_position (:176-217).  DepthSGBM: the reference's own loop -- Frame::find_3d on the dense disparity map,
// keep 10 < Z < 400, reliable when Z < 40.  DepthStereoMatch: the depth source swapped for the north_star stage (right-image
// ORB, L/R cross-check match, rectified DLT).  Same outputs either way: pts_3d (world), keypoints/descriptors compacted in
// place, reliable flags.
std::vector<bool> VO::set_ref_3d_position(std::vector<Point3f>& pts_3d, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors, Frame& frame) {
    pts_3d.clear();
    std::vector<bool> reliable_depth;
    if (depth_source_ == DepthSGBM) {
        disparity_map(frame, frame.disparity_); // :377 / :505
        const size_t n = keypoints.size();
        std::vector<float> xyz(3 * std::max<size_t>(n, 1));
        std::vector<uint8_t> valid(std::max<size_t>(n, 1)), rel(std::max<size_t>(n, 1));
        if (n) check(vslam_find_3d_disparity(ctx_, reinterpret_cast<const vslam_keypoint*>(keypoints.data()), (int)n, frame.disparity_.data(), frame.left_img_.cols,
                                            frame.left_img_.rows, frame.left_img_.cols, frame.T_c_w_.data(), xyz.data(), valid.data(), rel.data(), nullptr), "find_3d");
        std::vector<KeyPoint> kept_k; DescriptorMat kept_d;
        for (size_t i = 0; i < n; ++i) {
            if (!valid[i]) continue;
            pts_3d.emplace_back(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
            kept_k.push_back(keypoints[i]);
            kept_d.push_back(descriptors.row((int)i));
            reliable_depth.push_back(rel[i] != 0);
        }
        keypoints.swap(kept_k); descriptors = kept_d;
        return reliable_depth;
    }
    std::vector<KeyPoint> kps_r; DescriptorMat desc_r;
    if (feature_detection(frame.right_img_, kps_r, desc_r) != 0 || descriptors.rows == 0) { keypoints.clear(); descriptors.clear(); return reliable_depth; }
    std::vector<DMatch> lr((size_t)descriptors.rows);
    int n = 0;
    check(vslam_feature_matching(ctx_, descriptors.data.data(), descriptors.rows, desc_r.data.data(), desc_r.rows, 1.0, 1, reinterpret_cast<vslam_dmatch*>(lr.data()), &n), "L/R matching");
    lr.resize((size_t)n);
    std::vector<float> uvL(2 * lr.size()), uvR(2 * lr.size()), xyz(3 * lr.size());
    std::vector<uint8_t> valid(lr.size()), rel(lr.size());
    for (size_t i = 0; i < lr.size(); ++i) {
        const KeyPoint& l = keypoints[(size_t)lr[i].queryIdx]; const KeyPoint& r = kps_r[(size_t)lr[i].trainIdx];
        uvL[2 * i] = l.pt.x; uvL[2 * i + 1] = l.pt.y; uvR[2 * i] = r.pt.x; uvR[2 * i + 1] = r.pt.y;
    }
    if (!lr.empty()) check(vslam_triangulate(ctx_, uvL.data(), uvR.data(), (int)lr.size(), frame.T_c_w_.data(), xyz.data(), valid.data(), rel.data(), nullptr), "triangulate");
    std::vector<KeyPoint> kept_k; DescriptorMat kept_d;
    for (size_t i = 0; i < lr.size(); ++i) { // matches are ascending in queryIdx: the surviving keypoints keep their order
        if (!valid[i]) continue;             // 10 < Z < 400 (:194)
        pts_3d.emplace_back(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        kept_k.push_back(keypoints[(size_t)lr[i].queryIdx]);
        kept_d.push_back(descriptors.row(lr[i].queryIdx));
        reliable_depth.push_back(rel[i] != 0); // Z < 40 (:201)
    }
    keypoints.swap(kept_k); descriptors = kept_d; // :213-214
    return reliable_depth;
}

void VO::motion_estimation(Frame& frame) {
    std::vector<float> pts3d, pts2d;
    std::vector<size_t> rows; // features with a live landmark
    for (size_t i = 0; i < frame.features_.size(); ++i) { // :260-270
        auto it = my_map_.landmarks_.find((unsigned long)frame.features_[i].landmark_id_);
        if (it == my_map_.landmarks_.end()) { std::cout << "No landmark associated!" << std::endl; continue; } // reference: .at(-1) throws (Q7)
        const Point3f& p = it->second.pt_3d_;
        pts3d.insert(pts3d.end(), {p.x, p.y, p.z});
        pts2d.insert(pts2d.end(), {frame.features_[i].keypoint_.pt.x, frame.features_[i].keypoint_.pt.y});
        rows.push_back(i);
    }
    num_inliers_ = 0;
    SE3 T = frame_last_.T_c_w_; // initial guess: previous pose (solvePnPRansac starts from scratch; LM needs a start)
    std::vector<uint8_t> inlier(rows.size(), 0);
    if (rows.size() >= 4) {
        int n_in = 0;
        if (pnp_mode_ == PnpRansac) // the reference's call: solvePnPRansac(..., false, 100, 4.0, 0.99, inliers) (:277)
            check(vslam_pnp_ransac(ctx_, pts3d.data(), pts2d.data(), (int)rows.size(), T.data(), 100, 4.0, 0.99, ransac_refine_iterations_, inlier.data(), &n_in, nullptr), "motion_estimation (RANSAC)");
        else
            check(vslam_pnp_motion_only(ctx_, pts3d.data(), pts2d.data(), (int)rows.size(), T.data(), pnp_iterations_, inlier.data(), &n_in, nullptr), "motion_estimation");
        num_inliers_ = n_in;
    }
    T_c_w_ = T; // :290-292
    for (size_t k = 0; k < rows.size(); ++k) frame.features_[rows[k]].is_inlier = inlier[k] != 0; // :295-303
    frame.features_.erase(std::remove_if(frame.features_.begin(), frame.features_.end(), [](const Feature& x) { return !x.is_inlier; }), frame.features_.end()); // :306-311
}

bool VO::check_motion_estimation() {
    const double frame_gap = frame_current_.frame_id_ - frame_last_.frame_id_;
    const bool ok = vslam_check_motion(num_inliers_, T_c_l_.data(), frame_gap) != 0; // :319-334
    if (!ok) {
        std::cout << "Frame id: " << frame_last_.frame_id_ << " and " << frame_current_.frame_id_ << std::endl;
        if (num_inliers_ < 10) std::cout << "Rejected - inliers not enough: " << num_inliers_ << std::endl;
        else std::cout << "Rejected - motion is too large: " << norm6(T_c_l_.log()) << std::endl;
    }
    return ok;
}

bool VO::insert_key_frame(bool check_ok, std::vector<Point3f>& pts_3d, std::vector<KeyPoint>& keypoints, DescriptorMat& descriptors) {
    // enough inliers and no (signed, quirk Q2) turn, or a rejected frame: not a keyframe (:353)
    if ((num_inliers_ >= 80 && T_c_l_.angleY() < 0.03) || !check_ok) return false;
    frame_current_.is_keyframe_ = true;
    frame_current_.keyframe_id_ = curr_keyframe_id_;
    for (Feature& f : frame_current_.features_) { // observations of the tracked landmarks (:363-368)
        Landmark& lm = my_map_.landmarks_.at((unsigned long)f.landmark_id_);
        lm.observed_times_++;
        lm.observations_.emplace_back(frame_current_.keyframe_id_, f.feature_id_);
    }
    // new landmarks from this keyframe's stereo depth (:377-378)
    std::vector<bool> reliable_depth = set_ref_3d_position(pts_3d, keypoints, descriptors, frame_current_);
    std::unordered_multimap<uint64_t, size_t> by_pixel; // replaces the O(N*M) exact-equality scan (:385-401)
    for (size_t j = 0; j < frame_current_.features_.size(); ++j) by_pixel.emplace(pixel_key(frame_current_.features_[j].keypoint_.pt), j);
    int feature_id = (int)frame_current_.features_.size(); // :383 (post-erase size: ids may collide, quirk Q1)
    for (size_t i = 0; i < keypoints.size(); ++i) {
        auto range = by_pixel.equal_range(pixel_key(keypoints[i].pt));
        bool exist = false;
        for (auto it = range.first; it != range.second; ++it) {
            exist = true;
            Landmark& lm = my_map_.landmarks_.at((unsigned long)frame_current_.features_[it->second].landmark_id_);
            if (!lm.reliable_depth_ && reliable_depth[i]) { lm.pt_3d_ = pts_3d[i]; lm.reliable_depth_ = true; } // :395-399
        }
        if (exist) continue;
        Feature f(feature_id, frame_current_.frame_id_, keypoints[i], descriptors.row((int)i));
        f.landmark_id_ = curr_landmark_id_;
        frame_current_.features_.push_back(f);
        my_map_.insert_landmark(Landmark(curr_landmark_id_, pts_3d[i], descriptors.row((int)i), reliable_depth[i], Observation(frame_current_.keyframe_id_, feature_id)));
        curr_landmark_id_++;
        feature_id++;
    }
    curr_keyframe_id_++;
    my_map_.insert_keyframe(frame_current_); // :426
    return true;
}

bool VO::initialization() {
    frame_last_ = Frame();
    if (read_img(0, frame_last_.left_img_, frame_last_.right_img_) != 0) return false;
    frame_last_.frame_id_ = 0;
    std::vector<KeyPoint> keypoints; DescriptorMat descriptors; std::vector<Point3f> pts_3d;
    if (feature_detection(frame_last_.left_img_, keypoints, descriptors) != 0) return false;
    std::vector<bool> reliable_depth = set_ref_3d_position(pts_3d, keypoints, descriptors, frame_last_); // :505-507
    for (size_t i = 0; i < keypoints.size(); ++i) { // :509-525
        Feature f((int)i, 0, keypoints[i], descriptors.row((int)i));
        f.landmark_id_ = curr_landmark_id_;
        frame_last_.features_.push_back(f);
        my_map_.insert_landmark(Landmark(curr_landmark_id_, pts_3d[i], descriptors.row((int)i), reliable_depth[i], Observation(0, (int)i)));
        curr_landmark_id_++;
    }
    frame_last_.fill_frame(SE3(), true, curr_keyframe_id_); // :536
    curr_keyframe_id_++;
    my_map_.insert_keyframe(frame_last_);
    return true;
}

bool VO::tracking(bool& if_insert_keyframe) {
    frame_current_ = Frame();
    if (frame_last_.is_keyframe_) frame_last_ = my_map_.keyframes_.at((unsigned long)frame_last_.keyframe_id_); // BA-refined pose (:553-556)
    if (read_img(seq_, frame_current_.left_img_, frame_current_.right_img_) != 0) { seq_++; return false; }
    frame_current_.frame_id_ = seq_;
    std::vector<KeyPoint> keypoints_detected; DescriptorMat descriptors_detected;
    feature_detection(frame_current_.left_img_, keypoints_detected, descriptors_detected);
    DescriptorMat descriptors_last;
    for (const Feature& f : frame_last_.features_) descriptors_last.push_back(f.descriptor_); // :568-574
    std::vector<DMatch> feature_matches;
    feature_matching(descriptors_last, descriptors_detected, feature_matches);
    last_num_detected_ = (int)keypoints_detected.size(); last_num_matches_ = (int)feature_matches.size();
    last_match_hash_ = 1469598103934665603ULL;
    for (const DMatch& m : feature_matches)
        for (uint32_t v : {(uint32_t)m.queryIdx, (uint32_t)m.trainIdx, (uint32_t)m.distance}) { last_match_hash_ ^= v; last_match_hash_ *= 1099511628211ULL; }
    for (size_t i = 0; i < feature_matches.size(); ++i) { // :587-598
        const DMatch& m = feature_matches[i];
        Feature f((int)i, seq_, keypoints_detected[(size_t)m.trainIdx], descriptors_detected.row(m.trainIdx));
        f.landmark_id_ = frame_last_.features_[(size_t)m.queryIdx].landmark_id_;
        frame_current_.features_.push_back(f);
    }
    motion_estimation(frame_current_);
    frame_current_.T_c_w_ = T_c_w_;
    T_c_l_ = frame_current_.T_c_w_ * frame_last_.T_c_w_.inverse(); // :615
    const bool ok = check_motion_estimation();
    std::vector<Point3f> pts_3d;
    if_insert_keyframe = insert_key_frame(ok, pts_3d, keypoints_detected, descriptors_detected);
    if (ok) move_frame(); // a rejected frame is skipped: the next one matches across a wider gap (:630-637)
    seq_++;
    return ok;
}

bool VO::pipeline(bool& if_insert_keyframe) {
    switch (state_) {
    case Init:
        if (initialization()) state_ = Track;
        else if (++num_lost_ > 10) state_ = Lost;
        break;
    case Track:
        if (tracking(if_insert_keyframe)) num_lost_ = 0;
        else if (++num_lost_ > 10) state_ = Lost;
        break;
    case Lost:
    default:
        std::cout << "VO IS LOST" << std::endl;
        return false;
    }
    return true;
}

} // namespace vslam
