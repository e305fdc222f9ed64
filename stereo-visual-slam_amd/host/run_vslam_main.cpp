// run_vslam_main.cpp -- ROS-free driver with the loop and BA schedule of the reference's node
// (/root/reference/src/run_vslam.cpp:17-92).
//
//   run_vslam <dataset_dir/> <n_frames> [if_write_pose=1] [anms_num=500] [traj_path=estimated_traj.txt] [q1=1]
//             [depth=1] [pnp=1] [trace_path]
//
// dataset_dir holds image_0/%06d.png + image_1/%06d.png (KITTI's 8-bit gray PNGs, what the reference reads at
// visual_odometry.cpp:44-50) or the same names with .pgm (binary P5).  q1: reproduce the reference's use of
// Feature::feature_id_ as a vector index.  depth: 1 = the reference's own depth source (SGBM
// disparity + Frame::find_3d), 0 = the north_star stage (right-image ORB, L/R match, rectified DLT).  pnp: 1 = the
// reference's solvePnPRansac control flow, 0 = motion-only Huber LM.  The defaults are the reference's algorithm.
// trace_path: one line per frame / per BA run with the integer decisions and the f64 poses (%.17g), for the
// CPU-path vs GPU-path parity test (tests/test_gpu_host_driver.py).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

#include "ba_host.hpp"
#include "map_host.hpp"
#include "vo_host.hpp"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <dataset_dir/> <n_frames> [if_write_pose] [anms_num] [traj_path] [q1] [depth: 1 = SGBM, 0 = L/R match + DLT] "
                             "[pnp: 1 = RANSAC, 0 = motion-only LM] [trace_path]\n", argv[0]);
        return 2;
    }
    const std::string dataset = argv[1];
    const int n_frames = std::atoi(argv[2]);
    const bool if_write_pose = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const int anms_num = argc > 4 ? std::atoi(argv[4]) : 500;
    const std::string traj = argc > 5 ? argv[5] : "estimated_traj.txt";
    const bool q1 = argc > 6 ? std::atoi(argv[6]) != 0 : true;
    const bool sgbm = argc > 7 ? std::atoi(argv[7]) != 0 : true;
    const bool ransac = argc > 8 ? std::atoi(argv[8]) != 0 : true;
    std::FILE* trace = argc > 9 ? std::fopen(argv[9], "w") : nullptr;

    vslam::Image probe_l, probe_r;
    if (vslam::ImageSource(dataset).read(0, probe_l, probe_r) != 0) return 1;
    vslam_params p;
    vslam_default_params(&p);
    p.img_w = probe_l.cols; p.img_h = probe_l.rows; p.anms_num = anms_num; p.max_batch = 1;
    vslam_ctx* ctx = nullptr;
    if (vslam_create(&p, 0, nullptr, &ctx) != VSLAM_OK) { std::fprintf(stderr, "vslam_create: %s\n", vslam_last_error()); return 1; }

    std::remove(traj.c_str());
    vslam::Map my_map(if_write_pose, traj);
    vslam::VO my_VO(dataset, ctx, my_map);
    vslam::set_optimizer_backend(ctx, q1); // VO's constructor already bound ctx; the Q1 switch is this driver's option
    my_VO.depth_source_ = sgbm ? vslam::DepthSGBM : vslam::DepthStereoMatch;
    my_VO.pnp_mode_ = ransac ? vslam::PnpRansac : vslam::PnpMotionOnlyLM;
    // run_vslam.cpp:34-38: K and the baseline are constants of the node (the same numbers Frame carries, types_def.hpp:53-54)
    const vslam::Mat33 K = {p.cam[0], 0, p.cam[2], 0, p.cam[1], p.cam[3], 0, 0, 1};
    int n_keyframes = 0, n_ok = 0, n_ba = 0;
    const auto loop_t0 = std::chrono::steady_clock::now(); // (the frame loop: image read + pipeline + BA schedule, what README.md:90 of the reference times per keyframe)
    for (int ite = 0; ite < n_frames; ite++) { // run_vslam.cpp:40
        bool if_insert_keyframe = false;
        const bool not_lost = my_VO.pipeline(if_insert_keyframe);
        if (if_insert_keyframe) ++n_keyframes;
        if (not_lost) ++n_ok;
        if (trace) {
            const double* T = my_VO.T_c_w_.data();
            std::fprintf(trace, "frame %d state %d kf %d det %d matches %d match_hash %016llx inliers %d landmarks %zu keyframes %zu T", ite, (int)my_VO.state_,
                         if_insert_keyframe ? 1 : 0, my_VO.last_num_detected_, my_VO.last_num_matches_, (unsigned long long)my_VO.last_match_hash_,
                         my_VO.num_inliers_, my_map.landmarks_.size(), my_map.keyframes_.size());
            for (int i = 0; i < 7; ++i) std::fprintf(trace, " %.17g", T[i]);
            std::fprintf(trace, "\n");
        }
        if (if_insert_keyframe && my_map.keyframes_.size() >= 10) { // :58-71
            // reject the outliers, then do the optimization; then pose only -- the four calls exactly as run_vslam.cpp:61-70 spells them
            vslam::optimize_map(my_map.keyframes_, my_map.landmarks_, K, false, false, 5);
            vslam::optimize_map(my_map.keyframes_, my_map.landmarks_, K, false, false, 5);
            vslam::optimize_map(my_map.keyframes_, my_map.landmarks_, K, true, false, 10);
            vslam::optimize_pose_only(my_map.keyframes_, my_map.landmarks_, K, true, 10);
            ++n_ba;
            if (trace) {
                size_t n_in = 0;
                for (const auto& kv : my_map.landmarks_) n_in += kv.second.is_inlier ? 1 : 0;
                std::map<unsigned long, const vslam::Frame*> ordered;
                for (const auto& kv : my_map.keyframes_) ordered[kv.first] = &kv.second;
                std::fprintf(trace, "ba %d inlier_landmarks %zu", ite, n_in);
                for (const auto& kv : ordered) {
                    std::fprintf(trace, " kf %lu", kv.first);
                    for (int i = 0; i < 7; ++i) std::fprintf(trace, " %.17g", kv.second->T_c_w_.data()[i]);
                }
                std::fprintf(trace, "\n");
            }
        }
        if (!not_lost) break;
    }
    const double loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_t0).count();
    if (if_write_pose) my_map.write_remaining_pose(); // :84-87
    if (trace) std::fclose(trace);
    std::printf("frames %d keyframes_inserted %d ba_runs %d map_keyframes %zu landmarks %zu last_inliers %d\n", my_VO.seq_, n_keyframes, n_ba,
                my_map.keyframes_.size(), my_map.landmarks_.size(), my_VO.num_inliers_);
    const auto t = my_VO.T_c_w_.inverse().translation();
    std::printf("final_position %.6f %.6f %.6f\n", t[0], t[1], t[2]);
    std::printf("timing loop_s %.6f frames %d frames_per_s %.3f keyframes_per_s %.3f s_per_keyframe %.6f\n", loop_s, my_VO.seq_, my_VO.seq_ / loop_s,
                n_keyframes / loop_s, n_keyframes > 0 ? loop_s / n_keyframes : 0.0);
    vslam::clear_optimizer_backend(ctx); // (my_VO outlives this statement: unbind before the context goes)
    vslam_destroy(ctx);
    return 0;
}
