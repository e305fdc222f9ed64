// run_vslam_main.cpp -- ROS-free driver with the loop and BA schedule of the reference's node
// (/root/reference/src/run_vslam.cpp:17-92).  Usage: run_vslam <dataset_dir/> <n_frames> [if_write_pose=1] [anms_num=500] [traj_path]
// The dataset directory holds image_0/%06d.pgm and image_1/%06d.pgm (binary PGM instead of KITTI's PNG).
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "ba_host.hpp"
#include "map_host.hpp"
#include "vo_host.hpp"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s <dataset_dir/> <n_frames> [if_write_pose] [anms_num] [traj_path] [q1] [depth: 0 = L/R match + DLT, 1 = SGBM] [pnp: 0 = motion-only LM, 1 = RANSAC]\n", argv[0]); return 2; }
    const std::string dataset = argv[1];
    const int n_frames = std::atoi(argv[2]);
    const bool if_write_pose = argc > 3 ? std::atoi(argv[3]) != 0 : true;
    const int anms_num = argc > 4 ? std::atoi(argv[4]) : 500;
    const std::string traj = argc > 5 ? argv[5] : "estimated_traj.txt";
    const bool q1 = argc > 6 ? std::atoi(argv[6]) != 0 : true; // reproduce the reference's feature_id-as-index behaviour

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
    my_VO.depth_source_ = (argc > 7 && std::atoi(argv[7]) != 0) ? vslam::DepthSGBM : vslam::DepthStereoMatch;
    my_VO.pnp_mode_ = (argc > 8 && std::atoi(argv[8]) != 0) ? vslam::PnpRansac : vslam::PnpMotionOnlyLM;
    int n_keyframes = 0, n_ok = 0, n_ba = 0;
    for (int ite = 0; ite < n_frames; ite++) { // run_vslam.cpp:40
        bool if_insert_keyframe = false;
        const bool not_lost = my_VO.pipeline(if_insert_keyframe);
        if (if_insert_keyframe) ++n_keyframes;
        if (not_lost) ++n_ok;
        if (if_insert_keyframe && my_map.keyframes_.size() >= 10) { // :58-71
            vslam::optimize_map(ctx, my_map.keyframes_, my_map.landmarks_, false, false, 5, q1);
            vslam::optimize_map(ctx, my_map.keyframes_, my_map.landmarks_, false, false, 5, q1);
            vslam::optimize_map(ctx, my_map.keyframes_, my_map.landmarks_, true, false, 10, q1);
            vslam::optimize_pose_only(ctx, my_map.keyframes_, my_map.landmarks_, true, 10, q1);
            ++n_ba;
        }
        if (!not_lost) break;
    }
    if (if_write_pose) my_map.write_remaining_pose(); // :84-87
    std::printf("frames %d keyframes_inserted %d ba_runs %d map_keyframes %zu landmarks %zu last_inliers %d\n", my_VO.seq_, n_keyframes, n_ba,
                my_map.keyframes_.size(), my_map.landmarks_.size(), my_VO.num_inliers_);
    const auto t = my_VO.T_c_w_.inverse().translation();
    std::printf("final_position %.6f %.6f %.6f\n", t[0], t[1], t[2]);
    vslam_destroy(ctx);
    return 0;
}
