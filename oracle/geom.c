/*
 * oracle/geom.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see vo_oracle.h): SE3 algebra, stereo depth ->
 * 3-D (row A7), DLT triangulation (north_star stage K8), motion sanity gate (row A9).
 *
 * [UPSTREAM] Sophus::SE3d / SO3d exp, log, product, inverse, angleY restated from sophus/se3.hpp, so3.hpp
 * (not in the image; parity unpinned, pinned by round-trip and closed-form tests).
 * [REF] Frame::find_3d types_def.cpp:9-18; VO::set_ref_3d_position visual_odometry.cpp:176-217;
 * VO::check_motion_estimation visual_odometry.cpp:316-346.
 */
#include "vo_oracle.h"

#include <math.h>
#include <string.h>

#define SOPHUS_EPS 1e-10

/* quaternion stored x,y,z,w */
static void quat_mul(const double a[4], const double b[4], double c[4]) {
    double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by - ax * bz + ay * bw + az * bx;
    c[2] = aw * bz + ax * by - ay * bx + az * bw;
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
}
static void quat_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

void vo_se3_rotmat(const double T[7], double R[9]) {
    double x = T[0], y = T[1], z = T[2], w = T[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
           tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void vo_se3_act(const double T[7], const double p[3], double out[3]) {
    double R[9];
    vo_se3_rotmat(T, R);
    double x = p[0], y = p[1], z = p[2];
    out[0] = R[0] * x + R[1] * y + R[2] * z + T[4];
    out[1] = R[3] * x + R[4] * y + R[5] * z + T[5];
    out[2] = R[6] * x + R[7] * y + R[8] * z + T[6];
}

void vo_se3_mul(const double A[7], const double B[7], double C[7]) {
    double q[4], t[3];
    quat_mul(A, B, q);
    quat_normalize(q);
    vo_se3_act(A, B + 4, t);
    C[0] = q[0]; C[1] = q[1]; C[2] = q[2]; C[3] = q[3];
    C[4] = t[0]; C[5] = t[1]; C[6] = t[2];
}

void vo_se3_inv(const double A[7], double C[7]) {
    double Ti[7] = {-A[0], -A[1], -A[2], A[3], 0, 0, 0};
    double t[3];
    vo_se3_act(Ti, A + 4, t);
    C[0] = Ti[0]; C[1] = Ti[1]; C[2] = Ti[2]; C[3] = Ti[3];
    C[4] = -t[0]; C[5] = -t[1]; C[6] = -t[2];
}

static void hat_sq(const double w[3], double O[9], double O2[9]) {
    O[0] = 0; O[1] = -w[2]; O[2] = w[1];
    O[3] = w[2]; O[4] = 0; O[5] = -w[0];
    O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
            O2[i * 3 + j] = s;
        }
}

void vo_se3_exp(const double xi[6], double T[7]) {
    const double* ups = xi;
    const double* om = xi + 3;
    /* SO3::expAndTheta */
    double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double theta = sqrt(theta_sq), half = 0.5 * theta, imag, real;
    if (theta < SOPHUS_EPS) {
        double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - 0.125 * theta_sq + (1.0 / 384.0) * t4;
    } else {
        imag = sin(half) / theta;
        real = cos(half);
    }
    T[0] = imag * om[0]; T[1] = imag * om[1]; T[2] = imag * om[2]; T[3] = real;
    quat_normalize(T);
    /* V */
    double O[9], O2[9], V[9];
    hat_sq(om, O, O2);
    if (theta < SOPHUS_EPS) {
        vo_se3_rotmat(T, V); /* "V = so3.matrix()" */
    } else {
        double a = (1 - cos(theta)) / theta_sq, b = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = a * O[i] + b * O2[i];
        V[0] += 1; V[4] += 1; V[8] += 1;
    }
    for (int i = 0; i < 3; ++i) T[4 + i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

void vo_se3_log(const double T[7], double xi[6]) {
    /* SO3::logAndTheta */
    double sq = T[0] * T[0] + T[1] * T[1] + T[2] * T[2], n = sqrt(sq), w = T[3], two_atan;
    if (n < SOPHUS_EPS) {
        two_atan = 2.0 / w - 2.0 * sq / (w * w * w);
    } else if (fabs(w) < SOPHUS_EPS) {
        two_atan = (w > 0 ? M_PI : -M_PI) / n;
    } else {
        two_atan = 2.0 * atan(n / w) / n;
    }
    double theta = two_atan * n;
    double om[3] = {two_atan * T[0], two_atan * T[1], two_atan * T[2]};
    double O[9], O2[9], Vi[9];
    hat_sq(om, O, O2);
    double c;
    if (fabs(theta) < SOPHUS_EPS) c = 1.0 / 12.0;
    else {
        double half = 0.5 * theta;
        c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta);
    }
    for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
    Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
    for (int i = 0; i < 3; ++i) xi[i] = Vi[i * 3] * T[4] + Vi[i * 3 + 1] * T[5] + Vi[i * 3 + 2] * T[6];
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

double vo_se3_angle_y(const double T[7]) {
    double R[9];
    vo_se3_rotmat(T, R);
    return atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
}

/* ------------------------------------------------------------------ A7: depth -> world points ------- */

static void gate_and_store(const double rel[3], const double T_c_w[7], int i, float* xyz_w, uint8_t* valid,
                           uint8_t* reliable, int* nvalid) {
    double Tinv[7], pw[3];
    vo_se3_inv(T_c_w, Tinv);
    vo_se3_act(Tinv, rel, pw);
    /* visual_odometry.cpp:194: keep 10 < Z < 400 ; :201 reliable = Z < 40 */
    int ok = (rel[2] > 10 && rel[2] < 400);
    valid[i] = (uint8_t)ok;
    reliable[i] = (uint8_t)(ok && rel[2] < 40);
    xyz_w[3 * i] = (float)pw[0]; xyz_w[3 * i + 1] = (float)pw[1]; xyz_w[3 * i + 2] = (float)pw[2]; /* cv::Point3f */
    if (ok) ++*nvalid;
}

int vo_find_3d_disparity(const vo_keypoint* kps, int n, const float* disparity, int w, int h, int dstride,
                         const double T_c_w[7], const double cam[5], float* xyz_w, uint8_t* valid,
                         uint8_t* reliable) {
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], b = cam[4];
    int nvalid = 0;
    for (int i = 0; i < n; ++i) {
        /* types_def.cpp:11-13; at<float>(kp.pt.y, kp.pt.x) truncates float -> int (quirk Q3) */
        double x = ((double)kps[i].x - cx) / fx;
        double y = ((double)kps[i].y - cy) / fy;
        int r = (int)kps[i].y, c = (int)kps[i].x;
        if (r < 0 || r >= h || c < 0 || c >= w) { /* the reference would read out of bounds */
            valid[i] = reliable[i] = 0; xyz_w[3 * i] = xyz_w[3 * i + 1] = xyz_w[3 * i + 2] = 0.f;
            continue;
        }
        double depth = fx * b / (double)disparity[(size_t)r * dstride + c];
        double rel[3] = {x * depth, y * depth, depth};
        gate_and_store(rel, T_c_w, i, xyz_w, valid, reliable, &nvalid);
    }
    return nvalid;
}

int vo_triangulate_dlt(const float* uvL, const float* uvR, int n, const double T_c_w[7], const double cam[5], double row_tol,
                       float* xyz_w, uint8_t* valid, uint8_t* reliable) {
    const double fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], b = cam[4];
    int nvalid = 0;
    for (int i = 0; i < n; ++i) {
        /* P_L = K[I|0], P_R = K[I|-b e1]; rows  u*P3 - P1, v*P3 - P2 for both views; X = (X,Y,Z,1):
         *   fx X          - (uL-cx) Z = 0
         *          fy Y   - (vL-cy) Z = 0
         *   fx X          - (uR-cx) Z = fx b
         *          fy Y   - (vR-cy) Z = 0
         * least squares via the 3x3 normal equations (structure: N = [[2fx^2,0,n02],[0,2fy^2,n12],[.,.,n22]]) */
        double aL = (double)uvL[2 * i] - cx, bL = (double)uvL[2 * i + 1] - cy;
        double aR = (double)uvR[2 * i] - cx, bR = (double)uvR[2 * i + 1] - cy;
        double n00 = 2 * fx * fx, n11 = 2 * fy * fy;
        double n02 = -fx * (aL + aR), n12 = -fy * (bL + bR);
        double n22 = aL * aL + bL * bL + aR * aR + bR * bR;
        double r0 = fx * fx * b, r1 = 0.0, r2 = -aR * fx * b;
        /* eliminate X, Y */
        double s22 = n22 - n02 * n02 / n00 - n12 * n12 / n11;
        double s2 = r2 - n02 * r0 / n00 - n12 * r1 / n11;
        double Z = s2 / s22;
        double X = (r0 - n02 * Z) / n00;
        double Y = (r1 - n12 * Z) / n11;
        double rel[3] = {X, Y, Z};
        if (!(s22 > 0) || !isfinite(Z)) { rel[0] = rel[1] = 0; rel[2] = -1; }
        /* epipolar gate of a rectified pair: same row within row_tol px, positive disparity (row_tol < 0: off).  The reference has
         * no descriptor-matched stereo stage (its depth is SGBM, which searches along the row by construction); a cross-checked
         * L/R descriptor match has no such constraint built in, so the stage that replaces SGBM applies it explicitly. */
        if (row_tol >= 0 && (!(fabs((double)uvL[2 * i + 1] - (double)uvR[2 * i + 1]) <= row_tol) || !(uvL[2 * i] > uvR[2 * i]))) { rel[0] = rel[1] = 0; rel[2] = -1; }
        gate_and_store(rel, T_c_w, i, xyz_w, valid, reliable, &nvalid);
    }
    return nvalid;
}

int vo_check_motion(int num_inliers, const double T_c_l[7], double frame_gap) {
    if (num_inliers < 10) return 0; /* visual_odometry.cpp:319 */
    double xi[6];
    vo_se3_log(T_c_l, xi);          /* :327 */
    double nrm = 0;
    for (int i = 0; i < 6; ++i) nrm += xi[i] * xi[i];
    nrm = sqrt(nrm);
    if (nrm > 5.0 * frame_gap) return 0; /* :329 */
    return 1;
}
