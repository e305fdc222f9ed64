// se3_device.h -- SE3 algebra used by the kernels and by the C++ host mirror.
// Storage: 7 doubles, unit quaternion (x,y,z,w) then translation (the memory order of Sophus::SE3d, which the
// reference uses for every pose: /root/reference/include/stereo_visual_slam_main/library_include.hpp:18).
// Tangent vectors are [upsilon(3); omega(3)] (translation first), the convention optimization.cpp:26-32 relies on.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VS_HD __host__ __device__ inline
#else
#define VS_HD inline
#endif

namespace vslam {
namespace se3 {

constexpr double kEps = 1e-10;

VS_HD void rotmat(const double* T, double R[9]) {
    const double x = T[0], y = T[1], z = T[2], w = T[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
                 tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

VS_HD void act(const double* T, const double p[3], double out[3]) {
    double R[9];
    rotmat(T, R);
    out[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + T[4];
    out[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + T[5];
    out[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + T[6];
}

VS_HD void mul(const double* A, const double* B, double* C) {
    const double ax = A[0], ay = A[1], az = A[2], aw = A[3], bx = B[0], by = B[1], bz = B[2], bw = B[3];
    double q0 = aw * bx + ax * bw + ay * bz - az * by;
    double q1 = aw * by - ax * bz + ay * bw + az * bx;
    double q2 = aw * bz + ax * by - ay * bx + az * bw;
    double q3 = aw * bw - ax * bx - ay * by - az * bz;
    const double n = sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    double t[3];
    act(A, B + 4, t);
    C[0] = q0 / n; C[1] = q1 / n; C[2] = q2 / n; C[3] = q3 / n;
    C[4] = t[0]; C[5] = t[1]; C[6] = t[2];
}

VS_HD void inverse(const double* A, double* C) {
    const double Ti[7] = {-A[0], -A[1], -A[2], A[3], 0, 0, 0};
    double t[3];
    act(Ti, A + 4, t);
    C[0] = Ti[0]; C[1] = Ti[1]; C[2] = Ti[2]; C[3] = Ti[3];
    C[4] = -t[0]; C[5] = -t[1]; C[6] = -t[2];
}

VS_HD void hat_sq(const double w[3], double O[9], double O2[9]) {
    O[0] = 0; O[1] = -w[2]; O[2] = w[1];
    O[3] = w[2]; O[4] = 0; O[5] = -w[0];
    O[6] = -w[1]; O[7] = w[0]; O[8] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
}

// Sophus::SE3d::exp
VS_HD void exp(const double xi[6], double* T) {
    const double* ups = xi;
    const double* om = xi + 3;
    const double theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const double theta = sqrt(theta_sq), half = 0.5 * theta;
    double imag, real;
    if (theta < kEps) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - 0.125 * theta_sq + (1.0 / 384.0) * t4;
    } else {
        imag = sin(half) / theta;
        real = cos(half);
    }
    double q[4] = {imag * om[0], imag * om[1], imag * om[2], real};
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    T[0] = q[0] / n; T[1] = q[1] / n; T[2] = q[2] / n; T[3] = q[3] / n;
    double O[9], O2[9], V[9];
    hat_sq(om, O, O2);
    if (theta < kEps) {
        rotmat(T, V);
    } else {
        const double a = (1 - cos(theta)) / theta_sq, b = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = a * O[i] + b * O2[i];
        V[0] += 1; V[4] += 1; V[8] += 1;
    }
    for (int i = 0; i < 3; ++i) T[4 + i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

// Sophus::SE3d::log
VS_HD void log(const double* T, double xi[6]) {
    const double sq = T[0] * T[0] + T[1] * T[1] + T[2] * T[2], n = sqrt(sq), w = T[3];
    double two_atan;
    if (n < kEps) two_atan = 2.0 / w - 2.0 * sq / (w * w * w);
    else if (fabs(w) < kEps) two_atan = (w > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
    else two_atan = 2.0 * atan(n / w) / n;
    const double theta = two_atan * n;
    const double om[3] = {two_atan * T[0], two_atan * T[1], two_atan * T[2]};
    double O[9], O2[9], Vi[9];
    hat_sq(om, O, O2);
    double c;
    if (fabs(theta) < kEps) c = 1.0 / 12.0;
    else { const double half = 0.5 * theta; c = (1.0 - theta * cos(half) / (2.0 * sin(half))) / (theta * theta); }
    for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * O[i] + c * O2[i];
    Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
    for (int i = 0; i < 3; ++i) xi[i] = Vi[i * 3] * T[4] + Vi[i * 3 + 1] * T[5] + Vi[i * 3 + 2] * T[6];
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

// Sophus::SO3d::angleY
VS_HD double angle_y(const double* T) {
    double R[9];
    rotmat(T, R);
    return atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
}

} // namespace se3
} // namespace vslam
