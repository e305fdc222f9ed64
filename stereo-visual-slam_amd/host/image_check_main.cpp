// image_check <a.png> <a.pgm>: decodes both with the host driver's readers and compares them (used by tests/test_host_io.py)
#include "vo_host.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    vslam::Image a, b;
    if (vslam::ImageSource::read_png(argv[1], a) || vslam::ImageSource::read_pgm(argv[2], b)) { std::puts("read failed"); return 1; }
    if (a.cols != b.cols || a.rows != b.rows || a.data != b.data) { std::puts("MISMATCH"); return 2; }
    std::printf("png == pgm (%d x %d)\n", a.cols, a.rows);
    return 0;
}
