"""how long do pageable host <-> device copies take on this box (the host-buffer tier pays them on every call)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import stereo_visual_slam_amd as pkg
vo = pkg.VO(device=0, max_batch=1)
lib = vo.lib
for nbytes in (4, 4096, 467 * 1024, 1866 * 1024, 8 << 20):
    d = C.c_void_p()
    assert lib.vslam_dev_alloc(C.byref(d), C.c_size_t(nbytes)) == 0
    h = np.zeros(nbytes, np.uint8)
    hp = h.ctypes.data_as(C.c_void_p)
    for name, fn in (("upload", lambda: lib.vslam_dev_upload(vo.h, d, hp, C.c_size_t(nbytes))), ("download", lambda: lib.vslam_dev_download(vo.h, hp, d, C.c_size_t(nbytes)))):
        fn()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        print("%-8s %9d B  %8.3f ms" % (name, nbytes, (time.perf_counter() - t0) / 20 * 1e3))
    lib.vslam_dev_free(d)
vo.close()
