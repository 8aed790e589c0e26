"""The C++ facade (include/svo/visual_odometry.hpp) compiles with plain g++ against the C-ABI, and on a GPU
the reference's run_tests() restated with it (tests/cpp/facade_kat.cpp) passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_kat.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "facade_kat")


def build():
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L" + os.path.join(ROOT, "stereo_visual_odometry_amd"), "-lsvo_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "stereo_visual_odometry_amd")]
    subprocess.check_call(cmd)


def test_facade_compiles_and_links_with_gxx():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_reference_run_tests_through_facade():
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL TESTS PASS" in out.stdout


def test_stereo_synchronizer_host_logic():
    """ROS-callback-shaped adapter (include/svo/stereo_sync.hpp): pairing + bounded queues; no GPU needed."""
    exe = os.path.join(ROOT, "tests", "cpp", "sync_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sync_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "stereo_visual_odometry_amd"), "-lsvo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "stereo_visual_odometry_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "SYNC OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_ros_callback_binding_on_the_gpu(tmp_path):
    """synchroniser -> VisualOdometry::stereo_callback (src/stereo_vo.cpp:53-62) end to end on the HIP path:
    jittered stamped frames through make_stereo_vo_callback give the poses of direct stereo_callback calls on the same pairs."""
    import numpy as np
    from stereo_visual_odometry_amd import synthetic as syn
    cal = dict(syn.KITTI00, width=320, height=160, cx=160.0, cy=80.0)
    seq = syn.StereoSequence(cal=cal, n_frames=7, seed=21, step=0.3)
    path = tmp_path / "frames.bin"
    with open(path, "wb") as f:
        f.write(np.array([seq.n_frames, 160, 320], np.int32).tobytes())
        for l, r in zip(seq.left, seq.right):
            f.write(l.tobytes()); f.write(r.tobytes())
    exe = os.path.join(ROOT, "tests", "cpp", "sync_gpu_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sync_gpu_test.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "stereo_visual_odometry_amd"), "-lsvo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "stereo_visual_odometry_amd")])
    out = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "SYNC GPU OK" in out.stdout, out.stdout + out.stderr
