"""Single-stream latency without Python in the loop: renders the bench scene's frames, builds tools/svo_latency.cpp (plain g++
against the C-ABI) and runs it.   python tools/latency_cpp.py [win] [calls]   (MI355X box)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereo_visual_odometry_amd import synthetic as syn   # noqa: E402

win = sys.argv[1] if len(sys.argv) > 1 else "21"
calls = sys.argv[2] if len(sys.argv) > 2 else "200"
exe = os.path.join(ROOT, "tools", "svo_latency")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "svo_latency.cpp"),
                       "-o", exe, "-L" + os.path.join(ROOT, "stereo_visual_odometry_amd"), "-lsvo_hip", "-Wl,-rpath," + os.path.join(ROOT, "stereo_visual_odometry_amd")])
cal = syn.KITTI00
for movers in (0.0, 0.3):
    seq = syn.StereoSequence(cal=cal, n_frames=8, seed=0x5EED0002, step=0.5, cell_px=17.6, movers=movers)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(np.array([seq.n_frames, cal["height"], cal["width"]], np.int32).tobytes())
        for l, r in zip(seq.left, seq.right):
            f.write(l.tobytes()); f.write(r.tobytes())
        path = f.name
    print("movers %.1f, window %s" % (movers, win), flush=True)
    subprocess.check_call([exe, path, win, calls])
    os.unlink(path)
