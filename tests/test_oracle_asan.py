"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; SURVEY.md §5 "race detection /
sanitizers" row: the reference has none).  oracle/Makefile's `asan` target builds the same sources with
-fsanitize=address,undefined; the reference's known-answer tests and a short full-pipeline run then execute against that
library in a child process (the sanitizer runtime has to be preloaded into the interpreter).  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PIPE = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import oracle_lib as orc
from stereo_visual_odometry_amd import synthetic as syn
cal = dict(syn.KITTI00, width=320, height=160, cx=160.0, cy=80.0)
seq = syn.StereoSequence(cal=cal, n_frames=4, seed=11, step=0.4, movers=0.3)
Pl, Pr = syn.projection_matrices(cal)
for win, cn in ((21, 1), (10, 1), (7, 3)):
    o = orc.VisualOdometry(orc.default_config(win_w=win, win_h=win, max_level=3, max_translation_norm=2.0, channels=cn))
    o.initalize_projection_matricies(Pl, Pr)
    oks = 0
    for k in range(4):
        L, R = seq.left[k], seq.right[k]
        if cn == 3:
            L = np.ascontiguousarray(np.stack([L, np.roll(L, 1, 1), L[::-1]], -1)); R = np.ascontiguousarray(np.stack([R, np.roll(R, 1, 1), R[::-1]], -1))
        ok, T = o.stereo_callback(L, R)
        oks += bool(ok)
    print("win", win, "cn", cn, "poses", oks)
# border / degenerate inputs: black frames, then texture again
o = orc.VisualOdometry(orc.default_config(win_w=21, win_h=21)); o.initalize_projection_matricies(Pl, Pr)
z = np.zeros((160, 320), np.uint8)
for L, R in ((seq.left[0], seq.right[0]), (z, z), (z, z), (seq.left[1], seq.right[1]), (seq.left[2], seq.right[2])):
    o.stereo_callback(L, R)
print("asan pipeline ok")
"""


def _asan_env():
    lib = os.path.join(ROOT, "oracle", "libsvo_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no libasan runtime in this image")
    env = dict(os.environ)
    env.update(SVO_ORACLE_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    return env


def test_reference_kats_under_asan_ubsan():
    env = _asan_env()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kat.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_pipeline_under_asan_ubsan():
    env = _asan_env()
    r = subprocess.run([sys.executable, "-c", PIPE % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "asan pipeline ok" in r.stdout and "runtime error" not in r.stderr
