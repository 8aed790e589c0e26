#!/usr/bin/env python3
"""Generates tests/golden/run1_frames_0_7.npz from the reference's bundled data set (run1/, the input of
BASELINE.json configs[0] `vo 400 run1`).  Data only — no reference source is copied:
  * left/right frames 0..7 (512x288 RGB PNG) decoded with PIL and converted to 8-bit gray with OpenCV's
    BGR2GRAY integer formula: (B*1868 + G*9617 + R*4899 + 8192) >> 14;
  * rows 0..7 of run1/result.csv (the trajectory the reference recorded) and of run1/gt.csv.
Run in the build container only (needs /root/reference): python tests/golden/make_run1_fixture.py
"""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/run1"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "run1_frames_0_7.npz")


def gray(path):
    a = np.asarray(Image.open(path).convert("RGB")).astype(np.int64)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def main():
    n = 8
    left = np.stack([gray("%s/left/frame%06d.png" % (REF, i)) for i in range(n)])
    right = np.stack([gray("%s/right/frame%06d.png" % (REF, i)) for i in range(n)])
    res = np.loadtxt(REF + "/result.csv", delimiter=",", skiprows=1)[:n]
    gt = np.loadtxt(REF + "/gt.csv", delimiter=",", skiprows=1)[:n]
    np.savez_compressed(OUT, left=left, right=right, result_csv=res, gt_csv=gt,
                        result_header="x,y,z,gtx,gty", gt_header="time,x,y,dx,dy")
    print(OUT, os.path.getsize(OUT), left.shape)


if __name__ == "__main__":
    main()
