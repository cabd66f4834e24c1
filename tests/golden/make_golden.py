#!/usr/bin/env python3
"""Regenerates tests/golden/frame_320x240_seed11.npz from the CPU oracle.

The reference ships no golden vectors and cannot be built or imported here (C++ on OpenCV/Eigen/Pangolin), so this
fixture pins the ORACLE's outputs (regression guard for the restatement), not the reference's: parity stays
"unpinned" in the sense of SURVEY.md F2.  Inputs come from the integer-only synthetic generator."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from orb_line_slam_amd import synth

left, right = synth.stereo_pair(11, 320, 240)
p = oracle.full_params(500, 100, 300.0, 40.0)
o = oracle.stereo_points(left, right, p)
ol, orr = oracle.line_extract(left, p.line), oracle.line_extract(right, p.line)
m, disp, le = oracle.stereo_lines(ol["kls"], ol["desc"], orr["kls"], orr["desc"], 320, 240, p.stereo)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frame_320x240_seed11.npz"),
                    crc_left=zlib.crc32(left.tobytes()), crc_right=zlib.crc32(right.tobytes()), kpsL=o["kpsL"], descL=o["descL"],
                    uRight=o["uRight"], depth=o["depth"], klsL=ol["kls"], ldescL=ol["desc"], lm12=m, ldisp=disp, lle=le)
print(len(o["kpsL"]), "kps", (o["uRight"] >= 0).sum(), "stereo;", len(ol["kls"]), "lines", (m >= 0).sum(), "stereo lines")
