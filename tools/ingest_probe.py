"""Diagnostic (GPU box): motion-library ingestion throughput — BASELINE configs[4]'s library size, 2048 synthetic clips made
from the shipped 29-DoF clip (time-warped / shifted variants, bench.synth_library): host concatenation + one upload + ONE batched
FK / velocity / filter / pack launch set (`pbhc_motion_build_batch`).  Prints clips, frames, rows' bytes, wall time, frames/s."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pbhc_amd.motion_lib import MotionLib
from pbhc_amd.skeleton import Skeleton
from tests.helpers import GOLDEN, clip_from_env_golden

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = dict(np.load(os.path.join(GOLDEN, "env_v2_teacher29.npz")))
sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_29dof_rev_1_0.json"))
clips = bench.synth_library(clip_from_env_golden(g), M, seed=7)
MotionLib(sk, clips[:4], 8, "cuda:0")                         # warm-up (module load)
torch.cuda.synchronize()
t0 = time.perf_counter()
ml = MotionLib(sk, clips, 4096, "cuda:0")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
F = int(ml.frames.shape[0])
print(f"library ingestion: {M} clips, {F} frames, {ml.frames.numel() * 4 / 1e6:.1f} MB of packed rows in {dt * 1e3:.1f} ms "
      f"= {F / dt / 1e6:.2f} M frames/s, {M / dt:.0f} clips/s (host concatenation + upload + one batched FK / filter / pack launch set)")
t0 = time.perf_counter()
ml.load_motions(random_sample=True)
torch.cuda.synchronize()
print(f"load_motions (slot -> clip resampling of 4096 env slots; the reference re-runs one Python FK per slot here): {(time.perf_counter() - t0) * 1e3:.2f} ms")
