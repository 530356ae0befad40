#!/usr/bin/env python3
"""SURVEY §8(d) CPU-baseline item (1): the REFERENCE ITSELF timed on the host cores of the build container.

Build-container only (it imports the unmodified reference from /root/reference through oracle/ref_harness — the same stand-ins and
replay fake simulator the golden fixtures were generated with; it never travels to the GPU box and nothing in tests / bench / smoke calls
it).  Times full PPO iterations — 24-step rollout (`MHPPO._rollout_step`: policy forward + `LeggedRobotMotionTracking.step` + storage) and
`_training_step` (5 epochs x 4 minibatches) — of the reference's v1 agent on the BASELINE configs' shapes:

    python tools/time_reference_cpu.py            # N = 64 (config 1, Horse-stance) and N = 4096 (config 2, g1_walk_45cms)

Median of `--iters` iterations after one warm-up, all host cores (`torch.get_num_threads()` printed).  Prints one JSON line per case and
appends them to profiles/round2_reference_cpu_timing.jsonl.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--envs", type=int, nargs="*", default=[64, 4096])
    a = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        raise SystemExit("the reference is not present: this tool runs in the build container only")
    import torch

    from oracle.ref_harness import gen_golden as G  # noqa: F401  (installs the stand-ins, chdir to the reference)
    from oracle.ref_harness.gen_env_golden import V1_CFG, build_env, make_cfg, make_replay, oracle_motion_lib
    from humanoidverse.agents.mh_ppo.mh_ppo import MHPPO

    torch.set_num_threads(os.cpu_count() or 1)
    out_path = os.path.join(ROOT, "profiles", "round2_reference_cpu_timing.jsonl")
    for N in a.envs:
        # config 1: 64 envs on Horse-stance_pose (the composed config the reference ships); config 2's shape: 4096 envs on g1_walk_45cms
        motion = None if N <= 64 else "motion_data/g1_walk_45cms_23dof.pkl"
        extra = {} if motion is None else {"rewards.reward_scales.teleop_contact_mask": 0.0}      # the walk clip has no contact_mask
        cfg = make_cfg(V1_CFG, N, motion_file=motion, extra=extra)
        t0 = time.perf_counter()
        env = build_env(cfg, seed=3)
        build_s = time.perf_counter() - t0
        skel, clip, ml = oracle_motion_lib(cfg)
        torch.manual_seed(4)
        algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device="cpu")
        algo.setup()
        T = algo.num_steps_per_env
        root, qp, qv, cf = make_replay(env, ml, (a.iters + 1) * T + 2, seed=9, script=False)
        env.simulator.set_replay(root, qp, qv, cf, start_frame=0)
        obs = env.reset_all()
        algo._train_mode()
        roll, upd = [], []
        for it in range(a.iters + 1):
            algo.start_time = time.time()
            t0 = time.perf_counter()
            obs = algo._rollout_step(obs)
            t1 = time.perf_counter()
            algo._training_step()
            t2 = time.perf_counter()
            if it > 0:                                  # iteration 0 = warm-up
                roll.append(t1 - t0); upd.append(t2 - t1)
            print(f"  N={N} iteration {it}: rollout {t1 - t0:.2f} s, update {t2 - t1:.2f} s", flush=True)
        r, u = statistics.median(roll), statistics.median(upd)
        rec = {"what": "reference (unmodified humanoidverse MHPPO + LeggedRobotMotionTracking on the replay fake simulator), CPU",
               "num_envs": N, "steps_per_iter": T, "rollout_s": r, "update_s": u, "env_steps_per_s": N * T / (r + u),
               "iters_timed": a.iters, "threads": torch.get_num_threads(), "cores": os.cpu_count(), "env_build_s": build_s,
               "torch": torch.__version__, "motion": motion or "example/motion_data/Horse-stance_pose.pkl"}
        print(json.dumps(rec), flush=True)
        with open(out_path, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
