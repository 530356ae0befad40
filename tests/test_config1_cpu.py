"""BASELINE configs[0]: 64 envs, G1 23-DoF, Horse-stance_pose, CPU PyTorch PPO on replayed sim-stub tensors ("plumbing, no GPU") —
one full PPO iteration of the CPU oracle end to end: 24 control steps of the restated LeggedRobotMotionTracking.step (pinned step by step
against the reference's own trace in tests/test_oracle_env.py) + GAE + 5 x 4 minibatch updates of the restated MHPPO update (pinned
against the reference's weights in tests/test_oracle_ppo.py).  The reference itself on this configuration in the build container:
tools/time_reference_cpu.py (profiles/round2_reference_cpu_timing.jsonl)."""
import math

import torch

from oracle.cpu_loop import run_iteration
from tests.helpers import clip_from_env_golden, fixture_config, load_env_golden, skel_from_golden


def test_config1_64_envs_horse_stance_one_ppo_iteration_on_cpu():
    torch.set_num_threads(4)
    cfg = fixture_config("v1_g1_23dof_horse_stance.yaml", 64)
    r = run_iteration(cfg, skel_from_golden(), clip_from_env_golden(load_env_golden("horse")), 64, seed=0, return_state=True)
    T = cfg.algo.config.num_steps_per_env
    assert r["env_steps"] == 64 * T == 1536
    assert r["seconds"] > 0 and math.isfinite(r["seconds"])
    st = r["state"]
    # the rollout buffer is full and finite; horse-stance rewards are 21 columns (20 terms + the always-zero head, SURVEY §8 note 1)
    assert tuple(st["rewards"].shape) == (T, 64, 21) and bool(torch.isfinite(st["rewards"]).all())
    assert float(st["rewards"][..., 20].abs().max()) == 0.0
    assert tuple(st["actor_obs"].shape) == (T, 64, 380) and tuple(st["critic_obs"].shape) == (T, 64, 630)
    assert bool(torch.isfinite(st["actor_obs"]).all()) and bool(torch.isfinite(st["critic_obs"]).all())
    assert bool(torch.isfinite(st["advantages"]).all()) and abs(float(st["advantages"].mean())) < 1e-4
    assert abs(float(st["advantages"].std()) - 1.0) < 1e-3                       # (A - mean) / (std + 1e-8) over all T*N samples
    # 20 optimiser steps moved the weights and kept them finite; the adaptive learning rate stayed inside the reference's clamp
    assert all(bool(torch.isfinite(v).all()) for v in r["actor_params"].values())
    assert float((r["actor_params"]["actor_module.module.0.weight"] - r["actor_params0"]["actor_module.module.0.weight"]).abs().max()) > 0
    assert 1e-5 <= r["lr"] <= 1e-2
