"""bench.py prints ONE JSON line with the contract's keys (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload + roofline + cpu_baseline at N=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--envs", "512"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1 and r["higher_is_better"] is True and r["scaling"] == "weak" and r["dtype"] == "f32"
    assert r["unit"] == "env-steps/s" and r["value"] > 0 and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 512 * 24 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-6
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and "traffic" in rf
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "env-steps/s" and "sample" in cb


def test_bench_secondary_object_is_configs_2():
    """the default invocation appends BASELINE.json configs[2] (general tracking, 29-DoF, 256-clip library) as `secondary`"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--envs", "256", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    s = r["secondary"]
    assert s["value"] > 0 and s["unit"] == "env-steps/s" and "29-DoF" in s["config"]["workload"] and "256 clips" in s["config"]["workload"]
    assert s["roofline"]["bound"] == "hbm" and s["roofline"]["kernel_ms"] > 0 and s["roofline_update"]["bound"] == "mfma"


def test_rccl_code_path_on_one_rank():
    """PBHC_DIST_FORCE=1: a ONE-rank RCCL process group takes the data-parallel code path — broadcast of the initial weights, ONE averaged
    all-reduce of the actor + critic gradient bucket (KL mean in its last slot) per optimiser step, the advantage-moment exchange and the
    once-per-rollout exchange of the env statistics, all next to hipGraph replays of the policy forward.  What a one-GPU box can rehearse
    of the N-GPU run."""
    env = dict(os.environ, PBHC_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--envs", "512", "--no-cpu-baseline", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c = r["collectives"]
    assert c["backend"] == "nccl" and r["n_gpus"] == 1
    # per iteration: 20 optimiser steps x 1 gradient bucket (the KL scalar rides in it) + 1 advantage-moment exchange + 1 env-statistics exchange
    assert c["all_reduces_per_iter"] == 20 + 1 + 1, c
    # actor 380-512-256-128-23 (+ std) and critic 630-768-512-128-R, R = reward columns of the walk config (19 terms + termination = 20)
    assert c["grad_bucket_bytes"] == 4 * (362286 + 944000 + 129 * 20) and c["grad_allreduce_ms"] > 0
