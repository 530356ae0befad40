"""bench.py starts its own ranks: `python bench.py --gpus N` with no launcher environment must come back as ONE JSON line from N
ranks (n_gpus == N, parallelism dpN) and never as a silent single-rank run.  CPU rehearsal: PBHC_BENCH_DRYRUN=1 runs the rendezvous and
the collective pattern of one PPO iteration over gloo (the product itself has no CPU path)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_2_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"PBHC_BENCH_DRYRUN": "1", "PBHC_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                         # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_envs"] == 2 * out["config"]["envs_per_gpu"]
    assert out["collectives"]["dry_run_all_reduces"] == 3 and out["collectives"]["grad_allreduce_ms"] > 0
    assert out["scaling"] == "weak" and out["higher_is_better"] is True


def test_bench_refuses_a_world_size_that_is_not_gpus():
    # a launcher that started ONE rank for --gpus 2 must not produce an n_gpus: 1 line
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"PBHC_BENCH_DRYRUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
