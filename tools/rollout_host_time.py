"""Host-side vs device-side time of one MHPPO rollout (24 control steps at 4096 envs): is the rollout launch-bound?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg, env, Algo = bench.build(N, "cuda:0", 0)
algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0")
algo.setup()
obs = env.reset_all()
env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 8 + 2, seed=1))
algo._train_mode()
for _ in range(3):
    obs = algo._rollout_step(obs); algo.storage.clear()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs = algo._rollout_step(obs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    algo.storage.clear()
    print(f"rollout: host enqueue {1e3 * (t1 - t0):.2f} ms, until device idle {1e3 * (t2 - t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
obs = algo._rollout_step(obs)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
