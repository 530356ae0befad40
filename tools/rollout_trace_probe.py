"""A few MHPPO rollouts at 4096 envs for a rocprofv3 --kernel-trace timeline (no update phase): where does a control step's time go?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg, env, Algo = bench.build(N, "cuda:0", 0)
algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0")
algo.setup()
obs = env.reset_all()
env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 6 + 2, seed=1))
algo._train_mode()
for _ in range(5):
    obs = algo._rollout_step(obs)
    algo.storage.clear()
torch.cuda.synchronize()
