"""A few ppo_mimic rollouts (29-DoF teacher, 256 clips, 4096 envs) for a rocprofv3 --kernel-trace timeline of one control step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

cfg, env, Algo = bench.build(4096, "cuda:0", 0, workload="v2_teacher29", num_clips=256)
algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0")
algo.setup()
obs = env.reset_all()
env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 5 + 2, seed=1))
algo._train_mode()
for _ in range(4):
    obs = algo._rollout_step(obs)
    algo.storage.clear()
torch.cuda.synchronize()
