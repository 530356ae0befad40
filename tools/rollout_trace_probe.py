"""A few rollouts at 4096 envs for a rocprofv3 --kernel-trace timeline (no update phase): where does a control step's time go?
python3 tools/rollout_trace_probe.py [envs] [workload] [clips]   (PBHC_ROLLOUT_GRAPH=0 for the step-by-step loop: per-step kernel boundaries)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
WL = sys.argv[2] if len(sys.argv) > 2 else "v1_walk"
CLIPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg, env, Algo = bench.build(N, "cuda:0", 0, WL, CLIPS)
algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0")
algo.setup()
obs = env.reset_all()
env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 6 + 2, seed=1))
algo._train_mode()
for _ in range(5):
    obs = algo._rollout_step(obs)
    algo.storage.clear()
torch.cuda.synchronize()
