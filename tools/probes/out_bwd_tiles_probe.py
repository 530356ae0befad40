"""Update time per iteration (v1, 4096 envs) with k_out_bwd_mfma taking V 32-row tiles per workgroup: python3 tools/probes/out_bwd_tiles_probe.py V"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pbhc_amd import _lib
v = int(sys.argv[1])
_lib.lib().pbhc_debug_out_bwd_variant(1 | (v << 8))
cfg, env, Algo = bench.build(4096, "cuda:0", 0)
algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0"); algo.setup()
obs = env.reset_all()
env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 8 + 2, seed=1))
algo._train_mode()
ts = []
for i in range(7):
    obs = algo._rollout_step(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); algo._training_step(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(v, "update ms", sorted(ts[2:])[len(ts[2:]) // 2])
