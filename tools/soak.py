"""Soak run: N PPO iterations of each agent at 4096 envs (noise on, resets, curricula); parameters and logs must stay finite, memory flat."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for wl in ("v1_walk", "v2_teacher29", "v2_student23"):
    cfg, env, Algo = bench.build(4096, "cuda:0", 0, workload=wl)
    algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device="cuda:0")
    algo.setup()
    env.reset_all()
    env.simulator.set_replay(*bench.make_replay_on_device(env, 24 * 8 + 2, seed=1))
    torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); t0 = time.time()
    algo.learn(num_iterations=iters)
    torch.cuda.synchronize()
    params = list(algo.alg.parameters()) if hasattr(algo, "alg") else list(algo.actor.parameters()) + list(algo.critic.parameters())
    ok = all(torch.isfinite(p).all().item() for p in params)
    log = env.read_log()
    print(f"{wl}: {iters} iterations in {time.time() - t0:.1f} s, finite={ok}, reward_mean={log['reward_mean']:.4f}, avg_ep_len={log['average_episode_length']:.1f}, "
          f"reset_frac={log.get('reset_frac', float('nan'))}, mem {m0 / 2**20:.0f} -> {torch.cuda.memory_allocated() / 2**20:.0f} MiB (peak {torch.cuda.max_memory_allocated() / 2**20:.0f})", flush=True)
    del algo, env
    torch.cuda.empty_cache()
