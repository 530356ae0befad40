"""Golden for the MHPPO maths: one rollout (24 steps) + one _training_step (5 epochs x 4
minibatches) of the unmodified reference MHPPO on the reference env / ReplayFakeSim (CPU)."""
import numpy as np
import torch

from oracle.ref_harness import gen_golden as G
from oracle.ref_harness.gen_env_golden import V1_CFG, build_env, make_cfg, make_replay, oracle_motion_lib


def main():
    from humanoidverse.agents.mh_ppo.mh_ppo import MHPPO

    N = 8
    cfg = make_cfg(V1_CFG, N)
    # narrow hidden layers keep the fixture small; the maths under test does not depend on width
    cfg.algo.config.module_dict.actor.layer_config.hidden_dims = [64, 48, 32]
    cfg.algo.config.module_dict.critic.layer_config.hidden_dims = [64, 48, 32]
    env = build_env(cfg, seed=3)
    skel, clip, ml = oracle_motion_lib(cfg)
    torch.manual_seed(4)
    algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device="cpu")
    algo.setup()
    T = algo.num_steps_per_env
    root, qp, qv, cf = make_replay(env, ml, T + 2, seed=9, script=False)
    env.simulator.set_replay(root, qp, qv, cf, start_frame=0)
    obs_dict = env.reset_all()
    algo._train_mode()
    w0 = {"actor__" + k: v.clone() for k, v in algo.actor.state_dict().items()}
    w0.update({"critic__" + k: v.clone() for k, v in algo.critic.state_dict().items()})
    torch.manual_seed(21)
    algo.start_time = 0
    last_obs = algo._rollout_step(obs_dict)
    st = {k: getattr(algo.storage, k).clone() for k in algo.storage.stored_keys}
    torch.manual_seed(33)
    perm = torch.randperm(T * N)
    torch.manual_seed(33)
    loss = algo._training_step()
    w1 = {"actor1__" + k: v.clone() for k, v in algo.actor.state_dict().items()}
    w1.update({"critic1__" + k: v.clone() for k, v in algo.critic.state_dict().items()})
    G.save(
        "ppo_v1.npz", perm=perm, hidden_dims=np.array([64, 48, 32]), last_critic_obs=last_obs["critic_obs"],
        lr_actor=np.float64(algo.actor_learning_rate), lr_critic=np.float64(algo.critic_learning_rate),
        **{"loss__" + k: np.float64(v) for k, v in loss.items()},
        **{"st__" + k: v for k, v in st.items()}, **w0, **w1,
    )
