"""Golden for the ppo_mimic.PPO maths: one rollout + one _training_step + one _training_step_dagger of the unmodified
reference PPO on the reference general-tracking env / ReplayFakeSim (CPU), teacher-29 configuration with narrowed layers."""
import numpy as np
import torch

from oracle.ref_harness import gen_golden as G
from oracle.ref_harness import gen_env_golden as G1
from oracle.ref_harness import gen_env_v2_golden as V

NARROW = {
    "algo.config.module_dict.actor.layer_config.hidden_dims": [64, 48, 32],
    "algo.config.module_dict.critic.layer_config.hidden_dims": [64, 48, 32],
    "algo.config.module_dict.actor.motion_encoder.hidden_dim": 12,
    "algo.config.module_dict.actor.motion_encoder.output_dim": 16,
    "algo.config.module_dict.actor.history_encoder.hidden_dim": 10,
    "algo.config.module_dict.actor.history_encoder.output_dim": 8,
    "algo.config.module_dict.actor.priv_encoder.layer_config.hidden_dims": [16],
    "algo.config.num_steps_per_env": 8,
}


def algo_node():
    """The algo node of config/algo/ppo_mimic.yaml (the student config carries the same file composed)."""
    return V._yaml("algo/ppo_mimic.yaml")["algo"]


def main():
    from humanoidverse.agents.ppo.ppo_mimic import PPO

    N = 8
    cfg = V.make_cfg("teacher29", N, extra=NARROW)
    env = V.build_env(cfg, seed=3)
    skel, clip, ml = G1.oracle_motion_lib(cfg)
    torch.manual_seed(4)
    algo = PPO(env=env, config=cfg.algo.config, log_dir=None, device="cpu")
    algo.setup()
    T = algo.num_steps_per_env
    root, qp, qv, cf = G1.make_replay(env, ml, T + 2, seed=9, script=False)
    env.simulator.set_replay(root, qp, qv, cf, start_frame=0)
    obs_dict = env.reset_all()
    algo._train_mode()
    algo.counter = 3500                       # priv_reg coefficient 0.05 (schedule [0, 0.1, 2000, 3000])
    algo.hist_encoding = False
    w0 = {"w0__" + k: v.clone() for k, v in algo.alg.state_dict().items()}
    torch.manual_seed(21)
    algo.start_time = 0
    last_obs = algo._rollout_step(obs_dict)
    st = {k: getattr(algo.storage, k).clone() for k in algo.storage.stored_keys if not k.startswith("next_")}
    with torch.no_grad():
        infer_hist = algo.alg.act_inference(last_obs, hist_encoding=True).clone()
        last_values = algo.alg.evaluate(last_obs).clone()
    torch.manual_seed(33)
    perm1 = torch.randperm(T * N)
    torch.manual_seed(33)
    loss1 = algo._training_step()
    w1 = {"w1__" + k: v.clone() for k, v in algo.alg.state_dict().items()}
    lr1 = algo.learning_rate
    torch.manual_seed(34)
    perm2 = torch.randperm(T * N)
    torch.manual_seed(34)
    loss2 = algo._training_step_dagger()
    w2 = {"w2__" + k: v.clone() for k, v in algo.alg.state_dict().items()}
    G.save(
        "ppo_v2.npz", perm1=perm1, perm2=perm2, lr1=np.float64(lr1), counter0=np.int64(3500), counter2=np.int64(algo.counter),
        infer_hist=infer_hist, last_values=last_values,
        **{"last__" + k: v for k, v in last_obs.items()},
        **{"loss1__" + k: np.float64(v) for k, v in loss1.items()}, **{"loss2__" + k: np.float64(v) for k, v in loss2.items()},
        **{"st__" + k: v for k, v in st.items()}, **w0, **w1, **w2,
    )
    return cfg


if __name__ == "__main__":
    main()
