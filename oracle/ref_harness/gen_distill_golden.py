"""Golden for the student-distillation branch of the reference ppo_mimic.PPO (learn_distill: teacher actor acting on the teacher
observation groups, DAgger-only behaviour cloning of the student; ppo_mimic.py:121-145,157-191,343-357,533-549,711-724).

A 23-DoF teacher (student tree + the obs node of obs_ppo_teacher.yaml, narrowed layers, random init) is built with the reference
PPO, its checkpoint and composed config are written to a scratch directory (our own files), and the unmodified reference student
(`teacher_model_path` set, `dagger_only: True`) runs one rollout + one _training_step_distill on the ReplayFakeSim."""
import os

import numpy as np
import torch
import yaml

from oracle.ref_harness import gen_golden as G
from oracle.ref_harness import gen_env_golden as G1
from oracle.ref_harness import gen_env_v2_golden as V
from oracle.ref_harness.gen_ppo_v2_golden import NARROW
from pbhc_amd.utils.config import _wrap, resolve, set_by_path

SCRATCH = os.path.join(G.REPO, "oracle", "_ref", "distill_teacher")


def teacher23_tree():
    c = V.unresolved_tree("student23")
    c["obs"] = _wrap(V._yaml("obs/motion_tracking/obs_ppo_teacher.yaml")["obs"])
    return c


def plain(n):
    if isinstance(n, dict):
        return {k: plain(v) for k, v in n.items()}
    if isinstance(n, list):
        return [plain(v) for v in n]
    return n


def main():
    from humanoidverse.agents.ppo.ppo_mimic import PPO

    N = 8
    os.makedirs(SCRATCH, exist_ok=True)
    # ---- teacher: reference PPO (RL mode), random init, saved with the reference's own save()
    tt = teacher23_tree()
    for k, v in dict(V.COMMON, num_envs=N, **NARROW).items():
        set_by_path(tt, k, v)
    with open(os.path.join(SCRATCH, "config.yaml"), "w") as f:
        yaml.safe_dump(plain(tt), f, sort_keys=False)
    tcfg = resolve(tt, now="golden")
    for k in list(tcfg.obs.noise_scales.keys()):
        tcfg.obs.noise_scales[k] = 0.0
    tenv = V.build_env(tcfg, seed=5)
    torch.manual_seed(6)
    teacher = PPO(env=tenv, config=tcfg.algo.config, log_dir=None, device="cpu")
    teacher.setup()
    ckpt = os.path.join(SCRATCH, "model_0.pt")
    teacher.save(ckpt)
    tsd = {k: v.clone() for k, v in teacher.alg.state_dict().items()}
    # ---- student: the shipped phuma_student config, teacher path -> our scratch checkpoint
    extra = dict(NARROW)
    extra.update({"algo.config.teacher_model_path": ckpt, "algo.config.dagger_only": True})
    cfg = V.make_cfg("student23", N, extra=extra)
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
    w0 = {"w0__" + k: v.clone() for k, v in algo.alg.state_dict().items()}
    torch.manual_seed(21)
    algo.start_time = 0
    algo.hist_encoding = True
    last_obs = algo._rollout_step(obs_dict)
    st = {k: getattr(algo.storage, k).clone() for k in algo.storage.stored_keys if not k.startswith("next_")}
    torch.manual_seed(33)
    perm = torch.randperm(T * N)
    torch.manual_seed(33)
    loss = algo._training_step_distill()
    w1 = {"w1__" + k: v.clone() for k, v in algo.alg.state_dict().items()}
    G.save("ppo_distill.npz", perm=perm, **{"loss__" + k: np.float64(v) for k, v in loss.items()},
           **{"teacher__" + k: v for k, v in tsd.items()}, **{"st__" + k: v for k, v in st.items()}, **w0, **w1,
           obs_keys=np.array(list(obs_dict.keys())), algo_obs_dims=np.array([f"{k}={v}" for k, v in algo.algo_obs_dim_dict.items()]),
           teacher_actor_obs_keys=np.array(list(env.config.obs.obs_dict["teacher_actor_obs"])),
           teacher_future_keys=np.array(list(env.config.obs.obs_dict["teacher_future_motion_targets"])))
    print("obs groups:", list(obs_dict.keys()), "loss", loss)
    V.dump_fixture_tree(teacher23_tree(), "v2_g1_23dof_teacher.yaml")


if __name__ == "__main__":
    main()
