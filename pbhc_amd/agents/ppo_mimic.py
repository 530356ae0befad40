"""PPO (ppo_mimic) — drop-in for the reference's KungfuBot2 general-tracking agent.

Same class surface as the reference (reference: humanoidverse/agents/ppo/ppo_mimic.py:34-975 on the BaseAlgo API):
`__init__(env, config, log_dir=None, device)`, `setup()`, `load(path)`, `save(path, infos)`, `learn()`, `inference_model`;
the checkpoint dict (`model_state_dict`, `optimizer_state_dict` in torch.optim.AdamW's format, `iter`, `infos`) and every
state_dict key are the reference's.  Select with `algo._target_: pbhc_amd.agents.ppo_mimic.PPO`.

Both branches of the reference are implemented: the teacher (RL) path — PPO on the privileged latent with the `priv_reg` term
and, every `dagger_update_freq` iterations, the DAgger regression of the history encoder (ppo_mimic.py:270-300,596-709) — and the
student distillation path (`teacher_model_path` set, `dagger_only: True`; ppo_mimic.py:121-191,313-357,533-549,711-724): the frozen
teacher actor acts on the teacher's observation groups, which are added to the env after construction as the reference does, and
the student actor is regressed onto it (DAgger-only behaviour cloning; PPO-with-distillation raises in the reference too).

MI355X-first differences (same maths, pinned by tests/golden/ppo_v2.npz): as pbhc_amd/agents/mh_ppo.py — no host
synchronisation inside an iteration, fused sample / bootstrap / GAE / loss / clip+AdamW kernels over flat parameter buffers,
the motion embedding computed once per forward for actor and critic, env observations written straight into the rollout slabs,
one flat RCCL gradient all-reduce per optimiser step when envs are sharded over ranks.
"""
from __future__ import annotations

import os
import time
from collections import deque

import torch
import torch.distributed as dist

from .. import _lib
from .. import dist as pdist
from .agent_modules import Actor, ActorCritic
from .mh_ppo import PhaseTimer, _load_checkpoint, _make_writer, policy_forward_graphs
from .modules import BaseModule, RolloutStorage, apply_cat, apply_into


class _FlatAdamWView:
    """torch.optim.AdamW-format state_dict()/load_state_dict() over a set of parameters of the flat buffers."""

    def __init__(self, algo, which):
        self.algo, self.which = algo, which            # which: 0 = self.optimizer (all parameters), 1 = hist_encoder_optimizer

    def _entries(self):
        a = self.algo
        names = [n for n, _ in a.alg.named_parameters()]
        if self.which == 1:
            names = [n for n in names if n.startswith("actor_module.history_encoder.")]
        elif a.dagger_only:                                  # optim.AdamW(self.alg.actor.parameters()) (ppo_mimic.py:186-187)
            names = [n for n in names if n.startswith("actor_module.")]
        return names

    def state_dict(self):
        a = self.algo
        state = {}
        stepped = float(a._adam_step[self.which]) > 0
        for i, n in enumerate(self._entries()):
            if not stepped or (self.which == 0 and not a._is_main(n)):       # never stepped: torch keeps no state for it
                continue
            o, k, shape = a._slice_of[n]
            state[i] = {"step": a._adam_step[self.which].detach().clone().cpu(), "exp_avg": a._mflat[self.which][o:o + k].view(shape).clone(),
                        "exp_avg_sq": a._vflat[self.which][o:o + k].view(shape).clone()}
        group = {"lr": float(a._lr[0]) if self.which == 0 else float(a._lr_hist[0]), "betas": tuple(a.betas), "eps": a.adam_eps, "weight_decay": a.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self._entries())))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        a = self.algo
        for i, n in enumerate(self._entries()):
            if i in sd["state"]:
                e = sd["state"][i]
                o, k, _ = a._slice_of[n]
                a._mflat[self.which][o:o + k].copy_(e["exp_avg"].reshape(-1).to(a.device))
                a._vflat[self.which][o:o + k].copy_(e["exp_avg_sq"].reshape(-1).to(a.device))
                a._adam_step[self.which] = float(e["step"])
        if self.which == 0:
            a._lr[:] = float(sd["param_groups"][0]["lr"])

    @property
    def param_groups(self):
        return [{"lr": float(self.algo._lr[0])}]


class PPO:
    def __init__(self, env, config, log_dir=None, device="cpu"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PbhcError("pbhc_amd.agents.ppo_mimic.PPO runs on the GPU only")
        self.env = env
        self.config = config
        self.log_dir = log_dir
        self.writer = _make_writer(log_dir)
        self.start_time = self.stop_time = 0
        self.collection_time = self.learn_time = 0
        self._timer = PhaseTimer()
        self._init_config()
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.ep_infos = []
        self.rewbuffer = deque(maxlen=100)
        self.lenbuffer = deque(maxlen=100)
        N = self.env.num_envs
        self.cur_reward_sum = torch.zeros(N, dtype=torch.float, device=self.device)
        self.cur_episode_length = torch.zeros(N, dtype=torch.float, device=self.device)
        self._ep_stats = torch.zeros(3, dtype=torch.float64, device=self.device)
        self.world_size, self.rank = pdist.world(), pdist.rank()
        self._dp = pdist.active()                    # data-parallel exchanges on (more than one rank, or a forced one-rank rehearsal)
        # algo.config.sync_env_statistics: "rollout" (default; True means the same) | "step" (exact single-process equivalence) | False
        self._stat_mode = {True: "rollout", False: None, None: None}.get(config.get("sync_env_statistics", "rollout"), config.get("sync_env_statistics", "rollout"))
        if self._dp and self._stat_mode and hasattr(self.env, "enable_global_statistics"):
            self.env.enable_global_statistics(mode=self._stat_mode)     # sigma / episode-length curricula from the batch of all ranks' envs
        _ = self.env.reset_all()
        self.learn = self.learn_RL if not self.train_distill else self.learn_distill

    def _init_config(self):
        c = self.config
        self.num_envs = self.env.config.num_envs
        self.algo_obs_dim_dict = self.env.config.robot.algo_obs_dim_dict
        self.num_act = self.env.config.robot.actions_dim
        self.save_interval = c.save_interval
        self.logging_interval = c.get("logging_interval", 10)
        self.num_steps_per_env = c.num_steps_per_env
        self.load_optimizer = c.load_optimizer
        self.num_learning_iterations = c.num_learning_iterations
        self.init_at_random_ep_len = c.init_at_random_ep_len
        self.desired_kl = c.desired_kl
        self.schedule = c.schedule
        self.learning_rate = c.learning_rate
        self.clip_param = c.clip_param
        self.num_learning_epochs = c.num_learning_epochs
        self.num_mini_batches = c.num_mini_batches
        self.gamma = c.gamma
        self.lam = c.lam
        self.value_loss_coef = c.value_loss_coef
        self.entropy_coef = c.entropy_coef
        self.max_grad_norm = c.max_grad_norm
        self.use_clipped_value_loss = c.use_clipped_value_loss
        self.num_rew_fn = self.env.num_rew_fn
        self.priv_reg_coef_schedual = c.priv_reg_coef_schedual
        self.counter = 0
        self.train_distill = c.get("teacher_model_path", None) is not None
        self.dagger_only = bool(c.get("dagger_only", False))
        if self.dagger_only != self.train_distill:
            # the reference: distillation without dagger_only raises NotImplementedError in _update_distill (ppo_mimic.py:724);
            # dagger_only without a teacher stores no values/log-probs and fails in _update_ppo
            raise NotImplementedError("ppo_mimic supports dagger_only=False without a teacher (RL) or dagger_only=True with teacher_model_path (distillation)")
        if self.train_distill:
            self._preprocess_teacher_config()
        if c.module_dict.get("actor", {}).get("type", "MLP") != "MLP" or c.module_dict.get("critic", {}).get("type", "MLP") != "MLP":
            raise NotImplementedError("MoEMLP actors/critics")
        self.dagger_update_freq = c.get("dagger_update_freq", 20)
        self.hist_encoding = False

    def _preprocess_teacher_config(self):
        """ppo_mimic.py:121-145: the teacher's actor / future-target observation groups are added to the env under `teacher_*` names and
        the teacher's module config is rewritten to read them."""
        from pathlib import Path

        from ..envs import env_config
        from ..utils.config import load_config

        tc = load_config(str(Path(self.config.teacher_model_path).parent / "config.yaml"))
        groups, _, _ = env_config.determine_obs_dim(tc)
        od = self.env.config.obs.obs_dict
        od["teacher_actor_obs"] = list(tc.obs.obs_dict["actor_obs"])
        od["teacher_future_motion_targets"] = list(tc.obs.obs_dict["future_motion_targets"])
        self.env.rebuild_observations()
        self.algo_obs_dim_dict = self.env.config.robot.algo_obs_dim_dict
        assert self.algo_obs_dim_dict["teacher_actor_obs"] == groups["actor_obs"] and self.algo_obs_dim_dict["teacher_future_motion_targets"] == groups["future_motion_targets"]
        md = tc.algo.config.module_dict
        swap = lambda dims: ["teacher_actor_obs" if x == "actor_obs" else x for x in dims]
        md.actor.input_dim = swap(list(md.actor.input_dim))
        md.critic.input_dim = swap(list(md.critic.input_dim))
        md.actor.motion_encoder.input_dim = ["teacher_future_motion_targets"]
        self.config.teacher_module_dict = md

    # ------------------------------------------------------------------------------------
    def setup(self):
        from .gemm_tuning import enable as _enable_gemm_tuning

        _enable_gemm_tuning()
        self._setup_models_and_optimizer()
        self._setup_storage()

    def _setup_models_and_optimizer(self):
        c = self.config
        if self.env.config.use_vec_reward:
            c.module_dict.critic["output_dim"][-1] = self.num_rew_fn
        if self.train_distill:
            self.teacher_actor = Actor(self.algo_obs_dim_dict, c.teacher_module_dict.actor, self.num_act).to(self.device)
            sd = torch.load(c.teacher_model_path, map_location=self.device, weights_only=True)
            self.teacher_actor.load_state_dict({k[len("actor_module."):]: v for k, v in sd["model_state_dict"].items() if k.startswith("actor_module.")}, strict=True)
            for p in self.teacher_actor.parameters():
                p.requires_grad = False
            self.teacher_actor.eval()
        self.alg = ActorCritic(self.algo_obs_dim_dict, c.module_dict, self.num_act, c.init_noise_std).to(self.device)
        if self.train_distill:
            self.alg.actor.history_encoder.load_state_dict(self.teacher_actor.history_encoder.state_dict())
            for p in self.alg.actor.history_encoder.parameters():
                p.requires_grad_(False)
        if self._dp:
            for p in self.alg.parameters():
                pdist.broadcast(p.data, src=0)
        self._flatten_parameters()

    def _flatten_parameters(self):
        """Flat fp32 buffers [main parameters | history-encoder parameters]: `self.optimizer` (AdamW over every parameter; the history
        encoder never has a gradient in the PPO step, so torch skips it) steps the first segment, `hist_encoder_optimizer` the second."""
        dev = self.device
        named = list(self.alg.named_parameters())
        if self.dagger_only:       # distillation: only the actor MLP and the motion encoder ever have a gradient (ppo_mimic.py:711-721)
            self._is_main = lambda n: n.startswith("actor_module.actor_module.") or n.startswith("actor_module.motion_encoder.")
        else:
            self._is_main = lambda n: not n.startswith("actor_module.history_encoder.")
        main = [(n, p) for n, p in named if self._is_main(n)]
        hist = [(n, p) for n, p in named if not self._is_main(n)]
        self._n_main = sum(p.numel() for _, p in main)
        self._n_hist = sum(p.numel() for _, p in hist)
        # four spare floats between the two segments (in every flat buffer, so that one offset serves them all): slot `_n_main` of the
        # GRADIENT buffer carries the minibatch KL mean through the data-parallel all-reduce of the main segment — one collective per
        # optimiser step, every rank then takes the same learning-rate branch (as MHPPO does, mh_ppo.py: `_gflat[na + nc]`)
        self._o_hist = self._n_main + 4
        n = self._o_hist + self._n_hist
        self._pflat = torch.zeros(n, device=dev)
        self._gflat = torch.zeros(n, device=dev)
        self._mflat = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]     # per optimiser (the hist segment of [0] stays unused)
        self._vflat = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
        self._slice_of = {}
        o = 0
        for nme, p in main + hist:
            if hist and p is hist[0][1]:
                o = self._o_hist
            k = p.numel()
            self._pflat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self._pflat[o:o + k].view_as(p)
            p.grad = self._gflat[o:o + k].view_as(p)
            self._slice_of[nme] = (o, k, tuple(p.shape))
            o += k
        self._std_slice = self._slice_of["std"][:2]
        self._main_has_std = self._is_main("std")
        self._lr = torch.full((2,), float(self.learning_rate), device=dev)          # [0] is THE learning rate (the loss kernel adapts both)
        self._lr_hist = torch.full((1,), float(self.learning_rate), device=dev)     # hist_encoder_optimizer keeps its initial lr (ppo_mimic.py:184)
        self._adam_step = torch.zeros(2, device=dev)
        self._adam_scratch = torch.zeros(2, 512, dtype=torch.float64, device=dev)
        self._grad_norms = torch.zeros(2, device=dev)
        self._loss_scalars = torch.zeros(4, device=dev)
        self._g_sigma = torch.zeros(self.num_act, device=dev)
        self.betas, self.adam_eps, self.weight_decay = (0.9, 0.999), 1e-8, 0.01     # torch.optim.AdamW defaults
        self.optimizer = _FlatAdamWView(self, 0)
        self.hist_encoder_optimizer = _FlatAdamWView(self, 1)
        # every update zeroes its segment of `_gflat` before its backward: the MLP stacks may store their gradients into it directly
        from . import fused_mlp
        from .modules import BaseModule

        self._direct_stacks = {"main": [], "hist": []}
        lo = self._gflat.data_ptr()
        for m in self.alg.modules():
            if isinstance(m, BaseModule):
                fused_mlp.grad_direct(m.module)
                off = (next(m.module.parameters()).grad.data_ptr() - lo) // 4
                self._direct_stacks["main" if off < self._n_main else "hist"].append(m.module)

    def _zero_grads(self, which):
        """zero one segment of the flat gradient buffer ("main": everything but the history encoder, "hist": the history encoder) and tell
        the declared stacks living in it that their next backward may store instead of accumulate"""
        from . import fused_mlp

        (self._gflat[: self._n_main] if which == "main" else self._gflat[self._o_hist:]).zero_()
        for q in self._direct_stacks[which]:
            fused_mlp.grads_zeroed(q)

    def _setup_storage(self):
        st = self.storage = RolloutStorage(self.env.num_envs, self.num_steps_per_env, self.device)
        S = len(self.env.tar_obs_steps)
        self._obs_width = {}
        for k, d in self.algo_obs_dim_dict.items():
            w = d * S if k in ("future_motion_targets", "teacher_future_motion_targets") else d      # ppo_mimic.py:206-216
            self._obs_width[k] = w
            st.register_key(k, shape=(w,), dtype=torch.float, pad_rows=True, tail_slab=True)
        st.register_key("actions", shape=(self.num_act,), dtype=torch.float)
        st.register_key("rewards", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("dones", shape=(1,), dtype=torch.bool)
        st.register_key("values", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("returns", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("advantages", shape=(1,), dtype=torch.float)
        st.register_key("actions_log_prob", shape=(1,), dtype=torch.float)
        st.register_key("action_mean", shape=(self.num_act,), dtype=torch.float)
        st.register_key("action_sigma", shape=(self.num_act,), dtype=torch.float)
        if self.train_distill:
            st.register_key("teacher_actions", shape=(self.num_act,), dtype=torch.float)
        T, N = self.num_steps_per_env, self.env.num_envs
        self._gae_stats = torch.zeros(2 * ((T * N + 255) // 256) + 4, dtype=torch.float64, device=self.device)
        self._last_obs = {k: st.with_tail(k)[T] for k in self._obs_width}      # the observations after the last step: slab T of the same buffers
        self._sample_seed = pdist.rank_seed(int(torch.randint(0, 2**62, (1,)).item()))
        if not hasattr(self.env, "globals") or not hasattr(self.env, "set_obs_outputs"):
            raise _lib.PbhcError("pbhc_amd PPO drives the fused pbhc_amd env (needs env.globals / env.set_obs_outputs)")
        self._mb = (T * N) // self.num_mini_batches
        self._loss_scratch = torch.zeros(_lib.lib().pbhc_ppo_loss_scratch_floats(self._mb), device=self.device)
        self._grad_mu = torch.zeros(self._mb, self.num_act, device=self.device)
        self._grad_value = torch.zeros(self._mb, self.num_rew_fn, device=self.device)

    def _eval_mode(self):
        self.alg.eval()

    def _train_mode(self):
        self.alg.train()

    # ---- checkpoints (ppo_mimic.py:237-265) --------------------------------------------------
    def load(self, ckpt_path):
        if ckpt_path is None:
            return None
        d = _load_checkpoint(ckpt_path, self.device)
        self.alg.load_state_dict(d["model_state_dict"])
        if self.load_optimizer:
            self.optimizer.load_state_dict(d["optimizer_state_dict"])
            self.learning_rate = d["optimizer_state_dict"]["param_groups"][0]["lr"]
            self.set_learning_rate(self.learning_rate)
        self.current_learning_iteration = d["iter"]
        return d["infos"]

    def save(self, path, infos=None):
        torch.save({"model_state_dict": self.alg.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def set_learning_rate(self, learning_rate):
        self.learning_rate = learning_rate
        self._lr[:] = float(learning_rate)

    def update_counter(self):
        self.counter += 1

    # ---- learn loop (ppo_mimic.py:267-311) ---------------------------------------------------
    def learn_RL(self, num_iterations=None):
        if self.init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        obs_dict = self.env.reset_all()
        self._train_mode()
        n = self.num_learning_iterations if num_iterations is None else num_iterations
        tot_iter = self.current_learning_iteration + n
        for it in range(self.current_learning_iteration, tot_iter):
            self.hist_encoding = it % self.dagger_update_freq == 0
            self._timer.start()
            obs_dict = self._rollout_step(obs_dict)           # ends with _timer.split(): collection | learning
            loss_dict = self._training_step()
            if self.hist_encoding:
                loss_dict = self._training_step_dagger()
            self._timer.split()
            self._post_epoch_logging(dict(it=it, loss_dict=loss_dict, num_learning_iterations=n))
            if self.log_dir is not None and it % self.save_interval == 0 and self.rank == 0:
                self.current_learning_iteration = it
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
            self.ep_infos.clear()
        self.current_learning_iteration = tot_iter
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    # ---- distillation (ppo_mimic.py:313-357,533-549,711-724) ----------------------------------
    def learn_distill(self, num_iterations=None):
        if self.init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        obs_dict = self.env.reset_all()
        self._train_mode()
        n = self.num_learning_iterations if num_iterations is None else num_iterations
        tot_iter = self.current_learning_iteration + n
        for it in range(self.current_learning_iteration, tot_iter):
            self.hist_encoding = True
            self._timer.start()
            obs_dict = self._rollout_step_distill(obs_dict)
            loss_dict = self._training_step_distill()
            self._timer.split()
            self._post_epoch_logging(dict(it=it, loss_dict=loss_dict, num_learning_iterations=n))
            if self.log_dir is not None and it % self.save_interval == 0 and self.rank == 0:
                self.current_learning_iteration = it
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
            self.ep_infos.clear()
        self.current_learning_iteration = tot_iter
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def teacher_actor_act_step(self, obs_dict, hist_encoding=True):
        return self.teacher_actor(obs_dict, hist_encoding, obs_key="teacher_actor_obs", target_key="teacher_future_motion_targets")

    def _rollout_step_distill(self, obs_dict):
        """DAgger-only rollout: the student acts with its mean on the history latent, the teacher's actions are recorded."""
        st, env, lib = self.storage, self.env, _lib.lib()
        T, N, R = self.num_steps_per_env, env.num_envs, self.num_rew_fn
        keys = list(self._obs_width.keys())
        stream = _lib.current_stream()
        with torch.inference_mode():
            for k in keys:
                getattr(st, k)[0].copy_(obs_dict[k])
            for t in range(T):
                b = {k: getattr(st, k)[t] for k in keys}
                st.teacher_actions[t].copy_(self.teacher_actor_act_step(b, hist_encoding=True))
                st.actions[t].copy_(self.alg.act_inference(b, hist_encoding=True))
                env.set_obs_outputs({k: getattr(st, k)[t + 1] for k in keys} if t + 1 < T else self._last_obs)
                nxt, rewards, dones, infos = env.step({"actions": st.actions[t]})
                _lib.check(lib.pbhc_rollout_post(rewards.data_ptr(), st.values[t].data_ptr(), dones.data_ptr(), infos["time_outs"].data_ptr(), N, R,
                                                 0.0, st.rewards[t].data_ptr(), st.dones[t].data_ptr(), self.cur_reward_sum.data_ptr(),
                                                 self.cur_episode_length.data_ptr(), self._ep_stats.data_ptr(), stream), "pbhc_rollout_post")
            st.step = T
            if self._dp and self._stat_mode == "rollout":
                self.env.sync_globals()                # sigma / curricula / log means: the mean over the ranks, once per rollout
            self._timer.split()
        return self._last_obs

    def _training_step_distill(self, indices=None):
        loss = {"bc_loss": torch.zeros((), device=self.device)}
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs,
                                                       keys=["actor_obs", "future_motion_targets", "prop_history", "teacher_actions"], indices=indices):
            self._update_distill(batch, loss)
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        out = {k: v / n for k, v in loss.items()}
        for k in ["Value", "Entropy", "Surrogate", "Actor_Load_Balancing_Loss", "Critic_Load_Balancing_Loss"]:
            out[k] = torch.zeros((), device=self.device)
        return out

    def _update_distill(self, b, loss):
        mu = self.alg.act_inference(b, hist_encoding=True)
        bc = (b["teacher_actions"] - mu).norm(p=2, dim=1).mean()
        self._zero_grads("main")
        bc.backward()
        if self._dp:
            pdist.allreduce_mean_(self._gflat[: self._n_main])
        self._adam(0, 0, self._n_main, self._lr[0:1])
        loss["bc_loss"] += bc.detach()
        return loss

    # ---- forward pieces ---------------------------------------------------------------------
    def _forward(self, b, hist_encoding, want_value=True):
        """mu, value (and the motion embedding computed ONCE for both; the reference encodes it twice with the same weights)."""
        a = self.alg.actor
        emb = a.motion_encoding(b["future_motion_targets"])
        latent = a.history_encoding(b["prop_history"]) if hist_encoding else a.priv_encoding(b["priv_obs"])
        # (the stacks read [observations | encoder outputs]: only the encoder columns carry a gradient — modules.apply_cat / apply_into)
        if "_xin_actor" in b and torch.is_grad_enabled():
            # the update: the observation columns of both stacks' inputs were laid out once, behind the minibatch shuffle (_assemble_inputs);
            # per optimiser step only the encoder outputs are copied in
            mu = apply_into(a.actor_module, b["_xin_actor"], b["actor_obs"].shape[1], [emb, latent])
            value = apply_into(self.alg.critic, b["_xin_critic"], b["actor_obs"].shape[1] + b["priv_obs"].shape[1], [emb]) if want_value else None
            return mu, value, latent
        mu = apply_cat(a.actor_module, b["actor_obs"], torch.cat([emb, latent], dim=-1))
        value = apply_cat(self.alg.critic, torch.cat([b["actor_obs"], b["priv_obs"]], dim=-1), emb) if want_value else None
        return mu, value, latent

    def _assemble_inputs(self, shuffled):
        """once per update, on the shuffled [T * N, C] tensors: the actor's and the critic's first-layer inputs with their observation columns in
        place — [actor_obs | (motion embedding, latent)] and [actor_obs | priv_obs | (motion embedding)] — so that an optimiser step copies 19 + 13 MB
        of encoder outputs where four `torch.cat` moved 110 MB (PBHC_ASSEMBLE_INPUTS=0: the concatenations)"""
        if os.environ.get("PBHC_ASSEMBLE_INPUTS", "1") == "0" or not all(k in shuffled for k in ("actor_obs", "priv_obs")):
            return
        a = self.alg.actor
        ao, po = shuffled["actor_obs"], shuffled["priv_obs"]
        if not (isinstance(a.actor_module, BaseModule) and a.actor_module._fused and isinstance(self.alg.critic, BaseModule) and self.alg.critic._fused and ao.is_cuda):
            return
        rows, wa, wp = ao.shape[0], ao.shape[1], po.shape[1]
        ka, kc = a.actor_module.module[0].in_features, self.alg.critic.module[0].in_features
        bufs = self.__dict__.get("_xin")
        if bufs is None or bufs[0].shape != (rows, ka) or bufs[1].shape != (rows, kc) or bufs[0].device != ao.device:
            bufs = self._xin = (torch.empty(rows, ka, device=ao.device), torch.empty(rows, kc, device=ao.device))
        bufs[0][:, :wa].copy_(ao)
        bufs[1][:, :wa].copy_(ao)
        bufs[1][:, wa:wa + wp].copy_(po)
        shuffled["_xin_actor"], shuffled["_xin_critic"] = bufs

    def _rollout_step(self, obs_dict):
        """ppo_mimic.py:371-438.  Per control step: encoders + actor + critic forward, ONE sample/log-prob/buffer-write kernel, the fused env
        step writing the next observations into the next rollout slab, ONE bootstrap/done/episode-stat kernel.  Round 4, as MHPPO's rollout
        (mh_ppo.py `_rollout_step`): the MLP stacks (actor, critic, privileged encoder) run as ONE launch each from packed weights — constant
        over the rollout — and the T control steps are ONE captured hipGraph (fork / join edges instead of stream events; the steps read the
        replay frame from the device-side cursor); the first rollout, and any rollout the env cannot promise to be free of host-side events
        (`rollout_graph_safe`), runs the loop step by step with one captured forward per step."""
        from . import fused_mlp
        from .mh_ppo import MHPPO

        st, env, lib = self.storage, self.env, _lib.lib()
        T, N, A, R = self.num_steps_per_env, env.num_envs, self.num_act, self.num_rew_fn
        keys = list(self._obs_width.keys())
        K = _lib.K
        counter = env.globals[K["PBHC_G_STEP_COUNTER"]:].data_ptr()
        with torch.inference_mode():
            sigma = self.__dict__.get("_sigma_buf")
            if sigma is None:
                sigma = self._sigma_buf = torch.empty(A, device=self.device)          # fixed address: the captured sampling kernel reads it
            sigma.copy_(self.alg.sigma())
            for k in keys:
                getattr(st, k)[0].copy_(obs_dict[k])
            mode = bool(self.hist_encoding)                    # the captured forward depends on the latent source
            a = self.alg.actor
            split = os.environ.get("PBHC_ROLLOUT_SPLIT", "1") != "0" and hasattr(env, "set_finalize_stream")
            # The critic's values feed only the time-out bootstrap and GAE, both after the rollout: evaluated ONCE over all T + 1 slabs
            # (whole-chip GEMM tiles) on the motion embeddings the per-step forwards left in `_emb_buf`, as MHPPO does (PBHC_CRITIC_BATCHED=0:
            # the critic stack inside every control step)
            batched = split and os.environ.get("PBHC_CRITIC_BATCHED", "1") != "0"
            nets = (a.actor_module, a.priv_encoder) if batched else (a.actor_module, self.alg.critic, a.priv_encoder)
            stacks = [m.module for m in nets if m is not None and isinstance(m, BaseModule) and m._fused]
            stacks = [q for q in stacks if os.environ.get("PBHC_STACK_NETS_V2", "1") != "0" and fused_mlp.pack_stack(q)]
            encoders = [e for e in (a.motion_encoder, a.history_encoder if mode else None) if e is not None and hasattr(e, "prepare_inference")]
            for e in encoders:
                e.prepare_inference()                          # weights re-laid-out once per rollout, in place (the captured graph reads them)
            if batched:
                E = a.motion_encoder.output_dim
                if self.__dict__.get("_emb_buf") is None or self._emb_buf.shape != (T + 1, N, E):
                    self._emb_buf = torch.zeros(T + 1, N, E, device=self.device)
                    self._time_outs = torch.zeros(T, N, 1, dtype=torch.bool, device=self.device)
            fuse_sample = False
            try:
                if batched:
                    # the actor stack reads [actor_obs | motion embedding | latent] as three column segments (no concatenated copy) and samples in
                    # its last epilogue, keyed by a snapshot of the step counter + the step index: the keys pbhc_policy_sample forms from the live
                    # counter, without waiting for the previous step's reduction (as MHPPO; PBHC_FUSED_SAMPLE=0: the separate sampling kernel)
                    a_seq = a.actor_module.module if isinstance(a.actor_module, BaseModule) else None
                    cat_ok = a_seq is not None and any(q is a_seq for q in stacks)
                    fuse_sample = cat_ok and os.environ.get("PBHC_FUSED_SAMPLE", "1") != "0"
                    if fuse_sample:
                        if self.__dict__.get("_ctr0") is None:
                            self._ctr0 = torch.zeros(1, dtype=torch.float64, device=self.device)
                        env.wait_finalize()
                        self._ctr0.copy_(env.globals[K["PBHC_G_STEP_COUNTER"]:K["PBHC_G_STEP_COUNTER"] + 1])

                    def eager_fwd(t):
                        b = {k: getattr(st, k)[t] for k in keys}
                        emb = a.motion_encoder(b["future_motion_targets"], out=self._emb_buf[t])
                        latent = a.history_encoding(b["prop_history"]) if mode else a.priv_encoding(b["priv_obs"])
                        xs = [b["actor_obs"], emb, latent]
                        if fuse_sample:
                            smp = dict(std=sigma, seed=self._sample_seed, counter=self._ctr0.data_ptr(), counter_offset=t, actions=st.actions[t],
                                       action_mean=st.action_mean[t], action_sigma=st.action_sigma[t], logp=st.actions_log_prob[t])
                            if fused_mlp.forward_cat_inference(a_seq, xs, sample=smp) is False:
                                raise _lib.PbhcError("pbhc_mlp_fwd_cat does not apply to this policy (PBHC_FUSED_SAMPLE=0)")
                            return st.action_mean[t], None
                        mu = fused_mlp.forward_cat_inference(a_seq, xs) if cat_ok else False
                        return (a.actor_module(torch.cat(xs, dim=-1)) if mu is False else mu), None
                else:
                    eager_fwd = lambda t: self._forward({k: getattr(st, k)[t] for k in keys}, mode)[:2]
                # the dependent chain of a control step is env step -> policy forward -> sampling -> env step; the env step's one-workgroup
                # reduction and the bootstrap / episode-statistics kernel run next to the policy forward on a branch stream (joined before the
                # sampling kernel, which reads the step counter the reduction advances)
                cur = torch.cuda.current_stream()
                br = None
                if split:
                    br = self.__dict__.get("_branch_stream") or torch.cuda.Stream(device=self.device)
                    self._branch_stream = br
                    env.set_finalize_stream(br)
                post_done = [self.__dict__.setdefault("_post_done", torch.cuda.Event())]
                sc = self.__dict__.get("_step_ptrs")
                if sc is None or sc[0] is not st or sc[2] != batched:
                    P = lambda x: x.data_ptr()
                    sc = (st, [dict(sample=(P(st.actions[t]), P(st.action_mean[t]), P(st.action_sigma[t]), P(st.actions_log_prob[t]), None if batched else P(st.values[t])),
                                    post=(P(st.rewards[t]), P(st.dones[t])), values=None if batched else P(st.values[t]),
                                    tout=P(self._time_outs[t]) if batched else None, act={"actions": st.actions[t]},
                                    obs_out={k: getattr(st, k)[t + 1] for k in keys} if t + 1 < T else self._last_obs) for t in range(T)], batched)
                    self._step_ptrs = sc
                steps = sc[1]
                sum_p, len_p, stat_p, gamma = self.cur_reward_sum.data_ptr(), self.cur_episode_length.data_ptr(), self._ep_stats.data_ptr(), float(self.gamma)

                def run_loop(cur, fwd_call):
                    stream = cur.cuda_stream
                    if split:
                        br.wait_stream(cur)
                    for t in range(T):
                        sp = steps[t]
                        mu, value = fwd_call(t)
                        if split and t > 0:
                            cur.wait_event(post_done[0])
                            env.finalize_joined()
                        if not fuse_sample:
                            _lib.check(lib.pbhc_policy_sample(mu.data_ptr(), sigma.data_ptr(), None if batched else value.data_ptr(), N, A, R, self._sample_seed, counter,
                                                              *sp["sample"], stream), "pbhc_policy_sample")
                        env.set_obs_outputs(sp["obs_out"])
                        nxt, rewards, dones, infos = env.step(sp["act"])
                        ps = br.cuda_stream if split else stream
                        # (batched critic: values == NULL — the time-out bootstrap is added after the loop — and the step's time-out flags are kept)
                        _lib.check(lib.pbhc_rollout_post2(rewards.data_ptr(), sp["values"], dones.data_ptr(), infos["time_outs"].data_ptr(), N, R, gamma, *sp["post"],
                                                          sum_p, len_p, stat_p, sp["tout"], ps), "pbhc_rollout_post2")
                        if split:
                            post_done[0].record(br)
                    if split:
                        cur.wait_stream(br)

                graph_ok = (os.environ.get("PBHC_ROLLOUT_GRAPH", "1") != "0" and split and self.__dict__.get("_rollouts_done", 0) >= 1
                            and not self.__dict__.get("_rollout_graph_failed", False) and hasattr(env, "rollout_graph_safe") and env.rollout_graph_safe(T))
                ran = False
                if graph_ok:
                    env.simulator.use_device_cursor()
                    key = (id(st), env._io_epoch, env.simulator.replay_version, bool(getattr(env, "is_specialised", False)), N, T, mode, bool(stacks), batched, fuse_sample)
                    gc = self.__dict__.get("_rollout_graph")
                    if gc is None or gc[0] != key:
                        gc = MHPPO._capture_rollout(self, key, run_loop, eager_fwd, env, post_done, T)
                    if gc is not None:
                        gc[1].replay()
                        env.after_graph_steps(T)
                        ran = True
                self._rollout_used_graph = ran
                if not ran:
                    run_loop(cur, policy_forward_graphs(self, eager_fwd, key=(mode, bool(stacks), batched, fuse_sample)))      # (one captured forward per step)
                if split:
                    env.set_finalize_stream(None)
            finally:
                for q in stacks:
                    fused_mlp.release_stack(q)
                for e in encoders:
                    e.release_inference()
            last_values = None
            if batched:
                # ppo_mimic.py:384-386, 425-431 for all steps at once: values of every slab + the bootstrap values of GAE (slab T: the
                # observations after the last step) from one launch set, then rewards += gamma * values * time_outs
                self._emb_buf[T].copy_(a.motion_encoding(self._last_obs["future_motion_targets"]))
                rows = lambda k: st.with_tail(k).flatten(0, 1)
                vals = self.alg.critic(torch.cat([rows("actor_obs"), rows("priv_obs"), self._emb_buf.flatten(0, 1)], dim=-1)).view(T + 1, N, R)
                st.values.copy_(vals[:T])
                st.rewards.addcmul_(st.values, self._time_outs.to(torch.float32), value=float(self.gamma))
                last_values = vals[T]
            st.step = T
            self._rollouts_done = self.__dict__.get("_rollouts_done", 0) + 1
            if self._dp and self._stat_mode == "rollout":
                self.env.sync_globals()                # sigma / curricula / log means: the mean over the ranks, once per rollout
            self._timer.split()
            self._compute_returns(self._last_obs, last_values=last_values)
        return self._last_obs

    def _compute_returns(self, last_obs_dict, last_values=None):
        """ppo_mimic.py:443-491 in one HIP pass (scalar reward: R = 1, normalisation over all [T,N] entries)."""
        st = self.storage
        if last_values is None:
            last_values = self.alg.evaluate(last_obs_dict).detach()
        last_values = last_values.contiguous()
        T, N, R = self.num_steps_per_env, self.env.num_envs, self.num_rew_fn
        adv = st.advantages
        _lib.check(_lib.lib().pbhc_gae(st.rewards.data_ptr(), st.values.data_ptr(), st.dones.data_ptr(), last_values.data_ptr(), T, N, R,
                                       float(self.gamma), float(self.lam), st.returns.data_ptr(), adv.data_ptr(), self._gae_stats.data_ptr(),
                                       _lib.current_stream()), "pbhc_gae")
        if self._dp:
            nb = (T * N + 255) // 256
            mean_l, std_l = self._gae_stats[2 * nb].float(), self._gae_stats[2 * nb + 1].float()
            adv.copy_(pdist.global_normalize_(adv * (std_l + 1e-8) + mean_l))
        return st.returns, adv

    # ---- updates (ppo_mimic.py:493-709) ------------------------------------------------------
    UPDATE_KEYS = ["actor_obs", "priv_obs", "future_motion_targets", "prop_history", "actions", "values", "advantages", "returns", "actions_log_prob",
                   "action_mean", "action_sigma"]

    def _training_step(self, indices=None):
        names = ["Value", "Entropy", "Surrogate", "priv_reg_loss", "Actor_Load_Balancing_Loss", "Critic_Load_Balancing_Loss"]
        meters = torch.zeros(len(names) + 4, device=self.device)          # one fill: the meters and, behind them, the loss kernel's running sums
        loss = {k: meters[i] for i, k in enumerate(names[:4])}
        loss["_acc"] = meters[len(names):]                     # {surrogate, value, entropy, kl} summed by the loss kernel itself
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, keys=self.UPDATE_KEYS, indices=indices,
                                                       on_gather=self._assemble_inputs):
            self._update_ppo(batch, loss)
        acc = loss.pop("_acc")
        loss["Surrogate"] += acc[0]; loss["Value"] += acc[1]; loss["Entropy"] += acc[2]
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        self.update_counter()
        self.learning_rate = self._lr[0:1]
        means = meters[:len(names)] / n                        # (the two load-balancing meters stay zero: no mixture-of-experts stacks here)
        return {k: means[i] for i, k in enumerate(names)}

    def _training_step_dagger(self, indices=None):
        loss = {"hist_latent_loss": torch.zeros((), device=self.device)}
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, keys=["priv_obs", "prop_history"], indices=indices):
            self._update_dagger(batch, loss)
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        self.update_counter()
        return {k: v / n for k, v in loss.items()}

    def _adam(self, which, o, n, lr):
        _lib.check(_lib.lib().pbhc_adam_clip(self._pflat[o:o + n].data_ptr(), self._gflat[o:o + n].data_ptr(), self._mflat[which][o:o + n].data_ptr(),
                                             self._vflat[which][o:o + n].data_ptr(), n, lr.data_ptr(), self._adam_step[which:which + 1].data_ptr(),
                                             float(self.max_grad_norm), self.betas[0], self.betas[1], self.adam_eps, self.weight_decay,
                                             self._adam_scratch[which].data_ptr(), self._grad_norms[which:which + 1].data_ptr(), _lib.current_stream()), "pbhc_adam_clip")

    def _update_ppo(self, b, loss):
        lib = _lib.lib()
        alg = self.alg
        mu, value, priv_latent = self._forward(b, hist_encoding=False)
        with torch.no_grad():
            hist_latent = alg.actor.history_encoding(b["prop_history"])
        priv_reg = (priv_latent - hist_latent).norm(p=2, dim=1).mean()
        sch = self.priv_reg_coef_schedual
        stage = min(max(self.counter - sch[2], 0) / sch[3], 1)
        coef = stage * (sch[1] - sch[0]) + sch[0]
        B = mu.shape[0]
        if B != self._mb:
            raise _lib.PbhcError("minibatch size changed")
        sigma = alg.sigma().detach().contiguous()
        self._zero_grads("main")
        adapt = int(self.desired_kl is not None and self.schedule == "adaptive")
        flags = (adapt if not self._dp else 0) | 2                      # bit 1: the ppo_mimic KL form
        st = _lib.current_stream()
        _lib.check(lib.pbhc_ppo_loss(mu.data_ptr(), sigma.data_ptr(), value.data_ptr(), b["actions"].data_ptr(), b["actions_log_prob"].data_ptr(),
                                     b["action_mean"].data_ptr(), b["action_sigma"].data_ptr(), b["advantages"].data_ptr(), b["returns"].data_ptr(),
                                     b["values"].data_ptr(), B, self.num_act, self.num_rew_fn, float(self.clip_param), float(self.value_loss_coef),
                                     float(self.entropy_coef), int(self.use_clipped_value_loss), float(self.desired_kl or 0.0), flags,
                                     self._grad_mu.data_ptr(), self._grad_value.data_ptr(), self._g_sigma.data_ptr(), self._loss_scalars.data_ptr(),
                                     loss["_acc"].data_ptr() if "_acc" in loss else None, self._lr.data_ptr(), self._loss_scratch.data_ptr(), st), "pbhc_ppo_loss")
        heads, grads = [mu, value], [self._grad_mu, self._grad_value]
        if coef != 0.0:
            heads.append(priv_reg * coef)
            grads.append(torch.ones((), device=self.device))
        torch.autograd.backward(heads, grads)
        if alg.std.requires_grad:                                               # d sigma / d std of clamp(std, min, max)
            so, sn = self._std_slice
            std = alg.std.detach()
            self._gflat[so:so + sn] = self._g_sigma * ((std >= alg.min_sigma) & (std <= alg.max_sigma))
        if self._dp:
            # ONE all-reduce per optimiser step: the main segment's gradients and, in the slot behind them, this rank's minibatch KL mean —
            # averaged by the collective itself; the learning-rate rule (ppo_mimic.py:617-630) then runs on the all-rank KL as one launch
            nb = self._n_main + adapt
            if adapt:
                self._gflat[self._n_main:nb].copy_(self._loss_scalars[3:4])
            pdist.allreduce_mean_(self._gflat[:nb])
            if adapt:
                _lib.check(lib.pbhc_kl_lr_rule(self._lr.data_ptr(), 2, self._gflat[self._n_main:].data_ptr(), float(self.desired_kl), st), "pbhc_kl_lr_rule")
        self._adam(0, 0, self._n_main, self._lr[0:1])
        if "_acc" in loss:
            pass                                           # (summed by the loss kernel's finishing block)
        else:
            loss["Value"] += self._loss_scalars[1]; loss["Surrogate"] += self._loss_scalars[0]; loss["Entropy"] += self._loss_scalars[2]
        loss["priv_reg_loss"] += priv_reg.detach()
        return loss

    def _update_dagger(self, b, loss):
        a = self.alg.actor
        with torch.no_grad():
            priv_latent = a.priv_encoding(b["priv_obs"])
        hist_loss = (priv_latent - a.history_encoding(b["prop_history"])).norm(p=2, dim=1).mean()
        self._zero_grads("hist")
        hist_loss.backward()
        if self._dp:
            pdist.allreduce_mean_(self._gflat[self._o_hist:])
        self._adam(1, self._o_hist, self._n_hist, self._lr_hist)
        loss["hist_latent_loss"] += hist_loss.detach()
        return loss

    # ---- evaluation / export surface --------------------------------------------------------
    @property
    def inference_model(self):
        return {"actor": self.alg.actor}

    def get_example_obs(self):
        obs = self.env.reset_all()
        return {k: v.clone() for k, v in obs.items()}

    @torch.no_grad()
    def evaluate_policy_steps(self, Nsteps):
        self._eval_mode()
        self.env.set_is_evaluating()
        obs = self.env.reset_all()
        for _ in range(Nsteps):
            obs, _, _, _ = self.env.step({"actions": self.alg.act_inference(obs, hist_encoding=True)})
        return obs

    def evaluate_policy(self):
        return self.evaluate_policy_steps(int(self.env.max_episode_length))

    def _post_epoch_logging(self, log, width=80, pad=40):
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs * self.world_size
        if log["it"] % self.logging_interval != 0:
            return
        for c, l in self._timer.resolve():                  # device time of every iteration since the last logging interval
            self.collection_time, self.learn_time = c, l
            self.tot_time += c + l
        log["collection_time"], log["learn_time"] = self.collection_time, self.learn_time
        it_time = self.collection_time + self.learn_time
        if self.rank != 0:
            return
        stats = self._ep_stats.tolist()
        self._ep_stats.zero_()
        fps = int(self.num_steps_per_env * self.env.num_envs * self.world_size / max(it_time, 1e-9))
        it, w = log["it"], self.writer
        for k, v in log["loss_dict"].items():
            w.add_scalar("Loss/" + k, float(v), it)
        w.add_scalar("Loss/learning_rate", float(self._lr[0]), it)
        w.add_scalar("Policy/mean_noise_std", float(self.alg.std.detach().mean()), it)
        w.add_scalar("Perf/total_fps", fps, it)
        w.add_scalar("Perf/collection_time", log["collection_time"], it)
        w.add_scalar("Perf/learning_time", log["learn_time"], it)
        if stats[2] > 0:
            w.add_scalar("Train/mean_reward", stats[0] / stats[2], it)
            w.add_scalar("Train/mean_episode_length", stats[1] / stats[2], it)
        for k, v in (self.env.read_log() if hasattr(self.env, "read_log") else {}).items():
            w.add_scalar("Env/" + k, float(v), it)
        ld = ", ".join(f"{k} {float(v):.4f}" for k, v in log["loss_dict"].items() if "Load_Balancing" not in k)
        print(f"[it {it}] fps {fps}  collect {log['collection_time']:.3f}s  learn {log['learn_time']:.3f}s  {ld}  lr {float(self._lr[0]):.2e}  "
              f"ep_rew {stats[0] / max(stats[2], 1):.3f}  ep_len {stats[1] / max(stats[2], 1):.1f}", flush=True)
