"""MHPPO — drop-in for the reference's KungfuBot PPO (multi-head / vector-reward critic).

Same class surface as the reference (reference: humanoidverse/agents/mh_ppo/mh_ppo.py:26-775 on the
BaseAlgo API agents/base_algo/base_algo.py:15-47): `__init__(env, config, log_dir=None, device)`,
`setup()`, `load(path)`, `save(path, infos)`, `learn()`, `evaluate_policy()`, `inference_model`;
checkpoint dict and state_dict key names are the reference's, so trained policies still export and
deploy through the reference's tooling.  Select with
`algo._target_: pbhc_amd.agents.mh_ppo.MHPPO`.

MI355X-first differences (same maths, pinned by tests/golden/ppo_v1.npz):
  * no host synchronisation inside an iteration: the adaptive-KL learning rate, the loss meters and
    the episode statistics live on the device (`torch.where`, capturable Adam with tensor lr);
  * GAE / head-summed advantage / normalisation run in `pbhc_gae` (HIP) on the `[T,N,R]` slab;
  * the minibatch gather moves only the keys the update reads;
  * envs shard over ranks (one process per GPU): ONE flat-bucket RCCL all-reduce of the actor+critic
    gradients per optimiser step, plus two tiny all-reduces (advantage moments, KL mean) so that
    every rank takes the same normalisation and learning-rate branch as one big batch would.
"""
from __future__ import annotations

import os
import time
from collections import deque

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim

from .. import _lib
from .. import dist as pdist
from .modules import PPOActor, PPOCritic, RolloutStorage


class _NullWriter:
    def __getattr__(self, n):
        return lambda *a, **k: None


def _make_writer(log_dir):
    if log_dir is None:
        return _NullWriter()
    try:
        from torch.utils.tensorboard import SummaryWriter

        return SummaryWriter(log_dir=log_dir, flush_secs=10)
    except Exception:
        return _NullWriter()


class PhaseTimer:
    """`Perf/collection_time` / `Perf/learning_time` (mh_ppo.py:223-230,325-327) as DEVICE time.  The reference's host clock deltas mean
    "time the phase took" only because its rollout synchronises with the host every step; an iteration here is fully asynchronous, so the
    phases are bracketed by HIP events on the compute stream and read back when a logging interval ends (one synchronisation per
    interval, none per iteration)."""

    def __init__(self):
        self._cur, self._pending = None, []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._cur = [e]

    def split(self):
        """end of the phase that started at the previous mark (no-op outside learn())"""
        if self._cur is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._cur.append(e)
            if len(self._cur) == 3:
                self._pending.append(self._cur)
                self._cur = None

    def resolve(self):
        """-> [(collection_s, learn_s)] of the iterations finished since the last call (synchronises on the last one)."""
        out = []
        if self._pending:
            self._pending[-1][2].synchronize()
            out = [(a.elapsed_time(b) * 1e-3, b.elapsed_time(c) * 1e-3) for a, b, c in self._pending]
            self._pending = []
        return out


def _load_checkpoint(path, device):
    """Checkpoints — ours, the reference's, a third party's `model_*.pt` — hold tensors, numbers, strings, tuples, lists, dicts and None
    (state dicts, torch.optim state, `iter`, `infos`): they are read with the non-executing loader only.  A file that needs more than
    that is refused, never unpickled."""
    try:
        return torch.load(path, map_location=device, weights_only=True)
    except Exception as e:          # pickle.UnpicklingError / RuntimeError from the restricted unpickler
        raise _lib.PbhcError(f"checkpoint {path}: not loadable with torch.load(weights_only=True) ({type(e).__name__}: {str(e)[:300]}); "
                             "pbhc_amd does not unpickle arbitrary objects — re-save the file with plain tensors / numbers in `infos`") from e


def policy_forward_graphs(self, eager, key=0):
    """The rollout is launch-bound on the host (≈18 launches per control step): the policy forward of step t — 14 of them, reading the
    fixed rollout slab t and the in-place-updated flat weights — is captured once as a hipGraph per step index and replayed with one
    launch.  The first rollout runs eagerly (GEMM selection happens there); PBHC_FWD_GRAPHS=0 keeps everything eager.
    `key`: one set of graphs per forward variant (ppo_mimic: history / privileged latent)."""
    if os.environ.get("PBHC_FWD_GRAPHS", "1") == "0":
        return eager
    seen = self.__dict__.setdefault("_fwd_seen", set())
    if key not in seen:                        # first rollout of this variant: eager (online GEMM selection must not run inside a capture)
        seen.add(key)
        return eager
    cache = self.__dict__.setdefault("_fwd_graph_cache", {})
    if key not in cache:
        T = self.num_steps_per_env
        graphs, outs = [], []
        pool = None
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for t in range(T):
                g = torch.cuda.CUDAGraph()
                # thread_local: the RCCL watchdog thread polls its events while we capture; only this thread's calls are policed
                with torch.cuda.graph(g, pool=pool, stream=side, capture_error_mode="thread_local"):
                    out = eager(t)
                pool = g.pool() if pool is None else pool
                graphs.append(g); outs.append(out)
        torch.cuda.current_stream().wait_stream(side)
        cache[key] = (graphs, outs)
    graphs, outs = cache[key]

    def replay(t):
        graphs[t].replay()
        return outs[t]

    replay.outs = outs                         # static output tensors per step index (another graph may read them)
    return replay



class _FlatAdamView:
    """torch.optim.Adam-format state_dict()/load_state_dict() over one network's slice of the flat Adam buffers, so
    checkpoints keep the reference's `*_optimizer_state_dict` entries (mh_ppo.py:195-204)."""

    def __init__(self, algo, which):
        self.algo, self.which = algo, which

    def _range(self):
        a = self.algo
        n_actor_params = len(list(a.actor.parameters()))
        sl = a._slices[:n_actor_params] if self.which == 0 else a._slices[n_actor_params:]
        return sl

    def state_dict(self):
        a = self.algo
        state = {}
        for i, (o, k) in enumerate(self._range()):
            shape = a._params[i if self.which == 0 else i + len(list(a.actor.parameters()))].shape
            state[i] = {"step": a._adam_step[self.which].detach().clone().cpu(), "exp_avg": a._mflat[o:o + k].view(shape).clone(),
                        "exp_avg_sq": a._vflat[o:o + k].view(shape).clone()}
        group = {"lr": float(a._lr[self.which]), "betas": tuple(a.betas), "eps": a.adam_eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        a = self.algo
        for i, (o, k) in enumerate(self._range()):
            if i in sd["state"]:
                e = sd["state"][i]
                a._mflat[o:o + k].copy_(e["exp_avg"].reshape(-1).to(a.device))
                a._vflat[o:o + k].copy_(e["exp_avg_sq"].reshape(-1).to(a.device))
                a._adam_step[self.which] = float(e["step"])
        a._lr[self.which] = float(sd["param_groups"][0]["lr"])

    @property
    def param_groups(self):
        return [{"lr": float(self.algo._lr[self.which])}]


class MHPPO:
    def __init__(self, env, config, log_dir=None, device="cpu"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PbhcError("pbhc_amd.agents.mh_ppo.MHPPO runs on the GPU only")
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
        # device-side episode statistics: [sum of returns, sum of lengths, count] of finished episodes
        self._ep_stats = torch.zeros(3, dtype=torch.float64, device=self.device)
        self.world_size, self.rank = pdist.world(), pdist.rank()
        self._dp = pdist.active()                    # data-parallel exchanges on (more than one rank, or a forced one-rank rehearsal)
        self._dp_buckets = int(os.environ.get("PBHC_DP_GRAD_BUCKETS", "1"))
        # algo.config.sync_env_statistics: "rollout" (default; True means the same) | "step" (exact single-process equivalence) | False
        self._stat_mode = {True: "rollout", False: None, None: None}.get(config.get("sync_env_statistics", "rollout"), config.get("sync_env_statistics", "rollout"))
        if self._dp and self._stat_mode and hasattr(self.env, "enable_global_statistics"):
            self.env.enable_global_statistics(mode=self._stat_mode)     # sigma / episode-length curricula from the batch of all ranks' envs
        _ = self.env.reset_all()

    def _init_config(self):
        c = self.config
        self.num_envs = self.env.num_envs
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
        self.actor_learning_rate = c.actor_learning_rate
        self.critic_learning_rate = c.critic_learning_rate
        self.clip_param = c.clip_param
        self.num_learning_epochs = c.num_learning_epochs
        self.num_mini_batches = c.num_mini_batches
        self.gamma = c.gamma
        self.lam = c.lam
        self.value_loss_coef = c.value_loss_coef
        self.entropy_coef = c.entropy_coef
        self.max_grad_norm = c.max_grad_norm
        self.use_clipped_value_loss = c.use_clipped_value_loss
        self.cfg_l2c2 = c.l2c2 if "l2c2" in c else None
        self.num_rew_fn = self.env.num_rew_fn

    # ------------------------------------------------------------------------------------
    def setup(self):
        from .gemm_tuning import enable as _enable_gemm_tuning

        _enable_gemm_tuning()
        self._setup_models_and_optimizer()
        self._setup_storage()

    def _setup_models_and_optimizer(self):
        c = self.config
        if "phase_embed" in c and c.phase_embed.type != "Original":
            # the reference's branch (mh_ppo.py:127-142) constructs `PhaseAwareActorV2` / `PhaseAwareCriticV2`, which no file of the reference
            # defines or imports: it raises NameError there, so there is no behaviour to reproduce
            raise NotImplementedError("phase_embed.type != 'Original': the reference's PhaseAwareActorV2 / PhaseAwareCriticV2 do not exist (mh_ppo.py:127-142)")
        c.module_dict.critic["output_dim"][-1] = self.num_rew_fn
        self.actor = PPOActor(obs_dim_dict=self.algo_obs_dim_dict, module_config_dict=c.module_dict.actor, num_actions=self.num_act,
                              init_noise_std=c.init_noise_std).to(self.device)
        self.critic = PPOCritic(obs_dim_dict=self.algo_obs_dim_dict, module_config_dict=c.module_dict.critic).to(self.device)
        if self._dp:      # replicas start from rank 0's weights
            for p in list(self.actor.parameters()) + list(self.critic.parameters()):
                pdist.broadcast(p.data, src=0)
        self._flatten_parameters()

    def _flatten_parameters(self):
        """All actor+critic parameters (and their grads / Adam moments) live in ONE flat fp32 buffer each:
        one RCCL all-reduce over the gradient bucket, one clip+Adam launch per network."""
        dev = self.device
        pa, pc = list(self.actor.parameters()), list(self.critic.parameters())
        self._params = pa + pc
        self._n_actor = sum(p.numel() for p in pa)
        self._n_critic = sum(p.numel() for p in pc)
        n = self._n_actor + self._n_critic
        self._pflat = torch.zeros(n, device=dev)
        self._gflat = torch.zeros(n + 1, device=dev)         # + one slot behind the critic's segment: the minibatch KL rides in its all-reduce
        self._mflat = torch.zeros(n, device=dev)
        self._vflat = torch.zeros(n, device=dev)
        o = 0
        self._slices = []
        for p in self._params:
            k = p.numel()
            self._pflat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self._pflat[o:o + k].view_as(p)
            p.grad = self._gflat[o:o + k].view_as(p)
            self._slices.append((o, k))
            o += k
        std_idx = [i for i, (nme, _) in enumerate(self.actor.named_parameters()) if nme == "std"][0]
        self._std_slice = self._slices[std_idx]
        self._lr = torch.tensor([float(self.actor_learning_rate), float(self.critic_learning_rate)], device=dev)
        self._lr_a, self._lr_c = self._lr[0:1], self._lr[1:2]
        self._adam_step = torch.zeros(2, device=dev)          # [actor, critic] step counts (float, like torch's `step` tensors)
        self._adam_scratch = torch.zeros(2, 512, dtype=torch.float64, device=dev)
        self._grad_norms = torch.zeros(2, device=dev)
        self._loss_scalars = torch.zeros(4, device=dev)
        self.betas, self.adam_eps = (0.9, 0.999), 1e-8
        # kept for checkpoint (de)serialisation in torch.optim.Adam's format
        self.actor_optimizer = _FlatAdamView(self, 0)
        self.critic_optimizer = _FlatAdamView(self, 1)
        # every update zeroes `_gflat` before its backward: the MLP stacks may store their gradients into it directly
        from . import fused_mlp
        from .modules import BaseModule

        self._direct_stacks = []
        for m in list(self.actor.modules()) + list(self.critic.modules()):
            if isinstance(m, BaseModule):
                fused_mlp.grad_direct(m.module)
                self._direct_stacks.append(m.module)

    def _zero_grads(self):
        """zero the flat gradient buffer and tell the declared stacks (their next backward may store instead of accumulate)"""
        from . import fused_mlp

        # INVARIANT behind the shortcut: between the Adam pass of one optimiser step (pbhc_adam_clip2(zero_grad=1) leaves the buffer zeroed)
        # and this call nothing runs a backward through a module whose .grad views the flat buffer.  _update_ppo is the only writer on the
        # training path; load() and the eager update clear the flag; PBHC_CHECK_GRAD_CLEAN=1 verifies it (one reduction + a sync per step).
        if getattr(self, "_gflat_clean", False) and os.environ.get("PBHC_CHECK_GRAD_CLEAN", "0") == "1":
            assert float(self._gflat[: self._n_actor + self._n_critic].abs().sum()) == 0.0, "a gradient was written outside _update_ppo"
        if not getattr(self, "_gflat_clean", False):      # (clean: the last Adam pass left it zeroed and nothing has written it since)
            self._gflat.zero_()
        self._gflat_clean = False
        for q in self._direct_stacks:
            fused_mlp.grads_zeroed(q)

    def _setup_storage(self):
        st = self.storage = RolloutStorage(self.env.num_envs, self.num_steps_per_env, self.device)
        self._need_next = bool(self.cfg_l2c2 is not None and self.cfg_l2c2.enable)
        if self._need_next:                      # the L2C2 terms run each network twice per graph: plain autograd accumulation
            self.actor.actor_module._fused = False
            self.critic.critic_module._fused = False
        for k, d in self.algo_obs_dim_dict.items():
            st.register_key(k, shape=(d,), dtype=torch.float, pad_rows=True, tail_slab=True)
            if self._need_next:
                st.register_key("next_" + k, shape=(d,), dtype=torch.float)
        st.register_key("actions", shape=(self.num_act,), dtype=torch.float)
        st.register_key("rewards", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("dones", shape=(1,), dtype=torch.bool)
        st.register_key("values", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("returns", shape=(self.num_rew_fn,), dtype=torch.float)
        st.register_key("advantages", shape=(1,), dtype=torch.float)
        st.register_key("actions_log_prob", shape=(1,), dtype=torch.float)
        st.register_key("action_mean", shape=(self.num_act,), dtype=torch.float)
        st.register_key("action_sigma", shape=(self.num_act,), dtype=torch.float)
        T, N = self.num_steps_per_env, self.env.num_envs
        self._gae_stats = torch.zeros(2 * ((T * N + 255) // 256) + 4, dtype=torch.float64, device=self.device)
        self._last_obs = {k: st.with_tail(k)[T] for k in self.algo_obs_dim_dict}       # the observations after the last step: slab T of the same buffers
        self._sample_seed = pdist.rank_seed(int(torch.randint(0, 2**62, (1,)).item()))
        self._branch_stream = torch.cuda.Stream(device=self.device)
        self._update_streams = os.environ.get("PBHC_UPDATE_STREAMS", "0") == "1"      # measured slower (37.0 vs 34.5 ms per update): off
        if not hasattr(self.env, "globals") or not hasattr(self.env, "set_obs_outputs"):
            raise _lib.PbhcError("pbhc_amd MHPPO drives the fused pbhc_amd env (needs env.globals / env.set_obs_outputs)")
        mb = (T * N) // self.num_mini_batches
        self._mb = mb
        self._loss_scratch = torch.zeros(_lib.lib().pbhc_ppo_loss_scratch_floats(mb), device=self.device)
        self._grad_mu = torch.zeros(mb, self.num_act, device=self.device)
        self._grad_value = torch.zeros(mb, self.num_rew_fn, device=self.device)

    def _eval_mode(self):
        self.actor.eval(); self.critic.eval()

    def _train_mode(self):
        self.actor.train(); self.critic.train()

    # ---- checkpoints: the reference's dict (mh_ppo.py:176-204) -----------------------------
    def load(self, ckpt_path):
        if ckpt_path is None:
            return None
        d = _load_checkpoint(ckpt_path, self.device)
        self._gflat_clean = False            # (whatever touched the gradients meanwhile: the next update zeroes the flat buffer itself)
        self.actor.load_state_dict(d["actor_model_state_dict"])
        self.critic.load_state_dict(d["critic_model_state_dict"])
        if self.load_optimizer:
            self.actor_optimizer.load_state_dict(d["actor_optimizer_state_dict"])
            self.critic_optimizer.load_state_dict(d["critic_optimizer_state_dict"])
            self.set_learning_rate(float(d["actor_optimizer_state_dict"]["param_groups"][0]["lr"]),
                                   float(d["critic_optimizer_state_dict"]["param_groups"][0]["lr"]))
        self.current_learning_iteration = d["iter"]
        return d["infos"]

    def _opt_state_for_save(self, opt, lr):
        return opt.state_dict()

    def save(self, path, infos=None):
        torch.save({
            "actor_model_state_dict": self.actor.state_dict(),
            "critic_model_state_dict": self.critic.state_dict(),
            "actor_optimizer_state_dict": self._opt_state_for_save(self.actor_optimizer, self._lr_a),
            "critic_optimizer_state_dict": self._opt_state_for_save(self.critic_optimizer, self._lr_c),
            "iter": self.current_learning_iteration,
            "infos": infos,
        }, path)

    def set_learning_rate(self, actor_learning_rate, critic_learning_rate):
        self.actor_learning_rate, self.critic_learning_rate = actor_learning_rate, critic_learning_rate
        self._lr[0] = float(actor_learning_rate)
        self._lr[1] = float(critic_learning_rate)

    # ---- learn loop (mh_ppo.py:206-250) ----------------------------------------------------
    def learn(self, num_iterations=None):
        if self.init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf, high=int(self.env.max_episode_length))
        obs_dict = self.env.reset_all()
        self._train_mode()
        n = self.num_learning_iterations if num_iterations is None else num_iterations
        tot_iter = self.current_learning_iteration + n
        for it in range(self.current_learning_iteration, tot_iter):
            self._timer.start()
            obs_dict = self._rollout_step(obs_dict)           # ends with _timer.split(): collection | learning
            loss_dict = self._training_step()
            self._timer.split()
            self._post_epoch_logging(dict(it=it, loss_dict=loss_dict, num_learning_iterations=n))
            if self.log_dir is not None and it % self.save_interval == 0 and self.rank == 0:
                self.current_learning_iteration = it
                self.save(os.path.join(self.log_dir, f"model_{it}.pt"))
            self.ep_infos.clear()
        self.current_learning_iteration = tot_iter
        if self.log_dir is not None and self.rank == 0:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def _actor_act_step(self, obs_dict):
        return self.actor.act(obs_dict["actor_obs"])

    def _critic_eval_step(self, obs_dict):
        return self.critic.evaluate(obs_dict["critic_obs"])

    def _prefetch_permutation(self, n):
        """The update's minibatch permutation (data_utils.py:139, `torch.randperm(batch_size)`) does not depend on the rollout's data: its
        ten sort launches (~125 us) are queued on a stream of their own before the rollout starts and run beside it.  Same generator, same
        draw per iteration — nothing else draws from torch's device generator in between (policy sampling and env resets are Philox streams
        keyed by their own counters)."""
        if self.device.type != "cuda" or os.environ.get("PBHC_PERM_PREFETCH", "1") == "0":
            return
        if self.__dict__.get("_perm_stream") is None:
            self._perm_stream = torch.cuda.Stream(device=self.device)
            self._perm_buf = torch.empty(n, dtype=torch.int64, device=self.device)
        if self._perm_buf.numel() != n:
            self._perm_buf = torch.empty(n, dtype=torch.int64, device=self.device)
        ps = self._perm_stream
        ps.wait_stream(torch.cuda.current_stream())           # (the previous update's gather has read the buffer)
        with torch.cuda.stream(ps):
            torch.randperm(n, device=self.device, out=self._perm_buf)
        self._perm_event = ps.record_event()

    def _take_permutation(self, n):
        ev = self.__dict__.get("_perm_event")
        if ev is None or self._perm_buf.numel() != n:
            return None
        self._perm_event = None
        torch.cuda.current_stream().wait_event(ev)
        return self._perm_buf

    def _rollout_step(self, obs_dict):
        """mh_ppo.py:270-342.  Per control step: the policy stack + sampling / log-prob / buffer writes (one launch), the fused env step —
        which writes the next observations straight into the next rollout-buffer slab — and ONE done / episode-statistics kernel; the
        critic over all slabs and the time-out bootstrap after the loop.  No host synchronisation."""
        st, env, lib = self.storage, self.env, _lib.lib()
        T, N, A, R = self.num_steps_per_env, env.num_envs, self.num_act, self.num_rew_fn
        keys = list(obs_dict.keys())
        K = _lib.K
        counter = env.globals[K["PBHC_G_STEP_COUNTER"]:].data_ptr()
        self._prefetch_permutation(T * N)
        std = self.actor.std
        stream = _lib.current_stream()
        with torch.inference_mode():
            for k in keys:
                getattr(st, k)[0].copy_(obs_dict[k])
            # Per control step the dependent chain is env step -> policy forward (+ sampling in its last epilogue) -> env step.  Everything
            # else of a step runs on a branch stream NEXT to that chain: the env step's one-workgroup reduction (sigma EMA, curricula, step
            # counter) and the done / episode-statistics kernel.  The chain waits for the branch once per step, right before the next env step.
            # The critic's values are consumed only by the time-out bootstrap and by GAE, both after the rollout: evaluated ONCE over all T
            # slabs (98 304 rows: whole-chip GEMM tiles at ~120 TFLOP/s) it costs 1.5 ms, against 24 x 85 us for per-step forwards that
            # share the chip with the step -> actor chain (a control step's kernels add up to its duration: overlap buys ~10 %).
            # PBHC_CRITIC_BATCHED=0: the critic of slab t on the branch stream next to the actor, one hipGraph launch per step.
            from . import fused_mlp

            cur, br = torch.cuda.current_stream(), self._branch_stream
            split = os.environ.get("PBHC_ROLLOUT_SPLIT", "1") != "0" and hasattr(env, "set_finalize_stream")
            batched = split and os.environ.get("PBHC_CRITIC_BATCHED", "1") != "0"
            # the weights are constant over the rollout: the networks named here run as ONE launch per step from a packed copy (pbhc_mlp_fwd);
            # packed BEFORE the step graphs are captured below — the capture records whichever kernels the forward launches
            stack_nets = [n_ for n_ in os.environ.get("PBHC_STACK_NETS", "actor").split(",") if n_ and not (batched and n_ == "critic")]
            stacks = [m.module for n_, m in (("actor", self.actor.actor_module), ("critic", self.critic.critic_module)) if n_ in stack_nets and m._fused]
            stacks = [q for q in stacks if fused_mlp.pack_stack(q)]
            try:
                # ... and the sampling kernel in that launch's last epilogue, keyed by a snapshot of the step counter + the step index (the same
                # keys pbhc_policy_sample forms from the live counter, without waiting for the previous step's reduction)
                fuse_sample = bool(stacks) and stacks[0] is self.actor.actor_module.module and os.environ.get("PBHC_FUSED_SAMPLE", "1") != "0"
                if fuse_sample:
                    if self.__dict__.get("_ctr0") is None:
                        self._ctr0 = torch.zeros(1, dtype=torch.float64, device=self.device)
                    env.wait_finalize()
                    self._ctr0.copy_(env.globals[K["PBHC_G_STEP_COUNTER"]:K["PBHC_G_STEP_COUNTER"] + 1])
                    ctr0_p, a_seq = self._ctr0.data_ptr(), self.actor.actor_module.module

                    def actor_eager(t):
                        if not fused_mlp.forward_sample(a_seq, getattr(st, "actor_obs")[t], std, self._sample_seed, ctr0_p, t, st.actions[t], st.action_mean[t],
                                                        st.action_sigma[t], st.actions_log_prob[t]):
                            raise _lib.PbhcError("pbhc_mlp_fwd_sample does not apply to this policy (PBHC_FUSED_SAMPLE=0)")
                        return st.action_mean[t]
                else:
                    actor_eager = lambda t: self.actor.actor_module(getattr(st, "actor_obs")[t])
                actor_fwd = policy_forward_graphs(self, actor_eager, key="actor_s" if fuse_sample else "actor")
                critic_fwd = None if batched else policy_forward_graphs(self, lambda t: self.critic.critic_module(getattr(st, "critic_obs")[t]), key="critic")
                if split:
                    env.set_finalize_stream(br)
                post_done = self.__dict__.setdefault("_post_done", torch.cuda.Event())
                if batched and self.__dict__.get("_time_outs") is None:
                    self._time_outs = torch.zeros(T, N, 1, dtype=torch.bool, device=self.device)
                # per-step device addresses, formed once (the host's share of a control step is what bounds the loop once the critic is out of it)
                sc = self.__dict__.get("_step_ptrs")
                if sc is None or sc[0] is not st or sc[2] != batched:
                    P = lambda x: x.data_ptr()
                    sc = (st, [dict(sample=(P(st.actions[t]), P(st.action_mean[t]), P(st.action_sigma[t]), P(st.actions_log_prob[t])),
                                    post=(P(st.rewards[t]), P(st.dones[t])), values=P(st.values[t]),
                                    tout=P(self._time_outs[t]) if batched else None,
                                    act={"actions": st.actions[t]},
                                    obs_out={k: getattr(st, k)[t + 1] for k in keys} if t + 1 < T else self._last_obs) for t in range(T)], batched)
                    self._step_ptrs = sc
                steps = sc[1]
                std_p, sum_p, len_p, stat_p, gamma = std.data_ptr(), self.cur_reward_sum.data_ptr(), self.cur_episode_length.data_ptr(), self._ep_stats.data_ptr(), float(self.gamma)
                br_h = br.cuda_stream
                post_done = [post_done]                  # (a cell: the capture below swaps the event it used for a fresh one)

                def run_loop(cur, actor_call):
                    stream = cur.cuda_stream
                    br.wait_stream(cur)
                    for t in range(T):
                        sp = steps[t]
                        if not batched:
                            with torch.cuda.stream(br):
                                st.values[t].copy_(critic_fwd(t))
                        mu = actor_call(t)
                        if split and t > 0:
                            cur.wait_event(post_done[0])          # reduction + book-keeping kernel of step t-1 (13 us of work, issued ~60 us ago)
                            env.finalize_joined()
                        if not fuse_sample:
                            _lib.check(lib.pbhc_policy_sample(mu.data_ptr(), std_p, None, N, A, R, self._sample_seed, counter, *sp["sample"], None, stream), "pbhc_policy_sample")
                        env.set_obs_outputs(sp["obs_out"])
                        nxt, rewards, dones, infos = env.step(sp["act"])
                        if self._need_next:
                            for k in keys:
                                getattr(st, "next_" + k)[t].copy_(nxt[k])
                        if split:
                            # branch: [reduction of step t, queued by env.step] -> done / episode-statistics kernel of step t (per-step critic: values[t]
                            # were produced earlier on this stream and the bootstrap is added here) -> critic of slab t+1 (next iteration)
                            _lib.check(lib.pbhc_rollout_post2(rewards.data_ptr(), None if batched else sp["values"], dones.data_ptr(), infos["time_outs"].data_ptr(), N, R,
                                                              gamma, *sp["post"], sum_p, len_p, stat_p, sp["tout"], br_h), "pbhc_rollout_post2")
                            post_done[0].record(br)
                        else:
                            cur.wait_stream(br)
                            _lib.check(lib.pbhc_rollout_post(rewards.data_ptr(), sp["values"], dones.data_ptr(), infos["time_outs"].data_ptr(), N, R,
                                                             gamma, *sp["post"], sum_p, len_p, stat_p, stream), "pbhc_rollout_post")
                            br.wait_stream(cur)
                    cur.wait_stream(br)

                # ONE hipGraph for the whole loop (PBHC_ROLLOUT_GRAPH=0: the eager loop above all): T x (policy stack + sampling, fused env step,
                # its reduction and the done / episode-statistics kernel on the branch stream) with fork / join edges instead of stream events
                # and 4 dispatch gaps per step — the steps read the replay frame from the device-side cursor, their addresses (rollout slabs)
                # are fixed, and the env's host-side events (DR re-draw, motion resample) are checked for the whole window before
                # (`rollout_graph_safe`); a rollout that contains one, or that is being timed launch by launch, runs eagerly.
                graph_ok = (os.environ.get("PBHC_ROLLOUT_GRAPH", "1") != "0" and split and batched and fuse_sample and not self._need_next
                            and self.__dict__.get("_rollouts_done", 0) >= 1 and not self.__dict__.get("_rollout_graph_failed", False)
                            and hasattr(env, "rollout_graph_safe") and env.rollout_graph_safe(T))
                ran = False
                if graph_ok:
                    env.simulator.use_device_cursor()
                    # everything a captured env step froze: the rollout slabs, the env's io struct (its epoch moves with every pointer the env
                    # re-points), the simulator's replay window (a new one is picked up lazily inside the next env.step(), i.e. AFTER this key
                    # is read: the version itself belongs to the key) and which kernel — generic or specialised — the launch names
                    key = (id(st), env._io_epoch, env.simulator.replay_version, bool(getattr(env, "is_specialised", False)), N, T)
                    gc = self.__dict__.get("_rollout_graph")
                    if gc is None or gc[0] != key:
                        gc = self._capture_rollout(key, run_loop, actor_eager, env, post_done, T)
                    if gc is not None:
                        gc[1].replay()
                        env.after_graph_steps(T)
                        ran = True
                self._rollout_used_graph = ran
                if not ran:
                    run_loop(cur, actor_fwd)
                if split:
                    env.set_finalize_stream(None)
            finally:
                # (also when a step raises: a stack left marked valid would serve stale weights to every later no-grad forward)
                for q in stacks:
                    fused_mlp.release_stack(q)
            if batched:
                # mh_ppo.py:286-305 for all steps at once: values of every slab, then rewards += gamma * values * time_outs
                # ... and the bootstrap values of GAE from the same launch set: the observations after the last step are slab T of the buffer
                vals = self.critic.critic_module(st.with_tail("critic_obs").flatten(0, 1)).view(T + 1, N, R)
                st.values.copy_(vals[:T])
                st.rewards.addcmul_(st.values, self._time_outs.to(torch.float32), value=float(self.gamma))
            st.step = T
            self._rollouts_done = self.__dict__.get("_rollouts_done", 0) + 1
            if self._dp and self._stat_mode == "rollout":
                env.sync_globals()                     # sigma / curricula / log means: the mean over the ranks, once per rollout
            self._timer.split()
            self._compute_returns(self._last_obs, last_values=vals[T] if batched else None)
        return self._last_obs

    def _capture_rollout(self, key, run_loop, actor_eager, env, post_done, T):
        """record the rollout loop into one hipGraph (nothing executes during the capture: the caller replays it).  On any failure the agent
        stays on the eager loop for good."""
        g = torch.cuda.CUDAGraph()
        side = self.__dict__.setdefault("_graph_stream", torch.cuda.Stream(device=self.device))
        counter0 = env.common_step_counter
        try:
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                # thread_local: the RCCL watchdog thread polls its events while we capture; only this thread's calls are policed
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    run_loop(side, actor_eager)
            torch.cuda.current_stream().wait_stream(side)
        except Exception as e:                                   # noqa: BLE001 (whatever the capture objects to: report once, go on eagerly)
            print(f"[pbhc] rollout graph capture failed ({type(e).__name__}: {e}); the rollout stays eager")
            self._rollout_graph_failed = True
            self._rollout_graph = None
            g = None
        finally:
            # the env's and the loop's events were recorded INSIDE the capture: they are edges of the graph now, not events a later eager
            # step may wait on; the host-side counters the captured env.step() calls advanced are advanced again after every replay
            env.common_step_counter = counter0
            env._step_done, env._fin_done, env._fin_pending = torch.cuda.Event(), torch.cuda.Event(), False
            post_done[0] = torch.cuda.Event()
            self._post_done = post_done[0]
        if g is None:
            return None
        self._rollout_graph = (key, g)
        return self._rollout_graph

    def _compute_returns(self, last_obs_dict, last_values=None):
        """mh_ppo.py:348-395 in one HIP pass over the [T,N,R] slab."""
        st = self.storage
        if last_values is None:
            last_values = self.critic.evaluate(last_obs_dict["critic_obs"]).detach()
        last_values = last_values.contiguous()
        T, N, R = self.num_steps_per_env, self.env.num_envs, self.num_rew_fn
        adv = st.advantages
        _lib.check(_lib.lib().pbhc_gae(st.rewards.data_ptr(), st.values.data_ptr(), st.dones.data_ptr(), last_values.data_ptr(), T, N, R,
                                       float(self.gamma), float(self.lam), st.returns.data_ptr(), adv.data_ptr(), self._gae_stats.data_ptr(),
                                       _lib.current_stream()), "pbhc_gae")
        if self._dp:
            # same normalisation as one big batch: undo the local one, re-normalise with global moments
            nb = (T * N + 255) // 256
            mean_l, std_l = self._gae_stats[2 * nb].float(), self._gae_stats[2 * nb + 1].float()
            raw = adv * (std_l + 1e-8) + mean_l
            adv.copy_(pdist.global_normalize_(raw))
        return st.returns, adv

    # ---- update (mh_ppo.py:397-533) --------------------------------------------------------
    UPDATE_KEYS = ["actor_obs", "critic_obs", "actions", "values", "advantages", "returns", "actions_log_prob", "action_mean", "action_sigma"]

    def _training_step(self, indices=None):
        names = ["Value", "Surrogate", "Entropy", "L2C2_Value", "L2C2_Policy"]
        meters = torch.zeros(len(names) + 4, device=self.device)          # one fill: the five meters and, behind them, the loss kernel's running sums
        loss = {k: meters[i] for i, k in enumerate(names)}
        keys = list(self.UPDATE_KEYS)
        if self._need_next:
            keys += ["next_actor_obs", "next_critic_obs"]
        loss["_acc"] = meters[len(names):]                     # {surrogate, value, entropy, kl} summed by the loss kernel itself, one slot per scalar
        if indices is None:
            indices = self._take_permutation(self.storage.num_envs * self.storage.num_transitions_per_env)      # (None: drawn by the generator now)
        for batch in self.storage.mini_batch_generator(self.num_mini_batches, self.num_learning_epochs, keys=keys, indices=indices):
            self._update_ppo(batch, loss)
        acc = loss.pop("_acc")
        loss["Surrogate"] += acc[0]; loss["Value"] += acc[1]; loss["Entropy"] += acc[2]
        n = self.num_learning_epochs * self.num_mini_batches
        self.storage.clear()
        self.actor_learning_rate = self._lr_a        # tensors; read back lazily by the logger
        self.critic_learning_rate = self._lr_c
        means = meters[:len(names)] / n
        return {k: means[i] for i, k in enumerate(names)}

    def _allreduce_grads(self):
        """ONE RCCL all-reduce of the flat actor+critic gradient buffer (≈5 MB fp32), then average."""
        pdist.allreduce_mean_(self._gflat)

    def _update_ppo(self, b, loss):
        if self._need_next:
            return self._update_ppo_eager(b, loss)
        lib = _lib.lib()
        c = self
        # The two networks are independent until the loss kernel: the critic's forward runs on the branch stream next to the actor's, and
        # autograd replays each backward on the stream its forward ran on — the narrow layers of one network (128 / 23 / 21 columns: fewer
        # output tiles than CUs) share the chip with the wide layers of the other.  Measured on MI355X (4096 envs): 37.0 ms per update
        # against 34.5 ms on one stream — the wide GEMMs are tuned to own the chip and lose more than the narrow ones gain — so this
        # stays an experiment behind PBHC_UPDATE_STREAMS=1.
        two = self._update_streams
        cur, br = torch.cuda.current_stream(), self._branch_stream
        if two:
            br.wait_stream(cur)
            with torch.cuda.stream(br):
                value = self.critic.critic_module(b["critic_obs"])
            mu = self.actor.actor_module(b["actor_obs"])
            cur.wait_stream(br)
            value.record_stream(cur)                     # allocated on the branch stream, read by the loss kernel on this one
        else:
            mu = self.actor.actor_module(b["actor_obs"])
            value = self.critic.critic_module(b["critic_obs"])
        B = mu.shape[0]
        if B != self._mb:
            raise _lib.PbhcError("minibatch size changed")
        self._zero_grads()
        so, sn = self._std_slice
        adapt = int(self.desired_kl is not None and self.schedule == "adaptive")
        on_device_lr = adapt if not self._dp else 0
        st = _lib.current_stream()
        _lib.check(lib.pbhc_ppo_loss(mu.data_ptr(), self.actor.std.data_ptr(), value.data_ptr(), b["actions"].data_ptr(), b["actions_log_prob"].data_ptr(),
                                     b["action_mean"].data_ptr(), b["action_sigma"].data_ptr(), b["advantages"].data_ptr(), b["returns"].data_ptr(),
                                     b["values"].data_ptr(), B, self.num_act, self.num_rew_fn, float(self.clip_param), float(self.value_loss_coef),
                                     float(self.entropy_coef), int(self.use_clipped_value_loss), float(self.desired_kl or 0.0), on_device_lr,
                                     self._grad_mu.data_ptr(), self._grad_value.data_ptr(), self._gflat[so:so + sn].data_ptr(), self._loss_scalars.data_ptr(),
                                     loss["_acc"].data_ptr() if "_acc" in loss else None, self._lr.data_ptr(), self._loss_scratch.data_ptr(), st), "pbhc_ppo_loss")
        na, nc = self._n_actor, self._n_critic
        if self._dp:
            # ONE all-reduce per optimiser step (north_star: "a single RCCL all-reduce of policy gradients per PPO update"): actor + critic
            # segments and, in the slot behind them, the minibatch KL mean (the adaptive learning-rate rule, mh_ppo.py:455-466, needs the
            # mean over ALL ranks' samples) — averaged by the collective itself (ReduceOp.AVG), 5.2 MB, latency-bound on xGMI.
            # PBHC_DP_GRAD_BUCKETS=2: the round-2 form — the critic's 3.8 MB exchanged while the actor's backward runs, the actor's 1.45 MB
            # exposed — one more collective launch per step for ~25 us of hidden wire time; bench.py's dp1_rehearsal is the meter.
            self._gflat[na + nc:na + nc + 1].copy_(self._loss_scalars[3:4])
            if self._dp_buckets == 2:
                torch.autograd.backward([value], [self._grad_value])
                h_c = pdist.all_reduce(self._gflat[na:na + nc + 1], async_op=True)
                torch.autograd.backward([mu], [self._grad_mu])
                h_a = pdist.all_reduce(self._gflat[:na], async_op=True)
                h_a.wait(); h_c.wait()
                self._gflat.div_(self.world_size)
            else:
                self._backward_both(mu, value)
                pdist.allreduce_mean_(self._gflat[:na + nc + 1])
            if adapt:                                    # the rule on the all-rank KL mean: one launch (pdist.kl_lr_rule_ is its host-tensor form)
                _lib.check(lib.pbhc_kl_lr_rule(self._lr.data_ptr(), 2, self._gflat[na + nc:].data_ptr(), float(self.desired_kl), st), "pbhc_kl_lr_rule")
        else:
            if two:
                br.wait_stream(cur)                      # the loss kernel's gradients are ready for the critic's backward on its stream
                torch.autograd.backward([mu, value], [self._grad_mu, self._grad_value])
                cur.wait_stream(br)
            else:
                self._backward_both(mu, value)
        # both networks' clip_grad_norm_ + Adam in one launch pair (two segments of the flat buffers, each clipped by its own norm); the pass
        # leaves the gradient buffer zeroed — the next step's zero_grad()
        _lib.check(lib.pbhc_adam_clip2(self._pflat.data_ptr(), self._gflat.data_ptr(), self._mflat.data_ptr(), self._vflat.data_ptr(), na, nc,
                                       self._lr.data_ptr(), self._adam_step.data_ptr(), float(self.max_grad_norm), self.betas[0], self.betas[1],
                                       self.adam_eps, 0.0, 1, self._adam_scratch.data_ptr(), self._grad_norms.data_ptr(), st), "pbhc_adam_clip2")
        self._gflat_clean = True
        if "_acc" not in loss:                            # ("_acc": summed by the loss kernel's finishing block)
            loss["Value"] += self._loss_scalars[1]; loss["Surrogate"] += self._loss_scalars[0]; loss["Entropy"] += self._loss_scalars[2]
        return loss

    def _backward_both(self, mu, value):
        """both networks' backward on this stream, their finishing column-sum launches merged into one (fused_mlp.finish_deferred)"""
        from . import fused_mlp

        fused_mlp.begin_deferred_finish()
        try:
            torch.autograd.backward([mu, value], [self._grad_mu, self._grad_value])
        finally:
            fused_mlp.finish_deferred()

    def _update_ppo_eager(self, b, loss):
        """Eager PyTorch form of the update (used only for the optional L2C2 regulariser, mh_ppo.py:488-507)."""
        self.actor.update_distribution(b["actor_obs"])
        logp = self.actor.get_actions_log_prob(b["actions"])
        value = self.critic.evaluate(b["critic_obs"])
        mu, sigma, entropy = self.actor.action_mean, self.actor.action_std, self.actor.entropy
        if self.desired_kl is not None and self.schedule == "adaptive":
            with torch.no_grad():
                old_s, old_m = b["action_sigma"], b["action_mean"]
                kl = torch.sum(torch.log(sigma / old_s + 1.0e-5) + (old_s.square() + (old_m - mu).square()) / (2.0 * sigma.square()) - 0.5, axis=-1)
                kl_mean = kl.mean()
                if self._dp:
                    pdist.all_reduce(kl_mean)
                    kl_mean = kl_mean / self.world_size
                up = kl_mean > self.desired_kl * 2.0
                down = (kl_mean < self.desired_kl / 2.0) & (kl_mean > 0.0)
                self._lr.copy_(torch.where(up, torch.clamp(self._lr / 1.5, min=1e-5), torch.where(down, torch.clamp(self._lr * 1.5, max=1e-2), self._lr)))
        adv = b["advantages"].squeeze(-1)
        ratio = torch.exp(logp - b["actions_log_prob"].squeeze(-1))
        surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        if self.use_clipped_value_loss:
            vclip = b["values"] + (value - b["values"]).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - b["returns"]).pow(2), (vclip - b["returns"]).pow(2)).sum(dim=-1).mean()
        else:
            value_loss = (b["returns"] - value).pow(2).sum(dim=-1).mean()
        entropy_loss = entropy.mean()
        l2c2_v = torch.zeros((), device=self.device)
        l2c2_p = torch.zeros((), device=self.device)
        if self._need_next:
            u = torch.rand(*b["actor_obs"].shape[:-1], 1, device=self.device) * 2 - 1
            u_mu = self.actor.act_inference(b["actor_obs"] + u * (b["next_actor_obs"] - b["actor_obs"]))
            u_val = self.critic.evaluate(b["critic_obs"] + u * (b["next_critic_obs"] - b["critic_obs"]))
            l2c2_v = self.cfg_l2c2.lambda_value * (value - u_val).pow(2).mean()
            l2c2_p = self.cfg_l2c2.lambda_policy * (b["actions"] - u_mu).pow(2).mean()
        actor_loss = surrogate - self.entropy_coef * entropy_loss + l2c2_p
        critic_loss = self.value_loss_coef * value_loss + l2c2_v
        self._zero_grads()
        actor_loss.backward()
        critic_loss.backward()
        if self._dp:
            self._allreduce_grads()
        lib, st = _lib.lib(), _lib.current_stream()
        na, nc = self._n_actor, self._n_critic
        # both networks' clip_grad_norm_ + Adam in one launch pair (two segments of the flat buffers, each clipped by its own norm)
        _lib.check(lib.pbhc_adam_clip2(self._pflat.data_ptr(), self._gflat.data_ptr(), self._mflat.data_ptr(), self._vflat.data_ptr(), na, nc,
                                       self._lr.data_ptr(), self._adam_step.data_ptr(), float(self.max_grad_norm), self.betas[0], self.betas[1],
                                       self.adam_eps, 0.0, 0, self._adam_scratch.data_ptr(), self._grad_norms.data_ptr(), st), "pbhc_adam_clip2")
        with torch.no_grad():
            loss["Value"] += value_loss.detach(); loss["Surrogate"] += surrogate.detach(); loss["Entropy"] += entropy_loss.detach()
            loss["L2C2_Value"] += l2c2_v.detach(); loss["L2C2_Policy"] += l2c2_p.detach()
        return loss

    # ---- evaluation / export surface --------------------------------------------------------
    @property
    def inference_model(self):
        return {"actor": self.actor, "critic": self.critic}

    def get_example_obs(self):
        obs = self.env.reset_all()
        return {k: v.clone() for k, v in obs.items()}

    @torch.no_grad()
    def evaluate_policy_steps(self, Nsteps):
        self._eval_mode()
        self.env.set_is_evaluating()
        obs = self.env.reset_all()
        for _ in range(Nsteps):
            obs, _, _, _ = self.env.step({"actions": self.actor.act_inference(obs["actor_obs"])})
        return obs

    def evaluate_policy(self):
        return self.evaluate_policy_steps(int(self.env.max_episode_length))

    # ---- logging (mh_ppo.py:547-700, reduced to the Perf/* + Loss/* + Train/* scalars) ------
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
        stats = self._ep_stats.tolist()           # the only read-back, once per logging interval
        self._ep_stats.zero_()
        fps = int(self.num_steps_per_env * self.env.num_envs * self.world_size / max(it_time, 1e-9))
        it = log["it"]
        w = self.writer
        for k, v in log["loss_dict"].items():
            w.add_scalar("Loss/" + k, float(v), it)
        w.add_scalar("Loss/actor_learning_rate", float(self._lr_a), it)
        w.add_scalar("Loss/critic_learning_rate", float(self._lr_c), it)
        w.add_scalar("Policy/mean_noise_std", float(self.actor.std.detach().mean()), it)
        w.add_scalar("Perf/total_fps", fps, it)
        w.add_scalar("Perf/collection_time", log["collection_time"], it)
        w.add_scalar("Perf/learning_time", log["learn_time"], it)
        if stats[2] > 0:
            w.add_scalar("Train/mean_reward", stats[0] / stats[2], it)
            w.add_scalar("Train/mean_episode_length", stats[1] / stats[2], it)
        envlog = self.env.read_log() if hasattr(self.env, "read_log") else {}
        for k, v in envlog.items():
            w.add_scalar("Env/" + k, float(v), it)
        print(f"[it {it}] fps {fps}  collect {log['collection_time']:.3f}s  learn {log['learn_time']:.3f}s  "
              f"value {float(log['loss_dict']['Value']):.4f}  surr {float(log['loss_dict']['Surrogate']):.4f}  "
              f"lr {float(self._lr_a):.2e}  ep_rew {stats[0] / max(stats[2], 1):.3f}  ep_len {stats[1] / max(stats[2], 1):.1f}", flush=True)
