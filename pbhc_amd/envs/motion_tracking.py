"""LeggedRobotMotionTracking — drop-in for the reference's KungfuBot (v1) env, running the whole
`step()` as one fused HIP launch.

Same constructor, methods, attributes and return values as the reference class
(reference: humanoidverse/envs/motion_tracking/motion_tracking.py:97-135 on top of
legged_robot_base.py:29-37,239-265 and base_task.py:19-93), selected the same way:
`env._target_: pbhc_amd.envs.motion_tracking.LeggedRobotMotionTracking`.
`step(actor_state)` returns `(obs_dict, rew_buf, reset_buf, extras)` with env-owned tensors that
are overwritten by the next step, as in the reference.  There is no eager / CPU path: every step
goes through `pbhc_env_step`; construction fails if libpbhc_hip.so is missing.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
import sys
from collections.abc import Mapping

import numpy as np
import torch

from .. import _lib
from .. import dist as pdist
from ..motion_lib import MotionLib
from . import env_config

K = _lib.K


class EpisodeExtras(Mapping):
    """`extras["episode"]` of a step (legged_robot_base.py:510-515): `rew_<term>` = the finished episodes' reward sums / max_episode_length_s
    and `end_epis_length`, one entry per env reset in that step.  Their shape depends on how many envs reset, which the reference learns with
    a host synchronisation every step (`nonzero`); here the kernel leaves the per-env values on the device and this mapping gathers them on
    first access.  It reads env-owned buffers of ITS step: `env.step()` materialises a mapping somebody still holds before it launches the
    next step, so a consumer may keep it (the reference's agents append it to `ep_infos` and read it at logging time)."""

    def __init__(self, env):
        self._env, self._data = env, None

    def materialise(self):
        if self._data is None:
            e = self._env
            ids = e.reset_buf.nonzero(as_tuple=False).flatten()
            d = {"rew_" + k: e._episode_rew_out[ids, i] for i, k in enumerate(e.layout.sum_names)}
            d["end_epis_length"] = e.last_episode_length_buf[ids]
            self._data, self._env = d, None
        return self._data

    def __getitem__(self, k):
        return self.materialise()[k]

    def __iter__(self):
        return iter(self.materialise())

    def __len__(self):
        return len(self.materialise())


class StepExtras(dict):
    """The env's `extras` mapping (one object, refilled every step, like the reference's `self.extras`).  `ref_body_pos_extend` /
    `ref_body_rot_extend` — the reference motion's extended body poses of the last step (motion_tracking.py:645-650) — are read by
    evaluation callbacks only, so the fused step does not store them: the kernel leaves the motion time of its lookup (4 B per env instead
    of 756) and the two tensors are rebuilt by the same lerp / slerp (pbhc_motion_state) when one of them is read — valid until the next
    `step()`, exactly as long as the reference's env-owned tensors are."""
    LAZY = {"ref_body_pos_extend": 0, "ref_body_rot_extend": 1}

    def __init__(self, env):
        super().__init__()
        self._env = env

    def _fill(self):
        pos, rot = self._env._reference_bodies()
        dict.__setitem__(self, "ref_body_pos_extend", pos)
        dict.__setitem__(self, "ref_body_rot_extend", rot)

    def invalidate(self):
        for k in self.LAZY:
            dict.__setitem__(self, k, None)

    def __getitem__(self, k):
        if k in self.LAZY and dict.get(self, k) is None:
            self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __iter__(self):
        # (an overridden __iter__ takes `dict(extras)` / `{**extras}` / `other.update(extras)` off CPython's storage-level fast path: they
        # then read through keys() + __getitem__, i.e. they see the lazy members materialised, not None)
        return iter(list(dict.keys(self)))

    def __reduce__(self):
        self._fill()
        return (dict, (dict(dict.items(self)),))

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def copy(self):
        self._fill()
        return dict(self)


class ReinitSchedule:
    """domain_rand.reinit_epis_rand > 0: the episodic domain randomisation of EVERY env is re-drawn at exponentially distributed intervals
    (legged_robot_base.py:142,390-395): the counter starts at 0 — the first re-draw fires in the very first step — and every firing draws
    the next interval with ONE np.random.rand.  Host logic only (tests/test_reinit_schedule.py replays the reference's lines)."""

    def __init__(self, mean_interval):
        self.mean = float(mean_interval)
        self.counter = 0.0 if self.mean > 0 else float("inf")

    def due(self, common_step_counter):
        """`common_step_counter`: the value AFTER this step's increment (the reference tests it in `_update_tasks_callback`, behind
        `_update_counters_each_step`)."""
        if common_step_counter >= self.counter:
            self.counter = common_step_counter + float(-np.log(np.random.rand(1))[0] * self.mean)
            return True
        return False


def get_class(path):
    mod, name = path.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


class LeggedRobotMotionTracking:
    TRACKING_MODE = 0          # selects the kernel instantiation (see PbhcEnvConfig.tracking_mode)

    def __init__(self, config, device):
        """config = cfg.env.config (with .robot/.obs/.rewards/.domain_rand/.terrain/.simulator aliased
        to the top-level nodes, as Hydra composes it)."""
        self.init_done = False
        self.config = config
        self._lib = _lib.lib()
        self.is_evaluating = False
        # ---- BaseTask.__init__ (base_task.py:20-64)
        self.simulator = get_class(config.simulator._target_)(config=config, device=device)
        self.headless = config.headless
        self.simulator.set_headless(self.headless)
        self.simulator.setup()
        self.device = torch.device(self.simulator.sim_device)
        self.sim_dt = self.simulator.sim_dt
        self.up_axis_idx = 2
        self.dt = config.simulator.config.sim.control_decimation * self.sim_dt
        self.max_episode_length_s = config.max_episode_length_s
        self.max_episode_length = np.ceil(self.max_episode_length_s / self.dt)
        self.num_envs = N = config.num_envs
        self.dim_actions = config.robot.actions_dim
        self.simulator.setup_terrain(config.terrain.mesh_type)
        self.num_dof, self.num_bodies, self.dof_names, self.body_names = self.simulator.load_assets()
        assert self.num_dof == self.dim_actions, "Number of DOFs must be equal to number of actions"
        self.num_dofs = self.num_dof
        dev = self.device
        rc = config.robot
        base = list(rc.init_state.pos) + list(rc.init_state.rot) + list(rc.init_state.lin_vel) + list(rc.init_state.ang_vel)
        self.base_init_state = torch.tensor(base, dtype=torch.float, device=dev)
        self._get_env_origins()
        self.simulator.create_envs(N, self.env_origins, self.base_init_state)
        self.dof_pos_limits, self.dof_vel_limits, self.torque_limits = self.simulator.get_dof_limits_properties()
        self.simulator.prepare_sim()
        self.viewer = None
        # ---- motion library (motion_tracking.py:171-199)
        self.skeleton = self.simulator.skeleton
        rc.motion.step_dt = self.dt
        self.max_len = int(getattr(rc.motion, "motion_max_len", -1)) if self.TRACKING_MODE == 1 else -1     # general_tracking.py:60
        self._motion_lib = MotionLib.from_config(rc.motion, self.skeleton, N, dev, max_len=self.max_len)
        # host-side draws (slot -> clip sampling, start phases and episodic DR of reset_all): the torch global generator on rank 0 — as the
        # reference — and a rank-keyed generator on the other ranks of a data-parallel run, so that equal seeds do not replicate envs
        self._gen = pdist.host_generator(dev)
        self._motion_lib.generator = self._gen
        if self._gen is not None:
            import random

            self._motion_lib.pyrand = random.Random(self._gen.initial_seed())
        self._load_motions_initial()
        for e in rc.motion.get("extend_config", []):
            self.simulator._body_list.append(e["joint_name"])           # motion_tracking.py:226
        self.num_extend_bodies = len(rc.motion.get("extend_config", []))
        # ---- static config -> device
        # Philox key of this env shard: the torch-seeded draw, with the rank mixed in (same config.seed on every rank must not replicate streams)
        self._seed = pdist.rank_seed(int(torch.randint(0, 2**31 - 1, (1,)).item()), bits=31)
        self._c, self.layout = env_config.build(_TopView(config), self.skeleton, self._motion_lib, N, dev, self.simulator._link_mass_scale.shape[1],
                                                seed=self._seed, mode=self.TRACKING_MODE)
        L = self.layout
        self.reward_names = L.reward_names
        self.reward_scales = L.reward_scales
        self.feet_indices = torch.tensor(L.feet, dtype=torch.long, device=dev)
        self.penalised_contact_indices = torch.tensor(L.penalised, dtype=torch.long, device=dev)
        self.motion_tracking_id, self.lower_body_id, self.upper_body_id = L.track, L.lower, L.upper
        self.globals = torch.tensor(L.globals0, dtype=torch.float64, device=dev)
        # reference yaw of env 0 at t = dt (motion_tracking.py:186-187); host-side, once
        ref0 = self._motion_lib.get_motion_state(torch.zeros(1, dtype=torch.long, device=dev), torch.full((1,), self.dt, device=dev), offset=self.env_origins[:1])
        q = ref0["root_rot"][0].tolist()
        self.ref_init_yaw = float(np.arctan2(2.0 * (q[3] * q[2] + q[0] * q[1]), q[3] * q[3] + q[0] * q[0] - q[1] * q[1] - q[2] * q[2]))
        self._c.ref_init_yaw = self.ref_init_yaw
        self._env = C.c_void_p()
        _lib.check(self._lib.pbhc_env_create(C.byref(self._c), C.byref(self._motion_lib.table), self.globals.data_ptr(), C.byref(self._env)), "pbhc_env_create")
        # the step kernel specialised to this config (a ~3 s hipcc build on the first env of a config, cached in pbhc_amd/_spec):
        # PBHC_SPECIALISE=off keeps the generic kernel, =cached never compiles
        self.specialise(os.environ.get("PBHC_SPECIALISE", "jit"))
        self._totals, self._stat_pending, self._stat_group, self._num_envs_total = None, None, None, float(N)     # enable_global_statistics()
        self._finalize_stream, self._step_done, self._fin_done, self._fin_pending = None, None, None, False        # set_finalize_stream()
        self._init_buffers()
        self._init_obs_buffers()
        self._build_io()
        self.log_dict = {}
        self.extras = StepExtras(self)
        self.common_step_counter = 0
        self._resample_motion_times(torch.arange(N, device=dev))
        if config.get("resample_motion_when_training", False):
            self.resample_time_interval = np.ceil(config.resample_time_interval_s / self.dt)
        # domain_rand.reinit_epis_rand > 0: the episodic DR of EVERY env is re-drawn at exponentially distributed intervals
        # (legged_robot_base.py:149-151,390-395)
        self.reinit_epis_rand = float(config.domain_rand.get("reinit_epis_rand", -1))
        self._reinit = ReinitSchedule(self.reinit_epis_rand)
        self.init_done = True

    def specialise(self, mode="jit", verbose=False):
        """Run this env's steps on a build of k_env_step specialised to its config (pbhc_amd/specialise.py): "jit" compiles on a cache
        miss (~3 s of hipcc, once per config, source and compiler version), "cached" only uses an existing object, "off" returns to the generic
        kernel.  Same source, same results; returns True when a specialised kernel is attached."""
        from .. import specialise as _spec

        self._specialise_mode = mode
        self._io_epoch = getattr(self, "_io_epoch", 0) + 1         # a captured graph of steps names the kernel it was recorded with
        return _spec.attach(self._env, mode, verbose=verbose)

    @property
    def is_specialised(self):
        return bool(self._lib.pbhc_env_is_specialised(self._env))

    def _load_motions_initial(self):
        self._motion_lib.load_motions(random_sample=not self.is_evaluating)            # motion_tracking.py:176-180

    # ------------------------------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "_env", None):
                self._lib.pbhc_env_destroy(self._env)
                self._env = None
        except Exception:
            pass

    def _get_env_origins(self):
        # base_task.py:102-138, plane terrain branch
        N, dev = self.num_envs, self.device
        self.custom_origins = False
        self.env_origins = torch.zeros(N, 3, device=dev)
        num_cols = np.floor(np.sqrt(N))
        num_rows = np.ceil(N / num_cols)
        xx, yy = torch.meshgrid(torch.arange(num_rows), torch.arange(num_cols), indexing="ij")
        spacing = self.config.env_spacing
        self.env_origins[:, 0] = (spacing * xx.flatten()[:N]).to(dev)
        self.env_origins[:, 1] = (spacing * yy.flatten()[:N]).to(dev)

    def _init_buffers(self):
        N, D, dev, L = self.num_envs, self.num_dof, self.device, self.layout
        f = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.actions, self.last_actions, self.actions_after_delay = f(N, D), f(N, D), f(N, D)
        self.action_queue = f(N, self._c.queue_len, D)
        dr = self.config.domain_rand
        if dr.randomize_ctrl_delay:
            self.action_delay_idx = torch.randint(dr.ctrl_delay_step_range[0], dr.ctrl_delay_step_range[1] + 1, (N,), device=dev, generator=self._gen)
        else:
            self.action_delay_idx = torch.zeros(N, dtype=torch.long, device=dev)
        self.last_dof_pos, self.last_dof_vel, self.torques = f(N, D), f(N, D), f(N, D)
        self.feet_air_time, self.contacts, self.contacts_filt = f(N, 2), f(N, 2), f(N, 2)
        self.last_contacts, self.last_contacts_filt = f(N, 2), f(N, 2)
        self._kp_scale, self._kd_scale = torch.ones(N, D, device=dev), torch.ones(N, D, device=dev)
        self._rfi_lim_scale, self._rao_scale = torch.ones(N, D, device=dev), torch.ones(N, D, device=dev)
        self.motion_start_times, self.motion_len, self.end_time_ratio_buf = f(N), f(N), f(N)
        self._episode_sums = f(N, len(L.sum_names))
        self.episode_sums = {name: self._episode_sums[:, i] for i, name in enumerate(L.sum_names)}
        self._episode_rew_out = f(N, len(L.sum_names))
        self._hist = _padded_rows(N, L.hist_dim, dev)            # rows start on 128-B lines
        self._episode_length_buf = torch.zeros(N, dtype=torch.long, device=dev)
        self.last_episode_length_buf = torch.zeros(N, dtype=torch.long, device=dev)
        self.reset_buf = torch.ones(N, dtype=torch.long, device=dev)
        self.time_out_buf = torch.zeros(N, dtype=torch.bool, device=dev)
        self.motion_ids = torch.arange(N, device=dev)
        self.rew_buf = f(N, L.num_rew_fn) if self.config.use_vec_reward else f(N)
        self._ref_time = f(N)                                # motion time of the last step's reference lookup (lazy reference bodies)
        self._ref_bodies = None
        self.default_dof_pos = torch.tensor([self._c.default_dof_pos[i] for i in range(D)], device=dev).repeat(N, 1)
        self.raw_default_dof_pos = self.default_dof_pos.clone()                    # legged_robot_base.py:81-93
        self.p_gains = torch.tensor([self._c.p_gains[i] for i in range(D)], device=dev)
        self.d_gains = torch.tensor([self._c.d_gains[i] for i in range(D)], device=dev)

    def _init_obs_buffers(self):
        L = self.layout
        self._own_obs = {g: _padded_rows(self.num_envs, L.group_dims[g], self.device) for g in L.group_names[:-1]}
        self.obs_buf_dict = dict(self._own_obs)

    def rebuild_observations(self):
        """Re-derive the observation maps after `config.obs.obs_dict` changed (the reference re-reads the dict every step, so
        ppo_mimic's distillation adds the teacher's observation groups after the env exists, ppo_mimic.py:131-134).  State is kept;
        the set of history keys must not change."""
        old = self.layout
        self._flush_statistics()
        c, L = env_config.build(_TopView(self.config), self.skeleton, self._motion_lib, self.num_envs, self.device, self.simulator._link_mass_scale.shape[1],
                                seed=self._seed, mode=self.TRACKING_MODE)
        if (L.hist_keys, L.hist_len, L.hist_dim) != (old.hist_keys, old.hist_len, old.hist_dim) or L.sum_names != old.sum_names:
            raise _lib.PbhcError("rebuild_observations: history keys / reward terms changed")
        c.ref_init_yaw = self.ref_init_yaw
        env = C.c_void_p()
        _lib.check(self._lib.pbhc_env_create(C.byref(c), C.byref(self._motion_lib.table), self.globals.data_ptr(), C.byref(env)), "pbhc_env_create")
        self._lib.pbhc_env_destroy(self._env)
        self._env, self._c, self.layout = env, c, L
        self._init_obs_buffers()
        self._build_io()
        if getattr(self, "_specialise_mode", None) not in (None, "off"):
            self.specialise(self._specialise_mode)                   # the new env object has a new config: its own specialised build

    @property
    def history(self):
        """per-key views [N, len, dim] of the packed history state (HistoryHandler.history)."""
        L = self.layout
        return {k: self._hist[:, L.hist_off[k]:L.hist_off[k] + L.hist_len[k] * L.obs_dims[k]].view(self.num_envs, L.hist_len[k], L.obs_dims[k]) for k in L.hist_keys}

    @property
    def episode_length_buf(self):
        return self._episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, v):           # MHPPO.learn assigns it (mh_ppo.py:208)
        self._episode_length_buf.copy_(v.to(self._episode_length_buf.dtype))

    @property
    def num_rew_fn(self):
        return self.layout.num_rew_fn

    def _build_io(self):
        s = self.simulator
        io = _lib.PbhcStepIO()
        p = lambda t: t.data_ptr()
        io.root_states, io.dof_state = p(s.robot_root_states), p(s.dof_state)
        # the simulator surface's rigid-body state and contact forces: re-derived by the stub on first access (ReplaySimStub.mark_step)
        io.rigid_body_state, io.contact_forces = None, None
        io.actions, io.last_actions, io.actions_after_delay, io.action_queue = p(self.actions), p(self.last_actions), p(self.actions_after_delay), p(self.action_queue)
        io.last_dof_pos, io.last_dof_vel, io.torques = p(self.last_dof_pos), p(self.last_dof_vel), p(self.torques)
        io.feet_air_time, io.contacts, io.contacts_filt = p(self.feet_air_time), p(self.contacts), p(self.contacts_filt)
        io.last_contacts, io.last_contacts_filt = p(self.last_contacts), p(self.last_contacts_filt)
        io.kp_scale, io.kd_scale, io.rfi_lim_scale, io.rao_scale = p(self._kp_scale), p(self._kd_scale), p(self._rfi_lim_scale), p(self._rao_scale)
        io.default_dof_pos = p(self.default_dof_pos) if self._c.randomize_default_dof_pos else None      # per-env defaults only when they are randomised
        io.motion_start_times, io.motion_len, io.end_time_ratio_buf = p(self.motion_start_times), p(self.motion_len), p(self.end_time_ratio_buf)
        io.episode_sums, io.hist = p(self._episode_sums), p(self._hist)
        io.episode_length_buf, io.last_episode_length_buf = p(self._episode_length_buf), p(self.last_episode_length_buf)
        io.reset_buf, io.action_delay_idx = p(self.reset_buf), p(self.action_delay_idx)
        self._slot_clip = self._motion_lib.slot_table.contiguous()
        io.motion_ids = p(self._slot_clip)
        io.time_out_buf = p(self.time_out_buf)
        io.env_origins = p(self.env_origins)
        self._friction_flat = s.friction_coeffs.reshape(self.num_envs, -1).contiguous()
        io.dr_base_com, io.dr_link_mass, io.dr_friction = p(s._base_com_bias), p(s._link_mass_scale.contiguous()), p(self._friction_flat)
        io.dr_base_mass = p(s._base_mass_scale)
        L = self.layout
        for i, g in enumerate(L.group_names[:-1]):
            io.obs[i] = p(self.obs_buf_dict[g])
            io.obs_pitch[i] = self.obs_buf_dict[g].stride(0)
        io.obs[len(L.group_names) - 1] = p(self._hist)
        io.obs_pitch[len(L.group_names) - 1] = io.hist_pitch = self._hist.stride(0)
        io.rew_buf = p(self.rew_buf)
        io.ref_body_pos_extend, io.ref_body_rot_extend = None, None          # lazy: rebuilt from ref_time_out on first access (StepExtras)
        io.ref_time_out = p(self._ref_time)
        io.episode_rew_out = p(self._episode_rew_out)
        io.totals_out = self._totals.data_ptr() if getattr(self, "_totals", None) is not None else None
        self._io = io
        self._io_epoch = getattr(self, "_io_epoch", 0) + 1        # captured graphs of the step hold these addresses: a new epoch invalidates them
        self._overrides = {}
        self._replay_version = -1

    def set_profiling(self, on=True):
        """per-launch HIP event pairs on k_env_step (bench.py's roofline meter; read with pbhc_env_profile_read).  While it is on the step
        launches through the hipExt call, which a stream capture cannot record: `rollout_graph_safe` says no."""
        _lib.check(self._lib.pbhc_env_profile(self._env, 1 if on else 0), "pbhc_env_profile")
        self._profiling = bool(on)

    def rollout_graph_safe(self, num_steps):
        """May the next `num_steps` control steps run as ONE captured graph?  Only if nothing in them needs the host: no per-step exchange of
        the batch statistics (data-parallel "step" mode), no re-draw / motion-resample event due inside the window, no per-launch event
        pairs, a replay window in place (the steps then read the frame index from the device-side cursor)."""
        if getattr(self, "_profiling", False) or self._totals is not None or self.simulator.replay is None or self.reinit_epis_rand > 0:
            return False
        if self.config.get("resample_motion_when_training", False):
            nxt = (self.common_step_counter // self.resample_time_interval + 1) * self.resample_time_interval
            if nxt - self.common_step_counter <= num_steps:
                return False
        return True

    def after_graph_steps(self, num_steps):
        """host-side book-keeping of `num_steps` control steps that ran inside a graph replay"""
        self.common_step_counter += num_steps
        self._ref_bodies = None
        self.extras.invalidate()
        self.simulator.mark_step(-1)
        self.extras["episode"] = EpisodeExtras(self)

    def set_eager_outputs(self, on=True):
        """Have the fused step STORE the optional outputs (rigid-body state, contact forces, reference bodies) instead of leaving them to be
        re-derived on access: for a consumer that reads them every step (an evaluation callback), and for the test that holds the two forms
        equal.  `env._eager` holds the four tensors."""
        N, B, Bx, dev = self.num_envs, self.num_bodies, self.skeleton.num_bodies_ext, self.device
        if on:
            self._eager = dict(rigid_body_state=torch.zeros(N, B, 13, device=dev), contact_forces=torch.zeros(N, B, 3, device=dev),
                               ref_body_pos_extend=torch.zeros(N, Bx, 3, device=dev), ref_body_rot_extend=torch.zeros(N, Bx, 4, device=dev))
        else:
            self._eager = None
        for k in ("rigid_body_state", "contact_forces", "ref_body_pos_extend", "ref_body_rot_extend"):
            setattr(self._io, k, self._eager[k].data_ptr() if on else None)
        self._io_epoch += 1

    def set_obs_outputs(self, tensors):
        """Point the observation outputs of the next step(s) at caller-owned `[N, dim]` tensors (e.g. the rollout-buffer slab of
        the next step) instead of the env-owned ones; `None` restores the env-owned buffers."""
        L = self.layout
        for i, g in enumerate(L.group_names[:-1]):
            t = self._own_obs[g] if tensors is None else _lib.require_gpu_rows(tensors[g], g, torch.float32, (self.num_envs, L.group_dims[g]))
            self.obs_buf_dict[g] = t
            self._io.obs[i] = t.data_ptr()
            self._io.obs_pitch[i] = t.stride(0)

    # ---- test / replay hooks: inject the random draws instead of the in-kernel Philox ---------
    def set_injected_draws(self, u_rfi=None, start_time=None, kp=None, kd=None, rfi_lim=None, rao=None, delay=None, dof_pos_bias=None, gate_u=None):
        """Keeps the tensors alive and points the kernel at them (None -> in-kernel RNG)."""
        self._overrides = dict(u_rfi=u_rfi, ovr_start_time=start_time, ovr_kp=kp, ovr_kd=kd, ovr_rfi_lim=rfi_lim, ovr_rao=rao, ovr_delay=delay,
                               ovr_dof_pos_bias=dof_pos_bias, ovr_gate_u=gate_u)
        self._io_epoch += 1
        for k, v in self._overrides.items():
            setattr(self._io, k, None if v is None else v.data_ptr())

    # ---- data-parallel runs: batch statistics over ALL ranks' envs ---------------------------
    def enable_global_statistics(self, group=None, mode="rollout"):
        """One process per GPU, envs sharded over ranks: adaptive sigma, average episode length, the curricula keyed on it and the logged
        means are batch statistics of the reference's single process (motion_tracking.py:1030-1048, legged_robot_base.py:875-900).
          mode "rollout" (default): every step updates them from THIS rank's shard (4096 envs: already a tight estimate), and
                 `sync_globals()` — called by the agents once per rollout — replaces every rank's copy by the mean over the ranks: ONE
                 1 KB all-reduce per PPO iteration, nothing on the step -> policy -> step chain;
          mode "step": each step writes its shard's batch sums (PBHC_NUM_TOTALS doubles), ONE 512-byte all-reduce sums them over the ranks
                 while the next policy forward runs, and `pbhc_env_finalize` applies them right before the next step launches: every rank
                 then holds exactly the sigma / curriculum state ONE process with all the envs would (tests/test_gpu_dist.py) — at 24
                 latency-critical collectives per iteration."""
        if not pdist.active(group):
            return False
        assert mode in ("rollout", "step"), mode
        self._flush_statistics()
        self._stat_group, self._stat_mode = group, mode
        if mode == "step":
            n = torch.tensor([float(self.num_envs)], dtype=torch.float64, device=self.device)
            pdist.all_reduce(n, group=group)
            self._num_envs_total = float(n)
            self._totals = torch.zeros(K["PBHC_NUM_TOTALS"], dtype=torch.float64, device=self.device)
            self._io.totals_out = self._totals.data_ptr()
        return True

    def sync_globals(self):
        """mode "rollout": the batch-statistics state (sigma, EMA, curricula, log means: the `globals` vector) becomes the mean over the ranks"""
        if getattr(self, "_stat_mode", None) != "rollout":
            return
        self.wait_finalize()
        pdist.allreduce_mean_(self.globals, group=self._stat_group)
        # the step counter (the kernels' Philox counter, read as (uint32_t)) is the same integer on every rank; a mean formed as
        # sum(x / n) may come back one ulp short for world sizes that are not powers of two — truncation would then replay a counter
        c = K["PBHC_G_STEP_COUNTER"]
        self.globals[c:c + 1].round_()

    def set_finalize_stream(self, stream):
        """Run every step's one-workgroup reduction (`pbhc_env_step_finish`: sigma EMA, curricula, log means, step counter) on `stream`
        instead of the stepping stream (None: back to one stream).  The env orders it after its fused launch and joins it before the next
        one; a caller that reads the globals on the stepping stream in between (pbhc_policy_sample reads the step counter) calls
        `wait_finalize()` first."""
        if self._finalize_stream is not None:
            self.wait_finalize()
        self._finalize_stream = stream
        if stream is not None and self._step_done is None:
            self._step_done, self._fin_done = torch.cuda.Event(), torch.cuda.Event()

    def wait_finalize(self):
        """the current stream waits for the last step's reduction (no-op on one stream)"""
        if self._fin_pending:
            torch.cuda.current_stream().wait_event(self._fin_done)
            self._fin_pending = False

    def finalize_joined(self):
        """The caller has made the stepping stream wait for something it queued on the finalize stream AFTER the last step's reduction
        (the rollout's book-keeping kernel): that wait covers the reduction, a second cross-stream wait (~10 us of dispatch latency each,
        signalled or not) is not needed."""
        self._fin_pending = False

    def _flush_statistics(self):
        h = self._stat_pending
        if h is not None:
            self._stat_pending = None
            h.wait()
            _lib.check(self._lib.pbhc_env_finalize(self._env, self._totals.data_ptr(), self._num_envs_total, _lib.current_stream()), "pbhc_env_finalize")

    # ------------------------------------------------------------------------------------
    def set_is_evaluating(self):
        self.is_evaluating = True

    def _resample_motion_times(self, env_ids):
        # motion_tracking.py:369-378
        if len(env_ids) == 0:
            return
        self.motion_len[env_ids] = self._motion_lib.get_motion_length(self.motion_ids[env_ids])
        if self.is_evaluating and not self.config.enforce_randomize_motion_start_eval:
            self.motion_start_times[env_ids] = 0.0
        else:
            self.motion_start_times[env_ids] = self._motion_lib.sample_time(self.motion_ids[env_ids])

    def _reference_bodies(self):
        """(ref_body_pos_extend [N,Bx,3], ref_body_rot_extend [N,Bx,4]) of the LAST step: the lookup the fused kernel did at `_ref_time`
        (before a reset), repeated with the stand-alone lookup kernel — cached until the next step"""
        if self._ref_bodies is None:
            ref = self._motion_lib.get_motion_state(self.motion_ids, self._ref_time, offset=self.env_origins)
            self._ref_bodies = (ref["rg_pos_t"], ref["rg_rot_t"])
        return self._ref_bodies

    @property
    def ref_body_pos_extend(self):
        return self._reference_bodies()[0]

    @property
    def ref_body_rot_extend(self):
        return self._reference_bodies()[1]

    def resample_motion(self, keep_reset_buf=False):
        """motion_tracking.py:385-389 / general_tracking.py:291-297"""
        if self.common_step_counter > 0:
            self._reference_bodies()                 # the last step's lazy reference bodies belong to the OLD slot -> clip table: freeze them
        self._motion_lib.load_motions(random_sample=True, max_len=self.max_len)
        self.curr_motion_ids = self._motion_lib.slot_clip
        self._reset_all_state(keep_reset_buf=keep_reset_buf)

    def reset_all(self):
        """base_task.py:83-93: reset every env, then one step with zero actions.  Start-up path,
        done with torch ops on the device (the per-step reset of terminated envs is in the kernel)."""
        self._reset_all_state()
        obs_dict, _, _, _ = self.step({"actions": torch.zeros(self.num_envs, self.dim_actions, device=self.device)})
        return obs_dict

    def _reset_all_state(self, keep_reset_buf=False):
        """reset_envs_idx(arange(N)) (legged_robot_base.py:491-517).  keep_reset_buf: the periodic resample inside step() — the reference's
        resample_motion() resets every env WITHOUT touching reset_buf, the dones of that step stay those of its own _check_termination."""
        self.wait_finalize()
        N, dev = self.num_envs, self.device
        ids = torch.arange(N, device=dev)
        g = self.globals
        # _reset_buffers_callback (legged_robot_base.py:670-686)
        for t in (self.actions, self.last_actions, self.actions_after_delay, self.last_dof_pos, self.last_dof_vel, self.feet_air_time,
                  self.contacts, self.contacts_filt, self.last_contacts, self.last_contacts_filt, self._hist):
            t.zero_()
        self._flush_statistics()
        cur = torch.mean(self.last_episode_length_buf, dtype=torch.float)
        n_all = N
        if self._totals is not None:                       # every rank resets all its envs: the mean over all ranks' envs
            m = torch.stack([self.last_episode_length_buf.sum().double(), torch.tensor(float(N), dtype=torch.float64, device=dev)])
            pdist.all_reduce(m, group=self._stat_group)
            cur, n_all = (m[0] / m[1]).float(), float(m[1])
        frac = n_all / self._c.num_compute_average_epl
        avg = g[K["PBHC_G_AVG_EP_LEN"]].float() * (1 - frac) + cur * frac
        g[K["PBHC_G_AVG_EP_LEN"]] = avg.double()
        self._episode_length_buf.zero_()
        if not keep_reset_buf:
            self.reset_buf.fill_(1)
        self._episodic_domain_randomization_all()
        # curricula keyed on average_episode_length (legged_robot_base.py:882-900, motion_tracking.py:309-317)
        c = self._c
        avg_f = float(avg)
        if c.penalty_curriculum:
            p = float(g[K["PBHC_G_PENALTY_SCALE"]])
            p *= (1 - c.penalty_degree) if avg_f < c.penalty_down else ((1 + c.penalty_degree) if avg_f > c.penalty_up else 1.0)
            g[K["PBHC_G_PENALTY_SCALE"]] = float(np.clip(p, c.penalty_min, c.penalty_max))
        if c.noise_curriculum:
            v = float(g[K["PBHC_G_NOISE_CURRICULUM"]])
            v *= (1 - c.noise_degree) if avg_f < c.noise_down else ((1 + c.noise_degree) if avg_f > c.noise_up else 1.0)
            g[K["PBHC_G_NOISE_CURRICULUM"]] = float(np.clip(v, c.noise_min, c.noise_max))
        end_time = self.last_episode_length_buf * self.dt + self.motion_start_times
        self.end_time_ratio_buf.copy_(end_time / self.motion_len.clamp(min=1e-9))
        self._resample_motion_times(ids)
        if c.terminate_when_motion_far and c.motion_far_curriculum:
            t = float(g[K["PBHC_G_MOTION_FAR_THR"]])
            t *= (1 + c.motion_far_degree) if avg_f < c.motion_far_down else ((1 - c.motion_far_degree) if avg_f > c.motion_far_up else 1.0)
            g[K["PBHC_G_MOTION_FAR_THR"]] = float(np.clip(t, c.motion_far_min, c.motion_far_max))
        # _reset_dofs / _reset_root_states from the reference frame at (0+1)*dt + start
        ref = self._motion_lib.get_motion_state(self.motion_ids, (self._episode_length_buf + 1) * self.dt + self.motion_start_times, offset=self.env_origins)
        s = self.simulator
        ref_dof = ref if self.TRACKING_MODE == 0 else self._motion_lib.get_motion_state(       # general_tracking.py:463-476: t = ep_len*dt + start
            self.motion_ids, self._episode_length_buf * self.dt + self.motion_start_times, offset=self.env_origins)
        s.dof_pos.copy_(ref_dof["dof_pos"]); s.dof_vel.copy_(ref_dof["dof_vel"])
        s.robot_root_states[:, 0:3] = ref["root_pos"]; s.robot_root_states[:, 3:7] = ref["root_rot"]
        s.robot_root_states[:, 7:10] = ref["root_vel"]; s.robot_root_states[:, 10:13] = ref["root_ang_vel"]
        self.extras["episode"] = {"rew_" + k: (v / self.max_episode_length_s).clone() for k, v in self.episode_sums.items()}
        self.extras["episode"]["end_epis_length"] = self.last_episode_length_buf.clone()
        self._episode_sums.zero_()
        self.extras["time_outs"] = self.time_out_buf

    def _episodic_domain_randomization_all(self):
        """_episodic_domain_randomization(arange(N)) (legged_robot_base.py:599-635): kp / kd / rfi-limit / rao scales, control delay."""
        N, dev, D = self.num_envs, self.device, self.num_dof
        dr = self.config.domain_rand
        u = lambda lo, hi: (hi - lo) * torch.rand(N, D, device=dev, generator=self._gen) + lo
        if dr.randomize_pd_gain:
            self._kp_scale.copy_(u(dr.kp_range[0], dr.kp_range[1]))
            self._kd_scale.copy_(u(dr.kd_range[0], dr.kd_range[1]))
        if dr.randomize_rfi_lim:
            self._rfi_lim_scale.copy_(u(dr.rfi_lim_range[0], dr.rfi_lim_range[1]))
        if dr.use_rao:
            self._rao_scale.copy_(u(-dr.rao_lim, dr.rao_lim))
        if dr.randomize_ctrl_delay:
            self.action_queue.zero_()
            self.action_delay_idx.copy_(torch.randint(dr.ctrl_delay_step_range[0], dr.ctrl_delay_step_range[1] + 1, (N,), device=dev, generator=self._gen))
        if dr.get("randomize_default_dof_pos", False):                            # legged_robot_base.py:632-635
            self.default_dof_pos.copy_(u(dr.dof_pos_range[0], dr.dof_pos_range[1]) + self.raw_default_dof_pos)

    def step(self, actor_state):
        """legged_robot_base.py:239-265 — one fused launch."""
        actions = actor_state["actions"]
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(self.device, torch.float32).contiguous()
        if tuple(actions.shape) != (self.num_envs, self.num_dof):
            raise _lib.PbhcError(f"actions must be [{self.num_envs},{self.num_dof}], got {tuple(actions.shape)}")
        self._actions_in = actions
        s = self.simulator
        io = self._io
        io.actions_in = actions.data_ptr()
        if self._replay_version != s.replay_version or s.replay is None:
            s.ensure_replay()
            r = s.replay
            io.frame_root, io.frame_dof_pos = r["root"].data_ptr(), r["dof_pos"].data_ptr()
            io.frame_dof_vel, io.frame_contact = r["dof_vel"].data_ptr(), r["contact"].data_ptr()
            io.frame_cursor, io.num_frames = s.frame_cursor.data_ptr(), s.replay_len
            self._replay_version = s.replay_version
            self._io_epoch += 1
        io.frame_index = s.take_host_frame()
        # _update_tasks_callback's re-draw of every env's episodic DR (legged_robot_base.py:390-395): decided on the host, done by the kernel
        # in this very step (after its torques, before its observations — where the reference does it)
        io.redraw_all = int(self._reinit.due(self.common_step_counter + 1))
        self._ref_bodies = None                      # the lazy members of last step's extras end here
        self.extras.invalidate()
        s.mark_step(io.frame_index)                  # the stub's rigid-body state / contact forces of this step: re-derived on first access
        # the previous step's lazy extras["episode"]: gather it now if somebody kept it (its buffers are about to be overwritten)
        prev = self.extras.pop("episode", None)
        if isinstance(prev, EpisodeExtras) and prev._data is None and sys.getrefcount(prev) > 2:
            prev.materialise()
        del prev
        self._flush_statistics()
        fs = self._finalize_stream
        if fs is None:
            _lib.check(self._lib.pbhc_env_step(self._env, C.byref(io), _lib.current_stream()), "pbhc_env_step")
        else:
            # the fused launch here, its one-workgroup reduction on the side stream (set_finalize_stream): this stream goes straight on to
            # the policy forward of the new observations; the reduction is joined again before the next launch (wait_finalize)
            cur = torch.cuda.current_stream()
            self.wait_finalize()
            _lib.check(self._lib.pbhc_env_step_launch(self._env, C.byref(io), cur.cuda_stream), "pbhc_env_step_launch")
            self._step_done.record(cur)
            fs.wait_event(self._step_done)
            _lib.check(self._lib.pbhc_env_step_finish(self._env, C.byref(io), fs.cuda_stream), "pbhc_env_step_finish")
            self._fin_done.record(fs)
            self._fin_pending = True
        if self._totals is not None:
            if fs is None:
                self._stat_pending = pdist.all_reduce(self._totals, group=self._stat_group, async_op=True)
            else:
                with torch.cuda.stream(fs):      # the shard's totals are written by the reduction on the side stream: the exchange follows it there
                    self._stat_pending = pdist.all_reduce(self._totals, group=self._stat_group, async_op=True)
        self.common_step_counter += 1
        # _update_tasks_callback (motion_tracking.py:320-325, general_tracking.py:216-222): periodic slot -> clip resampling + reset of every
        # env.  The reference does it inside the step, before termination and reward of that step; here it follows the fused launch, i.e.
        # it takes effect one control step later — once every resample_time_interval (50 000 - 100 000 steps in the shipped configs).
        # The dones returned for this step stay the kernel's own (the reference's resample_motion does not touch reset_buf).
        if self.config.get("resample_motion_when_training", False) and self.common_step_counter % self.resample_time_interval == 0:
            self.resample_motion(keep_reset_buf=True)
        else:
            self.extras["episode"] = EpisodeExtras(self)
        self.extras["time_outs"] = self.time_out_buf
        self.extras["to_log"] = self.log_dict
        self.extras["episode_rew"] = self._episode_rew_out
        return self.obs_buf_dict, self.rew_buf, self.reset_buf, self.extras

    # ---- logging: device-side means, read back on demand (no per-step sync) ----------------
    def read_log(self):
        self.wait_finalize()
        self._flush_statistics()
        g = self.globals.cpu().numpy()
        L0 = K["PBHC_G_LOG"]
        out = {
            "upper_body_diff_norm": g[L0 + K["PBHC_L_UPPER_BODY_DIFF_NORM"]], "lower_body_diff_norm": g[L0 + K["PBHC_L_LOWER_BODY_DIFF_NORM"]],
            "vr_3point_diff_norm": g[L0 + K["PBHC_L_VR_3POINT_DIFF_NORM"]], "joint_pos_diff_norm": g[L0 + K["PBHC_L_JOINT_POS_DIFF_NORM"]],
            "action_clip_frac": g[L0 + K["PBHC_L_ACTION_CLIP_FRAC"]], "terminate_by_gravity": g[L0 + K["PBHC_L_TERM_GRAVITY"]],
            "terminate_by_motion_far": g[L0 + K["PBHC_L_TERM_MOTION_FAR"]], "terminate_by_time_out": g[L0 + K["PBHC_L_TERM_TIME_OUT"]],
            "terminate_by_motion_end": g[L0 + K["PBHC_L_TERM_MOTION_END"]], "end_time_ratio": g[L0 + K["PBHC_L_END_TIME_RATIO"]],
            "end_time_ratio_std": g[L0 + K["PBHC_L_END_TIME_RATIO_STD"]], "penalty_scale": g[K["PBHC_G_PENALTY_SCALE"]], **({"current_noise_curriculum_value": g[K["PBHC_G_NOISE_CURRICULUM"]]} if self._c.noise_curriculum else {}),
            "average_episode_length": g[K["PBHC_G_AVG_EP_LEN"]], "terminate_when_motion_far_threshold": g[K["PBHC_G_MOTION_FAR_THR"]],
            "reward_mean": g[L0 + K["PBHC_L_REW_MEAN"]],
        }
        for flag, slot, name in ((self._c.soft_pos_curriculum, "PBHC_G_SOFT_POS_VAL", "soft_dof_pos"), (self._c.soft_vel_curriculum, "PBHC_G_SOFT_VEL_VAL", "soft_dof_vel"),
                                 (self._c.soft_tau_curriculum, "PBHC_G_SOFT_TAU_VAL", "soft_torque")):
            if flag:                                      # legged_robot_base.py:753-758
                out[name + "_curriculum_value"] = g[K[slot]]
        if self._c.terminate_by_contact:
            out["terminate_by_contact"] = g[L0 + K["PBHC_L_TERM_CONTACT"]]
        if self._c.terminate_by_low_height:
            out["terminate_by_low_height"] = g[L0 + K["PBHC_L_TERM_LOW_HEIGHT"]]
        for flag, key, name in ((self._c.terminate_close_pos, "PBHC_L_TERM_DOF_POS_LIMIT", "dof_pos_limit"), (self._c.terminate_close_vel, "PBHC_L_TERM_DOF_VEL_LIMIT", "dof_vel_limit"),
                                (self._c.terminate_close_tau, "PBHC_L_TERM_TORQUE_LIMIT", "torque_limit")):
            if flag:
                out["terminate_by_" + name] = g[L0 + K[key]]
        for i, k in enumerate(env_config.SIGMA_KEYS):
            out["adp_sigma_" + k] = g[K["PBHC_G_SIGMA"] + i]
            out["error_ema_" + k] = g[K["PBHC_G_EMA"] + i]
        self.log_dict.update({k: torch.tensor(float(v)) for k, v in out.items()})
        return out


def _padded_rows(n, dim, device, dtype=torch.float32):
    """[n, dim] view of an [n, ceil32(dim)] buffer: every row starts on a 128-byte line."""
    return torch.zeros(n, _lib.padded_width(dim), dtype=dtype, device=device)[:, :dim]


class _TopView:
    """env.config holds aliases of the top-level nodes; present them under the top-level names the
    config builder uses (cfg.env.config, cfg.robot, cfg.obs, ...)."""

    def __init__(self, env_cfg):
        self.env = _NS(config=env_cfg)
        self.robot, self.obs, self.rewards = env_cfg.robot, env_cfg.obs, env_cfg.rewards
        self.domain_rand, self.terrain, self.simulator = env_cfg.domain_rand, env_cfg.terrain, env_cfg.simulator


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)
