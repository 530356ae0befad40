"""ReplaySimStub — the thin simulator that replaces Isaac Gym on the hot path.

Implements the surface of the reference's BaseSimulator that the env touches (reference:
humanoidverse/simulator/base_simulator/base_simulator.py:6-171; tensor views as in
simulator/isaacgym/isaacgym.py:574-618, DR observables :234-239) but, instead of stepping PhysX,
replays recorded / synthetic `(root state, q, q-dot, contact forces)` tensors `[T, N, ...]` that are
resident in HBM; advancing one control step is a pointer bump.  Rigid-body pose/twist — which
Isaac Gym supplied — come from the HIP FK kernel (`pbhc_sim_fk`, or fused inside `pbhc_env_step`).
Select it with `simulator._target_: pbhc_amd.simulator.replay_stub.ReplaySimStub`.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .. import dist as pdist
from ..skeleton import Skeleton


class ReplaySimStub:
    def __init__(self, config, device):
        self.config = config
        self.env_config = config
        self.robot_config = config.robot
        self.sim_device = device
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PbhcError("ReplaySimStub runs on the GPU only (device must be cuda:N)")
        _lib.lib()                      # fail loudly now if the HIP extension is missing
        self.headless = True
        self.viewer = None
        sim = config.simulator.config.sim
        self.sim_dt = 1.0 / sim.fps
        self.decimation = sim.control_decimation
        self._substep = 0
        self._frame = 0
        self.replay = None
        self.replay_version = 0
        self.frame_cursor = torch.zeros(1, dtype=torch.int32, device=self.device)     # device-side cursor, advanced by the step kernel
        self._host_frame = 0                                                           # its host mirror (-1: unknown, the kernel reads the device cursor)

    # ---- bring-up ------------------------------------------------------------------------
    def set_headless(self, headless):
        self.headless = headless

    def setup(self):
        pass

    def setup_terrain(self, mesh_type):
        if mesh_type not in ("plane", None):
            raise NotImplementedError("ReplaySimStub: only the plane terrain")

    def load_assets(self):
        self.skeleton = Skeleton.from_motion_config(self.robot_config.motion)
        self._csk = self.skeleton.to_c()
        self.body_names = list(self.skeleton.body_names)
        self.dof_names = list(self.robot_config.dof_names)
        if "body_names" in self.robot_config and list(self.robot_config.body_names) != self.body_names:
            raise _lib.PbhcError("robot.body_names do not match the skeleton's DFS body order")
        self.num_dof = len(self.dof_names)
        self.num_bodies = len(self.body_names)
        self._body_list = list(self.body_names)
        return self.num_dof, self.num_bodies, self.dof_names, self.body_names

    def create_envs(self, num_envs, env_origins, base_init_state):
        self.num_envs = num_envs
        self.env_origins = env_origins
        self.base_init_state = base_init_state
        dr = self.env_config.domain_rand
        N, dev = num_envs, self.device
        gen = pdist.host_generator(dev)          # rank-keyed on ranks > 0 of a data-parallel run
        u = lambda *s: torch.rand(*s, device=dev, generator=gen)
        self._base_com_bias = torch.zeros(N, 3, device=dev)
        if dr.get("randomize_base_com", False):
            r = dr.base_com_range
            for i, ax in enumerate(("x", "y", "z")):
                self._base_com_bias[:, i] = u(N) * (r[ax][1] - r[ax][0]) + r[ax][0]
        nl = len(dr.get("randomize_link_body_names", []))
        self._link_mass_scale = torch.ones(N, nl, device=dev)
        if dr.get("randomize_link_mass", False):
            self._link_mass_scale = u(N, nl) * (dr.link_mass_range[1] - dr.link_mass_range[0]) + dr.link_mass_range[0]
        self._base_mass_scale = torch.ones(N, 1, device=dev)
        self.friction_coeffs = torch.ones(N, 1, 1, device=dev)
        if dr.get("randomize_friction", False):
            self.friction_coeffs = u(N, 1, 1) * (dr.friction_range[1] - dr.friction_range[0]) + dr.friction_range[0]
        self._process_dof_props()

    def _process_dof_props(self):
        rc, dev = self.robot_config, self.device
        lo = torch.tensor(list(rc.dof_pos_lower_limit_list), dtype=torch.float, device=dev)
        hi = torch.tensor(list(rc.dof_pos_upper_limit_list), dtype=torch.float, device=dev)
        self.hard_dof_pos_limits = torch.stack([lo, hi], -1)
        self.dof_vel_limits = torch.tensor(list(rc.dof_vel_limit_list), dtype=torch.float, device=dev)
        self.torque_limits = torch.tensor(list(rc.dof_effort_limit_list), dtype=torch.float, device=dev)
        m = (lo + hi) / 2
        r = hi - lo
        s = self.env_config.rewards.reward_limit.soft_dof_pos_limit
        self.dof_pos_limits = torch.stack([m - 0.5 * r * s, m + 0.5 * r * s], -1)
        s = self.env_config.termination_scales.termination_close_to_dof_pos_limit
        self.dof_pos_limits_termination = torch.stack([m - 0.5 * r * s, m + 0.5 * r * s], -1)

    def get_dof_limits_properties(self):
        return self.dof_pos_limits, self.dof_vel_limits, self.torque_limits

    def find_rigid_body_indice(self, body_name):
        return self.body_names.index(body_name) if body_name in self.body_names else -1      # -1 like Isaac Gym

    def prepare_sim(self):
        N, D, B, dev = self.num_envs, self.num_dof, self.num_bodies, self.device
        self.all_root_states = torch.zeros(N, 13, device=dev)
        self.all_root_states[:] = self.base_init_state
        self.all_root_states[:, :3] += self.env_origins
        self.robot_root_states = self.all_root_states
        self.base_quat = self.robot_root_states[..., 3:7]
        self.dof_state = torch.zeros(N * D, 2, device=dev)
        self.dof_pos = self.dof_state.view(N, D, 2)[..., 0]
        self.dof_vel = self.dof_state.view(N, D, 2)[..., 1]
        # rigid-body state and contact forces: LAZY under the fused step (see mark_step) — the storage, and what they are pending on
        self._cf_buf = torch.zeros(N, B, 3, device=dev)
        self._rb_buf = torch.zeros(N, B, 13, device=dev)
        self._rb_buf[..., 6] = 1.0
        self._pending = None
        self.refresh_sim_tensors()

    # ---- replay ----------------------------------------------------------------------------
    def set_replay(self, root, dof_pos, dof_vel, contact, start_frame=0):
        """root [T,N,13], dof_pos / dof_vel [T,N,D], contact [T,N,B,3] — device tensors, kept resident."""
        N, D, B = self.num_envs, self.num_dof, self.num_bodies
        T = root.shape[0]
        chk = _lib.require_gpu_tensor
        self.replay = dict(root=chk(root, "root", torch.float32, (T, N, 13)), dof_pos=chk(dof_pos, "dof_pos", torch.float32, (T, N, D)),
                           dof_vel=chk(dof_vel, "dof_vel", torch.float32, (T, N, D)), contact=chk(contact, "contact", torch.float32, (T, N, B, 3)))
        self.replay_len = T
        self._frame = start_frame
        self.frame_cursor.fill_(start_frame % T)
        self._host_frame = start_frame % T
        self.replay_version += 1

    def ensure_replay(self):
        if self.replay is None:
            # nothing to replay yet (e.g. the reset_all() inside MHPPO.__init__): hold the current state
            self.replay = dict(root=self.robot_root_states.clone()[None], dof_pos=self.dof_pos.clone()[None].contiguous(),
                               dof_vel=self.dof_vel.clone()[None].contiguous(), contact=self.contact_forces.clone()[None])
            self.replay_len = 1
            self.frame_cursor.zero_()
            self._host_frame = 0
            self.replay_version += 1

    def use_device_cursor(self):
        """from now on the fused step reads the frame index from the device-side cursor (which its reduction has been advancing all along):
        what a captured graph of several steps needs — a frame index passed by value would be frozen into the graph"""
        self._host_frame = -1

    def take_host_frame(self):
        """Frame index of the fused step about to be launched, by value (the step kernel advances the device cursor itself)."""
        k = self._host_frame
        if k >= 0:
            self._host_frame = (k + 1) % self.replay_len
        return k

    def next_frame_index(self):
        """Host-side stepping (reference-style `simulate` loop only): frame index, cursor advanced on both sides."""
        self.ensure_replay()
        k = int(self.frame_cursor.item())
        self.frame_cursor.copy_((self.frame_cursor + 1) % self.replay_len)
        self._host_frame = (k + 1) % self.replay_len
        return k

    # ---- rigid-body state / contact forces of the simulator surface (reference names, isaacgym.py:574-618) -----------------------------
    # Nothing on the training path reads them, so the fused step does not store them (9.4 MB per step at 4096 envs): the env tells the stub
    # which replay frame the step consumed (mark_step) and the tensors are re-derived on first access — pbhc_sim_fk of that frame, i.e. the
    # arithmetic the fused kernel ran, and a copy of the frame's contact forces.  Values are those of the frame BEFORE a reset wrote new
    # root / dof states, as the reference's tensors are until its next refresh.
    def mark_step(self, frame_index):
        self._pending = (self.replay, frame_index)

    def _materialise(self):
        p = self._pending
        if p is None:
            return
        self._pending = None
        rep, k = p
        if k < 0:                                    # device-side cursor (the step advanced it): the frame it consumed is the one before
            k = (int(self.frame_cursor.item()) - 1) % rep["root"].shape[0]
        _lib.check(_lib.lib().pbhc_sim_fk(C.byref(self._csk), rep["root"][k].data_ptr(), rep["dof_pos"][k].data_ptr(), rep["dof_vel"][k].data_ptr(),
                                          1, self.num_envs, self._rb_buf.data_ptr(), _lib.current_stream()), "pbhc_sim_fk")
        self._cf_buf.copy_(rep["contact"][k])

    @property
    def _rigid_body_state(self):
        self._materialise()
        return self._rb_buf

    @property
    def contact_forces(self):
        self._materialise()
        return self._cf_buf

    _rigid_body_pos = property(lambda self: self._rigid_body_state[..., 0:3])
    _rigid_body_rot = property(lambda self: self._rigid_body_state[..., 3:7])
    _rigid_body_vel = property(lambda self: self._rigid_body_state[..., 7:10])
    _rigid_body_ang_vel = property(lambda self: self._rigid_body_state[..., 10:13])

    def refresh_sim_tensors(self):
        """Rigid-body pose/twist of the current (root, q, q-dot) via the HIP FK kernel."""
        self._pending = None
        _lib.check(_lib.lib().pbhc_sim_fk(C.byref(self._csk), self.robot_root_states.data_ptr(), self.dof_pos.data_ptr(),
                                          self.dof_vel.data_ptr(), 2, self.num_envs, self._rb_buf.data_ptr(),
                                          _lib.current_stream()), "pbhc_sim_fk")

    def apply_torques_at_dof(self, torques):
        self.applied_torques = torques

    def simulate_at_each_physics_step(self):
        """Reference-style substepping: the replay frame lands on the last substep of a control step."""
        self._substep += 1
        if self._substep % self.decimation != 0:
            return
        k = self.next_frame_index()
        self.robot_root_states.copy_(self.replay["root"][k])
        self.dof_pos.copy_(self.replay["dof_pos"][k])
        self.dof_vel.copy_(self.replay["dof_vel"][k])
        self._pending = None
        self._cf_buf.copy_(self.replay["contact"][k])
        self.refresh_sim_tensors()

    def set_actor_root_state_tensor(self, set_env_ids, root_states):
        pass        # state tensors are written in place; the next replay frame overrides them

    def set_dof_state_tensor(self, set_env_ids, dof_states):
        pass

    def clear_lines(self):
        pass

    def draw_sphere(self, *a, **k):
        pass

    def render(self, sync_frame_time=True):
        pass
