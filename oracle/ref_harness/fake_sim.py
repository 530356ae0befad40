"""Replay simulator that satisfies the reference's BaseSimulator surface (reference:
humanoidverse/simulator/base_simulator/base_simulator.py:6-171; tensor views as in
simulator/isaacgym/isaacgym.py:574-618) so the unmodified reference env can run on CPU.

State per control step is taken from replay tensors (root13, dof_pos, dof_vel, contact forces);
rigid-body pose/twist come from oracle.fk.sim_fk.  Like Isaac Gym, body tensors change only in
`simulate`, not when the env writes reset states in place.
"""
import numpy as np
import torch

from oracle.skeleton import parse_mjcf
from oracle.fk import sim_fk


class ReplayFakeSim:
    def __init__(self, config, device):
        self.config = config
        self.env_config = config
        self.robot_config = config.robot
        self.sim_device = device
        self.device = device
        self.headless = True
        self.viewer = None
        self.sim_dt = 1.0 / config.simulator.config.sim.fps
        self.decimation = config.simulator.config.sim.control_decimation
        self._substep = 0
        self._frame = 0
        self.replay = None

    # --- bring-up ---------------------------------------------------------------
    def set_headless(self, h):
        self.headless = h

    def setup(self):
        pass

    def setup_terrain(self, mesh_type):
        pass

    def load_assets(self):
        m = self.robot_config.motion
        self.skel = parse_mjcf(str(m.asset.assetRoot) + "/" + m.asset.assetFileName, [dict(e) for e in m.extend_config])
        self.body_names = list(self.skel["body_names"])
        self.dof_names = list(self.robot_config.dof_names)
        assert self.body_names == list(self.robot_config.body_names)
        self.num_dof = len(self.dof_names)
        self.num_bodies = len(self.body_names)
        self._body_list = list(self.body_names)
        return self.num_dof, self.num_bodies, self.dof_names, self.body_names

    def create_envs(self, num_envs, env_origins, base_init_state):
        self.num_envs = num_envs
        self.env_origins = env_origins
        self.base_init_state = base_init_state
        g = torch.Generator().manual_seed(777)
        dr = self.env_config.domain_rand
        N = num_envs
        u = lambda *s: torch.rand(*s, generator=g)
        r = dr.base_com_range
        self._base_com_bias = torch.stack([u(N) * (r.x[1] - r.x[0]) + r.x[0], u(N) * (r.y[1] - r.y[0]) + r.y[0], u(N) * (r.z[1] - r.z[0]) + r.z[0]], -1)
        self._link_mass_scale = u(N, len(dr.randomize_link_body_names)) * (dr.link_mass_range[1] - dr.link_mass_range[0]) + dr.link_mass_range[0]
        self._base_mass_scale = torch.ones(N, 1)
        self.friction_coeffs = (u(N, 1, 1) * (dr.friction_range[1] - dr.friction_range[0]) + dr.friction_range[0])
        self._process_dof_props()

    def _process_dof_props(self):
        # reference: isaacgym.py:_process_dof_props (soft limits from config)
        rc = self.robot_config
        D = self.num_dof
        lo = torch.tensor(rc.dof_pos_lower_limit_list, dtype=torch.float)
        hi = torch.tensor(rc.dof_pos_upper_limit_list, dtype=torch.float)
        self.hard_dof_pos_limits = torch.stack([lo, hi], -1)
        self.dof_vel_limits = torch.tensor(rc.dof_vel_limit_list, dtype=torch.float)
        self.torque_limits = torch.tensor(rc.dof_effort_limit_list, dtype=torch.float)
        self.dof_pos_limits = torch.zeros(D, 2)
        self.dof_pos_limits_termination = torch.zeros(D, 2)
        for i in range(D):
            m = (self.hard_dof_pos_limits[i, 0] + self.hard_dof_pos_limits[i, 1]) / 2
            r = self.hard_dof_pos_limits[i, 1] - self.hard_dof_pos_limits[i, 0]
            s = self.env_config.rewards.reward_limit.soft_dof_pos_limit
            self.dof_pos_limits[i, 0] = m - 0.5 * r * s
            self.dof_pos_limits[i, 1] = m + 0.5 * r * s
            s = self.env_config.termination_scales.termination_close_to_dof_pos_limit
            self.dof_pos_limits_termination[i, 0] = m - 0.5 * r * s
            self.dof_pos_limits_termination[i, 1] = m + 0.5 * r * s

    def get_dof_limits_properties(self):
        return self.dof_pos_limits, self.dof_vel_limits, self.torque_limits

    def find_rigid_body_indice(self, name):
        return self.body_names.index(name) if name in self.body_names else -1

    def prepare_sim(self):
        N, D, B = self.num_envs, self.num_dof, self.num_bodies
        self.all_root_states = torch.zeros(N, 13)
        self.all_root_states[:] = self.base_init_state
        self.all_root_states[:, :3] += self.env_origins
        self.robot_root_states = self.all_root_states
        self.base_quat = self.robot_root_states[..., 3:7]
        self.dof_state = torch.zeros(N * D, 2)
        self.dof_pos = self.dof_state.view(N, D, 2)[..., 0]
        self.dof_vel = self.dof_state.view(N, D, 2)[..., 1]
        self.contact_forces = torch.zeros(N, B, 3)
        self._rigid_body_pos = torch.zeros(N, B, 3)
        self._rigid_body_rot = torch.zeros(N, B, 4)
        self._rigid_body_vel = torch.zeros(N, B, 3)
        self._rigid_body_ang_vel = torch.zeros(N, B, 3)
        self._fk()

    def _fk(self):
        p, r, v, w = sim_fk(self.skel, self.robot_root_states, self.dof_pos, self.dof_vel)
        self._rigid_body_pos[:] = p
        self._rigid_body_rot[:] = r
        self._rigid_body_vel[:] = v
        self._rigid_body_ang_vel[:] = w

    # --- replay -----------------------------------------------------------------
    def set_replay(self, root, dof_pos, dof_vel, contact, start_frame=0):
        """root [T,N,13], dof_pos/dof_vel [T,N,D], contact [T,N,B,3]."""
        self.replay = dict(root=root, dof_pos=dof_pos, dof_vel=dof_vel, contact=contact)
        self._frame = start_frame

    def refresh_sim_tensors(self):
        pass

    def apply_torques_at_dof(self, torques):
        self.applied_torques = torques

    def simulate_at_each_physics_step(self):
        self._substep += 1
        if self._substep % self.decimation != 0:
            return
        if self.replay is not None:
            k = self._frame % self.replay["root"].shape[0]
            self.robot_root_states[:] = self.replay["root"][k]
            self.dof_pos[:] = self.replay["dof_pos"][k]
            self.dof_vel[:] = self.replay["dof_vel"][k]
            self.contact_forces[:] = self.replay["contact"][k]
            self._frame += 1
        self._fk()

    def set_actor_root_state_tensor(self, ids, states):
        pass

    def set_dof_state_tensor(self, ids, states):
        pass

    def clear_lines(self):
        pass

    def draw_sphere(self, *a, **k):
        pass

    def render(self, *a, **k):
        pass
