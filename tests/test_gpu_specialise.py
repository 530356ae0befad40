"""The step kernel specialised to the env's config (pbhc_amd/specialise.py, csrc/pbhc_env_step_spec.hip) against the generic one.

The `-m gpu` suite runs on the specialised kernel — what an env uses by default and what `bench.py` measures.  Here: the GENERIC kernel
(`PBHC_SPECIALISE=off`: run-time config, per-element observation maps) replays the same four reference traces; the two kernels, stepped
side by side on the judged workload's config WITH observation noise, produce the same outputs (same source, same arithmetic per element,
the same noise stream: keyed by env / step / row / element, not by the writing lane); and the safety of the mechanism — an object built
from another config is refused, the modes do what they say."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.helpers import build_hip_env
from tests.test_gpu_parity import CASES as V1_CASES, close
from tests.test_gpu_parity_v2 import CASES as V2_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag,cfgname", V1_CASES)
def test_generic_kernel_replays_the_reference_traces(monkeypatch, tag, cfgname):
    monkeypatch.setenv("PBHC_SPECIALISE", "off")
    from tests.test_gpu_parity import test_env_step_matches_reference_trace

    test_env_step_matches_reference_trace(tag, cfgname)


@pytest.mark.parametrize("tag,cfgname,robot", V2_CASES)
def test_generic_kernel_replays_the_general_tracking_traces(monkeypatch, tag, cfgname, robot):
    monkeypatch.setenv("PBHC_SPECIALISE", "off")
    from tests.test_gpu_parity_v2 import test_general_tracking_step_matches_reference_trace

    test_general_tracking_step_matches_reference_trace(tag, cfgname, robot)


@pytest.mark.parametrize("cfgname,general,N", [("v1_g1_23dof_walk.yaml", False, 1024), ("v2_g1_29dof_teacher.yaml", True, 256)])
def test_specialised_and_generic_kernels_agree_with_noise_on(cfgname, general, N):
    """two envs of one config and one seed — one on the generic kernel, one specialised — fed the same replay window and actions for 6 steps
    (in-kernel resets and their Philox draws included, observation noise as shipped): observations, rewards, resets and state agree to
    2e-6 (the FK chain's fused multiply-adds may be paired differently by the two compilations; everything else is the same arithmetic)."""
    envs = []
    for mode in ("off", "jit"):
        torch.manual_seed(7)
        np.random.seed(7)
        cfg, env = build_hip_env(cfgname, N, general=general, noise_off=False, overrides={"domain_rand.push_robots": False})
        assert env.specialise(mode) == (mode == "jit") and env.is_specialised == (mode == "jit")
        envs.append(env)
    _step_side_by_side(envs[0], envs[1], N, 2e-6)


# measurement switches of the specialised build that stay in the source (profiles/round4_k_env_step_variants.txt): each has to compute what the
# default build computes — the same arithmetic per element, another schedule / cache policy / writer (the walk instead of the pointer-jumping FK
# pairs its fused multiply-adds differently: twists to 1e-5 — tests/test_gpu_fk.py — which the velocity-difference features scale up to 2.3e-5)
SWITCHES = [("-DPBHC_NO_NT_STORES", {}, 0.0), ("-DPBHC_NT_LOADS", {}, 0.0), ("-DPBHC_EARLY_OPERANDS", {}, 0.0), ("-DPBHC_WIDE_ROWS", {}, 0.0),
            ("-DPBHC_NO_HISTB", {}, 0.0), ("-DPBHC_NO_XCD_MAP -DPBHC_PTR_BURST", {}, 0.0), ("", {"PBHC_ROW_HELP_SHARE": "0.23"}, 0.0), ("-DPBHC_FK_WALK", {}, 5e-5)]


@pytest.mark.parametrize("defs,envvars,tol", SWITCHES)
def test_build_switches_of_the_specialised_kernel_compute_the_same_step(monkeypatch, defs, envvars, tol):
    N = 512
    envs = []
    for variant in (False, True):
        if variant:
            monkeypatch.setenv("PBHC_SPEC_DEFINES", defs)
            for k, v in envvars.items():
                monkeypatch.setenv(k, v)
        torch.manual_seed(7)
        np.random.seed(7)
        cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", N, noise_off=False, overrides={"domain_rand.push_robots": False})
        assert env.specialise("jit") and env.is_specialised
        envs.append(env)
    if envvars.get("PBHC_ROW_HELP_SHARE"):
        assert envs[1].layout.helper_elements > 100 and envs[0].layout.helper_elements == 0       # (the variant really hands runs to the dynamics waves)
    _step_side_by_side(envs[0], envs[1], N, tol)


def _step_side_by_side(a, b, N, tol):
    import bench

    assert a._seed == b._seed
    obs_a, obs_b = a.reset_all(), b.reset_all()
    rep = bench.make_replay_on_device(a, 8, seed=5)
    a.simulator.set_replay(*rep)
    b.simulator.set_replay(*[t.clone() for t in rep])
    # copy the complete state of `a` into `b` (the two reset_all() calls drew their start phases from the same torch stream position only
    # if nothing else drew in between: make it explicit)
    for name in ("motion_start_times", "motion_len", "_kp_scale", "_kd_scale", "_rfi_lim_scale", "_rao_scale", "action_delay_idx", "_episode_length_buf",
                 "last_episode_length_buf", "end_time_ratio_buf", "actions", "last_actions", "actions_after_delay", "action_queue", "last_dof_pos", "last_dof_vel", "_hist",
                 "_episode_sums", "feet_air_time", "last_contacts", "last_contacts_filt", "contacts", "contacts_filt", "globals", "env_origins"):
        getattr(b, name).copy_(getattr(a, name))
    b.simulator.robot_root_states.copy_(a.simulator.robot_root_states)
    b.simulator.dof_state.copy_(a.simulator.dof_state)
    for t_a, t_b in zip((a.simulator._base_com_bias, a.simulator._link_mass_scale, a.simulator.friction_coeffs, a.simulator._base_mass_scale),
                        (b.simulator._base_com_bias, b.simulator._link_mass_scale, b.simulator.friction_coeffs, b.simulator._base_mass_scale)):
        t_b.copy_(t_a)
    b._friction_flat.copy_(a._friction_flat)
    gen = torch.Generator(device=DEV).manual_seed(11)
    nreset = 0
    for k in range(6):
        if k == 3:
            a.episode_length_buf[::9] = 10 ** 6
            b.episode_length_buf[::9] = 10 ** 6
        act = 0.4 * torch.randn(N, a.num_dof, device=DEV, generator=gen)
        oa, ra, da, _ = a.step({"actions": act})
        ob, rb, db, _ = b.step({"actions": act.clone()})
        torch.cuda.synchronize()
        w = f"step {k}: "
        assert torch.equal(da, db), w + "resets"
        nreset += int(da.sum())
        same = (lambda x, y, what: close(x, y, tol, what, rtol=tol)) if tol > 0.0 else (lambda x, y, what: _bit_equal(x, y, what))
        same(rb, ra, w + "rewards")
        for g in oa:
            same(ob[g], oa[g], w + g)
        for name in ("_hist", "_episode_sums", "torques", "motion_start_times", "_kp_scale", "action_queue", "feet_air_time"):
            same(getattr(b, name), getattr(a, name), w + name)
        assert torch.equal(a.action_delay_idx, b.action_delay_idx) and torch.equal(a.episode_length_buf, b.episode_length_buf)
        a.wait_finalize(); b.wait_finalize()
        close(b.globals, a.globals, 1e-9, w + "globals (sigma, curricula, log means)", rtol=max(tol, 2e-6))
    assert nreset >= N // 9


def _bit_equal(x, y, what):
    assert torch.equal(x, y), f"{what}: max abs difference {(x.double() - y.double()).abs().max().item():.3e}"


def test_attach_refuses_an_object_built_from_another_config():
    from pbhc_amd import _lib
    from pbhc_amd import specialise as S

    _, walk = build_hip_env("v1_g1_23dof_walk.yaml", 64)
    _, horse = build_hip_env("v1_g1_23dof_horse_stance.yaml", 64)          # another reward table (teleop_contact_mask on)
    lib = _lib.lib()
    c = _lib.PbhcEnvConfig()
    _lib.check(lib.pbhc_env_get_config(horse._env, C.byref(c)))
    so_horse = S.ensure(c, "jit")
    walk.specialise("off")
    assert lib.pbhc_env_attach_specialised(walk._env, so_horse.encode()) == _lib.K["PBHC_EINVAL"]
    assert b"different config" in lib.pbhc_last_error() and not walk.is_specialised
    assert lib.pbhc_env_attach_specialised(walk._env, b"/nonexistent/file.so") == _lib.K["PBHC_EINVAL"]
    assert walk.specialise("jit") and walk.is_specialised
    # one object serves every env count of a config (num_envs is a run-time member)
    _, walk2 = build_hip_env("v1_g1_23dof_walk.yaml", 96)
    c2 = _lib.PbhcEnvConfig()
    _lib.check(lib.pbhc_env_get_config(walk2._env, C.byref(c2)))
    _lib.check(lib.pbhc_env_get_config(walk._env, C.byref(c)))
    assert S.key_of(S.emit_header(c)) == S.key_of(S.emit_header(c2))


def test_cached_mode_never_compiles(monkeypatch):
    from pbhc_amd import specialise as S

    monkeypatch.setenv("PBHC_SPECIALISE", "off")                                                                           # (not at construction either)
    _, env = build_hip_env("v1_g1_23dof_walk.yaml", 32, overrides={"env.config.normalization.clip_observations": 77.0})     # a config nobody built yet
    assert not env.is_specialised
    called = []
    monkeypatch.setattr(S, "compile_object", lambda *a, **k: called.append(1))
    assert env.specialise("cached") is False and not called and not env.is_specialised
