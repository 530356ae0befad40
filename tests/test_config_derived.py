"""The product's config path — `pbhc_amd.utils.config.load_config` (interpolation resolver) + `pbhc_amd/envs/env_config.build` (what the
HIP env is parameterised with) — against the values the REFERENCE derived from the same trees: tests/golden/config_derived.json, recorded
off the live reference env object by oracle/ref_harness/gen_config_golden.py with the reference's config resolved by an independent
resolver (oracle/ref_harness/ref_config.py).  Closes the common-mode link of the env goldens (their generator composes the reference's
config through the product's resolver): a wrong `${eval:...}` resolution or a wrong dt / scale / gain / index list now fails here.
Reference: utils/helpers.py:47-126, envs/legged_base_task/legged_robot_base.py:39-110,167-233, envs/motion_tracking/motion_tracking.py:98-170.
"""
import glob
import json
import os

import numpy as np
import pytest

from pbhc_amd.envs import env_config
from pbhc_amd.skeleton import Skeleton
from pbhc_amd.utils.config import load_config
from tests.helpers import GOLDEN

DERIVED = json.load(open(os.path.join(GOLDEN, "config_derived.json")))
CONFIGS = sorted(DERIVED.keys())


def _plain(n):
    if isinstance(n, dict):
        return {k: _plain(v) for k, v in n.items()}
    if isinstance(n, list):
        return [_plain(v) for v in n]
    return n


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "configs", "*.yaml"))))
def test_two_independent_resolvers_agree_on_the_whole_tree(name):
    """every resolved value of the fixture tree, product resolver vs the independently written one (different algorithm: innermost-first
    textual substitution with literal-evaluated arguments vs recursive descent)"""
    from oracle.ref_harness import ref_config as RC

    path = os.path.join(GOLDEN, "configs", name)
    a = _plain(RC.load(path, {"num_envs": 64}, now="t"))
    b = _plain(load_config(path, {"num_envs": 64}, now="t"))
    assert json.dumps(a, sort_keys=True) == json.dumps(b, sort_keys=True)


def _build(name):
    from pbhc_amd.envs.motion_tracking import _TopView

    cfg = load_config(os.path.join(GOLDEN, "configs", name), {"num_envs": 8}, now="t")
    general = name.startswith("v2_")
    if general:
        cfg.env.config["_target_"] = "humanoidverse.envs.motion_tracking.general_tracking.LeggedRobotGeneralTracking"
    sk = Skeleton.from_motion_config(cfg.robot.motion)

    class ML:
        has_contact_mask = "walk" not in name

    c, L = env_config.build(_TopView(cfg.env.config), sk, ML(), 8, "cpu", len(cfg.domain_rand.randomize_link_body_names), seed=0, mode=1 if general else 0)
    return cfg, sk, c, L


@pytest.mark.parametrize("name", CONFIGS)
def test_env_config_reproduces_what_the_reference_derived(name):
    ref = DERIVED[name]
    cfg, sk, c, L = _build(name)
    D = ref["num_dof"]
    # sizes, timing
    assert (sk.num_dof, sk.num_bodies) == (ref["num_dof"], ref["num_bodies"]) and ref["dim_actions"] == D
    assert sk.num_bodies_ext - sk.num_bodies == ref["num_extend_bodies"]
    assert c.dt == pytest.approx(ref["dt"], rel=1e-7) and L.max_episode_length == ref["max_episode_length"]
    assert c.max_episode_length_s == pytest.approx(ref["max_episode_length_s"], rel=1e-7)
    # observation bookkeeping (pre_process_config): per-key dims, group dims, the slices of every key inside its group
    for k, v in ref["obs_dims"].items():
        assert int(L.obs_dims[k]) == v, k
    groups, _, _ = env_config.determine_obs_dim(cfg)              # what the algo sizes its networks with (robot.algo_obs_dim_dict)
    assert {g: int(v) for g, v in groups.items()} == ref["algo_obs_dim_dict"]
    S = int(c.future_num_steps)
    for g, slices in ref["obs_slices"].items():
        pos = 0
        for key in sorted(cfg.obs.obs_dict[g]):                 # legged_robot_base.py:787-793: sorted-key concatenation
            a, b = slices[key]
            assert a == pos, (g, key)
            pos = b
        assert pos == ref["algo_obs_dim_dict"][g], g
        # the row the env hands out: future keys list their PER-STEP dim, the tensor is [N, steps x dim] (ppo_mimic.py:206-216)
        per_step = all(k.startswith("future_motion_") for k in cfg.obs.obs_dict[g]) and S > 0
        assert L.group_dims[g] == pos * (S if per_step else 1), g
    # history buffers (HistoryHandler, history_handler.py:10-31): [length, dim] per key
    assert {k: [int(L.hist_len[k]), int(L.obs_dims[k])] for k in L.hist_keys} == ref["history"]
    # rewards (_prepare_reward_function): zero scales dropped, the rest x dt, loop order = dict order, `termination` outside the loop
    assert list(L.reward_scales.keys()) == ref["reward_scale_order"]
    for k, v in ref["reward_scales_dt"].items():
        assert L.reward_scales[k] == pytest.approx(v, rel=1e-12), k
    assert L.reward_names == ref["reward_names"]
    for i, k in enumerate(L.reward_names):
        assert c.term_scale[i] == pytest.approx(ref["reward_scales_dt"][k], rel=1e-6), k
    assert bool(c.use_vec_reward) == ref["use_vec_reward"]
    # control: gains, defaults, limits, action scaling (legged_robot_base.py:74-110; limits: simulator get_dof_limits_properties)
    f32 = lambda v: np.asarray(v, np.float32)
    arr = lambda a: np.asarray([a[i] for i in range(D)], np.float32)
    np.testing.assert_array_equal(arr(c.p_gains), f32(ref["p_gains"]))
    np.testing.assert_array_equal(arr(c.d_gains), f32(ref["d_gains"]))
    np.testing.assert_array_equal(arr(c.default_dof_pos), f32(ref["default_dof_pos"]))
    np.testing.assert_array_equal(arr(c.torque_limits), f32(ref["torque_limits"]))
    np.testing.assert_array_equal(arr(c.dof_vel_limits), f32(ref["dof_vel_limits"]))
    np.testing.assert_array_equal(arr(c.action_scale), f32(ref["action_scale"]))
    soft = np.asarray([[c.soft_dof_pos_limits[i][0], c.soft_dof_pos_limits[i][1]] for i in range(D)], np.float32)
    np.testing.assert_allclose(soft, f32(ref["dof_pos_limits"]), rtol=0, atol=2e-7)      # m -/+ 0.5 r s in fp32 (one rounding of the product order)
    assert c.action_clip_value == pytest.approx(ref["action_clip_value"], rel=1e-7)
    # body index lists
    ints = lambda v: [int(x) for x in v]
    assert L.feet == ints(ref["feet_indices"]) and L.penalised == ints(ref["penalised_contact_indices"])
    assert L.termination_contact == ints(ref["termination_contact_indices"])
    assert L.upper == ints(ref["upper_body_id"]) and L.lower == ints(ref["lower_body_id"]) and L.track == ints(ref["motion_tracking_id"])
    if "key_body_id" in ref:
        assert L.key == ints(ref["key_body_id"])
    if "anchor_index" in ref:
        assert c.anchor_index == ref["anchor_index"]
    if "tar_obs_steps" in ref:
        assert [c.future_steps[i] for i in range(c.future_num_steps)] == ints(ref["tar_obs_steps"])
