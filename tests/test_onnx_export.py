"""Deploy-side contract: policies export to the ONNX graph signature the reference's exporters produce
(utils/inference_helpers.py:13-52,95-138) and the exported graph computes what the actor computes.

Fixture: tests/golden/onnx/ref_horse_stance_pose_model_50000.onnx is the policy file the reference ships
(example/pretrained_horse_stance_pose/exported/model_50000.onnx, written by the reference's own exporter) — data only: a graph of
Gemm/Elu nodes and the trained weights.  It pins (1) the ONNX reader/evaluator, (2) our PPOActor definition (the reference's trained
weights load into it by name and reproduce the graph's output) and (3) the graph signature of our exporter.  `onnx`/`onnxruntime` are not
installed: files are read back with pbhc_amd.utils.onnx_lite (tolerance 1e-5 absolute on actions of magnitude ~1, fp32 GEMMs summed in
different orders)."""
import os

import numpy as np
import torch

from pbhc_amd.agents.agent_modules import ActorCritic
from pbhc_amd.agents.modules import PPOActor
from pbhc_amd.envs.env_config import determine_obs_dim
from pbhc_amd.utils import inference_helpers as ih
from pbhc_amd.utils import onnx_lite
from tests.helpers import GOLDEN, fixture_config

REF_ONNX = os.path.join(GOLDEN, "onnx", "ref_horse_stance_pose_model_50000.onnx")
ATOL = 1e-5


def _v1_actor_with_reference_weights():
    ref = onnx_lite.read_model(REF_ONNX)
    cfg = fixture_config("v1_g1_23dof_horse_stance.yaml", 4)
    determine_obs_dim(cfg)
    actor = PPOActor(cfg.robot.algo_obs_dim_dict, cfg.algo.config.module_dict.actor, cfg.robot.actions_dim, cfg.algo.config.init_noise_std)
    sd = actor.state_dict()
    for k, v in ref["initializers"].items():
        assert k.startswith("actor.")
        sd[k[len("actor."):]] = torch.from_numpy(v.copy())
    actor.load_state_dict(sd, strict=True)
    return ref, actor


def test_reference_exported_policy_runs_on_our_actor():
    ref, actor = _v1_actor_with_reference_weights()
    assert ref["opset"] == 13 and ref["inputs"] == [("actor_obs", [1, 380])] and ref["outputs"] == [("action", [1, 23])]
    assert [n["op_type"] for n in ref["nodes"]] == ["Gemm", "Elu", "Gemm", "Elu", "Gemm", "Elu", "Gemm"]
    x = torch.randn(1, 380, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = actor.act_inference(x).numpy()
    got = onnx_lite.run(ref, {"actor_obs": x.numpy()})[0]
    assert np.abs(want).max() > 0.1 and np.abs(got - want).max() <= ATOL


def test_v1_export_has_the_reference_graph_signature(tmp_path):
    ref, actor = _v1_actor_with_reference_weights()
    example = {"actor_obs": torch.randn(4, 380, generator=torch.Generator().manual_seed(1))}
    file = ih.export_policy_as_onnx({"actor": actor}, str(tmp_path), "model_0.onnx", example)
    ours = onnx_lite.read_model(file)
    assert (ours["opset"], ours["inputs"], ours["outputs"]) == (ref["opset"], ref["inputs"], ref["outputs"])
    assert [(n["op_type"], n["attrs"]) for n in ours["nodes"]] == [(n["op_type"], n["attrs"]) for n in ref["nodes"]]
    assert list(ours["initializers"]) == list(ref["initializers"])
    for k, v in ref["initializers"].items():
        assert np.array_equal(ours["initializers"][k], v), k
    assert ih.check_onnx(file, {"actor": actor}, example, atol=ATOL) <= ATOL


def test_v2_export_policy_and_encoder(tmp_path):
    cfg = fixture_config("v2_g1_23dof_student.yaml", 4)
    determine_obs_dim(cfg)
    torch.manual_seed(3)
    ac = ActorCritic(cfg.robot.algo_obs_dim_dict, cfg.algo.config.module_dict, cfg.robot.actions_dim, cfg.algo.config.init_noise_std)
    dims = cfg.robot.algo_obs_dim_dict
    S = cfg.algo.config.module_dict.actor.motion_encoder.tsteps              # the env emits S future steps of the per-step width
    width = {"actor_obs": dims["actor_obs"], "future_motion_targets": dims["future_motion_targets"] * S, "prop_history": dims["prop_history"]}
    example = {k: torch.randn(2, w) for k, w in width.items()}
    file = ih.export_policy_and_encoder_as_onnx({"actor": ac.actor}, str(tmp_path), "model_0.onnx", example)
    m = onnx_lite.read_model(file)
    assert [n for n, _ in m["inputs"]] == ["actor_obs", "future_motion_targets", "prop_history"]
    assert m["inputs"][0][1] == [1, dims["actor_obs"]] and m["outputs"] == [("action", [1, cfg.robot.actions_dim])]
    ops = [n["op_type"] for n in m["nodes"]]
    assert ops.count("Conv") == 4 and "Concat" in ops                      # two Conv1d per encoder, as the reference's modules export
    assert ih.check_onnx(file, {"actor": ac.actor}, example, atol=ATOL) <= ATOL
    # the GPU formulation of the encoders (unfolded-window GEMMs) computes the same function as the exported Conv graph
    ac.actor.motion_encoder.unfold_gemm = ac.actor.history_encoder.unfold_gemm = True
    with torch.no_grad():
        t = {k: v[:1] for k, v in example.items()}
        want = ac.actor(dict(t), True).numpy()
    got = onnx_lite.run(m, {k: v.numpy() for k, v in t.items()})[0]
    assert np.abs(got - want).max() <= ATOL
