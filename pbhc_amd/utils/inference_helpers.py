"""Policy exporters for deployment, with the reference's names and file contracts
(reference: humanoidverse/utils/inference_helpers.py:6-52 `export_policy_as_jit` / `export_policy_as_onnx`, :95-138
`export_policy_and_encoder_as_onnx`; consumer: deploy/urcirobot.py:326-374, which feeds the named inputs to onnxruntime).

Same graph signature as the reference's exported files: input `actor_obs` (and `future_motion_targets`, `prop_history` for the
encoder policies), output `action`, opset 13, parameters embedded.  The actor is deep-copied to the CPU first, as the reference does, so
the export traces plain `nn.Linear` / `nn.Conv1d` modules (none of the HIP training paths).  `torch.onnx.export` post-processes the
serialised graph with the `onnx` package only to splice in onnxscript functions, which these policies do not have; when `onnx` is not
installed that one step is bypassed.  `check_onnx` re-reads the file with `onnx_lite` and compares it with the live actor.
"""
from __future__ import annotations

import copy
import importlib.util
import os
import warnings

import numpy as np
import torch
from torch import nn

from . import onnx_lite


class _ActorInference(nn.Module):
    def __init__(self, actor):
        super().__init__()
        self.actor = actor

    def forward(self, actor_obs):
        return self.actor.act_inference(actor_obs)


class _EncoderActorInference(nn.Module):
    """motion encoder + history encoder + actor MLP on three named inputs (inference_helpers.py:101-118)."""

    def __init__(self, actor):
        super().__init__()
        self.actor = actor

    def forward(self, actor_obs, future_motion_targets, prop_history):
        z = torch.cat([actor_obs, self.actor.motion_encoder(future_motion_targets), self.actor.history_encoder(prop_history)], dim=-1)
        return self.actor.actor_module(z)


def _onnx_export(module, args, file, input_names, output_names):
    kw = dict(input_names=input_names, output_names=output_names, export_params=True, opset_version=13, do_constant_folding=True, dynamo=False)
    module.eval()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if importlib.util.find_spec("onnx") is not None:
            torch.onnx.export(module, args, file, **kw)
            return
        from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as pu

        splice = pu._add_onnxscript_fn
        pu._add_onnxscript_fn = lambda proto, custom_opsets: proto
        try:
            torch.onnx.export(module, args, file, **kw)
        finally:
            pu._add_onnxscript_fn = splice


def _cpu_example(example_obs_dict, key):
    return example_obs_dict[key].detach().to("cpu", torch.float32)[:1].contiguous()


def export_policy_as_jit(actor_critic, path, exported_policy_name):
    os.makedirs(path, exist_ok=True)
    model = copy.deepcopy(actor_critic.actor).to("cpu")
    torch.jit.script(model).save(os.path.join(path, exported_policy_name))


def export_policy_as_onnx(inference_model, path, exported_policy_name, example_obs_dict):
    os.makedirs(path, exist_ok=True)
    file = os.path.join(path, exported_policy_name)
    wrapper = _ActorInference(copy.deepcopy(inference_model["actor"]).to("cpu"))
    _onnx_export(wrapper, (_cpu_example(example_obs_dict, "actor_obs"),), file, ["actor_obs"], ["action"])
    return file


def export_policy_and_encoder_as_onnx(inference_model, path, exported_policy_name, example_obs_dict):
    os.makedirs(path, exist_ok=True)
    file = os.path.join(path, exported_policy_name)
    names = ["actor_obs", "future_motion_targets", "prop_history"]
    wrapper = _EncoderActorInference(copy.deepcopy(inference_model["actor"]).to("cpu"))
    _onnx_export(wrapper, tuple(_cpu_example(example_obs_dict, k) for k in names), file, names, ["action"])
    return file


def check_onnx(file, inference_model, example_obs_dict, atol=1e-5):
    """Evaluate the exported graph (numpy, `onnx_lite`) on the example observation and compare with the live actor; returns max |diff|."""
    m = onnx_lite.read_model(file)
    feeds = {n: _cpu_example(example_obs_dict, n).numpy() for n, _ in m["inputs"]}
    got = onnx_lite.run(m, feeds)[0]
    actor = copy.deepcopy(inference_model["actor"]).to("cpu")
    with torch.no_grad():
        t = {k: torch.from_numpy(v) for k, v in feeds.items()}
        want = (_EncoderActorInference(actor)(**t) if len(feeds) == 3 else _ActorInference(actor)(**t)).numpy()
    err = float(np.abs(got - want).max())
    if not err <= atol:
        raise RuntimeError(f"{file}: exported graph differs from the actor by {err}")
    return err
