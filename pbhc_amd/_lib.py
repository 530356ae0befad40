"""ctypes binding of libpbhc_hip.so (include/pbhc_hip.h).

The ctypes Structures are generated from the header text itself, so the Python view of the ABI
cannot drift from the C one; sizes are cross-checked against `pbhc_sizeof_*()` at load.
There is no fallback: if the library is missing or was built for another ABI, using the
product raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "pbhc_hip.h")
LIB_PATH = os.environ.get("PBHC_LIB", os.path.join(_HERE, "libpbhc_hip.so"))   # PBHC_LIB: diagnostic builds only

_CTYPES = {
    "int32_t": C.c_int32, "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "float": C.c_float, "double": C.c_double, "uint8_t": C.c_uint8, "int": C.c_int,
}


def _strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def _parse_header(path):
    text = _strip_comments(open(path).read())
    consts = {}
    for m in re.finditer(r"#define\s+(PBHC_\w+)\s+\(?(-?\d+)\)?", text):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        val = -1
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                val = int(eval(v, {}, consts))
            else:
                name, val = item, val + 1
            consts[name] = val
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            base, first_ptr, rest = mm.group(2), mm.group(3), mm.group(4)
            variables = [v.strip() for v in rest.split(",")]
            variables[0] = first_ptr + variables[0]
            for var in variables:
                is_ptr = var.startswith("*")
                var = var.lstrip("* ")
                dims = [int(eval(d, {}, consts)) for d in re.findall(r"\[([^\]]+)\]", var)]
                vname = re.match(r"^(\w+)", var).group(1)
                if is_ptr:
                    ct = C.c_void_p
                elif base in _CTYPES:
                    ct = _CTYPES[base]
                elif base in structs:
                    ct = structs[base]
                else:
                    raise ValueError(f"unknown type {base} in {name}")
                for d in reversed(dims):
                    ct = ct * d
                fields.append((vname, ct))
        structs[name] = type(name, (C.Structure,), {"_fields_": fields})
    return consts, structs


K, _S = _parse_header(HEADER)
PbhcSkeleton = _S["PbhcSkeleton"]
PbhcOutMap = _S["PbhcOutMap"]
PbhcEnvConfig = _S["PbhcEnvConfig"]
PbhcMotionTable = _S["PbhcMotionTable"]
PbhcStepIO = _S["PbhcStepIO"]
PbhcMlpSample = _S["PbhcMlpSample"]
PbhcMlpInput = _S["PbhcMlpInput"]
PbhcConvEncoder = _S["PbhcConvEncoder"]

EXPORTS = ["pbhc_abi_version", "pbhc_last_error", "pbhc_sizeof_env_config", "pbhc_sizeof_step_io", "pbhc_motion_build",
           "pbhc_motion_state", "pbhc_sim_fk", "pbhc_env_create", "pbhc_env_destroy", "pbhc_env_step", "pbhc_gae",
           "pbhc_env_profile", "pbhc_env_profile_read", "pbhc_env_profile_overhead", "pbhc_ppo_loss", "pbhc_ppo_loss_scratch_floats", "pbhc_adam_clip",
           "pbhc_policy_sample", "pbhc_rollout_post", "pbhc_act_bwd_bias", "pbhc_env_finalize", "pbhc_act_bwd_partials", "pbhc_colsum_final", "pbhc_adam_clip2", "pbhc_debug_rotations", "pbhc_motion_build_batch",
           "pbhc_linear_act_fwd", "pbhc_env_config_lds_bytes", "pbhc_linear_act_fwd_out", "pbhc_debug_out_bwd_variant", "pbhc_gather_rows", "pbhc_linear_dgrad_act", "pbhc_gemm_debug_force_shape", "pbhc_linear_wgrad", "pbhc_linear_wgrad_parts", "pbhc_linear_act_fwd_strided",
           "pbhc_env_step_launch", "pbhc_env_step_finish", "pbhc_mlp_fwd", "pbhc_mlp_fwd_lds_bytes", "pbhc_mlp_pack", "pbhc_mlp_packed_floats", "pbhc_rollout_post2", "pbhc_mlp_fwd_sample", "pbhc_linear_out_bwd",
           "pbhc_env_get_config", "pbhc_env_attach_specialised", "pbhc_env_is_specialised", "pbhc_env_config_finalize", "pbhc_kl_lr_rule", "pbhc_debug_fk", "pbhc_mlp_fwd_cat", "pbhc_conv_encoder_fwd", "pbhc_conv_encoder_lds_bytes"]


class PbhcError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise PbhcError(
            f"{LIB_PATH} not found: the HIP extension is required (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pbhc_amd/csrc`."
        )
    lib = C.CDLL(LIB_PATH)
    lib.pbhc_last_error.restype = C.c_char_p
    if lib.pbhc_abi_version() != K["PBHC_ABI_VERSION"]:
        raise PbhcError("libpbhc_hip.so ABI version does not match include/pbhc_hip.h")
    if lib.pbhc_sizeof_env_config() != C.sizeof(PbhcEnvConfig) or lib.pbhc_sizeof_step_io() != C.sizeof(PbhcStepIO):
        raise PbhcError("struct layout mismatch between libpbhc_hip.so and include/pbhc_hip.h — rebuild")
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    lib.pbhc_motion_build.argtypes = [C.POINTER(PbhcSkeleton), vp, vp, vp, i, f, vp, vp, vp]
    lib.pbhc_motion_build_batch.argtypes = [C.POINTER(PbhcSkeleton), vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp]
    lib.pbhc_motion_state.argtypes = [C.POINTER(PbhcMotionTable), i, i, vp, vp, vp, i, vp, vp]
    lib.pbhc_sim_fk.argtypes = [C.POINTER(PbhcSkeleton), vp, vp, vp, i, i, vp, vp]
    lib.pbhc_debug_fk.argtypes = [C.POINTER(PbhcSkeleton), vp, vp, vp, i, i, vp, vp]
    lib.pbhc_env_create.argtypes = [C.POINTER(PbhcEnvConfig), C.POINTER(PbhcMotionTable), vp, C.POINTER(vp)]
    lib.pbhc_env_destroy.argtypes = [vp]
    lib.pbhc_env_destroy.restype = None
    lib.pbhc_env_get_config.argtypes = [vp, C.POINTER(PbhcEnvConfig)]
    lib.pbhc_env_attach_specialised.argtypes = [vp, C.c_char_p]
    lib.pbhc_env_is_specialised.argtypes = [vp]
    lib.pbhc_env_config_finalize.argtypes = [C.POINTER(PbhcEnvConfig), C.POINTER(PbhcEnvConfig)]
    lib.pbhc_env_config_lds_bytes.argtypes = [C.POINTER(PbhcEnvConfig)]
    lib.pbhc_env_step.argtypes = [vp, C.POINTER(PbhcStepIO), vp]
    lib.pbhc_env_step_launch.argtypes = [vp, C.POINTER(PbhcStepIO), vp]
    lib.pbhc_env_step_finish.argtypes = [vp, C.POINTER(PbhcStepIO), vp]
    lib.pbhc_env_finalize.argtypes = [vp, vp, C.c_double, vp]
    lib.pbhc_env_profile.argtypes = [vp, i]
    lib.pbhc_env_profile_read.argtypes = [vp, C.POINTER(C.c_float), i, C.POINTER(C.c_int)]
    lib.pbhc_env_profile_overhead.argtypes = [vp, vp, C.POINTER(C.c_float)]
    lib.pbhc_ppo_loss.argtypes = [vp] * 10 + [i, i, i, f, f, f, i, f, i] + [vp] * 7 + [vp]
    lib.pbhc_ppo_loss_scratch_floats.argtypes = [i]
    lib.pbhc_kl_lr_rule.argtypes = [vp, i, vp, f, vp]
    lib.pbhc_act_bwd_bias.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp]
    lib.pbhc_act_bwd_partials.argtypes = [vp, vp, i, i, i, vp, vp, C.POINTER(C.c_int), vp]
    lib.pbhc_colsum_final.argtypes = [vp, i, vp]
    lib.pbhc_linear_out_bwd.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, vp, vp, vp, C.POINTER(C.c_int), vp]
    lib.pbhc_linear_act_fwd.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    lib.pbhc_linear_act_fwd_out.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp, vp, i, vp, vp]
    lib.pbhc_linear_dgrad_act.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int), i, i, i, i, vp]
    lib.pbhc_gemm_debug_force_shape.argtypes = [i]
    lib.pbhc_debug_out_bwd_variant.argtypes = [i]
    lib.pbhc_gather_rows.argtypes = [vp, i, vp, i, vp]
    lib.pbhc_debug_out_bwd_variant.restype = None
    lib.pbhc_linear_wgrad_parts.argtypes = [i, i, i]
    lib.pbhc_linear_act_fwd_strided.argtypes = [vp, i, C.c_longlong, vp, vp, vp, vp, i, C.c_longlong, i, i, i, i, i, vp]
    lib.pbhc_linear_wgrad.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    lib.pbhc_mlp_fwd.argtypes = [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(i), i, i, vp, i, i, vp]
    lib.pbhc_mlp_fwd_sample.argtypes = [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(i), i, i, i, C.POINTER(PbhcMlpSample), vp]
    lib.pbhc_conv_encoder_lds_bytes.argtypes = [C.POINTER(PbhcConvEncoder)]
    lib.pbhc_conv_encoder_lds_bytes.restype = C.c_size_t
    lib.pbhc_conv_encoder_fwd.argtypes = [vp, i, C.POINTER(PbhcConvEncoder), vp, i, i, vp]
    lib.pbhc_mlp_fwd_cat.argtypes = [C.POINTER(PbhcMlpInput), C.POINTER(vp), C.POINTER(vp), C.POINTER(i), i, i, vp, i, i, C.POINTER(PbhcMlpSample), vp]
    lib.pbhc_mlp_fwd_lds_bytes.argtypes = [C.POINTER(i), i]
    lib.pbhc_mlp_fwd_lds_bytes.restype = C.c_size_t
    lib.pbhc_mlp_pack.argtypes = [vp, i, i, vp, vp]
    lib.pbhc_mlp_packed_floats.argtypes = [i, i]
    lib.pbhc_mlp_packed_floats.restype = C.c_size_t
    lib.pbhc_gemm_debug_force_shape.restype = None
    lib.pbhc_adam_clip.argtypes = [vp, vp, vp, vp, i, vp, vp, f, f, f, f, f, vp, vp, vp]
    lib.pbhc_adam_clip2.argtypes = [vp, vp, vp, vp, i, i, vp, vp, f, f, f, f, f, i, vp, vp, vp]
    lib.pbhc_policy_sample.argtypes = [vp, vp, vp, i, i, i, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    lib.pbhc_rollout_post.argtypes = [vp, vp, vp, vp, i, i, f, vp, vp, vp, vp, vp, vp]
    lib.pbhc_rollout_post2.argtypes = [vp, vp, vp, vp, i, i, f, vp, vp, vp, vp, vp, vp, vp]
    lib.pbhc_gae.argtypes = [vp, vp, vp, vp, i, i, i, f, f, vp, vp, vp, vp]
    lib.pbhc_debug_rotations.argtypes = [i, vp, vp, vp, i, vp, vp]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise PbhcError(f"{what} failed ({rc}): {lib().pbhc_last_error().decode()}")


def ptr(t):
    """raw device pointer of a torch tensor; None -> NULL"""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def padded_width(dim):
    """Row pitch (floats) of the [N, dim] tensors the env kernel writes: a multiple of 32 floats (rows start on 128-byte lines) with at
    least two floats of padding (lanes past the end of a row store there instead of being predicated off)."""
    return (dim + 2 + 31) // 32 * 32


def require_gpu_rows(t, name, dtype=None, shape=None):
    """as require_gpu_tensor, but rows may be padded: [N, C] with stride (pitch >= C, 1)"""
    import torch

    if not (torch.is_tensor(t) and t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        raise PbhcError(f"{name}: expected a CUDA(HIP) [N, C] tensor with unit inner stride")
    if dtype is not None and t.dtype != dtype:
        raise PbhcError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise PbhcError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def require_gpu_tensor(t, name, dtype=None, shape=None):
    """host-side shape/dtype/device check before a raw pointer crosses the ABI"""
    import torch

    if not (torch.is_tensor(t) and t.is_cuda and t.is_contiguous()):
        raise PbhcError(f"{name}: expected a contiguous CUDA(HIP) tensor")
    if dtype is not None and t.dtype != dtype:
        raise PbhcError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise PbhcError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t
