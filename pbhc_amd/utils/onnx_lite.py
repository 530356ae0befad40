"""Minimal ONNX reader + numpy evaluator for exported policies.

The deploy side of the reference feeds the exported `.onnx` policy to onnxruntime (reference: deploy/urcirobot.py:326-374 on
utils/inference_helpers.py:13-52,95-138).  Neither `onnx` nor `onnxruntime` is a dependency of this package; to check that an exported
file computes what the trained actor computes, this module decodes the protobuf wire format of the handful of ONNX messages a policy file
uses (ModelProto / GraphProto / NodeProto / AttributeProto / TensorProto / ValueInfoProto, onnx.proto3 field numbers) and evaluates the
graph with numpy.  Reading executes nothing from the file.  Operator coverage is what MLP / Conv1d-encoder policies export to (opset 13):
Gemm, MatMul, Add, Mul, Relu, Elu, Tanh, Sigmoid, Concat, Conv (1-D), Reshape, Flatten, Transpose, Constant, Identity, Shape, Gather,
Unsqueeze, Squeeze, Slice.
"""
from __future__ import annotations

import struct

import numpy as np

_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def _varint(b, i):
    x = s = 0
    while True:
        c = b[i]
        i += 1
        x |= (c & 0x7F) << s
        if c < 0x80:
            return x, i
        s += 7


def _fields(b):
    """Yield (field number, wire type, value) of one message; length-delimited values are memoryviews."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = bytes(b[i:i + 8]); i += 8
        elif w == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]; i += ln
        elif w == 5:
            v = bytes(b[i:i + 4]); i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {w}")
        yield f, w, v


def _sint(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(w, v):
    if w == 0:
        return [_sint(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_sint(x))
    return out


def _tensor(b):
    dims, dtype, name, raw, f32, i32, i64, f64 = [], 1, "", None, [], [], [], []
    for f, w, v in _fields(b):
        if f == 1:
            dims += _packed_ints(w, v)
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            f32 += list(np.frombuffer(bytes(v), dtype="<f4")) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 5:
            i32 += _packed_ints(w, v)
        elif f == 7:
            i64 += _packed_ints(w, v)
        elif f == 10:
            f64 += list(np.frombuffer(bytes(v), dtype="<f8")) if w == 2 else [struct.unpack("<d", v)[0]]
        elif f in (13, 14) and (f == 13 or v != 0):
            raise ValueError("external tensor data is not supported")
    np_dt = _DT[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dt).newbyteorder("<")).astype(np_dt)
    else:
        arr = np.asarray({1: f32, 11: f64, 7: i64}.get(dtype, i32), dtype=np_dt)
    return name, arr.reshape(dims)


def _attribute(b):
    name, val, floats, ints, strings = "", None, [], [], []
    for f, w, v in _fields(b):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _sint(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(np.frombuffer(bytes(v), dtype="<f4")) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_ints(w, v)
        elif f == 9:
            strings.append(bytes(v))
    if val is None:
        val = ints or floats or strings or []
    return name, val


def _node(b):
    n = dict(input=[], output=[], name="", op_type="", attrs={})
    for f, w, v in _fields(b):
        if f == 1:
            n["input"].append(bytes(v).decode())
        elif f == 2:
            n["output"].append(bytes(v).decode())
        elif f == 3:
            n["name"] = bytes(v).decode()
        elif f == 4:
            n["op_type"] = bytes(v).decode()
        elif f == 5:
            k, a = _attribute(v)
            n["attrs"][k] = a
    return n


def _value_info(b):
    name, shape = "", None
    for f, w, v in _fields(b):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            for f2, _, v2 in _fields(v):
                if f2 == 1:                                   # TypeProto.tensor_type
                    for f3, _, v3 in _fields(v2):
                        if f3 == 2:                           # TensorShapeProto
                            shape = []
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:
                                    d = None
                                    for f5, w5, v5 in _fields(v4):
                                        d = v5 if f5 == 1 else bytes(v5).decode()
                                    shape.append(d)
    return name, shape


def read_model(path):
    """-> dict(producer, opset, nodes, initializers{name: ndarray}, inputs[(name, shape)], outputs[(name, shape)]); graph inputs that are
    initializers are dropped from `inputs`."""
    with open(path, "rb") as fh:
        b = memoryview(fh.read())
    m = dict(producer="", opset=None, nodes=[], initializers={}, inputs=[], outputs=[])
    for f, w, v in _fields(b):
        if f == 2:
            m["producer"] = bytes(v).decode()
        elif f == 8:
            dom, ver = "", None
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                m["opset"] = ver
        elif f == 7:
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    m["nodes"].append(_node(v2))
                elif f2 == 5:
                    k, a = _tensor(v2)
                    m["initializers"][k] = a
                elif f2 == 11:
                    m["inputs"].append(_value_info(v2))
                elif f2 == 12:
                    m["outputs"].append(_value_info(v2))
    m["inputs"] = [(n, s) for n, s in m["inputs"] if n not in m["initializers"]]
    return m


def _conv1d(x, w, b, strides, pads, dilations, group):
    if group != 1 or any(p != 0 for p in pads) or any(d != 1 for d in dilations):
        raise NotImplementedError("Conv: only group=1, no padding, no dilation")
    s, k = strides[0], w.shape[2]
    L = (x.shape[2] - k) // s + 1
    win = np.stack([x[:, :, i * s:i * s + k] for i in range(L)], axis=1)       # [B, L, C, k]
    y = np.einsum("blck,ock->bol", win, w, optimize=True)
    return y + b[None, :, None] if b is not None else y


def run(model, feeds):
    """Evaluate the graph on numpy inputs {name: ndarray}; returns the graph outputs as a list."""
    env = dict(model["initializers"])
    for name, _ in model["inputs"]:
        env[name] = np.asarray(feeds[name])
    for n in model["nodes"]:
        op, a = n["op_type"], n["attrs"]
        x = [env[i] if i else None for i in n["input"]]
        if op == "Gemm":
            A = x[0].T if a.get("transA", 0) else x[0]
            B = x[1].T if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ B)
            if len(x) > 2 and x[2] is not None:
                y = y + a.get("beta", 1.0) * x[2]
        elif op == "MatMul":
            y = x[0] @ x[1]
        elif op == "Add":
            y = x[0] + x[1]
        elif op == "Mul":
            y = x[0] * x[1]
        elif op == "Relu":
            y = np.maximum(x[0], 0)
        elif op == "Elu":
            al = a.get("alpha", 1.0)
            y = np.where(x[0] > 0, x[0], al * np.expm1(np.minimum(x[0], 0)))
        elif op == "Tanh":
            y = np.tanh(x[0])
        elif op == "Sigmoid":
            y = 1.0 / (1.0 + np.exp(-x[0]))
        elif op == "Concat":
            y = np.concatenate(x, axis=a["axis"])
        elif op == "Conv":
            y = _conv1d(x[0], x[1], x[2] if len(x) > 2 else None, a.get("strides", [1]), a.get("pads", [0, 0]), a.get("dilations", [1]), a.get("group", 1))
        elif op == "Reshape":
            shp = [int(x[0].shape[i]) if d == 0 else int(d) for i, d in enumerate(x[1])]
            y = x[0].reshape(shp)
        elif op == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax], dtype=np.int64)), -1)
        elif op == "Transpose":
            y = np.transpose(x[0], a.get("perm"))
        elif op == "Constant":
            y = np.asarray(a["value"])
        elif op == "Identity":
            y = x[0]
        elif op == "Shape":
            y = np.asarray(x[0].shape, dtype=np.int64)
        elif op == "Gather":
            y = np.take(x[0], x[1], axis=a.get("axis", 0))
        elif op == "Unsqueeze":
            y = x[0]
            for ax in sorted(int(v) for v in (x[1] if len(x) > 1 else a["axes"])):
                y = np.expand_dims(y, ax)
        elif op == "Squeeze":
            axes = x[1] if len(x) > 1 and x[1] is not None else a.get("axes")
            y = np.squeeze(x[0], axis=None if axes is None else tuple(int(v) for v in axes))
        elif op == "Slice":
            starts, ends = x[1], x[2]
            axes = x[3] if len(x) > 3 and x[3] is not None else np.arange(len(starts))
            steps = x[4] if len(x) > 4 and x[4] is not None else np.ones(len(starts), dtype=np.int64)
            sl = [slice(None)] * x[0].ndim
            for s, e, ax, st in zip(starts, ends, axes, steps):
                sl[int(ax)] = slice(int(s), int(min(e, np.iinfo(np.int64).max)), int(st))
            y = x[0][tuple(sl)]
        else:
            raise NotImplementedError(f"ONNX operator {op}")
        env[n["output"][0]] = y
    return [env[name] for name, _ in model["outputs"]]
