"""Static (non-executing) reader for joblib-pickled motion files.

The reference stores reference motions as joblib pickles (dict name -> dict of
numpy arrays + ``fps``), written by its retargeting tools and read back with
``joblib.load`` (reference: humanoidverse/utils/motion_lib/motion_lib_base.py:63-66,
humanoidverse/utils/helpers.py:102-106).  Unpickling executes whatever the file
says, so this reader does NOT unpickle: it walks the pickle opcode stream with
the opcode table of ``pickletools`` as plain data, never imports a module named
in the file, never calls a callable from the file, and only materialises

* str / int / float / bool / None / tuple / list / dict
* numpy arrays stored by joblib's ``NumpyArrayWrapper`` (raw C-order bytes that
  follow the wrapper's BUILD opcode; dtype limited to plain numeric dtypes)

Anything else (an unknown global, a REDUCE of something that is not
``numpy.dtype``, an object dtype, a compressed file) raises ``UnsafePickleError``.
"""
from __future__ import annotations

import io
import pickletools
from typing import Any, BinaryIO

import numpy as np


class UnsafePickleError(RuntimeError):
    pass


class _Global:
    __slots__ = ("module", "name")

    def __init__(self, module: str, name: str):
        self.module, self.name = module, name

    def __repr__(self):
        return f"<global {self.module}.{self.name}>"


class _Obj:
    """Symbolic result of NEWOBJ / REDUCE: never instantiated."""

    __slots__ = ("cls", "args", "state")

    def __init__(self, cls, args):
        self.cls, self.args, self.state = cls, args, None


_MARK = object()
_NUMERIC_KINDS = set("fiub")
_ALLOWED_DTYPE_STR = {f"{k}{n}" for k in "fiu" for n in (1, 2, 4, 8)} | {"b1", "f2"}


def _dtype_from_symbolic(o) -> np.dtype:
    if not (isinstance(o, _Obj) and isinstance(o.cls, _Global) and (o.cls.module, o.cls.name) == ("numpy", "dtype")):
        raise UnsafePickleError(f"array dtype is not a plain numpy.dtype reduce: {o!r}")
    code = o.args[0]
    if code not in _ALLOWED_DTYPE_STR:
        raise UnsafePickleError(f"dtype {code!r} not allowed")
    dt = np.dtype(code)
    if o.state is not None:
        order = o.state[1]
        if order in ("<", ">"):
            dt = dt.newbyteorder(order)
    if dt.kind not in _NUMERIC_KINDS:
        raise UnsafePickleError(f"dtype kind {dt.kind!r} not allowed")
    return dt


def _read_wrapped_array(f: BinaryIO, state: dict) -> np.ndarray:
    sub = state.get("subclass")
    if not (isinstance(sub, _Global) and (sub.module, sub.name) == ("numpy", "ndarray")):
        raise UnsafePickleError(f"unsupported array subclass {sub!r}")
    dt = _dtype_from_symbolic(state["dtype"])
    shape = tuple(int(s) for s in state["shape"])
    order = state.get("order", "C")
    align = state.get("numpy_array_alignment_bytes")
    if align is not None:
        pad = f.read(1)[0]
        if pad:
            f.read(pad)
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    nbytes = count * dt.itemsize
    raw = f.read(nbytes)
    if len(raw) != nbytes:
        raise UnsafePickleError("truncated array payload")
    arr = np.frombuffer(raw, dtype=dt, count=count).copy()
    arr = arr.reshape(shape, order="F" if order == "F" else "C")
    return arr.astype(dt.newbyteorder("="), copy=False)


def load(path_or_file) -> Any:
    """Parse a joblib pickle without executing it; returns plain python/numpy data."""
    if hasattr(path_or_file, "read"):
        return _load(path_or_file)
    with open(path_or_file, "rb") as fh:
        data = fh.read()
    if data[:2] in (b"\x78\x9c", b"\x78\x01", b"\x78\xda") or data[:3] == b"\x1f\x8b\x08" or data[:4] == b"ZF0x":
        raise UnsafePickleError("compressed joblib files are not supported by the static reader")
    return _load(io.BytesIO(data))


def _load(f: BinaryIO) -> Any:
    stack: list = []
    memo: dict = {}
    ops = pickletools.code2op
    while True:
        c = f.read(1)
        if not c:
            raise UnsafePickleError("EOF before STOP")
        op = ops.get(c.decode("latin-1"))
        if op is None:
            raise UnsafePickleError(f"unknown opcode {c!r} at {f.tell() - 1}")
        arg = op.arg.reader(f) if op.arg is not None else None
        name = op.name
        if name in ("PROTO", "FRAME"):
            continue
        if name == "STOP":
            return stack.pop()
        if name == "MARK":
            stack.append(_MARK)
        elif name in ("EMPTY_DICT",):
            stack.append({})
        elif name in ("EMPTY_LIST",):
            stack.append([])
        elif name in ("EMPTY_TUPLE",):
            stack.append(())
        elif name == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif name in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif name in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif name in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE",
                      "BININT", "BININT1", "BININT2", "INT", "LONG", "LONG1", "LONG4",
                      "BINFLOAT", "FLOAT", "SHORT_BINSTRING", "BINSTRING",
                      "SHORT_BINBYTES", "BINBYTES", "BINBYTES8"):
            stack.append(arg)
        elif name == "NONE":
            stack.append(None)
        elif name == "NEWTRUE":
            stack.append(True)
        elif name == "NEWFALSE":
            stack.append(False)
        elif name == "TUPLE1":
            stack[-1:] = [(stack[-1],)]
        elif name == "TUPLE2":
            stack[-2:] = [(stack[-2], stack[-1])]
        elif name == "TUPLE3":
            stack[-3:] = [(stack[-3], stack[-2], stack[-1])]
        elif name in ("TUPLE", "LIST"):
            i = _last_mark(stack)
            items = stack[i + 1:]
            del stack[i:]
            stack.append(tuple(items) if name == "TUPLE" else list(items))
        elif name == "DICT":
            i = _last_mark(stack)
            items = stack[i + 1:]
            del stack[i:]
            stack.append({items[k]: items[k + 1] for k in range(0, len(items), 2)})
        elif name == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif name == "APPENDS":
            i = _last_mark(stack)
            items = stack[i + 1:]
            del stack[i:]
            stack[-1].extend(items)
        elif name == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            _setitem(stack[-1], k, v)
        elif name == "SETITEMS":
            i = _last_mark(stack)
            items = stack[i + 1:]
            del stack[i:]
            for k in range(0, len(items), 2):
                _setitem(stack[-1], items[k], items[k + 1])
        elif name == "STACK_GLOBAL":
            n_ = stack.pop()
            m_ = stack.pop()
            stack.append(_Global(m_, n_))
        elif name == "GLOBAL":
            m_, n_ = arg.split(" ")
            stack.append(_Global(m_, n_))
        elif name == "NEWOBJ":
            args = stack.pop()
            cls = stack.pop()
            stack.append(_Obj(cls, args))
        elif name == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            stack.append(_reduce(fn, args))
        elif name == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, _Obj):
                raise UnsafePickleError(f"BUILD on {type(tgt).__name__}")
            tgt.state = state
            if isinstance(tgt.cls, _Global) and (tgt.cls.module, tgt.cls.name) == ("joblib.numpy_pickle", "NumpyArrayWrapper"):
                if not isinstance(state, dict):
                    raise UnsafePickleError("NumpyArrayWrapper state is not a dict")
                stack[-1] = _read_wrapped_array(f, state)
                # the array replaces the wrapper in the memo as well
                for k, v in list(memo.items()):
                    if v is tgt:
                        memo[k] = stack[-1]
        else:
            raise UnsafePickleError(f"opcode {name} not allowed in a motion file")


def _last_mark(stack) -> int:
    for i in range(len(stack) - 1, -1, -1):
        if stack[i] is _MARK:
            return i
    raise UnsafePickleError("MARK not found")


def _setitem(d, k, v):
    if not isinstance(d, dict):
        raise UnsafePickleError("SETITEM on non-dict")
    d[k] = v


def _reduce(fn, args):
    if isinstance(fn, _Global):
        key = (fn.module, fn.name)
        if key == ("numpy", "dtype"):
            return _Obj(fn, args)
        if key in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
            dt = _dtype_from_symbolic(args[0])
            return np.frombuffer(args[1], dtype=dt, count=1)[0].item()
        if key == ("collections", "OrderedDict") and args in ((), ([],)):
            return {}
    raise UnsafePickleError(f"REDUCE of {fn!r} not allowed")
