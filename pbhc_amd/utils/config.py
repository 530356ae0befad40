"""YAML config tree loader with the OmegaConf interpolation subset the reference's saved
``config.yaml`` files use.

The reference composes its config with Hydra/OmegaConf (reference:
humanoidverse/train_agent.py:18-104) and registers the resolvers ``eval, if, eq, sqrt, sum,
ceil, int, len, sum_list`` (reference: humanoidverse/utils/config_utils.py:5-13).  Neither
library is available here, so this module resolves ``${path}``, ``${resolver:args}`` and the
Hydra ``${now:fmt}`` resolver itself.  A value that is exactly one ``${path}`` reference is
resolved to the *same node object* (aliasing), mirroring OmegaConf's lazy interpolation:
the reference mutates ``config.robot`` through ``config.env.config.robot``
(reference: humanoidverse/utils/helpers.py:56,77).
"""
from __future__ import annotations

import copy
import math
import re
import time
from typing import Any

import yaml


class AttrDict(dict):
    """dict with attribute access (``cfg.a.b``), the container type of the config tree."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __deepcopy__(self, memo):
        out = AttrDict()
        memo[id(self)] = out
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out


_INTERP = re.compile(r"\$\{")


def _wrap(node):
    if isinstance(node, dict):
        return AttrDict({k: _wrap(v) for k, v in node.items()})
    if isinstance(node, list):
        return [_wrap(v) for v in node]
    return node


def _find_close(s: str, start: int) -> int:
    """index of the '}' closing the '${' that starts at ``start``."""
    depth = 0
    i = start
    while i < len(s):
        if s.startswith("${", i):
            depth += 1
            i += 2
            continue
        if s[i] == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ValueError(f"unbalanced interpolation in {s!r}")


class _Resolver:
    def __init__(self, root: AttrDict, now: str | None):
        self.root = root
        self._t = time.localtime()
        self._now = now
        self._active: set = set()

    # -- lookups --------------------------------------------------------------------
    def _lookup(self, path: str):
        cur: Any = self.root
        keys = path.split(".")
        for i, k in enumerate(keys):
            if isinstance(cur, list):
                k = int(k)
                v = cur[k]
            else:
                if k not in cur:
                    raise KeyError(f"interpolation key {path!r} not found")
                v = cur[k]
            if isinstance(v, str) and _INTERP.search(v):
                v = self._resolve_str(v, ".".join(keys[: i + 1]))
                cur[k] = v
            cur = v
        return cur

    def _call(self, name: str, argstr: str):
        if name == "now":
            return self._now if self._now is not None else time.strftime(argstr, self._t)
        if name == "eval":
            expr = self._resolve_str(argstr, None)
            expr = str(expr).strip()
            if len(expr) >= 2 and expr[0] == expr[-1] and expr[0] in "'\"":
                expr = expr[1:-1]
            return eval(expr, {"__builtins__": {}}, {"math": math, "min": min, "max": max, "abs": abs, "int": int, "float": float, "len": len, "sum": sum, "round": round})
        args = [self._resolve_str(a.strip(), None) for a in _split_args(argstr)]
        if name == "len":
            return len(args[0])
        if name in ("sum", "sum_list"):
            return sum(args[0])
        if name == "int":
            return int(args[0])
        if name == "ceil":
            return math.ceil(args[0])
        if name == "sqrt":
            return math.sqrt(float(args[0]))
        if name == "eq":
            return str(args[0]).lower() == str(args[1]).lower()
        if name == "if":
            return args[1] if args[0] else args[2]
        raise KeyError(f"unknown resolver {name!r}")

    def _resolve_one(self, inner: str):
        m = re.match(r"^([A-Za-z_][A-Za-z_0-9]*):(.*)$", inner, re.S)
        if m and not re.match(r"^[A-Za-z_0-9.]+$", inner):
            return self._call(m.group(1), m.group(2))
        if m and m.group(1) in ("now", "eval", "len", "sum", "sum_list", "int", "ceil", "sqrt", "eq", "if"):
            return self._call(m.group(1), m.group(2))
        inner = self._resolve_str(inner, None) if _INTERP.search(inner) else inner
        return self._lookup(str(inner))

    def _resolve_str(self, s: str, where: str | None):
        if where is not None:
            if where in self._active:
                raise ValueError(f"interpolation cycle at {where}")
            self._active.add(where)
        try:
            s_strip = s
            if s_strip.startswith("${") and _find_close(s_strip, 0) == len(s_strip) - 1:
                return self._resolve_one(s_strip[2:-1])
            out = []
            i = 0
            while i < len(s):
                j = s.find("${", i)
                if j < 0:
                    out.append(s[i:])
                    break
                out.append(s[i:j])
                e = _find_close(s, j)
                out.append(str(self._resolve_one(s[j + 2 : e])))
                i = e + 1
            return "".join(out)
        finally:
            if where is not None:
                self._active.discard(where)

    # -- whole tree -----------------------------------------------------------------
    def resolve_tree(self):
        seen: set = set()

        def walk(node, path):
            if id(node) in seen:
                return
            seen.add(id(node))
            it = node.items() if isinstance(node, dict) else enumerate(node)
            for k, v in list(it):
                p = f"{path}.{k}" if path else str(k)
                if isinstance(v, str) and _INTERP.search(v):
                    v = self._resolve_str(v, p)
                    node[k] = v
                if isinstance(v, (dict, list)):
                    walk(v, p)

        walk(self.root, "")
        return self.root


def _split_args(s: str):
    out, depth, cur = [], 0, []
    i = 0
    while i < len(s):
        if s.startswith("${", i):
            depth += 1
            cur.append("${")
            i += 2
            continue
        c = s[i]
        if c == "}" and depth:
            depth -= 1
        if c == "," and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
        i += 1
    out.append("".join(cur))
    return out


def load_unresolved(path: str) -> AttrDict:
    """The YAML tree as written (interpolations kept as text) — what the reference saves
    next to checkpoints (reference: humanoidverse/train_agent.py:103-104)."""
    with open(path, "r") as f:
        return _wrap(yaml.safe_load(f))


def resolve(cfg: AttrDict, now: str | None = None) -> AttrDict:
    return _Resolver(cfg, now).resolve_tree()


def load_config(path: str, overrides: dict | None = None, now: str | None = None) -> AttrDict:
    """Load + apply dotted-key overrides (``{"num_envs": 64, "robot.motion.motion_file": ...}``)
    + resolve interpolations."""
    cfg = load_unresolved(path)
    for k, v in (overrides or {}).items():
        set_by_path(cfg, k, v)
    return resolve(cfg, now)


def set_by_path(cfg, dotted: str, value):
    keys = dotted.split(".")
    cur = cfg
    for k in keys[:-1]:
        if k not in cur:
            cur[k] = AttrDict()
        cur = cur[k]
    cur[keys[-1]] = _wrap(value)


def save_unresolved(cfg_unresolved: AttrDict, path: str):
    def plain(n):
        if isinstance(n, dict):
            return {k: plain(v) for k, v in n.items()}
        if isinstance(n, list):
            return [plain(v) for v in n]
        return n

    with open(path, "w") as f:
        yaml.safe_dump(plain(cfg_unresolved), f, sort_keys=False)
