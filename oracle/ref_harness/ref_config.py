"""An interpolation resolver for the reference's saved `config.yaml` files, written INDEPENDENTLY of the product's
(`pbhc_amd/utils/config.py`) — test infrastructure, build container only.

Why it exists: the reference run that generates the env goldens used to get its config tree from the product's own resolver, so a wrong
`${eval:...}` resolution would have been invisible (same mistake on both sides).  `gen_config_golden.py` builds the reference env from THIS
resolver instead and records what the reference derived from the tree; `tests/test_config_derived.py` then holds the product's resolver
and `envs/env_config.py` to those numbers.

The algorithm differs from the product's on purpose: innermost-first textual substitution (a regex finds an interpolation with no nested
`${` inside, resolves it, splices the result back as text — or, when it is the whole value, keeps the referenced OBJECT so that
`env.config.robot` aliases `robot` as OmegaConf's lazy interpolation does, helpers.py:56,77) instead of the product's recursive-descent
scanner; resolver arguments are evaluated as Python literals.  Resolvers: the reference's own list, utils/config_utils.py:5-13, + Hydra's
`now`.
"""
import ast
import math
import re

import yaml

_INNER = re.compile(r"\$\{([^${}]*)\}")          # an interpolation that contains no other one
_CALL = re.compile(r"^\s*([A-Za-z_]\w*)\s*:(.*)$", re.S)
_RESOLVERS = ("eval", "if", "eq", "sqrt", "sum", "ceil", "int", "len", "sum_list", "now")


class Node(dict):
    """attribute-access dict (what the reference's code expects of a DictConfig: attribute get / set, `in`, `.get`, iteration)"""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _nodes(x):
    if isinstance(x, dict):
        return Node({k: _nodes(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_nodes(v) for v in x]
    return x


def _literal(text):
    text = text.strip()
    if len(text) >= 2 and text[0] == text[-1] and text[0] in "'\"":
        return text[1:-1]
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _split_top(argtext):
    """split on commas that are not inside brackets / quotes"""
    out, depth, cur, quote = [], 0, "", None
    for ch in argtext:
        if quote:
            cur += ch
            if ch == quote:
                quote = None
            continue
        if ch in "'\"":
            quote = ch
        elif ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return out


class Resolver:
    def __init__(self, root, now="golden"):
        self.root, self.now, self.busy = root, now, set()

    def path(self, dotted):
        cur = self.root
        walked = []
        for part in dotted.strip().split("."):
            walked.append(part)
            holder = cur
            key = int(part) if isinstance(cur, list) else part
            cur = holder[key]
            if isinstance(cur, str) and "${" in cur:
                where = ".".join(walked)
                if where in self.busy:
                    raise ValueError("interpolation cycle at " + where)
                self.busy.add(where)
                cur = self.value(cur)
                self.busy.discard(where)
                holder[key] = cur
        return cur

    def call(self, name, argtext):
        if name == "now":
            return self.now
        if name == "eval":
            return eval(_literal(argtext) if argtext.strip()[:1] in "'\"" else argtext, {"__builtins__": {}, "math": math, "len": len, "int": int, "float": float,
                                                                                   "min": min, "max": max, "abs": abs, "sum": sum, "round": round})
        args = [_literal(a) for a in _split_top(argtext)]
        if name == "if":
            return args[1] if args[0] else args[2]
        if name == "eq":
            return str(args[0]).lower() == str(args[1]).lower()
        if name == "sqrt":
            return math.sqrt(float(args[0]))
        if name in ("sum", "sum_list"):
            return sum(args[0])
        if name == "ceil":
            return math.ceil(args[0])
        if name == "int":
            return int(args[0])
        if name == "len":
            return len(args[0])
        raise KeyError(name)

    def one(self, inner):
        m = _CALL.match(inner)
        if m and m.group(1) in _RESOLVERS:
            return self.call(m.group(1), m.group(2))
        return self.path(inner)

    def value(self, text):
        while True:
            m = _INNER.search(text)
            if not m:
                return text
            got = self.one(m.group(1))
            if m.start() == 0 and m.end() == len(text):
                return got                                   # the whole value: keep the object (aliasing) / the number's type
            text = text[:m.start()] + (got if isinstance(got, str) else repr(got)) + text[m.end():]

    def tree(self):
        seen = set()

        def walk(node):
            if id(node) in seen:
                return
            seen.add(id(node))
            for k in (list(node.keys()) if isinstance(node, dict) else range(len(node))):
                v = node[k]
                if isinstance(v, str) and "${" in v:
                    v = node[k] = self.value(v)
                if isinstance(v, (dict, list)):
                    walk(v)

        walk(self.root)
        return self.root


def load(path, overrides=None, now="golden"):
    with open(path) as f:
        root = _nodes(yaml.safe_load(f))
    for dotted, v in (overrides or {}).items():
        cur = root
        parts = dotted.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, Node())
        cur[parts[-1]] = _nodes(v)
    return Resolver(root, now).tree()
