"""Skeleton tables of the robot (parents, offsets, local rotations, joint axes, extended bodies).

Host-side, load-time only.  Same information the reference extracts in
Humanoid_Batch.__init__/from_mjcf (reference: humanoidverse/utils/motion_lib/torch_humanoid_batch.py:44-165):
DFS order of <body> elements, `pos`/`quat` attributes, the hinge axes of all joints after the free
joint, and the `extend_config` bodies appended at the end.  Tables can also be loaded from a small
JSON file (used where the MJCF is not available).
"""
from __future__ import annotations

import json
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

from . import _lib


@dataclass
class Skeleton:
    body_names: list
    body_names_ext: list
    parents: np.ndarray          # [Bx] int32, -1 for the root
    offsets: np.ndarray          # [Bx,3]
    local_rot_wxyz: np.ndarray   # [Bx,4]
    dof_axis: np.ndarray         # [D,3]
    depth: np.ndarray = field(default=None)

    def __post_init__(self):
        self.parents = np.asarray(self.parents, dtype=np.int32)
        self.offsets = np.asarray(self.offsets, dtype=np.float32)
        self.local_rot_wxyz = np.asarray(self.local_rot_wxyz, dtype=np.float32)
        self.dof_axis = np.asarray(self.dof_axis, dtype=np.float32)
        depth = np.zeros(len(self.parents), dtype=np.int32)
        for i, p in enumerate(self.parents):
            depth[i] = 0 if p < 0 else depth[p] + 1
        self.depth = depth

    @property
    def num_bodies(self):
        return len(self.body_names)

    @property
    def num_bodies_ext(self):
        return len(self.body_names_ext)

    @property
    def num_dof(self):
        return self.dof_axis.shape[0]

    # ------------------------------------------------------------------
    @classmethod
    def from_mjcf(cls, path, extend_config=()):
        root = ET.parse(path).getroot()
        world = root.find("worldbody")
        if world is None or world.find("body") is None:
            raise ValueError(f"{path}: no <worldbody>/<body>")
        names, parents, offs, rots = [], [], [], []
        stack = [(world.find("body"), -1)]
        # explicit-stack DFS, children visited in document order
        while stack:
            node, parent = stack.pop()
            idx = len(names)
            names.append(node.attrib.get("name"))
            parents.append(parent)
            offs.append([float(x) for x in node.attrib.get("pos", "0 0 0").split()])
            rots.append([float(x) for x in node.attrib.get("quat", "1 0 0 0").split()])
            for child in reversed(node.findall("body")):
                stack.append((child, idx))
        joints = world.findall(".//joint")
        first = joints[0].attrib
        if first.get("type") == "free":
            hinge = joints[1:]
        elif "type" not in first:
            hinge = joints
        else:
            hinge = joints[6:]
        axes = [[float(a) for a in j.attrib["axis"].split()] for j in hinge]
        ext_names = list(names)
        for e in extend_config:
            parents.append(names.index(e["parent_name"]))
            offs.append(list(e["pos"]))
            rots.append(list(e["rot"]))
            ext_names.append(e["joint_name"])
        return cls(names, ext_names, parents, offs, rots, axes)

    @classmethod
    def from_json(cls, path):
        d = json.load(open(path))
        return cls(d["body_names"], d["body_names_ext"], d["parents"], d["offsets"], d["local_rot_wxyz"], d["dof_axis"])

    def to_json(self, path):
        json.dump(dict(body_names=self.body_names, body_names_ext=self.body_names_ext, parents=self.parents.tolist(),
                       offsets=self.offsets.tolist(), local_rot_wxyz=self.local_rot_wxyz.tolist(), dof_axis=self.dof_axis.tolist()),
                  open(path, "w"), indent=1)

    @classmethod
    def from_motion_config(cls, mcfg):
        """mcfg = config.robot.motion (asset.assetRoot/assetFileName + extend_config)."""
        import os

        path = os.path.join(str(mcfg.asset.assetRoot), str(mcfg.asset.assetFileName))
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(_lib.ROOT, path)
        if path.endswith(".json"):
            return cls.from_json(path)
        return cls.from_mjcf(path, [dict(e) for e in mcfg.get("extend_config", [])])

    # ------------------------------------------------------------------
    def to_c(self):
        K = _lib.K
        if self.num_bodies_ext > K["PBHC_MAX_BODIES"] or self.num_dof > K["PBHC_MAX_DOF"]:
            raise _lib.PbhcError("skeleton larger than the compiled maxima")
        if self.num_dof != self.num_bodies - 1:
            raise _lib.PbhcError("expected one hinge per non-root body")
        s = _lib.PbhcSkeleton()
        s.num_bodies, s.num_bodies_ext, s.num_dof = self.num_bodies, self.num_bodies_ext, self.num_dof
        s.max_depth = int(self.depth[: self.num_bodies].max())
        for i in range(self.num_bodies_ext):
            s.parent[i] = int(self.parents[i])
            s.depth[i] = int(self.depth[i])
            for k in range(3):
                s.offset[i][k] = float(self.offsets[i, k])
            for k in range(4):
                s.local_rot_wxyz[i][k] = float(self.local_rot_wxyz[i, k])
        for d in range(self.num_dof):
            for k in range(3):
                s.dof_axis[d][k] = float(self.dof_axis[d, k])
        for b in range(self.num_bodies_ext):
            node = b if b < self.num_bodies else int(self.parents[b])
            path = []
            while node > 0:
                path.append(node)
                node = int(self.parents[node])
            path.reverse()
            if len(path) > K["PBHC_MAX_DEPTH"]:
                raise _lib.PbhcError("kinematic chain deeper than PBHC_MAX_DEPTH")
            s.chain_len[b] = len(path)
            for k, a in enumerate(path):
                s.chain[b][k] = a
        return s
