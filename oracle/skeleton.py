"""MJCF body tree -> constant skeleton tables.  Test infrastructure.

Restates Humanoid_Batch.__init__/from_mjcf (reference:
humanoidverse/utils/motion_lib/torch_humanoid_batch.py:44-102,104-165): DFS over <body>
elements gives node order, parent index, local translation, local rotation (wxyz); the joint
axes come from every <joint> under <worldbody> in document order (skipping a leading free
joint); the `extend_config` bodies (hands, head) are appended with fixed parents.
"""
import xml.etree.ElementTree as ET
import numpy as np


def parse_mjcf(path, extend_config=()):
    root = ET.parse(path).getroot()
    world = root.find("worldbody")
    body_root = world.find("body")
    names, parents, offs, rots = [], [], [], []

    def add(node, parent):
        idx = len(names)
        names.append(node.attrib.get("name"))
        parents.append(parent)
        offs.append(np.array(node.attrib.get("pos", "0 0 0").split(), dtype=np.float64))
        rots.append(np.array(node.attrib.get("quat", "1 0 0 0").split(), dtype=np.float64))
        for child in node.findall("body"):
            add(child, idx)

    add(body_root, -1)
    joints = world.findall(".//joint")
    # reference :73-84: a leading free joint (explicit type, or an untyped first joint) is skipped
    if joints[0].attrib.get("type") == "free":
        hinge = joints[1:]
    elif "type" not in joints[0].attrib:
        hinge = joints
    else:
        hinge = joints[6:]
    axes = np.array([[int(float(a)) for a in j.attrib["axis"].split()] for j in hinge], dtype=np.float32)
    num_bodies = len(names)
    names_ext = list(names)
    for e in extend_config:
        parents.append(names.index(e["parent_name"]))
        offs.append(np.array(e["pos"], dtype=np.float64))
        rots.append(np.array(e["rot"], dtype=np.float64))
        names_ext.append(e["joint_name"])
    return dict(
        body_names=names,
        body_names_ext=names_ext,
        num_bodies=num_bodies,
        parents=np.array(parents, dtype=np.int32),
        offsets=np.array(offs, dtype=np.float32),
        local_rot_wxyz=np.array(rots, dtype=np.float32),
        dof_axis=axes,
    )
