"""`Skeleton.from_mjcf` RUN, not its stored output (torch_humanoid_batch.py:104-165 is what it restates): a small
hand-written MJCF with hand-derived tables runs everywhere; the two G1 files of the reference are parsed where /root/reference exists
(this container) and compared with the tables the reference's own `Humanoid_Batch` produced (tests/golden/skeleton_fk_g1_*.npz, written by
oracle/ref_harness/gen_golden.py from the unmodified reference) and with the JSON the fixtures ship."""
import os

import numpy as np
import pytest

from pbhc_amd import _lib
from pbhc_amd.skeleton import Skeleton
from tests.helpers import GOLDEN, fixture_config

REF_G1 = "/root/reference/description/robots/g1"


def test_from_mjcf_on_a_hand_written_file():
    ext = [dict(joint_name="l_hand", parent_name="l_arm", pos=[0.2, 0.0, 0.0], rot=[1.0, 0.0, 0.0, 0.0]),
           dict(joint_name="crown", parent_name="head", pos=[0.0, 0.0, 0.1], rot=[0.7071068, 0.0, 0.0, 0.7071068])]
    sk = Skeleton.from_mjcf(os.path.join(GOLDEN, "mini_biped.xml"), ext)
    # depth-first, children in document order (Humanoid_Batch._add_xml_node recursion, torch_humanoid_batch.py:128-145)
    assert sk.body_names == ["pelvis", "l_hip", "l_knee", "l_foot", "torso", "l_arm", "head", "r_hip"]
    assert sk.body_names_ext == sk.body_names + ["l_hand", "crown"]
    assert sk.parents.tolist() == [-1, 0, 1, 2, 0, 4, 4, 0, 5, 6]
    assert sk.depth.tolist() == [0, 1, 2, 3, 1, 2, 2, 1, 3, 3]
    want_off = [[0, 0, 0.8], [0, 0.1, -0.05], [0, 0, -0.3], [0, 0, 0], [0, 0, 0], [0.02, 0.15, 0.25], [0, 0, 0.4], [0, -0.1, -0.05], [0.2, 0, 0], [0, 0, 0.1]]
    assert np.allclose(sk.offsets, np.array(want_off, dtype=np.float32))
    want_rot = [[1, 0, 0, 0], [0.9238795, 0, 0.3826834, 0], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [0.7071068, 0.7071068, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0],
                [1, 0, 0, 0], [0.7071068, 0, 0, 0.7071068]]
    assert np.allclose(sk.local_rot_wxyz, np.array(want_rot, dtype=np.float32))
    # one hinge per non-root body, in document order = DFS order, the free joint dropped (:147-160)
    assert sk.num_dof == 7 == sk.num_bodies - 1
    assert np.allclose(sk.dof_axis, np.array([[0, 1, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1], [0.6, 0, 0.8], [0, 0, 1], [0, 1, 0]], dtype=np.float32))
    # what the kernels walk: root -> body chains (an extended body walks to its parent)
    c = sk.to_c()
    assert (c.num_bodies, c.num_bodies_ext, c.num_dof, c.max_depth) == (8, 10, 7, 3)
    chains = [[c.chain[b][k] for k in range(c.chain_len[b])] for b in range(10)]
    assert chains == [[], [1], [1, 2], [1, 2, 3], [4], [4, 5], [4, 6], [7], [4, 5], [4, 6]]
    # JSON round trip = the same object
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        sk.to_json(os.path.join(d, "s.json"))
        sk2 = Skeleton.from_json(os.path.join(d, "s.json"))
    for a in ("body_names", "body_names_ext"):
        assert getattr(sk, a) == getattr(sk2, a)
    for a in ("parents", "offsets", "local_rot_wxyz", "dof_axis", "depth"):
        assert np.array_equal(getattr(sk, a), getattr(sk2, a)), a


def test_from_mjcf_refuses_a_file_without_bodies(tmp_path):
    p = tmp_path / "empty.xml"
    p.write_text("<mujoco><worldbody/></mujoco>")
    with pytest.raises(ValueError):
        Skeleton.from_mjcf(str(p))


def test_skeleton_larger_than_the_kernel_maxima_is_refused():
    n = _lib.K["PBHC_MAX_DEPTH"] + 2
    sk = Skeleton([f"b{i}" for i in range(n)], [f"b{i}" for i in range(n)], [-1] + list(range(n - 1)), np.zeros((n, 3)), np.tile([1.0, 0, 0, 0], (n, 1)), np.tile([0, 0, 1.0], (n - 1, 1)))
    with pytest.raises(_lib.PbhcError):
        sk.to_c()


@pytest.mark.skipif(not os.path.isdir(REF_G1), reason="the reference's MJCF files exist in the build container only")
@pytest.mark.parametrize("xml,cfgname,robot,json_name", [
    ("g1_23dof_lock_wrist_fitmotionONLY.xml", "v1_g1_23dof_walk.yaml", "g1_23dof", "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"),
    ("g1_29dof_rev_1_0.xml", "v2_g1_29dof_teacher.yaml", "g1_29dof", "skeleton_g1_29dof_rev_1_0.json")])
def test_from_mjcf_on_the_reference_robots(xml, cfgname, robot, json_name):
    cfg = fixture_config(cfgname, 4)
    ext = [dict(e) for e in cfg.robot.motion.get("extend_config", [])]
    sk = Skeleton.from_mjcf(os.path.join(REF_G1, xml), ext)
    # ... against the tables the reference's Humanoid_Batch holds for the same file and extend_config
    g = np.load(os.path.join(GOLDEN, f"skeleton_fk_{robot}.npz"))
    assert sk.body_names_ext == [str(x) for x in g["body_names"]]
    assert sk.num_bodies == int(g["num_bodies"])
    assert np.array_equal(sk.parents, g["parents"])
    assert np.allclose(sk.offsets, g["offsets"], atol=1e-7) and np.allclose(sk.local_rot_wxyz, g["local_rot_wxyz"], atol=1e-7)
    assert np.allclose(sk.dof_axis, g["dof_axis"].astype(np.float32), atol=1e-7)
    # ... and against the JSON the fixture configs load (written by this parser at fixture time: a later edit of from_mjcf fails here)
    js = Skeleton.from_json(os.path.join(GOLDEN, json_name))
    assert js.body_names_ext == sk.body_names_ext and js.body_names == sk.body_names
    for a in ("parents", "offsets", "local_rot_wxyz", "dof_axis"):
        assert np.array_equal(getattr(js, a), getattr(sk, a)), a
