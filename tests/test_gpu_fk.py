"""The rigid-body chain of the sim-stub (what Isaac Gym's rigid-body state tensor held in the reference, isaacgym.py:574-605; the reference's
own FK of a pose is torch_humanoid_batch.py:248-252) in its two device forms — the chain WALK (`pbhc_sim_fk`, the generic step kernel for robots
of more than 32 bodies) and POINTER JUMPING (the step kernel otherwise: csrc/pbhc_env_step.h fk_jump_wave) — against each other and against the
CPU oracle on skeletons the config-driven env tests never reach: random trees of every depth class (1, 2, 3 and 4 jumping rounds), hinges on
skew axes, local rotations, extended bodies hanging on the root, on leaves and on inner bodies."""
import ctypes as C

import numpy as np
import pytest
import torch

from pbhc_amd import _lib
from pbhc_amd.skeleton import Skeleton

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _random_skeleton(rng, num_bodies, max_depth, num_ext):
    """a random tree with EXACTLY `max_depth` as its deepest chain (body b's parent is drawn among earlier bodies that are not too deep)"""
    parents, depth = [-1], [0]
    spine = list(range(0, max_depth + 1))                      # bodies 0..max_depth form one chain of the wanted depth
    for b in range(1, num_bodies):
        if b <= max_depth:
            p = b - 1
        else:
            cand = [a for a in range(b) if depth[a] < max_depth]
            p = int(rng.choice(cand))
        parents.append(p); depth.append(depth[p] + 1)
    assert max(depth) == max_depth
    q = rng.normal(size=(num_bodies, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[::3] = [1.0, 0.0, 0.0, 0.0]                                # (a third of the bodies without a local rotation, as in the G1)
    off = rng.uniform(-0.3, 0.3, size=(num_bodies, 3))
    axes = rng.normal(size=(num_bodies - 1, 3))
    axes[::2] = np.eye(3)[rng.integers(0, 3, size=len(axes[::2]))]
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)            # (MJCF hinge axes are unit vectors; the kernels normalise them once, the reference per call)
    names = [f"b{i}" for i in range(num_bodies)]
    ext_par = [0, num_bodies - 1] + [int(x) for x in rng.integers(0, num_bodies, size=max(num_ext - 2, 0))]
    ext_names = list(names)
    for k, p in enumerate(ext_par[:num_ext]):
        parents.append(p)
        off = np.vstack([off, rng.uniform(-0.2, 0.2, size=(1, 3))])
        eq = rng.normal(size=4); eq /= np.linalg.norm(eq)
        q = np.vstack([q, eq[None]])
        ext_names.append(f"e{k}")
    return Skeleton(names, ext_names, parents, off, q, axes)


def _oracle_with_ext(sk, root, qp, qv):
    """oracle.fk.sim_fk for the real bodies + the reference's extended-body formula (motion_tracking.py:619-643) on top"""
    from oracle import rotations as R
    from oracle.fk import sim_fk

    skel = dict(parents=sk.parents, offsets=sk.offsets, local_rot_wxyz=sk.local_rot_wxyz, dof_axis=sk.dof_axis, num_bodies=sk.num_bodies)
    pos, rot, vel, ang = sim_fk(skel, root, qp, qv)
    rot_chain = rot.clone()
    rot_chain[:, 0] = R.normalize(root[:, 3:7])                 # (the chain hangs on the unit root rotation; body 0 reports the frame's own)
    P, Q, V, W = [pos], [rot], [vel], [ang]
    for b in range(sk.num_bodies, sk.num_bodies_ext):
        p = int(sk.parents[b])
        off = torch.tensor(sk.offsets[b]).expand(len(root), 3)
        eq = R.wxyz_to_xyzw(torch.tensor(sk.local_rot_wxyz[b]))[None].expand(len(root), 4)
        P.append((R.quat_rotate(eq, R.quat_rotate(rot_chain[:, p], off)) + pos[:, p])[:, None])
        Q.append(R.quat_mul(rot_chain[:, p], eq)[:, None])
        V.append((vel[:, p] + torch.cross(ang[:, p], off, dim=-1))[:, None])
        W.append(ang[:, p][:, None])
    return torch.cat([torch.cat(P, 1), torch.cat(Q, 1), torch.cat(V, 1), torch.cat(W, 1)], -1)


@pytest.mark.parametrize("num_bodies,max_depth,num_ext", [(3, 1, 1), (6, 3, 2), (20, 7, 3), (24, 7, 3), (29, 12, 3), (14, 12, 2), (32, 9, 0)])
def test_pointer_jumping_equals_the_walk_and_the_oracle(num_bodies, max_depth, num_ext):
    rng = np.random.default_rng(num_bodies * 100 + max_depth)
    sk = _random_skeleton(rng, num_bodies, max_depth, num_ext)
    csk = sk.to_c()
    assert csk.max_depth == max_depth
    N, D, Bx = 37, sk.num_dof, sk.num_bodies_ext                  # (37: a partly filled last workgroup)
    g = torch.Generator().manual_seed(num_bodies)
    root = torch.zeros(N, 13)
    root[:, :3] = torch.randn(N, 3, generator=g)
    rq = torch.randn(N, 4, generator=g); root[:, 3:7] = rq / rq.norm(dim=-1, keepdim=True)
    root[::4, 3:7] *= 1.0003                                      # a replay frame's quaternion may be off unit length (the reference slerp's scale error)
    root[:, 7:] = torch.randn(N, 6, generator=g)
    qp, qv = 1.5 * torch.randn(N, D, generator=g), 4.0 * torch.randn(N, D, generator=g)
    lib = _lib.lib()
    outs = []
    d_root, d_qp, d_qv = root.to(DEV).contiguous(), qp.to(DEV).contiguous(), qv.to(DEV).contiguous()
    for method in (0, 1):
        out = torch.zeros(N, Bx, 13, device=DEV)
        _lib.check(lib.pbhc_debug_fk(C.byref(csk), d_root.data_ptr(), d_qp.data_ptr(), d_qv.data_ptr(), N, method, out.data_ptr(), _lib.current_stream()), "pbhc_debug_fk")
        torch.cuda.synchronize()
        outs.append(out.cpu())
    walk, jump = outs
    want = _oracle_with_ext(sk, root, qp, qv)
    scale = 1.0 + want.abs()
    for name, got in (("walk", walk), ("jump", jump)):
        err = ((got - want).abs() / scale)
        assert float(err.max()) < 2e-5, (name, float(err.max()), err.argmax())
    # the two device forms: the same arithmetic up to the association order (joint rates of +-4 rad/s in these draws carry the rotations' ulps into the twists)
    assert float(((walk - jump).abs() / scale).max()) < 1e-5


def test_pointer_jumping_is_refused_beyond_its_limits():
    rng = np.random.default_rng(5)
    sk = _random_skeleton(rng, 30, 5, 3)                           # 33 bodies incl. extended: more than the 32 lanes of an env
    lib = _lib.lib()
    assert lib.pbhc_debug_fk(C.byref(sk.to_c()), None, None, None, 0, 1, None, None) == _lib.K["PBHC_EINVAL"]
