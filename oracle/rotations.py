"""Quaternion / rotation algebra, xyzw unless stated (torch CPU fp32).  Test infrastructure.

Follows humanoidverse/isaac_utils/isaac_utils/rotations.py and humanoidverse/utils/torch_utils.py
of the reference; each function names the lines it restates.  Three algebraically different
"rotate a vector" formulas exist in the reference and differ in rounding — each is kept.
"""
import math
import torch


def normalize(x, eps: float = 1e-9):
    # reference: isaac_utils/maths.py:6-8
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_rotate(q, v):
    # reference: rotations.py:82-97 (== my_quat_rotate :244-253, torch_utils.py:61-68)
    qw = q[..., 3]
    qv = q[..., :3]
    a = v * (2.0 * qw * qw - 1.0).unsqueeze(-1)
    b = torch.cross(qv, v, dim=-1) * qw.unsqueeze(-1) * 2.0
    c = qv * (qv * v).sum(-1, keepdim=True) * 2.0
    return a + b + c


def quat_rotate_inverse(q, v):
    # reference: rotations.py:101-116, torch_utils.py:72-79
    qw = q[..., 3]
    qv = q[..., :3]
    a = v * (2.0 * qw * qw - 1.0).unsqueeze(-1)
    b = torch.cross(qv, v, dim=-1) * qw.unsqueeze(-1) * 2.0
    c = qv * (qv * v).sum(-1, keepdim=True) * 2.0
    return a - b + c


def quat_apply(q, v):
    # reference: rotations.py:28-39, torch_utils.py:51-57
    xyz = q[..., :3]
    t = torch.cross(xyz, v, dim=-1) * 2
    return v + q[..., 3:] * t + torch.cross(xyz, t, dim=-1)


def quat_conjugate(q):
    # reference: rotations.py:57-63
    return torch.cat((-q[..., :3], q[..., 3:]), dim=-1)


def quat_mul(a, b):
    # reference: rotations.py:414-441 (the 9-multiplication form)
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([x, y, z, w], dim=-1)


def slerp(q0, q1, t):
    # reference: rotations.py:210-232.  t broadcasts against [..., 1].
    cos_half = torch.sum(q0 * q1, dim=-1, keepdim=True)
    q1 = torch.where(cos_half < 0, -q1, q1)
    cos_half = torch.abs(cos_half)
    half = torch.acos(cos_half)
    sin_half = torch.sqrt(1.0 - cos_half * cos_half)
    ra = torch.sin((1 - t) * half) / sin_half
    rb = torch.sin(t * half) / sin_half
    new_q = ra * q0 + rb * q1
    new_q = torch.where(torch.abs(sin_half) < 0.001, 0.5 * q0 + 0.5 * q1, new_q)
    new_q = torch.where(torch.abs(cos_half) >= 1, q0, new_q)
    return new_q


def calc_heading(q):
    # reference: rotations.py:257-268
    ref = torch.zeros_like(q[..., :3])
    ref[..., 0] = 1
    rot = quat_rotate(q, ref)
    return torch.atan2(rot[..., 1], rot[..., 0])


def quat_from_angle_axis(angle, axis):
    # reference: rotations.py:138-145
    theta = (angle / 2).unsqueeze(-1)
    xyz = normalize(axis) * theta.sin()
    w = theta.cos()
    return normalize(torch.cat([xyz, w], dim=-1))


def calc_heading_quat(q):
    # reference: rotations.py:281-291
    axis = torch.zeros_like(q[..., :3])
    axis[..., 2] = 1
    return quat_from_angle_axis(calc_heading(q), axis)


def calc_heading_quat_inv(q):
    # reference: rotations.py:296-306
    axis = torch.zeros_like(q[..., :3])
    axis[..., 2] = 1
    return quat_from_angle_axis(-calc_heading(q), axis)


def calc_yaw_heading_quat_inv(yaw):
    # reference: rotations.py:309-322 ; yaw [N,1]
    h = yaw[..., 0] * 0.5
    z = torch.zeros_like(h)
    return torch.stack([z, z, -torch.sin(h), torch.cos(h)], dim=-1)


def get_euler_xyz(q):
    # reference: rotations.py:368-387 (get_euler_xyz_in_tensor) + maths.copysign :16-19
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sinr_cosp = 2.0 * (qw * qx + qy * qz)
    cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz
    roll = torch.atan2(sinr_cosp, cosr_cosp)
    sinp = 2.0 * (qw * qy - qz * qx)
    half_pi = torch.full_like(sinp, math.pi / 2.0)
    pitch = torch.where(torch.abs(sinp) >= 1, torch.abs(half_pi) * torch.sign(sinp), torch.asin(sinp))
    siny_cosp = 2.0 * (qw * qz + qx * qy)
    cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz
    yaw = torch.atan2(siny_cosp, cosy_cosp)
    return torch.stack((roll, pitch, yaw), dim=-1)


def normalize_angle(x):
    # reference: rotations.py:176-177
    return torch.atan2(torch.sin(x), torch.cos(x))


def quat_to_angle_axis(q):
    # reference: rotations.py:185-207
    min_theta = 1e-5
    sin_theta = torch.sqrt(1 - q[..., 3] * q[..., 3])
    angle = normalize_angle(2 * torch.acos(q[..., 3]))
    axis = q[..., :3] / sin_theta.unsqueeze(-1)
    mask = torch.abs(sin_theta) > min_theta
    default_axis = torch.zeros_like(axis)
    default_axis[..., -1] = 1
    angle = torch.where(mask, angle, torch.zeros_like(angle))
    axis = torch.where(mask.unsqueeze(-1), axis, default_axis)
    return angle, axis


def quat_angle_axis(x):
    # reference: rotations.py:119-134 (angle in [0, pi], axis normalised with clamp 1e-9)
    w = x[..., 3]
    axis = x[..., :3]
    s = 2 * (w ** 2) - 1
    angle = s.clamp(-1, 1).arccos()
    axis = axis / axis.norm(p=2, dim=-1, keepdim=True).clamp(min=1e-9)
    return angle, axis


# ---- wxyz helpers used by the motion-library FK (reference "FROM PHC rotation_conversions") ----

def axis_angle_to_quaternion_wxyz(aa):
    # reference: rotations.py:554-578
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    s = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / safe)
    return torch.cat([torch.cos(half), aa * s], dim=-1)


def quaternion_to_matrix_wxyz(q):
    # reference: rotations.py:519-550
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quaternion_wxyz(m):
    # reference: rotations.py:589-636 (+ _sqrt_positive_part :639-647): best-conditioned candidate
    batch = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(batch + (9,)), dim=-1)
    x = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))
    cand = torch.stack(
        [
            torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
            torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
            torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
            torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
        ],
        dim=-2,
    )
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(batch + (1, 4))).squeeze(-2)


def wxyz_to_xyzw(q):
    return q[..., [1, 2, 3, 0]]


def xyzw_to_wxyz(q):
    return q[..., [3, 0, 1, 2]]
