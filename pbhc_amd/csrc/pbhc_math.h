// pbhc_math.h — device-side quaternion / rotation algebra (xyzw), the "device functions of every
// kernel" (SURVEY §8 a1).  Each function follows one function of the reference's
// humanoidverse/isaac_utils/isaac_utils/rotations.py (line numbers cited) op for op, so that the
// three algebraically different rotate formulas keep their own rounding.  Compiled with
// -ffp-contract=off: the reference's eager torch ops are not fused either.
#pragma once
#include <hip/hip_runtime.h>

namespace pbhc {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm3(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

// sin and cos of one argument in ~25 VALU instructions: three-term Cody-Waite reduction by pi/2 (k * c1, k * c2 exact for |k| < 2^13, i.e.
// |x| < ~1.2e4 rad: joint angles, half headings, slerp arcs) + the Cephes single-precision minimax polynomials on [-pi/4, pi/4]
// (max error ~1 ulp, like the library's).  The library's sinf / cosf inline a Payne-Hanek large-argument path at every call site — 185
// static instructions each, ~60 executed per call and two calls where one reduction serves both — which made them 15 % of the step kernel's
// code.  Explicit fmaf: the file is built with -ffp-contract=off (reference op order elsewhere), the fused steps here are deliberate.
__device__ __forceinline__ void sincos_cw(float x, float* sn, float* cs) {
  const float k = __builtin_rintf(x * 0.636619772367581343f);
  float r = __builtin_fmaf(k, -1.5703125f, x);
  r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
  r = __builtin_fmaf(k, -7.54978995489188216e-8f, r);
  const float z = r * r;
  float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  ps = __builtin_fmaf(ps * z, r, r);                               // sin(r)
  float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));  // cos(r)
  const int q = (int)k;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ float sin_cw(float x) { float s, c; sincos_cw(x, &s, &c); return s; }

// rotations.py:82-97 / :244-253 (my_quat_rotate):  v(2w^2-1) + 2w(q x v) + 2 q (q.v)
__device__ __forceinline__ f3 quat_rotate(f4 q, f3 v) {
  float s = 2.0f * q.w * q.w - 1.0f;
  f3 qv = mk3(q.x, q.y, q.z);
  f3 c = cross3(qv, v);
  float d = dot3(qv, v);
  return mk3(v.x * s + c.x * q.w * 2.0f + q.x * d * 2.0f,
             v.y * s + c.y * q.w * 2.0f + q.y * d * 2.0f,
             v.z * s + c.z * q.w * 2.0f + q.z * d * 2.0f);
}
// rotations.py:101-116:  a - b + c
__device__ __forceinline__ f3 quat_rotate_inverse(f4 q, f3 v) {
  float s = 2.0f * q.w * q.w - 1.0f;
  f3 qv = mk3(q.x, q.y, q.z);
  f3 c = cross3(qv, v);
  float d = dot3(qv, v);
  return mk3(v.x * s - c.x * q.w * 2.0f + q.x * d * 2.0f,
             v.y * s - c.y * q.w * 2.0f + q.y * d * 2.0f,
             v.z * s - c.z * q.w * 2.0f + q.z * d * 2.0f);
}
// rotations.py:414-441: Hamilton product, 9-multiplication form
__device__ __forceinline__ f4 quat_mul(f4 a, f4 b) {
  float ww = (a.z + a.x) * (b.x + b.y);
  float yy = (a.w - a.y) * (b.w + b.z);
  float zz = (a.w + a.y) * (b.w - b.z);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  f4 r;
  r.w = qq - ww + (a.z - a.y) * (b.y - b.z);
  r.x = qq - xx + (a.x + a.w) * (b.x + b.w);
  r.y = qq - yy + (a.w - a.x) * (b.y + b.z);
  r.z = qq - zz + (a.z + a.y) * (b.w - b.x);
  return r;
}
__device__ __forceinline__ f4 quat_conj(f4 q) { return mk4(-q.x, -q.y, -q.z, q.w); }
// maths.py:6-8 normalize (clamp 1e-9)
__device__ __forceinline__ f4 quat_unit(f4 q) {
  float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  n = fmaxf(n, 1e-9f);
  return mk4(q.x / n, q.y / n, q.z / n, q.w / n);
}
// The same normalisation with ONE hardware reciprocal square root (v_rsq_f32, 1 ulp) instead of a correctly rounded square root (~15
// instructions) and four correctly rounded divisions (~10 each): for the sim-stub's rigid-body chain — Isaac Gym's job in the reference, so
// there is no reference op order to mirror — where it runs once per joint of every body's chain.  Differs from quat_unit by <= 2 ulp per
// component; the 1e-9 clamp of the norm becomes a 1e-18 clamp of its square (a chain quaternion is a product of unit quaternions).
__device__ __forceinline__ f4 quat_unit_fast(f4 q) {
  const float r = __builtin_amdgcn_rsqf(fmaxf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, 1e-18f));
  return mk4(q.x * r, q.y * r, q.z * r, q.w * r);
}
// quat_from_angle_axis for a UNIT axis (the skeleton image holds normalised joint axes): (axis sin(a/2), cos(a/2)), unit by construction —
// the reference form below normalises the axis and the result again (two square roots, seven divisions)
__device__ __forceinline__ f4 quat_from_angle_unit_axis(float angle, f3 axis) {
  float s, c;
  sincos_cw(0.5f * angle, &s, &c);
  return mk4(axis.x * s, axis.y * s, axis.z * s, c);
}
__device__ __forceinline__ f3 normalize3(f3 v) {
  float n = fmaxf(norm3(v), 1e-9f);
  return mk3(v.x / n, v.y / n, v.z / n);
}
// rotations.py:210-232 slerp with the <0.001 sin and >=1 cos fall-backs
__device__ __forceinline__ f4 slerp(f4 q0, f4 q1, float t) {
  float c = q0.x * q1.x + q0.y * q1.y + q0.z * q1.z + q0.w * q1.w;
  if (c < 0.0f) { q1.x = -q1.x; q1.y = -q1.y; q1.z = -q1.z; q1.w = -q1.w; }
  c = fabsf(c);
  float half = acosf(c);
  float s = sqrtf(1.0f - c * c);
  const float rs = __builtin_amdgcn_rcpf(s);             // (inf for s = 0: that pair takes the `c >= 1` branch below)
  float ra = sin_cw((1.0f - t) * half) * rs;
  float rb = sin_cw(t * half) * rs;
  f4 r = mk4(ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w);
  if (fabsf(s) < 0.001f) r = mk4(0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w);
  if (fabsf(c) >= 1.0f) r = q0;
  return r;
}
// rotations.py:257-268 heading = atan2 of the rotated x axis
__device__ __forceinline__ float calc_heading(f4 q) {
  f3 r = quat_rotate(q, mk3(1.0f, 0.0f, 0.0f));
  return atan2f(r.y, r.x);
}
// rotations.py:138-145 with axis = z (as calc_heading_quat(_inv) :281-306 call it)
__device__ __forceinline__ f4 quat_from_angle_z(float angle) {
  float th = angle / 2.0f;
  // normalize(axis) = (0,0,1)/max(1,1e-9)
  float sn, cs;
  sincos_cw(th, &sn, &cs);
  f4 q = mk4(0.0f * sn, 0.0f * sn, 1.0f * sn, cs);
  return quat_unit(q);
}
__device__ __forceinline__ f4 quat_from_angle_axis(float angle, f3 axis) {
  float th = angle / 2.0f;
  f3 a = normalize3(axis);
  float s, c;
  sincos_cw(th, &s, &c);
  return quat_unit(mk4(a.x * s, a.y * s, a.z * s, c));
}
// rotations.py:368-387 (+ maths.copysign :16-19).  The arguments of the three inverse functions are exposed so that a caller can spread
// them over lanes (k_env_step evaluates one atan2 per lane); euler_xyz is the plain composition.
__device__ __forceinline__ void euler_xyz_args(f4 q, float* sinr, float* cosr, float* sinp, float* siny, float* cosy) {
  *sinr = 2.0f * (q.w * q.x + q.y * q.z);
  *cosr = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
  *sinp = 2.0f * (q.w * q.y - q.z * q.x);
  *siny = 2.0f * (q.w * q.z + q.x * q.y);
  *cosy = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
}
__device__ __forceinline__ f3 euler_xyz(f4 q) {
  float sinr, cosr, sinp, siny, cosy;
  euler_xyz_args(q, &sinr, &cosr, &sinp, &siny, &cosy);
  float roll = atan2f(sinr, cosr);
  float sgn = (sinp > 0.0f) ? 1.0f : ((sinp < 0.0f) ? -1.0f : 0.0f);
  float pitch = (fabsf(sinp) >= 1.0f) ? (1.5707963267948966f * sgn) : asinf(sinp);
  return mk3(roll, pitch, atan2f(siny, cosy));
}

// ---- arithmetic of the sim-stub's rigid-body chain (fk_walk / fk_walk_wave), CONTRACTED on purpose.  The rest of this file is built with
// -ffp-contract=off because it mirrors the reference's op order; the rigid-body state, though, was Isaac Gym's output in the reference —
// there is no reference arithmetic to mirror — and the chain is the longest dependent stretch of k_env_step: with fused multiply-adds a
// level of the walk is ~75 VALU instructions instead of ~130 (rotate 18 instead of 30, Hamilton product 16 instead of 28).
#pragma clang fp contract(fast)
// v + 2 w (u x v) + 2 u x (u x v), u = q.xyz, for a unit q
__device__ __forceinline__ f3 fk_rotate(f4 q, f3 v) {
  const float tx = 2.0f * (q.y * v.z - q.z * v.y), ty = 2.0f * (q.z * v.x - q.x * v.z), tz = 2.0f * (q.x * v.y - q.y * v.x);
  return mk3(v.x + q.w * tx + (q.y * tz - q.z * ty), v.y + q.w * ty + (q.z * tx - q.x * tz), v.z + q.w * tz + (q.x * ty - q.y * tx));
}
// Hamilton product a (x) b, xyzw, renormalised with one hardware rsq (as quat_unit_fast)
__device__ __forceinline__ f4 fk_mul_unit(f4 a, f4 b) {
  const float x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  const float y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  const float z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  const float w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  const float r = __builtin_amdgcn_rsqf(fmaxf(x * x + y * y + z * z + w * w, 1e-18f));
  return mk4(x * r, y * r, z * r, w * r);
}
// a + s b, a + b x c
// the pieces of the pointer-jumping chain (fk_jump_wave): Hamilton product without the renormalisation (done once, after the last round),
// and ONE quaternion turned into its rotation matrix for the three vectors a composition rotates by it (15 + 3 x 9 instead of 3 x 18)
__device__ __forceinline__ f4 fk_mul(f4 a, f4 b) {
  return mk4(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
struct fkm33 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };
__device__ __forceinline__ fkm33 fk_matrix(f4 q) {
  const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
  const float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2, xy = q.x * y2, xz = q.x * z2, yz = q.y * z2, wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
  return fkm33{1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0f - (xx + yy)};
}
__device__ __forceinline__ f3 fk_mv(const fkm33& m, f3 v) {
  return mk3(m.m00 * v.x + m.m01 * v.y + m.m02 * v.z, m.m10 * v.x + m.m11 * v.y + m.m12 * v.z, m.m20 * v.x + m.m21 * v.y + m.m22 * v.z);
}
__device__ __forceinline__ f3 fk_axpy(f3 a, float s, f3 b) { return mk3(a.x + s * b.x, a.y + s * b.y, a.z + s * b.z); }
__device__ __forceinline__ f3 fk_add_cross(f3 a, f3 b, f3 c) {
  return mk3(a.x + (b.y * c.z - b.z * c.y), a.y + (b.z * c.x - b.x * c.z), a.z + (b.x * c.y - b.y * c.x));
}
#pragma clang fp contract(off)

// ---- general-tracking helpers (humanoidverse/utils/torch_utils.py) ------------------------------
// torch_utils.py:51-57 quat_apply: b + w t + xyz x t, t = 2 (xyz x b)
__device__ __forceinline__ f3 quat_apply(f4 q, f3 b) {
  f3 xyz = mk3(q.x, q.y, q.z);
  f3 t = mul3(cross3(xyz, b), 2.0f);
  f3 u = cross3(xyz, t);
  return mk3(b.x + q.w * t.x + u.x, b.y + q.w * t.y + u.y, b.z + q.w * t.z + u.z);
}
// torch_utils.py:239-270 yaw_quat (xyzw)
__device__ __forceinline__ f4 yaw_quat(f4 q) {
  float yaw = atan2f(2.0f * (q.w * q.z + q.x * q.y), 1.0f - 2.0f * (q.y * q.y + q.z * q.z));
  float sn, cs;
  sincos_cw(yaw / 2.0f, &sn, &cs);
  return quat_unit(mk4(0.0f, 0.0f, sn, cs));
}
// torch_utils.py:274-296 matrix_from_quat(...)[..., :2] flattened: m00 m01 m10 m11 m20 m21
__device__ __forceinline__ void quat_to_mat6(f4 q, float* o) {
  const float i = q.x, j = q.y, k = q.z, r = q.w;
  const float two_s = 2.0f / (i * i + j * j + k * k + r * r);
  o[0] = 1.0f - two_s * (j * j + k * k); o[1] = two_s * (i * j - k * r);
  o[2] = two_s * (i * j + k * r);        o[3] = 1.0f - two_s * (i * i + k * k);
  o[4] = two_s * (i * k - j * r);        o[5] = two_s * (j * k + i * r);
}
// rotations.py:185-207 quat_to_angle_axis(q)[0]: normalize_angle(2 acos w), zero where sqrt(1-w^2) <= 1e-5 (or NaN)
__device__ __forceinline__ float quat_angle(f4 q) {
  const float st = sqrtf(1.0f - q.w * q.w);
  float a = 2.0f * acosf(q.w);
  float sn, cs;
  sincos_cw(a, &sn, &cs);
  a = atan2f(sn, cs);
  return (fabsf(st) > 1e-5f) ? a : 0.0f;
}

// ---- wxyz helpers of the motion-library FK (rotations.py:519-636) --------------------------
struct m33 { float m[9]; };
__device__ __forceinline__ m33 quat_wxyz_to_matrix(float r, float i, float j, float k) {
  float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  m33 o;
  o.m[0] = 1.0f - two_s * (j * j + k * k); o.m[1] = two_s * (i * j - k * r); o.m[2] = two_s * (i * k + j * r);
  o.m[3] = two_s * (i * j + k * r); o.m[4] = 1.0f - two_s * (i * i + k * k); o.m[5] = two_s * (j * k - i * r);
  o.m[6] = two_s * (i * k - j * r); o.m[7] = two_s * (j * k + i * r); o.m[8] = 1.0f - two_s * (i * i + j * j);
  return o;
}
__device__ __forceinline__ m33 matmul33(const m33& a, const m33& b) {
  m33 c;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int col = 0; col < 3; ++col)
      c.m[r * 3 + col] = a.m[r * 3 + 0] * b.m[0 * 3 + col] + a.m[r * 3 + 1] * b.m[1 * 3 + col] + a.m[r * 3 + 2] * b.m[2 * 3 + col];
  return c;
}
// rotations.py:554-578 axis-angle -> wxyz quaternion (Taylor below 1e-6)
__device__ __forceinline__ void axis_angle_to_quat_wxyz(float ax, float ay, float az, float* q) {
  float angle = sqrtf(ax * ax + ay * ay + az * az);
  float half = angle * 0.5f;
  float sh, ch;
  sincos_cw(half, &sh, &ch);
  float s = (fabsf(angle) < 1e-6f) ? (0.5f - (angle * angle) / 48.0f) : (sh / angle);
  q[0] = ch; q[1] = ax * s; q[2] = ay * s; q[3] = az * s;
}
// rotations.py:589-636 best-conditioned candidate, output xyzw
__device__ __forceinline__ f4 matrix_to_quat_xyzw(const m33& M) {
  const float* m = M.m;
  float x0 = 1.0f + m[0] + m[4] + m[8], x1 = 1.0f + m[0] - m[4] - m[8], x2 = 1.0f - m[0] + m[4] - m[8], x3 = 1.0f - m[0] - m[4] + m[8];
  float qa[4] = {x0 > 0.0f ? sqrtf(x0) : 0.0f, x1 > 0.0f ? sqrtf(x1) : 0.0f, x2 > 0.0f ? sqrtf(x2) : 0.0f, x3 > 0.0f ? sqrtf(x3) : 0.0f};
  int best = 0;
  for (int i = 1; i < 4; ++i) if (qa[i] > qa[best]) best = i;   // argmax, first on ties
  float den = 2.0f * fmaxf(qa[best], 0.1f);
  float c[4];
  if (best == 0) { c[0] = qa[0] * qa[0]; c[1] = m[7] - m[5]; c[2] = m[2] - m[6]; c[3] = m[3] - m[1]; }
  else if (best == 1) { c[0] = m[7] - m[5]; c[1] = qa[1] * qa[1]; c[2] = m[3] + m[1]; c[3] = m[2] + m[6]; }
  else if (best == 2) { c[0] = m[2] - m[6]; c[1] = m[3] + m[1]; c[2] = qa[2] * qa[2]; c[3] = m[5] + m[7]; }
  else { c[0] = m[3] - m[1]; c[1] = m[6] + m[2]; c[2] = m[7] + m[5]; c[3] = qa[3] * qa[3]; }
  return mk4(c[1] / den, c[2] / den, c[3] / den, c[0] / den);
}

// ---- Philox4x32-7 counter RNG (own; the reference draws from torch's global generator, so noise-on runs are compared
// statistically, never bitwise).  7 rounds is the smallest Crush-resistant member of the family (Salmon et al., "Parallel random
// numbers: as easy as 1, 2, 3", SC'11, Table 2); the step kernel is instruction-bound and runs the generator at four call sites. ----
#define PBHC_PHILOX_ROUNDS 7
__device__ __forceinline__ void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
#pragma unroll
  for (int i = 0; i < PBHC_PHILOX_ROUNDS; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform in [0,1): 24 mantissa bits
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// four uniforms of one Philox call, keyed (env, step, stream, idx): for draws that are consumed together
__device__ __forceinline__ void rng_uniform4(uint64_t seed, uint32_t env, uint32_t step, uint32_t stream, uint32_t idx, float u[4]) {
  uint32_t o[4];
  philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), env, step, stream, idx, o);
  u[0] = u01(o[0]); u[1] = u01(o[1]); u[2] = u01(o[2]); u[3] = u01(o[3]);
}
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint32_t env, uint32_t step, uint32_t stream, uint32_t idx) {
  uint32_t o[4];
  philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), env, step, stream, idx >> 2, o);
  return u01(o[idx & 3]);
}

}  // namespace pbhc
