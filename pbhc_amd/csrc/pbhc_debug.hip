// pbhc_debug.hip — test-only entry: runs the device functions of pbhc_math.h (the quaternion / rotation algebra every kernel of this
// library inlines) elementwise over caller-supplied arrays, so that tests can pin them DIRECTLY against the reference's own outputs
// (tests/golden/rotations.npz, generated from humanoidverse/isaac_utils/isaac_utils/rotations.py) instead of only through env traces.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/pbhc_hip.h"
#include "pbhc_math.h"

using namespace pbhc;

extern thread_local char g_pbhc_err[512];

__global__ void k_debug_rotations(int fn, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto q4 = [&](const float* p) { return mk4(p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]); };
  auto v3 = [&](const float* p) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); };
  auto o3 = [&](f3 v) { out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; };
  auto o4 = [&](f4 v) { out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w; };
  switch (fn) {
    case PBHC_DBG_QUAT_ROTATE: o3(quat_rotate(q4(a), v3(b))); break;                       // rotations.py:82-97, :244-253
    case PBHC_DBG_QUAT_ROTATE_INVERSE: o3(quat_rotate_inverse(q4(a), v3(b))); break;       // :101-116, torch_utils.py:71-79
    case PBHC_DBG_QUAT_APPLY: o3(quat_apply(q4(a), v3(b))); break;                         // :28-39
    case PBHC_DBG_QUAT_MUL: o4(quat_mul(q4(a), q4(b))); break;                             // :414-441
    case PBHC_DBG_QUAT_CONJ: o4(quat_conj(q4(a))); break;                                  // :57-63
    case PBHC_DBG_SLERP: o4(slerp(q4(a), q4(b), c[i])); break;                             // :210-232
    case PBHC_DBG_CALC_HEADING: out[i] = calc_heading(q4(a)); break;                       // :257-268
    case PBHC_DBG_CALC_HEADING_QUAT: o4(quat_from_angle_z(calc_heading(q4(a)))); break;    // :281-293
    case PBHC_DBG_CALC_HEADING_QUAT_INV: o4(quat_from_angle_z(-calc_heading(q4(a)))); break;   // :296-306 (as k_env_step phase C calls it)
    case PBHC_DBG_EULER_XYZ: o3(euler_xyz(q4(a))); break;                                  // :368-387
    case PBHC_DBG_QUAT_FROM_ANGLE_AXIS: o4(quat_from_angle_axis(a[i], v3(b))); break;      // :138-145
    case PBHC_DBG_QUAT_ANGLE: out[i] = quat_angle(q4(a)); break;                           // :185-207, the angle
    case PBHC_DBG_AXIS_ANGLE_TO_QUAT_WXYZ: axis_angle_to_quat_wxyz(a[3 * i], a[3 * i + 1], a[3 * i + 2], out + 4 * i); break;   // :554-578
    case PBHC_DBG_QUAT_TO_MATRIX: {                                                        // :519-550 (input xyzw)
      const f4 q = q4(a);
      const m33 m = quat_wxyz_to_matrix(q.w, q.x, q.y, q.z);
      for (int k = 0; k < 9; ++k) out[9 * i + k] = m.m[k];
    } break;
    case PBHC_DBG_MATRIX_TO_QUAT: {                                                        // :589-636 (output xyzw)
      m33 m;
      for (int k = 0; k < 9; ++k) m.m[k] = a[9 * i + k];
      o4(matrix_to_quat_xyzw(m));
    } break;
    case PBHC_DBG_YAW_QUAT: o4(yaw_quat(q4(a))); break;                                    // torch_utils.py:239-270
    case PBHC_DBG_QUAT_TO_MAT6: quat_to_mat6(q4(a), out + 6 * i); break;                   // torch_utils.py:274-296, [..., :2]
    case PBHC_DBG_QUAT_UNIT: o4(quat_unit(q4(a))); break;                                  // maths.py:6-8
    default: break;
  }
}

extern "C" int pbhc_debug_rotations(int fn, const float* a, const float* b, const float* c, int n, float* out, void* stream) {
  if (fn < 0 || fn >= PBHC_DBG_NUM || !a || !out || n < 0) {
    snprintf(g_pbhc_err, sizeof(g_pbhc_err), "pbhc_debug_rotations: bad argument (fn %d, n %d)", fn, n);
    return PBHC_EINVAL;
  }
  const bool need_b = fn == PBHC_DBG_QUAT_ROTATE || fn == PBHC_DBG_QUAT_ROTATE_INVERSE || fn == PBHC_DBG_QUAT_APPLY || fn == PBHC_DBG_QUAT_MUL ||
                      fn == PBHC_DBG_SLERP || fn == PBHC_DBG_QUAT_FROM_ANGLE_AXIS;
  if ((need_b && !b) || (fn == PBHC_DBG_SLERP && !c)) {
    snprintf(g_pbhc_err, sizeof(g_pbhc_err), "pbhc_debug_rotations: fn %d needs more operands", fn);
    return PBHC_EINVAL;
  }
  if (n == 0) return PBHC_OK;
  hipLaunchKernelGGL(k_debug_rotations, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, fn, a, b, c, n, out);
  if (hipGetLastError() != hipSuccess) return PBHC_EHIP;
  return PBHC_OK;
}
