// pbhc_kernels.hip — hand-written gfx950 (CDNA4) kernels + the C ABI of include/pbhc_hip.h.
//
// Hot path: one fused launch per control step (k_env_step) + a 1-block finalize.  Mapping:
// 32 lanes (half a wave64) own one env, lane <-> body / dof / obs element; PBHC_EPB envs per
// 128-thread workgroup, so 4096 envs = 1024 workgroups = 4 per CU.  Per-env rows are read and
// written as contiguous rows (reference [N,C] layouts, which is also what the MLP GEMMs consume);
// the joint chain, the reference frames and the feature row are staged in LDS.
// This is HBM-bound float work: no MFMA on purpose.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>
#include <dlfcn.h>

#include "../../include/pbhc_hip.h"
#include "pbhc_math.h"

using namespace pbhc;



thread_local char g_pbhc_err[512] = "";
#define g_err g_pbhc_err

#define HIP_CHECK(x)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      snprintf(g_err, sizeof(g_err), "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return PBHC_EHIP;                                                                       \
    }                                                                                         \
  } while (0)
#define ARG_CHECK(c)                                                                    \
  do {                                                                                  \
    if (!(c)) {                                                                         \
      snprintf(g_err, sizeof(g_err), "%s:%d bad argument: %s", __FILE__, __LINE__, #c); \
      return PBHC_EINVAL;                                                               \
    }                                                                                   \
  } while (0)


#include "pbhc_env_step.h"

// =================================================================================================
//  k_env_finalize: the scalars the reference updates on the host each step
// =================================================================================================
#define PBHC_FIN_CHUNKS 16
__global__ __launch_bounds__(64 * PBHC_FIN_CHUNKS) void k_env_finalize(const PbhcEnvConfig* __restrict__ cfgp, double* __restrict__ glob, const float* __restrict__ partials, int nblocks, int32_t* frame_cursor, int num_frames,
                                                                    const double* __restrict__ ext_tot, double* __restrict__ tot_out, double n_total) {
  // three uses: (partials) reduce + apply [single process]; (partials, tot_out) reduce only, sums to tot_out [data-parallel, first half];
  // (ext_tot) apply sums that the caller has added up over ranks [second half]
  __shared__ double acc[PBHC_FIN_CHUNKS][64];
  __shared__ double tot[PBHC_NP];
  const PbhcEnvConfig& c = *cfgp;
  const int k = threadIdx.x & 63, chunk = threadIdx.x >> 6;
  // column sums of the workgroup partials in a FIXED order (chunk-strided, then chunk order): deterministic
  double s = 0.0;
  if (k < PBHC_NP && !ext_tot) {
    int b = chunk;
    for (; b + 7 * PBHC_FIN_CHUNKS < nblocks; b += 8 * PBHC_FIN_CHUNKS) {       // 8 independent loads in flight, summed in the same fixed order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(b + u * PBHC_FIN_CHUNKS) * PBHC_NP + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nblocks; b += PBHC_FIN_CHUNKS) s += (double)partials[(size_t)b * PBHC_NP + k];
  }
  acc[chunk][k] = s;
  __syncthreads();
  if (threadIdx.x < PBHC_NP) {
    double t = 0.0;
    for (int ch = 0; ch < PBHC_FIN_CHUNKS; ++ch) t += acc[ch][threadIdx.x];
    if (ext_tot) t = ext_tot[threadIdx.x];
    tot[threadIdx.x] = t;
    if (tot_out) tot_out[threadIdx.x] = t;
  }
  __syncthreads();
  // the scalar updates are independent chains of dependent global loads / double divisions: one lane (of different waves) per chain
  const double N = ext_tot ? n_total : (double)c.num_envs;
  const int tid = threadIdx.x;
  if (tot_out) {                                     // reduce-only half: the per-launch counters still advance here
    if (tid == 64) {
      glob[PBHC_G_STEP_COUNTER] += 1.0;
      frame_cursor[0] = (frame_cursor[0] + 1) % num_frames;
    }
    return;
  }
  // adaptive sigma (motion_tracking.py:1030-1048, general_tracking.py:972-996): lane i of wave 0 owns term i
  if (tid < PBHC_NUM_SIGMA) {
    const int i = tid;
    if (c.adaptive_sigma && c.sigma_active[i]) {
      double mean = (double)(float)(tot[P_ERR + i] / N);
      double ema = glob[PBHC_G_EMA + i] * (1.0 - (double)c.adaptive_alpha) + mean * (double)c.adaptive_alpha;
      glob[PBHC_G_EMA + i] = ema;
      const double sg = glob[PBHC_G_SIGMA + i];
      double nsg;
      switch (c.adaptive_type) {
        case 1: nsg = (fmin(ema, sg) + ema) / 2.0; break;
        case 2: nsg = fmin(ema * (double)c.adaptive_scale, sg); break;
        case 3: nsg = ema; break;
        default: nsg = fmin(ema, sg);
      }
      glob[PBHC_G_SIGMA + i] = nsg;
    }
    return;
  }
  double* L = glob + PBHC_G_LOG;
  const double nreset = tot[P_RESET_CNT];
  const double rfrac = nreset / N;
  if (tid == 64) {
    L[PBHC_L_UPPER_BODY_DIFF_NORM] = tot[P_UPPER_NORM] / N; L[PBHC_L_LOWER_BODY_DIFF_NORM] = tot[P_LOWER_NORM] / N;
    L[PBHC_L_VR_3POINT_DIFF_NORM] = tot[P_VR_NORM] / N; L[PBHC_L_JOINT_POS_DIFF_NORM] = tot[P_JOINT_NORM] / N;
    L[PBHC_L_ACTION_CLIP_FRAC] = tot[P_CLIP_CNT] / (N * (double)c.skel.num_dof);
    L[PBHC_L_RESET_FRAC] = rfrac; L[PBHC_L_NUM_RESETS] = nreset;
    L[PBHC_L_REW_MEAN] = tot[P_REW_SUM] / N;
    if (!ext_tot) {
      glob[PBHC_G_STEP_COUNTER] += 1.0;
      frame_cursor[0] = (frame_cursor[0] + 1) % num_frames;
    }
  } else if (tid == 128) {
    L[PBHC_L_TERM_GRAVITY] = (tot[P_TERM_GRAVITY] / N) / (rfrac + 1e-15); L[PBHC_L_TERM_MOTION_FAR] = (tot[P_TERM_FAR] / N) / (rfrac + 1e-15);
    L[PBHC_L_TERM_TIME_OUT] = (tot[P_TERM_TIMEOUT] / N) / (rfrac + 1e-15); L[PBHC_L_TERM_MOTION_END] = (tot[P_TERM_END] / N) / (rfrac + 1e-15);
    L[PBHC_L_TERM_CONTACT] = (tot[P_TERM_CONTACT] / N) / (rfrac + 1e-15); L[PBHC_L_TERM_LOW_HEIGHT] = (tot[P_TERM_LOWH] / N) / (rfrac + 1e-15);
    L[PBHC_L_TERM_DOF_POS_LIMIT] = (tot[P_TERM_POSLIM] / N) / (rfrac + 1e-15); L[PBHC_L_TERM_DOF_VEL_LIMIT] = (tot[P_TERM_VELLIM] / N) / (rfrac + 1e-15);
    L[PBHC_L_TERM_TORQUE_LIMIT] = (tot[P_TERM_TAULIM] / N) / (rfrac + 1e-15);
    if (c.tracking_mode) {
      L[PBHC_L_TERM_REF_POS_Z] = (tot[P_TERM_REFZ] / N) / (rfrac + 1e-15); L[PBHC_L_TERM_REF_ORI] = (tot[P_TERM_REFORI] / N) / (rfrac + 1e-15);
      L[PBHC_L_TERM_BODY_Z] = (tot[P_TERM_BODYZ] / N) / (rfrac + 1e-15);
    }
  } else if (tid == 192) {
    if (c.tracking_mode) {
      L[PBHC_L_KEY_BODY_DIFF_NORM] = tot[P_KEY_NORM] / N; L[PBHC_L_LOCAL_UPPER_BODY_DIFF_NORM] = tot[P_LUP_NORM] / N;
      L[PBHC_L_LOCAL_LOWER_BODY_DIFF_NORM] = tot[P_LLO_NORM] / N; L[PBHC_L_LOCAL_VR_3POINT_DIFF_NORM] = tot[P_LVR_NORM] / N;
      L[PBHC_L_LOCAL_KEY_BODY_DIFF_NORM] = tot[P_LKEY_NORM] / N;
    }
  } else if (tid == 256 && nreset > 0.0) {
    // _update_average_episode_length (legged_robot_base.py:875-879), fp32 like the reference's 0-dim tensor
    float cur = (float)(tot[P_RESET_EPLEN] / nreset);
    double frac = nreset / (double)c.num_compute_average_epl;
    float avg = (float)glob[PBHC_G_AVG_EP_LEN] * (float)(1.0 - frac) + cur * (float)frac;
    glob[PBHC_G_AVG_EP_LEN] = (double)avg;
    if (c.penalty_curriculum) {           // legged_robot_base.py:882-900
      double p = glob[PBHC_G_PENALTY_SCALE];
      if (avg < c.penalty_down) p *= (1.0 - (double)c.penalty_degree);
      else if (avg > c.penalty_up) p *= (1.0 + (double)c.penalty_degree);
      glob[PBHC_G_PENALTY_SCALE] = fmin(fmax(p, (double)c.penalty_min), (double)c.penalty_max);
    }
    {                                     // _update_reward_limits_curriculum, legged_robot_base.py:902-939 (python floats: double)
      const int on[3] = {c.soft_pos_curriculum, c.soft_vel_curriculum, c.soft_tau_curriculum};
      const int slot[3] = {PBHC_G_SOFT_POS_VAL, PBHC_G_SOFT_VEL_VAL, PBHC_G_SOFT_TAU_VAL};
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (!on[q]) continue;
        double v = glob[slot[q]];
        if (avg < c.soft_cur_down[q]) v *= (1.0 + (double)c.soft_cur_degree[q]);
        else if (avg > c.soft_cur_up[q]) v *= (1.0 - (double)c.soft_cur_degree[q]);
        glob[slot[q]] = fmin(fmax(v, (double)c.soft_cur_min[q]), (double)c.soft_cur_max[q]);
      }
    }
    if (c.noise_curriculum) {             // _update_obs_noise_curriculum, legged_robot_base.py:1117-1126 (after the penalty / limit curricula)
      double v = glob[PBHC_G_NOISE_CURRICULUM];
      if (avg < c.noise_down) v *= (1.0 - (double)c.noise_degree);
      else if (avg > c.noise_up) v *= (1.0 + (double)c.noise_degree);
      glob[PBHC_G_NOISE_CURRICULUM] = fmin(fmax(v, (double)c.noise_min), (double)c.noise_max);
    }
    if (c.terminate_when_motion_far && c.motion_far_curriculum) {   // motion_tracking.py:309-317
      double t = glob[PBHC_G_MOTION_FAR_THR];
      if (avg < c.motion_far_down) t *= (1.0 + (double)c.motion_far_degree);
      else if (avg > c.motion_far_up) t *= (1.0 - (double)c.motion_far_degree);
      glob[PBHC_G_MOTION_FAR_THR] = fmin(fmax(t, (double)c.motion_far_min), (double)c.motion_far_max);
    }
  } else if (tid == 320 && nreset > 0.0) {
    double mean = tot[P_ETR_SUM] / N;
    L[PBHC_L_END_TIME_RATIO] = mean;
    double var = (tot[P_ETR_SQ] - N * mean * mean) / (N - 1.0);
    L[PBHC_L_END_TIME_RATIO_STD] = var > 0.0 ? sqrt(var) : 0.0;
  }
}

// =================================================================================================
//  standalone kernels: sim FK, motion state, load-time motion build
// =================================================================================================
__global__ __launch_bounds__(PBHC_G* PBHC_EPB) void k_sim_fk(PbhcSkeleton sk, const float* __restrict__ root_states, const float* __restrict__ dof_pos,
                                                            const float* __restrict__ dof_vel, int dof_stride, int n, float* __restrict__ out) {
  __shared__ float sm[PBHC_EPB][16 + 64 + PBHC_MAX_BODIES * 17];
  __shared__ float skc[SKC_WORDS];
  const int lane = threadIdx.x & (PBHC_G - 1), le = threadIdx.x / PBHC_G, env = blockIdx.x * PBHC_EPB + le;
  const bool valid = env < n;
  float* root = sm[le];
  float *q = root + 16, *qd = q + 32, *bp = qd + 32, *bq = bp + 3 * PBHC_MAX_BODIES, *bv = bq + 4 * PBHC_MAX_BODIES, *bw = bv + 3 * PBHC_MAX_BODIES;
  float* relq = bw + 3 * PBHC_MAX_BODIES;
  stage_skeleton(sk, skc);

  if (valid) {
    if (lane < 13) root[lane] = root_states[(size_t)env * 13 + lane];
    for (int d = lane; d < sk.num_dof; d += PBHC_G) {
      q[d] = dof_pos[((size_t)env * sk.num_dof + d) * dof_stride];
      qd[d] = dof_vel[((size_t)env * sk.num_dof + d) * dof_stride];
    }
  }
  __syncthreads();
  fk_walk(skc, sk.num_bodies, sk.num_bodies, lane, valid, root, q, qd, relq, bp, bq, bv, bw);
  if (valid)
    for (int b = lane; b < sk.num_bodies; b += PBHC_G) {
      float* o = out + ((size_t)env * sk.num_bodies + b) * 13;
      st3(o, ld3(bp + 3 * b)); st4(o + 3, ld4(bq + 4 * b)); st3(o + 7, ld3(bv + 3 * b)); st3(o + 10, ld3(bw + 3 * b));
    }
}

__global__ __launch_bounds__(PBHC_G* PBHC_EPB) void k_motion_state(PbhcMotionTable tbl, int Bx, int D, const int64_t* __restrict__ ids, const float* __restrict__ times,
                                                                  const float* __restrict__ offset, int n, float* __restrict__ out) {
  __shared__ float sm[PBHC_EPB][64 + 8 + PBHC_MAX_BODIES * 13];
  const int lane = threadIdx.x & (PBHC_G - 1), le = threadIdx.x / PBHC_G, i = blockIdx.x * PBHC_EPB + le;
  const bool valid = i < n;
  float* rdof = sm[le];
  float *rdofv = rdof + 32, *rc = rdofv + 32, *rp = rc + 8, *rq = rp + 3 * PBHC_MAX_BODIES, *rv = rq + 4 * PBHC_MAX_BODIES, *rw = rv + 3 * PBHC_MAX_BODIES;
  if (valid) {
    f3 off = offset ? ld3(offset + (size_t)i * 3) : mk3(0.f, 0.f, 0.f);
    motion_lookup(tbl, D, Bx, lane, (int)ids[i], times[i], off, true, rdof, rdofv, rc, rp, rq, rv, rw);
  }
  __syncthreads();
  if (!valid) return;
  float* o = out + (size_t)i * tbl.row;
  for (int d = lane; d < D; d += PBHC_G) { o[d] = rdof[d]; o[D + d] = rdofv[d]; }
  if (lane < 2) o[2 * D + lane] = rc[lane];
  int o_pos = 2 * D + 2, o_rot = o_pos + 3 * Bx, o_vel = o_rot + 4 * Bx, o_ang = o_vel + 3 * Bx;
  for (int k = lane; k < 3 * Bx; k += PBHC_G) { o[o_pos + k] = rp[k]; o[o_vel + k] = rv[k]; o[o_ang + k] = rw[k]; }
  for (int k = lane; k < 4 * Bx; k += PBHC_G) o[o_rot + k] = rq[k];
}

// load-time FK: one thread per frame walks the chain (Humanoid_Batch.fk_batch +
// forward_kinematics_batch, torch_humanoid_batch.py:168-269)
__global__ void k_motion_fk(PbhcSkeleton sk, const float* __restrict__ pose_aa, const float* __restrict__ trans, int F,
                            float* __restrict__ pos, float* __restrict__ rot) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int Bx = sk.num_bodies_ext;
  m33 W[PBHC_MAX_BODIES];
  float P[PBHC_MAX_BODIES][3];
  for (int i = 0; i < Bx; ++i) {
    const float* aa = pose_aa + ((size_t)f * Bx + i) * 3;
    float qw[4];
    axis_angle_to_quat_wxyz(aa[0], aa[1], aa[2], qw);
    m33 J = quat_wxyz_to_matrix(qw[0], qw[1], qw[2], qw[3]);
    int p = sk.parent[i];
    if (p < 0) {
      W[i] = J;
      P[i][0] = trans[f * 3 + 0]; P[i][1] = trans[f * 3 + 1]; P[i][2] = trans[f * 3 + 2];
    } else {
      const float* o = sk.offset[i];
      const m33& Rp = W[p];
      P[i][0] = Rp.m[0] * o[0] + Rp.m[1] * o[1] + Rp.m[2] * o[2] + P[p][0];
      P[i][1] = Rp.m[3] * o[0] + Rp.m[4] * o[1] + Rp.m[5] * o[2] + P[p][1];
      P[i][2] = Rp.m[6] * o[0] + Rp.m[7] * o[1] + Rp.m[8] * o[2] + P[p][2];
      m33 Lm = quat_wxyz_to_matrix(sk.local_rot_wxyz[i][0], sk.local_rot_wxyz[i][1], sk.local_rot_wxyz[i][2], sk.local_rot_wxyz[i][3]);
      W[i] = matmul33(Rp, matmul33(Lm, J));
    }
    float* po = pos + ((size_t)f * Bx + i) * 3;
    po[0] = P[i][0]; po[1] = P[i][1]; po[2] = P[i][2];
    st4(rot + ((size_t)f * Bx + i) * 4, matrix_to_quat_xyzw(W[i]));
  }
}

// raw velocities: np.gradient/dt and angle-axis of q_{t+1} * conj(q_t) (torch_humanoid_batch.py:272-290)
// `frame_clip` / `clip_start` / `clip_dt` (may be NULL: one clip of F frames): frame -> clip, first frame of every clip (+ the total),
// frame time per clip — the batched build of a whole library in one launch set (pbhc_motion_build_batch)
__global__ void k_motion_rawvel(const float* __restrict__ pos, const float* __restrict__ rot, int Ftot, int Bx, float dt1,
                                const int32_t* __restrict__ frame_clip, const int32_t* __restrict__ clip_start, const float* __restrict__ clip_dt,
                                float* __restrict__ vel, float* __restrict__ ang) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Ftot * Bx) return;
  int fg = i / Bx, b = i % Bx;
  int base = 0, F = Ftot;
  float dt = dt1;
  if (frame_clip) { const int cidx = frame_clip[fg]; base = clip_start[cidx]; F = clip_start[cidx + 1] - base; dt = clip_dt[cidx]; }
  const int f = fg - base;
  auto P = [&](int ff, int k) { return pos[((size_t)(base + ff) * Bx + b) * 3 + k]; };
  for (int k = 0; k < 3; ++k) {
    float g;
    if (F == 1) g = 0.0f;
    else if (f == 0) g = (P(1, k) - P(0, k)) / 1.0f;
    else if (f == F - 1) g = (P(F - 1, k) - P(F - 2, k)) / 1.0f;
    else g = (P(f + 1, k) - P(f - 1, k)) / 2.0f;
    vel[(size_t)i * 3 + k] = g / dt;
  }
  f4 dq = mk4(0.f, 0.f, 0.f, 1.f);
  if (f < F - 1) dq = quat_unit(quat_mul(ld4(rot + ((size_t)(fg + 1) * Bx + b) * 4), quat_conj(ld4(rot + ((size_t)fg * Bx + b) * 4))));
  float s = 2.0f * (dq.w * dq.w) - 1.0f;
  float angle = acosf(clampf(s, -1.0f, 1.0f));
  float n = fmaxf(sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), 1e-9f);
  ang[(size_t)i * 3 + 0] = dq.x / n * angle / dt;
  ang[(size_t)i * 3 + 1] = dq.y / n * angle / dt;
  ang[(size_t)i * 3 + 2] = dq.z / n * angle / dt;
}

// scipy.ndimage.gaussian_filter1d(sigma=2, mode="nearest", truncate=4) along time, double accumulation,
// then pack the frame rows
__global__ void k_motion_pack(PbhcSkeleton sk, const float* __restrict__ pose_aa, const float* __restrict__ contact, const float* __restrict__ pos,
                              const float* __restrict__ rot, const float* __restrict__ vel, const float* __restrict__ ang, int Ftot, float dt1, int row,
                              const int32_t* __restrict__ frame_clip, const int32_t* __restrict__ clip_start, const float* __restrict__ clip_dt,
                              float* __restrict__ out) {
  const int Bx = sk.num_bodies_ext, D = sk.num_dof, B = sk.num_bodies;
  const int fg = blockIdx.x;
  int base = 0, F = Ftot;
  float dt = dt1;
  if (frame_clip) { const int cidx = frame_clip[fg]; base = clip_start[cidx]; F = clip_start[cidx + 1] - base; dt = clip_dt[cidx]; }
  const int f = fg - base;
  // everything below indexes frames of THIS clip: shift the per-frame arrays to its first frame
  pose_aa += (size_t)base * Bx * 3; pos += (size_t)base * Bx * 3; rot += (size_t)base * Bx * 4; vel += (size_t)base * Bx * 3; ang += (size_t)base * Bx * 3;
  if (contact) contact += (size_t)base * 2;
  float* o = out + (size_t)fg * row;
  double w[9];
  double wsum = 0.0;
  for (int k = 0; k <= 8; ++k) { w[k] = exp(-0.5 * (double)(k * k) / 4.0); wsum += (k == 0 ? 1.0 : 2.0) * w[k]; }
  const int o_pos = 2 * D + 2, o_rot = o_pos + 3 * Bx, o_vel = o_rot + 4 * Bx, o_ang = o_vel + 3 * Bx;
  for (int i = threadIdx.x; i < Bx * 3; i += blockDim.x) {
    double av = 0.0, aw = 0.0;
    for (int k = -8; k <= 8; ++k) {
      int ff = min(max(f + k, 0), F - 1);
      double wk = w[k < 0 ? -k : k] / wsum;
      av += wk * (double)vel[(size_t)ff * Bx * 3 + i];
      aw += wk * (double)ang[(size_t)ff * Bx * 3 + i];
    }
    o[o_vel + i] = (float)av; o[o_ang + i] = (float)aw;
    o[o_pos + i] = pos[(size_t)f * Bx * 3 + i];
  }
  for (int i = threadIdx.x; i < Bx * 4; i += blockDim.x) o[o_rot + i] = rot[(size_t)f * Bx * 4 + i];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    auto dofp = [&](int ff) {
      const float* aa = pose_aa + ((size_t)ff * Bx + (d + 1)) * 3;
      return aa[0] + aa[1] + aa[2];                                   // pose_aa.sum(-1)[1:B]  (:216)
    };
    o[d] = dofp(f);
    float dv;
    if (F < 2) dv = 0.0f;
    else if (f < F - 1) dv = (dofp(f + 1) - dofp(f)) / dt;
    else { int g = F >= 3 ? F - 3 : 0; dv = (dofp(g + 1) - dofp(g)) / dt; }   // `dof_vel[:, -2:-1]`, sic (:224)
    o[D + d] = dv;
  }
  if (threadIdx.x < 2) o[2 * D + threadIdx.x] = contact ? contact[(size_t)f * 2 + threadIdx.x] : 0.0f;
  (void)B;
}

// =================================================================================================
//  GAE (MHPPO._compute_returns mh_ppo.py:348-395)
// =================================================================================================
#define GAE_TPB 256
__global__ __launch_bounds__(GAE_TPB) void k_gae(const float* __restrict__ rewards, const float* __restrict__ values, const uint8_t* __restrict__ dones,
                                                 const float* __restrict__ last_values, int T, int N, int R, float gamma, float lam,
                                                 float* __restrict__ returns) {
  // thread <-> (env, head): consecutive threads walk consecutive floats of every [t] slab
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * R) return;
  int n = (int)(i / R);
  float adv = 0.0f;
  float next = last_values[i];
  for (int t = T - 1; t >= 0; --t) {
    size_t o = (size_t)t * N * R + i;
    float nt = 1.0f - (dones[(size_t)t * N + n] ? 1.0f : 0.0f);
    float v = values[o];
    float delta = rewards[o] + nt * gamma * next - v;
    adv = delta + nt * gamma * lam * adv;
    returns[o] = adv + v;
    next = v;
  }
}
// advantages = sum over heads of (returns - values); block partial sums of x and x^2 (double)
__global__ __launch_bounds__(GAE_TPB) void k_adv_sum(const float* __restrict__ returns, const float* __restrict__ values, size_t TN, int R,
                                                     float* __restrict__ adv, double* __restrict__ part) {
  __shared__ double s1[GAE_TPB], s2[GAE_TPB];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0.0;
  if (i < TN) {
    float s = 0.0f;
    for (int r = 0; r < R; ++r) s += returns[i * R + r] - values[i * R + r];
    adv[i] = s;
    a = (double)s;
  }
  s1[threadIdx.x] = a; s2[threadIdx.x] = a * a;
  __syncthreads();
  for (int st = GAE_TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s1[threadIdx.x] += s1[threadIdx.x + st]; s2[threadIdx.x] += s2[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = s1[0]; part[2 * blockIdx.x + 1] = s2[0]; }
}
// mean / unbiased std of the advantages from the block partials, then the normalisation — in ONE launch: every block re-sums the (few hundred)
// partial pairs itself, all in the same fixed order (thread i takes partials i, i + 256, ...; then a tree), so every block normalises with the
// same two numbers; block 0 also leaves them behind the partials (part[2 nblocks], part[2 nblocks + 1]: the data-parallel path re-normalises
// with the all-rank moments).  (Was a one-thread serial sum, 34 us, and a launch of its own.)
__global__ __launch_bounds__(GAE_TPB) void k_adv_norm(float* __restrict__ adv, size_t TN, double* __restrict__ part, int nblocks) {
  __shared__ double s1[GAE_TPB], s2[GAE_TPB];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += GAE_TPB) { a += part[2 * i]; b += part[2 * i + 1]; }
  s1[threadIdx.x] = a; s2[threadIdx.x] = b;
  __syncthreads();
  for (int st = GAE_TPB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s1[threadIdx.x] += s1[threadIdx.x + st]; s2[threadIdx.x] += s2[threadIdx.x + st]; }
    __syncthreads();
  }
  const double n = (double)TN;
  const double mean_d = s1[0] / n;
  const double var = (s2[0] - n * mean_d * mean_d) / (n - 1.0);       // torch.std: unbiased
  const double sd_d = sqrt(var > 0.0 ? var : 0.0);
  if (blockIdx.x == 0 && threadIdx.x == 0) { part[2 * nblocks] = mean_d; part[2 * nblocks + 1] = sd_d; }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TN) return;
  const float mean = (float)mean_d, sd = (float)sd_d;
  adv[i] = (adv[i] - mean) / (sd + 1e-8f);
}

// =================================================================================================
//  C ABI
// =================================================================================================
// LDS slot of the reduction row that IS the raw value of a reward term, or -1 for the terms computed by a case of the kernel's switch
static int term_source(int id) {
  switch (id) {
    case PBHC_R_TELEOP_MAX_JOINT_POSITION: return R_EXP0 + PBHC_S_MAX_JOINT_POS;
    case PBHC_R_TELEOP_VR_3POINT: return R_EXP0 + PBHC_S_VR_3POINT_POS;
    case PBHC_R_TELEOP_BODY_POSITION_FEET: return R_EXP0 + PBHC_S_FEET_POS;
    case PBHC_R_TELEOP_BODY_ROTATION_EXTEND: return R_EXP0 + PBHC_S_BODY_ROT;
    case PBHC_R_TELEOP_BODY_ANG_VELOCITY_EXTEND: return R_EXP0 + PBHC_S_BODY_ANG_VEL;
    case PBHC_R_TELEOP_BODY_VELOCITY_EXTEND: return R_EXP0 + PBHC_S_BODY_VEL;
    case PBHC_R_TELEOP_JOINT_POSITION: return R_EXP0 + PBHC_S_JOINT_POS;
    case PBHC_R_TELEOP_JOINT_VELOCITY: return R_EXP0 + PBHC_S_JOINT_VEL;
    case PBHC_R_PENALTY_TORQUES: return R_TAU2;
    case PBHC_R_PENALTY_DOF_VEL: return R_QD2;
    case PBHC_R_PENALTY_DOF_ACC: return R_QACC2;
    case PBHC_R_PENALTY_ACTION_RATE: return R_ARATE;
    case PBHC_R_LIMITS_DOF_POS: return R_LIMPOS;
    case PBHC_R_LIMITS_DOF_VEL: return R_LIMVEL;
    case PBHC_R_LIMITS_TORQUE: return R_LIMTAU;
    case PBHC_R_COLLISION: return R_COLL;
    case PBHC_R_TELEOP_KEY_BODY_POSITION: return R_EXP0 + PBHC_S_KEY_BODY_POS;
    case PBHC_R_TELEOP_ANCHOR_BODY_POSITION: return R_EXP0 + PBHC_S_ANCHOR_BODY_POS;
    case PBHC_R_TELEOP_ANCHOR_BODY_ROTATION: return R_EXP0 + PBHC_S_ANCHOR_BODY_ROT;
    case PBHC_R_LOCAL_KEY_BODY_POSITION: return R_EXP0 + PBHC_S_LOCAL_KEY_BODY_POS;
    case PBHC_R_LOCAL_KEY_BODY_ROTATION: return R_EXP0 + PBHC_S_LOCAL_KEY_BODY_ROT;
    case PBHC_R_KEY_BODY_VELOCITY: return R_EXP0 + PBHC_S_KEY_BODY_VEL;
    case PBHC_R_KEY_BODY_ANG_VELOCITY: return R_EXP0 + PBHC_S_KEY_BODY_ANG_VEL;
    case PBHC_R_TELEOP_ROOT_VEL: return R_EXP0 + PBHC_S_ROOT_VEL;
    case PBHC_R_TELEOP_ROOT_POSE: return R_EXP0 + PBHC_S_ROOT_POSE;
    default: return -1;
  }
}

struct PbhcEnv {
  PbhcEnvConfig cfg;
  PbhcEnvConfig* d_cfg;
  PbhcMotionTable tbl;
  double* d_glob;
  float* d_partials;
  float* d_skc;
  float* d_skj;              // per-body rows of the pointer-jumping chain (fk_jump_wave: skj_word)
  int nblocks;
  int lds_stride;
  size_t lds_bytes;
  uint32_t step_ctr;
  int profile;
  int prof_count;
  hipEvent_t ev0[PBHC_PROFILE_RING], ev1[PBHC_PROFILE_RING];
  const void* spec_fn;       // config-specialised k_env_step (pbhc_env_attach_specialised), or nullptr: the generic kernel
  void* spec_dl;
  int spec_lds_stride;       // ... and its own LDS plan (step_lds_plan)
  size_t spec_lds_bytes;
  int spec_hist_wide;        // ... which may read the history rows 16 bytes per lane
};

extern "C" {

int pbhc_abi_version(void) { return PBHC_ABI_VERSION; }
#ifdef PBHC_STAMPS
int pbhc_debug_read_stamps(unsigned long long* out, int n) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 64 ? n : 64)));
  return PBHC_OK;
}
int pbhc_debug_read_wg_times(unsigned long long* out, int num_workgroups) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 2 * (num_workgroups < 4096 ? num_workgroups : 4096)));
  return PBHC_OK;
}
#endif
const char* pbhc_last_error(void) { return g_err; }
int pbhc_sizeof_env_config(void) { return (int)sizeof(PbhcEnvConfig); }
int pbhc_sizeof_step_io(void) { return (int)sizeof(PbhcStepIO); }

static int check_skel(const PbhcSkeleton* sk) {
  ARG_CHECK(sk != nullptr);
  ARG_CHECK(sk->num_bodies >= 1 && sk->num_bodies <= sk->num_bodies_ext && sk->num_bodies_ext <= PBHC_MAX_BODIES);
  ARG_CHECK(sk->num_dof == sk->num_bodies - 1 && sk->num_dof <= PBHC_MAX_DOF);
  ARG_CHECK(sk->parent[0] == -1);
  for (int i = 1; i < sk->num_bodies_ext; ++i) ARG_CHECK(sk->parent[i] >= 0 && sk->parent[i] < i && sk->parent[i] < sk->num_bodies);
  for (int i = 1; i < sk->num_bodies; ++i) ARG_CHECK(sk->depth[i] == sk->depth[sk->parent[i]] + 1 && sk->depth[i] <= sk->max_depth);
  for (int b = 0; b < sk->num_bodies_ext; ++b) {
    const int node = b < sk->num_bodies ? b : sk->parent[b];
    ARG_CHECK(sk->chain_len[b] == sk->depth[node] && sk->chain_len[b] <= PBHC_MAX_DEPTH);
    int cur = node;
    for (int k = sk->chain_len[b] - 1; k >= 0; --k) { ARG_CHECK(sk->chain[b][k] == cur && cur >= 1 && cur < sk->num_bodies); cur = sk->parent[cur]; }
    ARG_CHECK(cur == 0);
  }
  return PBHC_OK;
}

int pbhc_motion_build(const PbhcSkeleton* skel, const float* pose_aa, const float* trans, const float* contact, int F, float dt,
                      float* out_rows, float* scratch, void* stream) {
  int rc = check_skel(skel);
  if (rc) return rc;
  ARG_CHECK(pose_aa && trans && out_rows && scratch && F >= 1 && dt > 0.0f);
  hipStream_t st = (hipStream_t)stream;
  const int Bx = skel->num_bodies_ext, D = skel->num_dof;
  const int row = 2 * D + 2 + 13 * Bx;
  float* pos = scratch;
  float* rot = pos + (size_t)F * Bx * 3;
  float* vel = rot + (size_t)F * Bx * 4;
  float* ang = vel + (size_t)F * Bx * 3;
  hipLaunchKernelGGL(k_motion_fk, dim3((F + 63) / 64), dim3(64), 0, st, *skel, pose_aa, trans, F, pos, rot);
  hipLaunchKernelGGL(k_motion_rawvel, dim3((F * Bx + 127) / 128), dim3(128), 0, st, pos, rot, F, Bx, dt, (const int32_t*)nullptr, (const int32_t*)nullptr,
                     (const float*)nullptr, vel, ang);
  hipLaunchKernelGGL(k_motion_pack, dim3(F), dim3(128), 0, st, *skel, pose_aa, contact, pos, rot, vel, ang, F, dt, row, (const int32_t*)nullptr,
                     (const int32_t*)nullptr, (const float*)nullptr, out_rows);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_motion_build_batch(const PbhcSkeleton* skel, const float* pose_aa, const float* trans, const float* contact, int total_frames, int num_clips,
                            const int32_t* frame_clip, const int32_t* clip_start, const float* clip_dt, float* out_rows, float* scratch, void* stream) {
  int rc = check_skel(skel);
  if (rc) return rc;
  ARG_CHECK(pose_aa && trans && out_rows && scratch && frame_clip && clip_start && clip_dt && total_frames >= 1 && num_clips >= 1 && num_clips <= total_frames);
  hipStream_t st = (hipStream_t)stream;
  const int Bx = skel->num_bodies_ext, D = skel->num_dof, F = total_frames;
  const int row = 2 * D + 2 + 13 * Bx;
  float* pos = scratch;
  float* rot = pos + (size_t)F * Bx * 3;
  float* vel = rot + (size_t)F * Bx * 4;
  float* ang = vel + (size_t)F * Bx * 3;
  hipLaunchKernelGGL(k_motion_fk, dim3((F + 63) / 64), dim3(64), 0, st, *skel, pose_aa, trans, F, pos, rot);
  hipLaunchKernelGGL(k_motion_rawvel, dim3((unsigned)(((size_t)F * Bx + 127) / 128)), dim3(128), 0, st, pos, rot, F, Bx, 0.0f, frame_clip, clip_start, clip_dt, vel, ang);
  hipLaunchKernelGGL(k_motion_pack, dim3(F), dim3(128), 0, st, *skel, pose_aa, contact, pos, rot, vel, ang, F, 0.0f, row, frame_clip, clip_start, clip_dt, out_rows);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_motion_state(const PbhcMotionTable* tbl, int Bx, int D, const int64_t* ids, const float* times, const float* offset, int n,
                      float* out, void* stream) {
  ARG_CHECK(tbl && tbl->frames && ids && times && out && n >= 0);
  ARG_CHECK(Bx >= 1 && Bx <= PBHC_MAX_BODIES && D >= 1 && D <= PBHC_MAX_DOF && tbl->row == 2 * D + 2 + 13 * Bx);
  if (n == 0) return PBHC_OK;
  hipLaunchKernelGGL(k_motion_state, dim3((n + PBHC_EPB - 1) / PBHC_EPB), dim3(PBHC_G * PBHC_EPB), 0, (hipStream_t)stream, *tbl, Bx, D, ids, times, offset, n, out);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_sim_fk(const PbhcSkeleton* skel, const float* root_states, const float* dof_pos, const float* dof_vel, int dof_stride, int n,
                float* out, void* stream) {
  int rc = check_skel(skel);
  if (rc) return rc;
  ARG_CHECK(root_states && dof_pos && dof_vel && out && n >= 0 && dof_stride >= 1);
  if (n == 0) return PBHC_OK;
  hipLaunchKernelGGL(k_sim_fk, dim3((n + PBHC_EPB - 1) / PBHC_EPB), dim3(PBHC_G * PBHC_EPB), 0, (hipStream_t)stream, *skel, root_states, dof_pos, dof_vel, dof_stride, n, out);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

// Test-only: the rigid-body state of every body INCLUDING the extended ones by either form of the chain — method 0 the walk (fk_walk, what
// pbhc_sim_fk runs), 1 pointer jumping (fk_jump_wave, what the step kernel runs for robots of <= 32 bodies) — on any skeleton: random trees of
// every depth class (1..4 rounds) in tests/test_gpu_fk.py, where the envs' config-driven tests only reach the G1's three rounds.
__global__ __launch_bounds__(PBHC_G* PBHC_EPB) void k_debug_fk(PbhcSkeleton sk, const float* __restrict__ root_states, const float* __restrict__ dof_pos,
                                                              const float* __restrict__ dof_vel, int n, int method, float* __restrict__ out) {
  __shared__ float sm[PBHC_EPB][16 + 64 + PBHC_MAX_BODIES * 17];
  __shared__ float skc[SKC_WORDS];
  const int lane = threadIdx.x & (PBHC_G - 1), le = threadIdx.x / PBHC_G, env = blockIdx.x * PBHC_EPB + le;
  const bool valid = env < n;
  float* root = sm[le];
  float *q = root + 16, *qd = q + 32, *bp = qd + 32, *bq = bp + 3 * PBHC_MAX_BODIES, *bv = bq + 4 * PBHC_MAX_BODIES, *bw = bv + 3 * PBHC_MAX_BODIES;
  float* relq = bw + 3 * PBHC_MAX_BODIES;
  stage_skeleton(sk, skc);
  if (valid) {
    if (lane < 13) root[lane] = root_states[(size_t)env * 13 + lane];
    for (int d = lane; d < sk.num_dof; d += PBHC_G) { q[d] = dof_pos[(size_t)env * sk.num_dof + d]; qd[d] = dof_vel[(size_t)env * sk.num_dof + d]; }
  }
  __syncthreads();
  const int B = sk.num_bodies, Bx = sk.num_bodies_ext;
  if (method == 0) {
    fk_walk(skc, B, Bx, lane, valid, root, q, qd, relq, bp, bq, bv, bw);
  } else {
    float4 kr[5];
    const int lb = min(lane, Bx - 1);
#pragma unroll
    for (int u = 0; u < 5; ++u) kr[u] = make_float4(skj_word(sk, SKJ_W * lb + 4 * u), skj_word(sk, SKJ_W * lb + 4 * u + 1), skj_word(sk, SKJ_W * lb + 4 * u + 2), skj_word(sk, SKJ_W * lb + 4 * u + 3));
    fk_jump_wave(kr, skel_fk_rounds(sk.max_depth), B, Bx, lane, valid, root, q, qd, bp, bq, bv, bw);
    __syncthreads();
  }
  if (valid)
    for (int b = lane; b < Bx; b += PBHC_G) {
      float* o = out + ((size_t)env * Bx + b) * 13;
      st3(o, ld3(bp + 3 * b)); st4(o + 3, ld4(bq + 4 * b)); st3(o + 7, ld3(bv + 3 * b)); st3(o + 10, ld3(bw + 3 * b));
    }
}
extern "C" int pbhc_debug_fk(const PbhcSkeleton* skel, const float* root_states, const float* dof_pos, const float* dof_vel, int n, int method, float* out, void* stream) {
  ARG_CHECK(skel);
  int rc = check_skel(skel);
  if (rc) return rc;
  ARG_CHECK(root_states && dof_pos && dof_vel && out && n >= 0 && (method == 0 || method == 1));
  if (method == 1 && !skel_fk_jump(skel->num_bodies_ext, skel->max_depth)) {
    snprintf(g_err, sizeof(g_err), "pbhc_debug_fk: pointer jumping needs <= %d bodies and a chain of <= 15 joints", PBHC_G);
    return PBHC_EINVAL;
  }
  if (n == 0) return PBHC_OK;
  hipLaunchKernelGGL(k_debug_fk, dim3((n + PBHC_EPB - 1) / PBHC_EPB), dim3(PBHC_G * PBHC_EPB), 0, (hipStream_t)stream, *skel, root_states, dof_pos, dof_vel, n, method, out);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

// Validation + the members pbhc_env_create derives (term_src, sum_col_term, whether the compact observation maps fit in LDS): host
// arithmetic only, so that a specialised kernel can be generated for a config without a device at hand (pbhc_env_config_finalize).
static int finalize_config(const PbhcEnvConfig* cfg, PbhcEnvConfig* fin, int* lds_stride, size_t* lds_bytes) {
  ARG_CHECK(cfg && fin && lds_stride && lds_bytes);
  ARG_CHECK(cfg->abi_version == PBHC_ABI_VERSION);
  int rc = check_skel(&cfg->skel);
  if (rc) return rc;
  const int Bx = cfg->skel.num_bodies_ext;
  ARG_CHECK(cfg->num_envs >= 1);
  ARG_CHECK(cfg->num_feet >= 1 && cfg->num_feet <= PBHC_MAX_FEET);
  ARG_CHECK(cfg->num_terms >= 0 && cfg->num_terms <= PBHC_MAX_TERMS && cfg->num_terms <= PBHC_G && cfg->num_rew_cols <= PBHC_G);
  ARG_CHECK(cfg->num_groups >= 1 && cfg->num_groups <= PBHC_MAX_GROUPS);
  ARG_CHECK(cfg->num_term_contact >= 0 && cfg->num_term_contact <= PBHC_MAX_IDX);
  for (int i = 0; i < cfg->num_term_contact; ++i) ARG_CHECK(cfg->term_contact[i] >= 0 && cfg->term_contact[i] < cfg->skel.num_bodies);
  ARG_CHECK(cfg->queue_len >= 1 && cfg->queue_len <= PBHC_MAX_QUEUE);
  ARG_CHECK(cfg->feat_dim > 0 && cfg->feat_dim < 16384);
  // the step kernel addresses every per-env tensor with 32-bit element offsets from its base pointer
  ARG_CHECK((uint64_t)cfg->num_envs * (uint64_t)(cfg->skel.num_bodies * 13 > cfg->hist_dim + 64 ? cfg->skel.num_bodies * 13 : cfg->hist_dim + 64) < (1ull << 30));
  for (int g = 0; g < cfg->num_groups; ++g) ARG_CHECK(cfg->groups[g].dim > 0 && cfg->groups[g].src && cfg->groups[g].scale && cfg->groups[g].noise);
  for (int i = 0; i < PBHC_F_NUM; ++i) ARG_CHECK(cfg->feat_off[i] >= 0 && cfg->feat_off[i] < cfg->feat_dim);
  *fin = *cfg;
  for (int i = 0; i < PBHC_MAX_TERMS; ++i) fin->term_src[i] = i < cfg->num_terms ? term_source(cfg->term_id[i]) : -1;
  ARG_CHECK(cfg->num_sum_cols >= 1 && cfg->num_sum_cols <= 32);
  for (int i = 0; i < 32; ++i) fin->sum_col_term[i] = -1;
  for (int i = 0; i < cfg->num_terms; ++i) {
    ARG_CHECK(cfg->term_sum_col[i] >= 0 && cfg->term_sum_col[i] < cfg->num_sum_cols && fin->sum_col_term[cfg->term_sum_col[i]] == -1);
    fin->sum_col_term[cfg->term_sum_col[i]] = i;
  }
  ARG_CHECK(cfg->tracking_mode == 0 || cfg->tracking_mode == 1);
  if (cfg->tracking_mode) {
    ARG_CHECK(cfg->num_key >= 1 && cfg->num_key <= PBHC_MAX_IDX && cfg->anchor_index >= 0 && cfg->anchor_index < Bx);
    ARG_CHECK(cfg->future_num_steps >= 0 && cfg->future_num_steps <= PBHC_MAX_FUTURE);
    for (int k = 0; k < cfg->num_key; ++k) ARG_CHECK(cfg->key[k] >= 0 && cfg->key[k] < Bx);
    for (int k = 0; k < cfg->future_num_steps; ++k) ARG_CHECK(cfg->future_steps[k] >= 0);
  }
  *lds_stride = Lds(Bx, cfg->tracking_mode).feat + ((cfg->feat_dim + 3) & ~3);
  ARG_CHECK(cfg->map_lds_words >= 0);
  if (cfg->map_lds_words > 0) {
    int need = 0;
    for (int g = 0; g < cfg->num_groups; ++g) {
      ARG_CHECK(cfg->groups[g].dst == nullptr && cfg->groups[g].lds_off == need && cfg->groups[g].map_words >= PBHC_MAP_HDR + ((cfg->groups[g].dim + 1) >> 1));
      need += cfg->groups[g].map_words;
    }
    ARG_CHECK(need == cfg->map_lds_words && cfg->feat_dim <= 4096 && cfg->map_image != nullptr);
  }
  {
    // the compact maps must not cost a resident workgroup per CU (160 KB LDS): otherwise the per-element maps stay in global memory
    const size_t base = ((size_t)PBHC_EPB * *lds_stride + (size_t)((Bx * SKC_W + 3) & ~3)) * sizeof(float);
    const size_t with = base + (size_t)cfg->map_lds_words * sizeof(float);
    if (cfg->map_lds_words > 0 && (160 * 1024) / with < (160 * 1024) / base) fin->map_lds_words = 0;
    *lds_bytes = fin->map_lds_words > 0 ? with : base;
  }
  if (*lds_bytes > 160 * 1024) { snprintf(g_err, sizeof(g_err), "feature row too large for LDS"); return PBHC_EINVAL; }
  return PBHC_OK;
}

// `out` = `cfg` as pbhc_env_create would store it.  No device is touched.
int pbhc_env_config_finalize(const PbhcEnvConfig* cfg, PbhcEnvConfig* out) {
  int lds_stride = 0;
  size_t lds_bytes = 0;
  return finalize_config(cfg, out, &lds_stride, &lds_bytes);
}

// dynamic LDS of one k_env_step workgroup (4 envs) for this config, or a negative error code.  No device is touched.
int pbhc_env_config_lds_bytes(const PbhcEnvConfig* cfg) {
  PbhcEnvConfig fin;
  int lds_stride = 0;
  size_t lds_bytes = 0;
  const int rc = finalize_config(cfg, &fin, &lds_stride, &lds_bytes);
  return rc != PBHC_OK ? -rc : (int)lds_bytes;
}

int pbhc_env_create(const PbhcEnvConfig* cfg, const PbhcMotionTable* tbl, double* globals, PbhcEnv** out) {
  ARG_CHECK(cfg && tbl && globals && out);
  PbhcEnv* e = new (std::nothrow) PbhcEnv();
  if (!e) return PBHC_ENOMEM;
  int rc = finalize_config(cfg, &e->cfg, &e->lds_stride, &e->lds_bytes);
  if (rc == PBHC_OK && !(tbl->row == 2 * cfg->skel.num_dof + 2 + 13 * cfg->skel.num_bodies_ext && tbl->frames && tbl->num_motions >= 1)) {
    snprintf(g_err, sizeof(g_err), "pbhc_env_create: motion table does not match the skeleton (row %d)", tbl->row);
    rc = PBHC_EINVAL;
  }
  if (rc != PBHC_OK) { delete e; return rc; }
  const int Bx = cfg->skel.num_bodies_ext;
  e->tbl = *tbl;
  e->d_glob = globals;
  e->nblocks = (cfg->num_envs + PBHC_EPB - 1) / PBHC_EPB;
  e->step_ctr = 0;
  e->profile = 0;
  e->prof_count = 0;
  e->spec_fn = nullptr;
  e->spec_dl = nullptr;
  e->spec_lds_stride = 0;
  e->spec_lds_bytes = 0;
  e->spec_hist_wide = 0;
  if (hipMalloc(&e->d_cfg, sizeof(PbhcEnvConfig)) != hipSuccess) { delete e; return PBHC_ENOMEM; }
  if (hipMalloc(&e->d_partials, (size_t)e->nblocks * 2 * PBHC_NP * sizeof(float)) != hipSuccess) { (void)hipFree(e->d_cfg); delete e; return PBHC_ENOMEM; }
  if (hipMalloc(&e->d_skc, SKC_WORDS * sizeof(float)) != hipSuccess) { (void)hipFree(e->d_cfg); (void)hipFree(e->d_partials); delete e; return PBHC_ENOMEM; }
  {
    float img[SKC_WORDS];
    for (int i = 0; i < SKC_WORDS; ++i) img[i] = i < Bx * SKC_W ? skel_word(cfg->skel, i) : 0.0f;
    HIP_CHECK(hipMemcpy(e->d_skc, img, sizeof(img), hipMemcpyHostToDevice));
  }
  if (hipMalloc(&e->d_skj, PBHC_MAX_BODIES * SKJ_W * sizeof(float)) != hipSuccess) { (void)hipFree(e->d_cfg); (void)hipFree(e->d_partials); (void)hipFree(e->d_skc); delete e; return PBHC_ENOMEM; }
  {
    float img[PBHC_MAX_BODIES * SKJ_W];
    for (int i = 0; i < PBHC_MAX_BODIES * SKJ_W; ++i) img[i] = i < Bx * SKJ_W ? skj_word(cfg->skel, i) : 0.0f;
    HIP_CHECK(hipMemcpy(e->d_skj, img, sizeof(img), hipMemcpyHostToDevice));
  }
  HIP_CHECK(hipMemcpy(e->d_cfg, &e->cfg, sizeof(PbhcEnvConfig), hipMemcpyHostToDevice));
  if (e->lds_bytes > 64 * 1024)
    HIP_CHECK(hipFuncSetAttribute(cfg->tracking_mode ? (const void*)k_env_step<1> : (const void*)k_env_step<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes));
  *out = e;
  return PBHC_OK;
}

// the config as pbhc_env_create finalised it (term_src, sum_col_term, map_lds_words): what a specialised kernel is generated from
int pbhc_env_get_config(PbhcEnv* e, PbhcEnvConfig* out) {
  ARG_CHECK(e && out);
  *out = e->cfg;
  return PBHC_OK;
}

// config without the members k_env_step reads from the run-time copy even when specialised (pointers, env count, seed, reference yaw)
static void strip_runtime_members(PbhcEnvConfig* c) {
  c->num_envs = 0; c->seed = 0; c->ref_init_yaw = 0.0f; c->map_image = nullptr;
  for (int g = 0; g < PBHC_MAX_GROUPS; ++g) { c->groups[g].dst = nullptr; c->groups[g].src = nullptr; c->groups[g].scale = nullptr; c->groups[g].noise = nullptr; }
}

// Attach a config-specialised build of k_env_step (csrc/pbhc_env_step_spec.hip compiled with this env's config as constants).  The
// object is accepted only if the config baked into it equals this env's, member by member; afterwards pbhc_env_step launches it
// instead of the generic kernel.  `so_path` = NULL detaches.
int pbhc_env_attach_specialised(PbhcEnv* e, const char* so_path) {
  ARG_CHECK(e);
  if (!so_path) { e->spec_fn = nullptr; return PBHC_OK; }          // (the object stays loaded: another env may share it)
  void* dl = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) { snprintf(g_err, sizeof(g_err), "pbhc_env_attach_specialised: dlopen(%s): %s", so_path, dlerror()); return PBHC_EINVAL; }
  typedef int (*fn_i)(void);
  typedef const PbhcEnvConfig* (*fn_c)(void);
  typedef const void* (*fn_k)(void);
  fn_i abi = (fn_i)dlsym(dl, "pbhc_spec_abi_version"), mode = (fn_i)dlsym(dl, "pbhc_spec_mode");
  fn_c cfgf = (fn_c)dlsym(dl, "pbhc_spec_config");
  fn_k kern = (fn_k)dlsym(dl, "pbhc_spec_kernel");
  fn_i lds_stride = (fn_i)dlsym(dl, "pbhc_spec_lds_stride"), lds_bytes = (fn_i)dlsym(dl, "pbhc_spec_lds_bytes"), hist_wide = (fn_i)dlsym(dl, "pbhc_spec_hist_wide");
  if (!abi || !mode || !cfgf || !kern || !lds_stride || !lds_bytes || !hist_wide || abi() != PBHC_ABI_VERSION) {
    dlclose(dl);
    snprintf(g_err, sizeof(g_err), "pbhc_env_attach_specialised: %s is not a specialised step kernel of this ABI version", so_path);
    return PBHC_EINVAL;
  }
  PbhcEnvConfig* a = new (std::nothrow) PbhcEnvConfig;
  PbhcEnvConfig* b = new (std::nothrow) PbhcEnvConfig;
  if (!a || !b) { delete a; delete b; dlclose(dl); return PBHC_ENOMEM; }
  memcpy(a, cfgf(), sizeof(PbhcEnvConfig));
  memcpy(b, &e->cfg, sizeof(PbhcEnvConfig));
  strip_runtime_members(a); strip_runtime_members(b);
  const bool same = mode() == e->cfg.tracking_mode && memcmp(a, b, sizeof(PbhcEnvConfig)) == 0;
  delete a; delete b;
  if (!same) {
    dlclose(dl);
    snprintf(g_err, sizeof(g_err), "pbhc_env_attach_specialised: %s was built from a different config", so_path);
    return PBHC_EINVAL;
  }
  const void* fn = kern();
  e->spec_lds_stride = lds_stride();
  e->spec_lds_bytes = (size_t)lds_bytes();
  e->spec_hist_wide = hist_wide();
  if (e->spec_lds_bytes > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->spec_lds_bytes));
  e->spec_fn = fn;
  e->spec_dl = dl;
  return PBHC_OK;
}
int pbhc_env_is_specialised(PbhcEnv* e) { return e && e->spec_fn ? 1 : 0; }

void pbhc_env_destroy(PbhcEnv* e) {
  if (!e) return;
  if (e->profile)
    for (int i = 0; i < PBHC_PROFILE_RING; ++i) { (void)hipEventDestroy(e->ev0[i]); (void)hipEventDestroy(e->ev1[i]); }
  (void)hipFree(e->d_cfg);
  (void)hipFree(e->d_partials);
  (void)hipFree(e->d_skc);
  (void)hipFree(e->d_skj);
  delete e;
}

int pbhc_env_profile(PbhcEnv* e, int enable) {
  ARG_CHECK(e);
  if (enable && !e->profile) {
    for (int i = 0; i < PBHC_PROFILE_RING; ++i) { HIP_CHECK(hipEventCreate(&e->ev0[i])); HIP_CHECK(hipEventCreate(&e->ev1[i])); }
    e->profile = 1;
  }
  if (!enable && e->profile) {
    for (int i = 0; i < PBHC_PROFILE_RING; ++i) { (void)hipEventDestroy(e->ev0[i]); (void)hipEventDestroy(e->ev1[i]); }
    e->profile = 0;
  }
  e->prof_count = 0;
  return PBHC_OK;
}

// One wave that spins for `ticks` of the 100 MHz wall clock: a kernel of KNOWN duration, to calibrate the dispatch-attached event pair
__global__ void k_profile_spin(long long ticks, long long* sink) {
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();          // bounded: the wall clock always advances
  if (sink && threadIdx.x == 0) *sink = t - t0;
}
int pbhc_env_profile_overhead(PbhcEnv* e, void* stream, float* overhead_ms) {
  ARG_CHECK(e && overhead_ms && e->profile);
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  float v[33];
  const long long ticks = 2000;                             // 20 us
  for (int i = 0; i < 33; ++i) {
    hipExtLaunchKernelGGL(k_profile_spin, dim3(1), dim3(64), 0, st, a, b, 0, ticks, (long long*)nullptr);
    HIP_CHECK(hipEventSynchronize(b));
    HIP_CHECK(hipEventElapsedTime(&v[i], a, b));
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  for (int i = 1; i < 33; ++i) { float x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }
  *overhead_ms = v[16] - 0.020f;                            // median reading minus the spin's own 20 us
  return PBHC_OK;
}
int pbhc_env_profile_read(PbhcEnv* e, float* ms_out, int max_count, int* count) {
  ARG_CHECK(e && ms_out && count && e->profile);
  int n = e->prof_count < PBHC_PROFILE_RING ? e->prof_count : PBHC_PROFILE_RING;
  if (n > max_count) n = max_count;
  for (int i = 0; i < n; ++i) {
    int slot = (e->prof_count - n + i) % PBHC_PROFILE_RING;
    HIP_CHECK(hipEventSynchronize(e->ev1[slot]));
    HIP_CHECK(hipEventElapsedTime(&ms_out[i], e->ev0[slot], e->ev1[slot]));
  }
  *count = n;
  return PBHC_OK;
}

static int env_step_check(PbhcEnv* e, const PbhcStepIO* io) {
  ARG_CHECK(e && io);
  ARG_CHECK(io->actions_in && io->frame_root && io->frame_dof_pos && io->frame_dof_vel && io->frame_contact && io->frame_cursor && io->num_frames >= 1);
  ARG_CHECK(io->root_states && io->dof_state && io->actions && io->last_actions && io->actions_after_delay && io->action_queue);
  ARG_CHECK(io->last_dof_pos && io->last_dof_vel && io->torques && io->feet_air_time && io->contacts && io->contacts_filt);
  ARG_CHECK(io->last_contacts && io->last_contacts_filt && io->kp_scale && io->kd_scale && io->rfi_lim_scale && io->rao_scale);
  ARG_CHECK(io->motion_start_times && io->motion_len && io->end_time_ratio_buf && io->episode_sums && io->hist);
  ARG_CHECK(io->episode_length_buf && io->last_episode_length_buf && io->reset_buf && io->action_delay_idx && io->motion_ids && io->time_out_buf);
  ARG_CHECK(io->env_origins && io->dr_base_com && io->dr_link_mass && io->dr_friction && io->rew_buf);
  for (int g = 0; g < e->cfg.num_groups; ++g) {
    ARG_CHECK(io->obs[g] != nullptr && (io->obs_pitch[g] == 0 || io->obs_pitch[g] >= e->cfg.groups[g].pitch));
    ARG_CHECK((uint64_t)e->cfg.num_envs * (uint64_t)(io->obs_pitch[g] ? io->obs_pitch[g] : e->cfg.groups[g].pitch) < (1ull << 30));   // 32-bit row offsets in the kernel
  }
  ARG_CHECK(io->hist_pitch == 0 || io->hist_pitch >= e->cfg.hist_dim);
  return PBHC_OK;
}

// first half of a step: the fused launch (per-env work + per-workgroup partial sums)
int pbhc_env_step_launch(PbhcEnv* e, const PbhcStepIO* io, void* stream) {
  const int rc = env_step_check(e, io);
  if (rc != PBHC_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int slot = e->prof_count % PBHC_PROFILE_RING;
  // profiling: the event pair is attached to the dispatch itself (hipExtLaunchKernelGGL: start / stop taken from the kernel's own
  // begin / end timestamps), so the reading is the kernel's execution time, without the dispatch gap a hipEventRecord pair would add
  hipEvent_t pe0 = e->profile ? e->ev0[slot] : nullptr, pe1 = e->profile ? e->ev1[slot] : nullptr;
  // (profiling off: the plain launch calls — what a stream capture of the whole rollout records; the dispatch-attached event pair of the
  // hipExt forms exists for bench.py's per-launch timing only)
  const PbhcEnvConfig* a_cfg = e->d_cfg;
  const double* a_glob = e->d_glob;
  const float* a_skc = e->d_skc;
  const uint32_t* a_map = e->cfg.map_image;
  const float* a_skj = e->d_skj;
  if (e->spec_fn && e->spec_hist_wide) {
    const int hp = io->hist_pitch ? io->hist_pitch : e->cfg.hist_dim;
    if (((uintptr_t)io->hist & 15) != 0 || (hp & 3) != 0 || hp < ((e->cfg.hist_dim + 3) & ~3)) {
      snprintf(g_err, sizeof(g_err), "pbhc_env_step: the specialised kernel reads history rows 16 bytes per lane: io.hist must be 16-byte aligned with a pitch that is a "
                                     "multiple of 4 floats >= hist_dim rounded up to 4 (pad the rows, or detach the specialised kernel)");
      return PBHC_EINVAL;
    }
  }
  PbhcStepIO a_io = *io;
  a_io.obs_wide = 1;
  for (int g = 0; g < e->cfg.num_groups; ++g) {
    const int pitch = io->obs_pitch[g] ? io->obs_pitch[g] : e->cfg.groups[g].pitch;
    if (((uintptr_t)io->obs[g] & 15) != 0 || (pitch & 3) != 0 || pitch < ((e->cfg.groups[g].dim + 3) & ~3)) a_io.obs_wide = 0;
  }
  int a_stride = e->spec_fn ? e->spec_lds_stride : e->lds_stride;
  const size_t a_lds = e->spec_fn ? e->spec_lds_bytes : e->lds_bytes;
  const long long* a_ep = (const long long*)io->episode_length_buf;
  const float* a_st = io->motion_start_times;
  const float *a_fr = io->frame_root, *a_fq = io->frame_dof_pos, *a_fqd = io->frame_dof_vel;
  const int32_t* a_cur = io->frame_cursor;
  int a_fi = io->frame_index, a_n = e->cfg.num_envs;
  void* args[] = {(void*)&a_ep, (void*)&a_st, (void*)&a_fr, (void*)&a_fq, (void*)&a_fqd, (void*)&a_cur, (void*)&a_fi, (void*)&a_n, (void*)&a_cfg, (void*)&e->tbl, (void*)&a_io, (void*)&a_glob, (void*)&e->d_partials, (void*)&a_stride, (void*)&a_skc, (void*)&a_map, (void*)&a_skj};
  const void* fn = e->spec_fn ? e->spec_fn : (e->cfg.tracking_mode ? (const void*)k_env_step<1> : (const void*)k_env_step<0>);
  if (e->profile) HIP_CHECK(hipExtLaunchKernel(fn, dim3(e->nblocks), dim3(PBHC_TPB), args, a_lds, st, pe0, pe1, 0));
  else HIP_CHECK(hipLaunchKernel(fn, dim3(e->nblocks), dim3(PBHC_TPB), args, a_lds, st));
  if (e->profile) e->prof_count++;
  e->step_ctr++;
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

// second half: the one-workgroup reduction of the partial sums into the globals (adaptive sigma, curricula, log means, step counter) — or,
// with io->totals_out, into this shard's totals for the ranks' exchange.  `stream` may differ from the launch's: the caller orders it
// after the fused launch and before the next one (the rollout runs it next to the policy forward, off the step -> policy -> step chain).
int pbhc_env_step_finish(PbhcEnv* e, const PbhcStepIO* io, void* stream) {
  ARG_CHECK(e && io && io->frame_cursor && io->num_frames >= 1);
  hipLaunchKernelGGL(k_env_finalize, dim3(1), dim3(64 * PBHC_FIN_CHUNKS), 0, (hipStream_t)stream, e->d_cfg, e->d_glob, e->d_partials, 2 * e->nblocks, io->frame_cursor,
                     io->num_frames, (const double*)nullptr, io->totals_out, 0.0);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_env_step(PbhcEnv* e, const PbhcStepIO* io, void* stream) {
  const int rc = pbhc_env_step_launch(e, io, stream);
  return rc != PBHC_OK ? rc : pbhc_env_step_finish(e, io, stream);
}

int pbhc_env_finalize(PbhcEnv* e, const double* totals, double num_envs_total, void* stream) {
  ARG_CHECK(e && totals && num_envs_total >= 1.0);
  hipLaunchKernelGGL(k_env_finalize, dim3(1), dim3(64 * PBHC_FIN_CHUNKS), 0, (hipStream_t)stream, e->d_cfg, e->d_glob, (const float*)nullptr, 0, (int32_t*)nullptr, 1,
                     totals, (double*)nullptr, num_envs_total);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, int T, int N, int R, float gamma,
             float lam, float* returns, float* advantages, double* stats, void* stream) {
  ARG_CHECK(rewards && values && dones && last_values && returns && advantages && stats && T >= 1 && N >= 1 && R >= 1);
  hipStream_t st = (hipStream_t)stream;
  size_t NR = (size_t)N * R, TN = (size_t)T * N;
  int nb = (int)((TN + GAE_TPB - 1) / GAE_TPB);
  hipLaunchKernelGGL(k_gae, dim3((unsigned)((NR + GAE_TPB - 1) / GAE_TPB)), dim3(GAE_TPB), 0, st, rewards, values, dones, last_values, T, N, R, gamma, lam, returns);
  hipLaunchKernelGGL(k_adv_sum, dim3(nb), dim3(GAE_TPB), 0, st, returns, values, TN, R, advantages, stats);
  hipLaunchKernelGGL(k_adv_norm, dim3(nb), dim3(GAE_TPB), 0, st, advantages, TN, stats, nb);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

}  // extern "C"
