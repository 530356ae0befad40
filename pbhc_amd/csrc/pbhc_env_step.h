// pbhc_env_step.h — the fused env-step kernel (k_env_step) and the device helpers it shares with the stand-alone kernels of
// pbhc_kernels.hip (rigid-body FK, reference-motion lookup).  Included by pbhc_kernels.hip (the generic kernel: every config scalar read
// from the run-time PbhcEnvConfig) and by pbhc_env_step_spec.hip (a config-specialised build of the same source: -DPBHC_STATIC_CFG=<header
// with a constexpr copy of the env's config>, compiled when an env asks for it — pbhc_amd/specialise.py — and attached with
// pbhc_env_attach_specialised).  Expects <hip/hip_runtime.h>, include/pbhc_hip.h, pbhc_math.h and `using namespace pbhc` before it.
#pragma once

#ifndef PBHC_G
#define PBHC_G 32     // lanes per env
#endif
#ifndef PBHC_EPB
#define PBHC_EPB 4    // envs per workgroup
#endif
#define PBHC_NP 64    // partial sums per workgroup

#ifdef PBHC_STATIC_CFG
#include PBHC_STATIC_CFG
#endif

// Diagnostic build only (-DPBHC_STAMPS, libpbhc_hip_stamps.so): shader-clock stamps of workgroup 0 at the phase
// boundaries of k_env_step, written to a buffer nothing else reads.  The product build contains none of this.
#ifdef PBHC_STAMPS
__device__ unsigned long long g_stamps[64];                  // [0,32): role A (thread 0 of workgroup 0), [32,64): role B (thread 128)
__device__ unsigned long long g_wg_times[2 * 4096];          // [workgroup][entry, exit] on the constant 100 MHz clock (comparable across CUs)
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = clock64(); } while (0)
#define STAMPB(i) do { if (threadIdx.x == 128 && blockIdx.x == 0) g_stamps[32 + (i)] = clock64(); } while (0)
#define WG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_wg_times[2 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMPB(i) do { } while (0)
#define WG_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// partial-sum columns written per workgroup by k_env_step, reduced by k_env_finalize
enum {
  P_ERR = 0,            // [PBHC_NUM_SIGMA] tracking errors (adaptive sigma)
  P_UPPER_NORM = PBHC_NUM_SIGMA, P_LOWER_NORM, P_VR_NORM, P_JOINT_NORM, P_CLIP_CNT, P_RESET_CNT, P_TERM_GRAVITY, P_TERM_FAR,
  P_TERM_TIMEOUT, P_TERM_END, P_RESET_EPLEN, P_ETR_SUM, P_ETR_SQ, P_REW_SUM,
  P_KEY_NORM, P_LUP_NORM, P_LLO_NORM, P_LVR_NORM, P_LKEY_NORM, P_TERM_REFZ, P_TERM_REFORI, P_TERM_BODYZ,      // general tracking
  P_TERM_CONTACT, P_TERM_LOWH, P_TERM_POSLIM, P_TERM_VELLIM, P_TERM_TAULIM,
  P_NUM
};
static_assert(P_NUM <= PBHC_NP, "partials");
static_assert(PBHC_NP == PBHC_NUM_TOTALS, "totals");

// reduction slots per env (LDS)
enum {
  R_ERR0 = 0,                              // [PBHC_NUM_SIGMA] the tracking errors, in sigma order (what exp(-err/sigma) and the sigma EMA read)
  R_MAXJP = R_ERR0 + PBHC_S_MAX_JOINT_POS, R_UP = R_ERR0 + PBHC_S_UPPER_BODY_POS, R_LO = R_ERR0 + PBHC_S_LOWER_BODY_POS,
  R_VR = R_ERR0 + PBHC_S_VR_3POINT_POS, R_FEET = R_ERR0 + PBHC_S_FEET_POS, R_ROT = R_ERR0 + PBHC_S_BODY_ROT, R_VEL = R_ERR0 + PBHC_S_BODY_VEL,
  R_ANG = R_ERR0 + PBHC_S_BODY_ANG_VEL, R_JPM = R_ERR0 + PBHC_S_JOINT_POS, R_JVM = R_ERR0 + PBHC_S_JOINT_VEL,
  R_KEY = R_ERR0 + PBHC_S_KEY_BODY_POS, R_APOS = R_ERR0 + PBHC_S_ANCHOR_BODY_POS, R_AROT = R_ERR0 + PBHC_S_ANCHOR_BODY_ROT,
  R_LKEY = R_ERR0 + PBHC_S_LOCAL_KEY_BODY_POS, R_LKROT = R_ERR0 + PBHC_S_LOCAL_KEY_BODY_ROT, R_KVEL = R_ERR0 + PBHC_S_KEY_BODY_VEL,
  R_KANG = R_ERR0 + PBHC_S_KEY_BODY_ANG_VEL, R_RVEL = R_ERR0 + PBHC_S_ROOT_VEL, R_RPOSE = R_ERR0 + PBHC_S_ROOT_POSE,
  R_MAXNORM = R_ERR0 + PBHC_NUM_SIGMA, R_UPN, R_LON, R_VRN, R_JP2, R_TAU2, R_ARATE, R_QD2, R_QACC2, R_LIMPOS, R_LIMVEL, R_LIMTAU, R_COLL, R_CLIPCNT,
  R_KEYN, R_LKEYN, R_LUPN, R_LLON, R_LVRN, R_BODYZ,                 // general tracking: log norms, body_z flag
  R_EXP0,                                  // [PBHC_NUM_SIGMA] exp(-err_k / sigma_k)
  R_FOOT0 = R_EXP0 + PBHC_NUM_SIGMA,       // per foot f: +4f: |F|, |F_xy|, F_z, |v|   (+8: |v_xy| x2; +10: |heading - root heading| x2; +12: |gravity_xy| in the foot frame x2)
  R_NUM = R_FOOT0 + 14
};
static_assert(R_NUM <= 80, "RED region");

// per-env LDS layout (floats); the body arrays are sized for the robot at hand (Bx rounded up to 4)
struct Lds {
  enum {
    ACT = 0, ACTD = 32, TAU = 64, Q = 96, QD = 128, RDOF = 160, RDOFV = 192,          // 7 x 32
    ROOT = 224,                                                                      // 16
    MISC = 240,                                                                      // 56: scalars
    CF = 296,                                                                        // 108 contact forces
    BP = 404,                                                                        // body pos3/quat4/vel3/ang3 x Bxp, then the reference's, same shapes
    RED_WORDS = 80, FUT_WORDS = 10 * PBHC_MAX_FUTURE                                 // reductions; general tracking: per-step future scratch
  };
  int bq = 0, bv = 0, bw = 0, rp = 0, rq = 0, rv = 0, rw = 0, red = 0, fut = 0, feat = 0;
  __host__ __device__ constexpr explicit Lds(int Bx, int mode) {
    const int p = (Bx + 3) & ~3;
    bq = BP + 3 * p; bv = bq + 4 * p; bw = bv + 3 * p;
    rp = bw + 3 * p; rq = rp + 3 * p; rv = rq + 4 * p; rw = rv + 3 * p;
    red = rw + 3 * p;
    fut = red + RED_WORDS;
    feat = fut + (mode ? FUT_WORDS : 0);
  }
};
// MISC slots
enum {
  M_HINV = 0,   // 4 heading-inverse quaternion
  M_MLEN = 4, M_START, M_RESET, M_TIMEOUT, M_EPLEN, M_CONTACT0, M_CONTACT1, M_CFILT0, M_CFILT1,
  M_RCONTACT0, M_RCONTACT1, M_GRAV, M_FAR, M_END, M_TOUT_LEN, M_LASTEP, M_NEWSTART, M_DELAY, M_FAT0, M_FAT1,
  M_LASTC0, M_LASTC1, M_GX, M_GY, M_GZ,
  M_REFZ, M_REFORI, M_BODYZ, M_ADZ, M_AORI,                     // general tracking: termination causes, anchor z / gravity-z differences
  M_CLIPCNT,                                                    // clipped actions of this step (role B -> reduction row)
  M_TCONTACT, M_TLOWH,                                          // termination causes: contact on a terminating body, low base height
  M_TPOSLIM, M_TVELLIM, M_TTAULIM, M_TGATE,                     // ... close to a joint position / velocity / torque limit (role B), any of them
  M_ORIGIN0, M_ORIGIN1, M_ORIGIN2, M_CLIP_LEN, M_CLIP_DT, M_CLIP_NF, M_CLIP_ROW0,     // env origin + clip meta (role B's prologue loads) for role A's reset path
  M_PJOINT, M_PEPLEN, M_PETR, M_PETRSQ, M_PREW, M_ZERO,                               // partial-sum columns that are computed values; a zero for the unused ones
  M_NZB                                                                               // the observation-noise base word of this env and step (dynamics -> reference waves)
};
static_assert(M_NZB < 56, "MISC region");

// Where column k of an env's partial-sum row comes from: a slot of the env's reduction row (RED) or of its scalar block (MISC).  Lane k of
// the dynamics wave copies it at the end of the step (one lane writing the ~35 columns in turn was 1.8 k cycles of every workgroup's tail).
#define PSRC_RED(slot) ((uint16_t)(slot))
#define PSRC_MISC(slot) ((uint16_t)(0x8000u | (slot)))
struct PartTab { uint16_t v[PBHC_NP]; };
constexpr PartTab make_part_tab(bool general, bool close_any) {
  PartTab t{};
  for (int k = 0; k < PBHC_NP; ++k) t.v[k] = PSRC_MISC(M_ZERO);
  for (int k = 0; k < PBHC_NUM_SIGMA; ++k) t.v[P_ERR + k] = PSRC_RED(R_ERR0 + k);
  t.v[P_UPPER_NORM] = PSRC_RED(R_UPN); t.v[P_LOWER_NORM] = PSRC_RED(R_LON); t.v[P_VR_NORM] = PSRC_RED(R_VRN);
  t.v[P_JOINT_NORM] = PSRC_MISC(M_PJOINT); t.v[P_CLIP_CNT] = PSRC_RED(R_CLIPCNT);
  t.v[P_RESET_CNT] = PSRC_MISC(M_RESET); t.v[P_TERM_GRAVITY] = PSRC_MISC(M_GRAV); t.v[P_TERM_FAR] = PSRC_MISC(M_FAR);
  t.v[P_TERM_TIMEOUT] = PSRC_MISC(M_TIMEOUT); t.v[P_TERM_END] = PSRC_MISC(M_END); t.v[P_RESET_EPLEN] = PSRC_MISC(M_PEPLEN);
  t.v[P_ETR_SUM] = PSRC_MISC(M_PETR); t.v[P_ETR_SQ] = PSRC_MISC(M_PETRSQ); t.v[P_REW_SUM] = PSRC_MISC(M_PREW);
  t.v[P_TERM_CONTACT] = PSRC_MISC(M_TCONTACT); t.v[P_TERM_LOWH] = PSRC_MISC(M_TLOWH);
  if (close_any) { t.v[P_TERM_POSLIM] = PSRC_MISC(M_TPOSLIM); t.v[P_TERM_VELLIM] = PSRC_MISC(M_TVELLIM); t.v[P_TERM_TAULIM] = PSRC_MISC(M_TTAULIM); }
  if (general) {
    t.v[P_KEY_NORM] = PSRC_RED(R_KEYN); t.v[P_LUP_NORM] = PSRC_RED(R_LUPN); t.v[P_LLO_NORM] = PSRC_RED(R_LLON); t.v[P_LVR_NORM] = PSRC_RED(R_LVRN);
    t.v[P_LKEY_NORM] = PSRC_RED(R_LKEYN); t.v[P_TERM_REFZ] = PSRC_MISC(M_REFZ); t.v[P_TERM_REFORI] = PSRC_MISC(M_REFORI); t.v[P_TERM_BODYZ] = PSRC_MISC(M_BODYZ);
  }
  return t;
}
__device__ const PartTab kPartTab[4] = {make_part_tab(false, false), make_part_tab(false, true), make_part_tab(true, false), make_part_tab(true, true)};

// Workgroup barrier that orders LDS traffic only: waits for this wave's LDS ops (lgkmcnt) and leaves global loads AND stores in
// flight (a __syncthreads() would also drain vmcnt, i.e. stall on the early fire-and-forget stores).  Waves of a workgroup share
// data through LDS only; same-address global accesses stay inside one wave, where program order holds.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// Hand-off through LDS between lanes of ONE wave (the step kernel keeps an env's lanes inside a wave): the hardware executes a wave's DS
// instructions in issue order, so no instruction is needed — but the COMPILER must not move LDS accesses across the hand-off (without this
// it forwards `if (lane == 0) x[i] = v;  ... = x[i]` per thread: the other lanes' load is hoisted above the store).
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

// Reductions over the 32 lanes of an env (half a wave64), every lane receiving the result.  `__shfl_xor` lowers to ds_bpermute_b32 — an LDS
// round trip plus an address VGPR per step, five steps per reduction, ~30 reductions per env step.  The DPP forms below stay in the VALU:
// quad_perm x2 and row_half_mirror / row_mirror fold the 16 lanes of a DPP row into every lane of it (the compiler fuses each move into
// v_add_f32_dpp / v_max_f32_dpp), and one v_permlane16_swap_b32 (gfx950) exchanges the odd rows of one copy with the even rows of the
// other, so rows {0,1} and {2,3} — the two envs of the wave — each end up holding their 32-lane result: 7 instructions, no LDS.
// Inactive lanes read as 0 (bound_ctrl), a row never mixes envs, and the swap pairs row 0 with 1 and row 2 with 3 only.
#if PBHC_G == 32
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<0xB1>(v);        // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E>(v);        // quad_perm:[2,3,0,1]
  v += dpp_mov<0x141>(v);       // row_half_mirror
  v += dpp_mov<0x140>(v);       // row_mirror
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float group_max(float v) {      // operands are >= 0 wherever this is used, so the 0 of an inactive lane is neutral
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return fmaxf(__int_as_float(r[0]), __int_as_float(r[1]));
}
#else
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = PBHC_G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, PBHC_G);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int m = PBHC_G / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, PBHC_G));
  return v;
}
#endif
// element `i` of a tensor at a UNIFORM base pointer with the byte offset formed in 32 bits: the compiler emits the SGPR-base form of
// global_load / global_store (one 32-bit offset VGPR per access instead of a 64-bit address built in the VALU).  Callers keep
// (elements x sizeof) below 2^32 (checked on the host at pbhc_env_create / pbhc_env_step).
// Outputs of the step are STREAMING stores (round 4): nothing in this launch reads them back, and as ordinary stores their 27 MB per 4096 envs
// allocate in — and wash out — the XCDs' L2 that also serves the launch's own reads: 18.0 -> 17.1 us at 4096 envs, 100 -> 89.6 us at 32 768 on
// one box (profiles/round4_k_env_step_variants.txt (j)).  -DPBHC_NO_NT_STORES: ordinary stores.
#ifndef PBHC_NO_NT_STORES
#define NTST(lhs, v) __builtin_nontemporal_store((v), &(lhs))
#else
#define NTST(lhs, v) ((lhs) = (v))
#endif
// ... and the per-env inputs (replay frame, env state, history: each read by exactly one lane of one launch) as streaming loads
// (-DPBHC_NT_LOADS, measurement: see (k) in the variants file); the motion table's rows and the constant tables stay ordinary loads — those
// are what the L2 is for
#ifdef PBHC_NT_LOADS
#define NTLD(expr) __builtin_nontemporal_load(&(expr))
#else
#define NTLD(expr) (expr)
#endif
typedef float pbhc_f32x4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T& at(T* p, unsigned int i) { return *(T*)((char*)p + (size_t)(unsigned int)(i * (unsigned int)sizeof(T))); }
template <class T> __device__ __forceinline__ const T& at(const T* p, unsigned int i) { return *(const T*)((const char*)p + (size_t)(unsigned int)(i * (unsigned int)sizeof(T))); }
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ f4 ld4(const float* p) { return mk4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void st3(float* p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ void st4(float* p, f4 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// global -> LDS copy by one env group, 8 independent loads in flight per lane before the first store
__device__ __forceinline__ void copy_g2l(float* dst, const float* __restrict__ src, int n, int lane) {
  for (int i0 = lane; i0 < n; i0 += 8 * PBHC_G) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * PBHC_G; v[u] = i < n ? src[i] : 0.0f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * PBHC_G; if (i < n) dst[i] = v[u]; }
  }
}

// ---- skeleton constants staged once per workgroup in LDS (shared by its envs) -------------------
// per body b: off[3] lq_xyzw[4] axis[3] chain_len chain[PBHC_MAX_DEPTH]
#define SKC_W (11 + PBHC_MAX_DEPTH)
#define SKC_WORDS (PBHC_MAX_BODIES * SKC_W)
__host__ __device__ __forceinline__ float skel_word(const PbhcSkeleton& sk, int i) {
  int b = i / SKC_W, w = i - b * SKC_W;
  if (w < 3) return sk.offset[b][w];
  if (w < 7) return sk.local_rot_wxyz[b][(w - 3 + 1) & 3];                 // wxyz -> xyzw
  if (w < 10) {                                                              // joint axis, normalised (rotations.py:138-145 normalises it per call)
    if (!(b >= 1 && b < sk.num_bodies)) return 0.0f;
    const float* ax = sk.dof_axis[b - 1];
    const float nrm = fmaxf(sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]), 1e-9f);
    return ax[w - 7] / nrm;
  }
  int iv = (w == 10) ? sk.chain_len[b] : sk.chain[b][w - 11];
  float fv;
  memcpy(&fv, &iv, sizeof(fv));
  return fv;
}
__device__ __forceinline__ void stage_skeleton(const PbhcSkeleton& sk, float* skc) {
  const int n = sk.num_bodies_ext * SKC_W;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x; v[u] = i < n ? skel_word(sk, i) : 0.0f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + u * blockDim.x; if (i < n) skc[i] = v[u]; }
  }
}
// the same in two halves for the step kernel (128 threads, <= 8 words each): loads first, LDS stores after the other prologue loads are issued
#define SKC_REGS ((SKC_WORDS + PBHC_G * PBHC_EPB - 1) / (PBHC_G * PBHC_EPB))
// `img`: the SKC image built on the host at pbhc_env_create (plain coalesced loads; building it word by word from the struct in the kernel
// cost one dependent memory round trip per register)
__device__ __forceinline__ void stage_skeleton_load(const float* __restrict__ img, int n, float* v) {
#pragma unroll
  for (int u = 0; u < SKC_REGS; ++u) { const int i = threadIdx.x + u * (PBHC_G * PBHC_EPB); v[u] = i < n ? img[i] : 0.0f; }
}
__device__ __forceinline__ void stage_skeleton_store(int num_bodies_ext, float* skc, const float* v) {
  const int n = num_bodies_ext * SKC_W;
#pragma unroll
  for (int u = 0; u < SKC_REGS; ++u) { const int i = threadIdx.x + u * (PBHC_G * PBHC_EPB); if (i < n) skc[i] = v[u]; }
}

// ---- rigid-body pose + twist of every body (what Isaac Gym's rigid-body state tensor held in the
// reference, isaacgym.py:574-605).  Same chain as forward_kinematics_batch
// (torch_humanoid_batch.py:248-252) with pose_aa = axis*q; twist propagated analytically:
// w_i = w_par + R_i axis_i qd_i, v_i = v_par + w_par x (p_i - p_par).
// Step 1: lane b composes relq[b] = q_local[b] * q_joint(q[b-1]) (one sincos per lane).
// Step 2: lane b walks ITS OWN root->b chain from LDS — no barriers inside the chain.
// Extended bodies (motion_tracking.py:619-643) walk to their parent, then apply
// p = R_ext(R_par off) + p_par, q = q_par*q_ext, w = w_par, v = v_par + w_par x off (offset NOT rotated, sic).
// All threads of the workgroup must call it (two barriers).
__device__ __forceinline__ void fk_walk(const float* skc, int B, int Bx, int lane, bool valid, const float* root, const float* q, const float* qd,
                                        float* relq, float* bp, float* bq, float* bv, float* bw) {
  if (valid)
    for (int b = 1 + lane; b < B; b += PBHC_G) {
      const float* k = skc + b * SKC_W;
      st4(relq + 4 * b, quat_mul(ld4(k + 3), quat_from_angle_unit_axis(q[b - 1], ld3(k + 7))));
    }
  LDS_BARRIER();
  if (valid)
    for (int b = lane; b < Bx; b += PBHC_G) {
      const float* kb = skc + b * SKC_W;
      const int n = __float_as_int(kb[10]);
      f3 p = ld3(root), v = ld3(root + 7), w = ld3(root + 10);
      // the chain starts from the UNIT root rotation (a replay frame's quaternion may be off unit length by the reference slerp's scale
      // error, up to 4e-4: a rotation is a rotation); the root body itself keeps the frame's quaternion as it is
      const f4 rraw = ld4(root + 3);
      f4 r = quat_unit_fast(rraw);
      for (int i = 0; i < n; ++i) {
        const int a = __float_as_int(kb[11 + i]);
        const float* ka = skc + a * SKC_W;
        f3 axis = ld3(ka + 7);
        const f3 rp = fk_rotate(r, ld3(ka));
        f4 rn = fk_mul_unit(r, ld4(relq + 4 * a));
        f3 wn = fk_axpy(w, qd[a - 1], fk_rotate(rn, axis));
        v = fk_add_cross(v, w, rp);
        p = add3(p, rp); r = rn; w = wn;
      }
      if (b >= B) {
        f3 off = ld3(kb);
        f4 eq = ld4(kb + 3);
        f3 pe = add3(quat_rotate(eq, quat_rotate(r, off)), p);
        v = add3(v, cross3(w, off));
        r = quat_mul(r, eq);
        p = pe;
      }
      if (b == 0) r = rraw;
      st3(bp + 3 * b, p); st4(bq + 4 * b, r); st3(bv + 3 * b, v); st3(bw + 3 * b, w);
    }
  LDS_BARRIER();
}

// The same for ONE wave (the step kernel, where the 32 lanes of an env belong to one wave): no workgroup barrier.  A wave's DS instructions
// execute in issue order, so the relative joint quaternions written in step 1 are visible to every lane's chain walk, and they can live in
// `bq` itself: the world quaternions are stored only after EVERY walk of the wave has read its last relative one (up to two bodies per lane,
// results held in registers until then).
__device__ __forceinline__ void fk_walk_wave(const float* skc, int B, int Bx, int lane, bool valid, const float* root, const float* q, const float* qd,
                                             float* bp, float* bq, float* bv, float* bw) {
  static_assert(PBHC_MAX_BODIES <= 2 * PBHC_G, "two bodies per lane");
  float* relq = bq;
  if (valid)
    for (int b = 1 + lane; b < B; b += PBHC_G) {
      const float* k = skc + b * SKC_W;
      st4(relq + 4 * b, quat_mul(ld4(k + 3), quat_from_angle_unit_axis(q[b - 1], ld3(k + 7))));       // (the image's axes are unit vectors: skel_word)
    }
  WAVE_LDS_FENCE();
  STAMP(20);
  if (!valid) return;
  f3 P[2], V[2], W[2];
  f4 R[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int b = lane + it * PBHC_G;
    P[it] = mk3(0, 0, 0); V[it] = P[it]; W[it] = P[it]; R[it] = mk4(0, 0, 0, 1);
    if (b < Bx) {
      const float* kb = skc + b * SKC_W;
      const int n = __float_as_int(kb[10]);
      f3 p = ld3(root), v = ld3(root + 7), w = ld3(root + 10);
      const f4 rraw = ld4(root + 3);
      f4 r = quat_unit_fast(rraw);                                 // (unit root rotation for the chain: see fk_walk)
      // The walk is a chain of dependent levels, each needing the joint's constants (an index read, then reads addressed by it): software
      // pipelined by hand — the index two levels ahead and the constants one level ahead are requested before this level's arithmetic, so
      // both LDS round trips run under ~500 cycles of VALU work instead of in front of it.  Reads past the chain's end repeat the last joint.
      int a1 = __float_as_int(kb[11 + min(1, max(n - 1, 0))]);
      int a0 = __float_as_int(kb[11]);
      const float* k0 = skc + a0 * SKC_W;
      f3 off_c = ld3(k0), ax_c = ld3(k0 + 7);
      f4 rq_c = ld4(relq + 4 * a0);
      float qd_c = qd[max(a0 - 1, 0)];
      for (int i = 0; i < n; ++i) {
        const int a2 = __float_as_int(kb[11 + min(i + 2, n - 1)]);
        const float* k1 = skc + a1 * SKC_W;
        const f3 off_n = ld3(k1), ax_n = ld3(k1 + 7);
        const f4 rq_n = ld4(relq + 4 * a1);
        const float qd_n = qd[max(a1 - 1, 0)];
        WAVE_LDS_FENCE();                                        // (compiler fence: keep the requests above this level's arithmetic)
        const f3 rp = fk_rotate(r, off_c);                      // (pbhc_math.h: the chain's arithmetic is fused on purpose)
        const f4 rn = fk_mul_unit(r, rq_c);
        const f3 wn = fk_axpy(w, qd_c, fk_rotate(rn, ax_c));
        v = fk_add_cross(v, w, rp);
        p = add3(p, rp); r = rn; w = wn;
        off_c = off_n; ax_c = ax_n; rq_c = rq_n; qd_c = qd_n; a1 = a2;
      }
      if (b >= B) {
        f3 off = ld3(kb);
        f4 eq = ld4(kb + 3);
        f3 pe = add3(quat_rotate(eq, quat_rotate(r, off)), p);
        v = add3(v, cross3(w, off));
        r = quat_mul(r, eq);
        p = pe;
      }
      if (b == 0) r = rraw;
      P[it] = p; V[it] = v; W[it] = w; R[it] = r;
    }
  }
  WAVE_LDS_FENCE();
  STAMP(21);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int b = lane + it * PBHC_G;
    if (b < Bx) { st3(bp + 3 * b, P[it]); st4(bq + 4 * b, R[it]); st3(bv + 3 * b, V[it]); st3(bw + 3 * b, W[it]); }
  }
  STAMP(22);
}

// ---- the same rigid-body state by POINTER JUMPING (round 4; config-specialised builds of robots with <= 32 bodies incl. extended).
// The walk above is the longest dependent stretch of the step: every lane composes its body's whole root -> body chain, depth-of-the-deepest-
// body levels of ~130 instructions each (7 for the 23-DoF G1: ~5.5 k cycles of a 21 k-cycle workgroup).  Rigid transforms with twists
// compose associatively, so the chain is evaluated as a prefix product instead: lane b starts with the segment parent(b) -> b
//     R = q_local (x) q_joint(q),  p = offset,  w = R(q_local) axis * q-dot,  v = 0          (pose / twist of b relative to, and in the frame of, its parent;
//                                                                                           body 0: the root state relative to the world)
// and in round r = 0, 1, 2, ... replaces it by  S(anc) o S(b)  where anc is b's ancestor 2^r levels up, whose segment — fetched from ITS
// lane's registers with ds_bpermute_b32 (the LDS crossbar, no LDS memory, no barrier) — reaches 2^r levels further:
//     R = R1 R2,  p = p1 + R1 p2,  w = w1 + R1 w2,  v = v1 + w1 x (R1 p2) + R1 v2 .
// After ceil(log2(depth + 1)) rounds (3 for the 23-DoF G1, 4 for 29 DoF) every segment starts at the world: 3 x ~100 instructions instead of
// 7 x ~130, same arithmetic as the walk up to the association order (the stand-alone FK kernel keeps the walk: the two agree to 2e-6,
// tests/test_gpu_parity.py: lazy vs stored rigid-body state).  Extended bodies take their parent's finished state afterwards, with the
// reference's own formula (see fk_walk).  The per-body constants come from a 20-float row of `skj` per lane, straight into registers:
// no skeleton image in LDS, no bar0.
//   skj row b: [off.xyz, anc1 | q_local.xyzw | axis.xyz, anc2 | u.xyz (= R(q_local) axis), anc4 | anc8, ext_parent (-1: a real body), -, -]
#define SKJ_W 20
__host__ __device__ inline float skj_word(const PbhcSkeleton& sk, int i) {
  const int b = i / SKJ_W, w = i - b * SKJ_W;
  const bool ext = b >= sk.num_bodies;
  auto anc = [&](int dist) -> int {                            // ancestor `dist` levels up, -1 if b is less than `dist` below the root (or extended)
    if (ext || sk.depth[b] < dist) return -1;
    int a = b;
    for (int k = 0; k < dist; ++k) a = sk.parent[a];
    return a;
  };
  auto asf = [](int v) { float f; memcpy(&f, &v, sizeof(f)); return f; };
  float ax[3] = {0.0f, 0.0f, 0.0f};
  if (b >= 1 && !ext) {
    const float* a = sk.dof_axis[b - 1];
    const float nrm = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-9f);
    for (int k = 0; k < 3; ++k) ax[k] = a[k] / nrm;
  }
  const float* lw = sk.local_rot_wxyz[b];
  const float qx = lw[1], qy = lw[2], qz = lw[3], qw = lw[0];
  if (w < 3) return sk.offset[b][w];
  if (w == 3) return asf(anc(1));
  if (w < 8) return w == 4 ? qx : w == 5 ? qy : w == 6 ? qz : qw;
  if (w < 11) return ax[w - 8];
  if (w == 11) return asf(anc(2));
  if (w < 15) {                                                 // u = R(q_local) axis (double precision on the host)
    const double x = qx, y = qy, z = qz, ww = qw, vx = ax[0], vy = ax[1], vz = ax[2];
    const double tx = 2.0 * (y * vz - z * vy), ty = 2.0 * (z * vx - x * vz), tz = 2.0 * (x * vy - y * vx);
    const double u[3] = {vx + ww * tx + (y * tz - z * ty), vy + ww * ty + (z * tx - x * tz), vz + ww * tz + (x * ty - y * tx)};
    return (float)u[w - 12];
  }
  if (w == 15) return asf(anc(4));
  if (w == 16) return asf(anc(8));
  if (w == 17) return asf(ext ? sk.parent[b] : -1);
  return 0.0f;
}
__host__ __device__ constexpr bool skel_fk_jump(int num_bodies_ext, int max_depth) {
#ifdef PBHC_FK_WALK           // (measurement aid: the chain walk)
  return false;
#else
  return num_bodies_ext <= PBHC_G && max_depth + 1 <= 16;
#endif
}
constexpr bool cfg_fk_jump(const PbhcEnvConfig& c) { return skel_fk_jump(c.skel.num_bodies_ext, c.skel.max_depth); }
__host__ __device__ constexpr int skel_fk_rounds(int max_depth) { return max_depth + 1 <= 2 ? 1 : max_depth + 1 <= 4 ? 2 : max_depth + 1 <= 8 ? 3 : 4; }
__device__ __forceinline__ float fk_lane_get(float v, int byte_addr) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v))); }
#pragma clang fp contract(fast)
// kr: this lane's skj row (5 x float4), rounds = ceil(log2(max_depth + 1)) <= 4 (a literal in specialised builds); every lane of the wave
// calls it (bpermute reads inactive lanes as 0)
__device__ __forceinline__ void fk_jump_wave(const float4 (&kr)[5], int rounds, int B, int Bx, int lane, bool valid, const float* root, const float* q, const float* qd,
                                             float* bp, float* bq, float* bv, float* bw) {
  const int b = lane;
  const int half = (int)(threadIdx.x & 32u);                   // this env's 32 lanes inside the wave
  const bool real = valid && b < B, extb = valid && b >= B && b < Bx;
  f4 R = mk4(0.0f, 0.0f, 0.0f, 1.0f);
  f3 P = mk3(0.0f, 0.0f, 0.0f), W = P, V = P;
  if (real) {
    if (b == 0) { P = ld3(root); R = quat_unit_fast(ld4(root + 3)); V = ld3(root + 7); W = ld3(root + 10); }     // (unit root rotation for the chain: see fk_walk)
    else {
      const float qj = q[b - 1], qdj = qd[b - 1];
      R = fk_mul(mk4(kr[1].x, kr[1].y, kr[1].z, kr[1].w), quat_from_angle_unit_axis(qj, mk3(kr[2].x, kr[2].y, kr[2].z)));
      P = mk3(kr[0].x, kr[0].y, kr[0].z);
      W = mk3(kr[3].x * qdj, kr[3].y * qdj, kr[3].z * qdj);
    }
  }
  STAMP(20);
  const int ancs[4] = {__float_as_int(kr[0].w), __float_as_int(kr[2].w), __float_as_int(kr[3].w), __float_as_int(kr[4].x)};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= rounds) break;
    const int a = ancs[r];
    const bool act = real && a >= 0;
    const int addr = (half + max(a, 0)) << 2;
    const f4 R1 = mk4(fk_lane_get(R.x, addr), fk_lane_get(R.y, addr), fk_lane_get(R.z, addr), fk_lane_get(R.w, addr));
    const f3 P1 = mk3(fk_lane_get(P.x, addr), fk_lane_get(P.y, addr), fk_lane_get(P.z, addr));
    const f3 W1 = mk3(fk_lane_get(W.x, addr), fk_lane_get(W.y, addr), fk_lane_get(W.z, addr));
    const f3 V1 = mk3(fk_lane_get(V.x, addr), fk_lane_get(V.y, addr), fk_lane_get(V.z, addr));
    const fkm33 M = fk_matrix(R1);
    const f3 rp = fk_mv(M, P), rw = fk_mv(M, W), rv = fk_mv(M, V);
    const f4 Rn = fk_mul(R1, R);
    const f3 Vn = add3(fk_add_cross(V1, W1, rp), rv);
    if (act) { R = Rn; P = add3(P1, rp); W = add3(W1, rw); V = Vn; }
  }
  if (real) {
    const float n = __builtin_amdgcn_rsqf(fmaxf(R.x * R.x + R.y * R.y + R.z * R.z + R.w * R.w, 1e-18f));
    R = mk4(R.x * n, R.y * n, R.z * n, R.w * n);
  }
  STAMP(21);
  const f4 Rchain = R;                                         // (what the extended bodies hang on: the root's is the unit one)
  if (real && b == 0) R = ld4(root + 3);                       // the root body keeps the frame's quaternion as it is
  {
    // extended bodies (motion_tracking.py:619-643): the parent's finished state, then p = R_ext(R_par off) + p_par, q = q_par q_ext,
    // w = w_par, v = v_par + w_par x off (offset NOT rotated, sic) — as fk_walk
    const int ep = __float_as_int(kr[4].y);
    const int addr = (half + max(ep, 0)) << 2;
    const f4 R1 = mk4(fk_lane_get(Rchain.x, addr), fk_lane_get(Rchain.y, addr), fk_lane_get(Rchain.z, addr), fk_lane_get(Rchain.w, addr));
    const f3 P1 = mk3(fk_lane_get(P.x, addr), fk_lane_get(P.y, addr), fk_lane_get(P.z, addr));
    const f3 W1 = mk3(fk_lane_get(W.x, addr), fk_lane_get(W.y, addr), fk_lane_get(W.z, addr));
    const f3 V1 = mk3(fk_lane_get(V.x, addr), fk_lane_get(V.y, addr), fk_lane_get(V.z, addr));
    if (extb) {
      const f3 off = mk3(kr[0].x, kr[0].y, kr[0].z);
      const f4 eq = mk4(kr[1].x, kr[1].y, kr[1].z, kr[1].w);
      P = add3(quat_rotate(eq, quat_rotate(R1, off)), P1);
      V = add3(V1, cross3(W1, off));
      R = quat_mul(R1, eq);
      W = W1;
    }
  }
  if (valid && b < Bx) { st3(bp + 3 * b, P); st4(bq + 4 * b, R); st3(bv + 3 * b, V); st3(bw + 3 * b, W); }
  STAMP(22);
}
#pragma clang fp contract(off)

// ---- frame blend (motion_lib_base.py:503-513) -------------------------------------------------
__device__ __forceinline__ void frame_blend(float t, float len, int nf, float dt, int* f0, int* f1, float* blend) {
  float phase = clampf(t / len, 0.0f, 1.0f);
  if (t < 0.0f) t = 0.0f;
  int i0 = (int)(phase * (float)(nf - 1));
  int i1 = min(i0 + 1, nf - 1);
  *f0 = i0; *f1 = i1;
  *blend = clampf((t - (float)i0 * dt) / dt, 0.0f, 1.0f);
}

// ---- phase lookup of one env: lerp/slerp of the two packed frame rows into LDS
// (MotionLibBase.get_motion_state motion_lib_base.py:123-259) ------------------------------------
// clip meta (length, frames, dt, first row) by value: the step kernel has them in registers since its prologue, so a lookup on the reset
// path is ONE memory round trip (the two rows) instead of two (table meta, then rows)
__device__ __forceinline__ void motion_lookup_meta(const PbhcMotionTable& tbl, int D, int Bx, int lane, float m_len, int m_nf, float m_dt, int m_row0, float t, f3 off,
                                                   bool bodies, float* rdof, float* rdofv, float* rcontact, float* rp, float* rq, float* rv, float* rw);
__device__ __forceinline__ void motion_lookup(const PbhcMotionTable& tbl, int D, int Bx, int lane, int mid, float t, f3 off, bool bodies,
                                              float* rdof, float* rdofv, float* rcontact, float* rp, float* rq, float* rv, float* rw) {
  motion_lookup_meta(tbl, D, Bx, lane, tbl.motion_len[mid], tbl.num_frames[mid], tbl.motion_dt[mid], tbl.length_starts[mid], t, off, bodies, rdof, rdofv, rcontact,
                     rp, rq, rv, rw);
}
__device__ __forceinline__ void motion_lookup_meta(const PbhcMotionTable& tbl, int D, int Bx, int lane, float m_len, int m_nf, float m_dt, int m_row0, float t, f3 off,
                                                   bool bodies, float* rdof, float* rdofv, float* rcontact, float* rp, float* rq, float* rv, float* rw) {
  int f0, f1; float b;
  frame_blend(t, m_len, m_nf, m_dt, &f0, &f1, &b);
  const float* r0 = tbl.frames + (size_t)(m_row0 + f0) * tbl.row;
  const float* r1 = tbl.frames + (size_t)(m_row0 + f1) * tbl.row;
  float a = 1.0f - b;
  for (int d = lane; d < D; d += PBHC_G) {
    rdof[d] = a * r0[d] + b * r1[d];
    rdofv[d] = a * r0[D + d] + b * r1[D + d];
  }
  if (lane < 2) rcontact[lane] = a * r0[2 * D + lane] + b * r1[2 * D + lane];
  int o_pos = 2 * D + 2, o_rot = o_pos + 3 * Bx, o_vel = o_rot + 4 * Bx, o_ang = o_vel + 3 * Bx;
  int nb = bodies ? Bx : 1;
  for (int i = lane; i < nb; i += PBHC_G) {
    f3 p0 = ld3(r0 + o_pos + 3 * i), p1 = ld3(r1 + o_pos + 3 * i);
    st3(rp + 3 * i, mk3(a * p0.x + b * p1.x + off.x, a * p0.y + b * p1.y + off.y, a * p0.z + b * p1.z + off.z));
    st4(rq + 4 * i, slerp(ld4(r0 + o_rot + 4 * i), ld4(r1 + o_rot + 4 * i), b));
    f3 v0 = ld3(r0 + o_vel + 3 * i), v1 = ld3(r1 + o_vel + 3 * i);
    st3(rv + 3 * i, mk3(a * v0.x + b * v1.x, a * v0.y + b * v1.y, a * v0.z + b * v1.z));
    f3 w0 = ld3(r0 + o_ang + 3 * i), w1 = ld3(r1 + o_ang + 3 * i);
    st3(rw + 3 * i, mk3(a * w0.x + b * w1.x, a * w0.y + b * w1.y, a * w0.z + b * w1.z));
  }
}

// =================================================================================================
//  k_env_step: LeggedRobotBase.step (legged_robot_base.py:239-338) for LeggedRobotMotionTracking
// =================================================================================================
// One launch, 256-thread workgroups = 4 waves for PBHC_EPB = 4 envs, 32 lanes (half a wave64) per env and ROLE:
//   role A "dynamics"    (waves 0,1; envs {0,1} / {2,3}): replay frame -> rigid-body FK -> body differences + reductions -> termination ->
//                        rewards -> reset of terminated envs -> post-reset features -> state write-back.  This is the dependent chain
//                        that sets the kernel's duration.
//   role B "reference"   (waves 2,3; the same envs): everything that does not depend on the FK of the new frame — pre-physics step +
//                        torques, per-env scalars (heading, base velocities, gravity, contacts), reference-frame lookup (lerp / slerp),
//                        future reference targets (general tracking), the joint-space halves of the reductions, the optional state
//                        outputs, and the observation elements whose sources are ready before the chain ends (history: ~80 % of them).
// The round-1 kernel ran both roles back to back in ONE wave per env pair (2 waves per SIMD on the chip, ~65 k cycles per wave, of
// which the observation write-out and the load phase were 45 %); split, the chain is ~40 % shorter and a SIMD holds 4 waves.
// All LDS traffic of an env stays inside ITS two waves: within a wave the hardware executes DS instructions in order, so phases of
// one role need no barrier at all; the five workgroup barriers below are the points where the roles exchange data.
#define STATE_WRITEBACK()                                                                                                          \
      for (int dd = lane; dd < D; dd += PBHC_G) {                                                                                                     \
        NTST(at(io.actions, eD + dd), act[dd]);                                                                                                            \
        NTST(at(io.last_actions, eD + dd), act[dd]);                                                                                                       \
        NTST(at(io.actions_after_delay, eD + dd), actd[dd]);                                                                                               \
        NTST(at(io.torques, eD + dd), tau[dd]);                                                                                                            \
        NTST(at(io.dof_state, (eD + dd) * 2), q[dd]);                                                                                                      \
        NTST(at(io.dof_state, (eD + dd) * 2 + 1), qd[dd]);                                                                                                 \
        NTST(at(io.last_dof_pos, eD + dd), q[dd]);                                                                                                         \
        NTST(at(io.last_dof_vel, eD + dd), qd[dd]);                                                                                                        \
      }                                                                                                                                               \
      if (lane < 13) NTST(at(io.root_states, (u32)env * 13u + (u32)lane), root[lane]);                                                                     \
      if (lane < NF) {                                                                                                                                \
        const u32 fo = (u32)env * (u32)NF + (u32)lane;                                                                                                \
        NTST(at(io.feet_air_time, fo), misc[M_FAT0 + lane]);                                                                                               \
        NTST(at(io.contacts, fo), misc[M_CONTACT0 + lane]);                                                                                                \
        NTST(at(io.contacts_filt, fo), misc[M_CFILT0 + lane]);                                                                                             \
        NTST(at(io.last_contacts, fo), misc[M_CONTACT0 + lane]);                                                                                           \
        NTST(at(io.last_contacts_filt, fo), misc[M_CFILT0 + lane]);                                                                                        \
      }                                                                                                                                               \
      if (lane == 0) {                                                                                                                                \
        NTST(io.episode_length_buf[env], (long long)(misc[M_EPLEN]));                                                                                        \
        NTST(io.last_episode_length_buf[env], (long long)(misc[M_LASTEP]));                                                                                  \
        NTST(io.reset_buf[env], (long long)(misc[M_RESET] != 0.0f ? 1 : 0));                                                                                            \
        NTST(io.time_out_buf[env], (unsigned char)(misc[M_TIMEOUT] != 0.0f ? 1 : 0));                                                                                       \
      }                                                                                                                                               
extern __shared__ float smem[];

#define PBHC_TPB (2 * PBHC_G * PBHC_EPB)                   // threads per workgroup of k_env_step: two roles x 32 lanes x 4 envs
#define PBHC_HREG (384 / PBHC_G)                           // history words per lane held in registers (hist_dim <= 384)
#define PBHC_MAPREG 8                                      // map words per role-B thread held in registers while staged (the rest: a loop)
#define PBHC_MAP_HDR 36                                    // compact map block: [16 scales][16 noises][nn_early][nn_late][n_early][n_late][u16 pair list][noisy][pairs]

// Observation elements of group block `mg`: out[j] = clip(feat[src[j]] * scale[seg[j]]) for the element pairs named by entries [k0, k1) of
// the block's pair list (16-bit pair indices, grouped on the host by readiness class).  `nl` lanes (32 or 64) of this env cooperate, `l` is
// this lane's index among them; every pair costs one list read, one map word, four feature / scale reads and ONE 8-byte store.  Lanes past
// the end recompute the last entry and store the same value to the same address (branch-free batches).
template <int BATCH>
__device__ __forceinline__ void obs_write_list(const uint32_t* mg, int k0, int k1, int l, int nl, const float* feat, float* __restrict__ outg, unsigned int ob,
                                               int dim, int pitch_g, int clip, float clipobs) {
  const float* segs = (const float*)mg;
  const int nn = (int)(mg[32] + mg[33]);
  const int nlist = (int)(mg[34] + mg[35]);
  const uint16_t* list = (const uint16_t*)(mg + PBHC_MAP_HDR);
  const uint32_t* m32 = mg + PBHC_MAP_HDR + ((nlist + 1) >> 1) + nn;
  const bool pad_ok = pitch_g >= dim + 1;                  // a trailing odd element stores its pair's second half into the row padding
  for (int e0 = k0 + l; e0 < k1; e0 += BATCH * nl) {
    int p[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) p[u] = (int)list[min(e0 + u * nl, k1 - 1)];
    uint32_t w[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) w[u] = m32[p[u]];
    float xa[BATCH], xb[BATCH], sa[BATCH], sb[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const uint32_t lo = w[u] & 0xFFFFu, hi = w[u] >> 16;
      xa[u] = feat[lo & 0xFFFu]; sa[u] = segs[lo >> 12];
      xb[u] = feat[hi & 0xFFFu]; sb[u] = segs[hi >> 12];
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int j = 2 * p[u];
      float va = xa[u] * sa[u], vb = xb[u] * sb[u];
      if (clip) { va = __builtin_amdgcn_fmed3f(va, -clipobs, clipobs); vb = __builtin_amdgcn_fmed3f(vb, -clipobs, clipobs); }
      if (j + 1 < dim || pad_ok) *reinterpret_cast<float2*>(&at(outg, ob + (unsigned int)j)) = make_float2(va, vb);
      else at(outg, ob + (unsigned int)j) = va;
    }
  }
}

// The same element by element, for caller-owned rows that are not 8-byte aligned.
__device__ __forceinline__ void obs_write_list_unaligned(const uint32_t* mg, int k0, int k1, int l, int nl, const float* feat, float* __restrict__ outg, unsigned int ob,
                                                         int dim, int clip, float clipobs) {
  const float* segs = (const float*)mg;
  const int nn = (int)(mg[32] + mg[33]);
  const int nlist = (int)(mg[34] + mg[35]);
  const uint16_t* list = (const uint16_t*)(mg + PBHC_MAP_HDR);
  const uint16_t* m16 = (const uint16_t*)(mg + PBHC_MAP_HDR + ((nlist + 1) >> 1) + nn);
  for (int e = 2 * k0 + l; e < 2 * k1; e += nl) {
    const int j = 2 * (int)list[e >> 1] + (e & 1);
    if (j < dim) {
      const uint32_t w = m16[j];
      float v = feat[w & 0xFFFu] * segs[w >> 12];
      if (clip) v = clampf(v, -clipobs, clipobs);
      at(outg, ob + (unsigned int)j) = v;
    }
  }
}

// Noisy elements [k0, k1) of the block's noise list — and the other element of a pair that holds one: such pairs belong to no pair list —
// out[j] = clip((feat[src] + (2U - 1) * noise * curriculum) * scale), four entries per lane per Philox4x32 call (helpers.py:128-152).
// Uniforms: ONE Philox4x32 quad per lane and step (`base`, keyed by env / step / lane, computed in the prologue while the wave waits for
// its loads: seven dependent rounds per group and pass cost ~1.4 k cycles each where they stood) is spread over the groups and list
// quads by a bijective 32-bit finaliser (two multiply / xor-shift rounds) of base[u] ^ f(group, quad): distinct (env, step, lane, u, group,
// quad) tuples give decorrelated words, which is all observation noise asks for (helpers.py:152 draws torch.rand_like).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
constexpr bool cfg_has_term(const PbhcEnvConfig& c, int id) {
  for (int i = 0; i < c.num_terms; ++i)
    if (c.term_id[i] == id) return true;
  return false;
}
__device__ __forceinline__ bool is_special_term(int id) {     // the reward terms k_env_step evaluates by formula (the others: one slot of the reduction row)
  return id == PBHC_R_TELEOP_CONTACT_MASK || id == PBHC_R_TELEOP_CONTACT_MASK_V2 || id == PBHC_R_TELEOP_BODY_POSITION_EXTEND || id == PBHC_R_PENALTY_ORIENTATION ||
         id == PBHC_R_FEET_AIR_TIME || id == PBHC_R_PENALTY_FEET_CONTACT_FORCES || id == PBHC_R_PENALTY_STUMBLE || id == PBHC_R_PENALTY_SLIPPAGE ||
         id == PBHC_R_FOOT_SLIP_PENALTY || id == PBHC_R_ALIVE || id == PBHC_R_FEET_HEADING_ALIGNMENT || id == PBHC_R_FEET_HEADING_ALIGNMENT_CONTACT ||
         id == PBHC_R_PENALTY_FEET_ORI || id == PBHC_R_PENALTY_FEET_ORI_CONTACT;
}
constexpr bool obs_runs_complete(const PbhcEnvConfig& c) {
  for (int g = 0; g < c.num_groups; ++g)
    if (c.groups[g].num_runs < 0) return false;
  return true;
}
// LDS plan of one k_env_step workgroup (PBHC_EPB envs) for a config.  `use_runs`: the build writes the observation rows as unrolled
// runs (the config-specialised kernel whose groups all have a run table) — such a build stages no compact maps, and it may keep the
// HISTORY block out of the feature row: the old history then waits in the reference waves' registers and is staged, after bar2, over the
// rigid-body arrays of the simulator state, which are dead by then (`hist_in_bodies`; needs the history to be the last block of the
// feature index space, every row written by the reference waves, and a history no larger than those arrays / the registers).  That is
// 5 KB of LDS per workgroup for the walk config: 29.6 KB instead of 39.5 — FIVE workgroups per CU instead of four.  The launch is bound by
// what a CU holds in flight (an env's chain is ~11 us whatever the env count), so occupancy is throughput.
#define PBHC_SEG 128                                      // floats of an observation row composed in LDS and stored 16 bytes per lane at a time
struct StepLds { int stride, hist_in_bodies, map_words, bytes, stage; };
__host__ __device__ constexpr StepLds step_lds_plan(const PbhcEnvConfig& c, bool use_runs) {
  const int Bx = c.skel.num_bodies_ext, p = (Bx + 3) & ~3;
  const Lds lo(Bx, c.tracking_mode);
  const int hoff = c.feat_off[PBHC_F_HISTORY];
  bool all_b = true;
  for (int g = 0; g < c.num_groups; ++g)
    if (c.groups[g].role != 1) all_b = false;
#ifdef PBHC_NO_HISTB          // (measurement aid: the round-3 LDS plan)
  const bool hb = false;
#else
  const bool hb = use_runs && all_b && hoff + c.hist_dim == c.feat_dim && ((c.hist_dim + 3) & ~3) <= 13 * p && c.hist_dim <= (384 / PBHC_G) * PBHC_G;
#endif
  const int feat_words = hb ? hoff : c.feat_dim;
  // (hist_in_bodies builds also compose their rows in a PBHC_SEG-float staging segment per env: obs_write_wide)
  const int stage = lo.feat + ((feat_words + 3) & ~3);
#ifdef PBHC_WIDE_ROWS
  const int stride = stage + (hb ? PBHC_SEG : 0);
#else
  const int stride = stage;
#endif
  const int mapw = use_runs ? 0 : c.map_lds_words;
#ifdef PBHC_WIDE_ROWS
  const bool skc_lds = true;                                  // (the second staging segment lives there)
#else
  const bool skc_lds = !cfg_fk_jump(c);                       // the pointer-jumping chain keeps its constants in registers (fk_jump_wave)
#endif
  const int words = PBHC_EPB * stride + (skc_lds ? ((Bx * (11 + PBHC_MAX_DEPTH) + 3) & ~3) : 0) + mapw;
  return StepLds{stride, hb ? 1 : 0, mapw, words * 4, stage};
}
// the uniform of element j of row `stream`: the first word of the env's Philox quad of this step (keyed by env / step only: every lane
// computes the same one, so the value does not depend on WHICH lane writes element j), re-keyed by (row, j) and passed through a bijective
// 32-bit finaliser — every writer of observation noise (list, per-element map, unrolled runs) uses this one function, so the generic and
// the specialised kernel draw the same noise, and a test can restate it on the host (tests/helpers.py: expected_obs_noise).  (One word on purpose: choosing among the four by j & 3 turns
// the quad into an indexed array, i.e. scratch memory.)
__device__ __forceinline__ float obs_noise_u(const uint32_t* pre, uint32_t stream, uint32_t j) {
  return u01(mix32(pre[0] ^ (stream * 0x9E3779B9u + j * 0x85EBCA6Bu)));
}
__device__ __forceinline__ void obs_write_noisy(const uint32_t* mg, int k0, int k1, int l, int nl, const float* feat, float* __restrict__ outg, unsigned int ob,
                                                int clip, float clipobs, float noise_cur, uint32_t stream, const uint32_t* pre) {
  const float* segs = (const float*)mg;
  const int nlist = (int)(mg[34] + mg[35]);
  const uint32_t* noisy = mg + PBHC_MAP_HDR + ((nlist + 1) >> 1);
  for (int kb = (k0 & ~3) + 4 * l; kb < k1; kb += 4 * nl) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (kb + u >= k0 && kb + u < k1) {
        const uint32_t e = noisy[kb + u];
        const uint32_t w = e >> 16;
        const int seg = w >> 12;
        float v = (feat[w & 0xFFFu] + (obs_noise_u(pre, stream, e & 0xFFFFu) * 2.0f - 1.0f) * (segs[16 + seg] * noise_cur)) * segs[seg];
        if (clip) v = __builtin_amdgcn_fmed3f(v, -clipobs, clipobs);
        at(outg, ob + (e & 0xFFFFu)) = v;
      }
  }
}

// Observation row as RUNS (PbhcObsRun), for the config-specialised build: `m` is a compile-time constant there, both loops unroll and what
// is left per element is one LDS read at an immediate offset, the scale as a literal, the clip and one store — no list, no map word, no
// segment table (the per-element paths above: ~25 instructions per element pair).  WHICH: 0 every run, 1 the runs that read no post-reset
// feature, 2 the ones that do.  Same arithmetic per element as the map paths: (x + noise) * scale, clip.
// `fhist`: where feature indices >= `hoff` (the HISTORY block) live, as a base for the same index (feat itself unless the block is staged
// elsewhere: step_lds_plan); the table builder never lets a run straddle `hoff`
// WHO (round 4, builds whose rows all belong to the reference waves): 0 every run; 1 all but the runs the host marked for the DYNAMICS waves
// (PbhcObsRun.late bit 1: runs that read no history — those waves idle after their reward / reset phases while the reference waves write
// 1 010 elements per env, the longer path after bar2), 2 only those.
template <int WHICH, int WHO = 0>
__device__ __forceinline__ void obs_write_runs(const PbhcOutMap& m, uint32_t stream, int lane, const float* feat, const float* fhist, int hoff, float* __restrict__ outg,
                                               unsigned int ob, float clipobs, float noise_cur, const uint32_t* pre) {
#pragma unroll
  for (int r = 0; r < m.num_runs; ++r) {
    const PbhcObsRun& R = m.runs[r];
    if ((WHICH == 1 && (R.late & 1)) || (WHICH == 2 && !(R.late & 1))) continue;
    if ((WHO == 1 && (R.late & 2)) || (WHO == 2 && !(R.late & 2))) continue;
#pragma unroll
    for (int i0 = 0; i0 < R.len; i0 += PBHC_G) {
      // lanes past the run's end repeat its last element (the same value to the same address): no exec-mask region per partial
      // iteration, so the whole row is ONE basic block and the LDS reads of many iterations are in flight together (with a branch per
      // run tail every iteration waited out its own LDS round trip: ~100 cycles each)
      const int i = min(i0 + lane, R.len - 1);
      float x = (R.src >= hoff ? fhist : feat)[R.src + i];
      if (R.noise != 0.0f) x = x + (obs_noise_u(pre, stream, (uint32_t)(R.dst + i)) * 2.0f - 1.0f) * (R.noise * noise_cur);
      x = x * R.scale;
      if (m.clip) x = __builtin_amdgcn_fmed3f(x, -clipobs, clipobs);
      NTST(at(outg, ob + (unsigned int)(R.dst + i)), x);
    }
  }
}

#ifdef PBHC_STATIC_CFG
constexpr int spec_min_waves() {
  const int n = (160 * 1024) / step_lds_plan(kStaticCfg, obs_runs_complete(kStaticCfg)).bytes;
  return n < 1 ? 1 : (n > 5 ? 5 : n);
}
#endif
// The same rows 16 BYTES PER LANE (round 4).  At 32 768 envs the launch is bound by the vector-memory pipeline, not by bytes or arithmetic
// (profiles/round4_k_env_step_memory_pipeline.txt: the address unit of a CU busy 65 % of the launch, 92 wave-instructions per wave of which
// nearly all carried 4 bytes per lane), and a third of those instructions were the dword stores of these rows.  A row is composed PBHC_SEG
// floats at a time in a staging segment of the env's LDS — every run piece that falls into the segment written by the lanes as above
// (lane <-> element: consecutive LDS words), same arithmetic, same noise words — and leaves as ONE ds_read_b128 + global_store_dwordx4 per
// segment: a quarter of the store instructions, whole 128-byte lines.  DS instructions of a wave execute in order, so one segment buffer
// serves all segments of all rows.  PASS 0: every segment for an env that keeps its state, the segments without post-reset features for
// a terminated one; PASS 1 (after bar3): the remaining segments of a terminated env.  Needs io.obs_wide (rows 16-byte aligned, padded).
__device__ __forceinline__ void wide_compose(const PbhcOutMap& m, int s0, uint32_t stream, int lane, const float* feat, const float* fhist, int hoff, float* stg,
                                             float clipobs, float noise_cur, const uint32_t* pre) {
  const int s1 = min(s0 + PBHC_SEG, m.dim);
#pragma unroll
  for (int r = 0; r < m.num_runs; ++r) {
    const PbhcObsRun& R = m.runs[r];
    const int lo = max(R.dst, s0), hi = min(R.dst + R.len, s1);            // this run's piece of the segment, in row elements
    if (lo >= hi) continue;
#pragma unroll
    for (int i0 = lo; i0 < hi; i0 += PBHC_G) {
      const int j = min(i0 + lane, hi - 1);                                   // (lanes past the piece repeat its last element: one basic block)
      float x = (R.src >= hoff ? fhist : feat)[R.src + (j - R.dst)];
      if (R.noise != 0.0f) x = x + (obs_noise_u(pre, stream, (uint32_t)j) * 2.0f - 1.0f) * (R.noise * noise_cur);
      x = x * R.scale;
      if (m.clip) x = __builtin_amdgcn_fmed3f(x, -clipobs, clipobs);
      stg[j - s0] = x;
    }
  }
  if (s1 == m.dim && (m.dim & 3) != 0 && lane < 4 - (m.dim & 3)) stg[m.dim - s0 + lane] = 0.0f;      // the row's padding up to a whole quad
}
__device__ __forceinline__ void wide_flush(const PbhcOutMap& m, int s0, int lane, const float* stg, float* __restrict__ outg, unsigned int ob) {
  const int nq = (min(s0 + PBHC_SEG, m.dim) - s0 + 3) >> 2;
  const int q = min(lane, nq - 1);
  const float4 v = *reinterpret_cast<const float4*>(stg + 4 * q);
  *reinterpret_cast<float4*>(&at(outg, ob + (unsigned int)(s0 + 4 * q))) = v;
}
constexpr bool seg_has_late(const PbhcOutMap& m, int s0) {
  for (int r = 0; r < m.num_runs; ++r)
    if ((m.runs[r].late & 1) && m.runs[r].dst < s0 + PBHC_SEG && m.runs[r].dst + m.runs[r].len > s0) return true;
  return false;
}
template <int PASS>
__device__ __forceinline__ void obs_write_wide(const PbhcOutMap& m, uint32_t stream, int lane, const float* feat, const float* fhist, int hoff, float* stagel,
                                               float* __restrict__ outg, unsigned int ob, float clipobs, float noise_cur, const uint32_t* pre, bool rs) {
#pragma unroll
  for (int s0 = 0; s0 < m.dim; s0 += PBHC_SEG) {
    const bool has_late = seg_has_late(m, s0);
    if (PASS == 1 && !has_late) continue;
    const bool doit = PASS == 0 ? (!rs || !has_late) : rs;
    if (doit) {
      wide_compose(m, s0, stream, lane, feat, fhist, hoff, stagel, clipobs, noise_cur, pre);
      WAVE_LDS_FENCE();
      wide_flush(m, s0, lane, stagel, outg, ob);
      WAVE_LDS_FENCE();
    }
  }
}
// The common case — neither env of the wave resets — as ONE basic block: every segment of the row, composed alternately in two staging
// buffers, segment k leaving (ds_read_b128 + store) after segment k+1 has been composed: the flush's LDS round trip and the next
// segment's feature reads are in flight together (with one buffer every segment waited out compose -> read-back -> store: ~300 cycles
// each, 11 segments per env).  `stg2`: the second buffer (the skeleton image's LDS, dead since the FK).
__device__ __forceinline__ void obs_write_wide_all(const PbhcOutMap& m, uint32_t stream, int lane, const float* feat, const float* fhist, int hoff, float* stg0, float* stg1,
                                                   float* __restrict__ outg, unsigned int ob, float clipobs, float noise_cur, const uint32_t* pre, int* parity) {
  int par = *parity;
#pragma unroll
  for (int s0 = 0; s0 < m.dim; s0 += PBHC_SEG) {
    wide_compose(m, s0, stream, lane, feat, fhist, hoff, par ? stg1 : stg0, clipobs, noise_cur, pre);
    WAVE_LDS_FENCE();
    wide_flush(m, s0, lane, par ? stg1 : stg0, outg, ob);
    par ^= 1;
  }
  *parity = par;
}

// MODE 0: LeggedRobotMotionTracking (motion_tracking.py), MODE 1: LeggedRobotGeneralTracking (general_tracking.py)
template <int MODE>
// waves per SIMD the register allocation must allow: the v1 kernel's LDS footprint admits 4 workgroups = 16 waves per CU (<= 128 VGPRs);
// general tracking holds twice the LDS per env (2 workgroups per CU)
#ifndef PBHC_MIN_WAVES
#ifdef PBHC_STATIC_CFG
// a workgroup puts one wave on every SIMD: waves per SIMD = workgroups per CU = what the config's LDS plan admits (160 KB), at most 5 here —
// 6 would leave 80 VGPRs
#define PBHC_MIN_WAVES spec_min_waves()
#else
#define PBHC_MIN_WAVES (MODE ? 2 : 4)
#endif
#endif
// (waves_per_eu pins the allocation target too: LDS admits no more than PBHC_MIN_WAVES waves per SIMD, so aiming at a higher occupancy
// — the compiler stopped at 96 VGPRs and spilled — buys nothing)
// The first eight parameters are the addresses (and two scalars) the first memory round trip of a workgroup hangs on — a copy of what `io`
// and the config also hold — so that each role can request the episode clock / the replay frame first thing.  Measured and NOT adopted:
// preloading them into SGPRs at wave launch (PBHC_KERNARG_PRELOAD=14 -> -mllvm -amdgpu-kernarg-preload-count): the reference waves then
// enter their role 395 cycles into the launch instead of 1 525 and issue these loads at once — and the data is back at the same 6 k cycles:
// with every workgroup of the chip in its prologue the first round trip is the HBM burst itself (12.7 MB at ~5 TB/s), not the issue time
// (profiles/round4_k_env_step_prologue.txt).
__global__ __launch_bounds__(PBHC_TPB, PBHC_MIN_WAVES) __attribute__((amdgpu_waves_per_eu(PBHC_MIN_WAVES, PBHC_MIN_WAVES))) void k_env_step(
                                                              const long long* __restrict__ a_ep_len, const float* __restrict__ a_start, const float* __restrict__ a_frame_root,
                                                              const float* __restrict__ a_frame_q, const float* __restrict__ a_frame_qd, const int32_t* __restrict__ a_cursor,
                                                              int a_frame_index, int a_num_envs,
                                                              const PbhcEnvConfig* __restrict__ cfgp, PbhcMotionTable tbl, PbhcStepIO io,
                                                              const double* __restrict__ glob, float* __restrict__ partials,
                                                              int lds_stride, const float* __restrict__ skc_img, const uint32_t* __restrict__ map_img,
                                                              const float* __restrict__ skj_img) {
  // `rt`: the run-time config (device memory).  `c`: the same values, or — in a config-specialised build — a constexpr copy
  // whose scalars fold into the instruction stream; pointers, seed, env count and reference yaw always come from `rt`.
  // The config is read through the CONSTANT address space: the kernel never writes it, and saying so lets the compiler keep its
  // scalars in SGPRs across the kernel's global stores.
  WG_STAMP(0);
#ifdef PBHC_STAGGER
  // experiment: every other workgroup starts PBHC_STAGGER x ~1 us late, so that one half of a CU's workgroups is in its load burst while
  // the other half computes (see DESIGN §4 for what it measured)
  if (blockIdx.x & 1)
    for (int i = 0; i < PBHC_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif
  typedef const PbhcEnvConfig __attribute__((address_space(4))) ConstCfg;
  ConstCfg& rt = *(ConstCfg*)cfgp;
#ifdef PBHC_STATIC_CFG
  const PbhcEnvConfig& c = kStaticCfg;
#else
  ConstCfg& c = rt;
#endif
  const auto& sk = c.skel;
  const int N = a_num_envs, D = sk.num_dof, B = sk.num_bodies, Bx = sk.num_bodies_ext, NF = c.num_feet;
  const int lane = threadIdx.x & (PBHC_G - 1);
  const int wave = threadIdx.x >> 6;
  const bool roleB = wave >= 2;                                            // wave-uniform (alternating the roles' waves between co-resident
                                                                           // workgroups, so that every SIMD holds two waves of each role, measured no change)
  const int le = ((wave & 1) << 1) | ((threadIdx.x >> 5) & 1);             // env slot of this half-wave in the workgroup
  typedef unsigned int u32;
  // Which 4 envs this workgroup steps.  The dispatcher deals workgroups to the 8 XCDs round-robin (workgroup b -> XCD b % 8), and every XCD
  // has its own L2: with env block = b, the [N, 23]-float state arrays — 92-byte rows, 368 bytes per workgroup — put every 128-byte line
  // that two neighbouring workgroups share into TWO L2s, each of which fetches it and writes its half back as a partial line.  XCD x
  // takes a CONTIGUOUS range of env blocks instead (bijective for any block count), so neighbours meet in one L2: whole-line write-backs,
  // one fetch per line.  (The partial-sum rows stay indexed by env block: k_env_finalize adds them up in the same order as ever.)
#ifdef PBHC_NO_XCD_MAP
  const u32 wgb = blockIdx.x;
#else
  const u32 nb_ = gridDim.x, xq_ = nb_ >> 3, xr_ = nb_ & 7u, xcd_ = blockIdx.x & 7u;
  const u32 wgb = xcd_ * xq_ + min(xcd_, xr_) + (blockIdx.x >> 3);
#endif
  const int env = (int)wgb * PBHC_EPB + le;
  const bool valid = env < N;
  float* S = smem + (size_t)le * lds_stride;
  float *act = S + Lds::ACT, *actd = S + Lds::ACTD, *tau = S + Lds::TAU, *q = S + Lds::Q, *qd = S + Lds::QD;
  float *rdof = S + Lds::RDOF, *rdofv = S + Lds::RDOFV, *root = S + Lds::ROOT, *misc = S + Lds::MISC, *cf = S + Lds::CF;
  const Lds lo(Bx, MODE);
  float *bp = S + Lds::BP, *bq = S + lo.bq, *bv = S + lo.bv, *bw = S + lo.bw;
  float *rp = S + lo.rp, *rq = S + lo.rq, *rv = S + lo.rv, *rw = S + lo.rw;
  float *red = S + lo.red, *feat = S + lo.feat;
  // replay frame of this step: named by the host, or read from the device-side cursor
  const size_t fk = (size_t)(a_frame_index >= 0 ? a_frame_index : a_cursor[0] % io.num_frames) * (size_t)N;
  // Addressing: every per-env tensor is indexed as <uniform 64-bit base> + <32-bit unsigned lane offset> (`at`), which the compiler emits
  // as the SGPR-base form of global_load / global_store (pbhc_env_create / pbhc_env_step check that num_envs x row pitch < 2^30 elements).
  const uint32_t step_ctr = (uint32_t)glob[PBHC_G_STEP_COUNTER];                 // RNG counter: advanced by k_env_finalize
  float* skc = smem + (size_t)PBHC_EPB * lds_stride;          // [SKC_WORDS] skeleton constants, shared by the workgroup
  uint32_t* mapl = (uint32_t*)(skc + ((Bx * SKC_W + 3) & ~3));    // [map_lds_words] compact observation maps, shared by the workgroup
  const float dt = c.dt;
  const u32 eD = (u32)env * (u32)D;
  const int envc = valid ? env : N - 1;                       // a tail workgroup's missing envs load env N-1 (and store nothing)
  const u32 eDc = (u32)envc * (u32)D;
  const int d = lane;                                         // D <= 32: one dof per lane
  const int dc = min(lane, D - 1);
  const int hoff = c.feat_off[PBHC_F_HISTORY];
  const int Q = c.queue_len;
  // how the observation rows are written: unrolled runs (specialised build whose groups all have a run table), else the compact maps
  // staged in LDS, else (a feature row too large for them) the per-element maps in global memory after bar3
#ifdef PBHC_STATIC_CFG
  constexpr bool use_runs = obs_runs_complete(kStaticCfg);
#else
  const bool use_runs = false;
#endif
  const int map_words = use_runs ? 0 : c.map_lds_words;
  const bool obs_by_role = use_runs || map_words > 0;
  // where the HISTORY block of the feature index space lives (step_lds_plan): in the feature row, or — staged by the reference waves after
  // bar2 — over the simulator body arrays
#ifdef PBHC_STATIC_CFG
  constexpr bool hist_b = step_lds_plan(kStaticCfg, use_runs).hist_in_bodies != 0;
  constexpr bool fk_jump = cfg_fk_jump(kStaticCfg);                             // the rigid-body chain by pointer jumping (fk_jump_wave)
  constexpr int fk_rounds = skel_fk_rounds(kStaticCfg.skel.max_depth);
#else
  const bool hist_b = false;
  const bool fk_jump = skel_fk_jump(Bx, sk.max_depth);                          // (the generic kernel: the same algorithm, decided per launch)
  const int fk_rounds = skel_fk_rounds(sk.max_depth);
#endif
  // hist_b builds: the per-dof episodic DR draws of a reset (gain / torque-noise scales, queue) are the reference waves' — they own the
  // kp / kd features and idle at bar3 in exactly the workgroups that finish last (the ones with a terminated env), while the dynamics
  // waves still have the reward phase and the rest of the reset in front of them
  const bool dr_on_b = hist_b;
  // ... and the runs the host marked (PbhcObsRun.late bit 1) are written by the dynamics waves after their phase H (OBS_GROUPS_HELP_RUNS)
#if defined(PBHC_STATIC_CFG) && !defined(PBHC_WIDE_ROWS) && !defined(PBHC_NO_ROW_HELP)
  constexpr bool row_help = hist_b;
#else
  constexpr bool row_help = false;
#endif
  constexpr int RUN_WHO = row_help ? 1 : 0;
  float* const histl = hist_b ? bp : feat + hoff;
  const float* const fhist = histl - hoff;
#ifdef PBHC_STATIC_CFG
  float* const stagel = S + step_lds_plan(kStaticCfg, use_runs).stage;       // (hist_b builds only)
  float* const stage2 = skc + le * PBHC_SEG;                                  // second staging segment: the skeleton image, dead since the FK (bar1)
  static_assert(!hist_b || PBHC_EPB * PBHC_SEG <= SKC_WORDS, "second staging buffer");
#ifndef PBHC_WIDE_ROWS        // the 16-byte form is opt-in (-DPBHC_WIDE_ROWS): measured, not adopted — see obs_write_wide
  const bool wide = false;
#else
  const bool wide = hist_b && io.obs_wide != 0;
#endif
#define OBS_GROUPS_WIDE_ALL()                                                                                                       \
  { int par_ = 0;                                                                                                                   \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups) continue;                                                                                                \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    obs_write_wide_all(c.groups[g], 16 + g, lane, feat, fhist, hoff, stagel, stage2, io.obs[g], (u32)env * (u32)pitch_g, clipobs, noise_cur, nzb, &par_); \
  } }
#define OBS_GROUPS_WIDE(PASS, RS)                                                                                                   \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups) continue;                                                                                                \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    obs_write_wide<PASS>(c.groups[g], 16 + g, lane, feat, fhist, hoff, stagel, io.obs[g], (u32)env * (u32)pitch_g, clipobs, noise_cur, nzb, RS); \
  }
#else
  const bool wide = false;
#define OBS_GROUPS_WIDE(PASS, RS)
#define OBS_GROUPS_WIDE_ALL()
#endif
  const int o_pos = 2 * D + 2, o_rot = o_pos + 3 * Bx, o_vel = o_rot + 4 * Bx, o_ang = o_vel + 3 * Bx;     // columns of a packed motion-table row
  STAMP(0);

  // ---------------- per-env scalars (tiny loads; each role loads what it uses, the reset path re-reads the clip meta) -----------------
#define LOAD_CLIP_ID()                                                                                                 \
  const int mid = (int)io.motion_ids[envc];                                                                            \
  const f3 origin = mk3(at(io.env_origins, (u32)envc * 3u), at(io.env_origins, (u32)envc * 3u + 1u), at(io.env_origins, (u32)envc * 3u + 2u));
#define LOAD_CLIP_META_OF_ID()                                                                                         \
  float m_len = tbl.single_len, m_dt = tbl.single_dt;          /* this env's clip: length, frame time, frames, first table row */ \
  int m_nf = tbl.single_num_frames, m_row0 = 0;                                                                        \
  if (tbl.num_motions != 1) {                                  /* single clip: its meta travels in the kernel arguments */ \
    m_len = tbl.motion_len[mid]; m_nf = tbl.num_frames[mid]; m_dt = tbl.motion_dt[mid]; m_row0 = tbl.length_starts[mid]; \
  }
#define LOAD_CLIP_META() LOAD_CLIP_ID() LOAD_CLIP_META_OF_ID()

  long long ep1 = 0;                                          // role B: the episode clock of this step (loaded first thing in its prologue)
  float start = 0.0f;
  // role-A registers that live across phases
  uint32_t psrc0 = 0, psrc1 = 0;                              // sources of this lane's two partial-sum columns (kPartTab)
  float sumrow = 0.0f, pf_tscale = 0.0f, pf_sigma = 1.0f, pf_pen_scale = 1.0f, pf_far_thr = 0.0f, kpA = 1.0f, kdA = 1.0f, dpA = 0.0f, etr_old = 0.0f;
  int pf_tid = 0, pf_tpen = 0, pf_tsrc = -1, pf_colterm = -1;
  long long adelay = 0;
  // role-B registers that live across phases (loads issued in its prologue, consumed after bar1)
  float pf_last_act = 0.0f, pf_last_qd = 0.0f;
  float tref = 0.0f;
  float qold[PBHC_MAX_QUEUE];
  float hreg[PBHC_HREG];                                      // role B: the env's history row, requested before bar1, copied to LDS after it
  // ... hist_b builds: the same row 16 bytes per lane (3 load instructions instead of 10: at 4096 envs every workgroup of the chip is in its
  // prologue at once and the CU's address unit, one wave-instruction at a time whatever its width, is what the load burst queues behind)
#define PBHC_HREG4 ((384 / 4 + PBHC_G - 1) / PBHC_G)
  float4 hreg4[PBHC_HREG4];
  // Philox quads computed ahead of their use, while the wave waits for its loads: the first quad of every group's noise list (both roles;
  // role A needs role B's rows too after a reset), and role A's reset draws
  uint32_t nzb[4];
  float a_in = 0, qp = 0, qv = 0, kp = 1, kd = 1, rfs = 1, ras = 0, u_inj = 0, bmass = 1, lmreg = 0, combias = 0, fric = 0;
  float u_rfi = 0.5f, k_tl = 0.0f, k_dp = 0.0f, clipcnt = 0.0f;
  // terminate_when_close_to_{dof_pos,dof_vel,torque}_limit (legged_robot_base.py:449-479; off in the shipped yamls): role B owns the joint-space
  // quantities, so it raises these causes before bar2; role A folds them into the reset flag right after bar2, role B keeps its own copy
  const bool close_any = c.terminate_close_pos || c.terminate_close_vel || c.terminate_close_tau;
  bool gateB = false;
  long long adelayB = 0;
  int didx = 0;
  const u32 qoff = (u32)envc * (u32)(Q * D) + (u32)dc;
  // reductions of the two roles (role A: body sums, role B: joint-space sums); declared here, reduced after their loops
  float s_up = 0, s_lo = 0, s_vr = 0, s_feet = 0, s_rot = 0, s_vel = 0, s_ang = 0, s_maxn = 0, s_upn = 0, s_lon = 0, s_vrn = 0;
  float s_key = 0, s_keyn = 0, s_lkey = 0, s_lkeyn = 0, s_lkrot = 0, s_kvel = 0, s_kang = 0, s_lupn = 0, s_llon = 0, s_lvrn = 0, s_bodyz = 0;

  if (!roleB) {
    // =============== role A, interval 0: the replay frame -> LDS (what the FK chain waits for), then the loads of its later phases.
    // The skeleton constants are staged by the reference waves (bar0 below): round 3 had EACH dynamics wave load the whole 2.5 KB image
    // for itself — 10 of its 25 load instructions and 9 % of the bytes a workgroup pulls through the CU's vector-memory pipeline, which
    // is what bounds the launch at large env counts (profiles/round4_k_env_step_memory_pipeline.txt).
    // the replay frame first: its addresses are in SGPRs since the wave was launched
    const float fq = NTLD(at(a_frame_q + fk * D, eDc + dc)), fqd = NTLD(at(a_frame_qd + fk * D, eDc + dc));
    const float froot = NTLD(at(a_frame_root + fk * 13, (u32)envc * 13u + (u32)min(lane, 12)));
    __builtin_amdgcn_sched_barrier(0);
    float4 kr[5];                                              // fk_jump: this lane's body constants, straight into registers
    if (fk_jump) {
      const float4* __restrict__ row = reinterpret_cast<const float4*>(skj_img + SKJ_W * min(lane, Bx - 1));
#pragma unroll
      for (int u = 0; u < 5; ++u) kr[u] = row[u];
    }
    {
      const int tl_ = min(lane, PBHC_MAX_TERMS - 1);
      pf_tid = c.term_id[tl_]; pf_tscale = c.term_scale[tl_]; pf_tpen = c.term_penalty[tl_]; pf_tsrc = c.term_src[tl_];
      pf_colterm = c.sum_col_term[lane];                     // lane i <-> episode_sums column i: the term that accumulates into it
      sumrow = NTLD(at(io.episode_sums, (u32)envc * (u32)c.num_sum_cols + (u32)min(lane, c.num_sum_cols - 1)));
      pf_sigma = (float)glob[PBHC_G_SIGMA + min(lane, PBHC_NUM_SIGMA - 1)];
      pf_pen_scale = (float)glob[PBHC_G_PENALTY_SCALE]; pf_far_thr = (float)glob[PBHC_G_MOTION_FAR_THR];
      kpA = NTLD(at(io.kp_scale, eDc + dc)); kdA = NTLD(at(io.kd_scale, eDc + dc));                // phase H (a reset replaces them in registers)
      dpA = io.default_dof_pos ? at(io.default_dof_pos, eDc + dc) : c.default_dof_pos[dc];
      adelay = io.action_delay_idx[envc];
      etr_old = io.end_time_ratio_buf[envc];
      const PartTab& pt = kPartTab[(MODE ? 2 : 0) + (close_any ? 1 : 0)];
      psrc0 = pt.v[lane]; psrc1 = pt.v[lane + PBHC_G];
    }

    if (!fk_jump) LDS_BARRIER();                               // bar0: the skeleton image is in LDS (an L2 hit: long before this wave's frame is)
    STAMP(1);
    if (valid) {
      if (d < D) { q[d] = fq; qd[d] = fqd; }
      if (lane < 13) root[lane] = froot;
    }
    WAVE_LDS_FENCE();
    // =============== role A, interval 1: rigid-body state of the new frame (sim-stub FK), wave-local ==============================
#ifndef PBHC_ABL_FK
    if (fk_jump) fk_jump_wave(kr, fk_rounds, B, Bx, lane, valid, root, q, qd, bp, bq, bv, bw);
    else fk_walk_wave(skc, B, Bx, lane, valid, root, q, qd, bp, bq, bv, bw);
#else       // (timing ablations, -DPBHC_ABL_*: what a phase costs is read off the launch time without it; results are meaningless)
    if (valid && lane < Bx) { st3(bp + 3 * lane, ld3(root)); st4(bq + 4 * lane, ld4(root + 3)); st3(bv + 3 * lane, ld3(root + 7)); st3(bw + 3 * lane, ld3(root + 10)); }
#endif
    // ---- phase C: per-env scalars of the root state (legged_robot_base.py:346-380).  One lane per quantity and ONE code path per function:
    // lanes 0-2 evaluate the three atan2 (yaw, heading, roll), lane 3 the asin of the pitch, lanes 4-6 the three base-frame rotations (lane 0
    // doing all of it in turn was ~450 instructions; roll and pitch only exist where an observation reads them: general tracking).
#ifdef PBHC_ABL_PHASEC
    if (false) {
#else
    if (valid) {
#endif
      const f4 rq4 = ld4(root + 3);
      if (lane < 3) {
        float sinr, cosr, sinp, siny, cosy;
        euler_xyz_args(rq4, &sinr, &cosr, &sinp, &siny, &cosy);
        const f3 hx = quat_rotate(rq4, mk3(1.0f, 0.0f, 0.0f));                  // calc_heading rotations.py:257-268
        const float ang = atan2f(lane == 0 ? siny : (lane == 1 ? hx.y : sinr), lane == 0 ? cosy : (lane == 1 ? hx.x : cosr));
        if (lane == 0) feat[c.feat_off[PBHC_F_RELYAW]] = ang - rt.ref_init_yaw;
        else if (lane == 1) st4(misc + M_HINV, quat_from_angle_z(-ang));        // calc_heading_quat_inv rotations.py:296-306
        else if (MODE) feat[c.feat_off[PBHC_F_ROLL_PITCH]] = ang;
      } else if (lane == 3) {
        if (MODE) feat[c.feat_off[PBHC_F_ROLL_PITCH] + 1] = euler_xyz(rq4).y;
      } else if (lane <= 6) {
        // the same rotation of three different vectors
        const f3 vin = lane == 4 ? ld3(root + 7) : (lane == 5 ? ld3(root + 10) : mk3(0.0f, 0.0f, -1.0f));
        const f3 vo = quat_rotate_inverse(rq4, vin);
        const int off = lane == 4 ? c.feat_off[PBHC_F_BASE_LIN_VEL] : (lane == 5 ? c.feat_off[PBHC_F_BASE_ANG_VEL] : c.feat_off[PBHC_F_PROJECTED_GRAVITY]);
        st3(feat + off, vo);
        if (lane == 6) { misc[M_GX] = vo.x; misc[M_GY] = vo.y; misc[M_GZ] = vo.z; }
      }
    }
    // the observation-noise base of this env and step (obs_noise_u): ONE Philox call per env, here — these waves wait at bar1 —, handed
    // to whichever wave writes a row through LDS
#ifndef PBHC_ABL_RNG
    philox4x32((uint32_t)rt.seed, (uint32_t)(rt.seed >> 32), env, step_ctr, 16, 0u, nzb);
#else
    nzb[0] = env ^ step_ctr; nzb[1] = nzb[2] = nzb[3] = 0;
#endif
    if (valid && lane == 0) misc[M_NZB] = __uint_as_float(nzb[0]);
    WAVE_LDS_FENCE();
    STAMP(2);
  } else {
    STAMPB(9);
    // the episode clock first: its addresses are in SGPRs since the wave was launched, and the reference rows' addresses hang on it
    ep1 = a_ep_len[envc] + 1;
    start = a_start[envc];
    const float frootB = NTLD(at(a_frame_root + fk * 13, (u32)envc * 13u + (u32)min(lane, 12)));      // phase C below
    __builtin_amdgcn_sched_barrier(0);
    // the skeleton constants for the dynamics waves' FK: ONE copy per workgroup, the first thing these two waves request (an L2 hit)
#define SKC_REGSB ((SKC_WORDS + 2 * 64 - 1) / (2 * 64))
    float skregB[SKC_REGSB];
    if (!fk_jump) {
      const int n = Bx * SKC_W, wl = threadIdx.x & 127;
#pragma unroll
      for (int u = 0; u < SKC_REGSB; ++u) skregB[u] = skc_img[min(wl + u * 128, n - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef PBHC_PTR_BURST
    // (measurement aid, neutral: the ~30 tensor addresses of this prologue fetched from the argument segment as ONE burst of scalar loads
    // instead of one or two at a time in front of their first use)
    asm volatile("" ::"s"(io.episode_length_buf), "s"(io.motion_start_times), "s"(io.motion_ids), "s"(io.env_origins), "s"(io.frame_root), "s"(io.feet_air_time),
                 "s"(io.last_contacts), "s"(io.frame_contact), "s"(io.action_queue), "s"(io.actions_in), "s"(io.dof_state), "s"(io.kp_scale), "s"(io.kd_scale),
                 "s"(io.rfi_lim_scale));
    asm volatile("" ::"s"(io.rao_scale), "s"(io.u_rfi), "s"(io.last_actions), "s"(io.last_dof_vel), "s"(io.action_delay_idx), "s"(io.motion_len), "s"(io.dr_base_com),
                 "s"(io.dr_friction), "s"(io.dr_link_mass), "s"(tbl.frames));
#endif
    STAMPB(10);
    LOAD_CLIP_ID();
    // =============== role B, interval 0: every other load of the step.  ORDER (loads return in issue order): (1) the env scalars the
    // reference rows' addresses hang on — episode length, start time, clip id: issued above; (2) every load that depends on nothing but
    // the env index, back to back (indices clamped, not predicated: one basic block); (3) — once (1) is back, with (2) still in flight — the
    // two table rows.  Round 3 issued (2) BEHIND (3), i.e. after the first round trip had come back: two full trips to memory (6.5 k cycles
    // to this role's first stamp) where one and an L2 hit do.  Then _pre_physics_step (motion_tracking.py:749-768) and the torques from the
    // pre-step state (legged_robot_base.py:795-838).
    const float fat = NTLD(at(io.feet_air_time, (u32)envc * (u32)NF + (u32)min(lane, NF - 1))), lastc = NTLD(at(io.last_contacts, (u32)envc * (u32)NF + (u32)min(lane, NF - 1)));
    // the frame's contact forces (3 B floats per env): whole quads 16 bytes per lane + the remainder, two load instructions instead of four
    float4 creg4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float cregr = 0.0f;
    {
      const float* __restrict__ csrc = io.frame_contact + fk * (size_t)(B * 3);
      const u32 cbase = (u32)envc * (u32)(B * 3);
      const int nq = (B * 3) >> 2;
      const float* qa = &at(csrc, cbase + 4u * (u32)min(lane, nq - 1));
      creg4 = make_float4(NTLD(qa[0]), NTLD(qa[1]), NTLD(qa[2]), NTLD(qa[3]));               // (4-byte aligned rows: the compiler's own dwordx4, as for the table rows)
      if ((B * 3) & 3) cregr = NTLD(at(csrc, cbase + (u32)min(4 * nq + lane, B * 3 - 1)));
    }
    // operands of the pre-physics step / torques / joint-space sums: consumed after bar1 — requested behind phase D, where the reference rows'
    // 32 registers are free again: 13 registers off this prologue's peak (92 -> 76 VGPRs; 18.0 / 95.0 -> 18.0 / 92.0 us at 4096 / 32 768 envs,
    // profiles/round4_k_env_step_variants.txt (i); -DPBHC_EARLY_OPERANDS: with the prologue's other loads, as before).  A sixth workgroup per CU,
    // which 76 registers admit (-DPBHC_MIN_WAVES=6: resident by hipOccupancyMaxActiveBlocksPerMultiprocessor), measured no further gain.
#define LOAD_STEP_OPERANDS()                                                                                                          \
    _Pragma("unroll") for (int k = 0; k < PBHC_MAX_QUEUE; ++k) qold[k] = NTLD(at(io.action_queue, qoff + (u32)(min(k, Q - 1) * D)));       \
    a_in = NTLD(at(io.actions_in, eDc + dc));                                                                                               \
    qp = NTLD(at(io.dof_state, (eDc + dc) * 2)); qv = NTLD(at(io.dof_state, (eDc + dc) * 2 + 1));                                               \
    kp = NTLD(at(io.kp_scale, eDc + dc)); kd = NTLD(at(io.kd_scale, eDc + dc)); rfs = NTLD(at(io.rfi_lim_scale, eDc + dc)); ras = NTLD(at(io.rao_scale, eDc + dc)); \
    u_inj = NTLD(at(io.u_rfi ? io.u_rfi : io.actions_in, eDc + dc));                                                                        \
    pf_last_act = NTLD(at(io.last_actions, eDc + dc)); pf_last_qd = NTLD(at(io.last_dof_vel, eDc + dc));
#ifdef PBHC_EARLY_OPERANDS
    LOAD_STEP_OPERANDS()
#endif
    adelayB = io.action_delay_idx[envc];
    const float mlenB = io.motion_len[envc];
    didx = c.randomize_ctrl_delay ? (int)adelayB : 0;
    bmass = (MODE && io.dr_base_mass) ? io.dr_base_mass[envc] : 1.0f;
    {
      const int nlm = max(c.dr_link_mass_dim, 1);
      lmreg = at(c.dr_link_mass_dim > 0 ? io.dr_link_mass : io.dr_base_com, (u32)envc * (u32)nlm + (u32)min(lane, nlm - 1));
    }
    combias = at(io.dr_base_com, (u32)envc * 3u + (u32)min(lane, 2));
    fric = io.dr_friction[envc];
    __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise sinks (2) below the wait for (1))
    STAMPB(6);
    if (!fk_jump) {
      const int n = Bx * SKC_W, wl = threadIdx.x & 127;
#pragma unroll
      for (int u = 0; u < SKC_REGSB; ++u) { const int i = wl + u * 128; if (i < n) skc[i] = skregB[u]; }
      LDS_BARRIER();                                            // bar0
    }
    LOAD_CLIP_META_OF_ID();                                     // (a library: one more dependent trip, clip id -> clip meta, with (2) in flight)
    // reference rows: address from the env scalars, loads issued now, consumed in phase D
    float blend = 0.0f;
    const float* r0 = tbl.frames;
    const float* r1 = tbl.frames;
    f3 rp0 = mk3(0, 0, 0), rp1 = rp0, rv0 = rp0, rv1 = rp0, rw0 = rp0, rw1 = rp0;
    f4 rq0 = mk4(0, 0, 0, 1), rq1 = rq0;
    float rd0 = 0, rd1 = 0, rdv0 = 0, rdv1 = 0, rc0 = 0, rc1 = 0;
    tref = (float)(ep1 + 1) * dt + start;                       // motion_tracking.py:554,588
    if (io.ref_time_out && valid && lane == 0) io.ref_time_out[env] = tref;     // (what a lazy extras["ref_body_*_extend"] is rebuilt from)
    {
      int f0, f1;
      frame_blend(tref, m_len, m_nf, m_dt, &f0, &f1, &blend);
      r0 = tbl.frames + (size_t)(m_row0 + f0) * tbl.row;
      r1 = tbl.frames + (size_t)(m_row0 + f1) * tbl.row;
      const int lb = min(lane, Bx - 1), lc = min(lane, 1);
      rp0 = ld3(r0 + o_pos + 3 * lb); rp1 = ld3(r1 + o_pos + 3 * lb);
      rq0 = ld4(r0 + o_rot + 4 * lb); rq1 = ld4(r1 + o_rot + 4 * lb);
      rv0 = ld3(r0 + o_vel + 3 * lb); rv1 = ld3(r1 + o_vel + 3 * lb);
      rw0 = ld3(r0 + o_ang + 3 * lb); rw1 = ld3(r1 + o_ang + 3 * lb);
      rd0 = r0[dc]; rd1 = r1[dc]; rdv0 = r0[D + dc]; rdv1 = r1[D + dc];
      rc0 = r0[2 * D + lc]; rc1 = r1[2 * D + lc];
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMPB(7);
#ifndef PBHC_ABL_RNG
    if (c.randomize_torque_rfi) u_rfi = io.u_rfi ? 0.5f : rng_uniform(rt.seed, env, step_ctr, 1, d);     // in-kernel draw: computed while the loads fly
#endif
    STAMPB(8);
    if (valid) {
      // contact forces and the previous contacts
      {
        const int nq = (B * 3) >> 2;
        if (lane < nq) *reinterpret_cast<float4*>(cf + 4 * lane) = creg4;
        if (lane < ((B * 3) & 3)) cf[4 * nq + lane] = cregr;
      }
      if (lane < NF) { misc[M_FAT0 + lane] = fat; misc[M_LASTC0 + lane] = lastc; }
      if (lane == 0 && NF < 2) { misc[M_FAT1] = 0.0f; misc[M_LASTC1] = 0.0f; }
    }
    WAVE_LDS_FENCE();
    // =============== role B, interval 1: per-env scalars, reference frame ============================================================
    // ---- contacts (legged_robot_base.py:371-380); the other per-env scalars are role A's (behind its FK chain)
    if (valid && lane < NF) {
      const int f = lane;
      float cn = norm3(ld3(cf + 3 * c.feet[f])) > 1.0f ? 1.0f : 0.0f;
      misc[M_CONTACT0 + f] = cn;
      misc[M_CFILT0 + f] = (cn != 0.0f || lastc != 0.0f) ? 1.0f : 0.0f;
    }
    if (c.terminate_by_contact) {                              // legged_robot_base.py:434-436
      float hit = 0.0f;
      if (valid)
        for (int i = lane; i < c.num_term_contact; i += PBHC_G)
          if (norm3(ld3(cf + 3 * c.term_contact[i])) > 1.0f) hit = 1.0f;
      hit = group_max(hit);
      if (valid && lane == 0) misc[M_TCONTACT] = hit;
    }
    STAMPB(1);
    // ---- phase D: reference frame: lerp / slerp of the two frame rows (MotionLibBase.get_motion_state motion_lib_base.py:123-259)
#ifdef PBHC_ABL_PHASED
    if (false) {
#else
    if (valid) {
#endif
      const float a = 1.0f - blend, bb = blend;
      if (lane < D) { rdof[lane] = a * rd0 + bb * rd1; rdofv[lane] = a * rdv0 + bb * rdv1; }
      if (lane < 2) {
        misc[M_RCONTACT0 + lane] = a * rc0 + bb * rc1;
        if (MODE) feat[c.feat_off[PBHC_F_REF_CONTACT_MASK] + lane] = a * rc0 + bb * rc1;
      }
      if (lane == 0) {       // for role A's reset path (phase G, after bar2): the env origin and the clip meta without another memory round trip
        misc[M_ORIGIN0] = origin.x; misc[M_ORIGIN1] = origin.y; misc[M_ORIGIN2] = origin.z;
        misc[M_CLIP_LEN] = m_len; misc[M_CLIP_DT] = m_dt; misc[M_CLIP_NF] = __int_as_float(m_nf); misc[M_CLIP_ROW0] = __int_as_float(m_row0);
      }
      if (lane < Bx) {
        st3(rp + 3 * lane, mk3(a * rp0.x + bb * rp1.x + origin.x, a * rp0.y + bb * rp1.y + origin.y, a * rp0.z + bb * rp1.z + origin.z));
        st4(rq + 4 * lane, slerp(rq0, rq1, bb));
        st3(rv + 3 * lane, mk3(a * rv0.x + bb * rv1.x, a * rv0.y + bb * rv1.y, a * rv0.z + bb * rv1.z));
        st3(rw + 3 * lane, mk3(a * rw0.x + bb * rw1.x, a * rw0.y + bb * rw1.y, a * rw0.z + bb * rw1.z));
      }
      for (int i = lane + PBHC_G; i < Bx; i += PBHC_G) {            // bodies beyond the 32 lanes (29-DoF robots)
        f3 p0 = ld3(r0 + o_pos + 3 * i), p1 = ld3(r1 + o_pos + 3 * i);
        st3(rp + 3 * i, mk3(a * p0.x + bb * p1.x + origin.x, a * p0.y + bb * p1.y + origin.y, a * p0.z + bb * p1.z + origin.z));
        st4(rq + 4 * i, slerp(ld4(r0 + o_rot + 4 * i), ld4(r1 + o_rot + 4 * i), bb));
        f3 v0 = ld3(r0 + o_vel + 3 * i), v1 = ld3(r1 + o_vel + 3 * i);
        st3(rv + 3 * i, mk3(a * v0.x + bb * v1.x, a * v0.y + bb * v1.y, a * v0.z + bb * v1.z));
        f3 w0 = ld3(r0 + o_ang + 3 * i), w1 = ld3(r1 + o_ang + 3 * i);
        st3(rw + 3 * i, mk3(a * w0.x + bb * w1.x, a * w0.y + bb * w1.y, a * w0.z + bb * w1.z));
      }
    }
    STAMPB(2);
#ifndef PBHC_EARLY_OPERANDS
    LOAD_STEP_OPERANDS()
#endif
    // ---- the history row (40 % of this role's loaded bytes, read by nobody before bar2) is REQUESTED here, behind the loads that bar1 waits
    // for — issued with them it sat in the same in-order queue and the burst of all workgroups' prologues landed ~2 k cycles later —
    // and copied to the feature row after bar1
    {
      const u32 hbase = (u32)envc * (u32)(io.hist_pitch ? io.hist_pitch : c.hist_dim);
      const int hlast = c.hist_dim - 1;
#pragma unroll
#ifndef PBHC_ABL_HIST
      for (int u = 0; u < PBHC_HREG; ++u)
        if (!hist_b) hreg[u] = at(io.hist, hbase + (u32)min(lane + u * PBHC_G, hlast));
      if (hist_b) {
        // (rows are 16-byte aligned with a pitch that is a multiple of 4 — checked by the host: hist_wide — so the last quad may reach into
        // the row's padding; quads past it repeat the last one)
        const int nq = (c.hist_dim + 3) >> 2;
#pragma unroll
        for (int u = 0; u < PBHC_HREG4; ++u)
          if (u * PBHC_G < nq) {
            const pbhc_f32x4 hv = NTLD(*reinterpret_cast<const pbhc_f32x4*>(&at(io.hist, hbase + 4u * (u32)min(lane + u * PBHC_G, nq - 1))));
            hreg4[u] = make_float4(hv[0], hv[1], hv[2], hv[3]);
          }
      }
#else
      for (int u = 0; u < PBHC_HREG; ++u) hreg[u] = 0.0f;
      for (int u = 0; u < PBHC_HREG4; ++u) hreg4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    }
    // per-dof constants of the config: one batch of loads at the head of the interval
    k_tl = c.torque_limits[dc]; k_dp = io.default_dof_pos ? at(io.default_dof_pos, eDc + dc) : c.default_dof_pos[dc];
    const float k_pg = c.p_gains[dc], k_dg = c.d_gains[dc], k_as = c.action_scale[dc];
    // ---- _pre_physics_step (motion_tracking.py:749-768) and the torques from the pre-step state (legged_robot_base.py:795-838)
    if (c.randomize_torque_rfi && io.u_rfi) u_rfi = u_inj;
    if (valid) {
      if (d < D) {
        const float tl = k_tl;
        const float a = clampf(a_in, -c.action_clip_value, c.action_clip_value);
        if (fabsf(a) == c.action_clip_value) clipcnt += 1.0f;
        act[d] = a;
        float delayed = a;
        if (c.randomize_ctrl_delay) {            // queue[k] <- queue[k-1], queue[0] <- a ; delayed = queue[delay_idx]
#pragma unroll
          for (int k = 0; k < PBHC_MAX_QUEUE; ++k)
            if (k < Q) {
              float nv = (k == 0) ? a : qold[k > 0 ? k - 1 : 0];
              NTST(at(io.action_queue, qoff + (u32)(k * D)), nv);
              if (k == didx) delayed = nv;
            }
        }
        actd[d] = delayed;
        const float a_sc = delayed * k_as;
        float tq;
        if (c.control_type == 0) tq = kp * k_pg * (a_sc + k_dp - qp) - kd * k_dg * qv;
        else if (c.control_type == 1) tq = kp * k_pg * (a_sc - qv) - kd * k_dg * (qv - pf_last_qd) / c.sim_dt;      // "V" (legged_robot_base.py:812-813)
        else tq = a_sc;                                                                                                 // "T" (:814-815)
        if (c.randomize_torque_rfi) tq = tq + (u_rfi * 2.0f - 1.0f) * c.rfi_lim * rfs * tl;
        if (c.use_rao) tq = tq + ras * tl;
        if (c.clip_torques) tq = clampf(tq, -tl, tl);
        tau[d] = tq;
      }
      if (lane < 3) feat[c.feat_off[PBHC_F_DR_BASE_COM] + lane] = combias;
      if (lane < c.dr_link_mass_dim) feat[c.feat_off[PBHC_F_DR_LINK_MASS] + lane] = lmreg;
      if (lane == 0) {
        feat[c.feat_off[PBHC_F_DR_FRICTION]] = fric;
        feat[c.feat_off[PBHC_F_ZERO]] = 0.0f;
        if (MODE) feat[c.feat_off[PBHC_F_DR_BASE_MASS]] = bmass;
      }
    }
    clipcnt = group_sum(clipcnt);
    WAVE_LDS_FENCE();
    // ---- phase C, this role's share: the reference time / phase (motion_tracking.py:554,588).  The scalars derived from the root state
    // (heading, base-frame velocities, gravity: legged_robot_base.py:346-380) are the dynamics waves' since round 4 — with the chain
    // evaluated by pointer jumping those waves reach bar1 first (the reverse of round 3).
    if (valid && lane < 13) root[lane] = frootB;
    if (valid && lane == 7) {
      const float t = (float)(ep1 + 1) * dt + start;
      misc[M_EPLEN] = (float)ep1;
      misc[M_START] = start; misc[M_MLEN] = mlenB;
      feat[c.feat_off[PBHC_F_REF_MOTION_PHASE]] = t / mlenB;
    }
    WAVE_LDS_FENCE();
    STAMPB(3);
  }
  LDS_BARRIER();                                               // bar1: A's body state and B's reference frame / scalars are in LDS
  STAMP(3);

  if (!roleB) {
    // =============== role A, interval 2a: tracking differences over the bodies + lane-parallel partial sums ======================
    // (motion_tracking.py:645-731 and the body-space reductions of the _reward_* terms)
    if (valid) {
      f4 hinv = ld4(misc + M_HINV);
      f3 rootp = ld3(root);
      // general tracking: anchor ("beyondmimic") frame, general_tracking.py:738-767 — every lane derives it redundantly (no LDS
      // round trip).  delta_pos aliases robot_anchor_pos in the reference (:748-749): (robot x, robot y, REF z) is used by both.
      f4 a_rq = mk4(0, 0, 0, 1), a_bq = a_rq, dori = a_rq, ainv = a_rq;
      f3 a_rp = mk3(0, 0, 0), a_bp = a_rp, dpos = a_rp;
      if (MODE) {
        const int an = c.anchor_index;
        a_rq = ld4(rq + 4 * an); a_bq = ld4(bq + 4 * an); a_rp = ld3(rp + 3 * an); a_bp = ld3(bp + 3 * an);
        dori = yaw_quat(quat_mul(a_bq, quat_conj(a_rq)));
        dpos = mk3(a_bp.x, a_bp.y, a_rp.z);
        ainv = quat_conj(a_bq);
      }
      const int o_lbp = c.feat_off[PBHC_F_LOCAL_BODY_POS], o_lbr = c.feat_off[PBHC_F_LOCAL_BODY_ROT];
      const int o_dif = c.feat_off[PBHC_F_DIF_LOCAL_RIGID_BODY_POS], o_loc = c.feat_off[PBHC_F_LOCAL_REF_RIGID_BODY_POS];
      const int o_vr = c.feat_off[PBHC_F_VR_3POINT_POS], o_lv = c.feat_off[PBHC_F_LOCAL_REF_RIGID_BODY_VEL], o_gv = c.feat_off[PBHC_F_GLOBAL_REF_RIGID_BODY_VEL];
#ifdef PBHC_ABL_EBODY
      for (int b = lane; b < 0; b += PBHC_G) {
#else
      for (int b = lane; b < Bx; b += PBHC_G) {
#endif
        f3 rpos = ld3(rp + 3 * b);
        f3 dp = sub3(rpos, ld3(bp + 3 * b));
        float n2 = dp.x * dp.x + dp.y * dp.y + dp.z * dp.z;
        float msq = n2 * (1.0f / 3.0f);
        float nrm = sqrtf(n2);
        int fl = c.body_flags[b];
        if (fl & 1) { s_up += msq; s_upn += nrm; }
        if (fl & 2) { s_lo += msq; s_lon += nrm; }
        if (fl & 4) { s_vr += msq; s_vrn += nrm; }
        if (fl & 8) s_feet += msq;
        s_maxn = fmaxf(s_maxn, nrm);
        f4 dq = ld4(rq + 4 * b), cq = ld4(bq + 4 * b);
        f3 dv = sub3(ld3(rv + 3 * b), ld3(bv + 3 * b));
        const float dv2 = (dv.x * dv.x + dv.y * dv.y + dv.z * dv.z) * (1.0f / 3.0f);
        s_vel += dv2;
        f3 dw3 = sub3(ld3(rw + 3 * b), ld3(bw + 3 * b));
        const float dw2 = (dw3.x * dw3.x + dw3.y * dw3.y + dw3.z * dw3.z) * (1.0f / 3.0f);
        s_ang += dw2;
        if (!MODE) {
          float dx = dq.x - cq.x, dy = dq.y - cq.y, dz = dq.z - cq.z, dw = dq.w - cq.w;   // quaternion SUBTRACTION, sic (motion_tracking.py:651)
          s_rot += (dx * dx + dy * dy + dz * dz + dw * dw) * 0.25f;
        } else {
          // true quaternion difference + its angle (general_tracking.py:643-647,1144,1203); anchor-relative target :750-767
          const f3 bpos = ld3(bp + 3 * b);
          const float ang = quat_angle(quat_mul(dq, quat_conj(cq)));
          s_rot += ang * ang;
          if (b == c.anchor_index) { red[R_AROT] = ang * ang; red[R_APOS] = msq; }
          const f3 dl = sub3(add3(dpos, quat_apply(dori, sub3(rpos, a_rp))), bpos);
          const float l2 = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z, lnrm = sqrtf(l2);
          const float lang = quat_angle(quat_mul(quat_mul(dori, dq), quat_conj(cq)));
          if (fl & 16) {
            s_key += msq; s_keyn += nrm; s_lkey += l2 * (1.0f / 3.0f); s_lkeyn += lnrm;
            s_lkrot += lang * lang; s_kvel += dv2; s_kang += dw2;
          }
          if (fl & 1) s_lupn += lnrm;
          if (fl & 2) s_llon += lnrm;
          if (fl & 4) s_lvrn += lnrm;
          if ((fl & 32) && fabsf(dl.z) > c.body_z_threshold) s_bodyz = 1.0f;
          st3(feat + o_lbp + 3 * b, quat_apply(ainv, sub3(bpos, dpos)));           // :779-782
          quat_to_mat6(quat_mul(ainv, cq), feat + o_lbr + 6 * b);                 // :771-778
        }
        st3(feat + o_dif + 3 * b, quat_rotate(hinv, dp));
        f3 gl = sub3(rpos, rootp);
        f3 loc = quat_rotate(hinv, gl);
        st3(feat + o_loc + 3 * b, loc);
        if (c.track_slot[b] >= 0) st3(feat + o_vr + 3 * c.track_slot[b], loc);
        f3 rvel = ld3(rv + 3 * b);
        st3(feat + o_gv + 3 * b, rvel);
        st3(feat + o_lv + 3 * b, quat_rotate(hinv, rvel));
      }
      if (MODE && lane == 0) {
        // root differences (general_tracking.py:655-666) and the anchor observations / termination signals (:784-803)
        const f4 rootq = ld4(root + 3);
        const f3 drv = sub3(quat_rotate_inverse(ld4(rq), ld3(rv)), quat_rotate_inverse(rootq, ld3(root + 7)));
        st3(feat + c.feat_off[PBHC_F_DIF_ROOT_VELOCITY], drv);
        red[R_RVEL] = (drv.x * drv.x + drv.y * drv.y + drv.z * drv.z) / 3.0f;
        const f4 drr = quat_mul(ld4(rq), quat_conj(rootq));
        st4(feat + c.feat_off[PBHC_F_DIF_ROOT_ROT], drr);
        const float drh = rp[2] - root[2];
        feat[c.feat_off[PBHC_F_DIF_ROOT_HEIGHT]] = drh;
        const float ra = quat_angle(drr);
        red[R_RPOSE] = ra * ra + drh * drh;
        quat_to_mat6(quat_mul(ainv, a_rq), feat + c.feat_off[PBHC_F_ANCHOR_REF_ROT]);
        st3(feat + c.feat_off[PBHC_F_ANCHOR_REF_POS], quat_apply(ainv, sub3(a_rp, a_bp)));
        const f3 gv = mk3(0.0f, 0.0f, -1.0f);
        misc[M_ADZ] = a_rp.z - a_bp.z;
        misc[M_AORI] = quat_rotate_inverse(a_rq, gv).z - quat_rotate_inverse(a_bq, gv).z;
      }
    }
#define GSUM(v) v = group_sum(v)
    STAMP(23);
    GSUM(s_up); GSUM(s_lo); GSUM(s_vr); GSUM(s_feet); GSUM(s_rot); GSUM(s_vel); GSUM(s_ang); GSUM(s_upn); GSUM(s_lon); GSUM(s_vrn);
    s_maxn = group_max(s_maxn);
    if (MODE) {
      GSUM(s_key); GSUM(s_keyn); GSUM(s_lkey); GSUM(s_lkeyn); GSUM(s_lkrot); GSUM(s_kvel); GSUM(s_kang); GSUM(s_lupn); GSUM(s_llon); GSUM(s_lvrn);
      s_bodyz = group_max(s_bodyz);
    }
    STAMP(24);
    // the means over body sets: multiplications by the set sizes' reciprocals (literals in the specialised build; a correctly rounded
    // division is ~10 instructions, and this lane-0 block sits on the chain — the quotient differs from x / n by <= 1 ulp)
    const float inv_upper = 1.0f / (float)c.num_upper, inv_lower = 1.0f / (float)c.num_lower, inv_track = 1.0f / (float)c.num_track;
    const float inv_feet = 1.0f / (float)NF, inv_bx = 1.0f / (float)Bx, inv_key = MODE ? 1.0f / (float)c.num_key : 1.0f;
    if (valid && lane == 0) {
      if (MODE) {
        red[R_KEY] = s_key * inv_key; red[R_KEYN] = s_keyn * inv_key; red[R_LKEY] = s_lkey * inv_key; red[R_LKEYN] = s_lkeyn * inv_key; red[R_LKROT] = s_lkrot * inv_key;
        red[R_KVEL] = s_kvel * inv_key; red[R_KANG] = s_kang * inv_key;
        red[R_LUPN] = s_lupn * inv_upper; red[R_LLON] = s_llon * inv_lower; red[R_LVRN] = s_lvrn * inv_track;
      }
      red[R_UP] = s_up * inv_upper; red[R_LO] = s_lo * inv_lower; red[R_VR] = s_vr * inv_track;
      red[R_FEET] = s_feet * inv_feet; red[R_ROT] = s_rot * inv_bx; red[R_VEL] = s_vel * inv_bx; red[R_ANG] = s_ang * inv_bx;
      red[R_MAXNORM] = s_maxn; red[R_UPN] = s_upn * inv_upper; red[R_LON] = s_lon * inv_lower; red[R_VRN] = s_vrn * inv_track;
      if (!MODE) for (int k = PBHC_S_KEY_BODY_POS; k < PBHC_NUM_SIGMA; ++k) red[R_ERR0 + k] = 0.0f;
      // ---- _check_termination (legged_robot_base.py:408-489, motion_tracking.py:330-357)
      float grav = 0.0f, far = 0.0f, tlen = 0.0f, tend = 0.0f;
      if (c.terminate_by_gravity) grav = sqrtf(misc[M_GX] * misc[M_GX] + misc[M_GY] * misc[M_GY]) > c.termination_gravity ? 1.0f : 0.0f;
      if (c.terminate_when_motion_far) far = s_maxn > pf_far_thr ? 1.0f : 0.0f;
      tlen = misc[M_EPLEN] > c.max_episode_length ? 1.0f : 0.0f;
      if (c.terminate_when_motion_end) tend = (misc[M_EPLEN] * dt + misc[M_START]) > misc[M_MLEN] ? 1.0f : 0.0f;
      float tout = (tlen != 0.0f || tend != 0.0f) ? 1.0f : 0.0f;
      float tcontact = 0.0f, tlowh = 0.0f;
      if (c.terminate_by_contact) tcontact = misc[M_TCONTACT];
      if (c.terminate_by_low_height) tlowh = root[2] < c.termination_min_base_height ? 1.0f : 0.0f;         // :442-444
      misc[M_TCONTACT] = tcontact; misc[M_TLOWH] = tlowh;
      float refz = 0.0f, refori = 0.0f, bodyz = 0.0f;
      if (MODE) {                                   // general_tracking.py:241-254
        if (c.terminate_by_ref_pos_z) refz = fabsf(misc[M_ADZ]) > c.ref_pos_z_threshold ? 1.0f : 0.0f;
        if (c.terminate_by_ref_ori) refori = fabsf(misc[M_AORI]) > c.ref_ori_threshold ? 1.0f : 0.0f;
        if (c.terminate_by_body_z) bodyz = s_bodyz;
        misc[M_REFZ] = refz; misc[M_REFORI] = refori; misc[M_BODYZ] = bodyz;
      }
      misc[M_GRAV] = grav; misc[M_FAR] = far; misc[M_END] = tend; misc[M_TOUT_LEN] = tlen;
      misc[M_TIMEOUT] = tout;
      misc[M_RESET] = (grav != 0.0f || far != 0.0f || tout != 0.0f || refz != 0.0f || refori != 0.0f || bodyz != 0.0f || tcontact != 0.0f || tlowh != 0.0f) ? 1.0f : 0.0f;
#ifdef PBHC_ABL_NORESET
      misc[M_RESET] = 0.0f;
#endif
    }
    WAVE_LDS_FENCE();
    STAMP(4);
  } else {
    // =============== role B, interval 2a: pre-physics step + torques, joint-space differences + reductions, foot norms, the
    // post-reset features of a NON-terminated env (what phase H of role A computes after a reset), the observation maps -> LDS ========
#ifdef PBHC_ABL_HIST
    if (false) {
#else
    if (valid && !hist_b) {                                      // the history row -> feature row (hist_b: it stays in registers until bar2)
#endif
#pragma unroll
      for (int u = 0; u < PBHC_HREG; ++u) { const int i = lane + u * PBHC_G; if (i < c.hist_dim) feat[hoff + i] = hreg[u]; }
      if (c.hist_dim > PBHC_HREG * PBHC_G)
        copy_g2l(feat + hoff + PBHC_HREG * PBHC_G, io.hist + (size_t)env * (io.hist_pitch ? io.hist_pitch : c.hist_dim) + PBHC_HREG * PBHC_G, c.hist_dim - PBHC_HREG * PBHC_G, lane);
    }
    uint32_t mreg[PBHC_MAPREG];
    if (map_words > 0) {
      const int wl = threadIdx.x & (2 * PBHC_G * 2 - 1);            // 0..127 over the two role-B waves
#pragma unroll
      for (int u = 0; u < PBHC_MAPREG; ++u) mreg[u] = map_img[min(wl + u * 128, map_words - 1)];
    }
    const float k_vl = c.dof_vel_limits[dc];
    const float k_lo = c.soft_pos_curriculum ? c.hard_dof_pos_limits[dc][0] : c.soft_dof_pos_limits[dc][0];
    const float k_hi = c.soft_pos_curriculum ? c.hard_dof_pos_limits[dc][1] : c.soft_dof_pos_limits[dc][1];
    float s_maxjp = 0, s_jp2 = 0, s_jv2 = 0, s_tau2 = 0, s_ar = 0, s_qd2 = 0, s_qacc2 = 0, s_lpos = 0, s_lvel = 0, s_ltau = 0, s_coll = 0;
    float g_pos = 0.0f, g_vel = 0.0f, g_tau = 0.0f;
    if (valid) {
      // the optional simulator-surface outputs (the rigid-body arrays are re-used after bar2: step_lds_plan)
      if (io.rigid_body_state)
        for (int b = lane; b < B; b += PBHC_G) {
          float* o = &at(io.rigid_body_state, ((u32)env * (u32)B + (u32)b) * 13u);
          st3(o, ld3(bp + 3 * b)); st4(o + 3, ld4(bq + 4 * b)); st3(o + 7, ld3(bv + 3 * b)); st3(o + 10, ld3(bw + 3 * b));
        }
      if (io.contact_forces)
        for (int i = lane; i < B * 3; i += PBHC_G) at(io.contact_forces, (u32)env * (u32)(B * 3) + (u32)i) = cf[i];
      // outputs of the PRE-reset reference (a reset rewrites the root entries of rp / rq after bar2)
      if (io.ref_body_pos_extend)
        for (int i = lane; i < Bx * 3; i += PBHC_G) at(io.ref_body_pos_extend, (u32)env * (u32)(Bx * 3) + (u32)i) = rp[i];
      if (io.ref_body_rot_extend)
        for (int i = lane; i < Bx * 4; i += PBHC_G) at(io.ref_body_rot_extend, (u32)env * (u32)(Bx * 4) + (u32)i) = rq[i];
      const float soft_pos = (float)glob[PBHC_G_SOFT_POS_VAL], soft_vel = (float)glob[PBHC_G_SOFT_VEL_VAL], soft_tau = (float)glob[PBHC_G_SOFT_TAU_VAL];
      const float inv_dt = 1.0f / dt;
      const int o_dja = c.feat_off[PBHC_F_DIF_JOINT_ANGLES], o_djv = c.feat_off[PBHC_F_DIF_JOINT_VELOCITIES];
#ifdef PBHC_ABL_JOINT
      if (false) {
#else
      if (d < D) {
#endif
        const int dd = d;
        float dj = rdof[dd] - q[dd], djv = rdofv[dd] - qd[dd];
        feat[o_dja + dd] = dj; feat[o_djv + dd] = djv;
        s_maxjp = fmaxf(s_maxjp, fabsf(dj));
        s_jp2 += dj * dj; s_jv2 += djv * djv;
        s_tau2 += tau[dd] * tau[dd];
        float la = pf_last_act - act[dd];
        s_ar += la * la;
        s_qd2 += qd[dd] * qd[dd];
        float acc = (pf_last_qd - qd[dd]) * inv_dt;
        s_qacc2 += acc * acc;
        float lo_l = k_lo, hi_l = k_hi;
        if (c.soft_pos_curriculum) {
          float m = (k_lo + k_hi) / 2.0f;
          float r = k_hi - k_lo;
          lo_l = m - 0.5f * r * soft_pos; hi_l = m + 0.5f * r * soft_pos;
        }
        s_lpos += -fminf(q[dd] - lo_l, 0.0f) + fmaxf(q[dd] - hi_l, 0.0f);
        float vlim = k_vl * (c.soft_vel_curriculum ? soft_vel : c.soft_dof_vel_limit);
        s_lvel += clampf(fabsf(qd[dd]) - vlim, 0.0f, 1.0f);
        if (c.soft_tau_curriculum) s_ltau += clampf(fabsf(tau[dd]) - k_tl * soft_tau, 0.0f, 1.0f);
        else s_ltau += fmaxf(fabsf(tau[dd]) - k_tl * c.soft_torque_limit, 0.0f);
        if (close_any) {                                      // legged_robot_base.py:449-479 (the gates' per-step draws: below)
          if (c.terminate_close_pos && (q[dd] < c.dof_pos_limits_termination[dd][0] || q[dd] > c.dof_pos_limits_termination[dd][1])) g_pos = 1.0f;
          if (c.terminate_close_vel && fabsf(qd[dd]) - k_vl * c.term_close_vel_scale > 0.0f) g_vel = 1.0f;
          if (c.terminate_close_tau && fabsf(tau[dd]) - k_tl * c.term_close_tau_scale > 0.0f) g_tau = 1.0f;
        }
      }
      for (int i = lane; i < c.num_penalised; i += PBHC_G)
        if (norm3(ld3(cf + 3 * c.penalised[i])) > 0.1f) s_coll += 1.0f;
      if (lane >= PBHC_G - NF) {                              // per-foot norms for the contact rewards (phase F of role A)
        const int f = lane - (PBHC_G - NF);
        const float* fc = cf + 3 * c.feet[f];
        const float* fv = bv + 3 * c.feet[f];
        red[R_FOOT0 + 4 * f + 0] = norm3(ld3(fc));
        red[R_FOOT0 + 4 * f + 1] = sqrtf(fc[0] * fc[0] + fc[1] * fc[1]);
        red[R_FOOT0 + 4 * f + 2] = fc[2];
        red[R_FOOT0 + 4 * f + 3] = norm3(ld3(fv));
        red[R_FOOT0 + 8 + f] = sqrtf(fv[0] * fv[0] + fv[1] * fv[1]);
        if (c.foot_ori_terms) {
          // per-foot heading against the root's (feet_heading_alignment*, legged_robot_base.py:1030-1045,1054-1069) and tilt
          // (penalty_feet_ori*, :1047-1052,1071-1079): torch_utils.quat_apply of the forward axis, atan2, wrap_to_pi (rotations.py:50-53)
          const f4 fq = ld4(bq + 4 * c.feet[f]);
          const f3 ff = quat_apply(fq, mk3(1.0f, 0.0f, 0.0f)), rf = quat_apply(ld4(root + 3), mk3(1.0f, 0.0f, 0.0f));
          float dh = fmodf(atan2f(ff.y, ff.x) - atan2f(rf.y, rf.x), 6.2831855f);       // torch's %: the sign of the divisor
          if (dh < 0.0f) dh += 6.2831855f;
          if (dh > 3.1415927f) dh -= 6.2831855f;
          red[R_FOOT0 + 10 + f] = fabsf(dh);
          const f3 fg = quat_rotate_inverse(fq, mk3(0.0f, 0.0f, -1.0f));
          red[R_FOOT0 + 12 + f] = sqrtf(fg.x * fg.x + fg.y * fg.y);
        }
      }
    }
    GSUM(s_jp2); GSUM(s_jv2); GSUM(s_tau2); GSUM(s_ar); GSUM(s_qd2); GSUM(s_qacc2); GSUM(s_lpos); GSUM(s_lvel); GSUM(s_ltau); GSUM(s_coll);
    s_maxjp = group_max(s_maxjp);
    if (close_any) {
      // one uniform per gate and STEP (torch.rand(1) < p on the host in the reference): keyed on the step counter only, the same for every env
      float ug[4];
      pbhc::rng_uniform4(rt.seed, 0xFFFFFFFFu, step_ctr, 11, 0, ug);
      if (io.ovr_gate_u) { ug[0] = io.ovr_gate_u[0]; ug[1] = io.ovr_gate_u[1]; ug[2] = io.ovr_gate_u[2]; }
      g_pos = (ug[0] < c.term_close_prob[0]) ? group_max(g_pos) : 0.0f;
      g_vel = (ug[1] < c.term_close_prob[1]) ? group_max(g_vel) : 0.0f;
      g_tau = (ug[2] < c.term_close_prob[2]) ? group_max(g_tau) : 0.0f;
      gateB = valid && (g_pos != 0.0f || g_vel != 0.0f || g_tau != 0.0f);
      if (valid && lane == 0) { misc[M_TPOSLIM] = g_pos; misc[M_TVELLIM] = g_vel; misc[M_TTAULIM] = g_tau; misc[M_TGATE] = gateB ? 1.0f : 0.0f; }
    }
    if (valid && lane == 0) {
      red[R_MAXJP] = s_maxjp; red[R_JP2] = s_jp2; red[R_JPM] = s_jp2 / (float)D; red[R_JVM] = s_jv2 / (float)D; red[R_TAU2] = s_tau2; red[R_ARATE] = s_ar; red[R_QD2] = s_qd2;
      red[R_QACC2] = s_qacc2; red[R_LIMPOS] = s_lpos; red[R_LIMVEL] = s_lvel; red[R_LIMTAU] = s_ltau; red[R_COLL] = s_coll; red[R_CLIPCNT] = clipcnt;
    }
    // post-reset features as they stand WITHOUT a reset (phase H recomputes them for a terminated env, after bar2): with them every
    // observation element of a surviving env can be written while the dynamics chain is still in its reward phase
    if (valid) {
      const int o_q = c.feat_off[PBHC_F_DOF_POS], o_qd = c.feat_off[PBHC_F_DOF_VEL], o_a = c.feat_off[PBHC_F_ACTIONS];
      const int o_kp = c.feat_off[PBHC_F_DR_KP], o_kd = c.feat_off[PBHC_F_DR_KD];
      if (d < D) {
        feat[o_q + d] = q[d] - k_dp;
        feat[o_qd + d] = qd[d];
        feat[o_a + d] = act[d];
        feat[o_kp + d] = kp;
        feat[o_kd + d] = kd;
      }
      if (lane == 0) {
        feat[c.feat_off[PBHC_F_DR_CTRL_DELAY]] = (float)adelayB;
        feat[c.feat_off[PBHC_F_BASE_POS_Z]] = root[2];
      }
      if (MODE && lane < NF) feat[c.feat_off[PBHC_F_CONTACT_MASK] + lane] = misc[M_CFILT0 + lane];
    }
    if (map_words > 0) {
      const int wl = threadIdx.x & (2 * PBHC_G * 2 - 1);
#pragma unroll
      for (int u = 0; u < PBHC_MAPREG; ++u) { const int i = wl + u * 128; if (i < map_words) mapl[i] = mreg[u]; }
      for (int i = wl + PBHC_MAPREG * 128; i < map_words; i += 128) mapl[i] = map_img[i];
    }
    STAMPB(4);
  }
  LDS_BARRIER();                                               // bar2: both halves of the reduction row + termination flags are in LDS
  STAMP(5);
  if (close_any && !roleB) {                                   // role A: role B's causes join the reset flag (every later reader of M_RESET in this wave follows)
    if (valid && lane == 0 && misc[M_TGATE] != 0.0f) misc[M_RESET] = 1.0f;
    WAVE_LDS_FENCE();
  }
  // a terminated env's history is zero in the observations of this very step (history_handler.py:33-38 via reset_envs_idx).  The copy of
  // the history row into the feature row is role B's, between bar1 and bar2; each role zeroes it for itself here, ahead of its own
  // observation passes (both store zeros: no order between the two is needed)
  if (hist_b) {
    // the old history leaves the reference waves' registers only now, over the simulator body arrays (nobody reads those after bar2);
    // every row that holds history is written by these waves, so the hand-over is wave-local
    if (roleB && valid) {
      const bool z = misc[M_RESET] != 0.0f || gateB;
      {
        const int nq = (c.hist_dim + 3) >> 2;
#pragma unroll
        for (int u = 0; u < PBHC_HREG4; ++u) {
          const int qd_ = lane + u * PBHC_G;
          if (u * PBHC_G < nq && qd_ < nq) *reinterpret_cast<float4*>(histl + 4 * qd_) = z ? make_float4(0.f, 0.f, 0.f, 0.f) : hreg4[u];
        }
      }
    }
    WAVE_LDS_FENCE();
  } else if (valid && (misc[M_RESET] != 0.0f || (roleB && gateB))) {
    for (int i = lane; i < c.hist_dim; i += PBHC_G) histl[i] = 0.0f;
    WAVE_LDS_FENCE();
  }

  if (roleB) nzb[0] = __float_as_uint(misc[M_NZB]);        // (the dynamics waves' Philox word; they keep their own copy in registers)
  if (dr_on_b && roleB && valid && (misc[M_RESET] != 0.0f || gateB || io.redraw_all != 0)) {
    // _episodic_domain_randomization of a terminated (or re-drawn) env, per-dof part (legged_robot_base.py:599-631): the same Philox stream
    // and injected-draw overrides as the dynamics waves' form (phase G), the new gains straight into the features the late rows read
    const int o_kp = c.feat_off[PBHC_F_DR_KP], o_kd = c.feat_off[PBHC_F_DR_KD];
    for (int dd = lane; dd < D; dd += PBHC_G) {
      float ur[4];
      pbhc::rng_uniform4(rt.seed, env, step_ctr, 2, dd, ur);
      if (c.randomize_pd_gain) {
        const float kpn = io.ovr_kp ? at(io.ovr_kp, eD + dd) : (c.kp_range[1] - c.kp_range[0]) * ur[0] + c.kp_range[0];
        const float kdn = io.ovr_kd ? at(io.ovr_kd, eD + dd) : (c.kd_range[1] - c.kd_range[0]) * ur[1] + c.kd_range[0];
        at(io.kp_scale, eD + dd) = kpn;
        at(io.kd_scale, eD + dd) = kdn;
        feat[o_kp + dd] = kpn;
        feat[o_kd + dd] = kdn;
      }
      if (c.randomize_rfi_lim)
        at(io.rfi_lim_scale, eD + dd) = io.ovr_rfi_lim ? at(io.ovr_rfi_lim, eD + dd) : (c.rfi_lim_range[1] - c.rfi_lim_range[0]) * ur[2] + c.rfi_lim_range[0];
      if (c.use_rao)
        at(io.rao_scale, eD + dd) = io.ovr_rao ? at(io.ovr_rao, eD + dd) : (c.rao_lim - (-c.rao_lim)) * ur[3] + (-c.rao_lim);
      if (c.randomize_ctrl_delay)
        for (int k = 0; k < Q; ++k) at(io.action_queue, ((u32)env * (u32)Q + (u32)k) * (u32)D + (u32)dd) = 0.0f;     // queue *= 0 (finite values)
    }
    WAVE_LDS_FENCE();
  }
  const float clipobs = c.clip_observations;     // config scalars used inside the store loops live in locals
  const int ngroups = c.num_groups;
  // Observation rows: group g is written by the role the host assigned it to (PbhcOutMap.role, balanced by row width), 32 lanes per env.
  // `late_too`: also the pairs / noisy elements that read post-reset features — valid when this role has them (role A after its phase H;
  // role B for a surviving env, whose no-reset values it wrote itself; for a terminated env role B defers them past bar3).
  const float noise_cur = (float)glob[PBHC_G_NOISE_CURRICULUM];
  // specialised build: the rows as unrolled runs (obs_write_runs); the per-element maps are not even staged
#ifndef PBHC_STATIC_CFG
#define OBS_GROUPS_RUNS(ROLE, LATE_TOO)
#define OBS_GROUPS_LATE_RUNS(ROLE)
#else
#define OBS_GROUPS_RUNS(ROLE, LATE_TOO)                                                                                                \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups || c.groups[g].role != (ROLE)) continue;                                                                  \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    if (LATE_TOO) obs_write_runs<0, RUN_WHO>(c.groups[g], 16 + g, lane, feat, fhist, hoff, outg, ob, clipobs, noise_cur, nzb);      \
    else obs_write_runs<1, RUN_WHO>(c.groups[g], 16 + g, lane, feat, fhist, hoff, outg, ob, clipobs, noise_cur, nzb);               \
  }
#define OBS_GROUPS_HELP_RUNS()                                                                                                      \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups || c.groups[g].role != 1) continue;                                                                       \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    obs_write_runs<0, 2>(c.groups[g], 16 + g, lane, feat, feat, hoff, outg, ob, clipobs, noise_cur, nzb);                           \
  }
#define OBS_GROUPS_LATE_RUNS(ROLE)                                                                                                  \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups || c.groups[g].role != (ROLE)) continue;                                                                  \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    obs_write_runs<2, RUN_WHO>(c.groups[g], 16 + g, lane, feat, fhist, hoff, outg, ob, clipobs, noise_cur, nzb);                    \
  }
#endif
#define OBS_GROUPS_MAP(ROLE, LATE_TOO)                                                                                              \
  for (int g = 0; g < ngroups; ++g) {                                                                                               \
    if (c.groups[g].role != (ROLE)) continue;                                                                                       \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    const uint32_t* mg = mapl + c.groups[g].lds_off;                                                                                \
    const int n_early = (int)mg[34], nlist = n_early + (int)mg[35], nn_early = (int)mg[32], nn = nn_early + (int)mg[33];            \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    if (((pitch_g & 1) == 0) && ((reinterpret_cast<uintptr_t>(outg) & 7) == 0))                                                     \
      obs_write_list<8>(mg, 0, (LATE_TOO) ? nlist : n_early, lane, PBHC_G, feat, outg, ob, c.groups[g].dim, pitch_g, c.groups[g].clip, clipobs); \
    else                                                                                                                            \
      obs_write_list_unaligned(mg, 0, (LATE_TOO) ? nlist : n_early, lane, PBHC_G, feat, outg, ob, c.groups[g].dim, c.groups[g].clip, clipobs);   \
    obs_write_noisy(mg, 0, (LATE_TOO) ? nn : nn_early, lane, PBHC_G, feat, outg, ob, c.groups[g].clip, clipobs, noise_cur, 16 + g, nzb); \
  }
#define OBS_GROUPS_LATE_MAP(ROLE)                                                                                                   \
  for (int g = 0; g < ngroups; ++g) {                                                                                               \
    if (c.groups[g].role != (ROLE)) continue;                                                                                       \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    const uint32_t* mg = mapl + c.groups[g].lds_off;                                                                                \
    const int n_early = (int)mg[34], nlist = n_early + (int)mg[35], nn_early = (int)mg[32], nn = nn_early + (int)mg[33];            \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    if (((pitch_g & 1) == 0) && ((reinterpret_cast<uintptr_t>(outg) & 7) == 0))                                                     \
      obs_write_list<4>(mg, n_early, nlist, lane, PBHC_G, feat, outg, ob, c.groups[g].dim, pitch_g, c.groups[g].clip, clipobs);     \
    else                                                                                                                            \
      obs_write_list_unaligned(mg, n_early, nlist, lane, PBHC_G, feat, outg, ob, c.groups[g].dim, c.groups[g].clip, clipobs);       \
    obs_write_noisy(mg, nn_early, nn, lane, PBHC_G, feat, outg, ob, c.groups[g].clip, clipobs, noise_cur, 16 + g, nzb); \
  }
#ifdef PBHC_ABL_OBSX4
// timing experiment: the same rows, the same bytes, stored 16 bytes per lane straight from the feature row (contents meaningless)
#define OBS_GROUPS(ROLE, LATE_TOO) do {                                                                                             \
  _Pragma("unroll") for (int g = 0; g < PBHC_MAX_GROUPS; ++g) {                                                                     \
    if (g >= c.num_groups || c.groups[g].role != (ROLE)) continue;                                                                  \
    const int pitch_g = io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch;                                                      \
    float* __restrict__ const outg = io.obs[g];                                                                                     \
    const u32 ob = (u32)env * (u32)pitch_g;                                                                                         \
    _Pragma("unroll") for (int q0 = 0; q0 < (c.groups[g].dim + 3) / 4; q0 += PBHC_G) {                                             \
      const int q = min(q0 + lane, (c.groups[g].dim + 3) / 4 - 1);                                                                  \
      float4 v = *reinterpret_cast<const float4*>(feat + 4 * (q & 127));                                                            \
      v.x *= clipobs;                                                                                                               \
      *reinterpret_cast<float4*>(&at(outg, ob + (unsigned int)(4 * q))) = v;                                                        \
    }                                                                                                                               \
  } } while (0)
#define OBS_GROUPS_LATE(ROLE) do { } while (0)
#elif !defined(PBHC_ABL_OBS)
#define OBS_GROUPS(ROLE, LATE_TOO) do { if (use_runs) { OBS_GROUPS_RUNS(ROLE, LATE_TOO) } else { OBS_GROUPS_MAP(ROLE, LATE_TOO) } } while (0)
#define OBS_GROUPS_LATE(ROLE) do { if (use_runs) { OBS_GROUPS_LATE_RUNS(ROLE) } else { OBS_GROUPS_LATE_MAP(ROLE) } } while (0)
#else
#define OBS_GROUPS(ROLE, LATE_TOO) do { } while (0)
#define OBS_GROUPS_LATE(ROLE) do { } while (0)
#endif
  float rew_total = 0.0f, etr_val = 0.0f;

  if (!roleB) {
    // =============== role A, interval 2b: _compute_reward (legged_robot_base.py:715-761): lane i <-> term i ========================
    // (a) lane k < 10: e_k = exp(-err_k / sigma_k).  (b) lane i <-> term i: cheap selects.
    if (valid && lane < PBHC_NUM_SIGMA) {
      const float e = red[R_ERR0 + lane];
#ifndef PBHC_ABL_F
      // exp(-err / sigma) (motion_tracking.py:1154-1290) as 2^(-err log2(e) / sigma) with the hardware reciprocal and 2^x (1 ulp each; the
      // library's expf + a correctly rounded division are ~35 instructions of the reward chain): relative error <= ~(2 + |x|) 6e-8, i.e.
      // < 2e-6 for the arguments the tracking terms see — inside the 3e-5 / 1e-4 the reward columns are held to
      red[R_EXP0 + lane] = __builtin_amdgcn_exp2f((-e * __builtin_amdgcn_rcpf(pf_sigma)) * 1.44269504088896341f);
#else
      red[R_EXP0 + lane] = e * pf_sigma;
#endif
    }
    WAVE_LDS_FENCE();
    STAMP(25);
    if (valid) {
      const float pen_scale = pf_pen_scale;
      float myrew = 0.0f;
      {
        // Most terms are one value of the reduction row: `term_src` (filled by pbhc_env_create, see term_source()) is its LDS slot.  The
        // terms that combine several values are evaluated STRAIGHT-LINE for the whole wave — operands fetched up front by broadcast LDS
        // reads (one round trip), every value computed by every lane, each lane keeping the one its term id names — instead of a switch per
        // lane, whose cases ran one after the other, each behind its own LDS reads (1.7 k cycles of the chain).  The specialised build drops
        // the terms its config does not have (TERM folds to a literal there).
#ifdef PBHC_ABL_F
#define TERM(name) false
#elif defined(PBHC_STATIC_CFG)
#define TERM(name) cfg_has_term(kStaticCfg, PBHC_R_##name)
#else
#define TERM(name) true
#endif
        const int id = pf_tid;
        const float* ex = red + R_EXP0;
        const float* ft = red + R_FOOT0;
        float raw = red[max(pf_tsrc, 0)];
        float cfl[PBHC_MAX_FEET], rcn[PBHC_MAX_FEET], lsc[PBHC_MAX_FEET], fat[PBHC_MAX_FEET];
        float fF[PBHC_MAX_FEET], fXY[PBHC_MAX_FEET], fZ[PBHC_MAX_FEET], fV[PBHC_MAX_FEET], fVxy[PBHC_MAX_FEET], fHd[PBHC_MAX_FEET], fOri[PBHC_MAX_FEET];
#pragma unroll
        for (int f = 0; f < PBHC_MAX_FEET; ++f) {
          const int g = min(f, NF - 1);
          cfl[f] = misc[M_CFILT0 + g]; rcn[f] = misc[M_RCONTACT0 + g]; lsc[f] = misc[M_LASTC0 + g]; fat[f] = misc[M_FAT0 + g];
          fF[f] = ft[4 * g]; fXY[f] = ft[4 * g + 1]; fZ[f] = ft[4 * g + 2]; fV[f] = ft[4 * g + 3]; fVxy[f] = ft[8 + g];
          fHd[f] = c.foot_ori_terms ? ft[10 + g] : 0.0f; fOri[f] = c.foot_ori_terms ? ft[12 + g] : 0.0f;
        }
        const float gx = misc[M_GX], gy = misc[M_GY], exl = ex[PBHC_S_LOWER_BODY_POS], exu = ex[PBHC_S_UPPER_BODY_POS];
        if (TERM(TELEOP_CONTACT_MASK) || TERM(TELEOP_CONTACT_MASK_V2)) {
          float e = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF) e += fabsf(cfl[f] - rcn[f]);
          const float q = e / (float)NF;
          raw = id == PBHC_R_TELEOP_CONTACT_MASK ? 1.0f - q : (id == PBHC_R_TELEOP_CONTACT_MASK_V2 ? 0.5f - q : raw);
        }
        if (TERM(TELEOP_BODY_POSITION_EXTEND)) {
          const float v = exl * c.body_pos_lower_weight + exu * c.body_pos_upper_weight;
          raw = id == PBHC_R_TELEOP_BODY_POSITION_EXTEND ? v : raw;
        }
        if (TERM(PENALTY_ORIENTATION)) raw = id == PBHC_R_PENALTY_ORIENTATION ? gx * gx + gy * gy : raw;
        if (TERM(FEET_AIR_TIME)) {                               // stateful (motion_tracking.py:1307-1319): the air times advance only where the term is configured
          float v = 0.0f, nf[PBHC_MAX_FEET];
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) {
            const bool cfilt = fZ[f] > 1.0f || lsc[f] != 0.0f;
            const float first = (fat[f] > 0.0f && cfilt) ? 1.0f : 0.0f;
            const float t = fat[f] + dt;
            if (f < NF) v += (t - c.desired_feet_air_time) * first;
            nf[f] = cfilt ? t * 0.0f : t;
          }
          const bool mine = id == PBHC_R_FEET_AIR_TIME && lane < c.num_terms;
          if (mine) {
#pragma unroll
            for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF) misc[M_FAT0 + f] = nf[f];
          }
          raw = id == PBHC_R_FEET_AIR_TIME ? v : raw;
        }
        if (TERM(PENALTY_FEET_CONTACT_FORCES)) {
          float v = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF) v += fmaxf(fF[f] - c.max_contact_force, 0.0f);
          raw = id == PBHC_R_PENALTY_FEET_CONTACT_FORCES ? v : raw;
        }
        if (TERM(PENALTY_STUMBLE)) {
          float v = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF && fXY[f] > 5.0f * fabsf(fZ[f])) v = 1.0f;
          raw = id == PBHC_R_PENALTY_STUMBLE ? v : raw;
        }
        if (TERM(PENALTY_SLIPPAGE)) {
          float v = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF) v += fV[f] * (fF[f] > 1.0f ? 1.0f : 0.0f);
          raw = id == PBHC_R_PENALTY_SLIPPAGE ? v : raw;
        }
        if (TERM(FOOT_SLIP_PENALTY)) {
          float v = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f) if (f < NF) v += (fF[f] > 1.0f ? 1.0f : 0.0f) * fVxy[f];
          raw = id == PBHC_R_FOOT_SLIP_PENALTY ? v : raw;
        }
        if (TERM(ALIVE)) raw = id == PBHC_R_ALIVE ? 1.0f : raw;
        if (c.foot_ori_terms) {                                  // legged_robot_base.py:1030-1079: left + right, the *_contact forms weighted by contacts_filt
          float vh = 0.0f, vhc = 0.0f, vo = 0.0f, voc = 0.0f;
#pragma unroll
          for (int f = 0; f < PBHC_MAX_FEET; ++f)
            if (f < NF) { vh += fHd[f]; vhc += fHd[f] * cfl[f]; vo += fOri[f]; voc += fOri[f] * cfl[f]; }
          raw = id == PBHC_R_FEET_HEADING_ALIGNMENT ? vh : (id == PBHC_R_FEET_HEADING_ALIGNMENT_CONTACT ? vhc : (id == PBHC_R_PENALTY_FEET_ORI ? vo : (id == PBHC_R_PENALTY_FEET_ORI_CONTACT ? voc : raw)));
        }
#undef TERM
        if (pf_tsrc < 0 && !is_special_term(id)) raw = 0.0f;     // (an id with neither a slot nor a formula: 0, as before)
        myrew = lane < c.num_terms ? raw * pf_tscale : 0.0f;
        if (pf_tpen) myrew = myrew * pen_scale;
      }
      STAMP(26);
      // episode_sums (legged_robot_base.py:733-747): lane i owns COLUMN i — the row loaded in the prologue + the reward of the term that
      // accumulates into it (one gather) + the termination reward in its column — and stores the whole row in one coalesced access; a
      // terminated env stores zeros and hands the finished sums to extras["episode"] (reset_envs_idx :510-514) from the same registers
      {
        const float contrib = __shfl(myrew, max(pf_colterm, 0), PBHC_G);
        const float tr_ = c.has_termination ? (misc[M_RESET] != 0.0f && misc[M_TIMEOUT] == 0.0f ? 1.0f : 0.0f) * c.termination_scale : 0.0f;
        float newsum = sumrow + (pf_colterm >= 0 ? contrib : 0.0f);
        if (c.has_termination && lane == c.termination_sum_col) newsum = (c.use_vec_reward ? sumrow : newsum) + tr_;
        if (lane < c.num_sum_cols) {
          const bool rs = misc[M_RESET] != 0.0f;
          const u32 so = (u32)env * (u32)c.num_sum_cols + (u32)lane;
          at(io.episode_sums, so) = rs ? 0.0f : newsum;
          if (rs && io.episode_rew_out) at(io.episode_rew_out, so) = newsum / c.max_episode_length_s;
        }
      }
      STAMP(27);
      if (c.use_vec_reward) {
        if (lane < c.num_rew_cols) {
          float v = lane < c.num_terms ? myrew : 0.0f;
          if (c.only_positive_rewards) v = fmaxf(v, 0.0f);
          if (c.has_termination && lane == c.num_terms - 1) {      // column of the last loop term, sic (:743-744)
            float tr = (misc[M_RESET] != 0.0f && misc[M_TIMEOUT] == 0.0f ? 1.0f : 0.0f) * c.termination_scale;
            v += tr;
          }
          NTST(at(io.rew_buf, (u32)env * (u32)c.num_rew_cols + (u32)lane), v);
          rew_total = v;
        }
        rew_total = group_sum(rew_total);
      } else {
        float v = group_sum(lane < c.num_terms ? myrew : 0.0f);
        if (c.only_positive_rewards) v = fmaxf(v, 0.0f);
        if (lane == 0) {
          if (c.has_termination) {
            float tr = (misc[M_RESET] != 0.0f && misc[M_TIMEOUT] == 0.0f ? 1.0f : 0.0f) * c.termination_scale;
            v += tr;
          }
          io.rew_buf[env] = v;
        }
        rew_total = v;
      }
    }
    // the episode_sums stores above and the reset path's loads / stores of the same row below are issued by the same lanes' wave in
    // program order; the reset path waits for them explicitly
    WAVE_LDS_FENCE();
    STAMP(6);
    // ---------------- phase G: reset_envs_idx for terminated envs (legged_robot_base.py:491-517,
    // 599-686; motion_tracking.py:265-287,369-378,445-543) ------------------------------------------
    const bool do_reset = valid && misc[M_RESET] != 0.0f;
    // io.redraw_all (domain_rand.reinit_epis_rand fired, legged_robot_base.py:390-395): the episodic DR block of the reset path for EVERY env,
    // a surviving env keeping everything else — after this step's torques, before its observations, as `_update_tasks_callback` sits
    const bool do_dr = do_reset || (valid && io.redraw_all != 0);
    if (valid && lane == 0) { misc[M_LASTEP] = misc[M_EPLEN]; misc[M_DELAY] = (float)adelay; }
    if (do_dr) {
      // env origin + clip meta: role B's prologue loads, handed over in LDS before bar1
      const f3 origin = mk3(misc[M_ORIGIN0], misc[M_ORIGIN1], misc[M_ORIGIN2]);
      const float m_len = misc[M_CLIP_LEN], m_dt = misc[M_CLIP_DT];
      const int m_nf = __float_as_int(misc[M_CLIP_NF]), m_row0 = __float_as_int(misc[M_CLIP_ROW0]);
      // The start phase and the control delay of the new episode first (one Philox call, computed by every lane: no hand-over), so that the two
      // table rows of the second lookup — (0+1)*dt + new start: dof + root only (kick_motion_res after the cache was invalidated,
      // motion_tracking.py:378,536-543,477-507) — are REQUESTED before the per-dof draws, the stores and the episode book-keeping below and
      // consumed after them: the memory round trip of the lookup runs under ~2.5 k cycles of work instead of behind it.
      float ue[4];
      pbhc::rng_uniform4(rt.seed, env, step_ctr, 6, 0, ue);
      const float mlen = m_len;                              // = tbl.motion_len[mid]
      const float ns = io.ovr_start_time ? io.ovr_start_time[env] : ue[0] * mlen;   // sample_time motion_lib_base.py:486-495
      const float t2 = (0.0f + 1.0f) * dt + ns;
      float lk_b = 0.0f, lq0 = 0.0f, lq1 = 0.0f, lv0 = 0.0f, lv1 = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lr0 = 0.0f, lr1 = 0.0f;
      if (!MODE && do_reset) {
        int f0, f1;
        frame_blend(t2, m_len, m_nf, m_dt, &f0, &f1, &lk_b);
        const float* r0 = tbl.frames + (size_t)(m_row0 + f0) * tbl.row;
        const float* r1 = tbl.frames + (size_t)(m_row0 + f1) * tbl.row;
        lq0 = r0[dc]; lq1 = r1[dc]; lv0 = r0[D + dc]; lv1 = r1[D + dc];
        const int cl = 2 * D + min(lane, 1);
        lc0 = r0[cl]; lc1 = r1[cl];
        // the root body's 13 values, one per lane: pos (lanes 0-2), rot (3-6), vel (7-9), ang vel (10-12)
        const int l13 = min(lane, 12);
        const int col = l13 < 3 ? o_pos + l13 : (l13 < 7 ? o_rot + l13 - 3 : (l13 < 10 ? o_vel + l13 - 7 : o_ang + l13 - 10));
        lr0 = r0[col]; lr1 = r1[col];
      }
      for (int dd = lane; dd < D; dd += PBHC_G) {
        if (do_reset) { act[dd] = 0.0f; actd[dd] = 0.0f; }
        float ur[4] = {0.0f, 0.0f, 0.0f, 0.0f};          // the four episodic draws of this dof (kp, kd, rfi limit, rao) from one Philox call
        if (!dr_on_b) pbhc::rng_uniform4(rt.seed, env, step_ctr, 2, dd, ur);
        if (dr_on_b) {
        } else {
        if (c.randomize_pd_gain) {
          kpA = io.ovr_kp ? at(io.ovr_kp, eD + dd) : (c.kp_range[1] - c.kp_range[0]) * ur[0] + c.kp_range[0];
          kdA = io.ovr_kd ? at(io.ovr_kd, eD + dd) : (c.kd_range[1] - c.kd_range[0]) * ur[1] + c.kd_range[0];
          at(io.kp_scale, eD + dd) = kpA;
          at(io.kd_scale, eD + dd) = kdA;
        }
        if (c.randomize_rfi_lim)
          at(io.rfi_lim_scale, eD + dd) = io.ovr_rfi_lim ? at(io.ovr_rfi_lim, eD + dd) : (c.rfi_lim_range[1] - c.rfi_lim_range[0]) * ur[2] + c.rfi_lim_range[0];
        if (c.use_rao)
          at(io.rao_scale, eD + dd) = io.ovr_rao ? at(io.ovr_rao, eD + dd) : (c.rao_lim - (-c.rao_lim)) * ur[3] + (-c.rao_lim);
        if (c.randomize_ctrl_delay)
          for (int k = 0; k < Q; ++k) at(io.action_queue, ((u32)env * (u32)Q + (u32)k) * (u32)D + (u32)dd) = 0.0f;     // queue *= 0 (finite values)
        }
        if (c.randomize_default_dof_pos && io.default_dof_pos) {    // legged_robot_base.py:632-635 (its own Philox stream: off in the shipped yamls)
          float ub[4];
          pbhc::rng_uniform4(rt.seed, env, step_ctr, 9, dd, ub);
          const float bias = io.ovr_dof_pos_bias ? at(io.ovr_dof_pos_bias, eD + dd) : (c.dof_pos_range[1] - c.dof_pos_range[0]) * ub[0] + c.dof_pos_range[0];
          dpA = bias + c.default_dof_pos[dd];
          at(io.default_dof_pos, eD + dd) = dpA;
        }
      }
      if (lane == 0) {
        if (do_reset) {
          misc[M_FAT0] = 0.0f; misc[M_FAT1] = 0.0f;
          misc[M_CONTACT0] = 0.0f; misc[M_CONTACT1] = 0.0f; misc[M_CFILT0] = 0.0f; misc[M_CFILT1] = 0.0f;
          float old_start = misc[M_START];
          float etr = (misc[M_LASTEP] * dt + old_start) / misc[M_MLEN];
          io.end_time_ratio_buf[env] = etr;
          etr_val = etr;
          io.motion_len[env] = mlen;
          io.motion_start_times[env] = ns;
          misc[M_NEWSTART] = ns;
        } else {
          etr_val = etr_old;
        }
        if (c.randomize_ctrl_delay) {
          long long nd = io.ovr_delay ? io.ovr_delay[env]
                                      : (long long)c.ctrl_delay_range[0] + (long long)(ue[1] * (float)(c.ctrl_delay_range[1] + 1 - c.ctrl_delay_range[0]));
          io.action_delay_idx[env] = nd;
          misc[M_DELAY] = (float)nd;
        }
        if (do_reset) misc[M_EPLEN] = 0.0f;
      }
      WAVE_LDS_FENCE();
      if (!do_reset) {
      } else if (MODE) {
        // general tracking: _reset_dofs looks up at ep_len*dt + start = start (general_tracking.py:463-476), _reset_root_states at
        // (ep_len+1)*dt + start (kick_motion_res :398,486-496): dofs from the first lookup, root from the second
        motion_lookup_meta(tbl, D, Bx, lane, m_len, m_nf, m_dt, m_row0, 0.0f * dt + misc[M_NEWSTART], origin, false, q, qd, misc + M_RCONTACT0, rp, rq, rv, rw);
        motion_lookup_meta(tbl, D, Bx, lane, m_len, m_nf, m_dt, m_row0, t2, origin, false, rdof, rdofv, misc + M_RCONTACT0, rp, rq, rv, rw);
      } else {
        // motion_lookup_meta(..., bodies = false, q, qd, ...) on the rows requested above (the same lerp / slerp arithmetic, lane <-> value)
        const float la = 1.0f - lk_b;
        if (lane < D) { q[lane] = la * lq0 + lk_b * lq1; qd[lane] = la * lv0 + lk_b * lv1; }
        if (lane < 2) misc[M_RCONTACT0 + lane] = la * lc0 + lk_b * lc1;
        const float lin = la * lr0 + lk_b * lr1;
        if (lane < 3) rp[lane] = lin + (lane == 0 ? origin.x : (lane == 1 ? origin.y : origin.z));
        else if (lane >= 7 && lane < 10) rv[lane - 7] = lin;
        else if (lane >= 10 && lane < 13) rw[lane - 10] = lin;
        const f4 q0r = mk4(__shfl(lr0, 3, PBHC_G), __shfl(lr0, 4, PBHC_G), __shfl(lr0, 5, PBHC_G), __shfl(lr0, 6, PBHC_G));
        const f4 q1r = mk4(__shfl(lr1, 3, PBHC_G), __shfl(lr1, 4, PBHC_G), __shfl(lr1, 5, PBHC_G), __shfl(lr1, 6, PBHC_G));
        if (lane == 0) st4(rq, slerp(q0r, q1r, lk_b));
      }
      WAVE_LDS_FENCE();
      if (do_reset && lane == 0) {
        st3(root, ld3(rp));
        st4(root + 3, quat_mul(mk4(0.f, 0.f, 0.f, 1.f), ld4(rq)));       // quat_mul(small_random_quaternions(max_angle=0), root_rot)
        st3(root + 7, ld3(rv));
        st3(root + 10, ld3(rw));
      }
    } else if (valid && lane == 0) {
      etr_val = etr_old;
    }
    WAVE_LDS_FENCE();
    STAMP(7);
    // ---------------- phase H: post-reset features (role B wrote the no-reset values before bar2) ------------------------------------
    if (do_dr) {
      const int o_q = c.feat_off[PBHC_F_DOF_POS], o_qd = c.feat_off[PBHC_F_DOF_VEL], o_a = c.feat_off[PBHC_F_ACTIONS];
      const int o_kp = c.feat_off[PBHC_F_DR_KP], o_kd = c.feat_off[PBHC_F_DR_KD];
      for (int dd = lane; dd < D; dd += PBHC_G) {
        feat[o_q + dd] = q[dd] - dpA;                            // (a reset replaced it in registers, like kpA)
        feat[o_qd + dd] = qd[dd];
        feat[o_a + dd] = act[dd];
        if (!dr_on_b) {
          feat[o_kp + dd] = kpA;                                 // D <= 32: lane dd owns dof dd in the prologue load and in the reset path alike
          feat[o_kd + dd] = kdA;
        }
      }
      if (lane == 0) {
        feat[c.feat_off[PBHC_F_DR_CTRL_DELAY]] = misc[M_DELAY];
        feat[c.feat_off[PBHC_F_BASE_POS_Z]] = root[2];              // a live view of the root state in the reference: reset envs show the reset height
      }
      if (MODE && lane < NF) feat[c.feat_off[PBHC_F_CONTACT_MASK] + lane] = misc[M_CFILT0 + lane];
    }
    WAVE_LDS_FENCE();
    STAMP(8);
#ifndef PBHC_ABL_WB
    if (hist_b && valid) { STATE_WRITEBACK(); }                  // phase J on THIS role (see interval 3)
#endif
#if defined(PBHC_STATIC_CFG) && !defined(PBHC_ABL_OBS) && !defined(PBHC_ABL_OBSX4)
    // ... and its share of the reference waves' rows: every source is final for this role (its own phase H included), history excluded
    if (row_help && valid && obs_by_role) { OBS_GROUPS_HELP_RUNS() }
#endif
    // ---------------- observation rows of the groups assigned to this role (helpers.py:128-152, legged_robot_base.py:787-793,326-331,
    // history_handler.py:40-44): every source is final for THIS role now (its own phase H included)
    if (valid && obs_by_role) {
      OBS_GROUPS(0, true);
    }
  } else {
    // =============== role B, interval 2b: state outputs, future targets, the observation rows assigned to this role ================
    // ---- general tracking: future reference targets (general_tracking.py:500-565) ----------------------------------------------
    // S lookups at motion_times + steps[s]*dt with motion_times = ep_len*dt + start (ep_len already incremented).
    // Pass 1, lane s <-> step s: frame pair, root-frame quantities, anchor pose -> features + per-step scratch.
    // Pass 2/3, lane <-> (step, dof) / (step, key body): lerps from the two frame rows.  Wave-local (no barrier between the passes).
    if (MODE && c.future_num_steps > 0) {
      LOAD_CLIP_META();
      const int NS = c.future_num_steps, Kn = c.num_key, an = c.anchor_index;
      float* fut = S + lo.fut;                                    // per step: f0 f1 blend | anchor quat (4) | anchor pos (3)
      const int row0 = m_row0, nf_c = m_nf;
      const float len_c = m_len, dt_c = m_dt;
      const float tb = (float)ep1 * dt + start;
      if (valid)
        for (int st = lane; st < NS; st += PBHC_G) {
          const float t = (float)c.future_steps[st] * dt + tb;
          int f0, f1; float bl;
          frame_blend(t, len_c, nf_c, dt_c, &f0, &f1, &bl);
          const float* q0 = tbl.frames + (size_t)(row0 + f0) * tbl.row;
          const float* q1 = tbl.frames + (size_t)(row0 + f1) * tbl.row;
          const float al = 1.0f - bl;
          const f4 rr = slerp(ld4(q0 + o_rot), ld4(q1 + o_rot), bl);
          const f3 v0 = ld3(q0 + o_vel), v1 = ld3(q1 + o_vel), w0 = ld3(q0 + o_ang), w1 = ld3(q1 + o_ang);
          const f3 e = euler_xyz(rr);
          feat[c.feat_off[PBHC_F_FUT_ROOT_HEIGHT] + st] = al * q0[o_pos + 2] + bl * q1[o_pos + 2] + origin.z;
          feat[c.feat_off[PBHC_F_FUT_ROLL_PITCH] + 2 * st] = e.x;
          feat[c.feat_off[PBHC_F_FUT_ROLL_PITCH] + 2 * st + 1] = e.y;
          st3(feat + c.feat_off[PBHC_F_FUT_BASE_LIN_VEL] + 3 * st, quat_rotate_inverse(rr, mk3(al * v0.x + bl * v1.x, al * v0.y + bl * v1.y, al * v0.z + bl * v1.z)));
          st3(feat + c.feat_off[PBHC_F_FUT_BASE_ANG_VEL] + 3 * st, quat_rotate_inverse(rr, mk3(al * w0.x + bl * w1.x, al * w0.y + bl * w1.y, al * w0.z + bl * w1.z)));
          const f4 aq = an == 0 ? rr : slerp(ld4(q0 + o_rot + 4 * an), ld4(q1 + o_rot + 4 * an), bl);
          const f3 p0 = ld3(q0 + o_pos + 3 * an), p1 = ld3(q1 + o_pos + 3 * an);
          float* fs = fut + 10 * st;
          fs[0] = __int_as_float(f0); fs[1] = __int_as_float(f1); fs[2] = bl;
          st4(fs + 3, quat_conj(aq));
          st3(fs + 7, mk3(al * p0.x + bl * p1.x + origin.x, al * p0.y + bl * p1.y + origin.y, al * p0.z + bl * p1.z + origin.z));
        }
      WAVE_LDS_FENCE();
      if (valid) {
        const int o_fd = c.feat_off[PBHC_F_FUT_DOF_POS], o_fk = c.feat_off[PBHC_F_FUT_LOCAL_KEY_POS];
        int st = 0, dd = lane;                                       // (step, dof) without divisions: D may be < 32
        while (dd >= D) { dd -= D; ++st; }
        for (; st < NS;) {
          const float* fs = fut + 10 * st;
          const float* q0 = tbl.frames + (size_t)(row0 + __float_as_int(fs[0])) * tbl.row;
          const float* q1 = tbl.frames + (size_t)(row0 + __float_as_int(fs[1])) * tbl.row;
          feat[o_fd + st * D + dd] = (1.0f - fs[2]) * q0[dd] + fs[2] * q1[dd];
          dd += PBHC_G;
          while (dd >= D) { dd -= D; ++st; }
        }
        for (int i = lane; i < NS * Kn; i += PBHC_G) {
          const int st2 = i / Kn, k = i - st2 * Kn, body = c.key[k];
          const float* fs = fut + 10 * st2;
          const float* q0 = tbl.frames + (size_t)(row0 + __float_as_int(fs[0])) * tbl.row;
          const float* q1 = tbl.frames + (size_t)(row0 + __float_as_int(fs[1])) * tbl.row;
          const float bl = fs[2], al = 1.0f - bl;
          const f3 p0 = ld3(q0 + o_pos + 3 * body), p1 = ld3(q1 + o_pos + 3 * body);
          const f3 pw = mk3(al * p0.x + bl * p1.x + origin.x, al * p0.y + bl * p1.y + origin.y, al * p0.z + bl * p1.z + origin.z);
          st3(feat + o_fk + 3 * i, quat_apply(ld4(fs + 3), sub3(pw, ld3(fs + 7))));
        }
      }
    }
    WAVE_LDS_FENCE();
    if (valid && obs_by_role) {
      // a surviving env: every pair; a terminated env: all but the pairs that read post-reset features (after bar3)
      const bool rs = misc[M_RESET] != 0.0f || gateB || io.redraw_all != 0;
      if (wide) {
        // (a wave holds two envs: the one-block form when neither resets — wave-uniform, ~98 % of the waves)
#ifdef PBHC_WIDE_ONEBUF       // (measurement aid)
        { OBS_GROUPS_WIDE(0, rs) }
#else
        if (__builtin_amdgcn_ballot_w64(rs) == 0) { OBS_GROUPS_WIDE_ALL() } else { OBS_GROUPS_WIDE(0, rs) }
#endif
      } else if (rs) { OBS_GROUPS(1, false); } else { OBS_GROUPS(1, true); }
    }
    STAMPB(5);
  }
  LDS_BARRIER();                                               // bar3: post-reset features are in LDS
  STAMP(9);
  // a terminated env (~1 % of them): the pairs of role B's rows that read post-reset features, by role B itself — it is idle from here to bar4
  // while role A, the chain that sets the kernel's duration, writes the state back (round 2: role A wrote them before bar3, +2.8 k cycles on
  // exactly the workgroups that finish last)
  if (roleB && valid && obs_by_role && (misc[M_RESET] != 0.0f || io.redraw_all != 0)) {
    if (wide) { OBS_GROUPS_WIDE(1, true) } else { OBS_GROUPS_LATE(1); }
  }

  // =============== interval 3: state write-back (role A) ================================================================================
  if (valid) {
    if (obs_by_role) {
    } else {
      // per-element maps in global memory (a feature row too large for the compact LDS maps): both roles, 64 lanes per env
      const int l64 = lane + (roleB ? PBHC_G : 0);
      for (int g = 0; g < c.num_groups; ++g) {
        const int dim = c.groups[g].dim, clip = c.groups[g].clip;
        const int* __restrict__ mdst = rt.groups[g].dst;
        const int* __restrict__ msrc = rt.groups[g].src;
        const float* __restrict__ mscale = rt.groups[g].scale;
        const float* __restrict__ mnoise = rt.groups[g].noise;
        float* __restrict__ out = io.obs[g] + (size_t)env * (io.obs_pitch[g] ? io.obs_pitch[g] : c.groups[g].pitch);
        for (int j0 = l64; j0 < dim; j0 += 8 * 2 * PBHC_G) {
          int si[8], di[8]; float sc[8], ns[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 2 * PBHC_G;
            const bool ok = j < dim;
            si[u] = ok ? msrc[j] : 0; sc[u] = ok ? mscale[j] : 0.0f; ns[u] = ok ? mnoise[j] : 0.0f;
            di[u] = ok ? (mdst ? mdst[j] : j) : 0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * 2 * PBHC_G;
            if (j < dim) {
              float x = feat[si[u]];
              if (ns[u] != 0.0f) x = x + (obs_noise_u(nzb, 16 + g, (uint32_t)(mdst ? di[u] : j)) * 2.0f - 1.0f) * (ns[u] * noise_cur);
              x = x * sc[u];
              if (clip) x = clampf(x, -c.clip_observations, c.clip_observations);
              out[di[u]] = x;
            }
          }
        }
      }
    }
#ifdef PBHC_ABL_WB
    if (false) {
#else
    if (roleB && !hist_b) {
#endif
      // ---------------- phase J: state write-back (_post_compute_observations_callback :398-405), by the reference waves: everything it
      // stores is final in LDS since bar3 (a reset's new state included), these waves have nothing else left, and the arrays are the ones
      // they loaded in their prologue — while the dynamics waves go straight on to the partial sums (1.3 k cycles off the chain).
      // (hist_b builds: the reference waves write every observation row and are the longer role after bar2 — there the dynamics waves
      // do this right after their phase H, while they would otherwise wait at bar3.)
      STATE_WRITEBACK();
    }
  }

  STAMP(10);
  STAMP(11);
  // ---------------- partial sums for the host-side scalars of the reference: one row per dynamics WAVE (its two envs added up by one
  // cross-half exchange), lane <-> column, sources from the table loaded in the prologue.  No workgroup barrier, no LDS staging: the
  // reference waves are done, and k_env_finalize adds up 2 rows per workgroup in its fixed order.
  if (!roleB) {
    if (valid && lane == 0) {
      misc[M_PJOINT] = sqrtf(red[R_JP2]);
      misc[M_PEPLEN] = misc[M_RESET] != 0.0f ? misc[M_LASTEP] : 0.0f;
      misc[M_PETR] = etr_val; misc[M_PETRSQ] = etr_val * etr_val;
      misc[M_PREW] = rew_total;
      misc[M_ZERO] = 0.0f;
    }
    WAVE_LDS_FENCE();
    float v0 = 0.0f, v1 = 0.0f;
    if (valid) {
      v0 = S[(psrc0 & 0x8000u) ? Lds::MISC + (int)(psrc0 & 0x7FFFu) : lo.red + (int)psrc0];
      v1 = S[(psrc1 & 0x8000u) ? Lds::MISC + (int)(psrc1 & 0x7FFFu) : lo.red + (int)psrc1];
    }
    v0 += __shfl_xor(v0, PBHC_G, 2 * PBHC_G);                 // env 0 + env 1 of this wave
    v1 += __shfl_xor(v1, PBHC_G, 2 * PBHC_G);
    if ((threadIdx.x & PBHC_G) == 0) {
      float* prow = partials + (wgb * 2u + (u32)wave) * (u32)PBHC_NP;
      prow[lane] = v0;
      prow[lane + PBHC_G] = v1;
    }
  }
  STAMP(12);
  WG_STAMP(1);
}
