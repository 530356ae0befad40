// fp32 GEMMs of the PPO update's Linear / activation stacks on the gfx950 matrix cores, with the work the library GEMMs leave to extra
// passes folded into the epilogue (reference: the nn.Sequential of Linear + ELU built by agents/modules/modules.py:47-63 and differentiated
// by autograd in agents/ppo.py:391-410):
//   forward   y = act(x W^T + b)                       — bias + ELU/ReLU applied to the accumulators (no separate in-place ELU pass)
//   backward  dx = (dy W) * act'(saved), colsum(dx)    — the activation derivative of the layer below and the row-block column sums of
//                                                         its bias gradient, taken from the accumulators (no `pbhc_act_bwd_bias` pass)
// `v_mfma_f32_32x32x2_f32`: f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain (64 FLOP/clk/SIMD = 157 TFLOP/s on 256 CUs).
// It issues once per 64 cycles, so feeding it is cheap — one ds_read_b128 per operand tile per four MFMAs — and what matters is that
// nothing stalls the issue: double-buffered LDS stages with ONE barrier per stage, next stage's global loads in flight during the
// MFMAs, two workgroups per CU so that one's barrier / epilogue hides under the other's MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pbhc_hip.h"

extern thread_local char g_pbhc_err[512];
#define GEMM_FAIL(code, ...) do { snprintf(g_pbhc_err, sizeof(g_pbhc_err), __VA_ARGS__); return (code); } while (0)
#define GEMM_ARG(cond) do { if (!(cond)) GEMM_FAIL(PBHC_EINVAL, "%s: argument check failed: %s", __func__, #cond); } while (0)
#define GEMM_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) GEMM_FAIL(PBHC_EHIP, "%s: %s: %s", __func__, #expr, hipGetErrorString(e_)); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GEMM_T 256

// exp as 2^(x log2 e) on the hardware's v_exp_f32 (1 ulp; the product adds |x| 6e-8 relative): 2 instructions instead of the library expf's ~12.
// The epilogue's VALU work displaces MFMA issue of the co-resident workgroups (profiles/round4_fused_gemm_ablation.txt: "epilogue without global
// stores"), and the exp of ELU / SiLU was most of it: update 28.5-28.6 -> 28.0-28.1 ms, critic pass of the rollout 3.56 -> 3.46 ms, A/B inside one
// call (tools/probes/fast_exp_ab.sh).  The activations differ from ATen's by <= 1e-7 absolute (-DPBHC_GEMM_LIBM_EXP: the library expf).
#ifndef PBHC_GEMM_LIBM_EXP
#define GEMM_EXP(x) __builtin_amdgcn_exp2f((x) * 1.44269504088896341f)
#define GEMM_RCP(x) __builtin_amdgcn_rcpf(x)                    // (SiLU's 1 / (1 + e^-x): v_rcp_f32, 1 ulp, for an IEEE division's ~10 instructions)
#else
#define GEMM_EXP(x) expf(x)
#define GEMM_RCP(x) (1.0f / (x))
#endif
__device__ __forceinline__ float gemm_act(int act, float v) {
  if (act == 1) return v > 0.0f ? v : GEMM_EXP(v) - 1.0f;   // ATen's GPU ELU: exp(x) - 1 in f32 (ActivationEluKernel.cu)
  if (act == 2) return v * GEMM_RCP(1.0f + GEMM_EXP(-v));         // SiLU as ATen computes it: x / (1 + exp(-x))
  if (act == 3) return v > 0.0f ? v : 0.0f;
  return v;
}
__device__ __forceinline__ float gemm_act_grad(int act, float s) {      // ELU (alpha 1), ReLU: from the activation OUTPUT; SiLU: from the PRE-activation
  if (act == 1) return s > 0.0f ? 1.0f : s + 1.0f;
  if (act == 2) { const float sg = GEMM_RCP(1.0f + GEMM_EXP(-s)); return sg * (1.0f + s * (1.0f - sg)); }
  if (act == 3) return s > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}

// The same on four values at a time, written for the instruction count (every VALU instruction of the epilogue displaces ~12 cycles of the
// co-resident workgroups' MFMA issue; SQ counters: 235 VALU per wave and tile in the 64 x 128 forward epilogue, 160 of them the ELU): the
// multiplies / adds as packed pairs (v_pk_mul_f32 / v_pk_add_f32), and
//   ELU(v)  = med3(v, e^v - 1, 0)     — e^v - 1 > v for every v != 0, so the median is v for v > 0 and e^v - 1 below (one v_med3_f32 for a compare + select)
//   ELU'(y) = min(y + 1, 1)           — y >= -1 is the activation's output
// bit for bit the values of the scalar forms above (a -0 input aside).
template <int ACT>
__device__ __forceinline__ f32x4 gemm_act4(f32x4 v) {
#ifndef PBHC_GEMM_LIBM_EXP
  if (ACT == 1 || ACT == 2) {
    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const float sgn = ACT == 1 ? 1.44269504088896341f : -1.44269504088896341f;
    const f32x2 tl = lo * sgn, th = hi * sgn;
    f32x2 el = {__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])}, eh = {__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])};
    if (ACT == 1) {
      el = el - 1.0f; eh = eh - 1.0f;
      return f32x4{__builtin_amdgcn_fmed3f(v[0], el[0], 0.0f), __builtin_amdgcn_fmed3f(v[1], el[1], 0.0f), __builtin_amdgcn_fmed3f(v[2], eh[0], 0.0f),
                   __builtin_amdgcn_fmed3f(v[3], eh[1], 0.0f)};
    }
    el = el + 1.0f; eh = eh + 1.0f;
    const f32x2 rl = {__builtin_amdgcn_rcpf(el[0]), __builtin_amdgcn_rcpf(el[1])}, rh = {__builtin_amdgcn_rcpf(eh[0]), __builtin_amdgcn_rcpf(eh[1])};
    const f32x2 ol = lo * rl, oh = hi * rh;
    return f32x4{ol[0], ol[1], oh[0], oh[1]};
  }
#endif
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = gemm_act(ACT, v[q]);
  return o;
}
// v * act'(s) for four values
template <int ACT>
__device__ __forceinline__ f32x4 gemm_mul_grad4(f32x4 v, f32x4 sv) {
  if (ACT == 1) {
    const f32x2 sl = {sv[0], sv[1]}, sh = {sv[2], sv[3]};
    const f32x2 gl = sl + 1.0f, gh = sh + 1.0f;
    const f32x2 vl = {v[0], v[1]}, vh = {v[2], v[3]};
    const f32x2 ml = {fminf(gl[0], 1.0f), fminf(gl[1], 1.0f)}, mh = {fminf(gh[0], 1.0f), fminf(gh[1], 1.0f)};
    const f32x2 ol = vl * ml, oh = vh * mh;
    return f32x4{ol[0], ol[1], oh[0], oh[1]};
  }
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = v[q] * gemm_act_grad(ACT, sv[q]);
  return o;
}

// four consecutive floats of a row starting at column c (row length `len`), zeros beyond the row's end.  Rows are only 4-byte aligned in
// general (630-wide critic observations, 23-wide action gradients): global_load_dwordx4 needs dword alignment only.
struct __attribute__((packed, aligned(4))) F4U { f32x4 v; };
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, bool rowok, int c, int len) {
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (rowok && c + 4 <= len) {
    t = reinterpret_cast<const F4U*>(p)->v;
  } else if (rowok && c < len) {                           // the chunk that straddles the row's end (once per row, last K stage only)
    t[0] = p[0];
    if (c + 1 < len) t[1] = p[1];
    if (c + 2 < len) t[2] = p[2];
  }
  return t;
}

// MODE 0 ("NT", forward):  C[M,N] = act(A[M,K] . B[N,K]^T + bias[N])
// MODE 1 ("NN", backward): C[M,N] = (A[M,K] . B[K,N]) * act'(S[M,N]);  part[blockRow][N] = column sums of C over the block's rows
// Workgroup = 4 waves as WM x WN, a wave owns TM x TN tiles of 32x32; BM = 32 WM TM, BN = 32 WN TN; BK floats of K per LDS stage.
// K order inside a block of 8: lane (r, kk) holds k = 4 kk + s at MFMA step s for BOTH operands, so an operand tile is one 16-byte LDS read
// per four steps (any k permutation common to A and B leaves the product unchanged; the fmaf order is fixed, hence deterministic).
template <int MODE, int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(GEMM_T, 2) void k_gemm(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                     const float* __restrict__ S, float* __restrict__ Cout, float* __restrict__ Pre, float* __restrict__ part,
                                                     int M, int N, int K, int act, int tiles_n, int dbg) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int AS = BK + 4;                              // A row stride in LDS (floats): 16-byte aligned rows, conflict-free b128 reads
  constexpr int BS = MODE == 0 ? BK + 4 : BN + 8;         // B: [BN][BK+4] (k-contiguous) or [BK][BN+8] (n-contiguous; 4 rows apart = 32 banks)
  constexpr int A_STAGE = BM * AS, B_STAGE = MODE == 0 ? BN * BS : BK * BS;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 (A_STAGE + B_STAGE) floats
  float* As = lds;
  float* Bs = lds + 2 * A_STAGE;

  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous run of row panels so that the
  // column tiles of one row panel (which share the A panel) meet in one L2
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r = lane & 31, kk = lane >> 5;

  // ---- global -> register staging of one K stage ----
  constexpr int A_TPR = BK / 4;                           // threads per A row (16 bytes each)
  constexpr int A_N = (BM * A_TPR + GEMM_T - 1) / GEMM_T;
  constexpr int B_TPR = (MODE == 0 ? BK : BN) / 4;        // threads per B row (a row = one n for MODE 0, one k for MODE 1)
  constexpr int B_ROWS = MODE == 0 ? BN : BK;
  constexpr int B_N = (B_ROWS * B_TPR + GEMM_T - 1) / GEMM_T;
  f32x4 ra[A_N], rb[B_N];

  auto load_stage = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int e = tid + i * GEMM_T, row = e / A_TPR, kq = (e - row * A_TPR) * 4;
      const bool ok = (BM * A_TPR) % GEMM_T == 0 || e < BM * A_TPR;
      const int gr = row0 + row, gk = k0 + kq;
      ra[i] = load4(A + (size_t)gr * K + gk, ok && gr < M, gk, K);
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int e = tid + i * GEMM_T, row = e / B_TPR, q = (e - row * B_TPR) * 4;
      const bool ok = (B_ROWS * B_TPR) % GEMM_T == 0 || e < B_ROWS * B_TPR;
      if (MODE == 0) {
        const int gn = col0 + row, gk = k0 + q;
        rb[i] = load4(B + (size_t)gn * K + gk, ok && gn < N, gk, K);
      } else {
        const int gk = k0 + row, gn = col0 + q;
        rb[i] = load4(B + (size_t)gk * N + gn, ok && gk < K, gn, N);
      }
    }
  };
  auto store_stage = [&](int buf) {
    float* as = As + buf * A_STAGE;
    float* bs = Bs + buf * B_STAGE;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int e = tid + i * GEMM_T, row = e / A_TPR, kq = (e - row * A_TPR) * 4;
      if ((BM * A_TPR) % GEMM_T == 0 || e < BM * A_TPR) *reinterpret_cast<f32x4*>(as + row * AS + kq) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int e = tid + i * GEMM_T, row = e / B_TPR, q = (e - row * B_TPR) * 4;
      if ((B_ROWS * B_TPR) % GEMM_T == 0 || e < B_ROWS * B_TPR) *reinterpret_cast<f32x4*>(bs + row * BS + q) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

  const int nk = (K + BK - 1) / BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int kb = 0; kb < nk; ++kb) {
    const int cur = (dbg & 1) ? 0 : kb & 1;
    if (kb + 1 < nk && !(dbg & 1)) load_stage((kb + 1) * BK);           // in flight during this stage's MFMAs
    const float* as = As + cur * A_STAGE + (wm * TM * 32 + r) * AS + kk * 4;
    const float* bs = MODE == 0 ? Bs + cur * B_STAGE + (wn * TN * 32 + r) * BS + kk * 4 : Bs + cur * B_STAGE + (kk * 4) * BS + wn * TN * 32 + r;
#pragma unroll
    for (int k8 = 0; k8 < BK / 8; ++k8) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * AS + k8 * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (MODE == 0) {
          b[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * BS + k8 * 8);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) b[j][s] = bs[(k8 * 8 + s) * BS + j * 32];
        }
      }
      if (!(dbg & 2))
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nk && !(dbg & 1)) store_stage(cur ^ 1);                // the other buffer was last read before the previous barrier
    if (!(dbg & 4)) __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 tile: element e of lane (r, kk) is row 8 (e / 4) + 4 kk + (e % 4), column r ----
  float cs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + (wn * TN + j) * 32 + r;
    const bool cok = col < N;
    const float bv = (MODE == 0 && bias && cok) ? bias[col] : 0.0f;
    cs[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rb0 = row0 + (wm * TM + i) * 32 + 4 * kk;
      if (MODE == 1 && act) {
        float sv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = rb0 + 8 * (e >> 2) + (e & 3);
          sv[e] = (cok && row < M) ? S[(size_t)row * N + col] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] *= gemm_act_grad(act, sv[e]);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rb0 + 8 * (e >> 2) + (e & 3);
        float v = acc[i][j][e];
        if (MODE == 0) {
          v += bv;
          if (Pre && cok && row < M) Pre[(size_t)row * N + col] = v;
          v = gemm_act(act, v);
        }
        if (cok && row < M) Cout[(size_t)row * N + col] = v;
        if (MODE == 1) cs[j] += v;                        // rows >= M hold zeros (their A rows were loaded as zeros)
      }
    }
  }
  if (MODE == 1 && part) {
    // column sums over the block's BM rows in a fixed order: lane halves (kk), then the WM waves of a column through LDS
    float* red = lds;                                     // every wave is past the last barrier of the K loop: the stages are free
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      cs[j] += __shfl_xor(cs[j], 32);
      if (kk == 0) red[wm * BN + (wn * TN + j) * 32 + r] = cs[j];
    }
    __syncthreads();
    for (int c = tid; c < BN; c += GEMM_T) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < WM; ++w) t += red[w * BN + c];
      if (col0 + c < N) part[(size_t)tile_m * N + col0 + c] = t;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Version 2 of the same two GEMMs: the K stages go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no ds_write
// pass), NS stages deep in a ring, ONE raw s_barrier per stage and a counted `s_waitcnt vmcnt` so that NS - 2 stages stay in flight across
// the barrier.  An LDS-DMA wave-instruction fills 1 KiB lane-linearly, so the bank swizzle is applied to the per-lane SOURCE address and
// again on the read (the same involution on both sides):
//   k-contiguous operands (A; B of MODE 0): [row][BK] with the 16-byte chunk index XORed by (row >> 2) & 3 (BK 16) / (row >> 1) & 7 (BK 32)
//   — conflict-free for ds_read_b128's four 16-lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32);
//   n-contiguous B of MODE 1: [k][128] with the chunk index rotated by 8 for k rows with bit 2 set (the two k halves of a wave's read
//   then sit 32 banks apart).
// Rows / columns beyond the matrix are clamped to the last valid one (their results are never stored); K beyond the end reads a
// 16-byte zero constant; the one chunk that straddles K (K % 4 != 0) is loaded from [K - 4, K) — inside the row — and rotated into place
// in LDS by its owner before the last stage is used.  Needs K >= 4 and, for MODE 1, N % 4 == 0 (version 1 handles the rest).
__device__ __attribute__((aligned(16))) float g_gemm_zeros[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)l, 16, 0, 0);
}
// the same with the address split as SGPR base + 32-bit VGPR byte offset, so that advancing the base per stage is scalar work; M0 (the
// LDS destination) is written in the statement that reads it and restored (hipcc reserves it).  hipcc does not count an asm load in its
// own s_waitcnt bookkeeping: the K loop retires them with explicit counted waits and ends on vmcnt(0).
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, const float* l) {
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)l;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ void lds_barrier_raw() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int V> struct IC { static constexpr int value = V; };

// The narrow output layer of a stack (23 actions / 20 value heads behind a 128-wide hidden layer) riding in the epilogue of that hidden
// layer's forward GEMM: out[M, no] = y[M, 128] . w[no, 128]^T + b[no] with y the activated tile rows (no <= 32; out == nullptr: nothing).
struct GemmOutLayer { const float* w; const float* b; float* out; int no; };

// The fp32 MFMA moves 34 registers per lane in its 64 cycles (32x32x2: 16 accumulators in, 16 out, one A, one B), so it runs at the VGPR
// file's pace and every other VALU instruction a co-resident wave issues displaces it: measured, a K stage with ~57 address / bookkeeping
// VALU instructions per 32 MFMAs ran the matrix pipe at 67 %.  Hence the shape of this loop: per-lane addresses are stage-invariant 32-bit
// offsets (the stage's advance lives in the SGPR base of `global_load_lds_dwordx4 v_off, s[base]`; LDS fragment addresses are constant
// VGPRs + immediates because the ring position is a template constant), loop control is scalar, and nothing else touches the VALU.
template <int MODE, int WM, int WN, int TM, int TN, int BK, int NS>
__global__ __launch_bounds__(GEMM_T, 2) void k_gemm2(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                      const float* __restrict__ S, float* __restrict__ Cout, float* __restrict__ Pre, float* __restrict__ part,
                                                      int M, int N, int K, int act, int tiles_n, int dbg, int lda, int ldc, long long a_batch, long long c_batch,
                                                      GemmOutLayer fo) {
  // MODE 0 only: row strides of A / of the outputs (floats; lda >= K, ldc >= N) and a batch over blockIdx.y with element strides a_batch / c_batch
  // — the L output positions of a Conv1d window GEMM in one launch (agents/agent_modules.py).  MODE 1 is launched with lda = K, ldc = N, one batch.
  // fo (32 x 128 forward tiles with N = 128 only): the stack's narrow OUTPUT layer applied to the tile's activated rows in the epilogue.
  A += (size_t)blockIdx.y * a_batch;
  Cout += (size_t)blockIdx.y * c_batch;
  if (Pre) Pre += (size_t)blockIdx.y * c_batch;
  const long long dbg_c0 = (dbg & 8) ? clock64() : 0, dbg_w0 = (dbg & 8) ? wall_clock64() : 0;
  if (dbg & 0xE0) {
    // diagnosis (tools/gemm_ablation.py): de-phase the co-resident workgroups of the first round — the second / third workgroup of a CU starts
    // 1 x / 2 x ((dbg >> 5) * 4 us) late — to see what the launch pays for every tile's epilogue falling into the same moment
    const int type = (blockIdx.x >> 8) % 3;
    if (blockIdx.x < 768 && type) {
      const long long t0 = wall_clock64(), d = (long long)type * ((dbg >> 5) & 7) * 400;      // wall clock: 100 MHz
      while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(64);
    }
  }
  static_assert(WM * WN == 4, "four waves");
  static_assert(BK == 16 || BK == 32, "BK");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static_assert(MODE == 0 || BN == 128, "MODE 1: 128 columns per tile");
  constexpr int CPR = BK / 4;                              // 16-byte chunks per k-contiguous row
  constexpr int RPP = 64 / CPR;                            // rows per 1 KiB piece
  constexpr int A_PIECES = BM / RPP;
  constexpr int B_PIECES = MODE == 0 ? BN / RPP : BK / 2;  // MODE 1: two k rows of 128 floats per piece
  constexpr int PIECES = A_PIECES + B_PIECES;
  constexpr int LPW = (PIECES + 3) / 4;                    // LDS-DMA instructions per wave per stage (pieces wrap: a duplicate writes the same bytes)
  constexpr int A_STAGE = BM * BK, STAGE = (BM + BN) * BK;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // NS * STAGE floats

  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: everything derived from it stays off the VALU
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r = lane & 31, kk = lane >> 5;
  const int nk = (K + BK - 1) / BK;
  const int ktail0 = (nk - 1) * BK;                        // first k of the last stage
  const bool tail = (K - ktail0) < BK;                     // the last stage is partial

  // ---- stage-invariant per-lane byte offsets of this wave's pieces (relative to A / B at the stage's first k) ----
  auto piece_geom = [&](int i, int& prow_or_kr, int& c, bool& kcontig, bool& isA) {
    const int p = (wave + 4 * i) % PIECES;
    isA = p < A_PIECES;
    kcontig = isA || MODE == 0;
    if (kcontig) {
      prow_or_kr = (isA ? p : p - A_PIECES) * RPP + lane / CPR;          // row inside the tile
      const int sw = BK == 16 ? (prow_or_kr >> 2) & 3 : (prow_or_kr >> 1) & 7;
      c = (lane % CPR) ^ sw;                                             // source chunk that lives in this lane's LDS slot
    } else {
      prow_or_kr = 2 * (p - A_PIECES) + (lane >> 5);                     // k row inside the stage
      c = ((lane & 31) - 8 * ((prow_or_kr >> 2) & 1)) & 31;
    }
  };
  unsigned voff[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    int q, c; bool kc, isA;
    piece_geom(i, q, c, kc, isA);
    if (kc) {
      const int grow = min((isA ? row0 : col0) + q, (isA ? M : N) - 1);
      voff[i] = ((unsigned)grow * (unsigned)(isA ? lda : K) + 4u * c) * 4u;
    } else {
      voff[i] = ((unsigned)q * (unsigned)N + (unsigned)min(col0 + 4 * c, N - 4)) * 4u;
    }
  }
  auto issue = [&](int st, auto slot_c) {                  // stage st -> ring slot
    constexpr int SLOT = decltype(slot_c)::value;
    const int k0 = st * BK;
    if (tail && st == nk - 1) {                            // once per tile: per-lane pointers with the K edge handled
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const int p = (wave + 4 * i) % PIECES;
        int q, c; bool kc, isA;
        piece_geom(i, q, c, kc, isA);
        const float* g;
        if (kc) {
          const int grow = min((isA ? row0 : col0) + q, (isA ? M : N) - 1);
          const float* rowbase = (isA ? A : B) + (size_t)grow * (isA ? lda : K);
          const int gk = k0 + 4 * c;
          g = gk + 4 <= K ? rowbase + gk : gk < K ? rowbase + (K - 4) : g_gemm_zeros;
        } else {
          g = k0 + q < K ? B + (size_t)(k0 + q) * N + min(col0 + 4 * c, N - 4) : g_gemm_zeros;
        }
        glds16(g, &lds[SLOT * STAGE + p * 256]);
      }
    } else {
      const char* sA = reinterpret_cast<const char*>(A + k0);
      const char* sB = reinterpret_cast<const char*>(MODE == 0 ? B + k0 : B + (size_t)k0 * N);
#pragma unroll
      for (int i = 0; i < LPW; ++i) {
        const int p = (wave + 4 * i) % PIECES;
        const char* sbase = p < A_PIECES ? sA : sB;         // scalar select
        glds16_sv(sbase, voff[i], &lds[SLOT * STAGE + p * 256]);
      }
    }
  };

  // The accumulators start at the BIAS of their column (C/D map: lane (r, kk) holds column r of the tile in all 16 elements): the epilogue then has no
  // bias add — every VALU instruction outside the K loop displaces ~12 cycles of the co-resident workgroups' MFMA issue (SQ counters of the 768 x 630
  // forward: 543 non-MFMA VALU instructions per wave and tile against 84 % matrix-pipe utilisation) — and y = (b + sum_k x w), one rounding apart
  // from (sum_k x w) + b
  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int colj = col0 + (wn * TN + j) * 32 + r;
    const float bj = (MODE == 0 && bias && colj < N) ? bias[colj] : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = bj;
  }

  // constant per-lane fragment indices (floats) inside a stage; tile rows start at multiples of 32, so the row swizzle depends on r only
  const int sw_r = BK == 16 ? (r >> 2) & 3 : (r >> 1) & 7;
  int fa[BK / 8], fb[MODE == 0 ? BK / 8 : TN];
#pragma unroll
  for (int k8 = 0; k8 < BK / 8; ++k8) {
    fa[k8] = (wm * TM * 32 + r) * BK + 4 * ((2 * k8 + kk) ^ sw_r);
    if (MODE == 0) fb[k8] = A_STAGE + (wn * TN * 32 + r) * BK + 4 * ((2 * k8 + kk) ^ sw_r);
  }
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = A_STAGE + 4 * kk * BN + (((wn * TN + j) * 32 + r + 32 * kk) & 127);
  }

  auto stage_body = [&](auto slot_c, int kb) {
    constexpr int SLOT = decltype(slot_c)::value;
    // stage kb has landed for this wave once at most the younger stages' DMAs are outstanding ...
    if (kb + NS - 2 < nk) wait_vmcnt<(NS - 2) * LPW>(); else wait_vmcnt<0>();
    lds_barrier_raw();                                     // ... and for every wave; every wave is also done reading stage kb - 1
    if (kb + NS - 1 < nk && !(dbg & 1)) issue(kb + NS - 1, IC<(SLOT + NS - 1) % NS>{});
    if (tail && kb == nk - 1 && (K & 3)) {                 // rotate the straddling chunk [K-4, K) into [K - K%4, ...) and zero its end
      const int rem = K & 3, cst = (K - ktail0) >> 2;      // chunk index inside the stage row
      const int nrows = MODE == 0 ? BM + BN : BM;
      for (int t = tid; t < nrows; t += GEMM_T) {
        const int sw = BK == 16 ? (t >> 2) & 3 : (t >> 1) & 7;    // BM is a multiple of 32: B rows (t - BM) swizzle like t
        f32x4* q = reinterpret_cast<f32x4*>(&lds[SLOT * STAGE + t * BK + 4 * (cst ^ sw)]);
        const f32x4 o = *q;
        // (selects, not an if / else-if chain: hipcc 7.2 lowered the three-way chain to a flow whose rem == 3 arm stored unset registers)
        f32x4 n;
        n[0] = rem == 1 ? o[3] : rem == 2 ? o[2] : o[1];
        n[1] = rem == 1 ? 0.0f : rem == 2 ? o[3] : o[2];
        n[2] = rem == 3 ? o[3] : 0.0f;
        n[3] = 0.0f;
        *q = n;
      }
      lds_barrier_raw();
    }
    // fragments of k block k8 + 1 are read while the MFMAs of block k8 run (two register sets): one exposed LDS latency per stage, not per block
    f32x4 a[2][TM], b[2][TN];
    auto read_frags = [&](int k8, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(&lds[SLOT * STAGE + i * 32 * BK + fa[k8]]);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (MODE == 0) {
          bf[j] = *reinterpret_cast<const f32x4*>(&lds[SLOT * STAGE + j * 32 * BK + fb[k8]]);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[j][s] = lds[SLOT * STAGE + (k8 * 8 + s) * BN + fb[j]];
        }
      }
    };
    read_frags(0, a[0], b[0]);
#pragma unroll
    for (int k8 = 0; k8 < BK / 8; ++k8) {
      if (k8 + 1 < BK / 8) read_frags(k8 + 1, a[(k8 + 1) & 1], b[(k8 + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch ahead of this block's MFMAs (the scheduler sinks it otherwise)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k8 & 1][i][s], b[k8 & 1][j][s], acc[i][j], 0, 0, 0);
      // (s_setprio 1 / 3 around this block — favouring waves that are issuing MFMAs over co-resident waves in their prologue / epilogue —
      // measured in the update: 29.64 / 29.67 / 29.63 ms per update without / with priority 1 / 3: nothing)
    }
  };

  if (0 < nk) issue(0, IC<0>{});
  if (NS > 2 && 1 < nk) issue(1, IC<1 % NS>{});
  if (NS > 3 && 2 < nk) issue(2, IC<2 % NS>{});
  for (int kb = 0; kb < nk; kb += NS) {
    stage_body(IC<0>{}, kb);
    if (kb + 1 < nk) stage_body(IC<1 % NS>{}, kb + 1);
    if (NS > 2 && kb + 2 < nk) stage_body(IC<2 % NS>{}, kb + 2);
    if (NS > 3 && kb + 3 < nk) stage_body(IC<3 % NS>{}, kb + 3);
  }

  // ---- epilogue: through LDS, so that global traffic is 16 bytes per lane along rows (a 128-column tile row = 512 contiguous bytes per
  // half wave) instead of the accumulator layout's 4-byte pieces.  The accumulators go to a [BM][BN + 4] image as they are (C/D map: element e
  // of lane (r, kk) is row 8 (e / 4) + 4 kk + (e % 4), column r); bias / activation / activation derivative are applied on the way out.
  if ((dbg & 16) && acc[0][0][0] != 12345.678f) return;
  constexpr int CS = BN + 4;                               // row stride of the image (floats): 16-byte rows, ds_write_b32 by 32 consecutive columns
  constexpr bool CAN_FUSE_OUT = MODE == 0 && BM == 32 && BN == 128;
  const bool fuse_out = CAN_FUSE_OUT && fo.out != nullptr;
  f32x4 wo[3];                                             // the output layer's weights on their way to LDS: 32 x 128 floats = 3 x 16 B per thread (+ a rest)
  if (CAN_FUSE_OUT && fuse_out) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int ch = tid + GEMM_T * q;                     // 16-byte chunk: row n = ch / 32, floats 4 (ch % 32) ...
      const int n = ch >> 5;
      wo[q] = n < fo.no ? reinterpret_cast<const F4U*>(fo.w + (size_t)n * 128 + 4 * (ch & 31))->v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  lds_barrier_raw();                                       // every wave is out of the K loop (its last stage waited vmcnt(0)): the ring is free
  if (CAN_FUSE_OUT && fuse_out) {                          // W_out image [32][CS] behind the tile image (rows n >= no are zeros)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = tid + GEMM_T * q;
      const int n = ch >> 5;
      f32x4 v = q < 3 ? wo[q < 3 ? q : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (q == 3 && n < fo.no) v = reinterpret_cast<const F4U*>(fo.w + (size_t)n * 128 + 4 * (ch & 31))->v;
      *reinterpret_cast<f32x4*>(&lds[BM * CS + n * CS + 4 * (ch & 31)]) = v;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        lds[((wm * TM + i) * 32 + 8 * (e >> 2) + 4 * kk + (e & 3)) * CS + (wn * TN + j) * 32 + r] = acc[i][j][e];
  lds_barrier_raw();
  {
    constexpr int RP = GEMM_T / (BN / 4);                  // rows per pass: a thread owns 4 columns and every RP-th row
    constexpr int NP = BM / RP;
    const int c4 = (tid % (BN / 4)) * 4, rr = tid / (BN / 4);
    const int col = col0 + c4;
    const bool vec = (N & 3) == 0;                         // whole 16-byte pieces inside the row (dword-aligned 16-byte accesses)
    const bool cok = vec ? col < N : col < N;              // first column inside
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 sv[NP];
    if (MODE == 1 && act) {                                // the lower layer's saved activations for all of this thread's rows, in flight together
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int row = row0 + p * RP + rr;
        sv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < M && cok) {
          const float* sp = S + (size_t)row * N + col;
          if (vec) sv[p] = reinterpret_cast<const F4U*>(sp)->v;
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (col + q < N) sv[p][q] = sp[q];
          }
        }
      }
    }
    // The row pass with the activation and the store width as COMPILE-TIME constants (dispatched once below): with `act` / `vec` read per element
    // the loop was a chain of scalar compares and branches around every group of four values (86 s_cmp + 270 branches in the epilogue of the
    // 64 x 128 forward tile), which is latency the co-resident workgroups' MFMAs do not hide for this workgroup
    auto row_pass = [&](auto act_c, auto vec_c) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool VEC = (decltype(vec_c)::value & 1) != 0;
      constexpr bool INTERIOR = (decltype(vec_c)::value & 2) != 0;      // the whole tile lies inside the matrix: no row / column tests
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = row0 + p * RP + rr;
      f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(p * RP + rr) * CS + c4]);
      const bool ok = INTERIOR || (row < M && cok);
      if (MODE == 0) {
        if (Pre && ok) {
          float* pp = Pre + (size_t)row * ldc + col;
          if (VEC) reinterpret_cast<F4U*>(pp)->v = v;
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (col + q < N) pp[q] = v[q];
          }
        }
        v = gemm_act4<ACT>(v);
        if (CAN_FUSE_OUT && fuse_out) *reinterpret_cast<f32x4*>(&lds[(p * RP + rr) * CS + c4]) = v;   // the activated row stays for the output layer
      } else if (ACT) {
        v = gemm_mul_grad4<ACT>(v, sv[p]);
      }
      if (ok && !((dbg & 2) && v[0] != 12345.678f)) {      // (dbg 2: the epilogue without its global stores)
        float* cp = Cout + (size_t)row * ldc + col;
        if (VEC && (dbg & 4)) __builtin_nontemporal_store(v, &reinterpret_cast<F4U*>(cp)->v);      // (dbg 4: C as streaming stores — measurement)
        else if (VEC) reinterpret_cast<F4U*>(cp)->v = v;
        else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (col + q < N) cp[q] = v[q];
        }
        if (MODE == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) csum[q] += v[q];      // rows in the fixed order p = 0, 1, ...
        }
      }
    }
    };
    const bool interior = row0 + BM <= M && col0 + BN <= N;
    if (vec && interior) {
      if (act == 1) row_pass(IC<1>{}, IC<3>{}); else if (act == 2) row_pass(IC<2>{}, IC<3>{}); else if (act == 3) row_pass(IC<3>{}, IC<3>{}); else row_pass(IC<0>{}, IC<3>{});
    } else if (vec) {
      if (act == 1) row_pass(IC<1>{}, IC<1>{}); else if (act == 2) row_pass(IC<2>{}, IC<1>{}); else if (act == 3) row_pass(IC<3>{}, IC<1>{}); else row_pass(IC<0>{}, IC<1>{});
    } else {
      if (act == 1) row_pass(IC<1>{}, IC<0>{}); else if (act == 2) row_pass(IC<2>{}, IC<0>{}); else if (act == 3) row_pass(IC<3>{}, IC<0>{}); else row_pass(IC<0>{}, IC<0>{});
    }
    if ((dbg & 8) && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) {
      __builtin_amdgcn_s_waitcnt(0);
      Cout[0] = (float)(clock64() - dbg_c0);
      Cout[1] = (float)(wall_clock64() - dbg_w0);
    }
    if constexpr (CAN_FUSE_OUT) {
      if (fuse_out) {
        // out[32, 32] = y[32, 128] . W_out^T: wave w takes k in [32 w, 32 w + 32) (16 MFMAs, operand fragments as in the K loop: lane (r, kk) holds
        // k = 8 k8 + 4 kk + s), the four partial tiles meet in LDS — in the tile image's place, [4][32][33] floats are exactly its 32 x 132 —
        // and are summed in the fixed order w = 0..3 (+ bias).  Rows / columns beyond M / no are computed on zeros and not stored.
        lds_barrier_raw();                                 // activated image + W_out image complete
        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.0f;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
          const f32x4 ya = *reinterpret_cast<const f32x4*>(&lds[r * CS + 32 * wave + 8 * k8 + 4 * kk]);
          const f32x4 wb = *reinterpret_cast<const f32x4*>(&lds[BM * CS + r * CS + 32 * wave + 8 * k8 + 4 * kk]);
#pragma unroll
          for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[s], wb[s], o, 0, 0, 0);
        }
        lds_barrier_raw();                                 // every wave has read its fragments: the image may be overwritten
#pragma unroll
        for (int e = 0; e < 16; ++e) lds[wave * (32 * 33) + (8 * (e >> 2) + 4 * kk + (e & 3)) * 33 + r] = o[e];
        lds_barrier_raw();
        const int m = tid >> 3, q0 = tid & 7;
        if (row0 + m < M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = q0 + 8 * j;
            if (n < fo.no) {
              float t = lds[m * 33 + n];
#pragma unroll
              for (int w = 1; w < 4; ++w) t += lds[w * (32 * 33) + m * 33 + n];
              fo.out[(size_t)(row0 + m) * fo.no + n] = t + (fo.b ? fo.b[n] : 0.0f);
            }
          }
        }
      }
    }
    if (MODE == 1 && part) {
      // column sums over the tile's rows: the RP threads of a column group through LDS, fixed order
      lds_barrier_raw();                                   // the image has been read
      float* red = lds;
#pragma unroll
      for (int q = 0; q < 4; ++q) red[rr * BN + c4 + q] = csum[q];
      lds_barrier_raw();
      for (int c = tid; c < BN; c += GEMM_T) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < RP; ++w) t += red[w * BN + c];
        if (col0 + c < N) part[(size_t)tile_m * N + col0 + c] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a Linear: dW[n, k] = sum_m dy[m, n] x[m, k] (autograd's `dy.t() @ x`, agents/modules/modules.py:47-63 under torch.autograd).
// The reduction runs over the minibatch rows (24 576) and the output is small, so the rows are split over P workgroup groups that each write a
// partial [N, K] image, summed in a fixed order by k_wgrad_sum (what the library path does as a split-K batched GEMM + torch.sum).
// Both operands are m-major (a row of dy / x per sample), i.e. the MFMA's reduction index runs across LDS rows.  Instead of gathering each
// lane's four k values with 4-byte reads, the 32x32 tiles INTERLEAVE their rows: a lane reads 8 contiguous bytes of one m row — columns
// 2i, 2i+1 — and feeds column 2i to tile 0 and column 2i+1 to tile 1 (tile q owns the output rows / columns = q mod 2).  One ds_read_b64 per
// operand then serves four MFMAs of a k step (2 x 2 tiles), the reads of a 32-lane group are 256 contiguous bytes (conflict-free without a
// swizzle), and the interleave is undone when the accumulators are written to the epilogue's LDS image.
// Tile: 128 (n) x 128 (k) per workgroup = 2 x 2 waves of 64 x 64; stages of 32 rows by LDS-DMA (2 m rows per 1 KiB piece), 2-deep ring, two
// workgroups per CU; same VALU-free K loop as k_gemm2.  K % 4 != 0 (380- / 630-wide inputs): the chunk that straddles a row's end is fetched
// from [K-4, K) and its columns are rotated back when the partial image is written.
#define WG_BK 32
#define WG_BM 128
#define WG_BN 128
__global__ __launch_bounds__(GEMM_T, 2) void k_wgrad(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part,
                                                      int N, int K, int stages_total, int tiles_k, int tiles, int P) {
  constexpr int A_STAGE = WG_BK * WG_BM, STAGE = WG_BK * (WG_BM + WG_BN);     // floats
  constexpr int A_PIECES = A_STAGE / 256, PIECES = STAGE / 256, LPW = PIECES / 4;   // 16 + 16 pieces, 8 per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];                // 2 stages (64 KB); reused by the epilogue image [128][132]
  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs; the tiles of one row group read the same dy / x rows, so a
  // row group lives on ONE XCD (its rows are fetched into that L2 once and hit by the group's other tiles) — P is a multiple of 8 when >= 8
  int tile, p_idx;
  if ((P & 7) == 0) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; p_idx = xcd + 8 * (slot / tiles); tile = slot - (slot / tiles) * tiles; }
  else { p_idx = blockIdx.x / tiles; tile = blockIdx.x - p_idx * tiles; }
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int n0 = tile_n * WG_BM, k0c = tile_k * WG_BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, kk = lane >> 5;
  // this group's run of K stages (uneven split: any P fills the chip)
  const int st0 = (int)((long long)p_idx * stages_total / P), st1 = (int)((long long)(p_idx + 1) * stages_total / P);
  const int nk = st1 - st0;
  const float* dYp = dY + (size_t)st0 * WG_BK * N;
  const float* Xp = X + (size_t)st0 * WG_BK * K;

  unsigned voff[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int p = wave + 4 * i;                            // 32 pieces over 4 waves: no wrap; a piece = 2 m rows x 128 columns
    const bool isA = p < A_PIECES;
    const int row = (isA ? p : p - A_PIECES) * 2 + (lane >> 5);
    const int c = lane & 31;
    // beyond the matrix (and the chunk that straddles a row's end): the last whole chunk of the row — never stored, or rotated back below
    const int col = isA ? min(n0 + 4 * c, N - 4) : min(k0c + 4 * c, K - 4);
    voff[i] = ((unsigned)row * (unsigned)(isA ? N : K) + (unsigned)col) * 4u;
  }
  auto issue = [&](int st, auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    const char* sA = reinterpret_cast<const char*>(dYp + (size_t)st * WG_BK * N);
    const char* sB = reinterpret_cast<const char*>(Xp + (size_t)st * WG_BK * K);
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int p = wave + 4 * i;
      glds16_sv(p < A_PIECES ? sA : sB, voff[i], &lds[SLOT * STAGE + p * 256]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][p][e] = 0.0f;
  // constant per-lane fragment indices inside a stage (floats): m row kk of the k step, columns 2r, 2r+1 of this wave's 64
  const int fa = kk * WG_BM + wm * 64 + 2 * r;
  const int fb = A_STAGE + kk * WG_BN + wn * 64 + 2 * r;

  auto stage_body = [&](auto slot_c, int kb) {
    constexpr int SLOT = decltype(slot_c)::value;
    wait_vmcnt<0>();
    lds_barrier_raw();
    if (kb + 1 < nk) issue(kb + 1, IC<(SLOT + 1) % 2>{});
    // four k steps (8 m rows) of fragments per group, the next group read while this one's MFMAs run
    f32x2 a[2][4], b[2][4];
    auto read_frags = [&](int g, f32x2 (&af)[4], f32x2 (&bf)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *reinterpret_cast<const f32x2*>(&lds[SLOT * STAGE + (8 * g + 2 * t) * WG_BM + fa]);
        bf[t] = *reinterpret_cast<const f32x2*>(&lds[SLOT * STAGE + (8 * g + 2 * t) * WG_BN + fb]);
      }
    };
    read_frags(0, a[0], b[0]);
#pragma unroll
    for (int g = 0; g < WG_BK / 8; ++g) {
      if (g + 1 < WG_BK / 8) read_frags(g + 1, a[(g + 1) & 1], b[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) acc[q][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][t][q], b[g & 1][t][p], acc[q][p], 0, 0, 0);
    }
  };
  if (nk > 0) issue(0, IC<0>{});
  for (int kb = 0; kb < nk; kb += 2) {
    stage_body(IC<0>{}, kb);
    if (kb + 1 < nk) stage_body(IC<1>{}, kb + 1);
  }

  // ---- epilogue: accumulators -> [128][132] image (tile (q, p) element e of lane (r, kk): row 2 (8 (e/4) + 4 kk + e%4) + q, column 2 r + p) ->
  // partial[p_idx][n][k] with 16-byte stores
  constexpr int CS = WG_BN + 4;
  lds_barrier_raw();
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 64 + 2 * (8 * (e >> 2) + 4 * kk + (e & 3)) + q;
      *reinterpret_cast<f32x2*>(&lds[row * CS + wn * 64 + 2 * r]) = f32x2{acc[q][0][e], acc[q][1][e]};
    }
  lds_barrier_raw();
  float* out = part + (size_t)p_idx * N * K;
  const int c4 = (tid & 31) * 4, rr = tid >> 5;            // 8 rows per pass
  const int col = k0c + c4;
  const int rem = K & 3;
  const bool straddle = rem != 0 && col < K && col + 4 > K;
#pragma unroll
  for (int p = 0; p < WG_BM / 8; ++p) {
    const int row = n0 + p * 8 + rr;
    f32x4 v = *reinterpret_cast<const f32x4*>(&lds[(p * 8 + rr) * CS + c4]);
    if (row < N && col < K) {
      float* cp = out + (size_t)row * K + col;
      if (!straddle) {
        reinterpret_cast<F4U*>(cp)->v = v;
      } else {                                             // this chunk was computed on x[:, K-4 .. K): column col + q sits at slot q + 4 - rem
        cp[0] = rem == 1 ? v[3] : rem == 2 ? v[2] : v[1];
        if (rem >= 2) cp[1] = rem == 2 ? v[3] : v[2];
        if (rem == 3) cp[2] = v[3];
      }
    }
  }
}

// dW[i] = sum_p part[p][i], p in order (deterministic); 16 bytes per lane
__global__ __launch_bounds__(256) void k_wgrad_sum(const float* __restrict__ part, float* __restrict__ dW, int n, int P) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (; p + 8 <= P; p += 8) {
      f32x4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = reinterpret_cast<const F4U*>(part + (size_t)(p + u) * n + i)->v;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += q[u];
    }
    for (; p < P; ++p) s += reinterpret_cast<const F4U*>(part + (size_t)p * n + i)->v;
    reinterpret_cast<F4U*>(dW + i)->v = s;
  } else {
    for (int j = i; j < n; ++j) {
      float s = 0.0f;
      for (int p = 0; p < P; ++p) s += part[(size_t)p * n + j];
      dW[j] = s;
    }
  }
}

static int g_dbg = 0;
template <int MODE, int WM, int WN, int TM, int TN, int BK>
static hipError_t gemm_launch(const float* A, const float* B, const float* bias, const float* S, float* C, float* pre, float* part, int M, int N, int K, int act,
                              hipStream_t st) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int LDS_BYTES = 2 * (BM * (BK + 4) + (MODE == 0 ? BN * (BK + 4) : BK * (BN + 8))) * 4;
  static bool attr_set = LDS_BYTES <= 65536;               // more than 64 KB of LDS per workgroup has to be granted once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<MODE, WM, WN, TM, TN, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm<MODE, WM, WN, TM, TN, BK>), dim3(tm * tn), dim3(GEMM_T), LDS_BYTES, st, A, B, bias, S, C, pre, part, M, N, K, act, tn, g_dbg);
  return hipGetLastError();
}

// tile shapes: 0 = 128x128 (2x2 waves of 2x2 tiles), 1 = 96x128 (1x4 waves of 3x1 tiles: 24576 rows = 256 panels, one per CU when N = 128),
// 2 = 64x128 (1x4 waves of 2x1)
static int tile_bm(int shape) { return shape == 0 ? 128 : shape == 1 ? 96 : shape == 4 ? 32 : 64; }   // (shape 3, forward only: 64 x 64, 2x2 waves of one tile)

template <int MODE, int WM, int WN, int TM, int TN, int BK, int NS>
static hipError_t gemm2_launch(const float* A, const float* B, const float* bias, const float* S, float* C, float* pre, float* part, int M, int N, int K, int act,
                               hipStream_t st, int lda = 0, int ldc = 0, long long a_batch = 0, long long c_batch = 0, int batches = 1,
                               GemmOutLayer fo = GemmOutLayer{nullptr, nullptr, nullptr, 0}) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int RING_BYTES = NS * (BM + BN) * BK * 4, IMAGE_BYTES = (BM + (MODE == 0 && BM == 32 && BN == 128 ? 32 : 0)) * (BN + 4) * 4;   // K stages; the epilogue's [BM][BN + 4] image (+ the fused output layer's weights)
  constexpr int LDS_BYTES = RING_BYTES > IMAGE_BYTES ? RING_BYTES : IMAGE_BYTES;
  static bool attr_set = LDS_BYTES <= 65536;               // (the shapes pick_shape() chooses stay below 64 KB: nothing to set, safe under stream capture)
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2<MODE, WM, WN, TM, TN, BK, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  hipLaunchKernelGGL((k_gemm2<MODE, WM, WN, TM, TN, BK, NS>), dim3(tm * tn, batches), dim3(GEMM_T), LDS_BYTES, st, A, B, bias, S, C, pre, part, M, N, K, act, tn, g_dbg,
                     lda ? lda : K, ldc ? ldc : N, a_batch, c_batch, fo);
  return hipGetLastError();
}

#ifndef PBHC_GEMM_BK
#define PBHC_GEMM_BK 32
#endif
static int g_variant = 0;                                  // diagnosis: 0 automatic (version 2 where it applies); 1 force version 1 (register staging); 2 version 2 with BK 16 x 3 stages
template <int MODE, int WM, int WN, int TM, int TN>
static hipError_t gemm_variant(const float* A, const float* B, const float* bias, const float* S, float* C, float* pre, float* part, int M, int N, int K, int act,
                               hipStream_t st) {
  // version 2: 32-bit byte offsets into both operands, K >= 4, whole 16-byte chunks along n for MODE 1
  const bool v2ok = K >= 4 && (MODE == 0 || (N & 3) == 0) && (size_t)M * K < (1u << 30) && (size_t)(MODE == 0 ? N : K) * (MODE == 0 ? K : N) < (1u << 30);
  if (g_variant == 1 || !v2ok) return gemm_launch<MODE, WM, WN, TM, TN, PBHC_GEMM_BK>(A, B, bias, S, C, pre, part, M, N, K, act, st);
  if (g_variant == 2) return gemm2_launch<MODE, WM, WN, TM, TN, 16, 3>(A, B, bias, S, C, pre, part, M, N, K, act, st);
  return gemm2_launch<MODE, WM, WN, TM, TN, 32, 2>(A, B, bias, S, C, pre, part, M, N, K, act, st);
}
template <int MODE>
static hipError_t gemm_dispatch(int shape, const float* A, const float* B, const float* bias, const float* S, float* C, float* pre, float* part, int M, int N, int K,
                                int act, hipStream_t st) {
  if (shape == 0) return gemm_variant<MODE, 2, 2, 2, 2>(A, B, bias, S, C, pre, part, M, N, K, act, st);
  if (shape == 1) return gemm_variant<MODE, 1, 4, 3, 1>(A, B, bias, S, C, pre, part, M, N, K, act, st);
  if constexpr (MODE == 0) {
    if (shape == 3) return gemm_variant<MODE, 2, 2, 1, 1>(A, B, bias, S, C, pre, part, M, N, K, act, st);
  }
  if (shape == 4) return gemm_variant<MODE, 1, 4, 1, 1>(A, B, bias, S, C, pre, part, M, N, K, act, st);        // 32 x 128
  return gemm_variant<MODE, 1, 4, 2, 1>(A, B, bias, S, C, pre, part, M, N, K, act, st);
}

// Tile choice (measured on MI355X at the update's 24 576-row minibatch, tools/gemm_probe.py): 64 x 128 tiles, three workgroups per CU, are the
// fastest for every N >= 256 (1 536 / 1 024 / 768 / 384 ... tiles fill 768 slots in whole rounds); a 128-column layer has one column tile, and
// 96-row tiles then put exactly one tile on each of the 256 CUs (24 576 = 256 x 96).
static int pick_shape(int M, int N, int K, int forced, bool forward) {
  if (forced >= 0 && forced <= 4) return forced;
  if (forward && N <= 64) return 3;                        // one 64-column tile holds every output column (the encoders' 60- / 30-wide per-step Linear)
  // 32 x 128 tiles (one 32 x 32 tile per wave, three to four workgroups per CU) where the K loop is short — the 128-column layers' forward
  // (K 256 / 512) and the input gradients of layers with <= 128 outputs (K = out_features): more workgroups in flight hide a tile's prologue and
  // epilogue, which is most of such a launch.  Chosen from the update's own trace (profiles/round2_update_step_timeline.txt: forward 128x256
  // 28.5 -> 20.0 us, 128x512 36.3 -> 31.9; input gradient 21 -> 128 17.2 -> 10.5, 23 -> 128 15.7 -> 9.2, 128 -> 512 42.3 -> 39.8); the wider layers
  // lose 2-10 % on it there although a back-to-back probe (tools/gemm_probe.py --shape 2,4, operands warm in L2) shows them 3-8 % faster
  if (M <= 32 * PBHC_ACT_MAX_BLOCKS && M >= 8192 && (forward ? N <= 128 : K <= 128)) return 4;
  if (N <= 128 && M % 96 == 0 && (M / 96) % 256 == 0) return 1;
  // few rows (the rollout's 4 096-row policy / critic forward): 64 x 64 tiles keep every CU busy where 64 x 128 tiles would leave fewer than two
  // workgroups per CU (measured at 4 096 rows: the six hidden layers 139 us against 181 us for library GEMM + ELU)
  if (forward && (long)((M + 63) / 64) * ((N + 127) / 128) < 512) return 3;
  return 2;
}
static int g_force_shape = -1;

extern "C" {

void pbhc_gemm_debug_force_shape(int shape) {
  if (shape < 0) { g_force_shape = -1; g_dbg = 0; g_variant = 0; return; }
  g_force_shape = (shape & 0xff) == 0xff ? -1 : shape & 0xff;
  g_dbg = (shape >> 8) & 0xff;
  g_variant = (shape >> 16) & 0xff;
}

int pbhc_linear_act_fwd(const float* x, const float* w, const float* bias, float* y, float* pre, int M, int N, int K, int act, void* stream) {
  GEMM_ARG(x && w && y && M >= 1 && N >= 1 && K >= 1 && act >= 0 && act <= 3);
  GEMM_ARG(((uintptr_t)x & 3) == 0 && ((uintptr_t)w & 3) == 0);
  GEMM_HIP(gemm_dispatch<0>(pick_shape(M, N, K, g_force_shape, true), x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, (hipStream_t)stream));
  return PBHC_OK;
}

int pbhc_linear_act_fwd_out(const float* x, const float* w, const float* bias, float* y, float* pre, int M, int N, int K, int act, const float* w_out,
                            const float* b_out, int NO, float* out, void* stream) {
  GEMM_ARG(x && w && y && w_out && out && M >= 1 && N == 128 && K >= 4 && NO >= 1 && NO <= 32 && act >= 0 && act <= 3);
  GEMM_ARG(((uintptr_t)x & 3) == 0 && ((uintptr_t)w & 3) == 0 && ((uintptr_t)w_out & 3) == 0);
  GEMM_ARG((size_t)M * K < (1u << 30) && (size_t)N * K < (1u << 30));
  const GemmOutLayer fo{w_out, b_out, out, NO};
  hipStream_t st = (hipStream_t)stream;
  // (g_variant 2 / 3, measurement: BK 16 with a 3- / 4-deep ring for this short-K tile)
  if (g_variant == 2) GEMM_HIP((gemm2_launch<0, 1, 4, 1, 1, 16, 3>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, 0, 0, 0, 0, 1, fo)));
  else if (g_variant == 3) GEMM_HIP((gemm2_launch<0, 1, 4, 1, 1, 16, 4>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, 0, 0, 0, 0, 1, fo)));
  else GEMM_HIP((gemm2_launch<0, 1, 4, 1, 1, 32, 2>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, 0, 0, 0, 0, 1, fo)));
  return PBHC_OK;
}

int pbhc_linear_dgrad_act(const float* dy, const float* w, const float* saved, float* dx, float* scratch, int* num_row_blocks, int M, int N, int K,
                          int act, void* stream) {
  GEMM_ARG(dy && w && dx && M >= 1 && N >= 1 && K >= 1 && act >= 0 && act <= 3 && (act == 0 || saved) && (!scratch || num_row_blocks));
  GEMM_ARG(((uintptr_t)dy & 3) == 0 && ((uintptr_t)w & 3) == 0);
  int shape = pick_shape(M, N, K, g_force_shape, false);
  if (shape == 3) shape = 2;
  if ((M + tile_bm(shape) - 1) / tile_bm(shape) > PBHC_ACT_MAX_BLOCKS) shape = 0;
  const int nb = (M + tile_bm(shape) - 1) / tile_bm(shape);
  GEMM_ARG(!scratch || nb <= PBHC_ACT_MAX_BLOCKS);
  if (num_row_blocks) *num_row_blocks = nb;
  // A = dy [M, K] (K = out_features of the layer, the reduction), B = W [K, N] (N = in_features, contiguous)
  GEMM_HIP(gemm_dispatch<1>(shape, dy, w, nullptr, saved, dx, nullptr, scratch, M, N, K, act, (hipStream_t)stream));
  return PBHC_OK;
}


int pbhc_linear_act_fwd_strided(const float* x, int lda, long long x_batch_stride, const float* w, const float* bias, float* y, float* pre, int ldc,
                                long long y_batch_stride, int batches, int M, int N, int K, int act, void* stream) {
  GEMM_ARG(x && w && y && M >= 1 && N >= 1 && K >= 4 && act >= 0 && act <= 3 && lda >= K && ldc >= N && batches >= 1 && batches <= 65535);
  GEMM_ARG(((uintptr_t)x & 3) == 0 && ((uintptr_t)w & 3) == 0 && x_batch_stride >= 0 && y_batch_stride >= 0);
  GEMM_ARG((size_t)M * (size_t)lda < (1u << 30) && (size_t)N * (size_t)K < (1u << 30));      // 32-bit byte offsets inside one batch
  hipStream_t st = (hipStream_t)stream;
  const int shape = pick_shape(M, N, K, g_force_shape, true);
  hipError_t e;
  if (shape == 3) e = gemm2_launch<0, 2, 2, 1, 1, 32, 2>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, lda, ldc, x_batch_stride, y_batch_stride, batches);
  else if (shape == 1) e = gemm2_launch<0, 1, 4, 3, 1, 32, 2>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, lda, ldc, x_batch_stride, y_batch_stride, batches);
  else if (shape == 0) e = gemm2_launch<0, 2, 2, 2, 2, 32, 2>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, lda, ldc, x_batch_stride, y_batch_stride, batches);
  else e = gemm2_launch<0, 1, 4, 2, 1, 32, 2>(x, w, bias, nullptr, y, pre, nullptr, M, N, K, act, st, lda, ldc, x_batch_stride, y_batch_stride, batches);
  GEMM_HIP(e);
  return PBHC_OK;
}

int pbhc_linear_wgrad_parts(int M, int N, int K) {
  // number of row groups: as many workgroups as fill the 512 resident slots (two 64 KB workgroups per CU) in one round; a group takes a run of
  // 32-row stages (uneven runs allowed), at least four of them
  if (M < 32 || (M & 31) || N < 4 || (N & 3) || K < 4) return 0;
  const int tiles = ((N + WG_BM - 1) / WG_BM) * ((K + WG_BN - 1) / WG_BN);
  const int stages = M / WG_BK;
  int P = 512 / tiles;
  if (P > stages / 4) P = stages / 4;
  if (P > 256) P = 256;
  if (P >= 8 && (P & ~7) * 10 >= P * 9) P &= ~7;           // whole row groups per XCD (k_wgrad's workgroup order) when that costs < 10 % of the workgroups
  return P < 1 ? 1 : P;
}

int pbhc_linear_wgrad(const float* dy, const float* x, float* dw, float* scratch, int M, int N, int K, void* stream) {
  GEMM_ARG(dy && x && dw && scratch && M >= 1 && N >= 1 && K >= 1);
  GEMM_ARG(((uintptr_t)dy & 3) == 0 && ((uintptr_t)x & 3) == 0);
  const int P = pbhc_linear_wgrad_parts(M, N, K);
  GEMM_ARG(P >= 1 && (M & 31) == 0 && (N & 3) == 0 && K >= 4 && (size_t)M * (size_t)(N > K ? N : K) < (1u << 30));
  hipStream_t st = (hipStream_t)stream;
  const int tn = (N + WG_BM - 1) / WG_BM, tk = (K + WG_BN - 1) / WG_BN;
  constexpr int RING = 2 * WG_BK * (WG_BM + WG_BN) * 4, IMAGE = WG_BM * (WG_BN + 4) * 4;
  constexpr int LDS_BYTES = RING > IMAGE ? RING : IMAGE;          // 67 584 B: above the 64 KB default
  static bool attr_set = false;
  if (!attr_set) {
    GEMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_wgrad, dim3(tn * tk * P), dim3(GEMM_T), LDS_BYTES, st, dy, x, scratch, N, K, M / WG_BK, tk, tn * tk, P);
  const int n = N * K;
  hipLaunchKernelGGL(k_wgrad_sum, dim3((n / 4 + 256) / 256), dim3(256), 0, st, scratch, dw, n, P);
  GEMM_HIP(hipGetLastError());
  return PBHC_OK;
}

}  // extern "C"
