// Whole-stack forward of a Linear / activation stack (the policy and the critic of the rollout: `BaseModule.forward` under no_grad,
// reference agents/modules/modules.py:47-63 called from mh_ppo.py:286-290 once per control step) as ONE launch on the gfx950 matrix cores.
//
// At the rollout's 4 096 rows a layer-by-layer GEMM chain is launch- and tail-bound: 380 -> 512 -> 256 -> 128 -> 23 is four launches whose
// later layers have 128 / 64 / 16 output tiles for 256 CUs, and every activation makes a round trip through HBM (76 us for 3 GFLOP).  Here a
// workgroup owns 16 ROWS and carries them through every layer:
//   * 4 096 rows = 256 workgroups = one per CU, 8 waves each; the activations of the 16 rows live in two LDS images (ping-pong), never in HBM;
//   * a wave owns a run of 16-column output tiles of the layer (`v_mfma_f32_16x16x4_f32`, f32 in / f32 accumulate: the same arithmetic class
//     as the library GEMM, k-ordered fmaf chains); its B operand — rows of the weight matrix, used by this wave only — goes STRAIGHT from
//     global memory (L2-resident: 1.4 MB / 3.8 MB per network) to the MFMA's registers, from a PACKED copy of the weights (`pbhc_mlp_pack`,
//     once per rollout: the weights are constant over its 24 steps) in which the 64 lanes' 16-byte fragments of one (tile, k-step) are 1 KiB
//     contiguous: lane (j, g) holds W[n0 + j][k0 + 4 g .. + 3], i.e. the k index of MFMA step s is k0 + 4 g + s for lane group g — the A
//     fragment is read from LDS with the same permutation (one ds_read_b128 per 16 k), and a k permutation common to both operands leaves
//     the product unchanged.  (Reading the fragments from the nn.Linear layout — 16 rows x 64 bytes per wave-instruction, consecutive lanes
//     on different rows — cost ~77 cycles per load in the texture addresser: 51 us for the actor at ANY row count from 64 to 4 096;)
//   * three k-steps of B loads are in flight per wave (register ring of four), the accumulators get bias + activation and are stored as the next
//     layer's A image (row pitch = 4 mod 32 words: conflict-free ds_write_b32 per 32-lane half); the last layer goes to global memory.
// Rounding differs from the layer-by-layer path by summation order only (pinned against fp64 by tests/test_gpu_gemm.py).
// Tried and not kept: requesting a layer's first ring stages before the previous layer's epilogue and barrier (a by-value register ring carried
// across the tile runs): 37.7 -> 42.8 us — the conditional use of the carried ring makes the first round of every run drain the load queue.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pbhc_hip.h"
#include "pbhc_math.h"

extern thread_local char g_pbhc_err[512];
#define MLP_FAIL(code, ...) do { snprintf(g_pbhc_err, sizeof(g_pbhc_err), __VA_ARGS__); return (code); } while (0)
#define MLP_ARG(cond) do { if (!(cond)) MLP_FAIL(PBHC_EINVAL, "%s: argument check failed: %s", __func__, #cond); } while (0)
#define MLP_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) MLP_FAIL(PBHC_EHIP, "%s: %s: %s", __func__, #expr, hipGetErrorString(e_)); } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) MlpF4U { f32x4 v; };

#ifndef MLP_WAVES
#define MLP_WAVES 8
#endif
#define MLP_T (64 * MLP_WAVES)
#define MLP_ROWS 16
#ifndef MLP_NST
#define MLP_NST 4                                          // B register ring: k-steps of loads in flight per wave (2 / 3 / 4 measured: 38.6 / 38.2 / 37.1 us)
#endif

struct MlpArgs {
  const float* w[PBHC_MLP_MAX_LAYERS];
  const float* b[PBHC_MLP_MAX_LAYERS];
  int dim[PBHC_MLP_MAX_LAYERS + 1];
  int nl, act, M, ldy, pitch0, pitch1;                     // pitch0 / pitch1: row pitch (floats) of the even / odd activation image
  // the input rows: up to three column segments laid side by side (`torch.cat([...], -1)` without the copy: observation slab | encoder outputs)
  const float* xs[PBHC_MLP_MAX_SEGS];
  int xw[PBHC_MLP_MAX_SEGS], xld[PBHC_MLP_MAX_SEGS];
  int nseg, vec;                                           // vec: every segment can be read in 16-byte pieces (checked on the host)
  float* y;
  // optional sampling epilogue of the last layer (the rollout's policy: a ~ Normal(mu, std), pbhc_policy_sample's arithmetic and Philox keys)
  const float* std;
  const double* counter;
  unsigned long long seed;
  int counter_offset;
  float *actions, *action_mean, *action_sigma, *logp;
};

__host__ __device__ __forceinline__ int mlp_pitch(int k) { return ((k + 15) & ~15) + 4; }   // zero-padded to a whole k-step; 4 mod 32 words where it matters

// (exp as in csrc/pbhc_gemm.hip: the hardware's 2^x on x log2 e, so that the rollout's stack and the update's layer kernels apply the same activation)
#ifndef PBHC_GEMM_LIBM_EXP
#define MLP_EXP(x) __builtin_amdgcn_exp2f((x) * 1.44269504088896341f)
#define MLP_RCP(x) __builtin_amdgcn_rcpf(x)                    // (SiLU's 1 / (1 + e^-x): v_rcp_f32, 1 ulp, for an IEEE division's ~10 instructions)
#else
#define MLP_EXP(x) expf(x)
#define MLP_RCP(x) (1.0f / (x))
#endif
__device__ __forceinline__ float mlp_act(int act, float v) {
  if (act == 1) return v > 0.0f ? v : MLP_EXP(v) - 1.0f;    // ELU: exp(x) - 1 in f32
  if (act == 2) return v * MLP_RCP(1.0f + MLP_EXP(-v));          // SiLU: x / (1 + exp(-x))
  if (act == 3) return v > 0.0f ? v : 0.0f;
  return v;
}

// T output tiles (16 columns each) of one layer for this wave: acc[t] += A[16, Kp] . Wp[tile_t]^T, Kp = K rounded up to 16 (both operands are
// zero there), `nks` = Kp / 16 k-steps, Wp = the layer's packed weights [tile][k-step][lane][4].
template <int T>
__device__ __forceinline__ void mlp_tiles(const float* __restrict__ Wp, int nks, const int (&tile)[T], const float* __restrict__ A, int P, f32x4 (&acc)[T]) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  unsigned int woff[T];                                    // byte offset of this lane's fragment of the tile's k-step 0 (< 2^30: checked on the host)
#pragma unroll
  for (int t = 0; t < T; ++t) woff[t] = ((unsigned int)(tile[t] * nks) * 64u + (unsigned int)lane) * 16u;
  const float* arow = A + j * P + 4 * g;
  f32x4 breg[MLP_NST][T];
  // every ring load is UNCONDITIONAL (k-step clamped to the last one: a few redundant loads at the end of a tile run): a load under a
  // branch makes the compiler's waitcnt pass assume the worst at the join and drain the ring (vmcnt(0)) once per unrolled round
  const int klast = nks - 1;
#pragma unroll
  for (int s = 0; s < MLP_NST - 1; ++s) {
#pragma unroll
    for (int t = 0; t < T; ++t) breg[s][t] = *reinterpret_cast<const f32x4*>((const char*)Wp + woff[t] + 1024u * (unsigned int)min(s, klast));
  }
  f32x4 a = *reinterpret_cast<const f32x4*>(arow);         // A fragment of the step at hand; the next one is read under this step's MFMAs
  for (int k0 = 0; k0 < nks; k0 += MLP_NST) {
#pragma unroll
    for (int u = 0; u < MLP_NST; ++u) {
      const int k = k0 + u;
#pragma unroll
      for (int t = 0; t < T; ++t)
        breg[(u + MLP_NST - 1) % MLP_NST][t] = *reinterpret_cast<const f32x4*>((const char*)Wp + woff[t] + 1024u * (unsigned int)min(k + MLP_NST - 1, klast));
      if (k < nks) {
        const f32x4 an = *reinterpret_cast<const f32x4*>(arow + 16 * min(k + 1, klast));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], breg[u][t][s], acc[t], 0, 0, 0);
        }
        a = an;
      }
    }
  }
}

// tiles [first, first + T) of layer `l` for this wave: product, bias, activation, store (next A image, or global rows for the last layer)
template <int T>
__device__ __forceinline__ void mlp_run(const MlpArgs& a, int l, int first, int ntiles, const float* __restrict__ A, int P, float* __restrict__ O, int PO, int row0) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int K = a.dim[l], N = a.dim[l + 1];
  int tile[T];
#pragma unroll
  for (int t = 0; t < T; ++t) tile[t] = min(first + t, ntiles - 1);
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  mlp_tiles<T>(a.w[l], (K + 15) >> 4, tile, A, P, acc);
  const bool last = l == a.nl - 1;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (first + t >= ntiles) continue;                      // (a clamped duplicate of the wave's last tile)
    const int n = 16 * tile[t] + j;
    const float bias = (a.b[l] && n < N) ? a.b[l][n] : 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = 4 * g + v;
      float val = acc[t][v] + bias;
      if (!last) {
        val = n < N ? mlp_act(a.act, val) : 0.0f;         // columns [N, ceil16(N)) of the next A image are zeros
        O[r * PO + n] = val;
      } else {
        const bool ok = n < N && row0 + r < a.M;
        if (ok && a.y) a.y[(size_t)(row0 + r) * a.ldy + n] = val;
        if (a.actions) {
          // mh_ppo.py:286-296 on the accumulators: action = mu + std * z (Box-Muller on one Philox call keyed by row / step counter / column,
          // exactly k_policy_sample's), the column's log-prob term -> the idle activation image, summed per row after the barrier
          float lp = 0.0f;
          if (ok) {
            const size_t i = (size_t)(row0 + r) * N + n;
            const float m = val, sg = a.std[n];
            uint32_t o[4];
            pbhc::philox4x32((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)(row0 + r), (uint32_t)a.counter[0] + (uint32_t)a.counter_offset, 0x5A4Du, (uint32_t)n, o);
            const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
            const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
            const float act = m + sg * z;
            a.actions[i] = act; a.action_mean[i] = m; a.action_sigma[i] = sg;
            const float d = act - m;
            lp = -(d * d) / (2.0f * sg * sg) - logf(sg) - 0.9189385332046727f;
          }
          O[r * PO + n] = lp;
        }
      }
    }
  }
}

__global__ __launch_bounds__(MLP_T) void k_mlp_fwd(MlpArgs a) {
  extern __shared__ float mlp_smem[];
  float* const img0 = mlp_smem;                            // (offsets into the one shared array, so that every access stays a DS instruction)
  const int off1 = MLP_ROWS * a.pitch0;
  const int row0 = blockIdx.x * MLP_ROWS;
  const int wave = threadIdx.x >> 6;
  {
    // the 16 input rows -> image 0, zero-padded to the 16-wide k-step (rows past M: the last row again, never stored)
    const int K = a.dim[0], P = a.pitch0, Kp = (K + 15) & ~15;
    // column c of the concatenated row -> (segment, column inside it); segments before the last are whole 16-byte pieces on the vector path
    auto seg_of = [&](int c, int& cc) {
      int sg = 0;
      cc = c;
      if (a.nseg > 1 && cc >= a.xw[0]) { cc -= a.xw[0]; sg = 1; }
      if (a.nseg > 2 && sg == 1 && cc >= a.xw[1]) { cc -= a.xw[1]; sg = 2; }
      return sg;
    };
    if (a.vec) {
      const int cpr = Kp >> 2;                             // 16-byte chunks per row
      for (int i = threadIdx.x; i < MLP_ROWS * cpr; i += MLP_T) {
        const int r = i / cpr, c = (i - r * cpr) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < K) {
          int cc;
          const int sg = seg_of(c, cc);
          const float* base = sg == 0 ? a.xs[0] : sg == 1 ? a.xs[1] : a.xs[2];
          const int ld = sg == 0 ? a.xld[0] : sg == 1 ? a.xld[1] : a.xld[2];
          v = *reinterpret_cast<const f32x4*>(base + (size_t)min(row0 + r, a.M - 1) * ld + cc);      // may cover row padding: masked below
          if (c + 1 >= K) v[1] = 0.f;
          if (c + 2 >= K) v[2] = 0.f;
          if (c + 3 >= K) v[3] = 0.f;
        }
        *reinterpret_cast<f32x4*>(img0 + r * P + c) = v;
      }
    } else {
      for (int i = threadIdx.x; i < MLP_ROWS * Kp; i += MLP_T) {
        const int r = i / Kp, c = i - r * Kp;
        float v = 0.0f;
        if (c < K) {
          int cc;
          const int sg = seg_of(c, cc);
          const float* base = sg == 0 ? a.xs[0] : sg == 1 ? a.xs[1] : a.xs[2];
          const int ld = sg == 0 ? a.xld[0] : sg == 1 ? a.xld[1] : a.xld[2];
          v = base[(size_t)min(row0 + r, a.M - 1) * ld + cc];
        }
        img0[r * P + c] = v;
      }
    }
  }
  __syncthreads();
  for (int l = 0; l < a.nl; ++l) {
    const float* A = mlp_smem + ((l & 1) ? off1 : 0);
    float* O = mlp_smem + ((l & 1) ? 0 : off1);
    const int P = (l & 1) ? a.pitch1 : a.pitch0, PO = (l & 1) ? a.pitch0 : a.pitch1;
    const int ntiles = (a.dim[l + 1] + 15) >> 4;
    const int per = (ntiles + MLP_WAVES - 1) / MLP_WAVES;   // tiles per wave (wave-uniform run [wave * per, ...))
    int first = wave * per;
    const int end = min(first + per, ntiles);
    while (first < end) {
      const int n = end - first;
      if (n >= 4) { mlp_run<4>(a, l, first, ntiles, A, P, O, PO, row0); first += 4; }
      else if (n == 3) { mlp_run<3>(a, l, first, ntiles, A, P, O, PO, row0); first += 3; }
      else if (n == 2) { mlp_run<2>(a, l, first, ntiles, A, P, O, PO, row0); first += 2; }
      else { mlp_run<1>(a, l, first, ntiles, A, P, O, PO, row0); first += 1; }
    }
    __syncthreads();
    if (l == a.nl - 1 && a.actions && threadIdx.x < MLP_ROWS && row0 + (int)threadIdx.x < a.M) {
      // log-prob of the row: its columns' terms in column order (a fixed order: deterministic)
      const int N = a.dim[a.nl];
      float lp = 0.0f;
      for (int n = 0; n < N; ++n) lp += O[threadIdx.x * PO + n];
      a.logp[row0 + threadIdx.x] = lp;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The general-tracking ConvEncoder (reference: agents/modules/encoder_modules.py:22-107 — Linear + ReLU per time step, two Conv1d + activation
// over the time axis that leave 3 positions, a Linear on the flattened result) under no_grad as ONE launch: the rollout evaluates it for
// 4 096 rows per control step, where its four layers as separate GEMM launches are 62 us of launch / tile overhead for 1.8 GFLOP.
// A workgroup owns 16 envs and carries them through all four layers with the tile routine of the stack kernel above (packed weights straight
// from L2 into the MFMA's B registers):
//   layer 1  rows = the 16 envs at ONE time step: the A fragments come straight from the observation slab (global, 16 bytes per lane; the
//            columns a k-step reads past the step's `d` are the next step's values against zero weights) -> image h [env][t * H + c];
//   conv 1/2 output position l reads k consecutive time steps = ONE contiguous run of k * C floats of the image below: A = image + l * s * C,
//            same row pitch -> images c1 [env][l * O1 + c], c2 [env][l * O2 + c];
//   output   Linear on c2's rows as they lie (its columns re-ordered to (l, c) by the caller) -> global [env][E].
// Every image row ends in zeroed padding that covers the last k-step's over-read.  Jobs (position x pair of 16-column tiles) go round-robin
// over the 8 waves.  Rounding differs from the layer-by-layer path by summation order only.
struct EncArgs {
  const float* x; int ldx;
  const float *w1, *b1, *wc1, *bc1, *wc2, *bc2, *wo, *bo;      // packed weights (pbhc_mlp_pack), biases
  float* y; int ldy;
  int M, T, d, H, O1, k1, s1, L1, O2, k2, s2, L2, E, act;
  int ph, pc1, pc2;                                        // row pitches (floats) of the three LDS images
};

struct EncTrue { static constexpr bool value = true; };
struct EncFalse { static constexpr bool value = false; };
// one job: acc[t] = A[16, K] . Wp[tile_t]^T for two column tiles; A either an LDS image (row pitch P) or, GLOBAL_A, global rows `ga` (per-lane
// pointer to row j, column 4 g of the first k-step; <= 8 k-steps, fetched up front)
template <bool GLOBAL_A>
__device__ __forceinline__ void enc_job(const float* __restrict__ Wp, int nks, const int (&tile)[2], const float* __restrict__ A, int P, const float* __restrict__ ga,
                                        f32x4 (&acc)[2]) {
  if (!GLOBAL_A) {
    mlp_tiles<2>(Wp, nks, tile, A, P, acc);
  } else {
    const int lane = threadIdx.x & 63;
    f32x4 areg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) areg[k] = reinterpret_cast<const MlpF4U*>(ga + 16 * min(k, nks - 1))->v;
    unsigned int woff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) woff[t] = ((unsigned int)(tile[t] * nks) * 64u + (unsigned int)lane) * 16u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < nks) {
        f32x4 b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = *reinterpret_cast<const f32x4*>((const char*)Wp + woff[t] + 1024u * (unsigned int)k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[k][q], b[t][q], acc[t], 0, 0, 0);
        }
      }
    }
  }
}

#ifndef ENC_WAVES
#define ENC_WAVES 16                                       // the layers are chains of dependent k-steps fed from L2: more waves per workgroup hide more of it (8: 44.7 us)
#endif
#define ENC_T (64 * ENC_WAVES)
__global__ __launch_bounds__(ENC_T) void k_conv_encoder_fwd(EncArgs a) {
  extern __shared__ float mlp_smem[];
  float* const himg = mlp_smem;
  float* const c1 = himg + MLP_ROWS * a.ph;
  float* const c2 = c1 + MLP_ROWS * a.pc1;
  const int row0 = blockIdx.x * MLP_ROWS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  // zero the padding behind every image row (read by the last k-step of the layer above, against zero weights: it has to be finite)
  {
    const int wh = a.T * a.H, w1 = a.L1 * a.O1, w2 = a.L2 * a.O2;
    for (int i = threadIdx.x; i < MLP_ROWS * (a.ph - wh); i += ENC_T) himg[(i / (a.ph - wh)) * a.ph + wh + i % (a.ph - wh)] = 0.0f;
    for (int i = threadIdx.x; i < MLP_ROWS * (a.pc1 - w1); i += ENC_T) c1[(i / (a.pc1 - w1)) * a.pc1 + w1 + i % (a.pc1 - w1)] = 0.0f;
    for (int i = threadIdx.x; i < MLP_ROWS * (a.pc2 - w2); i += ENC_T) c2[(i / (a.pc2 - w2)) * a.pc2 + w2 + i % (a.pc2 - w2)] = 0.0f;
  }
  // stage: `npos` positions x pairs of column tiles; out(r, pos, n, v) stores one element
  auto stage = [&](auto global_a, const float* Wp, const float* bias, int K, int N, int npos, int act, const float* Aimg, int P, int astep, float* O, int PO, int ostep,
                   bool to_global) {
    constexpr bool GA = decltype(global_a)::value;
    const int nks = (K + 15) >> 4, ntiles = (N + 15) >> 4, npairs = (ntiles + 1) >> 1;
    for (int job = wave; job < npos * npairs; job += ENC_WAVES) {
      const int pos = job / npairs, pr = job - pos * npairs;
      const int tile[2] = {2 * pr, min(2 * pr + 1, ntiles - 1)};
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const float* ga = GA ? a.x + (size_t)min(row0 + j, a.M - 1) * a.ldx + pos * astep + 4 * g : nullptr;
      enc_job<GA>(Wp, nks, tile, GA ? nullptr : Aimg + pos * astep, P, ga, acc);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && 2 * pr + 1 >= ntiles) continue;           // (a clamped duplicate of the last tile)
        const int n = 16 * tile[t] + j;
        const float bv = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 4 * g + v;
          const float val = mlp_act(act, acc[t][v] + bv);
          if (n < N) {
            if (!to_global) O[r * PO + pos * ostep + n] = val;
            else if (row0 + r < a.M) a.y[(size_t)(row0 + r) * a.ldy + n] = val;
          }
        }
      }
    }
  };
  stage(EncTrue{}, a.w1, a.b1, a.d, a.H, a.T, 3, nullptr, 0, a.d, himg, a.ph, a.H, false);
  __syncthreads();
  stage(EncFalse{}, a.wc1, a.bc1, a.k1 * a.H, a.O1, a.L1, a.act, himg, a.ph, a.s1 * a.H, c1, a.pc1, a.O1, false);
  __syncthreads();
  stage(EncFalse{}, a.wc2, a.bc2, a.k2 * a.O1, a.O2, a.L2, a.act, c1, a.pc1, a.s2 * a.O1, c2, a.pc2, a.O2, false);
  __syncthreads();
  stage(EncFalse{}, a.wo, a.bo, a.L2 * a.O2, a.E, 1, 0, c2, a.pc2, 0, nullptr, 0, 0, true);
}

// nn.Linear.weight [N, K] -> [ceil(N/16)][ceil(K/16)][64 lanes][4]: lane (j = l % 16, g = l / 16) of (tile, k-step) holds
// W[16 tile + j][16 kstep + 4 g .. + 3], zeros outside the matrix
__global__ void k_mlp_pack(const float* __restrict__ w, int N, int K, float* __restrict__ out, int nks, size_t total4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int lane = (int)(i & 63);
  const size_t ts = i >> 6;
  const int ks = (int)(ts % (size_t)nks), tile = (int)(ts / (size_t)nks);
  const int n = 16 * tile + (lane & 15), k = 16 * ks + 4 * (lane >> 4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    const float* p = w + (size_t)n * K + k;
    if (k < K) v[0] = p[0];
    if (k + 1 < K) v[1] = p[1];
    if (k + 2 < K) v[2] = p[2];
    if (k + 3 < K) v[3] = p[3];
  }
  reinterpret_cast<f32x4*>(out)[i] = v;
}

extern "C" {

size_t pbhc_mlp_packed_floats(int N, int K) { return (size_t)((N + 15) / 16) * (size_t)((K + 15) / 16) * 256; }

int pbhc_mlp_pack(const float* w, int N, int K, float* packed, void* stream) {
  MLP_ARG(w && packed && N >= 1 && K >= 1 && (((uintptr_t)packed) & 15) == 0);
  const size_t total4 = pbhc_mlp_packed_floats(N, K) / 4;
  hipLaunchKernelGGL(k_mlp_pack, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, N, K, packed, (K + 15) / 16, total4);
  MLP_HIP(hipGetLastError());
  return PBHC_OK;
}

size_t pbhc_mlp_fwd_lds_bytes(const int* dims, int num_layers) {
  int p0 = 0, p1 = 0;
  for (int l = 0; l < num_layers; ++l) {                   // image (l & 1) holds layer l's input
    const int p = mlp_pitch(dims[l]);
    if (l & 1) p1 = p > p1 ? p : p1; else p0 = p > p0 ? p : p0;
  }
  return (size_t)MLP_ROWS * (size_t)(p0 + p1) * sizeof(float);
}

static int mlp_launch(const PbhcMlpInput* in, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, float* y, int ldy,
                      int M, const PbhcMlpSample* smp, void* stream) {
  MLP_ARG(in && weights && biases && dims && (y || smp) && M >= 1 && num_layers >= 1 && num_layers <= PBHC_MLP_MAX_LAYERS && act >= 0 && act <= 3);
  MLP_ARG(in->nseg >= 1 && in->nseg <= PBHC_MLP_MAX_SEGS && (!y || ldy >= dims[num_layers]));
  MlpArgs a;
  {
    int total = 0;
    a.vec = 1;
    for (int sgi = 0; sgi < PBHC_MLP_MAX_SEGS; ++sgi) {
      const bool used = sgi < in->nseg;
      a.xs[sgi] = used ? in->x[sgi] : in->x[0];
      a.xw[sgi] = used ? in->width[sgi] : 0;
      a.xld[sgi] = used ? in->ld[sgi] : 0;
      if (!used) continue;
      MLP_ARG(in->x[sgi] && in->width[sgi] >= 1 && in->ld[sgi] >= in->width[sgi] && (((uintptr_t)in->x[sgi]) & 3) == 0);
      total += in->width[sgi];
      // 16-byte pieces: aligned base and pitch, the piece that straddles the segment's end stays inside its row, inner segments end on a piece
      if ((in->ld[sgi] & 3) || (((uintptr_t)in->x[sgi]) & 15) || in->ld[sgi] < ((in->width[sgi] + 3) & ~3) || (sgi + 1 < in->nseg && (in->width[sgi] & 3))) a.vec = 0;
    }
    MLP_ARG(total == dims[0]);
    a.nseg = in->nseg;
  }
  a.std = nullptr; a.counter = nullptr; a.seed = 0; a.counter_offset = 0;
  a.actions = a.action_mean = a.action_sigma = a.logp = nullptr;
  if (smp) {
    MLP_ARG(smp->std && smp->counter && smp->actions && smp->action_mean && smp->action_sigma && smp->logp);
    a.std = smp->std; a.counter = smp->counter; a.seed = smp->seed; a.counter_offset = smp->counter_offset;
    a.actions = smp->actions; a.action_mean = smp->action_mean; a.action_sigma = smp->action_sigma; a.logp = smp->logp;
  }
  a.nl = num_layers; a.act = act; a.M = M; a.ldy = ldy; a.y = y;
  a.pitch0 = a.pitch1 = 0;
  for (int l = 0; l <= num_layers; ++l) {
    MLP_ARG(dims[l] >= 1 && dims[l] <= 4096);
    a.dim[l] = dims[l];
  }
  for (int l = 0; l < num_layers; ++l) {
    MLP_ARG(weights[l] && (((uintptr_t)weights[l]) & 15) == 0 && pbhc_mlp_packed_floats(dims[l + 1], dims[l]) < (1u << 28));      // packed (pbhc_mlp_pack)
    a.w[l] = weights[l];
    a.b[l] = biases[l];
    const int p = mlp_pitch(dims[l]);
    if (l & 1) a.pitch1 = p > a.pitch1 ? p : a.pitch1; else a.pitch0 = p > a.pitch0 ? p : a.pitch0;
  }
  if (a.pitch1 == 0) a.pitch1 = 4;
  const size_t lds = (size_t)MLP_ROWS * (size_t)(a.pitch0 + a.pitch1) * sizeof(float);
  MLP_ARG(lds <= 160 * 1024);
  if (lds > 64 * 1024) {
    // the attribute is per DEVICE: raised once on each device that runs a wide stack (the first rollout runs outside any capture)
    static std::atomic<unsigned long long> raised{0};       // bit d: device d has it
    int dev = 0;
    MLP_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
      MLP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(k_mlp_fwd, dim3((M + MLP_ROWS - 1) / MLP_ROWS), dim3(MLP_T), lds, (hipStream_t)stream, a);
  MLP_HIP(hipGetLastError());
  return PBHC_OK;
}

static PbhcMlpInput one_segment(const float* x, int ldx, const int* dims) {
  PbhcMlpInput in;
  for (int i = 0; i < PBHC_MLP_MAX_SEGS; ++i) { in.x[i] = nullptr; in.ld[i] = 0; in.width[i] = 0; }
  in.x[0] = x; in.ld[0] = ldx; in.width[0] = dims ? dims[0] : 0; in.nseg = 1;
  return in;
}

int pbhc_mlp_fwd(const float* x, int ldx, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, float* y, int ldy,
                 int M, void* stream) {
  MLP_ARG(y && dims);
  const PbhcMlpInput in = one_segment(x, ldx, dims);
  return mlp_launch(&in, weights, biases, dims, num_layers, act, y, ldy, M, nullptr, stream);
}

int pbhc_mlp_fwd_sample(const float* x, int ldx, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, int M,
                        const PbhcMlpSample* sample, void* stream) {
  MLP_ARG(sample && dims);
  const PbhcMlpInput in = one_segment(x, ldx, dims);
  return mlp_launch(&in, weights, biases, dims, num_layers, act, nullptr, 0, M, sample, stream);
}

// image row pitch: the row + 16 floats of zeroed over-read room, rounded up to 4 mod 8 words... to a multiple of 4 that is 4, 12, 20 or 28 mod 32
// (16-byte aligned rows whose ds_read_b128 fragments of 8 consecutive rows fall into different banks)
static int enc_pitch(int w) {
  int p = ((w + 16 + 3) & ~3);
  while ((p & 7) != 4) p += 4;
  return p;
}

size_t pbhc_conv_encoder_lds_bytes(const PbhcConvEncoder* e) {
  if (!e) return 0;
  const int L1 = (e->T - e->k1) / e->s1 + 1, L2 = (L1 - e->k2) / e->s2 + 1;
  return (size_t)MLP_ROWS * (size_t)(enc_pitch(e->T * e->H) + enc_pitch(L1 * e->O1) + enc_pitch(L2 * e->O2)) * sizeof(float);
}

int pbhc_conv_encoder_fwd(const float* x, int ldx, const PbhcConvEncoder* e, float* y, int ldy, int M, void* stream) {
  MLP_ARG(x && e && y && M >= 1 && e->w1 && e->wc1 && e->wc2 && e->wo && e->act >= 0 && e->act <= 3);
  MLP_ARG(e->T >= 1 && e->d >= 1 && e->d <= 128 && e->H >= 1 && e->O1 >= 1 && e->O2 >= 1 && e->E >= 1 && e->k1 >= 1 && e->s1 >= 1 && e->k2 >= 1 && e->s2 >= 1);
  const int L1 = (e->T - e->k1) / e->s1 + 1, L2 = e->T >= e->k1 ? (L1 - e->k2) / e->s2 + 1 : 0;
  MLP_ARG(e->T >= e->k1 && L1 >= e->k2 && L2 >= 1 && ldy >= e->E);
  // layer 1 reads whole k-steps of 16 floats from the slab: the last step's over-read has to stay inside the row's pitch (and the allocation)
  MLP_ARG(ldx >= (e->T - 1) * e->d + ((e->d + 15) & ~15) && (((uintptr_t)x) & 3) == 0);
  // the conv windows are read as 16-byte LDS fragments: their starts have to be 16-byte aligned
  MLP_ARG(((e->s1 * e->H) & 3) == 0 && ((e->s2 * e->O1) & 3) == 0);
  for (const float* w : {e->w1, e->wc1, e->wc2, e->wo}) MLP_ARG((((uintptr_t)w) & 15) == 0);
  MLP_ARG(pbhc_mlp_packed_floats(e->O1, e->k1 * e->H) < (1u << 28) && pbhc_mlp_packed_floats(e->E, L2 * e->O2) < (1u << 28));
  EncArgs a;
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.M = M;
  a.w1 = e->w1; a.b1 = e->b1; a.wc1 = e->wc1; a.bc1 = e->bc1; a.wc2 = e->wc2; a.bc2 = e->bc2; a.wo = e->wo; a.bo = e->bo;
  a.T = e->T; a.d = e->d; a.H = e->H; a.O1 = e->O1; a.k1 = e->k1; a.s1 = e->s1; a.L1 = L1; a.O2 = e->O2; a.k2 = e->k2; a.s2 = e->s2; a.L2 = L2; a.E = e->E; a.act = e->act;
  a.ph = enc_pitch(e->T * e->H); a.pc1 = enc_pitch(L1 * e->O1); a.pc2 = enc_pitch(L2 * e->O2);
  const size_t lds = pbhc_conv_encoder_lds_bytes(e);
  MLP_ARG(lds <= 160 * 1024);
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    MLP_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
      MLP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_encoder_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(k_conv_encoder_fwd, dim3((M + MLP_ROWS - 1) / MLP_ROWS), dim3(ENC_T), lds, (hipStream_t)stream, a);
  MLP_HIP(hipGetLastError());
  return PBHC_OK;
}

int pbhc_mlp_fwd_cat(const PbhcMlpInput* in, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, float* y, int ldy,
                     int M, const PbhcMlpSample* sample, void* stream) {
  return mlp_launch(in, weights, biases, dims, num_layers, act, y, ldy, M, sample, stream);
}

}  // extern "C"
