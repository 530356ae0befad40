// pbhc_ppo.hip — fused PPO-update kernels (gfx950): the elementwise / reduction work of
// MHPPO._update_ppo (reference: humanoidverse/agents/mh_ppo/mh_ppo.py:433-533) that eager PyTorch runs as
// ~100 tiny launches per minibatch.  The MLP GEMMs stay on PyTorch-ROCm (rocBLAS/hipBLASLt); these kernels
// consume the network outputs (mu, value) and hand back d(loss)/d(mu), d(loss)/d(value), d(loss)/d(std).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

#include "../../include/pbhc_hip.h"

extern thread_local char g_pbhc_err[512];
#define HIP_CHECK(x)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (x);                                                                                 \
    if (e_ != hipSuccess) {                                                                              \
      snprintf(g_pbhc_err, sizeof(g_pbhc_err), "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return PBHC_EHIP;                                                                                  \
    }                                                                                                    \
  } while (0)
#define ARG_CHECK(c)                                                                         \
  do {                                                                                       \
    if (!(c)) {                                                                              \
      snprintf(g_pbhc_err, sizeof(g_pbhc_err), "%s:%d bad argument: %s", __FILE__, __LINE__, #c); \
      return PBHC_EINVAL;                                                                    \
    }                                                                                        \
  } while (0)

#define LOSS_ROWS 8          // rows (samples) per 256-thread workgroup, 32 lanes per row
#define LOSS_NP 8            // scalar partial sums per workgroup: surrogate, value, kl, (pad)

__device__ __forceinline__ float gsum32(float v) {
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
  return v;
}

// One row per 32-lane group: lane a < A <-> action dim a, lane r < R <-> value head r.
// Every 32-lane group keeps its own running sums over the tiles it visits (no barrier inside the tile loop: the loads of consecutive tiles
// overlap); at the end of the block the eight groups are summed in a fixed order into
//   partial[blk][0..2] = sum over the block's rows of (max(s1,s2), sum_r max(l1,l2), kl)
//   gstd_part[blk][a]  = sum over the block's rows of d(surrogate)/d(sigma_a) (before the 1/B).
// (A "last block finishes the job" ticket was tried instead of the second launch: the device-scope release / acquire it needs is an L2
// write-back + invalidate per block on this 8-XCD part — 73 us for the pair against 25 us as two launches.)
__global__ __launch_bounds__(256) void k_ppo_loss(const float* __restrict__ mu, const float* __restrict__ stdp, const float* __restrict__ value,
                                                  const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ old_mu,
                                                  const float* __restrict__ old_sigma, const float* __restrict__ adv, const float* __restrict__ returns,
                                                  const float* __restrict__ old_values, int B, int A, int R, float clip, float value_coef,
                                                  int clipped_value, int kl_v2, float* __restrict__ grad_mu, float* __restrict__ grad_value,
                                                  float* __restrict__ partial, float* __restrict__ gstd_part) {
  __shared__ float sh_s[LOSS_ROWS][4];
  __shared__ float sh_g[LOSS_ROWS][32];
  const int lane = threadIdx.x & 31, lr_ = threadIdx.x >> 5;
  const float invB = 1.0f / (float)B;
  const int ntiles = (B + LOSS_ROWS - 1) / LOSS_ROWS;
  const float sigma = lane < A ? stdp[lane] : 1.0f;            // Normal(mean, mean*0 + std): sigma is the parameter itself
  const float var = sigma * sigma, lsig = logf(sigma);
  float a_surr = 0.0f, a_vl = 0.0f, a_kl = 0.0f, a_sig = 0.0f;  // lane 0: the group's scalar sums; lane a < A: its d sigma sum
  // LOSS_U tiles per trip, ALL their loads requested before the first is used: with one tile per trip a block's six tiles were six dependent
  // round trips to memory (17.7 -> 12.0 us for 20 MB at two blocks per CU; six tiles per trip: the same)
  constexpr int LOSS_U = 3;
  struct TileIn { float m, x, om, os, a, olp, v, ov, ret; };
  for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += LOSS_U * gridDim.x) {
    TileIn in[LOSS_U];
#pragma unroll
    for (int u = 0; u < LOSS_U; ++u) {
      const int tile = tile0 + u * gridDim.x;
      const int row = tile * LOSS_ROWS + lr_;
      const bool valid = tile < ntiles && row < B;
      in[u] = TileIn{0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (valid && lane < A) {
        const size_t i = (size_t)row * A + lane;
        in[u].m = mu[i]; in[u].x = actions[i]; in[u].om = old_mu[i]; in[u].os = old_sigma[i];
      }
      if (valid) { in[u].a = adv[row]; in[u].olp = old_logp[row]; }
      if (valid && lane < R) {
        const size_t i = (size_t)row * R + lane;
        in[u].v = value[i]; in[u].ov = old_values[i]; in[u].ret = returns[i];
      }
    }
#pragma unroll
    for (int u = 0; u < LOSS_U; ++u) {
    const int tile = tile0 + u * gridDim.x;
    const int row = tile * LOSS_ROWS + lr_;
    const bool valid = tile < ntiles && row < B;
    float lp = 0.0f, kl = 0.0f, dmu_c = 0.0f, dsig_c = 0.0f;     // per-lane pieces
    if (valid && lane < A) {
      const float m = in[u].m, x = in[u].x, om = in[u].om, os = in[u].os;
      const float d = x - m;
      // torch.distributions.Normal.log_prob: -((x-mu)^2)/(2 var) - log(sigma) - log(sqrt(2 pi))
      lp = -(d * d) / (2.0f * var) - lsig - 0.9189385332046727f;
      // mh_ppo.py:453 adds the 1e-5 to the ratio, ppo_mimic.py:624 to the old sigma
      kl = (kl_v2 ? logf(sigma / (os + 1.0e-5f)) : logf(sigma / os + 1.0e-5f)) + (os * os + (om - m) * (om - m)) / (2.0f * sigma * sigma) - 0.5f;
      dmu_c = d / var;                       // d logp / d mu
      dsig_c = d * d / (var * sigma) - 1.0f / sigma;   // d logp / d sigma
    }
    const float logp = gsum32(lp);
    const float klrow = gsum32(kl);
    float surr = 0.0f, coef = 0.0f;
    if (valid) {
      const float a = in[u].a;
      const float ratio = expf(logp - in[u].olp);
      const float rc = fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
      const float s1 = -a * ratio, s2 = -a * rc;
      surr = fmaxf(s1, s2);
      // d max(s1,s2)/d logp : torch.max splits ties evenly; inside the clip range s2 == s1 and d s2 = d s1
      const bool inrange = (ratio >= 1.0f - clip) && (ratio <= 1.0f + clip);
      float g;
      if (s1 > s2) g = -a * ratio;
      else if (s1 < s2) g = inrange ? -a * ratio : 0.0f;
      else g = 0.5f * (-a * ratio) + 0.5f * (inrange ? -a * ratio : 0.0f);
      coef = g * invB;
    }
    if (valid && lane < A) grad_mu[(size_t)row * A + lane] = coef * dmu_c;
    // value loss (summed over heads, mean over rows) and its gradient
    float vl = 0.0f;
    if (valid && lane < R) {
      const size_t i = (size_t)row * R + lane;
      const float v = in[u].v, ov = in[u].ov, ret = in[u].ret;
      float g;
      if (clipped_value) {
        const float dv = v - ov;
        const float dc = fminf(fmaxf(dv, -clip), clip);
        const float vc = ov + dc;
        const float l1 = (v - ret) * (v - ret), l2 = (vc - ret) * (vc - ret);
        vl = fmaxf(l1, l2);
        const float g1 = 2.0f * (v - ret);
        const float g2 = (dv >= -clip && dv <= clip) ? 2.0f * (vc - ret) : 0.0f;
        g = l1 > l2 ? g1 : (l1 < l2 ? g2 : 0.5f * g1 + 0.5f * g2);
      } else {
        vl = (ret - v) * (ret - v);
        g = 2.0f * (v - ret);
      }
      grad_value[i] = value_coef * g * invB;
    }
    const float vlrow = gsum32(vl);
    if (valid) {
      a_surr += surr;
      a_vl += vlrow;
      a_kl += klrow;
      if (lane < A) a_sig += coef * dsig_c;
    }
    }
  }
  if (lane == 0) { sh_s[lr_][0] = a_surr; sh_s[lr_][1] = a_vl; sh_s[lr_][2] = a_kl; }
  sh_g[lr_][lane] = lane < A ? a_sig : 0.0f;
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.0f;
    for (int g = 0; g < LOSS_ROWS; ++g) t += sh_s[g][threadIdx.x];
    partial[(size_t)blockIdx.x * LOSS_NP + threadIdx.x] = t;
  }
  if (threadIdx.x >= 32 && threadIdx.x < 64) {
    const int a = threadIdx.x - 32;
    float t = 0.0f;
    for (int g = 0; g < LOSS_ROWS; ++g) t += sh_g[g][a];
    gstd_part[(size_t)blockIdx.x * 32 + a] = t;
  }
}

// Fixed-order second stage: loss scalars, d(loss)/d(std) (surrogate part + entropy bonus), and the adaptive-KL learning-rate rule
// (mh_ppo.py:455-466) on the device.  One block of 1024 threads = 32 chunks of blocks x 32 columns.  The partials were written by the previous
// launch on all eight XCDs and arrive through memory (~2 us a trip), so a thread issues ALL its loads (<= 16 with the entry's 512-block cap)
// before the first add, and sums them in block order.
#define RED_T 1024
#define RED_CH (RED_T / 32)
__global__ __launch_bounds__(RED_T) void k_ppo_reduce(const float* __restrict__ partial, const float* __restrict__ gstd_part, const float* __restrict__ stdp,
                                                      int nblocks, int B, int A, float entropy_coef, float desired_kl, int adapt_lr,
                                                      float* __restrict__ grad_std, float* __restrict__ scalars, float* __restrict__ lr,
                                                      float* __restrict__ scalars_acc) {
  __shared__ double sh_g[RED_CH][33];
  __shared__ double sh_s[RED_CH][4];
  __shared__ float sh_e[32];
  __shared__ float sh_m[4];
  const int col = threadIdx.x & 31, chunk = threadIdx.x >> 5;
  const float lr0 = threadIdx.x < 2 ? lr[threadIdx.x] : 0.0f;       // in flight with the partials
  const float sig = col < A ? stdp[col] : 1.0f;
  float gq[16], sq[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int b = chunk + u * RED_CH;
    gq[u] = b < nblocks ? gstd_part[(size_t)b * 32 + col] : 0.0f;
    sq[u] = (b < nblocks && col < 3) ? partial[(size_t)b * LOSS_NP + col] : 0.0f;
  }
  double sg = 0.0, ss = 0.0;
#pragma unroll
  for (int u = 0; u < 16; ++u) { sg += (double)gq[u]; ss += (double)sq[u]; }
  for (int b = chunk + 16 * RED_CH; b < nblocks; b += RED_CH) {            // (not reached with the entry's 512-block cap)
    sg += (double)gstd_part[(size_t)b * 32 + col];
    if (col < 3) ss += (double)partial[(size_t)b * LOSS_NP + col];
  }
  sh_g[chunk][col] = sg;
  if (col < 3) sh_s[chunk][col] = ss;
  if (chunk == 0) sh_e[col] = col < A ? 0.5f + 0.9189385332046727f + logf(sig) : 0.0f;   // entropy terms (mean over rows of identical values)
  __syncthreads();
  if (threadIdx.x < A) {
    double t = 0.0;
    for (int ch = 0; ch < RED_CH; ++ch) t += sh_g[ch][threadIdx.x];
    // actor_loss = surrogate - entropy_coef * entropy ; entropy = sum_a (0.5 + 0.5 log(2 pi) + log sigma_a)
    grad_std[threadIdx.x] = (float)t - entropy_coef / sig;
  }
  if (threadIdx.x >= 32 && threadIdx.x < 35) {              // wave 0's upper half: the three scalar columns, each its own lane
    const int k = threadIdx.x - 32;
    double t = 0.0;
    for (int ch = 0; ch < RED_CH; ++ch) t += sh_s[ch][k];
    const float mean = (float)(t / (double)B);
    const int slot = k == 2 ? 3 : k;                        // surrogate loss, value loss, (entropy), kl
    scalars[slot] = mean;
    if (scalars_acc) scalars_acc[slot] += mean;
    sh_m[k] = mean;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ent = 0.0f;
    for (int a = 0; a < A; ++a) ent += sh_e[a];
    scalars[2] = ent;
    if (scalars_acc) scalars_acc[2] += ent;
  }
  if (threadIdx.x < 2 && adapt_lr) {
    const float kl_mean = sh_m[2];
    float l = lr0;
    if (kl_mean > desired_kl * 2.0f) l = fmaxf(1e-5f, l / 1.5f);
    else if (kl_mean < desired_kl / 2.0f && kl_mean > 0.0f) l = fminf(1e-2f, l * 1.5f);
    lr[threadIdx.x] = l;
  }
}

// ---- activation backward fused with the bias gradient ------------------------------------------------
// dz = dy * act'(saved) (written in place over dy when dz == dy) and grad_bias[c] = sum_rows dz[r][c], one pass over the [B,n] slab:
// what eager PyTorch does as an elu_backward / silu_backward launch plus a column-sum launch per layer.
// act: 0 none, 1 ELU(alpha 1) from the activation OUTPUT (elu' = y > 0 ? 1 : y + 1), 2 SiLU from the PRE-activation, 3 ReLU from the output.
// Thread <-> column (coalesced rows), ACT_RPB row groups per block, fixed-order two-stage column sums.
#define ACT_T 256
// (the SiLU derivative's exp as in csrc/pbhc_gemm.hip: 2^(x log2 e) on v_exp_f32; -DPBHC_GEMM_LIBM_EXP: the library expf)
#ifndef PBHC_GEMM_LIBM_EXP
#define ACT_EXP(x) __builtin_amdgcn_exp2f((x) * 1.44269504088896341f)
#define ACT_RCP(x) __builtin_amdgcn_rcpf(x)                    // (SiLU's 1 / (1 + e^-x): v_rcp_f32, 1 ulp, for an IEEE division's ~10 instructions)
#else
#define ACT_EXP(x) expf(x)
#define ACT_RCP(x) (1.0f / (x))
#endif
__device__ __forceinline__ float act_grad(int act, float s) {
  if (act == 1) return s > 0.0f ? 1.0f : s + 1.0f;
  if (act == 2) { const float sg = ACT_RCP(1.0f + ACT_EXP(-s)); return sg * (1.0f + s * (1.0f - sg)); }
  if (act == 3) return s > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}
// grid (row blocks, column blocks of ACT_T): thread <-> column, `groups` row groups per block when n < ACT_T; rows unrolled by 8 so that
// eight independent loads are in flight per thread.
__global__ __launch_bounds__(ACT_T) void k_act_bwd_bias(const float* __restrict__ dy, const float* __restrict__ saved, int B, int n, int act,
                                                        float* __restrict__ dz, float* __restrict__ part, int rows_per_block) {
  __shared__ float sh[ACT_T];
  const int cpp = n < ACT_T ? n : ACT_T;              // columns per block
  const int groups = ACT_T / cpp;                     // row groups working in parallel (1 when n >= 256)
  const int g = threadIdx.x / cpp, c = blockIdx.y * ACT_T + (threadIdx.x - g * cpp);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(B, r0 + rows_per_block);
  const bool write = act != 0 || dz != dy;
  float s = 0.0f;
  if (g < groups && c < n) {
    int r = r0 + g;
    for (; r + 7 * groups < r1; r += 8 * groups) {
      float v[8], sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const size_t i = (size_t)(r + u * groups) * n + c; v[u] = dy[i]; sv[u] = act ? saved[i] : 0.0f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (act) v[u] *= act_grad(act, sv[u]);
        if (write) dz[(size_t)(r + u * groups) * n + c] = v[u];
        s += v[u];
      }
    }
    for (; r < r1; r += groups) {
      const size_t i = (size_t)r * n + c;
      float v = dy[i];
      if (act) v *= act_grad(act, saved[i]);
      if (write) dz[i] = v;
      s += v;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < cpp && blockIdx.y * ACT_T + threadIdx.x < n) {
    float t = 0.0f;
    for (int k = 0; k < groups; ++k) t += sh[k * cpp + threadIdx.x];
    part[(size_t)blockIdx.x * n + blockIdx.y * ACT_T + threadIdx.x] = t;
  }
}
// second stage: 32 columns x 8 slices per block, fixed order
__device__ __forceinline__ void colsum_final_block(const float* __restrict__ part, int nblocks, int n, float* __restrict__ out, int block);
__global__ __launch_bounds__(ACT_T) void k_colsum_final(const float* __restrict__ part, int nblocks, int n, float* __restrict__ out) {
  colsum_final_block(part, nblocks, n, out, blockIdx.x);
}
struct ColsumJobs { PbhcColsumJob job[PBHC_MAX_COLSUM_JOBS]; int first_block[PBHC_MAX_COLSUM_JOBS]; };
// Two job shapes.  TALL: many partial rows, few columns (bias gradients, the output layer's weight gradient: <= 1024 rows x <= 3k columns) —
// 32 columns x 8 row slices per block.  WIDE: few partial images, many columns (the split-K partials of a hidden layer's weight gradient:
// 2..32 images of up to 768 x 630) — a thread owns four consecutive columns, 1024 per block, and walks the images in order with all loads of
// a run of eight in flight.
__host__ __device__ __forceinline__ bool colsum_wide(int num_row_blocks, int n) { return num_row_blocks <= 64 && n >= 4096 && (n & 3) == 0; }
__host__ __device__ __forceinline__ int colsum_blocks(int num_row_blocks, int n) { return colsum_wide(num_row_blocks, n) ? (n + 1023) / 1024 : (n + 31) / 32; }
__device__ __forceinline__ void colsum_wide_block(const float* __restrict__ part, int P, int n, float* __restrict__ out, int block) {
  const int c = block * 1024 + 4 * (int)threadIdx.x;
  if (c >= n) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = 0;
  for (; b + 8 <= P; b += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (size_t)(b + u) * n + c);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
  }
  for (; b < P; ++b) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)b * n + c);
    s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
  }
  if (((uintptr_t)out & 15) == 0) {
    *reinterpret_cast<float4*>(out + c) = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
  } else {                                                         // a .grad view at an odd offset of the flat gradient buffer
    out[c] = (float)s0; out[c + 1] = (float)s1; out[c + 2] = (float)s2; out[c + 3] = (float)s3;
  }
}
// the second stage of several layers in one launch (the bias gradients of a whole MLP backward)
__global__ __launch_bounds__(ACT_T) void k_colsum_final_multi(ColsumJobs J, int num_jobs) {
  int j = 0;
  while (j + 1 < num_jobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
  const PbhcColsumJob job = J.job[j];
  const int block = (int)blockIdx.x - J.first_block[j];
  if (colsum_wide(job.num_row_blocks, job.n) && ((uintptr_t)job.part & 15) == 0) {
    colsum_wide_block(job.part, job.num_row_blocks, job.n, job.out, block);
  } else if (colsum_wide(job.num_row_blocks, job.n)) {             // (unaligned partial images: the tall form over this block's 1024 columns)
    for (int sub = 0; sub < 32; ++sub)
      if ((block * 32 + sub) * 32 < job.n) { colsum_final_block(job.part, job.num_row_blocks, job.n, job.out, block * 32 + sub); __syncthreads(); }
  } else {
    colsum_final_block(job.part, job.num_row_blocks, job.n, job.out, block);
  }
}
__device__ __forceinline__ void colsum_final_block(const float* __restrict__ part, int nblocks, int n, float* __restrict__ out, int block) {
  // 32 columns x 8 row slices per block (a slice reads rows slice, slice + 8, ...: eight independent loads in flight per thread), then the
  // slices in order 0..7: fixed order, deterministic
  __shared__ double sh[8][32];
  const int col = threadIdx.x & 31, slice = threadIdx.x >> 5, c = block * 32 + col;
  double s = 0.0;
  if (c < n) {
    int b = slice;
    for (; b + 56 < nblocks; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 8 * u) * n + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nblocks; b += 8) s += (double)part[(size_t)b * n + c];
  }
  sh[slice][col] = s;
  __syncthreads();
  if (slice == 0 && c < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][col];
    out[c] = (float)t;
  }
}

// ---- global-norm clipping + Adam over a flat parameter segment ---------------------------------
#define ADAM_T 256
// One or two flat segments (actor, critic) per launch pair.  Segment s: blocks [first_block[s], first_block[s+1]).
#define ADAM_MAX_SEGS 2
struct AdamSegs {
  float *p[ADAM_MAX_SEGS], *g[ADAM_MAX_SEGS], *m[ADAM_MAX_SEGS], *v[ADAM_MAX_SEGS];
  const float* lr[ADAM_MAX_SEGS];
  float* step[ADAM_MAX_SEGS];
  double* part[ADAM_MAX_SEGS];
  float* norm_out[ADAM_MAX_SEGS];
  int n[ADAM_MAX_SEGS], nparts[ADAM_MAX_SEGS], first_norm_block[ADAM_MAX_SEGS + 1], first_adam_block[ADAM_MAX_SEGS + 1];
  int num;
};
struct __attribute__((packed, aligned(4))) AdamF4 { float v[4]; };
__global__ __launch_bounds__(ADAM_T) void k_sqnorm_partial(AdamSegs S) {
  __shared__ double sh[ADAM_T];
  const int seg = (S.num > 1 && (int)blockIdx.x >= S.first_norm_block[1]) ? 1 : 0;
  const int b = blockIdx.x - S.first_norm_block[seg], nb = S.first_norm_block[seg + 1] - S.first_norm_block[seg];
  const float* __restrict__ g = S.g[seg];
  const size_t n = (size_t)S.n[seg];
  double s = 0.0;
  // 16 bytes per lane (the flat segments are only 4-byte aligned relative to each other: dword-aligned dwordx4 accesses)
  for (size_t i = ((size_t)b * ADAM_T + threadIdx.x) * 4; i < n; i += (size_t)nb * ADAM_T * 4) {
    if (i + 4 <= n) {
      const AdamF4 q = *reinterpret_cast<const AdamF4*>(g + i);
      s += (double)q.v[0] * (double)q.v[0] + (double)q.v[1] * (double)q.v[1] + (double)q.v[2] * (double)q.v[2] + (double)q.v[3] * (double)q.v[3];
    } else {
      for (size_t j = i; j < n; ++j) s += (double)g[j] * (double)g[j];
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = ADAM_T / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    S.part[seg][b] = sh[0];
    if (b == 0) S.step[seg][0] += 1.0f;               // Adam's step count: advanced here, read (post-increment) by every block of k_adam_clip
  }
}

// torch.optim.Adam (no amsgrad, no weight decay), after nn.utils.clip_grad_norm_(max_norm):
//   clip = min(1, max_norm / (||g|| + 1e-6));  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(ADAM_T) void k_adam_clip(AdamSegs S, float max_norm, float b1, float b2, float eps, float weight_decay, int zero_grad) {
  __shared__ float s_clip, s_bc1, s_bc2s, s_lr;
  __shared__ double sh[ADAM_T];
  const int seg = (S.num > 1 && (int)blockIdx.x >= S.first_adam_block[1]) ? 1 : 0;
  const int b = blockIdx.x - S.first_adam_block[seg], nb = S.first_adam_block[seg + 1] - S.first_adam_block[seg];
  {                                                    // ||g||^2 from the block partials: strided loads + a fixed-order tree (deterministic)
    const double* __restrict__ part = S.part[seg];
    double t = 0.0;
    for (int i = threadIdx.x; i < S.nparts[seg]; i += ADAM_T) t += part[i];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int st = ADAM_T / 2; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(sh[0]);
    const float c = max_norm / (norm + 1e-6f);
    s_clip = c < 1.0f ? c : 1.0f;
    const float st = S.step[seg][0];                  // already advanced by k_sqnorm_partial
    s_bc1 = 1.0f - powf(b1, st);
    s_bc2s = sqrtf(1.0f - powf(b2, st));
    s_lr = S.lr[seg][0];
    if (b == 0 && S.norm_out[seg]) S.norm_out[seg][0] = norm;
  }
  __syncthreads();
  float* __restrict__ p = S.p[seg];
  float* __restrict__ g = S.g[seg];
  float* __restrict__ m = S.m[seg];
  float* __restrict__ v = S.v[seg];
  const size_t n = (size_t)S.n[seg];
  const float clipc = s_clip, step_size = s_lr / s_bc1, bc2s = s_bc2s;
  const float decay = 1.0f - s_lr * weight_decay;              // torch.optim.AdamW: param.mul_(1 - lr * weight_decay) before the Adam update
  auto one = [&](float& pi, float& gi_, float& mi_, float& vi_) {
    const float gi = gi_ * clipc;
    gi_ = zero_grad ? 0.0f : gi;                      // clip_grad_norm_ scales .grad in place; or the next step's zero_grad() right here
    const float mi = mi_ + (gi - mi_) * (1.0f - b1);
    const float vi = vi_ * b2 + (1.0f - b2) * gi * gi;
    mi_ = mi; vi_ = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi = pi * decay - step_size * (mi / denom);
  };
  for (size_t i = ((size_t)b * ADAM_T + threadIdx.x) * 4; i < n; i += (size_t)nb * ADAM_T * 4) {
    if (i + 4 <= n) {
      AdamF4 qp = *reinterpret_cast<const AdamF4*>(p + i), qg = *reinterpret_cast<const AdamF4*>(g + i);
      AdamF4 qm = *reinterpret_cast<const AdamF4*>(m + i), qv = *reinterpret_cast<const AdamF4*>(v + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) one(qp.v[u], qg.v[u], qm.v[u], qv.v[u]);
      *reinterpret_cast<AdamF4*>(p + i) = qp; *reinterpret_cast<AdamF4*>(g + i) = qg;
      *reinterpret_cast<AdamF4*>(m + i) = qm; *reinterpret_cast<AdamF4*>(v + i) = qv;
    } else {
      for (size_t j = i; j < n; ++j) one(p[j], g[j], m[j], v[j]);
    }
  }
}

// ---- rollout-side fusions (mh_ppo.py:270-342) ------------------------------------------------------
#include "pbhc_math.h"
__global__ __launch_bounds__(256) void k_policy_sample(const float* __restrict__ mu, const float* __restrict__ stdp, const float* __restrict__ value, int N, int A,
                                                       int R, uint64_t seed, const double* __restrict__ counter, float* __restrict__ actions,
                                                       float* __restrict__ action_mean, float* __restrict__ action_sigma, float* __restrict__ logp,
                                                       float* __restrict__ values_out) {
  const int lane = threadIdx.x & 31, row = blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool valid = row < N;
  const uint32_t ctr = (uint32_t)counter[0];
  float lp = 0.0f;
  if (valid && lane < A) {
    const size_t i = (size_t)row * A + lane;
    const float m = mu[i], sg = stdp[lane];
    uint32_t o[4];
    pbhc::philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)row, ctr, 0x5A4Du, (uint32_t)lane, o);
    const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);           // Box-Muller
    const float a = m + sg * z;
    actions[i] = a; action_mean[i] = m; action_sigma[i] = sg;
    const float d = a - m;
    lp = -(d * d) / (2.0f * sg * sg) - logf(sg) - 0.9189385332046727f;
  }
  lp = gsum32(lp);
  if (valid && lane == 0) logp[row] = lp;
  if (value && valid && lane < R) values_out[(size_t)row * R + lane] = value[(size_t)row * R + lane];
}

__global__ __launch_bounds__(256) void k_rollout_post(const float* __restrict__ rew, const float* __restrict__ values, const int64_t* __restrict__ reset_buf,
                                                      const uint8_t* __restrict__ time_outs, int N, int R, float gamma, float* __restrict__ rewards_out,
                                                      uint8_t* __restrict__ dones_out, float* __restrict__ cur_rew, float* __restrict__ cur_len,
                                                      double* __restrict__ ep_stats, uint8_t* __restrict__ time_outs_out) {
  __shared__ double sh[3][8];
  const int lane = threadIdx.x & 31, lr = threadIdx.x >> 5, row = blockIdx.x * 8 + lr;
  const bool valid = row < N;
  float r = 0.0f;
  const float to = valid ? (time_outs[row] ? 1.0f : 0.0f) : 0.0f;
  if (valid && lane < R) {
    const size_t i = (size_t)row * R + lane;
    r = rew[i];
    rewards_out[i] = values ? r + gamma * values[i] * to : r;      // values == NULL: the caller adds the time-out bootstrap later (batched critic)
  }
  const float rsum = gsum32(r);
  double a = 0.0, b = 0.0, c = 0.0;
  if (valid && lane == 0) {
    const bool done = reset_buf[row] > 0;
    dones_out[row] = done ? 1 : 0;
    if (time_outs_out) time_outs_out[row] = time_outs[row];
    const float cr = cur_rew[row] + rsum, cl = cur_len[row] + 1.0f;
    if (done) { a = (double)cr; b = (double)cl; c = 1.0; }
    cur_rew[row] = done ? 0.0f : cr;
    cur_len[row] = done ? 0.0f : cl;
  }
  if (lane == 0) { sh[0][lr] = a; sh[1][lr] = b; sh[2][lr] = c; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int k = 0; k < 8; ++k) t += sh[threadIdx.x][k];
    if (t != 0.0) atomicAdd(&ep_stats[threadIdx.x], t);        // logging statistics only
  }
}

// ---- backward of a stack's narrow OUTPUT layer in one pass --------------------------------------------------------------------------------
// y = h W^T + b with A = out_features <= 32 (23 actions / 21 value heads) and K = in_features in {64, 128, 192, 256}: autograd runs the weight
// gradient dy^T h (a [A, K] result over a 24 576-long reduction: 20 us of library split-K for 0.14 GFLOP), the bias gradient, the input gradient
// dy W and the activation backward of the layer below as four launches.  Here a workgroup streams a run of rows once: thread (rl, c) owns column
// c of the K inputs for the rows rl, rl + RPI, ... of the run, keeps W[:, c] and its dW[:, c] partial in registers (padded to 32 with zeros: no
// branches), reads the row's dy from an LDS tile (broadcast reads), and writes dh = (dy W) * act'(saved) plus the partial column sums of dh (the
// bias gradient of the layer below), dW and db — one partial row per (workgroup, rl), finished by pbhc_colsum_final like every other column sum.
#define OUTB_T 256
#define OUTB_HR 16                   // rows a thread requests together (h / saved values in flight per thread)
#define OUTB_CHUNK 128               // rows of dy staged in LDS at once (a workgroup's whole run at 24 576 rows)
// AP: A rounded up to a multiple of 8 — the register arrays and the FMA loops are this long (zeros beyond A).  KT: K as a compile-time constant
// (128, 256) or 0 for a run-time K: with it the row-lane count RPI = 256 / K is a constant, every one of a thread's OUTB_HR request slots maps to
// a row of the tile (run-time K: a tile is 16 rows and at K = 128 half the slots stay empty), and the guards fold away.
template <int AP, int KT>
__global__ __launch_bounds__(OUTB_T) void k_out_layer_bwd(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ saved,
                                                          const float* __restrict__ w, int M, int A, int Krt, int act, int rows_per_block,
                                                          float* __restrict__ dh, float* __restrict__ part_dw, float* __restrict__ part_db,
                                                          float* __restrict__ part_cs) {
  __shared__ __attribute__((aligned(16))) float sdy[OUTB_CHUNK][AP];
  const int K = KT > 0 ? KT : Krt;
  const int RPI = OUTB_T / K;                                                  // K divides 256 or K = 192 (then 64 threads idle)
  const int TR = KT > 0 ? OUTB_HR * (OUTB_T / (KT > 0 ? KT : 1)) : OUTB_HR;    // rows per tile (a divisor of OUTB_CHUNK)
  const int c = threadIdx.x % K, rl = threadIdx.x / K;
  const bool live = rl < RPI;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float wc[AP], acc[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) { wc[a] = (live && a < A) ? w[(size_t)a * K + c] : 0.0f; acc[a] = 0.0f; }
  float cs = 0.0f, dbacc = 0.0f;
  for (int rc = r0; rc < r1; rc += OUTB_CHUNK) {
    const int re = min(r1, rc + OUTB_CHUNK);
    __syncthreads();
    for (int i = threadIdx.x; i < OUTB_CHUNK * AP; i += OUTB_T) {
      const int rr = i / AP, a = i - rr * AP;
      sdy[rr][a] = (a < A && rc + rr < re) ? dy[(size_t)(rc + rr) * A + a] : 0.0f;
    }
    __syncthreads();
    for (int rt = rc; rt < re; rt += TR) {
      // this thread's rows of the tile, ALL requested before the first is used (a load -> fma -> store chain per row leaves one load in flight
      // per wave: the first version of this kernel ran at 0.8 TB/s)
      float hv[OUTB_HR], sv[OUTB_HR];
#pragma unroll
      for (int u = 0; u < OUTB_HR; ++u) {
        const int rr = rl + u * RPI;
        const bool ok = live && rr < TR && rt + rr < re;
        const size_t o = (size_t)(rt + (ok ? rr : 0)) * K + c;
        hv[u] = ok ? h[o] : 0.0f;
        sv[u] = saved ? (ok ? saved[o] : 0.0f) : hv[u];
      }
#pragma unroll
      for (int u = 0; u < OUTB_HR; ++u) {
        const int rr = rl + u * RPI;
        if (live && rr < TR && rt + rr < re) {
          const float* dr = sdy[rt - rc + rr];
          float d[AP];
#pragma unroll
          for (int q = 0; q < AP / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(dr + 4 * q);
            d[4 * q] = t.x; d[4 * q + 1] = t.y; d[4 * q + 2] = t.z; d[4 * q + 3] = t.w;
          }
          float g = 0.0f;
#pragma unroll
          for (int a = 0; a < AP; ++a) { g += d[a] * wc[a]; acc[a] += d[a] * hv[u]; }
          const float s_ = sv[u];
          float gr = 1.0f;
          if (act == 1) gr = s_ > 0.0f ? 1.0f : s_ + 1.0f;                   // ELU' from the output, SiLU' from the pre-activation (as k_act_bwd_bias)
          else if (act == 2) { const float sg = ACT_RCP(1.0f + ACT_EXP(-s_)); gr = sg * (1.0f + s_ * (1.0f - sg)); }
          else if (act == 3) gr = s_ > 0.0f ? 1.0f : 0.0f;
          g *= gr;
          dh[(size_t)(rt + rr) * K + c] = g;
          cs += g;
          if (c < A) dbacc += dr[c < AP ? c : 0];
        }
      }
    }
  }
  if constexpr (KT == 128) {
    // the two row lanes of a column meet in LDS (fixed order: lane 0 + lane 1): ONE partial row per workgroup — half the partial traffic
    // (24 576 rows, 23 x 128 weights: 6 MB written here and read by pbhc_colsum_final instead of 12)
    __shared__ float sred[(AP + 2) * 128];
    if (rl == 1) {
#pragma unroll
      for (int a = 0; a < AP; ++a) sred[a * 128 + c] = acc[a];
      sred[AP * 128 + c] = cs;
      sred[(AP + 1) * 128 + c] = dbacc;
    }
    __syncthreads();
    if (rl == 0) {
      const size_t pb = blockIdx.x;
#pragma unroll
      for (int a = 0; a < AP; ++a)
        if (a < A) part_dw[pb * (size_t)(A * K) + (size_t)a * K + c] = acc[a] + sred[a * 128 + c];
      part_cs[pb * K + c] = cs + sred[AP * 128 + c];
      if (c < A) part_db[pb * A + c] = dbacc + sred[(AP + 1) * 128 + c];
    }
  } else if (live) {
    const size_t pb = (size_t)blockIdx.x * RPI + rl;                          // this thread group's partial row
#pragma unroll
    for (int a = 0; a < AP; ++a)
      if (a < A) part_dw[pb * (size_t)(A * K) + (size_t)a * K + c] = acc[a];
    part_cs[pb * K + c] = cs;
    if (c < A) part_db[pb * A + c] = dbacc;
  }
}

// The same pass for K = 128 inputs and a derivative taken from the layer's OUTPUT (ELU / ReLU stacks: the 23-action and 20-value-head tails of
// the update) on the matrix cores.  A workgroup takes 32-row tiles; wave w owns input columns [32 w, 32 w + 32).  Per tile, per wave, two
// 32 x 32 products of depth 32 (`v_mfma_f32_32x32x2_f32`, 16 instructions each):
//   dh[row, col]  = sum_n  dy[row, n] W[n, col]        A = the dy tile (LDS image [32][36], zero-padded to 32 outputs), B = W in registers
//   dW[n, col]   += sum_row dy[row, n] h[row, col]     A = the same image read transposed, B = the tile of h in registers
// and the second product's B fragment is, register for register, the accumulator layout of the first (element e of lane (r, kk) <-> row
// 8 (e / 4) + 4 kk + e % 4, column r): ONE set of 16 loads of h per lane serves as the matrix operand and as act'(h) for the elementwise
// factor.  Nothing but the dy tile goes through LDS.  One partial row per workgroup (dW, db, column sums of dh), finished by pbhc_colsum_final.
typedef float outb_f32x16 __attribute__((ext_vector_type(16)));
typedef float outb_f32x4 __attribute__((ext_vector_type(4)));
#define OUTM_DS 36                   // row stride of the dy image (floats)
// KT = 128 / 256 inputs: KT / 32 waves per workgroup, one 32-column block each
template <bool SAVED, int KT>
__global__ __launch_bounds__(2 * KT) void k_out_bwd_mfma(const float* __restrict__ dy, const float* __restrict__ h, const float* __restrict__ saved,
                                                      const float* __restrict__ w, int M, int A, int act, int ntiles, float* __restrict__ dh,
                                                      float* __restrict__ part_dw, float* __restrict__ part_db, float* __restrict__ part_cs) {
  // SAVED: the derivative's argument is a second tensor (SiLU: the pre-activation) — 16 more loads per lane in the same layout
  __shared__ __attribute__((aligned(16))) float dyimg[32 * OUTM_DS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, kk = lane >> 5;
  const int col = 32 * wave + r;
  constexpr int NT = 2 * KT;                                               // threads per workgroup
  for (int i = tid; i < 32 * OUTM_DS; i += NT) dyimg[i] = 0.0f;          // the padding (outputs A..35) stays zero
  float wb[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = 8 * (e >> 2) + 4 * kk + (e & 3);
    wb[e] = n < A ? w[(size_t)n * KT + col] : 0.0f;
  }
  outb_f32x16 accw;
#pragma unroll
  for (int e = 0; e < 16; ++e) accw[e] = 0.0f;
  float cs = 0.0f, dbacc = 0.0f;
  const int tile_elems = 32 * A;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int row0 = t * 32;
    float hv[16], sv[SAVED ? 16 : 1];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + 8 * (e >> 2) + 4 * kk + (e & 3);
      hv[e] = row < M ? h[(size_t)row * KT + col] : 0.0f;
      if (SAVED) sv[e] = row < M ? saved[(size_t)row * KT + col] : 0.0f;
    }
    __syncthreads();                                                     // the previous tile's image has been read (first tile: the zero fill is complete)
    {
      const float* src = dy + (size_t)row0 * A;
      const int lim = min(tile_elems, (M - row0) * A);
      for (int i = tid; i < tile_elems; i += NT) {
        const int rr = i / A, n = i - rr * A;
        dyimg[rr * OUTM_DS + n] = i < lim ? src[i] : 0.0f;
      }
    }
    __syncthreads();
    outb_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) {
      const outb_f32x4 a = *reinterpret_cast<const outb_f32x4*>(&dyimg[r * OUTM_DS + 8 * k8 + 4 * kk]);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_], wb[4 * k8 + s_], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float a2 = dyimg[(8 * (e >> 2) + 4 * kk + (e & 3)) * OUTM_DS + r];
      accw = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, hv[e], accw, 0, 0, 0);
    }
    if (wave == 0 && kk == 0) {                                          // db: column sums of the dy tile (rows beyond M are zeros)
      float t_ = 0.0f;
#pragma unroll 8
      for (int rr = 0; rr < 32; ++rr) t_ += dyimg[rr * OUTM_DS + r];
      dbacc += t_;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + 8 * (e >> 2) + 4 * kk + (e & 3);
      const float hh = SAVED ? sv[e] : hv[e];
      float gr = 1.0f;
      if (act == 1) gr = hh > 0.0f ? 1.0f : hh + 1.0f;                    // ELU' / ReLU' from the output, SiLU' from the pre-activation (as k_act_bwd_bias)
      else if (act == 2) { const float sg = ACT_RCP(1.0f + ACT_EXP(-hh)); gr = sg * (1.0f + hh * (1.0f - sg)); }
      else if (act == 3) gr = hh > 0.0f ? 1.0f : 0.0f;
      const float g = acc[e] * gr;
      if (row < M) { dh[(size_t)row * KT + col] = g; cs += g; }
    }
  }
  const size_t pb = blockIdx.x;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = 8 * (e >> 2) + 4 * kk + (e & 3);
    if (n < A) part_dw[pb * (size_t)(A * KT) + (size_t)n * KT + col] = accw[e];
  }
  cs += __shfl_xor(cs, 32, 64);
  if (kk == 0) part_cs[pb * KT + col] = cs;
  if (wave == 0 && kk == 0 && r < A) part_db[pb * A + r] = dbacc;
}

// ---- the minibatch shuffle: every key's rows gathered by one permutation, one launch ------------------------------------------------------------
struct GatherJobs { PbhcGatherJob job[PBHC_MAX_GATHER_JOBS]; int vec[PBHC_MAX_GATHER_JOBS]; int n; };
template <int V>
__device__ __forceinline__ void gather_row(const float* __restrict__ src, float* __restrict__ dst, int width, int lane) {
  typedef float vt __attribute__((ext_vector_type(V)));
  const int nv = width / V;
  for (int c = lane; c < nv; c += 64) reinterpret_cast<vt*>(dst)[c] = reinterpret_cast<const vt*>(src)[c];
}
// a wave per destination row (4 rows per workgroup), all jobs of the row: 16- / 8- / 4-byte pieces as the job's widths and addresses allow
__global__ __launch_bounds__(256) void k_gather_rows(GatherJobs J, const int64_t* __restrict__ index, int nrows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int64_t s = index[row];
  for (int j = 0; j < J.n; ++j) {
    const PbhcGatherJob job = J.job[j];
    const float* src = job.src + (size_t)s * job.src_pitch;
    float* dst = job.dst + (size_t)row * job.width;
    if (J.vec[j] == 4) gather_row<4>(src, dst, job.width, lane);
    else if (J.vec[j] == 2) gather_row<2>(src, dst, job.width, lane);
    else for (int c = lane; c < job.width; c += 64) dst[c] = src[c];
  }
}

template <int KT>
static void out_bwd_launch(int grid, hipStream_t st, const float* dy, const float* h, const float* saved, const float* w, int M, int A, int K, int act, int rpb,
                           float* dh, float* part_dw, float* part_db, float* part_cs) {
  if (A <= 8) hipLaunchKernelGGL((k_out_layer_bwd<8, KT>), dim3(grid), dim3(OUTB_T), 0, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  else if (A <= 16) hipLaunchKernelGGL((k_out_layer_bwd<16, KT>), dim3(grid), dim3(OUTB_T), 0, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  else if (A <= 24) hipLaunchKernelGGL((k_out_layer_bwd<24, KT>), dim3(grid), dim3(OUTB_T), 0, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  else hipLaunchKernelGGL((k_out_layer_bwd<32, KT>), dim3(grid), dim3(OUTB_T), 0, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
}

extern "C" {

int pbhc_policy_sample(const float* mu, const float* std, const float* value, int N, int A, int R, uint64_t seed, const double* counter,
                       float* actions, float* action_mean, float* action_sigma, float* logp, float* values_out, void* stream) {
  ARG_CHECK(mu && std && (!value || values_out) && counter && actions && action_mean && action_sigma && logp && N >= 1 && A >= 1 && A <= 32 && R >= 1 && R <= 32);
  hipLaunchKernelGGL(k_policy_sample, dim3((N + 7) / 8), dim3(256), 0, (hipStream_t)stream, mu, std, value, N, A, R, seed, counter, actions, action_mean,
                     action_sigma, logp, values_out);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_rollout_post2(const float* rew, const float* values, const int64_t* reset_buf, const uint8_t* time_outs, int N, int R, float gamma,
                       float* rewards_out, uint8_t* dones_out, float* cur_reward_sum, float* cur_episode_length, double* ep_stats, uint8_t* time_outs_out,
                       void* stream) {
  ARG_CHECK(rew && reset_buf && time_outs && rewards_out && dones_out && cur_reward_sum && cur_episode_length && ep_stats && N >= 1 && R >= 1 && R <= 32);
  hipLaunchKernelGGL(k_rollout_post, dim3((N + 7) / 8), dim3(256), 0, (hipStream_t)stream, rew, values, reset_buf, time_outs, N, R, gamma, rewards_out, dones_out,
                     cur_reward_sum, cur_episode_length, ep_stats, time_outs_out);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_rollout_post(const float* rew, const float* values, const int64_t* reset_buf, const uint8_t* time_outs, int N, int R, float gamma,
                      float* rewards_out, uint8_t* dones_out, float* cur_reward_sum, float* cur_episode_length, double* ep_stats, void* stream) {
  ARG_CHECK(values);
  return pbhc_rollout_post2(rew, values, reset_buf, time_outs, N, R, gamma, rewards_out, dones_out, cur_reward_sum, cur_episode_length, ep_stats, nullptr, stream);
}

int pbhc_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* old_logp, const float* old_mu,
                  const float* old_sigma, const float* adv, const float* returns, const float* old_values, int B, int A, int R, float clip,
                  float value_coef, float entropy_coef, int use_clipped_value_loss, float desired_kl, int adapt_lr, float* grad_mu, float* grad_value,
                  float* grad_std, float* scalars, float* scalars_acc, float* lr, float* scratch, void* stream) {
  ARG_CHECK(mu && std && value && actions && old_logp && old_mu && old_sigma && adv && returns && old_values);
  ARG_CHECK(grad_mu && grad_value && grad_std && scalars && lr && scratch);
  ARG_CHECK(B >= 1 && A >= 1 && A <= 32 && R >= 1 && R <= 32);
  hipStream_t st = (hipStream_t)stream;
  int nb = (B + LOSS_ROWS - 1) / LOSS_ROWS;
  if (nb > 512) nb = 512;                         // grid-stride over row tiles: the 1-block second stage sums <= 512 partials per column
  float* partial = scratch;                       // [nb][LOSS_NP]
  float* gstd_part = scratch + (size_t)nb * LOSS_NP;   // [nb][32]
  hipLaunchKernelGGL(k_ppo_loss, dim3(nb), dim3(256), 0, st, mu, std, value, actions, old_logp, old_mu, old_sigma, adv, returns, old_values, B, A, R,
                     clip, value_coef, use_clipped_value_loss, (adapt_lr >> 1) & 1, grad_mu, grad_value, partial, gstd_part);
  hipLaunchKernelGGL(k_ppo_reduce, dim3(1), dim3(RED_T), 0, st, partial, gstd_part, std, nb, B, A, entropy_coef, desired_kl, adapt_lr & 1, grad_std, scalars, lr, scalars_acc);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

static int act_bwd_launch(const float* dy, const float* saved, int B, int n, int act, float* dz, float* scratch, hipStream_t st) {
  // row blocks: enough workgroups to fill 256 CUs (x column blocks when n > 256), at least 32 rows each
  const int colblocks = (n + ACT_T - 1) / ACT_T;
  int nb = (B + 31) / 32;
  const int want = colblocks >= 3 ? 256 : PBHC_ACT_MAX_BLOCKS;
  if (nb > want) nb = want;
  const int rpb = (B + nb - 1) / nb;
  nb = (B + rpb - 1) / rpb;
  hipLaunchKernelGGL(k_act_bwd_bias, dim3(nb, colblocks), dim3(ACT_T), 0, st, dy, saved, B, n, act, dz, scratch, rpb);
  return nb;
}

int pbhc_act_bwd_bias(const float* dy, const float* saved, int B, int n, int act, float* dz, float* grad_bias, float* scratch, void* stream) {
  ARG_CHECK(dy && dz && grad_bias && scratch && B >= 1 && n >= 1 && act >= 0 && act <= 3 && (act == 0 || saved));
  hipStream_t st = (hipStream_t)stream;
  const int nb = act_bwd_launch(dy, saved, B, n, act, dz, scratch, st);
  hipLaunchKernelGGL(k_colsum_final, dim3((n + 31) / 32), dim3(ACT_T), 0, st, scratch, nb, n, grad_bias);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_act_bwd_partials(const float* dy, const float* saved, int B, int n, int act, float* dz, float* scratch, int* num_row_blocks, void* stream) {
  ARG_CHECK(dy && dz && scratch && num_row_blocks && B >= 1 && n >= 1 && act >= 0 && act <= 3 && (act == 0 || saved));
  *num_row_blocks = act_bwd_launch(dy, saved, B, n, act, dz, scratch, (hipStream_t)stream);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

static int g_out_bwd_mfma = 1, g_out_bwd_tiles_per_wg = 0;
int pbhc_gather_rows(const PbhcGatherJob* jobs, int num_jobs, const int64_t* index, int nrows, void* stream) {
  ARG_CHECK(jobs && index && num_jobs >= 1 && num_jobs <= PBHC_MAX_GATHER_JOBS && nrows >= 1);
  GatherJobs J;
  J.n = num_jobs;
  for (int j = 0; j < num_jobs; ++j) {
    ARG_CHECK(jobs[j].src && jobs[j].dst && jobs[j].width >= 1 && jobs[j].src_pitch >= jobs[j].width);
    J.job[j] = jobs[j];
    const uintptr_t a = (uintptr_t)jobs[j].src | (uintptr_t)jobs[j].dst;
    const int wp = jobs[j].width | jobs[j].src_pitch;
    J.vec[j] = ((wp & 3) == 0 && (a & 15) == 0) ? 4 : ((wp & 1) == 0 && (a & 7) == 0) ? 2 : 1;
  }
  hipLaunchKernelGGL(k_gather_rows, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, J, index, nrows);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

void pbhc_debug_out_bwd_variant(int mfma) { g_out_bwd_mfma = mfma & 0xff; g_out_bwd_tiles_per_wg = (mfma >> 8) & 0xff; }       // test / measurement aid: 0 = the streaming VALU form for every shape

int pbhc_linear_out_bwd(const float* dy, const float* h, const float* saved, const float* w, int M, int A, int K, int act, float* dh, float* part_dw,
                        float* part_db, float* part_cs, int* num_row_blocks, void* stream) {
  ARG_CHECK(dy && h && w && dh && part_dw && part_db && part_cs && num_row_blocks && M >= 1 && A >= 1 && A <= 32 && act >= 0 && act <= 3);
  ARG_CHECK(K == 64 || K == 128 || K == 192 || K == 256);
  hipStream_t st = (hipStream_t)stream;
  if ((K == 128 || K == 256) && (saved || act != 2) && g_out_bwd_mfma) {
    const int ntiles = (M + 31) / 32;
    const int tpw = g_out_bwd_tiles_per_wg > 0 ? g_out_bwd_tiles_per_wg : 1;        // tiles per workgroup (measurement aid; 1: every tile its own workgroup)
    int grid = (ntiles + tpw - 1) / tpw;
    if (grid > PBHC_ACT_MAX_BLOCKS) grid = PBHC_ACT_MAX_BLOCKS;
    *num_row_blocks = grid;
    if (K == 128) {
      if (saved) hipLaunchKernelGGL((k_out_bwd_mfma<true, 128>), dim3(grid), dim3(256), 0, st, dy, h, saved, w, M, A, act, ntiles, dh, part_dw, part_db, part_cs);
      else hipLaunchKernelGGL((k_out_bwd_mfma<false, 128>), dim3(grid), dim3(256), 0, st, dy, h, saved, w, M, A, act, ntiles, dh, part_dw, part_db, part_cs);
    } else {
      if (saved) hipLaunchKernelGGL((k_out_bwd_mfma<true, 256>), dim3(grid), dim3(512), 0, st, dy, h, saved, w, M, A, act, ntiles, dh, part_dw, part_db, part_cs);
      else hipLaunchKernelGGL((k_out_bwd_mfma<false, 256>), dim3(grid), dim3(512), 0, st, dy, h, saved, w, M, A, act, ntiles, dh, part_dw, part_db, part_cs);
    }
    HIP_CHECK(hipGetLastError());
    return PBHC_OK;
  }
  const int RPI = OUTB_T / K;
  int grid = (M + OUTB_HR - 1) / OUTB_HR;
  const int cap = PBHC_ACT_MAX_BLOCKS / RPI < 512 ? PBHC_ACT_MAX_BLOCKS / RPI : 512;     // two workgroups per CU; partial rows = grid x RPI <= the scratch cap
  if (grid > cap) grid = cap;
  int rpb = (M + grid - 1) / grid;
  rpb = (rpb + OUTB_HR - 1) / OUTB_HR * OUTB_HR;
  grid = (M + rpb - 1) / rpb;
  *num_row_blocks = K == 128 ? grid : grid * RPI;          // (K = 128: the two row lanes are summed inside the workgroup)
  if (K == 128) out_bwd_launch<128>(grid, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  else if (K == 256) out_bwd_launch<256>(grid, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  else out_bwd_launch<0>(grid, st, dy, h, saved, w, M, A, K, act, rpb, dh, part_dw, part_db, part_cs);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_colsum_final(const PbhcColsumJob* jobs, int num_jobs, void* stream) {
  ARG_CHECK(jobs && num_jobs >= 1 && num_jobs <= PBHC_MAX_COLSUM_JOBS);
  ColsumJobs J;
  int blocks = 0;
  for (int j = 0; j < PBHC_MAX_COLSUM_JOBS; ++j) {
    if (j < num_jobs) {
      ARG_CHECK(jobs[j].part && jobs[j].out && jobs[j].num_row_blocks >= 1 && jobs[j].n >= 1);
      J.job[j] = jobs[j];
      blocks += colsum_blocks(jobs[j].num_row_blocks, jobs[j].n);
    } else {
      J.job[j] = PbhcColsumJob{nullptr, nullptr, 0, 0};
    }
  }
  int acc = 0;
  for (int j = 0; j < PBHC_MAX_COLSUM_JOBS; ++j) { J.first_block[j] = acc; acc += colsum_blocks(J.job[j].num_row_blocks, J.job[j].n); }
  hipLaunchKernelGGL(k_colsum_final_multi, dim3(blocks), dim3(ACT_T), 0, (hipStream_t)stream, J, num_jobs);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

// The adaptive-KL learning-rate rule (mh_ppo.py:455-466) on a KL mean that is ready on the device — the data-parallel update applies it to
// the mean over ALL ranks, which exists only after the gradient bucket's all-reduce; as eight tiny torch launches it was 40 us of every
// optimiser step.  Same arithmetic as the rule inside k_ppo_reduce.
__global__ void k_kl_lr_rule(float* __restrict__ lr, int n, const float* __restrict__ kl_mean_p, float desired_kl) {
  if (threadIdx.x < n) {
    const float kl_mean = kl_mean_p[0];
    float l = lr[threadIdx.x];                                   // each rate from its own value (mh_ppo.py:457-461)
    if (kl_mean > desired_kl * 2.0f) l = fmaxf(1e-5f, l / 1.5f);
    else if (kl_mean < desired_kl / 2.0f && kl_mean > 0.0f) l = fminf(1e-2f, l * 1.5f);
    lr[threadIdx.x] = l;
  }
}
int pbhc_kl_lr_rule(float* lr, int n, const float* kl_mean, float desired_kl, void* stream) {
  ARG_CHECK(lr && kl_mean && n >= 1 && n <= 64);
  hipLaunchKernelGGL(k_kl_lr_rule, dim3(1), dim3(64), 0, (hipStream_t)stream, lr, n, kl_mean, desired_kl);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_ppo_loss_scratch_floats(int B) { return ((B + LOSS_ROWS - 1) / LOSS_ROWS) * (LOSS_NP + 32); }

static void adam_fill(AdamSegs& S, int k, float* param, float* grad, float* m, float* v, int n, const float* lr, float* step, double* scratch, float* norm_out) {
  S.p[k] = param; S.g[k] = grad; S.m[k] = m; S.v[k] = v; S.n[k] = n; S.lr[k] = lr; S.step[k] = step; S.part[k] = scratch; S.norm_out[k] = norm_out;
  int nb = (n + ADAM_T * 8 - 1) / (ADAM_T * 8);
  if (nb > 512) nb = 512;
  int nb2 = (n + ADAM_T * 4 - 1) / (ADAM_T * 4);
  if (nb2 > 1024) nb2 = 1024;
  S.nparts[k] = nb;
  S.first_norm_block[k + 1] = S.first_norm_block[k] + nb;
  S.first_adam_block[k + 1] = S.first_adam_block[k] + nb2;
}

int pbhc_adam_clip(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int n, const float* lr, float* step, float max_norm, float beta1,
                   float beta2, float eps, float weight_decay, double* scratch, float* norm_out, void* stream) {
  ARG_CHECK(param && grad && exp_avg && exp_avg_sq && lr && step && scratch && n >= 1);
  hipStream_t st = (hipStream_t)stream;
  AdamSegs S = {};
  S.num = 1;
  adam_fill(S, 0, param, grad, exp_avg, exp_avg_sq, n, lr, step, scratch, norm_out);
  S.first_norm_block[2] = S.first_norm_block[1]; S.first_adam_block[2] = S.first_adam_block[1];
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(S.first_norm_block[1]), dim3(ADAM_T), 0, st, S);
  hipLaunchKernelGGL(k_adam_clip, dim3(S.first_adam_block[1]), dim3(ADAM_T), 0, st, S, max_norm, beta1, beta2, eps, weight_decay, 0);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

int pbhc_adam_clip2(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int n0, int n1, const float* lr, float* step, float max_norm, float beta1,
                    float beta2, float eps, float weight_decay, int zero_grad, double* scratch, float* norm_out, void* stream) {
  ARG_CHECK(param && grad && exp_avg && exp_avg_sq && lr && step && scratch && n0 >= 1 && n1 >= 1);
  hipStream_t st = (hipStream_t)stream;
  AdamSegs S = {};
  S.num = 2;
  adam_fill(S, 0, param, grad, exp_avg, exp_avg_sq, n0, lr, step, scratch, norm_out);
  adam_fill(S, 1, param + n0, grad + n0, exp_avg + n0, exp_avg_sq + n0, n1, lr + 1, step + 1, scratch + 512, norm_out ? norm_out + 1 : nullptr);
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(S.first_norm_block[2]), dim3(ADAM_T), 0, st, S);
  hipLaunchKernelGGL(k_adam_clip, dim3(S.first_adam_block[2]), dim3(ADAM_T), 0, st, S, max_norm, beta1, beta2, eps, weight_decay, zero_grad);
  HIP_CHECK(hipGetLastError());
  return PBHC_OK;
}

}  // extern "C"
