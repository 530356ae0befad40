// Sustained fp32-MFMA rate and shader clock on the box: what a GEMM at 100 % MFMA issue would reach (diagnosis; built by tools/mfma_peak_probe.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k_peak(float* out, long long* clk, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
extern "C" int mfma_peak(float* out, long long* clk, int blocks, int iters, int nacc, void* stream) {
  if (nacc == 4) hipLaunchKernelGGL(k_peak<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, clk, iters);
  else hipLaunchKernelGGL(k_peak<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, clk, iters);
  return (int)hipGetLastError();
}
