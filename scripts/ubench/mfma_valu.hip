// ubench: do fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 VALU of a co-resident wave overlap on one SIMD?
// block = 512 threads = 8 waves = 2 per SIMD.  mode 0: all 8 waves MFMA.  mode 1: waves 0-3 MFMA, 4-7 idle exit.
// mode 2: waves 0-3 MFMA, waves 4-7 VALU fma chain.  mode 3: waves 4-7 VALU only (0-3 exit).  mode 4: waves 4-7 LDS+global store loop
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void __launch_bounds__(512) k(int mode, int iters, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool is_mfma = wave < 4;
  unsigned long long t0 = __builtin_readcyclecounter();
  float res = 0.f;
  if (is_mfma && mode != 3) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    res = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (!is_mfma && (mode == 0)) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    res = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (!is_mfma && (mode == 2 || mode == 3)) {
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters * 16; ++i) {   // 8 independent fma per trip
      v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
      v4 = fmaf(v4, c, d); v5 = fmaf(v5, c, d); v6 = fmaf(v6, c, d); v7 = fmaf(v7, c, d);
    }
    res = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  unsigned long long h[8];
  const int iters = 4096;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc); hipDeviceSynchronize(); }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: wave cycles:", mode);
    for (int w = 0; w < 8; ++w) printf(" %llu", h[w]);
    printf("   (mfma-only ideal per wave: %d cycles; valu trip = 8 fma)\n", iters * 4 * 64);
  }
  return 0;
}
