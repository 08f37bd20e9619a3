// ubench: would TWO waves per SIMD hide the layer kernel's VALU-only phases?  (the structural lever left for
// b3::k_layer.)  Today one wave per SIMD owns 32 tokens (v_mfma_f32_32x32x16_bf16, 512 registers); its LayerNorm / GELU /
// epilogue phases leave the matrix pipe idle.  With 16 tokens per wave (v_mfma_f32_16x16x32_bf16, <= 256 registers) two waves
// share a SIMD and one wave's VALU phase can run under the other's MFMAs - if the hardware interleaves them at full rate.
//
// Every wave runs `iters` x [ P "MFMA blocks" ; V VALU-only instructions ].  An MFMA block = 12 MFMAs on two alternating
// accumulator chains with F independent v_fma fillers behind each (the layer kernel's structure).  Work per SIMD is the same
// in every mode (the 16-token waves do blocks of half the cycles and half the VALU each); the second wave of a SIMD starts
// with its VALU phase, so the phases of the two waves alternate.
//   mode 0: 1 wave / SIMD,  32x32x16            (today)
//   mode 1: 2 waves / SIMD, 16x16x32            (the proposal)
//   mode 2: 2 waves / SIMD, 32x32x16, each wave half the iterations   (does it need the smaller tile at all?)
// Output: cycles per iteration of the slowest wave and the MFMA-busy fraction = MFMA cycles per SIMD / cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// fillers / VALU phase as volatile asm: the compiler would otherwise merge them into v_pk_fma_f32 (which shares the matrix
// pipe) and move them away from their MFMA
__device__ __forceinline__ void vfma(float& x, float c, float d) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d)); }
template <int F>
__device__ __forceinline__ void fillers(float (&v)[8], float c, float d) {
#pragma unroll
  for (int i = 0; i < F; ++i) vfma(v[i], c, d);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int F>
__global__ void __launch_bounds__(MODE == 0 ? 256 : 512) k(int iters, int P, int V, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool second = wave >= 4;                       // the second wave of its SIMD (waves w and w + 4 share SIMD w & 3)
  const u32x4 wa = {threadIdx.x * 3u + 1u, 0x3f803f80u, 0x3f813f82u, 0x3f833f84u};
  const u32x4 xa = {0x3f803f80u, threadIdx.x * 5u + 7u, 0x3f853f86u, 0x3f873f88u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, wa), b = __builtin_bit_cast(bf16x8, xa);
  float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, float(threadIdx.x)};
  const float c = 1.0001f, d = 0.5f;
  f32x16 A0 = {0}, A1 = {0};
  f32x4 B0 = {0}, B1 = {0}, B2 = {0}, B3 = {0};
  const int my_iters = MODE == 2 ? iters / 2 : iters;
  auto valu_phase = [&](int n) {
    for (int i = 0; i < n / 32; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 8; ++u) vfma(v[u], c, d);
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (second) valu_phase(MODE == 1 ? V / 2 : V);       // offset the two waves of a SIMD by one phase
  for (int it = 0; it < my_iters; ++it) {
    for (int p = 0; p < P; ++p) {
      if constexpr (MODE == 1) {
        // 12 MFMAs of 16 cycles on four alternating chains (a 16x16 tile is 4 accumulator registers)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          B0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, B0, 0, 0, 0); fillers<F>(v, c, d);
          B1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, B1, 0, 0, 0); fillers<F>(v, c, d);
          B2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, B2, 0, 0, 0); fillers<F>(v, c, d);
          B3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, B3, 0, 0, 0); fillers<F>(v, c, d);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, A0, 0, 0, 0); fillers<F>(v, c, d);
          A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, A1, 0, 0, 0); fillers<F>(v, c, d);
        }
      }
    }
    valu_phase(MODE == 1 ? V / 2 : V);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float res = A0[0] + A1[1] + B0[0] + B1[1] + B2[2] + B3[3];
  for (int u = 0; u < 8; ++u) res += v[u];
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int F>
void run(int iters, int P, int V, float* out, unsigned long long* cyc) {
  unsigned long long h[8 * 256];
  const int nt = MODE == 0 ? 256 : 512, nw = nt / 64;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE, F>), dim3(256), dim3(nt), 0, 0, iters, P, V, out, cyc);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long worst = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < nw; ++w) worst = h[b * 8 + w] > worst ? h[b * 8 + w] : worst;
  // MFMA cycles per SIMD: iters x P blocks x 12 MFMAs x 32 cycles (mode 1: two waves x 12 x 16; mode 2: two waves x iters/2)
  const double mfma = double(iters) * P * 12 * 32;
  printf("mode %d  F=%d  P=%d  V=%d : %8.0f cycles per iteration, MFMA-busy %.3f\n", MODE, F, P, V, double(worst) / iters, mfma / double(worst));
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 400;
  // the layer kernel per tile: ~83 stages x 8 blocks = 664 MFMA blocks and ~27 K cycles of VALU-only phases (~6.7 K v_* at 4
  // cycles); per "phase pair" here: P blocks then V VALU instructions, same ratio (P = 16 -> V = 160) and a heavier one
  const int cfg[3][2] = {{16, 160}, {16, 320}, {64, 640}};
  for (auto& pv : cfg) {
    run<0, 0>(iters, pv[0], pv[1], out, cyc);
    run<0, 3>(iters, pv[0], pv[1], out, cyc);
    run<0, 5>(iters, pv[0], pv[1], out, cyc);
    run<1, 0>(iters, pv[0], pv[1], out, cyc);
    run<1, 1>(iters, pv[0], pv[1], out, cyc);
    run<1, 2>(iters, pv[0], pv[1], out, cyc);
    run<1, 3>(iters, pv[0], pv[1], out, cyc);
    run<2, 0>(iters, pv[0], pv[1], out, cyc);
    run<2, 3>(iters, pv[0], pv[1], out, cyc);
    run<2, 5>(iters, pv[0], pv[1], out, cyc);
  }
  return 0;
}
