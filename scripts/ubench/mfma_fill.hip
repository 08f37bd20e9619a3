// How many independent filler instructions hide behind one v_mfma_f32_32x32x16_bf16 when a SIMD runs ONE wave?
// Kernel: 256 threads (one wave per SIMD), loop of [MFMA on accumulator A or B alternately ; K fillers], fillers =
// fp32 VALU (v_fma; independent or in 1 / 2 / 3 dependent chains), v_exp_f32, ds_read_b128 (+ a counted wait), or SALU.  Prints cycles per MFMA for K = 0..10.
// hipcc --offload-arch=gfx950 -O2 mfma_fill.hip -o mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int K, int KIND>
__global__ void __launch_bounds__(256, 1) k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = float(i);
  __syncthreads();
  f32x16 a = {}, b = {};
  u32x4 w = {threadIdx.x, 1u, 2u, 3u}, x = {3u, 2u, 1u, threadIdx.x};
  float f[12];
  for (int i = 0; i < 12; ++i) f[i] = float(threadIdx.x + i);
  u32x4 lr[4] = {};
  f32x2 p[6];
  for (int i = 0; i < 6; ++i) p[i] = f32x2{float(threadIdx.x + i), float(i)};
  const f32x2 kc = {1.0009765625f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u & 1) b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), b, 0, 0, 0);
      else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), a, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if (KIND == 0) {
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q % 12]));
        } else if (KIND == 1) {
          if (q % 2 == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(lr[(q / 2) % 4]) : "v"((threadIdx.x & 63) * 16 + (q / 2) * 1024));
          else asm volatile("s_waitcnt lgkmcnt(3)");
        } else if (KIND == 2) {
          asm volatile("s_nop 0");
        } else if (KIND == 3) {          // ONE dependent chain: every filler needs the previous one's result
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[0]));
        } else if (KIND == 4) {          // two interleaved dependent chains (distance 2)
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q & 1]));
        } else if (KIND == 5) {          // three interleaved dependent chains (distance 3)
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[q % 3]));
        } else if (KIND == 6) {          // transcendental (quarter rate): v_exp_f32, independent
          asm volatile("v_exp_f32 %0, %0" : "+v"(f[q % 12]));
        } else if (KIND == 7) {          // packed fp32 (two values per lane and instruction), independent
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[q % 6]));
        } else if (KIND == 8) {          // packed fp32, two dependent chains
          asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[q & 1]));
        } else if (KIND == 9) {          // packed fp32 with an SGPR-pair constant operand (splat through op_sel_hi)
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[q % 6]) : "s"(kc));
        } else {                         // packed multiply
          asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[q % 6]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + b[i];
  for (int i = 0; i < 12; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += float(lr[i][0]);
  for (int i = 0; i < 6; ++i) s += p[i][0] + p[i][1];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int KIND>
void run(unsigned long long* d_out, float* d_sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
  hipLaunchKernelGGL((k<K, KIND>), dim3(256), dim3(256), 0, 0, d_out, d_sink, iters);
  unsigned long long h = 0;
  hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
  printf("  K=%2d: %.1f cycles (s_memtime ticks) per MFMA\n", K, double(h) / (iters * 8.0));
}
template <int KIND>
void sweep(const char* name, unsigned long long* d_out, float* d_sink) {
  printf("%s fillers behind each MFMA (two accumulators alternating, one wave per SIMD):\n", name);
  run<0, KIND>(d_out, d_sink); run<1, KIND>(d_out, d_sink); run<2, KIND>(d_out, d_sink); run<3, KIND>(d_out, d_sink);
  run<4, KIND>(d_out, d_sink); run<5, KIND>(d_out, d_sink); run<6, KIND>(d_out, d_sink); run<8, KIND>(d_out, d_sink);
  run<10, KIND>(d_out, d_sink);
}
int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 8); hipMalloc(&d_sink, 256 * 256 * 4);
  sweep<0>("v_fma_f32", d_out, d_sink);
  sweep<1>("ds_read_b128 / s_waitcnt", d_out, d_sink);
  sweep<2>("s_nop", d_out, d_sink);
  sweep<3>("v_fma_f32, ONE dependent chain", d_out, d_sink);
  sweep<4>("v_fma_f32, two dependent chains interleaved", d_out, d_sink);
  sweep<5>("v_fma_f32, three dependent chains interleaved", d_out, d_sink);
  sweep<6>("v_exp_f32 (independent)", d_out, d_sink);
  sweep<7>("v_pk_fma_f32 (independent)", d_out, d_sink);
  sweep<8>("v_pk_fma_f32, two dependent chains interleaved", d_out, d_sink);
  sweep<9>("v_pk_fma_f32 with SGPR constant (op_sel_hi splat)", d_out, d_sink);
  sweep<10>("v_pk_mul_f32 (independent)", d_out, d_sink);
  return 0;
}
