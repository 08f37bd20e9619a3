// What does the WHOLE chip sustain on v_mfma_f32_32x32x16_bf16 when nothing else happens - no memory, no LDS, operands
// and accumulators in registers - and at what package power and shader clock?  The layer kernel's roofline (2500 TFLOP/s
// dense bf16 = 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz) assumes the nominal clock; under the 1400 W package cap the
// part does not hold it while the matrix pipes are busy.  This loop measures that ceiling directly:
//   mfma_chip <waves per SIMD: 1|2> <accumulator chains per wave: 2|4> <seconds> [operand pattern: 0 zeros | 1 random bits]
//             [ds_read_b128 per two MFMAs: 0|1|2] [fp32 VALU fillers per MFMA: 0|2|3|4]
// The last two add what the layer kernel does around its MFMAs - the A fragments come out of LDS (random bits, a different
// KiB every read) and independent v_fma_f32 fill the issue slots behind each MFMA - still without any global memory: how
// much clock do those cost at the cap?
// Persistent grid (one block of 256 x waves-per-SIMD threads per CU); prints the achieved TFLOP/s and the cycles per MFMA
// per SIMD from s_memtime.  scripts/power_calibration.py runs it while sampling amdsmi.
// hipcc --offload-arch=gfx950 -O2 mfma_chip.hip -o mfma_chip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int CHAINS, int LDSR, int VALU>
__global__ void __launch_bounds__(512, 1) k(float* sink, unsigned long long* cyc, int iters, int pattern) {
  __shared__ u32x4 lds[LDSR ? 3072 : 1];                  // 48 KiB: one stage image of the layer kernel (32 KiB of it are read)
  // operands: zeros (pattern 0: the multiplier array does not toggle) or pseudo-random bf16 bit patterns (random sign and mantissa, exponent of
  // 1.0 (pattern 1: what the split pieces of real activations look like to the datapath)
  unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() {          // two bf16 per dword: random sign, exponent of 1.0, random 7-bit mantissa
    s = s * 1664525u + 1013904223u;
    return 0x3F803F80u | ((s >> 8) & 0x807F807Fu);
  };
  // four operand pairs used in rotation, so that with pattern 1 the inputs of the multiplier array change with every MFMA
  u32x4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = pattern ? u32x4{rnd(), rnd(), rnd(), rnd()} : u32x4{0, 0, 0, 0};
    b[i] = pattern ? u32x4{rnd(), rnd(), rnd(), rnd()} : u32x4{0, 0, 0, 0};
  }
  if (LDSR) {
    for (int i = threadIdx.x; i < 3072; i += blockDim.x) lds[i] = pattern ? u32x4{rnd(), rnd(), rnd(), rnd()} : u32x4{0, 0, 0, 0};
    __syncthreads();
  }
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = 1.0f + float(threadIdx.x + i) * 1e-3f;
  const float fc = 0.99993896484375f;
  const unsigned lane = threadIdx.x & 63;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) u32x4*)lds;
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x16{};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      // the fragment read here is consumed two operand rotations later
      // asm, so that the read stays HERE (the compiler would sink it next to its use and expose the LDS latency).  The counted
      // wait only bounds the reads in flight (a wait for the previous read - lgkmcnt(1) - stalls the wave for most of an LDS
      // round trip every two MFMAs: 43 cycles per MFMA instead of 33, r03l); the MFMAs may consume a fragment register whose
      // read is still in flight - stale bits, which is all a power benchmark needs
      if (LDSR && (u % 2 == 0 || LDSR == 2)) {
        const unsigned addr = lds_base + (((unsigned(it) * 16u + u) * 64u + lane) & 2047u) * 16u     /* (a power of two: no integer division in the loop) */;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(3)" : "=v"(a[(u / CHAINS + 2) & 3]) : "v"(addr));
      }
      acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u / CHAINS) & 3]), __builtin_bit_cast(bf16x8, b[(u / CHAINS + u) & 3]), acc[u % CHAINS], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < VALU; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(u * VALU + q) & 7]) : "v"(fc));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) r += acc[c][e];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += f[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;     // wall time of the loop in 10-ns ticks
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1, chains = argc > 2 ? atoi(argv[2]) : 2;
  const double seconds = argc > 3 ? atof(argv[3]) : 3.0;
  const int pattern = argc > 4 ? atoi(argv[4]) : 1;
  const int ldsr = argc > 5 ? atoi(argv[5]) : 0, valu = argc > 6 ? atoi(argv[6]) : 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  const int cus = p.multiProcessorCount, threads = 256 * wps;
  float* sink;
  unsigned long long* cyc;
  if (hipMalloc(&sink, size_t(cus) * threads * 4) != hipSuccess || hipMalloc(&cyc, size_t(cus) * threads / 64 * 8) != hipSuccess) return 1;
  void (*kern)(float*, unsigned long long*, int, int) = nullptr;
  if (chains == 4 && !ldsr && !valu) kern = k<4, 0, 0>;
  else if (chains == 2 && ldsr == 0 && valu == 0) kern = k<2, 0, 0>;
  else if (chains == 2 && ldsr == 1 && valu == 0) kern = k<2, 1, 0>;
  else if (chains == 2 && ldsr == 0 && valu == 2) kern = k<2, 0, 2>;
  else if (chains == 2 && ldsr == 0 && valu == 4) kern = k<2, 0, 4>;
  else if (chains == 2 && ldsr == 1 && valu == 2) kern = k<2, 1, 2>;
  else if (chains == 2 && ldsr == 1 && valu == 3) kern = k<2, 1, 3>;
  else if (chains == 2 && ldsr == 2 && valu == 3) kern = k<2, 2, 3>;
  else if (chains == 2 && ldsr == 1 && valu == 4) kern = k<2, 1, 4>;
  if (!kern) {
    fprintf(stderr, "combination not instantiated\n");
    return 2;
  }
  auto launch = [&](int iters) { hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters, pattern); };
  // calibrate the iteration count of one launch to ~50 ms, then launch back to back for the requested time
  launch(1000);
  (void)hipDeviceSynchronize();
  auto c0 = std::chrono::steady_clock::now();
  launch(20000);
  (void)hipDeviceSynchronize();
  const double t_probe = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
  const int iters = int(20000 * 0.05 / t_probe) + 1;
  const int launches = int(seconds / 0.05) + 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < launches; ++i) launch(iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = double(launches) * iters * 16.0 * cus * 4.0 * wps;     // wave-level instructions
  const double tflops = n_mfma * 32768.0 / (ms * 1e-3) / 1e12;
  // (s_memtime counts at a fixed 100 MHz on gfx950: cycles per MFMA = sampled shader clock / mfma_per_simd_per_s, by the caller)
  printf("{\"waves_per_simd\": %d, \"chains\": %d, \"pattern\": %d, \"lds_reads_per_2_mfma\": %d, \"valu_per_mfma\": %d, \"cus\": %d, \"seconds\": %.3f, \"tflops_bf16\": %.1f, \"frac_of_2500\": %.4f, "
         "\"mfma_per_simd_per_s\": %.4e}\n",
         wps, chains, pattern, ldsr, valu, cus, ms * 1e-3, tflops, tflops / 2500.0, n_mfma / (cus * 4.0) / (ms * 1e-3));
  return 0;
}
