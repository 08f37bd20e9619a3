// What does the WHOLE chip sustain on v_mfma_f32_32x32x16_bf16 when nothing else happens - no memory, no LDS, operands
// and accumulators in registers - and at what package power and shader clock?  The layer kernel's roofline (2500 TFLOP/s
// dense bf16 = 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz) assumes the nominal clock; under the 1400 W package cap the
// part does not hold it while the matrix pipes are busy.  This loop measures that ceiling directly:
//   mfma_chip <waves per SIMD: 1|2> <accumulator chains per wave: 2|4> <seconds> [operand pattern: 0 zeros | 1 random bits]
//             [ds_read_b128 per two MFMAs: 0|1|2] [fp32 VALU fillers per MFMA: 0|2|3|4] [weight stream by LDS-DMA: 0|1] [HBM streams: 0|1]
//             [operand sharing: 0|1|2|3]
// Round 5: does it matter, in joules per MFMA, WHICH operands consecutive MFMAs share?  Registers only,
// two accumulator chains, random operand bits:  0 = the rotation above (A changes every 2nd MFMA, B every MFMA);  1 = three
// consecutive MFMAs share their A registers (the layer kernel's "same weight fragment x the three activation pieces", issued
// back to back), B changes every MFMA;  2 = three consecutive share B, A changes every MFMA;  3 = every MFMA reads the same A and
// the same B (random bits that never change: what the datapath costs when no input toggles).
// The last two add what the layer kernel does around its MFMAs - the A fragments come out of LDS (random bits, a different
// KiB every read) and independent v_fma_f32 fill the issue slots behind each MFMA - still without any global memory: how
// much clock do those cost at the cap?
// Round 4 (what is between this mix at 0.66 / 2.08 GHz and the layer kernel at 0.54 / 1.81 GHz?) adds the
// layer kernel's two kinds of data movement, at its own densities, on top of the mix:
//   weight stream   one global_load_lds_dwordx4 (1 KiB per wave, M0-addressed LDS-DMA) per 8 MFMAs = 128 B per MFMA, walking
//                   a 4 MB buffer every block shares (the per-layer stream: resident in each XCD's 4 MB L2), into a
//                   144 KiB LDS ring - 8.4 GB per 1.55 ms launch in the real kernel = 5.4 TB/s of L2 -> LDS traffic;
//   HBM streams     per 128 MFMAs one 1-KiB non-temporal global_load_dwordx4 and one 1-KiB + (every 4th) a second
//                   global_store_dwordx4 per wave, walking 1 GiB buffers (re-reference distance > the 256 MB MALL): the
//                   activation fragments, 2 KiB read + 2.4 KiB written per token = ~0.9 TB/s at the kernel's rate.
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

template <int CHAINS, int LDSR, int VALU, int DMA = 0, int HBM = 0, int SHARE = 0>
__global__ void __launch_bounds__(512, 1) k(float* sink, unsigned long long* cyc, int iters, int pattern, const u32x4* wstream,
                                            const u32x4* hbm_in, u32x4* hbm_out) {
  extern __shared__ u32x4 lds[];                          // 48 KiB stage image read by the fragments (+ 96 KiB more = the 3-slot ring with DMA)
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
  // weight stream: a block's 4 waves fetch 4 KiB per piece, consecutive pieces walk the 4 MB buffer; blocks start spread over it
  const unsigned long long wbase = (unsigned long long)(size_t)wstream;
  unsigned wpos = (blockIdx.x * 16384u) & (4u * 1024 * 1024 - 1);
  const unsigned wlane = (threadIdx.x >> 6) * 1024u + lane * 16u;
  unsigned ring = 0;
  // HBM streams: 1 KiB per wave and event, the waves of the chip interleaved, wrapping at 1 GiB
  const unsigned gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned nw = gridDim.x * (blockDim.x >> 6);
  unsigned hpos = gw;
  u32x4 pend = u32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (HBM && (it & 7) == 0) {
      // consume the fragment fetched 128 MFMAs ago (its wait is free by now), store a result fragment, fetch the next one
      // (inline asm: a compiler-visible load / store makes hipcc put "s_waitcnt vmcnt(0..1)" in front of the next use of their
      // registers, which - vector memory completes in order - drains every DMA piece issued since; the real kernel avoids that
      // by front-loading its pieces.  Here the wait is explicit and counted: 16 DMA pieces have been issued since
      // the fetch consumed now, so vmcnt(16) = "the fetch and the stores of the previous event have landed", the pieces stay in
      // flight.  Without any wait a late-landing fetch overwrites a re-used address register: r04b faulted exactly that way.)
      if (DMA) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const u32x4 got = pend;
      const size_t slot = size_t(hpos & (1u << 20) - 1) * 64 + lane;                    // 2^20 KiB = 1 GiB
      const u32x4* src = hbm_in + slot;
      u32x4* dst = hbm_out + slot;
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(pend) : "v"(src) : "memory");
      u32x4 st = a[0];
      st[0] ^= got[0] ^ got[1] ^ got[2] ^ got[3];
      asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(st) : "memory");
      if ((it & 31) == 0) {
        u32x4* dst2 = hbm_out + (slot ^ (1u << 19) * 64);                                // 2.4 KiB written per 2 KiB read
        asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst2), "v"(st) : "memory");
      }
      a[3][3] = (a[3][3] & 0x807F807Fu) | 0x3F803F80u | (got[0] & 0x007F007Fu);       // the fetched bits reach the multiplier array
      hpos += nw;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (DMA && (u & 7) == 0) {
        // the layer kernel's stream_piece: M0 = ring slot, 64 lanes x 16 B from (base + voff); at most 24 pieces in flight = the
        // two stages of look-ahead of the kernel's 3-slot ring (with 12 the loop is bound by DMA latency x concurrency, not by power:
        // profiles/r04b_power_components_12_in_flight.json - 1.7 - 2.2 us per piece under load, 5.6 TB/s instead of the 7.1 demanded)
        const unsigned m0 = lds_base + 49152u + ring + (threadIdx.x >> 6) * 1024u;
        const unsigned voff = ((wpos & (4u * 1024 * 1024 - 1)) + wlane);
        asm volatile(
            "s_waitcnt vmcnt(23)\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, %1"
            :
            : "v"(voff), "s"(wbase), "s"(__builtin_amdgcn_readfirstlane(m0))
            : "memory");
        wpos += 4096u;
        ring = ring + 4096u >= 98304u ? 0u : ring + 4096u;
      }
      // the fragment read here is consumed two operand rotations later
      // asm, so that the read stays HERE (the compiler would sink it next to its use and expose the LDS latency).  The counted
      // wait only bounds the reads in flight (a wait for the previous read - lgkmcnt(1) - stalls the wave for most of an LDS
      // round trip every two MFMAs: 43 cycles per MFMA instead of 33, r03l); the MFMAs may consume a fragment register whose
      // read is still in flight - stale bits, which is all a power benchmark needs
      if (LDSR && (u % 2 == 0 || LDSR == 2)) {
        const unsigned addr = lds_base + (((unsigned(it) * 16u + u) * 64u + lane) & 2047u) * 16u     /* (a power of two: no integer division in the loop) */;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(3)" : "=v"(a[(u / CHAINS + 2) & 3]) : "v"(addr));
      }
      constexpr int dummy = 0;
      (void)dummy;
      const int ai = SHARE == 0 ? (u / CHAINS) & 3 : SHARE == 1 ? (u / 3) & 3 : SHARE == 2 ? u & 3 : 0;
      const int bi = SHARE == 0 ? (u / CHAINS + u) & 3 : SHARE == 1 ? u & 3 : SHARE == 2 ? (u / 3) & 3 : 0;
      acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ai]), __builtin_bit_cast(bf16x8, b[bi]), acc[u % CHAINS], 0, 0, 0);
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
  const int dma = argc > 7 ? atoi(argv[7]) : 0, hbm = argc > 8 ? atoi(argv[8]) : 0, share = argc > 9 ? atoi(argv[9]) : 0;
  u32x4 *wstream = nullptr, *hbm_in = nullptr, *hbm_out = nullptr;
  if (dma) {
    if (hipMalloc(&wstream, 4u << 20) != hipSuccess) return 1;
    unsigned* hw = (unsigned*)malloc(4u << 20);
    unsigned sd = 99991u;
    for (size_t i = 0; i < (1u << 20); ++i) {
      sd = sd * 1664525u + 1013904223u;
      hw[i] = 0x3F803F80u | ((sd >> 8) & 0x807F807Fu);
    }
    (void)hipMemcpy(wstream, hw, 4u << 20, hipMemcpyHostToDevice);
    free(hw);
  }
  if (hbm) {
    if (hipMalloc(&hbm_in, size_t(1) << 30) != hipSuccess || hipMalloc(&hbm_out, size_t(1) << 30) != hipSuccess) return 1;
    (void)hipMemset(hbm_in, 0x3c, size_t(1) << 30);
  }
  void (*kern)(float*, unsigned long long*, int, int, const u32x4*, const u32x4*, u32x4*) = nullptr;
  if (share) {
    if (chains == 2 && !ldsr && !valu && !dma && !hbm) kern = share == 1 ? k<2, 0, 0, 0, 0, 1> : share == 2 ? k<2, 0, 0, 0, 0, 2> : k<2, 0, 0, 0, 0, 3>;
  } else if (dma || hbm) {
    if (chains == 2 && ldsr == 1 && valu == 3 && dma == 1 && hbm == 0) kern = k<2, 1, 3, 1, 0>;
    else if (chains == 2 && ldsr == 1 && valu == 3 && dma == 1 && hbm == 1) kern = k<2, 1, 3, 1, 1>;
    else if (chains == 2 && ldsr == 1 && valu == 3 && dma == 0 && hbm == 1) kern = k<2, 1, 3, 0, 1>;
    else if (chains == 2 && ldsr == 0 && valu == 0 && dma == 1 && hbm == 0) kern = k<2, 0, 0, 1, 0>;
  } else if (chains == 4 && !ldsr && !valu) kern = k<4, 0, 0>;
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
  const int lds_bytes = dma ? 147456 : (ldsr ? 49152 : 16);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 3;
  auto launch = [&](int iters) {
    hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), lds_bytes, 0, sink, cyc, iters, pattern, wstream, hbm_in, hbm_out);
  };
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
  // data moved per second by the two added streams (derived from the instruction counts, not measured)
  const double mfma_per_wave_s = n_mfma / (cus * 4.0 * wps) / (ms * 1e-3);
  const double dma_tbs = dma ? mfma_per_wave_s / 8.0 * 1024.0 * cus * 4.0 * wps / 1e12 : 0.0;
  const double hbm_tbs = hbm ? mfma_per_wave_s / 128.0 * (1024.0 + 1024.0 * 1.25) * cus * 4.0 * wps / 1e12 : 0.0;
  printf("{\"waves_per_simd\": %d, \"chains\": %d, \"pattern\": %d, \"lds_reads_per_2_mfma\": %d, \"valu_per_mfma\": %d, \"lds_dma_weight_stream\": %d, \"hbm_streams\": %d, \"operand_sharing\": %d, \"cus\": %d, \"seconds\": %.3f, \"tflops_bf16\": %.1f, \"frac_of_2500\": %.4f, "
         "\"mfma_per_simd_per_s\": %.4e, \"l2_to_lds_tb_per_s\": %.2f, \"hbm_tb_per_s\": %.2f}\n",
         wps, chains, pattern, ldsr, valu, dma, hbm, share, cus, ms * 1e-3, tflops, tflops / 2500.0, n_mfma / (cus * 4.0) / (ms * 1e-3), dma_tbs, hbm_tbs);
  return 0;
}
