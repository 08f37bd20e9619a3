// What does the L1 / texture-address path of one MI355X CU sustain for the access pattern of the deformable-attention
// gather (k_msda_gather_sb_pad: every lane loads 16 B, 8 lanes cover one 128-B head slice of a value-map pixel, the 8
// lane groups of a wave sit at 8 unrelated pixels of a small window), and how does it depend on
//   * loads in flight per wave (IN_FLIGHT = 4 / 8 / 16 / 32),
//   * waves per SIMD (blocks of 512 threads; LDS padding limits the number of resident blocks),
//   * the layout of the value map (SEG = 128: token-major as today; SEG = 256: head-major, where the two x-adjacent
//     corners of a bilinear stencil are one contiguous 256-B piece)?
// Prints bytes / clock / CU for each combination; the kernel's 26.9 B/clk/CU (0.319 ms for 4.3 GB of taps) is to be read
// against these.  hipcc --offload-arch=gfx950 -O3 gather_ta.hip -o gather_ta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;

// map: (rows x cols) pixels of 1 KiB (8 heads x 128 B); a wave walks tokens, per token 16 taps per head
template <int IN_FLIGHT, int SEG>
__global__ void __launch_bounds__(512) k(const float* __restrict__ map, float* __restrict__ sink, int cols, int rows, int tokens_per_wave,
                                          int lds_pad_words) {
  extern __shared__ float pad[];
  if (lds_pad_words < 0) pad[threadIdx.x] = 0.f;       // keeps the dynamic LDS allocation alive
  const int lane = threadIdx.x & 63;
  const int gwave = (blockIdx.x * 512 + threadIdx.x) >> 6;
  const int hd = lane >> 3;
  const unsigned lane_off = unsigned(hd * 128 + (lane & 7) * 16);
  unsigned seed = gwave * 2654435761u + hd * 40503u + 12345u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int t0 = gwave * tokens_per_wave;
  for (int t = 0; t < tokens_per_wave; ++t) {
    const int tok = t0 + t;
    const int px = tok % cols, py = (tok / cols) % rows;
#pragma unroll
    for (int g = 0; g < 16 / IN_FLIGHT + (16 % IN_FLIGHT ? 1 : 0); ++g) {
      f32x4 v[IN_FLIGHT < 16 ? IN_FLIGHT : 16];
#pragma unroll
      for (int i = 0; i < (IN_FLIGHT < 16 ? IN_FLIGHT : 16); ++i) {
        // per head: a pixel within +-4 of the token (the offsets of a head), corners (dy, dx) of the bilinear stencil
        const int tap = g * (IN_FLIGHT < 16 ? IN_FLIGHT : 16) + i;
        if ((tap & 3) == 0) seed = seed * 1664525u + 1013904223u;
        const int ox = int((seed >> 8) & 7) - 4, oy = int((seed >> 16) & 7) - 4;
        int x = px + ox + (tap & 1), y = py + oy + ((tap >> 1) & 1);
        x = min(max(x, 0), cols - 1);
        y = min(max(y, 0), rows - 1);
        // SEG = 128: token-major map (pixel = 1 KiB, head slice = 128 B: today's layout); SEG = 256: head-major map
        // ([head][y][x][32 ch]: the two x-adjacent corners of a bilinear stencil are one contiguous 256-B piece)
        const unsigned off = SEG == 256 ? (unsigned(hd * rows + y) * unsigned(cols) + unsigned(x)) * 128u + unsigned(lane & 7) * 16u
                                        : unsigned(y * cols + x) * 1024u + lane_off;
        v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(map) + off);
      }
#pragma unroll
      for (int i = 0; i < (IN_FLIGHT < 16 ? IN_FLIGHT : 16); ++i) acc += v[i];
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int IN_FLIGHT, int SEG>
void run(const float* map, float* sink, int cols, int rows, int blocks_per_cu, int n_cu) {
  // resident blocks per CU are limited through the dynamic LDS size: 160 KiB / blocks_per_cu
  const int lds = (160 * 1024) / blocks_per_cu - 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<IN_FLIGHT, SEG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tokens_per_wave = 64;
  const int blocks = n_cu * blocks_per_cu * 4;         // 4 rounds of resident blocks
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<IN_FLIGHT, SEG>), dim3(blocks), dim3(512), lds, 0, map, sink, cols, rows, tokens_per_wave, 0);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<IN_FLIGHT, SEG>), dim3(blocks), dim3(512), lds, 0, map, sink, cols, rows, tokens_per_wave, 0);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = double(blocks) * 8 * tokens_per_wave * 16 * 1024.0;     // 16 wave-loads of 1 KiB per token
  const double clk = 1.95e9;                                                    // sustained clock under load
  printf("  in flight %2d/wave, %d waves/SIMD, %3d-B pieces: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n", IN_FLIGHT, blocks_per_cu * 2, SEG, ms,
         bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / clk / n_cu);
}

int main() {
  int n_cu = 256, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  const int cols = 258, rows = 130 * 8;                // the padded value maps of the headline workload: 8 x 130 x 258 pixels of 1 KiB
  float *map, *sink;
  (void)hipMalloc(&map, size_t(cols) * rows * 1024 + 4096);
  (void)hipMemset(map, 0, size_t(cols) * rows * 1024 + 4096);
  (void)hipMalloc(&sink, size_t(n_cu) * 16 * 4 * 512 * 4);
  printf("gather tap traffic on %d CUs (value map %d x %d pixels of 1 KiB = %.0f MB):\n", n_cu, rows, cols, cols * rows / 1024.0);
  for (int bpc = 1; bpc <= 4; ++bpc) {
    run<4, 128>(map, sink, cols, rows, bpc, n_cu);
    run<8, 128>(map, sink, cols, rows, bpc, n_cu);
    run<16, 128>(map, sink, cols, rows, bpc, n_cu);
    run<16, 256>(map, sink, cols, rows, bpc, n_cu);
  }
  return 0;
}
