// hbm_calib.hip - known-byte streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md "HBM":
// FETCH_SIZE reports half the bytes of a wide coalesced read, "other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Each kernel moves exactly BYTES (1 GiB, four times the 256-MiB Infinity Cache) in
// the access patterns the library's kernels use:
//   k_store16      a wave stores 1 KiB pieces (64 lanes x 16 B), plain            - fragment-major q / u / probabilities
//   k_store16_nt   the same with the non-temporal hint                            - the layer kernel's q' / v' / sample-table stores at C2
//   k_store_row32  a lane pair stores 32 B of a 1-KiB row, 32 rows per instruction - the padded value map's rows (v_out)
//   k_load16       a wave loads 1 KiB pieces, plain / k_load16_nt non-temporal     - every activation stream
//   k_copy16       load + store                                                    - the sum must equal the two singles
// Run under  rocprofv3 --pmc WRITE_SIZE --kernel-trace  and  --pmc FETCH_SIZE --kernel-trace  (separate passes);
// scripts/collect_profiles.py turns the per-kernel counters into  factor = known bytes / counted bytes.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/hbm_calib scripts/ubench/hbm_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t BYTES = size_t(1) << 30;
constexpr int THREADS = 256;

__global__ void __launch_bounds__(THREADS) k_store16(f32x4* __restrict__ p, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * THREADS + threadIdx.x; i < n16; i += size_t(gridDim.x) * THREADS) p[i] = f32x4{1.f, 2.f, 3.f, float(i)};
}
__global__ void __launch_bounds__(THREADS) k_store16_nt(f32x4* __restrict__ p, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * THREADS + threadIdx.x; i < n16; i += size_t(gridDim.x) * THREADS)
    __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, float(i)}, p + i);
}
// rows of 1 KiB; a wave's instruction covers 32 rows x 32 B (lane = (row j, half h), 16 B each), 32 instructions finish the rows
__global__ void __launch_bounds__(THREADS) k_store_row32(float* __restrict__ p, size_t rows) {
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const size_t wave = (size_t(blockIdx.x) * THREADS + threadIdx.x) >> 6, nw = (size_t(gridDim.x) * THREADS) >> 6;
  for (size_t g = wave; g * 32 < rows; g += nw) {
    float* row = p + (g * 32 + j) * 256 + 4 * h;
#pragma unroll
    for (int c = 0; c < 32; ++c) __builtin_nontemporal_store(f32x4{1.f, 2.f, float(c), float(j)}, reinterpret_cast<f32x4*>(row + 8 * c));
  }
}
__global__ void __launch_bounds__(THREADS) k_load16(const f32x4* __restrict__ p, size_t n16, float* __restrict__ sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = size_t(blockIdx.x) * THREADS + threadIdx.x; i < n16; i += size_t(gridDim.x) * THREADS) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
__global__ void __launch_bounds__(THREADS) k_load16_nt(const f32x4* __restrict__ p, size_t n16, float* __restrict__ sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = size_t(blockIdx.x) * THREADS + threadIdx.x; i < n16; i += size_t(gridDim.x) * THREADS) acc += __builtin_nontemporal_load(p + i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];
}
__global__ void __launch_bounds__(THREADS) k_copy16(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n16) {
  for (size_t i = size_t(blockIdx.x) * THREADS + threadIdx.x; i < n16; i += size_t(gridDim.x) * THREADS) b[i] = a[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  float *a = nullptr, *b = nullptr, *sink = nullptr;
  CK(hipMalloc(&a, BYTES));
  CK(hipMalloc(&b, BYTES));
  CK(hipMalloc(&sink, 256));
  CK(hipMemset(a, 0, BYTES));
  CK(hipMemset(b, 0, BYTES));
  const size_t n16 = BYTES / 16, rows = BYTES / 1024;
  const int grid = 256 * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[6] = {"k_store16", "k_store16_nt", "k_store_row32", "k_load16", "k_load16_nt", "k_copy16"};
  for (int k = 0; k < 6; ++k) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      switch (k) {
        case 0: hipLaunchKernelGGL(k_store16, dim3(grid), dim3(THREADS), 0, 0, reinterpret_cast<f32x4*>(a), n16); break;
        case 1: hipLaunchKernelGGL(k_store16_nt, dim3(grid), dim3(THREADS), 0, 0, reinterpret_cast<f32x4*>(a), n16); break;
        case 2: hipLaunchKernelGGL(k_store_row32, dim3(grid), dim3(THREADS), 0, 0, a, rows); break;
        case 3: hipLaunchKernelGGL(k_load16, dim3(grid), dim3(THREADS), 0, 0, reinterpret_cast<const f32x4*>(a), n16, sink); break;
        case 4: hipLaunchKernelGGL(k_load16_nt, dim3(grid), dim3(THREADS), 0, 0, reinterpret_cast<const f32x4*>(a), n16, sink); break;
        default: hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(THREADS), 0, 0, reinterpret_cast<const f32x4*>(a), reinterpret_cast<f32x4*>(b), n16); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("{\"kernel\": \"%s\", \"bytes_read\": %zu, \"bytes_written\": %zu, \"best_ms\": %.4f, \"tb_per_s\": %.3f}\n", names[k],
           k >= 3 ? BYTES : size_t(0), (k < 3 || k == 5) ? BYTES : size_t(0), best, double(BYTES) * (k == 5 ? 2 : 1) / (best * 1e-3) / 1e12);
  }
  return 0;
}
