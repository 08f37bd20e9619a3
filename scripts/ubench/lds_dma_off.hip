// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// hipcc --offload-arch=gfx950 -O2 lds_dma_off.hip -o lds_dma_off && ./lds_dma_off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) float lds_float_t;
__global__ void k(const float* g, float* out) {
  __shared__ __attribute__((aligned(16))) float sm[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) sm[i] = -1.f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)sm;
  const unsigned voff = threadIdx.x * 16;
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
      "s_waitcnt vmcnt(0)"
      :: "v"(voff), "s"(g), "s"(lds0) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = sm[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = float(i);
  float *g, *o;
  hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
  std::vector<float> r(2048);
  hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
  int first = -1; for (int i = 0; i < 2048; ++i) if (r[i] >= 0.f) { first = i; break; }
  printf("first written LDS float index = %d (256 => LDS side moved by the 1024-B offset), value there = %.0f (256 => global side moved)\n", first, first >= 0 ? r[first] : -1.f);
  int cnt = 0; for (int i = 0; i < 2048; ++i) cnt += r[i] >= 0.f;
  printf("floats written = %d\n", cnt);
  return 0;
}
