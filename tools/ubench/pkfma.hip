// What does a packed fp32 FMA cost against two plain ones?  (The depthwise phases of the towers are 200 v_pk_fma_f32 per wave and chunk.)
// cycles per instruction and wave, 8 independent chains, 1 and 2 waves per SIMD; operand B a VGPR pair or an SGPR pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64, 1) void k(float* out, long long* cyc, int iters, float sa, float sb) {
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = (f32x2){(float)threadIdx.x + i, (float)threadIdx.x - i};
  f32x2 b = (f32x2){threadIdx.x * 0.002f + 1.f, 0.5f};
  f32x2 x = (f32x2){0.25f * threadIdx.x, 0.125f};
  f32x2 sc = (f32x2){sa, sb};
  asm volatile("" : "+s"(sc));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = u & 7;
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(x), "v"(b));
      else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(x), "s"(sc));
      else if (MODE == 2) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %4, %1" : "+v"(v[i][0]), "+v"(v[i][1]) : "v"(x[0]), "v"(b[0]), "v"(b[1]));
      else if (MODE == 3) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %4, %1" : "+v"(v[i][0]), "+v"(v[i][1]) : "v"(x[0]), "s"(sc[0]), "s"(sc[1]));
      else if (MODE == 4) asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(v[i][0]), "+v"(v[i][1]) : "s"(sc[0]), "v"(b[0]), "v"(b[1]));
      else if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(v[i]) : "v"(x), "v"(b));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(int grid, const char* what, int ninstr) {
  float* out; long long* cyc;
  hipMalloc(&out, grid * 64 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 1000;
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.5f, 0.75f); hipDeviceSynchronize();
  std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= grid;
  printf("%-44s %d wave(s)/SIMD: %.2f cycles per 128 FMAs per wave (%.2f per instruction)\n", what, grid / 1024, mean / (iters * 32.0), mean / (iters * 32.0 * ninstr));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int grid : {1024, 2048, 4096}) {
    run<0>(grid, "v_pk_fma_f32 v, v, v", 1);
    run<1>(grid, "v_pk_fma_f32 v, v, s", 1);
    run<2>(grid, "2 x v_fma_f32 v, v, v", 2);
    run<3>(grid, "2 x v_fma_f32 v, v, s", 2);
    run<4>(grid, "2 x v_fmac_f32 v, s, v", 2);
    run<5>(grid, "v_pk_mul_f32 v, v, v", 1);
  }
  return 0;
}
