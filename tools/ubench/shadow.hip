// How many independent VALU ops fit "for free" in the shadow of an fp32 MFMA issued by the SAME wave?
// cycles per (1 MFMA + N VALU) group for the 8-cycle 4x4x1 and the 32-cycle 16x16x4, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BIG, int NV>
__global__ __launch_bounds__(64, 1) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u) {
      const int i = u % 6;
      if (BIG) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[n % 8]) : "v"(b));   // 8 independent chains
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int BIG, int NV>
void run(int grid) {
  float* out; long long* cyc;
  hipMalloc(&out, grid * 64 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 1000;
  hipLaunchKernelGGL((k<BIG, NV>), dim3(grid), dim3(64), 0, 0, out, cyc, iters); hipDeviceSynchronize();
  std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= grid;
  printf("%s + %d VALU, %d wave(s)/SIMD: %.2f cycles per group per wave\n", BIG ? "16x16x4" : "4x4x1  ", NV, grid / 1024, mean / (iters * 24.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int grid : {1024, 2048}) {
    run<0, 0>(grid); run<0, 1>(grid); run<0, 2>(grid); run<0, 3>(grid); run<0, 4>(grid);
    run<1, 0>(grid); run<1, 1>(grid); run<1, 2>(grid); run<1, 4>(grid); run<1, 6>(grid); run<1, 8>(grid); run<1, 12>(grid);
  }
  return 0;
}
