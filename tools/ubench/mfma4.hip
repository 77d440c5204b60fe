// Microbenchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 / v_mfma_f32_16x16x4_f32 and cost of DPP wave shifts beside them.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4.hip -o tools/ubench/mfma4 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>   // MODE 0: 4x4x1, 1: 16x16x4, 2: 4x4x1 + one dpp wave_shr per 3 mfma, 3: 4x4x1 + one v_max per mfma, 4: 4x4x1 + 1 v_max per mfma with row_shr dpp
__global__ __launch_bounds__(64, 1) void k(float* out, long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f, c = 0.5f + threadIdx.x, d = 0.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        if (MODE == 3) { d = fmaxf(d, c); c += 1.0f; }
        if (MODE == 4) { c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x111, 0xf, 0xf, true)); d = fmaxf(d, c); }
      }
      if (MODE == 2) c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138, 0xf, 0xf, true)) + 1.0f;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = c + d;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void layout(float* out) {  // D for a = lane, b = 100*lane
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)threadIdx.x, 100.f * threadIdx.x, z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = d[r];
}

template <int ABID>
__global__ void layout_bcast(float* out) {  // cbsz = 4: every block takes its A from block ABID
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)threadIdx.x, 100.f * threadIdx.x, z, 4, ABID, 0);
  for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = d[r];
}
template <int NACC>
__global__ __launch_bounds__(64, 1) void kb(float* out, long long* cyc, int iters) {   // rate with cbsz=4 broadcast
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 5, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
void run(const char* name, int grid) {
  float* out; long long* cyc;
  hipMalloc(&out, grid * 64 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(64), 0, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(64), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 24 * NACC;
  // readcyclecounter = s_memtime (100 MHz constant clock on gfx9?) -> report both
  printf("%-44s grid %5d: %.3f ms, %.2f ns per mfma per wave, counter ticks/mfma %.3f\n", name, grid, ms, ms * 1e6 / n, h[0] / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  float* o; hipMalloc(&o, 256 * 4); hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, o);
  std::vector<float> h(256); hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float e = (float)(4 * (l / 4) + r) * 100.f * l; if (h[l * 4 + r] != e) ++bad; }
  printf("layout D[lane 4b+j][reg i] = A[4b+i]*B[4b+j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  {
    int badb = 0;
    hipLaunchKernelGGL(layout_bcast<5>, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float e = (float)(4 * 5 + r) * 100.f * l; if (h[l * 4 + r] != e) ++badb; }
    hipLaunchKernelGGL(layout_bcast<13>, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float e = (float)(4 * 13 + r) * 100.f * l; if (h[l * 4 + r] != e) ++badb; }
    printf("cbsz=4 abid=j: D[lane][reg i] = A[4j+i]*B[lane]: %s (%d mismatches)\n", badb ? "NO" : "yes", badb);
    float* out; long long* cyc; hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    hipLaunchKernelGGL((kb<6>), dim3(1024), dim3(64), 0, 0, out, cyc, 2000); hipDeviceSynchronize();
    long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    printf("4x4x1 6 acc with cbsz=4 broadcast: ticks/mfma %.3f\n", c0 / (2000.0 * 24 * 6));
  }
  for (int grid : {1024}) {
    run<6, 0>("4x4x1 6 acc", grid);
    run<3, 0>("4x4x1 3 acc", grid);
    run<2, 0>("4x4x1 2 acc", grid);
    run<1, 0>("4x4x1 1 acc", grid);
    run<4, 1>("16x16x4 4 acc", grid);
    run<6, 2>("4x4x1 6 acc + wave_shr/6", grid);
    run<6, 3>("4x4x1 6 acc + 2 valu per mfma", grid);
    run<6, 4>("4x4x1 6 acc + row_shr dpp + max per mfma", grid);
  }
  return 0;
}
