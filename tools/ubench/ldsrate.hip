// LDS read / write throughput per CU: 512-thread workgroups (8 waves, one per CU), every wave issues N reads back to back (conflict-free: lane l reads
// slot l of its wave's run).  cycles per wave-instruction as seen by the CU = elapsed / (N x 8 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += WAVES * 64) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* p = lds + wv * 1024 + lane * (MODE == 0 ? 4 : MODE == 1 ? 2 : MODE == 3 ? 4 : 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"((unsigned)(size_t)p), "n"(0));
      else if (MODE == 1) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)p)); v[u] = (f32x4){t[0], t[1], 0.f, 0.f}; }
      else if (MODE == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)p)); v[u] = (f32x4){t, 0.f, 0.f, 0.f}; }
      else if (MODE == 3) { asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)p), "v"(acc) : "memory"); v[u] = acc; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int WAVES>
void run(const char* what) {
  const int grid = 256;
  float* out; long long* cyc;
  (void)hipMalloc(&out, grid * WAVES * 64 * 4); (void)hipMalloc(&cyc, grid * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 65536, 0, out, cyc, iters); (void)hipDeviceSynchronize();
  std::vector<long long> h(grid); (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= grid;
  printf("%-16s %d waves/CU: %.2f cycles per wave-instruction at the CU (%.1f per wave)\n", what, WAVES, mean / (iters * 8.0 * WAVES), mean / (iters * 8.0));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 4>("ds_read_b128"); run<0, 8>("ds_read_b128"); run<0, 16>("ds_read_b128");
  run<1, 4>("ds_read_b64"); run<1, 8>("ds_read_b64"); run<1, 16>("ds_read_b64");
  run<2, 8>("ds_read_b32");
  run<3, 4>("ds_write_b128"); run<3, 8>("ds_write_b128");
  return 0;
}
