// How is a wave64 ds_read_b128 serviced?  Full-wave address patterns, 8 waves per CU hammering; cycles per wave-instruction at the CU (4.0 = no conflict).
//   contiguous      lane l -> slot l                                                   (free under every hypothesis)
//   guide groups    16 distinct slot residues inside each of {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 - but NOT inside 16 consecutive lanes
//   consecutive     16 distinct residues inside lanes 16k .. 16k+15 - but not inside the guide's groups
//   half-wave pairs lane l and l+32 on the same banks (different addresses)
//   quad stride     lane l and l+8 on the same banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, const int* slot_of_lane) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const unsigned addr = slot_of_lane[lane] * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
double run(const std::vector<int>& slots) {
  const int grid = 256;
  static float* out = nullptr; static long long* cyc = nullptr; static int* d = nullptr;
  if (!out) { (void)hipMalloc(&out, grid * 512 * 4); (void)hipMalloc(&cyc, grid * 8); (void)hipMalloc(&d, 256); }
  (void)hipMemcpy(d, slots.data(), 256, hipMemcpyHostToDevice);
  const int iters = 500;
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), 65536, 0, out, cyc, iters, d); (void)hipDeviceSynchronize();
  std::vector<long long> h(grid); (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : h) mean += c; mean /= grid;
  return mean / (iters * 8.0 * 8);
}
int main(int argc, char** argv) {
  std::vector<int> s(64);
  for (int l = 0; l < 64; ++l) s[l] = l;
  printf("contiguous:                       %.2f\n", run(s));
  const int G[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                        {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  for (int g = 0; g < 4; ++g) for (int i = 0; i < 16; ++i) s[G[g][i]] = i + 16 * g + 64 * (i & 3);   // residues 0..15 per guide group, addresses all different
  printf("free for the guide's groups only: %.2f\n", run(s));
  // consecutive-16 free, guide groups conflicting: lanes 16k+i -> residue i, but e.g. lanes 0-3 and 12-15 ... are distinct anyway; make guide groups collide:
  // inside G0 = {0-3, 12-15, 20-27}: lane 20 -> residue 4 (collides with nothing in consecutive view) - use residue = (l % 16) rotated per 16-lane block by 8
  for (int l = 0; l < 64; ++l) s[l] = ((l + 8 * (l / 16)) & 15) + 16 * (l / 16) + 64 * (l & 3);
  printf("free for consecutive 16 (rotated): %.2f\n", run(s));
  for (int l = 0; l < 64; ++l) s[l] = (l & 31) + 64 * (l >> 5);
  printf("lanes l and l+32 on the same banks: %.2f\n", run(s));
  for (int l = 0; l < 64; ++l) s[l] = (l & 7) + 16 * (l >> 3) ;
  printf("lanes l and l+8 share banks (8 distinct residues per 16 lanes): %.2f\n", run(s));
  for (int l = 0; l < 64; ++l) s[l] = (l & 3) + 16 * (l >> 2);
  printf("lanes l and l+4 share banks (4 distinct residues): %.2f\n", run(s));
  for (int l = 0; l < 64; ++l) s[l] = 16 * l;
  printf("all lanes on the same banks: %.2f\n", run(s));
  // the towers' natural patch order: slot = py * 26 + px, patch = lane (11 per row)
  for (int l = 0; l < 64; ++l) s[l] = (l / 11) * 26 + l % 11;
  printf("towerp natural order, pitch 26: %.2f\n", run(s));
  for (int l = 0; l < 64; ++l) s[l] = (l / 11) * 27 + l % 11;
  printf("patch-row pitch 27 (towerh's): %.2f\n", run(s));
  for (int P : {28, 30, 32, 34, 36, 38, 40, 42, 44}) { for (int l = 0; l < 64; ++l) s[l] = (l / 11) * P + l % 11; printf("patch-row pitch %d: %.2f\n", P, run(s)); }
  return 0;
}
