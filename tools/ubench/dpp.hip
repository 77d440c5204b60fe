// DPP semantics probe: row_shr:1 / row_shl:1 / wave_shr:1 with bound_ctrl on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* in) {
  f32x4 v = reinterpret_cast<const f32x4*>(in)[threadIdx.x];
  // NOTE: bit_cast(int, v[3]) on the element lvalue reads element 0 with hipcc 7.2 (prints l instead of l+300 below)
  float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[3]), 0x111, 0xf, 0xf, true));
  float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[3]), 0x101, 0xf, 0xf, true));
  float c = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[3]), 0x138, 0xf, 0xf, true));
  out[threadIdx.x * 3] = a; out[threadIdx.x * 3 + 1] = b; out[threadIdx.x * 3 + 2] = c;
}
int main() {
  float h[256], *d, *o, r[192];
  for (int i = 0; i < 256; ++i) h[i] = i / 4 + 100.f * (i % 4);   // v[3] of lane l = l + 300
  hipMalloc(&d, 1024); hipMalloc(&o, 768); hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d); hipMemcpy(r, o, 768, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 15, 16, 17, 31, 32, 63}) printf("lane %2d: row_shr1 %.0f row_shl1 %.0f wave_shr1 %.0f\n", l, r[3 * l], r[3 * l + 1], r[3 * l + 2]);
  return 0;
}
