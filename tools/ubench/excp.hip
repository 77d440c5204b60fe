// Does the wave's sticky exception word (HW_REG_TRAPSTS.EXCP, accumulated whether or not traps are enabled) see the events the
// fp16x3 split can produce on gfx950?  (a) v_cvt_pk_f16_f32 of a value beyond fp16's range (the "cliff" of include/yfv2.h),
// (b) the same through the scalar v_cvt_f16_f32, (c) a clean run, (d) an MFMA fed with Inf operands (Inf - Inf inside the
// matrix core), (e) a wave that only rounds (inexact must not look like overflow).  Prints the 9 EXCP bits per case:
// bit0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 underflow, 5 inexact, 6 int div0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define TRAPSTS_EXCP (3 | (0 << 6) | (8 << 11))
__global__ void k(unsigned* out, const float* in, int mode) {
  const int lane = threadIdx.x;
  f2 v = {in[2 * lane], in[2 * lane + 1]};
  unsigned keep = 0;
  if (mode == 0 || mode == 2 || mode == 4) {          // packed convert
    h2 t = __builtin_convertvector(v, h2);
    keep = __builtin_bit_cast(unsigned, t);
  } else if (mode == 1) {                               // scalar convert
    _Float16 t = (_Float16)v[0];
    keep = __builtin_bit_cast(unsigned short, t);
  } else if (mode == 3) {                               // MFMA on Inf operands
    h2 t = __builtin_convertvector(v, h2);
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (i & 1) ? t[0] : -t[0]; b[i] = (_Float16)1.0f; }
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    keep = __builtin_bit_cast(unsigned, acc[0]);
  }
  const unsigned ts = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
  out[lane * 2] = ts;
  out[lane * 2 + 1] = keep;
}
int main() {
  float h[128], *d; unsigned *o, r[128];
  hipMalloc(&d, 512); hipMalloc(&o, 512);
  const char* names[] = {"packed cvt of 1e6 (lane 5 only)", "scalar cvt of 1e6 (lane 5 only)", "packed cvt, all values < 100 (clean)", "MFMA with +-Inf operands (lane 5's 1e6)", "packed cvt of 65504.0 and 65519.9 (largest that still round to finite)"};
  for (int mode = 0; mode < 5; ++mode) {
    for (int i = 0; i < 128; ++i) h[i] = 0.37f * (i + 1);
    if (mode == 0 || mode == 1 || mode == 3) h[10] = 1e6f;
    if (mode == 4) { h[10] = 65504.0f; h[11] = 65519.9f; }
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, mode); hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    printf("%-75s EXCP = 0x%03x  (overflow bit %d, invalid bit %d, inexact bit %d)  lane5 result bits 0x%08x\n", names[mode], r[0], (r[0] >> 3) & 1, r[0] & 1, (r[0] >> 5) & 1, r[11]);
  }
  return 0;
}
