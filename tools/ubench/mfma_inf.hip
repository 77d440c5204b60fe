// How does v_mfma_f32_16x16x32_f16 treat non-finite fp16 operands on gfx950?  B = all ones; A row r holds special values:
//   row 1: one +Inf, rest 1        -> IEEE: +Inf          row 2: +Inf and -Inf      -> IEEE: NaN
//   row 3: one NaN                 -> IEEE: NaN           row 4: +Inf times B = 0 in that slot (0 * Inf) -> IEEE: NaN
//   row 5: 65504 (largest finite) x 4                     row 6: the fp16x3 overflow pattern: {x1 = +Inf, x2 = -Inf} against w = {1, 2^-11}
// Prints D[r][0] for each row (as bits and as a float).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const unsigned short* abits, const unsigned short* bbits) {
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(_Float16, abits[l * 32 + 8 * g + i]);     // A[row l][k = 8g + i]
    b[i] = __builtin_bit_cast(_Float16, bbits[l * 32 + 8 * g + i]);     // B[k = 8g + i][col l]
  }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + l] = acc[r];       // D[row 4g + r][col l]
}
int main() {
  unsigned short A[16 * 32], Bm[16 * 32];
  const unsigned short ONE = 0x3c00, PINF = 0x7c00, NINF = 0xfc00, QNAN = 0x7e00, MAXF = 0x7bff, TINY = 0x1000 /* 2^-11 */;
  for (int i = 0; i < 512; ++i) { A[i] = ONE; Bm[i] = ONE; }
  A[1 * 32 + 3] = PINF;
  A[2 * 32 + 3] = PINF; A[2 * 32 + 9] = NINF;
  A[3 * 32 + 5] = QNAN;
  A[4 * 32 + 7] = PINF;                       // and B[k = 7][col 0] = 0 below
  for (int c = 0; c < 1; ++c) Bm[c * 32 + 7] = 0;   // column 0 only: rows other than 4 just lose one term
  for (int kk = 0; kk < 4; ++kk) A[5 * 32 + kk] = MAXF;
  // row 6: slots 0, 1 = the two products w1 * x2 + w2 * x1 of one overflowed operand: A = {w1 = 1, w2 = 2^-11}, B col 0 = {x2 = -Inf, x1 = +Inf}
  A[6 * 32 + 0] = ONE; A[6 * 32 + 1] = TINY; Bm[0 * 32 + 0] = NINF; Bm[0 * 32 + 1] = PINF;
  unsigned short *da, *db; float* dout; float h[256];
  hipMalloc(&da, sizeof(A)); hipMalloc(&db, sizeof(Bm)); hipMalloc(&dout, 1024);
  hipMemcpy(da, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(db, Bm, sizeof(Bm), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db); hipMemcpy(h, dout, 1024, hipMemcpyDeviceToHost);
  const char* what[] = {"row 0: 32 ones (col 0: 31, slot 7 of B is 0)", "row 1: one +Inf", "row 2: +Inf and -Inf", "row 3: one NaN", "row 4: +Inf x 0", "row 5: 4 x 65504 + 28",
                        "row 6: {1, 2^-11} x {-Inf, +Inf} + 30"};
  for (int r = 0; r < 7; ++r) { unsigned u; memcpy(&u, &h[r * 16], 4); printf("%-50s D[r][0] = %g (0x%08x)   D[r][1] = %g\n", what[r], h[r * 16], u, h[r * 16 + 1]); }
  return 0;
}
