// bf16x6.hip - can the bf16 matrix cores evaluate an fp32 pointwise conv at fp32 accuracy, and how much faster?
// x and w are split EXACTLY into three bf16 terms each (truncation: 8 + 8 + 8 significant bits = fp32's 24), six of the
// nine cross products (everything above 2^-23 relative) go through v_mfma_f32_16x16x32_bf16 with fp32 accumulation:
//   MFMA 1:  A = {w.hi, w.mid}   B = {x.hi, x.hi }     MFMA 2:  A = {w.lo, w.hi}   B = {x.hi, x.mid}
//   MFMA 3:  A = {w.mid, w.hi}   B = {x.mid, x.lo}
// (the 8 k-slots of a lane group = the lane's 4 channels x 2 terms).  Compared against v_mfma_f32_16x16x4_f32 on the same
// data: max error of both vs a float64 host result, and cycles per (16 channels x MT x NT) step of one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x6 bf16x6.hip && ./bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
constexpr int MT = 5, NT = 2, K16 = 12, M = 16 * MT, K = 16 * K16, NPX = 16 * NT;

__device__ __forceinline__ unsigned pack_hi(float a, float b) {   // (bf16 trunc of b) << 16 | bf16 trunc of a   (element 0 in the low half)
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u); }

// A fragments, fp32: [mt][s][lane] float4;   split: [mt][s][lane] {lo01, lo23, hi01, hi23, mid01, mid23} (6 dwords)
__global__ __launch_bounds__(64) void k_f32(const f32x4* Ag, const f32x4* Xg, f32x4* D, int reps, long long* cyc) {
  const int lane = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4* A = reinterpret_cast<f32x4*>(lds);
  f32x4* X = A + MT * K16 * 64;
  for (int i = lane; i < MT * K16 * 64; i += 64) A[i] = Ag[i];
  for (int i = lane; i < NT * K16 * 64; i += 64) X[i] = Xg[i];
  __syncthreads();
  f32x4 acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r)
#pragma unroll 2
    for (int s = 0; s < K16; ++s) {
      f32x4 af[MT], b[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = A[(mt * K16 + s) * 64 + lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = X[(nt * K16 + s) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
    }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[0] = t1 - t0;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) D[(mt * NT + nt) * 64 + lane] = acc[mt][nt];
}

__global__ __launch_bounds__(64) void k_bf16x6(const unsigned* A6g, const f32x4* Xg, f32x4* D, int reps, long long* cyc) {
  const int lane = threadIdx.x;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* A6 = reinterpret_cast<unsigned*>(lds);
  f32x4* X = reinterpret_cast<f32x4*>(lds + MT * K16 * 64 * 8);
  for (int i = lane; i < MT * K16 * 64 * 8; i += 64) A6[i] = A6g[i];
  for (int i = lane; i < NT * K16 * 64; i += 64) X[i] = Xg[i];
  __syncthreads();
  f32x4 acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r)
#pragma unroll 2
    for (int s = 0; s < K16; ++s) {
      // A: two aligned quads per lane, {lo01 lo23 hi01 hi23} and {mid01 mid23 hi01 hi23}; B: three quads built in registers
      //   MFMA a: A{lo,hi}  x B{hi,lo}     MFMA b: A{mid,hi} x B{hi,hi}     MFMA c: A{mid,hi} x B{mid,mid}
      u32x4 aq0[MT], aq1[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const u32x4* p = reinterpret_cast<const u32x4*>(A6 + ((size_t)(mt * K16 + s) * 64 + lane) * 8);
        aq0[mt] = p[0]; aq1[mt] = p[1];
      }
      u32x4 ba[NT], bb[NT], bc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 x = X[(nt * K16 + s) * 64 + lane];
        float r1[4], r2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { r1[c] = x[c] - trunc_bf16(x[c]); r2[c] = r1[c] - trunc_bf16(r1[c]); }
        ba[nt] = (u32x4){pack_hi(x[0], x[1]), pack_hi(x[2], x[3]), pack_hi(r2[0], r2[1]), pack_hi(r2[2], r2[3])};
        bb[nt] = (u32x4){pack_hi(x[0], x[1]), pack_hi(x[2], x[3]), pack_hi(x[0], x[1]), pack_hi(x[2], x[3])};
        bc[nt] = (u32x4){pack_hi(r1[0], r1[1]), pack_hi(r1[2], r1[3]), pack_hi(r1[0], r1[1]), pack_hi(r1[2], r1[3])};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq0[mt]), __builtin_bit_cast(bf16x8, ba[nt]), acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq1[mt]), __builtin_bit_cast(bf16x8, bc[nt]), acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, aq1[mt]), __builtin_bit_cast(bf16x8, bb[nt]), acc[mt][nt], 0, 0, 0);
    }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[0] = t1 - t0;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) D[(mt * NT + nt) * 64 + lane] = acc[mt][nt];
}

static unsigned short bf16_trunc(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf16_val(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  std::vector<float> W(M * K), Xp(NPX * K);
  srand(1);
  for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
  for (auto& v : Xp) v = (rand() / (float)RAND_MAX) * 6.0f;           // post-ReLU-like activations
  std::vector<float> A((size_t)MT * K16 * 64 * 4), X((size_t)NT * K16 * 64 * 4);
  std::vector<unsigned> A6((size_t)MT * K16 * 64 * 8);
  for (int mt = 0; mt < MT; ++mt) for (int s = 0; s < K16; ++s) for (int l = 0; l < 64; ++l) {
    unsigned short hi[4], mid[4], lo[4];
    for (int c = 0; c < 4; ++c) {
      const float w = W[(16 * mt + (l & 15)) * K + 16 * s + 4 * (l >> 4) + c];
      A[((mt * K16 + s) * 64 + l) * 4 + c] = w;
      hi[c] = bf16_trunc(w); const float r1 = w - bf16_val(hi[c]);
      mid[c] = bf16_trunc(r1); const float r2 = r1 - bf16_val(mid[c]);
      lo[c] = bf16_trunc(r2);
    }
    unsigned* p = &A6[((size_t)(mt * K16 + s) * 64 + l) * 8];
    p[0] = lo[0] | (unsigned)lo[1] << 16; p[1] = lo[2] | (unsigned)lo[3] << 16;
    p[2] = hi[0] | (unsigned)hi[1] << 16; p[3] = hi[2] | (unsigned)hi[3] << 16;
    p[4] = mid[0] | (unsigned)mid[1] << 16; p[5] = mid[2] | (unsigned)mid[3] << 16; p[6] = p[2]; p[7] = p[3];   // quad 0 {lo,hi}, quad 1 {mid,hi}
  }
  for (int nt = 0; nt < NT; ++nt) for (int s = 0; s < K16; ++s) for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c)
    X[((nt * K16 + s) * 64 + l) * 4 + c] = Xp[(16 * nt + (l & 15)) * K + 16 * s + 4 * (l >> 4) + c];
  float *dA, *dX, *dD; unsigned* dA6; long long* dc;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dD, MT * NT * 64 * 16); hipMalloc(&dA6, A6.size() * 4); hipMalloc(&dc, 8);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dA6, A6.data(), A6.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> ref(M * NPX);
  for (int m = 0; m < M; ++m) for (int n = 0; n < NPX; ++n) { double a = 0; for (int k = 0; k < K; ++k) a += (double)W[m * K + k] * Xp[n * K + k]; ref[m * NPX + n] = a; }
  for (int which = 0; which < 2; ++which) {
    std::vector<float> D(MT * NT * 64 * 4);
    long long c1 = 0, c20 = 0;
    for (int reps : {1, 20}) {
      if (which == 0) { hipFuncSetAttribute((const void*)k_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), (MT + NT) * K16 * 64 * 16, 0, (const f32x4*)dA, (const f32x4*)dX, (f32x4*)dD, reps, dc); }
      else { hipFuncSetAttribute((const void*)k_bf16x6, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(k_bf16x6, dim3(1), dim3(64), MT * K16 * 64 * 32 + NT * K16 * 64 * 16, 0, (const unsigned*)dA6, (const f32x4*)dX, (f32x4*)dD, reps, dc); }
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      if (reps == 1) { c1 = c; hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost); } else c20 = c;
    }
    double err = 0, mag = 0;
    for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      const int m = 16 * mt + 4 * (l >> 4) + r, n = 16 * nt + (l & 15);
      err = fmax(err, fabs(D[((mt * NT + nt) * 64 + l) * 4 + r] - ref[m * NPX + n])); mag = fmax(mag, fabs(ref[m * NPX + n]));
    }
    printf("%-8s max |err| vs float64 %.3e (largest output %.1f)   %.1f cycles per 16-channel step of %d x %d tiles (one wave)\n",
           which ? "bf16x6" : "f32", err, mag, (double)(c20 - c1) / (19.0 * K16), MT, NT);
  }
  return 0;
}
