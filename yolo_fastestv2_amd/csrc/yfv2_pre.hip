// yfv2_pre.hip - the pre-process in front of the path: bilinear resize of uint8 HWC frames
// to the network input size, the arithmetic of `cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR)`
// (test.py:35, utils/datasets.py:107) for 8-bit images: OpenCV's fixed-point path (imgproc/resize.cpp, pinned by the
// reference at opencv_python 4.2.0.34): per output column  fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx),
// coefficients round_half_even((1 - fx) * 2048), round_half_even(fx * 2048) as int16 (sx < 0 or sx >= src_w - 1:
// clamp, fx = 0); the same per output row except that the two row INDICES are clipped instead of the weight; horizontal
// pass S = p[sx] * a0 + p[sx + 1] * a1 (int32), vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// OpenCV itself is not in this image: the tests bit-compare this kernel with a numpy restatement of the same published
// algorithm - "parity unpinned" against cv2 proper.
//
// HBM-bound byte work.  One workgroup = one output row of one frame: the two source rows it blends are staged in LDS
// with aligned 4-byte loads (source row starts are arbitrary byte addresses), the blend reads LDS bytes, the output row
// is assembled in LDS and leaves as aligned 4-byte stores (width % 4 == 0, so every output row starts 4-byte aligned).
// Built with -ffp-contract=off: the coefficient arithmetic must round like the separate C operations it restates.
#include "yfv2_internal.h"

struct RowCoef { int s0, s1, c0, c1; };

__device__ __forceinline__ RowCoef yfv2_axis_coef(int d, double scale, int n, bool clamp_weight) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  RowCoef r;
  if (clamp_weight) {            // columns: the weight is dropped at the borders
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    r.s0 = s; r.s1 = min(s + 1, n - 1);
  } else {                       // rows: both indices are clipped, the weights stay
    r.s0 = min(max(s, 0), n - 1); r.s1 = min(max(s + 1, 0), n - 1);
  }
  r.c0 = __float2int_rn((1.f - f) * 2048.f);
  r.c1 = __float2int_rn(f * 2048.f);
  return r;
}

__global__ __launch_bounds__(256) void resize_u8_kernel(ResizeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.H, oy = blockIdx.x - b * a.H;
  const RowCoef ry = yfv2_axis_coef(oy, a.scale_y, a.SH, false);
  const int row_bytes = a.SW * 3;
  const int slot = (row_bytes + 3 + 3) & ~3;          // staged bytes per source row (head misalignment + tail round-up)
  unsigned char* R0 = smem;
  unsigned char* R1 = smem + slot;
  unsigned char* OUT = smem + 2 * slot;
  // ---- stage the two source rows: aligned dwords covering [g, g + row_bytes)
  const size_t total = (size_t)a.B * a.SH * row_bytes;
  const size_t g0 = ((size_t)b * a.SH + ry.s0) * row_bytes, g1 = ((size_t)b * a.SH + ry.s1) * row_bytes;
  const uintptr_t base = reinterpret_cast<uintptr_t>(a.src);
  const int m0 = (int)((base + g0) & 3), m1 = (int)((base + g1) & 3);
  const int nd0 = (m0 + row_bytes + 3) >> 2, nd1 = (m1 + row_bytes + 3) >> 2;
  auto stage = [&](unsigned char* dst, size_t g, int mis, int nd) {
    const unsigned char* p = a.src + g - mis;        // 4-byte aligned
    for (int i = tid; i < nd; i += 256) {
      // the first / last dword of the very first / last row of the buffer may reach outside the allocation: bytes there
      const bool inside = (i > 0 || mis == 0 || g >= (size_t)mis) && (g - mis + 4 * (size_t)i + 4 <= total);
      unsigned v;
      if (inside) v = *reinterpret_cast<const unsigned*>(p + 4 * i);
      else {
        v = 0;
        for (int k = 0; k < 4; ++k) {
          const long long off = (long long)g - mis + 4ll * i + k;
          if (off >= 0 && (size_t)off < total) v |= (unsigned)a.src[off] << (8 * k);
        }
      }
      *reinterpret_cast<unsigned*>(dst + 4 * i) = v;
    }
  };
  stage(R0, g0, m0, nd0);
  stage(R1, g1, m1, nd1);
  __syncthreads();
  // ---- blend: one thread per output pixel (3 channels)
  const unsigned char* r0 = R0 + m0;
  const unsigned char* r1 = R1 + m1;
  for (int ox = tid; ox < a.W; ox += 256) {
    const RowCoef rx = yfv2_axis_coef(ox, a.scale_x, a.SW, true);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int S0 = (int)r0[rx.s0 * 3 + c] * rx.c0 + (int)r0[rx.s1 * 3 + c] * rx.c1;
      const int S1 = (int)r1[rx.s0 * 3 + c] * rx.c0 + (int)r1[rx.s1 * 3 + c] * rx.c1;
      const int v = (((ry.c0 * (S0 >> 4)) >> 16) + ((ry.c1 * (S1 >> 4)) >> 16) + 2) >> 2;
      OUT[ox * 3 + c] = (unsigned char)min(max(v, 0), 255);
    }
  }
  __syncthreads();
  unsigned* drow = reinterpret_cast<unsigned*>(a.dst + ((size_t)b * a.H + oy) * a.W * 3);
  for (int i = tid; i < (a.W * 3) >> 2; i += 256) drow[i] = reinterpret_cast<const unsigned*>(OUT)[i];
}

size_t yfv2_resize_lds_bytes(int SW, int W) { return 2 * (size_t)((SW * 3 + 6) & ~3) + (size_t)W * 3; }

void yfv2_launch_resize(const ResizeArgs& a, hipStream_t s) {
  const size_t lds = yfv2_resize_lds_bytes(a.SW, a.W);
  static std::atomic<unsigned long long> lds_ok0{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&resize_u8_kernel), lds_ok0);
  hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)(a.B * a.H)), dim3(256), lds, s, a);
}
