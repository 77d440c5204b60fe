// yfv2_stem.hip - gfx950 (CDNA4, wave64) stem of the Yolo-FastestV2 forward:
//   conv3x3 s2 (3->24) + BN + ReLU + maxpool3x3 s2, NCHW image in -> NHWC (B, H/4, W/4, 24) out
// on v_mfma_f32_4x4x1_16b_f32, one pixel per lane, no LDS, no barriers.
//
// Reference layers (read for behaviour only): model/backbone/shufflenetv2.py:74-80, 102-104.
//
// Why the 16-block 4x4x1 MFMA: the 16x16x4 tile pads M = 24 -> 32 and K = 27 -> 28 (14 MFMAs x 32
// cycles per 16 pixels = 1792 cycles per 64 pixels; an LDS-ring implicit-GEMM kernel built on it
// measured 274 us per 256 images).  In the 4x4x1 form a block is 4 adjacent lanes and
// D_blk[i][j] += A_blk[i] * B_blk[j]: with B = "the tap value of this lane's own pixel" and
// A = filter[4m + i][k] one instruction is 64 pixels x 4 output channels x 1 tap, so 6 x 27 = 162
// instructions x 8 cycles = 1296 cycles per 64 pixels with no padding at all, and the accumulator
// comes out as 4 consecutive channels of the lane's own pixel (= one 16-byte NHWC store).
//
// This file is compiled with -fno-honor-nans: max-pooling is v_max3_f32 chains on MFMA results and
// the default NaN-quieting canonicalisation (v_max x, x, x before every max) doubles their cost.
#include "yfv2_internal.h"
#include <utility>

// NB: pass the value through a by-value float parameter: __builtin_bit_cast applied directly to a vector
// ELEMENT lvalue (bit_cast(int, q[3])) reads element 0 with this compiler (seen in tools/ubench/dpp.hip).
__device__ __forceinline__ float yfv2_row_shr1(float v) {   // lane l <- lane l-1 inside its 16-lane row, 0 at r = 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}

// ----------------------------------------------------------------------------
// lane = pooled column (two conv columns per lane), 16-lane rows = independent strips
// ----------------------------------------------------------------------------
// A single wave cannot overlap VALU with the 8-cycle 4x4x1 MFMA (tools/ubench/mfma4.hip: MFMA + 2
// VALU = 18 cycles) and DPP wave shifts cost ~4 VALU slots (a first version with one conv column per
// lane and wave_shr/wave_shl pooling measured 192 us).  So the layout minimises VALU per MFMA: a lane owns pooled column px = 15*strip + r (r = lane & 15), i.e.
// conv columns 2px and 2px+1, whose five input columns 4px-1 .. 4px+3 are one aligned 16-byte buffer
// load plus the left neighbour's last value (full-rate DPP row_shr:1, whose zero fill at r = 0 is
// exactly the image's left padding).  Horizontal pooling needs only conv column 2px-1 = the left
// neighbour's second column: one row_shr per channel.  Lane r = 0 of strips > 0 is a halo lane (15
// new pooled columns per 16 lanes); the four 16-lane rows of a wave are four (strip, band) units of
// the same image, so the buffer resource (base = image, out-of-range lanes read zeros) is wave-uniform.
//
// Filter registers: the 4x4x1 MFMA can broadcast the A operand of ONE block to all 16 (cbsz = 4,
// abid = block).  Block j of filter register q holds W[4m + i][k] for (m, k) = q*16 + j, so the whole
// 24x27 filter is 11 VGPRs per lane instead of 162 and the (m, k) pair is picked by an immediate.
// Per pooled row and lane: 12 loads, 4 x 162 MFMAs, ~170 VALU, 6 stores.
struct StemRow { float v[3][5]; };                 // [ci][left, x0, x1, x2, x3]

template <int K, int COL>
__device__ __forceinline__ void stem_px_step(const float (&wq)[11], const f32x4 (&sh)[6], const StemRow& r0, const StemRow& r1,
                                             const StemRow& r2, f32x4 (&acc)[6]) {
  constexpr int ky = K / 9, ci = (K % 9) / 3, kx = K % 3;
  const StemRow& rw = ky == 0 ? r0 : (ky == 1 ? r1 : r2);
  const float bv = rw.v[ci][2 * COL + kx];
#define YFV2_STEM_MM(M)                                                                                                 \
  acc[M] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[((M) * 27 + K) >> 4], bv, K == 0 ? sh[M] : acc[M], 4, ((M) * 27 + K) & 15, 0)
  YFV2_STEM_MM(0); YFV2_STEM_MM(1); YFV2_STEM_MM(2); YFV2_STEM_MM(3); YFV2_STEM_MM(4); YFV2_STEM_MM(5);
#undef YFV2_STEM_MM
}
template <int COL, int... Ks>
__device__ __forceinline__ void stem_px_conv_impl(const float (&wq)[11], const f32x4 (&sh)[6], const StemRow& r0, const StemRow& r1,
                                                  const StemRow& r2, f32x4 (&acc)[6], std::integer_sequence<int, Ks...>) {
  (stem_px_step<Ks, COL>(wq, sh, r0, r1, r2, acc), ...);
}
// conv of one pixel: input rows r0 (ky = 0), r1, r2; COL = 0 (conv column 2px) or 1 (2px+1)
template <int COL>
__device__ __forceinline__ void stem_px_conv(const float (&wq)[11], const f32x4 (&sh)[6], const StemRow& r0, const StemRow& r1,
                                             const StemRow& r2, f32x4 (&acc)[6]) {
  stem_px_conv_impl<COL>(wq, sh, r0, r1, r2, acc, std::make_integer_sequence<int, 27>{});
}

// PPOUT: output layout pair planes [12][PH][PW][2] (for s2px_kernel) or NHWC.
// U8IN: the input is the camera/decoder layout uint8 (B,H,W,3) (test.py:34-38 before its permute/float()/255): a lane's four
// columns x three channels are 12 contiguous bytes = ONE load per input row instead of three 16-byte ones and a
// quarter of the traffic; bytes become floats with v_cvt_f32_ubyteN and the 1/255 is folded into the filter.
template <bool PPOUT, bool U8IN>
__global__ __launch_bounds__(64, 1) void stem_px_kernel(StemArgs a) {
  const int H = a.H, W = a.W, PH = H >> 2, PW = W >> 2;
  const int strips = (PW - 1 + 14) / 15;
  const int bands = PH / a.R;                      // a.R divides PH
  const int units = strips * bands, wpi = (units + 3) >> 2;
  // workgroup ids are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of waves, so
  // the waves of one image (which share halo rows/columns and DRAM pages) sit behind one L2
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);   // keep the buffer resource in SGPRs
  const int lane = threadIdx.x, r = lane & 15;
  const int uid = wi * 4 + (lane >> 4);
  const int band = uid % bands, strip = uid / bands;   // (adjacent strips per wave instead measured the same)
  const int px = 15 * strip + r;
  const bool lvalid = uid < units && px < PW;
  const int py0 = band * a.R;
  const bool st_ok = lvalid && (r > 0 || strip == 0);

  constexpr int ESZ = U8IN ? 3 : 4;               // bytes per input column in a row (u8: 3 interleaved channels; fp32: one plane)
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x + (size_t)b * 3 * H * W * (U8IN ? 1 : 4)), 0,
                                                                    3 * H * W * (U8IN ? 1 : 4), 0x00020000);
  const int rowb = W * ESZ;                        // bytes per input row
  // voffset of (row 4py0, column 4px); invalid lanes sit beyond num_records and read zeros
  int voff = lvalid ? (4 * py0 * W + 4 * px) * ESZ : (int)0x80000000;

  float wq[11];
  f32x4 shiftv[6];
  const float* wimg = U8IN ? a.img_u8 : a.img;
#pragma unroll
  for (int q = 0; q < 11; ++q) wq[q] = wimg[q * 64 + lane];
  {
    const f32x4* sh = reinterpret_cast<const f32x4*>(wimg + 11 * 64);
#pragma unroll
    for (int m = 0; m < 6; ++m) shiftv[m] = sh[m];
  }

  // one input row of the lane's four columns: fp32 = three 16-byte plane pieces, u8 = 12 interleaved bytes
  struct RawRow { f32x4 f[U8IN ? 1 : 3]; u32x3 u; };
  auto load4 = [&](int vo, RawRow (&raw)[4]) {     // input rows vo .. vo+3 (all three channels)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if constexpr (U8IN) {
        raw[rr].u = __builtin_amdgcn_raw_buffer_load_b96(rsrc, vo, rr * rowb, 0);
      } else {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
          raw[rr].f[ci] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, (ci * H + rr) * rowb, 0));
      }
    }
  };
  auto unpack = [&](const RawRow& raw, StemRow& o) {
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      float v[4];
      if constexpr (U8IN) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {              // byte 3c + ci of the 12
          const unsigned w = (3 * c + ci) >> 2 == 0 ? raw.u[0] : ((3 * c + ci) >> 2 == 1 ? raw.u[1] : raw.u[2]);
          const int k = (3 * c + ci) & 3;
          v[c] = (float)((w >> (8 * k)) & 0xffu);   // -> v_cvt_f32_ubyte{k}
        }
      } else {
        v[0] = raw.f[ci][0]; v[1] = raw.f[ci][1]; v[2] = raw.f[ci][2]; v[3] = raw.f[ci][3];
      }
      o.v[ci][0] = yfv2_row_shr1(v[3]);
      o.v[ci][1] = v[0]; o.v[ci][2] = v[1]; o.v[ci][3] = v[2]; o.v[ci][4] = v[3];
    }
  };

  // carried state: input row 4py-1 and the raw (pre-ReLU) conv row 2py-1 of both columns; max commutes
  // with ReLU, which is applied once to the pooled value, and 0 stands in for the -inf padding
  StemRow carry;
  f32x4 cv0[6], cv1[6];
  RawRow bufA[4], bufB[4];
  {
    RawRow raw[4];
    load4(py0 > 0 ? voff - 4 * rowb : (int)0x80000000, raw);   // rows 4py0-4 .. 4py0-1 (row -4 unused)
    load4(voff, bufA);                                          // first pooled row's inputs fly during the halo conv
    StemRow r1, r2;
    unpack(raw[1], r1); unpack(raw[2], r2); unpack(raw[3], carry);
    stem_px_conv<0>(wq, shiftv, r1, r2, carry, cv0);
    stem_px_conv<1>(wq, shiftv, r1, r2, carry, cv1);
    if (py0 == 0) {
#pragma unroll
      for (int m = 0; m < 6; ++m) { cv0[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; cv1[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
  }
  float* __restrict__ ob = PPOUT ? a.out + (size_t)b * 24 * PH * PW + ((size_t)py0 * PW + (st_ok ? px : 0)) * 2
                                 : a.out + (((size_t)b * PH + py0) * PW + (st_ok ? px : 0)) * 24;

  f32x4 pend[6];
  auto compute = [&](const RawRow (&cur)[4]) {
    StemRow r0, r1, r2, r3;
    unpack(cur[0], r0); unpack(cur[1], r1); unpack(cur[2], r2); unpack(cur[3], r3);
    f32x4 A[6], B[6], m0[6];
    stem_px_conv<0>(wq, shiftv, carry, r0, r1, A);
    stem_px_conv<0>(wq, shiftv, r1, r2, r3, B);
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) { m0[m][e] = __builtin_fmaxf(__builtin_fmaxf(cv0[m][e], A[m][e]), B[m][e]); cv0[m][e] = B[m][e]; }
    stem_px_conv<1>(wq, shiftv, carry, r0, r1, A);
    stem_px_conv<1>(wq, shiftv, r1, r2, r3, B);
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(cv1[m][e], A[m][e]), B[m][e]);
        cv1[m][e] = B[m][e];
        const float lf = yfv2_row_shr1(m1);
        const float hm = __builtin_fmaxf(__builtin_fmaxf(m0[m][e], m1), lf);
        o[e] = __builtin_amdgcn_fmed3f(hm, 0.f, __builtin_inff());   // ReLU
      }
      pend[m] = o;
    }
    carry = r3;
  };
  // The pooled row is stored at the START of the next step, before that step's prefetch loads: gfx9
  // has one in-order vmcnt for loads and stores, so a store issued after the prefetch would have to be
  // acknowledged before the prefetched data may be used (measured: ~1 us of stall per pooled row).
  bool have = false;
  auto flush = [&]() {
    if (have) {
      if (st_ok) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          if constexpr (PPOUT) {
            *reinterpret_cast<f32x2*>(ob + (size_t)(2 * m) * PH * PW * 2) = (f32x2){pend[m][0], pend[m][1]};
            *reinterpret_cast<f32x2*>(ob + (size_t)(2 * m + 1) * PH * PW * 2) = (f32x2){pend[m][2], pend[m][3]};
          } else {
            *reinterpret_cast<f32x4*>(ob + 4 * m) = pend[m];
          }
        }
      }
      ob += PPOUT ? (size_t)PW * 2 : (size_t)PW * 24;
    }
    have = true;
    __builtin_amdgcn_sched_barrier(0);
  };

  // One wave per SIMD: the next pooled row's loads fly during this one's MFMAs.  The prefetch is
  // unconditional (a conditional one makes the compiler's waitcnt merge wait for the loads it has just
  // issued); past the band's end it re-reads the last rows, which stay in range.
  const int vlast = voff + (a.R - 1) * 4 * rowb;
  int t = 0;
  for (; t + 1 < a.R; t += 2) {
    flush();
    voff += 4 * rowb; load4(voff, bufB);
    __builtin_amdgcn_sched_barrier(0);
    compute(bufA);
    flush();
    voff = min(voff + 4 * rowb, vlast); load4(voff, bufA);
    __builtin_amdgcn_sched_barrier(0);
    compute(bufB);
  }
  if (t < a.R) { flush(); compute(bufA); }
  flush();
}

void yfv2_launch_stem(const StemArgs& a, hipStream_t s) {
  if (a.img16) { yfv2_launch_stem16(a, s); return; }   // the f16-matrix-core kernels (yfv2_stem16.hip): fp32 NCHW or uint8 HWC input
  StemArgs b = a;
  const int PH = a.H / 4, PW = a.W / 4;
  int nb = 8;                                       // bands per image: R must divide PH
  while (nb > 1 && (PH % nb || PH / nb < 4)) nb >>= 1;
  b.R = PH / nb;
  const int strips = (PW - 1 + 14) / 15;
  const dim3 grid(a.B * ((strips * nb + 3) / 4));
  if (a.u8_in) {
    if (a.pp_out) YFV2_LAUNCH((stem_px_kernel<true, true>), grid, dim3(64), 0, s, b);
    else YFV2_LAUNCH((stem_px_kernel<false, true>), grid, dim3(64), 0, s, b);
  } else {
    if (a.pp_out) YFV2_LAUNCH((stem_px_kernel<true, false>), grid, dim3(64), 0, s, b);
    else YFV2_LAUNCH((stem_px_kernel<false, false>), grid, dim3(64), 0, s, b);
  }
}
