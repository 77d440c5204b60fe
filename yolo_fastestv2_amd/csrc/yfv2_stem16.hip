// yfv2_stem16.hip - the stem (conv3x3 s2 3->24 + BN + ReLU + maxpool3x3 s2, model/backbone/shufflenetv2.py:74-80,
// 102-104; behaviour only) as an implicit GEMM on the f16 matrix cores, fp32 accuracy kept by splitting every operand
// EXACTLY-to-2^-24 into two fp16 terms ("fp16x3"): round 3's replacement of yfv2_stem.hip's 4x4x1 fp32-MFMA kernel for
// fp32 input (that kernel stays for the uint8 entry points and as the YFV2_BF6=0 plan).
//
// Why: gfx950's fp32 MFMA issues at the fp32 VECTOR rate and shares that datapath with the VALU (tools/ubench/shadow.hip),
// so the old kernel's time was MFMA + VALU summed (PMC: MFMA busy 55 %, issue-wait 40 %, 144 us = 0.49 of the HBM roofline
// on its 571 MB).  v_mfma_f32_16x16x32_f16 is 16x faster per MAC and runs beside the VALU.
//
// Arithmetic.  For a = fl32, h1 = RN16(a), h2 = RN16(a - fl32(h1)) (the subtraction is exact): |a - h1 - h2| <= 2^-24 |a|
// as long as |a| < 65504 and h2 is a normal fp16, and <= 2^-25 absolutely below that.  w x = w1 x1 + w1 x2 + w2 x1 + O(2^-22)
// w2 x2: three f16 x f16 products, each EXACT in fp32 (11 + 11 significant bits), accumulated in fp32 by the matrix core -
// the error per product is below the one rounding v_mfma_f32_4x4x1_f32 makes, measured in tests/test_stem16_host_model.py
// (numpy model of this arithmetic vs float64 and vs the fp32 conv).  The filter (BN scale folded) is split on the HOST and
// pre-scaled by a power of two 2^sw that puts its largest entry near 2^14 (exact; undone with the same power of two after
// pooling), so that the second terms of small weights stay normal numbers; the image is split in the kernel, unscaled:
// valid for |x| < 65504, far beyond any pixel scaling (the reference feeds [0, 1], test.py:38).
//
// GEMM shape.  D[channel][pixel] += W[channel][k] X[k][pixel], 16 x 16 x 32 per instruction: 24 channels = two channel tiles
// (the second half empty), K = 27 taps in 32 slots, N = 16 pixels.  A wave = ONE strip of 16 lanes' worth of pooled columns
// (15 new + a halo lane, as in yfv2_stem.hip) walking down a band of pooled rows; lane = (p = lane & 15: pooled column
// px = 15 strip + p, g = lane >> 4: K group).  Per conv row the wave multiplies TWO pixel tiles - tile E: the even conv
// columns 2px, tile O: the odd ones 2px + 1 - so that the horizontal max-pool needs ONE neighbour access
// (max(O[p-1], E[p], O[p])) and every lane ends up with pooled values to store.
//
// K slots.  Lane group g < 3 carries input channel g: one ALIGNED 16-byte load per input row gives columns 4px .. 4px+3 =
// v0..v3, the column 4px-1 = vm1 is the left neighbour's v3 (DPP row_shr:1, zero fill = the image's left padding).  With
// X0, X1, X2 = input rows 2y-1, 2y, 2y+1 of conv row y the eight slots of a lane are
//     tile E:  X0.v0 X0.v1 | X1.v0 X1.v1 | X0.vm1 X1.vm1 | X2.vm1 X2.v0        taps (ky,kx): (0,1)(0,2) (1,1)(1,2) (0,0)(1,0) (2,0)(2,1)
//     tile O:  X0.v2 X0.v3 | X1.v2 X1.v3 | X0.v1  X1.v1  | X2.v1  X2.v2        (same taps, columns shifted by two)
// - the first two register pairs are the packed conversion of a loaded pair as it stands.  That is 8 of a channel's 9
// taps; the ninth, (2,2), of all three channels lives in lane group 3, which runs the SAME instructions on other data:
// its X0, X1, X2 are row 2y+1 of channels 0, 1, 2, the last one loaded one column to the right, which puts (2,2) of the
// three channels into slots 1, 3 and 7 of both tiles; the filter image is zero in its other slots.
// No LDS, no barriers, nothing carried between conv rows except the pooled maxima.
#include "yfv2_internal.h"

typedef _Float16 yfv2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 yfv2_h2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ unsigned dpp_row_shr1_u(unsigned v) {   // lane l <- lane l-1 inside its 16-lane row, 0 at p = 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ float dpp_row_shr1_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
// two fp32 -> (h1, h2) packed pairs; h1 + h2 reproduces each value to 2^-24 (see header)
__device__ __forceinline__ void split2(float a, float b, unsigned& h1, unsigned& h2) {
  const f32x2 v = {a, b};
  const yfv2_h2 t1 = __builtin_convertvector(v, yfv2_h2);                       // v_cvt_pk_f16_f32 (RN)
  const f32x2 r = v - __builtin_convertvector(t1, f32x2);                       // exact
  const yfv2_h2 t2 = __builtin_convertvector(r, yfv2_h2);
  h1 = __builtin_bit_cast(unsigned, t1);
  h2 = __builtin_bit_cast(unsigned, t2);
}
// low halves / high halves / mixed picks of two packed registers
__device__ __forceinline__ unsigned pack_lo_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // {a.lo, b.lo}
__device__ __forceinline__ unsigned pack_hi_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // {a.hi, b.hi}
__device__ __forceinline__ unsigned pack_hi_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }   // {a.hi, b.lo}

struct Row16 {          // one input row of the lane, both terms: pairs (v0,v1), (v2,v3) and the left neighbour's v3 (in the HIGH half of m)
  unsigned p01[2], p23[2], m[2];
};
__device__ __forceinline__ void split_row(const f32x4 v, Row16& o) {
  split2(v[0], v[1], o.p01[0], o.p01[1]);
  split2(v[2], v[3], o.p23[0], o.p23[1]);
  o.m[0] = dpp_row_shr1_u(o.p23[0]);     // high half = the neighbour's v3 = this lane's column 4px-1
  o.m[1] = dpp_row_shr1_u(o.p23[1]);
}

}  // namespace

template <bool PPOUT>
__global__ __launch_bounds__(64, 3) void stem_h3_kernel(StemArgs a) {
  const int H = a.H, W = a.W, PH = H >> 2, PW = W >> 2;
  const int strips = (PW - 1 + 14) / 15;
  const int bands = PH / a.R;                      // a.R divides PH
  const int wpi = strips * bands;                  // waves per image
  // workgroup ids are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of waves, so that the waves of one
  // image (which share halo rows / columns and DRAM pages) sit behind one L2
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int band = wi % bands, strip = wi / bands;
  const int lane = threadIdx.x, p = lane & 15, g = lane >> 4;
  const int px = 15 * strip + p;
  const bool lvalid = px < PW;
  const int py0 = band * a.R;
  const bool st_ok = lvalid && (p > 0 || strip == 0);

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x + (size_t)b * 3 * H * W * 4), 0, 3 * H * W * 4, 0x00020000);
  const int rowb = W * 4;
  constexpr int OOB = (int)0x80000000;
  // byte offsets of the lane's three loads for conv row y: g < 3: rows 2y-1, 2y, 2y+1 of channel g; g = 3: row 2y+1 of
  // channels 0, 1, 2, the last one a column to the right
  const int colb = 4 * px * 4;
  const int o0 = g < 3 ? (g * H - 1) * rowb + colb : (0 * H + 1) * rowb + colb;
  const int o1 = g < 3 ? (g * H + 0) * rowb + colb : (1 * H + 1) * rowb + colb;
  const int o2 = g < 3 ? (g * H + 1) * rowb + colb : (2 * H + 1) * rowb + colb + 4;

  // filter: [tile 2][term 2][64 lanes][4 dwords] fp16 pairs, then shift * 2^sw [32], then 2^-sw
  yfv2_h8 wa[2][2];
  {
    const u32x4* wimg = reinterpret_cast<const u32x4*>(a.img16);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) wa[t][k] = __builtin_bit_cast(yfv2_h8, wimg[(t * 2 + k) * 64 + lane]);
  }
  const float* cst = a.img16 + 2 * 2 * 64 * 4;
  const f32x4 sh0 = *reinterpret_cast<const f32x4*>(cst + 4 * g), sh1 = *reinterpret_cast<const f32x4*>(cst + 16 + 4 * g);
  const float unscale = cst[32];

  auto load3 = [&](int y, f32x4 (&raw)[3]) {       // conv row y of this lane
    const int base = lvalid ? 2 * y * rowb : OOB;
    // the row above the image (y = 0, lane groups 0..2) is padding: its offset would land in the previous channel plane
    const int off0 = (y == 0 && g < 3) ? OOB : base + o0;
    raw[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off0, 0, 0));
    raw[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lvalid ? base + o1 : OOB, 0, 0));
    raw[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lvalid ? base + o2 : OOB, 0, 0));
  };

  // conv row -> horizontally pooled raw values hp[tile][4] = max(O[p-1], E[p], O[p]) (BN shift inside, pre-ReLU, x 2^sw)
  auto conv_row = [&](const f32x4 (&raw)[3], f32x4 (&hp)[2]) {
    Row16 x0, x1, x2;
    split_row(raw[0], x0); split_row(raw[1], x1); split_row(raw[2], x2);
    yfv2_h8 be[2], bo[2];                           // B operands of tile E / tile O, per term
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32x4 e = {x0.p01[k], x1.p01[k], pack_hi_hi(x0.m[k], x1.m[k]), pack_hi_lo(x2.m[k], x2.p01[k])};
      const u32x4 o = {x0.p23[k], x1.p23[k], pack_hi_hi(x0.p01[k], x1.p01[k]), pack_hi_lo(x2.p01[k], x2.p23[k])};
      be[k] = __builtin_bit_cast(yfv2_h8, e);
      bo[k] = __builtin_bit_cast(yfv2_h8, o);
    }
    f32x4 ae[2] = {sh0, sh1}, ao[2] = {sh0, sh1};
    // w1 x2, w2 x1, w1 x1 - smallest terms first; the four accumulators of a product are independent
#pragma unroll
    for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], be[1], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], bo[1], ao[t], 0, 0, 0); }
#pragma unroll
    for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], be[0], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], bo[0], ao[t], 0, 0, 0); }
#pragma unroll
    for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], be[0], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], bo[0], ao[t], 0, 0, 0); }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)     // 0 from the DPP at the image's left edge stands for the -inf padding: ReLU follows the pooling
        hp[t][e] = __builtin_fmaxf(__builtin_fmaxf(dpp_row_shr1_f(ao[t][e]), ae[t][e]), ao[t][e]);
  };

  // carried: hp of the odd conv row above the current pooled row (0 above the image: post-ReLU equivalent of the padding)
  f32x4 up[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  f32x4 bufA[3], bufB[3];
  if (py0 > 0) {
    load3(2 * py0 - 1, bufA);
    load3(2 * py0, bufB);
    conv_row(bufA, up);
  } else {
    load3(0, bufB);
  }
  float* __restrict__ ob = PPOUT ? a.out + (size_t)b * 24 * PH * PW + ((size_t)py0 * PW + (st_ok ? px : 0)) * 2
                                 : a.out + (((size_t)b * PH + py0) * PW + (st_ok ? px : 0)) * 24;
  const int ylast = (H >> 1) - 1;
#pragma unroll 1
  for (int t = 0; t < a.R; ++t) {
    const int y = 2 * (py0 + t);
    f32x4 h0[2], h1[2];
    load3(y + 1, bufA);                            // odd conv row of this pooled row: in flight during the even one
    conv_row(bufB, h0);
    load3(min(y + 2, ylast), bufB);                // next pooled row's even conv row (past the band: a re-read that stays in range)
    conv_row(bufA, h1);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = __builtin_fmaxf(__builtin_fmaxf(up[tt][e], h0[tt][e]), h1[tt][e]);
        o[e] = __builtin_fmaxf(m, 0.f) * unscale;  // ReLU, then the exact power of two back
      }
      up[tt] = h1[tt];
      if (st_ok && (tt == 0 || g < 2)) {           // channel tile 1 holds channels 16..23 in lane groups 0, 1
        if constexpr (PPOUT) {
          const int q = 8 * tt + 2 * g;            // channels 16 tt + 4 g .. +3 = pairs q, q + 1
          *reinterpret_cast<f32x2*>(ob + (size_t)q * PH * PW * 2) = (f32x2){o[0], o[1]};
          *reinterpret_cast<f32x2*>(ob + (size_t)(q + 1) * PH * PW * 2) = (f32x2){o[2], o[3]};
        } else {
          *reinterpret_cast<f32x4*>(ob + 16 * tt + 4 * g) = o;
        }
      }
    }
    ob += PPOUT ? (size_t)PW * 2 : (size_t)PW * 24;
  }
}

void yfv2_launch_stem16(const StemArgs& a, hipStream_t s) {
  StemArgs b = a;
  const int PH = a.H / 4, PW = a.W / 4;
  int nb = 8;                                       // bands per image: R must divide PH
  while (nb > 1 && (PH % nb || PH / nb < 4)) nb >>= 1;
  b.R = PH / nb;
  const int strips = (PW - 1 + 14) / 15;
  const dim3 grid(a.B * strips * nb);
  if (a.pp_out) hipLaunchKernelGGL((stem_h3_kernel<true>), grid, dim3(64), 0, s, b);
  else hipLaunchKernelGGL((stem_h3_kernel<false>), grid, dim3(64), 0, s, b);
}
