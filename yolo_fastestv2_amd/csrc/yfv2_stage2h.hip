// yfv2_stage2h.hip - ShuffleNetV2 stage 2's stride-1 blocks (44x44 maps, 24-channel branch: pw1+BN+ReLU -> dw3x3+BN ->
// pw2+BN+ReLU; model/backbone/shufflenetv2.py:19-32,48-51,57-63, behaviour only) with both pointwise convs on the f16
// matrix cores - the "fp16x3" arithmetic of yfv2_stem16.hip: every operand split into two fp16 terms whose sum
// reproduces it to 2^-24, three exact f16 x f16 products per MAC, fp32 accumulation.  Round 3's replacement of
// s1px_kernel (yfv2_stage2.hip: one pixel per lane on the 4x4x1 fp32 MFMA, 300 MFMAs of 8 cycles + ~260 VALU per 64 pixels
// on ONE shared datapath: 35-37 us per block at 256 images for 23 us of HBM traffic; kept as the YFV2_BF6=0 plan).
//
// Layout of the work: a wave = one strip of 16 columns x a band of rows of one image; lane = (l = lane & 15: column
// 14 strip + l, g = lane >> 4: channel group).  v_mfma_f32_16x16x32_f16 with the filter as A and 16 pixels as B leaves lane
// (l, g) with output channels 4g..4g+3 of channel tile 0 and 16+4g..16+4g+3 of tile 1 (lane groups 0, 1 only: 24
// channels) of ITS pixel - and those eight values, as K slots 8g..8g+7, are exactly the B operand of the NEXT pointwise
// conv: the chain pw1 -> depthwise -> pw2 needs no data movement between lanes except the depthwise's own horizontal
// neighbours (DPP row shifts of column sums, zero fill = the conv's zero padding at the image edge).
//   * loads: the eight input channels of a lane are four 8-byte pairs of stage 2's pair planes (yfv2_stage2.hip header;
//     which pairs, in which buffer, and where the results go: per-lane byte offsets packed by the host), converted pair by
//     pair (v_cvt_pk_f16_f32) into B-operand dwords as they stand;
//   * BN: scale folded into the filters / taps, shift = the accumulators' initial value; ReLU = v_med3 against a per-lane
//     0 / +inf limit that also zeroes pw1's output outside the image (the depthwise's padding);
//   * depthwise taps: four to a register across the lanes of a quad (all quads of a lane group alike), applied with
//     v_fmac_f32_dpp quad_perm broadcasts as in yfv2_stage2.hip - 18 registers instead of 72;
//   * powers of two: filters carry 2^sw (largest entry near 2^14), activations are scaled by 2^4 before the split (fp16's
//     absolute floor 2^-25 -> 2^-29; valid for |x| < 4094, an order of magnitude above anything this network produces:
//     stage-2 activations are < 10 with the COCO weights and with random-init ones), all undone exactly: pw1's inside the
//     taps, pw2's after its ReLU.
#include "yfv2_internal.h"
#include <atomic>
#include <utility>

typedef _Float16 yfv2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 yfv2_h2 __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float row_shr1(float v) {   // lane l <- lane l-1 inside its 16-lane row, 0 at l = 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_shl1(float v) {   // lane l <- lane l+1 inside its 16-lane row, 0 at l = 15
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
// tap K of the lane's quad times v (see yfv2_stage2.hip for the two gfx9 hazards the asm forms guard against)
template <int K>
__device__ __forceinline__ float quad_mul(float tap4, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tap4), K * 0x55, 0xf, 0xf, false)) * v;
}
#define YFV2_QP(n) "quad_perm:[%" #n ",%" #n ",%" #n ",%" #n "] row_mask:0xf bank_mask:0xf\n\t"
// a0 += t3*u + t6*w; a1 += t4*u + t7*w; a2 += t5*u + t8*w   (rows dy = 1, 2 of a stride-1 window, all three dx)
template <int K3, int K4, int K5, int K6, int K7, int K8>
__device__ __forceinline__ void quad_fmac6(float& a0, float& a1, float& a2, float t3, float t4, float t5, float t6, float t7, float t8,
                                           float u, float w) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %9 " YFV2_QP(11) "v_fmac_f32_dpp %1, %4, %9 " YFV2_QP(12) "v_fmac_f32_dpp %2, %5, %9 " YFV2_QP(13)
      "v_fmac_f32_dpp %0, %6, %10 " YFV2_QP(14) "v_fmac_f32_dpp %1, %7, %10 " YFV2_QP(15) "v_fmac_f32_dpp %2, %8, %10 " YFV2_QP(16)
      : "+v"(a0), "+v"(a1), "+v"(a2)
      : "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(t8), "v"(u), "v"(w), "n"(K3), "n"(K4), "n"(K5), "n"(K6), "n"(K7), "n"(K8));
}
// s += ta*v0; q += tb*v1; s += tc*v1   (one vertical tap row of a stride-2 window: dx = 1, 0, 2)
template <int KA, int KB, int KC>
__device__ __forceinline__ void quad_fmac3(float& s, float& q, float ta, float tb, float tc, float v0, float v1) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %5 " YFV2_QP(7) "v_fmac_f32_dpp %1, %3, %6 " YFV2_QP(8) "v_fmac_f32_dpp %0, %4, %6 " YFV2_QP(9)
      : "+v"(s), "+v"(q) : "v"(ta), "v"(tb), "v"(tc), "v"(v0), "v"(v1), "n"(KA), "n"(KB), "n"(KC));
}
template <int KA>
__device__ __forceinline__ void quad_fmac1(float& s, float ta, float v) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 " YFV2_QP(3) : "+v"(s) : "v"(ta), "v"(v), "n"(KA));
}
#undef YFV2_QP
__device__ __forceinline__ void dpp_src_ready(float& v) { asm volatile("s_nop 1" : "+v"(v)); }

// (a, b) -> packed fp16 pairs h1, h2 with h1 + h2 = (a, b) to 2^-24
__device__ __forceinline__ void split2(f32x2 v, unsigned& h1, unsigned& h2) {
  const yfv2_h2 t1 = __builtin_convertvector(v, yfv2_h2);                       // v_cvt_pk_f16_f32 (RN)
  const f32x2 r = v - __builtin_convertvector(t1, f32x2);                       // exact
  const yfv2_h2 t2 = __builtin_convertvector(r, yfv2_h2);
  h1 = __builtin_bit_cast(unsigned, t1);
  h2 = __builtin_bit_cast(unsigned, t2);
}
// 24 -> 24 pointwise conv of 16 pixels: in = the lane's eight K slots as four pairs (already carrying their 2^4), init = BN
// shift (scaled) of the lane's output channels; w[tile][term].  w1 x2, w2 x1, w1 x1: smallest terms first.
__device__ __forceinline__ void pw_h3(const yfv2_h8 (&w)[2][2], const f32x2 (&in)[4], const f32x4 (&init)[2], f32x4 (&acc)[2], Yfv2Watch& watch) {
  u32x4 b1, b2;
#pragma unroll
  for (int k = 0; k < 4; ++k) { unsigned h1, h2; split2(in[k], h1, h2); b1[k] = h1; b2[k] = h2; }
  const yfv2_h8 x1 = __builtin_bit_cast(yfv2_h8, b1), x2 = __builtin_bit_cast(yfv2_h8, b2);
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t][0], x2, init[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t][1], x1, acc[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t][0], x1, acc[t], 0, 0, 0);
  watch.see(acc[0][0]);   // range guard (yfv2_internal.h): an operand beyond fp16's range makes every output channel of its pixel NaN
}

}  // namespace

// image offsets (floats): W1 | W2 [tile 2][term 2][64 lanes][4 dwords] | taps [18 regs][64 lanes] | sh1 * 2^(sw1+4) [32] |
// bias2 * 2^(sw2+4) [32] | 2^-(sw2+4) | pad | per-lane byte offsets: src [4][64], dst [4][64] (0x80000000 = no such pair)
constexpr int S1H_W1 = 0, S1H_W2 = 1024, S1H_TAPS = 2048, S1H_CST = 3200, S1H_OFFS = 3272;

__global__ __launch_bounds__(64, 3) void s1h_kernel(S1PxArgs a) {
  const int H = a.H, W = a.W;
  const int nstrips = a.nstrips, nb = a.nb, R = a.R;
  const int wpi = nstrips * nb;
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);   // XCD-contiguous (yfv2_stage2.hip)
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int strip = wi % nstrips, band = wi / nstrips;
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const int x = 14 * strip + l;
  const bool xok = x < W;
  const bool st_lane = xok && (l > 0 || strip == 0) && (l < 15 || strip == nstrips - 1);
  const int y0 = band * R, y1 = min(H, y0 + R);
  constexpr int OOB = (int)0x80000000;

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.act + (size_t)b * a.img_stride), 0, a.num_records, 0x00020000);
  const float* img = a.img16;
  yfv2_h8 w1[2][2], w2[2][2];
  {
    const u32x4* q = reinterpret_cast<const u32x4*>(img);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        w1[t][k] = __builtin_bit_cast(yfv2_h8, q[(S1H_W1 / 4) + (t * 2 + k) * 64 + lane]);
        w2[t][k] = __builtin_bit_cast(yfv2_h8, q[(S1H_W2 / 4) + (t * 2 + k) * 64 + lane]);
      }
  }
  Yfv2Watch watch;
  float tq[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) tq[q] = img[S1H_TAPS + q * 64 + lane];
  const f32x4 sh1[2] = {*reinterpret_cast<const f32x4*>(img + S1H_CST + 4 * g), *reinterpret_cast<const f32x4*>(img + S1H_CST + 16 + 4 * g)};
  const f32x4 bi2[2] = {*reinterpret_cast<const f32x4*>(img + S1H_CST + 32 + 4 * g), *reinterpret_cast<const f32x4*>(img + S1H_CST + 48 + 4 * g)};
  const float unscale2 = img[S1H_CST + 64];
  int soff[4], doff[4];
  {
    const int* po = reinterpret_cast<const int*>(img + S1H_OFFS);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int so = po[k * 64 + lane], dd = po[(4 + k) * 64 + lane];
      soff[k] = (xok && so != OOB) ? so + x * 8 : OOB;
      doff[k] = (st_lane && dd != OOB) ? dd + x * 8 : OOB;
    }
  }
  const int rowb = W * 8;                          // bytes per row of one pair plane

  auto load_row = [&](int r, f32x2 (&v)[4]) {      // the lane's four input pairs of row r, times 2^4 (zeros outside the image)
    const bool rok = r >= 0 && r < H;              // wave-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (rok && soff[k] != OOB) ? soff[k] + r * rowb : OOB, 0, 0));
  };

  f32x2 cur[4], nxt[4];
  float tA[8], tB[8], tC[8];
  f32x2 pend[4];
  int pend_row = -1;
#pragma unroll
  for (int k = 0; k < 4; ++k) pend[k] = (f32x2){0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) { tA[c] = 0.f; tB[c] = 0.f; tC[c] = 0.f; }
  load_row(y0 - 1, cur);

  // step j: pw1 of row r = y0-1+j into tn; for j >= 2 the block's output row r-1 from the t rows (tp2, tp1, tn)
  auto step = [&](int j, const float (&tp2)[8], const float (&tp1)[8], float (&tn)[8]) {
    const int r = y0 - 1 + j;
    // last step's outputs go out first, then this step's prefetch (one in-order vmcnt for loads and stores)
    {
      const bool pok = pend_row >= 0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pend[k]), rsrc, (pok && doff[k] != OOB) ? doff[k] + pend_row * rowb : OOB, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_row(r + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[2];
    {
      f32x2 in[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) in[k] = cur[k] * 16.0f;
      pw_h3(w1, in, sh1, acc, watch);
    }
    const float lim = (xok && r >= 0 && r < H) ? __builtin_inff() : 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) tn[c] = __builtin_amdgcn_fmed3f(acc[c >> 2][c & 3], 0.f, lim);   // ReLU, and 0 outside the image

    // depthwise 3x3 (taps carry BN scale, pw1's 2^-(sw1+4) and pw2's 2^4; its BN shift sits in pw2's bias): vertical taps
    // are per-lane FMAs over the three t rows, the dx = 0 / dx = 2 column sums move one lane right / left
    f32x2 d[4];
    auto dw_ch = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
#define YFV2_TQ(t) tq[(c * 9 + (t)) >> 2]
#define YFV2_TK(t) ((c * 9 + (t)) & 3)
      float a0 = quad_mul<YFV2_TK(0)>(YFV2_TQ(0), tp2[c]), a1 = quad_mul<YFV2_TK(1)>(YFV2_TQ(1), tp2[c]), a2 = quad_mul<YFV2_TK(2)>(YFV2_TQ(2), tp2[c]);
      quad_fmac6<YFV2_TK(3), YFV2_TK(4), YFV2_TK(5), YFV2_TK(6), YFV2_TK(7), YFV2_TK(8)>(a0, a1, a2, YFV2_TQ(3), YFV2_TQ(4), YFV2_TQ(5), YFV2_TQ(6),
                                                                                       YFV2_TQ(7), YFV2_TQ(8), tp1[c], tn[c]);
#undef YFV2_TQ
#undef YFV2_TK
      dpp_src_ready(a2);
      d[c >> 1][c & 1] = a1 + row_shr1(a0) + row_shl1(a2);
    };
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) { (dw_ch(std::integral_constant<int, Cs>{}), ...); }(std::make_integer_sequence<int, 8>{});
    pw_h3(w2, d, bi2, acc, watch);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pend[k][0] = __builtin_fmaxf(acc[k >> 1][2 * (k & 1)], 0.f) * unscale2;
      pend[k][1] = __builtin_fmaxf(acc[k >> 1][2 * (k & 1) + 1], 0.f) * unscale2;
    }
    pend_row = (j >= 2 && r - 1 < y1) ? r - 1 : -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
  };

  const int nsteps = R + 2;
  for (int j = 0; j < nsteps; j += 3) {
    step(j, tB, tC, tA);
    if (j + 1 < nsteps) step(j + 1, tC, tA, tB);
    if (j + 2 < nsteps) step(j + 2, tA, tB, tC);
  }
  {
    const bool pok = pend_row >= 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pend[k]), rsrc, (pok && doff[k] != OOB) ? doff[k] + pend_row * rowb : OOB, 0, 0);
  }
  watch.report(a.nonfinite);
}

void yfv2_launch_s1h(const S1PxArgs& a0, hipStream_t s) {
  S1PxArgs a = a0;
  a.nstrips = a.W <= 16 ? 1 : (a.W - 2 + 13) / 14;
  a.nb = a.H >= 16 ? 4 : 1;   // (2, 3, 5, 6, 8, 11 bands measured 35-41 us against 31-33; a memory-only build of this kernel - same
                              // 8-byte loads and stores, no arithmetic - takes 29-30 us: the launch is bound by its access pattern)
                              // (next row's loads issued BEFORE the pending stores, and two rows of look-ahead on top of that: 30-32 us
                              // each, same box as 30-32 - neither the stores' place in the in-order vmcnt queue nor bytes in flight limit it)
  a.R = (a.H + a.nb - 1) / a.nb;
  a.nb = (a.H + a.R - 1) / a.R;
  YFV2_LAUNCH(s1h_kernel, dim3(a.B * a.nstrips * a.nb), dim3(64), 0, s, a);
}

// ============================================================================
// stage2.0: the stride-2 block 24 -> 48 (88x88 -> 44x44) in ONE launch that reads its input once
// ============================================================================
// Reference (shufflenetv2.py:19-44,52-55): proj = pw(dw3x3 s2(x)), main = pw2(dw3x3 s2(pw1(x))), out = cat(proj, main).
// Rounds 1-2 ran the two branches as two wave ROLES / two kernels (yfv2_stage2.hip: ~250 registers of state per branch in
// the one-pixel-per-lane form), both reading the stem's output: 580 MB of HBM traffic for 285 MB of input + output, 108-117
// us.  In the layout above a lane holds eight channels instead of 24, so one wave carries BOTH branches:
//   lane (l, g): output column ox = 15 strip + l (lane 0 of an inner strip is a halo lane, as in the stem kernels) = input
//   columns 2ox, 2ox+1: ONE 16-byte load per pair plane and input row brings both columns of two channels; tile E = the
//   even input columns, tile O = the odd ones.  Per input row: pw1 of both tiles (main), ReLU / padding mask, and both
//   branches' vertical taps into the column sums S (dx = 1, 2) and Q (dx = 0: it belongs to the right neighbour's window
//   and travels there with one DPP row shift) of the current output row; the odd input row is also the next output row's
//   dy = 0 row (its pw1 output and raw values are carried).  Per output row: the two depthwise results go through their
//   pointwise convs (proj pw, pw2) and out to stage 2's pair planes: channel positions 0..15 of either branch are its eight
//   whole pairs, positions 16..23 pair up ACROSS the branches (proj in element 0, main in element 1: the slot map of
//   yfv2_stage2_channel), so every store is a full 8-byte pair.
// image offsets (floats): W1 | Wproj | W2 (1024 each) | taps main [18][64] | taps proj [18][64] | sh1, bias_proj, bias2 (x 2^(sw+4))
// [3][32] | 2^-(swp+4), 2^-(sw2+4) | pad | per-lane byte offsets: load [4][64], store proj-whole [2], main-whole [2], mixed [4] x [64]
constexpr int S2H_W1 = 0, S2H_WP = 1024, S2H_W2 = 2048, S2H_TM = 3072, S2H_TP = 4224, S2H_CST = 5376, S2H_OFFS = 5480;

__global__ __launch_bounds__(64, 2) void s2h_kernel(S2PxArgs a) {
  const int IH = a.IH, IW = a.IW, OH = IH >> 1, OW = IW >> 1;
  const int nstrips = a.nstrips, nb = a.nb, R = a.R;
  const int wpi = nstrips * nb;
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int strip = wi % nstrips, band = wi / nstrips;
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const int ox = 15 * strip + l;
  const bool xok = ox < OW;
  const bool st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R, y1 = min(OH, y0 + R);
  constexpr int OOB = (int)0x80000000;

  __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * a.in_stride), 0, a.in_records, 0x00020000);
  __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(a.act + (size_t)b * a.out_stride), 0, a.out_records, 0x00020000);
  const float* img = a.img16;
  yfv2_h8 w1[2][2], wp[2][2], w2[2][2];
  {
    const u32x4* q = reinterpret_cast<const u32x4*>(img);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        w1[t][k] = __builtin_bit_cast(yfv2_h8, q[(S2H_W1 / 4) + (t * 2 + k) * 64 + lane]);
        wp[t][k] = __builtin_bit_cast(yfv2_h8, q[(S2H_WP / 4) + (t * 2 + k) * 64 + lane]);
        w2[t][k] = __builtin_bit_cast(yfv2_h8, q[(S2H_W2 / 4) + (t * 2 + k) * 64 + lane]);
      }
  }
  Yfv2Watch watch;
  float tm[18], tp[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) { tm[q] = img[S2H_TM + q * 64 + lane]; tp[q] = img[S2H_TP + q * 64 + lane]; }
  const f32x4 sh1[2] = {*reinterpret_cast<const f32x4*>(img + S2H_CST + 4 * g), *reinterpret_cast<const f32x4*>(img + S2H_CST + 16 + 4 * g)};
  const f32x4 bip[2] = {*reinterpret_cast<const f32x4*>(img + S2H_CST + 32 + 4 * g), *reinterpret_cast<const f32x4*>(img + S2H_CST + 48 + 4 * g)};
  const f32x4 bi2[2] = {*reinterpret_cast<const f32x4*>(img + S2H_CST + 64 + 4 * g), *reinterpret_cast<const f32x4*>(img + S2H_CST + 80 + 4 * g)};
  const float unscale_p = img[S2H_CST + 96], unscale_2 = img[S2H_CST + 97];
  // input: the stem's [IH][IW][24] (a pixel's 96 bytes in one run; yfv2_stem16.hip) - this lane's eight channel positions are the
  // 16-byte pieces g (all lane groups: the stem's channel tile 0) and 4 + g (lane groups 0, 1: tile 1) of two adjacent pixels
  int loff[2], soff[8];
  loff[0] = xok ? 2 * ox * 96 + 16 * g : OOB;
  loff[1] = (xok && g < 2) ? 2 * ox * 96 + 64 + 16 * g : OOB;
  constexpr int coff = 96;                         // the lane's second pixel
  {
    const int* po = reinterpret_cast<const int*>(img + S2H_OFFS);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int v = po[(4 + k) * 64 + lane]; soff[k] = (st_lane && v != OOB) ? v + ox * 8 : OOB; }
  }
  const int irowb = IW * 96, orowb = OW * 8;

  // one input row: X[2t + c] = the four channel positions 4t .. 4t+3 of column 2ox + c, times 2^4 later - four 16-byte loads
  auto load_row = [&](int iy, f32x4 (&X)[4]) {
    const bool rok = iy >= 0 && iy < IH;           // wave-uniform
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        X[2 * t + c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rok && loff[t] != OOB) ? loff[t] + iy * irowb : OOB, c * coff, 0));   // (an out-of-range voffset stays out of range)
  };
  // the two branches' depthwise inputs of one input row: raw values x 2^4 (proj) and relu(pw1) x 2^(sw1+4) (main, 0 outside the image)
  auto columns = [&](const f32x4 (&X)[4], float lim, float (&xe)[8], float (&xo)[8], float (&te)[8], float (&to)[8]) {
    f32x2 ine[4], ino[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 ve = X[2 * t] * 16.0f, vo = X[2 * t + 1] * 16.0f;
      ine[2 * t] = (f32x2){ve[0], ve[1]}; ine[2 * t + 1] = (f32x2){ve[2], ve[3]};
      ino[2 * t] = (f32x2){vo[0], vo[1]}; ino[2 * t + 1] = (f32x2){vo[2], vo[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) { xe[4 * t + e] = ve[e]; xo[4 * t + e] = vo[e]; }
    }
    f32x4 ae[2], ao[2];
    pw_h3(w1, ine, sh1, ae, watch);
    pw_h3(w1, ino, sh1, ao, watch);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      te[c] = __builtin_amdgcn_fmed3f(ae[c >> 2][c & 3], 0.f, lim);
      to[c] = __builtin_amdgcn_fmed3f(ao[c >> 2][c & 3], 0.f, lim);
    }
  };
#define YFV2_TQ(T, c, t) T[((c) * 9 + (t)) >> 2]
#define YFV2_TK(c, t) (((c) * 9 + (t)) & 3)
  // vertical tap row DY of a 3x3 stride-2 window: S += w[dy][1]*v0 + w[dy][2]*v1, Q += w[dy][0]*v1
  auto acc_row = [&](auto dyc, const float (&T)[18], const float (&v0)[8], const float (&v1)[8], float (&S)[8], float (&Q)[8]) {
    constexpr int DY = decltype(dyc)::value;
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) {
      ((DY == 0 ? (void)(S[Cs] = quad_mul<YFV2_TK(Cs, 1)>(YFV2_TQ(T, Cs, 1), v0[Cs]), Q[Cs] = quad_mul<YFV2_TK(Cs, 0)>(YFV2_TQ(T, Cs, 0), v1[Cs]),
                         quad_fmac1<YFV2_TK(Cs, 2)>(S[Cs], YFV2_TQ(T, Cs, 2), v1[Cs]))
                : (void)quad_fmac3<YFV2_TK(Cs, DY * 3 + 1), YFV2_TK(Cs, DY * 3), YFV2_TK(Cs, DY * 3 + 2)>(
                      S[Cs], Q[Cs], YFV2_TQ(T, Cs, DY * 3 + 1), YFV2_TQ(T, Cs, DY * 3), YFV2_TQ(T, Cs, DY * 3 + 2), v0[Cs], v1[Cs])), ...);
    }(std::make_integer_sequence<int, 8>{});
  };

  f32x4 X[4], Y[4];
  float cxe[8], cxo[8], cte[8], cto[8];            // the odd input row above the current output row (dy = 0): raw and pw1'd
  {
    const int iy = 2 * y0 - 1;
    load_row(iy, X);
    load_row(iy + 1, Y);
    columns(X, (xok && iy >= 0) ? __builtin_inff() : 0.f, cxe, cxo, cte, cto);
    load_row(iy + 2, X);
  }
  const float limx = xok ? __builtin_inff() : 0.f;
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
    float Sm[8], Qm[8], Sp[8], Qp[8], xe[8], xo[8], te[8], to[8];
    acc_row(std::integral_constant<int, 0>{}, tm, cte, cto, Sm, Qm);
    acc_row(std::integral_constant<int, 0>{}, tp, cxe, cxo, Sp, Qp);
    columns(Y, limx, xe, xo, te, to);              // even input row 2oy: dy = 1
    acc_row(std::integral_constant<int, 1>{}, tm, te, to, Sm, Qm);
    acc_row(std::integral_constant<int, 1>{}, tp, xe, xo, Sp, Qp);
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 2, Y);                       // next step's even row
    __builtin_amdgcn_sched_barrier(0);
    columns(X, limx, cxe, cxo, cte, cto);          // odd input row 2oy+1: dy = 2, and the next output row's dy = 0
    acc_row(std::integral_constant<int, 2>{}, tm, cte, cto, Sm, Qm);
    acc_row(std::integral_constant<int, 2>{}, tp, cxe, cxo, Sp, Qp);
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 3, X);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 dm[4], dp[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dpp_src_ready(Qm[c]); dpp_src_ready(Qp[c]);
      dm[c >> 1][c & 1] = Sm[c] + row_shr1(Qm[c]);
      dp[c >> 1][c & 1] = Sp[c] + row_shr1(Qp[c]);
    }
    f32x4 am[2], ap[2];
    pw_h3(wp, dp, bip, ap, watch);
    pw_h3(w2, dm, bi2, am, watch);
    const bool rowok = oy < y1;                    // wave-uniform
    f32x4 op[2], om[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) { op[t][e] = __builtin_fmaxf(ap[t][e], 0.f) * unscale_p; om[t][e] = __builtin_fmaxf(am[t][e], 0.f) * unscale_2; }
    const int ro = oy * orowb;
    auto st = [&](int k, float v0, float v1) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v0, v1}), rout, (rowok && soff[k] != OOB) ? soff[k] + ro : OOB, 0, 0);
    };
    st(0, op[0][0], op[0][1]); st(1, op[0][2], op[0][3]);          // proj: positions 4g..4g+3 = two whole pairs
    st(2, om[0][0], om[0][1]); st(3, om[0][2], om[0][3]);          // main: the same
#pragma unroll
    for (int e = 0; e < 4; ++e) st(4 + e, op[1][e], om[1][e]);     // positions 16 + 4g + e: proj | main halves of a mixed pair
  }
  watch.report(a.nonfinite);
#undef YFV2_TQ
#undef YFV2_TK
}

void yfv2_launch_s2h(const S2PxArgs& a0, hipStream_t s) {
  S2PxArgs a = a0;
  const int OW = a.IW / 2, OH = a.IH / 2;
  a.nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  a.nb = OH >= 16 ? 4 : 1;
  a.R = (OH + a.nb - 1) / a.nb;
  a.nb = (OH + a.R - 1) / a.R;
  YFV2_LAUNCH(s2h_kernel, dim3(a.B * a.nstrips * a.nb), dim3(64), 0, s, a);
}

// ============================================================================
// stem + stage2.0 in ONE wave (round 5): conv3x3 s2 3 -> 24 + BN + ReLU + maxpool3x3 s2 (yfv2_stem16.hip's arithmetic) feeding
// the stride-2 block above without the [H/4][W/4][24] tensor between them ever leaving the registers
// ============================================================================
// Why: plain streaming costs 130 pJ per byte above idle on these boxes (tools/traffic_energy_probe.py), the two
// launches move 856 MB of which 380 are the stem's output written and read back: ~50 mJ of a 850 mJ pipelined step that runs at
// the package's power cap, and 117 + 72 us on one stream for what a single pass over 381 MB in + 95 MB out could do.
// How: the matrix core's D layout of the stem - lane (p, g): channels 16 t + 4 g .. + 3 of ITS pooled pixel - is, record for
// record, what s2h_kernel loads per lane: X[2 t + c] = channel positions 4 t .. 4 t + 3 of pooled column 2 ox + c.  So a lane of
// this kernel owns TWO adjacent pooled columns A = 2 ox, B = 2 ox + 1 (input columns 8 ox .. 8 ox + 7: two aligned 16-byte loads
// per input row and channel), runs the stem's implicit GEMM for both (four pixel tiles per conv row instead of two: even / odd
// conv columns of A and of B), pools, and hands the result to the stride-2 block's code above as if it had been loaded.  What a
// lane needs from a neighbour: the input column left of A (= the left lane's B, one DPP row shift of its converted pair, as in
// the stem) and the odd conv column left of A for the horizontal max (one DPP row shift of the left lane's accumulator).  Lane 0
// of an inner strip has no left lane: its column A is wrong, its column B is right - and B is all lane 1 needs from it (the
// depthwise's dx = 0 tap); lane 0 stores nothing (the halo lane of s2h_kernel).  Pooled rows are produced in exactly the order
// the block consumes them (2 y0 - 1, 2 y0, ..): none is ever held beside another.  Every value goes through the same
// instructions on the same operands in the same order as in stem_h3_kernel + s2h_kernel: the results are bit-identical
// (tests/test_gpu_parity.py compares the two plans).
namespace {
struct FCol { unsigned p01[2], p23[2], m[2]; };   // a 16-byte run of one input row, both fp16 terms: (v0, v1), (v2, v3); m: high half = the column left of v0
__device__ __forceinline__ unsigned f_dpp_shr1_u(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ float f_dpp_shr1_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); }
__device__ __forceinline__ int f_f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float f_i2f(int v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ void f_split2(float a, float b, unsigned& h1, unsigned& h2) {   // the stem's split: x 2^8 first (yfv2_stem16.hip)
  const f32x2 v = (f32x2){a, b} * 256.0f;
  const yfv2_h2 t1 = __builtin_convertvector(v, yfv2_h2);
  const f32x2 r = v - __builtin_convertvector(t1, f32x2);
  const yfv2_h2 t2 = __builtin_convertvector(r, yfv2_h2);
  h1 = __builtin_bit_cast(unsigned, t1);
  h2 = __builtin_bit_cast(unsigned, t2);
}
__device__ __forceinline__ unsigned f_hi_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // {a.hi, b.hi}
__device__ __forceinline__ unsigned f_hi_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040302u); }   // {a.hi, b.lo}
__device__ __forceinline__ void f_split(const f32x4 v, FCol& o) {
  f_split2(v[0], v[1], o.p01[0], o.p01[1]);
  f_split2(v[2], v[3], o.p23[0], o.p23[1]);
}
// one pooled column's two conv columns (tile E: even, tile O: odd) of one conv row: rows x0 (carried), x1, x2 - stem_h3_kernel's K slots
__device__ __forceinline__ void f_conv_col(const FCol& x0, const FCol& x1, const FCol& x2, const yfv2_h8 (&wa)[2][2], const f32x4 sh0, const f32x4 sh1,
                                           f32x4 (&ae)[2], f32x4 (&ao)[2]) {
  yfv2_h8 be[2], bo[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const u32x4 e = {x0.p01[k], x1.p01[k], f_hi_hi(x0.m[k], x1.m[k]), f_hi_lo(x2.m[k], x2.p01[k])};
    const u32x4 o = {x0.p23[k], x1.p23[k], f_hi_hi(x0.p01[k], x1.p01[k]), f_hi_lo(x2.p01[k], x2.p23[k])};
    be[k] = __builtin_bit_cast(yfv2_h8, e);
    bo[k] = __builtin_bit_cast(yfv2_h8, o);
  }
  ae[0] = sh0; ae[1] = sh1; ao[0] = sh0; ao[1] = sh1;
#pragma unroll
  for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], be[1], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], bo[1], ao[t], 0, 0, 0); }
#pragma unroll
  for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], be[0], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], bo[0], ao[t], 0, 0, 0); }
#pragma unroll
  for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], be[0], ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], bo[0], ao[t], 0, 0, 0); }
}
}  // namespace


// front2_kernel: that wave program at TWO waves per SIMD.  Its first form (front_kernel, round 5, removed in round 6) held ~400
// registers (both kernels' filters, depthwise taps and BatchNorm constants, two sets of input buffers) and ran one wave per SIMD,
// where nothing covered its dependency and LDS stalls and the matrix core's results had to be fetched from the accumulation
// registers one by one: 157 us.  Here a workgroup of four waves shares ONE copy of the filters, taps and constants in LDS (26 KB) and
// every wave reads what it needs right where it needs it (a compiler-level memory barrier in front of each read keeps the
// compiler from hoisting the loop-invariant reads back into registers); the input look-ahead is ONE set of eight 16-byte buffers
// refilled conv row by conv row right behind their use.  Same instructions on the same operands in the same order: bit-identical.
constexpr int F2_WA = 0, F2_S2 = 1024, F2_TAPS = F2_S2 + 3072, F2_CST = F2_TAPS + 36 * 64, F2_SCST = F2_CST + 104, F2_FLOATS = F2_SCST + 40;
// U8: the input is uint8 (B,H,W,3) (yfv2_forward_u8 / yfv2_detect_u8) and the stem part is stem_h3u_kernel's (yfv2_stem16.hip): a pixel
// is exactly ONE fp16 term (two products per MAC), HWC puts the three channels of a lane's four columns into one aligned 12-byte
// load - a lane's two pooled columns are 24 consecutive bytes - lane group g picks its channel's bytes with v_perm_b32, lane group 3
// picks tap (2,2)'s values out of its own load (no ds_bpermute), the 1 / 255 rides in the final unscale.  Bit-identical to
// stem_h3u_kernel + s2h_kernel.
typedef unsigned yfv2_u3x __attribute__((ext_vector_type(3)));
namespace {
struct FCol8 { unsigned p01, p23, m; };
__device__ __forceinline__ unsigned f_u8pair(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned v = __builtin_amdgcn_perm(hi, lo, sel) | 0x64006400u;
  const yfv2_h2 h = __builtin_bit_cast(yfv2_h2, v) - (yfv2_h2){(_Float16)1024.0f, (_Float16)1024.0f};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void f_split_u8(const yfv2_u3x d, unsigned sel_a, unsigned sel_b, FCol8& o) {
  o.p01 = f_u8pair(d[1], d[0], sel_a);
  o.p23 = f_u8pair(d[2], d[1], sel_b);
}
__device__ __forceinline__ void f_conv_col8(const FCol8& x0, const FCol8& x1, const FCol8& x2, const yfv2_h8 (&wa)[2][2], const f32x4 sh0, const f32x4 sh1,
                                            f32x4 (&ae)[2], f32x4 (&ao)[2]) {
  const u32x4 e = {x0.p01, x1.p01, f_hi_hi(x0.m, x1.m), f_hi_lo(x2.m, x2.p01)};
  const u32x4 o = {x0.p23, x1.p23, f_hi_hi(x0.p01, x1.p01), f_hi_lo(x2.p01, x2.p23)};
  const yfv2_h8 be = __builtin_bit_cast(yfv2_h8, e), bo = __builtin_bit_cast(yfv2_h8, o);
  ae[0] = sh0; ae[1] = sh1; ao[0] = sh0; ao[1] = sh1;
#pragma unroll
  for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], be, ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], bo, ao[t], 0, 0, 0); }
#pragma unroll
  for (int t = 0; t < 2; ++t) { ae[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], be, ae[t], 0, 0, 0); ao[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], bo, ao[t], 0, 0, 0); }
}
}  // namespace
template <bool U8>
__global__ __launch_bounds__(256, 2) void front2_kernel(FrontArgs fa) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const S2PxArgs& a = fa.s2;
  {   // the shared images -> LDS: stem filter | W1 WP W2 | taps transposed to [lane][tm 18 | tp 18] | stage2.0's constants | the stem's
    const int tid = threadIdx.x;
    const f32x4* sw = reinterpret_cast<const f32x4*>(fa.img_stem);
    const f32x4* sq = reinterpret_cast<const f32x4*>(a.img16);
    f32x4 t0 = sw[tid], t1 = sq[tid], t2 = sq[256 + tid], t3 = sq[512 + tid];
    float tp9[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { const int e = tid + 256 * j, q = e >> 6, ln = e & 63; tp9[j] = a.img16[S2H_TM + q * 64 + ln]; }   // e = q * 64 + lane over tm then tp (tp follows tm: 18 * 64 floats)
    const float c0 = tid < 104 ? a.img16[S2H_CST + tid] : 0.f, c1 = tid < 33 ? fa.img_stem[1024 + (U8 ? 36 : 0) + tid] : 0.f;   // (the uint8 constants follow the fp32 ones)
    reinterpret_cast<f32x4*>(lds + F2_WA)[tid] = t0;
    reinterpret_cast<f32x4*>(lds + F2_S2)[tid] = t1; reinterpret_cast<f32x4*>(lds + F2_S2)[256 + tid] = t2; reinterpret_cast<f32x4*>(lds + F2_S2)[512 + tid] = t3;
#pragma unroll
    for (int j = 0; j < 9; ++j) { const int e = tid + 256 * j, q = e >> 6, ln = e & 63; lds[F2_TAPS + ln * 36 + q] = tp9[j]; }
    if (tid < 104) lds[F2_CST + tid] = c0;
    if (tid < 33) lds[F2_SCST + tid] = c1;
  }
  __syncthreads();
  const int IH = a.IH, IW = a.IW, OH = IH >> 1, OW = IW >> 1;      // IH x IW = the pooled map the stem produces (H/4 x W/4)
  const int H = fa.H, W = fa.W;
  const int nstrips = a.nstrips, nb = a.nb, R = a.R;
  const int wpi = nstrips * nb;
  const int nwg = gridDim.x;
  const int wgid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int wid = __builtin_amdgcn_readfirstlane(wgid * 4 + (int)(threadIdx.x >> 6));
  if (wid >= a.B * wpi) return;                    // (after the barrier above: the workgroup's last waves may have no unit)
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int strip = wi % nstrips, band = wi / nstrips;
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  const int ox = 15 * strip + l;
  const bool xok = ox < OW;
  const bool st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R, y1 = min(OH, y0 + R);
  constexpr int OOB = (int)0x80000000;

  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)fa.x + (size_t)b * 3 * H * W * (U8 ? 1 : 4)), 0, 3 * H * W * (U8 ? 1 : 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(a.act + (size_t)b * a.out_stride), 0, a.out_records, 0x00020000);
  // ---- stage2.0's state (s2h_kernel)
  const float* img = a.img16;
  Yfv2Watch watch;
  // filters, taps and constants: read from the workgroup's LDS copy at the point of use
  auto ld_filter = [&](int off, yfv2_h8 (&w)[2][2]) {   // [tile 2][term 2][64 lanes][4 dwords]
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) w[t][k] = __builtin_bit_cast(yfv2_h8, reinterpret_cast<const u32x4*>(lds + off)[(t * 2 + k) * 64 + lane]);
  };
  auto ld_taps = [&](float (&tm)[18], float (&tp)[18]) {
    asm volatile("" ::: "memory");
    const f32x4* q = reinterpret_cast<const f32x4*>(lds + F2_TAPS + lane * 36);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const f32x4 v = q[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int i = 4 * j + e; if (i < 18) tm[i] = v[e]; else tp[i - 18] = v[e]; }
    }
  };
  auto ld_c2 = [&](int off, f32x4 (&c)[2]) {             // two channel tiles of a per-channel constant: entries 4 g .. and 16 + 4 g ..
    asm volatile("" ::: "memory");
    c[0] = *reinterpret_cast<const f32x4*>(lds + off + 4 * g); c[1] = *reinterpret_cast<const f32x4*>(lds + off + 16 + 4 * g);
  };
  const float unscale_p = img[S2H_CST + 96], unscale_2 = img[S2H_CST + 97];
  int soff[8];
  {
    const int* po = reinterpret_cast<const int*>(img + S2H_OFFS);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int v = po[(4 + k) * 64 + lane]; soff[k] = (st_lane && v != OOB) ? v + ox * 8 : OOB; }
  }
  const int orowb = OW * 8;
  // ---- the stem's state (stem_h3_kernel): filter [tile 2][term 2][64 lanes][4 dwords], shift x 2^(sw+8) [32], 2^-(sw+8)
  const float* cst = fa.img_stem + 2 * 2 * 64 * 4 + (U8 ? 36 : 0);
  const float sunscale = cst[32];
  const int rowb = U8 ? W * 3 : W * 4;
  // fp32: lane groups 0..2 = input channel g, columns 8 ox .. 8 ox + 7; uint8: every lane group loads the same 24 bytes (columns 8 ox .. + 7, three channels each)
  const int chan_off = U8 ? (xok ? 24 * ox : OOB) : ((xok && g < 3) ? g * H * rowb + 8 * ox * 4 : OOB);
  // uint8 byte selectors (stem_h3u_kernel): lane group g < 3: channel g of columns 0..3; lane group 3: X1 = (ch0, ch1) of columns 1 | 3, X2 = (ch2, 0)
  constexpr unsigned Z = 0x0c;
  const unsigned sel1a = g < 3 ? (unsigned)g | (Z << 8) | ((unsigned)(g + 3) << 16) | (Z << 24) : 3u | (Z << 8) | (4u << 16) | (Z << 24);
  const unsigned sel1b = g < 3 ? (unsigned)(g + 2) | (Z << 8) | ((unsigned)(g + 5) << 16) | (Z << 24) : 5u | (Z << 8) | (6u << 16) | (Z << 24);
  const unsigned sel2a = g < 3 ? sel1a : 5u | (Z << 8) | (Z << 16) | (Z << 24);
  const unsigned sel2b = g < 3 ? sel1b : 7u | (Z << 8) | (Z << 16) | (Z << 24);
  const int src0 = (0 * 16 + l) * 4, src1 = (1 * 16 + l) * 4, src2 = (2 * 16 + l) * 4;   // ds_bpermute byte addresses of lanes (l, 0..2)
  const int hlast = H - 1;

  // input rows 4 r .. 4 r + 3 (the two conv rows of pooled row r), columns A | B: eight 16-byte loads
  struct InSet { f32x4 a[4], b[4]; };                 // (uint8: the first three dwords of each hold the 12-byte load)
  auto load_ab = [&](int off, f32x4& va, f32x4& vb) {
    if constexpr (U8) {
      const yfv2_u3x ua = __builtin_bit_cast(yfv2_u3x, __builtin_amdgcn_raw_buffer_load_b96(rx, off, 0, 0));
      const yfv2_u3x ub = __builtin_bit_cast(yfv2_u3x, __builtin_amdgcn_raw_buffer_load_b96(rx, off, 12, 0));
      va = __builtin_bit_cast(f32x4, (u32x4){ua[0], ua[1], ua[2], 0u}); vb = __builtin_bit_cast(f32x4, (u32x4){ub[0], ub[1], ub[2], 0u});
    } else {
      va = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      vb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 16, 0));
    }
  };
  auto issue_half = [&](int r, InSet& s, int hf) {
#pragma unroll
    for (int i = 2 * hf; i < 2 * hf + 2; ++i) {
      const int row = min(4 * r + i, hlast);
      load_ab(chan_off != OOB ? chan_off + row * rowb : OOB, s.a[i], s.b[i]);
    }
  };
  auto issue = [&](int r, InSet& s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(4 * r + i, hlast);          // (a row past the image: a re-read that stays in range, never used)
      load_ab(chan_off != OOB ? chan_off + row * rowb : OOB, s.a[i], s.b[i]);
    }
  };
  // one conv row of both pooled columns from the carried rows (cA, cB) and the fresh rows r1 (2c), r2 (2c + 1) -> horizontally
  // pooled raw values (BN shift inside, pre-ReLU, x 2^(sw+8)); cA, cB <- the split row 2c + 1
  auto conv_row2 = [&](FCol& cA, FCol& cB, f32x4 r1A, f32x4 r2A, f32x4 r1B, f32x4 r2B, f32x4 (&hA)[2], f32x4 (&hB)[2]) {
    if constexpr (U8) {
      // (the carried columns live in the first term's slots of cA / cB: p01[0], p23[0], m[0])
      const u32x4 q1A = __builtin_bit_cast(u32x4, r1A), q2A = __builtin_bit_cast(u32x4, r2A), q1B = __builtin_bit_cast(u32x4, r1B), q2B = __builtin_bit_cast(u32x4, r2B);
      const yfv2_u3x d1A = {q1A[0], q1A[1], q1A[2]}, d2A = {q2A[0], q2A[1], q2A[2]}, d1B = {q1B[0], q1B[1], q1B[2]}, d2B = {q2B[0], q2B[1], q2B[2]};
      FCol8 x0A = {cA.p01[0], cA.p23[0], cA.m[0]}, x0B = {cB.p01[0], cB.p23[0], cB.m[0]}, x1A, x2A, x1B, x2B;
      f_split_u8(g == 3 ? d2A : d1A, sel1a, sel1b, x1A); f_split_u8(d2A, sel2a, sel2b, x2A);
      f_split_u8(g == 3 ? d2B : d1B, sel1a, sel1b, x1B); f_split_u8(d2B, sel2a, sel2b, x2B);
      x1A.m = f_dpp_shr1_u(x1B.p23); x2A.m = f_dpp_shr1_u(x2B.p23);
      x1B.m = x1A.p23; x2B.m = x2A.p23;
      f32x4 aeA[2], aoA[2], aeB[2], aoB[2];
      {
        yfv2_h8 wa[2][2]; f32x4 ssh[2];
        ld_filter(F2_WA, wa); ld_c2(F2_SCST, ssh);
        f_conv_col8(x0A, x1A, x2A, wa, ssh[0], ssh[1], aeA, aoA);
        f_conv_col8(x0B, x1B, x2B, wa, ssh[0], ssh[1], aeB, aoB);
      }
      cA.p01[0] = x2A.p01; cA.p23[0] = x2A.p23; cA.m[0] = x2A.m; cB.p01[0] = x2B.p01; cB.p23[0] = x2B.p23; cB.m[0] = x2B.m;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hA[t][e] = __builtin_fmaxf(__builtin_fmaxf(f_dpp_shr1_f(aoB[t][e]), aeA[t][e]), aoA[t][e]);
          hB[t][e] = __builtin_fmaxf(__builtin_fmaxf(aoA[t][e], aeB[t][e]), aoB[t][e]);
        }
      return;
    }
    {   // lane group 3: tap (2, 2) of the three channels = columns + 1 | + 3 of row 2c + 1, out of the registers of lanes (l, 0..2)
      const int a1 = f_f2i(r2A[1]), a3 = f_f2i(r2A[3]), b1 = f_f2i(r2B[1]), b3 = f_f2i(r2B[3]);
      const int e0 = __builtin_amdgcn_ds_bpermute(src0, a1), e1 = __builtin_amdgcn_ds_bpermute(src1, a1), e2 = __builtin_amdgcn_ds_bpermute(src2, a1);
      const int q0 = __builtin_amdgcn_ds_bpermute(src0, a3), q1 = __builtin_amdgcn_ds_bpermute(src1, a3), q2 = __builtin_amdgcn_ds_bpermute(src2, a3);
      const int f0 = __builtin_amdgcn_ds_bpermute(src0, b1), f1 = __builtin_amdgcn_ds_bpermute(src1, b1), f2 = __builtin_amdgcn_ds_bpermute(src2, b1);
      const int u0 = __builtin_amdgcn_ds_bpermute(src0, b3), u1 = __builtin_amdgcn_ds_bpermute(src1, b3), u2 = __builtin_amdgcn_ds_bpermute(src2, b3);
      if (g == 3) {
        r1A = (f32x4){f_i2f(e0), f_i2f(e1), f_i2f(q0), f_i2f(q1)};
        r2A = (f32x4){f_i2f(e2), 0.f, f_i2f(q2), 0.f};
        r1B = (f32x4){f_i2f(f0), f_i2f(f1), f_i2f(u0), f_i2f(u1)};
        r2B = (f32x4){f_i2f(f2), 0.f, f_i2f(u2), 0.f};
      }
    }
    FCol x1A, x2A, x1B, x2B;
    f_split(r1A, x1A); f_split(r2A, x2A); f_split(r1B, x1B); f_split(r2B, x2B);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      x1A.m[k] = f_dpp_shr1_u(x1B.p23[k]); x2A.m[k] = f_dpp_shr1_u(x2B.p23[k]);   // the column left of A: the left lane's B
      x1B.m[k] = x1A.p23[k]; x2B.m[k] = x2A.p23[k];                               // the column left of B: this lane's A
    }
    f32x4 aeA[2], aoA[2], aeB[2], aoB[2];
    {
      yfv2_h8 wa[2][2]; f32x4 ssh[2];
      ld_filter(F2_WA, wa); ld_c2(F2_SCST, ssh);
      f_conv_col(cA, x1A, x2A, wa, ssh[0], ssh[1], aeA, aoA);
      f_conv_col(cB, x1B, x2B, wa, ssh[0], ssh[1], aeB, aoB);
    }
    cA = x2A; cB = x2B;
    watch.see(aeA[0][0]); watch.see(aoA[0][0]); watch.see(aeB[0][0]); watch.see(aoB[0][0]);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // 0 from the DPP at the image's left edge stands for the -inf padding: ReLU follows the pooling
        hA[t][e] = __builtin_fmaxf(__builtin_fmaxf(f_dpp_shr1_f(aoB[t][e]), aeA[t][e]), aoA[t][e]);
        hB[t][e] = __builtin_fmaxf(__builtin_fmaxf(aoA[t][e], aeB[t][e]), aoB[t][e]);
      }
  };
  FCol cA, cB;
  f32x4 upA[2], upB[2];
  // pooled row from its input set -> X[2 t + c] as s2h_kernel's load_row delivers it
  auto pooled = [&](InSet& s, f32x4 (&X)[4], int rnext) {
    f32x4 h0A[2], h0B[2], h1A[2], h1B[2];
    conv_row2(cA, cB, s.a[0], s.a[1], s.b[0], s.b[1], h0A, h0B);
    __builtin_amdgcn_sched_barrier(0); issue_half(rnext, s, 0); __builtin_amdgcn_sched_barrier(0);
    conv_row2(cA, cB, s.a[2], s.a[3], s.b[2], s.b[3], h1A, h1B);
    __builtin_amdgcn_sched_barrier(0); issue_half(rnext, s, 1); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mA = __builtin_fmaxf(__builtin_fmaxf(upA[t][e], h0A[t][e]), h1A[t][e]);
        const float mB = __builtin_fmaxf(__builtin_fmaxf(upB[t][e], h0B[t][e]), h1B[t][e]);
        X[2 * t][e] = __builtin_fmaxf(mA, 0.f) * sunscale;
        X[2 * t + 1][e] = __builtin_fmaxf(mB, 0.f) * sunscale;
      }
      upA[t] = h1A[t]; upB[t] = h1B[t];
    }
    if (g >= 2 || !xok) { X[2] = (f32x4){0.f, 0.f, 0.f, 0.f}; X[3] = X[2]; }   // channel tile 1 holds channels 16..23 in lane groups 0, 1
    if (!xok) { X[0] = X[2]; X[1] = X[2]; }
  };

  // ---- stage2.0's row machinery (s2h_kernel)
  auto columns = [&](const f32x4 (&X)[4], float lim, float (&xe)[8], float (&xo)[8], float (&te)[8], float (&to)[8]) {
    f32x2 ine[4], ino[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 ve = X[2 * t] * 16.0f, vo = X[2 * t + 1] * 16.0f;
      ine[2 * t] = (f32x2){ve[0], ve[1]}; ine[2 * t + 1] = (f32x2){ve[2], ve[3]};
      ino[2 * t] = (f32x2){vo[0], vo[1]}; ino[2 * t + 1] = (f32x2){vo[2], vo[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) { xe[4 * t + e] = ve[e]; xo[4 * t + e] = vo[e]; }
    }
    f32x4 ae[2], ao[2];
    {
      yfv2_h8 w1[2][2]; f32x4 sh1[2];
      ld_filter(F2_S2 + S2H_W1, w1); ld_c2(F2_CST, sh1);
      pw_h3(w1, ine, sh1, ae, watch);
      pw_h3(w1, ino, sh1, ao, watch);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      te[c] = __builtin_amdgcn_fmed3f(ae[c >> 2][c & 3], 0.f, lim);
      to[c] = __builtin_amdgcn_fmed3f(ao[c >> 2][c & 3], 0.f, lim);
    }
  };
#define YFV2_TQ(T, c, t) T[((c) * 9 + (t)) >> 2]
#define YFV2_TK(c, t) (((c) * 9 + (t)) & 3)
  auto acc_row = [&](auto dyc, const float (&T)[18], const float (&v0)[8], const float (&v1)[8], float (&S)[8], float (&Q)[8]) {
    constexpr int DY = decltype(dyc)::value;
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) {
      ((DY == 0 ? (void)(S[Cs] = quad_mul<YFV2_TK(Cs, 1)>(YFV2_TQ(T, Cs, 1), v0[Cs]), Q[Cs] = quad_mul<YFV2_TK(Cs, 0)>(YFV2_TQ(T, Cs, 0), v1[Cs]),
                         quad_fmac1<YFV2_TK(Cs, 2)>(S[Cs], YFV2_TQ(T, Cs, 2), v1[Cs]))
                : (void)quad_fmac3<YFV2_TK(Cs, DY * 3 + 1), YFV2_TK(Cs, DY * 3), YFV2_TK(Cs, DY * 3 + 2)>(
                      S[Cs], Q[Cs], YFV2_TQ(T, Cs, DY * 3 + 1), YFV2_TQ(T, Cs, DY * 3), YFV2_TQ(T, Cs, DY * 3 + 2), v0[Cs], v1[Cs])), ...);
    }(std::make_integer_sequence<int, 8>{});
  };

  InSet s0;
  f32x4 P[4];
  float cxe[8], cxo[8], cte[8], cto[8];            // the odd pooled row above the current output row (dy = 0): raw and pw1'd
  const float limx = xok ? __builtin_inff() : 0.f;
  {
    const int r0 = 2 * y0 - 1;                      // the first pooled row this band needs
    if (r0 > 0) {
      // the conv row above pooled row r0 (2 r0 - 1: input rows 4 r0 - 3 .. 4 r0 - 1) gives the carried maxima and the carried split row
      issue(r0 - 1, s0);                            // rows 4 r0 - 4 .. 4 r0 - 1 (the first of them is not needed)
      if constexpr (U8) {                           // a carried row is an X2 (stem_h3u_kernel)
        const u32x4 qa = __builtin_bit_cast(u32x4, s0.a[1]), qb = __builtin_bit_cast(u32x4, s0.b[1]);
        FCol8 ta, tb;
        f_split_u8((yfv2_u3x){qa[0], qa[1], qa[2]}, sel2a, sel2b, ta); f_split_u8((yfv2_u3x){qb[0], qb[1], qb[2]}, sel2a, sel2b, tb);
        ta.m = f_dpp_shr1_u(tb.p23); tb.m = ta.p23;
        cA.p01[0] = ta.p01; cA.p23[0] = ta.p23; cA.m[0] = ta.m; cB.p01[0] = tb.p01; cB.p23[0] = tb.p23; cB.m[0] = tb.m;
        cA.p01[1] = cA.p23[1] = cA.m[1] = 0u; cB.p01[1] = cB.p23[1] = cB.m[1] = 0u;
      } else {
        f_split(s0.a[1], cA); f_split(s0.b[1], cB);
#pragma unroll
        for (int k = 0; k < 2; ++k) { cA.m[k] = f_dpp_shr1_u(cB.p23[k]); cB.m[k] = cA.p23[k]; }
      }
      conv_row2(cA, cB, s0.a[2], s0.a[3], s0.b[2], s0.b[3], upA, upB);
      __builtin_amdgcn_sched_barrier(0);
      issue(r0, s0);
      __builtin_amdgcn_sched_barrier(0);
      pooled(s0, P, r0 + 1);                        // (refills the set with pooled row r0 + 1 = 2 y0 behind itself)
      columns(P, limx, cxe, cxo, cte, cto);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // band 0: pooled row -1 is the block's zero padding; the stem starts above the image (zero carried row, zero maxima: the
      // post-ReLU equivalent of the max-pool's padding)
      issue(0, s0);
      f_split((f32x4){0.f, 0.f, 0.f, 0.f}, cA); cB = cA;
#pragma unroll
      for (int k = 0; k < 2; ++k) { cA.m[k] = 0u; cB.m[k] = 0u; }
#pragma unroll
      for (int t = 0; t < 2; ++t) { upA[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; upB[t] = upA[t]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) P[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      columns(P, 0.f, cxe, cxo, cte, cto);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // here: the set holds (or awaits) the inputs of pooled row 2 y0
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
    float Sm[8], Qm[8], Sp[8], Qp[8], xe[8], xo[8], te[8], to[8];
    {
      float tm[18], tp[18]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 0>{}, tm, cte, cto, Sm, Qm);
      acc_row(std::integral_constant<int, 0>{}, tp, cxe, cxo, Sp, Qp);
    }
    __builtin_amdgcn_sched_barrier(0);
    pooled(s0, P, 2 * oy + 1);                      // even pooled row 2 oy: dy = 1 (the set is refilled with row 2 oy + 1 behind it)
    columns(P, limx, xe, xo, te, to);
    {
      float tm[18], tp[18]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 1>{}, tm, te, to, Sm, Qm);
      acc_row(std::integral_constant<int, 1>{}, tp, xe, xo, Sp, Qp);
    }
    __builtin_amdgcn_sched_barrier(0);
    pooled(s0, P, 2 * oy + 2);                      // odd pooled row 2 oy + 1: dy = 2, and the next output row's dy = 0
    columns(P, limx, cxe, cxo, cte, cto);
    {
      float tm[18], tp[18]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 2>{}, tm, cte, cto, Sm, Qm);
      acc_row(std::integral_constant<int, 2>{}, tp, cxe, cxo, Sp, Qp);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 dm[4], dp[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      dpp_src_ready(Qm[c]); dpp_src_ready(Qp[c]);
      dm[c >> 1][c & 1] = Sm[c] + row_shr1(Qm[c]);
      dp[c >> 1][c & 1] = Sp[c] + row_shr1(Qp[c]);
    }
    f32x4 am[2], ap[2];
    {
      yfv2_h8 wp[2][2]; f32x4 bip[2];
      ld_filter(F2_S2 + S2H_WP, wp); ld_c2(F2_CST + 32, bip);
      pw_h3(wp, dp, bip, ap, watch);
    }
    {
      yfv2_h8 w2[2][2]; f32x4 bi2[2];
      ld_filter(F2_S2 + S2H_W2, w2); ld_c2(F2_CST + 64, bi2);
      pw_h3(w2, dm, bi2, am, watch);
    }
    const bool rowok = oy < y1;                    // wave-uniform
    f32x4 op[2], om[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) { op[t][e] = __builtin_fmaxf(ap[t][e], 0.f) * unscale_p; om[t][e] = __builtin_fmaxf(am[t][e], 0.f) * unscale_2; }
    const int ro = oy * orowb;
    auto st = [&](int k, float v0, float v1) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v0, v1}), rout, (rowok && soff[k] != OOB) ? soff[k] + ro : OOB, 0, 0);
    };
    st(0, op[0][0], op[0][1]); st(1, op[0][2], op[0][3]);
    st(2, om[0][0], om[0][1]); st(3, om[0][2], om[0][3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) st(4 + e, op[1][e], om[1][e]);
  }
  watch.report(a.nonfinite);
#undef YFV2_TQ
#undef YFV2_TK
}

void yfv2_launch_front(const FrontArgs& a0, hipStream_t s) {
  FrontArgs a = a0;
  const int OW = a.s2.IW / 2, OH = a.s2.IH / 2;
  a.s2.nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  // Bands per image.  A band of R output rows costs R + ~0.75 row times (its first pooled row and the conv row above it are computed
  // again), so fewer bands are less work - but the launch is bound by instruction issue and wants two waves on every SIMD for
  // most of its duration: 1.5 x (two 4-wave workgroups per CU) workgroups or more.  256 images: 4 bands (768 workgroups; 5 bands
  // measured the same 130 us, 768 workgroups being three per CU either way); 512 images: 2; one image: 16 bands (47 -> 19.5 us).
  static std::atomic<int> cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (cus <= 0) { if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; cus_of[dev].store(cus, std::memory_order_relaxed); }
  const long long want = 3LL * cus;
  int best_nb = 1;
  for (int nb = 1; nb <= OH && nb <= 16; ++nb) {
    const int R = (OH + nb - 1) / nb, nbe = (OH + R - 1) / R;
    if (R < 2 && OH >= 2) break;
    best_nb = nbe;
    if (((long long)a.s2.B * a.s2.nstrips * nbe + 3) / 4 >= want) break;
  }
  a.s2.nb = best_nb;
  a.s2.R = (OH + a.s2.nb - 1) / a.s2.nb;
  a.s2.nb = (OH + a.s2.R - 1) / a.s2.R;
  const unsigned units = a.s2.B * a.s2.nstrips * a.s2.nb;
  if (a.u8_in) YFV2_LAUNCH(front2_kernel<true>, dim3((units + 3) / 4), dim3(256), F2_FLOATS * sizeof(float), s, a);
  else YFV2_LAUNCH(front2_kernel<false>, dim3((units + 3) / 4), dim3(256), F2_FLOATS * sizeof(float), s, a);
}

// ============================================================================
// stage3.0: the stride-2 block 48 -> 96 (44x44 -> 22x22) in the same streaming form
// ============================================================================
// Reads stage 2's pair planes (two buffers, slot bookkeeping folded into the filter columns / tap channels on the host,
// as for block_s2_kernel<48, .., PPIN>), writes plain NHWC (B, 22, 22, 96) for the stage-3 chain.  Same dataflow as
// s2h_kernel with 48 channels: a lane holds 12 channel positions q = 4t + e <-> channel 16t + 4g + e (t = 0..2: the three
// channel tiles; as K slots: chunk q / 8, slot q % 8 - two K chunks, the second half empty), a pointwise conv is 3 tiles x 2
// chunks x 3 products = 18 MFMAs per pixel tile.  The three 48 x 48 filters are 36 two-term operands per lane - 144
// registers - so they live in LDS (36 KB, fragment-major: the 64 lanes of an operand read hit 64 consecutive 16-byte slots)
// and are read once per input row; a workgroup = the four (strip, band) waves of ONE image sharing that image, one wave
// per SIMD (the two branches' carried rows, column sums and taps are ~320 registers per lane), no barrier after the
// prologue.  Rounds 1-2: block_s2_kernel<48> (tile staged in LDS, 8 waves per image, barriers per phase) 73-77 us.
// image (floats): W1 | Wproj | W2, each [tile 3][chunk 2][term 2][64 lanes][4 dwords] = 3072 | taps main [27][64] | taps proj
// [27][64] | sh1, bias_proj, bias2 (x 2^(sw+4)) [3][48] | 2^-(swp+4), 2^-(sw2+4) | pad | per-lane load byte offsets [6][64]
constexpr int S3H_WFL = 3072, S3H_TM = 9216, S3H_TP = 9216 + 1728, S3H_CST = 9216 + 3456, S3H_OFFS = S3H_CST + 148;

namespace {
// 48 -> 48 pointwise conv of 16 pixels, filter operands from LDS
__device__ __forceinline__ void pw_h3_48(const float* W, int lane, const f32x2 (&in)[6], const f32x4 (&init)[3], f32x4 (&acc)[3], Yfv2Watch& watch) {
  u32x4 b1[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, b2[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
  for (int k = 0; k < 6; ++k) { unsigned h1, h2; split2(in[k], h1, h2); b1[k >> 2][k & 3] = h1; b2[k >> 2][k & 3] = h2; }
  const u32x4* Wq = reinterpret_cast<const u32x4*>(W) + lane;
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = init[t];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const yfv2_h8 x1 = __builtin_bit_cast(yfv2_h8, b1[c]), x2 = __builtin_bit_cast(yfv2_h8, b2[c]);
    yfv2_h8 wa[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) wa[t][k] = __builtin_bit_cast(yfv2_h8, Wq[((t * 2 + c) * 2 + k) * 64]);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], x2, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], x1, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], x1, acc[t], 0, 0, 0);
  }
  watch.see(acc[0][0]);
}
}  // namespace

// s3h2_kernel (round 5): the wave program described above at TWO waves per SIMD.  Its first form (s3h_kernel, rounds 3-5, removed in
// round 6) held 54 depthwise taps + 36 BatchNorm constants + state in 316 registers; with the taps and constants read from the
// workgroup's LDS copy at the point of use (as front2_kernel does) the wave fits 256 registers, two workgroups share a CU and an
// image is cut into eight (strip, band) units instead of four: the launch is a latency chain (36 us for ONE image, 45 for 256)
// that a second wave per SIMD overlaps.  Bit-identical to that first form (round 5's same-box check, profiles/r05_experiments.txt).
constexpr int S3H2_TAPS = 3 * S3H_WFL, S3H2_TSTRIDE = 60, S3H2_CST = S3H2_TAPS + 64 * S3H2_TSTRIDE, S3H2_FLOATS = S3H2_CST + 160;
__global__ __launch_bounds__(256, 2) void s3h2_kernel(BlockS2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int IH = a.H, IW = a.W, OH = IH >> 1, OW = IW >> 1;
  const int nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  const int R = a.R, nb = (OH + R - 1) / R;
  const int units = nstrips * nb, wpi = (units + 3) >> 2;     // workgroups per image
  const int b = blockIdx.x / wpi, wi = blockIdx.x - b * wpi;
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 15, g = lane >> 4;
  const int uid = wi * 4 + (tid >> 6);
  const float* img = a.img16;
  {   // the three filters -> LDS (straight 16-byte copy, every load issued before the first store)
    const f32x4* src = reinterpret_cast<const f32x4*>(img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    f32x4 tmp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) tmp[k] = src[tid + k * 256];
#pragma unroll
    for (int k = 0; k < 9; ++k) dst[tid + k * 256] = tmp[k];
    float tv[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) { const int e = tid + 256 * j; tv[j] = e < 54 * 64 ? img[S3H_TM + e] : 0.f; }   // e = q * 64 + lane over tm [27] then tp [27]
    const float cv = tid < 146 ? img[S3H_CST + tid] : 0.f;
#pragma unroll
    for (int j = 0; j < 14; ++j) { const int e = tid + 256 * j; if (e < 54 * 64) lds[S3H2_TAPS + (e & 63) * S3H2_TSTRIDE + (e >> 6)] = tv[j]; }
    if (tid < 146) lds[S3H2_CST + tid] = cv;
  }
  __syncthreads();
  if (uid >= units) return;
  const int strip = uid % nstrips, band = uid / nstrips;
  const int ox = 15 * strip + l;
  const bool xok = ox < OW;
  const bool st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R, y1 = min(OH, y0 + R);
  constexpr int OOB = (int)0x80000000;
  const float* W1 = lds; const float* WP = lds + S3H_WFL; const float* W2 = lds + 2 * S3H_WFL;

  // input: stage 2's two pair-plane buffers of this image (adjacent: buffer 1 at + pp_bufstride floats, the next image at + pp_imgstride)
  __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * (size_t)a.pp_imgstride), 0,
                                                                   (int)((a.pp_bufstride + 48LL * IH * IW) * 4), 0x00020000);
  Yfv2Watch watch;
  auto ld_taps = [&](float (&tm)[27], float (&tp)[27]) {   // [lane][54] at a pitch of 60 floats (16-byte reads, the 16 lanes of a read group on 16 bank quads)
    asm volatile("" ::: "memory");
    const f32x4* q = reinterpret_cast<const f32x4*>(lds + S3H2_TAPS + lane * S3H2_TSTRIDE);
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const f32x4 v = q[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int i = 4 * j + e; if (i < 27) tm[i] = v[e]; else if (i < 54) tp[i - 27] = v[e]; }
    }
  };
  auto ld_c3 = [&](int off, f32x4 (&c)[3]) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 3; ++t) c[t] = *reinterpret_cast<const f32x4*>(lds + S3H2_CST + off + 16 * t + 4 * g);
  };
  const float unscale_p = img[S3H_CST + 144], unscale_2 = img[S3H_CST + 145];
  int loff[6];
  {
    const int* po = reinterpret_cast<const int*>(img + S3H_OFFS);
#pragma unroll
    for (int k = 0; k < 6; ++k) loff[k] = xok ? po[k * 64 + lane] + 2 * ox * 8 : OOB;
  }
  const int irowb = IW * 8;
  float* __restrict__ outp = a.out + ((size_t)b * OH * OW + (st_lane ? ox : 0)) * 96 + 4 * g;

  auto load_row = [&](int iy, f32x4 (&X)[6]) {
    const bool rok = iy >= 0 && iy < IH;           // wave-uniform
#pragma unroll
    for (int k = 0; k < 6; ++k)
      X[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (rok && loff[k] != OOB) ? loff[k] + iy * irowb : OOB, 0, 0));
  };
  auto columns = [&](const f32x4 (&X)[6], float lim, float (&xe)[12], float (&xo)[12], float (&te)[12], float (&to)[12]) {
    f32x2 ine[6], ino[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x4 v = X[k] * 16.0f;
      ine[k] = (f32x2){v[0], v[1]}; ino[k] = (f32x2){v[2], v[3]};
      xe[2 * k] = v[0]; xe[2 * k + 1] = v[1]; xo[2 * k] = v[2]; xo[2 * k + 1] = v[3];
    }
    f32x4 ae[3], ao[3];
    {
      f32x4 sh1[3]; ld_c3(0, sh1);
      pw_h3_48(W1, lane, ine, sh1, ae, watch);
      pw_h3_48(W1, lane, ino, sh1, ao, watch);
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      te[c] = __builtin_amdgcn_fmed3f(ae[c >> 2][c & 3], 0.f, lim);
      to[c] = __builtin_amdgcn_fmed3f(ao[c >> 2][c & 3], 0.f, lim);
    }
  };
#define YFV2_TQ(T, c, t) T[((c) * 9 + (t)) >> 2]
#define YFV2_TK(c, t) (((c) * 9 + (t)) & 3)
  auto acc_row = [&](auto dyc, const float (&T)[27], const float (&v0)[12], const float (&v1)[12], float (&S)[12], float (&Q)[12]) {
    constexpr int DY = decltype(dyc)::value;
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) {
      ((DY == 0 ? (void)(S[Cs] = quad_mul<YFV2_TK(Cs, 1)>(YFV2_TQ(T, Cs, 1), v0[Cs]), Q[Cs] = quad_mul<YFV2_TK(Cs, 0)>(YFV2_TQ(T, Cs, 0), v1[Cs]),
                         quad_fmac1<YFV2_TK(Cs, 2)>(S[Cs], YFV2_TQ(T, Cs, 2), v1[Cs]))
                : (void)quad_fmac3<YFV2_TK(Cs, DY * 3 + 1), YFV2_TK(Cs, DY * 3), YFV2_TK(Cs, DY * 3 + 2)>(
                      S[Cs], Q[Cs], YFV2_TQ(T, Cs, DY * 3 + 1), YFV2_TQ(T, Cs, DY * 3), YFV2_TQ(T, Cs, DY * 3 + 2), v0[Cs], v1[Cs])), ...);
    }(std::make_integer_sequence<int, 12>{});
  };

  f32x4 X[6], Y[6];
  float cxe[12], cxo[12], cte[12], cto[12];
  {
    const int iy = 2 * y0 - 1;
    load_row(iy, X);
    load_row(iy + 1, Y);
    columns(X, (xok && iy >= 0) ? __builtin_inff() : 0.f, cxe, cxo, cte, cto);
    load_row(iy + 2, X);
  }
  const float limx = xok ? __builtin_inff() : 0.f;
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
    float Sm[12], Qm[12], Sp[12], Qp[12], xe[12], xo[12], te[12], to[12];
    {
      float tm[27], tp[27]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 0>{}, tm, cte, cto, Sm, Qm);
      acc_row(std::integral_constant<int, 0>{}, tp, cxe, cxo, Sp, Qp);
    }
    columns(Y, limx, xe, xo, te, to);
    {
      float tm[27], tp[27]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 1>{}, tm, te, to, Sm, Qm);
      acc_row(std::integral_constant<int, 1>{}, tp, xe, xo, Sp, Qp);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 2, Y);
    __builtin_amdgcn_sched_barrier(0);
    columns(X, limx, cxe, cxo, cte, cto);
    {
      float tm[27], tp[27]; ld_taps(tm, tp);
      acc_row(std::integral_constant<int, 2>{}, tm, cte, cto, Sm, Qm);
      acc_row(std::integral_constant<int, 2>{}, tp, cxe, cxo, Sp, Qp);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 3, X);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 dm[6], dp[6];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      dpp_src_ready(Qm[c]); dpp_src_ready(Qp[c]);
      dm[c >> 1][c & 1] = Sm[c] + row_shr1(Qm[c]);
      dp[c >> 1][c & 1] = Sp[c] + row_shr1(Qp[c]);
    }
    f32x4 am[3], ap[3];
    { f32x4 bip[3]; ld_c3(48, bip); pw_h3_48(WP, lane, dp, bip, ap, watch); }
    { f32x4 bi2[3]; ld_c3(96, bi2); pw_h3_48(W2, lane, dm, bi2, am, watch); }
    if (st_lane && oy < y1) {
      float* o = outp + (size_t)oy * OW * 96;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x4 vp, vm;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vp[e] = __builtin_fmaxf(ap[t][e], 0.f) * unscale_p; vm[e] = __builtin_fmaxf(am[t][e], 0.f) * unscale_2; }
        *reinterpret_cast<f32x4*>(o + 16 * t) = vp;          // proj: channels 0..47
        *reinterpret_cast<f32x4*>(o + 48 + 16 * t) = vm;     // main: channels 48..95
      }
    }
  }
  watch.report(a.nonfinite);
#undef YFV2_TQ
#undef YFV2_TK
}

bool yfv2_s3h_supported(int H, int W) { return H >= 4 && W >= 4 && !(H & 1) && !(W & 1) && W / 2 <= 16 * 15; }

void yfv2_launch_s3h(const BlockS2Args& a0, hipStream_t s) {
  BlockS2Args a = a0;
  const int OH = a.H / 2, OW = a.W / 2;
  const int nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  int nb = (8 + nstrips - 1) / nstrips;             // about eight (strip, band) waves per image: two workgroups
  if (nb > OH) nb = OH;
  a.R = (OH + nb - 1) / nb;
  nb = (OH + a.R - 1) / a.R;
  const int units = nstrips * nb;
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&s3h2_kernel), lds_ok);
  YFV2_LAUNCH(s3h2_kernel, dim3(a.B * ((units + 3) / 4)), dim3(256), S3H2_FLOATS * sizeof(float), s, a);
}

// ============================================================================
// stage4.0: the stride-2 block 96 -> 192 (22x22 -> 11x11), streaming form with two wave ROLES
// ============================================================================
// NHWC in (stage 3's output, in whatever channel order the chain kernel leaves it: folded into the filter columns / tap
// channels on the host), NHWC out.  96 channels = 24 positions per lane (q = 4t + e <-> channel 16t + 4g + e, t = 0..5; K
// chunk q / 8, slot q % 8: three full chunks), a pointwise conv = 6 tiles x 3 chunks x 3 products = 54 MFMAs per pixel tile.
// The carried rows, column sums and taps of BOTH branches would be ~600 registers per lane, so the branches are two wave
// roles of one workgroup (role = wave & 1; they share the input rows through L1 / L2 and the three filters - 108 KB of
// two-term fp16 operands - through LDS); a workgroup = 2 bands x 2 roles = the four waves of ONE image, one per SIMD.
// Rounds 1-2: block_s2w_kernel (bands of two output rows, four barrier-separated phases, Wproj / W2 streamed through one
// LDS slot) 61-67 us.
// image (floats): W1 | Wproj | W2, each [tile 6][chunk 3][term 2][64 lanes][4 dwords] = 9216 | taps main [54][64] | taps proj
// [54][64] | sh1, bias_proj, bias2 (x 2^(sw+4)) [3][96] | 2^-(swp+4), 2^-(sw2+4)
constexpr int S4H_WFL = 9216, S4H_TM = 27648, S4H_TP = 27648 + 3456, S4H_CST = 27648 + 6912;

namespace {
// 96 -> 96 pointwise conv of 16 pixels, filter operands from LDS (one 16-byte read per (tile, chunk, term), used at once)
__device__ __forceinline__ void pw_h3_96(const float* W, int lane, const f32x2 (&in)[12], const f32x4 (&init)[6], f32x4 (&acc)[6], Yfv2Watch& watch) {
  const u32x4* Wq = reinterpret_cast<const u32x4*>(W) + lane;
#pragma unroll
  for (int t = 0; t < 6; ++t) acc[t] = init[t];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    u32x4 b1, b2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { unsigned h1, h2; split2(in[4 * c + k], h1, h2); b1[k] = h1; b2[k] = h2; }
    const yfv2_h8 x1 = __builtin_bit_cast(yfv2_h8, b1), x2 = __builtin_bit_cast(yfv2_h8, b2);
    yfv2_h8 wa[6][2];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) wa[t][k] = __builtin_bit_cast(yfv2_h8, Wq[((t * 3 + c) * 2 + k) * 64]);
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], x2, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][1], x1, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[t][0], x1, acc[t], 0, 0, 0);
  }
  watch.see(acc[0][0]);
}

template <bool MAIN>
__device__ __forceinline__ void s4h_body(const BlockS2Args& a, const float* lds, int b, int strip, int band, int R, int lane) {
  const int IH = a.H, IW = a.W, OH = IH >> 1, OW = IW >> 1;
  const int l = lane & 15, g = lane >> 4;
  const int ox = 15 * strip + l;
  const bool xok = ox < OW;
  const bool st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R, y1 = min(OH, y0 + R);
  constexpr int OOB = (int)0x80000000;
  const float* img = a.img16;
  const float* W1 = lds; const float* WB = lds + (MAIN ? 2 : 1) * S4H_WFL;   // this role's second filter: W2 (main) / Wproj

  __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * 96 * IH * IW), 0, 96 * IH * IW * 4, 0x00020000);
  Yfv2Watch watch;
  float tq[54];
#pragma unroll
  for (int q = 0; q < 54; ++q) tq[q] = img[(MAIN ? S4H_TM : S4H_TP) + q * 64 + lane];
  f32x4 sh1[6], bib[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    sh1[t] = *reinterpret_cast<const f32x4*>(img + S4H_CST + 16 * t + 4 * g);
    bib[t] = *reinterpret_cast<const f32x4*>(img + S4H_CST + (MAIN ? 192 : 96) + 16 * t + 4 * g);
  }
  const float unscale = img[S4H_CST + 288 + (MAIN ? 1 : 0)];
  const int lbase = xok ? (2 * ox * 96 + 4 * g) * 4 : OOB;     // byte offset of (column 2ox, channel 4g) inside an input row
  const int irowb = IW * 96 * 4;
  float* __restrict__ outp = a.out + ((size_t)b * OH * OW + (st_lane ? ox : 0)) * 192 + (MAIN ? 96 : 0) + 4 * g;

  // one input row: X[t] = channels 16t+4g..+3 of column 2ox, X[6 + t] = the same of column 2ox+1
  auto load_row = [&](int iy, f32x4 (&X)[12]) {
    const bool rok = iy >= 0 && iy < IH;           // wave-uniform
    const int base = (rok && lbase != OOB) ? lbase + iy * irowb : OOB;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      X[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, base, t * 64, 0));          // (an out-of-range voffset stays out of range)
      X[6 + t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, base, 384 + t * 64, 0));
    }
  };
  // this role's depthwise input of one input row, both columns: raw x 2^4 (proj) / relu(pw1) x 2^(sw1+4), 0 outside the image (main)
  auto columns = [&](const f32x4 (&X)[12], float lim, float (&ve)[24], float (&vo)[24]) {
    if constexpr (MAIN) {
      f32x2 in[12];
      f32x4 acc[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) { const f32x4 v = X[t] * 16.0f; in[2 * t] = (f32x2){v[0], v[1]}; in[2 * t + 1] = (f32x2){v[2], v[3]}; }
      pw_h3_96(W1, lane, in, sh1, acc, watch);
#pragma unroll
      for (int c = 0; c < 24; ++c) ve[c] = __builtin_amdgcn_fmed3f(acc[c >> 2][c & 3], 0.f, lim);
#pragma unroll
      for (int t = 0; t < 6; ++t) { const f32x4 v = X[6 + t] * 16.0f; in[2 * t] = (f32x2){v[0], v[1]}; in[2 * t + 1] = (f32x2){v[2], v[3]}; }
      pw_h3_96(W1, lane, in, sh1, acc, watch);
#pragma unroll
      for (int c = 0; c < 24; ++c) vo[c] = __builtin_amdgcn_fmed3f(acc[c >> 2][c & 3], 0.f, lim);
    } else {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const f32x4 e = X[t] * 16.0f, o = X[6 + t] * 16.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { ve[4 * t + k] = e[k]; vo[4 * t + k] = o[k]; }
      }
    }
  };
#define YFV2_TQ(c, t) tq[((c) * 9 + (t)) >> 2]
#define YFV2_TK(c, t) (((c) * 9 + (t)) & 3)
  auto acc_row = [&](auto dyc, const float (&v0)[24], const float (&v1)[24], float (&S)[24], float (&Q)[24]) {
    constexpr int DY = decltype(dyc)::value;
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) {
      ((DY == 0 ? (void)(S[Cs] = quad_mul<YFV2_TK(Cs, 1)>(YFV2_TQ(Cs, 1), v0[Cs]), Q[Cs] = quad_mul<YFV2_TK(Cs, 0)>(YFV2_TQ(Cs, 0), v1[Cs]),
                         quad_fmac1<YFV2_TK(Cs, 2)>(S[Cs], YFV2_TQ(Cs, 2), v1[Cs]))
                : (void)quad_fmac3<YFV2_TK(Cs, DY * 3 + 1), YFV2_TK(Cs, DY * 3), YFV2_TK(Cs, DY * 3 + 2)>(
                      S[Cs], Q[Cs], YFV2_TQ(Cs, DY * 3 + 1), YFV2_TQ(Cs, DY * 3), YFV2_TQ(Cs, DY * 3 + 2), v0[Cs], v1[Cs])), ...);
    }(std::make_integer_sequence<int, 24>{});
  };

  f32x4 X[12];
  float ce[24], co[24];                             // the odd input row above the current output row (dy = 0)
  {
    const int iy = 2 * y0 - 1;
    load_row(iy, X);
    columns(X, (xok && iy >= 0) ? __builtin_inff() : 0.f, ce, co);
    load_row(iy + 1, X);
  }
  const float limx = xok ? __builtin_inff() : 0.f;
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
    float S[24], Q[24], ve[24], vo[24];
    acc_row(std::integral_constant<int, 0>{}, ce, co, S, Q);
    columns(X, limx, ve, vo);                       // even input row 2oy: dy = 1
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 1, X);
    __builtin_amdgcn_sched_barrier(0);
    acc_row(std::integral_constant<int, 1>{}, ve, vo, S, Q);
    columns(X, limx, ce, co);                       // odd input row 2oy+1: dy = 2, and the next output row's dy = 0
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 2, X);
    __builtin_amdgcn_sched_barrier(0);
    acc_row(std::integral_constant<int, 2>{}, ce, co, S, Q);
    f32x2 d[12];
#pragma unroll
    for (int c = 0; c < 24; ++c) { dpp_src_ready(Q[c]); d[c >> 1][c & 1] = S[c] + row_shr1(Q[c]); }
    f32x4 acc[6];
    pw_h3_96(WB, lane, d, bib, acc, watch);
    if (st_lane && oy < y1) {
      float* o = outp + (size_t)oy * OW * 192;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(acc[t][e], 0.f) * unscale;
        *reinterpret_cast<f32x4*>(o + 16 * t) = v;
      }
    }
  }
  watch.report(a.nonfinite);
#undef YFV2_TQ
#undef YFV2_TK
}
}  // namespace

__global__ __launch_bounds__(256, 1) void s4h_kernel(BlockS2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int OH = a.H >> 1, OW = a.W >> 1;
  const int nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  const int R = a.R, nb = (OH + R - 1) / R;
  const int units = nstrips * nb, wpi = (units + 1) >> 1;     // workgroups per image: two (strip, band) units x two roles each
  const int b = blockIdx.x / wpi, wi = blockIdx.x - b * wpi;
  const int tid = threadIdx.x, wave = tid >> 6;
  {   // the three filters -> LDS in three rounds of nine 16-byte loads per thread
    const f32x4* src = reinterpret_cast<const f32x4*>(a.img16);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
      f32x4 tmp[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) tmp[k] = src[(r * 9 + k) * 256 + tid];
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[(r * 9 + k) * 256 + tid] = tmp[k];
    }
  }
  __syncthreads();
  if (a.s4_main_bands > 0) {
    // Round 5.  The launch is as long as its longest wave: per output row the main branch issues 3331 instructions (pw1 of two input
    // rows of two columns, depthwise, pw2), the proj branch 672 - with two bands x two roles the two proj waves were done after a
    // fifth of the launch and the two main waves walked six rows each: 36 us for ONE image.  Now three main waves walk four rows
    // each and one proj wave walks all eleven (7.4 k instructions against 13.3 k + the band's extra pw1 row).  Same arithmetic per
    // output row whatever the band: bit-identical.
    const int mb = a.s4_main_bands;
    if (wave < mb) s4h_body<true>(a, lds, (int)blockIdx.x, 0, wave, (OH + mb - 1) / mb, tid & 63);
    else if (wave == 3) s4h_body<false>(a, lds, (int)blockIdx.x, 0, 0, OH, tid & 63);
    return;
  }
  const int uid = wi * 2 + (wave >> 1);
  if (uid >= units) return;
  const int strip = uid % nstrips, band = uid / nstrips;
  if (wave & 1) s4h_body<true>(a, lds, b, strip, band, R, tid & 63);
  else s4h_body<false>(a, lds, b, strip, band, R, tid & 63);
}

bool yfv2_s4h_supported(int H, int W) { return H >= 4 && W >= 4 && !(H & 1) && !(W & 1); }

void yfv2_launch_s4h(const BlockS2Args& a0, hipStream_t s) {
  BlockS2Args a = a0;
  const int OH = a.H / 2, OW = a.W / 2;
  const int nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
  int nb = (2 + nstrips - 1) / nstrips;             // about two (strip, band) units per image: with the two roles one workgroup
  if (nb > OH) nb = OH;
  a.R = (OH + nb - 1) / nb;
  nb = (OH + a.R - 1) / a.R;
  const int units = nstrips * nb;
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&s4h_kernel), lds_ok);
  a.s4_main_bands = (nstrips == 1 && OH >= 6) ? 3 : 0;   // (wider or very low maps: two bands x two roles per workgroup, round 4's form)
  if (a.s4_main_bands) { YFV2_LAUNCH(s4h_kernel, dim3(a.B), dim3(256), 3 * S4H_WFL * sizeof(float), s, a); return; }
  YFV2_LAUNCH(s4h_kernel, dim3(a.B * ((units + 1) / 2)), dim3(256), 3 * S4H_WFL * sizeof(float), s, a);
}
