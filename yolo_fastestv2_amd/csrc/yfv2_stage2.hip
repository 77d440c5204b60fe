// yfv2_stage2.hip - gfx950 (CDNA4, wave64): ShuffleNetV2 stage 2 (44x44 maps, 24-channel branches) in
// "one pixel per lane" form on v_mfma_f32_4x4x1_16b_f32.
//
// Reference layers (read for behaviour only): model/backbone/shufflenetv2.py:19-63 (ShuffleV2Block,
// channel_shuffle), :102-109 (stage loop).
//
// Why a second formulation for this stage: with 24-channel branches the 16x16x4 tile pads M and K
// from 24 to 32 and the depthwise 3x3 between the two pointwise convs has to go through LDS; the fused
// LDS kernel (yfv2_block.hip) measured 70 us per block at 256 images.  In the 16-block 4x4x1 MFMA a
// block is 4 adjacent lanes and D_blk[i][j] += A_blk[i] * B_blk[j]: with B = "channel k of this lane's
// own pixel" and A = W[4m + i][k] (broadcast from one block, cbsz/abid) one instruction is 64 pixels x 4
// output channels x 1 input channel - no padding at any channel count that is a multiple of 4 - and
// a lane ends up holding all channels of ITS pixel in registers.  The depthwise conv is then plain
// per-lane FMAs over three register rows (vertical taps) plus two DPP row shifts (horizontal taps):
// no LDS anywhere, no barriers.
//
// Activation layout of stage 2 ("pair planes"): [image][24 pairs][H][W][2] fp32.  A lane = a pixel reads /
// writes one 8-byte pair per instruction, 16 lanes = 128 contiguous bytes.  Which logical channel sits in
// which (pair, element) slot, and in which of the two stage buffers a pair currently lives, is tracked on
// the host (yfv2_api.hip, Stage2Layout): a stride-1 block only ever reads the 12 pairs that hold its odd
// (branch) channels and writes the branch result into the other buffer's copy of the same 12 pairs; the
// even (pass-through) channels are never touched, channel_shuffle / concat are pure bookkeeping that is
// folded into the order of the filter columns / rows when the weights are packed.
#include "yfv2_internal.h"
#include <utility>

namespace {

__device__ __forceinline__ float row_shr1(float v) {   // lane l <- lane l-1 inside its 16-lane row, 0 at l = 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
// tap k (0..3) of the lane's quad: the depthwise taps live four to a register in the lanes of every quad
// (all quads identical), DPP quad_perm:[k,k,k,k] hands tap k to all lanes without touching SGPRs or LDS
template <int K>
__device__ __forceinline__ float quad_mul(float tap4, float v) {   // tap4[quad lane K] * v
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tap4), K * 0x55, 0xf, 0xf, false)) * v;
}
// acc += tap4[quad lane K] * v as ONE v_fmac_f32_dpp (hipcc folds the DPP into v_mul but not into v_fmac: it
// emits v_mov 0 + v_mov_dpp + v_fmac).  Inline asm hides two gfx9 hazards from the compiler's hazard recognizer,
// both met in practice (wrong depthwise sums in one wave role only):
//   * the tap register may just have been written by a VALU op the compiler inserted (a v_accvgpr_read of a
//     spilled tap): "VALU writes VGPR -> DPP reads it" needs 2 wait states -> every asm block starts with s_nop 1;
//   * the asm WRITES acc with a VALU op, and a compiler-generated DPP (row_shr1/row_shl1 of a column sum) may
//     read it next -> dpp_src_ready() below.
// Groups of FMAs share one block (one s_nop).  KA.. are the quad lanes of the taps (immediates).
#define YFV2_QP(n) "quad_perm:[%" #n ",%" #n ",%" #n ",%" #n "] row_mask:0xf bank_mask:0xf\n\t"
template <int KA>
__device__ __forceinline__ void quad_fmac1(float& s, float ta, float v) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 " YFV2_QP(3) : "+v"(s) : "v"(ta), "v"(v), "n"(KA));
}
// s += ta*v0; q += tb*v1; s += tc*v1   (one vertical tap row of a stride-2 window: dx = 1, 0, 2)
template <int KA, int KB, int KC>
__device__ __forceinline__ void quad_fmac3(float& s, float& q, float ta, float tb, float tc, float v0, float v1) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %2, %5 " YFV2_QP(7) "v_fmac_f32_dpp %1, %3, %6 " YFV2_QP(8) "v_fmac_f32_dpp %0, %4, %6 " YFV2_QP(9)
      : "+v"(s), "+v"(q) : "v"(ta), "v"(tb), "v"(tc), "v"(v0), "v"(v1), "n"(KA), "n"(KB), "n"(KC));
}
// a0 += t3*u + t6*w; a1 += t4*u + t7*w; a2 += t5*u + t8*w   (rows dy = 1, 2 of a stride-1 window, all three dx)
template <int K3, int K4, int K5, int K6, int K7, int K8>
__device__ __forceinline__ void quad_fmac6(float& a0, float& a1, float& a2, float t3, float t4, float t5, float t6, float t7, float t8,
                                           float u, float w) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %9 " YFV2_QP(11) "v_fmac_f32_dpp %1, %4, %9 " YFV2_QP(12) "v_fmac_f32_dpp %2, %5, %9 " YFV2_QP(13)
      "v_fmac_f32_dpp %0, %6, %10 " YFV2_QP(14) "v_fmac_f32_dpp %1, %7, %10 " YFV2_QP(15) "v_fmac_f32_dpp %2, %8, %10 " YFV2_QP(16)
      : "+v"(a0), "+v"(a1), "+v"(a2)
      : "v"(t3), "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(t8), "v"(u), "v"(w), "n"(K3), "n"(K4), "n"(K5), "n"(K6), "n"(K7), "n"(K8));
}
// 2 wait states between an inline-asm VALU write of v and a DPP read of v (gfx9 "VALU writes VGPR -> DPP reads it")
__device__ __forceinline__ void dpp_src_ready(float& v) { asm volatile("s_nop 1" : "+v"(v)); }
__device__ __forceinline__ float row_shl1(float v) {   // lane l <- lane l+1 inside its 16-lane row, 0 at l = 15
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// 24 -> 24 pointwise conv of the lane's pixel: acc[m] (4 consecutive output positions) = sum_k W[4m+i][k] * in[k]
// + shift, as 6 x 25 MFMAs: input "channel" 24 is the constant 1 and carries the BN shift, so the
// accumulator starts from the inline constant 0 and no shift registers exist.  wq[q], lane 4j+i holds the
// (BN-scale-folded) filter entry (m, k) with m*25 + k = 16q + j.
template <int K>
__device__ __forceinline__ void pw24_step(const float (&wq)[10], float bv, f32x4 (&acc)[6]) {
#define YFV2_PW24_MM(M)                                                                                        \
  acc[M] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[((M) * 25 + K) >> 4], bv, K == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[M], 4, \
                                               ((M) * 25 + K) & 15, 0)
  YFV2_PW24_MM(0); YFV2_PW24_MM(1); YFV2_PW24_MM(2); YFV2_PW24_MM(3); YFV2_PW24_MM(4); YFV2_PW24_MM(5);
#undef YFV2_PW24_MM
}
template <int... Ks>
__device__ __forceinline__ void pw24_impl(const float (&wq)[10], const float (&in)[24], float one, f32x4 (&acc)[6],
                                          std::integer_sequence<int, Ks...>) {
  (pw24_step<Ks>(wq, in[Ks], acc), ...);
  pw24_step<24>(wq, one, acc);
}
__device__ __forceinline__ void pw24(const float (&wq)[10], const float (&in)[24], float one, f32x4 (&acc)[6]) {
  pw24_impl(wq, in, one, acc, std::make_integer_sequence<int, 24>{});
}

// image offsets (floats)
constexpr int S1PX_W1 = 0, S1PX_W2 = 640, S1PX_TAPS = 1280, S1PX_FL = 1280 + 54 * 64;   // taps: [54 regs][64 lanes], lane&3 = k holds tap 4q+k, flat tap index = c*9 + dy*3 + dx

}  // namespace

// ----------------------------------------------------------------------------
// stride-1 ShuffleV2 block, 24-channel branch: pw1+BN+ReLU -> dw3x3+BN -> pw2+BN+ReLU on the 12 branch pairs
// ----------------------------------------------------------------------------
// Work unit = 16 lanes = (column strip, row band) of one image; a wave = four units of the same image
// (the buffer resource is wave-uniform).  Strip s covers columns 14s .. 14s+15: the outer lanes of an inner
// strip edge are halo lanes (their pw1 output is needed by the neighbour's 3x3 window, their own output
// is not stored); at the image edge the DPP zero fill IS the conv's zero padding, so 44 columns are
// exactly three strips.  A lane slides down its column: per step it runs pw1 on row r (24 inputs -> t),
// the depthwise 3x3 of row r-1 from the t rows r-2, r-1, r held in registers, pw2, and hands the 24
// outputs to the next step, which stores them before it issues its own prefetch (one in-order vmcnt).
// Rows above/below the image and columns >= W get t = 0 (v_med3 against a per-lane 0 / +inf limit).
__global__ __launch_bounds__(64, 1) void s1px_kernel(S1PxArgs a) {
  const int H = a.H, W = a.W;
  const int nstrips = a.nstrips, nb = a.nb, R = a.R;
  const int units = nstrips * nb, wpi = (units + 3) >> 2;
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int lane = threadIdx.x, l = lane & 15;
  const int uid = wi * 4 + (lane >> 4);
  const int band = uid % nb, strip = uid / nb;
  const int x = 14 * strip + l;
  const bool xok = uid < units && x < W;
  const bool st_lane = xok && (l > 0 || strip == 0) && (l < 15 || strip == nstrips - 1);
  const int y0 = band * R, y1 = min(H, y0 + R);

  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.act + (size_t)b * a.img_stride), 0, a.num_records, 0x00020000);
  const int rowb = W * 8;                         // bytes per row of one pair plane
  const int OOB = (int)0x80000000;

  float w1q[10], w2q[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) { w1q[q] = a.img[S1PX_W1 + q * 64 + lane]; w2q[q] = a.img[S1PX_W2 + q * 64 + lane]; }
  float tq[54];
#pragma unroll
  for (int q = 0; q < 54; ++q) tq[q] = a.img[S1PX_TAPS + q * 64 + lane];
  const float one = 1.0f;

  auto load_row = [&](int r, float (&in)[24]) {
    const int vo = (xok && r >= 0 && r < H) ? (r * W + x) * 8 : OOB;
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) {
      const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, a.src_off[kk], 0));
      in[2 * kk] = v[0]; in[2 * kk + 1] = v[1];
    }
  };

  float cur[24], nxt[24];
  float tA[24], tB[24], tC[24];
  f32x4 pend[6];
  int pend_vo = OOB;
#pragma unroll
  for (int m = 0; m < 6; ++m) pend[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 24; ++c) { tA[c] = 0.f; tB[c] = 0.f; tC[c] = 0.f; }
  load_row(y0 - 1, cur);

  // step j: pw1 of row r = y0-1+j into tn; for j >= 2 the block output of row r-1 from (tp2, tp1, tn)
  auto step = [&](int j, const float (&tp2)[24], const float (&tp1)[24], float (&tn)[24]) {
    const int r = y0 - 1 + j;
    // last step's outputs go out first, then this step's prefetch
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){pend[m][0], pend[m][1]}), rsrc, pend_vo, a.dst_off[2 * m], 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){pend[m][2], pend[m][3]}), rsrc, pend_vo, a.dst_off[2 * m + 1], 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_row(r + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[6];
    pw24(w1q, cur, one, acc);
    const float lim = (xok && r >= 0 && r < H) ? __builtin_inff() : 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) tn[c] = __builtin_amdgcn_fmed3f(acc[c >> 2][c & 3], 0.f, lim);   // ReLU, and 0 outside the image

    float d[24];
    // depthwise 3x3 (BN scale folded into the taps, BN shift folded into pw2's bias): vertical taps are
    // per-lane FMAs over the three t rows, the dx = 0 / dx = 2 column sums move one lane right / left
    auto dw_ch = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
#define YFV2_TQ(t) tq[(c * 9 + (t)) >> 2]
#define YFV2_TK(t) ((c * 9 + (t)) & 3)
      float a0 = quad_mul<YFV2_TK(0)>(YFV2_TQ(0), tp2[c]), a1 = quad_mul<YFV2_TK(1)>(YFV2_TQ(1), tp2[c]), a2 = quad_mul<YFV2_TK(2)>(YFV2_TQ(2), tp2[c]);
      quad_fmac6<YFV2_TK(3), YFV2_TK(4), YFV2_TK(5), YFV2_TK(6), YFV2_TK(7), YFV2_TK(8)>(a0, a1, a2, YFV2_TQ(3), YFV2_TQ(4), YFV2_TQ(5), YFV2_TQ(6),
                                                                                       YFV2_TQ(7), YFV2_TQ(8), tp1[c], tn[c]);
#undef YFV2_TQ
#undef YFV2_TK
      dpp_src_ready(a2);   // a2 is the asm block's last write and the next DPP source (a0 has two instructions behind it)
      d[c] = a1 + row_shr1(a0) + row_shl1(a2);
    };
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) { (dw_ch(std::integral_constant<int, Cs>{}), ...); }(std::make_integer_sequence<int, 24>{});
    pw24(w2q, d, one, acc);
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) pend[m][e] = __builtin_fmaxf(acc[m][e], 0.f);
    pend_vo = (st_lane && j >= 2 && r - 1 < y1) ? ((r - 1) * W + x) * 8 : OOB;
#pragma unroll
    for (int c = 0; c < 24; ++c) cur[c] = nxt[c];
  };

  const int nsteps = R + 2;
  for (int j = 0; j < nsteps; j += 3) {
    step(j, tB, tC, tA);
    if (j + 1 < nsteps) step(j + 1, tC, tA, tB);
    if (j + 2 < nsteps) step(j + 2, tA, tB, tC);
  }
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){pend[m][0], pend[m][1]}), rsrc, pend_vo, a.dst_off[2 * m], 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){pend[m][2], pend[m][3]}), rsrc, pend_vo, a.dst_off[2 * m + 1], 0);
  }
}

// ----------------------------------------------------------------------------
// stride-2 ShuffleV2 block 24 -> 48 (stage2.0): proj = dw3x3s2+BN -> pw+BN+ReLU, main = pw1+BN+ReLU -> dw3x3s2+BN -> pw2+BN+ReLU
// ----------------------------------------------------------------------------
// A lane owns output column ox, i.e. input columns 2ox and 2ox+1 (one aligned 16-byte load per pair plane and
// input row gives both columns of two channels); input column 2ox-1 is the left neighbour's second column,
// so the dx = 0 column sums travel one lane to the right (DPP row_shr, zero fill = left padding) and only a
// LEFT halo lane is needed: 44 output columns = strips of 16 + 15 + 13.  A lane slides down its column one
// output row (two input rows) per step; input rows are consumed as they arrive - pw1 (main role) and the
// vertical taps are accumulated into the column sums S (dx = 1, 2) and Q (dx = 0) of the current output
// row, the odd input row is kept as the dy = 0 row of the next one - so no t or raw rows pile up.
// The two branches need ~250 registers of state each, so they are two wave ROLES over the same units
// (role = workgroup parity; both read the same input rows through L2) writing disjoint output slots.
constexpr int S2PX_PW_A = 0, S2PX_PW_B = 640, S2PX_TAPS_MAIN = 1280, S2PX_TAPS_PROJ = 640;

// proj role: its depthwise inputs are registers written by buffer loads and read ONLY by inline asm - hipcc inserts
// s_waitcnt for its own instructions' operands, not for asm operands, so without this the FMAs could run on a row
// that has not landed yet (it "worked" only because the loads are issued half a step ahead).  The wait is an asm
// statement that takes the row's 12 registers as read-write operands, so no reader can be scheduled above it.
// When a row is due the only younger loads are the 12 of the other row buffer: vmcnt(12) is exact if stores are not
// counted in vmcnt and merely conservative if they are.
#define YFV2_WAIT_ROW(Z)                                                                                                             \
  asm volatile("s_waitcnt vmcnt(12)"                                                                                                  \
               : "+v"(Z[0]), "+v"(Z[1]), "+v"(Z[2]), "+v"(Z[3]), "+v"(Z[4]), "+v"(Z[5]), "+v"(Z[6]), "+v"(Z[7]), "+v"(Z[8]), "+v"(Z[9]), \
                 "+v"(Z[10]), "+v"(Z[11]))
template <bool MAIN>
__device__ __forceinline__ void s2px_body(const S2PxArgs& a, int wid) {
  constexpr int ROLE = MAIN ? 1 : 0;
  const int IH = a.IH, IW = a.IW, OH = IH >> 1, OW = IW >> 1;
  const int nstrips = a.nstrips, nb = a.nb, R = a.R;
  const int units = nstrips * nb, wpi = (units + 3) >> 2;
  const int b = __builtin_amdgcn_readfirstlane(wid / wpi), wi = __builtin_amdgcn_readfirstlane(wid - b * wpi);
  const int lane = threadIdx.x, l = lane & 15;
  const int uid = wi * 4 + (lane >> 4);
  const int band = uid % nb, strip = uid / nb;
  const int ox = 15 * strip + l;
  const bool xok = uid < units && ox < OW;
  const bool st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R, y1 = min(OH, y0 + R);
  const int OOB = (int)0x80000000;

  __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * a.in_stride), 0, a.in_records, 0x00020000);
  __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(a.act + (size_t)b * a.out_stride), 0, a.out_records, 0x00020000);
  const int plane = IH * IW * 8;                  // bytes per input pair plane

  const float* img = a.img[ROLE];
  float wA[10], wB[10], tq[54];
#pragma unroll
  for (int q = 0; q < 10; ++q) { wA[q] = img[S2PX_PW_A + q * 64 + lane]; wB[q] = MAIN ? img[S2PX_PW_B + q * 64 + lane] : 0.f; }
#pragma unroll
  for (int q = 0; q < 54; ++q) tq[q] = img[(MAIN ? S2PX_TAPS_MAIN : S2PX_TAPS_PROJ) + q * 64 + lane];
  const float one = 1.0f;

  // one input row: X[q] = {col 2ox: ch 2q, 2q+1 ; col 2ox+1: ch 2q, 2q+1}
  auto load_row = [&](int iy, f32x4 (&X)[12]) {
    const int vo = (xok && iy >= 0 && iy < IH) ? (iy * IW + 2 * ox) * 8 : OOB;
#pragma unroll
    for (int q = 0; q < 12; ++q) X[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, q * plane, 0));
  };
  // the branch's depthwise input of one input row: col = 0/1 -> 24 channels (main: pw1+BN+ReLU of the raw column, 0 outside the image)
  auto column = [&](const f32x4 (&X)[12], int col, float lim, float (&v)[24]) {
    float raw[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) raw[k] = X[k >> 1][2 * col + (k & 1)];
    if constexpr (MAIN) {
      f32x4 acc[6];
      pw24(wA, raw, one, acc);
#pragma unroll
      for (int c = 0; c < 24; ++c) v[c] = __builtin_amdgcn_fmed3f(acc[c >> 2][c & 3], 0.f, lim);
    } else {
#pragma unroll
      for (int c = 0; c < 24; ++c) v[c] = raw[c];
    }
  };
#define YFV2_TQ(c, t) tq[((c) * 9 + (t)) >> 2]
#define YFV2_TK(c, t) (((c) * 9 + (t)) & 3)
  // vertical tap row DY of the 3x3 (taps index dy*3 + dx): S += w[dy][1]*v0 + w[dy][2]*v1, Q += w[dy][0]*v1
  auto acc_row = [&](auto dyc, const float (&v0)[24], const float (&v1)[24], float (&S)[24], float (&Q)[24]) {
    constexpr int DY = decltype(dyc)::value;
    [&]<int... Cs>(std::integer_sequence<int, Cs...>) {
      ((DY == 0 ? (void)(S[Cs] = quad_mul<YFV2_TK(Cs, 1)>(YFV2_TQ(Cs, 1), v0[Cs]), Q[Cs] = quad_mul<YFV2_TK(Cs, 0)>(YFV2_TQ(Cs, 0), v1[Cs]),
                         quad_fmac1<YFV2_TK(Cs, 2)>(S[Cs], YFV2_TQ(Cs, 2), v1[Cs]))
                : (void)quad_fmac3<YFV2_TK(Cs, DY * 3 + 1), YFV2_TK(Cs, DY * 3), YFV2_TK(Cs, DY * 3 + 2)>(
                      S[Cs], Q[Cs], YFV2_TQ(Cs, DY * 3 + 1), YFV2_TQ(Cs, DY * 3), YFV2_TQ(Cs, DY * 3 + 2), v0[Cs], v1[Cs])), ...);
    }(std::make_integer_sequence<int, 24>{});
  };

  f32x4 X[12], Y[12];
  float T0[24], T1[24];                            // the odd input row above the current output row (dy = 0)
  {
    const int iy = 2 * y0 - 1;
    load_row(iy, X);
    load_row(iy + 1, Y);
    const float lim = (xok && iy >= 0) ? __builtin_inff() : 0.f;
    if constexpr (!MAIN) YFV2_WAIT_ROW(X);         // (MAIN: the rows go through MFMA builtins, which get the compiler's own waits)
    column(X, 0, lim, T0);
    column(X, 1, lim, T1);
    load_row(iy + 2, X);
  }
  const float limx = xok ? __builtin_inff() : 0.f;
  // (proj role: stores are deferred to the next step like the stem's)
  f32x4 pend[6];
  int pend_vo = OOB;
#pragma unroll
  for (int m = 0; m < 6; ++m) pend[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto flush = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){pend[i >> 1][2 * (i & 1)], pend[i >> 1][2 * (i & 1) + 1]}), rout, pend_vo, a.st2_off[ROLE][i], 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float t = pend[4 + (i >> 2)][i & 3];   // by value: bit_cast on a vector ELEMENT lvalue reads element 0 (hipcc 7.2)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), rout, pend_vo, a.st1_off[ROLE][i], 0);
    }
  };
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
    float S[24], Q[24], v0[24], v1[24];
    if constexpr (!MAIN) { flush(); __builtin_amdgcn_sched_barrier(0); }
    acc_row(std::integral_constant<int, 0>{}, T0, T1, S, Q);
    // even input row 2oy (in Y), dy = 1
    if constexpr (!MAIN) YFV2_WAIT_ROW(Y);
    column(Y, 0, limx, v0);
    column(Y, 1, limx, v1);
    acc_row(std::integral_constant<int, 1>{}, v0, v1, S, Q);
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 2, Y);                       // next step's even row
    __builtin_amdgcn_sched_barrier(0);
    // odd input row 2oy+1 (in X), dy = 2; it is the next output row's dy = 0 row
    if constexpr (!MAIN) YFV2_WAIT_ROW(X);
    column(X, 0, limx, T0);
    column(X, 1, limx, T1);
    acc_row(std::integral_constant<int, 2>{}, T0, T1, S, Q);
    __builtin_amdgcn_sched_barrier(0);
    load_row(2 * oy + 3, X);
    __builtin_amdgcn_sched_barrier(0);
    float d[24];
#pragma unroll
    for (int c = 0; c < 24; ++c) { dpp_src_ready(Q[c]); d[c] = S[c] + row_shr1(Q[c]); }
    f32x4 acc[6];
    pw24(MAIN ? wB : wA, d, one, acc);
    const int vo = (st_lane && oy < y1) ? (oy * OW + ox) * 8 : OOB;
#pragma unroll
    for (int m = 0; m < 6; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) pend[m][e] = __builtin_fmaxf(acc[m][e], 0.f);
    pend_vo = vo;
    if constexpr (MAIN) flush();
  }
  if constexpr (!MAIN) flush();
#undef YFV2_TQ
#undef YFV2_TK
}
#undef YFV2_QP
#undef YFV2_WAIT_ROW

// The two roles are two kernels launched back to back, each with its own decomposition (main: 5 row bands per image
// = 960 one-wave workgroups at 256 images; proj: 4 bands).  A proj-role wave has almost no arithmetic between its
// row loads and spends its life in memory latency: timed alone it takes 50 us, the main role 58-63 us.  As two roles
// of ONE kernel (each wave holding a whole SIMD: 256 + 31 registers) they added up to 108 us AND left the next
// launch 16 us slower than its twins; as two kernels the pair costs 116 us and that penalty is gone (-8 us net).
// Tried and measured worse: proj compiled for two waves per SIMD (needs 84 bytes of scratch: 143-165 us for the
// pair), more/shorter proj bands (122-130 us).  A proj wave cannot share a SIMD with a main wave either
// (287 + >=232 registers > 512), so overlapping the two kernels on two streams has no room to work with.
template <bool MAIN>
__device__ __forceinline__ void s2px_dispatch(const S2PxArgs& a) {
  // Workgroup ids are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of waves, image by image.
  // Neighbouring strips overlap by one column and are not 128-byte aligned (a strip is 240 input bytes wide), and
  // both roles read the same rows: waves that run side by side behind ONE L2 turn those re-reads into hits
  // (measured fabric reads: 598 MB per launch without this ordering, 299 MB with it, for 190 MB of input).
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  s2px_body<MAIN>(a, wid);
}
__global__ __launch_bounds__(64, 1) void s2px_main_kernel(S2PxArgs a) { s2px_dispatch<true>(a); }
__global__ __launch_bounds__(64, 1) void s2px_proj_kernel(S2PxArgs a) { s2px_dispatch<false>(a); }

void yfv2_launch_s2px(const S2PxArgs& a0, hipStream_t s) {
  if (a0.img16) { yfv2_launch_s2h(a0, s); return; }   // both branches in one wave, input read once (yfv2_stage2h.hip)
  const int OW = a0.IW / 2, OH = a0.IH / 2;
  for (int role = 0; role < 2; ++role) {
    S2PxArgs a = a0;
    a.nstrips = OW <= 16 ? 1 : (OW - 1 + 14) / 15;
    a.nb = role ? 5 : 4;
    a.R = (OH + a.nb - 1) / a.nb;
    a.nb = (OH + a.R - 1) / a.R;
    const int units = a.nstrips * a.nb;
    const dim3 grid(a.B * ((units + 3) / 4));
    if (role) YFV2_LAUNCH(s2px_main_kernel, grid, dim3(64), 0, s, a);
    else YFV2_LAUNCH(s2px_proj_kernel, grid, dim3(64), 0, s, a);
  }
}

bool yfv2_s1px_supported(int H, int W) { return H >= 8 && W >= 16 && (long)48 * H * W * 4 < (1L << 28); }

void yfv2_launch_s1px(const S1PxArgs& a0, hipStream_t s) {
  if (a0.img16) { yfv2_launch_s1h(a0, s); return; }   // both pointwise convs on the f16 matrix cores (yfv2_stage2h.hip)
  S1PxArgs a = a0;
  a.nstrips = a.W <= 16 ? 1 : (a.W - 2 + 13) / 14;
  a.nb = 5;   // 3 strips x 5 bands = 15 units = 4 waves per image: 1024 waves at 256 images, 11 steps each (4 bands: 768 waves x 13 steps measured 7 % slower, 6 or 8 bands 15-30 % slower)
  a.R = (a.H + a.nb - 1) / a.nb;
  const int units = a.nstrips * a.nb;
  YFV2_LAUNCH(s1px_kernel, dim3(a.B * ((units + 3) / 4)), dim3(64), 0, s, a);
}
