// yfv2_towerh.hip - the FPN towers (model/fpn.py:12-25 DWConvblock, behaviour only) and the output convs they feed
// (model/detector.py:25-31), round 3's replacement of tower2_kernel (yfv2_block.hip; kept for YFV2_BF6=0 and for maps
// this kernel's lane grid does not cover).
//
// One launch = one half of a DWConvblock: dw5x5 (pad 2) + BN + ReLU -> pw 72->72 + BN [-> biased output conv, NCHW].
// One workgroup of eight waves per image, the image's 72 channels walked in five chunks of 16 (four channel quads).
//
// What was wrong with tower2_kernel (per-wave stamps): its depthwise read every tap of every output from
// LDS - lane = (pixel of a 16-pixel MFMA tile, channel quad), 25 window reads + 25/NT filter reads of 16 bytes per output:
// 125 ds_read_b128 per wave and chunk, the LDS port busy 6.8 k of a 10.4 k-cycle chunk period at 22x22, and its pointwise
// split both operands into three bf16 terms at run time (six MFMAs per product, ~30 VALU per filter fragment).  Here:
//   * depthwise phase: a wave = ONE channel quad x half of the image, a lane = a 2x2 patch of pixels (one pixel at
//     11x11).  The lane reads its 6x6 window once (36 reads for four outputs instead of 100) and the 25 taps + BN constants
//     of the quad are wave-uniform: scalar loads, SGPR operands of the packed FMAs - no filter traffic on the LDS port at
//     all (the table is pulled into the scalar cache in the prologue, each tap row requested one window row ahead).  The
//     staged input slice is kept as [quad][row parity][row / 2][column parity: slots 0.. / 14..][column / 2] with a row
//     pitch of 27 16-byte slots, so that the 64 lanes of a read hit consecutive slots of ONE parity plane (patch-row
//     pitch 27 = 11 mod 16: the four 16-lane groups ds_read_b128 is serviced in are conflict-free for 11 patches per row).
//   * exchange: the BN'd, ReLU'd result is split ONCE into two fp16 terms (x16 first: fp16's absolute floor) and written
//     as one 16-byte slot {h1 x4, h2 x4} per (pixel, quad) - which is exactly the B operand (K = 4g..4g+3) of
//     v_mfma_f32_16x16x16_f16 for lane (pixel, g): one ds_read_b128 per pixel tile and chunk in the pointwise phase.
//   * pointwise phase: fp16x3 (yfv2_stem16.hip / yfv2_stage2h.hip): filters pre-split on the host into two fp16 terms
//     (scaled by 2^sw, largest entry near 2^14), the three products w1 x2 + w2 x1 + w1 x1 in 1.5 K=32 MFMAs per 16-channel
//     chunk (cross terms of a chunk in one instruction, main terms of two chunks in another), fp32 accumulators of a
//     wave's NT pixel tiles x five output-channel tiles live in registers across the chunks.  The scales are undone exactly
//     inside the BN scale.  Valid for |activation| < 4094 (as every fp16x3 kernel of the plan).
//   * chained output conv: the BN'd accumulator tile s IS the data fragment of chunk s (D layout = operand layout at
//     K = 16); split in place and used as the A operand - the product is computed TRANSPOSED, so that a lane ends up with
//     four consecutive pixels of one output channel: 16-byte stores straight into the NCHW logit tensors (dword stores
//     when H*W % 4 != 0).  tower2_kernel's form (lane = pixel, four channels: dword stores, 64 bytes per channel and
//     lane group) cost 33 k of the 84 k cycles of the 22x22 obj+cls launch.
// Two barriers per chunk (input slice ready / exchange ready); the next chunk's slice is in flight in registers during
// the pointwise phase, the next image's first slice during the last chunk.
#include "yfv2_internal.h"
#include <atomic>
#include <algorithm>
#include <array>
#include <cstddef>
#include <map>
#include <mutex>
#include <vector>

typedef _Float16 yfv2_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 yfv2_h8 __attribute__((ext_vector_type(8)));
typedef unsigned yfv2_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const f32x4 yfv2_cf4;   // constant address space: uniform loads become s_load

#define YFV2_WSTAMP(i) do { if (a.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.trace[64 + 32 * (threadIdx.x >> 6) + (i)] = (long long)__builtin_readcyclecounter(); } while (0)

namespace {

// f32x4 -> {h1.lo, h1.hi, h2.lo, h2.hi}: two fp16 terms per element, h1 + h2 = v to 2^-24 relative (v already carries 2^4)
__device__ __forceinline__ u32x4 split4(f32x4 v) {
  const yfv2_h4 t1 = __builtin_convertvector(v, yfv2_h4);                       // v_cvt_pk_f16_f32 (RN)
  const f32x4 r = v - __builtin_convertvector(t1, f32x4);                       // exact
  const yfv2_h4 t2 = __builtin_convertvector(r, yfv2_h4);
  const yfv2_u2 a = __builtin_bit_cast(yfv2_u2, t1), b = __builtin_bit_cast(yfv2_u2, t2);
  return (u32x4){a[0], a[1], b[0], b[1]};
}
// One 16-byte entry = {first fp16 term x4, second term x4} of four consecutive K positions.  v_mfma_f32_16x16x32_f16 gives a
// lane 8 K slots: with the filter entry {w1, w2} as A and the data entry swapped to {x2, x1} as B, ONE instruction adds the
// two cross products w1 x2 + w2 x1 of a 16-channel chunk; the main products w1 x1 of TWO chunks share another one.
__device__ __forceinline__ yfv2_h8 pack8(unsigned a0, unsigned a1, unsigned b0, unsigned b1) { return __builtin_bit_cast(yfv2_h8, (u32x4){a0, a1, b0, b1}); }
__device__ __forceinline__ f32x4 mfma_cross(u32x4 w, u32x4 x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(w[0], w[1], w[2], w[3]), pack8(x[2], x[3], x[0], x[1]), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_main2(u32x4 wa, u32x4 wb, yfv2_u2 xa, u32x4 xb, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(wa[0], wa[1], wb[0], wb[1]), pack8(xa[0], xa[1], xb[0], xb[1]), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_main1(u32x4 w, u32x4 x, f32x4 acc) {    // the odd chunk out: upper K half of A = 0 (B's is finite)
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(pack8(w[0], w[1], 0u, 0u), pack8(x[0], x[1], x[0], x[1]), acc, 0, 0, 0);
}

}  // namespace

// image (floats), packed by WeightPacker::image_towerh (yfv2_api.hip):
//   WP   [mt 5][s 5][64 lanes][4 dwords]   pointwise filter x 2^sw: dwords 0,1 = first fp16 term of W[16mt + l%16][16s + 4(l/16) .. +3], 2,3 = second
//   CS   [4][96]                           pw BN scale x 2^-(sw+4) | pw BN shift | output-conv bias | [0] = 2^-(swh+4)
//   WH   [m MH][s 5][64][4]                output conv x 2^swh, as WP
//   ---- the part above is copied to LDS ----
//   TAPS [s 5][quad 4][27][4]              depthwise taps t = ky*5 + kx of channels 16s + 4q .. +3, then BN scale x 16, BN shift x 16 (zeros past channel 71); 16 floats of padding
constexpr int TH_KC = 5, TH_C = 72;
constexpr int TH_CS = TH_KC * TH_KC * 256;
constexpr int TH_WH = TH_CS + 4 * 96;
__host__ __device__ constexpr int th_lds_img(int MH) { return TH_WH + MH * TH_KC * 256; }   // = offset of TAPS
constexpr int TH_Q = 27;   // parity-plane row pitch (16-byte slots)

constexpr int TH_XOFF = 13;   // the odd-column plane of a row starts here (13 of the 27 slots of a row are used by each parity).  13, not 14: the
                              // staging stores (eight consecutive pixels per 8-lane store group: slots k, 13+k, k+1, 14+k, ..) then collide in one
                              // bank pair instead of two
// exchange-buffer slot of pixel q: bit 0 flipped in every other run of eight.  A 2x2-patch lane stores pixels two apart (eight
// lanes of a ds_write_b128 group: slots q, q+2, .. q+14 - the same four bank quads twice); with the flip the second four land on
// the odd quads.  A pixel tile's 16 consecutive slots stay a permutation of themselves: the pointwise reads are unaffected.
__device__ __forceinline__ int th_xslot(int q) { return q ^ ((q >> 3) & 1); }

template <int PS> struct ThGeom {
  static constexpr int YH = PS == 2 ? 13 : 16;                // plane rows: (H + 4) / PS rounded up (H <= 22 / H <= 11)
  static constexpr int SP = (YH * TH_Q + 15) & ~15;           // (quad, row parity) plane size (slots)
  static constexpr int TIN_SLOTS = 4 * PS * SP;
};

template <int MH, int PS, int NT>
__global__ __launch_bounds__(512) void towerh_kernel(TowerJobs jobs) {   // runs jobs.j[0], or (jobs.par) INDEPENDENT jobs side by side:
                                                                        // workgroups [k gpj, (k + 1) gpj) run job k (job LISTS - one workgroup running
                                                                        // dependent jobs in turn - are towers_kernel's, below)
  struct { int B, H, W; long long* trace; } a;                 // geometry, batch and trace buffer are the same for every job
  a.B = jobs.j[0].B; a.H = jobs.j[0].H; a.W = jobs.j[0].W; a.trace = jobs.j[0].trace;
  constexpr int KC = TH_KC, C = TH_C, WS = PS + 4, NPAR = PS * PS, Q = TH_Q, SP = ThGeom<PS>::SP, XOFF = TH_XOFF;
  constexpr int XP = 16 * NT * 8;                              // exchange slots per quad
  constexpr int NPF = NT;                                      // staged 16-byte pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* WP_ = lds;
  float* CS = lds + TH_CS;
  float* WH = lds + TH_WH;
  float* XB = lds + th_lds_img(MH);
  float* TIN = XB + 4 * XP * 4;
  const int H = a.H, W = a.W, HW = H * W;
  const float invW = 1.0f / (float)W;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Independent jobs in one launch (the a halves of the cls and reg towers both read the FPN map, the b halves each their own a
  // half): a CU's next workgroup starts the moment its current one ends, instead of the whole chip draining between two launches.
  const int gpj = jobs.par ? jobs.gpj : (int)gridDim.x;        // workgroups per job = the image loop's stride
  const int bid = blockIdx.x;
  const int jidx = __builtin_amdgcn_readfirstlane(jobs.par ? (bid >= gpj) + (bid >= 2 * gpj) + (bid >= 3 * gpj) : 0);
  const int grid = gpj;
  YFV2_WSTAMP(0);

  // ---- staging map: piece i of a chunk -> (pixel, quad): eight consecutive lanes = eight consecutive pixels of one quad, the
  // next three 8-lane groups the other quads of the same pixels (a wave reads whole 64-byte channel runs)
  int s_src[NPF], s_dst[NPF];                                  // s_dst < 0: no such pixel
  const int my_c4 = (tid >> 3) & 3;
#pragma unroll
  for (int j = 0; j < NPF; ++j) {
    const int i = tid + j * 512;
    const int px = (i & 7) + 8 * (i >> 5);
    const bool ok = px < HW;
    const int y = ok ? yfv2_fdiv(px, invW) : 0, x = ok ? px - y * W : 0;
    const int yy = y + 2, xx = x + 2;
    s_src[j] = ok ? px * C : 0;                                // (a pixel that does not exist loads pixel 0 and stores nothing)
    s_dst[j] = ok ? (((my_c4 * PS + (yy % PS)) * SP + (yy / PS) * Q + (xx % PS) * XOFF + xx / PS) * 4) : -1;
  }
  // straight-line loads (no branch, no select between issue and use: a conditional load makes the compiler wait for it
  // on the spot).  Channels past 71 (quads 2, 3 of the last chunk) load channel 0 and are zeroed when stored.
  auto stage_load = [&](const float* in, int bb, int sl, f32x4 (&pre)[NPF]) {
    const int ch = 16 * sl + 4 * my_c4;
    const float* img = in + (size_t)bb * HW * C + (ch < C ? ch : 0);
#pragma unroll
    for (int j = 0; j < NPF; ++j) pre[j] = *reinterpret_cast<const f32x4*>(img + s_src[j]);
  };
  auto stage_store = [&](int sl, const f32x4 (&pre)[NPF]) {
    const bool live = 16 * sl + 4 * my_c4 < C;
#pragma unroll
    for (int j = 0; j < NPF; ++j)
      if (s_dst[j] >= 0) *reinterpret_cast<f32x4*>(TIN + s_dst[j]) = live ? pre[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  const int qq = wv >> 1, hh = wv & 1;                         // depthwise role: quad qq of the chunk, half hh of the patches
  // ---- depthwise role: lane -> patch (py, pxx)
  const int PWn = (W + PS - 1) / PS, PHn = (H + PS - 1) / PS;
  const int pid = 64 * hh + lane;
  const bool pvalid = pid < PWn * PHn;
  const int pidc = pvalid ? pid : 0;
  const int py = yfv2_fdiv(pidc, 1.0f / (float)PWn), pxx = pidc - py * PWn;
  const float* win = TIN + ((qq * PS) * SP + py * Q + pxx) * 4;             // window (r, c): + ((r % PS) * SP + (r / PS) * Q + (c % PS) * XOFF + c / PS) * 4
  int xdst[NPAR];                                                           // exchange slot of the patch's pixels (float offset), -1 = outside
#pragma unroll
  for (int k = 0; k < NPAR; ++k) {
    const int y = PS * py + k / PS, x = PS * pxx + k % PS;
    xdst[k] = (pvalid && y < H && x < W) ? (qq * XP + th_xslot(y * W + x)) * 4 : -1;
  }
  // ---- pointwise role: NT pixel tiles of 16
  int opix[NT];
  bool pv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int q = 16 * (wv * NT + nt) + p;
    pv[nt] = q < HW;
    opix[nt] = q;                                                           // < XP always: pixels past HW read the zeroed exchange tail (th_xslot(q) stays inside q's aligned pair)
  }

  // (The four halves of a 22x22 level as a job list of ONE launch were measured: 145 us against 132 as four launches - a job is
  // 60-75 k cycles either way, and the wider kernel - every job on the six-tile LDS layout, both epilogues - spills.)
  {
  const __attribute__((address_space(4))) TowerArgs& ja = ((const __attribute__((address_space(4))) TowerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[jidx];
  const yfv2_cf4* taps = (const yfv2_cf4*)(ja.img16 + th_lds_img(MH));
  int b = bid - jidx * gpj;
  Yfv2Watch watch;                                             // range guard of the fp16x3 products (yfv2_internal.h)
  f32x4 pre[NPF];
  stage_load(ja.in, b < a.B ? b : 0, 0, pre);                  // the first slice flies during the job's prologue

  // ---- prologue.  Only what the first depthwise phase needs is waited for here: zeroed planes (the halo stays zero),
  // slice 0, the tap table in the scalar cache.  The filter image (th_lds_img floats: pointwise + output conv + constants)
  // is requested now but (11x11 kernels) lands in LDS only after that phase, before its closing barrier: its trip from
  // L2 / HBM is off the critical path of the job's start.
  constexpr int N4 = th_lds_img(MH) / 4, NIT = (N4 + 511) / 512;
  f32x4 tmp[NIT];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(ja.img16);
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * 512; tmp[k] = src[i < N4 ? i : 0]; }
    constexpr int NZ = 4 * XP + ThGeom<PS>::TIN_SLOTS;
    for (int i = tid; i < NZ; i += 512) reinterpret_cast<f32x4*>(XB)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the tap table of this wave's five quads -> scalar cache: one request per 64-byte line, all issued back to back (as C++
    // loads the compiler serialises them in groups of eight, a trip to L2 / HBM each)
    const float* tq = ja.img16 + th_lds_img(MH) + qq * 108;
#define YFV2_L(o) "s_load_dword s40, %0, " #o "\n\t"
#define YFV2_B(o) YFV2_L(o + 0) YFV2_L(o + 64) YFV2_L(o + 128) YFV2_L(o + 192) YFV2_L(o + 256) YFV2_L(o + 320) YFV2_L(o + 384)
    asm volatile(YFV2_B(0) YFV2_B(1728) YFV2_B(3456) YFV2_B(5184) YFV2_B(6912) "s_waitcnt lgkmcnt(0)" ::"s"(tq) : "s40", "memory");
#undef YFV2_B
#undef YFV2_L
  }
  constexpr bool DEFER = PS == 1;                              // (the 2x2-patch kernels have no registers to keep the image in flight)
  bool image_pending = DEFER;
  if constexpr (!DEFER) {
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * 512; if (i < N4) dst[i] = tmp[k]; }
  }
  __syncthreads();                                             // (the zeroing and the slice store below touch the same cells; the previous job is done with LDS)
  YFV2_WSTAMP(1);
  if (b < a.B) {
    stage_store(0, pre);
    stage_load(ja.in, b, 1, pre);
  }

  const int hmh = MH ? ja.mh : 0;                             // output channels of the merged matrix (MH > 0)
  for (; b < a.B; b += grid) {
    // MH == 0: acc[mt][nt] = output-channel tile mt x pixel tile nt of the 72 -> 72 pointwise conv (lane: 4 channels of 1 pixel).
    // MH > 0 (the half ends in an output conv): pointwise conv + BN and the output conv are both linear and nothing sits between
    // them (fpn.py:16-17,23-24; detector.py:25-31), so the host multiplied them into ONE matrix (WeightPacker::image_towerh:
    // Wh diag(scale) Wp, bias Wh shift + b, in double) - the kernel applies that matrix to the depthwise result directly,
    // TRANSPOSED (the data entries as A, rows = the tile's 16 pixels; the filter entries as B, columns = 16 output channels:
    // both operands have the same lane layout, so this is the same registers with the arguments swapped): acc[m][nt] = pixel
    // tile nt x output-channel tile m, lane (c = lane & 15, g) holds pixels 4g .. 4g+3 of ONE channel - a 16-byte run of the
    // NCHW tensor.  (Until round 4 the 72 x 72 product ran first and its BN'd, re-split result fed the output conv in the
    // epilogue: 25 + 30 filter tiles instead of 30 for obj + cls, 25 + 5 instead of 5 for reg, and 13-15 k cycles of a 67 k
    // cycle job spent in that epilogue.)
    constexpr int NA = MH == 0 ? KC : MH;
    f32x4 acc[NA][NT];
#pragma unroll
    for (int mt = 0; mt < NA; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    yfv2_u2 xprev[NT];                                                     // first terms of the even chunk, for the pair's main-product MFMA
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xprev[nt] = (yfv2_u2){0u, 0u};
#pragma unroll 1
    for (int s = 0; s < KC; ++s) {
      const bool dw_on = 4 * s + qq < C / 4;
      const yfv2_cf4* tp = taps + (dw_on ? s * 4 + qq : 0) * 27;
      f32x4 wt[3][5];                                                       // tap rows ky % 3 (wave-uniform: SGPRs)
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) wt[0][kx] = tp[kx];                    // requested before the barrier
      __syncthreads();                                                      // slice s complete in TIN; exchange free
      YFV2_WSTAMP(2 + 3 * s);
      if (dw_on) {
        f32x4 d[PS][PS];
#pragma unroll
        for (int k = 0; k < NPAR; ++k) d[k / PS][k % PS] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // two window rows and three tap rows in flight (left alone the scheduler hoists all 36 reads - 144 registers on top
        // of the 80 accumulators - and all 27 scalar loads - more SGPRs than there are)
        f32x4 row[2][WS];
        auto load_row = [&](int r, f32x4 (&dst)[WS]) {
#pragma unroll
          for (int c = 0; c < WS; ++c) dst[c] = *reinterpret_cast<const f32x4*>(win + ((r % PS) * SP + (r / PS) * Q + (c % PS) * XOFF + c / PS) * 4);
        };
        load_row(0, row[0]);
#pragma unroll
        for (int r = 0; r < WS; ++r) {
          if (r + 1 < WS) load_row(r + 1, row[(r + 1) & 1]);
          if (r + 1 < 5) {
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) wt[(r + 1) % 3][kx] = tp[(r + 1) * 5 + kx];
          }
#pragma unroll
          for (int dy = 0; dy < PS; ++dy) {
            const int ky = r - dy;
            if (ky < 0 || ky >= 5) continue;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
              for (int dx = 0; dx < PS; ++dx) d[dy][dx] = __builtin_elementwise_fma(row[r & 1][dx + kx], wt[ky % 3][kx], d[dy][dx]);
          }
          // pins this row's FMAs here (the DAG scheduler otherwise sinks all of them below all the reads)
          if constexpr (PS == 2) asm volatile("" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) :: "memory");
          else asm volatile("" : "+v"(d[0][0]) :: "memory");
        }
        const f32x4 sc = tp[25], sh = tp[26];
#pragma unroll
        for (int k = 0; k < NPAR; ++k) {
          f32x4 u = __builtin_elementwise_fma(d[k / PS][k % PS], sc, sh);
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = u[e] > 0.f ? u[e] : 0.f;                    // (the BN constants carry the 2^4)
          if (xdst[k] >= 0) *reinterpret_cast<u32x4*>(XB + xdst[k]) = split4(u);
        }
      }
      if (DEFER && image_pending) {                                         // first chunk of the job's first image only
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
#pragma unroll
        for (int k = 0; k < NIT; ++k) { const int i = tid + k * 512; if (i < N4) dst[i] = tmp[k]; }
        image_pending = false;
      }
      __syncthreads();                                                      // exchange complete; TIN free
      YFV2_WSTAMP(3 + 3 * s);
      u32x4 xb[NT];                                                         // (requested before the staging traffic below)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const u32x4*>(XB + (g * XP + th_xslot(opix[nt])) * 4);
      if (s == 0) YFV2_WSTAMP(22);
      // the next slice -> TIN, the one after into registers (of the next image after the last chunk)
      if (s + 1 < KC || b + grid < a.B) stage_store(s + 1 < KC ? s + 1 : 0, pre);
      {
        int ns = s + 2, nb = b;
        if (ns >= KC) { ns -= KC; nb += grid; }
        stage_load(ja.in, nb < a.B ? nb : b, ns, pre);
      }
      if constexpr (MH == 0) {
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) {
        const u32x4 wf = *reinterpret_cast<const u32x4*>(WP_ + ((mt * KC + s) * 64 + lane) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_cross(wf, xb[nt], acc[mt][nt]);
        if (s & 1) {
          const u32x4 w0 = *reinterpret_cast<const u32x4*>(WP_ + ((mt * KC + s - 1) * 64 + lane) * 4);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_main2(w0, wf, xprev[nt], xb[nt], acc[mt][nt]);
        } else if (s == KC - 1) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_main1(wf, xb[nt], acc[mt][nt]);
        }
      }
      } else {
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        if (16 * m < hmh) {                                                 // (wave-uniform: a narrower output conv leaves zero tiles)
          const u32x4 wf = *reinterpret_cast<const u32x4*>(WH + ((m * KC + s) * 64 + lane) * 4);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_cross(xb[nt], wf, acc[m][nt]);
          if (s & 1) {
            const u32x4 w0 = *reinterpret_cast<const u32x4*>(WH + ((m * KC + s - 1) * 64 + lane) * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = mfma_main2((u32x4){xprev[nt][0], xprev[nt][1], 0u, 0u}, xb[nt], (yfv2_u2){w0[0], w0[1]}, wf, acc[m][nt]);
          } else if (s == KC - 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_main1(xb[nt], wf, acc[m][nt]);
          }
        }
      }
      }
      if (!(s & 1)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xprev[nt] = (yfv2_u2){xb[nt][0], xb[nt][1]};
      }
      YFV2_WSTAMP(4 + 3 * s);
    }
    YFV2_WSTAMP(17);
    if constexpr (MH == 0) {
      // pointwise BN (no ReLU: fpn.py:16-17,23-24); the scale carries 2^-(sw+4)
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 0 * 96 + 16 * mt + 4 * g);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 1 * 96 + 16 * mt + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_elementwise_fma(acc[mt][nt], sc, sh);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) watch.see(acc[0][nt][0]);   // a depthwise result beyond fp16's range: NaN in every channel of its pixel
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (!pv[nt]) continue;
        float* dst = ja.out + ((size_t)b * HW + opix[nt]) * C;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt)
          if (16 * mt + 4 * g < C) *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * g) = acc[mt][nt];
      }
    } else {
      const float us = CS[3 * 96];                                          // 2^-(swh+4)
      const bool vec = (HW & 3) == 0;                                       // (alignment of every channel plane)
      // (both plane pointers in scalar registers BEFORE the lane-dependent choice: left to itself the compiler turns the choice
      // between two kernel-argument fields into ONE vector load from a chosen address - a full load latency in front of every
      // output tile's stores)
      float* hn0 = ja.nchw0; float* hn1 = ja.nchw1;
      asm volatile("" : "+s"(hn0), "+s"(hn1));
      const int hsplit = ja.split;
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        if (16 * m >= hmh) break;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) watch.see((acc[m][nt][0] + acc[m][nt][1]) + (acc[m][nt][2] + acc[m][nt][3]));   // transposed: a lane's four values are four PIXELS
        const int co = 16 * m + p;
        if (co < hmh) {
          const float bias = CS[2 * 96 + co];
          float* plane = co < hsplit ? hn0 + ((size_t)b * hsplit + co) * HW : hn1 + ((size_t)b * (hmh - hsplit) + (co - hsplit)) * HW;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int px0 = 16 * (wv * NT + nt) + 4 * g;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(acc[m][nt][r], us, bias);
            if (vec) {
              if (px0 < HW) *reinterpret_cast<f32x4*>(plane + px0) = y;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (px0 + r < HW) plane[px0 + r] = y[r];
            }
          }
        }
      }
    }
    YFV2_WSTAMP(18);
  }
  watch.report(ja.nonfinite);
  }
}

// ============================================================================
// Small maps (up to 11x11): towers_kernel - the whole image in one pass, depthwise with lane = (row, channel)
// ============================================================================
// At 11x11 a tower half is 121 pixels.  The chunked kernel above, at one pixel per lane, reads every input value 25 times from
// LDS (871 KB per job through one CU's LDS port: ~10 k cycles, measured) and goes through eleven barriers and five dependent
// rounds of global loads per job: the four jobs of an image take 44 us even at batch 1.  Here:
//   * the image is staged ONCE as it lies in memory ([pixel][72 channels]: one contiguous 35 KB run, fully coalesced);
//   * depthwise: a lane = (output row y, channel c), 792 lane units = 13 wave units over eight waves (two per wave).  The
//     lane reads the five input rows of its channel into registers (55 values, one LDS round trip) and slides the taps over
//     them: an input value is read five times (once per output row it feeds) instead of 25, every read is 64 consecutive
//     channels of one pixel (256 contiguous bytes: conflict-free), the 25 taps and the BN constants are per-lane registers
//     read from the tap table the job's filter image brings into LDS;
//   * exchange: BN + ReLU results as fp32 [pixel][72] (TS_CP below: the pitch that keeps the pointwise phase's 16-byte reads of 16
//     pixels x 4 lane groups conflict-free in ds_read_b128's service groups), split into two fp16 terms by the reader;
//   * pointwise (K = 72 in one go) and the transposed output conv as in towerh_kernel.
// Three barriers per job.
// exchange pitch (floats per pixel).  The pointwise phase reads 16 bytes at slot 18 p + 4 chunk + g per lane (p, g); ds_read_b128's service
// groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: tools/ubench/ldsgroups.hip) mix the pixels {0-3, 12-15} of channel group g with
// the pixels {4-11} of g + 1: a pitch of 2 mod 16 slots puts the former on even, the latter on odd slot residues, each set distinct.
// (Until round 6: 76 floats = 19 slots, free of conflicts for 16 CONSECUTIVE lanes - which is not how the instruction is serviced.)
constexpr int TS_CP = 72;
template <int MH>
__global__ __launch_bounds__(512) void towers_kernel(TowerJobs jobs) {
  struct { int B, H, W; long long* trace; } a;
  a.B = jobs.j[0].B; a.H = jobs.j[0].H; a.W = jobs.j[0].W; a.trace = jobs.j[0].trace;
  constexpr int KC = TH_KC, C = TH_C, NQ = C / 4, MAXW = 11, CP = TS_CP, MAXPX = 128;
  constexpr int TAPS_FL = NQ * 27 * 4;
  constexpr int LDS_IMG = th_lds_img(MH) + TAPS_FL;               // pointwise filter, constants, output-conv filter, tap table: one straight copy
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* WP_ = lds;
  float* CS = lds + TH_CS;
  float* WH = lds + TH_WH;
  float* IN = lds + LDS_IMG;                                      // [row + 2][MAXW][72]: two zero rows above and below the image, zero columns right of it
                                                                  // (a fixed pitch: every row read of the depthwise is unconditional, two pixels per ds_read2_b64)
  float* X32 = IN + (MAXW + 4) * MAXW * C;                        // [pixel][TS_CP]; rows past H*W stay zero
  const int H = a.H, W = a.W, HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grid = gridDim.x;
  YFV2_WSTAMP(0);
  const int opix = 16 * wv + p;                                   // pointwise tile: 16 pixels per wave
  const bool pv = opix < HW;
  constexpr int N4 = LDS_IMG / 4, NIT = (N4 + 511) / 512;
  const __attribute__((address_space(4))) TowerArgs* kj = (const __attribute__((address_space(4))) TowerArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  Yfv2Watch watch;                                                // range guard of the fp16x3 products (yfv2_internal.h)
  f32x4 tmp[NIT];                                                 // the NEXT job's filter image, in flight while this job computes
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(kj[0].img16);
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * 512; tmp[k] = src[i < N4 ? i : 0]; }
  }
  // an item's image: requested while the item before it computes (its own first thing otherwise: 3 k cycles of exposed latency per job)
  constexpr int NPF = (MAXPX * NQ + 511) / 512;                   // 16-byte pieces of the image per thread (5)
  f32x4 pre[NPF];
  auto load_pre = [&](const float* in, int bb) __attribute__((always_inline)) {
    const f32x4* img = reinterpret_cast<const f32x4*>(in + (size_t)bb * HW * C);
#pragma unroll
    for (int j = 0; j < NPF; ++j) { const int i = tid + j * 512; pre[j] = img[i < HW * NQ ? i : 0]; }
  };
  bool have_pre = false;
#pragma unroll
  for (int j = 0; j < NPF; ++j) pre[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!(kj[0].chain & 1) && (int)blockIdx.x < a.B) { load_pre(kj[0].in, blockIdx.x); have_pre = true; }
  // (the zeroing behind the requests, not in front of them)
  for (int i = tid; i < ((MAXW + 4) * MAXW * C + MAXPX * CP) / 4; i += 512) reinterpret_cast<f32x4*>(IN)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int ji = 0; ji < jobs.n; ++ji) {
  const __attribute__((address_space(4))) TowerArgs& ja = kj[ji];
  const float* tapsf = lds + th_lds_img(MH);                      // [quad][27][4]: 25 taps, BN scale x 16, BN shift x 16 (in LDS: as per-lane
                                                                  // global gathers the 54 loads of a wave cost the job ~7 k cycles of address traffic)
#pragma unroll 1
  for (int b = blockIdx.x; b < a.B; b += grid) {
    // ---- everything the job needs from memory is requested at once
    const bool in_lds = (ja.chain & 1) != 0, out_lds = (ja.chain & 2) != 0;   // half a -> half b of a tower: the 72-channel tensor between them never leaves LDS
    if (!in_lds && !have_pre) load_pre(ja.in, b);
    const bool first_image = b == (int)blockIdx.x;
    // this lane's depthwise unit: (output row, channel PAIR) - H x 36 units <= 396 lanes, ONE round (round 6; until then (row, channel):
    // 792 units on 512 lanes, the second round 45 % empty, every FMA a plain one)
    const int UL = 64 * wv + lane;
    const bool uon = 64 * wv < H * (C / 2), ulane = UL < H * (C / 2);
    const int uy = yfv2_fdiv(ulane ? UL : 0, 1.0f / (float)(C / 2)), uc = 2 * ((ulane ? UL : 0) - uy * (C / 2));
    __syncthreads();                                              // the previous job / image is done with LDS
    // (every thread stores every register it carries, on every path - what is not wanted goes where nothing reads: behind a predicate the
    // compiler keeps "this register may still have a load in flight" alive on the other path and answers with s_waitcnt vmcnt(0) at the
    // next requests further down - i.e. it waits for the first prefetch before it issues the second: a full memory latency per job)
    float* const dump = X32 + MAXPX * CP + 4 * lane;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      const int i = tid + j * 512;
      const bool ok = !in_lds && i < HW * NQ;
      const int px = yfv2_fdiv(ok ? i : 0, 1.0f / (float)NQ), qd = i - px * NQ;
      const int y = yfv2_fdiv(px, 1.0f / (float)W), x = px - y * W;
      *reinterpret_cast<f32x4*>(ok ? IN + ((y + 2) * MAXW + x) * C + 4 * qd : dump) = pre[j];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * 512; *reinterpret_cast<f32x4*>(first_image && i < N4 ? lds + 4 * i : dump) = tmp[k]; }
    __syncthreads();                                              // image and filters in LDS
    YFV2_WSTAMP(1 + 4 * ji);
    // ---- prefetches, issued BETWEEN the depthwise rows (all 13 requests of a wave up front keep it 2-3 k cycles in front of the
    // vector-memory issue port before its first FMA: 8 waves x 13 KB at 64 bytes per cycle): (1) the next item's image, unless this
    // job is what produces it (an a half whose b half reads it back from memory); (2) the next job's filter image
    const bool pf_same = b + grid < a.B;
    const int pf_jn = pf_same ? ji : ji + 1, pf_bn = pf_same ? b + grid : (int)blockIdx.x;
    have_pre = pf_jn < jobs.n && !(kj[pf_jn].chain & 1) && (pf_same || kj[pf_jn].in != ja.out);
    const f32x4* pf_img = reinterpret_cast<const f32x4*>((have_pre ? kj[pf_jn].in : ja.in) + (size_t)(have_pre ? pf_bn : b) * HW * C);
    const f32x4* pf_flt = reinterpret_cast<const f32x4*>(kj[ji + 1 < jobs.n ? ji + 1 : ji].img16);
    const bool pf_filters = first_image && ji + 1 < jobs.n;
    auto prefetch = [&](int k) __attribute__((always_inline)) {     // k = 0 .. NPF + NIT - 1 (wave-uniform conditions)
      if (k < NPF) { if (have_pre) { const int i = tid + k * 512; pre[k] = pf_img[i < HW * NQ ? i : 0]; } }
      else if (k < NPF + NIT) { if (pf_filters) { const int i = tid + (k - NPF) * 512; tmp[k - NPF] = pf_flt[i < N4 ? i : 0]; } }
    };
    constexpr int PF_ROW = (NPF + NIT + 4) / 5;                   // requests per depthwise row
    YFV2_WSTAMP(17 + 3 * ji);
    // ---- depthwise: a lane = (output row, channel pair): packed FMAs on (c, c + 1) against the tap pair (the table holds a quad's
    // taps side by side: the pair is one 8-byte read), row by row with the next input row and tap row in flight.  Per output the
    // products are added in the order tap row, tap column - as before, bit for bit
    if (uon) {                                                    // wave-uniform
      const int y = uy, c = uc;
      const float* tl = tapsf + ((c >> 2) * 27) * 4 + (c & 3);
      auto ld_taps = [&](int r, f32x2 (&t)[5]) {
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) t[kx] = *reinterpret_cast<const f32x2*>(tl + 4 * (r * 5 + kx));
      };
      auto ld_row = [&](int r, f32x2 (&v)[MAXW]) {
        const float* rp = IN + (y + r) * MAXW * C + c;            // row y + r - 2 of the image: the halo rows and the columns right of the image are zero
#pragma unroll
        for (int x = 0; x < MAXW; ++x) v[x] = *reinterpret_cast<const f32x2*>(rp + x * C);
      };
      f32x2 acc[MAXW];
#pragma unroll
      for (int x = 0; x < MAXW; ++x) acc[x] = (f32x2){0.f, 0.f};
      f32x2 v[2][MAXW], tw[2][5];
      ld_row(0, v[0]); ld_taps(0, tw[0]);
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        if (r + 1 < 5) { ld_row(r + 1, v[(r + 1) & 1]); ld_taps(r + 1, tw[(r + 1) & 1]); }
        // (tap column outermost: eleven independent accumulators between two FMAs of one output - with the output outermost the
        // compiler keeps the source order and issues every output's five FMAs back to back, each waiting for the one before it)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
#pragma unroll
          for (int x = 0; x < MAXW; ++x) {
            const int xi = x + kx - 2;
            if (xi >= 0 && xi < MAXW) acc[x] = __builtin_elementwise_fma(v[r & 1][xi], tw[r & 1][kx], acc[x]);
          }
        // (two rows in flight: left alone the scheduler requests all five rows and tap rows first - 164 registers)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]) :: "memory");
#pragma unroll
        for (int k = 0; k < PF_ROW; ++k) prefetch(r * PF_ROW + k);
      }
      YFV2_WSTAMP(18 + 3 * ji);
      const f32x2 bsc = *reinterpret_cast<const f32x2*>(tl + 4 * 25), bsh = *reinterpret_cast<const f32x2*>(tl + 4 * 26);
      if (ulane) {
#pragma unroll
        for (int x = 0; x < MAXW; ++x)
          if (x < W) {
            f32x2 uu = __builtin_elementwise_fma(acc[x], bsc, bsh);                                            // (the BN constants carry the 2^4)
            uu[0] = uu[0] > 0.f ? uu[0] : 0.f; uu[1] = uu[1] > 0.f ? uu[1] : 0.f;
            *reinterpret_cast<f32x2*>(X32 + (y * W + x) * CP + c) = uu;
          }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NPF + NIT; ++k) prefetch(k);
    }
    YFV2_WSTAMP(19 + 3 * ji);
    __syncthreads();                                              // exchange complete
    YFV2_WSTAMP(2 + 4 * ji);

    // ---- pointwise: K = 72 in one go.  A half that ends in an output conv applies the host-merged matrix (output conv x BN x
    // pointwise conv, see towerh_kernel) to these operand entries directly, transposed.
    u32x4 xb[KC];
#pragma unroll
    for (int sc = 0; sc < KC; ++sc) {
      const f32x4 v = 16 * sc + 4 * g < C ? *reinterpret_cast<const f32x4*>(X32 + opix * CP + 16 * sc + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
      xb[sc] = split4(v);
    }
    if (MH == 0 || !ja.has_head) {
    f32x4 acc[KC];
#pragma unroll
    for (int mt = 0; mt < KC; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) {
        u32x4 wf[KC];
#pragma unroll
        for (int sc = 0; sc < KC; ++sc) wf[sc] = *reinterpret_cast<const u32x4*>(WP_ + ((mt * KC + sc) * 64 + lane) * 4);
#pragma unroll
        for (int sc = 0; sc < KC; ++sc) acc[mt] = mfma_cross(wf[sc], xb[sc], acc[mt]);
#pragma unroll
        for (int sc = 0; sc + 1 < KC; sc += 2) acc[mt] = mfma_main2(wf[sc], wf[sc + 1], (yfv2_u2){xb[sc][0], xb[sc][1]}, xb[sc + 1], acc[mt]);
        acc[mt] = mfma_main1(wf[KC - 1], xb[KC - 1], acc[mt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < KC; ++mt) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 0 * 96 + 16 * mt + 4 * g);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 1 * 96 + 16 * mt + 4 * g);
      acc[mt] = __builtin_elementwise_fma(acc[mt], sc, sh);
    }
    watch.see(acc[0][0]);
    YFV2_WSTAMP(3 + 4 * ji);
      if (pv) {                                                   // (every depthwise read of IN is behind the exchange barrier)
        const int oy = yfv2_fdiv(opix, 1.0f / (float)W);
        float* dst = out_lds ? IN + ((oy + 2) * MAXW + (opix - oy * W)) * C : ja.out + ((size_t)b * HW + opix) * C;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt)
          if (16 * mt + 4 * g < C) *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * g) = acc[mt];
      }
    } else {
      YFV2_WSTAMP(3 + 4 * ji);
      const float us = CS[3 * 96];
      const bool vec = (HW & 3) == 0;
      // (both plane pointers in scalar registers BEFORE the lane-dependent choice: left to itself the compiler turns the choice
      // between two kernel-argument fields into ONE vector load from a chosen address - a full load latency in front of every
      // output tile's stores)
      float* hn0 = ja.nchw0; float* hn1 = ja.nchw1;
      asm volatile("" : "+s"(hn0), "+s"(hn1));
      const int hmh = ja.mh, hsplit = ja.split;
      // two output tiles per step: two independent chains of eight MFMAs (one tile at a time, every MFMA waits for the one before it:
      // 1.5 k cycles per tile, 9.4 k for the six tiles of the cls tower's obj + cls conv)
      constexpr int TW = MH > 1 ? 2 : 1;
#pragma unroll 1
      for (int m = 0; m < MH; m += TW) {
        if (16 * m >= hmh) break;
        u32x4 wf[TW][KC];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int sc = 0; sc < KC; ++sc) wf[t][sc] = *reinterpret_cast<const u32x4*>(WH + (((m + t < MH ? m + t : m) * KC + sc) * 64 + lane) * 4);
        f32x4 hacc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) hacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sc = 0; sc < KC; ++sc)
#pragma unroll
          for (int t = 0; t < TW; ++t) hacc[t] = mfma_cross(xb[sc], wf[t][sc], hacc[t]);
#pragma unroll
        for (int sc = 0; sc + 1 < KC; sc += 2)
#pragma unroll
          for (int t = 0; t < TW; ++t) hacc[t] = mfma_main2(xb[sc], xb[sc + 1], (yfv2_u2){wf[t][sc][0], wf[t][sc][1]}, wf[t][sc + 1], hacc[t]);
#pragma unroll
        for (int t = 0; t < TW; ++t) hacc[t] = mfma_main1(xb[KC - 1], wf[t][KC - 1], hacc[t]);
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          if (16 * (m + t) >= hmh) break;
          watch.see((hacc[t][0] + hacc[t][1]) + (hacc[t][2] + hacc[t][3]));      // transposed: a lane's four values are four PIXELS
          const int co = 16 * (m + t) + p;
          if (co < hmh) {
            const float bias = CS[2 * 96 + co];
            float* plane = co < hsplit ? hn0 + ((size_t)b * hsplit + co) * HW : hn1 + ((size_t)b * (hmh - hsplit) + (co - hsplit)) * HW;
            const int px0 = 16 * wv + 4 * g;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(hacc[t][r], us, bias);
            // (a lane's four pixels as ONE 16-byte store also where the planes are only 4-byte aligned - 11x11 = 121 floats per plane:
            // global memory takes dword-aligned multi-dword stores; four predicated dword stores per tile were 24 scattered store
            // instructions per wave for the cls tower's six tiles)
            typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            if (vec || px0 + 3 < HW) {
              if (px0 < HW) *reinterpret_cast<f32x4_a4*>(plane + px0) = y;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (px0 + r < HW) plane[px0 + r] = y[r];
            }
          }
        }
      }
    }
    YFV2_WSTAMP(4 + 4 * ji);
  }
  __syncthreads();                                                // the next job reads what this one wrote for the same image: workgroup scope is
                                                                  // enough (same CU, same L1); an agent-scope __threadfence() here wrote back and
                                                                  // invalidated L2 on every XCD - 4x the launch time
  }
  watch.report(kj[0].nonfinite);
}

// ============================================================================
// towerp_kernel (round 6): the 2x2-patch maps (up to 22x22) with the depthwise on CHANNEL-PAIR units
// ============================================================================
// What the per-wave stamps and the disassembly of towerh_kernel<.., 2, 4> showed (profiles/r06_tower_stamps.txt): a job is 55-58 k
// cycles, the same 3.7 k per chunk in the depthwise phase and 2.5-3.4 k in the pointwise phase at ONE image as at 256 - on-CU latency,
// not memory: (1) the depthwise took its taps from the scalar cache row by row INSIDE the window loop; scalar loads and LDS reads
// share one counter and return out of order, so every `s_waitcnt` in that loop was lgkmcnt(0): the row just requested was waited
// for on the spot, six exposed LDS round trips per wave and chunk for 1.6 k cycles of packed FMAs per SIMD; (2) the pointwise phase
// read each filter fragment right in front of the four MFMAs that use it (ten exposed round trips), and with the chunk loop rolled
// the even / odd / last forms were run-time branches between MFMA groups, accumulator copies and `s_nop 7` included.  Here:
//   * depthwise unit = (channel PAIR, 64 patches): 25 tap pairs + the BN pair = 54 SGPRs, requested ONCE per chunk before the
//     chunk's first barrier - no scalar load is in flight while the window is read, the compiler counts the LDS reads down one by
//     one.  A wave owns one pair of the 16-channel chunk and runs both 64-patch rounds (last chunk, 4 live pairs: pair wv & 3,
//     round wv >> 2).  The staged slice is eight pair planes [pair][row 26][col 26] of 8-byte slots (zero halo of 2): a lane's 6x6
//     window is 18 ds_read_b128 (two adjacent columns x two channels each) instead of 36, all in flight at once (72 registers);
//     the 16-lane service groups of ds_read_b128 are conflict-free through a host-computed lane -> patch table (TowerJobs::lane_patch:
//     every group's patches have distinct 16-byte slot residues mod 16).  v_pk_fma_f32 on (channel, channel + 1) with the SGPR
//     tap pair: the same products in the same order as towerh_kernel - BIT-identical results.
//   * pointwise: the chunk loop is unrolled by pairs (even, odd) + the last, every form straight-line: all filter fragments of
//     the chunk requested up front (the depthwise's window registers are free by then), then the MFMAs back to back.
// Everything else (exchange layout, fp16x3 products, merged output matrix, epilogues) as in towerh_kernel above.
constexpr int TP_P = 26, TP_ROWS = 26, TP_PLANE = TP_ROWS * TP_P;     // pair plane: rows x pitch, 8-byte slots (5408 bytes = 32 mod 64: the two
                                                                       // planes a staged 16-byte piece is split into fall into different bank halves)
constexpr int TP_TIN_FL = 8 * TP_PLANE * 2;                            // floats
constexpr int TP_DUMP_FL = 2 * TP_PLANE + 64;                          // where staged pieces of pixels that do not exist go (both halves of a piece)
constexpr int TP_XS = 2 * 512 + 16;                                    // exchange buffer: [pair 8][pixel 512]{first fp16 terms of the pair's two channels, second terms}
                                                                       // (8 bytes), pair stride 4160 bytes = 64 mod 128: the pointwise phase's two lane groups of a
                                                                       // ds_read_b64 (channel groups g, g + 1 = pairs 2 apart) fall into different bank halves
constexpr int TP_XB_FL = 8 * TP_XS;
constexpr int TP_TAPS2 = TH_KC * 4 * 27 * 4 + 16;                      // offset of the pair tap table behind th_lds_img(MH): [s 5][pair 8][64]:
                                                                       // floats 2t + e = tap t of channel 16s + 2 pair + e, 50 + e = BN scale x 16, 52 + e = BN shift x 16
__host__ __device__ constexpr int tp_lds_floats(int MH) { return th_lds_img(MH) + TP_XB_FL + TP_TIN_FL + TP_DUMP_FL + 2 * 384; }
typedef float yfv2_f8 __attribute__((ext_vector_type(8)));
typedef float yfv2_f16v __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) const yfv2_f16v yfv2_cf16;
typedef __attribute__((address_space(4))) const yfv2_f8 yfv2_cf8;

template <int MH, bool BOTH>
__global__ __launch_bounds__(512) void towerp_kernel(TowerJobs jobs) {
  // BOTH (round 6, third step): ONE launch for a whole level - a workgroup's items are the four halves of its image, cls a, reg a, cls b,
  // reg b.  The b halves read what the a halves of the SAME workgroup wrote two items earlier (same CU, same L1, tens of thousands
  // of cycles and many barriers apart), so no launch boundary is needed between them: one prologue per image instead of two, no drain /
  // ramp between two launches.  An item's FORM (0: 72 -> 72 pointwise conv, a halves; 1: merged matrix, b halves) is then a
  // per-item property (TowerArgs::has_head); the two forms are two instantiations of the item body below.
  static_assert(!BOTH || MH > 0, "the b halves end in an output conv");
  struct { int B, H, W; long long* trace; } a;
  a.B = jobs.j[0].B; a.H = jobs.j[0].H; a.W = jobs.j[0].W; a.trace = jobs.j[0].trace;
  constexpr int KC = TH_KC, C = TH_C, NT = 4, P = TP_P;
  // form 0: the 72 -> 72 pointwise conv (accumulators = output-channel tiles x pixel tiles); form 1: the merged matrix, transposed (a lane
  // holds four pixels of one output channel).  Filter / accumulator tiles: KC / MH.  Without BOTH every item has the form MH implies.
  constexpr int FORM1 = MH == 0 ? 0 : 1;
  auto form_of = [&](int j) { return BOTH ? (int)(((const __attribute__((address_space(4))) TowerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[j].has_head != 0) : FORM1; };
  auto nf_of = [](int form) { return form == 0 ? KC : MH; };
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* WP_ = lds;
  float* WH = lds + TH_WH;
  float* XB = lds + th_lds_img(MH);
  float* TIN = XB + TP_XB_FL;
  const int H = a.H, W = a.W, HW = H * W;
  const float invW = 1.0f / (float)W;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work items of a workgroup (round 6, second step): with side-by-side jobs (jobs.par: the cls and the reg tower's halves) a workgroup
  // runs ALL of them for its image, one after the other - item (job j, image b) is followed by (j + 1, b), then (0, b + grid) - and an
  // item's last two chunks already request the next item's first slices, filter fragments, constants and taps: one prologue per
  // workgroup instead of one per job (12 k of a 47-60 k cycle job at batch 256), and the second job's input is the first one's
  // (the FPN map: L2 hits).
  // (Small batches - fewer items than CUs - get one workgroup per item instead: the launcher's grid is then nj x B.)
  const int nj = jobs.par ? jobs.n : 1;
  const bool spread = !BOTH && (int)gridDim.x == nj * a.B && nj > 1;     // one item per workgroup, job-major (independent jobs only)
  const int grid = (int)gridDim.x;
  const int bid = blockIdx.x;
  YFV2_WSTAMP(0);

  // ---- staging map (as towerh_kernel: eight consecutive lanes = eight consecutive pixels of one quad); a piece = (pixel, quad) is
  // stored as two 8-byte halves into the planes of pairs 2 quad and 2 quad + 1
  int s_dst[NT];
  const int my_c4 = (tid >> 3) & 3;
  const int px0 = (tid & 7) + 8 * (tid >> 5);                  // piece j = pixel px0 + 128 j
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int px = px0 + 128 * j;
    const bool ok = px < HW;
    const int y = ok ? yfv2_fdiv(px, invW) : 0, x = ok ? px - y * W : 0;
    s_dst[j] = ok ? (2 * my_c4 * TP_PLANE + (y + 2) * P + (x + 2)) * 2 : TP_TIN_FL + 4 * (tid & 15);
  }
  auto stage_load = [&](const float* in, int bb, int sl, f32x4 (&pre)[NT]) {
    const int ch = 16 * sl + 4 * my_c4;
    const float* img = in + (size_t)bb * HW * C + (ch < C ? ch : 0);
    int pxo = px0;
    asm volatile("" : "+v"(pxo));                              // (the four source offsets are recomputed here: three registers fewer across the job)
#pragma unroll
    for (int j = 0; j < NT; ++j) pre[j] = *reinterpret_cast<const f32x4*>(img + (pxo + 128 * j < HW ? pxo + 128 * j : 0) * C);
  };
  auto stage_store = [&](int sl, const f32x4 (&pre)[NT]) {
    const bool live = 16 * sl + 4 * my_c4 < C;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 v = live ? pre[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x2*>(TIN + s_dst[j]) = (f32x2){v[0], v[1]};
      *reinterpret_cast<f32x2*>(TIN + s_dst[j] + 2 * TP_PLANE) = (f32x2){v[2], v[3]};
    }
  };

  // ---- depthwise role: the lane's patch in either round (host table: conflict-free 16-lane groups), ONE register per round
  // (py << 8 | px, or -1); window corner and exchange slots are recomputed from it where they are used (~30 VALU per round
  // against 400 of packed FMAs) - held in registers across the job they cost 10 of the 256 this kernel has
  int patch[2];
  {
    const int PWn = (W + 1) / 2;
    const __attribute__((address_space(4))) unsigned char* tbl =
        (const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TowerJobs, lane_patch);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pid = tbl[64 * r + lane];
      const int pidc = pid != 255 ? pid : 0;
      const int py = yfv2_fdiv(pidc, 1.0f / (float)PWn), pxx = pidc - py * PWn;
      patch[r] = pid != 255 ? (py << 8 | pxx) : -1;
    }
  }
  // ---- pointwise role: NT pixel tiles of 16
  int opix[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) opix[nt] = 16 * (wv * NT + nt) + p;

  typedef const __attribute__((address_space(4))) TowerArgs JobArgs;
  JobArgs* kj = (JobArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  // offset of the pair tap table in a job's image (a job is packed for the launch it was planned for: under BOTH the a halves as a launch
  // without output conv)
  auto taps_off = [&](int form) { return th_lds_img((BOTH && form == 0) ? 0 : MH) + TP_TAPS2; };
  int b = spread ? bid % a.B : bid, jc = spread ? bid / a.B : 0;   // the current item: image b, job jc
  Yfv2Watch watch;
  f32x4 pre[NT];
  stage_load(kj[jc].in, b < a.B ? b : 0, 0, pre);

  // ---- the filter fragments a wave carries into LDS: chunk s's pointwise phase reads only that chunk's fragments (NF tiles x 1 KB, + the
  // previous chunk's in odd chunks), so set s + 1 is written during chunk s's pointwise phase from a register loaded a chunk earlier (wave
  // m < NF carries tile m's fragment, one 16-byte piece per lane).  form 0: the 72 x 72 filter WP (5 tiles), form 1: the merged matrix WH
  const int fdump = th_lds_img(MH) + TP_XB_FL + TP_TIN_FL + 4 * (lane & 15);
  auto frag_off = [&](int form, int sc) { return (form == 0 ? 0 : TH_WH) + (((wv < nf_of(form) ? wv : 0) * KC + sc) * 64 + lane) * 4; };
  auto frag_dst = [&](int form, int sc) { return wv < nf_of(form) ? frag_off(form, sc) : fdump; };
  f32x4 fpre;
  // the 1.5 KB of constants: waves 6, 7 carry their 96 16-byte pieces (lanes 0..63, 0..31; the fragment carriers are waves 0 .. 5 at
  // most); TWO copies in LDS (behind the dump area), item i reads copy i & 1 while the next item's is written under its last chunk
  static_assert(MH <= 6, "waves 6 and 7 carry the constants");
  const bool cs_w = wv == 6 || (wv == 7 && lane < 32);
  const int cs_i = cs_w ? (wv == 6 ? lane : 64 + lane) : 0;
  constexpr int CS2 = th_lds_img(MH) + TP_XB_FL + TP_TIN_FL + TP_DUMP_FL;      // float offset of the two copies (2 x 384)
  f32x4 cpre;
  // ---- prologue: fragment set 0 and the constants -> LDS, exchange and planes zeroed (the halo stays zero for the whole job)
  {
    // this wave's tap records -> scalar cache: one request per 64-byte line, issued back to back NOW (nothing waits for them:
    // the results are dropped).  Left to the first use, each of a chunk's four loads is a miss to L2 / HBM - and in the job's
    // set-up, where scalar registers are short, the compiler serialises them: 2.8 k cycles in front of the first chunk
#define YFV2_L(p, o) "s_load_dword s40, %" #p ", " #o "\n\t"
#define YFV2_R(p, o) YFV2_L(p, o + 0) YFV2_L(p, o + 64) YFV2_L(p, o + 128) YFV2_L(p, o + 192)
#pragma unroll 1
    for (int j = 0; j < nj; ++j) {
      const float* tq = kj[j].img16 + taps_off(form_of(j)) + wv * 64;
      const float* tq4 = kj[j].img16 + taps_off(form_of(j)) + (4 * 8 + (wv & 3)) * 64;
      asm volatile(YFV2_R(0, 0) YFV2_R(0, 2048) YFV2_R(0, 4096) YFV2_R(0, 6144) YFV2_R(1, 0) :: "s"(tq), "s"(tq4) : "s40", "memory");
    }
#undef YFV2_R
#undef YFV2_L
    // Only set 0 and the 1.5 KB of constants are fetched here: 7.5 KB instead of 27 / 58 KB (the merged-matrix forms never read the
    // 72 x 72 filter at all) in the burst in which every CU of the chip fills at ~11 bytes per cycle.
    constexpr int NZ = (TP_XB_FL + TP_TIN_FL + TP_DUMP_FL) / 4;
    for (int i = tid; i < NZ; i += 512) reinterpret_cast<f32x4*>(XB)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int f0m = form_of(jc);
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(kj[jc].img16 + (cs_w ? TH_CS + 4 * cs_i : frag_off(f0m, 0)));
    fpre = *reinterpret_cast<const f32x4*>(kj[jc].img16 + frag_off(f0m, 1));
    // (every lane stores: what is not part of the image goes where nothing reads - a store behind a lane predicate leaves the
    // compiler a path on which the load is still pending, and it then guards the registers with vmcnt(0) behind the NEXT requests)
    *reinterpret_cast<f32x4*>(lds + (cs_w ? CS2 + 4 * cs_i : frag_dst(f0m, 0))) = f0;
  }
  __syncthreads();
  YFV2_WSTAMP(1);
  if (b < a.B) {
    stage_store(0, pre);
    stage_load(kj[jc].in, b, 1, pre);
  }

  // this wave's depthwise unit of the coming chunk: the pair's 25 taps + BN constants in 56 SGPRs
  yfv2_f16v t0, t1, t2;
  yfv2_f8 t3;
  auto load_taps = [&](const float* img16, int form, int s) __attribute__((always_inline)) {
    const int pair = s == KC - 1 ? (wv & 3) : wv;
    const yfv2_cf16* tp = (const yfv2_cf16*)(img16 + taps_off(form) + (s * 8 + pair) * 64);
    t0 = tp[0]; t1 = tp[1]; t2 = tp[2];
    t3 = *(const yfv2_cf8*)(tp + 3);
    __builtin_amdgcn_sched_barrier(0);                         // (requested HERE: invariant loads move freely otherwise)
  };
  load_taps(kj[jc].img16, form_of(jc), 0);
  int item = 0;
  // one item: five chunks and the form's epilogue
  auto run_item = [&]<int FORM>() __attribute__((always_inline)) {
    constexpr int NF = FORM == 0 ? KC : MH;
    // this item's job and the next item (its first slices, fragments, constants and taps are requested under this item's last chunks)
    JobArgs& ja = kj[jc];
    const int jn = spread ? jc : (jc + 1 < nj ? jc + 1 : 0), bn = spread ? a.B : (jn ? b : b + grid);
    const bool more = bn < a.B;
    JobArgs& jx = kj[more ? jn : jc];                             // (no next item: harmless re-requests of this one's)
    const int bx = more ? bn : b;
    const int fx = form_of(more ? jn : jc);                       // the next item's form
    const float* CS = lds + CS2 + (item & 1) * 384;
    const int hmh = FORM ? ja.mh : 0;
    const int mlive = (hmh + 15) >> 4;                           // output-channel tiles of the merged matrix (wave-uniform)
    f32x4 acc[NF][NT];                                         // form 0: output-channel tile x pixel tile (lane: 4 channels of 1 pixel);
                                                               // form 1: pixel tile x output-channel tile, transposed (lane: 4 pixels of 1 channel)
#pragma unroll
    for (int mt = 0; mt < NF; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    yfv2_u2 xprev[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xprev[nt] = (yfv2_u2){0u, 0u};

    // One chunk: [barrier] depthwise of slice s (both rounds of this wave's channel pair) [barrier] pointwise of slice s
    auto chunk = [&]<bool ODD, bool LAST>(int s, int stamp0) __attribute__((always_inline)) {
      const int pair = LAST ? (wv & 3) : wv;
      auto tap = [&](int t) -> f32x2 {
        const int i = 2 * t;
        return i < 16 ? (f32x2){t0[i & 15], t0[(i & 15) + 1]} : i < 32 ? (f32x2){t1[i & 15], t1[(i & 15) + 1]}
             : i < 48 ? (f32x2){t2[i & 15], t2[(i & 15) + 1]} : (f32x2){t3[i & 7], t3[(i & 7) + 1]};
      };
      __syncthreads();                                                      // slice s complete in TIN; exchange free
      YFV2_WSTAMP(stamp0);
      auto round = [&](int pq) __attribute__((always_inline)) {
        asm volatile("" : "+v"(pq));                                        // (opaque: what is derived from it below is not hoisted out of the job)
        const bool pvalid = pq >= 0;
        const int py = pvalid ? pq >> 8 : 0, pxx = pvalid ? pq & 255 : 0;
        const float* wp = TIN + pair * (TP_PLANE * 2) + ((2 * py) * P + 2 * pxx) * 2;
        // three window rows in flight (36 registers), row r + 3 requested behind row r's FMAs
        constexpr int WIN = 3;
        f32x4 w[6][3];
        auto read_row = [&](int r) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < 3; ++j) w[r][j] = *reinterpret_cast<const f32x4*>(wp + (r * P + 2 * j) * 2);
        };
#pragma unroll
        for (int r = 0; r < WIN; ++r) read_row(r);
        f32x2 d[2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k >> 1][k & 1] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int kx = 0; kx < 5; ++kx)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
              const int ky = r - dy;
              if (ky < 0 || ky >= 5) continue;
#pragma unroll
              for (int dx = 0; dx < 2; ++dx) {
                const int c = dx + kx;
                const f32x2 v = (c & 1) ? (f32x2){w[r][c >> 1][2], w[r][c >> 1][3]} : (f32x2){w[r][c >> 1][0], w[r][c >> 1][1]};
                d[dy][dx] = __builtin_elementwise_fma(v, tap(ky * 5 + kx), d[dy][dx]);
              }
            }
          // keeps the four accumulation chains interleaved, row by row in source order (left alone the scheduler runs the chains one
          // after the other: 25 dependent packed FMAs with a wait state between each two) and the window reads behind it (left alone
          // all 18 are hoisted to the top: 72 registers).  Per output the order is tap row, then tap column - towerh_kernel's, bit for bit
          asm volatile("" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) :: "memory");
          if (r + WIN < 6) read_row(r + WIN);
        }
        const f32x2 sc = tap(25), sh = tap(26);
        // BN + ReLU, ONE split into two fp16 terms; a patch row = two adjacent pixels = one 16-byte record {h1, h2, h1, h2} of the
        // pair's exchange run (H and W are even here: both rows of a patch exist, the record is aligned).  A patch that does not
        // exist stores where nothing reads (a store behind a lane predicate makes the compiler sink the pixel's 25 FMAs into the
        // branch: one serial chain)
        const int xdump = TP_XB_FL + TP_TIN_FL + 4 * (lane & 15);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          u32x4 rec;
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            f32x2 u = __builtin_elementwise_fma(d[dy][dx], sc, sh);
            u[0] = u[0] > 0.f ? u[0] : 0.f; u[1] = u[1] > 0.f ? u[1] : 0.f;           // (the BN constants carry the 2^4)
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            const h2_t h1 = __builtin_convertvector(u, h2_t);
            const f32x2 rr = u - __builtin_convertvector(h1, f32x2);                   // exact
            const h2_t h2 = __builtin_convertvector(rr, h2_t);
            rec[2 * dx] = __builtin_bit_cast(unsigned, h1);
            rec[2 * dx + 1] = __builtin_bit_cast(unsigned, h2);
          }
          const int q = (2 * py + dy) * W + 2 * pxx;
          *reinterpret_cast<u32x4*>(XB + (pvalid ? pair * TP_XS + 2 * q : xdump)) = rec;
        }
      };
      if constexpr (!LAST) {
        round(patch[0]);
        __builtin_amdgcn_sched_barrier(0);                                  // (round 1's reads stay behind round 0's FMAs)
        round(patch[1]);
      } else {
        round(wv < 4 ? patch[0] : patch[1]);
      }
      __syncthreads();                                                      // exchange complete; TIN free
      YFV2_WSTAMP(stamp0 + 1);

      // ---- pointwise: everything the chunk needs from LDS requested up front, then the MFMAs back to back
      u32x4 xb[NT];                                                         // {first terms of channels 4g .. 4g+3, second terms}: pairs 2g and 2g + 1
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(XB + (2 * g) * TP_XS + 2 * opix[nt]);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(XB + (2 * g + 1) * TP_XS + 2 * opix[nt]);
        xb[nt] = (u32x4){lo[0], hi[0], lo[1], hi[1]};
      }
      const float* FB = FORM == 0 ? WP_ : WH;
      // filter fragments: tile m + 1's requested in front of tile m's MFMAs (two tiles = 16 registers in flight, not all 2 x NF: 201 / 227
      // registers instead of 252 / 255).  All tiles, also where a job's output conv is narrower - the image holds zero tiles there
      auto frag = [&](int m, int sc) { return *reinterpret_cast<const u32x4*>(FB + ((m * KC + sc) * 64 + lane) * 4); };
      u32x4 wfn = frag(0, s), w0n = {0u, 0u, 0u, 0u};
      if constexpr (ODD) w0n = frag(0, s - 1);
      // the next chunk's filter fragments -> LDS (requested a chunk ago), the set after that into the register
      *reinterpret_cast<f32x4*>(lds + (s + 1 < KC ? frag_dst(FORM, s + 1) : frag_dst(fx, 0))) = fpre;
      fpre = *reinterpret_cast<const f32x4*>(s + 2 < KC ? ja.img16 + frag_off(FORM, s + 2) : jx.img16 + frag_off(fx, s + 2 - KC));
      if constexpr (LAST) {   // the next item's constants -> the other copy
        *reinterpret_cast<f32x4*>(lds + (cs_w ? CS2 + ((item + 1) & 1) * 384 + 4 * cs_i : fdump)) = cpre;
      } else if (s == KC - 2) {
        cpre = *reinterpret_cast<const f32x4*>(jx.img16 + TH_CS + 4 * cs_i);
      }
      // the next slice -> TIN, the one after into registers (the next ITEM's behind this one's last)
      if (s + 1 < KC || more) stage_store(s + 1 < KC ? s + 1 : 0, pre);
      if (s + 2 < KC) stage_load(ja.in, b, s + 2, pre);
      else stage_load(jx.in, bx, s + 2 - KC, pre);
#pragma unroll
      for (int m = 0; m < NF; ++m) {
        const u32x4 wf = wfn, w0 = w0n;
        __builtin_amdgcn_sched_barrier(0);                                  // (tile by tile: in straight-line code every fragment read is hoisted to the top otherwise)
        if (m + 1 < NF) {
          wfn = frag(m + 1, s);
          if constexpr (ODD) w0n = frag(m + 1, s - 1);
        }
        if constexpr (FORM == 0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_cross(wf, xb[nt], acc[m][nt]);
          if constexpr (ODD) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_main2(w0, wf, xprev[nt], xb[nt], acc[m][nt]);
          } else if constexpr (LAST) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_main1(wf, xb[nt], acc[m][nt]);
          }
        } else if (m < mlive) {                                           // (wave-uniform: a narrower output conv leaves zero tiles)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_cross(xb[nt], wf, acc[m][nt]);
          if constexpr (ODD) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = mfma_main2((u32x4){xprev[nt][0], xprev[nt][1], 0u, 0u}, xb[nt], (yfv2_u2){w0[0], w0[1]}, wf, acc[m][nt]);
          } else if constexpr (LAST) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma_main1(xb[nt], wf, acc[m][nt]);
          }
        }
      }
      if constexpr (!ODD) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xprev[nt] = (yfv2_u2){xb[nt][0], xb[nt][1]};
      }
      // the NEXT chunk's taps (of the next image's first chunk behind the last): requested here, behind this phase's last LDS
      // wait - scalar loads and LDS reads share one counter, a scalar load in flight turns every LDS wait into "wait for everything" -
      // and landed (a scalar-cache miss is an L2 round trip) by the time the matrix pipe has drained and the barrier opens
      load_taps(LAST ? jx.img16 : ja.img16, LAST ? fx : FORM, LAST ? 0 : s + 1);
      YFV2_WSTAMP(stamp0 + 2);
    };

#pragma unroll 1
    for (int sp = 0; sp < 2; ++sp) {
      chunk.template operator()<false, false>(2 * sp, 2 + 6 * sp);
      chunk.template operator()<true, false>(2 * sp + 1, 5 + 6 * sp);
    }
    chunk.template operator()<false, true>(KC - 1, 14);
    YFV2_WSTAMP(17);

    // (the store addresses are formed HERE, from opaque copies: hoisted out of the job they are spilled, and a scratch reload between the
    // stores waits for every store before it)
    int pe = p, ge = g;
    asm volatile("" : "+v"(pe), "+v"(ge));
    if constexpr (FORM == 0) {
      // pointwise BN (no ReLU: fpn.py:16-17,23-24); the scale carries 2^-(sw+4)
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 0 * 96 + 16 * mt + 4 * g);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 1 * 96 + 16 * mt + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_elementwise_fma(acc[mt][nt], sc, sh);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) watch.see(acc[0][nt][0]);   // a depthwise result beyond fp16's range: NaN in every channel of its pixel
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int q = 16 * (wv * NT + nt) + pe;
        if (q >= HW) continue;
        float* dst = ja.out + ((size_t)b * HW + q) * C;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt)
          if (16 * mt + 4 * ge < C) *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * ge) = acc[mt][nt];
      }
    } else {
      const float us = CS[3 * 96];                                          // 2^-(swh+4)
      const bool vec = (HW & 3) == 0;                                       // (alignment of every channel plane)
      // (both plane pointers in scalar registers BEFORE the lane-dependent choice: left to itself the compiler turns the choice
      // between two kernel-argument fields into ONE vector load from a chosen address - a full load latency in front of every
      // output tile's stores)
      float* hn0 = ja.nchw0; float* hn1 = ja.nchw1;
      asm volatile("" : "+s"(hn0), "+s"(hn1));
      const int hsplit = ja.split;
#pragma unroll
      for (int m = 0; m < (FORM ? MH : 1); ++m) {
        if (m >= mlive) break;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) watch.see((acc[m][nt][0] + acc[m][nt][1]) + (acc[m][nt][2] + acc[m][nt][3]));   // transposed: a lane's four values are four PIXELS
        const int co = 16 * m + pe;
        if (co < hmh) {
          const float bias = CS[2 * 96 + co];
          float* plane = co < hsplit ? hn0 + ((size_t)b * hsplit + co) * HW : hn1 + ((size_t)b * (hmh - hsplit) + (co - hsplit)) * HW;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int px0 = 16 * (wv * NT + nt) + 4 * ge;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(acc[m][nt][r], us, bias);
            if (vec) {
              if (px0 < HW) *reinterpret_cast<f32x4*>(plane + px0) = y;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (px0 + r < HW) plane[px0 + r] = y[r];
            }
          }
        }
      }
    }
    YFV2_WSTAMP(18);
    jc = jn; b = bn;
  };
  for (; b < a.B; ++item) {
    if constexpr (BOTH) {
      if (form_of(jc) == 0) run_item.template operator()<0>();
      else run_item.template operator()<1>();
    } else {
      run_item.template operator()<FORM1>();
    }
  }
  watch.report(kj[0].nonfinite);
}

template <int MH>
static void launch_towers(const TowerJobs& jobs, hipStream_t s) {
  const int B = jobs.j[0].B;
  const size_t lds = sizeof(float) * ((size_t)th_lds_img(MH) + (TH_C / 4) * 27 * 4 + 15 * 11 * TH_C + 128 * TS_CP + 256);   // (+ 1 KB where unwanted stores go)
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&towers_kernel<MH>), lds_ok);
  YFV2_LAUNCH((towers_kernel<MH>), dim3(B < 256 ? B : 256), dim3(512), lds, s, jobs);
}

template <int MH, int PS, int NT>
static void launch_towerh(TowerJobs jobs, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)th_lds_img(MH) + 4 * (4 * 16 * NT * 8 + ThGeom<PS>::TIN_SLOTS));
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&towerh_kernel<MH, PS, NT>), lds_ok);
  const int B = jobs.j[0].B;
  jobs.gpj = B < 256 ? B : 256;
  YFV2_LAUNCH((towerh_kernel<MH, PS, NT>), dim3(jobs.gpj * (jobs.par ? jobs.n : 1)), dim3(512), lds, s, jobs);
}

// towerp_kernel's lane -> patch table.  ds_read_b128 is serviced in four groups of 16 lanes, one LDS cycle per group when the 16
// slots are distinct mod 16 (64 banks x 4 bytes).  A patch (py, px) reads slots py * TP_P + px + const: patches are dealt to the
// eight groups of the two rounds so that no group holds two patches of one residue class (a class has at most ceil(patches / 16)
// + 1 members; with more than eight the extra one costs its group one cycle).
static void towerp_lane_patches(int H, int W, unsigned char (&tbl)[128]) {
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  const int PWn = (W + 1) / 2, PHn = (H + 1) / 2, NP = PWn * PHn;
  // step 1 (reads): patches -> the eight 16-lane groups of the two rounds, one patch per residue class and group
  int member[8][16], fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool has[8][16] = {};
  std::vector<int> late;
  for (int pid = 0; pid < NP; ++pid) {
    const int res = ((pid / PWn) * TP_P + pid % PWn) & 15;
    int gi = 0;
    while (gi < 8 && (has[gi][res] || fill[gi] >= 16)) ++gi;
    if (gi == 8) { late.push_back(pid); continue; }
    has[gi][res] = true;
    member[gi][fill[gi]++] = pid;
  }
  for (int pid : late)
    for (int gi = 0; gi < 8; ++gi)
      if (fill[gi] < 16) { member[gi][fill[gi]++] = pid; break; }
  // step 2 (the exchange stores, ds_write_b128: eight groups of eight CONTIGUOUS lanes, 32 banks): inside a round, which lane of
  // its group a patch sits on is free - swap until the eight lanes 8k .. 8k+7 store to distinct 16-byte slots mod 8 (slot of a
  // patch row = ((2 py + dy) W + 2 px) / 2: the same residue pattern for dy = 0 and 1), as far as a fixed-seed local search gets
  for (int i = 0; i < 128; ++i) tbl[i] = 255;
  unsigned rs = 12345u;
  auto rnd = [&](unsigned n) { rs = rs * 1664525u + 1013904223u; return (rs >> 8) % n; };
  for (int r = 0; r < 2; ++r) {
    int at[64];
    for (int l = 0; l < 64; ++l) at[l] = -1;
    for (int gi = 0; gi < 4; ++gi)
      for (int i = 0; i < fill[4 * r + gi]; ++i) at[grp[gi][i]] = member[4 * r + gi][i];
    auto wres = [&](int pid) { return ((2 * (pid / PWn)) * (W / 2) + pid % PWn) & 7; };
    auto cost = [&]() {
      int c = 0;
      for (int k = 0; k < 8; ++k) {
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx = 1;
        for (int l = 8 * k; l < 8 * k + 8; ++l)
          if (at[l] >= 0) mx = std::max(mx, ++cnt[wres(at[l])]);
        c += mx - 1;
      }
      return c;
    };
    int cur = cost();
    for (int it = 0; it < 40000 && cur > 0; ++it) {
      const int gi = (int)rnd(4), l1 = grp[gi][rnd(16)], l2 = grp[gi][rnd(16)];
      if (l1 == l2) continue;
      std::swap(at[l1], at[l2]);
      const int c = cost();
      if (c <= cur) cur = c;
      else std::swap(at[l1], at[l2]);
    }
    for (int l = 0; l < 64; ++l)
      if (at[l] >= 0) tbl[64 * r + l] = (unsigned char)at[l];
  }
}

template <int MH, bool BOTH>
static void launch_towerp(TowerJobs jobs, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)tp_lds_floats(MH);
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&towerp_kernel<MH, BOTH>), lds_ok);
  const int B = jobs.j[0].B;
  // a workgroup runs every side-by-side job of its images (one prologue, the next item's data requested under the current one) - unless
  // there are fewer items than CUs: then one workgroup per item
  const int njobs = jobs.par ? jobs.n : 1;
  jobs.gpj = (!BOTH && njobs > 1 && njobs * B <= 256) ? njobs * B : (B < 256 ? B : 256);
  {
    static std::mutex mu;
    static std::map<int, std::array<unsigned char, 128>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const int key = jobs.j[0].H * 64 + jobs.j[0].W;
    auto it = cache.find(key);
    if (it == cache.end()) {
      unsigned char t[128];
      towerp_lane_patches(jobs.j[0].H, jobs.j[0].W, t);
      std::array<unsigned char, 128> arr;
      std::copy(t, t + 128, arr.begin());
      it = cache.emplace(key, arr).first;
    }
    std::copy(it->second.begin(), it->second.end(), jobs.lane_patch);
  }
  YFV2_LAUNCH((towerp_kernel<MH, BOTH>), dim3(jobs.gpj), dim3(512), lds, s, jobs);
}

// 2x2 patches: up to 22x22 with at most 128 patches; single pixels: up to 11x11
bool yfv2_towerh_supported(int H, int W) {
  if (H < 1 || W < 1) return false;
  if (H <= 11 && W <= 11) return true;
  return H <= 22 && W <= 22 && ((H + 1) / 2) * ((W + 1) / 2) <= 128;
}
// several tower halves in one launch: maps up to 11x11 only (towers_kernel)
bool yfv2_towerh_multi(int H, int W) { return H >= 1 && W >= 1 && H <= 11 && W <= 11; }

// The kernel is instantiated for 0, 1 or 6 output-conv tiles; every job's image must be packed for `mh_tiles` of them
// (WeightPacker::image_towerh pads with zero tiles) - the LDS layout depends on it.
bool yfv2_launch_towerh(const TowerJobs& jobs, int mh_tiles, hipStream_t s) {
  if (jobs.n < 1 || jobs.n > 4) return false;
  const TowerArgs& a = jobs.j[0];
  for (int i = 0; i < jobs.n; ++i)
    if (!jobs.j[i].img16 || jobs.j[i].H != a.H || jobs.j[i].W != a.W || jobs.j[i].B != a.B) return false;
  if (!yfv2_towerh_supported(a.H, a.W) || (mh_tiles != 0 && mh_tiles != 1 && mh_tiles != 6)) return false;
  if (jobs.n > 1 && !jobs.par) {
    if (!yfv2_towerh_multi(a.H, a.W) || mh_tiles == 0) return false;
    if (mh_tiles == 1) launch_towers<1>(jobs, s);
    else launch_towers<6>(jobs, s);
  } else if (a.H <= 11 && a.W <= 11) {
    // single halves of a small map: only what the planner leaves unmerged - a class head wider than 96 channels (its b half without merged
    // matrix: 0 tiles, the output convs as pointwise launches) beside the reg tower (1 tile).  Everything else at this size is towers_kernel's
    if (mh_tiles == 0) launch_towerh<0, 1, 1>(jobs, s);
    else if (mh_tiles == 1) launch_towerh<1, 1, 1>(jobs, s);
    else return false;
  } else if ((a.H & 1) || (a.W & 1)) {   // odd maps (13x13 at 416x416): towerh_kernel<.., 2, 4> (towerp_kernel's 16-byte patch-row records want even sizes)
    if (mh_tiles == 0) launch_towerh<0, 2, 4>(jobs, s);
    else if (mh_tiles == 1) launch_towerh<1, 2, 4>(jobs, s);
    else launch_towerh<6, 2, 4>(jobs, s);
  } else if (jobs.par && jobs.n == 4 && mh_tiles != 0 && !jobs.j[0].has_head && !jobs.j[1].has_head && jobs.j[2].has_head && jobs.j[3].has_head) {
    // a whole level: {cls a, reg a, cls b, reg b}.  Batches that fill the chip (two workgroups' worth of items per CU and more): ONE launch, a
    // workgroup runs its image's four halves back to back; smaller batches: the a halves and the b halves as two launches of independent
    // items (one workgroup per item - a single image's four halves in sequence would take twice as long as two rounds of two)
    if (2 * a.B > 256) {
      if (mh_tiles == 1) launch_towerp<1, true>(jobs, s);
      else launch_towerp<6, true>(jobs, s);
    } else {
      TowerJobs ja = jobs, jb = jobs;
      ja.n = 2; jb.n = 2; jb.j[0] = jobs.j[2]; jb.j[1] = jobs.j[3];
      launch_towerp<0, false>(ja, s);
      if (mh_tiles == 1) launch_towerp<1, false>(jb, s);
      else launch_towerp<6, false>(jb, s);
    }
  } else {
    if (mh_tiles == 0) launch_towerp<0, false>(jobs, s);
    else if (mh_tiles == 1) launch_towerp<1, false>(jobs, s);
    else launch_towerp<6, false>(jobs, s);
  }
  return true;
}
