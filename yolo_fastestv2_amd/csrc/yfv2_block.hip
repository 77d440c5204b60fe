// yfv2_block.hip - the fused ShuffleV2 blocks of stages 3 and 4 and the tower halves for gfx950:
//   block_s1chain6_kernel   stage 3's seven stride-1 blocks as ONE launch (bf16x6 pointwise convs)
//   block_s1pool_kernel     stage 4's three stride-1 blocks as one launch, the activation resident in LDS
//   block_s2_kernel         stride-2 block with the input tile staged in LDS (stage3.0; stage2.0 / stage 3 of odd sizes)
//   block_s2w_kernel        stage4.0 (96 -> 192) in one launch over bands of two output rows
//   tower2_kernel           a DWConvblock half (+ the chained output conv)
// Shapes none of them covers run layer by layer (yfv2_conv.hip: pw_kernel / dw_kernel), which is also the YFV2_FUSED=0 plan.
//
// Reference stride-1 block (model/backbone/shufflenetv2.py:19-32,48-51,57-63), c = 2*C2 channels:
//   pass = x[:, 0::2]                      (even channels, untouched)
//   y    = x[:, 1::2]                      (odd channels)
//   y    = ReLU(BN(pw1(y)))   C2 -> C2
//   y    = BN(dw3x3(y))       pad 1, stride 1
//   y    = ReLU(BN(pw2(y)))   C2 -> C2
//   out  = cat(pass, y)
// (Rounds 1-2 also carried single-block, two-block and fp32-MFMA chain kernels; they were superseded by the two chains and
// removed in round 3 - git history has them.)
#include <cstdlib>
#include <type_traits>

#include "yfv2_internal.h"

typedef _Float16 yfv2_h4c __attribute__((ext_vector_type(4)));
typedef _Float16 yfv2_h8c __attribute__((ext_vector_type(8)));
typedef unsigned yfv2_u2c __attribute__((ext_vector_type(2)));

template <int C2>
struct S1Cfg {
  static constexpr int KC = (C2 + 15) / 16;  // 16-channel chunks (also M tiles: M == K == C2)
  static constexpr int DW_FL = 9 * KC * 16;  // depthwise taps [9][KC*16]
  static constexpr int CST_FL = 6 * KC * 16; // sc1, sh1, scd, shd, sc2, sh2
};

#define YFV2_STAMP(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = (long long)__builtin_readcyclecounter(); } while (0)
// per-wave stamps of workgroup 0 (YFV2_TRACE=1, tools/trace_waves.py): trace[64 + 32 * wave + i]
#define YFV2_WSTAMP(i) do { if (a.trace && blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.trace[64 + 32 * (threadIdx.x >> 6) + (i)] = (long long)__builtin_readcyclecounter(); } while (0)

// ============================================================================
// A CHAIN of stride-1 blocks in one launch (C2 = 48, whole image, plane-per-quad tile): stage 3's blocks 1..7
// ============================================================================
// Reference: N consecutive ShuffleV2Block(stride 1) (shufflenetv2.py:19-32,48-51,57-63).  Per image the chain reads its
// input X once and writes its output Z once; everything in between stays on the chip or in a few parked dwords per pixel:
//   * the 48 channels that go through a block's branch live in the LDS tile: 12 planes [quad][haloed row][W + 1][4 floats]
//     (a row's left halo column is the previous row's right halo - one shared zero slot - so the image is one linear run of
//     16-byte slots per plane and both phases tile that run, 16 consecutive slots per wave tile; planes a multiple of 256
//     bytes apart: no ds_read_b128 of a 16-lane group collides);
//   * a value that passes k blocks before it becomes a branch input is, by the time it is needed,
//       k = 0  a fresh branch output picked straight out of the accumulators (elements 1, 3 of every quad),
//       k = 1  three registers per pixel slot (elements 0: held for one block; X[4i+2] for the second block),
//       k >= 2 "parked": elements 2 (twelve values per pixel and block) and X[4i] are stored as single dwords into Z -
//              which is free until the very end - GROUPED BY THE BLOCK THAT CONSUMES THEM: Z[12 (k - 2) .. + 11] holds
//              the twelve parked inputs of block k, so that each lane fetches its three with one 12-byte load during
//              the phase A of the block before; values no block of the chain consumes are parked at their final place;
//   * which logical channel sits in which lane / element / tile position / Z position is decided on the host
//     (PlanBuilder::s1chain_block): it permutes the input columns of every pw1 and the output rows of every pw2 so that
//     the kernel's fixed, lane-uniform data movement below is the reference's channel_shuffle; the park positions ride
//     at the end of each block's LDS image, and the consumers of the stage's output read Z through the channel
//     permutation the plan reports.
// Per block image: W1 | W2 (pre-split, below) | dw taps | 6 BN vectors | int tables PS[3][4], XS[6][4] (+ pad).
constexpr int CH_TBL_FL = 64;

// ============================================================================
// The chain on bf16x6 (default): block_s1chain6_kernel
// ============================================================================
// The dataflow above with the two pointwise convs of every block on the bf16 matrix cores.  (The fp32-MFMA chain of round 2
// was bound by its matrix-core time - 2232 MFMAs of 32 cycles per block and image: 25 k of a block's 38 k ticks - and, in
// phase B, by LDS reads: taps re-read per tile.)  Here:
//  * W1 / W2 arrive PRE-SPLIT (WeightPacker::append_s1_bf6): K = 48 = one chunk PAIR + one single chunk.  Per output tile
//    six 16-byte operands: hi / mid / lo quads of the pair (32 k-slots = chunks 0, 1: six MFMAs, no duplication) and the
//    {hi,hi} {mid,mid} {hi,lo} quads of chunk 2 (three MFMAs against {hi,mid} {lo,hi} of the activations) - nine
//    v_mfma_f32_16x16x32_bf16 of 16 cycles per (output tile, pixel tile) instead of twelve fp32 MFMAs of 32, and no VALU
//    work on the filter side;
//  * the filter operands are read from LDS per PAIR of pixel tiles (not held in registers: 72 VGPRs), the depthwise taps of
//    a chunk once per pair of tiles;
//  * an image is 40 KB (10000 floats) instead of 21.6: ONE image buffer.  The next block's image travels through five
//    registers per thread during phases A and B and is stored in the exchange phase, after the barrier that retires phase
//    B's reads; the park table of the current block is read into registers before that barrier.
// Round 3: the same dataflow on fp16x3 (make_b / mfma9 below): five MFMAs instead of nine, a two-term split, half the filter
// image (the BN scales of the two pointwise convs carry the exact 2^-(sw+4)).
// Per block image: W1 [3 mt][{w1,w1} pair | {w2,w2} pair | {w1,w2} chunk 2][64][4] x 2^sw1 | W2 likewise | dw taps [9][48] |
// 6 BN vectors | int tables (as above).
constexpr int CH6_WP_FL = 3 * 3 * 256;
constexpr int CH6_IMG_FL = 2 * CH6_WP_FL + S1Cfg<48>::DW_FL + S1Cfg<48>::CST_FL + CH_TBL_FL;

// row pitch of the patch-parity tile in 16-byte slots: both column parities of a haloed row (2 (PW + 1) slots), = PW mod 16
__host__ __device__ constexpr int s1chain_pitch(int PW) { return 2 * (PW + 1) + ((PW - 2 * (PW + 1)) % 16 + 16) % 16; }
// slots per plane, a multiple of 16 (round 6): ds_read_b128 / ds_write_b128 are serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32
// (tools/ubench/ldsgroups.hip) - a group mixes lanes of two channel groups g, g + 1, i.e. of two PLANES.  With consecutive patches on
// consecutive slots a group is conflict-free only if the plane stride is 0 mod 16 slots (22x22: 2 x 12 x 27 = 648 = 8 mod 16 put lanes
// 20-27 on exactly the banks of lanes 0-3, 12-15: every such access took two passes)
__host__ __device__ constexpr int s1chain_plane(int PH, int PW) { return (2 * (PH + 1) * s1chain_pitch(PW) + 15) & ~15; }

template <int THREADS>
__global__ __launch_bounds__(THREADS, 2) void block_s1chain6_kernel(BlockS1Args a) {
  constexpr int C2 = 48;
  using Cfg = S1Cfg<C2>;
  constexpr int KC = Cfg::KC, NQ = C2 / 4, C = 2 * C2, NT = 4;
  constexpr int NW = THREADS / 64;
  constexpr int N4 = CH6_IMG_FL / 4, NIT = (N4 + THREADS - 1) / THREADS;   // float4 per image, per thread (5)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* IM = lds;                                        // the current block's image
  float* T1 = lds + CH6_IMG_FL;
  const float* W1P = IM;
  const float* W2P = IM + CH6_WP_FL;
  const float* WD = IM + 2 * CH6_WP_FL;
  const float* CS = WD + Cfg::DW_FL;
  const int H = a.H, W = a.W, HW = H * W, NB = a.nblk;
  // Tile geometry (round 4): a lane owns a 2x2 PATCH of pixels - its four pixel tiles nt = 2 dy + dx are the patch's four pixels -
  // so that phase B's depthwise reads the patch's 4x4 window once (16 quads for four outputs instead of 36) and a chunk's taps
  // once for all four.  A plane is kept as [row parity][haloed row / 2][column parity][haloed column / 2] (the towers'
  // layout, yfv2_towerh.hip): every read or write instruction addresses ONE parity pair, where consecutive lanes = consecutive
  // patches sit in consecutive 16-byte slots; the row pitch is = PW mod 16, so that the 16 lanes ds_read_b128 is serviced
  // in stay on 16 different slot residues when they wrap into the next patch row.
  const int PH = H >> 1, PW = W >> 1;
  const int CP1 = PW + 1;                                   // slots of one column parity (haloed columns 0 .. 2 PW + 1)
  const int RPT = s1chain_pitch(PW);
  const int PL = s1chain_plane(PH, PW);                     // slots per plane
  const float invPW = 1.0f / (float)PW;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, wave = tid >> 6;
  YFV2_WSTAMP(0);

  for (int i = tid; i < NQ * PL; i += THREADS) reinterpret_cast<f32x4*>(T1)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};  // halo rows / columns stay zero

  const int patch = 16 * wave + p;
  const bool pvalid = patch < PH * PW;
  const int pi = pvalid ? yfv2_fdiv(patch, invPW) : 0, pj = pvalid ? patch - pi * PW : 0;
  // haloed row hr -> (hr & 1) * (PH + 1) + (hr >> 1), haloed column hc -> (hc & 1) * CP1 + (hc >> 1); pixel (r, c) = haloed (r + 1, c + 1)
  const int wbase = pi * RPT + pj;                          // window position (0, 0) = haloed (2 pi, 2 pj): both parities 0
  // Parked values (see above) that a later block consumes - positions 0 .. NPARK of Z - do not go to Z any more (round 4) but to a
  // scratch laid out for the lanes: [position][patch][the patch's four pixels], ONE 16-byte store / load per lane and position
  // (16 lanes = 256 contiguous bytes) where the NHWC positions of Z took four dword accesses to four different lines.  Values
  // no block consumes (positions >= NPARK) are still stored at their final place in Z.
  const int NPARK = 12 * (NB - 2);
  const int PP4 = ((PH * PW + 15) & ~15) * 4;
  int sl[NT], pix[NT];
  bool valid[NT], real[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int dy = nt >> 1, dx = nt & 1;                    // haloed row 2 pi + dy + 1, haloed column 2 pj + dx + 1
    sl[nt] = wbase + ((dy ^ 1) * (PH + 1) + dy) * RPT + (dx ^ 1) * CP1 + dx;
    valid[nt] = pvalid; real[nt] = pvalid;
    pix[nt] = pvalid ? (2 * pi + dy) * W + 2 * pj + dx : 0;
  }
  float* Tg = T1 + (size_t)g * PL * 4;                    // plane of quad g; quad 4 s + g is 4 s planes further
  // (a slot index goes through an opaque register wherever an LDS address is formed from it: addresses hoisted out of the
  // block loop are what the register allocator spills first, and every scratch reload waits for the image loads in flight)
  auto opq = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
  // The workgroup communicates through LDS only (a pixel's parked values are stored and loaded back by lanes of ONE wave, in
  // program order): its barriers wait for the LDS counter, not for the global stores in flight - __syncthreads() would
  // also drain vmcnt, i.e. stall every block's exchange phase until its park stores are acknowledged by L2.
  auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  Yfv2Watch watch;   // range guard of the fp16x3 products (yfv2_internal.h): one accumulator element per pixel tile and pointwise conv

  // fp16x3 (round 3; yfv2_stem16.hip): every operand = two fp16 terms, w1 x2 + w2 x1 + w1 x1 with fp32 accumulation.
  // B operands of one pixel tile from its three chunk fragments (x 2^4 first): the pair (chunks 0, 1) as {x1, x1} and
  // {x2, x2}, chunk 2 as {x2, x1} - against the filter's {w1, w1} {w2, w2} of the pair and {w1, w2} of chunk 2 that is five
  // v_mfma_f32_16x16x32_f16 per (output tile, pixel tile): the pair's two cross products, chunk 2's cross products in ONE
  // instruction, the pair's main product, chunk 2's main product ({0, w1} against the same {x2, x1}).  Round 2's bf16x6 form
  // took nine, a three-term split per activation quad and twice the filter image.
  struct BOps { yfv2_h8c p1, p2, s; };
  auto make_b = [&](f32x4 c0, f32x4 c1, f32x4 c2v) __attribute__((always_inline)) {
    unsigned h[3][2], l[3][2];
    const f32x4 cs[3] = {c0 * 16.0f, c1 * 16.0f, c2v * 16.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const yfv2_h4c t1 = __builtin_convertvector(cs[k], yfv2_h4c);                                   // v_cvt_pk_f16_f32 (RN)
      const yfv2_h4c t2 = __builtin_convertvector(cs[k] - __builtin_convertvector(t1, f32x4), yfv2_h4c);   // the difference is exact
      const yfv2_u2c a1 = __builtin_bit_cast(yfv2_u2c, t1), a2 = __builtin_bit_cast(yfv2_u2c, t2);
      h[k][0] = a1[0]; h[k][1] = a1[1]; l[k][0] = a2[0]; l[k][1] = a2[1];
    }
    BOps b;
    b.p1 = __builtin_bit_cast(yfv2_h8c, (u32x4){h[0][0], h[0][1], h[1][0], h[1][1]});
    b.p2 = __builtin_bit_cast(yfv2_h8c, (u32x4){l[0][0], l[0][1], l[1][0], l[1][1]});
    b.s = __builtin_bit_cast(yfv2_h8c, (u32x4){l[2][0], l[2][1], h[2][0], h[2][1]});
    return b;
  };
  // acc[n] += W[mt] x B[n] for the lane's four pixel tiles: five products, the small ones first, the tiles interleaved (the
  // three filter operands of ONE output tile in registers at a time, read once for all four tiles)
  auto mfma9 = [&](const float* WP, int mt, const BOps (&b)[NT], f32x4 (&acc)[NT]) __attribute__((always_inline)) {
    const float* wq = WP + ((mt * 3) * 64 + lane) * 4;
    const u32x4 q1 = *reinterpret_cast<const u32x4*>(wq), q2 = *reinterpret_cast<const u32x4*>(wq + 256), qs = *reinterpret_cast<const u32x4*>(wq + 512);
    const yfv2_h8c w1p = __builtin_bit_cast(yfv2_h8c, q1), w2p = __builtin_bit_cast(yfv2_h8c, q2), ws = __builtin_bit_cast(yfv2_h8c, qs);
    const yfv2_h8c wm = __builtin_bit_cast(yfv2_h8c, (u32x4){0u, 0u, qs[0], qs[1]});
#define CH6_EACH(EXPR) _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[n] = EXPR;
    CH6_EACH(__builtin_amdgcn_mfma_f32_16x16x32_f16(w1p, b[n].p2, acc[n], 0, 0, 0))
    CH6_EACH(__builtin_amdgcn_mfma_f32_16x16x32_f16(w2p, b[n].p1, acc[n], 0, 0, 0))
    CH6_EACH(__builtin_amdgcn_mfma_f32_16x16x32_f16(ws, b[n].s, acc[n], 0, 0, 0))
    CH6_EACH(__builtin_amdgcn_mfma_f32_16x16x32_f16(w1p, b[n].p1, acc[n], 0, 0, 0))
    CH6_EACH(__builtin_amdgcn_mfma_f32_16x16x32_f16(wm, b[n].s, acc[n], 0, 0, 0))
#undef CH6_EACH
  };

  auto phase_a = [&]() __attribute__((always_inline)) {
    BOps b[NT];
    int slo[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) slo[n] = opq(sl[n]);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x4 bf[KC];
#pragma unroll
      for (int s2 = 0; s2 < KC; ++s2) bf[s2] = *reinterpret_cast<const f32x4*>(Tg + ((size_t)(4 * s2) * PL + slo[n]) * 4);
      b[n] = make_b(bf[0], bf[1], bf[2]);
    }
#pragma unroll
    for (int mt = 0; mt < KC; ++mt) {
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mfma9(W1P, mt, b, acc);
      if (mt == 0) { watch.see(acc[0][0]); watch.see(acc[1][0]); watch.see(acc[2][0]); watch.see(acc[3][0]); }
      const f32x4 sc1 = *reinterpret_cast<const f32x4*>(CS + 0 * KC * 16 + 16 * mt + 4 * g);
      const f32x4 sh1 = *reinterpret_cast<const f32x4*>(CS + 1 * KC * 16 + 16 * mt + 4 * g);
      if (pvalid) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4 y;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float uu = __builtin_fmaf(acc[n][c], sc1[c], sh1[c]);
            y[c] = uu > 0.f ? uu : 0.f;
          }
          // (in place: the lane's twelve input quads were read above, before its first output quad is stored - and only this
          // lane reads or writes these slots in phase A)
          *reinterpret_cast<f32x4*>(T1 + ((size_t)(4 * mt + g) * PL + slo[n]) * 4) = y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);                  // one output tile's operands at a time
    }
  };
  // window row wr (haloed row 2 pi + wr) / column wc of the patch, as slot offsets from wbase: wave-uniform
  const int wro[4] = {0, (PH + 1) * RPT, RPT, (PH + 2) * RPT};
  const int wco[4] = {0, CP1, 1, CP1 + 1};
  auto phase_b = [&](f32x4 (&bo)[KC][NT]) __attribute__((always_inline)) {
    f32x4 dwv[KC][NT];
    const int wb = opq(wbase);
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int cb = 16 * s + 4 * g;
      f32x4 wl[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) wl[k] = *reinterpret_cast<const f32x4*>(WD + k * KC * 16 + cb);   // once per chunk, for the four outputs
      const f32x4 lsc = *reinterpret_cast<const f32x4*>(CS + 2 * KC * 16 + cb);
      const f32x4 lsh = *reinterpret_cast<const f32x4*>(CS + 3 * KC * 16 + cb);
      const float* pl = Tg + ((size_t)(4 * s) * PL + wb) * 4;
      f32x4 d[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) d[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int wr = 0; wr < 4; ++wr) {
        f32x4 row[4];
#pragma unroll
        for (int wc = 0; wc < 4; ++wc) row[wc] = *reinterpret_cast<const f32x4*>(pl + (size_t)(wro[wr] + wco[wc]) * 4);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int ky = wr - dy;
          if (ky < 0 || ky > 2) continue;
#pragma unroll
          for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
              d[2 * dy + dx] = __builtin_elementwise_fma(row[dx + kx], wl[3 * ky + kx], d[2 * dy + dx]);   // packed: two v_pk_fma_f32 per tap
        }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) dwv[s][n] = __builtin_elementwise_fma(d[n], lsc, lsh);
      __builtin_amdgcn_sched_barrier(0);                  // one chunk's taps and window at a time
    }
    BOps b[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = make_b(dwv[0][n], dwv[1][n], dwv[2][n]);
#pragma unroll
    for (int mt = 0; mt < KC; ++mt) {
      f32x4 acc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mfma9(W2P, mt, b, acc);
      if (mt == 0) { watch.see(acc[0][0]); watch.see(acc[1][0]); watch.see(acc[2][0]); watch.see(acc[3][0]); }
      const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 4 * KC * 16 + 16 * mt + 4 * g);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 5 * KC * 16 + 16 * mt + 4 * g);
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = __builtin_fmaf(acc[n][k], sc[k], sh[k]);
          bo[mt][n][k] = u > 0.f ? u : 0.f;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the three quads of the next block's branch input (planes g, 4 + g, 8 + g) at this lane's slots
  auto write_tile = [&](int nt, f32x4 q0, f32x4 q1, f32x4 q2) __attribute__((always_inline)) {
    if (!real[nt]) { q0 = (f32x4){0.f, 0.f, 0.f, 0.f}; q1 = q0; q2 = q0; }   // halo slots stay the zero column
    const int so = opq(sl[nt]);
    *reinterpret_cast<f32x4*>(T1 + ((size_t)(0 + g) * PL + so) * 4) = q0;
    *reinterpret_cast<f32x4*>(T1 + ((size_t)(4 + g) * PL + so) * 4) = q1;
    *reinterpret_cast<f32x4*>(T1 + ((size_t)(8 + g) * PL + so) * 4) = q2;
  };
  auto tbl = [&](int i) __attribute__((always_inline)) {                                 // entry i of this lane group in the CURRENT image: PS[mt] = 0..2, XS[c] = 3..8
    return reinterpret_cast<const int*>(IM + 2 * CH6_WP_FL + Cfg::DW_FL + Cfg::CST_FL)[i * 4 + g];
  };
  // The next block's image travels global -> registers -> LDS in two parts, each held in registers across ONE phase only:
  // its W1 during this block's phase A (stored after the barrier that retires phase A's filter reads - W1 is dead then),
  // the rest (W2, taps, BN vectors, tables) during phase B (stored in the exchange phase).
  constexpr int P1_4 = CH6_WP_FL / 4, P2_4 = (CH6_IMG_FL - CH6_WP_FL) / 4;
  constexpr int NI1 = (P1_4 + THREADS - 1) / THREADS, NI2 = (P2_4 + THREADS - 1) / THREADS;   // 3, 3
  // (Loads and waits are UNCONDITIONAL - the last block re-reads image 0, a thread past the part's end its last quad: a load or
  // a wait behind a predicate leaves a path on which the compiler must assume the load still pending, and it then guards the
  // destination registers wherever they are reused - with a vmcnt(0) right behind the NEXT requests, i.e. a full load latency
  // at the start of phase A and of phase B of every block: per-wave stamps, 8.4 k cycles per phase A against 6.5 k.)
  auto part_issue = [&](int kb_next, int off4, int n4, auto& regs) __attribute__((always_inline)) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.img + (size_t)(kb_next < NB ? kb_next : 0) * CH6_IMG_FL) + off4;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(regs) / sizeof(regs[0])); ++k) { const int i = tid + k * THREADS; regs[k] = src[i < n4 ? i : n4 - 1]; }
  };
  auto part_commit = [&](int off4, int n4, const auto& regs) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the part has landed (and every older request of this wave)
#pragma unroll
    for (int k = 0; k < (int)(sizeof(regs) / sizeof(regs[0])); ++k) { const int i = tid + k * THREADS; if (i < n4) reinterpret_cast<f32x4*>(IM)[off4 + i] = regs[k]; }
  };

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float* ximg = a.in + (size_t)b * HW * C;
    float* zimg = a.out + (size_t)b * HW * C;
    float* pimg = a.park + (size_t)b * NPARK * PP4;       // this image's park scratch: [position 0 .. NPARK)[patch][4 pixels of the patch]
    // ---- everything this image needs from memory up front is requested at once: the image of block 0, and X
    float hold2[6][NT];                                   // X[16 c + 4 g + 2]: branch input 4 c + g of the second block
    {
      f32x4 xq[NT][6];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int c = 0; c < 6; ++c)
          xq[nt][c] = *reinterpret_cast<const f32x4*>(ximg + (size_t)pix[nt] * C + 16 * c + 4 * g);   // halo slots read pixel 0 and are zeroed below
      {
        f32x4 i1[NI1], i2[NI2];
        part_issue(0, 0, P1_4, i1);
        part_issue(0, P1_4, P2_4, i2);
        __builtin_amdgcn_sched_barrier(0);                // all requests are out before the first use
        part_commit(0, P1_4, i1);
        part_commit(P1_4, P2_4, i2);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (!real[nt]) {
#pragma unroll
          for (int c = 0; c < 6; ++c) xq[nt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (valid[nt]) {
#pragma unroll
          for (int j = 0; j < KC; ++j)
            *reinterpret_cast<f32x4*>(T1 + ((size_t)(4 * j + g) * PL + sl[nt]) * 4) = (f32x4){xq[nt][2 * j][1], xq[nt][2 * j][3], xq[nt][2 * j + 1][1], xq[nt][2 * j + 1][3]};
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) hold2[c][nt] = xq[nt][c][2];
      }
      YFV2_WSTAMP(12);
      lds_barrier();                                    // image 0 (with its XS table) is in LDS
      YFV2_WSTAMP(13);
      // X[16 c + 4 g] pass at least two blocks: parked into the group of the block that consumes them
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int pos = tbl(3 + c);
        if (pvalid) {
          if (pos < NPARK) {
            *reinterpret_cast<f32x4*>(pimg + (size_t)pos * PP4 + 4 * patch) = (f32x4){xq[0][c][0], xq[1][c][0], xq[2][c][0], xq[3][c][0]};
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) zimg[(size_t)pix[nt] * C + pos] = xq[nt][c][0];
          }
        }
      }
    }
    YFV2_WSTAMP(1);

    f32x4 bo[KC][NT];
    float Hd[KC][NT];                                     // element 0 of every accumulator quad: branch input of the block after next
    // one block; FIRST (compile time): the peeled first block, whose exchange draws on X's held elements (hold2 dies with it)
    auto run_block = [&](const int kb, auto first_tag) __attribute__((always_inline)) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const bool more = kb + 1 < NB;
      {
        f32x4 n1[NI1];
        part_issue(kb + 1, 0, P1_4, n1);                  // the next block's W1 flies during phase A
        __builtin_amdgcn_sched_barrier(0);
        phase_a();
        __builtin_amdgcn_sched_barrier(0);
        if (kb == 1) YFV2_WSTAMP(6);
        if (FIRST) YFV2_WSTAMP(2);
        lds_barrier();                                  // phase A's W1 reads are done: W1 may be replaced
        if (kb == 1) YFV2_WSTAMP(7);
        if (FIRST) YFV2_WSTAMP(3);
        __builtin_amdgcn_s_waitcnt(0x0F70);               // (see part_issue)
        if (more) part_commit(0, P1_4, n1);
      }
      const int ps0 = tbl(0), ps1 = tbl(1), ps2 = tbl(2);   // this block's park positions: read before the tables are replaced
      f32x4 n2[NI2];
      part_issue(kb + 1, P1_4, P2_4, n2);                 // the rest of the next image flies during phase B
      float plq[3][NT];                                   // parked input i of the patch's pixel nt (consumed in the exchange)
      if constexpr (!FIRST) {
        // the three parked inputs of the next block (group kb - 1: parked two or more blocks ago), used in the exchange
        if (more) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            // FOUR dword loads of the 16-byte record, through opaque offsets so that they are not merged into one: a
            // 4-register tuple held across phase B gets split by the register allocator - v_movs behind a vmcnt(0), i.e.
            // every wave waiting out the load latency here (measured: +19 us per launch)
            const float* rec = pimg + (size_t)(12 * (kb - 1) + 3 * g + i) * PP4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) plq[i][nt] = rec[opq(4 * patch + nt)];   // (lanes without a patch: in bounds, unused)
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      phase_b(bo);
      if (kb == 1) YFV2_WSTAMP(8);
      if (FIRST) YFV2_WSTAMP(4);
      // Everything this wave has in flight is consumed right below.  Said HERE, on a path every lane of every block takes: the
      // consuming code sits behind lane predicates and behind `more`, so the compiler must assume the loads are still pending at
      // the loop's back edge and guards their destination registers at the top of the next block - a vmcnt(0) right behind the
      // next image's requests, i.e. a full load latency at the start of every phase A (per-wave stamps: 8.4 k cycles, 6.5 k in
      // the peeled first block).
      __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0)
      if (more) {
        lds_barrier();                                  // every window, filter and table read of this block is done
        part_commit(P1_4, P2_4, n2);                      // the rest of the next block's image
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (valid[nt]) {
            if constexpr (FIRST)
              write_tile(nt, (f32x4){hold2[0][nt], hold2[1][nt], hold2[2][nt], hold2[3][nt]},
                         (f32x4){hold2[4][nt], hold2[5][nt], bo[0][nt][1], bo[0][nt][3]},
                         (f32x4){bo[1][nt][1], bo[1][nt][3], bo[2][nt][1], bo[2][nt][3]});
            else
              write_tile(nt, (f32x4){Hd[0][nt], Hd[1][nt], Hd[2][nt], plq[0][nt]},
                         (f32x4){plq[1][nt], plq[2][nt], bo[0][nt][1], bo[0][nt][3]},
                         (f32x4){bo[1][nt][1], bo[1][nt][3], bo[2][nt][1], bo[2][nt][3]});
          }
#pragma unroll
          for (int mt = 0; mt < KC; ++mt) Hd[mt][nt] = bo[mt][nt][0];
        }
        if (pvalid) {   // elements 2: parked for the block after next (one 16-byte store per position), or - nobody consumes them - at their final place
          const int ps[3] = {ps0, ps1, ps2};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (ps[i] < NPARK) {
              *reinterpret_cast<f32x4*>(pimg + (size_t)ps[i] * PP4 + 4 * patch) = (f32x4){bo[i][0][2], bo[i][1][2], bo[i][2][2], bo[i][3][2]};
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) zimg[(size_t)pix[nt] * C + ps[i]] = bo[i][nt][2];
            }
          }
        }
        lds_barrier();
        if (kb == 1) YFV2_WSTAMP(9);
        if (FIRST) YFV2_WSTAMP(5);
      }
    };
    run_block(0, std::true_type{});
#pragma unroll 1
    for (int kb = 1; kb < NB; ++kb) run_block(kb, std::false_type{});
    YFV2_WSTAMP(10);
    // ---- the last block's output in accumulator order, the held elements of the block before it behind them
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      if (real[nt]) {
        float* zp = zimg + (size_t)pix[nt] * C;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) *reinterpret_cast<f32x4*>(zp + 16 * mt + 4 * g) = bo[mt][nt];
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) zp[C2 + 3 * g + mt] = Hd[mt][nt];
      }
    YFV2_WSTAMP(11);
    __syncthreads();                                      // tile and image buffer are rewritten by the next image
  }
  watch.report(a.nonfinite);
}

static long s1chain_lds_floats(int H, int W) {
  const long pl = s1chain_plane(H / 2, W / 2);               // 16-byte slots per plane: [row parity][haloed row / 2][pitch], rounded up to 16
  return (long)CH6_IMG_FL + 12L * pl * 4;
}

long yfv2_s1chain_park_floats(int H, int W, int nblk) { return 12L * (nblk - 2) * ((((H / 2) * (W / 2) + 15) & ~15) * 4); }

int yfv2_s1chain_image_floats() { return CH6_IMG_FL; }

bool yfv2_s1chain_supported(int c2, int H, int W) {
  if (c2 != 48 || H < 2 || (H & 1) || (W & 1)) return false;
  if ((H / 2) * (W / 2) > 16 * 8) return false;            // one 2x2 patch per (wave, lane & 15): 8 waves
  if (yfv2_s1chain_park_floats(H, W, 7) > 24L * (4 * H) * (4 * W)) return false;   // the park scratch is a stem-sized temporary (yfv2_api.hip: t1)
  return s1chain_lds_floats(H, W) * 4 <= 160 * 1024;
}

bool yfv2_launch_block_s1chain(const BlockS1Args& a, hipStream_t s) {
  if (!yfv2_s1chain_supported(48, a.H, a.W) || a.nblk < 2) return false;
  const size_t lds = sizeof(float) * (size_t)s1chain_lds_floats(a.H, a.W);
  const int blocks = a.B < 256 ? a.B : 256;
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s1chain6_kernel<512>), lds_ok);
  YFV2_LAUNCH((block_s1chain6_kernel<512>), dim3(blocks), dim3(512), lds, s, a);
  return true;
}

// ============================================================================
// A chain of stride-1 blocks with the WHOLE activation resident in LDS (C2 = 96, maps up to 128 pixels): stage 4's blocks 1..3
// ============================================================================
// Reference: N consecutive ShuffleV2Block(stride 1) at 192 channels (shufflenetv2.py:19-32,48-51,57-63).  An 11x11x192
// image is 93 KB: it fits one CU's LDS whole, in the reference's own channel order, so the chain needs no channel
// bookkeeping at all: per block
//   pw1 reads the ODD channels of the pool (two 16-byte reads per lane and chunk, odd elements kept), in three passes of
//   32 output channels each (the filters do not fit next to the pool: every pass brings its own 27 KB image - 32 rows of
//   W1, 32 columns of W2, their taps and BN vectors - prefetched into registers during the pass before);
//   a pass's pw1 output (+BN+ReLU) goes to a small zero-bordered tile T[13][13][32], the depthwise 3x3 (+BN) is formed in
//   registers from it as the MFMA B fragment and multiplied into the pw2 accumulators (K split over the passes);
//   after the third pass the accumulators (+BN+ReLU) are the block's 96 fresh channels; the pool is then rewritten in place
//   to cat(even channels, fresh) = the next block's input (evens gathered through registers between two barriers).
// One wave = one 16-pixel tile (8 waves, 121 pixels).  Per image: one coalesced read and one coalesced write of the
// activation instead of three of each, one launch instead of three.
constexpr int P96_C2 = 96, P96_C = 192, P96_KC = 6, P96_TH = 3, P96_TC = 32;     // thirds of 32 channels = 2 M tiles = 2 chunks
constexpr int P96_CPP = P96_C + 4;                                               // floats per pool pixel
constexpr int P96_TP = P96_TC + 4;                                               // floats per T pixel
constexpr int P96_W1_FL = 2 * P96_KC * 256, P96_W2_FL = P96_KC * 2 * 256;        // fragment-major, per third
constexpr int P96_IMG_FL = P96_W1_FL + P96_W2_FL + 9 * P96_TC + 4 * P96_TC + 2 * P96_C2;   // + taps, sc1 sh1 scd shd, sc2 sh2
// PRE: W1 / W2 pre-split into bf16 hi / mid / lo operand quads per chunk pair (WeightPacker, as block_s2w_kernel's W1):
// [mt][pair][term][64][4] - W1 of a third = 2 tiles x 3 pairs, W2 = 6 tiles x 1 pair; 1.5x the fp32 fragments
constexpr int P96_W1P_FL = 2 * (P96_KC / 2) * 2 * 256, P96_W2P_FL = P96_KC * 1 * 2 * 256;   // PRE: two fp16 terms per chunk pair (fp16x3)
constexpr int P96_IMGP_FL = P96_W1P_FL + P96_W2P_FL + 9 * P96_TC + 4 * P96_TC + 2 * P96_C2;
constexpr int P96_MAXPX = 128;

// PRE (bf16x6 on pre-split filters): the kernel is bound by its fp32 MFMAs (6912 per image: 55 k SIMD-cycles of the launch's
// 115 k) - the six-product form needs 2.67x fewer matrix-core cycles, and with the filters split on the host the only VALU
// work added is the split of the B operands: pw1's once per BLOCK (the pool's odd channels do not change during its three
// passes: nine operand quads stay in registers), pw2's depthwise result once per pass.
template <int THREADS, bool PRE>
__global__ __launch_bounds__(THREADS, 2) void block_s1pool_kernel(BlockS1Args a) {
  constexpr int NW = THREADS / 64;
  constexpr int IMG_FL = PRE ? P96_IMGP_FL : P96_IMG_FL;
  constexpr int W1_FL = PRE ? P96_W1P_FL : P96_W1_FL, W2_FL = PRE ? P96_W2P_FL : P96_W2_FL;
  constexpr int N4 = IMG_FL / 4, NIT = (N4 + THREADS - 1) / THREADS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* POOL = lds;                                   // [H * W][P96_CPP]
  float* T = POOL + a.H * a.W * P96_CPP;               // [(H+2)][(W+2)][P96_TP], border zero
  float* IMG = T + (a.H + 2) * (a.W + 2) * P96_TP;     // one third's image
  Yfv2Watch watch;                                     // range guard of the fp16x3 products (yfv2_internal.h)
  const int H = a.H, W = a.W, HW = H * W, WP = W + 2, NB = a.nblk;
  const float invW = 1.0f / (float)W;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int px = 16 * wave + p;
  const bool pv = px < HW;
  const int pxc = pv ? px : HW - 1;
  const int py = yfv2_fdiv(pxc, invW), pxx = pxc - py * W;
  float* tpix = T + ((py + 1) * WP + pxx + 1) * P96_TP;            // this lane's pixel in T
  const float* twin = T + (py * WP + pxx) * P96_TP;                // top-left of its 3x3 window
  float* ppix = POOL + pxc * P96_CPP;
  static_assert(NW * 16 >= P96_MAXPX, "one tile per wave");

  YFV2_WSTAMP(0);
  for (int i = tid; i < (H + 2) * WP * P96_TP / 4; i += THREADS) reinterpret_cast<f32x4*>(T)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    // ---- the image into the pool (coalesced 16-byte copy), the first third's filters into IMG
    {
      // every load of a batch is issued before its first store (the plain `dst[i] = src[i]` loop compiles to one global
      // round trip per 16 bytes and thread: 12 + 5 of them here)
      const f32x4* src = reinterpret_cast<const f32x4*>(a.in + (size_t)b * HW * P96_C);
      const f32x4* isrc = reinterpret_cast<const f32x4*>(a.img);
      constexpr int NPX = (P96_MAXPX * (P96_C / 4) + THREADS - 1) / THREADS;   // 12
      const int n4 = HW * (P96_C / 4);
      f32x4 ti[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; ti[k] = i < N4 ? isrc[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int k0 = 0; k0 < NPX; k0 += 6) {
        f32x4 tp[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { const int i = tid + (k0 + k) * THREADS; tp[k] = i < n4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int i = tid + (k0 + k) * THREADS;
          const int ipx = i / (P96_C / 4), q = i - ipx * (P96_C / 4);
          if (i < n4) *reinterpret_cast<f32x4*>(POOL + ipx * P96_CPP + 4 * q) = tp[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; if (i < N4) reinterpret_cast<f32x4*>(IMG)[i] = ti[k]; }
    }
    __syncthreads();
    YFV2_WSTAMP(1);

#pragma unroll 1
    for (int blk = 0; blk < NB; ++blk) {
      f32x4 acc2[P96_KC];
#pragma unroll
      for (int mt = 0; mt < P96_KC; ++mt) acc2[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // PRE: pw1's B operands for the whole block - the 96 odd pool channels of this lane's pixel, per chunk pair
      // (fp16x3, round 3: x 2^4, two fp16 terms; round 2's form was bf16x6 - three terms, six products)
      yfv2_h8c b1a[PRE ? P96_KC / 2 : 1], b1b[PRE ? P96_KC / 2 : 1];
      auto split_pair = [&](f32x4 c0, f32x4 c1, yfv2_h8c& t1, yfv2_h8c& t2) __attribute__((always_inline)) {
        u32x4 u1, u2;
        const f32x4 cs[2] = {c0 * 16.0f, c1 * 16.0f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const yfv2_h4c h1 = __builtin_convertvector(cs[e], yfv2_h4c);
          const yfv2_h4c h2 = __builtin_convertvector(cs[e] - __builtin_convertvector(h1, f32x4), yfv2_h4c);
          const yfv2_u2c a1 = __builtin_bit_cast(yfv2_u2c, h1), a2 = __builtin_bit_cast(yfv2_u2c, h2);
          u1[2 * e] = a1[0]; u1[2 * e + 1] = a1[1]; u2[2 * e] = a2[0]; u2[2 * e + 1] = a2[1];
        }
        t1 = __builtin_bit_cast(yfv2_h8c, u1); t2 = __builtin_bit_cast(yfv2_h8c, u2);
      };
      if constexpr (PRE) {
#pragma unroll
        for (int sp = 0; sp < P96_KC / 2; ++sp) {
          f32x4 c[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int s2 = 2 * sp + e;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(ppix + 32 * s2 + 8 * g);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(ppix + 32 * s2 + 8 * g + 4);
            c[e] = (f32x4){q0[1], q0[3], q1[1], q1[3]};
          }
          split_pair(c[0], c[1], b1a[sp], b1b[sp]);
        }
      }
#pragma unroll 1
      for (int th = 0; th < P96_TH; ++th) {
        const int t = blk * P96_TH + th;
        const bool more = t + 1 < NB * P96_TH;
        const float* W1t = IMG;
        const float* W2t = IMG + W1_FL;
        const float* TAPS = W2t + W2_FL;               // [9][32]
        const float* CSV = TAPS + 9 * P96_TC;          // sc1 sh1 scd shd [32] | sc2 sh2 [96]
        // the next third's image travels through registers while this one computes
        f32x4 nimg[NIT];
        {
          const f32x4* isrc = reinterpret_cast<const f32x4*>(a.img + (size_t)(t + 1) * IMG_FL);
#pragma unroll
          for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; nimg[k] = (more && i < N4) ? isrc[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        // ---- pw1 (+BN+ReLU): output channels 32 th .. +31 of the branch, K = the 96 odd channels of the pool
        {
          f32x4 acc1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          if constexpr (PRE) {
#pragma unroll
            for (int sp = 0; sp < P96_KC / 2; ++sp) {
              yfv2_h8c a1[2], a2[2];
#pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                const float* wq2 = W1t + (((mt * (P96_KC / 2) + sp) * 2) * 64 + lane) * 4;
                a1[mt] = __builtin_bit_cast(yfv2_h8c, *reinterpret_cast<const u32x4*>(wq2));
                a2[mt] = __builtin_bit_cast(yfv2_h8c, *reinterpret_cast<const u32x4*>(wq2 + 256));
              }
#define P96_PROD(A_, B_) _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[mt], B_, acc1[mt], 0, 0, 0);
              P96_PROD(a1, b1b[sp]) P96_PROD(a2, b1a[sp]) P96_PROD(a1, b1a[sp])
#undef P96_PROD
            }
            watch.see(acc1[0][0]);
          } else {
#pragma unroll
          for (int s2 = 0; s2 < P96_KC; ++s2) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(ppix + 32 * s2 + 8 * g);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(ppix + 32 * s2 + 8 * g + 4);
            const f32x4 bf = {q0[1], q0[3], q1[1], q1[3]};               // branch inputs 16 s2 + 4 g .. +3 = odd pool channels
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(W1t + ((0 * P96_KC + s2) * 64 + lane) * 4);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(W1t + ((1 * P96_KC + s2) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bf[j], acc1[0], 0, 0, 0);
              acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], bf[j], acc1[1], 0, 0, 0);
            }
          }
          }
          if (pv) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const f32x4 sc = *reinterpret_cast<const f32x4*>(CSV + 0 * P96_TC + 16 * mt + 4 * g);
              const f32x4 sh = *reinterpret_cast<const f32x4*>(CSV + 1 * P96_TC + 16 * mt + 4 * g);
              f32x4 y;
#pragma unroll
              for (int c = 0; c < 4; ++c) { const float u = __builtin_fmaf(acc1[mt][c], sc[c], sh[c]); y[c] = u > 0.f ? u : 0.f; }
              *reinterpret_cast<f32x4*>(tpix + 16 * mt + 4 * g) = y;
            }
          }
        }
        if (blk == 0) YFV2_WSTAMP(2 + 4 * th);
        __syncthreads();   // the depthwise windows reach into the neighbours' pixels
        if (blk == 0) YFV2_WSTAMP(3 + 4 * th);
        // ---- dw3x3 (+BN) in registers -> pw2 partial sums over this third's 32 input channels
        if constexpr (PRE) {
          f32x4 bfr2[2];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const int cb = 16 * c2 + 4 * g;
            f32x4 win[9], wl[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
              win[k] = *reinterpret_cast<const f32x4*>(twin + ((k / 3) * WP + (k % 3)) * P96_TP + cb);
              wl[k] = *reinterpret_cast<const f32x4*>(TAPS + k * P96_TC + cb);
            }
            const f32x4 dsc = *reinterpret_cast<const f32x4*>(CSV + 2 * P96_TC + cb);
            const f32x4 dsh = *reinterpret_cast<const f32x4*>(CSV + 3 * P96_TC + cb);
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
              for (int c = 0; c < 4; ++c) d[c] = __builtin_fmaf(win[k][c], wl[k][c], d[c]);
#pragma unroll
            for (int c = 0; c < 4; ++c) bfr2[c2][c] = __builtin_fmaf(d[c], dsc[c], dsh[c]);
          }
          yfv2_h8c x1, x2;
          split_pair(bfr2[0], bfr2[1], x1, x2);
          yfv2_h8c a1[P96_KC], a2[P96_KC];
#pragma unroll
          for (int mt = 0; mt < P96_KC; ++mt) {
            const float* wq2 = W2t + ((mt * 2) * 64 + lane) * 4;
            a1[mt] = __builtin_bit_cast(yfv2_h8c, *reinterpret_cast<const u32x4*>(wq2));
            a2[mt] = __builtin_bit_cast(yfv2_h8c, *reinterpret_cast<const u32x4*>(wq2 + 256));
          }
#define P96_PROD(A_, B_) _Pragma("unroll") for (int mt = 0; mt < P96_KC; ++mt) acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[mt], B_, acc2[mt], 0, 0, 0);
          P96_PROD(a1, x2) P96_PROD(a2, x1) P96_PROD(a1, x1)
#undef P96_PROD
        } else {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int cb = 16 * c2 + 4 * g;
          f32x4 win[9], wl[9], af[P96_KC];
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            win[k] = *reinterpret_cast<const f32x4*>(twin + ((k / 3) * WP + (k % 3)) * P96_TP + cb);
            wl[k] = *reinterpret_cast<const f32x4*>(TAPS + k * P96_TC + cb);
          }
          const f32x4 dsc = *reinterpret_cast<const f32x4*>(CSV + 2 * P96_TC + cb);
          const f32x4 dsh = *reinterpret_cast<const f32x4*>(CSV + 3 * P96_TC + cb);
#pragma unroll
          for (int mt = 0; mt < P96_KC; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(W2t + ((mt * 2 + c2) * 64 + lane) * 4);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 d = {0.f, 0.f, 0.f, 0.f}, bfr;
#pragma unroll
          for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[c] = __builtin_fmaf(win[k][c], wl[k][c], d[c]);
#pragma unroll
          for (int c = 0; c < 4; ++c) bfr[c] = __builtin_fmaf(d[c], dsc[c], dsh[c]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < P96_KC; ++mt) acc2[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][j], bfr[j], acc2[mt], 0, 0, 0);
        }
        }
        if (th == P96_TH - 1) {                         // pw2's BN + ReLU (the vectors leave with this image)
          if constexpr (PRE) watch.see(acc2[0][0]);
#pragma unroll
          for (int mt = 0; mt < P96_KC; ++mt) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(CSV + 4 * P96_TC + 16 * mt + 4 * g);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(CSV + 4 * P96_TC + P96_C2 + 16 * mt + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float u = __builtin_fmaf(acc2[mt][c], sc[c], sh[c]); acc2[mt][c] = u > 0.f ? u : 0.f; }
          }
        }
        if (blk == 0) YFV2_WSTAMP(4 + 4 * th);
        __syncthreads();   // T and IMG are free
        if (more) {
#pragma unroll
          for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; if (i < N4) reinterpret_cast<f32x4*>(IMG)[i] = nimg[k]; }
        }
        if (th < P96_TH - 1) __syncthreads();            // (after the last third the barriers of the pool rewrite follow)
        if (blk == 0) YFV2_WSTAMP(5 + 4 * th);
      }
      // ---- pool <- cat(even channels, fresh): evens gathered through registers between two barriers
      {
        constexpr int NQ = P96_C2 / 4;                  // 24 output quads of pass-through channels per pixel
        constexpr int NEV = (P96_MAXPX * NQ + THREADS - 1) / THREADS;   // (registers for up to 128 pixels; the pool itself holds H * W)
        f32x4 ev[NEV];
#pragma unroll
        for (int k = 0; k < NEV; ++k) {
          const int i = tid + k * THREADS;
          const int ipx = i / NQ, q = i - ipx * NQ;
          ev[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (ipx < HW) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(POOL + ipx * P96_CPP + 8 * q);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(POOL + ipx * P96_CPP + 8 * q + 4);
            ev[k] = (f32x4){q0[0], q0[2], q1[0], q1[2]};
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NEV; ++k) {
          const int i = tid + k * THREADS;
          const int ipx = i / NQ, q = i - ipx * NQ;
          if (ipx < HW) *reinterpret_cast<f32x4*>(POOL + ipx * P96_CPP + 4 * q) = ev[k];
        }
        if (pv) {
#pragma unroll
          for (int mt = 0; mt < P96_KC; ++mt) *reinterpret_cast<f32x4*>(ppix + P96_C2 + 16 * mt + 4 * g) = acc2[mt];
        }
        __syncthreads();
        YFV2_WSTAMP(14 + blk);
      }
    }
    // ---- the pool out (coalesced), then the barrier that frees it for the next image
    {
      f32x4* dst = reinterpret_cast<f32x4*>(a.out + (size_t)b * HW * P96_C);
      for (int i = tid; i < HW * (P96_C / 4); i += THREADS) {
        const int ipx = i / (P96_C / 4), q = i - ipx * (P96_C / 4);
        dst[i] = *reinterpret_cast<const f32x4*>(POOL + ipx * P96_CPP + 4 * q);
      }
    }
    YFV2_WSTAMP(17);
    __syncthreads();
  }
  watch.report(a.nonfinite);
}

static long s1pool_lds_floats(int H, int W, bool pre) { return (long)H * W * P96_CPP + (long)(H + 2) * (W + 2) * P96_TP + (pre ? P96_IMGP_FL : P96_IMG_FL); }
int yfv2_s1pool_image_floats(bool presplit) { return presplit ? P96_IMGP_FL : P96_IMG_FL; }
bool yfv2_s1pool_supported(int c2, int H, int W) {
  if (c2 != P96_C2 || H * W > P96_MAXPX || H * W < 1) return false;
  return s1pool_lds_floats(H, W, true) * 4 <= 160 * 1024;
}

bool yfv2_launch_block_s1pool(const BlockS1Args& a, hipStream_t s) {
  if (!yfv2_s1pool_supported(P96_C2, a.H, a.W) || a.nblk < 1) return false;
  const size_t lds = sizeof(float) * (size_t)s1pool_lds_floats(a.H, a.W, a.presplit != 0);
  const int blocks = a.B < 256 ? a.B : 256;
  static std::atomic<unsigned long long> lds_ok0{0}, lds_ok1{0};
  if (a.presplit) {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s1pool_kernel<512, true>), lds_ok1);
    YFV2_LAUNCH((block_s1pool_kernel<512, true>), dim3(blocks), dim3(512), lds, s, a);
  } else {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s1pool_kernel<512, false>), lds_ok0);
    YFV2_LAUNCH((block_s1pool_kernel<512, false>), dim3(blocks), dim3(512), lds, s, a);
  }
  return true;
}

// ============================================================================
// fused DWConvblock half ("tower half"), 72 channels: shared constants (kernel: tower2_kernel below)
// ============================================================================
constexpr int TW_C = 72, TW_KC = 5;
constexpr int TW_WP_FL = TW_KC * TW_KC * 256;   // pointwise filter, fragment-major
constexpr int TW_WH_FL = TW_KC * 256;           // per output-conv M tile

// ============================================================================
// fused ShuffleV2 stride-2 block (first block of a stage)
// ============================================================================
// Reference (model/backbone/shufflenetv2.py:19-44,52-55), input c = CIN channels at HxW:
//   proj = ReLU(BN(pw(BN(dw3x3 s2(x)))))                       CIN -> CIN at H/2 x W/2
//   main = ReLU(BN(pw2(BN(dw3x3 s2(ReLU(BN(pw1(x))))))))       CIN -> CIN
//   out  = cat(proj, main)                                     2*CIN channels
// Work item = (image, R output rows).  stage: the 2R+1 raw input rows the tile's depthwise
// windows touch are copied into the LDS tile T1 by all threads at once (left zero column, zero
// top row at the image edge).  proj: per 16 output pixels each lane forms the stride-2
// depthwise (+BN) of its pixel / its 4 channels in registers from T1 = B fragment of the proj
// pointwise.  pw1 (+BN+ReLU) then overwrites T1 in place, and the main branch repeats the
// depthwise -> pointwise step on it.  The input leaves HBM once (the first version gathered the
// proj taps from global: 1.8x the algorithmic traffic by PMC); one launch replaces five and
// removes four intermediate tensors.
template <int CIN>
struct S2Cfg {
  static constexpr int KC = (CIN + 15) / 16;
  static constexpr int CP = CIN + 4;
  static constexpr int W_FL = KC * KC * 256;  // fragment-major filter (see S1Cfg)
  static constexpr int DW_FL = 9 * KC * 16;
  static constexpr int NCS = 10;  // sc1 sh1 scd shd sc2 sh2 scpd shpd scpp shpp
  // static bounds of the staged tile (512 threads, 8 waves), enforced by yfv2_block_s2_rows:
  static constexpr int MAXP = 14;                     // staged 16-byte quads per thread
  static constexpr int MAXT = CIN <= 24 ? 10 : 6;     // pw1 pixel tiles per wave
};

template <int CIN, int THREADS, bool PPIN>
__global__ __launch_bounds__(THREADS) void block_s2_kernel(BlockS2Args a) {
  using Cfg = S2Cfg<CIN>;
  constexpr int KC = Cfg::KC, CP = Cfg::CP, KS = KC * 16;
  constexpr int NW = THREADS / 64;
  constexpr int CO = 2 * CIN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W1 = lds;
  float* W2 = W1 + Cfg::W_FL;
  float* WJ = W2 + Cfg::W_FL;       // proj pointwise
  float* WD = WJ + Cfg::W_FL;       // main depthwise taps [9][KS]
  float* WE = WD + Cfg::DW_FL;      // proj depthwise taps [9][KS]
  float* CS = WE + Cfg::DW_FL;      // [10][KS]
  float* T1 = CS + Cfg::NCS * KS;
  const int H = a.H, W = a.W, R = a.R, OH = H >> 1, OW = W >> 1;
  const float invW = 1.0f / (float)W, invOW = 1.0f / (float)OW;
  const int WP = W + 1;
  const int t1_fl = (2 * R + 1) * WP * CP + 16;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, wave = tid >> 6;

  // ---- cooperative staging of one item's raw input tile (input rows 2*y0-1 .. 2*y0+2*rows-1):
  // every thread requests up to MAXP 16-byte quads at once -> one global-latency round per item,
  // overlapped with the LDS prologue (first item) or the previous item's tail
  const int tiles_per_img = (OH + R - 1) / R;
  const int n_items = a.B * tiles_per_img;
  constexpr int QPP = CIN / 4;
  constexpr int MAXP = Cfg::MAXP;
  // PPIN: the input is stage 2's pair-plane layout (yfv2_stage2.hip): per pair plane the tile's rows are ONE
  // contiguous run of 8-byte pairs, staged into channel positions 2p, 2p+1 of the tile (the filters were
  // re-ordered to slot order on the host).
  // Thread t owns tile pixels t and t + THREADS (a tile has at most 14*THREADS/QPP <= 2*THREADS pixels) in EVERY
  // pair plane: the pixel -> (row, column) split is done once per thread, the plane loop only adds uniform offsets.
  constexpr int NPL = CIN / 2;
  constexpr int NST = PPIN ? 2 * NPL : 2 * MAXP;   // f32x2 registers (NHWC mode: MAXP float4 as two consecutive entries)
  f32x2 st[NST];
  auto stage_issue = [&](int item_, bool active) {  // always (re)defines every staged register
    const int item = active ? item_ : 0;
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R, rows = min(R, OH - y0);
    if constexpr (PPIN) {
      const int npx = active ? (2 * rows + 1) * W : 0;    // pixels of the tile (rows 2*y0-1 .. 2*y0+2*rows-1): one contiguous run per plane
      const int g0 = (2 * y0 - 1) * W;                    // plane pixel index of the tile's first pixel; < 0 only in image row -1
      const bool okA = tid < npx && g0 + tid >= 0, okB = tid + THREADS < npx && g0 + tid + THREADS >= 0;
      const float* pA = a.in + (size_t)b * (size_t)a.pp_imgstride + (okA ? (size_t)(g0 + tid) * 2 : 0);
      const float* pB = a.in + (size_t)b * (size_t)a.pp_imgstride + (okB ? (size_t)(g0 + tid + THREADS) * 2 : 0);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const size_t po = (size_t)pl * H * W * 2 + (((a.pp_mask >> pl) & 1u) ? (size_t)a.pp_bufstride : 0);   // uniform
        st[2 * pl] = okA ? *reinterpret_cast<const f32x2*>(pA + po) : (f32x2){0.f, 0.f};
        st[2 * pl + 1] = okB ? *reinterpret_cast<const f32x2*>(pB + po) : (f32x2){0.f, 0.f};
      }
    } else {
      const int nq = active ? (2 * rows + 1) * W * QPP : 0;
      const size_t in_px = (size_t)b * H * W;
#pragma unroll
      for (int j = 0; j < MAXP; ++j) {
        const int i = tid + j * THREADS;
        const int pix = i / QPP, q = i - pix * QPP;
        const int r = yfv2_fdiv(pix, invW), x = pix - r * W;
        const int gy = 2 * y0 - 1 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < nq && gy >= 0 && gy < H) v = *reinterpret_cast<const f32x4*>(a.in + (in_px + (size_t)gy * W + x) * CIN + 4 * q);
        st[2 * j] = (f32x2){v[0], v[1]};
        st[2 * j + 1] = (f32x2){v[2], v[3]};
      }
    }
  };
  auto stage_commit = [&](int item) {
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R, rows = min(R, OH - y0);
    if constexpr (PPIN) {
      const int npx = (2 * rows + 1) * W;
      const int rA = yfv2_fdiv(tid, invW), xA = tid - rA * W, rB = yfv2_fdiv(tid + THREADS, invW), xB = tid + THREADS - rB * W;
      float* tA = T1 + (rA * WP + xA + 1) * CP;
      float* tB = T1 + (rB * WP + xB + 1) * CP;
      const bool okA = tid < npx, okB = tid + THREADS < npx;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        if (okA) *reinterpret_cast<f32x2*>(tA + 2 * pl) = st[2 * pl];
        if (okB) *reinterpret_cast<f32x2*>(tB + 2 * pl) = st[2 * pl + 1];
      }
    } else {
      const int nq = (2 * rows + 1) * W * QPP;
#pragma unroll
      for (int j = 0; j < MAXP; ++j) {
        const int i = tid + j * THREADS;
        if (i >= nq) continue;
        const int pix = i / QPP, q = i - pix * QPP;
        const int r = yfv2_fdiv(pix, invW), x = pix - r * W;
        *reinterpret_cast<f32x4*>(T1 + (r * WP + x + 1) * CP + 4 * q) = (f32x4){st[2 * j][0], st[2 * j][1], st[2 * j + 1][0], st[2 * j + 1][1]};
      }
    }
  };
  stage_issue(blockIdx.x, (int)blockIdx.x < n_items);

  // prologue: the LDS image (filters, taps, BN constants - padded and zero-filled on the host,
  // yfv2_load_weights) is one straight coalesced 16-byte copy
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    constexpr int N4 = (3 * Cfg::W_FL + 2 * Cfg::DW_FL + Cfg::NCS * KS) / 4;
    constexpr int NIT = (N4 + THREADS - 1) / THREADS;  // <= 11: every load is issued before the first store
    f32x4 tmp[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; tmp[k] = i < N4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; if (i < N4) dst[i] = tmp[k]; }
  }
  for (int i = tid; i < t1_fl / 4; i += THREADS) reinterpret_cast<f32x4*>(T1)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};  // column 0 (input col -1) stays zero
  __syncthreads();

  int it = 0;   // per-wave stamps of the workgroup's first three items (tools/trace_waves.py): 1 + 7 * it + {staged, proj, barrier, pw1 (+ barrier), main, barrier}
#define S2_STAMP(k) do { if (it < 3) YFV2_WSTAMP(1 + 7 * it + (k)); } while (0)
  YFV2_WSTAMP(0);
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R;                 // first output row
    const int rows = min(R, OH - y0);
    const int iy0 = 2 * y0 - 1;            // first input row of T1
    const size_t in_px = (size_t)b * H * W;
    const size_t out_px = (size_t)b * OH * OW;
    const int npxA = (2 * rows + 1) * W;
    const int npxB = rows * OW;

    // ---- stage: the raw input rows iy0 .. iy0+2*rows (requested before the prologue / at the
    // end of the previous item) land in T1; rows above the image and column -1 are zero
    stage_commit(item);
    __syncthreads();
    S2_STAMP(0);

    // ---- one depthwise(s2, 3x3, +BN) -> pointwise(+BN+ReLU) branch over the tile's output pixels,
    // depthwise taps read from T1: branch 1 = proj on the RAW input (T1 as staged),
    // branch 0 = main on pw1's output (T1 after the in-place pass below)
    auto dw_pw_branch = [&](const int branch) {
      const float* taps = branch == 0 ? WD : WE;
      const float* wmat = branch == 0 ? W2 : WJ;
      const float* dsc_p = CS + (branch == 0 ? 2 : 6) * KS;
      const float* dsh_p = CS + (branch == 0 ? 3 : 7) * KS;
      const float* psc_p = CS + (branch == 0 ? 4 : 8) * KS;
      const float* psh_p = CS + (branch == 0 ? 5 : 9) * KS;
      for (int t = wave; t * 16 < npxB; t += NW) {
        const int q = 16 * t + p;
        const bool pv = q < npxB;
        const int qc = pv ? q : npxB - 1;
        const int r = yfv2_fdiv(qc, invOW), x = qc - r * OW;
        const int oy = y0 + r;
        const float* tp = T1 + ((2 * r) * WP + 2 * x) * CP;  // window rows 2r..2r+2, T1 cols 2x..2x+2
        f32x4 bfr[KC];
#pragma unroll 1
        for (int s = 0; s < KC; ++s) {
          const int cb = 16 * s + 4 * g;
          f32x4 win[9], wk[9];  // the chunk's window and taps in flight together, then the FMAs
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            win[k] = *reinterpret_cast<const f32x4*>(tp + ((k / 3) * WP + (k % 3)) * CP + cb);
            wk[k] = *reinterpret_cast<const f32x4*>(taps + k * KS + cb);
          }
          const f32x4 dsc = *reinterpret_cast<const f32x4*>(dsc_p + cb);
          const f32x4 dsh = *reinterpret_cast<const f32x4*>(dsh_p + cb);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[c] = __builtin_fmaf(win[k][c], wk[k][c], d[c]);
          f32x4 y;
#pragma unroll
          for (int c = 0; c < 4; ++c) y[c] = cb < CIN ? __builtin_fmaf(d[c], dsc[c], dsh[c]) : 0.f;
          if (s == 0) bfr[0] = y;
          if (KC > 1 && s == 1) bfr[KC > 1 ? 1 : 0] = y;
          if (KC > 2 && s == 2) bfr[KC > 2 ? 2 : 0] = y;
        }
        float* dst = a.out + (out_px + (size_t)oy * OW + x) * CO + (branch == 0 ? CIN : 0);
        f32x4 afB[KC][KC], accB[KC];
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) {
          accB[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KC; ++s) afB[mt][s] = *reinterpret_cast<const f32x4*>(wmat + ((mt * KC + s) * 64 + lane) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KC; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < KC; ++mt)
              accB[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afB[mt][s][j], bfr[s][j], accB[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) {
          const int cb = 16 * mt + 4 * g;
          if (pv && cb < CIN) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(psc_p + cb);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(psh_p + cb);
            f32x4 y;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float u = __builtin_fmaf(accB[mt][k], sc[k], sh[k]);
              y[k] = u > 0.f ? u : 0.f;
            }
            *reinterpret_cast<f32x4*>(dst + cb) = y;
          }
        }
      }
    };

    // ================= proj branch first: its depthwise reads the raw tile
    dw_pw_branch(1);
    S2_STAMP(1);
    __syncthreads();
    S2_STAMP(2);

    // ================= pw1 (+BN+ReLU) IN PLACE over the staged tile.  A wave takes whole 16-pixel tiles with the
    // filter's KC*KC A fragments held in registers: a tile's output overwrites exactly the pixels its own B fragments
    // came from, so no barrier separates reads from writes; the next tile's B fragments are fetched before the
    // current tile's MFMAs (same scheme as block_s1_kernel's phase A).
    {
      f32x4 aw[KC][KC], sc1[KC], sh1[KC];
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) {
#pragma unroll
        for (int s = 0; s < KC; ++s) {
          const f32x4 fr = *reinterpret_cast<const f32x4*>(W1 + ((mt * KC + s) * 64 + lane) * 4);
          aw[mt][s] = fr;
        }
        sc1[mt] = *reinterpret_cast<const f32x4*>(CS + 0 * KS + 16 * mt + 4 * g);
        sh1[mt] = *reinterpret_cast<const f32x4*>(CS + 1 * KS + 16 * mt + 4 * g);
      }
      auto tile_off = [&](int t) {
        const int q = 16 * t + p;
        const int qc = q < npxA ? q : npxA - 1;
        const int r = yfv2_fdiv(qc, invW), x = qc - r * W;
        return (r * WP + x + 1) * CP;
      };
      f32x4 bf[KC], bn[KC];
      int t = wave;
      if (t * 16 < npxA) {
        const int o = tile_off(t);
#pragma unroll
        for (int s = 0; s < KC; ++s) bf[s] = (16 * s + 4 * g < CIN) ? *reinterpret_cast<const f32x4*>(T1 + o + 16 * s + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      for (; t * 16 < npxA; t += NW) {
        const int tn = t + NW;
        const int on = tile_off(tn * 16 < npxA ? tn : t);
#pragma unroll
        for (int s = 0; s < KC; ++s) bn[s] = (16 * s + 4 * g < CIN) ? *reinterpret_cast<const f32x4*>(T1 + on + 16 * s + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int q = 16 * t + p;
        const bool valid = q < npxA;
        const int r = yfv2_fdiv(valid ? q : 0, invW);
        const int gy = iy0 + r;
        const bool inimg = valid && gy >= 0 && gy < H;
        float* dst = T1 + tile_off(t);
        f32x4 accA[KC];
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) accA[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
#pragma unroll
        for (int s = 0; s < KC; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < KC; ++mt) accA[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[mt][s][j], bf[s][j], accA[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) {
          const int cb = 16 * mt + 4 * g;
          if (valid && cb < CIN) {
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float u = __builtin_fmaf(accA[mt][c], sc1[mt][c], sh1[mt][c]);
              y[c] = (inimg && u > 0.f) ? u : 0.f;  // input row -1 is the depthwise zero padding
            }
            *reinterpret_cast<f32x4*>(dst + cb) = y;
          }
        }
#pragma unroll
        for (int s = 0; s < KC; ++s) bf[s] = bn[s];
      }
      S2_STAMP(3);
      __syncthreads();  // the main branch's windows reach into the neighbours' tiles
      S2_STAMP(4);
    }

    // ================= main branch: depthwise on pw1's output
    dw_pw_branch(0);
    S2_STAMP(5);
    stage_issue(item + gridDim.x, item + (int)gridDim.x < n_items);  // next item's tile flies across the barrier
    __syncthreads();  // T1 is restaged by the next item
    S2_STAMP(6);
    ++it;
  }
#undef S2_STAMP
}

template <int CIN>
static size_t s2_lds_floats(int R, int W) {
  using Cfg = S2Cfg<CIN>;
  return (size_t)3 * Cfg::W_FL + 2 * Cfg::DW_FL + (size_t)Cfg::NCS * Cfg::KC * 16 + (size_t)(2 * R + 1) * (W + 1) * Cfg::CP + 16;
}

static int yfv2_block_s2_rows_small(int cin, int H, int W) {
  const int OH = H / 2;
  int best = 0;
  for (int r = 1; r <= OH; ++r) {
    const size_t fl = cin == 24 ? s2_lds_floats<24>(r, W) : s2_lds_floats<48>(r, W);
    if (fl * 4 <= 158 * 1024) best = r;
  }
  if (best == 0) return 0;
  const int tiles = (OH + best - 1) / best;
  int R = (OH + tiles - 1) / tiles;  // even split
  // static bounds of the staged kernel (512 threads, 8 waves)
  const int maxt = cin == 24 ? S2Cfg<24>::MAXT : S2Cfg<48>::MAXT;
  while (R > 0 && ((long)(2 * R + 1) * W * (cin / 4) > 14L * 512 || ((2 * R + 1) * W + 15) / 16 > 8 * maxt)) --R;
  return R;
}

template <int CIN>
static void launch_s2(const BlockS2Args& a, hipStream_t s) {
  const size_t lds = s2_lds_floats<CIN>(a.R, a.W) * sizeof(float);
  const int tiles = (a.H / 2 + a.R - 1) / a.R;
  int blocks = a.B * tiles;
  if (blocks > 256) blocks = 256;
  static std::atomic<unsigned long long> lds_ok0{0}, lds_ok1{0}, lds_ok2{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s2_kernel<CIN, 512, false>), lds_ok0);
  if constexpr (CIN == 48) {   // the pair-plane input forms: stage3.0 behind the lane-per-pixel stage 2 (stage2.0 reads the stem's planes in s2px / s2h / front2)
    // (fp32 MFMA whatever the handle's arithmetic: with fp16x3 this block is s3h2_kernel's - the staged form is reached only by the
    // fp32-matrix plan and by stage-2 maps wider than 480 columns, where it is correct at any range.  Its bf16x6 variant is gone.)
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s2_kernel<CIN, 512, true>), lds_ok1);
    if (a.pp_in) { YFV2_LAUNCH((block_s2_kernel<CIN, 512, true>), dim3(blocks), dim3(512), lds, s, a); return; }
  }
  if (a.pp_in) return;   // (never planned: WeightPacker sets pp_in for the 48-channel block only)
  YFV2_LAUNCH((block_s2_kernel<CIN, 512, false>), dim3(blocks), dim3(512), lds, s, a);
}

// ============================================================================
// fused stride-2 block at 96 channels (stage4.0): filters streamed through one LDS slot
// ============================================================================
// Same block as block_s2_kernel (proj on the raw tile, pw1 in place, main on pw1's output), re-cut for 96 channels:
//  * three 96x96 filters (110 KB) do not fit LDS next to a tile: W1 stays resident and ONE 36.8 KB slot holds Wproj
//    while the proj pointwise runs and W2 while the main pointwise runs.  The slot's next content is fetched into
//    registers (five 16-byte loads per thread - L2 hits, every workgroup reads the same two filters) two phases ahead
//    and stored after the barrier that retires the slot's readers;
//  * a row band has only two or three 16-pixel output tiles, so the depthwise is NOT done in MFMA-fragment form by the
//    wave that consumes it (three waves per pixel tile would each repeat it: LDS-bound): thread = (output pixel, channel
//    quad) over the whole band, results to a small tile D, and the pointwise then runs as (pixel tile, pair of
//    output-channel tiles) units with B fragments from D;
//  * pw1 takes whole 16-pixel tiles, the tile's six B fragments in registers before its first write (in place without a
//    barrier); it shares its phase with the proj pointwise.  (Two tiles per wave on four waves with the proj pointwise on
//    their SIMD partners was slower: 70 us against 64 - the pw1 waves then bound the phase.)  pw1 is 70 % of the block's MFMA work and runs as bf16x6
//    (yfv2_internal.h) with W1 PRE-SPLIT on the host (WeightPacker::image_s2w): per (output tile, chunk pair, term) one
//    16-byte operand whose 32 k-slots are the two chunks, so the six products hi.hi hi.mid mid.hi hi.lo lo.hi mid.mid are
//    six MFMAs per chunk pair with no operand duplication and no VALU work on the filter side (splitting W1's 36
//    fragments per tile on the fly cost more VALU time than the fp32 MFMAs it replaced: first version, 85 us).  The two
//    streamed filters stay fp32 (a pre-split slot would not fit) and their pointwise uses the fp32 MFMA.
// Four phases / four barriers per band:
//    P1  slot <- Wproj (from registers) | depthwise(proj): T1 raw -> D          | registers <- W2
//    P2  pw1 in place on T1 | proj pointwise: D x slot -> out[.., 0:96]         | registers <- next band's rows
//    P3  slot <- W2                     | depthwise(main): T1 -> D              | registers <- Wproj
//    P4  T1 <- next band's rows         | main pointwise: D x slot -> out[.., 96:192]
// The input leaves HBM once and pw1's output never does: 279 KB per image instead of 836 KB for the three launches (proj
// tail, pw1, main tail) this replaces.
struct S2WCfg {
  static constexpr int CIN = 96, KC = 6, CP = 100, KS = 96;
  static constexpr int W_FL = KC * KC * 256;
  static constexpr int W1P_FL = KC * (KC / 2) * 3 * 256;  // W1 pre-split: [mt][chunk pair][hi, mid, lo][64 lanes][4 dwords]
  static constexpr int DW_FL = 9 * KS;
  static constexpr int NCS = 10;
  static constexpr int CONST_FL = 2 * DW_FL + NCS * KS;   // WD | WE | CS
  static constexpr int MAXP = 8;                          // staged 16-byte quads per thread (512 threads)
  static constexpr int MPER = 3;                          // output-channel tiles per pointwise unit
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void block_s2w_kernel(BlockS2Args a) {
  using Cfg = S2WCfg;
  constexpr int CIN = Cfg::CIN, KC = Cfg::KC, CP = Cfg::CP, KS = Cfg::KS;
  constexpr int NW = THREADS / 64;
  constexpr int CO = 2 * CIN;
  constexpr int MPER = Cfg::MPER, MG = KC / MPER;
  constexpr int QPP = CIN / 4;
  constexpr int MAXP = Cfg::MAXP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W1 = lds;                  // pre-split (bf16 hi/mid/lo operand quads)
  float* SL = W1 + Cfg::W1P_FL;     // the slot: Wproj / W2 (fp32 fragments)
  float* WD = SL + Cfg::W_FL;       // main depthwise taps [9][KS]
  float* WE = WD + Cfg::DW_FL;      // proj depthwise taps [9][KS]
  float* CS = WE + Cfg::DW_FL;      // [10][KS]
  float* T1 = CS + Cfg::NCS * KS;
  const int H = a.H, W = a.W, R = a.R, OH = H >> 1, OW = W >> 1;
  const float invW = 1.0f / (float)W, invOW = 1.0f / (float)OW;
  const int WP = W + 1;
  const int t1_fl = (2 * R + 1) * WP * CP + 16;
  float* D = T1 + t1_fl;            // depthwise output of the band [R*OW][CP]
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int tiles_per_img = (OH + R - 1) / R;
  const int n_items = a.B * tiles_per_img;

  // global image (WeightPacker::image_s2w): W1 pre-split | W2 | Wproj | WD | WE | CS
  const f32x4* gW2 = reinterpret_cast<const f32x4*>(a.img + Cfg::W1P_FL);
  const f32x4* gWJ = reinterpret_cast<const f32x4*>(a.img + Cfg::W1P_FL + Cfg::W_FL);
  constexpr int WQ4 = Cfg::W_FL / 4;                       // 2304 quads
  constexpr int NWQ = (WQ4 + THREADS - 1) / THREADS;       // 5
  f32x4 wq[NWQ];
  // (the thread index goes through an opaque register in the staging lambdas: per-quad addresses hoisted out of the band
  // loop get spilled around pw1, and a scratch reload next to the global loads waits for every load in flight)
  auto opaque_tid = [&]() { unsigned t = tid; asm volatile("" : "+v"(t)); return t; };
  auto slot_issue = [&](const f32x4* src) {
    const unsigned t = opaque_tid();
#pragma unroll
    for (int k = 0; k < NWQ; ++k) { const unsigned i = t + k * THREADS; wq[k] = i < (unsigned)WQ4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
  };
  auto slot_commit = [&]() {
    const unsigned t = opaque_tid();
#pragma unroll
    for (int k = 0; k < NWQ; ++k) { const unsigned i = t + k * THREADS; if (i < (unsigned)WQ4) reinterpret_cast<f32x4*>(SL)[i] = wq[k]; }
  };

  f32x4 st[MAXP];
  auto stage_issue = [&](int item_, bool active) {  // always (re)defines every staged register
    const int item = active ? item_ : 0;
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R, rows = min(R, OH - y0);
    // the band's 2*rows+1 input rows are whole rows: ONE contiguous run of 16-byte quads in NHWC (row -1 of the first band
    // is the zero padding: its quads are skipped, the pointer below is only dereferenced past them)
    const int nq = active ? (2 * rows + 1) * W * QPP : 0;
    const int skip = y0 == 0 ? W * QPP : 0;
    const float* src = a.in + ((long long)b * H * W + (long long)(2 * y0 - 1) * W) * CIN;
    const unsigned t = opaque_tid();
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const unsigned i = t + j * THREADS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < (unsigned)nq && i >= (unsigned)skip) v = *reinterpret_cast<const f32x4*>(src + 4u * i);
      st[j] = v;
    }
  };
  auto stage_commit = [&](int item) {   // rows above the image arrive as zeros; column 0 of T1 (input col -1) is never written
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R, rows = min(R, OH - y0);
    const int nq = (2 * rows + 1) * W * QPP;
    const int t = (int)opaque_tid();
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
      const int i = t + j * THREADS;
      if (i >= nq) continue;
      const int pix = i / QPP, q = i - pix * QPP;
      const int r = yfv2_fdiv(pix, invW), x = pix - r * W;
      *reinterpret_cast<f32x4*>(T1 + (r * WP + x + 1) * CP + 4 * q) = st[j];
    }
  };
  stage_issue(blockIdx.x, (int)blockIdx.x < n_items);
  YFV2_WSTAMP(0);

  // prologue: W1 -> W1, Wproj -> slot, taps + BN constants behind them; every load issued before the first store
  {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(a.img);
    constexpr int P4 = Cfg::W1P_FL / 4;                    // 3456 quads
    constexpr int NP = (P4 + THREADS - 1) / THREADS;       // 7
    constexpr int C4 = Cfg::CONST_FL / 4;                  // 672 quads
    constexpr int NC = (C4 + THREADS - 1) / THREADS;       // 2
    f32x4 t1[NP], t2[NWQ], t3[NC];
#pragma unroll
    for (int k = 0; k < NP; ++k) { const int i = tid + k * THREADS; t1[k] = i < P4 ? g4[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NWQ; ++k) { const int i = tid + k * THREADS; t2[k] = i < WQ4 ? gWJ[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NC; ++k) { const int i = tid + k * THREADS; t3[k] = i < C4 ? g4[P4 + 2 * WQ4 + i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NP; ++k) { const int i = tid + k * THREADS; if (i < P4) reinterpret_cast<f32x4*>(W1)[i] = t1[k]; }
#pragma unroll
    for (int k = 0; k < NWQ; ++k) { const int i = tid + k * THREADS; if (i < WQ4) reinterpret_cast<f32x4*>(SL)[i] = t2[k]; }
#pragma unroll
    for (int k = 0; k < NC; ++k) { const int i = tid + k * THREADS; if (i < C4) reinterpret_cast<f32x4*>(WD)[i] = t3[k]; }
  }
  for (int i = tid; i < t1_fl / 4; i += THREADS) reinterpret_cast<f32x4*>(T1)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};  // column 0 stays zero
  __syncthreads();
  if ((int)blockIdx.x < n_items) stage_commit(blockIdx.x);
  __syncthreads();
  YFV2_WSTAMP(1);
  int it = 0;   // stamps 2.. of the workgroup's SECOND band (steady state)
#define S2W_STAMP(k) do { if (it == 1) YFV2_WSTAMP(2 + (k)); } while (0)

  // ---- depthwise 3x3 s2 (+BN) of the band: thread = (output pixel, channel quad), T1 -> D
  auto depthwise = [&](const float* taps, const float* dsc_p, const float* dsh_p, int npxB) {
    for (int i = tid; i < npxB * QPP; i += THREADS) {
      const int q = i / QPP, cq = i - q * QPP;
      const int r = yfv2_fdiv(q, invOW), x = q - r * OW;
      const float* tp = T1 + ((2 * r) * WP + 2 * x) * CP + 4 * cq;   // window rows 2r..2r+2, T1 cols 2x..2x+2
      f32x4 win[9], wk[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        win[k] = *reinterpret_cast<const f32x4*>(tp + ((k / 3) * WP + (k % 3)) * CP);
        wk[k] = *reinterpret_cast<const f32x4*>(taps + k * KS + 4 * cq);
      }
      const f32x4 dsc = *reinterpret_cast<const f32x4*>(dsc_p + 4 * cq);
      const f32x4 dsh = *reinterpret_cast<const f32x4*>(dsh_p + 4 * cq);
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = __builtin_fmaf(win[k][c], wk[k][c], d[c]);
      f32x4 y;
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = __builtin_fmaf(d[c], dsc[c], dsh[c]);
      *reinterpret_cast<f32x4*>(D + q * CP + 4 * cq) = y;
    }
  };
  // ---- pointwise (+BN+ReLU) on D with the slot's filter (fp32 MFMA); unit = (16 output pixels, MPER output-channel tiles).
  // The unit's 18 filter fragments are requested before the first MFMA.
  // `after_pw1`: the phase ran pw1 first (tile t on wave t % NW; SIMD = wave & 3 holds waves s and s + 4): the wave pairs
  // with the fewest tiles take the units - for the seven tiles of a 5 x 22 band wave 7 (no tile) takes units 0 and 1,
  // waves 0 and 1 units 2 and 3, which leaves every SIMD with three pieces of work but one.
  auto pointwise = [&](const float* psc_p, const float* psh_p, float* out_base, int y0, int npxB, bool after_pw1) {
    const int ntB = (npxB + 15) >> 4;
    for (int u = 0; u < ntB * MG; ++u) {
      const int owner = !after_pw1 ? u % NW : (u < 2 ? NW - 1 : (u - 2) % NW);
      if (owner != wave) continue;
      const int t = u / MG, mg = u - t * MG;
      const int q = 16 * t + p;
      const bool pv = q < npxB;
      const int qc = pv ? q : npxB - 1;
      f32x4 bf[KC], af[MPER][KC];
#pragma unroll
      for (int s = 0; s < KC; ++s) bf[s] = *reinterpret_cast<const f32x4*>(D + qc * CP + 16 * s + 4 * g);
#pragma unroll
      for (int s = 0; s < KC; ++s)
#pragma unroll
        for (int m = 0; m < MPER; ++m) af[m][s] = *reinterpret_cast<const f32x4*>(SL + (((MPER * mg + m) * KC + s) * 64 + lane) * 4);
      f32x4 acc[MPER];
#pragma unroll
      for (int m = 0; m < MPER; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KC; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < MPER; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s][j], bf[s][j], acc[m], 0, 0, 0);
      const int r = yfv2_fdiv(qc, invOW), x = qc - r * OW;
      float* dst = out_base + ((size_t)(y0 + r) * OW + x) * CO;
#pragma unroll
      for (int m = 0; m < MPER; ++m) {
        const int cb = 16 * (MPER * mg + m) + 4 * g;
        if (pv) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(psc_p + cb);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(psh_p + cb);
          f32x4 y;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float v = __builtin_fmaf(acc[m][k], sc[k], sh[k]);
            y[k] = v > 0.f ? v : 0.f;
          }
          *reinterpret_cast<f32x4*>(dst + cb) = y;
        }
      }
    }
  };

  bool slot_pending = false;   // registers hold the next Wproj (every band but the workgroup's first: the prologue loaded it)
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / tiles_per_img, ti = item - b * tiles_per_img;
    const int y0 = ti * R;                 // first output row
    const int rows = min(R, OH - y0);
    const int iy0 = 2 * y0 - 1;            // first input row of T1
    const int npxA = (2 * rows + 1) * W;
    const int npxB = rows * OW;
    float* out_img = a.out + (size_t)b * OH * OW * CO;

    // ================= P1
    if (slot_pending) slot_commit();
    slot_issue(gW2);
    S2W_STAMP(0);
    depthwise(WE, CS + 6 * KS, CS + 7 * KS, npxB);
    S2W_STAMP(1);
    __syncthreads();
    S2W_STAMP(2);

    // ================= P2: pw1 in place, one 16-pixel tile per wave and pass (tile t on wave t % NW), then the proj pointwise
    {
      auto tile_off = [&](int t) {
        const int q = 16 * t + p;
        const int qc = q < npxA ? q : npxA - 1;
        const int r = yfv2_fdiv(qc, invW), x = qc - r * W;
        return (r * WP + x + 1) * CP;
      };
      for (int t = wave; t * 16 < npxA; t += NW) {
        const int o = tile_off(t);
        f32x4 bf[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) bf[s] = *reinterpret_cast<const f32x4*>(T1 + o + 16 * s + 4 * g);
        const int q = 16 * t + p;
        const bool valid = q < npxA;
        const int r = yfv2_fdiv(valid ? q : 0, invW);
        const int gy = iy0 + r;
        const bool inimg = valid && gy >= 0 && gy < H;
        f32x4 accA[KC];
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) accA[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        S2W_STAMP(3);
        // six stages = (chunk pair sp, half of the output tiles): a stage's nine operand quads are requested one stage ahead
        // of the 18 MFMAs that use them
        constexpr int HM = KC / 2;
        yfv2_bf16x8 aop[2][3][HM];     // [buffer][hi, mid, lo][mt within the half]
        auto load_stage = [&](int k, yfv2_bf16x8 (&dst)[3][HM]) {
          const int sp = k >> 1, m0 = (k & 1) * HM;
#pragma unroll
          for (int m = 0; m < HM; ++m) {
            const float* wq3 = W1 + ((((m0 + m) * (KC / 2) + sp) * 3) * 64 + lane) * 4;
#pragma unroll
            for (int term = 0; term < 3; ++term) dst[term][m] = __builtin_bit_cast(yfv2_bf16x8, *reinterpret_cast<const u32x4*>(wq3 + 256 * term));
          }
        };
        load_stage(0, aop[0]);
        yfv2_bf16x8 bh, bm, bl;
#pragma unroll
        for (int k = 0; k < 2 * (KC / 2); ++k) {
          const int sp = k >> 1, m0 = (k & 1) * HM;
          if (k + 1 < 2 * (KC / 2)) load_stage(k + 1, aop[(k + 1) & 1]);
          if ((k & 1) == 0) {
            // B side: split the pair's two chunks into bf16 terms; operand = {chunk 2sp: 4 k-slots, chunk 2sp+1: 4 k-slots}
            unsigned h0[2], m0_[2], l0[2], h1[2], m1[2], l1[2];
            yfv2_split3(bf[2 * sp], h0, m0_, l0);
            yfv2_split3(bf[2 * sp + 1], h1, m1, l1);
            bh = __builtin_bit_cast(yfv2_bf16x8, (u32x4){h0[0], h0[1], h1[0], h1[1]});
            bm = __builtin_bit_cast(yfv2_bf16x8, (u32x4){m0_[0], m0_[1], m1[0], m1[1]});
            bl = __builtin_bit_cast(yfv2_bf16x8, (u32x4){l0[0], l0[1], l1[0], l1[1]});
          }
          // six products, the small ones first; mt innermost: no MFMA waits for the one before it
#define S2W_PROD(T_, B_) _Pragma("unroll") for (int m = 0; m < HM; ++m) accA[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aop[k & 1][T_][m], B_, accA[m0 + m], 0, 0, 0);
          S2W_PROD(2, bh) S2W_PROD(0, bl) S2W_PROD(1, bm) S2W_PROD(1, bh) S2W_PROD(0, bm) S2W_PROD(0, bh)
#undef S2W_PROD
          __builtin_amdgcn_sched_barrier(0);  // keeps the schedule from hoisting every stage's loads (spills)
          if (k & 1) S2W_STAMP(4 + sp);
        }
        float* dst = T1 + o;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) {
          const int cb = 16 * mt + 4 * g;
          if (valid) {
            const f32x4 sc1 = *reinterpret_cast<const f32x4*>(CS + 0 * KS + cb);
            const f32x4 sh1 = *reinterpret_cast<const f32x4*>(CS + 1 * KS + cb);
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float v = __builtin_fmaf(accA[mt][c], sc1[c], sh1[c]);
              y[c] = (inimg && v > 0.f) ? v : 0.f;  // input row -1 is the depthwise zero padding
            }
            *reinterpret_cast<f32x4*>(dst + cb) = y;
          }
        }
      }
    }
    S2W_STAMP(7);
    stage_issue(item + gridDim.x, item + (int)gridDim.x < n_items);   // the next band's rows fly until P4 (issued after pw1: registers)
    S2W_STAMP(8);
    pointwise(CS + 8 * KS, CS + 9 * KS, out_img, y0, npxB, true);
    S2W_STAMP(9);
    __syncthreads();
    S2W_STAMP(10);

    // ================= P3
    slot_commit();                         // Wproj's readers are done: the slot becomes W2
    slot_issue(gWJ);
    slot_pending = true;
    S2W_STAMP(11);
    depthwise(WD, CS + 2 * KS, CS + 3 * KS, npxB);
    S2W_STAMP(12);
    __syncthreads();
    S2W_STAMP(13);

    // ================= P4
    if (item + (int)gridDim.x < n_items) stage_commit(item + gridDim.x);
    S2W_STAMP(14);
    pointwise(CS + 4 * KS, CS + 5 * KS, out_img + CIN, y0, npxB, false);
    S2W_STAMP(15);
    __syncthreads();
    S2W_STAMP(16);
    ++it;
  }
#undef S2W_STAMP
}

static size_t s2w_lds_floats(int R, int W) {
  return (size_t)S2WCfg::W1P_FL + S2WCfg::W_FL + S2WCfg::CONST_FL + (size_t)(2 * R + 1) * (W + 1) * S2WCfg::CP + 16 + (size_t)R * (W / 2) * S2WCfg::CP;
}

// rows per work item of the 96-channel kernel: the largest band that fits LDS and the staging registers and keeps pw1
// at one tile per wave (<= 128 pixels) when any band does; 0 = not supported
static int s2w_rows(int H, int W) {
  if ((H & 1) || (W & 1)) return 0;
  const int OH = H / 2;
  int best = 0, best128 = 0;
  for (int r = 1; r <= OH; ++r) {
    if (s2w_lds_floats(r, W) * 4 > 158 * 1024) break;
    if ((long)(2 * r + 1) * W * (S2WCfg::CIN / 4) > (long)S2WCfg::MAXP * 512) break;
    best = r;
    if ((2 * r + 1) * W <= 128) best128 = r;
  }
  return best128 ? best128 : best;
}

int yfv2_block_s2_rows(int cin, int H, int W) {
  if (cin == 96) return s2w_rows(H, W);
  return yfv2_block_s2_rows_small(cin, H, W);
}

static void launch_s2w(const BlockS2Args& a, hipStream_t s) {
  const size_t lds = s2w_lds_floats(a.R, a.W) * sizeof(float);
  const int tiles = (a.H / 2 + a.R - 1) / a.R;
  int blocks = a.B * tiles;
  if (blocks > 256) blocks = 256;
  static std::atomic<unsigned long long> lds_ok{0};
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&block_s2w_kernel<512>), lds_ok);
  YFV2_LAUNCH((block_s2w_kernel<512>), dim3(blocks), dim3(512), lds, s, a);
}

bool yfv2_launch_block_s2(int cin, const BlockS2Args& a, hipStream_t s) {
  if (cin == 24) { launch_s2<24>(a, s); return true; }
  if (cin == 48) { launch_s2<48>(a, s); return true; }
  if (cin == 96 && !a.pp_in && a.bf6) { launch_s2w(a, s); return true; }   // planned only with bf16x6 on (image_s2w)
  return false;
}

// ============================================================================
// tower half, version 2: one image per workgroup, channel-chunk loop outermost
// ============================================================================
// Reference (model/fpn.py:12-25): DWConvblock = [dw5x5+BN+ReLU -> pw72+BN] x 2, and
// model/detector.py:25-31 applies a biased 1x1 output conv to the block's result.  One launch =
// one half: dw5x5 (pad 2) + BN + ReLU -> pw 72->72 + BN [-> output conv + bias, chained in
// registers: the BN'd accumulator tile t IS the B fragment of chunk t of the output conv].
// A wave owns NT fixed 16-pixel tiles of the image and
// keeps their pointwise accumulators in registers while the workgroup walks the five
// 16-channel chunks: per chunk only that channel slice of the (zero-haloed) input image
// is in LDS (four 26-row x 38-slot quad planes at 22x22, see below, instead of a 10-row x 72-channel tile), staged with a
// one-chunk-ahead register prefetch.  No halo re-staging, every wave busy, and the
// accumulators never leave registers until the (chained) output conv is done.
// The staged slice is kept as four PLANES, one per channel quad: [quad][row][W + 16][4 floats], each plane padded to
// a multiple of 256 bytes.  ds_read_b128 is serviced in four 16-lane groups that mix two quads ({0-3,12-15,20-27}, ...):
// with the quad offset a multiple of 256 bytes the 16 lanes of a group read 16 consecutive 16-byte slots (a row pitch
// of W + 16 slots keeps a 16-pixel run consecutive mod 16 across a row wrap) - no bank conflicts; the pixel-major
// [pixel][20] tile this replaces had three colliding slots in every group (SQ_LDS_BANK_CONFLICT = 50 % of LDS-active).
__host__ __device__ constexpr int tw2_row_pitch(int W) { return W + 16; }                                        // 16-byte slots
__host__ __device__ constexpr int tw2_plane_slots(int H, int W) { return ((H + 4) * tw2_row_pitch(W) + 15) & ~15; }

template <int MH, int THREADS, int NT, int NPF, bool BF6>
__global__ __launch_bounds__(THREADS) void tower2_kernel(TowerArgs a) {
  constexpr int KC = TW_KC, C = TW_C;
  constexpr int NW = THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* WP_ = lds;
  float* WH = WP_ + TW_WP_FL;
  float* WD = WH + MH * TW_WH_FL;
  float* CS = WD + 25 * KC * 16;
  float* TIN = CS + 5 * 96;
  const int H = a.H, W = a.W, HW = H * W;
  const float invW = 1.0f / (float)W;
  const int RP = tw2_row_pitch(W), PL = tw2_plane_slots(H, W);   // row pitch / plane size in 16-byte slots
  const int tin_fl = 4 * PL * 4;
  const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, wave = tid >> 6;
  YFV2_WSTAMP(0);

  // prologue: the LDS image (filters, taps, BN constants - padded and zero-filled on the host,
  // yfv2_load_weights) is one straight coalesced 16-byte copy
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.img);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    constexpr int N4 = (TW_WP_FL + MH * TW_WH_FL + 25 * KC * 16 + 5 * 96) / 4;
    constexpr int NIT = (N4 + THREADS - 1) / THREADS;  // <= 11: every load is issued before the first store
    f32x4 tmp[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; tmp[k] = i < N4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int k = 0; k < NIT; ++k) { const int i = tid + k * THREADS; if (i < N4) dst[i] = tmp[k]; }
  }
  for (int i = tid; i < tin_fl / 4; i += THREADS) reinterpret_cast<f32x4*>(TIN)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the 2-pixel halo stays zero
  __syncthreads();
  YFV2_WSTAMP(1);

  // this wave's pixel tiles (fixed for every image)
  int base[NT], opix[NT];
  bool pv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int q = 16 * (wave * NT + nt) + p;
    pv[nt] = q < HW;
    const int qc = pv[nt] ? q : HW - 1;
    const int r = yfv2_fdiv(qc, invW), x = qc - r * W;
    base[nt] = (r * RP + x) * 4;  // top-left of the 5x5 window (float offset inside a plane)
    opix[nt] = qc;
  }
  // staging slots of this thread: float4 i -> (pixel, quad) of the chunk slice.  Eight consecutive lanes take eight
  // consecutive pixels of ONE quad (ds_write_b128 is serviced in contiguous 8-lane groups: eight consecutive slots of a
  // plane), the four quads of those pixels go to the next three 8-lane groups - the wave still reads whole 64-byte
  // channel runs of 16 pixels from global memory.
  int s_src[NPF], s_dst[NPF];
  const int my_c4 = (tid >> 3) & 3;          // THREADS is a multiple of 32: the same quad for every j
#pragma unroll
  for (int j = 0; j < NPF; ++j) {
    const int i = tid + j * THREADS;
    const int px = (i & 7) + 8 * (i >> 5);
    const bool ok = px < HW;
    const int y = ok ? yfv2_fdiv(px, invW) : 0, x = ok ? px - y * W : 0;
    s_src[j] = ok ? px * C + 4 * my_c4 : -1;
    s_dst[j] = (my_c4 * PL + (y + 2) * RP + x + 2) * 4;
  }
  auto stage_load = [&](const float* img, int s, f32x4 (&pre)[NPF]) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      pre[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (s_src[j] >= 0 && 16 * s + 4 * my_c4 < C)
        pre[j] = *reinterpret_cast<const f32x4*>(img + s_src[j] + 16 * s);
    }
  };

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float* img = a.in + (size_t)b * HW * C;
    f32x4 acc[KC][NT];
#pragma unroll
    for (int mt = 0; mt < KC; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 pre[NPF];
    stage_load(img, 0, pre);
#pragma unroll 1
    for (int s = 0; s < KC; ++s) {
      __syncthreads();  // the previous chunk's (or image's) readers are done with TIN
#pragma unroll
      for (int j = 0; j < NPF; ++j)
        if (s_src[j] >= 0) *reinterpret_cast<f32x4*>(TIN + s_dst[j]) = pre[j];
      if (s + 1 < KC) stage_load(img, s + 1, pre);  // flies during this chunk's compute
      __syncthreads();
      YFV2_WSTAMP(2 + 3 * s);
      const int cb = 16 * s + 4 * g;
      f32x4 d[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) d[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ky = 0; ky < 5; ++ky) {
        const float* wrow = WD + ky * 5 * KC * 16 + cb;
        const float* trow = TIN + (g * PL + ky * RP) * 4;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + kx * KC * 16);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(trow + base[nt] + kx * 4);
            if constexpr (BF6) {
              d[nt] = __builtin_elementwise_fma(v, w, d[nt]);   // two v_pk_fma_f32: with the pointwise on the bf16 pipe the fp32 datapath is the depthwise's alone
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k) d[nt][k] = __builtin_fmaf(v[k], w[k], d[nt][k]);
            }
          }
        }
      }
      const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 0 * 96 + cb);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 1 * 96 + cb);
      f32x4 bfr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = __builtin_fmaf(d[nt][k], sc[k], sh[k]);  // channels >= 72: sc = sh = 0 -> 0
          bfr[nt][k] = (cb < C && u > 0.f) ? u : 0.f;
        }
      if constexpr (BF6) {
        // bf16x6 (yfv2_internal.h): the depthwise result is split once per pixel tile, each filter fragment once per
        // output-channel tile; three MFMAs per (mt, nt) - the NT accumulators of one mt are independent of each other
        Bf3B b3[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b3[nt] = yfv2_split_b(bfr[nt]);
        YFV2_WSTAMP(3 + 3 * s);
#pragma unroll
        for (int mt = 0; mt < KC; ++mt) {
          const Bf3A a3 = yfv2_split_a(*reinterpret_cast<const f32x4*>(WP_ + ((mt * KC + s) * 64 + lane) * 4));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = yfv2_mfma6_step<0>(a3, b3[nt], acc[mt][nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = yfv2_mfma6_step<1>(a3, b3[nt], acc[mt][nt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = yfv2_mfma6_step<2>(a3, b3[nt], acc[mt][nt]);
        }
      } else {
      f32x4 afP[KC];
#pragma unroll
      for (int mt = 0; mt < KC; ++mt) afP[mt] = *reinterpret_cast<const f32x4*>(WP_ + ((mt * KC + s) * 64 + lane) * 4);
      __builtin_amdgcn_sched_barrier(0);
      YFV2_WSTAMP(3 + 3 * s);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < KC; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afP[mt][j], bfr[nt][j], acc[mt][nt], 0, 0, 0);
      }
      YFV2_WSTAMP(4 + 3 * s);
    }
    // pointwise BN (no ReLU: fpn.py:16-17,23-24)
#pragma unroll
    for (int mt = 0; mt < KC; ++mt) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(CS + 2 * 96 + 16 * mt + 4 * g);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(CS + 3 * 96 + 16 * mt + 4 * g);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[mt][nt][k] = __builtin_fmaf(acc[mt][nt][k], sc[k], sh[k]);
    }
    YFV2_WSTAMP(17);
    if constexpr (MH == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (!pv[nt]) continue;
        float* dst = a.out + ((size_t)b * HW + opix[nt]) * C;
#pragma unroll
        for (int mt = 0; mt < KC; ++mt)
          if (16 * mt + 4 * g < C) *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * g) = acc[mt][nt];
      }
    } else {
      // chained output conv: the BN'd accumulator tile s IS the B fragment of chunk s.  Output-channel tiles in pairs, so
      // that (bf16x6) one split of a B fragment feeds two tiles' MFMAs
      constexpr int MP = MH >= 2 ? 2 : 1;
#pragma unroll 1
      for (int m0 = 0; m0 < MH; m0 += MP) {
        f32x4 hacc[MP][NT];
#pragma unroll
        for (int q = 0; q < MP; ++q)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) hacc[q][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF6) {
#pragma unroll
          for (int s = 0; s < KC; ++s) {
            Bf3A a3[MP];
#pragma unroll
            for (int q = 0; q < MP; ++q) a3[q] = yfv2_split_a(*reinterpret_cast<const f32x4*>(WH + (((m0 + q) * KC + s) * 64 + lane) * 4));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const Bf3B b3 = yfv2_split_b(acc[s][nt]);
#pragma unroll
              for (int q = 0; q < MP; ++q) hacc[q][nt] = yfv2_mfma6_step<0>(a3[q], b3, hacc[q][nt]);
#pragma unroll
              for (int q = 0; q < MP; ++q) hacc[q][nt] = yfv2_mfma6_step<1>(a3[q], b3, hacc[q][nt]);
#pragma unroll
              for (int q = 0; q < MP; ++q) hacc[q][nt] = yfv2_mfma6_step<2>(a3[q], b3, hacc[q][nt]);
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < MP; ++q) {
            f32x4 afH[KC];
#pragma unroll
            for (int s = 0; s < KC; ++s) afH[s] = *reinterpret_cast<const f32x4*>(WH + (((m0 + q) * KC + s) * 64 + lane) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KC; ++s)
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  hacc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afH[s][j], acc[s][nt][j], hacc[q][nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < MP; ++q) {
          const int m = m0 + q;
          const f32x4 bias = *reinterpret_cast<const f32x4*>(CS + 4 * 96 + 16 * m + 4 * g);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (!pv[nt]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = 16 * m + 4 * g + r;
              if (co < a.mh) {
                const float y = hacc[q][nt][r] + bias[r];
                if (co < a.split)
                  a.nchw0[((size_t)b * a.split + co) * HW + opix[nt]] = y;
                else
                  a.nchw1[((size_t)b * (a.mh - a.split) + (co - a.split)) * HW + opix[nt]] = y;
              }
            }
          }
        }
      }
    }
    YFV2_WSTAMP(18);
  }
}

template <int MH, int THREADS, int NT, int NPF>
static void launch_tower2(const TowerArgs& a, hipStream_t s) {
  const size_t lds = sizeof(float) * ((size_t)TW_WP_FL + (size_t)MH * TW_WH_FL + 25 * TW_KC * 16 + 5 * 96 +
                                      (size_t)16 * tw2_plane_slots(a.H, a.W));
  int blocks = a.B < 256 ? a.B : 256;
  static std::atomic<unsigned long long> lds_ok0{0}, lds_ok1{0};
  if (a.bf6) {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&tower2_kernel<MH, THREADS, NT, NPF, true>), lds_ok1);
    YFV2_LAUNCH((tower2_kernel<MH, THREADS, NT, NPF, true>), dim3(blocks), dim3(THREADS), lds, s, a);
    return;
  }
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&tower2_kernel<MH, THREADS, NT, NPF, false>), lds_ok0);
  YFV2_LAUNCH((tower2_kernel<MH, THREADS, NT, NPF, false>), dim3(blocks), dim3(THREADS), lds, s, a);
}

bool yfv2_tower2_supported(int H, int W) { return H * W <= 16 * 4 * 8; }

// whole-image variant: needs H*W <= 16 * NT * waves and the staged slice to fit the thread grid
bool yfv2_launch_tower2(const TowerArgs& a, hipStream_t s) {
  const int mh_tiles = a.has_head ? (a.mh + 15) / 16 : 0;
  const int hw = a.H * a.W;
  if (hw <= 16 * 4 * 8 && hw * 4 <= 4 * 512) {           // up to 22x22: 512 threads, 4 tiles per wave
    if (hw > 16 * 1 * 8) {
      // (a 16-wave x 2-tile variant at <= 128 VGPRs measured 5-8 % slower than 8 waves x 4 tiles)
      if (mh_tiles == 0) { launch_tower2<0, 512, 4, 4>(a, s); return true; }
      if (mh_tiles == 1) { launch_tower2<1, 512, 4, 4>(a, s); return true; }
      if (mh_tiles <= 6) { launch_tower2<6, 512, 4, 4>(a, s); return true; }
    } else {                                              // up to 11x11: 512 threads, 1 tile per wave
      if (mh_tiles == 0) { launch_tower2<0, 512, 1, 1>(a, s); return true; }
      if (mh_tiles == 1) { launch_tower2<1, 512, 1, 1>(a, s); return true; }
      if (mh_tiles <= 6) { launch_tower2<6, 512, 1, 1>(a, s); return true; }
    }
  }
  return false;
}
