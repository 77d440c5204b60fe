// yfv2_conv.hip - gfx950 (CDNA4, wave64) kernels of the Yolo-FastestV2 forward:
//   (the stem lives in yfv2_stem.hip)
//   pw_kernel   : every pointwise 1x1 conv (+BN, +ReLU) on v_mfma_f32_16x16x4_f32,
//                 with channel-shuffle / concat / nearest-upsample / NCHW-head
//                 folded into its operand loads and stores
//   dw_kernel   : depthwise 3x3 / 5x5 (+BN, +ReLU), NHWC float4 over channels
//
// Reference layers (read for behaviour only): model/backbone/shufflenetv2.py:19-63,
// 74-80; model/fpn.py:12-25,35-43,51-64; model/detector.py:17-31.
//
// Data layout: all internal activations are NHWC fp32 (pixel-major, channels
// contiguous) so that a 16-lane group reads one pixel's channel quad per lane
// and stores are 16 B per lane; the API surface stays NCHW (input image and the
// six logit maps).
#include <cstdlib>

#include "yfv2_internal.h"

typedef _Float16 yfv2_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 yfv2_h8 __attribute__((ext_vector_type(8)));
typedef unsigned yfv2_u2 __attribute__((ext_vector_type(2)));

// ============================================================================
// pointwise 1x1 conv on the fp32 matrix cores
// ============================================================================
// GEMM view per wave:  D[co][pixel] = sum_ci W[co][ci] * X[pixel][ci]
//   A operand = W   (M = output channels, 16 per tile)  - from LDS, staged once per block
//   B operand = X^T (N = pixels, 16 per tile)           - straight from global NHWC
// v_mfma_f32_16x16x4_f32 fragment maps (wave64): A lane l holds A[i=l&15][k=l>>4],
// B lane l holds B[k=l>>4][j=l&15], D lane l reg r holds D[i=4*(l>>4)+r][j=l&15].
// K order inside a 16-channel chunk is permuted so that every lane fetches its
// four k-values as one 16-byte load: MFMA step j of chunk s consumes channel
// 16*s + 4*(l>>4) + j from BOTH operands (any bijection of K is legal as long as
// A and B agree).  D gives each lane 4 consecutive output channels of one pixel
// -> one 16-byte NHWC store per tile.
//
// Modes fold the reference's data-movement ops into the operand traffic:
//   PW_SHUFFLE  channel_shuffle (shufflenetv2.py:57-63): B reads the ODD input
//               channels; the EVEN ones (the pass-through branch) are written
//               to out[copy_off + j] by the same lanes - shuffle + cat, no copy kernel
//   PW_FPN      F.interpolate(x2, nearest) + torch.cat (fpn.py:57-58): channels
//               [0,192) gathered from C3 at (y/2, x/2), [192,288) from C2
//   PW_HEAD     the three biased output convs (detector.py:17-19,25-31): stores
//               NCHW logits into two destination tensors split at `split`
// PRE (the streamed large-K forms: fpn.conv1x1_2 / conv1x1_3): fp16x3 (yfv2_stem16.hip) - the filter arrives from the host
// as two fp16 terms x 2^sw (WeightPacker::image_pw), per (output tile, PAIR of chunks, term) one 16-byte operand whose 32
// k-slots are the two chunks; the activations are scaled by 2^4 and split into two fp16 terms per chunk pair; w1 x2 + w2 x1
// + w1 x1 = three v_mfma_f32_16x16x32_f16 per (output tile, pixel tile, chunk pair), the scales undone exactly inside the BN
// scale.  (Round 2's form of this path was bf16x6 with a host-split hi / mid / lo filter: six MFMAs per pair - 13.6 of the
// 36 us of conv1x1_2 were matrix-core time - 135 KB of LDS and a three-term split per activation quad.)  |x| < 4094.
template <int K, int MT, int NT, int MODE, int THREADS = 256, bool STREAM = false, bool BF6 = false, bool PRE = false>
__global__ __launch_bounds__(THREADS) void pw_kernel(PwArgs a) {
  constexpr int K16 = K / 16;
  constexpr int KT = K % 16;
  static_assert(KT == 0 || KT == 8, "K must be 16*n or 16*n+8");
  static_assert(!PRE || (STREAM && BF6 && K % 32 == 0), "PRE: streamed bf16x6 form, whole chunk pairs");
  // filter in LDS fragment-major (host-packed, WeightPacker::image_pw): frag (mt, s), lane l holds
  // W[16mt + (l&15)][16s + 4(l>>4) .. +3] - the 64 lanes of a fragment read touch 64 consecutive 16-byte slots, no bank
  // conflicts whatever 16-lane groups the hardware forms (a row-padded [M][K+4] image collides 5-7 slots per group);
  // an 8-channel tail is MT fragments of 8 bytes per lane behind them
  constexpr int FRAG_FL = MT * K16 * 256;   // (PRE: [mt][chunk pair][2 terms][256] - the same size)
  constexpr int FILT_FL = FRAG_FL + (KT ? MT * 128 : 0);
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int tid = threadIdx.x;

  // PW_DUAL: two convs on the same input in one launch - output tiles 0 .. MT/2-1 -> out (ReLU as asked), tiles MT/2 .. MT-1 -> copy (no
  // ReLU).  A wave loads and splits its pixels ONCE for both (as two interleaved grids of MT/2-tile workgroups: 17.6 us against 13.2 for
  // the one conv; this form: the operand traffic, the K loop's round trips and the prologue are shared)
  static_assert(MODE != PW_DUAL || MT % 2 == 0, "two convs of MT / 2 tiles each");
  const float* const img = a.img;
  {  // prologue: one coalesced copy of the image, up to nine 16-byte loads per thread in flight at a time (the plain
     // load -> wait -> store loop the compiler makes of `dst[i] = src[i]` is one global round trip per 16 bytes and thread:
     // 17 of them for the 135 KB pre-split K = 288 filter)
    const f32x4* src = reinterpret_cast<const f32x4*>(img);
    f32x4* dst = reinterpret_cast<f32x4*>(wl);
    constexpr int N4 = FILT_FL / 4;
    constexpr int PER = (N4 + THREADS - 1) / THREADS;            // THREADS == blockDim.x (pw_launch)
    constexpr int BATCH = PER < 9 ? PER : 9;
#pragma unroll 1
    for (int k0 = 0; k0 < PER; k0 += BATCH) {
      f32x4 t[BATCH];
#pragma unroll
      for (int k = 0; k < BATCH; ++k) { const int i = tid + (k0 + k) * THREADS; t[k] = i < N4 ? src[i] : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) { const int i = tid + (k0 + k) * THREADS; if (i < N4) dst[i] = t[k]; }
    }
  }
  __syncthreads();

  const int lane = tid & 63, p = lane & 15, g = lane >> 4;
  const int wave = tid >> 6, nwaves = blockDim.x >> 6;

  f32x4 sc[MT], sh[MT];  // padded to MT*16 on the host: unconditional 16-byte loads
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    sc[mt] = *reinterpret_cast<const f32x4*>(img + FILT_FL + 16 * mt + 4 * g);
    sh[mt] = *reinterpret_cast<const f32x4*>(img + FILT_FL + MT * 16 + 16 * mt + 4 * g);
  }

  Yfv2Watch watch;   // range guard of the fp16x3 form (yfv2_internal.h)
  const int n_super = (a.P + NT * 16 - 1) / (NT * 16);
  for (int st = blockIdx.x * nwaves + wave; st < n_super; st += gridDim.x * nwaves) {
    const int pix0 = st * (NT * 16);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pixv[NT];
    f32x4 qv[MODE == PW_FPNQ ? MT : 1];

    if constexpr (STREAM) {
      // ---- large K (192 / 288): stream the 16-channel chunks with a one-chunk-ahead
      // prefetch instead of holding K/4 registers per pixel tile
      static_assert(!STREAM || (KT == 0 && (MODE == PW_PLAIN || MODE == PW_FPN || MODE == PW_DUAL || MODE == PW_FPNQ)), "STREAM: K % 16 == 0, plain/fpn only");
      const float* src0[NT];
      const float* src1[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int pix = pix0 + nt * 16 + p;
        pixv[nt] = pix;
        const int pc = pix < a.P ? pix : a.P - 1;
        if constexpr (MODE == PW_FPN) {
          const int hw = a.H * a.W;
          const int b = pc / hw, rem = pc - b * hw;
          const int y = rem / a.W, x = rem - y * a.W;
          src0[nt] = a.in + ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * 192 + 4 * g;  // chunks 0..11
          src1[nt] = a.in2 + (size_t)pc * 96 + 4 * g - 16 * 12;                                        // chunks 12..17
        } else {
          src0[nt] = a.in + (size_t)pc * a.in_stride + a.in_off + 4 * g;
          src1[nt] = src0[nt];
        }
      }
      constexpr int SPLIT = MODE == PW_FPN ? 12 : K16;
      if constexpr (MODE == PW_FPNQ) {   // the coarse map's share of the sum, requested in front of the K loop
        static_assert(NT == 1, "one pixel tile per wave");
        const int pc = pixv[0] < a.P ? pixv[0] : a.P - 1;
        const int hw = a.H * a.W;
        const int b = pc / hw, rem = pc - b * hw;
        const int y = rem / a.W, x = rem - y * a.W;
        const float* q = a.in2 + ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * a.M + 4 * g;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) qv[mt] = 16 * mt + 4 * g < a.M ? *reinterpret_cast<const f32x4*>(q + 16 * mt) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (PRE) {
        constexpr int KP = K16 / 2;
        static_assert(SPLIT % 2 == 0, "a chunk pair does not straddle the two inputs");
        // one chunk pair: the pair's filter fragments from LDS, the pixel's two quads split into two fp16 terms, three products
        auto pair_step = [&](int sp, const f32x4 (&bq)[NT][2]) __attribute__((always_inline)) {
          yfv2_h8 b1[NT], b2[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            u32x4 t1, t2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const f32x4 v = bq[nt][e] * 16.0f;                                          // fp16's absolute floor: 2^-25 -> 2^-29
              const yfv2_h4 h1 = __builtin_convertvector(v, yfv2_h4);                       // v_cvt_pk_f16_f32 (RN)
              const yfv2_h4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4), yfv2_h4);   // the difference is exact
              const yfv2_u2 u1 = __builtin_bit_cast(yfv2_u2, h1), u2 = __builtin_bit_cast(yfv2_u2, h2);
              t1[2 * e] = u1[0]; t1[2 * e + 1] = u1[1];
              t2[2 * e] = u2[0]; t2[2 * e + 1] = u2[1];
            }
            b1[nt] = __builtin_bit_cast(yfv2_h8, t1);
            b2[nt] = __builtin_bit_cast(yfv2_h8, t2);
          }
          // three products, the small ones first; MG * NT independent accumulators: no MFMA waits for the one before it.  Output tiles in
          // groups of at most five (PW_DUAL's ten: 80 fragment registers at once otherwise)
          constexpr int MG = MT > 5 ? MT / 2 : MT;
#pragma unroll
          for (int m0 = 0; m0 < MT; m0 += MG) {
            if (m0) __builtin_amdgcn_sched_barrier(0);
            yfv2_h8 a1[MG], a2[MG];
#pragma unroll
            for (int mt = 0; mt < MG; ++mt) {
              const float* wq2 = wl + ((((m0 + mt) * KP + sp) * 2) * 64 + lane) * 4;
              a1[mt] = __builtin_bit_cast(yfv2_h8, *reinterpret_cast<const u32x4*>(wq2));
              a2[mt] = __builtin_bit_cast(yfv2_h8, *reinterpret_cast<const u32x4*>(wq2 + 256));
            }
#define PW_PROD(A_, B_)                                                \
  _Pragma("unroll") for (int mt = 0; mt < MG; ++mt)                    \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[m0 + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[mt], B_[nt], acc[m0 + mt][nt], 0, 0, 0);
            PW_PROD(a1, b2) PW_PROD(a2, b1) PW_PROD(a1, b1)
#undef PW_PROD
          }
        };
        if constexpr (KP <= 3) {
          // short K (fpn.conv1x1_2's C2 part, K = 96): the pixel's whole K requested at once - one memory round trip per tile
          f32x4 ball[KP][NT][2];
#pragma unroll
          for (int sp = 0; sp < KP; ++sp)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int e = 0; e < 2; ++e) ball[sp][nt][e] = *reinterpret_cast<const f32x4*>((2 * sp < SPLIT ? src0[nt] : src1[nt]) + 16 * (2 * sp + e));
#pragma unroll
          for (int sp = 0; sp < KP; ++sp) {
            __builtin_amdgcn_sched_barrier(0);                     // (pair by pair: the fragment reads of all pairs would be hoisted to the top otherwise)
            pair_step(sp, ball[sp]);
          }
        } else {
        f32x4 bcur[NT][2], bnxt[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int e = 0; e < 2; ++e) bcur[nt][e] = *reinterpret_cast<const f32x4*>(src0[nt] + 16 * e);
#pragma unroll 1
        for (int sp = 0; sp < KP; ++sp) {
          if (sp + 1 < KP) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int e = 0; e < 2; ++e) bnxt[nt][e] = *reinterpret_cast<const f32x4*>((2 * sp + 2 < SPLIT ? src0[nt] : src1[nt]) + 16 * (2 * sp + 2 + e));
          }
          pair_step(sp, bcur);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 2; ++e) bcur[nt][e] = bnxt[nt][e];
        }
        }
      } else {
      f32x4 bcur[NT], bnxt[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bcur[nt] = *reinterpret_cast<const f32x4*>(src0[nt]);
#pragma unroll 2
      for (int s = 0; s < K16; ++s) {
        if (s + 1 < K16) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            bnxt[nt] = *reinterpret_cast<const f32x4*>((s + 1 < SPLIT ? src0[nt] : src1[nt]) + 16 * (s + 1));
        }
        // all MT filter fragments of the chunk first (distinct registers), then MT*NT independent
        // MFMA chains interleaved - no LDS wait between MFMAs
        if constexpr (BF6) {
          Bf3A a3[MT];
          Bf3B b3[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a3[mt] = yfv2_split_a(*reinterpret_cast<const f32x4*>(wl + ((mt * K16 + s) * 64 + lane) * 4));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b3[nt] = yfv2_split_b(bcur[nt]);
          yfv2_mfma6_tiles<MT, NT>(a3, b3, acc);
        } else {
        f32x4 afs[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) afs[mt] = *reinterpret_cast<const f32x4*>(wl + ((mt * K16 + s) * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afs[mt][j], bcur[nt][j], acc[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
      }
      }
    } else {
    // ---- B fragments: all K channels of this lane's pixel(s), 16 B per load
    f32x4 bf[NT][K16 > 0 ? K16 : 1];
    f32x2 bt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pix = pix0 + nt * 16 + p;
      pixv[nt] = pix;
      const int pc = pix < a.P ? pix : a.P - 1;  // clamp: tail lanes load valid memory, never store
      if constexpr (MODE == PW_PLAIN || MODE == PW_HEAD) {
        const float* src = a.in + (size_t)pc * a.in_stride + a.in_off;
#pragma unroll
        for (int s = 0; s < K16; ++s) bf[nt][s] = *reinterpret_cast<const f32x4*>(src + 16 * s + 4 * g);
        if constexpr (KT) bt[nt] = *reinterpret_cast<const f32x2*>(src + 16 * K16 + 2 * g);
      } else if constexpr (MODE == PW_SHUFFLE) {
        const float* src = a.in + (size_t)pc * a.in_stride;
        float* cp = a.copy + (size_t)pc * a.copy_stride + a.copy_off;
        const bool ok = pix < a.P;
#pragma unroll
        for (int s = 0; s < K16; ++s) {
          const f32x4 q0 = *reinterpret_cast<const f32x4*>(src + 2 * (16 * s + 4 * g));
          const f32x4 q1 = *reinterpret_cast<const f32x4*>(src + 2 * (16 * s + 4 * g) + 4);
          bf[nt][s] = (f32x4){q0[1], q0[3], q1[1], q1[3]};
          if (ok) *reinterpret_cast<f32x4*>(cp + 16 * s + 4 * g) = (f32x4){q0[0], q0[2], q1[0], q1[2]};
        }
        if constexpr (KT) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(src + 2 * (16 * K16 + 2 * g));
          bt[nt] = (f32x2){q[1], q[3]};
          if (ok) *reinterpret_cast<f32x2*>(cp + 16 * K16 + 2 * g) = (f32x2){q[0], q[2]};
        }
      } else {  // PW_FPN: K = 288 = 192 (C3, upsampled) + 96 (C2)
        static_assert(MODE != PW_FPN || K == 288, "PW_FPN expects K=288");
        const int hw = a.H * a.W;
        const int b = pc / hw, rem = pc - b * hw;
        const int y = rem / a.W, x = rem - y * a.W;
        const float* s3 = a.in + ((size_t)(b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1)) * 192;
        const float* s2 = a.in2 + (size_t)pc * 96;
#pragma unroll
        for (int s = 0; s < 12; ++s) bf[nt][s] = *reinterpret_cast<const f32x4*>(s3 + 16 * s + 4 * g);
#pragma unroll
        for (int s = 12; s < K16; ++s) bf[nt][s] = *reinterpret_cast<const f32x4*>(s2 + 16 * (s - 12) + 4 * g);
      }
    }

#pragma unroll
    for (int s = 0; s < K16; ++s) {
      if constexpr (BF6) {
        Bf3A a3[MT];
        Bf3B b3[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a3[mt] = yfv2_split_a(*reinterpret_cast<const f32x4*>(wl + ((mt * K16 + s) * 64 + lane) * 4));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b3[nt] = yfv2_split_b(bf[nt][s]);
        yfv2_mfma6_tiles<MT, NT>(a3, b3, acc);
      } else {
      f32x4 afs[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) afs[mt] = *reinterpret_cast<const f32x4*>(wl + ((mt * K16 + s) * 64 + lane) * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afs[mt][j], bf[nt][s][j], acc[mt][nt], 0, 0, 0);
      }
    }
    if constexpr (KT) {  // 8-channel tail: group g owns channels 16*K16 + 2g, +1
      if constexpr (BF6) {   // as a chunk whose elements 2, 3 are zero
        Bf3A a3[MT];
        Bf3B b3[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x2 af = *reinterpret_cast<const f32x2*>(wl + FRAG_FL + (mt * 64 + lane) * 2);
          a3[mt] = yfv2_split_a((f32x4){af[0], af[1], 0.f, 0.f});
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b3[nt] = yfv2_split_b((f32x4){bt[nt][0], bt[nt][1], 0.f, 0.f});
        yfv2_mfma6_tiles<MT, NT>(a3, b3, acc);
      } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x2 af = *reinterpret_cast<const f32x2*>(wl + FRAG_FL + (mt * 64 + lane) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bt[nt][j], acc[mt][nt], 0, 0, 0);
      }
      }
    }

    }  // !STREAM

    // ---- epilogue: BN scale/shift (or bias), ReLU, store
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int pix = pixv[nt];
      if constexpr (PRE) watch.see(acc[0][nt][0]);
      if (pix >= a.P) continue;
      if constexpr (MODE == PW_HEAD) {
        const int b = pix / a.HW, hw = pix - b * a.HW;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 16 * mt + 4 * g + r;
            if (co < a.M) {
              const float y = __builtin_fmaf(acc[mt][nt][r], sc[mt][r], sh[mt][r]);
              if (co < a.split)
                a.nchw0[((size_t)b * (a.ctot0 ? a.ctot0 : a.split) + a.coff0 + co) * a.HW + hw] = y;
              else
                a.nchw1[((size_t)b * (a.M - a.split) + (co - a.split)) * a.HW + hw] = y;
            }
          }
      } else {
        float* dst = a.out + (size_t)pix * a.out_stride + a.out_off;
        float* dst2 = MODE == PW_DUAL ? a.copy + (size_t)pix * a.copy_stride + a.copy_off : dst;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          constexpr int MH = MODE == PW_DUAL ? MT / 2 : MT;
          const bool second = mt >= MH;                               // (compile-time per unrolled tile)
          const bool relu = a.relu && !second;
          if (16 * (mt % MH) + 4 * g < a.M) {  // M % 4 == 0 for every NHWC destination
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              y[r] = __builtin_fmaf(acc[mt][nt][r], sc[mt][r], MODE == PW_FPNQ ? qv[mt][r] : sh[mt][r]);
              if (relu) y[r] = y[r] > 0.f ? y[r] : 0.f;
            }
            *reinterpret_cast<f32x4*>((second ? dst2 : dst) + 16 * (mt % MH) + 4 * g) = y;
          }
        }
      }
    }
  }
  if constexpr (PRE) watch.report(a.nonfinite);
}

// FORMS: 0 = whatever the handle's plan asks for (fp32 MFMA / split on the fly / pre-split filter); 1 = the pre-split fp16x3 form only,
// 2 = the fp32-MFMA form only (instantiations the launcher picks for one plan: nothing else is compiled into the library)
template <int K, int MT, int NT, int MODE, int THREADS = 256, bool STREAM = false, int FORMS = 0>
static void pw_launch(const PwArgs& a, hipStream_t s) {
  const size_t lds = (size_t)MT * 16 * K * sizeof(float);   // K/16 fragments of 256 floats + (K%16 == 8) 128 per M tile
  const int n_super = (a.P + NT * 16 - 1) / (NT * 16);
  int blocks = (n_super + THREADS / 64 - 1) / (THREADS / 64);
  // persistent-ish grid: enough blocks to fill 256 CUs a few times over, few
  // enough that the per-block weight staging (L2 -> LDS) stays amortised
  const int cap = lds > 64 * 1024 ? 256 : (lds > 32 * 1024 ? 512 : 1024);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  static std::atomic<unsigned long long> lds_ok0{0}, lds_ok1{0};
  // bf16x6 for the instantiations the default plans use (the streamed large-K forms and the small biased heads); the fully
  // unrolled 6-tile forms of the layer-by-layer fallback would spill with the split operands and stay on the fp32 MFMA
  constexpr bool kBf6 = STREAM || (MT * (K / 16 + 1) <= 12);
  if constexpr (FORMS == 1) {   // only the pre-split fp16x3 form (yfv2_launch_pw asks for nothing else)
    static std::atomic<unsigned long long> lds_ok2{0};
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true, true>), lds_ok2);
    YFV2_LAUNCH((pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true, true>), dim3(blocks), dim3(THREADS), lds, s, a);
  } else if constexpr (FORMS == 2) {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&pw_kernel<K, MT, NT, MODE, THREADS, STREAM>), lds_ok0);
    YFV2_LAUNCH((pw_kernel<K, MT, NT, MODE, THREADS, STREAM>), dim3(blocks), dim3(THREADS), lds, s, a);
  } else {
  if constexpr (STREAM && K % 32 == 0 && K >= 192) if (a.bf6) {   // (the planner packs these filters pre-split whenever the handle runs fp16x3)
    static std::atomic<unsigned long long> lds_ok2{0};
    const size_t lds_pre = lds;   // two fp16 terms: the size of the fp32 image
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true, true>), lds_ok2);
    YFV2_LAUNCH((pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true, true>), dim3(blocks), dim3(THREADS), lds_pre, s, a);
    return;
  }
  if constexpr (kBf6 && !(STREAM && K % 32 == 0 && K >= 192)) if (a.bf6) {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true>), lds_ok1);
    YFV2_LAUNCH((pw_kernel<K, MT, NT, MODE, THREADS, STREAM, true>), dim3(blocks), dim3(THREADS), lds, s, a);
    return;
  }
  yfv2_allow_full_lds(reinterpret_cast<const void*>(&pw_kernel<K, MT, NT, MODE, THREADS, STREAM>), lds_ok0);
  YFV2_LAUNCH((pw_kernel<K, MT, NT, MODE, THREADS, STREAM>), dim3(blocks), dim3(THREADS), lds, s, a);
  }
}

// does yfv2_launch_pw have a pre-split (PRE) instantiation for this launch?  (the planner then packs the filter pre-split)
bool yfv2_pw_presplit_supported(int K, int mode, int M) {
  const int MT = (M + 15) / 16;
  return MT == 5 && (((mode == PW_PLAIN || mode == PW_DUAL) && K == 192) || (mode == PW_FPN && K == 288) || (mode == PW_FPNQ && K == 96));   // (PW_DUAL: M = each conv's rows)
}

// M tiles of the instantiation yfv2_launch_pw picks: the host packs the filter image for exactly that many
int yfv2_pw_tiles(int K, int mode, int M) {
  const int MT = (M + 15) / 16;
  if (mode == PW_HEAD && K == 72 && MT > 1 && MT <= 6) return 6;
  return MT;
}

bool yfv2_launch_pw(int K, int mode, const PwArgs& a, hipStream_t s) {
  const int MT = (a.M + 15) / 16;
  if (mode == PW_PLAIN) {
    if (K == 24 && MT == 2) { pw_launch<24, 2, 4, PW_PLAIN>(a, s); return true; }
    if (K == 48 && MT == 3) { pw_launch<48, 3, 4, PW_PLAIN>(a, s); return true; }
    if (K == 96 && MT == 6) { pw_launch<96, 6, 2, PW_PLAIN, 256, true>(a, s); return true; }
    if (K == 72 && MT == 5) { pw_launch<72, 5, 2, PW_PLAIN>(a, s); return true; }
    if (K == 192 && MT == 5) { pw_launch<192, 5, 1, PW_PLAIN, 512, true>(a, s); return true; }   // one pixel tile per wave: 121 pixels x 256 images are 1936 tiles for 2048 waves (two tiles: 15 -> 13.4 us)
  } else if (mode == PW_SHUFFLE) {
    if (K == 24 && MT == 2) { pw_launch<24, 2, 4, PW_SHUFFLE>(a, s); return true; }
    if (K == 48 && MT == 3) { pw_launch<48, 3, 4, PW_SHUFFLE>(a, s); return true; }
    if (K == 96 && MT == 6) { pw_launch<96, 6, 2, PW_SHUFFLE>(a, s); return true; }
  } else if (mode == PW_FPN) {
    // the default (pre-split fp16x3) form as 1024-thread workgroups with ONE pixel tile per wave: 16 waves per CU instead of 8 hide
    // more of the one-chunk-pair look-ahead's latency (32.3 -> 30.6 us same box; 115 registers).  Not faster: two tiles per wave
    // at 1024 threads (28 spilled registers), four tiles per wave at 512 (the same 31 us), and a form with a tile's whole K in
    // flight in two register sets (pwf_kernel, round 4: 32.0 us - the launch is not bound by that latency).  The fp32-matrix plan
    // runs the 512-thread form below.
    if (K == 288 && MT == 5 && a.bf6 && a.presplit) { pw_launch<288, 5, 1, PW_FPN, 1024, true, 1>(a, s); return true; }
    if (K == 288 && MT == 5 && !a.bf6) { pw_launch<288, 5, 2, PW_FPN, 512, true, 2>(a, s); return true; }
  } else if (mode == PW_DUAL) {
    if (K == 192 && MT == 5 && a.bf6 && a.presplit) { pw_launch<192, 10, 1, PW_DUAL, 512, true, 1>(a, s); return true; }
  } else if (mode == PW_FPNQ) {
    if (K == 96 && MT == 5 && a.bf6 && a.presplit) { pw_launch<96, 5, 1, PW_FPNQ, 1024, true, 1>(a, s); return true; }
  } else if (mode == PW_HEAD) {
    if (K == 72 && MT == 1) { pw_launch<72, 1, 4, PW_HEAD>(a, s); return true; }
    if (K == 72 && MT <= 6) { pw_launch<72, 6, 2, PW_HEAD>(a, s); return true; }
  }
  return false;
}

// ============================================================================
// depthwise kxk conv + BN (+ReLU), NHWC
// ============================================================================
// Thread = (output pixel, channel quad): KS*KS 16-byte loads that are contiguous
// across the lanes of a pixel (C*4 bytes) and across consecutive pixels, so
// every wave-level load is a dense run of 1 KiB; the KS-fold re-reads of
// neighbouring rows are served by L1/L2.  Pure HBM-bound work: no MFMA shape.
template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
  constexpr int PAD = KS / 2;
  const int C4 = a.C >> 2;
  const size_t total = (size_t)a.B * a.OH * a.OW * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    size_t pix = idx / C4;
    const int ox = (int)(pix % a.OW);
    size_t t = pix / a.OW;
    const int oy = (int)(t % a.OH);
    const int b = (int)(t / a.OH);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* inb = a.in + (size_t)b * a.H * a.W * a.in_stride + a.in_off + 4 * c4;
    const float* wp = a.w + 4 * c4;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const int iy = oy * STRIDE - PAD + ky;
      if (iy < 0 || iy >= a.H) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int ix = ox * STRIDE - PAD + kx;
        if (ix < 0 || ix >= a.W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(inb + ((size_t)iy * a.W + ix) * a.in_stride);
        const f32x4 w = *reinterpret_cast<const f32x4*>(wp + (ky * KS + kx) * a.C);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(v[k], w[k], acc[k]);
      }
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * c4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * c4);
    f32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      y[k] = __builtin_fmaf(acc[k], sc[k], sh[k]);
      if (a.relu) y[k] = y[k] > 0.f ? y[k] : 0.f;
    }
    *reinterpret_cast<f32x4*>(a.out + pix * a.out_stride + a.out_off + 4 * c4) = y;
  }
}

bool yfv2_launch_dw(int ksize, int stride, const DwArgs& a, hipStream_t s) {
  const size_t total = (size_t)a.B * a.OH * a.OW * (a.C >> 2);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride the rest
  if (blocks < 1) blocks = 1;
  if (ksize == 3 && stride == 1) { YFV2_LAUNCH((dw_kernel<3, 1>), dim3(blocks), dim3(256), 0, s, a); return true; }
  if (ksize == 3 && stride == 2) { YFV2_LAUNCH((dw_kernel<3, 2>), dim3(blocks), dim3(256), 0, s, a); return true; }
  if (ksize == 5 && stride == 1) { YFV2_LAUNCH((dw_kernel<5, 1>), dim3(blocks), dim3(256), 0, s, a); return true; }
  return false;
}
