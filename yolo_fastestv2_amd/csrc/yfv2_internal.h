// yfv2_internal.h - launch-argument structs shared by the kernel translation
// units and the host-side plan (yfv2_api.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));   // three floats at any dword address (global_load/store_dwordx3)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// n / d for 0 <= n < 2^16 and 8 <= d <= 1024 through the float pipe (cvt, fma, cvt instead of the ~25-instruction
// integer division): (n + 0.5) / d is at least 0.5/d away from an integer, far more than the fp32 error of the product.
#ifdef __HIPCC__
__device__ __forceinline__ int yfv2_fdiv(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(function, device) attribute: raise it to the 160 KiB cap the first
// time a function is launched on each device (`done` = the call site's bit mask of devices already served).
#include <atomic>

// Every launch of the forward path goes through YFV2_LAUNCH.  Normally a plain launch; while yfv2_profile_forward runs a plan step, the
// step's FIRST launch records its own begin and every launch its own end into the calling thread's event pair (hipExtLaunchKernel: the
// dispatch's own timestamps - what rocprofv3 reports - instead of hipEventRecord packets in front of and behind the launch, which
// read 3-12 us more per launch: round 6, profiles/r06_experiments.txt 14).
struct Yfv2LaunchProbe { hipEvent_t start = nullptr, stop = nullptr; int launches = 0; };
extern thread_local Yfv2LaunchProbe yfv2_launch_probe;
#define YFV2_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                        \
  do {                                                                                                                             \
    Yfv2LaunchProbe& yfv2_lp_ = yfv2_launch_probe;                                                                                 \
    hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, yfv2_lp_.launches++ == 0 ? yfv2_lp_.start : nullptr, yfv2_lp_.stop, 0, \
                          __VA_ARGS__);                                                                                            \
  } while (0)
inline void yfv2_allow_full_lds(const void* fn, std::atomic<unsigned long long>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  const unsigned long long bit = 1ull << dev;
  if (done.load(std::memory_order_relaxed) & bit) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  done.fetch_or(bit, std::memory_order_relaxed);
}

// ---- fp32 pointwise convs on the bf16 matrix cores ("bf16x6")
// gfx950's fp32 MFMA runs at the fp32 VECTOR rate (16x16x4: 32 cycles per 1024 MACs per SIMD); its bf16 MFMA is 16x
// faster (16x16x32: ~17 cycles per 8192 MACs).  An fp32 value is EXACTLY the sum of three bf16 terms (truncation: 8 + 8 + 8
// significant bits), so w * x = sum of nine bf16 x bf16 products, each exact in fp32; the six largest (everything above
// 2^-23 of |w x|, i.e. below one fp32 rounding of the product) are accumulated in fp32 by three MFMAs whose eight k-slots
// per lane group are the lane's 4 channels x 2 terms:
//     A {w.hi, w.hi} x B {x.hi, x.mid}      A {w.mid, w.mid} x B {x.hi, x.mid}      A {w.hi, w.lo} x B {x.lo, x.hi}
// Same fragment maps as v_mfma_f32_16x16x4_f32 (A lane l: row l&15; B lane l: pixel l&15; lane group l>>4 owns 4 channels of
// the 16-channel chunk; D lane l reg r: row 4(l>>4)+r, column l&15), so it drops into the existing loops: split the A
// fragment once per (tile of output channels, chunk), the B fragment once per (pixel tile, chunk), then mfma6().
// Measured (tools/ubench/bf16x6.hip, K = 192, 5 x 2 tiles): max error vs float64 7.1e-6 against 8.6e-6 for the fp32 MFMA on
// the same data; 30 MFMAs in ~520 cycles against 40 in 1280; the whole network's logits: 1.39e-5 vs 1.41e-5 from float64.
#ifdef __HIPCC__
typedef __bf16 yfv2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Bf3A { u32x4 hh, mm, hl; };   // {hi,hi}, {mid,mid}, {hi,lo}
struct Bf3B { u32x4 hm, lh; };       // {hi,mid}, {lo,hi}
__device__ __forceinline__ unsigned yfv2_pack_hi16(float a, float b) {   // high halves of a (low 16 bits) and b (high 16 bits): two truncated bf16
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float yfv2_trunc_bf16(float a) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u); }
__device__ __forceinline__ void yfv2_split3(f32x4 v, unsigned (&h)[2], unsigned (&m)[2], unsigned (&l)[2]) {
  float r1[4], r2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { r1[c] = v[c] - yfv2_trunc_bf16(v[c]); r2[c] = r1[c] - yfv2_trunc_bf16(r1[c]); }   // exact subtractions
  h[0] = yfv2_pack_hi16(v[0], v[1]); h[1] = yfv2_pack_hi16(v[2], v[3]);
  m[0] = yfv2_pack_hi16(r1[0], r1[1]); m[1] = yfv2_pack_hi16(r1[2], r1[3]);
  l[0] = yfv2_pack_hi16(r2[0], r2[1]); l[1] = yfv2_pack_hi16(r2[2], r2[3]);
}
__device__ __forceinline__ Bf3A yfv2_split_a(f32x4 w) {
  unsigned h[2], m[2], l[2];
  yfv2_split3(w, h, m, l);
  return {(u32x4){h[0], h[1], h[0], h[1]}, (u32x4){m[0], m[1], m[0], m[1]}, (u32x4){h[0], h[1], l[0], l[1]}};
}
__device__ __forceinline__ Bf3B yfv2_split_b(f32x4 x) {
  unsigned h[2], m[2], l[2];
  yfv2_split3(x, h, m, l);
  return {(u32x4){h[0], h[1], m[0], m[1]}, (u32x4){l[0], l[1], h[0], h[1]}};
}
// one of the three MFMAs (k = 0: the small terms, 1: w.mid, 2: w.hi): call sites run k outermost over their independent
// accumulators so that no MFMA waits for the one before it
template <int k>
__device__ __forceinline__ f32x4 yfv2_mfma6_step(const Bf3A& a, const Bf3B& b, f32x4 acc) {
  if constexpr (k == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(yfv2_bf16x8, a.hl), __builtin_bit_cast(yfv2_bf16x8, b.lh), acc, 0, 0, 0);
  else if constexpr (k == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(yfv2_bf16x8, a.mm), __builtin_bit_cast(yfv2_bf16x8, b.hm), acc, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(yfv2_bf16x8, a.hh), __builtin_bit_cast(yfv2_bf16x8, b.hm), acc, 0, 0, 0);
}
// acc[mt][nt] += A[mt] x B[nt] for all tiles of a 16-channel chunk
template <int MT_, int NT_>
__device__ __forceinline__ void yfv2_mfma6_tiles(const Bf3A (&a)[MT_], const Bf3B (&b)[NT_], f32x4 (&acc)[MT_][NT_]) {
#pragma unroll
  for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = yfv2_mfma6_step<0>(a[mt], b[nt], acc[mt][nt]);
#pragma unroll
  for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = yfv2_mfma6_step<1>(a[mt], b[nt], acc[mt][nt]);
#pragma unroll
  for (int mt = 0; mt < MT_; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = yfv2_mfma6_step<2>(a[mt], b[nt], acc[mt][nt]);
}
#endif

// ---- range guard of the fp16x3 arithmetic (include/yfv2.h yfv2_nonfinite)
// An operand beyond fp16's range (|activation| * 2^4 or |pixel| * 2^8 >= 65520) splits into (+Inf, -Inf); the three
// products of ANY filter entry with it are +-Inf or NaN and sum to NaN, for every output channel of that pixel - and the ReLU
// behind almost every pointwise conv (v_max_f32 / v_med3_f32 return the non-NaN operand) would turn that NaN into a clean,
// silent 0.  So every kernel of the fp16x3 plan looks at ONE accumulator element per lane of every pointwise product before
// its ReLU: a shift and an unsigned compare into a scalar register pair OR-ed into a sticky wave-uniform mask (no branch),
// and reports once at its end.  Non-finite INPUT values are caught by the same test.
#ifdef __HIPCC__
struct Yfv2Watch {
  unsigned long long m = 0;
  // an integer test of the exponent field (all ones: NaN or Inf), two VALU instructions: the kernel files are compiled with
  // -fno-honor-nans (bare v_max for the ReLUs), under which `v != v` and the class intrinsics fold to false
  __device__ __forceinline__ void see(float v) {
    const unsigned b = __builtin_bit_cast(unsigned, v);
    m |= __builtin_amdgcn_ballot_w64((b << 1) >= 0xff000000u);
  }
  __device__ __forceinline__ void report(int* flag) const {
    // the rare path.  A plain store of the constant 1 (every writer stores the same value): the word lives in host-mapped,
    // coherent memory (yfv2_create), where a store needs no PCIe atomic - the host can then LOOK at it without waiting for
    // any stream (yfv2_nonfinite_peek)
    if (m != 0 && flag != nullptr) *reinterpret_cast<volatile int*>(flag) = 1;
  }
};
#endif


// ---- stem: conv3x3 s2 (3->24) + BN + ReLU + maxpool3x3 s2, NCHW in -> NHWC out
struct StemArgs {
  const void* x;       // fp32 (B,3,H,W) in [0,1], or (u8_in) uint8 (B,H,W,3) in 0..255
  float* out;          // (B,H/4,W/4,24) NHWC, or pair planes (pp_out)
  const float* img;    // broadcast-form filter [11 regs][64 lanes] (BN scale folded) + shift[24], see WeightPacker::image_stem
  int B, H, W;
  int R;               // pooled rows per band (set by the launcher, divides H/4)
  int u8_in;           // input is uint8 NHWC; img_u8 = the filter image with 1/255 folded in
  const float* img_u8;
  int pp_out;          // 1: write pair planes [B][12][H/4][W/4][2] (stage 2 in lane-per-pixel form), 0: NHWC
  const float* img16;  // yfv2_stem16.hip (fp32 input on the f16 matrix cores): filter as two fp16 terms in MFMA A-operand order
                       // [tile 2][term 2][64 lanes][4 dwords] | shift * 2^sw [32] | 2^-sw (WeightPacker::image_stem16); null: use the 4x4x1 kernel
  int* nonfinite;      // the handle's sticky range-guard word (Yfv2Watch), or null
};

// ---- pointwise 1x1 conv on the fp32 MFMA, NHWC
enum { PW_PLAIN = 0, PW_SHUFFLE = 1, PW_FPN = 2, PW_HEAD = 3,
       // round 6: fpn.conv1x1_2 without its 192 upsampled channels.  A 1x1 conv commutes with the nearest-neighbour upsample, so the C3 part of
       // conv1x1_2 (fpn.py:57-59) is computed ONCE per coarse pixel - Q = scale2 * (W2[:, :192] C3) + shift2, by the launch that computes
       // conv1x1_3 from the same C3 (PW_DUAL: two filter images, even workgroups the first, odd workgroups the second -> `copy`, no ReLU) -
       // and the fine-map launch is a K = 96 conv over C2 whose epilogue adds Q at (y / 2, x / 2) (PW_FPNQ: `in2` = Q)
       PW_DUAL = 4, PW_FPNQ = 5 };
struct PwArgs {
  const float* in;     // NHWC activations (PW_FPN: C3, coarse map)
  const float* in2;    // PW_FPN only: C2, fine map
  float* out;          // NHWC output (unused for PW_HEAD)
  const float* img;    // filter fragments [MT][K/16][64][4] (+ 8-channel tail [MT][64][2]), then scale[MT*16], shift[MT*16]
  int P;               // pixels = B*H*W
  int M;               // real output channels (<= 16*MT)
  int in_stride;       // floats per input pixel
  int in_off;          // first input channel (PW_PLAIN)
  int out_stride;      // floats per output pixel
  int out_off;         // first output channel
  int relu;
  float* copy;         // PW_SHUFFLE: even input channels (pass-through branch) are copied to
  int copy_stride;     //   copy[pix*copy_stride + copy_off + j]
  int copy_off;
  int H, W;            // PW_FPN: fine map size; PW_HEAD: H*W in HW
  int HW;
  float* nchw0;        // PW_HEAD: co <  split -> nchw0[b][co][hw]
  float* nchw1;        // PW_HEAD: co >= split -> nchw1[b][co-split][hw]
  int split;
  int ctot0, coff0;    // PW_HEAD writing a channel RANGE of a wider tensor (more than 93 classes: the class head in slices of 96):
                       // nchw0 has ctot0 channels per image (0: = split) and this launch's channel co lands at coff0 + co
  int bf6;             // run the MFMAs as bf16x6 where the instantiation has that form (handle flag, YFV2_BF6=0 at create time clears it)
  int presplit;        // img holds the filter PRE-SPLIT into bf16 hi/mid/lo operand quads per chunk pair (WeightPacker::image_pw, streamed K = 192 / 288 forms; needs bf6)
  int* nonfinite;      // range-guard word (Yfv2Watch), or null
};

// ---- depthwise kxk conv + BN (+ReLU), NHWC, float4 over channels
struct DwArgs {
  const float* in;
  float* out;
  const float* w;      // [k*k][C]
  const float* scale;  // [C]
  const float* shift;  // [C]
  int B, H, W, C;      // input size, channels (C % 4 == 0)
  int OH, OW;
  int in_stride, in_off;
  int out_stride, out_off;
  int relu;
};

// ---- chains of fused ShuffleV2 stride-1 blocks (yfv2_block.hip)
struct BlockS1Args {
  const float* in;   // (B,H,W,2*C2) NHWC
  float* out;        // (B,H,W,2*C2) NHWC, distinct from in
  const float* img;  // the blocks' LDS images back to back (host-packed, WeightPacker::append_s1_bf6 / PlanBuilder::s1pool_block)
  long long* trace;  // debug: workgroup 0 / thread 0 writes s_memtime stamps at phase boundaries (or null)
  int B, H, W;
  int R;             // rows per work item (H % R == 0)
  int nblk;          // block_s1chain_kernel: blocks in the chain (img = their images back to back)
  int presplit;      // block_s1pool_kernel: the images hold W1 / W2 pre-split for bf16x6 (yfv2_s1pool_image_floats(true) floats each)
  int* nonfinite;    // range-guard word (Yfv2Watch), or null
  float* park;       // block_s1chain6_kernel: scratch for the parked values, yfv2_s1chain_park_floats() floats per image (any dead buffer)
};

// chain of N stride-1 blocks in one launch (block_s1chain6_kernel, C2 = 48, bf16x6 pointwise convs on host-pre-split
// filters); see PlanBuilder::s1chain_block for the channel bookkeeping shared by host and kernel
bool yfv2_s1chain_supported(int c2, int H, int W);
int yfv2_s1chain_image_floats();                                    // floats per block image (incl. the two int tables)
long yfv2_s1chain_park_floats(int H, int W, int nblk);              // park scratch per image
bool yfv2_launch_block_s1chain(const BlockS1Args& a, hipStream_t s);
// chain of stride-1 blocks with the whole 192-channel activation resident in LDS (block_s1pool_kernel, stage 4 at 11x11):
// natural channel order, no bookkeeping; img = per block three images of yfv2_s1pool_image_floats() floats (one per third)
bool yfv2_s1pool_supported(int c2, int H, int W);
int yfv2_s1pool_image_floats(bool presplit);
bool yfv2_launch_block_s1pool(const BlockS1Args& a, hipStream_t s);

// ---- fused ShuffleV2 stride-2 block (yfv2_block.hip)
struct BlockS2Args {
  const float* in;   // (B,H,W,CIN) NHWC
  float* out;        // (B,H/2,W/2,2*CIN) NHWC
  const float* img;  // LDS image: W1 | W2 | Wproj | main dw taps | proj dw taps | 10 BN vectors [10][KS]
  int B, H, W;       // input size
  int R;             // output rows per work item
  int s4_main_bands; // s4h_kernel, maps up to 16 columns wide: > 0 = one workgroup per image, waves 0 .. n-1 run the MAIN branch on n bands of rows,
                     // wave 3 the PROJ branch on all rows (the main branch is five times the proj branch's instructions); 0 = two units x two roles
  // pair-plane input (stage 2 in lane-per-pixel form, yfv2_stage2.hip): in = buffer 0 of the stage, pair p of
  // image b lives at in + b*pp_imgstride + p*H*W*2 (+ pp_bufstride floats if bit p of pp_mask is set): an image's two buffers are
  // adjacent (pp_bufstride = CIN*H*W, pp_imgstride twice that), so no offset grows with the batch
  int pp_in;
  unsigned pp_mask;
  long long pp_bufstride;
  long long pp_imgstride;
  int bf6;           // pw1 as bf16x6 (pair-plane input form)
  long long* trace;  // debug: per-wave cycle stamps of workgroup 0 (block_s2w_kernel; or null)
  const float* img16;  // s3h_kernel's image (stage3.0 from pair planes in streaming form, yfv2_stage2h.hip; WeightPacker::image_s3h) or null
  int* nonfinite;      // range-guard word (Yfv2Watch), or null
};
bool yfv2_s3h_supported(int H, int W);
void yfv2_launch_s3h(const BlockS2Args& a, hipStream_t s);
bool yfv2_s4h_supported(int H, int W);                         // stage4.0 (96 -> 192, NHWC in) in streaming form with two wave roles
void yfv2_launch_s4h(const BlockS2Args& a, hipStream_t s);

// ---- stage 2 in lane-per-pixel form (yfv2_stage2.hip)
// slot s = 2*pair + element of the pair-plane layout -> logical channel of the stage's FIRST block output
// (the stride-2 block) stored there.  Slots are grouped by the low three bits of the channel: stride-1
// block j of the stage reads exactly the pairs whose slot label has bit j set (see Stage2Layout).
__host__ __device__ inline int yfv2_stage2_channel(int slot) {
  const int p = slot >> 1, e = slot & 1;
  const int b0 = p / 12, b1 = (p % 12) / 6, b2 = (p % 6) / 3, h = 2 * (p % 3) + e;
  return b0 + 2 * b1 + 4 * b2 + 8 * h;
}
struct S1PxArgs {
  float* act;          // the stage: [max_batch][2 buffers][24 pairs][H][W][2] (src_off / dst_off carry the buffer: + 48*H*W floats for buffer 1)
  const float* img;    // w1q[10][64] | w2q[10][64] | (unused 24) | dw taps [9][24]  (yfv2_api.hip image_s1px)
  int B, H, W;
  int nstrips, nb, R;  // set by the launcher
  int img_stride;      // floats per image (2*48*H*W: both buffers)
  int num_records;     // bytes addressable from an image base (covers its copy in buffer 1)
  int src_off[12];     // byte offsets (from the image base in buffer 0) of the 12 branch pairs: where they are read ...
  int dst_off[12];     // ... and where the block's output for the same pairs is written (the other buffer)
  const float* img16;  // s1h_kernel's image (yfv2_stage2h.hip, WeightPacker::image_s1h); null: s1px_kernel
  int* nonfinite;      // range-guard word (Yfv2Watch), or null
};
// stride-2 block of stage 2 (24 -> 48 channels) in lane-per-pixel form; two wave roles (proj / main branch)
struct S2PxArgs {
  const float* in;     // stem output in pair planes: [B][12 pairs][IH][IW][2]
  float* act;          // stage-2 buffer 0: [max_batch][24 pairs][OH][OW][2]
  const float* img[2]; // role images: [0] proj: pwq[10][64] | taps[54][64]; [1] main: w1q | w2q | taps
  int B, IH, IW;
  int nstrips, nb, R;  // set by the launcher
  int in_stride, out_stride;       // floats per image
  int in_records, out_records;     // bytes addressable from an image base
  int st2_off[2][8];   // per role: byte offsets of the 8 pair planes its output positions 0..15 fill (8-byte stores)
  int st1_off[2][8];   // per role: byte offsets (plane + element) of output positions 16..23 (4-byte stores)
  const float* img16;  // s2h_kernel's image (yfv2_stage2h.hip, WeightPacker::image_s2h: both branches in one wave); null: the two role kernels
  int* nonfinite;      // range-guard word (Yfv2Watch), or null
};
// stem + stage2.0 in one wave (front_kernel, yfv2_stage2h.hip): the fp32 input image straight to stage 2's pair planes
struct FrontArgs {
  const void* x;         // fp32 (B,3,H,W), or (u8_in, front2_kernel only) uint8 (B,H,W,3)
  int H, W;
  int u8_in;
  const float* img_stem; // WeightPacker::image_stem16
  S2PxArgs s2;           // as for s2h_kernel (IH x IW = H/4 x W/4; in unused)
};
void yfv2_launch_front(const FrontArgs& a, hipStream_t s);
void yfv2_launch_s2px(const S2PxArgs& a, hipStream_t s);   // a.img16 set: yfv2_launch_s2h (one kernel); else two kernels (proj role, main role)
void yfv2_launch_s2h(const S2PxArgs& a, hipStream_t s);
bool yfv2_s1px_supported(int H, int W);
void yfv2_launch_s1px(const S1PxArgs& a, hipStream_t s);   // a.img16 set: yfv2_launch_s1h
void yfv2_launch_s1h(const S1PxArgs& a, hipStream_t s);

// ---- fused DWConvblock half (yfv2_block.hip): dw5x5+BN+ReLU -> pw72+BN [-> output conv]
struct TowerArgs {
  const float* in;   // (B,H,W,72) NHWC
  float* out;        // (B,H,W,72) NHWC when there is no chained output conv
  const float* img;  // LDS image: pw [80][84] | output conv [mh16][84] (if any) | dw taps [25][80] | scd shd scp shp bias [5][96]
  int has_head;      // chained output conv present
  int mh, split;     // co < split -> nchw0[b][co][hw], else nchw1[b][co-split][hw]
  float* nchw0; float* nchw1;
  int B, H, W;
  long long* trace;  // debug: per-wave cycle stamps of workgroup 0 (or null)
  int bf6;           // pointwise + chained output conv as bf16x6
  const float* img16;  // towerh_kernel's image (yfv2_towerh.hip), or null: tower2_kernel
  int chain;           // towers_kernel job list: bit 0 = the input is what the previous job left in LDS, bit 1 = the output stays in LDS for the next job (no global round trip)
  int* nonfinite;      // range-guard word (Yfv2Watch), or null
};

// ---- decode (handel_preds) and NMS
struct DecodeArgs {
  const float* reg[2];
  const float* obj[2];
  const float* cls[2];
  float* boxes;        // (B, rows, 5+classes), or
  float* cand;         // (B, rows, 8) compact candidate rows (non-null selects the compact kernel)
  int B, classes;
  int fh[2], fw[2];    // feature map sizes
  float stride[2];     // cfg.height / fh  (fp32, like the reference's python float)
  double anchors[12];
  int rows;            // 3*(fh0*fw0 + fh1*fw1)
};

struct NmsArgs {
  const float* boxes;  // (B, rows, 5+classes), or (B, rows, 8) when compact
  int compact;
  float* dets;         // (B, 300, 6)
  int32_t* idx;        // (B, 300)
  int32_t* count;      // (B)
  const int32_t* classes;  // optional class filter (device), may be null
  int n_classes;
  int B, rows, nc;
  float conf_thres;
  double iou_thres;
  long long* trace;    // debug: workgroup 0 / thread 0 writes cycle stamps at phase boundaries (or null)
};

// launchers (defined next to the kernels)
void yfv2_launch_stem(const StemArgs& a, hipStream_t s);      // picks yfv2_stem16.hip's kernel for fp32 input when a.img16 is set
void yfv2_launch_stem16(const StemArgs& a, hipStream_t s);
// K in {24,48,72,96,192,288}; mode PW_*; returns false if the (K, mode, M) combination has no kernel
bool yfv2_launch_pw(int K, int mode, const PwArgs& a, hipStream_t s);
int yfv2_pw_tiles(int K, int mode, int M);   // M tiles of the kernel instantiation yfv2_launch_pw uses (the host packs for that many)
bool yfv2_pw_presplit_supported(int K, int mode, int M);   // the launch has a pre-split (bf16 hi/mid/lo operand quads) form
bool yfv2_launch_dw(int ksize, int stride, const DwArgs& a, hipStream_t s);
int yfv2_block_s2_rows(int cin, int H, int W);
bool yfv2_launch_block_s2(int cin, const BlockS2Args& a, hipStream_t s);
bool yfv2_tower2_supported(int H, int W);                    // whole-image tower kernel: maps up to 22x22
bool yfv2_launch_tower2(const TowerArgs& a, hipStream_t s);
bool yfv2_towerh_supported(int H, int W);
bool yfv2_towerh_multi(int H, int W);
struct TowerJobs {   // up to four tower halves of one map size in ONE launch
  TowerArgs j[4];
  int n;
  int par;   // 0: a job LIST - every workgroup runs the jobs one after the other on its image (towers_kernel, maps up to 11x11);
             // 1: INDEPENDENT jobs side by side - workgroups [k gpj, (k + 1) gpj) run job k (towerh_kernel)
  int gpj;   // par: workgroups per job (set by the launcher)
  unsigned char lane_patch[128];   // towerp_kernel: the 2x2 patch of lane l in depthwise round r at [64 r + l] (255: none); set by the launcher
};
bool yfv2_launch_towerh(const TowerJobs& jobs, int mh_tiles, hipStream_t s);
// ---- evaluation statistics (get_batch_statistics): which detections are true positives
struct StatsArgs {
  const float* dets;     // (B, 300, 6) x1,y1,x2,y2,conf,cls rows of yfv2_nms / yfv2_detect
  const int32_t* count;  // (B)
  const float* targets;  // (T, 6) image index, label, x1, y1, x2, y2 (pixels), the layout evaluation() builds (utils.py:372-376)
  int32_t* tp;           // (B, 300) 1 = true positive
  int32_t* overflow;     // set to 1 if an image has more than STATS_MAX_TARGETS targets (result then invalid)
  int B, T;
  float iou_thres;
};
constexpr int STATS_MAX_TARGETS = 1024;
void yfv2_launch_stats(const StatsArgs& a, hipStream_t s);
// ---- training loss and its gradient w.r.t. the logits (yfv2_loss.hip; utils/loss.py:8-208)
struct LossMatch {            // one (scale, offset candidate, anchor, label) slot of build_target
  int valid, b, a, gj, gi, cls;
  float tb[4];                // target box: (gx - cell x, gy - cell y, gw, gh) in grid units, fp32
  double aw, ah;              // anchor / stride, float64
};
struct LossArgs {
  const float* reg[2]; const float* obj[2]; const float* cls[2];   // the six logit maps (NCHW)
  float* grad_reg[2]; float* grad_obj[2]; float* grad_cls[2];      // d total / d logits (all null: forward only); reg / cls pre-zeroed
  const float* targets;       // (T, 6) image, class, cx, cy, w, h (normalised), device
  LossMatch* matches;         // 2 * 5 * 3 * T slots
  unsigned char* tobj[2];     // objectness targets (B, 3, H, W), pre-zeroed
  int* nb;                    // matches per scale [2], pre-zeroed
  double* sums;               // per scale: sum(1 - ciou), sum(objectness BCE), sum(class CE)  [2][3], pre-zeroed
  float* losses;              // out: lbox, lobj, lcls, total
  int B, T, classes;
  int fh[2], fw[2];
  double stride[2];           // cfg.width / fw
  double anchors[12];
};
void yfv2_launch_loss(const LossArgs& a, hipStream_t s);
// ---- pre-process: bilinear resize of uint8 HWC frames (yfv2_pre.hip)
struct ResizeArgs {
  const unsigned char* src;  // (B, SH, SW, 3)
  unsigned char* dst;        // (B, H, W, 3), 4-byte aligned, W % 4 == 0
  int B, SH, SW, H, W;
  double scale_x, scale_y;   // 1 / (W / SW), 1 / (H / SH) in double, like cv::resize derives them
};
size_t yfv2_resize_lds_bytes(int SW, int W);
void yfv2_launch_resize(const ResizeArgs& a, hipStream_t s);
void yfv2_launch_decode(const DecodeArgs& a, hipStream_t s);
void yfv2_launch_nms(const NmsArgs& a, hipStream_t s);
void yfv2_launch_decode_nms(const DecodeArgs& d, const NmsArgs& a, hipStream_t s);   // yfv2_detect: decode + NMS in one launch
bool yfv2_post_fusable(int classes, int rows);   // ... which exists for up to 96 classes and 2048 decode rows; beyond: two launches
int yfv2_nms_max_rows();                         // decode rows per image nms_kernel handles (4096)
// ---- measurement: effective shader clock (yfv2_probe.hip)
struct ClockProbeArgs {
  unsigned long long* out;       // [workgroups][4]: shader cycles, reference ticks, XCC id, (unused)
  unsigned long long ref_ticks;  // how long to stay, in ticks of the constant reference clock (s_memrealtime)
  int busy;                      // 1: dependent FMAs between the stamps, 0: s_sleep (runs beside other work without disturbing it)
};
bool yfv2_launch_clock_probe(const ClockProbeArgs& a, int workgroups, hipStream_t s);

// ---- training path (yfv2_train.hip): its state hangs off the handle through an opaque slot owned by yfv2_api.hip
struct yfv2_ctx;
struct yfv2_config;
void** yfv2_ctx_train_slot(yfv2_ctx* h);                       // null handle -> null
const yfv2_config* yfv2_ctx_config(yfv2_ctx* h);
int yfv2_ctx_fail(yfv2_ctx* h, int code, const char* msg);      // records the message, returns code
void yfv2_train_release(void* train_state);                    // called by yfv2_destroy
