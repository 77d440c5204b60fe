// yfv2_post.hip - anchor decode and class-aware NMS for gfx950.
// Built with -ffp-contract=off: every fp32 operation below must round exactly
// like the reference's separate torch/numpy ops (no FMA contraction), see
// SURVEY.md App. B.  Reference behaviour: utils/utils.py:67-74 (xywh2xyxy),
// :232-296 (non_max_suppression), :298-358 (make_grid, handel_preds) and
// torchvision.ops.nms (called at :286).
#include "yfv2_internal.h"

// fp32 sigmoid as ATen's CPU kernel evaluates it: 1 / (1 + exp(-x)), true division
__device__ __forceinline__ float sigmoid_f32(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

// One compact candidate row (cx, cy, w, h | obj, conf, class, 0) of image b, scale sc, grid cell `cell`, anchor `part`
// (part 3 of a cell's 4-lane group only helps with the softmax).  Called by 4 consecutive lanes per cell with `cc` = the
// cell clamped into the map, so that the group's shuffles stay convergent; the result is meaningful for ok && part < 3.
// Arithmetic = handel_preds (utils/utils.py:303-358) followed by the first two steps of non_max_suppression
// (conf = max_j fl32(cls_j * obj) with the FIRST maximal j, :261,267), SURVEY.md App. B.
// `skip_ct` (the fused NMS launch): when none of the cell's three anchors has obj > skip_ct - their logits arrive in objl,
// fetched one pass ahead - the class slice is not read at all: non_max_suppression drops rows with obj <= conf_thres before it
// looks at a class (utils.py:254), so conf / class of such rows are never used (they are written as 0).  With trained weights
// that is nearly every cell: the launch's decode phase stops being a 164 MB read.
// MAXPER >= ceil(classes / 4): 24 for up to 96 classes (every register count and schedule note below refers to that form), 64
// for up to 255 (the general form: only reached through decode_kernel, 256 threads per workgroup).
template <bool SKIP = false, int MAXPER = 24>
__device__ __forceinline__ void yfv2_compact_row(const DecodeArgs& a, int b, int sc, int cc, int part, f32x4& r0, f32x4& r1,
                                                 const float (&objl)[3] = {0.f, 0.f, 0.f}, float skip_ct = 0.f) {
  const int fh = a.fh[sc], fw = a.fw[sc], hw = fh * fw;
  const int nc = a.classes;
  const int per = (nc + 3) >> 2;
  const int c_lo = part * per, c_hi = min(nc, c_lo + per);
  const float* cls = a.cls[sc] + (size_t)b * nc * hw;   // wave-uniform base: the per-lane part stays a 32-bit offset (24 addresses live)
  // The lane's class slice is loaded ONCE, all 24 loads in flight together (masked slots re-read a valid class), and the
  // three sweeps (max, sum, best product) run on registers; the second sweep leaves exp(v - max) in the logit's register,
  // the third divides it (this runs inside the 1024-thread NMS workgroup: 128 registers, and it is VALU-bound there -
  // four waves per SIMD - not bandwidth-bound: 24 fewer expf per lane and pass).  The first fused version re-read the
  // logits in every sweep in batches of four: 18 dependent round trips per pass instead of one.
  // Every value is computed by the same expression as in decode_kernel<false>, so the results are identical.
  float best[3] = {0.f, 0.f, 0.f};
  int bj[3] = {0, 0, 0};
  bool need = true;   // (the four lanes of a cell agree: the shuffles below stay inside a converged group of four)
  if constexpr (SKIP) need = sigmoid_f32(objl[0]) > skip_ct || sigmoid_f32(objl[1]) > skip_ct || sigmoid_f32(objl[2]) > skip_ct;
  if (need) {
    float lv[MAXPER];
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
      const int c = c_lo + i;
      lv[i] = cls[(unsigned)((c < c_hi ? c : (c_lo < nc ? c_lo : 0)) * hw + cc)];
    }
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i)
      if (c_lo + i < c_hi) m = fmaxf(m, lv[i]);
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
      const float e = expf(__fsub_rn(lv[i], m));
      lv[i] = e;                                            // the logit is dead from here on: its register keeps the exponential
      if (c_lo + i < c_hi) sum = __fadd_rn(sum, e);
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0); // four exps at a time (a free schedule interleaves all 24: spills)
    }
    sum = __fadd_rn(sum, __shfl_xor(sum, 1));
    sum = __fadd_rn(sum, __shfl_xor(sum, 2));
    float obj3[3];
#pragma unroll
    for (int an = 0; an < 3; ++an) {
      obj3[an] = sigmoid_f32(a.obj[sc][((size_t)b * 3 + an) * hw + cc]);
      best[an] = -INFINITY;
      bj[an] = 0x7fffffff;
    }
    // conf = max_j fl32(fl32(e_j / sum) * obj) and its FIRST maximal j (utils.py:261,267).  The map e -> fl32(fl32(e / sum) * obj)
    // is non-decreasing (division by and multiplication with a non-negative constant, round to nearest), and the largest
    // exponential is exactly 1 (its logit IS the maximum), so conf = fl32(fl32(1 / sum) * obj) - one division per cell instead
    // of one per class - and a class can only tie with it if its exponential is within two roundings of 1: >= 1 - 2^-20 is
    // a safe superset while the products are normal numbers.  Round 4: this sweep was a division, three multiplications and
    // three compare-selects per class, two thirds of the launch's decode phase (VALU-bound: 68 k of its 163 k ticks).
    // One such class in the cell (the usual case): it is the answer for all three anchors.  More than one (a near-tie of two
    // logits) or a product down among the denormals (ties reach further there; conf <= 2^-100 only matters for
    // conf_thres <= 0): the exact sweep, as before - rare, so its wave-level divergence costs nothing.
    const float ev1 = __fdiv_rn(1.0f, sum);
    int first = 0x7fffffff, cnt = 0;
#pragma unroll
    for (int i = 0; i < MAXPER; ++i) {
      const bool near_max = c_lo + i < c_hi && lv[i] >= 0.99999904632568359375f;   // 1 - 2^-20
      first = near_max ? min(first, c_lo + i) : first;
      cnt += near_max ? 1 : 0;
    }
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    first = min(first, __shfl_xor(first, 1));
    first = min(first, __shfl_xor(first, 2));
    bool exact = cnt != 1;
#pragma unroll
    for (int an = 0; an < 3; ++an) {
      best[an] = __fmul_rn(ev1, obj3[an]);
      bj[an] = first;
      exact |= !(best[an] > 7.888609052210118e-31f);   // 2^-100 (also catches a NaN logit: every comparison false, as in the exact sweep)
    }
    if (exact) {   // the four lanes of a cell agree on this (cnt and obj3 are cell-wide)
#pragma unroll
      for (int an = 0; an < 3; ++an) { best[an] = -INFINITY; bj[an] = 0x7fffffff; }
#pragma unroll
      for (int i = 0; i < MAXPER; ++i) {
        if (c_lo + i < c_hi) {
          const float ev = __fdiv_rn(lv[i], sum);   // the class probability, as decode_kernel<false> stores it
#pragma unroll
          for (int an = 0; an < 3; ++an) {
            const float pj = __fmul_rn(ev, obj3[an]);
            if (pj > best[an]) { best[an] = pj; bj[an] = c_lo + i; }  // strict: first maximal index of this slice
          }
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int an = 0; an < 3; ++an)
#pragma unroll
        for (int msk = 1; msk < 4; msk <<= 1) {
          const float ob_ = __shfl_xor(best[an], msk);
          const int oi = __shfl_xor(bj[an], msk);
          if (ob_ > best[an] || (ob_ == best[an] && oi < bj[an])) { best[an] = ob_; bj[an] = oi; }
        }
    }
  }
  const int an = part < 3 ? part : 0;
  const int y = cc / fw, x = cc - y * fw;
  const float* reg = a.reg[sc] + ((size_t)b * 12 + an * 4) * hw + cc;
  const float t0 = reg[0], t1 = reg[(size_t)hw], t2 = reg[(size_t)2 * hw], t3 = reg[(size_t)3 * hw];
  const float ob = a.obj[sc][((size_t)b * 3 + an) * hw + cc];
  const float st = a.stride[sc];
  const float qw = __fmul_rn(sigmoid_f32(t2), 2.0f), qh = __fmul_rn(sigmoid_f32(t3), 2.0f);
  r0 = (f32x4){__fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sigmoid_f32(t0), 2.0f), 0.5f), (float)x), st),
               __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sigmoid_f32(t1), 2.0f), 0.5f), (float)y), st),
               (float)((double)__fmul_rn(qw, qw) * a.anchors[(sc * 3 + an) * 2 + 0]),
               (float)((double)__fmul_rn(qh, qh) * a.anchors[(sc * 3 + an) * 2 + 1])};
  r1 = (f32x4){sigmoid_f32(ob), part == 0 ? best[0] : (part == 1 ? best[1] : best[2]), (float)(part == 0 ? bj[0] : (part == 1 ? bj[1] : bj[2])), 0.f};
}

// ============================================================================
// decode  (handel_preds)
// ============================================================================
// One block = up to 64 consecutive grid cells of one (image, scale).  4 lanes
// share a cell's 80-way softmax (20 classes each, 4-lane xor-shuffle reduce);
// lanes 0..2 of the group additionally decode anchor a = lane.  The block
// assembles its 64 x 3 x 85 output rows in LDS and streams them out as one
// contiguous, fully coalesced span (rows of consecutive cells are adjacent in
// the (y, x, anchor) row order).
constexpr int DEC_CELLS = 64;
constexpr int DEC_THREADS = 256;

template <bool COMPACT, int MAXPER = 24>
__global__ __launch_bounds__(DEC_THREADS) void decode_kernel(DecodeArgs a, int blocks0, int blocks1) {
  constexpr int CELLS = MAXPER > 24 ? 32 : DEC_CELLS, THREADS = 4 * CELLS;   // (the wide form halves the cells: [32][3][260] floats of LDS)
  extern __shared__ float stage[];  // [CELLS][3][5+classes]
  const int per_img = blocks0 + blocks1;
  const int b = blockIdx.x / per_img;
  int blk = blockIdx.x - b * per_img;
  const int sc = blk >= blocks0 ? 1 : 0;
  if (sc) blk -= blocks0;
  const int fh = a.fh[sc], fw = a.fw[sc], hw = fh * fw;
  const int cell0 = blk * CELLS;
  const int ncell = min(CELLS, hw - cell0);
  const int nc = a.classes, rowlen = 5 + nc;
  const int tid = threadIdx.x;
  const int lc = tid >> 2, part = tid & 3;  // local cell, quarter
  const int cell = cell0 + lc;
  const bool ok = lc < ncell;
  const int cc = ok ? cell : cell0;  // clamp so that shuffles stay convergent

  // ---- class softmax (fp32): exp(x - max) / sum.  The lane's logits are fetched with one
  // batch of independent loads (fixed trip count, masked) instead of a load per loop turn.
  const int per = (nc + 3) >> 2;
  const int c_lo = part * per, c_hi = min(nc, c_lo + per);
  const float* cls = a.cls[sc] + (size_t)b * nc * hw;   // wave-uniform base: the per-lane part stays a 32-bit offset (24 addresses live)
  float lv[MAXPER];
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    const int c = c_lo + i;
    lv[i] = cls[(unsigned)((c < c_hi ? c : (c_lo < nc ? c_lo : 0)) * hw + cc)];   // masked slots re-read a valid class (fewer than 4 classes: quarters 1-3 are empty)
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXPER; ++i)
    if (c_lo + i < c_hi) m = fmaxf(m, lv[i]);
  m = fmaxf(m, __shfl_xor(m, 1));
  m = fmaxf(m, __shfl_xor(m, 2));
  float sum = 0.f;
  float ev[MAXPER];
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) {
    ev[i] = 0.f;
    if (c_lo + i < c_hi) {
      ev[i] = expf(__fsub_rn(lv[i], m));
      sum = __fadd_rn(sum, ev[i]);
    }
  }
  sum = __fadd_rn(sum, __shfl_xor(sum, 1));
  sum = __fadd_rn(sum, __shfl_xor(sum, 2));
#pragma unroll
  for (int i = 0; i < MAXPER; ++i) ev[i] = __fdiv_rn(ev[i], sum);  // class probabilities of this lane's slice

  // ---- box + objectness of anchor `an` at this cell (App. B steps 2-4)
  auto box_of = [&](int an, float (&o)[5]) {
    const int y = cell / fw, x = cell - y * fw;
    const float* reg = a.reg[sc] + ((size_t)b * 12 + an * 4) * hw + cell;
    const float t0 = reg[0], t1 = reg[(size_t)hw], t2 = reg[(size_t)2 * hw], t3 = reg[(size_t)3 * hw];
    const float ob = a.obj[sc][((size_t)b * 3 + an) * hw + cell];
    const float st = a.stride[sc];
    o[0] = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sigmoid_f32(t0), 2.0f), 0.5f), (float)x), st);
    o[1] = __fmul_rn(__fadd_rn(__fsub_rn(__fmul_rn(sigmoid_f32(t1), 2.0f), 0.5f), (float)y), st);
    const float qw = __fmul_rn(sigmoid_f32(t2), 2.0f), qh = __fmul_rn(sigmoid_f32(t3), 2.0f);
    // fp32 square, then a float64 multiply by the float64 anchor, one rounding to fp32
    o[2] = (float)((double)__fmul_rn(qw, qw) * a.anchors[(sc * 3 + an) * 2 + 0]);
    o[3] = (float)((double)__fmul_rn(qh, qh) * a.anchors[(sc * 3 + an) * 2 + 1]);
    o[4] = sigmoid_f32(ob);
  };
  const size_t row0 = (size_t)b * a.rows + (sc ? 3 * a.fh[0] * a.fw[0] : 0) + (size_t)cell0 * 3;

  if constexpr (COMPACT) {
    // compact candidate rows (8 floats) instead of the 85-wide tensor: what the two-launch form of yfv2_detect consumes
    f32x4 r0, r1;
    yfv2_compact_row<false, MAXPER>(a, b, sc, cc, part, r0, r1);
    if (ok && part < 3) {
      float* d = a.cand + (row0 + (size_t)lc * 3 + part) * 8;
      *reinterpret_cast<f32x4*>(d) = r0;
      *reinterpret_cast<f32x4*>(d + 4) = r1;
    }
  } else {
    float* srow = stage + (size_t)lc * 3 * rowlen;
    if (ok) {
#pragma unroll
      for (int i = 0; i < MAXPER; ++i) {
        const int c = c_lo + i;
        if (c < c_hi) {
          srow[5 + c] = ev[i];
          srow[rowlen + 5 + c] = ev[i];  // the 3 anchors share the class scores (utils.py:324-326)
          srow[2 * rowlen + 5 + c] = ev[i];
        }
      }
      if (part < 3) {
        float o[5];
        box_of(part, o);
        float* d = srow + part * rowlen;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = o[k];
      }
    }
    __syncthreads();
    float* dst = a.boxes + row0 * rowlen;
    const int n = ncell * 3 * rowlen;
    for (int i = tid; i < n; i += THREADS) dst[i] = stage[i];
  }
}

void yfv2_launch_decode(const DecodeArgs& a, hipStream_t s) {
  const int cells = a.classes <= 96 ? DEC_CELLS : 32;
  const int b0 = (a.fh[0] * a.fw[0] + cells - 1) / cells;
  const int b1 = (a.fh[1] * a.fw[1] + cells - 1) / cells;
  const size_t lds = (size_t)cells * 3 * (5 + a.classes) * sizeof(float);
  static std::atomic<unsigned long long> lds_ok0{0}, lds_ok1{0}, lds_ok2{0}, lds_ok3{0};
  if (a.classes <= 96) {
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&decode_kernel<false>), lds_ok0);
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&decode_kernel<true>), lds_ok1);
    if (a.cand)
      YFV2_LAUNCH(decode_kernel<true>, dim3(a.B * (b0 + b1)), dim3(DEC_THREADS), 0, s, a, b0, b1);
    else
      YFV2_LAUNCH(decode_kernel<false>, dim3(a.B * (b0 + b1)), dim3(DEC_THREADS), lds, s, a, b0, b1);
  } else {   // up to 255 classes: 64 class slots per lane, 32 cells per workgroup ([32][3][260] floats of LDS)
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&decode_kernel<false, 64>), lds_ok2);
    yfv2_allow_full_lds(reinterpret_cast<const void*>(&decode_kernel<true, 64>), lds_ok3);
    if (a.cand)
      YFV2_LAUNCH((decode_kernel<true, 64>), dim3(a.B * (b0 + b1)), dim3(4 * cells), 0, s, a, b0, b1);
    else
      YFV2_LAUNCH((decode_kernel<false, 64>), dim3(a.B * (b0 + b1)), dim3(4 * cells), lds, s, a, b0, b1);
  }
}
// the fused decode + NMS launch (nms_kernel<2>) holds a lane's class slice in 24 registers and the image's rows in LDS
bool yfv2_post_fusable(int classes, int rows) { return classes <= 96 && rows <= 2048; }
int yfv2_nms_max_rows() { return 4096; }

// ============================================================================
// class-aware greedy NMS  (non_max_suppression + torchvision.ops.nms)
// ============================================================================
// One workgroup per image, everything in LDS:
//   1. filter   obj > ct ; conf = max_j fl32(cls_j*obj) (first max) ; conf > ct ; class filter
//   2. sort     bitonic sort of 64-bit keys (conf bits << 32 | ~row): descending
//               conf, ties -> lower row first (stable order of the reference's
//               filtered list); conf > ct >= 0 so the fp32 bit pattern is monotone
//   3. prepare  xywh -> xyxy, + cls*4096 offset, area - all in fp32 exactly as
//               utils.py:67-74,283-285 and torchvision compute them
//   4. greedy   walk the sorted list; a kept box suppresses later boxes whose
//               (double)iou > iou_thres; stop after max_det (300) kept boxes
//               (the reference truncates the full result to 300, same set)
constexpr int NMS_THREADS = 1024;  // 16 waves: one workgroup per image is latency-bound, occupancy is what hides it
constexpr int NMS_NQ = NMS_THREADS / 64;  // thread groups per 64-candidate chunk
constexpr int NMS_CAP = 2048;      // >= rows (1815); power of two for the bitonic network (KPT = 2 keys per thread; KPT = 4: 4096 rows)
constexpr int NMS_MAX_DET = 300;   // utils/utils.py:243 (== YFV2_MAX_DET)

// SRC 0: the (B, rows, 5 + classes) decoded tensor (yfv2_nms); 1: compact candidate rows in global memory (decode_kernel<true>);
// 2: the logits themselves - the workgroup decodes its image into compact rows in LDS first (yfv2_detect: one launch for
// handel_preds + non_max_suppression, the candidate rows never exist in HBM)
#define NMS_STAMP(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = (long long)__builtin_readcyclecounter(); } while (0)
// KPT = sort keys per thread: 2 (up to 2048 rows: everything the notes below measured) or 4 (up to 4096 rows - inputs beyond
// 352x352, e.g. 416x416 = 2535 rows, 512x512 = 3840; the sort's second exchange buffer then shares LDS with the box arrays,
// which are written after it, so that 4096 candidates fit into 132 KB)
template <int SRC, int KPT = 2>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(NmsArgs a, DecodeArgs dec) {
  constexpr bool COMPACT = SRC >= 1;
  constexpr int CAP = KPT * NMS_THREADS;
  static_assert(KPT == 2 || (KPT == 4 && SRC != 2), "two keys per thread, or four without the in-LDS decode");
  extern __shared__ __attribute__((aligned(16))) float crow[];   // SRC == 2: [rows][8]
  __shared__ unsigned long long key[CAP];
  __shared__ __attribute__((aligned(16))) float geo[(KPT == 2 ? 7 : 5) * CAP];
  unsigned long long* const key2 = reinterpret_cast<unsigned long long*>(geo);   // second exchange buffer of the sort
  float* const bx1 = geo + (KPT == 2 ? 2 * CAP : 0);
  float* const by1 = bx1 + CAP; float* const bx2 = by1 + CAP; float* const by2 = bx2 + CAP; float* const area = by2 + CAP;
  __shared__ unsigned char supp[CAP];
  __shared__ unsigned char cls_of_row[CAP];
  __shared__ int keep[NMS_MAX_DET];
  __shared__ float kx1[NMS_MAX_DET], kx2[NMS_MAX_DET];   // x interval of the kept boxes, in kept order (greedy step's first test)
  __shared__ unsigned long long pmask[NMS_NQ][64];
  __shared__ int n_cand, n_keep, n_obj;

  const int tid = threadIdx.x, b = blockIdx.x;
  const int rowlen = COMPACT ? 8 : 5 + a.nc;  // COMPACT rows: cx,cy,w,h,obj,conf,cls,0 (decode_kernel<true>)
  const float* img = SRC == 2 ? crow : a.boxes + (size_t)b * a.rows * rowlen;
  const float ct = a.conf_thres;
  NMS_STAMP(0);
  if constexpr (SRC == 2) {
    // decode this image: 4 lanes per grid cell (yfv2_compact_row), 256 cells per pass over the workgroup
    constexpr int CPP = NMS_THREADS / 4;                 // cells per pass
    const int hw0 = dec.fh[0] * dec.fw[0], hw1 = dec.fh[1] * dec.fw[1];
    const int np0 = (hw0 + CPP - 1) / CPP, np = np0 + (hw1 + CPP - 1) / CPP;
    const int part = tid & 3;
    // the three objectness logits of this lane's cell, one pass ahead of their use (they decide whether the pass reads
    // its class logits at all)
    auto obj_of = [&](int ps, float (&o)[3]) {
      const int sc = ps < np0 ? 0 : 1, hw = sc ? hw1 : hw0;
      const int cell = (ps - (sc ? np0 : 0)) * CPP + (tid >> 2);
      const int cc = cell < hw ? cell : hw - 1;
#pragma unroll
      for (int an = 0; an < 3; ++an) o[an] = ps < np ? dec.obj[sc][((size_t)b * 3 + an) * hw + cc] : 0.f;
    };
    float ocur[3], onxt[3];
    obj_of(0, ocur);
    for (int ps = 0; ps < np; ++ps) {
      obj_of(ps + 1, onxt);
      const int sc = ps < np0 ? 0 : 1, hw = sc ? hw1 : hw0;
      const int row_base = sc ? 3 * hw0 : 0;
      const int cell = (ps - (sc ? np0 : 0)) * CPP + (tid >> 2);
      const bool ok = cell < hw;
      f32x4 r0, r1;
      yfv2_compact_row<true>(dec, b, sc, ok ? cell : hw - 1, part, r0, r1, ocur, ct);
      if (ok && part < 3) {
        float* d = crow + (size_t)(row_base + cell * 3 + part) * 8;
        *reinterpret_cast<f32x4*>(d) = r0;
        *reinterpret_cast<f32x4*>(d + 4) = r1;
      }
#pragma unroll
      for (int an = 0; an < 3; ++an) ocur[an] = onxt[an];
    }
    __syncthreads();
  }
  NMS_STAMP(1);   // decoded
  unsigned short* cand_row = reinterpret_cast<unsigned short*>(bx1);  // step 1 only; bx1 is written in step 3
  if (tid == 0) { n_cand = 0; n_keep = 0; n_obj = 0; }
  __syncthreads();

  // ---- 1. filter.  (a) rows with obj > ct are compacted into cand_row[] (order is
  // irrelevant: the sort key carries the row).  (b) each 16-lane group takes one such
  // row: lane l scans classes l, l+16, ... (64-byte coalesced reads), keeps its first
  // maximum of fl32(cls_j*obj), and a 4-step xor-shuffle picks the group's maximum with
  // the LOWEST class index among equals (= torch.max's first-index rule, utils.py:267).
  if constexpr (COMPACT) {
    // (one LDS atomic per WAVE, not per candidate: with every row a candidate - the bench's random weights - 1815
    // atomics on one word serialise into tens of microseconds)
    for (int n0 = 0; n0 < a.rows; n0 += NMS_THREADS) {
      const int n = n0 + tid;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (n < a.rows) t = *reinterpret_cast<const f32x4*>(img + (size_t)n * 8 + 4);  // obj, conf, cls
      const bool pass = n < a.rows && t[0] > ct && t[1] > ct;
      const unsigned long long m = __ballot(pass);
      int base = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&n_cand, (int)__popcll(m));
      base = __shfl(base, 0);
      if (pass) {
        const int slot = base + (int)__popcll(m & ((1ull << (tid & 63)) - 1ull));
        key[slot] = ((unsigned long long)__float_as_uint(t[1]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
        cls_of_row[n] = (unsigned char)(int)t[2];
      }
    }
  } else {
    for (int n0 = 0; n0 < a.rows; n0 += NMS_THREADS) {
      const int n = n0 + tid;
      const bool pass = n < a.rows && img[(size_t)n * rowlen + 4] > ct;
      const unsigned long long m = __ballot(pass);
      int base = 0;
      if ((tid & 63) == 0 && m) base = atomicAdd(&n_obj, (int)__popcll(m));
      base = __shfl(base, 0);
      if (pass) cand_row[base + (int)__popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)n;
    }
    __syncthreads();
    {
      const int l16 = tid & 15, grp = tid >> 4, ngrp = NMS_THREADS >> 4;
      const int nobj = n_obj;
      for (int q = grp; q < nobj; q += ngrp) {  // uniform within a 16-lane group
        const int n = cand_row[q];
        const float* r = img + (size_t)n * rowlen;
        const float obj = r[4];
        float best = -INFINITY;
        int bj = 0x7fffffff;
        for (int j = l16; j < a.nc; j += 16) {
          const float pj = __fmul_rn(r[5 + j], obj);
          if (pj > best) { best = pj; bj = j; }  // strict: first maximal index of this lane
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          const float ob = __shfl_xor(best, m, 16);
          const int oj = __shfl_xor(bj, m, 16);
          if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (l16 == 0 && best > ct) {
          bool hit = true;
          if (a.classes) {
            hit = false;
            for (int k = 0; k < a.n_classes; ++k) hit |= (a.classes[k] == bj);
          }
          if (hit) {
            const int slot = atomicAdd(&n_cand, 1);
            key[slot] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)n);
            cls_of_row[n] = (unsigned char)bj;
          }
        }
      }
    }
  }
  __syncthreads();
  NMS_STAMP(2);   // filtered
  const int n = n_cand;
  if (n == 0) {
    if (tid == 0) a.count[b] = 0;
    return;
  }

  // ---- 2. sort (descending) on the next power of two >= n, zero keys pad the end.  Bitonic network with the keys in
  // REGISTERS (thread t owns indices t and t + 1024): a compare-exchange distance below 64 is a lane shuffle inside the
  // wave, distance 1024 is the thread's own second key, and only distances 64..512 go through LDS - 14 barriers for
  // 2048 keys instead of 66, alternating two key buffers so that one barrier per LDS step is enough.
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + tid; i < np2; i += NMS_THREADS) key[i] = 0ull;
  __syncthreads();
  if constexpr (KPT == 2) {
    static_assert(NMS_CAP == 2 * NMS_THREADS, "two keys per thread");
    const int i0 = tid, i1 = tid + NMS_THREADS;
    unsigned long long k0 = i0 < np2 ? key[i0] : 0ull, k1 = i1 < np2 ? key[i1] : 0ull;
    auto cx = [](unsigned long long x, unsigned long long y, bool take_max) { return take_max == (x > y) ? x : y; };   // x != y or both zero
    int pb = 0;
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        const bool d0 = (i0 & k) == 0, d1 = (i1 & k) == 0;          // descending run?
        if (j >= NMS_THREADS) {                                     // partner = this thread's other key (i0 < i1)
          const unsigned long long hi = k0 > k1 ? k0 : k1, lo = k0 > k1 ? k1 : k0;
          k0 = d0 ? hi : lo; k1 = d0 ? lo : hi;
        } else if (j >= 64) {                                       // partner in another wave: through LDS
          unsigned long long* buf = pb ? key2 : key;
          buf[i0] = k0; buf[i1] = k1;
          __syncthreads();
          const unsigned long long y0 = buf[i0 ^ j], y1 = buf[i1 ^ j];
          const bool low = (i0 & j) == 0;                           // same bit for i0 and i1 (j < 1024)
          k0 = cx(k0, y0, low == d0); k1 = cx(k1, y1, low == d1);
          pb ^= 1;
        } else {                                                    // partner = lane ^ j of the same wave
          const unsigned long long y0 = __shfl_xor(k0, j), y1 = __shfl_xor(k1, j);
          const bool low = (i0 & j) == 0;
          k0 = cx(k0, y0, low == d0); k1 = cx(k1, y1, low == d1);
        }
      }
    }
    __syncthreads();                                                // readers of either buffer are done
    if (i0 < np2) key[i0] = k0;
    if (i1 < np2) key[i1] = k1;
    __syncthreads();
  } else {
    // the same network with KPT keys per thread: thread t owns indices t + 1024 r; a distance >= 1024 pairs two of its own keys
    unsigned long long k[KPT];
#pragma unroll
    for (int r = 0; r < KPT; ++r) { const int i = tid + r * NMS_THREADS; k[r] = i < np2 ? key[i] : 0ull; }
    auto cx = [](unsigned long long x, unsigned long long y, bool take_max) { return take_max == (x > y) ? x : y; };
    int pb = 0;
    for (int kk = 2; kk <= np2; kk <<= 1) {
      for (int j = kk >> 1; j > 0; j >>= 1) {
        if (j >= NMS_THREADS) {
          const int jr = j / NMS_THREADS;                           // 1 or 2: partner key r ^ jr of the same thread
#pragma unroll
          for (int r = 0; r < KPT; ++r) {
            if ((r & jr) == 0) {
              const int r2 = r | jr;
              const bool desc = ((tid + r * NMS_THREADS) & kk) == 0;  // (the same for both keys: jr * 1024 < kk)
              const unsigned long long hi = k[r] > k[r2] ? k[r] : k[r2], lo = k[r] > k[r2] ? k[r2] : k[r];
              k[r] = desc ? hi : lo; k[r2] = desc ? lo : hi;
            }
          }
        } else if (j >= 64) {
          unsigned long long* buf = pb ? key2 : key;
#pragma unroll
          for (int r = 0; r < KPT; ++r) buf[tid + r * NMS_THREADS] = k[r];
          __syncthreads();
          const bool low = (tid & j) == 0;
#pragma unroll
          for (int r = 0; r < KPT; ++r) {
            const int i = tid + r * NMS_THREADS;
            k[r] = cx(k[r], buf[i ^ j], low == ((i & kk) == 0));
          }
          pb ^= 1;
        } else {
          const bool low = (tid & j) == 0;
#pragma unroll
          for (int r = 0; r < KPT; ++r) {
            const int i = tid + r * NMS_THREADS;
            k[r] = cx(k[r], __shfl_xor(k[r], j), low == ((i & kk) == 0));
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KPT; ++r) { const int i = tid + r * NMS_THREADS; if (i < np2) key[i] = k[r]; }
    __syncthreads();
  }

  NMS_STAMP(3);   // sorted
  // ---- 3. per-candidate geometry in sorted order
  for (int i = tid; i < n; i += NMS_THREADS) {
    const unsigned row = 0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull);
    const float* r = img + (size_t)row * rowlen;
    const float cx = r[0], cy = r[1], hw_ = __fdiv_rn(r[2], 2.0f), hh = __fdiv_rn(r[3], 2.0f);
    const float c = __fmul_rn((float)cls_of_row[row], 4096.0f);
    const float x1 = __fadd_rn(__fsub_rn(cx, hw_), c), y1 = __fadd_rn(__fsub_rn(cy, hh), c);
    const float x2 = __fadd_rn(__fadd_rn(cx, hw_), c), y2 = __fadd_rn(__fadd_rn(cy, hh), c);
    bx1[i] = x1; by1[i] = y1; bx2[i] = x2; by2[i] = y2;
    area[i] = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    supp[i] = 0;
  }
  __syncthreads();

  NMS_STAMP(4);   // geometry
  // ---- 4. greedy suppression, one wave width (64 sorted candidates) at a time.
  // A candidate is kept iff no EARLIER KEPT candidate overlaps it by more than the
  // threshold (torchvision's loop).  Per chunk:  (a) all threads test the chunk
  // members against the boxes kept so far (16 threads per member, early exit);  (b) the
  // same threads build, per member j, the 64-bit mask of earlier chunk members i < j with
  // iou(i, j) > thr;  (c) wave 0 walks the chunk in order with scalar bit operations -
  // 3 barriers per 64 candidates instead of one per kept box.  Stops at max_det kept
  // (the reference truncates the full greedy result to its first 300: same rows).
  // (boxes of different classes sit 4096 apart: their x intervals never meet.  Then w = 0, inter = 0 or NaN and the IoU is
  // not above any threshold >= 0 - decided after four LDS reads instead of ten and a division)
  const bool thr_nonneg = a.iou_thres >= 0.0;
  auto overlaps = [&](int i, int j) -> bool {
    const float xx1 = fmaxf(bx1[i], bx1[j]), xx2 = fminf(bx2[i], bx2[j]);
    if (thr_nonneg && !(xx2 > xx1)) return false;
    const float yy1 = fmaxf(by1[i], by1[j]), yy2 = fminf(by2[i], by2[j]);
    const float w = fmaxf(0.0f, __fsub_rn(xx2, xx1)), h = fmaxf(0.0f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area[i], area[j]), inter));  // i = kept box
    return (double)ovr > a.iou_thres;
  };
  const int j = tid & 63, q = tid >> 6;  // chunk member, quarter
  long long t_tests = 0, t_walk = 0, t_bar = 0; int n_chunks = 0;   // debug stamps (thread 0 of workgroup 0, a.trace)
  const bool tr = a.trace && blockIdx.x == 0 && tid == 0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int nk = n_keep;  // stable: written only between the barriers below
    if (nk >= NMS_MAX_DET) break;
    const long long ts0 = tr ? (long long)__builtin_readcyclecounter() : 0;
    const int m = min(64, n - c0);
    unsigned long long bits = 0ull;
    {
      // Wave q tests the chunk's 64 members (one per lane) against kept boxes k = q, q + 16, .. and against chunk members
      // i = 4 q .. 4 q + 3: the OTHER box of every test is the same for all lanes of the wave.  Its x interval therefore
      // comes out of a register by v_readlane (lane t of the wave loaded kept entry q + 16 t; lane i holds member i's own
      // interval) - the first test of a pair costs no LDS access at all.  Boxes of other classes sit 4096 apart and drop
      // out here; only the survivors go through the full test on the LDS arrays.  History: one dependent chain keep[k] ->
      // bx1[keep[k]] -> compare -> break per k (250 cycles each), then 38 broadcast LDS reads per lane: the 16 waves of
      // the workgroup were LDS-issue bound (5.8 k ticks per chunk).
      constexpr int KMAX = (NMS_MAX_DET + NMS_NQ - 1) / NMS_NQ;   // 19
      constexpr int PER = 64 / NMS_NQ;                            // earlier chunk members examined by this wave
      auto lane_f = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
      const int cj = c0 + j;
      const bool live = j < m;
      const float jx1 = live ? bx1[cj] : INFINITY, jx2 = live ? bx2[cj] : -INFINITY;
      const int kl = q + NMS_NQ * j;
      const bool kv = j < KMAX && kl < nk;
      const float rk1 = kv ? kx1[kl] : INFINITY, rk2 = kv ? kx2[kl] : -INFINITY;
      unsigned cand = 0u;
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const float a1 = lane_f(rk1, t), a2 = lane_f(rk2, t);
        if (q + NMS_NQ * t < nk && (!thr_nonneg || fminf(a2, jx2) > fmaxf(a1, jx1))) cand |= 1u << t;
      }
      unsigned icand = 0u;
#pragma unroll
      for (int c = 0; c < PER; ++c) {
        const int i = PER * q + c;
        const float a1 = lane_f(jx1, i), a2 = lane_f(jx2, i);
        if (i < j && (!thr_nonneg || fminf(a2, jx2) > fmaxf(a1, jx1))) icand |= 1u << c;
      }
      if (live) {
        while (cand) {
          const int t = __builtin_ctz(cand);
          cand &= cand - 1u;
          if (overlaps(keep[q + NMS_NQ * t], cj)) { supp[cj] = 1; break; }
        }
        while (icand) {
          const int c = __builtin_ctz(icand);
          icand &= icand - 1u;
          if (overlaps(c0 + PER * q + c, cj)) bits |= 1ull << (PER * q + c);
        }
      }
    }
    pmask[q][j] = bits;
    const long long ts1 = tr ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    const long long ts2 = tr ? (long long)__builtin_readcyclecounter() : 0;
    if (tid < 64) {
      unsigned long long S = 0ull;
#pragma unroll
      for (int qq = 0; qq < NMS_NQ; ++qq) S |= pmask[qq][tid];
      const bool alive = tid < m && !supp[c0 + tid];
      const unsigned long long alive_mask = __ballot(alive);
      // Member j is kept iff it is alive and no KEPT earlier member of the chunk overlaps it: K_j = alive_j & !(S_j & K).
      // The equation has one solution (induction on j); iterating it from K = alive fixes members 0 .. t-1 after t rounds,
      // and a K that reproduces itself IS the solution - a handful of ballots (the depth of the longest suppression chain
      // in the chunk) instead of a 64-step scalar walk (137 ticks per step measured: 25 us of the 98 at 300 kept boxes).
      unsigned long long kmask = alive_mask;
      for (int it = 0; it < 64; ++it) {
        const unsigned long long k2 = __ballot(alive && !(S & kmask));
        if (k2 == kmask) break;
        kmask = k2;
      }
      if ((kmask >> tid) & 1ull) {
        const int pos = nk + __popcll(kmask & ((1ull << tid) - 1ull));
        if (pos < NMS_MAX_DET) { keep[pos] = c0 + tid; kx1[pos] = bx1[c0 + tid]; kx2[pos] = bx2[c0 + tid]; }
      }
      if (tid == 0) n_keep = min(NMS_MAX_DET, nk + __popcll(kmask));
    }
    const long long ts3 = tr ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    if (tr) { const long long ts4 = (long long)__builtin_readcyclecounter(); t_tests += ts1 - ts0; t_walk += ts3 - ts2; t_bar += (ts2 - ts1) + (ts4 - ts3); ++n_chunks; }
  }
  if (tr) { a.trace[9] = n_chunks; a.trace[10] = t_tests; a.trace[11] = t_walk; a.trace[12] = t_bar; }
  __syncthreads();
  NMS_STAMP(5);   // greedy
  const int kept = n_keep;

  // ---- output rows: un-offset box, conf, float(cls); idx = decode row
  for (int k = tid; k < kept; k += NMS_THREADS) {
    const int i = keep[k];
    const unsigned long long kk = key[i];
    const unsigned row = 0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull);
    const float* r = img + (size_t)row * rowlen;
    const float cx = r[0], cy = r[1], hw_ = __fdiv_rn(r[2], 2.0f), hh = __fdiv_rn(r[3], 2.0f);
    float* d = a.dets + ((size_t)b * NMS_MAX_DET + k) * 6;
    d[0] = __fsub_rn(cx, hw_); d[1] = __fsub_rn(cy, hh);
    d[2] = __fadd_rn(cx, hw_); d[3] = __fadd_rn(cy, hh);
    d[4] = __uint_as_float((unsigned)(kk >> 32));
    d[5] = (float)cls_of_row[row];
    a.idx[(size_t)b * NMS_MAX_DET + k] = (int)row;
  }
  if (tid == 0) a.count[b] = kept;
  NMS_STAMP(6);   // output
  if (a.trace && blockIdx.x == 0 && tid == 0) { a.trace[7] = n; a.trace[8] = kept; }
}

void yfv2_launch_nms(const NmsArgs& a, hipStream_t s) {
  const DecodeArgs none{};
  if (a.rows <= NMS_CAP) {
    if (a.compact)
      YFV2_LAUNCH(nms_kernel<1>, dim3(a.B), dim3(NMS_THREADS), 0, s, a, none);
    else
      YFV2_LAUNCH(nms_kernel<0>, dim3(a.B), dim3(NMS_THREADS), 0, s, a, none);
  } else {   // up to 4096 rows: four sort keys per thread (the configuration check admits no more)
    if (a.compact)
      YFV2_LAUNCH((nms_kernel<1, 4>), dim3(a.B), dim3(NMS_THREADS), 0, s, a, none);
    else
      YFV2_LAUNCH((nms_kernel<0, 4>), dim3(a.B), dim3(NMS_THREADS), 0, s, a, none);
  }
}

// handel_preds + non_max_suppression of yfv2_detect as ONE launch: a.boxes is unused, the rows are decoded into LDS
void yfv2_launch_decode_nms(const DecodeArgs& d, const NmsArgs& a, hipStream_t s) {
  // 85 KB of static LDS (sort keys, box arrays, masks) + up to 64 KB of candidate rows: the dynamic part's cap is what
  // is left of the CU's 160 KB, not 160 KB (yfv2_allow_full_lds would be refused)
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
  if (!(done.load() & (1ull << dev))) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, NMS_CAP * 8 * (int)sizeof(float));
    done.fetch_or(1ull << dev);
  }
  YFV2_LAUNCH(nms_kernel<2>, dim3(a.B), dim3(NMS_THREADS), (size_t)a.rows * 8 * sizeof(float), s, a, d);
}

// ============================================================================
// get_batch_statistics (utils/utils.py:194-230) with bbox_iou (:76-108, the "+1 pixel" convention)
// ============================================================================
// One wave per image.  The reference walks the image's detections in order (score-descending, as NMS returned them):
//   stop once every target has been matched; skip a detection whose label is not among the image's target labels;
//   otherwise take the target with the largest IoU over ALL of the image's targets (first one on ties), and count
//   the detection as a true positive if IoU >= iou_threshold and that target has not been matched yet.
// The walk is sequential through the matched set; lanes run over the targets (kept in LDS with their matched flag).
// fp32 arithmetic in the reference's operation order (this file is compiled with -ffp-contract=off); the
// threshold is compared in fp32, as torch does for a float32 tensor against a Python float.
__global__ __launch_bounds__(64) void stats_kernel(StatsArgs a) {
  __shared__ float tx1[STATS_MAX_TARGETS], ty1[STATS_MAX_TARGETS], tx2[STATS_MAX_TARGETS], ty2[STATS_MAX_TARGETS], tlab[STATS_MAX_TARGETS];
  __shared__ int tdone[STATS_MAX_TARGETS];
  const int b = blockIdx.x, lane = threadIdx.x;
  // this image's targets, in their order of appearance (box_index of the reference = position in this list)
  int nt = 0;
  for (int t0 = 0; t0 < a.T; t0 += 64) {
    const int t = t0 + lane;
    const bool mine = t < a.T && a.targets[(size_t)t * 6] == (float)b;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      const int pos = nt + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < STATS_MAX_TARGETS) {
        tlab[pos] = a.targets[(size_t)t * 6 + 1];
        tx1[pos] = a.targets[(size_t)t * 6 + 2]; ty1[pos] = a.targets[(size_t)t * 6 + 3];
        tx2[pos] = a.targets[(size_t)t * 6 + 4]; ty2[pos] = a.targets[(size_t)t * 6 + 5];
        tdone[pos] = 0;
      }
    }
    nt += __popcll(m);
  }
  if (nt > STATS_MAX_TARGETS) { if (lane == 0) *a.overflow = 1; nt = STATS_MAX_TARGETS; }
  __syncthreads();
  const int n = a.count[b];
  int* tp = a.tp + (size_t)b * NMS_MAX_DET;
  for (int i = lane; i < NMS_MAX_DET; i += 64) tp[i] = 0;
  if (nt == 0) return;
  const float thr = a.iou_thres;
  int matched = 0;
  for (int i = 0; i < n && matched < nt; ++i) {
    const float* d = a.dets + ((size_t)b * NMS_MAX_DET + i) * 6;
    const float bx1 = d[0], by1 = d[1], bx2 = d[2], by2 = d[3], lab = d[5];
    const float barea = __fmul_rn(__fadd_rn(__fsub_rn(bx2, bx1), 1.f), __fadd_rn(__fsub_rn(by2, by1), 1.f));
    bool has = false;
    float best = -1.f;
    int bidx = 0x7fffffff;
    for (int t = lane; t < nt; t += 64) {
      has |= tlab[t] == lab;
      const float ix1 = fmaxf(bx1, tx1[t]), iy1 = fmaxf(by1, ty1[t]), ix2 = fminf(bx2, tx2[t]), iy2 = fminf(by2, ty2[t]);
      const float iw = fmaxf(__fadd_rn(__fsub_rn(ix2, ix1), 1.f), 0.f), ih = fmaxf(__fadd_rn(__fsub_rn(iy2, iy1), 1.f), 0.f);
      const float inter = __fmul_rn(iw, ih);
      const float tarea = __fmul_rn(__fadd_rn(__fsub_rn(tx2[t], tx1[t]), 1.f), __fadd_rn(__fsub_rn(ty2[t], ty1[t]), 1.f));
      const float iou = __fdiv_rn(inter, __fadd_rn(__fsub_rn(__fadd_rn(barea, tarea), inter), 1e-16f));
      if (iou > best) { best = iou; bidx = t; }   // ascending t per lane: the first maximum stays
    }
    if (__ballot(has) == 0ull) continue;          // label not among the target labels (wave-uniform)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {      // (max IoU, lowest index) over the wave
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bidx, off);
      if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (best >= thr && tdone[bidx] == 0) {        // wave-uniform
      if (lane == 0) { tp[i] = 1; tdone[bidx] = 1; }
      ++matched;
    }
    __syncthreads();
  }
}

void yfv2_launch_stats(const StatsArgs& a, hipStream_t s) { YFV2_LAUNCH(stats_kernel, dim3(a.B), dim3(64), 0, s, a); }
