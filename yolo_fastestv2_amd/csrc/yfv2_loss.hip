// yfv2_loss.hip - the detector's training loss and its gradient w.r.t. the six logit maps for gfx950.
// Reference behaviour (utils/loss.py): build_target :53-124 (anchor-ratio test against the float64 anchors, the centre
// cell plus up to two neighbour cells per label and anchor), bbox_iou(..., CIoU=True) :8-51 in float64 (the anchors are
// float64, :60-61, so :153's predicted width/height and everything after it promote), compute_loss :130-208
// (BCEWithLogits over every objectness cell with balance 1.0 / 0.4, CrossEntropy over the matched cells' class logits
// divided by the class count, gains 3.2 / 64 / 32).  The backward half is what autograd derives from those lines, with
// CIoU's alpha held constant (:47-48).  SURVEY.md 8(f) row 3, first slice: the loss end of the training path.
//
// Four launches on the caller's stream, no host synchronisation:
//   loss_targets_kernel  one thread per (scale, offset candidate, anchor, label): match slot + objectness target map
//   loss_match_kernel    one thread per slot: CIoU (float64, forward-mode derivatives), class cross-entropy, gradients
//   loss_obj_kernel      every objectness cell: BCE-with-logits sum and gradient
//   loss_final_kernel    one thread: means, balances, gains -> lbox, lobj, lcls, total
#include "yfv2_internal.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// value + gradient w.r.t. the predicted box (px, py, pw, ph), float64
struct D4 {
  double v, g[4];
};
__device__ __forceinline__ D4 cst(double v) { return {v, {0, 0, 0, 0}}; }
__device__ __forceinline__ D4 var(double v, int i) { D4 r = cst(v); r.g[i] = 1.0; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) { D4 r; r.v = a.v / b.v; for (int i = 0; i < 4; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) / b.v; return r; }
__device__ __forceinline__ D4 scale(D4 a, double s) { D4 r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * s; return r; }
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 clamp0(D4 a) { return a.v >= 0.0 ? a : cst(0.0); }
__device__ __forceinline__ D4 datan(D4 a) { D4 r; r.v = atan(a.v); const double d = 1.0 / (1.0 + a.v * a.v); for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] * d; return r; }

// utils/loss.py:8-51, x1y1x2y2=False, CIoU=True; box2 = (tx, ty, tw, th) fp32 values taken as constants
__device__ D4 ciou(double px, double py, double pw, double ph, float tx, float ty, float tw, float th) {
  const D4 X = var(px, 0), Y = var(py, 1), Wd = var(pw, 2), Hd = var(ph, 3);
  const D4 b1x1 = X - scale(Wd, 0.5), b1x2 = X + scale(Wd, 0.5), b1y1 = Y - scale(Hd, 0.5), b1y2 = Y + scale(Hd, 0.5);
  // the target's corners are float32 arithmetic in the reference (box2 is a float32 tensor)
  const float hx = tw / 2, hy = th / 2;
  const D4 b2x1 = cst((double)(tx - hx)), b2x2 = cst((double)(tx + hx)), b2y1 = cst((double)(ty - hy)), b2y2 = cst((double)(ty + hy));
  const D4 inter = clamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * clamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
  const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1;
  const float w2f = (tx + hx) - (tx - hx), h2f = (ty + hy) - (ty - hy);
  const D4 w2 = cst((double)w2f), h2 = cst((double)h2f);
  const D4 uni = (w1 * h1 + cst(1e-16)) + cst((double)(w2f * h2f)) - inter;
  const D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1), ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
  const D4 c2 = cw * cw + ch * ch + cst(1e-16);
  const D4 sx = cst((double)((tx - hx) + (tx + hx))) - (b1x1 + b1x2), sy = cst((double)((ty - hy) + (ty + hy))) - (b1y1 + b1y2);
  const D4 rho2 = scale(sx * sx, 0.25) + scale(sy * sy, 0.25);
  const D4 da = cst((double)atanf(w2f / h2f)) - datan(w1 / h1);     // atan(w2 / h2) is float32 in the reference
  const D4 v = scale(da * da, 4.0 / (M_PI * M_PI));
  const double alpha = v.v / (1.0 - iou.v + v.v);                     // no_grad (:47-48)
  return iou - (rho2 / c2 + scale(v, alpha));
}

}  // namespace

__global__ __launch_bounds__(256) void loss_targets_kernel(LossArgs a) {
  const int per_scale = 5 * 3 * a.T;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * per_scale) return;
  const int l = i / per_scale, r = i - l * per_scale;
  const int k = r / (3 * a.T), an = (r / a.T) % 3, n = r % a.T;
  LossMatch m{};
  const float* t = a.targets + (size_t)n * 6;
  const int H = a.fh[l], W = a.fw[l];
  const float gx = t[2] * (float)W, gy = t[3] * (float)H, gw = t[4] * (float)W, gh = t[5] * (float)H;   // targets * gain, fp32 (:88)
  const double aw = a.anchors[(l * 3 + an) * 2 + 0] / a.stride[l], ah = a.anchors[(l * 3 + an) * 2 + 1] / a.stride[l];
  const double rw = (double)gw / aw, rh = (double)gh / ah;
  bool ok = fmax(fmax(rw, 1.0 / rw), fmax(rh, 1.0 / rh)) < 2.0;       // :93-94 (float64)
  float ox = 0.f, oy = 0.f;
  const float ix = (float)W - gx, iy = (float)H - gy;
  if (k == 1) { ok = ok && (fmodf(gx, 1.0f) < 0.5f) && gx > 1.0f; ox = 0.5f; }
  if (k == 2) { ok = ok && (fmodf(gy, 1.0f) < 0.5f) && gy > 1.0f; oy = 0.5f; }
  if (k == 3) { ok = ok && (fmodf(ix, 1.0f) < 0.5f) && ix > 1.0f; ox = -0.5f; }
  if (k == 4) { ok = ok && (fmodf(iy, 1.0f) < 0.5f) && iy > 1.0f; oy = -0.5f; }
  const int b = (int)t[0];
  const int lab = (int)t[1];
  // a label whose image index or class is out of range is skipped (the reference's CrossEntropyLoss raises on such a class;
  // here it must not become an out-of-bounds read of the class logits - precondition stated in include/yfv2.h)
  if (ok && b >= 0 && b < a.B && lab >= 0 && lab < a.classes) {
    const int cx = (int)(gx - ox), cy = (int)(gy - oy);                // .long(): truncation (:112)
    m.valid = 1; m.b = b; m.a = an; m.cls = (int)t[1];
    m.gi = min(max(cx, 0), W - 1); m.gj = min(max(cy, 0), H - 1);      // :119 (the clamp the reference needs int() bounds for)
    // :115-120: gi, gj are views of gij, so :119's in-place clamp_ reaches gij before `gxy - gij`: the CLAMPED cell
    m.tb[0] = gx - (float)m.gi; m.tb[1] = gy - (float)m.gj; m.tb[2] = gw; m.tb[3] = gh;
    m.aw = aw; m.ah = ah;
    atomicAdd(&a.nb[l], 1);
    a.tobj[l][((size_t)(b * 3 + an) * H + m.gj) * W + m.gi] = 1;
  }
  a.matches[i] = m;
}

__global__ __launch_bounds__(128) void loss_match_kernel(LossArgs a) {
  const int per_scale = 5 * 3 * a.T;
  const int i = blockIdx.x * 128 + threadIdx.x;
  double box_term = 0.0, cls_term = 0.0;
  int l = 0;
  if (i < 2 * per_scale) {
    l = i / per_scale;
    const LossMatch m = a.matches[i];
    if (m.valid) {
      const int H = a.fh[l], W = a.fw[l], HW = H * W;
      const size_t cell = (size_t)m.gj * W + m.gi;
      const float* reg = a.reg[l] + ((size_t)m.b * 12 + m.a * 4) * HW + cell;
      float s[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] = sigmoidf_(reg[(size_t)c * HW]);
      const float pxf = s[0] * 2.0f - 0.5f, pyf = s[1] * 2.0f - 0.5f;   // :152 fp32
      const float qw = s[2] * 2.0f, qh = s[3] * 2.0f;
      const double pw = (double)(qw * qw) * m.aw, ph = (double)(qh * qh) * m.ah;   // :153: fp32 square, float64 anchor product
      const D4 c = ciou((double)pxf, (double)pyf, pw, ph, m.tb[0], m.tb[1], m.tb[2], m.tb[3]);
      box_term = 1.0 - c.v;
      const int nb = a.nb[l];
      if (a.grad_reg[l]) {
        // d(3.2 * mean(1 - ciou)) / d logit = -3.2 / nb * dciou/dbox * dbox/dlogit
        const double k = -3.2 / (double)nb;
        const double ds[4] = {2.0 * s[0] * (1.0 - s[0]), 2.0 * s[1] * (1.0 - s[1]),
                              m.aw * 8.0 * (double)s[2] * s[2] * (1.0 - s[2]), m.ah * 8.0 * (double)s[3] * s[3] * (1.0 - s[3])};
        float* g = a.grad_reg[l] + ((size_t)m.b * 12 + m.a * 4) * HW + cell;
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(g + (size_t)q * HW, (float)(k * c.g[q] * ds[q]));
      }
      if (a.classes > 1) {
        const float* cl = a.cls[l] + (size_t)m.b * a.classes * HW + cell;
        float mx = -INFINITY;
        for (int q = 0; q < a.classes; ++q) mx = fmaxf(mx, cl[(size_t)q * HW]);
        float sum = 0.f;
        for (int q = 0; q < a.classes; ++q) sum += expf(cl[(size_t)q * HW] - mx);
        const float lse = mx + logf(sum);
        cls_term = (double)(lse - cl[(size_t)m.cls * HW]);
        if (a.grad_cls[l]) {
          const float k = 32.0f / ((float)a.classes * (float)nb);
          float* g = a.grad_cls[l] + (size_t)m.b * a.classes * HW + cell;
          for (int q = 0; q < a.classes; ++q) {
            const float pq = expf(cl[(size_t)q * HW] - lse);
            atomicAdd(g + (size_t)q * HW, k * (pq - (q == m.cls ? 1.0f : 0.0f)));
          }
        }
      }
    }
  }
  // block sums (a block never straddles the two scales' slot ranges unless per_scale % 128 != 0: add per scale)
  __shared__ double sb[2][2];
  if (threadIdx.x < 4) sb[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  __syncthreads();
  if (box_term != 0.0) atomicAdd(&sb[l][0], box_term);
  if (cls_term != 0.0) atomicAdd(&sb[l][1], cls_term);
  __syncthreads();
  if (threadIdx.x < 4) {
    const double v = sb[threadIdx.x >> 1][threadIdx.x & 1];
    if (v != 0.0) atomicAdd(&a.sums[(threadIdx.x >> 1) * 3 + (threadIdx.x & 1 ? 2 : 0)], v);
  }
}

__global__ __launch_bounds__(256) void loss_obj_kernel(LossArgs a) {
  // objectness BCE-with-logits over every cell of both maps (pos_weight = 1: (1 - t) x + log1p(exp(-|x|)) + max(-x, 0))
  const int n0 = a.B * 3 * a.fh[0] * a.fw[0], n1 = a.B * 3 * a.fh[1] * a.fw[1];
  double acc[2] = {0.0, 0.0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += gridDim.x * 256) {
    const int l = i >= n0 ? 1 : 0, j = l ? i - n0 : i;
    const float x = a.obj[l][j], t = a.tobj[l][j] ? 1.0f : 0.0f;
    acc[l] += (double)((1.0f - t) * x + (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f)));
    if (a.grad_obj[l]) a.grad_obj[l][j] = (64.0f * (l ? 0.4f : 1.0f) / (float)(l ? n1 : n0)) * (sigmoidf_(x) - t);
  }
  __shared__ double sb[2];
  if (threadIdx.x < 2) sb[threadIdx.x] = 0.0;
  __syncthreads();
  // wave reduction first
  for (int l = 0; l < 2; ++l) {
    double v = acc[l];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v != 0.0) atomicAdd(&sb[l], v);
  }
  __syncthreads();
  if (threadIdx.x < 2 && sb[threadIdx.x] != 0.0) atomicAdd(&a.sums[threadIdx.x * 3 + 1], sb[threadIdx.x]);
}

__global__ void loss_final_kernel(LossArgs a) {
  if (threadIdx.x || blockIdx.x) return;
  float lbox = 0.f, lobj = 0.f, lcls = 0.f;
  for (int l = 0; l < 2; ++l) {
    const int nb = a.nb[l];
    const double ncell = (double)a.B * 3 * a.fh[l] * a.fw[l];
    if (nb) {
      lbox = (float)((double)lbox + a.sums[l * 3 + 0] / (double)nb);             // float32 accumulator += float64 mean (:156)
      if (a.classes > 1) lcls += (float)(a.sums[l * 3 + 2] / (double)nb) / (float)a.classes;
    }
    lobj += (float)(a.sums[l * 3 + 1] / ncell) * (l ? 0.4f : 1.0f);
  }
  lbox *= 3.2f; lobj *= 64.0f; lcls *= 32.0f;
  a.losses[0] = lbox; a.losses[1] = lobj; a.losses[2] = lcls; a.losses[3] = lbox + lobj + lcls;
}

void yfv2_launch_loss(const LossArgs& a, hipStream_t s) {
  const int slots = 2 * 5 * 3 * a.T;
  if (slots > 0) {
    hipLaunchKernelGGL(loss_targets_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_match_kernel, dim3((slots + 127) / 128), dim3(128), 0, s, a);
  }
  const int ncell = a.B * 3 * (a.fh[0] * a.fw[0] + a.fh[1] * a.fw[1]);
  int blocks = (ncell + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(loss_obj_kernel, dim3(blocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, a);
}
