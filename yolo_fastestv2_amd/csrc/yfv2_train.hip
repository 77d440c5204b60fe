// yfv2_train.hip - SURVEY.md 8(f) row 3, the rest of the training path: one iteration of train.py:96-123 on the device.
//   yfv2_train_forward   Detector.forward in train() mode (model/backbone/shufflenetv2.py:19-63,74-80,102-109, model/fpn.py,
//                        model/detector.py:21-47 with every nn.BatchNorm2d on BATCH statistics, running statistics moved by
//                        momentum 0.1 with the unbiased variance)
//   yfv2_train_backward  what total_loss.backward() (train.py:110) derives for the 225 parameters from the gradient of the
//                        loss w.r.t. the six logit maps (yfv2_loss)
//   yfv2_sgd_step        torch.optim.SGD(momentum, weight_decay, dampening 0, no Nesterov) as train.py:81-85 builds it
// Behaviour only was read from the reference; autograd's derivatives are restated from the layer definitions.
//
// This is the correctness-first slice of the row: plain NCHW fp32 kernels, one thread per output element (reductions: one
// block per channel / filter entry, float64 accumulators), a tape of closures instead of an autograd engine.  It is NOT the
// throughput path (BASELINE.json's metric is inference); it exists so that the reference's training loop runs end to end
// on the MI355X through this library and matches tests/golden/golden_train.npz (the reference's own modules on CPU):
// gradients to 1e-5 of each tensor's largest entry, updated weights to 1e-6.  Parameters, gradients and BatchNorm buffers are
// the CALLER's device tensors in the reference's own layouts ((co, ci, k, k), (c,)): nothing is copied or re-laid-out.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/yfv2.h"
#include "yfv2_internal.h"

namespace {

// a (B, C, H, W) window into an NCHW tensor with Ctot channels: channel c of the view = channel coff + c * cstride of the tensor
struct V {
  float* p; int Ctot, coff, cstride, C, H, W;
  __host__ __device__ size_t at(int b, int c, int y, int x) const { return (((size_t)b * Ctot + coff + (size_t)c * cstride) * H + y) * W + x; }
};

struct ConvP { V in, out; const float* w; const float* bias; int k, stride, pad, dw, B; };


// Depthwise convs (3 x 3 and 5 x 5, stride 1 or 2; pad K / 2), forward and data gradient: a block stays inside ONE (image, channel)
// plane - the filter and the plane bases are wave-uniform, a thread divides once - with the generic kernels' summation order
// (row, then column; padded taps skipped), so the results are theirs bit for bit.  grid (position chunks of 256, B * C planes)
template <int K, int S>
__global__ __launch_bounds__(256) void dw_fwd_kernel(ConvP a) {
  constexpr int P = K / 2;
  const int OH = a.out.H, OW = a.out.W, IH = a.in.H, IW = a.in.W, C = a.out.C;
  const int pl = blockIdx.y, b = pl / C, c = pl - b * C;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= OH * OW) return;
  const int oy = o / OW, ox = o - oy * OW;
  const float* ip = a.in.p + a.in.at(b, c, 0, 0);
  const float* w = a.w + c * K * K;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * S - P + ky;
    if (iy < 0 || iy >= IH) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * S - P + kx;
      if (ix >= 0 && ix < IW) acc = __builtin_fmaf(w[ky * K + kx], ip[iy * IW + ix], acc);
    }
  }
  a.out.p[a.out.at(b, c, 0, 0) + o] = acc;
}
template <int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(ConvP a) {   // a.in = gradient view of the input (accumulated into), a.out = gradient of the output
  constexpr int P = K / 2;
  const int OH = a.out.H, OW = a.out.W, IH = a.in.H, IW = a.in.W, C = a.in.C;
  const int pl = blockIdx.y, b = pl / C, c = pl - b * C;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= IH * IW) return;
  const int iy = i / IW, ix = i - iy * IW;
  const float* gp = a.out.p + a.out.at(b, c, 0, 0);
  const float* w = a.w + c * K * K;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ny = iy + P - ky;
    if (ny < 0 || (S > 1 && (ny % S))) continue;
    const int oy = ny / S;
    if (oy >= OH) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int nx = ix + P - kx;
      if (nx < 0 || (S > 1 && (nx % S))) continue;
      const int ox = nx / S;
      if (ox >= OW) continue;
      acc = __builtin_fmaf(w[ky * K + kx], gp[oy * OW + ox], acc);
    }
  }
  a.in.p[a.in.at(b, c, 0, 0) + i] += acc;
}
// The stem conv (3 x 3, stride 2, pad 1, 3 -> 24 channels): conv_fwd_kernel read its 27 inputs once per OUTPUT CHANNEL (0.96 ms at
// batch 64); here a thread owns an output pixel, reads them once and keeps the 24 sums in registers - the same fmaf chain per
// channel (input channel, then row, then column; padded taps skipped), the filter through the scalar cache.
__global__ __launch_bounds__(256) void stem_fwd_kernel(ConvP a) {
  const int OH = a.out.H, OW = a.out.W, IH = a.in.H, IW = a.in.W;
  const size_t n = (size_t)a.B * OH * OW, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ox = i % OW, oy = (i / OW) % OH, b = i / ((size_t)OW * OH);
  float acc[24];
#pragma unroll
  for (int co = 0; co < 24; ++co) acc[co] = 0.f;
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= IH) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= IW) continue;
        const float v = a.in.p[a.in.at(b, ci, iy, ix)];
#pragma unroll
        for (int co = 0; co < 24; ++co) acc[co] = __builtin_fmaf(a.w[((co * 3 + ci) * 3 + ky) * 3 + kx], v, acc[co]);
      }
    }
#pragma unroll
  for (int co = 0; co < 24; ++co) a.out.p[a.out.at(b, co, oy, ox)] = acc[co];
}


// segments for a reduction of n elements next to `others` independent blocks: about 2048 blocks in all, at least 2048 elements each
inline unsigned reduce_segments(size_t n, size_t others) {
  size_t s = others >= 2048 ? 1 : 2048 / (others ? others : 1);
  if (s > 64) s = 64;
  if (s > n / 2048) s = n / 2048;
  return (unsigned)(s < 1 ? 1 : s);
}

__global__ __launch_bounds__(256) void bias_bwd_kernel(V g, int B, float* db) {
  __shared__ double red[256];
  const int co = blockIdx.x, n = B * g.H * g.W;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += (double)g.p[g.at(i / (g.H * g.W), co, (i / g.W) % g.H, i % g.W)];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) db[co] += (float)red[0];
}

// BatchNorm2d, training mode: per-channel batch mean / biased variance of y (B, C, H, W contiguous); saves mean and
// 1/sqrt(var + eps); running = 0.9 running + 0.1 (mean | unbiased variance)
// (All four kernels walk (plane, position) incrementally - no division per element - and the two elementwise ones take 1024
// elements per block; round 3's forms divided four times per element and spent a launch per layer on the per-channel finish.)
// per channel and segment: sum and sum of squares of y in double, into acc[4 c + 0, 1] (zeroed per forward)
__global__ __launch_bounds__(256) void bn_stats_part_kernel(const float* y, int B, int C, int HW, double* acc) {
  __shared__ double r0[256], r1[256];
  const int c = blockIdx.x, n = B * HW, nseg = gridDim.y, len = (n + nseg - 1) / nseg;
  const int i0 = blockIdx.y * len, i1 = i0 + len < n ? i0 + len : n;
  double a0 = 0.0, a1 = 0.0;
  int i = i0 + threadIdx.x;
  if (i < i1) {
    int b = i / HW, r = i - b * HW;
    const size_t pstride = (size_t)C * HW;
    const float* p = y + ((size_t)b * C + c) * HW;
    for (; i < i1; i += 256) {
      const double v = (double)p[r];
      a0 += v; a1 += v * v;
      r += 256;
      while (r >= HW) { r -= HW; p += pstride; }
    }
  }
  r0[threadIdx.x] = a0; r1[threadIdx.x] = a1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; } __syncthreads(); }
  if (threadIdx.x == 0) { atomicAdd(&acc[4 * c + 0], r0[0]); atomicAdd(&acc[4 * c + 1], r1[0]); }
}
// mean, 1 / sqrt(biased variance + eps) (variance = E[y^2] - mean^2 in double: the inputs are fp32) and the running statistics are
// finished inside the apply kernel: a block of BN_EPB consecutive NCHW elements touches a few (image, channel) planes, its first
// threads finish those planes' statistics into LDS; the block that holds a channel's FIRST element (image 0) also stores
// mean / invstd for the backward pass and moves the running statistics - exactly once per channel and forward.
constexpr int BN_EPB = 1024, BN_CH = BN_EPB + 2;   // elements per block; planes a block can touch when H * W == 1 (+ slack)
struct BnWalk {     // (plane, position) of element i0 + tid + 256 j, advanced without divisions
  size_t pl; int k, c, b, r;
  __device__ BnWalk(size_t i, size_t ch0, int HW, int C) { pl = i / HW; r = (int)(i - pl * HW); k = (int)(pl - ch0); c = (int)(pl % C); b = (int)(pl / C); }
  __device__ void step(int HW, int C) { r += 256; while (r >= HW) { r -= HW; ++pl; ++k; if (++c == C) { c = 0; ++b; } } }
};
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* y, V z, int B, const double* acc, float* mean, float* invstd, float* run_mean, float* run_var,
                                                       const float* gamma, const float* beta, int relu) {
  __shared__ float s_mean[BN_CH], s_inv[BN_CH], s_gamma[BN_CH], s_beta[BN_CH];
  const int HW = z.H * z.W, C = z.C;
  const size_t n = (size_t)B * C * HW, i0 = (size_t)blockIdx.x * BN_EPB;
  const size_t ilast = i0 + BN_EPB - 1 < n ? i0 + BN_EPB - 1 : n - 1;
  const size_t ch0 = i0 / HW, nch = ilast / HW - ch0 + 1;               // (image, channel) planes this block touches
  const int nb = B * HW;
  for (size_t k = threadIdx.x; k < nch; k += 256) {
    const int c = (int)((ch0 + k) % C);
    const double m = acc[4 * c] / nb;
    double var = acc[4 * c + 1] / nb - m * m;
    if (var < 0.0) var = 0.0;
    const float fm = (float)m, fi = (float)(1.0 / sqrt(var + 1e-5));
    s_mean[k] = fm; s_inv[k] = fi; s_gamma[k] = gamma[c]; s_beta[k] = beta[c];
    if (ch0 + k < (size_t)C) {                                            // this plane belongs to image 0: the channel's one writer
      const size_t first = (ch0 + k) * HW;
      if (first >= i0 && first <= ilast) {
        mean[c] = fm; invstd[c] = fi;
        run_mean[c] = (float)(0.9 * (double)run_mean[c] + 0.1 * m);
        run_var[c] = (float)(0.9 * (double)run_var[c] + 0.1 * (nb > 1 ? var * nb / (nb - 1) : var));
      }
    }
  }
  __syncthreads();
  size_t i = i0 + threadIdx.x;
  if (i > ilast) return;
  BnWalk w(i, ch0, HW, C);
  for (; i <= ilast; i += 256) {
    float v = (y[i] - s_mean[w.k]) * s_inv[w.k] * s_gamma[w.k] + s_beta[w.k];
    if (relu && !(v > 0.f)) v = 0.f;
    z.p[z.at(w.b, w.c, 0, 0) + w.r] = v;
    w.step(HW, C);
  }
}

// per channel: sum of dz' and of dz' * xhat (dz' = dz where the ReLU let the value through); -> d beta, d gamma, and the two
// sums for bn_bwd_apply
__global__ __launch_bounds__(256) void bn_bwd_part_kernel(const float* y, V z, V dz, int B, const float* mean, const float* invstd, int relu, double* acc) {
  __shared__ double r0[256], r1[256];
  const int c = blockIdx.x, HW = z.H * z.W, n = B * HW, nseg = gridDim.y, len = (n + nseg - 1) / nseg;
  const int i0 = blockIdx.y * len, i1 = i0 + len < n ? i0 + len : n;
  double a0 = 0.0, a1 = 0.0;
  int i = i0 + threadIdx.x;
  if (i < i1) {
    int b = i / HW, r = i - b * HW;
    const float mc = mean[c], ic = invstd[c];
    for (; i < i1; i += 256) {
      float g = dz.p[dz.at(b, c, 0, 0) + r];
      if (relu && !(z.p[z.at(b, c, 0, 0) + r] > 0.f)) g = 0.f;
      const float xh = (y[((size_t)b * z.C + c) * HW + r] - mc) * ic;
      a0 += (double)g; a1 += (double)g * (double)xh;
      r += 256;
      while (r >= HW) { r -= HW; ++b; }
    }
  }
  r0[threadIdx.x] = a0; r1[threadIdx.x] = a1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; } __syncthreads(); }
  if (threadIdx.x == 0) { atomicAdd(&acc[4 * c + 2], r0[0]); atomicAdd(&acc[4 * c + 3], r1[0]); }
}
// dy = gamma * invstd * (dz' - sum(dz') / N - xhat * sum(dz' xhat) / N)   (dy is the conv output's gradient: single consumer, overwritten)
// The two sums arrive in double (bn_bwd_part_kernel); a block converts those of the planes it touches (as bn_apply_kernel does
// with the statistics), and the block that holds a channel's first element adds them to d beta / d gamma - once per channel.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* y, V z, V dz, float* dy, int B, const float* mean, const float* invstd, const float* gamma,
                                                          int relu, const double* acc, float* dgamma, float* dbeta) {
  __shared__ float s_mean[BN_CH], s_inv[BN_CH], s_gamma[BN_CH], s_s0[BN_CH], s_s1[BN_CH];
  const int HW = z.H * z.W, C = z.C;
  const size_t n = (size_t)B * C * HW, i0 = (size_t)blockIdx.x * BN_EPB;
  const size_t ilast = i0 + BN_EPB - 1 < n ? i0 + BN_EPB - 1 : n - 1;
  const size_t ch0 = i0 / HW, nch = ilast / HW - ch0 + 1;
  for (size_t k = threadIdx.x; k < nch; k += 256) {
    const int c = (int)((ch0 + k) % C);
    const float s0 = (float)acc[4 * c + 2], s1 = (float)acc[4 * c + 3];
    s_mean[k] = mean[c]; s_inv[k] = invstd[c]; s_gamma[k] = gamma[c]; s_s0[k] = s0; s_s1[k] = s1;
    if (ch0 + k < (size_t)C) {
      const size_t first = (ch0 + k) * HW;
      if (first >= i0 && first <= ilast) { dbeta[c] += s0; dgamma[c] += s1; }
    }
  }
  __syncthreads();
  size_t i = i0 + threadIdx.x;
  if (i > ilast) return;
  const float inv_n = 1.0f / (float)(B * HW);
  BnWalk w(i, ch0, HW, C);
  for (; i <= ilast; i += 256) {
    float g = dz.p[dz.at(w.b, w.c, 0, 0) + w.r];
    if (relu && !(z.p[z.at(w.b, w.c, 0, 0) + w.r] > 0.f)) g = 0.f;
    const float xh = (y[i] - s_mean[w.k]) * s_inv[w.k];
    dy[i] = s_gamma[w.k] * s_inv[w.k] * (g - s_s0[w.k] * inv_n - xh * s_s1[w.k] * inv_n);
    w.step(HW, C);
  }
}

// max_pool2d(3, 2, 1) as ATen's CPU kernel scans it: window rows then columns, a later value replaces the maximum only if it is
// greater - the FIRST maximum keeps the gradient
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, float* out, int* arg, int B, int C, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const size_t n = (size_t)B * C * OH * OW, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ox = i % OW, oy = (i / OW) % OH;
  const size_t plane = i / ((size_t)OW * OH);
  const float* xp = x + plane * H * W;
  float best = -INFINITY; int bi = -1;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const float v = xp[iy * W + ix];
      if (v > best || bi < 0) { best = v; bi = iy * W + ix; }
    }
  }
  out[i] = best; arg[i] = bi;
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dout, const int* arg, float* dx, int B, int C, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const size_t n = (size_t)B * C * H * W, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ix = i % W, iy = (i / W) % H;
  const size_t plane = i / ((size_t)W * H);
  float acc = 0.f;
  for (int oy = iy / 2; oy <= (iy + 1) / 2 && oy < OH; ++oy)
    for (int ox = ix / 2; ox <= (ix + 1) / 2 && ox < OW; ++ox)
      if (arg[plane * OH * OW + oy * OW + ox] == iy * W + ix) acc += dout[plane * OH * OW + oy * OW + ox];
  dx[i] = acc;   // (the stem's ReLU output has this one consumer; the arena was zeroed)
}

// dst view (+)= src view, same (C, H, W)
__global__ __launch_bounds__(256) void view_copy_kernel(V dst, V src, int B, int accumulate) {
  const size_t n = (size_t)B * dst.C * dst.H * dst.W, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = i % dst.W, y = (i / dst.W) % dst.H, c = (i / ((size_t)dst.W * dst.H)) % dst.C, b = i / ((size_t)dst.W * dst.H * dst.C);
  const float v = src.p[src.at(b, c, y, x)];
  if (accumulate) dst.p[dst.at(b, c, y, x)] += v; else dst.p[dst.at(b, c, y, x)] = v;
}
// fpn.py:57-58: dst (2H x 2W) = nearest x2 of src; backward: d src += its four children
__global__ __launch_bounds__(256) void upsample2_kernel(V dst, V src, int B) {
  const size_t n = (size_t)B * dst.C * dst.H * dst.W, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = i % dst.W, y = (i / dst.W) % dst.H, c = (i / ((size_t)dst.W * dst.H)) % dst.C, b = i / ((size_t)dst.W * dst.H * dst.C);
  dst.p[dst.at(b, c, y, x)] = src.p[src.at(b, c, y >> 1, x >> 1)];
}
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(V dsrc, V ddst, int B) {
  const size_t n = (size_t)B * dsrc.C * dsrc.H * dsrc.W, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = i % dsrc.W, y = (i / dsrc.W) % dsrc.H, c = (i / ((size_t)dsrc.W * dsrc.H)) % dsrc.C, b = i / ((size_t)dsrc.W * dsrc.H * dsrc.C);
  dsrc.p[dsrc.at(b, c, y, x)] += ddst.p[ddst.at(b, c, 2 * y, 2 * x)] + ddst.p[ddst.at(b, c, 2 * y, 2 * x + 1)] + ddst.p[ddst.at(b, c, 2 * y + 1, 2 * x)] +
                                 ddst.p[ddst.at(b, c, 2 * y + 1, 2 * x + 1)];
}



// ---- round 4: the pointwise (1x1, stride 1) convolutions - 51 of the 79 convs and 73 % of the step's arithmetic - as GEMMs on
// the fp32 matrix instruction (v_mfma_f32_16x16x4_f32), straight on the NCHW views (channel planes are contiguous pixel runs).
// Round 3 ran them on conv_fwd / conv_bwd_data / conv_bwd_weight_kernel above: one thread per output element looping over the
// channels, one BLOCK per filter entry for the weight gradient (every block re-read the whole activation and gradient: 250-300 us
// per layer whatever its size) - 13.6 + 21.2 ms of the 42 ms iteration at batch 64 (rocprofv3, tools/train_probe.py).
typedef float yfv2_f4 __attribute__((ext_vector_type(4)));

// out[b][r][p] (+)= sum_k Wm(r, k) in[b][k][p] (+ bias[r]);  Wm(r, k) = TRANS ? w[k * wld + r] : w[r * wld + k]
//   forward:        in = x,     out = y,      w = (Cout, Cin), wld = Cin
//   data gradient:  in = dy,    out = dx (+=), TRANS, w = the same (Cout, Cin) array, wld = Cin: dx[ci] += sum_co w[co][ci] dy[co]
// A wave = one image x 64 pixels x up to four 16-row tiles of output channels; lane (l, g): B operand = in[k0 + g][p0 + 16 e + l]
// for the four pixel tiles e (four coalesced dword loads), A operand = Wm(16 t + l, k0 + g); D lane (l, g) reg r = out row
// 16 t + 4 g + r, pixel p0 + 16 e + l.
template <bool TRANS, bool ACCUM, int MT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(V in, V out, const float* __restrict__ w, const float* __restrict__ bias, int wld, int B) {
  constexpr int KS = 4;                            // MFMA steps (4 input channels each) per round of loads
  const int HW = out.H * out.W, K = in.C, M = out.C;
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const int p0 = (blockIdx.x * 4 + wave) * 64, b = blockIdx.y, r0 = blockIdx.z * (16 * MT);
  if (p0 >= HW) return;
  yfv2_f4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = (yfv2_f4){0.f, 0.f, 0.f, 0.f};
  const float* inb = in.p + ((size_t)b * in.Ctot + in.coff) * HW;
  const size_t kstride = (size_t)in.cstride * HW;
  // A round = the operands of KS MFMA steps (16 input channels), requested before the MFMAs of the round before it run.  The
  // first form fetched ONE step (4 channels) ahead: a layer was K / 4 dependent round trips to L2 with sixteen MFMAs each behind
  // them - 6 .. 72 of them, the launch's whole duration (the 51 pointwise convs of an iteration: 4.7 ms of its 12.3).  The
  // MFMAs run in the same order on the same operands: the results are bit-identical to that form.
  auto fetch = [&](int k0, float (&bv)[KS][4], float (&av)[KS][MT]) {
    // Every load is unconditional, from a clamped address, and what must not count is removed by an AND with a lane mask on the
    // FILTER operand only (a channel past K: filter entry 0 times a real, finite activation; rows past M and pixels past HW are
    // never stored).  A load behind a lane predicate - or a select the compiler turns back into one - is a branch, and the
    // compiler then waits for every load where it stands: the first version of this round waited twice per pair of loads.
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = k0 + 4 * j + g;
      const int kc = k < K ? k : K - 1;
      const unsigned km = k < K ? 0xffffffffu : 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int p = p0 + 16 * e + l; bv[j][e] = inb[(size_t)kc * kstride + (p < HW ? p : HW - 1)]; }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int r = r0 + 16 * t + l, rc = r < M ? r : M - 1;
        const float v = TRANS ? w[(size_t)kc * wld + rc] : w[(size_t)rc * wld + kc];
        av[j][t] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & km);
      }
    }
  };
  auto mac = [&](int k0, const float (&bv)[KS][4], const float (&av)[KS][MT]) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (k0 + 4 * j >= K) break;                  // (wave-uniform: a step of zeros adds nothing, but costs its MFMAs)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], bv[j][e], acc[t][e], 0, 0, 0);
    }
  };
  float b0[KS][4], a0[KS][MT], b1[KS][4], a1[KS][MT];
  fetch(0, b0, a0);
  for (int k0 = 0; k0 < K; k0 += 8 * KS) {
    fetch(k0 + 4 * KS, b1, a1);
    __builtin_amdgcn_sched_barrier(0);
    mac(k0, b0, a0);
    if (k0 + 4 * KS >= K) break;
    fetch(k0 + 8 * KS, b0, a0);
    __builtin_amdgcn_sched_barrier(0);
    mac(k0 + 4 * KS, b1, a1);
  }
  float* outb = out.p + ((size_t)b * out.Ctot + out.coff) * HW;
  const size_t rstride = (size_t)out.cstride * HW;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = r0 + 16 * t + 4 * g + rr;
      if (r >= M) continue;
      const float bi = bias ? bias[r] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = p0 + 16 * e + l;
        if (p < HW) { float* d = outb + (size_t)r * rstride + p; *d = ACCUM ? *d + acc[t][e][rr] : acc[t][e][rr] + bi; }
      }
    }
}

// weight gradient: scratch[co * Cin + ci] += sum over one image's pixel segment of dy[b][co][p] x[b][ci][p] (double atomics: the
// fp32 accumulation runs over at most `seg` pixels).  A wave = up to 3 x 3 tiles of 16 co x 16 ci x `upw` consecutive (image, segment)
// UNITS (round 5): each unit's fp32 sums are added to DOUBLE registers and the wave issues ONE set of atomics for all of them - a
// layer's ~2000 waves used to send ~2000 atomics to every filter entry (rocprofv3: 33 us per layer for ~5 us of matrix-core work; a
// sweep of the wave count alone gave 11.1 -> 10.6 ms per iteration at the price of four times longer fp32 sums - this form keeps
// the sums at `seg` pixels).  Per 16 pixels lane (l, g) holds pixels q + 4 g .. + 3 of channel row l of every tile: MFMA step e
// contracts pixel q + 4 g + e.
__global__ __launch_bounds__(64) void pw_wgrad_kernel(V x, V dy, double* __restrict__ scratch, int seg, int ci_blocks, int nseg, int units, int upw) {
  constexpr int T = 3;
  const int HW = x.H * x.W, Cin = x.C, Cout = dy.C;
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const int co0 = (blockIdx.z / ci_blocks) * (16 * T), ci0 = (blockIdx.z % ci_blocks) * (16 * T);
  const size_t xs = (size_t)x.cstride * HW, gs = (size_t)dy.cstride * HW;
  double dacc[T][T][4];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dacc[i][j][rr] = 0.0;
#pragma unroll 1
  for (int u = blockIdx.x * upw; u < min(units, (int)(blockIdx.x + 1) * upw); ++u) {
  const int b = u / nseg, q0 = (u - b * nseg) * seg, q1 = min(HW, q0 + seg);
  yfv2_f4 acc[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j) acc[i][j] = (yfv2_f4){0.f, 0.f, 0.f, 0.f};
  const float* xb = x.p + ((size_t)b * x.Ctot + x.coff) * HW;
  const float* gb = dy.p + ((size_t)b * dy.Ctot + dy.coff) * HW;
  auto fetch = [&](int q, float (&av)[T][4], float (&bv)[T][4]) {
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const int co = co0 + 16 * i + l, ci = ci0 + 16 * i + l;
      const int coc = co < Cout ? co : Cout - 1, cic = ci < Cin ? ci : Cin - 1;   // (unconditional loads, see pw_gemm_kernel: rows past
                                                                                 // Cout / Cin are never stored; a pixel past the segment
                                                                                 // gets a zero dy against a real, finite x)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int p = q + 4 * g + e, pc = p < q1 ? p : q1 - 1;
        const unsigned pm = p < q1 ? 0xffffffffu : 0u;
        av[i][e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, gb[(size_t)coc * gs + pc]) & pm);
        bv[i][e] = xb[(size_t)cic * xs + pc];
      }
    }
  };
  auto mac = [&](const float (&av)[T][4], const float (&bv)[T][4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
  };
  float a0[T][4], b0[T][4], a1[T][4], b1[T][4];
  fetch(q0, a0, b0);
  for (int q = q0; q < q1; q += 32) {
    fetch(q + 16, a1, b1);                        // (past q1: zeros)
    __builtin_amdgcn_sched_barrier(0);
    mac(a0, b0);
    if (q + 16 >= q1) break;
    fetch(q + 32, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mac(a1, b1);
  }
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) dacc[i][j][rr] += (double)acc[i][j][rr];
  }
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = co0 + 16 * i + 4 * g + rr, ci = ci0 + 16 * j + l;
        if (co < Cout && ci < Cin) atomicAdd(&scratch[(size_t)co * Cin + ci], dacc[i][j][rr]);
      }
}
// Every layer's weight gradient is summed in double in its own range of one scratch (zeroed per backward) and added to the bound
// gradient tensors by ONE launch at the end of the backward pass (round 4: it was a launch per layer, 79 of them).
struct FinishItem { const double* scr; float* dw; int n; int pad; };
struct FinishChunk { FinishItem it[96]; };
__global__ __launch_bounds__(256) void wgrad_finish_multi_kernel(FinishChunk c) {
  const FinishItem& t = c.it[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < t.n) t.dw[i] += (float)t.scr[i];
}

// the stem's weight gradient (3 x 3, stride 2, pad 1, 3 -> 24 channels, 176 x 176 outputs per image: the longest reduction of
// the network - 2 M terms per filter entry at batch 64; conv_bwd_weight_kernel: 3.9 ms): a thread walks output pixels with a
// stride, holds 8 output channels x 9 taps of one input channel as fp32 partial sums over at most 16 pixels, the block meets in
// LDS, one double atomic per entry and block.  grid (pixel blocks, 3 input channels, 3 groups of 8 output channels)
__global__ __launch_bounds__(256) void stem_wgrad_kernel(ConvP a, double* __restrict__ scratch) {
  const int OH = a.out.H, OW = a.out.W, IH = a.in.H, IW = a.in.W;
  const int ci = blockIdx.y, cog = blockIdx.z;
  const long long n = (long long)a.B * OH * OW;
  float acc[8][9];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  const long long i0 = (long long)blockIdx.x * 4096;
  for (int it = 0; it < 16; ++it) {
    const long long i = i0 + it * 256 + threadIdx.x;
    if (i >= n) break;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), b = (int)(i / ((long long)OW * OH));
    float xv[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
        xv[ky * 3 + kx] = (iy >= 0 && iy < IH && ix >= 0 && ix < IW) ? a.in.p[a.in.at(b, ci, iy, ix)] : 0.f;
      }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float gv = a.out.p[a.out.at(b, 8 * cog + c, oy, ox)];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[c][t] = __builtin_fmaf(gv, xv[t], acc[c][t]);
    }
  }
  __shared__ float red[4][72];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[c][t];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
      if (lane == 0) red[wave][c * 9 + t] = v;
    }
  __syncthreads();
  if (threadIdx.x < 72) {
    const int c = threadIdx.x / 9, t = threadIdx.x % 9;
    const double v = (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x];
    atomicAdd(&scratch[((size_t)(8 * cog + c) * a.in.C + ci) * 9 + t], v);
  }
}


// depthwise weight gradient (k x k, stride 1 or 2): dW[c][ky][kx] += sum_{b, oy, ox} dy[b][c][oy][ox] x[b][c][s oy - pad + ky][..].
// grid (pixel blocks of 4096, channels): a thread walks 16 output pixels of its channel with k*k fp32 partial sums, the block
// meets by wave shuffles + LDS, one double atomic per tap and block (conv_bwd_weight_kernel: a double LDS tree per tap, 98 us per
// layer at batch 64)
template <int KS>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(ConvP a, double* __restrict__ scratch) {
  constexpr int KK = KS * KS;
  const int OH = a.out.H, OW = a.out.W, IH = a.in.H, IW = a.in.W, c = blockIdx.y;
  const long long n = (long long)a.B * OH * OW;
  float acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = 0.f;
  const long long i0 = (long long)blockIdx.x * 4096;
  for (int it = 0; it < 16; ++it) {
    const long long i = i0 + it * 256 + threadIdx.x;
    if (i >= n) break;
    const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), b = (int)(i / ((long long)OW * OH));
    const float gv = a.out.p[a.out.at(b, c, oy, ox)];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
        const float xv = (iy >= 0 && iy < IH && ix >= 0 && ix < IW) ? a.in.p[a.in.at(b, c, iy, ix)] : 0.f;
        acc[ky * KS + kx] = __builtin_fmaf(gv, xv, acc[ky * KS + kx]);
      }
  }
  __shared__ float red[4][KK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    float v = acc[t];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    if (lane == 0) red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < KK)
    atomicAdd(&scratch[(size_t)c * KK + threadIdx.x], (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
}

struct SgdChunk { yfv2_sgd_item it[96]; };
__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdChunk c, float lr, float momentum, float wd) {
  const yfv2_sgd_item& t = c.it[blockIdx.y];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= t.n) return;
  const float d = t.grad[i] + wd * t.param[i];
  const float bb = t.first_step ? d : momentum * t.momentum_buf[i] + d;
  t.momentum_buf[i] = bb;
  t.param[i] = t.param[i] - lr * bb;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
inline unsigned bn_blocks(size_t n) { return (unsigned)((n + BN_EPB - 1) / BN_EPB); }

struct Tens { size_t off = 0; int C = 0, H = 0, W = 0; };   // offset (floats) into the activation arena; its gradient sits at the same offset of the gradient arena

struct Train {
  yfv2_config cfg{};
  int B = 0;
  float* acts = nullptr; float* grads = nullptr; size_t arena_floats = 0, used = 0;
  int* pool_arg = nullptr; size_t pool_arg_n = 0;
  double* dscr = nullptr; size_t dscr_n = 0, dscr_used = 0;   // four doubles per BatchNorm channel: sum, sum of squares, the two backward sums (zeroed per forward)
  std::map<std::string, float*> param, pgrad;
  double* wscr = nullptr; size_t wscr_n = 0, wscr_used = 0;   // double partial sums of every layer's weight gradient (zeroed per backward)
  std::vector<FinishItem> fin;                                // what wgrad_finish_multi_kernel adds where, collected while the tape runs
  // a layer's range of the scratch; a tensor used twice in one backward (the output convs serve both scales) keeps ONE range
  double* wgrad_scratch(float* dw, int n) {
    for (const auto& f : fin) if (f.dw == dw) return const_cast<double*>(f.scr);
    if (wscr_used + (size_t)n > wscr_n) { if (err.empty()) err = "yfv2_train: weight-gradient scratch exhausted"; return nullptr; }
    double* p = wscr + wscr_used; wscr_used += ((size_t)n + 1) & ~(size_t)1;
    fin.push_back(FinishItem{p, dw, n, 0});
    return p;
  }
  std::vector<std::function<void(hipStream_t)>> tape;
  std::map<std::string, V> relu_out;   // conv name -> the view its ReLU wrote (yfv2_debug_train_relu_output)
  std::string err;
  bool sizing = false;
  size_t per_image_floats = 0;

  Tens alloc(int C, int H, int W) {
    Tens t; t.off = used; t.C = C; t.H = H; t.W = W;
    used += (((size_t)B * C * H * W) + 63) & ~(size_t)63;
    return t;
  }
  V view(const Tens& t, bool grad, int coff = 0, int cstride = 1, int C = -1) const {
    return V{(grad ? grads : acts) + t.off, t.C, coff, cstride, C < 0 ? t.C : C, t.H, t.W};
  }
  float* P(const std::string& n) { auto it = param.find(n); if (it == param.end() || !it->second) { if (err.empty()) err = "yfv2_train: tensor '" + n + "' is not bound"; return nullptr; } return it->second; }
  float* G(const std::string& n) { auto it = pgrad.find(n); if (it == pgrad.end() || !it->second) { if (err.empty()) err = "yfv2_train: gradient of '" + n + "' is not bound"; return nullptr; } return it->second; }

  // pointwise conv launchers (round 4): forward / data gradient as one GEMM, weight gradient = GEMM into the double scratch + finish
  void pw_forward(const V& in, const V& out, const float* w, const float* bias, hipStream_t s) const {
    const int HW = out.H * out.W;
    if (out.C <= 32) hipLaunchKernelGGL((pw_gemm_kernel<false, false, 2>), dim3((HW + 255) / 256, B, 1), dim3(256), 0, s, in, out, w, bias, in.C, B);   // (stage 2's 24 channels: two row tiles)
    else hipLaunchKernelGGL((pw_gemm_kernel<false, false, 4>), dim3((HW + 255) / 256, B, (out.C + 63) / 64), dim3(256), 0, s, in, out, w, bias, in.C, B);
  }
  void pw_data_grad(const V& din, const V& dout, const float* w, int Bc, hipStream_t s) const {   // din += w^T dout
    const int HW = din.H * din.W;
    if (din.C <= 32) hipLaunchKernelGGL((pw_gemm_kernel<true, true, 2>), dim3((HW + 255) / 256, Bc, 1), dim3(256), 0, s, dout, din, w, (const float*)nullptr, din.C, Bc);
    else hipLaunchKernelGGL((pw_gemm_kernel<true, true, 4>), dim3((HW + 255) / 256, Bc, (din.C + 63) / 64), dim3(256), 0, s, dout, din, w, (const float*)nullptr, din.C, Bc);
  }
  void pw_weight_grad(const V& x, const V& dout, float* dw, int Bc, hipStream_t s) {
    const int HW = x.H * x.W, cob = (dout.C + 47) / 48, cib = (x.C + 47) / 48;
    int seg = 256;                                     // pixels per wave: enough waves for the machine, fp32 partial sums over few terms
    while (seg > 64 && (long long)((HW + seg - 1) / seg) * Bc * cob * cib < 2048) seg >>= 1;
    double* scr = wgrad_scratch(dw, dout.C * x.C);
    const int nseg = (HW + seg - 1) / seg, units = nseg * Bc;
    int upw = (int)(((long long)units * cob * cib) / 512);     // about 512 waves per layer (a launch below ~256 waves leaves CUs idle)
    if (upw < 1) upw = 1;
    if (upw > 16) upw = 16;
    if (scr) hipLaunchKernelGGL(pw_wgrad_kernel, dim3((units + upw - 1) / upw, 1, cob * cib), dim3(64), 0, s, x, dout, scr, seg, cib, nseg, units, upw);
  }

  // conv (no bias) + BatchNorm (batch statistics) [+ ReLU]: in view -> zout view (a channel slice of some tensor)
  void conv_bn(const Tens& tin, int in_coff, int in_cstride, int Cin, const Tens& tout, int out_coff, int Cout, const std::string& conv, const std::string& bn,
               int k, int stride, int pad, bool dw, bool relu, bool need_din, hipStream_t s) {
    const int OH = (tin.H + 2 * pad - k) / stride + 1, OW = (tin.W + 2 * pad - k) / stride + 1;
    Tens y = alloc(Cout, OH, OW);
    Tens st = alloc(1, 1, 4 * Cout);   // mean | invstd | the two backward sums (2 per channel)
    if (sizing) return;
    float* w = P(conv + ".weight"); float* gam = P(bn + ".weight"); float* bet = P(bn + ".bias"); float* rm = P(bn + ".running_mean"); float* rv = P(bn + ".running_var");
    float* gw = G(conv + ".weight"); float* gg = G(bn + ".weight"); float* gb = G(bn + ".bias");
    if (!w || !gam || !bet || !rm || !rv || !gw || !gg || !gb) return;
    float* mean = acts + st.off; float* invstd = mean + Cout; float* sums = invstd + Cout;
    ConvP c{view(tin, false, in_coff, in_cstride, Cin), view(y, false), w, nullptr, k, stride, pad, dw ? 1 : 0, B};
    const bool pw = k == 1 && stride == 1 && pad == 0 && !dw;
    if (pw) pw_forward(c.in, c.out, w, nullptr, s);
    else if (dw && pad == k / 2 && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
      const dim3 grid((OH * OW + 255) / 256, B * Cout);
      if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_fwd_kernel<3, 1>), grid, dim3(256), 0, s, c);
      else if (k == 3) hipLaunchKernelGGL((dw_fwd_kernel<3, 2>), grid, dim3(256), 0, s, c);
      else if (stride == 1) hipLaunchKernelGGL((dw_fwd_kernel<5, 1>), grid, dim3(256), 0, s, c);
      else { if (err.empty()) err = "yfv2_train: no kernel for a 5x5 stride-2 depthwise conv (" + conv + ")"; return; }
    } else { if (err.empty()) err = "yfv2_train: no kernel for conv " + conv + " (pointwise, depthwise 3x3 / 5x5 and the stem only)"; return; }   // (round 3's
    // one-thread-per-output conv_fwd / conv_bwd_data / conv_bwd_weight kernels served every layer; nothing of this network reaches them since round 4)
    double* ds = dscr + dscr_used; dscr_used += 4 * (size_t)Cout;
    if (dscr_used > dscr_n) { if (err.empty()) err = "yfv2_train: BatchNorm scratch exhausted"; return; }
    const unsigned nseg = reduce_segments((size_t)B * OH * OW, Cout);
    hipLaunchKernelGGL(bn_stats_part_kernel, dim3(Cout, nseg), dim3(256), 0, s, acts + y.off, B, Cout, OH * OW, ds);
    const V z = view(tout, false, out_coff, 1, Cout), dz = view(tout, true, out_coff, 1, Cout);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_blocks((size_t)B * Cout * OH * OW)), dim3(256), 0, s, acts + y.off, z, B, ds, mean, invstd, rm, rv, gam, bet, relu ? 1 : 0);
    if (relu) relu_out[conv] = z;
    const int Bc = B;
    tape.push_back([=, this](hipStream_t st2) {
      float* yv = acts + y.off; float* dy = grads + y.off;
      hipLaunchKernelGGL(bn_bwd_part_kernel, dim3(Cout, nseg), dim3(256), 0, st2, yv, z, dz, Bc, mean, invstd, relu ? 1 : 0, ds);
      hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_blocks((size_t)Bc * Cout * OH * OW)), dim3(256), 0, st2, yv, z, dz, dy, Bc, mean, invstd, gam, relu ? 1 : 0, ds, gg, gb);
      ConvP cw{view(tin, false, in_coff, in_cstride, Cin), view(y, true), nullptr, nullptr, k, stride, pad, dw ? 1 : 0, Bc};
      if (pw) pw_weight_grad(cw.in, cw.out, gw, Bc, st2);
      else if (dw && (k == 3 || k == 5)) {
        const dim3 grid((unsigned)(((size_t)Bc * OH * OW + 4095) / 4096), Cout);
        double* scr = wgrad_scratch(gw, Cout * k * k);
        if (scr && k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3>), grid, dim3(256), 0, st2, cw, scr);
        else if (scr) hipLaunchKernelGGL((dw_wgrad_kernel<5>), grid, dim3(256), 0, st2, cw, scr);
      }
      if (need_din) {
        ConvP cd{view(tin, true, in_coff, in_cstride, Cin), view(y, true), w, nullptr, k, stride, pad, dw ? 1 : 0, Bc};
        if (pw) pw_data_grad(cd.in, cd.out, w, Bc, st2);
        else if (dw && pad == k / 2 && (k == 3 || k == 5) && (stride == 1 || stride == 2)) {
          const dim3 grid((tin.H * tin.W + 255) / 256, Bc * Cin);
          if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_bwd_data_kernel<3, 1>), grid, dim3(256), 0, st2, cd);
          else if (k == 3) hipLaunchKernelGGL((dw_bwd_data_kernel<3, 2>), grid, dim3(256), 0, st2, cd);
          else hipLaunchKernelGGL((dw_bwd_data_kernel<5, 1>), grid, dim3(256), 0, st2, cd);
        }
      }
    });
  }

  // biased 1x1 output conv (detector.py:25-31): tower -> caller's NCHW logit tensor; its gradient arrives from the caller
  struct HeadUse { Tens tin; std::string name; int Cout; float* out; const float* dout; };
  std::vector<HeadUse> heads;
  void head(const Tens& tin, const std::string& name, int Cout, float* out, hipStream_t s) {
    if (sizing) return;
    float* w = P(name + ".weight"); float* bi = P(name + ".bias");
    if (!w || !bi) return;
    V o{out, Cout, 0, 1, Cout, tin.H, tin.W};
    pw_forward(view(tin, false), o, w, bi, s);
    heads.push_back(HeadUse{tin, name, Cout, out, nullptr});
  }
};

}  // namespace

// the training state hangs off the handle through an opaque slot (yfv2_api.hip owns the handle)
static Train* train_of(yfv2_handle h, bool create) {
  void** slot = yfv2_ctx_train_slot(h);
  if (!slot) return nullptr;
  if (!*slot && create) { Train* t = new Train(); t->cfg = *yfv2_ctx_config(h); *slot = t; }
  return static_cast<Train*>(*slot);
}
void yfv2_train_release(void* p) {
  Train* t = static_cast<Train*>(p);
  if (!t) return;
  if (t->acts) (void)hipFree(t->acts);
  if (t->grads) (void)hipFree(t->grads);
  if (t->pool_arg) (void)hipFree(t->pool_arg);
  if (t->dscr) (void)hipFree(t->dscr);
  if (t->wscr) (void)hipFree(t->wscr);
  delete t;
}

namespace {
// element count of every floating-point state_dict entry of the architecture (model/detector.py:8-19 and the modules it builds:
// shufflenetv2.py:19-46,66-100, fpn.py:5-49) for this configuration
std::map<std::string, int64_t> expected_numels(const yfv2_config& cfg) {
  std::map<std::string, int64_t> m;
  auto conv = [&](const std::string& n, int co, int ci_per_group, int k) { m[n + ".weight"] = (int64_t)co * ci_per_group * k * k; };
  auto bn = [&](const std::string& n, int c) { for (const char* leaf : {".weight", ".bias", ".running_mean", ".running_var"}) m[n + leaf] = c; };
  conv("backbone.first_conv.0", 24, 3, 3); bn("backbone.first_conv.1", 24);
  const int repeats[3] = {4, 8, 4}, chans[4] = {24, 48, 96, 192};
  int cin = 24;
  for (int si = 0; si < 3; ++si) {
    const int cout = chans[si + 1], mid = cout / 2;
    for (int i = 0; i < repeats[si]; ++i) {
      const std::string p = "backbone.stage" + std::to_string(si + 2) + "." + std::to_string(i);
      const int inp = i == 0 ? cin : cin / 2;
      conv(p + ".branch_main.0", mid, inp, 1); bn(p + ".branch_main.1", mid);
      conv(p + ".branch_main.3", mid, 1, 3); bn(p + ".branch_main.4", mid);
      conv(p + ".branch_main.5", cout - inp, mid, 1); bn(p + ".branch_main.6", cout - inp);
      if (i == 0) {
        conv(p + ".branch_proj.0", inp, 1, 3); bn(p + ".branch_proj.1", inp);
        conv(p + ".branch_proj.2", inp, inp, 1); bn(p + ".branch_proj.3", inp);
      }
      cin = cout;
    }
  }
  conv("fpn.conv1x1_2.0", 72, 288, 1); bn("fpn.conv1x1_2.1", 72);
  conv("fpn.conv1x1_3.0", 72, 192, 1); bn("fpn.conv1x1_3.1", 72);
  for (const char* head : {"cls_head_2", "reg_head_2", "reg_head_3", "cls_head_3"}) {
    const std::string p = std::string("fpn.") + head + ".block";
    conv(p + ".0", 72, 1, 5); bn(p + ".1", 72); conv(p + ".3", 72, 72, 1); bn(p + ".4", 72);
    conv(p + ".5", 72, 1, 5); bn(p + ".6", 72); conv(p + ".8", 72, 72, 1); bn(p + ".9", 72);
  }
  const int A = cfg.anchor_num;
  const std::pair<const char*, int> outs[3] = {{"output_reg_layers", 4 * A}, {"output_obj_layers", A}, {"output_cls_layers", cfg.classes}};
  for (const auto& o : outs) { m[std::string(o.first) + ".weight"] = (int64_t)o.second * 72; m[std::string(o.first) + ".bias"] = o.second; }
  return m;
}
}  // namespace

extern "C" {

int yfv2_train_bind(yfv2_handle h, const yfv2_tensor_desc* tensors, int32_t n, const yfv2_tensor_desc* grads, int32_t ng) {
  if (!h || !tensors || n <= 0 || !grads || ng <= 0) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_train_bind: bad argument");
  Train* t = train_of(h, true);
  // every bound buffer is indexed with the shapes the handle's configuration implies: a mis-sized one is refused here
  // (YFV2_ERR_WEIGHTS, like yfv2_load_weights), not read or written out of bounds later
  const std::map<std::string, int64_t> want = expected_numels(t->cfg);
  auto check = [&](const yfv2_tensor_desc& d, const char* what) -> int {
    auto it = want.find(d.name);
    if (it == want.end()) return YFV2_OK;               // not a tensor of this model: never looked up
    if (!d.data || d.numel != it->second)
      return yfv2_ctx_fail(h, YFV2_ERR_WEIGHTS, (std::string("yfv2_train_bind: ") + what + " '" + d.name + "' has " + std::to_string((long long)d.numel) +
                                                 " elements, expected " + std::to_string((long long)it->second)).c_str());
    return YFV2_OK;
  };
  for (int i = 0; i < n; ++i) if (tensors[i].name) if (int rc = check(tensors[i], "tensor")) return rc;
  for (int i = 0; i < ng; ++i) if (grads[i].name) if (int rc = check(grads[i], "gradient buffer")) return rc;
  t->param.clear(); t->pgrad.clear();
  for (int i = 0; i < n; ++i) if (tensors[i].name) t->param[tensors[i].name] = const_cast<float*>(tensors[i].data);
  for (int i = 0; i < ng; ++i) if (grads[i].name) t->pgrad[grads[i].name] = const_cast<float*>(grads[i].data);
  return YFV2_OK;
}

int yfv2_train_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], void* stream) {
  if (!h || !x || !out6 || B < 1) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_train_forward: bad argument");
  Train* t = train_of(h, false);
  if (!t || t->param.empty()) return yfv2_ctx_fail(h, YFV2_ERR_STATE, "yfv2_train_forward: yfv2_train_bind has not been called");
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipSetDevice(t->cfg.device);
  const int H = t->cfg.height, W = t->cfg.width, A = t->cfg.anchor_num, nc = t->cfg.classes;
  t->B = B; t->err.clear(); t->tape.clear(); t->heads.clear(); t->relu_out.clear();

  auto build = [&](bool sizing) {
    t->sizing = sizing; t->used = 0;
    // the input is NOT copied: a view over the caller's tensor (it needs no gradient)
    Tens stem = t->alloc(24, H / 2, W / 2), pooled = t->alloc(24, H / 4, W / 4);
    {
      // stem conv reads the caller's x directly
      if (!sizing) {
        Tens y = t->alloc(24, H / 2, W / 2); Tens st = t->alloc(1, 1, 4 * 24);
        float* w = t->P("backbone.first_conv.0.weight"); float* gam = t->P("backbone.first_conv.1.weight"); float* bet = t->P("backbone.first_conv.1.bias");
        float* rm = t->P("backbone.first_conv.1.running_mean"); float* rv = t->P("backbone.first_conv.1.running_var");
        float* gw = t->G("backbone.first_conv.0.weight"); float* gg = t->G("backbone.first_conv.1.weight"); float* gb = t->G("backbone.first_conv.1.bias");
        if (w && gam && bet && rm && rv && gw && gg && gb) {
          const int OH = H / 2, OW = W / 2;
          float* mean = t->acts + st.off; float* invstd = mean + 24; float* sums = invstd + 24;
          V xin{const_cast<float*>(x), 3, 0, 1, 3, H, W};
          ConvP c{xin, t->view(y, false), w, nullptr, 3, 2, 1, 0, B};
          hipLaunchKernelGGL(stem_fwd_kernel, dim3(blocks_for((size_t)B * OH * OW)), dim3(256), 0, s, c);
          double* ds = t->dscr + t->dscr_used; t->dscr_used += 4 * 24;
          const unsigned nseg = reduce_segments((size_t)B * OH * OW, 24);
          hipLaunchKernelGGL(bn_stats_part_kernel, dim3(24, nseg), dim3(256), 0, s, t->acts + y.off, B, 24, OH * OW, ds);
          const V z = t->view(stem, false), dz = t->view(stem, true);
          hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_blocks((size_t)B * 24 * OH * OW)), dim3(256), 0, s, t->acts + y.off, z, B, ds, mean, invstd, rm, rv, gam, bet, 1);
          t->relu_out["backbone.first_conv.0"] = z;
          Train* tt = t;
          t->tape.push_back([=](hipStream_t st2) {
            float* yv = tt->acts + y.off; float* dy = tt->grads + y.off;
            hipLaunchKernelGGL(bn_bwd_part_kernel, dim3(24, nseg), dim3(256), 0, st2, yv, z, dz, B, mean, invstd, 1, ds);
            hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_blocks((size_t)B * 24 * OH * OW)), dim3(256), 0, st2, yv, z, dz, dy, B, mean, invstd, gam, 1, ds, gg, gb);
            ConvP cw{xin, tt->view(y, true), nullptr, nullptr, 3, 2, 1, 0, B};
            double* scr = tt->wgrad_scratch(gw, 24 * 3 * 9);
            if (scr) hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)(((size_t)B * OH * OW + 4095) / 4096), 3, 3), dim3(256), 0, st2, cw, scr);
          });
          // maxpool
          const size_t np = (size_t)B * 24 * (H / 4) * (W / 4);
          int* arg = t->pool_arg;
          hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks_for(np)), dim3(256), 0, s, t->acts + stem.off, t->acts + pooled.off, arg, B, 24, H / 2, W / 2);
          t->tape.push_back([=](hipStream_t st2) {
            hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks_for((size_t)B * 24 * (H / 2) * (W / 2))), dim3(256), 0, st2, tt->grads + pooled.off, arg, tt->grads + stem.off, B, 24,
                               H / 2, W / 2);
          });
        }
      } else {
        t->alloc(24, H / 2, W / 2); t->alloc(1, 1, 4 * 24);
      }
    }
    Tens cur = pooled;
    Tens feats[3];
    const int repeats[3] = {4, 8, 4}, chans[4] = {24, 48, 96, 192};
    int cin = 24;
    for (int si = 0; si < 3; ++si) {
      const int cout = chans[si + 1], mid = cout / 2;
      for (int i = 0; i < repeats[si]; ++i) {
        const std::string p = "backbone.stage" + std::to_string(si + 2) + "." + std::to_string(i);
        if (i == 0) {   // stride 2: out = cat(proj(x), main(x))
          const int inp = cin, oh = cur.H / 2, ow = cur.W / 2;
          Tens out = t->alloc(cout, oh, ow);
          Tens pd = t->alloc(inp, oh, ow);
          t->conv_bn(cur, 0, 1, inp, pd, 0, inp, p + ".branch_proj.0", p + ".branch_proj.1", 3, 2, 1, true, false, true, s);
          t->conv_bn(pd, 0, 1, inp, out, 0, inp, p + ".branch_proj.2", p + ".branch_proj.3", 1, 1, 0, false, true, true, s);
          Tens m1 = t->alloc(mid, cur.H, cur.W), m2 = t->alloc(mid, oh, ow);
          t->conv_bn(cur, 0, 1, inp, m1, 0, mid, p + ".branch_main.0", p + ".branch_main.1", 1, 1, 0, false, true, true, s);
          t->conv_bn(m1, 0, 1, mid, m2, 0, mid, p + ".branch_main.3", p + ".branch_main.4", 3, 2, 1, true, false, true, s);
          t->conv_bn(m2, 0, 1, mid, out, inp, cout - inp, p + ".branch_main.5", p + ".branch_main.6", 1, 1, 0, false, true, true, s);
          cur = out;
        } else {        // stride 1: even channels pass, odd channels -> main; out = cat(pass, main)
          const int c = cout, c2 = c / 2;
          Tens out = t->alloc(c, cur.H, cur.W);
          if (!sizing) {
            const V dst = t->view(out, false, 0, 1, c2), src = t->view(cur, false, 0, 2, c2);
            hipLaunchKernelGGL(view_copy_kernel, dim3(blocks_for((size_t)B * c2 * cur.H * cur.W)), dim3(256), 0, s, dst, src, B, 0);
            const V gdst = t->view(cur, true, 0, 2, c2), gsrc = t->view(out, true, 0, 1, c2);
            const int hh = cur.H, ww = cur.W;
            t->tape.push_back([=](hipStream_t st2) { hipLaunchKernelGGL(view_copy_kernel, dim3(blocks_for((size_t)B * c2 * hh * ww)), dim3(256), 0, st2, gdst, gsrc, B, 1); });
          }
          Tens m1 = t->alloc(c2, cur.H, cur.W), m2 = t->alloc(c2, cur.H, cur.W);
          t->conv_bn(cur, 1, 2, c2, m1, 0, c2, p + ".branch_main.0", p + ".branch_main.1", 1, 1, 0, false, true, true, s);
          t->conv_bn(m1, 0, 1, c2, m2, 0, c2, p + ".branch_main.3", p + ".branch_main.4", 3, 1, 1, true, false, true, s);
          t->conv_bn(m2, 0, 1, c2, out, c2, c2, p + ".branch_main.5", p + ".branch_main.6", 1, 1, 0, false, true, true, s);
          cur = out;
        }
      }
      feats[si] = cur;
      cin = cout;
    }
    const Tens c2 = feats[1], c3 = feats[2];
    Tens s3 = t->alloc(72, c3.H, c3.W);
    t->conv_bn(c3, 0, 1, 192, s3, 0, 72, "fpn.conv1x1_3.0", "fpn.conv1x1_3.1", 1, 1, 0, false, true, true, s);
    auto tower = [&](const std::string& p, const Tens& in) {
      Tens a1 = t->alloc(72, in.H, in.W), a2 = t->alloc(72, in.H, in.W), a3 = t->alloc(72, in.H, in.W), a4 = t->alloc(72, in.H, in.W);
      t->conv_bn(in, 0, 1, 72, a1, 0, 72, p + ".0", p + ".1", 5, 1, 2, true, true, true, s);
      t->conv_bn(a1, 0, 1, 72, a2, 0, 72, p + ".3", p + ".4", 1, 1, 0, false, false, true, s);
      t->conv_bn(a2, 0, 1, 72, a3, 0, 72, p + ".5", p + ".6", 5, 1, 2, true, true, true, s);
      t->conv_bn(a3, 0, 1, 72, a4, 0, 72, p + ".8", p + ".9", 1, 1, 0, false, false, true, s);
      return a4;
    };
    const Tens cls3 = tower("fpn.cls_head_3.block", s3), reg3 = tower("fpn.reg_head_3.block", s3);
    Tens p2 = t->alloc(288, c2.H, c2.W);
    if (!sizing) {
      const V up = t->view(p2, false, 0, 1, 192), src3 = t->view(c3, false);
      hipLaunchKernelGGL(upsample2_kernel, dim3(blocks_for((size_t)B * 192 * c2.H * c2.W)), dim3(256), 0, s, up, src3, B);
      const V dst2 = t->view(p2, false, 192, 1, 96), src2 = t->view(c2, false);
      hipLaunchKernelGGL(view_copy_kernel, dim3(blocks_for((size_t)B * 96 * c2.H * c2.W)), dim3(256), 0, s, dst2, src2, B, 0);
      const V gup = t->view(p2, true, 0, 1, 192), g3 = t->view(c3, true), g2 = t->view(c2, true), gp2 = t->view(p2, true, 192, 1, 96);
      const int h3 = c3.H, w3 = c3.W, h2 = c2.H, w2 = c2.W;
      t->tape.push_back([=](hipStream_t st2) {
        hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(blocks_for((size_t)B * 192 * h3 * w3)), dim3(256), 0, st2, g3, gup, B);
        hipLaunchKernelGGL(view_copy_kernel, dim3(blocks_for((size_t)B * 96 * h2 * w2)), dim3(256), 0, st2, g2, gp2, B, 1);
      });
    }
    Tens s2 = t->alloc(72, c2.H, c2.W);
    t->conv_bn(p2, 0, 1, 288, s2, 0, 72, "fpn.conv1x1_2.0", "fpn.conv1x1_2.1", 1, 1, 0, false, true, true, s);
    const Tens cls2 = tower("fpn.cls_head_2.block", s2), reg2 = tower("fpn.reg_head_2.block", s2);
    t->head(reg2, "output_reg_layers", 4 * A, out6[0], s);
    t->head(cls2, "output_obj_layers", A, out6[1], s);
    t->head(cls2, "output_cls_layers", nc, out6[2], s);
    t->head(reg3, "output_reg_layers", 4 * A, out6[3], s);
    t->head(cls3, "output_obj_layers", A, out6[4], s);
    t->head(cls3, "output_cls_layers", nc, out6[5], s);
  };

  build(true);
  const size_t need = t->used;
  if (need > t->arena_floats) {
    if (hipDeviceSynchronize() != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: device error before growing the workspace");
    if (t->acts) (void)hipFree(t->acts);
    if (t->grads) (void)hipFree(t->grads);
    t->acts = t->grads = nullptr; t->arena_floats = 0;
    if (hipMalloc(reinterpret_cast<void**>(&t->acts), need * sizeof(float)) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&t->grads), need * sizeof(float)) != hipSuccess)
      return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: out of device memory for the training workspace");
    t->arena_floats = need;
  }
  const size_t npool = (size_t)B * 24 * (H / 4) * (W / 4);
  if (npool > t->pool_arg_n) {
    if (t->pool_arg) { (void)hipDeviceSynchronize(); (void)hipFree(t->pool_arg); }
    if (hipMalloc(reinterpret_cast<void**>(&t->pool_arg), npool * sizeof(int)) != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: out of device memory");
    t->pool_arg_n = npool;
  }
  if (!t->dscr) {
    t->dscr_n = 1 << 16;   // 4 doubles x 16 K BatchNorm channels (the network has ~5 K)
    if (hipMalloc(reinterpret_cast<void**>(&t->dscr), t->dscr_n * sizeof(double)) != hipSuccess) { t->dscr = nullptr; return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: out of device memory"); }
  }
  if (hipMemsetAsync(t->dscr, 0, t->dscr_n * sizeof(double), s) != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: memset failed");
  t->dscr_used = 0;
  if (!t->wscr) {
    t->wscr_n = (size_t)1 << 19;   // every conv weight of the network (~0.24 M entries; the class head: classes (<= 255) x 72) in double
    if (hipMalloc(reinterpret_cast<void**>(&t->wscr), t->wscr_n * sizeof(double)) != hipSuccess) { t->wscr = nullptr; return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: out of device memory"); }
  }
  build(false);
  if (!t->err.empty()) { t->tape.clear(); return yfv2_ctx_fail(h, YFV2_ERR_WEIGHTS, t->err.c_str()); }
  if (hipGetLastError() != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_forward: launch failed");
  return YFV2_OK;
}

int yfv2_train_backward(yfv2_handle h, const float* const grad6[6], void* stream) {
  Train* t = h ? train_of(h, false) : nullptr;
  if (!t || t->tape.empty() || t->heads.size() != 6) return yfv2_ctx_fail(h, YFV2_ERR_STATE, "yfv2_train_backward: no recorded yfv2_train_forward");
  if (!grad6) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_train_backward: null pointer");
  for (int i = 0; i < 6; ++i) if (!grad6[i]) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_train_backward: null gradient tensor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int B = t->B;
  if (hipMemsetAsync(t->grads, 0, t->used * sizeof(float), s) != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_backward: memset failed");
  if (hipMemsetAsync(t->wscr, 0, t->wscr_n * sizeof(double), s) != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_backward: memset failed");
  t->wscr_used = 0; t->fin.clear();
  // the output convs: weights shared by the two scales -> their gradients accumulate (the caller zeroed the parameter gradients)
  for (int i = 0; i < 6; ++i) {
    const auto& u = t->heads[i];
    float* w = t->P(u.name + ".weight"); float* gw = t->G(u.name + ".weight"); float* gb = t->G(u.name + ".bias");
    if (!w || !gw || !gb) return yfv2_ctx_fail(h, YFV2_ERR_WEIGHTS, t->err.c_str());
    V go{const_cast<float*>(grad6[i]), u.Cout, 0, 1, u.Cout, u.tin.H, u.tin.W};
    hipLaunchKernelGGL(bias_bwd_kernel, dim3(u.Cout), dim3(256), 0, s, go, B, gb);
    t->pw_weight_grad(t->view(u.tin, false), go, gw, B, s);
    t->pw_data_grad(t->view(u.tin, true), go, w, B, s);
  }
  for (size_t i = t->tape.size(); i-- > 0;) t->tape[i](s);
  t->tape.clear(); t->heads.clear();
  if (!t->err.empty()) return yfv2_ctx_fail(h, YFV2_ERR_STATE, t->err.c_str());
  for (size_t i0 = 0; i0 < t->fin.size(); i0 += 96) {   // every weight gradient: scratch (double) -> the bound tensors
    FinishChunk c{};
    const int m = (int)std::min<size_t>(96, t->fin.size() - i0);
    int nmax = 0;
    for (int k = 0; k < m; ++k) { c.it[k] = t->fin[i0 + k]; nmax = std::max(nmax, c.it[k].n); }
    hipLaunchKernelGGL(wgrad_finish_multi_kernel, dim3(blocks_for((size_t)nmax), m), dim3(256), 0, s, c);
  }
  t->fin.clear();
  if (hipGetLastError() != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_train_backward: launch failed");
  return YFV2_OK;
}

int64_t yfv2_debug_train_relu_output(yfv2_handle h, const char* conv_name, float* host_dst, int64_t capacity) {
  Train* t = h ? train_of(h, false) : nullptr;
  if (!t || !conv_name) { (void)yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_debug_train_relu_output: bad argument"); return -1; }
  const auto it = t->relu_out.find(conv_name);
  if (it == t->relu_out.end()) { (void)yfv2_ctx_fail(h, YFV2_ERR_STATE, "yfv2_debug_train_relu_output: no such ReLU output in the last yfv2_train_forward"); return -1; }
  const V z = it->second;
  const int64_t n = (int64_t)t->B * z.C * z.H * z.W;
  if (!host_dst || capacity < n) return n;
  float* dense = nullptr;
  if (hipDeviceSynchronize() != hipSuccess || hipMalloc(reinterpret_cast<void**>(&dense), (size_t)n * sizeof(float)) != hipSuccess) {
    (void)yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_debug_train_relu_output: device error"); return -1;
  }
  const V d{dense, z.C, 0, 1, z.C, z.H, z.W};
  hipLaunchKernelGGL(view_copy_kernel, dim3(blocks_for((size_t)n)), dim3(256), 0, nullptr, d, z, t->B, 0);
  const hipError_t e = hipMemcpy(host_dst, dense, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(dense);
  if (e != hipSuccess) { (void)yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_debug_train_relu_output: copy failed"); return -1; }
  return n;
}

int yfv2_sgd_step(yfv2_handle h, float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                  int32_t first_step, void* stream) {
  if (!h || !param || !grad || !momentum_buf || n < 0) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_sgd_step: bad argument");
  if (n == 0) return YFV2_OK;
  const yfv2_sgd_item one{param, grad, momentum_buf, n, first_step ? 1 : 0, 0};   // (one tensor = a table of one: the same kernel)
  return yfv2_sgd_step_multi(h, &one, 1, lr, momentum, weight_decay, stream);
}

int yfv2_sgd_step_multi(yfv2_handle h, const yfv2_sgd_item* items, int32_t n_items, float lr, float momentum, float weight_decay, void* stream) {
  if (!h || !items || n_items < 0) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_sgd_step_multi: bad argument");
  for (int i0 = 0; i0 < n_items; i0 += 96) {
    SgdChunk c{};
    const int m = n_items - i0 < 96 ? n_items - i0 : 96;
    int64_t nmax = 0;
    for (int k = 0; k < m; ++k) {
      const yfv2_sgd_item& t = items[i0 + k];
      if (!t.param || !t.grad || !t.momentum_buf || t.n < 0) return yfv2_ctx_fail(h, YFV2_ERR_ARG, "yfv2_sgd_step_multi: bad item");
      c.it[k] = t;
      if (t.n > nmax) nmax = t.n;
    }
    if (nmax == 0) continue;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(blocks_for((size_t)nmax), m), dim3(256), 0, static_cast<hipStream_t>(stream), c, lr, momentum, weight_decay);
  }
  if (hipGetLastError() != hipSuccess) return yfv2_ctx_fail(h, YFV2_ERR_DEVICE, "yfv2_sgd_step_multi: launch failed");
  return YFV2_OK;
}

}  // extern "C"
