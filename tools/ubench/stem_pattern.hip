// What does the stem's ACCESS PATTERN cost, and which part of it?  (a memory-only build of stem_h3_kernel takes as
// long as the kernel.)  Same launch geometry as yfv2_launch_stem16 - a wave = (image, strip of pooled columns, band of 11 pooled
// rows), lane = (pooled column p, lane group g), lane groups 0..2 load 16 bytes of their channel plane per input row, four input
// rows per pooled row, the next pooled row's loads in flight; pooled results go out as 8-byte pair-plane stores - no arithmetic.
//   bit 0: loads      bit 1: stores      bit 2: strips of 16 pooled columns on 256-byte boundaries instead of 15 (240-byte stride)
//   bit 3: NHWC-24 output with 16-byte stores instead of pair planes
//   bit 4: pair planes padded to 96 columns per row, strips of 16 -> every store instruction writes whole aligned 128-byte lines
//   bit 5: quad planes (what the kernel does since round 4)      bit 6: non-temporal loads      bit 7: non-temporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V>
__global__ __launch_bounds__(64, 3) void k(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W, int R) {
  constexpr bool LD = V & 1, ST = V & 2, AL = (V & 4) || (V & 16), NHWC = V & 8, PADPP = V & 16, QUAD = V & 32, NTL = V & 64, NTS = V & 128;
  const int PH = H >> 2, PW = W >> 2;
  const int SW = AL ? 16 : 15;
  const int strips = AL ? (PW + 15) / 16 : (PW - 1 + 14) / 15;
  const int bands = PH / R, wpi = strips * bands;
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = wid / wpi, wi = wid - b * wpi;
  const int strip = wi % strips, band = wi / strips;
  const int lane = threadIdx.x, p = lane & 15, g = lane >> 4;
  const int px = SW * strip + p;
  const bool lvalid = px < PW;
  const int py0 = band * R;
  const bool st_ok = lvalid && (AL || p > 0 || strip == 0);
  const float* xb = x + (size_t)b * 3 * H * W;
  const bool ldl = lvalid && g < 3;
  const float* src = xb + (size_t)(g < 3 ? g : 0) * H * W + 4 * (lvalid ? px : 0);
  const int PWP = PADPP ? 96 : PW;
  float* ob = NHWC ? out + (((size_t)b * PH + py0) * PW + (st_ok ? px : 0)) * 24
              : QUAD ? out + (size_t)b * 24 * PH * PW + ((size_t)py0 * PW + (st_ok ? px : 0)) * 4
                   : out + (size_t)b * 24 * PH * PWP + ((size_t)py0 * PWP + (st_ok ? px : 0)) * 2;
  f32x4 cur[4], nxt[4];
  auto load4 = [&](int y, f32x4 (&v)[4]) {   // input rows 4y .. 4y+3 (clamped), this lane's 16 bytes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(4 * y + r, H - 1);
      v[r] = (LD && ldl) ? (NTL ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)row * W)) : *reinterpret_cast<const f32x4*>(src + (size_t)row * W)) : (f32x4){1.f, 2.f, 3.f, 4.f};
    }
  };
  load4(py0, cur);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < R; ++t) {
    load4(min(py0 + t + 1, PH - 1), nxt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += cur[r];
    if (ST) {
      if (NHWC) {
        if (st_ok) {
          *reinterpret_cast<f32x4*>(ob + 4 * g) = acc;
          if (g < 2) *reinterpret_cast<f32x4*>(ob + 16 + 4 * g) = acc;
        }
        ob += (size_t)PW * 24;
      } else if (QUAD) {   // [6 planes of four channels][PH][PW][4]: one 16-byte store per channel tile
        if (st_ok) {
          if (NTS) {
            __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(ob + (size_t)g * PH * PW * 4));
            if (g < 2) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(ob + (size_t)(4 + g) * PH * PW * 4));
          } else {
          *reinterpret_cast<f32x4*>(ob + (size_t)g * PH * PW * 4) = acc;
          if (g < 2) *reinterpret_cast<f32x4*>(ob + (size_t)(4 + g) * PH * PW * 4) = acc;
          }
        }
        ob += (size_t)PW * 4;
      } else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          if (st_ok && (tt == 0 || g < 2)) {
            const int q = 8 * tt + 2 * g;
            *reinterpret_cast<f32x2*>(ob + (size_t)q * PH * PWP * 2) = (f32x2){acc[0], acc[1]};
            *reinterpret_cast<f32x2*>(ob + (size_t)(q + 1) * PH * PWP * 2) = (f32x2){acc[2], acc[3]};
          }
        ob += (size_t)PWP * 2;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
  }
  if (!ST && acc[0] == 12345.678f) out[lane] = acc[1];   // keep the loads alive
}
template <int NT>
__global__ void copyk(const f32x4* __restrict__ a, f32x4* __restrict__ o, size_t nin, size_t nout) {   // the same byte counts, fully coalesced
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  f32x4 s = {0, 0, 0, 0};
  for (size_t j = i; j < nin; j += n) s += (NT & 1) ? __builtin_nontemporal_load(a + j) : a[j];
  for (size_t j = i; j < nout; j += n) { if (NT & 2) __builtin_nontemporal_store(s, o + j); else o[j] = s; }
}
// reads and writes interleaved per thread (two quads in, one out), as a streaming kernel with a 2:1 ratio does it
template <int NT>
__global__ void copyk2(const f32x4* __restrict__ a, f32x4* __restrict__ o, size_t nout) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x;
  for (size_t j = i; j < nout; j += n) {
    const f32x4 u = (NT & 1) ? __builtin_nontemporal_load(a + 2 * j) : a[2 * j], v = (NT & 1) ? __builtin_nontemporal_load(a + 2 * j + 1) : a[2 * j + 1];
    if (NT & 2) __builtin_nontemporal_store(u + v, o + j); else o[j] = u + v;
  }
}
template <int V> float run(const float* x, float* out, int B, int H, int W, int iters) {
  const int PH = H / 4, PW = W / 4, R = 11;
  const bool AL = (V & 4) || (V & 16);
  const int strips = AL ? (PW + 15) / 16 : (PW - 1 + 14) / 15;
  const dim3 grid(B * strips * (PH / R));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, grid, dim3(64), 0, 0, x, out, B, H, W, R);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<V>, grid, dim3(64), 0, 0, x, out, B, H, W, R);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}
// stage2.0's side of the same tensor: s2h_kernel's loads (lane = output column = two input columns; per output row two input rows,
// four 16-byte loads each) from PAIR planes (2 pixels x 2 channels per load: as today) or from QUAD planes (1 pixel x 4 channels
// per load: lanes 32 bytes apart), and its eight 8-byte pair-plane stores per output row (unchanged)
template <int Q>
__global__ __launch_bounds__(64, 2) void k2(const float* __restrict__ in, float* __restrict__ out, int B, int IH, int IW, int R, int nstrips, int nb) {
  const int OH = IH >> 1, OW = IW >> 1, wpi = nstrips * nb;
  const int nwg = gridDim.x;
  const int wid = (nwg & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int b = wid / wpi, wi = wid - b * wpi, strip = wi % nstrips, band = wi / nstrips;
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const int ox = 15 * strip + l;
  const bool xok = ox < OW, st_lane = xok && (l > 0 || strip == 0);
  const int y0 = band * R;
  const float* ib = in + (size_t)b * 24 * IH * IW;
  float* ob = out + (size_t)b * 48 * OH * OW;
  auto load_row = [&](int iy, f32x4 (&X)[4]) {
    const int r = min(max(iy, 0), IH - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool on = xok && (k < 2 || g < 2);
      size_t off;
      if (Q) { const int plane = k < 2 ? g : 4 + g, c = k & 1; off = ((size_t)plane * IH * IW + (size_t)r * IW + 2 * ox + c) * 4; }
      else { const int pair = k < 2 ? 2 * g + k : 8 + 2 * g + (k - 2); off = ((size_t)pair * IH * IW + (size_t)r * IW + 2 * ox) * 2; }
      X[k] = on ? *reinterpret_cast<const f32x4*>(ib + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 X[4], Y[4], acc = {0.f, 0.f, 0.f, 0.f};
  load_row(2 * y0 - 1, X); load_row(2 * y0, Y);
  for (int j = 0; j < R; ++j) {
    const int oy = y0 + j;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += X[k] + Y[k];
    load_row(2 * oy + 1, X); load_row(2 * oy + 2, Y);
    __builtin_amdgcn_sched_barrier(0);
    if (st_lane && oy < OH) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < 4 || g < 2) *reinterpret_cast<f32x2*>(ob + ((size_t)(k < 4 ? 4 * g + k : 16 + 2 * g + (k - 4)) * OH * OW + (size_t)oy * OW + ox) * 2) = (f32x2){acc[0], acc[1]};
    }
  }
}
template <int Q> float run2(const float* in, float* out, int B, int iters) {
  const int IH = 88, IW = 88, OW = 44, OH = 44;
  const int nstrips = (OW - 1 + 14) / 15, nb = 4, R = 11;
  const dim3 grid(B * nstrips * nb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k2<Q>, grid, dim3(64), 0, 0, in, out, B, IH, IW, R, nstrips, nb);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k2<Q>, grid, dim3(64), 0, 0, in, out, B, IH, IW, R, nstrips, nb);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / iters;
}
int main() {
  const int B = 256, H = 352, W = 352;
  const size_t nin = (size_t)B * 3 * H * W, nout = (size_t)B * 24 * 88 * 96;
  float *x, *out; hipMalloc(&x, nin * 4); hipMalloc(&out, nout * 4); hipMemset(x, 0, nin * 4);
  for (int rep = 0; rep < 2; ++rep) {
    printf("as the kernel (15-column strips, pair planes)   loads+stores %6.1f us   loads only %6.1f   stores only %6.1f\n", run<3>(x, out, B, H, W, 20), run<1>(x, out, B, H, W, 20), run<2>(x, out, B, H, W, 20));
    printf("16-column strips on 256-byte boundaries          loads+stores %6.1f us   loads only %6.1f   stores only %6.1f\n", run<7>(x, out, B, H, W, 20), run<5>(x, out, B, H, W, 20), run<6>(x, out, B, H, W, 20));
    printf("15-column strips, NHWC-24 output (16-byte stores) loads+stores %6.1f us                       stores only %6.1f\n", run<11>(x, out, B, H, W, 20), run<10>(x, out, B, H, W, 20));
    printf("16-column strips, NHWC-24 output                  loads+stores %6.1f us                       stores only %6.1f\n", run<15>(x, out, B, H, W, 20), run<14>(x, out, B, H, W, 20));
    printf("15-column strips, QUAD planes (16-byte stores)     loads+stores %6.1f us                       stores only %6.1f\n", run<35>(x, out, B, H, W, 20), run<34>(x, out, B, H, W, 20));
    printf("16-column strips, pair planes padded to 96 columns loads+stores %6.1f us                      stores only %6.1f\n", run<19>(x, out, B, H, W, 20), run<18>(x, out, B, H, W, 20));
    printf("stage2.0's pattern (190 MB in, 95 MB out): input in pair planes %6.1f us   in quad planes %6.1f us\n", run2<0>(out, x, B, 20), run2<1>(out, x, B, 20));
    printf("QUAD planes with non-temporal loads %6.1f us   stores %6.1f   both %6.1f\n", run<35 + 64>(x, out, B, H, W, 20), run<35 + 128>(x, out, B, H, W, 20), run<35 + 192>(x, out, B, H, W, 20));
    printf("non-temporal loads ALONE: 15-column strips %6.1f us   16-column strips %6.1f;   16-column strips + NHWC-24 stores, non-temporal loads %6.1f\n",
           run<1 + 64>(x, out, B, H, W, 20), run<5 + 64>(x, out, B, H, W, 20), run<15 + 64>(x, out, B, H, W, 20));
    const size_t n4i = nin / 4, n4o = (size_t)B * 24 * 88 * 88 / 4;
    auto timeit = [&](auto launch) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      return 1e3f * ms / 20;
    };
    const float c0 = timeit([&] { hipLaunchKernelGGL(copyk<0>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4i, n4o); });
    const float c1 = timeit([&] { hipLaunchKernelGGL(copyk<1>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4i, n4o); });
    const float c2 = timeit([&] { hipLaunchKernelGGL(copyk<2>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4i, n4o); });
    const float c3 = timeit([&] { hipLaunchKernelGGL(copyk<3>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4i, n4o); });
    printf("plain streaming kernel, the same 381 MB in + 190 MB out: %6.1f us   nt loads %6.1f   nt stores %6.1f   both %6.1f\n", c0, c1, c2, c3);
    const float d0 = timeit([&] { hipLaunchKernelGGL(copyk2<0>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4o); });
    const float d3 = timeit([&] { hipLaunchKernelGGL(copyk2<3>, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4o); });
    const float d4 = timeit([&] { hipLaunchKernelGGL(copyk2<0>, dim3(256 * 32), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)out, n4o); });
    printf("the same bytes with reads and writes interleaved per thread: %6.1f us   non-temporal %6.1f   four times the workgroups %6.1f\n", d0, d3, d4);
  }
  return 0;
}
