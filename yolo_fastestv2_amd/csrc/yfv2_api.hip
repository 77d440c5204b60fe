// yfv2_api.hip - host side of libyfv2.so: handle, weight folding/re-layout, the
// forward launch plan and the extern "C" entry points declared in include/yfv2.h.
//
// The forward is a static list of launches ("plan") built once per handle from
// the model configuration; executing it enqueues the launches on the caller's
// stream, nothing else.  Reference dataflow followed by the plan (behaviour
// only): model/backbone/shufflenetv2.py:102-109, model/fpn.py:51-64,
// model/detector.py:21-47.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/yfv2.h"
#include "yfv2_internal.h"

thread_local Yfv2LaunchProbe yfv2_launch_probe;   // (yfv2_internal.h: YFV2_LAUNCH)

namespace {

thread_local std::string g_tls_error;
thread_local bool g_creating_lane = false;   // yfv2_create called for a child handle of a laned handle (create_lanes)

struct Folded { size_t w = 0, scale = 0, shift = 0; };  // offsets (floats) into the param blob

enum StepKind { STEP_STEM = 0, STEP_PW = 1, STEP_DW = 2, STEP_TOWER = 4, STEP_S2 = 5, STEP_S1PX = 7, STEP_S2PX = 8, STEP_S1CHAIN = 11, STEP_S1POOL = 12 };

struct Step {
  int kind = 0;
  int K = 0, mode = 0;        // pw
  int ksize = 0, stride = 0;  // dw
  StemArgs stem{};
  PwArgs pw{};
  DwArgs dw{};
  BlockS1Args s1{};
  TowerArgs tw{};
  BlockS2Args s2{};
  S1PxArgs s1px{};
  S2PxArgs s2px{};
  size_t img_off2 = 0;        // STEP_S2PX: main-role image (img_off = proj role); STEP_STEM: filter image for uint8 input
  size_t img_off3 = 0;        // STEP_TOWER: towerh_kernel's image; STEP_STEM / STEP_S2PX / STEP_S2 (stage3.0): the two-term fp16 image of stem_h3_kernel / s2h_kernel / s3h_kernel (0: none)
  bool has_head = false;
  int c2 = 0;                 // fused s1 block
  // offsets into the param blob, resolved to pointers after the upload
  size_t w_off = 0, scale_off = 0, shift_off = 0;
  size_t img_off = 0;         // host-packed LDS image of this launch
  int px_per_img = 0;         // pw: pixels per image (P = B * px_per_img)
  int head0 = -1, head1 = -1; // PW_HEAD: indices into out6
  std::string name;
  double flops = 0, bytes = 0;  // algorithmic, per image; bytes = per-LAYER accounting (every reference layer the launch covers reads its input and writes its output once)
  double bytes_ext = -1;        // SURVEY.md 8(d) for fused launches: EXTERNAL reads + writes of the launch only (-1: same as bytes)
  std::vector<Step> jobs;       // STEP_TOWER on towerh_kernel: the tower halves this ONE launch runs one after the other (empty: the step is its own job)
  bool par = false;             // ... or (par) side by side: independent halves (cls a | reg a, cls b | reg b) as workgroup ranges of one launch
  int tw_tiles = 0;             // STEP_TOWER on towerh_kernel: output-conv tiles the images of the launch are packed for (0, 1 or 6)
  std::string name_plain;       // STEP_S2PX with front: the step's name as a launch of its own
  bool front = false;           // STEP_S2PX: the launch starts from the IMAGE (front_kernel: stem + stage2.0 in one wave); the stem's own step lives in yfv2_ctx::stem_aside
};

struct Buf { float* p = nullptr; size_t per_img = 0; };

}  // namespace

struct yfv2_ctx {
  yfv2_config cfg{};
  int device = 0;
  std::string err;
  bool weights_loaded = false;
  int rows = 0;
  int fh[2] = {0, 0}, fw[2] = {0, 0};

  float* d_params = nullptr;
  size_t n_params = 0;
  std::vector<Step> plan;

  // workspace (NHWC fp32), sized for cfg.max_batch
  Buf a1, s2[2], s3[2], s4[2], t1, t2, t3, f2, f3, fq, ta, tb;
  Buf s2pp;  // stage 2 in pair planes: an image's two buffers back to back, [max_batch][2][24 pairs][H/8][W/8][2] (every offset a kernel adds to an
             // image base stays below 2 x 48 x H/8 x W/8 floats whatever max_batch is: no batch bound from 32-bit buffer offsets)
  // pair-plane bookkeeping at the END of stage 2 (for the stride-2 consumer and for yfv2_debug_activation)
  bool s2_px = false;
  // stem + stage2.0 as ONE launch (front_kernel, yfv2_stage2h.hip): plan[0] is then that launch and the stem's own step is kept HERE, for
  // uint8 input (stem_h3u_kernel, then plan[0] as plain s2h_kernel) and for yfv2_debug_activation(0), which re-runs it on the last input
  bool front_wanted = true;    // YFV2_FRONT=0 at create time: the two-launch form
  bool front_fused = false;
  Step stem_aside{};
  const void* last_x = nullptr; int last_B = 0; bool last_u8 = false;   // the input of the last forward run on THIS handle's workspace (a raw caller pointer: yfv2_debug_activation(0) re-reads it)
  bool stem_pp = false;     // the stem writes pair planes [12][H/4][W/4][2] for stage2.0 (the fp32-matrix plan: stem_px -> s2px kernels)
  int s2_label[48] = {0};   // logical channel stored in slot 2*pair + element
  int s2_buf[24] = {0};     // which of the two buffers holds pair p
  bool bf6 = true;          // pointwise convs on the bf16 matrix cores where a kernel has that form (YFV2_BF6=0 at create time: fp32 MFMA)
  yfv2_plan plan_sw{};      // the caller's plan switches (yfv2_create_ex); the library reads no environment
  bool postfuse = true;     // yfv2_detect: decode + NMS as one launch (yfv2_plan.post_two_launches: two launches)
  bool c2_permuted = false; // stage 3's output (C2) is stored in the chain kernel's order:
  int c2_label[96] = {0};   //   physical channel position k holds logical channel c2_label[k]
  Buf logits[6];
  Buf cand;  // (rows, 8) compact candidate rows of yfv2_detect
  int32_t* d_classes = nullptr;  // class filter scratch (<= 256 entries), then one int32 of its own for the statistics overflow flag
  int32_t* d_stats_flag = nullptr;  // = d_classes + 256
  // the sticky range-guard word of the fp16x3 plan (yfv2_nonfinite): ONE int32 in host-mapped, coherent memory.  The kernels
  // store 1 into it through d_nonfinite (the rare path, a plain store); the host reads h_nonfinite - after waiting for a stream
  // (yfv2_nonfinite: exact) or without waiting (yfv2_nonfinite_peek: what has landed so far).  A lane uses its parent's word.
  int32_t* h_nonfinite = nullptr;
  int32_t* d_nonfinite = nullptr;
  unsigned long long* d_probe = nullptr;   // yfv2_clock_probe_*: [probe_wgs][4] stamps of the last probe launch
  int probe_wgs = 0;
  // training-loss workspace (yfv2_loss): match slots for loss_cap labels, objectness target maps for max_batch images,
  // counters and float64 sums; grown on demand (a growth waits for the device)
  void* d_loss_ws = nullptr;
  size_t loss_ws_bytes = 0;
  void* train = nullptr;         // training state (yfv2_train.hip), created by yfv2_train_bind
  long long* d_trace = nullptr;  // YFV2_TRACE=1: cycle stamps of the last fused s1 launch (debug)
  int trace_step = -1;           // YFV2_TRACE_STEP=i: only launch i of the plan writes stamps (towers: only then)
  // which buffers hold the stage outputs of the last forward (for debug/parity)
  float* dbg[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t dbg_per_img[6] = {0, 0, 0, 0, 0, 0};
  int dbg_c[6] = {0, 0, 0, 0, 0, 0};
  // LANES (YFV2_LANES=N in the environment of yfv2_create): yfv2_forward / yfv2_detect (and their uint8
  // forms) of at least lane_min images cut the batch into N contiguous slices; slice i is run by child handle lanes[i] (own
  // workspace sized max_batch / N, own plan, its own copy of the 1 MB weight blob) on stream lane_stream[i] - lane 0 on the
  // caller's stream - forked from and joined back into the caller's stream with events INSIDE the call: the caller still
  // orders against one stream.  Images are independent, so the result is bit-identical to the unsliced call;
  // what changes is that the one-workgroup-per-image launches of one slice (stages 3 / 4, towers, decode + NMS) share the
  // machine with the streaming launches of another instead of each leaving it under-filled.
  std::vector<yfv2_ctx*> lanes;
  std::vector<hipStream_t> lane_stream;    // [n_lanes - 1]
  std::vector<hipEvent_t> lane_join;       // [n_lanes - 1]
  hipEvent_t lane_fork = nullptr;
  int lane_min = 0;
  bool in_lane = false;                    // this handle IS a lane of another one (never laned itself)
  std::vector<int> last_split;             // slice sizes of the last forward if it ran on the lanes (yfv2_debug_activation)
};

namespace {

int fail(yfv2_ctx* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_tls_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return fail(h, YFV2_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));    \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// ---------------------------------------------------------------------------
// weights: reference state_dict -> one device blob of kernel-ready parameters
// ---------------------------------------------------------------------------
struct WeightPacker {
  std::map<std::string, const yfv2_tensor_desc*> byname;
  std::vector<float> blob;
  std::string missing;

  const float* get(const std::string& name, int64_t numel) {
    auto it = byname.find(name);
    if (it == byname.end() || it->second->data == nullptr) {
      if (missing.empty()) missing = "missing tensor '" + name + "'";
      return nullptr;
    }
    if (it->second->numel != numel) {
      if (missing.empty())
        missing = "tensor '" + name + "' has " + std::to_string(it->second->numel) + " elements, expected " +
                  std::to_string(numel);
      return nullptr;
    }
    return it->second->data;
  }
  size_t reserve(size_t n) {  // 16-byte aligned slots
    size_t off = (blob.size() + 3) & ~size_t(3);
    blob.resize(off + n, 0.f);
    return off;
  }
  // eval-mode BatchNorm2d -> y = x*scale + shift  (ATen: alpha = gamma*invstd, beta = bias - mean*alpha)
  bool bn(const std::string& name, int c, Folded* f) {
    const float* g = get(name + ".weight", c);
    const float* b = get(name + ".bias", c);
    const float* m = get(name + ".running_mean", c);
    const float* v = get(name + ".running_var", c);
    if (!g || !b || !m || !v) return false;
    f->scale = reserve(c);
    f->shift = reserve(c);
    for (int i = 0; i < c; ++i) {
      const float invstd = 1.0f / std::sqrt(v[i] + 1e-5f);
      const float alpha = g[i] * invstd;
      blob[f->scale + i] = alpha;
      blob[f->shift + i] = b[i] - m[i] * alpha;
    }
    return true;
  }
  // pointwise conv weight (co, ci, 1, 1) is already the [M][K] row-major A operand
  bool pw(const std::string& conv, const std::string& bnname, int co, int ci, Folded* f) {
    const float* w = get(conv + ".weight", (int64_t)co * ci);
    if (!w) return false;
    f->w = reserve((size_t)co * ci);
    std::memcpy(&blob[f->w], w, sizeof(float) * co * ci);
    return bn(bnname, co, f);
  }
  // depthwise weight (C,1,k,k) -> [k*k][C] so that a channel quad is one 16-byte load
  bool dw(const std::string& conv, const std::string& bnname, int c, int k, Folded* f) {
    const float* w = get(conv + ".weight", (int64_t)c * k * k);
    if (!w) return false;
    f->w = reserve((size_t)c * k * k);
    for (int ch = 0; ch < c; ++ch)
      for (int t = 0; t < k * k; ++t) blob[f->w + (size_t)t * c + ch] = w[(size_t)ch * k * k + t];
    return bn(bnname, c, f);
  }
  // stem weight (24,3,3,3) -> [27 taps][24 co]
  bool stem(const std::string& conv, const std::string& bnname, Folded* f) {
    const float* w = get(conv + ".weight", 24 * 27);
    if (!w) return false;
    f->w = reserve(24 * 27);
    for (int co = 0; co < 24; ++co)
      for (int t = 0; t < 27; ++t) blob[f->w + (size_t)t * 24 + co] = w[co * 27 + t];
    return bn(bnname, 24, f);
  }
  // biased output convs: rows of several (co_i, 72) matrices stacked; scale = 1, shift = bias
  bool heads(const std::vector<std::pair<std::string, int>>& parts, int ci, Folded* f) {
    int total = 0;
    for (auto& p : parts) total += p.second;
    f->w = reserve((size_t)total * ci);
    f->scale = reserve(total);
    f->shift = reserve(total);
    int row = 0;
    for (auto& p : parts) {
      const float* w = get(p.first + ".weight", (int64_t)p.second * ci);
      const float* b = get(p.first + ".bias", p.second);
      if (!w || !b) return false;
      std::memcpy(&blob[f->w + (size_t)row * ci], w, sizeof(float) * p.second * ci);
      for (int i = 0; i < p.second; ++i) {
        blob[f->scale + row + i] = 1.0f;
        blob[f->shift + row + i] = b[i];
      }
      row += p.second;
    }
    return true;
  }

  // rows [r0, r0 + n) of one biased output conv (the class head of a model with more classes than one launch's 96 rows)
  bool heads_range(const std::string& name, int rows_total, int r0, int n, int ci, Folded* f) {
    const float* w = get(name + ".weight", (int64_t)rows_total * ci);
    const float* b = get(name + ".bias", rows_total);
    if (!w || !b || r0 < 0 || r0 + n > rows_total) return false;
    f->w = reserve((size_t)n * ci); f->scale = reserve(n); f->shift = reserve(n);
    std::memcpy(&blob[f->w], w + (size_t)r0 * ci, sizeof(float) * n * ci);
    for (int i = 0; i < n; ++i) { blob[f->scale + i] = 1.0f; blob[f->shift + i] = b[r0 + i]; }
    return true;
  }

  // ---- LDS images: the exact, zero-padded block of floats a kernel copies into LDS (or its
  // registers) in its prologue.  Built from the arrays packed above.
  // fragment-major filter: frag (mt, s), lane l -> W[16mt + (l&15)][16s + 4(l>>4) .. +3]  (zero outside M x K)
  static void push_frag(std::vector<float>& im, const float* w, int M, int K, int MT, int KC) {
    for (int mt = 0; mt < MT; ++mt)
      for (int s = 0; s < KC; ++s)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j) {
            const int r = 16 * mt + (l & 15), c = 16 * s + 4 * (l >> 4) + j;
            im.push_back((r < M && c < K) ? w[(size_t)r * K + c] : 0.f);
          }
  }
  static void push_rows(std::vector<float>& im, const float* w, int nrows, int C, int KS) {  // [nrows][C] -> [nrows][KS]
    for (int r = 0; r < nrows; ++r)
      for (int c = 0; c < KS; ++c) im.push_back(c < C ? w[(size_t)r * C + c] : 0.f);
  }
  static void push_vec(std::vector<float>& im, const float* v, int n, int padded) {
    for (int i = 0; i < padded; ++i) im.push_back((v && i < n) ? v[i] : 0.f);
  }
  size_t put(const std::vector<float>& im) {
    const size_t off = reserve(im.size());
    std::memcpy(&blob[off], im.data(), sizeof(float) * im.size());
    return off;
  }
  // pw_kernel: filter fragments [MT][K/16][64 lanes][4] (+ an 8-channel tail [MT][64 lanes][2]), scale[MT*16], shift[MT*16]
  size_t image_pw(const Folded& f, int M, int K, int MT /* the M tiles of the kernel instantiation, yfv2_pw_tiles */, bool presplit = false) {
    std::vector<float> im;
    build_pw(im, f, M, K, MT, presplit);
    return put(im);
  }
  // PW_DUAL: two convs on the same input as ONE image of 2 MT output tiles - fragments of the first, fragments of the second (each filter
  // with its own power-of-two scale), then scale[2 MT 16], shift[2 MT 16]
  size_t image_pw_dual(const Folded& f0, const Folded& f1, int M, int K, int MT) {
    std::vector<float> i0, i1, im;
    build_pw(i0, f0, M, K, MT, true);
    build_pw(i1, f1, M, K, MT, true);
    const size_t fr = i0.size() - 2 * (size_t)MT * 16, r = (size_t)MT * 16;
    im.insert(im.end(), i0.begin(), i0.begin() + fr);
    im.insert(im.end(), i1.begin(), i1.begin() + fr);
    im.insert(im.end(), i0.begin() + fr, i0.begin() + fr + r);
    im.insert(im.end(), i1.begin() + fr, i1.begin() + fr + r);
    im.insert(im.end(), i0.begin() + fr + r, i0.end());
    im.insert(im.end(), i1.begin() + fr + r, i1.end());
    return put(im);
  }
  void build_pw(std::vector<float>& im, const Folded& f, int M, int K, int MT, bool presplit) {
    const int rows = MT * 16, K16 = K / 16;
    int sw = 0;
    if (presplit) {
      // pw_kernel<.., PRE>: the filter x 2^sw as two fp16 terms, [mt][chunk pair][term][64 lanes][4 dwords]: dwords 0,1 = K
      // positions 4g..4g+3 of the pair's first chunk, 2,3 = of its second (one A operand of v_mfma_f32_16x16x32_f16)
      float mx = 0.f;
      for (int i = 0; i < M * K; ++i) mx = std::fmax(mx, std::fabs(blob[f.w + i]));
      sw = pow2_for(mx);
      for (int mt = 0; mt < MT; ++mt)
        for (int sp = 0; sp < K16 / 2; ++sp)
          for (int term = 0; term < 2; ++term)
            for (int l = 0; l < 64; ++l)
              for (int d = 0; d < 4; ++d) {
                unsigned packed = 0;
                for (int e = 0; e < 2; ++e) {
                  const int r = 16 * mt + (l & 15), c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1) + e;
                  const float v = (r < M && c < K) ? std::ldexp(blob[f.w + (size_t)r * K + c], sw) : 0.f;
                  const float h1 = rn_f16(v);
                  packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);
                }
                float fb; std::memcpy(&fb, &packed, 4);
                im.push_back(fb);
              }
    } else {
      push_frag(im, &blob[f.w], M, K, MT, K16);
    }
    if (K % 16)
      for (int mt = 0; mt < MT; ++mt)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 2; ++j) {
            const int r = 16 * mt + (l & 15), c = 16 * K16 + 2 * (l >> 4) + j;
            im.push_back((r < M && c < K) ? blob[f.w + (size_t)r * K + c] : 0.f);
          }
    if (presplit) for (int i = 0; i < rows; ++i) im.push_back(i < M ? std::ldexp(blob[f.scale + i], -(sw + 4)) : 0.f);   // the accumulators carry 2^(sw+4): undone exactly inside the BN scale
    else push_vec(im, &blob[f.scale], M, rows);
    push_vec(im, &blob[f.shift], M, rows);
  }
  // columns [c0, c0 + n) of a folded conv's filter as a conv of its own (same BN scale / shift)
  Folded pw_columns(const Folded& f, int co, int ci, int c0, int n) {
    Folded g = f;
    g.w = reserve((size_t)co * n);
    for (int r = 0; r < co; ++r)
      for (int k = 0; k < n; ++k) blob[g.w + (size_t)r * n + k] = blob[f.w + (size_t)r * ci + c0 + k];
    return g;
  }
  // block_s1chain6_kernel: a 48x48 filter as [mt (3)][six 16-byte operands][64 lanes][4 dwords]: hi / mid / lo quads of the
  // chunk PAIR (chunks 0, 1: the 32 k-slots of one bf16 MFMA), then {hi,hi} {mid,mid} {hi,lo} of the single chunk 2 (the
  // register form of yfv2_split_a); every dword = two truncated bf16, low half first
  // block_s1chain6_kernel: a 48x48 filter x 2^sw as two fp16 terms (w1 = RN16, w2 = RN16 of the rest), [mt (3)][three 16-byte
  // operands][64 lanes][4 dwords]: {w1 chunk 0, w1 chunk 1}, {w2 chunk 0, w2 chunk 1}, {w1 chunk 2, w2 chunk 2}; a lane's two
  // dwords of a chunk = K positions 4g..4g+3.  Returns sw.
  static int push_chain6_filter(std::vector<float>& im, const float* w /* [48][48] */) {
    float mx = 0.f;
    for (int i = 0; i < 48 * 48; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    const int sw = pow2_for(mx);
    auto term = [&](int r, int c, int t) {               // packed (value c, value c + 1) of row r, fp16 term t
      unsigned u = 0;
      for (int e = 0; e < 2; ++e) {
        const float v = std::ldexp(w[(size_t)r * 48 + c + e], sw), h1 = rn_f16(v);
        u |= f16_bits(t == 0 ? h1 : v - h1) << (16 * e);
      }
      return u;
    };
    auto put_u = [&](unsigned u) { float f; std::memcpy(&f, &u, 4); im.push_back(f); };
    for (int mt = 0; mt < 3; ++mt)
      for (int op = 0; op < 3; ++op)
        for (int l = 0; l < 64; ++l)
          for (int d = 0; d < 4; ++d) {
            const int r = 16 * mt + (l & 15), kq = 4 * (l >> 4) + 2 * (d & 1);
            if (op < 2) put_u(term(r, 16 * (d >> 1) + kq, op));   // the pair: dwords 0, 1 = chunk 0, dwords 2, 3 = chunk 1
            else put_u(term(r, 32 + kq, d >> 1));                 // chunk 2: dwords 0, 1 = first term, 2, 3 = second
          }
    return sw;
  }
  void append_s1_bf6(std::vector<float>& im, const Folded& f1, const Folded& fd, const Folded& f2) {
    const int sw1 = push_chain6_filter(im, &blob[f1.w]);
    const int sw2 = push_chain6_filter(im, &blob[f2.w]);
    push_rows(im, &blob[fd.w], 9, 48, 48);
    for (const Folded* f : {&f1, &fd, &f2}) {
      const int un = f == &f1 ? sw1 + 4 : (f == &f2 ? sw2 + 4 : 0);   // the pointwise accumulators carry 2^(sw+4): undone exactly inside the BN scale
      for (int i = 0; i < 48; ++i) im.push_back(std::ldexp(blob[f->scale + i], -un));
      push_vec(im, &blob[f->shift], 48, 48);
    }
  }
  // block_s2_kernel<CIN>: W1 | W2 | Wproj | main dw taps | proj dw taps | sc1 sh1 scd shd sc2 sh2 scpd shpd scpp shpp
  size_t image_s2(const Folded& f1, const Folded& fd, const Folded& f2, const Folded& fpd, const Folded& fpp, int cin) {
    const int KC = (cin + 15) / 16, KS = 16 * KC;
    std::vector<float> im;
    push_frag(im, &blob[f1.w], cin, cin, KC, KC);
    push_frag(im, &blob[f2.w], cin, cin, KC, KC);
    push_frag(im, &blob[fpp.w], cin, cin, KC, KC);
    push_rows(im, &blob[fd.w], 9, cin, KS);
    push_rows(im, &blob[fpd.w], 9, cin, KS);
    for (const Folded* f : {&f1, &fd, &f2, &fpd, &fpp}) { push_vec(im, &blob[f->scale], cin, KS); push_vec(im, &blob[f->shift], cin, KS); }
    return put(im);
  }
  // block_s2w_kernel (96 channels): W1 pre-split for bf16x6 | W2 | Wproj | taps | taps | BN vectors as image_s2.
  // W1 pre-split: an fp32 weight is the exact sum of three truncated bf16 terms (hi, mid, lo).  Per (tile of 16 output
  // channels mt, PAIR of 16-channel chunks sp, term) one 16-byte lane quad: {term(w0),term(w1)} {term(w2),term(w3)} of chunk
  // 2sp, then the same of chunk 2sp+1 - the A operand of one v_mfma_f32_16x16x32_bf16 whose 32 k-slots are the two chunks.
  static unsigned bf16_trunc_bits(float v) { unsigned u; std::memcpy(&u, &v, 4); return u >> 16; }
  static float bf16_trunc(float v) { unsigned u; std::memcpy(&u, &v, 4); u &= 0xffff0000u; float r; std::memcpy(&r, &u, 4); return r; }
  // the same layout from an element accessor (sub-matrices): MT output tiles x KP chunk pairs, el(row, column)
  template <class Fn>
  static void push_split3_fn(std::vector<float>& im, int MT, int KP, Fn el) {
    for (int mt = 0; mt < MT; ++mt)
      for (int sp = 0; sp < KP; ++sp)
        for (int term = 0; term < 3; ++term)
          for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
              const int r = 16 * mt + (l & 15), c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1);
              unsigned packed = 0;
              for (int e = 0; e < 2; ++e) {
                float v = el(r, c + e);
                for (int t = 0; t < term; ++t) v = v - bf16_trunc(v);   // exact in fp32
                packed |= bf16_trunc_bits(v) << (16 * e);
              }
              float f; std::memcpy(&f, &packed, 4);
              im.push_back(f);
            }
  }
  // fp16x3 form of the same layout: [mt][chunk pair][term 2][64 lanes][4 dwords of fp16 pairs], el(row, column) already
  // carrying its power of two
  template <class Fn>
  static void push_h2_fn(std::vector<float>& im, int MT, int KP, Fn el) {
    for (int mt = 0; mt < MT; ++mt)
      for (int sp = 0; sp < KP; ++sp)
        for (int term = 0; term < 2; ++term)
          for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
              const int r = 16 * mt + (l & 15), c = 16 * (2 * sp + (d >> 1)) + 4 * (l >> 4) + 2 * (d & 1);
              unsigned packed = 0;
              for (int e = 0; e < 2; ++e) {
                const float v = el(r, c + e), h1 = rn_f16(v);
                packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);
              }
              float f; std::memcpy(&f, &packed, 4);
              im.push_back(f);
            }
  }
  static void push_frag_split3(std::vector<float>& im, const float* w, int M, int K, int MT, int KC) {
    for (int mt = 0; mt < MT; ++mt)
      for (int sp = 0; sp < KC / 2; ++sp)
        for (int term = 0; term < 3; ++term)
          for (int l = 0; l < 64; ++l)
            for (int d = 0; d < 4; ++d) {
              const int s = 2 * sp + (d >> 1);
              const int r = 16 * mt + (l & 15), c = 16 * s + 4 * (l >> 4) + 2 * (d & 1);
              unsigned packed = 0;
              for (int e = 0; e < 2; ++e) {
                float v = (r < M && c + e < K) ? w[(size_t)r * K + c + e] : 0.f;
                for (int t = 0; t < term; ++t) v = v - bf16_trunc(v);   // exact in fp32
                packed |= bf16_trunc_bits(v) << (16 * e);
              }
              float f; std::memcpy(&f, &packed, 4);
              im.push_back(f);
            }
  }
  size_t image_s2w(const Folded& f1, const Folded& fd, const Folded& f2, const Folded& fpd, const Folded& fpp) {
    const int cin = 96, KC = 6, KS = 96;
    std::vector<float> im;
    push_frag_split3(im, &blob[f1.w], cin, cin, KC, KC);
    push_frag(im, &blob[f2.w], cin, cin, KC, KC);
    push_frag(im, &blob[fpp.w], cin, cin, KC, KC);
    push_rows(im, &blob[fd.w], 9, cin, KS);
    push_rows(im, &blob[fpd.w], 9, cin, KS);
    for (const Folded* f : {&f1, &fd, &f2, &fpd, &fpp}) { push_vec(im, &blob[f->scale], cin, KS); push_vec(im, &blob[f->shift], cin, KS); }
    return put(im);
  }
  // tower kernels: pw [80][84] | output conv [mh16][84] | dw taps [25][80] | scd shd scp shp bias [5][96]
  size_t image_tower(const Folded& fd, const Folded& fp, const Folded* fh, int mh) {
    std::vector<float> im;
    push_frag(im, &blob[fp.w], 72, 72, 5, 5);
    const int mh_tiles = fh ? ((mh + 15) / 16 <= 1 ? 1 : 6) : 0;  // kernels are instantiated for 1 or 6 output tiles
    if (fh) push_frag(im, &blob[fh->w], mh, 72, mh_tiles, 5);
    push_rows(im, &blob[fd.w], 25, 72, 80);
    push_vec(im, &blob[fd.scale], 72, 96); push_vec(im, &blob[fd.shift], 72, 96);
    push_vec(im, &blob[fp.scale], 72, 96); push_vec(im, &blob[fp.shift], 72, 96);
    push_vec(im, fh ? &blob[fh->shift] : nullptr, mh, 96);
    return put(im);
  }
  // towerh_kernel (yfv2_towerh.hip): WP [5][5][64][4 dwords of fp16 pairs: term 1, term 1, term 2, term 2] | CS [4][96] |
  // WH [mh tiles][5][64][4] | TAPS [5][4][27][4] + 16
  template <class Fn>
  static void push_a16(std::vector<float>& im, Fn el, int MT) {   // el(row, col) already scaled
    for (int mt = 0; mt < MT; ++mt)
      for (int s = 0; s < 5; ++s)
        for (int l = 0; l < 64; ++l) {
          unsigned dw[4] = {0, 0, 0, 0};
          for (int e = 0; e < 4; ++e) {
            const float v = el(16 * mt + (l & 15), 16 * s + 4 * (l >> 4) + e);
            const float h1 = rn_f16(v);
            dw[e >> 1] |= f16_bits(h1) << (16 * (e & 1));
            dw[2 + (e >> 1)] |= f16_bits(v - h1) << (16 * (e & 1));
          }
          for (int d = 0; d < 4; ++d) { float fb; std::memcpy(&fb, &dw[d], 4); im.push_back(fb); }
        }
  }
  size_t image_towerh(const Folded& fd, const Folded& fp, const Folded* fh, int mh, int mh_tiles) {
    std::vector<float> im;
    const float* wpw = &blob[fp.w];
    float mp = 0.f, mhd = 0.f;
    for (int i = 0; i < 72 * 72; ++i) mp = std::fmax(mp, std::fabs(wpw[i]));
    const int sw = pow2_for(mp);
    push_a16(im, [&](int r, int c) { return (r < 72 && c < 72) ? std::ldexp(wpw[(size_t)r * 72 + c], sw) : 0.f; }, 5);
    // A half that ends in an output conv: pointwise conv, its BatchNorm and the biased output conv are three linear maps in a row
    // (fpn.py:16-17,23-24 - no activation behind the block's last BN; detector.py:25-31), so the kernels apply their PRODUCT to the
    // depthwise result: M = Wh diag(scale) Wp (mh x 72), bias = Wh shift + b, both formed here in double and rounded once - closer
    // to the exact value than the reference's own two fp32 steps.  (The 72 x 72 filter above stays in the image: a launch has one
    // LDS layout for all its jobs, and the halves WITHOUT an output conv use it.)
    std::vector<float> mw, mb;
    int swh = 0;
    if (fh) {
      mw.assign((size_t)mh * 72, 0.f); mb.assign((size_t)mh, 0.f);
      for (int o = 0; o < mh; ++o) {
        double bacc = blob[fh->shift + o];
        for (int k = 0; k < 72; ++k) bacc += (double)blob[fh->w + (size_t)o * 72 + k] * (double)blob[fp.shift + k];
        mb[o] = (float)bacc;
        for (int c = 0; c < 72; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 72; ++k) acc += (double)blob[fh->w + (size_t)o * 72 + k] * (double)blob[fp.scale + k] * (double)wpw[(size_t)k * 72 + c];
          mw[(size_t)o * 72 + c] = (float)acc;
          mhd = std::fmax(mhd, std::fabs((float)acc));
        }
      }
      swh = pow2_for(mhd);
    }
    for (int c = 0; c < 96; ++c) im.push_back(c < 72 ? std::ldexp(blob[fp.scale + c], -(sw + 4)) : 0.f);
    push_vec(im, &blob[fp.shift], 72, 96);
    push_vec(im, fh ? mb.data() : nullptr, mh, 96);
    for (int c = 0; c < 96; ++c) im.push_back(c == 0 ? std::ldexp(1.0f, -(swh + 4)) : 0.f);
    push_a16(im, [&](int r, int c) { return (fh && r < mh && c < 72) ? std::ldexp(mw[(size_t)r * 72 + c], swh) : 0.f; }, mh_tiles);   // zero tiles where a job has no (or a narrower) output conv: one LDS layout per launch
    for (int s = 0; s < 5; ++s)
      for (int q = 0; q < 4; ++q)
        for (int t = 0; t < 27; ++t)
          for (int e = 0; e < 4; ++e) {
            const int ch = 16 * s + 4 * q + e;
            im.push_back(ch >= 72 ? 0.f : t < 25 ? blob[fd.w + (size_t)t * 72 + ch] : 16.0f * (t == 25 ? blob[fd.scale + ch] : blob[fd.shift + ch]));   // x 2^4: exact
          }
    for (int i = 0; i < 16; ++i) im.push_back(0.f);   // the scalar-cache warm-up reads whole 64-byte lines
    // towerp_kernel's table (round 6): the same numbers per channel PAIR, one 256-byte record per (chunk, pair) = what a wave's
    // depthwise unit pulls into 54 SGPRs: floats 2 t + e = tap t of channel 16 s + 2 pair + e, 50 + e = BN scale x 16, 52 + e = BN shift x 16
    for (int s = 0; s < 5; ++s)
      for (int pr = 0; pr < 8; ++pr)
        for (int i = 0; i < 64; ++i) {
          const int t = i >> 1, ch = 16 * s + 2 * pr + (i & 1);
          im.push_back((ch >= 72 || t > 26) ? 0.f : t < 25 ? blob[fd.w + (size_t)t * 72 + ch] : 16.0f * (t == 25 ? blob[fd.scale + ch] : blob[fd.shift + ch]));
        }
    return put(im);
  }
  // ---- stage 2 in lane-per-pixel form (yfv2_stage2.hip)
  // pointwise 24->24 in the 4x4x1 broadcast form [10][64]: register q, lane 4j+i holds entry (m, k) of
  // output position 4m+i, (m*25 + k) = 16q + j; k = 24 is the bias column.  row(n) / col(k) give the
  // filter row / column of output position n / input position k.
  template <class RowFn, class ColFn, class BiasFn>
  static void push_pw24_bcast(std::vector<float>& im, RowFn row, ColFn col, BiasFn bias, const float* w, const float* scale) {
    const size_t base = im.size();
    im.resize(base + 640, 0.f);
    for (int m = 0; m < 6; ++m)
      for (int i = 0; i < 4; ++i) {
        const int r = row(4 * m + i);
        for (int k = 0; k < 25; ++k) {
          const int idx = m * 25 + k;
          im[base + (idx >> 4) * 64 + 4 * (idx & 15) + i] = k < 24 ? w[(size_t)r * 24 + col(k)] * scale[r] : bias(r);
        }
      }
  }
  // s1px_kernel image: w1q | w2q | depthwise taps [54][64] (lane&3 = k holds scaled tap 4q+k, flat index c*9 + dy*3 + dx).
  // order[n] = which branch-input channel sits at input position n = which branch-output channel goes to output position n
  size_t image_s1px(const Folded& f1, const Folded& fd, const Folded& f2, const int (&order)[24]) {
    std::vector<float> im;
    const float* w1 = &blob[f1.w]; const float* w2 = &blob[f2.w]; const float* wd = &blob[fd.w];
    const float* sc1 = &blob[f1.scale]; const float* sh1 = &blob[f1.shift];
    const float* scd = &blob[fd.scale]; const float* shd = &blob[fd.shift];
    const float* sc2 = &blob[f2.scale]; const float* sh2 = &blob[f2.shift];
    push_pw24_bcast(im, [](int n) { return n; }, [&](int k) { return order[k]; }, [&](int r) { return sh1[r]; }, w1, sc1);
    // the depthwise BN shift goes through pw2 (linear): bias2 = shift2 + scale2 * (W2 . shiftd)
    push_pw24_bcast(im, [&](int n) { return order[n]; }, [](int k) { return k; },
                    [&](int r) { double acc = 0; for (int k = 0; k < 24; ++k) acc += (double)w2[(size_t)r * 24 + k] * shd[k]; return sh2[r] + sc2[r] * (float)acc; },
                    w2, sc2);
    push_taps_quad(im, wd, scd);
    return put(im);
  }
  // ---- s1h_kernel (yfv2_stage2h.hip): the same block with both pointwise convs as two-term fp16 operands in the A-operand
  // order of v_mfma_f32_16x16x32_f16.  Lane (l, g = lane >> 4) owns channel POSITIONS npos(g, j), j = 0..7: 4g + j for j < 4
  // (channel tile 0), 16 + 4g + (j - 4) for j >= 4 and g < 2 (tile 1), none otherwise - as K slots of the B operand and as
  // rows 4g..4g+3 of the D tiles alike.  Position n = pair n / 2, element n & 1 of the 12 branch pairs; order[] as image_s1px.
  static int s1h_npos(int g, int j) { return j < 4 ? 4 * g + j : (g < 2 ? 16 + 4 * g + (j - 4) : -1); }
  static int pow2_for(float mx) {   // sw with mx * 2^sw in (2^13, 2^14]
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int sw = 14 - (int)std::ceil(std::log2(mx));
    return sw > 24 ? 24 : (sw < -14 ? -14 : sw);
  }
  // el(row position, K position) -> [tile 2][term 2][64][4 dwords] of fp16 pairs
  template <class Fn>
  static void push_h3_filter(std::vector<float>& im, Fn el) {
    for (int t = 0; t < 2; ++t)
      for (int term = 0; term < 2; ++term)
        for (int l = 0; l < 64; ++l)
          for (int d = 0; d < 4; ++d) {
            unsigned packed = 0;
            for (int e = 0; e < 2; ++e) {
              const int n = s1h_npos(l >> 4, 2 * d + e), r = 16 * t + (l & 15);
              const float v = (n >= 0 && r < 24) ? el(r, n) : 0.f;
              const float h1 = rn_f16(v);
              packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);
            }
            float fb; std::memcpy(&fb, &packed, 4);
            im.push_back(fb);
          }
  }
  size_t image_s1h(const Folded& f1, const Folded& fd, const Folded& f2, const int (&order)[24], const int (&src_off)[12], const int (&dst_off)[12]) {
    const float* w1 = &blob[f1.w]; const float* w2 = &blob[f2.w]; const float* wd = &blob[fd.w];
    const float* sc1 = &blob[f1.scale]; const float* sh1 = &blob[f1.shift];
    const float* scd = &blob[fd.scale]; const float* shd = &blob[fd.shift];
    const float* sc2 = &blob[f2.scale]; const float* sh2 = &blob[f2.shift];
    auto e1 = [&](int r, int n) { return w1[(size_t)r * 24 + order[n]] * sc1[r]; };                 // pw1: natural output channel r, input position n
    auto e2 = [&](int r, int n) { return w2[(size_t)order[r] * 24 + n] * sc2[order[r]]; };          // pw2: output position r, natural input channel n
    float m1 = 0.f, m2 = 0.f;
    for (int r = 0; r < 24; ++r)
      for (int n = 0; n < 24; ++n) { m1 = std::fmax(m1, std::fabs(e1(r, n))); m2 = std::fmax(m2, std::fabs(e2(r, n))); }
    const int sw1 = pow2_for(m1), sw2 = pow2_for(m2);
    std::vector<float> im;
    push_h3_filter(im, [&](int r, int n) { return std::ldexp(e1(r, n), sw1); });
    push_h3_filter(im, [&](int r, int n) { return std::ldexp(e2(r, n), sw2); });
    // taps [18][64]: lane (l, g), register q holds tap f = 4q + (l & 3) = cs * 9 + dy * 3 + dx of the lane's channel slot cs;
    // they see relu(pw1) * 2^(sw1+4) and must hand pw2 its input times 2^4: BN scale * 2^-sw1
    for (int q = 0; q < 18; ++q)
      for (int l = 0; l < 64; ++l) {
        const int f = 4 * q + (l & 3), cs = f / 9, tt = f % 9, n = s1h_npos(l >> 4, cs);
        im.push_back(n >= 0 ? std::ldexp(wd[(size_t)tt * 24 + n] * scd[n], -sw1) : 0.f);
      }
    for (int n = 0; n < 32; ++n) im.push_back(n < 24 ? std::ldexp(sh1[n], sw1 + 4) : 0.f);
    for (int n = 0; n < 32; ++n) {   // the depthwise BN shift goes through pw2 (linear): bias2 = shift2 + scale2 * (W2 . shiftd)
      float v = 0.f;
      if (n < 24) {
        const int r = order[n];
        double acc = 0; for (int k = 0; k < 24; ++k) acc += (double)w2[(size_t)r * 24 + k] * shd[k];
        v = std::ldexp(sh2[r] + sc2[r] * (float)acc, sw2 + 4);
      }
      im.push_back(v);
    }
    im.push_back(std::ldexp(1.0f, -(sw2 + 4)));
    while (im.size() < 3272) im.push_back(0.f);
    for (int which = 0; which < 2; ++which)           // per-lane byte offsets of the lane's four pairs: read from / written to
      for (int k = 0; k < 4; ++k)
        for (int l = 0; l < 64; ++l) {
          const int g = l >> 4, kk = k < 2 ? 2 * g + k : (g < 2 ? 8 + 2 * g + (k - 2) : -1);
          const int v = kk < 0 ? (int)0x80000000 : (which ? dst_off[kk] : src_off[kk]);
          float fb; std::memcpy(&fb, &v, 4);
          im.push_back(fb);
        }
    return put(im);
  }
  // s2h_kernel (yfv2_stage2h.hip): stage2.0 with both branches in one wave.  Input positions = natural channels (the stem's
  // pair planes); output position r of a branch = its channel pos[r] (pos[][] of PlanBuilder::s2px_block: 0..15 = the branch's
  // eight whole pairs, 16..23 = its halves of the eight pairs that mix a proj and a main channel).
  static void push_quad_taps(std::vector<float>& im, const float* wd, const float* scd, int shift_pow2) {
    for (int q = 0; q < 18; ++q)
      for (int l = 0; l < 64; ++l) {
        const int f = 4 * q + (l & 3), cs = f / 9, tt = f % 9, n = s1h_npos(l >> 4, cs);
        im.push_back(n >= 0 ? std::ldexp(wd[(size_t)tt * 24 + n] * scd[n], shift_pow2) : 0.f);
      }
  }
  size_t image_s2h(const Folded& f1, const Folded& fd, const Folded& f2, const Folded& fpd, const Folded& fpp, const int (&pos)[2][24],
                   const int (&st2_off)[2][8], const int (&st1_off)[2][8], int IH, int IW) {
    const float* w1 = &blob[f1.w]; const float* w2 = &blob[f2.w]; const float* wq = &blob[fpp.w];
    const float* sc1 = &blob[f1.scale]; const float* sc2 = &blob[f2.scale]; const float* scq = &blob[fpp.scale];
    auto e1 = [&](int r, int n) { return w1[(size_t)r * 24 + n] * sc1[r]; };
    auto ep = [&](int r, int n) { return wq[(size_t)pos[0][r] * 24 + n] * scq[pos[0][r]]; };
    auto e2 = [&](int r, int n) { return w2[(size_t)pos[1][r] * 24 + n] * sc2[pos[1][r]]; };
    float m1 = 0.f, mp = 0.f, m2 = 0.f;
    for (int r = 0; r < 24; ++r)
      for (int n = 0; n < 24; ++n) { m1 = std::fmax(m1, std::fabs(e1(r, n))); mp = std::fmax(mp, std::fabs(ep(r, n))); m2 = std::fmax(m2, std::fabs(e2(r, n))); }
    const int sw1 = pow2_for(m1), swp = pow2_for(mp), sw2 = pow2_for(m2);
    std::vector<float> im;
    push_h3_filter(im, [&](int r, int n) { return std::ldexp(e1(r, n), sw1); });
    push_h3_filter(im, [&](int r, int n) { return std::ldexp(ep(r, n), swp); });
    push_h3_filter(im, [&](int r, int n) { return std::ldexp(e2(r, n), sw2); });
    push_quad_taps(im, &blob[fd.w], &blob[fd.scale], -sw1);     // main: sees relu(pw1) * 2^(sw1+4), hands pw2 its input * 2^4
    push_quad_taps(im, &blob[fpd.w], &blob[fpd.scale], 0);      // proj: sees the raw input * 2^4
    for (int n = 0; n < 32; ++n) im.push_back(n < 24 ? std::ldexp(blob[f1.shift + n], sw1 + 4) : 0.f);
    auto bias = [&](const float* w, const Folded& fp_, const Folded& fdw, int c) {   // shift + scale * (W . depthwise shift)
      double acc = 0; for (int k = 0; k < 24; ++k) acc += (double)w[(size_t)c * 24 + k] * blob[fdw.shift + k];
      return blob[fp_.shift + c] + blob[fp_.scale + c] * (float)acc;
    };
    for (int n = 0; n < 32; ++n) im.push_back(n < 24 ? std::ldexp(bias(wq, fpp, fpd, pos[0][n]), swp + 4) : 0.f);
    for (int n = 0; n < 32; ++n) im.push_back(n < 24 ? std::ldexp(bias(w2, f2, fd, pos[1][n]), sw2 + 4) : 0.f);
    im.push_back(std::ldexp(1.0f, -(swp + 4)));
    im.push_back(std::ldexp(1.0f, -(sw2 + 4)));
    while (im.size() < 5480) im.push_back(0.f);
    auto put_i = [&](int v) { float fb; std::memcpy(&fb, &v, 4); im.push_back(fb); };
    const int NONE = (int)0x80000000;
    for (int k = 0; k < 4; ++k)                      // loads: the lane's four input pair planes
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, kk = k < 2 ? 2 * g + k : (g < 2 ? 8 + 2 * g + (k - 2) : -1);
        put_i(kk < 0 ? NONE : kk * IH * IW * 8);
      }
    for (int k = 0; k < 8; ++k)                      // stores: proj whole pairs (2), main whole pairs (2), mixed pairs (4)
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4;
        int v = NONE;
        if (k < 2) v = st2_off[0][2 * g + k];
        else if (k < 4) v = st2_off[1][2 * g + (k - 2)];
        else if (g < 2) v = st1_off[0][4 * g + (k - 4)];   // the pair's base: proj sits in element 0, main in element 1
        put_i(v);
      }
    return put(im);
  }
  // s3h_kernel (yfv2_stage2h.hip): stage3.0 (48 -> 96) in the same form.  Lane (l, g) owns positions n48(g, q) = 16 (q / 4) +
  // 4g + q % 4, q = 0..11; as K slots: chunk q / 8, slot q % 8.  Input positions = stage 2's pair-plane slots (the Folded
  // objects passed in already have their input columns / depthwise channels in slot order), outputs natural.
  static int n48(int g, int q) { return 16 * (q >> 2) + 4 * g + (q & 3); }
  size_t image_s3h(const Folded& f1, const Folded& fd, const Folded& f2, const Folded& fpd, const Folded& fpp, unsigned pp_mask,
                   long long pp_bufstride, int IH, int IW) {
    const Folded* fs[3] = {&f1, &fpp, &f2};
    int sw[3];
    std::vector<float> im;
    for (int f = 0; f < 3; ++f) {
      const float* w = &blob[fs[f]->w]; const float* sc = &blob[fs[f]->scale];
      float mx = 0.f;
      for (int r = 0; r < 48; ++r)
        for (int n = 0; n < 48; ++n) mx = std::fmax(mx, std::fabs(w[(size_t)r * 48 + n] * sc[r]));
      sw[f] = pow2_for(mx);
      for (int t = 0; t < 3; ++t)
        for (int c = 0; c < 2; ++c)
          for (int term = 0; term < 2; ++term)
            for (int l = 0; l < 64; ++l)
              for (int d = 0; d < 4; ++d) {
                unsigned packed = 0;
                for (int e = 0; e < 2; ++e) {
                  const int q = 8 * c + 2 * d + e, r = 16 * t + (l & 15);
                  const float v = q < 12 ? std::ldexp(w[(size_t)r * 48 + n48(l >> 4, q)] * sc[r], sw[f]) : 0.f;
                  const float h1 = rn_f16(v);
                  packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);
                }
                float fb; std::memcpy(&fb, &packed, 4);
                im.push_back(fb);
              }
    }
    auto taps = [&](const Folded& fdw, int shift_pow2) {     // [27][64]: lane (l, g), register q' holds tap f = 4q' + (l & 3) = q * 9 + dy * 3 + dx
      for (int qq = 0; qq < 27; ++qq)
        for (int l = 0; l < 64; ++l) {
          const int f = 4 * qq + (l & 3), q = f / 9, tt = f % 9, n = n48(l >> 4, q);
          im.push_back(std::ldexp(blob[fdw.w + (size_t)tt * 48 + n] * blob[fdw.scale + n], shift_pow2));
        }
    };
    taps(fd, -sw[0]);      // main: sees relu(pw1) * 2^(sw1+4), hands pw2 its input * 2^4
    taps(fpd, 0);          // proj: sees the raw input * 2^4
    for (int n = 0; n < 48; ++n) im.push_back(std::ldexp(blob[f1.shift + n], sw[0] + 4));
    auto bias = [&](const Folded& fp_, const Folded& fdw, int c) {
      double acc = 0; for (int k = 0; k < 48; ++k) acc += (double)blob[fp_.w + (size_t)c * 48 + k] * blob[fdw.shift + k];
      return blob[fp_.shift + c] + blob[fp_.scale + c] * (float)acc;
    };
    for (int n = 0; n < 48; ++n) im.push_back(std::ldexp(bias(fpp, fpd, n), sw[1] + 4));
    for (int n = 0; n < 48; ++n) im.push_back(std::ldexp(bias(f2, fd, n), sw[2] + 4));
    im.push_back(std::ldexp(1.0f, -(sw[1] + 4)));
    im.push_back(std::ldexp(1.0f, -(sw[2] + 4)));
    im.push_back(0.f); im.push_back(0.f);
    for (int k = 0; k < 6; ++k)                      // per-lane byte offset of input pair 8t + 2g + h, k = 2t + h
      for (int l = 0; l < 64; ++l) {
        const int pair = 8 * (k >> 1) + 2 * (l >> 4) + (k & 1);
        const long long off = (((pp_mask >> pair) & 1u) ? pp_bufstride * 4 : 0) + (long long)pair * IH * IW * 8;
        const int v = (int)off;
        float fb; std::memcpy(&fb, &v, 4);
        im.push_back(fb);
      }
    return put(im);
  }
  // s4h_kernel (yfv2_stage2h.hip): stage4.0 (96 -> 192).  Lane (l, g) owns positions n96(g, q) = 16 (q / 4) + 4g + q % 4,
  // q = 0..23; K chunk q / 8, slot q % 8.  Input positions = physical NHWC channel positions of stage 3's output (the Folded
  // objects passed in already have their input columns / depthwise channels in that order), outputs natural.
  size_t image_s4h(const Folded& f1, const Folded& fd, const Folded& f2, const Folded& fpd, const Folded& fpp) {
    const Folded* fs[3] = {&f1, &fpp, &f2};
    int sw[3];
    std::vector<float> im;
    for (int f = 0; f < 3; ++f) {
      const float* w = &blob[fs[f]->w]; const float* sc = &blob[fs[f]->scale];
      float mx = 0.f;
      for (int r = 0; r < 96; ++r)
        for (int n = 0; n < 96; ++n) mx = std::fmax(mx, std::fabs(w[(size_t)r * 96 + n] * sc[r]));
      sw[f] = pow2_for(mx);
      for (int t = 0; t < 6; ++t)
        for (int c = 0; c < 3; ++c)
          for (int term = 0; term < 2; ++term)
            for (int l = 0; l < 64; ++l)
              for (int d = 0; d < 4; ++d) {
                unsigned packed = 0;
                for (int e = 0; e < 2; ++e) {
                  const int q = 8 * c + 2 * d + e, r = 16 * t + (l & 15);
                  const float v = std::ldexp(w[(size_t)r * 96 + n48(l >> 4, q)] * sc[r], sw[f]);   // n48's formula is n96's
                  const float h1 = rn_f16(v);
                  packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);
                }
                float fb; std::memcpy(&fb, &packed, 4);
                im.push_back(fb);
              }
    }
    auto taps = [&](const Folded& fdw, int shift_pow2) {     // [54][64]: lane (l, g), register q' holds tap f = 4q' + (l & 3) = q * 9 + dy * 3 + dx
      for (int qq = 0; qq < 54; ++qq)
        for (int l = 0; l < 64; ++l) {
          const int f = 4 * qq + (l & 3), q = f / 9, tt = f % 9, n = n48(l >> 4, q);
          im.push_back(std::ldexp(blob[fdw.w + (size_t)tt * 96 + n] * blob[fdw.scale + n], shift_pow2));
        }
    };
    taps(fd, -sw[0]);
    taps(fpd, 0);
    for (int n = 0; n < 96; ++n) im.push_back(std::ldexp(blob[f1.shift + n], sw[0] + 4));
    auto bias = [&](const Folded& fp_, const Folded& fdw, int c) {
      double acc = 0; for (int k = 0; k < 96; ++k) acc += (double)blob[fp_.w + (size_t)c * 96 + k] * blob[fdw.shift + k];
      return blob[fp_.shift + c] + blob[fp_.scale + c] * (float)acc;
    };
    for (int n = 0; n < 96; ++n) im.push_back(std::ldexp(bias(fpp, fpd, n), sw[1] + 4));
    for (int n = 0; n < 96; ++n) im.push_back(std::ldexp(bias(f2, fd, n), sw[2] + 4));
    im.push_back(std::ldexp(1.0f, -(sw[1] + 4)));
    im.push_back(std::ldexp(1.0f, -(sw[2] + 4)));
    im.push_back(0.f); im.push_back(0.f);
    return put(im);
  }
  // depthwise 3x3 taps of 24 channels, BN scale folded: [54][64], lane&3 = k of register q holds tap 4q+k, flat index c*9 + dy*3 + dx
  static void push_taps_quad(std::vector<float>& im, const float* wd, const float* scd) {
    const size_t base = im.size();
    im.resize(base + 54 * 64, 0.f);
    for (int c = 0; c < 24; ++c)
      for (int t = 0; t < 9; ++t) {
        const int f = c * 9 + t;
        for (int quad = 0; quad < 16; ++quad) im[base + (f >> 2) * 64 + 4 * quad + (f & 3)] = wd[(size_t)t * 24 + c] * scd[c];
      }
  }
  // s2px_kernel role images.  pos[n] = branch-local output channel at output position n of the role's last pointwise conv
  size_t image_s2px_proj(const Folded& fpd, const Folded& fpp, const int (&pos)[24]) {
    std::vector<float> im;
    const float* w = &blob[fpp.w]; const float* sc = &blob[fpp.scale]; const float* sh = &blob[fpp.shift]; const float* shd = &blob[fpd.shift];
    push_pw24_bcast(im, [&](int n) { return pos[n]; }, [](int k) { return k; },
                    [&](int r) { double acc = 0; for (int k = 0; k < 24; ++k) acc += (double)w[(size_t)r * 24 + k] * shd[k]; return sh[r] + sc[r] * (float)acc; }, w, sc);
    push_taps_quad(im, &blob[fpd.w], &blob[fpd.scale]);
    return put(im);
  }
  size_t image_s2px_main(const Folded& f1, const Folded& fd, const Folded& f2, const int (&pos)[24]) {
    std::vector<float> im;
    const float* w2 = &blob[f2.w]; const float* sc2 = &blob[f2.scale]; const float* sh2 = &blob[f2.shift]; const float* shd = &blob[fd.shift];
    const float* sh1 = &blob[f1.shift];
    push_pw24_bcast(im, [](int n) { return n; }, [](int k) { return k; }, [&](int r) { return sh1[r]; }, &blob[f1.w], &blob[f1.scale]);
    push_pw24_bcast(im, [&](int n) { return pos[n]; }, [](int k) { return k; },
                    [&](int r) { double acc = 0; for (int k = 0; k < 24; ++k) acc += (double)w2[(size_t)r * 24 + k] * shd[k]; return sh2[r] + sc2[r] * (float)acc; }, w2, sc2);
    push_taps_quad(im, &blob[fd.w], &blob[fd.scale]);
    return put(im);
  }
  // copies with the INPUT channels re-ordered: position k takes logical channel label[k]
  Folded permuted_pw_inputs(const Folded& f, int co, int ci, const int* label) {
    Folded g = f;
    g.w = reserve((size_t)co * ci);
    for (int r = 0; r < co; ++r)
      for (int k = 0; k < ci; ++k) blob[g.w + (size_t)r * ci + k] = blob[f.w + (size_t)r * ci + label[k]];
    return g;
  }
  // copy with the OUTPUT channels re-ordered: row r (and its BN scale / shift) takes logical output channel label[r]
  Folded permuted_pw_outputs(const Folded& f, int co, int ci, const int* label) {
    Folded g;
    g.w = reserve((size_t)co * ci); g.scale = reserve(co); g.shift = reserve(co);
    for (int r = 0; r < co; ++r) {
      for (int k = 0; k < ci; ++k) blob[g.w + (size_t)r * ci + k] = blob[f.w + (size_t)label[r] * ci + k];
      blob[g.scale + r] = blob[f.scale + label[r]];
      blob[g.shift + r] = blob[f.shift + label[r]];
    }
    return g;
  }
  Folded permuted_dw_channels(const Folded& f, int c, int kk, const int* label) {
    Folded g;
    g.w = reserve((size_t)c * kk); g.scale = reserve(c); g.shift = reserve(c);
    for (int k = 0; k < c; ++k) {
      for (int t = 0; t < kk; ++t) blob[g.w + (size_t)t * c + k] = blob[f.w + (size_t)t * c + label[k]];
      blob[g.scale + k] = blob[f.scale + label[k]];
      blob[g.shift + k] = blob[f.shift + label[k]];
    }
    return g;
  }
  // stem_px_kernel: filter registers in the 4x4x1 broadcast form [11][64]: register q, lane 4j+i holds
  // scale[co] * W[co = 4m+i][k] for (m*27 + k) = 16q + j, k = ky*9 + ci*3 + kx; then shift[24]
  // in_scale: 1 for fp32 input in [0,1]; 1/255 for the uint8 entry points (test.py:38's float()/255 folded into the filter)
  size_t image_stem(const Folded& f, float in_scale = 1.0f) {
    std::vector<float> im(11 * 64 + 24, 0.f);
    const float* w = &blob[f.w];  // [27 taps t = ci*9 + ky*3 + kx][24 co]
    for (int idx = 0; idx < 162; ++idx)
      for (int i = 0; i < 4; ++i) {
        const int co = 4 * (idx / 27) + i, k = idx % 27, ky = k / 9, ci = (k % 9) / 3, kx = k % 3;
        im[(idx >> 4) * 64 + 4 * (idx & 15) + i] = w[(ci * 9 + ky * 3 + kx) * 24 + co] * blob[f.scale + co] * in_scale;
      }
    for (int co = 0; co < 24; ++co) im[11 * 64 + co] = blob[f.shift + co];
    return put(im);
  }
  // stem_h3_kernel (yfv2_stem16.hip): the BN-folded filter times 2^sw as TWO fp16 terms (w = h1 + h2 to 2^-24, round to
  // nearest) in the A-operand order of v_mfma_f32_16x16x32_f16: [channel tile 2][term 2][lane 64][dword 4], lane = 16 g + r
  // holds output channel 16 t + r, K slots 8 g .. 8 g + 7, two halves per dword (low half = even slot).  Slot -> tap:
  //   g < 3 (input channel g): (ky,kx) = (0,1) (0,2) (1,1) (1,2) (0,0) (1,0) (2,0) (2,1);   g = 3: slots 2, 3, 7 = tap (2,2)
  //   of input channels 0, 1, 2, the rest zero.       Then shift * 2^(sw+8) [32 channels, zero beyond 24] and 2^-(sw+8).
  static float rn_f16(float v) { return (float)(_Float16)v; }
  static unsigned f16_bits(float v) { const _Float16 h = (_Float16)v; unsigned short u; std::memcpy(&u, &h, 2); return u; }
  size_t image_stem16(const Folded& f) {
    const float* w = &blob[f.w];   // [27 taps t = ci*9 + ky*3 + kx][24 co]
    auto folded = [&](int co, int ci, int ky, int kx) { return w[(ci * 9 + ky * 3 + kx) * 24 + co] * blob[f.scale + co]; };
    float mx = 0.f;
    for (int co = 0; co < 24; ++co)
      for (int t = 0; t < 27; ++t) mx = std::fmax(mx, std::fabs(w[t * 24 + co] * blob[f.scale + co]));
    int sw = 0;
    if (mx > 0.f && std::isfinite(mx)) { sw = 14 - (int)std::ceil(std::log2(mx)); if (sw > 24) sw = 24; if (sw < -14) sw = -14; }
    const float up = std::ldexp(1.0f, sw);
    static const int TAP[8][2] = {{0, 1}, {0, 2}, {1, 1}, {1, 2}, {0, 0}, {1, 0}, {2, 0}, {2, 1}};
    auto slot_value = [&](int co, int g, int j) -> float {
      if (co >= 24) return 0.f;
      if (g < 3) return folded(co, g, TAP[j][0], TAP[j][1]) * up;
      if (j == 2) return folded(co, 0, 2, 2) * up;
      if (j == 3) return folded(co, 1, 2, 2) * up;
      if (j == 7) return folded(co, 2, 2, 2) * up;
      return 0.f;
    };
    std::vector<float> im;
    for (int t = 0; t < 2; ++t)
      for (int term = 0; term < 2; ++term)
        for (int l = 0; l < 64; ++l)
          for (int d = 0; d < 4; ++d) {
            unsigned packed = 0;
            for (int e = 0; e < 2; ++e) {
              const float v = slot_value(16 * t + (l & 15), l >> 4, 2 * d + e);
              const float h1 = rn_f16(v);
              packed |= f16_bits(term == 0 ? h1 : v - h1) << (16 * e);   // v - h1 is exact in fp32
            }
            float fb; std::memcpy(&fb, &packed, 4);
            im.push_back(fb);
          }
    // the kernel scales the image by 2^8 before splitting it (yfv2_stem16.hip): accumulators carry 2^(sw+8)
    for (int co = 0; co < 32; ++co) im.push_back(co < 24 ? std::ldexp(blob[f.shift + co], sw + 8) : 0.f);
    im.push_back(std::ldexp(1.0f, -(sw + 8)));
    while (im.size() % 4) im.push_back(0.f);
    // stem_h3u_kernel (uint8 pixels 0..255 as they are, one exact fp16 term): accumulators carry 2^sw 255
    for (int co = 0; co < 32; ++co) im.push_back(co < 24 ? (float)(std::ldexp((double)blob[f.shift + co], sw) * 255.0) : 0.f);
    im.push_back((float)(std::ldexp(1.0, -sw) / 255.0));
    while (im.size() % 4) im.push_back(0.f);
    return put(im);
  }
};

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct PlanBuilder {
  yfv2_ctx* h;
  WeightPacker& wp;
  bool ok = true;

  void add_stem(const Buf& out, bool pp_out) {
    Folded f;
    ok &= wp.stem("backbone.first_conv.0", "backbone.first_conv.1", &f);
    Step s;
    s.kind = STEP_STEM;
    s.stem.out = out.p;
    s.stem.H = h->cfg.height;
    s.stem.W = h->cfg.width;
    s.stem.R = 0;  // bands are chosen by the launcher
    s.stem.pp_out = pp_out ? 1 : 0;
    s.img_off = wp.image_stem(f);
    s.img_off2 = wp.image_stem(f, 1.0f / 255.0f);
    s.img_off3 = wp.image_stem16(f);
    s.name = "stem conv3x3s2+bn+relu+maxpool3x3s2";
    const double ch = h->cfg.height / 2.0, cw = h->cfg.width / 2.0;
    s.flops = 2.0 * ch * cw * 27 * 24;
    s.bytes = 4.0 * (3.0 * h->cfg.height * h->cfg.width + (ch / 2) * (cw / 2) * 24);
    h->plan.push_back(s);
  }

  // generic pointwise launch; bn_name empty => Folded given by caller (heads)
  Step& add_pw(const std::string& name, int K, int mode, int M, int px, const float* in, int in_stride, int in_off,
               float* out, int out_stride, int out_off, bool relu, const Folded& f) {
    Step s;
    s.kind = STEP_PW;
    s.K = K; s.mode = mode;
    s.pw.in = in; s.pw.in2 = nullptr; s.pw.out = out;
    s.pw.M = M;
    s.pw.in_stride = in_stride; s.pw.in_off = in_off;
    s.pw.out_stride = out_stride; s.pw.out_off = out_off;
    s.pw.relu = relu ? 1 : 0;
    s.pw.copy = nullptr; s.pw.copy_stride = 0; s.pw.copy_off = 0;
    s.pw.H = 0; s.pw.W = 0; s.pw.HW = px;
    s.pw.nchw0 = nullptr; s.pw.nchw1 = nullptr; s.pw.split = 0; s.pw.ctot0 = 0; s.pw.coff0 = 0;
    s.px_per_img = px;
    s.pw.presplit = (h->bf6 && yfv2_pw_presplit_supported(K, mode, M)) ? 1 : 0;
    s.img_off = wp.image_pw(f, M, K, yfv2_pw_tiles(K, mode, M), s.pw.presplit != 0);
    s.name = name;
    s.flops = 2.0 * px * K * M;
    s.bytes = 4.0 * px * (K + M);
    h->plan.push_back(s);
    return h->plan.back();
  }

  void add_dw(const std::string& name, int k, int stride, int C, int H, int W, const float* in, int in_stride,
              float* out, int out_stride, bool relu, const Folded& f) {
    Step s;
    s.kind = STEP_DW;
    s.ksize = k; s.stride = stride;
    s.dw.in = in; s.dw.out = out;
    s.dw.H = H; s.dw.W = W; s.dw.C = C;
    s.dw.OH = H / stride; s.dw.OW = W / stride;
    s.dw.in_stride = in_stride; s.dw.in_off = 0;
    s.dw.out_stride = out_stride; s.dw.out_off = 0;
    s.dw.relu = relu ? 1 : 0;
    s.w_off = f.w; s.scale_off = f.scale; s.shift_off = f.shift;
    s.name = name;
    s.flops = 2.0 * s.dw.OH * s.dw.OW * C * k * k;
    s.bytes = 4.0 * C * ((double)H * W + (double)s.dw.OH * s.dw.OW);
    h->plan.push_back(s);
  }

  // ShuffleV2Block stride 2 (shufflenetv2.py:19-44,52-55): out = cat(proj(x), main(x))
  // pp_label != nullptr: the input is stage 2's pair-plane layout (slot k holds logical channel pp_label[k],
  // pair p lives in buffer pp_buf[p]); only the fused kernel reads it
  // in_label != nullptr: the NHWC input holds logical channel in_label[k] at position k (stage 3 written by the chain kernel)
  void block_s2(const std::string& p, int cin, int H, int W, const Buf& x, const Buf& y, const int* pp_label = nullptr,
                const int* pp_buf = nullptr, long long pp_bufstride = 0, const int* in_label = nullptr) {
    Folded f;
    const int oh = H / 2, ow = W / 2, co = 2 * cin;
    const bool layer_plan = h->plan_sw.layer_by_layer != 0;
    const int rfused = (cin == 24 || cin == 48 || (cin == 96 && !pp_label && h->bf6)) ? yfv2_block_s2_rows(cin, H, W) : 0;   // 96: block_s2w_kernel (its pw1 is bf16x6 only)
    if (!layer_plan && rfused > 0) {
      Folded f1, fd, f2, fpd, fpp;
      ok &= wp.dw(p + ".branch_proj.0", p + ".branch_proj.1", cin, 3, &fpd);
      ok &= wp.pw(p + ".branch_proj.2", p + ".branch_proj.3", cin, cin, &fpp);
      ok &= wp.pw(p + ".branch_main.0", p + ".branch_main.1", cin, cin, &f1);
      ok &= wp.dw(p + ".branch_main.3", p + ".branch_main.4", cin, 3, &fd);
      ok &= wp.pw(p + ".branch_main.5", p + ".branch_main.6", cin, cin, &f2);
      Step s;
      s.kind = STEP_S2;
      s.c2 = cin;
      s.s2.in = x.p; s.s2.out = y.p;
      s.s2.H = H; s.s2.W = W; s.s2.R = rfused;
      if (pp_label && ok) {  // channel position k of the staged tile = slot k: re-order every per-input-channel parameter
        f1 = wp.permuted_pw_inputs(f1, cin, cin, pp_label);
        fpd = wp.permuted_dw_channels(fpd, cin, 9, pp_label);
        fpp = wp.permuted_pw_inputs(fpp, cin, cin, pp_label);
        s.s2.pp_in = 1;
        s.s2.pp_bufstride = pp_bufstride;
        s.s2.pp_imgstride = 2 * pp_bufstride;
        for (int q = 0; q < cin / 2; ++q) if (pp_buf[q]) s.s2.pp_mask |= 1u << q;
      } else if (in_label && ok) {
        f1 = wp.permuted_pw_inputs(f1, cin, cin, in_label);
        fpd = wp.permuted_dw_channels(fpd, cin, 9, in_label);
        fpp = wp.permuted_pw_inputs(fpp, cin, cin, in_label);
      }
      s.img_off = cin == 96 ? wp.image_s2w(f1, fd, f2, fpd, fpp) : wp.image_s2(f1, fd, f2, fpd, fpp, cin);
      if (cin == 48 && pp_label && ok && h->bf6 && yfv2_s3h_supported(H, W))   // the streaming form on the f16 matrix cores (yfv2_stage2h.hip)
        s.img_off3 = wp.image_s3h(f1, fd, f2, fpd, fpp, s.s2.pp_mask, pp_bufstride, H, W);
      if (cin == 96 && !pp_label && ok && h->bf6 && yfv2_s4h_supported(H, W))
        s.img_off3 = wp.image_s4h(f1, fd, f2, fpd, fpp);
      s.name = p + " fused s2 block: proj(dw3x3s2+bn -> pw+bn+relu) | main(pw1+bn+relu -> dw3x3s2+bn -> pw2+bn+relu) | cat";
      s.flops = 2.0 * ((double)H * W * cin * cin + 2.0 * oh * ow * cin * cin + 2.0 * oh * ow * 9 * cin);
      s.bytes = 4.0 * ((double)H * W * cin + (double)oh * ow * co);
      h->plan.push_back(s);
      return;
    }
    ok &= wp.dw(p + ".branch_proj.0", p + ".branch_proj.1", cin, 3, &f);
    if (in_label && ok) f = wp.permuted_dw_channels(f, cin, 9, in_label);
    add_dw(p + ".proj.dw3x3s2+bn", 3, 2, cin, H, W, x.p, cin, h->t3.p, cin, false, f);
    ok &= wp.pw(p + ".branch_proj.2", p + ".branch_proj.3", cin, cin, &f);
    if (in_label && ok) f = wp.permuted_pw_inputs(f, cin, cin, in_label);
    add_pw(p + ".proj.pw+bn+relu", cin, PW_PLAIN, cin, oh * ow, h->t3.p, cin, 0, y.p, co, 0, true, f);
    ok &= wp.pw(p + ".branch_main.0", p + ".branch_main.1", cin, cin, &f);
    if (in_label && ok) f = wp.permuted_pw_inputs(f, cin, cin, in_label);
    add_pw(p + ".main.pw1+bn+relu", cin, PW_PLAIN, cin, H * W, x.p, cin, 0, h->t1.p, cin, 0, true, f);
    ok &= wp.dw(p + ".branch_main.3", p + ".branch_main.4", cin, 3, &f);
    add_dw(p + ".main.dw3x3s2+bn", 3, 2, cin, H, W, h->t1.p, cin, h->t2.p, cin, false, f);
    ok &= wp.pw(p + ".branch_main.5", p + ".branch_main.6", cin, cin, &f);
    add_pw(p + ".main.pw2+bn+relu", cin, PW_PLAIN, cin, oh * ow, h->t2.p, cin, 0, y.p, co, cin, true, f);
  }

  // ---- stage 2 in lane-per-pixel form (yfv2_stage2.hip).  Bookkeeping of the pair-plane layout:
  // label[slot] = logical channel (numbered as the input of the NEXT block) stored in slot 2*pair + element,
  // buf[pair] = which of the two stage buffers holds the pair.  A stride-1 block (shufflenetv2.py:57-63,
  // 48-51) sends its even input channels 2j to output channel j untouched and its odd input channels 2i+1
  // through the branch to output channel c2+i: in slot terms the even-labelled pairs are simply re-labelled
  // (label /= 2) and the odd-labelled pairs are read, transformed and written to the OTHER buffer's copy of
  // the same pair (no in-place halo races), re-labelled c2 + (label-1)/2.  yfv2_stage2_channel() places the
  // stride-2 block's 48 outputs so that every pair stays wholly even or wholly odd for all three blocks.
  struct Stage2Layout {
    int label[48];
    int buf[24];
  };
  // stage2.0 in lane-per-pixel form: reads the stem's pair planes, writes logical channel c to slot(c) of buffer 0
  void s2px_block(const std::string& p, int IH, int IW) {
    Folded f1, fd, f2, fpd, fpp;
    ok &= wp.dw(p + ".branch_proj.0", p + ".branch_proj.1", 24, 3, &fpd);
    ok &= wp.pw(p + ".branch_proj.2", p + ".branch_proj.3", 24, 24, &fpp);
    ok &= wp.pw(p + ".branch_main.0", p + ".branch_main.1", 24, 24, &f1);
    ok &= wp.dw(p + ".branch_main.3", p + ".branch_main.4", 24, 3, &fd);
    ok &= wp.pw(p + ".branch_main.5", p + ".branch_main.6", 24, 24, &f2);
    const int OH = IH / 2, OW = IW / 2;
    int slot_of[48];
    for (int k = 0; k < 48; ++k) slot_of[yfv2_stage2_channel(k)] = k;
    Step s;
    s.kind = STEP_S2PX;
    // output positions: 0..15 = the role's eight whole pairs, 16..23 = its halves of the eight mixed pairs
    int pos[2][24];
    for (int j = 0; j < 8; ++j) {
      pos[0][2 * j] = j;      pos[0][2 * j + 1] = 8 + j;   pos[0][16 + j] = 16 + j;   // proj: logical channels 0..23
      pos[1][2 * j] = 8 + j;  pos[1][2 * j + 1] = 16 + j;  pos[1][16 + j] = j;        // main: logical 24 + (..)
    }
    for (int role = 0; role < 2; ++role) {
      for (int i = 0; i < 8; ++i) {
        const int s0 = slot_of[24 * role + pos[role][2 * i]], s1 = slot_of[24 * role + pos[role][2 * i + 1]];
        if ((s0 & 1) || s1 != s0 + 1) { ok = false; return; }
        s.s2px.st2_off[role][i] = (s0 >> 1) * OH * OW * 8;
        const int ss = slot_of[24 * role + pos[role][16 + i]];
        s.s2px.st1_off[role][i] = (ss >> 1) * OH * OW * 8 + (ss & 1) * 4;
      }
    }
    s.s2px.in = h->a1.p; s.s2px.act = h->s2pp.p;
    s.s2px.IH = IH; s.s2px.IW = IW;
    s.s2px.in_stride = 24 * IH * IW; s.s2px.out_stride = 2 * 48 * OH * OW;   // (an image owns both of its stage-2 buffers; this block fills buffer 0)
    s.s2px.in_records = 24 * IH * IW * 4; s.s2px.out_records = 48 * OH * OW * 4;
    if (ok) {
      s.img_off = wp.image_s2px_proj(fpd, fpp, pos[0]); s.img_off2 = wp.image_s2px_main(f1, fd, f2, pos[1]);
      for (int i = 0; i < 8; ++i)   // s2h_kernel stores a mixed pair whole: proj must sit in element 0, main right behind it
        if ((s.s2px.st1_off[0][i] & 7) != 0 || s.s2px.st1_off[1][i] != s.s2px.st1_off[0][i] + 4) ok = false;
      if (ok) s.img_off3 = wp.image_s2h(f1, fd, f2, fpd, fpp, pos, s.s2px.st2_off, s.s2px.st1_off, IH, IW);
    }
    s.name = p + " s2 block, lane-per-pixel: proj(dw3x3s2+bn -> pw+bn+relu) | main(pw1+bn+relu -> dw3x3s2+bn -> pw2+bn+relu) -> pair planes";
    s.flops = 2.0 * ((double)IH * IW * 24 * 24 + 2.0 * OH * OW * 24 * 24 + 2.0 * OH * OW * 9 * 24);
    s.bytes = 4.0 * ((double)IH * IW * 24 + (double)OH * OW * 48);
    h->plan.push_back(s);
  }
  void s1px_block(const std::string& p, int H, int W, Stage2Layout& L, long long bufstride) {
    Folded f1, fd, f2;
    ok &= wp.pw(p + ".branch_main.0", p + ".branch_main.1", 24, 24, &f1);
    ok &= wp.dw(p + ".branch_main.3", p + ".branch_main.4", 24, 3, &fd);
    ok &= wp.pw(p + ".branch_main.5", p + ".branch_main.6", 24, 24, &f2);
    Step s;
    s.kind = STEP_S1PX;
    int order[24], kk = 0;
    for (int q = 0; q < 24; ++q) {
      const bool odd0 = L.label[2 * q] & 1, odd1 = L.label[2 * q + 1] & 1;
      if (odd0 != odd1) { ok = false; return; }   // cannot happen with yfv2_stage2_channel's placement
      if (!odd0) continue;
      if (kk >= 12) { ok = false; return; }
      order[2 * kk] = (L.label[2 * q] - 1) / 2;
      order[2 * kk + 1] = (L.label[2 * q + 1] - 1) / 2;
      s.s1px.src_off[kk] = (int)(((long long)L.buf[q] * bufstride + (long long)q * H * W * 2) * 4);
      s.s1px.dst_off[kk] = (int)(((long long)(1 - L.buf[q]) * bufstride + (long long)q * H * W * 2) * 4);
      ++kk;
    }
    if (kk != 12) { ok = false; return; }
    for (int q = 0; q < 24; ++q) {
      if (L.label[2 * q] & 1) {
        L.label[2 * q] = 24 + (L.label[2 * q] - 1) / 2;
        L.label[2 * q + 1] = 24 + (L.label[2 * q + 1] - 1) / 2;
        L.buf[q] ^= 1;
      } else {
        L.label[2 * q] /= 2;
        L.label[2 * q + 1] /= 2;
      }
    }
    s.s1px.act = h->s2pp.p;
    s.s1px.H = H; s.s1px.W = W;
    s.s1px.img_stride = 2 * 48 * H * W;
    s.s1px.num_records = (int)((bufstride + 48LL * H * W) * 4);
    if (ok) { s.img_off = wp.image_s1px(f1, fd, f2, order); s.img_off2 = wp.image_s1h(f1, fd, f2, order, s.s1px.src_off, s.s1px.dst_off); }
    s.name = p + " s1 block, lane-per-pixel: pw1+bn+relu -> dw3x3+bn -> pw2+bn+relu on the 12 branch pairs (shuffle/pass/cat = bookkeeping)";
    s.flops = 2.0 * H * W * (2.0 * 24 * 24 + 9.0 * 24);
    s.bytes = 4.0 * H * W * (2.0 * 48);  // the layer's logical input + output; the launch itself moves half of it
    s.bytes_ext = 4.0 * H * W * (2.0 * 24);   // the 12 branch pairs in, the 12 fresh pairs out; the pass-through half never moves
    h->plan.push_back(s);
  }

  // ---- a chain of stride-1 blocks as ONE launch (block_s1chain_kernel, yfv2_block.hip).  The kernel moves data in a
  // fixed, lane-uniform way (pixel slot owned by the 4 lanes g of a 16-lane row; per block and lane: accumulator elements
  // 1, 3 -> next tile, element 0 -> held one block, element 2 -> parked in Z; next tile quads = (held 3 + parked 1 |
  // parked 2 + fresh 2 | fresh 4)); this planner decides which LOGICAL channel each of those positions carries so that
  // the whole thing is the reference's channel_shuffle / pass-through / cat chain (shufflenetv2.py:48-51,57-63):
  //   logical activation A_k (96 channels) before block k:  branch input i = A_k[2i+1],  A_{k+1} = [A_k[0::2], F_k]
  // A fresh output F_k[j] (index 48 + j in A_{k+1}) becomes a branch input after L blocks, L = 1 + trailing zeros of its
  // index: odd j at once (24 values -> elements 1, 3), j = 2 mod 4 after one pass (12 -> element 0), j = 0 mod 4 later
  // (12 -> element 2, parked).  X[2i+1] feed block 1 from the load, X[4i+2] are held for block 2, X[4i] stay in memory.
  // Outputs: the blocks' images (pw1 input columns / pw2 output rows permuted, tables PS / PL appended) and
  // z_label[pos] = logical channel of the chain's output stored at Z position pos.
  struct ChainLoc { int kind = 0, blk = 0, mt = 0, g = 0, e = 0, off = 0; };   // kind 0: X[off] (loaded up front), 1: accumulator of block blk, 2: parked at Z[off]
  void s1chain_block(const std::vector<std::string>& names, int c, int H, int W, const Buf& x, const Buf& y, int* z_label) {
    const int c2 = c / 2, NB = (int)names.size();
    std::vector<Folded> f1(NB), fd(NB), f2(NB);
    for (int k = 0; k < NB; ++k) {
      ok &= wp.pw(names[k] + ".branch_main.0", names[k] + ".branch_main.1", c2, c2, &f1[k]);
      ok &= wp.dw(names[k] + ".branch_main.3", names[k] + ".branch_main.4", c2, 3, &fd[k]);
      ok &= wp.pw(names[k] + ".branch_main.5", names[k] + ".branch_main.6", c2, c2, &f2[k]);
    }
    std::vector<float> im;
    if (ok && c2 == 48 && NB >= 3 && NB <= 7) {
      // Z positions: [12 (k - 2), + 12) = the parked inputs of block k (k = 2 .. NB-1; all of it below 60 and free until
      // the final stores), [60, 96) = parked values no block consumes (they are already where the output wants them)
      std::vector<int> group_n(NB, 0);
      int next_final = 60;
      // consumer of the value that has index idx in the activation entering block kn: the block that takes it as a
      // branch input, or NB if it survives the chain
      auto consumer = [&](int idx, int kn) {
        int steps = 0;
        while (!(idx & 1) && idx != 0) { idx >>= 1; ++steps; }
        return (idx == 0 || kn + steps > NB - 1) ? NB : kn + steps;
      };
      // n consecutive Z positions for values with consumer kc
      auto park_slots = [&](int kc, int n) {
        if (kc >= NB) { const int p0 = next_final; next_final += n; if (next_final > 96) ok = false; return p0; }
        if (kc < 2 || group_n[kc] + n > 12) { ok = false; return 0; }
        const int p0 = 12 * (kc - 2) + group_n[kc];
        group_n[kc] += n;
        return p0;
      };
      std::vector<ChainLoc> act(96);                     // where logical channel o of the current activation lives
      std::vector<std::vector<int>> tables(NB, std::vector<int>(36, 0));   // per block: PS[i][g] | (block 0) XS[c][g]
      for (int o = 0; o < 96; ++o) { act[o].kind = 0; act[o].off = o; }
      // X[16 cq + 4 g] are parked at load time.  Lane groups 1..3: the kernel stores (cq = 0,1,2) and (cq = 3,4,5) as two
      // 12-byte runs, so each triple must share a consumer; lane group 0: six single dwords (X[0] passes every block: Z[95])
      for (int g = 0; g < 4 && ok; ++g) {
        if (g == 0) {
          for (int cq = 0; cq < 6; ++cq) {
            const int o = 16 * cq;
            act[o].kind = 2;
            act[o].off = o == 0 ? 95 : park_slots(consumer(o, 0), 1);
            tables[0][12 + cq * 4 + 0] = act[o].off;
          }
          if (next_final > 95) ok = false;                // Z[95] is X[0]'s
        } else {
          for (int t = 0; t < 2; ++t) {
            const int kc = consumer(16 * (3 * t) + 4 * g, 0);
            for (int i = 1; i < 3; ++i) if (consumer(16 * (3 * t + i) + 4 * g, 0) != kc) ok = false;
            const int p0 = park_slots(kc, 3);
            for (int i = 0; i < 3; ++i) {
              const int o = 16 * (3 * t + i) + 4 * g;
              act[o].kind = 2; act[o].off = p0 + i;
              tables[0][12 + (3 * t + i) * 4 + g] = p0 + i;
            }
          }
        }
      }
      auto tile_of_fresh = [](int mt, int e) {           // accumulator (mt, element 1 | 3) -> tile (quad j, element)
        const int hi = e == 3 ? 1 : 0;
        if (mt == 0) return std::make_pair(1, 2 + hi);
        if (mt == 1) return std::make_pair(2, 0 + hi);
        return std::make_pair(2, 2 + hi);
      };
      for (int k = 0; k < NB && ok; ++k) {
        // ---- where does branch input i of this block sit in the tile?  label[physical column 16 j + 4 g + e] = i
        int label[48];
        for (int q = 0; q < 48; ++q) label[q] = -1;
        int parked_n = 0;
        for (int i = 0; i < 48; ++i) {
          const ChainLoc& L = act[2 * i + 1];
          int j = -1, g = -1, e = -1;
          if (k == 0) {                                   // loaded from X: quad cq = off / 16, lane group, element 1 | 3
            if (L.kind != 0 || !(L.off & 1)) { ok = false; break; }
            const int cq = L.off / 16; g = (L.off % 16) / 4;
            j = cq / 2; e = (cq & 1) * 2 + ((L.off & 3) == 3 ? 1 : 0);
          } else if (L.kind == 1 && L.blk == k - 1 && (L.e == 1 || L.e == 3)) {   // fresh output of the previous block
            const auto t = tile_of_fresh(L.mt, L.e); j = t.first; e = t.second; g = L.g;
          } else if (k == 1 && L.kind == 0 && (L.off & 3) == 2) {                  // X element 2, held since the load
            const int cq = L.off / 16; g = (L.off % 16) / 4;
            if (cq < 4) { j = 0; e = cq; } else { j = 1; e = cq - 4; }
          } else if (k >= 2 && L.kind == 1 && L.blk == k - 2 && L.e == 0) {        // element 0 of the block before the previous one
            j = 0; g = L.g; e = L.mt;
          } else if (k >= 2 && L.kind == 2 && L.off >= 12 * (k - 2) && L.off < 12 * (k - 2) + 12) {   // parked in this block's group
            const int n = L.off - 12 * (k - 2), pi = n % 3;
            g = n / 3; ++parked_n;
            if (pi == 0) { j = 0; e = 3; } else { j = 1; e = pi - 1; }
          } else { ok = false; break; }
          if (label[16 * j + 4 * g + e] != -1) { ok = false; break; }
          label[16 * j + 4 * g + e] = i;
        }
        if (!ok) break;
        if (k >= 2 && parked_n != 12) { ok = false; break; }
        for (int q = 0; q < 48; ++q) if (label[q] < 0) ok = false;
        if (!ok) break;
        // ---- which logical fresh channel lands in accumulator (mt, g, e)?  rowlab[16 mt + 4 g + e] = j
        int rowlab[48];
        if (k == NB - 1) {
          for (int q = 0; q < 48; ++q) rowlab[q] = q;   // last block: natural order (Z[0..47] = logical 48..95)
        } else {
          int n13 = 0, n0 = 0;
          for (int j = 0; j < 48; ++j) {
            if ((j & 3) == 0) continue;
            int slot, e;
            if (j & 1) { slot = n13 / 2; e = (n13 & 1) ? 3 : 1; ++n13; }
            else { slot = n0++; e = 0; }
            rowlab[16 * (slot / 4) + 4 * (slot % 4) + e] = j;   // slot = 4 mt + g
          }
          // the twelve j = 0 mod 4 go to elements 2: lane groups 0..2 get three values with ONE consumer each (the kernel
          // parks them with one 12-byte store), lane group 3 takes whatever is left (three dwords)
          std::vector<std::vector<int>> by_consumer(NB + 1);
          for (int j = 0; j < 48; j += 4) by_consumer[consumer(48 + j, k + 1)].push_back(j);
          std::vector<std::vector<int>> triples;
          std::vector<int> left;
          for (auto& v : by_consumer) {
            size_t t = 0;
            for (; t + 3 <= v.size(); t += 3) triples.push_back({v[t], v[t + 1], v[t + 2]});
            for (; t < v.size(); ++t) left.push_back(v[t]);
          }
          while (triples.size() > 3) { for (int q : triples.back()) left.push_back(q); triples.pop_back(); }
          if (triples.size() != 3 || left.size() != 3) { ok = false; break; }
          for (int g = 0; g < 3; ++g)
            for (int mt = 0; mt < 3; ++mt) rowlab[16 * mt + 4 * g + 2] = triples[g][mt];
          for (int mt = 0; mt < 3; ++mt) rowlab[16 * mt + 4 * 3 + 2] = left[mt];
        }
        const Folded f1k = wp.permuted_pw_inputs(f1[k], c2, c2, label);
        const Folded f2k = wp.permuted_pw_outputs(f2[k], c2, c2, rowlab);
        // ---- next activation; park positions of the element-2 values
        std::vector<ChainLoc> nxt(96);
        for (int i = 0; i < 48; ++i) nxt[i] = act[2 * i];
        int trip_base[3] = {0, 0, 0};
        if (k < NB - 1)
          for (int g = 0; g < 3; ++g) trip_base[g] = park_slots(consumer(48 + rowlab[4 * g + 2], k + 1), 3);   // its three share the consumer
        for (int q = 0; q < 48; ++q) {
          const int mt = q / 16, g = (q % 16) / 4, e = q % 4, j = rowlab[q];
          ChainLoc L; L.kind = 1; L.blk = k; L.mt = mt; L.g = g; L.e = e;
          if (k < NB - 1 && e == 2) {
            L.kind = 2;
            L.off = g < 3 ? trip_base[g] + mt : park_slots(consumer(48 + j, k + 1), 1);
            tables[k][mt * 4 + g] = L.off;
          }
          nxt[48 + j] = L;
        }
        act.swap(nxt);
        wp.append_s1_bf6(im, f1k, fd[k], f2k);
        for (int t = 0; t < 64; ++t) {                  // int tables as raw bits behind the BN vectors
          float fbits; const int v = t < 36 ? tables[k][t] : 0;
          std::memcpy(&fbits, &v, sizeof(float));
          im.push_back(fbits);
        }
      }
      if (ok) {
        for (int k = 2; k < NB; ++k) if (group_n[k] != 12) ok = false;
        // ---- where the chain's output lives in Z
        for (int pos = 0; pos < 96; ++pos) z_label[pos] = -1;
        for (int o = 0; o < 96 && ok; ++o) {
          const ChainLoc& L = act[o];
          int pos = -1;
          if (L.kind == 1 && L.blk == NB - 1) pos = 16 * L.mt + 4 * L.g + L.e;               // last block's accumulators
          else if (L.kind == 1 && L.blk == NB - 2 && L.e == 0) pos = 48 + 3 * L.g + L.mt;    // held elements of the block before
          else if (L.kind == 2 && L.off >= 60) pos = L.off;
          if (pos < 0 || z_label[pos] != -1) { ok = false; break; }
          z_label[pos] = o;
        }
        if ((int)(im.size() / NB) != yfv2_s1chain_image_floats()) ok = false;
      }
    } else {
      ok = false;
    }
    Step s;
    s.kind = STEP_S1CHAIN;
    s.c2 = c2;
    s.s1.in = x.p; s.s1.out = y.p;
    s.s1.H = H; s.s1.W = W; s.s1.R = H; s.s1.nblk = NB;
    s.s1.presplit = 1;
    s.s1.park = h->t1.p;     // a temporary of the layer-by-layer blocks: nothing else runs while the chain does
    if ((size_t)yfv2_s1chain_park_floats(H, W, NB) > h->t1.per_img) ok = false;
    s.img_off = wp.put(im);
    s.name = names.front() + " .. " + names.back().substr(names.back().rfind('.') + 1) + " chain of " + std::to_string(NB) +
             " fused s1 blocks in one launch (activations between them stay on chip)";
    s.flops = NB * 2.0 * H * W * (2.0 * c2 * c2 + 9.0 * c2);
    s.bytes = NB * 4.0 * H * W * (2.0 * c);   // per-layer accounting: every block reads and writes c channels
    s.bytes_ext = 4.0 * H * W * (2.0 * c);    // the launch reads the activation once and writes it once (parked dwords are internal traffic)
    h->plan.push_back(s);
  }

  // ---- a chain of stride-1 blocks with the whole activation resident in LDS (block_s1pool_kernel, yfv2_block.hip): natural
  // channel order, so the only host work is cutting every block's filters into the three 32-channel passes the kernel runs:
  // per pass W1 rows 32 t .. +31 (fragment-major [2][6][64][4]) | W2 columns 32 t .. +31 ([6][2][64][4]) | depthwise taps
  // [9][32] | sc1 sh1 scd shd [32] | sc2 sh2 [96]
  void s1pool_block(const std::vector<std::string>& names, int c, int H, int W, const Buf& x, const Buf& y) {
    const int c2 = c / 2, NB = (int)names.size();
    const bool pre = h->bf6;   // bf16x6 on pre-split filters; YFV2_BF6=0: the fp32-MFMA form of the same kernel
    std::vector<float> im;
    for (int k = 0; k < NB && ok; ++k) {
      Folded f1, fd, f2;
      ok &= wp.pw(names[k] + ".branch_main.0", names[k] + ".branch_main.1", c2, c2, &f1);
      ok &= wp.dw(names[k] + ".branch_main.3", names[k] + ".branch_main.4", c2, 3, &fd);
      ok &= wp.pw(names[k] + ".branch_main.5", names[k] + ".branch_main.6", c2, c2, &f2);
      if (!ok) break;
      const float* w1 = &wp.blob[f1.w]; const float* w2 = &wp.blob[f2.w]; const float* wd = &wp.blob[fd.w];
      int sw1 = 0, sw2 = 0;   // fp16x3: one power of two per filter; the BN scales below carry the exact 2^-(sw+4)
      if (pre) {
        float m1 = 0.f, m2 = 0.f;
        for (int i = 0; i < c2 * c2; ++i) { m1 = std::fmax(m1, std::fabs(w1[i])); m2 = std::fmax(m2, std::fabs(w2[i])); }
        sw1 = WeightPacker::pow2_for(m1); sw2 = WeightPacker::pow2_for(m2);
      }
      const int un1 = pre ? sw1 + 4 : 0, un2 = pre ? sw2 + 4 : 0;
      for (int t = 0; t < 3; ++t) {
        const size_t start = im.size();
        if (pre) {   // two fp16 terms x 2^sw per chunk pair (block_s1pool_kernel<.., PRE>: fp16x3)
          WeightPacker::push_h2_fn(im, 2, 3, [&](int r, int cc) { return std::ldexp(w1[(size_t)(32 * t + r) * c2 + cc], sw1); });    // W1 rows 32 t .. +31, K = 96
          WeightPacker::push_h2_fn(im, 6, 1, [&](int r, int cc) { return std::ldexp(w2[(size_t)r * c2 + 32 * t + cc], sw2); });      // W2 columns 32 t .. +31
        } else {
        for (int mt = 0; mt < 2; ++mt)
          for (int s = 0; s < 6; ++s)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 4; ++j) im.push_back(w1[(size_t)(32 * t + 16 * mt + (l & 15)) * c2 + 16 * s + 4 * (l >> 4) + j]);
        for (int mt = 0; mt < 6; ++mt)
          for (int s = 0; s < 2; ++s)
            for (int l = 0; l < 64; ++l)
              for (int j = 0; j < 4; ++j) im.push_back(w2[(size_t)(16 * mt + (l & 15)) * c2 + 32 * t + 16 * s + 4 * (l >> 4) + j]);
        }
        for (int tap = 0; tap < 9; ++tap)
          for (int ch = 0; ch < 32; ++ch) im.push_back(wd[(size_t)tap * c2 + 32 * t + ch]);
        for (const size_t* v : {&f1.scale, &f1.shift, &fd.scale, &fd.shift})
          for (int ch = 0; ch < 32; ++ch) im.push_back(v == &f1.scale ? std::ldexp(wp.blob[*v + 32 * t + ch], -un1) : wp.blob[*v + 32 * t + ch]);
        for (int ch = 0; ch < c2; ++ch) im.push_back(std::ldexp(wp.blob[f2.scale + ch], -un2));
        for (int ch = 0; ch < c2; ++ch) im.push_back(wp.blob[f2.shift + ch]);
        if ((int)(im.size() - start) != yfv2_s1pool_image_floats(pre)) ok = false;
      }
    }
    Step s;
    s.kind = STEP_S1POOL;
    s.c2 = c2;
    s.s1.in = x.p; s.s1.out = y.p;
    s.s1.H = H; s.s1.W = W; s.s1.R = H; s.s1.nblk = NB;
    s.s1.presplit = pre ? 1 : 0;
    s.img_off = wp.put(im);
    s.name = names.front() + " .. " + names.back().substr(names.back().rfind('.') + 1) + " chain of " + std::to_string(NB) +
             " fused s1 blocks in one launch (whole activation resident in LDS)";
    s.flops = NB * 2.0 * H * W * (2.0 * c2 * c2 + 9.0 * c2);
    s.bytes = NB * 4.0 * H * W * (2.0 * c);
    s.bytes_ext = 4.0 * H * W * (2.0 * c);
    h->plan.push_back(s);
  }

  // ShuffleV2Block stride 1 (shufflenetv2.py:48-51,57-63), layer by layer: even channels pass through (copied by the pw1
  // launch), odd channels -> main; out = cat(pass, main).  The general plan for shapes the chains do not cover.
  void block_s1(const std::string& p, int c, int H, int W, const Buf& x, const Buf& y) {
    Folded f;
    const int c2 = c / 2;
    ok &= wp.pw(p + ".branch_main.0", p + ".branch_main.1", c2, c2, &f);
    Step& s = add_pw(p + ".shuffle+pass+main.pw1+bn+relu", c2, PW_SHUFFLE, c2, H * W, x.p, c, 0, h->t1.p, c2, 0, true, f);
    s.pw.copy = y.p; s.pw.copy_stride = c; s.pw.copy_off = 0;
    s.bytes = 4.0 * H * W * (c + c2 + c2);  // reads both halves, writes pass-through half + pw1 output
    ok &= wp.dw(p + ".branch_main.3", p + ".branch_main.4", c2, 3, &f);
    add_dw(p + ".main.dw3x3+bn", 3, 1, c2, H, W, h->t1.p, c2, h->t2.p, c2, false, f);
    ok &= wp.pw(p + ".branch_main.5", p + ".branch_main.6", c2, c2, &f);
    add_pw(p + ".main.pw2+bn+relu", c2, PW_PLAIN, c2, H * W, h->t2.p, c2, 0, y.p, c, c2, true, f);
  }

  // Maps larger than 11x11 run a tower half per launch (towerh_kernel's 2x2-patch form).  The cls and the reg tower of a level are
  // independent of each other, so their a halves (both read the FPN map) and their b halves (each reads its own a half) go side
  // by side as workgroup ranges of ONE launch each: four launches -> two, and a CU starts its next workgroup when its current one
  // ends instead of waiting for the slowest image of the launch (YFV2_TPAIR=0: four launches).  Needs the chained output convs
  // on both towers (anchors + classes <= 96) and a second intermediate buffer (tb: unused on this path otherwise).
  bool pair_level(int H, int W) const {
    if (h->plan_sw.towers_unpaired || h->plan_sw.layer_by_layer) return false;
    return yfv2_tower2_supported(H, W) && yfv2_towerh_supported(H, W) && !yfv2_towerh_multi(H, W) && h->cfg.anchor_num + h->cfg.classes <= 96;
  }

  // DWConvblock (fpn.py:12-25) + the output convs fed by this tower (detector.py:25-31)
  void tower_half(const std::string& name, int H, int W, const float* in, float* out, const Folded& fd, const Folded& fp,
                  const Folded* fh, int mh, int split, int head0, int head1) {
    Step s;
    s.kind = STEP_TOWER;
    s.tw.in = in; s.tw.out = out;
    s.tw.H = H; s.tw.W = W;
    s.tw.mh = mh; s.tw.split = split;
    s.img_off = wp.image_tower(fd, fp, fh, mh);
    // one LDS layout per launch: where the four halves of a map size share a launch (merge_tower_launches) every image is
    // packed for the widest output conv of the level (obj + cls), else for the step's own
    // (with more than 93 classes the class head runs as separate launches: the level's halves are never merged)
    const bool merged_level = yfv2_towerh_multi(H, W) && h->cfg.anchor_num + h->cfg.classes <= 96;
    s.tw_tiles = merged_level ? ((h->cfg.anchor_num + h->cfg.classes + 15) / 16 <= 1 ? 1 : 6) : (fh ? ((mh + 15) / 16 <= 1 ? 1 : 6) : 0);
    // paired level (pair_level): the two b halves share a launch, so both are packed for the wider of the two output convs
    if (pair_level(H, W) && fh) s.tw_tiles = ((h->cfg.anchor_num + h->cfg.classes + 15) / 16 <= 1 && (4 * h->cfg.anchor_num + 15) / 16 <= 1) ? 1 : 6;
    if (yfv2_towerh_supported(H, W)) s.img_off3 = wp.image_towerh(fd, fp, fh, mh, s.tw_tiles);
    s.has_head = fh != nullptr;
    s.head0 = head0; s.head1 = head1;
    s.name = name;
    s.flops = 2.0 * H * W * (25.0 * 72 + 72.0 * 72 + (fh ? 72.0 * mh : 0.0));
    s.bytes = 4.0 * H * W * (72.0 + (fh ? mh : 72.0));
    h->plan.push_back(s);
  }

  // obj + cls output convs of a model with more than 93 classes, from the finished cls tower in h->tb: the objectness head
  // and the class head in slices of up to 96 output channels, each a PW_HEAD launch writing its channel range of the NCHW tensor
  void wide_cls_heads(const std::string& p, int px, int scale_idx) {
    const int A = h->cfg.anchor_num, nc = h->cfg.classes;
    Folded f;
    ok &= wp.heads({{"output_obj_layers", A}}, 72, &f);
    {
      Step& s = add_pw(p + " -> output_obj (bias, NCHW)", 72, PW_HEAD, A, px, h->tb.p, 72, 0, nullptr, 0, 0, false, f);
      s.pw.split = A; s.head0 = scale_idx * 3 + 1; s.head1 = -1;
    }
    for (int c0 = 0; c0 < nc; c0 += 96) {
      const int n = std::min(96, nc - c0);
      ok &= wp.heads_range("output_cls_layers", nc, c0, n, 72, &f);
      Step& s = add_pw(p + " -> output_cls channels " + std::to_string(c0) + ".." + std::to_string(c0 + n - 1) + " (bias, NCHW)", 72, PW_HEAD, n, px,
                       h->tb.p, 72, 0, nullptr, 0, 0, false, f);
      s.pw.split = n; s.pw.ctot0 = nc; s.pw.coff0 = c0; s.head0 = scale_idx * 3 + 2; s.head1 = -1;
    }
  }

  void tower(const std::string& p, int H, int W, const Buf& s_in, bool is_cls, int scale_idx) {
    Folded f;
    const int px = H * W;
    {
      if (!h->plan_sw.layer_by_layer && yfv2_tower2_supported(H, W)) {
        Folded fd1, fp1, fd2, fp2, fh;
        ok &= wp.dw(p + ".0", p + ".1", 72, 5, &fd1);
        ok &= wp.pw(p + ".3", p + ".4", 72, 72, &fp1);
        ok &= wp.dw(p + ".5", p + ".6", 72, 5, &fd2);
        ok &= wp.pw(p + ".8", p + ".9", 72, 72, &fp2);
        const int A = h->cfg.anchor_num, nc = h->cfg.classes;
        float* mid = (!is_cls && pair_level(H, W)) ? h->tb.p : h->ta.p;   // (paired level: both towers' a halves are alive at once)
        tower_half(p + " half a: dw5x5+bn+relu -> pw+bn", H, W, s_in.p, mid, fd1, fp1, nullptr, 0, 0, -1, -1);
        if (is_cls && A + nc > 96) {   // more output channels than a chained output conv holds: the tower ends in memory, the heads follow as launches
          tower_half(p + " half b: dw5x5+bn+relu -> pw+bn", H, W, mid, h->tb.p, fd2, fp2, nullptr, 0, 0, -1, -1);
          wide_cls_heads(p, px, scale_idx);
        } else if (is_cls) {
          ok &= wp.heads({{"output_obj_layers", A}, {"output_cls_layers", nc}}, 72, &fh);
          tower_half(p + " half b: dw5x5+bn+relu -> pw+bn -> output_obj+output_cls (bias, NCHW)", H, W, mid, nullptr, fd2,
                     fp2, &fh, A + nc, A, scale_idx * 3 + 1, scale_idx * 3 + 2);
        } else {
          ok &= wp.heads({{"output_reg_layers", 4 * A}}, 72, &fh);
          tower_half(p + " half b: dw5x5+bn+relu -> pw+bn -> output_reg (bias, NCHW)", H, W, mid, nullptr, fd2, fp2, &fh,
                     4 * A, 4 * A, scale_idx * 3 + 0, -1);
        }
        return;
      }
    }
    ok &= wp.dw(p + ".0", p + ".1", 72, 5, &f);
    add_dw(p + ".dw5x5+bn+relu(a)", 5, 1, 72, H, W, s_in.p, 72, h->ta.p, 72, true, f);
    ok &= wp.pw(p + ".3", p + ".4", 72, 72, &f);
    add_pw(p + ".pw+bn(a)", 72, PW_PLAIN, 72, px, h->ta.p, 72, 0, h->tb.p, 72, 0, false, f);
    ok &= wp.dw(p + ".5", p + ".6", 72, 5, &f);
    add_dw(p + ".dw5x5+bn+relu(b)", 5, 1, 72, H, W, h->tb.p, 72, h->ta.p, 72, true, f);
    ok &= wp.pw(p + ".8", p + ".9", 72, 72, &f);
    add_pw(p + ".pw+bn(b)", 72, PW_PLAIN, 72, px, h->ta.p, 72, 0, h->tb.p, 72, 0, false, f);
    const int A = h->cfg.anchor_num, nc = h->cfg.classes;
    if (is_cls && A + nc > 96) {
      wide_cls_heads(p, px, scale_idx);
    } else if (is_cls) {
      ok &= wp.heads({{"output_obj_layers", A}, {"output_cls_layers", nc}}, 72, &f);
      Step& s = add_pw(p + " -> output_obj+output_cls (bias, NCHW)", 72, PW_HEAD, A + nc, px, h->tb.p, 72, 0, nullptr,
                       0, 0, false, f);
      s.pw.split = A;
      s.head0 = scale_idx * 3 + 1;
      s.head1 = scale_idx * 3 + 2;
    } else {
      ok &= wp.heads({{"output_reg_layers", 4 * A}}, 72, &f);
      Step& s = add_pw(p + " -> output_reg (bias, NCHW)", 72, PW_HEAD, 4 * A, px, h->tb.p, 72, 0, nullptr, 0, 0, false, f);
      s.pw.split = 4 * A;
      s.head0 = scale_idx * 3 + 0;
      s.head1 = -1;
    }
  }

  // towerh_kernel's single-pixel form (maps up to 11x11) runs the four tower halves of a map size in ONE launch (each
  // workgroup: cls a, cls b, reg a, reg b of its image, in the order the separate launches had): runs of four consecutive
  // such steps become one step.
  void merge_tower_launches() {
    std::vector<Step> out;
    for (size_t i = 0; i < h->plan.size();) {
      auto mergeable = [&](const Step& t) { return t.kind == STEP_TOWER && t.img_off3 != 0 && yfv2_towerh_multi(t.tw.H, t.tw.W) && t.tw.H == h->plan[i].tw.H && t.tw.W == h->plan[i].tw.W &&
                                                   h->cfg.anchor_num + h->cfg.classes <= 96; };
      size_t n = 0;
      while (i + n < h->plan.size() && n < 4 && mergeable(h->plan[i + n])) ++n;
      if (n == 4) {
        Step m = h->plan[i];
        m.jobs.assign(h->plan.begin() + i, h->plan.begin() + i + 4);
        m.name = "fpn towers " + std::to_string(m.tw.H) + "x" + std::to_string(m.tw.W) + ": cls_head (dw5+bn+relu -> pw+bn, twice) -> output_obj+output_cls | reg_head -> output_reg, four jobs in one launch";
        m.flops = 0; m.bytes = 0;
        double ext = 0;
        for (const Step& j : m.jobs) {
          m.flops += j.flops; m.bytes += j.bytes;
          ext += 4.0 * j.tw.H * j.tw.W * (j.has_head ? (double)j.tw.mh : 72.0);   // half a reads the FPN map, half b writes logits; the 72-channel tensor between them is the launch's own scratch
        }
        m.bytes_ext = ext;
        out.push_back(m);
        i += 4;
      } else if (i + 4 <= h->plan.size() && h->plan[i].kind == STEP_TOWER && pair_level(h->plan[i].tw.H, h->plan[i].tw.W) && pairable(i) &&
                 !((h->plan[i].tw.H | h->plan[i].tw.W) & 1)) {
        // cls a, cls b, reg a, reg b  ->  ONE step {cls a, reg a, cls b, reg b} (towerp_kernel: even maps).  At batches that fill the chip it is
        // one launch whose workgroups run their image's four halves back to back (yfv2_launch_towerh decides per call: small batches run
        // the a halves and the b halves as two launches of independent items)
        Step m = h->plan[i + 1];
        m.jobs = {h->plan[i], h->plan[i + 2], h->plan[i + 1], h->plan[i + 3]};
        m.par = true;
        m.tw_tiles = std::max(h->plan[i + 1].tw_tiles, h->plan[i + 3].tw_tiles);
        m.name = "fpn towers " + std::to_string(m.tw.H) + "x" + std::to_string(m.tw.W) + ": cls_head (dw5x5+bn+relu -> pw+bn, twice) -> output_obj+output_cls | reg_head -> output_reg, the four halves of an image in one workgroup";
        m.flops = 0; m.bytes = 0;
        double ext = 0;
        for (const Step& j : m.jobs) {
          m.flops += j.flops; m.bytes += j.bytes;
          ext += 4.0 * j.tw.H * j.tw.W * (j.has_head ? (double)j.tw.mh : 72.0);   // the a halves read the FPN map, the b halves write logits; the tensors between them are the launch's own scratch
        }
        m.bytes_ext = ext;
        out.push_back(m);
        i += 4;
      } else if (i + 4 <= h->plan.size() && h->plan[i].kind == STEP_TOWER && pair_level(h->plan[i].tw.H, h->plan[i].tw.W) && pairable(i)) {
        // cls a, cls b, reg a, reg b  ->  (cls a | reg a), (cls b | reg b)
        for (int half = 0; half < 2; ++half) {
          Step m = h->plan[i + half];
          m.jobs = {h->plan[i + half], h->plan[i + 2 + half]};
          m.par = true;
          m.name = "fpn towers " + std::to_string(m.tw.H) + "x" + std::to_string(m.tw.W) + (half == 0 ? ": cls_head half a | reg_head half a (dw5x5+bn+relu -> pw+bn), side by side in one launch"
                                                                                                        : ": cls_head half b -> output_obj+output_cls | reg_head half b -> output_reg, side by side in one launch");
          m.flops = 0; m.bytes = 0;
          for (const Step& j : m.jobs) { m.flops += j.flops; m.bytes += j.bytes; }
          m.bytes_ext = -1;   // every job reads and writes memory: external = bytes
          out.push_back(m);
        }
        i += 4;
      } else {
        out.push_back(h->plan[i]);
        ++i;
      }
    }
    h->plan.swap(out);
  }
  // four consecutive tower steps of one level in the order tower() emits them, all on towerh_kernel with the same image layout per pair
  bool pairable(size_t i) const {
    const Step *ca = &h->plan[i], *cb = &h->plan[i + 1], *ra = &h->plan[i + 2], *rb = &h->plan[i + 3];
    for (const Step* t : {ca, cb, ra, rb})
      if (t->kind != STEP_TOWER || !t->img_off3 || t->tw.H != ca->tw.H || t->tw.W != ca->tw.W) return false;
    return !ca->has_head && !ra->has_head && cb->has_head && rb->has_head && ca->tw_tiles == ra->tw_tiles && cb->tw_tiles == rb->tw_tiles &&
           cb->tw.in == ca->tw.out && rb->tw.in == ra->tw.out && ca->tw.out != ra->tw.out;
  }

  void build() {
    const int H = h->cfg.height, W = h->cfg.width;
    int hh = H / 4, ww = W / 4, cin = 24;
    const long long pp_bufstride = 48LL * (H / 8) * (W / 8);   // floats from an image's copy in buffer 0 to its copy in buffer 1 (the image stride is twice that)
    const bool fused = !h->plan_sw.layer_by_layer;   // yfv2_plan.layer_by_layer: every layer its own launch (the general plan)
    const bool stage2_px = fused && h->s2pp.p && yfv2_s1px_supported(hh / 2, ww / 2) &&
                           yfv2_block_s2_rows(48, hh / 2, ww / 2) > 0;
    // the stem's output for s2h_kernel: [H/4][W/4][24] (a pixel's 96 bytes in one run: every lane group's 16-byte store lands in
    // the same 1.5 KB of a wave's row) - 126 -> 120 us against the quad planes of round 4's first half on the same box, stage2.0
    // unchanged (72.7 us either way).  The fp32-matrix plan keeps its pair planes.
    const bool stem_nhwc = stage2_px && h->bf6;
    add_stem(h->a1, stage2_px && !stem_nhwc);
    h->stem_pp = stage2_px && !stem_nhwc;
    h->front_fused = stem_nhwc && h->front_wanted;   // YFV2_FRONT=0: the stem and stage2.0 as two launches (the form every other plan and the uint8 entry points use)
    Buf* stage_bufs[3] = {h->s2, h->s3, h->s4};
    const int repeats[3] = {4, 8, 4};
    const Buf* x = &h->a1;
    h->dbg[0] = h->a1.p; h->dbg_per_img[0] = h->a1.per_img; h->dbg_c[0] = 24;
    Stage2Layout L2{};
    bool px_pending = false;   // the next stride-2 block reads stage 2's pair planes
    for (int si = 0; si < 3; ++si) {
      const int cout = cin * 2;
      int cur = 0;
      const bool use_px = si == 0 && stage2_px;
      for (int i = 0; i < repeats[si]; ++i) {
        const std::string p = "backbone.stage" + std::to_string(si + 2) + "." + std::to_string(i);
        const Buf* y = &stage_bufs[si][cur];
        if (i == 0) {
          if (use_px) s2px_block(p, hh, ww);
          else if (px_pending) block_s2(p, cin, hh, ww, h->s2pp, *y, L2.label, L2.buf, pp_bufstride);
          else if (si == 2 && h->c2_permuted) block_s2(p, cin, hh, ww, *x, *y, nullptr, nullptr, 0, h->c2_label);
          else block_s2(p, cin, hh, ww, *x, *y);
          px_pending = false;
          hh /= 2; ww /= 2;
          if (use_px) {   // logical channel c sits at slot(c) of buffer 0
            for (int k = 0; k < 48; ++k) L2.label[k] = yfv2_stage2_channel(k);
            for (int q = 0; q < 24; ++q) L2.buf[q] = 0;
          }
        } else if (use_px) {
          s1px_block(p, hh, ww, L2, pp_bufstride);
        } else if (fused && h->bf6 && i == 1 && repeats[si] == 8 && yfv2_s1chain_supported(cout / 2, hh, ww)) {
          std::vector<std::string> names;
          for (int q = 1; q < repeats[si]; ++q) names.push_back("backbone.stage" + std::to_string(si + 2) + "." + std::to_string(q));
          s1chain_block(names, cout, hh, ww, *x, *y, h->c2_label);     // blocks 1..7 of the stage as one launch
          h->c2_permuted = ok;
          i = repeats[si] - 1;
        } else if (fused && i == 1 && si == 2 && yfv2_s1pool_supported(cout / 2, hh, ww)) {
          std::vector<std::string> names;
          for (int q = 1; q < repeats[si]; ++q) names.push_back("backbone.stage" + std::to_string(si + 2) + "." + std::to_string(q));
          s1pool_block(names, cout, hh, ww, *x, *y);                     // stage 4's blocks 1..3 as one launch
          i = repeats[si] - 1;
        } else {
          block_s1(p, cout, hh, ww, *x, *y);
        }
        x = y;
        cur ^= 1;
      }
      if (use_px) {
        px_pending = true;
        h->s2_px = true;
        for (int k = 0; k < 48; ++k) h->s2_label[k] = L2.label[k];
        for (int q = 0; q < 24; ++q) h->s2_buf[q] = L2.buf[q];
        h->dbg[1] = h->s2pp.p; h->dbg_per_img[1] = (size_t)48 * hh * ww; h->dbg_c[1] = cout;
      } else {
        h->dbg[1 + si] = x->p; h->dbg_per_img[1 + si] = x->per_img; h->dbg_c[1 + si] = cout;
      }
      cin = cout;
    }
    const Buf* c2 = nullptr; const Buf* c3 = x;
    // stage3 output: 8 blocks -> last written buffer index is (8-1)&1 ... recover from dbg
    Buf c2b; c2b.p = h->dbg[2]; c2b.per_img = h->dbg_per_img[2]; c2 = &c2b;
    const int h3 = H / 32, w3 = W / 32, h2 = H / 16, w2 = W / 16;
    Folded f;
    ok &= wp.pw("fpn.conv1x1_3.0", "fpn.conv1x1_3.1", 72, 192, &f);
    add_pw("fpn.conv1x1_3 pw192->72+bn+relu", 192, PW_PLAIN, 72, h3 * w3, c3->p, 192, 0, h->f3.p, 72, 0, true, f);
    ok &= wp.pw("fpn.conv1x1_2.0", "fpn.conv1x1_2.1", 72, 288, &f);
    if (h->c2_permuted && ok) {   // columns 192.. read C2 in the chain kernel's channel order
      int lab[288];
      for (int k = 0; k < 192; ++k) lab[k] = k;
      for (int k = 0; k < 96; ++k) lab[192 + k] = 192 + h->c2_label[k];
      f = wp.permuted_pw_inputs(f, 72, 288, lab);
    }
    // Default plan (round 6): a 1x1 conv commutes with the nearest-neighbour upsample (fpn.py:57-59), so conv1x1_2's 192 upsampled channels
    // are applied ONCE per coarse pixel - Q = scale2 (W2[:, :192] C3) + shift2, by the launch that computes conv1x1_3 from the same C3 - and
    // the fine-map launch is a K = 96 conv over C2 whose epilogue adds Q at (y / 2, x / 2): a third of the matrix-core work and 25 MB less
    // traffic than the K = 288 form (which the fp32-matrix and the layer-by-layer plans keep).  Like the BatchNorm folding a
    // re-association inside one linear map: S2 = relu(scale2 (W2b C2) + Q) instead of relu(scale2 (W2a up(C3) + W2b C2) + shift2).
    const bool fpn_split = fused && h->bf6 && ok && h->fq.p && yfv2_pw_presplit_supported(192, PW_DUAL, 72) && yfv2_pw_presplit_supported(96, PW_FPNQ, 72);
    if (fpn_split) {
      Step& s3 = h->plan.back();                                     // conv1x1_3 just added: it becomes the dual launch
      Folded fa = wp.pw_columns(f, 72, 288, 0, 192), fb = wp.pw_columns(f, 72, 288, 192, 96);
      Folded f3;
      ok &= wp.pw("fpn.conv1x1_3.0", "fpn.conv1x1_3.1", 72, 192, &f3);
      s3.mode = PW_DUAL;
      s3.pw.copy = h->fq.p; s3.pw.copy_stride = 72; s3.pw.copy_off = 0;
      s3.pw.presplit = 1;
      s3.img_off = wp.image_pw_dual(f3, fa, 72, 192, 5);
      s3.name = "fpn.conv1x1_3 pw192->72+bn+relu | the C3 part of fpn.conv1x1_2 (W2[:, :192] C3, scale + shift of its bn), one launch";
      Step& s = add_pw("fpn.conv1x1_2 pw96 over C2 + the C3 part at (y/2, x/2) +bn+relu  [= up2x(C3)+cat(C2)+pw288->72+bn+relu]", 96, PW_FPNQ, 72, h2 * w2,
                       c2->p, 96, 0, h->f2.p, 72, 0, true, fb);
      s.pw.in2 = h->fq.p;
      s.pw.H = h2; s.pw.W = w2;
      s.flops = 2.0 * h2 * w2 * 288 * 72;                             // (the reference layer's count, as for every fused or re-associated launch)
      s.bytes = 4.0 * (h3 * w3 * 192.0 + h2 * w2 * 96.0 + h2 * w2 * 72.0);
    } else {
      Step& s = add_pw("fpn.conv1x1_2 up2x(C3)+cat(C2)+pw288->72+bn+relu", 288, PW_FPN, 72, h2 * w2, c3->p, 192, 0,
                       h->f2.p, 72, 0, true, f);
      s.pw.in2 = c2->p;
      s.pw.H = h2; s.pw.W = w2;
      s.bytes = 4.0 * (h3 * w3 * 192.0 + h2 * w2 * 96.0 + h2 * w2 * 72.0);
    }
    h->dbg[4] = h->f2.p; h->dbg_per_img[4] = h->f2.per_img; h->dbg_c[4] = 72;
    h->dbg[5] = h->f3.p; h->dbg_per_img[5] = h->f3.per_img; h->dbg_c[5] = 72;
    tower("fpn.cls_head_3.block", h3, w3, h->f3, true, 1);
    tower("fpn.reg_head_3.block", h3, w3, h->f3, false, 1);
    tower("fpn.cls_head_2.block", h2, w2, h->f2, true, 0);
    tower("fpn.reg_head_2.block", h2, w2, h->f2, false, 0);
    merge_tower_launches();
    if (h->front_fused && ok && h->plan.size() > 1 && h->plan[0].kind == STEP_STEM && h->plan[1].kind == STEP_S2PX && h->plan[1].img_off3) {
      h->stem_aside = h->plan[0];
      h->plan.erase(h->plan.begin());
      Step& f = h->plan[0];
      f.front = true;
      f.name_plain = f.name;
      f.name = "stem + backbone.stage2.0 in one launch: conv3x3s2+bn+relu+maxpool3x3s2 -> s2 block, lane-per-pixel (proj | main) -> pair planes";
      f.flops += h->stem_aside.flops;
      f.bytes += h->stem_aside.bytes;                     // per-layer accounting: both layers' reads and writes
      f.bytes_ext = 4.0 * (3.0 * H * W + 48.0 * (H / 8) * (W / 8));   // the image in, stage 2's 48 channels out
    } else {
      h->front_fused = false;
    }
  }
};

// kernel (family) a plan step launches, as it appears in a rocprofv3 kernel trace (prefix of the symbol name)
std::string step_kernel(const Step& st) {
  switch (st.kind) {
    case STEP_STEM: return "stem_h3_kernel";   // fp32 input, default plan (uint8 input: stem_h3u_kernel; YFV2_BF6=0: stem_px_kernel)
    case STEP_PW: return "pw_kernel<" + std::to_string(st.K) + ",";
    case STEP_DW: return "dw_kernel<" + std::to_string(st.ksize) + ", " + std::to_string(st.stride) + ">";
    case STEP_TOWER:
      if (st.img_off3 && !st.jobs.empty() && !st.par) return "towers_kernel<" + std::to_string(st.tw_tiles) + ">";   // default plan, maps up to 11x11
      if (st.img_off3 && (st.tw.H > 11 || st.tw.W > 11) && !((st.tw.H | st.tw.W) & 1))   // default plan, even maps up to 22x22
        return "towerp_kernel<" + std::to_string(st.tw_tiles) + (st.par && st.jobs.size() == 4 ? ", true>" : ", false>");
      if (st.img_off3) return "towerh_kernel<" + std::to_string(st.tw_tiles) + ", " + (st.tw.H > 11 || st.tw.W > 11 ? "2, 4>" : "1, 1>");
      return "tower2_kernel<" + std::to_string(!st.has_head ? 0 : ((st.tw.mh + 15) / 16 <= 1 ? 1 : 6)) + ", 512, " + (st.tw.H * st.tw.W > 128 ? "4, 4," : "1, 1,");
    case STEP_S2: return st.img_off3 ? std::string(st.c2 == 96 ? "s4h_kernel" : "s3h2_kernel") : (st.c2 == 96 ? std::string("block_s2w_kernel<") : "block_s2_kernel<" + std::to_string(st.c2) + ",");
    case STEP_S1PX: return "s1h_kernel";   // default plan (YFV2_BF6=0: s1px_kernel)
    case STEP_S2PX: return st.front ? "front2_kernel" : "s2h_kernel";   // default plan (YFV2_BF6=0: s2px_proj_kernel + s2px_main_kernel; uint8 input under front: stem_h3u_kernel + s2h_kernel)
    case STEP_S1CHAIN: return "block_s1chain6_kernel";
    case STEP_S1POOL: return "block_s1pool_kernel";
  }
  return "?";
}

// The plan switches, read from the environment when a handle is created (and by the host-only dry runs):
//   YFV2_FUSED=0     every reference layer its own launch (the general plan; also what shapes outside a fused kernel's
//                    static bounds get, block by block)                                 - read by PlanBuilder::build
//   YFV2_BF6=0       every pointwise conv on the fp32 MFMA; blocks whose fused kernel exists only in the bf16x6 form
//                    (the stage-3 chain, stage4.0) then run layer by layer
//   YFV2_POSTFUSE=0  yfv2_detect decodes and suppresses in two launches
//   YFV2_FRONT=0     the stem and stage2.0 as two launches (stem_h3_kernel, s2h_kernel) instead of front_kernel's one
//   YFV2_TPAIR=0     the tower halves of a level larger than 11x11 as four launches instead of two side-by-side pairs
//                                                                                       - read by PlanBuilder::pair_level
void read_plan_switches(yfv2_ctx* h, const yfv2_plan* plan) {
  h->plan_sw = yfv2_plan{};
  if (plan) {   // (a caller built against an older, shorter struct: the fields it does not have stay 0)
    const size_t n = plan->struct_size > 0 && (size_t)plan->struct_size < sizeof(yfv2_plan) ? (size_t)plan->struct_size : sizeof(yfv2_plan);
    std::memcpy(&h->plan_sw, plan, n);
  }
  h->plan_sw.struct_size = (int32_t)sizeof(yfv2_plan);
  h->bf6 = !h->plan_sw.fp32_matrix;
  h->postfuse = !h->plan_sw.post_two_launches;
  h->front_wanted = !h->plan_sw.front_two_launches;
}

int alloc_buf(yfv2_ctx* h, Buf* b, size_t per_img) {
  b->per_img = per_img;
  HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&b->p), per_img * sizeof(float) * (size_t)h->cfg.max_batch));
  return YFV2_OK;
}

void free_buf(Buf* b) {
  if (b->p) (void)hipFree(b->p);
  b->p = nullptr;
}

size_t logit_elems(const yfv2_ctx* h, int i) {
  const int sc = i / 3, k = i % 3;
  const int c = k == 0 ? 4 * h->cfg.anchor_num : (k == 1 ? h->cfg.anchor_num : h->cfg.classes);
  return (size_t)c * h->fh[sc] * h->fw[sc];
}

int run_plan(yfv2_ctx* h, const void* x, bool x_u8, int B, float* const out6[6], hipStream_t main_stream, hipEvent_t* ev /*nullable: 2 per step*/,
             int only_step = -1 /* >= 0: this launch alone (yfv2_debug_repeat_step) */) {
  // the front kernels read the image with 16-byte (fp32) / 12-byte-at-4-byte-alignment (uint8) buffer loads: a base address that
  // is not so aligned would be read at the wrong offsets without any fault (include/yfv2.h yfv2_forward)
  if (reinterpret_cast<uintptr_t>(x) & (x_u8 ? 3u : 15u))
    return fail(h, YFV2_ERR_ARG, x_u8 ? "input images: the uint8 tensor must be 4-byte aligned" : "input images: the fp32 tensor must be 16-byte aligned");
  const float* params = h->d_params;
  const hipStream_t s = main_stream;
  struct ProbeScope { ~ProbeScope() { yfv2_launch_probe = Yfv2LaunchProbe{}; } } probe_scope;   // (cleared on every path out, error returns included)
  for (size_t i = 0; i < h->plan.size(); ++i) {
    if (only_step >= 0 && (int)i != only_step) continue;
    Step& st = h->plan[i];
    yfv2_launch_probe = ev ? Yfv2LaunchProbe{ev[2 * i], ev[2 * i + 1], 0} : Yfv2LaunchProbe{};   // (profile pass: the step's launches stamp themselves)
    auto stem_args = [&](const Step& ss) {
      StemArgs a = ss.stem;
      a.x = x; a.B = B; a.u8_in = x_u8 ? 1 : 0;
      a.img = params + ss.img_off;
      a.img_u8 = params + ss.img_off2;
      a.img16 = h->bf6 ? params + ss.img_off3 : nullptr;
      a.nonfinite = h->d_nonfinite;
      return a;
    };
    if (st.kind == STEP_STEM) {
      StemArgs a = st.stem;
      a.x = x; a.B = B; a.u8_in = x_u8 ? 1 : 0;
      a.img = params + st.img_off;
      a.img_u8 = params + st.img_off2;
      a.img16 = h->bf6 ? params + st.img_off3 : nullptr;   // YFV2_BF6=0: the 4x4x1 fp32-MFMA stem
      a.nonfinite = h->d_nonfinite;
      yfv2_launch_stem(a, s);
    } else if (st.kind == STEP_PW) {
      PwArgs a = st.pw;
      a.P = B * st.px_per_img;
      a.img = params + st.img_off;
      a.bf6 = h->bf6 ? 1 : 0;
      a.nonfinite = h->d_nonfinite;
      if (st.mode == PW_HEAD) {
        a.nchw0 = out6[st.head0];
        a.nchw1 = st.head1 >= 0 ? out6[st.head1] : nullptr;
      }
      if (!yfv2_launch_pw(st.K, st.mode, a, s))
        return fail(h, YFV2_ERR_CONFIG, "no pointwise kernel for step '" + st.name + "'");
    } else if (st.kind == STEP_S2) {
      BlockS2Args a = st.s2;
      a.B = B;
      a.img = params + st.img_off;
      a.bf6 = h->bf6 ? 1 : 0;
      a.trace = (h->trace_step == (int)i) ? h->d_trace : nullptr;
      a.img16 = (st.img_off3 && h->bf6) ? params + st.img_off3 : nullptr;
      a.nonfinite = h->d_nonfinite;
      if (a.img16 && st.c2 == 48) yfv2_launch_s3h(a, s);
      else if (a.img16 && st.c2 == 96) yfv2_launch_s4h(a, s);
      else if (!yfv2_launch_block_s2(st.c2, a, s))
        return fail(h, YFV2_ERR_CONFIG, "no fused stride-2 kernel for step '" + st.name + "'");
    } else if (st.kind == STEP_TOWER) {
      auto args_of = [&](const Step& t) {
        TowerArgs a = t.tw;
        a.B = B;
        a.img = params + t.img_off;
        a.has_head = t.has_head ? 1 : 0;
        a.nchw0 = nullptr; a.nchw1 = nullptr;
        a.trace = (h->trace_step == (int)i) ? h->d_trace : nullptr;
        a.bf6 = h->bf6 ? 1 : 0;
        a.img16 = (t.img_off3 && h->bf6) ? params + t.img_off3 : nullptr;   // YFV2_BF6=0: tower2_kernel on the fp32 MFMA
        a.nonfinite = h->d_nonfinite;
        if (t.has_head) {
          a.nchw0 = out6[t.head0];
          a.nchw1 = t.head1 >= 0 ? out6[t.head1] : nullptr;
        }
        return a;
      };
      bool done = false;
      if (st.img_off3 && h->bf6) {
        TowerJobs jobs{};
        if (st.jobs.empty()) { jobs.j[0] = args_of(st); jobs.n = 1; }
        else { jobs.n = (int)st.jobs.size(); for (int k = 0; k < jobs.n; ++k) jobs.j[k] = args_of(st.jobs[k]); }
        // half a -> half b of a tower inside one launch: the tensor between them stays in the workgroup's LDS - as long as every
        // workgroup has ONE image (the job loop is outside the image loop)
        jobs.par = st.par ? 1 : 0;
        for (int k = 0; k + 1 < jobs.n && !st.par; ++k)
          if (B <= 256 && !jobs.j[k].has_head && jobs.j[k].out == jobs.j[k + 1].in) { jobs.j[k].chain |= 2; jobs.j[k + 1].chain |= 1; }
        done = yfv2_launch_towerh(jobs, st.tw_tiles, s);
      }
      if (!done) {   // tower2_kernel, one launch per half
        const size_t nj = st.jobs.empty() ? 1 : st.jobs.size();
        for (size_t k = 0; k < nj; ++k)
          if (!yfv2_launch_tower2(args_of(st.jobs.empty() ? st : st.jobs[k]), s))
            return fail(h, YFV2_ERR_CONFIG, "no tower kernel for step '" + st.name + "'");
      }
    } else if (st.kind == STEP_S1POOL) {
      BlockS1Args a = st.s1;
      a.B = B;
      a.img = params + st.img_off;
      a.trace = (h->trace_step == (int)i) ? h->d_trace : nullptr;
      a.nonfinite = h->d_nonfinite;
      if (!yfv2_launch_block_s1pool(a, s))
        return fail(h, YFV2_ERR_CONFIG, "no pool-chain kernel for step '" + st.name + "'");
    } else if (st.kind == STEP_S1CHAIN) {
      BlockS1Args a = st.s1;
      a.B = B;
      a.img = params + st.img_off;
      a.trace = (h->trace_step < 0 || h->trace_step == (int)i) ? h->d_trace : nullptr;
      a.nonfinite = h->d_nonfinite;
      if (!yfv2_launch_block_s1chain(a, s))
        return fail(h, YFV2_ERR_CONFIG, "no chain kernel for step '" + st.name + "'");
    } else if (st.kind == STEP_S2PX) {
      S2PxArgs a = st.s2px;
      a.B = B;
      a.img[0] = params + st.img_off;
      a.img[1] = params + st.img_off2;
      a.img16 = h->bf6 ? params + st.img_off3 : nullptr;   // YFV2_BF6=0: the two role kernels on the 4x4x1 fp32 MFMA
      a.nonfinite = h->d_nonfinite;
      if (st.front) {
        FrontArgs f{};
        f.x = x; f.H = h->cfg.height; f.W = h->cfg.width; f.u8_in = x_u8 ? 1 : 0;
        f.img_stem = params + h->stem_aside.img_off3;
        f.s2 = a;
        yfv2_launch_front(f, s);
      } else {
        if (st.front) yfv2_launch_stem(stem_args(h->stem_aside), s);   // uint8 input: stem_h3u_kernel, then stage2.0 on its own
        yfv2_launch_s2px(a, s);
      }
    } else if (st.kind == STEP_S1PX) {
      S1PxArgs a = st.s1px;
      a.B = B;
      a.img = params + st.img_off;
      a.img16 = h->bf6 ? params + st.img_off2 : nullptr;   // YFV2_BF6=0: s1px_kernel on the 4x4x1 fp32 MFMA
      a.nonfinite = h->d_nonfinite;
      yfv2_launch_s1px(a, s);
    } else {
      DwArgs a = st.dw;
      a.B = B;
      a.w = params + st.w_off; a.scale = params + st.scale_off; a.shift = params + st.shift_off;
      if (!yfv2_launch_dw(st.ksize, st.stride, a, s))
        return fail(h, YFV2_ERR_CONFIG, "no depthwise kernel for step '" + st.name + "'");
    }
    yfv2_launch_probe = Yfv2LaunchProbe{};
  }
  if (only_step < 0) { h->last_x = x; h->last_B = B; h->last_u8 = x_u8; }
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int check_call(yfv2_ctx* h, int B, bool need_weights) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (B < 1 || B > h->cfg.max_batch)
    return fail(h, YFV2_ERR_BATCH, "batch " + std::to_string(B) + " outside [1, max_batch=" + std::to_string(h->cfg.max_batch) + "]");
  if (need_weights && !h->weights_loaded) return fail(h, YFV2_ERR_STATE, "yfv2_load_weights has not been called");
  return YFV2_OK;
}

// configuration checks shared by yfv2_create and the host-only dry run; returns the decode row count through *rows
int check_config(const yfv2_config* cfg, int* rows_out) {
  if (cfg->anchor_num != 3)
    return fail(nullptr, YFV2_ERR_CONFIG, "anchor_num must be 3 (the reference decode hard-codes 3 anchors per scale)");
  // The reference sizes everything from its .data file (utils/utils.py:13-65).  What is static here: the class index travels
  // as one byte through the NMS kernel (255 classes), and that kernel sorts an image's candidates in LDS (4096 decode rows =
  // any input up to 512x512, or e.g. 640x384).  Shapes and class counts outside the fused kernels' own bounds run on the
  // general plan, block by block (PlanBuilder); decode + NMS then run as two launches.
  if (cfg->classes < 1 || cfg->classes > 255)
    return fail(nullptr, YFV2_ERR_CONFIG, "classes must be in [1, 255]");
  if (cfg->height < 32 || cfg->width < 32 || cfg->height % 32 || cfg->width % 32)
    return fail(nullptr, YFV2_ERR_CONFIG, "height/width must be multiples of 32");
  if (cfg->max_batch < 1) return fail(nullptr, YFV2_ERR_CONFIG, "max_batch must be >= 1");
  const long long rows_ll = 3LL * ((long long)(cfg->height / 16) * (cfg->width / 16) + (long long)(cfg->height / 32) * (cfg->width / 32));
  if (rows_ll > yfv2_nms_max_rows())
    return fail(nullptr, YFV2_ERR_CONFIG, "more than " + std::to_string(yfv2_nms_max_rows()) + " decode rows per image (" + std::to_string(rows_ll) +
                                          ": inputs beyond 512x512) is not supported by the NMS kernel");
  const int rows = (int)rows_ll;
  *rows_out = rows;
  return YFV2_OK;
}

// geometry + workspace of a handle; `alloc` is alloc_buf (device memory) or the dry run's address generator
template <class Alloc>
int setup_ctx(yfv2_ctx* h, const yfv2_config* cfg, int rows, Alloc alloc) {
  h->cfg = *cfg;
  h->device = cfg->device;
  h->rows = rows;
  h->fh[0] = cfg->height / 16; h->fw[0] = cfg->width / 16;
  h->fh[1] = cfg->height / 32; h->fw[1] = cfg->width / 32;
  const size_t H = cfg->height, W = cfg->width;
  int rc = YFV2_OK;
  auto A = [&](Buf* b, size_t n) { if (rc == YFV2_OK) rc = alloc(h, b, n); };
  A(&h->a1, (H / 4) * (W / 4) * 24);
  for (int i = 0; i < 2; ++i) {
    A(&h->s2[i], (H / 8) * (W / 8) * 48);
    A(&h->s3[i], (H / 16) * (W / 16) * 96);
    A(&h->s4[i], (H / 32) * (W / 32) * 192);
  }
  A(&h->s2pp, 2 * (H / 8) * (W / 8) * 48);
  A(&h->t1, (H / 4) * (W / 4) * 24);
  A(&h->t2, (H / 4) * (W / 4) * 24);
  A(&h->t3, (H / 4) * (W / 4) * 24);
  A(&h->f2, (H / 16) * (W / 16) * 72);
  A(&h->f3, (H / 32) * (W / 32) * 72);
  A(&h->fq, (H / 32) * (W / 32) * 72);   // the C3 part of fpn.conv1x1_2 (PW_DUAL -> PW_FPNQ)
  A(&h->ta, (H / 16) * (W / 16) * 72);
  A(&h->tb, (H / 16) * (W / 16) * 72);
  for (int i = 0; i < 6; ++i) A(&h->logits[i], logit_elems(h, i));
  A(&h->cand, (size_t)rows * 8);
  return rc;
}

constexpr int LANES_DEFAULT = 1, LANES_MAX = 8, LANE_MIN_IMAGES = 32;   // per slice: below that a slice is pure latency (tools/scale_probe.py)

int create_lanes(yfv2_ctx* h) {
  int n = h->plan_sw.lanes > 1 ? h->plan_sw.lanes : LANES_DEFAULT;
  if (h->d_trace || h->in_lane) n = 1;             // cycle stamps are taken on the parent's own launches
  n = n < 1 ? 1 : (n > LANES_MAX ? LANES_MAX : n);
  if (n == 1 || h->cfg.max_batch < n * LANE_MIN_IMAGES) return YFV2_OK;
  h->lane_min = n * LANE_MIN_IMAGES;
  yfv2_config c = h->cfg;
  c.max_batch = (h->cfg.max_batch + n - 1) / n;
  for (int i = 0; i < n; ++i) {
    yfv2_ctx* lane = nullptr;
    yfv2_plan lp = h->plan_sw;            // a lane runs the parent's plan, without lanes or stamps of its own
    lp.lanes = 0; lp.trace = 0;
    g_creating_lane = true;
    const int rc = yfv2_create_ex(&lane, &c, &lp);
    g_creating_lane = false;
    if (rc) return fail(h, rc, "lane " + std::to_string(i) + ": " + g_tls_error);
    lane->d_nonfinite = h->d_nonfinite;   // the parent's word (its h_nonfinite stays null: only the parent owns and frees it)
    h->lanes.push_back(lane);
  }
  HIP_TRY(h, hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
  for (int i = 1; i < n; ++i) {
    hipStream_t st; hipEvent_t ev;
    HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));   // non-blocking: no implicit ordering against the NULL stream, events only
    h->lane_stream.push_back(st);
    HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    h->lane_join.push_back(ev);
  }
  return YFV2_OK;
}

// run `f(lane, first image, images, stream)` for every slice of a batch of B images; lane 0 on the caller's stream `s`
template <class F>
int run_lanes(yfv2_ctx* h, int B, hipStream_t s, F f) {
  const int n = (int)h->lanes.size();
  const int per = (B + n - 1) / n;
  h->last_split.clear();
  h->last_x = nullptr; h->last_B = 0;   // (this call's input is recorded slice by slice on the lanes; yfv2_debug_activation follows last_split - the
                                        // parent's own record would be a stale pointer)
  HIP_TRY(h, hipEventRecord(h->lane_fork, s));   // nothing is enqueued on a lane stream yet: a plain return is safe here
  int rc = YFV2_OK;
  auto hip_ok = [&](hipError_t e, const char* what) {     // a HIP error inside the fork / join region must NOT return early:
    if (e != hipSuccess && rc == YFV2_OK) rc = fail(h, YFV2_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return e == hipSuccess;                               // work already queued on the lane streams still has to be joined
  };
  std::vector<char> forked((size_t)n, 0);
  // lanes 1.. first: their launches are queued behind the fork before lane 0's own launches occupy the caller's stream
  for (int i = 1; i < n && rc == YFV2_OK; ++i) {
    const int off = i * per, cnt = std::min(per, B - off);
    if (cnt <= 0) break;
    if (!hip_ok(hipStreamWaitEvent(h->lane_stream[i - 1], h->lane_fork, 0), "hipStreamWaitEvent(lane fork)")) break;
    forked[(size_t)i] = 1;
    const int lrc = f(h->lanes[i], off, cnt, h->lane_stream[i - 1]);
    if (lrc) { rc = lrc; h->err = h->lanes[i]->err; }
  }
  if (rc == YFV2_OK) {
    rc = f(h->lanes[0], 0, std::min(per, B), s);
    if (rc) h->err = h->lanes[0]->err;
  }
  // the join, on every path: whatever reached a lane stream (a slice that failed half way included) is ordered in front of
  // the caller's later use of the output and workspace buffers
  for (int i = 1; i < n; ++i) {
    if (!forked[(size_t)i]) continue;
    if (hip_ok(hipEventRecord(h->lane_join[i - 1], h->lane_stream[i - 1]), "hipEventRecord(lane join)"))
      hip_ok(hipStreamWaitEvent(s, h->lane_join[i - 1], 0), "hipStreamWaitEvent(lane join)");
    else
      (void)hipStreamSynchronize(h->lane_stream[i - 1]);   // no event to wait for: the host waits instead
  }
  if (rc == YFV2_OK)
    for (int i = 0; i < n; ++i) { const int cnt = std::min(per, B - i * per); if (cnt > 0) h->last_split.push_back(cnt); }
  return rc;
}

bool use_lanes(const yfv2_ctx* h, int B) { return !h->lanes.empty() && B >= h->lane_min; }

}  // namespace

void** yfv2_ctx_train_slot(yfv2_ctx* h) { return h ? &h->train : nullptr; }
const yfv2_config* yfv2_ctx_config(yfv2_ctx* h) { return &h->cfg; }
int yfv2_ctx_fail(yfv2_ctx* h, int code, const char* msg) { return fail(h, code, msg ? msg : ""); }

// ===========================================================================
// extern "C" surface
// ===========================================================================
extern "C" {

int yfv2_abi_version(void) { return YFV2_ABI_VERSION; }

const char* yfv2_last_error(yfv2_handle h) { return h ? h->err.c_str() : g_tls_error.c_str(); }

int yfv2_create(yfv2_handle* out, const yfv2_config* cfg) { return yfv2_create_ex(out, cfg, nullptr); }

int yfv2_create_ex(yfv2_handle* out, const yfv2_config* cfg, const yfv2_plan* plan) {
  if (!out || !cfg) return fail(nullptr, YFV2_ERR_ARG, "yfv2_create: null argument");
  *out = nullptr;
  int rows = 0;
  if (int rc = check_config(cfg, &rows)) return rc;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return fail(nullptr, YFV2_ERR_DEVICE, "no usable HIP device (this library has no CPU fallback)");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
    return fail(nullptr, YFV2_ERR_DEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, YFV2_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");

  yfv2_ctx* h = new yfv2_ctx();
  h->in_lane = g_creating_lane;
  DeviceGuard guard(cfg->device);
  int rc = setup_ctx(h, cfg, rows, alloc_buf);
  if (rc == YFV2_OK && hipMalloc(reinterpret_cast<void**>(&h->d_classes), 258 * sizeof(int32_t)) != hipSuccess)
    rc = fail(h, YFV2_ERR_DEVICE, "hipMalloc(class filter) failed");
  if (rc == YFV2_OK) {
    h->d_stats_flag = h->d_classes + 256;
    if (hipMemset(h->d_stats_flag, 0, 2 * sizeof(int32_t)) != hipSuccess) rc = fail(h, YFV2_ERR_DEVICE, "hipMemset(flags) failed");
  }
  if (rc == YFV2_OK && !h->in_lane) {
    void* hp = nullptr; void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
      if (hp) (void)hipHostFree(hp);
      rc = fail(h, YFV2_ERR_DEVICE, "hipHostMalloc(range-guard word) failed");
    } else {
      h->h_nonfinite = static_cast<int32_t*>(hp);
      h->d_nonfinite = static_cast<int32_t*>(dp);
      *reinterpret_cast<volatile int32_t*>(h->h_nonfinite) = 0;
    }
  }
  if (rc != YFV2_OK) {
    g_tls_error = h->err;
    yfv2_destroy(h);
    return rc;
  }
  read_plan_switches(h, plan);
  if (h->plan_sw.trace) {
    h->trace_step = h->plan_sw.trace_step;
    (void)hipMalloc(reinterpret_cast<void**>(&h->d_trace), 8192 * sizeof(long long));
    (void)hipMemset(h->d_trace, 0, 8192 * sizeof(long long));
  }
  if (int rc2 = create_lanes(h)) { g_tls_error = h->err; yfv2_destroy(h); return rc2; }
  *out = h;
  return YFV2_OK;
}

// Host-only test hook (CPU suite): validate `cfg`, build the launch plan and pack the weights exactly as
// yfv2_create + yfv2_load_weights do, but without a device - the workspace gets made-up addresses that are only ever
// used for pointer arithmetic.  Reports the number of launches and the size of the packed parameter blob.
int yfv2_debug_plan_dryrun(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t* n_steps, int64_t* blob_floats) {
  return yfv2_debug_plan_dryrun_ex(cfg, nullptr, tensors, n, n_steps, blob_floats);
}

int yfv2_debug_plan_dryrun_ex(const yfv2_config* cfg, const yfv2_plan* plan, const yfv2_tensor_desc* tensors, int32_t n, int32_t* n_steps, int64_t* blob_floats) {
  if (!cfg || !tensors || n <= 0) return fail(nullptr, YFV2_ERR_ARG, "yfv2_debug_plan_dryrun: bad argument");
  int rows = 0;
  if (int rc = check_config(cfg, &rows)) return rc;
  yfv2_ctx ctx;
  uintptr_t next = 0x100000000ull;
  auto fake = [&](yfv2_ctx* hh, Buf* b, size_t per_img) {
    b->per_img = per_img;
    b->p = reinterpret_cast<float*>(next);
    next += (per_img * sizeof(float) * (size_t)hh->cfg.max_batch + 4095) & ~(uintptr_t)4095;
    return (int)YFV2_OK;
  };
  setup_ctx(&ctx, cfg, rows, fake);
  read_plan_switches(&ctx, plan);
  WeightPacker wp;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) wp.byname[tensors[i].name] = &tensors[i];
  PlanBuilder pb{&ctx, wp};
  pb.build();
  if (!pb.ok || !wp.missing.empty()) return fail(nullptr, YFV2_ERR_WEIGHTS, wp.missing.empty() ? "weight packing failed" : wp.missing);
  for (const Step& st : ctx.plan)
    if (st.img_off > wp.blob.size()) return fail(nullptr, YFV2_ERR_WEIGHTS, "step '" + st.name + "': image offset outside the blob");
  if (n_steps) *n_steps = (int32_t)ctx.plan.size();
  if (blob_floats) *blob_floats = (int64_t)wp.blob.size();
  return YFV2_OK;
}

// Host-only test hook: the packed LDS image of launch `step` (or of one of its jobs, see below) of the plan the dry run builds (at most `cap` floats from the
// image's start to the end of the blob), and the launch's name.  Lets the CPU suite check host packing against a
// numpy model of a kernel's dataflow.  Returns the number of floats copied or a negative error code.
int64_t yfv2_debug_plan_image(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t step, char* name, int32_t name_cap,
                              float* dst, int64_t cap) {
  return yfv2_debug_plan_image_ex(cfg, nullptr, tensors, n, step, name, name_cap, dst, cap);
}

int64_t yfv2_debug_plan_image_ex(const yfv2_config* cfg, const yfv2_plan* plan, const yfv2_tensor_desc* tensors, int32_t n, int32_t step, char* name,
                                 int32_t name_cap, float* dst, int64_t cap) {
  if (!cfg || !tensors || n <= 0 || (step != -1 && (!dst || cap <= 0))) return fail(nullptr, YFV2_ERR_ARG, "yfv2_debug_plan_image: bad argument");
  int rows = 0;
  if (int rc = check_config(cfg, &rows)) return rc;
  yfv2_ctx ctx;
  uintptr_t next = 0x100000000ull;
  auto fake = [&](yfv2_ctx* hh, Buf* b, size_t per_img) {
    b->per_img = per_img;
    b->p = reinterpret_cast<float*>(next);
    next += (per_img * sizeof(float) * (size_t)hh->cfg.max_batch + 4095) & ~(uintptr_t)4095;
    return (int)YFV2_OK;
  };
  setup_ctx(&ctx, cfg, rows, fake);
  read_plan_switches(&ctx, plan);
  WeightPacker wp;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) wp.byname[tensors[i].name] = &tensors[i];
  PlanBuilder pb{&ctx, wp};
  pb.build();
  if (!pb.ok || !wp.missing.empty()) return fail(nullptr, YFV2_ERR_WEIGHTS, wp.missing.empty() ? "weight packing failed" : wp.missing);
  // step + 1000 (k + 1): job k of a launch that runs several tower halves (towers_kernel's list, towerh_kernel's side-by-side pair)
  const int job = step >= 1000 ? step / 1000 - 1 : -1;
  if (step >= 1000) step %= 1000;
  // the images are those of the launches as packed: under front_kernel (one launch for the stem and stage2.0) the stem's step is
  // put back in front and stage2.0 answers to its own name - step indices are those of the two-launch plan
  std::vector<Step> view = ctx.plan;
  if (ctx.front_fused) { view.insert(view.begin(), ctx.stem_aside); view[1].name = view[1].name_plain; }
  if (step == -1) return (int64_t)view.size();   // the number of steps of THIS index space (launch plan + 1 where the front is one launch)
  if (step < 0 || step >= (int32_t)view.size()) return fail(nullptr, YFV2_ERR_ARG, "yfv2_debug_plan_image: step out of range");
  if (job >= (int)view[step].jobs.size()) return fail(nullptr, YFV2_ERR_ARG, "yfv2_debug_plan_image: job out of range");
  const Step& st = job >= 0 ? view[step].jobs[job] : view[step];
  if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", st.name.c_str());
  const int64_t avail = (int64_t)wp.blob.size() - (int64_t)st.img_off;
  const int64_t cnt = avail < cap ? avail : cap;
  if (cnt > 0) std::memcpy(dst, &wp.blob[st.img_off], sizeof(float) * (size_t)cnt);
  return cnt;
}

// Host-only test hook: the channel order in which the plan stores stage 3's output (C2).  label[k] = logical channel at
// physical position k; returns 1 if the plan permutes (chain kernel), 0 if C2 is plain NHWC, or a negative error code.
int yfv2_debug_plan_c2_label(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t* label) {
  if (!cfg || !tensors || n <= 0 || !label) return fail(nullptr, YFV2_ERR_ARG, "yfv2_debug_plan_c2_label: bad argument");
  int rows = 0;
  if (int rc = check_config(cfg, &rows)) return rc;
  yfv2_ctx ctx;
  uintptr_t next = 0x100000000ull;
  auto fake = [&](yfv2_ctx* hh, Buf* b, size_t per_img) {
    b->per_img = per_img;
    b->p = reinterpret_cast<float*>(next);
    next += (per_img * sizeof(float) * (size_t)hh->cfg.max_batch + 4095) & ~(uintptr_t)4095;
    return (int)YFV2_OK;
  };
  setup_ctx(&ctx, cfg, rows, fake);
  read_plan_switches(&ctx, nullptr);
  WeightPacker wp;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) wp.byname[tensors[i].name] = &tensors[i];
  PlanBuilder pb{&ctx, wp};
  pb.build();
  if (!pb.ok || !wp.missing.empty()) return fail(nullptr, YFV2_ERR_WEIGHTS, wp.missing.empty() ? "weight packing failed" : wp.missing);
  for (int k = 0; k < 96; ++k) label[k] = ctx.c2_permuted ? ctx.c2_label[k] : k;
  return ctx.c2_permuted ? 1 : 0;
}

void yfv2_destroy(yfv2_handle h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  free_buf(&h->a1);
  for (int i = 0; i < 2; ++i) { free_buf(&h->s2[i]); free_buf(&h->s3[i]); free_buf(&h->s4[i]); }
  free_buf(&h->s2pp);
  free_buf(&h->t1); free_buf(&h->t2); free_buf(&h->t3);
  free_buf(&h->f2); free_buf(&h->f3); free_buf(&h->fq); free_buf(&h->ta); free_buf(&h->tb);
  for (int i = 0; i < 6; ++i) free_buf(&h->logits[i]);
  free_buf(&h->cand);
  if (h->d_classes) (void)hipFree(h->d_classes);
  if (h->h_nonfinite) (void)hipHostFree(h->h_nonfinite);
  if (h->d_probe) (void)hipFree(h->d_probe);
  if (h->d_loss_ws) (void)hipFree(h->d_loss_ws);
  if (h->train) { yfv2_train_release(h->train); h->train = nullptr; }
  if (h->d_params) (void)hipFree(h->d_params);
  for (yfv2_ctx* lane : h->lanes) yfv2_destroy(lane);
  for (hipStream_t st : h->lane_stream) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (hipEvent_t ev : h->lane_join) (void)hipEventDestroy(ev);
  if (h->lane_fork) (void)hipEventDestroy(h->lane_fork);
  delete h;
}

int yfv2_load_weights(yfv2_handle h, const yfv2_tensor_desc* tensors, int32_t n) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (!tensors || n <= 0) return fail(h, YFV2_ERR_ARG, "yfv2_load_weights: no tensors");
  DeviceGuard guard(h->device);
  WeightPacker wp;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) wp.byname[tensors[i].name] = &tensors[i];
  h->plan.clear();
  h->c2_permuted = false;
  PlanBuilder pb{h, wp};
  pb.build();
  if (!pb.ok || !wp.missing.empty()) {
    h->plan.clear();
    h->weights_loaded = false;
    return fail(h, YFV2_ERR_WEIGHTS, wp.missing.empty() ? "weight packing failed" : wp.missing);
  }
  HIP_TRY(h, hipDeviceSynchronize());  // nothing of ours may still read the old blob
  if (h->d_params && h->n_params < wp.blob.size()) { (void)hipFree(h->d_params); h->d_params = nullptr; }
  if (!h->d_params) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_params), wp.blob.size() * sizeof(float)));
    h->n_params = wp.blob.size();
  }
  HIP_TRY(h, hipMemcpy(h->d_params, wp.blob.data(), wp.blob.size() * sizeof(float), hipMemcpyHostToDevice));
  h->weights_loaded = true;
  for (yfv2_ctx* lane : h->lanes)
    if (int rc = yfv2_load_weights(lane, tensors, n)) { h->weights_loaded = false; return fail(h, rc, lane->err); }
  return YFV2_OK;
}

int yfv2_set_anchors(yfv2_handle h, const double anchors[12]) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (!anchors) return fail(h, YFV2_ERR_ARG, "yfv2_set_anchors: null pointer");
  for (int i = 0; i < 12; ++i) h->cfg.anchors[i] = anchors[i];
  for (yfv2_ctx* lane : h->lanes) (void)yfv2_set_anchors(lane, anchors);
  return YFV2_OK;
}

int yfv2_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (!x || !out6) return fail(h, YFV2_ERR_ARG, "yfv2_forward: null pointer");
  for (int i = 0; i < 6; ++i)
    if (!out6[i]) return fail(h, YFV2_ERR_ARG, "yfv2_forward: null output tensor");
  DeviceGuard guard(h->device);
  if (use_lanes(h, B))
    return run_lanes(h, B, static_cast<hipStream_t>(stream), [&](yfv2_ctx* lane, int off, int cnt, hipStream_t st) {
      float* o6[6];
      for (int i = 0; i < 6; ++i) o6[i] = out6[i] + (size_t)off * logit_elems(h, i);
      return yfv2_forward(lane, x + (size_t)off * 3 * h->cfg.height * h->cfg.width, cnt, o6, st);
    });
  h->last_split.clear();
  return run_plan(h, x, false, B, out6, static_cast<hipStream_t>(stream), nullptr);
}

int yfv2_forward_u8(yfv2_handle h, const uint8_t* x, int32_t B, float* const out6[6], void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (!x || !out6) return fail(h, YFV2_ERR_ARG, "yfv2_forward_u8: null pointer");
  for (int i = 0; i < 6; ++i)
    if (!out6[i]) return fail(h, YFV2_ERR_ARG, "yfv2_forward_u8: null output tensor");
  DeviceGuard guard(h->device);
  if (use_lanes(h, B))
    return run_lanes(h, B, static_cast<hipStream_t>(stream), [&](yfv2_ctx* lane, int off, int cnt, hipStream_t st) {
      float* o6[6];
      for (int i = 0; i < 6; ++i) o6[i] = out6[i] + (size_t)off * logit_elems(h, i);
      return yfv2_forward_u8(lane, x + (size_t)off * 3 * h->cfg.height * h->cfg.width, cnt, o6, st);
    });
  h->last_split.clear();
  return run_plan(h, x, true, B, out6, static_cast<hipStream_t>(stream), nullptr);
}

static DecodeArgs decode_args(yfv2_handle h, const float* const out6[6], int32_t B) {
  DecodeArgs a{};
  for (int sc = 0; sc < 2; ++sc) {
    a.reg[sc] = out6[sc * 3 + 0];
    a.obj[sc] = out6[sc * 3 + 1];
    a.cls[sc] = out6[sc * 3 + 2];
    a.fh[sc] = h->fh[sc];
    a.fw[sc] = h->fw[sc];
    // utils.py:332  stride = cfg["height"] / r.shape[0]  (python float -> fp32 scalar multiply)
    a.stride[sc] = (float)((double)h->cfg.height / (double)h->fh[sc]);
  }
  for (int i = 0; i < 12; ++i) a.anchors[i] = h->cfg.anchors[i];
  a.B = B;
  a.classes = h->cfg.classes;
  a.rows = h->rows;
  return a;
}
// handel_preds + non_max_suppression behind a forward whose logits sit in h->logits: one launch that decodes each image into
// LDS (default), or decode_kernel<compact> + nms_kernel over candidate rows in HBM (YFV2_POSTFUSE=0, the A/B reference)
static int post_impl(yfv2_handle h, int32_t B, float conf_thres, double iou_thres, float* dets, int32_t* idx, int32_t* count, void* stream);
static int decode_impl(yfv2_handle h, const float* const out6[6], int32_t B, float* boxes, float* cand, void* stream);
static int nms_impl(yfv2_handle h, const float* boxes, int compact, int32_t B, float conf_thres, double iou_thres,
                    const int32_t* classes, int32_t n_classes, float* dets, int32_t* idx, int32_t* count, void* stream);

int yfv2_decode(yfv2_handle h, const float* const out6[6], int32_t B, float* boxes, void* stream) {
  return decode_impl(h, out6, B, boxes, nullptr, stream);
}

static int decode_impl(yfv2_handle h, const float* const out6[6], int32_t B, float* boxes, float* cand, void* stream) {
  int rc = check_call(h, B, false);
  if (rc) return rc;
  if (!out6 || (!boxes && !cand)) return fail(h, YFV2_ERR_ARG, "yfv2_decode: null pointer");
  for (int i = 0; i < 6; ++i)
    if (!out6[i]) return fail(h, YFV2_ERR_ARG, "yfv2_decode: null logit tensor");
  DeviceGuard guard(h->device);
  DecodeArgs a = decode_args(h, out6, B);
  a.boxes = boxes;
  a.cand = cand;
  yfv2_launch_decode(a, static_cast<hipStream_t>(stream));
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int yfv2_nms(yfv2_handle h, const float* boxes, int32_t B, float conf_thres, double iou_thres, const int32_t* classes,
             int32_t n_classes, float* dets, int32_t* idx, int32_t* count, void* stream) {
  return nms_impl(h, boxes, 0, B, conf_thres, iou_thres, classes, n_classes, dets, idx, count, stream);
}

static int nms_impl(yfv2_handle h, const float* boxes, int compact, int32_t B, float conf_thres, double iou_thres,
                    const int32_t* classes, int32_t n_classes, float* dets, int32_t* idx, int32_t* count, void* stream) {
  int rc = check_call(h, B, false);
  if (rc) return rc;
  if (!boxes || !dets || !idx || !count) return fail(h, YFV2_ERR_ARG, "yfv2_nms: null pointer");
  if (n_classes < 0 || n_classes > 256 || (n_classes > 0 && !classes))
    return fail(h, YFV2_ERR_ARG, "yfv2_nms: bad class filter");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  NmsArgs a{};
  a.boxes = boxes; a.compact = compact; a.dets = dets; a.idx = idx; a.count = count;
  a.classes = nullptr; a.n_classes = 0;
  if (n_classes > 0) {  // classes is a HOST array (python list in the reference, utils.py:271-272)
    HIP_TRY(h, hipMemcpyAsync(h->d_classes, classes, sizeof(int32_t) * n_classes, hipMemcpyHostToDevice, s));
    a.classes = h->d_classes; a.n_classes = n_classes;
  }
  a.B = B; a.rows = h->rows; a.nc = h->cfg.classes;
  a.conf_thres = conf_thres; a.iou_thres = iou_thres;
  yfv2_launch_nms(a, s);
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int yfv2_detect(yfv2_handle h, const float* x, int32_t B, float conf_thres, double iou_thres, float* dets, int32_t* idx,
                int32_t* count, void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (use_lanes(h, B)) {
    if (!x || !dets || !idx || !count) return fail(h, YFV2_ERR_ARG, "yfv2_detect: null pointer");
    DeviceGuard guard(h->device);
    return run_lanes(h, B, static_cast<hipStream_t>(stream), [&](yfv2_ctx* lane, int off, int cnt, hipStream_t st) {
      return yfv2_detect(lane, x + (size_t)off * 3 * h->cfg.height * h->cfg.width, cnt, conf_thres, iou_thres, dets + (size_t)off * YFV2_MAX_DET * 6,
                         idx + (size_t)off * YFV2_MAX_DET, count + off, st);
    });
  }
  float* out6[6];
  for (int i = 0; i < 6; ++i) out6[i] = h->logits[i].p;
  rc = yfv2_forward(h, x, B, out6, stream);
  if (rc) return rc;
  return post_impl(h, B, conf_thres, iou_thres, dets, idx, count, stream);
}

static int post_impl(yfv2_handle h, int32_t B, float conf_thres, double iou_thres, float* dets, int32_t* idx, int32_t* count, void* stream) {
  if (!dets || !idx || !count) return fail(h, YFV2_ERR_ARG, "yfv2_detect: null pointer");
  float* out6[6];
  for (int i = 0; i < 6; ++i) out6[i] = h->logits[i].p;
  if (!h->postfuse || !yfv2_post_fusable(h->cfg.classes, h->rows)) {
    // compact candidate rows instead of the (B,1815,85) tensor: same arithmetic, 10x less traffic
    const int rc = decode_impl(h, out6, B, nullptr, h->cand.p, stream);
    if (rc) return rc;
    return nms_impl(h, h->cand.p, 1, B, conf_thres, iou_thres, nullptr, 0, dets, idx, count, stream);
  }
  DeviceGuard guard(h->device);
  const DecodeArgs d = decode_args(h, out6, B);
  NmsArgs a{};
  a.boxes = nullptr; a.compact = 1; a.dets = dets; a.idx = idx; a.count = count;
  a.classes = nullptr; a.n_classes = 0;
  a.B = B; a.rows = h->rows; a.nc = h->cfg.classes;
  a.conf_thres = conf_thres; a.iou_thres = iou_thres;
  a.trace = h->trace_step == -2 ? h->d_trace : nullptr;   // YFV2_TRACE=1 YFV2_TRACE_STEP=-2: stamps of the post launch (tools/trace_post.py)
  yfv2_launch_decode_nms(d, a, static_cast<hipStream_t>(stream));
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int yfv2_detect_u8(yfv2_handle h, const uint8_t* x, int32_t B, float conf_thres, double iou_thres, float* dets, int32_t* idx,
                   int32_t* count, void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (use_lanes(h, B)) {
    if (!x || !dets || !idx || !count) return fail(h, YFV2_ERR_ARG, "yfv2_detect_u8: null pointer");
    DeviceGuard guard(h->device);
    return run_lanes(h, B, static_cast<hipStream_t>(stream), [&](yfv2_ctx* lane, int off, int cnt, hipStream_t st) {
      return yfv2_detect_u8(lane, x + (size_t)off * 3 * h->cfg.height * h->cfg.width, cnt, conf_thres, iou_thres, dets + (size_t)off * YFV2_MAX_DET * 6,
                            idx + (size_t)off * YFV2_MAX_DET, count + off, st);
    });
  }
  float* out6[6];
  for (int i = 0; i < 6; ++i) out6[i] = h->logits[i].p;
  rc = yfv2_forward_u8(h, x, B, out6, stream);
  if (rc) return rc;
  return post_impl(h, B, conf_thres, iou_thres, dets, idx, count, stream);
}

// enqueue only: the overflow flag is sticky in the handle until yfv2_batch_statistics_overflow reads it
int yfv2_batch_statistics_async(yfv2_handle h, const float* dets, const int32_t* count, int32_t B, const float* targets, int32_t T,
                                float iou_threshold, int32_t* tp, void* stream) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (B < 1) return fail(h, YFV2_ERR_BATCH, "yfv2_batch_statistics: B < 1");   // no workspace involved: B is not bound by max_batch
  if (!dets || !count || !tp || T < 0 || (T > 0 && !targets)) return fail(h, YFV2_ERR_ARG, "yfv2_batch_statistics: bad argument");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  StatsArgs a{};
  a.dets = dets; a.count = count; a.targets = targets; a.tp = tp; a.overflow = h->d_stats_flag;
  a.B = B; a.T = T; a.iou_thres = iou_threshold;
  yfv2_launch_stats(a, s);
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int yfv2_batch_statistics_overflow(yfv2_handle h, int32_t* overflowed, void* stream) {
  if (!h || !overflowed) return fail(h, YFV2_ERR_ARG, "yfv2_batch_statistics_overflow: null pointer");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int32_t over = 0;
  HIP_TRY(h, hipMemcpyAsync(&over, h->d_stats_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemsetAsync(h->d_stats_flag, 0, sizeof(int32_t), s));
  HIP_TRY(h, hipStreamSynchronize(s));
  *overflowed = over;
  return YFV2_OK;
}

// the sticky range-guard word of the fp16x3 plan (yfv2_internal.h Yfv2Watch): waits for `stream`, reports and clears it
int yfv2_nonfinite(yfv2_handle h, int32_t* flag, void* stream) {
  if (!h || !flag) return fail(h, YFV2_ERR_ARG, "yfv2_nonfinite: null argument");
  if (!h->h_nonfinite) return fail(h, YFV2_ERR_STATE, "yfv2_nonfinite: this handle owns no guard word");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(h, hipStreamSynchronize(s));        // a kernel's stores to coherent host memory are visible once it has completed
  volatile int32_t* w = h->h_nonfinite;
  const int32_t v = *w;
  if (v) *w = 0;
  *flag = v ? 1 : 0;
  return YFV2_OK;
}

// ... and without waiting for anything: what has landed in the word so far (not cleared).  A caller that never synchronises
// with the host (a detect loop that hands device tensors on) looks before every call and learns of a tripped guard one call late
// instead of never.
int yfv2_nonfinite_peek(yfv2_handle h, int32_t* flag) {
  if (!h || !flag) return fail(h, YFV2_ERR_ARG, "yfv2_nonfinite_peek: null argument");
  if (!h->h_nonfinite) return fail(h, YFV2_ERR_STATE, "yfv2_nonfinite_peek: this handle owns no guard word");
  *flag = *reinterpret_cast<volatile int32_t*>(h->h_nonfinite) ? 1 : 0;
  return YFV2_OK;
}

// effective shader clock, measured by the shader (yfv2_probe.hip): enqueue on `stream` ...
int yfv2_clock_probe_begin(yfv2_handle h, int32_t workgroups, float milliseconds, int32_t busy, void* stream) {
  if (!h || workgroups < 1 || workgroups > 4096 || !(milliseconds > 0.f) || milliseconds > 10000.f)
    return fail(h, YFV2_ERR_ARG, "yfv2_clock_probe_begin: bad argument");
  DeviceGuard guard(h->device);
  if (!h->d_probe) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_probe), 4096 * 4 * sizeof(unsigned long long)));
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
  ClockProbeArgs a{};
  a.out = h->d_probe; a.busy = busy ? 1 : 0;
  a.ref_ticks = (unsigned long long)((double)milliseconds * (double)khz);
  h->probe_wgs = workgroups;
  if (!yfv2_launch_clock_probe(a, workgroups, static_cast<hipStream_t>(stream))) return fail(h, YFV2_ERR_DEVICE, "clock probe launch failed");
  return YFV2_OK;
}

// ... and read it back (waits for `stream`): out[0..2] = min / mean / max over the probe's workgroups of
// (shader cycles / reference ticks) x reference clock, in MHz; out[3] = the reference clock in MHz; out[4] = mean measured
// interval in milliseconds; out[5] = number of distinct XCDs the workgroups ran on
int yfv2_clock_probe_end(yfv2_handle h, double out[6], void* stream) {
  if (!h || !out) return fail(h, YFV2_ERR_ARG, "yfv2_clock_probe_end: null argument");
  if (!h->d_probe || h->probe_wgs < 1) return fail(h, YFV2_ERR_STATE, "yfv2_clock_probe_end without yfv2_clock_probe_begin");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<unsigned long long> st((size_t)h->probe_wgs * 4);
  HIP_TRY(h, hipMemcpyAsync(st.data(), h->d_probe, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
  const double ref_mhz = khz * 1e-3;
  double mn = 1e30, mx = 0., sum = 0., ms = 0.;
  unsigned xcds = 0;
  for (int i = 0; i < h->probe_wgs; ++i) {
    const double cyc = (double)st[4 * i], ref = (double)st[4 * i + 1];
    if (!(ref > 0.)) return fail(h, YFV2_ERR_DEVICE, "clock probe: a workgroup reported no reference ticks");
    const double mhz = cyc / ref * ref_mhz;
    mn = std::min(mn, mhz); mx = std::max(mx, mhz); sum += mhz; ms += ref / ref_mhz * 1e-3;
    xcds |= 1u << (unsigned)(st[4 * i + 2] & 15);
  }
  out[0] = mn; out[1] = sum / h->probe_wgs; out[2] = mx; out[3] = ref_mhz; out[4] = ms / h->probe_wgs; out[5] = (double)__builtin_popcount(xcds);
  h->probe_wgs = 0;
  return YFV2_OK;
}

int yfv2_batch_statistics(yfv2_handle h, const float* dets, const int32_t* count, int32_t B, const float* targets, int32_t T,
                          float iou_threshold, int32_t* tp, void* stream) {
  int rc = yfv2_batch_statistics_async(h, dets, count, B, targets, T, iou_threshold, tp, stream);
  if (rc) return rc;
  int32_t over = 0;
  rc = yfv2_batch_statistics_overflow(h, &over, stream);
  if (rc) return rc;
  if (over) return fail(h, YFV2_ERR_ARG, "yfv2_batch_statistics: an image has more than 1024 targets");
  return YFV2_OK;
}

int yfv2_loss(yfv2_handle h, const float* const out6[6], int32_t B, const float* targets, int32_t T, float* losses,
              float* const grad6[6], void* stream) {
  int rc = check_call(h, B, false);
  if (rc) return rc;
  if (!out6 || !losses || T < 0 || (T > 0 && !targets)) return fail(h, YFV2_ERR_ARG, "yfv2_loss: bad argument");
  for (int i = 0; i < 6; ++i)
    if (!out6[i] || (grad6 && !grad6[i])) return fail(h, YFV2_ERR_ARG, "yfv2_loss: null logit / gradient tensor");
  if (T > (1 << 20)) return fail(h, YFV2_ERR_ARG, "yfv2_loss: more than 2^20 labels in one batch");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t A = (size_t)h->cfg.anchor_num;
  const size_t cells0 = (size_t)B * A * h->fh[0] * h->fw[0], cells1 = (size_t)B * A * h->fh[1] * h->fw[1];
  // layout: [sums 6 doubles][nb 2 ints + pad][tobj0][tobj1][pad][matches]
  const size_t off_nb = 6 * sizeof(double), off_t0 = off_nb + 16, off_t1 = off_t0 + cells0;
  const size_t zero_bytes = (off_t1 + cells1 + 15) & ~(size_t)15;
  const size_t need = zero_bytes + sizeof(LossMatch) * (size_t)(2 * 5 * 3) * (size_t)(T > 0 ? T : 1);
  if (need > h->loss_ws_bytes) {
    HIP_TRY(h, hipDeviceSynchronize());               // an earlier yfv2_loss may still be using the old block
    if (h->d_loss_ws) { (void)hipFree(h->d_loss_ws); h->d_loss_ws = nullptr; h->loss_ws_bytes = 0; }
    const size_t cap = need + need / 2;
    HIP_TRY(h, hipMalloc(&h->d_loss_ws, cap));
    h->loss_ws_bytes = cap;
  }
  char* ws = static_cast<char*>(h->d_loss_ws);
  HIP_TRY(h, hipMemsetAsync(ws, 0, zero_bytes, s));
  LossArgs a{};
  for (int l = 0; l < 2; ++l) {
    a.reg[l] = out6[3 * l]; a.obj[l] = out6[3 * l + 1]; a.cls[l] = out6[3 * l + 2];
    a.grad_reg[l] = grad6 ? grad6[3 * l] : nullptr; a.grad_obj[l] = grad6 ? grad6[3 * l + 1] : nullptr; a.grad_cls[l] = grad6 ? grad6[3 * l + 2] : nullptr;
    a.fh[l] = h->fh[l]; a.fw[l] = h->fw[l];
    a.stride[l] = (double)h->cfg.width / (double)h->fw[l];        // utils/loss.py:82
    if (grad6) {                                                   // reg / cls gradients are accumulated with atomics: start from zero
      HIP_TRY(h, hipMemsetAsync(grad6[3 * l], 0, sizeof(float) * (size_t)B * 4 * A * h->fh[l] * h->fw[l], s));
      HIP_TRY(h, hipMemsetAsync(grad6[3 * l + 2], 0, sizeof(float) * (size_t)B * h->cfg.classes * h->fh[l] * h->fw[l], s));
    }
  }
  for (int i = 0; i < 12; ++i) a.anchors[i] = h->cfg.anchors[i];
  a.targets = targets;
  a.sums = reinterpret_cast<double*>(ws);
  a.nb = reinterpret_cast<int*>(ws + off_nb);
  a.tobj[0] = reinterpret_cast<unsigned char*>(ws + off_t0);
  a.tobj[1] = reinterpret_cast<unsigned char*>(ws + off_t1);
  a.matches = reinterpret_cast<LossMatch*>(ws + zero_bytes);
  a.losses = losses;
  a.B = B; a.T = T; a.classes = h->cfg.classes;
  yfv2_launch_loss(a, s);
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int yfv2_resize_u8(yfv2_handle h, const uint8_t* src, int32_t B, int32_t src_h, int32_t src_w, uint8_t* dst, void* stream) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (!src || !dst || B < 1 || src_h < 1 || src_w < 1) return fail(h, YFV2_ERR_ARG, "yfv2_resize_u8: bad argument");
  if ((reinterpret_cast<uintptr_t>(dst) & 3) != 0) return fail(h, YFV2_ERR_ARG, "yfv2_resize_u8: dst must be 4-byte aligned");
  if (yfv2_resize_lds_bytes(src_w, h->cfg.width) > 160 * 1024 || (long long)B * h->cfg.height > 0x7fffffffll)
    return fail(h, YFV2_ERR_ARG, "yfv2_resize_u8: source rows wider than " + std::to_string((160 * 1024 - 3 * h->cfg.width) / 6 - 2) + " pixels are not supported");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ResizeArgs a{};
  a.src = src; a.dst = dst; a.B = B; a.SH = src_h; a.SW = src_w; a.H = h->cfg.height; a.W = h->cfg.width;
  a.scale_x = 1.0 / ((double)a.W / (double)src_w);      // cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale
  a.scale_y = 1.0 / ((double)a.H / (double)src_h);
  yfv2_launch_resize(a, s);
  HIP_TRY(h, hipGetLastError());
  return YFV2_OK;
}

int32_t yfv2_num_rows(yfv2_handle h) { return h ? h->rows : 0; }

int32_t yfv2_num_stages(yfv2_handle h) { return h ? (int32_t)h->plan.size() : 0; }

int yfv2_stage_info(yfv2_handle h, int32_t i, char* name, int32_t name_cap, double* flops_per_image, double* bytes_per_image,
                    double* external_bytes_per_image) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (i < 0 || i >= (int32_t)h->plan.size()) return fail(h, YFV2_ERR_ARG, "stage index out of range");
  const Step& s = h->plan[i];
  if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", s.name.c_str());
  if (flops_per_image) *flops_per_image = s.flops;
  if (bytes_per_image) *bytes_per_image = s.bytes;
  if (external_bytes_per_image) *external_bytes_per_image = s.bytes_ext >= 0 ? s.bytes_ext : s.bytes;
  return YFV2_OK;
}

int yfv2_stage_kernel(yfv2_handle h, int32_t i, char* name, int32_t name_cap) {
  if (!h) return fail(nullptr, YFV2_ERR_ARG, "null handle");
  if (i < 0 || i >= (int32_t)h->plan.size() || !name || name_cap < 1) return fail(h, YFV2_ERR_ARG, "yfv2_stage_kernel: bad argument");
  std::snprintf(name, (size_t)name_cap, "%s", step_kernel(h->plan[i]).c_str());
  return YFV2_OK;
}

int yfv2_profile_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], int32_t iters, float* ms, void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (!x || !out6 || !ms || iters < 1) return fail(h, YFV2_ERR_ARG, "yfv2_profile_forward: bad argument");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // One untimed pass, then `iters` timed passes queued back to back and ONE synchronisation at the end: a pass's first launch
  // follows the previous pass's last one, as in a running loop.  (Synchronising after every pass - the first form - put the stem
  // behind an idle device each time: 127 us by these events against 114 us in a rocprofv3 trace of the bench loop on the same box.)
  const size_t n = h->plan.size();
  // events and the post launch's output buffers are released on EVERY path out of this function (HIP_TRY returns early)
  struct Scratch {
    std::vector<hipEvent_t> ev;
    float* dets = nullptr; int32_t* idx = nullptr; int32_t* cnt = nullptr;
    ~Scratch() {
      for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
      if (dets) (void)hipFree(dets);
      if (idx) (void)hipFree(idx);
      if (cnt) (void)hipFree(cnt);
    }
  } sc;
  sc.ev.assign(2 * n * (size_t)iters, nullptr);
  for (auto& e : sc.ev) HIP_TRY(h, hipEventCreate(&e));
  std::vector<hipEvent_t>& ev = sc.ev;
  // Between two passes the post launch runs (untimed, on the logits just written, test.py's thresholds 0.3 / 0.4), as it does
  // between two forwards of a detect loop: a pass's first launch then meets the memory system in the state it meets there (behind
  // the last tower launch's 47 MB of logit stores instead, the stem took 121 us by these events against 109 us in the trace).
  const bool with_post = h->postfuse && yfv2_post_fusable(h->cfg.classes, h->rows);
  if (with_post) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&sc.dets), (size_t)B * YFV2_MAX_DET * 6 * sizeof(float)));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&sc.idx), (size_t)B * YFV2_MAX_DET * sizeof(int32_t)));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&sc.cnt), (size_t)B * sizeof(int32_t)));
  }
  float* const p_dets = sc.dets; int32_t* const p_idx = sc.idx; int32_t* const p_cnt = sc.cnt;
  auto post = [&]() {
    if (!with_post) return;
    const DecodeArgs d = decode_args(h, out6, B);
    NmsArgs a{};
    a.boxes = nullptr; a.compact = 1; a.dets = p_dets; a.idx = p_idx; a.count = p_cnt;
    a.classes = nullptr; a.n_classes = 0;
    a.B = B; a.rows = h->rows; a.nc = h->cfg.classes;
    a.conf_thres = 0.3f; a.iou_thres = 0.4;
    a.trace = nullptr;
    yfv2_launch_decode_nms(d, a, s);
  };
  rc = run_plan(h, x, false, B, out6, s, nullptr);
  post();
  for (int it = 0; it < iters && rc == YFV2_OK; ++it) { rc = run_plan(h, x, false, B, out6, s, ev.data() + 2 * n * (size_t)it); post(); }
  if (rc == YFV2_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(h, YFV2_ERR_DEVICE, "yfv2_profile_forward: synchronize failed");
  if (rc != YFV2_OK) (void)hipStreamSynchronize(s);   // nothing may still be writing the post buffers when Scratch frees them
  std::vector<double> acc(n, 0.0);
  for (int it = 0; it < iters && rc == YFV2_OK; ++it)
    for (size_t i = 0; i < n; ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev[2 * n * (size_t)it + 2 * i], ev[2 * n * (size_t)it + 2 * i + 1]) != hipSuccess) { rc = fail(h, YFV2_ERR_DEVICE, "yfv2_profile_forward: event query failed"); break; }
      acc[i] += t;
    }
  if (rc) return rc;
  for (size_t i = 0; i < n; ++i) ms[i] = (float)(acc[i] / iters);
  return YFV2_OK;
}

// Measurement helper: one whole forward (so that every launch's inputs exist), then launch `step` of the plan `iters` times back to
// back on `stream` (every launch reads its inputs and writes its outputs in place again: idempotent).  Enqueue only - the caller
// times it, or reads the device's power sensor while it runs (tools/power_probe.py).
int yfv2_debug_repeat_step(yfv2_handle h, const float* x, int32_t B, float* const out6[6], int32_t step, int32_t iters, void* stream) {
  int rc = check_call(h, B, true);
  if (rc) return rc;
  if (!x || !out6 || iters < 0 || step < 0 || step >= (int32_t)h->plan.size()) return fail(h, YFV2_ERR_ARG, "yfv2_debug_repeat_step: bad argument");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = run_plan(h, x, false, B, out6, s, nullptr);
  for (int it = 0; it < iters && rc == YFV2_OK; ++it) rc = run_plan(h, x, false, B, out6, s, nullptr, step);
  return rc;
}

int64_t yfv2_debug_activation(yfv2_handle h, int32_t which, int32_t B, float* host_dst, int64_t cap) {
  if (h && which == 100 && h->d_trace && host_dst && cap >= 128) {  // debug: cycle stamps as int64 (2 floats each), as many as fit (<= 8192)
    (void)hipDeviceSynchronize();
    const int64_t n64 = cap / 2 < 8192 ? cap / 2 : 8192;
    (void)hipMemcpy(host_dst, h->d_trace, (size_t)n64 * sizeof(long long), hipMemcpyDeviceToHost);
    return n64;
  }
  if (h && which == 101 && h->s2_px && host_dst) {  // debug: both raw stage-2 pair-plane buffers, B images each
    const size_t per = h->dbg_per_img[1], nn = (size_t)B * per;     // -> [buffer][image][..]; on the device an image's two copies are adjacent
    if (cap < (int64_t)(2 * nn)) return YFV2_ERR_ARG;
    (void)hipDeviceSynchronize();
    for (int k = 0; k < 2; ++k)
      (void)hipMemcpy2D(host_dst + (size_t)k * nn, per * sizeof(float), h->s2pp.p + (size_t)k * per, 2 * per * sizeof(float), per * sizeof(float), (size_t)B,
                        hipMemcpyDeviceToHost);
    return (int64_t)(2 * nn);
  }
  if (!h || which < 0 || which > 5 || !h->dbg[which] || B < 1 || B > h->cfg.max_batch) {
    fail(h, YFV2_ERR_ARG, "yfv2_debug_activation: bad argument");
    return YFV2_ERR_ARG;
  }
  DeviceGuard guard(h->device);
  const int64_t n = (int64_t)h->dbg_per_img[which] * B;
  if (!host_dst) return n;
  if (!h->last_split.empty()) {   // the last forward ran on the lanes: every lane holds its slice
    if (cap < n) { fail(h, YFV2_ERR_ARG, "yfv2_debug_activation: destination too small"); return YFV2_ERR_ARG; }
    int off = 0;
    for (size_t i = 0; i < h->last_split.size() && off < B; ++i) {
      const int cnt = std::min(h->last_split[i], B - off);
      const int64_t got = yfv2_debug_activation(h->lanes[i], which, cnt, host_dst + (size_t)off * h->dbg_per_img[which], (int64_t)h->dbg_per_img[which] * cnt);
      if (got < 0) return got;
      off += cnt;
    }
    return n;
  }
  if (cap < n) { fail(h, YFV2_ERR_ARG, "yfv2_debug_activation: destination too small"); return YFV2_ERR_ARG; }
  if (which == 0 && h->front_fused) {
    // front_kernel never writes the stem's output: run the stem's own launch on the last forward's input (which the caller must still hold)
    if (!h->last_x || h->last_B < B) { fail(h, YFV2_ERR_STATE, "yfv2_debug_activation(0): no forward of at least this batch has run on the handle"); return YFV2_ERR_STATE; }
    StemArgs a = h->stem_aside.stem;
    a.x = h->last_x; a.B = B; a.u8_in = h->last_u8 ? 1 : 0;
    a.img = h->d_params + h->stem_aside.img_off; a.img_u8 = h->d_params + h->stem_aside.img_off2;
    a.img16 = h->bf6 ? h->d_params + h->stem_aside.img_off3 : nullptr;
    a.nonfinite = h->d_nonfinite;
    if (hipDeviceSynchronize() != hipSuccess) { fail(h, YFV2_ERR_DEVICE, "yfv2_debug_activation: synchronize failed"); return YFV2_ERR_DEVICE; }
    yfv2_launch_stem(a, nullptr);
  }
  if (which == 0 && h->stem_pp) {  // stem output in pair planes [12][PH*PW][2] (stem_px_kernel, YFV2_BF6=0) -> NHWC
    const size_t per = h->dbg_per_img[0], hw = per / 24;
    std::vector<float> tmp((size_t)n);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(tmp.data(), h->dbg[0], (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
      fail(h, YFV2_ERR_DEVICE, "yfv2_debug_activation: copy failed");
      return YFV2_ERR_DEVICE;
    }
    for (int b = 0; b < B; ++b)
      for (int q = 0; q < 12; ++q)
        for (size_t px = 0; px < hw; ++px)
          for (int e = 0; e < 2; ++e) host_dst[((size_t)b * hw + px) * 24 + 2 * q + e] = tmp[(size_t)b * per + ((size_t)q * hw + px) * 2 + e];
    return n;
  }
  if (which == 1 && h->s2_px) {  // stage 2 lives in pair planes: gather the logical NHWC tensor on the host
    const size_t per = h->dbg_per_img[1], hw = per / 48;
    std::vector<float> tmp(2 * (size_t)n);     // [buffer][image][pair][pixel][2]; on the device an image's two copies are adjacent
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy2D(tmp.data(), per * sizeof(float), h->s2pp.p, 2 * per * sizeof(float), per * sizeof(float), (size_t)B, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy2D(tmp.data() + n, per * sizeof(float), h->s2pp.p + per, 2 * per * sizeof(float), per * sizeof(float), (size_t)B, hipMemcpyDeviceToHost) != hipSuccess) {
      fail(h, YFV2_ERR_DEVICE, "yfv2_debug_activation: copy failed");
      return YFV2_ERR_DEVICE;
    }
    for (int b = 0; b < B; ++b)
      for (int q = 0; q < 24; ++q) {
        const float* src = tmp.data() + (size_t)h->s2_buf[q] * n + (size_t)b * per + (size_t)q * hw * 2;
        for (size_t px = 0; px < hw; ++px)
          for (int e = 0; e < 2; ++e) host_dst[((size_t)b * hw + px) * 48 + h->s2_label[2 * q + e]] = src[px * 2 + e];
      }
    return n;
  }
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(host_dst, h->dbg[which], (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
    fail(h, YFV2_ERR_DEVICE, "yfv2_debug_activation: copy failed");
    return YFV2_ERR_DEVICE;
  }
  if (which == 2 && h->c2_permuted) {   // stage 3 lives in the chain kernel's channel order: back to logical NHWC
    float tmp[96];
    for (int64_t px = 0; px < n / 96; ++px) {
      float* row = host_dst + px * 96;
      for (int k = 0; k < 96; ++k) tmp[h->c2_label[k]] = row[k];
      std::memcpy(row, tmp, sizeof(tmp));
    }
  }
  return n;
}

}  // extern "C"
