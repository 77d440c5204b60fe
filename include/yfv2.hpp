/* yfv2.hpp - C++ host class over the C ABI of libyfv2.so (include/yfv2.h): no Python, no torch.
 *
 * Counterpart of the reference's only native component, the ncnn sample class
 * `yoloFastestv2` (sample/ncnn/src/yolo-fastestv2.cpp:185-221: loadModel(param, bin) /
 * detection(srcImg, dstBoxes, thresh)), SURVEY.md section 8(f) row 4: same call shape, same TargetBox
 * record, but the arithmetic is the Python path's (utils/utils.py handel_preds + non_max_suppression at
 * conf/IoU thresholds 0.3/0.4, test.py:48-49) because that is what libyfv2 implements and pins - the ncnn
 * sample's own integer NMS (IoU 0.25, explicit class compare) is a different algorithm.
 *
 * detection() = one upload of the source frame as it is (uint8 HWC) -> yfv2_resize_u8 (cv2.resize INTER_LINEAR's 8-bit
 * arithmetic on the device, skipped when the frame already has the network size) -> yfv2_detect_u8 -> one download of
 * the padded rows -> boxes scaled back to the source image.
 *
 * Weights come from a flat container written by `yolo_fastestv2_amd.export_weights(state_dict, path)`:
 *   "YFV2W1\0\0" | int32 n | n x { int32 name_len | name bytes | int64 numel | numel x float32 }.
 */
#ifndef YFV2_HPP
#define YFV2_HPP

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "yfv2.h"

namespace yfv2 {

struct TargetBox {   // sample/ncnn/src/yolo-fastestv2.h TargetBox
  int x1, y1, x2, y2;
  int cate;
  float score;
};

class Detector {
 public:
  Detector(int classes, const double (&anchors)[12], int width = 352, int height = 352, int device = 0, const yfv2_plan* plan = nullptr)
      : width_(width), height_(height) {
    yfv2_config cfg{};
    cfg.classes = classes; cfg.anchor_num = 3; cfg.height = height; cfg.width = width;
    for (int i = 0; i < 12; ++i) cfg.anchors[i] = anchors[i];
    cfg.max_batch = 1; cfg.device = device;
    rc_ = yfv2_create_ex(&h_, &cfg, plan);   // plan: NULL = the default (fp16x3) plan; {.fp32_matrix = 1} = fp32 matrix instructions, no range limit
    if (rc_ != YFV2_OK) { err_ = yfv2_last_error(nullptr); return; }
    const size_t img = (size_t)width * height * 3;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d_img_, img) != hipSuccess || hipMalloc(&d_dets_, 300 * 6 * sizeof(float)) != hipSuccess ||
        hipMalloc(&d_idx_, 300 * sizeof(int32_t)) != hipSuccess || hipMalloc(&d_cnt_, sizeof(int32_t)) != hipSuccess ||
        hipStreamCreate(&stream_) != hipSuccess) {
      rc_ = YFV2_ERR_DEVICE; err_ = "hipMalloc / hipStreamCreate failed";
    }
  }
  ~Detector() {
    if (d_img_) (void)hipFree(d_img_);
    if (d_src_) (void)hipFree(d_src_);
    if (d_dets_) (void)hipFree(d_dets_);
    if (d_idx_) (void)hipFree(d_idx_);
    if (d_cnt_) (void)hipFree(d_cnt_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (h_) yfv2_destroy(h_);
  }
  Detector(const Detector&) = delete;
  Detector& operator=(const Detector&) = delete;

  bool ok() const { return rc_ == YFV2_OK; }
  const char* lastError() const { return err_.c_str(); }

  /* yoloFastestv2::loadModel: here ONE file, the flat tensor container described above */
  int loadModel(const char* path) {
    if (!ok()) return rc_;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(YFV2_ERR_ARG, std::string("cannot open ") + path);
    char magic[8];
    int32_t n = 0;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "YFV2W1\0\0", 8) != 0 || std::fread(&n, 4, 1, f) != 1 || n <= 0 || n > 4096) {
      std::fclose(f);
      return fail(YFV2_ERR_ARG, "not a YFV2W1 weight container");
    }
    std::vector<std::string> names(n);
    std::vector<std::vector<float>> data(n);
    for (int i = 0; i < n; ++i) {
      int32_t len = 0; int64_t numel = 0;
      if (std::fread(&len, 4, 1, f) != 1 || len <= 0 || len > 512) { std::fclose(f); return fail(YFV2_ERR_ARG, "corrupt weight container"); }
      names[i].resize(len);
      if (std::fread(&names[i][0], 1, len, f) != (size_t)len || std::fread(&numel, 8, 1, f) != 1 || numel < 0 || numel > (1 << 26)) {
        std::fclose(f); return fail(YFV2_ERR_ARG, "corrupt weight container");
      }
      data[i].resize((size_t)numel);
      if (numel && std::fread(data[i].data(), 4, (size_t)numel, f) != (size_t)numel) { std::fclose(f); return fail(YFV2_ERR_ARG, "truncated weight container"); }
    }
    std::fclose(f);
    std::vector<yfv2_tensor_desc> descs(n);
    for (int i = 0; i < n; ++i) { descs[i].name = names[i].c_str(); descs[i].data = data[i].data(); descs[i].numel = (int64_t)data[i].size(); }
    const int rc = yfv2_load_weights(h_, descs.data(), n);
    if (rc != YFV2_OK) return fail(rc, yfv2_last_error(h_));
    return YFV2_OK;
  }

  /* yoloFastestv2::detection(srcImg, dstBoxes, thresh): bgr = rows x cols x 3 uint8 (cv::Mat BGR data) */
  int detection(const unsigned char* bgr, int cols, int rows, std::vector<TargetBox>& dst, float thresh = 0.3f, float iou_thresh = 0.4f) {
    dst.clear();
    if (!ok()) return rc_;
    if (!bgr || cols <= 0 || rows <= 0) return fail(YFV2_ERR_ARG, "detection: bad image");
    const float scaleW = (float)cols / (float)width_, scaleH = (float)rows / (float)height_;   // yolo-fastestv2.cpp:189-190
    const size_t src_bytes = (size_t)cols * rows * 3;
    if (cols == width_ && rows == height_) {
      if (hipMemcpyAsync(d_img_, bgr, src_bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail(YFV2_ERR_DEVICE, "upload failed");
    } else {
      if (src_bytes > src_cap_) {
        if (d_src_) (void)hipFree(d_src_);
        d_src_ = nullptr; src_cap_ = 0;
        if (hipMalloc(&d_src_, src_bytes) != hipSuccess) return fail(YFV2_ERR_DEVICE, "hipMalloc for the source frame failed");
        src_cap_ = src_bytes;
      }
      if (hipMemcpyAsync(d_src_, bgr, src_bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail(YFV2_ERR_DEVICE, "upload failed");
      const int rr = yfv2_resize_u8(h_, static_cast<const uint8_t*>(d_src_), 1, rows, cols, static_cast<uint8_t*>(d_img_), stream_);
      if (rr != YFV2_OK) return fail(rr, yfv2_last_error(h_));
    }
    const int rc = yfv2_detect_u8(h_, static_cast<const uint8_t*>(d_img_), 1, thresh, (double)iou_thresh, static_cast<float*>(d_dets_),
                                  static_cast<int32_t*>(d_idx_), static_cast<int32_t*>(d_cnt_), stream_);
    if (rc != YFV2_OK) return fail(rc, yfv2_last_error(h_));
    int32_t cnt = 0;
    float rowsbuf[300 * 6];
    if (hipMemcpyAsync(&cnt, d_cnt_, sizeof(cnt), hipMemcpyDeviceToHost, stream_) != hipSuccess ||
        hipMemcpyAsync(rowsbuf, d_dets_, sizeof(rowsbuf), hipMemcpyDeviceToHost, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess)
      return fail(YFV2_ERR_DEVICE, "download failed");
    /* the range guard of the default plan (yfv2.h yfv2_nonfinite): the stream was waited for just above, so the query costs a
       host memory read.  A fine-tuned model whose activations leave +-4094 must come back as an ERROR here, not as boxes -
       the reference's fp32 convolutions (model/backbone/shufflenetv2.py:19-32) have no such cliff. */
    int32_t tripped = 0;
    const int gr = yfv2_nonfinite(h_, &tripped, stream_);
    if (gr != YFV2_OK) return fail(gr, yfv2_last_error(h_));
    if (tripped)
      return fail(YFV2_ERR_RANGE, "detection: an activation left the range of the default (fp16x3) plan - |activation| >= 4094; the result is "
                                  "invalid.  Run this model on a handle created with yfv2_plan.fp32_matrix = 1 (the former YFV2_BF6=0: fp32 matrix instructions, no such bound)");
    for (int i = 0; i < cnt; ++i) {
      const float* r = rowsbuf + 6 * i;
      TargetBox b;
      b.x1 = (int)(r[0] * scaleW); b.y1 = (int)(r[1] * scaleH); b.x2 = (int)(r[2] * scaleW); b.y2 = (int)(r[3] * scaleH);
      b.score = r[4]; b.cate = (int)r[5];
      dst.push_back(b);
    }
    return YFV2_OK;
  }

 private:
  int fail(int rc, const std::string& msg) { err_ = msg; return rc; }
  yfv2_handle h_ = nullptr;
  int rc_ = YFV2_OK;
  int width_, height_;
  std::string err_;
  void* d_img_ = nullptr; void* d_dets_ = nullptr; void* d_idx_ = nullptr; void* d_cnt_ = nullptr;
  hipStream_t stream_ = nullptr;
  void* d_src_ = nullptr;
  size_t src_cap_ = 0;
};

}  // namespace yfv2

#endif /* YFV2_HPP */
