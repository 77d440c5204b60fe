// Test driver for include/yfv2.hpp (built by __graft_entry__.build(), run by tests/test_gpu_parity.py):
//   yfv2_cpp_test <weights.yfv2w> <anchors: 12 numbers, comma separated> <image.raw> <cols> <rows> [thresh] [iou]
// prints one line per detection: x1 y1 x2 y2 cate score
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/yfv2.hpp"

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s weights anchors image.raw cols rows [thresh] [iou]\n", argv[0]); return 2; }
  double anchors[12];
  {
    const char* p = argv[2];
    for (int i = 0; i < 12; ++i) { char* e; anchors[i] = std::strtod(p, &e); p = (*e == ',') ? e + 1 : e; }
  }
  const int cols = std::atoi(argv[4]), rows = std::atoi(argv[5]);
  const float thresh = argc > 6 ? (float)std::atof(argv[6]) : 0.3f, iou = argc > 7 ? (float)std::atof(argv[7]) : 0.4f;
  std::vector<unsigned char> img((size_t)cols * rows * 3);
  FILE* f = std::fopen(argv[3], "rb");
  if (!f || std::fread(img.data(), 1, img.size(), f) != img.size()) { std::fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
  std::fclose(f);
  yfv2::Detector det(80, anchors);
  if (!det.ok()) { std::fprintf(stderr, "create: %s\n", det.lastError()); return 3; }
  if (det.loadModel(argv[1]) != 0) { std::fprintf(stderr, "loadModel: %s\n", det.lastError()); return 3; }
  std::vector<yfv2::TargetBox> boxes;
  if (det.detection(img.data(), cols, rows, boxes, thresh, iou) != 0) { std::fprintf(stderr, "detection: %s\n", det.lastError()); return 3; }
  for (const auto& b : boxes) std::printf("%d %d %d %d %d %.9g\n", b.x1, b.y1, b.x2, b.y2, b.cate, b.score);
  return 0;
}
