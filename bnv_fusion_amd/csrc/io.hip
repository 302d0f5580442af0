// Host-side helpers for the dataset formats around the hot path (SURVEY.md section 8 f: "data formats either
// side").  The reference reads its depth frames with cv2.imread(path, -1) (src/utils/common.py:93): 16-bit
// greyscale PNGs in millimetres.  OpenCV is not a dependency here; the container format is parsed in Python
// (bnv_fusion_amd/datasets.py: chunks + zlib) and only the per-scanline filter reversal -- a byte-serial loop --
// lives here.  No GPU code in this file.
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/bnv_fusion.h"

extern "C" {

// PNG filter reversal (RFC 2083 section 6).  `raw`: height scanlines, each 1 filter-type byte + row_bytes data
// bytes (the inflated IDAT stream); `out`: height * row_bytes bytes; bpp = bytes per complete pixel (>= 1).
int bnv_png_unfilter(const uint8_t* raw, int height, int row_bytes, int bpp, uint8_t* out) {
  if (!raw || !out || height < 0 || row_bytes < 0 || bpp < 1) return BNV_ERR_INVALID_ARGUMENT;
  for (int y = 0; y < height; ++y) {
    const uint8_t* in = raw + (size_t)y * (row_bytes + 1);
    const int ft = in[0];
    ++in;
    uint8_t* cur = out + (size_t)y * row_bytes;
    const uint8_t* up = y ? cur - row_bytes : nullptr;
    for (int x = 0; x < row_bytes; ++x) {
      const int a = x >= bpp ? cur[x - bpp] : 0;
      const int b = up ? up[x] : 0;
      const int c = (up && x >= bpp) ? up[x - bpp] : 0;
      int pred;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return BNV_ERR_INVALID_ARGUMENT;
      }
      cur[x] = (uint8_t)(in[x] + pred);
    }
  }
  return BNV_OK;
}

}  // extern "C"
