// cv2.resize for 8-bit HWC images on the GPU (second half): the two interpolations
// `annotator/util.py:28-38` (`resize_image`) asks for -- INTER_LANCZOS4 when enlarging, INTER_AREA otherwise.
// Byte / integer work, HBM-bound: one thread per output element gathers its taps through the coefficient tables
// (built on the host from the image geometry: O(W + H) entries) and evaluates OpenCV's arithmetic:
//   Lanczos4: 8 x 8 taps, short coefficients (x 2048), 32-bit integer sums, (sum + 2^21) >> 22, saturate  -- the separable
//             passes of `HResizeLanczos4` / `VResizeLanczos4` are exact integer arithmetic, so one fused evaluation gives
//             the same integer;
//   area:     float weights, multiply and add kept separate (no fused multiply-add: OpenCV's generic C++ loop has none)
//             and in the table order of `resizeArea_`, cvRound (ties to even), saturate.
#include "kernels.h"

namespace sdeo {

__global__ __launch_bounds__(256) void resize_lanczos4_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int h, int w,
                                                              int c, int dh, int dw, const int* __restrict__ x0,
                                                              const short* __restrict__ ax, const int* __restrict__ y0,
                                                              const short* __restrict__ by) {
  const int64_t total = (int64_t)dh * dw * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int dx = (int)((i / c) % dw);
    const int dy = (int)(i / ((int64_t)c * dw));
    int xi[8], xa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int s = x0[dx] + k;
      xi[k] = (s < 0 ? 0 : (s > w - 1 ? w - 1 : s)) * c + ch;
      xa[k] = ax[dx * 8 + k];
    }
    int acc = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int sy = y0[dy] + r;
      const uint8_t* row = src + (size_t)(sy < 0 ? 0 : (sy > h - 1 ? h - 1 : sy)) * w * c;
      int rs = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) rs += (int)row[xi[k]] * xa[k];
      acc += rs * (int)by[dy * 8 + r];
    }
    const int v = (acc + (1 << 21)) >> 22;
    dst[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

__global__ __launch_bounds__(256) void resize_area_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int w, int c, int dh,
                                                          int dw, const int* __restrict__ xstart, const int* __restrict__ xidx,
                                                          const float* __restrict__ xw, const int* __restrict__ ystart,
                                                          const int* __restrict__ yidx, const float* __restrict__ yw) {
#pragma clang fp contract(off)      // OpenCV's loop multiplies, rounds, then adds: a fused multiply-add would round once
  const int64_t total = (int64_t)dh * dw * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int dx = (int)((i / c) % dw);
    const int dy = (int)(i / ((int64_t)c * dw));
    float sum = 0.f;
    for (int yk = ystart[dy]; yk < ystart[dy + 1]; ++yk) {
      const uint8_t* row = src + (size_t)yidx[yk] * w * c + ch;
      float buf = 0.f;
      for (int xk = xstart[dx]; xk < xstart[dx + 1]; ++xk) buf = buf + (float)row[(size_t)xidx[xk] * c] * xw[xk];
      const float t = buf * yw[yk];
      sum = yk == ystart[dy] ? t : sum + t;
    }
    const float r = rintf(sum);
    dst[i] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
  }
}

// INTER_AREA when BOTH axes shrink by integer factors (sy = h / dh, sx = w / dw): OpenCV leaves `resizeArea_` for `resizeAreaFast_`
// (resize.cpp: `is_area_fast`), which is integer arithmetic -- restated here, parity unpinned like the rest of this file (no cv2 in
// the image):  2 x 2 cells of 8-bit images: (a + b + c + d + 2) >> 2 (`ResizeAreaFastVec`, round half UP);  any other cell: the
// integer sum of the sy x sx cell times the float 1 / (sx * sy), cvRound (ties to even), saturate (`ResizeAreaFast_Invoker`).
__global__ __launch_bounds__(256) void resize_area_fast_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int w, int c,
                                                               int dh, int dw, int sy, int sx) {
  const int64_t total = (int64_t)dh * dw * c;
  const float scale = 1.f / (float)(sx * sy);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    const int dx = (int)((i / c) % dw);
    const int dy = (int)(i / ((int64_t)c * dw));
    const uint8_t* cell = src + ((size_t)dy * sy * w + (size_t)dx * sx) * c + ch;
    int sum = 0;
    for (int y = 0; y < sy; ++y)
      for (int x = 0; x < sx; ++x) sum += (int)cell[((size_t)y * w + x) * c];
    int v;
    if (sx == 2 && sy == 2) {
      v = (sum + 2) >> 2;
    } else {
      const float r = rintf((float)sum * scale);
      v = (int)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
    }
    dst[i] = (uint8_t)v;
  }
}

static inline dim3 resize_grid(int64_t n) {
  int64_t b = cdiv64(n, 256);
  return dim3((unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)));
}

int resize_lanczos4_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, const int* x0, const short* ax,
                       const int* y0, const short* by, hipStream_t stream) {
  SDEO_CHECK(dst && src && x0 && ax && y0 && by, "resize_lanczos4: null operand");
  SDEO_CHECK(h > 0 && w > 0 && c >= 1 && c <= 4 && dh > 0 && dw > 0, "resize_lanczos4: bad geometry %dx%dx%d -> %dx%d", h, w, c, dh, dw);
  hipLaunchKernelGGL(resize_lanczos4_kernel, resize_grid((int64_t)dh * dw * c), dim3(256), 0, stream, dst, src, h, w, c, dh, dw, x0, ax, y0, by);
  SDEO_HIP(hipGetLastError());
  return 0;
}

int resize_area_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, const int* xstart, const int* xidx,
                   const float* xw, const int* ystart, const int* yidx, const float* yw, hipStream_t stream) {
  SDEO_CHECK(dst && src && xstart && xidx && xw && ystart && yidx && yw, "resize_area: null operand");
  SDEO_CHECK(h > 0 && w > 0 && c >= 1 && c <= 4 && dh > 0 && dw > 0, "resize_area: bad geometry %dx%dx%d -> %dx%d", h, w, c, dh, dw);
  hipLaunchKernelGGL(resize_area_kernel, resize_grid((int64_t)dh * dw * c), dim3(256), 0, stream, dst, src, w, c, dh, dw, xstart, xidx, xw,
                     ystart, yidx, yw);
  SDEO_HIP(hipGetLastError());
  return 0;
}

int resize_area_fast_u8(uint8_t* dst, const uint8_t* src, int h, int w, int c, int dh, int dw, hipStream_t stream) {
  SDEO_CHECK(dst && src, "resize_area_fast: null operand");
  SDEO_CHECK(h > 0 && w > 0 && c >= 1 && c <= 4 && dh > 0 && dw > 0 && h % dh == 0 && w % dw == 0,
             "resize_area_fast: %dx%dx%d -> %dx%d is not a shrink by integer factors", h, w, c, dh, dw);
  hipLaunchKernelGGL(resize_area_fast_kernel, resize_grid((int64_t)dh * dw * c), dim3(256), 0, stream, dst, src, w, c, dh, dw, h / dh, w / dw);
  SDEO_HIP(hipGetLastError());
  return 0;
}

}  // namespace sdeo
