// Canny edge detector on gfx950: `cv2.Canny(img, low, high)` as the reference calls it (`annotator/canny/__init__.py:4-6`,
// `canny2image_torch.py:33`): aperture 3, L1 gradient magnitude, uint8 HWC input of 1..4 channels, uint8 0 / 255 output.
// OpenCV itself (pinned opencv-contrib-python 4.3.0.36, `environment.yaml:15`) is not in the reference tree; the integer
// algorithm of its imgproc/canny.cpp is restated in oracle/canny_oracle.py and these kernels are bit-exact to that oracle.
//
// HBM-bound byte work, four small kernels over the image (no reshaping into GEMMs):
//   gradient   3x3 Sobel per channel (replicated border), |dx| + |dy|, channel of the largest magnitude per pixel
//   nms        fixed-point sector test (TG22 = 13573), thresholds -> map {0 candidate, 1 no edge, 2 edge}
//   hysteresis 32x32 tiles (+1 halo) iterate to a local fixed point in LDS; the launch is repeated until no tile changes
//              (edges are the 8-connected components of candidates that contain an edge: order-independent, deterministic)
//   finalize   255 / 0 (+ optional fp32 control tensor [3][H][W] = edges / 255, `canny2image_torch.py:34-38`)
#include "kernels.h"

namespace sdeo {

static inline dim3 grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

__global__ __launch_bounds__(256) void canny_grad_kernel(const uint8_t* __restrict__ img, int H, int W, int C, short* __restrict__ dx,
                                                         short* __restrict__ dy, short* __restrict__ mag) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const int xm = max(x - 1, 0), xp = min(x + 1, W - 1), ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  int bdx = 0, bdy = 0, bm = -1;
  for (int k = 0; k < C; ++k) {
    auto px = [&](int yy, int xx) { return (int)img[((size_t)yy * W + xx) * C + k]; };
    const int a = px(ym, xm), b = px(ym, x), c = px(ym, xp), d = px(y, xm), f = px(y, xp), g = px(yp, xm), h = px(yp, x), i = px(yp, xp);
    const int gx = (c + 2 * f + i) - (a + 2 * d + g), gy = (g + 2 * h + i) - (a + 2 * b + c);
    const int m = abs(gx) + abs(gy);
    if (m > bm) { bm = m; bdx = gx; bdy = gy; }      // strictly greater: the first channel wins a tie
  }
  const size_t o = (size_t)y * W + x;
  dx[o] = (short)bdx; dy[o] = (short)bdy; mag[o] = (short)bm;
}

__global__ __launch_bounds__(256) void canny_nms_kernel(const short* __restrict__ dx, const short* __restrict__ dy,
                                                        const short* __restrict__ mag, int H, int W, int low, int high,
                                                        uint8_t* __restrict__ map) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  auto M = [&](int yy, int xx) { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : (int)mag[(size_t)yy * W + xx]; };
  const size_t o = (size_t)y * W + x;
  const int m = mag[o];
  uint8_t r = 1;
  if (m > low) {
    const int xs = dx[o], ys = dy[o];
    const long ax = abs(xs), ay = (long)abs(ys) << 15;
    const long tg22x = ax * 13573;
    bool keep;
    if (ay < tg22x) keep = m > M(y, x - 1) && m >= M(y, x + 1);
    else {
      const long tg67x = tg22x + (ax << 16);
      if (ay > tg67x) keep = m > M(y - 1, x) && m >= M(y + 1, x);
      else {
        const int s = (xs ^ ys) < 0 ? -1 : 1;
        keep = m > M(y - 1, x - s) && m > M(y + 1, x + s);
      }
    }
    if (keep) r = m > high ? 2 : 0;
  }
  map[o] = r;
}

__global__ __launch_bounds__(1024) void canny_hyst_kernel(uint8_t* __restrict__ map, int H, int W, int* __restrict__ changed) {
  __shared__ uint8_t t[34][36];
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < 34 * 34; i += 1024) {
    const int ty = i / 34, tx = i - ty * 34;
    const int gy = y0 + ty - 1, gx = x0 + tx - 1;
    t[ty][tx] = (gy < 0 || gy >= H || gx < 0 || gx >= W) ? 1 : map[(size_t)gy * W + gx];
  }
  __syncthreads();
  const int gx = x0 + lx, gy = y0 + ly;
  const bool inside = gx < W && gy < H;
  bool mine = false;
  for (;;) {
    bool ch = false;
    if (inside && t[ly + 1][lx + 1] == 0) {
      // monotone 0 -> 2 updates: a neighbour read that races with its writer sees either value, both are valid states
      const bool s = t[ly][lx] == 2 || t[ly][lx + 1] == 2 || t[ly][lx + 2] == 2 || t[ly + 1][lx] == 2 || t[ly + 1][lx + 2] == 2 ||
                     t[ly + 2][lx] == 2 || t[ly + 2][lx + 1] == 2 || t[ly + 2][lx + 2] == 2;
      if (s) { t[ly + 1][lx + 1] = 2; ch = true; mine = true; }
    }
    if (!__syncthreads_or(ch)) break;
  }
  if (mine) {
    map[(size_t)gy * W + gx] = 2;
    *changed = 1;
  }
}

__global__ __launch_bounds__(256) void canny_final_kernel(const uint8_t* __restrict__ map, int64_t n, uint8_t* __restrict__ edges,
                                                          float* __restrict__ control) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool e = map[i] == 2;
    if (edges) edges[i] = e ? 255 : 0;
    if (control) {
      const float v = e ? 1.0f : 0.0f;             // 255 / 255
      control[i] = v; control[n + i] = v; control[2 * n + i] = v;
    }
  }
}

size_t canny_workspace_bytes(int H, int W) { return (size_t)H * W * 7 + 512; }

int canny_u8(const uint8_t* img, int H, int W, int C, float low_threshold, float high_threshold, uint8_t* edges, float* control,
             void* workspace, size_t workspace_bytes, hipStream_t stream) {
  SDEO_CHECK(img && (edges || control) && workspace, "canny: null argument");
  SDEO_CHECK(H >= 1 && W >= 1 && C >= 1 && C <= 4, "canny: bad image %dx%dx%d (1..4 channels)", H, W, C);
  SDEO_CHECK(workspace_bytes >= canny_workspace_bytes(H, W), "canny: workspace too small (%zu < %zu)", workspace_bytes,
             canny_workspace_bytes(H, W));
  int low = (int)floorf(low_threshold), high = (int)floorf(high_threshold);
  if (low > high) { const int t = low; low = high; high = t; }
  const size_t n = (size_t)H * W;
  char* ws = static_cast<char*>(workspace);
  int* changed = reinterpret_cast<int*>(ws);
  short* dx = reinterpret_cast<short*>(ws + 512);
  short* dy = dx + n;
  short* mag = dy + n;
  uint8_t* map = reinterpret_cast<uint8_t*>(mag + n);
  const dim3 g8(cdiv(W, 32), cdiv(H, 8)), g32(cdiv(W, 32), cdiv(H, 32));
  hipLaunchKernelGGL(canny_grad_kernel, g8, dim3(256), 0, stream, img, H, W, C, dx, dy, mag);
  hipLaunchKernelGGL(canny_nms_kernel, g8, dim3(256), 0, stream, dx, dy, mag, H, W, low, high, map);
  SDEO_HIP(hipGetLastError());
  // hysteresis to a global fixed point: every launch settles each tile; a chain that crosses a tile boundary needs one more
  // launch per crossing (it may re-enter a tile many times), so the loop runs until a launch changes nothing.  Every launch
  // that continues the loop turns at least one candidate into an edge: n + 1 launches is a true upper bound.  The flag read
  // makes this entry point synchronising.
  const size_t max_iter = n + 1;
  for (size_t it = 0; it < max_iter; ++it) {
    int h_changed = 0;
    SDEO_HIP(hipMemsetAsync(changed, 0, sizeof(int), stream));
    hipLaunchKernelGGL(canny_hyst_kernel, g32, dim3(1024), 0, stream, map, H, W, changed);
    SDEO_HIP(hipGetLastError());
    SDEO_HIP(hipMemcpyAsync(&h_changed, changed, sizeof(int), hipMemcpyDeviceToHost, stream));
    SDEO_HIP(hipStreamSynchronize(stream));
    if (!h_changed) break;
  }
  hipLaunchKernelGGL(canny_final_kernel, grid_for((int64_t)n), dim3(256), 0, stream, map, (int64_t)n, edges, control);
  SDEO_HIP(hipGetLastError());
  return 0;
}

}  // namespace sdeo
