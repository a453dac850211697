// Canny edge detector on gfx950: `cv2.Canny(img, low, high)` as the reference calls it (`annotator/canny/__init__.py:4-6`,
// `canny2image_torch.py:33`): aperture 3, L1 gradient magnitude, uint8 HWC input of 1..4 channels, uint8 0 / 255 output.
// OpenCV itself (pinned opencv-contrib-python 4.3.0.36, `environment.yaml:15`) is not in the reference tree; the integer
// algorithm of its imgproc/canny.cpp is restated in oracle/canny_oracle.py and these kernels are bit-exact to that oracle.
//
// HBM-bound byte work, six small kernels over the image (no reshaping into GEMMs, no host loop: hipGraph-capturable):
//   gradient   3x3 Sobel per channel (replicated border), |dx| + |dy|, channel of the largest magnitude per pixel
//   nms        fixed-point sector test (TG22 = 13573), thresholds -> map {0 candidate, 1 no edge, 2 edge}
//   hysteresis the 8-connected components of {candidate, edge} pixels by union-find on pixel indices (init / merge / mark the
//              components that hold an edge): a candidate is an edge iff its component holds one -- order-independent,
//              deterministic, and the same fixed point OpenCV's stack-based flood reaches
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

// ---- hysteresis without a host loop: a candidate pixel (map 0) becomes an edge when its 8-connected component of
// {candidates, strong edges} contains a strong edge (map 2) -- exactly the fixed point cv2.Canny's stack-based flood reaches.
// Components by union-find on pixel indices (parents only ever decrease: atomicMin), three launches, no iteration count that
// depends on the image, nothing for the host to read back: the entry point is hipGraph-capturable.
// Reads of a parent that race with an atomicMin may return an older parent: still an element of the same set with a smaller or
// equal index, so `find` terminates and `unite` (which re-validates through the value its atomic returns) stays correct.
__device__ __forceinline__ int uf_find(const int* L, int i) {
  for (;;) {
    const int pnt = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pnt == i) return i;
    i = pnt;
  }
}
__device__ __forceinline__ void uf_unite(int* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }                  // a < b: hang b under a
    const int old = atomicMin(L + b, a);
    if (old == b) return;                                           // b was still a root: done
    b = old;                                                        // somebody re-parented b meanwhile: merge a with that parent
  }
}

__global__ __launch_bounds__(256) void canny_label_init_kernel(const uint8_t* __restrict__ map, int64_t n, int* __restrict__ L,
                                                               int* __restrict__ strong) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    L[i] = map[i] != 1 ? (int)i : -1;
    strong[i] = 0;
  }
}

__global__ __launch_bounds__(256) void canny_label_merge_kernel(const uint8_t* __restrict__ map, int H, int W, int* __restrict__ L) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= W || y >= H) return;
  const int i = y * W + x;
  if (map[i] == 1) return;
  // the four neighbours that precede the pixel in raster order: every adjacent pair is visited once
  if (x > 0 && map[i - 1] != 1) uf_unite(L, i, i - 1);
  if (y > 0) {
    if (map[i - W] != 1) uf_unite(L, i, i - W);
    if (x > 0 && map[i - W - 1] != 1) uf_unite(L, i, i - W - 1);
    if (x + 1 < W && map[i - W + 1] != 1) uf_unite(L, i, i - W + 1);
  }
}

__global__ __launch_bounds__(256) void canny_label_strong_kernel(const uint8_t* __restrict__ map, int64_t n, const int* __restrict__ L,
                                                                 int* __restrict__ strong) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if (map[i] == 2) strong[uf_find(L, (int)i)] = 1;               // benign race: every writer stores 1
}

__global__ __launch_bounds__(256) void canny_final_kernel(const uint8_t* __restrict__ map, int64_t n, const int* __restrict__ L,
                                                          const int* __restrict__ strong, uint8_t* __restrict__ edges,
                                                          float* __restrict__ control) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool e = map[i] != 1 && strong[uf_find(L, (int)i)] != 0;
    if (edges) edges[i] = e ? 255 : 0;
    if (control) {
      const float v = e ? 1.0f : 0.0f;             // 255 / 255
      control[i] = v; control[n + i] = v; control[2 * n + i] = v;
    }
  }
}

size_t canny_workspace_bytes(int H, int W) { return (size_t)H * W * 15 + 512; }

int canny_u8(const uint8_t* img, int H, int W, int C, float low_threshold, float high_threshold, uint8_t* edges, float* control,
             void* workspace, size_t workspace_bytes, hipStream_t stream) {
  SDEO_CHECK(img && (edges || control) && workspace, "canny: null argument");
  SDEO_CHECK(H >= 1 && W >= 1 && C >= 1 && C <= 4 && (int64_t)H * W < (1ll << 31), "canny: bad image %dx%dx%d (1..4 channels)", H, W, C);
  SDEO_CHECK(workspace_bytes >= canny_workspace_bytes(H, W), "canny: workspace too small (%zu < %zu)", workspace_bytes,
             canny_workspace_bytes(H, W));
  SDEO_CHECK((reinterpret_cast<uintptr_t>(workspace) & 3) == 0, "canny: workspace must be 4-byte aligned");
  int low = (int)floorf(low_threshold), high = (int)floorf(high_threshold);
  if (low > high) { const int t = low; low = high; high = t; }
  const size_t n = (size_t)H * W;
  char* ws = static_cast<char*>(workspace);
  int* L = reinterpret_cast<int*>(ws + 512);                // labels / strong flags first: 4-byte aligned
  int* strong = L + n;
  short* dx = reinterpret_cast<short*>(strong + n);
  short* dy = dx + n;
  short* mag = dy + n;
  uint8_t* map = reinterpret_cast<uint8_t*>(mag + n);
  const dim3 g8(cdiv(W, 32), cdiv(H, 8));
  hipLaunchKernelGGL(canny_grad_kernel, g8, dim3(256), 0, stream, img, H, W, C, dx, dy, mag);
  hipLaunchKernelGGL(canny_nms_kernel, g8, dim3(256), 0, stream, dx, dy, mag, H, W, low, high, map);
  hipLaunchKernelGGL(canny_label_init_kernel, grid_for((int64_t)n), dim3(256), 0, stream, map, (int64_t)n, L, strong);
  hipLaunchKernelGGL(canny_label_merge_kernel, g8, dim3(256), 0, stream, map, H, W, L);
  hipLaunchKernelGGL(canny_label_strong_kernel, grid_for((int64_t)n), dim3(256), 0, stream, map, (int64_t)n, L, strong);
  hipLaunchKernelGGL(canny_final_kernel, grid_for((int64_t)n), dim3(256), 0, stream, map, (int64_t)n, L, strong, edges, control);
  SDEO_HIP(hipGetLastError());
  return 0;
}

}  // namespace sdeo
