/* Private entry points of libsdeo.so: tuning and measurement hooks used by tools/ and a few op tests.  NOT part of the
 * drop-in boundary (include/sdeo.h): nothing a reference-side binding needs is declared here. */
#ifndef SDEO_INTERNAL_H
#define SDEO_INTERNAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* force tile config / split-K of the following conv/GEMM launches (-1, 0 = plan table / heuristic) */
void sdeo_debug_force_gemm_plan(int tile, int splitk);
void sdeo_debug_force_gemm_order(int order); /* -1 heuristic, 0 M-fastest, 1 N-fastest tile order within an XCD */
/* name of the kernel instantiation sdeo_conv2d_nhwc_f16 would launch for this problem (plan table / forced plan / heuristic) */
const char* sdeo_debug_conv2d_kernel_name(int n, int h, int w, int cin, int cout, int ksize, int stride, int upsample2x);
/* measurement builds only (python -m stablediffusioneo_amd.build --debug, SDEO_DBG_GEMM bit 6): per-workgroup phase stamps of
 * the last GEMM (which = 0) / halo conv (1) launch, 8 x uint64 per workgroup in 10 ns units */
int sdeo_debug_read_stamps(int which, unsigned long long* out, int n);
/* y = silu(x) on n fp16 elements: launch-floor probe for tools/launch_floor.py */
int sdeo_debug_silu(void* y, const void* x, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDEO_INTERNAL_H */
