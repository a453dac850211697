// TEMPORARY stubs (replaced by the executor).
#include "../../include/sdeo.h"
#include "kernels.h"
using namespace sdeo;
extern "C" {
int sdeo_create(const sdeo_config*, sdeo_handle*) { return fail("not implemented"); }
int sdeo_destroy(sdeo_handle) { return fail("not implemented"); }
int sdeo_load_weight(sdeo_handle, const char*, const float*, const int64_t*, int, int) { return fail("not implemented"); }
int sdeo_finalize_weights(sdeo_handle) { return fail("not implemented"); }
int sdeo_num_weights(sdeo_handle) { return 0; }
int sdeo_weight_info(sdeo_handle, int, const char**, int64_t*, int*) { return fail("not implemented"); }
int sdeo_configure(sdeo_handle, int, int, int) { return fail("not implemented"); }
int sdeo_controlnet_forward(sdeo_handle, const float*, const float*, const int64_t*, const float*, float* const*, int, void*) { return fail("not implemented"); }
int sdeo_unet_forward(sdeo_handle, const float*, const int64_t*, const float*, const float* const*, const float*, int, float*, void*) { return fail("not implemented"); }
int sdeo_apply_model(sdeo_handle, const float*, const float*, const int64_t*, const float*, const float*, int, int, float*, void*) { return fail("not implemented"); }
int sdeo_vae_decode(sdeo_handle, const float*, int, float*, uint8_t*, void*) { return fail("not implemented"); }
size_t sdeo_device_bytes(sdeo_handle) { return 0; }
}
