// TEMPORARY: raster entry points return W3D_ERR_UNSUPPORTED until raster_fwd/bwd.hip land.
#include "common.h"
extern "C" size_t w3d_raster_geom_bytes(int) { return 0; }
extern "C" size_t w3d_raster_image_bytes(int, int) { return 0; }
extern "C" size_t w3d_raster_binning_bytes(int64_t) { return 0; }
extern "C" int w3d_raster_forward(const W3dRasterIn *, void *, void *, w3d_alloc_fn, void *, float *, float *, int32_t *, int *, w3d_stream_t) { return W3D_ERR_UNSUPPORTED; }
extern "C" int w3d_raster_backward(const W3dRasterIn *, const int32_t *, const W3dRasterGradIn *, W3dRasterGradOut *, const void *, const void *, const void *, int, w3d_stream_t) { return W3D_ERR_UNSUPPORTED; }
extern "C" int w3d_mark_visible(const float *, const float *, const float *, int, uint8_t *, w3d_stream_t) { return W3D_ERR_UNSUPPORTED; }
extern "C" int w3d_activations_fwd(const float *, const float *, const float *, int, float *, float *, float *, w3d_stream_t) { return W3D_ERR_UNSUPPORTED; }
extern "C" int w3d_activations_bwd(const float *, const float *, const float *, int, const float *, const float *, const float *, float *, float *, float *, w3d_stream_t) { return W3D_ERR_UNSUPPORTED; }
