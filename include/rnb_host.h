/* rnb_host.h — C-ABI of librnb_host.so: the CPU-side helpers the Python data-preparation / albedo-scaling stages call
 * where the reference uses cv2 / trimesh+embree (neither exists on the target image). No GPU code, no torch types.
 *
 *   PNG I/O        replaces cv2.imread(path, IMREAD_UNCHANGED) / cv2.imwrite for 8/16-bit PNG
 *                  (rnb_neus2/image_io.py:15-73, rnb_neus2/prepare.py:23-42,150-205)
 *   ray casting    replaces trimesh.ray.intersects_location (embree) in rnb_neus2/albedo_scaling.py:289-330
 *
 * All functions return 0 on success, negative on error (message via rnb_host_last_error()).
 */
#ifndef RNB_HOST_H
#define RNB_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* rnb_host_last_error(void);

/* width/height and the layout an unchanged decode reports: channels 1 (grey), 2 (grey+alpha), 3, 4; depth 8 or 16. */
int rnb_png_info(const char* path, uint32_t* width, uint32_t* height, int32_t* channels, int32_t* depth);
/* Decode to RGBA, 4 x uint16 per pixel, row-major, with the widening rules of stbi_load_16 (8-bit v -> v*257; grey
 * replicated; missing alpha = 65535). out must hold width*height*4 uint16. */
int rnb_png_read_rgba16(const char* path, uint16_t* out);
/* Encode interleaved samples (RGB order). channels 1..4, depth 8 (uint8) or 16 (uint16), zlib level 0..9. */
int rnb_png_write(const char* path, const void* data, uint32_t width, uint32_t height, int32_t channels, int32_t depth, int32_t level);

/* Triangle-mesh ray casting (bounding-volume hierarchy, double-precision intersection tests, OpenMP over rays). */
typedef struct rnb_bvh rnb_bvh;
int rnb_bvh_create(const float* vertices, uint32_t n_vertices, const uint32_t* triangles, uint32_t n_triangles, rnb_bvh** out);
int rnb_bvh_destroy(rnb_bvh* bvh);
/* Nearest hit with t > 0 per ray. t_out[i] = distance along the (not necessarily unit) direction, or +inf; tri_out[i] = triangle or -1. */
int rnb_bvh_first_hit(const rnb_bvh* bvh, const double* origins, const double* directions, uint32_t n_rays, double* t_out, int32_t* tri_out);
/* occluded_out[i] = 1 when any triangle is hit with 0 < t < t_max[i]. */
int rnb_bvh_occluded(const rnb_bvh* bvh, const double* origins, const double* directions, const double* t_max, uint32_t n_rays, uint8_t* occluded_out);

#ifdef __cplusplus
}
#endif
#endif
