/* host.h -- C interface of libpt_host.so: the host-side mirror of the reference's scene executables and
 * Render_command (the parts of the reference that sit ABOVE the integrator boundary). */
#ifndef PT_HOST_H
#define PT_HOST_H
#include <stdint.h>

#include "../../include/ptx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene; /* owns the arrays a ptx_scene_desc points into */

const ptx_scene_desc* pth_scene_desc(pth_scene* s);
void pth_scene_free(pth_scene* s);

/* Camera.create (path_tracer/src/camera.ml:58-83): view = the four fields Camera.ray reads;
 * look_at (optional) = Mat4.look_at rows, row-major 4x4 */
void pth_camera_create(const double eye[3], const double target[3], const double up[3], double aspect, double fov_deg,
                       ptx_camera* view_out, double look_at_out[16]);

/* shirley_spheres/bin/main.ml (Random.init seed; the reference uses 42) */
pth_scene* pth_scene_shirley(int32_t width, int32_t height, int32_t no_simd, int64_t seed);
/* cornell-box/bin/main.ml geometry + documented ceiling emitter */
pth_scene* pth_scene_cornell(int32_t width, int32_t height, double ceiling_emit);
/* ganesha/bin/main.ml camera / floor / material over a synthetic mesh of ~n_target triangles */
pth_scene* pth_scene_ganesha_like(int32_t width, int32_t height, int32_t n_target, uint64_t seed);

/* lights of the photon-mapped scenes (camera space): cornell-box/bin/main.ml:225-228, ganesha/bin/main.ml:267-282 */
int32_t pth_lights_cornell(int32_t width, int32_t height, ptx_light* out);      /* writes 1 */
int32_t pth_lights_ganesha(pth_scene* ganesha_scene, ptx_light* out /* 2 */); /* writes 2 */
/* save_image's gamma (progressive_photon_map.ml:398-410): avg = (sum / n) ** (1 / 2.2), in place into out */
void pth_ppm_gamma(const double* img_sum, int64_t count, int32_t n, double* out);

/* ---- PLY (ply_format/src/ply.ml) ---- */
typedef struct pth_ply pth_ply;
const char* pth_last_error(void);
/* Ply.of_bigstring over the file's bytes: NULL + pth_last_error() on failure */
pth_ply* pth_ply_load(const char* path);
void pth_ply_free(pth_ply* p);
/* number of rows of an element (fixed-width) or of a list PROPERTY (keyed by property name, ply.ml:234); -1 if absent */
int64_t pth_ply_count(const pth_ply* p, const char* key);
const double* pth_ply_floats(const pth_ply* p, const char* element, const char* property);  /* Column.Floats */
const int64_t* pth_ply_ints(const pth_ply* p, const char* element, const char* property);   /* Column.Ints */
const int64_t* pth_ply_rows(pth_ply* p, const char* list_property, const int32_t** lengths_out); /* Column.Rows, flattened */
/* ganesha/bin/main.ml with -ganesha-ply PATH: Mesh.create + floor + camera; sky background (extension) */
pth_scene* pth_scene_ganesha_ply(const char* path, int32_t width, int32_t height);
/* the synthetic stand-in mesh written as a PLY file with the real model's layout */
int32_t pth_write_ganesha_like_ply(const char* path, int32_t n_target, uint64_t seed);

/* Bimage_unix.Stb.write of the f64 image (render_command.ml:66-70): 8-bit RGB PNG, v -> int(v*255) clamped.
 * returns 0 on success */
int32_t pth_write_png(const char* path, int32_t width, int32_t height, const double* rgb);

#ifdef __cplusplus
}
#endif
#endif
