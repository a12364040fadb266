/* host_capi.h -- C entry points of librt_host.so, the C++ host side (scene build, frame loop), for
 * callers that cannot link C++ (the Python test and benchmark plumbing). */
#ifndef RT_HOST_CAPI_H
#define RT_HOST_CAPI_H

#include <stddef.h>
#include <stdint.h>

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* A zero in n_obj, the thresholds, width, height or ray_count selects the reference's own value (256 instances,
 * thresholds of 128, 1024 x 1024 pixels, 1,048,576 ray slots: main.cpp:270-272, 510, 516); band_rows 0 = the whole
 * image; grid k > 0 = k*k*k instances on a lattice instead of the random field of n_obj. */
typedef struct {
    const char *model_files; /* ';'-separated list of ASCII PLY paths */
    uint32_t n_obj, tri_threshold, aabb_threshold;
    uint32_t width, height, ray_count;
    uint32_t band_first, band_rows;
    uint32_t grid;
} rth_scene_spec;

/* Builds the scene of main.cpp:268-343.  Returns NULL and writes a message into err on failure. */
rth_scene *rth_scene_build(const rth_scene_spec *spec, char *err, size_t err_len);
void rth_scene_free(rth_scene *scene);

const rt_global_data *rth_scene_global(const rth_scene *scene);
const rt_group *rth_scene_groups(const rth_scene *scene, size_t *count);
const rt_matrix *rth_scene_matrices(const rth_scene *scene, size_t *count);
const rt_aabb *rth_scene_aabbs(const rth_scene *scene, size_t *count);
const rt_vertex *rth_scene_vertices(const rth_scene *scene, size_t *count);
const uint32_t *rth_scene_triangles(const rth_scene *scene, size_t *count);
uint32_t rth_scene_pool_size(const rth_scene *scene);
uint32_t rth_scene_row_offset(const rth_scene *scene);
uint64_t rth_scene_resident_triangles(const rth_scene *scene);
/* out[0] = seconds spent reading the model files, out[1] = seconds of hierarchy build and scene assembly,
 * out[2] = number of model files whose parsed arrays came from the binary cache (Model::load). */
void rth_scene_timings(const rth_scene *scene, double out[3]);

/* Writes the dragon stand-in mesh as ASCII PLY; 0 on success. */
int rth_write_standin_ply(const char *path, uint32_t around, uint32_t rings);

/* Look-at camera of model.cpp:11-21. */
void rth_set_camera(rt_camera *cam, uint32_t width, uint32_t height, float tan_fov, const float pos[3],
                    const float view[3], const float up[3]);

#ifdef __cplusplus
}
#endif

#endif
