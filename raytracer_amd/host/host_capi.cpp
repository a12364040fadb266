// host_capi.cpp -- C entry points over the C++ scene builder (see host_capi.h).

#include "host_capi.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "scene.h"

struct rth_scene {
    rt::Scene scene;
};

namespace {

void set_error(char *err, size_t err_len, const std::string &text)
{
    if (err && err_len) std::snprintf(err, err_len, "%s", text.c_str());
}

}  // namespace

extern "C" {

rth_scene *rth_scene_build(const rth_scene_spec *in, char *err, size_t err_len)
{
    if (!in || !in->model_files) {
        set_error(err, err_len, "null scene spec");
        return nullptr;
    }
    rt::SceneSpec spec;
    std::string list = in->model_files;
    for (size_t at = 0; at <= list.size();) {
        size_t end = list.find(';', at);
        if (end == std::string::npos) end = list.size();
        if (end > at) spec.model_files.push_back(list.substr(at, end - at));
        at = end + 1;
    }
    if (in->n_obj) spec.n_obj = in->n_obj;
    if (in->tri_threshold) spec.tri_threshold = in->tri_threshold;
    if (in->aabb_threshold) spec.aabb_threshold = in->aabb_threshold;
    if (in->width) spec.width = in->width;
    if (in->height) spec.height = in->height;
    if (in->ray_count) spec.ray_count = in->ray_count;
    spec.band_first = in->band_first;
    spec.band_rows = in->band_rows;
    spec.grid = in->grid;

    // No exception crosses the C boundary: an allocation that fails is a failed build with a message.
    rth_scene *out = nullptr;
    try {
        out = new rth_scene;
        std::string message;
        if (rt::build_scene(spec, out->scene, message)) return out;
        set_error(err, err_len, message);
    } catch (const std::exception &e) {
        set_error(err, err_len, std::string("scene build: ") + e.what());
    }
    delete out;
    return nullptr;
}

void rth_scene_free(rth_scene *scene)
{
    delete scene;
}

const rt_global_data *rth_scene_global(const rth_scene *s) { return &s->scene.data; }

const rt_group *rth_scene_groups(const rth_scene *s, size_t *count)
{
    *count = s->scene.groups.size();
    return s->scene.groups.data();
}

const rt_matrix *rth_scene_matrices(const rth_scene *s, size_t *count)
{
    *count = s->scene.matrices.size();
    return s->scene.matrices.data();
}

const rt_aabb *rth_scene_aabbs(const rth_scene *s, size_t *count)
{
    *count = s->scene.aabbs.size();
    return s->scene.aabbs.data();
}

const rt_vertex *rth_scene_vertices(const rth_scene *s, size_t *count)
{
    *count = s->scene.vertices.size();
    return s->scene.vertices.data();
}

const uint32_t *rth_scene_triangles(const rth_scene *s, size_t *count)
{
    *count = s->scene.triangles.size();
    return s->scene.triangles.data();
}

uint32_t rth_scene_pool_size(const rth_scene *s) { return s->scene.pool_size; }
uint32_t rth_scene_row_offset(const rth_scene *s) { return s->scene.row_offset; }
uint64_t rth_scene_resident_triangles(const rth_scene *s) { return s->scene.resident_triangles; }
void rth_scene_timings(const rth_scene *s, double out[3])
{
    out[0] = s->scene.ingest_seconds;
    out[1] = s->scene.build_seconds;
    out[2] = double(s->scene.models_from_cache);
}

int rth_write_standin_ply(const char *path, uint32_t around, uint32_t rings)
{
    return rt::write_standin_ply(path, around, rings) ? 0 : -1;
}

void rth_set_camera(rt_camera *cam, uint32_t width, uint32_t height, float tan_fov, const float pos[3],
                    const float view[3], const float up[3])
{
    rt::set_camera(*cam, width, height, tan_fov, rt::Vec3(pos[0], pos[1], pos[2]), rt::Vec3(view[0], view[1], view[2]),
                   rt::Vec3(up[0], up[1], up[2]));
}

}  // extern "C"
