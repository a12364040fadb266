// scene.cpp -- the benchmark scene: a field of instanced models, two materials, one camera.
//
// Follows the reference's RayTracer::create_buffers() (main.cpp:268-343) step by step so that the
// arrays match it byte for byte for the same model files, then generalises it (any model list, any
// resolution, a row band per GPU, an optional regular grid of instances).

#include "scene.h"
#include "parallel.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace rt {

namespace {

// Instance transforms of the reference scene: a rotation about world z that maps model y to world z,
// and a translation inside [-2,2] x [0,4] x [-1,1]; four glibc random() draws per instance in the
// order angle, tx, ty, tz (main.cpp:274-284).  srandom(1) is glibc's state before any seeding.
void random_field(std::vector<rt_matrix> &mat)
{
    srandom(1);
    for (rt_matrix &m : mat) {
        std::memset(&m, 0, sizeof(m));
        const double alpha = 2 * 3.14159265359 * random() / RAND_MAX;
        m.y[2] = float(std::cos(alpha));
        m.x[0] = m.y[2];
        m.y[0] = float(std::sin(alpha));
        m.x[2] = -m.y[0];
        m.z[1] = 1;
        m.x[3] = float(4.0 * random() / RAND_MAX - 2);
        m.y[3] = float(4.0 * random() / RAND_MAX);
        m.z[3] = float(2.0 * random() / RAND_MAX - 1);
    }
}

// k*k*k upright instances on a regular lattice in the same volume.
void grid_field(std::vector<rt_matrix> &mat, uint32_t k)
{
    size_t i = 0;
    for (uint32_t iz = 0; iz < k; iz++)
        for (uint32_t iy = 0; iy < k; iy++)
            for (uint32_t ix = 0; ix < k; ix++, i++) {
                rt_matrix &m = mat[i];
                std::memset(&m, 0, sizeof(m));
                m.x[0] = 1;
                m.y[2] = 1;
                m.z[1] = 1;
                m.x[3] = float(-2 + 4.0 * (ix + 0.5) / k);
                m.y[3] = float(0.4 + 3.6 * (iy + 0.5) / k);
                m.z[3] = float(-1 + 2.0 * (iz + 0.5) / k - 0.1);
            }
}

}  // namespace

bool build_scene(const SceneSpec &spec, Scene &scene, std::string &error)
{
    if (spec.model_files.empty()) {
        error = "no model files given";
        return false;
    }
    const uint32_t n_obj = spec.grid ? spec.grid * spec.grid * spec.grid : spec.n_obj;
    if (!n_obj || !spec.width || !spec.height) {
        error = "empty scene or image";
        return false;
    }
    const uint32_t rows = spec.band_rows ? spec.band_rows : spec.height;
    if (uint64_t(spec.band_first) + rows > spec.height) {
        error = "band exceeds the image";
        return false;
    }

    std::vector<rt_matrix> mat(n_obj);
    if (spec.grid) grid_field(mat, spec.grid);
    else random_field(mat);

    ResourceManager mngr;
    mngr.reserve_groups(6);  // spawn, sky, light, two materials, scene root (main.cpp:285, 310-313)
    mngr.reserve_aabbs(n_obj);

    const auto seconds_since = [](std::chrono::steady_clock::time_point from) {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - from).count();
    };
    const auto build_start = std::chrono::steady_clock::now();
    std::vector<std::unique_ptr<Model>> models;
    for (const std::string &file : spec.model_files) {
        models.emplace_back(new Model);
        const auto load_start = std::chrono::steady_clock::now();
        if (!models.back()->load(file.c_str())) {
            error = "cannot load model " + file;
            return false;
        }
        scene.ingest_seconds += seconds_since(load_start);
        scene.models_from_cache += models.back()->loaded_from_cache() ? 1 : 0;
        try {
            models.back()->subdivide(spec.tri_threshold, spec.aabb_threshold);
            models.back()->reserve(mngr);
        } catch (const std::exception &e) {
            error = file + ": " + e.what();
            return false;
        }
    }

    mngr.alloc();
    mngr.get_groups(3);  // predefined: spawn, sky, light
    const uint32_t green_id = make_group_id(mngr.get_groups(1), RT_TR_NONE, RT_SH_MATERIAL);
    const uint32_t red_id = make_group_id(mngr.get_groups(1), RT_TR_NONE, RT_SH_MATERIAL);
    const uint32_t root_id = make_group_id(mngr.get_groups(1), RT_TR_IDENTITY, RT_SH_AABB);

    rt_group *grp = mngr.group(root_id & RT_GROUP_ID_MASK);
    grp->aabb.aabb_offs = mngr.get_aabbs(n_obj);
    grp->aabb.aabb_count = n_obj;
    grp->aabb.flags = RT_F_LOCAL0;
    rt_aabb *instance_box = mngr.aabb(grp->aabb.aabb_offs);

    const float green[4] = {0.2f, 0.9f, 0.2f, 0.1f}, red[4] = {0.9f, 0.2f, 0.2f, 0.1f};  // main.cpp:319-323
    std::memcpy(mngr.group(green_id & RT_GROUP_ID_MASK)->material.color, green, sizeof(green));
    std::memcpy(mngr.group(red_id & RT_GROUP_ID_MASK)->material.color, red, sizeof(red));

    for (size_t k = 0; k < models.size(); k++) models[k]->fill(mngr, k & 1 ? red_id : green_id);
    scene.resident_triangles = 0;
    for (uint32_t i = 0; i < n_obj; i++) scene.resident_triangles += models[i % models.size()]->triangle_count();
    // every instance's world-space box comes from all of its model's vertices (model.cpp:216-222): one item each
    parallel_for(n_obj, [&](size_t i) { models[i % models.size()]->put(instance_box[i], mat[i], uint32_t(i)); });
    if (!mngr.full()) {
        error = "internal: reserved and emitted sizes differ";
        return false;
    }

    scene.pool_size = uint32_t(align_up(spec.ray_count, RT_SORT_TILE));
    scene.row_offset = spec.band_first;

    rt_global_data &data = scene.data;
    std::memset(&data, 0, sizeof(data));
    data.ray_count = scene.pool_size;
    data.group_count = uint32_t(align_up(mngr.group_count() + 1, 512));  // main.cpp:331
    // Camera of main.cpp:334-339 (eye below the field looking along +y, image plane y = 1),
    // stretched horizontally for non-square images so that pixels stay square.
    const float eye[3] = {0, float(-0.3), 0};
    const float top_left[3] = {float(-0.5 * spec.width / spec.height), 1, float(-0.5)};
    const float dx[3] = {float(1.0 / spec.height), 0, 0}, dy[3] = {0, 0, float(1.0 / spec.height)};
    std::memcpy(data.cam.eye, eye, sizeof(eye));
    std::memcpy(data.cam.top_left, top_left, sizeof(top_left));
    std::memcpy(data.cam.dx, dx, sizeof(dx));
    std::memcpy(data.cam.dy, dy, sizeof(dy));
    data.cam.width = spec.width;
    data.cam.height = rows;
    data.cam.root_group = root_id;
    data.cam.root_local = 0;

    scene.groups.swap(mngr.groups());
    scene.matrices.swap(mat);
    scene.aabbs.swap(mngr.aabbs());
    scene.vertices.swap(mngr.vertices());
    scene.triangles.swap(mngr.triangles());
    scene.build_seconds = seconds_since(build_start) - scene.ingest_seconds;
    return true;
}

// A closed genus-0 surface with layered bumps, sized and placed like the Stanford dragon scan
// (about 0.2 x 0.14 x 0.09 units, resting above y = 0.05).  `rings` latitude rings of `around`
// vertices plus two poles: around*rings + 2 vertices and 2*around*rings triangles.
bool write_standin_ply(const char *path, uint32_t around, uint32_t rings)
{
    if (around < 3 || rings < 1) return false;
    FILE *out = std::fopen(path, "w");
    if (!out) return false;
    const size_t n_vtx = size_t(around) * rings + 2, n_tri = size_t(2) * around * rings;
    std::fprintf(out,
                 "ply\nformat ascii 1.0\ncomment synthetic stand-in for dragon_vrip\n"
                 "element vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %zu\nproperty list uchar int vertex_indices\nend_header\n",
                 n_vtx, n_tri);

    const double pi = 3.14159265358979323846;
    auto emit = [&](double theta, double phi) {
        // theta in (0, pi) from pole to pole, phi around.  Radius with three octaves of bumps that
        // vanish at the poles, then an anisotropic squash and a bend so leaves are not convex shells.
        const double s = std::sin(theta), c = std::cos(theta);
        const double bumps = 0.22 * std::sin(5 * phi) * std::sin(4 * theta) + 0.10 * std::sin(13 * phi + 2 * theta) * s +
                             0.05 * std::sin(31 * phi) * std::sin(29 * theta) * s;
        const double r = 1 + s * bumps;
        double x = r * s * std::cos(phi), y = r * c, z = r * s * std::sin(phi);
        x += 0.35 * y * y;  // bend
        std::fprintf(out, "%.6g %.6g %.6g\n", 0.095 * x - 0.01, 0.068 * y + 0.125, 0.042 * z - 0.005);
    };
    emit(0, 0);
    for (uint32_t j = 1; j <= rings; j++)
        for (uint32_t i = 0; i < around; i++) emit(pi * j / (rings + 1), 2 * pi * i / around);
    emit(pi, 0);

    auto ring_vertex = [&](uint32_t ring, uint32_t i) { return size_t(1) + size_t(ring) * around + i % around; };
    for (uint32_t i = 0; i < around; i++)  // cap at the first pole
        std::fprintf(out, "3 0 %zu %zu\n", ring_vertex(0, i + 1), ring_vertex(0, i));
    for (uint32_t j = 0; j + 1 < rings; j++)
        for (uint32_t i = 0; i < around; i++) {
            const size_t a = ring_vertex(j, i), b = ring_vertex(j, i + 1), c = ring_vertex(j + 1, i), d = ring_vertex(j + 1, i + 1);
            std::fprintf(out, "3 %zu %zu %zu\n3 %zu %zu %zu\n", a, b, c, b, d, c);
        }
    for (uint32_t i = 0; i < around; i++)  // cap at the second pole
        std::fprintf(out, "3 %zu %zu %zu\n", n_vtx - 1, ring_vertex(rings - 1, i), ring_vertex(rings - 1, i + 1));
    return std::fclose(out) == 0;
}

}  // namespace rt
