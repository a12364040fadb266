// scene.h -- assembly of the benchmark scenes into the flat arrays rt_upload_scene() takes.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "model.h"

namespace rt {

struct SceneSpec {
    std::vector<std::string> model_files;  // instance i uses model_files[i % size] (main.cpp:326 with two models)
    uint32_t n_obj = 256;                  // instances (main.cpp:272)
    uint32_t tri_threshold = 128;          // main.cpp:270
    uint32_t aabb_threshold = 128;
    uint32_t width = 1024, height = 1024;  // full image the camera spans (main.cpp:510)
    uint32_t ray_count = 1024 * 1024;      // requested pool size (main.cpp:516)
    uint32_t band_first = 0, band_rows = 0;  // rows [band_first, band_first + band_rows) for this GPU; 0 rows = all
    // Placement of the instances: 0 = the reference's random field from glibc random() with its
    // default seed (main.cpp:274-284); k > 0 = a regular k*k*k grid of upright instances.
    uint32_t grid = 0;
};

struct Scene {
    rt_global_data data;
    std::vector<rt_group> groups;
    std::vector<rt_matrix> matrices;
    std::vector<rt_aabb> aabbs;
    std::vector<rt_vertex> vertices;
    std::vector<uint32_t> triangles;
    uint32_t pool_size = 0;    // ray slots after alignment (main.cpp:188)
    uint32_t row_offset = 0;   // first row of the band (rt_config.row_offset)
    uint64_t resident_triangles = 0;  // sum over instances of their model's triangle count
    double ingest_seconds = 0;        // reading the model files (text parse, or the binary cache of Model::load)
    double build_seconds = 0;         // hierarchy build and scene assembly
    uint32_t models_from_cache = 0;
};

inline size_t align_up(size_t value, size_t unit)
{
    return (value + unit - 1) / unit * unit;
}

// Builds the scene of main.cpp:268-343 for the given models.  Returns false and fills `error`.
bool build_scene(const SceneSpec &spec, Scene &scene, std::string &error);

// Writes an ASCII PLY stand-in for dragon_vrip.ply (which cannot be shipped): a closed, bumpy,
// dragon-sized surface with 2*around*(rings) triangles, in the dialect Model::load reads.
bool write_standin_ply(const char *path, uint32_t around, uint32_t rings);

}  // namespace rt
