// model.h -- host-side scene build: triangle mesh -> wide bounding-box hierarchy -> flat arrays.
//
// Same public surface as the reference's ResourceManager (model.h:19-140), make_group_id
// (model.h:13-16), set_camera (model.h:186-187) and Model (model.h:245-286), re-implemented on
// index-based storage.  The flat arrays it emits are byte-compatible with the reference's
// (ray-tracer.h), which is what rt_upload_scene() consumes.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "rt_hip.h"
#include "vecmath.h"

namespace rt {

// Packs table index, transform type and shader type into a 32-bit group id (model.h:13-16).
inline uint32_t make_group_id(size_t index, int transform, int shader)
{
    return uint32_t(index) | uint32_t(transform) << RT_GROUP_TR_SHIFT | uint32_t(shader) << RT_GROUP_SH_SHIFT;
}

// Two-phase arena for the four flat scene arrays: reserve_*() while sizes are being counted, alloc()
// once, then get_*() hands out consecutive ranges (model.h:19-140).  Storage is zero-initialised.
class ResourceManager {
public:
    void reserve_groups(size_t n) { want_[0] += n; }
    void reserve_aabbs(size_t n) { want_[1] += n; }
    void reserve_vertices(size_t n) { want_[2] += n; }
    void reserve_triangles(size_t n) { want_[3] += n; }

    void alloc();
    bool allocated() const { return allocated_; }
    bool full() const;

    uint32_t get_groups(size_t n) { return take(0, n); }
    uint32_t get_aabbs(size_t n) { return take(1, n); }
    uint32_t get_vertices(size_t n) { return take(2, n); }
    uint32_t get_triangles(size_t n) { return take(3, n); }

    rt_group *group(size_t index) { return &groups_.at(index); }
    rt_aabb *aabb(size_t index) { return &aabbs_.at(index); }
    rt_vertex *vertex(size_t index) { return &vertices_.at(index); }
    uint32_t *triangle(size_t index) { return &triangles_.at(index); }

    size_t group_count() const { return want_[0]; }
    size_t aabb_count() const { return want_[1]; }
    size_t vertex_count() const { return want_[2]; }
    size_t triangle_count() const { return want_[3]; }

    std::vector<rt_group> &groups() { return groups_; }
    std::vector<rt_aabb> &aabbs() { return aabbs_; }
    std::vector<rt_vertex> &vertices() { return vertices_; }
    std::vector<uint32_t> &triangles() { return triangles_; }

private:
    uint32_t take(int which, size_t n);

    size_t want_[4] = {0, 0, 0, 0};
    size_t used_[4] = {0, 0, 0, 0};
    bool allocated_ = false;
    std::vector<rt_group> groups_;
    std::vector<rt_aabb> aabbs_;
    std::vector<rt_vertex> vertices_;
    std::vector<uint32_t> triangles_;
};

// Look-at camera (model.cpp:11-21).  The reference never calls it; kept for non-default views.
void set_camera(rt_camera &cam, size_t width, size_t height, float tan_fov, const Vec3 &pos, const Vec3 &view,
                const Vec3 &up = Vec3(0, 0, 1));

class Model {
public:
    // Reads an ASCII PLY file of the dialect the reference accepts (model.cpp:143-160): one comment
    // line, float x y z plus up to four more float vertex properties, triangular faces.
    bool load(const char *file);
    bool loaded_from_cache() const { return from_cache_; }  // load() found the file's parsed arrays in the binary cache
    // Takes a mesh that is already in memory.
    bool assign(const std::vector<Vec3> &positions, const std::vector<std::array<uint32_t, 3>> &faces);

    // Median split until a node holds fewer than tri_threshold triangles; a node becomes a
    // bounding-box group when at least aabb_threshold leaves hang below it (model.cpp:25-47).
    void subdivide(size_t tri_threshold, size_t aabb_threshold);
    void reserve(ResourceManager &mngr);
    void fill(ResourceManager &mngr, uint32_t material_id);
    // World-space box of one instance, from every vertex (model.cpp:216-222).
    void put(rt_aabb &aabb, const rt_matrix &mat, uint32_t local_id) const;

    size_t vertex_count() const { return pos_.size(); }
    size_t triangle_count() const { return faces_.size(); }
    uint32_t group_id() const { return group_id_; }

private:
    bool load_file(const char *file);  // load() without its guard against running out of memory
    bool from_cache_ = false;
    struct Face {
        uint32_t v[3];
        Vec3 center;
    };
    struct Node {
        uint32_t first = 0, count = 0;  // range of order_
        int child[2] = {-1, -1};
        uint32_t aabb_count = 0, vtx_count = 0;
        Bounds box;
    };

    void prepare();
    void split_node(int node, int first_child);
    size_t count_boxes(int node, size_t aabb_threshold, bool is_root);
    void reserve_node(int node, ResourceManager &mngr);
    uint32_t fill_node(int node, ResourceManager &mngr, uint32_t material_id, uint32_t *aabb_cursor);

    std::vector<Vec3> pos_, norm_;
    std::vector<int> slot_;  // leaf-local vertex number while a leaf is being emitted, else -1
    std::vector<Face> faces_;
    std::vector<uint32_t> order_;  // triangle permutation the splits sort in place
    std::vector<Node> nodes_;
    uint32_t group_id_ = 0;
};

}  // namespace rt
