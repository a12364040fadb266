// ref_model_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the REFERENCE's own host-side scene builder.  It is compiled together
// with /root/reference/model.cpp (sources stay where they lie; see oracle/Makefile, target
// _ref/libref_model.so) and only calls the reference's classes: Model::load / subdivide / reserve /
// fill / put and ResourceManager (reference model.h:19-140, 245-286; model.cpp:25-222), in the order
// RayTracer::create_buffers() uses them (reference main.cpp:285-327).  The tests use its output to
// pin raytracer_amd/host/model.cpp byte for byte.  Nothing in the product may link or load this.

#include "model.h"  // the reference's header, found through -I/root/reference

#include <cstddef>
#include <cstdio>
#include <cstring>
#include <vector>

struct refm_scene {
    ResourceManager mngr;
    std::vector<Model *> models;
    cl_uint root_id, green_id, red_id;

    ~refm_scene()
    {
        for (Model *m : models) delete m;
    }
};

extern "C" {

// files: n_files PLY paths; mat: n_obj instance matrices in the reference's Matrix layout.
refm_scene *refm_build(const char *const *files, int n_files, const Matrix *mat, int n_obj, int tri_threshold,
                       int aabb_threshold)
{
    refm_scene *s = new refm_scene;
    s->mngr.reserve_groups(6);
    s->mngr.reserve_aabbs(n_obj);
    for (int k = 0; k < n_files; k++) {
        Model *m = new Model;
        s->models.push_back(m);
        if (!m->load(files[k])) {
            delete s;
            return 0;
        }
        m->subdivide(tri_threshold, aabb_threshold);
        m->reserve(s->mngr);
    }
    s->mngr.alloc();
    s->mngr.get_groups(3);
    s->green_id = make_group_id(s->mngr.get_groups(1), tr_none, sh_material);
    s->red_id = make_group_id(s->mngr.get_groups(1), tr_none, sh_material);
    s->root_id = make_group_id(s->mngr.get_groups(1), tr_identity, sh_aabb);

    Group *grp = s->mngr.group(s->root_id & GROUP_ID_MASK);
    AABB *aabb = s->mngr.aabb(grp->aabb.aabb_offs = s->mngr.get_aabbs(n_obj));
    grp->aabb.aabb_count = n_obj;
    grp->aabb.flags = f_local0;

    for (int k = 0; k < n_files; k++) s->models[k]->fill(s->mngr, k & 1 ? s->red_id : s->green_id);
    for (int i = 0; i < n_obj; i++) s->models[i % n_files]->put(aabb[i], mat[i], i);
    if (!s->mngr.full()) {
        delete s;
        return 0;
    }
    return s;
}

void refm_free(refm_scene *s)
{
    delete s;
}

size_t refm_group_count(refm_scene *s) { return s->mngr.group_count(); }
size_t refm_aabb_count(refm_scene *s) { return s->mngr.aabb_count(); }
size_t refm_vertex_count(refm_scene *s) { return s->mngr.vertex_count(); }
size_t refm_triangle_count(refm_scene *s) { return s->mngr.triangle_count(); }
const void *refm_groups(refm_scene *s) { return s->mngr.group(0); }
const void *refm_aabbs(refm_scene *s) { return s->mngr.aabb(0); }
const void *refm_vertices(refm_scene *s) { return s->mngr.vertex(0); }
const void *refm_triangles(refm_scene *s) { return s->mngr.triangle(0); }
unsigned refm_root_id(refm_scene *s) { return s->root_id; }

// Camera helper of the reference (model.cpp:11-21), for pinning rt::set_camera.
void refm_set_camera(Camera *cam, size_t width, size_t height, float tan_fov, const float pos[3], const float view[3],
                     const float up[3])
{
    set_camera(*cam, width, height, tan_fov, Vector(pos[0], pos[1], pos[2]), Vector(view[0], view[1], view[2]),
               Vector(up[0], up[1], up[2]));
}

// Sizes, member offsets and constants of the reference's shared data layout (ray-tracer.h, compiled as C++ with the
// image's CL headers: cl_float3 is a 16-byte type), as "name=value;" pairs, for pinning include/rt_hip.h.
const char *refm_layout()
{
    static char text[4096];
    int at = 0;
#define PUT(name, value) at += std::snprintf(text + at, sizeof(text) - at, "%s=%zu;", name, size_t(value))
#define SIZE(T) PUT("sizeof " #T, sizeof(T))
#define OFFS(T, m) PUT(#T "." #m, offsetof(T, m))
    SIZE(MatShader); SIZE(AABB); SIZE(AABBShader); SIZE(Vertex); SIZE(MeshShader); SIZE(Matrix); SIZE(Group);
    SIZE(Camera); SIZE(GlobalData); SIZE(GroupData); SIZE(Ray); SIZE(RayHit); SIZE(RayQueue);
    OFFS(AABB, min); OFFS(AABB, max); OFFS(AABB, group_id); OFFS(AABB, local_id);
    OFFS(AABBShader, aabb_offs); OFFS(AABBShader, aabb_count); OFFS(AABBShader, flags);
    OFFS(Vertex, pos); OFFS(Vertex, norm);
    OFFS(MeshShader, vtx_offs); OFFS(MeshShader, tri_offs); OFFS(MeshShader, tri_count); OFFS(MeshShader, material_id);
    OFFS(Matrix, x); OFFS(Matrix, y); OFFS(Matrix, z);
    OFFS(Camera, eye); OFFS(Camera, top_left); OFFS(Camera, dx); OFFS(Camera, dy); OFFS(Camera, width); OFFS(Camera, height);
    OFFS(Camera, root_group); OFFS(Camera, root_local);
    OFFS(GlobalData, pixel_offset); OFFS(GlobalData, pixel_count); OFFS(GlobalData, group_count); OFFS(GlobalData, old_count);
    OFFS(GlobalData, ray_count); OFFS(GlobalData, cam);
    OFFS(GroupData, base); OFFS(GroupData, count); OFFS(GroupData, offset);
    OFFS(Ray, start_min); OFFS(Ray, min); OFFS(Ray, dir_max); OFFS(Ray, max);
    OFFS(RayHit, pos); OFFS(RayHit, group_id); OFFS(RayHit, local_id);
    OFFS(RayQueue, weight); OFFS(RayQueue, pixel); OFFS(RayQueue, type); OFFS(RayQueue, material_id); OFFS(RayQueue, queue_len);
    OFFS(RayQueue, ray); OFFS(RayQueue, norm); OFFS(RayQueue, root); OFFS(RayQueue, orig); OFFS(RayQueue, queue);
    PUT("GROUP_ID_MASK", GROUP_ID_MASK); PUT("GROUP_TR_SHIFT", GROUP_TR_SHIFT); PUT("GROUP_TR_MASK", GROUP_TR_MASK);
    PUT("GROUP_SH_SHIFT", GROUP_SH_SHIFT); PUT("GROUP_SH_MASK", GROUP_SH_MASK);
    PUT("tr_none", tr_none); PUT("tr_identity", tr_identity); PUT("tr_ortho", tr_ortho); PUT("tr_affine", tr_affine);
    PUT("sh_spawn", sh_spawn); PUT("sh_sky", sh_sky); PUT("sh_light", sh_light); PUT("sh_material", sh_material);
    PUT("sh_aabb", sh_aabb); PUT("sh_mesh", sh_mesh);
    PUT("spawn_group", spawn_group); PUT("sky_group", sky_group); PUT("light_group", light_group);
    PUT("f_local0", f_local0); PUT("f_local1", f_local1); PUT("rt_primary", rt_primary); PUT("rt_shadow", rt_shadow);
    PUT("MAX_HITS", MAX_HITS); PUT("MAX_QUEUE_LEN", MAX_QUEUE_LEN); PUT("RADIX_SHIFT", RADIX_SHIFT);
#undef OFFS
#undef SIZE
#undef PUT
    return text;
}

}  // extern "C"
