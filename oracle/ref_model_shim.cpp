// ref_model_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" driver around the REFERENCE's own host-side scene builder.  It is compiled together
// with /root/reference/model.cpp (sources stay where they lie; see oracle/Makefile, target
// _ref/libref_model.so) and only calls the reference's classes: Model::load / subdivide / reserve /
// fill / put and ResourceManager (reference model.h:19-140, 245-286; model.cpp:25-222), in the order
// RayTracer::create_buffers() uses them (reference main.cpp:285-327).  The tests use its output to
// pin raytracer_amd/host/model.cpp byte for byte.  Nothing in the product may link or load this.

#include "model.h"  // the reference's header, found through -I/root/reference

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

}  // extern "C"
