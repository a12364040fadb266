// vecmath.h -- float 3-vectors for the host-side scene build.
//
// Covers what the reference keeps in vec3d.h:10-111 and model.h:144-184.  The scene arrays must come
// out bit-identical to the reference's, so the operations that differ from the obvious form are kept:
// division multiplies by the rounded reciprocal (vec3d.h:77-85) and length is sqrt((xx + yy) + zz)
// (vec3d.h:102-105).  Build with -ffp-contract=off.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>

#include "rt_hip.h"

namespace rt {

struct Vec3 {
    float x, y, z;

    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}

    float &operator[](int axis) { return axis == 0 ? x : axis == 1 ? y : z; }
    float operator[](int axis) const { return axis == 0 ? x : axis == 1 ? y : z; }
};

inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 scale(const Vec3 &v, float s) { return Vec3(s * v.x, s * v.y, s * v.z); }
inline Vec3 divide(const Vec3 &v, float d) { return scale(v, 1 / d); }
inline float dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(const Vec3 &v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline Vec3 normalized(const Vec3 &v) { return divide(v, length(v)); }

// Axis-aligned bounds, empty when constructed (model.h:146-171).
struct Bounds {
    Vec3 lo, hi;

    Bounds()
    {
        const float inf = std::numeric_limits<float>::infinity();
        lo = Vec3(inf, inf, inf);
        hi = Vec3(-inf, -inf, -inf);
    }

    void add(const Vec3 &p)
    {
        lo = Vec3(std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z));
        hi = Vec3(std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z));
    }

    void add(const Bounds &b)
    {
        lo = Vec3(std::min(lo.x, b.lo.x), std::min(lo.y, b.lo.y), std::min(lo.z, b.lo.z));
        hi = Vec3(std::max(hi.x, b.hi.x), std::max(hi.y, b.hi.y), std::max(hi.z, b.hi.z));
    }
};

// Instance transform applied to a point: rows (x, y, z) of an rt_matrix, translation in lane 3 (model.h:173-179).
inline Vec3 transform_point(const rt_matrix &m, const Vec3 &v)
{
    const float x = m.x[0] * v.x + m.x[1] * v.y + m.x[2] * v.z + m.x[3];
    const float y = m.y[0] * v.x + m.y[1] * v.y + m.y[2] * v.z + m.y[3];
    const float z = m.z[0] * v.x + m.z[1] * v.y + m.z[2] * v.z + m.z[3];
    return Vec3(x, y, z);
}

inline void store3(float dst[], const Vec3 &v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

}  // namespace rt
