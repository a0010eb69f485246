// rt_vec.h — binary32 vector helpers shared by host scene construction and device kernels.
//
// Every helper fixes ONE evaluation order — the order GLM 0.9.9.9 (vendored by the reference
// under RayTracing/Depend/include/glm/, "glm/" below) uses for the same operation — because that
// order is the numerical contract of the render loop.  The library is compiled
// with -ffp-contract=off, so nothing here fuses; kernels that want FMAs call fmaf explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>

#define RT_HD __host__ __device__ __forceinline__

namespace rt {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { vec4 c[4]; };   // column-major, c[col] — glm::mat4 storage order

RT_HD vec3 make3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD vec3 operator-(vec3 a, vec3 b) { return make3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD vec3 operator+(vec3 a, vec3 b) { return make3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD vec3 operator*(vec3 a, float s) { return make3(a.x * s, a.y * s, a.z * s); }

// glm/detail/func_geometric.inl:48-55 — tmp = a*b; (tmp.x + tmp.y) + tmp.z
RT_HD float dot(vec3 a, vec3 b) {
	float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return (tx + ty) + tz;
}

// glm/detail/func_geometric.inl:74-77
RT_HD vec3 cross(vec3 x, vec3 y) {
	return make3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}

// glm/detail/func_geometric.inl:88 with inversesqrt = 1/sqrt (func_exponential.inl:136-139).
// Both the division and the square root are correctly rounded (hipcc default for HIP; never
// build this library with -ffast-math).
RT_HD vec3 normalize(vec3 v) {
	float inv = 1.0f / sqrtf(dot(v, v));
	return v * inv;
}

// glm/detail/func_geometric.inl:108 — I - (N * dot(N, I)) * 2
RT_HD vec3 reflect(vec3 I, vec3 N) {
	float d = dot(N, I);
	return I - (N * d) * 2.0f;
}

// Source.cpp:443-453 newNormal — note dot == 0 flips
RT_HD vec3 face_normal(vec3 dir, vec3 n) {
	return dot(dir, n) < 0.0f ? n : n * -1.0f;
}

RT_HD vec4 make4(float x, float y, float z, float w) { vec4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_HD vec4 operator*(vec4 a, float s) { return make4(a.x * s, a.y * s, a.z * s, a.w * s); }
RT_HD vec4 operator+(vec4 a, vec4 b) { return make4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// glm/detail/type_mat4x4.inl:561-572 — (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
RT_HD vec4 mul(const mat4& m, vec4 v) {
	vec4 add0 = m.c[0] * v.x + m.c[1] * v.y;
	vec4 add1 = m.c[2] * v.z + m.c[3] * v.w;
	return add0 + add1;
}

} // namespace rt
