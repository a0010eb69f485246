// rt_lbvh_trav.inl — LBVH traversal (device side), included by rt_kernels.hip.
//
// castRay (Source.cpp:277-405) over the tree built by rt_lbvh.hip instead of over every triangle.
// The closest hit of the reference's loop is the accepted triangle with the smallest t and, among
// equal t, the smallest index (its strict `t < closest.distance` keeps the first one met in index
// order); acceptance of a triangle does not depend on the others.  Any visiting order therefore
// gives the same result as long as (1) every triangle that the reference would accept with
// t <= current best is visited, and (2) ties are resolved towards the smaller index.  Boxes
// (padded, see rt_lbvh.hip) only skip subtrees that cannot contain such a triangle; the triangle
// test at a leaf is the reference's arithmetic, unfused, in its order.
//
// Per-lane traversal stack: RT_STACK_LDS entries in LDS (entry k of lane l at [k * blockDim + l]:
// conflict-free), the rest — a Karras tree over 62 key bits is at most 62 deep — in a private
// array that is only touched by unusually deep paths.
#pragma once

#define RT_STACK_LDS 32
#define RT_STACK_MAX 64

namespace rt {

struct Slab {
	vec3 inv, oi;
};

__device__ __forceinline__ void slab2(const float4 q0, const float4 q1, const float4 q2, const Slab& s, float lim_hi,
                                      float& tminL, float& tmaxL, float& tminR, float& tmaxR) {
	// left box lo = (q0.x q0.y q0.z) hi = (q0.w q1.x q1.y); right box lo = (q1.z q1.w q2.x) hi = (q2.y q2.z q2.w)
	const float lx0 = __builtin_fmaf(q0.x, s.inv.x, s.oi.x), lx1 = __builtin_fmaf(q0.w, s.inv.x, s.oi.x);
	const float ly0 = __builtin_fmaf(q0.y, s.inv.y, s.oi.y), ly1 = __builtin_fmaf(q1.x, s.inv.y, s.oi.y);
	const float lz0 = __builtin_fmaf(q0.z, s.inv.z, s.oi.z), lz1 = __builtin_fmaf(q1.y, s.inv.z, s.oi.z);
	const float rx0 = __builtin_fmaf(q1.z, s.inv.x, s.oi.x), rx1 = __builtin_fmaf(q2.y, s.inv.x, s.oi.x);
	const float ry0 = __builtin_fmaf(q1.w, s.inv.y, s.oi.y), ry1 = __builtin_fmaf(q2.z, s.inv.y, s.oi.y);
	const float rz0 = __builtin_fmaf(q2.x, s.inv.z, s.oi.z), rz1 = __builtin_fmaf(q2.w, s.inv.z, s.oi.z);
	tminL = fmaxf(fmaxf(fminf(lx0, lx1), fminf(ly0, ly1)), fmaxf(fminf(lz0, lz1), 0.0f));
	tmaxL = fminf(fminf(fmaxf(lx0, lx1), fmaxf(ly0, ly1)), fminf(fmaxf(lz0, lz1), lim_hi));
	tminR = fmaxf(fmaxf(fminf(rx0, rx1), fminf(ry0, ry1)), fmaxf(fminf(rz0, rz1), 0.0f));
	tmaxR = fminf(fminf(fmaxf(rx0, rx1), fmaxf(ry0, ry1)), fminf(fmaxf(rz0, rz1), lim_hi));
}

template <bool COUNT>
__device__ __forceinline__ Hit cast_ray_lbvh(const DevScene& sc, int* lds_stack, vec3 o, vec3 d, int skip, bool live,
                                             bool shadow, float dist, LaneCounters& lc) {
	Hit h;
	h.t = 1e9f;                                                          // Source.cpp:278
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	const bool cull = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;   // false for NaN too
	float lim = cull ? fminf(exit_bound(sc, o, d), 1e9f) : 1e9f;
	if (shadow) lim = fminf(lim, __uint_as_float(__float_as_uint(dist) + 1u));          // Source.cpp:476
	lim = fminf(lim, 1e9f);
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	Slab s;
	{
		const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
		s.inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
		s.oi = make3(-(o.x * s.inv.x), -(o.y * s.inv.y), -(o.z * s.inv.z));
	}
	const int SENT = 0x7fffffff;
	const int bd = blockDim.x, tid = threadIdx.x;
	int ovf[RT_STACK_MAX - RT_STACK_LDS];
	int sp = 0;
	int node = live ? sc.bvh_root : SENT;
	const float4* __restrict__ planes = sc.planes;
	const int stride = sc.stride;

	auto pop = [&]() -> int {
		if (sp == 0) return SENT;
		sp--;
		return sp < RT_STACK_LDS ? lds_stack[sp * bd + tid] : ovf[sp - RT_STACK_LDS];
	};

	while (node != SENT) {
		if (node >= 0) {
			const float4* nd = sc.bvh_nodes + (size_t)node * 4;
			const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
			float tminL, tmaxL, tminR, tmaxR;
			slab2(q0, q1, q2, s, lim_hi, tminL, tmaxL, tminR, tmaxR);
			const bool hitL = !(tminL > tmaxL) || !cull;
			const bool hitR = !(tminR > tmaxR) || !cull;
			if (COUNT) lc.box += 2;
			const int l = __float_as_int(q3.x), r = __float_as_int(q3.y);
			if (hitL && hitR) {
				const bool swap = tminR < tminL;
				const int far = swap ? l : r;
				node = swap ? r : l;
				if (sp < RT_STACK_LDS) lds_stack[sp * bd + tid] = far;
				else if (sp < RT_STACK_MAX) ovf[sp - RT_STACK_LDS] = far;
				sp++;
			} else if (hitL) {
				node = l;
			} else if (hitR) {
				node = r;
			} else {
				node = pop();
			}
		} else {
			const int i = ~node;
			const float4 N4 = planes[PL_N * stride + i];
			const float4 P14 = planes[PL_P1 * stride + i];
			const vec3 nn = xyz(N4), p1 = xyz(P14);
			const float t = dot(p1 - o, nn) / dot(d, nn);                                 // Source.cpp:383
			const bool visit = (i != skip);                                               // Source.cpp:302
			const bool better = (t < lim) || (t == h.t && i < h.idx);                     // Source.cpp:391 + index order
			const bool cand = visit && (t > RT_EPS_BELOW) && better;                      // Source.cpp:385, 391
			if (COUNT) { lc.plane += visit; lc.point += visit && !(t < 0.0f); lc.edge += cand; }
			bool found = false;
			if (cand) {
				const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);        // Source.cpp:389
				const float4 E0 = planes[PL_E0 * stride + i];
				if (dot(nn, cross(xyz(E0), I - p1)) > 0.0f) {                             // Source.cpp:392
					if (COUNT) lc.edge++;
					const float4 P24 = planes[PL_P2 * stride + i];
					const float4 E1 = planes[PL_E1 * stride + i];
					if (dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f) {                   // Source.cpp:393
						if (COUNT) lc.edge++;
						const float4 P34 = planes[PL_P3 * stride + i];
						const float4 E2 = planes[PL_E2 * stride + i];
						if (dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f) {               // Source.cpp:394
							h.t = t; h.idx = i; h.p = I;                                  // Source.cpp:396
							lim = t;
							lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
							found = true;
						}
					}
				}
			}
			node = (found && shadow) ? SENT : pop();      // a shadow ray only asks whether a hit with t <= dist exists
		}
	}
	return h;
}

} // namespace rt
