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
// The tree is walked in its 4-wide form (rt_lbvh.hip k_collapse4): one 128-byte fetch, four slab
// tests, children visited nearest first.  A leaf test reads the triangle's AoS record in one burst.
//
// Per-lane traversal stack: RT_STACK_LDS entries in LDS (entry k of lane l at [k * blockDim + l]:
// conflict-free), the rest in a private array that is only touched by unusually deep paths.  The
// nearest-first walk of a 4-wide tree keeps up to 3 entries per level, so the stack needs
// 3 * (depth of the 4-wide tree) entries in the worst case; lbvh_build measures that depth at
// commit (rt_lbvh.hip tree_depths) and a tree with 3 * depth > RT_STACK_MAX is never handed to this
// traversal (DevScene::per_lane_ok: the launchers walk it by packets, whose stack holds one entry
// per binary level, at most 63).  The guards in push_any / pop_any keep the indices inside the
// arrays regardless.
#pragma once

#ifndef RT_STACK_LDS
#define RT_STACK_LDS 16      /* measured: 12-16 beats 24 and 32 (occupancy); deeper paths spill to a private array */
#endif
#define RT_STACK_MAX RT_STACK_LIMIT   /* include/rt355.h: 64 */
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 1        /* lanes that must be waiting on a parked leaf before the wave makes a leaf trip */
#endif
#define RT_STACK_PITCH 64    /* lanes per workgroup the LDS stack is laid out for: every kernel that walks per lane is launched in workgroups of one wavefront (rt_kernels.hip launch_render) */

namespace rt {

struct Slab {
	vec3 inv, oi;
};

// compare-exchange of (key, ref) pairs, ascending by key
__device__ __forceinline__ void cswap(float& ka, int& ra, float& kb, int& rb) {
	const bool sw = kb < ka;
	const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
	const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
	ka = k0; kb = k1; ra = r0; rb = r1;
}

template <bool COUNT>
__device__ __forceinline__ Hit cast_ray_lbvh(const DevScene& sc, int* lds_stack, vec3 o, vec3 d, int skip, bool live,
                                             bool shadow, float dist, LaneCounters& lc, WaveProf& wp, HoistCtx hc = HoistCtx(), bool idle_exit = false) {
	RT_PROF(wp, WP_TRACE);
	Hit h;
	h.t = 1e9f;                                                          // Source.cpp:278
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	const bool cull = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) < sc.far_guard;   // false for NaN too
	Slab s;
	{
		const vec3 ds = make3(fabsf(d.x) < 1e-30f ? 1e-30f : d.x, fabsf(d.y) < 1e-30f ? 1e-30f : d.y, fabsf(d.z) < 1e-30f ? 1e-30f : d.z);
		s.inv = make3(__builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y), __builtin_amdgcn_rcpf(ds.z));
		s.oi = make3(-(o.x * s.inv.x), -(o.y * s.inv.y), -(o.z * s.inv.z));
	}
	// where the ray leaves the padded scene box: no hit lies beyond (same arithmetic as the slabs)
	// (idle_exit, wave-uniform: the rays start inside the box, where the bound has nothing to skip — rt_kernels.hip exit_bound_is_idle)
	float lim = 1e9f;
	if (!idle_exit) {
		const float ex = fminf(fminf(fmaxf(__builtin_fmaf(sc.bb_lo[0], s.inv.x, s.oi.x), __builtin_fmaf(sc.bb_hi[0], s.inv.x, s.oi.x)),
		                             fmaxf(__builtin_fmaf(sc.bb_lo[1], s.inv.y, s.oi.y), __builtin_fmaf(sc.bb_hi[1], s.inv.y, s.oi.y))),
		                       fmaxf(__builtin_fmaf(sc.bb_lo[2], s.inv.z, s.oi.z), __builtin_fmaf(sc.bb_hi[2], s.inv.z, s.oi.z)));
		if (cull) lim = fminf(ex + fabsf(ex) * 1e-4f, 1e9f);
	}
	if (shadow) lim = __uint_as_float(__float_as_uint(dist) + 1u);                      // Source.cpp:476 (the exit bound has nothing to add: the lockstep kernel drops it)
	lim = fminf(lim, 1e9f);
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	if (!cull) { s.inv = make3(0.f, 0.f, 0.f); s.oi = make3(0.f, 0.f, 0.f); }           // every (non-empty) box tests as hit
	bool walk = live;
	if (sc.hoist_count > 0) {
		if (hc.by_mask) {                                                   // (rt_kernels.hip hoisted_by_mask: by the lanes' direction masks, no boxes)
			float l2 = live ? lim : -1.0f, l2_hi = l2 + fabsf(l2) * RT_PREFILTER_REL;
			hoisted_by_mask<COUNT>(sc, hc.hm, o, d, skip, shadow, h, l2, l2_hi, lc, wp);
			if (live) { lim = l2; lim_hi = l2_hi; }
		} else hoisted_tests<COUNT>(sc, o, d, s.inv, s.oi, skip, live, shadow, h, lim, lim_hi, lc, wp);
		walk = live && lim >= 0.0f;                                      // a shadow ray that has its hit is done
	}
	const int SENT = 0x7fffffff;
	const int tid = threadIdx.x;
	int ovf[RT_STACK_MAX - RT_STACK_LDS];
	int sp = 0;
	int node = walk ? sc.bvh_root : SENT;
	int* const slot0 = lds_stack + tid;                                  // entry k of this lane: slot0[k * RT_STACK_PITCH]

	// General stack access (any depth); the loop below uses it only when some lane of the wave is
	// near or beyond the LDS part, which is wave-uniformly tested so the common path has no branches.
	auto pop_any = [&]() -> int {
		if (sp == 0) return SENT;
		sp--;
		return sp < RT_STACK_LDS ? slot0[sp * RT_STACK_PITCH] : ovf[min(sp, RT_STACK_MAX - 1) - RT_STACK_LDS];
	};
	auto push_any = [&](int v) {
		if (sp < RT_STACK_LDS) slot0[sp * RT_STACK_PITCH] = v;
		else if (sp < RT_STACK_MAX) ovf[sp - RT_STACK_LDS] = v;
		sp += sp < RT_STACK_MAX;          // (full: unreachable for the trees the launchers send here, see the header)
	};

	// pop for a wave whose lanes may or may not all be within the LDS part
	auto pop = [&]() -> int {
		if (__builtin_amdgcn_ballot_w64(sp > RT_STACK_LDS) == 0ull) {
			const int top = slot0[(sp > 0 ? sp - 1 : 0) * RT_STACK_PITCH];
			const int v = sp > 0 ? top : SENT;
			sp -= sp > 0;
			return v;
		}
		return pop_any();
	};

	const float inf = __builtin_huge_valf();
	// A lane that reaches a leaf parks it in `pend` and keeps walking; the wave tests parked leaves
	// together once RT_LEAF_MIN lanes cannot go on without it (or nobody has an inner node left), so
	// a leaf trip serves many lanes.  The order in which a ray meets its triangles does not matter
	// (see the header), only that all of them are met before the ray finishes.
	int pend = SENT;
	const bool pos_x = s.inv.x >= 0.0f, pos_y = s.inv.y >= 0.0f, pos_z = s.inv.z >= 0.0f;
	while (node != SENT || pend != SENT) {
		const bool inner = (unsigned int)node < (unsigned int)SENT;
#ifdef RT_PROFILE
		if (wp.g) {   // exact wave-level counts (slow: one atomic per event from the first active lane); counters 20..27
			const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
			const unsigned long long mn = __builtin_amdgcn_ballot_w64(inner);
			unsigned long long todo = mn;
			unsigned int distinct = 0;
			while (todo) {
				const int first = __builtin_amdgcn_readlane(node, __builtin_ctzll(todo));
				todo &= ~__builtin_amdgcn_ballot_w64(node == first);
				distinct++;
			}
			if (__builtin_ctzll(act) == (int)__lane_id()) {
				atomicAdd(&wp.g[20], 1ull);                                   // loop trips
				if (mn) { atomicAdd(&wp.g[21], 1ull); atomicAdd(&wp.g[22], (unsigned long long)__popcll(mn)); }   // node trips, node lane-visits
				atomicAdd(&wp.g[25], (unsigned long long)distinct);           // different nodes fetched
				if (distinct == 1) atomicAdd(&wp.g[26], 1ull);                // node trips with one node for the whole wave
				if (distinct <= 4 && distinct > 0) atomicAdd(&wp.g[27], 1ull);
			}
		}
#endif
		if (inner) {
			// (uniform base + 32-bit byte offset: the load's own addressing form, no 64-bit address arithmetic per lane)
			const char* nd = (const char*)sc.bvh4 + (unsigned int)node * 64u;
			const uint4 w0 = *(const uint4*)nd, w1 = *(const uint4*)(nd + 16), w2 = *(const uint4*)(nd + 32);
			const uint4 w3 = *(const uint4*)(nd + 48);
			// per axis: t(q) = (p + q * 2^e) * inv + oi = q * A + B
			const float Ax = __uint_as_float(w0.w) * s.inv.x, Bx = __builtin_fmaf(__uint_as_float(w0.x), s.inv.x, s.oi.x);
			const float Ay = __uint_as_float(w3.z) * s.inv.y, By = __builtin_fmaf(__uint_as_float(w0.y), s.inv.y, s.oi.y);
			const float Az = __uint_as_float(w3.w) * s.inv.z, Bz = __builtin_fmaf(__uint_as_float(w0.z), s.inv.z, s.oi.z);
			float key[4];
			int ref[4] = { (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w };
			// A's sign is the ray direction's: the near slab plane is `lo` when it is positive, `hi` otherwise.
			// Picking the words once per node replaces the per-child min/max of the two slab distances.
			// (asked of inv, not of A = 2^e inv: the same sign whenever it matters — with 2^e = 0 both choices give t = B — and a
			// condition that does not change along the ray, so its three compares leave the loop)
			const unsigned int nx = pos_x ? w2.x : w2.w, fx = pos_x ? w2.w : w2.x;
			const unsigned int ny = pos_y ? w2.y : w3.x, fy = pos_y ? w3.x : w2.y;
			const unsigned int nz = pos_z ? w2.z : w3.y, fz = pos_z ? w3.y : w2.z;
#define RT_QB(word, c) ((float)(((word) >> (8 * (c))) & 0xffu))      /* v_cvt_f32_ubyteN */
			const float lim_q = __builtin_canonicalizef(lim_hi);           // once per node instead of once per child (fminf quiets its operands)
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const float tmin = fmaxf(fmaxf(__builtin_fmaf(RT_QB(nx, c), Ax, Bx), __builtin_fmaf(RT_QB(ny, c), Ay, By)),
				                         fmaxf(__builtin_fmaf(RT_QB(nz, c), Az, Bz), 0.0f));
				const float tmax = fminf(fminf(__builtin_fmaf(RT_QB(fx, c), Ax, Bx), __builtin_fmaf(RT_QB(fy, c), Ay, By)),
				                         fminf(__builtin_fmaf(RT_QB(fz, c), Az, Bz), lim_q));
				const bool present = ref[c] != RT_BVH4_EMPTY;
				key[c] = (present && !(tmin > tmax)) ? tmin : inf;            // A, B are finite: no NaN here
				if (COUNT) lc.box += present;
			}
#undef RT_QB
			// nearest first: descend into the nearest hit, push the others
			// (three compare-exchanges put the nearest hit first; the order of the other three on the stack is left as it
			// falls — the full 5-comparator sort cost more than its better culling saved: 41.0 | 40.2 us at 800x800 and
			// 375 | 365 at 3840x2160 on the 102k-triangle scene, sort | this; no ordering at all: 42.0, 377)
			cswap(key[0], ref[0], key[1], ref[1]);
			cswap(key[2], ref[2], key[3], ref[3]);
			cswap(key[0], ref[0], key[2], ref[2]);
			const bool deep = __builtin_amdgcn_ballot_w64(sp > RT_STACK_LDS - 3) != 0ull;
			if (!deep) {
				// every lane has room for three entries: store unconditionally, advance only for hits
				slot0[sp * RT_STACK_PITCH] = ref[3]; sp += key[3] < inf;
				slot0[sp * RT_STACK_PITCH] = ref[2]; sp += key[2] < inf;
				slot0[sp * RT_STACK_PITCH] = ref[1]; sp += key[1] < inf;
				const int top = slot0[(sp > 0 ? sp - 1 : 0) * RT_STACK_PITCH];
				const bool hit0 = key[0] < inf;
				node = hit0 ? ref[0] : (sp > 0 ? top : SENT);
				sp -= (!hit0 && sp > 0);
			} else {
				if (key[3] < inf) push_any(ref[3]);
				if (key[2] < inf) push_any(ref[2]);
				if (key[1] < inf) push_any(ref[1]);
				node = key[0] < inf ? ref[0] : pop_any();
			}
		}
		if (node < 0 && pend == SENT) { pend = node; node = pop(); }                  // park the leaf, keep walking
		const bool has = pend != SENT;
		const bool walking = (unsigned int)node < (unsigned int)SENT;
		const unsigned long long stuck = __builtin_amdgcn_ballot_w64(has && !walking);
		const bool leaf_trip = __popcll(stuck) >= RT_LEAF_MIN || (stuck != 0ull && __builtin_amdgcn_ballot_w64(walking) == 0ull);
#ifdef RT_PROFILE
		if (wp.g && leaf_trip) {
			const unsigned long long ml = __builtin_amdgcn_ballot_w64(has);
			if (__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)) == (int)__lane_id()) { atomicAdd(&wp.g[23], 1ull); atomicAdd(&wp.g[24], (unsigned long long)__popcll(ml)); }
		}
#endif
		if (leaf_trip && has) {
			const int i = ~pend;
			const float4* T = (const float4*)((const char*)sc.tri_aos + (unsigned int)i * (RT_AOS_F4 * 16u));
			const float4 N4 = T[0], P14 = T[1], M0 = T[2], M1 = T[3], M2 = T[4];             // (the edge data only if the filter leaves a lane open)
			const vec3 nn = xyz(N4), p1 = xyz(P14);
			const float t = dot(p1 - o, nn) / dot(d, nn);                                 // Source.cpp:383
			// lane predicates as wave masks (rt_kernels.hip lanemask); the lanes of this leaf trip only
			const lanemask visit = lanes(i != skip);                                      // Source.cpp:302
			const lanemask better = lanes(t < lim) | (lanes(t == h.t) & lanes(i < h.idx)); // Source.cpp:391 + index order
			const lanemask cand = visit & lanes(t > RT_EPS_BELOW) & better;               // Source.cpp:385, 391
			if (COUNT) { lc.plane += lane_in(visit); lc.point += lane_in(visit) && !(t < 0.0f); lc.edge += lane_in(cand); }
			lanemask found_m = 0;
			if (cand != 0) {
				const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);        // Source.cpp:389
				const vec3 w = I - p1;
				lanemask open = cand;
				if (!COUNT) found_m = edge_filter(M0, M1, M2, w, cand, open);             // rt_kernels.hip; the counting kernel takes the reference's steps
				if (open != 0) {
					const float4 P24 = T[5], P34 = T[6], E0 = T[7], E1 = T[8], E2 = T[9];
					const lanemask c0 = open & lanes(dot(nn, cross(xyz(E0), w)) > 0.0f);  // Source.cpp:392
					if (COUNT) lc.edge += lane_in(c0);
					if (c0 != 0) {
						const lanemask c1 = c0 & lanes(dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
						if (COUNT) lc.edge += lane_in(c1);
						if (c1 != 0) found_m |= c1 & lanes(dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);   // Source.cpp:394
					}
				}
				if (lane_in(found_m)) {
					h.t = t; h.idx = i; h.p = I;                                          // Source.cpp:396
					lim = t;
					lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
				}
			}
			const bool found = lane_in(found_m);
			pend = SENT;
			if (found && shadow) { node = SENT; sp = 0; }                                 // a shadow ray only asks whether a hit with t <= dist exists
			else if (node < 0) { pend = node; node = pop(); }
		}
	}
	return h;
}

} // namespace rt
