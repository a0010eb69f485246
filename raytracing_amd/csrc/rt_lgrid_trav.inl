// rt_lgrid_trav.inl — shadow rays through a light's direction grid (device side; the build is rt_lgrid.hip, whose header
// says why the lists are complete).  Included by rt_kernels.hip after rt_lbvh_trav.inl.
//
// processLighting's question (Source.cpp:473-476) for one light and the rays of a wavefront: is any triangle accepted
// with t <= dist?  Per ray: (1) the light's few "always" triangles, wave-uniform from scalar loads; (2) the list of
// the ONE cell the ray's direction, seen from the light, falls into — per lane, nearest the light first, ending at the
// first entry that lies farther from the light than the ray's own origin; (3) of the scene-spanning triangles kept out
// of the tree, those the light's coarse direction table lists for this ray's direction, for the lanes still unblocked.
// Every triangle test is the reference's arithmetic (Source.cpp:383-396) with the edge-term sign filter of edge_filter();
// the first accepted triangle ends the lane's search (any-hit: which triangle blocks does not matter to Source.cpp:476).
#pragma once

namespace rt {

// One lane, one triangle each (lanes in `act`): Source.cpp:383-396 on the triangle's AoS record.
template <bool COUNT>
__device__ __forceinline__ void lane_triangle_any_hit(const DevScene& sc, int i, bool act, vec3 o, vec3 d, int skip, Hit& h, float& lim, LaneCounters& lc) {
	const float4* T = (const float4*)((const char*)sc.tri_aos + (unsigned int)i * (RT_AOS_F4 * 16u));
	const float4 N4 = T[0], P14 = T[1], M0 = T[2], M1 = T[3], M2 = T[4];
	const vec3 nn = xyz(N4), p1 = xyz(P14);
	const float t = dot(p1 - o, nn) / dot(d, nn);                                   // Source.cpp:383
	const lanemask visit = lanes(act) & lanes(i != skip);                           // Source.cpp:302
	const lanemask cand = visit & lanes(t > RT_EPS_BELOW) & lanes(t < lim);         // Source.cpp:385, 391 (lim: the first float past dist)
	if (COUNT) { lc.plane += lane_in(visit); lc.point += lane_in(visit) && !(t < 0.0f); lc.edge += lane_in(cand); }
	if (cand == 0) return;
	const vec3 I = make3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);              // Source.cpp:389
	const vec3 w = I - p1;
	lanemask open = cand, found = 0;
	if (!COUNT) found = edge_filter(M0, M1, M2, w, cand, open);
	if (open != 0) {
		const float4 P24 = T[5], P34 = T[6], E0 = T[7], E1 = T[8], E2 = T[9];
		const lanemask c0 = open & lanes(dot(nn, cross(xyz(E0), w)) > 0.0f);        // Source.cpp:392
		if (COUNT) lc.edge += lane_in(c0);
		if (c0 != 0) {
			const lanemask c1 = c0 & lanes(dot(nn, cross(xyz(E1), I - xyz(P24))) > 0.0f);   // Source.cpp:393
			if (COUNT) lc.edge += lane_in(c1);
			if (c1 != 0) found |= c1 & lanes(dot(nn, cross(xyz(E2), I - xyz(P34))) > 0.0f);   // Source.cpp:394
		}
	}
	if (lane_in(found)) { h.t = t; h.idx = i; h.p = I; lim = -1.0f; }               // Source.cpp:396; nothing more to ask for this ray
}

// li: wave-uniform light index with a grid (G.R > 0).  L: that light's position.  `live` lanes hold a shadow ray from o
// along d = normalize(L - o), dist = |L - o| (Source.cpp:470-474).
template <bool COUNT>
__device__ __forceinline__ Hit grid_shadow(const DevScene& sc, int li, vec3 L, vec3 o, vec3 d, int skip, bool live, float dist, LaneCounters& lc, WaveProf& wp) {
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	float lim = live ? fminf(__uint_as_float(__float_as_uint(dist) + 1u), 1e9f) : -1.0f;    // Source.cpp:476; (dist == inf -> NaN -> 1e9)
	float lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
	// the record of this light: scalar loads
	const __attribute__((address_space(4))) LightGrid* G = (const __attribute__((address_space(4))) LightGrid*)(unsigned long long)(sc.lgrid + li);
	const int R = G->R, n_over = G->n_over;
	const float half_R = G->half_R;
	// (1) triangles tested for every ray of this light
	if (n_over > 0) {
		const c4ptr aos = c4ptr_of(sc.tri_aos);
		const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
		for (int k = 0; k < n_over; k++) {
			const int i = G->over[k];
			packet_leaf<COUNT>(aos.offset_floats((unsigned int)i * (RT_AOS_F4 * 4u)), i, o, d, skip, true, eps_lo, h, lim, lim_hi, lc, wp);
		}
	}
	// (2) the cell this ray's direction falls into, as seen from the light
	float far;
	unsigned int hm = 0u;
	{
		const vec3 q = o - L;
		const float ax = fabsf(q.x), ay = fabsf(q.y), az = fabsf(q.z);
		const bool xm = ax >= ay && ax >= az, ym = !xm && ay >= az;
		const float qa = xm ? q.x : (ym ? q.y : q.z), qb = xm ? q.y : (ym ? q.z : q.x), qc = xm ? q.z : (ym ? q.x : q.y);
		const int face = (xm ? 0 : (ym ? 2 : 4)) + (qa < 0.0f ? 1 : 0);
		const float inv = __builtin_amdgcn_rcpf(fabsf(qa));
		const float top = (float)(R - 1);
		// (a ray that starts AT the light has u = v = NaN: fmaxf drops it, cell 0 of the face; its list is cut off at once by distance)
		const int iu = (int)fminf(fmaxf(__builtin_fmaf(qb * inv, half_R, half_R), 0.0f), top);
		const int iv = (int)fminf(fmaxf(__builtin_fmaf(qc * inv, half_R, half_R), 0.0f), top);
		const unsigned int cell = ((unsigned int)face * (unsigned int)R + (unsigned int)iv) * (unsigned int)R + (unsigned int)iu;
		const int* cs = G->cell_start;
		const int2* entries = G->entries;
		int k = 0, kend = 0;
		if (live) { k = cs[cell]; kend = cs[cell + 1]; }
#ifdef RT_EXPERIMENT
		if (sc.debug & 1) kend = k;
#endif
		far = __builtin_fmaf(dist, 1e-5f, dist) + G->slack;                          // entries beyond lie past the ray's own origin
		if (sc.hoist_count > 0 && live) {                                            // the hoisted triangles this direction can meet
			const float hc = 0.5f * (float)RT_LGRID_HC, htop = (float)(RT_LGRID_HC - 1);
			const int cu = (int)fminf(fmaxf(__builtin_fmaf(qb * inv, hc, hc), 0.0f), htop);
			const int cv = (int)fminf(fmaxf(__builtin_fmaf(qc * inv, hc, hc), 0.0f), htop);
			hm = G->hmask[(face * RT_LGRID_HC + cv) * RT_LGRID_HC + cu];
		}
		int2 e = make_int2(0x7f800000, 0);
		if (k < kend && lim >= 0.0f) e = entries[k];
		for (;;) {
			bool act = k < kend && lim >= 0.0f;
			if (act && !(__int_as_float(e.x) <= far)) { act = false; k = kend; }       // sorted by distance: nothing nearer follows
			if (__builtin_amdgcn_ballot_w64(act) == 0ull) break;
			const int i = act ? e.y : 0;
			k++;
			if (k < kend) e = entries[k];                                            // the next entry travels while this triangle is tested
			RT_PROF(wp, WP_TRI);
			lane_triangle_any_hit<COUNT>(sc, i, act, o, d, skip, h, lim, lc);
		}
	}
	// (3) the triangles kept out of the tree that this ray's direction can meet and that are not beyond its origin
#ifdef RT_EXPERIMENT
	if (sc.debug & 2) hm = 0u;
#endif
	if (sc.hoist_count > 0 && __builtin_amdgcn_ballot_w64(lim >= 0.0f && hm != 0u) != 0ull) {
		const c4ptr hrec = c4ptr_of(sc.hoist);
		const float eps_lo = RT_EPS_BELOW * (1.0f - RT_PREFILTER_REL);
		lim_hi = lim + fabsf(lim) * RT_PREFILTER_REL;
		// the lanes of a tile nearly always share one mask: take a lane's mask as a scalar, walk its set bits, repeat for what is left
		unsigned int left = lim >= 0.0f ? hm : 0u;
		for (;;) {
			const unsigned long long any = __builtin_amdgcn_ballot_w64(left != 0u);
			if (any == 0ull) break;
			unsigned int m = __builtin_amdgcn_readlane(left, __builtin_ctzll(any));
			left &= ~m;
			while (m != 0u) {
				const int j = __builtin_ctz(m);
				m &= m - 1u;
				const int i = G->hidx[j];
				const bool want = lim >= 0.0f && ((hm >> j) & 1u) != 0u && G->hmind[j] <= far && i != skip;
				if (__builtin_amdgcn_ballot_w64(want) == 0ull) continue;
				packet_leaf<COUNT>(hrec.offset_floats((unsigned int)j * (RT_AOS_F4 * 4u)), i, o, d, skip, true, eps_lo, h, lim, lim_hi, lc, wp);
			}
		}
	}
	return h;
}

} // namespace rt
