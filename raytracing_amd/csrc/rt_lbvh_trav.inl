// rt_lbvh_trav.inl — LBVH traversal (device side).  Included by rt_kernels.hip.
#pragma once

namespace rt {

template <bool COUNT>
__device__ __forceinline__ Hit cast_ray_lbvh(const DevScene& sc, vec3 o, vec3 d, int skip, bool live, bool shadow, float dist, LaneCounters& lc) {
	(void)sc; (void)o; (void)d; (void)skip; (void)live; (void)shadow; (void)dist; (void)lc;
	Hit h;
	h.t = 1e9f;
	h.idx = -1;
	h.p = make3(0.f, 0.f, 0.f);
	return h;
}

} // namespace rt
