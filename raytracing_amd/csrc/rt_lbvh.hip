// rt_lbvh.hip — GPU LBVH build (Morton sort + Karras).  Placeholder until the build lands.
#include "rt_internal.h"

namespace rt {

int lbvh_build(rt_scene* s, DeviceState& d, hipStream_t stream) {
	(void)s; (void)d; (void)stream;
	set_error("LBVH build is not available in this build");
	return RT_ERR_LIMIT;
}

void lbvh_free(DeviceState& d) {
	if (d.bvh_nodes) (void)hipFree(d.bvh_nodes);
	if (d.bvh_prim) (void)hipFree(d.bvh_prim);
	d.bvh_nodes = nullptr;
	d.bvh_prim = nullptr;
}

} // namespace rt
