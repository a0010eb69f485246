// rt_shards.hip — the exchange step's de-interleave (rt355.h rt_scatter_shards): after the gather, rank 0 holds
// every rank's COMPACT shard (the rows of its interleaved tiles in increasing y, as rt_render_tiles_device writes them) at
// shards + r * pitch; the frame the reference's viewer uploads (Source.cpp:1270) has row y at y * row_bytes.  One kernel
// moves every row to its place: row y belongs to tile t = y / tile_rows, which rank t % n rendered as its (t / n)-th tile.
// Pure data movement, HBM-bound: 16 bytes per lane when rows and pitches allow it, else 4 (pixels are 4 or 12 bytes).
#include "rt_internal.h"

namespace rt {

template <typename V>
__global__ void __launch_bounds__(256) scatter_shards_kernel(const V* __restrict__ shards, size_t pitch_v, int n, int row_v, int height, int tile_rows,
                                                             V* __restrict__ frame) {
	const int y = (int)blockIdx.y;
	const int t = y / tile_rows, r = t % n, local_row = (t / n) * tile_rows + (y - t * tile_rows);
	const V* src = shards + (size_t)r * pitch_v + (size_t)local_row * (size_t)row_v;
	V* dst = frame + (size_t)y * (size_t)row_v;
	for (int x = (int)(blockIdx.x * blockDim.x + threadIdx.x); x < row_v; x += (int)(gridDim.x * blockDim.x)) dst[x] = src[x];
}

int launch_scatter_shards(const void* shards, size_t pitch_bytes, int n, size_t row_bytes, int height, int tile_rows, void* frame, hipStream_t stream) {
	if (height <= 0 || row_bytes == 0) return RT_OK;
	const bool wide = row_bytes % 16 == 0 && pitch_bytes % 16 == 0 && ((size_t)shards | (size_t)frame) % 16 == 0;
	const int row_v = (int)(row_bytes / (wide ? 16 : 4));
	const dim3 grid((unsigned int)std::min(64, (row_v + 255) / 256), (unsigned int)height), block(256);
	if (wide) hipLaunchKernelGGL((scatter_shards_kernel<uint4>), grid, block, 0, stream, (const uint4*)shards, pitch_bytes / 16, n, row_v, height, tile_rows, (uint4*)frame);
	else hipLaunchKernelGGL((scatter_shards_kernel<unsigned int>), grid, block, 0, stream, (const unsigned int*)shards, pitch_bytes / 4, n, row_v, height, tile_rows, (unsigned int*)frame);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return hip_fail(e, "shard scatter launch");
	return RT_OK;
}

} // namespace rt
