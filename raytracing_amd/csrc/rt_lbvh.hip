// rt_lbvh.hip — GPU-built linear BVH for large scenes (BASELINE config 4: 100k triangles).
//
// The reference has no acceleration structure: castRay walks every triangle (Source.cpp:301).
// north_star replaces that O(N) walk for large scenes by an LBVH built ON the GPU:
//
//   1. k_morton      one lane per triangle: centroid of its vertex box, normalised by the scene
//                    box, 21 bits per axis interleaved to a 63-bit Morton code; key = (as many
//                    leading code bits as fit) : (triangle index in ceil(log2 N) bits), 63 bits in
//                    all, so keys are unique (Karras 2012, §4: "augment each key with its index")
//                    and equal codes still split into a balanced subtree.  At 100k triangles that
//                    is 15 code bits per axis.
//   2. radix sort    rocPRIM device radix sort of the 64-bit keys (the bits in use).
//   3. k_karras      one lane per internal node i in [0, N-2]: direction, range and split from
//                    the longest-common-prefix function delta(i, j) = clz(key_i ^ key_j)
//                    (Karras 2012, Alg. in §3); writes the two child references and the parent
//                    links.  Node 0 is the root.
//   4. k_refit       one lane per leaf walks to the root; the second lane to arrive at a node
//                    (atomic counter) merges the two child boxes and carries on.  Each node stores
//                    ITS CHILDREN's boxes, so one 64-byte fetch during traversal yields both slab
//                    tests.
//
// Node (4 x float4 = 64 B, 64-byte aligned):
//   q0 = (L.lo.x, L.lo.y, L.lo.z, L.hi.x)   q1 = (L.hi.y, L.hi.z, R.lo.x, R.lo.y)
//   q2 = (R.lo.z, R.hi.x, R.hi.y, R.hi.z)   q3 = (left ref, right ref, 0, 0) as int bits
// A child reference >= 0 is an internal node index; < 0 is a leaf holding triangle ~ref.
// Leaf boxes are the triangle's vertex box padded by the scene pad (1e-3 of the scene scale): boxes
// only ever skip work, the triangle test at the leaf is the reference's own arithmetic.
//
// Cross-workgroup hand-off in k_refit (MI355X: per-CU L1 is not coherent, per-XCD L2s are not
// coherent with each other): the arriving lane stores its box with agent-scope (sc1, write-through)
// stores, issues an agent release fence + explicit vmcnt(0) wait, then the returning agent-scope
// atomic add; the second arriver issues an agent acquire fence and reads the sibling's slots with
// agent-scope (L1-bypassing) loads.
#include "rt_internal.h"

#include <cstring>
#include <cstdlib>
#include <chrono>
#include <rocprim/device/device_radix_sort.hpp>

namespace rt {

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ u64 expand_bits21(u64 v) {
	// 21 bits -> every third bit of 63
	v = (v | (v << 32)) & 0x001f00000000ffffull;
	v = (v | (v << 16)) & 0x001f0000ff0000ffull;
	v = (v | (v << 8)) & 0x100f00f00f00f00full;
	v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
	v = (v | (v << 2)) & 0x1249249249249249ull;
	return v;
}

__device__ __forceinline__ void tri_box(const float4* __restrict__ planes, int stride, int i, float lo[3], float hi[3]) {
	const float4 a = planes[PL_P1 * stride + i], b = planes[PL_P2 * stride + i], c = planes[PL_P3 * stride + i];
	lo[0] = fminf(a.x, fminf(b.x, c.x)); hi[0] = fmaxf(a.x, fmaxf(b.x, c.x));
	lo[1] = fminf(a.y, fminf(b.y, c.y)); hi[1] = fmaxf(a.y, fmaxf(b.y, c.y));
	lo[2] = fminf(a.z, fminf(b.z, c.z)); hi[2] = fmaxf(a.z, fmaxf(b.z, c.z));
}

struct BoxF { float lo[3], hi[3]; };

// Padded box of ONE triangle as the tree's leaves, the hoisted list and the per-lane leaf resolution use
// it.  Boxes only ever skip work; what the pad has to cover is (1) the rounding of the slab arithmetic,
// about 5e-7 of the scene scale S for origins inside the scene and 30 times that at the far_guard of an
// LBVH commit (30 S, rt_api.cpp), and (2) hit points the reference's own edge tests accept beyond a
// needle's tips, about 1e-7 * aspect ratio * its length L.  pad = 1e-4 S + 1e-3 L covers (1) six times over
// and (2) up to an aspect of 1:10^4, as the flat 1e-3 S of round 1 did for scene-sized needles — and for
// the 0.05-unit triangles of the 100k-triangle scene it is 0.002 instead of 0.02: boxes of 0.054 instead
// of 0.09 a side, about a third of the box hits at the leaves.
#ifndef RT_LEAF_PAD_SCENE
#define RT_LEAF_PAD_SCENE 0.1f     /* of the commit's pad (1e-3 S) */
#endif
__device__ __forceinline__ void tri_box_padded(const float4* __restrict__ planes, int stride, int i, float pad, float lo[3], float hi[3]) {
	tri_box(planes, stride, i, lo, hi);
	const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
	const float p = pad * RT_LEAF_PAD_SCENE + 1e-3f * sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
	for (int k = 0; k < 3; k++) { lo[k] -= p; hi[k] += p; }
}

// key = (Morton code, as many leading bits as fit) : (triangle index in `index_bits` bits); 63 bits in all
__global__ void __launch_bounds__(256) k_morton(const float4* __restrict__ planes, int stride, int n, BoxF scene, int index_bits,
                                               u64* __restrict__ keys) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float lo[3], hi[3];
	tri_box(planes, stride, i, lo, hi);
	u64 code = 0;
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const float ext = scene.hi[k] - scene.lo[k];
		const float c = 0.5f * (lo[k] + hi[k]);
		float u = ext > 0.f ? (c - scene.lo[k]) / ext : 0.5f;
		u = fminf(fmaxf(u * 2097152.0f, 0.0f), 2097151.0f);
		code |= expand_bits21((u64)(unsigned int)u) << (2 - k);
	}
	const int code_bits = (63 - index_bits) / 3 * 3;                     // whole x/y/z triples
	keys[i] = ((code >> (63 - code_bits)) << index_bits) | (u64)(unsigned int)i;
}

// delta(i, j): common-prefix length of keys i and j, -1 outside [0, n)
__device__ __forceinline__ int delta(const u64* __restrict__ keys, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	return __clzll((long long)(keys[i] ^ keys[j]));      // keys are unique, the xor is never 0
}

__global__ void __launch_bounds__(256) k_karras(const u64* __restrict__ keys, int n, unsigned int index_mask, float4* __restrict__ nodes,
                                               int* __restrict__ parent_int, int* __restrict__ parent_leaf) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1) return;
	// direction of the range
	const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
	const int dmin = delta(keys, n, i, i - d);
	int lmax = 2;
	while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1)
		if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	// split position
	const int dnode = delta(keys, n, i, j);
	int s = 0;
	int t = l;
	do {
		t = (t + 1) >> 1;
		if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	const int gamma = i + s * d + min(d, 0);
	const int first = min(i, j), last = max(i, j);
	const int left = (first == gamma) ? ~(int)((unsigned int)keys[gamma] & index_mask) : gamma;
	const int right = (last == gamma + 1) ? ~(int)((unsigned int)keys[gamma + 1] & index_mask) : gamma + 1;
	nodes[(size_t)i * 4 + 3] = make_float4(__int_as_float(left), __int_as_float(right), 0.f, 0.f);
	// parent links: (node << 1) | side
	if (first == gamma) parent_leaf[gamma] = (i << 1); else parent_int[gamma] = (i << 1);
	if (last == gamma + 1) parent_leaf[gamma + 1] = (i << 1) | 1; else parent_int[gamma + 1] = (i << 1) | 1;
	if (i == 0) parent_int[0] = -1;
}

__device__ __forceinline__ void store_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// float offsets of child `side`'s box inside a node: lo xyz, hi xyz
__device__ __forceinline__ void store_child_box(float* node, int side, const float lo[3], const float hi[3]) {
	const int b = side * 6;
#pragma unroll
	for (int k = 0; k < 3; k++) { store_agent(node + b + k, lo[k]); store_agent(node + b + 3 + k, hi[k]); }
}

__global__ void __launch_bounds__(256) k_refit(const float4* __restrict__ planes, int stride, int n, const u64* __restrict__ keys,
                                              unsigned int index_mask, float* __restrict__ nodes, const int* __restrict__ parent_int,
                                              const int* __restrict__ parent_leaf, unsigned int* __restrict__ flags, float pad) {
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n) return;
	const int tri = (int)((unsigned int)keys[p] & index_mask);
	float lo[3], hi[3];
	tri_box_padded(planes, stride, tri, pad, lo, hi);
	int link = parent_leaf[p];
	// every lane's walk ends: at a node it reaches first, or at the root
	for (int guard = 0; guard < 128 && link >= 0; guard++) {
		const int node = link >> 1, side = link & 1;
		float* nd = nodes + (size_t)node * 16;
		store_child_box(nd, side, lo, hi);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned int old = __hip_atomic_fetch_add(&flags[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (old == 0) return;                          // the sibling subtree is not done: its lane carries on
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		const int ob = (side ^ 1) * 6;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			lo[k] = fminf(lo[k], load_agent(nd + ob + k));
			hi[k] = fmaxf(hi[k], load_agent(nd + ob + 3 + k));
		}
		link = parent_int[node];
	}
}

// 5. k_collapse4: the binary tree walked two levels at a time, into 64-byte quantised nodes.
//    4-wide node i holds the grandchildren of binary node i (a child that is a leaf stands for
//    itself), so a ray fetches half as many nodes in sequence.  Every binary node gets a 4-wide
//    twin; only those at even depth are ever reached from the root, the rest are never read.
//    Traversal is bound by the texture addresser (one address-processing slot per load
//    instruction, measured 75-80 % busy), so the node is made to fit FOUR 16-byte loads instead of
//    seven: child boxes are stored as 8-bit offsets from the node's own corner p in units of
//    2^e per axis, rounded outward (lo down, hi up) — a quantised box contains the padded box it
//    stands for, and boxes only ever skip work.  Layout (4 x 16 B):
//      w0 = (p.x, p.y, p.z, scale.x)      scale_a = 2^e_a as a binary32 (round 1 packed three exponent bytes into w0.w:
//                                         six integer instructions per node to unpack, and w3 had the room)
//      w1 = the 4 child references (RT_BVH4_EMPTY for an unused slot)
//      w2 = (lo.x of children 0..3 as 4 bytes, lo.y, lo.z, hi.x)     w3 = (hi.y, hi.z, scale.y, scale.z)
__device__ __forceinline__ unsigned int quant_axis(const float lo[4], const float hi[4], int cnt, float& p, unsigned int& qlo, unsigned int& qhi) {
	float mn = lo[0], mx = hi[0];
	for (int c = 1; c < cnt; c++) { mn = fminf(mn, lo[c]); mx = fmaxf(mx, hi[c]); }
	p = mn;
	// smallest power of two s with 255 * s >= extent (a little slack for the roundings below)
	const float step = fmaxf((mx - mn) * (1.0f / 254.0f), 1e-30f);
	int e;
	(void)frexpf(step, &e);                      // step = m * 2^e, m in [0.5, 1)  ->  2^e >= step
	const float s = ldexpf(1.0f, e);
	qlo = 0; qhi = 0;
	for (int c = 0; c < 4; c++) {
		unsigned int ql = 255u, qh = 0u;         // unused slot (never consulted: its reference is RT_BVH4_EMPTY)
		if (c < cnt) {
			float fl = floorf((lo[c] - mn) / s), fh = ceilf((hi[c] - mn) / s);
			// the subtraction rounds: step outward until the decoded corner really encloses the box
			while (fl > 0.0f && mn + fl * s > lo[c]) fl -= 1.0f;
			while (fh < 255.0f && mn + fh * s < hi[c]) fh += 1.0f;
			ql = (unsigned int)fminf(fmaxf(fl, 0.0f), 255.0f);
			qh = (unsigned int)fminf(fmaxf(fh, 0.0f), 255.0f);
		}
		qlo |= ql << (8 * c);
		qhi |= qh << (8 * c);
	}
	return (unsigned int)(e + 127);
}

__global__ void __launch_bounds__(256) k_collapse4(const float4* __restrict__ nodes, int n_nodes, uint4* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_nodes) return;
	float lo[3][4], hi[3][4];
	int ref[4];
	int cnt = 0;
	const float* nd = (const float*)(nodes + (size_t)i * 4);
#ifdef RT_COLLAPSE_FIXED
	for (int side = 0; side < 2; side++) {
		const int c = __float_as_int(nd[12 + side]);
		if (c >= 0) {
			const float* cd = (const float*)(nodes + (size_t)c * 4);
			for (int s2 = 0; s2 < 2; s2++) {
				for (int k = 0; k < 3; k++) { lo[k][cnt] = cd[s2 * 6 + k]; hi[k][cnt] = cd[s2 * 6 + 3 + k]; }
				ref[cnt++] = __float_as_int(cd[12 + s2]);
			}
		} else {
			for (int k = 0; k < 3; k++) { lo[k][cnt] = nd[side * 6 + k]; hi[k][cnt] = nd[side * 6 + 3 + k]; }
			ref[cnt++] = c;
		}
	}
#else
	// Greedy by surface area: start from the two children, twice replace the inner child with the
	// largest box by its own two children.  Big children are opened first, so the four boxes a ray
	// is tested against are of similar size and the node is full whenever the subtree allows.
	for (int side = 0; side < 2; side++) {
		for (int k = 0; k < 3; k++) { lo[k][cnt] = nd[side * 6 + k]; hi[k][cnt] = nd[side * 6 + 3 + k]; }
		ref[cnt++] = __float_as_int(nd[12 + side]);
	}
	for (int round = 0; round < 2; round++) {
		int best = -1;
		float best_area = -1.0f;
		for (int c = 0; c < cnt; c++) {
			if (ref[c] < 0) continue;                                   // a leaf stands for itself
			const float dx = hi[0][c] - lo[0][c], dy = hi[1][c] - lo[1][c], dz = hi[2][c] - lo[2][c];
			const float area = dx * dy + dy * dz + dz * dx;
			if (area > best_area) { best_area = area; best = c; }
		}
		if (best < 0) break;
		const float* cd = (const float*)(nodes + (size_t)ref[best] * 4);
		for (int k = 0; k < 3; k++) { lo[k][cnt] = cd[6 + k]; hi[k][cnt] = cd[9 + k]; }
		ref[cnt] = __float_as_int(cd[13]);
		for (int k = 0; k < 3; k++) { lo[k][best] = cd[k]; hi[k][best] = cd[3 + k]; }
		ref[best] = __float_as_int(cd[12]);
		cnt++;
	}
#endif
	for (int c = cnt; c < 4; c++) ref[c] = RT_BVH4_EMPTY;
	float p[3];
	unsigned int ql[3], qh[3], eb[3];
	for (int k = 0; k < 3; k++) eb[k] = quant_axis(lo[k], hi[k], cnt, p[k], ql[k], qh[k]);
	uint4* o = out + (size_t)i * 4;
	o[0] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), eb[0] << 23);
	o[1] = make_uint4((unsigned int)ref[0], (unsigned int)ref[1], (unsigned int)ref[2], (unsigned int)ref[3]);
	o[2] = make_uint4(ql[0], ql[1], ql[2], qh[0]);
	o[3] = make_uint4(qh[1], qh[2], eb[1] << 23, eb[2] << 23);
}

__device__ __forceinline__ void aos_record(const float4* __restrict__ planes, int stride, int i, float4* __restrict__ o) {
	o[0] = planes[PL_N * stride + i];
	o[1] = planes[PL_P1 * stride + i];
	o[2] = planes[PL_M0 * stride + i];
	o[3] = planes[PL_M1 * stride + i];
	o[4] = planes[PL_M2 * stride + i];
	o[5] = planes[PL_P2 * stride + i];
	o[6] = planes[PL_P3 * stride + i];
	o[7] = planes[PL_E0 * stride + i];
	o[8] = planes[PL_E1 * stride + i];
	o[9] = planes[PL_E2 * stride + i];
}

// 6. k_aos: the SoA planes regrouped per triangle so that a leaf test is ONE burst of loads
//    (N, P1 and, for candidates, the edge data) instead of up to four dependent round trips.
__global__ void __launch_bounds__(256) k_aos(const float4* __restrict__ planes, int stride, int n, float4* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	aos_record(planes, stride, i, out + (size_t)i * RT_AOS_F4);
}

// 7. k_packet: per octant, the node's two children with the one nearer along that octant first and
//    the box faces ordered near/far per axis (rt_internal.h DevScene::skiplist): what the
//    wave-uniform traversal reads with one 64-byte scalar load per node.
__global__ void __launch_bounds__(256) k_packet(const float4* __restrict__ nodes, int n_nodes, float4* __restrict__ out) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_nodes) return;
	const float* nd = (const float*)(nodes + (size_t)i * 4);
	float lo[2][3], hi[2][3];
	int ref[2];
	for (int side = 0; side < 2; side++) {
		for (int k = 0; k < 3; k++) { lo[side][k] = nd[side * 6 + k]; hi[side][k] = nd[side * 6 + 3 + k]; }
		ref[side] = __float_as_int(nd[12 + side]);
	}
	// the axis on which the two centres are farthest apart decides which child is nearer
	int axis = 0;
	float best = -1.f;
	for (int k = 0; k < 3; k++) {
		const float sep = fabsf((lo[0][k] + hi[0][k]) - (lo[1][k] + hi[1][k]));
		if (sep > best) { best = sep; axis = k; }
	}
	const bool left_lower = (lo[0][axis] + hi[0][axis]) <= (lo[1][axis] + hi[1][axis]);
	for (int oct = 0; oct < 8; oct++) {
		const bool neg_axis = (oct >> axis) & 1;
		const int a = (left_lower != neg_axis) ? 0 : 1, b = 1 - a;
		float v[16];
		// per child: near.x near.y far.x far.y near.z far.z — the pairs the traversal's packed multiply-adds take as they lie
		for (int k = 0; k < 3; k++) {
			const bool neg = (oct >> k) & 1;
			const int at_near = k < 2 ? k : 4, at_far = k < 2 ? 2 + k : 5;
			v[at_near] = neg ? hi[a][k] : lo[a][k];
			v[at_far] = neg ? lo[a][k] : hi[a][k];
			v[8 + at_near] = neg ? hi[b][k] : lo[b][k];
			v[8 + at_far] = neg ? lo[b][k] : hi[b][k];
		}
		v[6] = __int_as_float(ref[a]);
		v[7] = __int_as_float(ref[b]);
		v[14] = 0.f; v[15] = 0.f;
		float4* o = out + ((size_t)oct * n_nodes + i) * 4;
		o[0] = make_float4(v[0], v[1], v[2], v[3]);
		o[1] = make_float4(v[4], v[5], v[6], v[7]);
		o[2] = make_float4(v[8], v[9], v[10], v[11]);
		o[3] = make_float4(v[12], v[13], v[14], v[15]);
	}
}

// 9. depth of a tree on the device (what the traversal stacks have to hold), level by level from the root: a launch of k_level
//    gives every inner child of the nodes at `level` the level + 1 and notes that there was one.  `words` are the nodes as 16
//    ints each, the `width` child references start at word `first`; a reference r with 0 <= r < n_nodes is an inner node,
//    anything else a leaf or an empty slot.  (Walking up parent pointers instead does not work for the 4-wide collapse: every
//    binary node has a 4-wide record, reachable or not, and several of them may name the same child.)
__global__ void __launch_bounds__(256) k_level(const int* __restrict__ words, int n_nodes, int first, int width, int level, int* __restrict__ depth, int* __restrict__ deepest) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_nodes || depth[i] != level) return;
	for (int c = 0; c < width; c++) {
		const int r = words[(size_t)i * 16 + first + c];
		if (r >= 0 && r < n_nodes) { depth[r] = level + 1; *deepest = level + 1; }      // (every writer of a launch stores the same value)
	}
}

// 8. hoisted triangles (DevScene::hoist): their keys get the top bit, so they sort behind every
//    triangle of the tree; their records are copied out in index order, followed by their padded
//    boxes ordered near/far for each octant.  Two neighbours of the list whose boxes coincide (the two
//    halves of a quad, Source.cpp:815-816: both span the quad's whole box) share ONE box test: the first
//    carries the union and the triangle index of the second in its spare slot (-1: none), the second
//    is stepped over by the traversals.
__global__ void __launch_bounds__(64) k_hoist(const float4* __restrict__ planes, int stride, const int* __restrict__ idx, int k, float pad,
                                             u64* __restrict__ keys, float4* __restrict__ out) {
	__shared__ float s_lo[RT_HOIST_MAX][3], s_hi[RT_HOIST_MAX][3];
	__shared__ int s_second[RT_HOIST_MAX];       // triangle index of the partner that follows, or -1
	const int j = threadIdx.x;
	float4* boxes = out + (size_t)k * RT_AOS_F4;
	int i = 0;
	if (j < k) {
		i = idx[j];
		keys[i] = (1ull << 63) | (u64)(unsigned int)i;
		aos_record(planes, stride, i, out + (size_t)j * RT_AOS_F4);
		out[(size_t)j * RT_AOS_F4 + 1].w = __int_as_float(i);      // (a hoisted record carries its triangle's index where P1's plane has a colour: hoisted_by_mask reads it with N and P1)
		float lo[3], hi[3];
		tri_box_padded(planes, stride, i, pad, lo, hi);
		for (int a = 0; a < 3; a++) { s_lo[j][a] = lo[a]; s_hi[j][a] = hi[a]; }
		s_second[j] = -1;
	}
	__syncthreads();
	if (j == 0) {                                  // greedy pairing, left to right (k <= RT_HOIST_MAX)
		for (int a = 0; a + 1 < k;) {
			bool same = true;
			for (int c = 0; c < 3; c++)
				same = same && fabsf(s_lo[a][c] - s_lo[a + 1][c]) <= pad && fabsf(s_hi[a][c] - s_hi[a + 1][c]) <= pad;
			if (same) {
				for (int c = 0; c < 3; c++) { s_lo[a][c] = fminf(s_lo[a][c], s_lo[a + 1][c]); s_hi[a][c] = fmaxf(s_hi[a][c], s_hi[a + 1][c]); }
				s_second[a] = idx[a + 1];
				a += 2;
			} else a++;
		}
	}
	__syncthreads();
	if (j >= k) return;
	for (int oct = 0; oct < 8; oct++) {
		float nr[3], fr[3];
		for (int a = 0; a < 3; a++) {
			const bool neg = (oct >> a) & 1;
			nr[a] = neg ? s_hi[j][a] : s_lo[j][a];
			fr[a] = neg ? s_lo[j][a] : s_hi[j][a];
		}
		boxes[((size_t)oct * k + j) * 2] = make_float4(nr[0], nr[1], nr[2], fr[0]);
		boxes[((size_t)oct * k + j) * 2 + 1] = make_float4(fr[1], fr[2], __int_as_float(s_second[j]), __int_as_float(i));
	}
}

// ---- RT_BUILD_PLOC: the tree by parallel locally-ordered clustering (Meister & Bittner 2018) ON the GPU ----
// Bottom-up over the Morton order the radix sort leaves: every cluster (at first: every triangle's padded box)
// looks RT_PLOC_R places to either side for the neighbour whose union with it has the smallest surface area; two
// clusters that choose each other merge into a new inner node; the survivors are compacted in order and the search
// repeats until one cluster — the root — is left.  The agglomeration sees the actual boxes, not the bits of a grid,
// so the tree is of binned-SAH quality (102 412 triangles: sum of child-box areas over the root's 8.83, host SAH build
// 8.73, Morton/Karras 10.70; scripts/ploc_proto.py is the CPU prototype these kernels follow) at GPU build speed.
//   * Ties go to the lower index, so the globally closest pair always chooses each other: every pass merges at
//     least one pair.  (A scene whose areas overflow to inf/NaN may stall; the pass budget then gives up and the
//     commit falls back to the Morton/Karras build.)
//   * A pass over more than RT_PLOC_TAIL clusters is two launches: k_ploc_search — a workgroup loads 256 consecutive
//     clusters to LDS, finds the neighbour of the middle 240, decides for its own 224 (keep / lead a merge / merged
//     away) and ranks them (wave ballots); k_ploc_merge — sums the ranks of the workgroups before it (no look-back
//     spinning: the launch boundary is the grid barrier), writes the compacted clusters to the other buffer and the
//     new nodes: both children's boxes and references, so no refit pass is needed.  Nodes are numbered from nt-2
//     DOWN in merge order, so the last merge — the root — is node 0, where every consumer of the tree expects it.
//   * From RT_PLOC_TAIL clusters on, ONE workgroup finishes the tree in LDS (k_ploc_tail): the last ~25 passes
//     cost a barrier each instead of two launches.
// Pass state (cluster count, merges so far, which buffer is current) lives on the device, one record per pass; the
// host reads it back every few passes only to size the next launches.
#ifndef RT_PLOC_R
#define RT_PLOC_R 8
#endif
#define RT_PLOC_T 256
#define RT_PLOC_C (RT_PLOC_T - 4 * RT_PLOC_R)
#define RT_PLOC_TAIL 1024
#define RT_PLOC_MAX_PASSES 384

struct PlocState { int n, merges, cur, fail; };

__device__ __forceinline__ float union_area(float4 alo, float4 ahi, float4 blo, float4 bhi) {
	const float x = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), y = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y), z = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
	return x * y + y * z + z * x;
}

__device__ __forceinline__ void write_merged_node(float4* __restrict__ nodes, int node, float4 alo, float4 ahi, float4 blo, float4 bhi) {
	float4* nd = nodes + (size_t)node * 4;
	nd[0] = make_float4(alo.x, alo.y, alo.z, ahi.x);
	nd[1] = make_float4(ahi.y, ahi.z, blo.x, blo.y);
	nd[2] = make_float4(blo.z, bhi.x, bhi.y, bhi.z);
	nd[3] = make_float4(alo.w, blo.w, 0.f, 0.f);                 // (the references travel in lo.w)
}

// leaves in Morton order -> the first cluster buffer: (lo.xyz, reference ~triangle), (hi.xyz, -)
__global__ void __launch_bounds__(256) k_ploc_init(const float4* __restrict__ planes, int stride, int nt, const u64* __restrict__ keys,
                                                  unsigned int index_mask, float pad, float4* __restrict__ lo, float4* __restrict__ hi,
                                                  PlocState* __restrict__ state) {
	const int p = blockIdx.x * blockDim.x + threadIdx.x;
	if (p == 0) { PlocState s; s.n = nt; s.merges = 0; s.cur = 0; s.fail = 0; state[0] = s; }
	if (p >= nt) return;
	const int tri = (int)((unsigned int)keys[p] & index_mask);
	float l[3], h[3];
	tri_box_padded(planes, stride, tri, pad, l, h);
	lo[p] = make_float4(l[0], l[1], l[2], __int_as_float(~tri));
	hi[p] = make_float4(h[0], h[1], h[2], 0.f);
}

// kind of a cluster after the search: 0 stays, 1 leads a merge (the lower index of a mutual pair), 2 merged away
__global__ void __launch_bounds__(RT_PLOC_T) k_ploc_search(const float4* __restrict__ lo0, const float4* __restrict__ hi0, const float4* __restrict__ lo1,
                                                          const float4* __restrict__ hi1, const PlocState* __restrict__ state, int pass,
                                                          int2* __restrict__ dec, int2* __restrict__ totals) {
	const PlocState st = state[pass];
	const int n = st.n;
	const int base = (int)blockIdx.x * RT_PLOC_C;
	if (n <= RT_PLOC_TAIL || base >= n || st.fail) return;
	const float4* __restrict__ lo = st.cur ? lo1 : lo0;
	const float4* __restrict__ hi = st.cur ? hi1 : hi0;
	__shared__ float4 s_lo[RT_PLOC_T], s_hi[RT_PLOC_T];
	__shared__ int s_nn[RT_PLOC_T];
	__shared__ int s_ws[RT_PLOC_T / 64], s_wm[RT_PLOC_T / 64];
	const int tid = (int)threadIdx.x;
	const int p = base - 2 * RT_PLOC_R + tid;
	const bool valid = p >= 0 && p < n;
	if (valid) { s_lo[tid] = lo[p]; s_hi[tid] = hi[p]; }
	__syncthreads();
	int nn = -1;
	if (valid && tid >= RT_PLOC_R && tid < RT_PLOC_T - RT_PLOC_R) {
		const float4 alo = s_lo[tid], ahi = s_hi[tid];
		float best = 0.f;
#pragma unroll
		for (int d = -RT_PLOC_R; d <= RT_PLOC_R; d++) {
			if (d == 0) continue;
			const int q = p + d;
			if (q < 0 || q >= n) continue;
			const float a = union_area(alo, ahi, s_lo[tid + d], s_hi[tid + d]);
			if (nn < 0 || a < best) { best = a; nn = q; }            // ascending q, strict <: the lower index on ties
		}
	}
	s_nn[tid] = nn;
	__syncthreads();
	const bool owned = valid && tid >= 2 * RT_PLOC_R && tid < RT_PLOC_T - 2 * RT_PLOC_R;
	int kind = 0;
	if (owned && nn >= 0 && s_nn[nn - (base - 2 * RT_PLOC_R)] == p) kind = p < nn ? 1 : 2;
	const bool survive = owned && kind != 2, lead = owned && kind == 1;
	const unsigned long long ms = __builtin_amdgcn_ballot_w64(survive), mm = __builtin_amdgcn_ballot_w64(lead);
	const int lane = tid & 63, wv = tid >> 6;
	const unsigned long long below = (1ull << lane) - 1ull;
	int rs = __popcll(ms & below), rm = __popcll(mm & below);
	if (lane == 0) { s_ws[wv] = __popcll(ms); s_wm[wv] = __popcll(mm); }
	__syncthreads();
	int ts = 0, tm = 0;
#pragma unroll
	for (int w = 0; w < RT_PLOC_T / 64; w++) {
		if (w < wv) { rs += s_ws[w]; rm += s_wm[w]; }
		ts += s_ws[w]; tm += s_wm[w];
	}
	if (owned) dec[p] = make_int2((rs << 2) | kind, (rm << 8) | (nn - p + 128));
	if (tid == 0) totals[blockIdx.x] = make_int2(ts, tm);
}

__global__ void __launch_bounds__(RT_PLOC_T) k_ploc_merge(float4* __restrict__ lo0, float4* __restrict__ hi0, float4* __restrict__ lo1, float4* __restrict__ hi1,
                                                         PlocState* __restrict__ state, int pass, const int2* __restrict__ dec,
                                                         const int2* __restrict__ totals, float4* __restrict__ nodes, int nt) {
	const PlocState st = state[pass];
	const int n = st.n;
	const int base = (int)blockIdx.x * RT_PLOC_C;
	const int tid = (int)threadIdx.x;
	if (n <= RT_PLOC_TAIL || st.fail) {                       // nothing to do in this pass (the tail kernel finishes): hand the state on
		if (blockIdx.x == 0 && tid == 0) state[pass + 1] = st;
		return;
	}
	if (base >= n) return;
	const float4* __restrict__ lo = st.cur ? lo1 : lo0;
	const float4* __restrict__ hi = st.cur ? hi1 : hi0;
	float4* __restrict__ olo = st.cur ? lo0 : lo1;
	float4* __restrict__ ohi = st.cur ? hi0 : hi1;
	// survivors and merges of the workgroups before this one
	__shared__ int s_s[RT_PLOC_T / 64], s_m[RT_PLOC_T / 64];
	int ps = 0, pm = 0;
	for (int k = tid; k < (int)blockIdx.x; k += RT_PLOC_T) { const int2 t = totals[k]; ps += t.x; pm += t.y; }
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { ps += __shfl_down(ps, off); pm += __shfl_down(pm, off); }
	if ((tid & 63) == 0) { s_s[tid >> 6] = ps; s_m[tid >> 6] = pm; }
	__syncthreads();
	ps = 0; pm = 0;
#pragma unroll
	for (int w = 0; w < RT_PLOC_T / 64; w++) { ps += s_s[w]; pm += s_m[w]; }
	const int n_blocks = (n + RT_PLOC_C - 1) / RT_PLOC_C;
	if ((int)blockIdx.x == n_blocks - 1 && tid == 0) {
		const int2 mine = totals[blockIdx.x];
		PlocState nx;
		nx.n = ps + mine.x; nx.merges = st.merges + pm + mine.y; nx.cur = st.cur ^ 1;
		nx.fail = (mine.y + pm) == 0;                          // no pair merged: the search cannot make progress (non-finite areas)
		state[pass + 1] = nx;
	}
	const int p = base + tid;
	if (tid >= RT_PLOC_C || p >= n) return;
	const int2 d = dec[p];
	const int kind = d.x & 3;
	if (kind == 2) return;
	const int pos = ps + (d.x >> 2);
	float4 alo = lo[p], ahi = hi[p];
	if (kind == 1) {
		const int q = p + ((d.y & 255) - 128);
		const float4 blo = lo[q], bhi = hi[q];
		const int node = nt - 2 - (st.merges + pm + (d.y >> 8));
		write_merged_node(nodes, node, alo, ahi, blo, bhi);
		alo = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), __int_as_float(node));
		ahi = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
	}
	olo[pos] = alo;
	ohi[pos] = ahi;
}

// the last RT_PLOC_TAIL clusters (or a whole small scene): one workgroup, clusters in LDS, a barrier per step
__global__ void __launch_bounds__(RT_PLOC_TAIL) k_ploc_tail(const float4* __restrict__ lo0, const float4* __restrict__ hi0, const float4* __restrict__ lo1,
                                                           const float4* __restrict__ hi1, PlocState* __restrict__ state, int pass,
                                                           float4* __restrict__ nodes, int nt) {
	__shared__ float4 s_lo[RT_PLOC_TAIL], s_hi[RT_PLOC_TAIL];
	__shared__ int s_nn[RT_PLOC_TAIL];
	__shared__ int s_ws[RT_PLOC_TAIL / 64], s_wm[RT_PLOC_TAIL / 64];
	PlocState st = state[pass];
	const int tid = (int)threadIdx.x;
	if (st.n > RT_PLOC_TAIL || st.fail) {                      // (the host only launches this once the count is small enough)
		if (tid == 0) { st.fail = 1; state[pass + 1] = st; }
		return;
	}
	int n = st.n, merges = st.merges;
	{
		const float4* __restrict__ lo = st.cur ? lo1 : lo0;
		const float4* __restrict__ hi = st.cur ? hi1 : hi0;
		if (tid < n) { s_lo[tid] = lo[tid]; s_hi[tid] = hi[tid]; }
	}
	__syncthreads();
	const int lane = tid & 63, wv = tid >> 6;
	const unsigned long long below = (1ull << lane) - 1ull;
	int fail = 0;
	while (n > 1) {
		const bool valid = tid < n;
		float4 alo = make_float4(0.f, 0.f, 0.f, 0.f), ahi = alo;
		int nn = -1;
		if (valid) {
			alo = s_lo[tid]; ahi = s_hi[tid];
			float best = 0.f;
			for (int d = -RT_PLOC_R; d <= RT_PLOC_R; d++) {
				const int q = tid + d;
				if (d == 0 || q < 0 || q >= n) continue;
				const float a = union_area(alo, ahi, s_lo[q], s_hi[q]);
				if (nn < 0 || a < best) { best = a; nn = q; }
			}
		}
		s_nn[tid] = nn;
		__syncthreads();
		int kind = 0;
		if (valid && nn >= 0 && s_nn[nn] == tid) kind = tid < nn ? 1 : 2;
		const bool survive = valid && kind != 2, lead = kind == 1;
		const unsigned long long ms = __builtin_amdgcn_ballot_w64(survive), mm = __builtin_amdgcn_ballot_w64(lead);
		int rs = __popcll(ms & below), rm = __popcll(mm & below);
		if (lane == 0) { s_ws[wv] = __popcll(ms); s_wm[wv] = __popcll(mm); }
		float4 blo = alo, bhi = ahi;
		if (lead) { blo = s_lo[nn]; bhi = s_hi[nn]; }
		__syncthreads();                                       // every read of the old clusters is done
		int ts = 0, tm = 0;
		for (int w = 0; w < RT_PLOC_TAIL / 64; w++) {
			if (w < wv) { rs += s_ws[w]; rm += s_wm[w]; }
			ts += s_ws[w]; tm += s_wm[w];
		}
		if (lead) {
			const int node = nt - 2 - (merges + rm);
			write_merged_node(nodes, node, alo, ahi, blo, bhi);
			alo = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), __int_as_float(node));
			ahi = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
		}
		if (survive) { s_lo[rs] = alo; s_hi[rs] = ahi; }
		__syncthreads();                                       // (also keeps s_ws / s_wm for the next step's writers)
		if (tm == 0) { fail = 1; break; }                      // wave-uniform and workgroup-uniform: everyone leaves together
		n = ts;
		merges += tm;
	}
	if (tid == 0) {
		PlocState nx;
		nx.n = n; nx.merges = merges; nx.cur = st.cur; nx.fail = fail;
		state[pass + 1] = nx;
	}
}

// Which triangles stay out of the tree: those whose vertex box has at least an eighth of the
// surface area of the scene's box.  A ray meets such a box with that probability or more at every
// level of the path from its leaf to the root, so leaving it in costs more box tests than the one
// triangle test per ray that hoisting costs.  At most RT_HOIST_MAX of them and a quarter of the
// scene (beyond that the scene is made of large triangles and the tree is what it is).
void pick_hoisted(const std::vector<float>& tris, const float bb_lo[3], const float bb_hi[3], std::vector<int>& out) {
	out.clear();
	const int n = (int)(tris.size() / RT_TRI_FLOATS);
#ifdef RT_EXPERIMENT
	if (std::getenv("RT355_NO_HOIST")) return;
#endif
	const double ex = (double)bb_hi[0] - bb_lo[0], ey = (double)bb_hi[1] - bb_lo[1], ez = (double)bb_hi[2] - bb_lo[2];
	const double scene_area = ex * ey + ey * ez + ez * ex;
	if (!(scene_area > 0.0) || !(scene_area < 1e300)) return;
	for (int i = 0; i < n; i++) {
		const float* v = tris.data() + (size_t)i * RT_TRI_FLOATS + 6;      // p1 p2 p3
		double e[3];
		for (int k = 0; k < 3; k++) {
			const float lo = fminf(v[k], fminf(v[3 + k], v[6 + k])), hi = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
			e[k] = (double)hi - (double)lo;
		}
		const double area = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
		if (area * 8.0 >= scene_area) {
			if ((int)out.size() == RT_HOIST_MAX) { out.clear(); return; }
			out.push_back(i);
		}
	}
	if ((long long)out.size() * 4 > n || n - (int)out.size() < 2) out.clear();
}

template <typename T>
int dev_alloc(T** p, size_t count) {
	hipError_t e = hipMalloc((void**)p, count * sizeof(T));
	if (e != hipSuccess) return hip_fail(e, "hipMalloc (LBVH build)");
	return RT_OK;
}

} // namespace

void lbvh_free(DeviceState& d) {
	if (d.bvh_nodes) (void)hipFree(d.bvh_nodes);
	if (d.bvh4) (void)hipFree(d.bvh4);
	if (d.tri_aos) (void)hipFree(d.tri_aos);
	if (d.skiplist) (void)hipFree(d.skiplist);
	if (d.hoist) (void)hipFree(d.hoist);
	d.hoist = nullptr;
	d.hoist_count = 0;
	d.bvh_node_count = 0;
	d.hoisted.clear();
	d.bvh_nodes = nullptr;
	d.bvh4 = nullptr;
	d.tri_aos = nullptr;
	d.skiplist = nullptr;
	d.skip_count = 0;
}

int lbvh_build(rt_scene* s, DeviceState& d, hipStream_t stream) {
	const int n = (int)(s->tris.size() / RT_TRI_FLOATS);
	if (n < 2) { set_error("LBVH needs at least two triangles"); return RT_ERR_LIMIT; }
	for (int k = 0; k < 3; k++)
		if (!(d.bb_hi[k] - d.bb_lo[k] < __builtin_huge_valf())) {
			set_error("LBVH: the scene has a non-finite vertex; commit it with RT_ACCEL_BRUTE");
			return RT_ERR_LIMIT;
		}
	// (the packet walk addresses its 8 octant copies of the node records, 512 B per node, as a 64-bit base + 32-bit byte offset)
	if ((unsigned long long)n * 512ull >= 0xffffffffull) { set_error("LBVH: %d triangles need packet records of 4 GiB or more", n); return RT_ERR_LIMIT; }
	lbvh_free(d);
	std::vector<int> hoisted;
	pick_hoisted(s->tris, d.bb_lo, d.bb_hi, hoisted);
	const int nh = (int)hoisted.size();
	const int nt = n - nh;                       // triangles in the tree
	const size_t hoist_float4s = (size_t)nh * (RT_AOS_F4 + 8 * 2);   // records, then 8 octants of boxes
	int* hoist_idx = nullptr;
	float4* hoist = nullptr;
	u64 *keys = nullptr, *keys_sorted = nullptr;
	int *parent_int = nullptr, *parent_leaf = nullptr;
	unsigned int* flags = nullptr;
	void* temp = nullptr;
	float4 *nodes = nullptr, *aos = nullptr, *pkt = nullptr;
	float4* cl[4] = { nullptr, nullptr, nullptr, nullptr };          // PLOC: cluster lo/hi, two buffers
	int2 *dec = nullptr, *totals = nullptr;
	PlocState* pstate = nullptr;
	uint4* nodes4 = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	int rc = RT_OK;
	auto cleanup = [&]() {
		if (keys) (void)hipFree(keys);
		if (keys_sorted) (void)hipFree(keys_sorted);
		if (parent_int) (void)hipFree(parent_int);
		if (parent_leaf) (void)hipFree(parent_leaf);
		if (flags) (void)hipFree(flags);
		if (temp) (void)hipFree(temp);
		if (hoist_idx) (void)hipFree(hoist_idx);
		for (int k = 0; k < 4; k++) if (cl[k]) (void)hipFree(cl[k]);
		if (dec) (void)hipFree(dec);
		if (totals) (void)hipFree(totals);
		if (pstate) (void)hipFree(pstate);
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
	};
	auto drop_results = [&]() {
		if (nodes) (void)hipFree(nodes);
		if (nodes4) (void)hipFree(nodes4);
		if (aos) (void)hipFree(aos);
		if (pkt) (void)hipFree(pkt);
		if (hoist) (void)hipFree(hoist);
	};
#define LB_TRY(x) do { rc = (x); if (rc != RT_OK) { cleanup(); drop_results(); return rc; } } while (0)
#define LB_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = hip_fail(e_, #x); cleanup(); drop_results(); return rc; } } while (0)
	const int ploc_blocks_max = (nt + RT_PLOC_C - 1) / RT_PLOC_C;
	LB_TRY(dev_alloc(&keys, (size_t)n));
	LB_TRY(dev_alloc(&keys_sorted, (size_t)n));
	LB_TRY(dev_alloc(&parent_int, (size_t)n));
	LB_TRY(dev_alloc(&parent_leaf, (size_t)n));
	LB_TRY(dev_alloc(&flags, (size_t)n));
	LB_TRY(dev_alloc(&nodes, (size_t)(nt - 1) * 4));
	LB_TRY(dev_alloc(&nodes4, (size_t)(nt - 1) * 4));
	LB_TRY(dev_alloc(&aos, (size_t)n * RT_AOS_F4));
	LB_TRY(dev_alloc(&pkt, (size_t)(nt - 1) * 4 * 8));
	if (s->build == RT_BUILD_PLOC) {
		for (int k = 0; k < 4; k++) LB_TRY(dev_alloc(&cl[k], (size_t)nt));
		LB_TRY(dev_alloc(&dec, (size_t)nt));
		LB_TRY(dev_alloc(&totals, (size_t)ploc_blocks_max));
		LB_TRY(dev_alloc(&pstate, (size_t)RT_PLOC_MAX_PASSES + 2));
	}
	if (nh > 0) {
		LB_TRY(dev_alloc(&hoist_idx, (size_t)nh));
		LB_TRY(dev_alloc(&hoist, hoist_float4s));
		LB_HIP(hipMemcpy(hoist_idx, hoisted.data(), (size_t)nh * sizeof(int), hipMemcpyHostToDevice));
	}
	size_t temp_bytes = 0;
	int index_bits = 1;
	while (index_bits < 31 && (1ll << index_bits) < (long long)n) index_bits++;
	const unsigned int index_mask = (unsigned int)((1ull << index_bits) - 1ull);
	const unsigned int key_bits = nh > 0 ? 64u : (unsigned int)((63 - index_bits) / 3 * 3 + index_bits);   // bit 63: hoisted
	LB_HIP(rocprim::radix_sort_keys(nullptr, temp_bytes, keys, keys_sorted, (size_t)n, 0, key_bits, stream));
	LB_HIP(hipMalloc(&temp, temp_bytes > 0 ? temp_bytes : 16));
	LB_HIP(hipEventCreate(&e0));
	LB_HIP(hipEventCreate(&e1));

	// Morton codes are taken over the box of the triangles that go into the tree
	BoxF scene;
	for (int k = 0; k < 3; k++) { scene.lo[k] = d.bb_lo[k]; scene.hi[k] = d.bb_hi[k]; }
	if (nh > 0) {
		for (int k = 0; k < 3; k++) { scene.lo[k] = __builtin_huge_valf(); scene.hi[k] = -__builtin_huge_valf(); }
		size_t next = 0;
		for (int i = 0; i < n; i++) {
			if (next < hoisted.size() && hoisted[next] == i) { next++; continue; }
			const float* v = s->tris.data() + (size_t)i * RT_TRI_FLOATS + 6;
			for (int k = 0; k < 3; k++) {
				scene.lo[k] = fminf(scene.lo[k], fminf(v[k], fminf(v[3 + k], v[6 + k])));
				scene.hi[k] = fmaxf(scene.hi[k], fmaxf(v[k], fmaxf(v[3 + k], v[6 + k])));
			}
		}
	}
	const float pad = d.pad;
	const float4* planes = (const float4*)d.planes;
	const unsigned int blocks_n = (unsigned int)((n + 255) / 256);
	const unsigned int blocks_t = (unsigned int)((nt + 255) / 256);

	float ms = 0.f, host_ms = 0.f;
	int depth2 = 0, depth4 = 0;
	int built = s->build;                            // what the tree was really built by (PLOC may fall back)
	for (int attempt = 0; attempt < 2; attempt++) {
		const bool sah = built == RT_BUILD_SAH, ploc = built == RT_BUILD_PLOC;
		std::vector<u64> h_keys;
		std::vector<int> h_child, h_pint, h_pleaf, h_nodes;
		host_ms = 0.f;
		if (sah) {
			const auto t0 = std::chrono::steady_clock::now();
			try { sah_topology(s->tris, hoisted, nt, h_keys, h_child, h_pint, h_pleaf); }
			catch (...) { set_error("LBVH (SAH build): out of host memory or threads"); rc = RT_ERR_NOMEM; }
			LB_TRY(rc);
			// whole nodes with only the child references set (k_refit fills the boxes): one contiguous upload
			try {
				h_nodes.assign((size_t)(nt - 1) * 16, 0);
				for (size_t i = 0; i < (size_t)(nt - 1); i++) { h_nodes[i * 16 + 12] = h_child[i * 2]; h_nodes[i * 16 + 13] = h_child[i * 2 + 1]; }
			} catch (...) { set_error("LBVH (SAH build): out of host memory"); rc = RT_ERR_NOMEM; }
			LB_TRY(rc);
			host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
		}
		LB_HIP(hipEventRecord(e0, stream));
		if (!ploc) {
			LB_HIP(hipMemsetAsync(flags, 0, (size_t)n * sizeof(unsigned int), stream));
			LB_HIP(hipMemsetAsync(nodes, 0, (size_t)(nt - 1) * 4 * sizeof(float4), stream));
		}
		hipLaunchKernelGGL(k_morton, dim3(blocks_n), dim3(256), 0, stream, planes, d.stride, n, scene, index_bits, keys);
		LB_HIP(hipGetLastError());
		if (nh > 0) {
			hipLaunchKernelGGL(k_hoist, dim3(1), dim3(64), 0, stream, planes, d.stride, (const int*)hoist_idx, nh, pad, keys, hoist);
			LB_HIP(hipGetLastError());
		}
		unsigned int leaf_mask = index_mask;
		bool ploc_failed = false;
		if (sah) {
			// the host's topology in place of sort + k_karras: leaf order, child references (the fourth float4 of a node), parent links
			LB_HIP(hipMemcpyAsync(keys_sorted, h_keys.data(), (size_t)nt * sizeof(u64), hipMemcpyHostToDevice, stream));
			LB_HIP(hipMemcpyAsync(nodes, h_nodes.data(), (size_t)(nt - 1) * 64, hipMemcpyHostToDevice, stream));
			LB_HIP(hipMemcpyAsync(parent_int, h_pint.data(), (size_t)nt * sizeof(int), hipMemcpyHostToDevice, stream));
			LB_HIP(hipMemcpyAsync(parent_leaf, h_pleaf.data(), (size_t)nt * sizeof(int), hipMemcpyHostToDevice, stream));
			leaf_mask = 0xffffffffu;
		} else {
			LB_HIP(rocprim::radix_sort_keys(temp, temp_bytes, keys, keys_sorted, (size_t)n, 0, key_bits, stream));
			// (the hoisted keys sorted to the tail: the tree is built over the first nt)
		}
		if (ploc) {
			hipLaunchKernelGGL(k_ploc_init, dim3(blocks_t), dim3(256), 0, stream, planes, d.stride, nt, (const u64*)keys_sorted, index_mask, pad,
			                   cl[0], cl[1], pstate);
			LB_HIP(hipGetLastError());
			int pass = 0, n_upper = nt;
			PlocState hs = { nt, 0, 0, 0 };
			while (n_upper > RT_PLOC_TAIL && !ploc_failed) {
				if (pass >= RT_PLOC_MAX_PASSES) { ploc_failed = true; break; }
				const int batch = RT_PLOC_MAX_PASSES - pass < 8 ? RT_PLOC_MAX_PASSES - pass : 8;
				const dim3 grid((unsigned int)((n_upper + RT_PLOC_C - 1) / RT_PLOC_C));
				for (int b = 0; b < batch; b++, pass++) {
					hipLaunchKernelGGL(k_ploc_search, grid, dim3(RT_PLOC_T), 0, stream, (const float4*)cl[0], (const float4*)cl[1], (const float4*)cl[2],
					                   (const float4*)cl[3], (const PlocState*)pstate, pass, dec, totals);
					hipLaunchKernelGGL(k_ploc_merge, grid, dim3(RT_PLOC_T), 0, stream, cl[0], cl[1], cl[2], cl[3], pstate, pass, (const int2*)dec,
					                   (const int2*)totals, nodes, nt);
				}
				LB_HIP(hipGetLastError());
				LB_HIP(hipMemcpyAsync(&hs, pstate + pass, sizeof hs, hipMemcpyDeviceToHost, stream));
				LB_HIP(hipStreamSynchronize(stream));
				if (hs.fail || hs.n < 1 || hs.n > n_upper) ploc_failed = true;
				n_upper = hs.n;
			}
			if (!ploc_failed) {
				hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(RT_PLOC_TAIL), 0, stream, (const float4*)cl[0], (const float4*)cl[1], (const float4*)cl[2],
				                   (const float4*)cl[3], pstate, pass, nodes, nt);
				LB_HIP(hipGetLastError());
				pass++;
				LB_HIP(hipMemcpyAsync(&hs, pstate + pass, sizeof hs, hipMemcpyDeviceToHost, stream));
				LB_HIP(hipStreamSynchronize(stream));
				if (hs.fail || hs.n != 1 || hs.merges != nt - 1) ploc_failed = true;
			}
		} else {
			if (!sah) {
				hipLaunchKernelGGL(k_karras, dim3(blocks_t), dim3(256), 0, stream, keys_sorted, nt, index_mask, nodes, parent_int, parent_leaf);
				LB_HIP(hipGetLastError());
			}
			hipLaunchKernelGGL(k_refit, dim3(blocks_t), dim3(256), 0, stream, planes, d.stride, nt, keys_sorted, leaf_mask, (float*)nodes,
			                   parent_int, parent_leaf, flags, pad);
			LB_HIP(hipGetLastError());
		}
		if (ploc_failed) { built = RT_BUILD_FAST; continue; }      // clustering made no progress (areas that overflow): the Morton/Karras tree instead
		hipLaunchKernelGGL(k_collapse4, dim3(blocks_t), dim3(256), 0, stream, (const float4*)nodes, nt - 1, nodes4);
		LB_HIP(hipGetLastError());
		hipLaunchKernelGGL(k_aos, dim3(blocks_n), dim3(256), 0, stream, planes, d.stride, n, aos);
		LB_HIP(hipGetLastError());
		hipLaunchKernelGGL(k_packet, dim3(blocks_t), dim3(256), 0, stream, (const float4*)nodes, nt - 1, pkt);
		LB_HIP(hipGetLastError());
		LB_HIP(hipEventRecord(e1, stream));
		LB_HIP(hipEventSynchronize(e1));
		LB_HIP(hipEventElapsedTime(&ms, e0, e1));
		if (!ploc) {
			// the root must have been reached by exactly two arrivals
			unsigned int root_flag = 0;
			LB_HIP(hipMemcpy(&root_flag, flags, sizeof root_flag, hipMemcpyDeviceToHost));
			if (root_flag != 2) {
				set_error("LBVH refit did not reach the root (flag %u)", root_flag);
				cleanup();
				drop_results();
				return RT_ERR_DEVICE;
			}
		}
		// Depths of the binary tree and of its 4-wide collapse (what the traversal stacks have to hold: rt_lbvh_trav.inl header),
		// measured on the device — reading both trees back for a host walk cost 2 ms of a commit's 12 at 100k triangles
		{
			int* dscratch = nullptr;                                   // levels (nt - 1), then the deepest level reached
			LB_TRY(dev_alloc(&dscratch, (size_t)(nt - 1) + 1));
			int h_depth[2] = { 0, 0 };
			const unsigned int nb = (unsigned int)((nt - 1 + 255) / 256);
			const int one = 1;
			hipError_t de = hipSuccess;
			for (int which = 0; which < 2 && de == hipSuccess; which++) {
				// binary node: q3 = (left, right, 0, 0) at words 12, 13; 4-wide node: w1 = the 4 child references at words 4..7
				const int* words = which == 0 ? (const int*)nodes : (const int*)nodes4;
				const int first = which == 0 ? 12 : 4, width = which == 0 ? 2 : 4;
				de = hipMemsetAsync(dscratch, 0, ((size_t)(nt - 1) + 1) * sizeof(int), stream);
				if (de == hipSuccess) de = hipMemcpyAsync(dscratch, &one, sizeof one, hipMemcpyHostToDevice, stream);               // the root is level 1
				if (de == hipSuccess) de = hipMemcpyAsync(dscratch + (nt - 1), &one, sizeof one, hipMemcpyHostToDevice, stream);
				int level = 1, deepest = 1;
				while (de == hipSuccess) {
					const int batch = which == 0 ? 32 : 16;              // levels per read-back
					for (int k = 0; k < batch; k++)
						hipLaunchKernelGGL(k_level, dim3(nb), dim3(256), 0, stream, words, nt - 1, first, width, level + k, dscratch, dscratch + (nt - 1));
					level += batch;
					de = hipGetLastError();
					if (de == hipSuccess) de = hipMemcpyAsync(&deepest, dscratch + (nt - 1), sizeof deepest, hipMemcpyDeviceToHost, stream);
					if (de == hipSuccess) de = hipStreamSynchronize(stream);
					if (deepest < level) break;                           // the last level of the batch had no inner children
					if (level > 4096) { deepest = -1; break; }            // not a tree
				}
				h_depth[which] = deepest;
			}
			(void)hipFree(dscratch);
			if (de != hipSuccess) { rc = hip_fail(de, "tree depth"); }
			depth2 = h_depth[0];
			depth4 = h_depth[1];
		}
		LB_TRY(rc);
		if (ploc && depth2 > 63 && depth4 >= 1) { built = RT_BUILD_FAST; continue; }    // an agglomerated chain deeper than the packet stack: Karras' tree is at most 63 deep
		break;
	}
	if (depth2 < 1 || depth4 < 1 || depth2 > 63) {
		// (63-bit unique keys bound the Karras tree at 63 levels, RT_SAH_MEDIAN_DEPTH the SAH tree: the packet stack has 64 entries)
		set_error("LBVH: malformed tree (binary depth %d, 4-wide depth %d)", depth2, depth4);
		cleanup();
		drop_results();
		return RT_ERR_DEVICE;
	}
#undef LB_TRY
#undef LB_HIP
	cleanup();
	d.bvh_nodes = nodes;
	d.bvh4 = nodes4;
	d.tri_aos = aos;
	d.skiplist = pkt;
	d.skip_count = pkt ? nt - 1 : 0;
	d.hoist = hoist;
	d.hoist_count = nh;
	d.bvh_node_count = nt - 1;
	d.hoisted = hoisted;
	d.bvh_root = 0;
	d.bvh_depth = depth2;
	d.bvh4_depth = depth4;
	d.per_lane_ok = 3 * depth4 <= RT_STACK_LIMIT;
	d.build_ms = ms + host_ms;
	d.built_by = built;
	return RT_OK;
}

} // namespace rt
