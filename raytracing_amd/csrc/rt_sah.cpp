// rt_sah.cpp — host side of the RT_BUILD_SAH tree build (rt_scene_set_build): the topology only; boxes and
// everything derived from them stay on the GPU (rt_lbvh.hip).
#include "rt_internal.h"

#include <cmath>
#include <functional>
#include <thread>

namespace rt {

typedef unsigned long long u64;

namespace {
// ---- the tree's topology by binned surface-area heuristic, on the host (rt_scene_set_build RT_BUILD_SAH) ----
// The Morton/Karras tree (rt_lbvh.hip) is built in half a millisecond but splits space at the planes of a fixed
// grid; a top-down build that picks, per node, the plane that minimises area(L) * count(L) + area(R) * count(R)
// over 16 bins on each axis (Wald 2007) gives a tree a ray crosses in fewer nodes (sum of child-box areas over
// the root's: 8.7 instead of 10.7 for the 102k-triangle scene) at the price of 15-45 ms of host time per 100k
// triangles and commit — the choice for a scene that is committed once and rendered for many frames.
// Only the TOPOLOGY is made here, in the arrays k_karras writes on the GPU: child references, parent links and
// the leaf order; boxes, the 4-wide collapse, packet records and hoisting are rt_lbvh.hip's kernels either way.
// Numbering: inner nodes in depth-first preorder (a subtree of m leaves owns m-1 consecutive node indices, its
// leaves m consecutive positions), so both halves of a split know their index ranges before they are built and
// large subtrees are built on separate threads.  Depth: below RT_SAH_MEDIAN_DEPTH levels splits fall back to
// the median of the leaf order, so the depth is at most that + ceil(log2 n) <= 63 for n < 2^23.
#define RT_SAH_BINS 16
#define RT_SAH_MEDIAN_DEPTH 40
struct SahCtx {
	const float* lo;          // nt x 3: vertex boxes of the tree's triangles, in tree-local numbering
	const float* hi;
	const int* tri;           // tree-local number -> triangle index
	int* order;               // leaf position -> tree-local number (partitioned in place)
	int* child;               // 2 per inner node
	int* parent_int;
	int* parent_leaf;
};

inline float half_area(const float l[3], const float h[3]) {
	const float x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
	return x * y + y * z + z * x;
}

// Splits order[a, b) in place; returns m with a < m < b.
int sah_split(const SahCtx& c, int a, int b, int depth) {
	const int cnt = b - a;
	if (cnt == 2) return a + 1;
	int* ord = c.order;
	if (depth < RT_SAH_MEDIAN_DEPTH) {
		float clo[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() }, chi[3] = { -clo[0], -clo[0], -clo[0] };
		for (int p = a; p < b; p++) {
			const float* l = c.lo + (size_t)ord[p] * 3; const float* h = c.hi + (size_t)ord[p] * 3;
			for (int k = 0; k < 3; k++) { const float m = 0.5f * (l[k] + h[k]); clo[k] = fminf(clo[k], m); chi[k] = fmaxf(chi[k], m); }
		}
		float best = __builtin_huge_valf();
		int best_axis = -1, best_bin = 0;
		float best_scale = 0.f;
		for (int k = 0; k < 3; k++) {
			const float ext = chi[k] - clo[k];
			if (!(ext > 0.f) || !(ext < __builtin_huge_valf())) continue;
			const float scale = (float)RT_SAH_BINS / ext;
			int n[RT_SAH_BINS] = { 0 };
			float bl[RT_SAH_BINS][3], bh[RT_SAH_BINS][3];
			for (int i = 0; i < RT_SAH_BINS; i++) for (int q = 0; q < 3; q++) { bl[i][q] = __builtin_huge_valf(); bh[i][q] = -__builtin_huge_valf(); }
			for (int p = a; p < b; p++) {
				const float* l = c.lo + (size_t)ord[p] * 3; const float* h = c.hi + (size_t)ord[p] * 3;
				int bin = (int)((0.5f * (l[k] + h[k]) - clo[k]) * scale);
				bin = bin < 0 ? 0 : (bin >= RT_SAH_BINS ? RT_SAH_BINS - 1 : bin);
				n[bin]++;
				for (int q = 0; q < 3; q++) { bl[bin][q] = fminf(bl[bin][q], l[q]); bh[bin][q] = fmaxf(bh[bin][q], h[q]); }
			}
			float right_area[RT_SAH_BINS];
			int right_n[RT_SAH_BINS];
			float l3[3] = { __builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf() }, h3[3] = { -l3[0], -l3[0], -l3[0] };
			int acc = 0;
			for (int i = RT_SAH_BINS - 1; i >= 1; i--) {
				if (n[i]) for (int q = 0; q < 3; q++) { l3[q] = fminf(l3[q], bl[i][q]); h3[q] = fmaxf(h3[q], bh[i][q]); }
				acc += n[i];
				right_n[i] = acc;
				right_area[i] = acc ? half_area(l3, h3) : 0.f;
			}
			for (int q = 0; q < 3; q++) { l3[q] = __builtin_huge_valf(); h3[q] = -__builtin_huge_valf(); }
			acc = 0;
			for (int i = 0; i + 1 < RT_SAH_BINS; i++) {          // split after bin i
				if (n[i]) for (int q = 0; q < 3; q++) { l3[q] = fminf(l3[q], bl[i][q]); h3[q] = fmaxf(h3[q], bh[i][q]); }
				acc += n[i];
				if (acc == 0 || right_n[i + 1] == 0) continue;
				const float cost = half_area(l3, h3) * (float)acc + right_area[i + 1] * (float)right_n[i + 1];
				if (cost < best) { best = cost; best_axis = k; best_bin = i; best_scale = scale; }
			}
		}
		if (best_axis >= 0) {
			const int k = best_axis;
			int i = a, j = b - 1;
			while (i <= j) {
				const float* l = c.lo + (size_t)ord[i] * 3; const float* h = c.hi + (size_t)ord[i] * 3;
				int bin = (int)((0.5f * (l[k] + h[k]) - clo[k]) * best_scale);
				bin = bin < 0 ? 0 : (bin >= RT_SAH_BINS ? RT_SAH_BINS - 1 : bin);
				if (bin <= best_bin) i++;
				else { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; j--; }
			}
			if (i > a && i < b) return i;
		}
	}
	return a + cnt / 2;          // coincident centres, non-finite costs or a very deep path: halve the run as it stands
}

void sah_range(const SahCtx& c, int node, int a, int b, int depth, int link, int threads_left) {
	c.parent_int[node] = link;
	const int m = sah_split(c, a, b, depth);
	const int left_n = m - a, right_n = b - m;
	const int right_node = node + left_n;        // (the left subtree owns left_n - 1 inner nodes after this one)
	if (left_n == 1) { c.child[2 * node] = ~c.tri[c.order[a]]; c.parent_leaf[a] = node << 1; }
	else c.child[2 * node] = node + 1;
	if (right_n == 1) { c.child[2 * node + 1] = ~c.tri[c.order[m]]; c.parent_leaf[m] = (node << 1) | 1; }
	else c.child[2 * node + 1] = right_node;
	if (left_n > 1 && right_n > 1 && threads_left > 1 && left_n + right_n > 16384) {
		std::thread t;
		bool spawned = false;
		try { t = std::thread(sah_range, std::cref(c), node + 1, a, m, depth + 1, node << 1, threads_left / 2); spawned = true; } catch (...) { }
		if (spawned) {                               // (no thread to be had: both halves on this one, below)
			sah_range(c, right_node, m, b, depth + 1, (node << 1) | 1, threads_left - threads_left / 2);
			t.join();
			return;
		}
	}
	if (left_n > 1) sah_range(c, node + 1, a, m, depth + 1, node << 1, 1);
	if (right_n > 1) sah_range(c, right_node, m, b, depth + 1, (node << 1) | 1, 1);
}

} // namespace

// Fills keys (leaf position -> triangle index), child references and parent links for the nt triangles that are
// not in `hoisted` (ascending).
void sah_topology(const std::vector<float>& tris, const std::vector<int>& hoisted, int nt, std::vector<u64>& keys,
                  std::vector<int>& child, std::vector<int>& parent_int, std::vector<int>& parent_leaf) {
	const int n = (int)(tris.size() / RT_TRI_FLOATS);
	std::vector<float> lo((size_t)nt * 3), hi((size_t)nt * 3);
	std::vector<int> tri(nt), order(nt);
	size_t next = 0;
	int j = 0;
	for (int i = 0; i < n; i++) {
		if (next < hoisted.size() && hoisted[next] == i) { next++; continue; }
		const float* v = tris.data() + (size_t)i * RT_TRI_FLOATS + 6;
		for (int k = 0; k < 3; k++) {
			lo[(size_t)j * 3 + k] = fminf(v[k], fminf(v[3 + k], v[6 + k]));
			hi[(size_t)j * 3 + k] = fmaxf(v[k], fmaxf(v[3 + k], v[6 + k]));
		}
		tri[j] = i;
		order[j] = j;
		j++;
	}
	child.assign((size_t)(nt - 1) * 2, 0);
	parent_int.assign((size_t)nt, -1);
	parent_leaf.assign((size_t)nt, -1);
	SahCtx c = { lo.data(), hi.data(), tri.data(), order.data(), child.data(), parent_int.data(), parent_leaf.data() };
	unsigned int hw = std::thread::hardware_concurrency();
	sah_range(c, 0, 0, nt, 1, -1, (int)(hw < 1 ? 1 : (hw > 16 ? 16 : hw)));
	keys.resize((size_t)nt);
	for (int p = 0; p < nt; p++) keys[p] = (u64)(unsigned int)tri[order[p]];
}


} // namespace rt

// Test hook (not part of include/rt355.h): the topology for a committed-or-not scene's triangles, so that the CPU
// suite can check it without a GPU.  hoisted: ascending triangle indices left out; arrays sized by the caller:
// keys nt, child 2 * (nt - 1), parent_int nt, parent_leaf nt with nt = n - n_hoisted.
extern "C" int rt_internal_sah_topology(const rt_scene* s, const int* hoisted, int n_hoisted, unsigned long long* keys, int* child,
                                        int* parent_int, int* parent_leaf) {
	if (!s || !keys || !child || !parent_int || !parent_leaf || n_hoisted < 0) return RT_ERR_INVALID;
	const int n = (int)(s->tris.size() / RT_TRI_FLOATS), nt = n - n_hoisted;
	if (nt < 2) return RT_ERR_LIMIT;
	try {
		std::vector<int> h(hoisted, hoisted + n_hoisted);
		std::vector<rt::u64> k;
		std::vector<int> c, pi, pl;
		rt::sah_topology(s->tris, h, nt, k, c, pi, pl);
		std::copy(k.begin(), k.end(), keys);
		std::copy(c.begin(), c.end(), child);
		std::copy(pi.begin(), pi.end(), parent_int);
		std::copy(pl.begin(), pl.end(), parent_leaf);
	} catch (...) { return RT_ERR_NOMEM; }
	return RT_OK;
}
