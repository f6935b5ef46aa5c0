// rt_build.hip -- device build of the float record stream (see rt_build.h).
#include "rt_build.h"

#include <hipcub/hipcub.hpp>
#include <cstdlib>

namespace rtamd {
namespace {

constexpr int TPB = 256;

__device__ inline uint32_t spread10(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// key = 30-bit Morton code of the leaf's centroid | leaf index: unique, so the radix tree needs no tie handling
__global__ void keys_kernel(const BuildLeaf *L, int n, float lx, float ly, float lz, float sx, float sy, float sz, unsigned long long *keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BuildLeaf l = L[i];
    auto cell = [](float lo, float hi, float base, float scale) {
        float q = (0.5f * (lo + hi) - base) * scale;
        q = q != q ? 0.0f : q; // NaN centroid (an unbounded leaf): first cell
        return (uint32_t)fminf(fmaxf(q, 0.0f), 1023.0f);
    };
    const uint32_t m = (spread10(cell(l.lo[0], l.hi[0], lx, sx)) << 2) | (spread10(cell(l.lo[1], l.hi[1], ly, sy)) << 1) |
                       spread10(cell(l.lo[2], l.hi[2], lz, sz));
    keys[i] = ((unsigned long long)m << 32) | (unsigned)i;
}

struct Tree { // n - 1 internal nodes over n sorted leaves; a child c >= 0 is internal node c, c < 0 is sorted leaf ~c
    int *left, *right, *parent_i, *parent_l; // parent of internal node / of sorted leaf
    int *flag;                               // arrivals at an internal node (bottom-up pass)
    int *size, *count, *first_is_right;      // units of the subtree's records WITHOUT a BOX record of its own, its leaves, emission order of the children
    int *boxed;                              // the node gets a BOX record (decided when its parent is fitted)
    float *lo, *hi;                          // 3 floats per internal node
};

__device__ inline int delta(const unsigned long long *k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(k[i] ^ k[j]));
}

// Karras, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees" (HPG 2012), section 3
__global__ void tree_kernel(const unsigned long long *k, int n, Tree t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(k, n, i, i + 1) - delta(k, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(k, n, i, i - d);
    int lmax = 2;
    while (delta(k, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (delta(k, n, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(k, n, i, j);
    int s = 0, step = l;
    do {
        step = (step + 1) >> 1;
        if (delta(k, n, i, i + (s + step) * d) > dnode) s += step;
    } while (step > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lc = lo == gamma ? ~gamma : gamma, rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    t.left[i] = lc;
    t.right[i] = rc;
    if (lc < 0) t.parent_l[~lc] = i; else t.parent_i[lc] = i;
    if (rc < 0) t.parent_l[~rc] = i; else t.parent_i[rc] = i;
    if (i == 0) t.parent_i[0] = -1;
}

__device__ inline float box_area(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// bottom-up: the second thread to arrive at a node finds both children complete and fits the node
__global__ void fit_kernel(const unsigned long long *k, const BuildLeaf *L, int n, Tree t, int order, float thr2, float thrn) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int node = t.parent_l[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&t.flag[node], 1) == 0) return;
        __threadfence();
        float lo[2][3], hi[2][3];
        int size[2], count[2];
        for (int c = 0; c < 2; c++) {
            const int ch = c == 0 ? t.left[node] : t.right[node];
            if (ch < 0) {
                const BuildLeaf l = L[(unsigned)k[~ch]];
                for (int a = 0; a < 3; a++) { lo[c][a] = l.lo[a]; hi[c][a] = l.hi[a]; }
                size[c] = l.n;
                count[c] = 1;
            } else {
                for (int a = 0; a < 3; a++) { lo[c][a] = t.lo[3 * ch + a]; hi[c][a] = t.hi[3 * ch + a]; }
                size[c] = t.size[ch];
                count[c] = t.count[ch];
            }
        }
        float nlo[3], nhi[3];
        for (int a = 0; a < 3; a++) {
            nlo[a] = fminf(lo[0][a], lo[1][a]);
            nhi[a] = fmaxf(hi[0][a], hi[1][a]);
            t.lo[3 * node + a] = nlo[a];
            t.hi[3 * node + a] = nhi[a];
        }
        // A child whose box is nearly as large as this node's gets no BOX record of its own: its children stand in this node's
        // sequence (the host build's rule, rt_flatten.h BuildKnobs thr2 / thrn, against the direct parent's box).
        const float a0 = box_area(lo[0], hi[0]), a1 = box_area(lo[1], hi[1]), an = box_area(nlo, nhi);
        int tot[2];
        for (int c = 0; c < 2; c++) {
            const int ch = c == 0 ? t.left[node] : t.right[node];
            tot[c] = size[c];
            if (ch >= 0) {
                const bool b = (c == 0 ? a0 : a1) < (count[c] == 2 ? thr2 : thrn) * an;
                t.boxed[ch] = b ? 1 : 0;
                tot[c] += b ? 2 : 0;
            }
        }
        t.size[node] = tot[0] + tot[1];
        t.count[node] = count[0] + count[1];
        // the child with the larger box area per leaf goes first (rt_flatten.h, BuildKnobs)
        t.first_is_right[node] = order == 1 ? (a1 > a0 ? 1 : 0) : order == 3 ? (count[1] < count[0] ? 1 : 0) : (a1 * (float)count[0] > a0 * (float)count[1] ? 1 : 0);
        if (t.parent_i[node] < 0) t.boxed[node] = 1; // the root of the tree: the leaves that span the scene stand in front of it
        node = t.parent_i[node];
    }
}

// where a node's records start: two units per boxed ancestor, plus the first child's subtree wherever the path goes
// through a second child
__device__ inline int offset_of(const unsigned long long *k, const BuildLeaf *L, const Tree &t, int child /* encoded */, int parent) {
    int acc = 0;
    while (parent >= 0) {
        acc += t.boxed[parent] ? 2 : 0;
        const int first = t.first_is_right[parent] ? t.right[parent] : t.left[parent];
        if (first != child) acc += first < 0 ? L[(unsigned)k[~first]].n : t.size[first] + (t.boxed[first] ? 2 : 0);
        child = parent;
        parent = t.parent_i[parent];
    }
    return acc;
}

// units of a record by its kind (rt_scene.h)
__device__ inline int record_units(int kind) {
    return kind == K_QUAD ? 5 : (kind == K_TRI || kind == K_MSPHERE) ? 3 : kind == K_OBOX ? 4 : 2;
}
// Records moved from unit `from` of the leaf stream to unit `to` of the finished one: the positions they hold -- skip links of
// the BOX records inside an instance, a MEDIUM_GEN's sub-stream -- move along.  sub_from / sub_to: where the boundary
// sub-streams stood / stand (units).
__device__ inline void relocate(UnitT<float> *stream, int to, int n, int from, int sub_from, int sub_to) {
    const int U = (int)sizeof(UnitT<float>);
    for (int u = 0; u + 1 < n;) {
        const int kind = dec_int(stream[to + u + 1].w[3]) & 15;
        if (kind == K_BOX) stream[to + u + 1].w[2] = enc_int(dec_int(stream[to + u + 1].w[2]) + (to - from) * U, 0.0f);
        else if (kind == K_MEDIUM_GEN) stream[to + u].w[0] = enc_int(dec_int(stream[to + u].w[0]) + (sub_to - sub_from) * U, 0.0f);
        u += kind == K_READY ? 2 : record_units(kind);
    }
}

__global__ void emit_kernel(const unsigned long long *k, const BuildLeaf *L, int n, Tree t, const UnitT<float> *leaf_units,
                            const UnitT<float> *leaf_cold, UnitT<float> *stream, UnitT<float> *cold, float box_pad, int link_base,
                            int sub_from, int sub_units) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const UnitT<float> zero{{0.0f, 0.0f, 0.0f, 0.0f}};
    const UnitT<float> no_node{{0.0f, 0.0f, 0.0f, enc_int(-1, 0.0f)}};
    // the tree's records, then the sentinel, then the boundary sub-streams of general media
    const int tree_units = n > 1 ? t.size[0] + 2 : L[(unsigned)k[0]].n;
    const int sub_to = link_base + tree_units + 2; // as a position in the WHOLE stream (stream[] starts link_base units into it)
    if (g < n) { // sorted leaf g: its record(s); positions inside move with them
        const BuildLeaf l = L[(unsigned)k[g]];
        const int o = n > 1 ? offset_of(k, L, t, ~g, t.parent_l[g]) : 0;
        for (int u = 0; u < l.n; u++) {
            stream[o + u] = leaf_units[l.off + u];
            cold[o + u] = leaf_cold[l.off + u];
        }
        if (l.n > 2) relocate(stream, o, l.n, l.off - link_base, sub_from - link_base, sub_to - link_base);
        else if ((dec_int(stream[o + 1].w[3]) & 15) == K_MEDIUM_GEN) relocate(stream, o, l.n, l.off - link_base, sub_from - link_base, sub_to - link_base);
    } else if (g < 2 * n - 1) { // internal node: a BOX record, if it has one
        const int i = g - n;
        if (!t.boxed[i]) return;
        const int o = offset_of(k, L, t, i, t.parent_i[i]);
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const float pad = (t.hi[3 * i + a] - t.lo[3 * i + a]) < 2.0f * box_pad ? box_pad : 0.0f;
            lo[a] = t.lo[3 * i + a] - pad;
            hi[a] = t.hi[3 * i + a] + pad;
        }
        stream[o] = UnitT<float>{{lo[0], hi[0], lo[1], hi[1]}};
        stream[o + 1] = UnitT<float>{{lo[2], hi[2], enc_int((link_base + o + 2 + t.size[i]) * (int)sizeof(UnitT<float>), 0.0f), enc_int(make_hdr(K_BOX, 0), 0.0f)}};
        cold[o] = no_node;
        cold[o + 1] = zero;
    } else if (g == 2 * n - 1) { // the sentinel, and behind it the sub-streams (a few records: one thread)
        stream[tree_units] = zero;
        stream[tree_units + 1] = UnitT<float>{{0.0f, 0.0f, 0.0f, enc_int(make_hdr(K_READY, 0), 0.0f)}};
        cold[tree_units] = no_node;
        cold[tree_units + 1] = zero;
        for (int u = 0; u < sub_units; u++) {
            stream[tree_units + 2 + u] = leaf_units[sub_from + u];
            cold[tree_units + 2 + u] = leaf_cold[sub_from + u];
        }
        if (sub_units > 0) relocate(stream, tree_units + 2, sub_units, sub_from - link_base, sub_from - link_base, sub_to - link_base);
    }
}

} // namespace

hipError_t build_stream_on_device(const BuildLeaf *d_leaves, int n, const UnitT<float> *d_leaf_units, const UnitT<float> *d_leaf_cold,
                                  int leaf_units, int sub_from, int sub_units, UnitT<float> *d_stream, UnitT<float> *d_cold, const float c_lo[3],
                                  const float c_hi[3], float box_pad, int link_base, int *built_units) {
    if (n < 1 || !built_units) return hipErrorInvalidValue;
    (void)leaf_units;
    // one scratch allocation: keys in/out, the tree, the sort's own storage
    size_t sort_bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, n, 0, 64, 0);
    if (e != hipSuccess) return e;
    const size_t ni = (size_t)(n > 1 ? n - 1 : 1);
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t o_k0 = 0, o_k1 = o_k0 + up(8 * (size_t)n), o_int = o_k1 + up(8 * (size_t)n);
    const size_t o_f = o_int + up(4 * (8 * ni + (size_t)n)), o_sort = o_f + up(4 * 6 * ni), bytes = o_sort + up(sort_bytes);
    unsigned char *tmp = nullptr;
    e = hipMalloc((void **)&tmp, bytes);
    if (e != hipSuccess) return e;
    unsigned long long *k0 = (unsigned long long *)(tmp + o_k0), *k1 = (unsigned long long *)(tmp + o_k1);
    int *ip = (int *)(tmp + o_int);
    Tree t;
    t.left = ip; t.right = ip + ni; t.parent_i = ip + 2 * ni; t.flag = ip + 3 * ni; t.size = ip + 4 * ni; t.count = ip + 5 * ni;
    t.first_is_right = ip + 6 * ni; t.boxed = ip + 7 * ni; t.parent_l = ip + 8 * ni;
    t.lo = (float *)(tmp + o_f);
    t.hi = t.lo + 3 * ni;
    auto scale = [&](int a) { const float ext = c_hi[a] - c_lo[a]; return ext > 0.0f && ext < 3.0e38f ? 1024.0f / ext : 0.0f; };
    const int blocks = (n + TPB - 1) / TPB;
    int order = 5;
    if (const char *e = dev_knob("RT_AMD_DEV_ORDER")) order = atoi(e); // development knob
    float thr2 = 0.8f, thrn = 0.95f; // as the host build's (rt_flatten.h BuildKnobs)
    if (const char *e = dev_knob("RT_AMD_DEV_T2")) thr2 = (float)atof(e);
    if (const char *e = dev_knob("RT_AMD_DEV_TN")) thrn = (float)atof(e);
    int root_size = 0, one_leaf_units = 0;
    do {
        e = hipMemsetAsync(tmp + o_int, 0xff, up(4 * (8 * ni + (size_t)n)), 0); // parents = -1 ...
        if (e != hipSuccess) break;
        e = hipMemsetAsync(t.flag, 0, 4 * ni, 0);                               // ... arrival counters = 0
        if (e != hipSuccess) break;
        keys_kernel<<<blocks, TPB, 0, 0>>>(d_leaves, n, c_lo[0], c_lo[1], c_lo[2], scale(0), scale(1), scale(2), k0);
        e = hipcub::DeviceRadixSort::SortKeys(tmp + o_sort, sort_bytes, k0, k1, n, 0, 64, 0);
        if (e != hipSuccess) break;
        if (n > 1) {
            tree_kernel<<<blocks, TPB, 0, 0>>>(k1, n, t);
            fit_kernel<<<blocks, TPB, 0, 0>>>(k1, d_leaves, n, t, order, thr2, thrn);
        }
        emit_kernel<<<(2 * n + TPB - 1) / TPB, TPB, 0, 0>>>(k1, d_leaves, n, t, d_leaf_units, d_leaf_cold, d_stream, d_cold, box_pad, link_base, sub_from,
                                                          sub_units);
        e = hipGetLastError();
        if (e != hipSuccess) break;
        // how long the stream came out (nodes whose box is nearly their parent's have no BOX record): one word back
        if (n > 1) e = hipMemcpy(&root_size, t.size, sizeof(int), hipMemcpyDeviceToHost);
        else e = hipMemcpy(&one_leaf_units, &d_leaves[0].n, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) break;
        e = hipStreamSynchronize(0);
    } while (false);
    (void)hipFree(tmp);
    *built_units = (n > 1 ? root_size + 2 : one_leaf_units) + 2 + sub_units;
    return e;
}

} // namespace rtamd
