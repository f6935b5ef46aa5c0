// rt_build.hip -- device build of the float record stream (see rt_build.h).
#include "rt_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace rtamd {
namespace {

constexpr int TPB = 256;

struct Tree { // n - 1 internal nodes over n leaves in their final order; a child c >= 0 is internal node c, c < 0 is the leaf at place ~c
    int *left, *right, *parent_i, *parent_l; // parent of an internal node / of the leaf at a place
    int *flag;                               // arrivals at an internal node (bottom-up pass)
    int *size, *first_is_right;              // units of the subtree's records WITHOUT a BOX record of its own; emission order of the children
    int *boxed;                              // the node gets a BOX record
    double *lo, *hi;                         // 3 per internal node
};

// double -> float rounded down / up (rt_flatten.h cvt_lo / cvt_hi)
__device__ inline float cvt_lo_f(double x) {
    float o = (float)x;
    if ((double)o > x) o = nextafterf(o, -INFINITY);
    return o;
}
__device__ inline float cvt_hi_f(double x) {
    float o = (float)x;
    if ((double)o < x) o = nextafterf(o, INFINITY);
    return o;
}

// ---- the host's hierarchy, built on the device ------------------------------------------------------------------------------
// Top-down binned surface-area-heuristic build: the algorithm of Flattener::build (rt_flatten.h) -- bins over the centroids on
// three axes, cost = area(L) |L| + area(R) |R| + overlap * area(L ^ R) (|L| + |R|), the earliest cheapest (axis, bin), a BOX
// record only for a node whose area is below thr2 / thrn of its closest boxed ancestor's, the child with the larger box first
// -- with the same arithmetic, in double on the leaves' double bounds, so the stream comes out as the host's does (the test
// compares them unit for unit).  One kernel launch per tree level, one wave per node: a node's leaves are a range of the index
// array, a split is a stable partition of that range into the other of two index arrays; the nodes of the next level are
// appended to a work list.  (Rounds 2 and 3 linked a binary radix tree over Morton keys here -- Karras 2012 --: built in
// two thirds of the time, final_scene rendered 22 % slower on it, mesh_scene 7 %: DESIGN.md section 7.)
struct SahWork {
    int begin, n;       // the node's leaves: idx[begin, begin + n)
    int parent, info;   // parent node (-1: the root); bit 0: this is the parent's right child, bits 1..: depth
    double parent_area; // area of the closest ancestor with a BOX record
};
struct SahKnobs {
    int bins, order;
    double thr2, thrn, overlap;
};
constexpr int SAH_MAX_BINS = 16, SAH_WAVES = 4, SAH_BIN_WORDS = 7; // a bin: lo[3], hi[3] as ordered 64-bit integers, count
constexpr int SAH_MAX_LEVELS = 64 + 32 + 2;                         // (splits below depth 64 are halves: rt_flatten.h)

typedef unsigned long long u64;
__device__ inline u64 f_ord(double f) { // double -> unsigned, order-preserving (for integer atomic min / max)
    const u64 u = (u64)__double_as_longlong(f);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline double f_unord(u64 u) { return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u)); }
__device__ inline double area3(const double lo[3], const double hi[3]) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
__device__ inline double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// counts[l]: nodes of level l (zero at the start; counts[0] is taken as 1); n_nodes: internal nodes handed out so far
__global__ __launch_bounds__(SAH_WAVES * 64) void sah_level_kernel(const BuildLeaf *L, int n, int level, Tree t, unsigned long long *final_keys,
                                                                   int *idx_a, int *idx_b, SahWork *work_a, SahWork *work_b, int *counts,
                                                                   int *n_nodes, SahKnobs kn) {
    __shared__ u64 s_bins[SAH_WAVES][3][SAH_MAX_BINS][SAH_BIN_WORDS]; // per wave
    __shared__ u64 s_suf[SAH_WAVES][3][SAH_MAX_BINS][SAH_BIN_WORDS];  // per wave: bins b .. NB-1 united
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NB = kn.bins;
    const int cnt = level == 0 ? 1 : counts[level];
    const int *idx_in = (level & 1) ? idx_b : idx_a; // (level 0: the identity)
    int *idx_out = (level & 1) ? idx_a : idx_b;
    const SahWork *cur = (level & 1) ? work_b : work_a;
    SahWork *nxt = (level & 1) ? work_a : work_b;
    for (int w = blockIdx.x * SAH_WAVES + wave; w < cnt; w += gridDim.x * SAH_WAVES) {
        const SahWork wk = level == 0 ? SahWork{0, n, -1, 0, 0.0} : cur[w];
        const int begin = wk.begin, m = wk.n, depth = wk.info >> 1;
        // the node's leaves, four per lane at a time (the loads of a round are independent: one latency, not four)
        auto for_leaves = [&](auto &&fn) {
            for (int base = 0; base < m; base += 256) {
                int id[4];
                BuildLeaf l[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = base + j * 64 + lane;
                    id[j] = k < m ? (level == 0 ? k : idx_in[begin + k]) : -1;
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (id[j] >= 0) l[j] = L[id[j]];
#pragma unroll
                for (int j = 0; j < 4; j++) fn(base + j * 64 + lane, id[j], l[j]); // (called by every lane: id < 0 = no leaf)
            }
        };
        // bounds of the node and of its leaves' centroids
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for_leaves([&](int, int id, const BuildLeaf &l) {
            if (id < 0) return;
            for (int a = 0; a < 3; a++) {
                lo[a] = fmin(lo[a], l.lo[a]);
                hi[a] = fmax(hi[a], l.hi[a]);
                const double c = 0.5 * (l.lo[a] + l.hi[a]);
                clo[a] = fmin(clo[a], c);
                chi[a] = fmax(chi[a], c);
            }
        });
        for (int a = 0; a < 3; a++) {
            lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]);
            clo[a] = wave_min(clo[a]); chi[a] = wave_max(chi[a]);
        }
        const double area = area3(lo, hi);
        const double parent_area = wk.parent < 0 ? area : wk.parent_area; // (the world's own box gets no BOX record: rt_flatten.h emit_group)
        const bool boxed = area < (m == 2 ? kn.thr2 : kn.thrn) * parent_area;
        int node = 0;
        if (lane == 0) node = level == 0 ? 0 : 1 + atomicAdd(n_nodes, 1); // (the root is node 0: emit_kernel starts there)
        node = __shfl(node, 0, 64);
        if (lane == 0) {
            t.boxed[node] = boxed ? 1 : 0;
            t.parent_i[node] = wk.parent;
            if (wk.parent >= 0) ((wk.info & 1) ? t.right : t.left)[wk.parent] = node;
        }
        int n_left = -1, best_axis = -1, best_bin = -1;
        double l_lo[3], l_hi[3], r_lo[3], r_hi[3]; // the two sides' boxes
        if (m > 2 && depth < 64) {
            u64(*bins)[SAH_MAX_BINS][SAH_BIN_WORDS] = s_bins[wave];
            u64(*suf)[SAH_MAX_BINS][SAH_BIN_WORDS] = s_suf[wave];
            for (int i = lane; i < 3 * SAH_MAX_BINS; i += 64) {
                u64 *b = bins[i / SAH_MAX_BINS][i % SAH_MAX_BINS];
                b[0] = b[1] = b[2] = ~0ull;
                b[3] = b[4] = b[5] = 0ull;
                b[6] = 0ull;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double sc[3];
            bool use[3];
            for (int a = 0; a < 3; a++) {
                const double ext = chi[a] - clo[a];
                use[a] = ext > 0.0 && isfinite(ext);
                sc[a] = use[a] ? (double)NB / ext : 0.0;
            }
            for_leaves([&](int, int id, const BuildLeaf &l) {
                if (id < 0) return;
                for (int a = 0; a < 3; a++) {
                    if (!use[a]) continue;
                    int b = (int)((0.5 * (l.lo[a] + l.hi[a]) - clo[a]) * sc[a]);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    u64 *bb = bins[a][b];
                    for (int c = 0; c < 3; c++) {
                        atomicMin(&bb[c], f_ord(l.lo[c]));
                        atomicMax(&bb[3 + c], f_ord(l.hi[c]));
                    }
                    atomicAdd(&bb[6], 1ull);
                }
            });
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // lane a sweeps axis a
            double best = INFINITY;
            int my_bin = -1;
            double bl_lo[3] = {0, 0, 0}, bl_hi[3] = {0, 0, 0};
            int bl_n = 0;
            if (lane < 3 && use[lane]) {
                const int a = lane;
                double alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
                unsigned cn = 0;
                for (int b = NB - 1; b > 0; b--) {
                    for (int c = 0; c < 3; c++) {
                        alo[c] = fmin(alo[c], f_unord(bins[a][b][c]));
                        ahi[c] = fmax(ahi[c], f_unord(bins[a][b][3 + c]));
                        suf[a][b][c] = f_ord(alo[c]);
                        suf[a][b][3 + c] = f_ord(ahi[c]);
                    }
                    cn += (unsigned)bins[a][b][6];
                    suf[a][b][6] = cn;
                }
                for (int c = 0; c < 3; c++) { alo[c] = INFINITY; ahi[c] = -INFINITY; }
                cn = 0;
                for (int b = 0; b + 1 < NB; b++) { // left = bins 0 .. b
                    for (int c = 0; c < 3; c++) {
                        alo[c] = fmin(alo[c], f_unord(bins[a][b][c]));
                        ahi[c] = fmax(ahi[c], f_unord(bins[a][b][3 + c]));
                    }
                    cn += (unsigned)bins[a][b][6];
                    const unsigned rn = (unsigned)suf[a][b + 1][6];
                    if (cn == 0 || rn == 0) continue;
                    double ll[3], lh[3], rl[3], rh[3];
                    for (int c = 0; c < 3; c++) {
                        ll[c] = alo[c]; lh[c] = ahi[c];
                        rl[c] = f_unord(suf[a][b + 1][c]); rh[c] = f_unord(suf[a][b + 1][3 + c]);
                    }
                    double cost = area3(ll, lh) * (double)cn + area3(rl, rh) * (double)rn;
                    if (kn.overlap > 0.0) { // a ray through the part the two sides share walks both
                        double ol[3], oh[3];
                        bool any = true;
                        for (int c = 0; c < 3; c++) {
                            ol[c] = fmax(ll[c], rl[c]);
                            oh[c] = fmin(lh[c], rh[c]);
                            any = any && oh[c] >= ol[c];
                        }
                        if (any) cost += kn.overlap * area3(ol, oh) * (double)(cn + rn);
                    }
                    if (cost < best) {
                        best = cost;
                        my_bin = b;
                        bl_n = (int)cn;
                        for (int c = 0; c < 3; c++) { bl_lo[c] = alo[c]; bl_hi[c] = ahi[c]; }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (the suffix boxes are read by every lane below)
            __builtin_amdgcn_wave_barrier();
            // the earliest cheapest (axis, bin)
            double bc = INFINITY;
            for (int a = 0; a < 3; a++) {
                const double ca = __shfl(best, a, 64);
                const int ba = __shfl(my_bin, a, 64);
                if (ba >= 0 && ca < bc) { bc = ca; best_axis = a; best_bin = ba; }
            }
            if (best_axis >= 0) {
                n_left = __shfl(bl_n, best_axis, 64);
                for (int c = 0; c < 3; c++) {
                    l_lo[c] = __shfl(bl_lo[c], best_axis, 64);
                    l_hi[c] = __shfl(bl_hi[c], best_axis, 64);
                    r_lo[c] = f_unord(suf[best_axis][best_bin + 1][c]);
                    r_hi[c] = f_unord(suf[best_axis][best_bin + 1][3 + c]);
                }
            }
            __builtin_amdgcn_wave_barrier(); // (the next node of this wave clears the bins)
        }
        const bool split = n_left > 0 && n_left < m;
        const int child_l = begin, child_r = begin + (split ? n_left : m / 2);
        const int nl = child_r - begin, nr = m - nl;
        if (split) {
            // stable partition by the bin on the chosen axis
            const int a = best_axis;
            const double sca = (double)NB / (chi[a] - clo[a]);
            int cl = 0, cr = 0;
            for_leaves([&](int, int id, const BuildLeaf &l) {
                bool left = false;
                if (id >= 0) {
                    int b = (int)((0.5 * (l.lo[a] + l.hi[a]) - clo[a]) * sca);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    left = b <= best_bin;
                }
                const unsigned long long ml = __ballot(id >= 0 && left), mr = __ballot(id >= 0 && !left);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (id >= 0) {
                    const int pos = left ? child_l + cl + __popcll(ml & below) : child_r + cr + __popcll(mr & below);
                    idx_out[pos] = id;
                    if ((left && nl == 1) || (!left && nr == 1)) { // a child of one leaf: the leaf's place in the tree
                        final_keys[pos] = (unsigned long long)(unsigned)id;
                        t.parent_l[pos] = node;
                    }
                }
                cl += __popcll(ml);
                cr += __popcll(mr);
            });
        } else {
            // no useful split (two leaves, equal centroids, depth guard).  Two leaves: the lower centroid along the widest
            // centroid axis is the left one, as the host's nth_element has it; more: halves in index order (the host takes the
            // median along that axis -- the two agree when the centroids are equal).
            int a = 0;
            for (int k = 1; k < 3; k++)
                if (chi[k] - clo[k] > chi[a] - clo[a]) a = k;
            bool swap2 = false;
            if (m == 2) {
                const int i0 = level == 0 ? 0 : idx_in[begin], i1 = level == 0 ? 1 : idx_in[begin + 1];
                const BuildLeaf l0 = L[i0], l1 = L[i1];
                const double c0 = l0.lo[a] + l0.hi[a], c1 = l1.lo[a] + l1.hi[a];
                swap2 = !(c0 < c1 || (c0 == c1 && i0 < i1));
            }
            double bl[3] = {INFINITY, INFINITY, INFINITY}, bh[3] = {-INFINITY, -INFINITY, -INFINITY};
            double cl2[3] = {INFINITY, INFINITY, INFINITY}, ch2[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = lane; k < m; k += 64) {
                const int src = swap2 ? 1 - k : k;
                const int id = level == 0 ? src : idx_in[begin + src];
                idx_out[begin + k] = id;
                if ((k < nl && nl == 1) || (k >= nl && nr == 1)) {
                    final_keys[begin + k] = (unsigned long long)(unsigned)id;
                    t.parent_l[begin + k] = node;
                }
                const BuildLeaf l = L[id]; // the two sides' boxes, for the order rule
                for (int c = 0; c < 3; c++) {
                    if (k < nl) { bl[c] = fmin(bl[c], l.lo[c]); bh[c] = fmax(bh[c], l.hi[c]); }
                    else { cl2[c] = fmin(cl2[c], l.lo[c]); ch2[c] = fmax(ch2[c], l.hi[c]); }
                }
            }
            for (int c = 0; c < 3; c++) {
                l_lo[c] = wave_min(bl[c]); l_hi[c] = wave_max(bh[c]);
                r_lo[c] = wave_min(cl2[c]); r_hi[c] = wave_max(ch2[c]);
            }
        }
        if (lane == 0) {
            // which side goes first in the stream (Flattener::build): 0 as split, 1 / 2 larger / smaller box, 3 / 4 more / fewer
            // leaves, 5 larger box area per leaf
            const double al = area3(l_lo, l_hi), ar = area3(r_lo, r_hi);
            int first = 0;
            if (kn.order == 1 || kn.order == 2) first = ((kn.order == 1) == (al >= ar)) ? 0 : 1;
            else if (kn.order == 3 || kn.order == 4) first = ((kn.order == 3) == (nl >= nr)) ? 0 : 1;
            else if (kn.order >= 5) first = al * (double)nr >= ar * (double)nl ? 0 : 1;
            t.first_is_right[node] = first;
            if (nl == 1) t.left[node] = ~child_l;
            if (nr == 1) t.right[node] = ~child_r;
            const double pa = boxed ? area : parent_area;
            const int n_more = (nl > 1 ? 1 : 0) + (nr > 1 ? 1 : 0);
            if (n_more && level + 1 < SAH_MAX_LEVELS) {
                int at = atomicAdd(&counts[level + 1], n_more);
                if (nl > 1) nxt[at++] = SahWork{child_l, nl, node, (depth + 1) << 1, pa};
                if (nr > 1) nxt[at] = SahWork{child_r, nr, node, ((depth + 1) << 1) | 1, pa};
            }
        }
    }
}

// bottom-up: the second thread to arrive at a node finds both children complete and fits the node's box and size (which nodes
// get a BOX record and which child goes first was decided on the way down)
__global__ void fit_kernel(const unsigned long long *k, const BuildLeaf *L, int n, Tree t) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int node = t.parent_l[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&t.flag[node], 1) == 0) return;
        __threadfence();
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int size = 0;
        for (int c = 0; c < 2; c++) {
            const int ch = c == 0 ? t.left[node] : t.right[node];
            if (ch < 0) {
                const BuildLeaf l = L[(unsigned)k[~ch]];
                for (int a = 0; a < 3; a++) { lo[a] = fmin(lo[a], l.lo[a]); hi[a] = fmax(hi[a], l.hi[a]); }
                size += l.n;
            } else {
                for (int a = 0; a < 3; a++) { lo[a] = fmin(lo[a], t.lo[3 * ch + a]); hi[a] = fmax(hi[a], t.hi[3 * ch + a]); }
                size += t.size[ch] + (t.boxed[ch] ? 2 : 0);
            }
        }
        for (int a = 0; a < 3; a++) {
            t.lo[3 * node + a] = lo[a];
            t.hi[3 * node + a] = hi[a];
        }
        t.size[node] = size;
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
                            const UnitT<float> *leaf_cold, UnitT<float> *stream, UnitT<float> *cold, double box_pad,
                            int sub_from, int sub_units, int *tree_units_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const UnitT<float> zero{{0.0f, 0.0f, 0.0f, 0.0f}};
    const UnitT<float> no_node{{0.0f, 0.0f, 0.0f, enc_int(-1, 0.0f)}};
    // the tree's records, then the sentinel, then the boundary sub-streams of general media
    const int tree_units = n > 1 ? t.size[0] + (t.boxed[0] ? 2 : 0) : L[(unsigned)k[0]].n;
    const int sub_to = tree_units + 2;
    if (g < n) { // sorted leaf g: its record(s); positions inside move with them
        const BuildLeaf l = L[(unsigned)k[g]];
        const int o = n > 1 ? offset_of(k, L, t, ~g, t.parent_l[g]) : 0;
        for (int u = 0; u < l.n; u++) {
            stream[o + u] = leaf_units[l.off + u];
            cold[o + u] = leaf_cold[l.off + u];
        }
        if (l.n > 2) relocate(stream, o, l.n, l.off, sub_from, sub_to);
        else if ((dec_int(stream[o + 1].w[3]) & 15) == K_MEDIUM_GEN) relocate(stream, o, l.n, l.off, sub_from, sub_to);
    } else if (g < 2 * n - 1) { // internal node: a BOX record, if it has one
        const int i = g - n;
        if (!t.boxed[i]) return;
        const int o = offset_of(k, L, t, i, t.parent_i[i]);
        float lo[3], hi[3]; // rounded outward, flat sides padded: as Flattener::build writes a BOX record
        for (int a = 0; a < 3; a++) {
            const double pad = (t.hi[3 * i + a] - t.lo[3 * i + a]) < 2.0 * box_pad ? box_pad : 0.0;
            lo[a] = cvt_lo_f(t.lo[3 * i + a] - pad);
            hi[a] = cvt_hi_f(t.hi[3 * i + a] + pad);
        }
        stream[o] = UnitT<float>{{lo[0], hi[0], lo[1], hi[1]}};
        stream[o + 1] = UnitT<float>{{lo[2], hi[2], enc_int((o + 2 + t.size[i]) * (int)sizeof(UnitT<float>), 0.0f), enc_int(make_hdr(K_BOX, 0), 0.0f)}};
        cold[o] = no_node;
        cold[o + 1] = zero;
    } else if (g == 2 * n - 1) { // the sentinel, and behind it the sub-streams (a few records: one thread)
        *tree_units_out = tree_units;
        stream[tree_units] = zero;
        stream[tree_units + 1] = UnitT<float>{{0.0f, 0.0f, 0.0f, enc_int(make_hdr(K_READY, 0), 0.0f)}};
        cold[tree_units] = no_node;
        cold[tree_units + 1] = zero;
        for (int u = 0; u < sub_units; u++) {
            stream[tree_units + 2 + u] = leaf_units[sub_from + u];
            cold[tree_units + 2 + u] = leaf_cold[sub_from + u];
        }
        if (sub_units > 0) relocate(stream, tree_units + 2, sub_units, sub_from, sub_from, sub_to);
    }
}

} // namespace

hipError_t build_stream_on_device(const BuildLeaf *d_leaves, int n, const UnitT<float> *d_leaf_units, const UnitT<float> *d_leaf_cold,
                                  int sub_from, int sub_units, UnitT<float> *d_stream, UnitT<float> *d_cold, double box_pad,
                                  const DeviceBuildKnobs &kn, int *built_units) {
    if (n < 1 || !built_units) return hipErrorInvalidValue;
    // one scratch allocation: the leaves' final order, the two index arrays, the tree, the work lists
    const size_t ni = (size_t)(n > 1 ? n - 1 : 1), nw = (size_t)n / 2 + 1;
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t o_keys = 0, o_idx = o_keys + up(8 * (size_t)n), o_int = o_idx + up(8 * (size_t)n);
    const size_t n_int = 7 * ni + (size_t)n + 1 + SAH_MAX_LEVELS + 1;
    const size_t o_f = o_int + up(4 * n_int), o_work = o_f + up(8 * 6 * ni), bytes = o_work + 2 * up(sizeof(SahWork) * nw);
    unsigned char *tmp = nullptr;
    hipError_t e = hipMalloc((void **)&tmp, bytes);
    if (e != hipSuccess) return e;
    unsigned long long *keys = (unsigned long long *)(tmp + o_keys);
    int *idx_a = (int *)(tmp + o_idx), *idx_b = idx_a + n;
    int *ip = (int *)(tmp + o_int);
    Tree t;
    t.left = ip; t.right = ip + ni; t.parent_i = ip + 2 * ni; t.flag = ip + 3 * ni; t.size = ip + 4 * ni;
    t.first_is_right = ip + 5 * ni; t.boxed = ip + 6 * ni; t.parent_l = ip + 7 * ni;
    int *d_tree_units = ip + 7 * ni + (size_t)n, *counts = d_tree_units + 1, *n_nodes = counts + SAH_MAX_LEVELS;
    t.lo = (double *)(tmp + o_f);
    t.hi = t.lo + 3 * ni;
    SahWork *wa = (SahWork *)(tmp + o_work), *wb = (SahWork *)(tmp + o_work + up(sizeof(SahWork) * nw));
    const int blocks = (n + TPB - 1) / TPB;
    int tree_units = 0;
    do {
        e = hipMemsetAsync(tmp + o_int, 0xff, 4 * (7 * ni + (size_t)n), 0); // links and parents = -1 ...
        if (e != hipSuccess) break;
        e = hipMemsetAsync(t.flag, 0, 4 * ni, 0);                           // ... arrival counters, level counts, node counter = 0
        if (e != hipSuccess) break;
        e = hipMemsetAsync(d_tree_units, 0, 4 * (1 + SAH_MAX_LEVELS + 1), 0);
        if (e != hipSuccess) break;
        if (n > 1) {
            const SahKnobs sk{kn.bins < 2 ? 2 : (kn.bins > SAH_MAX_BINS ? SAH_MAX_BINS : kn.bins), kn.order, kn.thr2, kn.thrn, kn.overlap};
            const int grid = std::min(256, (n / 2 + SAH_WAVES - 1) / SAH_WAVES + 1);
            // Levels are launched without waiting: a first batch sized for a balanced tree and a half, then the host looks, every
            // eight levels, whether the work list ran dry (a level without nodes costs a launch of a few microseconds).
            int level = 0, more = 1;
            for (int batch = std::min(28, 3 + (int)std::ceil(1.5 * std::log2((double)n))); more && level < SAH_MAX_LEVELS; batch = 8) {
                for (int k = 0; k < batch && level < SAH_MAX_LEVELS; k++, level++)
                    sah_level_kernel<<<grid, SAH_WAVES * 64, 0, 0>>>(d_leaves, n, level, t, keys, idx_a, idx_b, wa, wb, counts, n_nodes, sk);
                more = 0;
                if (level < SAH_MAX_LEVELS) e = hipMemcpy(&more, counts + level, sizeof(int), hipMemcpyDeviceToHost);
                if (e != hipSuccess) break;
            }
            if (e != hipSuccess) break;
            fit_kernel<<<blocks, TPB, 0, 0>>>(keys, d_leaves, n, t);
        } else {
            e = hipMemsetAsync(keys, 0, 8, 0); // the one leaf
            if (e != hipSuccess) break;
        }
        emit_kernel<<<(2 * n + TPB - 1) / TPB, TPB, 0, 0>>>(keys, d_leaves, n, t, d_leaf_units, d_leaf_cold, d_stream, d_cold, box_pad, sub_from, sub_units,
                                                          d_tree_units);
        e = hipGetLastError();
        if (e != hipSuccess) break;
        // how long the stream came out (nodes whose box is nearly their parent's have no BOX record): one word back
        e = hipMemcpy(&tree_units, d_tree_units, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) break;
        e = hipStreamSynchronize(0);
    } while (false);
    (void)hipFree(tmp);
    *built_units = tree_units + 2 + sub_units;
    return e;
}

} // namespace rtamd
