// rt_build.h -- the float record stream built ON THE DEVICE: the host hands over the scene's leaf
// records (what the flattener would put under its own hierarchy) with their bounds; the device sorts them along a Morton
// curve, links a binary radix tree over the sorted keys (Karras 2012), fits the boxes bottom-up and writes the threaded
// stream of rt_scene.h -- BOX records with skip links, leaves in place, sentinel -- straight into the scene's blob.
// Replaces, for scenes that are rebuilt per frame, the host's bvh_node build (reference bvh.h:15-41; the reference's CUDA
// tree builds on the device too, src/cuda/bvh.h:140-188) + flattening.  world.hit() is the closest hit over all
// primitives, so the result does not depend on the hierarchy (ties and grazing rays aside: tests/f32_bounds.py).
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_scene.h"

namespace rtamd {

struct BuildLeaf {   // one leaf record of the leaf stream
    float lo[3], hi[3]; // bounds in the world frame, rounded outward
    int32_t off, n;     // its units in the leaf stream
};

// upper bound of the units of the finished stream: the leaves, a BOX record per internal node of the binary tree, the sentinel
// (nodes whose box is nearly as large as their parent's get no BOX record: the stream usually comes out shorter)
inline size_t built_stream_units(size_t leaf_units, size_t n_leaves) { return leaf_units + (n_leaves > 1 ? 2 * (n_leaves - 1) : 0) + 2; }

// d_leaves / d_leaf_units / d_leaf_cold: device copies of the leaf table and the leaf stream (+ its cold units); units
// [sub_from, sub_from + sub_units) of the leaf stream are the boundary sub-streams of general media (placed behind the
// sentinel).  d_stream / d_cold: room for built_stream_units() units each.  c_lo / c_hi: bounds of the leaves' centroids (for
// the Morton grid).  box_pad: see Flattener::run (flat float boxes).  link_base: units in front of d_stream[0] in the scene's
// stream (leaves the host put ahead of the tree): links are positions in the whole stream.  *built_units: units written from
// d_stream[0] on (tree, sentinel, sub-streams).  Synchronous: returns after the stream is complete.
hipError_t build_stream_on_device(const BuildLeaf *d_leaves, int n_leaves, const UnitT<float> *d_leaf_units, const UnitT<float> *d_leaf_cold,
                                  int leaf_units, int sub_from, int sub_units, UnitT<float> *d_stream, UnitT<float> *d_cold, const float c_lo[3],
                                  const float c_hi[3], float box_pad, int link_base, int *built_units);

} // namespace rtamd
#endif
