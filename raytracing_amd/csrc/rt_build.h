// rt_build.h -- the float record stream built ON THE DEVICE: the host hands over the scene's leaf
// records (what the flattener would put under its own hierarchy) with their bounds; the device builds the hierarchy over
// them -- the host's own top-down binned surface-area-heuristic build (Flattener::build, rt_flatten.h), one kernel launch
// per tree level --, fits the boxes bottom-up and writes the threaded stream of rt_scene.h -- BOX records with skip links,
// leaves in place, sentinel -- straight into the scene's blob.  The stream comes out as the host's does, unit for unit.
// Replaces, for scenes that are rebuilt per frame, the host's bvh_node build (reference bvh.h:15-41; the reference's CUDA
// tree builds on the device too, src/cuda/bvh.h:140-188) + flattening.  world.hit() is the closest hit over all
// primitives, so the result does not depend on the hierarchy (ties and grazing rays aside: tests/f32_bounds.py).
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_scene.h"

namespace rtamd {

struct BuildLeaf {      // one leaf of the leaf stream
    double lo[3], hi[3]; // bounds in the world frame (the host's doubles: the build decides as the host's does)
    int32_t off, n;      // its units in the leaf stream
};

// upper bound of the units of the finished stream: the leaves, a BOX record per internal node of the binary tree, the sentinel
// (nodes whose box is nearly as large as their boxed ancestor's get no BOX record: the stream usually comes out shorter)
inline size_t built_stream_units(size_t leaf_units, size_t n_leaves) { return leaf_units + (n_leaves > 1 ? 2 * (n_leaves - 1) : 0) + 2; }

struct DeviceBuildKnobs { // as BuildKnobs (rt_flatten.h)
    int bins = 12, order = 1;
    double thr2 = 0.8, thrn = 0.95, overlap = 0.1;
};

// d_leaves / d_leaf_units / d_leaf_cold: device copies of the leaf table and the leaf stream (+ its cold units); units
// [sub_from, sub_from + sub_units) of the leaf stream are the boundary sub-streams of general media (placed behind the
// sentinel).  d_stream / d_cold: room for built_stream_units() units each.  box_pad: see Flattener::run (flat float boxes).
// *built_units: units written (tree, sentinel, sub-streams).  Synchronous: returns after the stream is complete.
hipError_t build_stream_on_device(const BuildLeaf *d_leaves, int n_leaves, const UnitT<float> *d_leaf_units, const UnitT<float> *d_leaf_cold,
                                  int sub_from, int sub_units, UnitT<float> *d_stream, UnitT<float> *d_cold, double box_pad,
                                  const DeviceBuildKnobs &kn, int *built_units);

} // namespace rtamd
#endif
