// scenes.h -- the reference's scenes, restated against the scene API (include/rtamd/scene_api.h).
#ifndef RTAMD_SCENES_H
#define RTAMD_SCENES_H
#include <string>

#include "../../include/rtamd/scene_api.h"

namespace rtamd {

struct BuiltScene {
    hittable_list world;
    hittable_list lights;
    camera cam;
};

// scene ids follow cpuRenderer.cpp:453-458 (5 = one frame of the bouncing-spheres animation, cpuRenderer.cpp:19)
bool build_scene(int scene, bool use_bvh, int frame, const std::string &asset_dir, BuiltScene &out);

} // namespace rtamd
#endif
