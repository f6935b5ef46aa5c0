// host_abi.cpp -- C entry points of librtamd_host.so (include/rtamd_host.h).
#include "../../include/rtamd_host.h"

#include "scenes.h"

struct rtamd_host_scene {
    rtamd::BuiltScene built;
    rtamd::FlatScene flat;
    rt_scene_desc desc{};
    rt_camera cam{};
    camera fresh; // the scene's camera as the scene function leaves it, before any initialize()
};

extern "C" {

rtamd_host_scene *rtamd_host_build_scene(int scene, int image_width, int image_height, int samples_per_pixel,
                                         int max_depth, int use_bvh, int frame, const char *asset_dir) {
    if (image_width <= 0 || image_height <= 0 || samples_per_pixel <= 0 || max_depth < 0) return nullptr;
    auto *s = new rtamd_host_scene;
    std::srand(1); // a fresh reference process starts from the default seed
    if (!rtamd::build_scene(scene, use_bvh != 0, frame, asset_dir ? asset_dir : "", s->built)) {
        delete s;
        return nullptr;
    }
    camera &c = s->built.cam;
    c.image_width = image_width;
    c.aspect_ratio = double(image_width) / double(image_height); // cpuRenderer.cpp:437
    c.samples_per_pixel = samples_per_pixel;
    c.max_depth = max_depth;
    s->fresh = c;
    s->cam = c.frame();
    camera::flatten(s->built.world, s->built.lights, s->flat);
    s->desc = s->flat.desc();
    return s;
}

const rt_scene_desc *rtamd_host_scene_desc(const rtamd_host_scene *s) { return s ? &s->desc : nullptr; }

int rtamd_host_scene_camera(const rtamd_host_scene *s, rt_camera *out) {
    if (!s || !out) return RT_ERR_INVALID;
    *out = s->cam;
    return RT_OK;
}

int rtamd_host_scene_animation(const rtamd_host_scene *s, int method, rt_camera *out, int cap) {
    if (!s || method < 0 || method > 1) return RT_ERR_INVALID;
    camera c = s->fresh; // every call replays the sequence from the scene function's camera
    c.animation_method = method;
    const int n = c.animation_frames();
    if (!out) return n;
    if (cap < n) return RT_ERR_INVALID;
    for (int f = 0; f < n; f++) out[f] = c.animation_step(f);
    return n;
}

int rtamd_host_load_image(const char *path, int *width, int *height, unsigned char *out, size_t cap) {
    if (!path || !width || !height) return RT_ERR_INVALID;
    rtw_image img;
    if (!img.load(path)) return RT_ERR_INVALID;
    *width = img.width();
    *height = img.height();
    if (out) {
        if (cap < img.bytes().size()) return RT_ERR_INVALID;
        std::memcpy(out, img.bytes().data(), img.bytes().size());
    }
    return RT_OK;
}

int rtamd_host_scene_save(const rtamd_host_scene *s, const char *path) {
    if (!s || !path) return RT_ERR_INVALID;
    return s->flat.save(path, &s->cam) ? RT_OK : RT_ERR_INVALID;
}

void rtamd_host_scene_free(rtamd_host_scene *s) { delete s; }

} // extern "C"
