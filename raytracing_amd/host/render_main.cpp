// render_main.cpp -- command-line driver with the reference CPU renderer's positional argv contract
// (cpuRenderer.cpp:426-438): scene spp depth animation anim_method use_openmp use_bvh num_threads W H
// (use_openmp / num_threads are accepted and ignored).  Writes ./images/<test.ppm | animation/imageN.ppm> like the
// reference and prints the same two timing lines (cpuRenderer.cpp:466-467).
#include <chrono>
#include <string>

#include "scenes.h"

int main(int argc, char *argv[]) {
    int scene = 1, spp = 200, depth = 30, method = 1, W = 600, H = 600;
    bool animation = false, use_bvh = true;
    if (argc > 10) {
        scene = std::atoi(argv[1]);
        spp = std::atoi(argv[2]);
        depth = std::atoi(argv[3]);
        animation = std::string(argv[4]) == "true";
        method = std::atoi(argv[5]);
        use_bvh = std::string(argv[7]) == "true";
        W = std::atoi(argv[9]);
        H = std::atoi(argv[10]);
    } else if (argc > 1) {
        std::cerr << "usage: rtamd_render scene spp depth animation anim_method use_openmp use_bvh num_threads W H\n";
        return 2;
    }
    const char *assets = getenv("RTAMD_ASSETS");
    const std::string asset_dir = assets ? assets : "raytracing_amd/assets";
    std::clog << "samples_per_pixel: " << spp << " \nmax_depth: " << depth << " \nanimation: " << animation
              << " \nanimation_method: " << method << " \nuse_bvh: " << use_bvh << " \n";
    auto t0 = std::chrono::high_resolution_clock::now();
    try {
        auto setup = [&](rtamd::BuiltScene &s) {
            s.cam.image_width = W;
            s.cam.aspect_ratio = double(W) / double(H);
            s.cam.samples_per_pixel = spp;
            s.cam.max_depth = depth;
            s.cam.use_bvh = use_bvh;
        };
        if (animation && method == 2) { // cpuRenderer.cpp:447-451: 20 frames, scene rebuilt per frame
            for (int frame = 0; frame < 20; frame++) {
                rtamd::BuiltScene s;
                if (!rtamd::build_scene(5, use_bvh, frame, asset_dir, s)) return 2;
                setup(s);
                s.cam.render(s.world, s.lights);
            }
        } else {
            rtamd::BuiltScene s;
            if (!rtamd::build_scene(scene, use_bvh, 0, asset_dir, s)) { std::cerr << "bad scene\n"; return 2; }
            setup(s);
            s.cam.filename = "test.ppm";
            if (animation) {
                s.cam.animation_method = method;
                s.cam.render_animation(s.world, s.lights);
            } else {
                s.cam.render(s.world, s.lights);
            }
            std::clog << "kernel ms: " << s.cam.last_stats.kernel_ms << "  samples: " << s.cam.last_stats.samples
                      << "  rays: " << s.cam.last_stats.rays << "\n";
        }
    } catch (const std::exception &e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    auto t1 = std::chrono::high_resolution_clock::now();
    double total = (double)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
    std::clog << "Total render time (ms): " << total << "\n";
    std::clog << "Average time per pixel (ms): " << total / double(W * H) << "\n";
    return 0;
}
