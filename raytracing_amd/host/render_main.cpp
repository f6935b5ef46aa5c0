// render_main.cpp -- command-line driver with the reference CPU renderer's positional argv contract
// (cpuRenderer.cpp:426-438): scene spp depth animation anim_method use_openmp use_bvh num_threads W H
// (use_openmp / num_threads are accepted and ignored).  Writes ./images/<test.ppm | animation/imageN.ppm> like the
// reference and prints the same two timing lines (cpuRenderer.cpp:466-467), preceded by where the time went.
// Extension: a leading or trailing `--gpus N` renders every frame on GPUs 0..N-1 of this node (the strata of a frame are
// split over them, rt_render_multi); $RTAMD_DEVICES="0,1,.." names the devices explicitly.
#include <chrono>
#include <string>

#include "scenes.h"

int main(int argc_, char *argv_[]) {
    std::vector<char *> args;
    for (int k = 0; k < argc_; k++) {
        if (std::string(argv_[k]) == "--gpus" && k + 1 < argc_) {
            const int n = std::atoi(argv_[++k]);
            if (n < 1 || n > 64) { std::cerr << "--gpus: expected 1..64\n"; return 2; }
            std::string list;
            for (int d = 0; d < n; d++) list += (d ? "," : "") + std::to_string(d);
            setenv("RTAMD_DEVICES", list.c_str(), /*overwrite=*/0);
        } else {
            args.push_back(argv_[k]);
        }
    }
    const int argc = (int)args.size();
    char **argv = args.data();
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
                s.cam.device_build = true; // a scene that is rebuilt for every frame: its hierarchy is built on the GPU
                s.cam.render(s.world, s.lights);
            }
        } else {
            rtamd::BuiltScene s;
            const double tb0 = rtamd::now_ms();
            if (!rtamd::build_scene(scene, use_bvh, 0, asset_dir, s)) { std::cerr << "bad scene\n"; return 2; }
            const double build_ms = rtamd::now_ms() - tb0;
            const double ti0 = rtamd::now_ms();
            const size_t n_gpus = rtamd::sessions().size(); // creates the GPU context(s): not part of any frame
            const double init_ms = rtamd::now_ms() - ti0;
            setup(s);
            s.cam.filename = "test.ppm";
            if (animation) {
                s.cam.animation_method = method;
                s.cam.render_animation(s.world, s.lights);
            } else {
                s.cam.render(s.world, s.lights);
            }
            const rtamd::FrameTiming &ft = s.cam.last_timing;
            std::clog << "gpus: " << n_gpus << "  gpu init ms: " << init_ms << "\n";
            std::clog << "scene build ms: " << build_ms << "  flatten ms: " << ft.flatten_ms << "  upload ms: " << ft.upload_ms
                      << "  render+readback ms: " << ft.render_ms << "  ppm write ms: " << ft.write_ms << "\n";
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
