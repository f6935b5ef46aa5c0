// scenes.cpp -- inputs of the benchmark/parity configurations: the reference's five scenes (cpuRenderer.cpp:19-401)
// written against the scene API.  Geometry is a function of the std::rand stream; draws are taken in the order the
// reference build takes them (g++ evaluates call arguments right to left, hence the explicit temporaries), which
// tests/test_host_api.py checks byte for byte against scenes serialised from the reference's own object graph.
#include "scenes.h"

namespace rtamd {

namespace {

void weekend_camera(camera &c, double vfov, point3 from, double focus) { // cpuRenderer.cpp:143-155 / 69-81
    c.background = color(0.70, 0.80, 1.00);
    c.vfov = vfov;
    c.lookfrom = from;
    c.lookat = point3(0, 0, 0);
    c.vup = vec3(0, 1, 0);
    c.defocus_angle = 0.6;
    c.focus_dist = focus;
}

void cornell_camera(camera &c, point3 from) { // cpuRenderer.cpp:205-216, 272-283, 382-393
    c.background = color(0, 0, 0);
    c.vfov = 40;
    c.lookfrom = from;
    c.lookat = point3(278, 278, 0);
    c.vup = vec3(0, 1, 0);
    c.defocus_angle = 0;
}

shared_ptr<hittable> ground_sphere() {
    auto checker = make_shared<checker_texture>(0.32, color(.8, .1, .1), color(.9, .9, .9));
    return make_shared<sphere>(point3(0, -1000, 0), 1000, make_shared<lambertian>(checker));
}

void wrap_bvh(hittable_list &world, bool use_bvh) {
    if (use_bvh) world = hittable_list(make_shared<bvh_node>(world));
}

void weekend_light(hittable_list &lights) { // cpuRenderer.cpp:133-135
    lights.add(make_shared<sphere>(point3(0, -1000, 0), 1, shared_ptr<material>()));
}

void first_scene(BuiltScene &s, bool use_bvh) { // cpuRenderer.cpp:86
    s.world.add(ground_sphere());
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            const double choose_mat = random_double();
            const double jz = random_double(), jx = random_double(); // point3(a + .9*rd(), .2, b + .9*rd()): z drawn first
            const point3 center(a + 0.9 * jx, 0.2, b + 0.9 * jz);
            if ((center - point3(4, 0.2, 0)).length() <= 0.9) continue;
            shared_ptr<material> m;
            if (choose_mat < 0.8) {
                const color c2 = color::random(), c1 = color::random();
                m = make_shared<lambertian>(c1 * c2);
            } else if (choose_mat < 0.95) {
                const color albedo = color::random(0.5, 1);
                const double fuzz = random_double(0, 0.5);
                m = make_shared<metal>(albedo, fuzz);
            } else {
                m = make_shared<dielectric>(1.5);
            }
            s.world.add(make_shared<sphere>(center, 0.2, m));
        }
    s.world.add(make_shared<sphere>(point3(0, 1, 0), 1.0, make_shared<dielectric>(1.5)));
    s.world.add(make_shared<sphere>(point3(-4, 1, 0), 1.0, make_shared<lambertian>(color(0.4, 0.2, 0.1))));
    s.world.add(make_shared<sphere>(point3(4, 1, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    wrap_bvh(s.world, use_bvh);
    weekend_light(s.lights);
    weekend_camera(s.cam, 20, point3(13, 2, 3), 10.0);
}

void bounce_frame(BuiltScene &s, bool use_bvh, int frame) { // cpuRenderer.cpp:19
    double h1 = 1.0, h2 = 1.0, h3 = 1.0;
    if (frame < 5) h1 += 0.2 * frame;
    else if (frame < 10) { h1 += 0.2 * (9 - frame); h2 += 0.2 * (frame - 5); }
    else if (frame < 15) { h2 += 0.2 * (14 - frame); h3 += 0.2 * (frame - 10); }
    else if (frame < 20) h3 += 0.2 * (19 - frame);
    s.world.add(ground_sphere());
    s.world.add(make_shared<sphere>(point3(0, h2, 0), 1.0, make_shared<dielectric>(1.5)));
    s.world.add(make_shared<sphere>(point3(-4, h1, 0), 1.0, make_shared<lambertian>(color(0.4, 0.2, 0.1))));
    s.world.add(make_shared<sphere>(point3(4, h3, 0), 1.0, make_shared<metal>(color(0.7, 0.6, 0.5), 0.0)));
    wrap_bvh(s.world, use_bvh);
    weekend_light(s.lights);
    weekend_camera(s.cam, 24, point3(17, 4, 5), 15.0);
    s.cam.filename = "animation/image" + std::to_string(frame) + ".ppm";
}

void cornell_walls(hittable_list &world) { // cpuRenderer.cpp:169-182 / 230-243
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(15, 15, 15));
    world.add(make_shared<quad>(point3(555, 0, 0), vec3(0, 0, 555), vec3(0, 555, 0), green));
    world.add(make_shared<quad>(point3(0, 0, 555), vec3(0, 0, -555), vec3(0, 555, 0), red));
    world.add(make_shared<quad>(point3(0, 555, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    world.add(make_shared<quad>(point3(0, 0, 555), vec3(555, 0, 0), vec3(0, 0, -555), white));
    world.add(make_shared<quad>(point3(555, 0, 555), vec3(-555, 0, 0), vec3(0, 555, 0), white));
    world.add(make_shared<quad>(point3(213, 554, 227), vec3(130, 0, 0), vec3(0, 0, 105), light));
}

void cornell_lights(hittable_list &lights) { // cpuRenderer.cpp:193-197 / 260-264
    auto none = shared_ptr<material>();
    lights.add(make_shared<quad>(point3(343, 554, 332), vec3(-130, 0, 0), vec3(0, 0, -105), none));
    lights.add(make_shared<sphere>(point3(190, 90, 190), 90, none));
}

void cornell_box(BuiltScene &s, bool use_bvh) { // cpuRenderer.cpp:227
    cornell_walls(s.world);
    auto mirror = make_shared<metal>(color(0.8, 0.85, 0.88), 0.0);
    shared_ptr<hittable> box1 = box(point3(0, 0, 0), point3(165, 330, 165), mirror);
    box1 = make_shared<rotate_y>(box1, 15);
    box1 = make_shared<translate>(box1, vec3(265, 0, 295));
    s.world.add(box1);
    s.world.add(make_shared<sphere>(point3(190, 90, 190), 90, make_shared<dielectric>(1.5)));
    wrap_bvh(s.world, use_bvh);
    cornell_lights(s.lights);
    cornell_camera(s.cam, point3(278, 278, -800));
}

// Scene 6, "cornell smoke": not one of the reference's scene functions, a scene its API permits (the book's cornell_smoke
// pattern): constant_mediums bounded by rotated + translated box() lists, by a box() list in world space and by a moving
// sphere.  oracle/refgen.cpp builds the same scene with the reference's own classes (fixtures tests/golden/*6.npz).
void cornell_smoke(BuiltScene &s, bool use_bvh) {
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(7, 7, 7));
    s.world.add(make_shared<quad>(point3(555, 0, 0), vec3(0, 0, 555), vec3(0, 555, 0), green));
    s.world.add(make_shared<quad>(point3(0, 0, 555), vec3(0, 0, -555), vec3(0, 555, 0), red));
    s.world.add(make_shared<quad>(point3(0, 555, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    s.world.add(make_shared<quad>(point3(0, 0, 555), vec3(555, 0, 0), vec3(0, 0, -555), white));
    s.world.add(make_shared<quad>(point3(555, 0, 555), vec3(-555, 0, 0), vec3(0, 555, 0), white));
    s.world.add(make_shared<quad>(point3(113, 554, 127), vec3(330, 0, 0), vec3(0, 0, 305), light));
    shared_ptr<hittable> box1 = box(point3(0, 0, 0), point3(165, 330, 165), white);
    box1 = make_shared<rotate_y>(box1, 15);
    box1 = make_shared<translate>(box1, vec3(265, 0, 295));
    shared_ptr<hittable> box2 = box(point3(0, 0, 0), point3(165, 165, 165), white);
    box2 = make_shared<rotate_y>(box2, -18);
    box2 = make_shared<translate>(box2, vec3(130, 0, 65));
    s.world.add(make_shared<constant_medium>(box1, 0.01, color(0, 0, 0)));
    s.world.add(make_shared<constant_medium>(box2, 0.01, color(1, 1, 1)));
    s.world.add(make_shared<constant_medium>(box(point3(30, 380, 300), point3(140, 470, 420), white), 0.02, color(.2, .4, .9)));
    s.world.add(make_shared<constant_medium>(make_shared<sphere>(point3(400, 420, 200), point3(430, 420, 200), 50, white), 0.05,
                                             color(.9, .6, .2)));
    s.world.add(make_shared<sphere>(point3(190, 90, 190), 60, make_shared<dielectric>(1.5)));
    wrap_bvh(s.world, use_bvh);
    auto none = shared_ptr<material>();
    s.lights.add(make_shared<quad>(point3(443, 554, 432), vec3(-330, 0, 0), vec3(0, 0, -305), none));
    cornell_camera(s.cam, point3(278, 278, -800));
}

void mesh_scene(BuiltScene &s, bool use_bvh, const std::string &asset_dir) { // cpuRenderer.cpp:166
    cornell_walls(s.world);
    auto mirror = make_shared<metal>(color(0.8, 0.85, 0.88), 0.0);
    const std::string obj = asset_dir + "/shapes/small_bunny.obj";
    hittable_list mesh1 = build_mesh(obj.c_str(), mirror, 230, use_bvh);
    s.world.add(make_shared<translate>(make_shared<rotate_y>(make_shared<hittable_list>(mesh1), 180), vec3(265, 120, 295)));
    wrap_bvh(s.world, use_bvh);
    cornell_lights(s.lights);
    cornell_camera(s.cam, point3(278, 278, -800));
}

void final_scene(BuiltScene &s, bool use_bvh) { // cpuRenderer.cpp:293
    hittable_list boxes1;
    auto ground = make_shared<lambertian>(color(0.48, 0.83, 0.53));
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++)
        for (int j = 0; j < boxes_per_side; j++) {
            const double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            const double x1 = x0 + w, y1 = random_double(1, 101), z1 = z0 + w;
            boxes1.add(box(point3(x0, y0, z0), point3(x1, y1, z1), ground));
        }
    hittable_list &world = s.world;
    if (use_bvh) world.add(make_shared<bvh_node>(boxes1));
    else world.add(make_shared<hittable_list>(boxes1));
    world.add(make_shared<quad>(point3(123, 554, 147), vec3(300, 0, 0), vec3(0, 0, 265), make_shared<diffuse_light>(color(15, 15, 15))));
    const point3 center1(400, 400, 200);
    world.add(make_shared<sphere>(center1, center1 + vec3(30, 0, 0), 50, make_shared<lambertian>(color(0.7, 0.3, 0.1))));
    world.add(make_shared<sphere>(point3(260, 150, 45), 50, make_shared<dielectric>(1.5)));
    world.add(make_shared<sphere>(point3(0, 150, 145), 50, make_shared<metal>(color(0.8, 0.8, 0.9), 1.0)));
    auto boundary = make_shared<sphere>(point3(360, 150, 145), 70, make_shared<dielectric>(1.5));
    world.add(boundary);
    world.add(make_shared<constant_medium>(boundary, 0.2, color(0.2, 0.4, 0.9)));
    boundary = make_shared<sphere>(point3(0, 0, 0), 5000, make_shared<dielectric>(1.5));
    world.add(make_shared<constant_medium>(boundary, .0001, color(1, 1, 1)));
    world.add(make_shared<sphere>(point3(400, 200, 400), 100, make_shared<lambertian>(make_shared<image_texture>("earthmap.jpg"))));
    world.add(make_shared<sphere>(point3(220, 280, 300), 80, make_shared<lambertian>(make_shared<noise_texture>(0.2))));
    hittable_list boxes2;
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    for (int j = 0; j < 1000; j++) boxes2.add(make_shared<sphere>(point3::random(0, 165), 10, white));
    shared_ptr<hittable> cluster;
    if (use_bvh) cluster = make_shared<bvh_node>(boxes2);
    else cluster = make_shared<hittable_list>(boxes2);
    world.add(make_shared<translate>(make_shared<rotate_y>(cluster, 15), vec3(-100, 270, 395)));
    wrap_bvh(world, use_bvh);
    s.lights.add(make_shared<quad>(point3(253, 554, 253), vec3(-300, 0, 0), vec3(0, 0, -265), shared_ptr<material>()));
    cornell_camera(s.cam, point3(478, 278, -600));
}

} // namespace

bool build_scene(int scene, bool use_bvh, int frame, const std::string &asset_dir, BuiltScene &out) {
    if (!asset_dir.empty()) setenv("RTW_IMAGES", asset_dir.c_str(), 1); // texture search path (stb_image_utils.h:25)
    switch (scene) {
    case 1: first_scene(out, use_bvh); return true;
    case 2: cornell_box(out, use_bvh); return true;
    case 3: final_scene(out, use_bvh); return true;
    case 4: mesh_scene(out, use_bvh, asset_dir); return true;
    case 5: bounce_frame(out, use_bvh, frame); return true;
    case 6: cornell_smoke(out, use_bvh); return true;
    }
    return false;
}

} // namespace rtamd
