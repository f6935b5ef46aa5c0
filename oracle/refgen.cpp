// refgen.cpp -- fixture generator linked against the REAL reference (Jiaqi1song/RayTracing src/cpu).
// TEST INFRASTRUCTURE ONLY; built by oracle/Makefile into oracle/_ref/refgen, only in a container that has
// /root/reference.  No reference source is copied: the reference's own translation unit
// (src/cpu/cpuRenderer.cpp) is #included from where it lies, with two preprocessor hooks:
//   * `private` -> `public`, so the harness can read the object graph the reference scene functions build;
//   * `renderer.render(world, lights)` -> a callback, so the harness gets the world/lights/camera the reference's
//     own scene functions (cpuRenderer.cpp:86-401) constructed, with the reference's own std::rand stream.
// Modes (one per process):
//   dump   : serialise the reference object graph + camera frame into the flat format of include/rt_amd.h
//   trace  : deterministic ray batches through the reference's world.hit(); records hit_record + primitive id
//   anim   : the reference's own first_scene(animation=true) -> camera::render_animation() (camera.h:46; its CPU build
//            always runs the 42-frame translate sequence, see camera.h:48,98); every frame's camera and int image
//   anim_rotate : the camera frames of the rotate + zoom sequence (camera.h:49-59, 93-96: dead code in the CPU build,
//            so the harness drives the reference's rotate()/zoom()/initialize() in that order itself)
//   render : the reference's pixel loop (get_ray / ray_color / write_color) single-threaded from srand(seed);
//            records the double-precision pixel sums and the int bytes; cross-checked against the PPM written by
//            the reference's real camera::render() in the same process.
#include <algorithm>
#include <array>
#include <assert.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <emmintrin.h>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits.h>
#include <limits>
#include <map>
#include <math.h>
#include <memory>
#include <omp.h>
#include <sstream>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unistd.h>
#include <vector>

#include "../include/rt_amd.h"

// ---- rand() interposition: lets the trace mode inject the uniform constant_medium::hit consumes -------------
static bool g_inject = false;
static int g_inject_value = 0;
extern "C" int rand(void) {
    if (g_inject) return g_inject_value;
    return (int)random(); // glibc rand() == (int)random(); same state, same default seed
}
extern "C" void srand(unsigned int seed) { srandom(seed); }

#define private public
#include "cpu/utils.h"
// every writePPMImage() call inside camera.h (render: 189, render_animation: 91, 151) goes through the harness, which
// records the camera frame and the int image of each animation frame the reference's own loop produces
class camera;
static void on_write_ppm(camera *cam, const Image *image, const char *path);
#define writePPMImage(img, path) on_write_ppm(this, (img), (path))
#include "cpu/camera.h"
#undef writePPMImage
#include "cpu/bvh.h"
#include "cpu/constant_medium.h"
#include "cpu/hittable_list.h"
#include "cpu/material.h"
#include "cpu/quad.h"
#include "cpu/sphere.h"
#include "cpu/texture.h"
#include "cpu/triangle.h"
#undef private

static void on_render(camera &cam, const hittable &world, const hittable &lights);

#define render(w, l) max_depth, on_render(renderer, (w), (l))
#define main ref_main
#include "cpu/cpuRenderer.cpp"
#undef main
#undef render

// ---- serialiser ------------------------------------------------------------------------------------------------
struct Flat {
    std::vector<rt_node> nodes;
    std::vector<int32_t> child_index;
    std::vector<rt_sphere> spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_triangle> triangles;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> image_bytes;
    std::vector<rt_perlin> perlins;
    std::map<const hittable *, int> node_of;
    std::map<const material *, int> mat_of;
    std::map<const texture *, int> tex_of;
    std::vector<hittable *> prim_objs; // node id -> object (for material mutation in trace mode)
    int world_root = -1, lights_root = -1;

    static void put3(double *d, const vec3 &v) { d[0] = v.x(); d[1] = v.y(); d[2] = v.z(); }
    static void putbb(double *d, const aabb &b) {
        d[0] = b.x.min; d[1] = b.x.max; d[2] = b.y.min; d[3] = b.y.max; d[4] = b.z.min; d[5] = b.z.max;
    }

    int add_texture(const texture *t) {
        if (!t) return -1;
        auto it = tex_of.find(t);
        if (it != tex_of.end()) return it->second;
        int id = (int)textures.size();
        textures.push_back(rt_texture{});
        tex_of[t] = id;
        rt_texture r{};
        if (auto s = dynamic_cast<const solid_color *>(t)) {
            r.kind = RT_TEX_SOLID;
            put3(r.color, s->albedo);
        } else if (auto c = dynamic_cast<const checker_texture *>(t)) {
            r.kind = RT_TEX_CHECKER;
            r.scale = c->inv_scale;
            r.a = add_texture(c->even.get());
            r.b = add_texture(c->odd.get());
        } else if (auto im = dynamic_cast<const image_texture *>(t)) {
            r.kind = RT_TEX_IMAGE;
            if (im->image.height() <= 0) r.a = -1;
            else {
                rt_image ri{};
                ri.width = im->image.image_width;
                ri.height = im->image.image_height;
                ri.offset = (int64_t)image_bytes.size();
                size_t nb = (size_t)ri.width * ri.height * 3;
                image_bytes.insert(image_bytes.end(), im->image.bdata, im->image.bdata + nb);
                r.a = (int)images.size();
                images.push_back(ri);
            }
        } else if (auto nz = dynamic_cast<const noise_texture *>(t)) {
            r.kind = RT_TEX_NOISE;
            r.scale = nz->scale;
            rt_perlin p{};
            for (int i = 0; i < 256; i++) {
                put3(p.randvec[i], nz->noise.randvec[i]);
                p.perm_x[i] = nz->noise.perm_x[i];
                p.perm_y[i] = nz->noise.perm_y[i];
                p.perm_z[i] = nz->noise.perm_z[i];
            }
            r.a = (int)perlins.size();
            perlins.push_back(p);
        } else {
            fprintf(stderr, "refgen: unknown texture type\n");
            exit(2);
        }
        textures[id] = r;
        return id;
    }

    int add_material(const material *m) {
        if (!m) return -1;
        auto it = mat_of.find(m);
        if (it != mat_of.end()) return it->second;
        rt_material r{};
        r.tex = -1;
        if (auto l = dynamic_cast<const lambertian *>(m)) {
            r.kind = RT_MAT_LAMBERTIAN;
            r.tex = add_texture(l->tex.get());
        } else if (auto mt = dynamic_cast<const metal *>(m)) {
            r.kind = RT_MAT_METAL;
            put3(r.albedo, mt->albedo);
            r.fuzz = mt->fuzz;
        } else if (auto d = dynamic_cast<const dielectric *>(m)) {
            r.kind = RT_MAT_DIELECTRIC;
            r.ior = d->refraction_index;
        } else if (auto dl = dynamic_cast<const diffuse_light *>(m)) {
            r.kind = RT_MAT_DIFFUSE_LIGHT;
            r.tex = add_texture(dl->tex.get());
        } else if (auto iso = dynamic_cast<const isotropic *>(m)) {
            r.kind = RT_MAT_ISOTROPIC;
            r.tex = add_texture(iso->tex.get());
        } else {
            fprintf(stderr, "refgen: unknown material type\n");
            exit(2);
        }
        int id = (int)materials.size();
        materials.push_back(r);
        mat_of[m] = id;
        return id;
    }

    int add_node(const hittable *h) {
        auto it = node_of.find(h);
        if (it != node_of.end()) return it->second;
        int id = (int)nodes.size();
        nodes.push_back(rt_node{});
        prim_objs.push_back(const_cast<hittable *>(h));
        node_of[h] = id;
        rt_node n{};
        putbb(n.bbox, h->bounding_box());
        if (auto b = dynamic_cast<const bvh_node *>(h)) {
            n.kind = RT_NODE_BVH;
            n.a = add_node(b->left.get());
            n.b = add_node(b->right.get());
        } else if (auto l = dynamic_cast<const hittable_list *>(h)) {
            n.kind = RT_NODE_LIST;
            std::vector<int> kids;
            for (auto &o : l->objects) kids.push_back(add_node(o.get()));
            n.a = (int)child_index.size();
            n.b = (int)kids.size();
            child_index.insert(child_index.end(), kids.begin(), kids.end());
        } else if (auto t = dynamic_cast<const translate *>(h)) {
            n.kind = RT_NODE_TRANSLATE;
            put3(n.p, t->offset);
            n.a = add_node(t->object.get());
        } else if (auto r = dynamic_cast<const rotate_y *>(h)) {
            n.kind = RT_NODE_ROTATE_Y;
            n.p[0] = r->sin_theta;
            n.p[1] = r->cos_theta;
            n.a = add_node(r->object.get());
        } else if (auto m = dynamic_cast<const constant_medium *>(h)) {
            n.kind = RT_NODE_MEDIUM;
            n.p[0] = m->neg_inv_density;
            n.b = add_material(m->phase_function.get());
            n.a = add_node(m->boundary.get());
        } else if (auto s = dynamic_cast<const sphere *>(h)) {
            n.kind = RT_NODE_SPHERE;
            rt_sphere r{};
            put3(r.center, s->center.orig);
            put3(r.dcenter, s->center.dir);
            r.radius = s->radius;
            r.mat = add_material(s->mat.get());
            n.a = (int)spheres.size();
            spheres.push_back(r);
        } else if (auto q = dynamic_cast<const quad *>(h)) {
            n.kind = RT_NODE_QUAD;
            rt_quad r{};
            put3(r.Q, q->Q); put3(r.u, q->u); put3(r.v, q->v); put3(r.w, q->w); put3(r.normal, q->normal);
            r.D = q->D;
            r.area = q->area;
            r.mat = add_material(q->mat.get());
            n.a = (int)quads.size();
            quads.push_back(r);
        } else if (auto tr = dynamic_cast<const triangle *>(h)) {
            n.kind = RT_NODE_TRIANGLE;
            rt_triangle r{};
            put3(r.v0, tr->vertices[0]); put3(r.v1, tr->vertices[1]); put3(r.v2, tr->vertices[2]);
            put3(r.normal, tr->normal);
            r.mat = add_material(tr->mat.get());
            r.cull = tr->backCulling ? 1 : 0;
            n.a = (int)triangles.size();
            triangles.push_back(r);
        } else {
            fprintf(stderr, "refgen: unknown hittable type\n");
            exit(2);
        }
        nodes[id] = n;
        return id;
    }
};

template <class T> static void wr_vec(FILE *f, const std::vector<T> &v) {
    int64_t n = (int64_t)v.size();
    fwrite(&n, sizeof(n), 1, f);
    if (n) fwrite(v.data(), sizeof(T), v.size(), f);
}

static rt_camera camera_frame(const camera &c) {
    rt_camera k{};
    k.image_width = c.image_width;
    k.image_height = c.image_height;
    k.sqrt_spp = c.sqrt_spp;
    k.max_depth = c.max_depth;
    Flat::put3(k.center, c.center);
    Flat::put3(k.pixel00_loc, c.pixel00_loc);
    Flat::put3(k.pixel_delta_u, c.pixel_delta_u);
    Flat::put3(k.pixel_delta_v, c.pixel_delta_v);
    Flat::put3(k.defocus_disk_u, c.defocus_disk_u);
    Flat::put3(k.defocus_disk_v, c.defocus_disk_v);
    k.defocus_angle = c.defocus_angle;
    Flat::put3(k.background, c.background);
    return k;
}

static void write_scene(const char *path, const Flat &f, const rt_camera &cam) {
    FILE *fp = fopen(path, "wb");
    if (!fp) { perror(path); exit(2); }
    fwrite("RTSCENE1", 1, 8, fp);
    int32_t roots[2] = {f.world_root, f.lights_root};
    fwrite(roots, sizeof(int32_t), 2, fp);
    wr_vec(fp, f.nodes);
    wr_vec(fp, f.child_index);
    wr_vec(fp, f.spheres);
    wr_vec(fp, f.quads);
    wr_vec(fp, f.triangles);
    wr_vec(fp, f.materials);
    wr_vec(fp, f.textures);
    wr_vec(fp, f.images);
    wr_vec(fp, f.image_bytes);
    wr_vec(fp, f.perlins);
    fwrite(&cam, sizeof(cam), 1, fp);
    fclose(fp);
}

// ---- modes ------------------------------------------------------------------------------------------------
static std::string g_mode, g_out;
static unsigned g_seed = 1;
static int g_nrays_extra = 0;

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static uint32_t lcg32() { // own deterministic stream for ray batches (not std::rand)
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg_state >> 32);
}
static double lcg01() { return (lcg32() >> 1) / 2147483648.0; }

static void set_unique_materials(Flat &f, std::map<const material *, int> &prim_of_mat) {
    for (size_t id = 0; id < f.nodes.size(); id++) {
        hittable *h = f.prim_objs[id];
        int kind = f.nodes[id].kind;
        auto fresh = [&]() {
            auto m = make_shared<lambertian>(color(0.5, 0.5, 0.5));
            prim_of_mat[m.get()] = (int)id;
            return m;
        };
        if (kind == RT_NODE_SPHERE) static_cast<sphere *>(h)->mat = fresh();
        else if (kind == RT_NODE_QUAD) static_cast<quad *>(h)->mat = fresh();
        else if (kind == RT_NODE_TRIANGLE) static_cast<triangle *>(h)->mat = fresh();
        else if (kind == RT_NODE_MEDIUM) prim_of_mat[static_cast<constant_medium *>(h)->phase_function.get()] = (int)id;
    }
}

struct AnimFrame {
    rt_camera cam;
    std::vector<int32_t> bytes;
};
static std::vector<AnimFrame> g_anim;

static void on_write_ppm(camera *cam, const Image *image, const char *path) {
    if (g_mode != "anim") {
        writePPMImage(image, path);
        return;
    }
    AnimFrame fr;
    fr.cam = camera_frame(*cam);
    fr.bytes.assign(image->data, image->data + (size_t)3 * image->width * image->height);
    g_anim.push_back(std::move(fr));
}

static void write_anim(bool with_images) {
    FILE *fp = fopen(g_out.c_str(), "wb");
    if (!fp) { perror(g_out.c_str()); exit(2); }
    fwrite("RTANIM01", 1, 8, fp);
    int32_t hdr[4] = {(int32_t)g_anim.size(), g_anim.empty() ? 0 : g_anim[0].cam.image_width,
                      g_anim.empty() ? 0 : g_anim[0].cam.image_height, with_images ? 1 : 0};
    fwrite(hdr, sizeof(int32_t), 4, fp);
    for (auto &fr : g_anim) fwrite(&fr.cam, sizeof(rt_camera), 1, fp);
    if (with_images)
        for (auto &fr : g_anim) fwrite(fr.bytes.data(), sizeof(int32_t), fr.bytes.size(), fp);
    fclose(fp);
    fprintf(stderr, "refgen: %zu animation frames\n", g_anim.size());
}

static void on_render(camera &cam, const hittable &world, const hittable &lights) {
    cam.use_openmp = false;
    if (g_mode == "anim_rotate") { // camera.h:49-59, 93-96
        vec3 direction = cam.lookfrom - cam.lookat;
        double theta = std::acos(direction.y() / direction.length());
        double phi = std::atan2(direction.x(), direction.z());
        for (int frame = 0; frame < cam.animation_sample; frame++) {
            cam.rotate(theta, phi);
            cam.zoom(1.005);
            cam.initialize();
            g_anim.push_back(AnimFrame{camera_frame(cam), {}});
            phi += delta_phi;
            if (phi >= 2 * pi) phi -= 2 * pi;
            theta -= 0.01;
            if (theta <= 0) theta = 0;
        }
        write_anim(false);
        return;
    }
    cam.initialize();
    Flat f;
    f.world_root = f.add_node(&world);
    f.lights_root = f.add_node(&lights);
    rt_camera kc = camera_frame(cam);

    if (g_mode == "dump") {
        write_scene(g_out.c_str(), f, kc);
        fprintf(stderr, "refgen: %zu nodes, %zu spheres, %zu quads, %zu triangles, %zu materials, %zu textures\n",
                f.nodes.size(), f.spheres.size(), f.quads.size(), f.triangles.size(), f.materials.size(),
                f.textures.size());
        return;
    }

    if (g_mode == "trace" || g_mode == "boxtrace") {
        std::map<const material *, int> prim_of_mat;
        set_unique_materials(f, prim_of_mat);
        std::vector<rt_ray> rays;
        std::vector<double> xis;
        std::vector<rt_hit> hits;
        const int W = cam.image_width, H = cam.image_height;
        // batch A: pixel-centre primary rays (un-normalised direction, like get_ray without jitter)
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                vec3 ps = cam.pixel00_loc + (double(i) * cam.pixel_delta_u) + (double(j) * cam.pixel_delta_v);
                vec3 d = ps - cam.center;
                rt_ray r{};
                Flat::put3(r.orig, cam.center);
                Flat::put3(r.dir, d);
                r.time = lcg01();
                rays.push_back(r);
            }
        size_t n_primary = rays.size();
        auto trace_one = [&](const rt_ray &rr, double &xi_out) {
            int rv = (int)(lcg32() >> 1);
            if (rv == 0) rv = 1;
            g_inject = true;
            g_inject_value = rv;
            xi_out = rv / (RAND_MAX + 1.0);
            ray r(point3(rr.orig[0], rr.orig[1], rr.orig[2]), vec3(rr.dir[0], rr.dir[1], rr.dir[2]), rr.time);
            hit_record rec;
            rec.u = rec.v = 0;
            rt_hit h{};
            if (world.hit(r, interval(0.001, infinity), rec)) {
                auto it = prim_of_mat.find(rec.mat.get());
                if (it == prim_of_mat.end()) { fprintf(stderr, "refgen: unmapped material\n"); exit(2); }
                h.prim = it->second;
                h.front_face = rec.front_face ? 1 : 0;
                h.t = rec.t;
                Flat::put3(h.p, rec.p);
                Flat::put3(h.normal, rec.normal);
                h.u = rec.u;
                h.v = rec.v;
                if (f.nodes[h.prim].kind == RT_NODE_MEDIUM) { h.u = 0; h.v = 0; } // stale/uninitialised in the reference
            } else {
                h.prim = -1;
                h.t = infinity;
            }
            g_inject = false;
            return h;
        };
        if (g_mode == "boxtrace") {
            // batch C: rays through the corners, edges and faces of box() lists that stand in world space (final_scene's
            // ground, cpuRenderer.cpp:298-312).  ray.time carries the class of the target: .1 corner, .2 edge midpoint
            // (both exactly ON a line two or three faces share: equal-distance candidates), .3 face centre, .4 a point
            // 1/1000 of the box's size inside a face next to an edge (no tie: the face must win).
            rays.clear();
            std::vector<int> boxes;
            for (size_t id = 0; id < f.nodes.size(); id++) {
                const rt_node &n = f.nodes[id];
                if (n.kind != RT_NODE_LIST || n.b != 6) continue;
                bool all_quads = true;
                for (int k = 0; k < 6; k++) all_quads = all_quads && f.nodes[f.child_index[n.a + k]].kind == RT_NODE_QUAD;
                if (all_quads) boxes.push_back((int)id);
            }
            const size_t step = boxes.size() > 64 ? boxes.size() / 64 : 1;
            for (size_t bi = 0; bi < boxes.size(); bi += step) {
                const rt_node &n = f.nodes[boxes[bi]];
                double lo[3] = {infinity, infinity, infinity}, hi[3] = {-infinity, -infinity, -infinity};
                for (int k = 0; k < 6; k++) {
                    const rt_quad &q = f.quads[f.nodes[f.child_index[n.a + k]].a];
                    for (int c = 0; c < 4; c++)
                        for (int a = 0; a < 3; a++) {
                            double x = q.Q[a] + ((c & 1) ? q.u[a] : 0.0) + ((c & 2) ? q.v[a] : 0.0);
                            lo[a] = std::fmin(lo[a], x);
                            hi[a] = std::fmax(hi[a], x);
                        }
                }
                const double mid[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
                const double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
                std::vector<std::pair<std::array<double, 3>, double>> targets;
                for (int c = 0; c < 27; c++) { // every combination of {lo, mid, hi} per axis except the centre of the box
                    const int s[3] = {c % 3, (c / 3) % 3, c / 9};
                    const int n_mid = (s[0] == 1) + (s[1] == 1) + (s[2] == 1);
                    if (n_mid == 3) continue;
                    std::array<double, 3> p;
                    for (int a = 0; a < 3; a++) p[a] = s[a] == 0 ? lo[a] : (s[a] == 1 ? mid[a] : hi[a]);
                    targets.push_back({p, n_mid == 0 ? 0.1 : (n_mid == 1 ? 0.2 : 0.3)});
                    if (n_mid == 1) { // an edge midpoint: also the two points just inside the faces that meet there
                        for (int a = 0; a < 3; a++) {
                            if (s[a] == 1) continue;
                            std::array<double, 3> q = p; // move along axis a into the box: stays on the OTHER face
                            q[a] += (s[a] == 0 ? 1.0 : -1.0) * 1e-3 * ext[a];
                            targets.push_back({q, 0.4});
                        }
                    }
                }
                const point3 origins[2] = {cam.center, point3(mid[0] + 37.0, 700.0, mid[2] - 41.0)};
                for (const point3 &o : origins)
                    for (auto &t : targets) {
                        rt_ray r{};
                        Flat::put3(r.orig, o);
                        for (int a = 0; a < 3; a++) r.dir[a] = (t.first[a] - r.orig[a]) * 0.37;
                        r.time = t.second;
                        rays.push_back(r);
                    }
            }
            n_primary = rays.size();
        }
        for (size_t k = 0; k < n_primary; k++) {
            double xi;
            hits.push_back(trace_one(rays[k], xi));
            xis.push_back(xi);
        }
        if (g_mode == "boxtrace") g_nrays_extra = 0;
        // batch B: secondary-like rays leaving primary hit points in arbitrary directions (inside objects, grazing, ...)
        for (int k = 0; k < g_nrays_extra; k++) {
            const rt_hit &h0 = hits[lcg32() % n_primary];
            if (h0.prim < 0) continue;
            rt_ray r{};
            for (int c = 0; c < 3; c++) r.orig[c] = h0.p[c];
            double len = 0.25 + 4.0 * lcg01();
            for (int c = 0; c < 3; c++) r.dir[c] = len * (2 * lcg01() - 1);
            r.time = lcg01();
            double xi;
            rt_hit h = trace_one(r, xi);
            rays.push_back(r);
            hits.push_back(h);
            xis.push_back(xi);
        }
        FILE *fp = fopen(g_out.c_str(), "wb");
        if (!fp) { perror(g_out.c_str()); exit(2); }
        fwrite("RTTRACE1", 1, 8, fp);
        int64_t n = (int64_t)rays.size();
        fwrite(&n, sizeof(n), 1, fp);
        fwrite(rays.data(), sizeof(rt_ray), rays.size(), fp);
        fwrite(xis.data(), sizeof(double), xis.size(), fp);
        fwrite(hits.data(), sizeof(rt_hit), hits.size(), fp);
        fclose(fp);
        size_t nh = 0;
        for (auto &h : hits) nh += h.prim >= 0;
        fprintf(stderr, "refgen: traced %zu rays, %zu hits\n", rays.size(), nh);
        return;
    }

    if (g_mode == "render" || g_mode == "render_nocheck") {
        // REFGEN_WINDOW="x0,y0,w,h" (render_nocheck only): just that pixel window of the camera's frame -- the reference's own
        // get_ray / ray_color on pixels (x0 + i, y0 + j) of the FULL-SIZE camera (a 64 x 64 window of the 1920 x 1080 frame
        // at tens of thousands of samples is what a pixel-level check of a headline frame can afford)
        int x0 = 0, y0 = 0, W = cam.image_width, H = cam.image_height;
        if (const char *w = getenv("REFGEN_WINDOW")) {
            if (g_mode != "render_nocheck" || sscanf(w, "%d,%d,%d,%d", &x0, &y0, &W, &H) != 4 || x0 < 0 || y0 < 0 || W < 1 || H < 1 ||
                x0 + W > cam.image_width || y0 + H > cam.image_height) {
                fprintf(stderr, "refgen: bad REFGEN_WINDOW\n");
                exit(2);
            }
        }
        std::vector<double> sums((size_t)W * H * 3);
        std::vector<int32_t> bytes((size_t)W * H * 3);
        srand(g_seed);
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) { // camera.h:172-185, single thread
                color pixel_color(0, 0, 0);
                for (int s_j = 0; s_j < cam.sqrt_spp; s_j++)
                    for (int s_i = 0; s_i < cam.sqrt_spp; s_i++) {
                        ray r = cam.get_ray(x0 + i, y0 + j, s_i, s_j);
                        pixel_color += cam.ray_color(r, cam.max_depth, world, lights);
                    }
                size_t o = 3 * ((size_t)j * W + i);
                sums[o] = pixel_color.x(); sums[o + 1] = pixel_color.y(); sums[o + 2] = pixel_color.z();
                int px[3];
                write_color(px, cam.pixel_samples_scale * pixel_color);
                bytes[o] = px[0]; bytes[o + 1] = px[1]; bytes[o + 2] = px[2];
            }
        // cross-check against the reference's real camera::render() + writePPMImage()
        if (g_mode == "render") {
        srand(g_seed);
        cam.filename = "refgen_check.ppm";
        cam.render(world, lights);
        FILE *pp = fopen("./images/refgen_check.ppm", "rb");
        if (!pp) { perror("refgen_check.ppm"); exit(2); }
        char magic[8];
        int w2, h2, mx;
        if (fscanf(pp, "%7s %d %d %d", magic, &w2, &h2, &mx) != 4 || w2 != W || h2 != H) { fprintf(stderr, "refgen: bad ppm\n"); exit(2); }
        size_t bad = 0;
        for (size_t k = 0; k < bytes.size(); k++) {
            int vv;
            if (fscanf(pp, "%d", &vv) != 1) { fprintf(stderr, "refgen: short ppm\n"); exit(2); }
            bad += (vv != bytes[k]);
        }
        fclose(pp);
        if (bad) { fprintf(stderr, "refgen: harness loop differs from camera::render() in %zu values\n", bad); exit(3); }
        }
        FILE *fp = fopen(g_out.c_str(), "wb");
        if (!fp) { perror(g_out.c_str()); exit(2); }
        fwrite("RTIMG001", 1, 8, fp);
        int32_t hdr[4] = {W, H, cam.sqrt_spp, cam.max_depth};
        fwrite(hdr, sizeof(int32_t), 4, fp);
        uint32_t seed = g_seed;
        fwrite(&seed, sizeof(seed), 1, fp);
        fwrite(sums.data(), sizeof(double), sums.size(), fp);
        fwrite(bytes.data(), sizeof(int32_t), bytes.size(), fp);
        fclose(fp);
        fprintf(stderr, "refgen: rendered %dx%d spp=%d^2 depth=%d, harness loop == camera::render() PPM\n", W, H,
                cam.sqrt_spp, cam.max_depth);
        return;
    }
    fprintf(stderr, "refgen: unknown mode %s\n", g_mode.c_str());
    exit(2);
}

// Scene 6, "cornell smoke": not one of the reference's own scene functions -- a scene a user of its API can write (it is
// the book's cornell_smoke pattern) -- built here with the REFERENCE's classes so that the reference's own
// constant_medium::hit runs boundary->hit() on boundaries that are not spheres: rotated + translated box() lists, a
// box() list in world space, a moving sphere.  raytracing_amd/host/scenes.cpp builds the same scene with the drop-in API.
static void cornell_smoke(int image_width, double aspect_ratio, int samples_per_pixel, int max_depth, bool use_bvh) {
    hittable_list world;
    auto red = make_shared<lambertian>(color(.65, .05, .05));
    auto white = make_shared<lambertian>(color(.73, .73, .73));
    auto green = make_shared<lambertian>(color(.12, .45, .15));
    auto light = make_shared<diffuse_light>(color(7, 7, 7));
    world.add(make_shared<quad>(point3(555, 0, 0), vec3(0, 0, 555), vec3(0, 555, 0), green));
    world.add(make_shared<quad>(point3(0, 0, 555), vec3(0, 0, -555), vec3(0, 555, 0), red));
    world.add(make_shared<quad>(point3(0, 555, 0), vec3(555, 0, 0), vec3(0, 0, 555), white));
    world.add(make_shared<quad>(point3(0, 0, 555), vec3(555, 0, 0), vec3(0, 0, -555), white));
    world.add(make_shared<quad>(point3(555, 0, 555), vec3(-555, 0, 0), vec3(0, 555, 0), white));
    world.add(make_shared<quad>(point3(113, 554, 127), vec3(330, 0, 0), vec3(0, 0, 305), light));
    shared_ptr<hittable> box1 = box(point3(0, 0, 0), point3(165, 330, 165), white);
    box1 = make_shared<rotate_y>(box1, 15);
    box1 = make_shared<translate>(box1, vec3(265, 0, 295));
    shared_ptr<hittable> box2 = box(point3(0, 0, 0), point3(165, 165, 165), white);
    box2 = make_shared<rotate_y>(box2, -18);
    box2 = make_shared<translate>(box2, vec3(130, 0, 65));
    world.add(make_shared<constant_medium>(box1, 0.01, color(0, 0, 0)));
    world.add(make_shared<constant_medium>(box2, 0.01, color(1, 1, 1)));
    world.add(make_shared<constant_medium>(box(point3(30, 380, 300), point3(140, 470, 420), white), 0.02, color(.2, .4, .9)));
    world.add(make_shared<constant_medium>(make_shared<sphere>(point3(400, 420, 200), point3(430, 420, 200), 50, white), 0.05,
                                           color(.9, .6, .2)));
    world.add(make_shared<sphere>(point3(190, 90, 190), 60, make_shared<dielectric>(1.5)));
    if (use_bvh) world = hittable_list(make_shared<bvh_node>(world));
    auto empty_material = shared_ptr<material>();
    hittable_list lights;
    lights.add(make_shared<quad>(point3(443, 554, 432), vec3(-330, 0, 0), vec3(0, 0, -305), empty_material));
    camera renderer;
    renderer.aspect_ratio = aspect_ratio;
    renderer.image_width = image_width;
    renderer.samples_per_pixel = samples_per_pixel;
    renderer.max_depth = max_depth;
    renderer.background = color(0, 0, 0);
    renderer.use_bvh = use_bvh;
    renderer.vfov = 40;
    renderer.lookfrom = point3(278, 278, -800);
    renderer.lookat = point3(278, 278, 0);
    renderer.vup = vec3(0, 1, 0);
    renderer.defocus_angle = 0;
    on_render(renderer, world, lights);
}

// usage: refgen <mode> <scene 1..4|5(bounce frame)|6(cornell smoke)> <W> <H> <spp> <depth> <use_bvh 0|1> <out> [seed|extra_rays] [frame]
#undef main
int main(int argc, char **argv) {
    if (argc == 4 && std::string(argv[1]) == "teximg") { // refgen teximg <image file> <out>: the reference's rtw_image::load
        rtw_image img;                                   // (stb_image_utils.h:46-55) -> "RTWTEX01", w, h, linear RGB8
        if (!img.load(argv[2])) {
            fprintf(stderr, "refgen: the reference loader rejects %s\n", argv[2]);
            return 3;
        }
        FILE *f = fopen(argv[3], "wb");
        if (!f) return 2;
        const int32_t wh[2] = {img.image_width, img.image_height};
        fwrite("RTWTEX01", 1, 8, f);
        fwrite(wh, 4, 2, f);
        fwrite(img.bdata, 1, (size_t)wh[0] * wh[1] * 3, f);
        fclose(f);
        return 0;
    }
    if (argc < 9) {
        fprintf(stderr, "usage: refgen dump|trace|render scene W H spp depth use_bvh out [seed|extra_rays] [frame]\n");
        return 2;
    }
    g_mode = argv[1];
    std::clog.setstate(std::ios_base::failbit); // silence the reference's per-pixel progress output
    std::cout.setstate(std::ios_base::failbit);
    int scene = atoi(argv[2]), W = atoi(argv[3]), H = atoi(argv[4]), spp = atoi(argv[5]), depth = atoi(argv[6]);
    bool use_bvh = atoi(argv[7]) != 0;
    g_out = argv[8];
    if (argc > 9) {
        if (g_mode == "trace" || g_mode == "boxtrace") g_nrays_extra = atoi(argv[9]);
        else g_seed = (unsigned)atoi(argv[9]);
    }
    int frame = argc > 10 ? atoi(argv[10]) : 0;
    double aspect = double(W) / double(H); // cpuRenderer.cpp:437
    switch (scene) {
    case 1:
        if (g_mode == "anim") { // the reference's own animation path, from the process's default rand() seed
            first_scene(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, true, 1);
            write_anim(true);
            break;
        }
        first_scene(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, false, 0);
        break;
    case 2: cornell_box(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, false, 0); break;
    case 3: final_scene(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, false, 0); break;
    case 4: mesh_scene(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, false, 0); break;
    case 5: first_scene_moving(W, aspect, spp, depth, false, 1, "x.ppm", use_bvh, true, 2, 20, frame); break;
    case 6: cornell_smoke(W, aspect, spp, depth, use_bvh); break;
    default: fprintf(stderr, "bad scene\n"); return 2;
    }
    return 0;
}
