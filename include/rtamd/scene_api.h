// scene_api.h -- host-side, source-compatible mirror of the reference's scene API (Jiaqi1song/RayTracing src/cpu/*.h).
//
// A scene function written for the reference's CPU renderer (cpuRenderer.cpp:86-401) compiles unchanged against this
// header (via the forwarding headers in include/rtamd/compat/) and renders on the GPU: the classes below keep the
// reference's names, constructor signatures and public fields, build the same object graph (including the
// reference's host-side BVH: median split on the longest axis with std::sort, bvh.h:15-41), and camera::render()
// flattens that graph into the C structs of include/rt_amd.h and calls the HIP library.
//
// What is deliberately NOT here: hit()/scatter()/value() -- intersection, shading and sampling run in the HIP kernels
// (raytracing_amd/csrc/rt_kernels.h).  User-defined hittable/material subclasses therefore cannot be rendered.
//
// Everything that feeds geometry is IEEE double and follows the reference's arithmetic so that the flattened scene is
// byte-identical to one serialised from the reference's own object graph (tests/test_host_api.py).
#ifndef RTAMD_SCENE_API_H
#define RTAMD_SCENE_API_H

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>
#include <iterator>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../rt_amd.h"
#include "jpeg_baseline.h"
#include "png_decode.h"

using std::make_shared;
using std::shared_ptr;

// ---- utils.h ---------------------------------------------------------------------------------------------------------
const double infinity = std::numeric_limits<double>::infinity();
const double pi = 3.1415926535897932385;
const double delta_phi = 0.1;
const double phi_init = 0;
const double zoom_scale = 0.5;
const double step_scale = 0.5;

inline double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }
// Scene geometry of the reference is a function of the std::rand stream (utils.h:33-46): keep that generator.
inline double random_double() { return std::rand() / (RAND_MAX + 1.0); }
inline double random_double(double min, double max) { return min + (max - min) * random_double(); }
inline int random_int(int min, int max) { return int(random_double(min, max + 1)); }
inline double clamp(const double &n, const double &lower, const double &upper) { return std::max(lower, std::min(n, upper)); }

// ---- vec3.h ----------------------------------------------------------------------------------------------------------
class vec3 {
  public:
    double e[3];
    vec3() : e{0, 0, 0} {}
    vec3(double e0, double e1, double e2) : e{e0, e1, e2} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    double operator[](int i) const { return e[i]; }
    double &operator[](int i) { return e[i]; }
    vec3 &operator+=(const vec3 &v) { e[0] += v.e[0]; e[1] += v.e[1]; e[2] += v.e[2]; return *this; }
    vec3 &operator*=(double t) { e[0] *= t; e[1] *= t; e[2] *= t; return *this; }
    vec3 &operator/=(double t) { return *this *= 1 / t; }
    double length() const { return std::sqrt(length_squared()); }
    double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    bool near_zero() const { auto s = 1e-8; return (std::fabs(e[0]) < s) && (std::fabs(e[1]) < s) && (std::fabs(e[2]) < s); }
    // The reference writes vec3(random_double(), random_double(), random_double()) (vec3.h:51-57); the order in which a
    // compiler evaluates those arguments is unspecified.  The reference build the fixtures are pinned to (g++ -O2)
    // draws z, then y, then x -- made explicit here so every compiler builds the same scenes.
    static vec3 random() { double c = random_double(), b = random_double(), a = random_double(); return vec3(a, b, c); }
    static vec3 random(double min, double max) {
        double c = random_double(min, max), b = random_double(min, max), a = random_double(min, max);
        return vec3(a, b, c);
    }
};
using point3 = vec3;
using color = vec3;

inline vec3 operator+(const vec3 &u, const vec3 &v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3 &u, const vec3 &v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3 &u, const vec3 &v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator*(double t, const vec3 &v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3 &v, double t) { return t * v; }
inline vec3 operator/(const vec3 &v, double t) { return (1 / t) * v; }
inline double dot(const vec3 &u, const vec3 &v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
inline vec3 cross(const vec3 &u, const vec3 &v) {
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3 &v) { return v / v.length(); }
inline vec3 random_unit_vector() {
    while (true) {
        auto p = vec3::random(-1, 1);
        auto lensq = p.length_squared();
        if (1e-160 < lensq && lensq <= 1.0) return p / sqrt(lensq);
    }
}
inline vec3 reflect(const vec3 &v, const vec3 &n) { return v - 2 * dot(v, n) * n; }

// ---- interval.h / aabb.h / ray.h -------------------------------------------------------------------------------------
class interval {
  public:
    double min, max;
    interval() : min(+infinity), max(-infinity) {}
    interval(double min, double max) : min(min), max(max) {}
    interval(const interval &a, const interval &b) {
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    double size() const { return max - min; }
    bool contains(double x) const { return min <= x && x <= max; }
    bool surrounds(double x) const { return min < x && x < max; }
    double clamp(double x) const { if (x < min) return min; if (x > max) return max; return x; }
    interval expand(double delta) const { auto padding = delta / 2; return interval(min - padding, max + padding); }
    static const interval empty, universe;
};
inline const interval interval::empty = interval(+infinity, -infinity);
inline const interval interval::universe = interval(-infinity, +infinity);
inline interval operator+(const interval &ival, double displacement) { return interval(ival.min + displacement, ival.max + displacement); }
inline interval operator+(double displacement, const interval &ival) { return ival + displacement; }

class ray {
  public:
    ray() {}
    ray(const point3 &origin, const vec3 &direction, double time) : orig(origin), dir(direction), tm(time) {}
    ray(const point3 &origin, const vec3 &direction) : ray(origin, direction, 0) {}
    const point3 &origin() const { return orig; }
    const vec3 &direction() const { return dir; }
    double time() const { return tm; }
    point3 at(double t) const { return orig + t * dir; }

  private:
    point3 orig;
    vec3 dir;
    double tm;
};

class aabb {
  public:
    interval x, y, z;
    aabb() {}
    aabb(const interval &x, const interval &y, const interval &z) : x(x), y(y), z(z) { pad_to_minimums(); }
    aabb(const point3 &a, const point3 &b) {
        x = (a[0] <= b[0]) ? interval(a[0], b[0]) : interval(b[0], a[0]);
        y = (a[1] <= b[1]) ? interval(a[1], b[1]) : interval(b[1], a[1]);
        z = (a[2] <= b[2]) ? interval(a[2], b[2]) : interval(b[2], a[2]);
        pad_to_minimums();
    }
    aabb(const aabb &box0, const aabb &box1) { // union: not padded (aabb.h:24-28)
        x = interval(box0.x, box1.x);
        y = interval(box0.y, box1.y);
        z = interval(box0.z, box1.z);
    }
    const interval &axis_interval(int n) const { if (n == 1) return y; if (n == 2) return z; return x; }
    int longest_axis() const {
        if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
        else return y.size() > z.size() ? 1 : 2;
    }
    static const aabb empty, universe;

  private:
    void pad_to_minimums() { // aabb.h:74
        double delta = 0.0001;
        if (x.size() < delta) x = x.expand(delta);
        if (y.size() < delta) y = y.expand(delta);
        if (z.size() < delta) z = z.expand(delta);
    }
};
inline const aabb aabb::empty = aabb(interval::empty, interval::empty, interval::empty);
inline const aabb aabb::universe = aabb(interval::universe, interval::universe, interval::universe);
inline aabb operator+(const aabb &bbox, const vec3 &offset) { return aabb(bbox.x + offset.x(), bbox.y + offset.y(), bbox.z + offset.z()); }
inline aabb operator+(const vec3 &offset, const aabb &bbox) { return bbox + offset; }

// ---- flattening: object graph -> rt_scene_desc ------------------------------------------------------------------------
namespace rtamd {

class FlatScene {
  public:
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
    int world_root = -1, lights_root = -1;
    std::map<const void *, int> node_of, mat_of, tex_of;

    static void put3(double *d, const vec3 &v) { d[0] = v.x(); d[1] = v.y(); d[2] = v.z(); }
    static void putbb(double *d, const aabb &b) {
        d[0] = b.x.min; d[1] = b.x.max; d[2] = b.y.min; d[3] = b.y.max; d[4] = b.z.min; d[5] = b.z.max;
    }
    rt_scene_desc desc() const {
        rt_scene_desc d{};
        d.abi_version = RT_AMD_ABI_VERSION;
        d.world_root = world_root;
        d.lights_root = lights_root;
        d.n_nodes = (int32_t)nodes.size(); d.nodes = nodes.data();
        d.n_child_index = (int32_t)child_index.size(); d.child_index = child_index.data();
        d.n_spheres = (int32_t)spheres.size(); d.spheres = spheres.data();
        d.n_quads = (int32_t)quads.size(); d.quads = quads.data();
        d.n_triangles = (int32_t)triangles.size(); d.triangles = triangles.data();
        d.n_materials = (int32_t)materials.size(); d.materials = materials.data();
        d.n_textures = (int32_t)textures.size(); d.textures = textures.data();
        d.n_images = (int32_t)images.size(); d.images = images.data();
        d.n_image_bytes = (int64_t)image_bytes.size(); d.image_bytes = image_bytes.data();
        d.n_perlins = (int32_t)perlins.size(); d.perlins = perlins.data();
        return d;
    }
    // "RTSCENE1" container (same as oracle/refgen.cpp and raytracing_amd/abi.py)
    bool save(const char *path, const rt_camera *cam) const {
        FILE *fp = std::fopen(path, "wb");
        if (!fp) return false;
        std::fwrite("RTSCENE1", 1, 8, fp);
        int32_t roots[2] = {world_root, lights_root};
        std::fwrite(roots, sizeof(int32_t), 2, fp);
        wr(fp, nodes); wr(fp, child_index); wr(fp, spheres); wr(fp, quads); wr(fp, triangles); wr(fp, materials);
        wr(fp, textures); wr(fp, images); wr(fp, image_bytes); wr(fp, perlins);
        if (cam) std::fwrite(cam, sizeof(*cam), 1, fp);
        std::fclose(fp);
        return true;
    }

  private:
    template <class T> static void wr(FILE *f, const std::vector<T> &v) {
        int64_t n = (int64_t)v.size();
        std::fwrite(&n, sizeof(n), 1, f);
        if (n) std::fwrite(v.data(), sizeof(T), v.size(), f);
    }
};

} // namespace rtamd

// ---- perlin.h / stb_image_utils.h / texture.h --------------------------------------------------------------------------
class perlin { // tables only; noise()/turb() run on the GPU
  public:
    perlin() { // perlin.h:6-14: same std::rand consumption order
        for (int i = 0; i < point_count; i++) randvec[i] = unit_vector(vec3::random(-1, 1));
        perlin_generate_perm(perm_x);
        perlin_generate_perm(perm_y);
        perlin_generate_perm(perm_z);
    }
    static const int point_count = 256;
    vec3 randvec[point_count];
    int perm_x[point_count], perm_y[point_count], perm_z[point_count];

  private:
    static void perlin_generate_perm(int *p) {
        for (int i = 0; i < point_count; i++) p[i] = i;
        for (int i = point_count - 1; i > 0; i--) { // perlin.h:66
            int target = random_int(0, i);
            int tmp = p[i];
            p[i] = p[target];
            p[target] = tmp;
        }
    }
};

// rtw_image (stb_image_utils.h:18-110): linearised RGB8 texels.  The reference decodes the file with the vendored
// stb_image (stbi_loadf: 8-bit samples -> linear float by pow(x/255, 2.2) -> byte by int(256 f)).  Here a JPEG is
// decoded by rtamd/jpeg_baseline.h (which reproduces that decoder's integer arithmetic), a PNG by rtamd/png_decode.h,
// and linearised the same way; any other format must be supplied pre-decoded by the same pipeline as "<name>.rtwtex"
// ("RTWTEX01", int32 w, int32 h, w*h*3 bytes) next to where the image would be.  Search path as the reference's.
class rtw_image {
  public:
    rtw_image() {}
    rtw_image(const char *image_filename) {
        auto filename = std::string(image_filename);
        auto imagedir = getenv("RTW_IMAGES");
        if (imagedir && load(std::string(imagedir) + "/" + image_filename)) return;
        if (load(filename)) return;
        std::string up = "images/";
        for (int k = 0; k < 7; k++) {
            if (load(up + filename)) return;
            up = "../" + up;
        }
        std::cerr << "ERROR: Could not load image file '" << image_filename << "'.\n";
    }
    bool load(const std::string &filename) { return load_encoded(filename) || load_predecoded(filename + ".rtwtex"); }
    bool load_encoded(const std::string &filename) {
        std::ifstream f(filename, std::ios::binary);
        if (!f.is_open()) return false;
        std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::vector<unsigned char> rgb;
        std::string why;
        int w = 0, h = 0;
        if (rtamd::jpeg::is_jpeg(file.data(), file.size())) {
            rtamd::jpeg::Image img;
            if (!rtamd::jpeg::decode(file.data(), file.size(), img, &why)) {
                std::cerr << "ERROR: '" << filename << "': " << why << ".\n";
                return false;
            }
            w = img.width, h = img.height, rgb = std::move(img.rgb);
        } else if (rtamd::png::is_png(file.data(), file.size())) {
            rtamd::png::Image img;
            if (!rtamd::png::decode(file.data(), file.size(), img, &why)) {
                std::cerr << "ERROR: '" << filename << "': " << why << ".\n";
                return false;
            }
            w = img.width, h = img.height, rgb = std::move(img.rgb);
        } else {
            return false;
        }
        // stb_image.h:1373 (ldr -> hdr, gamma 2.2, scale 1) then stb_image_utils.h:87-93 (float -> byte)
        unsigned char lin[256];
        const float gamma = 2.2f, scale = 1.0f;
        for (int v = 0; v < 256; v++) {
            const float value = (float)(std::pow(v / 255.0f, gamma) * scale);
            lin[v] = value <= 0.0 ? 0 : 1.0 <= value ? 255 : static_cast<unsigned char>(256.0 * value);
        }
        for (auto &t : rgb) t = lin[t];
        image_width = w;
        image_height = h;
        bdata = std::move(rgb);
        return true;
    }
    bool load_predecoded(const std::string &filename) {
        std::ifstream f(filename, std::ios::binary);
        if (!f.is_open()) return false;
        char magic[8];
        int32_t wh[2];
        f.read(magic, 8);
        f.read(reinterpret_cast<char *>(wh), 8);
        if (!f || std::memcmp(magic, "RTWTEX01", 8) != 0 || wh[0] <= 0 || wh[1] <= 0) return false;
        std::vector<unsigned char> data((size_t)wh[0] * wh[1] * 3);
        f.read(reinterpret_cast<char *>(data.data()), (std::streamsize)data.size());
        if (!f) return false;
        image_width = wh[0];
        image_height = wh[1];
        bdata = std::move(data);
        return true;
    }
    int width() const { return bdata.empty() ? 0 : image_width; }
    int height() const { return bdata.empty() ? 0 : image_height; }
    const std::vector<unsigned char> &bytes() const { return bdata; }

  private:
    std::vector<unsigned char> bdata;
    int image_width = 0, image_height = 0;
};

class texture {
  public:
    virtual ~texture() = default;
    virtual int flatten(rtamd::FlatScene &f) const = 0;

  protected:
    // reserve the id before visiting children, so ids follow first-encounter order
    int reserve(rtamd::FlatScene &f, bool &fresh) const {
        auto it = f.tex_of.find(this);
        if (it != f.tex_of.end()) { fresh = false; return it->second; }
        fresh = true;
        int id = (int)f.textures.size();
        f.textures.push_back(rt_texture{});
        f.tex_of[this] = id;
        return id;
    }
};

class solid_color : public texture {
  public:
    solid_color(const color &albedo) : albedo(albedo) {}
    solid_color(double red, double green, double blue) : solid_color(color(red, green, blue)) {}
    int flatten(rtamd::FlatScene &f) const override {
        bool fresh;
        int id = reserve(f, fresh);
        if (!fresh) return id;
        rt_texture r{};
        r.kind = RT_TEX_SOLID;
        rtamd::FlatScene::put3(r.color, albedo);
        f.textures[id] = r;
        return id;
    }

  private:
    color albedo;
};

class checker_texture : public texture {
  public:
    checker_texture(double scale, shared_ptr<texture> even, shared_ptr<texture> odd) : inv_scale(1.0 / scale), even(even), odd(odd) {}
    checker_texture(double scale, const color &c1, const color &c2)
        : checker_texture(scale, make_shared<solid_color>(c1), make_shared<solid_color>(c2)) {}
    int flatten(rtamd::FlatScene &f) const override {
        bool fresh;
        int id = reserve(f, fresh);
        if (!fresh) return id;
        rt_texture r{};
        r.kind = RT_TEX_CHECKER;
        r.scale = inv_scale;
        r.a = even->flatten(f);
        r.b = odd->flatten(f);
        f.textures[id] = r;
        return id;
    }

  private:
    double inv_scale;
    shared_ptr<texture> even, odd;
};

class image_texture : public texture {
  public:
    // The file is decoded on a helper thread while the scene function goes on building geometry (earthmap.jpg: 9 ms of
    // final_scene's 14 ms on the GPU box's host); flatten() -- the first reader -- waits for it.
    image_texture(const char *filename)
        : loading(std::async(std::launch::async, [name = std::string(filename)] { return rtw_image(name.c_str()); }).share()) {}
    int flatten(rtamd::FlatScene &f) const override {
        bool fresh;
        int id = reserve(f, fresh);
        if (!fresh) return id;
        const rtw_image &image = loading.get();
        rt_texture r{};
        r.kind = RT_TEX_IMAGE;
        if (image.height() <= 0) r.a = -1; // texture.h:62: no data -> solid cyan
        else {
            rt_image ri{};
            ri.width = image.width();
            ri.height = image.height();
            ri.offset = (int64_t)f.image_bytes.size();
            f.image_bytes.insert(f.image_bytes.end(), image.bytes().begin(), image.bytes().end());
            r.a = (int)f.images.size();
            f.images.push_back(ri);
        }
        f.textures[id] = r;
        return id;
    }

  private:
    std::shared_future<rtw_image> loading;
};

class noise_texture : public texture {
  public:
    noise_texture(double scale) : scale(scale) {}
    int flatten(rtamd::FlatScene &f) const override {
        bool fresh;
        int id = reserve(f, fresh);
        if (!fresh) return id;
        rt_texture r{};
        r.kind = RT_TEX_NOISE;
        r.scale = scale;
        rt_perlin p{};
        for (int i = 0; i < 256; i++) {
            rtamd::FlatScene::put3(p.randvec[i], noise.randvec[i]);
            p.perm_x[i] = noise.perm_x[i];
            p.perm_y[i] = noise.perm_y[i];
            p.perm_z[i] = noise.perm_z[i];
        }
        r.a = (int)f.perlins.size();
        f.perlins.push_back(p);
        f.textures[id] = r;
        return id;
    }

  private:
    perlin noise;
    double scale;
};

// ---- material.h ---------------------------------------------------------------------------------------------------------
class material {
  public:
    virtual ~material() = default;
    virtual rt_material describe(rtamd::FlatScene &f) const = 0;
    int flatten(rtamd::FlatScene &f) const {
        auto it = f.mat_of.find(this);
        if (it != f.mat_of.end()) return it->second;
        rt_material r = describe(f);
        int id = (int)f.materials.size();
        f.materials.push_back(r);
        f.mat_of[this] = id;
        return id;
    }
};
inline int rtamd_mat_id(rtamd::FlatScene &f, const shared_ptr<material> &m) { return m ? m->flatten(f) : -1; }

class lambertian : public material {
  public:
    lambertian(const color &albedo) : tex(make_shared<solid_color>(albedo)) {}
    lambertian(shared_ptr<texture> tex) : tex(tex) {}
    rt_material describe(rtamd::FlatScene &f) const override {
        rt_material r{};
        r.kind = RT_MAT_LAMBERTIAN;
        r.tex = tex->flatten(f);
        return r;
    }

  private:
    shared_ptr<texture> tex;
};

class metal : public material {
  public:
    metal(const color &albedo, double fuzz) : albedo(albedo), fuzz(fuzz < 1 ? fuzz : 1) {}
    rt_material describe(rtamd::FlatScene &) const override {
        rt_material r{};
        r.kind = RT_MAT_METAL;
        r.tex = -1;
        rtamd::FlatScene::put3(r.albedo, albedo);
        r.fuzz = fuzz;
        return r;
    }

  private:
    color albedo;
    double fuzz;
};

class dielectric : public material {
  public:
    dielectric(double refraction_index) : refraction_index(refraction_index) {}
    rt_material describe(rtamd::FlatScene &) const override {
        rt_material r{};
        r.kind = RT_MAT_DIELECTRIC;
        r.tex = -1;
        r.ior = refraction_index;
        return r;
    }

  private:
    double refraction_index;
};

class diffuse_light : public material {
  public:
    diffuse_light(shared_ptr<texture> tex) : tex(tex) {}
    diffuse_light(const color &emit) : tex(make_shared<solid_color>(emit)) {}
    rt_material describe(rtamd::FlatScene &f) const override {
        rt_material r{};
        r.kind = RT_MAT_DIFFUSE_LIGHT;
        r.tex = tex->flatten(f);
        return r;
    }

  private:
    shared_ptr<texture> tex;
};

class isotropic : public material {
  public:
    isotropic(const color &albedo) : tex(make_shared<solid_color>(albedo)) {}
    isotropic(shared_ptr<texture> tex) : tex(tex) {}
    rt_material describe(rtamd::FlatScene &f) const override {
        rt_material r{};
        r.kind = RT_MAT_ISOTROPIC;
        r.tex = tex->flatten(f);
        return r;
    }

  private:
    shared_ptr<texture> tex;
};

// ---- hittable.h ---------------------------------------------------------------------------------------------------------
class hittable {
  public:
    virtual ~hittable() = default;
    virtual aabb bounding_box() const = 0;
    // Append this object (and, first time only, its children) to the flat scene; returns its node id.
    int flatten(rtamd::FlatScene &f) const {
        auto it = f.node_of.find(this);
        if (it != f.node_of.end()) return it->second;
        int id = (int)f.nodes.size();
        f.nodes.push_back(rt_node{});
        f.node_of[this] = id;
        rt_node n{};
        rtamd::FlatScene::putbb(n.bbox, bounding_box());
        describe(f, n);
        f.nodes[id] = n;
        return id;
    }

  protected:
    virtual void describe(rtamd::FlatScene &f, rt_node &n) const = 0;
};

class translate : public hittable {
  public:
    translate(shared_ptr<hittable> object, const vec3 &offset) : object(object), offset(offset) { bbox = object->bounding_box() + offset; }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_TRANSLATE;
        rtamd::FlatScene::put3(n.p, offset);
        n.a = object->flatten(f);
    }

  private:
    shared_ptr<hittable> object;
    vec3 offset;
    aabb bbox;
};

class rotate_y : public hittable {
  public:
    rotate_y(shared_ptr<hittable> object, double angle) : object(object) { // hittable.h:78-108
        auto radians = degrees_to_radians(angle);
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        bbox = object->bounding_box();
        point3 min(infinity, infinity, infinity);
        point3 max(-infinity, -infinity, -infinity);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    auto x = i * bbox.x.max + (1 - i) * bbox.x.min;
                    auto y = j * bbox.y.max + (1 - j) * bbox.y.min;
                    auto z = k * bbox.z.max + (1 - k) * bbox.z.min;
                    auto newx = cos_theta * x + sin_theta * z;
                    auto newz = -sin_theta * x + cos_theta * z;
                    vec3 tester(newx, y, newz);
                    for (int c = 0; c < 3; c++) {
                        min[c] = std::fmin(min[c], tester[c]);
                        max[c] = std::fmax(max[c], tester[c]);
                    }
                }
        bbox = aabb(min, max);
    }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_ROTATE_Y;
        n.p[0] = sin_theta;
        n.p[1] = cos_theta;
        n.a = object->flatten(f);
    }

  private:
    shared_ptr<hittable> object;
    double sin_theta, cos_theta;
    aabb bbox;
};

// ---- hittable_list.h ----------------------------------------------------------------------------------------------------
class hittable_list : public hittable {
  public:
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    hittable_list(shared_ptr<hittable> object) { add(object); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) {
        objects.push_back(object);
        bbox = aabb(bbox, object->bounding_box());
    }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_LIST;
        std::vector<int32_t> kids;
        kids.reserve(objects.size());
        for (auto &o : objects) kids.push_back(o->flatten(f));
        n.a = (int32_t)f.child_index.size();
        n.b = (int32_t)kids.size();
        f.child_index.insert(f.child_index.end(), kids.begin(), kids.end());
    }

  private:
    aabb bbox;
};

// ---- bvh.h: built on the host exactly as the reference builds it ----------------------------------------------------------
// The reference's build (bvh.h:13-41): union box of the range, longest axis, std::sort of the range by the boxes' minimum on
// that axis, split in the middle; span 1 => left = right, span 2 => the pair unsorted.  Same tree, same order of equal
// keys (std::sort is not stable: the outcome is libstdc++'s introsort order, which depends on the comparison results
// only), but sorted as (box, index) items: the reference's comparator copies two shared_ptrs and makes two virtual
// bounding_box() calls per comparison, which is most of its build time (4 968 triangles: 25 ms -> 2 ms).
class bvh_node : public hittable {
    struct Item {
        aabb box;
        size_t index; // into the caller's vector as it was on entry
    };

  public:
    bvh_node(hittable_list list) : bvh_node(list.objects, 0, list.objects.size()) {}
    bvh_node(std::vector<shared_ptr<hittable>> &objects, size_t start, size_t end) {
        std::vector<Item> items;
        items.reserve(end - start);
        for (size_t k = start; k < end; k++) items.push_back(Item{objects[k]->bounding_box(), k});
        const std::vector<shared_ptr<hittable>> orig(objects.begin() + start, objects.begin() + end);
        build(items, 0, items.size(), orig, start);
        // the reference sorts the caller's vector in place (nested sub-ranges): leave it in that final order
        for (size_t k = 0; k < items.size(); k++) objects[start + k] = orig[items[k].index - start];
    }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_BVH;
        n.a = left->flatten(f);
        n.b = right->flatten(f);
    }

  private:
    shared_ptr<hittable> left, right;
    aabb bbox;
    struct Built {}; // tag: a node whose fields build() fills
    explicit bvh_node(Built) {}
    void build(std::vector<Item> &items, size_t start, size_t end, const std::vector<shared_ptr<hittable>> &orig, size_t base) {
        bbox = aabb::empty;
        for (size_t k = start; k < end; k++) bbox = aabb(bbox, items[k].box);
        const int axis = bbox.longest_axis();
        const size_t object_span = end - start;
        if (object_span == 1) {
            left = right = orig[items[start].index - base];
        } else if (object_span == 2) {
            left = orig[items[start].index - base];
            right = orig[items[start + 1].index - base];
        } else {
            std::sort(items.begin() + start, items.begin() + end, [axis](const Item &a, const Item &b) {
                return a.box.axis_interval(axis).min < b.box.axis_interval(axis).min; // bvh.h:60-64
            });
            const size_t mid = start + object_span / 2;
            auto l = shared_ptr<bvh_node>(new bvh_node(Built{}));
            l->build(items, start, mid, orig, base);
            auto r = shared_ptr<bvh_node>(new bvh_node(Built{}));
            r->build(items, mid, end, orig, base);
            left = l;
            right = r;
        }
    }
};

// ---- sphere.h / quad.h / triangle.h / constant_medium.h -------------------------------------------------------------------
class sphere : public hittable {
  public:
    sphere(const point3 &static_center, double radius, shared_ptr<material> mat)
        : center(static_center, vec3(0, 0, 0)), radius(std::fmax(0, radius)), mat(mat) {
        auto rvec = vec3(radius, radius, radius);
        bbox = aabb(static_center - rvec, static_center + rvec);
    }
    sphere(const point3 &center1, const point3 &center2, double radius, shared_ptr<material> mat)
        : center(center1, center2 - center1), radius(std::fmax(0, radius)), mat(mat) {
        auto rvec = vec3(radius, radius, radius);
        aabb box1(center.at(0) - rvec, center.at(0) + rvec);
        aabb box2(center.at(1) - rvec, center.at(1) + rvec);
        bbox = aabb(box1, box2);
    }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_SPHERE;
        rt_sphere r{};
        rtamd::FlatScene::put3(r.center, center.origin());
        rtamd::FlatScene::put3(r.dcenter, center.direction());
        r.radius = radius;
        r.mat = rtamd_mat_id(f, mat);
        n.a = (int32_t)f.spheres.size();
        f.spheres.push_back(r);
    }

  private:
    ray center;
    double radius;
    shared_ptr<material> mat;
    aabb bbox;
};

class quad : public hittable {
  public:
    quad(const point3 &Q, const vec3 &u, const vec3 &v, shared_ptr<material> mat) : Q(Q), u(u), v(v), mat(mat) {
        auto n = cross(u, v);
        normal = unit_vector(n);
        D = dot(normal, Q);
        w = n / dot(n, n);
        area = n.length();
        set_bounding_box();
    }
    virtual void set_bounding_box() {
        auto bbox_diagonal1 = aabb(Q, Q + u + v);
        auto bbox_diagonal2 = aabb(Q + u, Q + v);
        bbox = aabb(bbox_diagonal1, bbox_diagonal2);
    }
    aabb bounding_box() const override { return bbox; }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_QUAD;
        rt_quad r{};
        rtamd::FlatScene::put3(r.Q, Q);
        rtamd::FlatScene::put3(r.u, u);
        rtamd::FlatScene::put3(r.v, v);
        rtamd::FlatScene::put3(r.w, w);
        rtamd::FlatScene::put3(r.normal, normal);
        r.D = D;
        r.area = area;
        r.mat = rtamd_mat_id(f, mat);
        n.a = (int32_t)f.quads.size();
        f.quads.push_back(r);
    }

  private:
    point3 Q;
    vec3 u, v;
    vec3 w;
    shared_ptr<material> mat;
    aabb bbox;
    vec3 normal;
    double D;
    double area;
};

inline shared_ptr<hittable_list> box(const point3 &a, const point3 &b, shared_ptr<material> mat) { // quad.h:103
    auto sides = make_shared<hittable_list>();
    auto min = point3(std::fmin(a.x(), b.x()), std::fmin(a.y(), b.y()), std::fmin(a.z(), b.z()));
    auto max = point3(std::fmax(a.x(), b.x()), std::fmax(a.y(), b.y()), std::fmax(a.z(), b.z()));
    auto dx = vec3(max.x() - min.x(), 0, 0);
    auto dy = vec3(0, max.y() - min.y(), 0);
    auto dz = vec3(0, 0, max.z() - min.z());
    sides->add(make_shared<quad>(point3(min.x(), min.y(), max.z()), dx, dy, mat));  // front
    sides->add(make_shared<quad>(point3(max.x(), min.y(), max.z()), -dz, dy, mat)); // right
    sides->add(make_shared<quad>(point3(max.x(), min.y(), min.z()), -dx, dy, mat)); // back
    sides->add(make_shared<quad>(point3(min.x(), min.y(), min.z()), dz, dy, mat));  // left
    sides->add(make_shared<quad>(point3(min.x(), max.y(), max.z()), dx, -dz, mat)); // top
    sides->add(make_shared<quad>(point3(min.x(), min.y(), min.z()), dx, dz, mat));  // bottom
    return sides;
}

class triangle : public hittable {
  public:
    triangle(const vec3 vs[3], shared_ptr<material> mat, bool cull = false) : mat(mat), backCulling(cull) {
        for (int i = 0; i < 3; i++) vertices[i] = vs[i];
        vec3 edge1 = vertices[1] - vertices[0];
        vec3 edge2 = vertices[2] - vertices[0];
        normal = unit_vector(cross(edge1, edge2));
    }
    aabb bounding_box() const override { // triangle.h:63-73: extents are rounded to float
        float minX = std::min(vertices[0].x(), std::min(vertices[1].x(), vertices[2].x()));
        float minY = std::min(vertices[0].y(), std::min(vertices[1].y(), vertices[2].y()));
        float minZ = std::min(vertices[0].z(), std::min(vertices[1].z(), vertices[2].z()));
        float maxX = std::max(vertices[0].x(), std::max(vertices[1].x(), vertices[2].x()));
        float maxY = std::max(vertices[0].y(), std::max(vertices[1].y(), vertices[2].y()));
        float maxZ = std::max(vertices[0].z(), std::max(vertices[1].z(), vertices[2].z()));
        return aabb(point3(minX, minY, minZ), point3(maxX, maxY, maxZ));
    }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_TRIANGLE;
        rt_triangle r{};
        rtamd::FlatScene::put3(r.v0, vertices[0]);
        rtamd::FlatScene::put3(r.v1, vertices[1]);
        rtamd::FlatScene::put3(r.v2, vertices[2]);
        rtamd::FlatScene::put3(r.normal, normal);
        r.mat = rtamd_mat_id(f, mat);
        r.cull = backCulling ? 1 : 0;
        n.a = (int32_t)f.triangles.size();
        f.triangles.push_back(r);
    }

  private:
    vec3 vertices[3];
    vec3 normal;
    shared_ptr<material> mat;
    bool backCulling;
};

inline void parse_obj(const char *filename, std::vector<vec3> &points, std::vector<vec3> &idxVertex, int &nPoints, int &nTriangles) {
    // stb_image_utils.h:112-172: "v x y z" / "f a b c" lines; centre on the vertex mean, divide by the (float) max radius
    std::ifstream objFile(filename);
    if (!objFile.is_open()) {
        std::cerr << "Error: Cannot open the OBJ file: " << filename << std::endl;
        return;
    }
    size_t np = 0, nt = 0;
    std::string line;
    while (std::getline(objFile, line)) {
        std::stringstream ss(line);
        std::string label;
        ss >> label;
        if (label == "v") {
            vec3 vertex;
            ss >> vertex[0] >> vertex[1] >> vertex[2];
            if (np >= points.size()) points.resize(2 * points.size() + 1); // the reference overruns its 5000 slots instead
            points[np++] = vertex;
        } else if (label == "f") {
            vec3 idx;
            ss >> idx[0] >> idx[1] >> idx[2];
            idx[0] -= 1; idx[1] -= 1; idx[2] -= 1;
            if (nt >= idxVertex.size()) idxVertex.resize(2 * idxVertex.size() + 1);
            idxVertex[nt++] = idx;
        }
    }
    nPoints = (int)np;
    nTriangles = (int)nt;
    vec3 mean = vec3(0, 0, 0);
    for (int i = 0; i < nPoints; i++) mean += points[i];
    mean /= float(nPoints);
    for (int i = 0; i < nPoints; i++) points[i] += -mean;
    float maxDistance = 0.0f;
    for (int i = 0; i < nPoints; i++) {
        float dist = points[i].length();
        if (dist > maxDistance) maxDistance = dist;
    }
    for (int i = 0; i < nPoints; i++) points[i] /= maxDistance;
}

inline hittable_list build_mesh(const char *filename, shared_ptr<material> mat, double scale, bool use_bvh) { // triangle.h:89
    hittable_list mesh_obj;
    std::vector<vec3> vertices(5000);
    std::vector<vec3> indices(5000);
    int nPoints = 0, nTriangles = 0;
    parse_obj(filename, vertices, indices, nPoints, nTriangles);
    for (vec3 &p : vertices) p *= scale;
    for (int i = 0; i < nTriangles; i++) {
        vec3 v0 = vertices[static_cast<int>(indices[i].x())];
        vec3 v1 = vertices[static_cast<int>(indices[i].y())];
        vec3 v2 = vertices[static_cast<int>(indices[i].z())];
        auto obj = make_shared<triangle>(std::array<vec3, 3>{v0, v1, v2}.data(), mat, false);
        mesh_obj.add(obj);
    }
    if (use_bvh && nTriangles > 0) mesh_obj = hittable_list(make_shared<bvh_node>(mesh_obj));
    return mesh_obj;
}

class constant_medium : public hittable {
  public:
    constant_medium(shared_ptr<hittable> boundary, double density, shared_ptr<texture> tex)
        : boundary(boundary), neg_inv_density(-1 / density), phase_function(make_shared<isotropic>(tex)) {}
    constant_medium(shared_ptr<hittable> boundary, double density, const color &albedo)
        : boundary(boundary), neg_inv_density(-1 / density), phase_function(make_shared<isotropic>(albedo)) {}
    aabb bounding_box() const override { return boundary->bounding_box(); }

  protected:
    void describe(rtamd::FlatScene &f, rt_node &n) const override {
        n.kind = RT_NODE_MEDIUM;
        n.p[0] = neg_inv_density;
        n.b = rtamd_mat_id(f, phase_function);
        n.a = boundary->flatten(f);
    }

  private:
    shared_ptr<hittable> boundary;
    double neg_inv_density;
    shared_ptr<material> phase_function;
};

// ---- color.h ------------------------------------------------------------------------------------------------------------
struct Image {
    Image(int w, int h) : width(w), height(h), data(new int[3 * (size_t)w * h]) {}
    ~Image() { delete[] data; }
    Image(const Image &) = delete;
    Image &operator=(const Image &) = delete;
    void clear(int r, int g, int b) {
        for (size_t i = 0; i < (size_t)width * height; i++) { data[3 * i] = r; data[3 * i + 1] = g; data[3 * i + 2] = b; }
    }
    int width, height;
    int *data;
};

// color.h:72: ASCII P3, byte for byte what the reference's fprintf("%d %d %d\n") loop writes -- formatted into one
// buffer and written with a single fwrite (a 1080p frame is 6.2 M numbers: 1.3 s through fprintf, 25 ms this way).
inline void writePPMImage(const Image *image, const char *filename) {
    FILE *fp = fopen(filename, "wb");
    if (!fp) {
        fprintf(stderr, "Error: could not open %s for write\n", filename);
        exit(1);
    }
    fprintf(fp, "P3\n");
    fprintf(fp, "%d %d\n", image->width, image->height);
    fprintf(fp, "255\n");
    // 6.2 M integers at 1080p: formatted into memory by up to four threads (one band of pixels each), written in order
    const size_t n = (size_t)image->width * image->height;
    const int n_bands = n >= (size_t)1 << 18 ? 4 : 1;
    std::vector<std::vector<char>> bands((size_t)n_bands);
    auto format_band = [&](int b) {
        const size_t k0 = n * (size_t)b / (size_t)n_bands, k1 = n * (size_t)(b + 1) / (size_t)n_bands;
        std::vector<char> &buf = bands[(size_t)b];
        buf.resize((k1 - k0) * 36 + 1); // an int takes at most 11 characters
        char *o = buf.data();
        auto put = [&](int v, char sep) {
            if (v >= 0 && v < 1000) {
                if (v >= 100) *o++ = char('0' + v / 100);
                if (v >= 10) *o++ = char('0' + (v / 10) % 10);
                *o++ = char('0' + v % 10);
            } else {
                o += snprintf(o, 12, "%d", v);
            }
            *o++ = sep;
        };
        for (size_t k = k0; k < k1; k++) {
            const int *ptr = &image->data[3 * k];
            put(ptr[0], ' ');
            put(ptr[1], ' ');
            put(ptr[2], '\n');
        }
        buf.resize((size_t)(o - buf.data()));
    };
    std::vector<std::thread> workers;
    for (int b = 1; b < n_bands; b++) workers.emplace_back(format_band, b);
    format_band(0);
    for (std::thread &t : workers) t.join();
    for (const std::vector<char> &buf : bands) fwrite(buf.data(), 1, buf.size(), fp);
    fclose(fp);
}

// ---- camera.h -----------------------------------------------------------------------------------------------------------
namespace rtamd {
// The GPU contexts of this process, created on first use: one per entry of $RTAMD_DEVICES ("0,1,2,3": a frame's strata are
// split over them by rt_render_multi; an index may repeat), else the single device $RTAMD_DEVICE (default 0).
inline std::vector<rt_ctx *> &sessions() {
    static std::vector<rt_ctx *> ctxs;
    if (ctxs.empty()) {
        std::vector<int> devs;
        if (const char *list = getenv("RTAMD_DEVICES")) {
            std::stringstream ss(list);
            std::string tok;
            while (std::getline(ss, tok, ','))
                if (!tok.empty()) devs.push_back(atoi(tok.c_str()));
        }
        if (devs.empty()) {
            const char *dev = getenv("RTAMD_DEVICE");
            devs.push_back(dev ? atoi(dev) : 0);
        }
        for (int d : devs) {
            rt_ctx *c = nullptr;
            int rc = rt_init(d, &c);
            if (rc != RT_OK) {
                for (rt_ctx *x : ctxs) rt_shutdown(x);
                ctxs.clear();
                throw std::runtime_error("rtamd: rt_init(" + std::to_string(d) + ") failed (status " + std::to_string(rc) + "): no usable MI355X/HIP device");
            }
            ctxs.push_back(c);
        }
    }
    return ctxs;
}
inline rt_ctx *session() { return sessions()[0]; }

// A flattened scene uploaded to every context of sessions(); freed with the object.
struct UploadedScene {
    std::vector<rt_scene *> scenes;
    double flatten_ms = 0, upload_ms = 0;
    UploadedScene() = default;
    UploadedScene(const UploadedScene &) = delete;
    UploadedScene &operator=(const UploadedScene &) = delete;
    ~UploadedScene() {
        for (rt_scene *s : scenes) rt_scene_free(s);
    }
};
// Wall-clock pieces of the last camera::render / render_animation frame (milliseconds).
struct FrameTiming {
    double flatten_ms = 0, upload_ms = 0, render_ms = 0 /* kernels + reduce + D2H, host-side wall clock */, write_ms = 0;
};
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
} // namespace rtamd

enum CameraMovement { FORWARD, BACKWARD, LEFT, RIGHT, UP, DOWN };

class camera {
  public:
    double aspect_ratio = 1.0;
    int image_width = 100;
    int samples_per_pixel = 10;
    int max_depth = 10;
    color background;
    bool use_openmp = false; // accepted for source compatibility; parallelism is the GPU's
    int num_threads = 8;
    Image *image = NULL;
    char filepath[1024];
    std::string filename = "image.ppm";
    double vfov = 90;
    point3 lookfrom = point3(0, 0, 0);
    point3 lookat = point3(0, 0, -1);
    vec3 vup = vec3(0, 1, 0);
    double defocus_angle = 0;
    double focus_dist = 10;
    int animation_sample = int(2 * pi / delta_phi);
    int animation_method = 0;
    bool use_bvh = true;
    // extensions (not in the reference): RNG seed, and statistics of the last render
    uint64_t seed = 1984;
    bool device_build = false; // see upload()
    rt_stats last_stats{};
    rtamd::FrameTiming last_timing{};
    bool write_file = true;

    // camera.h:157: one still frame to ./images/<filename>
    void render(const hittable &world, const hittable &lights) {
        initialize();
        snprintf(filepath, sizeof(filepath), "./images/%s", filename.c_str());
        rtamd::UploadedScene up;
        upload(world, lights, up);
        render_frame(up);
        write_frame();
    }

    // camera.h:46.  The reference tests `if (animation_method = 0) ... else if (animation_method = 1)` (assignments), so
    // its CPU build always runs the 42-frame translate sequence whatever was requested; method 0 below is the rotate +
    // zoom sequence that code was written to run (camera.h:48-97), method 1 the translate sequence (98-153).
    void render_animation(const hittable &world, const hittable &lights) {
        const int n = animation_frames();
        rtamd::UploadedScene up; // the world does not change between frames: flattened and uploaded once, only the camera moves
        upload(world, lights, up);
        for (int frame = 0; frame < n; frame++) {
            animation_step(frame);
            snprintf(filepath, sizeof(filepath), "./images/animation/image%d.ppm", frame);
            render_frame(up);
            write_frame();
        }
    }

    // The camera path of render_animation() without the rendering, for drivers that shard frames over GPUs
    // (frames are independent: frame f -> GPU f mod G): animation_frames() is the length of the sequence,
    // animation_step(f), called for f = 0, 1, 2, ... in order, moves the camera to frame f and returns its frame.
    int animation_frames() {
        if (animation_method != 0) animation_sample = 42; // camera.h:100
        return animation_sample;
    }
    rt_camera animation_step(int frame) {
        if (animation_method == 0) { // camera.h:49-59, 93-96
            if (frame == 0) {
                vec3 direction = lookfrom - lookat;
                anim_theta_ = std::acos(direction.y() / direction.length());
                anim_phi_ = std::atan2(direction.x(), direction.z());
            }
            rotate(anim_theta_, anim_phi_);
            zoom(1.005);
            initialize();
            anim_phi_ += delta_phi;
            if (anim_phi_ >= 2 * pi) anim_phi_ -= 2 * pi;
            anim_theta_ -= 0.01;
            if (anim_theta_ <= 0) anim_theta_ = 0;
        } else { // camera.h:99-119.  u, v, w are those of the previous initialize(): zero vectors on a fresh camera,
                 // so the first FORWARD step only lengthens focus_dist, as in the reference
            const double step = 0.6;
            if (frame < 7) translate(FORWARD, step);
            else if (frame < 14) translate(BACKWARD, step);
            else if (frame < 21) translate(LEFT, step);
            else if (frame < 28) translate(RIGHT, step);
            else if (frame < 35) translate(UP, step);
            else if (frame < 42) translate(DOWN, step);
            initialize();
        }
        return frame_;
    }

    // The derived frame (camera.h:205-247) in the C-ABI form; public so that drivers can shard work themselves.
    rt_camera frame() {
        initialize();
        return frame_;
    }
    void initialize() {
        image_height = int(image_width / aspect_ratio);
        image_height = (image_height < 1) ? 1 : image_height;
        if (!image || image->width != image_width || image->height != image_height) {
            image_owner_ = std::make_shared<Image>(image_width, image_height);
            image = image_owner_.get();
        }
        image->clear(0, 0, 0);
        sqrt_spp = int(std::sqrt(samples_per_pixel));
        center = lookfrom;
        auto theta = degrees_to_radians(vfov);
        auto h = std::tan(theta / 2);
        auto viewport_height = 2 * h * focus_dist;
        auto viewport_width = viewport_height * (double(image_width) / image_height);
        w = unit_vector(lookfrom - lookat);
        u = unit_vector(cross(vup, w));
        v = cross(w, u);
        vec3 viewport_u = viewport_width * u;
        vec3 viewport_v = viewport_height * -v;
        pixel_delta_u = viewport_u / image_width;
        pixel_delta_v = viewport_v / image_height;
        auto viewport_upper_left = center - (focus_dist * w) - viewport_u / 2 - viewport_v / 2;
        pixel00_loc = viewport_upper_left + 0.5 * (pixel_delta_u + pixel_delta_v);
        auto defocus_radius = focus_dist * std::tan(degrees_to_radians(defocus_angle / 2));
        defocus_disk_u = u * defocus_radius;
        defocus_disk_v = v * defocus_radius;
        frame_ = rt_camera{};
        frame_.image_width = image_width;
        frame_.image_height = image_height;
        frame_.sqrt_spp = sqrt_spp;
        frame_.max_depth = max_depth;
        rtamd::FlatScene::put3(frame_.center, center);
        rtamd::FlatScene::put3(frame_.pixel00_loc, pixel00_loc);
        rtamd::FlatScene::put3(frame_.pixel_delta_u, pixel_delta_u);
        rtamd::FlatScene::put3(frame_.pixel_delta_v, pixel_delta_v);
        rtamd::FlatScene::put3(frame_.defocus_disk_u, defocus_disk_u);
        rtamd::FlatScene::put3(frame_.defocus_disk_v, defocus_disk_v);
        frame_.defocus_angle = defocus_angle;
        rtamd::FlatScene::put3(frame_.background, background);
    }

    static void flatten(const hittable &world, const hittable &lights, rtamd::FlatScene &f) {
        f.world_root = world.flatten(f);
        f.lights_root = lights.flatten(f);
    }

  private:
    int image_height = 0;
    int sqrt_spp = 1;
    point3 center, pixel00_loc;
    vec3 pixel_delta_u, pixel_delta_v, u, v, w, defocus_disk_u, defocus_disk_v;
    rt_camera frame_{};
    double anim_theta_ = 0, anim_phi_ = 0;
    std::shared_ptr<Image> image_owner_; // `image` points into it; copies of a camera share the buffer until re-initialised

    // flatten world / lights and upload the result to every context of the process
    void upload(const hittable &world, const hittable &lights, rtamd::UploadedScene &up) {
        const double t0 = rtamd::now_ms();
        rtamd::FlatScene f;
        flatten(world, lights, f);
        rt_scene_desc d = f.desc();
        const double t1 = rtamd::now_ms();
        for (rt_ctx *ctx : rtamd::sessions()) {
            rt_scene *scene = nullptr;
            // device_build (extension field; $RTAMD_DEVICE_BUILD sets it too): the hierarchy of the production stream is built
            // on the GPU -- for scenes that are rebuilt per frame (the same stream as the host build's); an empty world falls back to the host path
            int rc = RT_ERR_UNSUPPORTED;
            if (device_build || getenv("RTAMD_DEVICE_BUILD")) rc = rt_scene_upload_device_build(ctx, &d, &scene);
            if (rc == RT_ERR_UNSUPPORTED) rc = rt_scene_upload(ctx, &d, &scene);
            if (rc != RT_OK) throw std::runtime_error(std::string("rtamd: rt_scene_upload: ") + rt_last_error(ctx));
            up.scenes.push_back(scene);
        }
        up.flatten_ms = t1 - t0;
        up.upload_ms = rtamd::now_ms() - t1;
    }
    // one frame of the current camera on the GPU(s) -> image->data
    void render_frame(const rtamd::UploadedScene &up) {
        const double t0 = rtamd::now_ms();
        std::vector<uint8_t> rgb((size_t)image_width * image_height * 3);
        rt_render_params p{};
        p.seed = seed;
        p.s_end = RT_ALL_STRATA;
        p.precision = RT_PREC_F32;
        std::vector<rt_ctx *> &ctxs = rtamd::sessions();
        int rc;
        if (ctxs.size() == 1) rc = rt_render(ctxs[0], up.scenes[0], &frame_, &p, nullptr, rgb.data(), &last_stats);
        else rc = rt_render_multi(ctxs.data(), up.scenes.data(), (int)ctxs.size(), &frame_, &p, nullptr, rgb.data(), &last_stats);
        if (rc != RT_OK) throw std::runtime_error(std::string("rtamd: rt_render: ") + rt_last_error(ctxs[0]));
        for (size_t k = 0; k < rgb.size(); k++) image->data[k] = rgb[k];
        last_timing.flatten_ms = up.flatten_ms;
        last_timing.upload_ms = up.upload_ms;
        last_timing.render_ms = rtamd::now_ms() - t0;
    }
    void write_frame() {
        const double t0 = rtamd::now_ms();
        if (write_file) writePPMImage(image, filepath);
        last_timing.write_ms = rtamd::now_ms() - t0;
    }

    void rotate(double theta, double phi) { // camera.h:328
        double radialDistance = (lookfrom - lookat).length();
        lookfrom = vec3(radialDistance * std::sin(theta) * std::sin(phi), radialDistance * std::cos(theta),
                        radialDistance * std::sin(theta) * std::cos(phi)) + lookat;
    }
    void zoom(double zoom_scale) { // camera.h:337
        vec3 direction = lookfrom - lookat;
        lookfrom = direction * zoom_scale + lookat;
        focus_dist *= zoom_scale;
    }
    void translate(CameraMovement direction, double step_scale) { // camera.h:343
        if (direction == FORWARD) { lookfrom += w * step_scale; lookat += w * step_scale; focus_dist += step_scale; }
        if (direction == BACKWARD) { lookfrom += w * -step_scale; lookat += w * -step_scale; focus_dist -= step_scale; }
        if (direction == LEFT) { lookfrom += u * -step_scale; lookat += u * -step_scale; }
        if (direction == RIGHT) { lookfrom += u * step_scale; lookat += u * step_scale; }
        if (direction == UP) { lookfrom += v * step_scale; lookat += v * step_scale; }
        if (direction == DOWN) { lookfrom += v * -step_scale; lookat += v * -step_scale; }
    }
};

#endif // RTAMD_SCENE_API_H
