// png_decode.h -- PNG decoder for image_texture files (asset pipeline next to rtamd/jpeg_baseline.h).
//
// The reference reads textures through the vendored stb_image v2.06 with three requested components
// (stb_image_utils.h:49); this decoder returns the same 8-bit RGB: 1/2/4/8-bit grey (scaled to 0..255), grey + alpha,
// RGB, RGBA and palette images, Adam7 interlace, all five row filters; alpha and tRNS are dropped, not composited, a
// 16-bit file is refused (that stb version refuses it too), CRCs and the zlib checksum are not verified (nor are they
// there).  Inflate is RFC 1951 (stored, fixed and dynamic Huffman blocks).  Checked bit for bit against texels produced
// by the compiled reference (tests/test_png_decode.py).  Header-only, host code.
#ifndef RTAMD_PNG_DECODE_H
#define RTAMD_PNG_DECODE_H
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rtamd {
namespace png {

struct Image {
    int width = 0, height = 0;
    int file_components = 0;  // 1 grey, 2 grey + alpha, 3 RGB (also palette), 4 RGBA
    std::vector<uint8_t> rgb; // width * height * 3
};

namespace detail {

// ---- RFC 1951 inflate ---------------------------------------------------------------------------------------------
struct HuffTable { // canonical code, lengths 1..15
    uint16_t count[16] = {0}, symbol[288] = {0};
    bool build(const uint8_t *lengths, int n) {
        std::memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false; // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
};

struct Inflate {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    std::vector<uint8_t> &out;
    size_t limit;
    Inflate(const uint8_t *d, size_t len, std::vector<uint8_t> &o, size_t lim) : p(d), end(d + len), out(o), limit(lim) {}
    int bits(int k) { // LSB first
        while (n < k) {
            acc |= (uint32_t)(p < end ? *p++ : 0) << n;
            n += 8;
        }
        const int v = (int)(acc & ((1u << k) - 1u));
        acc >>= k;
        n -= k;
        return v;
    }
    int decode(const HuffTable &h) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= bits(1);
            const int c = h.count[l];
            if (code - c < first) return h.symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
    bool block(const HuffTable &lit, const HuffTable &dist) {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            const int s = decode(lit);
            if (s < 0) return false;
            if (s < 256) {
                if (out.size() >= limit) return false;
                out.push_back((uint8_t)s);
            } else if (s == 256) {
                return true;
            } else {
                if (s > 285) return false;
                const int len = lbase[s - 257] + bits(lext[s - 257]);
                const int d = decode(dist);
                if (d < 0 || d > 29) return false;
                const size_t back = (size_t)dbase[d] + (size_t)bits(dext[d]);
                if (back > out.size() || out.size() + (size_t)len > limit) return false;
                for (int k = 0; k < len; k++) out.push_back(out[out.size() - back]);
            }
        }
    }
    bool run() {
        int last;
        do {
            last = bits(1);
            const int type = bits(2);
            if (type == 0) {
                acc = 0;
                n = 0; // to the byte boundary
                if (end - p < 4) return false;
                const int len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
                p += 4;
                if ((len ^ 0xffff) != nlen || end - p < len || out.size() + (size_t)len > limit) return false;
                out.insert(out.end(), p, p + len);
                p += len;
            } else if (type == 1) {
                uint8_t l[288 + 30];
                int i = 0;
                for (; i < 144; i++) l[i] = 8;
                for (; i < 256; i++) l[i] = 9;
                for (; i < 280; i++) l[i] = 7;
                for (; i < 288; i++) l[i] = 8;
                for (i = 0; i < 30; i++) l[288 + i] = 5;
                HuffTable lit, dist;
                lit.build(l, 288);
                dist.build(l + 288, 30);
                if (!block(lit, dist)) return false;
            } else if (type == 2) {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlit = bits(5) + 257, ndist = bits(5) + 1, ncode = bits(4) + 4;
                if (nlit > 286 || ndist > 30) return false;
                uint8_t l[320];
                std::memset(l, 0, sizeof(l));
                for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)bits(3);
                HuffTable cl;
                if (!cl.build(l, 19)) return false;
                uint8_t lens[286 + 30];
                int i = 0;
                while (i < nlit + ndist) {
                    const int s = decode(cl);
                    if (s < 0) return false;
                    if (s < 16) {
                        lens[i++] = (uint8_t)s;
                    } else {
                        int rep, val = 0;
                        if (s == 16) {
                            if (i == 0) return false;
                            val = lens[i - 1];
                            rep = 3 + bits(2);
                        } else if (s == 17) {
                            rep = 3 + bits(3);
                        } else {
                            rep = 11 + bits(7);
                        }
                        if (i + rep > nlit + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                HuffTable lit, dist;
                if (!lit.build(lens, nlit) || !dist.build(lens + nlit, ndist)) return false;
                if (!block(lit, dist)) return false;
            } else {
                return false;
            }
        } while (!last);
        return true;
    }
};

inline bool zlib_inflate(const std::vector<uint8_t> &z, std::vector<uint8_t> &out, size_t limit) {
    if (z.size() < 2) return false;
    const int cmf = z[0], flg = z[1];
    if ((cmf * 256 + flg) % 31 != 0 || (flg & 32) || (cmf & 15) != 8) return false; // RFC 1950 header, no preset dictionary
    Inflate inf(z.data() + 2, z.size() - 2, out, limit);
    return inf.run();
}

// ---- PNG ----------------------------------------------------------------------------------------------------------
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

// Reverse the row filters of one (sub)image: raw = h * (1 + row_bytes) filtered bytes -> h * row_bytes bytes.
inline bool unfilter(const uint8_t *raw, size_t raw_len, int h, size_t row_bytes, int bpp, std::vector<uint8_t> &out) {
    if (raw_len < (size_t)h * (row_bytes + 1)) return false;
    out.assign((size_t)h * row_bytes, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t *in = raw + (size_t)y * (row_bytes + 1);
        const int f = *in++;
        if (f > 4) return false;
        uint8_t *cur = out.data() + (size_t)y * row_bytes;
        const uint8_t *up = y ? cur - row_bytes : nullptr;
        for (size_t k = 0; k < row_bytes; k++) {
            const int a = k >= (size_t)bpp ? cur[k - bpp] : 0, b = up ? up[k] : 0, c = (up && k >= (size_t)bpp) ? up[k - bpp] : 0;
            int v = in[k];
            switch (f) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: break;
            }
            cur[k] = (uint8_t)v;
        }
    }
    return true;
}

struct Decoder {
    std::string err;
    bool fail(const char *m) {
        if (err.empty()) err = m;
        return false;
    }
    static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

    bool run(const uint8_t *d, size_t n, Image &img) {
        static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
        if (n < 8 || std::memcmp(d, sig, 8) != 0) return fail("not a PNG file");
        size_t at = 8;
        uint32_t w = 0, h = 0;
        int depth = 0, color = 0, interlace = 0, comps = 0;
        bool have_hdr = false, done = false;
        uint8_t palette[256 * 3];
        std::memset(palette, 0, sizeof(palette));
        int pal_len = 0;
        std::vector<uint8_t> z;
        while (!done) {
            if (n - at < 8) return fail("truncated PNG");
            const uint32_t len = be32(d + at);
            const uint8_t *type = d + at + 4;
            at += 8;
            if (len > n - at) return fail("truncated PNG chunk");
            const uint8_t *body = d + at;
            auto is = [&](const char *t) { return std::memcmp(type, t, 4) == 0; };
            if (!have_hdr && !is("IHDR")) return fail("first chunk is not IHDR");
            if (is("IHDR")) {
                if (have_hdr || len != 13) return fail("bad IHDR");
                w = be32(body);
                h = be32(body + 4);
                depth = body[8];
                color = body[9];
                interlace = body[12];
                if (w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) return fail("bad image size");
                if (depth != 1 && depth != 2 && depth != 4 && depth != 8) return fail("PNG not supported: 1/2/4/8-bit only");
                if (color > 6 || color == 1 || color == 5) return fail("bad colour type");
                if (color == 3 ? false : (color != 0 && depth != 8)) return fail("bad bit depth for the colour type");
                if (body[10] != 0 || body[11] != 0 || interlace > 1) return fail("bad compression / filter / interlace method");
                comps = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;
                if ((uint64_t)w * h > (1u << 28)) return fail("image too large");
                have_hdr = true;
            } else if (is("PLTE")) {
                if (len > 256 * 3 || len % 3) return fail("bad PLTE");
                pal_len = (int)(len / 3);
                std::memcpy(palette, body, len);
            } else if (is("IDAT")) {
                if (color == 3 && pal_len == 0) return fail("no PLTE");
                z.insert(z.end(), body, body + len);
            } else if (is("IEND")) {
                done = true;
            } else if (is("CgBI")) {
                return fail("CgBI (iPhone) PNG is not supported");
            } else if (!(type[0] & 32)) {
                return fail("unknown critical chunk");
            } // other ancillary chunks (tRNS too: its alpha would be dropped anyway) are skipped
            at += (size_t)len + 4; // body + CRC
            if (at > n && !done) return fail("truncated PNG");
        }
        if (!have_hdr || z.empty()) return fail("no image data");
        const int bits_px = comps * depth, bpp = bits_px >= 8 ? bits_px / 8 : 1;
        auto row_bytes = [&](uint32_t pw) { return ((size_t)pw * bits_px + 7) >> 3; };
        // expected size of the filtered stream
        static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t want = 0;
        if (!interlace) want = (size_t)h * (row_bytes(w) + 1);
        else
            for (int p = 0; p < 7; p++) {
                const uint32_t pw = (w - xo[p] + xs[p] - 1) / xs[p], ph = (h - yo[p] + ys[p] - 1) / ys[p];
                if (pw && ph) want += (size_t)ph * (row_bytes(pw) + 1);
            }
        std::vector<uint8_t> raw;
        raw.reserve(std::min(want, (size_t)1 << 24)); // a header may claim any size: grow with the data that actually inflates
        if (!zlib_inflate(z, raw, want + 64) || raw.size() < want) return fail("corrupt compressed data");

        img.width = (int)w;
        img.height = (int)h;
        img.file_components = color == 3 ? 3 : comps;
        img.rgb.assign((size_t)w * h * 3, 0);
        static const uint8_t scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
        // write one decoded sub-image (pw x ph samples, unfiltered rows) into the RGB image
        auto emit = [&](const std::vector<uint8_t> &rows, uint32_t pw, uint32_t ph, int x0, int y0, int dx, int dy) {
            const size_t rb = row_bytes(pw);
            for (uint32_t y = 0; y < ph; y++) {
                const uint8_t *r = rows.data() + (size_t)y * rb;
                for (uint32_t x = 0; x < pw; x++) {
                    uint8_t s[4] = {0, 0, 0, 0};
                    if (depth == 8) {
                        for (int c = 0; c < comps; c++) s[c] = r[(size_t)x * comps + c];
                    } else { // one component per pixel (grey or palette index), MSB first
                        const size_t bit = (size_t)x * depth;
                        s[0] = (uint8_t)((r[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                        if (color == 0) s[0] = (uint8_t)(s[0] * scale[depth]);
                    }
                    uint8_t *o = img.rgb.data() + ((size_t)(y0 + y * dy) * w + (size_t)(x0 + x * dx)) * 3;
                    if (color == 3) {
                        o[0] = palette[s[0] * 3];
                        o[1] = palette[s[0] * 3 + 1];
                        o[2] = palette[s[0] * 3 + 2];
                    } else if (comps <= 2) {
                        o[0] = o[1] = o[2] = s[0];
                    } else {
                        o[0] = s[0];
                        o[1] = s[1];
                        o[2] = s[2];
                    }
                }
            }
        };
        std::vector<uint8_t> rows;
        if (!interlace) {
            if (!unfilter(raw.data(), raw.size(), (int)h, row_bytes(w), bpp, rows)) return fail("invalid filter");
            emit(rows, w, h, 0, 0, 1, 1);
        } else {
            size_t off = 0;
            for (int p = 0; p < 7; p++) {
                const uint32_t pw = (w - xo[p] + xs[p] - 1) / xs[p], ph = (h - yo[p] + ys[p] - 1) / ys[p];
                if (!pw || !ph) continue;
                const size_t len = (size_t)ph * (row_bytes(pw) + 1);
                if (!unfilter(raw.data() + off, raw.size() - off, (int)ph, row_bytes(pw), bpp, rows)) return fail("invalid filter");
                emit(rows, pw, ph, xo[p], yo[p], xs[p], ys[p]);
                off += len;
            }
        }
        return true;
    }
};

} // namespace detail

inline bool is_png(const uint8_t *d, size_t n) { return n >= 8 && d[0] == 137 && d[1] == 80 && d[2] == 78 && d[3] == 71; }

// Decode a PNG held in memory to 8-bit RGB.  Returns false and a message in *error for anything it cannot read.
inline bool decode(const uint8_t *data, size_t size, Image &out, std::string *error = nullptr) {
    detail::Decoder d;
    if (d.run(data, size, out)) return true;
    if (error) *error = d.err.empty() ? "corrupt PNG" : d.err;
    out = Image();
    return false;
}

} // namespace png
} // namespace rtamd
#endif
