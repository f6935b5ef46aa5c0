// jpeg_baseline.h -- Huffman JPEG decoder (baseline, extended sequential, progressive; 8-bit) for image_texture files
// (SURVEY §8f rank 4).
//
// The reference reads textures through the vendored stb_image (stb_image_utils.h:46-55: stbi_loadf -> linear float ->
// bytes).  The texels feed the render, so this decoder reproduces that pipeline's ARITHMETIC bit for bit -- the
// fixed-point inverse DCT (constants x4096, +512 >>10 after the column pass, +65536+(128<<17) >>17 after the row pass),
// the chroma upsampling filters (3:1 taps, centred), the 20-bit fixed-point YCbCr -> RGB conversion -- and is checked
// against texels produced by the compiled reference (tests/test_jpeg_decode.py, tests/golden/jpeg/).  What it does not
// read fails loudly: arithmetic-coded, lossless, 12-bit and CMYK files.
//
// Header-only, host code, no dependencies.  ITU T.81 section numbers are cited where a step follows the standard.
#ifndef RTAMD_JPEG_BASELINE_H
#define RTAMD_JPEG_BASELINE_H
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace rtamd {
namespace jpeg {

struct Image {
    int width = 0, height = 0;
    int file_components = 0;   // 1 (grey) or 3 (YCbCr) in the file
    std::vector<uint8_t> rgb;  // width * height * 3, sRGB-encoded bytes as stored
};

namespace detail {

// Canonical Huffman table (T.81 C.2, F.2.2.3): codes of length l occupy [first[l], first[l] + count[l]).
struct Huff {
    bool present = false;
    uint8_t count[17] = {0};
    uint8_t symbol[256] = {0};
    int32_t first_code[17] = {0}, first_index[17] = {0};
    // codes of up to 9 bits resolved by one lookup of the next 9 bits: length << 8 | symbol, 0 = longer code
    uint16_t fast[512] = {0};
    bool build(const uint8_t *counts16, const uint8_t *symbols, int n) {
        int code = 0, index = 0;
        for (int l = 1; l <= 16; l++) {
            count[l] = counts16[l - 1];
            first_code[l] = code;
            first_index[l] = index;
            code += count[l];
            if (count[l] && code - 1 >= (1 << l)) return false; // more codes than the length can hold
            index += count[l];
            code <<= 1;
        }
        if (index != n || n > 256) return false;
        std::memcpy(symbol, symbols, (size_t)n);
        std::memset(fast, 0, sizeof(fast));
        for (int l = 1; l <= 9; l++)
            for (int k = 0; k < count[l]; k++) {
                const int code = first_code[l] + k, lo = code << (9 - l);
                for (int j = 0; j < (1 << (9 - l)); j++) fast[lo + j] = (uint16_t)(l << 8 | symbol[first_index[l] + k]);
            }
        present = true;
        return true;
    }
};

// Entropy-coded segment reader: MSB-first bits, 0xFF00 unstuffing; once a marker is met the stream reads as zeros
// and the marker is kept for the scan loop.
struct Bits {
    const uint8_t *p = nullptr, *end = nullptr;
    uint32_t acc = 0;
    int n = 0;
    int marker = 0; // pending marker met inside the entropy-coded data (0 = none)
    void reset() { acc = 0; n = 0; marker = 0; }
    int byte() { return p < end ? *p++ : 0; }
    void fill() {
        while (n <= 24) {
            int b = marker ? 0 : byte();
            if (b == 0xff) {
                int c = byte();
                if (c != 0) { marker = c; b = 0; }
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int bit() {
        if (n < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1;
        n--;
        return b;
    }
    int raw(int s) { // s raw bits, MSB first (T.81 F.2.2.4 RECEIVE)
        if (s == 0) return 0;
        if (n < s) fill();
        int v = (int)(acc >> (32 - s));
        acc <<= s;
        n -= s;
        return v;
    }
    // T.81 F.2.2.1 RECEIVE + EXTEND
    int receive_extend(int s) {
        if (s == 0) return 0;
        if (n < s) fill();
        int v = (int)(acc >> (32 - s));
        acc <<= s;
        n -= s;
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    int decode(const Huff &h) { // T.81 F.2.2.3 DECODE; -1 = no such code
        if (n < 9) fill(); // (past the end of the data and behind a marker the stream reads as zeros, as bit() sees it)
        const uint16_t e = h.fast[acc >> 23];
        if (e) {
            acc <<= (e >> 8);
            n -= (e >> 8);
            return e & 255;
        }
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | bit();
            if (h.count[l] && code - h.first_code[l] < h.count[l] && code >= h.first_code[l])
                return h.symbol[h.first_index[l] + code - h.first_code[l]];
        }
        return -1;
    }
};

static const uint8_t k_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp_u8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

// One 8-point pass of the LL&M integer inverse DCT ("islow"), constants scaled by 4096 as the reference's decoder
// scales them (stb_image.h:1905-1944).  in: 8 samples; out: even part e[0..3] and odd part o[0..3], so that
// result[k] = e[k] + o[3-k] ... result[7-k] = e[k] - o[3-k] (k = 0..3) before rounding.
inline int fx(double x) { return (int)(x * 4096 + 0.5); }
inline void idct8(const int s[8], int e[4], int o[4]) {
    static const int c0541 = fx(0.5411961f), c1847 = fx(-1.847759065f), c0765 = fx(0.765366865f), c1175 = fx(1.175875602f),
                     c0298 = fx(0.298631336f), c2053 = fx(2.053119869f), c3072 = fx(3.072711026f), c1501 = fx(1.501321110f),
                     c0899 = fx(-0.899976223f), c2562 = fx(-2.562915447f), c1961 = fx(-1.961570560f), c0390 = fx(-0.390180644f);
    const int z = (s[2] + s[6]) * c0541;
    const int a2 = z + s[6] * c1847, a3 = z + s[2] * c0765;
    const int a0 = (s[0] + s[4]) * 4096, a1 = (s[0] - s[4]) * 4096;
    e[0] = a0 + a3;
    e[3] = a0 - a3;
    e[1] = a1 + a2;
    e[2] = a1 - a2;
    int b0 = s[7], b1 = s[5], b2 = s[3], b3 = s[1];
    int q3 = b0 + b2, q4 = b1 + b3, q1 = b0 + b3, q2 = b1 + b2;
    const int q5 = (q3 + q4) * c1175;
    b0 *= c0298;
    b1 *= c2053;
    b2 *= c3072;
    b3 *= c1501;
    q1 = q5 + q1 * c0899;
    q2 = q5 + q2 * c2562;
    q3 *= c1961;
    q4 *= c0390;
    o[3] = b3 + q1 + q4;
    o[2] = b2 + q2 + q3;
    o[1] = b1 + q2 + q4;
    o[0] = b0 + q1 + q3;
}

// Dequantised block (natural order) -> 8x8 samples at out (stb_image.h:1946-2003: columns first, two extra bits kept,
// an all-zero AC column short-cuts to dc << 2; level shift and rounding folded into the row pass).
inline void idct_block(uint8_t *out, int stride, const int16_t d[64]) {
    int v[64];
    for (int c = 0; c < 8; c++) {
        if (!(d[c + 8] | d[c + 16] | d[c + 24] | d[c + 32] | d[c + 40] | d[c + 48] | d[c + 56])) {
            const int dc = d[c] * 4;
            for (int r = 0; r < 8; r++) v[r * 8 + c] = dc;
            continue;
        }
        int s[8], e[4], o[4];
        for (int r = 0; r < 8; r++) s[r] = d[r * 8 + c];
        idct8(s, e, o);
        for (int k = 0; k < 4; k++) {
            v[k * 8 + c] = (e[k] + 512 + o[3 - k]) >> 10;
            v[(7 - k) * 8 + c] = (e[k] + 512 - o[3 - k]) >> 10;
        }
    }
    for (int r = 0; r < 8; r++) {
        int e[4], o[4];
        idct8(v + r * 8, e, o);
        uint8_t *row = out + (size_t)r * stride;
        for (int k = 0; k < 4; k++) {
            const int bias = 65536 + (128 << 17);
            row[k] = clamp_u8((e[k] + bias + o[3 - k]) >> 17);
            row[7 - k] = clamp_u8((e[k] + bias - o[3 - k]) >> 17);
        }
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;   // samples
    int w2 = 0, h2 = 0; // samples, padded to whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> data;
    std::vector<int16_t> coef; // (w2/8) x (h2/8) blocks of 64 coefficients, natural order (progressive: not yet dequantised)
    std::vector<uint8_t> done; // sequential files: blocks the scans have decoded (a truncated file leaves the rest black)
    int blocks_w = 0;
};

// The entropy-coded data is one serial bit stream; the inverse DCTs and the colour conversion that follow are not:
// they run on up to four threads (a 1024 x 512 texture: 12 ms -> 6 ms on one 2 GHz core's clock).
template <class F> inline void parallel_for(int n, int min_parallel, F f) {
    const unsigned hw = std::thread::hardware_concurrency();
    int threads = (int)(hw ? hw : 1u);
    if (threads > 4) threads = 4;
    if (threads > n) threads = n;
    if (threads <= 1 || n < min_parallel) {
        for (int i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<int> next{0};
    auto work = [&] {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            f(i);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}

// Chroma upsampling of one output row (stb_image.h:2839-2910, 3029): near / far are the two source rows that bracket it.
inline const uint8_t *upsample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, int w, int hs, int vs) {
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
    } else if (hs == 2 && vs == 1) {
        if (w == 1) {
            out[0] = out[1] = near[0];
            return out;
        }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        for (int i = 1; i < w - 1; i++) {
            const int n = 3 * near[i] + 2;
            out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[2 * (w - 1)] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = near[w - 1];
    } else if (hs == 2 && vs == 2) {
        int t1 = 3 * near[0] + far[0];
        if (w == 1) {
            out[0] = out[1] = (uint8_t)((t1 + 2) >> 2);
            return out;
        }
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
    } else { // any other ratio: nearest sample
        for (int i = 0; i < w; i++)
            for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    }
    return out;
}

// stb_image.h:3071-3095: 12-bit coefficients kept in 20-bit fixed point; the Cb term of green is truncated to its upper
// 16 bits before the sum.
inline void ycc_to_rgb_row(uint8_t *out, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, int count) {
    auto f2f = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
    static const int k_r = f2f(1.40200f), k_g_cr = -f2f(0.71414f), k_g_cb = -f2f(0.34414f), k_b = f2f(1.77200f);
    for (int i = 0; i < count; i++) {
        const int yf = (y[i] << 20) + (1 << 19);
        const int r_ = cr[i] - 128, b_ = cb[i] - 128;
        int r = yf + r_ * k_r;
        int g = yf + r_ * k_g_cr + (int)((unsigned)(b_ * k_g_cb) & 0xffff0000u);
        int b = yf + b_ * k_b;
        out[3 * i] = clamp_u8(r >> 20);
        out[3 * i + 1] = clamp_u8(g >> 20);
        out[3 * i + 2] = clamp_u8(b >> 20);
    }
}

struct Decoder {
    const uint8_t *p, *end;
    std::string err;
    uint8_t quant[4][64];
    bool have_quant[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];
    int ncomp = 0, width = 0, height = 0, h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0, restart_interval = 0;
    bool have_frame = false, progressive = false;
    int ss = 0, se = 63, ah = 0, al = 0, eob_run = 0; // progressive scan parameters (T.81 G.1)
    Bits bits;

    Decoder(const uint8_t *d, size_t n) : p(d), end(d + n) {}
    bool fail(const char *m) {
        if (err.empty()) err = m;
        return false;
    }
    int u8() { return p < end ? *p++ : 0; }
    int u16() {
        int a = u8();
        return (a << 8) | u8();
    }

    bool read_dqt(int len) { // T.81 B.2.4.1
        while (len > 0) {
            const int q = u8(), pq = q >> 4, t = q & 15;
            if (pq != 0) return fail("16-bit quantisation tables are not supported");
            if (t > 3) return fail("bad DQT table id");
            for (int i = 0; i < 64; i++) quant[t][k_zigzag[i]] = (uint8_t)u8();
            have_quant[t] = true;
            len -= 65;
        }
        return len == 0 || fail("bad DQT length");
    }
    bool read_dht(int len) { // T.81 B.2.4.2
        while (len > 0) {
            const int q = u8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) return fail("bad DHT header");
            uint8_t counts[16], symbols[256];
            int n = 0;
            for (int i = 0; i < 16; i++) n += (counts[i] = (uint8_t)u8());
            if (n > 256) return fail("bad DHT counts");
            for (int i = 0; i < n; i++) symbols[i] = (uint8_t)u8();
            if (!(tc == 0 ? dc[th] : ac[th]).build(counts, symbols, n)) return fail("bad code lengths");
            len -= 17 + n;
        }
        return len == 0 || fail("bad DHT length");
    }
    bool read_sof(int len) { // T.81 B.2.2
        if (have_frame) return fail("more than one frame");
        if (u8() != 8) return fail("only 8-bit samples are supported");
        height = u16();
        width = u16();
        ncomp = u8();
        if (height == 0) return fail("no header height");
        if (width == 0) return fail("zero width");
        if (ncomp != 1 && ncomp != 3) return fail("bad component count");
        if (len != 8 + 3 * ncomp) return fail("bad SOF length");
        h_max = v_max = 1;
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.id = u8();
            const int q = u8();
            c.h = q >> 4;
            c.v = q & 15;
            c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return fail("bad sampling factor");
            if (c.tq > 3) return fail("bad quantisation table id");
            if (c.h > h_max) h_max = c.h;
            if (c.v > v_max) v_max = c.v;
        }
        // the upsampler handles whole ratios only (1, 2, 3 = h_max / c.h, 4): a 3:2 frame would be read past its rows
        for (int i = 0; i < ncomp; i++)
            if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) return fail("non-integer chroma sampling ratio");
        if ((int64_t)width * height > (int64_t)1 << 28) return fail("image too large");
        mcus_x = (width + 8 * h_max - 1) / (8 * h_max);
        mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
        for (int i = 0; i < ncomp; i++) {
            Component &c = comp[i];
            c.x = (width * c.h + h_max - 1) / h_max;
            c.y = (height * c.v + v_max - 1) / v_max;
            c.w2 = mcus_x * c.h * 8;
            c.h2 = mcus_y * c.v * 8;
            c.data.assign((size_t)c.w2 * c.h2, 0);
            c.blocks_w = c.w2 / 8;
            c.coef.assign((size_t)c.w2 * c.h2, 0);
            if (!progressive) c.done.assign((size_t)c.blocks_w * (c.h2 / 8), 0);
        }
        have_frame = true;
        return true;
    }

    // T.81 F.2.2.1 / F.2.2.2 with the reference decoder's 16-bit products (stb_image.h:1711, 1728, 1741)
    bool decode_block(Component &c, int16_t d[64]) {
        const Huff &hd = dc[c.td], &ha = ac[c.ta];
        const uint8_t *q = quant[c.tq];
        std::memset(d, 0, 64 * sizeof(int16_t));
        const int t = bits.decode(hd);
        if (t < 0 || t > 15) return fail("bad huffman code");
        c.dc_pred += bits.receive_extend(t);
        d[0] = (int16_t)(c.dc_pred * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = bits.decode(ha);
            if (rs < 0) return fail("bad huffman code");
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break; // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) return fail("bad huffman code"); // run past the block
                const int z = k_zigzag[k++];
                d[z] = (int16_t)(bits.receive_extend(s) * q[z]);
            }
        }
        return true;
    }

    // Progressive scans (T.81 G.1.2), coefficient values as the reference's decoder keeps them (stb_image.h:1748-1890):
    // DC: first pass value << al, refinement adds one bit.
    bool prog_dc(Component &c, int16_t *d) {
        if (se != 0) return fail("can't merge dc and ac");
        if (ah == 0) {
            std::memset(d, 0, 64 * sizeof(int16_t));
            const int t = bits.decode(dc[c.td]);
            if (t < 0 || t > 15) return fail("bad huffman code");
            c.dc_pred += bits.receive_extend(t);
            d[0] = (int16_t)(c.dc_pred * (1 << al));
        } else if (bits.bit()) {
            d[0] = (int16_t)(d[0] + (int16_t)(1 << al));
        }
        return true;
    }
    // AC: spectral band [ss, se]; first pass with end-of-band runs, refinement with correction bits for the
    // coefficients that are already non-zero.
    bool prog_ac(Component &c, int16_t *d) {
        if (ss == 0) return fail("can't merge dc and ac");
        const Huff &h = ac[c.ta];
        if (ah == 0) {
            if (eob_run) {
                --eob_run;
                return true;
            }
            int k = ss;
            do {
                const int rs = bits.decode(h);
                if (rs < 0) return fail("bad huffman code");
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r);
                        if (r) eob_run += bits.raw(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) return fail("bad huffman code");
                    d[k_zigzag[k++]] = (int16_t)(bits.receive_extend(s) * (1 << al));
                }
            } while (k <= se);
            return true;
        }
        const int16_t bit = (int16_t)(1 << al);
        auto refine = [&](int16_t &v) { // one correction bit for a non-zero coefficient
            if (bits.bit() && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
        };
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; k++) {
                int16_t &v = d[k_zigzag[k]];
                if (v != 0) refine(v);
            }
            return true;
        }
        int k = ss;
        do {
            const int rs = bits.decode(h);
            if (rs < 0) return fail("bad huffman code");
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += bits.raw(r);
                    r = 64; // finish the band: only correction bits from here
                }
                // r == 15: sixteen zero coefficients are skipped below (a run of 15, then the "new" value 0)
            } else {
                if (s != 1) return fail("bad huffman code");
                s = bits.bit() ? bit : -bit;
            }
            while (k <= se) {
                int16_t &v = d[k_zigzag[k++]];
                if (v != 0) {
                    refine(v);
                } else {
                    if (r == 0) {
                        v = (int16_t)s;
                        break;
                    }
                    --r;
                }
            }
        } while (k <= se);
        return true;
    }
    // after the last scan: dequantise (16-bit products, as the reference's decoder) and transform every block
    void prog_finish() {
        for (int n = 0; n < ncomp; n++) {
            Component &c = comp[n];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            parallel_for(bh, 8, [&](int j) {
                for (int i = 0; i < bw; i++) {
                    int16_t *d = c.coef.data() + 64 * ((size_t)i + (size_t)j * c.blocks_w);
                    for (int k = 0; k < 64; k++) d[k] = (int16_t)(d[k] * quant[c.tq][k]);
                    idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, d);
                }
            });
        }
    }
    // sequential files: transform the blocks the scans decoded (their coefficients are already dequantised)
    void seq_finish() {
        for (int n = 0; n < ncomp; n++) {
            Component &c = comp[n];
            const int bh = c.h2 / 8;
            parallel_for(bh, 8, [&](int j) {
                for (int i = 0; i < c.blocks_w; i++)
                    if (c.done[(size_t)i + (size_t)j * c.blocks_w])
                        idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, c.coef.data() + 64 * ((size_t)i + (size_t)j * c.blocks_w));
            });
        }
    }

    void restart() {
        bits.reset();
        eob_run = 0;
        for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0;
    }

    bool read_scan(int len) { // T.81 B.2.3 + the entropy-coded segments that follow
        if (!have_frame) return fail("scan before frame");
        const int ns = u8();
        if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) return fail("bad SOS header");
        int order[3];
        for (int i = 0; i < ns; i++) {
            const int id = u8(), q = u8();
            int which = -1;
            for (int k = 0; k < ncomp; k++)
                if (comp[k].id == id) which = k;
            if (which < 0) return fail("bad SOS component");
            comp[which].td = q >> 4;
            comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad SOS table id");
            if (!have_quant[comp[which].tq]) return fail("missing quantisation table");
            order[i] = which;
        }
        ss = u8();
        se = u8();
        const int a = u8();
        ah = a >> 4;
        al = a & 15;
        if (progressive) {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return fail("bad SOS parameters");
            if (ss != 0 && ns != 1) return fail("interleaved AC scan");
        } else if (ss != 0 || se != 63 || a != 0) {
            return fail("bad SOS parameters");
        }
        for (int i = 0; i < ns; i++) {
            const Component &c = comp[order[i]];
            const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
            if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) return fail("missing huffman table");
        }
        bits.p = p;
        bits.end = end;
        restart();
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        auto seq_block = [&](Component &c, int bx, int by) { // one block of a sequential scan -> the coefficient store
            const size_t b = (size_t)bx + (size_t)by * c.blocks_w;
            if (!decode_block(c, c.coef.data() + 64 * b)) return false;
            c.done[b] = 1;
            return true;
        };
        // after each MCU: at a restart interval boundary the next thing in the stream must be RSTn, else stop decoding
        // this scan and keep what there is (as the reference's decoder does)
        auto mcu_done = [&]() -> int {
            if (--todo > 0) return 1;
            if (bits.n < 24) bits.fill();
            if (bits.marker < 0xd0 || bits.marker > 0xd7) return 0;
            restart();
            todo = restart_interval ? restart_interval : 0x7fffffff;
            return 1;
        };
        bool stop = false;
        if (ns == 1) { // non-interleaved: the component's own blocks in raster order (T.81 A.2.2)
            Component &c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh && !stop; j++)
                for (int i = 0; i < bw && !stop; i++) {
                    if (progressive) {
                        int16_t *blk = c.coef.data() + 64 * ((size_t)i + (size_t)j * c.blocks_w);
                        if (!(ss == 0 ? prog_dc(c, blk) : prog_ac(c, blk))) return false;
                    } else if (!seq_block(c, i, j)) {
                        return false;
                    }
                    if (!mcu_done()) stop = true;
                }
        } else { // interleaved MCUs (T.81 A.2.3)
            for (int j = 0; j < mcus_y && !stop; j++)
                for (int i = 0; i < mcus_x && !stop; i++) {
                    for (int k = 0; k < ns; k++) {
                        Component &c = comp[order[k]];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++) {
                                if (progressive) { // only DC scans are interleaved
                                    int16_t *blk = c.coef.data() + 64 * ((size_t)(i * c.h + x) + (size_t)(j * c.v + y) * c.blocks_w);
                                    if (!prog_dc(c, blk)) return false;
                                    continue;
                                }
                                if (!seq_block(c, i * c.h + x, j * c.v + y)) return false;
                            }
                    }
                    if (!mcu_done()) stop = true;
                }
        }
        p = bits.p;
        return true;
    }

    // next marker code, 0 at end of data
    int next_marker() {
        if (bits.marker) {
            const int m = bits.marker;
            bits.marker = 0;
            return m;
        }
        while (p < end) {
            if (u8() != 0xff) continue;
            int m = 0xff;
            while (m == 0xff && p < end) m = u8();
            if (m != 0 && m != 0xff) return m;
        }
        return 0;
    }

    bool run(Image &img) {
        if (end - p < 4 || p[0] != 0xff || p[1] != 0xd8) return fail("not a JPEG file");
        p += 2;
        bool seen_scan = false;
        for (;;) {
            const int m = next_marker();
            if (m == 0) {
                if (seen_scan) break; // truncated after the scans: keep what was decoded
                return fail("no image data");
            }
            if (m == 0xd9) break;                 // EOI
            if (m >= 0xd0 && m <= 0xd7) continue; // stray RSTn
            if (m == 0x01) continue;              // TEM
            const int len = u16();
            if (len < 2 || len - 2 > end - p) return fail("bad marker length");
            const uint8_t *seg_end = p + (len - 2);
            switch (m) {
            case 0xc0:
            case 0xc1:
            case 0xc2:
                progressive = m == 0xc2;
                if (!read_sof(len)) return false;
                break;
            case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                return fail("unsupported JPEG coding process");
            case 0xc4:
                if (!read_dht(len - 2)) return false;
                break;
            case 0xdb:
                if (!read_dqt(len - 2)) return false;
                break;
            case 0xdd:
                if (len != 4) return fail("bad DRI length");
                restart_interval = u16();
                break;
            case 0xda:
                if (!read_scan(len)) return false;
                seen_scan = true;
                continue; // p already stands behind the entropy-coded data
            default: break; // APPn, COM, ...: skipped
            }
            p = seg_end;
        }
        if (!have_frame || !seen_scan) return fail("no image data");
        if (progressive) prog_finish();
        else seq_finish();
        img.width = width;
        img.height = height;
        img.file_components = ncomp;
        img.rgb.assign((size_t)width * height * 3, 0);
        // rows: upsample each component to full resolution, then convert (stb_image.h:3319-3379).  Which two source rows
        // bracket an output row is a running state of the reference's loop: worked out for all rows first, so that the
        // rows themselves can be converted in parallel.
        struct Src {
            const uint8_t *near, *far;
        };
        std::vector<Src> src[3];
        int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, w_lores[3] = {0, 0, 0};
        for (int k = 0; k < ncomp; k++) {
            hs[k] = h_max / comp[k].h;
            vs[k] = v_max / comp[k].v;
            w_lores[k] = (width + hs[k] - 1) / hs[k];
            src[k].resize((size_t)height);
            int ystep = vs[k] >> 1, ypos = 0;
            const uint8_t *line0 = comp[k].data.data(), *line1 = line0;
            for (int j = 0; j < height; j++) {
                const bool bottom = ystep >= (vs[k] >> 1);
                src[k][(size_t)j] = Src{bottom ? line1 : line0, bottom ? line0 : line1};
                if (++ystep >= vs[k]) {
                    ystep = 0;
                    line0 = line1;
                    if (++ypos < comp[k].y) line1 += comp[k].w2;
                }
            }
        }
        const int band = 16; // rows per task
        parallel_for((height + band - 1) / band, 8, [&](int t) {
            std::vector<uint8_t> line[3];
            for (int k = 0; k < ncomp; k++) line[k].assign((size_t)width + 8, 0);
            for (int j = t * band; j < height && j < (t + 1) * band; j++) {
                const uint8_t *row[3] = {nullptr, nullptr, nullptr};
                for (int k = 0; k < ncomp; k++)
                    row[k] = upsample_row(line[k].data(), src[k][(size_t)j].near, src[k][(size_t)j].far, w_lores[k], hs[k], vs[k]);
                uint8_t *out = img.rgb.data() + (size_t)j * width * 3;
                if (ncomp == 3) ycc_to_rgb_row(out, row[0], row[1], row[2], width);
                else
                    for (int i = 0; i < width; i++) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = row[0][i];
            }
        });
        return true;
    }
};

} // namespace detail

// Decode a JPEG held in memory to 8-bit RGB.  Returns false and a message in *error for anything it cannot read.
inline bool decode(const uint8_t *data, size_t size, Image &out, std::string *error = nullptr) {
    detail::Decoder d(data, size);
    if (d.run(out)) return true;
    if (error) *error = d.err.empty() ? "corrupt JPEG" : d.err;
    out = Image();
    return false;
}

inline bool is_jpeg(const uint8_t *data, size_t size) { return size >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff; }

} // namespace jpeg
} // namespace rtamd
#endif
