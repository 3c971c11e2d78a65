// JPEG reader for the textures and cubemap atlases the reference's scenes name (the reference decodes them with
// the `image` crate, interpret.rs:687-696,1109-1119, which pulls in jpeg-decoder 0.1.22; neither is vendored).
// Huffman coding, 8-bit samples, one or three components, sampling factors 1 and 2, restart intervals.  Baseline
// (one interleaved scan) and progressive (spectral selection and successive approximation over any number of
// scans: the coefficients are gathered first and turned into samples when the file ends).
//
// The arithmetic follows the public-domain stb_image decoder that jpeg-decoder's was derived from: the integer
// inverse DCT with 12-bit constants, linear interpolation of subsampled chroma, BT.601 YCbCr in f32 with +0.5
// rounding.  Whether every pixel equals what the reference's decoder produces cannot be checked here (no Rust
// toolchain): decoded images are inputs of the render, and the tests compare against libjpeg within a few counts.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "scene.hpp"

namespace rth {
namespace {

struct JpegError { std::string msg; };
[[noreturn]] void jfail(const std::string& m) { throw JpegError{"jpeg: " + m}; }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t values[256];
    uint8_t size[257];
    uint16_t code[256];
    int maxcode[18];     // first code that is too large for length k, left-aligned to 16 bits
    int delta[17];
    uint8_t fast[512];   // 9-bit lookahead: index into values, 255 = longer code
    void build(const uint8_t counts[16]) {
        int k = 0;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < counts[i]; j++) size[k++] = (uint8_t)(i + 1);
        size[k] = 0;
        int c = 0;
        k = 0;
        for (int len = 1; len <= 16; len++) {
            delta[len] = k - c;
            if (size[k] == len) {
                while (size[k] == len) code[k++] = (uint16_t)c++;
                if (c - 1 >= (1 << len)) jfail("bad Huffman code lengths");
            }
            maxcode[len] = c << (16 - len);
            c <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        std::memset(fast, 255, sizeof(fast));
        for (int i = 0; i < k; i++) {
            const int s = size[i];
            if (s <= 9) {
                const int first = code[i] << (9 - s), n = 1 << (9 - s);
                for (int j = 0; j < n; j++) fast[first + j] = (uint8_t)i;
            }
        }
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int w2 = 0, h2 = 0;           // padded plane size in samples
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef;    // progressive: every block's 64 coefficients (natural order), w2/8 blocks per row
};

class Decoder {
   public:
    Decoder(const uint8_t* data, size_t n) : p_(data), end_(data + n) {}

    void decode(uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb) {
        if (get8() != 0xFF || get8() != 0xD8) jfail("not a JPEG file");
        bool have_frame = false, have_scan = false;
        while (true) {
            int m = next_marker();
            if (m == 0xD9) { if (!have_scan) jfail("no image data"); break; }
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) jfail("more than one frame");
                progressive_ = m == 0xC2;
                read_frame(); have_frame = true;
            }
            else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) jfail("unsupported JPEG coding process");
            else if (m == 0xC4) read_dht();
            else if (m == 0xDB) read_dqt();
            else if (m == 0xDD) { if (get16() != 4) jfail("bad DRI"); restart_ = get16(); }
            else if (m == 0xDA) {
                if (!have_frame) jfail("scan before frame header");
                if (progressive_) read_scan_progressive(); else read_scan();
                have_scan = true;
                if (!progressive_) break;
                // the bit reader stopped at (or before) the next marker
                bits_ = 0; nbits_ = 0; hit_marker_ = false;
                if (p_ >= end_) break;   // a file cut off after its last scan still shows what it holds
            } else skip_segment();
        }
        if (progressive_) finish_progressive();
        w = (uint32_t)width_; h = (uint32_t)height_;
        to_rgb(rgb);
    }

   private:
    const uint8_t* p_;
    const uint8_t* end_;
    int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0;
    bool progressive_ = false;
    int spec_start_ = 0, spec_end_ = 0, succ_high_ = 0, succ_low_ = 0, eob_run_ = 0;
    Component comp_[3];
    uint16_t quant_[4][64] = {};
    bool have_quant_[4] = {false, false, false, false};
    Huffman dc_[4], ac_[4];
    // entropy-coded bit reader
    uint32_t bits_ = 0;
    int nbits_ = 0;
    bool hit_marker_ = false;

    int get8() { if (p_ >= end_) jfail("unexpected end of file"); return *p_++; }
    int get16() { int a = get8(); return (a << 8) | get8(); }
    int next_marker() {
        int c = get8();
        while (c != 0xFF) c = get8();
        while (c == 0xFF) c = get8();
        return c;
    }
    void skip_segment() { int len = get16(); if (len < 2 || p_ + (len - 2) > end_) jfail("bad segment length"); p_ += len - 2; }

    void read_dqt() {
        int len = get16() - 2;
        while (len > 0) {
            const int q = get8(), prec = q >> 4, t = q & 15;
            if (t > 3 || prec > 1) jfail("bad quantisation table");
            for (int i = 0; i < 64; i++) quant_[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
            have_quant_[t] = true;
            len -= 65 + 64 * prec;
        }
        if (len != 0) jfail("bad DQT length");
    }
    void read_dht() {
        int len = get16() - 2;
        while (len > 0) {
            const int q = get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) jfail("bad Huffman table header");
            uint8_t counts[16];
            int total = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); total += counts[i]; }
            if (total > 256) jfail("bad Huffman table");
            Huffman& hf = tc ? ac_[th] : dc_[th];
            for (int i = 0; i < total; i++) hf.values[i] = (uint8_t)get8();
            hf.build(counts);
            len -= 17 + total;
        }
        if (len != 0) jfail("bad DHT length");
    }
    void read_frame() {
        const int len = get16();
        if (get8() != 8) jfail("only 8-bit samples are supported");
        height_ = get16(); width_ = get16(); ncomp_ = get8();
        if (height_ == 0 || width_ == 0) jfail("empty image");
        if (ncomp_ != 1 && ncomp_ != 3) jfail("only 1 or 3 components are supported");
        if (len != 8 + 3 * ncomp_) jfail("bad SOF length");
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            c.id = get8();
            const int q = get8();
            c.h = q >> 4; c.v = q & 15; c.tq = get8();
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) jfail("unsupported sampling factors");
            hmax_ = std::max(hmax_, c.h); vmax_ = std::max(vmax_, c.v);
        }
        if (ncomp_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }
        if ((uint64_t)width_ * (uint64_t)height_ > (1ull << 30)) jfail("image too large");
    }

    // ---- entropy-coded segment
    void fill() {
        while (nbits_ <= 24) {
            int b = 0;
            if (!hit_marker_) {
                if (p_ >= end_) hit_marker_ = true;
                else {
                    b = *p_++;
                    if (b == 0xFF) {
                        const int c = p_ < end_ ? *p_ : 0xD9;
                        if (c == 0) p_++;                     // stuffed zero: a data byte 0xFF
                        else { p_--; hit_marker_ = true; b = 0; }   // a marker: feed zeros from here on
                    }
                }
            }
            bits_ |= (uint32_t)b << (24 - nbits_);
            nbits_ += 8;
        }
    }
    int decode_symbol(const Huffman& h) {
        if (nbits_ < 16) fill();
        const int look = (int)(bits_ >> 23);
        const int k = h.fast[look];
        if (k < 255) {
            const int s = h.size[k];
            bits_ <<= s; nbits_ -= s;
            return h.values[k];
        }
        const int top = (int)(bits_ >> 16);
        int len = 10;
        while (top >= h.maxcode[len]) len++;
        if (len == 17) jfail("bad Huffman code");
        const int c = (int)(bits_ >> (32 - len)) + h.delta[len];
        if (c < 0 || c >= 256) jfail("bad Huffman code");
        bits_ <<= len; nbits_ -= len;
        return h.values[c];
    }
    int receive_extend(int n) {
        if (n == 0) return 0;
        if (nbits_ < n) fill();
        const int v = (int)(bits_ >> (32 - n));
        bits_ <<= n; nbits_ -= n;
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    void decode_block(Component& c, int coef[64]) {
        std::memset(coef, 0, 64 * sizeof(int));
        const Huffman& hd = dc_[c.td];
        const Huffman& ha = ac_[c.ta];
        const int t = decode_symbol(hd);
        if (t > 15) jfail("bad DC code");
        c.dc_pred += receive_extend(t);
        coef[0] = c.dc_pred * quant_[c.tq][0];
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(ha), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;   // end of block
                k += 16;
            } else {
                k += r;
                if (k > 63) jfail("bad AC run");
                const int z = kZigzag[k];
                coef[z] = receive_extend(s) * quant_[c.tq][z];
                k++;
            }
        }
    }

    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
    // integer inverse DCT (12-bit constants, 10 + 17 bits of scaling), output level-shifted and clamped
    static void idct(const int d[64], uint8_t* out, int stride) {
        auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
        static const int c0541 = f2f(0.5411961), cm184 = f2f(-1.847759065), c0765 = f2f(0.765366865), c1175 = f2f(1.175875602);
        static const int c0298 = f2f(0.298631336), c2053 = f2f(2.053119869), c3072 = f2f(3.072711026), c1501 = f2f(1.501321110);
        static const int cm089 = f2f(-0.899976223), cm256 = f2f(-2.562915447), cm196 = f2f(-1.961570560), cm039 = f2f(-0.390180644);
        int val[64];
        auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int& t0, int& t1, int& t2, int& t3, int& x0, int& x1, int& x2, int& x3) {
            int p2 = s2, p3 = s6;
            int p1 = (p2 + p3) * c0541;
            int tt2 = p1 + p3 * cm184, tt3 = p1 + p2 * c0765;
            p2 = s0; p3 = s4;
            int tt0 = (p2 + p3) * 4096, tt1 = (p2 - p3) * 4096;
            x0 = tt0 + tt3; x3 = tt0 - tt3; x1 = tt1 + tt2; x2 = tt1 - tt2;
            tt0 = s7; tt1 = s5; tt2 = s3; tt3 = s1;
            p3 = tt0 + tt2;
            int p4 = tt1 + tt3;
            p1 = tt0 + tt3; p2 = tt1 + tt2;
            const int p5 = (p3 + p4) * c1175;
            tt0 *= c0298; tt1 *= c2053; tt2 *= c3072; tt3 *= c1501;
            p1 = p5 + p1 * cm089; p2 = p5 + p2 * cm256; p3 *= cm196; p4 *= cm039;
            t3 = tt3 + p1 + p4; t2 = tt2 + p2 + p3; t1 = tt1 + p2 + p4; t0 = tt0 + p1 + p3;
        };
        for (int i = 0; i < 8; i++) {   // columns
            const int* c = d + i;
            int* v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dc = c[0] * 4;
                for (int k = 0; k < 8; k++) v[8 * k] = dc;
            } else {
                int t0, t1, t2, t3, x0, x1, x2, x3;
                pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], t0, t1, t2, t3, x0, x1, x2, x3);
                x0 += 512; x1 += 512; x2 += 512; x3 += 512;
                v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
                v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
                v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
                v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
            }
        }
        for (int i = 0; i < 8; i++) {   // rows
            const int* v = val + 8 * i;
            uint8_t* o = out + (size_t)i * (size_t)stride;
            int t0, t1, t2, t3, x0, x1, x2, x3;
            pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], t0, t1, t2, t3, x0, x1, x2, x3);
            const int bias = 65536 + (128 << 17);
            x0 += bias; x1 += bias; x2 += bias; x3 += bias;
            o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
            o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
            o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
            o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
        }
    }

    void read_scan() {
        const int len = get16();
        const int ns = get8();
        if (ns != ncomp_ || len != 6 + 2 * ns) jfail("only one interleaved scan over all components is supported");
        for (int i = 0; i < ns; i++) {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < ncomp_; k++) if (comp_[k].id == id) which = k;
            if (which != i) jfail("scan components out of order");
            comp_[i].td = q >> 4; comp_[i].ta = q & 15;
            if (comp_[i].td > 3 || comp_[i].ta > 3 || !dc_[comp_[i].td].present || !ac_[comp_[i].ta].present) jfail("scan refers to a missing Huffman table");
            if (!have_quant_[comp_[i].tq]) jfail("component refers to a missing quantisation table");
        }
        if (get8() != 0) jfail("bad spectral selection (not baseline)");
        get8();
        if (get8() != 0) jfail("bad successive approximation (not baseline)");

        const int mcu_w = 8 * hmax_, mcu_h = 8 * vmax_;
        const int mcus_x = (width_ + mcu_w - 1) / mcu_w, mcus_y = (height_ + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            c.w2 = mcus_x * c.h * 8; c.h2 = mcus_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * (size_t)c.h2, 0);
            c.dc_pred = 0;
        }
        int todo = restart_ ? restart_ : 0x7FFFFFFF;
        int coef[64];
        for (int my = 0; my < mcus_y; my++) {
            for (int mx = 0; mx < mcus_x; mx++) {
                for (int i = 0; i < ncomp_; i++) {
                    Component& c = comp_[i];
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++) {
                            decode_block(c, coef);
                            idct(coef, c.plane.data() + (size_t)((my * c.v + by) * 8) * (size_t)c.w2 + (size_t)((mx * c.h + bx) * 8), c.w2);
                        }
                }
                if (--todo <= 0) {
                    // a restart marker follows, unless this was the last unit
                    if (my == mcus_y - 1 && mx == mcus_x - 1) break;
                    if (nbits_ < 24) fill();
                    bits_ = 0; nbits_ = 0; hit_marker_ = false;
                    const int m = next_marker();
                    if (m < 0xD0 || m > 0xD7) jfail("missing restart marker");
                    for (int i = 0; i < ncomp_; i++) comp_[i].dc_pred = 0;
                    todo = restart_;
                }
            }
        }
    }

    // ---- progressive scans (ITU T.81 annex G)
    int get_bits(int n) {
        if (n == 0) return 0;
        if (nbits_ < n) fill();
        const int v = (int)(bits_ >> (32 - n));
        bits_ <<= n; nbits_ -= n;
        return v;
    }
    void block_dc(Component& c, int16_t* d) {
        if (succ_high_ == 0) {   // first pass: the value, scaled down by the bits still to come
            std::memset(d, 0, 64 * sizeof(int16_t));
            const int t = decode_symbol(dc_[c.td]);
            if (t > 15) jfail("bad DC code");
            c.dc_pred += receive_extend(t);
            d[0] = (int16_t)(c.dc_pred * (1 << succ_low_));
        } else if (get_bits(1)) d[0] = (int16_t)(d[0] + (1 << succ_low_));   // refinement: one more bit
    }
    void refine(int16_t& v, int bit) {   // one correction bit for a coefficient that is already non-zero
        if (get_bits(1) && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    }
    void block_ac(Component& c, int16_t* d) {
        const Huffman& ha = ac_[c.ta];
        if (succ_high_ == 0) {
            if (eob_run_) { --eob_run_; return; }
            int k = spec_start_;
            do {
                const int rs = decode_symbol(ha), r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                    if (r < 15) {   // end of band for this block and the next eob_run_ ones
                        eob_run_ = (1 << r) - 1;
                        if (r) eob_run_ += get_bits(r);
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) jfail("bad AC run");
                    d[kZigzag[k++]] = (int16_t)(receive_extend(sz) * (1 << succ_low_));
                }
            } while (k <= spec_end_);
            return;
        }
        const int bit = 1 << succ_low_;
        if (eob_run_) {
            --eob_run_;
            for (int k = spec_start_; k <= spec_end_; k++) { int16_t& v = d[kZigzag[k]]; if (v != 0) refine(v, bit); }
            return;
        }
        int k = spec_start_;
        do {
            const int rs = decode_symbol(ha);
            int r = rs >> 4, sz = rs & 15, val = 0;
            if (sz == 0) {
                if (r < 15) {
                    eob_run_ = (1 << r) - 1;
                    if (r) eob_run_ += get_bits(r);
                    r = 64;   // to the end of the band: only corrections remain
                }           // r == 15: sixteen zero coefficients are passed over
            } else {
                if (sz != 1) jfail("bad refinement code");
                val = get_bits(1) ? bit : -bit;
            }
            while (k <= spec_end_) {
                int16_t& v = d[kZigzag[k++]];
                if (v != 0) refine(v, bit);
                else {
                    if (r == 0) { v = (int16_t)val; break; }
                    --r;
                }
            }
        } while (k <= spec_end_);
    }

    void plane_sizes() {
        const int mcu_w = 8 * hmax_, mcu_h = 8 * vmax_;
        const int mcus_x = (width_ + mcu_w - 1) / mcu_w, mcus_y = (height_ + mcu_h - 1) / mcu_h;
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            c.w2 = mcus_x * c.h * 8; c.h2 = mcus_y * c.v * 8;
        }
    }
    void restart_if_due(int& todo, bool last) {
        if (--todo > 0 || last) return;
        if (nbits_ < 24) fill();
        bits_ = 0; nbits_ = 0; hit_marker_ = false;
        const int m = next_marker();
        if (m < 0xD0 || m > 0xD7) jfail("missing restart marker");
        for (int i = 0; i < ncomp_; i++) comp_[i].dc_pred = 0;
        eob_run_ = 0;
        todo = restart_;
    }
    void read_scan_progressive() {
        const int len = get16();
        const int ns = get8();
        if (ns < 1 || ns > ncomp_ || len != 6 + 2 * ns) jfail("bad scan header");
        int order[3];
        for (int i = 0; i < ns; i++) {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < ncomp_; k++) if (comp_[k].id == id) which = k;
            if (which < 0) jfail("scan names an unknown component");
            order[i] = which;
            comp_[which].td = q >> 4; comp_[which].ta = q & 15;
            if (comp_[which].td > 3 || comp_[which].ta > 3) jfail("bad Huffman table index");
        }
        spec_start_ = get8(); spec_end_ = get8();
        const int a = get8();
        succ_high_ = a >> 4; succ_low_ = a & 15;
        if (spec_start_ > 63 || spec_end_ > 63 || spec_start_ > spec_end_ || succ_high_ > 13 || succ_low_ > 13) jfail("bad progressive scan parameters");
        const bool dc_scan = spec_start_ == 0;
        if (dc_scan && spec_end_ != 0) jfail("a scan cannot hold both DC and AC coefficients");
        if (!dc_scan && ns != 1) jfail("an AC scan holds one component");
        for (int i = 0; i < ns; i++) {
            const Component& c = comp_[order[i]];
            if (dc_scan ? (succ_high_ == 0 && !dc_[c.td].present) : !ac_[c.ta].present) jfail("scan refers to a missing Huffman table");
        }
        if (comp_[0].coef.empty()) {
            plane_sizes();
            for (int i = 0; i < ncomp_; i++) comp_[i].coef.assign((size_t)comp_[i].w2 * (size_t)comp_[i].h2, 0);
        }
        for (int i = 0; i < ncomp_; i++) comp_[i].dc_pred = 0;
        eob_run_ = 0;
        bits_ = 0; nbits_ = 0; hit_marker_ = false;
        int todo = restart_ ? restart_ : 0x7FFFFFFF;
        if (ns == 1) {
            // one component: its blocks in raster order, only those that hold image samples
            Component& c = comp_[order[0]];
            const int cx = (width_ * c.h + hmax_ - 1) / hmax_, cy = (height_ * c.v + vmax_ - 1) / vmax_;
            const int bw = (cx + 7) >> 3, bh = (cy + 7) >> 3, row = c.w2 >> 3;
            for (int j = 0; j < bh; j++)
                for (int i = 0; i < bw; i++) {
                    int16_t* d = c.coef.data() + 64 * ((size_t)j * (size_t)row + (size_t)i);
                    if (dc_scan) block_dc(c, d); else block_ac(c, d);
                    restart_if_due(todo, j == bh - 1 && i == bw - 1);
                }
            return;
        }
        const int mcu_w = 8 * hmax_, mcu_h = 8 * vmax_;
        const int mcus_x = (width_ + mcu_w - 1) / mcu_w, mcus_y = (height_ + mcu_h - 1) / mcu_h;
        for (int my = 0; my < mcus_y; my++)
            for (int mx = 0; mx < mcus_x; mx++) {
                for (int i = 0; i < ns; i++) {
                    Component& c = comp_[order[i]];
                    const int row = c.w2 >> 3;
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++)
                            block_dc(c, c.coef.data() + 64 * ((size_t)(my * c.v + by) * (size_t)row + (size_t)(mx * c.h + bx)));
                }
                restart_if_due(todo, my == mcus_y - 1 && mx == mcus_x - 1);
            }
    }
    void finish_progressive() {
        int coef[64];
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            if (!have_quant_[c.tq]) jfail("component refers to a missing quantisation table");
            c.plane.assign((size_t)c.w2 * (size_t)c.h2, 0);
            const int row = c.w2 >> 3, rows = c.h2 >> 3;
            for (int j = 0; j < rows; j++)
                for (int b = 0; b < row; b++) {
                    const int16_t* d = c.coef.data() + 64 * ((size_t)j * (size_t)row + (size_t)b);
                    for (int k = 0; k < 64; k++) coef[k] = (int)d[k] * (int)quant_[c.tq][k];
                    idct(coef, c.plane.data() + (size_t)(j * 8) * (size_t)c.w2 + (size_t)(b * 8), c.w2);
                }
            c.coef.clear(); c.coef.shrink_to_fit();
        }
    }

    // one output row of component c at full resolution (linear interpolation of subsampled planes)
    void row_of(const Component& c, int y, std::vector<uint8_t>& out) const {
        const int hs = hmax_ / c.h, vs = vmax_ / c.v;
        const int cw = (width_ + hs - 1) / hs;   // samples of this component that carry image data, per row
        const int ch = (height_ + vs - 1) / vs;
        out.resize((size_t)width_ + 2);
        auto src = [&](int r) { r = r < 0 ? 0 : r >= ch ? ch - 1 : r; return c.plane.data() + (size_t)r * (size_t)c.w2; };
        if (hs == 1 && vs == 1) { std::memcpy(out.data(), src(y), (size_t)width_); return; }
        if (vs == 1) {   // h2v1: 3/4 near + 1/4 far
            const uint8_t* in = src(y);
            for (int x = 0; x < width_; x++) {
                const int i = x >> 1, far = (x & 1) ? std::min(i + 1, cw - 1) : std::max(i - 1, 0);
                out[(size_t)x] = (uint8_t)((3 * in[i] + in[far] + 2) >> 2);
            }
            return;
        }
        const int yy = y >> 1;
        const uint8_t* near = src(yy);
        const uint8_t* far = src((y & 1) ? yy + 1 : yy - 1);
        if (hs == 1) {   // h1v2
            for (int x = 0; x < width_; x++) out[(size_t)x] = (uint8_t)((3 * near[x] + far[x] + 2) >> 2);
            return;
        }
        for (int x = 0; x < width_; x++) {   // h2v2: 9/16, 3/16, 3/16, 1/16
            const int i = x >> 1, j = (x & 1) ? std::min(i + 1, cw - 1) : std::max(i - 1, 0);
            const int a = 3 * near[i] + far[i], b = 3 * near[j] + far[j];
            out[(size_t)x] = (uint8_t)((3 * a + b + 8) >> 4);
        }
    }

    void to_rgb(std::vector<uint8_t>& rgb) const {
        rgb.resize((size_t)width_ * (size_t)height_ * 3);
        std::vector<uint8_t> y, cb, cr;
        for (int row = 0; row < height_; row++) {
            uint8_t* o = rgb.data() + (size_t)row * (size_t)width_ * 3;
            row_of(comp_[0], row, y);
            if (ncomp_ == 1) {
                for (int x = 0; x < width_; x++) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = y[(size_t)x]; }
                continue;
            }
            row_of(comp_[1], row, cb);
            row_of(comp_[2], row, cr);
            for (int x = 0; x < width_; x++) {   // BT.601, f32, +0.5 and truncation
                const float yf = (float)y[(size_t)x], cbf = (float)cb[(size_t)x] - 128.f, crf = (float)cr[(size_t)x] - 128.f;
                const float r = yf + 1.40200f * crf, g = yf - 0.34414f * cbf - 0.71414f * crf, b = yf + 1.77200f * cbf;
                o[3 * x] = clamp8((int)(r + 0.5f)); o[3 * x + 1] = clamp8((int)(g + 0.5f)); o[3 * x + 2] = clamp8((int)(b + 0.5f));
            }
        }
    }
};

}  // namespace

bool read_jpeg(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "jpeg: cannot open " + path; return false; }
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        Decoder d(data.data(), data.size());
        d.decode(w, h, rgb);
        return true;
    } catch (const JpegError& e) {
        err = e.msg;
    } catch (const std::exception& e) {
        err = std::string("jpeg: ") + e.what();
    }
    return false;
}

bool read_image(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    unsigned char magic[2] = {0, 0};
    f.read(reinterpret_cast<char*>(magic), 2);
    f.close();
    if (magic[0] == 0xFF && magic[1] == 0xD8) return read_jpeg(path, w, h, rgb, err);
    return read_png(path, w, h, rgb, err);
}

}  // namespace rth
