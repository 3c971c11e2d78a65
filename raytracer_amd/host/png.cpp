// Minimal PNG encode/decode on zlib: the on-disk format either side of the
// render path.  Writer: what Scene::render_to saves (scene.rs:249-265, RGB8).
// Reader: what `image::open(..).into_rgb8()` yields for the texture files the
// SDL names (sdl/src/interpret.rs:690,1115) -- 8/16-bit gray, gray+alpha, RGB,
// RGBA and 1/2/4/8-bit palette images, non-interlaced; alpha is dropped.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scene.hpp"

namespace rth {

namespace {
void be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
uint32_t rd32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
bool chunk(FILE* f, const char* type, const uint8_t* data, uint32_t n) {
    uint8_t hdr[8]; be32(hdr, n); std::memcpy(hdr + 4, type, 4);
    uint32_t crc = (uint32_t)crc32(0, hdr + 4, 4);
    if (n) crc = (uint32_t)crc32(crc, data, n);
    uint8_t tail[4]; be32(tail, crc);
    return fwrite(hdr, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
}
const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
}  // namespace

bool write_png(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err) {
    std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* row = &raw[(size_t)y * (1 + (size_t)w * 3)];
        row[0] = 0;
        std::memcpy(row + 1, rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "png: deflate failed"; return false; }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "png: cannot open " + path; return false; }
    uint8_t ihdr[13]; be32(ihdr, w); be32(ihdr + 4, h); ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    bool ok = fwrite(SIG, 1, 8, f) == 8 && chunk(f, "IHDR", ihdr, 13) && chunk(f, "IDAT", comp.data(), (uint32_t)clen) && chunk(f, "IEND", nullptr, 0);
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "png: write failed: " + path;
    return ok;
}

bool read_png(const std::string& path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgb, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "png: cannot open " + path; return false; }
    std::vector<uint8_t> file;
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    if (file.size() < 8 || std::memcmp(file.data(), SIG, 8) != 0) { err = "png: bad signature: " + path; return false; }
    size_t p = 8;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (p + 12 <= file.size()) {
        uint32_t len = rd32(&file[p]);
        if (p + 12 + (size_t)len > file.size()) { err = "png: truncated chunk"; return false; }
        const uint8_t* type = &file[p + 4];
        const uint8_t* data = &file[p + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = rd32(data); h = rd32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) { err = "png: bad header"; return false; }
    if (interlace) { err = "png: interlaced images are not supported"; return false; }
    uint32_t channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || !(depth == 8 || depth == 16 || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) || (ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) {
        err = "png: unsupported colour type / bit depth"; return false;
    }
    size_t bpp_bits = (size_t)channels * depth;
    size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    size_t bpp = (bpp_bits + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rlen, idat.data(), (uLong)idat.size()) != Z_OK || rlen != raw.size()) { err = "png: inflate failed"; return false; }
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    rgb.assign((size_t)w * h * 3, 0);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = &raw[(size_t)y * (stride + 1)];
        uint8_t ft = row[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int x = row[1 + i], v;
            switch (ft) {
                case 0: v = x; break;
                case 1: v = x + a; break;
                case 2: v = x + b; break;
                case 3: v = x + ((a + b) >> 1); break;
                case 4: { int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); break; }
                default: err = "png: bad filter"; return false;
            }
            cur[i] = (uint8_t)v;
        }
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* o = &rgb[3 * ((size_t)y * w + x)];
            auto sample = [&](uint32_t ch) -> uint32_t {   // value of channel ch at its bit depth
                if (depth == 8) return cur[(size_t)x * channels + ch];
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2];   // high byte (image's 16->8 conversion keeps the top 8 bits)
                size_t bit = (size_t)x * depth;
                return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1);
            };
            if (ctype == 2 || ctype == 6) { o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2); }
            else if (ctype == 0 || ctype == 4) {
                uint32_t g = sample(0);
                if (depth < 8) g = g * 255 / ((1u << depth) - 1);
                o[0] = o[1] = o[2] = (uint8_t)g;
            } else {
                uint32_t idx = sample(0);
                if ((size_t)idx * 3 + 2 >= plte.size()) { err = "png: palette index out of range"; return false; }
                o[0] = plte[3 * idx]; o[1] = plte[3 * idx + 1]; o[2] = plte[3 * idx + 2];
            }
        }
        prev.swap(cur);
    }
    return true;
}

}  // namespace rth
