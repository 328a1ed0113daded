// image_io.cpp -- SURVEY 8f items 2 and 3 on the host side: PPM / PNG writers for the presented frame
// (what the reference's Bgra8UnormSrgb target holds, src/pipeline.rs:56-60 -- so output can be compared
// with screenshots of the real app) and a scripted fly-cam input file (src/fly_cam.rs:98-176 driven from
// text instead of winit events).  Plain C++, no GPU calls, no third-party code: the PNG is written with
// stored (uncompressed) deflate blocks.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_host.h"

namespace {

uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data) {
    put_be32(png, (uint32_t)data.size());
    size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_of(png.data() + at, png.size() - at, 0));
}
bool write_file(const char *path, const uint8_t *p, size_t n) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(p, 1, n, f) == n;
    return std::fclose(f) == 0 && ok;
}

}  // namespace

extern "C" {

// Binary PPM (P6), RGB from the BGRA8 frame, alpha dropped.  0 ok, -1 bad argument, -2 I/O error.
int rth_write_ppm(const char *path, const uint8_t *bgra, uint32_t width, uint32_t height) {
    if (!path || !bgra || !width || !height) return -1;
    char head[64];
    int hl = std::snprintf(head, sizeof head, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> out((size_t)hl + (size_t)width * height * 3);
    std::memcpy(out.data(), head, (size_t)hl);
    uint8_t *q = out.data() + hl;
    for (size_t i = 0, n = (size_t)width * height; i < n; ++i) { q[3 * i] = bgra[4 * i + 2]; q[3 * i + 1] = bgra[4 * i + 1]; q[3 * i + 2] = bgra[4 * i]; }
    return write_file(path, out.data(), out.size()) ? 0 : -2;
}

// PNG, 8-bit RGBA (colour type 6), sRGB chunk (the bytes ARE sRGB-encoded), stored deflate blocks.
int rth_write_png(const char *path, const uint8_t *bgra, uint32_t width, uint32_t height) {
    if (!path || !bgra || !width || !height) return -1;
    const size_t row = (size_t)width * 4 + 1;
    std::vector<uint8_t> raw(row * height);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *r = raw.data() + row * y;
        r[0] = 0;                                          // filter: none
        const uint8_t *s = bgra + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; ++x) { r[1 + 4 * x] = s[4 * x + 2]; r[2 + 4 * x] = s[4 * x + 1]; r[3 + 4 * x] = s[4 * x]; r[4 + 4 * x] = s[4 * x + 3]; }
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);                  // zlib header, no compression
    uint32_t a = 1, b = 0;                                 // Adler-32 of the raw stream
    for (size_t at = 0; at < raw.size();) {
        size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n == raw.size() ? 1 : 0);        // BFINAL, BTYPE = 00 (stored)
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        for (size_t i = at; i < at + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        at += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "sRGB", std::vector<uint8_t>{0});       // rendering intent: perceptual
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", std::vector<uint8_t>());
    return write_file(path, png.data(), png.size()) ? 0 : -2;
}

// Scripted fly-cam input.  One step per line:   dt  mouse_dx  mouse_dy  [keys]
// keys: any of W A S D (KeyBindings defaults, src/fly_cam.rs:47-50), ^ = Space (ascend, :51), v = ShiftLeft
// (descend, :52); '-' or nothing = no key.  '#' starts a comment; blank lines are skipped.  Every line is one
// rth_fly_cam_step, in order.  Returns the number of steps run, or -(line number) of the first malformed line
// (the camera then keeps the steps before it).
long rth_fly_cam_run_script(rth_app *app, const char *text) {
    if (!app || !text) return 0;
    long steps = 0, line_no = 0;
    const char *p = text;
    while (*p) {
        const char *e = std::strchr(p, '\n');
        std::string line(p, e ? (size_t)(e - p) : std::strlen(p));
        p = e ? e + 1 : p + line.size();
        ++line_no;
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        if (line.find_first_not_of(" \t\r") == std::string::npos) continue;
        float dt = 0.0f, mx = 0.0f, my = 0.0f;
        char keys[32] = "";
        int n = std::sscanf(line.c_str(), "%f %f %f %31s", &dt, &mx, &my, keys);
        if (n < 3) return -line_no;
        uint32_t mask = 0;
        for (const char *k = keys; *k; ++k) {
            switch (*k) {
                case 'W': case 'w': mask |= RTH_KEY_FORWARD; break;
                case 'S': case 's': mask |= RTH_KEY_BACKWARD; break;
                case 'A': case 'a': mask |= RTH_KEY_LEFT; break;
                case 'D': case 'd': mask |= RTH_KEY_RIGHT; break;
                case '^': mask |= RTH_KEY_ASCEND; break;
                case 'v': case 'V': mask |= RTH_KEY_DESCEND; break;
                case '-': break;
                default: return -line_no;
            }
        }
        rth_fly_cam_step(app, mx, my, mask, dt);
        ++steps;
    }
    return steps;
}

}  // extern "C"
