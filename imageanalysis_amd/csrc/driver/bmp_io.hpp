// Uncompressed Windows BMP in and out, the one codec the reference's data set needs (its
// driver goes through OpenCV's imread/imwrite, test/test.cu:271,133; the cell-smear images are
// 24-bit .BMP files).  Reading gives planar R,G,B as convertMatToRGBImage lays it out
// (test/test.cu:43-60); 8-bit palettised and 32-bit files are accepted as well.  Writing a
// single-channel image produces an 8-bit file with a grey palette, as OpenCV does.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace bmp_io {

struct Planar {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;  // R plane, G plane, B plane
};

inline uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline void wr32(uint8_t *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = v >> 24; }
inline void wr16(uint8_t *p, uint16_t v) { p[0] = v & 255; p[1] = v >> 8; }

// Returns "" on success, else what was wrong.
inline std::string read(const std::string &path, Planar *out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "cannot open";
    std::vector<uint8_t> d;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long size = ftell(f);
        if (size > 0) d.resize((size_t)size);
        rewind(f);
    }
    if (!d.empty()) {
        if (fread(d.data(), 1, d.size(), f) != d.size()) d.clear();
    } else {  // not seekable: read to the end in pieces
        uint8_t buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    }
    fclose(f);
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return "not a BMP file";
    const uint32_t data_off = rd32(&d[10]);
    const uint32_t hdr = rd32(&d[14]);
    if (hdr < 40) return "unsupported BMP header";
    const int32_t w = (int32_t)rd32(&d[18]);
    int32_t h = (int32_t)rd32(&d[22]);
    const uint16_t bpp = rd16(&d[28]);
    const uint32_t comp = rd32(&d[30]);
    if (comp != 0 && !(comp == 3 && bpp == 32)) return "compressed BMP not supported";
    if (bpp != 8 && bpp != 24 && bpp != 32) return "unsupported bit depth";
    const bool top_down = h < 0;
    if (top_down) h = -h;
    if (w <= 0 || h <= 0 || (long long)w * h > (1LL << 30)) return "bad dimensions";
    const size_t row_bytes = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)data_off + row_bytes * (size_t)h > d.size()) return "truncated pixel data";
    const uint8_t *pal = &d[14 + hdr];
    uint32_t pal_n = rd32(&d[46]);
    if (bpp == 8) {
        if (pal_n == 0) pal_n = 256;
        if (14 + (size_t)hdr + 4 * (size_t)pal_n > d.size()) return "truncated palette";
    }
    out->width = w;
    out->height = h;
    const size_t N = (size_t)w * h;
    out->rgb.assign(3 * N, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = &d[data_off + row_bytes * (size_t)(top_down ? y : h - 1 - y)];
        uint8_t *R = &out->rgb[(size_t)y * w], *G = R + N, *B = G + N;
        for (int x = 0; x < w; x++) {
            if (bpp == 8) {
                const uint32_t i = row[x] < pal_n ? row[x] : 0;
                B[x] = pal[4 * i]; G[x] = pal[4 * i + 1]; R[x] = pal[4 * i + 2];
            } else {
                const uint8_t *px = row + (size_t)x * (bpp / 8);
                B[x] = px[0]; G[x] = px[1]; R[x] = px[2];
            }
        }
    }
    return "";
}

inline std::string write_gray(const std::string &path, const uint8_t *img, int w, int h) {
    const size_t row_bytes = (((size_t)w + 3) / 4) * 4;
    const uint32_t off = 14 + 40 + 1024;
    std::vector<uint8_t> d(off + row_bytes * (size_t)h, 0);
    d[0] = 'B'; d[1] = 'M';
    wr32(&d[2], (uint32_t)d.size());
    wr32(&d[10], off);
    wr32(&d[14], 40);
    wr32(&d[18], (uint32_t)w);
    wr32(&d[22], (uint32_t)h);
    wr16(&d[26], 1);
    wr16(&d[28], 8);
    wr32(&d[34], (uint32_t)(row_bytes * (size_t)h));
    wr32(&d[46], 256);
    for (int i = 0; i < 256; i++) { d[54 + 4 * i] = d[54 + 4 * i + 1] = d[54 + 4 * i + 2] = (uint8_t)i; }
    for (int y = 0; y < h; y++) memcpy(&d[off + row_bytes * (size_t)(h - 1 - y)], img + (size_t)y * w, (size_t)w);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "cannot create";
    const size_t n = fwrite(d.data(), 1, d.size(), f);
    fclose(f);
    return n == d.size() ? "" : "short write";
}

}  // namespace bmp_io
