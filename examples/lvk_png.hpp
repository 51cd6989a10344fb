// lvk_png.hpp — a small PNG reader for the dataset drivers: file -> 8-bit grey image, the job cv::imread(path, 0) does in
// the reference's drivers (/root/reference/app/larvioMain.cpp:95).  OpenCV is not installed here; zlib is, so only the container,
// the five scanline filters, Adam7 and the sample conversions are written out.  Host-only (link with -lz).
//
// Conversions follow what OpenCV's PNG reader asks of libpng for IMREAD_GRAYSCALE: 16-bit samples keep their high byte, 1/2/4-bit
// grey is scaled to 0..255, palette entries are expanded, alpha is dropped, colour is reduced with 0.299/0.587/0.114 in 15-bit
// fixed point.  The EuRoC (8-bit grey) and TUM-VI (16-bit grey) images are the cases the hot path cares about and are exact.
#ifndef LVK_PNG_HPP
#define LVK_PNG_HPP
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace lvk {

struct GreyImage { int width, height; std::vector<uint8_t> data; GreyImage() : width(0), height(0) {} };

namespace png_detail {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the scanline filters of one (sub-)image in place; `src` holds rows of 1 filter byte + `stride` data bytes
inline bool unfilter(uint8_t* src, int rows, size_t stride, int bpp)
{
    const uint8_t* prev = nullptr;
    for (int y = 0; y < rows; ++y) {
        uint8_t* line = src + (size_t)y * (stride + 1);
        const int type = line[0]; uint8_t* cur = line + 1;
        switch (type) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: if (prev) for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            break;
        default: return false;
        }
        prev = cur;
    }
    return true;
}

struct Header { int w, h, depth, colour, interlace, channels; };

// one decoded pixel (x of a row of packed samples) -> grey
inline uint8_t to_grey(const Header& H, const uint8_t* row, int x, const uint8_t* plte, int n_plte)
{
    if (H.colour == 3) {                                            // palette index -> RGB -> grey
        int idx;
        if (H.depth == 8) idx = row[x];
        else { const int per = 8 / H.depth, sh = (per - 1 - x % per) * H.depth; idx = (row[x / per] >> sh) & ((1 << H.depth) - 1); }
        if (idx >= n_plte) return 0;
        const uint8_t* c = plte + 3 * idx;
        return (uint8_t)((c[0] * 9798 + c[1] * 19235 + c[2] * 3735 + 16384) >> 15);
    }
    if (H.colour == 0 || H.colour == 4) {                           // grey (+alpha)
        if (H.depth == 16) return row[(size_t)x * H.channels * 2];                  // high byte
        if (H.depth == 8) return row[(size_t)x * H.channels];
        const int per = 8 / H.depth, sh = (per - 1 - x % per) * H.depth, v = (row[x / per] >> sh) & ((1 << H.depth) - 1);
        return (uint8_t)(v * 255 / ((1 << H.depth) - 1));
    }
    const int bps = H.depth / 8; const uint8_t* p = row + (size_t)x * H.channels * bps;       // RGB(A), high bytes
    return (uint8_t)((p[0] * 9798 + p[bps] * 19235 + p[2 * bps] * 3735 + 16384) >> 15);
}

}  // namespace png_detail

// decode `bytes` (a whole .png file).  false + *err on anything malformed or unsupported.
inline bool decode_png_grey(const uint8_t* bytes, size_t size, GreyImage* out, std::string* err)
{
    using namespace png_detail;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (size < 8 || std::memcmp(bytes, sig, 8) != 0) { if (err) *err = "not a PNG file"; return false; }
    Header H = {0, 0, 0, 0, 0, 0}; bool have_ihdr = false, have_end = false;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8;
    while (pos + 12 <= size && !have_end) {
        const uint32_t len = be32(bytes + pos); const uint8_t* type = bytes + pos + 4; const uint8_t* data = bytes + pos + 8;
        if ((size_t)len > size - pos - 12) { if (err) *err = "truncated chunk"; return false; }
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len)) { if (err) *err = "chunk CRC mismatch"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { if (err) *err = "bad IHDR"; return false; }
            H.w = (int)be32(data); H.h = (int)be32(data + 4); H.depth = data[8]; H.colour = data[9]; H.interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || H.interlace > 1 || H.w <= 0 || H.h <= 0 || H.w > (1 << 16) || H.h > (1 << 16)) { if (err) *err = "unsupported IHDR"; return false; }
            switch (H.colour) { case 0: H.channels = 1; break; case 2: H.channels = 3; break; case 3: H.channels = 1; break;
                                case 4: H.channels = 2; break; case 6: H.channels = 4; break; default: if (err) *err = "bad colour type"; return false; }
            const bool ok_depth = (H.colour == 0) ? (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8 || H.depth == 16)
                                : (H.colour == 3) ? (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8) : (H.depth == 8 || H.depth == 16);
            if (!ok_depth) { if (err) *err = "bad bit depth"; return false; }
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) have_end = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || idat.empty()) { if (err) *err = "missing IHDR or IDAT"; return false; }
    if (H.colour == 3 && plte.empty()) { if (err) *err = "palette image without PLTE"; return false; }

    const int bits_pp = H.depth * H.channels, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // sub-images: one for a progressive file, seven Adam7 passes otherwise  {x0, y0, dx, dy}
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int (*pass)[4] = H.interlace ? adam7 : whole; const int n_pass = H.interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; ++k) {
        const int pw = (H.w - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (H.h - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (pw > 0 && ph > 0) raw_size += (size_t)ph * (1 + ((size_t)pw * bits_pp + 7) / 8);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf got = (uLongf)raw_size;
    const int zr = uncompress(raw.data(), &got, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || got != raw_size) { if (err) *err = "zlib stream does not match the image size"; return false; }

    out->width = H.w; out->height = H.h; out->data.assign((size_t)H.w * H.h, 0);
    size_t off = 0;
    for (int k = 0; k < n_pass; ++k) {
        const int pw = (H.w - pass[k][0] + pass[k][2] - 1) / pass[k][2], ph = (H.h - pass[k][1] + pass[k][3] - 1) / pass[k][3];
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        if (!unfilter(raw.data() + off, ph, stride, bpp)) { if (err) *err = "bad filter type"; return false; }
        for (int y = 0; y < ph; ++y) {
            const uint8_t* row = raw.data() + off + (size_t)y * (stride + 1) + 1;
            uint8_t* dst = out->data.data() + (size_t)(pass[k][1] + y * pass[k][3]) * H.w + pass[k][0];
            for (int x = 0; x < pw; ++x) dst[(size_t)x * pass[k][2]] = to_grey(H, row, x, plte.data(), (int)(plte.size() / 3));
        }
        off += (size_t)ph * (stride + 1);
    }
    return true;
}

inline bool read_png_grey(const std::string& path, GreyImage* out, std::string* err)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<uint8_t> bytes; uint8_t buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return decode_png_grey(bytes.data(), bytes.size(), out, err);
}

}  // namespace lvk
#endif
