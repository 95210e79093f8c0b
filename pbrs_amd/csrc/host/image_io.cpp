// pbrs_amd/csrc/host/image_io.cpp — the two image files the reference's front end writes (src/main.rs:28-53):
// `write_exr` (f32 RGB, what a render ends in, :245) and `write_image` (8-bit RGB PNG of `gamma_encode().to_u8()` pixels,
// radiometry/src/color.rs:13-23, :54-66).  The reference goes through the `exr` 1.4 and `png` 0.16 crates; only the file
// formats are shared with them: the EXR is a single-part scanline image with uncompressed FLOAT channels B, G, R, the PNG
// one zlib stream of unfiltered rows.  Pixel values are what is pinned (tests/test_image_io.py reads both back).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/pbrs_host.h"
#include "../../../include/pbrs_numeric.h"

namespace {

thread_local std::string g_io_error;

bool write_all(const char* path, const std::vector<unsigned char>& bytes) {
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        g_io_error = std::string("cannot create ") + path;
        return false;
    }
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (std::fclose(f) != 0 || !ok) {
        g_io_error = std::string("short write to ") + path;
        return false;
    }
    return true;
}
void put_u32(std::vector<unsigned char>& b, uint32_t v) {  // little endian (EXR)
    for (int k = 0; k < 4; ++k) b.push_back((unsigned char)(v >> (8 * k)));
}
void put_u64(std::vector<unsigned char>& b, uint64_t v) {
    for (int k = 0; k < 8; ++k) b.push_back((unsigned char)(v >> (8 * k)));
}
void put_str(std::vector<unsigned char>& b, const char* s) {  // with the terminating zero
    b.insert(b.end(), s, s + std::strlen(s) + 1);
}
void put_be32(std::vector<unsigned char>& b, uint32_t v) {  // big endian (PNG)
    for (int k = 3; k >= 0; --k) b.push_back((unsigned char)(v >> (8 * k)));
}
// radiometry/src/color.rs:13-23
unsigned char saturate_cast_u8(float f) {
    if (f > 1.0f) return 255;
    if (f >= 0.0f) return (unsigned char)(f * 255.0f);
    return 0;  // negative or NaN
}

}  // namespace

extern "C" {

const char* pbrs_host_io_error(void) { return g_io_error.c_str(); }

// src/main.rs:42-53: rgb is row-major, 3 floats per pixel, as pbrs_render_tile returns it
int pbrs_host_write_exr(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || width == 0 || height == 0) return PBRS_E_INVALID;
    std::vector<unsigned char> b;
    put_u32(b, 20000630u);  // magic
    put_u32(b, 2u);         // version 2, single-part scanline
    auto attr = [&](const char* name, const char* type, uint32_t size) {
        put_str(b, name);
        put_str(b, type);
        put_u32(b, size);
    };
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char* ch : {"B", "G", "R"}) {  // alphabetical, as the format requires
        put_str(b, ch);
        put_u32(b, 2u);  // FLOAT
        put_u32(b, 0u);  // pLinear + reserved
        put_u32(b, 1u);  // xSampling
        put_u32(b, 1u);  // ySampling
    }
    b.push_back(0);
    attr("compression", "compression", 1);
    b.push_back(0);  // NO_COMPRESSION
    for (const char* name : {"dataWindow", "displayWindow"}) {
        attr(name, "box2i", 16);
        put_u32(b, 0u);
        put_u32(b, 0u);
        put_u32(b, width - 1);
        put_u32(b, height - 1);
    }
    attr("lineOrder", "lineOrder", 1);
    b.push_back(0);  // INCREASING_Y
    attr("pixelAspectRatio", "float", 4);
    put_u32(b, pn_bits(1.0f));
    attr("screenWindowCenter", "v2f", 8);
    put_u32(b, 0u);
    put_u32(b, 0u);
    attr("screenWindowWidth", "float", 4);
    put_u32(b, pn_bits(1.0f));
    b.push_back(0);  // end of header
    const uint64_t row_bytes = 3ull * 4 * width, block = 8 + row_bytes;
    const uint64_t first = b.size() + 8ull * height;
    for (uint32_t y = 0; y < height; ++y) put_u64(b, first + block * y);  // scanline offset table
    for (uint32_t y = 0; y < height; ++y) {
        put_u32(b, y);
        put_u32(b, (uint32_t)row_bytes);
        for (int ch = 2; ch >= 0; --ch)  // B, G, R planes of the row
            for (uint32_t x = 0; x < width; ++x) put_u32(b, pn_bits(rgb[3 * ((size_t)y * width + x) + ch]));
    }
    return write_all(path, b) ? PBRS_OK : PBRS_E_INVALID;
}

// src/main.rs:28-40 fed with `color.gamma_encode().to_u8()` (:171): sqrt per channel, then the saturating cast
int pbrs_host_write_png(const char* path, const float* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || width == 0 || height == 0) return PBRS_E_INVALID;
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * (1 + 3 * (size_t)width));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter: none
        for (size_t k = 0; k < 3 * (size_t)width; ++k) raw.push_back(saturate_cast_u8(pn_sqrt(rgb[3 * (size_t)y * width + k])));
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        g_io_error = "deflate failed";
        return PBRS_E_INVALID;
    }
    z.resize(zlen);
    std::vector<unsigned char> b = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto chunk = [&](const char* type, const std::vector<unsigned char>& data) {
        put_be32(b, (uint32_t)data.size());
        const size_t start = b.size();
        b.insert(b.end(), type, type + 4);
        b.insert(b.end(), data.begin(), data.end());
        put_be32(b, (uint32_t)crc32(0L, b.data() + start, (uInt)(b.size() - start)));
    };
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8-bit RGB, deflate, adaptive filtering, no interlace
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    return write_all(path, b) ? PBRS_OK : PBRS_E_INVALID;
}

}  // extern "C"
