// host_png.cpp — minimal PNG reader/writer over zlib for the CLI and the OBJ loader.
//
// Replaces the reference's use of the `image` crate: `image::open` + `convert_color_space(SRGB_LINEAR)`
// for textures (tracing/src/material.rs:143-151) and `save_with_format(.., Png)` for the result
// (pathtracer-cli/src/main.rs:186-188).  8-bit, non-interlaced grey / grey+alpha / RGB / RGBA / palette.
// sRGB decode is the piecewise IEC 61966-2-1 EOTF evaluated in double and rounded once to f32 per
// 8-bit code (the `image`/`moxcms` arithmetic is not vendored: parity unpinned, tolerance-based).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "host_internal.hpp"

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t n) {
    put32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put32(out, (uint32_t)crc32(0, out.data() + start, (uInt)(n + 4)));
}
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

using namespace ptx;

extern "C" {

int ptx_png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) return fail(PTX_E_ARG, "ptx_png_write_rgb8: bad argument");
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + (size_t)y * width * 3, rgb + (size_t)(y + 1) * width * 3);
    }
    // One zlib stream assembled from independently deflated stripes (raw deflate, each but the last ended with
    // a sync flush so that the pieces concatenate), the way pigz does: the stripes compress in parallel.
    unsigned workers = std::thread::hardware_concurrency();
    workers = workers > 16 ? 16 : (workers < 1 ? 1 : workers);
    size_t n_stripes = raw.size() / (256u << 10);  // at least 256 KiB per stripe
    n_stripes = n_stripes < 1 ? 1 : (n_stripes > workers ? workers : n_stripes);
    const size_t stripe = (raw.size() + n_stripes - 1) / n_stripes;
    std::vector<std::vector<uint8_t>> parts(n_stripes);
    std::vector<int> ok(n_stripes, 0);
    auto deflate_stripe = [&](size_t k) {
        const size_t lo = k * stripe, hi = lo + stripe < raw.size() ? lo + stripe : raw.size();
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
        parts[k].resize(deflateBound(&zs, (uLong)(hi - lo)) + 16);
        zs.next_in = const_cast<Bytef*>(raw.data() + lo);
        zs.avail_in = (uInt)(hi - lo);
        zs.next_out = parts[k].data();
        zs.avail_out = (uInt)parts[k].size();
        const int rc = deflate(&zs, k + 1 == n_stripes ? Z_FINISH : Z_SYNC_FLUSH);
        ok[k] = (k + 1 == n_stripes ? rc == Z_STREAM_END : rc == Z_OK) && zs.avail_in == 0;
        parts[k].resize(zs.total_out);
        deflateEnd(&zs);
    };
    {
        std::vector<std::thread> pool;
        for (size_t k = 1; k < n_stripes; ++k) pool.emplace_back(deflate_stripe, k);
        deflate_stripe(0);
        for (auto& t : pool) t.join();
    }
    for (int good : ok)
        if (!good) return fail(PTX_E_IO, "zlib deflate failed");
    std::vector<uint8_t> z = {0x78, 0x9C};  // zlib header: deflate, 32 KiB window, default level
    for (auto& part : parts) z.insert(z.end(), part.begin(), part.end());
    const uLong adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    put32(z, (uint32_t)adler);
    uLongf zn = (uLongf)z.size();
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width);
    put32(ihdr, height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8-bit RGB
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zn);
    chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) return fail(PTX_E_IO, std::string("cannot create ") + path);
    size_t w = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return w == out.size() ? PTX_OK : fail(PTX_E_IO, std::string("short write to ") + path);
}

int ptx_png_read_linear(const char* path, float** rgb_out, uint32_t* width, uint32_t* height) {
    if (!path || !rgb_out || !width || !height) return fail(PTX_E_ARG, "ptx_png_read_linear: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PTX_E_IO, std::string("cannot open ") + path);
    std::vector<uint8_t> file;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return fail(PTX_E_IO, std::string(path) + ": not a PNG");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t n = be32(&file[pos]);
        if (pos + 12 + (size_t)n > file.size()) return fail(PTX_E_IO, std::string(path) + ": truncated chunk");
        const uint8_t* type = &file[pos + 4];
        const uint8_t* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4) && n >= 13) {
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + n);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + n);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)n;
    }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!w || !h || depth != 8 || !channels || interlace)
        return fail(PTX_E_IO, std::string(path) + ": unsupported PNG (need 8-bit, non-interlaced)");
    if ((uint64_t)w * h > (1ull << 28)) return fail(PTX_E_LIMIT, std::string(path) + ": image too large");
    size_t stride = (size_t)w * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawn = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawn, idat.data(), (uLong)idat.size()) != Z_OK || rawn != raw.size())
        return fail(PTX_E_IO, std::string(path) + ": inflate failed");
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* in = &raw[(stride + 1) * y];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        int filter = in[0];
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)channels ? cur[x - channels] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= (size_t)channels) ? up[x - channels] : 0;
            int v = in[1 + x];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return fail(PTX_E_IO, std::string(path) + ": bad PNG filter");
            }
            cur[x] = (uint8_t)v;
        }
    }
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    float* out = (float*)malloc((size_t)w * h * 3 * sizeof(float));
    if (!out) return fail(PTX_E_OOM, "ptx_png_read_linear: out of memory");
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t r, g, b;
        const uint8_t* p = &img[i * channels];
        if (ctype == 3) {
            if ((size_t)p[0] * 3 + 2 >= plte.size()) {
                free(out);
                return fail(PTX_E_IO, std::string(path) + ": palette index out of range");
            }
            r = plte[p[0] * 3], g = plte[p[0] * 3 + 1], b = plte[p[0] * 3 + 2];
        } else if (channels <= 2) {
            r = g = b = p[0];
        } else {
            r = p[0], g = p[1], b = p[2];
        }
        out[3 * i] = lut[r];
        out[3 * i + 1] = lut[g];
        out[3 * i + 2] = lut[b];
    }
    *rgb_out = out;
    *width = w;
    *height = h;
    return PTX_OK;
}

// Accumulation-buffer checkpoint (the reference has none: a run that stops loses its samples).
// Little-endian file: "PTXACC01", width u32, height u32, seed u64, samples u32, crc32 of the pixel bytes u32,
// then width * height * 3 f32 sums.  Samples are keyed by (seed, pixel, sample index) and added in index
// order, so rendering [samples, N) on top of a loaded buffer gives the bits of one uninterrupted run.
int ptx_accum_save(const char* path, const float* accum_rgb, uint32_t width, uint32_t height, uint64_t seed,
                   uint32_t samples) {
    if (!path || !accum_rgb || !width || !height) return fail(PTX_E_ARG, "ptx_accum_save: bad argument");
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    uLong crc = crc32(0, nullptr, 0);
    for (size_t off = 0; off < bytes; off += (size_t)1 << 30) {
        const size_t n = bytes - off < ((size_t)1 << 30) ? bytes - off : (size_t)1 << 30;
        crc = crc32(crc, reinterpret_cast<const Bytef*>(accum_rgb) + off, (uInt)n);
    }
    const std::string tmp = std::string(path) + ".tmp";  // written aside and renamed: never a torn checkpoint
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(PTX_E_IO, std::string("ptx_accum_save: cannot create ") + tmp);
    uint8_t head[32];
    memcpy(head, "PTXACC01", 8);
    memcpy(head + 8, &width, 4);
    memcpy(head + 12, &height, 4);
    memcpy(head + 16, &seed, 8);
    memcpy(head + 24, &samples, 4);
    const uint32_t c32 = (uint32_t)crc;
    memcpy(head + 28, &c32, 4);
    bool ok = fwrite(head, 1, sizeof head, f) == sizeof head && fwrite(accum_rgb, 1, bytes, f) == bytes;
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        remove(tmp.c_str());
        return fail(PTX_E_IO, std::string("ptx_accum_save: cannot write ") + path);
    }
    return PTX_OK;
}

// accum_rgb == NULL only reads the header (to size the buffer); otherwise the buffer must hold
// *width x *height as passed in, and the file must match them.
int ptx_accum_load(const char* path, float* accum_rgb, uint32_t* width, uint32_t* height, uint64_t* seed,
                   uint32_t* samples) {
    if (!path || !width || !height) return fail(PTX_E_ARG, "ptx_accum_load: bad argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PTX_E_IO, std::string("ptx_accum_load: cannot open ") + path);
    uint8_t head[32];
    if (fread(head, 1, sizeof head, f) != sizeof head || memcmp(head, "PTXACC01", 8) != 0) {
        fclose(f);
        return fail(PTX_E_IO, std::string("ptx_accum_load: not an accumulation checkpoint: ") + path);
    }
    uint32_t w, h, n, c32;
    uint64_t sd;
    memcpy(&w, head + 8, 4);
    memcpy(&h, head + 12, 4);
    memcpy(&sd, head + 16, 8);
    memcpy(&n, head + 24, 4);
    memcpy(&c32, head + 28, 4);
    if (seed) *seed = sd;
    if (samples) *samples = n;
    if (!accum_rgb) {
        *width = w, *height = h;
        fclose(f);
        return PTX_OK;
    }
    if (w != *width || h != *height) {
        fclose(f);
        return fail(PTX_E_ARG, "ptx_accum_load: the checkpoint is " + std::to_string(w) + "x" + std::to_string(h) +
                                   ", the buffer " + std::to_string(*width) + "x" + std::to_string(*height));
    }
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    const bool ok = fread(accum_rgb, 1, bytes, f) == bytes && fgetc(f) == EOF;
    fclose(f);
    if (!ok) return fail(PTX_E_IO, std::string("ptx_accum_load: truncated or oversized checkpoint: ") + path);
    uLong crc = crc32(0, nullptr, 0);
    for (size_t off = 0; off < bytes; off += (size_t)1 << 30) {
        const size_t k = bytes - off < ((size_t)1 << 30) ? bytes - off : (size_t)1 << 30;
        crc = crc32(crc, reinterpret_cast<const Bytef*>(accum_rgb) + off, (uInt)k);
    }
    if ((uint32_t)crc != c32) return fail(PTX_E_IO, std::string("ptx_accum_load: checksum mismatch in ") + path);
    return PTX_OK;
}

}  // extern "C"
