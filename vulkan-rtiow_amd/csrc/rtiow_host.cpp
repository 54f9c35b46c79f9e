// rtiow_host.cpp — host-side half of the C ABI that needs no GPU: the camera
// UBO the reference fills (RTCHAP06/main.cpp:101-120), the positionable camera,
// the scene builders, the row-tile arithmetic and the lossless frame writer
// that replaces saveScreenCap (RTCHAP06/Vulkan.cpp:625-766).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

#include "../../include/rtiow.h"
#include "rtiow_rng.h"

namespace {

void fill_material(RtMaterial& m, uint32_t kind, float r, float g, float b, float fuzz, float ior) {
    std::memset(&m, 0, sizeof m);
    m.kind = kind;
    m.albedo[0] = r;
    m.albedo[1] = g;
    m.albedo[2] = b;
    m.fuzz = fuzz;
    m.ior = ior;
}

struct SceneWriter {
    RtSphere* sph;
    RtMaterial* mat;
    uint32_t cap;
    uint32_t n = 0;
    bool push(float x, float y, float z, float r, uint32_t kind, float cr, float cg, float cb,
              float fuzz, float ior) {
        if (n >= cap) return false;
        sph[n] = RtSphere{x, y, z, r};
        fill_material(mat[n], kind, cr, cg, cb, fuzz, ior);
        ++n;
        return true;
    }
};

}  // namespace

extern "C" {

int rtAbiVersion(void) { return RTIOW_ABI_VERSION; }

int rtUboFromImage(uint32_t width, uint32_t height, RtUbo5* out) {
    if (!out || width == 0 || height == 0) return RT_ERR_INVALID;
    // RTCHAP06/main.cpp:103-120: the reference fixes the viewport WIDTH at 2.
    const float aspect = static_cast<float>(width) / static_cast<float>(height);
    out->imageWidth = static_cast<float>(width);
    out->imageHeight = static_cast<float>(width) / aspect;
    out->viewportWidth = 2.0f;
    out->viewportHeight = 2.0f / aspect;
    out->focalLength = 1.0f;
    return RT_OK;
}

int rtCameraFromUbo(const RtUbo5* ubo, RtCamera* out) {
    if (!ubo || !out) return RT_ERR_INVALID;
    // raytrace06.comp:53-56: origin at 0, viewport spanned by +x / +y, looking down -z.
    std::memset(out, 0, sizeof *out);
    out->horizontal[0] = ubo->viewportWidth;
    out->vertical[1] = ubo->viewportHeight;
    out->lower_left[0] = 0.0f - ubo->viewportWidth / 2;
    out->lower_left[1] = 0.0f - ubo->viewportHeight / 2;
    out->lower_left[2] = 0.0f - ubo->focalLength;
    out->u[0] = out->v[1] = out->w[2] = 1.0f;
    return RT_OK;
}

int rtMakeCamera(const float lookfrom[3], const float lookat[3], const float vup[3],
                 float vfov_deg, float aspect, float aperture, float focus_dist, RtCamera* out) {
    if (!lookfrom || !lookat || !vup || !out) return RT_ERR_INVALID;
    // Basis and viewport in double, rounded to float once (hard part 1:
    // tan() never runs on the device).
    const double theta = static_cast<double>(vfov_deg) * 3.14159265358979323846 / 180.0;
    const double half_h = std::tan(theta / 2.0);
    const double vp_h = 2.0 * half_h;
    const double vp_w = static_cast<double>(aspect) * vp_h;
    double w[3], u[3], v[3];
    for (int k = 0; k < 3; ++k) w[k] = double(lookfrom[k]) - double(lookat[k]);
    const double wl = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (!(wl > 0.0)) return RT_ERR_INVALID;
    for (double& c : w) c /= wl;
    u[0] = double(vup[1]) * w[2] - double(vup[2]) * w[1];
    u[1] = double(vup[2]) * w[0] - double(vup[0]) * w[2];
    u[2] = double(vup[0]) * w[1] - double(vup[1]) * w[0];
    const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (!(ul > 0.0)) return RT_ERR_INVALID;
    for (double& c : u) c /= ul;
    v[0] = w[1] * u[2] - w[2] * u[1];
    v[1] = w[2] * u[0] - w[0] * u[2];
    v[2] = w[0] * u[1] - w[1] * u[0];
    for (int k = 0; k < 3; ++k) {
        const double hor = double(focus_dist) * vp_w * u[k];
        const double ver = double(focus_dist) * vp_h * v[k];
        out->origin[k] = lookfrom[k];
        out->horizontal[k] = static_cast<float>(hor);
        out->vertical[k] = static_cast<float>(ver);
        out->lower_left[k] = static_cast<float>(double(lookfrom[k]) - hor / 2.0 - ver / 2.0 -
                                                double(focus_dist) * w[k]);
        out->u[k] = static_cast<float>(u[k]);
        out->v[k] = static_cast<float>(v[k]);
        out->w[k] = static_cast<float>(w[k]);
    }
    out->lens_radius = static_cast<float>(double(aperture) / 2.0);
    return RT_OK;
}

int rtMakeCoverScene(uint32_t seed, int grid_half, RtSphere* spheres, RtMaterial* materials,
                     uint32_t capacity, uint32_t* out_n) {
    if (!spheres || !materials || !out_n || grid_half < 0) return RT_ERR_INVALID;
    SceneWriter sw{spheres, materials, capacity};
    rtiow::Pcg rng(seed, 0x5CE7E5EEu, 0u);
    bool ok = sw.push(0.0f, -1000.0f, 0.0f, 1000.0f, RT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0, 0);
    for (int a = -grid_half; a < grid_half && ok; ++a) {
        for (int b = -grid_half; b < grid_half && ok; ++b) {
            const float choose = rng.uniform();
            const float cx = static_cast<float>(a) + 0.9f * rng.uniform();
            const float cz = static_cast<float>(b) + 0.9f * rng.uniform();
            const float dx = cx - 4.0f, dy = 0.2f - 0.2f, dz = cz - 0.0f;
            const float dist = std::sqrt(std::fmaf(dz, dz, std::fmaf(dy, dy, dx * dx)));
            if (!(dist > 0.9f)) continue;
            if (choose < 0.8f) {
                float p[3], q[3];
                for (float& c : p) c = rng.uniform();
                for (float& c : q) c = rng.uniform();
                ok = sw.push(cx, 0.2f, cz, 0.2f, RT_MAT_LAMBERTIAN, p[0] * q[0], p[1] * q[1],
                             p[2] * q[2], 0, 0);
            } else if (choose < 0.95f) {
                float p[3];
                for (float& c : p) c = 0.5f + 0.5f * rng.uniform();
                const float fuzz = 0.5f * rng.uniform();
                ok = sw.push(cx, 0.2f, cz, 0.2f, RT_MAT_METAL, p[0], p[1], p[2], fuzz, 0);
            } else {
                ok = sw.push(cx, 0.2f, cz, 0.2f, RT_MAT_DIELECTRIC, 1, 1, 1, 0, 1.5f);
            }
        }
    }
    ok = ok && sw.push(0.0f, 1.0f, 0.0f, 1.0f, RT_MAT_DIELECTRIC, 1, 1, 1, 0, 1.5f);
    ok = ok && sw.push(-4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0, 0);
    ok = ok && sw.push(4.0f, 1.0f, 0.0f, 1.0f, RT_MAT_METAL, 0.7f, 0.6f, 0.5f, 0, 0);
    if (!ok) return RT_ERR_INVALID;
    *out_n = sw.n;
    return RT_OK;
}

int rtMakeThreeSphereScene(int with_bubble, RtSphere* spheres, RtMaterial* materials,
                           uint32_t capacity, uint32_t* out_n) {
    if (!spheres || !materials || !out_n) return RT_ERR_INVALID;
    SceneWriter sw{spheres, materials, capacity};
    bool ok = sw.push(0.0f, -100.5f, -1.0f, 100.0f, RT_MAT_LAMBERTIAN, 0.8f, 0.8f, 0.0f, 0, 0);
    ok = ok && sw.push(0.0f, 0.0f, -1.0f, 0.5f, RT_MAT_LAMBERTIAN, 0.1f, 0.2f, 0.5f, 0, 0);
    ok = ok && sw.push(-1.0f, 0.0f, -1.0f, 0.5f, RT_MAT_DIELECTRIC, 1, 1, 1, 0, 1.5f);
    if (with_bubble)
        ok = ok && sw.push(-1.0f, 0.0f, -1.0f, -0.4f, RT_MAT_DIELECTRIC, 1, 1, 1, 0, 1.5f);
    ok = ok && sw.push(1.0f, 0.0f, -1.0f, 0.5f, RT_MAT_METAL, 0.8f, 0.6f, 0.2f, 0.0f, 0);
    if (!ok) return RT_ERR_INVALID;
    *out_n = sw.n;
    return RT_OK;
}

uint32_t rtTileRowCount(uint32_t height, uint32_t row_block, uint32_t tile_rank,
                        uint32_t tile_count) {
    if (tile_count <= 1) return height;
    if (row_block == 0) row_block = 1;
    // full cycles of tile_count blocks, then the ragged remainder
    const uint32_t cycle = row_block * tile_count;
    const uint32_t full = height / cycle;
    const uint32_t rem = height % cycle;
    uint32_t rows = full * row_block;
    const uint32_t start = tile_rank * row_block;
    if (rem > start) rows += (rem - start < row_block) ? rem - start : row_block;
    return rows;
}

uint32_t rtTileGlobalRow(uint32_t local_row, uint32_t row_block, uint32_t tile_rank,
                         uint32_t tile_count) {
    if (tile_count <= 1) return local_row;
    if (row_block == 0) row_block = 1;
    return ((local_row / row_block) * tile_count + tile_rank) * row_block + local_row % row_block;
}

int rtWritePPM(const char* path, const void* rgba8, uint32_t width, uint32_t height,
               size_t pitch) {
    if (!path || !rgba8 || width == 0 || height == 0 || pitch < size_t(width) * 4)
        return RT_ERR_INVALID;
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> line(size_t(width) * 3);
    const unsigned char* base = static_cast<const unsigned char*>(rgba8);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; ++y) {
        // rt.frag:8 samples at 1-t: the top line of the picture is buffer row H-1
        const unsigned char* row = base + size_t(height - 1 - y) * pitch;
        for (uint32_t x = 0; x < width; ++x) {
            line[3 * x + 0] = row[4 * x + 0];
            line[3 * x + 1] = row[4 * x + 1];
            line[3 * x + 2] = row[4 * x + 2];
        }
        ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_IO;
}

namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {  // PNG chunk CRC (ISO 3309), bitwise table-free
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back(static_cast<unsigned char>(x >> 24));
    v.push_back(static_cast<unsigned char>(x >> 16));
    v.push_back(static_cast<unsigned char>(x >> 8));
    v.push_back(static_cast<unsigned char>(x));
}
bool write_chunk(std::FILE* f, const char type[4], const std::vector<unsigned char>& data) {
    std::vector<unsigned char> head;
    put_be32(head, static_cast<uint32_t>(data.size()));
    head.insert(head.end(), type, type + 4);
    uint32_t crc = crc32_update(0xFFFFFFFFu, reinterpret_cast<const unsigned char*>(type), 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xFFFFFFFFu;
    std::vector<unsigned char> tail;
    put_be32(tail, crc);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}
}  // namespace

int rtWritePNG(const char* path, const void* rgba8, uint32_t width, uint32_t height, size_t pitch) {
    if (!path || !rgba8 || width == 0 || height == 0 || pitch < size_t(width) * 4) return RT_ERR_INVALID;
    // raw scanlines: filter byte 0 + RGB, top line of the picture (buffer row H-1, rt.frag:8) first
    const size_t line = 1 + size_t(width) * 3;
    std::vector<unsigned char> raw(line * height);
    const unsigned char* base = static_cast<const unsigned char*>(rgba8);
    for (uint32_t y = 0; y < height; ++y) {
        const unsigned char* row = base + size_t(height - 1 - y) * pitch;
        unsigned char* dst = raw.data() + line * y;
        dst[0] = 0;
        for (uint32_t x = 0; x < width; ++x) {
            dst[1 + 3 * x + 0] = row[4 * x + 0];
            dst[1 + 3 * x + 1] = row[4 * x + 1];
            dst[1 + 3 * x + 2] = row[4 * x + 2];
        }
    }
    // zlib stream of stored deflate blocks (<= 65535 bytes each) + Adler-32
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00
        z.push_back(static_cast<unsigned char>(n & 0xFF));
        z.push_back(static_cast<unsigned char>(n >> 8));
        z.push_back(static_cast<unsigned char>(~n & 0xFF));
        z.push_back(static_cast<unsigned char>((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) {
            s1 = (s1 + raw[off + i]) % 65521u;
            s2 = (s2 + s1) % 65521u;
        }
        if (raw.empty()) break;
    }
    put_be32(z, (s2 << 16) | s1);
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const unsigned char tail[5] = {8, 2, 0, 0, 0};  // 8 bits, colour type 2 (RGB), deflate, filter 0, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    ok = ok && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_IO;
}

}  // extern "C"
