// host_asan.cpp — the host side of the product and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer
// ("ASAN on host oracle"; GPU ASAN is not available on this pool).  Built by
// tests/host_asan/Makefile from the SAME sources the product uses -- rtiow_host.cpp (camera, scene builders, tiling,
// PPM/PNG writers), rtiow_clusters.cpp (the two-level list's build), host/scene_file.h (the scene-file parser) --
// plus oracle/rtiow_oracle.c, and run by tests/test_host_logic.py::test_host_side_under_sanitizers.  No GPU, no HIP
// runtime: the sources that launch kernels are not part of this build.  Exit status 0 = every check held and the
// sanitizers stayed silent.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "rtiow.h"
#include "rtiow_device.h"
#include "scene_file.h"

extern "C" {
int oracle_render(const RtSphere* spheres, const RtMaterial* materials, uint32_t n, const RtCamera* cam,
                  const RtParams* params, uint8_t* dst, size_t pitch, int nthreads, uint64_t* out_segments);
int oracle_render_ubo(const RtUbo5* ubo, uint32_t mode, uint8_t* dst, size_t pitch);
int oracle_make_cover_scene(uint32_t seed, int grid_half, RtSphere* sph, RtMaterial* mat, uint32_t cap, uint32_t* out_n);
}

static int g_failed = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond); \
            ++g_failed;                                                    \
        }                                                                  \
    } while (0)

static std::string write_tmp(const std::string& dir, const char* name, const std::string& text) {
    const std::string path = dir + "/" + name;
    std::ofstream(path) << text;
    return path;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";

    // ---- scene builders, camera, UBO -------------------------------------------------------------------
    std::vector<RtSphere> sph(6200);
    std::vector<RtMaterial> mat(6200);
    uint32_t n = 0;
    CHECK(rtMakeCoverScene(1, 11, sph.data(), mat.data(), 6200, &n) == RT_OK && n > 400 && n < 500);
    std::vector<RtSphere> osph(6200);
    std::vector<RtMaterial> omat(6200);
    uint32_t on = 0;
    CHECK(oracle_make_cover_scene(1, 11, osph.data(), omat.data(), 6200, &on) == RT_OK && on == n);
    CHECK(std::memcmp(sph.data(), osph.data(), n * sizeof(RtSphere)) == 0);
    CHECK(std::memcmp(mat.data(), omat.data(), n * sizeof(RtMaterial)) == 0);
    CHECK(rtMakeCoverScene(1, 11, sph.data(), mat.data(), 10, &n) != RT_OK);  // too little room: refused, not overrun
    CHECK(rtMakeCoverScene(1, 11, sph.data(), mat.data(), 6200, &n) == RT_OK);
    uint32_t n3 = 0;
    std::vector<RtSphere> s3(5);
    std::vector<RtMaterial> m3(5);
    CHECK(rtMakeThreeSphereScene(1, s3.data(), m3.data(), 5, &n3) == RT_OK && n3 == 5);
    CHECK(rtMakeThreeSphereScene(1, s3.data(), m3.data(), 4, &n3) != RT_OK);
    RtUbo5 ubo{};
    CHECK(rtUboFromImage(800, 608, &ubo) == RT_OK && ubo.viewportWidth == 2.0f);
    CHECK(rtUboFromImage(0, 608, &ubo) != RT_OK);
    RtCamera cam{};
    const float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    CHECK(rtMakeCamera(from, at, up, 20.0f, 1.5f, 0.1f, 10.0f, &cam) == RT_OK);
    CHECK(rtMakeCamera(from, from, up, 20.0f, 1.5f, 0.1f, 10.0f, &cam) != RT_OK);  // degenerate view direction
    CHECK(rtMakeCamera(from, at, up, 20.0f, 1.5f, 0.1f, 10.0f, &cam) == RT_OK);

    // ---- tiling arithmetic: every row owned exactly once, in ascending order ----------------------------
    for (uint32_t h : {1u, 7u, 800u, 2160u})
        for (uint32_t count : {1u, 2u, 3u, 8u, 64u})
            for (uint32_t block : {1u, 4u, 16u, 5000u}) {
                std::vector<int> seen(h, 0);
                for (uint32_t r = 0; r < count; ++r) {
                    const uint32_t rows = rtTileRowCount(h, block, r, count);
                    uint32_t prev = 0;
                    for (uint32_t lr = 0; lr < rows; ++lr) {
                        const uint32_t g = rtTileGlobalRow(lr, block, r, count);
                        CHECK(g < h && (lr == 0 || g > prev));
                        if (g < h) ++seen[g];
                        prev = g;
                    }
                }
                for (uint32_t y = 0; y < h; ++y) CHECK(seen[y] == 1);
            }

    // ---- the two-level list: every sphere in exactly one slot, padding marked, boxes finite --------------
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uni(-1.0f, 1.0f);
    for (uint32_t count : {1u, 15u, 16u, 17u, 64u, 486u, 1537u, 4099u, 6144u}) {
        std::vector<RtSphere> rs(count);
        for (uint32_t i = 0; i < count; ++i) {
            const float scale = i % 97 == 0 ? 40.0f : 1.0f;  // a few large ones
            rs[i] = RtSphere{uni(rng) * 30.0f, uni(rng) * 2.0f, uni(rng) * 30.0f, (0.05f + 0.3f * std::fabs(uni(rng))) * scale};
            if (i % 13 == 5) rs[i].radius = -rs[i].radius;  // hollow-glass style negative radii
        }
        for (double range : {2.0, 9.0, 64.0}) {
            rtiow::ClusterScene cs;
            rtiow::build_clusters(rs.data(), count, range, cs);
            CHECK(cs.slots.size() == cs.idx.size());
            CHECK(cs.slots.size() == size_t(cs.n_large_slots) + size_t(cs.n_clusters) * rtiow::kClusterStride);
            // (whole boxes: two entries each; then, in a flat scene, the boxes without the flat axis: one each)
            CHECK(cs.bounds.size() == (cs.flat_axis < 3u ? 3u : 2u) * (size_t(cs.n_clusters) + cs.n_super));
            CHECK(cs.n_clusters % rtiow::kSuperSize == 0);
            std::vector<int> seen(count, 0);
            for (size_t k = 0; k < cs.idx.size(); ++k) {
                if (cs.idx[k] == 0xFFFFFFFFu) {
                    CHECK(std::isinf(cs.slots[k].w) && cs.slots[k].w < 0);  // padding: r^2 = -inf, never hit
                } else {
                    CHECK(cs.idx[k] < count);
                    if (cs.idx[k] < count) ++seen[cs.idx[k]];
                }
            }
            for (uint32_t i = 0; i < count; ++i) CHECK(seen[i] == 1);
            for (const rtiow::ClusterF4& b : cs.bounds) CHECK(std::isfinite(b.x) && std::isfinite(b.y) && std::isfinite(b.z));
        }
    }

    // ---- writers: sizes, headers, refusal of bad arguments -------------------------------------------------
    {
        const uint32_t w = 37, h = 11;
        std::vector<uint8_t> img(size_t(w) * h * 4);
        for (size_t k = 0; k < img.size(); ++k) img[k] = uint8_t(k * 7);
        const std::string ppm = dir + "/asan.ppm", png = dir + "/asan.png";
        CHECK(rtWritePPM(ppm.c_str(), img.data(), w, h, size_t(w) * 4) == RT_OK);
        CHECK(rtWritePNG(png.c_str(), img.data(), w, h, size_t(w) * 4) == RT_OK);
        CHECK(rtWritePPM(ppm.c_str(), img.data(), w, h, size_t(w) * 4 - 1) != RT_OK);
        CHECK(rtWritePNG(nullptr, img.data(), w, h, size_t(w) * 4) != RT_OK);
        CHECK(rtWritePPM((dir + "/no/such/dir/x.ppm").c_str(), img.data(), w, h, size_t(w) * 4) == RT_ERR_IO);
        std::ifstream in(ppm, std::ios::binary);
        std::string magic;
        in >> magic;
        CHECK(magic == "P6");
        // a wide image: the PNG writer's stored-deflate blocks (64 KiB each) must split inside scanlines
        std::vector<uint8_t> wide(size_t(30000) * 3 * 4, 0x5A);
        CHECK(rtWritePNG(png.c_str(), wide.data(), 30000, 3, size_t(30000) * 4) == RT_OK);
    }

    // ---- the scene-file parser on well-formed and malformed input ----------------------------------------
    {
        std::vector<RtSphere> fs;
        std::vector<RtMaterial> fm;
        RtCamera fc{};
        bool have = false;
        const std::string good = write_tmp(dir, "good.txt",
            "# demo\ncamera -2 2 1  0 0 -1  0 1 0  40 0.05 3.4\nsphere 0 -100.5 -1 100 lambertian 0.8 0.8 0\n"
            "sphere 0 0 -1 0.5 metal 0.8 0.6 0.2 0.1   # trailing comment\nsphere -1 0 -1 -0.4 dielectric 1.5\n\n");
        CHECK(load_scene_file(good, 1.5f, fs, fm, fc, have) && have && fs.size() == 3 && fm[1].kind == RT_MAT_METAL);
        const char* bad[] = {
            "sphere 0 0 0\n",                                   // too few numbers
            "sphere 0 0 0 1 plastic 1 1 1\n",                   // unknown material
            "sphere 0 0 0 1 metal 0.5 0.5\n",                   // material cut short
            "camera 1 2 3\n",                                   // camera cut short
            "camera 0 0 0  0 0 0  0 1 0  40 0 1\nsphere 0 0 -1 0.5 lambertian 1 1 1\n",  // look-from == look-at
            "teapot 1 2 3\n",                                   // unknown item
            "# nothing but a comment\n",                        // no sphere at all
            "sphere nan nan nan nan dielectric\n",              // non-numbers where numbers belong
            "sphere 1e99999 0 0 1 lambertian 1 1 1\n",          // overflowing literal
            "sphere 0 0 0 1 lambertian 1 1 1 \xff\xfe\x00garbage", // binary junk after a valid line
        };
        int k = 0;
        for (const char* text : bad) {
            fs.clear();
            fm.clear();
            const std::string path = write_tmp(dir, ("bad" + std::to_string(k++) + ".txt").c_str(), text);
            const bool ok = load_scene_file(path, 1.5f, fs, fm, fc, have);
            // either refused, or accepted with arrays of equal length (the junk-after-a-valid-line case)
            CHECK(!ok || (fs.size() == fm.size() && !fs.empty()));
        }
        CHECK(!load_scene_file(dir + "/does_not_exist.txt", 1.5f, fs, fm, fc, have));
        std::string huge;  // many lines, long lines
        for (int i = 0; i < 5000; ++i) huge += "sphere " + std::to_string(i) + " 0 0 0.4 lambertian 0.5 0.5 0.5 " + std::string(i % 50, ' ') + "\n";
        fs.clear();
        fm.clear();
        CHECK(load_scene_file(write_tmp(dir, "huge.txt", huge), 1.5f, fs, fm, fc, have) && fs.size() == 5000);
    }

    // ---- the oracle itself on small frames (ragged tiles, one thread and several) --------------------------
    {
        const uint32_t w = 33, h = 17;
        RtParams p{};
        p.width = w; p.height = h; p.spp = 3; p.max_depth = 50; p.seed = 1; p.mode = RT_MODE_PATH; p.quantiser = RT_QUANT_BOOK;
        std::vector<uint8_t> a(size_t(w) * h * 4), b(size_t(w) * h * 4);
        uint64_t sa = 0, sb = 0;
        CHECK(oracle_render(sph.data(), mat.data(), n, &cam, &p, a.data(), size_t(w) * 4, 1, &sa) == RT_OK);
        CHECK(oracle_render(sph.data(), mat.data(), n, &cam, &p, b.data(), size_t(w) * 4, 4, &sb) == RT_OK);
        CHECK(a == b && sa == sb && sa >= uint64_t(w) * h * 3);
        p.row_block = 3; p.tile_count = 4; p.tile_rank = 3;
        const uint32_t rows = rtTileRowCount(h, 3, 3, 4);
        std::vector<uint8_t> t(size_t(w) * (rows ? rows : 1) * 4);
        CHECK(oracle_render(sph.data(), mat.data(), n, &cam, &p, t.data(), size_t(w) * 4, 2, &sb) == RT_OK);
        for (uint32_t lr = 0; lr < rows; ++lr)
            CHECK(std::memcmp(t.data() + size_t(lr) * w * 4, a.data() + size_t(rtTileGlobalRow(lr, 3, 3, 4)) * w * 4, size_t(w) * 4) == 0);
        std::vector<uint8_t> c(size_t(w) * h * 4);
        rtUboFromImage(w, h, &ubo);
        CHECK(oracle_render_ubo(&ubo, RT_MODE_CH06, c.data(), size_t(w) * 4) == RT_OK);
        CHECK(oracle_render_ubo(&ubo, 99, c.data(), size_t(w) * 4) != RT_OK);
    }

    if (g_failed) {
        std::fprintf(stderr, "host_asan: %d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("host_asan: all checks held\n");
    return 0;
}
