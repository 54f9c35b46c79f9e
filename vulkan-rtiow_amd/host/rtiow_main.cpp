// rtiow_main.cpp — C++ host harness standing where RTCHAP06/main.cpp stands: it
// fills the camera (main.cpp:101-120), uploads the scene, dispatches the render
// (main.cpp:313-325) and, instead of presenting to a swapchain (main.cpp:326-357)
// or taking a JPEG screenshot (Vulkan.cpp:625-766), writes a lossless PPM.
//
//   rtiow_main --scene cover|three|ch05|ch06 [--width W --height H --spp S --depth D
//              --seed N --chunk C --frames F --out file.ppm --device G]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtiow.h"

namespace {
int die(RtContext* ctx, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rtGetLastError(ctx));
    if (ctx) rtDestroy(ctx);
    return 1;
}
}  // namespace

int main(int argc, char** argv) {
    std::string scene = "cover", out = "frame.ppm";
    uint32_t width = 1200, height = 800, spp = 100, depth = 50, seed = 1, chunk = 10, frames = 1;
    int device = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char* v = argv[i + 1];
        if (k == "--scene") scene = v;
        else if (k == "--out") out = v;
        else if (k == "--width") width = std::strtoul(v, nullptr, 10);
        else if (k == "--height") height = std::strtoul(v, nullptr, 10);
        else if (k == "--spp") spp = std::strtoul(v, nullptr, 10);
        else if (k == "--depth") depth = std::strtoul(v, nullptr, 10);
        else if (k == "--seed") seed = std::strtoul(v, nullptr, 10);
        else if (k == "--chunk") chunk = std::strtoul(v, nullptr, 10);
        else if (k == "--frames") frames = std::strtoul(v, nullptr, 10);
        else if (k == "--device") device = std::atoi(v);
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }

    RtContext* ctx = nullptr;
    int rc = rtCreate(device, &ctx);
    if (rc != RT_OK) return die(nullptr, "rtCreate", rc);

    RtParams prm{};
    prm.width = width; prm.height = height; prm.spp = spp; prm.max_depth = depth; prm.seed = seed;
    prm.chunk_spp = chunk; prm.quantiser = RT_QUANT_BOOK; prm.mode = RT_MODE_PATH;
    RtCamera cam{};
    RtUbo5 ubo{};
    rtUboFromImage(width, height, &ubo);
    if (scene == "ch05" || scene == "ch06") {
        prm.mode = scene == "ch05" ? RT_MODE_CH05 : RT_MODE_CH06;
    } else {
        std::vector<RtSphere> sph(5000);
        std::vector<RtMaterial> mat(5000);
        uint32_t n = 0;
        if (scene == "three") {
            rc = rtMakeThreeSphereScene(1, sph.data(), mat.data(), 5000, &n);
            rtCameraFromUbo(&ubo, &cam);
        } else {
            const int half = scene == "cover4096" ? 32 : 11;
            rc = rtMakeCoverScene(seed, half, sph.data(), mat.data(), 5000, &n);
            const float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
            rtMakeCamera(from, at, up, 20.0f, float(width) / float(height), 0.1f, 10.0f, &cam);
        }
        if (rc != RT_OK) return die(ctx, "scene", rc);
        if ((rc = rtSetScene(ctx, sph.data(), mat.data(), n)) != RT_OK) return die(ctx, "rtSetScene", rc);
        std::printf("scene %s: %u spheres\n", scene.c_str(), n);
    }

    std::vector<uint8_t> frame(size_t(width) * height * 4);
    for (uint32_t f = 0; f < frames; ++f) {  // the frame loop of main.cpp:304-360
        const auto t0 = std::chrono::steady_clock::now();
        rc = rtRender(ctx, &cam, &prm, frame.data(), size_t(width) * 4, 0, nullptr);
        if (rc != RT_OK) return die(ctx, "rtRender", rc);
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        RtStats st{};
        rtGetStats(ctx, &st);
        const double nominal = double(width) * height * (prm.mode == RT_MODE_PATH ? double(spp) * depth : 1.0);
        std::printf("frame %u: kernel %.3f ms, wall %.3f ms, %.1f Mray/s nominal, %llu segments, %llu sphere tests\n",
                    f, st.kernel_ms, wall, nominal / (st.kernel_ms * 1e-3) / 1e6,
                    (unsigned long long)st.segments, (unsigned long long)st.sphere_tests);
    }
    if ((rc = rtWritePPM(out.c_str(), frame.data(), width, height, size_t(width) * 4)) != RT_OK)
        return die(ctx, "rtWritePPM", rc);
    std::printf("wrote %s\n", out.c_str());
    rtDestroy(ctx);
    return 0;
}
