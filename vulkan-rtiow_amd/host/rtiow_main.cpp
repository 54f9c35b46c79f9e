// rtiow_main.cpp — C++ host harness standing where RTCHAP06/main.cpp stands: it
// fills the camera (main.cpp:101-120), uploads the scene, dispatches the render
// (main.cpp:313-325) and, instead of presenting to a swapchain (main.cpp:326-357)
// or taking a JPEG screenshot (Vulkan.cpp:625-766), writes a lossless PPM.
//
//   rtiow_main --scene cover|cover4096|three|ch05|ch06|file [--file scene.txt] [--width W --height H
//              --spp S --depth D --seed N --kernel K --frames F --progressive 0|1 --out file.ppm|file.png
//              --device G | --gpus N | --devices g0,g1,... --json 0|1]
//
// --json 1 prints one JSON object per frame (W, H, spp, depth, spheres, GPUs, kernel and wall ms, nominal Mray/s, segments,
// sphere tests: the metrics line SURVEY section 5 asks of the harness; the reference logs nothing) instead of the text,
// and after the last frame one summary object with bench.py's fields -- metric, value, n_gpus, steps, warmup (--warmup W
// of the F frames are left out of it), ms_per_step, and under "config" the transport, every device's tile kernel ms,
// the frame's CRC-32 and, on the multi-GPU path, whether the gathered frame is the one device 0 renders alone -- so the
// C-ABI path (rtMultiRender) is timed by the same yardstick as the torch path:
//     rtiow_main --gpus 8 --frames 12 --warmup 2 --json 1        (rehearsal on one GPU: --devices 0,0,0,0,0,0,0,0)
//
// --gpus N renders every frame on devices 0..N-1 of this node from this one process (rtCreateMulti: block-cyclic
// row tiles, one RCCL gather to device 0, de-interleave there); --devices names them (a repeated device is the
// one-GPU rehearsal of the N-tile path).  The reference drives one device (RTCHAP06/Vulkan.cpp:87-97,122).
//
// --progressive 1 makes the F frames a running average (RtParams.accumulate): frame f adds spp new
// samples to the picture.  A scene file (parameters instead of the compile-time
// constants of main.cpp:15-16,101-102,116-120) is plain text, one item per line, '#' comments:
//   camera  fx fy fz  ax ay az  ux uy uz  vfov aperture focus
//   sphere  cx cy cz radius  lambertian r g b | metal r g b fuzz | dielectric ior
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtiow.h"
#include "scene_file.h"

namespace {
uint32_t crc32_of(const uint8_t* p, size_t n) {  // zlib's CRC-32 (reflected 0xEDB88320), as tests/golden/frame_golden.json holds it
    static uint32_t table[256];
    if (table[1] == 0u)
        for (uint32_t i = 0; i < 256u; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int die(RtContext* ctx, const char* what, int rc) {
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rtGetLastError(ctx));
    if (ctx) rtDestroy(ctx);
    return 1;
}
}  // namespace

int main(int argc, char** argv) {
    std::string scene = "cover", out = "frame.ppm", file;
    uint32_t width = 1200, height = 800, spp = 100, depth = 50, seed = 1, kernel = 0, frames = 1, progressive = 0, json = 0, warmup = 0;
    uint32_t n_spheres = 1;
    int device = 0;
    std::vector<int> devices;  // --gpus / --devices: the multi-GPU path
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
        else if (k == "--kernel") kernel = std::strtoul(v, nullptr, 10);
        else if (k == "--progressive") progressive = std::strtoul(v, nullptr, 10);
        else if (k == "--file") file = v;
        else if (k == "--json") json = std::strtoul(v, nullptr, 10);
        else if (k == "--frames") frames = std::strtoul(v, nullptr, 10);
        else if (k == "--warmup") warmup = std::strtoul(v, nullptr, 10);
        else if (k == "--device") device = std::atoi(v);
        else if (k == "--gpus") { devices.clear(); for (int g = 0; g < std::atoi(v); ++g) devices.push_back(g); }
        else if (k == "--devices") {
            devices.clear();
            std::istringstream ls(v);
            std::string tok;
            while (std::getline(ls, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
        }
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }

    RtContext* ctx = nullptr;
    RtMulti* multi = nullptr;
    int rc;
    if (!devices.empty()) {
        rc = rtCreateMulti(devices.data(), static_cast<int>(devices.size()), &multi);
        if (rc != RT_OK) { std::fprintf(stderr, "rtCreateMulti failed (%d): %s\n", rc, rtMultiGetLastError(nullptr)); return 1; }
        if (!json) std::printf("%d devices, transport %s\n", rtMultiDeviceCount(multi), rtMultiTransport(multi));
    } else {
        rc = rtCreate(device, &ctx);
        if (rc != RT_OK) return die(nullptr, "rtCreate", rc);
    }
    auto die_multi = [&](const char* what, int code) {
        std::fprintf(stderr, "%s failed (%d): %s\n", what, code, rtMultiGetLastError(multi));
        rtDestroyMulti(multi);
        return 1;
    };

    RtParams prm{};
    prm.width = width; prm.height = height; prm.spp = spp; prm.max_depth = depth; prm.seed = seed;
    prm.kernel = kernel; prm.quantiser = RT_QUANT_BOOK; prm.mode = RT_MODE_PATH;
    RtCamera cam{};
    RtUbo5 ubo{};
    rtUboFromImage(width, height, &ubo);
    if (scene == "ch05" || scene == "ch06") {
        prm.mode = scene == "ch05" ? RT_MODE_CH05 : RT_MODE_CH06;
        if (multi && devices.size() > 1) { std::fprintf(stderr, "ch05/ch06 are one 7-us dispatch: use --device\n"); rtDestroyMulti(multi); return 2; }
    } else {
        std::vector<RtSphere> sph(5000);
        std::vector<RtMaterial> mat(5000);
        uint32_t n = 0;
        if (scene == "file") {
            sph.clear();
            mat.clear();
            bool have_cam = false;
            if (!load_scene_file(file, float(width) / float(height), sph, mat, cam, have_cam)) { rtDestroy(ctx); rtDestroyMulti(multi); return 2; }
            if (!have_cam) rtCameraFromUbo(&ubo, &cam);
            n = static_cast<uint32_t>(sph.size());
            rc = RT_OK;
        } else if (scene == "three") {
            rc = rtMakeThreeSphereScene(1, sph.data(), mat.data(), 5000, &n);
            rtCameraFromUbo(&ubo, &cam);
        } else {
            const int half = scene == "cover4096" ? 32 : 11;
            rc = rtMakeCoverScene(seed, half, sph.data(), mat.data(), 5000, &n);
            const float from[3] = {13, 2, 3}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
            rtMakeCamera(from, at, up, 20.0f, float(width) / float(height), 0.1f, 10.0f, &cam);
        }
        if (rc != RT_OK) { rtDestroyMulti(multi); return die(ctx, "scene", rc); }
        if (multi) {
            if ((rc = rtMultiSetScene(multi, sph.data(), mat.data(), n)) != RT_OK) return die_multi("rtMultiSetScene", rc);
        } else if ((rc = rtSetScene(ctx, sph.data(), mat.data(), n)) != RT_OK) return die(ctx, "rtSetScene", rc);
        n_spheres = n;
        if (!json) std::printf("scene %s: %u spheres\n", scene.c_str(), n);
    }

    std::vector<uint8_t> frame(size_t(width) * height * 4);
    if (warmup >= frames) warmup = frames ? frames - 1 : 0;
    double sum_frame_ms = 0.0, sum_wall_ms = 0.0;  // over the frames after the warm-up
    std::vector<double> sum_dev_ms(devices.empty() ? 1 : devices.size(), 0.0);
    unsigned long long last_segments = 0, last_tests = 0;
    for (uint32_t f = 0; f < frames; ++f) {  // the frame loop of main.cpp:304-360
        const auto t0 = std::chrono::steady_clock::now();
        if (progressive && prm.mode == RT_MODE_PATH) {  // running average over the frames so far
            prm.accumulate = 1;
            prm.sample_offset = f * spp;
        }
        RtStats st{};
        if (multi) {
            rc = rtMultiRender(multi, &cam, &prm, frame.data(), size_t(width) * 4, 0);
            if (rc != RT_OK) return die_multi("rtMultiRender", rc);
        } else {
            rc = rtRender(ctx, &cam, &prm, frame.data(), size_t(width) * 4, 0, nullptr);
            if (rc != RT_OK) return die(ctx, "rtRender", rc);
        }
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (multi) {  // whole-frame time on the root; segments and tests summed over the devices' tiles
            double frame_ms = 0.0;
            RtStats part{};
            for (int g = 0; g < rtMultiDeviceCount(multi); ++g) {
                if ((rc = rtMultiGetStats(multi, g, &part, &frame_ms)) != RT_OK) return die_multi("rtMultiGetStats", rc);
                if (!json) std::printf("  device %d: tile kernel %.3f ms, %u rows\n", devices[g], part.kernel_ms, part.rows_rendered);
                if (f >= warmup) sum_dev_ms[g] += part.kernel_ms;
                st.segments += part.segments;
                st.sphere_tests += part.sphere_tests;
            }
            st.kernel_ms = frame_ms;
        } else {
            rtGetStats(ctx, &st);
            if (f >= warmup) sum_dev_ms[0] += st.kernel_ms;
        }
        if (f >= warmup) {
            sum_frame_ms += st.kernel_ms;
            sum_wall_ms += wall;
        }
        last_segments = st.segments;
        last_tests = st.sphere_tests;
        const double nominal = double(width) * height * (prm.mode == RT_MODE_PATH ? double(spp) * depth : 1.0);
        if (json)
            std::printf("{\"frame\": %u, \"scene\": \"%s\", \"width\": %u, \"height\": %u, \"spp\": %u, \"max_depth\": %u, "
                        "\"spheres\": %u, \"gpus\": %d, \"kernel_ms\": %.4f, \"wall_ms\": %.4f, \"mray_per_s_nominal\": %.1f, "
                        "\"segments\": %llu, \"sphere_tests\": %llu}\n",
                        f, scene.c_str(), width, height, prm.mode == RT_MODE_PATH ? spp : 1u, prm.mode == RT_MODE_PATH ? depth : 1u,
                        n_spheres, multi ? rtMultiDeviceCount(multi) : 1, st.kernel_ms, wall, nominal / (st.kernel_ms * 1e-3) / 1e6,
                        (unsigned long long)st.segments, (unsigned long long)st.sphere_tests);
        else
            std::printf("frame %u: kernel %.3f ms, wall %.3f ms, %.1f Mray/s nominal, %llu segments, %llu sphere tests\n",
                        f, st.kernel_ms, wall, nominal / (st.kernel_ms * 1e-3) / 1e6,
                        (unsigned long long)st.segments, (unsigned long long)st.sphere_tests);
    }
    if (json && frames > 0) {
        // the summary line: bench.py's fields for this path.  ms_per_step = device-side time of a frame, one frame in flight
        // (single GPU: the dispatch; several: first launch to assembled frame on the root); wall adds the copy to the host.
        const uint32_t steps = frames - warmup;
        const double ms = sum_frame_ms / steps;
        const double nominal = double(width) * height * (prm.mode == RT_MODE_PATH ? double(spp) * depth : 1.0);
        const uint32_t crc = crc32_of(frame.data(), frame.size());
        const char* vs_single = "not checked";
        if (multi && devices.size() > 1 && !progressive) {  // the same frame by device 0 alone, through the single-GPU entry point
            RtContext* one = nullptr;
            std::vector<uint8_t> whole(frame.size());
            bool ok = rtCreate(devices[0], &one) == RT_OK;
            if (ok && prm.mode == RT_MODE_PATH) {
                std::vector<RtSphere> sph(5000);
                std::vector<RtMaterial> mat(5000);
                uint32_t n = 0;
                if (scene == "file") {
                    bool have_cam = false;
                    RtCamera c2{};
                    sph.clear();
                    mat.clear();
                    ok = load_scene_file(file, float(width) / float(height), sph, mat, c2, have_cam);
                    n = static_cast<uint32_t>(sph.size());
                } else if (scene == "three") {
                    ok = rtMakeThreeSphereScene(1, sph.data(), mat.data(), 5000, &n) == RT_OK;
                } else {
                    ok = rtMakeCoverScene(seed, scene == "cover4096" ? 32 : 11, sph.data(), mat.data(), 5000, &n) == RT_OK;
                }
                ok = ok && rtSetScene(one, sph.data(), mat.data(), n) == RT_OK;
            }
            ok = ok && rtRender(one, &cam, &prm, whole.data(), size_t(width) * 4, 0, nullptr) == RT_OK;
            vs_single = !ok ? "single-GPU render failed" : (whole == frame ? "identical" : "DIFFERS");
            rtDestroy(one);
        }
        std::printf("{\"metric\": \"Mray/s (w*h*spp*depth / s)\", \"value\": %.1f, \"unit\": \"Mray/s\", \"n_gpus\": %d, \"steps\": %u, "
                    "\"warmup\": %u, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"strong\", \"dtype\": \"f32\", "
                    "\"data\": \"synthetic\", \"config\": {\"workload\": \"%s_%ux%u_%uspp\", \"path\": \"C ABI, one process: %s\", "
                    "\"transport\": \"%s\", \"frames_in_flight\": 1, \"destination\": \"host memory\", \"wall_ms_per_step\": %.4f, "
                    "\"device_kernel_ms\": [",
                    nominal / (ms * 1e-3) / 1e6, multi ? rtMultiDeviceCount(multi) : 1, steps, warmup, ms, scene.c_str(), width, height,
                    prm.mode == RT_MODE_PATH ? spp : 1u, multi ? "rtMultiRender" : "rtRender", multi ? rtMultiTransport(multi) : "single",
                    sum_wall_ms / steps);
        for (size_t g = 0; g < sum_dev_ms.size(); ++g) std::printf("%s%.4f", g ? ", " : "", sum_dev_ms[g] / steps);
        std::printf("], \"devices\": [");
        for (size_t g = 0; g < sum_dev_ms.size(); ++g) std::printf("%s%d", g ? ", " : "", devices.empty() ? device : devices[g]);
        RtSceneStats ss{};
        if (!multi && ctx) rtGetSceneStats(ctx, &ss);  // (single device: what rtSetScene cost and chose)
        std::printf("], \"segments_per_frame\": %llu, \"sphere_tests_per_frame\": %llu, \"frame_crc32\": %u, "
                    "\"scene_build_ms\": %.3f, \"cluster_builds\": %u, \"cluster_range_diags\": %.3g, "
                    "\"gathered_frame_vs_single_gpu_frame\": \"%s\"}}\n",
                    last_segments, last_tests, crc, ss.scene_build_ms, ss.cluster_builds, ss.range_diags, vs_single);
    }
    const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    rc = png ? rtWritePNG(out.c_str(), frame.data(), width, height, size_t(width) * 4)
             : rtWritePPM(out.c_str(), frame.data(), width, height, size_t(width) * 4);
    if (rc != RT_OK) { rtDestroyMulti(multi); return die(ctx, png ? "rtWritePNG" : "rtWritePPM", rc); }
    if (!json) std::printf("wrote %s\n", out.c_str());
    rtDestroy(ctx);
    rtDestroyMulti(multi);
    return 0;
}
