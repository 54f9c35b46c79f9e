// scene_file.h — the text scene format of the host harness (parameters instead of the compile-time
// constants of RTCHAP06/main.cpp:15-16,101-102,116-120).  One item per line, '#' starts a comment:
//   camera  fx fy fz  ax ay az  ux uy uz  vfov aperture focus
//   sphere  cx cy cz radius  lambertian r g b | metal r g b fuzz | dielectric ior
// Header-only so that the harness (rtiow_main.cpp) and the host-side sanitizer run (tests/host_asan.cpp) parse
// with the same code.
#pragma once
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rtiow.h"

// scene file -> arrays + camera; returns false with a message on a malformed line
inline bool load_scene_file(const std::string& path, float aspect, std::vector<RtSphere>& sph,
                            std::vector<RtMaterial>& mat, RtCamera& cam, bool& have_cam) {
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); return false; }
    std::string line;
    int ln = 0;
    have_cam = false;
    while (std::getline(in, line)) {
        ++ln;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ls(line);
        std::string what;
        if (!(ls >> what)) continue;
        if (what == "camera") {
            float f[3], a[3], u[3], vfov, aperture, focus;
            if (!(ls >> f[0] >> f[1] >> f[2] >> a[0] >> a[1] >> a[2] >> u[0] >> u[1] >> u[2] >> vfov >> aperture >> focus) ||
                rtMakeCamera(f, a, u, vfov, aspect, aperture, focus, &cam) != RT_OK) {
                std::fprintf(stderr, "%s:%d: bad camera\n", path.c_str(), ln);
                return false;
            }
            have_cam = true;
        } else if (what == "sphere") {
            RtSphere s{};
            RtMaterial m{};
            std::string kind;
            if (!(ls >> s.cx >> s.cy >> s.cz >> s.radius >> kind)) { std::fprintf(stderr, "%s:%d: bad sphere\n", path.c_str(), ln); return false; }
            bool ok = true;
            if (kind == "lambertian") { m.kind = RT_MAT_LAMBERTIAN; ok = bool(ls >> m.albedo[0] >> m.albedo[1] >> m.albedo[2]); }
            else if (kind == "metal") { m.kind = RT_MAT_METAL; ok = bool(ls >> m.albedo[0] >> m.albedo[1] >> m.albedo[2] >> m.fuzz); }
            else if (kind == "dielectric") { m.kind = RT_MAT_DIELECTRIC; m.albedo[0] = m.albedo[1] = m.albedo[2] = 1.0f; ok = bool(ls >> m.ior); }
            else ok = false;
            if (!ok) { std::fprintf(stderr, "%s:%d: bad material\n", path.c_str(), ln); return false; }
            sph.push_back(s);
            mat.push_back(m);
        } else {
            std::fprintf(stderr, "%s:%d: unknown item '%s'\n", path.c_str(), ln, what.c_str());
            return false;
        }
    }
    return !sph.empty();
}

