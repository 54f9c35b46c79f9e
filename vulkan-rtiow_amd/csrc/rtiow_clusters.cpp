// rtiow_clusters.cpp — host-side build of the two-level sphere list used by the clustered
// persistent kernel ("acceleration structure", not in the reference or
// in RTIOW book 1; book 2's bvh_node is the natural next step after hittable_list).
//
// Spheres are sorted along a Morton curve and cut into clusters of kClusterSize; each cluster gets
// a bounding sphere.  Very large spheres (the ground) become clusters of their own whose bound IS
// the sphere, bit for bit.  The kernel tests every cluster bound for every ray segment (wave in
// lock-step, LDS broadcast) and then, lane by lane, only the members of the clusters the ray can
// reach.  The result is the same closest hit the flat list gives — the bound of a cluster is
// inflated far beyond the rounding error of the bound test, so no hit can be culled, and the
// minimum over (distance, original index) does not depend on the order of the tests.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "rtiow_device.h"

namespace rtiow {

namespace {
uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
}  // namespace

void build_clusters(const RtSphere* sph, uint32_t n, ClusterScene& out) {
    out.slots.clear();
    out.idx.clear();
    out.bounds.clear();
    // radius scale: median |r|
    std::vector<float> radii(n);
    for (uint32_t i = 0; i < n; ++i) radii[i] = std::fabs(sph[i].radius);
    std::vector<float> sorted_r = radii;
    std::nth_element(sorted_r.begin(), sorted_r.begin() + n / 2, sorted_r.end());
    const float median = sorted_r[n / 2];
    std::vector<uint32_t> small, large;
    for (uint32_t i = 0; i < n; ++i) (radii[i] > 4.0f * median ? large : small).push_back(i);
    // extent of the small spheres (for Morton codes) and of everything (for the rounding margin)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    double alo[3] = {1e300, 1e300, 1e300}, ahi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < n; ++i) {
        const double c[3] = {sph[i].cx, sph[i].cy, sph[i].cz};
        const bool is_small = radii[i] <= 4.0f * median;
        for (int k = 0; k < 3; ++k) {
            if (is_small) {
                lo[k] = std::min(lo[k], c[k]);
                hi[k] = std::max(hi[k], c[k]);
            }
            // the huge spheres count with the part of them near the others only: use the centres of
            // the small ones and the surfaces of everything up to 16 median radii away
            alo[k] = std::min(alo[k], is_small ? c[k] - radii[i] : alo[k]);
            ahi[k] = std::max(ahi[k], is_small ? c[k] + radii[i] : ahi[k]);
        }
    }
    if (small.empty()) {
        for (int k = 0; k < 3; ++k) lo[k] = hi[k] = alo[k] = ahi[k] = 0.0;
    }
    double diag2 = 0.0;
    for (int k = 0; k < 3; ++k) diag2 += (ahi[k] - alo[k]) * (ahi[k] - alo[k]);
    // Rays start inside a few scene diameters of the spheres; |oc|^2 up to (4 diag)^2 = 16 diag2.
    // binary32 error of hb^2 - (|oc|^2 - R^2) is a few ulps of |oc|^2: keep 64 ulps of margin in R^2.
    const double r2_margin = 16.0 * diag2 * 64.0 * 5.96e-8 + 1e-6;
    for (int k = 0; k < 3; ++k) out.center[k] = static_cast<float>(0.5 * (alo[k] + ahi[k]));
    out.diag = static_cast<float>(std::sqrt(diag2));

    std::vector<uint32_t> code(n, 0u);
    for (uint32_t i : small) {
        uint32_t q[3];
        const double c[3] = {sph[i].cx, sph[i].cy, sph[i].cz};
        for (int k = 0; k < 3; ++k) {
            const double span = hi[k] - lo[k];
            q[k] = span > 0.0 ? static_cast<uint32_t>(std::min(1023.0, (c[k] - lo[k]) / span * 1023.0)) : 0u;
        }
        code[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
    std::stable_sort(small.begin(), small.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });

    auto emit = [&](const uint32_t* members, uint32_t count, bool exact) {
        // slots: kClusterStride per cluster, the first kClusterSize real or never-hit padding
        const size_t base = out.slots.size();
        out.slots.resize(base + kClusterStride, ClusterF4{0.0f, 0.0f, 0.0f, -1.0f});
        out.idx.resize(base + kClusterStride, 0xFFFFFFFFu);
        double c[3] = {0, 0, 0};
        for (uint32_t m = 0; m < count; ++m) {
            const RtSphere& s = sph[members[m]];
            out.slots[base + m] = ClusterF4{s.cx, s.cy, s.cz, s.radius * s.radius};
            out.idx[base + m] = members[m];
            c[0] += s.cx; c[1] += s.cy; c[2] += s.cz;
        }
        ClusterF4 b;
        if (exact) {  // a single large sphere: the bound test IS the sphere test, same floats
            const RtSphere& s = sph[members[0]];
            b = ClusterF4{s.cx, s.cy, s.cz, s.radius * s.radius};
        } else {
            for (double& v : c) v /= count;
            const float cf[3] = {static_cast<float>(c[0]), static_cast<float>(c[1]), static_cast<float>(c[2])};
            double r = 0.0;
            for (uint32_t m = 0; m < count; ++m) {
                const RtSphere& s = sph[members[m]];
                const double dx = s.cx - double(cf[0]), dy = s.cy - double(cf[1]), dz = s.cz - double(cf[2]);
                r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs(double(s.radius)));
            }
            const double r2 = r * r * (1.0 + 1e-5) + r2_margin;
            b = ClusterF4{cf[0], cf[1], cf[2], std::nextafter(static_cast<float>(r2), INFINITY)};
        }
        out.bounds.push_back(b);
    };
    for (uint32_t i : large) emit(&i, 1u, true);
    for (size_t k = 0; k < small.size(); k += kClusterSize)
        emit(&small[k], static_cast<uint32_t>(std::min<size_t>(kClusterSize, small.size() - k)), false);
    // pad the cluster count to a multiple of 8 (the unroll of the bound loop) with unreachable bounds
    while (out.bounds.size() % 8u) {
        out.bounds.push_back(ClusterF4{0.0f, 0.0f, 0.0f, -1.0f});
        out.slots.resize(out.slots.size() + kClusterStride, ClusterF4{0.0f, 0.0f, 0.0f, -1.0f});
        out.idx.resize(out.idx.size() + kClusterStride, 0xFFFFFFFFu);
    }
    out.n_clusters = static_cast<uint32_t>(out.bounds.size());
}

}  // namespace rtiow
