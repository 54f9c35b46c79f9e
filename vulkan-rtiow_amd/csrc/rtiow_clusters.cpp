// rtiow_clusters.cpp — host-side build of the two-level sphere list used by the clustered
// persistent kernel ("acceleration structure", not in the reference or
// in RTIOW book 1; book 2's bvh_node / aabb is the natural next step after hittable_list).
//
// Very large spheres (the ground) go into a short list of their own that every ray tests exactly.
// The others are grouped by recursive median splits into clusters of kClusterSize; each cluster gets
// an axis-aligned box, stored as centre + half extent.  The kernel tests every box for every ray
// segment (wave in lock-step, LDS broadcast, slab test) and then, lane by lane, only the members of
// the boxes the ray can reach.
//
// The result is the same closest hit the flat list gives:
//   * a member is found by the same discriminant / first-root-beyond-t_min arithmetic as in the flat
//     list, and the minimum over (distance, original index) does not depend on the order of the tests;
//   * a box can only cull spheres the exact test would reject.  binary32 evaluates the discriminant
//     hb^2 - (|oc|^2 - r^2) with an absolute error below ~22 eps |oc|^2 (eps = 2^-24),
//     so a ray the exact test accepts passes within sqrt(r^2 + E) of the centre, E = 48 eps |oc|max^2.
//     Every member is boxed with that radius plus the slab test's own rounding (a shift of the planes
//     by a few eps of the coordinates), for ray origins within rmax of the scene's centre (2 scene
//     diagonals, more when the camera stands farther out: range_diags); the kernel sends a ray that
//     starts beyond rmax through all clusters.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "rtiow_device.h"

namespace rtiow {

namespace {
float round_up(double v) {  // smallest float >= v
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafter(f, INFINITY);
    return f;
}
}  // namespace

static std::atomic<unsigned long long> g_cluster_builds{0};
unsigned long long cluster_build_count() { return g_cluster_builds.load(); }

void build_clusters(const RtSphere* sph, uint32_t n, double range_diags, ClusterScene& out) {
    g_cluster_builds.fetch_add(1);
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
    // extent of the small spheres' surfaces (for the margins)
    double alo[3] = {1e300, 1e300, 1e300}, ahi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i : small) {
        const double c[3] = {sph[i].cx, sph[i].cy, sph[i].cz};
        for (int k = 0; k < 3; ++k) {
            alo[k] = std::min(alo[k], c[k] - radii[i]);
            ahi[k] = std::max(ahi[k], c[k] + radii[i]);
        }
    }
    double diag2 = 0.0, cmax = 0.0;
    for (int k = 0; k < 3; ++k) {
        diag2 += (ahi[k] - alo[k]) * (ahi[k] - alo[k]);
        out.center[k] = static_cast<float>(0.5 * (alo[k] + ahi[k]));
        cmax = std::max(cmax, std::fabs(static_cast<double>(out.center[k])));
    }
    const double diag = std::sqrt(diag2);
    out.diag = static_cast<float>(diag);
    // (the padded cluster count, as computed below; more than super_from of them: a level of super-clusters above)
    uint32_t super_from = kSuperFrom;
    if (const char* v = debug_knob("RTIOW_DEBUG_SUPER_FROM")) super_from = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));  // tuning only
    const size_t clusters_to_be = ((small.size() + kClusterSize - 1u) / kClusterSize + kSuperSize - 1u) / kSuperSize * kSuperSize;
    const bool with_supers = clusters_to_be > super_from;
    // range_diags <= 0: the scene's own range -- kRangeTwoLevel for one with super-clusters, kRangeOneLevel otherwise (rtiow_device.h)
    if (!(range_diags > 0.0)) range_diags = with_supers ? kRangeTwoLevel : kRangeOneLevel;
    out.range_diags = std::max(kRangeFloor, range_diags);
    // rays that start within rmax of the centre use the boxes; |oc| <= rmax + diag/2 for them
    double rmax = out.range_diags * diag;
    if (const char* v = debug_knob("RTIOW_DEBUG_RANGE")) rmax = std::atof(v) * diag;  // tuning only (tools/range_ab.py)
    out.rmax2 = static_cast<float>(rmax * rmax);
    constexpr double kEps = 5.9604644775390625e-8;  // 2^-24
    const double oc_max = rmax * 1.05 + 0.5 * diag;  // rmax + diag/2, and the rounding of the kernel's own range check
    const double r2_margin = 48.0 * kEps * oc_max * oc_max;
    const double plane_margin = 64.0 * kEps * (cmax + rmax + diag) + 1e-30;
    // Rays that start farther out, q from the centre: the same two margins with q for rmax are sqrt(r^2 + 48 eps (1.05 q + diag/2)^2)
    // - r <= sqrt(48 eps) (1.05 q + diag/2) for the sphere and 64 eps (cmax + q + diag) for the planes, whatever the radius: a box
    // with every half extent enlarged by far_k q + far_c contains what such a ray's exact test can accept.  Linear in q, so the
    // kernel works it out per ray (trace_clustered, the expansion stage of the two-level trace: a ground plane that reaches the
    // horizon starts a few rays in a thousand out there, and each of them used to take EVERY cluster of a large scene).
    out.far_k = round_up((1.05 * std::sqrt(48.0 * kEps) + 64.0 * kEps) * 1.001);
    out.far_c = round_up((0.5 * diag * std::sqrt(48.0 * kEps) + 64.0 * kEps * (cmax + diag) + 1e-30) * 1.001);

    // Order the small spheres so that every run of kClusterSize is a compact group: split the set at
    // the median of its longest axis, the left part rounded to whole clusters, and recurse.  (Runs of a
    // Morton curve give boxes a ray meets 2.1x as often on the cover scene: tools/cull_sim.py.)
    struct Range { size_t lo, hi; };
    constexpr size_t kSuperSpan = size_t(kSuperSize) * kClusterSize;  // spheres under one super-cluster
    std::vector<Range> todo{{0, small.size()}};
    while (!todo.empty()) {
        const Range rg = todo.back();
        todo.pop_back();
        const size_t count = rg.hi - rg.lo;
        if (count <= kClusterSize) continue;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = rg.lo; k < rg.hi; ++k) {
            const float c[3] = {sph[small[k]].cx, sph[small[k]].cy, sph[small[k]].cz};
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], c[d]);
                mx[d] = std::max(mx[d], c[d]);
            }
        }
        int axis = 0;
        for (int d = 1; d < 3; ++d)
            if (mx[d] - mn[d] > mx[axis] - mn[axis]) axis = d;
        auto coord = [&](uint32_t i) { return axis == 0 ? sph[i].cx : (axis == 1 ? sph[i].cy : sph[i].cz); };
        std::stable_sort(small.begin() + rg.lo, small.begin() + rg.hi, [&](uint32_t x, uint32_t y) {
            const float cx = coord(x), cy = coord(y);
            return cx < cy || (cx == cy && x < y);
        });
        size_t left = (count / 2 + kClusterSize / 2) / kClusterSize * kClusterSize;
        if (left == 0) left = kClusterSize;
        // A scene that gets super-clusters (below) splits its larger ranges at whole SUPER-clusters, so that the kSuperSize
        // consecutive clusters under one super box are one subtree of the splits: rounded to whole clusters only, a super
        // straddled the boundary of two subtrees wherever the cluster count was not a power of two -- two distant corners under
        // one box -- and a 3138-sphere scene (200 clusters) made 103 tests per segment where the 4099-sphere one (256) makes 68.
        if (with_supers && count > kSuperSpan) {
            left = (count / 2 + kSuperSpan / 2) / kSuperSpan * kSuperSpan;
            if (left == 0) left = kSuperSpan;
        }
        if (left >= count) left = count - 1;
        todo.push_back({rg.lo, rg.lo + left});
        todo.push_back({rg.lo + left, rg.hi});
    }

    // padding: r^2 = -inf makes the discriminant -inf (or NaN) for every ray, so a padding slot is never a
    // candidate; a finite value would not do (far from the origin, hb^2 - |o|^2 exceeds it through rounding alone)
    const ClusterF4 never{0.0f, 0.0f, 0.0f, -INFINITY};
    // the large spheres: one group of slots, padded to whole clusters (the sparse trace scans all slots)
    out.n_large_slots = static_cast<uint32_t>((large.size() + kClusterSize - 1u) / kClusterSize * kClusterSize);
    out.slots.assign(out.n_large_slots, never);
    out.idx.assign(out.n_large_slots, 0xFFFFFFFFu);
    for (size_t m = 0; m < large.size(); ++m) {
        const RtSphere& s = sph[large[m]];
        out.slots[m] = ClusterF4{s.cx, s.cy, s.cz, s.radius * s.radius};
        out.idx[m] = large[m];
    }
    out.n_large = static_cast<uint32_t>(large.size());

    for (size_t k0 = 0; k0 < small.size(); k0 += kClusterSize) {
        const uint32_t count = static_cast<uint32_t>(std::min<size_t>(kClusterSize, small.size() - k0));
        const size_t base = out.slots.size();
        out.slots.resize(base + kClusterStride, never);
        out.idx.resize(base + kClusterStride, 0xFFFFFFFFu);
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (uint32_t m = 0; m < count; ++m) {
            const RtSphere& s = sph[small[k0 + m]];
            out.slots[base + m] = ClusterF4{s.cx, s.cy, s.cz, s.radius * s.radius};
            out.idx[base + m] = small[k0 + m];
            const double c[3] = {s.cx, s.cy, s.cz};
            const double ext = std::sqrt(double(s.radius) * double(s.radius) + r2_margin) + plane_margin;
            for (int k = 0; k < 3; ++k) {
                blo[k] = std::min(blo[k], c[k] - ext);
                bhi[k] = std::max(bhi[k], c[k] + ext);
            }
        }
        ClusterF4 mid{0, 0, 0, 0}, half{0, 0, 0, 0};
        float* mp[3] = {&mid.x, &mid.y, &mid.z};
        float* hp[3] = {&half.x, &half.y, &half.z};
        for (int k = 0; k < 3; ++k) {
            const float m = static_cast<float>(0.5 * (blo[k] + bhi[k]));
            *mp[k] = m;
            *hp[k] = round_up(std::max(bhi[k] - double(m), double(m) - blo[k]));
        }
        out.bounds.push_back(mid);
        out.bounds.push_back(half);
    }
    // pad the cluster count to a multiple of kSuperSize (and of the box loop's unroll of 4) with empty
    // clusters whose box is a point far outside any scene (reaching it by accident only costs 16
    // never-hit tests)
    const size_t n_real_clusters = out.bounds.size() / 2u;
    while ((out.bounds.size() / 2u) % kSuperSize) {
        out.bounds.push_back(ClusterF4{3e18f, 3e18f, 3e18f, 0.0f});
        out.bounds.push_back(ClusterF4{0.0f, 0.0f, 0.0f, 0.0f});
        out.slots.resize(out.slots.size() + kClusterStride, never);
        out.idx.resize(out.idx.size() + kClusterStride, 0xFFFFFFFFu);
    }
    out.n_clusters = static_cast<uint32_t>(out.bounds.size() / 2u);
    // Large scenes get a level above: super-clusters of kSuperSize consecutive clusters (neighbours in
    // the split tree), boxed by the union of their clusters' boxes.  The kernel then tests the super
    // boxes in lock-step and the cluster boxes only for the (ray, super-cluster) pairs that pass.
    out.n_super = 0;
    if (out.n_clusters > super_from) {
        out.n_super = out.n_clusters / kSuperSize;
        for (uint32_t sc = 0; sc < out.n_super; ++sc) {
            double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
            bool any = false;
            for (uint32_t j = 0; j < kSuperSize; ++j) {
                const size_t c = size_t(sc) * kSuperSize + j;
                if (c >= n_real_clusters) continue;  // padding holds nothing
                const ClusterF4 &mid = out.bounds[2u * c], &half = out.bounds[2u * c + 1u];
                const double m[3] = {mid.x, mid.y, mid.z}, h[3] = {half.x, half.y, half.z};
                for (int k = 0; k < 3; ++k) {
                    blo[k] = std::min(blo[k], m[k] - h[k]);
                    bhi[k] = std::max(bhi[k], m[k] + h[k]);
                }
                any = true;
            }
            ClusterF4 mid{3e18f, 3e18f, 3e18f, 0.0f}, half{0.0f, 0.0f, 0.0f, 0.0f};
            if (any) {
                float* mp[3] = {&mid.x, &mid.y, &mid.z};
                float* hp[3] = {&half.x, &half.y, &half.z};
                for (int k = 0; k < 3; ++k) {
                    const float m = static_cast<float>(0.5 * (blo[k] + bhi[k]));
                    *mp[k] = m;
                    // the child boxes already carry the slab test's margin; keep the union exact, rounded outwards
                    *hp[k] = round_up(std::max(bhi[k] - double(m), double(m) - blo[k]));
                }
            }
            out.bounds.push_back(mid);
            out.bounds.push_back(half);
        }
    }
    // The flat axis.  Scenes of this renderer mostly stand on a ground plane: every cluster box then spans (nearly) the
    // same interval along the up axis, and the kernel's slab test can take that axis ONCE per ray against the common
    // interval instead of once per box (rtiow_kernels.hip: slab_gap_flat, 10 instead of 14 instructions per ray and box,
    // one 16-byte LDS read per box instead of two).  A wider interval only ever lets more boxes through, so using the
    // union of the boxes' intervals is conservative by construction; it is taken when no real box is narrower than two
    // thirds of the union (the cull would lose too much otherwise) along the best such axis.
    out.flat_axis = 3u;
    out.flat_mid = out.flat_half = 0.0f;
    if (n_real_clusters > 0u) {
        const char* force = debug_knob("RTIOW_DEBUG_FLAT");  // (1: every scene flat along its best axis -- parity tests only)
        double best_ratio = force && std::atoi(force) == 1 ? 1e300 : 1.5;
        for (uint32_t axis = 0; axis < 3u; ++axis) {
            double lo = 1e300, hi = -1e300, narrowest = 1e300;
            for (size_t c = 0; c < n_real_clusters; ++c) {
                const ClusterF4 &mid = out.bounds[2u * c], &half = out.bounds[2u * c + 1u];
                const double m = axis == 0 ? mid.x : (axis == 1 ? mid.y : mid.z), h = axis == 0 ? half.x : (axis == 1 ? half.y : half.z);
                lo = std::min(lo, m - h);
                hi = std::max(hi, m + h);
                narrowest = std::min(narrowest, 2.0 * h);
            }
            const double ratio = (hi - lo) / std::max(narrowest, 1e-300);
            if (ratio <= best_ratio) {
                best_ratio = ratio;
                out.flat_axis = axis;
                out.flat_mid = static_cast<float>(0.5 * (lo + hi));
                out.flat_half = round_up(std::max(hi - double(out.flat_mid), double(out.flat_mid) - lo));
            }
        }
    }
    // the boxes again without the flat axis, one float4 {mid a, mid b, half a, half b} each (a, b: the other two axes in
    // x, y, z order), clusters then super-clusters: appended to `bounds` behind the full boxes
    if (out.flat_axis < 3u) {
        const size_t n_boxes = out.bounds.size() / 2u;
        for (size_t c = 0; c < n_boxes; ++c) {
            const ClusterF4 mid = out.bounds[2u * c], half = out.bounds[2u * c + 1u];
            const float m[3] = {mid.x, mid.y, mid.z}, h[3] = {half.x, half.y, half.z};
            const uint32_t ia = out.flat_axis == 0u ? 1u : 0u, ib = out.flat_axis == 2u ? 1u : 2u;
            out.bounds.push_back(ClusterF4{m[ia], m[ib], h[ia], h[ib]});
        }
    }
}

void build_scene_clusters(const RtSphere* spheres, uint32_t n, ClusterScene& out) {
    build_clusters(spheres, n, 0.0, out);  // the scene's own range, decided before a box is made: one build
    if (out.n_super != 0u && clustered_levels_that_fit(static_cast<uint32_t>(out.slots.size()), out.n_clusters, out.n_super, out.flat_axis < 3u) == 1)
        build_clusters(spheres, n, kRangeOneLevel, out);  // (the kernels will drop the super level: boxes for a one-level trace)
}

}  // namespace rtiow
