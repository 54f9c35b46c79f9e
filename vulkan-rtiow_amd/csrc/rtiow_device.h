// rtiow_device.h — launch interface between the C ABI (rtiow_capi.hip) and the
// gfx950 kernels (rtiow_kernels.hip).  Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "../../include/rtiow.h"

namespace rtiow {

// Tuning and test knobs (RTIOW_DEBUG_* environment variables) exist in the diagnostic builds only -- -DRTIOW_DEBUG_KNOBS:
// librtiow_hip_knobs.so (the shipped kernels + the knobs: what the A/B tools and two parity tests load) and librtiow_hip_dbg.so.
// The shipped library never reads the environment on its render path: no variable can change its kernel choice, and a frame's
// dispatch makes no getenv calls (rounds 1-3: about a dozen per frame).
#ifdef RTIOW_DEBUG_KNOBS
inline const char* debug_knob(const char* name) { return std::getenv(name); }
#else
inline const char* debug_knob(const char*) { return nullptr; }
#endif

// Device-side counters, one block per context (zeroed before every render).
// Head of one per-XCD pixel queue, alone on its 128-byte line: atomics on one line are serialised by
// its L2 channel (~90 per microsecond), eight heads on one line would share that budget.
struct alignas(128) QueueHead {
    unsigned int next;              // first virtual pixel not handed out yet
    unsigned int pad[31];
};

struct Counters {
    QueueHead xcd_head[8];          // persistent kernel: heads of the eight per-XCD pixel queues (their last part, handed
                                    // out in pools of a few pixels, counted from where the whole chunks end)
    QueueHead xcd_chunk[8];         // ... and of their first part, handed out in whole chunks: one atomic per fetch, no look
    unsigned int wg_done;           // persistent kernel: workgroups that have left (the last one out resets the OTHER
    unsigned int pad0;              // counter block for the next frame: no memset between frames)
    unsigned long long paths;
    unsigned long long segments;
    unsigned long long tests;       // ray-sphere and ray-bound tests performed
    unsigned long long debug[8];    // diagnostic builds (-DRTIOW_DEBUG_COUNTERS) only
    unsigned long long clk_cycles, clk_ticks;  // persistent kernels: shader cycles and ticks of the constant 100 MHz counter the first wave
                                    // of workgroup 0 saw from its start to its end -> RtStats.shader_clock_mhz
    unsigned long long not_t0;      // diagnostic builds: ~(earliest wave start), 100 MHz ticks
    unsigned int hist_dry[32];      // diagnostic builds: waves by the time their queue ran dry, 0.125 ms bins
    unsigned int hist_end[32];      // diagnostic builds: waves by the time they finished
    unsigned long long tail_iters, tail_ticks;  // diagnostic builds: iterations / 100 MHz ticks of all waves after running dry
    unsigned long long tail_sparse_iters, tail_sparse_ticks, tail_sparse_paths;  // the sparse ones among them
    unsigned long long tail_cyc[3];  // diagnostic builds: shader cycles of the tail iterations in refill / trace / shade
    unsigned long long pass_stats[10]; // diagnostic builds, primary pass: passes, camera rays, cluster trips, paths that go on, shader cycles, shade cycles,
                                       // cycles handing out samples, cycles in the camera code, cycles moving records into slots, times that was done
    // -DRTIOW_DEBUG_TIMELINE builds: waves by the time (50 us bins from the first wave's start) their queue ran dry,
    // they first took the sparse trace, and they finished; iterations after running dry
    unsigned int tl_hist[3][64];
    unsigned int tl_progress[10];   // 100 MHz ticks at which queue 0's head passed k/8 of its pixels
    unsigned int tl_tail_iters_max, tl_pad;
    unsigned long long tl_tail_iters_sum;
    unsigned long long tl_sparse_iters_sum, tl_sparse_paths_sum, tl_sparse_ticks_sum;  // from a wave's first sparse iteration on
    unsigned long long tl_starved_sum;      // iterations in which a wave could not open a pixel for want of an accumulator entry
    unsigned int tl_starved_max, tl_live_at_dry_max;
    unsigned int tl_live_at_dry_hist[9];    // waves by their live paths when they found the queues dry (bins of 16)
    unsigned int tl_late_dry_live_hist[9];  // the same for the waves that ran dry more than 100 us after the first one
    unsigned long long not_first_dry;       // ~(earliest dry stamp)
#ifdef RTIOW_DEBUG_TIMELINE
    // one record per wave (RTIOW_DEBUG_WAVELOG=file dumps them): dry, first sparse iteration, end (100 MHz ticks from the first
    // wave's start), iterations after dry, sparse ones among them, live paths at dry | paths summed over its sparse iterations << 8,
    // deepest path finished after dry, when the last path of 40+ segments finished
    unsigned int tl_wave[8192][8];
    unsigned long long tl_bucket[5][4];  // sparse iterations by paths held (1-2, 3-4, 5-8, 9-16, 17-32): count, ticks, trace cycles, shade cycles
    unsigned long long tl_start_sum[5];    // per wave, 10 ns ticks: entry -> staged (barrier passed); staged -> first camera rays made; waves that made any;
                                           // staged -> first pool fetched; staged -> first hand_out done
    unsigned int tl_start_max[2];
    unsigned long long tl_depth_hist[64];  // finished paths by the segments they took, from 8 on (63: that many or more)
#endif
#ifdef RTIOW_BLOCK_COUNTERS
    // tools/blockprof builds only: one execution counter per basic block of the instrumented kernel, each on a 128-byte line of its own
    // (written by s_atomic_add instructions that tools/blockprof/instrument.py puts into the kernel's assembly; RTIOW_BLOCK_DUMP=file
    // makes rtGetStats write them out)
    alignas(128) unsigned int block_counts[RTIOW_BLOCK_COUNTERS * 32];
#endif
};

// Device-side shading record of one sphere (32 B), built by rtSetScene from RtSphere + RtMaterial.
struct ShadeRec {
    float albedo[3]; // lambertian, metal; dielectric: 1 / ior, (1 - 1/ior) / (1 + 1/ior), (1 - ior) / (1 + ior)
    float param;     // fuzz (metal) or index of refraction (dielectric)
    float inv_r;     // 1 / radius, rounded once on the host
    uint32_t kind;   // RT_MAT_*
    uint32_t pad[2];
};

// Two-level sphere list of the clustered kernel (rtiow_clusters.cpp): the large spheres first, then
// clusters of kClusterSize members, kClusterStride slots apart.  With the lane-rotated member order of
// the kernel, a stride of 16 slots (256 B = one LDS bank row) makes the bank of a read depend on the
// lane only: conflict-free.
constexpr uint32_t kClusterSize = 16, kClusterStride = 16;
// Super-clusters (large scenes only): kSuperSize consecutive clusters -- one subtree of the build's splits -- under one box, from
// more than kSuperFrom clusters on.  (Rounds 1-3: 96, "the extra stage costs about as much as 80 box tests per ray".  With the
// splits rounded to whole super-clusters -- rtiow_clusters.cpp: before, a super box straddled two subtrees wherever the cluster
// count was not a power of two -- and the boxes tested without their flat axis, the level pays as soon as the scene is beyond
// the small-scene kernels (which are compiled without it and hold 32 clusters at most): cover scenes of 576 / 785 / 1026 spheres
// (40 / 56 / 64 clusters) 3.04 -> 2.8 / 3.40 -> 3.05 / 3.39 -> 2.98 ms at 32 spp, 44.8 -> 27 / 59.5 -> 30.5 / 62.9 -> 32.2 tests
// per segment.)
constexpr uint32_t kSuperSize = 8, kSuperFrom = 32;

struct PathArgs {
    const float4* spheres;       // n x {cx,cy,cz,radius} as uploaded (RtSphere)
    const ShadeRec* shade;       // n x 32 B
    uint32_t n;
    const float4* cslots;        // clustered: n_cslots x {cx,cy,cz,r*r}; padding never hit.  The first
                                 // n_large_slots hold the large spheres, then kClusterStride per cluster
    const uint32_t* cidx;        // clustered: original index of every slot (0xFFFFFFFF = padding)
    const float4* cbounds;       // clustered: n_clusters x {box centre, box half extent}, then the super-clusters' boxes; then
                                 // (flat_axis < 3) one float4 per box without the flat axis: {mid a, mid b, half a, half b}
    const float4* cbounds2;      // (flat_axis < 3) the boxes without the flat axis: n_clusters, then the super-clusters'
    uint32_t flat_axis;          // 0..2: every box spans the interval flat_mid +- flat_half along this axis (a union: conservative); 3: none
    float flat_mid, flat_half;
    uint32_t n_clusters;         // multiple of kSuperSize
    uint32_t n_super;            // super-clusters (0: none); their boxes follow the clusters' in cbounds
    uint32_t n_large;            // large spheres (tested exactly by every ray)
    uint32_t n_large_slots;      // their slots: n_large padded to a multiple of kClusterSize
    uint32_t n_cslots;           // n_large_slots + n_clusters * kClusterStride
    float ccenter[3];            // boxes are valid for ray origins with |o - ccenter|^2 <= crmax2
    float crmax2;
    float cfar_k, cfar_c;        // ... and, every half extent enlarged by cfar_k |o - ccenter| + cfar_c, for any origin (far_box_margin)
    RtCamera cam;
    uint32_t width, height;      // full image
    uint32_t spp, max_depth, seed, quantiser;
    uint32_t sample_offset;      // first sample index of this dispatch (progressive accumulation)
    unsigned long long* accum;   // progressive: local_pixels x 4 u64 {r,g,b,-} kept by the context, else null
    float inv_wm1, inv_hm1;      // 1/(width-1), 1/(height-1), rounded once on the host
    uint32_t row_block, tile_rank, tile_count;
    uint32_t local_rows;         // rows this call renders
    uint32_t* dst;               // local_rows x dst_stride words
    uint32_t dst_stride;         // in 32-bit words
    Counters* counters;
    Counters* next_counters;     // persistent kernels: the block the NEXT frame will use, zeroed by this frame's last workgroup
    // Cost-ordered dequeue (persistent kernels): the tile's pixels are handed out in chunks of 32; place s of the
    // chunk sequence holds the chunk chunk_order names for it (null: s itself; layout: launch_order_chunks).  Finished
    // pixels add the segments their samples took to chunk_cost[their chunk] (a wave that renders a whole chunk sums
    // in LDS and stores once); order_chunks() sorts the chunks by that for the next frame, dearest first, so that
    // the long paths of glass-heavy pixels start early and the frame ends on cheap ones.
    const uint32_t* chunk_order;
    unsigned long long* chunk_cost;  // null: not collected
    uint32_t order_stale;            // (host side only) chunk_order was made from another view: the camera has moved since
};

struct ChArgs {
    RtUbo5 ubo;
    uint32_t mode;
    uint32_t width, height;
    uint32_t* dst;
    uint32_t dst_stride;
    uint32_t tiles_form;     // != 0: one lane per pixel in 16x16 tiles, as the reference dispatches (RtParams.kernel = 1)
    uint32_t rows_per_wave;  // (set by launch_ch) rows of the frame a wave of ch_kernel_rows renders
    uint32_t vector_store;   // (set by launch_ch) destination rows are 16-byte aligned
    uint32_t row_blocks, row_block_stride;  // (set by launch_ch) two-phase kernel: grid row g renders row block g * stride mod row_blocks
};

// Chunks of the persistent kernels' pixel queue.  One chunk = 16 pixels = half a 128-byte line of the frame: all its
// stores come from the XCD whose queue holds it.  Also the unit of the cost order.  Rounds 2-4: 32 pixels (one eighth of the
// cover frame, tools/ab_bench.py: 256 pixels 1.71 ms, 128: 1.67, 64: 1.69, 32: 1.64).  Round 5: what ends a small frame is dear pixels
// that sit in chunks of middling cost and are therefore STARTED late (tools/wavelog.py), and a finer order starts them earlier: 16
// pixels take one eighth of the cover frame from 0.883 to 0.864 ms and a quarter from 1.61 to 1.57; the whole frame, a 16-spp frame and
// C5 (whole-chunk line stores of 64 bytes instead of 128) stay where they were; 8 pixels lose 5 % (profiles/r05_ab_log.txt).
#ifndef RTIOW_CHUNK_PIX
#define RTIOW_CHUNK_PIX 16
#endif
constexpr uint32_t kChunkPixels = RTIOW_CHUNK_PIX;
// LDS a wave of the persistent kernels has to itself (rtiow_kernels.hip: path_persistent_kernel lays it out, launch_path sizes it)
constexpr uint32_t kChunkPix = kChunkPixels;         // pixels per XCD-queue chunk
constexpr uint32_t kLineBufs = 4;                    // whole chunks a wave may be assembling: 32 RGBA8 pixels each +
constexpr uint32_t kLineMetaWords = 4;               // ... {pixels done, pixels expected, segments they took (u64)}
constexpr uint32_t kWaveLineBytes = kLineBufs * (kChunkPix + kLineMetaWords) * 4u;
constexpr uint32_t kAccEntries = 64;                 // pixels a wave may have in flight
constexpr uint32_t kAccEntriesCompact = 32, kLineBufsCompact = 2;  // ... and line buffers, in the compact per-wave area (large scenes at four waves per SIMD: rtiow_kernels.hip, COMPACT)
constexpr uint32_t kAccWords = 4;                    // u64 words per entry: r, g, b, samples done
constexpr uint32_t kWaveAccBytes = kAccEntries * kAccWords * 8u;  // 2 KiB of LDS per wave
constexpr uint32_t kWavePixBytes = kAccEntries * 4u;              // ... and the pixel of each entry
constexpr uint32_t kItemCap = 512;                   // (clustered) items per work list
constexpr uint32_t kWaveResultBytes = 128u * 8u;     // (clustered) one u64 key per path slot of the wave
__host__ __device__ constexpr uint32_t wave_item_bytes(bool two_level) { return kWaveResultBytes + kItemCap * 2u * (two_level ? 2u : 1u); }
constexpr size_t kLdsPerCu = 160u * 1024u - 192u;    // dynamic LDS a workgroup of the persistent kernels may ask for (they have no static __shared__)
constexpr uint32_t kGroupLdsBytes = 64u;             // ... of which the workgroup's own words: the leaving waves' sums and tick, one wave's clock stamps
// kernel variants selectable through RtParams.kernel (identical results)
struct ClusterF4 {
    float x, y, z, w;
};
struct ClusterScene {  // host-side result of build_clusters
    std::vector<ClusterF4> slots;
    std::vector<uint32_t> idx;
    std::vector<ClusterF4> bounds;  // two per cluster: centre, half extent; then two per super-cluster; then (flat_axis < 3)
                                    // one per cluster and super-cluster: the box without its flat axis {mid a, mid b, half a, half b}
    uint32_t flat_axis = 3;         // axis along which all cluster boxes span (nearly) one interval; 3: none
    float flat_mid = 0, flat_half = 0;  // that interval (the union of the boxes')
    uint32_t n_clusters = 0, n_super = 0;
    uint32_t n_large = 0, n_large_slots = 0;
    float center[3] = {0, 0, 0};  // of the clustered spheres
    float diag = 0;               // their extent
    float rmax2 = 0;              // (range_diags diag)^2: ray origins farther from the centre are outside
                                  // the rounding margin the boxes were inflated for
    float far_k = 0, far_c = 0;   // a ray that starts q from the centre may use the boxes enlarged by far_k q + far_c
    double range_diags = 0;       // the range the boxes were built for, in scene diagonals (what rmax2 came from)
};
// The range the boxes of a scene are inflated for, in scene diagonals from its centre, while the camera is within it (rtRender moves
// up a rung when it is not: range_for_camera): 2 for a scene with one level of boxes -- a ray from beyond costs its wave-wide test
// (65 instructions: cover frame 6.93 / 6.95 / 7.03 ms at 1.5 / 2 / 3) -- and 0.6 for a scene with super-clusters, where a ray from
// beyond costs next to nothing (its margin rides through the expansion stage) and the margins, which grow with the square of the
// range times the scene's extent, are what inflates the boxes: C5's scene 7.60 -> 6.96 ms at 64 spp, 53.0 -> 47.1 tests per segment
// (tools/range_ab.py; 0.5 diagonals are the scene's own bounding sphere, below that the camera rays' culls go).
constexpr double kRangeOneLevel = 2.0, kRangeTwoLevel = 0.6, kRangeFloor = 0.25;
// range_diags: ray origins up to this many scene diagonals from the scene's centre use the boxes (>= kRangeFloor); <= 0: the
// scene's own range -- kRangeTwoLevel if it gets super-clusters, kRangeOneLevel if not -- decided inside, before a box is made
void build_clusters(const RtSphere* spheres, uint32_t n, double range_diags, ClusterScene& out);
unsigned long long cluster_build_count();  // calls of build_clusters in this process so far (tests: a scene is boxed once)
// A slot's original sphere index as the kernels keep it in LDS: 16 bits (rtSetScene caps a scene at 6144 spheres; round 5 -- 32 bits before:
// 8 KB of C5's 87 KB of lists, twelve of a 6000-sphere scene's 128)
typedef unsigned short SlotIndex;
// Does the clustered list fit the LDS of a CU beside the per-wave areas of one 256-thread group: 2 = with its super-cluster
// level, 1 = without it (launch_path then drops the level), 0 = not at all (flat list).  Shared by launch_path and rtSetScene.
inline int clustered_levels_that_fit(uint32_t n_cslots, uint32_t n_clusters, uint32_t n_super, bool flat) {
    const uint32_t box_bytes = flat ? 16u : 32u;
    auto fits = [&](uint32_t supers) {
        return static_cast<size_t>(n_cslots) * (16u + sizeof(SlotIndex)) + static_cast<size_t>(n_clusters + supers) * box_bytes +
                   kGroupLdsBytes + 4u * (kWaveAccBytes + kWavePixBytes + kWaveLineBytes + wave_item_bytes(supers != 0u)) <= kLdsPerCu;
    };
    return fits(n_super) ? 2 : (fits(0u) ? 1 : 0);
}
// The lists rtSetScene uploads: built for the scene's own range -- and, when its super-cluster level will not fit the LDS (a band of
// a few thousand spheres just below the flat-list fallback), for the one-level range instead, since launch_path will trace one level.
void build_scene_clusters(const RtSphere* spheres, uint32_t n, ClusterScene& out);

// PATH mode: the persistent kernels count a path's segments in 19 bits of its slot's bookkeeping word (rtiow_kernels.hip, Slot)
constexpr uint32_t kMaxPathDepth = (1u << 19) - 1u;

enum : uint32_t {
    KERNEL_DEFAULT = 0,
    KERNEL_PIXEL = 1,      // one lane per pixel, spp loop inside (v1)
    KERNEL_PERSISTENT = 2, // persistent waves, flat sphere list (every ray tests every sphere)
    KERNEL_CLUSTERED = 3,  // persistent waves, two-level list: cluster boxes, then members per lane
    KERNEL_CLUSTERED_PASS = 4  // the same with the primary pass (camera rays traced where they are made) whatever the
                               // samples per pixel; 0 and 3 use it from 8 samples per pixel on (large scenes: always).
                               // Reported as 3.
};

// Is every camera ray's length within [2^-30, 2^40]?  rtRender's precondition since late round 4: the kernels normalise a camera ray
// by the short square root and reciprocal (newton_sqrt, newton_rcp: exact on [2^-96, 2^128) and [2^-64, 2^64), rtiow_kernels.hip), as
// they do the rays they scatter themselves.  A ray runs from the lens to a point A + u H + v V of the image plane (A = lower_left -
// origin): no shorter than the plane's distance from the origin less the lens offset's component along the plane's normal, no longer than the farthest corner (u, v <= 2: u = (i +
// xi) / (W - 1)) plus the lens; the float evaluation is off by a few ulps of the LONGEST term, which comes off the shortest ray.
// In double; false for a degenerate (H x V = 0) or non-finite camera.
inline bool camera_rays_moderate(const RtCamera& c) {
    double A[3], H[3], V[3], n[3];
    for (int k = 0; k < 3; ++k) {
        A[k] = double(c.lower_left[k]) - c.origin[k];
        H[k] = c.horizontal[k];
        V[k] = c.vertical[k];
    }
    n[0] = H[1] * V[2] - H[2] * V[1];
    n[1] = H[2] * V[0] - H[0] * V[2];
    n[2] = H[0] * V[1] - H[1] * V[0];
    auto norm = [](const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    const double nn = norm(n);
    if (!(nn > 0.0) || !std::isfinite(nn)) return false;
    const double uvec[3] = {c.u[0], c.u[1], c.u[2]}, vvec[3] = {c.v[0], c.v[1], c.v[2]};
    auto along_n = [&](const double* v) { return std::fabs(v[0] * n[0] + v[1] * n[1] + v[2] * n[2]) / nn; };
    const double lens = c.lens_radius > 0.0f ? double(c.lens_radius) * (norm(uvec) + norm(vvec)) : 0.0;
    // (the lens offset x u + y v, |x|, |y| <= lens_radius, shortens a ray only by its component along the plane's normal: none for the
    // usual camera, whose u and v lie in the plane -- a lens wider than the focus distance is fine)
    const double lens_n = c.lens_radius > 0.0f ? double(c.lens_radius) * (along_n(uvec) + along_n(vvec)) : 0.0;
    const double nearest = along_n(A) - lens_n;
    double farthest = 0.0, term = norm(A) + 2.0 * norm(H) + 2.0 * norm(V) + lens;
    for (int k = 0; k < 3; ++k) term = std::max(term, std::fabs(double(c.origin[k])) + std::fabs(double(c.lower_left[k])));
    for (int cu = 0; cu < 2; ++cu)
        for (int cv = 0; cv < 2; ++cv) {
            double P[3];
            for (int k = 0; k < 3; ++k) P[k] = A[k] + 2.0 * cu * H[k] + 2.0 * cv * V[k];
            farthest = std::max(farthest, norm(P) + lens);
        }
    // (each component of the ray is four rounded operations on terms no larger than `term`, u and v two more: off by less than 2^-21
    // term in length.  A camera a million units from the origin can still focus four units ahead.)
    return std::isfinite(term) && farthest <= 0x1p40 && term <= 0x1p40 && nearest - term * 0x1p-20 >= 0x1p-30;
}

hipError_t launch_ch(const ChArgs& a, hipStream_t stream);
// rtSelfTestChSkySteps: every float in [lo, hi] where the sky colour of raytrace06.comp:45-47 changes (see ch_sky_steps_kernel)
// rtSelfTestUnaryScan: floats of [lo, hi] (positive) on which the kernels' six-instruction square root differs from sqrtf (fn 0), their
// three-instruction reciprocal from 1.0f / x (fn 1)
hipError_t launch_unary_scan(uint32_t fn, float lo, float hi, unsigned long long* bad, uint32_t* first, uint32_t cap, hipStream_t stream);
hipError_t launch_ch_sky_steps(float lo, float hi, RtChSkyStep* out, uint32_t cap, uint32_t* count, hipStream_t stream);
// the primary pass's cone cull run on the host (rtConeSelfTestHost): see rtiow_kernels.hip
int cone_selftest_host(const RtCamera& cam, uint32_t width, uint32_t height, uint32_t pix_lo, uint32_t pix_hi,
                       const float* range_center, float range_rmax, const RtSphere* spheres, uint32_t n_spheres,
                       const float* boxes, uint32_t n_boxes, uint8_t* sphere_reach, uint8_t* box_reach);
// `resolved` receives the variant that was launched (KERNEL_DEFAULT resolves to one of the others)
hipError_t launch_path(const PathArgs& a, uint32_t kernel, uint32_t max_take, int num_cus,
                       hipStream_t stream, uint32_t* resolved);
inline uint32_t chunk_count(uint32_t local_rows, uint32_t width) {
    return static_cast<uint32_t>((static_cast<unsigned long long>(local_rows) * width + kChunkPixels - 1u) / kChunkPixels);
}
// Words of the chunk order: it is stored queue by queue, ceil(n / 8) places for each of the eight queues, so the last
// place of queue 7 lies at 8 * ceil(n / 8) - 1 whatever n % 8 is.
__host__ __device__ inline uint32_t chunk_order_words(uint32_t n_chunks) { return 8u * ((n_chunks + 7u) / 8u); }
__host__ __device__ inline uint32_t chunk_order_slot(uint32_t seq, uint32_t n_chunks) { return (seq & 7u) * ((n_chunks + 7u) / 8u) + (seq >> 3); }
// order = the chunks by descending cost (ties in no particular order), place s of that sequence stored at
// (s % 8) * ceil(n / 8) + s / 8 (queue by queue: queue s % 8 takes it as its (s / 8)-th chunk); cost[] is zeroed for the
// next frame
hipError_t launch_order_chunks(unsigned long long* cost, uint32_t* order, uint32_t n_chunks, uint32_t spp, hipStream_t stream);
hipError_t launch_arith(uint32_t op, const float* a, const float* b, const float* c, float* out,
                        uint32_t n, hipStream_t stream);

}  // namespace rtiow
