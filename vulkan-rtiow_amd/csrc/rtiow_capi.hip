// rtiow_capi.hip — the thin C-ABI layer over the gfx950 kernels: context
// lifetime, scene upload, the render dispatch and its statistics.  It stands
// where RTCHAP06/main.cpp:100-157 (resource setup) and :313-325 (per-frame
// compute submit) stand in the reference.  There is no CPU fallback: without a
// usable GPU every device entry point fails with RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <cmath>
#include <algorithm>
#include <chrono>

#include "../../include/rtiow.h"
#include "rtiow_device.h"

#include "rtiow_context.h"

namespace {

std::mutex g_err_mutex;
std::string g_error;  // errors with no context to hang them on

int fail(RtContext* ctx, int code, const std::string& msg) {
    if (ctx) {
        ctx->error = msg;
    } else {
        std::lock_guard<std::mutex> lock(g_err_mutex);
        g_error = msg;
    }
    return code;
}

int fail_hip(RtContext* ctx, hipError_t e, const char* what) {
    return fail(ctx, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define RT_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)

int ensure_bytes(RtContext* ctx, void** ptr, size_t* have, size_t need) {
    if (*have >= need) return RT_OK;
    if (*ptr) {
        RT_HIP(ctx, hipFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    hipError_t e = hipMalloc(ptr, need);
    if (e != hipSuccess) return fail(ctx, RT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    *have = need;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rtGetLastError(const RtContext* ctx) {
    if (ctx) return ctx->error.c_str();
    std::lock_guard<std::mutex> lock(g_err_mutex);
    static thread_local std::string copy;
    copy = g_error;
    return copy.c_str();
}

int rtCreate(int device_id, RtContext** out_ctx) {
    if (!out_ctx) return fail(nullptr, RT_ERR_INVALID, "rtCreate: out_ctx is null");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, RT_ERR_NO_DEVICE,
                    std::string("rtCreate: no HIP device (") + hipGetErrorString(e) + ")");
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, RT_ERR_NO_DEVICE, "rtCreate: device_id out of range");
    RtContext* ctx = new (std::nothrow) RtContext;
    if (!ctx) return fail(nullptr, RT_ERR_NOMEM, "rtCreate: out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&ctx->d_counters), 2 * sizeof(rtiow::Counters))) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_counters), sizeof(rtiow::Counters))) != hipSuccess) {
        int rc = fail_hip(nullptr, e, "rtCreate");
        rtDestroy(ctx);
        return rc;
    }
    ctx->num_cus = prop.multiProcessorCount;
    std::memset(ctx->h_counters, 0, sizeof(rtiow::Counters));
    *out_ctx = ctx;
    return RT_OK;
}

int rtDestroy(RtContext* ctx) {
    if (!ctx) return RT_OK;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_spheres) (void)hipFree(ctx->d_spheres);
    if (ctx->d_shade) (void)hipFree(ctx->d_shade);
    if (ctx->d_cslots) (void)hipFree(ctx->d_cslots);
    if (ctx->d_cidx) (void)hipFree(ctx->d_cidx);
    if (ctx->d_cbounds) (void)hipFree(ctx->d_cbounds);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    if (ctx->d_chunk_cost) (void)hipFree(ctx->d_chunk_cost);
    if (ctx->d_chunk_order) (void)hipFree(ctx->d_chunk_order);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

// The range the boxes are built for when the camera (with its lens) is `need` scene diagonals from the scene's centre: the
// scene's own (`base`: kRangeOneLevel, or kRangeTwoLevel for a scene that is traced with its super-cluster level), else the first of 1 and 2 diagonals that holds the camera with a tenth to spare,
// else twice its distance.
static double range_for_camera(double need, double base) {
    for (double r : {base, 1.0, 2.0})
        if (r >= base && need <= 0.9 * r) return r;
    return 2.0 * need;
}

// Builds the two-level list of the clustered kernel -- for ray origins up to range_diags scene diagonals from the scene's centre, or
// (range_diags <= 0: rtSetScene) for the scene's own range, ONE build (rtiow::build_scene_clusters) -- and uploads it (the context's
// stream must be idle).  On failure the context holds no clustered list at all (rtRender then falls back to nothing stale).
static int upload_clusters(RtContext* ctx, double range_diags) {
    const auto t0 = std::chrono::steady_clock::now();
    rtiow::ClusterScene cs;
    const uint32_t n = static_cast<uint32_t>(ctx->host_spheres.size());
    if (range_diags > 0.0) rtiow::build_clusters(ctx->host_spheres.data(), n, range_diags, cs);
    else rtiow::build_scene_clusters(ctx->host_spheres.data(), n, cs);
    ctx->n_clusters = ctx->n_super = ctx->n_large = ctx->n_large_slots = ctx->n_cslots = 0u;  // (until the new lists are in place)
    ++ctx->cluster_uploads;
    // (Re-)allocate only when the lists have grown -- re-boxing the same spheres for another range never does: hipFree
    // synchronises the whole device, and other contexts may have frames in flight on it.  The copies go through the
    // context's own stream (the caller has made sure no frame of this context still reads the old lists).
    auto room = [&](void** p, size_t* cap, size_t bytes) -> int {
        if (*cap >= bytes && *p) return RT_OK;
        if (*p) RT_HIP(ctx, hipFree(*p));
        *p = nullptr;
        *cap = 0;
        RT_HIP(ctx, hipMalloc(p, bytes));
        *cap = bytes;
        return RT_OK;
    };
    const size_t n_boxes = size_t(cs.n_clusters) + cs.n_super;
    int rc = room(reinterpret_cast<void**>(&ctx->d_cslots), &ctx->cslots_bytes, sizeof(float4) * cs.slots.size());
    if (rc == RT_OK) rc = room(reinterpret_cast<void**>(&ctx->d_cidx), &ctx->cidx_bytes, sizeof(uint32_t) * cs.idx.size());
    if (rc == RT_OK) rc = room(reinterpret_cast<void**>(&ctx->d_cbounds), &ctx->cbounds_bytes, sizeof(float4) * 3u * n_boxes);  // (whole boxes + flat ones)
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_cslots, cs.slots.data(), sizeof(float4) * cs.slots.size(), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_cidx, cs.idx.data(), sizeof(uint32_t) * cs.idx.size(), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_cbounds, cs.bounds.data(), sizeof(float4) * cs.bounds.size(), hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (`cs` goes out of scope; and any stream may render next)
    ctx->n_clusters = cs.n_clusters;
    ctx->n_super = cs.n_super;
    ctx->n_large = cs.n_large;
    ctx->n_large_slots = cs.n_large_slots;
    ctx->n_cslots = static_cast<uint32_t>(cs.slots.size());
    ctx->flat_axis = cs.flat_axis;
    ctx->flat_mid = cs.flat_mid;
    ctx->flat_half = cs.flat_half;
    for (int k = 0; k < 3; ++k) ctx->cluster_center[k] = cs.center[k];
    ctx->cluster_diag = cs.diag;
    ctx->cluster_rmax2 = cs.rmax2;
    ctx->cluster_far_k = cs.far_k;
    ctx->cluster_far_c = cs.far_c;
    ctx->cluster_range = cs.range_diags;
    ctx->last_cluster_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rtSetScene(RtContext* ctx, const RtSphere* spheres, const RtMaterial* materials,
               uint32_t n_spheres) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtSetScene: ctx is null");
    if (!spheres || !materials || n_spheres == 0)
        return fail(ctx, RT_ERR_INVALID, "rtSetScene: empty scene or null arrays");
    // the sphere list (16 B per sphere; clustered: 20 B per slot) must fit the 160 KiB LDS of a
    // gfx950 CU beside the accumulator entries of at least 8 waves
    if (n_spheres > 6144) return fail(ctx, RT_ERR_INVALID, "rtSetScene: more than 6144 spheres");
    for (uint32_t i = 0; i < n_spheres; ++i) {
        if (materials[i].kind > RT_MAT_DIELECTRIC)
            return fail(ctx, RT_ERR_INVALID, "rtSetScene: unknown material kind");
        if (!(spheres[i].radius != 0.0f))
            return fail(ctx, RT_ERR_INVALID, "rtSetScene: zero or NaN radius");
    }
    const auto t_scene = std::chrono::steady_clock::now();
    RT_HIP(ctx, hipSetDevice(ctx->device));
    // (no frame of this context may still read the old scene: its last render may have gone to a caller's stream)
    if (ctx->have_done) RT_HIP(ctx, hipEventSynchronize(ctx->ev_done));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_spheres) RT_HIP(ctx, hipFree(ctx->d_spheres));
    if (ctx->d_shade) RT_HIP(ctx, hipFree(ctx->d_shade));
    ctx->d_spheres = nullptr;
    ctx->d_shade = nullptr;
    ctx->n_spheres = 0;
    RT_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_spheres), sizeof(float4) * n_spheres));
    RT_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_shade), sizeof(rtiow::ShadeRec) * n_spheres));
    static_assert(sizeof(RtSphere) == sizeof(float4), "RtSphere must be 16 bytes");
    static_assert(sizeof(rtiow::ShadeRec) == 32, "ShadeRec must be 32 bytes");
    rtiow::ShadeRec* tmp = new (std::nothrow) rtiow::ShadeRec[n_spheres];
    if (!tmp) return fail(ctx, RT_ERR_NOMEM, "rtSetScene: out of host memory");
    for (uint32_t i = 0; i < n_spheres; ++i) {
        rtiow::ShadeRec& r = tmp[i];
        std::memset(&r, 0, sizeof r);
        const RtMaterial& m = materials[i];
        r.kind = m.kind;
        for (int k = 0; k < 3; ++k) r.albedo[k] = m.albedo[k];
        if (m.kind == RT_MAT_METAL) {  // fuzz clamped to [0,1] as the book's metal constructor does
            float fz = m.fuzz;
            if (!(fz < 1.0f)) fz = 1.0f;
            if (!(fz > 0.0f)) fz = 0.0f;
            r.param = fz;
        } else if (m.kind == RT_MAT_DIELECTRIC) {
            // glass has no albedo: its record carries the two quotients scatter() needs instead -- 1 / ior and Schlick's
            // (1 - ratio) / (1 + ratio) for either side -- by the oracle's own IEEE single operations (24 instructions
            // off the kernels' glass branch)
            r.param = m.ior;
            const float inv = 1.0f / m.ior;
            r.albedo[0] = inv;
            r.albedo[1] = (1.0f - inv) / (1.0f + inv);      // front face: ratio = 1 / ior
            r.albedo[2] = (1.0f - m.ior) / (1.0f + m.ior);  // back face: ratio = ior
        }
        r.inv_r = 1.0f / spheres[i].radius;  // IEEE single division, as the oracle's
    }
    hipError_t e = hipMemcpy(ctx->d_shade, tmp, sizeof(rtiow::ShadeRec) * n_spheres, hipMemcpyHostToDevice);
    delete[] tmp;
    if (e != hipSuccess) return fail_hip(ctx, e, "hipMemcpy(shading records)");
    RT_HIP(ctx, hipMemcpy(ctx->d_spheres, spheres, sizeof(float4) * n_spheres, hipMemcpyHostToDevice));
    ctx->host_spheres.assign(spheres, spheres + n_spheres);
    ctx->cluster_uploads = 0;
    int rc = upload_clusters(ctx, 0.0);  // the scene's own range (kRangeOneLevel / kRangeTwoLevel), decided before a box is made
    if (rc != RT_OK) return rc;
    ctx->scene_base_range = ctx->cluster_range;
    ctx->n_spheres = n_spheres;
    ctx->scene_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scene).count();
    return RT_OK;
}

// The cluster boxes are inflated for ray origins within cluster_range scene diagonals of the scene's centre
// (rtiow_clusters.cpp); the kernel sends any ray that starts farther out through every cluster (enlarged by its own margin
// where that is cheap).  A camera out there would lose the culls of its primary rays: the boxes are rebuilt for the next rung
// that holds it (range_for_camera: wider margins; rare, so this context's frames are simply drained first), and again once
// the camera is back well inside a lower rung.  Beyond 64 diagonals the margins swallow the boxes: flat list.
// Called at the top of rtRender, before anything of the frame is enqueued or recorded: the rebuild is host work (it is
// in RtSceneStats.last_cluster_build_ms, not in the frame's kernel_ms), and a failure leaves the frame sequence untouched.
static int boxes_for_camera(RtContext* ctx, const RtCamera* cam, uint32_t* kernel) {
    if (*kernel != rtiow::KERNEL_CLUSTERED && *kernel != rtiow::KERNEL_CLUSTERED_PASS && *kernel != rtiow::KERNEL_DEFAULT) return RT_OK;
    double d2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double d = double(cam->origin[k]) - double(ctx->cluster_center[k]);
        d2 += d * d;
    }
    const double need = (std::sqrt(d2) + double(cam->lens_radius)) / std::max(1e-30, double(ctx->cluster_diag));
    if (!(need <= 64.0)) {
        *kernel = rtiow::KERNEL_PERSISTENT;
        return RT_OK;
    }
    const double base = ctx->scene_base_range;  // (the range rtSetScene chose: build_scene_clusters)
    if (need > 0.95 * ctx->cluster_range || range_for_camera(2.0 * need, base) < ctx->cluster_range) {
        // (out of the range, or so far inside that a camera twice as far out would still fit the rung below)
        // (the old lists are read by this context's frames only: wait for the last of them -- ev_done marks the end
        // of everything the previous render enqueued -- not for the device, where other contexts' frames are in flight)
        if (ctx->have_done) RT_HIP(ctx, hipEventSynchronize(ctx->ev_done));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return upload_clusters(ctx, range_for_camera(need, base));
    }
    return RT_OK;
}

static int render_common(RtContext* ctx, bool is_ch, const RtUbo5* ubo, const RtCamera* cam,
                         const RtParams* prm, void* dst, size_t dst_pitch, int dst_is_device,
                         void* stream_handle) {
    const uint32_t W = prm->width, H = prm->height;
    if (W == 0 || H == 0) return fail(ctx, RT_ERR_INVALID, "rtRender: empty image");
    if (!dst) return fail(ctx, RT_ERR_INVALID, "rtRender: dst is null");
    if (dst_pitch < size_t(W) * 4 || (dst_pitch & 3u))
        return fail(ctx, RT_ERR_INVALID, "rtRender: dst_pitch must be >= 4*width and a multiple of 4");
    uint32_t tcount = prm->tile_count <= 1 ? 1u : prm->tile_count;
    uint32_t rblock = prm->row_block == 0 ? 1u : prm->row_block;
    if (tcount > 1 && prm->tile_rank >= tcount)
        return fail(ctx, RT_ERR_INVALID, "rtRender: tile_rank >= tile_count");
    const uint32_t rows = rtTileRowCount(H, rblock, prm->tile_rank, tcount);
    if (!is_ch && uint64_t(rows) * W >= (1ull << 29))  // (the kernels keep a pixel's number in 29 bits: rtiow_kernels.hip, kPixLineShift)
        return fail(ctx, RT_ERR_INVALID, "rtRender: more than 2^29 pixels in one tile");

    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_handle ? static_cast<hipStream_t>(stream_handle) : ctx->stream;
    uint32_t kernel = prm->kernel;
    if (!is_ch) {
        int rc = boxes_for_camera(ctx, cam, &kernel);
        if (rc != RT_OK) return rc;
    }

    uint32_t* out = static_cast<uint32_t*>(dst);
    uint32_t out_stride = static_cast<uint32_t>(dst_pitch / 4);
    if (!dst_is_device) {
        int rc = ensure_bytes(ctx, reinterpret_cast<void**>(&ctx->d_frame), &ctx->frame_bytes,
                              size_t(W) * 4 * (rows ? rows : 1));
        if (rc != RT_OK) return rc;
        out = ctx->d_frame;
        out_stride = W;
    }

    ctx->stats = RtStats{};
    ctx->stats.rows_rendered = rows;
    ctx->stats.bytes_written = uint64_t(rows) * W * 4;
    ctx->stats.n_spheres = is_ch ? 1u : ctx->n_spheres;
    ctx->have_timing = false;
    // One frame in flight per context: the counters (with the pixel queues' heads), the timing events, the
    // accumulators and the chunk order are the context's.  A render on another stream than the last one's
    // waits, on the device, for everything the last one enqueued.
    // (The reference's own kernels keep no state in the context -- no queues, no counters -- and their frame is
    // launch-bound: they neither wait nor leave an event behind.)
    if (!is_ch && ctx->have_done && stream != ctx->done_stream) RT_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_done, 0));
    ctx->last_stream = stream;
    // progressive accumulation: the accumulators belong to one (size, tile) frame; a new frame starts at
    // sample_offset 0.  (Checked and recorded before the empty-tile return: a rank that owns no rows still
    // follows the sequence of its peers.)
    const uint32_t n_chunks = is_ch ? 0u : rtiow::chunk_count(rows, W);
    const RtFrameShape shape{W, H, rblock, tcount > 1 ? prm->tile_rank : 0u, tcount, n_chunks};
    if (!is_ch && prm->accumulate) {
        if (prm->sample_offset != 0u && (shape != ctx->accum_shape || prm->sample_offset != ctx->accum_samples))
            return fail(ctx, RT_ERR_STATE, "rtRender: accumulate continues a different frame or sample count");
        ctx->accum_shape = shape;
        ctx->accum_samples = prm->sample_offset + prm->spp;
    }
    if (rows == 0) return RT_OK;

    // the reference's kernels keep no counters: their frame is launch-bound (7 us of kernel), so the
    // counter reset and read-back are left out of its dispatch (16 -> 9 us per frame, tools/ch_dispatch_rate.py)
    ctx->last_is_ch = is_ch;
    // PATH frames count in one of two counter blocks; a persistent kernel's last workgroup zeroes the other one for the
    // frame after it, so only the first frame (and one that follows the one-lane-per-pixel kernel) needs a memset
    rtiow::Counters* counters = ctx->d_counters + ctx->counter_index;
    if (!is_ch && !ctx->counter_clean) RT_HIP(ctx, hipMemsetAsync(counters, 0, sizeof(rtiow::Counters), stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev_start, stream));
    if (is_ch) {
        rtiow::ChArgs a{};
        a.ubo = *ubo;
        a.mode = prm->mode;
        a.width = W;
        a.height = H;
        a.dst = out;
        a.dst_stride = out_stride;
        a.tiles_form = prm->kernel == rtiow::KERNEL_PIXEL ? 1u : 0u;
        RT_HIP(ctx, rtiow::launch_ch(a, stream));
    } else {
        rtiow::PathArgs a{};
        a.spheres = ctx->d_spheres;
        a.shade = ctx->d_shade;
        a.cslots = ctx->d_cslots;
        a.cidx = ctx->d_cidx;
        a.cbounds = ctx->d_cbounds;
        a.n_clusters = ctx->n_clusters;
        a.n_super = ctx->n_super;
        a.n_large = ctx->n_large;
        a.n_large_slots = ctx->n_large_slots;
        a.n_cslots = ctx->n_cslots;
        a.flat_axis = ctx->flat_axis;
        a.cbounds2 = ctx->d_cbounds + 2u * size_t(ctx->n_clusters + ctx->n_super);
        a.flat_mid = ctx->flat_mid;
        a.flat_half = ctx->flat_half;
        for (int k = 0; k < 3; ++k) a.ccenter[k] = ctx->cluster_center[k];
        a.crmax2 = ctx->cluster_rmax2;
        a.cfar_k = ctx->cluster_far_k;
        a.cfar_c = ctx->cluster_far_c;
        a.n = ctx->n_spheres;
        a.cam = *cam;
        a.width = W;
        a.height = H;
        a.spp = prm->spp;
        a.max_depth = prm->max_depth;
        a.seed = prm->seed;
        a.quantiser = prm->quantiser;
        a.sample_offset = prm->accumulate ? prm->sample_offset : 0u;
        a.accum = nullptr;
        if (prm->accumulate) {
            int rc = ensure_bytes(ctx, reinterpret_cast<void**>(&ctx->d_accum), &ctx->accum_bytes,
                                  size_t(rows) * W * 4 * sizeof(unsigned long long));
            if (rc != RT_OK) return rc;
            a.accum = ctx->d_accum;
        }
        a.inv_wm1 = 1.0f / static_cast<float>(W - 1);
        a.inv_hm1 = 1.0f / static_cast<float>(H - 1);
        a.row_block = rblock;
        a.tile_rank = tcount > 1 ? prm->tile_rank : 0u;
        a.tile_count = tcount;
        a.local_rows = rows;
        a.dst = out;
        a.dst_stride = out_stride;
        a.counters = counters;
        a.next_counters = ctx->d_counters + (ctx->counter_index ^ 1u);
        // cost-ordered dequeue (persistent kernels): this frame is dealt in the order the last frame of the
        // same shape suggests, and leaves its own per-chunk costs for the next one
#ifdef RTIOW_NO_ORDER  // (tuning only)
        const bool ordered = false;
#else
        const bool ordered = kernel != rtiow::KERNEL_PIXEL && !rtiow::debug_knob("RTIOW_DEBUG_NO_ORDER");  // (tuning only)
#endif
        bool collect = false;
        if (ordered) {
            // (the order is stored queue by queue, ceil(n / 8) places per queue: rtiow::chunk_order_words)
            const size_t order_bytes = size_t(rtiow::chunk_order_words(n_chunks)) * 4u;
            const bool fresh = shape != ctx->order_shape || ctx->chunk_cost_bytes < size_t(n_chunks) * 8u ||
                               ctx->chunk_order_bytes < order_bytes;
            int rc = ensure_bytes(ctx, reinterpret_cast<void**>(&ctx->d_chunk_cost), &ctx->chunk_cost_bytes, size_t(n_chunks) * 8u);
            if (rc == RT_OK)
                rc = ensure_bytes(ctx, reinterpret_cast<void**>(&ctx->d_chunk_order), &ctx->chunk_order_bytes, order_bytes);
            if (rc != RT_OK) return rc;
            if (fresh) {
                RT_HIP(ctx, hipMemsetAsync(ctx->d_chunk_cost, 0, size_t(n_chunks) * 8u, stream));
                ctx->order_valid = false;
                ctx->order_shape = shape;
                ctx->order_frames = 0;
            }
            // The first two frames of a shape report their costs and have them sorted (the second one runs in the order
            // of the first), after that every eighth: a frame loop over a scene that changes slowly, if at all, does
            // not need a new order per frame, and seven frames in eight then pay neither the cost reports nor the
            // 32 us of sorting behind the path kernel.
            // ... unless the camera moves: while the view changes every frame reports, and the next one runs in ITS order -- one
            // view behind instead of up to eight.  Measured (tools/moving_ab.py, profiles/r04_moving_camera.txt: an orbit of the cover
            // scene, each frame against the same view standing still): +3.0 % at 3.75 degrees a frame and +2.2 % at 0.5 (every eighth
            // frame reporting: +3.4 / +2.9; a moved camera's frame in natural order instead: +2.1 / +3.3).
            const bool moved = std::memcmp(&ctx->order_cam, cam, sizeof(RtCamera)) != 0;
            ctx->order_cam = *cam;
            collect = ctx->order_frames < 2u || ctx->order_frames % 8u == 0u || moved;
            ++ctx->order_frames;
            a.chunk_cost = collect ? ctx->d_chunk_cost : nullptr;
            a.chunk_order = ctx->order_valid ? ctx->d_chunk_order : nullptr;
            a.order_stale = moved ? 1u : 0u;
        }
        RT_HIP(ctx, rtiow::launch_path(a, kernel, prm->chunk_spp, ctx->num_cus, stream, &ctx->last_kernel));
        RT_HIP(ctx, hipEventRecord(ctx->ev_stop, stream));
        ctx->stats_index = ctx->counter_index;
        ctx->counter_index ^= 1u;
        ctx->counter_clean = ctx->last_kernel != rtiow::KERNEL_PIXEL;  // (that kernel does not reset the other block)
        if (ordered && collect) {
            RT_HIP(ctx, rtiow::launch_order_chunks(ctx->d_chunk_cost, ctx->d_chunk_order, n_chunks, prm->spp, stream));
            ctx->order_valid = true;
        }
    }
    if (is_ch) RT_HIP(ctx, hipEventRecord(ctx->ev_stop, stream));
    ctx->have_timing = true;

    if (!dst_is_device) {
        RT_HIP(ctx, hipMemcpy2DAsync(dst, dst_pitch, ctx->d_frame, size_t(W) * 4, size_t(W) * 4,
                                     rows, hipMemcpyDeviceToHost, stream));
        RT_HIP(ctx, hipStreamSynchronize(stream));
    }
    if (!is_ch) {
        RT_HIP(ctx, hipEventRecord(ctx->ev_done, stream));
        ctx->have_done = true;
        ctx->done_stream = stream;
    }
    return RT_OK;
}

int rtRender(RtContext* ctx, const RtCamera* cam, const RtParams* params, void* dst,
             size_t dst_pitch, int dst_is_device, void* stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtRender: ctx is null");
    if (!params) return fail(ctx, RT_ERR_INVALID, "rtRender: params is null");
    if (params->mode == RT_MODE_CH05 || params->mode == RT_MODE_CH06) {
        // the reference's own kernels take their camera from the 5-float UBO
        // The image extent is the caller's integer size: main.cpp:106 truncates the UBO's float height back to
        // an integer, and 800/(800/608)-style quotients land just below the integer for ~5 % of sizes, which
        // would drop the last row.  The UBO floats are kept for the u/v arithmetic only.
        RtUbo5 ubo;
        if (rtUboFromImage(params->width, params->height, &ubo) != RT_OK || params->width > 65536u || params->height > 65536u)
            return fail(ctx, RT_ERR_INVALID, "rtRender: bad image size");
        RtParams p{};
        p.width = params->width;
        p.height = params->height;
        p.mode = params->mode;
        p.spp = 1;
        p.kernel = params->kernel == rtiow::KERNEL_PIXEL ? params->kernel : 0u;  // (1: one lane per pixel in 16x16 tiles, the reference's dispatch shape)
        return render_common(ctx, true, &ubo, nullptr, &p, dst, dst_pitch, dst_is_device, stream);
    }
    if (params->mode != RT_MODE_PATH) return fail(ctx, RT_ERR_INVALID, "rtRender: unknown mode");
    if (!cam) return fail(ctx, RT_ERR_INVALID, "rtRender: cam is null");
    if (ctx->n_spheres == 0) return fail(ctx, RT_ERR_STATE, "rtRender: PATH mode needs rtSetScene first");
    if (params->width < 2 || params->height < 2)
        return fail(ctx, RT_ERR_INVALID, "rtRender: PATH mode needs width,height >= 2");
    if (params->spp == 0 || params->spp > 65536)
        return fail(ctx, RT_ERR_INVALID, "rtRender: spp must be 1..65536");
    if (params->max_depth == 0)  // (a zero-initialised RtParams; the kernels' bounce loops differ on "no bounce at all")
        return fail(ctx, RT_ERR_INVALID, "rtRender: max_depth must be at least 1");
    if (params->max_depth > rtiow::kMaxPathDepth)
        return fail(ctx, RT_ERR_INVALID, "rtRender: max_depth must not exceed 524287");
    if (!rtiow::camera_rays_moderate(*cam))
        return fail(ctx, RT_ERR_INVALID, "rtRender: camera rays must be between 2^-30 and 2^40 long (and the image plane not degenerate)");
    if (params->accumulate && (params->sample_offset > 65536u - params->spp))
        return fail(ctx, RT_ERR_INVALID, "rtRender: sample_offset + spp must not exceed 65536");
    if (params->quantiser > RT_QUANT_BOOK) return fail(ctx, RT_ERR_INVALID, "rtRender: unknown quantiser");
    if (params->kernel > rtiow::KERNEL_CLUSTERED_PASS) return fail(ctx, RT_ERR_INVALID, "rtRender: unknown kernel variant (0..4)");
    return render_common(ctx, false, nullptr, cam, params, dst, dst_pitch, dst_is_device, stream);
}

int rtRenderUbo(RtContext* ctx, const RtUbo5* ubo, uint32_t mode, void* dst, size_t dst_pitch,
                int dst_is_device, void* stream) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtRenderUbo: ctx is null");
    if (!ubo) return fail(ctx, RT_ERR_INVALID, "rtRenderUbo: ubo is null");
    if (mode != RT_MODE_CH05 && mode != RT_MODE_CH06)
        return fail(ctx, RT_ERR_INVALID, "rtRenderUbo: mode must be RT_MODE_CH05 or RT_MODE_CH06");
    if (!(ubo->imageWidth >= 1.0f) || !(ubo->imageHeight >= 1.0f) || ubo->imageWidth > 65536.0f ||
        ubo->imageHeight > 65536.0f)
        return fail(ctx, RT_ERR_INVALID, "rtRenderUbo: image size out of range");
    RtParams p{};
    // main.cpp:106: the image extent is the UBO's float size truncated to uint32_t
    p.width = static_cast<uint32_t>(ubo->imageWidth);
    p.height = static_cast<uint32_t>(ubo->imageHeight);
    p.mode = mode;
    p.spp = 1;
    return render_common(ctx, true, ubo, nullptr, &p, dst, dst_pitch, dst_is_device, stream);
}

int rtSynchronize(RtContext* ctx) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtSynchronize: ctx is null");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->last_stream ? ctx->last_stream : ctx->stream));
    return RT_OK;
}

int rtGetStats(RtContext* ctx, RtStats* out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtGetStats: ctx is null");
    if (!out) return fail(ctx, RT_ERR_INVALID, "rtGetStats: out is null");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->have_timing) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->last_stream ? ctx->last_stream : ctx->stream));
        float ms = 0.0f;
        RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
        ctx->stats.kernel_ms = ms;
        if (ctx->last_is_ch) {
            std::memset(ctx->h_counters, 0, sizeof(rtiow::Counters));
        } else {  // (fetched here, not after every frame: one small copy less between two frames of a loop)
            RT_HIP(ctx, hipMemcpy(ctx->h_counters, ctx->d_counters + ctx->stats_index, sizeof(rtiow::Counters), hipMemcpyDeviceToHost));
        }
#ifdef RTIOW_BLOCK_COUNTERS
        if (const char* path = std::getenv("RTIOW_BLOCK_DUMP")) {  // (tools/blockprof builds only) one line per basic-block counter
            if (FILE* f = fopen(path, "w")) {
                // (entries, and the active lanes summed over them where the kernel was instrumented with `lanes`: a 64-bit sum eight bytes on)
                for (uint32_t b = 0; b < RTIOW_BLOCK_COUNTERS; ++b) {
                    const uint32_t* c = ctx->h_counters->block_counts + b * 32u;
                    fprintf(f, "%u %llu\n", c[0], static_cast<unsigned long long>(c[2]) | static_cast<unsigned long long>(c[3]) << 32);
                }
                fclose(f);
            }
        }
#endif
        ctx->stats.paths = ctx->h_counters->paths;
        ctx->stats.segments = ctx->h_counters->segments;
        // persistent kernels count the tests they perform; the one-lane-per-pixel kernel tests every
        // sphere for every segment
        ctx->stats.sphere_tests = ctx->h_counters->tests ? ctx->h_counters->tests
                                                         : ctx->stats.segments * ctx->stats.n_spheres;
        for (int k = 0; k < 8; ++k) ctx->stats.debug[k] = ctx->h_counters->debug[k];
        ctx->stats.shader_clock_mhz = ctx->h_counters->clk_ticks ? static_cast<uint32_t>(ctx->h_counters->clk_cycles * 100ull / ctx->h_counters->clk_ticks) : 0u;
#ifdef RTIOW_DEBUG_TIMELINE
        if (const char* path = rtiow::debug_knob("RTIOW_DEBUG_WAVELOG")) {  // one line per wave: see Counters::tl_wave
            if (FILE* f = fopen(path, "w")) {
                for (int wv = 0; wv < 8192; ++wv) {
                    const unsigned int* r = ctx->h_counters->tl_wave[wv];
                    if (r[2] == 0u) continue;
                    fprintf(f, "%d %u %u %u %u %u %u %u %u\n", wv, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
                }
                fclose(f);
            }
            static const char* held[5] = {"1-2", "3-4", "5-8", "9-16", "17-32"};
            for (int k = 0; k < 5; ++k) {
                const unsigned long long* b = ctx->h_counters->tl_bucket[k];
                if (b[0] != 0ull)
                    fprintf(stderr, "sparse iterations with %s paths: %llu, %.2f us each (trace %.0f + shade %.0f shader cycles)\n", held[k], b[0],
                            b[1] * 0.01 / b[0], double(b[2]) / b[0], double(b[3]) / b[0]);
            }
        }
        if (rtiow::debug_knob("RTIOW_DEBUG_HIST")) {
            const rtiow::Counters& c = *ctx->h_counters;
            static const char* names[3] = {"dry   ", "sparse", "done  "};
            for (int k = 0; k < 3; ++k) {
                fprintf(stderr, "waves %s (50 us bins):", names[k]);
                for (int b = 0; b < 64; ++b) fprintf(stderr, " %u", c.tl_hist[k][b]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "iterations after dry: max %u, sum %llu\nqueue 0 head at k/8 of its pixels (us):", c.tl_tail_iters_max, c.tl_tail_iters_sum);
            for (int k = 1; k <= 8; ++k) fprintf(stderr, " %.0f", c.tl_progress[k] * 0.01);
            fprintf(stderr, "\nentry-starved iterations: sum %llu, max per wave %u; live paths at dry: max %u, waves by 16s:", c.tl_starved_sum,
                    c.tl_starved_max, c.tl_live_at_dry_max);
            for (int b = 0; b < 9; ++b) fprintf(stderr, " %u", c.tl_live_at_dry_hist[b]);
            fprintf(stderr, "; of the waves dry > 100 us after the first:");
            for (int b = 0; b < 9; ++b) fprintf(stderr, " %u", c.tl_late_dry_live_hist[b]);
#ifdef RTIOW_DEBUG_DEPTH_HIST
            fprintf(stderr, "\npaths by segments taken (8, 9, ... 62, 63+):");
            for (int b = 8; b < 64; ++b) fprintf(stderr, " %llu", c.tl_depth_hist[b]);
#endif
            fprintf(stderr, "\nstart of the frame, per wave: entry -> scene staged, barrier passed %.2f us (max %.2f); then -> first pool fetched %.2f us -> first hand_out done %.2f us -> first camera rays made %.2f us (max %.2f; %llu waves)",
                    c.tl_start_sum[0] * 0.01 / 4096.0, c.tl_start_max[0] * 0.01, c.tl_start_sum[2] ? c.tl_start_sum[3] * 0.01 / c.tl_start_sum[2] : 0.0,
                    c.tl_start_sum[2] ? c.tl_start_sum[4] * 0.01 / c.tl_start_sum[2] : 0.0, c.tl_start_sum[2] ? c.tl_start_sum[1] * 0.01 / c.tl_start_sum[2] : 0.0, c.tl_start_max[1] * 0.01, c.tl_start_sum[2]);
            fprintf(stderr, "\nfrom the first sparse iteration on: %llu iterations, %.2f us each, %.1f paths each\n", c.tl_sparse_iters_sum,
                    c.tl_sparse_iters_sum ? c.tl_sparse_ticks_sum * 0.01 / c.tl_sparse_iters_sum : 0.0,
                    c.tl_sparse_iters_sum ? double(c.tl_sparse_paths_sum) / c.tl_sparse_iters_sum : 0.0);
        }
#endif
#ifdef RTIOW_DEBUG_COUNTERS
        if (rtiow::debug_knob("RTIOW_DEBUG_HIST")) {  // diagnostic builds: when waves ran dry / finished, 0.125 ms bins
            fprintf(stderr, "waves dry :");
            for (int k = 0; k < 32; ++k) fprintf(stderr, " %u", ctx->h_counters->hist_dry[k]);
            fprintf(stderr, "\nwaves done:");
            for (int k = 0; k < 32; ++k) fprintf(stderr, " %u", ctx->h_counters->hist_end[k]);
            const rtiow::Counters& c = *ctx->h_counters;
            fprintf(stderr, "\ntail: %llu iterations, %.2f us each; sparse among them %llu, trace %.2f us each, %.1f paths each\n",
                    c.tail_iters, c.tail_iters ? c.tail_ticks * 0.01 / c.tail_iters : 0.0, c.tail_sparse_iters,
                    c.tail_sparse_iters ? c.tail_sparse_ticks * 0.01 / c.tail_sparse_iters : 0.0,
                    c.tail_sparse_iters ? double(c.tail_sparse_paths) / c.tail_sparse_iters : 0.0);
            if (c.pass_stats[0])
                fprintf(stderr, "primary passes: %llu, %.1f camera rays each, %.2f cluster trips, %.1f paths go on, %.0f shader cycles (trace %.0f, shade %.0f)\n",
                        c.pass_stats[0], double(c.pass_stats[1]) / c.pass_stats[0], double(c.pass_stats[2]) / c.pass_stats[0],
                        double(c.pass_stats[3]) / c.pass_stats[0], double(c.pass_stats[4]) / c.pass_stats[0],
                        double(c.pass_stats[4] - c.pass_stats[5]) / c.pass_stats[0] , double(c.pass_stats[5]) / c.pass_stats[0]);
            if (c.pass_stats[0])
                fprintf(stderr, "   per pass: handing out %.0f cycles, camera %.0f; records -> slots %.0f cycles each, %.2f times per pass\n",
                        double(c.pass_stats[6]) / c.pass_stats[0], double(c.pass_stats[7]) / c.pass_stats[0],
                        c.pass_stats[9] ? double(c.pass_stats[8]) / c.pass_stats[9] : 0.0, double(c.pass_stats[9]) / c.pass_stats[0]);
            if (c.tail_iters)
                fprintf(stderr, "tail iteration, shader cycles: refill+merge %.0f trace %.0f shade %.0f\n",
                        double(c.tail_cyc[0]) / c.tail_iters, double(c.tail_cyc[1]) / c.tail_iters,
                        double(c.tail_cyc[2]) / c.tail_iters);
        }
#endif
    }
    *out = ctx->stats;
    return RT_OK;
}

int rtConeSelfTestHost(const RtCamera* cam, uint32_t width, uint32_t height, uint32_t pix_lo, uint32_t pix_hi,
                       const float* range_center, float range_rmax, const RtSphere* spheres, uint32_t n_spheres,
                       const float* boxes, uint32_t n_boxes, uint8_t* sphere_reach, uint8_t* box_reach) {
    if (!cam || !range_center || (n_spheres && (!spheres || !sphere_reach)) || (n_boxes && (!boxes || !box_reach)))
        return -fail(nullptr, RT_ERR_INVALID, "rtConeSelfTestHost: null argument");
    if (width < 2 || height < 2 || pix_lo > pix_hi || pix_hi / width != pix_lo / width || pix_hi / width >= height)
        return -fail(nullptr, RT_ERR_INVALID, "rtConeSelfTestHost: the span must lie in one row of the image");
    return rtiow::cone_selftest_host(*cam, width, height, pix_lo, pix_hi, range_center, range_rmax, spheres, n_spheres, boxes,
                                     n_boxes, sphere_reach, box_reach);
}

int rtClusterBuildHost(const RtSphere* spheres, uint32_t n_spheres, float range_diags, float* boxes, float* flat_boxes,
                       uint32_t box_cap, uint32_t* n_clusters, uint32_t* n_super, uint32_t* slot_index, uint32_t slot_cap,
                       uint32_t* n_slots, uint32_t* n_large_slots, uint32_t* flat_axis, float* flat_interval) {
    if (!spheres || n_spheres == 0 || !n_clusters || !n_super || !n_slots || !n_large_slots || !flat_axis || !flat_interval)
        return fail(nullptr, RT_ERR_INVALID, "rtClusterBuildHost: null argument or empty scene");
    rtiow::ClusterScene cs;
    rtiow::build_clusters(spheres, n_spheres, range_diags, cs);  // the very function rtSetScene calls
    *n_clusters = cs.n_clusters;
    *n_super = cs.n_super;
    *n_slots = static_cast<uint32_t>(cs.slots.size());
    *n_large_slots = cs.n_large_slots;
    *flat_axis = cs.flat_axis;
    flat_interval[0] = cs.flat_mid;
    flat_interval[1] = cs.flat_half;
    const uint32_t n_boxes = cs.n_clusters + cs.n_super;
    for (uint32_t b = 0; b < n_boxes && b < box_cap; ++b) {
        if (boxes) {
            const rtiow::ClusterF4 &mid = cs.bounds[2u * b], &half = cs.bounds[2u * b + 1u];
            const float v[6] = {mid.x, mid.y, mid.z, half.x, half.y, half.z};
            std::memcpy(boxes + 6u * b, v, sizeof v);
        }
        if (flat_boxes && cs.flat_axis < 3u) {
            const rtiow::ClusterF4& f = cs.bounds[2u * n_boxes + b];
            const float v[4] = {f.x, f.y, f.z, f.w};
            std::memcpy(flat_boxes + 4u * b, v, sizeof v);
        }
    }
    if (slot_index)
        for (uint32_t k = 0; k < cs.idx.size() && k < slot_cap; ++k) slot_index[k] = cs.idx[k];
    return RT_OK;
}

int rtCameraIsRenderable(const RtCamera* cam) {  // rtRender's precondition on the camera, for a caller to ask beforehand (no GPU needed)
    return cam != nullptr && rtiow::camera_rays_moderate(*cam) ? 1 : 0;
}

int rtSceneClusterSelfTestHost(const RtSphere* spheres, uint32_t n_spheres, uint32_t* builds_out, double* range_out,
                               uint32_t* n_super_out) {
    if (!spheres || n_spheres == 0 || !builds_out || !range_out || !n_super_out)
        return fail(nullptr, RT_ERR_INVALID, "rtSceneClusterSelfTestHost: null argument or empty scene");
    const unsigned long long before = rtiow::cluster_build_count();
    rtiow::ClusterScene cs;
    rtiow::build_scene_clusters(spheres, n_spheres, cs);  // the very function rtSetScene calls
    *builds_out = static_cast<uint32_t>(rtiow::cluster_build_count() - before);
    *range_out = cs.range_diags;
    *n_super_out = cs.n_super;
    return RT_OK;
}

int rtGetSceneStats(const RtContext* ctx, RtSceneStats* out) {
    if (!ctx || !out) return fail(nullptr, RT_ERR_INVALID, "rtGetSceneStats: null argument");
    *out = RtSceneStats{};
    out->scene_build_ms = ctx->scene_build_ms;
    out->last_cluster_build_ms = ctx->last_cluster_build_ms;
    out->range_diags = ctx->cluster_range;
    out->base_range_diags = ctx->scene_base_range;
    out->cluster_builds = ctx->cluster_uploads;
    out->n_clusters = ctx->n_clusters;
    out->n_super = ctx->n_super;
    out->n_large = ctx->n_large;
    out->flat_axis = ctx->flat_axis;
    out->n_spheres = ctx->n_spheres;
    return RT_OK;
}

int rtChunkOrderSelfTestHost(uint32_t n_chunks, uint32_t* words_out, uint32_t* max_slot_out) {
    if (!words_out || !max_slot_out) return fail(nullptr, RT_ERR_INVALID, "rtChunkOrderSelfTestHost: null argument");
    // the very functions rtRender sizes the buffer with and the kernels index it with
    *words_out = rtiow::chunk_order_words(n_chunks);
    uint32_t hi = 0;
    std::vector<uint8_t> seen(*words_out, 0);
    for (uint32_t s = 0; s < n_chunks; ++s) {
        const uint32_t slot = rtiow::chunk_order_slot(s, n_chunks);
        hi = slot > hi ? slot : hi;
        if (slot < *words_out) {
            if (seen[slot]) return fail(nullptr, RT_ERR_STATE, "rtChunkOrderSelfTestHost: two places share a slot");
            seen[slot] = 1;
        }
    }
    *max_slot_out = hi;
    return RT_OK;
}

int rtGetLastKernel(RtContext* ctx, uint32_t* kernel_out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtGetLastKernel: ctx is null");
    if (!kernel_out) return fail(ctx, RT_ERR_INVALID, "rtGetLastKernel: kernel_out is null");
    *kernel_out = ctx->last_kernel;
    return RT_OK;
}

// Arithmetic conformance probe: runs op over host arrays on the GPU
// (0 fma, 1 div, 2 sqrt, 3 mul, 4 add, 5 RNG draw, 6 fixed-point accumulate, 7 u64->float, 8 / 9 the kernels' lean square root / quotient, 10-16 the pieces of the two-phase CH pixels: include/rtiow.h).  Used by the parity tests to
// localise any CPU/GPU rounding difference to a single operation.
int rtSelfTestArith(RtContext* ctx, uint32_t op, const float* a, const float* b, const float* c,
                    float* out, uint32_t n) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtSelfTestArith: ctx is null");
    if (!a || !b || !c || !out || n == 0 || op > 18)
        return fail(ctx, RT_ERR_INVALID, "rtSelfTestArith: bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    float* d = nullptr;
    const size_t bytes = size_t(n) * sizeof(float);
    RT_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&d), bytes * 4));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * size_t(n), c, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rtiow::launch_arith(op, d, d + n, d + 2 * size_t(n), d + 3 * size_t(n), n, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d + 3 * size_t(n), bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, e, "rtSelfTestArith");
    return RT_OK;
}

int rtSelfTestChSkySteps(RtContext* ctx, float lo, float hi, RtChSkyStep* out, uint32_t cap, uint32_t* count) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtSelfTestChSkySteps: ctx is null");
    if (!out || !count || cap == 0 || !(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi))
        return fail(ctx, RT_ERR_INVALID, "rtSelfTestChSkySteps: bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    const size_t bytes = size_t(cap) * sizeof(RtChSkyStep);
    RT_HIP(ctx, hipMalloc(&d, bytes + 16));
    uint32_t* d_count = reinterpret_cast<uint32_t*>(static_cast<char*>(d) + bytes);
    hipError_t e = hipMemsetAsync(d_count, 0, 4, ctx->stream);
    if (e == hipSuccess) e = rtiow::launch_ch_sky_steps(lo, hi, static_cast<RtChSkyStep*>(d), cap, d_count, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(count, d_count, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out, d, size_t(*count < cap ? *count : cap) * sizeof(RtChSkyStep), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, e, "rtSelfTestChSkySteps");
    return RT_OK;
}

int rtSelfTestUnaryScan(RtContext* ctx, uint32_t fn, float lo, float hi, uint64_t* mismatches, uint32_t* first, uint32_t cap) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rtSelfTestUnaryScan: ctx is null");
    if (fn > 1u || !mismatches || (cap != 0 && !first) || !(lo <= hi) || !(lo >= 0.0f) || !std::isfinite(hi))
        return fail(ctx, RT_ERR_INVALID, "rtSelfTestUnaryScan: bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    void* d = nullptr;
    RT_HIP(ctx, hipMalloc(&d, 8 + size_t(cap) * 4));
    unsigned long long* d_bad = static_cast<unsigned long long*>(d);
    uint32_t* d_first = reinterpret_cast<uint32_t*>(d_bad + 1);
    hipError_t e = hipMemsetAsync(d, 0, 8 + size_t(cap) * 4, ctx->stream);
    if (e == hipSuccess) e = rtiow::launch_unary_scan(fn, lo, hi, d_bad, d_first, cap, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && cap != 0) e = hipMemcpy(first, d_first, size_t(cap) * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, e, "rtSelfTestUnaryScan");
    *mismatches = bad;
    return RT_OK;
}

}  // extern "C"
