/*
 * rtiow.h — C-ABI of the MI355X-native path tracer that replaces the compute
 * dispatch of baeng72/Vulkan-RTIOW.
 *
 * The reference has no plugin/FFI API; its de-facto boundary is the compute
 * dispatch block RTCHAP06/main.cpp:313-325 plus the two resources bound at
 * RTCHAP06/main.cpp:127-151 (binding 0: rgba8 storage image, binding 1: the
 * 20-byte camera UBO).  Every entry point below cites the reference interface
 * it stands in for.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions (mirroring Vulkan.h:111-133's caller-owned POD handles):
 *   - every function returns an int status (RT_OK == 0) and never aborts
 *     (the reference asserts: Vulkan.cpp:60-62; a C ABI cannot);
 *   - a context belongs to one GPU and one host thread at a time, and has ONE frame in flight: its
 *     pixel queues, counters, accumulators and chunk order are its own.  A render on another stream
 *     than the previous one's is ordered behind it on the device; to overlap frames -- the reference
 *     keeps one compute fence per swapchain image, RTCHAP06/main.cpp:94-98,313-316 -- use one context
 *     per frame in flight;
 *   - the framebuffer is W*H packed RGBA8 (bytes R,G,B,A; A == 0 exactly as
 *     imageStore(vec4(color,0.0)) writes it, raytrace06.comp:66), row 0 = the
 *     BOTTOM of the scene (raytrace06.comp:58; the display flips it in
 *     rt.frag:8, and so does rtWritePPM).
 */
#ifndef RTIOW_H
#define RTIOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTIOW_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------ */
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,   /* bad argument (null pointer, zero size, bad mode) */
    RT_ERR_NO_DEVICE = 2, /* no such GPU / HIP runtime unusable                */
    RT_ERR_HIP = 3,       /* a HIP call failed; see rtGetLastError            */
    RT_ERR_NOMEM = 4,
    RT_ERR_STATE = 5,     /* e.g. PATH render before rtSetScene               */
    RT_ERR_IO = 6
};

/* ---- render modes ------------------------------------------------------ */
enum {
    RT_MODE_CH05 = 5,  /* RTCHAP05/RTCHAP05/Shaders/raytrace05.comp:21-61: flat red sphere   */
    RT_MODE_CH06 = 6,  /* RTCHAP06/Shaders/raytrace06.comp:21-66: normal-shaded sphere       */
    RT_MODE_PATH = 13  /* sphere list + lambertian/metal/dielectric + spp accumulate
                          (BUILD-SPEC: not in the reference)          */
};

/* ---- quantisers (hard part 5) ---------------------- */
enum {
    RT_QUANT_UNORM8 = 0, /* clamp[0,1]*255 round-half-up: rgba8 imageStore, raytrace06.comp:3,66 */
    RT_QUANT_BOOK = 1    /* (int)(256*clamp(x,0,0.999)): the RTIOW book's write_color            */
};

/* ---- material kinds ---------------------------------------------------- */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2 };

/* The reference's camera UBO, bit-compatible (raytrace06.comp:4-10; host
 * mirror RTCHAP06/main.cpp:109-120): five packed floats, sizeof == 20. */
typedef struct RtUbo5 {
    float imageWidth;
    float imageHeight;
    float viewportWidth;
    float viewportHeight;
    float focalLength;
} RtUbo5;

/* One sphere of the hittable list (16 B).  radius may be negative (hollow
 * glass: the outward normal flips). */
typedef struct RtSphere {
    float cx, cy, cz;
    float radius;
} RtSphere;

/* One material per sphere (32 B). */
typedef struct RtMaterial {
    uint32_t kind;   /* RT_MAT_*                                   */
    float albedo[3]; /* lambertian, metal                          */
    float fuzz;      /* metal, clamped to [0,1] by rtSetScene      */
    float ior;       /* dielectric                                 */
    uint32_t pad[2];
} RtMaterial;

/* Positionable thin-lens camera in derived form.
 * Fill it with rtMakeCamera or rtCameraFromUbo. */
typedef struct RtCamera {
    float origin[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
} RtCamera;

/* Per-render parameters. */
typedef struct RtParams {
    uint32_t width;      /* full image width                                        */
    uint32_t height;     /* full image height                                       */
    uint32_t spp;        /* samples per pixel (PATH), 1..65536                      */
    uint32_t max_depth;  /* bounce limit (PATH): 1 .. 524287                        */
    uint32_t seed;       /* RNG seed (PATH)                                         */
    uint32_t mode;       /* RT_MODE_*                                               */
    uint32_t quantiser;  /* RT_QUANT_*                                              */
    uint32_t chunk_spp;  /* scheduling hint only: the most samples a lane takes from the
                            work queue at a time (0 = auto).  Never changes the image:
                            samples are accumulated in 32.32 fixed point, an integer
                            sum that is independent of order and grouping.             */
    /* Row tiling across GPUs: this call renders the rows r with
     * (r / row_block) % tile_count == tile_rank, packed in ascending order.
     * tile_count == 0 or 1 renders the whole image. */
    uint32_t row_block;
    uint32_t tile_rank;
    uint32_t tile_count;
    uint32_t kernel;     /* 0 = default (persistent waves; flat sphere list below 64 spheres, clustered
                            list from 64 on); 1 one lane per pixel; 2 persistent, flat list; 3 persistent,
                            clustered list; 4 = 3 with the primary pass (camera rays traced where they are
                            made, against the spheres their pixels' cones reach) at any spp -- 0 and 3 use
                            it from 8 samples per pixel on (scenes too large for their shading records to
                            sit in LDS: always); rtGetLastKernel reports 3.  RT_MODE_CH05 / RT_MODE_CH06 through
                            rtRender: 1 = one lane per pixel in 16x16 tiles, the shape raytrace06.comp:2 and
                            RTCHAP06/main.cpp:321 dispatch; anything else = four pixels of a row per lane
                            (rows kernel).  Frames are identical by contract. */
    /* Progressive accumulation, the frame loop of RTCHAP06/main.cpp:304-360 with a running
     * average: with accumulate != 0 this dispatch adds samples sample_offset .. sample_offset+spp-1
     * of every pixel to accumulators the context keeps (reset when sample_offset == 0) and writes
     * the average of all sample_offset+spp samples so far.  k dispatches of spp samples give, bit
     * for bit, the frame of one dispatch of k*spp samples.  sample_offset + spp <= 65536. */
    uint32_t sample_offset;
    uint32_t accumulate;
} RtParams;

typedef struct RtStats {
    double kernel_ms;        /* HIP-event time of the device work of the last render      */
    uint64_t paths;          /* camera paths started                                      */
    uint64_t segments;       /* ray segments traced                                       */
    uint64_t sphere_tests;   /* ray-sphere and ray-box tests executed, from the kernel's own counter:
                                segments * n_spheres for the flat list and the one-lane-per-pixel kernel,
                                far fewer for the clustered list (large spheres + cluster boxes +
                                members of the boxes a ray reaches; camera rays traced in the
                                primary pass: the cone tests of their pixels + the spheres the
                                cones reach)                                                */
    uint64_t bytes_written;  /* algorithmic framebuffer bytes of the last render          */
    uint32_t rows_rendered;
    uint32_t n_spheres;
    uint64_t debug[8];       /* kernel-internal counters of diagnostic builds (0 otherwise)   */
    uint32_t shader_clock_mhz; /* PATH, persistent kernels: the shader clock the frame's kernel ran at, measured by one of
                                  its waves (shader cycles per tick of the constant 100 MHz counter); 0 otherwise.  ABI 4. */
    uint32_t reserved;
} RtStats;

typedef struct RtContext RtContext;

/* ---- device lifetime: replaces initInstance/choosePhysicalDevice/initDevice
 *      (RTCHAP06/Vulkan.cpp:46-153) and their cleanup* twins ---------------- */
int rtCreate(int device_id, RtContext** out_ctx);
int rtDestroy(RtContext* ctx);
const char* rtGetLastError(const RtContext* ctx); /* ctx may be NULL: global error */
int rtAbiVersion(void);

/* ---- scene upload: BUILD-SPEC extension of the descriptor writes at
 *      RTCHAP06/main.cpp:140-151 (the reference binds no scene buffer) -----
 *      LIMIT: 1 <= n_spheres <= 6144, else RT_ERR_INVALID.  Every persistent kernel keeps the whole sphere list in the LDS of
 *      a CU (160 KB: 18 bytes per list slot + the cluster boxes + 4.4 - 6 KB of private area per wave), which is where north_star
 *      asks for it; 6144 spheres are 118 KB of list.  BASELINE's largest scene has 4099.  Inside the limit the speed has steps
 *      (profiles/r05_scene_size_sweep.txt): up to ~510 spheres the shading records share the LDS; up to ~4600 spheres sixteen
 *      waves run per CU around one copy of the list, with fewer and fewer waiting camera paths each; beyond that the list leaves
 *      room for fewer waves -- fourteen at 5185 spheres, ten at 6086, which takes 1.6x the time of a 4099-sphere scene. */
int rtSetScene(RtContext* ctx, const RtSphere* spheres, const RtMaterial* materials,
               uint32_t n_spheres);

/* ---- the dispatch: replaces vkCmdDispatch + vkQueueSubmit + fence wait
 *      (RTCHAP06/main.cpp:313-325).  dst receives rtTileRowCount(...) rows of
 *      dst_pitch bytes each (dst_pitch >= 4*width, multiple of 4).
 *      dst_is_device != 0: dst is a device pointer on ctx's GPU and the work
 *      is enqueued on `stream` (a hipStream_t, NULL = the context's stream)
 *      without a host sync; otherwise dst is host memory and the call returns
 *      when the frame is in it (fixing the unsynchronised compute->graphics
 *      hazard of main.cpp:324 vs :354).
 *      One exception to "enqueues and returns".  The cluster boxes are inflated for ray origins within a RANGE of the
 *      scene's centre, in scene diagonals (diagonal = extent of the small spheres): rtSetScene chooses the scene's own range --
 *      2 for a scene traced with one level of boxes, 0.6 for one traced with super-clusters (rtGetSceneStats reports it) -- and
 *      builds the lists once.  A camera (with its lens) beyond 0.95 of the current range has the boxes rebuilt inside this call
 *      for the first rung of {the scene's own range, 1, 2} that holds it with a tenth to spare, else for twice its distance; a
 *      camera so far inside that one twice as far out would still fit a lower rung has them rebuilt for that rung.  The rebuild
 *      happens before anything of the frame is enqueued: the host waits for THIS context's previous frame (other contexts'
 *      frames on the device are not touched), builds (RtSceneStats.last_cluster_build_ms; not part of the frame's kernel_ms)
 *      and re-uploads.  Beyond 64 diagonals the frame is rendered with the flat list.  Rare (a fly-away camera), never wrong.
 *      max_depth above 524287 is refused (the kernels count a path's segments in 19 bits), and so is a camera whose rays --
 *      from the lens to any point of its image plane -- could be shorter than 2^-30 or longer than 2^40, or whose image plane
 *      is degenerate (horizontal x vertical = 0): the kernels normalise a ray by a square root and a reciprocal that are
 *      exact inside that range (rtSelfTestUnaryScan). */
int rtRender(RtContext* ctx, const RtCamera* cam, const RtParams* params, void* dst,
             size_t dst_pitch, int dst_is_device, void* stream);

/* rtRender's precondition on the camera, for a caller to ask beforehand: 1 if every ray from the lens to a point of the image
 * plane is between 2^-30 and 2^40 long and the image plane is not degenerate, else 0 (also for NULL).  No context, no GPU. */
int rtCameraIsRenderable(const RtCamera* cam);

/* The reference's own call shape: 20-byte UBO in, image out
 * (RTCHAP06/main.cpp:109-124 + :313-325).  mode is RT_MODE_CH05/CH06.
 * The image is the shaders' byte for byte (raytrace05.comp / raytrace06.comp evaluated in IEEE float32 without contraction,
 * imageStore to rgba8 with alpha 0, row 0 at the bottom) for EVERY UBO; how it is computed depends on the UBO only in speed:
 * viewport and focal length within [2^-20, 2^20] in magnitude and image extents in [2, 2^24 + 1] take kernels whose roots
 * and quotients drop range handling that cannot trigger there, and whose pixels are computed from one-ulp approximations
 * wherever an error bound proves the stored byte the same (the exact arithmetic elsewhere); any other UBO takes the
 * compiler's full forms.  One-row or one-column images take one lane per pixel, as the reference dispatches. */
int rtRenderUbo(RtContext* ctx, const RtUbo5* ubo, uint32_t mode, void* dst, size_t dst_pitch,
                int dst_is_device, void* stream);

/* Waits for the context's outstanding work and fills stats of the last render. */
int rtGetStats(RtContext* ctx, RtStats* out);

/* What rtSetScene made of the scene (no device work; the reference uploads no scene, RTCHAP06/main.cpp:140-151). */
typedef struct RtSceneStats {
    double scene_build_ms;         /* host time of the last rtSetScene: shading records, cluster build, uploads       */
    double last_cluster_build_ms;  /* ... of the last cluster build + upload alone (rtSetScene, or a re-box in rtRender) */
    double range_diags;            /* range the boxes are inflated for now, in scene diagonals                          */
    double base_range_diags;       /* the scene's own range, chosen by rtSetScene (2: one level of boxes, 0.6: two)      */
    uint32_t cluster_builds;       /* times the lists of this scene were built and uploaded: 1 after rtSetScene, + 1 per re-box */
    uint32_t n_clusters, n_super, n_large;
    uint32_t flat_axis;            /* 0..2: the cluster boxes share an interval along this axis; 3: none                 */
    uint32_t n_spheres;
} RtSceneStats;
int rtGetSceneStats(const RtContext* ctx, RtSceneStats* out);
int rtSynchronize(RtContext* ctx);

/* Which kernel variant the last RT_MODE_PATH render ran (RtParams.kernel 0 lets the library choose):
 * 1 one lane per pixel, 2 persistent waves over the flat list, 3 persistent waves over the clustered
 * list.  All variants produce the same bytes.  Diagnostic; no reference counterpart (the reference
 * has one pipeline, RTCHAP06/main.cpp:153-157). */
int rtGetLastKernel(RtContext* ctx, uint32_t* kernel_out);

/* Arithmetic conformance probe (diagnostic): evaluates one operation per
 * element on the GPU over host arrays — op 0 fma(a,b,c), 1 a/b, 2 sqrt(a),
 * 3 a*b, 4 a+b, 5 RNG draw, 6 fixed-point accumulate, 7 u64->float, 8 / 9 the kernels' own shortened square root / quotient
 * (to be compared with 2 / 1 on operands inside their stated ranges) — so a CPU/GPU rounding difference can be pinned to
 * a single operation.  Ops 10-16: the pieces of the two-phase CH05/CH06 pixels (results are bit patterns, bit 31 set when
 * the exact second phase ran): 10 v_rsq_f32(a); 11 / 12 the sky colour of (dy = a, dot(dir, dir) = b), two-phase / exact;
 * 13 the sky colour of normalize(dir).y = a; 14 / 15 the normal colour of v = (a, b, c), two-phase / exact; 16 the sky
 * table's answer for a alone; 17 / 18 the kernels' six-instruction square root / three-instruction reciprocal (rtSelfTestUnaryScan).  No reference counterpart. */
int rtSelfTestArith(RtContext* ctx, uint32_t op, const float* a, const float* b, const float* c,
                    float* out, uint32_t n);

/* Every float unit_y in [lo, hi] at which the sky colour of raytrace06.comp:45-47 / raytrace05.comp:39-40 -- a function of
 * normalize(dir).y alone -- differs from the colour of the float before it, found by evaluating the kernels' own arithmetic
 * on every float of the range on the GPU (two billion for [-1, 1]; milliseconds).  The two-phase pixels of the CH kernels
 * rest on the table tools/gen_ch_sky_table.py derives from the same step list on the host; the tests compare the two.
 * `count` receives the number found, of which the first `cap` (in no particular order) are written.  Diagnostic. */
typedef struct RtChSkyStep {
    float unit_y;          /* first float with the new colour */
    uint32_t before, after; /* packed r | g << 8 | b << 16 */
} RtChSkyStep;
int rtSelfTestChSkySteps(RtContext* ctx, float lo, float hi, RtChSkyStep* out, uint32_t cap, uint32_t* count);

/* The kernels take square roots by v_rsq_f32 and one Newton step (six instructions; rtSelfTestArith op 17) and their one
 * reciprocal by v_rcp_f32 and one Newton step (three; op 18).  That these are sqrtf(x) and 1.0f / x is a property of the
 * GPU's two instructions, checkable because each function has ONE float argument: this call evaluates the short form
 * (fn 0: the square root, fn 1: the reciprocal) and the compiler's on every float of [lo, hi] (0 <= lo <= hi finite) on
 * the GPU and returns the number of floats on which they differ, the bit patterns of the first `cap` of them in `first`.
 * The tests run it over [2^-96, FLT_MAX] and [2^-64, 2^64]: zero.  Diagnostic. */
int rtSelfTestUnaryScan(RtContext* ctx, uint32_t fn, float lo, float hi, uint64_t* mismatches, uint32_t* first, uint32_t cap);

/* ---- several GPUs of one node, one process ---------------------------------------------------
 * The reference drives exactly one device (RTCHAP06/Vulkan.cpp:87-97; one queue family by assert,
 * :122); this is the extension BASELINE.json asks for.  device_ids[0] is the root.  Device g renders
 * the rows r with (r / row_block) % n == g into its own HBM (RtParams.row_block, 0 = 4); the finished
 * rows are gathered to the root over xGMI (one ncclGather per device in one RCCL group; librccl.so.1 is
 * opened here, not at library load) and one kernel there puts every row in its place.  The frame is the
 * single-GPU frame byte for byte.  n == 1 is rtRender itself (RTIOW_MULTI_TRANSPORT=rccl: through a one-rank
 * communicator instead, which exercises the RCCL bindings on a one-GPU box).  A list that names one device more than
 * once (a rehearsal of the n-tile path on fewer GPUs) or RTIOW_MULTI_TRANSPORT=peer moves the tiles with
 * hipMemcpyPeerAsync instead; there is no fallback when RCCL is asked for and fails. */
typedef struct RtMulti RtMulti;
int rtCreateMulti(const int* device_ids, int n_devices, RtMulti** out);
int rtDestroyMulti(RtMulti* m);
int rtMultiSetScene(RtMulti* m, const RtSphere* spheres, const RtMaterial* materials, uint32_t n_spheres);
/* dst: height rows of dst_pitch bytes, host memory (the call returns when the frame is in it) or device
 * memory on the root (enqueued on the root context's stream; rtMultiSynchronize waits).  params->tile_* must
 * be 0: the tiling is this call's.  RT_MODE_PATH only when n > 1. */
int rtMultiRender(RtMulti* m, const RtCamera* cam, const RtParams* params, void* dst, size_t dst_pitch,
                  int dst_is_device);
int rtMultiSynchronize(RtMulti* m);
/* Statistics of device number device_index (its tile); *frame_ms (may be NULL) = root-side time from the
 * first launch to the assembled frame. */
int rtMultiGetStats(RtMulti* m, int device_index, RtStats* out, double* frame_ms);
int rtMultiDeviceCount(const RtMulti* m);
const char* rtMultiTransport(const RtMulti* m); /* "single", "rccl" or "peer-copy" */
const char* rtMultiGetLastError(const RtMulti* m); /* m may be NULL: errors of rtCreateMulti */
/* CPU check of the partition / padded gather slots / de-interleave with memcpy standing in for the
 * communicator: cuts `full` (height x width words) into n_tiles tiles, gathers, reassembles into `out`. */
int rtMultiSelfTestHost(const uint32_t* full, uint32_t width, uint32_t height, uint32_t row_block,
                        uint32_t n_tiles, uint32_t* out);

/* CPU check of the primary pass's cull (the kernels' own functions compiled for the host): for the span of pixels
 * pix_lo..pix_hi (indices row * width + column, one row) of a width x height image, which of the n_spheres spheres and
 * of the n_boxes boxes (six floats each: centre, half extent) can a camera ray of the span reach?  range_center /
 * range_rmax: the ray origins the scene's boxes are valid for (rtSetScene: the scene's range around its centre, see rtRender).
 * sphere_reach / box_reach receive one byte each (1 = may be reached).  Returns 1, 0 when the cull is off for this
 * camera (everything reached), or a negative RT_ERR_* code.  The test: no ray of the span, sampled as the camera
 * code samples it, hits a sphere or enters a box marked 0. */
int rtConeSelfTestHost(const RtCamera* cam, uint32_t width, uint32_t height, uint32_t pix_lo, uint32_t pix_hi,
                       const float* range_center, float range_rmax, const RtSphere* spheres, uint32_t n_spheres,
                       const float* boxes, uint32_t n_boxes, uint8_t* sphere_reach, uint8_t* box_reach);

/* CPU view of the two-level list rtSetScene builds for the clustered kernels (no GPU involved): for `spheres`, the boxes
 * of the clusters and then of the super-clusters (six floats each: centre, half extent; at most box_cap of them are
 * written), the original sphere index of every slot of the list (0xFFFFFFFF = padding; the first *n_large_slots slots
 * hold the large spheres, then 16 per cluster), and the flat axis: *flat_axis = 0..2 when every cluster box spans
 * (nearly) the interval flat_interval[0] +- flat_interval[1] along that axis -- the kernels then test the boxes without
 * it -- or 3; flat_boxes receives four floats per box {centre a, centre b, half a, half b} of the other two axes.
 * The tests: every sphere lies in its cluster's box, every cluster box in its super-cluster's, the common interval
 * contains every box's.  No reference counterpart (the reference has one sphere, raytrace06.comp:39). */
int rtClusterBuildHost(const RtSphere* spheres, uint32_t n_spheres, float range_diags, float* boxes, float* flat_boxes,
                       uint32_t box_cap, uint32_t* n_clusters, uint32_t* n_super, uint32_t* slot_index, uint32_t slot_cap,
                       uint32_t* n_slots, uint32_t* n_large_slots, uint32_t* flat_axis, float* flat_interval);
/* CPU run of rtSetScene's own choice (no GPU involved): builds the lists exactly as rtSetScene does and reports how many
 * builds that took (*builds_out: 1, or 2 for a scene whose super-cluster level does not fit the LDS of a CU), the range it
 * chose (*range_out, scene diagonals) and the super-clusters of the result (*n_super_out). */
int rtSceneClusterSelfTestHost(const RtSphere* spheres, uint32_t n_spheres, uint32_t* builds_out, double* range_out,
                               uint32_t* n_super_out);

/* CPU check of the chunk-order layout (cost-ordered dequeue): for a tile of n_chunks 32-pixel chunks, the words
 * rtRender allocates for the order (*words_out) and the highest word the kernels index (*max_slot_out), computed with
 * the functions both sides use; fails if two places of the sequence share a word.  No reference counterpart (the
 * reference dispatches a fixed grid, RTCHAP06/main.cpp:321). */
int rtChunkOrderSelfTestHost(uint32_t n_chunks, uint32_t* words_out, uint32_t* max_slot_out);

/* ---- host-side helpers (no GPU needed) ---------------------------------- */

/* RTCHAP06/main.cpp:101-120: width/height -> the UBO the reference fills
 * (viewportWidth = 2, viewportHeight = 2/aspect, focalLength = 1). */
int rtUboFromImage(uint32_t width, uint32_t height, RtUbo5* out);
/* raytrace06.comp:53-56: origin 0, axis-aligned viewport, no lens. */
int rtCameraFromUbo(const RtUbo5* ubo, RtCamera* out);
/* SURVEY.md section 9.4 (book camera; vfov in degrees). */
int rtMakeCamera(const float lookfrom[3], const float lookat[3], const float vup[3],
                 float vfov_deg, float aspect, float aperture, float focus_dist, RtCamera* out);

/* Scene builders.  `capacity` is the room in both arrays; *out_n the count made.
 * rtMakeCoverScene: SURVEY.md section 9.6 random_scene on a (2*grid_half)^2 grid
 * (grid_half = 11 -> ~485 spheres, 32 -> ~4000) drawn from the shared counter RNG. */
int rtMakeCoverScene(uint32_t seed, int grid_half, RtSphere* spheres, RtMaterial* materials,
                     uint32_t capacity, uint32_t* out_n);
/* BASELINE config 2: ground + lambertian centre + dielectric left (+ optional
 * inner r=-0.4 bubble) + metal right. */
int rtMakeThreeSphereScene(int with_bubble, RtSphere* spheres, RtMaterial* materials,
                           uint32_t capacity, uint32_t* out_n);

/* Row tiling helpers (which rows a rank renders). */
uint32_t rtTileRowCount(uint32_t height, uint32_t row_block, uint32_t tile_rank,
                        uint32_t tile_count);
/* global row of local row `local_row` of this tile */
uint32_t rtTileGlobalRow(uint32_t local_row, uint32_t row_block, uint32_t tile_rank,
                         uint32_t tile_count);

/* Lossless writer replacing saveScreenCap (RTCHAP06/Vulkan.cpp:625-766): binary
 * PPM "P6", top scene row first, i.e. buffer row H-1 first (rt.frag:8's flip). */
int rtWritePPM(const char* path, const void* rgba8, uint32_t width, uint32_t height,
               size_t pitch);
/* The same picture as an 8-bit RGB PNG (zlib "stored" blocks: lossless, uncompressed), for viewers
 * that do not read PPM; same orientation.  Also stands where saveScreenCap's stbi_write_jpg
 * (RTCHAP06/Vulkan.cpp:743,761) stood. */
int rtWritePNG(const char* path, const void* rgba8, uint32_t width, uint32_t height,
               size_t pitch);

#ifdef __cplusplus
}
#endif
#endif /* RTIOW_H */
