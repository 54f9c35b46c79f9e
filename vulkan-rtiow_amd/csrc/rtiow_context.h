// rtiow_context.h — the state behind an RtContext handle.  Internal: shared by the C-ABI layer (rtiow_capi.hip)
// and the multi-GPU layer on top of it (rtiow_multi.hip); not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rtiow.h"
#include "rtiow_device.h"

// The (size, tile) a context-held per-frame state belongs to: the progressive accumulators and the chunk order are
// kept for exactly one of these and start afresh when any field changes (compared field by field: a packed key
// would alias, e.g. rank 1 of 2 with rank 0 of 66).
struct RtFrameShape {
    uint32_t width = 0, height = 0, row_block = 0, tile_rank = 0, tile_count = 0, n_chunks = 0;
    bool operator==(const RtFrameShape& o) const {
        return width == o.width && height == o.height && row_block == o.row_block && tile_rank == o.tile_rank &&
               tile_count == o.tile_count && n_chunks == o.n_chunks;
    }
    bool operator!=(const RtFrameShape& o) const { return !(*this == o); }
};

struct RtContext {
    int device = -1;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_done = nullptr;  // end of everything the last render enqueued (its stream may differ from the next one's)
    bool have_done = false;
    hipStream_t done_stream = nullptr;  // the stream ev_done was recorded on
    float4* d_spheres = nullptr;
    rtiow::ShadeRec* d_shade = nullptr;
    float4* d_cslots = nullptr;   // clustered list (rtiow_clusters.cpp)
    uint32_t* d_cidx = nullptr;
    float4* d_cbounds = nullptr;
    size_t cslots_bytes = 0, cidx_bytes = 0, cbounds_bytes = 0;  // capacities (re-boxing reuses the buffers)
    uint32_t n_clusters = 0, n_super = 0, n_large = 0, n_large_slots = 0, n_cslots = 0;
    uint32_t flat_axis = 3;       // (rtiow_clusters.cpp: the axis all cluster boxes share an interval along; 3: none)
    float flat_mid = 0, flat_half = 0;
    float cluster_center[3] = {0, 0, 0};
    float cluster_diag = 0, cluster_rmax2 = 0, cluster_far_k = 0, cluster_far_c = 0;
    uint32_t last_kernel = 0;     // variant the last PATH render launched
    std::vector<RtSphere> host_spheres;  // kept to re-box the clusters for a camera farther out
    double cluster_range = 0;     // range_diags the current boxes were built for
    double scene_base_range = 0;  // the scene's own range (kRangeOneLevel / kRangeTwoLevel), chosen by rtSetScene
    uint32_t cluster_uploads = 0; // times the lists of the current scene were built and uploaded (1 after rtSetScene; + re-boxes)
    double scene_build_ms = 0;    // host time of the last rtSetScene (shading records, cluster build, uploads)
    double last_cluster_build_ms = 0;  // ... and of the last cluster build + upload alone (rtSetScene or a re-box in rtRender)
    uint32_t n_spheres = 0;
    rtiow::Counters* d_counters = nullptr;  // TWO blocks: frame k counts in block k & 1 and its last workgroup zeroes the other
    uint32_t counter_index = 0;             // the block the next PATH frame uses ...
    bool counter_clean = false;             // ... which the previous frame's kernel has already zeroed
    uint32_t stats_index = 0;               // the block the last PATH frame counted in (read by rtGetStats)
    rtiow::Counters* h_counters = nullptr;  // pinned
    uint32_t* d_frame = nullptr;            // staging framebuffer for host destinations
    size_t frame_bytes = 0;
    unsigned long long* d_accum = nullptr;  // progressive accumulation: 4 x u64 per pixel of the tile
    size_t accum_bytes = 0;
    // cost-ordered dequeue: per-chunk cost of the last frame of this shape and the chunk order made from it
    unsigned long long* d_chunk_cost = nullptr;
    size_t chunk_cost_bytes = 0;
    uint32_t* d_chunk_order = nullptr;
    size_t chunk_order_bytes = 0;
    RtFrameShape order_shape;               // (size, tile) the order belongs to
    bool order_valid = false;
    uint32_t order_frames = 0;              // frames of this shape rendered so far
    RtCamera order_cam{};                   // the camera of the last of them (a camera that moves has every frame report its costs)
    RtFrameShape accum_shape;               // which frame the accumulators belong to
    uint32_t accum_samples = 0;             // samples accumulated so far
    bool have_timing = false;
    bool last_is_ch = false;      // the last render ran a CH05/CH06 kernel (no counters)
    hipStream_t last_stream = nullptr;
    RtStats stats{};
    std::string error;
};

