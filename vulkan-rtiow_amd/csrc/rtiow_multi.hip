// rtiow_multi.hip — one frame on the N GPUs of one node from ONE process, behind the C ABI
// (the reference is single-device by assert, RTCHAP06/Vulkan.cpp:122).
//
// Device g renders the rows r with (r / row_block) % N == g (block-cyclic row tiles: sky rows are far cheaper
// than sphere-dense rows) into a buffer of its own HBM; nothing is exchanged while rendering.  The finished
// RGBA8 rows are gathered to the first device over xGMI -- one ncclGather per device in one RCCL group -- into
// a buffer [N][rows_max][width], and one kernel there puts every row in its place in the frame.  Pixels do not
// depend on the partition (RNG and accumulation keyed by the global pixel), so the frame is the single-GPU
// frame byte for byte.
//
// Transports (what moves the tiles to the root):
//   rccl       ncclGather, grouped; the default when the devices are distinct.  librccl.so.1 is opened when the
//              first multi-device context is made, so single-GPU users do not load it.  If it cannot be opened
//              or a call fails, rtCreateMulti / rtMultiRender FAIL: there is no silent fallback.
//   peer-copy  hipMemcpyPeerAsync per tile + events; chosen by RTIOW_MULTI_TRANSPORT=peer, and whenever the list
//              names one device more than once (a rehearsal of the N-tile path on fewer GPUs, which RCCL refuses).
//   single     one device: rtRender itself (RTIOW_MULTI_TRANSPORT=rccl runs the rccl transport with a communicator of
//              one rank instead: the bindings, the group call and the de-interleave on a one-GPU box).
//   host       plain memcpy between host buffers: rtMultiSelfTestHost, the CPU test of partition, padding and
//              de-interleave against a host-memory stand-in for the communicator.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtiow.h"
#include "rtiow_context.h"

namespace {

// ---- the few RCCL entry points used, bound at run time (rccl.h:236,260,339,745,919-930) -------------------
struct Rccl {
    using Comm = void*;
    void* lib = nullptr;
    int (*CommInitAll)(Comm*, int, const int*) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    // ncclGather(sendbuff, recvbuff, sendcount, datatype, root, comm, stream)
    int (*Gather)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    static constexpr int kUint32 = 3;  // ncclUint32 (rccl.h:462)

    bool open(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("cannot open librccl.so.1: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* n) { return dlsym(lib, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Gather = reinterpret_cast<decltype(Gather)>(sym("ncclGather"));
        if (!CommInitAll || !CommDestroy || !GetErrorString || !GroupStart || !GroupEnd || !Gather) {
            err = "librccl.so.1 lacks ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd";
            return false;
        }
        return true;
    }
};
Rccl g_rccl;

enum class Transport { kSingle, kRccl, kPeerCopy };

}  // namespace

struct RtMulti {
    std::vector<int> devices;
    std::vector<RtContext*> ctx;         // one per entry of `devices`
    std::vector<uint32_t*> d_tile;       // device g's rows (g >= 1); the root renders into its slot of d_gather
    std::vector<size_t> tile_words;
    std::vector<hipEvent_t> ev_tile;     // peer-copy: tile g has arrived on the root
    std::vector<Rccl::Comm> comms;
    uint32_t* d_gather = nullptr;        // on the root: [n][rows_max][width]
    size_t gather_words = 0;
    uint32_t* d_frame = nullptr;         // on the root: staging for a host destination
    size_t frame_words = 0;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // on the root's stream: first launch .. frame assembled
    Transport transport = Transport::kSingle;
    double last_frame_ms = 0.0;
    bool have_timing = false;
    unsigned long long frames_done = 0;
    std::string error;
};

namespace {

std::string g_multi_error;

int mfail(RtMulti* m, int code, const std::string& msg) {
    if (m) m->error = msg;
    else g_multi_error = msg;
    return code;
}

#define RTM_HIP(m, call)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return mfail(m, RT_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

// rows device g of n owns, and the most any of them owns (the equal slot size of the gather buffer)
uint32_t rows_max(uint32_t height, uint32_t row_block, uint32_t n) {
    uint32_t m = 0;
    for (uint32_t g = 0; g < n; ++g) {
        const uint32_t r = rtTileRowCount(height, row_block, g, n);
        m = r > m ? r : m;
    }
    return m;
}

// One workgroup per (rank, local row): the row goes to its place in the frame.  128-bit copies when the row
// is a whole number of them and both sides are aligned, else words.
__global__ __launch_bounds__(256) void deinterleave_kernel(const uint32_t* gathered, uint32_t* frame, uint32_t frame_stride,
                                                            uint32_t width, uint32_t height, uint32_t row_block, uint32_t n,
                                                            uint32_t slot_rows) {
    const uint32_t g = blockIdx.x / slot_rows, lr = blockIdx.x % slot_rows;
    const uint32_t row = ((lr / row_block) * n + g) * row_block + lr % row_block;  // == rtTileGlobalRow
    if (row >= height) return;  // padding rows of a short tile
    const uint32_t* src = gathered + (static_cast<size_t>(g) * slot_rows + lr) * width;
    uint32_t* dst = frame + static_cast<size_t>(row) * frame_stride;
    if ((width & 3u) == 0u && (frame_stride & 3u) == 0u && (reinterpret_cast<uintptr_t>(frame) & 15u) == 0u &&
        (reinterpret_cast<uintptr_t>(gathered) & 15u) == 0u) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (uint32_t i = threadIdx.x; i < width / 4u; i += blockDim.x) d4[i] = s4[i];
    } else {
        for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) dst[i] = src[i];
    }
}

// the same on host memory (rtMultiSelfTestHost)
void deinterleave_host(const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height, uint32_t row_block,
                       uint32_t n, uint32_t slot_rows) {
    for (uint32_t g = 0; g < n; ++g)
        for (uint32_t lr = 0; lr < slot_rows; ++lr) {
            const uint32_t row = rtTileGlobalRow(lr, row_block, g, n);
            if (row >= height || lr >= rtTileRowCount(height, row_block, g, n)) continue;
            std::memcpy(frame + static_cast<size_t>(row) * width,
                        gathered + (static_cast<size_t>(g) * slot_rows + lr) * width, size_t(width) * 4);
        }
}

int ensure_words(RtMulti* m, int device, uint32_t** ptr, size_t* have, size_t need) {
    if (*have >= need) return RT_OK;
    RTM_HIP(m, hipSetDevice(device));
    if (*ptr) RTM_HIP(m, hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(ptr), need * 4);
    if (e != hipSuccess) return mfail(m, RT_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    *have = need;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rtMultiGetLastError(const RtMulti* m) { return m ? m->error.c_str() : g_multi_error.c_str(); }

int rtMultiDeviceCount(const RtMulti* m) { return m ? static_cast<int>(m->devices.size()) : 0; }

const char* rtMultiTransport(const RtMulti* m) {
    if (!m) return "";
    return m->transport == Transport::kRccl ? "rccl" : m->transport == Transport::kPeerCopy ? "peer-copy" : "single";
}

int rtDestroyMulti(RtMulti* m) {
    if (!m) return RT_OK;
    for (size_t g = 0; g < m->ctx.size(); ++g)
        if (m->ctx[g]) (void)rtSynchronize(m->ctx[g]);
    for (Rccl::Comm c : m->comms)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (size_t g = 0; g < m->devices.size(); ++g) {
        if (g >= m->ctx.size() || !m->ctx[g]) continue;  // (a device rtCreate refused: nothing of ours lives there)
        (void)hipSetDevice(m->devices[g]);
        if (g < m->d_tile.size() && m->d_tile[g]) (void)hipFree(m->d_tile[g]);
        if (g < m->ev_tile.size() && m->ev_tile[g]) (void)hipEventDestroy(m->ev_tile[g]);
    }
    if (!m->ctx.empty() && m->ctx[0]) {
        (void)hipSetDevice(m->devices[0]);
        if (m->d_gather) (void)hipFree(m->d_gather);
        if (m->d_frame) (void)hipFree(m->d_frame);
        if (m->ev_t0) (void)hipEventDestroy(m->ev_t0);
        if (m->ev_t1) (void)hipEventDestroy(m->ev_t1);
    }
    for (RtContext* c : m->ctx)
        if (c) (void)rtDestroy(c);
    (void)hipGetLastError();  // a failed teardown step must not surface in somebody's next launch check
    delete m;
    return RT_OK;
}

int rtCreateMulti(const int* device_ids, int n_devices, RtMulti** out) {
    if (!out) return mfail(nullptr, RT_ERR_INVALID, "rtCreateMulti: out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return mfail(nullptr, RT_ERR_INVALID, "rtCreateMulti: 1..64 devices");
    RtMulti* m = new (std::nothrow) RtMulti;
    if (!m) return mfail(nullptr, RT_ERR_NOMEM, "rtCreateMulti: out of host memory");
    const size_t n = static_cast<size_t>(n_devices);
    m->devices.assign(device_ids, device_ids + n_devices);
    m->ctx.assign(n, nullptr);
    m->d_tile.assign(n, nullptr);
    m->tile_words.assign(n, 0);
    m->ev_tile.assign(n, nullptr);
    bool distinct = true;
    for (size_t a = 0; a < n; ++a)
        for (size_t b = a + 1; b < n; ++b) distinct = distinct && m->devices[a] != m->devices[b];
    for (size_t g = 0; g < n; ++g) {
        int rc = rtCreate(m->devices[g], &m->ctx[g]);
        if (rc != RT_OK) {
            const std::string why = rtGetLastError(nullptr);
            rtDestroyMulti(m);
            return mfail(nullptr, rc, "rtCreateMulti: " + why);
        }
    }
    hipError_t e = hipSetDevice(m->devices[0]);
    if (e == hipSuccess) e = hipEventCreate(&m->ev_t0);
    if (e == hipSuccess) e = hipEventCreate(&m->ev_t1);
    for (size_t g = 1; g < n && e == hipSuccess; ++g) {
        e = hipSetDevice(m->devices[g]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_tile[g], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        const std::string why = hipGetErrorString(e);
        rtDestroyMulti(m);
        return mfail(nullptr, RT_ERR_HIP, "rtCreateMulti: " + why);
    }
    const char* want = getenv("RTIOW_MULTI_TRANSPORT");
    // One device is rtRender itself -- unless RTIOW_MULTI_TRANSPORT=rccl asks for the whole N-device machinery with a
    // communicator of one rank: librccl is opened, ncclCommInitAll, the grouped ncclGather (in place: nothing moves)
    // and the de-interleave all run.  That is how the RCCL bindings are exercised on a one-GPU box.
    if (n == 1 && !(want && std::strcmp(want, "rccl") == 0)) {
        m->transport = Transport::kSingle;
    } else {
        const bool peer = !distinct || (want && std::strcmp(want, "peer") == 0);
        if (peer) {
            m->transport = Transport::kPeerCopy;
            for (size_t g = 1; g < n; ++g) {  // direct xGMI copies need peer access switched on (a no-op on one device)
                if (m->devices[g] == m->devices[0]) continue;
                (void)hipSetDevice(m->devices[0]);
                (void)hipDeviceEnablePeerAccess(m->devices[g], 0);
                (void)hipSetDevice(m->devices[g]);
                (void)hipDeviceEnablePeerAccess(m->devices[0], 0);
            }
            (void)hipGetLastError();  // "already enabled" is not an error here
        } else {
            std::string err;
            if (!g_rccl.open(err)) {
                rtDestroyMulti(m);
                return mfail(nullptr, RT_ERR_NO_DEVICE, "rtCreateMulti: " + err);
            }
            m->comms.assign(n, nullptr);
            const int rc = g_rccl.CommInitAll(m->comms.data(), n_devices, m->devices.data());
            if (rc != 0) {
                const std::string why = g_rccl.GetErrorString(rc);
                m->comms.clear();
                rtDestroyMulti(m);
                return mfail(nullptr, RT_ERR_HIP, "rtCreateMulti: ncclCommInitAll: " + why);
            }
            m->transport = Transport::kRccl;
        }
    }
    *out = m;
    return RT_OK;
}

int rtMultiSetScene(RtMulti* m, const RtSphere* spheres, const RtMaterial* materials, uint32_t n_spheres) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "rtMultiSetScene: handle is null");
    for (size_t g = 0; g < m->ctx.size(); ++g) {  // every GPU holds the whole (tiny) scene
        const int rc = rtSetScene(m->ctx[g], spheres, materials, n_spheres);
        if (rc != RT_OK) return mfail(m, rc, std::string("rtMultiSetScene: ") + rtGetLastError(m->ctx[g]));
    }
    return RT_OK;
}

int rtMultiRender(RtMulti* m, const RtCamera* cam, const RtParams* params, void* dst, size_t dst_pitch, int dst_is_device) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "rtMultiRender: handle is null");
    if (!params || !dst) return mfail(m, RT_ERR_INVALID, "rtMultiRender: params or dst is null");
    if (params->tile_count > 1) return mfail(m, RT_ERR_INVALID, "rtMultiRender: the tiling is this call's; leave tile_count 0");
    const uint32_t W = params->width, H = params->height;
    if (W == 0 || H == 0) return mfail(m, RT_ERR_INVALID, "rtMultiRender: empty image");
    if (dst_pitch < size_t(W) * 4 || (dst_pitch & 3u)) return mfail(m, RT_ERR_INVALID, "rtMultiRender: bad dst_pitch");
    const uint32_t n = static_cast<uint32_t>(m->ctx.size());
    const int root = m->devices[0];
    m->have_timing = false;
    if (m->transport == Transport::kSingle) {  // one device: the single-GPU path, untouched
        const int rc = rtRender(m->ctx[0], cam, params, dst, dst_pitch, dst_is_device, nullptr);
        if (rc != RT_OK) return mfail(m, rc, std::string("rtMultiRender: ") + rtGetLastError(m->ctx[0]));
        return RT_OK;
    }
    if (params->mode != RT_MODE_PATH)
        return mfail(m, RT_ERR_INVALID, "rtMultiRender: CH05/CH06 frames are one 7-us dispatch; render them with rtRenderUbo");
    const uint32_t block = params->row_block ? params->row_block : 4u;
    const uint32_t slot_rows = rows_max(H, block, n);
    const size_t slot_words = size_t(slot_rows) * W;
    int rc = ensure_words(m, root, &m->d_gather, &m->gather_words, slot_words * n);
    if (rc != RT_OK) return rc;
    for (uint32_t g = 1; g < n; ++g) {
        rc = ensure_words(m, m->devices[g], &m->d_tile[g], &m->tile_words[g], slot_words);
        if (rc != RT_OK) return rc;
    }
    uint32_t* frame = static_cast<uint32_t*>(dst);
    uint32_t frame_stride = static_cast<uint32_t>(dst_pitch / 4);
    if (!dst_is_device) {
        rc = ensure_words(m, root, &m->d_frame, &m->frame_words, size_t(W) * H);
        if (rc != RT_OK) return rc;
        frame = m->d_frame;
        frame_stride = W;
    }
    hipStream_t root_stream = m->ctx[0]->stream;
    RTM_HIP(m, hipSetDevice(root));
    RTM_HIP(m, hipEventRecord(m->ev_t0, root_stream));
    // From the first launch on, an error leaves work enqueued on some devices and not on others.  The frame is then
    // abandoned in a defined state: every context is drained (so that nothing still writes d_gather / d_tile), the
    // frame counter and ev_t1 are left as the last complete frame set them, and the error goes to the caller.  No
    // other transport is tried.
    auto abandon = [&](int code, const std::string& what) {
        for (size_t g = 0; g < m->ctx.size(); ++g) (void)rtSynchronize(m->ctx[g]);
        (void)hipSetDevice(root);
        return mfail(m, code, what);
    };
    auto hip_check = [&](hipError_t e, const char* what) {
        return e == hipSuccess ? std::string() : std::string("rtMultiRender: ") + what + ": " + hipGetErrorString(e);
    };
    // every device renders its rows on its own context's stream; the root straight into its slot
    for (uint32_t g = 0; g < n; ++g) {
        RtParams p = *params;
        p.row_block = block;
        p.tile_rank = g;
        p.tile_count = n;
        uint32_t* tile = g == 0 ? m->d_gather : m->d_tile[g];
        rc = rtRender(m->ctx[g], cam, &p, tile, size_t(W) * 4, /*dst_is_device*/ 1, nullptr);
        if (rc != RT_OK)
            return abandon(rc, std::string("rtMultiRender: device ") + std::to_string(m->devices[g]) + ": " + rtGetLastError(m->ctx[g]));
    }
    // the gather
    std::string err;
    if (m->transport == Transport::kRccl) {
        int nrc = g_rccl.GroupStart();
        if (nrc != 0) return abandon(RT_ERR_HIP, std::string("rtMultiRender: ncclGroupStart: ") + g_rccl.GetErrorString(nrc));
        // (nothing returns between GroupStart and GroupEnd: an open group would swallow every later RCCL call)
        for (uint32_t g = 0; g < n && nrc == 0 && err.empty(); ++g) {
            err = hip_check(hipSetDevice(m->devices[g]), "hipSetDevice");
            if (!err.empty()) break;
            // in place on the root: its tile already sits at recvbuff + 0 * sendcount (rccl.h:733)
            const void* send = g == 0 ? m->d_gather : m->d_tile[g];
            nrc = g_rccl.Gather(send, g == 0 ? m->d_gather : nullptr, slot_words, Rccl::kUint32, /*root*/ 0, m->comms[g],
                                m->ctx[g]->stream);
        }
        const int erc = g_rccl.GroupEnd();
        if (nrc == 0) nrc = erc;
        if (!err.empty()) return abandon(RT_ERR_HIP, err);
        if (nrc != 0) return abandon(RT_ERR_HIP, std::string("rtMultiRender: ncclGather: ") + g_rccl.GetErrorString(nrc));
    } else {
        for (uint32_t g = 1; g < n && err.empty(); ++g) {
            hipStream_t s = m->ctx[g]->stream;
            err = hip_check(hipSetDevice(m->devices[g]), "hipSetDevice");
            // (the root may still be reading the previous frame out of the gather buffer)
            if (err.empty() && m->frames_done != 0) err = hip_check(hipStreamWaitEvent(s, m->ev_t1, 0), "hipStreamWaitEvent");
            if (err.empty())
                err = hip_check(hipMemcpyPeerAsync(m->d_gather + slot_words * g, root, m->d_tile[g], m->devices[g], slot_words * 4, s),
                                "hipMemcpyPeerAsync");
            if (err.empty()) err = hip_check(hipEventRecord(m->ev_tile[g], s), "hipEventRecord");
        }
        if (err.empty()) err = hip_check(hipSetDevice(root), "hipSetDevice");
        for (uint32_t g = 1; g < n && err.empty(); ++g) err = hip_check(hipStreamWaitEvent(root_stream, m->ev_tile[g], 0), "hipStreamWaitEvent");
        if (!err.empty()) return abandon(RT_ERR_HIP, err);
    }
    // rows to their places
    err = hip_check(hipSetDevice(root), "hipSetDevice");
    if (err.empty()) {
        hipLaunchKernelGGL(deinterleave_kernel, dim3(n * slot_rows), dim3(256), 0, root_stream, m->d_gather, frame, frame_stride,
                           W, H, block, n, slot_rows);
        err = hip_check(hipGetLastError(), "deinterleave_kernel");
    }
    if (err.empty()) err = hip_check(hipEventRecord(m->ev_t1, root_stream), "hipEventRecord");
    if (!err.empty()) return abandon(RT_ERR_HIP, err);
    m->have_timing = true;
    ++m->frames_done;
    if (!dst_is_device) {
        err = hip_check(hipMemcpy2DAsync(dst, dst_pitch, m->d_frame, size_t(W) * 4, size_t(W) * 4, H, hipMemcpyDeviceToHost, root_stream),
                        "hipMemcpy2DAsync");
        if (err.empty()) err = hip_check(hipStreamSynchronize(root_stream), "hipStreamSynchronize");
        if (!err.empty()) return abandon(RT_ERR_HIP, err);
    }
    return RT_OK;
}

int rtMultiSynchronize(RtMulti* m) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "rtMultiSynchronize: handle is null");
    for (size_t g = 0; g < m->ctx.size(); ++g) {
        const int rc = rtSynchronize(m->ctx[g]);
        if (rc != RT_OK) return mfail(m, rc, rtGetLastError(m->ctx[g]));
    }
    return RT_OK;
}

int rtMultiGetStats(RtMulti* m, int device_index, RtStats* out, double* frame_ms) {
    if (!m) return mfail(nullptr, RT_ERR_INVALID, "rtMultiGetStats: handle is null");
    if (device_index < 0 || device_index >= static_cast<int>(m->ctx.size()) || !out)
        return mfail(m, RT_ERR_INVALID, "rtMultiGetStats: bad device index or null out");
    int rc = rtMultiSynchronize(m);
    if (rc != RT_OK) return rc;
    rc = rtGetStats(m->ctx[device_index], out);
    if (rc != RT_OK) return mfail(m, rc, rtGetLastError(m->ctx[device_index]));
    if (frame_ms) {
        *frame_ms = out->kernel_ms;
        if (m->have_timing) {
            float ms = 0.0f;
            RTM_HIP(m, hipSetDevice(m->devices[0]));
            RTM_HIP(m, hipEventElapsedTime(&ms, m->ev_t0, m->ev_t1));
            *frame_ms = ms;
        }
    }
    return RT_OK;
}

// CPU test of the partition, the padded slots and the de-interleave: `full` (height x width words) is cut into
// the N tiles the devices would render, the tiles go through the host transport (memcpy standing in for the
// communicator) into a gather buffer, and the rows are put back.  `out` must come back equal to `full`.
int rtMultiSelfTestHost(const uint32_t* full, uint32_t width, uint32_t height, uint32_t row_block, uint32_t n_tiles,
                        uint32_t* out) {
    if (!full || !out || width == 0 || height == 0 || n_tiles == 0 || n_tiles > 64)
        return mfail(nullptr, RT_ERR_INVALID, "rtMultiSelfTestHost: bad arguments");
    const uint32_t block = row_block ? row_block : 4u;
    const uint32_t slot_rows = rows_max(height, block, n_tiles);
    const size_t slot_words = size_t(slot_rows) * width;
    std::vector<std::vector<uint32_t>> tiles(n_tiles);
    for (uint32_t g = 0; g < n_tiles; ++g) {  // what device g would have rendered, packed ascending; padding poisoned
        tiles[g].assign(slot_words, 0xDEADBEEFu);
        const uint32_t rows = rtTileRowCount(height, block, g, n_tiles);
        for (uint32_t lr = 0; lr < rows; ++lr)
            std::memcpy(tiles[g].data() + size_t(lr) * width, full + size_t(rtTileGlobalRow(lr, block, g, n_tiles)) * width,
                        size_t(width) * 4);
    }
    std::vector<uint32_t> gathered(slot_words * n_tiles, 0u);
    for (uint32_t g = 0; g < n_tiles; ++g)  // host transport: rank g's slot_words to offset g * slot_words on the root
        std::memcpy(gathered.data() + slot_words * g, tiles[g].data(), slot_words * 4);
    std::memset(out, 0, size_t(width) * height * 4);
    deinterleave_host(gathered.data(), out, width, height, block, n_tiles, slot_rows);
    return RT_OK;
}

}  // extern "C"
