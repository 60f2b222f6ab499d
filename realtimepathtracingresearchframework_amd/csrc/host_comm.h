// host_comm.h -- the path's one collective: a gather of tile radiance to rank 0 (include/rptr_hip.h "multi-GPU").
//
// Included once, at the end of rptr_hip.hip (it works on the handle's internals). No reference counterpart: the reference renders on
// one physical device (vulkan/render_vulkan_extensions.cpp:77-82); the partitioning is SURVEY 8e's -- screen stripes, scene replicas,
// resolved float4 radiance moves, nothing else.
//
// Transport. RCCL has no gather primitive; the idiom is grouped point-to-point: every rank r != 0 posts ONE ncclSend of its packed rows
// (the frame context's own image: its rows top to bottom are exactly the tile, no staging copy), rank 0 posts N-1 ncclRecv into one
// receive buffer, all inside ncclGroupStart/End. On xGMI every peer has its own link to rank 0, so the N-1 transfers run side by
// side (1080p: 4.1 MB per link and frame) -- this is not a ring collective and not bound by a ring's per-link rate. One HIP kernel then
// interleaves the stripes into the frame (HBM stream: 2 x 16 B per pixel). librccl is dlopen'ed (librccl.so.1: the copy already in the
// process -- PyTorch ships one -- or the system's), so single-GPU hosts and the build container never need it.
// One process with several handles can also move the rows with hipMemcpyPeerAsync (handles that share a device in test rigs, or
// RPTR_COMM_TRANSPORT=copy); same stream / event structure, same assembly kernel.
// Peer writes (round 4; RPTR_COMM_TRANSPORT=peer, one-process groups): RCCL's receive side costs rank 0 -- its proxy kernels hold CUs of
// an issue-bound renderer (+18 % on rank 0's frames, tools/gather_cost.py) and rank 0 then still runs the assembly pass over the whole
// frame. With peer access on (hipDeviceEnablePeerAccess) every rank instead SCATTERS its rows itself, straight into their places in rank
// 0's frame (rp_k_scatter_rows on the rank's own communication stream: 30 KB contiguous per row, full lines over the rank's own xGMI
// link): rank 0 runs no receive kernels and no assembly, only the scatter of its own rows; its stream waits for one event per peer.
// Batched gathers (round 4; rptr_hip_gather_batch / _gather_all_batch): "fewer, larger collectives" -- the frames of one launch sequence
// finish together and lie behind each other in the frame context's image array, so ONE transfer per rank (n x its rows) and ONE assembly
// pass move all n of them: measured beside a renderer whose frames take 0.2 ms (what a rank of an 8-way split renders), a gather per
// frame costs +44 % (RCCL) / +16 % (peer writes) of the frame rate, a gather per sequence of four a quarter of that (tools/gather_cost.py).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: nothing links against librccl

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    if (api.lib || !api.error.empty()) return api;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) // a copy that is already loaded wins (two RCCL instances in one process would each own the devices)
        if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.lib)
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.lib) {
        api.error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?");
        return api;
    }
    auto sym = [&](const char *name) {
        void *p = dlsym(api.lib, name);
        if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + name;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    return api;
}

enum { COMM_RCCL = 0, COMM_COPY = 1, COMM_PEER = 2, COMM_IPC = 3 };

// COMM_IPC: the peer-write transport for ONE PROCESS PER GPU (round 4, opt-in: rptr_hip_comm_ipc_export / _ipc_init). Rank 0 exports its two
// assembled-frame buffers and a small flag block through hipIpcGetMemHandle; every other rank maps them (hipIpcOpenMemHandle) and, per gather,
// scatters its rows straight into rank 0's frame from a kernel on its own communication stream -- no RCCL kernels on either side, no receive
// buffer, no assembly pass. Events do not cross processes reliably (a wait issued before the other side's record sees an unrecorded
// event), so the ordering is carried by monotone counters in the flag block (uncached device memory of rank 0):
//   gate      written by rank 0's communication stream when it reaches gather g: everything rank 0 queued before (the gather that last
//             used this slot, a read-back of it) is done -- the peers' writes of gather g wait for gate >= g (a one-lane kernel that polls);
//   done[r]   written by rank r's communication stream behind its scatter kernel: rank 0's stream waits for done[r] >= g of every peer.
// Every rank counts its gathers itself (a gather is a collective: the same sequence of calls on every rank).
struct RptrIpcFlags {
    uint32_t gate;
    uint32_t _pad[15];
    uint32_t done[240]; // one word per rank (world <= 240)
};
struct RptrIpcBlob { // what rptr_hip_comm_ipc_export hands to the other ranks (RPTR_COMM_IPC_BYTES)
    hipIpcMemHandle_t frame[2], flags;
    uint64_t npix;
    int32_t max_batch, world;
    uint32_t magic;
};

} // namespace

struct RptrComm {
    int transport = COMM_RCCL;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;     // everything of a gather runs here, behind the frame it sends
    hipEvent_t ev_src = nullptr;      // "the waited frame is visible" (recorded on the backend's stream)
    hipEvent_t ev_done = nullptr;     // the last gather of this rank has finished (send done / frame assembled)
    // rank 0: two slots, used in turn (gather g works on slot g % 2), so that a reader can hold frame i while frame i + 1 arrives and is
    // assembled, and so that a peer's copy for gather g + 1 never lands in the rows gather g is still assembling from
    float4 *recv[2] = {nullptr, nullptr};      // packed rows of the ranks 1..N-1 (RPTR_COMM_SELF: and of rank 0), rank after rank; a rank's block
                                               // holds `max_batch` images of its rows (a batched gather fills the first n)
    float4 *gathered[2] = {nullptr, nullptr};  // the assembled frame(s): max_batch images of width * height
    int max_batch = 1;                // frames one gather can move (the handle's max_batch_frames; 1 without frames in flight)
    int last_batch = 1;               // frames the last gather moved: rptr_hip_gathered_frame / _readback_gathered_f32 show the LAST of them
    hipEvent_t ev_slot[2] = {nullptr, nullptr}; // "the assembly that used this slot has finished" (recorded on rank 0's stream)
    bool slot_used[2] = {false, false};
    int last_slot = 0;                // the slot of the last gather: what rptr_hip_gathered_frame / _readback_gathered_f32 show
    size_t bytes_owned = 0;           // device bytes of the buffers above (counted in the handle's bytes_frame)
    unsigned long long *d_offsets = nullptr; // per rank: first float4 of its block in `recv`, then (world more entries) the pixels of one image of its rows
    std::vector<size_t> rank_pixels, rank_offset;
    bool self = false;                // RPTR_COMM_SELF=1 (diagnostic): rank 0's own rows also travel through ncclSend / ncclRecv
    // one process, several handles: the peers (rank order) and, for peer copies, the events that tell rank 0 a peer's rows have landed
    std::vector<rptr_hip *> peers;
    hipEvent_t ev_copied = nullptr;
    // COMM_IPC: rank 0 owns `ipc_flags`; the other ranks hold rank 0's buffers mapped into their address space
    RptrIpcFlags *ipc_flags = nullptr;
    uint32_t *ipc_gave_up = nullptr;  // a peer's own word: its gate wait timed out, the scatter and the done flag of that gather are skipped
    float4 *ipc_frame[2] = {nullptr, nullptr};
    bool ipc_mapped = false;          // (this rank opened the handles: close them on release)
    hipEvent_t ev_gate = nullptr;     // (rank 0, peer writes) what rank 0's communication stream held when a gather began: the peers' writes into the slot wait for it
    // statistics
    uint64_t gathers = 0, timed = 0;
    double total_ms = 0.0;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    bool timing_pending = false;
};

// rows of rank r's tile, packed top to bottom, into the frame
// nb images: image k of rank r's rows lies at its block + k * (pixels of one image of its rows) (own: rank 0's images, same stride)
__global__ __launch_bounds__(256) void rp_k_assemble(float4 *frame, const float4 *own, const float4 *recv, const unsigned long long *offsets, int width,
                                                     int height, int stripe_rows, int world, int self, int nb) {
    const size_t npix = (size_t)width * height, n = npix * (size_t)nb;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const size_t k = j / npix, i = j - k * npix;
        const int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
        const int s = y / stripe_rows, r = s % world;
        const int local_row = (s / world) * stripe_rows + (y - s * stripe_rows);
        const float4 *src = ((r == 0 && !self) ? own : recv + offsets[r]) + k * offsets[world + r];
        frame[j] = src[(size_t)local_row * width + x];
    }
}

// the peer-write transport: the rows of rank `rank` (packed top to bottom) go to their places in the frame, which may live on another device
// skip (COMM_IPC): this rank's gate wait gave up -- rank 0 may still be reading the slot: nothing is written
__global__ __launch_bounds__(256) void rp_k_scatter_rows(float4 *frame, const float4 *rows, int width, int height, int local_rows, int stripe_rows, int world,
                                                         int rank, int nb, const uint32_t *skip = nullptr) {
    if (skip && *skip != 0u) return;
    const size_t per = (size_t)width * local_rows, n = per * (size_t)nb;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const size_t k = j / per, i = j - k * per;
        const int lr = (int)(i / (size_t)width), x = (int)(i - (size_t)lr * width);
        const int s = lr / stripe_rows;
        const int y = (s * world + rank) * stripe_rows + (lr - s * stripe_rows);
        frame[k * ((size_t)width * height) + (size_t)y * width + x] = rows[j];
    }
}

// COMM_IPC flag kernels (one lane each). The flag block is uncached device memory of rank 0; a peer reaches it through its IPC mapping.
// skip: see rp_k_scatter_rows -- a rank that did not write its rows does not say it did (rank 0's wait for it then times out and reports)
__global__ void rp_k_flag_set(uint32_t *flag, uint32_t value, const uint32_t *skip = nullptr) {
    if (skip && *skip != 0u) return;
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // (behind the kernels queued before it: stream order)
}
// waits until flags[i * stride] >= value for i = 0..n-1; gives up after ~30 s (a rank that died must not wedge the others' GPUs) and says so
// (timed_out: a counter in rank 0's flag block, read back by the host -- rptr_hip_comm_stats / the gathered read-backs fail when it is not 0;
// gave_up: a word of THIS rank's own memory that makes the rest of its gather a no-op)
__global__ void rp_k_flag_wait(const uint32_t *flags, int n, int stride, uint32_t value, uint32_t *timed_out, uint32_t *gave_up = nullptr) {
    if (gave_up) *gave_up = 0u;
    for (int i = 0; i < n; ++i) {
        const uint32_t *f = flags + (size_t)i * stride;
        const long long t0 = wall_clock64();
        // (counters only grow; a signed difference keeps the compare right across a wrap after 2^31 gathers)
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > 3000000000ll) { // 30 s of the 100 MHz counter
                if (timed_out) atomicAdd(timed_out, 1u);
                if (gave_up) *gave_up = 1u;
                return;
            }
        }
    }
}

namespace {

int comm_fail_nccl(rptr_hip *h, const char *what, ncclResult_t r) {
    return fail(h, RPTR_E_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
}

#define NCCL_TRY(h, expr)                                          \
    do {                                                           \
        ncclResult_t _r = (expr);                                  \
        if (_r != ncclSuccess) return comm_fail_nccl(h, #expr, _r); \
    } while (0)

// what every rank needs besides the communicator: stream, events and (rank 0) the receive buffer, the frame, the offsets
int comm_setup_local(rptr_hip *h, int transport) {
    if (h->width == 0) return fail(h, RPTR_E_INVALID, "communicator before initialize (the receive buffers depend on the frame size)");
    HIP_TRY(h, hipSetDevice(h->device));
    RptrComm *c = new RptrComm();
    c->transport = transport;
    c->self = h->opt.v[OPT_COMM_SELF] != 0;
    h->comm = c;
    // the gather's kernels are tiny next to the persistent traversal launches they run beside: a high-priority queue gets them their CU
    // slots first (option "comm_priority" = 0: a plain stream)
    {
        int least = 0, greatest = 0;
        if (h->opt.v[OPT_COMM_PRIORITY] != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            HIP_TRY(h, hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
        else
            HIP_TRY(h, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_copied, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreate(&c->ev_t0));
    HIP_TRY(h, hipEventCreate(&c->ev_t1));
    c->max_batch = h->ctx.size() > 1 ? std::max(1, h->max_batch_frames) : 1;
    c->rank_pixels.assign((size_t)h->world, 0);
    c->rank_offset.assign((size_t)h->world, 0);
    size_t at = 0;
    for (int r = 0; r < h->world; ++r) {
        c->rank_pixels[(size_t)r] = (size_t)h->width * local_row_count(h->height, h->stripe_rows, r, h->world);
        c->rank_offset[(size_t)r] = at;
        if (r != 0 || c->self) at += c->rank_pixels[(size_t)r] * (size_t)c->max_batch;
    }
    if (h->rank == 0) {
        // the communicator owns its buffers (frame-sized: comm_release frees them, and the next initialize drops the communicator)
        const size_t npix = (size_t)h->width * h->height;
        auto own = [&](void **out, size_t bytes) -> int {
            bytes = std::max<size_t>(bytes, 16);
            hipError_t e = hipMalloc(out, bytes);
            if (e != hipSuccess) return fail(h, RPTR_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            c->bytes_owned += bytes;
            h->bytes_frame += bytes;
            h->bytes_allocated = h->bytes_scene + h->bytes_frame;
            return RPTR_OK;
        };
        int rc;
        for (int s = 0; s < 2; ++s) {
            if ((rc = own((void **)&c->recv[s], ((transport == COMM_PEER || transport == COMM_IPC) ? 0 : at) * sizeof(float4)))) return rc; // (peer writes land in the frame itself)
            if ((rc = own((void **)&c->gathered[s], npix * (size_t)c->max_batch * sizeof(float4)))) return rc;
            HIP_TRY(h, hipMemset(c->gathered[s], 0, npix * (size_t)c->max_batch * sizeof(float4)));
            HIP_TRY(h, hipEventCreateWithFlags(&c->ev_slot[s], hipEventDisableTiming));
        }
        if ((rc = own((void **)&c->d_offsets, 2 * (size_t)h->world * sizeof(unsigned long long)))) return rc;
        std::vector<unsigned long long> off(c->rank_offset.begin(), c->rank_offset.end());
        off.insert(off.end(), c->rank_pixels.begin(), c->rank_pixels.end());
        HIP_TRY(h, hipMemcpy(c->d_offsets, off.data(), off.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    }
    return RPTR_OK;
}

void comm_release(rptr_hip *h) {
    RptrComm *c = h->comm;
    if (!c) return;
    (void)hipSetDevice(h->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
    for (hipEvent_t e : {c->ev_src, c->ev_done, c->ev_copied, c->ev_gate, c->ev_t0, c->ev_t1, c->ev_slot[0], c->ev_slot[1]})
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (void *p : {(void *)c->recv[0], (void *)c->recv[1], (void *)c->gathered[0], (void *)c->gathered[1], (void *)c->d_offsets})
        if (p) (void)hipFree(p);
    if (c->ipc_gave_up) (void)hipFree(c->ipc_gave_up);
    if (c->ipc_mapped) {
        for (void *p : {(void *)c->ipc_frame[0], (void *)c->ipc_frame[1], (void *)c->ipc_flags})
            if (p) (void)hipIpcCloseMemHandle(p);
    } else if (c->ipc_flags)
        (void)hipFree(c->ipc_flags);
    h->bytes_frame -= std::min(h->bytes_frame, c->bytes_owned);
    h->bytes_allocated = h->bytes_scene + h->bytes_frame;
    for (FrameCtx &fc : h->ctx) fc.gather_pending = false;
    delete c;
    h->comm = nullptr;
}

// the slot (rank 0's receive buffer + assembled frame) the gather that is being issued works on
inline int comm_slot(const RptrComm *c) { return (int)(c->gathers & 1u); }

// the image the gather sends: the rows of the frame that was waited for last
// nb > 1: the last nb frames of the waited launch sequence (images output_index - nb + 1 .. output_index of its context)
const float4 *comm_source(rptr_hip *h, FrameCtx *&owner, int nb) {
    if (h->output_ctx >= 0) {
        owner = &h->ctx[(size_t)h->output_ctx];
        return owner->out_accum + (size_t)(h->output_index - (nb - 1)) * ((size_t)h->width * (size_t)std::max(h->local_rows, 1));
    }
    owner = &h->ctx[0];
    return h->accum;
}

void comm_collect_timing(RptrComm *c, bool wait) {
    if (!c->timing_pending) return;
    if (wait) (void)hipEventSynchronize(c->ev_t1);
    else if (hipEventQuery(c->ev_t1) != hipSuccess) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1) == hipSuccess) {
        c->total_ms += ms;
        c->timed++;
    }
    c->timing_pending = false;
}

// first half of a rank's gather: order the communication stream behind the waited frame
int comm_begin(rptr_hip *h, const float4 *&src, FrameCtx *&owner, int nb) {
    RptrComm *c = h->comm;
    if (!c) return fail(h, RPTR_E_INVALID, "rptr_hip_gather without a communicator (rptr_hip_comm_init_rank / _init_all)");
    if (nb < 1 || nb > c->max_batch)
        return fail(h, RPTR_E_INVALID, "a gather moves 1..%d frames (the handle's frames per launch sequence), not %d", c->max_batch, nb);
    if (nb > 1 && (h->output_ctx < 0 || h->output_index + 1 < nb))
        return fail(h, RPTR_E_INVALID, "a gather of %d frames wants the LAST frame of a launch sequence of at least %d frames to have been waited for (waited: frame %d of its sequence)", nb, nb, h->output_index);
    if (h->output_overwritten)
        return fail(h, RPTR_E_INVALID, "the image of the last waited frame is being overwritten by a newer frame on the same frame context: gather right "
                                       "after rptr_hip_wait");
    HIP_TRY(h, hipSetDevice(h->device));
    src = comm_source(h, owner, nb);
    HIP_TRY(h, hipEventRecord(c->ev_src, h->stream)); // finish_frame made the backend's stream wait for the frame
    HIP_TRY(h, hipStreamWaitEvent(c->stream, c->ev_src, 0));
    comm_collect_timing(c, false);
    if (!c->timing_pending) HIP_TRY(h, hipEventRecord(c->ev_t0, c->stream));
    return RPTR_OK;
}

// second half: rank 0 assembles; the sender's frame context learns when its image is free again
int comm_end(rptr_hip *h, const float4 *src, FrameCtx *owner, int nb) {
    RptrComm *c = h->comm;
    c->last_batch = nb;
    if (h->rank == 0) {
        const size_t npix = (size_t)h->width * h->height * (size_t)nb;
        const int slot = comm_slot(c);
        if (c->transport != COMM_PEER && c->transport != COMM_IPC) // (peer writes: every rank, this one included, has put its rows into the frame already)
            hipLaunchKernelGGL(rp_k_assemble, dim3(grid_for(h, npix)), dim3(256), 0, c->stream, c->gathered[slot], src, c->recv[slot], c->d_offsets, h->width,
                               h->height, h->stripe_rows, h->world, c->self ? 1 : 0, nb);
        HIP_TRY(h, hipEventRecord(c->ev_slot[slot], c->stream));
        c->slot_used[slot] = true;
        c->last_slot = slot;
    }
    if (!c->timing_pending) {
        HIP_TRY(h, hipEventRecord(c->ev_t1, c->stream));
        c->timing_pending = true;
    }
    HIP_TRY(h, hipEventRecord(c->ev_done, c->stream));
    if (!owner->ev_gather) HIP_TRY(h, hipEventCreateWithFlags(&owner->ev_gather, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(owner->ev_gather, c->stream)); // the next frame on this context waits for it before it touches the image
    owner->gather_pending = true;
    c->gathers++;
    HIP_TRY(h, hipGetLastError());
    return RPTR_OK;
}

// this rank's sends / receives (inside the caller's ncclGroupStart / End)
int comm_post_rccl(rptr_hip *h, const float4 *src, int nb) {
    RptrComm *c = h->comm;
    RcclApi &R = rccl();
    const size_t mine = c->rank_pixels[(size_t)h->rank] * (size_t)nb; // (the nb images of a rank's rows lie behind each other on both sides)
    if ((h->rank != 0 || c->self) && mine) NCCL_TRY(h, R.Send(src, mine * 4, ncclFloat, 0, c->nccl, c->stream));
    if (h->rank == 0)
        for (int r = c->self ? 0 : 1; r < h->world; ++r)
            if (c->rank_pixels[(size_t)r]) NCCL_TRY(h, R.Recv(c->recv[comm_slot(c)] + c->rank_offset[(size_t)r], c->rank_pixels[(size_t)r] * (size_t)nb * 4, ncclFloat, r, c->nccl, c->stream));
    return RPTR_OK;
}

} // namespace

extern "C" {

int rptr_hip_comm_get_unique_id(void *out_id128) {
    if (!out_id128) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    RcclApi &R = rccl();
    if (!R.error.empty()) return fail(nullptr, RPTR_E_UNSUPPORTED, "%s", R.error.c_str());
    ncclUniqueId id;
    static_assert(sizeof(id) == RPTR_COMM_ID_BYTES, "ncclUniqueId size");
    NCCL_TRY(nullptr, R.GetUniqueId(&id));
    memcpy(out_id128, &id, sizeof(id));
    return RPTR_OK;
}

int rptr_hip_comm_init_rank(rptr_hip_t *h, const void *id128) {
    if (!h || !id128) return fail(h, RPTR_E_INVALID, "NULL argument");
    RcclApi &R = rccl();
    if (!R.error.empty()) return fail(h, RPTR_E_UNSUPPORTED, "%s", R.error.c_str());
    comm_release(h);
    int rc = comm_setup_local(h, COMM_RCCL);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCL_TRY(h, R.CommInitRank(&h->comm->nccl, h->world, id, h->rank));
    return RPTR_OK;
}

int rptr_hip_comm_init_all(rptr_hip_t *const *handles, int n) {
    if (!handles || n < 1) return fail(nullptr, RPTR_E_INVALID, "bad argument");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        rptr_hip *h = handles[i];
        if (!h) return fail(nullptr, RPTR_E_INVALID, "handle %d is NULL", i);
        if (h->rank != i || h->world != n) return fail(h, RPTR_E_INVALID, "handle %d has rank %d of %d: handles[i] must be rank i of n = %d", i, h->rank, h->world, n);
        if (h->width != handles[0]->width || h->height != handles[0]->height || h->stripe_rows != handles[0]->stripe_rows)
            return fail(h, RPTR_E_INVALID, "handle %d: frame size / stripe rows differ from handle 0", i);
        for (int j = 0; j < i; ++j) distinct = distinct && handles[j]->device != h->device;
    }
    int transport = distinct ? COMM_RCCL : COMM_COPY; // RCCL refuses two ranks on one device: such rigs move the rows with copies
    switch (handles[0]->opt.v[OPT_COMM_TRANSPORT]) { // option "comm_transport" of handle 0 (0: the choice above)
    case 1: transport = COMM_RCCL; break;
    case 2: transport = COMM_COPY; break;
    case 3: transport = COMM_PEER; break;
    default: break;
    }
    RcclApi &R = rccl();
    if (transport == COMM_RCCL && !R.error.empty()) return fail(handles[0], RPTR_E_UNSUPPORTED, "%s", R.error.c_str());
    if (transport == COMM_PEER)
        for (int i = 1; i < n; ++i)
            if (handles[i]->device != handles[0]->device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, handles[i]->device, handles[0]->device) != hipSuccess || !can)
                    return fail(handles[i], RPTR_E_UNSUPPORTED, "comm_transport = peer: device %d cannot access device %d's memory", handles[i]->device,
                                handles[0]->device);
            }
    for (int i = 0; i < n; ++i) {
        comm_release(handles[i]);
        int rc = comm_setup_local(handles[i], transport);
        if (rc) return rc;
        handles[i]->comm->peers.assign(handles, handles + n);
    }
    if (transport == COMM_RCCL) {
        ncclUniqueId id;
        NCCL_TRY(handles[0], R.GetUniqueId(&id));
        // one thread, several devices: the rank initialisations must be fused into one group (rccl.h ncclCommInitRank)
        NCCL_TRY(handles[0], R.GroupStart());
        for (int i = 0; i < n; ++i) {
            HIP_TRY(handles[i], hipSetDevice(handles[i]->device));
            NCCL_TRY(handles[i], R.CommInitRank(&handles[i]->comm->nccl, n, id, i));
        }
        NCCL_TRY(handles[0], R.GroupEnd());
    } else {
        for (int i = 1; i < n; ++i) // peer access for the copies / the peer writes (a no-op error when it is already on or the device is the same)
            if (handles[i]->device != handles[0]->device) {
                (void)hipSetDevice(handles[i]->device);
                (void)hipDeviceEnablePeerAccess(handles[0]->device, 0);
                (void)hipSetDevice(handles[0]->device);
                (void)hipDeviceEnablePeerAccess(handles[i]->device, 0);
                (void)hipGetLastError();
            }
    }
    return RPTR_OK;
}

int rptr_hip_comm_ipc_export(rptr_hip_t *h, void *out_bytes) {
    static_assert(sizeof(RptrIpcBlob) <= RPTR_COMM_IPC_BYTES, "RPTR_COMM_IPC_BYTES");
    if (!h || !out_bytes) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (h->rank != 0) return fail(h, RPTR_E_INVALID, "rank 0 exports its frame (this handle is rank %d)", h->rank);
    if (h->world > 240) return fail(h, RPTR_E_UNSUPPORTED, "the IPC transport holds 240 ranks");
    comm_release(h);
    int rc = comm_setup_local(h, COMM_IPC);
    if (rc) return rc;
    RptrComm *c = h->comm;
    // the flag block: uncached, so that rank 0's polls see what the peers store over the fabric (and the peers' polls what rank 0 stores)
    hipError_t e = hipExtMallocWithFlags((void **)&c->ipc_flags, sizeof(RptrIpcFlags), hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags((void **)&c->ipc_flags, sizeof(RptrIpcFlags), hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) return fail(h, RPTR_E_NOMEM, "hipExtMallocWithFlags(flag block) failed: %s", hipGetErrorString(e));
    HIP_TRY(h, hipMemset(c->ipc_flags, 0, sizeof(RptrIpcFlags)));
    HIP_TRY(h, hipDeviceSynchronize());
    c->ipc_frame[0] = c->gathered[0];
    c->ipc_frame[1] = c->gathered[1];
    RptrIpcBlob blob;
    memset(&blob, 0, sizeof(blob));
    HIP_TRY(h, hipIpcGetMemHandle(&blob.frame[0], c->gathered[0]));
    HIP_TRY(h, hipIpcGetMemHandle(&blob.frame[1], c->gathered[1]));
    HIP_TRY(h, hipIpcGetMemHandle(&blob.flags, c->ipc_flags));
    blob.npix = (uint64_t)h->width * h->height;
    blob.max_batch = c->max_batch;
    blob.world = h->world;
    blob.magic = 0x52495043u; // "RIPC"
    memset(out_bytes, 0, RPTR_COMM_IPC_BYTES);
    memcpy(out_bytes, &blob, sizeof(blob));
    return RPTR_OK;
}

int rptr_hip_comm_ipc_init(rptr_hip_t *h, const void *bytes) {
    if (!h || !bytes) return fail(h, RPTR_E_INVALID, "NULL argument");
    RptrIpcBlob blob;
    memcpy(&blob, bytes, sizeof(blob));
    if (blob.magic != 0x52495043u) return fail(h, RPTR_E_INVALID, "not a blob of rptr_hip_comm_ipc_export");
    if (blob.world != h->world || blob.npix != (uint64_t)h->width * h->height)
        return fail(h, RPTR_E_INVALID, "the exported frame (%llu pixels, %d ranks) does not match this handle (%d x %d, %d ranks)", (unsigned long long)blob.npix, blob.world,
                    h->width, h->height, h->world);
    if (h->rank == 0) { // rank 0 made everything in rptr_hip_comm_ipc_export
        if (!h->comm || h->comm->transport != COMM_IPC) return fail(h, RPTR_E_INVALID, "rank 0: call rptr_hip_comm_ipc_export first");
        return RPTR_OK;
    }
    comm_release(h);
    int rc = comm_setup_local(h, COMM_IPC);
    if (rc) return rc;
    RptrComm *c = h->comm;
    // every rank moves the same number of frames per gather, and the ranks count their gathers themselves: the limit is rank 0's for all of
    // them (a rank that could hold fewer refuses here instead of failing a gather on its own later and falling out of step)
    if (c->max_batch < blob.max_batch)
        return fail(h, RPTR_E_INVALID, "this rank holds %d frames per launch sequence, rank 0's frame buffers %d: create every rank with the same "
                                       "\"max_batch_frames\"", c->max_batch, blob.max_batch);
    c->max_batch = blob.max_batch;
    HIP_TRY(h, hipMalloc((void **)&c->ipc_gave_up, sizeof(uint32_t)));
    HIP_TRY(h, hipMemset(c->ipc_gave_up, 0, sizeof(uint32_t)));
    c->ipc_mapped = true;
    HIP_TRY(h, hipIpcOpenMemHandle((void **)&c->ipc_frame[0], blob.frame[0], hipIpcMemLazyEnablePeerAccess));
    HIP_TRY(h, hipIpcOpenMemHandle((void **)&c->ipc_frame[1], blob.frame[1], hipIpcMemLazyEnablePeerAccess));
    HIP_TRY(h, hipIpcOpenMemHandle((void **)&c->ipc_flags, blob.flags, hipIpcMemLazyEnablePeerAccess));
    return RPTR_OK;
}

int rptr_hip_comm_destroy(rptr_hip_t *h) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    comm_release(h);
    return RPTR_OK;
}

int rptr_hip_gather_batch(rptr_hip_t *h, int n_frames) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (!h->comm) return fail(h, RPTR_E_INVALID, "rptr_hip_gather without a communicator (rptr_hip_comm_init_rank)");
    if (h->comm->peers.size() > 1) return fail(h, RPTR_E_INVALID, "this handle belongs to a one-process group: use rptr_hip_gather_all");
    const int nb = n_frames;
    const float4 *src = nullptr;
    FrameCtx *owner = nullptr;
    if (h->comm->transport == COMM_IPC && (!h->comm->ipc_flags || !h->comm->ipc_frame[0]))
        return fail(h, RPTR_E_INVALID, "rptr_hip_comm_ipc_init has not been called on this handle"); // (before anything is counted or begun)
    int rc = comm_begin(h, src, owner, nb);
    if (rc) {
        // (COMM_IPC: the ranks count their gathers themselves. A rank that cannot take part in gather g -- a caller error on this rank only --
        // still counts it, so that gather g + 1 means the same slot and the same flag values on every rank again, AND still publishes its flag
        // of gather g: rank 0 its gate -- the peers would otherwise spin in rp_k_flag_wait until the poll's own 30 s time-out, and their next
        // wait could then pass on gate g + 1 and scatter into a slot rank 0 is still reading --, a peer its done word, so that rank 0 assembles
        // the frame without this rank's rows instead of waiting for them. The error is this rank's return code.)
        RptrComm *c = h->comm;
        if (c && c->transport == COMM_IPC) {
            const uint32_t g = (uint32_t)(c->gathers + 1);
            if (c->stream) {
                uint32_t *flag = h->rank == 0 ? &c->ipc_flags->gate : &c->ipc_flags->done[h->rank];
                hipLaunchKernelGGL(rp_k_flag_set, dim3(1), dim3(1), 0, c->stream, flag, g, (const uint32_t *)nullptr);
                (void)hipGetLastError();
            }
            c->gathers++;
        }
        return rc;
    }
    if (h->comm->transport == COMM_IPC) {
        RptrComm *c = h->comm;
        const uint32_t g = (uint32_t)(c->gathers + 1); // the same number on every rank: a gather is a collective
        const int slot = comm_slot(c);
        const int rows = local_row_count(h->height, h->stripe_rows, h->rank, h->world);
        uint32_t *timed_out = &c->ipc_flags->_pad[0]; // (rank 0's block: counts the waits that gave up, on any rank; the hosts read it back)
        const uint32_t *skip = h->rank == 0 ? nullptr : c->ipc_gave_up;
        if (h->rank == 0) hipLaunchKernelGGL(rp_k_flag_set, dim3(1), dim3(1), 0, c->stream, &c->ipc_flags->gate, g, (const uint32_t *)nullptr);
        else hipLaunchKernelGGL(rp_k_flag_wait, dim3(1), dim3(1), 0, c->stream, (const uint32_t *)&c->ipc_flags->gate, 1, 1, g, timed_out, c->ipc_gave_up);
        if (rows > 0)
            hipLaunchKernelGGL(rp_k_scatter_rows, dim3(grid_for(h, (size_t)h->width * rows * nb)), dim3(256), 0, c->stream, c->ipc_frame[slot], src, h->width, h->height,
                               rows, h->stripe_rows, h->world, h->rank, nb, skip);
        if (h->rank == 0) {
            if (h->world > 1)
                hipLaunchKernelGGL(rp_k_flag_wait, dim3(1), dim3(1), 0, c->stream, (const uint32_t *)&c->ipc_flags->done[1], h->world - 1, 1, g, timed_out, (uint32_t *)nullptr);
        } else
            hipLaunchKernelGGL(rp_k_flag_set, dim3(1), dim3(1), 0, c->stream, &c->ipc_flags->done[h->rank], g, skip);
        HIP_TRY(h, hipGetLastError());
        return comm_end(h, src, owner, nb);
    }
    RcclApi &R = rccl();
    NCCL_TRY(h, R.GroupStart());
    rc = comm_post_rccl(h, src, nb);
    NCCL_TRY(h, R.GroupEnd());
    if (rc) return rc;
    return comm_end(h, src, owner, nb);
}

int rptr_hip_gather(rptr_hip_t *h) { return rptr_hip_gather_batch(h, 1); }

int rptr_hip_gather_all(rptr_hip_t *const *handles, int n) { return rptr_hip_gather_all_batch(handles, n, 1); }

int rptr_hip_gather_all_batch(rptr_hip_t *const *handles, int n, int n_frames) {
    const int nb = n_frames;
    if (!handles || n < 1 || !handles[0]) return fail(nullptr, RPTR_E_INVALID, "bad argument");
    for (int i = 0; i < n; ++i)
        if (!handles[i] || !handles[i]->comm || (int)handles[i]->comm->peers.size() != n || handles[i]->comm->peers[(size_t)i] != handles[i])
            return fail(handles[i], RPTR_E_INVALID, "handle %d is not rank %d of a group made by rptr_hip_comm_init_all", i, i);
    std::vector<const float4 *> src((size_t)n, nullptr);
    std::vector<FrameCtx *> owner((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
        int rc = comm_begin(handles[i], src[(size_t)i], owner[(size_t)i], nb);
        if (rc) return rc;
    }
    rptr_hip *h0 = handles[0];
    RptrComm *c0 = h0->comm;
    if (c0->transport == COMM_RCCL) {
        RcclApi &R = rccl();
        NCCL_TRY(h0, R.GroupStart()); // one group over all devices: sends and receives progress together
        int rc = RPTR_OK;
        for (int i = 0; i < n && !rc; ++i) {
            HIP_TRY(handles[i], hipSetDevice(handles[i]->device));
            rc = comm_post_rccl(handles[i], src[(size_t)i], nb);
        }
        NCCL_TRY(h0, R.GroupEnd());
        if (rc) return rc;
    } else if (c0->transport == COMM_PEER) {
        // peer writes: every rank scatters its rows into rank 0's frame on its OWN communication stream -- behind its frame (comm_begin)
        // and behind whatever rank 0's communication stream held when this gather began (the gather that last used this slot, a read-back
        // of it) --; rank 0's stream waits for every peer's event, then the frame is complete (comm_end records it)
        const int slot = comm_slot(c0);
        float4 *frame = c0->gathered[slot];
        HIP_TRY(h0, hipSetDevice(h0->device));
        HIP_TRY(h0, hipEventRecord(c0->ev_gate, c0->stream));
        for (int i = 0; i < n; ++i) {
            rptr_hip *h = handles[i];
            RptrComm *c = h->comm;
            const int rows = local_row_count(h0->height, h0->stripe_rows, i, n);
            if (rows <= 0) continue;
            HIP_TRY(h, hipSetDevice(h->device));
            if (i != 0) HIP_TRY(h, hipStreamWaitEvent(c->stream, c0->ev_gate, 0));
            hipLaunchKernelGGL(rp_k_scatter_rows, dim3(grid_for(h, (size_t)h0->width * rows * nb)), dim3(256), 0, c->stream, frame, src[(size_t)i], h0->width,
                               h0->height, rows, h0->stripe_rows, n, i, nb);
            HIP_TRY(h, hipGetLastError());
            if (i != 0) {
                HIP_TRY(h, hipEventRecord(c->ev_copied, c->stream));
                HIP_TRY(h0, hipSetDevice(h0->device));
                HIP_TRY(h0, hipStreamWaitEvent(c0->stream, c->ev_copied, 0));
            }
        }
    } else {
        // peer copies: rank r writes its rows into rank 0's receive buffer on its OWN communication stream (behind its frame, and behind
        // the assembly that last read this slot of the receive buffer), rank 0's stream waits for every copy before it assembles
        const int slot = comm_slot(c0);
        float4 *recv = c0->recv[slot];
        for (int i = c0->self ? 0 : 1; i < n; ++i) {
            rptr_hip *h = handles[i];
            RptrComm *c = h->comm;
            const size_t bytes = c0->rank_pixels[(size_t)i] * (size_t)nb * sizeof(float4);
            if (!bytes) continue;
            HIP_TRY(h, hipSetDevice(h->device));
            if (c0->slot_used[slot]) HIP_TRY(h, hipStreamWaitEvent(c->stream, c0->ev_slot[slot], 0));
            if (h->device == h0->device)
                HIP_TRY(h, hipMemcpyAsync(recv + c0->rank_offset[(size_t)i], src[(size_t)i], bytes, hipMemcpyDeviceToDevice, c->stream));
            else
                HIP_TRY(h, hipMemcpyPeerAsync(recv + c0->rank_offset[(size_t)i], h0->device, src[(size_t)i], h->device, bytes, c->stream));
            HIP_TRY(h, hipEventRecord(c->ev_copied, c->stream));
            HIP_TRY(h0, hipSetDevice(h0->device));
            HIP_TRY(h0, hipStreamWaitEvent(c0->stream, c->ev_copied, 0));
        }
    }
    for (int i = 0; i < n; ++i) {
        HIP_TRY(handles[i], hipSetDevice(handles[i]->device));
        int rc = comm_end(handles[i], src[(size_t)i], owner[(size_t)i], nb);
        if (rc) return rc;
    }
    return RPTR_OK;
}

int rptr_hip_gathered_frame(rptr_hip_t *h, const void **out_device_rgba32f) {
    if (!h || !out_device_rgba32f) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (!h->comm || h->rank != 0) return fail(h, RPTR_E_INVALID, "the assembled frame lives on rank 0 of a communicator");
    // the LAST image of the last gather; stays intact during the NEXT gather (two slots), is rewritten by the one after
    *out_device_rgba32f = h->comm->gathered[h->comm->last_slot] + (size_t)(h->comm->last_batch - 1) * ((size_t)h->width * h->height);
    return RPTR_OK;
}

extern "C++" {
// COMM_IPC: waits of the flag protocol that gave up (a rank that died or fell out of step) since the communicator was made -- the gathers
// they belonged to delivered stale or missing rows. Called once the communication stream has been waited for.
static int comm_ipc_check(rptr_hip *h) {
    RptrComm *c = h->comm;
    if (!c || c->transport != COMM_IPC || !c->ipc_flags) return RPTR_OK;
    uint32_t n = 0;
    HIP_TRY(h, hipMemcpy(&n, &c->ipc_flags->_pad[0], sizeof(n), hipMemcpyDeviceToHost));
    if (n != 0u)
        return fail(h, RPTR_E_HIP, "the IPC gather's flag protocol timed out %u time(s) (30 s each): a rank died or the ranks fell out of step; the assembled "
                                   "frames since then are incomplete -- make the communicator again (rptr_hip_comm_ipc_export / _ipc_init)", n);
    return RPTR_OK;
}
}

int rptr_hip_readback_gathered_frame_f32(rptr_hip_t *h, int index, float *rgba, size_t n_floats) {
    if (!h || !rgba) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (!h->comm || h->rank != 0) return fail(h, RPTR_E_INVALID, "the assembled frame lives on rank 0 of a communicator");
    if (index < 0 || index >= h->comm->last_batch) return fail(h, RPTR_E_INVALID, "the last gather moved %d frame(s): no frame %d", h->comm->last_batch, index);
    const size_t need = (size_t)h->width * h->height * 4;
    if (n_floats < need) return fail(h, RPTR_E_INVALID, "read-back buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(rgba, h->comm->gathered[h->comm->last_slot] + (size_t)index * ((size_t)h->width * h->height), need * sizeof(float), hipMemcpyDeviceToHost,
                              h->comm->stream));
    HIP_TRY(h, hipStreamSynchronize(h->comm->stream));
    return comm_ipc_check(h);
}

int rptr_hip_readback_gathered_f32(rptr_hip_t *h, float *rgba, size_t n_floats) {
    if (!h || !h->comm) return fail(h, RPTR_E_INVALID, "the assembled frame lives on rank 0 of a communicator");
    return rptr_hip_readback_gathered_frame_f32(h, h->comm->last_batch - 1, rgba, n_floats);
}

int rptr_hip_comm_stats(rptr_hip_t *h, uint64_t *out_gathers, float *out_mean_gather_ms) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (!h->comm) return fail(h, RPTR_E_INVALID, "no communicator");
    (void)hipSetDevice(h->device);
    comm_collect_timing(h->comm, true);
    if (out_gathers) *out_gathers = h->comm->gathers;
    if (out_mean_gather_ms) *out_mean_gather_ms = h->comm->timed ? (float)(h->comm->total_ms / (double)h->comm->timed) : 0.0f;
    if (h->comm->transport == COMM_IPC) { // (any rank can tell: the counter lives in rank 0's flag block, which every rank has mapped)
        HIP_TRY(h, hipStreamSynchronize(h->comm->stream));
        return comm_ipc_check(h);
    }
    return RPTR_OK;
}

const char *rptr_hip_comm_transport(rptr_hip_t *h) {
    if (!h || !h->comm) return nullptr;
    return h->comm->transport == COMM_RCCL ? "rccl" : h->comm->transport == COMM_COPY ? "copy" : h->comm->transport == COMM_PEER ? "peer" : "ipc";
}

} // extern "C"
