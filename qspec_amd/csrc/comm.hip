// One-shot all-reduce over peer-mapped buffers (xGMI), for the latency-bound messages of the tensor-parallel verify
// pass: [T, H] fp32 row-parallel partial sums, 64 KB .. 1 MB.
//
// Contract mirrored: vllm's custom all-reduce (csrc/custom_all_reduce.cuh, one-shot form for small messages;
// vllm/distributed/device_communicators/custom_all_reduce.py:50-56,242-255: world in {2,4,6,8}, < 8 MiB, every rank
// maps every peer's buffer through IPC handles, falls back to the library collective otherwise).  The reference QSpec
// model itself has no tensor parallelism: this is the build's own verify-pass design.
//
// MI355X form: xGMI is point to point (7 links per GPU), so a rank PUSHES its vector straight into a slot of every
// peer's buffer -- one hop per peer, all links busy at once -- instead of walking a ring (2 (p - 1) hops):
//
//   every workgroup w owns one slice of the vector and needs no other workgroup of its rank:
//     1. tag = gen[w] + 1                         (a device word only workgroup w of this rank touches: graph replays
//                                                  need no host-side counter)
//     2. store the slice into slot[parity][rank] of EVERY peer (16-byte stores over xGMI; the own copy stays local),
//        system-scope release, then flag[parity][rank][w] = tag on every peer
//     3. wait until the local flag[parity][p][w] == tag for every peer p   (bounded: a lost peer raises the error
//        word instead of hanging the GPU; a raised word ends every later wait at once)
//     4. sum the world slices in RANK ORDER in fp32 -> out: every rank computes the same bits, run after run
//     5. gen[w] = tag
//   Slots are double buffered by the parity of the tag: a peer can be at most one call ahead (it needs this rank's
//   flag of call n to leave call n), so it writes parity (n + 1) & 1 while this rank still reads parity n & 1.
//
// The buffer is allocated UNCACHED (hipDeviceMallocUncached): peers' stores land in memory and this GPU's reads are not
// served from a stale L2 line.  Handles travel through the host (torch.distributed store); see qspec_amd/parallel.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "kernels.h"

namespace qspec {

#define QS_AR_MAX_WORLD 8
#define QS_AR_GRID 32            // workgroups = slices of the vector
#define QS_AR_FLAG_WORDS (2 * QS_AR_MAX_WORLD * QS_AR_GRID)

struct OneShotCtx {
    int rank, world;
    size_t max_bytes;            // capacity of one slot
    char* local;                 // this rank's buffer (uncached device memory)
    char* peer[QS_AR_MAX_WORLD]; // every rank's buffer as mapped here (peer[rank] == local)
    bool opened[QS_AR_MAX_WORLD];
};

// buffer layout: [error word + gen[QS_AR_GRID] | flags[2][world][GRID] | slots[2][world][max_bytes]]
__host__ __device__ inline size_t ar_flags_off() { return 1024; }
__host__ __device__ inline size_t ar_slots_off() { return 1024 + QS_AR_FLAG_WORDS * sizeof(uint32_t); }
static size_t ar_total_bytes(size_t max_bytes) { return ar_slots_off() + (size_t)2 * QS_AR_MAX_WORLD * max_bytes; }

struct ArPeers {
    char* p[QS_AR_MAX_WORLD];
};

__global__ __launch_bounds__(256) void oneshot_all_reduce_f32_kernel(ArPeers peers, int rank, int world,
                                                                      size_t max_bytes, float* __restrict__ data, int n) {
    const int w = blockIdx.x, tid = threadIdx.x;
    char* local = peers.p[rank];
    uint32_t* gen = reinterpret_cast<uint32_t*>(local) + 16;
    const uint32_t tag = gen[w] + 1u;
    const int par = tag & 1u;
    // slice of workgroup w, in float4 units
    const int n4 = (n + 3) / 4;
    const int per = (n4 + QS_AR_GRID - 1) / QS_AR_GRID;
    const int lo = w * per, hi = min(n4, lo + per);
    const float4* src = reinterpret_cast<const float4*>(data);
    // 2. push to every peer (the tail of the last float4 may read past n: the caller's buffer is padded to 16 bytes)
    for (int pr = 0; pr < world; pr++) {
        if (pr == rank) continue;
        float4* dst = reinterpret_cast<float4*>(peers.p[pr] + ar_slots_off() + ((size_t)par * QS_AR_MAX_WORLD + rank) * max_bytes);
        for (int i = lo + tid; i < hi; i += 256) dst[i] = src[i];
    }
    __threadfence_system();
    __syncthreads();
    if (tid < world && tid != rank) {
        uint32_t* f = reinterpret_cast<uint32_t*>(peers.p[tid] + ar_flags_off()) + ((size_t)par * QS_AR_MAX_WORLD + rank) * QS_AR_GRID + w;
        __hip_atomic_store(f, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. wait for every peer's slice
    if (tid < world && tid != rank) {
        const uint32_t* f = reinterpret_cast<const uint32_t*>(local + ar_flags_off()) + ((size_t)par * QS_AR_MAX_WORLD + tid) * QS_AR_GRID + w;
        // The sticky error word short-circuits the wait: once ONE wait of this rank has timed out (or the host has raised
        // the word), every later all-reduce of the captured cycle returns at once with an invalid sum -- only the first
        // collective pays the bound, not 2 per layer x 32-80 layers x 20 s of a busy-spinning GPU.  The cycle's result is
        // discarded anyway: the word reaches the host with the tokens and the worker re-runs or raises.
        const uint32_t* errw = reinterpret_cast<const uint32_t*>(local);
        bool dead = __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        int guard = 0;
        while (!dead && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
            __builtin_amdgcn_s_sleep(8);
            ++guard;
            if ((guard & 1023) == 0 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                dead = true;   // another workgroup's (or an earlier call's) wait has already given up on the peer
            // ~20 s: a peer that is merely LATE (a long host step, a collector pause) is not a dead one -- vllm's custom
            // all-reduce waits without a bound; the bound here only keeps a lost peer from hanging the GPU for good
            if (guard > (1 << 25)) {
                __hip_atomic_store(reinterpret_cast<uint32_t*>(local), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    __threadfence_system();
    // 4. reduce in rank order
    float4* out = reinterpret_cast<float4*>(data);
    for (int i = lo + tid; i < hi; i += 256) {
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int pr = 0; pr < world; pr++) {
            float4 v;
            if (pr == rank) {
                v = src[i];
            } else {
                const float4* s = reinterpret_cast<const float4*>(local + ar_slots_off() + ((size_t)par * QS_AR_MAX_WORLD + pr) * max_bytes);
                v = s[i];
            }
            if (pr == 0) acc = v;
            else { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
        }
        out[i] = acc;
    }
    if (tid == 0) gen[w] = tag;
}

}  // namespace qspec

using qspec::OneShotCtx;

extern "C" {

int qspec_oneshot_create(int rank, int world, size_t max_bytes, void** ctx_out) {
    if (!ctx_out || world < 2 || world > QS_AR_MAX_WORLD || rank < 0 || rank >= world || max_bytes == 0 || max_bytes % 16) return 1;
    OneShotCtx* c = new OneShotCtx();
    memset(c, 0, sizeof(*c));
    c->rank = rank; c->world = world; c->max_bytes = max_bytes;
    void* p = nullptr;
    const size_t total = qspec::ar_total_bytes(max_bytes);
    if (hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return 2;
    }
    if (hipMemset(p, 0, qspec::ar_slots_off()) != hipSuccess) { (void)hipFree(p); delete c; return 3; }
    (void)hipDeviceSynchronize();
    c->local = static_cast<char*>(p);
    c->peer[rank] = c->local;
    *ctx_out = c;
    return 0;
}

int qspec_oneshot_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int qspec_oneshot_local_handle(void* ctx, void* handle_out) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    if (!c || !handle_out) return 1;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, c->local) != hipSuccess) { (void)hipGetLastError(); return 2; }
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

// handles: world x qspec_oneshot_handle_bytes(), rank-major (this rank's own entry is ignored)
int qspec_oneshot_open_peers(void* ctx, const void* handles) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    if (!c || !handles) return 1;
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return 2 + r; }
        c->peer[r] = static_cast<char*>(p);
        c->opened[r] = true;
    }
    return 0;
}

// data [n] fp32, in place; n * 4 <= max_bytes; the allocation behind `data` must extend to a multiple of 16 bytes.
int qspec_oneshot_all_reduce_f32(void* ctx, float* data, int n, void* stream) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    if (!c || !data || n < 0) return 1;
    if (n == 0) return 0;
    if ((size_t)((n + 3) / 4) * 16 > c->max_bytes || ((uintptr_t)data) % 16) return 2;
    qspec::ArPeers peers;
    for (int r = 0; r < QS_AR_MAX_WORLD; r++) peers.p[r] = r < c->world ? c->peer[r] : nullptr;
    for (int r = 0; r < c->world; r++)
        if (!peers.p[r]) return 3;
    hipLaunchKernelGGL(qspec::oneshot_all_reduce_f32_kernel, dim3(QS_AR_GRID), dim3(256), 0, static_cast<hipStream_t>(stream),
                       peers, c->rank, c->world, c->max_bytes, data, n);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

// sticky error word (a wait timed out): 0 = fine.  Synchronises the device.
int qspec_oneshot_error(void* ctx) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    if (!c) return -1;
    uint32_t v = 0;
    if (hipMemcpy(&v, c->local, 4, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    return (int)v;
}

void* qspec_oneshot_error_word(void* ctx) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    return c ? static_cast<void*>(c->local) : nullptr;
}

int qspec_oneshot_destroy(void* ctx) {
    OneShotCtx* c = static_cast<OneShotCtx*>(ctx);
    if (!c) return 0;
    for (int r = 0; r < c->world; r++)
        if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    (void)hipFree(c->local);
    delete c;
    return 0;
}

}  // extern "C"
