// Gradient exchange of the data-parallel step: a thin wrapper over RCCL (xGMI) behind the C ABI, plus the flat
// cast kernels of the bf16 gradient transport.  The reference is single-GPU (nb_train_iia.py:17 pins one device),
// so this is the build's own contract: one communicator per process, created from a unique-id
// byte blob that the host exchanges out of band, and bucket all-reduces enqueued on a caller-chosen HIP stream.
//
// RCCL is resolved at first use with dlopen (the copy PyTorch-ROCm already mapped when there is one), so a
// single-GPU process never needs the library and libru3d.so carries no link-time dependency on it.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>

namespace {

// the handful of RCCL declarations used here (rccl.h: ncclResult_t / ncclDataType_t / ncclRedOp_t are plain enums)
typedef struct { char internal[RU3D_COMM_ID_BYTES]; } nccl_unique_id;
typedef void* nccl_comm_t;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_SUM = 0, NCCL_AVG = 4 };
enum { NCCL_FLOAT32 = 7, NCCL_BFLOAT16 = 9 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* (*GetLastError)(nccl_comm_t) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {   // prefer an instance that is already mapped (PyTorch-ROCm ships its own)
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) {
        const char* env = getenv("RU3D_RCCL_LIB");
        if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    }
    for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        g_rccl.why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "not found");
        return;
    }
    g_rccl.handle = h;
#define RU3D_SYM(field, name)                                            \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);               \
    if (!g_rccl.field) {                                                 \
        g_rccl.why = std::string("librccl.so lacks symbol ") + name;     \
        return;                                                          \
    }
    RU3D_SYM(GetUniqueId, "ncclGetUniqueId")
    RU3D_SYM(CommInitRank, "ncclCommInitRank")
    RU3D_SYM(CommDestroy, "ncclCommDestroy")
    RU3D_SYM(AllReduce, "ncclAllReduce")
    RU3D_SYM(ReduceScatter, "ncclReduceScatter")
    RU3D_SYM(AllGather, "ncclAllGather")
    RU3D_SYM(GetErrorString, "ncclGetErrorString")
#undef RU3D_SYM
    g_rccl.GetLastError = (decltype(g_rccl.GetLastError))dlsym(h, "ncclGetLastError");   // optional
    g_rccl.ok = true;
}

int need_rccl(const char* what) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) return ru3d_fail(-2, "%s: %s", what, g_rccl.why.c_str());
    return 0;
}

int rccl_fail(const char* what, int rc, nccl_comm_t comm) {
    const char* detail = (g_rccl.GetLastError && comm) ? g_rccl.GetLastError(comm) : "";
    ru3d_fail(1000 + rc, "%s: RCCL error %d (%s) %s", what, rc, g_rccl.GetErrorString(rc), detail ? detail : "");
    return 1000 + rc;
}

struct Comm {
    nccl_comm_t comm;
    int world, rank, device;
};

// ---------------------------------------------------------------- flat casts for the bf16 gradient transport
// dst[i] = (bf16)(src[i] * scale): 8 elements per thread, 16-B stores
__global__ void __launch_bounds__(256) flat_f32_to_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst,
                                                               int64_t count, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < count; i += stride) {
        if (i + 8 <= count) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(src + i + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                o[j] = (bf16)(a[j] * scale);
                o[4 + j] = (bf16)(b[j] * scale);
            }
            *reinterpret_cast<bf16x8*>(dst + i) = o;
        } else {
            for (int64_t j = i; j < count; j++) dst[j] = (bf16)(src[j] * scale);
        }
    }
}

__global__ void __launch_bounds__(256) flat_bf16_to_f32_kernel(const bf16* __restrict__ src, float* __restrict__ dst,
                                                               int64_t count, float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < count; i += stride) {
        if (i + 8 <= count) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i);
            f32x4 a, b;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                a[j] = (float)v[j] * scale;
                b[j] = (float)v[4 + j] * scale;
            }
            *reinterpret_cast<f32x4*>(dst + i) = a;
            *reinterpret_cast<f32x4*>(dst + i + 4) = b;
        } else {
            for (int64_t j = i; j < count; j++) dst[j] = (float)src[j] * scale;
        }
    }
}

}  // namespace

extern "C" int ru3d_comm_unique_id(void* id_out) {
    RU3D_REQUIRE(id_out, "comm_unique_id: null pointer");
    if (int rc = need_rccl("comm_unique_id")) return rc;
    nccl_unique_id id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_unique_id", rc, nullptr);
    memcpy(id_out, id.internal, RU3D_COMM_ID_BYTES);
    return 0;
}

extern "C" int ru3d_comm_init(void** comm_out, const void* unique_id, int world, int rank, int device) {
    RU3D_REQUIRE(comm_out && unique_id, "comm_init: null pointer");
    RU3D_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank %d of %d", rank, world);
    if (int rc = need_rccl("comm_init")) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return ru3d_fail((int)e, "comm_init: %s", hipGetErrorString(e));
    RU3D_REQUIRE(device >= 0 && device < ndev, "comm_init: device %d of %d", device, ndev);
    e = hipSetDevice(device);   // the communicator binds to the calling thread's current device
    if (e != hipSuccess) return ru3d_fail((int)e, "comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
    nccl_unique_id id;
    memcpy(id.internal, unique_id, RU3D_COMM_ID_BYTES);
    nccl_comm_t c = nullptr;
    int rc = g_rccl.CommInitRank(&c, world, id, rank);
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_init", rc, nullptr);
    Comm* h = new Comm{c, world, rank, device};
    *comm_out = h;
    return 0;
}

extern "C" int ru3d_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream) {
    RU3D_REQUIRE(comm && buf && count > 0, "comm_allreduce: bad argument");
    RU3D_REQUIRE(dtype == RU3D_F32 || dtype == RU3D_BF16, "comm_allreduce: dtype must be f32 or bf16");
    Comm* h = (Comm*)comm;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != h->device) {
        hipError_t e = hipSetDevice(h->device);
        if (e != hipSuccess) return ru3d_fail((int)e, "comm_allreduce: hipSetDevice: %s", hipGetErrorString(e));
    }
    int rc = g_rccl.AllReduce(buf, buf, (size_t)count, dtype == RU3D_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16,
                              average ? NCCL_AVG : NCCL_SUM, h->comm, as_stream(stream));
    if (cur != h->device && cur >= 0) (void)hipSetDevice(cur);
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_allreduce", rc, h->comm);
    return 0;
}

// in place as RCCL defines it: the reduce-scatter's receive buffer is the rank's own shard of the send buffer, the
// all-gather's send buffer the rank's own shard of the receive buffer
extern "C" int ru3d_comm_reduce_scatter(void* comm, void* buf, int64_t count_per_rank, int dtype, int average, void* stream) {
    RU3D_REQUIRE(comm && buf && count_per_rank > 0, "comm_reduce_scatter: bad argument");
    RU3D_REQUIRE(dtype == RU3D_F32 || dtype == RU3D_BF16, "comm_reduce_scatter: dtype must be f32 or bf16");
    Comm* h = (Comm*)comm;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != h->device) {
        hipError_t e = hipSetDevice(h->device);
        if (e != hipSuccess) return ru3d_fail((int)e, "comm_reduce_scatter: hipSetDevice: %s", hipGetErrorString(e));
    }
    const size_t esz = dtype == RU3D_F32 ? 4 : 2;
    char* mine = (char*)buf + (size_t)h->rank * (size_t)count_per_rank * esz;
    int rc = g_rccl.ReduceScatter(buf, mine, (size_t)count_per_rank, dtype == RU3D_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16,
                                  average ? NCCL_AVG : NCCL_SUM, h->comm, as_stream(stream));
    if (cur != h->device && cur >= 0) (void)hipSetDevice(cur);
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_reduce_scatter", rc, h->comm);
    return 0;
}

extern "C" int ru3d_comm_all_gather(void* comm, void* buf, int64_t count_per_rank, int dtype, void* stream) {
    RU3D_REQUIRE(comm && buf && count_per_rank > 0, "comm_all_gather: bad argument");
    RU3D_REQUIRE(dtype == RU3D_F32 || dtype == RU3D_BF16, "comm_all_gather: dtype must be f32 or bf16");
    Comm* h = (Comm*)comm;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != h->device) {
        hipError_t e = hipSetDevice(h->device);
        if (e != hipSuccess) return ru3d_fail((int)e, "comm_all_gather: hipSetDevice: %s", hipGetErrorString(e));
    }
    const size_t esz = dtype == RU3D_F32 ? 4 : 2;
    const char* mine = (const char*)buf + (size_t)h->rank * (size_t)count_per_rank * esz;
    int rc = g_rccl.AllGather(mine, buf, (size_t)count_per_rank, dtype == RU3D_F32 ? NCCL_FLOAT32 : NCCL_BFLOAT16, h->comm,
                              as_stream(stream));
    if (cur != h->device && cur >= 0) (void)hipSetDevice(cur);
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_all_gather", rc, h->comm);
    return 0;
}

extern "C" int ru3d_comm_available(void) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) ru3d_fail(-2, "comm_available: %s", g_rccl.why.c_str());
    return g_rccl.ok ? 1 : 0;
}

// ---------------------------------------------------------------- CU budget of the persistent kernels
// A property of the DEVICE, not of the process: how many of a device's CUs the persistent one-workgroup-per-CU kernels
// launched on it may occupy (the rest is left to RCCL's reduction kernels on the side stream).  Two models on two devices
// of one process have two budgets; two models on one device share that device's CUs and therefore its budget.  The
// entry points look the value up for the device their stream belongs to (Ru3dDeviceGuard makes it current).
#define RU3D_MAX_DEVICES 64
static int g_cu_budget[RU3D_MAX_DEVICES];
static int current_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        dev = 0;
    }
    return (dev >= 0 && dev < RU3D_MAX_DEVICES) ? dev : 0;
}
extern "C" int ru3d_set_cu_budget_device(int device, int cus) {
    RU3D_REQUIRE(cus >= 0 && cus <= 1024 && device >= 0 && device < RU3D_MAX_DEVICES, "set_cu_budget: device %d, %d CUs", device, cus);
    g_cu_budget[device] = cus;
    return 0;
}
extern "C" int ru3d_set_cu_budget(int cus) { return ru3d_set_cu_budget_device(current_device_index(), cus); }
extern "C" int ru3d_get_cu_budget(void) {
    const int b = g_cu_budget[current_device_index()];
    return b > 0 ? b : 256;
}

// ---------------------------------------------------------------- kernel probe (bench.py's roofline figure)
// HIP-event pairs recorded by the LIBRARY on the launch stream, right around the main kernel of every 3x3x3 stride-1
// conv launch of one geometry (forward, input gradient, with or without fused statistics / backward sums) - not around
// the memsets, finalize and apply launches that share an entry point with it.  Off unless ru3d_probe_begin was called.
#include <mutex>
#include <vector>
static std::mutex g_probe_mu;
static struct {
    bool on = false;
    int key[6] = {0, 0, 0, 0, 0, 0};
    std::vector<hipEvent_t> ev;      // start, stop, start, stop, ...
} g_probe;

extern "C" int ru3d_probe_begin(int n, int d, int h, int w, int cin, int cout) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    for (hipEvent_t e : g_probe.ev) (void)hipEventDestroy(e);
    g_probe.ev.clear();
    const int k[6] = {n, d, h, w, cin, cout};
    for (int i = 0; i < 6; i++) g_probe.key[i] = k[i];
    g_probe.on = true;
    return 0;
}

extern "C" int ru3d_probe_end(int* launches, double* total_ms) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    g_probe.on = false;
    int n = 0;
    double sum = 0.0;
    for (size_t i = 0; i + 1 < g_probe.ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_probe.ev[i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_probe.ev[i], g_probe.ev[i + 1]) == hipSuccess) {
            n++;
            sum += (double)ms;
        }
    }
    (void)hipGetLastError();
    for (hipEvent_t e : g_probe.ev) (void)hipEventDestroy(e);
    g_probe.ev.clear();
    if (launches) *launches = n;
    if (total_ms) *total_ms = sum;
    return 0;
}

// called by the conv launchers (both storage builds): the stop event of a fresh pair whose start was just recorded, or NULL
void* ru3d_probe_start(int n, int d, int h, int w, int cin, int cout, hipStream_t st) {
    if (!g_probe.on) return nullptr;
    std::lock_guard<std::mutex> lk(g_probe_mu);
    const int k[6] = {n, d, h, w, cin, cout};
    for (int i = 0; i < 6; i++)
        if (g_probe.key[i] != k[i]) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {      // events inside a capture cannot be read
        (void)hipGetLastError();
        return nullptr;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return nullptr;
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return nullptr;
    }
    (void)hipEventRecord(e0, st);
    g_probe.ev.push_back(e0);
    g_probe.ev.push_back(e1);
    return (void*)e1;
}

void ru3d_probe_stop(void* ev, hipStream_t st) {
    if (ev) (void)hipEventRecord((hipEvent_t)ev, st);
}

extern "C" int ru3d_comm_destroy(void* comm) {
    if (!comm) return 0;
    Comm* h = (Comm*)comm;
    int rc = g_rccl.ok ? g_rccl.CommDestroy(h->comm) : NCCL_SUCCESS;
    delete h;
    if (rc != NCCL_SUCCESS) return rccl_fail("comm_destroy", rc, nullptr);
    return 0;
}

extern "C" int ru3d_flat_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t count, float scale,
                              void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(src && dst && count > 0, "flat_cast: bad argument");
    RU3D_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "flat_cast: pointers must be 16-byte aligned");
    int64_t blocks = (count + 2047) / 2048;
    if (blocks > 4096) blocks = 4096;
    if (src_dtype == RU3D_F32 && dst_dtype == RU3D_BF16)
        flat_f32_to_bf16_kernel<<<dim3((unsigned)blocks), 256, 0, as_stream(stream)>>>((const float*)src, (bf16*)dst,
                                                                                        count, scale);
    else if (src_dtype == RU3D_BF16 && dst_dtype == RU3D_F32)
        flat_bf16_to_f32_kernel<<<dim3((unsigned)blocks), 256, 0, as_stream(stream)>>>((const bf16*)src, (float*)dst,
                                                                                        count, scale);
    else
        return ru3d_fail(-1, "flat_cast: only f32 <-> bf16");
    return ru3d_check_launch("flat_cast");
}
