// Shared device/host helpers for libru3d (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "f16_names.h"
#include "../../include/ru3d.h"

// The 16-bit storage element of this build.  The kernel sources are compiled twice into libru3d.so: once with bfloat16
// (RU3D_BF16) and once, with -DRU3D_STORAGE_F16, with IEEE half (RU3D_F16: the reference's apex-O1 arithmetic,
// trainer.py:538-542); `bf16` names whichever it is, and RU3D_BF16 stands for "this build's 16-bit dtype code".
#ifdef RU3D_STORAGE_F16
typedef _Float16 bf16;
#define RU3D_BF16 RU3D_F16
#define RU3D_NS ru3d_f16
#define RU3D_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define RU3D_MFMA_ASM "v_mfma_f32_32x32x16_f16"
#define RU3D_FWD_F16(dt, call)
#else
typedef __bf16 bf16;
#define RU3D_NS ru3d_bf16
#define RU3D_MFMA_32X32X16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define RU3D_MFMA_ASM "v_mfma_f32_32x32x16_bf16"
#define RU3D_DECL_F16(name) extern "C" decltype(name) name##_f16;
RU3D_F16_APIS(RU3D_DECL_F16)
#define RU3D_FWD_F16(dt, call) \
    if ((dt) == RU3D_F16) return call
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
// ds_read_b64_tr_b16: the 16-bit transposed LDS read is type-agnostic; the i16 form serves both storage types
#define RU3D_DS_READ_TR16(p) __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(p)))
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RU3D_WAVE 64

// LDS-DMA: `buffer_load_dwordx4 ... lds` - every lane fetches 16 bytes at its own byte offset inside the buffer (zeros
// when the offset lies beyond the descriptor's range) and the wave's 64 pieces land in 1 KB of consecutive LDS starting
// at `lds_dst` (wave-uniform).  Issued as inline asm on purpose: for the builtin form the compiler drains vmcnt in front
// of the next LDS read it cannot prove disjoint - i.e. right behind the issue, which serialises the fill of the NEXT
// tile with the reads of the current one.  The caller owns the ordering: `s_waitcnt vmcnt(..)` + barrier before the
// destination is read (ru3d_dma_landed_barrier).  Counts in vmcnt like any vector-memory load.
typedef __attribute__((ext_vector_type(4))) int ru3d_i32x4;
__device__ __forceinline__ ru3d_i32x4 ru3d_buffer_rsrc(const void* base, int num_bytes) {
    const uint64_t p = (uint64_t)base;
    ru3d_i32x4 r = {(int)(uint32_t)p, (int)((uint32_t)(p >> 32) & 0xffffu), num_bytes, 0x00020000};
    return r;
}
__device__ __forceinline__ void ru3d_lds_dma16(const ru3d_i32x4& rsrc, const void* lds_dst, int byte_offset) {
    const uint32_t dst = (uint32_t)(uintptr_t)lds_dst;      // the low 32 bits of a flat LDS address are the LDS offset
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :
                 : "s"(dst), "v"(byte_offset), "s"(rsrc)
                 : "memory", "m0");
}
// every wave's LDS-DMA fills have landed and every wave has arrived: the filled buffer may be read, the buffer read
// before this point may be refilled
__device__ __forceinline__ void ru3d_dma_landed_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

int ru3d_fail(int code, const char* fmt, ...);   // sets the thread-local error string, returns code
int ru3d_check_launch(const char* what);         // hipGetLastError -> status

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

// kernel probe (comm.hip): event pair around the main kernel of a conv launch while ru3d_probe_begin's geometry is armed
void* ru3d_probe_start(int n, int d, int h, int w, int cin, int cout, hipStream_t st);
void ru3d_probe_stop(void* ev, hipStream_t st);

// Every launching entry point makes the stream's device current for the duration of the call (PyTorch runs backward
// on its per-device autograd threads and user code may hold a model on a device other than the thread's current
// one); a NULL stream means the current device's default stream.  Restores the previous device on exit.
struct Ru3dDeviceGuard {
    int prev, want;
    explicit Ru3dDeviceGuard(void* stream) : prev(-1), want(-1) {
        if (!stream) return;
        hipDevice_t dev;
        if (hipStreamGetDevice((hipStream_t)stream, &dev) != hipSuccess) { (void)hipGetLastError(); return; }
        want = (int)dev;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (prev != want) (void)hipSetDevice(want);
    }
    ~Ru3dDeviceGuard() {
        if (prev >= 0 && want >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};

static inline int64_t nvox(const ru3d_tensor* t) { return (int64_t)t->n * t->d * t->h * t->w; }

static inline int tensor_ok(const ru3d_tensor* t) {
    return t && t->ptr && t->n > 0 && t->d > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->ld >= t->c && t->cseg == 0;
}
// the few entry points that take the split (planar concat) layout: two or more dense segments of cseg channels
static inline int tensor_ok_split(const ru3d_tensor* t) {
    if (t && t->cseg == 0) return tensor_ok(t);
    return t && t->ptr && t->n > 0 && t->d > 0 && t->h > 0 && t->w > 0 && t->c > 0 && t->cseg > 0 && (t->c % t->cseg) == 0 &&
           t->ld >= t->cseg && t->seg_stride >= (int64_t)t->n * t->d * t->h * t->w * t->ld;
}

#define RU3D_REQUIRE(cond, ...)                        \
    do {                                               \
        if (!(cond)) return ru3d_fail(-1, __VA_ARGS__); \
    } while (0)

// --------------------------------------------------------------------------- device helpers
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// vector of VEC elements of T, loaded/stored as one access when aligned
template <typename T, int VEC> struct vec_t {
    T v[VEC];
};
template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, float (&out)[VEC]) {
    typedef __attribute__((ext_vector_type(VEC))) T VT;
    VT x = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < VEC; i++) out[i] = to_f32<T>(x[i]);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const float (&in)[VEC]) {
    typedef __attribute__((ext_vector_type(VEC))) T VT;
    VT x;
#pragma unroll
    for (int i = 0; i < VEC; i++) x[i] = from_f32<T>(in[i]);
    *reinterpret_cast<VT*>(p) = x;
}

// Statistics slabs (slab[(workgroup * 4 + wave)][n][cb][2] floats): a wave writes only the rows of the samples it meets, the
// fixed-order finalize reads every row.  Each wave clears its own N rows when the kernel starts (1 KB of stores, same
// wave -> program order with its later writes) - the callers' hipMemsetAsync per conv launch (23 a step, 5 us each) is gone.
__device__ __forceinline__ void ru3d_clear_own_slab_rows(float* slab, int64_t wave_row, int N, int row_floats) {
    float* base = slab + wave_row * (int64_t)N * row_floats;
    for (int i = threadIdx.x & 63; i < N * row_floats; i += 64) base[i] = 0.f;
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the zeros have left this wave before anything else does
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float lrelu_f(float x, float slope) { return x > 0.f ? x : x * slope; }
