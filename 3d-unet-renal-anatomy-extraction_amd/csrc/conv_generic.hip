// Generic (any channel count, fp32 or bf16 storage, fp32 accumulate) direct convolution kernels.
//
// These are the parity-mode kernels (fp32 storage, fixed summation order) and the fallback for
// shapes the MFMA implicit-GEMM kernels (conv_mfma.hip) do not take (Cin = 1 stem, Cout = 2..4
// head, F = 30 channel plans).  One kernel covers both data-movement forms the U-Net needs:
//   gather form      y[o]  = sum_tap x[o*s + tap - p] . W[tap]      (Conv3d fwd; s=1 dgrad; convT dgrad)
//   transposed form  y[o]  = sum_{tap : (o + p - tap) % s == 0} x[(o + p - tap)/s] . W[tap]
//                                                                  (ConvTranspose3d fwd; s=2 conv dgrad)
// Packed weight layout for these kernels: W[tap][cin][cout_pad] in the storage dtype, cout_pad =
// cout rounded up to the per-thread output-channel tile so the inner loop needs no masking.
#include "common.h"
#include "conv.h"

namespace RU3D_NS {

template <typename T, typename TO, int CO_T, int VEC, bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv_generic_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const TO* __restrict__ res, TO* __restrict__ y,
                                                           ConvGeom g) {
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t vo = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vo >= total) return;
    const int co0 = blockIdx.y * CO_T;
    int ow = (int)(vo % g.Wo);
    int64_t t = vo / g.Wo;
    int oh = (int)(t % g.Ho);
    t /= g.Ho;
    int od = (int)(t % g.Do);
    int n = (int)(t / g.Do);

    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; j++) acc[j] = 0.f;

    const int k = g.k, s = g.stride, p = g.pad;
    for (int kd = 0; kd < k; kd++) {
        int id;
        if (!TRANSPOSED) {
            id = od * s + kd - p;
            if (id < 0 || id >= g.Di) continue;
        } else {
            int q = od + p - kd;
            if (q < 0 || (q % s) != 0) continue;
            id = q / s;
            if (id >= g.Di) continue;
        }
        for (int kh = 0; kh < k; kh++) {
            int ih;
            if (!TRANSPOSED) {
                ih = oh * s + kh - p;
                if (ih < 0 || ih >= g.Hi) continue;
            } else {
                int q = oh + p - kh;
                if (q < 0 || (q % s) != 0) continue;
                ih = q / s;
                if (ih >= g.Hi) continue;
            }
            for (int kw = 0; kw < k; kw++) {
                int iw;
                if (!TRANSPOSED) {
                    iw = ow * s + kw - p;
                    if (iw < 0 || iw >= g.Wi) continue;
                } else {
                    int q = ow + p - kw;
                    if (q < 0 || (q % s) != 0) continue;
                    iw = q / s;
                    if (iw >= g.Wi) continue;
                }
                const int tap = (kd * k + kh) * k + kw;
                const T* xp = x + ((((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx;
                const int wtap = g.flip ? (k * k * k - 1 - tap) : tap;
                const T* wp = w + (int64_t)wtap * g.Cin * g.CoutPad + co0;
                for (int ci = 0; ci < g.Cin; ci += VEC) {
                    float xv[VEC];
                    load_vec<T, VEC>(xp + ci, xv);
#pragma unroll
                    for (int u = 0; u < VEC; u++) {
                        const T* wr = wp + (int64_t)(ci + u) * g.CoutPad;
#pragma unroll
                        for (int j = 0; j < CO_T; j++) acc[j] = fmaf(xv[u], to_f32<T>(wr[j]), acc[j]);
                    }
                }
            }
        }
    }
    const bool far = g.zero_far && (od == g.Do - 1 || oh == g.Ho - 1 || ow == g.Wo - 1);
#pragma unroll
    for (int j = 0; j < CO_T; j++) {
        const int co = co0 + j;
        if (co < g.Cout) {
            float v = acc[j];
            if (bias) v += bias[co];
            if (far) v = 0.f;
            if (res) v += to_f32<TO>(res[vo * g.ldr + co]);
            y[vo * g.ldy + co] = from_f32<TO>(v);
        }
    }
}

int generic_cot(int cout) { return cout <= 4 ? 4 : (cout % 16 == 0 ? 16 : 8); }
int generic_cout_pad(int cout) {
    int t = generic_cot(cout);
    return (cout + t - 1) / t * t;
}

template <typename T, typename TO, int CO_T, int VEC>
static int launch_generic2(const void* x, const void* w, const float* bias, const void* res, void* y,
                           const ConvGeom& g, hipStream_t st) {
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    dim3 grid((unsigned)((total + 255) / 256), (unsigned)(g.CoutPad / CO_T));
    if (g.transposed)
        hipLaunchKernelGGL((conv_generic_kernel<T, TO, CO_T, VEC, true>), grid, dim3(256), 0, st, (const T*)x,
                           (const T*)w, bias, (const TO*)res, (TO*)y, g);
    else
        hipLaunchKernelGGL((conv_generic_kernel<T, TO, CO_T, VEC, false>), grid, dim3(256), 0, st, (const T*)x,
                           (const T*)w, bias, (const TO*)res, (TO*)y, g);
    return ru3d_check_launch("conv_generic");
}

template <typename T, typename TO>
static int launch_generic1(const void* x, const void* w, const float* bias, const void* res, void* y,
                           const ConvGeom& g, hipStream_t st) {
    const bool v4 = (g.Cin % 4 == 0) && (g.ldx % 4 == 0) && (((uintptr_t)x) % (4 * sizeof(T)) == 0);
    const int cot = generic_cot(g.Cout);
#define GO(CO, V) return launch_generic2<T, TO, CO, V>(x, w, bias, res, y, g, st)
    if (cot == 4) { if (v4) GO(4, 4); else GO(4, 1); }
    if (cot == 8) { if (v4) GO(8, 4); else GO(8, 1); }
    if (v4) GO(16, 4); else GO(16, 1);
#undef GO
}

int conv_generic_launch(const void* x, const void* w, const float* bias, const void* res, void* y,
                        const ConvGeom& g, int dtype, int y_dtype, hipStream_t st) {
    if (dtype == RU3D_F32 && y_dtype == RU3D_F32) return launch_generic1<float, float>(x, w, bias, res, y, g, st);
    if (dtype == RU3D_BF16 && y_dtype == RU3D_BF16) return launch_generic1<bf16, bf16>(x, w, bias, res, y, g, st);
    if (dtype == RU3D_BF16 && y_dtype == RU3D_F32) return launch_generic1<bf16, float>(x, w, bias, res, y, g, st);
    return ru3d_fail(-1, "conv_generic: unsupported dtype pair (%d -> %d)", dtype, y_dtype);
}

// --------------------------------------------------------------------------- weight packing
// dst[tap'][ci'][co'] (cout padded) <- src[co'*s_o + ci'*s_i + tap], tap flipped for the s=1 dgrad.
template <typename T>
__global__ void pack_generic_kernel(const float* __restrict__ src, T* __restrict__ dst, int cin, int cout,
                                    int cout_pad, int taps, int64_t s_o, int64_t s_i, int flip) {
    const int64_t total = (int64_t)taps * cin * cout_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int co = (int)(i % cout_pad);
        int64_t t = i / cout_pad;
        int ci = (int)(t % cin);
        int tap = (int)(t / cin);
        float v = 0.f;
        if (co < cout) {
            int st = flip ? (taps - 1 - tap) : tap;
            v = src[co * s_o + ci * s_i + st];
        }
        dst[i] = from_f32<T>(v);
    }
}

int pack_generic_launch(const float* src, void* dst, int cin, int cout, int taps, int64_t s_o, int64_t s_i,
                        int flip, int dtype, hipStream_t st) {
    const int cp = generic_cout_pad(cout);
    const int64_t total = (int64_t)taps * cin * cp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(pack_generic_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, cin, cout, cp,
                           taps, s_o, s_i, flip);
    else
        hipLaunchKernelGGL(pack_generic_kernel<bf16>, dim3(blocks), dim3(256), 0, st, src, (bf16*)dst, cin, cout, cp,
                           taps, s_o, s_i, flip);
    return ru3d_check_launch("pack_generic");
}

// Batched packing: blockIdx.y selects the weight, both packed layouts are produced by the same kernel
// (MFMA fragment order: see conv_mfma.hip; generic: [tap][cin][cout_pad]).
//
// MFMA items whose source has one of the two contiguous-weight shapes (s_i == taps: [co][ci][tap], the forward
// roles; s_o == taps: [ci][co][tap] seen from the packed side, the input-gradient roles) are transposed through
// LDS: a block owns one (32 co) x (16 ci) fragment column for every tap, reads its source rows as contiguous runs
// (16*taps or 32*taps floats) and writes one 1-KiB fragment per tap.  The element-wise gather it replaces touched a
// different 64-byte line with every 4-byte read.
#define PACK_TILE_FLOATS (32 * (16 * 27 + 1))   // >= 16 * (32 * 27 + 1)
// Channel padding (PackOne::*_seg): a channel dimension of the packed weight may be made of segments of `real`
// source channels each padded with zeros to `pad` packed channels (F = 30 widths on the MFMA kernels: 30 -> 32,
// and the decoder's concat input 30 | 30 -> 32 | 32).  pad == 0: the dimension is packed as it is.
__device__ __forceinline__ int pack_src_index(int p, int real, int pad) {
    if (pad == 0) return p;
    const int seg = p / pad, off = p - seg * pad;
    return off < real ? seg * real + off : -1;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(PackBatch b) {
    __shared__ float tile[PACK_TILE_FLOATS];   // 32 rows of 16*taps(+1) floats, or 16 rows of 32*taps(+1)
    const PackOne& p = b.item[blockIdx.y];
    T* dst = (T*)p.dst;
    const int taps = p.taps;
    if (p.mfma == 3) return;   // packed by pack_pair_kernel
    if (p.mfma == 2) {   // bias vector: fp32, zero-padded per segment
        float* out = (float*)p.dst;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < p.cout; i += gridDim.x * 256) {
            const int sc = pack_src_index(i, p.co_real, p.co_pad);
            out[i] = sc >= 0 ? p.src[sc] : 0.f;
        }
        return;
    }
    const bool rows_co = (p.s_i == taps), rows_ci = (p.s_o == taps);
    if (p.mfma && sizeof(T) == 2 && taps <= 27 && (rows_co || rows_ci)) {
        const int KS = p.cin / 16, NTT = p.cout / 32;
        const int nrows = rows_co ? 32 : 16, rowlen = (rows_co ? 16 : 32) * taps, pitch = rowlen + 1;
        for (int unit = blockIdx.x; unit < KS * NTT; unit += gridDim.x) {
            const int ks = unit / NTT, nt = unit % NTT;
            // a 32-channel (16-channel) block never straddles a padded segment (segments are multiples of 32):
            // first source channel of the block and the number of real channels in it, per dimension
            const int co_s = pack_src_index(nt * 32, p.co_real, p.co_pad);
            const int ci_s = pack_src_index(ks * 16, p.ci_real, p.ci_pad);
            int co_n = 32, ci_n = 16;
            if (p.co_pad) co_n = co_s < 0 ? 0 : min(32, p.co_real - (nt * 32) % p.co_pad);
            if (p.ci_pad) ci_n = ci_s < 0 ? 0 : min(16, p.ci_real - (ks * 16) % p.ci_pad);
            const float* base = p.src + (int64_t)(co_s < 0 ? 0 : co_s) * p.s_o + (int64_t)(ci_s < 0 ? 0 : ci_s) * p.s_i;
            const int64_t row_stride = rows_co ? p.s_o : p.s_i;
            const int row_n = rows_co ? co_n : ci_n, col_n = (rows_co ? ci_n : co_n) * taps;
            __syncthreads();
            for (int e = threadIdx.x; e < nrows * rowlen; e += 256) {
                const int r = e / rowlen, j = e - r * rowlen;
                tile[r * pitch + j] = (r < row_n && j < col_n) ? base[(int64_t)r * row_stride + j] : 0.f;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < taps * 64; e += 256) {
                const int tap = e >> 6, lane = e & 63;
                const int co = lane & 31, ci0 = 8 * (lane >> 5);
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int ci = ci0 + j;
                    const float f = rows_co ? tile[co * pitch + ci * taps + tap] : tile[ci * pitch + co * taps + tap];
                    v[j] = (bf16)f;
                }
                *reinterpret_cast<bf16x8*>((bf16*)p.dst + ((((int64_t)tap * KS + ks) * NTT + nt) * 64 + lane) * 8) = v;
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.total; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        if (p.mfma) {
            const int KS = p.cin / 16, NTT = p.cout / 32;
            const int j = (int)(i & 7);
            const int lane = (int)((i >> 3) & 63);
            int64_t t = i >> 9;
            const int nt = (int)(t % NTT);
            t /= NTT;
            const int ks = (int)(t % KS);
            const int tap = (int)(t / KS);
            const int co = pack_src_index(nt * 32 + (lane & 31), p.co_real, p.co_pad);
            const int ci = pack_src_index(ks * 16 + 8 * (lane >> 5) + j, p.ci_real, p.ci_pad);
            if (co >= 0 && ci >= 0) v = p.src[co * p.s_o + ci * p.s_i + tap];
        } else {
            const int co = (int)(i % p.cout_pad);
            const int64_t t = i / p.cout_pad;
            const int ci = pack_src_index((int)(t % p.cin), p.ci_real, p.ci_pad);
            const int tap = (int)(t / p.cin);
            const int cs = co < p.cout ? pack_src_index(co, p.co_real, p.co_pad) : -1;
            if (cs >= 0 && ci >= 0) v = p.src[cs * p.s_o + ci * p.s_i + tap];
        }
        dst[i] = from_f32<T>(v);
    }
}

// ---- both MFMA roles of a weight from one read of the source (see PackPair in conv.h).  A workgroup owns one
// (32 a) x (32 b) x taps block: it reads 32 contiguous runs of 32 * taps floats, keeps the block in LDS as 16-bit
// [tap][a][b] (pitch 40: the 16-byte reads of the a-major form and the 2-byte strided reads of the b-major form both
// spread over the banks) and writes, per tap, two 1-KiB fragments of each packed form.  The per-role kernel above read
// the source once per role and ran at 1.5 TB/s; the weights are 0.4 GB of fp32 per step at BASELINE config 2.
#define PP_PITCH 40
#define PP_PLANE (32 * PP_PITCH + 8)   // per-tap plane: 644 dwords, so the tap-fastest fill spreads over the banks
template <int TAPS>   // compile-time (27 or 1): the fill loop's index arithmetic divides by it
__global__ __launch_bounds__(256) void pack_pair_kernel(PackPairBatch pb) {
    __shared__ __attribute__((aligned(16))) bf16 tile[TAPS * PP_PLANE];
    const PackPair& p = pb.item[blockIdx.y];
    if (p.taps != TAPS) return;
    constexpr int taps = TAPS;
    const int TA = p.adim / 32, TB = p.bdim / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int unit = blockIdx.x; unit < TA * TB; unit += gridDim.x) {
        const int ta = unit / TB, tb = unit % TB;
        // real source channels covered by this block (a 32-channel block never straddles a padded segment)
        const int a_s = pack_src_index(ta * 32, p.a_real, p.a_pad), b_s = pack_src_index(tb * 32, p.b_real, p.b_pad);
        int a_n = 32, b_n = 32;
        if (p.a_pad) a_n = a_s < 0 ? 0 : min(32, p.a_real - (ta * 32) % p.a_pad);
        if (p.b_pad) b_n = b_s < 0 ? 0 : min(32, p.b_real - (tb * 32) % p.b_pad);
        const float* base = p.src + (int64_t)(a_s < 0 ? 0 : a_s) * p.s_a + (int64_t)(b_s < 0 ? 0 : b_s) * taps;
        constexpr int run = 32 * taps;
        const int run_n = b_n * taps;
        __syncthreads();
        const bool vec_ok = ((p.s_a | run_n) & 3) == 0 && ((((uintptr_t)base) & 15) == 0);
        if (vec_ok) {
            // 16-byte loads, eight in flight per thread (128 B): with 4-byte loads and four in flight a CU had 8 KB on
            // the wire and the kernel ran at HBM latency, not bandwidth
            constexpr int run4 = run / 4;
            for (int e0 = threadIdx.x; e0 < 32 * run4; e0 += 8 * 256) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u * 256;
                    const int a = e / run4, j4 = e - a * run4;
                    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    v[u] = (e < 32 * run4 && a < a_n && 4 * j4 < run_n)
                               ? *reinterpret_cast<const f32x4*>(base + (int64_t)a * p.s_a + 4 * j4) : z4;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u * 256;
                    if (e < 32 * run4) {
                        const int a = e / run4, j4 = e - a * run4;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int j = 4 * j4 + q;
                            const int b = j / taps, tap = j - b * taps;
                            tile[tap * PP_PLANE + a * PP_PITCH + b] = (bf16)v[u][q];
                        }
                    }
                }
            }
        } else {
            // four loads in flight per thread before the first LDS store
            for (int e0 = threadIdx.x; e0 < 32 * run; e0 += 4 * 256) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256;
                    const int a = e / run, j = e - a * run;
                    v[u] = (e < 32 * run && a < a_n && j < run_n) ? base[(int64_t)a * p.s_a + j] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u * 256;
                    const int a = e / run, j = e - a * run;
                    const int b = j / taps, tap = j - b * taps;
                    if (e < 32 * run) tile[tap * PP_PLANE + a * PP_PITCH + b] = (bf16)v[u];
                }
            }
        }
        __syncthreads();
        // a-major form: fragment (tap, ks = 2 tb + kc, nt = ta): lane -> a = lane & 31, 8 consecutive b from 16 kc + 8 (lane >> 5)
        if (p.dst_a) {
            const int KS = p.bdim / 16, NTT = TA;
            for (int f = wave; f < taps * 2; f += 4) {
                const int tap = f >> 1, kc = f & 1;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(&tile[tap * PP_PLANE + (lane & 31) * PP_PITCH + 16 * kc + 8 * (lane >> 5)]);
                *reinterpret_cast<bf16x8*>((bf16*)p.dst_a + ((((int64_t)tap * KS + 2 * tb + kc) * NTT + ta) * 64 + lane) * 8) = v;
            }
        }
        // b-major form: fragment (tap, ks = 2 ta + kc, nt = tb): lane -> b = lane & 31, 8 consecutive a
        if (p.dst_b) {
            const int KS = p.adim / 16, NTT = TB;
            for (int f = wave; f < taps * 2; f += 4) {
                const int tap = f >> 1, kc = f & 1;
                const int a0 = 16 * kc + 8 * (lane >> 5);
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = tile[tap * PP_PLANE + (a0 + j) * PP_PITCH + (lane & 31)];
                *reinterpret_cast<bf16x8*>((bf16*)p.dst_b + ((((int64_t)tap * KS + 2 * ta + kc) * NTT + tb) * 64 + lane) * 8) = v;
            }
        }
    }
}

int pack_pair_launch(const PackPairBatch& b, hipStream_t st) {
    int64_t mx = 1;
    for (int i = 0; i < b.count; i++) {
        const int64_t units = (int64_t)(b.item[i].adim / 32) * (b.item[i].bdim / 32);
        mx = units > mx ? units : mx;
    }
    if (mx > 1024) mx = 1024;
    bool t27 = false, t1 = false;
    for (int i = 0; i < b.count; i++) {
        t27 = t27 || b.item[i].taps == 27;
        t1 = t1 || b.item[i].taps == 1;
    }
    if (t27) hipLaunchKernelGGL(pack_pair_kernel<27>, dim3((unsigned)mx, b.count), dim3(256), 0, st, b);
    if (t1) hipLaunchKernelGGL(pack_pair_kernel<1>, dim3((unsigned)mx, b.count), dim3(256), 0, st, b);
    return ru3d_check_launch("pack_pair");
}

// dst[co][ci][tap] = src[co_p][ci_p][tap]: drops the zero rows / columns of a weight gradient computed on padded channels
__global__ __launch_bounds__(256) void unpad_weight_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int cout, int cin, int taps, int co_real, int co_pad,
                                                           int ci_real, int ci_pad, int cin_p) {
    const int64_t total = (int64_t)cout * cin * taps;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int tap = (int)(i % taps);
        const int64_t t = i / taps;
        const int ci = (int)(t % cin), co = (int)(t / cin);
        const int cop = co_pad ? (co / co_real) * co_pad + co % co_real : co;
        const int cip = ci_pad ? (ci / ci_real) * ci_pad + ci % ci_real : ci;
        dst[i] = src[((int64_t)cop * cin_p + cip) * taps + tap];
    }
}

int unpad_weight_launch(const float* src, float* dst, int cout, int cin, int taps, int co_real, int co_pad, int ci_real,
                        int ci_pad, int cin_p, hipStream_t st) {
    const int64_t total = (int64_t)cout * cin * taps;
    int64_t blocks = (total + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(unpad_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, cout, cin, taps, co_real,
                       co_pad, ci_real, ci_pad, cin_p);
    return ru3d_check_launch("unpad_weight_grad");
}

int pack_batch_launch(const PackBatch& b, int dtype, hipStream_t st) {
    int64_t mx = 1;
    for (int i = 0; i < b.count; i++) mx = b.item[i].total > mx ? b.item[i].total : mx;
    int64_t blocks = (mx + 1023) / 1024;   // ~4 elements per thread for the largest item
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks, b.count);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(pack_batch_kernel<float>, grid, dim3(256), 0, st, b);
    else
        hipLaunchKernelGGL(pack_batch_kernel<bf16>, grid, dim3(256), 0, st, b);
    return ru3d_check_launch("pack_batch");
}

// --------------------------------------------------------------------------- generic wgrad
// dW[tap][ci][co] = sum_pos x[s*pos + tap - p][ci] * dy[pos][co], pos over N*Do*Ho*Wo.
// grid = (chunks, taps, ci_tiles*co_tiles); block = 64x64 (ci x co) tile, 16 positions per LDS stage,
// each thread a 4x4 register tile.  Partials go to the workspace and are summed in a fixed order by
// wgrad_reduce_kernel (deterministic: no atomics).
#define WG_PB 16
template <typename T>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            float* __restrict__ part, WgradGeom g) {
    __shared__ float xs[WG_PB][64 + 4];
    __shared__ float ds[WG_PB][64 + 4];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, tap = blockIdx.y;
    const int co_tiles = (g.Cout + 63) / 64;
    const int ci0 = (blockIdx.z / co_tiles) * 64, co0 = (blockIdx.z % co_tiles) * 64;
    const int kw = tap % g.k, kh = (tap / g.k) % g.k, kd = tap / (g.k * g.k);
    const int ty = tid / 16, tx = tid % 16;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;

    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t p_begin = (int64_t)chunk * g.chunk_len;
    int64_t p_end = p_begin + g.chunk_len;
    if (p_end > P) p_end = P;

    for (int64_t p0 = p_begin; p0 < p_end; p0 += WG_PB) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int e = tid + r * 256;
            const int pl = e / 64, c = e % 64;
            const int64_t pos = p0 + pl;
            float xv = 0.f, dv = 0.f;
            if (pos < p_end) {
                int ow = (int)(pos % g.Wo);
                int64_t t = pos / g.Wo;
                int oh = (int)(t % g.Ho);
                t /= g.Ho;
                int od = (int)(t % g.Do);
                int n = (int)(t / g.Do);
                if (co0 + c < g.Cout) dv = to_f32<T>(dy[pos * g.lddy + co0 + c]);
                const int id = od * g.stride + kd - g.pad, ih = oh * g.stride + kh - g.pad,
                          iw = ow * g.stride + kw - g.pad;
                if (ci0 + c < g.Cin && id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                    xv = to_f32<T>(x[((((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx + ci0 + c]);
            }
            xs[pl][c] = xv;
            ds[pl][c] = dv;
        }
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < WG_PB; pl++) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = xs[pl][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = ds[pl][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* pp = part + ((int64_t)chunk * g.taps + tap) * g.Cin * g.Cout;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int ci = ci0 + ty * 4 + i;
        if (ci >= g.Cin) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int co = co0 + tx * 4 + j;
            if (co < g.Cout) pp[(int64_t)ci * g.Cout + co] = acc[i][j];
        }
    }
}

// dw[co*s_o + ci*s_i + tap] = sum_chunk part[chunk][tap][ci][co]
// block = QX lanes x (4 consecutive outputs, one 16-byte load) x KY = 256 / QX chunk lanes; each lane keeps 4 independent
// slab loads in flight; the KY lane sums are combined in a fixed order -> deterministic.  QX = 64 / KY = 4 for the usual
// case; a small output under thousands of slabs (the 1x1x1 skip convs of the large levels: 2 K outputs x 2048 slabs ran
// 45 us as 8 blocks with 128 dependent rounds each) takes QX = 8 / KY = 32: eight times the blocks, an eighth of the rounds.
template <int QX>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           int chunks, int taps, int cin, int cout, int64_t s_o,
                                                           int64_t s_i) {
    constexpr int KY = 256 / QX;
    __shared__ f32x4 sh[KY][QX];
    const int64_t total = (int64_t)taps * cin * cout;   // multiple of 4 is NOT required: tail handled scalar
    const int ox = threadIdx.x % QX, ky = threadIdx.x / QX;
    const int64_t i0 = ((int64_t)blockIdx.x * QX + ox) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const bool full = (i0 + 3 < total) && ((total & 3) == 0);
    if (full) {
        f32x4 s0 = s, s1 = s, s2 = s, s3 = s;
        int c = ky;
        for (; c + 3 * KY < chunks; c += 4 * KY) {
            s0 += *reinterpret_cast<const f32x4*>(part + (int64_t)c * total + i0);
            s1 += *reinterpret_cast<const f32x4*>(part + (int64_t)(c + KY) * total + i0);
            s2 += *reinterpret_cast<const f32x4*>(part + (int64_t)(c + 2 * KY) * total + i0);
            s3 += *reinterpret_cast<const f32x4*>(part + (int64_t)(c + 3 * KY) * total + i0);
        }
        for (; c < chunks; c += KY) s0 += *reinterpret_cast<const f32x4*>(part + (int64_t)c * total + i0);
        s = (s0 + s1) + (s2 + s3);
    } else {
        for (int j = 0; j < 4; j++)
            if (i0 + j < total)
                for (int c = ky; c < chunks; c += KY) s[j] += part[(int64_t)c * total + i0 + j];
    }
    sh[ky][ox] = s;
    __syncthreads();
    if (ky == 0) {
        s = sh[0][ox];
#pragma unroll
        for (int k = 1; k < KY; k++) s += sh[k][ox];
        for (int j = 0; j < 4; j++) {
            const int64_t i = i0 + j;
            if (i >= total) break;
            const int co = (int)(i % cout);
            const int64_t t = i / cout;
            const int ci = (int)(t % cin);
            const int tap = (int)(t / cin);
            dw[co * s_o + ci * s_i + tap] = s[j];
        }
    }
}

// The same sum for the wide layers (Cin*Cout >= 128*128: few slabs, megabytes each), where the write side
// matters: slabs are [tap][ci][co] but dw is [co][ci][tap] (Conv3d) or [ci][co][tap] (ConvTranspose3d), so the
// kernel above stores 4 bytes per 108-byte stride.  Here a block owns all taps of a (4 ci) x (32 co) tile, sums
// the slabs in slab order (deterministic) with 128-byte row reads, transposes through LDS and writes runs of
// 4*taps (CO_MAJOR, Conv3d) or 32*taps (ConvTranspose3d) consecutive floats.
template <bool CO_MAJOR>
__device__ __forceinline__ void wgrad_reduce_tiled_body(const float* __restrict__ part, float* __restrict__ dw, int chunks,
                                                        int taps, int cin, int cout, int64_t s_o, int64_t s_i, int bid,
                                                        float* tile) {
    const int cot = bid % (cout / 32), cit = bid / (cout / 32);
    const int ci0 = cit * 4, co0 = cot * 32;
    const int64_t total = (int64_t)taps * cin * cout;
    const int nquads = taps * 32;   // (tap, ci_l, co4)
    const int pitch_co = 4 * taps + 1;
    // a thread's (up to) four quads side by side: 16 row reads in flight instead of 4 - with 128 blocks of 221 KB each at
    // 128 x 128 channels the kernel ran at the latency of its loads (29 us for 28 MB that sit in L2).  Per quad the same
    // four partial sums in the same order as before (c mod 4, then (s0 + s1) + (s2 + s3)): bit-identical results.
    constexpr int QPT = 4;      // 27 taps x 32 quads = 864 <= 4 x 256
    const float* src[QPT];
    bool live[QPT];
    f32x4 acc[QPT][4];
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        const int qid = threadIdx.x + 256 * k;
        live[k] = qid < nquads;
        const int qq = live[k] ? qid : 0;
        const int tap = qq >> 5, ci_l = (qq >> 3) & 3, co4 = qq & 7;
        src[k] = part + ((int64_t)tap * cin + ci0 + ci_l) * cout + co0 + co4 * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int c = 0;
    for (; c + 3 < chunks; c += 4) {
        f32x4 v[QPT][4];
#pragma unroll
        for (int k = 0; k < QPT; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) v[k][j] = *reinterpret_cast<const f32x4*>(src[k] + (int64_t)(c + j) * total);
#pragma unroll
        for (int k = 0; k < QPT; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[k][j] += v[k][j];
    }
    for (; c < chunks; c++) {
#pragma unroll
        for (int k = 0; k < QPT; k++) acc[k][0] += *reinterpret_cast<const f32x4*>(src[k] + (int64_t)c * total);
    }
#pragma unroll
    for (int k = 0; k < QPT; k++) {
        if (!live[k]) continue;
        const int qid = threadIdx.x + 256 * k;
        const int tap = qid >> 5, ci_l = (qid >> 3) & 3, co4 = qid & 7;
        const f32x4 s = (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int co = co4 * 4 + e;
            if (CO_MAJOR) tile[co * pitch_co + ci_l * taps + tap] = s[e];
            else tile[ci_l * (32 * taps + 1) + co * taps + tap] = s[e];
        }
    }
    __syncthreads();
    if (CO_MAJOR) {
        const int run = 4 * taps;     // dw[(co0+co)*s_o + ci0*taps + j], j < run
        for (int e = threadIdx.x; e < 32 * run; e += 256) {
            const int co = e / run, j = e - co * run;
            dw[(int64_t)(co0 + co) * s_o + (int64_t)ci0 * s_i + j] = tile[co * pitch_co + j];
        }
    } else {
        const int run = 32 * taps;    // dw[(ci0+ci)*s_i + co0*taps + j], j < run
        for (int e = threadIdx.x; e < 4 * run; e += 256) {
            const int ci = e / run, j = e - ci * run;
            dw[(int64_t)(ci0 + ci) * s_i + (int64_t)co0 * s_o + j] = tile[ci * (run + 1) + j];
        }
    }
}

template <bool CO_MAJOR>
__global__ __launch_bounds__(256) void wgrad_reduce_tiled_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                 int chunks, int taps, int cin, int cout, int64_t s_o,
                                                                 int64_t s_i) {
    __shared__ float tile[32 * (4 * 27 + 1)];
    wgrad_reduce_tiled_body<CO_MAJOR>(part, dw, chunks, taps, cin, cout, s_o, s_i, blockIdx.x, tile);
}

int wgrad_reduce_launch(const float* part, float* dw, int chunks, int taps, int cin, int cout, int64_t s_o, int64_t s_i,
                        hipStream_t st) {
    const int64_t total = (int64_t)taps * cin * cout;
    const bool tiled = (int64_t)cin * cout >= 128 * 128 && (cin % 4) == 0 && (cout % 32) == 0 && taps <= 27 &&
                       (s_i == taps || s_o == taps);
    if (tiled) {
        const int64_t blocks = (int64_t)(cin / 4) * (cout / 32);
        if (blocks > 0x3fffffff) return ru3d_fail(-1, "wgrad_reduce: grid too large");
        if (s_i == taps)
            hipLaunchKernelGGL(wgrad_reduce_tiled_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, part, dw, chunks, taps,
                               cin, cout, s_o, s_i);
        else
            hipLaunchKernelGGL(wgrad_reduce_tiled_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, part, dw, chunks, taps,
                               cin, cout, s_o, s_i);
        return ru3d_check_launch("wgrad_reduce_tiled");
    }
    // few outputs under many slabs: more chunk lanes per output (the launch is latency, not bandwidth)
    const bool deep = chunks >= 128 && (total + 255) / 256 < 512;
    const int qx = deep ? 8 : 64;
    const int64_t blocks = (total + 4 * qx - 1) / (4 * qx);
    if (blocks > 0x3fffffff) return ru3d_fail(-1, "wgrad_reduce: grid too large");
    if (deep)
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, part, dw, chunks, taps, cin, cout,
                           s_o, s_i);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, part, dw, chunks, taps, cin, cout,
                           s_o, s_i);
    return ru3d_check_launch("wgrad_reduce");
}

int wgrad_generic_chunks(const WgradGeom& g) {
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int tiles = ((g.Cin + 63) / 64) * ((g.Cout + 63) / 64);
    int64_t want = 2048 / ((int64_t)g.taps * tiles);
    if (want < 1) want = 1;
    int64_t maxc = (P + WG_PB * 16 - 1) / (WG_PB * 16);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

size_t wgrad_generic_ws_bytes(const WgradGeom& g) {
    return (size_t)wgrad_generic_chunks(g) * g.taps * g.Cin * g.Cout * sizeof(float);
}

int wgrad_generic_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, WgradGeom g, int dtype,
                         hipStream_t st) {
    const int chunks = wgrad_generic_chunks(g);
    const size_t need = wgrad_generic_ws_bytes(g);
    if (!ws || ws_bytes < need) return ru3d_fail(-1, "wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    int64_t len = (P + chunks - 1) / chunks;
    len = (len + WG_PB - 1) / WG_PB * WG_PB;
    g.chunk_len = len;
    const int tiles = ((g.Cin + 63) / 64) * ((g.Cout + 63) / 64);
    dim3 grid(chunks, g.taps, tiles);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(wgrad_generic_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)dy,
                           (float*)ws, g);
    else
        hipLaunchKernelGGL(wgrad_generic_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy,
                           (float*)ws, g);
    int rc = ru3d_check_launch("wgrad_generic");
    if (rc) return rc;
    return wgrad_reduce_launch((const float*)ws, dw, chunks, g.taps, g.Cin, g.Cout, g.s_o, g.s_i, st);
}

}  // namespace RU3D_NS
