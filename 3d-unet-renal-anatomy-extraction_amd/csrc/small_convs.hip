// Dedicated kernels for the two "thin" ends of the U-Net, where one side of the conv has 1..4 channels and
// an implicit GEMM would be almost all padding.  Both are HBM-bound by construction:
//   stem  : Conv3d(1 -> F, k3, p1)          forward and weight gradient  (reference network.py:541-544)
//   head  : Conv3d(F -> classes, k1)        weight gradient              (reference network.py:545-547)
// (The head forward / input gradient and the stem forward for other shapes stay on conv_generic.hip.)
#include "common.h"
#include "conv.h"

namespace RU3D_NS {

// --------------------------------------------------------------------------- stem forward (Cin == 1)
// thread = (voxel, group of 8 couts): 27 scalar x loads (neighbouring lanes share them through L1), 27x8 FMA,
// one 16-byte (bf16) / 32-byte (f32) store; consecutive lanes write consecutive channel groups of consecutive
// voxels -> fully coalesced NDHWC stores.  Weights [27][CoutPad] (generic packing) are staged in LDS as fp32.
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, ConvGeom g) {
    extern __shared__ float wl[];   // [27][Cout] + [Cout]
    const int Cout = g.Cout;
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) wl[i] = to_f32<T>(w[(i / Cout) * g.CoutPad + (i % Cout)]);
    for (int i = threadIdx.x; i < Cout; i += 256) wl[27 * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int groups = Cout / 8;
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo * groups;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cg = (int)(gid % groups);
    const int64_t vo = gid / groups;
    const int ow = (int)(vo % g.Wo);
    int64_t t = vo / g.Wo;
    const int oh = (int)(t % g.Ho);
    t /= g.Ho;
    const int od = (int)(t % g.Do);
    const int n = (int)(t / g.Do);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = wl[27 * Cout + cg * 8 + j];
#pragma unroll
    for (int kd = 0; kd < 3; kd++) {
        const int id = od + kd - 1;
        if (id < 0 || id >= g.Di) continue;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            const int ih = oh + kh - 1;
            if (ih < 0 || ih >= g.Hi) continue;
            const T* row = x + (((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi * g.ldx;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int iw = ow + kw - 1;
                if (iw < 0 || iw >= g.Wi) continue;
                const float xv = to_f32<T>(row[(int64_t)iw * g.ldx]);
                const float* wr = wl + ((kd * 3 + kh) * 3 + kw) * Cout + cg * 8;
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = fmaf(xv, wr[j], acc[j]);
            }
        }
    }
    store_vec<T, 8>(y + vo * g.ldy + cg * 8, acc);
}

// The same conv on MFMA for bf16 with H % 8 == 0, W % 32 == 0, Cout % 32 == 0: D[co][pos] = W[co][tap] x patch[tap][pos],
// K = 27 taps padded to 32 (two k-steps).  A persistent workgroup walks (1 x 8 x 32)-position tiles: the 3 x 10 x 34
// input patch goes to LDS, every thread expands ONE position into its 32-tap row (im2col in LDS, 80-byte rows), each
// wave runs 4 MFMAs for its two 32-position rows and writes them through a wave-private patch as 16-byte NDHWC
// stores.  The VALU form above issues 27 two-byte loads per thread; this one is bound by the 2F bytes it writes.
#define STEM_FWD_BLOCKS 1792   /* 7 workgroups of 22.5 KB LDS per CU: the tile loop is a chain of barriers, residency hides it */
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                            const float* __restrict__ bias, bf16* __restrict__ y,
                                                            ConvGeom g, int tiles_h, int tiles_w, int ntiles) {
    __shared__ __attribute__((aligned(16))) bf16 xs[3 * 10 * 34 + 4];
    // im2col rows: 32 taps + pad.  A wave's 64 rows double as its epilogue patch (same pitch; a row is rewritten only
    // behind the MFMAs that read it, and only by the wave that owns it)
    __shared__ __attribute__((aligned(16))) bf16 pm[256 * 40];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * 32;
    const int h = lane >> 5;
    // weight fragments: row = co = lane & 31, k = 16 ks + 8 h + j = tap (taps 27..31 are zero)
    bf16x8 afrag[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int tap = ks * 16 + 8 * h + j;
            afrag[ks][j] = tap < 27 ? w[tap * g.CoutPad + co0 + (lane & 31)] : (bf16)0.f;
        }
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        bq[q] = bias ? *reinterpret_cast<const f32x4*>(bias + co0 + 8 * q + 4 * h) : z4;
    }
    bf16* patch = pm + wave * (64 * 40);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tiles_w) * 32;
        t /= tiles_w;
        const int h0 = (t % tiles_h) * 8;
        t /= tiles_h;
        const int d = t % g.Do;
        const int n = t / g.Do;
        __syncthreads();   // previous tile's im2col rows consumed
        for (int e = tid; e < 3 * 10 * 34; e += 256) {
            const int c = e % 34, r = (e / 34) % 10, kd = e / 340;
            const int id = d + kd - 1, ih = h0 + r - 1, iw = w0 + c - 1;
            bf16 v = (bf16)0.f;
            if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                v = x[((((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx];
            xs[e] = v;
        }
        __syncthreads();
        {   // im2col: thread = position f = (row tid >> 5, column tid & 31)
            const int hh = tid >> 5, ww = tid & 31;
            bf16x8 row[4];
#pragma unroll
            for (int tap = 0; tap < 32; tap++) {
                bf16 v = (bf16)0.f;
                if (tap < 27) v = xs[((tap / 9) * 10 + hh + (tap / 3) % 3) * 34 + ww + tap % 3];
                row[tap >> 3][tap & 7] = v;
            }
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<bf16x8*>(&pm[tid * 40 + i * 8]) = row[i];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; m++) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.f;
            const int f = (wave * 2 + m) * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(&pm[f * 40 + ks * 16 + 8 * h]);
                acc = RU3D_MFMA_32X32X16(afrag[ks], bfrag, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = acc[q * 4 + i] + bq[q][i];
                store_vec<bf16, 4>(patch + (m * 32 + (lane & 31)) * 40 + 8 * q + 4 * h, v);
            }
        }
        const int64_t vrow = (((int64_t)n * g.Do + d) * g.Ho + h0 + 2 * wave) * (int64_t)g.Wo + w0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 2) + 16 * r, part = lane & 3;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + row * 40 + part * 8);
            const int64_t vox = vrow + (int64_t)(row >> 5) * g.Wo + (row & 31);
            // border tiles of extents that are not multiples of 8 x 32 (160 x 160 x 80: the last column is half empty)
            if (h0 + 2 * wave + (row >> 5) < g.Ho && w0 + (row & 31) < g.Wo)
                *reinterpret_cast<bf16x8*>(y + vox * g.ldy + co0 + part * 8) = v;
        }
    }
}

static bool stem_fwd_mfma_ok(const ConvGeom& g, int dtype) {
    static const int mode = getenv("RU3D_STEM_MFMA") ? atoi(getenv("RU3D_STEM_MFMA")) : 1;
    return mode && dtype == RU3D_BF16 && (g.Cout % 32) == 0 && g.Do == g.Di && g.Ho == g.Hi && g.Wo == g.Wi &&
           g.pad == 1 && (g.ldy % 8) == 0;
}

bool stem_fwd_eligible(const ConvGeom& g, int dtype, int y_dtype, const void* res) {
    return g.Cin == 1 && g.k == 3 && g.stride == 1 && !g.transposed && !g.flip && !g.zero_far && !res &&
           dtype == y_dtype && (g.Cout % 8) == 0 && g.Cout <= 256 && (g.ldy % 8) == 0;
}

int stem_fwd_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, int dtype,
                    hipStream_t st) {
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo * (g.Cout / 8);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return ru3d_fail(-1, "stem_fwd: grid too large");
    if (stem_fwd_mfma_ok(g, dtype) && (((uintptr_t)y) % 16) == 0 && (!bias || (((uintptr_t)bias) % 16) == 0)) {
        const int tiles_h = (g.Ho + 7) / 8, tiles_w = (g.Wo + 31) / 32;
        const int64_t ntiles = (int64_t)g.N * g.Do * tiles_h * tiles_w;
        if (ntiles <= 0x7fffffff) {
            const int nb = (int)(ntiles < STEM_FWD_BLOCKS ? ntiles : STEM_FWD_BLOCKS);
            hipLaunchKernelGGL(stem_fwd_mfma_kernel, dim3(nb, g.Cout / 32), dim3(256), 0, st, (const bf16*)x,
                               (const bf16*)w, bias, (bf16*)y, g, tiles_h, tiles_w, (int)ntiles);
            return ru3d_check_launch("stem_fwd_mfma");
        }
    }
    const size_t lds = (size_t)28 * g.Cout * sizeof(float);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, (const float*)x,
                           (const float*)w, bias, (float*)y, g);
    else
        hipLaunchKernelGGL(stem_fwd_kernel<bf16>, dim3((unsigned)blocks), dim3(256), lds, st, (const bf16*)x,
                           (const bf16*)w, bias, (bf16*)y, g);
    return ru3d_check_launch("stem_fwd");
}

// --------------------------------------------------------------------------- head forward (k1, Cout <= 4)
// logits[v][co] = b[co] + sum_ci x[v][ci] w[ci][co].  G = Cin/VEC lanes share a voxel: each loads one 16-byte
// piece (fully coalesced), forms its partial dot products and the G partials are combined by xor-shuffles.
template <typename T, typename TO, int VEC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                       const float* __restrict__ bias, TO* __restrict__ y, ConvGeom g) {
    // Streaming kernel (reads |x|, writes 4-12 bytes per voxel): what it needs is bytes in flight.  A thread keeps its
    // VEC x 4 weights in registers and handles HEAD_U voxels - all their 16-byte loads are issued before the first FMA -
    // and after the xor-shuffles every lane of a voxel's group holds the sums, so lane cg stores channel cg: a wave's
    // store instruction covers a contiguous run of logits instead of one lane in G writing three scattered values.
    constexpr int HEAD_U = 4;
    const int G = g.Cin / VEC;      // lanes per voxel: power of two, <= 64
    const int cg = threadIdx.x % G;
    const int slots = 256 / G;      // voxels per workgroup and round
    float wr[VEC][4];
#pragma unroll
    for (int i = 0; i < VEC; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) wr[i][c] = to_f32<T>(w[(cg * VEC + i) * 4 + c]);   // generic packing: [cin][cout_pad = 4]
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t v0 = (int64_t)blockIdx.x * (slots * HEAD_U) + threadIdx.x / G;
    float xv[HEAD_U][VEC];
#pragma unroll
    for (int u = 0; u < HEAD_U; u++) {
        const int64_t vo = v0 + u * slots;
        if (vo < total) load_vec<T, VEC>(x + vo * g.ldx + cg * VEC, xv[u]);
        else
#pragma unroll
            for (int i = 0; i < VEC; i++) xv[u][i] = 0.f;
    }
    const float bsel = (bias && cg < g.Cout) ? bias[cg] : 0.f;
#pragma unroll
    for (int u = 0; u < HEAD_U; u++) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < VEC; i++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = fmaf(xv[u][i], wr[i][c], acc[c]);
        if (G == 4) {
            // a voxel's four lanes are a DPP quad: the two butterfly steps as quad permutes (same pairs, same sums as the
            // xor-shuffles, which are LDS-crossbar instructions: 8 per voxel and lane)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                acc[c] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[c]), 0xB1, 0xf, 0xf, true));   // lane ^ 1
                acc[c] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[c]), 0x4E, 0xf, 0xf, true));   // lane ^ 2
            }
        } else {
            for (int o = 1; o < G; o <<= 1) {
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] += __shfl_xor(acc[c], o, 64);
            }
        }
        const int64_t vo = v0 + u * slots;
        if (vo < total) {
            if (G >= g.Cout) {
                const float mine = cg == 0 ? acc[0] : (cg == 1 ? acc[1] : (cg == 2 ? acc[2] : acc[3]));
                if (cg < g.Cout) y[vo * g.ldy + cg] = from_f32<TO>(mine + bsel);
            } else if (cg == 0) {
                for (int c = 0; c < g.Cout; c++) y[vo * g.ldy + c] = from_f32<TO>(acc[c] + (bias ? bias[c] : 0.f));
            }
        }
    }
}

bool head_fwd_eligible(const ConvGeom& g, int dtype, const void* res) {
    const int vec = dtype == RU3D_BF16 ? 8 : 4;
    const int G = g.Cin / vec;
    return g.k == 1 && g.stride == 1 && !g.transposed && !g.zero_far && !res && g.Cout <= 4 && (g.Cin % vec) == 0 &&
           G >= 1 && G <= 64 && (G & (G - 1)) == 0 && (g.ldx % vec) == 0;
}

int head_fwd_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, int dtype,
                    int y_dtype, hipStream_t st) {
    const int vec = dtype == RU3D_BF16 ? 8 : 4;
    const int64_t per_block = (256 / (g.Cin / vec)) * 4;      // voxels per workgroup: slots x HEAD_U
    const int64_t blocks = ((int64_t)g.N * g.Do * g.Ho * g.Wo + per_block - 1) / per_block;
    if (blocks > 0x7fffffff) return ru3d_fail(-1, "head_fwd: grid too large");
    if (((uintptr_t)x) % 16) return ru3d_fail(-1, "head_fwd: x must be 16-byte aligned");
    const size_t lds = 0;
    dim3 grid((unsigned)blocks);
    if (dtype == RU3D_F32 && y_dtype == RU3D_F32)
        hipLaunchKernelGGL((head_fwd_kernel<float, float, 4>), grid, dim3(256), lds, st, (const float*)x,
                           (const float*)w, bias, (float*)y, g);
    else if (dtype == RU3D_BF16 && y_dtype == RU3D_F32)
        hipLaunchKernelGGL((head_fwd_kernel<bf16, float, 8>), grid, dim3(256), lds, st, (const bf16*)x, (const bf16*)w,
                           bias, (float*)y, g);
    else if (dtype == RU3D_BF16 && y_dtype == RU3D_BF16)
        hipLaunchKernelGGL((head_fwd_kernel<bf16, bf16, 8>), grid, dim3(256), lds, st, (const bf16*)x, (const bf16*)w,
                           bias, (bf16*)y, g);
    else
        return ru3d_fail(-1, "head_fwd: unsupported dtype pair");
    return ru3d_check_launch("head_fwd");
}

// --------------------------------------------------------------------------- head input gradient (k1, Cin' <= 4)
// dx[v][c] = sum_{co<=4} dy[v][co] w[co][c]: thread = (voxel, 8 output channels), one 16-byte store.
template <typename T>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const T* __restrict__ dy, const T* __restrict__ w,
                                                         T* __restrict__ dx, ConvGeom g) {
    extern __shared__ float wl[];   // [Cin'][Cout']
    for (int i = threadIdx.x; i < g.Cin * g.Cout; i += 256) wl[i] = to_f32<T>(w[(i / g.Cout) * g.CoutPad + (i % g.Cout)]);
    __syncthreads();
    const int groups = g.Cout / 8;
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo * groups;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cg = (int)(gid % groups);
    const int64_t vo = gid / groups;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int ci = 0; ci < g.Cin; ci++) {
        const float d = to_f32<T>(dy[vo * g.ldx + ci]);
        const float* wr = wl + ci * g.Cout + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = fmaf(d, wr[j], acc[j]);
    }
    store_vec<T, 8>(dx + vo * g.ldy + cg * 8, acc);
}

bool head_dgrad_eligible(const ConvGeom& g, int dtype, int y_dtype, const void* res) {
    return g.k == 1 && g.stride == 1 && !g.transposed && !g.zero_far && !res && dtype == y_dtype && g.Cin <= 4 &&
           (g.Cout % 8) == 0 && g.Cout <= 1024 && (g.ldy % 8) == 0;
}

int head_dgrad_launch(const void* dy, const void* w, void* dx, const ConvGeom& g, int dtype, hipStream_t st) {
    const int64_t total = (int64_t)g.N * g.Do * g.Ho * g.Wo * (g.Cout / 8);
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return ru3d_fail(-1, "head_dgrad: grid too large");
    const size_t lds = (size_t)g.Cin * g.Cout * sizeof(float);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(head_dgrad_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, (const float*)dy,
                           (const float*)w, (float*)dx, g);
    else
        hipLaunchKernelGGL(head_dgrad_kernel<bf16>, dim3((unsigned)blocks), dim3(256), lds, st, (const bf16*)dy,
                           (const bf16*)w, (bf16*)dx, g);
    return ru3d_check_launch("head_dgrad");
}

// --------------------------------------------------------------------------- stem weight gradient (Cin == 1)
// dW[tap][co] = sum_pos x[pos + tap] * dy[pos][co].  Workgroup = a run of STEM_CHUNK flat positions x 32 couts;
// wave = 8 couts; lanes stride over positions; the three kd planes are processed one after the other so a
// lane holds 9 taps x 8 couts of partial sums (72 VGPRs).  Lane sums are combined with a shuffle tree and
// one slab per workgroup is reduced afterwards in fixed order (deterministic).
#define STEM_CHUNK 2048
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ part, WgradGeom g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co0 = blockIdx.y * 32 + wave * 8;
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t p0 = (int64_t)blockIdx.x * STEM_CHUNK;
    int64_t p1 = p0 + STEM_CHUNK;
    if (p1 > P) p1 = P;
    for (int kd = 0; kd < 3; kd++) {
        float acc[9][8];
#pragma unroll
        for (int a = 0; a < 9; a++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[a][j] = 0.f;
        // coordinates of the lane's first position, then advanced by 64 per iteration (no division in the loop)
        int ow, oh, od, n;
        {
            const uint32_t pp = (uint32_t)(p0 + lane);
            ow = (int)(pp % (uint32_t)g.Wo);
            uint32_t t = pp / (uint32_t)g.Wo;
            oh = (int)(t % (uint32_t)g.Ho);
            t /= (uint32_t)g.Ho;
            od = (int)(t % (uint32_t)g.Do);
            n = (int)(t / (uint32_t)g.Do);
        }
        ow -= 64;
        for (int64_t p = p0 + lane; p < p1; p += 64) {
            ow += 64;
            while (ow >= g.Wo) {
                ow -= g.Wo;
                if (++oh == g.Ho) {
                    oh = 0;
                    if (++od == g.Do) {
                        od = 0;
                        ++n;
                    }
                }
            }
            const int id = od + kd - 1;
            if (id < 0 || id >= g.Di) continue;
            float dv[8];
            load_vec<T, 8>(dy + p * g.lddy + co0, dv);
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const int ih = oh + kh - 1;
                if (ih < 0 || ih >= g.Hi) continue;
                const T* row = x + (((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi * g.ldx;
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const int iw = ow + kw - 1;
                    if (iw < 0 || iw >= g.Wi) continue;
                    const float xv = to_f32<T>(row[(int64_t)iw * g.ldx]);
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[kh * 3 + kw][j] = fmaf(xv, dv[j], acc[kh * 3 + kw][j]);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 9; a++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float s = wave_sum(acc[a][j]);
                if (lane == 0) part[((int64_t)blockIdx.x * 27 + kd * 9 + a) * g.Cout + co0 + j] = s;
            }
    }
}

// The same gradient on MFMA for bf16 with H % 8 == 0, W % 32 == 0: dW[tap][co] is a GEMM with M = 27 taps (padded to
// 32), N = 32 couts, K = positions.  A persistent workgroup walks (1 x 8 x 32)-position tiles; per tile it stages the
// dy rows (row-per-position, read back with the hardware 4x16 transpose) and three copies of the 3 x 10 x 32 input
// patch, pre-shifted by kw - 1 voxels so that every A fragment (8 consecutive positions of one tap) is one aligned
// ds_read_b128.  The 4 waves split the 16 k-steps of a tile.  dy is read exactly once; the MFMA time is ~1 % of it.
#define STEM_MF_BLOCKS 1024
typedef __attribute__((address_space(3))) bf16x4 stem_lds_bf16x4;
__device__ __forceinline__ bf16x8 stem_tr_frag(const bf16* p) {
    const bf16x4 lo = RU3D_DS_READ_TR16(p);
    const bf16x4 hi = RU3D_DS_READ_TR16(p + 4 * 32);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                              float* __restrict__ part, WgradGeom g, int tiles_h,
                                                              int tiles_w, int ntiles, float* __restrict__ bias_part) {
    __shared__ __attribute__((aligned(16))) bf16 ds[256 * 32];            // [position][32 co]
    __shared__ __attribute__((aligned(16))) bf16 xs[(3 * 3 * 10 + 2) * 32];   // [kw][kd][row][32 cols] + a zero row + a row of ones
    __shared__ float red[4][16][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * 32;
    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;
    // A fragment base of this lane: tap = lane & 31 (rows 27..31 read the zero row)
    const int tap = lane & 31;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    // rows 28..31 of the A operand are zeros; row 27 is ONES: its output row is sum_pos dy[pos][co] - the conv's bias
    // gradient from the pass that reads dy anyway (was a separate reduction over dy: 62 + 8 us at 2 x 128^3)
    const int abase = tap < 27 ? (((kw * 3 + kd) * 10 + kh) * 32 + 8 * h) : ((tap == 27 ? 91 : 90) * 32);
    if (tid < 32) {
        xs[90 * 32 + tid] = (bf16)0.f;
        xs[91 * 32 + tid] = (bf16)1.f;
    }

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int w0 = (t % tiles_w) * 32;
        t /= tiles_w;
        const int h0 = (t % tiles_h) * 8;
        t /= tiles_h;
        const int d = t % g.Do;
        const int n = t / g.Do;
        __syncthreads();   // previous tile consumed
        // dy rows: 256 positions x 4 pieces of 16 B
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + i * 256;
            const int f = c >> 2, piece = c & 3;
            const int64_t pos = (((int64_t)n * g.Do + d) * g.Ho + h0 + (f >> 5)) * g.Wo + w0 + (f & 31);
            bf16x8 v = {};      // positions of a border tile outside the volume contribute nothing
            if (h0 + (f >> 5) < g.Ho && w0 + (f & 31) < g.Wo)
                v = *reinterpret_cast<const bf16x8*>(dy + pos * g.lddy + co0 + piece * 8);
            *reinterpret_cast<bf16x8*>(&ds[f * 32 + piece * 8]) = v;
        }
        // the three shifted copies of the input patch: xs[kw][kd][r][c] = x[d + kd - 1][h0 + r - 1][w0 + c + kw - 1]
        for (int e = tid; e < 3 * 3 * 10 * 32; e += 256) {
            const int c = e & 31, r = (e >> 5) % 10, kk = (e >> 5) / 10;
            const int kdd = kk % 3, kww = kk / 3;
            const int id = d + kdd - 1, ih = h0 + r - 1, iw = w0 + c + kww - 1;
            bf16 v = (bf16)0.f;
            if (id >= 0 && id < g.Di && ih >= 0 && ih < g.Hi && iw >= 0 && iw < g.Wi)
                v = x[((((int64_t)n * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.ldx];
            xs[e] = v;
        }
        __syncthreads();
        // k-steps 4*wave .. 4*wave + 3: positions f0 = 16 ks + 8 h, row ks >> 1, columns 16 (ks & 1) + 8 h ..
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ks = wave * 4 + j;
            const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&xs[abase + (tap < 27 ? ((ks >> 1) * 32 + (ks & 1) * 16) : 0)]);
            const bf16x8 bfrag = stem_tr_frag(ds + (ks * 16 + 8 * h) * 32 + lane_off);
            acc = RU3D_MFMA_32X32X16(afrag, bfrag, acc, 0, 0, 0);
        }
    }
    // sum the 4 waves' partial tiles in a fixed order; D row = tap = (i & 3) + 8 (i >> 2) + 4 (lane >> 5), col = co
#pragma unroll
    for (int i = 0; i < 16; i++) red[wave][i][lane] = acc[i];
    __syncthreads();
    for (int e = tid; e < 16 * 64; e += 256) {
        const int i = e >> 6, ln = e & 63;
        const float s = (red[0][i][ln] + red[1][i][ln]) + (red[2][i][ln] + red[3][i][ln]);
        const int tp = (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5), co = ln & 31;
        if (tp < 27) part[((int64_t)blockIdx.x * 27 + tp) * g.Cout + co0 + co] = s;
        else if (tp == 27 && bias_part) bias_part[(int64_t)blockIdx.x * g.Cout + co0 + co] = s;
    }
}

static bool stem_wgrad_mfma_ok(const WgradGeom& g, int dtype) {
    static const int mode = getenv("RU3D_STEM_MFMA") ? atoi(getenv("RU3D_STEM_MFMA")) : 1;
    return mode && dtype == RU3D_BF16 && g.Do == g.Di && g.Ho == g.Hi && g.Wo == g.Wi && g.pad == 1;
}

static int stem_mfma_blocks(const WgradGeom& g) {
    const int64_t ntiles = (int64_t)g.N * g.Do * ((g.Ho + 7) / 8) * ((g.Wo + 31) / 32);
    return (int)(ntiles < STEM_MF_BLOCKS ? ntiles : STEM_MF_BLOCKS);
}

bool stem_wgrad_eligible(const WgradGeom& g) {
    return g.Cin == 1 && g.k == 3 && g.stride == 1 && (g.Cout % 32) == 0 && (g.lddy % 8) == 0 &&
           (int64_t)g.N * g.Do * g.Ho * g.Wo < (1ll << 31);
}

static int stem_chunks(const WgradGeom& g) {
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    return (int)((P + STEM_CHUNK - 1) / STEM_CHUNK);
}

size_t stem_wgrad_ws_bytes(const WgradGeom& g) {
    // upper bound over both kernels (the MFMA form writes at most STEM_MF_BLOCKS slabs, 28 rows each with the bias row)
    const size_t chunks = (size_t)stem_chunks(g) > (size_t)STEM_MF_BLOCKS ? (size_t)stem_chunks(g) : (size_t)STEM_MF_BLOCKS;
    return chunks * 28 * g.Cout * sizeof(float);
}

// the MFMA form can deliver the bias gradient (sum of dy over all positions) from the same pass
bool stem_wgrad_gives_bias(const WgradGeom& g, int dtype) {
    const int64_t ntiles = (int64_t)g.N * g.Do * ((g.Ho + 7) / 8) * ((g.Wo + 31) / 32);
    return stem_wgrad_eligible(g) && stem_wgrad_mfma_ok(g, dtype) && ntiles <= 0x7fffffff;
}

int stem_wgrad_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, const WgradGeom& g,
                      int dtype, hipStream_t st, float* db) {
    if (!ws || ws_bytes < stem_wgrad_ws_bytes(g)) return ru3d_fail(-1, "stem_wgrad: workspace too small");
    if (db && !stem_wgrad_gives_bias(g, dtype)) return ru3d_fail(-1, "stem_wgrad: no bias gradient from this form");
    if (stem_wgrad_mfma_ok(g, dtype)) {
        const int blocks = stem_mfma_blocks(g);
        const int tiles_h = (g.Ho + 7) / 8, tiles_w = (g.Wo + 31) / 32;
        const int64_t ntiles = (int64_t)g.N * g.Do * tiles_h * tiles_w;
        if (ntiles <= 0x7fffffff) {
            float* bias_part = db ? (float*)ws + (size_t)blocks * 27 * g.Cout : nullptr;
            hipLaunchKernelGGL(stem_wgrad_mfma_kernel, dim3(blocks, g.Cout / 32), dim3(256), 0, st, (const bf16*)x,
                               (const bf16*)dy, (float*)ws, g, tiles_h, tiles_w, (int)ntiles, bias_part);
            int rc = ru3d_check_launch("stem_wgrad_mfma");
            if (rc) return rc;
            rc = wgrad_reduce_launch((const float*)ws, dw, blocks, 27, 1, g.Cout, g.s_o, g.s_i, st);
            if (rc || !db) return rc;
            return wgrad_reduce_launch(bias_part, db, blocks, 1, 1, g.Cout, 1, 1, st);
        }
    }
    const int chunks = stem_chunks(g);
    dim3 grid(chunks, g.Cout / 32);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL(stem_wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)dy,
                           (float*)ws, g);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (const bf16*)dy,
                           (float*)ws, g);
    int rc = ru3d_check_launch("stem_wgrad");
    if (rc) return rc;
    return wgrad_reduce_launch((const float*)ws, dw, chunks, 27, 1, g.Cout, g.s_o, g.s_i, st);
}

// --------------------------------------------------------------------------- head weight gradient (k1, Cout <= 4)
// dW[co][ci] = sum_pos x[pos][ci] * dy[pos][co]: thread = (position lane, group of VEC input channels), 16-byte
// x loads, the 2..4 dy values of a position are shared by the channel-group lanes.
#define HEAD_SPAN 4096
template <typename T, int VEC>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ part, WgradGeom g) {
    __shared__ float sh[256][VEC * 4 + 1];
    const int G = g.Cin / VEC;          // channel groups (<= 256)
    const int vpb = 256 / G;
    const int tid = threadIdx.x;
    const int cg = tid % G, vl = tid / G;
    const bool active = vl < vpb;
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int64_t p0 = (int64_t)blockIdx.x * HEAD_SPAN;
    int64_t p1 = p0 + HEAD_SPAN;
    if (p1 > P) p1 = P;
    float acc[VEC][4];
#pragma unroll
    for (int i = 0; i < VEC; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[i][c] = 0.f;
    if (active) {
        // four positions per round, all their loads issued before the first FMA (a streaming kernel: what it needs is
        // bytes in flight; positions past the span read position p0 and contribute zeros)
        constexpr int WU = 4;
        for (int64_t p = p0 + vl; p < p1; p += (int64_t)WU * vpb) {
            float xv[WU][VEC], dv[WU][4];
#pragma unroll
            for (int u = 0; u < WU; u++) {
                const int64_t pu = p + (int64_t)u * vpb;
                const bool ok = pu < p1;
                const int64_t pp = ok ? pu : p0;
                load_vec<T, VEC>(x + pp * g.ldx + cg * VEC, xv[u]);
#pragma unroll
                for (int c = 0; c < 4; c++) dv[u][c] = (ok && c < g.Cout) ? to_f32<T>(dy[pp * g.lddy + c]) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < WU; u++)
#pragma unroll
                for (int i = 0; i < VEC; i++)
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[i][c] = fmaf(xv[u][i], dv[u][c], acc[i][c]);
        }
    }
#pragma unroll
    for (int i = 0; i < VEC; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) sh[tid][i * 4 + c] = active ? acc[i][c] : 0.f;
    __syncthreads();
    if (vl == 0) {
#pragma unroll
        for (int i = 0; i < VEC; i++)
            for (int c = 0; c < g.Cout; c++) {
                float s = 0.f;
                for (int l = 0; l < vpb; l++) s += sh[l * G + cg][i * 4 + c];
                // slab layout [chunk][tap = 0][ci][co]
                part[((int64_t)blockIdx.x * g.Cin + cg * VEC + i) * g.Cout + c] = s;
            }
    }
}

// --------------------------------------------------------------------------- head backward, fused (k1, Cout <= 4)
// The gradient of the logits arrives as fp32 from the loss kernel; the unfused path cast it to the storage type (one pass),
// then read that copy three times: weight gradient, bias gradient (a channel sum) and input gradient - 253 us in six
// launches at 2 x 128^3.  Here one pass reads a = the head's input and dlogits once, rounds dlogits to the storage type in
// registers (the value every unfused consumer saw) and produces
//     dx[v][ci]  = sum_co d[v][co] * w[co][ci]        (stored, 16 bytes per lane)
//     dW[co][ci] = sum_v  a[v][ci] * d[v][co]          (per-thread fp32 sums -> per-block partial -> fixed-order finalize)
//     db[co]     = sum_v  d[v][co]
// Thread = (voxel lane, 8 input channels); the G = Cin / 8 lanes of a voxel share its <= 4 gradient values.
constexpr int HB_U = 4;
__global__ __launch_bounds__(256) void head_bwd_kernel(const bf16* __restrict__ a, int lda, const float* __restrict__ dlog,
                                                       int ldd, const float* __restrict__ w, int cin_real, int Cin, int Cout,
                                                       bf16* __restrict__ dx, int lddx, float* __restrict__ part, int64_t P) {
    __shared__ float sh[256][33];
    const int G = Cin / 8, vpb = 256 / G;
    const int tid = threadIdx.x, cg = tid % G, vl = tid / G;
    float wr[4][8];      // the packed (16-bit) weight the unfused input gradient read
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int ci = cg * 8 + j;
            wr[c][j] = (c < Cout && ci < cin_real) ? (float)(bf16)w[c * cin_real + ci] : 0.f;
        }
    float aw[4][8], ab[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        ab[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) aw[c][j] = 0.f;
    }
    const int64_t stride = (int64_t)gridDim.x * vpb;
    for (int64_t p0 = (int64_t)blockIdx.x * vpb + vl; p0 < P; p0 += HB_U * stride) {
        bf16x8 xv[HB_U];
        float dv[HB_U][4];
#pragma unroll
        for (int u = 0; u < HB_U; u++) {
            const int64_t p = p0 + u * stride;
            const bool ok = p < P;
            const int64_t pp = ok ? p : p0;
            xv[u] = *reinterpret_cast<const bf16x8*>(a + pp * lda + cg * 8);
#pragma unroll
            for (int c = 0; c < 4; c++) dv[u][c] = (ok && c < Cout) ? (float)(bf16)dlog[pp * ldd + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < HB_U; u++) {
            const int64_t p = p0 + u * stride;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    o[j] = fmaf(dv[u][c], wr[c][j], o[j]);                  // head_dgrad_kernel's order
                    aw[c][j] = fmaf((float)xv[u][j], dv[u][c], aw[c][j]);
                }
                ab[c] += dv[u][c];
            }
            if (p < P) store_vec<bf16, 8>(dx + p * lddx + cg * 8, o);
        }
    }
    // per-block partial: the voxel lanes of a channel group in a fixed order
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) sh[tid][c * 8 + j] = aw[c][j];
    sh[tid][32] = 0.f;
    __syncthreads();
    float* pp = part + (int64_t)blockIdx.x * (4 * Cin + 4);
    for (int e = tid; e < 4 * Cin; e += 256) {
        const int c = e / Cin, ci = e % Cin, g8 = ci / 8, j = ci % 8;
        float s = 0.f;
        for (int l = 0; l < vpb; l++) s += sh[l * G + g8][c * 8 + j];
        pp[e] = s;
    }
    __syncthreads();
    if (cg == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) sh[vl][c] = ab[c];
    }
    __syncthreads();
    if (tid < 4) {
        float s = 0.f;
        for (int l = 0; l < vpb; l++) s += sh[l][tid];
        pp[4 * Cin + tid] = s;
    }
}

// dW[co][ci < cin_real], db[co] = sums of the per-block partials in block order (double)
__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ part, int blocks, int Cin,
                                                                int cin_real, int Cout, float* __restrict__ dw,
                                                                float* __restrict__ db) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per output value
    const int row = 4 * Cin + 4;
    if (e >= row) return;
    double acc = 0.0;
    for (int b = lane; b < blocks; b += 64) acc += (double)part[(int64_t)b * row + e];
    acc = wave_sum_d(acc);
    if (lane) return;
    if (e < 4 * Cin) {
        const int c = e / Cin, ci = e % Cin;
        if (c < Cout && ci < cin_real) dw[c * cin_real + ci] = (float)acc;
    } else if (e - 4 * Cin < Cout && db) {
        db[e - 4 * Cin] = (float)acc;
    }
}

static int head_bwd_blocks(int64_t P, int Cin) {
    const int vpb = 256 / (Cin / 8);
    int64_t b = (P + (int64_t)vpb * HB_U - 1) / ((int64_t)vpb * HB_U);
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

bool head_bwd_eligible(int Cin, int Cout, int dtype) {
    static const int mode = getenv("RU3D_HEAD_FUSED") ? atoi(getenv("RU3D_HEAD_FUSED")) : 1;
    const int G = Cin / 8;
    return mode && dtype == RU3D_BF16 && Cout >= 1 && Cout <= 4 && (Cin % 8) == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0;
}

size_t head_bwd_ws_bytes(int64_t P, int Cin) { return (size_t)head_bwd_blocks(P, Cin) * (4 * Cin + 4) * sizeof(float); }

int head_bwd_launch(const void* a, int lda, const float* dlog, int ldd, const float* w, int cin_real, int Cin, int Cout,
                    void* dx, int lddx, float* dw, float* db, void* ws, int64_t P, hipStream_t st) {
    const int blocks = head_bwd_blocks(P, Cin);
    hipLaunchKernelGGL(head_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16*)a, lda, dlog, ldd, w, cin_real, Cin, Cout,
                       (bf16*)dx, lddx, (float*)ws, P);
    int rc = ru3d_check_launch("head_bwd");
    if (rc) return rc;
    const int vals = 4 * Cin + 4;
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3((vals + 3) / 4), dim3(256), 0, st, (const float*)ws, blocks, Cin,
                       cin_real, Cout, dw, db);
    return ru3d_check_launch("head_bwd_finalize");
}

bool head_wgrad_eligible(const WgradGeom& g, int dtype) {
    const int vec = dtype == RU3D_BF16 ? 8 : 4;
    return g.k == 1 && g.stride == 1 && g.Cout <= 4 && (g.Cin % vec) == 0 && (g.Cin / vec) <= 256 &&
           (g.ldx % vec) == 0;
}

static int head_chunks(const WgradGeom& g) {
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    return (int)((P + HEAD_SPAN - 1) / HEAD_SPAN);
}

size_t head_wgrad_ws_bytes(const WgradGeom& g) { return (size_t)head_chunks(g) * g.Cin * g.Cout * sizeof(float); }

int head_wgrad_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, const WgradGeom& g,
                      int dtype, hipStream_t st) {
    if (!ws || ws_bytes < head_wgrad_ws_bytes(g)) return ru3d_fail(-1, "head_wgrad: workspace too small");
    if (((uintptr_t)x) % 16) return ru3d_fail(-1, "head_wgrad: x must be 16-byte aligned");
    const int chunks = head_chunks(g);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL((head_wgrad_kernel<float, 4>), dim3(chunks), dim3(256), 0, st, (const float*)x,
                           (const float*)dy, (float*)ws, g);
    else
        hipLaunchKernelGGL((head_wgrad_kernel<bf16, 8>), dim3(chunks), dim3(256), 0, st, (const bf16*)x,
                           (const bf16*)dy, (float*)ws, g);
    int rc = ru3d_check_launch("head_wgrad");
    if (rc) return rc;
    return wgrad_reduce_launch((const float*)ws, dw, chunks, 1, g.Cin, g.Cout, g.s_o, g.s_i, st);
}

}  // namespace RU3D_NS
