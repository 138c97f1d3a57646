// InstanceNorm3d + LeakyReLU on the SMALL levels (reference network.py:384-386, 411-416 at 16^3 / 8^3 voxels, hundreds of
// channels).
//
// norm.hip walks a tensor with thousands of blocks and needs three launches per normalisation (partial sums, fixed-order
// finalize, apply) - on a 1-4 MB tensor each of them is 4-6 us of launch ramp and drain around ~0.3 us of traffic, and a
// ResBlock of the two deepest levels spends as long in them (and in the split-K sum of its convs) as in the convolutions.
// Two forms replace them, by sample size V:
//
//  * V <= 1024 (8^3; 10x10x5): WHOLE-INSTANCE kernels.  One workgroup owns all voxels of 8 consecutive channels of one
//    sample (or of every sample, when a sum over the batch is wanted): a thread holds one or two 16-byte voxel pieces in
//    registers, the tensor is read ONCE, the two sums are combined inside the workgroup (DPP row sums in fp32, then the row
//    sums in double through LDS, fixed order: deterministic) and the apply runs from the registers.  One launch for
//    statistics + finalize + apply, forward and backward; the load phase can also sum the split-K slices of the deepest
//    convs (the fp32 slices conv3_s1_mfma_kernel / conv_ws leave in the workspace) with conv_ksplit_reduce_kernel's
//    arithmetic, so that launch disappears too.  (At 16^3 this form loses: 8 channels = 16 bytes of every 128-byte line,
//    eight times the tensor through the L1 of the 64 CUs that run it: 16-25 us, measured.)
//  * 1024 < V <= 8192 (16^3; 20x20x10): TWO coalesced kernels.  `sums`: a workgroup takes a slice of 256 / 512 voxels x
//    64 / 32 channels (whole 128-byte lines), leaves one row of double partials per slice; `apply`: the same decomposition,
//    every workgroup first adds the <= 32 partial rows of its channels in slice order (the finalize launch folded into the
//    apply: redundant per workgroup, a few KB), then applies.
//
// Same arithmetic per element as the norm.hip kernels (the apply expressions are copied), so a checkpointed block that
// recomputes its activation with ru3d_in_lrelu_fwd gets the same bits.
#include "common.h"
#include "conv.h"

namespace RU3D_NS {

namespace {

__device__ __forceinline__ float row_sum16(float v) {
    // inclusive scan by doubling over a DPP row of 16 lanes; zeros are shifted in (bound_ctrl): lane 15 holds the row's sum.
    // (A xor-butterfly of __shfl_xor compiles to ds_bpermute: 192 trips through the LDS crossbar for 16 doubles, 6 us.)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

// Sums of 16 per-thread values over each of the NS thread groups (TS threads each, TS % 16 == 0) of the workgroup, in a
// fixed order: tot[g * 16 + k].
__device__ __forceinline__ void group_sums16(const float (&v)[16], double* shw, double* tot, int TS, int NS) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float r = row_sum16(v[k]);
        if ((tid & 15) == 15) shw[(tid >> 4) * 16 + k] = (double)r;
    }
    __syncthreads();
    if (tid < 16 * NS) {
        const int g = tid >> 4, k = tid & 15, rows = TS >> 4;
        double s = 0.0;
        for (int w = 0; w < rows; w++) s += shw[(g * rows + w) * 16 + k];
        tot[tid] = s;
    }
    __syncthreads();
}

struct SmallSrc {
    const bf16* y;        // the tensor (pitch ldy) - or, KSPLIT, where the summed tensor is stored
    int ldy;
    const float* part;    // KSPLIT: split-K slices [z][N * V][C] fp32
    int ksplit;
    const float* bias;    // KSPLIT: added to the slice sum (may be NULL)
    bf16* ystore;         // KSPLIT: the rounded sum is stored here (NULL: nobody else needs it)
};

// 8 channels of voxel `row` as the STORED 16-bit values (what every later reader sees).  No branches: the caller clamps
// `row` into the sample (a thread beyond the last voxel re-reads the last one and discards it).
template <bool KSPLIT>
__device__ __forceinline__ bf16x8 small_load(const SmallSrc& s, int64_t row, int c0, int64_t NV, int C, bool store) {
    if (!KSPLIT) return *reinterpret_cast<const bf16x8*>(s.y + row * s.ldy + c0);
    // conv_ksplit_reduce_kernel's arithmetic: bias first, then the slices in order
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = s.bias ? s.bias[c0 + i] : 0.f;
    for (int z = 0; z < s.ksplit; z++) {
        const float* p = s.part + ((int64_t)z * NV + row) * C + c0;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] += lo[i];
            v[4 + i] += hi[i];
        }
    }
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (bf16)v[i];
    if (s.ystore && store) *reinterpret_cast<bf16x8*>(s.ystore + row * s.ldy + c0) = r;
    return r;
}

// ------------------------------------------------------------------------------------------- whole-instance kernels
// blockDim = TS * NS <= 1024: NS samples side by side (NS == 1: the sample is blockIdx.y), TS threads x P pieces >= V each.
template <int P, bool HAS_RES, bool KSPLIT>
__global__ __launch_bounds__(1024) void in_small_fwd_kernel(SmallSrc src, const float* __restrict__ drop,
                                                            const bf16* __restrict__ res, int ldr, bf16* __restrict__ out,
                                                            int ldo, float* __restrict__ mean, float* __restrict__ scale,
                                                            int V, int C, int N, double invV, float eps, float slope,
                                                            int TS, int NS) {
    __shared__ double shw[1024];
    __shared__ double tot[64];
    __shared__ float ms[64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 8;
    const int g = tid / TS, tl = tid - g * TS;
    const int n = NS > 1 ? g : (int)blockIdx.y;
    const int64_t base = (int64_t)n * V, NV = (int64_t)N * V;
    bf16x8 raw[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int v = tl + TS * p;
        raw[p] = small_load<KSPLIT>(src, base + (v < V ? v : V - 1), c0, NV, C, v < V);
    }
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const bool ok = tl + TS * p < V;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float f = ok ? (float)raw[p][i] : 0.f;
            s[i] += f;
            s[8 + i] = fmaf(f, f, s[8 + i]);
        }
    }
    group_sums16(s, shw, tot, TS, NS);
    if (tid < 16 * NS && (tid & 15) < 8) {
        const int gg = tid >> 4, c = tid & 15;
        const int nn = NS > 1 ? gg : n;
        const int i = nn * C + c0 + c;
        const double m = tot[gg * 16 + c] * invV;
        double var = tot[gg * 16 + 8 + c] * invV - m * m;
        if (var < 0.0) var = 0.0;
        const double sd = drop ? (double)drop[i] : 1.0;
        const float mf = (float)m, sf = (float)(sd / sqrt(sd * sd * var + (double)eps));
        mean[i] = mf;
        scale[i] = sf;
        ms[gg * 16 + c] = mf;
        ms[gg * 16 + 8 + c] = sf;
    }
    __syncthreads();
    float mu[8], sc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mu[i] = ms[g * 16 + i];
        sc[i] = ms[g * 16 + 8 + i];
    }
    bf16x8 rq[P];
    if (HAS_RES) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int v = tl + TS * p;
            rq[p] = *reinterpret_cast<const bf16x8*>(res + (base + (v < V ? v : V - 1)) * ldr + c0);
        }
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int v = tl + TS * p;
        float ov[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float t = ((float)raw[p][i] - mu[i]) * sc[i];      // in_lrelu_fwd_kernel's expression
            if (HAS_RES) t += (float)rq[p][i];
            ov[i] = lrelu_f(t, slope);
        }
        if (v < V) store_vec<bf16, 8>(out + (base + v) * ldo + c0, ov);
    }
}

// RESID: out = lrelu(IN(y) + res):  g' = gout * lrelu'(out) is rounded and stored (it is dL/dres), xhat from y
// else : out = lrelu(IN(y)):        xhat recovered from out itself, nothing else stored
template <int P, bool RESID, bool KSPLIT>
__global__ __launch_bounds__(1024) void in_small_bwd_kernel(SmallSrc gsrc, const bf16* __restrict__ outp, int ldo,
                                                            const bf16* __restrict__ y, int ldy,
                                                            const float* __restrict__ mean, const float* __restrict__ scale,
                                                            bf16* __restrict__ dy, int lddy, bf16* __restrict__ gpre,
                                                            int ldgp, float* __restrict__ gpre_sum, int V, int C, int N,
                                                            double invV, float slope, int zero_far, int D, int H, int W,
                                                            int TS, int NS) {
    __shared__ double shw[1024];
    __shared__ double tot[64];
    __shared__ float ms[64];
    const int tid = threadIdx.x, c0 = blockIdx.x * 8;
    const int g = tid / TS, tl = tid - g * TS;
    const int n = NS > 1 ? g : (int)blockIdx.y;
    const int64_t base = (int64_t)n * V, NV = (int64_t)N * V;
    const float inv_slope = 1.f / slope;
    float mu[8], sc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mu[i] = mean[n * C + c0 + i];
        sc[i] = scale[n * C + c0 + i];
    }
    bf16x8 qa[P], qb[P], qc[P];      // RESID: (gout -> g' rounded, out, y)   else: (gout, out, -)
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int v = tl + TS * p;
        const int64_t row = base + (v < V ? v : V - 1);
        qa[p] = small_load<KSPLIT>(gsrc, row, c0, NV, C, false);
        qb[p] = *reinterpret_cast<const bf16x8*>(outp + row * ldo + c0);
        if (RESID) qc[p] = *reinterpret_cast<const bf16x8*>(y + row * ldy + c0);
    }
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int v = tl + TS * p;
        const bool ok = v < V;
        if (RESID) {
            bf16x8 gp;
#pragma unroll
            for (int i = 0; i < 8; i++) gp[i] = (bf16)((float)qb[p][i] > 0.f ? (float)qa[p][i] : (float)qa[p][i] * slope);
            qa[p] = gp;
            if (ok) *reinterpret_cast<bf16x8*>(gpre + (base + v) * ldgp + c0) = gp;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float gp, xh;
            if (RESID) {
                gp = (float)qa[p][i];
                xh = ((float)qc[p][i] - mu[i]) * sc[i];
            } else {
                const float gv = (float)qa[p][i], ov = (float)qb[p][i];
                gp = ov > 0.f ? gv : gv * slope;
                xh = ov > 0.f ? ov : ov * inv_slope;
            }
            gp = ok ? gp : 0.f;
            s[i] += gp;
            s[8 + i] = fmaf(gp, xh, s[8 + i]);
        }
    }
    group_sums16(s, shw, tot, TS, NS);
    if (tid < 16 * NS) ms[tid] = (float)(tot[tid] * invV);      // m1 (k < 8), m2 (k >= 8) of group tid >> 4
    __syncthreads();
    float m1[8], m2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m1[i] = ms[g * 16 + i];
        m2[i] = ms[g * 16 + 8 + i];
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int v = tl + TS * p;
        bool far = false;
        if (zero_far) {
            const int w = v % W, h = (v / W) % H, d = v / (W * H);
            far = (w == W - 1) || (h == H - 1) || (d == D - 1);
        }
        float dv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float gp, xh;
            if (RESID) {
                gp = (float)qa[p][i];
                xh = ((float)qc[p][i] - mu[i]) * sc[i];
            } else {
                const float gv = (float)qa[p][i], ov = (float)qb[p][i];
                gp = ov > 0.f ? gv : gv * slope;
                xh = ov > 0.f ? ov : ov * inv_slope;
            }
            dv[i] = far ? 0.f : sc[i] * (gp - m1[i] - xh * m2[i]);      // in_lrelu_bwd_kernel's expression
        }
        if (v < V) store_vec<bf16, 8>(dy + (base + v) * lddy + c0, dv);
    }
    // the skip conv's bias gradient: V * sum over the samples (all of them sit in this workgroup) of the rounded m1, as
    // bwd_finalize_kernel forms it
    if (gpre_sum && tid < 8) {
        double b = 0.0;
        for (int gg = 0; gg < NS; gg++) b += (double)ms[gg * 16 + tid];
        gpre_sum[c0 + tid] = (float)(b * (double)V);
    }
}

// ------------------------------------------------------------------------------------------- two-kernel coalesced form
// Workgroup (256 threads) = PL piece lanes (CW = 8 * PL channels: whole 128- or 64-byte voxel rows) x VL = 256 / PL voxel
// lanes x IT voxels each; grid (slices, N, C / CW).  Partials: part[((n * S + slice) * C + c) * 2 + {0, 1}] as double.
constexpr int IT = 8;

struct TwoArgs {
    const bf16* a; int lda;     // fwd: y            bwd: gout (RESID apply: g')
    const bf16* b; int ldb;     // fwd apply: res    bwd: out
    const bf16* c; int ldc;     // bwd RESID: y
    bf16* o1; int ld1;          // fwd apply: out    bwd apply: dy      bwd sums RESID: g'
    double* part;
    const float* mean; const float* scale; const float* drop;
    float* mean_out; float* scale_out; float* gpre_sum;
    int V, C, N, S;
    double invV;
    float eps, slope;
    int zero_far, D, H, W;
};

// MODE 0: (sum y, sum y^2)   MODE 1: backward sums, out = lrelu(IN(y))   MODE 2: backward sums, residual form (stores g')
template <int PL, int MODE>
__global__ __launch_bounds__(256) void in_two_sums_kernel(TwoArgs t) {
    constexpr int VL = 256 / PL, CW = 8 * PL;
    __shared__ float sh[VL][PL * 16 + 1];
    const int tid = threadIdx.x, pl = tid % PL, vl = tid / PL;
    const int n = blockIdx.y, c0 = blockIdx.z * CW + pl * 8;
    const int v0 = blockIdx.x * (VL * IT);
    const int64_t base = (int64_t)n * t.V;
    const float inv_slope = 1.f / t.slope;
    float mu[8], sc[8];
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            mu[i] = t.mean[n * t.C + c0 + i];
            sc[i] = t.scale[n * t.C + c0 + i];
        }
    }
    bf16x8 qa[IT], qb[IT], qc[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int v = v0 + vl + VL * it;
        const int64_t row = base + (v < t.V ? v : t.V - 1);
        qa[it] = *reinterpret_cast<const bf16x8*>(t.a + row * t.lda + c0);
        if (MODE >= 1) qb[it] = *reinterpret_cast<const bf16x8*>(t.b + row * t.ldb + c0);
        if (MODE == 2) qc[it] = *reinterpret_cast<const bf16x8*>(t.c + row * t.ldc + c0);
    }
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0.f;
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int v = v0 + vl + VL * it;
        const bool ok = v < t.V;
        if (MODE == 2) {
            bf16x8 gp;
#pragma unroll
            for (int i = 0; i < 8; i++) gp[i] = (bf16)((float)qb[it][i] > 0.f ? (float)qa[it][i] : (float)qa[it][i] * t.slope);
            qa[it] = gp;
            if (ok) *reinterpret_cast<bf16x8*>(t.o1 + (base + v) * t.ld1 + c0) = gp;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float p, q;
            if (MODE == 0) {
                p = (float)qa[it][i];
                q = p;
            } else if (MODE == 2) {
                p = (float)qa[it][i];
                q = ((float)qc[it][i] - mu[i]) * sc[i];
            } else {
                const float gv = (float)qa[it][i], ov = (float)qb[it][i];
                p = ov > 0.f ? gv : gv * t.slope;
                q = ov > 0.f ? ov : ov * inv_slope;
            }
            p = ok ? p : 0.f;
            s[i] += p;
            s[8 + i] = fmaf(p, q, s[8 + i]);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) sh[vl][pl * 16 + k] = s[k];
    __syncthreads();
    if (tid < PL * 16) {
        double acc = 0.0;
#pragma unroll 8
        for (int l = 0; l < VL; l++) acc += (double)sh[l][tid];      // voxel-lane order: fixed
        const int c = blockIdx.z * CW + (tid >> 4) * 8 + (tid & 7), which = (tid >> 3) & 1;
        t.part[(((int64_t)n * t.S + blockIdx.x) * t.C + c) * 2 + which] = acc;
    }
}

// sums of the S partial rows of this workgroup's channels (sample n) -> fin[which][channel], slice order
template <int CW>
__device__ __forceinline__ void two_finalize(const double* __restrict__ part, int n, int S, int C, int cbase,
                                             double (*fin)[CW]) {
    const int tid = threadIdx.x;
    if (tid < 2 * CW) {
        const int cl = tid >> 1, which = tid & 1;
        const double* p = part + ((int64_t)n * S * C + cbase + cl) * 2 + which;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int s = 0;
        for (; s + 3 < S; s += 4) {      // four rows in flight; the additions keep slice order within each lane
            a0 += p[(int64_t)s * C * 2];
            a1 += p[(int64_t)(s + 1) * C * 2];
            a2 += p[(int64_t)(s + 2) * C * 2];
            a3 += p[(int64_t)(s + 3) * C * 2];
        }
        for (; s < S; s++) a0 += p[(int64_t)s * C * 2];
        fin[which][cl] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
}

// MODE 0: out = lrelu((y - mean) * scale (+ res))     MODE 1 / 2: dy = scale * (g' - m1 - xhat * m2)
template <int PL, int MODE, bool HAS_RES>
__global__ __launch_bounds__(256) void in_two_apply_kernel(TwoArgs t) {
    constexpr int VL = 256 / PL, CW = 8 * PL;
    __shared__ double fin[2][CW];
    __shared__ float ab[2][CW];
    const int tid = threadIdx.x, pl = tid % PL, vl = tid / PL;
    const int n = blockIdx.y, cbase = blockIdx.z * CW, c0 = cbase + pl * 8;
    const int v0 = blockIdx.x * (VL * IT);
    const int64_t base = (int64_t)n * t.V;
    const float inv_slope = 1.f / t.slope;
    // issue the tile's loads first: the partial rows' latency hides behind them
    bf16x8 qa[IT], qb[IT];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int v = v0 + vl + VL * it;
        const int64_t row = base + (v < t.V ? v : t.V - 1);
        qa[it] = *reinterpret_cast<const bf16x8*>(t.a + row * t.lda + c0);
        if (MODE == 1) qb[it] = *reinterpret_cast<const bf16x8*>(t.b + row * t.ldb + c0);
        if (MODE == 2) qb[it] = *reinterpret_cast<const bf16x8*>(t.c + row * t.ldc + c0);
        if (MODE == 0 && HAS_RES) qb[it] = *reinterpret_cast<const bf16x8*>(t.b + row * t.ldb + c0);
    }
    two_finalize<CW>(t.part, n, t.S, t.C, cbase, fin);
    if (tid < CW) {
        const int i = n * t.C + cbase + tid;
        if (MODE == 0) {
            const double m = fin[0][tid] * t.invV;
            double var = fin[1][tid] * t.invV - m * m;
            if (var < 0.0) var = 0.0;
            const double sd = t.drop ? (double)t.drop[i] : 1.0;
            const float mf = (float)m, sf = (float)(sd / sqrt(sd * sd * var + (double)t.eps));
            ab[0][tid] = mf;
            ab[1][tid] = sf;
            if (blockIdx.x == 0) {
                t.mean_out[i] = mf;
                t.scale_out[i] = sf;
            }
        } else {
            ab[0][tid] = (float)(fin[0][tid] * t.invV);
            ab[1][tid] = (float)(fin[1][tid] * t.invV);
        }
    }
    if (MODE == 2 && t.gpre_sum && blockIdx.x == 0 && n == 0 && tid >= 64 && tid < 64 + CW) {
        // the skip conv's bias gradient: V * sum_n of the rounded mean of g' (bwd_finalize_kernel's arithmetic)
        const int cl = tid - 64;
        double b = 0.0;
        for (int nn = 0; nn < t.N; nn++) {
            double a = 0.0;
            const double* p = t.part + ((int64_t)nn * t.S * t.C + cbase + cl) * 2;
            for (int s = 0; s < t.S; s++) a += p[(int64_t)s * t.C * 2];
            b += (double)(float)(a * t.invV);
        }
        t.gpre_sum[cbase + cl] = (float)(b * (double)t.V);
    }
    __syncthreads();
    float A[8], B[8], mu[8], sc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        A[i] = ab[0][pl * 8 + i];
        B[i] = ab[1][pl * 8 + i];
        if (MODE >= 1) {
            mu[i] = t.mean[n * t.C + c0 + i];
            sc[i] = t.scale[n * t.C + c0 + i];
        }
    }
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const int v = v0 + vl + VL * it;
        float ov[8];
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float x = ((float)qa[it][i] - A[i]) * B[i];      // in_lrelu_fwd_kernel's expression
                if (HAS_RES) x += (float)qb[it][i];
                ov[i] = lrelu_f(x, t.slope);
            }
        } else {
            bool far = false;
            if (t.zero_far) {
                const int w = v % t.W, h = (v / t.W) % t.H, d = v / (t.W * t.H);
                far = (w == t.W - 1) || (h == t.H - 1) || (d == t.D - 1);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float gp, xh;
                if (MODE == 2) {
                    gp = (float)qa[it][i];
                    xh = ((float)qb[it][i] - mu[i]) * sc[i];
                } else {
                    const float gv = (float)qa[it][i], o = (float)qb[it][i];
                    gp = o > 0.f ? gv : gv * t.slope;
                    xh = o > 0.f ? o : o * inv_slope;
                }
                ov[i] = far ? 0.f : sc[i] * (gp - A[i] - xh * B[i]);      // in_lrelu_bwd_kernel's expression
            }
        }
        if (v < t.V) store_vec<bf16, 8>(t.o1 + (base + v) * t.ld1 + c0, ov);
    }
}

inline bool piece_ok(const ru3d_tensor* t) { return !t || ((t->ld % 8) == 0 && (((uintptr_t)t->ptr) % 16) == 0); }

// whole-instance decomposition: threads per sample, pieces per thread, samples per workgroup; false: does not fit
inline bool whole_plan(int V, int N, bool all_samples, int* TS, int* P, int* NS) {
    const int ns = all_samples ? N : 1;
    for (int p = 1; p <= 2; p++) {
        int ts = 64;
        while (ts * p < V) ts *= 2;
        if (ts * ns <= 1024 && ns <= 4) {
            *TS = ts; *P = p; *NS = ns;
            return true;
        }
    }
    return false;
}

void two_geom(const ru3d_tensor* y, int* PL, int* S, dim3* grid) {
    const int V = y->d * y->h * y->w;
    *PL = (y->c % 64) == 0 ? 8 : 4;
    const int sv = (256 / *PL) * IT;
    *S = (V + sv - 1) / sv;
    *grid = dim3(*S, y->n, y->c / (8 * *PL));
}

}  // namespace

// 0: norm.hip's three launches; 1: whole-instance kernel; 2: two coalesced kernels.  all_samples: a sum over the batch
// (the skip conv's bias gradient) is wanted from the same call.
int in_small_mode(const ru3d_tensor* y, bool all_samples, const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* c,
                  const ru3d_tensor* d) {
    static const int mode = getenv("RU3D_IN_SMALL") ? atoi(getenv("RU3D_IN_SMALL")) : 1;
    if (!mode || !y) return 0;
    const int64_t V = (int64_t)y->d * y->h * y->w;
    if (V > 8192 || (y->c % 8) != 0) return 0;
    if (!(piece_ok(y) && piece_ok(a) && piece_ok(b) && piece_ok(c) && piece_ok(d))) return 0;
    int ts, p, ns;
    if (V <= 1024 && whole_plan((int)V, y->n, all_samples, &ts, &p, &ns)) return 1;
    if (V > 512 && (y->c % 32) == 0 && y->n <= 65535) return 2;
    return 0;
}

// workspace of the two-kernel form (partial rows); the whole-instance form needs none
size_t in_small_ws_bytes(const ru3d_tensor* y) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int64_t S = (V + 255) / 256;
    return (size_t)y->n * S * y->c * 2 * sizeof(double);
}

// part != NULL (whole-instance form only): y is produced here from `ksplit` fp32 slices (+ bias) and stored
int in_small_fwd_launch(const ru3d_tensor* y, const float* part, int ksplit, const float* bias, const float* drop,
                        float* mean, float* scale, const ru3d_tensor* res, const ru3d_tensor* out, void* ws, float eps,
                        float slope, hipStream_t st) {
    const int V = y->d * y->h * y->w;
    const double invV = 1.0 / (double)V;
    const int mode = in_small_mode(y, false, res, out);
    if (mode == 1) {
        int TS, P, NS;
        whole_plan(V, y->n, false, &TS, &P, &NS);
        SmallSrc src;
        src.y = (const bf16*)y->ptr; src.ldy = y->ld; src.part = part; src.ksplit = ksplit; src.bias = bias;
        src.ystore = part ? (bf16*)y->ptr : nullptr;
        dim3 grid(y->c / 8, y->n);
#define LAUNCH_F(PP, RR, KK)                                                                                              \
    hipLaunchKernelGGL((in_small_fwd_kernel<PP, RR, KK>), grid, dim3(TS * NS), 0, st, src, drop,                          \
                       res ? (const bf16*)res->ptr : (const bf16*)nullptr, res ? res->ld : 0, (bf16*)out->ptr, out->ld,   \
                       mean, scale, V, y->c, y->n, invV, eps, slope, TS, NS)
#define CALL(PP)                                     \
    if (res && part) LAUNCH_F(PP, true, true);       \
    else if (res) LAUNCH_F(PP, true, false);         \
    else if (part) LAUNCH_F(PP, false, true);        \
    else LAUNCH_F(PP, false, false)
        if (P == 1) { CALL(1); } else { CALL(2); }
#undef CALL
#undef LAUNCH_F
        return ru3d_check_launch("in_small_fwd");
    }
    if (mode != 2 || part) return ru3d_fail(-1, "in_small_fwd: shape not supported");
    TwoArgs t = {};
    int PL;
    dim3 grid;
    two_geom(y, &PL, &t.S, &grid);
    t.a = (const bf16*)y->ptr; t.lda = y->ld;
    t.b = res ? (const bf16*)res->ptr : nullptr; t.ldb = res ? res->ld : 0;
    t.o1 = (bf16*)out->ptr; t.ld1 = out->ld;
    t.part = (double*)ws;
    t.drop = drop; t.mean_out = mean; t.scale_out = scale;
    t.V = V; t.C = y->c; t.N = y->n; t.invV = invV; t.eps = eps; t.slope = slope;
    if (PL == 8) hipLaunchKernelGGL((in_two_sums_kernel<8, 0>), grid, dim3(256), 0, st, t);
    else hipLaunchKernelGGL((in_two_sums_kernel<4, 0>), grid, dim3(256), 0, st, t);
    int rc = ru3d_check_launch("in_two_sums");
    if (rc) return rc;
    if (PL == 8) {
        if (res) hipLaunchKernelGGL((in_two_apply_kernel<8, 0, true>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_apply_kernel<8, 0, false>), grid, dim3(256), 0, st, t);
    } else {
        if (res) hipLaunchKernelGGL((in_two_apply_kernel<4, 0, true>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_apply_kernel<4, 0, false>), grid, dim3(256), 0, st, t);
    }
    return ru3d_check_launch("in_two_apply");
}

// gpre != NULL: the residual form (y required); part != NULL (whole-instance form only): gout is the sum of `ksplit` fp32
// slices and is never stored
int in_small_bwd_launch(const ru3d_tensor* gout, const float* part, int ksplit, const ru3d_tensor* outp,
                        const ru3d_tensor* y, const float* mean, const float* scale, const ru3d_tensor* dy,
                        const ru3d_tensor* gpre, void* ws, float slope, int zero_far, float* gpre_sum, hipStream_t st) {
    const int V = outp->d * outp->h * outp->w;
    const double invV = 1.0 / (double)V;
    const int mode = in_small_mode(outp, gpre_sum != nullptr, gout, y, dy, gpre);
    if (mode == 1) {
        int TS, P, NS;
        whole_plan(V, outp->n, gpre_sum != nullptr, &TS, &P, &NS);
        SmallSrc src;
        src.y = gout ? (const bf16*)gout->ptr : nullptr; src.ldy = gout ? gout->ld : 0; src.part = part; src.ksplit = ksplit;
        src.bias = nullptr; src.ystore = nullptr;
        dim3 grid(outp->c / 8, NS > 1 ? 1 : outp->n);
#define LAUNCH_B(PP, RR, KK)                                                                                              \
    hipLaunchKernelGGL((in_small_bwd_kernel<PP, RR, KK>), grid, dim3(TS * NS), 0, st, src, (const bf16*)outp->ptr,        \
                       outp->ld, y ? (const bf16*)y->ptr : (const bf16*)nullptr, y ? y->ld : 0, mean, scale,              \
                       (bf16*)dy->ptr, dy->ld, gpre ? (bf16*)gpre->ptr : (bf16*)nullptr, gpre ? gpre->ld : 0, gpre_sum,   \
                       V, outp->c, outp->n, invV, slope, zero_far, outp->d, outp->h, outp->w, TS, NS)
#define CALL(PP)                                      \
    if (gpre && part) LAUNCH_B(PP, true, true);       \
    else if (gpre) LAUNCH_B(PP, true, false);         \
    else if (part) LAUNCH_B(PP, false, true);         \
    else LAUNCH_B(PP, false, false)
        if (P == 1) { CALL(1); } else { CALL(2); }
#undef CALL
#undef LAUNCH_B
        return ru3d_check_launch("in_small_bwd");
    }
    if (mode != 2 || part || !gout) return ru3d_fail(-1, "in_small_bwd: shape not supported");
    TwoArgs t = {};
    int PL;
    dim3 grid;
    two_geom(outp, &PL, &t.S, &grid);
    t.a = (const bf16*)gout->ptr; t.lda = gout->ld;
    t.b = (const bf16*)outp->ptr; t.ldb = outp->ld;
    t.c = gpre ? (const bf16*)y->ptr : nullptr; t.ldc = gpre ? y->ld : 0;
    t.part = (double*)ws;
    t.mean = mean; t.scale = scale; t.gpre_sum = gpre_sum;
    t.V = V; t.C = outp->c; t.N = outp->n; t.invV = invV; t.slope = slope;
    t.zero_far = zero_far; t.D = outp->d; t.H = outp->h; t.W = outp->w;
    if (gpre) {
        t.o1 = (bf16*)gpre->ptr; t.ld1 = gpre->ld;
        if (PL == 8) hipLaunchKernelGGL((in_two_sums_kernel<8, 2>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_sums_kernel<4, 2>), grid, dim3(256), 0, st, t);
    } else {
        if (PL == 8) hipLaunchKernelGGL((in_two_sums_kernel<8, 1>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_sums_kernel<4, 1>), grid, dim3(256), 0, st, t);
    }
    int rc = ru3d_check_launch("in_two_sums");
    if (rc) return rc;
    t.o1 = (bf16*)dy->ptr; t.ld1 = dy->ld;
    if (gpre) {
        t.a = (const bf16*)gpre->ptr; t.lda = gpre->ld;      // the apply reads the stored g'
        if (PL == 8) hipLaunchKernelGGL((in_two_apply_kernel<8, 2, false>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_apply_kernel<4, 2, false>), grid, dim3(256), 0, st, t);
    } else {
        if (PL == 8) hipLaunchKernelGGL((in_two_apply_kernel<8, 1, false>), grid, dim3(256), 0, st, t);
        else hipLaunchKernelGGL((in_two_apply_kernel<4, 1, false>), grid, dim3(256), 0, st, t);
    }
    return ru3d_check_launch("in_two_apply");
}

}  // namespace RU3D_NS
