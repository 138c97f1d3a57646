// InstanceNorm3d + LeakyReLU (+ residual, + folded Dropout3d) forward/backward, per-channel sums,
// channel-slice copy, add, layout repack.  All HBM-bound: 16-byte accesses, channels on the lanes,
// per-(n,c) parameters held in registers across a voxel loop, no integer division per element.
//
// Work decomposition shared by every kernel here ("channel loop"): a tensor is [N][V][C] with pitch
// ld.  Channels are cut into groups of VEC elements (one 16-byte access for bf16x8 / f32x4); a block
// of 256 threads owns Gb <= 256 channel groups x vpb = 256/Gb voxel lanes and walks a span of voxels.
// grid = (chunks, N, group blocks).
#include "common.h"
#include "conv.h"
#include <initializer_list>

namespace RU3D_NS {

struct ChanLoop {
    int V;      // voxels per sample
    int G;      // channel groups (C / VEC)
    int Gb;     // groups per block
    int vpb;    // voxel lanes per block
    int span;   // voxels per block
    int chunks; // blocks along the voxel axis
};

static ChanLoop make_chanloop(int64_t V, int C, int vec, int max_iters, int N = 1) {
    ChanLoop cl;
    cl.V = (int)V;
    cl.G = C / vec;
    cl.Gb = cl.G < 256 ? cl.G : 256;
    cl.vpb = 256 / cl.Gb;
    // small tensors (the deep levels: 8^3 .. 16^3 voxels, hundreds of channels): shorten the per-thread loop until
    // the launch has ~2048 blocks, otherwise a handful of blocks walk the tensor serially at load latency
    const int64_t gz = (cl.G + cl.Gb - 1) / cl.Gb;
    int64_t iters = (V * N * gz + (int64_t)cl.vpb * 2048 - 1) / ((int64_t)cl.vpb * 2048);
    if (iters < 4) iters = 4;
    if (iters > max_iters) iters = max_iters;
    int64_t span = (int64_t)cl.vpb * iters;
    int64_t chunks = (V + span - 1) / span;
    if (chunks > 4096) {
        chunks = 4096;
        span = (V + chunks - 1) / chunks;
        span = (span + cl.vpb - 1) / cl.vpb * cl.vpb;
        chunks = (V + span - 1) / span;
    }
    cl.span = (int)span;
    cl.chunks = (int)chunks;
    return cl;
}

template <typename T>
static int pick_vec(int c, std::initializer_list<const ru3d_tensor*> ts) {
    int vec = 16 / (int)sizeof(T);
    for (; vec > 1; vec >>= 1) {
        bool ok = (c % vec) == 0;
        for (const ru3d_tensor* t : ts) {
            if (!t) continue;
            ok = ok && (t->ld % vec == 0) && (((uintptr_t)t->ptr) % (vec * sizeof(T)) == 0);
        }
        if (ok) break;
    }
    return vec;
}

// --------------------------------------------------------------------------- two-sum reduction
// MODE 0: (sum y, sum y^2)                                 InstanceNorm statistics
// MODE 1: (sum gpre, sum gpre*xhat)                        InstanceNorm backward
// MODE 2: (sum t, 0)                                       bias gradient
// MODE 3: as MODE 1 with xhat recovered from the activation (no residual in the forward: out = lrelu(xhat))
// Partials: part[((n*chunks + chunk)*C + c)*2 + {0,1}] as double.
template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void reduce2_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b,
                                                      int ldb, const T* __restrict__ c3, int ldc,
                                                      const float* __restrict__ mean, const float* __restrict__ scale,
                                                      float slope, double* __restrict__ part, ChanLoop cl, int C,
                                                      T* __restrict__ gp_out = nullptr, int ldgp = 0,
                                                      const float* __restrict__ shift = nullptr) {
    // MODE 5 = MODE 4 for BatchNorm: xhat = y * scale + shift with per-(n, c) values (mean unused)
    // MODE 4 = MODE 1 that also stores g' = gout * lrelu'(out) (the pre-activation gradient): the apply pass then reads
    // (g', y) instead of (gout, out, y) - one tensor pass fewer per residual block backward
    __shared__ double sh[2][256][VEC > 4 ? 4 : VEC];  // reduced in two halves when VEC == 8
    const int tid = threadIdx.x;
    const int cgl = tid % cl.Gb, vl = tid / cl.Gb;
    const int cg = blockIdx.z * cl.Gb + cgl;
    const int n = blockIdx.y;
    const bool active = (vl < cl.vpb) && (cg < cl.G);
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) s1[i] = s2[i] = 0.f;
    if (active) {
        float mu[VEC], sc[VEC];
        if (MODE == 1 || MODE == 4 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < VEC; i++) {
                mu[i] = MODE == 5 ? shift[n * C + cg * VEC + i] : mean[n * C + cg * VEC + i];
                sc[i] = scale[n * C + cg * VEC + i];
            }
        }
        const int v0 = blockIdx.x * cl.span;
        int v1 = v0 + cl.span;
        if (v1 > cl.V) v1 = cl.V;
        for (int v = v0 + vl; v < v1; v += cl.vpb) {
            const int64_t row = (int64_t)n * cl.V + v;
            float av[VEC];
            load_vec<T, VEC>(a + row * lda + cg * VEC, av);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    s1[i] += av[i];
                    s2[i] = fmaf(av[i], av[i], s2[i]);
                }
            } else if (MODE == 1 || MODE == 4 || MODE == 5) {
                float ov[VEC], yv[VEC], pv[VEC];
                load_vec<T, VEC>(b + row * ldb + cg * VEC, ov);
                load_vec<T, VEC>(c3 + row * ldc + cg * VEC, yv);
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    float gp = ov[i] > 0.f ? av[i] : av[i] * slope;
                    if (MODE >= 4) gp = to_f32<T>(from_f32<T>(gp));      // the sums see the value the apply pass will read
                    pv[i] = gp;
                    const float xh = MODE == 5 ? fmaf(yv[i], sc[i], mu[i]) : (yv[i] - mu[i]) * sc[i];
                    s1[i] += gp;
                    s2[i] = fmaf(gp, xh, s2[i]);
                }
                if (MODE >= 4) store_vec<T, VEC>(gp_out + row * ldgp + cg * VEC, pv);
            } else if (MODE == 3) {
                float ov[VEC];
                load_vec<T, VEC>(b + row * ldb + cg * VEC, ov);
                const float inv_slope = 1.f / slope;
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    const float gp = ov[i] > 0.f ? av[i] : av[i] * slope;
                    const float xh = ov[i] > 0.f ? ov[i] : ov[i] * inv_slope;
                    s1[i] += gp;
                    s2[i] = fmaf(gp, xh, s2[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < VEC; i++) s1[i] += av[i];
            }
        }
    }
    // cross voxel-lane reduction in LDS (double), fixed order -> deterministic
    constexpr int HV = VEC > 4 ? 4 : VEC;
#pragma unroll
    for (int half = 0; half < VEC / HV; half++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HV; i++) {
            sh[0][tid][i] = active ? (double)s1[half * HV + i] : 0.0;
            sh[1][tid][i] = active ? (double)s2[half * HV + i] : 0.0;
        }
        __syncthreads();
        if (vl == 0 && cg < cl.G) {
#pragma unroll
            for (int i = 0; i < HV; i++) {
                double t1 = 0.0, t2 = 0.0;
                for (int l = 0; l < cl.vpb; l++) {
                    t1 += sh[0][l * cl.Gb + cgl][i];
                    t2 += sh[1][l * cl.Gb + cgl][i];
                }
                const int c = cg * VEC + half * HV + i;
                double* pp = part + (((int64_t)n * cl.chunks + blockIdx.x) * C + c) * 2;
                pp[0] = t1;
                pp[1] = t2;
            }
        }
    }
}

// Finalize kernels: 16 channel lanes x 16 chunk lanes per block; every lane sums a strided subset of the
// per-block partials in double, the 16 subsets are combined in a fixed order through LDS (deterministic).
#define FIN_CX 16
#define FIN_KY 16
__device__ __forceinline__ void finalize_sums(const double* __restrict__ part, int rows, int64_t row_stride, int c,
                                              bool ok, double& t1, double& t2) {
    // rows = number of partial rows for this (n) (or N*chunks for channel sums); row r at part + r*row_stride + c*2
    __shared__ double sh[2][FIN_KY][FIN_CX];
    const int cx = threadIdx.x % FIN_CX, ky = threadIdx.x / FIN_CX;
    double a1 = 0.0, a2 = 0.0;
    if (ok) {
        // eight independent partial rows in flight per lane (a serial chain of dependent loads cost ~10 us on the
        // 512-chunk tensors); the order of the additions is fixed, so the result is still deterministic
        const double* base = part + (int64_t)c * 2;
        int r = ky;
        for (; r + 7 * FIN_KY < rows; r += 8 * FIN_KY) {
            double u1[8], u2[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double* pp = base + (int64_t)(r + j * FIN_KY) * row_stride;
                u1[j] = pp[0];
                u2[j] = pp[1];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a1 += u1[j];
                a2 += u2[j];
            }
        }
        for (; r < rows; r += FIN_KY) {
            const double* pp = base + (int64_t)r * row_stride;
            a1 += pp[0];
            a2 += pp[1];
        }
    }
    sh[0][ky][cx] = a1;
    sh[1][ky][cx] = a2;
    __syncthreads();
    t1 = 0.0;
    t2 = 0.0;
    if (ky == 0) {
#pragma unroll
        for (int k = 0; k < FIN_KY; k++) {
            t1 += sh[0][k][cx];
            t2 += sh[1][k][cx];
        }
    }
}

// stats finalize: mean, scale = s / sqrt(s^2 var + eps)
__global__ __launch_bounds__(256) void stats_finalize_kernel(const double* __restrict__ part, int chunks, int C,
                                                             int NC, double invV, const float* __restrict__ drop,
                                                             float eps, float* __restrict__ mean,
                                                             float* __restrict__ scale) {
    const int i = blockIdx.x * FIN_CX + threadIdx.x % FIN_CX;
    const bool ok = i < NC;
    const int n = ok ? i / C : 0, c = ok ? i % C : 0;
    double t1, t2;
    finalize_sums(part + (int64_t)n * chunks * C * 2, chunks, (int64_t)C * 2, c, ok, t1, t2);
    if (!ok || threadIdx.x >= FIN_CX) return;
    const double m = t1 * invV;
    double var = t2 * invV - m * m;
    if (var < 0.0) var = 0.0;
    const double s = drop ? (double)drop[i] : 1.0;
    mean[i] = (float)m;
    scale[i] = (float)(s / sqrt(s * s * var + (double)eps));
}

// backward finalize: m1 = mean(gpre), m2 = mean(gpre * xhat), stored right after the partials.  A block takes FIN_CX
// channels and walks the samples, so the skip conv's bias gradient - sum over n and voxels of gpre = V * sum_n m1[n][c],
// formed from the rounded m1 like the separate pass it replaces - comes out of the same launch.
__global__ __launch_bounds__(256) void bwd_finalize_kernel(const double* __restrict__ part, int chunks, int C, int N,
                                                           double invV, float* __restrict__ m12, double V,
                                                           float* __restrict__ gpre_sum) {
    const int c = blockIdx.x * FIN_CX + threadIdx.x % FIN_CX;
    const bool ok = c < C;
    double s = 0.0;
    for (int n = 0; n < N; n++) {
        double t1, t2;
        finalize_sums(part + (int64_t)n * chunks * C * 2, chunks, (int64_t)C * 2, ok ? c : 0, ok, t1, t2);
        if (ok && threadIdx.x < FIN_CX) {
            const int i = n * C + c;
            const float m1 = (float)(t1 * invV);
            m12[2 * i] = m1;
            m12[2 * i + 1] = (float)(t2 * invV);
            s += (double)m1;
        }
        __syncthreads();   // finalize_sums' shared partials are reused by the next sample
    }
    if (gpre_sum && ok && threadIdx.x < FIN_CX) gpre_sum[c] = (float)(s * V);
}

__global__ __launch_bounds__(256) void chansum_finalize_kernel(const double* __restrict__ part, int chunks, int C,
                                                               int N, float* __restrict__ out) {
    const int c = blockIdx.x * FIN_CX + threadIdx.x % FIN_CX;
    const bool ok = c < C;
    double t1, t2;
    finalize_sums(part, N * chunks, (int64_t)C * 2, ok ? c : 0, ok, t1, t2);
    if (!ok || threadIdx.x >= FIN_CX) return;
    out[c] = (float)t1;
}

// --------------------------------------------------------------------------- apply kernels
// out = lrelu((y - mean) * scale (+ res))
// AFFINE (BatchNorm): out = lrelu(y * scale + shift (+ res)) with per-(n, c) scale / shift; `mean` unused
template <typename T, int VEC, bool HAS_RES, bool AFFINE = false>
__global__ __launch_bounds__(256) void in_lrelu_fwd_kernel(const T* __restrict__ y, int ldy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ scale, const T* __restrict__ res,
                                                           int ldr, T* __restrict__ out, int ldo, float slope,
                                                           ChanLoop cl, int C,
                                                           const float* __restrict__ shift = nullptr) {
    const int tid = threadIdx.x;
    const int cgl = tid % cl.Gb, vl = tid / cl.Gb;
    const int cg = blockIdx.z * cl.Gb + cgl;
    const int n = blockIdx.y;
    if (vl >= cl.vpb || cg >= cl.G) return;
    float mu[VEC], sc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) {
        mu[i] = AFFINE ? shift[n * C + cg * VEC + i] : mean[n * C + cg * VEC + i];
        sc[i] = scale[n * C + cg * VEC + i];
    }
    const int v0 = blockIdx.x * cl.span;
    int v1 = v0 + cl.span;
    if (v1 > cl.V) v1 = cl.V;
    for (int v = v0 + vl; v < v1; v += cl.vpb) {
        const int64_t row = (int64_t)n * cl.V + v;
        float yv[VEC], rv[VEC], ov[VEC];
        load_vec<T, VEC>(y + row * ldy + cg * VEC, yv);
        if (HAS_RES) load_vec<T, VEC>(res + row * ldr + cg * VEC, rv);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            float t = AFFINE ? fmaf(yv[i], sc[i], mu[i]) : (yv[i] - mu[i]) * sc[i];
            if (HAS_RES) t += rv[i];
            ov[i] = lrelu_f(t, slope);
        }
        store_vec<T, VEC>(out + row * ldo + cg * VEC, ov);
    }
}

// dy = scale * (gpre - m1 - xhat * m2), gpre = gout * lrelu'(out); optional gpre output; far planes zeroed
// FROM_GPRE: `gpre` is an INPUT (written by reduce2 MODE 4): dy from (gpre, y) alone
template <typename T, int VEC, bool HAS_GPRE, bool FROM_GPRE = false, bool AFFINE = false>
__global__ __launch_bounds__(256) void in_lrelu_bwd_kernel(const T* __restrict__ gout, int ldg,
                                                           const T* __restrict__ out, int ldo,
                                                           const T* __restrict__ y, int ldy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ m12, T* __restrict__ dy, int lddy,
                                                           T* __restrict__ gpre, int ldgp, float slope, int zero_far,
                                                           int D, int H, int W, ChanLoop cl, int C,
                                                           const float* __restrict__ shift = nullptr,
                                                           const float* __restrict__ oscale = nullptr,
                                                           double* __restrict__ dysum_part = nullptr) {
    // AFFINE (BatchNorm): xhat = y * scale + shift, dy = oscale * (...) with per-(n, c) values; `mean` unused
    // dysum_part: per-block sums of the stored dy (the bias gradient of the conv / transposed conv in front of the
    // norm is the sum of dy over samples and voxels) in reduce2_kernel's partial layout: no separate pass over dy
    __shared__ float dsh[256][VEC + 1];
    const int tid = threadIdx.x;
    const int cgl = tid % cl.Gb, vl = tid / cl.Gb;
    const int cg = blockIdx.z * cl.Gb + cgl;
    const int n = blockIdx.y;
    const bool active = vl < cl.vpb && cg < cl.G;
    float dsum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) dsum[i] = 0.f;
    if (active) {
    float mu[VEC], sc[VEC], m1[VEC], m2[VEC], osc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) {
        const int idx = n * C + cg * VEC + i;
        mu[i] = AFFINE ? shift[idx] : mean[idx];
        sc[i] = scale[idx];
        m1[i] = m12[2 * idx];
        m2[i] = m12[2 * idx + 1];
        osc[i] = AFFINE ? oscale[idx] : sc[i];
    }
    const float inv_slope = 1.f / slope;   // one division per thread, not eight per voxel
    const bool pow2 = (W & (W - 1)) == 0 && (H & (H - 1)) == 0;
    const int wshift = 31 - __clz(W);
    const int v0 = blockIdx.x * cl.span;
    int v1 = v0 + cl.span;
    if (v1 > cl.V) v1 = cl.V;
    for (int v = v0 + vl; v < v1; v += cl.vpb) {
        const int64_t row = (int64_t)n * cl.V + v;
        float gv[VEC], ov[VEC], yv[VEC], dv[VEC], pv[VEC];
        if (FROM_GPRE) {
            load_vec<T, VEC>(gpre + row * ldgp + cg * VEC, gv);
        } else {
            load_vec<T, VEC>(gout + row * ldg + cg * VEC, gv);
            load_vec<T, VEC>(out + row * ldo + cg * VEC, ov);
        }
        if (HAS_GPRE) load_vec<T, VEC>(y + row * ldy + cg * VEC, yv);   // no residual: xhat comes from `out`
        bool far = false;
        if (zero_far) {
            if (pow2) {   // extents are powers of two almost always: masks instead of three integer divisions per voxel
                far = ((v & (W - 1)) == W - 1) || (((v >> wshift) & (H - 1)) == H - 1) || (v >= (D - 1) * W * H);
            } else {
                const int w = v % W, h = (v / W) % H, d = v / (W * H);
                far = (w == W - 1) || (h == H - 1) || (d == D - 1);
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            const float gp = FROM_GPRE ? gv[i] : (ov[i] > 0.f ? gv[i] : gv[i] * slope);
            const float xh = AFFINE ? fmaf(yv[i], sc[i], mu[i])
                                    : (HAS_GPRE ? (yv[i] - mu[i]) * sc[i] : (ov[i] > 0.f ? ov[i] : ov[i] * inv_slope));
            pv[i] = gp;
            dv[i] = far ? 0.f : osc[i] * (gp - m1[i] - xh * m2[i]);
        }
        store_vec<T, VEC>(dy + row * lddy + cg * VEC, dv);
        if (HAS_GPRE && !FROM_GPRE) store_vec<T, VEC>(gpre + row * ldgp + cg * VEC, pv);
        if (dysum_part) {
#pragma unroll
            for (int i = 0; i < VEC; i++) dsum[i] += to_f32<T>(from_f32<T>(dv[i]));      // the stored value
        }
    }
    }
    if (!dysum_part) return;
#pragma unroll
    for (int i = 0; i < VEC; i++) dsh[tid][i] = active ? dsum[i] : 0.f;
    __syncthreads();
    if (vl == 0 && cg < cl.G) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            double t = 0.0;
            for (int l = 0; l < cl.vpb; l++) t += (double)dsh[l * cl.Gb + cgl][i];      // fixed order
            double* pp = dysum_part + (((int64_t)n * cl.chunks + blockIdx.x) * C + cg * VEC + i) * 2;
            pp[0] = t;
            pp[1] = 0.0;
        }
    }
}

// --------------------------------------------------------------------------- attention gate pointwise ops
// AttBlock (reference network.py:353-371): x = conv(x); g = conv(gate); rate = sigmoid(conv(lrelu(x + g))); x * rate.
// The three 1x1x1 convolutions run on the conv kernels; these are the elementwise pieces in between and their
// gradients.  OP: 0 o1 = lrelu(a)   1 o1 = a * sigmoid(b)   2 (o1, o2) = (c * sigmoid(b), c * a * sigmoid'(b))
//             3 (o1, o2) = (c * lrelu'(a), c * lrelu'(a) + b)       (a is the lrelu OUTPUT: its sign is the input's)
template <typename T, int VEC, int OP>
__global__ __launch_bounds__(256) void pointwise_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                        const T* __restrict__ c, int ldc, T* __restrict__ o1, int ld1,
                                                        T* __restrict__ o2, int ld2, float slope, ChanLoop cl) {
    const int tid = threadIdx.x;
    const int cgl = tid % cl.Gb, vl = tid / cl.Gb;
    const int cg = blockIdx.z * cl.Gb + cgl;
    const int n = blockIdx.y;
    if (vl >= cl.vpb || cg >= cl.G) return;
    const int v0 = blockIdx.x * cl.span;
    int v1 = v0 + cl.span;
    if (v1 > cl.V) v1 = cl.V;
    for (int v = v0 + vl; v < v1; v += cl.vpb) {
        const int64_t row = (int64_t)n * cl.V + v;
        float av[VEC], bv[VEC], cv[VEC], r1[VEC], r2[VEC];
        load_vec<T, VEC>(a + row * lda + cg * VEC, av);
        if (OP >= 1) load_vec<T, VEC>(b + row * ldb + cg * VEC, bv);
        if (OP >= 2) load_vec<T, VEC>(c + row * ldc + cg * VEC, cv);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
            if (OP == 0) {
                r1[i] = lrelu_f(av[i], slope);
            } else if (OP == 1) {
                r1[i] = av[i] / (1.f + __expf(-bv[i]));
            } else if (OP == 2) {
                const float sg = 1.f / (1.f + __expf(-bv[i]));
                r1[i] = cv[i] * sg;
                r2[i] = cv[i] * av[i] * sg * (1.f - sg);
            } else {
                r1[i] = av[i] > 0.f ? cv[i] : cv[i] * slope;
                r2[i] = r1[i] + bv[i];
            }
        }
        store_vec<T, VEC>(o1 + row * ld1 + cg * VEC, r1);
        if (OP >= 2) store_vec<T, VEC>(o2 + row * ld2 + cg * VEC, r2);
    }
}

// dst = a (+ b)
template <typename T, int VEC, bool ADD>
__global__ __launch_bounds__(256) void copy_add_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b,
                                                       int ldb, T* __restrict__ dst, int ldd, ChanLoop cl) {
    const int tid = threadIdx.x;
    const int cgl = tid % cl.Gb, vl = tid / cl.Gb;
    const int cg = blockIdx.z * cl.Gb + cgl;
    const int n = blockIdx.y;
    if (vl >= cl.vpb || cg >= cl.G) return;
    const int v0 = blockIdx.x * cl.span;
    int v1 = v0 + cl.span;
    if (v1 > cl.V) v1 = cl.V;
    for (int v = v0 + vl; v < v1; v += cl.vpb) {
        const int64_t row = (int64_t)n * cl.V + v;
        float av[VEC], bv[VEC];
        load_vec<T, VEC>(a + row * lda + cg * VEC, av);
        if (ADD) {
            load_vec<T, VEC>(b + row * ldb + cg * VEC, bv);
#pragma unroll
            for (int i = 0; i < VEC; i++) av[i] += bv[i];
        }
        store_vec<T, VEC>(dst + row * ldd + cg * VEC, av);
    }
}

// --------------------------------------------------------------------------- dispatch helpers
#define DISPATCH_VEC(T, vec, CALL)                 \
    switch (vec) {                                 \
        case 8: if constexpr (sizeof(T) == 2) { CALL(T, 8); } break; \
        case 4: CALL(T, 4); break;                 \
        case 2: CALL(T, 2); break;                 \
        default: CALL(T, 1); break;                \
    }

static bool same_shape(const ru3d_tensor* a, const ru3d_tensor* b) {
    return a->n == b->n && a->d == b->d && a->h == b->h && a->w == b->w && a->c == b->c;
}

static size_t reduce_ws_bytes(const ru3d_tensor* t) {
    // upper bound over the VEC choices: chunks is largest when vpb is smallest
    int64_t V = (int64_t)t->d * t->h * t->w;
    int64_t chunks = V < 4096 ? V : 4096;
    if (chunks < 1) chunks = 1;
    // reduction partials + m12 + (in_lrelu_bwd with dy_sum) the apply pass's dy-sum partials
    return 2 * (size_t)t->n * chunks * t->c * 2 * sizeof(double) + (size_t)t->n * t->c * 2 * sizeof(float) + 512;
}

#ifndef RU3D_STORAGE_F16
extern "C" size_t ru3d_reduce_workspace_bytes(const ru3d_tensor* t) { return t ? reduce_ws_bytes(t) : 0; }
#endif

template <typename T>
static int stats_impl(const ru3d_tensor* y, const float* drop, float* mean, float* scale, void* ws, float eps,
                      hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {y});
    ChanLoop cl = make_chanloop(V, y->c, vec, 64, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
    double* part = (double*)ws;
#define CALL(TT, VV)                                                                                              \
    hipLaunchKernelGGL((reduce2_kernel<TT, VV, 0>), grid, dim3(256), 0, st, (const TT*)y->ptr, y->ld, (const TT*)0, \
                       0, (const TT*)0, 0, (const float*)0, (const float*)0, 0.f, part, cl, y->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    int rc = ru3d_check_launch("instnorm_stats");
    if (rc) return rc;
    const int NC = y->n * y->c;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((NC + FIN_CX - 1) / FIN_CX), dim3(256), 0, st, (const double*)part, cl.chunks,
                       y->c, NC, 1.0 / (double)V, drop, eps, mean, scale);
    return ru3d_check_launch("instnorm_stats_finalize");
}

extern "C" int ru3d_instnorm_stats(const ru3d_tensor* y, const float* drop_scale, float* mean, float* scale, void* ws,
                                   size_t ws_bytes, float eps, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_instnorm_stats_f16(y, drop_scale, mean, scale, ws, ws_bytes, eps, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(y), "instnorm_stats: bad tensor");
    RU3D_REQUIRE(mean && scale && ws, "instnorm_stats: null output/workspace");
    RU3D_REQUIRE(ws_bytes >= reduce_ws_bytes(y), "instnorm_stats: workspace too small (%zu < %zu)", ws_bytes,
                 reduce_ws_bytes(y));
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "instnorm_stats: sample too large");
    if (dtype == RU3D_F32) return stats_impl<float>(y, drop_scale, mean, scale, ws, eps, as_stream(stream));
    if (dtype == RU3D_BF16) return stats_impl<bf16>(y, drop_scale, mean, scale, ws, eps, as_stream(stream));
    return ru3d_fail(-1, "instnorm_stats: bad dtype %d", dtype);
}

template <typename T>
static int in_fwd_impl(const ru3d_tensor* y, const float* mean, const float* scale, const ru3d_tensor* res,
                       const ru3d_tensor* out, float slope, hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {y, res, out});
    ChanLoop cl = make_chanloop(V, y->c, vec, 16, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
#define CALL(TT, VV)                                                                                               \
    if (res)                                                                                                       \
        hipLaunchKernelGGL((in_lrelu_fwd_kernel<TT, VV, true>), grid, dim3(256), 0, st, (const TT*)y->ptr, y->ld,   \
                           mean, scale, (const TT*)res->ptr, res->ld, (TT*)out->ptr, out->ld, slope, cl, y->c);    \
    else                                                                                                           \
        hipLaunchKernelGGL((in_lrelu_fwd_kernel<TT, VV, false>), grid, dim3(256), 0, st, (const TT*)y->ptr, y->ld,  \
                           mean, scale, (const TT*)0, 0, (TT*)out->ptr, out->ld, slope, cl, y->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    return ru3d_check_launch("in_lrelu_fwd");
}

extern "C" int ru3d_in_lrelu_fwd(const ru3d_tensor* y, const float* mean, const float* scale, const ru3d_tensor* res,
                                 const ru3d_tensor* out, float slope, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_in_lrelu_fwd_f16(y, mean, scale, res, out, slope, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(y) && tensor_ok(out) && same_shape(y, out), "in_lrelu_fwd: bad y/out");
    RU3D_REQUIRE(!res || (tensor_ok(res) && same_shape(y, res)), "in_lrelu_fwd: bad residual");
    RU3D_REQUIRE(mean && scale, "in_lrelu_fwd: null stats");
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "in_lrelu_fwd: sample too large");
    if (dtype == RU3D_F32) return in_fwd_impl<float>(y, mean, scale, res, out, slope, as_stream(stream));
    if (dtype == RU3D_BF16) return in_fwd_impl<bf16>(y, mean, scale, res, out, slope, as_stream(stream));
    return ru3d_fail(-1, "in_lrelu_fwd: bad dtype %d", dtype);
}

template <typename T>
static int in_bwd_impl(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y, const float* mean,
                       const float* scale, const ru3d_tensor* dy, const ru3d_tensor* gpre, void* ws, float slope,
                       int zero_far, float* gpre_sum, float* dy_sum, hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {gout, out, y, dy, gpre});
    ChanLoop cl = make_chanloop(V, y->c, vec, 64, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
    double* part = (double*)ws;
    float* m12 = (float*)((char*)ws + (size_t)y->n * cl.chunks * y->c * 2 * sizeof(double));
#define CALL(TT, VV)                                                                                                  \
    if (gpre)                                                                                                         \
        hipLaunchKernelGGL((reduce2_kernel<TT, VV, 4>), grid, dim3(256), 0, st, (const TT*)gout->ptr, gout->ld,       \
                           (const TT*)out->ptr, out->ld, (const TT*)y->ptr, y->ld, mean, scale, slope, part, cl,      \
                           y->c, (TT*)gpre->ptr, gpre->ld);                                                           \
    else                                                                                                              \
        hipLaunchKernelGGL((reduce2_kernel<TT, VV, 3>), grid, dim3(256), 0, st, (const TT*)gout->ptr, gout->ld,       \
                           (const TT*)out->ptr, out->ld, (const TT*)0, 0, mean, scale, slope, part, cl, y->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    int rc = ru3d_check_launch("in_lrelu_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bwd_finalize_kernel, dim3((y->c + FIN_CX - 1) / FIN_CX), dim3(256), 0, st, (const double*)part,
                       cl.chunks, y->c, y->n, 1.0 / (double)V, m12, (double)V, gpre_sum);
    rc = ru3d_check_launch("in_lrelu_bwd_finalize");
    if (rc) return rc;
    ChanLoop ca = make_chanloop(V, y->c, vec, 16, y->n);
    dim3 grida(ca.chunks, y->n, (ca.G + ca.Gb - 1) / ca.Gb);
    // dy-sum partials of the apply pass: behind the reduction partials and m12 (both still in use while it runs)
    double* dpart =
        dy_sum ? (double*)((char*)m12 + (((size_t)y->n * y->c * 2 * sizeof(float) + 255) / 256) * 256) : nullptr;
#define CALL(TT, VV)                                                                                                  \
    if (gpre)                                                                                                         \
        hipLaunchKernelGGL((in_lrelu_bwd_kernel<TT, VV, true, true>), grida, dim3(256), 0, st, (const TT*)gout->ptr,  \
                           gout->ld, (const TT*)out->ptr, out->ld, (const TT*)y->ptr, y->ld, mean, scale,            \
                           (const float*)m12, (TT*)dy->ptr, dy->ld, (TT*)gpre->ptr, gpre->ld, slope, zero_far, y->d,  \
                           y->h, y->w, ca, y->c, (const float*)nullptr, (const float*)nullptr, dpart);               \
    else                                                                                                              \
        hipLaunchKernelGGL((in_lrelu_bwd_kernel<TT, VV, false>), grida, dim3(256), 0, st, (const TT*)gout->ptr,       \
                           gout->ld, (const TT*)out->ptr, out->ld, (const TT*)y->ptr, y->ld, mean, scale,            \
                           (const float*)m12, (TT*)dy->ptr, dy->ld, (TT*)0, 0, slope, zero_far, y->d, y->h, y->w, ca, \
                           y->c, (const float*)nullptr, (const float*)nullptr, dpart)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    rc = ru3d_check_launch("in_lrelu_bwd_apply");
    if (rc || !dy_sum) return rc;
    hipLaunchKernelGGL(chansum_finalize_kernel, dim3((y->c + FIN_CX - 1) / FIN_CX), dim3(256), 0, st,
                       (const double*)dpart, ca.chunks, y->c, y->n, dy_sum);
    return ru3d_check_launch("in_lrelu_bwd_dysum_finalize");
}

extern "C" int ru3d_in_lrelu_bwd(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y,
                                 const float* mean, const float* scale, const ru3d_tensor* dy,
                                 const ru3d_tensor* gpre, void* ws, size_t ws_bytes, float slope, int zero_far,
                                 float* gpre_sum, float* dy_sum, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_in_lrelu_bwd_f16(gout, out, y, mean, scale, dy, gpre, ws, ws_bytes, slope, zero_far, gpre_sum, dy_sum, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(!gpre_sum || gpre, "in_lrelu_bwd: gpre_sum needs the residual form (gpre != NULL)");
    RU3D_REQUIRE(tensor_ok(gout) && tensor_ok(out) && tensor_ok(y) && tensor_ok(dy), "in_lrelu_bwd: bad tensor");
    RU3D_REQUIRE(same_shape(y, gout) && same_shape(y, out) && same_shape(y, dy), "in_lrelu_bwd: shape mismatch");
    RU3D_REQUIRE(!gpre || (tensor_ok(gpre) && same_shape(y, gpre)), "in_lrelu_bwd: bad gpre");
    RU3D_REQUIRE(mean && scale && ws, "in_lrelu_bwd: null stats/workspace");
    RU3D_REQUIRE(ws_bytes >= reduce_ws_bytes(y), "in_lrelu_bwd: workspace too small (%zu < %zu)", ws_bytes,
                 reduce_ws_bytes(y));
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "in_lrelu_bwd: sample too large");
    // the small levels: one launch instead of three (norm_small.hip)
    if (dtype == RU3D_BF16 && in_small_mode(out, gpre_sum != nullptr, gout, y, dy, gpre)) {
        int rc = in_small_bwd_launch(gout, nullptr, 0, out, y, mean, scale, dy, gpre, ws, slope, zero_far, gpre_sum, as_stream(stream));
        if (rc || !dy_sum) return rc;
        return ru3d_channel_sum(dy, dy_sum, ws, ws_bytes, dtype, stream);      // small tensors: a pass of its own
    }
    if (dtype == RU3D_F32)
        return in_bwd_impl<float>(gout, out, y, mean, scale, dy, gpre, ws, slope, zero_far, gpre_sum, dy_sum, as_stream(stream));
    if (dtype == RU3D_BF16)
        return in_bwd_impl<bf16>(gout, out, y, mean, scale, dy, gpre, ws, slope, zero_far, gpre_sum, dy_sum, as_stream(stream));
    return ru3d_fail(-1, "in_lrelu_bwd: bad dtype %d", dtype);
}

// the apply pass alone, with the two means given (ru3d_conv3d_dgrad_in_bwd: they come out of the conv's epilogue)
template <typename T>
static int in_bwd_apply_impl(const ru3d_tensor* gout, const ru3d_tensor* out, const float* mean, const float* scale,
                             const float* m12, const ru3d_tensor* dy, float slope, int zero_far, hipStream_t st) {
    const int64_t V = (int64_t)out->d * out->h * out->w;
    const int vec = pick_vec<T>(out->c, {gout, out, dy});
    ChanLoop ca = make_chanloop(V, out->c, vec, 16, out->n);
    dim3 grida(ca.chunks, out->n, (ca.G + ca.Gb - 1) / ca.Gb);
#define CALL(TT, VV)                                                                                                  \
    hipLaunchKernelGGL((in_lrelu_bwd_kernel<TT, VV, false>), grida, dim3(256), 0, st, (const TT*)gout->ptr, gout->ld, \
                       (const TT*)out->ptr, out->ld, (const TT*)out->ptr, out->ld, mean, scale, m12, (TT*)dy->ptr,    \
                       dy->ld, (TT*)0, 0, slope, zero_far, out->d, out->h, out->w, ca, out->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    return ru3d_check_launch("in_lrelu_bwd_apply");
}

extern "C" int ru3d_in_lrelu_bwd_apply(const ru3d_tensor* gout, const ru3d_tensor* out, const float* mean,
                                       const float* scale, const float* m12, const ru3d_tensor* dy, float slope,
                                       int zero_far, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_in_lrelu_bwd_apply_f16(gout, out, mean, scale, m12, dy, slope, zero_far, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(gout) && tensor_ok(out) && tensor_ok(dy), "in_lrelu_bwd_apply: bad tensor");
    RU3D_REQUIRE(same_shape(out, gout) && same_shape(out, dy), "in_lrelu_bwd_apply: shape mismatch");
    RU3D_REQUIRE(mean && scale && m12, "in_lrelu_bwd_apply: null stats");
    RU3D_REQUIRE((int64_t)out->d * out->h * out->w < (1ll << 31), "in_lrelu_bwd_apply: sample too large");
    if (dtype == RU3D_F32) return in_bwd_apply_impl<float>(gout, out, mean, scale, m12, dy, slope, zero_far, as_stream(stream));
    if (dtype == RU3D_BF16) return in_bwd_apply_impl<bf16>(gout, out, mean, scale, m12, dy, slope, zero_far, as_stream(stream));
    return ru3d_fail(-1, "in_lrelu_bwd_apply: bad dtype %d", dtype);
}

// --------------------------------------------------------------------------- BatchNorm3d, training mode
// Blocks built with norm_op=nn.BatchNorm3d (reference network.py:38-69, ResAttrBNUnet3D): statistics pooled over the
// batch.  With u = d[n][c] * y (Dropout3d's per-(n, c) factor d in front of the norm, network.py:411-413):
//     mu_c = sum_n d S1[n][c] / count,  var_c = sum_n d^2 S2[n][c] / count - mu_c^2,  r_c = 1 / sqrt(var_c + eps)
//     xhat = y * a[n][c] + b[n][c]          a = d r_c, b = -mu_c r_c
//     out  = lrelu(y * fscale + fshift (+ res))   fscale = gamma_c a, fshift = beta_c + gamma_c b
//     dy   = fscale * (gpre - M1_c - xhat * M2_c),  M1 = sum gpre / count, M2 = sum gpre xhat / count,
//     dgamma_c = sum gpre xhat, dbeta_c = sum gpre  (both over n and voxels).
// The pooled sums are handed back to the caller as doubles between the two halves of each direction, so that a
// multi-GPU caller can all-reduce them (SyncBN) - count is then the global element count.
__global__ __launch_bounds__(256) void bn_pool_kernel(const double* __restrict__ part, int chunks, int C, int N,
                                                      const float* __restrict__ drop, double* __restrict__ pooled) {
    const int c = blockIdx.x * FIN_CX + threadIdx.x % FIN_CX;
    const bool ok = c < C;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; n++) {
        double t1, t2;
        finalize_sums(part + (int64_t)n * chunks * C * 2, chunks, (int64_t)C * 2, ok ? c : 0, ok, t1, t2);
        if (ok && threadIdx.x < FIN_CX) {
            const double d = drop ? (double)drop[n * C + c] : 1.0;
            s1 += d * t1;
            s2 += d * d * t2;
        }
        __syncthreads();
    }
    if (ok && threadIdx.x < FIN_CX) {
        pooled[2 * c] = s1;
        pooled[2 * c + 1] = s2;
    }
}

__global__ void bn_m12_kernel(const double* __restrict__ pooled, int N, int C, double inv_count,
                              float* __restrict__ m12) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C;
    m12[2 * i] = (float)(pooled[2 * c] * inv_count);
    m12[2 * i + 1] = (float)(pooled[2 * c + 1] * inv_count);
}

#ifndef RU3D_STORAGE_F16
__global__ void bn_finalize_kernel(const double* __restrict__ pooled, int N, int C, int c_real, double inv_count,
                                   double unbias, const float* __restrict__ drop, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ fscale, float* __restrict__ fshift, float* __restrict__ a,
                                   float* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i % C;
    const double m = pooled[2 * c] * inv_count;
    double var = pooled[2 * c + 1] * inv_count - m * m;
    if (var < 0.0) var = 0.0;
    const double r = 1.0 / sqrt(var + (double)eps);
    const double d = drop ? (double)drop[i] : 1.0;
    const bool real = c < c_real;                    // channel padding: gamma = 1, beta = 0, no running statistics
    const double g = (real && gamma) ? (double)gamma[c] : 1.0;
    const double be = (real && beta) ? (double)beta[c] : 0.0;
    a[i] = (float)(d * r);
    b[i] = (float)(-m * r);
    fscale[i] = (float)(g * d * r);
    fshift[i] = (float)(be - g * m * r);
    if (n == 0 && real && running_mean && running_var) {
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * var * unbias);
    }
}

extern "C" int ru3d_batchnorm_stats_finalize(const double* pooled, int n, int c, int c_real, double count,
                                             const float* drop_scale, const float* gamma, const float* beta, float eps,
                                             float momentum, float* running_mean, float* running_var, float* fscale,
                                             float* fshift, float* a, float* b, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(pooled && fscale && fshift && a && b, "batchnorm_stats_finalize: null pointer");
    RU3D_REQUIRE(n > 0 && c > 0 && c_real > 0 && c_real <= c && count > 0.0, "batchnorm_stats_finalize: bad sizes");
    RU3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm_stats_finalize: half a running pair");
    const double unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((n * c + 255) / 256), dim3(256), 0, as_stream(stream), pooled, n, c, c_real,
                       1.0 / count, unbias, drop_scale, gamma, beta, eps, momentum, running_mean, running_var, fscale,
                       fshift, a, b);
    return ru3d_check_launch("batchnorm_stats_finalize");
}
#endif

template <typename T>
static int bn_pool_impl(const ru3d_tensor* y, const float* drop, double* pooled, void* ws, hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {y});
    ChanLoop cl = make_chanloop(V, y->c, vec, 64, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
    double* part = (double*)ws;
#define CALL(TT, VV)                                                                                              \
    hipLaunchKernelGGL((reduce2_kernel<TT, VV, 0>), grid, dim3(256), 0, st, (const TT*)y->ptr, y->ld, (const TT*)0, \
                       0, (const TT*)0, 0, (const float*)0, (const float*)0, 0.f, part, cl, y->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    int rc = ru3d_check_launch("batchnorm_stats_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_pool_kernel, dim3((y->c + FIN_CX - 1) / FIN_CX), dim3(256), 0, st, (const double*)part, cl.chunks,
                       y->c, y->n, drop, pooled);
    return ru3d_check_launch("batchnorm_stats_pool");
}

extern "C" int ru3d_batchnorm_stats_pool(const ru3d_tensor* y, const float* drop_scale, double* pooled, void* ws,
                                         size_t ws_bytes, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_batchnorm_stats_pool_f16(y, drop_scale, pooled, ws, ws_bytes, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(y), "batchnorm_stats_pool: bad tensor");
    RU3D_REQUIRE(pooled && ws, "batchnorm_stats_pool: null output/workspace");
    RU3D_REQUIRE(ws_bytes >= reduce_ws_bytes(y), "batchnorm_stats_pool: workspace too small (%zu < %zu)", ws_bytes,
                 reduce_ws_bytes(y));
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "batchnorm_stats_pool: sample too large");
    if (dtype == RU3D_F32) return bn_pool_impl<float>(y, drop_scale, pooled, ws, as_stream(stream));
    if (dtype == RU3D_BF16) return bn_pool_impl<bf16>(y, drop_scale, pooled, ws, as_stream(stream));
    return ru3d_fail(-1, "batchnorm_stats_pool: bad dtype %d", dtype);
}

// out = lrelu(y * scale[n][c] + shift[n][c] (+ res))
template <typename T>
static int affine_fwd_impl(const ru3d_tensor* y, const float* scale, const float* shift, const ru3d_tensor* res,
                           const ru3d_tensor* out, float slope, hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {y, res, out});
    ChanLoop cl = make_chanloop(V, y->c, vec, 16, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
#define CALL(TT, VV)                                                                                               \
    if (res)                                                                                                       \
        hipLaunchKernelGGL((in_lrelu_fwd_kernel<TT, VV, true, true>), grid, dim3(256), 0, st, (const TT*)y->ptr,    \
                           y->ld, (const float*)0, scale, (const TT*)res->ptr, res->ld, (TT*)out->ptr, out->ld,    \
                           slope, cl, y->c, shift);                                                                \
    else                                                                                                           \
        hipLaunchKernelGGL((in_lrelu_fwd_kernel<TT, VV, false, true>), grid, dim3(256), 0, st, (const TT*)y->ptr,   \
                           y->ld, (const float*)0, scale, (const TT*)0, 0, (TT*)out->ptr, out->ld, slope, cl, y->c, \
                           shift)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    return ru3d_check_launch("affine_lrelu_fwd");
}

extern "C" int ru3d_affine_lrelu_fwd(const ru3d_tensor* y, const float* scale, const float* shift, const ru3d_tensor* res,
                                     const ru3d_tensor* out, float slope, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_affine_lrelu_fwd_f16(y, scale, shift, res, out, slope, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(y) && tensor_ok(out) && same_shape(y, out), "affine_lrelu_fwd: bad y/out");
    RU3D_REQUIRE(!res || (tensor_ok(res) && same_shape(y, res)), "affine_lrelu_fwd: bad residual");
    RU3D_REQUIRE(scale && shift, "affine_lrelu_fwd: null scale/shift");
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "affine_lrelu_fwd: sample too large");
    if (dtype == RU3D_F32) return affine_fwd_impl<float>(y, scale, shift, res, out, slope, as_stream(stream));
    if (dtype == RU3D_BF16) return affine_fwd_impl<bf16>(y, scale, shift, res, out, slope, as_stream(stream));
    return ru3d_fail(-1, "affine_lrelu_fwd: bad dtype %d", dtype);
}

// backward, first half: gpre = gout * lrelu'(out) stored, pooled = (sum gpre, sum gpre * xhat) over n and voxels
template <typename T>
static int bn_bwd_pool_impl(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y, const float* a,
                            const float* b, const ru3d_tensor* gpre, double* pooled, void* ws, float slope,
                            hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {gout, out, y, gpre});
    ChanLoop cl = make_chanloop(V, y->c, vec, 64, y->n);
    dim3 grid(cl.chunks, y->n, (cl.G + cl.Gb - 1) / cl.Gb);
    double* part = (double*)ws;
#define CALL(TT, VV)                                                                                                  \
    hipLaunchKernelGGL((reduce2_kernel<TT, VV, 5>), grid, dim3(256), 0, st, (const TT*)gout->ptr, gout->ld,           \
                       (const TT*)out->ptr, out->ld, (const TT*)y->ptr, y->ld, (const float*)0, a, slope, part, cl,   \
                       y->c, (TT*)gpre->ptr, gpre->ld, b)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    int rc = ru3d_check_launch("batchnorm_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_pool_kernel, dim3((y->c + FIN_CX - 1) / FIN_CX), dim3(256), 0, st, (const double*)part, cl.chunks,
                       y->c, y->n, (const float*)0, pooled);
    return ru3d_check_launch("batchnorm_bwd_pool");
}

extern "C" int ru3d_batchnorm_bwd_pool(const ru3d_tensor* gout, const ru3d_tensor* out, const ru3d_tensor* y,
                                       const float* a, const float* b, const ru3d_tensor* gpre, double* pooled, void* ws,
                                       size_t ws_bytes, float slope, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_batchnorm_bwd_pool_f16(gout, out, y, a, b, gpre, pooled, ws, ws_bytes, slope, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(gout) && tensor_ok(out) && tensor_ok(y) && tensor_ok(gpre), "batchnorm_bwd_pool: bad tensor");
    RU3D_REQUIRE(same_shape(y, gout) && same_shape(y, out) && same_shape(y, gpre), "batchnorm_bwd_pool: shape mismatch");
    RU3D_REQUIRE(a && b && pooled && ws, "batchnorm_bwd_pool: null pointer");
    RU3D_REQUIRE(ws_bytes >= reduce_ws_bytes(y), "batchnorm_bwd_pool: workspace too small (%zu < %zu)", ws_bytes,
                 reduce_ws_bytes(y));
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "batchnorm_bwd_pool: sample too large");
    if (dtype == RU3D_F32) return bn_bwd_pool_impl<float>(gout, out, y, a, b, gpre, pooled, ws, slope, as_stream(stream));
    if (dtype == RU3D_BF16) return bn_bwd_pool_impl<bf16>(gout, out, y, a, b, gpre, pooled, ws, slope, as_stream(stream));
    return ru3d_fail(-1, "batchnorm_bwd_pool: bad dtype %d", dtype);
}

// backward, second half: dy = fscale * (gpre - M1 - xhat * M2) with M = pooled / count
static void bn_m12_launch(const double* pooled, int n, int c, double count, float* m12, hipStream_t st) {
    hipLaunchKernelGGL(bn_m12_kernel, dim3((n * c + 255) / 256), dim3(256), 0, st, pooled, n, c, 1.0 / count, m12);
}

template <typename T>
static int bn_bwd_apply_impl(const ru3d_tensor* gpre, const ru3d_tensor* y, const float* a, const float* b,
                             const float* fscale, const float* m12, const ru3d_tensor* dy, int zero_far, hipStream_t st) {
    const int64_t V = (int64_t)y->d * y->h * y->w;
    const int vec = pick_vec<T>(y->c, {gpre, y, dy});
    ChanLoop ca = make_chanloop(V, y->c, vec, 16, y->n);
    dim3 grida(ca.chunks, y->n, (ca.G + ca.Gb - 1) / ca.Gb);
#define CALL(TT, VV)                                                                                                  \
    hipLaunchKernelGGL((in_lrelu_bwd_kernel<TT, VV, true, true, true>), grida, dim3(256), 0, st, (const TT*)0, 0,      \
                       (const TT*)0, 0, (const TT*)y->ptr, y->ld, (const float*)0, a, m12, (TT*)dy->ptr, dy->ld,      \
                       (TT*)gpre->ptr, gpre->ld, 0.f, zero_far, y->d, y->h, y->w, ca, y->c, b, fscale)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    return ru3d_check_launch("batchnorm_bwd_apply");
}

extern "C" int ru3d_batchnorm_bwd_apply(const ru3d_tensor* gpre, const ru3d_tensor* y, const float* a, const float* b,
                                        const float* fscale, const double* pooled, double count, const ru3d_tensor* dy,
                                        void* ws, size_t ws_bytes, int zero_far, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_batchnorm_bwd_apply_f16(gpre, y, a, b, fscale, pooled, count, dy, ws, ws_bytes, zero_far, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(gpre) && tensor_ok(y) && tensor_ok(dy), "batchnorm_bwd_apply: bad tensor");
    RU3D_REQUIRE(same_shape(y, gpre) && same_shape(y, dy), "batchnorm_bwd_apply: shape mismatch");
    RU3D_REQUIRE(a && b && fscale && pooled && ws && count > 0.0, "batchnorm_bwd_apply: null pointer / bad count");
    RU3D_REQUIRE(ws_bytes >= (size_t)y->n * y->c * 2 * sizeof(float), "batchnorm_bwd_apply: workspace too small");
    RU3D_REQUIRE((int64_t)y->d * y->h * y->w < (1ll << 31), "batchnorm_bwd_apply: sample too large");
    float* m12 = (float*)ws;
    bn_m12_launch(pooled, y->n, y->c, count, m12, as_stream(stream));
    int rc = ru3d_check_launch("batchnorm_bwd_means");
    if (rc) return rc;
    if (dtype == RU3D_F32) return bn_bwd_apply_impl<float>(gpre, y, a, b, fscale, m12, dy, zero_far, as_stream(stream));
    if (dtype == RU3D_BF16) return bn_bwd_apply_impl<bf16>(gpre, y, a, b, fscale, m12, dy, zero_far, as_stream(stream));
    return ru3d_fail(-1, "batchnorm_bwd_apply: bad dtype %d", dtype);
}

template <typename T>
static int chansum_impl(const ru3d_tensor* t, float* out, void* ws, hipStream_t st) {
    const int64_t V = (int64_t)t->d * t->h * t->w;
    const int vec = pick_vec<T>(t->c, {t});
    ChanLoop cl = make_chanloop(V, t->c, vec, 64, t->n);
    dim3 grid(cl.chunks, t->n, (cl.G + cl.Gb - 1) / cl.Gb);
    double* part = (double*)ws;
#define CALL(TT, VV)                                                                                              \
    hipLaunchKernelGGL((reduce2_kernel<TT, VV, 2>), grid, dim3(256), 0, st, (const TT*)t->ptr, t->ld, (const TT*)0, \
                       0, (const TT*)0, 0, (const float*)0, (const float*)0, 0.f, part, cl, t->c)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    int rc = ru3d_check_launch("channel_sum");
    if (rc) return rc;
    hipLaunchKernelGGL(chansum_finalize_kernel, dim3((t->c + FIN_CX - 1) / FIN_CX), dim3(256), 0, st, (const double*)part,
                       cl.chunks, t->c, t->n, out);
    return ru3d_check_launch("channel_sum_finalize");
}

extern "C" int ru3d_channel_sum(const ru3d_tensor* t, float* out, void* ws, size_t ws_bytes, int dtype,
                                void* stream) {
    RU3D_FWD_F16(dtype, ru3d_channel_sum_f16(t, out, ws, ws_bytes, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(t) && out && ws, "channel_sum: bad argument");
    RU3D_REQUIRE(ws_bytes >= reduce_ws_bytes(t), "channel_sum: workspace too small");
    RU3D_REQUIRE((int64_t)t->d * t->h * t->w < (1ll << 31), "channel_sum: sample too large");
    if (dtype == RU3D_F32) return chansum_impl<float>(t, out, ws, as_stream(stream));
    if (dtype == RU3D_BF16) return chansum_impl<bf16>(t, out, ws, as_stream(stream));
    return ru3d_fail(-1, "channel_sum: bad dtype %d", dtype);
}

template <typename T>
static int copy_add_impl(const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* dst, hipStream_t st) {
    const int64_t V = (int64_t)a->d * a->h * a->w;
    const int vec = pick_vec<T>(a->c, {a, b, dst});
    ChanLoop cl = make_chanloop(V, a->c, vec, 16, a->n);
    dim3 grid(cl.chunks, a->n, (cl.G + cl.Gb - 1) / cl.Gb);
#define CALL(TT, VV)                                                                                                 \
    if (b)                                                                                                           \
        hipLaunchKernelGGL((copy_add_kernel<TT, VV, true>), grid, dim3(256), 0, st, (const TT*)a->ptr, a->ld,         \
                           (const TT*)b->ptr, b->ld, (TT*)dst->ptr, dst->ld, cl);                                    \
    else                                                                                                             \
        hipLaunchKernelGGL((copy_add_kernel<TT, VV, false>), grid, dim3(256), 0, st, (const TT*)a->ptr, a->ld,        \
                           (const TT*)0, 0, (TT*)dst->ptr, dst->ld, cl)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
    return ru3d_check_launch("copy_add");
}

template <typename T>
static int pointwise_impl(int op, const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* c, const ru3d_tensor* o1,
                          const ru3d_tensor* o2, float slope, hipStream_t st) {
    const int64_t V = (int64_t)a->d * a->h * a->w;
    const int vec = pick_vec<T>(a->c, {a, b, c, o1, o2});
    ChanLoop cl = make_chanloop(V, a->c, vec, 16, a->n);
    dim3 grid(cl.chunks, a->n, (cl.G + cl.Gb - 1) / cl.Gb);
#define PW(TT, VV, OPV)                                                                                                \
    hipLaunchKernelGGL((pointwise_kernel<TT, VV, OPV>), grid, dim3(256), 0, st, (const TT*)a->ptr, a->ld,                \
                       (const TT*)(b ? b->ptr : nullptr), b ? b->ld : 0, (const TT*)(c ? c->ptr : nullptr),              \
                       c ? c->ld : 0, (TT*)o1->ptr, o1->ld, (TT*)(o2 ? o2->ptr : nullptr), o2 ? o2->ld : 0, slope, cl)
#define CALL(TT, VV)              \
    if (op == 0) PW(TT, VV, 0);   \
    else if (op == 1) PW(TT, VV, 1); \
    else if (op == 2) PW(TT, VV, 2); \
    else PW(TT, VV, 3)
    DISPATCH_VEC(T, vec, CALL)
#undef CALL
#undef PW
    return ru3d_check_launch("pointwise");
}

extern "C" int ru3d_pointwise(int op, const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* c,
                              const ru3d_tensor* o1, const ru3d_tensor* o2, float slope, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_pointwise_f16(op, a, b, c, o1, o2, slope, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(op >= 0 && op <= 3, "pointwise: bad op %d", op);
    RU3D_REQUIRE(tensor_ok(a) && tensor_ok(o1) && same_shape(a, o1), "pointwise: bad a / o1");
    RU3D_REQUIRE(op < 1 || (tensor_ok(b) && same_shape(a, b)), "pointwise: op %d needs b", op);
    RU3D_REQUIRE(op < 2 || (tensor_ok(c) && same_shape(a, c) && tensor_ok(o2) && same_shape(a, o2)),
                 "pointwise: op %d needs c and o2", op);
    RU3D_REQUIRE((int64_t)a->d * a->h * a->w < (1ll << 31), "pointwise: sample too large");
    if (dtype == RU3D_F32) return pointwise_impl<float>(op, a, b, c, o1, o2, slope, as_stream(stream));
    if (dtype == RU3D_BF16) return pointwise_impl<bf16>(op, a, b, c, o1, o2, slope, as_stream(stream));
    return ru3d_fail(-1, "pointwise: bad dtype %d", dtype);
}

extern "C" int ru3d_copy_channels(const ru3d_tensor* src, const ru3d_tensor* dst, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_copy_channels_f16(src, dst, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(src) && tensor_ok(dst) && same_shape(src, dst), "copy_channels: bad tensors");
    RU3D_REQUIRE((int64_t)src->d * src->h * src->w < (1ll << 31), "copy_channels: sample too large");
    if (dtype == RU3D_F32) return copy_add_impl<float>(src, nullptr, dst, as_stream(stream));
    if (dtype == RU3D_BF16) return copy_add_impl<bf16>(src, nullptr, dst, as_stream(stream));
    return ru3d_fail(-1, "copy_channels: bad dtype %d", dtype);
}

extern "C" int ru3d_add(const ru3d_tensor* a, const ru3d_tensor* b, const ru3d_tensor* dst, int dtype,
                        void* stream) {
    RU3D_FWD_F16(dtype, ru3d_add_f16(a, b, dst, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(a) && tensor_ok(b) && tensor_ok(dst) && same_shape(a, b) && same_shape(a, dst),
                 "add: bad tensors");
    RU3D_REQUIRE((int64_t)a->d * a->h * a->w < (1ll << 31), "add: sample too large");
    if (dtype == RU3D_F32) return copy_add_impl<float>(a, b, dst, as_stream(stream));
    if (dtype == RU3D_BF16) return copy_add_impl<bf16>(a, b, dst, as_stream(stream));
    return ru3d_fail(-1, "add: bad dtype %d", dtype);
}

// fp32 -> storage dtype (small tensors: the 2..4-channel logits gradient)
template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int C,
                                int64_t rows) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        dst[r * ldd + c] = from_f32<T>(src[r * lds + c]);
    }
}

extern "C" int ru3d_cast_f32(const ru3d_tensor* src, const ru3d_tensor* dst, int dst_dtype, void* stream) {
    RU3D_FWD_F16(dst_dtype, ru3d_cast_f32_f16(src, dst, dst_dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(src) && tensor_ok(dst) && same_shape(src, dst), "cast_f32: bad tensors");
    const int64_t rows = nvox(src);
    int64_t b = (rows * src->c + 255) / 256;
    if (b > 8192) b = 8192;
    if (dst_dtype == RU3D_F32)
        hipLaunchKernelGGL(cast_f32_kernel<float>, dim3((unsigned)b), dim3(256), 0, as_stream(stream),
                           (const float*)src->ptr, src->ld, (float*)dst->ptr, dst->ld, src->c, rows);
    else if (dst_dtype == RU3D_BF16)
        hipLaunchKernelGGL(cast_f32_kernel<bf16>, dim3((unsigned)b), dim3(256), 0, as_stream(stream),
                           (const float*)src->ptr, src->ld, (bf16*)dst->ptr, dst->ld, src->c, rows);
    else
        return ru3d_fail(-1, "cast_f32: bad dtype %d", dst_dtype);
    return ru3d_check_launch("cast_f32");
}

// --------------------------------------------------------------------------- dropout mask
// Counter-based generator (splitmix64 of (seed, offset + i)); one draw per (n, c) volume.
__global__ void dropout_scale_kernel(float* __restrict__ scale, int count, float p, uint64_t seed,
                                     uint64_t offset, const uint64_t* __restrict__ offset_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (offset_base) offset += *offset_base;
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (offset + (uint64_t)i + 1) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0,1)
    scale[i] = (u >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

#ifndef RU3D_STORAGE_F16
extern "C" int ru3d_dropout3d_scale(float* scale, int count, float p, uint64_t seed, uint64_t offset,
                                    void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(scale && count > 0 && p >= 0.f && p < 1.f, "dropout3d_scale: bad argument");
    hipLaunchKernelGGL(dropout_scale_kernel, dim3((count + 255) / 256), dim3(256), 0, as_stream(stream), scale, count,
                       p, seed, offset, (const uint64_t*)nullptr);
    return ru3d_check_launch("dropout3d_scale");
}

extern "C" int ru3d_dropout3d_scale_dev(float* scale, int count, float p, uint64_t seed, uint64_t offset,
                                        const uint64_t* offset_base, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(scale && offset_base && count > 0 && p >= 0.f && p < 1.f, "dropout3d_scale_dev: bad argument");
    hipLaunchKernelGGL(dropout_scale_kernel, dim3((count + 255) / 256), dim3(256), 0, as_stream(stream), scale, count,
                       p, seed, offset, offset_base);
    return ru3d_check_launch("dropout3d_scale_dev");
}
#endif

// --------------------------------------------------------------------------- NCDHW <-> NDHWC
// Tiled transpose through LDS: [C][V] <-> [V][C] per sample; 32x32 tiles, padded rows.
template <typename T, bool TO_CL>
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ ncdhw_in, float* __restrict__ ncdhw_out,
                                                     const T* __restrict__ cl_in, T* __restrict__ cl_out, int ld,
                                                     int C, int64_t V) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int64_t v0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
    if (TO_CL) {
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r;
            const int64_t v = v0 + tx;
            tile[r][tx] = (c < C && v < V) ? ncdhw_in[((int64_t)n * C + c) * V + v] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int64_t v = v0 + r;
            const int c = c0 + tx;
            if (c < C && v < V) cl_out[((int64_t)n * V + v) * ld + c] = from_f32<T>(tile[tx][r]);
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int64_t v = v0 + r;
            const int c = c0 + tx;
            tile[r][tx] = (c < C && v < V) ? to_f32<T>(cl_in[((int64_t)n * V + v) * ld + c]) : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r;
            const int64_t v = v0 + tx;
            if (c < C && v < V) ncdhw_out[((int64_t)n * C + c) * V + v] = tile[tx][r];
        }
    }
}

extern "C" int ru3d_ncdhw_to_ndhwc(const float* src, const ru3d_tensor* dst, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_ncdhw_to_ndhwc_f16(src, dst, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(src && tensor_ok(dst), "ncdhw_to_ndhwc: bad argument");
    const int64_t V = (int64_t)dst->d * dst->h * dst->w;
    dim3 grid((unsigned)((V + 31) / 32), (dst->c + 31) / 32, dst->n);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL((repack_kernel<float, true>), grid, dim3(256), 0, as_stream(stream), src, (float*)0,
                           (const float*)0, (float*)dst->ptr, dst->ld, dst->c, V);
    else if (dtype == RU3D_BF16)
        hipLaunchKernelGGL((repack_kernel<bf16, true>), grid, dim3(256), 0, as_stream(stream), src, (float*)0,
                           (const bf16*)0, (bf16*)dst->ptr, dst->ld, dst->c, V);
    else
        return ru3d_fail(-1, "ncdhw_to_ndhwc: bad dtype %d", dtype);
    return ru3d_check_launch("ncdhw_to_ndhwc");
}

extern "C" int ru3d_ndhwc_to_ncdhw(const ru3d_tensor* src, float* dst, int dtype, void* stream) {
    RU3D_FWD_F16(dtype, ru3d_ndhwc_to_ncdhw_f16(src, dst, dtype, stream));
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(dst && tensor_ok(src), "ndhwc_to_ncdhw: bad argument");
    const int64_t V = (int64_t)src->d * src->h * src->w;
    dim3 grid((unsigned)((V + 31) / 32), (src->c + 31) / 32, src->n);
    if (dtype == RU3D_F32)
        hipLaunchKernelGGL((repack_kernel<float, false>), grid, dim3(256), 0, as_stream(stream), (const float*)0, dst,
                           (const float*)src->ptr, (float*)0, src->ld, src->c, V);
    else if (dtype == RU3D_BF16)
        hipLaunchKernelGGL((repack_kernel<bf16, false>), grid, dim3(256), 0, as_stream(stream), (const float*)0, dst,
                           (const bf16*)src->ptr, (bf16*)0, src->ld, src->c, V);
    else
        return ru3d_fail(-1, "ndhwc_to_ncdhw: bad dtype %d", dtype);
    return ru3d_check_launch("ndhwc_to_ncdhw");
}

}  // namespace RU3D_NS
