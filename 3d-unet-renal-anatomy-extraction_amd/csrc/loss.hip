// Fused softmax + focal + Tversky loss (forward sums, on-device finalize, backward) and the
// functional Tversky `dice`.  One read of logits + labels per pass; per-thread fp32 partials ->
// wave shuffles -> LDS -> one double partial per block -> fixed-order finalize (deterministic, no
// atomics, no host sync).
//
// Reference arithmetic restated (loss.py): with p = softmax(z) (sigmoid when C == 1), g = one_hot(t):
//   tp_c = sum p_c g_c, fn_c = sum (1-p_c) g_c = sg_c - tp_c, fp_c = sum p_c (1-g_c) = sp_c - tp_c
//   dice_c  = (tp_c + s) / (tp_c + alpha fn_c + beta fp_c + s)                      loss.py:32-48
//   focal_c = C * mean_v( -(1-p_c)^gamma g_c log p_c )                              loss.py:78-79, 240-241
//   w       = weight_v / |weight_v|_1   (weight_c and the presence mask are dead)   loss.py:69,155,237
//   Hybird  = sum_c w_c (1 - dice_c + focal_c); DiceLoss = sum w (1 - dice); Focal = sum w focal;
//   Dice    = sum w dice
#include "common.h"
#include <stddef.h>

#define RU3D_MAX_CLASSES 8

struct LossState {
    double sums[4][RU3D_MAX_CLASSES];  // tp, sp, sg, foc
    float qa[RU3D_MAX_CLASSES];        // dL/dp_c = qa_c * g_c + qb_c  (+ focal term)
    float qb[RU3D_MAX_CLASSES];
    float qf[RU3D_MAX_CLASSES];        // focal coefficient w_c * C / (N V)
    float loss;
    int bad_labels;
    int pad[2];
};

extern "C" size_t ru3d_loss_state_bytes(int num_classes) {
    (void)num_classes;
    return sizeof(LossState);
}

extern "C" size_t ru3d_loss_state_bad_labels_offset(void) { return offsetof(LossState, bad_labels); }

static int loss_blocks(int n, int64_t v) {
    int64_t total = (int64_t)n * v;
    int64_t b = (total + 256 * 8 - 1) / (256 * 8);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" size_t ru3d_loss_workspace_bytes(int n, int64_t v, int num_classes) {
    (void)num_classes;
    return (size_t)loss_blocks(n, v) * (4 * RU3D_MAX_CLASSES + 1) * sizeof(double);
}

__device__ __forceinline__ int load_label(const void* labels, int label_dtype, int64_t i) {
    if (label_dtype == RU3D_LABEL_I64) return (int)((const int64_t*)labels)[i];
    return (int)((const uint8_t*)labels)[i];
}

__device__ __forceinline__ float pow_gamma(float base, float gamma) {
    if (gamma == 2.f) return base * base;
    if (gamma == 1.f) return base;
    if (gamma == 0.f) return 1.f;
    return powf(base, gamma);
}

// probabilities + log-probabilities of one voxel (softmax over C, sigmoid for C == 1)
template <int C>
__device__ __forceinline__ void voxel_probs(const float* __restrict__ z, int64_t stride_c, float (&p)[C],
                                            float (&lp)[C]) {
    float zz[C];
#pragma unroll
    for (int c = 0; c < C; c++) zz[c] = z[c * stride_c];
    if (C == 1) {
        // F.sigmoid / torch.log(pt)  (loss.py:227-228)
        const float pr = 1.f / (1.f + __expf(-zz[0]));
        p[0] = pr;
        lp[0] = logf(pr);
        return;
    }
    float m = zz[0];
#pragma unroll
    for (int c = 1; c < C; c++) m = fmaxf(m, zz[c]);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) se += expf(zz[c] - m);
    const float lse = logf(se);
#pragma unroll
    for (int c = 0; c < C; c++) {
        lp[c] = zz[c] - m - lse;
        p[c] = expf(lp[c]);
    }
}

template <int C>
__global__ __launch_bounds__(256) void loss_sums_kernel(const float* __restrict__ logits, int64_t stride_n,
                                                        int64_t stride_c, int64_t stride_v,
                                                        const void* __restrict__ labels, int label_dtype, int n,
                                                        int64_t v, float gamma, double* __restrict__ part) {
    float tp[C], sp[C], sg[C], fo[C];
#pragma unroll
    for (int c = 0; c < C; c++) tp[c] = sp[c] = sg[c] = fo[c] = 0.f;
    int bad = 0;
    const int64_t total = (int64_t)n * v;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t ni = i / v, vi = i - ni * v;
        float p[C], lp[C];
        voxel_probs<C>(logits + ni * stride_n + vi * stride_v, stride_c, p, lp);
        int t = load_label(labels, label_dtype, i);
        if (t < 0 || t >= C) {
            bad++;
            t = -1;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            sp[c] += p[c];
            if (c == t) {
                tp[c] += p[c];
                sg[c] += 1.f;
                fo[c] += -pow_gamma(1.f - p[c], gamma) * lp[c];
            }
        }
    }
    __shared__ double sh[4][4 * C + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float a = wave_sum(tp[c]), b = wave_sum(sp[c]), d = wave_sum(sg[c]), e = wave_sum(fo[c]);
        if (lane == 0) {
            sh[wid][0 * C + c] = a;
            sh[wid][1 * C + c] = b;
            sh[wid][2 * C + c] = d;
            sh[wid][3 * C + c] = e;
        }
    }
    const float fb = wave_sum((float)bad);
    if (lane == 0) sh[wid][4 * C] = fb;
    __syncthreads();
    if (threadIdx.x < 4 * C + 1) {
        const int q = threadIdx.x;
        const double s = sh[0][q] + sh[1][q] + sh[2][q] + sh[3][q];
        // partial layout: [block][4*MAX + 1]
        const int dst = (q == 4 * C) ? 4 * RU3D_MAX_CLASSES : (q / C) * RU3D_MAX_CLASSES + (q % C);
        part[(int64_t)blockIdx.x * (4 * RU3D_MAX_CLASSES + 1) + dst] = s;
    }
}

struct LossParams {
    int kind, C, n;
    int64_t v;
    float gamma, alpha, beta, smooth;
    float w[RU3D_MAX_CLASSES];  // weight_v (un-normalised); all ones when the caller passed NULL
};

// One workgroup of 1024 threads: thread (g, q) sums the partials of quantity q over the blocks b = g, g + NG, ... (a
// wave-load covers consecutive q of one block: coalesced), then the NG group sums of a quantity are added in group order
// - fixed order, deterministic.  (Until round 4 the Q = 33 quantities were reduced one after the other, each a strided
// read and an 8-step block reduction: 42 us for 540 KB.)
constexpr int LF_THREADS = 1024;
__global__ __launch_bounds__(LF_THREADS) void loss_finalize_kernel(const double* __restrict__ part, int blocks,
                                                                   LossParams P, LossState* __restrict__ st,
                                                                   float* __restrict__ loss_out) {
    constexpr int Q = 4 * RU3D_MAX_CLASSES + 1;
    constexpr int NG = LF_THREADS / Q;
    __shared__ double red[NG][Q];
    __shared__ double tot[Q];
    {
        const int g = threadIdx.x / Q, q = threadIdx.x % Q;
        if (g < NG) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int b = g;
            for (; b + 3 * NG < blocks; b += 4 * NG) {
                s0 += part[(int64_t)b * Q + q];
                s1 += part[(int64_t)(b + NG) * Q + q];
                s2 += part[(int64_t)(b + 2 * NG) * Q + q];
                s3 += part[(int64_t)(b + 3 * NG) * Q + q];
            }
            for (; b < blocks; b += NG) s0 += part[(int64_t)b * Q + q];
            red[g][q] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (threadIdx.x < Q) {
            const int qq = threadIdx.x, c = qq % RU3D_MAX_CLASSES;
            double t = 0.0;
            if (!(qq < 4 * RU3D_MAX_CLASSES && c >= P.C))
                for (int k = 0; k < NG; k++) t += red[k][qq];
            tot[qq] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const int C = P.C;
    double wsum = 0.0;
    for (int c = 0; c < C; c++) wsum += fabs((double)P.w[c]);
    if (wsum < 1e-12) wsum = 1e-12;  // F.normalize eps
    const double NV = (double)P.n * (double)P.v;
    const bool has_dice = P.kind != RU3D_LOSS_FOCAL;
    const bool has_focal = (P.kind == RU3D_LOSS_HYBIRD) || (P.kind == RU3D_LOSS_FOCAL);
    const double dsign = (P.kind == RU3D_LOSS_DICE) ? -1.0 : 1.0;  // loss = const - dsign * dice
    const double dconst = (P.kind == RU3D_LOSS_HYBIRD || P.kind == RU3D_LOSS_DICELOSS) ? 1.0 : 0.0;
    double loss = 0.0;
    for (int c = 0; c < RU3D_MAX_CLASSES; c++) {
        st->qa[c] = st->qb[c] = st->qf[c] = 0.f;
        for (int k = 0; k < 4; k++) st->sums[k][c] = tot[k * RU3D_MAX_CLASSES + c];
    }
    for (int c = 0; c < C; c++) {
        const double w = (double)P.w[c] / wsum;
        const double tp = tot[0 * RU3D_MAX_CLASSES + c], sp = tot[1 * RU3D_MAX_CLASSES + c],
                     sg = tot[2 * RU3D_MAX_CLASSES + c], fo = tot[3 * RU3D_MAX_CLASSES + c];
        double term = 0.0;
        if (has_dice) {
            const double a = P.alpha, b = P.beta, s = P.smooth;
            const double den = tp + a * (sg - tp) + b * (sp - tp) + s;
            const double dice = (tp + s) / den;
            term += dconst - dsign * dice;
            // d dice / d p_c(v) = g * A - B
            const double A = (den - (tp + s) * (1.0 - a - b)) / (den * den);
            const double B = (tp + s) * b / (den * den);
            st->qa[c] = (float)(-w * dsign * A);
            st->qb[c] = (float)(w * dsign * B);
        }
        if (has_focal) {
            term += (double)C * fo / NV;
            st->qf[c] = (float)(w * (double)C / NV);
        }
        loss += w * term;
    }
    st->bad_labels = (int)tot[4 * RU3D_MAX_CLASSES];
    if (st->bad_labels > 0) loss = nan("");  // F.one_hot would have raised (loss.py:27)
    st->loss = (float)loss;
    loss_out[0] = (float)loss;
}

template <int C, typename TG>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits, int64_t stride_n,
                                                       int64_t stride_c, int64_t stride_v,
                                                       const void* __restrict__ labels, int label_dtype, int n,
                                                       int64_t v, float gamma, const LossState* __restrict__ st,
                                                       const float* __restrict__ grad_out, TG* __restrict__ dz) {
    float qa[C], qb[C], qf[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        qa[c] = st->qa[c];
        qb[c] = st->qb[c];
        qf[c] = st->qf[c];
    }
    const float go = grad_out ? grad_out[0] : 1.f;
    const int64_t total = (int64_t)n * v;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t ni = i / v, vi = i - ni * v;
        const int64_t base = ni * stride_n + vi * stride_v;
        float p[C], lp[C], u[C];
        voxel_probs<C>(logits + base, stride_c, p, lp);
        const int t = load_label(labels, label_dtype, i);
        float su = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            // u_c = p_c * dL/dp_c, written so that p -> 0 stays finite
            float uc = p[c] * qb[c];
            if (c == t) {
                const float om = 1.f - p[c];
                uc += p[c] * qa[c];
                float dfp;  // p * d/dp[ -(1-p)^g log p ] = g (1-p)^(g-1) p log p - (1-p)^g
                if (gamma == 2.f)
                    dfp = 2.f * om * p[c] * lp[c] - om * om;
                else if (gamma == 0.f)
                    dfp = -1.f;
                else
                    dfp = gamma * powf(om, gamma - 1.f) * p[c] * lp[c] - powf(om, gamma);
                uc += qf[c] * dfp;
            }
            u[c] = uc;
            su += uc;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            float d;
            if (C == 1)
                d = u[0] * (1.f - p[0]);  // sigmoid: dp/dz = p (1 - p)
            else
                d = u[c] - p[c] * su;
            dz[base + c * stride_c] = from_f32<TG>(d * go);
        }
    }
}

#define LOSS_DISPATCH_C(C, CALL) \
    switch (C) {                 \
        case 1: CALL(1); break;  \
        case 2: CALL(2); break;  \
        case 3: CALL(3); break;  \
        case 4: CALL(4); break;  \
        case 5: CALL(5); break;  \
        case 6: CALL(6); break;  \
        case 7: CALL(7); break;  \
        default: CALL(8); break; \
    }

extern "C" int ru3d_loss_fwd(const float* logits, int64_t stride_n, int64_t stride_c, int64_t stride_v,
                             const void* labels, int label_dtype, int n, int64_t v, int num_classes, int kind,
                             float gamma, const float* weight_v, float alpha, float beta, float smooth, void* state,
                             float* loss_out, void* ws, size_t ws_bytes, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(logits && labels && state && loss_out && ws, "loss_fwd: null pointer");
    RU3D_REQUIRE(n > 0 && v > 0, "loss_fwd: empty input");
    RU3D_REQUIRE(num_classes >= 1 && num_classes <= RU3D_MAX_CLASSES, "loss_fwd: %d classes unsupported (max %d)",
                 num_classes, RU3D_MAX_CLASSES);
    RU3D_REQUIRE(kind >= 0 && kind <= 3, "loss_fwd: bad kind %d", kind);
    RU3D_REQUIRE(label_dtype == RU3D_LABEL_I64 || label_dtype == RU3D_LABEL_U8, "loss_fwd: bad label dtype");
    RU3D_REQUIRE(ws_bytes >= ru3d_loss_workspace_bytes(n, v, num_classes), "loss_fwd: workspace too small");
    hipStream_t st = as_stream(stream);
    const int blocks = loss_blocks(n, v);
#define CALL(CC)                                                                                                    \
    hipLaunchKernelGGL(loss_sums_kernel<CC>, dim3(blocks), dim3(256), 0, st, logits, stride_n, stride_c, stride_v, \
                       labels, label_dtype, n, v, gamma, (double*)ws)
    LOSS_DISPATCH_C(num_classes, CALL)
#undef CALL
    int rc = ru3d_check_launch("loss_sums");
    if (rc) return rc;
    LossParams P;
    P.kind = kind;
    P.C = num_classes;
    P.n = n;
    P.v = v;
    P.gamma = gamma;
    P.alpha = alpha;
    P.beta = beta;
    P.smooth = smooth;
    for (int c = 0; c < RU3D_MAX_CLASSES; c++) P.w[c] = (c < num_classes) ? (weight_v ? weight_v[c] : 1.f) : 0.f;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LF_THREADS), 0, st, (const double*)ws, blocks, P,
                       (LossState*)state, loss_out);
    return ru3d_check_launch("loss_finalize");
}

extern "C" int ru3d_loss_bwd(const float* logits, int64_t stride_n, int64_t stride_c, int64_t stride_v,
                             const void* labels, int label_dtype, int n, int64_t v, int num_classes, float gamma,
                             const void* state, const float* grad_out, void* dlogits, int dlogits_dtype,
                             void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(logits && labels && state && dlogits, "loss_bwd: null pointer");
    RU3D_REQUIRE(n > 0 && v > 0, "loss_bwd: empty input");
    RU3D_REQUIRE(num_classes >= 1 && num_classes <= RU3D_MAX_CLASSES, "loss_bwd: %d classes unsupported",
                 num_classes);
    RU3D_REQUIRE(dlogits_dtype == RU3D_F32 || dlogits_dtype == RU3D_BF16, "loss_bwd: bad dlogits dtype");
    hipStream_t st = as_stream(stream);
    int64_t total = (int64_t)n * v;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
#define CALL(CC)                                                                                                    \
    if (dlogits_dtype == RU3D_F32)                                                                                  \
        hipLaunchKernelGGL((loss_bwd_kernel<CC, float>), dim3(blocks), dim3(256), 0, st, logits, stride_n, stride_c, \
                           stride_v, labels, label_dtype, n, v, gamma, (const LossState*)state, grad_out,           \
                           (float*)dlogits);                                                                        \
    else                                                                                                            \
        hipLaunchKernelGGL((loss_bwd_kernel<CC, bf16>), dim3(blocks), dim3(256), 0, st, logits, stride_n, stride_c,  \
                           stride_v, labels, label_dtype, n, v, gamma, (const LossState*)state, grad_out,           \
                           (bf16*)dlogits)
    LOSS_DISPATCH_C(num_classes, CALL)
#undef CALL
    return ru3d_check_launch("loss_bwd");
}

// --------------------------------------------------------------------------- functional dice
__global__ __launch_bounds__(256) void tversky_sums_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                           int64_t count, double* __restrict__ part) {
    float tp = 0.f, sp = 0.f, sg = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float a = p[i], b = g[i];
        tp = fmaf(a, b, tp);
        sp += a;
        sg += b;
    }
    __shared__ double sh[4][3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float a = wave_sum(tp), b = wave_sum(sp), c = wave_sum(sg);
    if (lane == 0) {
        sh[wid][0] = a;
        sh[wid][1] = b;
        sh[wid][2] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[(int64_t)blockIdx.x * 3 + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ void tversky_finalize_kernel(const double* __restrict__ part, int blocks, float alpha, float beta,
                                        float smooth, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double tp = 0.0, sp = 0.0, sg = 0.0;
    for (int b = 0; b < blocks; b++) {
        tp += part[b * 3];
        sp += part[b * 3 + 1];
        sg += part[b * 3 + 2];
    }
    out[0] = (float)((tp + smooth) / (tp + alpha * (sg - tp) + beta * (sp - tp) + smooth));
}

extern "C" int ru3d_tversky(const float* p, const float* g, int64_t count, float alpha, float beta, float smooth,
                            float* out, void* ws, size_t ws_bytes, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(p && g && out && ws && count > 0, "tversky: bad argument");
    int blocks = (int)((count + 2047) / 2048);
    if (blocks > 1024) blocks = 1024;
    RU3D_REQUIRE(ws_bytes >= (size_t)blocks * 3 * sizeof(double), "tversky: workspace too small");
    hipLaunchKernelGGL(tversky_sums_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, g, count, (double*)ws);
    int rc = ru3d_check_launch("tversky_sums");
    if (rc) return rc;
    hipLaunchKernelGGL(tversky_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), (const double*)ws, blocks,
                       alpha, beta, smooth, out);
    return ru3d_check_launch("tversky_finalize");
}

// --------------------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        // torch.optim.Adam: denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

extern "C" int ru3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                              float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                              float grad_scale, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(param && grad && exp_avg && exp_avg_sq && count > 0, "adam_step: bad argument");
    int64_t b = (count + 1023) / 1024;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, as_stream(stream), param, grad, exp_avg,
                       exp_avg_sq, count, lr, beta1, beta2, eps, bias_corr1, sqrtf(bias_corr2), grad_scale);
    return ru3d_check_launch("adam_step");
}

// multi-tensor form: one launch for the whole model (device-side tensor table + block map)
__device__ __forceinline__ void adam_multi_body(const ru3d_adam_tensor* __restrict__ tensors,
                                                const int32_t* __restrict__ block_map, int chunk_elems, float lr,
                                                float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    // no fused multiply-adds here: this body is compiled into two kernels (scalars as arguments / from device memory) and
    // into a vector and a scalar path - left to the compiler, the contraction of b2 * v + (1 - b2) * g * g differed
    // between them by one ulp, on the one parameter whose length is not a multiple of 4 (the head's bias)
#pragma clang fp contract(off)
    const ru3d_adam_tensor t = tensors[block_map[2 * blockIdx.x]];
    if (!t.grad) return;
    const int64_t begin = (int64_t)block_map[2 * blockIdx.x + 1] * chunk_elems;
    int64_t end = begin + chunk_elems;
    if (end > t.count) end = t.count;
    const float step = lr / bc1;
    const bool vec = ((((uintptr_t)t.param) | ((uintptr_t)t.grad) | ((uintptr_t)t.exp_avg) | ((uintptr_t)t.exp_avg_sq)) & 15) == 0;
    int64_t i = begin + (int64_t)threadIdx.x * 4;
    if (vec) {
        for (; i + 3 < end; i += 1024) {
            f32x4 p = *reinterpret_cast<const f32x4*>(t.param + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(t.grad + i);
            f32x4 m = *reinterpret_cast<const f32x4*>(t.exp_avg + i);
            f32x4 v = *reinterpret_cast<const f32x4*>(t.exp_avg_sq + i);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float gi = g[k] * gscale;
                m[k] = b1 * m[k] + (1.f - b1) * gi;
                v[k] = b2 * v[k] + (1.f - b2) * gi * gi;
                p[k] -= step * (m[k] / (sqrtf(v[k]) / bc2_sqrt + eps));
            }
            *reinterpret_cast<f32x4*>(t.param + i) = p;
            *reinterpret_cast<f32x4*>(t.exp_avg + i) = m;
            *reinterpret_cast<f32x4*>(t.exp_avg_sq + i) = v;
        }
    }
    // scalar tail (or unaligned tensors): this thread's remaining elements of its 4-wide slots
    for (; i < end; i += 1024)
        for (int k = 0; k < 4 && i + k < end; k++) {
            const float gi = t.grad[i + k] * gscale;
            const float mi = b1 * t.exp_avg[i + k] + (1.f - b1) * gi;
            const float vi = b2 * t.exp_avg_sq[i + k] + (1.f - b2) * gi * gi;
            t.exp_avg[i + k] = mi;
            t.exp_avg_sq[i + k] = vi;
            t.param[i + k] -= step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
        }
}


__global__ __launch_bounds__(256) void adam_multi_kernel(const ru3d_adam_tensor* __restrict__ tensors,
                                                         const int32_t* __restrict__ block_map, int chunk_elems,
                                                         float lr, float b1, float b2, float eps, float bc1,
                                                         float bc2_sqrt, float gscale) {
    adam_multi_body(tensors, block_map, chunk_elems, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
}

// the per-step scalars from device memory (a captured launch: see ru3d.h)
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const ru3d_adam_tensor* __restrict__ tensors,
                                                             const int32_t* __restrict__ block_map, int chunk_elems,
                                                             const float* __restrict__ hyper) {
    adam_multi_body(tensors, block_map, chunk_elems, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[7], hyper[6]);
}

extern "C" int ru3d_adam_multi(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                               float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                               float grad_scale, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensors && block_map && nblocks > 0 && chunk_elems >= 1024 && (chunk_elems % 1024) == 0,
                 "adam_multi: bad argument (chunk_elems must be a positive multiple of 1024)");
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), tensors, block_map,
                       chunk_elems, lr, beta1, beta2, eps, bias_corr1, sqrtf(bias_corr2), grad_scale);
    return ru3d_check_launch("adam_multi");
}

extern "C" int ru3d_adam_multi_dev(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks,
                                   int chunk_elems, const float* hyper, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensors && block_map && hyper && nblocks > 0 && chunk_elems >= 1024 && (chunk_elems % 1024) == 0,
                 "adam_multi_dev: bad argument (chunk_elems must be a positive multiple of 1024)");
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), tensors,
                       block_map, chunk_elems, hyper);
    return ru3d_check_launch("adam_multi_dev");
}

// ---- fp16 training inside a captured step: the loss scaler lives on the device (ru3d_amp_state, see ru3d.h).  The update
// kernel skips itself when the gradient check found an overflow, takes 1 / scale and the number of steps really taken
// from the state block (bias corrections from that count), and a one-thread kernel then moves the scaler: halve + reset
// on overflow, count a clean step and double after `growth_interval` of them otherwise - apex's schedule
// (reference trainer.py:492-493, 538-542), without the per-step read-back that kept the fp16 step out of a hipGraph.
__global__ __launch_bounds__(256) void adam_multi_amp_kernel(const ru3d_adam_tensor* __restrict__ tensors,
                                                             const int32_t* __restrict__ block_map, int chunk_elems,
                                                             const float* __restrict__ hyper,
                                                             const ru3d_amp_state* __restrict__ amp) {
    if (amp->found_inf != 0.f) return;                                   // overflow: the step is skipped
    const int t = __float_as_int(hyper[5]) + amp->steps + 1;             // Adam step number of this update
    // the betas in double (float value + residual in the slots the captured amp launch does not use otherwise): the bias
    // corrections then equal the host's `1 - beta ** t` to the last bit or two
    const double b1d = (double)hyper[1] + (double)hyper[4], b2d = (double)hyper[2] + (double)hyper[7];
    const double bc1 = 1.0 - pow(b1d, (double)t), bc2 = 1.0 - pow(b2d, (double)t);
    adam_multi_body(tensors, block_map, chunk_elems, hyper[0], hyper[1], hyper[2], hyper[3], (float)bc1, sqrtf((float)bc2),
                    amp->inv_scale);
}

__global__ void amp_update_kernel(ru3d_amp_state* amp, float growth, float backoff, int interval, float min_scale,
                                  float max_scale) {
    if (threadIdx.x || blockIdx.x) return;
    if (amp->found_inf != 0.f) {
        amp->scale = fmaxf(amp->scale * backoff, min_scale);
        amp->tracker = 0;
        amp->skipped += 1;
    } else {
        amp->steps += 1;
        amp->tracker += 1;
        if (amp->tracker >= interval) {
            amp->scale = fminf(amp->scale * growth, max_scale);
            amp->tracker = 0;
        }
    }
    amp->inv_scale = 1.f / amp->scale;
    amp->found_inf = 0.f;
}

extern "C" int ru3d_adam_multi_amp(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks, int chunk_elems,
                                   const float* hyper, const ru3d_amp_state* amp, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensors && block_map && hyper && amp && nblocks > 0 && chunk_elems >= 1024 && (chunk_elems % 1024) == 0,
                 "adam_multi_amp: bad argument (chunk_elems must be a positive multiple of 1024)");
    hipLaunchKernelGGL(adam_multi_amp_kernel, dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), tensors, block_map,
                       chunk_elems, hyper, amp);
    return ru3d_check_launch("adam_multi_amp");
}

extern "C" int ru3d_amp_update(ru3d_amp_state* amp, float growth_factor, float backoff_factor, int growth_interval,
                               float min_scale, float max_scale, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(amp && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0 &&
                     min_scale > 0.f && max_scale >= min_scale, "amp_update: bad argument");
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), amp, growth_factor, backoff_factor,
                       growth_interval, min_scale, max_scale);
    return ru3d_check_launch("amp_update");
}

// --------------------------------------------------------------------------- loss scaling (fp16 storage)
// Dynamic loss scaling of the reference's mixed-precision mode (apex O1, trainer.py:492-493, 538-542): gradients are
// computed on `scale * loss`; before the optimizer step every gradient is checked for inf / nan (an overflow skips the
// step and halves the scale) and multiplied by 1 / scale.  Same table / block-map layout as ru3d_adam_multi; only the
// `grad` and `count` fields are read.  scale == 1 checks without writing.
__global__ __launch_bounds__(256) void grad_scale_check_kernel(const ru3d_adam_tensor* __restrict__ tensors,
                                                               const int32_t* __restrict__ block_map, int chunk_elems,
                                                               float scale, float* __restrict__ found_inf) {
    const ru3d_adam_tensor t = tensors[block_map[2 * blockIdx.x]];
    if (!t.grad) return;
    float* g = const_cast<float*>(t.grad);
    const int64_t begin = (int64_t)block_map[2 * blockIdx.x + 1] * chunk_elems;
    int64_t end = begin + chunk_elems;
    if (end > t.count) end = t.count;
    const bool write = scale != 1.f;
    bool bad = false;
    int64_t i = begin + (int64_t)threadIdx.x * 4;
    if ((((uintptr_t)g) & 15) == 0) {
        for (; i + 3 < end; i += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bad = bad || !(fabsf(v[k]) <= 3.402823466e38f);   // false for inf and nan
                v[k] *= scale;
            }
            if (write) *reinterpret_cast<f32x4*>(g + i) = v;
        }
    }
    for (; i < end; i += 1024)
        for (int k = 0; k < 4 && i + k < end; k++) {
            const float v = g[i + k];
            bad = bad || !(fabsf(v) <= 3.402823466e38f);
            if (write) g[i + k] = v * scale;
        }
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.f;   // every writer stores the same value
}

extern "C" int ru3d_grad_scale_check(const ru3d_adam_tensor* tensors, const int32_t* block_map, int nblocks,
                                     int chunk_elems, float scale, float* found_inf, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensors && block_map && found_inf && nblocks > 0 && chunk_elems >= 1024 && (chunk_elems % 1024) == 0,
                 "grad_scale_check: bad argument (chunk_elems must be a positive multiple of 1024)");
    hipLaunchKernelGGL(grad_scale_check_kernel, dim3((unsigned)nblocks), dim3(256), 0, as_stream(stream), tensors,
                       block_map, chunk_elems, scale, found_inf);
    return ru3d_check_launch("grad_scale_check");
}
