// Sliding-window inference accumulators (reference trainer.py:17-98, predict_per_patch).
//
// The reference keeps two [C, X, Y, Z] fp32 volumes on the device (`result`, `result_n`), adds
// softmax(model(patch)) (sigmoid for one class) into the patch's window, divides, then -- unless one_hot --
// applies a SECOND softmax over the averaged probabilities and takes the argmax (trainer.py:77-80, 85-96).
// Here: `acc` is [X, Y, Z, C] (class-last, the layout of the logits and of the one-hot result the
// reference returns through to_numpy), `cnt` is one count per voxel (the reference's C copies are equal).
// Both kernels are pure streaming passes: one read of the logits + one read-modify-write of the window
// for accumulate; one read of acc/cnt and one write of the mask for merge (HBM-bound, coalesced along z/c).
#include "common.h"

#define RU3D_PREDICT_MAX_CLASSES 8

template <typename T, int C>
__global__ __launch_bounds__(256) void predict_accumulate_kernel(const T* __restrict__ z, int ld, int pd, int ph, int pw,
                                                                 float* __restrict__ acc, float* __restrict__ cnt,
                                                                 int Y, int Z, int ox, int oy, int oz) {
    const int64_t total = (int64_t)pd * ph * pw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % pw);
        const int64_t r = i / pw;
        const int j = (int)(r % ph);
        const int a = (int)(r / ph);
        const T* zp = z + i * ld;
        float p[C];
        if (C == 1) {
            p[0] = 1.f / (1.f + __expf(-to_f32<T>(zp[0])));          // torch.sigmoid (trainer.py:75)
        } else {
            float v[C];
#pragma unroll
            for (int c = 0; c < C; c++) v[c] = to_f32<T>(zp[c]);
            float m = v[0];
#pragma unroll
            for (int c = 1; c < C; c++) m = fmaxf(m, v[c]);
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++) {
                v[c] = expf(v[c] - m);
                se += v[c];
            }
#pragma unroll
            for (int c = 0; c < C; c++) p[c] = v[c] / se;             // torch.softmax(dim=1) (trainer.py:77)
        }
        const int64_t o = ((int64_t)(ox + a) * Y + (oy + j)) * Z + (oz + k);
        float* ap = acc + o * C;
#pragma unroll
        for (int c = 0; c < C; c++) ap[c] += p[c];                    // result[window] += output[0]
        cnt[o] += 1.f;                                                // result_n[window] += 1
    }
}

template <int C>
static int accumulate_launch(const ru3d_tensor* t, int dtype, int sample, float* acc, float* cnt, int Y, int Z,
                             int ox, int oy, int oz, hipStream_t st) {
    const int64_t vox = (int64_t)t->d * t->h * t->w;
    int blocks = (int)((vox + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (dtype == RU3D_F32) {
        const float* z = (const float*)t->ptr + (int64_t)sample * vox * t->ld;
        hipLaunchKernelGGL((predict_accumulate_kernel<float, C>), dim3(blocks), dim3(256), 0, st, z, t->ld, t->d, t->h,
                           t->w, acc, cnt, Y, Z, ox, oy, oz);
    } else {
        const bf16* z = (const bf16*)t->ptr + (int64_t)sample * vox * t->ld;
        hipLaunchKernelGGL((predict_accumulate_kernel<bf16, C>), dim3(blocks), dim3(256), 0, st, z, t->ld, t->d, t->h,
                           t->w, acc, cnt, Y, Z, ox, oy, oz);
    }
    return ru3d_check_launch("predict_accumulate");
}

extern "C" int ru3d_predict_accumulate(const ru3d_tensor* logits, int dtype, int sample, float* acc, float* cnt, int X,
                                       int Y, int Z, int ox, int oy, int oz, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(tensor_ok(logits) && acc && cnt, "predict_accumulate: bad argument");
    RU3D_REQUIRE(dtype == RU3D_F32 || dtype == RU3D_BF16, "predict_accumulate: dtype must be f32 or bf16");
    RU3D_REQUIRE(sample >= 0 && sample < logits->n, "predict_accumulate: sample %d outside batch of %d", sample,
                 logits->n);
    RU3D_REQUIRE(logits->c >= 1 && logits->c <= RU3D_PREDICT_MAX_CLASSES, "predict_accumulate: %d classes (max %d)",
                 logits->c, RU3D_PREDICT_MAX_CLASSES);
    RU3D_REQUIRE(ox >= 0 && oy >= 0 && oz >= 0 && ox + logits->d <= X && oy + logits->h <= Y && oz + logits->w <= Z,
                 "predict_accumulate: window [%d+%d, %d+%d, %d+%d] outside the %dx%dx%d volume", ox, logits->d, oy,
                 logits->h, oz, logits->w, X, Y, Z);
    hipStream_t st = as_stream(stream);
    switch (logits->c) {
        case 1: return accumulate_launch<1>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 2: return accumulate_launch<2>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 3: return accumulate_launch<3>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 4: return accumulate_launch<4>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 5: return accumulate_launch<5>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 6: return accumulate_launch<6>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        case 7: return accumulate_launch<7>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
        default: return accumulate_launch<8>(logits, dtype, sample, acc, cnt, Y, Z, ox, oy, oz, st);
    }
}

// result / result_n, then either the probabilities themselves (one_hot) or the class mask, written for the
// centre crop [cx, cx+sx) x [cy, cy+sy) x [cz, cz+sz) of the padded volume (crop_pad, trainer.py:98).
// Voxels no window covered are 0/0 = NaN in the reference: NaN probabilities in one-hot mode; in mask mode
// softmax(NaN) = NaN and torch.argmax returns the first NaN -> class 0 (uint8 cast of NaN -> 0 for C == 1).
template <int C>
__global__ __launch_bounds__(256) void predict_merge_kernel(const float* __restrict__ acc, const float* __restrict__ cnt,
                                                            int Y, int Z, int cx, int cy, int cz, int sx, int sy, int sz,
                                                            int one_hot, float* __restrict__ prob,
                                                            uint8_t* __restrict__ mask) {
    const int64_t total = (int64_t)sx * sy * sz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % sz);
        const int64_t r = i / sz;
        const int j = (int)(r % sy);
        const int a = (int)(r / sy);
        const int64_t o = ((int64_t)(cx + a) * Y + (cy + j)) * Z + (cz + k);
        const float n = cnt[o];
        float p[C];
#pragma unroll
        for (int c = 0; c < C; c++) p[c] = acc[o * C + c] / n;
        if (one_hot) {
#pragma unroll
            for (int c = 0; c < C; c++) prob[i * C + c] = p[c];
            continue;
        }
        if (C == 1) {
            const float v = rintf(p[0]);                              // np.round: half to even
            mask[i] = (v != v) ? (uint8_t)0 : (uint8_t)v;
            continue;
        }
        if (n == 0.f) {
            mask[i] = 0;
            continue;
        }
        float m = p[0];
#pragma unroll
        for (int c = 1; c < C; c++) m = fmaxf(m, p[c]);
        float e[C];
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            e[c] = expf(p[c] - m);
            se += e[c];
        }
        int best = 0;
        float bv = e[0] / se;                                         // second softmax (trainer.py:93)
#pragma unroll
        for (int c = 1; c < C; c++) {
            const float v = e[c] / se;
            if (v > bv) {                                             // first maximum wins, like torch.argmax
                bv = v;
                best = c;
            }
        }
        mask[i] = (uint8_t)best;
    }
}

template <int C>
static int merge_launch(const float* acc, const float* cnt, int Y, int Z, int cx, int cy, int cz, int sx, int sy, int sz,
                        int one_hot, void* out, hipStream_t st) {
    const int64_t vox = (int64_t)sx * sy * sz;
    int blocks = (int)((vox + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((predict_merge_kernel<C>), dim3(blocks), dim3(256), 0, st, acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz,
                       one_hot, one_hot ? (float*)out : nullptr, one_hot ? nullptr : (uint8_t*)out);
    return ru3d_check_launch("predict_merge");
}

extern "C" int ru3d_predict_merge(const float* acc, const float* cnt, int X, int Y, int Z, int num_classes, int cx,
                                  int cy, int cz, int sx, int sy, int sz, int one_hot, void* out, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(acc && cnt && out, "predict_merge: bad argument");
    RU3D_REQUIRE(num_classes >= 1 && num_classes <= RU3D_PREDICT_MAX_CLASSES, "predict_merge: %d classes (max %d)",
                 num_classes, RU3D_PREDICT_MAX_CLASSES);
    RU3D_REQUIRE(sx > 0 && sy > 0 && sz > 0 && cx >= 0 && cy >= 0 && cz >= 0 && cx + sx <= X && cy + sy <= Y &&
                     cz + sz <= Z,
                 "predict_merge: crop [%d+%d, %d+%d, %d+%d] outside the %dx%dx%d volume", cx, sx, cy, sy, cz, sz, X, Y,
                 Z);
    hipStream_t st = as_stream(stream);
    switch (num_classes) {
        case 1: return merge_launch<1>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 2: return merge_launch<2>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 3: return merge_launch<3>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 4: return merge_launch<4>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 5: return merge_launch<5>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 6: return merge_launch<6>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        case 7: return merge_launch<7>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
        default: return merge_launch<8>(acc, cnt, Y, Z, cx, cy, cz, sx, sy, sz, one_hot, out, st);
    }
}
