// On-device patch sampling + augmentation: the reference's training transform chain
//   RandomRescaleCrop -> RandomMirror -> RandomContrast -> RandomBrightness -> RandomGamma -> ToTensor
// (transform.py:573-652, 279-301, 176-259, 156-163; composed in nb_train_iia.py:30-39) as three small HBM-bound
// kernels per patch.  The random draws are made by the host in the reference's order (augment.py); the kernels are
// deterministic functions of (volume, parameters), so a patch can be replayed against the CPU pipeline.
//
//   presence   which label values occur in the crop box (decides the label resampling rule, transform.py:47-48, and
//              serves enforce_label_indices, :620-637)
//   resample   crop + constant pad + scipy.ndimage.zoom(order=1) restated: output sample o of an axis reads crop
//              coordinate o * (before - 1) / (patch - 1), trilinear interpolation in float64, float32 result;
//              labels: < 3 classes present: interpolate the label and truncate (transform.py:60-65), else one-hot per
//              class, interpolate, first maximum (:66-74); the mirror is folded into the output index; per-block
//              {sum, min, max} of the image for the intensity ops
//   intensity  contrast about the mean, brightness about the minimum, gamma on the [min, max] range, float32
//              arithmetic in the reference's order of operations; the minima / maxima after each monotone step are
//              derived from the resampled image's own, so one reduction serves all three ops
#include "common.h"

namespace {

struct Axis {
    double step;      // crop coordinate per output sample
    int before, lo, ext, patch, flip;
};

__device__ __forceinline__ void axis_sample(const Axis& a, int o, int& i0, int& i1, double& w) {
    const int s = a.flip ? a.patch - 1 - o : o;
    const double c = (double)s * a.step;
    int f = (int)floor(c);
    f = f < 0 ? 0 : (f > a.before - 1 ? a.before - 1 : f);
    w = c - (double)f;
    i0 = f;
    i1 = f + 1 > a.before - 1 ? a.before - 1 : f + 1;
}

template <typename L>
__device__ __forceinline__ int load_label(const L* lab, int64_t idx) { return (int)lab[idx]; }

template <typename L>
__global__ __launch_bounds__(256) void presence_kernel(const L* __restrict__ label, int X, int Y, int Z, int lx, int ly,
                                                       int lz, int bx, int by, int bz, int cval,
                                                       unsigned* __restrict__ mask) {
    const int64_t total = (int64_t)bx * by * bz;
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int z = (int)(i % bz);
        const int64_t t = i / bz;
        const int y = (int)(t % by), x = (int)(t / by);
        const int gx = lx + x, gy = ly + y, gz = lz + z;
        int v = cval;
        if (gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z) v = load_label(label, ((int64_t)gx * Y + gy) * Z + gz);
        m |= 1u << (v < 0 ? 0 : (v > 31 ? 31 : v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor((int)m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(mask, m);
}

template <typename L>
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ image, const L* __restrict__ label,
                                                       int X, int Y, int Z, int C, Axis ax, Axis ay, Axis az,
                                                       float image_cval, int label_cval,
                                                       const unsigned* __restrict__ mask, float* __restrict__ out_image,
                                                       int64_t* __restrict__ out_label, double* __restrict__ part) {
    __shared__ double red[3][4];
    const int64_t total = (int64_t)ax.patch * ay.patch * az.patch;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    if (i < total) {
        const int oz = (int)(i % az.patch);
        const int64_t t = i / az.patch;
        const int oy = (int)(t % ay.patch), ox = (int)(t / ay.patch);
        int x0, x1, y0, y1, z0, z1;
        double wx, wy, wz;
        axis_sample(ax, ox, x0, x1, wx);
        axis_sample(ay, oy, y0, y1, wy);
        axis_sample(az, oz, z0, z1, wz);
        const int xs[2] = {ax.lo + x0, ax.lo + x1}, ys[2] = {ay.lo + y0, ay.lo + y1}, zs[2] = {az.lo + z0, az.lo + z1};
        const double wxs[2] = {1.0 - wx, wx}, wys[2] = {1.0 - wy, wy}, wzs[2] = {1.0 - wz, wz};
        int64_t src[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int gx = xs[k >> 2], gy = ys[(k >> 1) & 1], gz = zs[k & 1];
            const bool in = gx >= 0 && gx < X && gy >= 0 && gy < Y && gz >= 0 && gz < Z;
            src[k] = in ? ((int64_t)gx * Y + gy) * Z + gz : -1;
        }
        // separable order of scipy's zoom: axis 0 first, then 1, then 2 (each a float64 lerp)
        for (int c = 0; image && c < C; c++) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[k] >= 0 ? (double)image[src[k] * C + c] : (double)image_cval;
            double a4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) a4[k] = v[k] * wxs[0] + v[k + 4] * wxs[1];
            const double b0 = a4[0] * wys[0] + a4[2] * wys[1], b1 = a4[1] * wys[0] + a4[3] * wys[1];
            const float r = (float)(b0 * wzs[0] + b1 * wzs[1]);
            out_image[(int64_t)c * total + i] = r;
            s += (double)r;
            mn = fminf(mn, r);
            mx = fmaxf(mx, r);
        }
        if (label && out_label) {
            int lv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) lv[k] = src[k] >= 0 ? load_label(label, src[k]) : label_cval;
            const unsigned m = mask ? *mask : 0xffffffffu;
            const int num_classes = 32 - __clz((int)m);          // highest label present + 1
            int64_t res;
            if (num_classes < 3) {
                double a4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) a4[k] = (double)lv[k] * wxs[0] + (double)lv[k + 4] * wxs[1];
                const double b0 = a4[0] * wys[0] + a4[2] * wys[1], b1 = a4[1] * wys[0] + a4[3] * wys[1];
                const float r = (float)(b0 * wzs[0] + b1 * wzs[1]);
                res = (int64_t)r;                                  // astype(integer dtype): truncation
            } else {
                int best = 0x7fffffff;
                float best_w = -1.f;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // the class plane's interpolated value, in the separable order used for images
                    double v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = lv[j] == lv[k] ? 1.0 : 0.0;
                    double a4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) a4[j] = v[j] * wxs[0] + v[j + 4] * wxs[1];
                    const double b0 = a4[0] * wys[0] + a4[2] * wys[1], b1 = a4[1] * wys[0] + a4[3] * wys[1];
                    const float tw = (float)(b0 * wzs[0] + b1 * wzs[1]);
                    if (tw > best_w || (tw == best_w && lv[k] < best)) {
                        best_w = tw;
                        best = lv[k];
                    }
                }
                // classes that are not among the 8 neighbours have plane value 0 and the maximum is positive (the
                // weights sum to one), so the first maximum over all classes is the best neighbour class
                res = best;
            }
            out_label[i] = res;
        }
    }
    // block partials: fixed order
    double ws = s;
    float wmn = mn, wmx = mx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ws += __shfl_xor(ws, o, 64);
        wmn = fminf(wmn, __shfl_xor(wmn, o, 64));
        wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = ws;
        red[1][wave] = (double)wmn;
        red[2][wave] = (double)wmx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, a = red[1][0], b = red[2][0];
        for (int k = 0; k < 4; k++) {
            t += red[0][k];
            a = fmin(a, red[1][k]);
            b = fmax(b, red[2][k]);
        }
        part[3 * blockIdx.x] = t;
        part[3 * blockIdx.x + 1] = a;
        part[3 * blockIdx.x + 2] = b;
    }
}

__global__ __launch_bounds__(256) void intensity_kernel(float* __restrict__ img, int64_t count,
                                                        const double* __restrict__ part, int nparts, int do_contrast,
                                                        float fc, int do_brightness, float fb, int do_gamma, float fg,
                                                        float eps) {
    __shared__ double red[3][256];
    double s = 0.0, a = INFINITY, b = -INFINITY;
    for (int k = threadIdx.x; k < nparts; k += 256) {
        s += part[3 * k];
        a = fmin(a, part[3 * k + 1]);
        b = fmax(b, part[3 * k + 2]);
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = a;
    red[2][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] = fmin(red[1][threadIdx.x], red[1][threadIdx.x + o]);
            red[2][threadIdx.x] = fmax(red[2][threadIdx.x], red[2][threadIdx.x + o]);
        }
        __syncthreads();
    }
    const float mean = (float)(red[0][0] / (double)count);
    float mn = (float)red[1][0], mx = (float)red[2][0];
    // every step below is monotone in the voxel value (also after float32 rounding), so the extrema follow the extrema
    auto contrast = [&](float v) { return (v - mean) * fc + mean; };
    float mn1 = mn, mx1 = mx;
    if (do_contrast) {
        const float lo = contrast(mn), hi = contrast(mx);
        mn1 = fminf(lo, hi);
        mx1 = fmaxf(lo, hi);
    }
    auto bright = [&](float v) { return (v - mn1) * fb + mn1; };
    float mn2 = mn1, mx2 = mx1;
    if (do_brightness) {
        const float lo = bright(mn1), hi = bright(mx1);
        mn2 = fminf(lo, hi);
        mx2 = fmaxf(lo, hi);
    }
    const float arange = mx2 - mn2 + eps;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float v = img[i];
        if (do_contrast) v = contrast(v);
        if (do_brightness) v = bright(v);
        if (do_gamma) v = powf((v - mn2) / arange, fg) * arange + mn2;
        img[i] = v;
    }
}

Axis make_axis(int lo, int before, int ext, int patch, int flip) {
    Axis a;
    a.lo = lo; a.before = before; a.ext = ext; a.patch = patch; a.flip = flip;
    a.step = patch > 1 ? (double)(before - 1) / (double)(patch - 1) : 0.0;
    return a;
}

}  // namespace

extern "C" size_t ru3d_augment_workspace_bytes(int px, int py, int pz) {
    if (px <= 0 || py <= 0 || pz <= 0) return 0;
    const int64_t total = (int64_t)px * py * pz;
    return (size_t)((total + 255) / 256) * 3 * sizeof(double) + 256;
}

extern "C" int ru3d_augment_label_presence(const void* label, int label_dtype, int X, int Y, int Z, const int32_t* lo,
                                           const int32_t* before, int label_cval, uint32_t* mask, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(label && lo && before && mask && X > 0 && Y > 0 && Z > 0, "augment_label_presence: bad argument");
    RU3D_REQUIRE(before[0] > 0 && before[1] > 0 && before[2] > 0, "augment_label_presence: empty crop box");
    hipError_t e = hipMemsetAsync(mask, 0, sizeof(uint32_t), as_stream(stream));
    if (e != hipSuccess) return ru3d_fail((int)e, "augment_label_presence: memset: %s", hipGetErrorString(e));
    const int64_t total = (int64_t)before[0] * before[1] * before[2];
    int64_t blocks = (total + 1023) / 1024;
    if (blocks > 1024) blocks = 1024;
    if (label_dtype == RU3D_LABEL_U8)
        hipLaunchKernelGGL(presence_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                           (const uint8_t*)label, X, Y, Z, lo[0], lo[1], lo[2], before[0], before[1], before[2],
                           label_cval, mask);
    else if (label_dtype == RU3D_LABEL_I64)
        hipLaunchKernelGGL(presence_kernel<int64_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                           (const int64_t*)label, X, Y, Z, lo[0], lo[1], lo[2], before[0], before[1], before[2],
                           label_cval, mask);
    else
        return ru3d_fail(-1, "augment_label_presence: bad label dtype %d", label_dtype);
    return ru3d_check_launch("augment_label_presence");
}

extern "C" int ru3d_augment_patch(const float* image, const void* label, int label_dtype, int X, int Y, int Z, int C,
                                  const ru3d_patch_params* p, const uint32_t* presence_mask, float* out_image,
                                  int64_t* out_label, void* ws, size_t ws_bytes, void* stream) {
    Ru3dDeviceGuard dev_guard(stream);
    RU3D_REQUIRE(p && ws && X > 0 && Y > 0 && Z > 0 && C > 0, "augment_patch: bad argument");
    RU3D_REQUIRE((image == nullptr) == (out_image == nullptr), "augment_patch: image and out_image go together");
    RU3D_REQUIRE((label == nullptr) == (out_label == nullptr), "augment_patch: label and out_label go together");
    RU3D_REQUIRE(image || label, "augment_patch: nothing to resample");
    for (int d = 0; d < 3; d++)
        RU3D_REQUIRE(p->before[d] > 0 && p->patch[d] > 0, "augment_patch: empty crop box / patch on axis %d", d);
    RU3D_REQUIRE(ws_bytes >= ru3d_augment_workspace_bytes(p->patch[0], p->patch[1], p->patch[2]),
                 "augment_patch: workspace too small");
    RU3D_REQUIRE((int64_t)X * Y * Z * C < (1ll << 40), "augment_patch: volume too large");
    const int64_t total = (int64_t)p->patch[0] * p->patch[1] * p->patch[2];
    RU3D_REQUIRE(total < (1ll << 31), "augment_patch: patch too large");
    const Axis ax = make_axis(p->lo[0], p->before[0], X, p->patch[0], p->flip[0]);
    const Axis ay = make_axis(p->lo[1], p->before[1], Y, p->patch[1], p->flip[1]);
    const Axis az = make_axis(p->lo[2], p->before[2], Z, p->patch[2], p->flip[2]);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    double* part = (double*)ws;
    if (!label || label_dtype == RU3D_LABEL_U8)
        hipLaunchKernelGGL(resample_kernel<uint8_t>, dim3(blocks), dim3(256), 0, as_stream(stream), image,
                           (const uint8_t*)label, X, Y, Z, C, ax, ay, az, p->image_cval, p->label_cval, presence_mask,
                           out_image, out_label, part);
    else if (label_dtype == RU3D_LABEL_I64)
        hipLaunchKernelGGL(resample_kernel<int64_t>, dim3(blocks), dim3(256), 0, as_stream(stream), image,
                           (const int64_t*)label, X, Y, Z, C, ax, ay, az, p->image_cval, p->label_cval, presence_mask,
                           out_image, out_label, part);
    else
        return ru3d_fail(-1, "augment_patch: bad label dtype %d", label_dtype);
    int rc = ru3d_check_launch("augment_resample");
    if (rc) return rc;
    if (!image || !(p->do_contrast || p->do_brightness || p->do_gamma)) return 0;
    const int64_t count = total * C;
    int64_t ib = (count + 1023) / 1024;
    if (ib > 512) ib = 512;
    hipLaunchKernelGGL(intensity_kernel, dim3((unsigned)ib), dim3(256), 0, as_stream(stream), out_image, count,
                       (const double*)part, (int)blocks, p->do_contrast, p->contrast, p->do_brightness, p->brightness,
                       p->do_gamma, p->gamma, p->gamma_eps);
    return ru3d_check_launch("augment_intensity");
}
