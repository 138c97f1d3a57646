// The tail of a decoder ResBlock in one pass (reference network.py:403 skip_conv k1 s1, :406-409, :414-416):
//     z = LeakyReLU( InstanceNorm(y2) + skip_conv(x) )
// The unfused path ran the 1x1x1 conv as its own launch (read x, write skip), then the apply pass (read y2, read skip,
// write z): 2 C + C | C + C + C channel-passes over the level's voxels.  Here the skip tensor never exists: a wave takes
// 16 voxels, reads their 2 C input channels straight into MFMA B fragments (16 bytes per lane, whole 128-byte voxel rows
// per request), multiplies with the register-resident weight (v_mfma_f32_16x16x32, output channels interleaved so that a
// lane ends up with 8 consecutive ones), and folds y2's normalisation, the sum and the activation into the epilogue:
// read x, read y2, write z.  Streaming kernel: no LDS, occupancy hides the latency.
// The skip value is rounded to the storage type before the sum, exactly where the unfused path stored it.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {
namespace {

#ifdef RU3D_STORAGE_F16
#define RU3D_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define RU3D_MFMA_16X16X32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef float f32x8 __attribute__((ext_vector_type(8)));

struct SkipArgs {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* y2;
    const float* mean;
    const float* scale;
    bf16* out;
    int N;
    int64_t V;              // voxels per sample (multiple of 16)
    int ldx, ldy, ldo;
    int cout;
    float slope;
    int x_cseg;             // split x (planar concat): 32-channel block ks lives in plane ks * 32 / x_cseg
    int64_t x_segstride;
};

// CIN input channels, NPB blocks of 32 output channels
template <int CIN, int NPB>
__global__ __launch_bounds__(256) void skip1x1_in_lrelu_fwd_kernel(SkipArgs a) {
    constexpr int KS = CIN / 32;
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, g4 = lane >> 4;
    const int NTT = a.cout / 32;

    // weights: [k-step][block][tile]; tile t row r <-> output channel 32 pb + 8 (r / 4) + 4 t + r % 4
    bf16x8 wreg[KS * NPB * 2];
    static_for<0, KS * NPB * 2>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ks = i / (NPB * 2), pb = (i / 2) % NPB, t = i & 1;
        const int co = pb * 32 + 8 * (p >> 2) + 4 * t + (p & 3);
        const int ci8 = ks * 4 + g4;
        wreg[i] = a.w[((ci8 >> 1) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (ci8 & 1)];
    });
    float bias8[NPB][8];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++)
#pragma unroll
        for (int i = 0; i < 8; i++) bias8[pb][i] = a.bias ? a.bias[pb * 32 + 8 * g4 + i] : 0.f;

    const int64_t groups_per_sample = a.V / 16;
    const int64_t total = groups_per_sample * a.N;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int cur_n = -1;
    float mu[NPB][8], sc[NPB][8];
    for (int64_t gi = wave_id; gi < total; gi += nwaves) {
        const int n = (int)(gi / groups_per_sample);
        if (n != cur_n) {
            cur_n = n;
#pragma unroll
            for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    mu[pb][i] = a.mean[n * a.cout + pb * 32 + 8 * g4 + i];
                    sc[pb][i] = a.scale[n * a.cout + pb * 32 + 8 * g4 + i];
                }
        }
        const int64_t vox = gi * 16 + p;               // global voxel row (samples are contiguous)
        bf16x8 xb[KS];
        static_for<0, KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            const bf16* xp = a.x_cseg ? a.x + (int64_t)((ks * 32) / a.x_cseg) * a.x_segstride + (ks * 32) % a.x_cseg
                                      : a.x + ks * 32;
            xb[ks] = *reinterpret_cast<const bf16x8*>(xp + vox * a.ldx + g4 * 8);
        });
        bf16x8 yv[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; pb++) yv[pb] = *reinterpret_cast<const bf16x8*>(a.y2 + vox * a.ldy + pb * 32 + 8 * g4);
        f32x4 acc[NPB][2];
#pragma unroll
        for (int pb = 0; pb < NPB; pb++) {
            acc[pb][0] = f32x4{bias8[pb][0], bias8[pb][1], bias8[pb][2], bias8[pb][3]};
            acc[pb][1] = f32x4{bias8[pb][4], bias8[pb][5], bias8[pb][6], bias8[pb][7]};
        }
        static_for<0, KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            static_for<0, NPB * 2>([&](auto tc) {
                constexpr int j = decltype(tc)::value;
                acc[j >> 1][j & 1] = RU3D_MFMA_16X16X32(wreg[ks * NPB * 2 + j], xb[ks], acc[j >> 1][j & 1], 0, 0, 0);
            });
        });
#pragma unroll
        for (int pb = 0; pb < NPB; pb++) {
            f32x4 a0 = acc[pb][0], a1 = acc[pb][1];
            // (wait states between the last 16x16x32 MFMA and a VALU read of its result: see conv_s2.hip)
            asm("s_nop 7\n\ts_nop 4" : "+v"(a0), "+v"(a1));
            const f32x8 s = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const bf16x8 sk = __builtin_convertvector(s, bf16x8);       // the skip tensor's stored value
            f32x8 o;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float t = ((float)yv[pb][i] - mu[pb][i]) * sc[pb][i] + (float)sk[i];
                o[i] = t > 0.f ? t : t * a.slope;
            }
            *reinterpret_cast<bf16x8*>(a.out + vox * a.ldo + pb * 32 + 8 * g4) = __builtin_convertvector(o, bf16x8);
        }
    }
}

}  // namespace

bool skip1x1_fused_eligible(const ru3d_tensor* x, const ru3d_tensor* y2, const ru3d_tensor* out, int dtype) {
    static const int mode = getenv("RU3D_FUSED_SKIP") ? atoi(getenv("RU3D_FUSED_SKIP")) : 1;
    if (!mode || dtype != RU3D_BF16) return false;
    if (!tensor_ok_split(x) || !tensor_ok(y2) || !tensor_ok(out)) return false;
    if (x->cseg && (x->cseg % 32)) return false;
    if (x->n != y2->n || x->d != y2->d || x->h != y2->h || x->w != y2->w) return false;
    if (out->n != y2->n || out->d != y2->d || out->h != y2->h || out->w != y2->w || out->c != y2->c) return false;
    if (!((x->c == 64 && y2->c == 32) || (x->c == 128 && y2->c == 64))) return false;
    if ((x->ld % 8) || (y2->ld % 8) || (out->ld % 8)) return false;
    if ((((uintptr_t)x->ptr) | ((uintptr_t)y2->ptr) | ((uintptr_t)out->ptr)) % 16) return false;
    const int64_t V = (int64_t)x->d * x->h * x->w;
    return (V % 16) == 0 && V * x->n >= 65536;          // small levels: the launches it saves are not the cost there
}

int skip1x1_fused_launch(const ru3d_tensor* x, const void* w, const float* bias, const ru3d_tensor* y2, const float* mean,
                         const float* scale, const ru3d_tensor* out, float slope, hipStream_t st) {
    SkipArgs a;
    a.x = (const bf16*)x->ptr; a.w = (const bf16x8*)w; a.bias = bias; a.y2 = (const bf16*)y2->ptr;
    a.mean = mean; a.scale = scale; a.out = (bf16*)out->ptr;
    a.N = x->n;
    a.V = (int64_t)x->d * x->h * x->w;
    a.ldx = x->ld; a.ldy = y2->ld; a.ldo = out->ld;
    a.cout = y2->c;
    a.slope = slope;
    a.x_cseg = x->cseg; a.x_segstride = x->seg_stride;
    const int64_t groups = a.V / 16 * a.N;
    int64_t blocks = (groups + 3) / 4;
    const int64_t cap = 256 * 8;                        // eight workgroups per CU, each wave walks its groups
    if (blocks > cap) blocks = cap;
    if (x->c == 64) hipLaunchKernelGGL((skip1x1_in_lrelu_fwd_kernel<64, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((skip1x1_in_lrelu_fwd_kernel<128, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    return ru3d_check_launch("skip1x1_in_lrelu_fwd");
}

}  // namespace RU3D_NS
