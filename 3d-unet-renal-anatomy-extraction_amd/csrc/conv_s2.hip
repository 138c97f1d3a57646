// The stride-2 forms of the TOP level boundary (128^3 <-> 64^3 of config 2: 32 <-> 64 channels) as LDS-tiled, D-sliding,
// persistent kernels on v_mfma_f32_16x16x32 (reference network.py:311-314 ConvTranspose3d k3 s2 p1 + far pad, :394 the
// pooling ResBlock's stride-2 conv1, :403 its 1x1x1 stride-2 skip_conv, and the input gradients of the three).
//
// SCOPE (t_plan / g_plan below): T form Cin = 64 with Cout 32 or 64, G form Cin = 32 with Cout 64 or 128 - config 2's
// 32 <-> 64 channel boundary in both directions and nothing else.  The 64^3 <-> 32^3 boundary (64 <-> 128
// channels) and everything deeper do NOT run here and are not meant to: the design keeps a consumer wave's weights in
// registers for the whole launch (T form: up to 8 taps x Cin x 32 couts), which at Cin = 128 / Cout = 64 is 8 x 16 KB
// per wave - four times the register file - and the LDS plane ring (5 planes x (TH+2) rows x W x Cin) no longer fits
// two workgroups per CU at 64 channels in, 3 input rows out.  Those shapes stay on conv_gather_mfma_kernel /
// convt_tile_mfma_kernel / conv_direct_mfma_kernel (conv_mfma.hip), 0.06-0.08 ms each, 0.23 ms of the step in
// profiles/r04_kernel_trace_by_grid.txt; they are weight-bandwidth bound like the other deep-level kernels
// (profiles/r04_pmc_deep_levels.txt), not HBM bound.
//
//   T form  in [n^3, Cin] -> out [(2n)^3, Cout]: ConvTranspose3d forward, input gradient of the stride-2 conv
//           (+ optionally, in the same launch: the input gradient of the 1x1x1 stride-2 skip conv, whose only non-zero
//           outputs sit on the even-even-even voxels, a residual operand, the far-plane zeros, the InstanceNorm sums
//           of the output);
//   G form  in [(2n)^3, Cin] -> out [n^3, Cout]: stride-2 conv forward, ConvTranspose3d input gradient
//           (+ optionally the 1x1x1 stride-2 skip conv of the same input as a second output - its operand is the centre
//           tap's activation fragment, which is in LDS anyway - and the InstanceNorm sums of the first output).
//
// The kernels they replace (conv_gather_mfma_kernel, convt_tile_mfma_kernel, conv_direct_mfma_kernel<.., true>) gathered
// 16-byte fragments straight from global memory or re-read the weights from L2 for every tile and wrote 8-byte pieces:
// 17-47 % of their HBM time.  Here
//   * a workgroup owns an (H, W) column of the output and slides along D: every input plane enters the CU once
//     (T: 1 new plane per step, ring of 3; G: 2 new planes per step, ring of 5), moved by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`): no staging registers, rows outside the volume come back as zeros;
//   * a PRODUCER wave issues the DMAs and is the only wave that waits for them (`vmcnt` counts a wave's loads AND stores
//     in order: a consumer that waited for "its" DMA would drain its output stores with it); four CONSUMER waves read
//     fragments from LDS, run the MFMAs and store; one LDS-only barrier per step;
//   * the consumers keep their weights in registers for the whole launch: T form - a wave owns whole output parity
//     classes ({7} | {3,5} | {6,1} | {2,4,0}: 8 | 8 | 6 | 5 of the 27 taps), G form - a wave owns 16 output channels;
//   * LDS rows are padded by 32 bytes (pitch 96 / 160 B): the 16-voxel x 32-channel fragment read (ds_read_b128) is
//     conflict-free for every row offset a tap needs; G form rows are de-interleaved by W parity (the DMA's per-lane
//     source address makes any order free), so that the 16 positions of a tap's fragment are 16 consecutive rows;
//   * T form: the output channels of the two MFMA tiles of a 32-channel block are interleaved (tile t row r <-> channel
//     8 (r / 4) + 4 t + r % 4) so that a lane ends up with 8 consecutive channels = one 16-byte store; a wave-store
//     covers whole 64-byte voxel rows.
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
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int NCONS = 4;                  // consumer waves (one per SIMD)
constexpr int TH = 4;                     // tile rows (T: half-resolution rows, G: output rows)
constexpr int OOB = (int)0x80000000;      // buffer offset beyond any descriptor range: loads return 0, stores are dropped

struct S2Args {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    bf16* y;
    float* stat_slab;
    // the fused 1x1x1 stride-2 partner: T form - second input x2 (Cin channels, the input's grid) and its weight; G form -
    // second weight, bias and output y2 (Cout channels, the output's grid)
    const bf16* x2;
    const bf16x8* w2;
    const float* bias2;
    bf16* y2;
    int N;
    int Di, Hi, Wi, Do, Ho, Wo;           // input / output extents of this data movement
    int ldx, ldy, ldr, ldx2, ldy2;
    int flip, zero_far;
    int cout_total;
    int tiles_h, tiles_w, dsplit, DL, units;
    int dbg;                              // diagnostics (RU3D_S2_DBG): 1 = drop the output stores (and take the per-lane path), 2 = no input loads
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int num_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, num_bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_load16(rsrc_t r, int off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store16(rsrc_t r, int off, const bf16x8& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, v), r, off, 0, 0);
}
__device__ __forceinline__ void buf_store8(rsrc_t r, int off, const bf16x4& v) {
    typedef int i32x2v __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2v, v), r, off, 0, 0);
}

// consumers: my LDS reads are done; producer: my DMAs have landed (it issues nothing else) - then everybody meets
__device__ __forceinline__ void consumer_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void producer_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }

// element (co, ci) of tap t in the library's packed (32x32x16 fragment) order, as an index of 16-byte units holding
// 8 consecutive ci:  ((t * KS16 + ci / 16) * NTT + co / 32) * 64 + (co % 32) + 32 * ((ci / 8) & 1)
__device__ __forceinline__ int packed_unit(int tap, int KS16, int NTT, int co, int ci8) {
    return ((tap * KS16 + (ci8 >> 1)) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (ci8 & 1);
}

// --------------------------------------------------------------------------------------------------- T form
template <int CIN, int TW>
struct TGeom {
    static constexpr int KS = CIN / 32;                       // MFMA k-steps per tap
    static constexpr int PITCH_B = CIN * 2 + 32;              // bytes per LDS row
    static constexpr int LW = TW + 1, ROWS = (TH + 1) * LW;   // halo rows per plane
    static constexpr int ND = (ROWS * PITCH_B + 1023) / 1024; // DMA wave-instructions per plane
    static constexpr int PLANE_B = ND * 1024;
    static constexpr int RING = 3;
    static constexpr int G = TH * TW / 16;                    // 16-position groups per step
    // fused second input: TH x TW rows, no halo, double-buffered
    static constexpr int X2ROWS = TH * TW;
    static constexpr int ND2 = (X2ROWS * PITCH_B + 1023) / 1024;
    static constexpr int X2_B = ND2 * 1024;
};

// classes of consumer wave WV, heaviest first
template <int WV> struct WaveClasses;
template <> struct WaveClasses<0> { static constexpr int n = 1; static constexpr int cls[3] = {7, 0, 0}; };
// (cost of a class per 16 positions ~ taps x 64 MFMA cycles + ~200 cycles of epilogue: 712 | 912 | 784 | 920)
template <> struct WaveClasses<1> { static constexpr int n = 2; static constexpr int cls[3] = {3, 5, 0}; };
template <> struct WaveClasses<2> { static constexpr int n = 2; static constexpr int cls[3] = {6, 1, 0}; };
template <> struct WaveClasses<3> { static constexpr int n = 3; static constexpr int cls[3] = {2, 4, 0}; };

constexpr int class_taps(int cl) { return (1 + (cl >> 2)) * (1 + ((cl >> 1) & 1)) * (1 + (cl & 1)); }
template <int WV> constexpr int wave_taps() {
    int t = 0;
    for (int i = 0; i < WaveClasses<WV>::n; i++) t += class_taps(WaveClasses<WV>::cls[i]);
    return t;
}
template <int WV> constexpr int class_tap_base(int idx) {
    int t = 0;
    for (int i = 0; i < idx; i++) t += class_taps(WaveClasses<WV>::cls[i]);
    return t;
}
// tap ti of class cl: kernel index per axis and the input offset (0 / 1) it reads
struct TapInfo { int tap, dd, dh, dw; };
constexpr TapInfo class_tap(int cl, int ti) {
    const int bd = cl >> 2, bh = (cl >> 1) & 1, bw = cl & 1;
    const int nkh = 1 + bh, nkw = 1 + bw;
    const int iw = ti % nkw, ih = (ti / nkw) % nkh, idd = ti / (nkw * nkh);
    // axis with class bit 1: index 0 -> k = 0 (input offset 1), index 1 -> k = 2 (offset 0); bit 0: k = 1 (offset 0)
    const int kd = bd ? 2 * idd : 1, kh = bh ? 2 * ih : 1, kw = bw ? 2 * iw : 1;
    return TapInfo{(kd * 3 + kh) * 3 + kw, bd ? 1 - idd : 0, bh ? 1 - ih : 0, bw ? 1 - iw : 0};
}

template <int CIN, int TW, bool HAS_RES, bool HAS_STATS, bool HAS_X2, int WV>
__device__ __forceinline__ void t_consumer(const S2Args& a, const char* lds, const char* lds_x2) {
    using GM = TGeom<CIN, TW>;
    using WC = WaveClasses<WV>;
    constexpr int KS = GM::KS;
    constexpr int NTAPS = wave_taps<WV>();
    constexpr bool X2W = HAS_X2 && WV == 3;            // the wave that owns class 0 adds the 1x1 partner
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, g4 = lane >> 4;
    const int cb = blockIdx.y;                         // 32-channel output block
    const int NTT = a.cout_total / 32;

    // ---- weights: [tap of the wave][k-step][tile], A fragment lane -> row r = lane & 15, k-block lane >> 4
    bf16x8 wreg[NTAPS * KS * 2];
    {
        const int co0 = cb * 32 + 8 * (p >> 2) + (p & 3);
        static_for<0, WC::n>([&](auto ic) {
            constexpr int ci = decltype(ic)::value;
            constexpr int cl = WC::cls[ci];
            static_for<0, class_taps(cl)>([&](auto tc) {
                constexpr int ti = decltype(tc)::value;
                constexpr TapInfo ti_ = class_tap(cl, ti);
                const int wtap = a.flip ? 26 - ti_.tap : ti_.tap;
                static_for<0, KS * 2>([&](auto kc) {
                    constexpr int ks = decltype(kc)::value >> 1, t = decltype(kc)::value & 1;
                    wreg[((class_tap_base<WV>(ci) + ti) * KS + ks) * 2 + t] =
                        a.w[packed_unit(wtap, CIN / 16, NTT, co0 + 4 * t, ks * 4 + g4)];
                });
            });
        });
    }
    bf16x8 w2reg[X2W ? KS * 2 : 1];
    if constexpr (X2W) {
        const int co0 = cb * 32 + 8 * (p >> 2) + (p & 3);
        static_for<0, KS * 2>([&](auto kc) {
            constexpr int ks = decltype(kc)::value >> 1, t = decltype(kc)::value & 1;
            w2reg[ks * 2 + t] = a.w2[packed_unit(0, CIN / 16, NTT, co0 + 4 * t, ks * 4 + g4)];
        });
    }
    float bias8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bias8[i] = a.bias ? a.bias[cb * 32 + 8 * g4 + i] : 0.f;

    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
    int cur_n = -1;
    if (HAS_STATS) ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + WV, a.N, 32 * 2);
    auto stat_flush = [&]() {
        if (!HAS_STATS || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + WV) * a.N + cur_n) * 32) * 2;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float s1 = st1[i], s2 = st2[i];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {       // lanes with equal g4 hold the same 8 channels
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (p == 0) {      // += : the XCD-wise deal of units can bring a workgroup back to a sample (the wave cleared its rows when the kernel started)
                dst[(8 * g4 + i) * 2] += s1;
                dst[(8 * g4 + i) * 2 + 1] += s2;
            }
            st1[i] = st2[i] = 0.f;
        }
    };

    const int y_lane = (2 * p * a.ldy + cb * 32 + 8 * g4) * 2, r_lane = (2 * p * a.ldr + cb * 32 + 8 * g4) * 2;
    const char* const lane_lds = lds + p * GM::PITCH_B + g4 * 16;
    const char* const lane_x2 = lds_x2 + p * GM::PITCH_B + g4 * 16;
    const int ysample_b = a.Do * a.Ho * a.Wo * a.ldy * 2, rsample_b = a.Do * a.Ho * a.Wo * a.ldr * 2;

    const int Gx = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (Gx % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += Gx) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, b0 = th_i * TH, c0 = tw_i * TW;
        int dl = a.Di - d0;
        if (dl > a.DL) dl = a.DL;
        if (HAS_STATS && n != cur_n) {
            stat_flush();
            cur_n = n;
        }
        const rsrc_t ry = make_rsrc(a.y + (int64_t)n * (ysample_b / 2), ysample_b);
        rsrc_t rr = ry;
        if constexpr (HAS_RES) rr = make_rsrc(a.res + (int64_t)n * (rsample_b / 2), rsample_b);

        // An output voxel (od, oh, ow) = (2 aa + bd, 2 (b0 + hb) + bh, 2 (c0 + wc + p) + bw): everything but the 2 p is
        // wave-uniform, so a store / residual load is `voffset` = the lane's constant (y_lane / r_lane) + `soffset` = the
        // scalar rest: no vector address arithmetic per group.  Tiles that stick out of the volume (or touch a far face that
        // must be zeroed) take the per-lane test; masked lanes get an offset beyond the descriptor's range.
        const bool edge = 2 * (b0 + TH) > a.Ho || 2 * (c0 + TW) > a.Wo || (a.dbg & 1);
        const bool far_hw = a.zero_far && (2 * (b0 + TH) >= a.Ho || 2 * (c0 + TW) >= a.Wo);
        auto lane_ok = [&](int od, int hb, int wc, int bh, int bw, bool& far) {
            const int oh = 2 * (b0 + hb) + bh, ow = 2 * (c0 + wc + p) + bw;
            far = od == a.Do - 1 || oh == a.Ho - 1 || ow == a.Wo - 1;
            return od < a.Do && oh < a.Ho && ow < a.Wo && !(a.dbg & 1);
        };
        auto uni_vox = [&](int od, int hb, int wc, int bh, int bw) {
            return (od * a.Ho + 2 * (b0 + hb) + bh) * a.Wo + 2 * (c0 + wc) + bw;
        };
        // residual rows are fetched eight groups ahead of their use: a wave's loads and stores retire in order (vmcnt),
        // so the wait for a residual row also waits for every older store - the distance is the number of stores that
        // may still be on their way.  rq[k]: group U j + k of the class being computed (U = 8 groups unrolled).
        constexpr int U = GM::G < 8 ? GM::G : 8;
        bf16x8 rq[U];
        auto res_fetch = [&](bf16x8& dst, int od, int g, int bh, int bw) {
            if constexpr (HAS_RES) {
                const int hb = g / (TW / 16), wc = (g % (TW / 16)) * 16;
                bool far;
                const bool ok = !(edge || od >= a.Do) || lane_ok(od, hb, wc, bh, bw, far);
                dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? r_lane : OOB,
                                                                                        uni_vox(od, hb, wc, bh, bw) * a.ldr * 2, 0));
            }
        };
        // item `ahead` groups behind group g of class index ci of step s: (od, group, bh, bw), or od < 0 when the unit ends
        auto item = [&](int s, int ci, int g, int& od, int& gg, int& bh, int& bw) {
            int c = ci, st = s;
            gg = g;
            while (gg >= GM::G) {
                gg -= GM::G;
                if (++c == WC::n) {
                    c = 0;
                    st++;
                }
            }
            const int cl = WC::cls[c];
            od = st < dl ? 2 * (d0 + st) + (cl >> 2) : -1;
            bh = (cl >> 1) & 1;
            bw = cl & 1;
        };
        if constexpr (HAS_RES) {
            static_for<0, U>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                int od, gg, bh, bw;
                item(0, 0, k, od, gg, bh, bw);
                if (od >= 0) res_fetch(rq[k], od, gg, bh, bw);
            });
        }

        consumer_barrier();      // prologue planes have landed
        for (int s = 0; s < dl; s++) {
            const int aa = d0 + s;                               // input plane of this step
            const int slot0 = (s % GM::RING) * GM::PLANE_B, slot1 = ((s + 1) % GM::RING) * GM::PLANE_B;
            const int x2slot = (s & 1) * GM::X2_B;
            static_for<0, WC::n>([&](auto ic) {
                constexpr int ci = decltype(ic)::value;
                constexpr int cl = WC::cls[ci];
                constexpr int bd = cl >> 2, bh = (cl >> 1) & 1, bw = cl & 1;
                const int od = 2 * aa + bd;
                const bool slow = edge || od >= a.Do || (a.zero_far && od == a.Do - 1) || far_hw;
                // two copies of the class body (the common one has no per-lane tests), each a two-stage pipeline written
                // out: stage k = the MFMAs of group k + 1 next to the epilogue of group k (two accumulator sets), with a
                // scheduling barrier between stages - left to itself the compiler interleaves all eight groups and spills
                auto run = [&](auto slowc) {
                constexpr bool SLOW = decltype(slowc)::value;
                static_assert(GM::G == U, "one chunk per class");
                f32x4 accs[2][2];
                auto mf = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int hb = k / (TW / 16), wc = (k % (TW / 16)) * 16;
                    // the accumulators start from the bias of the 8 channels this lane ends up with
                    f32x4 acc0 = {bias8[0], bias8[1], bias8[2], bias8[3]}, acc1 = {bias8[4], bias8[5], bias8[6], bias8[7]};
                    const char* base = lane_lds + (hb * GM::LW + wc) * GM::PITCH_B;
                    static_for<0, class_taps(cl)>([&](auto tc) {
                        constexpr int ti = decltype(tc)::value;
                        constexpr TapInfo ti_ = class_tap(cl, ti);
                        static_for<0, KS>([&](auto kk) {
                            constexpr int ks = decltype(kk)::value;
                            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(
                                base + (ti_.dd ? slot1 : slot0) + (ti_.dh * GM::LW + ti_.dw) * GM::PITCH_B + ks * 64);
                            constexpr int wi = ((class_tap_base<WV>(ci) + ti) * KS + ks) * 2;
                            acc0 = RU3D_MFMA_16X16X32(wreg[wi], xb, acc0, 0, 0, 0);
                            acc1 = RU3D_MFMA_16X16X32(wreg[wi + 1], xb, acc1, 0, 0, 0);
                        });
                    });
                    if constexpr (X2W && cl == 0) {
                        static_for<0, KS>([&](auto kk) {
                            constexpr int ks = decltype(kk)::value;
                            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(lane_x2 + x2slot + (hb * TW + wc) * GM::PITCH_B + ks * 64);
                            acc0 = RU3D_MFMA_16X16X32(w2reg[ks * 2], xb, acc0, 0, 0, 0);
                            acc1 = RU3D_MFMA_16X16X32(w2reg[ks * 2 + 1], xb, acc1, 0, 0, 0);
                        });
                    }
                    accs[k & 1][0] = acc0;
                    accs[k & 1][1] = acc1;
                };
                auto epi = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int hb = k / (TW / 16), wc = (k % (TW / 16)) * 16;
                    // ---- channels cb * 32 + 8 g4 .. of the lane's voxel
                    f32x4 acc0 = accs[k & 1][0], acc1 = accs[k & 1][1];
                    // (the wait states between a 16x16x32 MFMA and a VALU read of its last result registers: observed
                    // short by the compiler's own count when the conversion follows the MFMA directly - the last group)
                    if constexpr (k == U - 1) asm("s_nop 7\n\ts_nop 4" : "+v"(acc0), "+v"(acc1));
                    bool ok = true;
                    f32x8 v = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]};
                    if constexpr (SLOW) {
                        bool far;
                        ok = lane_ok(od, hb, wc, bh, bw, far);
                        if (a.zero_far) {
#pragma unroll
                            for (int i = 0; i < 8; i++) v[i] = far ? 0.f : v[i];
                        }
                    }
                    if constexpr (HAS_RES) {
#pragma unroll
                        for (int i = 0; i < 8; i++) v[i] += (float)rq[k][i];
                        // refill for the item U groups on (the next class, or the next step's first)
                        int od2, gg, bh2, bw2;
                        item(s, ci, k + U, od2, gg, bh2, bw2);
                        if (od2 >= 0) res_fetch(rq[k], od2, gg, bh2, bw2);
                    }
                    const bf16x8 o = __builtin_convertvector(v, bf16x8);
                    if constexpr (HAS_STATS) {
                        if (ok) {
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const float f = (float)o[i];
                                st1[i] += f;
                                st2[i] = fmaf(f, f, st2[i]);
                            }
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, o), ry, ok ? y_lane : OOB,
                                                           uni_vox(od, hb, wc, bh, bw) * a.ldy * 2, 0);
                };
                mf(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, U>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k + 1 < U) mf(std::integral_constant<int, k + 1>{});
                    epi(kc);
                    __builtin_amdgcn_sched_barrier(0);
                });
                };
                if (slow) run(std::true_type{});
                else run(std::false_type{});
            });
            consumer_barrier();      // done with plane aa; plane aa + 2 (and the next x2 tile) have landed
        }
    }
    stat_flush();
}

template <int CIN, int TW, bool HAS_X2>
__device__ __forceinline__ void t_producer(const S2Args& a, char* lds, char* lds_x2) {
    using GM = TGeom<CIN, TW>;
    const int lane = threadIdx.x & 63;
    // tile-invariant part of every DMA slot: row (zb, zc) and byte column, or -1 for a pad / tail slot
    int slot_id[GM::ND];
#pragma unroll
    for (int i = 0; i < GM::ND; i++) {
        const int byte = i * 1024 + lane * 16;
        const int row = byte / GM::PITCH_B, col = byte - row * GM::PITCH_B;
        const bool ok = row < GM::ROWS && col < CIN * 2;
        slot_id[i] = ok ? ((row / GM::LW) | ((row % GM::LW) << 8) | (col << 16)) : -1;
    }
    int slot2_id[HAS_X2 ? GM::ND2 : 1];
    if constexpr (HAS_X2) {
#pragma unroll
        for (int i = 0; i < GM::ND2; i++) {
            const int byte = i * 1024 + lane * 16;
            const int row = byte / GM::PITCH_B, col = byte - row * GM::PITCH_B;
            const bool ok = row < GM::X2ROWS && col < CIN * 2;
            slot2_id[i] = ok ? ((row / TW) | ((row % TW) << 8) | (col << 16)) : -1;
        }
    }
    const int plane_b = a.Hi * a.Wi * a.ldx * 2, sample_b = a.Di * plane_b;
    const int plane2_b = a.Hi * a.Wi * a.ldx2 * 2, sample2_b = a.Di * plane2_b;

    const int Gx = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (Gx % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += Gx) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, b0 = th_i * TH, c0 = tw_i * TW;
        int dl = a.Di - d0;
        if (dl > a.DL) dl = a.DL;

        int voff[GM::ND];
#pragma unroll
        for (int i = 0; i < GM::ND; i++) {
            const int id = slot_id[i];
            const int ib = b0 + (id & 255), ic = c0 + ((id >> 8) & 255);
            voff[i] = (id >= 0 && ib < a.Hi && ic < a.Wi && !(a.dbg & 2)) ? (ib * a.Wi + ic) * a.ldx * 2 + (id >> 16) : OOB;
        }
        int voff2[HAS_X2 ? GM::ND2 : 1];
        if constexpr (HAS_X2) {
#pragma unroll
            for (int i = 0; i < GM::ND2; i++) {
                const int id = slot2_id[i];
                const int ib = b0 + (id & 255), ic = c0 + ((id >> 8) & 255);
                voff2[i] = (id >= 0 && ib < a.Hi && ic < a.Wi) ? (ib * a.Wi + ic) * a.ldx2 * 2 + (id >> 16) : OOB;
            }
        }
        const bf16* xs = a.x + (int64_t)n * (sample_b / 2);
        auto issue_plane = [&](int pl, int slot) {          // input plane pl (may be Di: zeros) -> ring slot
            const ru3d_i32x4 rs = ru3d_buffer_rsrc(xs, pl < a.Di ? sample_b : 0);
            const int base = pl < a.Di ? pl * plane_b : 0;
            char* dst = lds + slot * GM::PLANE_B;
#pragma unroll
            for (int i = 0; i < GM::ND; i++) ru3d_lds_dma16(rs, dst + i * 1024, voff[i] + base);
        };
        auto issue_x2 = [&](int pl, int slot) {
            if constexpr (HAS_X2) {
                const ru3d_i32x4 rs = ru3d_buffer_rsrc(a.x2 + (int64_t)n * (sample2_b / 2), pl < a.Di ? sample2_b : 0);
                const int base = pl < a.Di ? pl * plane2_b : 0;
                char* dst = lds_x2 + slot * GM::X2_B;
#pragma unroll
                for (int i = 0; i < GM::ND2; i++) ru3d_lds_dma16(rs, dst + i * 1024, voff2[i] + base);
            }
        };
        // the previous unit's last barrier has passed: every consumer is done with the ring
        issue_plane(d0, 0);
        issue_plane(d0 + 1, 1);
        issue_x2(d0, 0);
        producer_barrier();
        for (int s = 0; s < dl; s++) {
            if (s + 1 < dl) {
                issue_plane(d0 + s + 2, (s + 2) % GM::RING);
                issue_x2(d0 + s + 1, (s + 1) & 1);
            }
            producer_barrier();
        }
    }
}

template <int CIN, int TW, bool HAS_RES, bool HAS_STATS, bool HAS_X2>
__global__ __launch_bounds__(320) void convt3_s2_tile_kernel(S2Args a) {
    using GM = TGeom<CIN, TW>;
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    char* lds = lds_raw;
    char* lds_x2 = lds_raw + GM::RING * GM::PLANE_B;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == NCONS) t_producer<CIN, TW, HAS_X2>(a, lds, lds_x2);
    else if (wave == 0) t_consumer<CIN, TW, HAS_RES, HAS_STATS, HAS_X2, 0>(a, lds, lds_x2);
    else if (wave == 1) t_consumer<CIN, TW, HAS_RES, HAS_STATS, HAS_X2, 1>(a, lds, lds_x2);
    else if (wave == 2) t_consumer<CIN, TW, HAS_RES, HAS_STATS, HAS_X2, 2>(a, lds, lds_x2);
    else t_consumer<CIN, TW, HAS_RES, HAS_STATS, HAS_X2, 3>(a, lds, lds_x2);
}

template <int CIN, int TW>
constexpr size_t t_lds_bytes(bool x2) {
    return (size_t)TGeom<CIN, TW>::RING * TGeom<CIN, TW>::PLANE_B + (x2 ? 2 * TGeom<CIN, TW>::X2_B : 0);
}


// --------------------------------------------------------------------------------------------------- G form
// out[od, oh, ow] = sum_k in[2 od - 1 + kd, 2 oh - 1 + kh, 2 ow - 1 + kw] . W[k]: a workgroup owns TH x GW output positions
// and slides along the output's D; step od needs the input planes 2 od - 1 .. 2 od + 1 (two new ones per step, ring of
// five).  A staged plane is 2 TH + 1 lines of 2 GW + 1 rows, every line de-interleaved by the parity of w: GW + 1 odd rows
// (w = 2 (c0 + e) - 1) followed by GW even ones (w = 2 (c0 + e)) - the 16 positions of a tap's fragment are then 16
// consecutive rows: kw = 0 -> odd rows e = p, kw = 1 -> even rows e = p, kw = 2 -> odd rows e = p + 1.
constexpr int GW = 16;
template <int CIN>
struct GGeom {
    static constexpr int KS = CIN / 32;
    static constexpr int PITCH_B = CIN * 2 + 32;
    static constexpr int LINE = 2 * GW + 1, NODD = GW + 1, LINES = 2 * TH + 1, ROWS = LINES * LINE;
    static constexpr int ND = (ROWS * PITCH_B + 1023) / 1024;
    static constexpr int PLANE_B = ND * 1024;
    static constexpr int RING = 5;
};

template <int CIN, bool HAS_STATS, bool HAS_Y2>
__device__ __forceinline__ void g_consumer(const S2Args& a, const char* lds, int wave) {
    using GM = GGeom<CIN>;
    constexpr int KS = GM::KS;
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, g4 = lane >> 4;
    const int cb = blockIdx.y * 64 + 16 * wave;        // the wave's 16 output channels
    const int NTT = a.cout_total / 32;

    // ---- weights: [tap][k-step], A fragment lane -> output channel cb + (lane & 15), k-block lane >> 4
    bf16x8 wreg[27 * KS];
    static_for<0, 27>([&](auto tc) {
        constexpr int tap = decltype(tc)::value;
        const int wtap = a.flip ? 26 - tap : tap;
        static_for<0, KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            wreg[tap * KS + ks] = a.w[packed_unit(wtap, CIN / 16, NTT, cb + p, ks * 4 + g4)];
        });
    });
    bf16x8 w2reg[HAS_Y2 ? KS : 1];
    if constexpr (HAS_Y2) {
        static_for<0, KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            w2reg[ks] = a.w2[packed_unit(0, CIN / 16, NTT, cb + p, ks * 4 + g4)];
        });
    }
    float bias4[4], bias24[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bias4[i] = a.bias ? a.bias[cb + 4 * g4 + i] : 0.f;
        bias24[i] = (HAS_Y2 && a.bias2) ? a.bias2[cb + 4 * g4 + i] : 0.f;
    }

    float st1[4], st2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) st1[i] = st2[i] = 0.f;
    int cur_n = -1;
    if (HAS_STATS) ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave, a.N, 64 * 2);
    auto stat_flush = [&]() {
        if (!HAS_STATS || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * a.N + cur_n) * 64) * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float s1 = st1[i], s2 = st2[i];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {       // lanes with equal g4 hold the same 4 channels
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (p == 0) {      // += : see the T form
                dst[(16 * wave + 4 * g4 + i) * 2] += s1;
                dst[(16 * wave + 4 * g4 + i) * 2 + 1] += s2;
            }
            st1[i] = st2[i] = 0.f;
        }
    };

    const char* const lane_lds = lds + p * GM::PITCH_B + g4 * 16;
    const int y_lane = (p * a.ldy + cb + 4 * g4) * 2, y2_lane = (p * a.ldy2 + cb + 4 * g4) * 2;
    const int ysample_b = a.Do * a.Ho * a.Wo * a.ldy * 2, y2sample_b = a.Do * a.Ho * a.Wo * a.ldy2 * 2;

    const int Gx = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (Gx % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += Gx) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, b0 = th_i * TH, c0 = tw_i * GW;
        int dl = a.Do - d0;
        if (dl > a.DL) dl = a.DL;
        if (HAS_STATS && n != cur_n) {
            stat_flush();
            cur_n = n;
        }
        const rsrc_t ry = make_rsrc(a.y + (int64_t)n * (ysample_b / 2), ysample_b);
        rsrc_t ry2 = ry;
        if constexpr (HAS_Y2) ry2 = make_rsrc(a.y2 + (int64_t)n * (y2sample_b / 2), y2sample_b);
        const bool lane_in = c0 + p < a.Wo && !(a.dbg & 1);

        consumer_barrier();      // prologue planes have landed
        for (int s = 0; s < dl; s++) {
            const int od = d0 + s;
            int slot[3];
#pragma unroll
            for (int kd = 0; kd < 3; kd++) slot[kd] = ((2 * s + kd) % GM::RING) * GM::PLANE_B;
            static_for<0, TH>([&](auto hc) {
                constexpr int hb = decltype(hc)::value;
                f32x4 acc = {bias4[0], bias4[1], bias4[2], bias4[3]};
                f32x4 acc2 = {bias24[0], bias24[1], bias24[2], bias24[3]};
                static_for<0, 27>([&](auto tc) {
                    constexpr int tap = decltype(tc)::value;
                    constexpr int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                    constexpr int row = (2 * hb + kh) * GM::LINE + (kw == 1 ? GM::NODD : (kw >> 1));
                    static_for<0, KS>([&](auto kc) {
                        constexpr int ks = decltype(kc)::value;
                        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(lane_lds + slot[kd] + row * GM::PITCH_B + ks * 64);
                        acc = RU3D_MFMA_16X16X32(wreg[tap * KS + ks], xb, acc, 0, 0, 0);
                        if constexpr (HAS_Y2 && tap == 13) acc2 = RU3D_MFMA_16X16X32(w2reg[ks], xb, acc2, 0, 0, 0);
                    });
                });
                // (wait states between the last MFMA and the VALU read of its result: see the T form)
                asm("s_nop 7\n\ts_nop 4" : "+v"(acc), "+v"(acc2));
                const bool ok = lane_in && b0 + hb < a.Ho;
                const int uni = (od * a.Ho + b0 + hb) * a.Wo + c0;
                const bf16x4 o = __builtin_convertvector(acc, bf16x4);
                if constexpr (HAS_STATS) {
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float f = (float)o[i];
                            st1[i] += f;
                            st2[i] = fmaf(f, f, st2[i]);
                        }
                    }
                }
                {
                    typedef int i32x2v __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2v, o), ry, ok ? y_lane : OOB, uni * a.ldy * 2, 0);
                    if constexpr (HAS_Y2) {
                        const bf16x4 o2 = __builtin_convertvector(acc2, bf16x4);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2v, o2), ry2, ok ? y2_lane : OOB,
                                                              uni * a.ldy2 * 2, 0);
                    }
                }
            });
            consumer_barrier();      // done with planes 2 s, 2 s + 1; the next two have landed
        }
    }
    stat_flush();
}

template <int CIN>
__device__ __forceinline__ void g_producer(const S2Args& a, char* lds) {
    using GM = GGeom<CIN>;
    const int lane = threadIdx.x & 63;
    // tile-invariant part of every DMA slot: (line zh, w relative to 2 c0 as dw + 1 in 0..2 GW, byte column) or -1
    int slot_id[GM::ND];
#pragma unroll
    for (int i = 0; i < GM::ND; i++) {
        const int byte = i * 1024 + lane * 16;
        const int row = byte / GM::PITCH_B, col = byte - row * GM::PITCH_B;
        const bool ok = row < GM::ROWS && col < CIN * 2;
        const int zh = row / GM::LINE, r = row - zh * GM::LINE;
        const int dw1 = r < GM::NODD ? 2 * r : 2 * (r - GM::NODD) + 1;      // (w - (2 c0 - 1)): odd rows 0, 2, .., even rows 1, 3, ..
        slot_id[i] = ok ? (zh | (dw1 << 8) | (col << 16)) : -1;
    }
    const int plane_b = a.Hi * a.Wi * a.ldx * 2, sample_b = a.Di * plane_b;

    const int Gx = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (Gx % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += Gx) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, b0 = th_i * TH, c0 = tw_i * GW;
        int dl = a.Do - d0;
        if (dl > a.DL) dl = a.DL;

        int voff[GM::ND];
#pragma unroll
        for (int i = 0; i < GM::ND; i++) {
            const int id = slot_id[i];
            const int ih = 2 * b0 - 1 + (id & 255), iw = 2 * c0 - 1 + ((id >> 8) & 255);
            voff[i] = (id >= 0 && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi && !(a.dbg & 2)) ? (ih * a.Wi + iw) * a.ldx * 2 + (id >> 16) : OOB;
        }
        const bf16* xs = a.x + (int64_t)n * (sample_b / 2);
        auto issue_plane = [&](int j) {          // relative plane j: input plane 2 d0 - 1 + j -> ring slot j % RING
            const int pl = 2 * d0 - 1 + j;
            const bool inside = pl >= 0 && pl < a.Di;
            const ru3d_i32x4 rs = ru3d_buffer_rsrc(xs, inside ? sample_b : 0);
            const int base = inside ? pl * plane_b : 0;
            char* dst = lds + (j % GM::RING) * GM::PLANE_B;
#pragma unroll
            for (int i = 0; i < GM::ND; i++) ru3d_lds_dma16(rs, dst + i * 1024, voff[i] + base);
        };
        issue_plane(0);
        issue_plane(1);
        issue_plane(2);
        producer_barrier();
        for (int s = 0; s < dl; s++) {
            if (s + 1 < dl) {
                issue_plane(2 * s + 3);
                issue_plane(2 * s + 4);
            }
            producer_barrier();
        }
    }
}

template <int CIN, bool HAS_STATS, bool HAS_Y2>
__global__ __launch_bounds__(320) void conv3_s2_tile_kernel(S2Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == NCONS) g_producer<CIN>(a, lds_raw);
    else g_consumer<CIN, HAS_STATS, HAS_Y2>(a, lds_raw, wave);
}

}  // namespace

// --------------------------------------------------------------------------------------------------- host: T form
// Work decomposition: units = N x dsplit x ceil(Hi / 4) x ceil(Wi / TW) columns of DL input planes each.
static bool t_plan(int N, int Di, int Hi, int Wi, int Cin, int Cout, int* tw_out, SlidePlan* out) {
    static const int mode = getenv("RU3D_CONV_S2") ? atoi(getenv("RU3D_CONV_S2")) : 1;
    if (!mode || Cin != 64 || (Cout % 32) || Cout > 64 || Di < 2) return false;
    const int ny = Cout / 32;
    int64_t best_cost = -1;
    int best_ds = 0, best_tw = 0;
    // 16-wide tiles only: the 32-wide instantiation (half the halo rows per voxel) measured slower on the same box - its
    // consumers spill and a step carries twice the work between barriers (189 vs 148 us, 103 vs 97 us at 64^3 -> 128^3)
    for (int tw = 16; tw <= 16; tw += 16) {
        const int64_t cols = (int64_t)N * ((Hi + TH - 1) / TH) * ((Wi + tw - 1) / tw);
        for (int ds = 1; ds <= Di; ds++) {
            const int dl = (Di + ds - 1) / ds;
            if ((int64_t)dl * (ds - 1) >= Di) continue;          // an empty last chunk
            const int64_t units = cols * ds;
            int64_t gx = ru3d_get_cu_budget() / ny;
            if (gx > units) gx = units;
            // a step of the wide tile moves twice the voxels of a narrow one
            const int64_t cost = ((units + gx - 1) / gx) * (int64_t)(dl + 2) * tw;
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best_ds = ds;
                best_tw = tw;
            }
        }
    }
    if (!best_ds) return false;
    const int64_t cols = (int64_t)N * ((Hi + TH - 1) / TH) * ((Wi + best_tw - 1) / best_tw);
    const int64_t units = cols * best_ds;
    if (units * ny < 128 || units > 0x7fffffff) return false;      // too small to fill the chip: the other kernels
    out->dsplit = best_ds;
    out->DL = (Di + best_ds - 1) / best_ds;
    out->tiles_h = (Hi + TH - 1) / TH;
    out->tiles_w = (Wi + best_tw - 1) / best_tw;
    out->units = (int)units;
    int g = units < ru3d_get_cu_budget() / ny ? (int)units : ru3d_get_cu_budget() / ny;
    if ((units % 8) == 0 && g >= 8) g = (g / 8) * 8;
    out->grid = g;
    out->ny = ny;
    *tw_out = best_tw;
    return true;
}

bool convt_s2_tile_eligible(const ConvGeom& g) {
    SlidePlan sp;
    int tw;
    if (!(g.transposed && g.k == 3 && g.stride == 2 && g.pad == 1)) return false;
    if ((g.ldx % 8) || (g.ldy % 8) || (g.ldr % 8)) return false;
    // 32-bit byte offsets inside a sample, top bit = "outside"
    if ((int64_t)g.Di * g.Hi * g.Wi * g.ldx >= (1ll << 30) || (int64_t)g.Do * g.Ho * g.Wo * g.ldy >= (1ll << 30) ||
        (int64_t)g.Do * g.Ho * g.Wo * g.ldr >= (1ll << 30))
        return false;
    if ((g.Do + 1) / 2 != g.Di || (g.Ho + 1) / 2 != g.Hi || (g.Wo + 1) / 2 != g.Wi) return false;
    return t_plan(g.N, g.Di, g.Hi, g.Wi, g.Cin, g.Cout, &tw, &sp);
}

size_t convt_s2_tile_slab_bytes(const ConvGeom& g) {
    SlidePlan sp;
    int tw;
    if (!t_plan(g.N, g.Di, g.Hi, g.Wi, g.Cin, g.Cout, &tw, &sp)) return 0;
    return (size_t)sp.grid * sp.ny * 4 * g.N * 32 * 2 * sizeof(float);
}

int convt_s2_tile_slab_geom(const ConvGeom& g, int* gx, int* cb) {
    SlidePlan sp;
    int tw;
    if (!t_plan(g.N, g.Di, g.Hi, g.Wi, g.Cin, g.Cout, &tw, &sp)) return -1;
    *gx = sp.grid;
    *cb = 32;
    return 0;
}

template <int CIN, int TW>
static int t_launch(const S2Args& a, const SlidePlan& p, bool has_res, bool has_stats, bool has_x2, hipStream_t st) {
    const dim3 grid(p.grid, p.ny), block(320);
    const size_t lds = t_lds_bytes<CIN, TW>(has_x2);
#define RU3D_T_LAUNCH(R, S, X)                                                                                    \
    do {                                                                                                          \
        auto kern = convt3_s2_tile_kernel<CIN, TW, R, S, X>;                                                      \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != \
                hipSuccess)                                                                                       \
                return ru3d_fail(-1, "convt_s2_tile: cannot raise the dynamic LDS limit");                        \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                        \
    } while (0)
    if (has_x2) {
        if (has_res) RU3D_T_LAUNCH(true, false, true);
        else RU3D_T_LAUNCH(false, false, true);
    } else if (has_stats) {
        RU3D_T_LAUNCH(false, true, false);
    } else if (has_res) {
        RU3D_T_LAUNCH(true, false, false);
    } else {
        RU3D_T_LAUNCH(false, false, false);
    }
#undef RU3D_T_LAUNCH
    return ru3d_check_launch("convt3_s2_tile");
}

// x2 / w2: the 1x1x1 stride-2 partner's gradient operand (on the input grid, Cin channels, pitch ldx2) and its packed
// input-gradient weight; stat_slab: InstanceNorm sums of the output (no residual then)
int convt_s2_tile_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                         float* stat_slab, const void* x2, int ldx2, const void* w2, hipStream_t st) {
    SlidePlan p;
    int tw;
    if (!convt_s2_tile_eligible(g) || !t_plan(g.N, g.Di, g.Hi, g.Wi, g.Cin, g.Cout, &tw, &p))
        return ru3d_fail(-1, "convt_s2_tile: shape not supported");
    if ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res) | ((uintptr_t)x2) | ((uintptr_t)w) | ((uintptr_t)w2)) % 16)
        return ru3d_fail(-1, "convt_s2_tile: operands must be 16-byte aligned");
    if (stat_slab && (res || x2)) return ru3d_fail(-1, "convt_s2_tile: statistics and residual / partner cannot be combined");
    if (x2 && ((ldx2 % 8) || !w2 || (int64_t)g.Di * g.Hi * g.Wi * ldx2 >= (1ll << 30)))
        return ru3d_fail(-1, "convt_s2_tile: bad partner operand");
    S2Args a;
    a.x = (const bf16*)x; a.w = (const bf16x8*)w; a.bias = bias; a.res = (const bf16*)res; a.y = (bf16*)y;
    a.stat_slab = stat_slab;
    a.x2 = (const bf16*)x2; a.w2 = (const bf16x8*)w2; a.bias2 = nullptr; a.y2 = nullptr;
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo;
    a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = res ? g.ldr : g.ldy; a.ldx2 = x2 ? ldx2 : g.ldx; a.ldy2 = 0;
    a.flip = g.flip; a.zero_far = g.zero_far; a.cout_total = g.Cout;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
    static const int dbg = getenv("RU3D_S2_DBG") ? atoi(getenv("RU3D_S2_DBG")) : 0;
    a.dbg = dbg;
    (void)tw;
    return t_launch<64, 16>(a, p, res != nullptr, stat_slab != nullptr, x2 != nullptr, st);
}

// --------------------------------------------------------------------------------------------------- host: G form
static bool g_plan(int N, int Do, int Ho, int Wo, int Cin, int Cout, SlidePlan* out) {
    static const int mode = getenv("RU3D_CONV_S2") ? atoi(getenv("RU3D_CONV_S2")) : 1;
    if (!mode || Cin != 32 || (Cout % 64) || Cout > 128 || Do < 2) return false;
    const int ny = Cout / 64;
    const int64_t cols = (int64_t)N * ((Ho + TH - 1) / TH) * ((Wo + GW - 1) / GW);
    int64_t best_cost = -1;
    int best_ds = 0;
    for (int ds = 1; ds <= Do; ds++) {
        const int dl = (Do + ds - 1) / ds;
        if ((int64_t)dl * (ds - 1) >= Do) continue;
        const int64_t units = cols * ds;
        int64_t gx = ru3d_get_cu_budget() / ny;
        if (gx > units) gx = units;
        const int64_t cost = ((units + gx - 1) / gx) * (int64_t)(2 * dl + 3);      // planes through a workgroup
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_ds = ds;
        }
    }
    if (!best_ds) return false;
    const int64_t units = cols * best_ds;
    if (units * ny < 128 || units > 0x7fffffff) return false;
    out->dsplit = best_ds;
    out->DL = (Do + best_ds - 1) / best_ds;
    out->tiles_h = (Ho + TH - 1) / TH;
    out->tiles_w = (Wo + GW - 1) / GW;
    out->units = (int)units;
    int g = units < ru3d_get_cu_budget() / ny ? (int)units : ru3d_get_cu_budget() / ny;
    if ((units % 8) == 0 && g >= 8) g = (g / 8) * 8;
    out->grid = g;
    out->ny = ny;
    return true;
}

bool conv_s2_tile_eligible(const ConvGeom& g) {
    SlidePlan sp;
    if (g.transposed || g.k != 3 || g.stride != 2 || g.pad != 1) return false;
    if ((g.ldx % 8) || (g.ldy % 4)) return false;
    if ((int64_t)g.Di * g.Hi * g.Wi * g.ldx >= (1ll << 30) || (int64_t)g.Do * g.Ho * g.Wo * g.ldy >= (1ll << 30)) return false;
    if (g.Do != (g.Di - 1) / 2 + 1 || g.Ho != (g.Hi - 1) / 2 + 1 || g.Wo != (g.Wi - 1) / 2 + 1) return false;
    return g_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp);
}

size_t conv_s2_tile_slab_bytes(const ConvGeom& g) {
    SlidePlan sp;
    if (!g_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return 0;
    return (size_t)sp.grid * sp.ny * 4 * g.N * 64 * 2 * sizeof(float);
}

int conv_s2_tile_slab_geom(const ConvGeom& g, int* gx, int* cb) {
    SlidePlan sp;
    if (!g_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return -1;
    *gx = sp.grid;
    *cb = 64;
    return 0;
}

// w2 / bias2 / y2 (pitch ldy2): the 1x1x1 stride-2 conv of the same input as a second output; stat_slab: sums of y
int conv_s2_tile_launch(const void* x, const void* w, const float* bias, void* y, const ConvGeom& g, float* stat_slab,
                        const void* w2, const float* bias2, void* y2, int ldy2, hipStream_t st) {
    SlidePlan p;
    if (!conv_s2_tile_eligible(g) || !g_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &p))
        return ru3d_fail(-1, "conv_s2_tile: shape not supported");
    if ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)w2)) % 16 || (((uintptr_t)y) | ((uintptr_t)y2)) % 8)
        return ru3d_fail(-1, "conv_s2_tile: operands must be 16-byte (x, w) / 8-byte (y) aligned");
    if (y2 && (!w2 || (ldy2 % 4) || (int64_t)g.Do * g.Ho * g.Wo * ldy2 >= (1ll << 30)))
        return ru3d_fail(-1, "conv_s2_tile: bad second output");
    S2Args a;
    a.x = (const bf16*)x; a.w = (const bf16x8*)w; a.bias = bias; a.res = nullptr; a.y = (bf16*)y;
    a.stat_slab = stat_slab;
    a.x2 = nullptr; a.w2 = (const bf16x8*)w2; a.bias2 = bias2; a.y2 = (bf16*)y2;
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo;
    a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldy; a.ldx2 = g.ldx; a.ldy2 = y2 ? ldy2 : g.ldy;
    a.flip = g.flip; a.zero_far = 0; a.cout_total = g.Cout;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
    static const int dbg = getenv("RU3D_S2_DBG") ? atoi(getenv("RU3D_S2_DBG")) : 0;
    a.dbg = dbg;
    const dim3 grid(p.grid, p.ny), block(320);
    const size_t lds = (size_t)GGeom<32>::RING * GGeom<32>::PLANE_B;
#define RU3D_G_LAUNCH(S, Y2)                                                                                      \
    do {                                                                                                          \
        auto kern = conv3_s2_tile_kernel<32, S, Y2>;                                                              \
        static bool attr = false;                                                                                 \
        if (!attr) {                                                                                              \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != \
                hipSuccess)                                                                                       \
                return ru3d_fail(-1, "conv_s2_tile: cannot raise the dynamic LDS limit");                         \
            attr = true;                                                                                          \
        }                                                                                                         \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                        \
    } while (0)
    if (stat_slab && y2) RU3D_G_LAUNCH(true, true);
    else if (stat_slab) RU3D_G_LAUNCH(true, false);
    else if (y2) RU3D_G_LAUNCH(false, true);
    else RU3D_G_LAUNCH(false, false);
#undef RU3D_G_LAUNCH
    return ru3d_check_launch("conv3_s2_tile");
}

}  // namespace RU3D_NS
