// 3x3x3 stride-1 conv, 32 -> 32 (or 64) channels, 16-bit MFMA, for the full-resolution level (reference
// network.py:391-403: conv1 / conv2 of the level-0 ResBlocks at 128^3, forward and input gradient) - the roofline kernel
// of bench.py.
//
// D-sliding plane ring as in conv_slide64.hip, laid out for K = 32:
//   * v_mfma_f32_16x16x32: one MFMA consumes the whole input-channel depth of a tap.  A wave owns ALL 32 output
//     channels (two 16-channel tiles) of a 4 x 16 (H, W) quarter of the workgroup's 8 x 32 column, and keeps the whole
//     weight - 27 taps x 2 channel tiles = 54 fragments = 216 registers - in AGPRs for the whole launch; Cout = 64 runs
//     as two slices (grid.y);
//   * every activation fragment read from LDS (16 voxels x 32 channels, 1 KB) feeds TWO MFMAs (the two channel tiles),
//     and the waves read disjoint quarters instead of all reading everything: 6 fragment reads per 24 MFMAs in a
//     (kd, kw) pass - half the LDS bytes per flop of the 32x32x16 form this kernel replaces (4 reads per 6 MFMAs of
//     twice the size), whose matrix pipe waited on them;
//   * 4-plane LDS ring of 10 x 34 halo rows x 64 B of channels (pitch 96 B: conflict-free fragment reads, see PITCH); one new 27 KB plane per 256 output voxels, staged with buffer loads (pieces outside
//     the volume come back as zeros: no select, no address clamp), one LDS-only barrier per plane;
//   * two accumulator sets: the epilogue of plane s (bias, rounding to the storage type, wave-private LDS patch, then
//     4 lanes per voxel -> 16-byte stores, residual, fused InstanceNorm sums) and the plane staging are cut into
//     micro-operations pinned behind the MFMA pairs of plane s+1 (the schedule is spelled out at the kernel).
// Packed weights are the library's ordinary 32x32x16 fragment order; the 16x16x32 fragments are gathered from it in the
// prologue.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {
namespace {
constexpr int TH = 8, TW = 32, HH = TH + 2, WW = TW + 2;
constexpr int RH = 4, RI = RH + 2;        // output rows of a wave, input rows it walks
constexpr int PROWS = HH * WW;            // 340 halo rows per plane
constexpr int NSTG = 6;                   // 16-byte pieces staged per thread and plane: ceil(340 * 4 / 256)
// 16-bit elements per staged row: 64 B of channels + 32 B pad.  ds_read_b128 serves four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) from 64 banks: lane (voxel v, k-block kb) of a fragment read starts at
// 16-byte unit 6 v + kb, and within every group the kb = even lanes land on 8 distinct even units, the kb = odd lanes
// on 8 distinct odd ones (mod 16) - conflict-free.  (80 B rows put three lanes of every group on an occupied unit.)
constexpr int PITCH = 48;
constexpr int PLANE = PROWS * PITCH;      // elements per plane (32,640 B)
constexpr int RING = 4;
constexpr int NP = 9;                     // (kd, kw) passes per step
constexpr int NJ = 12;                    // (input row, kh) MFMA pairs per pass: 24 MFMAs
constexpr int EP = 40;                    // 16-bit elements per epilogue-patch row (32 couts + 8 pad = 80 B)
static_assert(RING * PLANE * 2 + 4 * 64 * EP * 2 <= 160 * 1024, "LDS budget");
static_assert(NSTG * 64 >= PROWS, "staging pieces");
// pair j of a pass: input row JR[j], tap row JK[j] -> output row JR - JK; both channel tiles
constexpr int JR[NJ] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5};
constexpr int JK[NJ] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 1, 2, 2};
constexpr bool last_of_row(int j) { return j == 0 || j == 2 || j == 5 || j == 8 || j == 10 || j == 11; }

struct Slide32Args {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    const bf16x8* w2;                     // HAS_X2: packed input-gradient weight of the 1x1x1 partner
    bf16* y;
    float* stat_slab;
    int N, D, H, W;
    int ldx, ldy, ldr;
    int64_t yslice;         // elements between the 32-channel output slices (blockIdx.y): 32 dense, the plane distance when split
    int flip;
    int cout_total;                       // Cout of the conv (32 per grid.y slice)
    float slope, inv_slope;               // HAS_BST: LeakyReLU slope of the activation being differentiated
    int tiles_h, tiles_w, dsplit, DL, units;
#ifdef RU3D_SLIDE_STAMPS
    long long* stamps;
#endif
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
#ifdef RU3D_STORAGE_F16
#define RU3D_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define RU3D_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
// Operand classes as in conv_slide64.hip: all 54 weight fragments in AGPRs (the MFMA reads its A operand there), the
// accumulators and activation fragments in VGPRs; tools/isa_check.py checks that nothing is copied into an MFMA operand
// next to the MFMAs (the compiler's hazard recogniser does not see inline-asm MFMAs).
template <bool ZERO>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& w, const bf16x8& x) {
    const i32x4 wi = __builtin_bit_cast(i32x4, w), xi = __builtin_bit_cast(i32x4, x);
    if constexpr (ZERO) asm volatile(RU3D_MFMA16_ASM " %0, %1, %2, 0" : "=v"(acc) : "a"(wi), "v"(xi));
    else asm volatile(RU3D_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc) : "a"(wi), "v"(xi));
}

// The schedule of a step.  One wave per SIMD issues one instruction at a time, and a 16x16x32 MFMA holds the vector
// issue port for 8 of its 16 cycles: two single-issue instructions fit behind every MFMA for free, a clump of thirty
// between two MFMAs idles the matrix pipe for its whole length.  So everything that is not an MFMA is cut into
// micro-operations of three or four instructions and pinned (sched_barrier) behind a fixed MFMA pair of the step:
//   the previous plane's epilogue, output row by output row, on the pairs that carry no staging: the row's 2 accumulator
//     tiles -> bias, rounding, wave-private LDS patch (4 micro-ops per tile: two adds, two adds, two converts, the
//     write), then its row phase (10 micro-ops: patch read, 8 x one channel's convert / statistics / residual, pack +
//     16-byte store) - 72 micro-ops on the first 72 of the 96 free pairs;
//   passes 3-8: LDS-only barrier in front of pass 3; per pass one 16-byte piece of plane s+3 to LDS (pair 3), the same
//     piece of plane s+4 from HBM into the freed registers (pair 6); in pass 8 pairs 1, 4, 7, 10 also issue the residual
//     loads of the plane being computed;
//   behind the last pair of every input row: that row's fragment for the next pass.
// The previous plane's epilogue always runs - at the first step of a column it finishes the last plane of the PREVIOUS
// column (pointers and accumulators carry over), so there is no per-step condition and no separate tail except once at
// the end of the workgroup; the workgroup's very first step "finishes" garbage onto plane 0 of its first column, which
// step 1 overwrites with the real values (same lanes, program order), and the statistics are reset behind it.
// HAS_BST (input-gradient role in front of an InstanceNorm + LeakyReLU backward, reference network.py:411-416): the
// conv output IS the gradient wrt the activation a = lrelu(xhat), and the two sums that backward needs,
// sum g' and sum g' * xhat with g' = g * lrelu'(a), xhat recovered from a, are taken in the row phase - `res` / `ldr`
// carry the activation, the statistics slab the sums: the separate reduction pass over (g, a) disappears.
// HAS_X2 (decoder ResBlock's input gradient, reference network.py:403-411): the 1x1x1 skip conv's input gradient
// W_s^T g_pre is added as a 28th tap whose activation fragment is not a halo tile but the voxel's own row of a SECOND
// tensor (`res` / `ldr` carry g_pre, `w2` its packed weight): the rows of the plane being computed are loaded straight
// into MFMA B-fragment layout during pass 0 and consumed by 8 extra MFMAs behind pass 8 - the separate 1x1 launch, its
// output and the residual read of that output disappear.
template <bool HAS_RES, bool HAS_STATS, bool HAS_BST = false, bool HAS_X2 = false>
__global__ __launch_bounds__(256, 1) void conv3_s1_slide32_kernel(Slide32Args a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[RING * PLANE];
    __shared__ __attribute__((aligned(16))) bf16 est_s[4 * 64 * EP];    // epilogue patches (stored values), one per wave
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef RU3D_SLIDE_STAMPS
    // diagnostic build (tools/stamps.py): cycle stamps of wave 0 at every pass (or MFMA pair) of four steady-state steps
    __shared__ long long stamp_s[4 * 128];
#define SLIDE_STAMP(ph, i, s) \
    if ((s) >= 8 && (s) < 12 && tid == 0) stamp_s[(ph) * 128 + (i)] = clock64();
#define SLIDE_STAMP_RT(ph, i, s) \
    if ((s) >= 8 && (s) < 12 && tid == 0) stamp_s[(ph) * 128 + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define SLIDE_STAMP(ph, i, s)
#define SLIDE_STAMP_RT(ph, i, s)
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = wave & 1, hr = wave >> 1;           // the wave's quarter: W half, row half
    const int co_b = blockIdx.y * 32;                  // first output channel of this slice

    // ---- weights: fragment (tap, channel tile) gathered from the 32x32x16 fragment order
    //   element (co, ci) of tap t lives at ((t * 2 + ci / 16) * NTT + co / 32) * 64 + (co % 32) + 32 * ((ci / 8) & 1)
    //   16x16x32 A fragment: lane -> co = co_b + 16 ct + (lane & 15), ci = 8 * (lane >> 4) + j
    bf16x8 wreg[54];
    {
        const int NTT = a.cout_total / 32;
        const int kb = lane >> 4;
        static_for<0, 54>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int tap = f >> 1, ct = f & 1;
            const int st = a.flip ? 26 - tap : tap;
            const int co = co_b + 16 * ct + (lane & 15);
            wreg[f] = a.w[((st * 2 + (kb >> 1)) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (kb & 1)];
        });
    }

    bf16x8 wreg2[2];
    if constexpr (HAS_X2) {
        const int NTT = a.cout_total / 32;
        const int kb = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            const int co = co_b + 16 * ct + (lane & 15);
            wreg2[ct] = a.w2[((kb >> 1) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (kb & 1)];
        }
    }
    // B-fragment layout of a 16-voxel row of the second tensor: voxel lane & 15, k-block lane >> 4
    const int xvoff = ((lane & 15) * a.ldr + (lane >> 4) * 8) * 2;

    // ---- staging: piece c = tid + 256 i of a plane is halo row c >> 2 = (tid >> 2) + 64 i, 16-byte piece tid & 3
    bf16* const sdst = lds + (tid >> 2) * PITCH + (tid & 3) * 8;
    // ---- fragment address of this lane: voxel (lane & 15) of the wave's 16-voxel W-run, k-block lane >> 4, in the
    // wave's first input row; row, kw and plane are compile-time offsets
    const bf16* bl = lds + (RH * hr * WW + 16 * hw + (lane & 15)) * PITCH + (lane >> 4) * 8;

    // fused InstanceNorm statistics: slab[workgroup][wave][n][32][2]
    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
    int cur_n = -1;             // sample of the plane whose row phase is pending
    if (HAS_STATS || HAS_BST)
        ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave, a.N, 32 * 2);
    auto stat_flush = [&]() {
        if (!(HAS_STATS || HAS_BST) || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * a.N + cur_n) * 32) * 2;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float s1 = st1[i], s2 = st2[i];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {       // lanes with equal (lane & 3) hold the same 8 channels
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lane < 4) {
                const int c = lane * 8 + i;
                dst[c * 2] = s1;
                dst[c * 2 + 1] = s2;
            }
            st1[i] = st2[i] = 0.f;
        }
    };

    f32x4 acc[2][RH][2];        // [step parity][output row][channel tile]
    bf16x8 xq[RI];              // activation fragments of one pass: the wave's input rows 0..5
    bf16x8 stg[NSTG];           // the plane in flight from HBM to LDS
    bf16x8 rq[4];               // residual rows of the plane being computed (consumed by the next step)
    bf16x8 rv;                  // patch row of the row phase
    f32x4 bt;                   // accumulator tile + bias between its micro-ops
    bf16x4 btp;                 // ... rounded
    float ev[8];                // residual variant: the row's values between convert and pack
    bf16* est = est_s + wave * (64 * EP);
    // bias of the channels this lane holds in the accumulator layout: 16 ct + 4 (lane >> 4) + i
    // (the input-gradient role with backward sums has no bias, and no registers to spare for one)
    float bias4[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int i = 0; i < 4; i++) bias4[ct][i] = (!HAS_BST && a.bias) ? a.bias[co_b + 16 * ct + 4 * (lane >> 4) + i] : 0.f;
    // row phase: lane -> voxel lane >> 2 of the 16-run, channels 8 (lane & 3)..; byte offsets against a wave-uniform
    // row pointer
    const int yvoff = ((lane >> 2) * a.ldy + (lane & 3) * 8) * 2;
    const int rvoff = ((lane >> 2) * a.ldr + (lane & 3) * 8) * 2;
    const int64_t yrow_b = (int64_t)a.W * a.ldy * 2, yplane_b = (int64_t)a.H * yrow_b;
    const int64_t rrow_b = (int64_t)a.W * a.ldr * 2, rplane_b = (int64_t)a.H * rrow_b;
    char* ycur = nullptr;       // output row 0 of this wave's quarter in the plane being computed
    char* yprev = nullptr;      // ... in the plane whose epilogue is pending
    // W % 32 == 16: the far 16-voxel half of the last column lies outside the volume - its wave computes on the zero
    // halo and neither loads residual rows nor stores nor counts (wave-uniform flags, carried like ycur / yprev)
    bool vcur = true, vprev = true;
    const char* rcur = nullptr;
    bool first = true;

    // ---- micro-operations of the epilogue of the plane in acc[PARP] -------------------------------------------------
    auto micro_b = [&](auto parp, auto idc) {            // id = 0..31: tile id >> 2 = (m, ct), quarter id & 3
        constexpr int PARP = decltype(parp)::value, id = decltype(idc)::value;
        constexpr int m = id >> 3, ct = (id >> 2) & 1, qt = id & 3;
        if constexpr (qt < 2) {                              // two bias adds
#pragma unroll
            for (int i = 2 * qt; i < 2 * qt + 2; i++) bt[i] = HAS_BST ? acc[PARP][m][ct][i] : acc[PARP][m][ct][i] + bias4[ct][i];
        } else if constexpr (qt == 2) {                      // two packed converts
            btp = __builtin_convertvector(bt, bf16x4);
        } else {
            *reinterpret_cast<bf16x4*>(est + (m * 16 + (lane & 15)) * EP + 16 * ct + 4 * (lane >> 4)) = btp;
        }
    };
    auto micro_c = [&](auto cidc) {                      // cid = 10 p + k: output row p, k = 0 read, 1..8 channel, 9 store
        constexpr int cid = decltype(cidc)::value;
        constexpr int p = cid / 10, k = cid % 10;
        if constexpr (k == 0) {
            rv = *reinterpret_cast<const bf16x8*>(est + (16 * p + (lane >> 2)) * EP + (lane & 3) * 8);
        } else if constexpr (k <= 8) {
            constexpr int i = k - 1;
            if constexpr (HAS_STATS || HAS_RES || HAS_BST) {
                const float v = (float)rv[i];                                      // the stored value of conv + bias
                if constexpr (HAS_STATS) {
                    if (vprev) {
                        st1[i] += v;
                        st2[i] = fmaf(v, v, st2[i]);
                    }
                }
                if constexpr (HAS_RES) ev[i] = v + (float)rq[p][i];
                if constexpr (HAS_BST) {
                    const float o = (float)rq[p][i];                               // the activation lrelu(xhat)
                    const bool pos = o > 0.f;
                    const float gp = v * (pos ? 1.f : a.slope);
                    if (vprev) {
                        st1[i] += gp;
                        st2[i] = fmaf(gp, o * (pos ? 1.f : a.inv_slope), st2[i]);
                    }
                }
            }
        } else {
            char* dst = yprev + p * yrow_b + yvoff;
            if (vprev) {
                if constexpr (HAS_RES) {
                    f32x8 e;
#pragma unroll
                    for (int i = 0; i < 8; i++) e[i] = ev[i];
                    *reinterpret_cast<bf16x8*>(dst) = __builtin_convertvector(e, bf16x8);
                } else {
                    *reinterpret_cast<bf16x8*>(dst) = rv;
                }
            }
        }
    };

    const int G = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (G % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += G) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;   // XCD-contiguous deal: neighbouring columns share an L2
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, h0 = th_i * TH, w0 = tw_i * TW;

        // halo pieces of this column as byte offsets inside a plane; outside the volume: beyond the buffer's range, the
        // buffer load returns zeros (the plane as a whole is switched off through the record count)
        const int plane_b = a.H * a.W * a.ldx * 2;
        const int sample_b = a.D * plane_b;
        int voff[NSTG];
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const int r = (tid >> 2) + 64 * i, zh = r / WW, zw = r - zh * WW;
            const int gh = h0 - 1 + zh, gw = w0 - 1 + zw;
            const bool okv = r < PROWS && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            voff[i] = okv ? ((gh * a.W + gw) * a.ldx + (tid & 3) * 8) * 2 : (int)0x80000000;
        }
        const bf16* xs = a.x + (int64_t)n * a.D * (plane_b / 2);
        auto load_piece = [&](int pr, auto ic) {
            constexpr int i = decltype(ic)::value;
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xs, (short)0, dok ? sample_b : 0, 0x00020000);
            stg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], dok ? d * plane_b : 0, 0));
        };
        auto store_piece = [&](int slot, auto ic) {
            constexpr int i = decltype(ic)::value;
            // the last piece covers rows 320..383 of 340: threads 0..79 only
            if (i * 64 + 64 <= PROWS || tid < (PROWS - i * 64) * 4)
                *reinterpret_cast<bf16x8*>(sdst + slot * PLANE + i * 64 * PITCH) = stg[i];
        };
        auto load_plane = [&](int pr) { static_for<0, NSTG>([&](auto ic) { load_piece(pr, ic); }); };
        auto store_plane = [&](int slot) { static_for<0, NSTG>([&](auto ic) { store_piece(slot, ic); }); };

        __syncthreads();   // the previous column has left the ring
        load_plane(0); store_plane(0);
        load_plane(1); store_plane(1);
        load_plane(2); store_plane(2);
        load_plane(3);
        __syncthreads();

        // this wave's output rows in plane d0
        ycur = reinterpret_cast<char*>(a.y + (int64_t)blockIdx.y * a.yslice) +
               ((((int64_t)n * a.D + d0) * a.H + h0 + RH * hr) * a.W + w0 + 16 * hw) * (int64_t)a.ldy * 2;
        if constexpr (HAS_X2)
            rcur = reinterpret_cast<const char*>(a.res) +
                   ((((int64_t)n * a.D + d0) * a.H + h0 + RH * hr) * a.W + w0 + 16 * hw) * (int64_t)a.ldr * 2;
        if constexpr (HAS_RES || HAS_BST)
            rcur = reinterpret_cast<const char*>(a.res + co_b) +
                   ((((int64_t)n * a.D + d0) * a.H + h0 + RH * hr) * a.W + w0 + 16 * hw) * (int64_t)a.ldr * 2;
        const bool vnext = w0 + 16 * hw < a.W;
        if (first) {
            yprev = ycur;
            vprev = vnext;
        }
        vcur = vnext;

        // ---- activation fragment of pass q = (kd, kw) of the step with ring phase PHN: the wave's input row r
        auto frag_row = [&](auto phn, auto qc, auto rc) {
            constexpr int PHN = decltype(phn)::value, q = decltype(qc)::value, r = decltype(rc)::value;
            constexpr int kd = q / 3, kw = q % 3;
            xq[r] = *reinterpret_cast<const bf16x8*>(bl + ((PHN + kd) & 3) * PLANE + (r * WW + kw) * PITCH);
        };

        auto step = [&](auto phc, int s) {
            constexpr int PH = decltype(phc)::value;
            constexpr int PAR = PH & 1;
            // wait states in front of the inline-asm MFMA block (see conv_slide64.hip)
            asm volatile("s_nop 7");
            SLIDE_STAMP_RT(PH, 120, s)
            static_for<0, NP>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int kd = q / 3, kw = q % 3;
#ifndef RU3D_SLIDE_STAMP_PAIRS
                SLIDE_STAMP(PH, q, s)
#endif
                // Plane s+2, stored during the previous step, is complete (read from pass 6 on), and every wave is past
                // kd = 0 of the previous step: the slot plane s+3 goes to is free.  LDS-only: a __syncthreads() would
                // also drain vmcnt, i.e. wait for the epilogue stores just issued
                if constexpr (q == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                static_for<0, NJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int r = JR[j], kh = JK[j], m = r - kh;
                    constexpr int f = ((kd * 3 + kh) * 3 + kw) * 2;
#ifdef RU3D_SLIDE_STAMP_PAIRS
                    SLIDE_STAMP(PH, q * NJ + j, s)
#endif
                    mfma16<(q == 0 && kh == 0)>(acc[PAR][m][0], wreg[f], xq[r]);
                    mfma16<(q == 0 && kh == 0)>(acc[PAR][m][1], wreg[f + 1], xq[r]);
                    // row r is free behind its last pair: refill it for the next pass (behind the column's last step
                    // this reads a stale slot - the next column's prologue loads its own)
                    if constexpr (last_of_row(j)) {
                        if constexpr (q + 1 < NP) frag_row(std::integral_constant<int, PH>{}, std::integral_constant<int, q + 1>{}, std::integral_constant<int, r>{});
                        else frag_row(std::integral_constant<int, (PH + 1) & 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, r>{});
                    }
                    // the micro-operation pinned behind this pair
                    constexpr bool staging = q >= 3 && (j == 3 || j == 6);
                    if constexpr (staging) {
                        if constexpr (j == 3) store_piece((PH + 3) & 3, std::integral_constant<int, (q >= 3 ? q - 3 : 0)>{});
                        else load_piece(s + 4, std::integral_constant<int, (q >= 3 ? q - 3 : 0)>{});
                    } else {
                        // u-th pair without staging: 18 micro-ops per output row (8 tile quarters, 10 row phase)
                        constexpr int u = q < 3 ? q * NJ + j : 36 + (q - 3) * 10 + j - (j > 3) - (j > 6);
                        if constexpr (u < 72) {
                            constexpr int p = u / 18, w = u % 18;
                            if constexpr (w < 8) micro_b(std::integral_constant<int, PAR ^ 1>{}, std::integral_constant<int, 8 * p + w>{});
                            else micro_c(std::integral_constant<int, 10 * p + w - 8>{});
                        }
                        if constexpr ((HAS_RES || HAS_BST) && q == NP - 1 && j % 3 == 1) {
                            if (vcur) rq[(j - 1) / 3] = *reinterpret_cast<const bf16x8*>(rcur + ((j - 1) / 3) * rrow_b + rvoff);
                        }
                        if constexpr (HAS_X2 && q == 0 && j % 3 == 1) {
                            if (vcur) rq[(j - 1) / 3] = *reinterpret_cast<const bf16x8*>(rcur + ((j - 1) / 3) * rrow_b + xvoff);
                        }
                    }
                    if constexpr (HAS_X2 && q == NP - 1 && j == NJ - 1) {
                        // the partner's tap: output row m <- its own row of the second tensor
                        static_for<0, RH>([&](auto mc) {
                            constexpr int mm = decltype(mc)::value;
                            mfma16<false>(acc[PAR][mm][0], wreg2[0], rq[mm]);
                            mfma16<false>(acc[PAR][mm][1], wreg2[1], rq[mm]);
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            SLIDE_STAMP(PH, 127, s)
            SLIDE_STAMP_RT(PH, 121, s)
            yprev = ycur;
            vprev = vcur;
            ycur += yplane_b;
            if constexpr (HAS_RES || HAS_BST || HAS_X2) rcur += rplane_b;
        };

        // fragments of the first pass of step 0
        static_for<0, RI>([&](auto rc) { frag_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rc); });
        for (int s4 = 0; s4 < a.DL; s4 += 4) {
            step(std::integral_constant<int, 0>{}, s4);
            if (s4 == 0) {
                // step 0 has finished the previous column's last plane (or, the first time, garbage)
                if (first) {
#pragma unroll
                    for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
                    first = false;
                } else if ((HAS_STATS || HAS_BST) && n != cur_n) {
                    stat_flush();
                }
                cur_n = n;
            }
            step(std::integral_constant<int, 1>{}, s4 + 1);
            step(std::integral_constant<int, 2>{}, s4 + 2);
            step(std::integral_constant<int, 3>{}, s4 + 3);
        }
    }
    // the workgroup's last plane (parity 1: DL is a multiple of 4) has no next step to hide behind
    if (!first) {
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[1][3][0]), "+v"(acc[1][3][1]));
        static_for<0, 32>([&](auto idc) { micro_b(std::integral_constant<int, 1>{}, idc); });
        static_for<0, 40>([&](auto cidc) { micro_c(cidc); });
    }
    stat_flush();
#ifdef RU3D_SLIDE_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && a.stamps) { a.stamps[tid] = stamp_s[tid]; a.stamps[tid + 256] = stamp_s[tid + 256]; }
#endif
}
}  // namespace

#ifdef RU3D_SLIDE_STAMPS
static long long* g_slide_stamps = nullptr;
extern "C" void ru3d_debug_slide_stamps(long long* dev_buf) { g_slide_stamps = dev_buf; }
#endif

// Work decomposition: units = N x dsplit x (H/8) x (W/32) columns of DL = D/dsplit planes (+2 halo planes each),
// times Cout/32 output-channel slices (grid.y).  dsplit is the divisor of D (DL a multiple of 4) with the shortest
// makespan on 256 CUs.
bool slide_conv_plan(int N, int D, int H, int W, int Cin, int Cout, SlidePlan* out) {
    static const int mode = getenv("RU3D_CONV_SLIDE") ? atoi(getenv("RU3D_CONV_SLIDE")) : 1;
    if (mode == 0 || Cin != 32 || (Cout != 32 && Cout != 64) || (H % TH) || (W % 16) || D < 4) return false;
    // the staging loads address a sample with 30-bit element offsets; the input may sit in a buffer of twice its channels
    // (the decoder's concat): shapes that could exceed that go to the other kernels consistently (launch, slab, workspace)
    if ((int64_t)D * H * W * Cin * 2 >= (1ll << 30)) return false;
    const int ny = Cout / 32;
    const int64_t cols = (int64_t)N * (H / TH) * ((W + TW - 1) / TW);      // W % 32 == 16: the last column's far half idles
    int64_t best_cost = -1;
    int best = 0;
    for (int ds = 1; ds <= D / 4; ds++) {
        if (D % ds) continue;
        const int dl = D / ds;
        if (dl % 4) continue;
        const int64_t units = cols * ds;
        if (units * ny > 0x7fffffff) break;
        // grid.x = min(units, 256 / ny rounded to 8) persistent workgroups per slice
        int64_t gx = ru3d_get_cu_budget() / ny;
        if (gx > units) gx = units;
        const int64_t cost = ((units + gx - 1) / gx) * (dl + 3);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = ds;
        }
    }
    if (!best) return false;
    const int64_t units = cols * best;
    // worth it only when the 256 CUs are reasonably filled
    const double ideal = (double)cols * ny * D / (double)ru3d_get_cu_budget();
    if (units * ny < 128 || (double)best_cost > 1.6 * ideal + 8) return false;
    out->dsplit = best;
    out->DL = D / best;
    out->tiles_h = H / TH;
    out->tiles_w = (W + TW - 1) / TW;
    out->units = (int)units;
    int g = units < ru3d_get_cu_budget() / ny ? (int)units : ru3d_get_cu_budget() / ny;
    if ((units % 8) == 0 && g >= 8) g = (g / 8) * 8;
    out->grid = g;
    out->ny = ny;
    return true;
}

int conv_slide_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                      float* stat_slab, hipStream_t st, const void* bst_act, int bst_ld, float slope, const void* x2,
                      int ldx2, const void* w2) {
    SlidePlan p;
    if (!slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &p))
        return ru3d_fail(-1, "conv_slide: shape not supported");
    // the staging loads address a sample through a buffer descriptor with 32-bit byte offsets, the top bit marking
    // "outside the volume"
    if ((int64_t)g.Do * g.Ho * g.Wo * g.ldx >= (1ll << 30)) return ru3d_fail(-1, "conv_slide: sample too large");
    // residual + statistics together is not a combination any entry point produces (ru3d_conv3d_fwd_in has no residual)
    if (res && stat_slab) return ru3d_fail(-1, "conv_slide: residual and fused statistics cannot be combined");
    if (bst_act && (res || !stat_slab)) return ru3d_fail(-1, "conv_slide: the backward sums need a slab and no residual");
    if (x2 && (res || stat_slab || bst_act || !w2 || (ldx2 % 8) || (((uintptr_t)x2) | ((uintptr_t)w2)) % 16))
        return ru3d_fail(-1, "conv_slide: the 1x1 partner excludes residual / statistics and needs aligned operands");
    Slide32Args a;
    a.w2 = (const bf16x8*)w2;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.bias = bias;
    a.res = (const bf16*)res;
    a.y = (bf16*)y;
    a.stat_slab = stat_slab;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldr;
    a.yslice = 32;
    if (g.y_cseg) {      // planar concat gradient: slice s of the output is plane s
        if (g.y_cseg != 32 || res || stat_slab) return ru3d_fail(-1, "conv_slide: split output needs 32-channel slices, no residual, no statistics");
        a.yslice = g.y_segstride;
    }
    if (g.x_cseg) return ru3d_fail(-1, "conv_slide: split input not supported");
    a.flip = g.flip;
    a.cout_total = g.Cout;
    a.slope = slope;
    a.inv_slope = slope != 0.f ? 1.f / slope : 0.f;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
#ifdef RU3D_SLIDE_STAMPS
    a.stamps = g_slide_stamps;
#endif
    const dim3 grid(p.grid, p.ny), block(256);
    if (x2) {
        a.res = (const bf16*)x2;
        a.ldr = ldx2;
        hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, false, false, true>), grid, block, 0, st, a);
    } else if (bst_act) {
        a.res = (const bf16*)bst_act;
        a.ldr = bst_ld;
        hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, false, true>), grid, block, 0, st, a);
    } else if (res) hipLaunchKernelGGL((conv3_s1_slide32_kernel<true, false>), grid, block, 0, st, a);
    else if (stat_slab) hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, false>), grid, block, 0, st, a);
    return ru3d_check_launch("conv3_s1_slide32");
}
}  // namespace RU3D_NS
