// 3x3x3 stride-1 conv, 32 -> 32 channels, bf16 MFMA: the full-resolution level's conv (reference network.py:391-403
// conv1/conv2 of the level-0 ResBlocks, forward and input gradient).
//
// The halo-tile kernels of conv_mfma.hip re-read every input voxel ~3.2x (2x4x32 tile, 1-voxel halo on six faces)
// and stream the 55 KB weight once per tile; both arrive through the same ~10 B/clk/CU path and that, not the
// MFMA pipe, set their time.  This kernel removes both re-reads and most of the LDS traffic:
//   * the whole 32x32x27 weight lives in REGISTERS: 54 fragments x 4 VGPRs per lane, loaded once per persistent
//     workgroup (one wave per SIMD, 512 registers) - no weight bytes move after the prologue;
//   * a workgroup owns an (8 x 32) column in (H, W) and SLIDES along D: a ring of 4 input planes (10 x 34 halo
//     rows x 64 B) stays in LDS, each step loads ONE new plane (21.8 KB) for 256 output voxels - 1.33x
//     amplification instead of 3.2x - while the step's 432 MFMAs run (global -> registers at the top of the step,
//     registers -> LDS behind the step's barrier);
//   * a wave computes two output rows (h, h+1); their taps kh = 0..2 touch only 4 distinct input rows, so a
//     (kd, kw, k-step) group issues 4 activation-fragment reads for 6 MFMAs (0.67 KB of LDS reads per MFMA,
//     a third of the LDS array's rate at full MFMA speed).
// LDS rows are 80 bytes (64 B of channels + 16 B pad): conflict-free for the ds_read_b128 of a 32-voxel W-run, and
// every fragment address is one per-lane base plus a compile-time offset (no address registers per tap).
// The step is software-pipelined across planes: the epilogue of plane s (bf16 conversion, LDS transpose, statistics,
// stores) is issued between the MFMAs of plane s+1 (two accumulator sets), the fragment ring runs on into the next
// step, and the single barrier of a step sits two groups before the kd = 2 taps - only they read the newest
// plane, so its LDS store (behind the previous step's barrier) has two thirds of a step to land.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {

namespace {
constexpr int TH = 8, TW = 32, HH = TH + 2, WW = TW + 2;
constexpr int PROWS = HH * WW;        // 340 halo rows per plane
constexpr int PITCH = 40;             // bf16 elements per staged row: 64 B of channels + 16 B pad (conflict-free
                                      // ds_read_b128 over a 32-voxel W-run, and every fragment address is affine)
constexpr int PLANE = PROWS * PITCH;  // bf16 elements per plane
constexpr int RING = 4;
constexpr int WAVE_ROWS = PROWS / 4;  // 85 rows of a plane are staged by each wave
constexpr int EST_PITCH = 36;         // floats per epilogue-patch row (32 channels + 4 pad = 144 bytes)
constexpr int XG = 1;                 // activation-fragment register sets (rows are refilled one by one, see frag_row)
constexpr int NG = 18;                // (kd, kw, k-step) groups per step, 6 MFMAs each
constexpr int NSTG = 6;               // 16-byte pieces staged per thread and plane (85 rows x 4 pieces / 64 lanes)
static_assert(PROWS % 4 == 0, "plane rows split evenly over 4 waves");
static_assert(RING * PLANE * 2 + 4 * 64 * EST_PITCH * 4 <= 160 * 1024, "LDS budget");

struct SlideArgs {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    bf16* y;
    float* stat_slab;
    int N, D, H, W;
    int ldx, ldy, ldr;
    int flip;
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

// MFMA with the operand register classes pinned: accumulators and (most) weight fragments live in the AGPR half of
// the 512-register file, activation fragments arrive from LDS in VGPRs.  Left to itself the allocator parks the
// weights in AGPRs too but copies each fragment back to VGPRs before use (4 v_accvgpr_read per MFMA).
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <bool WA, bool ZERO>
__device__ __forceinline__ void mfma_w(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    const i32x4 wi = __builtin_bit_cast(i32x4, w), xi = __builtin_bit_cast(i32x4, x);
    if constexpr (ZERO) {
        if constexpr (WA) asm volatile(RU3D_MFMA_ASM " %0, %1, %2, 0" : "=a"(acc) : "a"(wi), "v"(xi));
        else asm volatile(RU3D_MFMA_ASM " %0, %1, %2, 0" : "=a"(acc) : "v"(wi), "v"(xi));
    } else {
        if constexpr (WA) asm volatile(RU3D_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "a"(wi), "v"(xi));
        else asm volatile(RU3D_MFMA_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(wi), "v"(xi));
    }
}
constexpr int WA_FRAGS = 48;          // weight fragments held in AGPRs (48 x 4 + 64 accumulator registers = 256)

__device__ __forceinline__ int piece_off(int row, int piece) { return row * PITCH + piece * 8; }

// HAS_RES / HAS_STATS are compile-time: a residual load that is only conditionally issued still makes the compiler
// guard every reuse of its destination registers with a vmcnt wait, and vmcnt retires in order - in the plain
// variant those waits landed on the plane loads that had just been issued (an HBM latency per step).
template <bool HAS_RES, bool HAS_STATS>
__global__ __launch_bounds__(256, 1) void conv3_s1_slide32_kernel(SlideArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[RING * PLANE];
    __shared__ __attribute__((aligned(16))) float est_s[4 * 64 * EST_PITCH];   // fp32 epilogue patches, one per wave
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef RU3D_SLIDE_STAMPS
    __shared__ long long stamp_s[4 * 24];
#define SLIDE_STAMP(ph, g, s) \
    if ((s) >= 8 && (s) < 12 && tid == 0) stamp_s[(ph) * 24 + (g)] = clock64();
#else
#define SLIDE_STAMP(ph, g, s)
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- the weight: fragment (tap, ks) of this lane, all 54 of them (the input-gradient role reads the taps
    // mirrored: applied here, not in the loop)
    bf16x8 wreg[54];
    static_for<0, 54>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        constexpr int tap = f >> 1, ks = f & 1;
        const int st = a.flip ? 26 - tap : tap;
        wreg[f] = a.w[((st * 2 + ks) * gridDim.y + blockIdx.y) * 64 + lane];
    });

    // ---- staging constants: this thread's pieces of the wave's 85 plane rows
    int srel[NSTG], sdst[NSTG], szh[NSTG], szw[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; i++) {
        const int cw = lane + 64 * i;
        const bool v = cw < WAVE_ROWS * 4;
        const int r = wave * WAVE_ROWS + (v ? (cw >> 2) : 0), part = cw & 3;
        szh[i] = v ? r / WW : -100000;      // invalid pieces fail every bounds test
        szw[i] = r % WW;
        srel[i] = ((r / WW) * a.W + (r % WW)) * a.ldx + part * 8;
        sdst[i] = piece_off(r, part);
    }
    // ---- fragment address of this lane: input row 2*wave (+ r = 0..3: the rows that serve output rows 2*wave and
    // 2*wave + 1), voxel (lane & 31) of the W-run (+ kw), k-half lane >> 5 (+ 2 per k-step); everything in
    // parentheses is a compile-time offset
    const bf16* bl = lds + piece_off((2 * wave) * WW + (lane & 31), lane >> 5);

    // fused InstanceNorm statistics (same slab layout as the producer/consumer kernel: [workgroup][wave][n][32][2])
    float st1[8], st2[8];
    int cur_n = -1;
    auto stat_flush = [&]() {
        if (!HAS_STATS || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * a.N + cur_n) * 32) * 2;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float s1 = st1[i], s2 = st2[i];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lane < 4) {
                const int c = lane * 8 + i;
                dst[c * 2] = s1;
                dst[c * 2 + 1] = s2;
            }
        }
    };

    f32x16 acc[2][2];   // [step parity][output row of the wave]
    bf16x8 xq[XG][4];   // activation fragments of XG groups: input rows 2*wave .. 2*wave + 3
    float* est = est_s + wave * (64 * EST_PITCH);
    // bias of the 8 channels this lane stores in the row phase of the epilogue
    float bias8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) bias8[i] = a.bias ? a.bias[blockIdx.y * 32 + (lane & 3) * 8 + i] : 0.f;
    // output channels [32 * blockIdx.y, +32): Cout = 64 runs as two independent 32-channel slices of the grid
    bf16* const ybase = a.y + blockIdx.y * 32;
    const bf16* const rbase = a.res + blockIdx.y * 32;

    const int G = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (G % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += G) {
        // XCD-contiguous deal: hardware workgroup ids round-robin over the 8 XCDs; neighbours in (h, w) share halo rows
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;
        const int tw_i = u % a.tiles_w;
        u /= a.tiles_w;
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, h0 = th_i * TH, w0 = tw_i * TW;

        // Plane loads are issued unconditionally (out-of-range pieces read the first bytes of the plane) and zeroed
        // when they are written to LDS a step later: a select on the loaded value at issue time would make the wave
        // wait for HBM right there.
        bool ok[NSTG];
        int soff[NSTG];
        const int base_hw = ((h0 - 1) * a.W + (w0 - 1)) * a.ldx;
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const int gh = h0 - 1 + szh[i], gw = w0 - 1 + szw[i];
            ok[i] = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            soff[i] = ok[i] ? base_hw + srel[i] : 0;
        }
        const int64_t plane_stride = (int64_t)a.H * a.W * a.ldx;

        auto load_plane = [&](int pr, bf16x8 (&stg)[NSTG]) {
            int d = d0 - 1 + pr;
            d = d < 0 ? 0 : (d >= a.D ? a.D - 1 : d);
            const bf16* src = a.x + ((int64_t)n * a.D + d) * plane_stride;
#pragma unroll
            for (int i = 0; i < NSTG; i++) stg[i] = *reinterpret_cast<const bf16x8*>(src + soff[i]);
        };
        auto store_plane = [&](int pr, int slot, const bf16x8 (&stg)[NSTG]) {
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
#pragma unroll
            for (int i = 0; i < NSTG; i++) {
                const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if (szh[i] >= 0) *reinterpret_cast<bf16x8*>(lds + slot * PLANE + sdst[i]) = (dok && ok[i]) ? stg[i] : z8;
            }
        };

        // one 16-byte piece at a time (steady state): the plane staging is spread over the MFMA groups behind the
        // step's barrier - as one block it kept the matrix pipe idle for ~1000 of a step's ~6500 cycles
        auto load_piece = [&](int pr, auto ic, bf16x8 (&stg)[NSTG]) {
            constexpr int i = decltype(ic)::value;
            int d = d0 - 1 + pr;
            d = d < 0 ? 0 : (d >= a.D ? a.D - 1 : d);
            const bf16* src = a.x + ((int64_t)n * a.D + d) * plane_stride;
            stg[i] = *reinterpret_cast<const bf16x8*>(src + soff[i]);
        };
        auto store_piece = [&](int pr, int slot, auto ic, const bf16x8 (&stg)[NSTG]) {
            constexpr int i = decltype(ic)::value;
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if (szh[i] >= 0) *reinterpret_cast<bf16x8*>(lds + slot * PLANE + sdst[i]) = (dok && ok[i]) ? stg[i] : z8;
        };

        __syncthreads();   // the previous unit has left the ring
        bf16x8 stg[NSTG];   // the plane in flight: loaded behind one step's barrier, stored behind the next one's
        load_plane(0, stg);
        store_plane(0, 0, stg);
        load_plane(1, stg);
        store_plane(1, 1, stg);
        load_plane(2, stg);
        store_plane(2, 2, stg);
        load_plane(3, stg);
        __syncthreads();

        if (HAS_STATS && n != cur_n) {
            stat_flush();
            cur_n = n;
#pragma unroll
            for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
        }

        // ---- activation fragments, continuous across the steps of a unit: group `g` = (kd, kw, ks) of the step with
        // phase PHN needs input rows 2*wave .. 2*wave + 3.  ONE register set: row r of the next group is fetched right
        // behind the last MFMA of the current group that reads row r (rows 0..3 fall free after MFMAs 0, 2, 4, 5), so
        // every LDS read has four to five MFMAs (128-160 cycles) to land - issued in a block behind the sixth MFMA they
        // left the matrix pipe idle for an LDS latency in every group.
        auto frag_row = [&](auto phn, auto gc, auto rc) {
            constexpr int PHN = decltype(phn)::value, g = decltype(gc)::value, r = decltype(rc)::value;
            constexpr int kd = g / 6, kw = (g % 6) >> 1, ks = g & 1;
            xq[0][r] = *reinterpret_cast<const bf16x8*>(bl + ((PHN + kd) & 3) * PLANE + (r * WW + kw) * PITCH + ks * 16);
        };
        auto frag_fetch = [&](auto phn, auto gc) {
            static_for<0, 4>([&](auto rc) { frag_row(phn, gc, rc); });
        };
        // ---- epilogue pieces of a finished step (accumulators `ac`, output plane d0 + sp): accumulator layout
        // (lane = voxel, 4 couts per 8 bytes) -> wave-private LDS patch -> 16-byte stores that cover whole
        // 64-byte channel rows; bias before, residual after the transpose.  The pieces are issued between the
        // MFMAs of the NEXT step, so the matrix pipe does not idle while a plane is written out.
        auto epi_write = [&](const f32x16 (&ac)[2], int m, int q) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = ac[m][q * 4 + i];
            *reinterpret_cast<f32x4*>(est + (m * 32 + (lane & 31)) * EST_PITCH + 8 * q + 4 * (lane >> 5)) = v;
        };
        auto epi_vox = [&](int sp, int r) {
            const int row = (lane >> 2) + 16 * r;
            return (((int64_t)n * a.D + d0 + sp) * a.H + h0 + 2 * wave + (row >> 5)) * (int64_t)a.W + w0 + (row & 31);
        };
        // row phase in two halves one group apart: the LDS read is in flight while the other group's MFMAs run
        auto epi_row_load = [&](int r, f32x4 (&rv)[2]) {
            const float* src = est + ((lane >> 2) + 16 * r) * EST_PITCH + (lane & 3) * 8;
            rv[0] = *reinterpret_cast<const f32x4*>(src);
            rv[1] = *reinterpret_cast<const f32x4*>(src + 4);
        };
        auto epi_row_finish = [&](int sp, int r, const f32x4 (&rv)[2], const bf16x8& rres) {
            const int part = lane & 3;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)(bf16)(rv[i >> 2][i & 3] + bias8[i]);   // the stored value
            if constexpr (HAS_STATS) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    st1[i] += v[i];
                    st2[i] = fmaf(v[i], v[i], st2[i]);
                }
            }
            if constexpr (HAS_RES) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] += (float)rres[i];
            }
            store_vec<bf16, 8>(ybase + epi_vox(sp, r) * a.ldy + part * 8, v);
        };

        auto step = [&](auto phc, int s) {
            constexpr int PH = decltype(phc)::value;
            constexpr int PAR = PH & 1;
            const bool has_prev = s > 0, last = s == a.DL - 1;
            // residual rows of the previous plane: used ~6 groups in (vmcnt retires in order; the only older loads
            // are the plane loads of the previous step, two thirds of a step old by then)
            f32x4 rv[2];
            bf16x8 rq[4];
            if constexpr (HAS_RES) {
                if (has_prev) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        rq[r] = *reinterpret_cast<const bf16x8*>(rbase + epi_vox(s - 1, r) * a.ldr + (lane & 3) * 8);
                }
            }
            // The MFMAs below are inline asm: the compiler's hazard recognizer does not see that they read AGPRs / VGPRs.
            // Anything it may have written just before (a register restored with v_accvgpr_write, a moved fragment) gets
            // its wait states here, once per 108 MFMAs; inside the loop every operand comes from a counted LDS read or
            // from registers written a step ago.
            asm volatile("s_nop 7\n\ts_nop 7");
            SLIDE_STAMP(PH, 0, s)
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int kd = g / 6, kw = (g % 6) >> 1, ks = g & 1;
                if constexpr (g == 11) {
                    // the kd = 2 groups (12..17) read the plane that was stored behind the PREVIOUS step's barrier, and
                    // their fragments are fetched during group 11; every wave is past the kd = 0 groups of this step,
                    // so the slot of plane s-1 is free
                    SLIDE_STAMP(PH, 19, s)
                    __syncthreads();
                    SLIDE_STAMP(PH, 20, s)
                    SLIDE_STAMP(PH, 21, s)
                }
                // output row m, tap row kh reads input row m + kh; the two accumulators alternate
                static_for<0, 6>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    constexpr int m = j & 1, kh = j >> 1;
                    constexpr int f = ((kd * 3 + kh) * 3 + kw) * 2 + ks;
                    mfma_w<(f < WA_FRAGS), (g == 0 && kh == 0)>(acc[PAR][m], wreg[f], xq[0][m + kh]);
                    // plane staging, one piece per group behind the barrier of group 11: piece i of plane s+3 goes to LDS
                    // in group 11 + i (its slot, that of plane s-1, is free behind the barrier; first read in the next
                    // step's kd = 2 groups, behind that step's barrier), and piece i of plane s+4 is loaded into the
                    // freed registers one group later.  Unconditional (planes beyond the unit are clamped / land in a
                    // slot nobody reads again): a conditional store hides from the compiler that the loads were waited for
                    if constexpr (j == 1 && g >= 11 && g < 11 + NSTG)
                        store_piece(s + 3, (PH + 3) & 3, std::integral_constant<int, g - 11>{}, stg);
                    if constexpr (j == 3 && g >= 12 && g < 12 + NSTG)
                        load_piece(s + 4, std::integral_constant<int, g - 12>{}, stg);
                    constexpr int row = j == 0 ? 0 : (j == 2 ? 1 : (j == 4 ? 2 : (j == 5 ? 3 : -1)));
                    if constexpr (row >= 0) {
                        if constexpr (g + 1 < NG) {
                            frag_row(std::integral_constant<int, PH>{}, std::integral_constant<int, g + 1>{},
                                     std::integral_constant<int, row>{});
                        } else {
                            if (!last)
                                frag_row(std::integral_constant<int, (PH + 1) & 3>{}, std::integral_constant<int, 0>{},
                                         std::integral_constant<int, row>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                // epilogue of the previous plane, spread over the groups: 8 patch writes, then 4 rows (load | finish)
                if constexpr (g <= 7) {
                    if (has_prev) epi_write(acc[PAR ^ 1], g >> 2, g & 3);
                }
                if constexpr (g >= 8 && g <= 15 && (g & 1) == 0) {
                    if (has_prev) epi_row_load((g - 8) >> 1, rv);
                }
                if constexpr (g >= 8 && g <= 15 && (g & 1) == 1) {
                    if (has_prev) epi_row_finish(s - 1, (g - 8) >> 1, rv, rq[(g - 8) >> 1]);
                }
                if constexpr (g == 16) {
                    SLIDE_STAMP(PH, 22, s)
                }
                SLIDE_STAMP(PH, g + 1, s)
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        // fragments of the first group of step 0
        frag_fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        for (int s4 = 0; s4 < a.DL; s4 += 4) {
            step(std::integral_constant<int, 0>{}, s4);
            step(std::integral_constant<int, 1>{}, s4 + 1);
            step(std::integral_constant<int, 2>{}, s4 + 2);
            step(std::integral_constant<int, 3>{}, s4 + 3);
        }
        // the last plane of the unit (phase 3, parity 1) has no next step to hide behind
        {
            bf16x8 rq[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                rq[r] = HAS_RES ? *reinterpret_cast<const bf16x8*>(rbase + epi_vox(a.DL - 1, r) * a.ldr + (lane & 3) * 8) : z8;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) epi_write(acc[1], j >> 2, j & 3);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                f32x4 rv[2];
                epi_row_load(r, rv);
                epi_row_finish(a.DL - 1, r, rv, rq[r]);
            }
        }
    }
    stat_flush();
#ifdef RU3D_SLIDE_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && tid < 96 && a.stamps) a.stamps[tid] = stamp_s[tid];
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
    if (mode == 0 || Cin != 32 || (Cout != 32 && Cout != 64) || (H % TH) || (W % TW) || D < 4) return false;
    const int ny = Cout / 32;
    const int64_t cols = (int64_t)N * (H / TH) * (W / TW);
    int64_t best_cost = -1;
    int best = 0;
    for (int ds = 1; ds <= D / 4; ds++) {
        if (D % ds) continue;
        const int dl = D / ds;
        if (dl % 4) continue;
        const int64_t units = cols * ds;
        if (units * ny > 0x7fffffff) break;
        // grid.x = min(units, 256 / ny rounded to 8) persistent workgroups per slice
        int64_t gx = 256 / ny;
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
    const double ideal = (double)cols * ny * D / 256.0;
    if (units * ny < 128 || (double)best_cost > 1.6 * ideal + 8) return false;
    out->dsplit = best;
    out->DL = D / best;
    out->tiles_h = H / TH;
    out->tiles_w = W / TW;
    out->units = (int)units;
    int g = units < 256 / ny ? (int)units : 256 / ny;
    if ((units % 8) == 0 && g >= 8) g = (g / 8) * 8;
    out->grid = g;
    out->ny = ny;
    return true;
}

int conv_slide_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                      float* stat_slab, hipStream_t st) {
    SlidePlan p;
    if (!slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &p))
        return ru3d_fail(-1, "conv_slide: shape not supported");
    if ((int64_t)g.Do * g.Ho * g.Wo * g.ldx >= (1ll << 31)) return ru3d_fail(-1, "conv_slide: sample too large");
    SlideArgs a;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.bias = bias;
    a.res = (const bf16*)res;
    a.y = (bf16*)y;
    a.stat_slab = stat_slab;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldr;
    a.flip = g.flip;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
#ifdef RU3D_SLIDE_STAMPS
    a.stamps = g_slide_stamps;
#endif
    // residual + statistics together is not a combination any entry point produces (ru3d_conv3d_fwd_in has no
    // residual), and its instantiation would not fit the register file
    if (res && stat_slab) return ru3d_fail(-1, "conv_slide: residual and fused statistics cannot be combined");
    if (res) hipLaunchKernelGGL((conv3_s1_slide32_kernel<true, false>), dim3(p.grid, p.ny), dim3(256), 0, st, a);
    else if (stat_slab) hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, true>), dim3(p.grid, p.ny), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3_s1_slide32_kernel<false, false>), dim3(p.grid, p.ny), dim3(256), 0, st, a);
    return ru3d_check_launch("conv3_s1_slide32");
}

}  // namespace RU3D_NS
