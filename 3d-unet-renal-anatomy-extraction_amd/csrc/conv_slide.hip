// 3x3x3 stride-1 conv, 32 -> 32 channels, bf16 MFMA: the full-resolution level's conv (reference network.py:391-403
// conv1/conv2 of the level-0 ResBlocks, forward and input gradient).
//
// The halo-tile kernels of conv_mfma.hip re-read every input voxel ~3.2x (2x4x32 tile, 1-voxel halo on six faces)
// and stream the 55 KB weight once per tile; both arrive through the same ~10 B/clk/CU path and that, not the
// MFMA pipe, set their time.  This kernel removes both re-reads:
//   * the whole 32x32x27 weight lives in LDS (fragment order, 55 KB) for the life of the persistent workgroup;
//   * a workgroup owns an (8 x 32) column in (H, W) and SLIDES along D: a ring of 4 input planes (10 x 34 halo
//     rows x 64 B) stays in LDS, each step loads ONE new plane (21.8 KB) for 256 output voxels - 1.33x
//     amplification instead of 3.2x, and it is loaded while the step's 432 MFMAs run (global -> registers at
//     the top of the step, registers -> LDS at the bottom, one barrier per step).
// LDS rows are 64 bytes with the 16-byte piece index XOR-ed by ((row >> 2) & 3): any 16 rows that are distinct
// mod 16 - which is what every ds_read_b128 lane group of a 32-voxel W-run touches - land in 16 distinct bank
// quads, for every tap shift.
// Per step and wave: 54 (tap, k-step) iterations x [1 weight fragment + 2 activation fragments from LDS, 2 MFMAs].
// The step is software-pipelined across planes: the epilogue of plane s (bf16 conversion, LDS transpose, statistics,
// stores) is issued between the MFMAs of plane s+1 (two accumulator sets), the fragment ring runs on into the next
// step, and the single barrier of a step sits before iteration 33 - only the kd = 2 taps (iterations 36..53) read
// the newest plane, so its LDS store (behind the previous step's barrier) has a third of a step to land.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace {
constexpr int TH = 8, TW = 32, HH = TH + 2, WW = TW + 2;
constexpr int PROWS = HH * WW;        // 340 halo rows per plane
constexpr int PLANE = PROWS * 32;     // bf16 elements per plane
constexpr int RING = 4;
constexpr int WTS = 54 * 64 * 8;      // 27 taps x 2 k-steps x 64 lanes x 8 bf16
constexpr int WAVE_ROWS = PROWS / 4;  // 85 rows of a plane are staged by each wave
constexpr int EST_PITCH = 40;         // bf16 elements per epilogue-patch row (80 bytes)
constexpr int XD = 3;                 // LDS fragment prefetch depth (iterations)
constexpr int NSTG = 6;               // 16-byte pieces staged per thread and plane (85 rows x 4 pieces / 64 lanes)
static_assert(PROWS % 4 == 0, "plane rows split evenly over 4 waves");
static_assert((RING * PLANE + WTS + 4 * 64 * EST_PITCH) * 2 <= 160 * 1024, "LDS budget");

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
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int piece_off(int row, int piece) { return row * 32 + ((piece ^ ((row >> 2) & 3)) << 3); }

__global__ __launch_bounds__(256, 1) void conv3_s1_slide32_kernel(SlideArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[RING * PLANE + WTS + 4 * 64 * EST_PITCH];
    bf16* wts = lds + RING * PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- weights -> LDS once (the input-gradient role reads the taps mirrored: done here, not in the loop)
    for (int c = tid; c < 54 * 64; c += 256) {
        const int frag = c >> 6, ln = c & 63;
        const int tap = frag >> 1, ks = frag & 1;
        const int st = a.flip ? 26 - tap : tap;
        *reinterpret_cast<bf16x8*>(wts + c * 8) = a.w[(st * 2 + ks) * 64 + ln];
    }

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
    // ---- fragment offsets of this lane: output rows 2*wave + m, voxel (lane & 31) of the W-run, k-half lane >> 5
    int boff[2][3][3][2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
                    boff[m][kh][kw][ks] = piece_off((2 * wave + m + kh) * WW + (lane & 31) + kw, 2 * ks + (lane >> 5));
    const bf16* wl = wts + lane * 8;

    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        bq[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 8 * q + 4 * (lane >> 5)) : z4;
    }

    // fused InstanceNorm statistics (same slab layout as the producer/consumer kernel: [workgroup][wave][n][32][2])
    float st1[8], st2[8];
    int cur_n = -1;
    auto stat_flush = [&]() {
        if (!a.stat_slab || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)blockIdx.x * 4 + wave) * a.N + cur_n) * 32) * 2;
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
    bf16x8 aq[XD], xq[XD][2];
    bf16* est = lds + RING * PLANE + WTS + wave * (64 * EST_PITCH);

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

        bool ok[NSTG];
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const int gh = h0 - 1 + szh[i], gw = w0 - 1 + szw[i];
            ok[i] = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        }
        const int64_t plane_stride = (int64_t)a.H * a.W * a.ldx;
        const int64_t base0 = (((int64_t)n * a.D) * a.H + (h0 - 1)) * (int64_t)a.W * a.ldx + (int64_t)(w0 - 1) * a.ldx;

        auto load_plane = [&](int pr, bf16x8 (&stg)[NSTG]) {
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
            const bf16* src = a.x + base0 + (int64_t)d * plane_stride;
#pragma unroll
            for (int i = 0; i < NSTG; i++) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (dok && ok[i]) v = *reinterpret_cast<const bf16x8*>(src + srel[i]);
                stg[i] = v;
            }
        };
        auto store_plane = [&](int slot, const bf16x8 (&stg)[NSTG]) {
#pragma unroll
            for (int i = 0; i < NSTG; i++)
                if (szh[i] >= 0) *reinterpret_cast<bf16x8*>(lds + slot * PLANE + sdst[i]) = stg[i];
        };

        __syncthreads();   // the previous unit has left the ring (first pass: nothing to wait for)
        {
            bf16x8 s0[NSTG], s1[NSTG], s2[NSTG];
            load_plane(0, s0);
            load_plane(1, s1);
            load_plane(2, s2);
            store_plane(0, s0);
            store_plane(1, s1);
            store_plane(2, s2);
        }
        __syncthreads();

        if (a.stat_slab && n != cur_n) {
            stat_flush();
            cur_n = n;
#pragma unroll
            for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
        }

        // ---- fragment ring, continuous across the steps of a unit: iteration `nx` of the step with phase PHN
        auto frag_fetch = [&](auto phn, auto nxc, int ring) {
            constexpr int PHN = decltype(phn)::value, nx = decltype(nxc)::value;
            constexpr int tap = nx >> 1, ks = nx & 1;
            constexpr int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            aq[ring] = *reinterpret_cast<const bf16x8*>(wl + nx * 512);
#pragma unroll
            for (int m = 0; m < 2; m++)
                xq[ring][m] = *reinterpret_cast<const bf16x8*>(lds + ((PHN + kd) & 3) * PLANE + boff[m][kh][kw][ks]);
        };
        // ---- epilogue pieces of a finished step (accumulators `ac`, output plane d0 + sp): accumulator layout
        // (lane = voxel, 4 couts per 8 bytes) -> wave-private LDS patch -> 16-byte stores that cover whole
        // 64-byte channel rows; bias before, residual after the transpose.  The pieces are issued between the
        // MFMAs of the NEXT step, so the matrix pipe does not idle while a plane is written out.
        auto epi_write = [&](const f32x16 (&ac)[2], int m, int q) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = ac[m][q * 4 + i] + bq[q][i];
            store_vec<bf16, 4>(est + (m * 32 + (lane & 31)) * EST_PITCH + 8 * q + 4 * (lane >> 5), v);
        };
        auto epi_vox = [&](int sp, int r) {
            const int row = (lane >> 2) + 16 * r;
            return (((int64_t)n * a.D + d0 + sp) * a.H + h0 + 2 * wave + (row >> 5)) * (int64_t)a.W + w0 + (row & 31);
        };
        auto epi_row = [&](int sp, int r, const bf16x8& rres) {
            const int row = (lane >> 2) + 16 * r, part = lane & 3;
            float v[8];
            load_vec<bf16, 8>(est + row * EST_PITCH + part * 8, v);
            if (a.stat_slab) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    st1[i] += v[i];
                    st2[i] = fmaf(v[i], v[i], st2[i]);
                }
            }
            if (a.res) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] += (float)rres[i];
            }
            store_vec<bf16, 8>(a.y + epi_vox(sp, r) * a.ldy + part * 8, v);
        };

        auto step = [&](auto phc, int s) {
            constexpr int PH = decltype(phc)::value;
            constexpr int PAR = PH & 1;
            const bool pre = s + 3 <= a.DL + 1, has_prev = s > 0, last = s == a.DL - 1;
            // long-latency loads first, in the order they are consumed: residual rows of the previous plane
            // (used ~10 iterations in), then the input plane that is stored to LDS behind this step's barrier
            bf16x8 rq[4];
            if (has_prev && a.res) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    rq[r] = *reinterpret_cast<const bf16x8*>(a.res + epi_vox(s - 1, r) * a.ldr + (lane & 3) * 8);
            }
            bf16x8 stg[NSTG];
            if (pre) load_plane(s + 3, stg);

            static_for<0, 54>([&](auto itc) {
                constexpr int it = decltype(itc)::value;
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    if (it == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[PAR][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[it % XD], xq[it % XD][m], z, 0, 0, 0);
                    } else {
                        acc[PAR][m] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[it % XD], xq[it % XD][m], acc[PAR][m], 0, 0, 0);
                    }
                }
                if (it == 36 - XD) {
                    // taps with kd = 2 (iterations 36..53) read the plane that was stored behind the PREVIOUS step's
                    // barrier; every wave is past iteration 17 of this step, so the slot of plane s-1 is free
                    __syncthreads();
                    if (pre) store_plane((PH + 3) & 3, stg);
                }
                if constexpr (it + XD < 54) {
                    frag_fetch(std::integral_constant<int, PH>{}, std::integral_constant<int, (it + XD) % 54>{}, it % XD);
                } else if (!last) {
                    frag_fetch(std::integral_constant<int, (PH + 1) & 3>{}, std::integral_constant<int, (it + XD) % 54>{},
                               it % XD);
                }
                if constexpr (it >= 1 && it <= 8) {
                    if (has_prev) epi_write(acc[PAR ^ 1], (it - 1) >> 2, (it - 1) & 3);
                }
                if constexpr (it >= 10 && it <= 13) {
                    if (has_prev) epi_row(s - 1, it - 10, rq[it - 10]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        // first fragments of step 0
        static_for<0, XD>([&](auto itc) {
            frag_fetch(std::integral_constant<int, 0>{}, itc, decltype(itc)::value);
        });
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
                rq[r] = a.res ? *reinterpret_cast<const bf16x8*>(a.res + epi_vox(a.DL - 1, r) * a.ldr + (lane & 3) * 8) : z8;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) epi_write(acc[1], j >> 2, j & 3);
#pragma unroll
            for (int r = 0; r < 4; r++) epi_row(a.DL - 1, r, rq[r]);
        }
    }
    stat_flush();
}
}  // namespace

// Work decomposition: units = N x dsplit x (H/8) x (W/32) columns of DL = D/dsplit planes (+2 halo planes each).
// dsplit is the divisor of D (DL a multiple of 4) with the shortest makespan on 256 CUs.
bool slide_conv_plan(int N, int D, int H, int W, int Cin, int Cout, SlidePlan* out) {
    static const int mode = getenv("RU3D_CONV_SLIDE") ? atoi(getenv("RU3D_CONV_SLIDE")) : 1;
    if (mode == 0 || Cin != 32 || Cout != 32 || (H % TH) || (W % TW) || D < 4) return false;
    const int64_t cols = (int64_t)N * (H / TH) * (W / TW);
    int64_t best_cost = -1;
    int best = 0;
    for (int ds = 1; ds <= D / 4; ds++) {
        if (D % ds) continue;
        const int dl = D / ds;
        if (dl % 4) continue;
        const int64_t units = cols * ds;
        if (units > 0x7fffffff) break;
        const int64_t cost = ((units + 255) / 256) * (dl + 3);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = ds;
        }
    }
    if (!best) return false;
    const int64_t units = cols * best;
    // worth it only when the 256 CUs are reasonably filled
    const double ideal = (double)cols * D / 256.0;
    if (units < 128 || (double)best_cost > 1.6 * ideal + 8) return false;
    out->dsplit = best;
    out->DL = D / best;
    out->tiles_h = H / TH;
    out->tiles_w = W / TW;
    out->units = (int)units;
    int g = units < 256 ? (int)units : 256;
    if ((units % 8) == 0 && g >= 8) g = (g / 8) * 8;
    out->grid = g;
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
    hipLaunchKernelGGL(conv3_s1_slide32_kernel, dim3(p.grid), dim3(256), 0, st, a);
    return ru3d_check_launch("conv3_s1_slide32");
}
