// Weight gradient of the 3x3x3 stride-1 conv on the large levels, D-sliding form (reference: autograd of
// network.py:391-403 conv1 / conv2):   dW[tap][ci][co] = sum_pos X[pos + tap][ci] * DY[pos][co].
//
// wgrad3_s1_mfma_kernel (conv_mfma.hip) stages a (2+2)x(4+2)x(32+2) halo tile of X per 256 positions: 3.2x read
// amplification, all of it through the ~10 B/clk/CU path that bounds that kernel.  Here a persistent workgroup owns one
// (32 ci) x (32 co) pair and an (8 x 32) column in (H, W), and slides along D: a ring of 4 X planes (10 x 34 rows of
// 64 B) stays in LDS, every step loads ONE new X plane and one DY plane (256 rows of 64 B) - 38 KB per 256 positions
// instead of 68 KB - into registers at the top of the step and stores them at the top of the next one (the ring slot of
// the plane that just died / the other DY buffer), so a whole step of MFMAs covers the load latency and one barrier per
// step is enough.  Both operands are K-major in memory (a position's channels are contiguous), so fragments come from
// row-per-position LDS images through ds_read_b64_tr_b16, as in the tile kernel; the 4 waves split the 27 taps
// (7/7/7/6) and keep their 32x32 fp32 accumulators in registers for the whole run; one slab per workgroup, reduced in
// fixed order afterwards (deterministic).
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {

namespace {
constexpr int TH = 8, TW = 32, HH = TH + 2, WW = TW + 2;
constexpr int PROWS = HH * WW;        // 340 rows per X plane
constexpr int XPLANE = PROWS * 32;    // bf16 elements (64-byte rows, no pad: conflict-free transposed reads)
constexpr int DPLANE = TH * TW * 32;  // DY plane: 256 rows
constexpr int NSX = 6;                // X pieces (16 B) staged per thread and plane: 1360 / 256
constexpr int NSD = 4;                // DY pieces per thread and plane: 1024 / 256
static_assert((4 * XPLANE + 4 * DPLANE) * 2 <= 160 * 1024, "LDS budget (PAIR: a second pair of DY planes)");

struct WSlideArgs {
    const bf16* x;
    const bf16* dy;
    float* part;
    int N, D, H, W;
    int Cin, Cout, ldx, lddy;
    int x_cseg;                                // split x (planar concat): ci tile t lives in plane t * 32 / x_cseg
    int64_t x_segstride;
    int tiles_h, tiles_w, dsplit, DL, units;   // units per (ci, co) pair
    int light_last;                            // EDGE: the last column's far W half lies outside the volume
    // PAIR: the weight gradient of a 1x1x1 conv of the SAME x (the decoder block's skip_conv, network.py:403-409) with its
    // own gradient dy2, in the one tap slot of the four waves' 28 that no tap of the 3x3x3 conv fills
    const bf16* dy2;
    int lddy2;
    float* part2;                              // [slab][ci][co]
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 tr_frag(const bf16* p) {
    // 8 consecutive K (positions) of this lane's channel: two 4-row transposed reads
    const bf16x4 lo = RU3D_DS_READ_TR16(p);
    const bf16x4 hi = RU3D_DS_READ_TR16(p + 4 * 32);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// EDGE: W is not a multiple of 32.  The dy positions beyond W read zeros (dvoff); a column whose far W half lies outside
// altogether skips the MFMAs of the odd k-steps (columns 16..31 of every row) - its staging runs as always - and such
// light columns are dealt last, so that a launch with between one and two units per workgroup ends on them.
// PAIR: wave 3's seventh slot (tap 27 does not exist; the plain kernel recomputes tap 26 there and drops it) multiplies the
// CENTRE tap's X fragment with the fragment of a second gradient tensor: dW2[ci][co] = sum_pos X[pos][ci] * DY2[pos][co],
// the 1x1x1 skip conv's weight gradient, from the X planes that are in LDS anyway.  Costs a second pair of DY plane
// buffers, four more staged pieces per thread and step, and one more fragment read per k-step in wave 3; saves the
// separate pass over x (64 channels at 128^3: 124 us).
template <bool EDGE, bool PAIR>
__global__ __launch_bounds__(256, 1) void wgrad3_s1_slide_kernel(WSlideArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[4 * XPLANE + (PAIR ? 4 : 2) * DPLANE];
    bf16* const dbuf = lds + 4 * XPLANE;
    bf16* const d2buf = lds + 4 * XPLANE + 2 * DPLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool pair_wave = PAIR && wave == 3;
    const int COT = a.Cout / 32;
    const int cit = blockIdx.y / COT, cot = blockIdx.y % COT;

    // this wave's taps: wave, wave + 4, ... (tap 27 = none); element offset of the tap inside a plane
    int toff[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        // tap 27 (wave 3's 7th) does not exist: it recomputes tap 26 and is dropped at the end - the MFMA loop
        // stays free of branches
        const int tap = wave + 4 * t < 27 ? wave + 4 * t : (PAIR ? 13 : 26);
        toff[t] = (((tap / 3) % 3) * WW + tap % 3) * 32;
    }
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;   // row q of the 4x16 block, columns 4p..4p+3

    // staging: X pieces c = tid + 256 i (row c >> 2 of the plane, 16-byte piece c & 3); DY pieces likewise.  Loads go
    // through buffer descriptors of the sample (byte offsets): a piece outside the volume carries an offset beyond the
    // range and comes back as zeros, a plane outside the sample is switched off through the record count - no selects,
    // no address clamps in the loop
    bf16* const sdst = lds + (tid >> 2) * 32 + (tid & 3) * 8;
    // this workgroup's 32 input channels: a channel offset in the dense tensor, a plane + offset in the split one
    const bf16* const xbase = a.x_cseg ? a.x + (int64_t)((cit * 32) / a.x_cseg) * a.x_segstride : a.x;
    const int xc0 = a.x_cseg ? (cit * 32) % a.x_cseg : cit * 32;
    const int xplane_b = a.H * a.W * a.ldx * 2, dplane_b = a.H * a.W * a.lddy * 2;
    const int xsample_b = a.D * xplane_b, dsample_b = a.D * dplane_b;

    for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
        int t = u;
        int tw_i;
        if (EDGE && a.light_last) {
            const int heavy = a.units / a.tiles_w * (a.tiles_w - 1);
            if (t < heavy) {
                tw_i = t % (a.tiles_w - 1);
                t /= (a.tiles_w - 1);
            } else {
                tw_i = a.tiles_w - 1;
                t -= heavy;
            }
        } else {
            tw_i = t % a.tiles_w;
            t /= a.tiles_w;
        }
        const int w0 = tw_i * TW;
        const bool skip_far = EDGE && a.W - w0 <= 16;     // the far half of every row is outside: odd k-steps add zeros
        const int h0 = (t % a.tiles_h) * TH;
        t /= a.tiles_h;
        const int d0 = (t % a.dsplit) * a.DL;
        const int n = t / a.dsplit;

        int xvoff[NSX], dvoff[NSD], d2voff[NSD];
#pragma unroll
        for (int i = 0; i < NSX; i++) {
            const int r = (tid >> 2) + 64 * i, zh = r / WW, zw = r - zh * WW;
            const int gh = h0 - 1 + zh, gw = w0 - 1 + zw;
            const bool ok = r < PROWS && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            xvoff[i] = ok ? ((gh * a.W + gw) * a.ldx + xc0 + (tid & 3) * 8) * 2 : (int)0x80000000;
        }
#pragma unroll
        for (int i = 0; i < NSD; i++) {
            const int f = (tid >> 2) + 64 * i;
            // a column of 32 may stick out of a W that is only a multiple of 16: those positions read zeros (no
            // contribution), as the X halo does
            dvoff[i] = (w0 + f % TW < a.W) ? (((h0 + f / TW) * a.W + w0 + f % TW) * a.lddy + cot * 32 + (tid & 3) * 8) * 2
                                           : (int)0x80000000;
            d2voff[i] = (PAIR && w0 + f % TW < a.W)
                            ? (((h0 + f / TW) * a.W + w0 + f % TW) * a.lddy2 + cot * 32 + (tid & 3) * 8) * 2
                            : (int)0x80000000;
        }
        const bf16* xs = xbase + (int64_t)n * a.D * (xplane_b / 2);
        const bf16* ds = a.dy + (int64_t)n * a.D * (dplane_b / 2);
        const int d2plane_b = PAIR ? a.H * a.W * a.lddy2 * 2 : 0, d2sample_b = a.D * d2plane_b;
        const bf16* ds2 = PAIR ? a.dy2 + (int64_t)n * a.D * (d2plane_b / 2) : nullptr;

        bf16x8 sx[NSX], sd[NSD], sd2[NSD];
        auto load_x1 = [&](int pr, auto ic) {      // piece i of X plane d0 - 1 + pr
            constexpr int i = decltype(ic)::value;
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xs, (short)0, dok ? xsample_b : 0, 0x00020000);
            sx[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, xvoff[i], dok ? d * xplane_b : 0, 0));
        };
        auto store_x1 = [&](int slot, auto ic) {
            constexpr int i = decltype(ic)::value;
            // the last piece covers rows 320..383 of 340: threads 0..79 only
            if (i * 64 + 64 <= PROWS || tid < (PROWS - i * 64) * 4)
                *reinterpret_cast<bf16x8*>(sdst + slot * XPLANE + i * 64 * 32) = sx[i];
        };
        auto load_d1 = [&](int s, auto ic) {       // piece i of DY plane d0 + s
            constexpr int i = decltype(ic)::value;
            const int d = d0 + s;
            const bool dok = d < a.D;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ds, (short)0, dok ? dsample_b : 0, 0x00020000);
            sd[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, dvoff[i], dok ? d * dplane_b : 0, 0));
        };
        auto store_d1 = [&](int buf, auto ic) {
            constexpr int i = decltype(ic)::value;
            *reinterpret_cast<bf16x8*>(sdst + 4 * XPLANE + buf * DPLANE + i * 64 * 32) = sd[i];
        };
        auto load_d2_1 = [&](int s, auto ic) {      // piece i of DY2 plane d0 + s
            constexpr int i = decltype(ic)::value;
            const int d = d0 + s;
            const bool dok = d < a.D;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ds2, (short)0, dok ? d2sample_b : 0, 0x00020000);
            sd2[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, d2voff[i], dok ? d * d2plane_b : 0, 0));
        };
        auto store_d2_1 = [&](int buf, auto ic) {
            constexpr int i = decltype(ic)::value;
            *reinterpret_cast<bf16x8*>(sdst + 4 * XPLANE + (2 + buf) * DPLANE + i * 64 * 32) = sd2[i];
        };
        auto load_x = [&](int pr) { static_for<0, NSX>([&](auto ic) { load_x1(pr, ic); }); };
        auto store_x = [&](int slot) { static_for<0, NSX>([&](auto ic) { store_x1(slot, ic); }); };
        auto load_d = [&](int s) { static_for<0, NSD>([&](auto ic) { load_d1(s, ic); }); };
        auto store_d = [&](int buf) { static_for<0, NSD>([&](auto ic) { store_d1(buf, ic); }); };

        __syncthreads();   // the previous unit has left the ring
        load_x(0); store_x(0);
        load_x(1); store_x(1);
        load_x(2); store_x(2);
        load_d(0); store_d(0);
        if constexpr (PAIR) {
            static_for<0, NSD>([&](auto ic) { load_d2_1(0, ic); });
            static_for<0, NSD>([&](auto ic) { store_d2_1(0, ic); });
        }
        load_x(3);
        load_d(1);
        if constexpr (PAIR) static_for<0, NSD>([&](auto ic) { load_d2_1(1, ic); });
        __syncthreads();

        auto step = [&](auto phc, int s) {
            constexpr int PH = decltype(phc)::value;
            const bf16* db = dbuf + (PH & 1) * DPLANE;
            const bf16* d2b = d2buf + (PH & 1) * DPLANE;
            // fragments of k-step ks + 1 are fetched before the MFMAs of k-step ks (one wave per SIMD: nothing else
            // hides the LDS latency)
            bf16x8 bq[3], bq2[3], aq[3][7];
            auto fetch = [&](auto ksc, int buf) {
                constexpr int ks = decltype(ksc)::value;
                // positions f0 = 16 ks + 8 h: row ks >> 1 of the 8 x 32 plane tile, columns 16 (ks & 1) + 8 h ..
                constexpr int rowb = (ks >> 1) * WW + (ks & 1) * 16;
                bq[buf] = tr_frag(db + (ks * 16 + 8 * h) * 32 + lane_off);
                if (pair_wave) bq2[buf] = tr_frag(d2b + (ks * 16 + 8 * h) * 32 + lane_off);
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const int tap = wave + 4 * t < 27 ? wave + 4 * t : (PAIR ? 13 : 26);
                    const int slot = (PH + tap / 9) & 3;
                    aq[buf][t] = tr_frag(lds + slot * XPLANE + (rowb + 8 * h) * 32 + toff[t] + lane_off);
                }
            };
            fetch(std::integral_constant<int, 0>{}, 0);
            fetch(std::integral_constant<int, 1>{}, 1);
            static_for<0, 16>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                // the fragments of k-step ks + 2 are read one tap behind each MFMA (two transposed LDS reads fit in an
                // MFMA's shadow; all 16 in front of the seven MFMAs held the issue port while the matrix pipe drained)
                constexpr int rowb2 = (((ks + 2) & 15) >> 1) * WW + (((ks + 2) & 15) & 1) * 16;
                static_for<0, 7>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    // the pair slot takes the second gradient's fragment (a wave-uniform choice)
                    const bf16x8 bsel = (PAIR && t == 6 && pair_wave) ? bq2[ks % 3] : bq[ks % 3];
                    if constexpr (EDGE && (ks & 1)) {
                        if (!skip_far) acc[t] = RU3D_MFMA_32X32X16(aq[ks % 3][t], bsel, acc[t], 0, 0, 0);
                    } else {
                        acc[t] = RU3D_MFMA_32X32X16(aq[ks % 3][t], bsel, acc[t], 0, 0, 0);
                    }
                    if constexpr (ks + 2 < 16) {
                        // (EDGE: the fragments of a k-step that issues no MFMAs are not fetched either)
                        if (!(EDGE && ((ks + 2) & 1)) || !skip_far) {
                            if constexpr (t == 0) bq[(ks + 2) % 3] = tr_frag(db + ((ks + 2) * 16 + 8 * h) * 32 + lane_off);
                            if constexpr (PAIR && t == 1) {
                                if (pair_wave) bq2[(ks + 2) % 3] = tr_frag(d2b + ((ks + 2) * 16 + 8 * h) * 32 + lane_off);
                            }
                            const int tap = wave + 4 * t < 27 ? wave + 4 * t : (PAIR ? 13 : 26);
                            const int slot = (PH + tap / 9) & 3;
                            aq[(ks + 2) % 3][t] = tr_frag(lds + slot * XPLANE + (rowb2 + 8 * h) * 32 + toff[t] + lane_off);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // Staging, one 16-byte piece behind each k-step's MFMAs (as one block at the top of the step it idled the
                // matrix pipe for its ~120 instructions): the X plane / DY rows loaded during the previous step go to the
                // ring slot and DY buffer that died with it, and the freed registers take the same piece of the next
                // plane.  Unconditional: planes past the column are never read, loads past the sample return zeros.
                if constexpr (ks < NSX) {
                    store_x1((PH + 3) & 3, std::integral_constant<int, (ks < NSX ? ks : 0)>{});
                    load_x1(s + 4, std::integral_constant<int, (ks < NSX ? ks : 0)>{});
                } else if constexpr (ks < NSX + NSD) {
                    store_d1((PH + 1) & 1, std::integral_constant<int, (ks >= NSX && ks < NSX + NSD ? ks - NSX : 0)>{});
                    load_d1(s + 2, std::integral_constant<int, (ks >= NSX && ks < NSX + NSD ? ks - NSX : 0)>{});
                } else if constexpr (PAIR && ks < NSX + 2 * NSD) {
                    store_d2_1((PH + 1) & 1, std::integral_constant<int, (ks >= NSX + NSD && ks < NSX + 2 * NSD ? ks - NSX - NSD : 0)>{});
                    load_d2_1(s + 2, std::integral_constant<int, (ks >= NSX + NSD && ks < NSX + 2 * NSD ? ks - NSX - NSD : 0)>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // LDS-only barrier: the step's ring / DY stores are visible, every wave is done with the slots the next step
            // overwrites (a __syncthreads() would also wait for the global loads just issued)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        for (int s4 = 0; s4 < a.DL; s4 += 4) {
            step(std::integral_constant<int, 0>{}, s4);
            step(std::integral_constant<int, 1>{}, s4 + 1);
            step(std::integral_constant<int, 2>{}, s4 + 2);
            step(std::integral_constant<int, 3>{}, s4 + 3);
        }
    }
    // partial slab: part[((slab * 27 + tap) * Cin + ci) * Cout + co]; D row = ci, col = co
    if (PAIR && pair_wave) {      // the skip conv's slab: [slab][ci][co]
        float* pp = a.part2 + (int64_t)blockIdx.x * a.Cin * a.Cout;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            const int co = cot * 32 + (lane & 31);
            pp[(int64_t)ci * a.Cout + co] = acc[6][i];
        }
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t;
        if (tap < 27) {
            float* pp = a.part + ((int64_t)blockIdx.x * 27 + tap) * a.Cin * a.Cout;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int co = cot * 32 + (lane & 31);
                pp[(int64_t)ci * a.Cout + co] = acc[t][i];
            }
        }
    }
}
}  // namespace

// units per pair = N x dsplit x (H/8) x (W/32); G persistent workgroups per pair (= slabs), pairs on grid.y.
bool wgrad_slide_plan(const WgradGeom& g, WgradSlidePlan* out) {
    static const int mode = getenv("RU3D_WGRAD_SLIDE") ? atoi(getenv("RU3D_WGRAD_SLIDE")) : 1;
    if (!mode || g.k != 3 || g.stride != 1 || (g.Cin % 32) || (g.Cout % 32) || (g.Ho % TH) || (g.Wo % 8) || g.Wo < 16) return false;   // dy positions beyond W are masked one by one
    if (g.Do != g.Di || g.Ho != g.Hi || g.Wo != g.Wi || (g.ldx % 8) || (g.lddy % 8)) return false;
    // buffer-descriptor byte offsets of a sample, the top bit marking "outside the volume"
    if ((int64_t)g.Do * g.Ho * g.Wo * (g.ldx > g.lddy ? g.ldx : g.lddy) >= (1ll << 30)) return false;
    const int pairs = (g.Cin / 32) * (g.Cout / 32);
    static const int max_pairs = getenv("RU3D_WGRAD_SLIDE_PAIRS") ? atoi(getenv("RU3D_WGRAD_SLIDE_PAIRS")) : 32;
    if (pairs > max_pairs) return false;      // up to 256 -> 128 channels (round 3: 8 -> 16 pairs, 89 -> 62 us at 32^3)
    const int tw_n = (g.Wo + TW - 1) / TW;
    const bool light = (g.Wo % TW) != 0 && (g.Wo % TW) <= 16 && tw_n > 1;
    const int64_t cols = (int64_t)g.N * (g.Ho / TH) * tw_n;
    const int gmax = ru3d_get_cu_budget() / pairs;             // one workgroup per CU in total
    int64_t best_cost = -1;
    int best = 0;
    for (int ds = 1; ds <= g.Do / 8; ds++) {
        if (g.Do % ds) continue;
        const int dl = g.Do / ds;
        if (dl % 4) continue;
        const int64_t units = cols * ds;
        const int64_t gx = units < gmax ? units : gmax;
        int64_t cost = ((units + gx - 1) / gx) * (dl + 4);
        if (light) {    // heavy-first deal: workgroup 0 carries the longest chain; light units run half the k-steps
            const int64_t heavy = units / tw_n * (tw_n - 1);
            cost = 0;
            for (int64_t u = 0; u < units; u += gx) cost += u < heavy ? dl + 4 : (dl * 9) / 16 + 4;
        }
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = ds;
        }
    }
    if (!best) return false;
    const int64_t units = cols * best;
    const double ideal = (double)cols * pairs * g.Do / (double)ru3d_get_cu_budget();
    if (units * pairs < 192 || (double)best_cost > 1.5 * ideal + 8) return false;
    out->dsplit = best;
    out->DL = g.Do / best;
    out->tiles_h = g.Ho / TH;
    out->tiles_w = (g.Wo + TW - 1) / TW;
    out->units = (int)units;
    out->G = (int)(units < gmax ? units : gmax);
    out->pairs = pairs;
    return true;
}

size_t wgrad_slide_ws_bytes(const WgradGeom& g) {
    WgradSlidePlan p;
    if (!wgrad_slide_plan(g, &p)) return 0;
    return (size_t)p.G * 27 * g.Cin * g.Cout * sizeof(float);
}

// the pair form's slabs: 27 + 1 taps
size_t wgrad_slide_pair_ws_bytes(const WgradGeom& g) {
    WgradSlidePlan p;
    if (!wgrad_slide_plan(g, &p)) return 0;
    return (size_t)p.G * 28 * g.Cin * g.Cout * sizeof(float);
}

int wgrad_slide_launch(const void* x, const void* dy, float* dw, void* ws, const WgradGeom& g, hipStream_t st) {
    return wgrad_slide_pair_launch(x, dy, nullptr, 0, dw, nullptr, ws, g, st);
}

// dy2 != nullptr: also dw2[co][ci] = sum_pos x[pos][ci] * dy2[pos][co] (a 1x1x1 conv of the same x), PAIR kernel
int wgrad_slide_pair_launch(const void* x, const void* dy, const void* dy2, int lddy2, float* dw, float* dw2, void* ws,
                            const WgradGeom& g, hipStream_t st) {
    WgradSlidePlan p;
    if (!wgrad_slide_plan(g, &p)) return ru3d_fail(-1, "wgrad_slide: shape not supported");
    WSlideArgs a;
    a.x = (const bf16*)x;
    a.dy = (const bf16*)dy;
    a.part = (float*)ws;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.lddy = g.lddy;
    a.x_cseg = g.x_cseg; a.x_segstride = g.x_segstride;
    if (g.x_cseg && (g.x_cseg % 32)) return ru3d_fail(-1, "wgrad_slide: split x needs segments of whole 32-channel tiles");
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
    a.light_last = (g.Wo % TW) != 0 && (g.Wo % TW) <= 16 && p.tiles_w > 1;
    a.dy2 = (const bf16*)dy2;
    a.lddy2 = lddy2;
    a.part2 = (float*)ws + (size_t)p.G * 27 * g.Cin * g.Cout;
    if (dy2) {
        if (g.Wo % TW) hipLaunchKernelGGL((wgrad3_s1_slide_kernel<true, true>), dim3(p.G, p.pairs), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad3_s1_slide_kernel<false, true>), dim3(p.G, p.pairs), dim3(256), 0, st, a);
    } else {
        if (g.Wo % TW) hipLaunchKernelGGL((wgrad3_s1_slide_kernel<true, false>), dim3(p.G, p.pairs), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wgrad3_s1_slide_kernel<false, false>), dim3(p.G, p.pairs), dim3(256), 0, st, a);
    }
    int rc = ru3d_check_launch("wgrad3_s1_slide");
    if (rc) return rc;
    rc = wgrad_reduce_launch((const float*)ws, dw, p.G, 27, g.Cin, g.Cout, g.s_o, g.s_i, st);
    if (rc || !dy2) return rc;
    // the 1x1x1 weight in the reference layout [Cout][Cin]
    return wgrad_reduce_launch(a.part2, dw2, p.G, 1, g.Cin, g.Cout, (int64_t)g.Cin, 1, st);
}

}  // namespace RU3D_NS
