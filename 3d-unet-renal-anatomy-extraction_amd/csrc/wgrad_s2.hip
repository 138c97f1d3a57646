// Weight gradient of the 3x3x3 stride-2 conv (pooling ResBlock conv1, network.py:391-403 with stride 2) and of
// ConvTranspose3d k3 s2 p1 (network.py:312; the same sum with the operands' roles swapped, api.hip make_convt_wgrad):
//     dW[tap][ci][co] = sum_pos X[2 pos + tap - 1][ci] * DY[pos][co]
// on the large levels.  wgrad_staged_mfma_kernel stages every tap's gathered rows separately (27 x 64 B per position,
// 1.7 KB of loads per position).  Here a workgroup stages the (2x2x32)-position tile's 5 x 5 x 65 input rows ONCE
// (0.87 KB per position), de-interleaved by W parity: a tap with kw = 1 reads the odd half of a line, kw = 0 / 2 the even
// half (shifted by one for kw = 2), so the 8 consecutive positions of a fragment are 8 consecutive LDS rows and the
// hardware-transposed read (ds_read_b64_tr_b16) works as in the stride-1 kernels.  The next tile's rows are loaded into
// registers before the MFMA loop of the current one; 4 waves split the 27 taps; slabs + fixed-order reduce.
#include "common.h"
#include "conv.h"

namespace RU3D_NS {

namespace {
constexpr int TDO = 2, THO = 2, TWO = 32;                  // dense-operand (output-side) tile: 128 positions
constexpr int LD = 2 * TDO + 1, LH = 2 * THO + 1, LW = 2 * TWO + 1;   // 5 x 5 lines of 65 gathered rows
constexpr int NE = TWO + 1;                                // even entries of a line (33), odd: 32
constexpr int XROWS = LD * LH * LW;                        // 1625
constexpr int NPOS = TDO * THO * TWO;                      // 128
constexpr int NSX = (XROWS * 4 + 255) / 256;               // 26 pieces per thread
constexpr int NSD = NPOS * 4 / 256;                        // 2 pieces per thread and 32-cout tile
static_assert((XROWS + 2 * NPOS) * 64 <= 160 * 1024, "LDS budget");

struct WS2Args {
    const bf16* x;
    const bf16* dy;
    float* part;
    int N, Di, Hi, Wi, Do, Ho, Wo;
    int Cin, Cout, ldx, lddy;
    int tiles_d, tiles_h, tiles_w, ntiles, G;
    // PAIR (DMA kernel): the 1x1x1 stride-2 skip conv's weight gradient (same x, its own gradient dy2 shaped like dy) in
    // the tap slot no tap of the 3x3x3 conv fills - the centre tap's gathered rows ARE the rows that conv reads
    const bf16* dy2;
    int lddy2;
    float* part2;      // [slab][ci][co]
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 tr_frag(const bf16* p) {
    const bf16x4 lo = RU3D_DS_READ_TR16(p);
    const bf16x4 hi = RU3D_DS_READ_TR16(p + 4 * 32);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

// NCO = 32-cout tiles per workgroup: the gathered tile is the expensive operand, so one staging of it serves NCO
// output-channel tiles (7 x NCO accumulators per wave).
template <int NCO>
__global__ __launch_bounds__(256, 1) void wgrad3_s2_tile_kernel(WS2Args a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[(XROWS + NCO * NPOS) * 32];
    bf16* const xs = lds;
    bf16* const ds = lds + XROWS * 32;   // [NCO][NPOS][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int COT = a.Cout / (32 * NCO);
    const int cit = blockIdx.y / COT, cot = (blockIdx.y % COT) * NCO;

    // this wave's taps (the non-existent tap 27 recomputes tap 26 and is dropped): row offset of the tap inside the tile
    int toff[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t < 27 ? wave + 4 * t : 26;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        toff[t] = ((kd * LH + kh) * LW + (kw == 1 ? NE : (kw >> 1))) * 32;
    }
    f32x16 acc[7][NCO];
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int o = 0; o < NCO; o++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][o][i] = 0.f;
    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;

    // gathered rows r = c >> 2 (line r / 65, entry r % 65), piece c & 3 for c = tid + 256 i: recomputed where needed
    // (26 x 4 registers of per-thread constants would push the accumulators out of the register file)
    bf16x8 sx[NSX], sd[NCO][NSD];
    auto load_tile = [&](int tile) {
        int tt = tile;
        const int ow0 = (tt % a.tiles_w) * TWO;
        tt /= a.tiles_w;
        const int oh0 = (tt % a.tiles_h) * THO;
        tt /= a.tiles_h;
        const int od0 = (tt % a.tiles_d) * TDO;
        const int n = tt / a.tiles_d;
#pragma unroll
        for (int i = 0; i < NSX; i++) {
            const int c = tid + 256 * i;
            const int r = c >> 2, line = r / LW, e = r - line * LW;
            const int id = 2 * od0 - 1 + line / LH, ih = 2 * oh0 - 1 + line % LH, iw = 2 * ow0 - 1 + e;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < XROWS * 4 && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                v = *reinterpret_cast<const bf16x8*>(a.x + ((((int64_t)n * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.ldx +
                                                     cit * 32 + ((tid + 256 * i) & 3) * 8);
            sx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NSD; i++) {
            const int c = tid + 256 * i;
            const int f = c >> 2;
            const int od = od0 + f / (THO * TWO), oh = oh0 + (f / TWO) % THO, ow = ow0 + f % TWO;
            const bool ok = od < a.Do && oh < a.Ho && ow < a.Wo;
            const bf16* src = a.dy + ((((int64_t)n * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.lddy + cot * 32 + (c & 3) * 8;
#pragma unroll
            for (int o = 0; o < NCO; o++) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const bf16x8*>(src + o * 32);
                sd[o][i] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NSX; i++) {
            const int c = tid + 256 * i;
            const int r = c >> 2, line = r / LW, e = r - line * LW;
            if (c < XROWS * 4)
                *reinterpret_cast<bf16x8*>(xs + (line * LW + ((e & 1) ? NE + (e >> 1) : (e >> 1))) * 32 + (c & 3) * 8) = sx[i];
        }
#pragma unroll
        for (int o = 0; o < NCO; o++)
#pragma unroll
            for (int i = 0; i < NSD; i++) {
                const int c = tid + 256 * i;
                *reinterpret_cast<bf16x8*>(ds + o * (NPOS * 32) + (c >> 2) * 32 + (c & 3) * 8) = sd[o][i];
            }
    };

    if ((int)blockIdx.x < a.ntiles) {
        load_tile(blockIdx.x);
        store_tile();
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.G) {
        __syncthreads();   // this tile's rows are in LDS
        const bool more = tile + a.G < a.ntiles;
        if (more) load_tile(tile + a.G);
#pragma unroll
        for (int ks = 0; ks < NPOS / 16; ks++) {
            // positions f0 = 16 ks + 8 h: output (ks >> 2, (ks >> 1) & 1, 16 (ks & 1) + 8 h ..): gathered line (2 od, 2 oh)
            const int rowb = ((2 * (ks >> 2)) * LH + 2 * ((ks >> 1) & 1)) * LW + (ks & 1) * 16 + 8 * h;
            bf16x8 bfrag[NCO];
#pragma unroll
            for (int o = 0; o < NCO; o++) bfrag[o] = tr_frag(ds + o * (NPOS * 32) + (ks * 16 + 8 * h) * 32 + lane_off);
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const bf16x8 afrag = tr_frag(xs + rowb * 32 + toff[t] + lane_off);
#pragma unroll
                for (int o = 0; o < NCO; o++)
                    acc[t][o] = RU3D_MFMA_32X32X16(afrag, bfrag[o], acc[t][o], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the k-steps apart: hoisting all 64 fragment reads spills
        }
        __syncthreads();   // every wave is done with this tile's rows
        if (more) store_tile();
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t;
        if (tap < 27) {
            float* pp = a.part + ((int64_t)blockIdx.x * 27 + tap) * a.Cin * a.Cout;
#pragma unroll
            for (int o = 0; o < NCO; o++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const int co = (cot + o) * 32 + (lane & 31);
                    pp[(int64_t)ci * a.Cout + co] = acc[t][o][i];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient with LDS-DMA staging.  The kernel above runs at the rate its CU takes bytes in: 104 KB of
// gathered input per 128 positions and per (ci, co) tile pair - 852 MB for the 32 -> 64 conv at 128^3 where 335 MB are
// needed - and cannot serve two cout tiles from one staging because 14 accumulators + 26 staging pieces do not fit the
// register file.  Here the tile is half as wide (2 x 2 x 16 positions, 5 x 5 lines of 33 rows = 52 KB), double-buffered
// in LDS, and filled by `buffer_load_dwordx4 ... lds`: a wave-instruction moves 64 x 16 bytes from 64 per-lane source
// addresses into 1 KB of consecutive LDS, so the de-interleaved row order is just the order of the slots, rows outside
// the volume come back as zeros (offset beyond the descriptor's range) and NO staging register exists - which makes
// room for NCO = 2 (half the staging per output) with the 14 accumulators where the compiler puts them.
constexpr int DW = 16;                                  // W extent of the output tile
constexpr int DLW = 2 * DW + 1, DNE = DW + 1;           // 33 gathered rows per line: 17 even entries, 16 odd
constexpr int DXROWS = LD * LH * DLW;                   // 825
constexpr int DNPOS = TDO * THO * DW;                   // 64 positions
constexpr int DXINSTR = (DXROWS * 4 + 63) / 64;         // 52 wave-instructions fill the input tile
constexpr int DXI = (DXINSTR + 3) / 4;                  // 13 per wave
constexpr int DXBUF = DXINSTR * 64 * 8;                 // elements per input buffer (53,248 B incl. the tail slots)
static_assert(2 * (DXBUF + 4 * DNPOS * 32) * 2 <= 160 * 1024, "LDS budget (PAIR: the second gradient's rows)");

template <int NCO, bool PAIR>
__global__ __launch_bounds__(256, 1) void wgrad3_s2_dma_kernel(WS2Args a) {
    constexpr int NDY = PAIR ? 2 * NCO : NCO;          // DY tiles per buffer: dy's, then dy2's
    constexpr int BUFE = DXBUF + NDY * DNPOS * 32;
    __shared__ __attribute__((aligned(16))) bf16 lds[2 * BUFE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int COT = a.Cout / (32 * NCO);
    const int cit = blockIdx.y / COT, cot = (blockIdx.y % COT) * NCO;
    const bool pair_wave = PAIR && wave == 3;

    int toff[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t < 27 ? wave + 4 * t : (PAIR ? 13 : 26);
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        toff[t] = ((kd * LH + kh) * DLW + (kw == 1 ? DNE : (kw >> 1))) * 32;
    }
    f32x16 acc[7][NCO];
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int o = 0; o < NCO; o++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][o][i] = 0.f;
    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;

    // slot k = (wave + 4 i) * 64 + lane of the input tile: LDS row k >> 2 (line, then the even entries, then the odd
    // ones), 16-byte piece k & 3.  Tile-invariant: byte offset against the tile's first gathered row, and the row's
    // (d, h, w) position inside the tile for the bounds test of border tiles
    int rel[DXI], zz[DXI];
#pragma unroll
    for (int i = 0; i < DXI; i++) {
        const int k = (wave + 4 * i) * 64 + lane;
        const int r = k >> 2, line = r / DLW, within = r - line * DLW;
        const int e = within < DNE ? 2 * within : 2 * (within - DNE) + 1;
        const int dz = line / LH, dh = line - dz * LH;
        const bool slot_ok = wave + 4 * i < DXINSTR && r < DXROWS;
        rel[i] = (((dz * a.Hi + dh) * a.Wi + e) * a.ldx + (k & 3) * 8) * 2;
        zz[i] = slot_ok ? (dz | (dh << 8) | (e << 16)) : -1;
    }
    const int xsample_b = a.Di * a.Hi * a.Wi * a.ldx * 2, dsample_b = a.Do * a.Ho * a.Wo * a.lddy * 2;
    const int d2sample_b = PAIR ? a.Do * a.Ho * a.Wo * a.lddy2 * 2 : 0;

    auto issue_tile = [&](int tile, int buf) {
        int tt = tile;
        const int ow0 = (tt % a.tiles_w) * DW;
        tt /= a.tiles_w;
        const int oh0 = (tt % a.tiles_h) * THO;
        tt /= a.tiles_h;
        const int od0 = (tt % a.tiles_d) * TDO;
        const int n = tt / a.tiles_d;
        const int id0 = 2 * od0 - 1, ih0 = 2 * oh0 - 1, iw0 = 2 * ow0 - 1;
        const bool interior = id0 >= 0 && id0 + LD <= a.Di && ih0 >= 0 && ih0 + LH <= a.Hi && iw0 >= 0 && iw0 + DLW <= a.Wi;
        const int org = (((id0 * a.Hi + ih0) * a.Wi + iw0) * a.ldx + cit * 32) * 2;   // may be negative on a border tile
        const ru3d_i32x4 rx = ru3d_buffer_rsrc(a.x + (int64_t)n * (xsample_b / 2), xsample_b);
        bf16* xb = lds + buf * BUFE;
#pragma unroll
        for (int i = 0; i < DXI; i++) {
            if (wave + 4 * i < DXINSTR) {                                            // wave-uniform
                bool ok = zz[i] >= 0;
                if (!interior) {
                    const int id = id0 + (zz[i] & 255), ih = ih0 + ((zz[i] >> 8) & 255), iw = iw0 + ((zz[i] >> 16) & 255);
                    ok = ok && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                }
                ru3d_lds_dma16(rx, xb + (wave + 4 * i) * 512, ok ? org + rel[i] : (int)0x80000000);
            }
        }
        // DY rows: instruction j = wave + 4 o of NCO * 4: cout tile o, positions 16 wave + (lane >> 2), piece lane & 3
        const ru3d_i32x4 rd = ru3d_buffer_rsrc(a.dy + (int64_t)n * (dsample_b / 2), dsample_b);
        const int pos = 16 * wave + (lane >> 2);
        const int od = od0 + (pos >> 5), oh = oh0 + ((pos >> 4) & 1), ow = ow0 + (pos & 15);
        const int doff = (((od * a.Ho + oh) * a.Wo + ow) * a.lddy + cot * 32 + (lane & 3) * 8) * 2;
        bf16* db = xb + DXBUF;
#pragma unroll
        for (int o = 0; o < NCO; o++)
            ru3d_lds_dma16(rd, db + o * (DNPOS * 32) + wave * 512, doff + o * 64);
        if constexpr (PAIR) {
            const ru3d_i32x4 rd2 = ru3d_buffer_rsrc(a.dy2 + (int64_t)n * (d2sample_b / 2), d2sample_b);
            const int d2off = (((od * a.Ho + oh) * a.Wo + ow) * a.lddy2 + cot * 32 + (lane & 3) * 8) * 2;
#pragma unroll
            for (int o = 0; o < NCO; o++)
                ru3d_lds_dma16(rd2, db + (NCO + o) * (DNPOS * 32) + wave * 512, d2off + o * 64);
        }
    };

    if ((int)blockIdx.x < a.ntiles) issue_tile(blockIdx.x, 0);
    int buf = 0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.G, buf ^= 1) {
        ru3d_dma_landed_barrier();   // this tile's rows have landed, the other buffer is free
        if (tile + a.G < a.ntiles) issue_tile(tile + a.G, buf ^ 1);
        const bf16* xs = lds + buf * BUFE;
        const bf16* ds = xs + DXBUF;
#pragma unroll
        for (int ks = 0; ks < DNPOS / 16; ks++) {
            // positions 16 ks + 8 h ..: output (ks >> 1, ks & 1, 8 h ..): gathered line (2 od, 2 oh), even entries 8 h ..
            const int rowb = ((2 * (ks >> 1)) * LH + 2 * (ks & 1)) * DLW + 8 * h;
            bf16x8 bfrag[NCO], bfrag2[NCO];
#pragma unroll
            for (int o = 0; o < NCO; o++) bfrag[o] = tr_frag(ds + o * (DNPOS * 32) + (ks * 16 + 8 * h) * 32 + lane_off);
            if (pair_wave) {
#pragma unroll
                for (int o = 0; o < NCO; o++)
                    bfrag2[o] = tr_frag(ds + (NCO + o) * (DNPOS * 32) + (ks * 16 + 8 * h) * 32 + lane_off);
            }
#pragma unroll
            for (int t = 0; t < 7; t++) {
                const bf16x8 afrag = tr_frag(xs + rowb * 32 + toff[t] + lane_off);
#pragma unroll
                for (int o = 0; o < NCO; o++) {
                    const bf16x8 bsel = (PAIR && t == 6 && pair_wave) ? bfrag2[o] : bfrag[o];
                    acc[t][o] = RU3D_MFMA_32X32X16(afrag, bsel, acc[t][o], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PAIR && pair_wave) {
        float* pp = a.part2 + (int64_t)blockIdx.x * a.Cin * a.Cout;
#pragma unroll
        for (int o = 0; o < NCO; o++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int co = (cot + o) * 32 + (lane & 31);
                pp[(int64_t)ci * a.Cout + co] = acc[6][o][i];
            }
    }
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t;
        if (tap < 27) {
            float* pp = a.part + ((int64_t)blockIdx.x * 27 + tap) * a.Cin * a.Cout;
#pragma unroll
            for (int o = 0; o < NCO; o++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const int co = (cot + o) * 32 + (lane & 31);
                    pp[(int64_t)ci * a.Cout + co] = acc[t][o][i];
                }
        }
    }
}

// ---- dispatch: the LDS-DMA kernel (16-wide tiles, two cout tiles per workgroup when Cout % 64 == 0) where W fits its
// tiles, the register-staged kernel elsewhere
static int s2_mode() { return 2; }
static bool s2_dma(const WgradGeom& g) {
    return s2_mode() >= 2 && (g.Wo % DW) == 0 &&
           (int64_t)g.Di * g.Hi * g.Wi * g.ldx < (1ll << 30) && (int64_t)g.Do * g.Ho * g.Wo * g.lddy < (1ll << 30);
}
static int s2_nco(const WgradGeom& g) { return (s2_dma(g) && (g.Cout % 64) == 0) ? 2 : 1; }
static int s2_tw(const WgradGeom& g) { return s2_dma(g) ? DW : TWO; }

static int s2_groups(const WgradGeom& g, int64_t ntiles) {
    const int pairs = (g.Cin / 32) * (g.Cout / (32 * s2_nco(g)));
    int64_t G = ru3d_get_cu_budget() / pairs;            // one workgroup per CU (112-123 KB of LDS each)
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
    return (int)G;
}
}  // namespace

bool wgrad_s2_eligible(const WgradGeom& g) {
    if (!s2_mode() || g.k != 3 || g.stride != 2 || g.pad != 1 || (g.Cin % 32) || (g.Cout % 32) || (g.ldx % 8) || (g.lddy % 8))
        return false;
    const int tw = s2_tw(g);
    if ((g.Wo % tw) || (g.Ho % THO) || (g.Do % TDO)) return false;
    const int pairs = (g.Cin / 32) * (g.Cout / (32 * s2_nco(g)));
    const int64_t ntiles = (int64_t)g.N * (g.Do / TDO) * (g.Ho / THO) * (g.Wo / tw);
    return pairs <= 32 && ntiles * pairs >= 192 && ntiles <= 0x7fffffff;
}

size_t wgrad_s2_ws_bytes(const WgradGeom& g) {
    if (!wgrad_s2_eligible(g)) return 0;
    const int64_t ntiles = (int64_t)g.N * (g.Do / TDO) * (g.Ho / THO) * (g.Wo / s2_tw(g));
    return (size_t)s2_groups(g, ntiles) * 27 * g.Cin * g.Cout * sizeof(float);
}

bool wgrad_s2_pair_eligible(const WgradGeom& g) { return wgrad_s2_eligible(g) && s2_dma(g) && !g.x_cseg; }

size_t wgrad_s2_pair_ws_bytes(const WgradGeom& g) {
    if (!wgrad_s2_pair_eligible(g)) return 0;
    const int64_t ntiles = (int64_t)g.N * (g.Do / TDO) * (g.Ho / THO) * (g.Wo / s2_tw(g));
    return (size_t)s2_groups(g, ntiles) * 28 * g.Cin * g.Cout * sizeof(float);
}

int wgrad_s2_launch(const void* x, const void* dy, float* dw, void* ws, const WgradGeom& g, hipStream_t st) {
    return wgrad_s2_pair_launch(x, dy, nullptr, 0, dw, nullptr, ws, g, st);
}

// dy2 != nullptr: also dw2[co][ci] = sum_pos x[2 pos][ci] * dy2[pos][co] (the 1x1x1 stride-2 conv of the same x)
int wgrad_s2_pair_launch(const void* x, const void* dy, const void* dy2, int lddy2, float* dw, float* dw2, void* ws,
                         const WgradGeom& g, hipStream_t st) {
    if (!wgrad_s2_eligible(g)) return ru3d_fail(-1, "wgrad_s2: shape not supported");
    if (dy2 && !wgrad_s2_pair_eligible(g)) return ru3d_fail(-1, "wgrad_s2: no pair form for this shape");
    const int tw = s2_tw(g), nco = s2_nco(g);
    WS2Args a;
    a.x = (const bf16*)x;
    a.dy = (const bf16*)dy;
    a.part = (float*)ws;
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.lddy = g.lddy;
    a.tiles_d = g.Do / TDO; a.tiles_h = g.Ho / THO; a.tiles_w = g.Wo / tw;
    a.ntiles = g.N * a.tiles_d * a.tiles_h * a.tiles_w;
    a.G = s2_groups(g, a.ntiles);
    a.dy2 = (const bf16*)dy2;
    a.lddy2 = lddy2;
    a.part2 = (float*)ws + (size_t)a.G * 27 * g.Cin * g.Cout;
    const dim3 grid(a.G, (g.Cin / 32) * (g.Cout / (32 * nco)));
    if (s2_dma(g)) {
        if ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dy2)) % 16) return ru3d_fail(-1, "wgrad_s2: x / dy must be 16-byte aligned");
        if (dy2) {
            if (nco == 2) hipLaunchKernelGGL((wgrad3_s2_dma_kernel<2, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((wgrad3_s2_dma_kernel<1, true>), grid, dim3(256), 0, st, a);
        } else {
            if (nco == 2) hipLaunchKernelGGL((wgrad3_s2_dma_kernel<2, false>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((wgrad3_s2_dma_kernel<1, false>), grid, dim3(256), 0, st, a);
        }
    } else {
        hipLaunchKernelGGL(wgrad3_s2_tile_kernel<1>, grid, dim3(256), 0, st, a);
    }
    int rc = ru3d_check_launch("wgrad3_s2_tile");
    if (rc) return rc;
    rc = wgrad_reduce_launch((const float*)ws, dw, a.G, 27, g.Cin, g.Cout, g.s_o, g.s_i, st);
    if (rc || !dy2) return rc;
    return wgrad_reduce_launch(a.part2, dw2, a.G, 1, g.Cin, g.Cout, (int64_t)g.Cin, 1, st);
}

}  // namespace RU3D_NS
