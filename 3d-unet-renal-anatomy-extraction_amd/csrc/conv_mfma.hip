// bf16 MFMA implicit-GEMM convolution kernels for gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// GEMM view of a 3x3x3 stride-1 conv on NDHWC data:  D[cout][voxel] = sum_{tap, ci} W[tap][cout][ci] * X[voxel+tap][ci]
//   A operand (rows = cout)   : weights, pre-packed in exact fragment order (one contiguous 1 KiB load per
//                               wave-instruction, served from L1/L2 - every workgroup reads the same weights)
//   B operand (cols = voxel)  : activations; a workgroup stages the (TD+2)x(TH+2)x(TW+2) halo tile of a
//                               32-channel chunk in LDS once and every tap reads its fragments from there with
//                               ds_read_b128 (voxel pitch 80 B = 64 B data + 16 B pad: conflict-free for a
//                               32-voxel W-run)
//   C/D                       : lane = voxel (l & 31), registers = 16 couts in groups of 4 consecutive channels
//                               -> 8-byte NDHWC stores, bias / residual fused in the epilogue.
// Workgroup: 256 threads = 4 waves, output tile TDxTHxTW = 256 voxels (8 MFMA column tiles, 2 per wave) x NT*32
// couts.  2 workgroups per CU (LDS <= 65 KB) so one group's staging overlaps the other's MFMA phase.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {

#define MF_PITCH 40  // bf16 elements per staged voxel (32 data + 8 pad) = 80 bytes

// ---------------------------------------------------------------------------------------------------------
// Weight packing: dst[((tap * KS + ks) * NTT + nt) * 64 + lane][j] = W[tap][co = nt*32 + (lane&31)][ci = ks*16 + 8*(lane>>5) + j]
// (KS = cin/16 k-steps, NTT = cout/32).  src index = co*s_o + ci*s_i + tap.
__global__ void pack_mfma_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int cin, int cout, int taps,
                                 int64_t s_o, int64_t s_i) {
    const int KS = cin / 16, NTT = cout / 32;
    const int64_t total = (int64_t)taps * KS * NTT * 64 * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        const int lane = (int)((i >> 3) & 63);
        int64_t t = i >> 9;
        const int nt = (int)(t % NTT);
        t /= NTT;
        const int ks = (int)(t % KS);
        const int tap = (int)(t / KS);
        const int co = nt * 32 + (lane & 31);
        const int ci = ks * 16 + 8 * (lane >> 5) + j;
        dst[i] = (bf16)src[co * s_o + ci * s_i + tap];
    }
}

size_t mfma_packed_bytes(int cin, int cout, int taps) { return (size_t)taps * cin * cout * 2; }

int pack_mfma_launch(const float* src, void* dst, int cin, int cout, int taps, int64_t s_o, int64_t s_i, int flip,
                     hipStream_t st) {
    (void)flip;
    const int64_t total = (int64_t)taps * cin * cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, (bf16*)dst, cin, cout, taps, s_o, s_i);
    return ru3d_check_launch("pack_mfma");
}

// Which (shape, dtype) combinations take the MFMA path.  Must agree between ru3d_pack_weight and the launchers.
bool mfma_conv_eligible(int cin, int cout, int k, int dtype, int y_dtype) {
    return dtype == RU3D_BF16 && y_dtype == RU3D_BF16 && (k == 1 || k == 3) && (cin % 16) == 0 && (cout % 32) == 0;
}

// ---------------------------------------------------------------------------------------------------------
struct MfmaConvArgs {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    bf16* y;
    int N, D, H, W;        // stride 1, pad 1: input extents == output extents
    int Cin, Cout, ldx, ldy, ldr;
    int tiles_d, tiles_h, tiles_w;
    int flip;
    int nblk;              // spatial tiles * N (grid.x)
    float* stat_slab;      // optional: per-(workgroup, wave, n, cout) {sum, sum of squares} of the stored outputs
    float* part;           // split-K (deep levels): fp32 partial outputs [blockIdx.z][voxel][Cout], no bias/residual
    int ksplit;            // number of 32-channel chunk groups (gridDim.z); 1 = none
    void* ws;              // caller-owned workspace for the split-K partials (may be null: no split-K)
    size_t ws_bytes;
    int* defer_ks;         // host side: leave the split-K partials in ws un-summed and report their count here (0 = y is final)
};

// T1 (bijective): consecutive hardware block ids round-robin over the 8 XCDs; give each XCD a contiguous
// run of tiles so that halo re-reads between neighbouring tiles hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// MT = MFMA column tiles (32 voxels) per wave: 2 for the large levels (256-voxel tile), 1 for the deep,
// spatially small levels where more, smaller workgroups are needed to fill 256 CUs.
// WD = depth (in (tap, k-step) steps) of the register ring that prefetches weight fragments: the loads of step
// s + WD are issued right after the MFMAs of step s, so ~WD*MT*NT MFMAs (>= 500 cycles) cover an L2 hit.
template <int TD, int TH, int TW, int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv3_s1_mfma_kernel(MfmaConvArgs a) {
    constexpr int HD = TD + 2, HH = TH + 2, WW = TW + 2;
    constexpr int HV = HD * HH * WW;
    constexpr int WD = (MT * NT >= 4) ? 6 : 12;   // weight-fragment ring depth (steps)
    constexpr int XD = 3;                          // activation-fragment ring depth (steps)
    constexpr int S = 54;   // 27 taps x 2 k-steps per 32-channel chunk
    static_assert(TD * TH * TW == 128 * MT, "tile must hold 128*MT voxels");
    static_assert(HV * MF_PITCH * 2 <= 65536, "halo tile must leave room for 2 workgroups per CU");
    __shared__ __attribute__((aligned(16))) bf16 lds[HV * MF_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (round 4: a 1-D launch in which every XCD owned a fixed slice of the cout tiles - its L2 holding 1/8 of the weight,
    // every weight byte leaving HBM once - changed nothing: 28.2 vs 28.4 us at 512 -> 512 on 8^3, 37.9 vs 37.9 at 16^3,
    // profiles/r04_deep_level_convs.txt.  Where the weight bytes are served from is not what bounds these kernels; the
    // CUs' L1 path is: four waves x 1 KB of weight fragments per 2 MFMAs.)
    const int by = blockIdx.y, bz = blockIdx.z, gz = gridDim.z;
    int tile = xcd_remap(blockIdx.x, a.nblk);
    const int tw_i = tile % a.tiles_w;
    tile /= a.tiles_w;
    const int th_i = tile % a.tiles_h;
    tile /= a.tiles_h;
    const int td_i = tile % a.tiles_d;
    const int n = tile / a.tiles_d;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    const int co_blk = by * (NT * 32);
    const int NTT = a.Cout / 32, KS = a.Cin / 16;

    // this lane's output voxels (one per MFMA column tile owned by the wave)
    int hv0[MT], od[MT], oh[MT], ow[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int f = (wave * MT + m) * 32 + (lane & 31);
        const int tdl = f / (TH * TW), thl = (f / TW) % TH, twl = f % TW;
        od[m] = d0 + tdl;
        oh[m] = h0 + thl;
        ow[m] = w0 + twl;
        hv0[m] = ((tdl * HH + thl) * WW + twl) * MF_PITCH + (lane >> 5) * 8;  // + k-half of the fragment
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;

    const bf16x8* wbase = a.w + (int64_t)by * NT * 64 + lane;
    const int nchunks = a.Cin / 32;
    // split-K: blockIdx.z owns a contiguous run of 32-channel chunks and writes an fp32 partial
    const int ch_lo = (nchunks * bz) / gz, ch_hi = (nchunks * (bz + 1)) / gz;
    for (int ch = ch_lo; ch < ch_hi; ch++) {
        // ---- weight ring prologue (independent of LDS: its latency hides under the staging below)
        bf16x8 wq[WD][NT];
#pragma unroll
        for (int s = 0; s < WD; s++) {
            const int tap = s >> 1, kc = s & 1;
            const int wtap = a.flip ? 26 - tap : tap;
#pragma unroll
            for (int t = 0; t < NT; t++) wq[s][t] = wbase[((int64_t)(wtap * KS + ch * 2 + kc) * NTT + t) * 64];
        }
        // ---- stage the halo tile of channels [32 ch, 32 ch + 32): all loads first, then all LDS writes
        constexpr int NIT = (HV * 4 + 255) / 256;
        bf16x8 stage[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int c = tid + i * 256;
            const int hv = c >> 2, part = c & 3;
            const int zw = hv % WW, zh = (hv / WW) % HH, zd = hv / (WW * HH);
            const int gd = d0 + zd - 1, gh = h0 + zh - 1, gw = w0 + zw - 1;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < HV * 4 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)
                v = *reinterpret_cast<const bf16x8*>(a.x + ((((int64_t)n * a.D + gd) * a.H + gh) * a.W + gw) * a.ldx +
                                                     ch * 32 + part * 8);
            stage[i] = v;
        }
        if (ch > ch_lo) __syncthreads();  // everyone done reading the previous chunk
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int c = tid + i * 256;
            if (c < HV * 4) *reinterpret_cast<bf16x8*>(&lds[(c >> 2) * MF_PITCH + (c & 3) * 8]) = stage[i];
        }
        __syncthreads();
        // ---- 27 taps x 2 k-steps of 16 channels.  Software pipeline pinned with sched_barrier: the LDS
        // fragments of step s+1 and the weight fragments of step s+WD are issued before / right after the
        // MFMAs of step s (hipcc otherwise sinks every load to just before its use: vmcnt(1) per step).
        // activation fragments: ring of XD steps read ahead from LDS (ds_read_b128 latency under load is
        // well above one MFMA pair, so a 1-step lookahead still exposes it)
        bf16x8 xq[XD][MT];
#pragma unroll
        for (int s = 0; s < XD; s++) {
            const int tap0 = s >> 1, kc0 = s & 1;
            const int toff0 = (((tap0 / 9) * HH + (tap0 / 3) % 3) * WW + tap0 % 3) * MF_PITCH;
#pragma unroll
            for (int m = 0; m < MT; m++) xq[s][m] = *reinterpret_cast<const bf16x8*>(&lds[hv0[m] + toff0 + kc0 * 16]);
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int m = 0; m < MT; m++)
                    acc[m][t] = RU3D_MFMA_32X32X16(wq[s % WD][t], xq[s % XD][m], acc[m][t], 0, 0, 0);
            if (s + XD < S) {
                const int tap1 = (s + XD) >> 1, kc1 = (s + XD) & 1;
                const int toff1 = (((tap1 / 9) * HH + (tap1 / 3) % 3) * WW + tap1 % 3) * MF_PITCH;
#pragma unroll
                for (int m = 0; m < MT; m++)
                    xq[s % XD][m] = *reinterpret_cast<const bf16x8*>(&lds[hv0[m] + toff1 + kc1 * 16]);
            }
            if (s + WD < S) {
                const int tap2 = (s + WD) >> 1, kc2 = (s + WD) & 1;
                const int wtap2 = a.flip ? 26 - tap2 : tap2;
#pragma unroll
                for (int t = 0; t < NT; t++)
                    wq[s % WD][t] = wbase[((int64_t)(wtap2 * KS + ch * 2 + kc2) * NTT + t) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: lane = voxel, 16 couts per accumulator in 4 groups of 4 consecutive channels
#pragma unroll
    for (int m = 0; m < MT; m++) {
        if (od[m] >= a.D || oh[m] >= a.H || ow[m] >= a.W) continue;
        const int64_t vox = (((int64_t)n * a.D + od[m]) * a.H + oh[m]) * a.W + ow[m];
        if (a.part) {
            float* pp = a.part + ((int64_t)bz * ((int64_t)a.N * a.D * a.H * a.W) + vox) * a.Cout;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
                    *reinterpret_cast<f32x4*>(pp + co_blk + t * 32 + 8 * q + 4 * (lane >> 5)) = v;
                }
            continue;
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = co_blk + t * 32 + 8 * q + 4 * (lane >> 5);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
                if (a.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += b[i];
                }
                if (a.res) {
                    float r[4];
                    load_vec<bf16, 4>(a.res + vox * a.ldr + c0, r);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += r[i];
                }
                store_vec<bf16, 4>(a.y + vox * a.ldy + c0, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Producer/consumer form of the same conv for the large levels (>= 2 tiles per CU):
//   * persistent workgroup of 8 waves, one per CU: waves 0-3 CONSUME (MFMA loop + epilogue, weights through
//     their own vmcnt-tracked register ring), waves 4-7 PRODUCE (global -> registers -> LDS staging of the
//     NEXT (tile, 32-channel chunk) item into the other of two LDS halo buffers).  vmcnt is per wave and
//     in-order, so keeping the long-latency halo loads out of the consumer waves is what lets the weight ring
//     run with counted waits.  One s_barrier per item hands the buffers over.
//   * the producers' per-piece global / LDS offsets are tile-invariant and computed once; interior tiles
//     skip the bounds checks.
//   * work is dealt XCD-contiguously: the blocks that share an XCD walk a contiguous run of tiles, so the halo
//     re-reads of neighbouring tiles meet in that XCD's L2.
// MT = column tiles per consumer wave (tile = 128*MT voxels); the 512-voxel form is conv3_s1_pc4_kernel below.
template <int TD, int TH, int TW, int MT, int NT>
__global__ __launch_bounds__(512, 2) void conv3_s1_pc_kernel(MfmaConvArgs a) {
    constexpr int PITCH = MF_PITCH;
    constexpr int HD = TD + 2, HH = TH + 2, WW = TW + 2;
    constexpr int HV = HD * HH * WW;
    // ring depths: ~700 cycles of MFMA work between a weight load and its use, ~2 MFMA groups for LDS reads
    constexpr int WD = (768 + 32 * MT * NT - 1) / (32 * MT * NT);
    constexpr int XD = (MT >= 4) ? 2 : 3;
    constexpr int S = 54;
    constexpr int NIT = (HV * 4 + 255) / 256;
    static_assert(TD * TH * TW == 128 * MT, "tile must hold 128*MT voxels");
    static_assert((2 * HV * PITCH + 4 * MT * 32 * MF_PITCH) * 2 <= 160 * 1024, "two halo buffers must fit in LDS");
    __shared__ __attribute__((aligned(16))) bf16 lds[2 * HV * PITCH + 4 * MT * 32 * MF_PITCH];   // 2 halo buffers + epilogue patches
    // element offset of 16-byte piece `p` of halo voxel `hv`
    auto piece_off = [](int hv, int p) { return hv * MF_PITCH + (p << 3); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int nchunks = a.Cin / 32;

    // ---- this workgroup's tiles: XCD-contiguous run, strided by the number of workgroups on that XCD label
    const int G = gridDim.x, b = blockIdx.x;
    const int L = G < 8 ? G : 8;                             // XCD labels in use
    const int xcd = b % L, idx = b / L;
    const int gx = (G - xcd + L - 1) / L;                    // workgroups with this label
    const int t_begin = (int)(((int64_t)a.nblk * xcd) / L), t_end = (int)(((int64_t)a.nblk * (xcd + 1)) / L);
    const int my_tiles = (t_begin + idx < t_end) ? (t_end - t_begin - idx + gx - 1) / gx : 0;
    const int nitems = my_tiles * nchunks;

    auto tile_origin = [&](int tile, int& n, int& d0, int& h0, int& w0) {
        const int tw_i = tile % a.tiles_w;
        tile /= a.tiles_w;
        const int th_i = tile % a.tiles_h;
        tile /= a.tiles_h;
        const int td_i = tile % a.tiles_d;
        n = tile / a.tiles_d;
        d0 = td_i * TD;
        h0 = th_i * TH;
        w0 = tw_i * TW;
    };

    if (producer) {
        // ------------------------------------------------------------------ producer waves
        const int pt = tid - 256;
        const int part = pt & 3;                 // c = pt + 256 i: the 16-byte piece index is fixed per thread
        int rel[NIT], zz[NIT];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int c = pt + i * 256;
            const int hv = (c < HV * 4) ? (c >> 2) : 0;
            const int zw = hv % WW, zh = (hv / WW) % HH, zd = hv / (WW * HH);
            rel[i] = ((zd * a.H + zh) * a.W + zw) * a.ldx + part * 8;
            zz[i] = (c < HV * 4) ? (zd | (zh << 8) | (zw << 16)) : -1;
        }
        constexpr int HALF = (NIT + 1) / 2;       // two batches of loads keep the staging registers at 4*HALF
        for (int it = 0; it <= nitems; it++) {
            if (it < nitems) {
                const int tile = t_begin + idx + (it / nchunks) * gx, ch = it % nchunks;
                int n, d0, h0, w0;
                tile_origin(tile, n, d0, h0, w0);
                const bf16* src = a.x + ((((int64_t)n * a.D + (d0 - 1)) * a.H + (h0 - 1)) * a.W + (w0 - 1)) * a.ldx + ch * 32;
                const bool interior = d0 >= 1 && d0 + TD + 1 <= a.D && h0 >= 1 && h0 + TH + 1 <= a.H && w0 >= 1 &&
                                      w0 + TW + 1 <= a.W;
                bf16* dst = lds + (it & 1) * (HV * PITCH);
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {
                    bf16x8 stage[HALF];
#pragma unroll
                    for (int j = 0; j < HALF; j++) {
                        const int i = hb * HALF + j;
                        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (i < NIT) {
                            bool ok = zz[i] >= 0;
                            if (!interior) {
                                const int gd = d0 - 1 + (zz[i] & 255), gh = h0 - 1 + ((zz[i] >> 8) & 255),
                                          gw = w0 - 1 + ((zz[i] >> 16) & 255);
                                ok = ok && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
                            }
                            if (ok) v = *reinterpret_cast<const bf16x8*>(src + rel[i]);
                        }
                        stage[j] = v;
                    }
#pragma unroll
                    for (int j = 0; j < HALF; j++) {
                        const int i = hb * HALF + j;
                        if (i < NIT && zz[i] >= 0) {
                            const int hv = (((zz[i] & 255) * HH) + ((zz[i] >> 8) & 255)) * WW + ((zz[i] >> 16) & 255);
                            *reinterpret_cast<bf16x8*>(dst + piece_off(hv, part)) = stage[j];
                        }
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int co_blk = blockIdx.y * (NT * 32);
    int hvl[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int f = (wave * MT + m) * 32 + (lane & 31);
        hvl[m] = ((f / (TH * TW)) * HH + (f / TW) % TH) * WW + f % TW;   // halo voxel index of tap (0,0,0)
    }
    const bf16x8* wbase = a.w + (int64_t)blockIdx.y * NT * 64 + lane;
    f32x16 acc[MT][NT];
    // bias of this lane's 16 couts per cout tile, loaded once (the epilogue must not wait on global loads)
    f32x4 bq[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            bq[t][q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co_blk + t * 32 + 8 * q + 4 * (lane >> 5)) : z4;
        }
    // fused InstanceNorm statistics: running {sum, sum^2} of this wave's stored values for the sample `cur_n`
    float st1[NT][8], st2[NT][8];
    int cur_n = -1;
    if (a.stat_slab)
        ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave, a.N, NT * 32 * 2);
    auto stat_flush = [&]() {
        if (!a.stat_slab || cur_n < 0) return;
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        float* dst = a.stat_slab + ((((int64_t)blk * 4 + wave) * a.N + cur_n) * (NT * 32)) * 2;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                float s1 = st1[t][i], s2 = st2[t][i];
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {   // lanes that share (lane & 3) hold the same 8 channels
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if (lane < 4) {
                    const int c = t * 32 + lane * 8 + i;
                    dst[c * 2] = s1;
                    dst[c * 2 + 1] = s2;
                }
            }
    };

    for (int it = 0; it <= nitems; it++) {
        if (it >= 1) {
            const int item = it - 1;
            const int tile = t_begin + idx + (item / nchunks) * gx, ch = item % nchunks;
            const bf16* buf = lds + (item & 1) * (HV * PITCH);
            const int ph = lane >> 5;   // k-half of the fragment
            if (ch == 0) {
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;
            }
            bf16x8 wq[WD][NT];
#pragma unroll
            for (int s = 0; s < WD; s++) {
                const int tap = s >> 1, kc = s & 1;
                const int wtap = a.flip ? 26 - tap : tap;
#pragma unroll
                for (int t = 0; t < NT; t++) wq[s][t] = wbase[((int64_t)(wtap * KS + ch * 2 + kc) * NTT + t) * 64];
            }
            bf16x8 xq[XD][MT];
#pragma unroll
            for (int s = 0; s < XD; s++) {
                const int tap0 = s >> 1, kc0 = s & 1;
                const int toff0 = ((tap0 / 9) * HH + (tap0 / 3) % 3) * WW + tap0 % 3;
#pragma unroll
                for (int m = 0; m < MT; m++)
                    xq[s][m] = *reinterpret_cast<const bf16x8*>(&buf[piece_off(hvl[m] + toff0, kc0 * 2 + ph)]);
            }
#pragma unroll
            for (int s = 0; s < S; s++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        acc[m][t] = RU3D_MFMA_32X32X16(wq[s % WD][t], xq[s % XD][m], acc[m][t], 0, 0, 0);
                if (s + XD < S) {
                    const int tap1 = (s + XD) >> 1, kc1 = (s + XD) & 1;
                    const int toff1 = ((tap1 / 9) * HH + (tap1 / 3) % 3) * WW + tap1 % 3;
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        xq[s % XD][m] = *reinterpret_cast<const bf16x8*>(&buf[piece_off(hvl[m] + toff1, kc1 * 2 + ph)]);
                }
                if (s + WD < S) {
                    const int tap2 = (s + WD) >> 1, kc2 = (s + WD) & 1;
                    const int wtap2 = a.flip ? 26 - tap2 : tap2;
#pragma unroll
                    for (int t = 0; t < NT; t++)
                        wq[s % WD][t] = wbase[((int64_t)(wtap2 * KS + ch * 2 + kc2) * NTT + t) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ch == nchunks - 1) {
                // ---- epilogue through a wave-private LDS patch: the accumulator layout (lane = voxel, 4 couts
                // per 8 bytes) would give 8-byte stores scattered over 32 rows per instruction (store-issue
                // bound); transposed through LDS every lane stores 16 bytes and a wave-instruction covers 16
                // whole 64-byte channel rows.  Bias is added before, the residual after the transpose.
                int n, d0, h0, w0;
                tile_origin(tile, n, d0, h0, w0);
                if (a.stat_slab && n != cur_n) {
                    stat_flush();
                    cur_n = n;
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int i = 0; i < 8; i++) st1[t][i] = st2[t][i] = 0.f;
                }
                bf16* est = lds + 2 * HV * PITCH + wave * (MT * 32 * MF_PITCH);
#pragma unroll
                for (int t = 0; t < NT; t++) {
#pragma unroll
                    for (int m = 0; m < MT; m++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int cl = 8 * q + 4 * (lane >> 5);
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
#pragma unroll
                            for (int i = 0; i < 4; i++) v[i] += bq[t][q][i];
                            store_vec<bf16, 4>(est + (m * 32 + (lane & 31)) * MF_PITCH + cl, v);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 2 * MT; r++) {
                        const int row = (lane >> 2) + 16 * r, part = lane & 3;
                        const int f = wave * (MT * 32) + row;
                        const int od = d0 + f / (TH * TW), oh = h0 + (f / TW) % TH, ow = w0 + f % TW;
                        float v[8];
                        load_vec<bf16, 8>(est + row * MF_PITCH + part * 8, v);
                        if (od < a.D && oh < a.H && ow < a.W) {
                            const int64_t vox = (((int64_t)n * a.D + od) * a.H + oh) * a.W + ow;
                            const int c0 = co_blk + t * 32 + part * 8;
                            if (a.stat_slab) {
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                    st1[t][i] += v[i];
                                    st2[t][i] = fmaf(v[i], v[i], st2[t][i]);
                                }
                            }
                            if (a.res) {
                                float rr[8];
                                load_vec<bf16, 8>(a.res + vox * a.ldr + c0, rr);
#pragma unroll
                                for (int i = 0; i < 8; i++) v[i] += rr[i];
                            }
                            store_vec<bf16, 8>(a.y + vox * a.ldy + c0, v);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    stat_flush();
}

// ---------------------------------------------------------------------------------------------------------
// pc4: the producer/consumer kernel with 4 voxel tiles x 2 cout tiles of accumulators per consumer wave.
// Why (profiles/r04_pc_ablation.txt): with 2 x 2 tiles per wave every v_mfma_32x32x16 needs 512 B of weights through
// the CU's vector L1 (64 B/clk) and 512 B of activations out of LDS; four consumer waves at the MFMA rate ask the L1
// for exactly its peak, so the weight loads, the producers' halo loads and the MFMAs end up in series: 71 us where
// the MFMAs alone need 23-27 us (weight reloads removed: -15 us, producers' global loads removed: -12 us).  Here
//   * a consumer wave owns a 128-voxel x 64-cout block: 8 accumulators (128 AGPRs), a weight fragment serves four
//     MFMAs: half the L1 bytes per MFMA, the same LDS bytes;
//   * channel chunks of 16 (one MFMA k-step, 27 steps per item) so that two 512-voxel halo buffers fit: 32-byte voxel
//     slots, rows padded from 34 to 40 slots so that the slot's 16-byte halves can be XOR-swapped by bit 2 of the W
//     position alone (a ds_read_b128 of 32 consecutive voxels then touches every bank equally, for every tap offset)
//     and every fragment address is one of three per-lane bases + an immediate;
//   * the halo is moved by LDS-DMA (no staging registers, out-of-volume pieces come back as zeros through the buffer
//     range check), issued by the producer waves one item ahead; producers are the only waves that wait for it;
//   * the weight ring runs across item boundaries (27 = 9 x 3 steps, ring depth 3), so a new item starts with its
//     first fragments already in registers;
//   * barriers wait for LDS traffic only (lgkmcnt): a consumer's weight loads stay in flight across them;
//   * InstanceNorm statistics are kept in LDS between tiles (wave-private rows), not in registers.
// Tile: 4 x 4 x 32 voxels (wave = D plane, accumulator m = H row, lane = W position).
template <int WD>
__global__ __launch_bounds__(512, 2) void conv3_s1_pc4_kernel(MfmaConvArgs a) {
    constexpr int TD = 4, TH = 4, TW = 32, MT = 4, NT = 2;
    constexpr int HH = TH + 2, HD = TD + 2, WW = TW + 2, WWP = 40;
    constexpr int SLOTS = HD * HH * WWP;              // 32-byte voxel slots of one halo buffer (pads included)
    constexpr int BUF = SLOTS * 16;                   // bf16 elements
    constexpr int NCH = (SLOTS * 2) / 64;             // 1 KB DMA chunks per buffer
    constexpr int NDMA = (NCH + 3) / 4;               // per producer wave
    constexpr int S = 27, XD = 2;
    constexpr int OOBV = (int)0x80000000;
    static_assert((SLOTS * 2) % 64 == 0 && S % WD == 0, "whole DMA chunks; the ring must close over an item");
    __shared__ __attribute__((aligned(1024))) bf16 lds[2 * BUF + 4 * MT * 32 * MF_PITCH];
    __shared__ float statl[4][NT * 32 * 2];
    __shared__ __attribute__((aligned(16))) float biasl[NT * 32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int nchunks = KS;

    const int G = gridDim.x, b = blockIdx.x;
    const int L = G < 8 ? G : 8;
    const int xcd = b % L, idx = b / L;
    const int gx = (G - xcd + L - 1) / L;
    const int t_begin = (int)(((int64_t)a.nblk * xcd) / L), t_end = (int)(((int64_t)a.nblk * (xcd + 1)) / L);
    const int my_tiles = (t_begin + idx < t_end) ? (t_end - t_begin - idx + gx - 1) / gx : 0;
    const int nitems = my_tiles * nchunks;

    auto tile_origin = [&](int tile, int& n, int& d0, int& h0, int& w0) {
        const int tw_i = tile % a.tiles_w;
        tile /= a.tiles_w;
        const int th_i = tile % a.tiles_h;
        tile /= a.tiles_h;
        const int td_i = tile % a.tiles_d;
        n = tile / a.tiles_d;
        d0 = td_i * TD;
        h0 = th_i * TH;
        w0 = tw_i * TW;
    };

    if (producer) {
        // ------------------------------------------------------------------ producer waves: LDS-DMA of the next item
        const int pw = wave - 4;
        int rel[NDMA], zz[NDMA];
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const int piece = (i * 4 + pw) * 64 + lane;            // 16-byte piece of the buffer this lane fills
            const int slot = piece >> 1;
            const int zw = slot % WWP, zh = (slot / WWP) % HH, zd = slot / (WWP * HH);
            const bool valid = (i * 4 + pw) < NCH && zw < WW;
            const int part = (piece & 1) ^ ((zw >> 2) & 1);        // which half of the voxel's 32 bytes lives here
            rel[i] = valid ? (((zd * a.H + zh) * a.W + zw) * a.ldx + part * 8) * 2 : OOBV;
            zz[i] = valid ? (zd | (zh << 8) | (zw << 16)) : -1;
        }
        const int64_t sample_elems = (int64_t)a.D * a.H * a.W * a.ldx;
        for (int it = 0; it <= nitems; it++) {
            if (it < nitems) {
                const int tile = t_begin + idx + (it / nchunks) * gx, ch = it % nchunks;
                int n, d0, h0, w0;
                tile_origin(tile, n, d0, h0, w0);
                const ru3d_i32x4 rsrc = ru3d_buffer_rsrc(a.x + n * sample_elems, (int)(sample_elems * 2));
                const int base = ((((d0 - 1) * a.H + (h0 - 1)) * a.W + (w0 - 1)) * a.ldx + ch * 16) * 2;
                const bool interior = d0 >= 1 && d0 + TD + 1 <= a.D && h0 >= 1 && h0 + TH + 1 <= a.H && w0 >= 1 &&
                                      w0 + TW + 1 <= a.W;
                const bf16* dst = lds + (it & 1) * BUF;
#pragma unroll
                for (int i = 0; i < NDMA; i++) {
                    if ((i * 4 + pw) < NCH) {
                        int off = zz[i] >= 0 ? base + rel[i] : OOBV;
                        if (!interior && zz[i] >= 0) {
                            const int gd = d0 - 1 + (zz[i] & 255), gh = h0 - 1 + ((zz[i] >> 8) & 255),
                                      gw = w0 - 1 + ((zz[i] >> 16) & 255);
                            if (!(gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)) off = OOBV;
                        }
                        ru3d_lds_dma16(rsrc, dst + (i * 4 + pw) * 512, off);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int co_blk = blockIdx.y * (NT * 32);
    const int ph = lane >> 5, wl = lane & 31;
    int xb[3];      // element offset of this lane's fragment piece for the three W offsets of a tap (wave's D plane)
#pragma unroll
    for (int tw = 0; tw < 3; tw++) {
        const int zw = wl + tw;
        xb[tw] = ((wave * HH) * WWP + zw) * 16 + ((ph ^ ((zw >> 2) & 1)) << 3);
    }
    const bf16x8* wbase = a.w + (int64_t)blockIdx.y * NT * 64 + lane;
    if (wave == 0) biasl[lane] = a.bias ? a.bias[co_blk + lane] : 0.f;
    statl[wave][lane] = 0.f;
    statl[wave][64 + lane] = 0.f;
    int cur_n = -1;
    if (a.stat_slab)
        ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave, a.N, NT * 32 * 2);
    auto stat_flush = [&]() {       // this wave's LDS sums -> its slab row of sample cur_n
        if (!a.stat_slab || cur_n < 0) return;
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        float* dst = a.stat_slab + ((((int64_t)blk * 4 + wave) * a.N + cur_n) * (NT * 32)) * 2;
        dst[lane] = statl[wave][lane];
        dst[64 + lane] = statl[wave][64 + lane];
        statl[wave][lane] = 0.f;
        statl[wave][64 + lane] = 0.f;
    };

    f32x16 acc[MT][NT];
    bf16x8 wq[WD][NT];
#pragma unroll
    for (int s = 0; s < WD; s++) {      // the ring's first fragments: (tap s, chunk 0)
        const int wtap = a.flip ? 26 - s : s;
#pragma unroll
        for (int t = 0; t < NT; t++) wq[s][t] = wbase[((int64_t)(wtap * KS) * NTT + t) * 64];
    }

    for (int it = 0; it <= nitems; it++) {
        if (it >= 1) {
            const int item = it - 1;
            const int tile = t_begin + idx + (item / nchunks) * gx, ch = item % nchunks;
            const int chn = ch + 1 == nchunks ? 0 : ch + 1;
            const bf16* buf = lds + (item & 1) * BUF;
            if (ch == 0) {
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;
            }
            bf16x8 xq[XD][MT];
#pragma unroll
            for (int s = 0; s < XD; s++)
#pragma unroll
                for (int m = 0; m < MT; m++)
                    xq[s][m] = *reinterpret_cast<const bf16x8*>(&buf[xb[s % 3] + (((s / 9) * HH + m + (s / 3) % 3) * WWP) * 16]);
#pragma unroll
            for (int s = 0; s < S; s++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        acc[m][t] = RU3D_MFMA_32X32X16(wq[s % WD][t], xq[s % XD][m], acc[m][t], 0, 0, 0);
                if (s + XD < S) {
                    const int s1 = s + XD;
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        xq[s % XD][m] =
                            *reinterpret_cast<const bf16x8*>(&buf[xb[s1 % 3] + (((s1 / 9) * HH + m + (s1 / 3) % 3) * WWP) * 16]);
                }
                {   // the ring never stops: the last WD steps fetch the first fragments of the next item's chunk
                    const int s2 = s + WD < S ? s + WD : s + WD - S;
                    const int c2 = s + WD < S ? ch : chn;
                    const int wtap2 = a.flip ? 26 - s2 : s2;
#pragma unroll
                    for (int t = 0; t < NT; t++) wq[s % WD][t] = wbase[((int64_t)(wtap2 * KS + c2) * NTT + t) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ch == nchunks - 1) {
                // ---- epilogue through a wave-private LDS patch (see conv3_s1_pc_kernel): bias before the transpose,
                // statistics and residual after it, 16-byte stores that cover whole 64-byte channel rows
                int n, d0, h0, w0;
                tile_origin(tile, n, d0, h0, w0);
                if (a.stat_slab && n != cur_n) {
                    stat_flush();
                    cur_n = n;
                }
                bf16* est = lds + 2 * BUF + wave * (MT * 32 * MF_PITCH);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    f32x4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4*>(&biasl[t * 32 + 8 * q + 4 * ph]);
#pragma unroll
                    for (int m = 0; m < MT; m++) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i] + bq[q][i];
                            store_vec<bf16, 4>(est + (m * 32 + wl) * MF_PITCH + 8 * q + 4 * ph, v);
                        }
                    }
                    float st1[8], st2[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
                    const int part = lane & 3;
#pragma unroll
                    for (int r = 0; r < 2 * MT; r++) {
                        const int row = (lane >> 2) + 16 * r;
                        const int od = d0 + wave, oh = h0 + (row >> 5), ow = w0 + (row & 31);
                        float v[8];
                        load_vec<bf16, 8>(est + row * MF_PITCH + part * 8, v);
                        if (od < a.D && oh < a.H && ow < a.W) {
                            const int64_t vox = (((int64_t)n * a.D + od) * a.H + oh) * a.W + ow;
                            const int c0 = co_blk + t * 32 + part * 8;
                            if (a.stat_slab) {
#pragma unroll
                                for (int i = 0; i < 8; i++) {
                                    st1[i] += v[i];
                                    st2[i] = fmaf(v[i], v[i], st2[i]);
                                }
                            }
                            if (a.res) {
                                float rr[8];
                                load_vec<bf16, 8>(a.res + vox * a.ldr + c0, rr);
#pragma unroll
                                for (int i = 0; i < 8; i++) v[i] += rr[i];
                            }
                            store_vec<bf16, 8>(a.y + vox * a.ldy + c0, v);
                        }
                    }
                    if (a.stat_slab) {
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            float s1 = st1[i], s2 = st2[i];
#pragma unroll
                            for (int o = 4; o < 64; o <<= 1) {   // lanes that share (lane & 3) hold the same 8 channels
                                s1 += __shfl_xor(s1, o, 64);
                                s2 += __shfl_xor(s2, o, 64);
                            }
                            if (lane < 4) {
                                const int c = t * 32 + lane * 8 + i;
                                statl[wave][c * 2] += s1;
                                statl[wave][c * 2 + 1] += s2;
                            }
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    stat_flush();
}


// ---------------------------------------------------------------------------------------------------------
// sk: the deep levels (16^3 x 256, 8^3 x 512 channels) with the accumulator blocking of pc4 and the reduction dimension
// split over the workgroup's four consumer waves.  The 256-voxel x 32-cout tiles of conv3_s1_mfma_kernel give a wave 1 x 2
// accumulators: 1 KB of activations from LDS and 512 B of weights through the L1 per MFMA, both pipes at their peak
// (profiles/r04_pmc_deep_levels.txt).  4 x 2 accumulators halve both, but a 128-voxel x 64-cout block per WAVE leaves only
// 64 blocks at 16^3 x 256: so all four waves of a workgroup work on the SAME 128 x 64 block, each on its own quarter of
// the input channels (private halo buffers, 16-channel chunks moved by the producer waves' LDS-DMA one chunk ahead,
// weight ring across chunks as in pc4); at the end the waves exchange accumulators through LDS - wave w keeps voxel tile
// w and receives the three other waves' partial sums for it, added in a fixed order - and each runs the epilogue of its
// 32 voxels.  gridDim.z > 1 splits the channels over workgroups as well (8^3: 4 x 4 = 16 slices of 32 channels): fp32
// partial slices in a.part, summed by the caller's next kernel or conv_ksplit_reduce_kernel.
// A wave numbers its accumulators from its own tile on (local m <-> voxel tile (m + wave) & 3): every register index is
// static.  Tile: 2 x (64 / TW) x TW voxels; a fragment's 32 voxels are 32 / TW rows of TW.
template <int TW>
__global__ __launch_bounds__(512, 2) void conv3_s1_sk_kernel(MfmaConvArgs a) {
    constexpr int TD = 2, TH = 64 / TW, R = 32 / TW, MT = 4, NT = 2, WD = 3;
    constexpr int HD = TD + 2, HH = TH + 2, WW = TW + 2;
    constexpr int SLOTS = HD * HH * WW;                  // real 32-byte voxel slots of one wave's halo
    constexpr int NCH = (SLOTS * 2 + 63) / 64;           // 1 KB DMA chunks per halo
    constexpr int BUF = NCH * 512;                       // bf16 elements per halo buffer (whole chunks)
    constexpr int S = 27, XD = 2;
    constexpr int OOBV = (int)0x80000000;
    constexpr int REDF = 4 * 3 * NT * 4 * 64 * 4;        // floats of the accumulator exchange
    static_assert(2 * 4 * BUF * 2 >= REDF * 4, "the exchange lives in the halo buffers");
    __shared__ __attribute__((aligned(1024))) bf16 lds[2 * 4 * BUF];
    __shared__ __attribute__((aligned(16))) bf16 patch[4][32 * MF_PITCH];
    __shared__ __attribute__((aligned(16))) float biasl[NT * 32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int cw = wave & 3;                              // consumer index (a producer feeds consumer cw)
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int cgs = a.Cout / 64;
    const int bz = blockIdx.z, gz = gridDim.z;
    const int cin_wave = a.Cin / (4 * gz);                // input channels of one wave
    const int nch = cin_wave / 16;
    const int k0 = (bz * 4 + cw) * nch;                   // first 16-channel chunk of this wave

    // units in [cout group][tile] order, a contiguous run per XCD: the workgroups of an XCD share their weights
    const int unit = xcd_remap(blockIdx.x, a.nblk * cgs);
    const int cg = unit / a.nblk;
    int tile = unit % a.nblk;
    const int tw_i = tile % a.tiles_w;
    tile /= a.tiles_w;
    const int th_i = tile % a.tiles_h;
    tile /= a.tiles_h;
    const int td_i = tile % a.tiles_d;
    const int n = tile / a.tiles_d;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;

    if (producer) {
        int rel[NCH], zz[NCH];
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int piece = i * 64 + lane;
            const int slot = piece >> 1;
            const int zw = slot % WW, zh = (slot / WW) % HH, zd = slot / (WW * HH);
            const bool valid = slot < SLOTS;
            const int part = (piece & 1) ^ ((zw >> 2) & 1);
            rel[i] = valid ? (((zd * a.H + zh) * a.W + zw) * a.ldx + part * 8) * 2 : OOBV;
            zz[i] = valid ? (zd | (zh << 8) | (zw << 16)) : -1;
        }
        const int64_t sample_elems = (int64_t)a.D * a.H * a.W * a.ldx;
        const ru3d_i32x4 rsrc = ru3d_buffer_rsrc(a.x + n * sample_elems, (int)(sample_elems * 2));
        const bool interior = d0 >= 1 && d0 + TD + 1 <= a.D && h0 >= 1 && h0 + TH + 1 <= a.H && w0 >= 1 && w0 + TW + 1 <= a.W;
        for (int it = 0; it <= nch; it++) {
            if (it < nch) {
                const int base = ((((d0 - 1) * a.H + (h0 - 1)) * a.W + (w0 - 1)) * a.ldx + (k0 + it) * 16) * 2;
                const bf16* dst = lds + ((it & 1) * 4 + cw) * BUF;
#pragma unroll
                for (int i = 0; i < NCH; i++) {
                    int off = zz[i] >= 0 ? base + rel[i] : OOBV;
                    if (!interior && zz[i] >= 0) {
                        const int gd = d0 - 1 + (zz[i] & 255), gh = h0 - 1 + ((zz[i] >> 8) & 255),
                                  gw = w0 - 1 + ((zz[i] >> 16) & 255);
                        if (!(gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)) off = OOBV;
                    }
                    ru3d_lds_dma16(rsrc, dst + i * 512, off);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int co_blk = cg * (NT * 32);
    const int ph = lane >> 5, wl = lane & 31;
    const int rl = wl / TW, wv = wl % TW;
    int xbm[MT][3];     // element offset (inside a buffer pair half) of this lane's piece: local tile m, W offset of the tap
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int gm = (m + wave) & 3;
#pragma unroll
        for (int tw = 0; tw < 3; tw++) {
            const int zw = wv + tw;
            xbm[m][tw] = ((((gm >> 1) * HH + (gm & 1) * R + rl) * WW) + zw) * 16 + ((ph ^ ((zw >> 2) & 1)) << 3) + wave * BUF;
        }
    }
    const bf16x8* wbase = a.w + (int64_t)cg * NT * 64 + lane;
    if (wave == 0) biasl[lane] = (a.bias && !a.part) ? a.bias[co_blk + lane] : 0.f;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;
    bf16x8 wq[WD][NT];
#pragma unroll
    for (int s = 0; s < WD; s++) {
        const int wtap = a.flip ? 26 - s : s;
#pragma unroll
        for (int t = 0; t < NT; t++) wq[s][t] = wbase[((int64_t)(wtap * KS + k0) * NTT + t) * 64];
    }

    for (int it = 0; it <= nch; it++) {
        if (it >= 1) {
            const int ch = it - 1;
            const int chn = ch + 1 == nch ? ch : ch + 1;      // the last chunk's tail fetches stay inside the slice
            const bf16* buf = lds + ((ch & 1) * 4) * BUF;
            bf16x8 xq[XD][MT];
#pragma unroll
            for (int s = 0; s < XD; s++)
#pragma unroll
                for (int m = 0; m < MT; m++)
                    xq[s][m] = *reinterpret_cast<const bf16x8*>(&buf[xbm[m][s % 3] + (((s / 9) * HH + (s / 3) % 3) * WW) * 16]);
#pragma unroll
            for (int s = 0; s < S; s++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        acc[m][t] = RU3D_MFMA_32X32X16(wq[s % WD][t], xq[s % XD][m], acc[m][t], 0, 0, 0);
                if (s + XD < S) {
                    const int s1 = s + XD;
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        xq[s % XD][m] =
                            *reinterpret_cast<const bf16x8*>(&buf[xbm[m][s1 % 3] + (((s1 / 9) * HH + (s1 / 3) % 3) * WW) * 16]);
                }
                {
                    const int s2 = s + WD < S ? s + WD : s + WD - S;
                    const int c2 = s + WD < S ? ch : chn;
                    const int wtap2 = a.flip ? 26 - s2 : s2;
#pragma unroll
                    for (int t = 0; t < NT; t++) wq[s % WD][t] = wbase[((int64_t)(wtap2 * KS + k0 + c2) * NTT + t) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- exchange: every halo has been read (the barrier above), the buffers now carry accumulators.
    // float4 slot [owner tile][source 0..2][t][q][lane]; a wave sends local tiles 1..3, which are tiles (m + wave) & 3
    f32x4* red = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int m = 1; m < MT; m++) {
        const int owner = (m + wave) & 3;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
                red[(((owner * 3 + (m - 1)) * NT + t) * 4 + q) * 64 + lane] = v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < 3; sl++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 v = red[(((wave * 3 + sl) * NT + t) * 4 + q) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; i++) acc[0][t][q * 4 + i] += v[i];
            }

    // ---- epilogue of this wave's 32 voxels (tile `wave`)
    const int od_w = d0 + (wave >> 1), hrow = h0 + (wave & 1) * R;
    if (a.part) {
        const int od = od_w, oh = hrow + rl, ow = w0 + wv;
        if (od < a.D && oh < a.H && ow < a.W) {
            const int64_t vox = (((int64_t)n * a.D + od) * a.H + oh) * a.W + ow;
            float* pp = a.part + ((int64_t)bz * ((int64_t)a.N * a.D * a.H * a.W) + vox) * a.Cout;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = acc[0][t][q * 4 + i];
                    *reinterpret_cast<f32x4*>(pp + co_blk + t * 32 + 8 * q + 4 * ph) = v;
                }
        }
        return;
    }
    bf16* est = patch[wave];
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&biasl[t * 32 + 8 * q + 4 * ph]);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = acc[0][t][q * 4 + i] + bq[i];
            store_vec<bf16, 4>(est + wl * MF_PITCH + 8 * q + 4 * ph, v);
        }
        const int part = lane & 3;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int row = (lane >> 2) + 16 * r;
            const int od = od_w, oh = hrow + row / TW, ow = w0 + row % TW;
            float v[8];
            load_vec<bf16, 8>(est + row * MF_PITCH + part * 8, v);
            if (od < a.D && oh < a.H && ow < a.W) {
                const int64_t vox = (((int64_t)n * a.D + od) * a.H + oh) * a.W + ow;
                const int c0 = co_blk + t * 32 + part * 8;
                if (a.res) {
                    float rr[8];
                    load_vec<bf16, 8>(a.res + vox * a.ldr + c0, rr);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] += rr[i];
                }
                store_vec<bf16, 8>(a.y + vox * a.ldy + c0, v);
            }
        }
    }
}


static void pc_grid(int cout, bool nt2, int64_t nblk, int* gx, int* gy) {
    *gy = cout / (nt2 ? 64 : 32);
    int g = ru3d_get_cu_budget() / *gy;   // one persistent workgroup per CU in total (N > 1: minus the CUs left to RCCL)
    g = g < 8 ? 8 : (g / 8) * 8;        // multiple of 8 so that blockIdx.x % 8 is the XCD label for every y
    if (g > nblk) g = (int)nblk;
    *gx = g;
}

template <int TD, int TH, int TW, int MT>
static int launch_s1_pc(const MfmaConvArgs& a0, bool nt2, hipStream_t st) {
    MfmaConvArgs a = a0;
    a.tiles_d = (a.D + TD - 1) / TD;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.tiles_w = (a.W + TW - 1) / TW;
    const int64_t nblk = (int64_t)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    if (nblk > 0x7fffffff) return ru3d_fail(-1, "conv_mfma: grid too large");
    a.nblk = (int)nblk;
    int gx, gy;
    pc_grid(a.Cout, nt2, nblk, &gx, &gy);
    dim3 grid(gx, gy);
    if (nt2)
        hipLaunchKernelGGL((conv3_s1_pc_kernel<TD, TH, TW, MT, 2>), grid, dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((conv3_s1_pc_kernel<TD, TH, TW, MT, 1>), grid, dim3(512), 0, st, a);
    return ru3d_check_launch("conv3_s1_pc");
}

static int launch_s1_pc4(const MfmaConvArgs& a0, hipStream_t st) {
    MfmaConvArgs a = a0;
    a.tiles_d = (a.D + 3) / 4;
    a.tiles_h = (a.H + 3) / 4;
    a.tiles_w = (a.W + 31) / 32;
    const int64_t nblk = (int64_t)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    if (nblk > 0x7fffffff) return ru3d_fail(-1, "conv_mfma: grid too large");
    a.nblk = (int)nblk;
    int gx, gy;
    pc_grid(a.Cout, true, nblk, &gx, &gy);
    hipLaunchKernelGGL((conv3_s1_pc4_kernel<3>), dim3(gx, gy), dim3(512), 0, st, a);
    return ru3d_check_launch("conv3_s1_pc4");
}

// y = bf16(sum_z part[z] + bias) (+ residual): fixed summation order, 8 channels (16 B) per thread
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(const float* __restrict__ part, int ksplit, int64_t V,
                                                                 int Cout, const float* __restrict__ bias,
                                                                 const bf16* __restrict__ res, int ldr,
                                                                 bf16* __restrict__ y, int ldy) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cg = Cout / 8;
    if (g >= V * cg) return;
    const int64_t vox = g / cg;
    const int c0 = (int)(g % cg) * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = bias ? bias[c0 + i] : 0.f;
    for (int z = 0; z < ksplit; z++) {
        const float* p = part + ((int64_t)z * V + vox) * Cout + c0;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] += lo[i];
            v[4 + i] += hi[i];
        }
    }
    if (res) {
        float r[8];
        load_vec<bf16, 8>(res + vox * ldr + c0, r);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (float)(bf16)v[i] + r[i];   // same two roundings as the fused epilogues
    }
    store_vec<bf16, 8>(y + vox * ldy + c0, v);
}

template <int TD, int TH, int TW, int MT>
static int launch_s1(const MfmaConvArgs& a0, bool nt2, hipStream_t st) {
    MfmaConvArgs a = a0;
    a.tiles_d = (a.D + TD - 1) / TD;
    a.tiles_h = (a.H + TH - 1) / TH;
    a.tiles_w = (a.W + TW - 1) / TW;
    const int64_t nblk = (int64_t)a.N * a.tiles_d * a.tiles_h * a.tiles_w;
    if (nblk > 0x7fffffff) return ru3d_fail(-1, "conv_mfma: grid too large");
    a.nblk = (int)nblk;
    const int ks = a.part ? a.ksplit : 1;
    dim3 grid((unsigned)nblk, a.Cout / (nt2 ? 64 : 32), ks);
    if (nt2) {
        hipLaunchKernelGGL((conv3_s1_mfma_kernel<TD, TH, TW, MT, 2>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((conv3_s1_mfma_kernel<TD, TH, TW, MT, 1>), grid, dim3(256), 0, st, a);
    }
    int rc = ru3d_check_launch("conv3_s1_mfma");
    if (rc || !a.part) return rc;
    if (a.defer_ks) {        // the caller's next kernel sums the slices (norm_small.hip)
        *a.defer_ks = ks;
        return 0;
    }
    const int64_t V = (int64_t)a.N * a.D * a.H * a.W;
    const int64_t groups = V * (a.Cout / 8);
    hipLaunchKernelGGL(conv_ksplit_reduce_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                       (const float*)a.part, ks, V, a.Cout, a.bias, a.res, a.ldr, a.y, a.ldy);
    return ru3d_check_launch("conv_ksplit_reduce");
}

// Tile choice: W decides the tile's aspect; if the 256-voxel x 64-cout decomposition yields fewer than ~2
// workgroups per CU (deep levels: 16^3, 8^3), fall back to 128-voxel tiles and then to 32-cout slices.
// >= 2 (tile, cout-slice) units per CU run on the persistent producer/consumer kernel.
struct S1Plan {
    int wclass;
    bool nt2, small, pc;
    bool pc4;          // 512-voxel tiles, 4 x 2 accumulators per wave (conv3_s1_pc4_kernel)
    int64_t nblk_pc;   // spatial tiles of the producer/consumer decomposition
};

static S1Plan s1_plan(int N, int D, int H, int W, int Cout, int ldx) {
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    S1Plan p;
    const bool can_nt2 = (Cout % 64) == 0;
    p.wclass = W >= 24 ? 32 : (W >= 12 ? 16 : 8);
    // extents that fit none of the tiles (config 4: 160 x 160 x 80 -> 40 x 40 x 20 on the 128-channel level): the tile
    // with the least padded volume (RU3D_CONV_TILEFIT=0: the width classes above)
    static const int fit_mode = getenv("RU3D_CONV_TILEFIT") ? atoi(getenv("RU3D_CONV_TILEFIT")) : 1;
    if (fit_mode) {
        const int64_t v32 = (int64_t)cdiv(D, 2) * 2 * cdiv(H, 4) * 4 * cdiv(W, 32) * 32;
        const int64_t v16 = (int64_t)cdiv(D, 2) * 2 * cdiv(H, 8) * 8 * cdiv(W, 16) * 16;
        const int64_t v8 = (int64_t)cdiv(D, 4) * 4 * cdiv(H, 8) * 8 * cdiv(W, 8) * 8;
        const int64_t cur = p.wclass == 32 ? v32 : (p.wclass == 16 ? v16 : v8);
        // a narrower tile has more halo per voxel: it must save an eighth of the padded volume to be chosen
        if (p.wclass == 32 && v16 * 8 < cur * 7 && v16 <= v8) p.wclass = 16;
        else if (p.wclass >= 16 && v8 * 8 < cur * 7) p.wclass = 8;
    }
    int64_t big;   // workgroups with MT = 2, widest cout slice
    if (p.wclass == 32) big = (int64_t)N * cdiv(D, 2) * cdiv(H, 4) * cdiv(W, 32);
    else if (p.wclass == 16) big = (int64_t)N * cdiv(D, 2) * cdiv(H, 8) * cdiv(W, 16);
    else big = (int64_t)N * cdiv(D, 4) * cdiv(H, 8) * cdiv(W, 8);
    p.nblk_pc = big;
    big *= Cout / (can_nt2 ? 64 : 32);
    p.small = big < 512;
    p.nt2 = can_nt2 && (!p.small || big * 2 >= 1024);
    static const int pc_mode = getenv("RU3D_CONV_PC") ? atoi(getenv("RU3D_CONV_PC")) : 1;   // 0 = off
    p.pc = !p.small && pc_mode >= 1 && (p.wclass >= 16 || fit_mode);
    // the 512-voxel form: no more padded volume than the 2 x 4 x 32 tile, at least 3/4 of the CUs busy, a sample that
    // a buffer descriptor can address (RU3D_CONV_PC4=0: off)
    static const int pc4_mode = getenv("RU3D_CONV_PC4") ? atoi(getenv("RU3D_CONV_PC4")) : 1;
    p.pc4 = false;
    if (p.pc && p.nt2 && p.wclass == 32 && pc4_mode && (D % 4) == 0 && (H % 4) == 0 && ldx > 0 && (ldx % 8) == 0) {
        const int64_t n4 = (int64_t)N * (D / 4) * (H / 4) * cdiv(W, 32);
        const int64_t sample_bytes = (int64_t)D * H * W * ldx * 2;
        if (n4 * (Cout / 64) * 4 >= (int64_t)ru3d_get_cu_budget() * 3 && sample_bytes < (1ll << 31)) {
            p.pc4 = true;
            p.nblk_pc = n4;
        }
    }
    return p;
}

// split-K factor of the deepest level (0/1 = none): the 256-voxel x 32-cout decomposition leaves CUs idle
static int s1_ksplit(const S1Plan& p, int N, int D, int H, int W, int Cin, int Cout) {
    const int64_t units = p.nblk_pc * (Cout / 32);
    const int nchunks = Cin / 32;
    // one workgroup per CU is the aim: splitting further (two resident workgroups per CU) measured slower in round 3
    // (16^3: 38.8 -> 43.1 us, 8^3: 26.8 -> 28.2 us) - these kernels are bound by what enters the CU, not by its occupancy
    const int64_t target = 256;
    if (!p.small || p.nt2 || units >= target || nchunks < 2) return 1;
    int ks = 2;
    while (ks * 2 <= nchunks && units * ks < target) ks *= 2;
    (void)N; (void)D; (void)H; (void)W;
    return ks;
}

// the deep levels' in-workgroup split-K form (conv3_s1_sk_kernel)
struct SkPlan {
    int tw, ks;
    int64_t tiles;
};

static bool sk_plan(int N, int D, int H, int W, int Cin, int Cout, int ldx, SkPlan* p) {
    static const int mode = getenv("RU3D_CONV_SK") ? atoi(getenv("RU3D_CONV_SK")) : 1;   // 0 = off
    if (!mode) return false;
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    // tile width with the smaller padded volume (ties: the wider one); ragged extents are masked in the kernel
    const int64_t v16 = (int64_t)cdiv(D, 2) * 2 * cdiv(H, 4) * 4 * cdiv(W, 16) * 16;
    const int64_t v8 = (int64_t)cdiv(D, 2) * 2 * cdiv(H, 8) * 8 * cdiv(W, 8) * 8;
    const int tw = v16 <= v8 ? 16 : 8;
    const int th = 64 / tw;
    if ((Cout % 64) || (Cin % 64) || ldx <= 0 || (ldx % 8)) return false;
    if ((int64_t)D * H * W * ldx * 2 >= (1ll << 31)) return false;
    if (mode < 2 && (int64_t)D * H * W * 10 < (tw == 16 ? v16 : v8) * 3) return false;   // under 30 % of the MFMAs useful
    const int64_t tiles = (int64_t)N * cdiv(D, 2) * cdiv(H, th) * cdiv(W, tw);
    const int64_t units = tiles * (Cout / 64);
    if (units > 512) return false;              // the large levels run on the persistent kernels
    int ks = 1;
    while (units * ks < 192 && ks < 8 && ((Cin / (ks * 2)) % 64) == 0) ks *= 2;
    if (units * ks < 96) return false;
    p->tw = tw;
    p->ks = ks;
    p->tiles = tiles;
    return true;
}

static int launch_s1_sk(const MfmaConvArgs& a0, const SkPlan& sp, hipStream_t st) {
    MfmaConvArgs a = a0;
    const int th = 64 / sp.tw;
    a.tiles_d = (a.D + 1) / 2;
    a.tiles_h = (a.H + th - 1) / th;
    a.tiles_w = (a.W + sp.tw - 1) / sp.tw;
    a.nblk = (int)sp.tiles;
    a.ksplit = sp.ks;
    a.part = sp.ks > 1 ? (float*)a.ws : nullptr;
    dim3 grid((unsigned)(sp.tiles * (a.Cout / 64)), 1, sp.ks);
    if (sp.tw == 16)
        hipLaunchKernelGGL((conv3_s1_sk_kernel<16>), grid, dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((conv3_s1_sk_kernel<8>), grid, dim3(512), 0, st, a);
    int rc = ru3d_check_launch("conv3_s1_sk");
    if (rc || sp.ks <= 1) return rc;
    if (a.defer_ks) {        // the caller's next kernel sums the slices (norm_small.hip)
        *a.defer_ks = sp.ks;
        return 0;
    }
    const int64_t V = (int64_t)a.N * a.D * a.H * a.W;
    const int64_t groups = V * (a.Cout / 8);
    hipLaunchKernelGGL(conv_ksplit_reduce_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                       (const float*)a.part, sp.ks, V, a.Cout, a.bias, a.res, a.ldr, a.y, a.ldy);
    return ru3d_check_launch("conv_ksplit_reduce");
}

// workspace a 3x3x3 stride-1 conv of this geometry can use (split-K partials); 0 = none needed
size_t conv_mfma_ws_bytes(const ConvGeom& g) {
    if (!(g.k == 3 && g.stride == 1 && !g.transposed && (g.Cin % 32) == 0 && (g.Cout % 32) == 0)) return 0;
    SlidePlan sp;
    if (slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return 0;
    if (slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return 0;
    if (const int wsl = conv_ws_slices(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout))
        return (size_t)wsl * g.N * g.Do * g.Ho * g.Wo * g.Cout * sizeof(float);
    const S1Plan p = s1_plan(g.N, g.Do, g.Ho, g.Wo, g.Cout, g.ldx);
    int ks = s1_ksplit(p, g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout);
    SkPlan sk;
    if (p.small && sk_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, g.ldx, &sk) && sk.ks > ks) ks = sk.ks;
    return ks > 1 ? (size_t)ks * g.N * g.Do * g.Ho * g.Wo * g.Cout * sizeof(float) : 0;
}

static int launch_s1_auto(const MfmaConvArgs& a, hipStream_t st) {
    // the deepest level: whole-sample workgroups that read their weight slice once (conv_ws.hip), then the split-K sum
    if (const int wsl = conv_ws_slices(a.N, a.D, a.H, a.W, a.Cin, a.Cout)) {
        const size_t bytes = (size_t)wsl * a.N * a.D * a.H * a.W * a.Cout * sizeof(float);
        if (!a.stat_slab && a.ws && a.ws_bytes >= bytes && (((uintptr_t)a.ws) % 16) == 0 && (a.ldy % 8) == 0 &&
            (!a.res || (a.ldr % 8) == 0)) {
            int rc = conv_ws_launch(a.x, a.w, (float*)a.ws, a.N, a.D, a.H, a.W, a.Cin, a.Cout, a.ldx, a.flip, wsl, st);
            if (rc) return rc;
            if (a.defer_ks) {
                *a.defer_ks = wsl;
                return 0;
            }
            const int64_t V = (int64_t)a.N * a.D * a.H * a.W;
            const int64_t groups = V * (a.Cout / 8);
            hipLaunchKernelGGL(conv_ksplit_reduce_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                               (const float*)a.ws, wsl, V, a.Cout, a.bias, a.res, a.ldr, a.y, a.ldy);
            return ru3d_check_launch("conv_ksplit_reduce");
        }
    }
    const S1Plan p = s1_plan(a.N, a.D, a.H, a.W, a.Cout, a.ldx);
    if (a.stat_slab && !p.pc) return ru3d_fail(-1, "conv_mfma: fused statistics need the producer/consumer kernel");
    if (!p.small) {
        if (p.pc4) return launch_s1_pc4(a, st);
        if (p.pc && p.wclass == 32) return launch_s1_pc<2, 4, 32, 2>(a, p.nt2, st);
        if (p.pc && p.wclass == 16) return launch_s1_pc<2, 8, 16, 2>(a, p.nt2, st);
        if (p.pc && p.wclass == 8) return launch_s1_pc<4, 8, 8, 2>(a, p.nt2, st);
        if (p.wclass == 32) return launch_s1<2, 4, 32, 2>(a, p.nt2, st);
        if (p.wclass == 16) return launch_s1<2, 8, 16, 2>(a, p.nt2, st);
        return launch_s1<4, 8, 8, 2>(a, p.nt2, st);
    }
    // deep levels, exact-fit extents: 128-voxel x 64-cout blocks with the input channels split over the workgroup's waves
    {
        SkPlan sk;
        if (!a.stat_slab && sk_plan(a.N, a.D, a.H, a.W, a.Cin, a.Cout, a.ldx, &sk)) {
            const size_t bytes = (size_t)sk.ks * a.N * a.D * a.H * a.W * a.Cout * sizeof(float);
            if (sk.ks == 1 || (a.ws && a.ws_bytes >= bytes && (((uintptr_t)a.ws) % 16) == 0 && (a.ldy % 8) == 0 &&
                               (!a.res || (a.ldr % 8) == 0)))
                return launch_s1_sk(a, sk, st);
        }
    }
    // deep levels are bound by the weight bytes every workgroup pulls into its CU: when 256-voxel tiles x 32-cout
    // slices still give one workgroup per CU, they halve that traffic against the 128-voxel tiles
    const int ks = s1_ksplit(p, a.N, a.D, a.H, a.W, a.Cin, a.Cout);
    if (ks <= 1 && !p.nt2 && p.nblk_pc * (a.Cout / 32) >= 256) {
        if (p.wclass == 32) return launch_s1<2, 4, 32, 2>(a, false, st);
        if (p.wclass == 16) return launch_s1<2, 8, 16, 2>(a, false, st);
        return launch_s1<4, 8, 8, 2>(a, false, st);
    }
    // still fewer than one workgroup per CU (8^3): keep the 256-voxel x 32-cout tiles and split the 32-channel
    // chunks over gridDim.z; fp32 partials in the caller's workspace, fixed-order reduce with the bias / residual fused
    if (ks > 1 && (a.ldy % 8) == 0 && (!a.res || (a.ldr % 8) == 0)) {
        const size_t bytes = (size_t)ks * a.N * a.D * a.H * a.W * a.Cout * sizeof(float);
        if (a.ws && a.ws_bytes >= bytes && (((uintptr_t)a.ws) % 16) == 0) {
            MfmaConvArgs b = a;
            b.part = (float*)a.ws;
            b.ksplit = ks;
            if (p.wclass == 32) return launch_s1<2, 4, 32, 2>(b, false, st);
            if (p.wclass == 16) return launch_s1<2, 8, 16, 2>(b, false, st);
            return launch_s1<4, 8, 8, 2>(b, false, st);
        }
    }
    if (p.wclass == 32) return launch_s1<1, 4, 32, 1>(a, p.nt2, st);
    if (p.wclass == 16) return launch_s1<1, 8, 16, 1>(a, p.nt2, st);
    return launch_s1<2, 8, 8, 1>(a, p.nt2, st);
}

// Fused InstanceNorm statistics (producer/consumer kernel only): slab[workgroup][wave][n][cout_local][2] floats.
bool mfma_conv_can_fuse_stats(const ConvGeom& g) {
    if (!(g.k == 3 && g.stride == 1 && !g.transposed && (g.Cin % 32) == 0 && (g.Cout % 32) == 0)) return false;
    SlidePlan sp;
    if (slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return true;
    if (slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) return true;
    return s1_plan(g.N, g.Do, g.Ho, g.Wo, g.Cout, g.ldx).pc;
}

static void stats_slab_geom(const ConvGeom& g, int* gx, int* gy, int* cb) {
    SlidePlan sp;
    if (slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) {
        *gx = sp.grid;
        *gy = sp.ny;
        *cb = 32;
        return;
    }
    if (slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) {
        *gx = sp.grid;
        *gy = sp.ny;
        *cb = g.Cout == 32 ? 32 : 64;
        return;
    }
    const S1Plan p = s1_plan(g.N, g.Do, g.Ho, g.Wo, g.Cout, g.ldx);
    pc_grid(g.Cout, p.nt2, p.nblk_pc, gx, gy);
    *cb = p.nt2 ? 64 : 32;
}

size_t mfma_conv_stats_slab_bytes(const ConvGeom& g) {
    int gx, gy, cb;
    stats_slab_geom(g, &gx, &gy, &cb);
    return (size_t)gx * gy * 4 * g.N * cb * 2 * sizeof(float);
}

// mean / scale from the slabs: one thread block per (n, 4 channels); 64 lanes per channel sum the
// (workgroup, wave) partials in a fixed order, in double.
__global__ __launch_bounds__(256) void stats_slab_finalize_kernel(const float* __restrict__ slab, int gx, int cb, int N,
                                                                  int C, double invV, const float* __restrict__ drop,
                                                                  float eps, float* __restrict__ mean,
                                                                  float* __restrict__ scale) {
    const int lane = threadIdx.x & 63, cx = threadIdx.x >> 6;
    const int n = blockIdx.y, c = blockIdx.x * 4 + cx;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        const int y = c / cb, cl = c % cb;
        const int parts = gx * 4;   // (workgroup x, wave) pairs that hold channel c
        for (int p = lane; p < parts; p += 64) {
            const float* e = slab + ((((int64_t)(y * gx + (p >> 2)) * 4 + (p & 3)) * N + n) * cb + cl) * 2;
            a1 += (double)e[0];
            a2 += (double)e[1];
        }
    }
    const double t1 = wave_sum_d(a1), t2 = wave_sum_d(a2);   // xor-butterfly: fixed order
    if (lane == 0 && c < C) {
        const int i = n * C + c;
        if (!scale) {            // backward sums: mean[] receives m12 = (mean g', mean g' xhat)
            mean[2 * i] = (float)(t1 * invV);
            mean[2 * i + 1] = (float)(t2 * invV);
            return;
        }
        const double m = t1 * invV;
        double var = t2 * invV - m * m;
        if (var < 0.0) var = 0.0;
        const double sdrop = drop ? (double)drop[i] : 1.0;
        mean[i] = (float)m;
        scale[i] = (float)(sdrop / sqrt(sdrop * sdrop * var + (double)eps));
    }
}

int stats_slab_finalize_launch(const float* slab, int gx, int cb, int N, int C, double invV, const float* drop, float eps,
                               float* mean, float* scale, hipStream_t st) {
    dim3 grid((C + 3) / 4, N);
    hipLaunchKernelGGL(stats_slab_finalize_kernel, grid, dim3(256), 0, st, slab, gx, cb, N, C, invV, drop, eps, mean, scale);
    return ru3d_check_launch("stats_slab_finalize");
}

bool mfma_conv_can_fuse_bwd_sums(const ConvGeom& g) {
    if (!(g.k == 3 && g.stride == 1 && !g.transposed)) return false;
    SlidePlan sp;
    if (g.Cin == 32) return slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp);
    return false;      // (the 64-channel kernel's variant was removed in round 4: time-neutral at best, see conv_slide64.hip)
}

int mfma_conv_bwd_sums_finalize(const ConvGeom& g, const float* slab, float* m12, hipStream_t st) {
    int gx, gy, cb;
    stats_slab_geom(g, &gx, &gy, &cb);
    dim3 grid((g.Cout + 3) / 4, g.N);
    const double invV = 1.0 / ((double)g.Do * g.Ho * g.Wo);
    hipLaunchKernelGGL(stats_slab_finalize_kernel, grid, dim3(256), 0, st, slab, gx, cb, g.N, g.Cout, invV,
                       (const float*)nullptr, 0.f, m12, (float*)nullptr);
    return ru3d_check_launch("bwd_sums_slab_finalize");
}

int mfma_conv_stats_finalize(const ConvGeom& g, const float* slab, const float* drop, float eps, float* mean,
                             float* scale, hipStream_t st) {
    int gx, gy, cb;
    stats_slab_geom(g, &gx, &gy, &cb);
    dim3 grid((g.Cout + 3) / 4, g.N);
    const double invV = 1.0 / ((double)g.Do * g.Ho * g.Wo);
    hipLaunchKernelGGL(stats_slab_finalize_kernel, grid, dim3(256), 0, st, slab, gx, cb, g.N, g.Cout, invV, drop, eps,
                       mean, scale);
    return ru3d_check_launch("stats_slab_finalize");
}

// ---------------------------------------------------------------------------------------------------------
// "Direct" MFMA conv: the activation fragment of every (tap, k-step) is loaded straight from global memory
// (16 bytes per lane, L1/L2 absorb the tap re-reads), no LDS tile.  Covers every remaining data-movement
// form of the U-Net with MFMA-friendly channel counts: 1x1x1 convs (stride 1/2), 3x3x3 stride-2 gather
// (pooling conv, ConvTranspose dgrad) and the transposed form (ConvTranspose fwd, stride-2 conv dgrad).
// Transposed form: a workgroup owns 32 half-resolution positions x the 8 output parity classes; a wave's
// column tile holds ONE parity class, so only that class' taps (1/2/4/8 of 27) are issued - no masked MFMAs.
struct DirectArgs {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    bf16* y;
    int N, Di, Hi, Wi, Do, Ho, Wo;
    int Cin, Cout, ldx, ldy, ldr;
    int k, stride, pad, flip, zero_far;
    int hd, hh, hw;          // transposed form: half-resolution extents ceil(Do/2)...
    int64_t total;           // gather: N*Do*Ho*Wo ; transposed: N*hd*hh*hw
    int rows16;              // y (and res) allow 16-byte row pieces: pitch % 8 == 0, base 16-byte aligned
};

// Epilogue of one 32-position column tile held in 32x32x16 accumulators (lane -> position lane & 31, output channels
// 8 q + 4 (lane >> 5) + i of tile t).  Written straight from that layout, a store instruction puts 8 bytes per lane at the
// pitch of a voxel row: a 64- or 128-byte row arrives as 4-8 partial write requests.  Through a wave-private fp32 LDS
// patch the lanes are re-dealt row-major - NT * 4 lanes x 16 bytes cover a position's NT * 32 channels - so every
// request carries a whole row (and consecutive positions, where the form has them, one contiguous run).  Same
// arithmetic as the direct form: (acc + bias) [zeroed on the far planes] + residual, rounded once.
template <int NT>
__device__ __forceinline__ void tile_epilogue_rows(const f32x16 (&acc)[NT], float* patch, int lane, int64_t vox_lane,
                                                   bool far_lane, const float* bias, const bf16* res, int ldr, bf16* y,
                                                   int ldy, int co_blk) {
    constexpr int PP = NT * 32 + 4;          // floats per patch row
    constexpr int PPR = NT * 4;              // 8-channel pieces per row
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v = {acc[t][q * 4], acc[t][q * 4 + 1], acc[t][q * 4 + 2], acc[t][q * 4 + 3]};
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + co_blk + t * 32 + 8 * q + 4 * (lane >> 5));
            if (far_lane) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(patch + (lane & 31) * PP + t * 32 + 8 * q + 4 * (lane >> 5)) = v;
        }
    const int vlo = (int)(vox_lane & 0xffffffff), vhi = (int)(vox_lane >> 32);
#pragma unroll
    for (int it = 0; it < 32 * PPR / 64; it++) {
        const int p = lane + 64 * it;
        const int row = p / PPR, chunk = p % PPR;
        const int64_t vox = ((int64_t)__shfl(vhi, row, 64) << 32) | (uint32_t)__shfl(vlo, row, 64);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + row * PP + chunk * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + row * PP + chunk * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (vox >= 0) {
            if (res) {
                float r[8];
                load_vec<bf16, 8>(res + vox * ldr + co_blk + chunk * 8, r);
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] += r[i];
            }
            store_vec<bf16, 8>(y + vox * ldy + co_blk + chunk * 8, v);
        }
    }
}

template <int NT, bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv_direct_mfma_kernel(DirectArgs a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int co_blk = blockIdx.y * (NT * 32);
    const int taps = a.k * a.k * a.k;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;

    int n_[2], od[2], oh[2], ow[2];
    bool valid[2];
    int ba[2], bb[2], bc[2];   // transposed: half-res coordinates
#pragma unroll
    for (int m = 0; m < 2; m++) {
        if (!TRANSPOSED) {
            const int64_t v = (int64_t)blockIdx.x * 256 + (wave * 2 + m) * 32 + (lane & 31);
            valid[m] = v < a.total;
            const int64_t vv = valid[m] ? v : 0;
            ow[m] = (int)(vv % a.Wo);
            int64_t t = vv / a.Wo;
            oh[m] = (int)(t % a.Ho);
            t /= a.Ho;
            od[m] = (int)(t % a.Do);
            n_[m] = (int)(t / a.Do);
            ba[m] = bb[m] = bc[m] = 0;
        } else {
            const int64_t hp = (int64_t)blockIdx.x * 32 + (lane & 31);
            const bool ok = hp < a.total;
            const int64_t vv = ok ? hp : 0;
            bc[m] = (int)(vv % a.hw);
            int64_t t = vv / a.hw;
            bb[m] = (int)(t % a.hh);
            t /= a.hh;
            ba[m] = (int)(t % a.hd);
            n_[m] = (int)(t / a.hd);
            const int cl = wave * 2 + m;
            od[m] = 2 * ba[m] + (cl >> 2);
            oh[m] = 2 * bb[m] + ((cl >> 1) & 1);
            ow[m] = 2 * bc[m] + (cl & 1);
            valid[m] = ok && od[m] < a.Do && oh[m] < a.Ho && ow[m] < a.Wo;
        }
    }

#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int cl = wave * 2 + m;   // parity class of this column tile (transposed form)
        for (int kd = 0; kd < a.k; kd++) {
            int id, qd = 0;
            if (TRANSPOSED) {
                qd = (cl >> 2) + a.pad - kd;
                if (qd & 1) continue;              // wave-uniform: this tap does not feed this parity class
                id = ba[m] + (qd >> 1);
            } else {
                id = od[m] * a.stride + kd - a.pad;
            }
            for (int kh = 0; kh < a.k; kh++) {
                int ih;
                if (TRANSPOSED) {
                    const int qh = ((cl >> 1) & 1) + a.pad - kh;
                    if (qh & 1) continue;
                    ih = bb[m] + (qh >> 1);
                } else {
                    ih = oh[m] * a.stride + kh - a.pad;
                }
                for (int kw = 0; kw < a.k; kw++) {
                    int iw;
                    if (TRANSPOSED) {
                        const int qw = (cl & 1) + a.pad - kw;
                        if (qw & 1) continue;
                        iw = bc[m] + (qw >> 1);
                    } else {
                        iw = ow[m] * a.stride + kw - a.pad;
                    }
                    const int tap = (kd * a.k + kh) * a.k + kw;
                    const int wtap = a.flip ? taps - 1 - tap : tap;
                    const bool inb = valid[m] && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                    const bf16* xp = a.x + ((((int64_t)n_[m] * a.Di + (inb ? id : 0)) * a.Hi + (inb ? ih : 0)) * a.Wi +
                                            (inb ? iw : 0)) * a.ldx + (lane >> 5) * 8;
                    const bf16x8* wrow = a.w + ((int64_t)wtap * KS * NTT + blockIdx.y * NT) * 64 + lane;
#pragma unroll 2
                    for (int ks = 0; ks < KS; ks++) {
                        bf16x8 xb = {0, 0, 0, 0, 0, 0, 0, 0};
                        if (inb) xb = *reinterpret_cast<const bf16x8*>(xp + ks * 16);
#pragma unroll
                        for (int t = 0; t < NT; t++) {
                            const bf16x8 wa = wrow[((int64_t)ks * NTT + t) * 64];
                            acc[m][t] = RU3D_MFMA_32X32X16(wa, xb, acc[m][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

#pragma unroll
    for (int m = 0; m < 2; m++) {
        if (!valid[m]) continue;
        const int64_t vox = (((int64_t)n_[m] * a.Do + od[m]) * a.Ho + oh[m]) * a.Wo + ow[m];
        const bool far = a.zero_far && (od[m] == a.Do - 1 || oh[m] == a.Ho - 1 || ow[m] == a.Wo - 1);
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = co_blk + t * 32 + 8 * q + 4 * (lane >> 5);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
                if (a.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += b[i];
                }
                if (far) {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = 0.f;
                }
                if (a.res) {
                    float r[4];
                    load_vec<bf16, 4>(a.res + vox * a.ldr + c0, r);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += r[i];
                }
                store_vec<bf16, 4>(a.y + vox * a.ldy + c0, v);
            }
        }
    }
}

// Gather form on the large levels (3x3x3 stride-2 pooling conv / ConvTranspose input gradient, 1x1x1 convs), second
// take: the kernel above recomputes a lane's input coordinates and bounds for every (tap, k-step) and reloads the
// weight fragment for each of its two column tiles.  Here the per-lane work of an iteration is one bit test and one
// 64-bit add (the lane's base address and its 27-bit mask of in-range taps are computed once; the tap's address delta
// is wave-uniform; out-of-range lanes read a 16-byte zero block), the two column tiles share every weight fragment, and
// the fragments of iteration j + 1 are in flight while iteration j's MFMAs issue.
// 16 zero bytes in the code object's data segment: out-of-range lanes load them instead of branching
__device__ __attribute__((aligned(16))) bf16 g_zero16[8];

template <int NT, int RD>
__global__ __launch_bounds__(256) void conv_gather_mfma_kernel(DirectArgs a) {
    const bf16* const zero16 = g_zero16;
    __shared__ __attribute__((aligned(16))) float patch_s[4][32 * (NT * 32 + 4)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int co_blk = blockIdx.y * (NT * 32);
    const int taps = a.k * a.k * a.k;

    int n_[2], od[2], oh[2], ow[2];
    bool valid[2];
    const bf16* xbase[2];
    uint32_t mask[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int64_t v = (int64_t)blockIdx.x * 256 + (wave * 2 + m) * 32 + (lane & 31);
        valid[m] = v < a.total;
        const int64_t vv = valid[m] ? v : 0;
        ow[m] = (int)(vv % a.Wo);
        int64_t t = vv / a.Wo;
        oh[m] = (int)(t % a.Ho);
        t /= a.Ho;
        od[m] = (int)(t % a.Do);
        n_[m] = (int)(t / a.Do);
        const int id0 = od[m] * a.stride - a.pad, ih0 = oh[m] * a.stride - a.pad, iw0 = ow[m] * a.stride - a.pad;
        // address of tap (0,0,0) - may lie outside the tensor, only dereferenced for taps whose mask bit is set
        xbase[m] = a.x + ((((int64_t)n_[m] * a.Di + id0) * a.Hi + ih0) * a.Wi + iw0) * (int64_t)a.ldx + (lane >> 5) * 8;
        uint32_t mk = 0;
        for (int kd = 0; kd < a.k; kd++)
            for (int kh = 0; kh < a.k; kh++)
                for (int kw = 0; kw < a.k; kw++) {
                    const int id = id0 + kd, ih = ih0 + kh, iw = iw0 + kw;
                    const bool inb = valid[m] && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
                    mk |= (inb ? 1u : 0u) << ((kd * a.k + kh) * a.k + kw);
                }
        mask[m] = mk;
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[m][t][i] = 0.f;

    const int T = taps * KS;
    // RD iterations of fragments in flight (one iteration of prefetch until round 4: every iteration then waited ~500
    // cycles for its L2 hits behind 128 cycles of MFMA - 216 iterations x 0.25 us were the 54 us of the 128 -> 256 conv at
    // 32^3).  The (tap, k-step) counters advance with the fetches (wave-uniform, no division in the loop); iterations past
    // T load the zero block and the last weight fragment again, so the ring body has no branch and the waits stay counted.
    // RD = 6 for the 3x3x3 forms; the 1x1x1 forms (4-32 iterations, HBM-bound on the large levels) keep RD = 2.
    int f_j = 0, f_tap = 0, f_ks = 0, f_kw = 0, f_kh = 0, f_kd = 0;
    auto fetch = [&](bf16x8 (&xb)[2], bf16x8 (&wa)[NT]) {
        const bool live = f_j < T;
        const int64_t delta = (((int64_t)f_kd * a.Hi + f_kh) * a.Wi + f_kw) * a.ldx + f_ks * 16;
        const int wtap = a.flip ? taps - 1 - f_tap : f_tap;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const bf16* p = (live && ((mask[m] >> f_tap) & 1u)) ? xbase[m] + delta : zero16;
            xb[m] = *reinterpret_cast<const bf16x8*>(p);
        }
        const bf16x8* wrow = a.w + (((int64_t)wtap * KS + f_ks) * NTT + blockIdx.y * NT) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++) wa[t] = wrow[t * 64];
        if (f_j + 1 < T) {
            f_ks++;
            if (f_ks == KS) {
                f_ks = 0;
                f_tap++;
                f_kw++;
                if (f_kw == a.k) {
                    f_kw = 0;
                    f_kh++;
                    if (f_kh == a.k) {
                        f_kh = 0;
                        f_kd++;
                    }
                }
            }
        }
        f_j++;
    };
    {
        bf16x8 xr[RD][2], wr[RD][NT];
#pragma unroll
        for (int r = 0; r < RD; r++) fetch(xr[r], wr[r]);
        for (int j = 0; j < T; j += RD) {
#pragma unroll
            for (int r = 0; r < RD; r++) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int m = 0; m < 2; m++)
                        acc[m][t] = RU3D_MFMA_32X32X16(wr[r][t], xr[r][m], acc[m][t], 0, 0, 0);
                fetch(xr[r], wr[r]);
            }
        }
    }

    if (a.rows16) {
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int64_t vox = (((int64_t)n_[m] * a.Do + od[m]) * a.Ho + oh[m]) * a.Wo + ow[m];
            const bool far = a.zero_far && (od[m] == a.Do - 1 || oh[m] == a.Ho - 1 || ow[m] == a.Wo - 1);
            tile_epilogue_rows<NT>(acc[m], patch_s[wave], lane, valid[m] ? vox : -1, far, a.bias, a.res, a.ldr, a.y, a.ldy,
                                   co_blk);
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < 2; m++) {
        if (!valid[m]) continue;
        const int64_t vox = (((int64_t)n_[m] * a.Do + od[m]) * a.Ho + oh[m]) * a.Wo + ow[m];
        const bool far = a.zero_far && (od[m] == a.Do - 1 || oh[m] == a.Ho - 1 || ow[m] == a.Wo - 1);
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = co_blk + t * 32 + 8 * q + 4 * (lane >> 5);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = acc[m][t][q * 4 + i];
                if (a.bias) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += b[i];
                }
                if (far) {
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = 0.f;
                }
                if (a.res) {
                    float r[4];
                    load_vec<bf16, 4>(a.res + vox * a.ldr + c0, r);
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] += r[i];
                }
                store_vec<bf16, 4>(a.y + vox * a.ldy + c0, v);
            }
        }
    }
}

// The same forms on the deep levels (8^3 .. 16^3 voxels, hundreds of channels): one 256-voxel workgroup per 64 couts
// leaves tens of workgroups, each walking thousands of dependent (load, MFMA) pairs.  Here a workgroup owns ONE
// 32-voxel column tile (transposed form: 32 half-resolution positions of one parity class, blockIdx.z) x NT cout
// tiles, its 4 waves split the (tap, k-step) iterations, every wave keeps the next iteration's fragments in flight,
// and the four partial tiles are summed through LDS in a fixed order before the fused bias / residual / store.
template <int NT, bool TRANSPOSED, int RD>
__global__ __launch_bounds__(256) void conv_direct_ksplit_kernel(DirectArgs a) {
    __shared__ float red[4][NT][64][17];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int co_blk = blockIdx.y * (NT * 32);
    const int taps = a.k * a.k * a.k;
    const int cl = TRANSPOSED ? (int)blockIdx.z : 0;

    // this lane's output voxel (column of the MFMA tile)
    int n_, od, oh, ow, ba = 0, bb = 0, bc = 0;
    bool valid;
    {
        const int64_t v = (int64_t)blockIdx.x * 32 + (lane & 31);
        const bool ok = v < a.total;
        const int64_t vv = ok ? v : 0;
        if (!TRANSPOSED) {
            ow = (int)(vv % a.Wo);
            int64_t t = vv / a.Wo;
            oh = (int)(t % a.Ho);
            t /= a.Ho;
            od = (int)(t % a.Do);
            n_ = (int)(t / a.Do);
            valid = ok;
        } else {
            bc = (int)(vv % a.hw);
            int64_t t = vv / a.hw;
            bb = (int)(t % a.hh);
            t /= a.hh;
            ba = (int)(t % a.hd);
            n_ = (int)(t / a.hd);
            od = 2 * ba + (cl >> 2);
            oh = 2 * bb + ((cl >> 1) & 1);
            ow = 2 * bc + (cl & 1);
            valid = ok && od < a.Do && oh < a.Ho && ow < a.Wo;
        }
    }
    // taps that feed this tile, per axis: gather form all k; transposed form those with (class bit + pad - k) even
    int nkd, nkh, nkw, k0d, k0h, k0w, kstep;
    if (!TRANSPOSED) {
        nkd = nkh = nkw = a.k;
        k0d = k0h = k0w = 0;
        kstep = 1;
    } else {
        auto axis = [&](int bit, int& nk, int& k0) {
            k0 = (bit + a.pad) & 1;                 // first k with the right parity
            nk = k0 < a.k ? (a.k - k0 + 1) / 2 : 0;
        };
        axis(cl >> 2, nkd, k0d);
        axis((cl >> 1) & 1, nkh, k0h);
        axis(cl & 1, nkw, k0w);
        kstep = 2;
    }
    const int T = nkd * nkh * nkw * KS;                       // iterations of this tile
    const int j_lo = (T * wave) / 4, j_hi = (T * (wave + 1)) / 4;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    // Fragments of iteration j: the (tap, k-step) counters advance with j (wave-uniform: scalar unit), no divisions in the
    // loop.  An iteration past the end re-reads the last valid one with a zero activation fragment (its MFMAs add zero): the
    // ring below then has no branch in its body and the compiler's vmcnt waits stay counted.
    // The ring is RD iterations deep: with one iteration of prefetch (rounds 1-3) every step waited ~500 cycles for an L2
    // hit behind 64 cycles of MFMA - 108 steps x 0.4 us were the whole 45-58 us of these launches.  RD = 8 for the 3x3x3
    // forms, 2 for the 1x1x1 forms (4-8 iterations per wave: a deeper ring would be mostly padding).
    int f_ks, f_kw, f_kh, f_kd, f_j;         // state of the NEXT fetch
    {
        f_j = j_lo;
        f_ks = j_lo % KS;
        int tq = j_lo / KS;
        f_kw = tq % nkw;
        tq /= nkw;
        f_kh = tq % nkh;
        f_kd = tq / nkh;
    }
    auto fetch = [&](bf16x8& xb, bf16x8 (&wa)[NT]) {
        const bool live = f_j < j_hi;
        const int ks = f_ks;
        const int kw = k0w + kstep * f_kw, kh = k0h + kstep * f_kh, kd = k0d + kstep * f_kd;
        int id, ih, iw;
        if (TRANSPOSED) {
            id = ba + (((cl >> 2) + a.pad - kd) >> 1);
            ih = bb + ((((cl >> 1) & 1) + a.pad - kh) >> 1);
            iw = bc + (((cl & 1) + a.pad - kw) >> 1);
        } else {
            id = od * a.stride + kd - a.pad;
            ih = oh * a.stride + kh - a.pad;
            iw = ow * a.stride + kw - a.pad;
        }
        const int tap = (kd * a.k + kh) * a.k + kw;
        const int wtap = a.flip ? taps - 1 - tap : tap;
        const bool inb = live && valid && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
        // out-of-range lanes load a 16-byte zero block: no select behind the load (the compiler gathered such selects at
        // the top of the unrolled ring body, i.e. waited for the youngest load there)
        const bf16* xp = inb ? a.x + ((((int64_t)n_ * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.ldx + (lane >> 5) * 8 + ks * 16
                             : (const bf16*)g_zero16;
        xb = *reinterpret_cast<const bf16x8*>(xp);
        const bf16x8* wrow = a.w + (((int64_t)wtap * KS + ks) * NTT + blockIdx.y * NT) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; t++) wa[t] = wrow[t * 64];
        if (live && f_j + 1 < j_hi) {         // advance (the last valid iteration is kept for the padding fetches)
            f_ks++;
            if (f_ks == KS) {
                f_ks = 0;
                f_kw++;
                if (f_kw == nkw) {
                    f_kw = 0;
                    f_kh++;
                    if (f_kh == nkh) {
                        f_kh = 0;
                        f_kd++;
                    }
                }
            }
        }
        f_j++;
    };

    if (j_lo < j_hi) {
        bf16x8 xb[RD], wa[RD][NT];
#pragma unroll
        for (int r = 0; r < RD; r++) fetch(xb[r], wa[r]);
        for (int j = j_lo; j < j_hi; j += RD) {
#pragma unroll
            for (int r = 0; r < RD; r++) {
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = RU3D_MFMA_32X32X16(wa[r][t], xb[r], acc[t], 0, 0, 0);
                fetch(xb[r], wa[r]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) red[wave][t][lane][i] = acc[t][i];
    __syncthreads();

    // thread -> (voxel, 8 consecutive couts); D layout: col = lane & 31 (voxel), row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int vox_l = tid >> 3, cg = tid & 7;
    if (cg * 8 >= NT * 32) return;
    // re-derive the voxel of column vox_l (it is lane vox_l's voxel)
    const int64_t v2 = (int64_t)blockIdx.x * 32 + vox_l;
    if (v2 >= a.total) return;
    int n2, d2, h2, w2;
    if (!TRANSPOSED) {
        w2 = (int)(v2 % a.Wo);
        int64_t t = v2 / a.Wo;
        h2 = (int)(t % a.Ho);
        t /= a.Ho;
        d2 = (int)(t % a.Do);
        n2 = (int)(t / a.Do);
    } else {
        const int c2 = (int)(v2 % a.hw);
        int64_t t = v2 / a.hw;
        const int b2 = (int)(t % a.hh);
        t /= a.hh;
        const int a2 = (int)(t % a.hd);
        n2 = (int)(t / a.hd);
        d2 = 2 * a2 + (cl >> 2);
        h2 = 2 * b2 + ((cl >> 1) & 1);
        w2 = 2 * c2 + (cl & 1);
        if (d2 >= a.Do || h2 >= a.Ho || w2 >= a.Wo) return;
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int co = cg * 8 + e;
        const int t = co >> 5, r = co & 31;
        const int h = (r >> 2) & 1, i = (r & 3) + 4 * (r >> 3);
        const int ln = vox_l + 32 * h;
        v[e] = (red[0][t][ln][i] + red[1][t][ln][i]) + (red[2][t][ln][i] + red[3][t][ln][i]);
    }
    const int c0 = co_blk + cg * 8;
    if (a.bias) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += a.bias[c0 + e];
    }
    if (a.zero_far && (d2 == a.Do - 1 || h2 == a.Ho - 1 || w2 == a.Wo - 1)) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = 0.f;
    }
    const int64_t vox = (((int64_t)n2 * a.Do + d2) * a.Ho + h2) * a.Wo + w2;
    if (a.res) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += (float)a.res[vox * a.ldr + c0 + e];
    }
    // y pitch and base are 8-byte aligned (checked by the launcher); two 4-element stores
    float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
    store_vec<bf16, 4>(a.y + vox * a.ldy + c0, lo);
    store_vec<bf16, 4>(a.y + vox * a.ldy + c0 + 4, hi);
}

// Transposed form (ConvTranspose3d k3 s2 p1 forward, stride-2 conv input gradient) on the large levels through an
// LDS tile: a workgroup owns 128 half-resolution positions (TD x TH x TW) and stages their (TD+1)(TH+1)(TW+1) input
// rows ONCE - all 8 output parity classes read the same 2x2x2 neighbourhoods, so the 27 taps cost one global read
// of the input instead of 27 L1/L2 gathers.  Waves take whole parity classes (8 | 4+2+1 | 4+2 | 4+2 taps); per class
// a wave holds the 4 column tiles of the workgroup, so every weight fragment (global, prefetched one iteration
// ahead) feeds 4 MFMAs per cout tile.  Row pitch Cin*2 + 16 bytes: conflict-free ds_read_b128 over a W-run.
template <int NT, int TD, int TH, int TW>
__global__ __launch_bounds__(256, 2) void convt_tile_mfma_kernel(DirectArgs a) {
    extern __shared__ __attribute__((aligned(16))) bf16 tile_lds[];
    static_assert((TD * TH * TW) % 32 == 0, "whole column tiles");
    constexpr int HD = TD + 1, HH = TH + 1, HW = TW + 1, ROWS = HD * HH * HW;
    constexpr int MT = TD * TH * TW / 32;       // column tiles of 32 positions: every weight fragment feeds MT MFMAs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NTT = a.Cout / 32, KS = a.Cin / 16;
    const int pitch = a.Cin + 8;                 // bf16 elements
    const int co_blk = blockIdx.y * (NT * 32);

    // tile origin in half-resolution coordinates
    const int tw_n = (a.hw + TW - 1) / TW, th_n = (a.hh + TH - 1) / TH, td_n = (a.hd + TD - 1) / TD;
    int t = blockIdx.x;
    const int c0 = (t % tw_n) * TW;
    t /= tw_n;
    const int b0 = (t % th_n) * TH;
    t /= th_n;
    const int a0 = (t % td_n) * TD;
    const int n = t / td_n;

    // ---- stage the input rows (zero outside the tensor)
    // eight pieces per thread in flight: written as load -> store per piece the loop ran at one memory round trip per
    // iteration (the store waits for its load), 16-31 of them per tile
    const int ppr = a.Cin / 8;                   // 16-byte pieces per row
    constexpr int SB = 8;
    for (int p0 = tid; p0 < ROWS * ppr; p0 += 256 * SB) {
        bf16x8 v[SB];
        int dst[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) {
            const int p = p0 + 256 * u;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            v[u] = z8;
            dst[u] = -1;
            if (p < ROWS * ppr) {
                const int r = p / ppr, piece = p - r * ppr;
                const int zc = r % HW, zb = (r / HW) % HH, za = r / (HW * HH);
                const int ia = a0 + za, ib = b0 + zb, ic = c0 + zc;
                dst[u] = r * pitch + piece * 8;
                if (ia < a.Di && ib < a.Hi && ic < a.Wi)
                    v[u] = *reinterpret_cast<const bf16x8*>(a.x + ((((int64_t)n * a.Di + ia) * a.Hi + ib) * a.Wi + ic) * a.ldx + piece * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < SB; u++)
            if (dst[u] >= 0) *reinterpret_cast<bf16x8*>(tile_lds + dst[u]) = v[u];
    }
    __syncthreads();

    // column tile m of the workgroup: positions f = m*32 + (lane & 31)
    int rowm[MT], pa[MT], pb[MT], pc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int f = m * 32 + (lane & 31);
        pc[m] = f % TW;
        pb[m] = (f / TW) % TH;
        pa[m] = f / (TW * TH);
        rowm[m] = ((pa[m] * HH + pb[m]) * HW + pc[m]) * pitch + (lane >> 5) * 8;
    }

    // classes of this wave, heaviest first: {7} | {3, 1, 0} | {5, 2} | {6, 4}   (taps 8 | 4+2+1 | 4+2 | 4+2)
    const int ncls = wave == 0 ? 1 : (wave == 1 ? 3 : 2);
    for (int ci = 0; ci < ncls; ci++) {
        const int cl = wave == 0 ? 7 : wave == 1 ? (ci == 0 ? 3 : (ci == 1 ? 1 : 0)) : wave == 2 ? (ci == 0 ? 5 : 2) : (ci == 0 ? 6 : 4);
        const int bd = cl >> 2, bh = (cl >> 1) & 1, bw = cl & 1;
        // taps of the class per axis: bit 0 -> k = 1 (input offset 0); bit 1 -> k = 0 (offset 1), k = 2 (offset 0)
        const int nkd = 1 + bd, nkh = 1 + bh, nkw = 1 + bw;
        const int T = nkd * nkh * nkw * KS;

        f32x16 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int tt = 0; tt < NT; tt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[m][tt][i] = 0.f;

        auto wfetch = [&](int j, bf16x8 (&wa)[NT], int& roff) {
            const int ks = j % KS;
            int tq = j / KS;
            const int iw = tq % nkw;
            tq /= nkw;
            const int ih = tq % nkh;
            const int idd = tq / nkh;
            // axis with class bit 1: index 0 -> k = 0 (offset 1), index 1 -> k = 2 (offset 0); bit 0: k = 1 (offset 0)
            const int kd = bd ? 2 * idd : 1, kh = bh ? 2 * ih : 1, kw = bw ? 2 * iw : 1;
            const int dd = bd ? 1 - idd : 0, dh = bh ? 1 - ih : 0, dw = bw ? 1 - iw : 0;
            const int tap = (kd * 3 + kh) * 3 + kw;
            const int wtap = a.flip ? 26 - tap : tap;
            const bf16x8* wrow = a.w + (((int64_t)wtap * KS + ks) * NTT + blockIdx.y * NT) * 64 + lane;
#pragma unroll
            for (int tt = 0; tt < NT; tt++) wa[tt] = wrow[tt * 64];
            roff = ((dd * HH + dh) * HW + dw) * pitch + ks * 16;
        };
        auto compute = [&](const bf16x8 (&wa)[NT], int roff) {
            bf16x8 xb[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) xb[m] = *reinterpret_cast<const bf16x8*>(tile_lds + rowm[m] + roff);
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int tt = 0; tt < NT; tt++)
                    acc[m][tt] = RU3D_MFMA_32X32X16(wa[tt], xb[m], acc[m][tt], 0, 0, 0);
        };
        // weight fragments RD iterations ahead (one until round 4: an L2 hit takes longer than the 4-8 MFMAs of an
        // iteration), activation fragments of the next iteration read from LDS in front of this iteration's MFMAs;
        // the fetches past the end repeat the last iteration's (branch-free body, counted waits)
        auto ring = [&](auto rd_tag) {
            constexpr int RD = decltype(rd_tag)::value;
            bf16x8 wq[RD][NT], xq[2][MT];
            int ro[RD];
            int fj = 0;
            auto wnext = [&](bf16x8 (&wa)[NT], int& roff) {
                wfetch(fj < T ? fj : T - 1, wa, roff);
                fj++;
            };
#pragma unroll
            for (int r = 0; r < RD; r++) wnext(wq[r], ro[r]);
#pragma unroll
            for (int m = 0; m < MT; m++) xq[0][m] = *reinterpret_cast<const bf16x8*>(tile_lds + rowm[m] + ro[0]);
            for (int j = 0; j < T; j += RD) {
#pragma unroll
                for (int r = 0; r < RD; r++) {
#pragma unroll
                    for (int m = 0; m < MT; m++)
                        xq[(r + 1) & 1][m] = *reinterpret_cast<const bf16x8*>(tile_lds + rowm[m] + ro[(r + 1) % RD]);
#pragma unroll
                    for (int m = 0; m < MT; m++)
#pragma unroll
                        for (int tt = 0; tt < NT; tt++)
                            acc[m][tt] = RU3D_MFMA_32X32X16(wq[r][tt], xq[r & 1][m], acc[m][tt], 0, 0, 0);
                    wnext(wq[r], ro[r]);
                }
            }
        };
        // (8 deep for the 4-MFMA iterations of the NT = 1 forms: 56 vs 54-55 us - no better)
        if ((T % 4) == 0) {
            ring(std::integral_constant<int, 4>{});
        } else {
            bf16x8 w0[NT], w1[NT];
            int r0, r1;
            wfetch(0, w0, r0);
            int j = 0;
            for (; j + 1 < T; j += 2) {
                wfetch(j + 1, w1, r1);
                compute(w0, r0);
                if (j + 2 < T) wfetch(j + 2, w0, r0);
                compute(w1, r1);
            }
            if (j < T) compute(w0, r0);
        }

        // ---- epilogue of the class: output voxel (2a + bd, 2b + bh, 2c + bw)
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const int od = 2 * (a0 + pa[m]) + bd, oh = 2 * (b0 + pb[m]) + bh, ow = 2 * (c0 + pc[m]) + bw;
            if (od >= a.Do || oh >= a.Ho || ow >= a.Wo) continue;
            const int64_t vox = (((int64_t)n * a.Do + od) * a.Ho + oh) * a.Wo + ow;
            const bool far = a.zero_far && (od == a.Do - 1 || oh == a.Ho - 1 || ow == a.Wo - 1);
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int cc = co_blk + tt * 32 + 8 * q + 4 * (lane >> 5);
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = acc[m][tt][q * 4 + i];
                    if (a.bias) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + cc);
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] += b[i];
                    }
                    if (far) {
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] = 0.f;
                    }
                    if (a.res) {
                        float r[4];
                        load_vec<bf16, 4>(a.res + vox * a.ldr + cc, r);
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] += r[i];
                    }
                    store_vec<bf16, 4>(a.y + vox * a.ldy + cc, v);
                }
            }
        }
    }
}

template <int NT, int TD, int TH, int TW>
static int launch_convt_tile(const DirectArgs& a, int N, hipStream_t st) {
    const int tw_n = (a.hw + TW - 1) / TW, th_n = (a.hh + TH - 1) / TH, td_n = (a.hd + TD - 1) / TD;
    const int64_t nblk = (int64_t)N * td_n * th_n * tw_n;
    if (nblk > 0x7fffffff) return ru3d_fail(-1, "convt_tile: grid too large");
    const size_t lds = (size_t)(TD + 1) * (TH + 1) * (TW + 1) * (a.Cin + 8) * sizeof(bf16);
    auto kern = convt_tile_mfma_kernel<NT, TD, TH, TW>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return ru3d_fail(-1, "convt_tile: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, a.Cout / (NT * 32)), dim3(256), lds, st, a);
    return ru3d_check_launch("convt_tile_mfma");
}

static int launch_direct(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                         hipStream_t st) {
    DirectArgs a;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.bias = bias;
    a.res = (const bf16*)res;
    a.y = (bf16*)y;
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldr;
    a.k = g.k; a.stride = g.stride; a.pad = g.pad; a.flip = g.flip; a.zero_far = g.zero_far;
    a.hd = (g.Do + 1) / 2; a.hh = (g.Ho + 1) / 2; a.hw = (g.Wo + 1) / 2;
    a.rows16 = (g.ldy % 8) == 0 && (((uintptr_t)y) % 16) == 0 && (!res || ((g.ldr % 8) == 0 && (((uintptr_t)res) % 16) == 0));
    const bool nt2 = (g.Cout % 64) == 0;
    int64_t nblk;
    if (g.transposed) {
        if (g.stride != 2) return ru3d_fail(-1, "conv_direct_mfma: transposed form needs stride 2");
        a.total = (int64_t)g.N * a.hd * a.hh * a.hw;
        nblk = (a.total + 31) / 32;
    } else {
        a.total = (int64_t)g.N * g.Do * g.Ho * g.Wo;
        nblk = (a.total + 255) / 256;
    }
    if (nblk > 0x7fffffff) return ru3d_fail(-1, "conv_direct_mfma: grid too large");
    // (first: the fused entry points pick these kernels by the same test, and a block recomputed under activation
    // checkpointing must get the same bits from the plain entry points)
    // the large levels' stride-2 forms: LDS-DMA plane ring + producer wave + weights in registers (conv_s2.hip)
    if (g.transposed && convt_s2_tile_eligible(g) &&
        ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)res)) % 16) == 0)
        return convt_s2_tile_launch(x, w, bias, res, y, g, nullptr, nullptr, 0, nullptr, st);
    if (!g.transposed && !res && conv_s2_tile_eligible(g) && (((uintptr_t)x) % 16) == 0 && (((uintptr_t)y) % 8) == 0)
        return conv_s2_tile_launch(x, w, bias, y, g, nullptr, nullptr, nullptr, nullptr, 0, st);
    if (nblk * (g.Cout / (nt2 ? 64 : 32)) < 384) {
        // few, long workgroups: split the reduction over the waves of 32-voxel workgroups instead
        const int64_t tiles = (a.total + 31) / 32;
        dim3 gk((unsigned)tiles, g.Cout / (nt2 ? 64 : 32), g.transposed ? 8 : 1);
        const bool deep = g.k >= 3;
#define RU3D_KSPLIT_LAUNCH(NTV, TRV)                                                                              \
    if (deep) hipLaunchKernelGGL((conv_direct_ksplit_kernel<NTV, TRV, 8>), gk, dim3(256), 0, st, a);             \
    else hipLaunchKernelGGL((conv_direct_ksplit_kernel<NTV, TRV, 2>), gk, dim3(256), 0, st, a)
        if (g.transposed) {
            if (nt2) { RU3D_KSPLIT_LAUNCH(2, true); } else { RU3D_KSPLIT_LAUNCH(1, true); }
        } else {
            if (nt2) { RU3D_KSPLIT_LAUNCH(2, false); } else { RU3D_KSPLIT_LAUNCH(1, false); }
        }
#undef RU3D_KSPLIT_LAUNCH
        return ru3d_check_launch("conv_direct_ksplit");
    }
    static const int tile_mode = getenv("RU3D_CONVT_TILE") ? atoi(getenv("RU3D_CONVT_TILE")) : 1;
    if (tile_mode && g.transposed && g.k == 3 && g.pad == 1) {
        // (TD+1)(TH+1)(TW+1) rows of Cin*2+16 bytes must fit in LDS: Cin <= 256 with the 16-wide tile, 128 with the 32-wide
        if (a.hw >= 24 && g.Cin <= 128) {
            // one cout tile: 256 positions per workgroup (8 column tiles per weight fragment) - the 110 KB of weights a
            // workgroup pulls through its CU's ~10 B/clk are the larger part of what enters it
            if (!nt2 && g.Cin <= 64 && a.hh >= 4) return launch_convt_tile<1, 2, 4, 32>(a, g.N, st);
            return nt2 ? launch_convt_tile<2, 2, 2, 32>(a, g.N, st) : launch_convt_tile<1, 2, 2, 32>(a, g.N, st);
        }
        if (a.hw >= 12 && g.Cin <= 256) {
            // 64-cout workgroups only when they still fill the chip (16^3 -> 32^3 at N = 2: 64 tiles)
            static const int nt_mode = getenv("RU3D_CONVT_NT") ? atoi(getenv("RU3D_CONVT_NT")) : 1;
            const int64_t t16 = (int64_t)g.N * ((a.hd + 1) / 2) * ((a.hh + 3) / 4) * ((a.hw + 15) / 16);
            const bool wide = nt2 && (nt_mode == 0 || t16 * (g.Cout / 64) >= 200);
            return wide ? launch_convt_tile<2, 2, 4, 16>(a, g.N, st) : launch_convt_tile<1, 2, 4, 16>(a, g.N, st);
        }
    }
    dim3 grid((unsigned)nblk, g.Cout / (nt2 ? 64 : 32));
    if (!g.transposed && g.k * g.k * g.k <= 27) {
        if (g.k >= 3) {
            if (nt2) hipLaunchKernelGGL((conv_gather_mfma_kernel<2, 6>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_mfma_kernel<1, 6>), grid, dim3(256), 0, st, a);
        } else {
            if (nt2) hipLaunchKernelGGL((conv_gather_mfma_kernel<2, 2>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_gather_mfma_kernel<1, 2>), grid, dim3(256), 0, st, a);
        }
        return ru3d_check_launch("conv_gather_mfma");
    }
    if (!g.transposed) return ru3d_fail(-1, "conv_direct_mfma: no gather kernel for k = %d", g.k);
    if (nt2) hipLaunchKernelGGL((conv_direct_mfma_kernel<2, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_direct_mfma_kernel<1, true>), grid, dim3(256), 0, st, a);
    return ru3d_check_launch("conv_direct_mfma");
}

static bool aligned_to(const void* p, size_t a) { return (((uintptr_t)p) % a) == 0; }

// Every data-movement form with MFMA-friendly channel counts runs on MFMA: the 3x3x3 stride-1 gather with
// Cin % 32 == 0 on the LDS-halo kernel, everything else on the direct kernel (same packed-weight layout).
bool mfma_conv_geometry_ok(const ConvGeom& g) { return !g.transposed || g.stride == 2; }

bool mfma_conv_can_fuse_partner(const ConvGeom& g) {
    if (!(g.k == 3 && g.stride == 1 && !g.transposed && g.Cin == 32)) return false;
    SlidePlan sp;
    return slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp);
}

static int conv_mfma_launch_inner(const void* x, const void* w, const float* bias, const void* res, void* y,
                                  const ConvGeom& g, hipStream_t st, float* stat_slab, void* ws, size_t ws_bytes,
                                  const void* bst_act, int bst_ld, float slope, const void* x2, int ldx2, const void* w2,
                                  int* defer_ks);

int conv_mfma_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                     hipStream_t st, float* stat_slab, void* ws, size_t ws_bytes, const void* bst_act, int bst_ld,
                     float slope, const void* x2, int ldx2, const void* w2, int* defer_ks) {
    // bench.py's kernel probe: an event pair around the conv kernel itself (ru3d_probe_begin; comm.hip)
    void* stop = (g.k == 3 && g.stride == 1 && !g.transposed)
                     ? ru3d_probe_start(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, st) : nullptr;
    const int rc = conv_mfma_launch_inner(x, w, bias, res, y, g, st, stat_slab, ws, ws_bytes, bst_act, bst_ld, slope, x2, ldx2,
                                          w2, defer_ks);
    ru3d_probe_stop(stop, st);
    return rc;
}

static int conv_mfma_launch_inner(const void* x, const void* w, const float* bias, const void* res, void* y,
                                  const ConvGeom& g, hipStream_t st, float* stat_slab, void* ws, size_t ws_bytes,
                                  const void* bst_act, int bst_ld, float slope, const void* x2, int ldx2, const void* w2,
                                  int* defer_ks) {
    if (defer_ks) *defer_ks = 0;
    if (!mfma_conv_geometry_ok(g)) return ru3d_fail(-1, "conv_mfma: geometry not supported");
    if (g.x_cseg || g.y_cseg) {
        // planar concat (see ru3d_tensor): the decoder block's conv1 on the 64-channel sliding kernel (split input), its
        // input gradient pair on the 32-channel sliding kernel (split output: one plane per 32-channel slice)
        SlidePlan sp;
        const bool in_ok = g.x_cseg && !g.y_cseg && g.k == 3 && g.stride == 1 && !g.transposed &&
                           slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp);
        const bool out_ok = g.y_cseg && !g.x_cseg && x2 && mfma_conv_can_fuse_partner(g);
        if (!in_ok && !out_ok) return ru3d_fail(-1, "conv_mfma: no kernel takes this split tensor");
    }
    if (x2 && !mfma_conv_can_fuse_partner(g)) return ru3d_fail(-1, "conv_mfma: no fused 1x1 partner for this shape");
    if ((g.ldx % 8) || (g.ldy % 4) || (res && (g.ldr % 4)) || !aligned_to(x, 16) || !aligned_to(y, 8) ||
        (res && !aligned_to(res, 8)) || (bias && !aligned_to(bias, 16)) || !aligned_to(w, 16))
        return ru3d_fail(-1, "conv_mfma: operands must be 16-byte (x, w, bias) / 8-byte (y, res) aligned");
    if (bst_act && !mfma_conv_can_fuse_bwd_sums(g))
        return ru3d_fail(-1, "conv_mfma: fused backward sums are not available for this shape");
    if (!(g.k == 3 && g.stride == 1 && !g.transposed && (g.Cin % 32) == 0)) {
        if (stat_slab) return ru3d_fail(-1, "conv_mfma: fused statistics not available for this form");
        return launch_direct(x, w, bias, res, y, g, st);
    }
    {
        SlidePlan sp;
        if (slide_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) {
            // 16-byte stores; a sample within the 30-bit element range of the staging loads' buffer offsets
            const bool aligned = (g.ldy % 8) == 0 && (!res || (g.ldr % 8) == 0) && aligned_to(y, 16) &&
                                 (!res || aligned_to(res, 16)) && (g.ldx % 8) == 0 && aligned_to(x, 16) &&
                                 (int64_t)g.Do * g.Ho * g.Wo * g.ldx < (1ll << 30);
            if (aligned && (!bst_act || ((bst_ld % 8) == 0 && aligned_to(bst_act, 16))))
                return conv_slide_launch(x, w, bias, res, y, g, stat_slab, st, bst_act, bst_ld, slope, x2, ldx2, w2);
            if (bst_act) return ru3d_fail(-1, "conv_mfma: the fused backward sums need 16-byte aligned operands");
            if (x2) return ru3d_fail(-1, "conv_mfma: the fused 1x1 partner needs 16-byte aligned operands");
            // the statistics slab (size, layout) was planned for the sliding kernel's grid: falling back to the
            // producer/consumer kernel here would fill it with another geometry
            if (stat_slab)
                return ru3d_fail(-1, "conv_mfma: fused statistics need y (and res) 16-byte aligned with a pitch that is "
                                     "a multiple of 8 on this shape");
        }
        if (slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &sp)) {
            const bool aligned = (g.ldy % 8) == 0 && (!res || (g.ldr % 8) == 0) && aligned_to(y, 16) &&
                                 (!res || aligned_to(res, 16)) && (g.ldx % 8) == 0 && aligned_to(x, 16) &&
                                 (int64_t)g.Do * g.Ho * g.Wo * g.ldx < (1ll << 30);
            if (aligned && (!bst_act || ((bst_ld % 8) == 0 && aligned_to(bst_act, 16))))
                return conv_slide64_launch(x, w, bias, res, y, g, stat_slab, st, bst_act, bst_ld, slope);
            if (bst_act) return ru3d_fail(-1, "conv_mfma: the fused backward sums need 16-byte aligned operands");
            if (stat_slab)
                return ru3d_fail(-1, "conv_mfma: fused statistics need y (and res) 16-byte aligned with a pitch that is "
                                     "a multiple of 8 on this shape");
        }
    }
    MfmaConvArgs a;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.bias = bias;
    a.res = (const bf16*)res;
    a.y = (bf16*)y;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldr;
    a.flip = g.flip;
    a.stat_slab = stat_slab;
    a.part = nullptr;
    a.ksplit = 1;
    a.defer_ks = (res || stat_slab) ? nullptr : defer_ks;
    a.ws = ws;
    a.ws_bytes = ws_bytes;
    return launch_s1_auto(a, st);
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 stride-1 conv on MFMA:
//     dW[tap][ci][co] = sum_pos X[pos + tap][ci] * DY[pos][co]
// is 27 GEMMs (M = ci, N = co, K = positions) that share the DY operand.  Both operands are K-major in
// memory (NDHWC: a position's channels are contiguous, MFMA wants 8 consecutive K per lane), so tiles are
// staged row-per-position in LDS (64-byte rows, no padding: conflict-free for the transposed read) and the
// fragments are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose).
// Workgroup = one (32 ci) x (32 co) pair; its 4 waves split the 27 taps (7/7/7/6), each wave keeping its
// taps' 32x32 fp32 accumulators in registers while the workgroup walks position tiles (persistent, stride
// G).  Partial sums go to a slab per workgroup and are reduced in fixed order (deterministic, no atomics).
struct MfmaWgradArgs {
    const bf16* x;
    const bf16* dy;
    float* part;
    int N, D, H, W;
    int Cin, Cout, ldx, lddy;
    int tiles_d, tiles_h, tiles_w, ntiles;
    int G;   // workgroups per channel pair (= number of partial slabs)
    float* dw;   // G == 1 on the deepest level: the gradient itself, [co][ci][27] (no slab, no reduce pass)
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const bf16* p) {
    // 8 consecutive K (positions) of this lane's channel: two 4-row transposed reads
    const bf16x4 lo = RU3D_DS_READ_TR16(p);
    const bf16x4 hi = RU3D_DS_READ_TR16(p + 4 * 32);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <int TD, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad3_s1_mfma_kernel(MfmaWgradArgs a) {
    constexpr int HD = TD + 2, HH = TH + 2, WW = TW + 2;
    constexpr int HV = HD * HH * WW;
    static_assert(TD * TH * TW == 256, "tile must hold 256 positions");
    static_assert((HV + 256) * 64 <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) bf16 lds[(HV + 256) * 32];
    bf16* xs = lds;             // [HV][32]  halo tile of 32 input channels
    bf16* ds = lds + HV * 32;   // [256][32] dy tile of 32 output channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int COT = a.Cout / 32;
    const int cit = blockIdx.y / COT, cot = blockIdx.y % COT;

    // this wave's taps: wave, wave + 4, ... (tap 27 = none)
    int toff[7];
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int tap = wave + 4 * t < 27 ? wave + 4 * t : 26;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        toff[t] = ((kd * HH + kh) * WW + kw) * 32;
    }
    f32x16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;

    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;   // row q of the 4x16 block, columns 4p..4p+3

    // The NEXT tile's global loads are issued before the MFMA loop of the current one and written to LDS after it:
    // the staging registers are the second buffer, the load latency hides under 112 MFMAs per wave.
    constexpr int NIT = (HV * 4 + 255) / 256;
    bf16x8 sx[NIT], sd[4];
    auto load_tile = [&](int tile) {
        int tt = tile;
        const int w0 = (tt % a.tiles_w) * TW;
        tt /= a.tiles_w;
        const int h0 = (tt % a.tiles_h) * TH;
        tt /= a.tiles_h;
        const int d0 = (tt % a.tiles_d) * TD;
        const int n = tt / a.tiles_d;
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int c = tid + i * 256;
            const int hv = c >> 2, part = c & 3;
            const int zw = hv % WW, zh = (hv / WW) % HH, zd = hv / (WW * HH);
            const int gd = d0 + zd - 1, gh = h0 + zh - 1, gw = w0 + zw - 1;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < HV * 4 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)
                v = *reinterpret_cast<const bf16x8*>(a.x + ((((int64_t)n * a.D + gd) * a.H + gh) * a.W + gw) * a.ldx +
                                                     cit * 32 + part * 8);
            sx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + i * 256;
            const int f = c >> 2, part = c & 3;
            const int gd = d0 + f / (TH * TW), gh = h0 + (f / TW) % TH, gw = w0 + f % TW;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gd < a.D && gh < a.H && gw < a.W)
                v = *reinterpret_cast<const bf16x8*>(a.dy + ((((int64_t)n * a.D + gd) * a.H + gh) * a.W + gw) * a.lddy +
                                                     cot * 32 + part * 8);
            sd[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int c = tid + i * 256;
            if (c < HV * 4) *reinterpret_cast<bf16x8*>(&xs[(c >> 2) * 32 + (c & 3) * 8]) = sx[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = tid + i * 256;
            *reinterpret_cast<bf16x8*>(&ds[(c >> 2) * 32 + (c & 3) * 8]) = sd[i];
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
        for (int ks = 0; ks < 16; ks++) {
            // first position of this lane's 8-position half of the k-step: f0 = 16 ks + 8 h
            int hvb, f0;
            if (TW >= 16) {
                constexpr int dummy = 0;
                (void)dummy;
                const int fb = ks * 16;
                hvb = ((fb / (TH * TW)) * HH + (fb / TW) % TH) * WW + fb % TW + 8 * h;
                f0 = fb + 8 * h;
            } else {
                const int fb = ks * 16;   // two rows of 8: row 2ks (h = 0) and 2ks + 1 (h = 1), same d-plane
                hvb = ((fb / (TH * TW)) * HH + (fb / TW) % TH) * WW + h * WW;
                f0 = fb + 8 * h;
            }
            const bf16x8 bfrag = tr_frag(ds + f0 * 32 + lane_off);
#pragma unroll
            for (int t = 0; t < 7; t++) {   // tap 27 (wave 3, t = 6) recomputes tap 26 and is dropped: no branches here
                const bf16x8 afrag = tr_frag(xs + hvb * 32 + toff[t] + lane_off);
                acc[t] = RU3D_MFMA_32X32X16(afrag, bfrag, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave is done with this tile's rows
        if (more) store_tile();
    }
    if (a.dw) {
        // One workgroup holds the pair's whole sum (G == 1: 512 -> 512 on 8^3 has 256 pairs = one workgroup per CU):
        // the 32 x 32 x 27 block goes out in the parameter's own layout dw[co][ci][tap] - per output channel one run
        // of 32 * 27 floats - transposed through LDS in four quarters of 8 output channels (row pitch 868 floats:
        // 16-byte aligned rows, the 8 channel lanes on distinct banks), 16-byte coalesced stores.  No slab, no
        // reduce pass (was 26 + 18 us per layer).
        constexpr int RP = 868;
        static_assert(8 * RP * 4 <= (HV + 256) * 64, "quarter block fits the tile buffer");
        float* const fl = reinterpret_cast<float*>(lds);
        const int co_l = lane & 7, qd_mine = (lane & 31) >> 3;
        for (int qd = 0; qd < 4; qd++) {
            __syncthreads();   // the tile rows (first quarter) / the previous quarter have been read
            if (qd_mine == qd) {
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const int tap = wave + 4 * t;
                    if (tap < 27) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int ci_l = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                            fl[co_l * RP + ci_l * 27 + tap] = acc[t][i];
                        }
                    }
                }
            }
            __syncthreads();
            for (int j = tid; j < 8 * 216; j += 256) {
                const int c = j / 216, r = j - c * 216;
                const f32x4 v = *reinterpret_cast<const f32x4*>(fl + c * RP + r * 4);
                *reinterpret_cast<f32x4*>(a.dw + ((int64_t)(cot * 32 + qd * 8 + c) * a.Cin + cit * 32) * 27 + r * 4) = v;
            }
        }
        return;
    }
    // partial slab: part[((chunk * 27 + tap) * Cin + ci) * Cout + co]; D row = ci, col = co
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

// ---------------------------------------------------------------------------------------------------------
// Weight gradient for the remaining forms (1x1x1 stride 1/2; 3x3x3 stride 2 = pooling conv and ConvTranspose):
// the gathered operand X[s*pos + tap - p] has no compact halo, so each tap's [positions][32 ci] tile is staged
// separately (row-per-position, transposed reads as above).
//   TAPS == 27: tile = 32 flat positions, all 27 tap tiles resident (55 KB); waves split the taps.
//   TAPS == 1 : tile = 256 flat positions; waves split the K (position) range and write one slab each.
struct StagedWgradArgs {
    const bf16* x;
    const bf16* dy;
    float* part;
    int N, Di, Hi, Wi, Do, Ho, Wo;
    int Cin, Cout, ldx, lddy;
    int stride, pad;
    int64_t P;       // N*Do*Ho*Wo
    int ntiles, G;
    int x_cseg;            // split x (planar concat): ci tile t lives in plane t * 32 / x_cseg
    int64_t x_segstride;
};

template <int TAPS, int PT>
__global__ __launch_bounds__(256, 2) void wgrad_staged_mfma_kernel(StagedWgradArgs a) {
    constexpr int K = (TAPS == 27) ? 3 : 1;
    __shared__ __attribute__((aligned(16))) bf16 lds[(TAPS * PT + PT) * 32];
    bf16* xs = lds;                    // [TAPS][PT][32]
    bf16* ds = lds + TAPS * PT * 32;   // [PT][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int COT = a.Cout / 32;
    const int cit = blockIdx.y / COT, cot = blockIdx.y % COT;
    constexpr int NACC = (TAPS == 27) ? 7 : 1;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const int h = lane >> 5, cg = (lane >> 4) & 1, q = (lane & 15) >> 2, p4 = lane & 3;
    const int lane_off = q * 32 + 16 * cg + 4 * p4;
    const int part = tid & 3;
    const bf16* const xbase = a.x_cseg ? a.x + (int64_t)((cit * 32) / a.x_cseg) * a.x_segstride : a.x;
    const int xc0 = a.x_cseg ? (cit * 32) % a.x_cseg : cit * 32;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.G) {
        const int64_t p_base = (int64_t)tile * PT;
        __syncthreads();
        // each thread stages a fixed 16-byte piece column (part) of fixed positions for every tap it owns
        constexpr int POS_PER_THREAD = (PT * 4 + 255) / 256;      // 4 for PT = 256, 1 for PT = 32
#pragma unroll
        for (int j = 0; j < POS_PER_THREAD; j++) {
            const int pl = ((tid >> 2) + 64 * j) % PT;
            const int64_t pos = p_base + pl;
            const bool pv = pos < a.P;
            const int64_t pp = pv ? pos : 0;
            const int ow = (int)(pp % a.Wo);
            int64_t t = pp / a.Wo;
            const int oh = (int)(t % a.Ho);
            t /= a.Ho;
            const int od = (int)(t % a.Do);
            const int n = (int)(t / a.Do);
            // dy piece (only the threads of the first 4*PT ids, i.e. every (pos, part) once)
            if (TAPS == 1 || tid < PT * 4) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (pv) v = *reinterpret_cast<const bf16x8*>(a.dy + pos * a.lddy + cot * 32 + part * 8);
                *reinterpret_cast<bf16x8*>(&ds[pl * 32 + part * 8]) = v;
            }
            // taps owned by this thread: TAPS == 27 -> (tid >> 7) + 2 i ; TAPS == 1 -> tap 0.
            // All gathers are issued first (registers), the LDS writes follow: one exposed latency per tile.
            constexpr int TAP_ITERS = (TAPS == 27) ? 14 : 1;
            bf16x8 sx[TAP_ITERS];
#pragma unroll
            for (int i = 0; i < TAP_ITERS; i++) {
                const int tap = (TAPS == 27) ? ((tid >> 7) + 2 * i) : 0;
                const int kd = tap / (K * K), kh = (tap / K) % K, kw = tap % K;
                const int id = od * a.stride + kd - a.pad, ih = oh * a.stride + kh - a.pad,
                          iw = ow * a.stride + kw - a.pad;
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (tap < TAPS && pv && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi)
                    v = *reinterpret_cast<const bf16x8*>(xbase + ((((int64_t)n * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.ldx +
                                                         xc0 + part * 8);
                sx[i] = v;
            }
#pragma unroll
            for (int i = 0; i < TAP_ITERS; i++) {
                const int tap = (TAPS == 27) ? ((tid >> 7) + 2 * i) : 0;
                if (tap < TAPS) *reinterpret_cast<bf16x8*>(&xs[(tap * PT + pl) * 32 + part * 8]) = sx[i];
            }
        }
        __syncthreads();
        if (TAPS == 27) {
#pragma unroll
            for (int ks = 0; ks < PT / 16; ks++) {
                const int f0 = ks * 16 + 8 * h;
                const bf16x8 bfrag = tr_frag(ds + f0 * 32 + lane_off);
#pragma unroll
                for (int t = 0; t < NACC; t++) {
                    // tap 27 (wave 3, t = 6) recomputes tap 26 and is dropped at the end: no branches between MFMAs
                    const int tap = wave + 4 * t < 27 ? wave + 4 * t : 26;
                    const bf16x8 afrag = tr_frag(xs + (tap * PT + f0) * 32 + lane_off);
                    acc[t] = RU3D_MFMA_32X32X16(afrag, bfrag, acc[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < PT / 64; kk++) {
                const int f0 = (wave * (PT / 64) + kk) * 16 + 8 * h;
                const bf16x8 bfrag = tr_frag(ds + f0 * 32 + lane_off);
                const bf16x8 afrag = tr_frag(xs + f0 * 32 + lane_off);
                acc[0] = RU3D_MFMA_32X32X16(afrag, bfrag, acc[0], 0, 0, 0);
            }
        }
    }
    // slabs: TAPS == 27: one per workgroup (chunk = blockIdx.x); TAPS == 1: one per wave (chunk = 4 blockIdx.x + wave)
#pragma unroll
    for (int t = 0; t < NACC; t++) {
        const int tap = (TAPS == 27) ? wave + 4 * t : 0;
        if (tap < TAPS) {
            const int64_t chunk = (TAPS == 27) ? blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
            float* pp = a.part + (chunk * TAPS + tap) * a.Cin * a.Cout;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ci = cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                const int co = cot * 32 + (lane & 31);
                pp[(int64_t)ci * a.Cout + co] = acc[t][i];
            }
        }
    }
}

static bool wgrad_is_halo_form(const WgradGeom& g) { return g.k == 3 && g.stride == 1; }

static int staged_groups(const WgradGeom& g) {
    const int64_t P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int pt = (g.k == 3) ? 32 : 256;
    const int64_t ntiles = (P + pt - 1) / pt;
    const int pairs = (g.Cin / 32) * (g.Cout / 32);
    int64_t G = 1024 / pairs;
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
    return (int)G;
}

bool mfma_wgrad_eligible(const WgradGeom& g, int dtype) {
    const bool form = (g.k == 3 && (g.stride == 1 || g.stride == 2)) || g.k == 1;
    return dtype == RU3D_BF16 && form && (g.Cin % 32) == 0 && (g.Cout % 32) == 0 && (g.ldx % 8) == 0 &&
           (g.lddy % 8) == 0;
}

static void wgrad_tiles(const WgradGeom& g, int* td, int* th, int* tw) {
    if (g.Wo >= 24) { *td = 2; *th = 4; *tw = 32; }
    else if (g.Wo >= 12) { *td = 2; *th = 8; *tw = 16; }
    else { *td = 4; *th = 8; *tw = 8; }
    // extents that fit none of the tiles (40 x 40 x 20): the tile with the least padded volume, as in s1_plan
    static const int fit_mode = getenv("RU3D_CONV_TILEFIT") ? atoi(getenv("RU3D_CONV_TILEFIT")) : 1;
    if (!fit_mode) return;
    auto cdiv = [](int x, int y) { return (x + y - 1) / y; };
    auto vol = [&](int a, int b, int c) { return (int64_t)cdiv(g.Do, a) * a * cdiv(g.Ho, b) * b * cdiv(g.Wo, c) * c; };
    const int64_t cur = vol(*td, *th, *tw);
    const int64_t v16 = vol(2, 8, 16), v8 = vol(4, 8, 8);
    if (*tw == 32 && v16 * 8 < cur * 7 && v16 <= v8) { *td = 2; *th = 8; *tw = 16; }
    else if (*tw >= 16 && v8 * 8 < cur * 7) { *td = 4; *th = 8; *tw = 8; }
}

static int wgrad_mfma_groups(const WgradGeom& g) {
    int td, th, tw;
    wgrad_tiles(g, &td, &th, &tw);
    const int64_t ntiles = (int64_t)g.N * ((g.Do + td - 1) / td) * ((g.Ho + th - 1) / th) * ((g.Wo + tw - 1) / tw);
    const int pairs = (g.Cin / 32) * (g.Cout / 32);
    int64_t G = 512 / pairs;      // 512 workgroups in all: the tuned optimum (256 / 384 / 768 all slower, round 3)
    if (G < 1) G = 1;
    if (G > ntiles) G = ntiles;
    // one workgroup per CU already and few tiles each (512 -> 512 on 8^3: 256 pairs x 4 tiles): a single workgroup per
    // pair writes the gradient itself (RU3D_WGRAD_DIRECT=0: two slabs + the reduce pass)
    static const int direct = getenv("RU3D_WGRAD_DIRECT") ? atoi(getenv("RU3D_WGRAD_DIRECT")) : 1;
    if (direct && pairs >= 224 && ntiles <= 8) G = 1;
    return (int)G;
}

size_t wgrad_mfma_ws_bytes(const WgradGeom& g) {
    if (wgrad_is_halo_form(g)) {
        const size_t tile = (size_t)wgrad_mfma_groups(g) * 27 * g.Cin * g.Cout * sizeof(float);
        const size_t slide = wgrad_slide_ws_bytes(g);
        return tile > slide ? tile : slide;
    }
    const int slabs = staged_groups(g) * (g.k == 1 ? 4 : 1);
    const size_t staged = (size_t)slabs * g.taps * g.Cin * g.Cout * sizeof(float);
    const size_t s2 = wgrad_s2_ws_bytes(g);
    return staged > s2 ? staged : s2;
}

static int wgrad_staged_launch(const void* x, const void* dy, float* dw, void* ws, const WgradGeom& g, hipStream_t st) {
    StagedWgradArgs a;
    a.x = (const bf16*)x;
    a.dy = (const bf16*)dy;
    a.part = (float*)ws;
    a.N = g.N; a.Di = g.Di; a.Hi = g.Hi; a.Wi = g.Wi; a.Do = g.Do; a.Ho = g.Ho; a.Wo = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.lddy = g.lddy;
    a.stride = g.stride; a.pad = g.pad;
    a.x_cseg = g.x_cseg; a.x_segstride = g.x_segstride;
    if (g.x_cseg && (g.x_cseg % 32)) return ru3d_fail(-1, "wgrad_staged: split x needs segments of whole 32-channel tiles");
    a.P = (int64_t)g.N * g.Do * g.Ho * g.Wo;
    const int pt = (g.k == 3) ? 32 : 256;
    const int64_t ntiles = (a.P + pt - 1) / pt;
    if (ntiles > 0x7fffffff) return ru3d_fail(-1, "wgrad_staged: too many tiles");
    a.ntiles = (int)ntiles;
    a.G = staged_groups(g);
    dim3 grid(a.G, (g.Cin / 32) * (g.Cout / 32));
    int slabs;
    if (g.k == 3) {
        hipLaunchKernelGGL((wgrad_staged_mfma_kernel<27, 32>), grid, dim3(256), 0, st, a);
        slabs = a.G;
    } else {
        hipLaunchKernelGGL((wgrad_staged_mfma_kernel<1, 256>), grid, dim3(256), 0, st, a);
        slabs = a.G * 4;
    }
    int rc = ru3d_check_launch("wgrad_staged_mfma");
    if (rc) return rc;
    return wgrad_reduce_launch((const float*)ws, dw, slabs, g.taps, g.Cin, g.Cout, g.s_o, g.s_i, st);
}

int wgrad_mfma_launch(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes, WgradGeom g,
                      hipStream_t st) {
    const size_t need = wgrad_mfma_ws_bytes(g);
    if (!ws || ws_bytes < need) return ru3d_fail(-1, "wgrad_mfma: workspace too small (%zu < %zu)", ws_bytes, need);
    if (!aligned_to(x, 16) || !aligned_to(dy, 16)) return ru3d_fail(-1, "wgrad_mfma: operands must be 16-byte aligned");
    if (!wgrad_is_halo_form(g)) {
        if (wgrad_s2_eligible(g)) {
            if (g.x_cseg) return ru3d_fail(-1, "wgrad_mfma: no stride-2 kernel takes a split x");
            return wgrad_s2_launch(x, dy, dw, ws, g, st);
        }
        return wgrad_staged_launch(x, dy, dw, ws, g, st);
    }
    {
        WgradSlidePlan sp;
        if (wgrad_slide_plan(g, &sp)) return wgrad_slide_launch(x, dy, dw, ws, g, st);
    }
    if (g.x_cseg) return ru3d_fail(-1, "wgrad_mfma: only the sliding kernel takes a split x");
    MfmaWgradArgs a;
    a.x = (const bf16*)x;
    a.dy = (const bf16*)dy;
    a.part = (float*)ws;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.Cin = g.Cin; a.Cout = g.Cout; a.ldx = g.ldx; a.lddy = g.lddy;
    int td, th, tw;
    wgrad_tiles(g, &td, &th, &tw);
    a.tiles_d = (g.Do + td - 1) / td;
    a.tiles_h = (g.Ho + th - 1) / th;
    a.tiles_w = (g.Wo + tw - 1) / tw;
    a.ntiles = g.N * a.tiles_d * a.tiles_h * a.tiles_w;
    a.G = wgrad_mfma_groups(g);
    // the gradient in the Conv3d layout [co][ci][27], 16-byte aligned: a lone workgroup per pair stores it directly
    a.dw = (a.G == 1 && g.s_i == 27 && g.s_o == (int64_t)g.Cin * 27 && aligned_to(dw, 16)) ? dw : nullptr;
    dim3 grid(a.G, (g.Cin / 32) * (g.Cout / 32));
    if (tw == 32)
        hipLaunchKernelGGL((wgrad3_s1_mfma_kernel<2, 4, 32>), grid, dim3(256), 0, st, a);
    else if (tw == 16)
        hipLaunchKernelGGL((wgrad3_s1_mfma_kernel<2, 8, 16>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((wgrad3_s1_mfma_kernel<4, 8, 8>), grid, dim3(256), 0, st, a);
    int rc = ru3d_check_launch("wgrad3_s1_mfma");
    if (rc || a.dw) return rc;
    return wgrad_reduce_launch((const float*)ws, dw, a.G, 27, g.Cin, g.Cout, g.s_o, g.s_i, st);
}

}  // namespace RU3D_NS
