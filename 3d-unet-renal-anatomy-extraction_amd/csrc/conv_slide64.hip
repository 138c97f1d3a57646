// 3x3x3 stride-1 conv, 64 -> 64 (or 128) channels, 16-bit MFMA: the D-sliding plane-ring design of conv_slide.hip
// for the 64-channel level (reference network.py:391-403: conv1 / conv2 of the level-1 ResBlocks at 64^3, forward and
// input gradient).
//
// The producer/consumer kernel that served this shape pulls 110 KB of weights + 65 KB of halo into the CU per
// (256-voxel tile, 32-channel chunk) item - 25 bytes per MFMA clock against the ~10 a CU ingests - and ran at a
// third of the matrix peak.  The sliding form keeps all weights resident and re-reads each input voxel 1.6x:
//   * v_mfma_f32_16x16x32: a wave owns 16 output channels and ALL of K.  Its weights - 27 taps x 2 k-steps of
//     (16 couts x 32 cin) - are 54 fragments x 4 registers = 216 registers, loaded once per persistent workgroup
//     (one wave per SIMD, 512 registers; all of them pinned to AGPRs through the inline-asm operand classes,
//     accumulators in VGPRs).  The four waves of a workgroup cover 64 output channels; Cout = 128 runs as two slices (grid.y);
//   * the workgroup owns a (4 x 32) column in (H, W) and slides along D over a ring of 4 input planes in LDS
//     (6 x 34 halo rows x 128 B of channels, pitch 160 B: conflict-free for the 16-voxel x 4 k-block fragment reads),
//     one new 26 KB plane per 128 output voxels: ~4 bytes per MFMA clock enter the CU;
//   * every wave reads every activation fragment (they differ in output channels, not in voxels): a (kd, kw, k-step)
//     group walks the six input rows once - 12 fragment reads for 24 MFMAs, row r's registers refilled for the next
//     group right behind the last MFMA that reads row r;
//   * the epilogue of plane s (wave-private fp32 LDS patch -> 16-byte stores, bias, residual, fused InstanceNorm sums)
//     and the staging of plane s+3 / the loads of plane s+4 are spread over the MFMA groups of plane s+1; one barrier
//     per plane.
// Packed weights are the library's ordinary 32x32x16 fragment order; the 16x16x32 fragments are gathered from it
// (16-byte pieces) in the prologue, so there is no second pack format.
#include "common.h"
#include "conv.h"

#include <type_traits>

namespace RU3D_NS {
namespace {
constexpr int TH = 4, TW = 32, HH = TH + 2, WW = TW + 2;
constexpr int PROWS = HH * WW;            // 204 halo rows per plane
constexpr int PITCH = 80;                 // bf16 elements per staged row: 128 B of channels + 32 B pad
constexpr int PLANE = PROWS * PITCH;      // elements per plane (32,640 B)
constexpr int RING = 4;
constexpr int NSTG = 7;                   // 16-byte pieces staged per thread and plane: ceil(204 * 8 / 256)
constexpr int NG = 18;                    // (kd, kw, k-step) groups per step, 24 MFMAs each
constexpr int EP = 24;                    // 16-bit elements per epilogue-patch row (16 couts + 8 pad = 48 B)
static_assert(RING * PLANE * 2 + 4 * 128 * EP * 2 <= 160 * 1024, "LDS budget");

struct Slide64Args {
    const bf16* x;
    const bf16x8* w;
    const float* bias;
    const bf16* res;
    bf16* y;
    float* stat_slab;
    int N, D, H, W;
    int ldx, ldy, ldr;
    int xsplit;             // planar concat input: channels 32..63 live xdelta_b bytes behind channels 0..31 (pitch ldx = 32)
    int xdelta_b;
    int flip;
    int cout_total;                       // Cout of the conv (64 per grid.y slice)
    int tiles_h, tiles_w, dsplit, DL, units;
    int light_last;                       // EDGE: the last column's far W half lies outside the volume (W % 32 in 1..16)
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifdef RU3D_STORAGE_F16
#define RU3D_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define RU3D_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
// Operand classes: ALL 54 weight fragments live in AGPRs (216 of the 256), accumulators and activation fragments in
// VGPRs - the MFMA takes its A operand from an AGPR directly.  With every weight on the "a" side the register allocator
// has no fragment to park elsewhere and copy back next to its use (a v_accvgpr_write into an MFMA operand right in front
// of the MFMA is a hazard the compiler's recogniser cannot see for inline asm: a build with 48 fragments + the
// accumulators in AGPRs returned stale data on the 64^3 shapes).  AGPRs 216.. are the allocator's spill space for
// loop-invariant VGPRs; tools/isa_check.py checks that no v_accvgpr_write between the MFMAs targets a0..a215.
template <bool ZERO>
__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& w, const bf16x8& x) {
    const i32x4 wi = __builtin_bit_cast(i32x4, w), xi = __builtin_bit_cast(i32x4, x);
    if constexpr (ZERO) asm volatile(RU3D_MFMA16_ASM " %0, %1, %2, 0" : "=v"(acc) : "a"(wi), "v"(xi));
    else asm volatile(RU3D_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc) : "a"(wi), "v"(xi));
}

// VG = 1: the four waves are four groups of 16 output channels (64 per workgroup), each wave walks both W halves of
// the column.  VG = 2 (Cout = 32: the decoder's 64 -> 32 conv on the full-resolution level): two channel groups x two
// voxel groups - a wave owns one W half, so the workgroup still keeps four matrix pipes busy on 32 output channels.
// EDGE: W is not a multiple of 32 (config 4: 160 x 160 x 80 gives W = 40 on this level): the last column of a row is
// partly outside the volume.  Its staged pieces already come back as zeros; the epilogue masks stores, residual reads and
// statistics per voxel, and a W half that lies outside entirely issues no MFMAs (its passes still carry their share of
// the staging and of the previous plane's epilogue).  A separate instantiation: the W % 32 == 0 code is unchanged.
// (Rounds 2-3 carried a HAS_BST variant - the InstanceNorm-backward sums of the following norm from this kernel's row phase, as
// in conv_slide32.hip.  It spilled 57 registers outside its loop and measured time-neutral to slightly slower in two
// same-box A/Bs (round 4: 16.39 vs 16.25 ms with it on); removed.)
template <bool HAS_RES, bool HAS_STATS, int VG, bool EDGE = false>
__global__ __launch_bounds__(256, 1) void conv3_s1_slide64_kernel(Slide64Args a) {
    constexpr bool SUMS = HAS_STATS;                   // st1 / st2 and the slab are in use
    constexpr bool LOADS = HAS_RES;                    // a second tensor is read in the row phase
    __shared__ __attribute__((aligned(16))) bf16 lds[RING * PLANE];
    __shared__ __attribute__((aligned(16))) bf16 est_s[4 * 128 * EP];   // epilogue patches (stored values), one per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NHF = VG == 1 ? 2 : 1;               // W halves a wave walks
    constexpr int CB = VG == 1 ? 64 : 32;              // output channels per workgroup
    const int cg = VG == 1 ? wave : (wave & 1);        // channel group of 16
    const int hfw = VG == 1 ? 0 : (wave >> 1);         // VG = 2: the wave's W half
    const int co_w = blockIdx.y * CB + cg * 16;        // first output channel of this wave

    // ---- weights: fragment (tap, ks32) of this wave's 16 couts, gathered from the 32x32x16 fragment order
    //   element (co, ci) of tap t lives at ((t * KS16 + ci / 16) * NTT + co / 32) * 64 + (co % 32) + 32 * ((ci / 8) & 1)
    //   16x16x32 A fragment: lane -> co = co_w + (lane & 15), ci = 32 * ks + 8 * (lane >> 4) + j
    bf16x8 wreg[54];
    {
        const int NTT = a.cout_total / 32;
        const int co = co_w + (lane & 15);
        const int kb = lane >> 4;                      // k-block of 8 inside the 32-deep k-step
        static_for<0, 54>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int tap = f >> 1, ks = f & 1;
            const int st = a.flip ? 26 - tap : tap;
            const int ks16 = 2 * ks + (kb >> 1);
            wreg[f] = a.w[((st * 4 + ks16) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (kb & 1)];
        });
    }

    // ---- staging: piece c = tid + 256 i of a plane is halo row c >> 3, 16-byte piece c & 7.  Nothing per piece is
    // kept in registers across the step loop except its global offset and validity (recomputing the LDS offset costs
    // two VALU ops per piece; the register file is the scarce resource here)
    auto piece_row = [&](int i) { return (tid + 256 * i) >> 3; };
    auto piece_valid = [&](int i) { return tid + 256 * i < PROWS * 8; };
    auto piece_dst = [&](int i) { return piece_row(i) * PITCH + ((tid + 256 * i) & 7) * 8; };
    // ---- fragment address of this lane: voxel (lane & 15) of a 16-voxel W-run, k-block lane >> 4; row, half, kw,
    // k-step and plane are compile-time offsets
    const bf16* bl = lds + ((lane & 15) + 16 * hfw) * PITCH + (lane >> 4) * 8;

    // fused InstanceNorm statistics: slab[workgroup][wave][n][CB][2]; a wave fills its own 16 channels, the rest of
    // its row stays at the zeros it wrote when the kernel started
    float st1[8], st2[8];
    int cur_n = -1;
    if (SUMS) ru3d_clear_own_slab_rows(a.stat_slab, (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave, a.N, CB * 2);
    auto stat_flush = [&]() {
        if (!SUMS || cur_n < 0) return;
        float* dst = a.stat_slab + ((((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * a.N + cur_n) * CB) * 2;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float s1 = st1[i], s2 = st2[i];
#pragma unroll
            for (int o = 2; o < 64; o <<= 1) {       // lanes with equal (lane & 1) hold the same 8 channels
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lane < 2) {
                const int c = cg * 16 + lane * 8 + i;
                if constexpr (EDGE) {       // heavy-first unit order: a workgroup may come back to a sample (slab rows start at zero)
                    dst[c * 2] += s1;
                    dst[c * 2 + 1] += s2;
                } else {
                    dst[c * 2] = s1;
                    dst[c * 2 + 1] = s2;
                }
            }
        }
    };

    f32x4 acc[TH][NHF];         // [output row][W half]: one set - a tile goes to the patch as soon as its last MFMA is issued
    bf16x8 xq[HH];              // activation fragments of one (group, W half): input rows 0..5
    bf16* est = est_s + wave * (128 * EP);
    // bias of the 4 channels this lane holds in the accumulator layout
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; i++) bias4[i] = a.bias ? a.bias[co_w + 4 * (lane >> 4) + i] : 0.f;
    bf16* const ybase = a.y + co_w + (lane & 1) * 8;
    const bf16* const rbase = a.res + co_w + (lane & 1) * 8;

    const int G = gridDim.x;
    const bool remap = (a.units % 8) == 0 && (G % 8) == 0;
    for (int ui = blockIdx.x; ui < a.units; ui += G) {
        int u = remap ? (ui % 8) * (a.units / 8) + ui / 8 : ui;   // XCD-contiguous deal (see conv_slide.hip)
        int tw_i;
        if (EDGE && a.light_last) {
            // the last column of a row issues half the MFMAs (its far W half is outside): all full columns first, the light
            // ones behind them, so that the second round of a launch with 1 < units / workgroups < 2 is made of light units
            const int heavy = a.units / a.tiles_w * (a.tiles_w - 1);
            u = ui;
            if (u < heavy) {
                tw_i = u % (a.tiles_w - 1);
                u /= (a.tiles_w - 1);
            } else {
                tw_i = a.tiles_w - 1;
                u -= heavy;
            }
        } else {
            tw_i = u % a.tiles_w;
            u /= a.tiles_w;
        }
        const int th_i = u % a.tiles_h;
        u /= a.tiles_h;
        const int dc = u % a.dsplit;
        const int n = u / a.dsplit;
        const int d0 = dc * a.DL, h0 = th_i * TH, w0 = tw_i * TW;
        const int wlim = a.W - w0;                       // EDGE: voxels of this column's rows inside the volume
        const bool far_ok = !EDGE || wlim > 16;          // the second W half holds at least one of them
        const bool wave_ok = !EDGE || VG == 1 || 16 * hfw < wlim;     // VG = 2: this wave's half

        // halo pieces of this column as byte offsets inside a plane; a piece outside the volume carries an offset beyond
        // the buffer's range and the buffer load returns zeros, a plane outside the sample is switched off through the
        // record count: no selects, no address clamps in the loop
        const int plane_b = a.H * a.W * a.ldx * 2;
        const int sample_b = a.D * plane_b;
        int voff[NSTG];
#pragma unroll
        for (int i = 0; i < NSTG; i++) {
            const int r = piece_row(i), zh = r / WW, zw = r - zh * WW;
            const int gh = h0 - 1 + zh, gw = w0 - 1 + zw;
            const bool okv = piece_valid(i) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            // (a thread's piece index is the same for every i: 256 i is a multiple of 8.  Split input: the voxel row is two
            // 64-byte halves in two planes of one buffer - pieces 4..7 carry the planes' distance in their offset)
            const int pc = (tid + 256 * i) & 7;
            const int coff = a.xsplit ? (pc >> 2) * a.xdelta_b + (pc & 3) * 16 : pc * 16;
            voff[i] = okv ? ((gh * a.W + gw) * a.ldx) * 2 + coff : (int)0x80000000;
        }
        const bf16* xs = a.x + (int64_t)n * a.D * (plane_b / 2);
        const int range_b = a.xsplit ? a.xdelta_b + sample_b : sample_b;

        bf16x8 stg[NSTG];
        auto load_piece = [&](int pr, auto ic) {
            constexpr int i = decltype(ic)::value;
            const int d = d0 - 1 + pr;
            const bool dok = d >= 0 && d < a.D;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xs, (short)0, dok ? range_b : 0, 0x00020000);
            stg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], dok ? d * plane_b : 0, 0));
        };
        auto store_piece = [&](int pr, int slot, auto ic) {
            constexpr int i = decltype(ic)::value;
            if (piece_valid(i)) *reinterpret_cast<bf16x8*>(lds + slot * PLANE + piece_dst(i)) = stg[i];
        };
        auto load_plane = [&](int pr) { static_for<0, NSTG>([&](auto ic) { load_piece(pr, ic); }); };
        auto store_plane = [&](int pr, int slot) { static_for<0, NSTG>([&](auto ic) { store_piece(pr, slot, ic); }); };

        __syncthreads();   // the previous unit has left the ring
        load_plane(0); store_plane(0, 0);
        load_plane(1); store_plane(1, 1);
        load_plane(2); store_plane(2, 2);
        load_plane(3);
        __syncthreads();

        if (SUMS && n != cur_n) {
            stat_flush();
            cur_n = n;
#pragma unroll
            for (int i = 0; i < 8; i++) st1[i] = st2[i] = 0.f;
        }

        // ---- activation fragments of group g = (kd, kw, ks) of the step with ring phase PHN: input row r, W half hf
        // a "pass" is one (group, W half): q = 2 g + hf, 36 per step, 12 MFMAs each
        auto frag_row = [&](auto phn, auto qc, auto rc) {
            constexpr int PHN = decltype(phn)::value, q = decltype(qc)::value, r = decltype(rc)::value;
            constexpr int g = q / NHF, hf = q % NHF;
            constexpr int kd = g / 6, kw = (g % 6) >> 1, ks = g & 1;
            xq[r] = *reinterpret_cast<const bf16x8*>(bl + ((PHN + kd) & 3) * PLANE + (r * WW + kw + 16 * hf) * PITCH + ks * 32);
        };

        // ---- epilogue.  Tile (m, hf) of the plane being computed is complete behind its last MFMA (tap row kh = 2, in the
        // step's last two passes): bias is added, the value rounded to the storage type and written to the wave-private
        // patch [128 voxels][16 couts] (accumulator layout: lane = voxel of the 16-run, 4 couts per lane).  The row
        // phase - lane pairs read a voxel's 2 x 8 channels, statistics, residual, 16-byte stores - is spread over the
        // first passes of the NEXT step.
        auto epi_write = [&](int m, int hf) {
            bf16x4 v;
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (bf16)(acc[m][hf][i] + bias4[i]);
            *reinterpret_cast<bf16x4*>(est + ((hf * TH + m) * 16 + (lane & 15)) * EP + 4 * (lane >> 4)) = v;
        };
        auto epi_vox = [&](int sp, int p) {     // p = 0..2 NHF - 1: patch rows 32 p + (lane >> 1); row = (hf * 4 + m) * 16 + w16
            const int row = 32 * p + (lane >> 1);
            const int hf = VG == 1 ? (row >> 6) : hfw, m = (row >> 4) & 3;
            return (((int64_t)n * a.D + d0 + sp) * a.H + h0 + m) * (int64_t)a.W + w0 + 16 * hf + (row & 15);
        };
        auto epi_row_load = [&](int p, bf16x8& rv) {
            rv = *reinterpret_cast<const bf16x8*>(est + (32 * p + (lane >> 1)) * EP + (lane & 1) * 8);
        };
        auto epi_in = [&](int p) {              // EDGE: is the voxel of patch row 32 p + (lane >> 1) inside the volume
            const int row = 32 * p + (lane >> 1);
            const int hf = VG == 1 ? (row >> 6) : hfw;
            return 16 * hf + (row & 15) < wlim;
        };
        auto epi_row_finish = [&](int sp, int p, const bf16x8& rv, const bf16x8& rres) {
            if constexpr (EDGE) {
                if (!epi_in(p)) return;
            }
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (float)rv[i];                         // the stored value of conv + bias
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
                store_vec<bf16, 8>(ybase + epi_vox(sp, p) * a.ldy, v);
            } else {
                *reinterpret_cast<bf16x8*>(ybase + epi_vox(sp, p) * a.ldy) = rv;
            }
        };
        // the row phase of one plane as 4 NHF pieces: (load, finish) x 2 NHF
        constexpr int NEP = 4 * NHF, NRQ = 2 * NHF;
        auto epi_piece = [&](auto pc, int sp, bf16x8& rv, const bf16x8 (&rq)[NRQ]) {
            constexpr int p = decltype(pc)::value;
            if constexpr ((p & 1) == 0) epi_row_load(p >> 1, rv);
            else epi_row_finish(sp, p >> 1, rv, rq[(p >> 1) < NRQ ? (p >> 1) : 0]);
        };

        auto step = [&](auto phc, int s) {
            constexpr int PH = decltype(phc)::value;
            const bool has_prev = s > 0, last = s == a.DL - 1;
            bf16x8 rv;
            bf16x8 rq[NRQ];
            if constexpr (LOADS) {
                if (has_prev) {
#pragma unroll
                    for (int k = 0; k < NRQ; k++)
                        if (!EDGE || epi_in(k)) rq[k] = *reinterpret_cast<const bf16x8*>(rbase + epi_vox(s - 1, k) * a.ldr);
                }
            }
            // wait states in front of the inline-asm MFMA block (see conv_slide.hip)
            asm volatile("s_nop 7\n\ts_nop 7");
            static_for<0, NHF * NG>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int g = q / NHF, hf = q % NHF;
                constexpr int kd = g / 6, kw = (g % 6) >> 1, ks = g & 1;
                // LDS-only: a __syncthreads() would also drain vmcnt, i.e. wait for the epilogue stores just issued
                if constexpr (q == 11 * NHF) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (in front of group 11) plane s+2, stored during the previous
                                                          // step, is complete; every wave is past kd = 0: the slot of
                                                          // plane s-1 is free
                // walk the six input rows; row r feeds output rows m = r - kh (kh = 0..2)
                static_for<0, HH>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    static_for<0, 3>([&](auto khc) {
                        constexpr int kh = decltype(khc)::value;
                        constexpr int m = r - kh;
                        if constexpr (m >= 0 && m < TH) {
                            constexpr int f = ((kd * 3 + kh) * 3 + kw) * 2 + ks;
                            if constexpr (!EDGE || (VG == 1 && hf == 0)) {
                                mfma16<(g == 0 && kh == 0)>(acc[m][hf], wreg[f], xq[r]);
                            } else {
                                if (VG == 1 ? far_ok : wave_ok) mfma16<(g == 0 && kh == 0)>(acc[m][hf], wreg[f], xq[r]);
                            }
                        }
                    });
                    // Finished tiles go to the patch (its rows of the previous plane were consumed in passes 0..14).  Tile
                    // m is complete behind row m + 2; its VALU reads are placed behind the MFMAs of the NEXT row - the
                    // wait states an MFMA result needs before a VALU read are not inserted for inline asm - and pinned
                    // there by the empty asm; the very last tile waits on explicit nops.
                    if constexpr (g == NG - 1) {
                        if constexpr (r >= 3) {
                            asm volatile("" : "+v"(acc[r - 3][hf]));
                            epi_write(r - 3, hf);
                        }
                        if constexpr (NHF == 2 && hf == 1 && r == 1) {
                            asm volatile("" : "+v"(acc[3][0]));
                            epi_write(3, 0);
                        }
                        if constexpr (hf == NHF - 1 && r == 5) {
                            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[3][hf]));
                            epi_write(3, hf);
                        }
                    }
                    // row r is free: refill it for the next pass (also in front of an EDGE pass that issues no MFMAs: a
                    // branch around the LDS reads cost more than the reads - 204 against 189 us on 2 x 80 x 80 x 40)
                    if constexpr (q + 1 < NHF * NG) {
                        frag_row(std::integral_constant<int, PH>{}, std::integral_constant<int, q + 1>{}, rc);
                    } else {
                        if (!last) frag_row(std::integral_constant<int, (PH + 1) & 3>{}, std::integral_constant<int, 0>{}, rc);
                    }
                    // plane staging behind the barrier: piece i of plane s+3 goes to LDS in pass 22 + 2 i (row 1), piece i
                    // of plane s+4 is loaded into the freed registers one pass later (row 3); unconditional (clamped)
                    if constexpr (r == 1 && hf == 0 && g >= 11 && g < 11 + NSTG) store_piece(s + 3, (PH + 3) & 3, std::integral_constant<int, (g >= 11 && g < 11 + NSTG ? g - 11 : 0)>{});
                    if constexpr (r == 3 && hf == NHF - 1 && g >= 11 && g < 11 + NSTG) load_piece(s + 4, std::integral_constant<int, (g >= 11 && g < 11 + NSTG ? g - 11 : 0)>{});
                    // row phase of the previous plane: its pieces over the first groups (row 4)
                    if constexpr (r == 4 && hf == 0 && g < NEP) {
                        if (has_prev) epi_piece(std::integral_constant<int, g>{}, s - 1, rv, rq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        };

        // fragments of the first group of step 0
        static_for<0, HH>([&](auto rc) { frag_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rc); });
        for (int s4 = 0; s4 < a.DL; s4 += 4) {
            step(std::integral_constant<int, 0>{}, s4);
            step(std::integral_constant<int, 1>{}, s4 + 1);
            step(std::integral_constant<int, 2>{}, s4 + 2);
            step(std::integral_constant<int, 3>{}, s4 + 3);
        }
        // the last plane of the unit has no next step to hide behind
        {
            bf16x8 rq[NRQ];
            bf16x8 rv;
#pragma unroll
            for (int k = 0; k < NRQ; k++) {
                const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                rq[k] = (LOADS && (!EDGE || epi_in(k))) ? *reinterpret_cast<const bf16x8*>(rbase + epi_vox(a.DL - 1, k) * a.ldr) : z8;
            }
            static_for<0, NEP>([&](auto pc) { epi_piece(pc, a.DL - 1, rv, rq); });
        }
    }
    stat_flush();
}
}  // namespace

// units = N x dsplit x (H/4) x (W/32) columns of DL = D/dsplit planes, times Cout/64 output slices (grid.y)
bool slide64_conv_plan(int N, int D, int H, int W, int Cin, int Cout, SlidePlan* out) {
    static const int mode = getenv("RU3D_CONV_SLIDE64") ? atoi(getenv("RU3D_CONV_SLIDE64")) : 1;
    if (mode == 0 || Cin != 64 || (Cout != 32 && Cout != 64 && Cout != 128) || (H % TH) || W < 8 || D < 4) return false;
    static const int edge_mode = getenv("RU3D_SLIDE64_EDGE") ? atoi(getenv("RU3D_SLIDE64_EDGE")) : 1;
    if ((W % TW) && (!edge_mode || (W % TW) <= 4)) return false;   // a sliver of a last column is not worth a column's work
    // the staging loads address a sample with 30-bit element offsets; the input may sit in a buffer of twice its channels
    // (the decoder's concat): shapes that could exceed that go to the other kernels consistently (launch, slab, workspace)
    if ((int64_t)D * H * W * Cin * 2 >= (1ll << 30)) return false;
    const int ny = Cout == 32 ? 1 : Cout / 64;
    const int tw_n = (W + TW - 1) / TW;
    const bool light = (W % TW) != 0 && (W % TW) <= 16 && tw_n > 1;
    const int64_t cols = (int64_t)N * (H / TH) * tw_n;
    int64_t best_cost = -1;
    int best = 0;
    for (int ds = 1; ds <= D / 4; ds++) {
        if (D % ds) continue;
        const int dl = D / ds;
        if (dl % 4) continue;
        const int64_t units = cols * ds;
        if (units * ny > 0x7fffffff) break;
        int64_t gx = ru3d_get_cu_budget() / ny;
        if (gx > units) gx = units;
        int64_t cost = ((units + gx - 1) / gx) * (dl + 4);
        if (light) {
            // heavy-first order (see the kernel): workgroup 0 carries the longest chain - units 0, gx, 2 gx, ... - of
            // which those beyond `heavy` issue about half the MFMAs
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
    const double ideal = (double)cols * ny * D / (double)ru3d_get_cu_budget();
    if (units * ny < 128 || (double)best_cost > 1.7 * ideal + 8) return false;
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

int conv_slide64_launch(const void* x, const void* w, const float* bias, const void* res, void* y, const ConvGeom& g,
                        float* stat_slab, hipStream_t st, const void* bst_act, int bst_ld, float slope) {
    SlidePlan p;
    if (!slide64_conv_plan(g.N, g.Do, g.Ho, g.Wo, g.Cin, g.Cout, &p))
        return ru3d_fail(-1, "conv_slide64: shape not supported");
    if ((int64_t)g.Do * g.Ho * g.Wo * g.ldx >= (1ll << 30)) return ru3d_fail(-1, "conv_slide64: sample too large");
    if (res && stat_slab) return ru3d_fail(-1, "conv_slide64: residual and fused statistics cannot be combined");
    if (bst_act) return ru3d_fail(-1, "conv_slide64: no fused backward sums on the 64-channel kernel");
    Slide64Args a;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.bias = bias;
    a.res = (const bf16*)res;
    a.y = (bf16*)y;
    a.stat_slab = stat_slab;
    a.N = g.N; a.D = g.Do; a.H = g.Ho; a.W = g.Wo;
    a.ldx = g.ldx; a.ldy = g.ldy; a.ldr = g.ldr;
    a.xsplit = 0; a.xdelta_b = 0;
    if (g.x_cseg) {
        // the two 32-channel halves of the concat as planes of one buffer; all offsets stay below 2^31 (top bit = "outside")
        const int64_t delta_b = g.x_segstride * 2, sample_b = (int64_t)g.Do * g.Ho * g.Wo * g.ldx * 2;
        if (g.x_cseg != 32 || g.Cin != 64 || g.ldx < 32 || delta_b <= 0 || delta_b + sample_b >= (1ll << 31) || (delta_b % 16))
            return ru3d_fail(-1, "conv_slide64: split input must be two 32-channel planes less than 2 GiB apart");
        a.xsplit = 1;
        a.xdelta_b = (int)delta_b;
    }
    if (g.y_cseg) return ru3d_fail(-1, "conv_slide64: split output not supported");
    a.flip = g.flip;
    a.cout_total = g.Cout;
    a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w; a.dsplit = p.dsplit; a.DL = p.DL; a.units = p.units;
    (void)bst_ld; (void)slope;
    a.light_last = (g.Wo % TW) != 0 && (g.Wo % TW) <= 16 && p.tiles_w > 1;
    const dim3 grid(p.grid, p.ny), block(256);
#define RU3D_S64_LAUNCH(VGV, EDGEV)                                                                                      \
    do {                                                                                                                 \
        if (res) hipLaunchKernelGGL((conv3_s1_slide64_kernel<true, false, VGV, EDGEV>), grid, block, 0, st, a);          \
        else if (stat_slab) hipLaunchKernelGGL((conv3_s1_slide64_kernel<false, true, VGV, EDGEV>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((conv3_s1_slide64_kernel<false, false, VGV, EDGEV>), grid, block, 0, st, a);             \
    } while (0)
    const bool edge = (g.Wo % TW) != 0;
    if (g.Cout == 32) {
        if (edge) RU3D_S64_LAUNCH(2, true);
        else RU3D_S64_LAUNCH(2, false);
    } else {
        if (edge) RU3D_S64_LAUNCH(1, true);
        else RU3D_S64_LAUNCH(1, false);
    }
#undef RU3D_S64_LAUNCH
    return ru3d_check_launch("conv3_s1_slide64");
}

}  // namespace RU3D_NS
