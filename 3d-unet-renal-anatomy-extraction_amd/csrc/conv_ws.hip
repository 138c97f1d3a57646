// 3x3x3 stride-1 conv of the deepest level (reference network.py:391-403 at 8^3: conv1 / conv2 of the 512-channel
// ResBlocks, forward and input gradient; 10 x 10 x 5 for the reference's own 160 x 160 x 80 patch), whole-sample form.
//
// At this level a sample is a few hundred voxels and a layer's weight is 14 MB: the tile kernels (conv_mfma.hip) cut the
// volume into 256-voxel boxes and every box's workgroups pull their weight slice into the CU again - 4 boxes at 8^3, 12
// at 10 x 10 x 5 (where the boxes are two thirds padding, too), all of it through the ~10 B/clk a CU takes in.  Here a
// workgroup owns (sample, 32 output channels, a slice of the input channels) and ALL voxels of the sample:
//   * the sample's zero-padded volume of one 32-channel chunk sits in LDS ((D+2)(H+2)(W+2) rows of 64 B at a pitch of
//     80 B: the 16 voxels of a fragment read - consecutive rows - start on 16 distinct multiples of four banks),
//     double-buffered: the next chunk's rows are loaded at the top of a chunk's MFMAs and written behind its first tap
//     planes; a fragment's address is the voxel's own row (a register per column tile) plus a per-(chunk, tap) scalar;
//   * v_mfma_f32_16x16x32: output voxels are taken in flat order, 16 to a column tile, tiles dealt round-robin to the 4
//     waves - no box, no padding beyond the last tile; an activation fragment (one ds_read_b128 per lane at the voxel's
//     padded row + the tap's row offset) feeds the two 16-channel weight tiles;
//   * weight fragments come straight from the packed weight (the library's ordinary fragment order) into three register
//     sets, one per kd plane of taps, each refilled for the next chunk right behind its last MFMA;
//   * the slices' fp32 partial outputs go to the caller's workspace and conv_ksplit_reduce_kernel (conv_mfma.hip) sums
//     them in fixed order with bias / residual - the tile kernels' split-K epilogue, unchanged.
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

constexpr int WS_ROWS = 1024;            // padded positions of a sample that fit
constexpr int WS_PITCH = 40;             // 16-bit elements per row: 64 B of channels + 16 B pad (2 x 80 KB = the CU's LDS)
constexpr int WS_MAXT = 8;               // column tiles (16 voxels) per wave: 512 voxels per sample
constexpr int WS_NSTG = WS_ROWS * 4 / 256;   // 16-byte pieces staged per thread and chunk

struct WsArgs {
    const bf16* x;
    const bf16x8* w;
    float* part;
    int N, D, H, W;
    int Cin, Cout, ldx;
    int flip;
    int nchunks;                          // 32-channel chunks of Cin, cut into gridDim.y contiguous slices
    unsigned mWP, mPP, mW, mHW;           // magic multipliers: x / d = (x * m) >> 32 for x < 65536 (d = WP, HP WP, W, H W)
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__global__ __launch_bounds__(256, 1) void conv3_s1_ws_kernel(WsArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 lds[2 * WS_ROWS * WS_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co_b = blockIdx.x * 32, slice = blockIdx.y, n = blockIdx.z;
    const int HP = a.H + 2, WP = a.W + 2;
    const int PV = (a.D + 2) * HP * WP;
    const int V = a.D * a.H * a.W;
    const int ntiles = (V + 15) >> 4;
    const int NTT = a.Cout / 32, KS16 = a.Cin / 16;
    const int kb = lane >> 4;

    // ---- staging plan of this thread: piece c = tid + 256 i = channels 8 q .. 8 q + 7 (q = c & 3) of padded position
    // c >> 2; halo rows read zeros (offset beyond the range)
    int voff[WS_NSTG];
#pragma unroll
    for (int i = 0; i < WS_NSTG; i++) {
        const int c = tid + 256 * i, row = c >> 2;
        const int q = c & 3;
        // (no integer divisions in the prologue)
        const int pd = (int)__umulhi((unsigned)row, a.mPP), rem = row - pd * (HP * WP);
        const int ph = (int)__umulhi((unsigned)rem, a.mWP), pw = rem - ph * WP;
        const bool ok = row < PV && pd >= 1 && pd <= a.D && ph >= 1 && ph <= a.H && pw >= 1 && pw <= a.W;
        voff[i] = ok ? ((((pd - 1) * a.H + ph - 1) * a.W + pw - 1) * a.ldx + q * 8) * 2 : (int)0x80000000;
    }
    const bf16* xs = a.x + (int64_t)n * V * a.ldx;
    const int sample_b = V * a.ldx * 2;
    const int npieces = (PV * 4 + 255) >> 8;          // pieces per thread actually needed (uniform)

    // ---- this lane's output voxels: tile wave + 4 t, voxel (lane & 15); padded row of the voxel at tap (0, 0, 0)
    int prow[WS_MAXT];
#pragma unroll
    for (int t = 0; t < WS_MAXT; t++) {
        int v = (wave + 4 * t) * 16 + (lane & 15);
        if (v >= V) v = 0;                            // idle lanes of the last tile read a valid row; never stored
        const int d = (int)__umulhi((unsigned)v, a.mHW), rem = v - d * (a.H * a.W);
        const int h = (int)__umulhi((unsigned)rem, a.mW), w = rem - h * a.W;
        prow[t] = ((d * HP + h) * WP + w) * WS_PITCH + kb * 8;       // element offset of this lane's k-block in that row
    }

    f32x4 acc[WS_MAXT][2];
#pragma unroll
    for (int t = 0; t < WS_MAXT; t++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weight fragments: set kd holds taps 9 kd .. 9 kd + 8, both 16-channel tiles
    //   element (co, ci) of tap t lives at ((t * KS16 + ci / 16) * NTT + co / 32) * 64 + (co % 32) + 32 * ((ci / 8) & 1)
    bf16x8 wq[3][9][2];
    auto load_w = [&](auto kdc, int ch) {
        constexpr int kd = decltype(kdc)::value;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int tap = kd * 9 + j;
            const int st = a.flip ? 26 - tap : tap;
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int co = co_b + 16 * ct + (lane & 15);
                wq[kd][j][ct] = a.w[((int64_t)(st * KS16 + ch * 2 + (kb >> 1)) * NTT + (co >> 5)) * 64 + (co & 31) + 32 * (kb & 1)];
            }
        }
    };

    bf16x8 stg[WS_NSTG / 2];
    bf16x8 bq[2][WS_MAXT];
    auto load_half = [&](auto hc, int ch) {            // pieces hc * 8 .. hc * 8 + 7 of chunk ch
        constexpr int h0 = decltype(hc)::value * (WS_NSTG / 2);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + ch * 32), (short)0, sample_b, 0x00020000);
#pragma unroll
        for (int i = 0; i < WS_NSTG / 2; i++)
            if (h0 + i < npieces) stg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[h0 + i], 0, 0));
    };
    auto store_half = [&](auto hc, int buf) {
        constexpr int h0 = decltype(hc)::value * (WS_NSTG / 2);
#pragma unroll
        for (int i = 0; i < WS_NSTG / 2; i++)
            if (h0 + i < npieces) {
                const int c = tid + 256 * (h0 + i);
                *reinterpret_cast<bf16x8*>(lds + buf * (WS_ROWS * WS_PITCH) + (c >> 2) * WS_PITCH + (c & 3) * 8) = stg[i];
            }
    };

    // slice s owns chunks [nchunks s / S, nchunks (s + 1) / S): 480 channels = 15 chunks over 8 slices is 2,2,2,2,2,2,2,1
    const int ch0 = (a.nchunks * slice) / (int)gridDim.y;
    const int nch = (a.nchunks * (slice + 1)) / (int)gridDim.y - ch0;
    // prologue: chunk 0 of the slice to buffer 0, its weights to the three sets
    load_half(std::integral_constant<int, 0>{}, ch0);
    load_w(std::integral_constant<int, 0>{}, ch0);
    store_half(std::integral_constant<int, 0>{}, 0);
    load_half(std::integral_constant<int, 1>{}, ch0);
    load_w(std::integral_constant<int, 1>{}, ch0);
    load_w(std::integral_constant<int, 2>{}, ch0);
    store_half(std::integral_constant<int, 1>{}, 0);
    __syncthreads();

    for (int j = 0; j < nch; j++) {
        const int boff = (j & 1) * (WS_ROWS * WS_PITCH);
        const bool more = j + 1 < nch;
        const int chn = ch0 + j + 1;
        if (more) load_half(std::integral_constant<int, 0>{}, chn);
        // activation fragments run one tap ahead of the MFMAs (two register sets): tap k's eight reads are issued between
        // the MFMA pairs of tap k - 1
        auto tap_soff = [&](int tap) {
            // scalar part of the fragment address: buffer + tap row offset (it changes with the chunk, so the 216 lane
            // addresses of a chunk are formed next to their reads instead of being kept as loop invariants)
            return __builtin_amdgcn_readfirstlane(boff + (((tap / 9) * HP + (tap / 3) % 3) * WP + tap % 3) * WS_PITCH);
        };
        {
            const int s0 = tap_soff(0);
#pragma unroll
            for (int t = 0; t < WS_MAXT; t++) bq[0][t] = *reinterpret_cast<const bf16x8*>(lds + prow[t] + s0);
        }
        static_for<0, 3>([&](auto kdc) {
            constexpr int kd = decltype(kdc)::value;
            static_for<0, 9>([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                constexpr int tap = kd * 9 + jj, cur = tap & 1;
                const int sn = tap_soff(tap + 1 < 27 ? tap + 1 : 26);
#pragma unroll
                for (int t = 0; t < WS_MAXT; t++) {
                    // (tiles beyond the sample's last one - prow = row 0 - compute garbage that is never stored: no
                    // control flow between the MFMAs)
                    if constexpr (tap + 1 < 27) bq[cur ^ 1][t] = *reinterpret_cast<const bf16x8*>(lds + prow[t] + sn);
                    acc[t][0] = RU3D_MFMA_16X16X32(wq[kd][jj][0], bq[cur][t], acc[t][0], 0, 0, 0);
                    acc[t][1] = RU3D_MFMA_16X16X32(wq[kd][jj][1], bq[cur][t], acc[t][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // this plane's weight set is free: refill it for the next chunk; the staged halves go to the other buffer
            // behind planes 0 and 1 (their loads have had a plane of MFMAs to land)
            if (more) {
                load_w(kdc, chn);
                if constexpr (kd == 0) {
                    store_half(std::integral_constant<int, 0>{}, (j + 1) & 1);
                    load_half(std::integral_constant<int, 1>{}, chn);
                }
                if constexpr (kd == 1) store_half(std::integral_constant<int, 1>{}, (j + 1) & 1);
            }
        });
        __syncthreads();      // the other buffer is complete; this one may be overwritten by the chunk after next
    }

    // ---- partial outputs: part[slice][n * V + v][Cout]; lane holds channels 4 (lane >> 4) .. + 3 of voxel lane & 15
    const int64_t Vtot = (int64_t)a.N * V;
#pragma unroll
    for (int t = 0; t < WS_MAXT; t++) {
        f32x4 r0 = acc[t][0], r1 = acc[t][1];
        // (wait states between the last 16x16x32 MFMA and a VALU / store read of its result: see conv_s2.hip)
        asm("s_nop 7\n\ts_nop 4" : "+v"(r0), "+v"(r1));
        const int v = (wave + 4 * t) * 16 + (lane & 15);
        if (wave + 4 * t < ntiles && v < V) {
            float* pp = a.part + ((int64_t)slice * Vtot + (int64_t)n * V + v) * a.Cout + co_b + 4 * kb;
            *reinterpret_cast<f32x4*>(pp) = r0;
            *reinterpret_cast<f32x4*>(pp + 16) = r1;
        }
    }
}

}  // namespace

// slices (split-K factor) of the whole-sample kernel for this geometry, 0 = not this kernel's shape
int conv_ws_slices(int N, int D, int H, int W, int Cin, int Cout) {
    static const int mode = getenv("RU3D_CONV_WS") ? atoi(getenv("RU3D_CONV_WS")) : 1;
    if (!mode || (Cin % 32) || (Cout % 32)) return 0;
    const int64_t V = (int64_t)D * H * W, PV = (int64_t)(D + 2) * (H + 2) * (W + 2);
    if (V > WS_MAXT * 64 || PV > WS_ROWS || V < 64) return 0;
    if (W < 2 || H * W < 2) return 0;      // the kernel's magic-number divisions hold for divisors >= 2
    // weights dominate: the point of the form is to read them once per sample
    if ((int64_t)Cin * Cout < 256 * 256) return 0;
    // ... which pays where the tile kernels cut a sample into many boxes: 8^3 is two 4 x 8 x 8 boxes (26.8 us there, 32.5
    // here), 10 x 10 x 5 is six, two thirds padding (55.9 us there, 30.1 here).  RU3D_CONV_WS=2: wherever the shape fits
    const int boxes = ((D + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8);
    if (mode < 2 && boxes < 4) return 0;
    const int nchunks = Cin / 32, units = N * (Cout / 32);
    if (units > 512) return 0;
    int ks = ru3d_get_cu_budget() / units;    // one workgroup per CU in total (its LDS is the CU's)
    if (ks > nchunks) ks = nchunks;
    if (ks < 1 || units * ks < 128) return 0;  // too few workgroups: the tile kernels' split does better
    return ks;
}

int conv_ws_launch(const void* x, const void* w, float* part, int N, int D, int H, int W, int Cin, int Cout, int ldx,
                   int flip, int slices, hipStream_t st) {
    WsArgs a;
    a.x = (const bf16*)x;
    a.w = (const bf16x8*)w;
    a.part = part;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.Cin = Cin; a.Cout = Cout; a.ldx = ldx;
    a.flip = flip;
    a.nchunks = Cin / 32;
    auto magic = [](int d) { return (unsigned)((0x100000000ull / (unsigned)d) + 1); };
    a.mWP = magic(W + 2); a.mPP = magic((H + 2) * (W + 2)); a.mW = magic(W); a.mHW = magic(H * W);
    if ((int64_t)D * H * W * ldx * 2 >= (1ll << 31)) return ru3d_fail(-1, "conv_ws: sample too large");
    hipLaunchKernelGGL(conv3_s1_ws_kernel, dim3(Cout / 32, slices, N), dim3(256), 0, st, a);
    return ru3d_check_launch("conv3_s1_ws");
}

}  // namespace RU3D_NS
