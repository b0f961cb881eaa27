// dsg_stream.h -- kernel set DSG_KSET_STREAM: weight-stationary, persistent GEMMs for large batches (bf16).
//
// Why (round-2 profile of the BLOCK set, profiles/r02_j_b64_kernel_stats.csv): at 5696 rows LN + QKV took 23.8 us, linear1 12.8-22.9,
// linear2 16.4 and the pose head 54.9 while moving ~180 MB per GEMM through the CUs' load paths -- 32 x 64 output blocks re-read an
// activation block 12-18 times (normalising it each time) and a weight slice 178 times (16 FLOP per byte pulled into a CU), and
// every fragment went through VGPRs.  Here:
//   * WEIGHTS ARE STATIONARY IN REGISTERS.  A workgroup owns one 128-column panel of W for its whole life; each of its 4 waves
//     (2 x 2 over a 64-row x 128-column block) keeps its 64 columns x K = 256 slice as v_mfma_f32_32x32x16_bf16 operand fragments
//     (128 VGPRs), loaded ONCE, and walks the row blocks  mb = group, group + G, ...  (persistent: grid = panels x groups ~ 2
//     workgroups per CU).  Per 64 x 128 block a CU pulls 32 KB of activations for 4.2 MFLOP: 128 FLOP per byte.
//   * ACTIVATIONS GO GLOBAL -> LDS DIRECTLY (global_load_lds_dwordx4, no VGPR staging), the whole block in one batch of 8
//     instructions per wave, DOUBLE BUFFERED: the next block's loads are issued before the current block's MFMA loop.  The A
//     operand is stored fragment-major by its producer (k_attn_op / k_ln_frag / the GELU epilogue), so a block is one contiguous
//     span and the LDS image is conflict-free for ds_read_b128 as it lies.
//   * LAYERNORM ONCE PER ROW (k_ln_frag): the LayerNorm-on-read GEMMs (QKV of layers > 0, pose head) get their operand from a
//     16-rows-per-workgroup pass that writes the normalised rows in bf16, fragment-major (+ fp32 for the attention kernel's
//     residual) -- one extra dispatch (5.5 us at 5696 rows) instead of 64 fp32 rows staged through registers per (block, panel).
//   * 32 x 32 x 16 MFMA: half the LDS operand bytes per FLOP of the 16 x 16 x 32 form (one 1 KB A fragment feeds two MFMAs of
//     32 K FLOP each), 4 waves x 1 KB per 64 cycles = 64 B/clk of the CU's 256 B/clk LDS read rate.
//   * linear2 (K = ff = 1024, k_ws2): the 4 waves split K; each keeps W[64 columns x its 256-wide K quarter] in registers, the 32-row
//     block of `hidden` (64 KB, fragment-major, contiguous) is double buffered in LDS, the four partial 32 x 64 blocks are reduced
//     through the retired buffer in a fixed order.
//   * V^T IN ALIGNED TOKEN GROUPS (vt_store_block, dsg_kernels.h): a V panel's block is transposed through the retired activation
//     buffer so that every 4-token group of a batch element is one 8-byte store (ntok = 89: three of four batch elements paid 4
//     element stores per lane and row quad -- QKV 16.8 -> 12.5 us at 5696 rows; the block kernels of dsg_batched.h do the same).
//   * the pose head (EPI_OUT) transposes each wave's 32 x 32 tile through LDS so that 8 lanes cover 128 contiguous bytes of the
//     [B][T][Jp] state; its registers (W 128 + accumulators + x_t / noise operands) take one workgroup per CU.
// Epilogues are the ones of dsg_kernels.h (gemm_prefetch_tile / gemm_epilogue_tile), called per accumulator quad with the
// (token, feature) the 32 x 32 C layout gives the lane -- same arithmetic per element; the set differs from the others only in the
// order of the k sums.
// Measured on MI355X (profiles/r03_c_*, r03_q_*_kernel_stats.csv; per launch, HIP-launch path under rocprofv3; step: AQL path):
//     rows (clips)      LN + QKV  blk -> k_ln_frag + k_ws   linear1  blk -> k_ws   linear2  blk_k -> k_ws2   head  lean -> k_ws     step, BLOCK -> STREAM
//     5696 (1 x 64)         23.0 -> 5.5 + 12.5 us               12.8 -> 10.1 us        16.3 -> 10.5 us        52.5 -> 51.6 us      611 -> 505 us (+21 % clips/s)
//     2848 (1 x 32)                                                                                                               372 -> 357 us
//     4 x 1424 (4 x 16)                                                                                                           500 -> 436 us (10.2k -> 11.7k frames/s)
//     4 x 2848 (4 x 32)                                                                                                           933 -> 768 us (11.0k -> 13.3k frames/s)
//     4 x 712 (4 x 8)                                                                                                             292 -> 308 us   (slower)
//     1424 (1 x 16)                                                                                                               240 -> 316 us   (slower: see below)
// i.e. it paid from ~2800 rows in one lane and ~1400 rows per lane with several lanes.  Below that a workgroup gets ONE block, so the
// register-resident panel is loaded for nothing and the persistent loop has nothing to overlap.
// ROUND 4: the step no longer launches k_ln_frag, k_ws<EPI_GELU> and k_ws2<EPI_RESID> per layer -- k_ffn (dsg_fused.h) does linear1 + GELU
// + linear2 + residual + LayerNorm2 in one kernel and writes the next QKV's / the pose head's operand (1 x 64: 422 -> 375 us, 256 clips
// 16.4k -> 20.2k frames/s; auto_kernel_set() picks the set from 2000 rows, 1000 per lane with several lanes).  Those three kernels remain
// for the last layer under fused classifier-free guidance and as the measured alternative (profiles/r04_l..o_*).
// An earlier form with LayerNorm-on-read INSIDE the weight-stationary GEMM (64 fp32 rows staged through registers, one workgroup
// per CU) was slower than the block kernels at every size (QKV 28.0 vs 23.0 us at 5696 rows): csrc/experiments/dsg_stream_ln.h.
// Reference arithmetic being replaced: the nn.Linear / LayerNorm calls of torch's TransformerEncoderLayer (main/model/mdm.py:79-86)
// and the pose head (mdm.py:233-236) at M = 89 B.
#pragma once
#include "dsg_batched.h"
#include <type_traits>

namespace dsg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 bytes per lane, global -> LDS without a VGPR round trip: LDS address = lds_base (wave-uniform) + lane * 16
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_base_uniform, int lane) {
#ifndef DSG_EMU
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
#else
    *(f32x4*)((char*)lds_base_uniform + lane * 16) = *(const f32x4*)gsrc_lane;
#endif
}
// D[i][j] += sum_k A[i][k] B[k][j], 32 x 32 x 16: lane l gives row / column (l & 31) and the 8 k-values of group (l >> 5) of
// either operand (same k map on both sides, so the hardware's k order is irrelevant); D: column j = l & 31,
// row i = (r & 3) + 8 (r >> 2) + 4 (l >> 5) for register r.
__device__ __forceinline__ f32x16 mma32(f32x4 a, f32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// wave-level wait for this wave's outstanding global -> LDS loads (they count as vector memory operations)
__device__ __forceinline__ void glds_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }       // vmcnt(0) expcnt(7) lgkmcnt(15)

// workgroup id -> (panel, group): the panels of one group share blockIdx.x & 7, i.e. (as workgroups are dealt) one XCD -- the
// activation block they all read is fetched into ONE L2 (cached mode); groups are spread over the XCDs
struct WsId { int panel, grp; bool work; };
__device__ __forceinline__ WsId ws_id(int n_panels, int G) {
    const int x = (int)blockIdx.x, t = x >> 3;
    WsId w;
    w.panel = t % n_panels;
    w.grp = (x & 7) + 8 * (t / n_panels);
    w.work = t < n_panels * (G >> 3);
    return w;
}
__host__ __device__ inline int ws_grid_x(int n_panels, int G) { return 8 * n_panels * (G >> 3); }

// ---------------------------------------------------------------------------------------------------------
// k_ln_frag: LayerNorm of the fp32 residual rows ONCE per row (the block kernels redo it in each of their 12 - 18 column groups):
// the normalised rows in fp32 (g.Xn: the attention kernel's residual; may be null) and in bf16, fragment-major (g.out) -- the A
// operand k_ws<EPI_QKV> / k_ws<EPI_OUT> stream global -> LDS.  Same arithmetic and rounding point as the LayerNorm-on-read
// prologue (ln_rows).  16 rows per workgroup.
// ---------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void k_ln_frag(const GemmArgs g) {
    DSG_TL_SCOPE();
    typedef PBF16 P;
    constexpr int D = 64 * NCH, PITCH = D * 2 + 16;
    __shared__ __attribute__((aligned(16))) char img[16 * PITCH];
    preload_kernargs(g);
    const int m0 = (int)blockIdx.x * 16, tid = threadIdx.x, row = tid >> 4, c = tid & 15;
    f32x4 v[8];
    ln_rows<P, NCH>(g, m0, tid, img, PITCH, v);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int col = c * 4 + 64 * i;
        if (g.Xn) *(f32x4*)(g.Xn + (size_t)(m0 + row) * D + col) = v[i];
        P::store4((P::elem*)g.out + qk_off<P>(m0 + row, col, D / P::KB), v[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_ws: a K = D <= 256 GEMM on fragment-major bf16 rows, 64 x 128 blocks, W panel stationary in registers.   KD16 = K / 16
//   EPI_GELU  linear1 (bias + GELU -> hidden, fragment-major)
//   EPI_QKV   the QKV projection (a 128-column panel is Q / K or V: the operand order -- D[feature][token] for Q / K, D[token][feature]
//             for V^T -- is a per-workgroup scalar branch OUTSIDE the MFMA loop; v_mfma ignores EXEC, a per-MFMA select compiles into a
//             branch around every MFMA)
//   EPI_OUT   the pose head + sampler update (8 extra workgroups: the first one does the step bookkeeping, see StepCtl)
//             Round 5 (ONE): a workgroup per (panel, row block) with ONE activation buffer -- 32 KB of LDS and 141 VGPRs, three workgroups per
//             CU instead of two.  Alone the kernel does not move (30.5 vs 30.6 us at 64 clips: neither the x_t round trip, nor the noise
//             chain's place, nor the 2-vs-1 block imbalance of the persistent groups is its bound -- experiments/README.md); with four
//             lanes in flight the extra residency is worth 2 % of the step (4 x 64 clips 25.7 k -> 26.3 k frames/s, profiles/r05_x_*)
// ---------------------------------------------------------------------------------------------------------
// bytes of LDS staging the epilogue of k_ws<EPI> needs per row block of BM rows
__host__ __device__ constexpr int ws_stage_bytes(int epi, int bm) {
    return epi == EPI_QKV ? 128 * (bm + 4) * 2 : (epi == EPI_OUT ? 4 * 32 * 36 * 4 : 0);
}
template <int EPI, int KD16, bool SW, bool ONE = false>       // SW: D[feature][token] (Q / K, linear1, pose head); !SW: D[token][feature] (V^T)
__device__ __forceinline__ void ws_body(const GemmArgs& g, const WsId id, char* lds) {     // ONE: one row block per workgroup (G >= MB), one buffer
    typedef PBF16 P;
    constexpr int K = 16 * KD16, KB = K / 32, BM = 64;
    constexpr int ABYTES = BM * K * 2;
    // epilogue staging (V^T transposition, pose-head tile transposition): the retired activation buffer when it is large enough
    // (K = 256), else its own region behind the two buffers (K = 128: 128 x 68 x 2 = 17408 B / 4 x 32 x 36 x 4 = 18432 B > 16384 --
    // round-3 advisor: at tiny dims with more than one block per workgroup the stage ran into the buffer being prefetched)
    constexpr int STAGE = ws_stage_bytes(EPI, BM);
    constexpr bool STAGE_IN_A = STAGE <= ABYTES;
    constexpr int NBUF = ONE ? 1 : 2;
    const int G = g.ws_G, MB = (g.M + BM - 1) / BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- this wave's 64 columns x K of W: 2 column tiles of 32, KD16 fragments each, resident for the life of the workgroup
    const f32x4* wbase = (const f32x4*)g.Wp;
    f32x4 wf[2][KD16];
    int nb[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) nb[ct] = id.panel * 128 + wn * 64 + ct * 32;
    // lane part of a fragment address in ONE 32-bit register; tile / k-step part wave-uniform (scalar base per load)
    const unsigned w_lane = (unsigned)((l31 >> 4) * KB * 64 + lhi * 16 + (lane & 15));
    auto load_w = [&](int opaque_zero) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const f32x4* wt = wbase + (size_t)__builtin_amdgcn_readfirstlane((nb[ct] >> 4) * KB * 64 + opaque_zero);
#pragma unroll
            for (int s = 0; s < KD16; ++s)
                wf[ct][s] = (wt + ((s >> 1) * 64 + 2 * (s & 1) * 16))[w_lane];
        }
    };
    // the pose head's epilogue (x_t, bias, Philox + Box-Muller, the sampler update) needs the registers the panel occupies: it
    // re-reads its panel per block (64 KB from the L2) and runs 2 workgroups per CU instead of 1
    constexpr bool W_PER_BLOCK = EPI == EPI_OUT;
    if constexpr (!W_PER_BLOCK) load_w(0);
    // activation block mb -> LDS buffer (fragment-major operand: one contiguous 4 KB-tiles x KB span)
    auto issue_a = [&](int mb, int buf) {
        const char* src = (const char*)g.A + (size_t)mb * (BM * K * 2);
        char* dst = lds + buf * ABYTES;
#pragma unroll
        for (int c = 0; c < (BM * K * 2) / 4096; ++c) {
            const int chunk = c * 4 + wave;                        // 1 KB per wave instruction
            glds16(src + chunk * 1024 + lane * 16, dst + chunk * 1024, lane);
        }
    };
    int cur = 0;
    issue_a(id.grp, 0);
    int step = 0;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f, k5 = 0.f;
    if constexpr (EPI == EPI_OUT) {
        if (g.out_mode != OUT_FORWARD) {
            step = ldw<P>(&g.ctl->stepB);
            k1 = ldwf<P>(&g.ctl->k1); k2 = ldwf<P>(&g.ctl->k2); k3 = ldwf<P>(&g.ctl->k3); k4 = ldwf<P>(&g.ctl->k4); k5 = ldwf<P>(&g.ctl->k5);
        }
    }
#pragma unroll 1
    for (int mb = id.grp; mb < MB; mb += G) {
        const int m0 = mb * BM;
        glds_wait();
        DSG_LDS_BARRIER();             // the block has landed for every wave; every wave is done with the other buffer
        if constexpr (ONE) DSG_TL_MARK(0);      // (pose head, one block per workgroup) the activation block is in LDS
        if constexpr (!ONE) { if (mb + G < MB) issue_a(mb + G, cur ^ 1); }
        if constexpr (W_PER_BLOCK) {
            int zero = 0;
#ifndef DSG_EMU
            asm volatile("" : "+s"(zero));         // keeps the (loop-invariant) panel loads inside the loop
#endif
            load_w(zero);
        }
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const char* abase = lds + cur * ABYTES + ((size_t)((2 * wm + (l31 >> 4)) * KB) * 64 + lhi * 16 + (lane & 15)) * 16;
#pragma unroll
        for (int s = 0; s < KD16; ++s) {
            const f32x4 a = *(const f32x4*)(abase + ((s >> 1) * 64 + 2 * (s & 1) * 16) * 16);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if constexpr (SW) acc[ct] = mma32(wf[ct][s], a, acc[ct]);      // 4 consecutive features per lane
                else acc[ct] = mma32(a, wf[ct][s], acc[ct]);                   // 4 consecutive tokens per lane
            }
        }
        const int mw = m0 + 32 * wm;
        if constexpr (EPI == EPI_QKV && !SW) {
            // ---- V^T: the block goes through the retired activation buffer and out in aligned token groups (vt_store_block)
            constexpr int SP = BM + 4;                             // 136-byte feature pitch: 8-byte aligned quads
            typedef typename P::elem elem;
            elem* stage = (elem*)(STAGE_IN_A ? lds + cur * ABYTES : lds + NBUF * ABYTES);
            static_assert(128 * SP * 2 <= (STAGE_IN_A ? ABYTES : STAGE), "V^T stage");
            DSG_LDS_BARRIER();                                     // every wave is done reading the activation block
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int f = wn * 64 + ct * 32 + l31;             // feature within the panel
                const float bias = g.bias[id.panel * 128 + f];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[ct][4 * q] + bias, acc[ct][4 * q + 1] + bias, acc[ct][4 * q + 2] + bias, acc[ct][4 * q + 3] + bias};
                    P::store4(stage + f * SP + 32 * wm + 8 * q + 4 * lhi, v);
                }
            }
            DSG_LDS_BARRIER();
            vt_store_block<P, 128>(g, stage, SP, m0, BM, id.panel * 128 - 2 * (g.H * g.hd), tid);
            cur ^= 1;
            continue;
        }
        if constexpr (EPI == EPI_OUT) {
            // ---- pose head: the C layout gives a lane 4 features of ONE token per quad, i.e. a wave store touches 32 tokens x 32 bytes
            //      of the [B][T][Jp] state -- and the x_t loads likewise.  Each wave transposes its 32 x 32 tile through its own slice of
            //      the retired activation buffer (LDS operations of a wave execute in order: no barrier between its write and read) so
            //      that 8 lanes cover 128 contiguous bytes of one token; the epilogue arithmetic is position-based (gemm_epilogue_tile).
            constexpr int TP = 36;                                 // floats per token: 144-byte pitch, conflict-free both ways
            float* st = (float*)(STAGE_IN_A ? lds + cur * ABYTES : lds + NBUF * ABYTES) + wave * (32 * TP);
            static_assert(4 * 32 * TP * 4 <= (STAGE_IN_A ? ABYTES : STAGE), "pose-head stage");
            if constexpr (ONE) DSG_TL_MARK(1);                     // MFMA loop issued (the W panel has landed)
            DSG_LDS_BARRIER();                                     // every wave is done reading the activation block
            if constexpr (ONE) DSG_TL_MARK(2);
            const int tok = lane >> 3, quad = lane & 7;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4*)(st + l31 * TP + 8 * q + 4 * lhi) = (f32x4){acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
                DSG_WAVE_LDS_SYNC();
                // x_t of the lane's 4 (token, feature quad) items up front: ONE memory round trip per column tile instead of four
                // dependent load -> update -> store sequences (round 4); the Philox draw and the update stay one item at a time
                // (the register budget of 2 workgroups per CU)
                f32x4 xt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xt[i] = out_xt_load<P>(g, mw + tok + 8 * i, nb[ct] + 4 * quad);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    TileOps ops;
                    const int t = tok + 8 * i;
                    const f32x4 v = *(const f32x4*)(st + t * TP + 4 * quad);
                    gemm_prefetch_tile<P, EPI>(g, mw + t, nb[ct] + 4 * quad, 0, 0, step, ops, &xt[i]);
                    gemm_epilogue_tile<P, EPI>(g, mw + t, nb[ct] + 4 * quad, 0, 0, 0, true, v, ops, k1, k2, k3, k4, k5);
                }
                if constexpr (ONE) { if (ct == 0) DSG_TL_MARK(3); else DSG_TL_MARK(4); }      // column tile ct: 4 x (Philox draw + update + stores)
                DSG_WAVE_LDS_SYNC();
            }
            cur ^= 1;
            continue;
        }
        // ---- epilogue: per column tile, the 4 accumulator quads of the lane, QG quads' operands in flight at a time
        constexpr int QG = EPI == EPI_GELU ? 4 : (EPI == EPI_QKV ? 2 : 1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += QG) {
                TileOps ops[QG];
#pragma unroll
                for (int j = 0; j < QG; ++j) {
                    const int q = q0 + j;
                    if constexpr (SW) gemm_prefetch_tile<P, EPI>(g, mw, nb[ct] + 8 * q + 4 * lhi, l31, 0, step, ops[j]);      // token mw + l31, 4 features
                    else gemm_prefetch_tile<P, EPI>(g, mw + 8 * q + 4 * lhi, nb[ct], l31, 0, step, ops[j]);                  // feature nb + l31, 4 tokens
                }
#pragma unroll
                for (int j = 0; j < QG; ++j) {
                    const int q = q0 + j;
                    const f32x4 v = {acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
                    if constexpr (SW) gemm_epilogue_tile<P, EPI>(g, mw, nb[ct] + 8 * q + 4 * lhi, l31, 0, 0, true, v, ops[j], k1, k2, k3, k4, k5);
                    else gemm_epilogue_tile<P, EPI>(g, mw + 8 * q + 4 * lhi, nb[ct], l31, 0, 0, false, v, ops[j], k1, k2, k3, k4, k5);
                }
            }
        cur ^= 1;
    }
}

template <int EPI, int KD16, bool ONE = false>      // ONE (pose head): a workgroup per (panel, row block), one activation buffer, three workgroups per CU
__global__ __launch_bounds__(256, KD16 > 16 ? 1 : (ONE ? 3 : 2)) void k_ws(const GemmArgs g) {      // (round 6: K = 384 / 512 -- the DSG+ widths: a 192 / 256-register panel, one workgroup per CU)
    DSG_TL_SCOPE();
    typedef PBF16 P;
    constexpr int K = 16 * KD16, BM = 64;
    static_assert(KD16 % 4 == 0 && KD16 <= 32, "K = 64 ... 512");
    static_assert(EPI == EPI_GELU || EPI == EPI_QKV || EPI == EPI_OUT, "GEMMs of the step with K = D");
    constexpr int ABYTES = BM * K * 2, STAGE = ws_stage_bytes(EPI, BM);
    static_assert(!ONE || EPI == EPI_OUT, "ONE belongs to the pose head");
    __shared__ __attribute__((aligned(16))) char lds[(ONE ? 1 : 2) * ABYTES + (STAGE <= ABYTES ? 0 : STAGE)];
    preload_kernargs(g);
    const int n_panels = g.NT >> 3;
    const WsId id = ws_id(n_panels, g.ws_G);
    if constexpr (EPI == EPI_OUT) {
        if (!id.work) {
            if (g.ctl && (int)blockIdx.x == ws_grid_x(n_panels, g.ws_G) && threadIdx.x == 0 && g.out_mode != OUT_FORWARD) step_advance_A<P>(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    if (!id.work || id.grp >= (g.M + BM - 1) / BM) return;
    if constexpr (EPI == EPI_QKV) {
        // a whole 128-column panel is Q / K or V: two straight-line bodies under ONE scalar branch (v_mfma ignores EXEC, so a per-MFMA
        // select becomes a branch around every MFMA; and one body with both operand orders keeps both epilogues' registers live: spills)
        if (id.panel * 128 >= 2 * (g.H * g.hd)) ws_body<EPI, KD16, false>(g, id, lds);
        else ws_body<EPI, KD16, true>(g, id, lds);
    } else {
        ws_body<EPI, KD16, true, ONE>(g, id, lds);
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_ws2: a large-K GEMM, 32 x 64 blocks, the 4 waves split K, W slice stationary in registers.   KW16 = K / 64
//   EPI_RESID    linear2 (K = ff): + bias + residual rows
//   EPI_PARTIAL  the pose embedding (K = Jp, the padded pose dimension) on the fragment-major state shadow: the plain product into
//                `partial[0]` (k_loc adds the conditioning); 8 extra workgroups, the first one does the step bookkeeping (StepCtl)
// ---------------------------------------------------------------------------------------------------------
//   KSPLIT (round 6, EPI_PARTIAL at the DSG+ pose widths, Jp = 2176 / 2304): K is split over KSPLIT workgroups as well -- "panel" then counts (column panel,
//                K part) pairs; a workgroup streams its K part of every row tile (two 32-row buffers of 68 / 72 KB in the LDS) and leaves partial[part], which k_loc sums
//                in a fixed order like the slabs of the tile kernels.  KW16 = K / (64 KSPLIT)
template <int EPI, int KW16, int KSPLIT = 1>
__global__ __launch_bounds__(256, 1) void k_ws2(const GemmArgs g) {
    DSG_TL_SCOPE();
    typedef PBF16 P;
    static_assert(EPI == EPI_RESID || EPI == EPI_PARTIAL, "linear2 or the pose embedding");
    static_assert(KSPLIT == 1 || EPI == EPI_PARTIAL, "split-K partials belong to the pose embedding");
    constexpr int K = 64 * KW16, KB = K / 32, BM = 32;      // K, KB: this workgroup's part
    constexpr int KBTOT = KB * KSPLIT;
    constexpr int ABYTES = BM * K * 2;
    constexpr int REDBYTES = 4 * 2 * 16 * 64 * 4;                 // 4 waves x 2 column tiles x 16 registers x 64 lanes, fp32
    constexpr bool RED_IN_A = ABYTES >= REDBYTES;                  // the partial sums go through the retired activation buffer
    __shared__ __attribute__((aligned(16))) char lds[2 * ABYTES + (RED_IN_A ? 0 : REDBYTES)];
    preload_kernargs(g);
    const int n_panels = (g.NT >> 2) * KSPLIT, G = g.ws_G;
    WsId id = ws_id(n_panels, G);
    const int kpart = KSPLIT > 1 ? id.panel % KSPLIT : 0;      // (the parts of a column panel are neighbours: same XCD, same group of row blocks)
    if constexpr (KSPLIT > 1) id.panel /= KSPLIT;
    const int MB = (g.M + BM - 1) / BM;
    if constexpr (EPI == EPI_PARTIAL) {
        if (!id.work) {
            if (g.ctl && (int)blockIdx.x == ws_grid_x(n_panels, G) && threadIdx.x == 0) step_advance_B<P>(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    if (!id.work || id.grp >= MB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int l31 = lane & 31, lhi = lane >> 5;
    const f32x4* wbase = (const f32x4*)g.Wp;
    // ---- W[64 columns x this wave's K quarter]: KW16 fragments per 32-column tile
    f32x4 wf[2][KW16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int nt = id.panel * 4 + ct * 2 + (l31 >> 4);
#pragma unroll
        for (int s = 0; s < KW16; ++s) {
            const int ks = kpart * 4 * KW16 + wave * KW16 + s;     // k16 step of the whole K range
            wf[ct][s] = wbase[((size_t)nt * KBTOT + (ks >> 1)) * 64 + (2 * (ks & 1) + lhi) * 16 + (lane & 15)];
        }
    }
    // ---- the 2 output quads this wave finishes: column tile wave >> 1, registers 8 (wave & 1) .. + 7
    const int ct_f = wave >> 1, q0 = 2 * (wave & 1);
    f32x4 pbias[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        pbias[j] = EPI == EPI_RESID ? *(const f32x4*)(g.bias + id.panel * 64 + ct_f * 32 + 8 * (q0 + j) + 4 * lhi) : (f32x4){0.f, 0.f, 0.f, 0.f};
    auto issue_a = [&](int mb, int buf) {
        const char* src = (const char*)g.A + (size_t)mb * (ABYTES * KSPLIT);      // fragment-major: 2 row tiles x KBTOT k-blocks, contiguous
        char* dst = lds + buf * ABYTES;
#pragma unroll
        for (int c = 0; c < (ABYTES + 4095) / 4096; ++c) {
            const int chunk = c * 4 + wave;                        // one k-block (1 KB) of one row tile
            if constexpr (KSPLIT == 1) {
                if (chunk * 1024 < ABYTES) glds16(src + chunk * 1024 + lane * 16, dst + chunk * 1024, lane);
            } else {                                               // this part's KB k-blocks of either row tile
                const int rt = chunk / KB, kb = chunk - rt * KB;
                if (chunk < 2 * KB) glds16(src + (size_t)((rt * KBTOT + kpart * KB + kb) * 1024 + lane * 16), dst + chunk * 1024, lane);
            }
        }
    };
    int cur = 0;
    issue_a(id.grp, 0);
#pragma unroll 1
    for (int mb = id.grp; mb < MB; mb += G) {
        const int m0 = mb * BM;
        glds_wait();
        DSG_LDS_BARRIER();
        if (mb + G < MB) issue_a(mb + G, cur ^ 1);
        // residual rows of this wave's quads: in flight during the MFMA loop
        f32x4 pres[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            pres[j] = EPI == EPI_RESID ? lda16<P>(g.R, ((size_t)(m0 + l31) * g.ldo + id.panel * 64 + ct_f * 32 + 8 * (q0 + j) + 4 * lhi) * sizeof(float))
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const char* abase = lds + cur * ABYTES + ((size_t)((l31 >> 4) * KB) * 64 + lhi * 16 + (lane & 15)) * 16;
#pragma unroll
        for (int s = 0; s < KW16; ++s) {
            const int ks = wave * KW16 + s;
            const f32x4 a = *(const f32x4*)(abase + ((ks >> 1) * 64 + 2 * (ks & 1) * 16) * 16);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = mma32(wf[ct][s], a, acc[ct]);      // D[feature][token]
        }
        // ---- reduce the 4 K quarters in a fixed order through LDS
        float* red = (float*)(RED_IN_A ? lds + cur * ABYTES : lds + 2 * ABYTES);
        if constexpr (RED_IN_A) DSG_LDS_BARRIER();                 // every wave is done reading the activation block
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * 2 + ct) * 16 + r) * 64 + lane] = acc[ct][r];
        DSG_LDS_BARRIER();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * (q0 + j) + e;
                float sum = red[((0 * 2 + ct_f) * 16 + r) * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) sum += red[((w2 * 2 + ct_f) * 16 + r) * 64 + lane];
                v[e] = sum;
            }
            const int m = m0 + l31, n = id.panel * 64 + ct_f * 32 + 8 * (q0 + j) + 4 * lhi;
            if (m < g.M) *(f32x4*)((float*)g.out + ((size_t)kpart * g.MT * 16 + m) * g.ldo + n) = v + pbias[j] + pres[j];      // (EPI_PARTIAL: slab `kpart`)
        }
        if constexpr (!RED_IN_A) DSG_LDS_BARRIER();                // the separate partial-sum area is rewritten by the next block
        cur ^= 1;
    }
}


// ---------------------------------------------------------------------------------------------------------
// k_clip_attn_w (round 6, ROWS at the DSG+ widths): k_clip_attn (dsg_fused.h) at latent_dim 384 / 512 -- the attention half of an encoder layer per
// (clip, head): the workgroup projects its head's Q / K / V from the clip's rows and runs the self-attention of all NKT query tiles on them; the QKV GEMM
// and k_attn disappear as dispatches, Q / K / V^T never reach memory.  What differs from the narrow kernel, because nothing fits as laid out there:
//   * the clip's rows (10 row tiles x 12 / 16 KB) pass through the LDS in chunks of XR row tiles, double-buffered with global -> LDS loads (no registers):
//     the next chunk is in flight while a chunk is projected;
//   * 8 waves, column tile t of the head's [Q | K | V] order by wave t % 8, ONE pass over the rows (ONEP): at latent_dim 384 18 tiles = three on waves 0 - 1, two
//     on the others (144 weight registers: 236 VGPRs); at 512 24 tiles = three on every wave (192 weight registers: the bias waits in the LDS and the A fragments
//     have no look-ahead -- 254 VGPRs).  Every wave reads ALL of the rows from the LDS once -- 8 x 120 / 160 KB at 128 B / clk is what bounds the pass.  The
//     TWO-pass form (!ONEP; kept as the A/B reference, -DDSG_X_TWH_TWOPASS): two tiles per wave (128 registers), pass A the 16 Q / K tiles, pass B the 8 V tiles
//     in pairs on waves 0 - 3 with the chunk's two row tiles side by side (four accumulation chains: one wave per SIMD) -- 1.5 x the LDS reads and a weight
//     reload in between: 475.1 vs 472.9 us per TWH step at 16 clips, 4 x 8 clips 601 vs 585;
//   * EVERY tile is computed as W . X^T (a lane holds 4 consecutive dims of one token) -- one operand order -- and a V tile is transposed when it moves
//     into V^T (four 2-byte LDS stores per row tile instead of one 8-byte store); the V tiles wait in registers, rounded, until every wave is done with the
//     rows: V^T takes their place;
//   * query tiles qt = wave, wave + 8 (10 query tiles on 8 waves); the K / V^T fragment reads of the attention are fenced per two key / dim tiles (hoisted
//     above their MFMAs they were 120 - 160 registers: 228 / 552 bytes of scratch in the first build).
// Bit-identical to the QKV GEMM + k_attn path.  Measured (profiles/r06_dn_*, r06_dq_*): the kernel is 17 / 24 us at 16 clips -- what the QKV GEMM + k_attn + their
// boundary took -- on 64 CUs instead of 192 + 256: BEAT 1 x 16 clips 374.7 -> 360.0 us per step, 4 x 16: 827 -> 604; TWH 1 x 16: 492 -> 475, 4 x 16: 1194 -> 871.
// Same rounding points as k_attn on the QKV GEMM's output (Q / K / V and P in bf16, softmax in fp32, 1 / sum applied to the fp32 P V).
// Reference arithmetic: nn.MultiheadAttention of torch's TransformerEncoderLayer (BEAT-TWH-main/model/mdm.py:134-146), in_proj + softmax(QK^T/sqrt(hd))V.
// ---------------------------------------------------------------------------------------------------------
template <int NJ, int CW, int KD, bool PF = true>      // NJ column tiles against one row tile: the A fragments four k-blocks at a time, the next four in flight (PF)
__device__ __forceinline__ void clip_w_proj(const f32x4 (&wf)[CW][KD], const f32x4 (*xrow)[64], int lane, f32x4 (&acc)[CW]) {
    static_assert(KD % 4 == 0, "k-blocks in groups of 4");
    f32x4 a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = xrow[i][lane];
#pragma unroll
    for (int kg = 0; kg < KD / 4; ++kg) {
        if (PF && kg + 1 < KD / 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[(kg + 1) & 1][i] = xrow[4 * (kg + 1) + i][lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = PBF16::mma(wf[j][4 * kg + i], a[PF ? (kg & 1) : 0][i], acc[j]);      // D[dim = 4 lg + r][token = lr]
        DSG_LOADS_ISSUED();
        if (!PF && kg + 1 < KD / 4) {      // (no room for the look-ahead: the other wave of the SIMD covers the LDS latency)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[0][i] = xrow[4 * (kg + 1) + i][lane];
        }
    }
}
// ... two column tiles against TWO row tiles side by side: four independent accumulation chains (pass B of the two-pass form: one wave per SIMD)
template <int KD>
__device__ __forceinline__ void clip_w_proj2(const f32x4 (&wf)[2][KD], const f32x4 (*xrow0)[64], const f32x4 (*xrow1)[64], int lane, f32x4 (&acc0)[2], f32x4 (&acc1)[2]) {
    f32x4 a[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { a[0][0][i] = xrow0[i][lane]; a[0][1][i] = xrow1[i][lane]; }
#pragma unroll
    for (int kg = 0; kg < KD / 2; ++kg) {
        if (kg + 1 < KD / 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[(kg + 1) & 1][0][i] = xrow0[2 * (kg + 1) + i][lane]; a[(kg + 1) & 1][1][i] = xrow1[2 * (kg + 1) + i][lane]; }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc0[j] = PBF16::mma(wf[j][2 * kg + i], a[kg & 1][0][i], acc0[j]);
                acc1[j] = PBF16::mma(wf[j][2 * kg + i], a[kg & 1][1][i], acc1[j]);
            }
        DSG_LOADS_ISSUED();
    }
}

template <int DT, int NKT, int XR, bool ONEP>      // D = 64 DT, H = 4, hd = 16 DT, Tp = 16 NKT; XR row tiles per LDS chunk; ONEP: one pass over the rows (three column tiles per wave fit)
__global__ __launch_bounds__(512, 1) void k_clip_attn_w(const ClipAttnArgs g) {
    DSG_TL_SCOPE();
    typedef PBF16 P;
    typedef P::elem elem;
    constexpr int NW = 8;
    constexpr int D = DT * 64, HD = DT * 16;
    constexpr int KD = D / P::KB, KDH = HD / P::KB, ND = HD / 16, NVF = NKT / 2;
    constexpr int CT = 3 * ND, NP = ND / 2;          // column tiles of the head; V tile pairs (pass B)
    constexpr int CW = ONEP ? 3 : 2;
    constexpr int NCH = (NKT + XR - 1) / XR, CF = XR * KD, CFW = (CF + NW - 1) / NW;
    static_assert(NKT % 2 == 0 && HD % P::KB == 0 && ND % 2 == 0 && 2 * ND <= 2 * NW && NP <= NW && CT <= 3 * NW && CT > 2 * NW, "shape");
    static_assert(ONEP || XR == 2, "two passes: pass B takes the row tiles in pairs");
    __shared__ __attribute__((aligned(16))) f32x4 xs[2][CF][64];
    __shared__ __attribute__((aligned(16))) f32x4 qs[NKT * KDH][64];
    __shared__ __attribute__((aligned(16))) f32x4 ks[NKT * KDH][64];
    f32x4 (* const vs)[64] = &xs[0][0];
    static_assert(ND * NVF <= 2 * CF, "V^T fits in the retired rows");
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    preload_kernargs(g);
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    // ---- step i of the row stream (chunk i % NCH) -> LDS buffer i & 1: fragment f = (row tile, k-block) by wave f % 8; a lane fetches the 16 bytes
    //      of ITS row (the clip's rows start anywhere in the flattened row tiles); rows past the clip: its last row (masked / dropped later)
    auto issue_x = [&](int i) {
        const int c = i % NCH;
#pragma unroll
        for (int k = 0; k < CFW; ++k) {
            const int f = wave + NW * k, rl = f / KD, kb = f - rl * KD, rt = c * XR + rl;
            if (f < CF && rt < NKT) {
                const int m = b * g.ntok + min(rt * 16 + lr, g.ntok - 1);
                glds16((const char*)g.X + (size_t)qk_off<P>(m, kb * P::KB + P::E * lg, KD) * sizeof(elem), &xs[i & 1][f][0], lane);
            }
        }
    };
    issue_x(0);
    const f32x4* wq = (const f32x4*)g.Wqkv + lane;
    f32x4 wf[CW][KD];
    constexpr bool LB = ONEP && KD > 12;             // (one pass at latent_dim 512: 192 weight registers -- the bias waits in the LDS, no look-ahead for the A fragments)
    f32x4 pb[LB ? 1 : CW];
    __shared__ __attribute__((aligned(16))) float bs[LB ? CT * 16 : 4];
    if constexpr (LB) {
        const int c4 = threadIdx.x;                  // four bias values of the head's [Q | K | V] columns
        if (c4 < CT * 4) { const int which = (4 * c4) / HD, within = 4 * c4 - which * HD; *(f32x4*)&bs[4 * c4] = *(const f32x4*)(g.bqkv + which * D + h * HD + within); }
    }
    bf16x4v vkeep[NKT][ONEP ? 1 : 2];                // the wave's V tile(s), 4 dims of one token per row tile
    // the wave's column tiles in the head's [Q | K | V] order (ND tiles each): packed in_proj column tile, weights, bias
    auto load_tiles = [&](const int (&t)[CW]) {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int which = t[j] / ND, d0 = (t[j] - which * ND) * 16;
            const int nt = which * (D / 16) + h * ND + d0 / 16;
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) wf[j][kb] = P::wload(wq, (size_t)nt * KD + kb);
            if constexpr (!LB) pb[j] = *(const f32x4*)(g.bqkv + nt * 16 + 4 * lg);
        }
        DSG_LOADS_ISSUED();
    };
    int dv[2] = {0, 0};                              // first dim of the parked V tile(s) inside the head
    bool vb = false;                                 // this wave holds V tiles
    if constexpr (ONEP) {
        // ---- one pass: tiles wave, wave + 8, wave + 16 (18 tiles: waves 0 - 1 three, the others two; at most one of them a V tile)
        int t[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] = min(wave + NW * j, CT - 1);
        const bool three = wave + 2 * NW < CT;
        load_tiles(t);
        DSG_TL_MARK(0);      // first chunk of rows + the wave's columns requested
        const int jv = t[1] >= 2 * ND ? 1 : 2;       // slot of the V tile, if any
        vb = t[1] >= 2 * ND || three;
        dv[0] = (t[jv] - 2 * ND) * 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            glds_wait();
            DSG_LDS_BARRIER();            // chunk c has landed for every wave; every wave is done with the other buffer
            if (c + 1 < NCH) issue_x(c + 1);
#pragma unroll
            for (int rl = 0; rl < XR; ++rl) {
                const int rt = c * XR + rl;
                if (rt < NKT) {
                    f32x4 acc[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                    if (three) clip_w_proj<3, 3, KD, !LB>(wf, &xs[c & 1][rl * KD], lane, acc);
                    else clip_w_proj<2, 3, KD, !LB>(wf, &xs[c & 1][rl * KD], lane, acc);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (j < 2 || three) {
                            const int which = t[j] / ND, d0 = (t[j] - which * ND) * 16;
                            f32x4 pbj;
                            if constexpr (LB) pbj = *(const f32x4*)&bs[t[j] * 16 + 4 * lg]; else pbj = pb[j];
                            const f32x4 y = acc[j] + pbj;
                            if (which < 2) P::store4((elem*)(which == 0 ? &qs[0][0] : &ks[0][0]) + qk_off<P>(rt * 16 + lr, d0 + 4 * lg, KDH), y);
                            else vkeep[rt][0] = __builtin_convertvector(y, bf16x4v);      // (P::store4's rounding)
                        }
                    }
                }
            }
        }
        DSG_TL_MARK(1);      // projection done
    } else {
        // ---- pass A: Q / K tiles wave, wave + 8
        const int ta[2] = {wave, min(wave + NW, 2 * ND - 1)};
        const int nja = wave + NW < 2 * ND ? 2 : 1;
        load_tiles(ta);
        DSG_TL_MARK(0);      // first chunk of rows + the Q / K columns requested
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            glds_wait();
            DSG_LDS_BARRIER();            // chunk c has landed for every wave; every wave is done with the other buffer
            issue_x(c + 1);               // (the last one: chunk 0 again, for pass B)
#pragma unroll
            for (int rl = 0; rl < XR; ++rl) {
                const int rt = c * XR + rl;
                if (rt < NKT) {
                    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                    if (nja == 2) clip_w_proj<2, 2, KD>(wf, &xs[c & 1][rl * KD], lane, acc);
                    else clip_w_proj<1, 2, KD>(wf, &xs[c & 1][rl * KD], lane, acc);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j < nja) {
                            const int which = ta[j] / ND, d0 = (ta[j] - which * ND) * 16;
                            P::store4((elem*)(which == 0 ? &qs[0][0] : &ks[0][0]) + qk_off<P>(rt * 16 + lr, d0 + 4 * lg, KDH), acc[j] + pb[j]);
                        }
                    }
                }
            }
        }
        DSG_TL_MARK(1);      // pass A done: Q / K of the head in LDS
        // ---- pass B: V tiles in pairs (waves 0 .. NP - 1), the chunk's two row tiles side by side (four accumulation chains: one wave per SIMD)
        vb = wave < NP;
        const int tb[2] = {2 * ND + min(wave, NP - 1), 2 * ND + min(wave, NP - 1) + NP};
        dv[0] = (tb[0] - 2 * ND) * 16; dv[1] = (tb[1] - 2 * ND) * 16;
        if (vb) load_tiles(tb);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = NCH + c;
            glds_wait();
            DSG_LDS_BARRIER();
            if (c + 1 < NCH) issue_x(i + 1);
            if (vb) {
                f32x4 acc0[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, acc1[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                static_assert(NKT % 2 == 0, "pairs");
                clip_w_proj2<KD>(wf, &xs[i & 1][0], &xs[i & 1][KD], lane, acc0, acc1);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    vkeep[2 * c][j] = __builtin_convertvector(acc0[j] + pb[j], bf16x4v);      // (P::store4's rounding)
                    vkeep[2 * c + 1][j] = __builtin_convertvector(acc1[j] + pb[j], bf16x4v);
                }
            }
        }
    }
    DSG_LDS_BARRIER();                                            // every wave is done with the rows: V^T moves in
    if (vb) {
#pragma unroll
        for (int j = 0; j < (ONEP ? 1 : 2); ++j) {
#pragma unroll
            for (int rt = 0; rt < NKT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) *((__bf16*)&vs[0][0] + vt_off<P>(dv[j] + 4 * lg + r, rt * 16 + lr, NVF)) = vkeep[rt][j][r];
        }
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(2);      // Q / K / V^T of the head in LDS
    // ---- attention of query tiles wave, wave + 8 (k_attn on LDS operands)
    const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll 1
    for (int qt = wave; qt < NKT; qt += NW) {
        f32x4 s[NKT], qf[KDH];
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) qf[kb] = qs[qt * KDH + kb][lane];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(ks[nt * KDH + kb][lane], qf[kb], s[nt]);   // D[key = 4 lg + r][query = lr]
            if (nt & 1) DSG_LOADS_ISSUED();      // (two key tiles' fragments in flight: all NKT x KDH of them hoisted were 120 - 160 registers -- scratch)
        }
        float mx = -DSG_FLT_MAX;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = nt * 16 + 4 * lg + r;
                const float v = key < g.ntok ? s[nt][r] * scale : -DSG_FLT_MAX;
                s[nt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = nt * 16 + 4 * lg + r;
                const float pv = key < g.ntok ? P::exp_sm(s[nt][r] - mx) : 0.f;
                s[nt][r] = pv;
                sum += pv;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        DSG_LOADS_ISSUED();
        f32x4 pfr[NVF];
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
        const int q = qt * 16 + lr;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NVF; ++kb) o = P::mma(vs[dt * NVF + kb][lane], pfr[kb], o);     // D[dim = 4 lg + r][query = lr]
            if (q < g.ntok) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
                P::store4_afrag((elem*)g.out, (size_t)qk_off<P>(b * g.ntok + q, h * HD + dt * 16 + 4 * lg, KD), y);      // the rounding point of the attention rows
            }
            if (dt & 1) DSG_LOADS_ISSUED();      // (two dim tiles' V^T fragments in flight)
        }
    }
    DSG_TL_MARK(3);      // attention + stores issued
}

}  // namespace dsg
