// dsg_stream_ln.h -- REJECTED (round 3): the weight-stationary persistent GEMM of dsg_stream.h with LayerNorm-on-read, for the
// QKV projection of layers > 0 and the pose head.  64 fp32 rows of a block are normalised once per (block, 128-column panel) into
// a padded row-major LDS image (conflict-free for the 32-row MFMA fragments: 528-byte pitch), W panel in registers, 32x32x16 MFMA,
// the epilogues of dsg_kernels.h per accumulator quad; the Q / K-versus-V operand order is a per-PANEL scalar branch outside the
// MFMA loop (a per-MFMA select is compiled into a branch around every v_mfma, which ignores EXEC: 708 accvgpr moves, 17 us).
// Parity-green on MI355X (the whole step vs the oracle, 7.5e-3 rel-L2 after 40 steps), but SLOWER than the block / lean kernels at
// every size (profiles/r03_c_stream_*_kernel_stats.csv, r03_e_*):
//                        1424 rows          5696 rows
//     LN + QKV           17.4 vs 8.4 us     28.0 vs 23.0 us
//     LN + pose head     18.5 vs 17.0 us    53.4 vs 52.3 us
// Why: the fp32 rows go through VGPRs (64 registers of staging next to 128 of weights: one workgroup per CU) and the phases
// load -> LayerNorm -> MFMA -> epilogue of a block run strictly one after the other; the block kernels keep 3 workgroups per CU in
// flight instead.  What would fix it is LayerNorm delivered by the PRODUCER (bf16 rows + row statistics out of linear2), so that
// these GEMMs take the global -> LDS path of k_ws as well; see DESIGN.md s5 "Next".
#pragma once
#include "dsg_stream.h"
#include <type_traits>

namespace dsg {

template <int EPI, int KD16>
__global__ __launch_bounds__(256, 1) void k_ws_ln(const GemmArgs g) {
    typedef PBF16 P;
    constexpr int K = 16 * KD16, KB = K / 32, BM = 64;
    constexpr int ROWB = K * 2 + 16;                               // padded row pitch of the LayerNorm image
    static_assert(EPI == EPI_QKV || EPI == EPI_OUT, "LayerNorm GEMMs of the step");
    __shared__ __attribute__((aligned(16))) char lds[BM * ROWB];
    preload_kernargs(g);
    const int n_panels = g.NT >> 3, G = g.ws_G;
    const WsId id = ws_id(n_panels, G);
    const int MB = (g.M + BM - 1) / BM;
    if constexpr (EPI == EPI_OUT) {
        if (!id.work) {          // 8 extra workgroups: the first one does the step bookkeeping (see StepCtl)
            if (g.ctl && (int)blockIdx.x == ws_grid_x(n_panels, G) && threadIdx.x == 0 && g.out_mode != OUT_FORWARD) step_advance_A<P>(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    if (!id.work || id.grp >= MB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const f32x4* wbase = (const f32x4*)g.Wp;
    f32x4 wf[2][KD16];
    int nb[2];
    const bool swp = !(EPI == EPI_QKV && id.panel * 128 >= 2 * (g.H * g.hd));       // a whole panel is Q / K or V
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        nb[ct] = id.panel * 128 + wn * 64 + ct * 32;
        const int nt = (nb[ct] >> 4) + (l31 >> 4);
#pragma unroll
        for (int s = 0; s < KD16; ++s)
            wf[ct][s] = wbase[((size_t)nt * KB + (s >> 1)) * 64 + (2 * (s & 1) + lhi) * 16 + (lane & 15)];
    }
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
        ln_rows_blk<P, KD16 / 4, 4>(g, m0, tid, lds, ROWB, g.Xn != nullptr && id.panel == 0);
        DSG_LDS_BARRIER();
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const char* abase = lds + (32 * wm + l31) * ROWB + 16 * lhi;
        auto mfma_loop = [&](auto sw) {
#pragma unroll
            for (int s = 0; s < KD16; ++s) {
                const f32x4 a = *(const f32x4*)(abase + s * 32);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    if constexpr (decltype(sw)::value) acc[ct] = mma32(wf[ct][s], a, acc[ct]);      // D[feature][token]
                    else acc[ct] = mma32(a, wf[ct][s], acc[ct]);                                   // D[token][feature]
                }
            }
        };
        if constexpr (EPI == EPI_QKV) {
            if (swp) mfma_loop(std::true_type{}); else mfma_loop(std::false_type{});
        } else {
            mfma_loop(std::true_type{});
        }
        const int mw = m0 + 32 * wm;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            TileOps ops[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (swp) gemm_prefetch_tile<P, EPI>(g, mw, nb[ct] + 8 * q + 4 * lhi, l31, 0, step, ops[q]);      // token mw + l31, 4 features
                else gemm_prefetch_tile<P, EPI>(g, mw + 8 * q + 4 * lhi, nb[ct], l31, 0, step, ops[q]);          // feature nb + l31, 4 tokens
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
                if (swp) gemm_epilogue_tile<P, EPI>(g, mw, nb[ct] + 8 * q + 4 * lhi, l31, 0, 0, true, v, ops[q], k1, k2, k3, k4, k5);
                else gemm_epilogue_tile<P, EPI>(g, mw + 8 * q + 4 * lhi, nb[ct], l31, 0, 0, false, v, ops[q], k1, k2, k3, k4, k5);
            }
        }
        DSG_LDS_BARRIER();            // single image: everybody is done with it before the next LayerNorm
    }
}

}  // namespace dsg
