// dsg_batched.h -- block GEMMs for the batched step (kernel set DSG_KSET_BLOCK): 32-row x 64-column output blocks per
// workgroup instead of dsg_kernels.h's one 16 x 16 tile per wave.
//
// Why: the latency GEMM (gemm_body) re-fetches a weight tile for every 16-row tile and recomputes the LayerNorm-on-read of
// its rows once per 64 output columns.  At batch 1 (6 row tiles) that redundancy is free; at batch 16 (89 row tiles) it is
// the step: rocprofv3 (profiles/r02_a_b16_kernel_stats.csv) shows ~10 us for each LayerNorm GEMM and 7.8 us for linear2 --
// 55-75 TFLOP/s -- with the per-CU load path (every wave-load of a 1 KB fragment costs ~25-100 cycles of texture-address
// issue) and 2-3 waves per SIMD as the limits, MFMA busy 0.5-1.7 %.  Here a workgroup owns 64 rows:
//   k_gemm_blk    K = D (<= 512): the 64 A rows are staged in LDS ONCE (LayerNorm-on-read applied on the way in, or a plain
//                 copy of the operand rows), wave w owns TNW 16-column tiles and ALL 4 row tiles: per k-block one weight
//                 fragment load feeds 4 MFMAs (16 x less weight traffic per row, 4 x less LayerNorm recompute per column).
//   k_gemm_blk_k  large K (linear2: K = ff, pose embedding: K = J): the 4 waves split K; each wave multiplies its K range of
//                 the whole 64 x 64 block (4 A + 4 B fragment loads feed 16 MFMAs, nothing is loaded twice in the
//                 workgroup), the 4 partial blocks are reduced through LDS in a fixed order.
// Arithmetic per output element is the same fp32-accumulated MFMA chain over k as in gemm_body, only the k order of the
// partial sums differs for k_gemm_blk_k (4 contiguous K ranges instead of one), i.e. last-bit fp32 differences vs the
// latency kernels -- the parity tests cover both kernel sets against the same goldens.
#pragma once
#include "dsg_kernels.h"

namespace dsg {

// LayerNorm of RT x 16 rows at once (256 threads, 16 lanes per row): all row loads of the RT tiles are in flight before the
// first reduction.  Rows go to LDS (row-major, `pitch` bytes apart) in the MFMA element type; `xn` (fp32 write-back of the
// normalised rows for the residual path) may be null.
template <class P, int NCH, int RT>
__device__ __forceinline__ void ln_rows_blk(const GemmArgs& g, int m0, int tid, char* lds_a, int pitch, bool wr) {
    typedef typename P::elem elem;
    const int D = g.D, row = tid >> 4, c = tid & 15;
    f32x4 v[RT][NCH];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[rt][i] = *(const f32x4*)(g.X + (size_t)(m0 + rt * 16 + row) * D + c * 4 + 64 * i);
    f32x4 gg[NCH], bb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) { gg[i] = *(const f32x4*)(g.ln_g + c * 4 + 64 * i); bb[i] = *(const f32x4*)(g.ln_b + c * 4 + 64 * i); }
    DSG_LOADS_ISSUED();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) s += (v[rt][i][0] + v[rt][i][1]) + (v[rt][i][2] + v[rt][i][3]);
        s = row16_sum(s);
        const float mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[rt][i][e] - mean; q += d * d; }
        q = row16_sum(q);
        const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
        const int r = rt * 16 + row;
        const bool w = wr && (m0 + r) < g.M;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[rt][i][e] - mean) * rstd * gg[i][e] + bb[i][e];
            P::store4((elem*)(lds_a + r * pitch) + c * 4 + 64 * i, y);
            if (w) *(f32x4*)(g.Xn + (size_t)(m0 + r) * D + c * 4 + 64 * i) = y;
        }
    }
}

// DMAX: widest row (= K) the LDS block is sized for (256 or 512); TNW: 16-column tiles per wave (1 or 2); RT: 16-row tiles
// per workgroup (2 or 4).  Everything a workgroup needs from memory -- weight fragments, the A rows, the epilogue operands of
// all its tiles (bias, residual rows, x_t, noise) -- is requested before the first dependent instruction: one memory round
// trip per workgroup, like the latency kernels (a first version that fetched the epilogue operands tile by tile after the
// MFMA loop was SLOWER than the 16 x 16 tile kernels: 407 vs 321 us per batch-16 step, profiles/r02_b_*).
template <class P, int PRO, int EPI, int DMAX, int TNW, int RT>
__global__ __launch_bounds__(256) void k_gemm_blk(const GemmArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    // CH: the whole K range in one batch of weight-fragment loads where it fits (K = 384 / 512 ran two serial load phases)
    constexpr int ES = (int)sizeof(elem), BM = 16 * RT, KBMAX = DMAX / P::KB, CH = KBMAX > 16 ? 16 : (KBMAX < 8 ? 8 : KBMAX);
    static_assert(EPI != EPI_PARTIAL, "split-K partials come from k_gemm_blk_k");
    static_assert(RT == 2 || RT == 4, "32- or 64-row blocks");
    __shared__ __attribute__((aligned(16))) char lds_a[BM * (DMAX * ES + 16)];
    preload_kernargs(g);
    const int NG = g.NT / (4 * TNW);
    int ng = xcd_ngroup<P>(), mb = blockIdx.y;
    const int MB = (g.MT + RT - 1) / RT;
    if constexpr (EPI == EPI_OUT) {
        if (mb >= MB) {      // extra grid row: step bookkeeping (see gemm_body)
            if (g.ctl && blockIdx.x == 0 && threadIdx.x == 0 && g.out_mode != OUT_FORWARD) step_advance_A(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    // (round 6: the QKV GEMM of the DSG+ widths is 18 column groups -- XCDs 0-1 held three, the others two: BEAT 16 clips 17.8 -> see profiles/r06_de_*)
    if (!xcd_deal_groups(NG, MB, ng, mb)) return;
    if (ng >= NG || mb >= MB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int m0 = mb * BM;
    const int nt0 = (ng * 4 + wave) * TNW;
    const int KBtot = g.KBtot, kb_last = KBtot - 1;
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    bool swapped[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) swapped[t] = !(EPI == EPI_QKV && ((nt0 + t) * 16) >= 2 * (g.H * g.hd));
    // ---- weight fragments of the first chunk: in flight while the A rows are staged
    f32x4 bf[CH][TNW];
    auto load_b = [&](int kb0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kb = min(kb0 + c, kb_last);
#pragma unroll
            for (int t = 0; t < TNW; ++t) bf[c][t] = wbase[((size_t)(nt0 + t) * KBtot + kb) * 64];
        }
    };
    load_b(0);
    int step = 0;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f, k5 = 0.f;
    if constexpr (EPI == EPI_OUT) {
        if (g.out_mode != OUT_FORWARD) {
            step = ldw<P>(&g.ctl->stepB);
            k1 = ldwf<P>(&g.ctl->k1); k2 = ldwf<P>(&g.ctl->k2); k3 = ldwf<P>(&g.ctl->k3); k4 = ldwf<P>(&g.ctl->k4); k5 = ldwf<P>(&g.ctl->k5);
        }
    }
    // ---- epilogue operands of every tile of this wave (they do not depend on the main loop)
    TileOps ops[RT][TNW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TNW; ++t) gemm_prefetch_tile<P, EPI>(g, min(m0 + rt * 16, (g.MT - 1) * 16), (nt0 + t) * 16, lr, lg, step, ops[rt][t]);
    // ---- stage the A rows in LDS (row-major, padded pitch: conflict-free fragment reads)
    const int K = KBtot * P::KB;
    const int pitch = K * ES + 16;
    if constexpr (PRO == PRO_LN) {
        const bool wr = g.Xn != nullptr && ng == 0;
        const int nch = g.D >> 6;
        if constexpr (DMAX <= 256) {       // one straight-line copy per row width the instantiation can see
            if (nch == 4) ln_rows_blk<P, 4, RT>(g, m0, tid, lds_a, pitch, wr);
            else if (nch == 2) ln_rows_blk<P, 2, RT>(g, m0, tid, lds_a, pitch, wr);
            else if (nch == 3) ln_rows_blk<P, 3, RT>(g, m0, tid, lds_a, pitch, wr);
            else ln_rows_blk<P, 1, RT>(g, m0, tid, lds_a, pitch, wr);
        } else {
#pragma unroll 1
            for (int half = 0; half < RT / 2; ++half) {
                if (nch == 8) ln_rows_blk<P, 8, 2>(g, m0 + 32 * half, tid, lds_a + 32 * half * pitch, pitch, wr);
                else if (nch == 6) ln_rows_blk<P, 6, 2>(g, m0 + 32 * half, tid, lds_a + 32 * half * pitch, pitch, wr);
                else if (nch == 5) ln_rows_blk<P, 5, 2>(g, m0 + 32 * half, tid, lds_a + 32 * half * pitch, pitch, wr);
                else ln_rows_blk<P, 7, 2>(g, m0 + 32 * half, tid, lds_a + 32 * half * pitch, pitch, wr);
            }
        }
    } else {
        // plain copy, 16 bytes per thread per pass; the source is row-major [rows][lda] or fragment-major (qk_off order)
        const int cpr = K * ES / 16;                       // 16-byte chunks per row
        const int total = BM * cpr;
        for (int e0 = 0; e0 < total; e0 += 256 * 4) {
            f32x4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * 256 + tid, total - 1);
                const char* src;
                if (g.a_frag) {       // chunk e of the block in fragment order: [row tile][k-block][lane][16 B]
                    src = (const char*)g.A + ((size_t)(m0 >> 4) * KBtot * 64 + e) * 16;
                } else {
                    const int r = e / cpr, cc = e - r * cpr;
                    src = (const char*)g.A + ((size_t)(m0 + r) * g.lda) * ES + cc * 16;
                }
                tmp[u] = *(const f32x4*)src;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 + tid;
                if (e < total) {
                    int r, boff;
                    if (g.a_frag) {
                        const int ln = e & 63, kb = (e >> 6) % KBtot, rt = (e >> 6) / KBtot;
                        r = rt * 16 + (ln & 15); boff = (kb * P::KB + P::E * (ln >> 4)) * ES;
                    } else {
                        r = e / cpr; boff = (e - r * cpr) * 16;
                    }
                    *(f32x4*)(lds_a + r * pitch + boff) = tmp[u];
                }
            }
        }
    }
    DSG_LDS_BARRIER();
    // ---- main loop: one weight fragment (per column tile) feeds the RT row tiles
    f32x4 acc[RT][TNW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < TNW; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb0 = 0; kb0 < KBtot; kb0 += CH) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool live = kb0 + c < KBtot;              // wave-uniform
            const int kb = min(kb0 + c, kb_last);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 a = *(const f32x4*)(lds_a + (rt * 16 + lr) * pitch + (kb * P::KB + P::E * lg) * ES);
                a = live ? a : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < TNW; ++t) acc[rt][t] = swapped[t] ? P::mma(bf[c][t], a, acc[rt][t]) : P::mma(a, bf[c][t], acc[rt][t]);
            }
        }
        if (kb0 + CH < KBtot) load_b(kb0 + CH);
    }
    // ---- epilogue
    if constexpr (EPI == EPI_QKV) {
        if (!swapped[0]) {
            // a V column group (64 TNW columns never straddle 2 D): the block's V^T goes through the retired row buffer and out in
            // aligned token groups (vt_store_block) instead of 4 element stores per lane and row quad
            constexpr int SP = BM + 4;
            elem* stage = (elem*)lds_a;
            DSG_LDS_BARRIER();                                  // every wave is done reading the A rows
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < TNW; ++t) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[rt][t][e] + ops[rt][t].pbs;
                    P::store4(stage + ((wave * TNW + t) * 16 + lr) * SP + rt * 16 + 4 * lg, y);
                }
            DSG_LDS_BARRIER();
            vt_store_block<P, 64 * TNW>(g, stage, SP, m0, BM, ng * 64 * TNW - 2 * (g.H * g.hd), tid);
            return;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int mt = m0 + rt * 16;
        if (mt >= g.MT * 16) continue;                      // wave-uniform: row tiles past the end of the batch
#pragma unroll
        for (int t = 0; t < TNW; ++t)
            gemm_epilogue_tile<P, EPI>(g, mt, (nt0 + t) * 16, lr, lg, 0, swapped[t], acc[rt][t], ops[rt][t], k1, k2, k3, k4, k5);
    }
}

// Large-K block GEMM (linear2: K = ff; pose embedding: K = J): a 32 x 32 output block per workgroup, the 4 waves split
// the k-blocks of the workgroup's K range (blockIdx.z of g.KS splits for EPI_PARTIAL) and every wave requests ALL fragments
// of its share -- 2 A + 2 B per k-block, up to KPW k-blocks -- before its first MFMA: a round trip to data another XCD
// produced costs 1.5-2 us at these sizes, so the kernel makes exactly one (a first version with 64 x 64 blocks that streamed
// its K range in four chunks took 10 us for linear2 against 7.8 us for the 16 x 16 tile kernel; profiles/r02_c_*).  Each
// fragment pair feeds 4 MFMAs (the 16 x 16 kernel: 1), halving the bytes pulled through L2 per output.  The 4 partial blocks
// are reduced through LDS in a fixed order; wave w finishes tile (w >> 1, w & 1).
template <class P, int EPI, int KPW>
__global__ __launch_bounds__(256) void k_gemm_blk_k(const GemmArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int RT = 2, CT = 2;
    static_assert(EPI == EPI_RESID || EPI == EPI_PARTIAL, "direct A operand, fp32 output");
    __shared__ __attribute__((aligned(16))) float red[4][RT * CT][64][4];      // every wave's partial 32 x 32 block
    preload_kernargs(g);
    const int NG = g.NT / CT;
    int ng = xcd_ngroup<P>(), mb = blockIdx.y;
    const int ks = blockIdx.z;
    const int MB = (g.MT + RT - 1) / RT;
    if constexpr (EPI == EPI_PARTIAL) {
        if (mb >= MB) {      // extra grid row: step bookkeeping (see gemm_body)
            if (g.ctl && blockIdx.x == 0 && ks == 0 && threadIdx.x == 0) step_advance_B(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    if (!xcd_deal_groups(NG, MB, ng, mb)) return;      // (round 6: the pose embedding of the DSG+ widths is 12 column groups: XCDs 0-3 held two, 4-7 one)
    if (ng >= NG || mb >= MB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int m0 = mb * 16 * RT, nt0 = ng * CT;
    const int KBtot = g.KBtot, kb_last = KBtot - 1;
    const int kb_lo_wg = ks * g.kb_per_split, kb_hi_wg = min(kb_lo_wg + g.kb_per_split, KBtot);
    const int per = (kb_hi_wg - kb_lo_wg + 3) >> 2;
    const int kb_lo = min(kb_lo_wg + wave * per, kb_hi_wg), kb_hi = min(kb_lo + per, kb_hi_wg);
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    const int mt_last = g.MT - 1;
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave finishes tile (wave >> 1, wave & 1): its bias / residual operands travel with the fragments
    TileOps ops;
    gemm_prefetch_tile<P, EPI>(g, min(m0 + (wave >> 1) * 16, mt_last * 16), (nt0 + (wave & 1)) * 16, lr, lg, 0, ops);
    for (int kb0 = kb_lo; kb0 < kb_hi; kb0 += KPW) {       // one pass when the wave's share fits (the sizes of the path)
        f32x4 af[KPW][RT], bf[KPW][CT];
#pragma unroll
        for (int c = 0; c < KPW; ++c) {
            const int kb = min(kb0 + c, kb_last);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int mt = min((m0 >> 4) + rt, mt_last);        // clamped (rows past the end are computed and dropped)
                const elem* ap = g.a_frag ? (const elem*)g.A + ((size_t)(mt * KBtot + kb) * 64 + lane) * P::E
                                          : (const elem*)g.A + (size_t)(mt * 16 + lr) * g.lda + (size_t)kb * P::KB + P::E * lg;
                af[c][rt] = *(const f32x4*)ap;
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) bf[c][t] = wbase[((size_t)(nt0 + t) * KBtot + kb) * 64];
        }
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int c = 0; c < KPW; ++c) {
            const bool live = kb0 + c < kb_hi;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 a = live ? af[c][rt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[rt][t] = P::mma(bf[c][t], a, acc[rt][t]);      // D[col 4lg+r][row lr]
            }
        }
    }
    // ---- reduce the 4 K ranges through LDS in a fixed order
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < CT; ++t) *(f32x4*)&red[wave][rt * CT + t][lane][0] = acc[rt][t];
    DSG_LDS_BARRIER();
    const int rt = wave >> 1, t = wave & 1;
    const int mt = m0 + rt * 16;
    if (mt >= g.MT * 16) return;
    f32x4 sum = *(const f32x4*)&red[0][rt * CT + t][lane][0];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) sum += *(const f32x4*)&red[w2][rt * CT + t][lane][0];
    gemm_epilogue_tile<P, EPI>(g, mt, (nt0 + t) * 16, lr, lg, ks, true, sum, ops, 0.f, 0.f, 0.f, 0.f, 0.f);
}

}  // namespace dsg
