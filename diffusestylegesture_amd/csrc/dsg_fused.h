// dsg_fused.h -- latency-mode kernels: the same arithmetic as dsg_kernels.h, regrouped so that one denoising step is
// 2 + 4*L dispatches (k_inloc, L x {QKV GEMM, k_attn, k_mid, linear2 GEMM}, pose head) instead of 3 + 5*L; with
// k_attn_mid (batch 1) 2 + 3*L.  At batch 1 every launch is a latency chain (kernel boundary + dependent loads +
// load -> MFMA -> store), so the lever is the NUMBER of dependent launches, not FLOPs: the fused kernels recompute
// small things redundantly (K/V of a head per query tile, out_proj + LayerNorm per hidden slice, the pose embedding
// of the previous window's frames) to avoid an all-to-all hand-off.
//
//   k_inloc     = k_in + k_loc      pose-embedding GEMM for {previous, own} window x one local head (K split over
//                                   the 4 waves, reduced in LDS) -> rotary -> local attention -> token -> rotary
//   k_mid       = out_proj + residual + LayerNorm1 (rows owned whole by the workgroup) + linear1 slice + GELU
//   (linear2 + residual and the pose head + sampler update stay k_gemm<RESID> / k_gemm<OUT> of dsg_kernels.h)
// Reference arithmetic: main/model/mdm.py:196-233 and torch's TransformerEncoderLayer (post-norm) -- see dsg_kernels.h.
#pragma once
#include "dsg_kernels.h"

namespace dsg {

// ---------------------------------------------------------------------------------------------------------
// k_inloc
// ---------------------------------------------------------------------------------------------------------
struct InLocArgs {
    const void* xs;         // [B*T (+pad)][Jp] P::elem state (bf16 shadow, or the fp32 master in fp32 mode)
    int Jp;
    const void* Wp;         // packed folded weight [D/16][KBtot][64][16 B]
    int KBtot;
    LocArgs loc;            // Cf, TE2, TE, emb1, ctr/tmodel/t_arr, rotary tables, mask, dims, X0/X0a (partial unused)
    StepCtl* ctl_upd;       // first kernel of a step: block 0 advances the B side of the step control at its end
    StepTables st; int n_tab;
};

template <class P, int HD, int W>
__device__ __forceinline__ void inloc_body(const InLocArgs& g) {
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    preload_kernargs(g);
    const LocArgs& a = g.loc;
    if (blockIdx.z == gridDim.z - 1) {      // an EXTRA grid slice: its first workgroup does the step bookkeeping (see StepCtl), off the critical path
        if (g.ctl_upd && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) step_advance_B<P>(g.ctl_upd, g.st, g.n_tab);
        return;
    }
    constexpr int W2 = 2 * W, half = HD / 2, NP1 = W2 * half, NPI = (NP1 + 255) / 256;
    constexpr int NSI = (W * 32 + 255) / 256, NP2 = W * half, NPO = (NP2 + 255) / 256;
    constexpr int NL = HD >= 16 ? HD / 16 : 1;                 // 16-col tiles covering the head
    __shared__ __attribute__((aligned(16))) float red[4][2][NL][64][4];      // per-wave partial accumulators
    __shared__ __attribute__((aligned(16))) float rot[W2][HD + 4];
    __shared__ float sc[W][W2 + 2];
    const int h = (int)blockIdx.x, w = blockIdx.y, b = blockIdx.z;            // grid (local heads, windows, batch [+1 bookkeeping])
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int col0 = h * HD, ntok = a.T + 1, f0 = (w - 1) * W;
    const int* tp = a.ctl ? &a.ctl->tA : a.t_arr + b;      // select the ADDRESS, then one unconditional load
    const int t = ldw<P>(tp);

    // ---- (1) window constants, rotary tables and the key mask: issued first, unconditionally (clamped indices)
    float lo[NPI], hi[NPI], c1[NPI], s1[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = min(tid + 256 * i, NP1 - 1);
        const int r = p / half, dd = p % half, f = max(f0 + r, 0);
        const size_t base = ((size_t)b * a.T + f) * a.D + col0 + dd;
        lo[i] = a.Cf[base];
        hi[i] = a.Cf[base + half];
        c1[i] = a.rcos[f * half + dd]; s1[i] = a.rsin[f * half + dd];
    }
    float c2[NPO], s2[NPO];
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
        const int p = min(tid + 256 * i, NP2 - 1);
        const int pos = w * W + p / half + 1;
        c2[i] = a.rcos[pos * half + p % half]; s2[i] = a.rsin[pos * half + p % half];
    }
    const int mrow = fdiv(b * a.Hl + h, a.inv_mask_div);
    bool keep[NSI];
#pragma unroll
    for (int i = 0; i < NSI; ++i) {
        const int idx = tid + 256 * i, q = idx >> 5, j = idx & 31, fk = f0 + j;
        const unsigned char mk = a.mask[(size_t)mrow * a.T + min(max(fk, 0), a.T - 1)];
        keep[i] = ((int)(q < W) & (int)(j < W2) & ((int)(a.nomask != 0) | ((int)(fk >= 0) & (int)(mk != 0)))) != 0;
    }
    const int tc = min(tid, HD - 1);
    float tokv = a.emb1[(size_t)b * a.D + col0 + tc];

    // ---- (2) pose-embedding GEMM for the 2W frames x this head's columns; this wave's share of K
    const int ntile0 = col0 / 16;
    const int kbw = (g.KBtot + 3) / 4;
    const int kb_lo = wave * kbw, kb_hi = min(kb_lo + kbw, g.KBtot);
    f32x4 acc[2][NL];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NL; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t arow[2];                                  // element offsets into g.xs
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int f = f0 + mt * 16 + lr;
        f = f < 0 ? 0 : (f > a.T - 1 ? a.T - 1 : f);          // rows outside the window pair are computed and ignored
        arow[mt] = ((size_t)b * a.T + f) * g.Jp + P::E * lg;
    }
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    constexpr int CH = 9;
    const int kb_last = g.KBtot - 1;
    f32x4 af[CH][2];
    typename P::wfrag bf[CH][NL];
    auto load_chunk = [&](int kb0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kb = min(kb0 + c, kb_last);                  // clamped, never predicated
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[c][mt] = lda16<P>(g.xs, (arow[mt] + (size_t)kb * P::KB) * ES);
#pragma unroll
            for (int nt = 0; nt < NL; ++nt) bf[c][nt] = P::wload(wbase, (size_t)(ntile0 + nt) * g.KBtot + kb);
        }
    };
    load_chunk(kb_lo);
    // the only loads that depend on the model timestep go out AFTER the GEMM fragments (t was requested first)
    float te_lo[NPI], te_hi[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = min(tid + 256 * i, NP1 - 1), dd = p % half;
        te_lo[i] = a.TE2[(size_t)t * a.D + col0 + dd];
        te_hi[i] = a.TE2[(size_t)t * a.D + col0 + dd + half];
    }
    tokv += a.TE[(size_t)t * a.D + col0 + tc];
    for (int kb0 = kb_lo; kb0 < kb_hi; kb0 += CH) {
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool live = kb0 + c < kb_hi;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 av = live ? af[c][mt] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NL; ++nt) acc[mt][nt] = P::mma_a(av, bf[c][nt], acc[mt][nt]);   // D[row 4lg+r][col lr]
            }
        }
        if (kb0 + CH < kb_hi) load_chunk(kb0 + CH);
    }
    if (w == 0 && tid < HD) {
        a.X0[(size_t)(b * ntok) * a.D + col0 + tid] = tokv;
        ((elem*)a.X0a)[a.x0a_frag ? (size_t)qk_off<P>(b * ntok, col0 + tid, a.D / P::KB) : (size_t)(b * ntok) * a.D + col0 + tid] = P::cvt(tokv);
    }

    // ---- (3) reduce the 4 K-slices, add the constants, apply the rotary -> rot
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NL; ++nt) *(f32x4*)&red[wave][mt][nt][lane][0] = acc[mt][nt];
    DSG_LDS_BARRIER();
    const int cshift = col0 - ntile0 * 16;                     // head narrower than a tile: offset inside the tile
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = tid + 256 * i;
        if (p < NP1) {
            const int r = p / half, dd = p % half, f = f0 + r;
            const int mt = r >> 4, rr = r & 15, rg = rr & 3;
            const int cl = dd + cshift, ch = dd + half + cshift;
            const int lnl = (rr >> 2) * 16 + (cl & 15), lnh = (rr >> 2) * 16 + (ch & 15);     // D[row 4*lg + reg][col lr]
            float vl = lo[i] + te_lo[i], vh = hi[i] + te_hi[i];
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { vl += red[wv][mt][cl >> 4][lnl][rg]; vh += red[wv][mt][ch >> 4][lnh][rg]; }
            rot[r][dd] = f >= 0 ? vl * c1[i] - vh * s1[i] : -1.0f;
            rot[r][dd + half] = f >= 0 ? vh * c1[i] + vl * s1[i] : -1.0f;
        }
    }
    DSG_LDS_BARRIER();
    local_attn_tail_valu<P, HD, W>(a, rot, sc, b, w, h, keep, c2, s2);
}
template <class P, int HD, int W>
__global__ __launch_bounds__(256) void k_inloc(const InLocArgs g) { DSG_TL_SCOPE(); inloc_body<P, HD, W>(g); }

// ---------------------------------------------------------------------------------------------------------
// k_mid: pre1 = R + attn.Wo^T + bo ; x1 = LayerNorm1(pre1) ; hidden[:, slice] = gelu(x1.W1[slice]^T + b1)
// ---------------------------------------------------------------------------------------------------------
// linear1 fragments per wave held in registers by k_mid / k_attn_mid (and the out_proj chunk of k_mid): all KD of them up to 16
// (fp32 at D = 256 used to stream them in two serial chunks: round 4), 4 at the widths whose W_o share already fills the file
__host__ __device__ constexpr int mid_ch(int dt, int kd) { return dt >= 6 ? 4 : (kd > 8 ? 16 : 8); }
struct MidArgs {
    const void* A;          // attention output rows P::elem [rows][D]
    const float* R;         // residual rows fp32 [rows][D]
    const void* Wo; const float* bo;
    const float* ln_g; const float* ln_b;
    const void* W1; const float* b1;
    float* X1;              // LayerNorm1 output rows (fp32), written by hidden-slice 0
    void* hidden;           // [rows][ff] P::elem
    int M, MT, ff;
};

// Shared tail of k_mid / k_attn_mid: acc = out_proj result tiles of this wave (D[n = 4*lg + r][row = lr]); adds bias +
// residual, LayerNorm1 over whole rows (the 4 waves hold a row between them), stages the normalised rows in LDS,
// linear1 slice + GELU -> hidden, X1 written by hidden-slice 0.
template <class P, int DT>
__device__ __forceinline__ void mid_tail(const MidArgs& g, f32x4 (&acc)[DT], const f32x4 (&pbo)[DT], const f32x4 (&pr)[DT],
                                         const f32x4 (&pg)[DT], const f32x4 (&pbt)[DT], const f32x4 pb1,
                                         const typename P::wfrag (&w1f)[(DT * 64 / P::KB) <= mid_ch(DT, DT * 64 / P::KB) ? (DT * 64 / P::KB) : 1],
                                         const f32x4* w1, char* a1, float (&red)[2][4][16], int m0, int ng, int n1t, int wave,
                                         int lr, int lg) {
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64;
    constexpr int KD = D / P::KB;
    constexpr int XP = D * ES + 16;
    constexpr int CH = mid_ch(DT, KD);
    // ---- residual + LayerNorm1 over whole rows (row lr: 4 lane groups x 4 waves hold its D values)
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) { acc[t] = acc[t] + pbo[t] + pr[t]; s += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]); }
    // (round 6: the statistics in ONE exchange -- ln_wave_moments, dsg_kernels.h -- one barrier instead of two: batch 1 107.9 -> 107.1 us per step, r06_cg_*)
    ln_wave_moments<DT>(acc, s, red[0][wave], red[1][wave], lr, lg);
    DSG_LDS_BARRIER();
    float mean, var;
    ln_combine_moments<4, D>(red[0], red[1], lr, mean, var);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    DSG_TL_MARK(7);      // mid_tail: residual + LayerNorm1 statistics (one barrier)
    const bool wr = ng == 0 && (m0 + lr) < g.M;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int n = (wave * DT + t) * 16 + 4 * lg;
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (acc[t][e] - mean) * rstd * pg[t][e] + pbt[t][e];
        P::store4_a((elem*)(a1 + lr * XP) + n, 16 * XP, y);
        acc[t] = y;                                   // written to X1 at the very end (keeps stores out of the vmcnt queue)
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(8);      // LayerNorm1 rows in LDS
    // ---- linear1 slice + GELU
    f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (KD <= CH) {
#pragma unroll
        for (int kb = 0; kb < KD; ++kb)
            c1 = P::mma_w(w1f[kb], P::aload(a1 + lr * XP + (kb * P::KB + P::E * lg) * ES, 16 * XP), c1);
    } else {
#pragma unroll
        for (int kb0 = 0; kb0 < KD; kb0 += CH) {
            typename P::wfrag bf[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) if (kb0 + c < KD) bf[c] = P::wload(w1, (size_t)n1t * KD + kb0 + c);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (kb0 + c < KD)
                    c1 = P::mma_w(bf[c], P::aload(a1 + lr * XP + ((kb0 + c) * P::KB + P::E * lg) * ES, 16 * XP), c1);
        }
    }
    DSG_TL_MARK(9);      // linear1 slice issued (W1 has landed)
    if (m0 + lr < g.M) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = gelu_erf<P>(c1[e] + pb1[e]);
        P::store4_afrag((elem*)g.hidden, (size_t)qk_off<P>(m0 + lr, n1t * 16 + 4 * lg, g.ff / P::KB), y);       // fragment-major: linear2's A operand
    }
    DSG_TL_MARK(10);     // GELU + `hidden` stores issued
    if (wr) {
#pragma unroll
        for (int t = 0; t < DT; ++t) *(f32x4*)(g.X1 + (size_t)(m0 + lr) * D + (wave * DT + t) * 16 + 4 * lg) = acc[t];
    }
}

// Activation and weight fragments are requested k-block by k-block, PD ahead of the MFMA that consumes them, so the first
// MFMA starts after ~25 loads instead of ~57 (measured: 10.7 k vs 12.3 k cycles per kernel).
template <class P, int DT>      // DT = D / 64 : 16-col tiles of the out_proj output per wave
__global__ __launch_bounds__(256) void k_mid(const MidArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64;
    constexpr int KD = D / P::KB;
    constexpr int XP = D * ES + 16;
    constexpr int CH = mid_ch(DT, KD);               // fragments in flight per chunk (DT tiles each): bounded by the register file
    __shared__ __attribute__((aligned(16))) char a1[16 * XP * P::AF];     // (PBF16W2: + the lo image)
    __shared__ float red[2][4][16];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];      // out_proj bias, LayerNorm1 scale / shift: one load per WORKGROUP
    preload_kernargs(g);
    const int NGH = g.ff / 64;
    const int ng = xcd_ngroup<P>(), mt = blockIdx.y;
    if (ng >= NGH) return;
    const int m0 = mt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const f32x4* wo = (const f32x4*)g.Wo + lane;
    const f32x4* w1 = (const f32x4*)g.W1 + lane;
    // The three per-column vectors are the same for every wave and row tile: fetched once per workgroup (one 16-byte load
    // by 3 D / 4 lanes) and read back from LDS, instead of 12 wave-wide loads per wave through the CU's load path, which
    // bounds this kernel (measured: -1060 cycles per kernel without those loads).
    constexpr int NV = (3 * D / 4 + 255) / 256;      // float4 loads per lane (1 for D <= 320, else 2)
    f32x4 vload[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = min(tid + 256 * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);      // clamped, never predicated
        vload[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
    }

    // ---- out_proj as a software pipeline.  The load phase of this kernel is bound by the CU's texture-address path
    //      (~65 x 1 KB wave loads per wave, 4 waves: measured ~90 cycles per load instruction), so the order of issue IS
    //      the schedule: weight / activation fragments first, PD k-blocks ahead of the MFMA that consumes them (the MFMAs
    //      then run in the shadow of the remaining load issue), and the operands of the later phases (bias + residual,
    //      LayerNorm scale/shift, linear1 fragments) last, in the order they are needed.
    constexpr int PD0 = DT >= 6 ? 2 : 4;             // k-blocks in flight ahead of the MFMA (register budget)
    constexpr int PD = PD0 < KD ? PD0 : KD;
    typename P::afrag af[KD];
    typename P::wfrag bf[KD][DT];
    f32x4 pbo[DT], pr[DT], pg[DT], pbt[DT], pb1;
    typename P::wfrag w1f[KD <= CH ? KD : 1];
    const int n1t = ng * 4 + wave;                   // this wave's 16-col tile of the hidden layer
#pragma unroll
    for (int kb = 0; kb < PD; ++kb) {
        af[kb] = P::aload_frag(g.A, (size_t)mt * KD + kb, lane);
#pragma unroll
        for (int t = 0; t < DT; ++t) bf[kb][t] = P::wload(wo, (size_t)(wave * DT + t) * KD + kb);
    }
    auto load_operands = [&]() {
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int n = (wave * DT + t) * 16 + 4 * lg;
            pr[t] = *(const f32x4*)(g.R + (size_t)(m0 + lr) * D + n);
        }
        if constexpr (KD <= CH) {
#pragma unroll
            for (int k2 = 0; k2 < KD; ++k2) w1f[k2] = P::wload(w1, (size_t)n1t * KD + k2);
        }
        pb1 = *(const f32x4*)(g.b1 + n1t * 16 + 4 * lg);
    };
    f32x4 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
        if (kb + PD < KD) {
            af[kb + PD] = P::aload_frag(g.A, (size_t)mt * KD + kb + PD, lane);
#pragma unroll
            for (int t = 0; t < DT; ++t) bf[kb + PD][t] = P::wload(wo, (size_t)(wave * DT + t) * KD + kb + PD);
        }
        if (kb + PD == KD) load_operands();           // all fragments requested: now the operands of the later phases
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = P::mma_w(bf[kb][t], af[kb], acc[t]);      // D[n 4lg+r][row lr]
        DSG_LOADS_ISSUED();
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = tid + 256 * i;
        if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];          // vecs[3][D] is contiguous: element e of the 3 D / 4 float4s
    }
    DSG_LDS_BARRIER();
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int n = (wave * DT + t) * 16 + 4 * lg;
        pbo[t] = *(const f32x4*)(&vecs[0][n]); pg[t] = *(const f32x4*)(&vecs[1][n]); pbt[t] = *(const f32x4*)(&vecs[2][n]);
    }
    mid_tail<P, DT>(g, acc, pbo, pr, pg, pbt, pb1, w1f, w1, a1, red, m0, ng, n1t, wave, lr, lg);
}


// ---------------------------------------------------------------------------------------------------------
// k_attn_mid: self-attention of the workgroup's 16 query rows (wave = head, H == 4, one batch element) staged in LDS,
// then exactly k_mid (out_proj + residual + LayerNorm1 + linear1 slice + GELU).  One launch less per layer: at batch 1
// a dependent launch costs ~3 us before any work, the attention itself ~1 us, and -- the 16 hidden-slices of a row tile
// recompute it -- idle CUs are free.  The load phase is bound by the CU's texture-address path, so the issue order is
// the schedule: Q/K/V^T fragments, then the first out_proj weight k-blocks (in flight during the softmax), then the
// operands of the later phases in the order they are needed.
// ---------------------------------------------------------------------------------------------------------
struct AttnMidArgs {
    MidArgs mid;                                      // mid.A is unused (the attention output never leaves LDS)
    const void* q; const void* k; const void* vt;     // [H][Tp][hd], [H][Tp][hd], [H][hd][Tp]  (batch element 0)
    int ntok, Tp;
};

template <class P, int DT, int NKT>      // D = 64 DT, H = 4, hd = 16 DT, Tp = 16 NKT
__device__ __forceinline__ void attn_mid_body(const AttnMidArgs& ga) {
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64, HD = DT * 16;
    constexpr int KD = D / P::KB;                    // k-blocks of the out_proj / linear1 reductions
    constexpr int KDH = HD / P::KB;                  // k-blocks over the head dim
    constexpr int XP = D * ES + 16;
    constexpr int CH = mid_ch(DT, KD);
    constexpr int ND = HD / 16;
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;   // PV k-blocks
    constexpr int PD = 4;                            // out_proj k-blocks in flight ahead of the MFMA
    constexpr int PDA = KD * P::WF <= 8 ? KD : PD;   // ... of which this many are requested before the attention math (register budget: fp32 and the
                                                     // two-register weight fragments of bf16w2 hold half as many k-blocks)
    static_assert(KDH >= 1 && PDA >= PD && PDA <= KD, "shape");
    static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
    __shared__ __attribute__((aligned(16))) char aT[16 * XP * P::AF];     // attention output rows (MFMA element type; PBF16W2: + the lo image)
    __shared__ __attribute__((aligned(16))) char a1[16 * XP * P::AF];     // LayerNorm1 output rows
    __shared__ float red[2][4][16];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];      // out_proj bias, LayerNorm1 scale / shift: one load per WORKGROUP (see k_mid)
    preload_kernargs(ga);
    const MidArgs& g = ga.mid;
    const int NGH = g.ff / 64;
    const int ng = xcd_ngroup<P>(), mt = blockIdx.y;
    if (ng >= NGH) return;
    const int m0 = mt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const f32x4* wo = (const f32x4*)g.Wo + lane;
    const f32x4* w1 = (const f32x4*)g.W1 + lane;
    const int h = wave;                              // head of this wave
    const size_t Q = (size_t)h * ga.Tp * HD, K = Q, VT = (size_t)h * HD * ga.Tp;      // element offsets of this head in q / k / vt

    // ---- (1) attention operand fragments
    f32x4 qf[KDH], kf[NKT][KDH], vfr[ND][NVF];
#pragma unroll
    for (int kb = 0; kb < KDH; ++kb) qf[kb] = lda16<P>(ga.q, (Q + (size_t)((mt * KDH + kb) * 64 + lane) * P::E) * ES);     // fragment-major (qk_off)
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) kf[nt][kb] = lda16<P>(ga.k, (K + (size_t)((nt * KDH + kb) * 64 + lane) * P::E) * ES);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) vfr[dt][kb] = lda16<P>(ga.vt, (VT + (size_t)((dt * NVF + kb) * 64 + lane) * P::E) * ES);       // fragment-major (vt_off)
    }
    DSG_LOADS_ISSUED();
    DSG_TL_MARK(0);      // Q / K / V^T requested
    // ---- (2) everything the later phases need is requested WHILE the attention math runs, a few loads per slot: a wave
    //      issues in order, and a burst of ~70 loads stalls it in the issue stage for as long as the texture path needs
    //      to drain them (~100 cycles each with 4 waves loading) -- time in which no softmax instruction can run.
    //      Spread between the softmax stages the same loads cost nothing and have arrived when out_proj starts.
    typename P::wfrag bf[KD][DT];
    f32x4 pbo[DT], pr[DT], pg[DT], pbt[DT], pb1;
    typename P::wfrag w1f[KD <= CH ? KD : 1];
    const int n1t = ng * 4 + wave;                   // this wave's 16-col tile of the hidden layer
    constexpr int G = PDA + 4;                       // load groups: PDA weight k-blocks, bias+residual, LN scale+shift, linear1 fragments, linear1 bias
    constexpr int S = 2 * NKT + ND;                  // slots: after each key tile of the max pass, of the exp pass, after each PV tile
    constexpr int NV = (3 * D / 4 + 255) / 256;
    f32x4 vload[NV];
    auto issue_group = [&](int gi) {
        if (gi < PDA) {
#pragma unroll
            for (int t = 0; t < DT; ++t) bf[gi][t] = P::wload(wo, (size_t)(wave * DT + t) * KD + gi);
        } else if (gi == PDA) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int n = (wave * DT + t) * 16 + 4 * lg;
                pr[t] = lda16<P>(g.R, ((size_t)(m0 + lr) * D + n) * sizeof(float));
            }
        } else if (gi == PDA + 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = min(tid + 256 * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);      // clamped, never predicated
                vload[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
            }
        } else if (gi == PDA + 2) {
            if constexpr (KD <= CH) {
#pragma unroll
                for (int k2 = 0; k2 < KD; ++k2) w1f[k2] = P::wload(w1, (size_t)n1t * KD + k2);
            }
        } else {
            pb1 = *(const f32x4*)(g.b1 + n1t * 16 + 4 * lg);
        }
    };
#define DSG_ISSUE_SLOT(slot)                                                                  \
    do {                                                                                      \
        _Pragma("unroll") for (int gi = ((slot) * G) / S; gi < (((slot) + 1) * G) / S; ++gi) issue_group(gi); \
        DSG_LOADS_ISSUED();                                                                   \
    } while (0)

    // ---- (3) S^T = K Q^T, softmax over the keys (D[key = 4*lg + r][query = lr])
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
        s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(kf[nt][kb], qf[kb], s[nt]);
    }
    DSG_TL_MARK(1);      // Q K^T issued (Q / K have landed)
    const float scale = 1.0f / sqrtf((float)HD);
    float mx = -DSG_FLT_MAX;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = nt * 16 + 4 * lg + r;
            const float v = key < ga.ntok ? s[nt][r] * scale : -DSG_FLT_MAX;
            s[nt][r] = v;
            mx = fmaxf(mx, v);
        }
        DSG_ISSUE_SLOT(nt);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    DSG_TL_MARK(2);      // scores scaled, row max
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = nt * 16 + 4 * lg + r;
            const float pv = key < ga.ntok ? P::exp_sm(s[nt][r] - mx) : 0.f;
            s[nt][r] = pv;
            sum += pv;
        }
        DSG_ISSUE_SLOT(NKT + nt);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    DSG_TL_MARK(3);      // exp + row sum
    f32x4 pfr[NVF];                                  // P^T fragments: exactly the values this lane already holds
#pragma unroll
    for (int kb = 0; kb < NVF; ++kb) {
        if constexpr (P::E == 4) {
            pfr[kb] = s[kb];
        } else {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
    }
    // ---- (4) O^T = V^T P^T  -> LDS rows (the same rounding point as the global attention buffer of k_attn)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) o = P::mma(vfr[dt][kb], pfr[kb], o);     // D[dim = 4*lg + r][query = lr]
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
        P::store4_a((elem*)(aT + lr * XP) + h * HD + dt * 16 + 4 * lg, 16 * XP, y);
        DSG_ISSUE_SLOT(2 * NKT + dt);
    }
    DSG_TL_MARK(4);      // P V issued, attention rows on their way to LDS
#undef DSG_ISSUE_SLOT
    // fp32 (KD = 16 at D = 256): K / V^T are dead now -- ALL remaining W_o k-blocks in one batch instead of a PD-deep pipeline of
    // exposed L2 round trips inside the out_proj loop (round 4: k_attn_mid<PF32, 4, 6> 20.4 us, 62 % of the fp32 step)
    constexpr bool ALL_AFTER = KD > PDA && (KD - PDA) * DT * P::WF <= 64;
    if constexpr (ALL_AFTER) {
#pragma unroll
        for (int kb = PDA; kb < KD; ++kb)
#pragma unroll
            for (int t = 0; t < DT; ++t) bf[kb][t] = P::wload(wo, (size_t)(wave * DT + t) * KD + kb);
        DSG_LOADS_ISSUED();
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(5);      // all four heads' rows are in LDS

    // ---- (5) out_proj from the LDS rows, remaining weight k-blocks PD ahead
    f32x4 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
        if (!ALL_AFTER && kb + PD >= PDA && kb + PD < KD) {
#pragma unroll
            for (int t = 0; t < DT; ++t) bf[kb + PD][t] = P::wload(wo, (size_t)(wave * DT + t) * KD + kb + PD);
        }
        const typename P::afrag af = P::aload(aT + lr * XP + (kb * P::KB + P::E * lg) * ES, 16 * XP);
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = P::mma_w(bf[kb][t], af, acc[t]);      // D[n 4lg+r][row lr]
    }
    DSG_TL_MARK(6);      // out_proj issued (W_o has landed)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = tid + 256 * i;
        if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];
    }
    DSG_LDS_BARRIER();
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int n = (wave * DT + t) * 16 + 4 * lg;
        pbo[t] = *(const f32x4*)(&vecs[0][n]); pg[t] = *(const f32x4*)(&vecs[1][n]); pbt[t] = *(const f32x4*)(&vecs[2][n]);
    }
    mid_tail<P, DT>(g, acc, pbo, pr, pg, pbt, pb1, w1f, w1, a1, red, m0, ng, n1t, wave, lr, lg);
}
template <class P, int DT, int NKT>
__global__ __launch_bounds__(256) void k_attn_mid(const AttnMidArgs ga) { DSG_TL_SCOPE(); attn_mid_body<P, DT, NKT>(ga); }

// ---------------------------------------------------------------------------------------------------------
// k_attn_op (batched step): self-attention of one (batch element, 16-query tile) for all 4 heads (wave = head), then
// out_proj + bias + residual + LayerNorm1 on the 16 full rows -- k_attn_mid without the linear1 slice, so nothing is
// recomputed per hidden slice and it scales with the batch (grid = query tiles x batch).  One dispatch and one round trip
// of the attention rows less per layer than k_attn -> out_proj GEMM, and linear1 reads LayerNorm1'ed rows in the GEMM type
// (fragment-major) instead of normalising fp32 rows again in each of its column groups.
// ---------------------------------------------------------------------------------------------------------
struct AttnOpArgs {
    const void* q; const void* k; const void* vt;     // [B][H][Tp][hd], [B][H][Tp][hd], [B][H][hd][Tp]
    const float* R;         // residual rows fp32 [rows][D]
    const void* Wo; const float* bo;
    const float* ln_g; const float* ln_b;
    float* X1;              // LayerNorm1 output rows fp32 [rows][D] (residual of linear2)
    void* X1a;              // the same rows in the GEMM type, fragment-major (linear1's A operand)
    int B, ntok, Tp;
};

template <class P, int DT, int NKT>      // D = 64 DT, H = 4, hd = 16 DT, Tp = 16 NKT
__global__ __launch_bounds__(256) void k_attn_op(const AttnOpArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64, HD = DT * 16;
    constexpr int KD = D / P::KB, KDH = HD / P::KB;
    constexpr int XP = D * ES + 16;
    constexpr int ND = HD / 16;
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;   // PV k-blocks
    static_assert(KDH >= 1 && KD <= 8, "shape");
    static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
    __shared__ __attribute__((aligned(16))) char aT[16 * XP];      // attention output rows (MFMA element type)
    __shared__ float red[2][4][16];
    preload_kernargs(g);
    const int qt = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int h = wave;
    const size_t bh = (size_t)b * 4 + h;
    const elem* Q = (const elem*)g.q + bh * g.Tp * HD;
    const elem* K = (const elem*)g.k + bh * g.Tp * HD;
    const elem* VT = (const elem*)g.vt + bh * HD * g.Tp;
    const f32x4* wo = (const f32x4*)g.Wo + lane;
    // ---- every load of the workgroup up front: attention operands, W_o fragments, bias / LayerNorm vectors, residual rows
    f32x4 qf[KDH], kf[NKT][KDH], vfr[ND][NVF];
#pragma unroll
    for (int kb = 0; kb < KDH; ++kb) qf[kb] = *(const f32x4*)(Q + (size_t)((qt * KDH + kb) * 64 + lane) * P::E);
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) kf[nt][kb] = *(const f32x4*)(K + (size_t)((nt * KDH + kb) * 64 + lane) * P::E);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) vfr[dt][kb] = *(const f32x4*)(VT + (size_t)((dt * NVF + kb) * 64 + lane) * P::E);
    const int tq = qt * 16 + lr;                                   // this lane's token (row of the 16-row tile)
    const bool rowok = tq < g.ntok;
    const size_t m = (size_t)b * g.ntok + (rowok ? tq : g.ntok - 1);      // clamped: unconditional loads, predicated stores
    f32x4 pr[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) pr[t] = *(const f32x4*)(g.R + m * D + (wave * DT + t) * 16 + 4 * lg);
    // out_proj bias, LayerNorm1 scale / shift: one 16-byte load per lane for the workgroup, read back from LDS (see k_mid)
    constexpr int NV = (3 * D / 4 + 255) / 256;
    f32x4 vload[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = min(tid + 256 * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);
        vload[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
    }
    DSG_LOADS_ISSUED();
    // ---- S^T = K Q^T, softmax over the keys (D[key = 4*lg + r][query = lr])
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
        s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(kf[nt][kb], qf[kb], s[nt]);
    }
    // W_o: the first half of the k-blocks of this wave's DT column tiles now (in flight during the softmax), the second half
    // once the V^T fragments are dead (register budget)
    constexpr int KH = (KD + 1) / 2;
    f32x4 bf[KD][DT];
#pragma unroll
    for (int kb = 0; kb < KH; ++kb)
#pragma unroll
        for (int t = 0; t < DT; ++t) bf[kb][t] = wo[((size_t)(wave * DT + t) * KD + kb) * 64];
    const float scale = 1.0f / sqrtf((float)HD);
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
    f32x4 pfr[NVF];
#pragma unroll
    for (int kb = 0; kb < NVF; ++kb) {
        if constexpr (P::E == 4) {
            pfr[kb] = s[kb];
        } else {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
    }
    // ---- O^T = V^T P^T -> LDS rows (the rounding point of the attention buffer of k_attn)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) o = P::mma(vfr[dt][kb], pfr[kb], o);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
        P::store4((elem*)(aT + lr * XP) + h * HD + dt * 16 + 4 * lg, y);
    }
#pragma unroll
    for (int kb = KH; kb < KD; ++kb)
#pragma unroll
        for (int t = 0; t < DT; ++t) bf[kb][t] = wo[((size_t)(wave * DT + t) * KD + kb) * 64];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = tid + 256 * i;
        if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];
    }
    DSG_LDS_BARRIER();
    // ---- out_proj from the LDS rows
    f32x4 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
        const f32x4 af = *(const f32x4*)(aT + lr * XP + (kb * P::KB + P::E * lg) * ES);
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[t] = P::mma(bf[kb][t], af, acc[t]);      // D[n 4lg+r][row lr]
    }
    // ---- residual + LayerNorm1 over whole rows (row lr: 4 lane groups x 4 waves hold its D values)
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const f32x4 pbo = *(const f32x4*)(&vecs[0][(wave * DT + t) * 16 + 4 * lg]);
        acc[t] = acc[t] + pbo + pr[t];
        sm += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    }
    sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
    if (lg == 0) red[0][wave][lr] = sm;
    DSG_LDS_BARRIER();
    const float mean = ((red[0][0][lr] + red[0][1][lr]) + (red[0][2][lr] + red[0][3][lr])) / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc[t][e] - mean; qv = __builtin_fmaf(d, d, qv); }      // explicit fma: k_attn_op2 must round alike
    qv += __shfl_xor(qv, 16); qv += __shfl_xor(qv, 32);
    if (lg == 0) red[1][wave][lr] = qv;
    DSG_LDS_BARRIER();
    const float var = ((red[1][0][lr] + red[1][1][lr]) + (red[1][2][lr] + red[1][3][lr])) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (rowok) {
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int n = (wave * DT + t) * 16 + 4 * lg;
            const f32x4 pg = *(const f32x4*)(&vecs[1][n]), pbt = *(const f32x4*)(&vecs[2][n]);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[t][e] - mean) * rstd, pg[e], pbt[e]);
            *(f32x4*)(g.X1 + m * D + n) = y;
            P::store4((elem*)g.X1a + qk_off<P>((int)m, n, D / P::KB), y);
        }
    }
}

// (Round 4: "bit-identical" with k_attn_op2 was only true under the emulator -- on the device the compiler contracted the LayerNorm's mul + add into
// fma in k_attn_op and not in k_attn_op2, one ulp apart in ~10 % of the rows (tools/debug_op2.py): the two fma sites above are spelled out since.
// k_attn_op2 itself -- two query tiles per workgroup -- is retired (round 6): experiments/dsg_rejected_kernels.h.)

// ---------------------------------------------------------------------------------------------------------
// k_attn_op_w (round 4): k_attn_op for the shapes whose W_o does not fit the register file next to the attention -- the DSG+ widths
// in bf16 (D = 384 / 512: 12 / 16 k-blocks x 6 / 8 column tiles per wave) and fp32 at the ZEGGS / tiny widths (16 / 8 k-blocks of 4
// values).  Same arithmetic, rounding points and k order as k_attn_op; W_o streams through two register buffers of KC k-blocks:
// chunk 0 is requested once K is dead (in flight during the softmax and PV), chunk 1 once V^T is dead, chunk c + 2 right after the
// MFMAs of chunk c.  The batched sets (TILE / BLOCK) at these shapes lose one dispatch and one LayerNorm recompute per layer, as
// the ZEGGS bf16 path did in round 2 (1 x 16: 292 -> 254 us).
// ---------------------------------------------------------------------------------------------------------
template <class P, int DT, int NKT>      // D = 64 DT, H = 4, hd = 16 DT, Tp = 16 NKT
__global__ __launch_bounds__(256) void k_attn_op_w(const AttnOpArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64, HD = DT * 16;
    constexpr int KD = D / P::KB, KDH = HD / P::KB;
    constexpr int XP = D * ES + 16;
    constexpr int ND = HD / 16;
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;   // PV k-blocks
    constexpr int KC = 4, NC = KD / KC;              // W_o chunk: KC k-blocks x DT column tiles per wave
    static_assert(KDH >= 1 && KD % KC == 0 && NC >= 2, "shape");
    static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
    __shared__ __attribute__((aligned(16))) char aT[16 * XP];      // attention output rows (MFMA element type)
    __shared__ float red[2][4][16];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];
    preload_kernargs(g);
    const int qt = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int h = wave;
    const size_t bh = (size_t)b * 4 + h;
    const elem* Q = (const elem*)g.q + bh * g.Tp * HD;
    const elem* K = (const elem*)g.k + bh * g.Tp * HD;
    const elem* VT = (const elem*)g.vt + bh * HD * g.Tp;
    const f32x4* wo = (const f32x4*)g.Wo + lane;
    f32x4 qf[KDH], kf[NKT][KDH], vfr[ND][NVF];
#pragma unroll
    for (int kb = 0; kb < KDH; ++kb) qf[kb] = *(const f32x4*)(Q + (size_t)((qt * KDH + kb) * 64 + lane) * P::E);
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) kf[nt][kb] = *(const f32x4*)(K + (size_t)((nt * KDH + kb) * 64 + lane) * P::E);
    // V^T with the other operands when the register file holds everything (one round trip), else once K is dead (D = 512)
    constexpr bool V_EARLY = (KDH + NKT * KDH + ND * NVF + DT + 4) * 4 <= 340;
    auto load_v = [&]() {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int kb = 0; kb < NVF; ++kb) vfr[dt][kb] = *(const f32x4*)(VT + (size_t)((dt * NVF + kb) * 64 + lane) * P::E);
    };
    if constexpr (V_EARLY) load_v();
    const int tq = qt * 16 + lr;
    const bool rowok = tq < g.ntok;
    const size_t m = (size_t)b * g.ntok + (rowok ? tq : g.ntok - 1);      // clamped: unconditional loads, predicated stores
    f32x4 pr[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) pr[t] = *(const f32x4*)(g.R + m * D + (wave * DT + t) * 16 + 4 * lg);
    constexpr int NV = (3 * D / 4 + 255) / 256;
    f32x4 vload[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = min(tid + 256 * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);
        vload[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
    }
    DSG_LOADS_ISSUED();
    // ---- S^T = K Q^T (D[key = 4*lg + r][query = lr])
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
        s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(kf[nt][kb], qf[kb], s[nt]);
    }
    f32x4 bf[2][KC][DT];
    auto load_wo = [&](int c, int buf) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int t = 0; t < DT; ++t) bf[buf][k][t] = wo[((size_t)(wave * DT + t) * KD + c * KC + k) * 64];
    };
    if constexpr (!V_EARLY) load_v();
    load_wo(0, 0);                                     // K is dead: chunk 0 in flight during the softmax and PV
    DSG_LOADS_ISSUED();
    const float scale = 1.0f / sqrtf((float)HD);
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
    f32x4 pfr[NVF];
#pragma unroll
    for (int kb = 0; kb < NVF; ++kb) {
        if constexpr (P::E == 4) {
            pfr[kb] = s[kb];
        } else {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
    }
    // ---- O^T = V^T P^T -> LDS rows (the rounding point of the attention buffer of k_attn)
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) o = P::mma(vfr[dt][kb], pfr[kb], o);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
        P::store4((elem*)(aT + lr * XP) + h * HD + dt * 16 + 4 * lg, y);
    }
    DSG_LOADS_ISSUED();
    load_wo(1, 1);                                     // V^T is dead
    DSG_LOADS_ISSUED();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = tid + 256 * i;
        if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];
    }
    DSG_LDS_BARRIER();
    // ---- out_proj from the LDS rows, W_o chunk by chunk (k order 0 .. KD - 1 as in k_attn_op)
    f32x4 acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const f32x4 af = *(const f32x4*)(aT + lr * XP + ((c * KC + k) * P::KB + P::E * lg) * ES);
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[t] = P::mma(bf[c & 1][k][t], af, acc[t]);      // D[n 4lg+r][row lr]
        }
        if (c + 2 < NC) { DSG_LOADS_ISSUED(); load_wo(c + 2, c & 1); DSG_LOADS_ISSUED(); }
    }
    // ---- residual + LayerNorm1 over whole rows (row lr: 4 lane groups x 4 waves hold its D values); fma sites as in k_attn_op
    float sm = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const f32x4 pbo = *(const f32x4*)(&vecs[0][(wave * DT + t) * 16 + 4 * lg]);
        acc[t] = acc[t] + pbo + pr[t];
        sm += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
    }
    sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
    if (lg == 0) red[0][wave][lr] = sm;
    DSG_LDS_BARRIER();
    const float mean = ((red[0][0][lr] + red[0][1][lr]) + (red[0][2][lr] + red[0][3][lr])) / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc[t][e] - mean; qv = __builtin_fmaf(d, d, qv); }
    qv += __shfl_xor(qv, 16); qv += __shfl_xor(qv, 32);
    if (lg == 0) red[1][wave][lr] = qv;
    DSG_LDS_BARRIER();
    const float var = ((red[1][0][lr] + red[1][1][lr]) + (red[1][2][lr] + red[1][3][lr])) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    if (rowok) {
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int n = (wave * DT + t) * 16 + 4 * lg;
            const f32x4 pg = *(const f32x4*)(&vecs[1][n]), pbt = *(const f32x4*)(&vecs[2][n]);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[t][e] - mean) * rstd, pg[e], pbt[e]);
            *(f32x4*)(g.X1 + m * D + n) = y;
            P::store4((elem*)g.X1a + qk_off<P>((int)m, n, D / P::KB), y);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// k_clip_attn (round 5, BLOCK set): the attention of an encoder layer per (clip, head) -- the workgroup projects ITS head's Q / K / V
// slices of the clip's rows (the LayerNorm2 rows of the previous layer, already in the GEMM type), keeps them in LDS and runs the
// self-attention of all NKT query tiles on them: the QKV GEMM disappears as a dispatch, K / V are projected once per (clip, head) and
// Q / K / V^T (2.2 MB of 8-byte uncached stores per launch at 16 clips) never cross the fabric.  Output: the attention rows in the
// GEMM type, fragment-major -- what k_attn writes; out_proj + residual + LayerNorm1 run as the prologue of k_ffn_part (OP = true).
// Every global byte is requested once per workgroup: 8 KB of rows per wave (staged through LDS: every wave needs all row tiles) and
// the head's 96 KB of W_qkv, column split over the waves.
// wave w: projection columns [w C/NW, (w+1) C/NW) of the head's C = 3 hd / 16 tiles for all row tiles (Q and K tiles as W . X^T: a lane holds
// 4 consecutive dims of one token; V tiles as X . W^T: 4 consecutive tokens of one dim = a V^T row), then the attention of query tile w
// (exactly k_attn on LDS operands: same rounding points -- Q / K / V and P in the GEMM type, softmax in fp32).  A row's result depends
// on nothing but its clip.  (First version, profiles/r05_h_*: with the head's share of out_proj as an fp32 slab per head the kernel was
// 10.7 us at 16 clips -- 1.9 us of it the 5.8 MB of slab stores, 3.8 the projection with a per-MFMA operand-order branch -- against
// 5.2 + 5.0 for the QKV GEMM + k_attn_op.)
// Reference arithmetic: nn.MultiheadAttention of torch's TransformerEncoderLayer (main/model/mdm.py:79-86), in_proj + softmax(QK^T/sqrt(hd))V.
// ---------------------------------------------------------------------------------------------------------
struct ClipAttnArgs {
    const void* X;          // [rows][D] rows in the GEMM type, fragment-major over the flattened (clip, token) rows (qk_off)
    const void* Wqkv; const float* bqkv;
    void* out;              // attention rows [rows][D] in the GEMM type, fragment-major (qk_off): out_proj's A operand
    int B, ntok;
};

// projection of one row tile's CW column tiles, operand order fixed at compile time (V tiles: un-swapped)
template <class P, int CW, int KD, bool VT>
__device__ __forceinline__ void clip_proj_tile(const typename P::wfrag (&wf)[CW][KD], const typename P::afrag (&a)[KD], f32x4 (&acc)[CW]) {
#pragma unroll
    for (int kb = 0; kb < KD; ++kb)
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[j] = VT ? P::mma_a(a[kb], wf[j][kb], acc[j]) : P::mma_w(wf[j][kb], a[kb], acc[j]);
}

// bf16w2 (round 6, ROWS set): the rows (LayerNorm2 of the previous layer / the embedding output) arrive as hi + lo images and W_qkv as hi + lo
// fragments -- three MFMAs per fragment pair, like every weight product of the mode; Q / K / V / P stay single bf16 (PBF16W2) and the attention rows
// leave as a hi + lo pair (k_attn's rounding point in this mode).  The rows' LDS stage doubles (96 KB at the ZEGGS widths): one workgroup per CU.
template <class P, int DT, int NKT>      // D = 64 DT, H = 4, hd = 16 DT, Tp = 16 NKT, NKT waves
__global__ __launch_bounds__(64 * NKT, P::W2 ? 1 : 2) void k_clip_attn(const ClipAttnArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64, HD = DT * 16, NW = NKT;
    constexpr int KD = D / P::KB, KDH = HD / P::KB;
    constexpr int ND = HD / 16;                      // 16-dim tiles of a head
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;   // PV k-blocks
    constexpr int CT = 3 * ND, CW = CT / NW;         // projection column tiles of the head, per wave
    static_assert(CT % NW == 0 && HD % P::KB == 0, "shape");
    static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
    // LDS: the clip's rows as A fragments [NKT][KD]; Q, K [NKT][KDH]; V^T [ND][NVF] (1 KB fragments).  V^T takes the place of the rows once
    // every wave is done projecting (the V tiles wait in registers, already rounded: 2 VGPRs each): 72 KB instead of 84 at the ZEGGS
    // widths, i.e. TWO workgroups per CU -- with several lanes in flight (1024 workgroups at 4 x 64 clips) the load -> project -> attend
    // chain of one workgroup runs under the other's
    __shared__ __attribute__((aligned(16))) f32x4 xs[P::AF * NKT * KD][64];      // (bf16w2: the lo images behind the hi images)
    constexpr int XS_LO = NKT * KD * 1024;
    __shared__ __attribute__((aligned(16))) f32x4 qs[NKT * KDH][64];
    __shared__ __attribute__((aligned(16))) f32x4 ks[NKT * KDH][64];
    f32x4 (* const vs)[64] = xs;
    static_assert(ND * NVF <= NKT * KD, "V^T fits in the retired rows");
    static_assert(ES == 2, "the parked V tiles are bf16 quads");
    typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
    bf16x4v vkeep[NKT][CW];
    preload_kernargs(g);
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    // ---- every global load of the wave up front: its row tile, its projection columns
    const int tok_ld = min(wave * 16 + lr, g.ntok - 1);          // (tokens past the clip: a valid row, masked / dropped later)
    const int m_ld = b * g.ntok + tok_ld;
    typename P::afrag xf[KD];
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) xf[kb] = P::aload_off(g.X, (size_t)qk_off<P>(m_ld, kb * P::KB + P::E * lg, KD));
    const f32x4* wq = (const f32x4*)g.Wqkv + lane;
    const int ct0 = wave * CW;                       // first of this wave's column tiles in the head's [Q | K | V] order (ND tiles each)
    typename P::wfrag wf[CW][KD];
    f32x4 pb[CW];
    float pbs[CW];
    int which[CW], d0[CW];                           // per tile: 0 Q, 1 K, 2 V (wave-uniform) and its first dim inside the head
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        which[j] = (ct0 + j) / ND; d0[j] = ((ct0 + j) - which[j] * ND) * 16;
        const int nt = which[j] * (D / 16) + h * ND + d0[j] / 16;                 // column tile of the packed [3D / 16] in_proj weight
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) wf[j][kb] = P::wload(wq, (size_t)nt * KD + kb);
        pb[j] = *(const f32x4*)(g.bqkv + nt * 16 + 4 * lg);
        pbs[j] = g.bqkv[nt * 16 + lr];
    }
    DSG_LOADS_ISSUED();
    DSG_TL_MARK(0);      // rows + projection columns requested
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
        if constexpr (P::W2) { xs[wave * KD + kb][lane] = xf[kb].h; xs[NKT * KD + wave * KD + kb][lane] = xf[kb].l; }
        else xs[wave * KD + kb][lane] = xf[kb];
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(1);      // the clip's rows are in LDS (they have landed for every wave)
    // ---- (1) projection of the head's Q / K / V for all row tiles -> LDS in the attention kernels' fragment order.  The operand order
    //      of a tile (V: un-swapped) is decided ONCE per wave, outside the MFMA loops: all of a wave's tiles are of one kind at the ZEGGS
    //      widths (waves 0-3 Q / K, 4-5 V); a wave with both kinds (tiny dims) takes the tile-by-tile form
    const bool all_qk = which[CW - 1] < 2, all_v = which[0] == 2;
#pragma unroll
    for (int rt = 0; rt < NKT; ++rt) {
        typename P::afrag a[KD];
        f32x4 acc[CW];
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) a[kb] = P::aload((const char*)&xs[rt * KD + kb][lane], XS_LO);
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (all_qk) clip_proj_tile<P, CW, KD, false>(wf, a, acc);
        else if (all_v) clip_proj_tile<P, CW, KD, true>(wf, a, acc);
        else {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                if (which[j] < 2) {
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb) acc[j] = P::mma_w(wf[j][kb], a[kb], acc[j]);
                } else {
#pragma unroll
                    for (int kb = 0; kb < KD; ++kb) acc[j] = P::mma_a(a[kb], wf[j][kb], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            if (which[j] < 2) {                                   // D[dim = 4 lg + r][token = lr]: 4 consecutive dims of one token
                elem* dst = (elem*)(which[j] == 0 ? &qs[0][0] : &ks[0][0]) + qk_off<P>(rt * 16 + lr, d0[j] + 4 * lg, KDH);
                P::store4(dst, acc[j] + pb[j]);
            } else {                                              // D[token = 4 lg + r][dim = lr]: 4 consecutive tokens of one dim
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = acc[j][e] + pbs[j];
                vkeep[rt][j] = __builtin_convertvector(y, bf16x4v);      // (P::store4's rounding)
            }
        }
    }
    DSG_TL_MARK(2);      // this wave's projection columns done for all row tiles (W_qkv has landed)
    DSG_LDS_BARRIER();                                            // every wave is done with the rows: V^T moves in
    DSG_TL_MARK(3);
#pragma unroll
    for (int rt = 0; rt < NKT; ++rt)
#pragma unroll
        for (int j = 0; j < CW; ++j)
            if (which[j] == 2) *(bf16x4v*)((elem*)&vs[0][0] + vt_off<P>(d0[j] + lr, rt * 16 + 4 * lg, NVF)) = vkeep[rt][j];
    DSG_LDS_BARRIER();
    DSG_TL_MARK(4);      // Q / K / V^T of the head in LDS
    // ---- (2) attention of query tile `wave` (k_attn on LDS operands)
    const int qt = wave;
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
        s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(ks[nt * KDH + kb][lane], qs[qt * KDH + kb][lane], s[nt]);   // D[key = 4 lg + r][query = lr]
    }
    DSG_TL_MARK(5);      // Q K^T issued
    const float scale = 1.0f / sqrtf((float)HD);
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
    DSG_TL_MARK(6);      // softmax
    f32x4 pfr[NVF];
#pragma unroll
    for (int kb = 0; kb < NVF; ++kb) {
        if constexpr (P::E == 4) {
            pfr[kb] = s[kb];
        } else {
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
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
            P::store4_afrag((elem*)g.out, (size_t)qk_off<P>(b * g.ntok + q, h * HD + dt * 16 + 4 * lg, KD), y);      // the rounding point of the attention rows (bf16w2: hi + lo)
        }
    }
    DSG_TL_MARK(7);      // P V + stores issued
}

// ---------------------------------------------------------------------------------------------------------
// k_ffn (round 4, BLOCK set): the whole feed-forward half of an encoder layer for a block of RT x 16 rows in ONE kernel --
//   hidden = gelu(x1 . W1^T + b1)   (hidden stays in LDS, in the GEMM type: the rounding point of the `hidden` buffer)
//   pre2   = x1 + hidden . W2^T + b2
//   xn     = LayerNorm2(pre2)        -> fp32 rows (the next attention kernel's residual) + GEMM-type fragment-major rows (the next
//                                       QKV projection's / the pose head's operand: they become DIRECT GEMMs, no LayerNorm-on-read)
// replacing linear1, linear2 and the LayerNorm prologue of the next kernel: one dispatch instead of two, the largest activation of
// a layer never crosses the fabric, and the 12 - 18 column groups of the next GEMM stop re-normalising the same fp32 rows.  The
// price: every workgroup streams ALL of W1 and W2 (1 MB in bf16 at D = 256, ff = 1024) through one CU for 16 RT rows.
// Wave w owns hidden columns [w ff/NW, (w+1) ff/NW) in phase 1 and output columns [w D/NW, (w+1) D/NW) in phase 2; weight fragments
// are double-buffered two column tiles (phase 1) / KC k-blocks (phase 2) ahead.  At the ZEGGS widths the workgroup is 8 waves (244 VGPRs with
// the residual rows requested after phase 1, LATE_R): twice the weight bytes in flight per CU -- 18.1 -> 16.4 us per launch at 5696 rows
// (profiles/r04_u_*, r04_v_*; walking the hidden-column groups in a per-workgroup rotated order, in case every CU of an XCD asking
// the same L2 channel for the same fragment at the same moment was the limit, changed nothing: 16.35 vs 16.32).  The k sums of linear2 run 0 .. ff - 1 in one
// chain (k_gemm_blk_k: four ranges, then summed), LayerNorm2 as in k_attn_op (explicit fma sites).
// Reference arithmetic: linear1 / activation / linear2 / norm2 of torch's TransformerEncoderLayer (main/model/mdm.py:79-86).
// ---------------------------------------------------------------------------------------------------------
struct FfnArgs {
    const void* A;          // LayerNorm1 rows in the GEMM type, fragment-major (k_attn_op's X1a)
    const float* R;         // the same rows in fp32 (residual)
    const void* W1; const float* b1;
    const void* W2; const float* b2;
    const float* ln_g; const float* ln_b;
    float* Xn;              // LayerNorm2 rows fp32 [rows][D]
    void* Xa;               // ... in the GEMM type, fragment-major
    int M, MT;
    // OP (round 5): A = the ATTENTION rows (k_clip_attn), R = the rows the attention block adds back; out_proj + bias + residual + LayerNorm1 run
    // as the prologue, the LayerNorm1 rows go to LDS (linear1's operand) and stay in registers (linear2's residual): no X1 / X1a round trip
    const void* Wo; const float* bo;
    const float* ln1_g; const float* ln1_b;
    float* X1;              // OP on 64-row blocks (no LDS left to park them): the fp32 LayerNorm1 rows, written by the prologue and read back after phase 1
};

template <class P, int DT, int FT, int RT, int NW, int LA = 2, bool LATE_R = false, bool OP = false, int RING = 0, bool WO_RING = false>      // D = 64 DT, ff = 64 FT, RT row tiles per workgroup, NW waves, LA tiles / 2 LA k-blocks of look-ahead
__global__ __launch_bounds__(64 * NW) void k_ffn(const FfnArgs g) {      // RING > 0: the weights of both phases as ONE stream through a rolling ring of RING fragments (see below)
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = 64 * DT, FF = 64 * FT, KD = D / P::KB, KF = FF / P::KB;
    constexpr int FW = FF / 16 / NW, DW = D / 16 / NW;            // column tiles per wave in phase 1 / phase 2
    constexpr int HP = FF * ES + 16;                 // LDS pitch of a hidden row
    constexpr int KC0 = (DW >= 4 ? 2 : 4) * LA, KC = KC0 < KF ? KC0 : KF, NC = KF / KC;   // phase 2: k-blocks per weight chunk (8 LA fragments in flight)
    constexpr int NT = 64 * NW;
    static_assert(FW >= LA && FW % LA == 0 && DW >= 1 && KF % KC == 0 && (FF / 16) % NW == 0 && (D / 16) % NW == 0, "shape");
    static_assert(!P::W2 || (OP && RING > 0 && RT < 4), "bf16w2: the ring forms with the prologue");
    constexpr int HID_LO = RT * 16 * HP;             // bf16w2: `hidden` as a hi + lo pair of images (P::store4_a / P::aload)
    __shared__ __attribute__((aligned(16))) char hid[P::AF * RT * 16 * HP];
    __shared__ float red[RT][2][NW][16];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];      // b2, LayerNorm2 scale / shift
    constexpr int XP = D * ES + 16;                  // OP: LDS pitch of a LayerNorm1 row
    constexpr bool BIG = OP && RT >= 4;              // 64-row blocks: `hidden` fills the LDS -- the LayerNorm1 rows alias its head (dead before phase 1
                                                     // writes: linear1's operand sits in registers by then) and the fp32 rows go through g.X1
    constexpr bool X1MEM = BIG || (OP && RT == 2 && D >= 512);      // ... and (round 6) on 32-row blocks at latent_dim 512, where x1f (66 KB) does not fit next to `hidden` and the rows either: the fp32 rows through g.X1 only
    static_assert(!BIG || RT * 16 * XP <= RT * 16 * HP, "alias");
    constexpr int XA_LO = RT * 16 * XP;              // (bf16w2: ... and the LayerNorm1 rows)
    __shared__ __attribute__((aligned(16))) char xa_own[(OP && !BIG) ? P::AF * RT * 16 * XP : 16];
    char* const xa = BIG ? hid : xa_own;
    __shared__ __attribute__((aligned(16))) float vecs1[OP ? 3 : 1][OP ? D : 4];      // OP: b_o, LayerNorm1 scale / shift
    constexpr int X1P = D + 4;                       // OP: pitch (floats) of the fp32 LayerNorm1 rows parked in LDS across phase 1 (register budget)
    __shared__ __attribute__((aligned(16))) float x1f[(OP && !X1MEM) ? RT * 16 * X1P : 4];
    preload_kernargs(g);
    const int mb = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int m0 = mb * 16 * RT, mt_last = g.MT - 1;
    // ---- loads that do not depend on phase 1: A fragments, first W1 tiles, residual rows, the per-column vectors
    constexpr int RH = (OP && RT >= 4) ? 2 : RT;     // OP on 64-row blocks: the prologue takes the row tiles two at a time (register budget)
    typename P::afrag af[RT][KD];
    // OP: the RH x KD attention-row fragments of a half are fetched ONCE per workgroup -- NFW of them per wave -- and handed round through LDS (the tail of
    // `hid`, free until phase 1): every wave needs all of them, and eight waves loading the same 16 KB from uncached memory were 23 MB of memory-side reads
    // per launch at 64 clips for 2.9 MB of rows
    constexpr int NFH = RH * KD, NFW = (NFH + NW - 1) / NW;
    f32x4* const stg = (f32x4*)(hid + P::AF * (RT * 16 * HP - NFH * 1024));      // [fragment][image][lane]
    static_assert(!OP || NFH * 1024 + (RT >= 4 ? RT * 16 * (D * ES + 16) : 0) <= RT * 16 * HP, "the staged fragments sit behind the (aliased) LayerNorm1 rows");
    typename P::afrag afw[OP ? NFW : 1];
    auto load_a = [&](int rt0) {
        if constexpr (OP) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) {
                const int f = min(wave * NFW + i, NFH - 1), rt = rt0 + f / KD, kb = f % KD;
                const int mt = min(mb * RT + rt, mt_last);   // clamped (rows past the end are computed and dropped)
                afw[i] = P::aload_frag(g.A, (size_t)(mt * KD + kb), lane);
            }
        } else {
#pragma unroll
            for (int rt = rt0; rt < rt0 + RH; ++rt) {
                const int mt = min(mb * RT + rt, mt_last);   // clamped (rows past the end are computed and dropped)
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) af[rt][kb] = P::aload_frag(g.A, (size_t)(mt * KD + kb), lane);
            }
        }
    };
    // (OP) afw -> LDS -> every wave's af[rt0 .. rt0 + RH)
    auto share_a = [&](int rt0) {
#pragma unroll
        for (int i = 0; i < NFW; ++i) if (wave * NFW + i < NFH) {
            if constexpr (P::W2) { stg[((wave * NFW + i) * 2) * 64 + lane] = afw[i].h; stg[((wave * NFW + i) * 2 + 1) * 64 + lane] = afw[i].l; }
            else stg[(wave * NFW + i) * 64 + lane] = afw[i];
        }
        DSG_LDS_BARRIER();
#pragma unroll
        for (int rt = rt0; rt < rt0 + RH; ++rt)
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) af[rt][kb] = P::aload((const char*)&stg[((rt - rt0) * KD + kb) * P::AF * 64 + lane], 1024);
    };
    load_a(0);
    const f32x4* w1 = (const f32x4*)g.W1 + lane;
    const f32x4* w2 = (const f32x4*)g.W2 + lane;
    f32x4 wb1[2][LA][KD], pb1[2][LA];                // [buffer][tile of the group][k-block]
    auto load1 = [&](int buf, int tp) {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int nt = wave * FW + LA * tp + j;
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) wb1[buf][j][kb] = w1[((size_t)nt * KD + kb) * 64];
            pb1[buf][j] = *(const f32x4*)(g.b1 + nt * 16 + 4 * lg);
        }
    };
    // RING (round 5): a wave's weight fragments of phase 1 (FW tiles x KD) and phase 2 (KF k-blocks x DW tiles) are ONE sequence; fragment i lives in
    // slot i % RING and slot s is refilled with fragment i + RING right after the MFMAs that read fragment i are issued.  The double-buffered groups of
    // the plain form ask for a group, compute the previous one, then WAIT for the group with nothing else in flight: a launch is (groups) x (latency +
    // transfer); the ring has RING - 1 .. RING fragments of every wave in flight at every moment, across tiles, chunks and the phase switch, in the same
    // registers.  Same MFMAs on the same operands, every accumulation chain in the same k order: bit-identical.
    // Measured (profiles/r05_s_*, one box each): 32-row form, 32 slots: 19.25 -> 18.25 us per launch at 64 clips, 1 x 64 clips 282.7 -> 276.0 us per step, 4 x 16 clips
    // 17.96 k -> 18.98 k frames/s, 4 x 32: 23.86 k -> 24.30 k; 64-row form, 12 slots (16: 32 B of scratch): 4 x 64 clips 26.2-26.3 k -> 26.6-26.9 k.  A whole chunk of
    // look-ahead instead (hidden in 128-column chunks, the A operand in LDS: 33.0 vs 30.5 us per 64-row launch) and 4 waves of 512 registers (36.4) were slower:
    // the kernel is not waiting for bytes in flight alone -- what the ring removes is the drain at every group boundary.
    // WO_RING (round 6, latent_dim 512): W_o joins the stream as its FIRST N0 fragments (k-block major: the order out_proj consumes them) instead of waiting whole
    // in registers -- DW x KD = 64 two-... fragments per wave do not fit next to the attention rows at that width.  The ring is then filled at kernel start and
    // stays full across LayerNorm1 (no quarter fills).
    constexpr int N0 = WO_RING ? DW * KD : 0;
    constexpr int N1 = FW * KD, N2 = KF * DW, NRING = RING > 0 ? RING : 1;
    static_assert(RING == 0 || (OP && RING <= N1), "ring");
    static_assert(!WO_RING || (RING > 0 && RING <= N0 && RT <= 2), "W_o in the ring: the 16- and 32-row forms");
    typename P::wfrag ring[NRING];
    const f32x4* wo_ring = (const f32x4*)g.Wo + lane;
    auto ring_load = [&](int ig) -> typename P::wfrag {      // ig: index in the whole stream [W_o | W1 | W2]
        if (ig < N0) return P::wload(wo_ring, (size_t)(wave * DW + ig % DW) * KD + ig / DW);
        const int i = ig - N0;
        if (i < N1) return P::wload(w1, (size_t)(wave * FW + i / KD) * KD + i % KD);
        const int k = (i - N1) / DW, t = (i - N1) % DW;
        return P::wload(w2, (size_t)(wave * DW + t) * KF + k);
    };
    if constexpr (!OP) load1(0, 0);
    f32x4 pr[RT][DW];
    // (row index of this lane in row tile rt, clamped to a valid row; recomputed where it is used: four 64-bit offsets kept live across
    //  the kernel were what spilled in the 64-row form with the prologue)
    auto rowok_of = [&](int rt) { return m0 + rt * 16 + lr < g.M; };
    auto mrow_of = [&](int rt) { return (unsigned)min(m0 + rt * 16 + lr, g.M - 1); };      // (32-bit: < 2^22 rows; byte offsets below stay 32-bit too)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if ((!LATE_R || OP) && rt < RH) {      // (OP: the rows the ATTENTION block adds back; linear2's residual is computed below)
#pragma unroll
            for (int t = 0; t < DW; ++t) pr[rt][t] = lda16<P>(g.R, (size_t)((mrow_of(rt) * D + (wave * DW + t) * 16 + 4 * lg) * 4u));
        }
    }
    constexpr int NV = (3 * D / 4 + NT - 1) / NT;
    f32x4 vload[NV], vload1[OP ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = min(tid + NT * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);
        vload[i] = ((const f32x4*)(vsel == 0 ? g.b2 : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
        if constexpr (OP) vload1[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln1_g : g.ln1_b)))[vidx];
    }
    if constexpr (OP) {
        // ---- prologue: pre1 = attention rows . W_o^T + b_o + residual; x1 = LayerNorm1(pre1) -> LDS in the GEMM type (linear1's operand)
        //      and, in fp32, this lane's registers: phase 2 adds exactly these (row, column) values back (same wave -> column map)
        const f32x4* wo = (const f32x4*)g.Wo + lane;
        typename P::wfrag wof[WO_RING ? 1 : DW][WO_RING ? 1 : KD];
        if constexpr (WO_RING) {
#pragma unroll
            for (int i = 0; i < RING; ++i) ring[i] = ring_load(i);
        } else {
#pragma unroll
            for (int t = 0; t < DW; ++t)
#pragma unroll
                for (int kb = 0; kb < KD; ++kb) wof[t][kb] = P::wload(wo, (size_t)(wave * DW + t) * KD + kb);
        }
        // (round 6, measured no: on 16-row tiles the register file has room for half of the weight ring next to W_o, but requesting it HERE puts
        //  W1 ahead of the attention rows and W_o in the CU's load path and delays out_proj: 1 x 16 clips 192.6 -> 195.4 us per step with 16 slots
        //  early, 198.2 with all 32 -- profiles/r06_z_*)
        DSG_LOADS_ISSUED();
        DSG_TL_MARK(0);      // attention rows (this wave's share), residual rows, W_o requested
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + NT * i;
            if (e < 3 * D / 4) { *(f32x4*)(&vecs1[0][0] + e * 4) = vload1[i]; *(f32x4*)(&vecs[0][0] + e * 4) = vload[i]; }      // (b2 / LayerNorm2 too: no registers across the phases)
        }
        f32x4 acc1[RT][DW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < DW; ++t) acc1[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r0 = 0; r0 < RT; r0 += RH) {
            if (r0 > 0) DSG_LDS_BARRIER();                    // every wave has read the previous half out of the stage
            share_a(r0);
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                if constexpr (WO_RING) {      // (RH == RT <= 2: a slot is consumed by the row tiles side by side; what follows W_o in the stream is W1)
#pragma unroll
                    for (int t = 0; t < DW; ++t) {
                        const int i = kb * DW + t;
#pragma unroll
                        for (int rt = r0; rt < r0 + RH; ++rt) acc1[rt][t] = P::mma_w(ring[i % NRING], af[rt][kb], acc1[rt][t]);
                        ring[i % NRING] = ring_load(i + RING);
                    }
                } else {
#pragma unroll
                    for (int rt = r0; rt < r0 + RH; ++rt)
#pragma unroll
                        for (int t = 0; t < DW; ++t) acc1[rt][t] = P::mma_w(wof[t][kb], af[rt][kb], acc1[rt][t]);      // D[n 4lg+r][row lr]
                }
            }
            if (r0 + RH < RT) {                               // (64-row blocks: the next two row tiles' attention rows and residual)
                load_a(r0 + RH);
#pragma unroll
                for (int rt = r0 + RH; rt < r0 + 2 * RH; ++rt)
#pragma unroll
                    for (int t = 0; t < DW; ++t) pr[rt][t] = lda16<P>(g.R, (size_t)((mrow_of(rt) * D + (wave * DW + t) * 16 + 4 * lg) * 4u));
                DSG_LOADS_ISSUED();
            }
        }
        DSG_TL_MARK(1);      // out_proj issued (attention rows shared through LDS, W_o landed)
        if constexpr (WO_RING) {
            // (the ring has been streaming since kernel start)
        } else if constexpr (RING > 0) {
#pragma unroll
            // Round 6: the ring's first fill goes out in FOUR quarters, one here and one behind each of LayerNorm1's next three stages.  A wave issues in
            // order: the 32 loads of a whole fill keep it in the issue stage while the CU's load path takes them (8 waves x 32 KB at 64 B / clk = 1.7 us) --
            // LayerNorm1 then ran AFTER the fill was issued, with the load path idle behind it (marks: 2.9 us for a 1.2 us LayerNorm).  Spread, the same
            // loads are issued while the other waves compute: 1 x 16 clips 192.6 -> 188.0 us per step (ROWS), 4 x 8: 207.3 -> 201.4, 1 x 64: 259.8 -> 255.2
            // (STREAM), bit-identical -- profiles/r06_ca_ab_dev{A,B}.log.  (k_attn_mid spreads its W_o / W1 loads over the softmax the same way.)
            for (int i = 0; i < RING / 4; ++i) ring[i] = ring_load(i);
        } else {
            load1(0, 0);                                      // (W_o and the attention rows are dead: the first W1 tiles arrive behind LayerNorm1)
        }
        DSG_LOADS_ISSUED();
        // (vecs1 was written before share_a's barrier: it is in place.  Round 6: LayerNorm1 / LayerNorm2 statistics in ONE exchange -- ln_wave_moments --
        //  and no barrier of its own for vecs1: two barriers less per LayerNorm1, one per LayerNorm2: 1 x 16 clips 186.2 -> 184.4 us per step,
        //  1 x 64: 252.9 -> 249.5, profiles/r06_cf_ab_dev{A,B}.log)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float sm = 0.f;
#pragma unroll
            for (int t = 0; t < DW; ++t) {
                const f32x4 pb = *(const f32x4*)(&vecs1[0][(wave * DW + t) * 16 + 4 * lg]);
                acc1[rt][t] = acc1[rt][t] + pb + pr[rt][t];
                sm += (acc1[rt][t][0] + acc1[rt][t][1]) + (acc1[rt][t][2] + acc1[rt][t][3]);
            }
            ln_wave_moments<DW>(acc1[rt], sm, red[rt][0][wave], red[rt][1][wave], lr, lg);
        }
        if constexpr (RING > 0 && !WO_RING) {
#pragma unroll
            for (int i = RING / 4; i < RING / 2; ++i) ring[i] = ring_load(i);
            DSG_LOADS_ISSUED();
        }
        DSG_LDS_BARRIER();
        float mean1[RT];
        float var1[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ln_combine_moments<NW, D>(red[rt][0], red[rt][1], lr, mean1[rt], var1[rt]);
        if constexpr (RING > 0 && !WO_RING) {
#pragma unroll
            for (int i = RING / 2; i < 3 * (RING / 4); ++i) ring[i] = ring_load(i);
            DSG_LOADS_ISSUED();
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float rstd = 1.0f / sqrtf(var1[rt] + 1e-5f);
#pragma unroll
            for (int t = 0; t < DW; ++t) {
                const int n = (wave * DW + t) * 16 + 4 * lg;
                const f32x4 pg = *(const f32x4*)(&vecs1[1][n]), pbt = *(const f32x4*)(&vecs1[2][n]);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc1[rt][t][e] - mean1[rt]) * rstd, pg[e], pbt[e]);
                P::store4_a((elem*)(xa + (rt * 16 + lr) * XP) + n, XA_LO, y);
                // linear2's residual: read back by the same lane after phase 2
                // (64-row blocks: through g.X1, un-clamped and unconditional -- the row buffers are padded past the last 64-row block)
                if constexpr (X1MEM) *(f32x4*)((char*)g.X1 + (size_t)(((unsigned)(m0 + rt * 16 + lr) * D + n) * 4u)) = y;
                else *(f32x4*)(x1f + (rt * 16 + lr) * X1P + n) = y;
            }
        }
        if constexpr (RING > 0 && !WO_RING) {
#pragma unroll
            for (int i = 3 * (RING / 4); i < RING; ++i) ring[i] = ring_load(i);
            DSG_LOADS_ISSUED();
        }
        DSG_LDS_BARRIER();                                    // (also: red is free for LayerNorm2)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) af[rt][kb] = P::aload(xa + (rt * 16 + lr) * XP + (kb * P::KB + P::E * lg) * ES, XA_LO);
        if constexpr (BIG) DSG_LDS_BARRIER();                 // every wave holds its operand: phase 1 may overwrite the aliased rows
        DSG_TL_MARK(2);      // LayerNorm1 (three barriers) -> linear1's operand registers
    } else {
        DSG_LOADS_ISSUED();
    }
    f32x4 acc[RT][DW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < DW; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (RING > 0) {
        static_assert(OP, "ring: the forms with the prologue");
        // ---- phase 1 on the ring: tile j = fragments j KD .. j KD + KD - 1; the row tiles' chains advance side by side (independent MFMAs back to back)
        f32x4 pbr[2];
        pbr[0] = *(const f32x4*)(g.b1 + (wave * FW) * 16 + 4 * lg);
#pragma unroll
        for (int j = 0; j < FW; ++j) {
            const int nt = wave * FW + j;
            if (j + 1 < FW) pbr[(j + 1) & 1] = *(const f32x4*)(g.b1 + (nt + 1) * 16 + 4 * lg);
            f32x4 c[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) c[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const int i = N0 + j * KD + kb;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) c[rt] = P::mma_w(ring[i % NRING], af[rt][kb], c[rt]);      // D[n 4lg+r][row lr]
                if (i + RING < N0 + N1 + N2) ring[i % NRING] = ring_load(i + RING);
                DSG_LOADS_ISSUED();
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = gelu_erf<P>(c[rt][e] + pbr[j & 1][e]);
                P::store4_a((elem*)(hid + (rt * 16 + lr) * HP) + nt * 16 + 4 * lg, HID_LO, y);
            }
        }
        if constexpr (X1MEM) {               // (64-row blocks: the fp32 LayerNorm1 rows come back from the padded X1 rows, as in the plain form)
            int lr_late = lr;
#ifndef DSG_EMU
            asm volatile("" : "+v"(lr_late));
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < DW; ++t) pr[rt][t] = lda16<P>((const float*)g.X1, (size_t)(((unsigned)(m0 + rt * 16 + lr_late) * D + (wave * DW + t) * 16 + 4 * lg) * 4u));
        }
        DSG_TL_MARK(3);      // phase 1 on the ring: linear1 + GELU -> LDS
        DSG_LDS_BARRIER();                 // `hidden` is complete (the ring keeps streaming: the barrier does not wait for vector memory)
        DSG_TL_MARK(4);
        // ---- phase 2 on the ring: k-block k = fragments N1 + k DW .. + DW - 1
        constexpr int AB = RT <= 2 ? 2 : 1;      // (32-row blocks: the next k-block's `hidden` fragments are read from LDS one step ahead)
        typename P::afrag a[AB][RT];
        if constexpr (AB == 2) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[0][rt] = P::aload(hid + (rt * 16 + lr) * HP + (P::E * lg) * ES, HID_LO);
        }
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            if (AB == 1 || k + 1 < KF) {
                const int kn = AB == 2 ? k + 1 : k;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a[kn % AB][rt] = P::aload(hid + (rt * 16 + lr) * HP + (kn * P::KB + P::E * lg) * ES, HID_LO);
            }
#pragma unroll
            for (int t = 0; t < DW; ++t) {
                const int i = N0 + N1 + k * DW + t;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][t] = P::mma_w(ring[i % NRING], a[k % AB][rt], acc[rt][t]);
                if (i + RING < N0 + N1 + N2) ring[i % NRING] = ring_load(i + RING);
            }
            DSG_LOADS_ISSUED();
        }
    } else {
        // ---- phase 1: hidden tiles of this wave, LA at a time, the next group's fragments in flight
    #pragma unroll
        for (int tp = 0; tp < FW / LA; ++tp) {
            if (tp + 1 < FW / LA) { load1((tp + 1) & 1, tp + 1); DSG_LOADS_ISSUED(); }
    #pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int nt = wave * FW + LA * tp + j;
    #pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int kb = 0; kb < KD; ++kb) c = P::mma(wb1[tp & 1][j][kb], af[rt][kb], c);      // D[n 4lg+r][row lr]
                    f32x4 y;
    #pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = gelu_erf<P>(c[e] + pb1[tp & 1][j][e]);
                    P::store4((elem*)(hid + (rt * 16 + lr) * HP) + nt * 16 + 4 * lg, y);
                }
            }
        }
        // ---- phase 2: first W2 chunk requested before the barrier
        f32x4 wb2[2][KC][DW];
        auto load2 = [&](int buf, int c) {
    #pragma unroll
            for (int k = 0; k < KC; ++k)
    #pragma unroll
                for (int t = 0; t < DW; ++t) wb2[buf][k][t] = w2[((size_t)(wave * DW + t) * KF + c * KC + k) * 64];
        };
        load2(0, 0);
        if constexpr ((LATE_R && !OP) || X1MEM) {      // (8 waves: the residual rows are requested only now -- the register budget of phase 1)
            int lr_late = lr;
            if constexpr (X1MEM) {
    #ifndef DSG_EMU
                asm volatile("" : "+v"(lr_late));      // opaque: the row offsets are RE-computed here -- kept live across phase 1 they were what spilled
    #endif
            }
    #pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const unsigned mr = X1MEM ? (unsigned)(m0 + rt * 16 + lr_late) : (unsigned)min(m0 + rt * 16 + lr_late, g.M - 1);
    #pragma unroll
                for (int t = 0; t < DW; ++t) pr[rt][t] = lda16<P>(X1MEM ? (const float*)g.X1 : g.R, (size_t)((mr * D + (wave * DW + t) * 16 + 4 * lg) * 4u));
            }
        }
        DSG_LOADS_ISSUED();
        if constexpr (!OP) {
    #pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = tid + NT * i;
                if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];
            }
        }
        DSG_LDS_BARRIER();
    #pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) { load2((c + 1) & 1, c + 1); DSG_LOADS_ISSUED(); }
    #pragma unroll
            for (int k = 0; k < KC; ++k) {
    #pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f32x4 a = *(const f32x4*)(hid + (rt * 16 + lr) * HP + ((c * KC + k) * P::KB + P::E * lg) * ES);
    #pragma unroll
                    for (int t = 0; t < DW; ++t) acc[rt][t] = P::mma(wb2[c & 1][k][t], a, acc[rt][t]);
                }
            }
        }
    }
    DSG_TL_MARK(5);          // phase 2: linear2 issued
    // ---- + bias + residual, LayerNorm2 over whole rows (row lr: 4 lane groups x NW waves hold its D values)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < DW; ++t) {
            const f32x4 pb = *(const f32x4*)(&vecs[0][(wave * DW + t) * 16 + 4 * lg]);
            if constexpr (OP && !X1MEM) pr[rt][t] = *(const f32x4*)(x1f + (rt * 16 + lr) * X1P + (wave * DW + t) * 16 + 4 * lg);
            acc[rt][t] = acc[rt][t] + pb + pr[rt][t];
            sm += (acc[rt][t][0] + acc[rt][t][1]) + (acc[rt][t][2] + acc[rt][t][3]);
        }
        ln_wave_moments<DW>(acc[rt], sm, red[rt][0][wave], red[rt][1][wave], lr, lg);
    }
    DSG_LDS_BARRIER();
    float mean[RT];
    float var2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ln_combine_moments<NW, D>(red[rt][0], red[rt][1], lr, mean[rt], var2[rt]);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const float rstd = 1.0f / sqrtf(var2[rt] + 1e-5f);
        if (rowok_of(rt)) {
#pragma unroll
            for (int t = 0; t < DW; ++t) {
                const int n = (wave * DW + t) * 16 + 4 * lg;
                const f32x4 pg = *(const f32x4*)(&vecs[1][n]), pbt = *(const f32x4*)(&vecs[2][n]);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[rt][t][e] - mean[rt]) * rstd, pg[e], pbt[e]);
                *(f32x4*)((char*)g.Xn + (size_t)((mrow_of(rt) * D + n) * 4u)) = y;
                P::store4_afrag((elem*)g.Xa, (size_t)qk_off<P>((int)mrow_of(rt), n, KD), y);
            }
        }
    }
    DSG_TL_MARK(6);          // LayerNorm2 (two barriers) + row stores issued
}

// ---------------------------------------------------------------------------------------------------------
// k_ffn_part + k_ffn_ln (round 4, BLOCK set below the STREAM threshold): k_ffn split S ways over the hidden dimension.
// k_ffn pulls all of W1 and W2 (1 MB) through one CU per 32 rows -- 17 us whatever the batch, which loses to linear1 + linear2 at
// 1424 rows (45 workgroups).  Here workgroup (row block mb, split s) computes  hidden[:, s ff/S .. (s+1) ff/S) = gelu(x1 W1^T + b1)
// for RT x 16 rows into LDS and multiplies it with the matching k-range of W2: a PARTIAL linear2 result, written as an fp32 slab
// part[s][row][D].  Its weights (ff/S columns of W1, the same k-range of W2: 128 KB at S = 8) are requested in ONE batch at kernel
// start, next to the A fragments -- one memory round trip, then 2 x FWS x RT x KD MFMAs per wave.  k_ffn_ln sums the S slabs in the
// fixed order 0 .. S-1, adds bias + residual, applies LayerNorm2 (two-pass, fp32) and writes what k_ffn writes: the fp32 rows and
// the GEMM-type fragment-major rows, so the next QKV projection is a DIRECT GEMM.  A row's result depends on S (a template constant
// of the set), never on the batch.
// Reference arithmetic: linear1 / activation / linear2 / norm2 of torch's TransformerEncoderLayer (main/model/mdm.py:79-86).
// ---------------------------------------------------------------------------------------------------------
struct FfnPartArgs {
    const void* A;          // LayerNorm1 rows in the GEMM type, fragment-major (k_attn_op's X1a); OP: the ATTENTION rows (k_clip_attn's output)
    const void* W1; const float* b1;
    const void* W2;
    float* part;            // [S][slab] fp32, slab >= M * D
    size_t slab;            // floats per slab
    int M, MT;
    // OP (round 5): out_proj + bias + residual + LayerNorm1 of the row block as the kernel's prologue (every split recomputes it: 128 KB of
    // W_o and 2 x DW x KD MFMAs per wave against a dispatch + a round trip of the rows); split 0 writes the fp32 rows k_ffn_ln adds back
    const float* R;         // residual rows fp32 [rows][D]
    const void* Wo; const float* bo;
    const float* ln_g; const float* ln_b;
    float* X1;              // LayerNorm1 rows fp32 [rows][D]
};

template <class P, int DT, int FT, int RT, int NW, int S, bool OP = false>
__global__ __launch_bounds__(64 * NW) void k_ffn_part(const FfnPartArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = 64 * DT, FF = 64 * FT, KD = D / P::KB, KF = FF / P::KB;
    constexpr int FS = FF / S, FWS = FS / 16 / NW, KFS = FS / P::KB, DW = D / 16 / NW;
    constexpr int HP = FS * ES + 16;                 // LDS pitch of a hidden row
    static_assert(FF % S == 0 && FS % (16 * NW) == 0 && FS % P::KB == 0 && (D / 16) % NW == 0 && FWS >= 1 && KFS >= 1, "shape");
    __shared__ __attribute__((aligned(16))) char hid[RT * 16 * HP];
    constexpr int XP = D * ES + 16;                  // OP: LDS pitch of a LayerNorm1 row
    __shared__ __attribute__((aligned(16))) char xa[OP ? RT * 16 * XP : 16];
    __shared__ float red[OP ? 2 * RT * NW * 16 : 1];
    __shared__ __attribute__((aligned(16))) float vecs1[OP ? 2 : 1][OP ? D : 4];      // OP: LayerNorm1 scale / shift
    preload_kernargs(g);
    const int mb = blockIdx.x / S, s = blockIdx.x - mb * S;
    const int lane = threadIdx.x & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int m0 = mb * 16 * RT, mt_last = g.MT - 1;
    // ---- every global load of the workgroup in one batch (OP: the prologue's operands and W1 first, W2 once W_o is dead)
    f32x4 af[RT][KD];
    // (every wave loads all RT x KD fragments itself: handing them round through LDS, as k_ffn<OP> does, measured 1.5 % SLOWER here -- 16 clips 202.3 vs 199.4 us)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int mt = min(mb * RT + rt, mt_last);           // clamped (rows past the end are computed and dropped)
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) af[rt][kb] = lda16<P>(g.A, ((size_t)(mt * KD + kb) * 64 + lane) * P::E * ES);
    }
    const f32x4* w1 = (const f32x4*)g.W1 + lane;
    const f32x4* w2 = (const f32x4*)g.W2 + lane;
    f32x4 wb1[FWS][KD], pb1[FWS];
    f32x4 wb2[KFS][DW];
    auto load_w1 = [&]() {
#pragma unroll
        for (int j = 0; j < FWS; ++j) {
            const int nt = s * (FS / 16) + wave * FWS + j;
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) wb1[j][kb] = w1[((size_t)nt * KD + kb) * 64];
            pb1[j] = *(const f32x4*)(g.b1 + nt * 16 + 4 * lg);
        }
    };
    auto load_w2 = [&]() {
#pragma unroll
        for (int k = 0; k < KFS; ++k)
#pragma unroll
            for (int t = 0; t < DW; ++t) wb2[k][t] = w2[((size_t)(wave * DW + t) * KF + s * KFS + k) * 64];
    };
    if constexpr (OP) {
        // ---- prologue: pre1 = attention rows . W_o^T + b_o + residual; x1 = LayerNorm1(pre1) -> LDS (GEMM type), fp32 rows by split 0.
        //      wave w owns output columns [w D/NW, (w+1) D/NW) of both row tiles (D[n = 4 lg + r][row = lr])
        const f32x4* wo = (const f32x4*)g.Wo + lane;
        f32x4 wof[DW][KD], pr[RT][DW], pbo[DW], pg[DW], pbt[DW];
        // LayerNorm1 scale / shift: ONE 16-byte load per lane for the workgroup, requested up front and parked in LDS until the rows are normalised
        // (per-wave copies requested up front cost 32 registers the stamps build did not have; requested after the MFMAs their latency showed)
        constexpr int NV1 = (2 * D / 4 + 64 * NW - 1) / (64 * NW);
        f32x4 vl1[NV1];
#pragma unroll
        for (int i = 0; i < NV1; ++i) {
            const int e = min((int)threadIdx.x + 64 * NW * i, 2 * D / 4 - 1);
            vl1[i] = ((const f32x4*)(e < D / 4 ? g.ln_g : g.ln_b))[e < D / 4 ? e : e - D / 4];
        }
#pragma unroll
        for (int t = 0; t < DW; ++t) {
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) wof[t][kb] = wo[((size_t)(wave * DW + t) * KD + kb) * 64];
            const int n = (wave * DW + t) * 16 + 4 * lg;
            pbo[t] = *(const f32x4*)(g.bo + n);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int m = min(m0 + rt * 16 + lr, g.M - 1);
                pr[rt][t] = lda16<P>(g.R, ((size_t)m * D + n) * sizeof(float));
            }
        }
        load_w1();
        DSG_LOADS_ISSUED();
        DSG_TL_MARK(0);      // attention rows, W_o, residual rows, W1 columns requested
        f32x4 acc[RT][DW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < DW; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KD; ++kb)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int t = 0; t < DW; ++t) acc[rt][t] = P::mma(wof[t][kb], af[rt][kb], acc[rt][t]);
        DSG_TL_MARK(1);      // out_proj issued (attention rows + W_o have landed)
        // (round 6, measured no: requesting W1 only HERE -- out_proj then starts after 56 instead of 98 loads per wave -- moves the phases (mark 0
        //  at 2.2 instead of 3.8 us) and not the kernel: the workgroup's 480 KB through the CU's load path are what it waits for, in any order;
        //  profiles/r06_c_*, r06_dA / r06_dB_*)
        load_w2();                                            // (W_o is dead: its registers take W2's k-range)
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int i = 0; i < NV1; ++i) {
            const int e = (int)threadIdx.x + 64 * NW * i;
            if (e < 2 * D / 4) *(f32x4*)(&vecs1[0][0] + e * 4) = vl1[i];      // visible after the two LayerNorm barriers below
        }
        float sm[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            sm[rt] = 0.f;
#pragma unroll
            for (int t = 0; t < DW; ++t) { acc[rt][t] = acc[rt][t] + pbo[t] + pr[rt][t]; sm[rt] += (acc[rt][t][0] + acc[rt][t][1]) + (acc[rt][t][2] + acc[rt][t][3]); }
            sm[rt] += __shfl_xor(sm[rt], 16); sm[rt] += __shfl_xor(sm[rt], 32);
            if (lg == 0) red[(rt * NW + wave) * 16 + lr] = sm[rt];
        }
        DSG_TL_MARK(2);      // bias + residual added (the residual rows have landed)
        DSG_LDS_BARRIER();
        float mean[RT], qv[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[(rt * NW + w) * 16 + lr];
            mean[rt] = a / (float)D;
            qv[rt] = 0.f;
#pragma unroll
            for (int t = 0; t < DW; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc[rt][t][e] - mean[rt]; qv[rt] = __builtin_fmaf(d, d, qv[rt]); }
            qv[rt] += __shfl_xor(qv[rt], 16); qv[rt] += __shfl_xor(qv[rt], 32);
            if (lg == 0) red[((RT + rt) * NW + wave) * 16 + lr] = qv[rt];
        }
        DSG_LDS_BARRIER();
        DSG_TL_MARK(3);      // LayerNorm1 statistics (two barriers)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[((RT + rt) * NW + w) * 16 + lr];
            const float rstd = 1.0f / sqrtf(a / (float)D + 1e-5f);
            const int m = m0 + rt * 16 + lr;
#pragma unroll
            for (int t = 0; t < DW; ++t) {
                const int n = (wave * DW + t) * 16 + 4 * lg;
                pg[t] = *(const f32x4*)(&vecs1[0][n]); pbt[t] = *(const f32x4*)(&vecs1[1][n]);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[rt][t][e] - mean[rt]) * rstd, pg[t][e], pbt[t][e]);
                P::store4((elem*)(xa + (rt * 16 + lr) * XP) + n, y);
                if (s == 0 && m < g.M) *(f32x4*)(g.X1 + (size_t)m * D + n) = y;
            }
        }
        DSG_LDS_BARRIER();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) af[rt][kb] = *(const f32x4*)(xa + (rt * 16 + lr) * XP + (kb * P::KB + P::E * lg) * ES);
        DSG_TL_MARK(4);      // LayerNorm1 rows through LDS into linear1's operand registers
    } else {
        load_w1();
        load_w2();
        DSG_LOADS_ISSUED();
        DSG_TL_MARK(0);
    }
    // ---- phase 1: this wave's hidden tiles of the split
#pragma unroll
    for (int j = 0; j < FWS; ++j) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) c = P::mma(wb1[j][kb], af[rt][kb], c);      // D[n 4lg+r][row lr]
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = gelu_erf<P>(c[e] + pb1[j][e]);
            P::store4((elem*)(hid + (rt * 16 + lr) * HP) + (wave * FWS + j) * 16 + 4 * lg, y);
        }
    }
    DSG_TL_MARK(5);      // phase 1: linear1 slice + GELU -> LDS (W1 has landed)
    DSG_LDS_BARRIER();
    DSG_TL_MARK(6);
    // ---- phase 2: partial linear2 over the split's k-range, all D columns (wave w: columns [w D/NW, (w+1) D/NW))
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        f32x4 acc[DW];
#pragma unroll
        for (int t = 0; t < DW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KFS; ++k) {
            const f32x4 a = *(const f32x4*)(hid + (rt * 16 + lr) * HP + (k * P::KB + P::E * lg) * ES);
#pragma unroll
            for (int t = 0; t < DW; ++t) acc[t] = P::mma(wb2[k][t], a, acc[t]);
        }
        const int m = m0 + rt * 16 + lr;
        if (m < g.M) {
            float* o = g.part + (size_t)s * g.slab + (size_t)m * D + wave * DW * 16 + 4 * lg;
#pragma unroll
            for (int t = 0; t < DW; ++t) *(f32x4*)(o + t * 16) = acc[t];
        }
    }
    DSG_TL_MARK(7);      // phase 2 + slab stores issued (W2 has landed)
}

struct FfnLnArgs {
    const float* part; size_t slab;
    const float* R;         // LayerNorm1 rows fp32 (residual)
    const float* b2; const float* ln_g; const float* ln_b;
    float* Xn;              // LayerNorm2 rows fp32 [rows][D]
    void* Xa;               // ... in the GEMM type, fragment-major
    int M;
};

// RW rows per workgroup, 16 lanes per row; lane c of a row owns columns 64 q + 4 c .. + 3 (q < DT): 256 contiguous bytes per (row, q)
template <class P, int DT, int S, int RW = 16>
__global__ __launch_bounds__(16 * RW) void k_ffn_ln(const FfnLnArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int D = 64 * DT, KD = D / P::KB;
    preload_kernargs(g);
    const int tid = threadIdx.x, c = tid & 15;
    const int m = blockIdx.x * RW + (tid >> 4);
    const bool ok = m < g.M;
    const size_t mr = (size_t)(ok ? m : g.M - 1);
    f32x4 p[S][DT], r[DT], pb[DT], pg[DT], pt[DT];
#pragma unroll
    for (int q = 0; q < DT; ++q) {
        const int n = 64 * q + 4 * c;
#pragma unroll
        for (int s = 0; s < S; ++s) p[s][q] = lda16<P>(g.part, ((size_t)s * g.slab + mr * D + n) * sizeof(float));
        r[q] = lda16<P>(g.R, (mr * D + n) * sizeof(float));
        pb[q] = *(const f32x4*)(g.b2 + n); pg[q] = *(const f32x4*)(g.ln_g + n); pt[q] = *(const f32x4*)(g.ln_b + n);
    }
    DSG_LOADS_ISSUED();
    DSG_TL_MARK(0);
    f32x4 v[DT];
    float sm = 0.f;
#pragma unroll
    for (int q = 0; q < DT; ++q) {
        f32x4 a = p[0][q];
#pragma unroll
        for (int s = 1; s < S; ++s) a = a + p[s][q];
        v[q] = a + pb[q] + r[q];
        sm += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
    const float mean = sm / (float)D;
    float qv = 0.f;
#pragma unroll
    for (int q = 0; q < DT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[q][e] - mean; qv = __builtin_fmaf(d, d, qv); }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) qv += __shfl_xor(qv, o);
    const float rstd = 1.0f / sqrtf(qv / (float)D + 1e-5f);
    DSG_TL_MARK(1);      // slabs summed, statistics (every load has landed)
    if (ok) {
#pragma unroll
        for (int q = 0; q < DT; ++q) {
            const int n = 64 * q + 4 * c;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((v[q][e] - mean) * rstd, pg[q][e], pt[q][e]);
            // (round 6: the boundary behind this kernel WAITS for these stores -- with them compiled out the gap to the next kernel drops from 2.78 to
            //  1.70 us and the 16-clip step from 208.6 to 195.6 us, profiles/r06_s_*: uncached stores drain at ~2 MB/us.  The ROWS set has no such pass.)
            *(f32x4*)(g.Xn + mr * D + n) = y;
            P::store4((elem*)g.Xa + qk_off<P>((int)mr, n, KD), y);
        }
    }
}


}  // namespace dsg
