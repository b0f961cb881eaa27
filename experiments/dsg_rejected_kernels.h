// dsg_rejected_kernels.h -- kernels that were built, parity-tested, MEASURED SLOWER on MI355X and taken out of the product
// library (round 3).  Kept compilable against the current headers (`make experiments`) as the record of what was tried; nothing
// in libdsg_hip.so references them.  Host-side drivers, env switches and tests lived in the tree up to commit 313453b (round 2).
//
//   k_gemm_tp      BM x 128 "textbook" blocks (A rows staged in LDS, 16 accumulators per wave): slower than the 32-row blocks at
//                  every size (5696 rows: QKV 31.8 vs 23.9 us, linear1 25.5 vs 23.0, pose head 70.7 vs 55.2;
//                  profiles/r02_k_b64_kernel_stats_tp.csv) -- weight fragments one k-block ahead from global memory, LayerNorm
//                  recomputed per column group, epilogue operands per tile.  Superseded by dsg_stream.h for the FFN GEMMs.
//   k_qkv_attn     LayerNorm + in_proj of one head + attention in one kernel: 192 KB of LDS traffic per workgroup and a 6x
//                  redundant K/V projection: 15 us against 6.2 + 4.6 for LN+QKV followed by k_attn (profiles/r01_c_*).
// Also removed, not kept as code (see git history at 313453b and the logs named):
//   gemm_body_mt   4 row tiles per workgroup on the 16 x 16 kernels: 513 vs 380 us/step at batch 16 (tools/b16_sweep.sh, round 1)
//   embedded-space state (k_loc_e, k_enoise, EPI_ESTEP, DSG_ECARRY): 115.9-118.7 vs 113.1-115.7 us/step (profiles/r02_f_ecarry_ab.log)
//   XCD-pinned lanes (PBF16X, k_*_x, PinTab, DSG_PIN): bit-identical, 203 vs 115 us/step for one lane (profiles/r02_n_pinned_lanes.log)
//   overlapped launches (DepWait, k_mid<OVL>, DSG_OVERLAP): 158 vs 144 us/step (round 1)
//   dsg_stream_ln.h (this directory): the weight-stationary GEMM with LayerNorm-on-read, round 3
#pragma once
#include "dsg_fused.h"
#include "dsg_batched.h"

namespace dsg {

// ---------------------------------------------------------------------------------------------------------
// k_gemm_tp: BM x 128 blocks (BM = 128, or 64 when the LDS block would not fit), the textbook throughput shape: the A rows are
// staged in LDS once (LayerNorm-on-read or copy), the 4 waves form a 2 x 2 grid of (BM/2) x 64 sub-blocks, per k-block a wave
// reads BM/32 A fragments from LDS and 4 weight fragments from global memory for (BM/32) x 4 MFMAs (16 accumulators).
// EXPERIMENT, off by default (DSG_GEMM_TP=1): on MI355X it is SLOWER than the 32-row blocks above at every batch size tried,
// including 5696 rows (batch 64: QKV 31.8 vs 23.9 us, linear1 25.5 vs 23.0, pose head 70.7 vs 55.2 us per launch).  These
// GEMMs have K = 256: eight k-blocks.  There is no long K loop to pipeline behind, so a workgroup's life is a chain of
// memory round trips of 1.5-2 us each (LayerNorm row loads, weight fragments one k-block ahead, per-tile epilogue operands)
// with 128 MFMAs per wave in between, and with 1-2 resident workgroups per CU nothing covers them.  The small shapes finish
// in one round trip per workgroup and let 3-4 workgroups per CU overlap.  What would be needed (all loads of a block in ONE
// batch: 128 rows x 1 KB + 32 weight fragments + column operands, ~260 VGPRs) no longer fits the register file.
// ---------------------------------------------------------------------------------------------------------
template <class P, int PRO, int EPI, int DMAX, int BM>
__global__ __launch_bounds__(256) void k_gemm_tp(const GemmArgs g) {
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem), WR = BM / 32, CT = 4;      // row tiles / column tiles per wave
    static_assert(EPI != EPI_PARTIAL, "split-K partials come from k_gemm_blk_k");
    static_assert(BM == 64 || BM == 128, "block rows");
    __shared__ __attribute__((aligned(16))) char lds_a[BM * (DMAX * ES + 16)];
    preload_kernargs(g);
    const int NGT = (g.NT + 7) / 8;                  // 128-column groups (the last one may be partial)
    const int ng = xcd_ngroup<P>(), mb = blockIdx.y;
    const int MB = (g.MT * 16 + BM - 1) / BM;
    if constexpr (EPI == EPI_OUT) {
        if (mb >= MB) {      // extra grid row: step bookkeeping (see gemm_body)
            if (g.ctl && blockIdx.x == 0 && threadIdx.x == 0 && g.out_mode != OUT_FORWARD) step_advance_A(g.ctl, g.st, g.n_tab);
            return;
        }
    }
    if (ng >= NGT || mb >= MB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = mb * BM;
    const int nt0 = ng * 8 + wc * 4;                 // first 16-column tile of this wave
    const int KBtot = g.KBtot, kb_last = KBtot - 1, nt_last = g.NT - 1;
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    // Q / K columns use the swapped product (4 consecutive features per lane), V the direct one; a wave's 64 columns never
    // straddle the boundary (H * hd is a multiple of 64)
    const bool swapped = !(EPI == EPI_QKV && (nt0 * 16) >= 2 * (g.H * g.hd));
    f32x4 bcur[CT], bnxt[CT];
    auto load_b = [&](f32x4 (&dst)[CT], int kb) {
        const int kc = min(kb, kb_last);
#pragma unroll
        for (int t = 0; t < CT; ++t) dst[t] = wbase[((size_t)min(nt0 + t, nt_last) * KBtot + kc) * 64];
    };
    load_b(bcur, 0);
    int step = 0;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f, k5 = 0.f;
    if constexpr (EPI == EPI_OUT) {
        if (g.out_mode != OUT_FORWARD) {
            step = ldw<P>(&g.ctl->stepB);
            k1 = ldwf<P>(&g.ctl->k1); k2 = ldwf<P>(&g.ctl->k2); k3 = ldwf<P>(&g.ctl->k3); k4 = ldwf<P>(&g.ctl->k4); k5 = ldwf<P>(&g.ctl->k5);
        }
    }
    // ---- stage the BM A rows in LDS
    const int K = KBtot * P::KB;
    const int pitch = K * ES + 16;
    if constexpr (PRO == PRO_LN) {
        const bool wrx = g.Xn != nullptr && ng == 0;
        const int nch = g.D >> 6;
#pragma unroll 1
        for (int r0 = 0; r0 < BM; r0 += 32) {
            if constexpr (DMAX <= 256) {
                if (nch == 4) ln_rows_blk<P, 4, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else if (nch == 2) ln_rows_blk<P, 2, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else if (nch == 3) ln_rows_blk<P, 3, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else ln_rows_blk<P, 1, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
            } else {
                if (nch == 8) ln_rows_blk<P, 8, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else if (nch == 6) ln_rows_blk<P, 6, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else if (nch == 5) ln_rows_blk<P, 5, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
                else ln_rows_blk<P, 7, 2>(g, m0 + r0, tid, lds_a + r0 * pitch, pitch, wrx);
            }
        }
    } else {
        const int cpr = K * ES / 16;                       // 16-byte chunks per row
        const int total = BM * cpr;
        for (int e0 = 0; e0 < total; e0 += 256 * 4) {
            f32x4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + u * 256 + tid, total - 1);
                const char* src;
                if (g.a_frag) src = (const char*)g.A + ((size_t)(m0 >> 4) * KBtot * 64 + e) * 16;
                else { const int r = e / cpr, cc = e - r * cpr; src = (const char*)g.A + ((size_t)(m0 + r) * g.lda) * ES + cc * 16; }
                tmp[u] = *(const f32x4*)src;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 + tid;
                if (e < total) {
                    int r, boff;
                    if (g.a_frag) { const int ln = e & 63, kb = (e >> 6) % KBtot, rt = (e >> 6) / KBtot; r = rt * 16 + (ln & 15); boff = (kb * P::KB + P::E * (ln >> 4)) * ES; }
                    else { r = e / cpr; boff = (e - r * cpr) * 16; }
                    *(f32x4*)(lds_a + r * pitch + boff) = tmp[u];
                }
            }
        }
    }
    DSG_LDS_BARRIER();
    // ---- main loop
    f32x4 acc[WR][CT];
#pragma unroll
    for (int rt = 0; rt < WR; ++rt)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* arow = lds_a + (wr * (BM / 2) + lr) * pitch + P::E * lg * ES;
    for (int kb = 0; kb < KBtot; ++kb) {
        load_b(bnxt, kb + 1);                              // next k-block's weight fragments under this block's MFMAs
        f32x4 a[WR];
#pragma unroll
        for (int rt = 0; rt < WR; ++rt) a[rt] = *(const f32x4*)(arow + rt * 16 * pitch + kb * P::KB * ES);
#pragma unroll
        for (int rt = 0; rt < WR; ++rt)
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[rt][t] = swapped ? P::mma(bcur[t], a[rt], acc[rt][t]) : P::mma(a[rt], bcur[t], acc[rt][t]);
#pragma unroll
        for (int t = 0; t < CT; ++t) bcur[t] = bnxt[t];
    }
    // ---- epilogue, tile by tile (throughput regime: other workgroups of the CU cover the operand latency)
#pragma unroll
    for (int rt = 0; rt < WR; ++rt) {
        const int mt = m0 + wr * (BM / 2) + rt * 16;
        if (mt >= g.MT * 16) continue;                      // wave-uniform
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if (nt0 + t > nt_last) continue;                // wave-uniform: partial last column group
            TileOps o;
            gemm_prefetch_tile<P, EPI>(g, mt, (nt0 + t) * 16, lr, lg, step, o);
            gemm_epilogue_tile<P, EPI>(g, mt, (nt0 + t) * 16, lr, lg, 0, swapped, acc[rt][t], o, k1, k2, k3, k4, k5);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// k_qkv_attn
// ---------------------------------------------------------------------------------------------------------
struct QkvAttnArgs {
    const void* Xa;         // layer 0: encoder input rows, P::elem [rows][D] (no LayerNorm)
    const float* X;         // layer > 0: pre-LayerNorm rows fp32 [rows][D]
    const float* ln_g; const float* ln_b;
    float* Xn;              // layer > 0: LayerNorm output rows (fp32), written by the head-0 workgroups
    const void* Wp;         // packed in_proj weight [3D/16][KD][64][16 B]
    const float* bias;      // [3D]
    void* out;              // [rows][D] P::elem attention output (heads concatenated)
    int B, H, ntok;
};

template <class P, int HD, int NKT, int DD>
__global__ __launch_bounds__(256) void k_qkv_attn(const QkvAttnArgs g) {
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int Tp = NKT * 16;
    constexpr int KD = DD / P::KB;                   // k-blocks of the projection
    constexpr int KH = HD / P::KB;                   // k-blocks of QK^T
    constexpr int NTH = HD / 16;                     // 16-col tiles per head (2 or 4)
    constexpr int MSPLIT = 4 / NTH;                  // waves sharing one n-tile split the row tiles
    static_assert(NTH == 2 || NTH == 4, "head dim 32 or 64");
    constexpr int XP = DD * ES + 16, KP = HD * ES + 16, VP = Tp * ES + 16;
    __shared__ __attribute__((aligned(16))) char lds[Tp * XP + Tp * KP + HD * VP + 16 * KP];
    char* const xs = lds;
    char* const kk = xs + Tp * XP;
    char* const vt = kk + Tp * KP;
    char* const qq = vt + HD * VP;

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;          // grid (query tiles, heads, batch)
    const size_t row0 = (size_t)b * g.ntok;

    // ---- (1) this wave's weight fragments: n-tile (wave % NTH) of Q, K and V of head h -- issued first
    const int wnt = wave % NTH, wms = wave / NTH;
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    f32x4 wf[3][KD];
#pragma unroll
    for (int mat = 0; mat < 3; ++mat) {
        const int nt = (mat * DD + h * HD) / 16 + wnt;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) wf[mat][kb] = wbase[((size_t)nt * KD + kb) * 64];
    }
    const f32x4 bq = *(const f32x4*)(g.bias + 0 * DD + h * HD + wnt * 16 + 4 * lg);
    const f32x4 bk = *(const f32x4*)(g.bias + 1 * DD + h * HD + wnt * 16 + 4 * lg);
    const float bv = g.bias[2 * DD + h * HD + wnt * 16 + lr];

    // ---- (2) token rows of this batch element -> LDS (LayerNorm-on-read for layers > 0)
    if (g.X) {
        constexpr int NCH = DD / 64;                 // float4 chunks per thread per row
        const int row = tid >> 4, c = tid & 15;
        f32x4 v[NKT][NCH];
#pragma unroll
        for (int p = 0; p < NKT; ++p)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                v[p][i] = *(const f32x4*)(g.X + (row0 + p * 16 + row) * DD + c * 4 + 64 * i);
        f32x4 gg[NCH], bb[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            gg[i] = *(const f32x4*)(g.ln_g + c * 4 + 64 * i);
            bb[i] = *(const f32x4*)(g.ln_b + c * 4 + 64 * i);
        }
#pragma unroll
        for (int p = 0; p < NKT; ++p) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) s += (v[p][i][0] + v[p][i][1]) + (v[p][i][2] + v[p][i][3]);
            s = row16_sum(s);
            const float mean = s / (float)DD;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[p][i][e] - mean; q += d * d; }
            q = row16_sum(q);
            const float rstd = 1.0f / sqrtf(q / (float)DD + 1e-5f);
            const int srow = p * 16 + row;
            const bool wr = g.Xn && h == 0 && p == qt && srow < g.ntok;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int col = c * 4 + 64 * i;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[p][i][e] - mean) * rstd * gg[i][e] + bb[i][e];
                P::store4((elem*)(xs + srow * XP) + col, y);
                if (wr) *(f32x4*)(g.Xn + (row0 + srow) * DD + col) = y;
            }
        }
    } else {
        constexpr int CPR = DD * ES / 16;            // 16-byte chunks per row
        constexpr int NCP = Tp * CPR / 256;
        static_assert((Tp * CPR) % 256 == 0, "copy tiling");
        f32x4 v[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
            const int e = tid + 256 * i, r = e / CPR, cc = e % CPR;
            v[i] = *(const f32x4*)((const char*)g.Xa + ((row0 + r) * DD) * ES + cc * 16);
        }
#pragma unroll
        for (int i = 0; i < NCP; ++i) {
            const int e = tid + 256 * i, r = e / CPR, cc = e % CPR;
            *(f32x4*)(xs + r * XP + cc * 16) = v[i];
        }
    }
    DSG_LDS_BARRIER();

    // ---- (3) K_h, V_h for every token and Q_h for this query tile (results stay in LDS)
    for (int mt = wms; mt < NKT; mt += MSPLIT) {
        f32x4 af[KD];
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) af[kb] = *(const f32x4*)(xs + (mt * 16 + lr) * XP + (kb * P::KB + P::E * lg) * ES);
        f32x4 ck = (f32x4){0.f, 0.f, 0.f, 0.f}, cv = ck;
#pragma unroll
        for (int kb = 0; kb < KD; ++kb) {
            ck = P::mma(wf[1][kb], af[kb], ck);      // D[dim 4lg+r][token lr]
            cv = P::mma(af[kb], wf[2][kb], cv);      // D[token 4lg+r][dim lr]
        }
        P::store4((elem*)(kk + (mt * 16 + lr) * KP) + wnt * 16 + 4 * lg, ck + bk);
        f32x4 vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[e] = cv[e] + bv;
        P::store4((elem*)(vt + (wnt * 16 + lr) * VP) + mt * 16 + 4 * lg, vv);
        if (mt == qt) {
            f32x4 cq = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) cq = P::mma(wf[0][kb], af[kb], cq);
            P::store4((elem*)(qq + lr * KP) + wnt * 16 + 4 * lg, cq + bq);
        }
    }
    DSG_LDS_BARRIER();

    // ---- (4) attention for the 16 queries of this tile; every wave forms the scores, wave w owns output dims
    f32x4 qf[KH];
#pragma unroll
    for (int kb = 0; kb < KH; ++kb) qf[kb] = *(const f32x4*)(qq + lr * KP + (kb * P::KB + P::E * lg) * ES);
    f32x4 s[NKT];
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt) {
        s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KH; ++kb) {
            const f32x4 kf = *(const f32x4*)(kk + (nt * 16 + lr) * KP + (kb * P::KB + P::E * lg) * ES);
            s[nt] = P::mma(kf, qf[kb], s[nt]);       // D[key 4lg+r][query lr]
        }
    }
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
            const float p = key < g.ntok ? P::exp_sm(s[nt][r] - mx) : 0.f;
            s[nt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    if (wave < NTH) {
        const int dt = wave;
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* vrow = vt + (dt * 16 + lr) * VP;
        if constexpr (P::E == 4) {
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt) o = P::mma(*(const f32x4*)(vrow + (nt * 16 + 4 * lg) * ES), s[nt], o);
        } else {
            static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
#pragma unroll
            for (int kb = 0; kb < NKT / 2; ++kb) {
                const f32x2 v0 = *(const f32x2*)(vrow + ((2 * kb) * 16 + 4 * lg) * ES);
                const f32x2 v1 = *(const f32x2*)(vrow + ((2 * kb + 1) * 16 + 4 * lg) * ES);
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                u16x8 pp;
#pragma unroll
                for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
                o = P::mma((f32x4){v0[0], v0[1], v1[0], v1[1]}, __builtin_bit_cast(f32x4, pp), o);
            }
        }
        const int q = qt * 16 + lr;
        if (q < g.ntok) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
            P::store4((elem*)g.out + qk_off<P>((int)(row0 + q), h * HD + dt * 16 + 4 * lg, DD / P::KB), y);     // fragment-major rows
        }
    }
}


// ---- round 6: retired with its switch (DSG_ATTN_OP2).  Two query tiles per workgroup; bit-identical to k_attn_op; won only in the round-4 STREAM set
//      (1 x 64: 478 -> 463 us), which k_clip_attn replaced in round 5.
// (Round 4: "bit-identical" was only true under the emulator -- on the device the compiler contracted the LayerNorm's mul + add into
// fma in k_attn_op and not in k_attn_op2, one ulp apart in ~10 % of the rows, enough to flip bf16 roundings downstream
// (tools/debug_op2.py).  Both kernels now spell the two fma sites out; tests/test_gpu_round4.py compares them on the device.)
// k_attn_op2: k_attn_op for TWO query tiles (32 queries) of a batch element per workgroup -- K, V^T (96 KB) and W_o (128 KB) are
// pulled through the CU's load path once per 32 rows instead of once per 16 (136 instead of 248 KB per tile), which is what bounds
// the kernel when the batch fills the GPU (>= 1400 token rows: STREAM set).  Row by row the arithmetic and its order are those of
// k_attn_op: bit-identical.
template <class P, int DT, int NKT>
__global__ __launch_bounds__(256) void k_attn_op2(const AttnOpArgs g) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int ES = (int)sizeof(elem);
    constexpr int D = DT * 64, HD = DT * 16;
    constexpr int KD = D / P::KB, KDH = HD / P::KB;
    constexpr int XP = D * ES + 16;
    constexpr int ND = HD / 16;
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;
    static_assert(KDH >= 1 && KD <= 8, "shape");
    static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
    __shared__ __attribute__((aligned(16))) char aT[2][16 * XP];
    __shared__ float red[2][2][4][16];
    __shared__ __attribute__((aligned(16))) float vecs[3][D];
    preload_kernargs(g);
    const int qt0 = 2 * (int)blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(), lr = lane & 15, lg = lane >> 4;
    const int h = wave;
    const size_t bh = (size_t)b * 4 + h;
    const elem* Q = (const elem*)g.q + bh * g.Tp * HD;
    const elem* K = (const elem*)g.k + bh * g.Tp * HD;
    const elem* VT = (const elem*)g.vt + bh * HD * g.Tp;
    const f32x4* wo = (const f32x4*)g.Wo + lane;
    const int nqt = g.Tp / 16;
    f32x4 qf[2][KDH], kf[NKT][KDH], vfr[ND][NVF];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) qf[j][kb] = *(const f32x4*)(Q + (size_t)((min(qt0 + j, nqt - 1) * KDH + kb) * 64 + lane) * P::E);
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KDH; ++kb) kf[nt][kb] = *(const f32x4*)(K + (size_t)((nt * KDH + kb) * 64 + lane) * P::E);
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) vfr[dt][kb] = *(const f32x4*)(VT + (size_t)((dt * NVF + kb) * 64 + lane) * P::E);
    bool rowok[2];
    size_t m[2];
    f32x4 pr[2][DT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tq = (qt0 + j) * 16 + lr;
        rowok[j] = tq < g.ntok;
        m[j] = (size_t)b * g.ntok + (rowok[j] ? tq : g.ntok - 1);      // clamped: unconditional loads, predicated stores
#pragma unroll
        for (int t = 0; t < DT; ++t) pr[j][t] = *(const f32x4*)(g.R + m[j] * D + (wave * DT + t) * 16 + 4 * lg);
    }
    constexpr int NV = (3 * D / 4 + 255) / 256;
    f32x4 vload[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = min(tid + 256 * i, 3 * D / 4 - 1), vsel = e / (D / 4), vidx = e % (D / 4);
        vload[i] = ((const f32x4*)(vsel == 0 ? g.bo : (vsel == 1 ? g.ln_g : g.ln_b)))[vidx];
    }
    DSG_LOADS_ISSUED();
    constexpr int KH = (KD + 1) / 2;
    f32x4 bf[KD][DT];
    const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // ---- S^T = K Q^T, softmax over the keys (D[key = 4*lg + r][query = lr])
        f32x4 s[NKT];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KDH; ++kb) s[nt] = P::mma(kf[nt][kb], qf[j][kb], s[nt]);
        }
        if (j == 0) {       // W_o: the first half of the k-blocks now (in flight during the softmax), the rest once V^T is dead
#pragma unroll
            for (int kb = 0; kb < KH; ++kb)
#pragma unroll
                for (int t = 0; t < DT; ++t) bf[kb][t] = wo[((size_t)(wave * DT + t) * KD + kb) * 64];
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
        // ---- O^T = V^T P^T -> LDS rows
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NVF; ++kb) o = P::mma(vfr[dt][kb], pfr[kb], o);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
            P::store4((elem*)(aT[j] + lr * XP) + h * HD + dt * 16 + 4 * lg, y);
        }
    }
#pragma unroll
    for (int kb = KH; kb < KD; ++kb)
#pragma unroll
        for (int t = 0; t < DT; ++t) bf[kb][t] = wo[((size_t)(wave * DT + t) * KD + kb) * 64];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = tid + 256 * i;
        if (e < 3 * D / 4) *(f32x4*)(&vecs[0][0] + e * 4) = vload[i];
    }
    DSG_LDS_BARRIER();
    // ---- out_proj from the LDS rows: one W_o fragment feeds both tiles
    f32x4 acc[2][DT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 af = *(const f32x4*)(aT[j] + lr * XP + (kb * P::KB + P::E * lg) * ES);
#pragma unroll
            for (int t = 0; t < DT; ++t) acc[j][t] = P::mma(bf[kb][t], af, acc[j][t]);      // D[n 4lg+r][row lr]
        }
    }
    // ---- residual + LayerNorm1 over whole rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float sm = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const f32x4 pbo = *(const f32x4*)(&vecs[0][(wave * DT + t) * 16 + 4 * lg]);
            acc[j][t] = acc[j][t] + pbo + pr[j][t];
            sm += (acc[j][t][0] + acc[j][t][1]) + (acc[j][t][2] + acc[j][t][3]);
        }
        sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
        if (lg == 0) red[j][0][wave][lr] = sm;
    }
    DSG_LDS_BARRIER();
    float mean[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        mean[j] = ((red[j][0][0][lr] + red[j][0][1][lr]) + (red[j][0][2][lr] + red[j][0][3][lr])) / (float)D;
        float qv = 0.f;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[j][t][e] - mean[j]; qv = __builtin_fmaf(d, d, qv); }
        qv += __shfl_xor(qv, 16); qv += __shfl_xor(qv, 32);
        if (lg == 0) red[j][1][wave][lr] = qv;
    }
    DSG_LDS_BARRIER();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float var = ((red[j][1][0][lr] + red[j][1][1][lr]) + (red[j][1][2][lr] + red[j][1][3][lr])) / (float)D;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (rowok[j]) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int n = (wave * DT + t) * 16 + 4 * lg;
                const f32x4 pg = *(const f32x4*)(&vecs[1][n]), pbt = *(const f32x4*)(&vecs[2][n]);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc[j][t][e] - mean[j]) * rstd, pg[e], pbt[e]);
                *(f32x4*)(g.X1 + m[j] * D + n) = y;
                P::store4((elem*)g.X1a + qk_off<P>((int)m[j], n, D / P::KB), y);
            }
        }
    }
}



}  // namespace dsg
