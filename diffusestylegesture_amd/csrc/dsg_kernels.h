// dsg_kernels.h -- hand-written gfx950 (CDNA4 / MI355X) kernels for the DiffuseStyleGesture sampling hot path.
//
// One denoising step = k_in (pose-embedding GEMM, split-K partials)
//                    -> k_loc (sum partials + per-window constants + time embedding, rotary, windowed causal
//                              local attention, token prepend, rotary)
//                    -> 8 x { k_gemm<QKV> (LayerNorm-on-read) , k_attn , k_gemm<RESID> (out_proj) ,
//                             k_gemm<GELU> (LayerNorm-on-read, linear1) , k_gemm<RESID> (linear2, in-WG split-K) }
//                    -> k_gemm<OUT> (LayerNorm-on-read, pose head, fused posterior / DDIM update with in-kernel
//                              Philox noise).
// Reference arithmetic being replaced (file:line under /root/reference):
//   MDM.forward                      main/model/mdm.py:166-233, :357      (BEAT-TWH-main/model/mdm.py:134-224)
//   rotary                           main/model/local_attention/rotary.py:8-27
//   LocalAttention.forward           main/model/local_attention/local_attention.py:91-199
//   nn.TransformerEncoderLayer x8    main/model/mdm.py:79-86 (PyTorch: post-norm, erf-GELU, eps 1e-5)
//   p_sample / q_posterior           main/diffusion/gaussian_diffusion.py:506-558, :256-278
//   ddim_sample                      main/diffusion/gaussian_diffusion.py:742-792
//   q_sample, xstart-from-eps        main/diffusion/gaussian_diffusion.py:236-254, :400-405
//
// MFMA usage.  Every contraction runs on the matrix cores through one abstraction: a "fragment" is 16 bytes per
// lane, taken at [row = lane&15][k-chunk = lane>>4] of a row-major operand.  For bf16 that is 8 k-values feeding one
// v_mfma_f32_16x16x32_bf16; for fp32 it is 4 k-values feeding four v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
// Because both operands use the same (lane>>4, j) -> k mapping, the hardware's internal k order never matters; only
// the C/D layout does: D[i = 4*(lane>>4) + r][j = lane&15], r = 0..3  (cdna_hip_programming.md s3).
// We mostly issue the "swapped" product D = W_tile . Act_tile^T so a lane ends up with FOUR CONSECUTIVE OUTPUT
// FEATURES of ONE token: 16-byte row-major stores, float4 bias loads, one Philox call per lane in the sampler epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dsg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

#define DSG_FLT_MAX 3.402823466e+38f

// ---------------------------------------------------------------------------------------------------------
// precision policies
// ---------------------------------------------------------------------------------------------------------
// float -> bf16, round-to-nearest-even: gfx950 has it in hardware (v_cvt_pk_bf16_f32, two values per instruction).
// A hand-rolled integer RNE with a NaN branch costs ~12 instructions and an EXEC-mask branch per value -- at one wave
// per SIMD every instruction is ~4 cycles of latency on the critical path, and k_attn alone converts 24 values/lane.
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, ((unsigned)h) << 16); }

// Per-lane index arithmetic in 32 bits on the 24-bit multiplier (round 6).  A 64-bit product of lane-varying operands -- `(size_t)row * pitch` -- costs
// the compiler a v_mad_u64_u32 + v_mul_lo_u32 group, all quarter rate (16 cycles per wave instruction): the STREAM pose head spent more cycles on
// such products (27 v_mul_lo + 12 v_mad_u64 per Philox call) than on Philox itself, k_loc a third of its VALU time.  v_mul_u32_u24 is full rate; every
// row / frame / feature count of the path is < 2^24 and every ELEMENT offset < 2^31 (dsg_create checks max_batch against that).
__device__ __forceinline__ unsigned imul24(int a, int b) { return __umul24((unsigned)a, (unsigned)b); }
// 16-lane ("DPP row") butterfly reductions: quad_perm xor-1, xor-2, row_half_mirror, row_mirror -- four DPP moves
// (a few cycles each) instead of four ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;     // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;     // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;    // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;   // row_mirror
    }
    return __builtin_bit_cast(float, r);
}
// 2^x as one v_exp_f32 (no denormal-range fix-up: callers' arguments stay far from it or do not care)
#ifndef DSG_EMU
__device__ __forceinline__ float dsg_exp2f(float x) { return __builtin_amdgcn_exp2f(x); }
#else
__device__ __forceinline__ float dsg_exp2f(float x) { return std::exp2(x); }
#endif
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f(v, 0); v += dpp_f(v, 1); v += dpp_f(v, 2); v += dpp_f(v, 3);
    return v;
}

// LayerNorm statistics of rows spread over the 4 lane groups x NW waves of a workgroup in ONE exchange (round 6): every (wave, lane) forms the sum and
// the centred second moment of its own 4 DW values, the lane groups of a wave merge theirs with two shuffle steps and the waves theirs through LDS --
// M2(a + b) = M2(a) + M2(b) + (mean_a - mean_b)^2 n_a n_b / (n_a + n_b) (Chan et al.), which is as stable as the two-pass form it replaces and needs no
// second round trip through LDS for the mean.  Every lane of a row evaluates the same expressions in the same order: identical statistics.
template <int DW>
__device__ __forceinline__ void ln_wave_moments(const f32x4 (&v)[DW], float s, float* red_s, float* red_m2, int lr, int lg) {
    constexpr float n0 = 4.0f * DW;
    const float c = s / n0;
    float m2 = 0.f;
#pragma unroll
    for (int t = 0; t < DW; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[t][e] - c; m2 = __builtin_fmaf(d, d, m2); }
    {
        const float so = __shfl_xor(s, 16), mo = __shfl_xor(m2, 16);
        const float d = (s - so) / n0;
        m2 = __builtin_fmaf(d * d, 0.5f * n0, m2 + mo); s += so;
    }
    {
        const float so = __shfl_xor(s, 32), mo = __shfl_xor(m2, 32);
        const float d = (s - so) / (2.0f * n0);
        m2 = __builtin_fmaf(d * d, n0, m2 + mo); s += so;
    }
    if (lg == 0) { red_s[lr] = s; red_m2[lr] = m2; }
}
template <int NW, int D>
__device__ __forceinline__ void ln_combine_moments(const float (&red_s)[NW][16], const float (&red_m2)[NW][16], int lr, float& mean, float& var) {
    constexpr float nw = (float)(D / NW);
    float sw[NW], tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sw[w] = red_s[w][lr]; tot += sw[w]; }
    mean = tot / (float)D;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const float d = sw[w] / nw - mean; m2 += __builtin_fmaf(d * d, nw, red_m2[w][lr]); }
    var = m2 / (float)D;
}

struct PF32 {                       // fp32 storage, v_mfma_f32_16x16x4_f32
    typedef float elem;
    static constexpr int E = 4;     // elements per 16-byte fragment
    static constexpr int KB = 16;   // k-values per k-block (4 lane groups x E)
    static __device__ __forceinline__ elem cvt(float f) { return f; }
    static __device__ __forceinline__ float up(elem e) { return e; }
    static __device__ __forceinline__ float exp_sm(float x) { return expf(x); }         // softmax exp: accurate
    static __device__ __forceinline__ f32x4 mma(f32x4 a, f32x4 b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ void store4(elem* p, f32x4 v) { *(f32x4*)p = v; }
    // weight fragments (see PBF16W2): one 16-byte fragment per k-block
    typedef f32x4 wfrag;
    static constexpr int WF = 1;    // 16-byte registers per weight fragment
    static constexpr bool W2 = false;
    static __device__ __forceinline__ wfrag wload(const f32x4* base, size_t frag) { return base[frag * 64]; }      // base = packed weights + lane
    static __device__ __forceinline__ f32x4 mma_w(const wfrag& w, f32x4 a, f32x4 c) { return mma(w, a, c); }     // D = W_tile . Act_tile^T
    static __device__ __forceinline__ f32x4 mma_a(f32x4 a, const wfrag& w, f32x4 c) { return mma(a, w, c); }     // D = Act_tile . W_tile^T
    // GEMM A operands that a kernel of the step produces itself (LayerNorm rows / attention rows in LDS, `hidden`): one image here,
    // a hi + lo pair in PBF16W2.  lo_off: byte distance of the lo image (LDS) / ignored
    typedef f32x4 afrag;
    static constexpr int AF = 1;
    static __device__ __forceinline__ afrag azero() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void store4_a(elem* p, int lo_off, f32x4 v) { (void)lo_off; store4(p, v); }
    static __device__ __forceinline__ afrag aload(const char* p, int lo_off) { (void)lo_off; return *(const f32x4*)p; }
    // fragment-major global A operand written by one kernel and read by the next (`hidden`): element offset of the PBF16 order -> here
    static __device__ __forceinline__ void store4_afrag(elem* base, size_t off, f32x4 v) { store4(base + off, v); }
    static __device__ __forceinline__ afrag aload_frag(const void* base, size_t frag, int lane) { return *(const f32x4*)((const char*)base + (frag * 64 + lane) * 16); }
    // ... at the element offset of a lane's 16 bytes (qk_off of a row that need not start a tile), and one element of it
    static __device__ __forceinline__ afrag aload_off(const void* base, size_t off) { return *(const f32x4*)((const char*)base + off * sizeof(elem)); }
    static __device__ __forceinline__ void store1_afrag(elem* base, size_t off, float v) { base[off] = cvt(v); }
};
struct PBF16 {                      // bf16 storage, v_mfma_f32_16x16x32_bf16, fp32 accumulate
    typedef bf16_t elem;
    static constexpr int E = 8;
    static constexpr int KB = 32;
    static __device__ __forceinline__ elem cvt(float f) { return f2bf(f); }
    static __device__ __forceinline__ float up(elem e) { return bf2f(e); }
    static __device__ __forceinline__ float exp_sm(float x) { return __expf(x); }       // softmax exp: v_exp_f32 (P is rounded to bf16 anyway)
    static __device__ __forceinline__ f32x4 mma(f32x4 a, f32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(elem* p, f32x4 v) {
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        *(bf16x4v*)p = __builtin_convertvector(v, bf16x4v);
    }
    typedef f32x4 wfrag;
    static constexpr int WF = 1;
    static constexpr bool W2 = false;
    static __device__ __forceinline__ wfrag wload(const f32x4* base, size_t frag) { return base[frag * 64]; }
    static __device__ __forceinline__ f32x4 mma_w(const wfrag& w, f32x4 a, f32x4 c) { return mma(w, a, c); }
    static __device__ __forceinline__ f32x4 mma_a(f32x4 a, const wfrag& w, f32x4 c) { return mma(a, w, c); }
    // GEMM A operands that a kernel of the step produces itself (LayerNorm rows / attention rows in LDS, `hidden`): one image here,
    // a hi + lo pair in PBF16W2.  lo_off: byte distance of the lo image (LDS) / ignored
    typedef f32x4 afrag;
    static constexpr int AF = 1;
    static __device__ __forceinline__ afrag azero() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void store4_a(elem* p, int lo_off, f32x4 v) { (void)lo_off; store4(p, v); }
    static __device__ __forceinline__ afrag aload(const char* p, int lo_off) { (void)lo_off; return *(const f32x4*)p; }
    // fragment-major global A operand written by one kernel and read by the next (`hidden`): element offset of the PBF16 order -> here
    static __device__ __forceinline__ void store4_afrag(elem* base, size_t off, f32x4 v) { store4(base + off, v); }
    static __device__ __forceinline__ afrag aload_frag(const void* base, size_t frag, int lane) { return *(const f32x4*)((const char*)base + (frag * 64 + lane) * 16); }
    // ... at the element offset of a lane's 16 bytes (qk_off of a row that need not start a tile), and one element of it
    static __device__ __forceinline__ afrag aload_off(const void* base, size_t off) { return *(const f32x4*)((const char*)base + off * sizeof(elem)); }
    static __device__ __forceinline__ void store1_afrag(elem* base, size_t off, float v) { base[off] = cvt(v); }
};
// "bf16w2" (round 5): bf16 activations, every WEIGHT as the sum of two bf16 numbers -- hi = bf16(w), lo = bf16(w - hi): 16 mantissa
// bits instead of 8 -- two v_mfma_f32_16x16x32_bf16 per fragment, fp32 accumulate.  The bf16 drift of a 1000-step chain is the
// weights' 8-bit mantissa (tests/bf16_ablation.py: 5.9e-3 of 6.2e-3), and at batch 1 the MFMA pipe is ~1 % busy; what the mode costs
// is the fp32-sized weight traffic.  Everything that is not a weight (activations, Q / K / V, softmax, the state shadow) is exactly
// PBF16.  Packed layout: fragment f of the PBF16 order becomes the two consecutive 1 KB blocks 2f (hi) and 2f + 1 (lo), so every
// fragment INDEX of the kernels stays what it is; a weight fragment is two registers quads (wfrag) loaded by wload.
struct PBF16W2 : PBF16 {
    struct wfrag { f32x4 h, l; };
    static constexpr int WF = 2;
    static constexpr bool W2 = true;
    static __device__ __forceinline__ wfrag wload(const f32x4* base, size_t frag) { wfrag w; w.h = base[frag * 128]; w.l = base[frag * 128 + 64]; return w; }
    static __device__ __forceinline__ f32x4 mma_w(const wfrag& w, f32x4 a, f32x4 c) { return PBF16::mma(w.l, a, PBF16::mma(w.h, a, c)); }
    static __device__ __forceinline__ f32x4 mma_a(f32x4 a, const wfrag& w, f32x4 c) { return PBF16::mma(a, w.l, PBF16::mma(a, w.h, c)); }
    // ... and the A operands the step's kernels hand to each other through LDS (LayerNorm1 / LayerNorm2 rows, the attention rows of
    // k_attn_mid) or as `hidden`: hi + lo as well, three MFMAs per fragment (w.h a.h + w.l a.h + w.h a.l; the lo x lo term is below
    // fp32 rounding).  The ablation (tests/bf16_ablation.py): with all weights exact the chain still drifts 3.1e-3 from exactly these
    // rounding points; what stays single bf16 is Q / K / V / P, x_t and the embedding output (1.4e-4 ... 4.1e-4 each).
    struct afrag { f32x4 h, l; };
    static constexpr int AF = 2;
    static __device__ __forceinline__ afrag azero() { afrag a; a.h = a.l = (f32x4){0.f, 0.f, 0.f, 0.f}; return a; }
    static __device__ __forceinline__ void split4(f32x4 v, u16x4& hi, u16x4& lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = f2bf(v[e]); lo[e] = f2bf(v[e] - bf2f(hi[e])); }
    }
    static __device__ __forceinline__ void store4_a(elem* p, int lo_off, f32x4 v) {
        u16x4 hi, lo; split4(v, hi, lo);
        *(u16x4*)p = hi; *(u16x4*)((char*)p + lo_off) = lo;
    }
    static __device__ __forceinline__ afrag aload(const char* p, int lo_off) { afrag a; a.h = *(const f32x4*)p; a.l = *(const f32x4*)(p + lo_off); return a; }
    // fragment f of the PBF16 order -> 1 KB blocks 2f (hi) and 2f + 1 (lo), like the weights
    static __device__ __forceinline__ void store4_afrag(elem* base, size_t off, f32x4 v) {
        u16x4 hi, lo; split4(v, hi, lo);
        const size_t o2 = off + (off / (64 * E)) * (64 * E);
        *(u16x4*)(base + o2) = hi; *(u16x4*)(base + o2 + 64 * E) = lo;
    }
    static __device__ __forceinline__ afrag aload_frag(const void* base, size_t frag, int lane) {
        afrag a; a.h = *(const f32x4*)((const char*)base + (frag * 128 + lane) * 16); a.l = *(const f32x4*)((const char*)base + (frag * 128 + 64 + lane) * 16); return a;
    }
    static __device__ __forceinline__ afrag aload_off(const void* base, size_t off) {
        const size_t o2 = off + (off / (64 * E)) * (64 * E);
        afrag a; a.h = *(const f32x4*)((const elem*)base + o2); a.l = *(const f32x4*)((const elem*)base + o2 + 64 * E); return a;
    }
    static __device__ __forceinline__ void store1_afrag(elem* base, size_t off, float v) {
        const size_t o2 = off + (off / (64 * E)) * (64 * E);
        const elem hi = f2bf(v);
        base[o2] = hi; base[o2 + 64 * E] = f2bf(v - bf2f(hi));
    }
    static __device__ __forceinline__ f32x4 mma_w(const wfrag& w, const afrag& a, f32x4 c) { return PBF16::mma(w.h, a.l, PBF16::mma(w.l, a.h, PBF16::mma(w.h, a.h, c))); }
    static __device__ __forceinline__ f32x4 mma_a(const afrag& a, const wfrag& w, f32x4 c) { return PBF16::mma(a.l, w.h, PBF16::mma(a.h, w.l, PBF16::mma(a.h, w.h, c))); }
};

// Loads of data another kernel of the step loop wrote.  The loop-written buffers live in uncached device memory (dsg_hip.cpp:
// uc_mode), so a plain load is coherent; the step control is a few words that must never come through the scalar cache
// (nothing invalidates it between the fence-free packets): agent-scope atomic loads are vector loads.
template <class P> __device__ __forceinline__ f32x4 lda16(const void* base, size_t off) { return *(const f32x4*)((const char*)base + off); }
template <class P> __device__ __forceinline__ int ldw(const int* p) {
#ifndef DSG_EMU
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}
template <class P> __device__ __forceinline__ float ldwf(const float* p) { return __builtin_bit_cast(float, ldw<P>((const int*)p)); }

// Q, K and V^T of the self-attention live in HBM in MFMA-FRAGMENT order, so that every fragment a wave loads is one
// contiguous 1 KB block (8 cache lines) instead of 16 rows x 64 B (16 half-used lines): the attention kernels' load
// phases are bound by the CU's load path, which works per line.  Inside one (batch, head) block:
//   Q / K : [token / 16][d / KB][lane = ((d % KB) / E) * 16 + token % 16][E]
//   V^T   : [dim / 16][kb][lane = lg * 16 + dim % 16][E]   with the tokens of PV k-block kb held by lane group lg:
//           fp32: token = 16 kb + 4 lg + r;  bf16: token = 16 (2 kb + e) + 4 lg + r at position 4 e + r -- the two key
//           tiles a bf16 PV k-block pairs (the values a lane's softmax registers already hold), one 16-byte load.
// The QKV epilogue scatters into this order (stores are off the critical path), the attention kernels read lane-linear.
template <class P>
__device__ __forceinline__ int qk_off(int tok, int d, int kdh) {
    return (((((int)imul24(tok >> 4, kdh)) + d / P::KB) * 64 + ((d % P::KB) / P::E) * 16 + (tok & 15)) * P::E) + (d % P::E);
}
template <class P>
__device__ __forceinline__ int vt_off(int dim, int tok, int nvf) {
    const int lg = (tok & 12) >> 2, r = tok & 3;
    if constexpr (P::E == 4) return ((((dim >> 4) * nvf + (tok >> 4)) * 64 + lg * 16 + (dim & 15)) * 4) + r;
    else return ((((dim >> 4) * nvf + (tok >> 5)) * 64 + lg * 16 + (dim & 15)) * 8) + ((tok >> 2) & 4) + r;
}
// element offset of (row, j .. j + 3) in the compute-dtype shadow of the sampler state: row-major [rows][Jp], or fragment-major
// (the GEMM operand order, qk_off) when the kernel set streams it
template <class P>
__device__ __forceinline__ size_t xs_off(int row, int j, int Jp, int frag) {
    return frag ? (size_t)qk_off<P>(row, j, Jp / P::KB) : (size_t)(imul24(row, Jp) + (unsigned)j);
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Load phases are written branch-free (clamped addresses, select afterwards): a load under a runtime predicate makes
// hipcc place `s_waitcnt vmcnt(0)` at every join, which turns N independent loads into N serial memory round trips
// (seen in the ISA as L w L w L w ...; the LayerNorm GEMMs had 17-19 of them).  LOADS_ISSUED() additionally stops the
// scheduler from sinking loads below the first use.
#define DSG_LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)
#ifndef DSG_BUILD_TAG
#define DSG_BUILD_TAG 0u
#endif
// checksum of the sources this device code was built from (Makefile); the AQL path compares the tag inside
// dsg_kernels.hsaco with the library's own before it trusts the code object with the library's argument structs
extern "C" __device__ const unsigned dsg_device_build_tag = DSG_BUILD_TAG;

// Timeline build (make stamps -> libdsg_hip_stamps.so + dsg_kernels_stamps.hsaco; tools/aql_timeline.py): every step kernel
// stamps the 100 MHz steady counter (s_memrealtime) when its first wave starts and when its last wave ends, into the slot the
// AQL submission wrote BEHIND the kernel's arguments for this particular packet (dsg_aql.h: Trace; slot < 0: an untraced
// packet, nothing happens).  One atomic min / max per wave; compiled out of the product library.
// Phase marks (round 6; -DDSG_STAMPS=2, `make marks` -> libdsg_hip_marks.so + dsg_kernels_marks.hsaco): DSG_TL_MARK(k), k = 0 .. 11, placed between
// the phases of a kernel, stores the same counter into slot k of the wave's entry when the wave PASSES that point (instruction issue: a mark
// behind the first use of a batch of loads reads "the loads have landed").  The table per kernel -- mean and last wave per mark, relative to the
// kernel's first wave -- is what tools/aql_timeline.py --lib marks prints.  Marks pin the instruction schedule around them (sched_barrier),
// so the marks build is for reading phases, the plain stamps build for kernel totals.
#ifdef DSG_STAMPS
constexpr int DSG_TL_NMARK = 12;
#if DSG_STAMPS >= 2
struct TlEntry { unsigned long long t0, t1; unsigned long long m[DSG_TL_NMARK]; };
#else
struct TlEntry { unsigned long long t0, t1; };
#endif
constexpr int DSG_TL_WAVES = 2048, DSG_TL_ARG_OFF = 128;       // entries (waves) per traced packet; slot index + ring pointer: reserved
                                                               // dwords 128 / 136 of the implicit-argument block
// the entry of this wave in the ring of its packet, or null (an untraced packet)
__device__ __forceinline__ TlEntry* tl_entry() {
    const char* ia = (const char*)__builtin_amdgcn_implicitarg_ptr();
    const int slot = *(const int*)(ia + DSG_TL_ARG_OFF);
    if (slot < 0) return nullptr;
    TlEntry* ring = *(TlEntry* const*)(ia + DSG_TL_ARG_OFF + 8);
    const unsigned w = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * ((blockDim.x + 63) >> 6) + (threadIdx.x >> 6);
    // grids of more than 2048 waves: the first 1024 keep their own entries (-> first start), the later ones share the other
    // 1024 round-robin; the last writers of those are the last waves of the grid (-> last end)
    const unsigned ent = w < 1024u ? w : 1024u + (w & 1023u);
    return ring + (size_t)slot * DSG_TL_WAVES + ent;
}
#if DSG_STAMPS >= 2
// marks are parked in LDS while the kernel runs (one 8-byte LDS store by lane 0 per mark: no address arithmetic, no vector-memory traffic, no
// live registers between marks) and copied to the wave's ring entry by its first 12 lanes when the wave ends
__device__ __forceinline__ unsigned long long* tl_lds() { __shared__ unsigned long long m[16 * DSG_TL_NMARK]; return m; }
#endif
struct TlScope {           // every wave stores its own start / end stamp (plain 8-byte stores to its own entry: no contention)
    bool on;
    __device__ __forceinline__ TlScope() : on(false) {
        TlEntry* p = tl_entry();
        on = p != nullptr;
        if (on) {
#if DSG_STAMPS >= 2
            if ((threadIdx.x & 63) < DSG_TL_NMARK) tl_lds()[(threadIdx.x >> 6) * DSG_TL_NMARK + (threadIdx.x & 63)] = 0ull;
#endif
            if ((threadIdx.x & 63) == 0) p->t0 = __builtin_readsteadycounter();
        }
    }
    __device__ __forceinline__ ~TlScope() {
        if (on) {
            const unsigned long long t = __builtin_readsteadycounter();       // (the stamp FIRST: the entry's address costs a scalar-cache miss)
            __builtin_amdgcn_sched_barrier(0);
            TlEntry* p = tl_entry();       // (recomputed: nothing of the scope stays live in vector registers across the kernel)
#if DSG_STAMPS >= 2
            if ((threadIdx.x & 63) < DSG_TL_NMARK) p->m[threadIdx.x & 63] = tl_lds()[(threadIdx.x >> 6) * DSG_TL_NMARK + (threadIdx.x & 63)];
#endif
            if ((threadIdx.x & 63) == 0) p->t1 = t;
        }
    }
};
#define DSG_TL_SCOPE() TlScope dsg_tl_scope_
#if DSG_STAMPS >= 2
__device__ __forceinline__ void tl_mark(int k) {
    if ((threadIdx.x & 63) == 0) tl_lds()[(threadIdx.x >> 6) * DSG_TL_NMARK + k] = __builtin_readsteadycounter();
}
#define DSG_TL_MARK(k) do { __builtin_amdgcn_sched_barrier(0); dsg::tl_mark(k); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSG_TL_MARK(k) ((void)0)
#endif
#else
#define DSG_TL_SCOPE() ((void)0)
#define DSG_TL_MARK(k) ((void)0)
#endif
// Workgroup barrier for LDS hand-offs only.  __syncthreads() also drains every outstanding VECTOR memory operation
// (s_waitcnt vmcnt(0)): in-flight weight loads and the acknowledgement of global stores issued before it.  Nothing in
// these kernels communicates through global memory inside a launch, so the fences are restricted to the LDS address
// space ("local"): they keep the compiler from moving LDS accesses across the barrier and emit lgkmcnt(0) only.
#define DSG_LDS_BARRIER()                                                      \
    do {                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");        \
        __builtin_amdgcn_s_barrier();                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");        \
    } while (0)
// Hand-off through LDS between the lanes of ONE wave (a wave's LDS instructions execute in order, so nothing is emitted: the
// fences only pin the compiler's order; the host emulator runs lanes as fibers and needs a real rendezvous).
#ifndef DSG_EMU
#define DSG_WAVE_LDS_SYNC()                                                    \
    do {                                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");        \
        __builtin_amdgcn_wave_barrier();                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");        \
    } while (0)
#else
#define DSG_WAVE_LDS_SYNC() ((void)__shfl_xor(0, 0))
#endif
// hipcc fetches kernel arguments lazily, one s_load batch (+ s_waitcnt lgkmcnt(0)) per region that first needs them;
// the kernarg segment is freshly written for every launch, so each batch is a scalar-cache MISS -- several serial
// misses per kernel.  Touching one dword per 64-byte line of the argument struct at the top of the kernel makes all
// those misses overlap; the later s_loads then hit the scalar cache.
template <class T>
__device__ __forceinline__ void preload_kernargs(const T& a) {
    const unsigned* p = (const unsigned*)&a;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i += 16) { const unsigned v = p[i]; asm volatile("" ::"r"(v)); }
    { const unsigned v = p[sizeof(T) / 4 - 1]; asm volatile("" ::"r"(v)); }
}
// The wave index must be PROVABLY wave-uniform (an SGPR): loop bounds and predicates derived from threadIdx.x >> 6 are
// otherwise treated as divergent, the compiler predicates the loop body through EXEC, and v_mfma IGNORES EXEC -- a
// "skipped" MFMA of a partial k-chunk then runs on uninitialised fragments (seen on gfx950: NaNs in the last split-K
// slice; invisible under emulation).  readfirstlane makes every such branch a scalar branch.
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: the framework's noise stream (restated for the CPU in oracle/philox.py)
// ---------------------------------------------------------------------------------------------------------
struct NoiseKey { unsigned k0, k1, s0, s1; };          // seed lo/hi, stream lo/hi

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// Box-Muller pieces for the restricted arguments of the noise stream (round 6).  The libm routines (logf, sqrtf, sincospif: ~180 instructions per
// Philox call with their general-argument handling) made the pose head of the STREAM set ALU-bound (2300 VALU instructions per wave, 13 us of
// VALU issue per launch at 64 clips -- tools/pmc_valu.sh); here u1 is in [2^-24, 1] and t = 2 u2 a multiple of 2^-23 in [0, 2):
//   r      = sqrt(-2 ln 2 . log2 u1)      v_log_f32 + v_sqrt_f32 (1 ulp each)
//   sincos = quarter-turn reduction (exact: t - n / 2 with n = rint(2 t)) + degree-7 / degree-8 polynomials on [-1/4, 1/4]: 1.74 ulp max over all
//            2^24 arguments (numpy float32 emulation); the oracle (oracle/philox.py) computes the transform in float64 and rounds.
// tools/noise_probe.cpp measures both against double arithmetic on the device, and the old against the new normals.
#ifndef DSG_EMU
__device__ __forceinline__ float dsg_log2f(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float dsg_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float dsg_log2f(float x) { return std::log2(x); }
__device__ __forceinline__ float dsg_sqrtf(float x) { return std::sqrt(x); }
#endif
__device__ __forceinline__ void dsg_sincospi_02(float t, float& s, float& c) {          // sin(pi t), cos(pi t) for t in [0, 2)
    const float n = __builtin_rintf(t + t);                                           // quarter turns 0 .. 4
    const float f = __builtin_fmaf(n, -0.5f, t), f2 = f * f;                          // exact; |f| <= 1/4
    const float sp = f * __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-0.5890104174613953f, f2, 2.5497608184814453f), f2, -5.167707443237305f), f2, 3.1415927410125732f);
    const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.231417715549469f, f2, -1.3350569009780884f), f2, 4.0587077140808105f), f2, -4.934802055358887f), f2, 1.0f);
    const int q = (int)n;
    const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
    s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) ^ ((unsigned)(q & 2) << 30));
    c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) ^ ((unsigned)((q + 1) & 2) << 30));
}
// four standard normals for element quad q (elements 4q..4q+3) of draw `draw`
__device__ __forceinline__ f32x4 philox_normal4(unsigned q, unsigned draw, NoiseKey key) {
    unsigned x[4];
    philox4x32_10(q, draw, key.s0, key.s1, key.k0, key.k1, x);
    const float sc = 5.9604644775390625e-08f;          // 2^-24
    float u1a = (float)((x[0] >> 8) + 1u) * sc, u2a = (float)(x[1] >> 8) * sc;
    float u1b = (float)((x[2] >> 8) + 1u) * sc, u2b = (float)(x[3] >> 8) * sc;
    float sa, ca, sb, cb;
    const float ra = dsg_sqrtf(-1.3862943611198906f * dsg_log2f(u1a)), rb = dsg_sqrtf(-1.3862943611198906f * dsg_log2f(u1b));
    dsg_sincospi_02(2.0f * u2a, sa, ca);
    dsg_sincospi_02(2.0f * u2b, sb, cb);
    f32x4 z; z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
    return z;
}

// ---------------------------------------------------------------------------------------------------------
// per-step control block (device memory): lets ONE captured hipGraph serve every step
// ---------------------------------------------------------------------------------------------------------
// Device-resident step control.  The FIRST kernel of a step reads {tA}; at its END its block 0 advances the B side
// (stepB, coefficients of that step).  The LAST kernel reads the B side; at its END its block 0 advances the A side
// (stepA, tA = model timestep of the next step).  No kernel reads what it writes, a kernel boundary separates every
// write from the next read, and readers need ONE load level (no ctr -> table -> table chains on the critical path).
struct StepCtl {
    int stepA, tA;             // read by k_loc / k_inloc
    int stepB;                 // read by the sampler epilogue (noise draw index, replayed-noise slot)
    float k1, k2, k3, k4, k5;  // coefficients of step stepB (see StepTables)
};

struct StepTables {            // execution-ordered, one entry per step that will run
    const int*   tmodel;       // model timestep fed to the denoiser (timestep_map[idx])
    const float* c1;           // DDPM: posterior_mean_coef1        | DDIM: sqrt_recip_alphas_cumprod
    const float* c2;           // DDPM: posterior_mean_coef2        | DDIM: sqrt_recipm1_alphas_cumprod
    const float* c3;           // DDPM: nonzero * exp(0.5*logvar)    | DDIM: sqrt(alpha_bar_prev)
    const float* c4;           //                                    | DDIM: sqrt(1 - abar_prev - sigma^2)
    const float* c5;           //                                    | DDIM: nonzero * sigma
};

template <class P = PF32>
__device__ __forceinline__ void step_advance_B(StepCtl* c, const StepTables& st, int n) {
    const int s = ldw<P>(&c->stepB) + 1, i = s < n ? s : n - 1;
    const float a1 = st.c1[i], a2 = st.c2[i], a3 = st.c3[i], a4 = st.c4[i], a5 = st.c5[i];   // loads first, then stores
    c->stepB = s;
    c->k1 = a1; c->k2 = a2; c->k3 = a3; c->k4 = a4; c->k5 = a5;
}
template <class P = PF32>
__device__ __forceinline__ void step_advance_A(StepCtl* c, const StepTables& st, int n) {
    const int s = ldw<P>(&c->stepA) + 1;
    c->stepA = s;
    c->tA = st.tmodel[s < n ? s : n - 1];
}
// `first`: the loop index the call starts at (0, or where a chain run in pieces resumes)
__global__ void k_ctl_init(StepCtl* c, const int* tmodel, int first) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { c->stepA = first; c->tA = tmodel[first]; c->stepB = first - 1; c->k1 = c->k2 = c->k3 = c->k4 = c->k5 = 0.f; }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM  (skinny-M, weight-stationary-per-XCD):  Out[m][n] = sum_k Act[m][k] * W[n][k]  (+ epilogue)
// ---------------------------------------------------------------------------------------------------------
enum { PRO_DIRECT = 0, PRO_LN = 1 };
enum { EPI_PARTIAL = 0, EPI_QKV = 1, EPI_RESID = 2, EPI_GELU = 3, EPI_OUT = 4 };
enum { OUT_FORWARD = 0, OUT_DDPM = 1, OUT_DDIM = 2 };

struct GemmArgs {
    // problem
    int M;                  // valid rows
    int MT;                 // row tiles (16 rows each)
    int NT;                 // 16-col output tiles
    int KBtot;              // k-blocks in the packed weight (K_padded / P::KB)
    int KS;                 // split-K across workgroups (EPI_PARTIAL only), else 1
    int kb_per_split;       // k-blocks per split (= KBtot when KS == 1)
    const void* Wp;         // packed weights: [NT][KBtot][64 lanes][16 B]
    const float* bias;      // [NT*16]
    // A operand
    const void* A;          // PRO_DIRECT: row-major P::elem [>=MT*16][lda]
    int lda;
    const float* X;         // PRO_LN: fp32 rows [>=MT*16][D]  (pre-LayerNorm residual stream)
    const float* ln_g;
    const float* ln_b;
    float* Xn;              // PRO_LN: normalised rows written back (fp32) by n-group 0, may be null
    int D;                  // PRO_LN: row width (= K)
    // epilogue
    void* out;              // EPI_GELU: P::elem [M][ldo] | EPI_RESID / EPI_PARTIAL: float [..][ldo]
    int ldo;
    const float* R;         // EPI_RESID residual [M][ldo]
    // EPI_QKV
    void* q; void* k; void* vt;
    int ntok, Tp, H, hd;    // tokens per batch element, padded tokens, heads, head dim
    unsigned inv_ntok, inv_hd;   // fastdiv_inv(ntok), fastdiv_inv(hd)
    // EPI_OUT
    int out_mode;           // OUT_FORWARD / OUT_DDPM / OUT_DDIM
    int J, Jp, Jq, T;       // pose dim, padded (row pitch of xs), noise pitch, frames
    float* xs32;            // [B][T][Jp] fp32 master state (in/out)
    void* xsA;              // [B][T][Jp] P::elem shadow for k_in (bf16 mode) or null
    float* fwd_out;         // OUT_FORWARD: [B][J][T]
    StepCtl* ctl;           // step control block (EPI_OUT reads the B side and advances the A side; EPI_PARTIAL
                            // advances the B side); null outside the sampling loop
    StepTables st;          // tables the block-0 bookkeeping reads (n_tab entries)
    int n_tab;
    const unsigned* dyn;    // device: {seed lo, seed hi, stream lo, stream hi, draw index of step 0} -- kept out of
                            // the kernel arguments so a captured graph is reusable across windows / clips
    const float* ext_noise; // optional [n_steps][B][J][T] replayed noise, else null
    int B;
    int const_noise;
    int ws_G;               // dsg_stream.h: row-block groups of the persistent grid (multiple of 8)
    int a_frag;             // PRO_DIRECT: A is stored fragment-major ([row tile][k-block][64 lanes][16 B], qk_off) -- hidden, attention rows
    int out_frag;           // EPI_GELU: the output goes out fragment-major (it is the next GEMM's A operand)
    // EPI_OUT, classifier-free guidance (main/model/cfg_sampler.py:8-31): the batch holds cfgB conditional elements followed
    // by their cfgB unconditional twins (same x_t); the workgroup of a conditional row tile also evaluates the twin rows
    // (cfg_off rows further down) and forms x0 = x0_u + scale[b] * (x0_c - x0_u) before the sampler update, which it writes to
    // BOTH halves of the state.  cfgB == 0: off.
    int cfgB, cfg_off;
    const float* cfg_scale; // [cfgB]
    int clip_x0;            // EPI_OUT: clamp x0 to [-1, 1] (clip_denoised=True, gaussian_diffusion.py:377-379)
    unsigned inv_ntok4;     // EPI_QKV block kernels: fastdiv_inv(ntok rounded up to 4) (vt_store_block)
    int xs_frag;            // EPI_OUT: the state shadow xsA is stored fragment-major (qk_off over [B T][Jp]): the STREAM set's pose
                            // embedding streams it like every other operand (k_ws2<EPI_PARTIAL>)
    int no_noise;           // EPI_OUT: the sampler adds no noise at any step of this call (DDIM with eta = 0: sigma = 0,
                            // gaussian_diffusion.py:782-791) -- the Philox draw is skipped, x_{t-1} = mean + 0 z bit for bit
};

// Workgroup -> n-group with the n-group pinned to an XCD (workgroups are dealt round-robin to the 8 XCDs in linear
// launch order; gridDim.x is a multiple of 8, so blockIdx.x & 7 is the XCD): a weight slice is always fetched through
// the same XCD's L2 and stays resident there across the 1000 steps.  Row tile / k-split come from blockIdx.y / .z --
// the hardware hands them over for free, whereas decomposing a flat id costs integer divisions (~40 instructions
// each, and at one wave per SIMD every instruction is ~2 ns of critical path).
template <class P> __device__ __forceinline__ int xcd_ngroup() { const int x = (int)blockIdx.x; return (x & 7) + 8 * (x >> 3); }
__host__ __device__ inline int xcd_grid_x(int NG) { return 8 * ((NG + 7) / 8); }

// (gemm_body, k_gemm_blk, k_gemm_blk_k) column group ng runs on XCD ng % 8 -- grid x is padded to a multiple of 8 --, which leaves the XCDs with NG / 8 or NG / 8 + 1
// groups each.  From 16 rows of the grid on, the NG % 8 groups past the last full round of 8 are dealt out by grid row over the 8 padded columns (item j = y * 8 + xcd),
// so every XCD gets the same share of them; the dead slots are the last rows of the grid.  Returns false for a dead slot.
__device__ __forceinline__ bool xcd_deal_groups(int NG, int n_rows, int& ng, int& row) {
    const int NG8 = NG & ~7, rem = NG - NG8;
    if (rem && ng >= NG8 && n_rows >= 16) {
        const int j = row * 8 + (ng - NG8);
        if (j >= rem * n_rows) return false;
        row = (int)((unsigned)j / (unsigned)rem);
        ng = NG8 + (j - row * rem);
    }
    return true;
}
// x / d for 0 <= x, x * d < 2^32, as one v_mul_hi_u32: inv = ceil(2^32 / d) (host: fastdiv_inv; d == 1 -> inv 0)
__device__ __forceinline__ int fdiv(int x, unsigned inv) { return inv ? (int)__umulhi((unsigned)x, inv) : x; }
__host__ __device__ inline unsigned fastdiv_inv(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1u) / (unsigned)d); }

// V^T of a whole row block, written in ALIGNED token groups.  A GEMM lane holds 4 consecutive ROWS of the batch for one feature; the
// fragment order stores tokens in groups of 4 per batch element, and with ntok = 89 a row quad is such a group only for every
// fourth batch element -- the other three paid 4 element stores per quad (2-byte uncached stores: LN + QKV 16.8 -> 9.6 us at
// 5696 rows when removed).  So V blocks go through LDS: `stage[f * SP + r]` = V(row m0 + r, feature n0 + f) in the GEMM type
// (already with bias), and the workgroup's threads walk the (feature, aligned group) pairs the block touches: one 8- / 16-byte
// store per group whose tokens all lie in the block (the tokens past ntok of an element's last group are written as 0 -- the
// attention kernels multiply them by P = 0), element stores only for the two groups a block boundary cuts.
// n0: first staged column, counted from the start of V (a multiple of 16 inside one head or spanning whole heads).
template <class P, int NCOLS>
__device__ __forceinline__ void vt_store_block(const GemmArgs& g, const typename P::elem* stage, int SP, int m0, int nrows, int n0, int tid) {
    typedef typename P::elem elem;
    typedef elem elem4 __attribute__((ext_vector_type(4)));
    static_assert(256 % NCOLS == 0, "threads per feature");
    const int ntok = g.ntok, ntok4 = (ntok + 3) & ~3;
    const int m_end = min(m0 + nrows, g.M);                  // rows [m0, m_end) are this block's
    if (m_end <= m0) return;
    const int b_lo = fdiv(m0, g.inv_ntok);
    const int rp_lo = (b_lo * ntok4 + (m0 - b_lo * ntok)) & ~3;          // first aligned slot in padded-row space (row' = b ntok4 + token)
    const int b_hi = fdiv(m_end - 1, g.inv_ntok);
    const int rp_hi = b_hi * ntok4 + (m_end - 1 - b_hi * ntok);
    const int ngroups = ((rp_hi - rp_lo) >> 2) + 1;
    const int nvf = P::E == 4 ? g.Tp / 16 : g.Tp / 32;
    const int f = tid % NCOLS;                               // one feature per thread, 256 / NCOLS threads share its groups
    const int nn = n0 + f, head = fdiv(nn, g.inv_hd), d = nn - head * g.hd;
    const elem* col = stage + f * SP;
    for (int j = tid / NCOLS; j < ngroups; j += 256 / NCOLS) {
        const int rp = rp_lo + 4 * j;
        const int b = fdiv(rp, g.inv_ntok4), s0 = rp - b * ntok4;        // tokens s0 .. s0 + 3 of batch element b
        elem* dst = (elem*)g.vt + ((size_t)b * g.H + head) * g.hd * g.Tp;
        const int mrow = b * ntok + s0 - m0;                 // block row of token s0
        bool whole = true;
        elem4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = mrow + e;
            const bool pad = s0 + e >= ntok;
            const bool own = !pad && r >= 0 && r < m_end - m0;
            whole = whole && (own || pad);
            v[e] = own ? col[r] : elem(0);
        }
        if (whole) {
            *(elem4*)(dst + vt_off<P>(d, s0, nvf)) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = mrow + e;
                if (s0 + e < ntok && r >= 0 && r < m_end - m0) dst[vt_off<P>(d, s0 + e, nvf)] = v[e];
            }
        }
    }
}

// Exact (erf) GELU.  fp32 kernels call erff.  bf16w2 (its `hidden` keeps 16 mantissa bits) uses Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7 in erf,
// ~17 instructions with v_rcp / v_exp) instead of the ~60-instruction branchy libm routine.  Plain bf16 rounds the result to 8 mantissa bits, and
// from 16 clips on the feed-forward kernels are BOUND by this function (round 6, phase marks: 65 536 evaluations per 64-row block of k_ffn = 8.5 us
// of VALU on a CU whose weight stream needs 7.8): there  gelu(x) = x Phi(x)  is evaluated as  x / (1 + 2^(x p(x^2)))  with the odd polynomial
// -log2(e) logit(Phi(x)) ~ x (c0 + c1 x^2 + c2 x^4) fitted in fp32 arithmetic on [-8, 8] (x clamped there; beyond, Phi is 0 / 1 to 1e-12):
// max |error| 2.6e-5 absolute, 1.2e-3 relative wherever |gelu| > 0.01 -- below half a bf16 ulp (2e-3) -- in 9 instructions (v_med3, 2 mul,
// 2 fma, v_exp, add, v_rcp, mul).  The tanh form of the literature has 4.7e-4 / 4.7e-2 for the same cost less one fma.
template <class P>
__device__ __forceinline__ float gelu_erf(float x) {
    if constexpr (sizeof(typename P::elem) == 4) {
        return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    } else if constexpr (P::W2) {
        const float ax = fabsf(x) * 0.70710678118654752440f;
        const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float er = 1.0f - poly * __expf(-ax * ax);
        return 0.5f * x * (1.0f + copysignf(er, x));
    } else {
        const float xc = fminf(fmaxf(x, -8.0f), 8.0f), x2 = xc * xc;
        const float t = xc * __builtin_fmaf(__builtin_fmaf(0.0010147836f, x2, -0.10677913f), x2, -2.301118f);
        return x * __builtin_amdgcn_rcpf(1.0f + dsg_exp2f(t));
    }
}

#define DSG_LDS_ROW_BYTES(K, ES) ((K) * (ES) + 16)

// LayerNorm of 16 rows by 256 lanes (16 lanes per row, NCH float4 chunks per lane), result to LDS in the MFMA element
// type and kept in v[] (fp32) for the write-back.  NCH > 0: exact width, no dead work.  NCH == 0: any D <= 512 that is
// a multiple of 64 (loads clamped to chunk 0 and weighted out, so the load phase stays branch-free).
// LEAN (batched path): the occupancy of these kernels is what bounds the batched step (rocprofv3 SQ counters,
// profiles/r01_k_pmc_sq_b1_b16.log: 2 waves per SIMD at ~200 VGPRs -> 2.8 rounds of workgroups at batch 16), so scale /
// shift are fetched chunk by chunk after the statistics instead of being held for the whole row, and the normalised rows
// are written back at once (`xn_out`) instead of staying live across the MFMA phase.
template <class P, int NCH, bool LEAN = false>
__device__ __forceinline__ void ln_rows(const GemmArgs& g, int m0, int tid, char* lds_a, int pitch, f32x4 (&v)[8], float* xn_out = nullptr, int lo_off = 0) {
    typedef typename P::elem elem;
    constexpr int N = NCH > 0 ? NCH : 8;
    const int D = g.D;
    const int row = tid >> 4, c = tid & 15;
    const size_t xr = (size_t)(m0 + row) * D;        // element offset of the row in g.X
    const int nch = NCH > 0 ? NCH : (D >> 6);
    f32x4 gg[LEAN ? 1 : N], bb[LEAN ? 1 : N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int col = c * 4 + 64 * (i < nch ? i : 0);
        v[i] = lda16<P>(g.X, (xr + col) * sizeof(float));
        if constexpr (!LEAN) {
            gg[i] = *(const f32x4*)(g.ln_g + col);
            bb[i] = *(const f32x4*)(g.ln_b + col);
        }
    }
    DSG_LOADS_ISSUED();

    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float wgt = (NCH > 0 || i < nch) ? 1.f : 0.f;
        s += wgt * ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3]));
    }
    s = row16_sum(s);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float wgt = (NCH > 0 || i < nch) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += wgt * d * d; }
    }
    q = row16_sum(q);
    const float rstd = 1.0f / sqrtf(q / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (NCH > 0 || i < nch) {
            f32x4 y;
            if constexpr (LEAN) {
                const f32x4 gi = *(const f32x4*)(g.ln_g + c * 4 + 64 * i), bi = *(const f32x4*)(g.ln_b + c * 4 + 64 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * gi[e] + bi[e];
                if (xn_out) *(f32x4*)(xn_out + c * 4 + 64 * i) = y;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * gg[i][e] + bb[i][e];
                v[i] = y;
            }
            P::store4_a((elem*)(lds_a + row * pitch) + c * 4 + 64 * i, lo_off, y);
        }
}

// Epilogue operands of one 16 x 16 output tile (bias, residual rows, x_t, noise): requested early, they do not depend on
// the main loop
struct TileOps { f32x4 pb, pr, pz; float pbs; bool ovalid; };
// x_t of (row m, features j0 .. j0 + 3) for the sampler epilogue: unconditional load from a clamped (always valid) row
template <class P>
__device__ __forceinline__ f32x4 out_xt_load(const GemmArgs& g, int m, int j0) {
    // row of (batch element b, frame sx - 1) in the state = b T + sx - 1 = m - b - 1 (ntok = T + 1); token rows (sx = 0) and rows past the batch are
    // clamped to a valid row, their value is never used
    const int b = fdiv(m, g.inv_ntok);
    const int row = min(max(m - b - 1, 0), g.B * g.T - 1);
    return lda16<P>(g.xs32, (size_t)((imul24(row, g.Jp) + (unsigned)j0) * 4u));
}
template <class P, int EPI>
__device__ __forceinline__ void gemm_prefetch_tile(const GemmArgs& g, int m0, int n0, int lr, int lg, int step, TileOps& o, const f32x4* xt_ready = nullptr) {
        o.pb = o.pr = o.pz = (f32x4){0.f, 0.f, 0.f, 0.f};
        o.pbs = 0.f; o.ovalid = false;
        if constexpr (EPI == EPI_RESID) {
            o.pb = *(const f32x4*)(g.bias + n0 + 4 * lg);
            o.pr = lda16<P>(g.R, ((size_t)(m0 + lr) * g.ldo + n0 + 4 * lg) * sizeof(float));     // rows are padded to the tile
        } else if constexpr (EPI == EPI_GELU) {
            o.pb = *(const f32x4*)(g.bias + n0 + 4 * lg);
        } else if constexpr (EPI == EPI_QKV) {
            o.pb = *(const f32x4*)(g.bias + n0 + 4 * lg);      // both forms loaded unconditionally (no branchy loads)
            o.pbs = g.bias[n0 + lr];
        } else if constexpr (EPI == EPI_OUT) {
            const int m = m0 + lr, j0 = n0 + 4 * lg;
            const int b = fdiv(m, g.inv_ntok), sx = m - (int)imul24(b, g.ntok);
            o.ovalid = m < g.M && sx > 0 && j0 < g.J;
            o.pb = *(const f32x4*)(g.bias + j0);
            // x_t: unconditional load from a clamped (always valid) row; unused when the lane is not `ovalid`
            o.pr = xt_ready ? *xt_ready : out_xt_load<P>(g, m, j0);
            if (o.ovalid && g.out_mode != OUT_FORWARD && !g.no_noise) {
                const int f = sx - 1;
                const int bn = g.const_noise ? 0 : b;
                if (g.ext_noise) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o.pz[e] = (j0 + e < g.J)
                                       ? g.ext_noise[(((size_t)step * g.B + bn) * g.J + j0 + e) * g.T + f] : 0.f;
                } else {
                    const NoiseKey nk = {g.dyn[0], g.dyn[1], g.dyn[2], g.dyn[3]};
                    o.pz = philox_normal4((imul24((int)imul24(bn, g.T) + f, g.Jq) + (unsigned)j0) >> 2, g.dyn[4] + (unsigned)step, nk);
                }
            }
        }
}

template <class P, int EPI>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmArgs& g, int m0, int n0, int lr, int lg, int ks, bool swapped, const f32x4& acc,
                                                   const TileOps& o, float k1, float k2, float k3, float k4, float k5,
                                                   const f32x4& acc_u = (f32x4){0.f, 0.f, 0.f, 0.f}) {
    typedef typename P::elem elem;
        if constexpr (EPI == EPI_PARTIAL) {
            const int m = m0 + lr;
            if (m < g.M) *(f32x4*)((float*)g.out + ((size_t)ks * g.MT * 16 + m) * g.ldo + n0 + 4 * lg) = acc;
        } else if constexpr (EPI == EPI_RESID) {
            const int m = m0 + lr, n = n0 + 4 * lg;
            if (m < g.M) *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = acc + o.pb + o.pr;
        } else if constexpr (EPI == EPI_GELU) {
            const int m = m0 + lr, n = n0 + 4 * lg;
            if (m < g.M) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = gelu_erf<P>(acc[e] + o.pb[e]);
                if (g.out_frag) P::store4_afrag((elem*)g.out, (size_t)qk_off<P>(m, n, g.ldo / P::KB), y);      // `hidden` (PBF16W2: hi + lo)
                else P::store4((elem*)g.out + (size_t)m * g.ldo + n, y);
            }
        } else if constexpr (EPI == EPI_QKV) {
            const int Dm = g.H * g.hd;
            const int which = (n0 >= Dm) + (n0 >= 2 * Dm), nn = n0 - which * Dm;
            const int head = fdiv(nn, g.inv_hd), d0 = nn - head * g.hd;
            if (swapped) {                    // Q or K: [B][H][Tp][hd], 4 consecutive dims of one token
                const int m = m0 + lr;
                if (m < g.M) {
                    const int b = fdiv(m, g.inv_ntok), sx = m - b * g.ntok;
                    elem* dst = (elem*)(which == 0 ? g.q : g.k) + ((size_t)b * g.H + head) * g.Tp * g.hd + qk_off<P>(sx, d0 + 4 * lg, g.hd / P::KB);
                    P::store4(dst, acc + o.pb);
                }
            } else {                             // V: transposed [B][H][hd][Tp], lane = one dim, 4 consecutive tokens
                // the 4 tokens of a lane are adjacent in the fragment order when they start a group of 4 inside one batch element
                // (always at batch 1): one 8- / 16-byte store instead of 4 element stores (the buffer is uncached memory)
                const int mq = m0 + 4 * lg;
                const int bq = fdiv(mq, g.inv_ntok), sq = mq - bq * g.ntok;
                if ((sq & 3) == 0 && sq + 3 < g.ntok && mq + 3 < g.M) {
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[e] + o.pbs;
                    P::store4((elem*)g.vt + ((size_t)bq * g.H + head) * g.hd * g.Tp + vt_off<P>(d0 + lr, sq, P::E == 4 ? g.Tp / 16 : g.Tp / 32), y);
                } else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + 4 * lg + e;
                    if (m < g.M) {
                        const int b = fdiv(m, g.inv_ntok), sx = m - b * g.ntok;
                        ((elem*)g.vt)[((size_t)b * g.H + head) * g.hd * g.Tp + vt_off<P>(d0 + lr, sx, P::E == 4 ? g.Tp / 16 : g.Tp / 32)] = P::cvt(acc[e] + o.pbs);
                    }
                }
            }
        } else if constexpr (EPI == EPI_OUT) {
            const int m = m0 + lr, j0 = n0 + 4 * lg;
            const int b = fdiv(m, g.inv_ntok), sx = m - (int)imul24(b, g.ntok);
            if (o.ovalid) {
                const int f = sx - 1;
                const int srow = m - b - 1;                 // = b T + f: row of the state
                f32x4 x0 = acc + o.pb;
                if (g.cfgB > 0) {                       // cfg_sampler.py:31: out_uncond + scale * (out - out_uncond)
                    const f32x4 xu = acc_u + o.pb;
                    const float sc = g.cfg_scale[b];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x0[e] = xu[e] + sc * (x0[e] - xu[e]);
                }
                if (g.clip_x0) {                        // gaussian_diffusion.py:377-379
#pragma unroll
                    for (int e = 0; e < 4; ++e) x0[e] = fminf(fmaxf(x0[e], -1.0f), 1.0f);
                }
                if (g.out_mode == OUT_FORWARD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j0 + e < g.J) g.fwd_out[((size_t)b * g.J + j0 + e) * g.T + f] = x0[e];
                } else {
                    const f32x4 xt = o.pr, z = o.pz;
                    f32x4 xn;
                    if (g.out_mode == OUT_DDPM) {       // gaussian_diffusion.py:264-267, :557
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mean = k1 * x0[e] + k2 * xt[e];
                            xn[e] = mean + k3 * z[e];
                        }
                    } else {                            // gaussian_diffusion.py:773-791
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float eps = (k1 * xt[e] - x0[e]) / k2;
                            const float mean = x0[e] * k3 + k4 * eps;
                            xn[e] = mean + k5 * z[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j0 + e >= g.J) xn[e] = 0.f;
                    *(f32x4*)((char*)g.xs32 + (size_t)((imul24(srow, g.Jp) + (unsigned)j0) * 4u)) = xn;
                    if (g.xsA) P::store4((elem*)g.xsA + xs_off<P>(srow, j0, g.Jp, g.xs_frag), xn);
                    if (g.cfgB > 0) {                   // the unconditional twin advances with the same x_{t-1}
                        const int trow = srow + (int)imul24(g.cfgB, g.T);
                        *(f32x4*)((char*)g.xs32 + (size_t)((imul24(trow, g.Jp) + (unsigned)j0) * 4u)) = xn;
                        if (g.xsA) P::store4((elem*)g.xsA + xs_off<P>(trow, j0, g.Jp, g.xs_frag), xn);
                    }
                }
            }
        }
}

// WN x WK = 4 waves: WN waves side by side along N (TNW 16-col tiles each), WK-way split of K inside the workgroup.
// CFG (EPI_OUT + PRO_LN only): two passes over the same weight fragments -- the unconditional twin rows first, then the
// conditional rows -- combined in the epilogue (see GemmArgs::cfgB).
// CH: k-blocks whose fragments are requested in one batch.  A wave's whole k range must fit ONE batch wherever it can: the next
// chunk's weight fragments are requested only after the current chunk's MFMAs, i.e. every further chunk is an exposed L2 round
// trip (~1 us at one wave per SIMD).  8 covers K = 256 in bf16 (ZEGGS); the DSG+ widths (K = 384 / 512: 12 / 16 k-blocks) get
// their own instantiations (round 4: the K = D GEMMs of BEAT / TWH ran two serial load phases each).
template <class P, int PRO, int EPI, int WN, int WK, int TNW, bool LEAN = false, bool CFG = false, int CH = 8>
__device__ __forceinline__ void gemm_body(const GemmArgs& g) {
    static_assert(!CFG || (EPI == EPI_OUT && PRO == PRO_LN && !LEAN), "guidance lives in the pose-head epilogue");
    typedef typename P::elem elem;
    static_assert(WN * WK == 4, "4 waves");
    constexpr int ES = (int)sizeof(elem);
    constexpr bool IS_LN = PRO == PRO_LN;
    constexpr int LDS_A = 16 * (512 * ES + 16);      // 16 rows of up to 512 elements (PBF16W2: a second image of the same size, the lo halves)
    __shared__ __attribute__((aligned(16))) char lds_a[IS_LN ? LDS_A * P::AF : 16];
    // PBF16W2: the A operand is a hi + lo pair where a kernel of the step produced it for this GEMM -- the LayerNorm rows (LDS) and
    // the two direct GEMMs with the residual epilogue: out_proj (k_attn's rows) and linear2 (`hidden`), both fragment-major
    // (round 6: ... and the pose head of the ROWS set, which reads the LayerNorm2 rows k_ffn wrote as a hi + lo pair)
    constexpr bool A2 = P::W2 && (IS_LN || (PRO == PRO_DIRECT && (EPI == EPI_RESID || EPI == EPI_OUT)));
    typedef typename std::conditional<A2, typename P::afrag, f32x4>::type AFrag;
    __shared__ __attribute__((aligned(16))) float lds_red[WK > 1 ? (WK - 1) * WN * TNW * 64 * 4 : 4];

    preload_kernargs(g);
    const int NG = g.NT / (WN * TNW);
    int ng = xcd_ngroup<P>();
    const int ks = blockIdx.z;
    int mt_first = blockIdx.y;
    if constexpr (EPI == EPI_PARTIAL || EPI == EPI_OUT) {
        // step bookkeeping runs in ONE EXTRA workgroup (first block of an extra grid row), concurrently with the real
        // work and off every critical path; see StepCtl for why this is race free
        if (mt_first >= g.MT) {
            if (g.ctl && blockIdx.x == 0 && ks == 0 && threadIdx.x == 0) {
                if constexpr (EPI == EPI_PARTIAL) step_advance_B<P>(g.ctl, g.st, g.n_tab);
                else if (g.out_mode != OUT_FORWARD) step_advance_A<P>(g.ctl, g.st, g.n_tab);
            }
            return;
        }
    }
    // Round 6: column group ng runs on XCD ng % 8 (grid x is padded to a multiple of 8: a group's weight columns stay in ONE L2), which left the
    // XCDs with NG / 8 or NG / 8 + 1 groups each -- the pose head at 16 clips: 18 groups, XCDs 0-1 three, the others two: 270 against 180
    // workgroups, the kernel as long as its two fullest XCDs (marks: loads requested at 5.1 us on average, 10.7 for the last wave).  The NG % 8
    // groups past the last full round of 8 are now dealt out by ROW TILE over the 8 padded columns of the grid (item j = y * 8 + xcd), so every
    // XCD gets the same share of them; the dead slots are the last rows of the grid and exit at once.  Same tiles, same arithmetic.
    // From 16 row tiles only: with few row tiles every XCD finishes in one round of its CUs anyway, and the fixed group -> XCD map keeps a group's
    // weight columns resident in one L2 across the steps (batch 1 with the remap: 106.9 -> 108.7 us per step; 16 clips 187.7 -> 185.6, 4 x 8 clips
    // 202.0 -> 199.4, TILE at 4 clips 157.4 -> 153.6 -- profiles/r06_cd_ab_dev{A,B}.log).
    if (!xcd_deal_groups(NG, g.MT, ng, mt_first)) return;
    if (ng >= NG || mt_first >= g.MT) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int wn = wave % WN, wk = wave / WN;
    const int lr = lane & 15, lg = lane >> 4;
    const int nt0 = (ng * WN + wn) * TNW;

    // ---- k range of this wave (split-K over workgroups may be uneven: the last split takes what is left)
    const int kb_lo_wg = ks * g.kb_per_split;
    const int kb_hi_wg = min(kb_lo_wg + g.kb_per_split, g.KBtot);
    const int kb_per_w = (kb_hi_wg - kb_lo_wg) / WK;          // host guarantees divisibility when WK > 1
    const int kb_lo = kb_lo_wg + wk * kb_per_w;
    const int kb_hi = WK > 1 ? kb_lo + kb_per_w : kb_hi_wg;

    // ---- main loop
    const f32x4* wbase = (const f32x4*)g.Wp + lane;
    // V tiles of the QKV projection use the un-swapped product (4 consecutive tokens per lane -> V^T rows)
    bool swapped[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) swapped[t] = !(EPI == EPI_QKV && ((nt0 + t) * 16) >= 2 * (g.H * g.hd));

    // All fragment loads of a chunk are issued before its first MFMA, and the NEXT chunk's weight fragments are
    // requested while the current chunk's MFMAs run: at these sizes a kernel is a latency chain, so every load that
    // does not depend on the prologue (weights, bias, residual, x_t, noise) is in flight before the LayerNorm starts.
    typename P::wfrag bf[CH][TNW];
    const int kb_last = g.KBtot - 1;
    auto load_b = [&](int kb0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kb = min(kb0 + c, kb_last);           // clamped, never predicated
#pragma unroll
            for (int t = 0; t < TNW; ++t) bf[c][t] = P::wload(wbase, (size_t)(nt0 + t) * g.KBtot + kb);
        }
    };
    load_b(kb_lo);
    int step = 0;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f, k5 = 0.f;
    if constexpr (EPI == EPI_OUT) {
        if (g.out_mode != OUT_FORWARD) {
            step = ldw<P>(&g.ctl->stepB);
            k1 = ldwf<P>(&g.ctl->k1); k2 = ldwf<P>(&g.ctl->k2); k3 = ldwf<P>(&g.ctl->k3); k4 = ldwf<P>(&g.ctl->k4); k5 = ldwf<P>(&g.ctl->k5);
        }
    }
    const int m0 = mt_first * 16;
    f32x4 acc[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t arow_off = 0;                             // element offset of this lane's row-major A fragment
    if constexpr (PRO == PRO_DIRECT) arow_off = (size_t)(m0 + lr) * g.lda + P::E * lg;
    // ---- epilogue operands (bias, residual, x_t, step coefficients, noise) do not depend on the main loop: fetch
    //      them now so their latency overlaps the weight / activation fragment loads
    TileOps ops[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) gemm_prefetch_tile<P, EPI>(g, m0, (nt0 + t) * 16, lr, lg, step, ops[t]);
    DSG_TL_MARK(0);          // weight fragments + epilogue operands requested (EPI_OUT: the Philox draw is behind this mark)

    // ---- prologue: LayerNorm-on-read (rows are owned whole: K == D)
    int pitch = 0;
    f32x4 v[IS_LN ? 8 : 1];
    bool wr = false;
    f32x4 acc_u[CFG ? TNW : 1];
#pragma unroll
    for (int t = 0; t < (CFG ? TNW : 1); ++t) acc_u[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NPASS = CFG ? 2 : 1;
#pragma unroll
    for (int pass = NPASS - 1; pass >= 0; --pass) {
    const int mp = m0 + pass * (CFG ? g.cfg_off : 0);     // pass 1 (first): the unconditional twin rows
    if constexpr (CFG) {
        if (pass == 0) {
#pragma unroll
            for (int t = 0; t < TNW; ++t) { acc_u[t] = acc[t]; acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            DSG_LDS_BARRIER();                            // every wave is done with the twin rows in lds_a
            if (kb_hi - kb_lo > CH) load_b(kb_lo);
        }
    }
    if constexpr (IS_LN) {
        pitch = DSG_LDS_ROW_BYTES(g.D, ES);
        wr = (g.Xn != nullptr) && ng == 0 && (mp + (tid >> 4)) < g.M;
        const int nch = g.D >> 6;                     // D / 64 float4 chunks per thread; one straight-line copy per width
        if constexpr (LEAN) {
            float* xn_out = wr ? g.Xn + (size_t)(mp + (tid >> 4)) * g.D : nullptr;
            if (nch == 4) ln_rows<P, 4, true>(g, mp, tid, lds_a, pitch, v, xn_out, LDS_A);
            else if (nch == 6) ln_rows<P, 6, true>(g, mp, tid, lds_a, pitch, v, xn_out, LDS_A);
            else if (nch == 8) ln_rows<P, 8, true>(g, mp, tid, lds_a, pitch, v, xn_out, LDS_A);
            else ln_rows<P, 0, true>(g, mp, tid, lds_a, pitch, v, xn_out, LDS_A);
            wr = false;                               // already written
        } else {
            if (nch == 4) ln_rows<P, 4>(g, mp, tid, lds_a, pitch, v, nullptr, LDS_A);
            else if (nch == 6) ln_rows<P, 6>(g, mp, tid, lds_a, pitch, v, nullptr, LDS_A);
            else if (nch == 8) ln_rows<P, 8>(g, mp, tid, lds_a, pitch, v, nullptr, LDS_A);
            else ln_rows<P, 0>(g, mp, tid, lds_a, pitch, v, nullptr, LDS_A);
        }
        DSG_LDS_BARRIER();
        DSG_TL_MARK(1);      // LayerNorm-on-read rows in LDS
    }

    for (int kb0 = kb_lo; kb0 < kb_hi; kb0 += CH) {
        AFrag af[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kb = min(kb0 + c, kb_last);
            if constexpr (PRO == PRO_DIRECT && A2) {
                af[c] = P::aload_frag(g.A, (size_t)(mt_first * g.KBtot + kb), lane);       // (attention rows / `hidden`: always fragment-major)
            } else if constexpr (PRO == PRO_DIRECT) {
                // fragment-major A: one contiguous 1 KB block per wave load (8 cache lines instead of 16 half-used ones)
                const size_t ao = g.a_frag ? ((size_t)(mt_first * g.KBtot + kb) * 64 + lane) * P::E : arow_off + (size_t)kb * P::KB;
                af[c] = lda16<P>(g.A, ao * ES);
            }
            else af[c] = P::aload(lds_a + lr * pitch + (kb * P::KB + P::E * lg) * ES, LDS_A);      // (IS_LN)
        }
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const bool live = kb0 + c < kb_hi;              // wave-uniform; out-of-range blocks contribute zeros
            AFrag a = af[c];
            if (!live) { if constexpr (A2) a = P::azero(); else a = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int t = 0; t < TNW; ++t)
                acc[t] = swapped[t] ? P::mma_w(bf[c][t], a, acc[t]) : P::mma_a(a, bf[c][t], acc[t]);
        }
        if (kb0 + CH < kb_hi) load_b(kb0 + CH);
    }
    }   // pass
    DSG_TL_MARK(2);          // main loop issued (operands have landed)

    // The normalised rows go back to global memory only now: a global store issued before the MFMA phase would sit
    // in the same vmcnt queue as the weight loads (stores and loads retire out of order with each other, so the
    // compiler has to wait vmcnt(0), i.e. for the store acknowledgements, before the first fragment is usable).
    if constexpr (IS_LN) {
        if (wr) {
            const int row = tid >> 4, c = tid & 15;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < (g.D >> 6)) *(f32x4*)(g.Xn + (size_t)(m0 + row) * g.D + c * 4 + 64 * i) = v[i];
        }
    }

    // ---- in-workgroup split-K reduction (deterministic order)
    if constexpr (WK > 1) {
        if (wk > 0) {
#pragma unroll
            for (int t = 0; t < TNW; ++t)
                *(f32x4*)(lds_red + ((((wk - 1) * WN + wn) * TNW + t) * 64 + lane) * 4) = acc[t];
        }
        DSG_LDS_BARRIER();
        if (wk == 0) {
#pragma unroll
            for (int w2 = 1; w2 < WK; ++w2)
#pragma unroll
                for (int t = 0; t < TNW; ++t)
                    acc[t] += *(const f32x4*)(lds_red + ((((w2 - 1) * WN + wn) * TNW + t) * 64 + lane) * 4);
        }
    }

    // ---- epilogue (split-K inside the workgroup: the k-slice-0 waves own it)
    if (WK == 1 || wk == 0) {
#pragma unroll
    for (int t = 0; t < TNW; ++t)
        gemm_epilogue_tile<P, EPI>(g, m0, (nt0 + t) * 16, lr, lg, ks, swapped[t], acc[t], ops[t], k1, k2, k3, k4, k5, acc_u[CFG ? t : 0]);
    }
    DSG_TL_MARK(3);          // epilogue stores issued
}

// batched LayerNorm-GEMM: compiled for 3 waves per SIMD (<= 168 VGPRs; at 128 it spills to scratch; requesting the weights
// only after the LayerNorm fits 5 waves per SIMD but measured slower: 370 vs 360 us/step at batch 16), see ln_rows LEAN
template <class P, int EPI, int CH = 8>
__global__ __launch_bounds__(256, (CH > 8 || P::W2) ? 2 : 3) void k_gemm_lean(const GemmArgs g) { DSG_TL_SCOPE(); gemm_body<P, PRO_LN, EPI, 4, 1, 1, true, false, CH>(g); }

// pose head with classifier-free guidance: conditional + unconditional rows per workgroup (GemmArgs::cfgB)
template <class P, int CH = 8>
__global__ __launch_bounds__(256) void k_gemm_cfg(const GemmArgs g) { DSG_TL_SCOPE(); gemm_body<P, PRO_LN, EPI_OUT, 4, 1, 1, false, true, CH>(g); }

template <class P, int PRO, int EPI, int WN, int WK, int TNW, int CH = 8>
__global__ __launch_bounds__(256) void k_gemm(const GemmArgs g) { DSG_TL_SCOPE(); gemm_body<P, PRO, EPI, WN, WK, TNW, false, false, CH>(g); }

// ---------------------------------------------------------------------------------------------------------
// k_loc: per (batch, window, local head).  h = sum_s partial_s + Cframe + TE2[t]; rotary(pos = frame);
//        causal local attention over {previous window, own window} with q = k = v; prepend token; rotary(pos+1)
// ---------------------------------------------------------------------------------------------------------
struct LocArgs {
    const float* partial;   // [KS][Min_pad][D]
    int KS, Min_pad;
    const float* Cf;        // [B][T][D]   window-invariant part of input_process2 (audio, biases, style/seed token)
    const float* TE2;       // [n_te][D]   W2a . time_embed(t)
    const float* TE;        // [n_te][D]   time_embed(t)
    const float* emb1;      // [B][D]      style (+ seed) token embedding
    const StepCtl* ctl;     // sampling: model timestep = ctl->tA ; null -> t_arr
    const int* t_arr;       // per-batch model timestep (forward)
    const float* rcos;      // [T+1][hd/2]
    const float* rsin;
    const unsigned char* mask;   // [mb][T] key mask (1 = keep)
    int mb;
    int nomask;                  // `mask=None` of LocalAttention.forward (local_attention.py:196): nothing is masked but the
                                 // causal future -- the look-back pad keys of window 0 take part with key = value = -1
    unsigned inv_mask_div;       // fastdiv_inv(B * Hl / mb): (b, head) -> mask row
    int B, T, D, Hl, hd, W;   // hd / W must match the kernel's template arguments
    float* X0;              // [M_pad][D] fp32, row = b*(T+1) + 1 + f ; row b*(T+1) = token
    void* X0a;              // same in P::elem (GEMM operand copy)
    int x0a_frag;           // ... stored fragment-major (qk_off over [rows][D]): the STREAM set streams it into the layer-0 QKV; 2 (bf16w2, ROWS): as a hi + lo pair
};

// Tail of k_inloc (LATENCY set: one workgroup per CU, all four SIMDs on the tile).  `rot` holds the rotary-embedded [2W][HD] tile (pad rows = -1).  Phase A: one thread
// per (query, key) pair forms the masked score, the 32 lanes of a query row reduce max / sum with shuffles; phase B:
// one thread per (query, dim pair) forms the attention output and applies the second rotary (position + 1).
template <class P, int HD, int W>
__device__ __forceinline__ void local_attn_tail_valu(const LocArgs& a, float (&rot)[2 * W][HD + 4], float (&sc)[W][2 * W + 2],
                                                int b, int w, int h, const bool (&keep)[(W * 32 + 255) / 256],
                                                const float (&c2)[(W * (HD / 2) + 255) / 256],
                                                const float (&s2)[(W * (HD / 2) + 255) / 256]) {
    typedef typename P::elem elem;
    constexpr int W2 = 2 * W, half = HD / 2, NSI = (W * 32 + 255) / 256, NP2 = W * half, NPO = (NP2 + 255) / 256;
    static_assert(W2 <= 32, "a query row of scores fits in 32 lanes");
    const int tid = threadIdx.x;
    const int f0 = (w - 1) * W;
    const float scale = 1.0f / sqrtf((float)HD);
#pragma unroll
    for (int i = 0; i < NSI; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx >> 5, j = idx & 31;
        const bool inq = q < W, valid = inq && j < W2;
        float sv = -DSG_FLT_MAX;
        if (valid) {
            // (round 5: 16-byte LDS reads -- the rows are 16-byte aligned now, pitch HD + 4 -- a quarter of the LDS instructions of the 4-byte loop;
            //  the products are added in the same order e = 0 .. HD - 1.  The same treatment of the P V loop -- 4 dims per thread, a quarter of the
            //  threads -- was SLOWER: k_loc 13.3 -> 15.6 us at 64 clips)
            float d = 0.f;
            const f32x4* qa = (const f32x4*)&rot[W + q][0];
            const f32x4* kb = (const f32x4*)&rot[j][0];
#pragma unroll
            for (int c = 0; c < HD / 4; ++c) {
                const f32x4 va = qa[c], vb = kb[c];
                d += va[0] * vb[0]; d += va[1] * vb[1]; d += va[2] * vb[2]; d += va[3] * vb[3];
            }
            const int fq = w * W + q, fk = f0 + j;
            const bool masked = ((fk >= 0) && (fq < fk)) || !keep[i];   // causal | key mask (pads are masked keys)
            sv = masked ? -DSG_FLT_MAX : d * scale;
        }
        float mx = sv;                                   // every lane takes part in the shuffles
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16));
        const float pv = valid ? expf(sv - mx) : 0.f;
        float sum = pv;
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 8); sum += __shfl_xor(sum, 16);
        if (valid) sc[q][j] = pv / sum;
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(2);          // local attention: scores + softmax
    const int ntok = a.T + 1, col0 = h * HD;
    // Round 5: the W x HD output tile is staged in LDS and leaves in 16-byte stores -- the rows used to go out as 4-byte (fp32) and
    // 2-byte (bf16) element stores, and an uncached narrow store is a fabric write of its own (4096 workgroups x 704 of them at 64
    // clips: k_loc 19.4 us of a 287 us step).  Same values, same rounding point.
    __shared__ __attribute__((aligned(16))) float ot[W][HD + 4];
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
        const int p = tid + 256 * i;
        if (p < NP2) {
            const int q = p / half, dd = p % half;
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int j = 0; j < W2; ++j) { const float pj = sc[q][j]; lo += pj * rot[j][dd]; hi += pj * rot[j][dd + half]; }
            ot[q][dd] = lo * c2[i] - hi * s2[i]; ot[q][dd + half] = hi * c2[i] + lo * s2[i];
        }
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(3);          // P V + second rotary -> staged tile
    constexpr int C4 = HD / 4, CE = HD / P::E;       // 16-byte chunks per row: fp32 rows / rows in the GEMM type
    for (int p = tid; p < W * C4; p += 256) {
        const int q = p / C4, c = p - q * C4;
        *(f32x4*)(a.X0 + (imul24(b * ntok + 1 + w * W + q, a.D) + (unsigned)(col0 + 4 * c))) = *(const f32x4*)&ot[q][4 * c];
    }
    for (int p = tid; p < W * CE; p += 256) {
        const int q = p / CE, c = p - q * CE, row = b * ntok + 1 + w * W + q;
        elem* dst = (elem*)a.X0a + (a.x0a_frag ? (unsigned)qk_off<P>(row, col0 + P::E * c, a.D / P::KB) : imul24(row, a.D) + (unsigned)(col0 + P::E * c));
        if constexpr (P::E == 4) {
            *(f32x4*)dst = *(const f32x4*)&ot[q][4 * c];
        } else {
            if constexpr (P::W2) {
                if (a.x0a_frag == 2) {       // bf16w2 in the ROWS set: hi + lo images, like every other A operand k_clip_attn reads
                    const size_t off = (size_t)qk_off<P>(row, col0 + P::E * c, a.D / P::KB);
                    P::store4_afrag((elem*)a.X0a, off, *(const f32x4*)&ot[q][8 * c]);
                    P::store4_afrag((elem*)a.X0a, off + 4, *(const f32x4*)&ot[q][8 * c + 4]);
                    continue;
                }
            }
            P::store4(dst, *(const f32x4*)&ot[q][8 * c]);
            P::store4(dst + 4, *(const f32x4*)&ot[q][8 * c + 4]);
        }
    }
}

// Tail of k_loc (every other set).  Round 6: the scores and the P V product of the (head, window) tile on fp32 MATRIX instructions (v_mfma_f32_16x16x4_f32),
// by ONE wave of the workgroup -- which is what lets the kernel run as one wave per item at large batches (dsg_hip.cpp: DSG_LOC1_DISPATCH); k_inloc keeps the
// VALU form above: alone on its CU at batch 1, four SIMDs on the tile beat 32 fp32 MFMAs on one (106.1 vs 106.8 us per step, profiles/r06_cm_*).
// The VALU form read the rotary tile out of LDS once per (query, key) pair and per (query, dim pair) -- ~108 KB of LDS reads per workgroup against a 3 KB
// tile: phase A alone was 1.8 of the kernel's 5.4 us at 16 clips and the kernel 13.3 us at 64 (4096 workgroups, two rounds of 8 per CU).  Here a lane reads
// one float per operand and MFMA: S^T[key][query] = K . Q^T over the head dim (A = key rows, B = query rows of the tile), softmax as in k_attn (a lane owns a
// query column: 8 keys in registers, two shuffle steps), O^T[dim][query] = V^T . P^T with the P values a lane already holds as its B operand (the k-slot of
// step s is key 4 lg + s: V^T is gathered in that order), raw O rows -> LDS, then the (query, dim pair) threads apply the second rotary in place.  The worker
// wave rotates with the block index (the workgroups of a CU do not all put it on the same SIMD).  Same masks, same pad / causal semantics; fp32 throughout.
template <class P, int HD, int W, int NT = 256>
__device__ __forceinline__ void local_attn_tail(const LocArgs& a, float (&rot)[2 * W][HD + 4], float (&sc)[W][2 * W + 2],
                                                int b, int w, int h, const bool (&keyok)[2][4],
                                                const float (&c2)[(W * (HD / 2) + NT - 1) / NT],
                                                const float (&s2)[(W * (HD / 2) + NT - 1) / NT]) {
    typedef typename P::elem elem;
    constexpr int W2 = 2 * W, half = HD / 2, NP2 = W * half, NPO = (NP2 + NT - 1) / NT;
    constexpr int NKT = 2, ND = (HD + 15) / 16, KS = HD / 4;
    static_assert(W2 <= 32 && W <= 16 && HD % 4 == 0, "one query tile, two key tiles");
    (void)sc;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
    const int f0 = (w - 1) * W;
    const float scale = 1.0f / sqrtf((float)HD);
    __shared__ __attribute__((aligned(16))) float ot[W][HD + 4];
    if (NT == 64 || wave_id() == ((b + w + h) & (NT / 64 - 1))) {      // (NT = 64: the workgroup IS one wave)
        const float* qrow = &rot[W + min(lr, W - 1)][0];
        f32x4 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const float* krow = &rot[min(16 * kt + lr, W2 - 1)][0];
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[4 * ks + lg], qrow[4 * ks + lg], s[kt], 0, 0, 0);   // D[key 4lg+r][query lr]
        }
        const int fq = w * W + lr;
        float mx = -DSG_FLT_MAX;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fk = f0 + 16 * kt + 4 * lg + r;
                const bool masked = ((fk >= 0) && (fq < fk)) || !keyok[kt][r];   // causal | key mask (pads are masked keys)
                const float v = masked ? -DSG_FLT_MAX : s[kt][r] * scale;
                s[kt][r] = v;
                mx = fmaxf(mx, (16 * kt + 4 * lg + r) < W2 ? v : -DSG_FLT_MAX);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = (16 * kt + 4 * lg + r) < W2 ? expf(s[kt][r] - mx) : 0.f;
                s[kt][r] = pv;
                sum += pv;
            }
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        DSG_TL_MARK(2);          // local attention: scores + softmax
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int dim = min(16 * dt + lr, HD - 1);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(rot[min(16 * kt + 4 * lg + st, W2 - 1)][dim], s[kt][st], o, 0, 0, 0);   // D[dim 4lg+r][query lr]
            if (lr < W && 16 * dt + 4 * lg < HD) {
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
                *(f32x4*)&ot[lr][16 * dt + 4 * lg] = y;
            }
        }
    }
    DSG_LDS_BARRIER();
    const int ntok = a.T + 1, col0 = h * HD;
    // second rotary (position + 1) in place: thread (query, dim pair) owns both elements it touches
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
        const int p = tid + NT * i;
        if (p < NP2) {
            const int q = p / half, dd = p % half;
            const float lo = ot[q][dd], hi = ot[q][dd + half];
            ot[q][dd] = lo * c2[i] - hi * s2[i]; ot[q][dd + half] = hi * c2[i] + lo * s2[i];
        }
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(3);          // P V + second rotary -> staged tile
    constexpr int C4 = HD / 4, CE = HD / P::E;       // 16-byte chunks per row: fp32 rows / rows in the GEMM type
    for (int p = tid; p < W * C4; p += NT) {
        const int q = p / C4, c = p - q * C4;
        *(f32x4*)(a.X0 + (imul24(b * ntok + 1 + w * W + q, a.D) + (unsigned)(col0 + 4 * c))) = *(const f32x4*)&ot[q][4 * c];
    }
    for (int p = tid; p < W * CE; p += NT) {
        const int q = p / CE, c = p - q * CE, row = b * ntok + 1 + w * W + q;
        elem* dst = (elem*)a.X0a + (a.x0a_frag ? (unsigned)qk_off<P>(row, col0 + P::E * c, a.D / P::KB) : imul24(row, a.D) + (unsigned)(col0 + P::E * c));
        if constexpr (P::E == 4) {
            *(f32x4*)dst = *(const f32x4*)&ot[q][4 * c];
        } else {
            if constexpr (P::W2) {
                if (a.x0a_frag == 2) {       // bf16w2 in the ROWS set: hi + lo images, like every other A operand k_clip_attn reads
                    const size_t off = (size_t)qk_off<P>(row, col0 + P::E * c, a.D / P::KB);
                    P::store4_afrag((elem*)a.X0a, off, *(const f32x4*)&ot[q][8 * c]);
                    P::store4_afrag((elem*)a.X0a, off + 4, *(const f32x4*)&ot[q][8 * c + 4]);
                    continue;
                }
            }
            P::store4(dst, *(const f32x4*)&ot[q][8 * c]);
            P::store4(dst + 4, *(const f32x4*)&ot[q][8 * c + 4]);
        }
    }
}

template <class P, int HD, int W, int NT = 256>
__device__ __forceinline__ void loc_body(const LocArgs& a, int h, int w, int b) {
    typedef typename P::elem elem;
    constexpr int W2 = 2 * W, half = HD / 2, NP1 = W2 * half, NPI = (NP1 + NT - 1) / NT;
    constexpr int NP2 = W * half, NPO = (NP2 + NT - 1) / NT, MAXKS = 9;
    __shared__ __attribute__((aligned(16))) float rot[W2][HD + 4];
    __shared__ float sc[W][W2 + 2];
    const int tid = threadIdx.x;
    const int* tp = a.ctl ? &a.ctl->tA : a.t_arr + b;      // select the ADDRESS, then one unconditional load
    const int t = ldw<P>(tp);
    const int col0 = h * HD, ntok = a.T + 1, f0 = (w - 1) * W;

    // ---- every global load of the block is issued up front, unconditionally (clamped indices, selects afterwards)
    float lo[NPI], hi[NPI], c1[NPI], s1[NPI];
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = min(tid + NT * i, NP1 - 1);
        const int r = p / half, dd = p % half, f = max(f0 + r, 0);
        const unsigned base = imul24(b * a.T + f, a.D) + (unsigned)(col0 + dd);      // (32-bit element offsets: imul24)
        lo[i] = a.Cf[base];
        hi[i] = a.Cf[base + half];
        c1[i] = a.rcos[f * half + dd]; s1[i] = a.rsin[f * half + dd];
        if (a.KS == 1) {             // (uniform; the streamed pose embedding of the batched sets leaves ONE slab: without this, eight clamped
                                     //  re-reads of it per element -- uncached memory, every one a trip to the memory side)
            lo[i] += a.partial[base]; hi[i] += a.partial[base + half];
        } else {
#pragma unroll
            for (int s = 0; s < MAXKS; ++s) {
                const float wgt = s < a.KS ? 1.f : 0.f;
                const unsigned pb = imul24(min(s, a.KS - 1) * a.Min_pad, a.D) + base;
                lo[i] += wgt * a.partial[pb]; hi[i] += wgt * a.partial[pb + half];
            }
        }
    }
    float c2[NPO], s2[NPO];
#pragma unroll
    for (int i = 0; i < NPO; ++i) {
        const int p = min(tid + NT * i, NP2 - 1);
        const int pos = w * W + p / half + 1;
        c2[i] = a.rcos[pos * half + p % half]; s2[i] = a.rsin[pos * half + p % half];
    }
    const int mrow = fdiv(b * a.Hl + h, a.inv_mask_div);
    bool keyok[2][4];                                    // the 8 keys of this lane in the S^T layout of local_attn_tail: key j = 16 kt + 4 lg + r
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * kt + 4 * ((tid & 63) >> 4) + r, fk = f0 + j;
            const unsigned char mk = a.mask[(unsigned)(mrow * a.T + min(max(fk, 0), a.T - 1))];
            keyok[kt][r] = ((int)(j < W2) & ((int)(a.nomask != 0) | ((int)(fk >= 0) & (int)(mk != 0)))) != 0;   // bitwise: keeps the load unconditional
        }
    const int tc = min(tid, HD - 1);
    float tokv = a.emb1[(unsigned)(b * a.D + col0 + tc)];
    // the loads that depend on the model timestep go out last (t itself was requested first)
    const float* te2 = a.TE2 + (size_t)t * a.D + col0;        // (t is wave-uniform: scalar arithmetic)
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = min(tid + NT * i, NP1 - 1), dd = p % half;
        lo[i] += te2[dd];
        hi[i] += te2[dd + half];
    }
    tokv += a.TE[(size_t)t * a.D + col0 + tc];
    DSG_LOADS_ISSUED();
    DSG_TL_MARK(0);          // k_loc: every load requested
    if (w == 0 && tid < HD) {                              // token row (position 0: rotary is the identity)
        a.X0[(size_t)(b * ntok) * a.D + col0 + tid] = tokv;
        if (P::W2 && a.x0a_frag == 2) P::store1_afrag((elem*)a.X0a, (size_t)qk_off<P>(b * ntok, col0 + tid, a.D / P::KB), tokv);
        else ((elem*)a.X0a)[a.x0a_frag ? (size_t)qk_off<P>(b * ntok, col0 + tid, a.D / P::KB) : (size_t)(b * ntok) * a.D + col0 + tid] = P::cvt(tokv);
    }
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = tid + NT * i;
        if (p < NP1) {
            const int r = p / half, dd = p % half, f = f0 + r;
            // rotary (rotary.py:20-27); rows before the sequence start are look_around's pad_value -1
            rot[r][dd] = f >= 0 ? lo[i] * c1[i] - hi[i] * s1[i] : -1.0f;
            rot[r][dd + half] = f >= 0 ? hi[i] * c1[i] + lo[i] * s1[i] : -1.0f;
        }
    }
    DSG_LDS_BARRIER();
    DSG_TL_MARK(1);          // rotary rows in LDS (the loads have landed)
    local_attn_tail<P, HD, W, NT>(a, rot, sc, b, w, h, keyok, c2, s2);
}

template <class P, int HD, int W, int NT = 256>      // NT = 128 (round 6): two waves per (head, window, clip) at large batches -- see dsg_hip.cpp: DSG_LOC1_DISPATCH
__global__ __launch_bounds__(NT) void k_loc(const LocArgs a) {
    DSG_TL_SCOPE();
    preload_kernargs(a);
    loc_body<P, HD, W, NT>(a, blockIdx.x, blockIdx.y, blockIdx.z);          // grid (local heads, windows, batch)
}

// ---------------------------------------------------------------------------------------------------------
// k_attn: one wavefront per (batch, head, 16-query tile); whole softmax(QK^T/sqrt(hd))V in registers, no LDS.
//   S^T = K_tile . Q^T  -> lane (query = lane&15) holds 4 keys per key tile: row max/sum = in-lane + 2 shuffles.
//   O^T = V^T_tile . P^T -> the P values a lane already holds ARE its B fragment (k-permutation: the V^T fragment
//   is gathered with the same key order), so P never leaves registers.
// ---------------------------------------------------------------------------------------------------------
struct AttnArgs {
    const void* q; const void* k; const void* vt;   // [B][H][Tp][hd], [B][H][Tp][hd], [B][H][hd][Tp]
    void* out;                                       // [M_pad][D] P::elem
    int B, H, ntok, Tp, D;
};

template <class P, int HD, int NKT>
__global__ __launch_bounds__(64) void k_attn(const AttnArgs a) {
    DSG_TL_SCOPE();
    typedef typename P::elem elem;
    constexpr int KD = HD / P::KB;                   // k-blocks over the head dim
    preload_kernargs(a);
    const int lane = threadIdx.x, lr = lane & 15, lg = lane >> 4;
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;          // grid (query tiles, heads, batch)
    const size_t bh = (size_t)b * a.H + h;
    const elem* Q = (const elem*)a.q + bh * a.Tp * HD;
    const elem* K = (const elem*)a.k + bh * a.Tp * HD;
    const elem* VT = (const elem*)a.vt + bh * HD * a.Tp;

    // ---- operand fragments.  When everything fits in registers (PRELOAD) all of Q, K, V^T is requested before the
    //      first MFMA (one memory round trip); the large DSG+ shapes would spill, so they stream K per key tile and
    //      fetch V^T while the softmax runs.
    constexpr int ND = HD / 16;                      // 16-dim output tiles
    constexpr int NVF = P::E == 4 ? NKT : NKT / 2;   // PV k-blocks
    constexpr bool PRELOAD = (KD + NKT * KD + ND * NVF) * 4 <= 360;
    f32x4 qf[KD];
#pragma unroll
    for (int kb = 0; kb < KD; ++kb) qf[kb] = *(const f32x4*)(Q + (size_t)((qt * KD + kb) * 64 + lane) * P::E);       // fragment-major (qk_off)
    auto load_v = [&](f32x4 (&vfr)[ND][NVF]) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
#pragma unroll
            for (int kb = 0; kb < NVF; ++kb) vfr[dt][kb] = *(const f32x4*)(VT + (size_t)((dt * NVF + kb) * 64 + lane) * P::E);   // fragment-major (vt_off)
        }
    };
    f32x4 vfr[ND][NVF];
    f32x4 s[NKT];
    if constexpr (PRELOAD) {
        f32x4 kf[NKT][KD];
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) kf[nt][kb] = *(const f32x4*)(K + (size_t)((nt * KD + kb) * 64 + lane) * P::E);
        load_v(vfr);
        DSG_LOADS_ISSUED();
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) s[nt] = P::mma(kf[nt][kb], qf[kb], s[nt]);   // D[key = 4*lg + r][query = lr]
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) {
            s[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KD; ++kb) {
                const f32x4 kf = *(const f32x4*)(K + (size_t)((nt * KD + kb) * 64 + lane) * P::E);
                s[nt] = P::mma(kf, qf[kb], s[nt]);
            }
        }
        load_v(vfr);
    }
    const float scale = 1.0f / sqrtf((float)HD);
    float mx = -DSG_FLT_MAX;
#pragma unroll
    for (int nt = 0; nt < NKT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = nt * 16 + 4 * lg + r;
            const float v = key < a.ntok ? s[nt][r] * scale : -DSG_FLT_MAX;
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
            const float p = key < a.ntok ? P::exp_sm(s[nt][r] - mx) : 0.f;
            s[nt][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;

    const int q = qt * 16 + lr;
    f32x4 pfr[NVF];                                  // P^T fragments: exactly the values this lane already holds
#pragma unroll
    for (int kb = 0; kb < NVF; ++kb) {
        if constexpr (P::E == 4) {
            pfr[kb] = s[kb];
        } else {
            static_assert(P::E == 4 || (NKT % 2) == 0, "bf16 pairs key tiles");
            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
            u16x8 pp;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = f2bf(s[2 * kb][e]); pp[4 + e] = f2bf(s[2 * kb + 1][e]); }
            pfr[kb] = __builtin_bit_cast(f32x4, pp);
        }
    }
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NVF; ++kb) o = P::mma(vfr[dt][kb], pfr[kb], o);     // D[dim = 4*lg + r][query = lr]
        if (q < a.ntok) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = o[e] * inv;
            // fragment-major rows (the next GEMM's A operand; PBF16W2: hi + lo)
            P::store4_afrag((elem*)a.out, (size_t)qk_off<P>(b * a.ntok + q, h * HD + dt * 16 + 4 * lg, a.D / P::KB), y);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// state layout conversion + sampler entry/exit arithmetic
// ---------------------------------------------------------------------------------------------------------
struct XInArgs {
    const float* x;         // optional [B][J][T] explicit start (dsg_forward input, or user `noise`)
    const float* init;      // optional init_image [B][J][T]  (q_sample start, gaussian_diffusion.py:706-713)
    int do_q;               // skip_timesteps / init_image: x_start = qa * init (0 if null) + qb * noise
    float qa, qb;
    int use_philox;         // draw noise 0 from the Philox stream when x == null
    NoiseKey nkey; unsigned draw;
    int B, J, Jp, Jq, T;
    float* xs32; void* xsA;
    int dupB;               // classifier-free guidance: batch element b is also written to row b + dupB (its unconditional twin)
    int xs_frag;            // xsA fragment-major (see GemmArgs::xs_frag)
};
template <class P>
__global__ void k_x_in(const XInArgs a) {
    typedef typename P::elem elem;
    const size_t n = (size_t)a.B * a.T * (a.Jp / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j0 = (int)(i % (a.Jp / 4)) * 4;
        const size_t bf = i / (a.Jp / 4);
        const int f = (int)(bf % a.T), b = (int)(bf / a.T);
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (j0 < a.J) {
            if (a.x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (j0 + e < a.J) z[e] = a.x[((size_t)b * a.J + j0 + e) * a.T + f];
            } else if (a.use_philox) {
                z = philox_normal4((unsigned)((((size_t)b * a.T + f) * a.Jq + j0) >> 2), a.draw, a.nkey);
            }
            if (a.do_q) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j0 + e < a.J)
                        z[e] = (a.init ? a.qa * a.init[((size_t)b * a.J + j0 + e) * a.T + f] : 0.f) + a.qb * z[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) if (j0 + e >= a.J) z[e] = 0.f;
        }
        *(f32x4*)(a.xs32 + ((size_t)b * a.T + f) * a.Jp + j0) = z;
        if (a.xsA) P::store4((elem*)a.xsA + xs_off<P>(b * a.T + f, j0, a.Jp, a.xs_frag), z);
        if (a.dupB > 0) {
            *(f32x4*)(a.xs32 + ((size_t)(b + a.dupB) * a.T + f) * a.Jp + j0) = z;
            if (a.xsA) P::store4((elem*)a.xsA + xs_off<P>((b + a.dupB) * a.T + f, j0, a.Jp, a.xs_frag), z);
        }
    }
}
// the framework's noise stream as a tensor: out [B][J][T] (the reference's [B, J, 1, T]) = draw `draw` of (seed, stream) --
// exactly what the fused sampler epilogue consumes for that draw index (generic sampling loop, tests)
__global__ void k_noise(float* out, int B, int J, int Jq, int T, NoiseKey key, unsigned draw) {
    const size_t n = (size_t)B * T * (Jq / 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int jq = (int)(i % (Jq / 4));
        const size_t bf = i / (Jq / 4);
        const int f = (int)(bf % T), b = (int)(bf / T);
        const f32x4 z = philox_normal4((unsigned)((((size_t)b * T + f) * Jq + 4 * jq) >> 2), draw, key);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * jq + e < J) out[((size_t)b * J + 4 * jq + e) * T + f] = z[e];
    }
}
__global__ void k_x_out(const float* xs32, float* out, int B, int J, int Jp, int T) {
    const size_t n = (size_t)B * J * T;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % T); const size_t bj = i / T;
        const int j = (int)(bj % J), b = (int)(bj / J);
        out[i] = xs32[((size_t)b * T + f) * Jp + j];
    }
}
// Self-check of the fence-free hand-off (dsg_hip.cpp: uc_selfcheck): `buf` is uncached device memory; the two kernels run as
// dependent AQL packets WITHOUT acquire / release, 64 times over.  Writer workgroup b fills chunk b with a pattern of the
// iteration; reader workgroup b verifies chunk b + 1 -- written on another XCD -- and counts stale words.  The iteration
// words follow the StepCtl protocol (no kernel reads what it writes; an extra workgroup advances the other kernel's word).
struct UcProbeArgs { unsigned* buf; int* ctl; unsigned* err; int n_wg; };
__device__ __forceinline__ unsigned uc_probe_pattern(unsigned i, unsigned it) { return (i * 2654435761u) ^ (it * 0x9E3779B9u) ^ 0x5bd1e995u; }
__global__ __launch_bounds__(256) void k_uc_probe_w(const UcProbeArgs a) {
    const int it = ldw<PF32>(&a.ctl[0]);
    const int b = blockIdx.x;
    if (b == a.n_wg) { if (threadIdx.x == 0) a.ctl[1] = it; return; }
    const unsigned i = (unsigned)b * 256u + threadIdx.x;
    a.buf[i] = uc_probe_pattern(i, (unsigned)it);
}
__global__ __launch_bounds__(256) void k_uc_probe_r(const UcProbeArgs a) {
    const int it = ldw<PF32>(&a.ctl[1]);
    const int b = blockIdx.x;
    if (b == a.n_wg) { if (threadIdx.x == 0) { a.ctl[0] = it + 1; a.err[1] = (unsigned)(it + 1); } return; }
    const unsigned i = (unsigned)((b + 1) % a.n_wg) * 256u + threadIdx.x;
    if (a.buf[i] != uc_probe_pattern(i, (unsigned)it)) __hip_atomic_fetch_add(&a.err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_ctr_set(int* ctr, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr = v; }
__global__ void k_ctr_inc(int* ctr) { if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += 1; }

// standalone fused sampler arithmetic on caller tensors (any layout, per-batch-element scalars):
//   out = a[b]*p + c[b]*q + s[b]*z      (q_sample, xstart-from-eps, posterior mean + noise)
__global__ void k_axpbypcz(float* out, const float* p, const float* q, const float* z, const float* a,
                           const float* c, const float* s, int B, size_t per) {
    const size_t n = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        float v = a[b] * p[i];
        if (q) v += c[b] * q[i];
        if (z) v += s[b] * z[i];
        out[i] = v;
    }
}
// DDIM update in the reference's evaluation order; coef = [B][5] {sqrt_recip, sqrt_recipm1, sqrt(abar_prev), dir, nz*sigma}
__global__ void k_ddim_step(float* out, const float* x0, const float* xt, const float* z, const float* coef, int B,
                            size_t per) {
    const size_t n = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float* c = coef + 5 * (i / per);
        const float eps = (c[0] * xt[i] - x0[i]) / c[1];
        out[i] = (x0[i] * c[2] + c[3] * eps) + c[4] * (z ? z[i] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// set-up kernels (once per checkpoint / once per window; fp32 data, double accumulation, one thread per output)
// ---------------------------------------------------------------------------------------------------------
// C[m][n] = act( sum_k A[m*sam + k*sak] * Bm[n*sbn + k*sbk] + bias[n] + add[(m/add_div)*sadd + n] )
struct MMArgs {
    float* C; int ldc;
    const float* A; long long sam, sak;
    const float* Bm; long long sbn, sbk;
    const float* bias; const float* add; long long sadd; int add_div;   // add[(m / add_div)*sadd + n]
    int M, N, K; int act;    // act: 0 none, 1 SiLU
};
__global__ void k_mm_naive(const MMArgs a) {
    const size_t n = (size_t)a.M * a.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int nn = (int)(i % a.N), m = (int)(i / a.N);
        double acc = 0.0;
        const float* pa = a.A + (long long)m * a.sam;
        const float* pb = a.Bm + (long long)nn * a.sbn;
        for (int k = 0; k < a.K; ++k) acc += (double)pa[(long long)k * a.sak] * (double)pb[(long long)k * a.sbk];
        if (a.bias) acc += (double)a.bias[nn];
        if (a.add) acc += (double)a.add[(long long)(m / a.add_div) * a.sadd + nn];
        float v = (float)acc;
        if (a.act == 1) v = v / (1.0f + expf(-v));
        a.C[(size_t)m * a.ldc + nn] = v;
    }
}
// the same product for LONG reductions (the seed-pose embedding: K = J * n_seed = 9128; the audio feature map: K = 1133 for every frame of the
// window): one WAVE per (row, 4 output columns), the 64 lanes stride over k, double accumulation, a fixed-order butterfly over the lanes.
// (Rounds 2-5 used one WORKGROUP per output with a __syncthreads tree: 130-200 us of conditioning per window at batch 16, which the host waits
// for before the first packet of the step loop -- 2 % of a 50-step DDIM window; round 6, tools/prof_host.py.)
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)u, m), hi = (unsigned)__shfl_xor((int)(unsigned)(u >> 32), m);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__global__ __launch_bounds__(256) void k_mm_wave(const MMArgs a) {
    const int lane = threadIdx.x & 63, n4s = (a.N + 3) / 4;
    const size_t nw = (size_t)a.M * n4s;
    for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (size_t)gridDim.x * 4) {
        const int n4 = (int)(w % n4s), m = (int)(w / n4s);
        const float* pa = a.A + (long long)m * a.sam;
        const float* pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pb[j] = a.Bm + (long long)min(4 * n4 + j, a.N - 1) * a.sbn;      // clamped: computed and dropped
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int k = lane; k < a.K; k += 64) {
            const double av = (double)pa[(long long)k * a.sak];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += av * (double)pb[j][(long long)k * a.sbk];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[j] += shfl_xor_f64(acc[j], o);
        if (lane < 4 && 4 * n4 + lane < a.N) {
            const int nn = 4 * n4 + lane;
            double r = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
            if (a.bias) r += (double)a.bias[nn];
            if (a.add) r += (double)a.add[(long long)(m / a.add_div) * a.sadd + nn];
            float v = (float)r;
            if (a.act == 1) v = v / (1.0f + expf(-v));
            a.C[(size_t)m * a.ldc + nn] = v;
        }
    }
}
// pack W[N][K] fp32 (row pitch ldw, column offset folded into the pointer) into MFMA fragment order
// [NT][KBtot][64][E] with zero padding for n >= N, k >= K
template <class P>
__global__ void k_pack_w(void* dst, const float* W, long long ldw, int N, int K, int NT, int KBtot) {
    typedef typename P::elem elem;
    const size_t n = (size_t)NT * KBtot * 64 * P::E;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % P::E); size_t r = i / P::E;
        const int lane = (int)(r % 64); r /= 64;
        const int kb = (int)(r % KBtot); const int nt = (int)(r / KBtot);
        const int nn = nt * 16 + (lane & 15);
        const int k = kb * P::KB + P::E * (lane >> 4) + j;
        const float v = (nn < N && k < K) ? W[(long long)nn * ldw + k] : 0.f;
        if constexpr (P::W2) {           // fragment f -> blocks 2f (hi) and 2f + 1 (lo) of 64 lanes x E elements
            const size_t frag = i / (64 * P::E), in = i % (64 * P::E);
            const elem hi = P::cvt(v);
            ((elem*)dst)[(2 * frag) * 64 * P::E + in] = hi;
            ((elem*)dst)[(2 * frag + 1) * 64 * P::E + in] = P::cvt(v - P::up(hi));
        } else {
            ((elem*)dst)[i] = P::cvt(v);
        }
    }
}
__global__ void k_fill_f32(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// dst[b][S + f][n] layout helpers are done with k_mm_naive strides; this copies / pads fp32 vectors
__global__ void k_copy_pad(float* dst, const float* src, int n_src, int n_dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += gridDim.x * blockDim.x)
        dst[i] = i < n_src ? src[i] : 0.f;
}

}  // namespace dsg
