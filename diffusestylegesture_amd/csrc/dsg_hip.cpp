// dsg_hip.cpp -- host side of libdsg_hip.so (C ABI in include/dsg.h): weight ingestion / repacking, per-window
// conditioning, the per-step launch sequence, hipGraph capture of the step loop, and the sampler entry points.
// See dsg_kernels.h for the kernels and the reference file:line each one replaces.
#include "dsg_fused.h"
#include "dsg_batched.h"
#include "dsg_stream.h"
#include "../../include/dsg.h"
#include "dsg_aql.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

using namespace dsg;

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(DSG_E_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_) + " @" +        \
                                           std::to_string(__LINE__));                                    \
    } while (0)
#define CHK(expr)                  \
    do {                           \
        int r_ = (expr);           \
        if (r_ != 0) return r_;    \
    } while (0)

extern "C" const char* dsg_last_error(void) { return g_err.c_str(); }
void dsg_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }      // for the other translation units of the library (dsg_bvh.cpp)
extern "C" int dsg_version(void) { return DSG_VERSION; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int rup(int a, int b) { return cdiv(a, b) * b; }

static bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

// ---------------------------------------------------------------------------------------------------------
// schedule tables (float64), GaussianDiffusion.__init__ (main/diffusion/gaussian_diffusion.py:161-198)
// ---------------------------------------------------------------------------------------------------------
struct Sched {
    int n = 0;
    std::vector<double> betas, ac, acp, sqrt_ac, sqrt_1mac, sqrt_recip, sqrt_recipm1, pvar, plogvar, coef1, coef2;
    std::vector<int> tmap;
};
static int build_sched(const double* betas, int n, Sched& s) {
    if (n <= 0) return fail(DSG_E_INVALID, "schedule: n <= 0");
    s.n = n;
    s.betas.assign(betas, betas + n);
    for (int i = 0; i < n; ++i)
        if (!(betas[i] > 0.0 && betas[i] <= 1.0)) return fail(DSG_E_INVALID, "schedule: betas must be in (0, 1]");
    auto rs = [&](std::vector<double>& v) { v.resize(n); };
    rs(s.ac); rs(s.acp); rs(s.sqrt_ac); rs(s.sqrt_1mac); rs(s.sqrt_recip); rs(s.sqrt_recipm1); rs(s.pvar);
    rs(s.plogvar); rs(s.coef1); rs(s.coef2);
    double c = 1.0;
    for (int i = 0; i < n; ++i) { c *= (1.0 - betas[i]); s.ac[i] = c; }
    for (int i = 0; i < n; ++i) s.acp[i] = i == 0 ? 1.0 : s.ac[i - 1];
    for (int i = 0; i < n; ++i) {
        s.sqrt_ac[i] = std::sqrt(s.ac[i]);
        s.sqrt_1mac[i] = std::sqrt(1.0 - s.ac[i]);
        s.sqrt_recip[i] = std::sqrt(1.0 / s.ac[i]);
        s.sqrt_recipm1[i] = std::sqrt(1.0 / s.ac[i] - 1);
        s.pvar[i] = betas[i] * (1.0 - s.acp[i]) / (1.0 - s.ac[i]);
        s.coef1[i] = betas[i] * std::sqrt(s.acp[i]) / (1.0 - s.ac[i]);
        s.coef2[i] = (1.0 - s.acp[i]) * std::sqrt(1.0 - betas[i]) / (1.0 - s.ac[i]);
    }
    for (int i = 0; i < n; ++i) s.plogvar[i] = std::log(n > 1 ? s.pvar[i == 0 ? 1 : i] : s.pvar[0]);
    return 0;
}
extern "C" int dsg_schedule_tables(const double* betas, int n, double* out) {
    Sched s;
    CHK(build_sched(betas, n, s));
    const std::vector<double>* t[11] = {&s.betas, &s.ac, &s.acp, &s.sqrt_ac, &s.sqrt_1mac, &s.sqrt_recip,
                                        &s.sqrt_recipm1, &s.pvar, &s.plogvar, &s.coef1, &s.coef2};
    for (int k = 0; k < 11; ++k) memcpy(out + (size_t)k * n, t[k]->data(), sizeof(double) * n);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------------
struct RawT { float* d = nullptr; std::vector<int64_t> shape; size_t n = 0; };
struct Layer {
    void *Wqkv = nullptr, *Wo = nullptr, *W1 = nullptr, *W2 = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *g1 = nullptr, *be1 = nullptr, *g2 = nullptr,
          *be2 = nullptr;
};

// Everything derived from the checkpoint alone (raw tensors, packed weights, tables): owned jointly by a handle and its
// clones (dsg_clone) -- several sampling lanes of one GPU read ONE copy of the weights, which therefore stays resident in the
// XCDs' L2 slices no matter how many clips are in flight.
struct SharedWeights {
    std::vector<void*> allocs;
    ~SharedWeights() { for (void* p : allocs) (void)hipFree(p); }
};

struct dsg_handle {
    dsg_config cfg;
    int prec = 0, es = 4, kbk = 16;      // element size / k-block of the precision policy
    int J, T, S, D, As, A, W, L, H, hd, ff, Hl, hdl, ntok, Tp, Jp, Jq, Bmax, Ta, n_te;
    int KSin = 4;                        // split-K of the pose-embedding GEMM: one split per 256 pose features
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    std::vector<void*> allocs;                       // per-lane buffers (state, activations, conditioning)
    std::vector<void*> allocs_uc;                    // ... those in uncached memory: returned to the process-wide pool, never to hipFree
    std::shared_ptr<SharedWeights> shared;           // checkpoint-derived buffers (see SharedWeights)
    bool alloc_shared = false;                       // dalloc target: true while loading / finalizing weights
    bool is_clone = false;
    std::map<std::string, RawT> raw;
    bool finalized = false, cond_set = false;
    int condB = 0;
    // weights
    void* Wp_in = nullptr; void* Wp_out = nullptr; float* b_out = nullptr;
    std::vector<Layer> layers;
    float *TE = nullptr, *TE2 = nullptr, *rcos = nullptr, *rsin = nullptr, *cbase = nullptr, *zero_bias = nullptr;
    // window conditioning
    float *emb1 = nullptr, *Cf = nullptr, *enc = nullptr, *cvec = nullptr;
    float *c_style = nullptr, *c_seed = nullptr, *c_audio = nullptr, *c_seed_last = nullptr;
    int seed_last_B = 0;                 // batch of the last dsg_set_seed_last (variant 5)
    unsigned char* mask = nullptr; int mb = 1; int nomask = 0;
    // classifier-free guidance (dsg_set_window_cond_cfg): cfgB user batch elements + their unconditional twins = condB rows
    int cfgB = 0; float* cfg_scale = nullptr;
    int last_path = -1;                  // submission path of the last dsg_sample: 0 HIP launches, 1 AQL packets, 2 hipGraph replay
    bool last_nofence = false;           // ... and whether its packets went without fences
    int kset_req = DSG_KSET_AUTO;        // dsg_set_kernel_set: the kernel set every step of this handle runs (AUTO: by batch, select_kernels)
    int last_kset = -1;                  // ... and the one the last dsg_forward / dsg_sample ran
    // state / activations
    float *xs32 = nullptr, *partial = nullptr, *X0 = nullptr, *pre1 = nullptr, *pre2 = nullptr, *Xn = nullptr,
          *X1 = nullptr, *fwd_out = nullptr, *io_tmp = nullptr, *io_tmp2 = nullptr, *ext_noise = nullptr;
    size_t ext_noise_cap = 0;
    void *xsA = nullptr, *X0a = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *attn = nullptr, *hidden = nullptr;
    int lanes_now = 1;                               // lanes of the dsg_sample_multi call in progress (1: dsg_sample / dsg_forward)
    float* ffn_part = nullptr; size_t ffn_slab = 0;      // [4][ffn_slab] partial linear2 results of k_ffn_part
    void* X1a = nullptr;                 // LayerNorm1 rows in the GEMM type, fragment-major (k_attn_op -> linear1)
    int* ctr = nullptr;                  // scratch counter for diagnostics
    StepCtl* ctl = nullptr;              // device-resident step control (dsg_kernels.h: StepCtl)
    int* t_arr = nullptr; unsigned* dyn = nullptr;
    int latency_mode = -1;               // dsg_config.latency_mode: -1 auto, 0 never the LATENCY set, 1 always
#ifndef DSG_EMU
    dsg_aql::Ctx aql;                    // hand-written AQL submission of the step loop (dsg_aql.h)
#endif
    // Fence-free step loop (DSG_UC, default 1): every buffer the loop WRITES (state, activations, step control) lives in
    // uncached device memory (MTYPE UC: neither the CUs' L1 nor the XCDs' L2 hold its lines, so a kernel on any XCD reads what
    // the previous kernel wrote without cache maintenance), the step control is read with vector loads (never through the
    // scalar cache), and the AQL packets of the loop carry NO acquire / release fence (first packet acquires, last releases).
    // Weights, tables and conditioning stay in cached memory -- nothing writes them during the loop.  Measured on MI355X
    // (profiles/r02_q_*): the fences are worth ~6 us per batch-1 step, the uncached buffers cost ~3 (117 us with uncached buffers
    // behind the usual fences, DSG_UC=2): 111.0 vs 114.2 us/step, bit-identical samples; 16 clips in 4 lanes: 5628 vs 5120 frames/s.
    // DSG_UC=0: cached buffers, agent-scope fences.  HIP launches / hipGraph keep the runtime's own fences either way.
    // The first fence-free run of a process checks the premise on the device it runs on (uc_selfcheck) and falls back to fenced
    // packets with a warning if a hand-off through uncached memory is ever seen stale.
    int uc_mode = 1;
    bool alloc_uc = false;               // dalloc target: a loop-written buffer
    bool state_fences = false;           // the state lives in cached memory: release on the step's last packet, acquire on its first
    int fence_next = 0;                  // step_launch: the next recorded packet acquires (1) / releases (2) at agent scope
    int aql_mode = 1;                    // DSG_AQL: 1 (default) = AQL packets for the eager step loop, 0 = HIP launches
    bool aql_warned = false;
    bool aql_timing = false;             // the last dsg_sample was timed by the host clock around the AQL run
    double aql_ms = 0.0;
    // A/B switches of the batched sets, read ONCE at dsg_create (round-4 advisor: they used to be getenv calls in select_kernels /
    // run_step, i.e. in the hot path and outside the hipGraph key): -1 = not set
    int env_loc64_from = 2048;           // DSG_LOC64_FROM=<workgroups>: k_loc as two waves per (head, window, clip) from that many of them (test hook / A/B)
    int env_ffn_rt4 = -1;                // DSG_FFN_RT4=<rows>: k_ffn on 64-row blocks from that many token rows at any lane count (0: never)
    int env_ffn_split = -1;              // DSG_FFN_SPLIT=0: linear1 + linear2 + LayerNorm-on-read instead of k_ffn_part + k_ffn_ln (BLOCK)
    int env_clip_attn = -1;              // DSG_CLIP_ATTN=0: QKV GEMM + k_attn_op instead of k_clip_attn + k_ffn_ln (BLOCK; differs in the last bits)
    int* st_tmodel = nullptr; float* st_c[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int st_cap = 0, n_run = 1;
    bool st_valid = false; int st_mode = -1, st_skip = -1; float st_eta = 0.f;      // what the device tables hold
    Sched sched;
    // graphs: key = (B, out_mode, mask batch, const_noise, steps, flags, kernel set) -> exec
    // n_run is part of the key: the captured kernels carry the step-table length as an argument (n_tab); so is the kernel set
    // (a graph captured under one set must never be replayed for a call that asked for another)
    // (round-4 advisor: n_run = the steps of THIS call, n_tab = the length of the whole chain = what the captured kernels clamp their
    // step-table reads with -- a chain run in pieces, first_step / max_steps, has n_run < n_tab, so both are part of the key)
    struct GKey { int B, mode, mb, cn, n_run, n_tab, flags, kset; bool operator<(const GKey& o) const {
        return std::tie(B, mode, mb, cn, n_run, n_tab, flags, kset) < std::tie(o.B, o.mode, o.mb, o.cn, o.n_run, o.n_tab, o.flags, o.kset); } };
    struct GVal { hipGraphExec_t exec; hipGraph_t graph; int steps; };
    std::map<GKey, GVal> graphs;
    float last_ms = -1.f; int last_steps = 0; bool timing_valid = false;
};

// bf16 activations (DSG_PREC_BF16, DSG_PREC_BF16W2): the compute-dtype shadows / fragment layouts of the bf16 kernels
static inline bool is_bf16(const dsg_handle* h) { return h->prec != DSG_PREC_FP32; }

// Uncached device memory is never handed back to the HIP allocator while a handle may still be created (round 4).  Found with
// tools/debug_rowdep*.py: after a handle with uncached loop buffers had been destroyed, a NEW handle whose buffers landed on the
// recycled range computed wrong rows (tiny dims, batch 200 after a batch-170 handle: every frame row >= 4096 -- exactly the part of
// `partial` beyond its first 2 MiB -- off by up to 100 %; DSG_UC=0 clean; same under every kernel set) -- memory that changed its
// caching attribute between two lives is not reliably coherent.
// Round 5 (verdict 8c / advisor): the pool is a SUB-ALLOCATOR over large arenas that stay uncached for their whole life (first fit,
// free neighbours coalesced), so a long-lived process that creates and destroys handles of varying max_batch holds the peak of what
// was alive at once, not the sum of every size class it ever asked for; DSG_UC_POOL_CAP_MB (default 16384) bounds the arenas of a
// device -- past it a handle gets cached loop buffers + fenced packets; dsg_trim() hands arenas without a live block back to HIP.
// Every FRESH arena is filled and read back by two kernels before its first use (uc_arena_ok): that is the observed symptom of a
// recycled range (wrong values between dependent launches), tested where it would show; a failing arena is quarantined (kept, never
// used, never freed) and another one is tried.
struct UcArena {
    char* base = nullptr; size_t size = 0, used = 0;
    std::map<size_t, size_t> free;       // offset -> length of the free runs
    bool quarantined = false;
};
struct UcPool {
    std::mutex mu;
    std::vector<UcArena> arenas[64];     // per device
    std::map<void*, std::pair<char*, size_t>> live;    // block -> (base of its arena, length)
    size_t trimmed = 0;                  // arenas returned by dsg_trim so far (diagnostics)
    // (allocation time only -- handle creation, never the step loop -- so the cap is read where it is applied)
    static size_t cap_bytes() { const char* e = getenv("DSG_UC_POOL_CAP_MB"); return (size_t)(e ? std::max(atoll(e), 0ll) : 16384ll) << 20; }
};
static UcPool& uc_pool() { static UcPool* p = new UcPool(); return *p; }      // (leaked on purpose: arenas outlive every handle)
constexpr size_t UC_ALIGN = 4096, UC_ARENA_MIN = (size_t)32 << 20, UC_ARENA_GRAN = (size_t)2 << 20;
#ifndef DSG_EMU
__global__ void k_uc_fill(unsigned* p, size_t n, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i * 2654435761u + salt;
}
__global__ void k_uc_verify(const unsigned* p, size_t n, unsigned salt, unsigned* bad) {
    unsigned b = 0;
    // (read by OTHER workgroups than the ones that wrote: the grid is walked from the far end)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = n - 1 - i;
        b += p[j] != (unsigned)j * 2654435761u + salt;
    }
    if (b) atomicAdd(bad, b);
}
// fill + read-back of a whole fresh arena, twice with different patterns, through ordinary launches on the null stream
static bool uc_arena_ok(char* base, size_t size) {
    unsigned* bad = nullptr;
    if (hipMalloc((void**)&bad, sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = hipMemset(bad, 0, sizeof(unsigned)) == hipSuccess;
    const size_t n = size / sizeof(unsigned);
    for (unsigned pass = 0; pass < 2 && ok; ++pass) {
        hipLaunchKernelGGL(k_uc_fill, dim3(1024), dim3(256), 0, 0, (unsigned*)base, n, 0x9E3779B9u * (pass + 1));
        hipLaunchKernelGGL(k_uc_verify, dim3(1024), dim3(256), 0, 0, (const unsigned*)base, n, 0x9E3779B9u * (pass + 1), bad);
        ok = hipGetLastError() == hipSuccess;
    }
    unsigned nbad = 1;
    ok = ok && hipMemcpy(&nbad, bad, sizeof nbad, hipMemcpyDeviceToHost) == hipSuccess && nbad == 0;
    (void)hipFree(bad);
    return ok;
}
#endif
// a block of `bytes` of uncached memory on device `dev`, or nullptr (no uncached memory here / pool cap reached)
static void* uc_pool_take(int dev, size_t bytes, bool ignore_cap = false) {
#ifdef DSG_EMU
    (void)dev; (void)bytes; (void)ignore_cap;
    return nullptr;
#else
    UcPool& P = uc_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto& av = P.arenas[dev & 63];
    bytes = (bytes + UC_ALIGN - 1) / UC_ALIGN * UC_ALIGN;
    for (int attempt = 0; attempt < 4; ++attempt) {
        for (size_t ai = 0; ai < av.size(); ++ai) {
            UcArena& A = av[ai];
            if (A.quarantined) continue;
            for (auto it = A.free.begin(); it != A.free.end(); ++it)
                if (it->second >= bytes) {
                    const size_t off = it->first, len = it->second;
                    A.free.erase(it);
                    if (len > bytes) A.free.emplace(off + bytes, len - bytes);
                    A.used += bytes;
                    P.live[A.base + off] = {A.base, bytes};
                    return A.base + off;
                }
        }
        size_t total = 0;
        for (const UcArena& A : av) total += A.size;
        const size_t want = std::max(UC_ARENA_MIN, (bytes + UC_ARENA_GRAN - 1) / UC_ARENA_GRAN * UC_ARENA_GRAN);
        if (!ignore_cap && total + want > UcPool::cap_bytes()) return nullptr;
        void* d = nullptr;
        // (round 6, measured no: FINE-GRAINED device memory instead of uncached is cached non-coherently across the XCDs -- the hand-off self-check
        //  fails with 4 M stale words, the fences come back: 16 clips 204 -> 221 us per step, profiles/r06_e_finegrained_vs_uncached.log)
        if (hipExtMallocWithFlags(&d, want, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        UcArena A;
        A.base = (char*)d; A.size = want;
        A.quarantined = !uc_arena_ok(A.base, want);
        if (A.quarantined)
            fprintf(stderr, "libdsg_hip: WARNING: a fresh %zu MB range of uncached device memory failed its fill / read-back check on device %d; "
                            "quarantined (kept, never used)\n", want >> 20, dev);
        else A.free.emplace(0, want);
        av.push_back(std::move(A));
    }
    return nullptr;
#endif
}
static void uc_pool_give(int dev, void* d) {
    UcPool& P = uc_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    auto lv = P.live.find(d);
    if (lv == P.live.end()) return;
    UcArena* Ap = nullptr;
    for (UcArena& X : P.arenas[dev & 63]) if (X.base == lv->second.first) Ap = &X;
    if (!Ap) return;
    UcArena& A = *Ap;
    size_t off = (size_t)((char*)d - A.base), len = lv->second.second;
    P.live.erase(lv);
    A.used -= len;
    auto nx = A.free.lower_bound(off);
    if (nx != A.free.end() && off + len == nx->first) { len += nx->second; nx = A.free.erase(nx); }
    if (nx != A.free.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second == off) { off = pv->first; len += pv->second; A.free.erase(pv); }
    }
    A.free.emplace(off, len);
}
// Hands every arena of `device` that holds no live block back to the HIP allocator (device < 0: all devices) and reports the bytes
// still held.  Meant for a long-lived service between bursts of work; arenas allocated afterwards are checked like any fresh one.
extern "C" int dsg_trim(int device, long long* bytes_released, long long* bytes_held) {
    UcPool& P = uc_pool();
    int dev_before = -1;
    if (hipGetDevice(&dev_before) != hipSuccess) { (void)hipGetLastError(); dev_before = -1; }
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{dev_before};      // the caller's current device is left as it was
    std::lock_guard<std::mutex> lock(P.mu);
    long long rel = 0, held = 0;
    for (int dev = 0; dev < 64; ++dev) {
        if (device >= 0 && dev != (device & 63)) { for (const UcArena& A : P.arenas[dev]) held += (long long)A.size; continue; }
        auto& av = P.arenas[dev];
        bool any = false;
        for (const UcArena& A : av) any = any || (A.used == 0 && !A.quarantined);
        if (!any) { for (const UcArena& A : av) held += (long long)A.size; continue; }
        HIPCHK(hipSetDevice(dev));
        HIPCHK(hipDeviceSynchronize());
        std::vector<UcArena> keep;
        for (size_t ai = 0; ai < av.size(); ++ai) {
            if (av[ai].used == 0 && !av[ai].quarantined) { (void)hipFree(av[ai].base); rel += (long long)av[ai].size; ++P.trimmed; }
            else { held += (long long)av[ai].size; keep.push_back(std::move(av[ai])); }
        }
        av = std::move(keep);
    }
    if (bytes_released) *bytes_released = rel;
    if (bytes_held) *bytes_held = held;
    return 0;
}
template <class T>
static int dalloc(dsg_handle* h, T** p, size_t n_elems, bool zero = true) {
    void* d = nullptr;
    size_t bytes = n_elems * sizeof(T);
    if (bytes == 0) bytes = 16;
    bool is_uc = false;
#ifndef DSG_EMU
    if (h->uc_mode && h->alloc_uc && !h->alloc_shared) {
        d = uc_pool_take(h->cfg.device, bytes);
        if (!d) h->uc_mode = 0;          // no uncached memory here (or the pool's cap is reached): cached buffers, fenced packets
        is_uc = d != nullptr;
    }
#endif
    if (!d) HIPCHK(hipMalloc(&d, bytes));
    // Zero-fill and WAIT for it.  hipMemset on device memory may return before the fill has run; the handle's stream is
    // non-blocking (no implicit ordering with the null stream), so a late fill could wipe what the first kernels on the
    // handle's stream (weight packing, conditioning) or a following synchronous copy have already written.  Seen once
    // as a wrong TWH forward in a full `pytest -m gpu` run.  Allocation happens at create / load time only.
    if (zero) {
        HIPCHK(hipMemsetAsync(d, 0, bytes, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (h->alloc_shared) h->shared->allocs.push_back(d);
    else if (is_uc) h->allocs_uc.push_back(d);
    else h->allocs.push_back(d);
    *p = (T*)d;
    return 0;
}
static int dalloc_bytes(dsg_handle* h, void** p, size_t bytes) {
    unsigned char* d = nullptr;
    CHK(dalloc(h, &d, bytes));
    *p = d;
    return 0;
}

// expected checkpoint tensors (name -> shape); the reference's state_dict contract
static std::map<std::string, std::vector<int64_t>> expected_tensors(const dsg_handle* h) {
    std::map<std::string, std::vector<int64_t>> m;
    const int64_t D = h->D, J = h->J, S = h->S, A = h->A, As = h->As, ff = h->ff;
    m["WavEncoder.audio_feature_map.weight"] = {A, As};
    m["WavEncoder.audio_feature_map.bias"] = {A};
    m["sequence_pos_encoder.pe"] = {h->cfg.pe_max_len, 1, D};
    m["input_process.poseEmbedding.weight"] = {D, J};
    m["input_process.poseEmbedding.bias"] = {D};
    for (int i = 0; i < h->L; ++i) {
        const std::string p = "seqTransEncoder.layers." + std::to_string(i) + ".";
        m[p + "self_attn.in_proj_weight"] = {3 * D, D};
        m[p + "self_attn.in_proj_bias"] = {3 * D};
        m[p + "self_attn.out_proj.weight"] = {D, D};
        m[p + "self_attn.out_proj.bias"] = {D};
        m[p + "linear1.weight"] = {ff, D};
        m[p + "linear1.bias"] = {ff};
        m[p + "linear2.weight"] = {D, ff};
        m[p + "linear2.bias"] = {D};
        m[p + "norm1.weight"] = {D}; m[p + "norm1.bias"] = {D};
        m[p + "norm2.weight"] = {D}; m[p + "norm2.bias"] = {D};
    }
    m["embed_timestep.time_embed.0.weight"] = {D, D};
    m["embed_timestep.time_embed.0.bias"] = {D};
    m["embed_timestep.time_embed.2.weight"] = {D, D};
    m["embed_timestep.time_embed.2.bias"] = {D};
    if (h->cfg.variant == 3) {
        m["embed_style.weight"] = {64, h->cfg.style_dim_in};
        m["embed_style.bias"] = {64};
        m["embed_text.weight"] = {D - 64, J * S};
        m["embed_text.bias"] = {D - 64};
    } else {
        m["embed_style.weight"] = {D, h->cfg.style_dim_in};
        m["embed_style.bias"] = {D};
        m["embed_text.weight"] = {A, J};
        m["embed_text.bias"] = {A};
        if (h->cfg.variant == 5) {          // DiffuseStyleGesture++ (BEAT-TWH mdm.py:85-89)
            m["embed_text_last.weight"] = {A, J};
            m["embed_text_last.bias"] = {A};
        }
    }
    m["output_process.poseFinal.weight"] = {J, D};
    m["output_process.poseFinal.bias"] = {J};
    m["rel_pos.inv_freq"] = {h->hdl / 2};
    m["input_process2.weight"] = {D, 2 * D + A};
    m["input_process2.bias"] = {D};
    return m;
}

static bool stream_set_ok(const dsg_handle* h);
static bool rows_w2_ok(const dsg_handle* h);
extern "C" int dsg_create(const dsg_config* c, dsg_handle** out) {
    if (!c || !out) return fail(DSG_E_INVALID, "dsg_create: null argument");
    if (c->variant < 3 || c->variant > 5) return fail(DSG_E_NOT_IMPLEMENTED, "variant must be 3, 4 or 5");
    if (c->variant == 5 && c->n_poses <= 2 * c->n_seed) return fail(DSG_E_INVALID, "variant 5 needs n_poses > 2 * n_seed");
    if (c->latent_dim % 64 || c->latent_dim > 512 || c->latent_dim <= 0)
        return fail(DSG_E_INVALID, "latent_dim must be a multiple of 64, <= 512");
    if (c->variant == 3 && c->latent_dim <= 64) return fail(DSG_E_INVALID, "variant 3 needs latent_dim > 64");
    if (c->window <= 0 || c->window > 16 || c->n_poses % c->window)
        return fail(DSG_E_INVALID, "n_poses must be a multiple of window (<= 16)");
    if (c->num_heads <= 0 || c->latent_dim % c->num_heads) return fail(DSG_E_INVALID, "num_heads");
    const int hd = c->latent_dim / c->num_heads;
    if (!(hd == 32 || hd == 64 || hd == 96 || hd == 128)) return fail(DSG_E_INVALID, "self-attention head dim must be 32/64/96/128");
    if (c->local_heads <= 0 || c->latent_dim % c->local_heads) return fail(DSG_E_INVALID, "local_heads");
    const int hdl = c->latent_dim / c->local_heads;
    if (hdl > 64 || (hdl & 1)) return fail(DSG_E_INVALID, "local head dim must be even and <= 64");
    if (c->ff_size % 128 || c->ff_size <= 0) return fail(DSG_E_INVALID, "ff_size must be a multiple of 128");
    if (c->audio_dim % 4) return fail(DSG_E_INVALID, "audio_dim must be a multiple of 4");
    if (c->max_batch <= 0 || c->njoints <= 0 || c->n_seed < 0 || c->n_seed >= c->n_poses)
        return fail(DSG_E_INVALID, "bad dims");
    if (c->precision != DSG_PREC_FP32 && c->precision != DSG_PREC_BF16 && c->precision != DSG_PREC_BF16W2) return fail(DSG_E_INVALID, "precision");
#ifdef DSG_DEV_BF16_ONLY
    if (c->precision != DSG_PREC_BF16) return fail(DSG_E_NOT_IMPLEMENTED, "development build (make dev): bf16 only");
#endif
    const int ntok = c->n_poses + 1, Tp = rup(ntok, 32);
    if (!(Tp == 32 || Tp == 96 || Tp == 160))
        return fail(DSG_E_NOT_IMPLEMENTED, "attention kernel is instantiated for n_poses+1 padded to 32, 96 or 160 tokens");
    // (the kernels index rows and features in 32 bits on the 24-bit multiplier -- dsg_kernels.h: imul24)
    {
        const size_t widest = (size_t)std::max(std::max(c->ff_size, c->latent_dim), (c->njoints + 127) / 128 * 128);
        if ((size_t)c->max_batch * (size_t)ntok >= ((size_t)1 << 23) || (size_t)c->max_batch * (size_t)ntok * widest >= ((size_t)1 << 30))
            return fail(DSG_E_INVALID, "max_batch too large: token rows x widest row must stay below 2^30 elements (use several lanes / handles)");
    }
    HIPCHK(hipSetDevice(c->device));

    dsg_handle* h = new dsg_handle();
    h->shared = std::make_shared<SharedWeights>();
    h->cfg = *c;
    h->prec = c->precision;
    h->es = is_bf16(h) ? 2 : 4;
    h->kbk = is_bf16(h) ? 32 : 16;
    h->J = c->njoints; h->T = c->n_poses; h->S = c->n_seed; h->D = c->latent_dim; h->As = c->audio_src_dim;
    h->A = c->audio_dim; h->W = c->window; h->L = c->num_layers; h->H = c->num_heads; h->hd = hd; h->ff = c->ff_size;
    h->Hl = c->local_heads; h->hdl = hdl; h->ntok = ntok; h->Tp = Tp;
    h->Jp = rup(h->J, 128); h->Jq = rup(h->J, 4); h->Bmax = c->max_batch;
    h->KSin = cdiv(h->Jp, 256);
    h->Ta = c->variant == 3 ? h->T : h->T - h->S * (c->variant == 5 ? 2 : 1);
    h->n_te = c->train_steps > 0 ? c->train_steps : 1000;
    if (h->n_te > c->pe_max_len) { delete h; return fail(DSG_E_INVALID, "train_steps > pe_max_len"); }
    h->layers.resize(h->L);
    h->latency_mode = c->latency_mode == 1 ? 0 : (c->latency_mode == 2 ? 1 : -1);
    // Environment switches of the release library (everything else is an API / config field):
    //   DSG_KSET  kernel set for every handle (1 latency, 2 tile, 3 block, 4 stream; overrides dsg_set_kernel_set)   [A/B runs]
    //   DSG_UC    0: cached loop buffers + fenced packets, 1: uncached + fence-free, 2: uncached + fenced
    //   DSG_AQL   0: HIP launches instead of hand-written AQL packets
    if (const char* e = getenv("DSG_KSET")) {
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end == e || *end != 0 || v < DSG_KSET_AUTO || v > DSG_KSET_ROWS) {
            delete h;
            return fail(DSG_E_INVALID, std::string("DSG_KSET must be 0 (auto) .. 5 (rows), got '") + e + "'");
        }
        h->kset_req = (int)v;
    }
    // Large batches on the BLOCK set re-read their activations from the L2 often enough that the uncached buffers cost what the
    // fences save (4 x 16: 9667 vs 9680 frames/s, 4 x 32: 10 430 vs 10 695): cached + fenced from batch 17 -- unless the handle can run
    // the STREAM set, which large batches select and which reads an activation block once per 128-column panel: fence-free wins there
    // at every size (1 x 64: 505 -> 479 us, 1 x 32: 356 -> 338, 4 x 32: 13.3k -> 13.8k frames/s; profiles/r03_q_uc_stream.log)
    if (c->max_batch > 16 && !stream_set_ok(h) && !rows_w2_ok(h)) h->uc_mode = 0;      // (bf16w2: ROWS at these sizes, every activation read once per workgroup)
    if (const char* e = getenv("DSG_UC")) h->uc_mode = atoi(e);
    if (const char* e = getenv("DSG_AQL")) h->aql_mode = atoi(e);
    else {
        // under a profiler the HSA queues are intercepted and rewritten (rocprofv3 crashed on hand-written packets), so
        // profiled runs use the HIP launches -- the same kernels, visible to the tool
        for (const char* v : {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_COUNTER_COLLECTION", "ROCPROF_KERNEL_TRACE", "LD_PRELOAD"}) {
            const char* val = getenv(v);
            if (val && (strstr(val, "rocprof") || strstr(v, "ROCPROF"))) h->aql_mode = 0;
        }
    }
    if (const char* e = getenv("DSG_FFN_RT4")) h->env_ffn_rt4 = std::max(atoi(e), 0);
    if (const char* e = getenv("DSG_LOC64_FROM")) h->env_loc64_from = std::max(atoi(e), 1);
    if (const char* e = getenv("DSG_FFN_SPLIT")) h->env_ffn_split = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("DSG_CLIP_ATTN")) h->env_clip_attn = atoi(e) != 0 ? 1 : 0;
    *out = h;

    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&h->ev_t0));
    HIPCHK(hipEventCreate(&h->ev_t1));

    const int B = h->Bmax, D = h->D;
    // row buffers carry one extra padded token block: the fused attention kernel reads Tp rows per batch element
    // (+64: the block GEMMs of dsg_batched.h read and LayerNorm whole 64-row blocks)
    const size_t Min_pad = rup(B * h->T, 16) + 16 + 128, M_pad = rup(B * ntok, 16) + Tp + 128;
    // The sampler STATE (x_t fp32 + its bf16 shadow) is the one loop-written buffer that crosses a STEP boundary only: written by
    // the last kernel of a step, read by the first (and last) kernel of the next.  It stays in cached memory and exactly those two
    // packets carry an agent-scope release / acquire (state_fences): the timeline of the fence-free path
    // (profiles/r03_aql_step_timeline_b16.json) showed 11.6 us between the pose head's last wave and the next step's first wave at
    // batch 16 -- 9.8 MB of state written as uncached stores have to be acknowledged one by one -- against ~1 us for a plain
    // boundary; a write-back of the same bytes from the L2 costs 1-2 us.  DSG_STATE_UC=1: the round-2 behaviour (state uncached).
    h->state_fences = h->uc_mode == 1 && !(getenv("DSG_STATE_UC") && atoi(getenv("DSG_STATE_UC")));
    h->alloc_uc = !h->state_fences;
    CHK(dalloc(h, &h->xs32, (size_t)B * h->T * h->Jp + 16 * h->Jp));
    if (is_bf16(h)) CHK(dalloc_bytes(h, &h->xsA, ((size_t)Min_pad * h->Jp) * h->es));
    h->alloc_uc = true;                  // ---- written AND read inside one step by the kernels of the loop
    CHK(dalloc(h, &h->partial, (size_t)h->KSin * Min_pad * D));
    CHK(dalloc(h, &h->X0, M_pad * D));
    CHK(dalloc_bytes(h, &h->X0a, M_pad * D * h->es * (h->prec == DSG_PREC_BF16W2 ? 2 : 1)));      // (bf16w2, ROWS: hi + lo images)
    CHK(dalloc(h, &h->pre1, M_pad * D));
    CHK(dalloc(h, &h->pre2, M_pad * D));
    CHK(dalloc(h, &h->Xn, M_pad * D));
    CHK(dalloc(h, &h->X1, M_pad * D));
    CHK(dalloc_bytes(h, &h->attn, M_pad * D * h->es * (h->prec == DSG_PREC_BF16W2 ? 2 : 1)));      // (bf16w2: hi + lo images)
    CHK(dalloc_bytes(h, &h->X1a, M_pad * D * h->es));
    CHK(dalloc_bytes(h, &h->hidden, M_pad * (size_t)h->ff * h->es * (h->prec == DSG_PREC_BF16W2 ? 2 : 1)));      // (bf16w2: hi + lo images)
    h->ffn_slab = M_pad * D;             // (the partial linear2 slabs of k_ffn_part are allocated on the first BLOCK call: ensure_set_buffers)
    const size_t qkv_elems = (size_t)B * h->H * Tp * hd;
    CHK(dalloc_bytes(h, &h->q, qkv_elems * h->es));
    CHK(dalloc_bytes(h, &h->k, qkv_elems * h->es));
    CHK(dalloc_bytes(h, &h->vt, qkv_elems * h->es));
    CHK(dalloc(h, &h->ctl, 1));
    h->alloc_uc = false;                 // ---- inputs / outputs / conditioning: constant while the loop runs
    CHK(dalloc(h, &h->fwd_out, (size_t)B * h->J * h->T));
    CHK(dalloc(h, &h->io_tmp, (size_t)B * h->J * h->T));
    CHK(dalloc(h, &h->io_tmp2, (size_t)B * h->J * h->T));
    CHK(dalloc(h, &h->emb1, (size_t)B * D));
    CHK(dalloc(h, &h->cvec, (size_t)B * D));
    CHK(dalloc(h, &h->Cf, (size_t)B * h->T * D));
    CHK(dalloc(h, &h->enc, (size_t)B * h->T * h->A));
    CHK(dalloc(h, &h->c_style, (size_t)B * c->style_dim_in));
    CHK(dalloc(h, &h->c_seed, (size_t)B * h->J * (h->S > 0 ? h->S : 1)));
    if (h->cfg.variant == 5) CHK(dalloc(h, &h->c_seed_last, (size_t)B * h->J * h->S));      // (guidance: rows [B/2, B) = the twins' copy)
    CHK(dalloc(h, &h->c_audio, (size_t)B * h->Ta * h->As));
    CHK(dalloc(h, &h->mask, (size_t)B * h->T));
    CHK(dalloc(h, &h->ctr, 8));
    CHK(dalloc(h, &h->cfg_scale, (size_t)B));
    CHK(dalloc(h, &h->dyn, 8));
    CHK(dalloc(h, &h->t_arr, (size_t)B));
    // the xs32 master is read as a GEMM operand in fp32 mode: rows padded to a 16-row tile exist (allocated above)
    return 0;
}

extern "C" int dsg_destroy(dsg_handle* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->graphs) { (void)hipGraphExecDestroy(kv.second.exec); (void)hipGraphDestroy(kv.second.graph); }
#ifndef DSG_EMU
    dsg_aql::destroy(h->aql);
#endif
    for (void* p : h->allocs) (void)hipFree(p);
    for (void* p : h->allocs_uc) uc_pool_give(h->cfg.device, p);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    if (h->ev_t0) (void)hipEventDestroy(h->ev_t0);
    if (h->ev_t1) (void)hipEventDestroy(h->ev_t1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

// A second sampling lane over the same weights: own stream / HSA queue, state, activations, conditioning and schedule, but
// the checkpoint-derived buffers (packed weights, time-embedding and rotary tables) are the source's, reference counted.
// `max_batch` <= 0 keeps the source's.
extern "C" int dsg_clone(dsg_handle* src, int max_batch, dsg_handle** out) {
    if (!src || !out) return fail(DSG_E_INVALID, "dsg_clone: null argument");
    if (!src->finalized) return fail(DSG_E_STATE, "dsg_clone before dsg_finalize_weights");
    dsg_config c = src->cfg;
    if (max_batch > 0) c.max_batch = max_batch;
    dsg_handle* h = nullptr;
    CHK(dsg_create(&c, &h));
    h->shared = src->shared;
    h->is_clone = true;
    h->raw = src->raw;
    h->Wp_in = src->Wp_in; h->Wp_out = src->Wp_out; h->b_out = src->b_out; h->layers = src->layers;
    h->TE = src->TE; h->TE2 = src->TE2; h->rcos = src->rcos; h->rsin = src->rsin; h->cbase = src->cbase;
    h->zero_bias = src->zero_bias;
    h->kset_req = src->kset_req;
    h->finalized = true;
    *out = h;
    return 0;
}

extern "C" int dsg_load_tensor(dsg_handle* h, const char* name, const void* data, const int64_t* shape, int ndim,
                               int dtype) {
    if (!h || !name || !data || !shape) return fail(DSG_E_INVALID, "dsg_load_tensor: null argument");
    if (h->is_clone) return fail(DSG_E_STATE, "dsg_load_tensor on a clone: load the weights into the source handle");
    if (dtype != 0) return fail(DSG_E_NOT_IMPLEMENTED, "dsg_load_tensor: only float32 (dtype 0)");
    HIPCHK(hipSetDevice(h->cfg.device));
    std::string nm(name);
    if (nm == "embed_timestep.sequence_pos_encoder.pe") nm = "sequence_pos_encoder.pe";   // aliased buffer
    auto exp = expected_tensors(h);
    auto it = exp.find(nm);
    if (it == exp.end()) return fail(DSG_E_UNEXPECTED_KEY, "unexpected key in state_dict: " + nm);
    size_t n = 1, ne = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    for (auto v : it->second) ne *= (size_t)v;
    bool same = (int)it->second.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = it->second[i] == shape[i];
    if (!same || n != ne) return fail(DSG_E_INVALID, "size mismatch for " + nm);
    RawT& r = h->raw[nm];
    if (!r.d) {
        h->alloc_shared = true;
        const int rc = dalloc(h, &r.d, n, false);
        h->alloc_shared = false;
        CHK(rc);
    }
    r.n = n; r.shape.assign(shape, shape + ndim);
    const bool dev_src = is_device_ptr(data);
    HIPCHK(hipMemcpy(r.d, data, n * sizeof(float), dev_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    // a device-to-device hipMemcpy may return before the copy has run (null stream); the packing kernels that read r.d
    // run on the handle's non-blocking stream, which is not ordered against it
    if (dev_src) HIPCHK(hipDeviceSynchronize());
    h->finalized = false;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------
static int launch_mm(dsg_handle* h, float* C, int ldc, const float* A, long long sam, long long sak, const float* Bm,
                     long long sbn, long long sbk, const float* bias, const float* add, long long sadd, int add_div,
                     int M, int N, int K, int act = 0) {
    MMArgs a;
    a.C = C; a.ldc = ldc; a.A = A; a.sam = sam; a.sak = sak; a.Bm = Bm; a.sbn = sbn; a.sbk = sbk; a.bias = bias;
    a.add = add; a.sadd = sadd; a.add_div = add_div < 1 ? 1 : add_div; a.M = M; a.N = N; a.K = K; a.act = act;
    const size_t n = (size_t)M * N;
    if (n == 0) return 0;
    if (K >= 512 && n <= (1u << 20)) {  // long reductions: a wave per (row, 4 columns) (k_mm_wave); the big set-up tables stay on k_mm_naive
        const size_t nw = (size_t)M * ((N + 3) / 4);
        hipLaunchKernelGGL(k_mm_wave, dim3((int)std::min<size_t>((nw + 3) / 4, 16384)), dim3(256), 0, h->stream, a);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int grid = (int)std::min<size_t>((n + 127) / 128, 4096);
    hipLaunchKernelGGL(k_mm_naive, dim3(grid), dim3(128), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
template <class P>
static int launch_pack(dsg_handle* h, void** dst, const float* W, long long ldw, int N, int K, int NT, int KBtot) {
    const size_t n = (size_t)NT * KBtot * 64 * P::E;
    CHK(dalloc_bytes(h, dst, n * sizeof(typename P::elem) * P::WF));
    const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL((k_pack_w<P>), dim3(grid), dim3(256), 0, h->stream, *dst, W, ldw, N, K, NT, KBtot);
    HIPCHK(hipGetLastError());
    return 0;
}
static int pack(dsg_handle* h, void** dst, const float* W, long long ldw, int N, int K, int Npad, int Kpad) {
    if (h->prec == DSG_PREC_BF16) return launch_pack<PBF16>(h, dst, W, ldw, N, K, Npad / 16, Kpad / 32);
    if (h->prec == DSG_PREC_BF16W2) return launch_pack<PBF16W2>(h, dst, W, ldw, N, K, Npad / 16, Kpad / 32);
    return launch_pack<PF32>(h, dst, W, ldw, N, K, Npad / 16, Kpad / 16);
}
static int padded_vec(dsg_handle* h, float** dst, const float* src, int n, int npad) {
    CHK(dalloc(h, dst, (size_t)npad));
    hipLaunchKernelGGL(k_copy_pad, dim3(cdiv(npad, 256)), dim3(256), 0, h->stream, *dst, src, n, npad);
    HIPCHK(hipGetLastError());
    return 0;
}

static int finalize_weights(dsg_handle* h);
extern "C" int dsg_finalize_weights(dsg_handle* h) {
    if (!h) return fail(DSG_E_INVALID, "null handle");
    if (h->is_clone) return fail(DSG_E_STATE, "dsg_finalize_weights on a clone");
    h->alloc_shared = true;
    const int rc = finalize_weights(h);
    h->alloc_shared = false;
    return rc;
}
static int finalize_weights(dsg_handle* h) {
    HIPCHK(hipSetDevice(h->cfg.device));
    auto exp = expected_tensors(h);
    for (auto& kv : exp)
        if (!h->raw.count(kv.first)) return fail(DSG_E_MISSING_KEY, "missing key in state_dict: " + kv.first);
    auto R = [&](const std::string& n) { return h->raw[n].d; };
    const int D = h->D, J = h->J, A = h->A, ff = h->ff, L = h->L;
    const int W2ld = 2 * D + A;
    const float* W2 = R("input_process2.weight");

    // time-embedding tables  (TimestepEmbedder, main/model/mdm.py:434-448)
    float* H1 = nullptr;
    CHK(dalloc(h, &H1, (size_t)h->n_te * D));
    CHK(dalloc(h, &h->TE, (size_t)h->n_te * D));
    CHK(dalloc(h, &h->TE2, (size_t)h->n_te * D));
    CHK(launch_mm(h, H1, D, R("sequence_pos_encoder.pe"), D, 1, R("embed_timestep.time_embed.0.weight"), D, 1,
                  R("embed_timestep.time_embed.0.bias"), nullptr, 0, 1, h->n_te, D, D, 1));
    CHK(launch_mm(h, h->TE, D, H1, D, 1, R("embed_timestep.time_embed.2.weight"), D, 1,
                  R("embed_timestep.time_embed.2.bias"), nullptr, 0, 1, h->n_te, D, D, 0));
    CHK(launch_mm(h, h->TE2, D, h->TE, D, 1, W2, W2ld, 1, nullptr, nullptr, 0, 1, h->n_te, D, D, 0));
    // fold input_process2[:, D:2D] . poseEmbedding   (mdm.py:198, :202-206) -> [D][J]
    float* Wfold = nullptr;
    CHK(dalloc(h, &Wfold, (size_t)D * J));
    CHK(launch_mm(h, Wfold, J, W2 + D, W2ld, 1, R("input_process.poseEmbedding.weight"), 1, J, nullptr, nullptr, 0, 1,
                  D, J, D, 0));
    CHK(dalloc(h, &h->cbase, (size_t)D));
    CHK(launch_mm(h, h->cbase, D, R("input_process.poseEmbedding.bias"), 0, 1, W2 + D, W2ld, 1,
                  R("input_process2.bias"), nullptr, 0, 1, 1, D, D, 0));
    CHK(pack(h, &h->Wp_in, Wfold, J, D, J, D, h->Jp));
    CHK(dalloc(h, &h->zero_bias, (size_t)std::max(D, h->Jp)));
    for (int i = 0; i < L; ++i) {
        const std::string p = "seqTransEncoder.layers." + std::to_string(i) + ".";
        Layer& l = h->layers[i];
        CHK(pack(h, &l.Wqkv, R(p + "self_attn.in_proj_weight"), D, 3 * D, D, 3 * D, D));
        CHK(pack(h, &l.Wo, R(p + "self_attn.out_proj.weight"), D, D, D, D, D));
        CHK(pack(h, &l.W1, R(p + "linear1.weight"), D, ff, D, ff, D));
        CHK(pack(h, &l.W2, R(p + "linear2.weight"), ff, D, ff, D, ff));
        l.bqkv = R(p + "self_attn.in_proj_bias"); l.bo = R(p + "self_attn.out_proj.bias");
        l.b1 = R(p + "linear1.bias"); l.b2 = R(p + "linear2.bias");
        l.g1 = R(p + "norm1.weight"); l.be1 = R(p + "norm1.bias");
        l.g2 = R(p + "norm2.weight"); l.be2 = R(p + "norm2.bias");
    }
    CHK(pack(h, &h->Wp_out, R("output_process.poseFinal.weight"), D, J, D, h->Jp, D));
    CHK(padded_vec(h, &h->b_out, R("output_process.poseFinal.bias"), J, h->Jp));

    // rotary tables (rotary.py:12-16): freqs = pos * inv_freq in fp32, cos/sin of that fp32 value
    {
        const int half = h->hdl / 2, npos = h->T + 1;
        std::vector<float> inv(half), c((size_t)npos * half), s((size_t)npos * half);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(inv.data(), R("rel_pos.inv_freq"), half * sizeof(float), hipMemcpyDeviceToHost));
        for (int p = 0; p < npos; ++p)
            for (int k = 0; k < half; ++k) {
                const float fr = (float)p * inv[k];
                c[(size_t)p * half + k] = (float)std::cos((double)fr);
                s[(size_t)p * half + k] = (float)std::sin((double)fr);
            }
        CHK(dalloc(h, &h->rcos, c.size()));
        CHK(dalloc(h, &h->rsin, s.size()));
        HIPCHK(hipMemcpy(h->rcos, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->rsin, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->finalized = true;
    return 0;
}

extern "C" int dsg_set_schedule(dsg_handle* h, const double* betas, const int64_t* tmap, int n) {
    if (!h || !betas || !tmap) return fail(DSG_E_INVALID, "dsg_set_schedule: null argument");
    Sched s;
    CHK(build_sched(betas, n, s));
    s.tmap.resize(n);
    for (int i = 0; i < n; ++i) {
        if (tmap[i] < 0 || tmap[i] >= h->n_te) return fail(DSG_E_INVALID, "timestep_map entry out of range");
        s.tmap[i] = (int)tmap[i];
    }
    h->sched = s;
    h->st_valid = false;
    // captured graphs carry the table pointers and the table length of the schedule they were captured under
    for (auto& kv : h->graphs) { (void)hipGraphExecDestroy(kv.second.exec); (void)hipGraphDestroy(kv.second.graph); }
    h->graphs.clear();
    if (n > h->st_cap) {
        HIPCHK(hipSetDevice(h->cfg.device));
        CHK(dalloc(h, &h->st_tmodel, (size_t)n));
        for (int k = 0; k < 5; ++k) CHK(dalloc(h, &h->st_c[k], (size_t)n));
        h->st_cap = n;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// per-window conditioning (mdm.py:180-190; BEAT-TWH mdm.py:145, :188-190): everything that does not depend on x_t or t
// ---------------------------------------------------------------------------------------------------------
static int upload(dsg_handle* h, void* dst, const void* src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    return 0;
}

static int order_after(dsg_handle* h, void* user_stream);
static int order_before(dsg_handle* h, void* user_stream);

// y['seed_last'] of DiffuseStyleGesture++ (cross_local_attention5, BEAT-TWH-main/model/mdm.py:229): [B, J, 1, S].  It is the same
// snippet for every window of a clip (BEAT-TWH sample.py:85-93), so it is handed over once and kept.
extern "C" int dsg_set_seed_last(dsg_handle* h, const float* seed_last, int B, void* stream) {
    if (!h || !seed_last) return fail(DSG_E_INVALID, "null argument");
    if (h->cfg.variant != 5) return fail(DSG_E_INVALID, "dsg_set_seed_last is for variant 5 (cross_local_attention5) only");
    if (B <= 0 || B > h->Bmax) return fail(DSG_E_INVALID, "batch exceeds max_batch");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(order_after(h, stream));
    CHK(upload(h, h->c_seed_last, seed_last, (size_t)B * h->J * h->S * sizeof(float)));
    h->seed_last_B = B;
    CHK(order_before(h, stream));     // the caller may release / overwrite its tensor once its stream has passed this point
    return 0;
}

// conditioning of batch rows [row0, row0 + B): inputs are the caller's B elements, `uncond` selects the masked variant
static int cond_rows(dsg_handle* h, const float* style, const float* seed, const float* audio, int row0, int B, int uncond) {
    auto R = [&](const std::string& n) { return h->raw[n].d; };
    const int D = h->D, J = h->J, S = h->S, A = h->A, As = h->As, T = h->T, sdi = h->cfg.style_dim_in;
    const int W2ld = 2 * D + A;
    const float* W2 = R("input_process2.weight");
    float* c_style = h->c_style + (size_t)row0 * sdi;
    float* c_seed = h->c_seed + (size_t)row0 * J * (S > 0 ? S : 1);
    float* c_audio = h->c_audio + (size_t)row0 * h->Ta * As;
    float* emb1 = h->emb1 + (size_t)row0 * D;
    float* enc = h->enc + (size_t)row0 * T * A;
    float* cvec = h->cvec + (size_t)row0 * D;
    float* Cf = h->Cf + (size_t)row0 * T * D;
    CHK(upload(h, c_style, style, (size_t)B * sdi * sizeof(float)));
    if (S > 0) CHK(upload(h, c_seed, seed, (size_t)B * J * S * sizeof(float)));
    CHK(upload(h, c_audio, audio, (size_t)B * h->Ta * As * sizeof(float)));
    const int sdo = h->cfg.variant == 3 ? 64 : D;
    if (uncond) {       // mask_cond(force_mask=True): zeros AFTER the style linear (mdm.py:156-159, :180)
        hipLaunchKernelGGL(k_fill_f32, dim3(cdiv(B * D, 256)), dim3(256), 0, h->stream, emb1, 0.f, (size_t)B * D);
        HIPCHK(hipGetLastError());
    } else {
        CHK(launch_mm(h, emb1, D, c_style, sdi, 1, R("embed_style.weight"), sdi, 1, R("embed_style.bias"),
                      nullptr, 0, 1, B, sdo, sdi));
    }
    if (h->cfg.variant == 3) {
        // embed_text(flattened seed) -> emb1[:, 64:]; with force_mask the seed is zeroed BEFORE the linear (bias only)
        CHK(launch_mm(h, emb1 + 64, D, c_seed, (long long)J * S, 1, R("embed_text.weight"), (long long)J * S, 1,
                      R("embed_text.bias"), nullptr, 0, 1, B, D - 64, uncond ? 0 : J * S));
        CHK(launch_mm(h, enc, A, c_audio, As, 1, R("WavEncoder.audio_feature_map.weight"), As, 1,
                      R("WavEncoder.audio_feature_map.bias"), nullptr, 0, 1, B * T, A, As));
    } else {
        for (int b = 0; b < B; ++b) {
            // per-frame seed embedding: rows 0..S-1 of the "audio" block  (BEAT-TWH mdm.py:188)
            CHK(launch_mm(h, enc + (size_t)b * T * A, A, c_seed + (size_t)b * J * S, 1, S, R("embed_text.weight"),
                          J, 1, R("embed_text.bias"), nullptr, 0, 1, S, A, J));
            CHK(launch_mm(h, enc + ((size_t)b * T + S) * A, A, c_audio + (size_t)b * h->Ta * As, As, 1,
                          R("WavEncoder.audio_feature_map.weight"), As, 1, R("WavEncoder.audio_feature_map.bias"),
                          nullptr, 0, 1, h->Ta, A, As));
            if (h->cfg.variant == 5)    // rows S+Ta .. T-1: embed_text_last(y['seed_last'])  (BEAT-TWH mdm.py:229-230)
                CHK(launch_mm(h, enc + ((size_t)b * T + S + h->Ta) * A, A, h->c_seed_last + (size_t)(row0 + b) * J * S, 1, S,
                              R("embed_text_last.weight"), J, 1, R("embed_text_last.bias"), nullptr, 0, 1, S, A, J));
        }
    }
    // cvec[b] = b2 + W2b.bp + W2a.emb1[b];  Cf[b,f] = W2c.enc[b,f] + cvec[b]
    CHK(launch_mm(h, cvec, D, emb1, D, 1, W2, W2ld, 1, nullptr, h->cbase, 0, 1, B, D, D));
    CHK(launch_mm(h, Cf, D, enc, A, 1, W2 + 2 * D, W2ld, 1, nullptr, cvec, D, T, B * T, D, A));
    return 0;
}

// cfg_scale == nullptr: plain conditioning (`uncond` as given).  Else classifier-free guidance: rows [0, B) conditional,
// rows [B, 2B) their unconditional twins, guidance scale per element (y['scale'], cfg_sampler.py:29-31).
static int set_window_cond(dsg_handle* h, const float* style, const float* seed, const float* audio,
                           const uint8_t* mask_local, int mask_batch, int B, int uncond, const float* cfg_scale, void* stream) {
    if (!h) return fail(DSG_E_INVALID, "null handle");
    if (!h->finalized) return fail(DSG_E_STATE, "dsg_set_window_cond before dsg_finalize_weights");
    const int rows = cfg_scale ? 2 * B : B;
    if (B <= 0 || rows > h->Bmax) return fail(DSG_E_INVALID, cfg_scale ? "classifier-free guidance needs max_batch >= 2 * batch" : "batch exceeds max_batch");
    if (!style || !audio || (h->S > 0 && !seed)) return fail(DSG_E_INVALID, "style/seed/audio required");
    if (mask_local && !(mask_batch >= 1 && (B * h->Hl) % mask_batch == 0))
        return fail(DSG_E_INVALID, "mask_local batch must divide B*heads");
    if (h->cfg.variant == 5 && h->seed_last_B != B)
        return fail(DSG_E_STATE, "variant 5: call dsg_set_seed_last with the same batch before dsg_set_window_cond (y['seed_last'])");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(order_after(h, stream));      // the caller's stream may still be producing seed / audio
    const int T = h->T;
    if (mask_local) {
        CHK(upload(h, h->mask, mask_local, (size_t)mask_batch * T));
        h->mb = mask_batch;
        if (cfg_scale && mask_batch > 1) {          // the twins use the same key masks: rows [B, 2B) repeat rows [0, B)
            if (mask_batch != B) return fail(DSG_E_NOT_IMPLEMENTED, "guidance: mask_local batch must be 1 or B");
            CHK(upload(h, h->mask + (size_t)B * T, mask_local, (size_t)B * T));
            h->mb = 2 * B;
        }
        h->nomask = 0;
    } else {
        // `mask=None` (local_attention.py:196-210 skipped): only the causal mask applies and the look-back pads of window 0
        // attend with key = value = -1
        HIPCHK(hipMemsetAsync(h->mask, 1, (size_t)T, h->stream));
        h->mb = 1;
        h->nomask = 1;
    }
    CHK(cond_rows(h, style, seed, audio, 0, B, cfg_scale ? 0 : uncond));
    if (cfg_scale) {
        if (h->cfg.variant == 5)      // the twins see the same closing snippet
            HIPCHK(hipMemcpyAsync(h->c_seed_last + (size_t)B * h->J * h->S, h->c_seed_last, (size_t)B * h->J * h->S * sizeof(float),
                                  hipMemcpyDeviceToDevice, h->stream));
        CHK(cond_rows(h, style, seed, audio, B, B, 1));
        CHK(upload(h, h->cfg_scale, cfg_scale, (size_t)B * sizeof(float)));
    }
    h->cfgB = cfg_scale ? B : 0;
    h->cond_set = true;
    h->condB = rows;
    CHK(order_before(h, stream));     // uploads are enqueued copies: the caller's buffers are free once its stream passes this point
    return 0;
}
extern "C" int dsg_set_window_cond(dsg_handle* h, const float* style, const float* seed, const float* audio,
                                   const uint8_t* mask_local, int mask_batch, int B, int uncond, void* stream) {
    return set_window_cond(h, style, seed, audio, mask_local, mask_batch, B, uncond, nullptr, stream);
}
extern "C" int dsg_set_window_cond_cfg(dsg_handle* h, const float* style, const float* seed, const float* audio,
                                       const uint8_t* mask_local, int mask_batch, int B, const float* scale, void* stream) {
    if (!scale) return fail(DSG_E_INVALID, "dsg_set_window_cond_cfg: scale required (y['scale'])");
    return set_window_cond(h, style, seed, audio, mask_local, mask_batch, B, 0, scale, stream);
}

// ---------------------------------------------------------------------------------------------------------
// Kernel sets.  ONE function decides which kernels a step of B batch rows runs (select_kernels); nothing else in this file
// looks at sizes.  A set is a property of the handle (dsg_set_kernel_set) or, with DSG_KSET_AUTO, a function of the batch
// alone -- never of how many lanes happen to run together -- so the same (handle, batch, seed) gives the same bits whether
// it is sampled by dsg_sample or as one lane of dsg_sample_multi.  What the sets are and the measurements behind the
// automatic rule (MI355X, ZEGGS dims, bf16; profiles/r02_u_kernel_sets.log, r02_i_lanes.log, r02_c_blk_sweep.log):
//
//   set      kernels of one step                                                  dispatches   wins at
//   LATENCY  k_inloc, L x {QKV 16x16, k_attn_mid | k_attn + k_mid, linear2}, head   2 + 3L      one lane, batch <= 2: 106.7 us vs
//                                                                                               122.0 (no k_attn_mid) / 125.5 (TILE)
//   TILE     k_in, k_loc, L x {QKV, k_attn_op | k_attn + out_proj, linear1,          3 + 4L      batch 3 .. 11 (182 vs 187 us at 3,
//            linear2} on 16 x 16 tiles (LayerNorm GEMMs "lean" from 512 rows), head              195 vs 197 at 4, 227 vs 235 at 8)
//   BLOCK    the same with 32-row block GEMMs for QKV / linear1 / linear2 / embedding 3 + 4L      >= 1000 rows in one lane (254 vs 293
//            (dsg_batched.h); out_proj and the pose head stay on 16 x 16 tiles                    us at 16); from 300 rows per lane when
//                                                                                               several lanes share the CUs (4 x 4:
//                                                                                               4691 vs 4400 frames/s, 4 x 16: 8243 vs 6447)
//   STREAM   weight-stationary persistent GEMMs for every K = D / K = ff product of a layer      3 + 5L      >= 2800 rows in one lane (batch 32: 372 ->
//            (dsg_stream.h: W panel in registers, activations global -> LDS, 32x32x16 MFMA):                357 us, batch 64: 611 -> 505); >= 1400 rows per
//            LayerNorm once per row (k_ln_frag), QKV, linear1, linear2, pose head; pose embedding         lane with several lanes (4 x 16: 10.2k -> 11.7k
//            and layer-0 QKV as in BLOCK                                                                  frames/s, 4 x 32: 11.0k -> 13.3k); slower below
//                                                                                                         (4 x 8: 8.3k vs 8.8k, batch 16: 316 vs 240 us)
//   With several lanes sharing the GPU the redundant recompute of LATENCY costs from batch 2 (4 x 2: 3507 frames/s TILE vs
//   3303 LATENCY): dsg_recommend_kernel_set(B, lanes) encodes the multi-lane column; the caller applies it to its lanes.
//   k_attn_op (attention + out_proj + LayerNorm1 in one kernel) replaces k_attn + out_proj in TILE / BLOCK wherever an
//   instantiation exists (bf16, 4 heads, ZEGGS / tiny dims): 1 x 16: 292 -> 254 us, 4 x 4: 4640 -> 5074 frames/s.  Within a set
//   nothing depends on the batch but the grid (and the register budget of the LayerNorm GEMMs from 512 rows, same arithmetic):
//   a row's result is bit-identical whatever the batch it rides in.
// ---------------------------------------------------------------------------------------------------------
struct KernelSel {
    int set = DSG_KSET_TILE;
    bool lat = false;           // LATENCY: fused first kernel (k_inloc) and k_mid / k_attn_mid
    bool attn_in_mid = false;   // ... with the attention inside k_mid (batch 1)
    bool blk = false;           // BLOCK: 32-row block GEMMs
    bool attn_op = false;       // k_attn_op instead of k_attn + out_proj
    bool stream = false;        // STREAM: BLOCK with the weight-stationary persistent GEMMs of dsg_stream.h (LayerNorm + QKV, linear1, linear2, pose head)
    bool ffn = false;           // STREAM (round 4): linear1 + GELU + linear2 + residual + LayerNorm2 in one kernel (k_ffn); QKV of the next layer
                                // and the pose head then read normalised rows: direct streaming GEMMs, no k_ln_frag
    bool ffn16_wide = false;    // ROWS at the DSG+ widths (round 6): k_clip_attn_w (or the direct QKV GEMM + k_attn) + k_ffn<OP> on 16-row tiles
    bool clip_w = false;        // ... with k_clip_attn_w instead of the QKV GEMM + k_attn
    bool ffn_rt2w = false;      // ... with k_ffn<OP> on 32-row blocks: >= 3 lanes whose row tiles together exceed one round of the CUs
    bool ffn16 = false;         // ROWS (round 6): k_ffn on 16-row tiles (one workgroup per row tile), behind k_clip_attn; everything else as BLOCK
    bool ffn_rt4 = false;       // ... on 64-row blocks: 4 lanes x >= 4000 token rows (4 x 64 clips: 981 -> 903 us per step of the 4 lanes; 1 x 64: 376 -> 432,
                                // 4 x 16: 334 -> 392, 4 x 32 even -- profiles/r04_y2_sweep_ffn_rt4_*.log).  Bit-identical to the 32-row form.
    bool ffn_split = false;     // BLOCK (round 4, bf16 ZEGGS / tiny dims): k_ffn split over the hidden dimension (k_ffn_part + k_ffn_ln); direct QKV / pose head
    bool clip_attn = false;     // BLOCK (round 5, with ffn_split): the attention half per (clip, head) -- k_clip_attn (QKV slices + attention + partial out_proj)
                                // + k_ffn_ln (head slabs + residual + LayerNorm1) instead of the QKV GEMM + k_attn_op
    bool xs_frag = false;       // BLOCK / STREAM (bf16, Jp 128 / 1152): the state shadow is fragment-major and the pose embedding streams it (k_ws2<EPI_PARTIAL>:
                                // 8.9 -> 4.3 us at 1424 rows, 29.9 -> 11.0 at 5632; 3.8 -> 4.1 at 356)
};
static bool have_attn_mid(const dsg_handle* h, int B) {
    return B == 1 && h->H == 4 && ((h->D == 256 && h->Tp == 96) || (h->D == 128 && h->Tp == 32));
}
// k_attn_op with all of W_o in registers: bf16 at the ZEGGS / tiny widths (the STREAM set and k_attn_op2 build on it)
static bool have_attn_op_narrow(const dsg_handle* h) {
    return h->prec == DSG_PREC_BF16 && h->H == 4 && ((h->D == 256 && h->Tp == 96) || (h->D == 128 && h->Tp == 32));
}
// ... or k_attn_op_w (W_o streamed in chunks, round 4): the DSG+ widths in bf16, the ZEGGS / tiny widths in fp32
static bool have_attn_op_wide(const dsg_handle* h) {
    if (h->H != 4 || h->prec == DSG_PREC_BF16W2) return false;
    if (h->prec == DSG_PREC_BF16) return (h->D == 384 || h->D == 512) && h->Tp == 160;
    return (h->D == 256 && h->Tp == 96) || (h->D == 128 && h->Tp == 32);
}
static bool have_attn_op(const dsg_handle* h) { return have_attn_op_narrow(h) || have_attn_op_wide(h); }
static bool latency_set_ok(const dsg_handle* h) {
    // k_mid pulls all of W_o (2 D^2 bytes) through every CU: at D = 512 the un-fused sets win (TWH: 219 vs 238 us)
    // and at D = 384 with the K = D GEMMs in one fragment batch (round 4): BEAT 163.9 (TILE) vs 175-179 us (LATENCY) at batch 1
    // (profiles/r04_b_timeline_beat_tile_ch12.json, r04_a_timeline_beat_latency.json).  An explicit dsg_set_kernel_set(LATENCY) still
    // runs k_mid at every width <= 512.
    const int dt = h->D / 64;
    return h->D <= 256 && (dt == 1 || dt == 2 || dt == 4);
}
static bool stream_set_ok(const dsg_handle* h) {
    // k_ws keeps 64 columns x K = D of W per wave in registers (D = 128 / 256), k_ws2 a quarter of K = ff (ff = 128 / 1024);
    // linear1 reads the fragment-major LayerNorm1 rows k_attn_op writes
    return have_attn_op_narrow(h) && (h->D == 256 || h->D == 128) && (h->ff == 1024 || h->ff == 128) && (h->Jp == 1152 || h->Jp == 128);
}
// k_ffn_part + k_ffn_ln exist for the shapes the STREAM set exists for and (round 5) for the DSG+ widths in bf16 (latent_dim 384 / 512, ff 1024:
// 8 ff-splits -- the two-batch weight fragments of these widths do not fit 4), behind k_attn_op_w
static bool ffn_split_wide(const dsg_handle* h) {
    if (h->prec == DSG_PREC_FP32) return have_attn_op_wide(h) && h->ff == 1024 && h->D == 256;      // (fp32, ZEGGS widths: K = 256 is 16 k-blocks of 16)
    return h->prec == DSG_PREC_BF16 && have_attn_op_wide(h) && h->ff == 1024 && (h->D == 384 || h->D == 512);
}
static bool ffn_split_ok(const dsg_handle* h) { return stream_set_ok(h) || ffn_split_wide(h); }
// ROWS at the DSG+ widths (round 6, bf16): k_clip_attn_w (dsg_stream.h: the clip's rows do not fit the LDS next to Q / K / V -- they pass through it in chunks; first
// version: the direct QKV GEMM + k_attn) feeds k_ffn<OP> on one 16-row tile (out_proj + LayerNorm1 as its prologue): no k_attn_op_w, no ff-split, no slabs, no slab-sum
// pass; streamed pose embedding (K over two workgroups) and pose head
static bool rows_wide_ok(const dsg_handle* h) {
    return h->prec == DSG_PREC_BF16 && h->H == 4 && h->Tp == 160 && h->ff == 1024 && (h->D == 384 || h->D == 512);
}
// bf16w2 (round 6): the ROWS pair -- k_clip_attn + k_ffn on one 16-row tile -- exists with the two-register fragments at the ZEGGS / tiny widths
// (not under fused guidance: its last layer runs the QKV GEMM + k_attn_op, which have no bf16w2 form)
static bool rows_w2_ok(const dsg_handle* h) {
    return h->prec == DSG_PREC_BF16W2 && h->H == 4 && ((h->D == 256 && h->Tp == 96 && h->ff == 1024) || (h->D == 128 && h->Tp == 32 && h->ff == 128));
}
// the streamed pose embedding at the DSG+ pose widths (round 6: k_ws2<EPI_PARTIAL, 17 / 18, 2>, K over two workgroups) -- in the ROWS set
static bool xs_frag_wide(const dsg_handle* h, int set) {
#ifdef DSG_X_NO_XS_WIDE      // (A/B: make dev DEVFLAGS=-DDSG_X_NO_XS_WIDE)
    return false;
#endif
    return (set == DSG_KSET_ROWS || set == DSG_KSET_BLOCK) && (h->Jp == 2176 || h->Jp == 2304) && (h->D == 384 || h->D == 512);
}
static int auto_kernel_set(const dsg_handle* h, int B, int lanes) {
    const int rows = B * h->ntok, MT = cdiv(rows, 16);
    const bool s_ok = stream_set_ok(h);       // (bf16, the ZEGGS / tiny widths: STREAM and ROWS exist)
    // Round 6 (one-box sweeps of every set, us per step; profiles/r06_u_*, r06_v_*, r06_w_* before, r06_cb_sweep_rows_gradual.log after the ring's first fill
    // was spread over LayerNorm1): ROWS -- k_ffn on one 16-row tile per workgroup -- wins wherever the row tiles of ALL lanes fit the 256 CUs in one round:
    // 1 x 10 clips 183.0 vs 186.0 (BLOCK), 1 x 12: 184.3 vs 190.3, 1 x 16: 188.0 vs 204, 1 x 24: 203 vs 298 (BLOCK) / 249 (STREAM), 1 x 46 (256 row tiles):
    // 228.7 vs 237.8 (STREAM); 1 x 48 (267 tiles: two rounds): 320 vs 240 -> STREAM; 4 x 3: 184.6 vs 187.1 (BLOCK), 4 x 4: 186.8 vs 194.2, 4 x 5: 189.9 vs
    // 198.7, 2 x 8: 186.4 vs 189.9, 4 x 12 (268 tiles): 225.5 vs 230.4 (STREAM), 4 x 14 (312): 243.4 = 240.7, 4 x 16: 264 vs 250 -> STREAM; 2 x 24: 229 vs
    // 238, 2 x 32: 274 vs 249 -> STREAM.  Below ~900 token rows over all lanes BLOCK's ff-split is even or ahead (1 x 8: 181.9 vs 180.8, 1 x 6: 180.6 vs
    // 177.6, 1 x 5: 180.3 vs 176.4 -- TILE 161.6 --, 2 x 4: 180.5 vs 178.1, 2 x 6: 184.0 = 183.8; 4 x 2: 181.7 vs 183.3, TILE 179.4).
    // (rounds 4-5, STREAM against BLOCK: profiles/r04_n_sweep_sets.log, r05_j_sweep_sets.log)
    if (lanes <= 1) {
        if (s_ok && rows >= 800) return MT <= 256 ? DSG_KSET_ROWS : DSG_KSET_STREAM;
        // (fp32, round 5: k_attn_mid recomputes out_proj per hidden slice, and an fp32 MFMA is 1/16 of a bf16 one -- the un-fused 16 x 16
        // tiles win at batch 1: 208.2 vs 218.5 us per step, profiles/r05_g_bench_fp32_*.log)
        if (B <= 2 && latency_set_ok(h) && h->prec != DSG_PREC_FP32) return DSG_KSET_LATENCY;
        // (DSG+ widths, round 5: with k_ffn_part + k_ffn_ln behind k_attn_op_w BLOCK wins from 4 clips -- BEAT 283 vs 303 us, TWH 310 vs 365; 2 clips: 259 vs 198)
        // (only there: ffn_split_wide() is also true for fp32 at the ZEGGS widths, which has no such measurement -- round-5 advisor)
        // (round 6, DSG+ widths: ROWS -- direct QKV + k_attn + k_ffn<OP> on 16-row tiles -- from 9 clips while the row tiles fit the CUs in one round: BEAT 1 x 16
        //  526 -> 443 us per step, 1 x 24: 701 -> 556, 1 x 26: 806 -> 594, 1 x 8: 370 -> 346; TWH 1 x 16: 609 -> 524, 1 x 24: 835 -> 660, 1 x 8: 417 -> 410; below: BLOCK
        //  (BEAT 1 x 6: 297 vs 305, TWH 1 x 6: 326 vs 366) -- profiles/r06_da_*, r06_db_*)
        //  (... and past one round of the CUs as well: BEAT 1 x 32 clips 702 vs 844 us BLOCK, 1 x 48: 822 vs 1179; TWH 1 x 32: 925 vs 1058 -- profiles/r06_dm_*)
        //  (with k_clip_attn_w -- 24 us at latent_dim 512 whatever the batch -- BLOCK keeps 9 .. 12 TWH clips: 1 x 9: 417 vs 452 us, 1 x 12: 457 vs 466, 1 x 14: 551 vs 472;
        //   BEAT 1 x 9: 356 vs 339 ROWS -- profiles/r06_dv_*)
        if (rows_wide_ok(h) && h->cfgB == 0 && rows >= (h->D == 512 ? 1960 : 1300)) return DSG_KSET_ROWS;
        if (ffn_split_wide(h) && h->prec == DSG_PREC_BF16 && (h->D == 384 || h->D == 512) && h->env_ffn_split != 0 && rows >= 600) return DSG_KSET_BLOCK;
        // (round 6, bf16 ZEGGS widths: BLOCK from 6 clips -- 1 x 6: 177.6 vs 190.1 TILE, 1 x 5: 176.4 vs 161.6, 1 x 4: 173.9 vs 157.4)
        return rows >= (s_ok ? 500 : 1000) ? DSG_KSET_BLOCK : DSG_KSET_TILE;
    }
    if (s_ok && rows >= 850 && lanes * MT > 300) return DSG_KSET_STREAM;
    if (s_ok && rows >= 250 && lanes * rows >= 1000 && lanes * MT <= 300) return DSG_KSET_ROWS;
    // (DSG+ widths with several lanes: ROWS from 4 clips per lane and 16 over all lanes -- BEAT 4 x 4: 399 vs 405, 2 x 8: 412 vs 445, 4 x 8: 627 vs 699; TWH 4 x 4: 509 vs 518,
    //  2 x 8: 514 vs 549, 4 x 8: 794 vs 949; below: BLOCK -- 4 x 3: 342 vs 363, 2 x 4: 297 vs 304, 4 x 2: 297 vs 319)
    if (rows_wide_ok(h) && h->cfgB == 0 && rows >= 600 && lanes * rows >= 2400) return DSG_KSET_ROWS;
    if (B <= 1 && latency_set_ok(h) && h->prec != DSG_PREC_FP32) return DSG_KSET_LATENCY;
    // (2 x 4: BLOCK 178.1 vs ROWS 180.5; 4 x 2: TILE 179.4 vs BLOCK 183.3; 2 x 3: 175.1 = 173.8)
    return rows >= (s_ok ? 250 : 300) ? DSG_KSET_BLOCK : DSG_KSET_TILE;
}
// what DSG_KSET_AUTO resolves to for `lanes` lanes of batch B: the measured table + dsg_config.latency_mode (1 = never LATENCY, 2 = always
// where LATENCY is a sensible choice at all: latent_dim <= 256, see latency_set_ok).  ONE function for select_kernels and
// dsg_recommend_kernel_set (round-4 advisor: the two used to disagree at the DSG+ widths)
static int resolve_auto_set(const dsg_handle* h, int B, int lanes) {
    int set = auto_kernel_set(h, B, lanes);
    if (h->latency_mode == 0 && set == DSG_KSET_LATENCY) set = DSG_KSET_TILE;
    // bf16w2: 16 x 16 tiles at every batch size.  Its weight fragments are twice the bytes, so the redundant W_o of the fused LATENCY
    // kernels costs more than the dispatches it saves: 168 vs 155 us per batch-1 step (profiles/r05_e_*); LATENCY stays selectable
    // (round 6: ... and ROWS -- k_clip_attn + k_ffn with the two-register fragments -- from 800 token rows in one lane / 1400 over several lanes of >= 300:
    //  1 x 16 clips 331 vs 453 us per step, 4 x 4: 330 vs 348, 1 x 8: 315 vs 311, 1 x 4: 310 vs 208 -- profiles/r06_bb_w2_rows.log.  Every workgroup
    //  streams W_o + W1 + W2 as hi + lo pairs, 2.3 MB through one CU's 64 B / clk load path: k_ffn is 23.7 us whatever the batch, r06_bc_marks_b16_rows_w2.log)
    if (h->prec == DSG_PREC_BF16W2) {
        const int rows = B * h->ntok;
        const bool big = rows >= 800 || (lanes > 1 && rows >= 300 && lanes * rows >= 1400);
        set = (rows_w2_ok(h) && h->cfgB == 0 && big) ? DSG_KSET_ROWS : DSG_KSET_TILE;
    }
    if (h->latency_mode == 1 && latency_set_ok(h)) set = DSG_KSET_LATENCY;
    return set;
}
static int select_kernels(const dsg_handle* h, int B, KernelSel& k) {
    int set = h->kset_req;
    if (set == DSG_KSET_AUTO) set = resolve_auto_set(h, B, 1);
    if (set == DSG_KSET_LATENCY && h->D > 512) return fail(DSG_E_NOT_IMPLEMENTED, "kernel set LATENCY: latent_dim > 512");
    if ((set == DSG_KSET_STREAM || set == DSG_KSET_ROWS) && !stream_set_ok(h) && !(set == DSG_KSET_ROWS && (rows_w2_ok(h) || rows_wide_ok(h))))
        return fail(DSG_E_NOT_IMPLEMENTED, "kernel sets STREAM / ROWS: bf16 (ROWS: bf16w2 as well), latent_dim 128 / 256, 4 heads, ff 128 / 1024 only");
    if (set < DSG_KSET_LATENCY || set > DSG_KSET_ROWS) return fail(DSG_E_INVALID, "unknown kernel set");
    if (h->prec == DSG_PREC_BF16W2 && ((set > DSG_KSET_TILE && !(set == DSG_KSET_ROWS && rows_w2_ok(h))) || (set == DSG_KSET_LATENCY && h->D > 256)))
        return fail(DSG_E_NOT_IMPLEMENTED, "precision bf16w2: kernel sets LATENCY (latent_dim <= 256), TILE and ROWS (latent_dim 128 / 256) only");
    k = KernelSel();
    k.set = set;
    k.lat = set == DSG_KSET_LATENCY;
    k.attn_in_mid = k.lat && have_attn_mid(h, B);
    k.stream = set == DSG_KSET_STREAM;
    k.ffn = k.stream;           // (round 4: k_ffn instead of k_ws<GELU> + k_ws2<RESID> + k_ln_frag)
    // ROWS (round 6): BLOCK's first and last kernels around STREAM's per-layer pair, with k_ffn on ONE 16-row tile per workgroup -- no ff-split,
    // no partial slabs, no k_ffn_ln: 3 + 2L dispatches.  Every workgroup streams W_o + W1 + W2 (1.15 MB) for 16 rows, so it pays while the row
    // tiles of all lanes fit the 256 CUs in one round
    k.ffn16 = set == DSG_KSET_ROWS;
    if (k.ffn16) k.ffn = true;
    {
        const int rows = B * h->ntok, e = h->env_ffn_rt4;      // (test hook / A/B: 64-row blocks from this many token rows at any lane count; 0: never)
        k.ffn_rt4 = k.ffn && (e >= 0 ? (e > 0 && rows >= e) : (h->lanes_now >= 4 && rows >= 4000));
    }
    k.blk = set == DSG_KSET_BLOCK || set == DSG_KSET_ROWS || k.stream;      // (STREAM / ROWS: pose embedding and layer-0 QKV as in BLOCK)
    // the wide form (W_o streamed: DSG+ widths, fp32) belongs to BLOCK / STREAM only -- a set's arithmetic never depends on the batch,
    // and at batch 1 its 10 workgroups per layer lose to k_attn + out_proj (BEAT: 200 vs 163 us/step; 16 clips: 3371 vs 2904 frames/s)
    if (k.ffn16 && h->prec == DSG_PREC_BF16W2 && h->cfgB > 0) return fail(DSG_E_NOT_IMPLEMENTED, "precision bf16w2, kernel set ROWS: no fused guidance (TILE has it)");
    k.attn_op = !k.lat && (have_attn_op_narrow(h) || (k.blk && have_attn_op_wide(h)) || (k.ffn16 && rows_w2_ok(h)));
    k.ffn16_wide = k.ffn16 && rows_wide_ok(h);
    if (k.ffn16_wide) {
#if !defined(DSG_X_NO_CLIP_W)      // (A/B: make dev DEVFLAGS=-DDSG_X_NO_CLIP_W)
        k.clip_w = h->H == 4 && h->Tp == 160;
#endif
#if defined(DSG_X_FFN_RT2W)      // (A/B: 0 never, 1 always)
        k.ffn_rt2w = DSG_X_FFN_RT2W != 0;
#else
        k.ffn_rt2w = h->lanes_now >= 3 && h->lanes_now * cdiv(B * h->ntok, 16) > 256;
#endif
        if (h->cfgB > 0) return fail(DSG_E_NOT_IMPLEMENTED, "kernel set ROWS at the DSG+ widths: no fused guidance (BLOCK has it)");
        k.attn_op = false;      // k_attn writes the attention rows, k_ffn<OP> does out_proj + LayerNorm1
    }
    k.xs_frag = k.blk && h->prec == DSG_PREC_BF16 && (h->Jp == 1152 || h->Jp == 128 || xs_frag_wide(h, set));
    if (set == DSG_KSET_BLOCK && ffn_split_ok(h)) k.ffn_split = h->env_ffn_split != 0;      // (A/B: DSG_FFN_SPLIT=0 = linear1 + linear2 + LayerNorm-on-read, round 3)
    k.clip_attn = (k.ffn_split || k.ffn) && have_attn_op_narrow(h) && h->env_clip_attn != 0;      // (A/B: DSG_CLIP_ATTN=0 = QKV GEMM + k_attn_op, round 4)
    if (k.ffn16 && h->prec == DSG_PREC_BF16W2) k.clip_attn = true;

    return 0;
}

// buffers only one kernel set needs, allocated when a call first runs that set (never inside a graph capture / packet recording:
// the callers invoke this right after select_kernels)
static int ensure_set_buffers(dsg_handle* h, const KernelSel& k) {
    if (k.ffn_split && !h->ffn_part) {
        const bool was = h->alloc_uc;
        h->alloc_uc = true;
        const int rc = dalloc(h, &h->ffn_part, (size_t)(ffn_split_wide(h) ? 8 : 4) * h->ffn_slab);      // [S][M_pad][D] fp32, loop-written: uncached like the other activations
        h->alloc_uc = was;
        if (rc) return rc;
    }
    return 0;
}

extern "C" int dsg_set_kernel_set(dsg_handle* h, int set) {
    if (!h) return fail(DSG_E_INVALID, "null handle");
    if (set < DSG_KSET_AUTO || set > DSG_KSET_ROWS) return fail(DSG_E_INVALID, "dsg_set_kernel_set: unknown kernel set");
    if (set == DSG_KSET_LATENCY && h->D > 512) return fail(DSG_E_NOT_IMPLEMENTED, "kernel set LATENCY: latent_dim > 512");
    if ((set == DSG_KSET_STREAM || set == DSG_KSET_ROWS) && !stream_set_ok(h) && !(set == DSG_KSET_ROWS && (rows_w2_ok(h) || rows_wide_ok(h))))
        return fail(DSG_E_NOT_IMPLEMENTED, "kernel sets STREAM / ROWS: bf16 (ROWS: bf16w2 as well), latent_dim 128 / 256, 4 heads, ff 128 / 1024 only");
    // (the same rule as select_kernels: round-5 advisor -- this entry point used to accept LATENCY at latent_dim 384 / 512, and every later call failed)
    if (h->prec == DSG_PREC_BF16W2 && ((set > DSG_KSET_TILE && !(set == DSG_KSET_ROWS && rows_w2_ok(h))) || (set == DSG_KSET_LATENCY && h->D > 256)))
        return fail(DSG_E_NOT_IMPLEMENTED, "precision bf16w2: kernel sets LATENCY (latent_dim <= 256), TILE and ROWS (latent_dim 128 / 256) only");
    if (getenv("DSG_KSET")) {                  // an A/B run pinned the set for the whole process: say so once, keep the pinned set
        static std::atomic<bool> said{false};
        if (set != h->kset_req && !said.exchange(true))
            fprintf(stderr, "libdsg_hip: WARNING: DSG_KSET=%s pins the kernel set of every handle; dsg_set_kernel_set(%d) ignored "
                            "(dsg_get_kernel_set reports the set in force)\n", getenv("DSG_KSET"), set);
        return 0;
    }
    h->kset_req = set;
    return 0;
}
// the set in force for the handle (what dsg_set_kernel_set / DSG_KSET / dsg_clone left): callers that change it for one call
// (sample.generate_clips_streams) restore it afterwards
extern "C" int dsg_get_kernel_set(dsg_handle* h, int* set) {
    if (!h || !set) return fail(DSG_E_INVALID, "null argument");
    *set = h->kset_req;
    return 0;
}
extern "C" int dsg_recommend_kernel_set(dsg_handle* h, int B, int lanes, int* set) {
    if (!h || !set || B <= 0 || lanes <= 0) return fail(DSG_E_INVALID, "dsg_recommend_kernel_set: bad argument");
    *set = resolve_auto_set(h, B, lanes);
    return 0;
}
extern "C" int dsg_last_kernel_set(dsg_handle* h, int* set) {
    if (!h || !set) return fail(DSG_E_INVALID, "null argument");
    if (h->last_kset < 0) return fail(DSG_E_STATE, "no step has run");
    *set = h->last_kset;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// one denoising step = 3 + 4*L dispatches (TILE / BLOCK with k_attn_op; 3 + 5*L without) or 2 + 3*L (LATENCY, batch 1)
// ---------------------------------------------------------------------------------------------------------
struct StepCtx {
    int B;                  // batch rows the kernels run on (with guidance: conditional elements + their twins)
    int out_mode; bool use_ctr; const float* ext_noise; int const_noise;
    int no_noise = 0;       // DDIM with eta = 0: no step adds noise (the pose head skips the Philox draw)
    int clip_x0 = 0;        // clip_denoised=True
    KernelSel ks;           // what select_kernels chose for this call
};

// Every kernel of the denoising step goes through here: a HIP launch on the handle's stream, or -- while dsg_sample is
// recording the step for the AQL path -- an entry of the packet plan (dsg_aql.h).
template <auto K, class A>
static int step_launch(dsg_handle* h, dim3 grid, dim3 block, const A& args) {
#ifndef DSG_EMU
    if (h->aql.recording) {
        const int fence = h->state_fences ? h->fence_next : 0;
        h->fence_next = 0;
        if (!dsg_aql::record(h->aql, (const void*)K, h->stream, grid, block, &args, sizeof(A), fence))
            return fail(DSG_E_RUNTIME, "AQL plan: " + h->aql.err);
        return 0;
    }
#endif
    h->fence_next = 0;
    hipLaunchKernelGGL(K, grid, block, 0, h->stream, args);
    HIPCHK(hipGetLastError());
    return 0;
}

// k-blocks per fragment batch of gemm_body (CH) for a wave whose k range is `per_wave` k-blocks: the fewest batches, then the
// fewest clamped (wasted) loads
static int pick_ch(int per_wave) {
    if (per_wave <= 8) return 8;
    if (per_wave <= 12) return 12;
    if (per_wave <= 16) return 16;
    if (per_wave % 16 == 0) return 16;
    if (per_wave % 12 == 0) return 12;
    return 8;
}
template <class P, int PRO, int EPI, int WN, int WK>
static int launch_gemm(dsg_handle* h, GemmArgs g) {
    if (g.KS == 1) g.kb_per_split = g.KBtot;
    const int NG = g.NT / WN;
    if (NG * WN != g.NT) return fail(DSG_E_INVALID, "gemm: NT not divisible by the workgroup tile");
    if (WK > 1 && (g.KS != 1 || g.KBtot % WK)) return fail(DSG_E_INVALID, "gemm: k-blocks not divisible by the wave split");
    if (g.KS < 1 || g.kb_per_split * g.KS < g.KBtot) return fail(DSG_E_INVALID, "gemm: split-K does not cover K");
    // EPI_PARTIAL / EPI_OUT carry one extra grid row whose first workgroup does the step bookkeeping
    const int extra = (EPI == EPI_PARTIAL || EPI == EPI_OUT) ? 1 : 0;
    g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
    const dim3 grid(xcd_grid_x(NG), g.MT + extra, g.KS);
    // one batch of fragment loads per wave wherever the wave's k range allows it (gemm_body: CH)
    if constexpr (EPI != EPI_PARTIAL) {
        const int ch = pick_ch(std::min(g.kb_per_split, g.KBtot) / WK);
        if (ch == 16) return step_launch<&k_gemm<P, PRO, EPI, WN, WK, 1, 16>>(h, grid, dim3(256), g);
        if (ch == 12) return step_launch<&k_gemm<P, PRO, EPI, WN, WK, 1, 12>>(h, grid, dim3(256), g);
    }
    return step_launch<&k_gemm<P, PRO, EPI, WN, WK, 1>>(h, grid, dim3(256), g);
}
// 32-row x 64-column blocks, K = D whole (dsg_batched.h).  Measured per GEMM in the real batch-16 step
// (profiles/r02_c_blk_sweep.log): QKV -13 us, linear1 -7, embedding -9.5 per step; out_proj +4 and the pose head +3.5 (few,
// long workgroups) stay on the 16 x 16 kernels.  Wider / taller blocks (128-column waves, 64 rows) lose (335 -> 380 / 364 us).
template <class P, int PRO, int EPI>
static int launch_blk(dsg_handle* h, GemmArgs g) {
    g.KS = 1; g.kb_per_split = g.KBtot;
    g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
    const int K = g.KBtot * P::KB;
    if (g.NT % 4) return fail(DSG_E_INVALID, "gemm: NT not divisible by the workgroup tile");
    const dim3 grid(xcd_grid_x(g.NT / 4), cdiv(g.MT, 2) + (EPI == EPI_OUT ? 1 : 0), 1);
    if (K <= 256) return step_launch<&k_gemm_blk<P, PRO, EPI, 256, 1, 2>>(h, grid, dim3(256), g);
    if (K > 512) return fail(DSG_E_NOT_IMPLEMENTED, "k_gemm_blk: K > 512");
    return step_launch<&k_gemm_blk<P, PRO, EPI, 512, 1, 2>>(h, grid, dim3(256), g);
}
template <class P, int EPI>
static int launch_blk_k(dsg_handle* h, GemmArgs g) {
    if (g.KS <= 1) { g.KS = 1; g.kb_per_split = g.KBtot; }
    g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
    if (g.NT % 2) return fail(DSG_E_INVALID, "gemm: NT not divisible by the workgroup tile");
    const int extra = EPI == EPI_PARTIAL ? 1 : 0;
    const dim3 grid(xcd_grid_x(g.NT / 2), cdiv(g.MT, 2) + extra, g.KS);
    // k-blocks per wave in one pass: 8 when the wave's share of the K range is that long (linear2 at ff = 1024 in bf16), else 4
    const int per = cdiv(std::min(g.kb_per_split, g.KBtot), 4);
    if (per > 4) return step_launch<&k_gemm_blk_k<P, EPI, 8>>(h, grid, dim3(256), g);
    return step_launch<&k_gemm_blk_k<P, EPI, 4>>(h, grid, dim3(256), g);
}

// STREAM (dsg_stream.h): persistent grid = 128-column panels x row-block groups.  Groups: enough workgroups for `occ` per CU,
// never more than there are row blocks, a multiple of 8 (one group per XCD slot)
static int ws_groups(int n_panels, int n_blocks, int occ) {
    int G = (256 * occ / n_panels) & ~7;
    G = std::max(G, 8);
    return std::min(G, rup(n_blocks, 8));
}
template <int EPI>
static int launch_ws(dsg_handle* h, GemmArgs g) {
    g.KS = 1; g.kb_per_split = g.KBtot;
    g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
    if (g.NT % 8) return fail(DSG_E_INVALID, "k_ws: N must be a multiple of 128");
    const int P = g.NT / 8, MB = cdiv(g.M, 64);
    g.ws_G = ws_groups(P, MB, 2);
    const int K = g.KBtot * 32;
    if constexpr (EPI == EPI_OUT) {
        // the pose head: one workgroup per (panel, row block), three per CU (round 5; the persistent row-block groups of rounds 3-4 lost on two
        // boxes and are retired -- experiments/README.md) + the bookkeeping workgroup (one XCD round)
        g.ws_G = rup(MB, 8);
        const dim3 grid1(ws_grid_x(P, g.ws_G) + 8);
        if (K == 256) return step_launch<&k_ws<EPI, 16, true>>(h, grid1, dim3(256), g);
        if (K == 128) return step_launch<&k_ws<EPI, 8, true>>(h, grid1, dim3(256), g);
        if (K == 384) return step_launch<&k_ws<EPI, 24, true>>(h, grid1, dim3(256), g);      // (round 6: the DSG+ widths, one workgroup per CU)
        if (K == 512) return step_launch<&k_ws<EPI, 32, true>>(h, grid1, dim3(256), g);
    } else {
        if constexpr (EPI == EPI_QKV) {      // (round 6: the DSG+ widths, one workgroup per CU)
            if (K == 384 || K == 512) {
                g.ws_G = ws_groups(P, MB, 1);
                const dim3 gridw(ws_grid_x(P, g.ws_G));
                if (K == 384) return step_launch<&k_ws<EPI, 24>>(h, gridw, dim3(256), g);
                return step_launch<&k_ws<EPI, 32>>(h, gridw, dim3(256), g);
            }
        }
        const dim3 grid(ws_grid_x(P, g.ws_G));
        if (K == 256) return step_launch<&k_ws<EPI, 16>>(h, grid, dim3(256), g);
        if (K == 128) return step_launch<&k_ws<EPI, 8>>(h, grid, dim3(256), g);
    }
    return fail(DSG_E_NOT_IMPLEMENTED, "k_ws: K must be 128 or 256");
}
// STREAM: LayerNorm once per row (k_ln_frag) -> fragment-major bf16 rows in X1a (free between linear1 and the next attention
// kernel) [+ fp32 rows in Xn], then the GEMM streams them like linear1.  Packet fences of the state (fence_next) belong to the GEMM.
template <int EPI>
static int launch_ln_ws(dsg_handle* h, GemmArgs g) {
    const int fence = h->fence_next;
    h->fence_next = 0;
    GemmArgs l = g;
    l.out = h->X1a;
    const dim3 grid(cdiv(rup(g.M, 64), 16));
    if (g.D == 256) CHK((step_launch<&k_ln_frag<4>>(h, grid, dim3(256), l)));
    else if (g.D == 128) CHK((step_launch<&k_ln_frag<2>>(h, grid, dim3(256), l)));
    else return fail(DSG_E_NOT_IMPLEMENTED, "k_ln_frag: latent_dim must be 128 or 256");
    h->fence_next = fence;
    g.A = h->X1a; g.lda = g.D; g.a_frag = 1; g.X = nullptr; g.Xn = nullptr;
    return launch_ws<EPI>(h, g);
}
template <int EPI>
static int launch_ws2(dsg_handle* h, GemmArgs g) {
    g.KS = 1; g.kb_per_split = g.KBtot;
    g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
    if (g.NT % 4) return fail(DSG_E_INVALID, "k_ws2: N must be a multiple of 64");
    const int K = g.KBtot * 32;
    const int ksplit = (EPI == EPI_PARTIAL && K > 1152) ? 2 : 1;      // (the DSG+ pose widths: K over two workgroups, two slabs for k_loc -- see k_ws2)
    const int P = g.NT / 4 * ksplit, MB = cdiv(g.M, 32);
    g.ws_G = ws_groups(P, MB, 1);
    g.KS = ksplit;
    const dim3 grid(ws_grid_x(P, g.ws_G) + (EPI == EPI_PARTIAL ? 8 : 0));      // EPI_PARTIAL: + the bookkeeping workgroup
    if constexpr (EPI == EPI_RESID) {
        if (K == 1024) return step_launch<&k_ws2<EPI, 16>>(h, grid, dim3(256), g);
        if (K == 128) return step_launch<&k_ws2<EPI, 2>>(h, grid, dim3(256), g);
    } else {
        if (K == 1152) return step_launch<&k_ws2<EPI, 18>>(h, grid, dim3(256), g);
        if (K == 128) return step_launch<&k_ws2<EPI, 2>>(h, grid, dim3(256), g);
        if (K == 2176) return step_launch<&k_ws2<EPI, 17, 2>>(h, grid, dim3(256), g);      // BEAT (round 6)
        if (K == 2304) return step_launch<&k_ws2<EPI, 18, 2>>(h, grid, dim3(256), g);      // TWH
    }
    return fail(DSG_E_NOT_IMPLEMENTED, "k_ws2: K must be 128 / 1024 (linear2) or 128 / 1152 / 2176 / 2304 (pose embedding)");
}

// a K = D GEMM of the un-fused sets: block kernel (BLOCK, the GEMMs it wins), else 16 x 16 tiles -- LayerNorm GEMMs from 512
// rows in the 3-waves-per-SIMD form (k_gemm_lean; tools/b16_lean.sh: batch 8: 250 vs 267 us/step, batch 16: 360 vs 378)
template <class P, int PRO, int EPI>
static int launch_gemm_w(dsg_handle* h, const GemmArgs& g, const KernelSel& ks) {
    constexpr bool blk_wins = EPI == EPI_QKV || EPI == EPI_GELU;
    if constexpr (sizeof(typename P::elem) == 2 && !P::W2 && PRO == PRO_DIRECT && (EPI == EPI_GELU || EPI == EPI_QKV || EPI == EPI_OUT)) {
        if (ks.stream && g.a_frag) return launch_ws<EPI>(h, g);      // STREAM: linear1 on the fragment-major LayerNorm1 rows of k_attn_op
        // (round 6, ROWS at the DSG+ widths: the weight-stationary QKV GEMM with a 192 / 256-register panel, one workgroup per CU -- bit-identical to the block form.
        //  latent 384 from 1200 rows: BEAT 1 x 16 clips 417.9 -> 402.7 us per step, 4 x 8: 616 -> 591, 4 x 16: 1161 -> 1050, 4 x 4 even; latent 512 only with
        //  several lanes: TWH 4 x 8: 799 -> 780, but 1 x 16: 526 -> 540, 1 x 24: 661 -> 682 -- profiles/r06_df_*)
        if constexpr (EPI == EPI_QKV) {
            if (ks.ffn16_wide && g.a_frag && g.M >= 1200 && (g.D == 384 || h->lanes_now >= 2)) return launch_ws<EPI>(h, g);
        }
        // (the streaming pose head below the STREAM sizes loses: BLOCK 1 x 16 clips 207.5 -> 211.6 us per step, 4 x 4: 197.5 -> 205.4 -- profiles/r06_h_*, round 6)
        // (... at the DSG+ widths it wins or is even everywhere in ROWS -- k_ws<OUT, 24 / 32, ONE>: 112 VGPRs + 32 AGPRs, the panel re-read per block: BEAT 1 x 16 clips
        //  358.3 -> 353.8 us per step, 4 x 8: 440.9 -> 423.3, 4 x 16: 615.6 -> 568.3; TWH 1 x 16 even, 4 x 16: 878 -> 844 -- profiles/r06_ds_*; 32 x 32 x 16 MFMAs: the
        //  pose head's last bits differ from the 16 x 16 tiles', so it belongs to the set, not to the lane count)
#ifndef DSG_X_NO_WS_OUT_WIDE
        if constexpr (EPI == EPI_OUT) {
            if (ks.ffn16_wide && g.a_frag) return launch_ws<EPI>(h, g);
        }
#endif
    }
    if constexpr (sizeof(typename P::elem) == 2 && !P::W2 && PRO == PRO_LN && (EPI == EPI_QKV || EPI == EPI_OUT)) {
        if (ks.stream) return launch_ln_ws<EPI>(h, g);                    // STREAM: LayerNorm once per row, then the same streaming GEMM
    }
    if constexpr (EPI != EPI_PARTIAL && !P::W2) {
        if (ks.blk && blk_wins && g.KBtot * P::KB <= 512) return launch_blk<P, PRO, EPI>(h, g);
    }
    if constexpr (PRO == PRO_LN) {
        if (g.M >= 512) {
            GemmArgs gl = g;
            gl.KS = 1; gl.kb_per_split = gl.KBtot;
            gl.inv_ntok = fastdiv_inv(gl.ntok); gl.inv_hd = fastdiv_inv(gl.hd);
            if (gl.NT % 4) return fail(DSG_E_INVALID, "gemm: NT not divisible by the workgroup tile");
            const dim3 grid(xcd_grid_x(gl.NT / 4), gl.MT + (EPI == EPI_OUT ? 1 : 0), 1);
            if constexpr (!P::W2) {      // (bf16w2: 16 k-blocks of two-register weight + activation fragments do not fit: two batches of 8)
                if (pick_ch(gl.KBtot) == 16) return step_launch<&k_gemm_lean<P, EPI, 16>>(h, grid, dim3(256), gl);
            }
            if (pick_ch(gl.KBtot) == 12) return step_launch<&k_gemm_lean<P, EPI, 12>>(h, grid, dim3(256), gl);
            return step_launch<&k_gemm_lean<P, EPI>>(h, grid, dim3(256), gl);
        }
    }
    return launch_gemm<P, PRO, EPI, 4, 1>(h, g);
}
// linear2: K = ff split over the 4 waves of the workgroup
template <class P>
static int launch_gemm_k4(dsg_handle* h, const GemmArgs& g, const KernelSel& ks) {
    if constexpr (sizeof(typename P::elem) == 2 && !P::W2) {
        if (ks.stream && g.a_frag) return launch_ws2<EPI_RESID>(h, g);
    }
    if constexpr (!P::W2) {
        if (ks.blk) return launch_blk_k<P, EPI_RESID>(h, g);
    }
    return launch_gemm<P, PRO_DIRECT, EPI_RESID, 1, 4>(h, g);
}

template <class P, int HD, int NKT>
static int launch_attn_t(dsg_handle* h, const AttnArgs& a) {
    const int nqt = cdiv(a.ntok, 16);
    return step_launch<&k_attn<P, HD, NKT>>(h, dim3(nqt, a.H, a.B), dim3(64), a);
}
template <class P>
static int launch_attn(dsg_handle* h, const AttnArgs& a) {
    const int key = h->hd * 1000 + h->Tp / 16;
    switch (key) {
        case 32 * 1000 + 2: return launch_attn_t<P, 32, 2>(h, a);
        case 64 * 1000 + 2: return launch_attn_t<P, 64, 2>(h, a);
        case 64 * 1000 + 6: return launch_attn_t<P, 64, 6>(h, a);
        case 96 * 1000 + 10: return launch_attn_t<P, 96, 10>(h, a);
        case 128 * 1000 + 10: return launch_attn_t<P, 128, 10>(h, a);
        default: return fail(DSG_E_NOT_IMPLEMENTED, "no attention instantiation for (head_dim, tokens) = (" +
                                                        std::to_string(h->hd) + ", " + std::to_string(h->Tp) + ")");
    }
}

// k_loc / k_inloc are instantiated per (local head dim, window)
#define DSG_LOC_DISPATCH(KERNEL, ARGS, GRID)                                                                         \
    do {                                                                                                             \
        const int key_ = h->hdl * 100 + h->W;                                                                        \
        if (key_ == 32 * 100 + 11) CHK((step_launch<&KERNEL<P, 32, 11>>(h, GRID, dim3(256), ARGS)));                 \
        else if (key_ == 48 * 100 + 15) CHK((step_launch<&KERNEL<P, 48, 15>>(h, GRID, dim3(256), ARGS)));            \
        else if (key_ == 64 * 100 + 15) CHK((step_launch<&KERNEL<P, 64, 15>>(h, GRID, dim3(256), ARGS)));            \
        else if (key_ == 16 * 100 + 11) CHK((step_launch<&KERNEL<P, 16, 11>>(h, GRID, dim3(256), ARGS)));            \
        else if (key_ == 8 * 100 + 15) CHK((step_launch<&KERNEL<P, 8, 15>>(h, GRID, dim3(256), ARGS)));              \
        else return fail(DSG_E_NOT_IMPLEMENTED, "no local-attention instantiation for (head dim, window) = (" +      \
                                                    std::to_string(h->hdl) + ", " + std::to_string(h->W) + ")");     \
    } while (0)

// k_loc with TWO waves per (head, window, clip) instead of four (round 6): see the call site
#define DSG_LOC1_DISPATCH(ARGS, GRID)                                                                                \
    do {                                                                                                             \
        const int key_ = h->hdl * 100 + h->W;                                                                        \
        if (key_ == 32 * 100 + 11) CHK((step_launch<&k_loc<P, 32, 11, 128>>(h, GRID, dim3(128), ARGS)));               \
        else if (key_ == 48 * 100 + 15) CHK((step_launch<&k_loc<P, 48, 15, 128>>(h, GRID, dim3(128), ARGS)));          \
        else if (key_ == 64 * 100 + 15) CHK((step_launch<&k_loc<P, 64, 15, 128>>(h, GRID, dim3(128), ARGS)));          \
        else if (key_ == 16 * 100 + 11) CHK((step_launch<&k_loc<P, 16, 11, 128>>(h, GRID, dim3(128), ARGS)));          \
        else if (key_ == 8 * 100 + 15) CHK((step_launch<&k_loc<P, 8, 15, 128>>(h, GRID, dim3(128), ARGS)));            \
        else return fail(DSG_E_NOT_IMPLEMENTED, "no local-attention instantiation for (head dim, window) = (" +      \
                                                    std::to_string(h->hdl) + ", " + std::to_string(h->W) + ")");     \
    } while (0)

template <class P>
static int launch_mid(dsg_handle* h, const MidArgs& a) {
    const dim3 grid(xcd_grid_x(a.ff / 64), a.MT);
    switch (h->D / 64) {
        case 1: return step_launch<&k_mid<P, 1>>(h, grid, dim3(256), a);
        case 2: return step_launch<&k_mid<P, 2>>(h, grid, dim3(256), a);
        case 4: return step_launch<&k_mid<P, 4>>(h, grid, dim3(256), a);
        default: break;
    }
    if constexpr (!P::W2) {      // (bf16w2: two-register weight fragments, LATENCY up to latent_dim 256 -- select_kernels)
        if (h->D == 384) return step_launch<&k_mid<P, 6>>(h, grid, dim3(256), a);
        if (h->D == 512) return step_launch<&k_mid<P, 8>>(h, grid, dim3(256), a);
    }
    return fail(DSG_E_NOT_IMPLEMENTED, "k_mid: latent_dim / 64 must be 1, 2, 4, 6 or 8 (bf16w2: 1, 2 or 4)");
}
// Attention fused into k_mid: one batch element, 4 heads (wave = head), D <= 256.  Bit-identical to k_attn + k_mid and
// one dispatch less per layer.  With HIP launches it does not pay (152.4 vs 151.3 us/step): the four heads' strided
// K / V^T fragment loads queue on ONE CU's load path (38 loads in ~6100 cycles) instead of running on 24 otherwise idle
// CUs, which costs what the saved launch gains.  With the AQL submission the balance tips (136.0 vs 140.8 us/step,
// measured twice on the same box), so it is what batch 1 runs (batch 2 of the same set runs k_attn + k_mid: tests compare the two).
template <class P>
static int launch_attn_mid(dsg_handle* h, const AttnMidArgs& a) {
    const dim3 grid(xcd_grid_x(a.mid.ff / 64), a.mid.MT);
    if (h->D == 256 && h->Tp == 96) return step_launch<&k_attn_mid<P, 4, 6>>(h, grid, dim3(256), a);
    if (h->D == 128 && h->Tp == 32) return step_launch<&k_attn_mid<P, 2, 2>>(h, grid, dim3(256), a);
    return fail(DSG_E_NOT_IMPLEMENTED, "no fused attention+mid instantiation");
}

static StepTables step_tables(const dsg_handle* h) {
    StepTables st;
    st.tmodel = h->st_tmodel; st.c1 = h->st_c[0]; st.c2 = h->st_c[1]; st.c3 = h->st_c[2]; st.c4 = h->st_c[3]; st.c5 = h->st_c[4];
    return st;
}

template <class P>
static int run_step(dsg_handle* h, const StepCtx& c) {
    const int B = c.B, D = h->D, T = h->T, ntok = h->ntok;
    const int Min = B * T, M = B * ntok;
    const int MTin = cdiv(Min, 16), MT = cdiv(M, 16);
    const int KB = P::KB;
    const KernelSel& ks = c.ks;
    GemmArgs z;
    memset(&z, 0, sizeof(z));
    z.KS = 1; z.kb_per_split = 0; z.B = B; z.ntok = ntok; z.Tp = h->Tp; z.H = h->H; z.hd = h->hd; z.T = T; z.J = h->J; z.Jp = h->Jp;
    z.Jq = h->Jq; z.D = D; z.inv_ntok4 = fastdiv_inv(rup(ntok, 4));

    LocArgs la;
    memset(&la, 0, sizeof(la));
    // split-K of the pose-embedding GEMM across workgroups: one split per 256 pose features for the 16 x 16 tile kernel; the
    // block kernel splits K over its 4 waves already, so 2 workgroup splits keep a wave's share at <= 8 k-blocks (one batch of
    // loads) without fragmenting the work 5 ways
    const int ks_in = ks.xs_frag ? (h->Jp > 1152 ? 2 : 1) : ((ks.blk && !P::W2) ? std::min(h->KSin, 2) : h->KSin);      // (streamed embedding: K stays whole, see k_ws2; bf16w2: the 16 x 16 tiles)
    la.partial = h->partial; la.KS = ks_in; la.Min_pad = MTin * 16; la.Cf = h->Cf; la.TE2 = h->TE2; la.TE = h->TE;
    la.emb1 = h->emb1; la.ctl = c.use_ctr ? h->ctl : nullptr; la.t_arr = h->t_arr;
    la.rcos = h->rcos; la.rsin = h->rsin; la.mask = h->mask; la.mb = h->mb; la.inv_mask_div = fastdiv_inv((int)((long long)B * h->Hl / h->mb)); la.B = B; la.T = T; la.D = D; la.Hl = h->Hl;
    la.hd = h->hdl; la.W = h->W; la.X0 = h->X0; la.X0a = h->X0a; la.nomask = h->nomask;
    la.x0a_frag = ((ks.stream || ks.clip_attn || ks.ffn16_wide) && sizeof(typename P::elem) == 2) ? (P::W2 ? 2 : 1) : 0;      // (bf16w2: as a hi + lo pair)
    h->fence_next = 1;         // the first packet of a step reads the state the previous step's last packet wrote (state_fences)
    if (ks.lat) {              // pose embedding + local attention in one launch (in the batched sets it loses: 1 x 16 clips 192.1 -> 198.5 us per
                               // step, 1 x 64: 258 -> 291, 4 x 4 even -- profiles/r06_ab_*, round 6)
        InLocArgs a;
        a.xs = is_bf16(h) ? h->xsA : (void*)h->xs32; a.Jp = h->Jp; a.Wp = h->Wp_in; a.KBtot = h->Jp / KB;
        a.loc = la; a.ctl_upd = c.use_ctr ? h->ctl : nullptr; a.st = step_tables(h); a.n_tab = h->n_run;
        DSG_LOC_DISPATCH(k_inloc, a, dim3(h->Hl, T / h->W, B + 1));
    } else {
        {   // k_in: partial[s] = xs[:, chunk s] . Wfold[:, chunk s]^T
            GemmArgs g = z;
            g.M = Min; g.MT = MTin; g.NT = D / 16; g.KBtot = h->Jp / KB; g.KS = ks_in; g.Wp = h->Wp_in;
            g.kb_per_split = cdiv(g.KBtot, g.KS);
            g.A = is_bf16(h) ? h->xsA : (void*)h->xs32; g.lda = h->Jp;
            g.out = h->partial; g.ldo = D;
            g.ctl = c.use_ctr ? h->ctl : nullptr; g.st = step_tables(h); g.n_tab = h->n_run;
            bool done = false;
            if constexpr (sizeof(typename P::elem) == 2 && !P::W2) {
                if (ks.xs_frag) { g.a_frag = 1; CHK(launch_ws2<EPI_PARTIAL>(h, g)); done = true; }      // the state shadow is fragment-major
            }
            if constexpr (!P::W2) {
                if (!done && ks.blk) { CHK((launch_blk_k<P, EPI_PARTIAL>(h, g))); done = true; }
            }
            if (!done) CHK((launch_gemm<P, PRO_DIRECT, EPI_PARTIAL, 4, 1>(h, g)));
        }
        // Round 6: from 2048 (head, window, clip) items TWO waves each instead of a 256-thread workgroup: the kernel is one ~5 us chain of dependent
        // phases per item, and 4096 workgroups of 4 waves (64 clips) need two rounds of the CUs' 8 workgroup slots where 4096 x 2 waves are all
        // resident at once (32 per CU) -- 1 x 64 clips 249.1 -> 245.0 us per step (one wave each: 246.4), 4 x 64: 705 -> 692-702 (29.2-29.6 k frames/s);
        // k_loc itself 13.3 -> 10.0 us at 5696 rows with one wave; below 2048 items the 4-wave form is ahead (16 clips: 184.1 vs 186.0).  Same arithmetic
        // (the matrix-instruction tail is one wave's work in every form): bit-identical.  profiles/r06_cj_*, r06_cl_*
        if (h->Hl * (T / h->W) * B >= h->env_loc64_from) DSG_LOC1_DISPATCH(la, dim3(h->Hl, T / h->W, B));
        else DSG_LOC_DISPATCH(k_loc, la, dim3(h->Hl, T / h->W, B));
    }
    for (int l = 0; l < h->L; ++l) {
        const Layer& ly = h->layers[l];
        // (guidance: the last layer leaves pre2 to the two-pass pose head, i.e. it runs the round-3 feed-forward kernels, which read k_attn_op's rows)
        const bool clip_l = ks.clip_attn && (l < h->L - 1 || h->cfgB == 0);
        bool clip_w_done = false;
        if constexpr (sizeof(typename P::elem) == 2 && !P::W2) {
            if (ks.clip_w) {      // ROWS at the DSG+ widths (round 6): QKV slices + attention per (clip, head) -- k_clip_attn_w (dsg_stream.h); X0a holds the embedding output /
                                  // LayerNorm2 of the previous layer, fragment-major
                ClipAttnArgs a;
                a.X = h->X0a; a.Wqkv = ly.Wqkv; a.bqkv = ly.bqkv; a.out = h->attn; a.B = B; a.ntok = ntok;
                // latent_dim 384: ONE pass over the rows, three column tiles on waves 0 - 1 (144 weight registers fit: 236 VGPRs)
                // (384 in the two-pass form: 1 x 16 clips 371.3 vs 360.0 us per step, 4 x 16: 614 vs 604 -- profiles/r06_dq_*)
                if (D == 384) CHK((step_launch<&k_clip_attn_w<6, 10, 4, true>>(h, dim3(4, B), dim3(512), a)));
                // 512: one pass as well -- 192 weight registers, so the bias waits in the LDS and the A fragments have no look-ahead (254 VGPRs); the two-pass form
                // (Q / K, then V in pairs on waves 0 - 3) reads the rows from the LDS 1.5 times and reloads weights in between: TWH 1 x 16 clips 475.1 vs 472.9 us per
                // step, 4 x 8: 601 vs 585, 4 x 16: 836.7 vs 830.1, bit-identical -- profiles/r06_dw_*
#ifdef DSG_X_TWH_TWOPASS
                else CHK((step_launch<&k_clip_attn_w<8, 10, 2, false>>(h, dim3(4, B), dim3(512), a)));
#else
                else CHK((step_launch<&k_clip_attn_w<8, 10, 2, true>>(h, dim3(4, B), dim3(512), a)));
#endif
                clip_w_done = true;
            }
        }
        if (!clip_l && !clip_w_done) {   // QKV projection (LayerNorm2 of the previous layer applied on read)
            GemmArgs g = z;
            g.M = M; g.MT = MT; g.NT = 3 * D / 16; g.KBtot = D / KB; g.Wp = ly.Wqkv; g.bias = ly.bqkv;
            g.q = h->q; g.k = h->k; g.vt = h->vt;
            if (l == 0) {
                g.A = h->X0a; g.lda = D; g.a_frag = la.x0a_frag;
                CHK((launch_gemm_w<P, PRO_DIRECT, EPI_QKV>(h, g, ks)));
            } else if (ks.ffn || ks.ffn_split) {      // k_ffn / k_ffn_ln left LayerNorm2(previous layer) in X0a, fragment-major
                g.A = h->X0a; g.lda = D; g.a_frag = 1;
                CHK((launch_gemm_w<P, PRO_DIRECT, EPI_QKV>(h, g, ks)));
            } else {
                g.X = h->pre2; g.ln_g = h->layers[l - 1].g2; g.ln_b = h->layers[l - 1].be2; g.Xn = h->Xn;
                CHK((launch_gemm_w<P, PRO_LN, EPI_QKV>(h, g, ks)));
            }
        }
        if (!ks.attn_in_mid && !ks.attn_op && !clip_w_done) {   // attention
            AttnArgs a;
            memset(&a, 0, sizeof(a));
            a.q = h->q; a.k = h->k; a.vt = h->vt; a.out = h->attn; a.B = B; a.H = h->H; a.ntok = ntok; a.Tp = h->Tp;
            a.D = D;
            CHK(launch_attn<P>(h, a));
        }
        if (ks.lat) {      // [attention +] out_proj + residual + LayerNorm1 + linear1 slice + GELU
            MidArgs a;
            memset(&a, 0, sizeof(a));
            a.A = h->attn; a.R = l == 0 ? h->X0 : h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln_g = ly.g1; a.ln_b = ly.be1;
            a.W1 = ly.W1; a.b1 = ly.b1; a.X1 = h->X1; a.hidden = h->hidden; a.M = M; a.MT = MT; a.ff = h->ff;
            if (ks.attn_in_mid) {
                AttnMidArgs am;
                am.mid = a; am.q = h->q; am.k = h->k; am.vt = h->vt; am.ntok = ntok; am.Tp = h->Tp;
                CHK(launch_attn_mid<P>(h, am));
            } else {
                CHK(launch_mid<P>(h, a));
            }
        } else if (ks.attn_op) {
          if constexpr (P::W2) {         // bf16w2 in the ROWS set (round 6): k_clip_attn + k_ffn on one 16-row tile, two-register fragments
            if (!clip_l || !ks.ffn16) return fail(DSG_E_NOT_IMPLEMENTED, "precision bf16w2: k_clip_attn + k_ffn in the ROWS set only");
            ClipAttnArgs a;
            a.X = h->X0a; a.Wqkv = ly.Wqkv; a.bqkv = ly.bqkv; a.out = h->attn; a.B = B; a.ntok = ntok;
            if (D == 256) CHK((step_launch<&k_clip_attn<P, 4, 6>>(h, dim3(4, B), dim3(384), a)));
            else CHK((step_launch<&k_clip_attn<P, 2, 2>>(h, dim3(4, B), dim3(128), a)));
            FfnArgs f;
            memset(&f, 0, sizeof(f));
            f.W1 = ly.W1; f.b1 = ly.b1; f.W2 = ly.W2; f.b2 = ly.b2; f.ln_g = ly.g2; f.ln_b = ly.be2; f.Xn = h->Xn; f.Xa = h->X0a; f.M = M; f.MT = MT;
            f.A = h->attn; f.R = l == 0 ? h->X0 : h->Xn; f.Wo = ly.Wo; f.bo = ly.bo; f.ln1_g = ly.g1; f.ln1_b = ly.be1; f.X1 = h->X1;
            if (D == 256) CHK((step_launch<&k_ffn<P, 4, 16, 1, 8, 2, true, true, 12>>(h, dim3(MT), dim3(512), f)));
            else CHK((step_launch<&k_ffn<P, 2, 2, 1, 4, 2, false, true, 8>>(h, dim3(MT), dim3(256), f)));
            continue;
          } else {                       // (bf16w2 runs LATENCY / TILE without k_attn_op: select_kernels)
            // attention + out_proj + residual + LayerNorm1 in one kernel per (query tile, batch element); linear1 reads the
            // normalised rows in the GEMM type
            if (clip_l) {
                // round 5: per (clip, head) -- QKV slices + attention in one kernel (k_clip_attn); out_proj + residual + LayerNorm1 are the
                // prologue of k_ffn_part below (OP): Q / K / V never leave the CU and the QKV GEMM is gone as a dispatch
                if constexpr (sizeof(typename P::elem) == 2) {
                    ClipAttnArgs a;
                    a.X = h->X0a; a.Wqkv = ly.Wqkv; a.bqkv = ly.bqkv; a.out = h->attn; a.B = B; a.ntok = ntok;
                    if (D == 256) CHK((step_launch<&k_clip_attn<P, 4, 6>>(h, dim3(4, B), dim3(384), a)));
                    else CHK((step_launch<&k_clip_attn<P, 2, 2>>(h, dim3(4, B), dim3(128), a)));
                }
            } else {
                AttnOpArgs a;
                a.q = h->q; a.k = h->k; a.vt = h->vt; a.R = l == 0 ? h->X0 : h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln_g = ly.g1; a.ln_b = ly.be1;
                a.X1 = h->X1; a.X1a = h->X1a; a.B = B; a.ntok = ntok; a.Tp = h->Tp;
                const dim3 grid(cdiv(ntok, 16), B);
                if constexpr (sizeof(typename P::elem) == 2) {
                    // (k_attn_op2 -- two query tiles per workgroup -- is retired: since round 5 this branch only serves the last layer under fused
                    //  guidance and DSG_CLIP_ATTN=0; experiments/dsg_rejected_kernels.h)
                    if (D == 256 && h->Tp == 96) CHK((step_launch<&k_attn_op<P, 4, 6>>(h, grid, dim3(256), a)));
                    else if (D == 128 && h->Tp == 32) CHK((step_launch<&k_attn_op<P, 2, 2>>(h, grid, dim3(256), a)));
                    else if (D == 384) CHK((step_launch<&k_attn_op_w<P, 6, 10>>(h, grid, dim3(256), a)));
                    else CHK((step_launch<&k_attn_op_w<P, 8, 10>>(h, grid, dim3(256), a)));
                } else {
                    if (D == 256) CHK((step_launch<&k_attn_op_w<P, 4, 6>>(h, grid, dim3(256), a)));
                    else CHK((step_launch<&k_attn_op_w<P, 2, 2>>(h, grid, dim3(256), a)));
                }
            }
            // (guidance: the last layer leaves pre2 to the two-pass pose head k_gemm_cfg, which normalises on read)
            if (ks.ffn && !(h->cfgB > 0 && l == h->L - 1)) {      // linear1 + GELU + linear2 + residual + LayerNorm2 (k_ffn): Xn fp32 + X0a in the GEMM type
                FfnArgs a;
                memset(&a, 0, sizeof(a));
                a.A = h->X1a; a.R = h->X1; a.W1 = ly.W1; a.b1 = ly.b1; a.W2 = ly.W2; a.b2 = ly.b2; a.ln_g = ly.g2; a.ln_b = ly.be2;
                a.Xn = h->Xn; a.Xa = h->X0a; a.M = M; a.MT = MT;
                if (clip_l) {      // out_proj + residual + LayerNorm1 as the prologue (OP): A = the attention rows
                    a.A = h->attn; a.R = l == 0 ? h->X0 : h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln1_g = ly.g1; a.ln1_b = ly.be1; a.X1 = h->X1;
                }
                if constexpr (sizeof(typename P::elem) == 2) {
                    const dim3 grid(cdiv(MT, 2));
                    if (clip_l) {
                        // (the weights of both phases on one rolling ring of fragments: 12 slots on 64-row blocks, 32 on 32-row blocks; the double-buffered
                        //  groups of rounds 4-5 are retired)
                        if (ks.ffn16) {         // ROWS: one 16-row tile per workgroup
                            if (D == 256) CHK((step_launch<&k_ffn<P, 4, 16, 1, 8, 2, true, true, 32>>(h, dim3(MT), dim3(512), a)));
                            else CHK((step_launch<&k_ffn<P, 2, 2, 1, 4, 2, false, true>>(h, dim3(MT), dim3(256), a)));
                            continue;
                        }
                        if (D == 256 && ks.ffn_rt4) CHK((step_launch<&k_ffn<P, 4, 16, 4, 8, 1, true, true, 12>>(h, dim3(cdiv(MT, 4)), dim3(512), a)));
                        else if (D == 256) CHK((step_launch<&k_ffn<P, 4, 16, 2, 8, 2, true, true, 32>>(h, grid, dim3(512), a)));
                        else CHK((step_launch<&k_ffn<P, 2, 2, 2, 4, 2, false, true>>(h, grid, dim3(256), a)));
                        continue;
                    }
                    // 64 rows per workgroup (bit-identical: same waves, same k order) when several lanes fill the GPU with large batches:
                    // half the weight bytes through the CUs' load paths per row, half the workgroups (ffn_rt4 in select_kernels)
                    if (D == 256 && ks.ffn_rt4) CHK((step_launch<&k_ffn<P, 4, 16, 4, 8, 1, true>>(h, dim3(cdiv(MT, 4)), dim3(512), a)));
                    else if (D == 256) CHK((step_launch<&k_ffn<P, 4, 16, 2, 8, 2, true>>(h, grid, dim3(512), a)));
                    else CHK((step_launch<&k_ffn<P, 2, 2, 2, 4>>(h, grid, dim3(256), a)));
                }
                continue;
            }
            // BLOCK below the STREAM threshold (round 4): k_ffn split 4 ways over the hidden dimension + the slab sum / LayerNorm2 pass; the
            // next QKV projection and the pose head become direct GEMMs.  (Guidance: the last layer leaves pre2 to k_gemm_cfg, which
            // normalises on read.)  profiles/r04_q_*: 1 x 16 230.9 -> 218.8 us, 4 x 4 222.4 -> 214.4, 4 x 8 288.9 -> 240.8.
            if (ks.ffn_split && (l < h->L - 1 || h->cfgB == 0)) {
                if constexpr (sizeof(typename P::elem) == 2) {
                    FfnPartArgs a;
                    memset(&a, 0, sizeof(a));
                    a.A = h->X1a; a.W1 = ly.W1; a.b1 = ly.b1; a.W2 = ly.W2; a.part = h->ffn_part; a.slab = h->ffn_slab; a.M = M; a.MT = MT;
                    if (clip_l) {      // out_proj + residual + LayerNorm1 as the prologue (OP)
                        a.A = h->attn; a.R = l == 0 ? h->X0 : h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln_g = ly.g1; a.ln_b = ly.be1; a.X1 = h->X1;
                    }
                    FfnLnArgs b;
                    b.part = h->ffn_part; b.slab = h->ffn_slab; b.R = h->X1; b.b2 = ly.b2; b.ln_g = ly.g2; b.ln_b = ly.be2; b.Xn = h->Xn; b.Xa = h->X0a; b.M = M;
                    if (D == 256) {
                        if (clip_l) CHK((step_launch<&k_ffn_part<P, 4, 16, 2, 4, 4, true>>(h, dim3(cdiv(MT, 2) * 4), dim3(256), a)));
                        else CHK((step_launch<&k_ffn_part<P, 4, 16, 2, 4, 4>>(h, dim3(cdiv(MT, 2) * 4), dim3(256), a)));
                        CHK((step_launch<&k_ffn_ln<P, 4, 4, 8>>(h, dim3(cdiv(M, 8)), dim3(128), b)));
                    } else if (D == 384) {      // DSG+ widths (round 5): 8 ff-splits, one row tile per workgroup-pair batch of fragments
                        CHK((step_launch<&k_ffn_part<P, 6, 16, 2, 4, 8>>(h, dim3(cdiv(MT, 2) * 8), dim3(256), a)));
                        CHK((step_launch<&k_ffn_ln<P, 6, 8, 8>>(h, dim3(cdiv(M, 8)), dim3(128), b)));
                    } else if (D == 512) {
                        CHK((step_launch<&k_ffn_part<P, 8, 16, 2, 4, 8>>(h, dim3(cdiv(MT, 2) * 8), dim3(256), a)));
                        CHK((step_launch<&k_ffn_ln<P, 8, 8, 8>>(h, dim3(cdiv(M, 8)), dim3(128), b)));
                    } else {
                        if (clip_l) CHK((step_launch<&k_ffn_part<P, 2, 2, 2, 4, 2, true>>(h, dim3(cdiv(MT, 2) * 2), dim3(256), a)));
                        else CHK((step_launch<&k_ffn_part<P, 2, 2, 2, 4, 2>>(h, dim3(cdiv(MT, 2) * 2), dim3(256), a)));
                        CHK((step_launch<&k_ffn_ln<P, 2, 2, 8>>(h, dim3(cdiv(M, 8)), dim3(128), b)));
                    }
                } else {
                    // fp32 at the ZEGGS widths (round 5, round-4 verdict item 7): the same split, 8 ways (16 k-blocks of 16 per K = 256 operand)
                    FfnPartArgs a;
                    memset(&a, 0, sizeof(a));
                    a.A = h->X1a; a.W1 = ly.W1; a.b1 = ly.b1; a.W2 = ly.W2; a.part = h->ffn_part; a.slab = h->ffn_slab; a.M = M; a.MT = MT;
                    FfnLnArgs b;
                    b.part = h->ffn_part; b.slab = h->ffn_slab; b.R = h->X1; b.b2 = ly.b2; b.ln_g = ly.g2; b.ln_b = ly.be2; b.Xn = h->Xn; b.Xa = h->X0a; b.M = M;
                    CHK((step_launch<&k_ffn_part<P, 4, 16, 2, 4, 8>>(h, dim3(cdiv(MT, 2) * 8), dim3(256), a)));
                    CHK((step_launch<&k_ffn_ln<P, 4, 8, 8>>(h, dim3(cdiv(M, 8)), dim3(128), b)));
                }
                continue;
            }
            {   // linear1 + GELU -> hidden
                GemmArgs g = z;
                g.M = M; g.MT = MT; g.NT = h->ff / 16; g.KBtot = D / KB; g.Wp = ly.W1; g.bias = ly.b1;
                g.A = h->X1a; g.lda = D; g.a_frag = 1; g.out = h->hidden; g.ldo = h->ff; g.out_frag = 1;
                CHK((launch_gemm_w<P, PRO_DIRECT, EPI_GELU>(h, g, ks)));
            }
          }
        } else if (ks.ffn16_wide) {
            if constexpr (sizeof(typename P::elem) == 2 && !P::W2) {
                FfnArgs a;
                memset(&a, 0, sizeof(a));
                a.W1 = ly.W1; a.b1 = ly.b1; a.W2 = ly.W2; a.b2 = ly.b2; a.ln_g = ly.g2; a.ln_b = ly.be2; a.Xn = h->Xn; a.Xa = h->X0a; a.M = M; a.MT = MT;
                a.A = h->attn; a.R = l == 0 ? h->X0 : h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln1_g = ly.g1; a.ln1_b = ly.be1; a.X1 = h->X1;
                // latent_dim 384: W_o (36 fragments per wave) waits in registers as at the ZEGGS widths; 512: 64 fragments do not fit -- W_o leads the weight ring
                // Round 6: with >= 3 lanes whose row tiles together need more than one round of the 256 CUs, 32-row blocks -- W_o | W1 | W2 (1.8 / 2.5 MB) streamed once
                // per 32 rows; W_o leads the ring at both widths (20 / 12 slots: 250 VGPRs), at 512 the fp32 LayerNorm1 rows wait in X1 instead of the LDS.
                // Bit-identical to the 16-row form.  BEAT 4 x 16 clips 946 -> 825 us per step, 4 x 8: 542 -> 516; 1 x 16: 375 -> 412, 4 x 4: 375 -> 398, 2 x 16 even
                // (profiles/r06_dj_*)
                if (ks.ffn_rt2w) {
                    if (D == 384) CHK((step_launch<&k_ffn<P, 6, 16, 2, 8, 2, true, true, 20, true>>(h, dim3(cdiv(MT, 2)), dim3(512), a)));
                    else CHK((step_launch<&k_ffn<P, 8, 16, 2, 8, 2, true, true, 12, true>>(h, dim3(cdiv(MT, 2)), dim3(512), a)));
                } else if (D == 384) CHK((step_launch<&k_ffn<P, 6, 16, 1, 8, 2, true, true, 24>>(h, dim3(MT), dim3(512), a)));      // (W_o leading the ring here too: 353.1 -> 357.8 us per step at 16 clips, r06_ea)
                else CHK((step_launch<&k_ffn<P, 8, 16, 1, 8, 2, true, true, 24, true>>(h, dim3(MT), dim3(512), a)));
                continue;
            }
        } else {
            {   // out_proj + residual -> pre1
                GemmArgs g = z;
                g.M = M; g.MT = MT; g.NT = D / 16; g.KBtot = D / KB; g.Wp = ly.Wo; g.bias = ly.bo;
                g.A = h->attn; g.lda = D; g.a_frag = 1; g.out = h->pre1; g.ldo = D; g.R = l == 0 ? h->X0 : h->Xn;
                CHK((launch_gemm_w<P, PRO_DIRECT, EPI_RESID>(h, g, ks)));
            }
            {   // LayerNorm1-on-read + linear1 + GELU -> hidden ; X1 = LN1(pre1)
                GemmArgs g = z;
                g.M = M; g.MT = MT; g.NT = h->ff / 16; g.KBtot = D / KB; g.Wp = ly.W1; g.bias = ly.b1;
                g.X = h->pre1; g.ln_g = ly.g1; g.ln_b = ly.be1; g.Xn = h->X1; g.out = h->hidden; g.ldo = h->ff; g.out_frag = 1;
                CHK((launch_gemm_w<P, PRO_LN, EPI_GELU>(h, g, ks)));
            }
        }
        {   // linear2 + residual -> pre2   (K = ff split over the 4 waves of the workgroup)
            GemmArgs g = z;
            g.M = M; g.MT = MT; g.NT = D / 16; g.KBtot = h->ff / KB; g.Wp = ly.W2; g.bias = ly.b2;
            g.A = h->hidden; g.lda = h->ff; g.a_frag = 1; g.out = h->pre2; g.ldo = D; g.R = h->X1;
            CHK(launch_gemm_k4<P>(h, g, ks));
        }
    }
    {   // final LayerNorm-on-read + pose head + sampler update
        GemmArgs g = z;
        g.M = M; g.MT = MT; g.NT = h->Jp / 16; g.KBtot = D / KB; g.Wp = h->Wp_out; g.bias = h->b_out;
        g.X = h->pre2; g.ln_g = h->layers[h->L - 1].g2; g.ln_b = h->layers[h->L - 1].be2; g.Xn = nullptr;
        g.out_mode = c.out_mode; g.xs32 = h->xs32; g.xsA = is_bf16(h) ? h->xsA : nullptr;
        g.fwd_out = h->fwd_out; g.ctl = c.use_ctr ? h->ctl : nullptr; g.st = step_tables(h); g.n_tab = h->n_run;
        g.dyn = h->dyn; g.ext_noise = c.ext_noise; g.const_noise = c.const_noise; g.clip_x0 = c.clip_x0; g.no_noise = c.no_noise;
        g.xs_frag = ks.xs_frag ? 1 : 0;
        h->fence_next = 2;     // the last packet of a step writes the state (state_fences)
        if (h->cfgB > 0) {      // guidance: one workgroup per CONDITIONAL row tile evaluates the twin rows as well (k_gemm_cfg)
            g.B = h->cfgB; g.M = h->cfgB * ntok; g.MT = cdiv(g.M, 16);
            g.cfgB = h->cfgB; g.cfg_off = h->cfgB * ntok; g.cfg_scale = h->cfg_scale;
            g.KS = 1; g.kb_per_split = g.KBtot; g.inv_ntok = fastdiv_inv(g.ntok); g.inv_hd = fastdiv_inv(g.hd);
            if (g.NT % 4) return fail(DSG_E_INVALID, "gemm: NT not divisible by the workgroup tile");
            const dim3 grid(xcd_grid_x(g.NT / 4), g.MT + 1, 1);
            // (bf16w2: 16 k-blocks of two-register weight + activation fragments do not fit -- two batches of 8, as in launch_gemm_w; round-5 advisor)
            bool done = false;
            if constexpr (!P::W2) {
                if (pick_ch(g.KBtot) == 16) { CHK((step_launch<&k_gemm_cfg<P, 16>>(h, grid, dim3(256), g))); done = true; }
            }
            if (!done && pick_ch(g.KBtot) == 12) { CHK((step_launch<&k_gemm_cfg<P, 12>>(h, grid, dim3(256), g))); done = true; }
            if (!done) CHK((step_launch<&k_gemm_cfg<P>>(h, grid, dim3(256), g)));
        } else if (ks.ffn || ks.ffn_split) {      // the rows are normalised already (k_ffn / k_ffn_ln of the last layer; cfgB == 0 here)
            g.X = nullptr; g.ln_g = nullptr; g.ln_b = nullptr; g.A = h->X0a; g.lda = D; g.a_frag = 1;
            CHK((launch_gemm_w<P, PRO_DIRECT, EPI_OUT>(h, g, ks)));
        } else {
            CHK((launch_gemm_w<P, PRO_LN, EPI_OUT>(h, g, ks)));
        }
    }
    h->last_kset = ks.set;
    return 0;
}
// ---- diagnostics: time a chain of ONE phase kernel (graph replay) to separate launch floor, kernel body and
//      weight coldness.  which: 0 null, 1 out_proj GEMM of layer 0 (same weights every launch), 2 out_proj cycling
//      over the layers, 3 LN+linear1+GELU cycling, 4 linear2 cycling, 5 k_attn, 6 k_loc, 7 k_in, 8 pose head (forward),
//      9 LN+QKV cycling, 10 k_mid cycling, 12 k_inloc
template <class P>
static int debug_launch(dsg_handle* h, int which, int i, int B) {
    const int D = h->D, T = h->T, ntok = h->ntok, M = B * ntok, MT = cdiv(M, 16), Min = B * T, MTin = cdiv(Min, 16);
    const int KB = P::KB;
    GemmArgs z;
    memset(&z, 0, sizeof(z));
    z.KS = 1; z.B = B; z.ntok = ntok; z.Tp = h->Tp; z.H = h->H; z.hd = h->hd; z.T = T; z.J = h->J; z.Jp = h->Jp;
    z.Jq = h->Jq; z.D = D; z.inv_ntok4 = fastdiv_inv(rup(ntok, 4));
    const Layer& ly = h->layers[(which == 1) ? 0 : i % h->L];
    LocArgs la;
    memset(&la, 0, sizeof(la));
    la.partial = h->partial; la.KS = h->KSin; la.Min_pad = MTin * 16; la.Cf = h->Cf; la.TE2 = h->TE2; la.TE = h->TE;
    la.emb1 = h->emb1; la.ctl = nullptr; la.t_arr = h->t_arr;
    la.rcos = h->rcos; la.rsin = h->rsin; la.mask = h->mask; la.mb = h->mb; la.inv_mask_div = fastdiv_inv((int)((long long)B * h->Hl / h->mb)); la.B = B; la.T = T; la.D = D; la.Hl = h->Hl;
    la.hd = h->hdl; la.W = h->W; la.X0 = h->X0; la.X0a = h->X0a; la.nomask = h->nomask;
    if (which == 20) return debug_launch<P>(h, (i & 1) ? 1 : 4, i, B);                 // alternate 2 kernels
    if (which == 21) { const int seq[4] = {1, 4, 7, 9}; return debug_launch<P>(h, seq[i & 3], i, B); }   // 4 kernels
    if (which == 22) { const int seq[6] = {12, 9, 5, 10, 4, 8}; return debug_launch<P>(h, seq[i % 6], i, B); } // the step's 6 kernels
    switch (which) {
        case 0: hipLaunchKernelGGL(k_ctr_inc, dim3(96), dim3(256), 0, h->stream, h->ctr); return 0;
        case 1: case 2: {
            GemmArgs g = z; g.M = M; g.MT = MT; g.NT = D / 16; g.KBtot = D / KB; g.Wp = ly.Wo; g.bias = ly.bo;
            g.A = h->attn; g.lda = D; g.a_frag = 1; g.out = (i & 1) ? h->pre1 : h->pre2; g.ldo = D; g.R = h->X0;
            return launch_gemm<P, PRO_DIRECT, EPI_RESID, 4, 1>(h, g); }
        case 3: {
            GemmArgs g = z; g.M = M; g.MT = MT; g.NT = h->ff / 16; g.KBtot = D / KB; g.Wp = ly.W1; g.bias = ly.b1;
            g.X = h->pre1; g.ln_g = ly.g1; g.ln_b = ly.be1; g.Xn = h->X1; g.out = h->hidden; g.ldo = h->ff; g.out_frag = 1;
            return launch_gemm<P, PRO_LN, EPI_GELU, 4, 1>(h, g); }
        case 4: {
            GemmArgs g = z; g.M = M; g.MT = MT; g.NT = D / 16; g.KBtot = h->ff / KB; g.Wp = ly.W2; g.bias = ly.b2;
            g.A = h->hidden; g.lda = h->ff; g.a_frag = 1; g.out = h->pre2; g.ldo = D; g.R = h->X1;
            return launch_gemm<P, PRO_DIRECT, EPI_RESID, 1, 4>(h, g); }
        case 5: {
            AttnArgs a; memset(&a, 0, sizeof(a)); a.q = h->q; a.k = h->k; a.vt = h->vt; a.out = h->attn; a.B = B; a.H = h->H; a.ntok = ntok;
            a.Tp = h->Tp; a.D = D; return launch_attn<P>(h, a); }
        case 6: DSG_LOC_DISPATCH(k_loc, la, dim3(h->Hl, T / h->W, B)); return 0;
        case 7: {
            GemmArgs g = z; g.M = Min; g.MT = MTin; g.NT = D / 16; g.KBtot = h->Jp / KB; g.KS = h->KSin; g.Wp = h->Wp_in;
            g.kb_per_split = cdiv(g.KBtot, g.KS); g.A = is_bf16(h) ? h->xsA : (void*)h->xs32; g.lda = h->Jp;
            g.out = h->partial; g.ldo = D; return launch_gemm<P, PRO_DIRECT, EPI_PARTIAL, 4, 1>(h, g); }
        case 8: {
            GemmArgs g = z; g.M = M; g.MT = MT; g.NT = h->Jp / 16; g.KBtot = D / KB; g.Wp = h->Wp_out; g.bias = h->b_out;
            g.X = h->pre2; g.ln_g = ly.g2; g.ln_b = ly.be2; g.out_mode = OUT_FORWARD; g.xs32 = h->xs32; g.fwd_out = h->fwd_out;
            g.ctl = nullptr; g.st = step_tables(h); g.n_tab = 1; g.dyn = h->dyn;
            return launch_gemm<P, PRO_LN, EPI_OUT, 4, 1>(h, g); }
        case 9: {
            GemmArgs g = z; g.M = M; g.MT = MT; g.NT = 3 * D / 16; g.KBtot = D / KB; g.Wp = ly.Wqkv; g.bias = ly.bqkv;
            g.q = h->q; g.k = h->k; g.vt = h->vt; g.X = h->pre2; g.ln_g = ly.g2; g.ln_b = ly.be2; g.Xn = h->Xn;
            return launch_gemm<P, PRO_LN, EPI_QKV, 4, 1>(h, g); }
        case 10: {
            MidArgs a; memset(&a, 0, sizeof(a)); a.A = h->attn; a.R = h->Xn; a.Wo = ly.Wo; a.bo = ly.bo; a.ln_g = ly.g1; a.ln_b = ly.be1;
            a.W1 = ly.W1; a.b1 = ly.b1; a.X1 = h->X1; a.hidden = h->hidden; a.M = M; a.MT = MT; a.ff = h->ff;
            return launch_mid<P>(h, a); }
        case 12: {
            InLocArgs a; a.xs = is_bf16(h) ? h->xsA : (void*)h->xs32; a.Jp = h->Jp; a.Wp = h->Wp_in;
            a.KBtot = h->Jp / KB; a.loc = la; a.ctl_upd = nullptr; a.st = step_tables(h); a.n_tab = 1;
            DSG_LOC_DISPATCH(k_inloc, a, dim3(h->Hl, T / h->W, B + 1)); return 0; }
        default: return fail(DSG_E_INVALID, "debug_chain: unknown kernel id");
    }
}
// copies an internal buffer to the host (raw bytes) -- used to bisect GPU-vs-emulator differences
extern "C" int dsg_debug_read(dsg_handle* h, const char* name, void* out, long long max_bytes, long long* n_bytes) {
    if (!h || !name || !out) return fail(DSG_E_INVALID, "null");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t B = h->Bmax, D = h->D, M = rup((int)(B * h->ntok), 16), Min = rup((int)(B * h->T), 16), es = h->es;
    std::map<std::string, std::pair<const void*, size_t>> m = {
        {"partial", {h->partial, (size_t)h->KSin * Min * D * 4}}, {"X0", {h->X0, M * D * 4}}, {"X0a", {h->X0a, M * D * es}},
        {"q", {h->q, B * h->H * h->Tp * h->hd * es}}, {"k", {h->k, B * h->H * h->Tp * h->hd * es}},
        {"vt", {h->vt, B * h->H * h->Tp * h->hd * es}}, {"attn", {h->attn, M * D * es}}, {"pre1", {h->pre1, M * D * 4}},
        {"X1", {h->X1, M * D * 4}}, {"Xn", {h->Xn, M * D * 4}}, {"hidden", {h->hidden, M * h->ff * es}},
        {"pre2", {h->pre2, M * D * 4}}, {"fwd_out", {h->fwd_out, B * h->J * h->T * 4}},
        {"xs32", {h->xs32, B * h->T * h->Jp * 4}}, {"xsA", {h->xsA, h->xsA ? Min * h->Jp * es : 0}},
        {"Cf", {h->Cf, B * h->T * D * 4}}, {"emb1", {h->emb1, B * D * 4}}};
    auto it = m.find(name);
    if (it == m.end() || !it->second.first) return fail(DSG_E_INVALID, "unknown buffer");
    size_t n = std::min<size_t>(it->second.second, (size_t)max_bytes);
    HIPCHK(hipMemcpy(out, it->second.first, n, hipMemcpyDeviceToHost));
    if (n_bytes) *n_bytes = (long long)n;
    return 0;
}

extern "C" int dsg_debug_chain(dsg_handle* h, int which, int n, int use_graph, int B, float* us_per_launch) {
    if (!h || !h->finalized || !h->cond_set) return fail(DSG_E_STATE, "debug_chain needs a finalized, conditioned handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->prec == DSG_PREC_BF16W2) return fail(DSG_E_NOT_IMPLEMENTED, "dsg_debug_chain: precision bf16w2");
#ifdef DSG_DEV_BF16_ONLY
    auto one = [&](int i) { return debug_launch<PBF16>(h, which, i, B); };
#else
    auto one = [&](int i) { return h->prec == DSG_PREC_BF16 ? debug_launch<PBF16>(h, which, i, B) : debug_launch<PF32>(h, which, i, B); };
#endif
    const int G = 64;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    if (use_graph) {
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < G; ++i) CHK(one(i));
        HIPCHK(hipStreamEndCapture(h->stream, &graph));
        HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIPCHK(hipEventRecord(h->ev_t0, h->stream));
        if (use_graph) for (int i = 0; i < n / G; ++i) HIPCHK(hipGraphLaunch(exec, h->stream));
        else for (int i = 0; i < n; ++i) CHK(one(i));
        HIPCHK(hipEventRecord(h->ev_t1, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        float ms; HIPCHK(hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1));
        best = std::min(best, 1000.f * ms / (use_graph ? (n / G) * G : n));
    }
    if (exec) { (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); }
    *us_per_launch = best;
    return 0;
}

// ---- diagnostics: per-packet timeline of the AQL step loop (dsg_aql.h: Trace).  dsg_debug_trace_arm before a dsg_sample traces
//      steps [first, first + n) of that call; dsg_debug_trace_get returns, per traced step and packet, start / end in us (relative
//      to the first traced dispatch) + the kernel names, ';'-separated.
extern "C" int dsg_debug_trace_arm(dsg_handle* h, int first, int n) {
    if (!h || first < 0 || n <= 0) return fail(DSG_E_INVALID, "dsg_debug_trace_arm: bad argument");
#ifndef DSG_EMU
    h->aql.trace.armed = true; h->aql.trace.first = first; h->aql.trace.n = n;
    return 0;
#else
    return fail(DSG_E_NOT_IMPLEMENTED, "no AQL path under emulation");
#endif
}
extern "C" int dsg_debug_trace_get(dsg_handle* h, double* us, int cap, int* n_steps, int* n_packets, char* names, int names_cap) {
    if (!h || !us || !n_steps || !n_packets) return fail(DSG_E_INVALID, "null argument");
#ifndef DSG_EMU
    const dsg_aql::Trace& t = h->aql.trace;
    const int L = (int)t.names.size();
    if (L == 0 || t.us.empty()) return fail(DSG_E_STATE, "no trace: arm it, then sample through the AQL path");
    *n_packets = L; *n_steps = (int)(t.us.size() / 2 / L);
    if ((int)t.us.size() > cap) return fail(DSG_E_INVALID, "trace buffer too small");
    memcpy(us, t.us.data(), t.us.size() * sizeof(double));
    if (names && names_cap > 0) {
        std::string all;
        for (const auto& n : t.names) { all += n; all += ';'; }
        snprintf(names, (size_t)names_cap, "%s", all.c_str());
    }
    return 0;
#else
    return fail(DSG_E_NOT_IMPLEMENTED, "no AQL path under emulation");
#endif
}

// marks build (-DDSG_STAMPS=2): per traced step, packet and mark k < 12: {waves that passed the mark, mean, last wave} in us after the packet's
// first wave start (dsg_kernels.h: DSG_TL_MARK)
extern "C" int dsg_debug_trace_marks(dsg_handle* h, double* out, int cap, int* n_marks) {
    if (!h || !out || !n_marks) return fail(DSG_E_INVALID, "null argument");
#if !defined(DSG_EMU) && defined(DSG_STAMPS) && DSG_STAMPS >= 2
    const dsg_aql::Trace& t = h->aql.trace;
    if (t.marks.empty()) return fail(DSG_E_STATE, "no marks: arm a trace, then sample through the AQL path");
    if ((int)t.marks.size() > cap) return fail(DSG_E_INVALID, "marks buffer too small");
    memcpy(out, t.marks.data(), t.marks.size() * sizeof(double));
    *n_marks = DSG_TL_NMARK;
    return 0;
#else
    (void)cap;
    return fail(DSG_E_NOT_IMPLEMENTED, "phase marks exist in the marks build only (make marks)");
#endif
}

static int run_step_p(dsg_handle* h, const StepCtx& c) {
#ifndef DSG_DEV_BF16_ONLY
    if (h->prec == DSG_PREC_BF16W2) return run_step<PBF16W2>(h, c);
    if (h->prec == DSG_PREC_FP32) return run_step<PF32>(h, c);
#endif
    return run_step<PBF16>(h, c);
}

static int launch_x_in(dsg_handle* h, const float* x, const float* init, int do_q, float qa, float qb, int use_philox,
                       NoiseKey nk, unsigned draw, int B, const KernelSel& ks) {
    XInArgs a;
    a.dupB = h->cfgB; a.xs_frag = ks.xs_frag ? 1 : 0;
    a.x = x; a.init = init; a.do_q = do_q; a.qa = qa; a.qb = qb; a.use_philox = use_philox; a.nkey = nk; a.draw = draw;
    a.B = B; a.J = h->J; a.Jp = h->Jp; a.Jq = h->Jq; a.T = h->T; a.xs32 = h->xs32;
    a.xsA = is_bf16(h) ? h->xsA : nullptr;
    const size_t n = (size_t)B * h->T * (h->Jp / 4);
    const int grid = (int)std::min<size_t>((n + 255) / 256, 2048);
    if (is_bf16(h)) hipLaunchKernelGGL((k_x_in<PBF16>), dim3(grid), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL((k_x_in<PF32>), dim3(grid), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    return 0;
}
static int launch_x_out(dsg_handle* h, float* dst_dev, int B) {
    const size_t n = (size_t)B * h->J * h->T;
    hipLaunchKernelGGL(k_x_out, dim3((int)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, h->stream, h->xs32,
                       dst_dev, B, h->J, h->Jp, h->T);
    HIPCHK(hipGetLastError());
    return 0;
}

// `user_stream` is the caller's hipStream_t; NULL is the legacy default stream (torch's default on ROCm), which does
// NOT implicitly synchronise with the handle's non-blocking stream -- so the ordering events are always recorded.
static int order_after(dsg_handle* h, void* user_stream) {
    HIPCHK(hipEventRecord(h->ev_in, (hipStream_t)user_stream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_in, 0));
    return 0;
}
static int order_before(dsg_handle* h, void* user_stream) {
    HIPCHK(hipEventRecord(h->ev_out, h->stream));
    HIPCHK(hipStreamWaitEvent((hipStream_t)user_stream, h->ev_out, 0));
    return 0;
}
// bring a caller tensor to the device (returns the device pointer to use)
static int to_dev(dsg_handle* h, const float* src, float* staging, size_t n, const float** out) {
    if (!src) { *out = nullptr; return 0; }
    if (is_device_ptr(src)) { *out = src; return 0; }
    HIPCHK(hipMemcpyAsync(staging, src, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    *out = staging;
    return 0;
}
static int from_dev(dsg_handle* h, float* dst, const float* src_dev, size_t n) {
    if (is_device_ptr(dst)) {
        HIPCHK(hipMemcpyAsync(dst, src_dev, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    } else {
        HIPCHK(hipMemcpyAsync(dst, src_dev, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

// rows the kernels run on for a user batch of B (guidance doubles it); checks B against the conditioning
static int rows_for(dsg_handle* h, int B, int* rows) {
    const int want = h->cfgB > 0 ? h->cfgB : h->condB;
    if (B != want) return fail(DSG_E_INVALID, "batch differs from the batch of dsg_set_window_cond");
    *rows = h->condB;
    return 0;
}

extern "C" int dsg_forward(dsg_handle* h, const float* x, const int64_t* t, float* out, int B, void* stream) {
    if (!h || !x || !t || !out) return fail(DSG_E_INVALID, "dsg_forward: null argument");
    if (!h->finalized || !h->cond_set) return fail(DSG_E_STATE, "dsg_forward before finalize / set_window_cond");
    int rows = 0;
    CHK(rows_for(h, B, &rows));
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(order_after(h, stream));
    std::vector<int> tt(rows);
    if (is_device_ptr(t)) {
        std::vector<int64_t> th(B);
        HIPCHK(hipMemcpy(th.data(), t, B * sizeof(int64_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) tt[i] = (int)th[i];
    } else {
        for (int i = 0; i < B; ++i) tt[i] = (int)t[i];
    }
    for (int i = B; i < rows; ++i) tt[i] = tt[i - B];      // guidance: the twins share the timestep
    for (int i = 0; i < B; ++i)
        if (tt[i] < 0 || tt[i] >= h->n_te) return fail(DSG_E_INVALID, "timestep out of range");
    HIPCHK(hipMemcpyAsync(h->t_arr, tt.data(), rows * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));     // tt is a stack-lifetime staging buffer
    const size_t n = (size_t)B * h->J * h->T;
    const float* xd = nullptr;
    CHK(to_dev(h, x, h->io_tmp, n, &xd));
    NoiseKey nk = {0, 0, 0, 0};
    StepCtx c; c.B = rows; c.out_mode = OUT_FORWARD; c.use_ctr = false; c.ext_noise = nullptr; c.const_noise = 0;
    CHK(select_kernels(h, rows, c.ks));
    CHK(ensure_set_buffers(h, c.ks));
    CHK(launch_x_in(h, xd, nullptr, 0, 0.f, 0.f, 0, nk, 0, B, c.ks));
    CHK(run_step_p(h, c));
    CHK(from_dev(h, out, h->fwd_out, n));
    CHK(order_before(h, stream));
    return 0;
}

// per-step coefficient tables in execution order (gaussian_diffusion.py:1617 `.float()` of the float64 tables)
static int build_step_tables(dsg_handle* h, int mode, int skip, float eta, int* n_run_out) {
    const Sched& s = h->sched;
    if (s.n == 0) return fail(DSG_E_STATE, "dsg_sample before dsg_set_schedule");
    if (skip < 0 || skip >= s.n) return fail(DSG_E_INVALID, "skip_timesteps out of range");
    const int n_run = s.n - skip;
    // every window of a clip asks for the same tables: they stay on the device until schedule / mode / skip / eta change
    if (h->st_valid && h->st_mode == mode && h->st_skip == skip && h->st_eta == eta) {
        *n_run_out = n_run; h->n_run = n_run;
        return 0;
    }
    std::vector<int> tm(n_run);
    std::vector<float> c[5];
    for (auto& v : c) v.assign(n_run, 0.f);
    for (int i = 0; i < n_run; ++i) {
        const int idx = n_run - 1 - i;
        tm[i] = s.tmap[idx];
        const float nz = idx == 0 ? 0.f : 1.f;
        if (mode == DSG_MODE_DDPM) {
            c[0][i] = (float)s.coef1[idx];
            c[1][i] = (float)s.coef2[idx];
            c[2][i] = nz * expf(0.5f * (float)s.plogvar[idx]);
        } else {
            const float ab = (float)s.ac[idx], abp = (float)s.acp[idx];
            const float sigma = eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
            c[0][i] = (float)s.sqrt_recip[idx];
            c[1][i] = (float)s.sqrt_recipm1[idx];
            c[2][i] = sqrtf(abp);
            c[3][i] = sqrtf(1.f - abp - sigma * sigma);
            c[4][i] = nz * sigma;
        }
    }
    HIPCHK(hipMemcpyAsync(h->st_tmodel, tm.data(), n_run * sizeof(int), hipMemcpyHostToDevice, h->stream));
    for (int k = 0; k < 5; ++k)
        HIPCHK(hipMemcpyAsync(h->st_c[k], c[k].data(), n_run * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n_run_out = n_run;
    h->n_run = n_run;
    h->st_valid = true; h->st_mode = mode; h->st_skip = skip; h->st_eta = eta;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// dsg_sample = prepare (x_T, step tables, control block, [AQL packet plan]) -> step loop -> finish (x_0 out).
// dsg_sample_multi runs the step loops of several handles ("lanes": own HSA queue each, shared weights) concurrently from one
// host thread.
// ---------------------------------------------------------------------------------------------------------
struct SampleJob {
    StepCtx c;
    int n_run = 0, B = 0, done = 0;      // n_run: steps THIS call runs; done: how many of them have been issued
    int first = 0;                       // absolute loop index of its first step (dsg_sample_args.first_step: a resumed chain)
    bool dumping = false, aql = false;
    int spg = -1;
};

#ifndef DSG_EMU
// One-time check of the premise of the fence-free loop on the device it is about to run on (advisor, round 2): a hand-off
// through uncached memory between two dependent AQL packets WITHOUT acquire / release must never be stale, whichever XCDs
// the writer and the reader run on.  64 x {k_uc_probe_w, k_uc_probe_r} on the handle's own queue, the same protocol as the
// step loop (device-resident iteration word, read with vector loads, advanced by an extra workgroup of the other kernel).
// A mismatch (another ASIC / ROCm version mapping hipDeviceMallocUncached differently) turns fence-free submission off for
// the process, with a warning; dsg_last_sample_fence_free then reports 0.
static std::atomic<int> g_uc_checked[64];      // per device: 0 not yet, 1 ok, 2 broken
static std::mutex g_uc_mutex;                  // one probe at a time: two host threads creating handles on one device run it once
static bool uc_selfcheck(dsg_handle* h) {
    const int dev = h->cfg.device & 63;
    if (int v = g_uc_checked[dev].load(std::memory_order_acquire)) return v == 1;
    std::lock_guard<std::mutex> lock(g_uc_mutex);
    if (int v = g_uc_checked[dev].load(std::memory_order_acquire)) return v == 1;
    const int n_wg = 256, iters = 64;
    // The 257 KB probe buffer comes from outside the pool's cap (round-5 advisor: a pool that is full at this moment is a TRANSIENT condition,
    // but the verdict stored here is permanent for the process); it goes back to the pool like every uncached block.  No uncached memory at
    // all on this device: that IS permanent.
    unsigned* buf = (unsigned*)uc_pool_take(h->cfg.device, (size_t)(n_wg * 256 + 64) * sizeof(unsigned), /*ignore_cap=*/true);
    if (!buf) {
        g_uc_checked[dev].store(2, std::memory_order_release);
        return false;
    }
    bool ok = hipMemset(buf, 0, (size_t)(n_wg * 256 + 64) * sizeof(unsigned)) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    UcProbeArgs a;
    a.buf = buf; a.ctl = (int*)(buf + n_wg * 256); a.err = buf + n_wg * 256 + 16; a.n_wg = n_wg;
    const bool trace_was_armed = h->aql.trace.armed;      // (a timeline trace armed for the caller's sample is not for this run)
    h->aql.trace.armed = false;
    if (ok) {
        dsg_aql::begin(h->aql);
        ok = dsg_aql::record(h->aql, (const void*)&k_uc_probe_w, h->stream, dim3(n_wg + 1), dim3(256), &a, sizeof a) &&
             dsg_aql::record(h->aql, (const void*)&k_uc_probe_r, h->stream, dim3(n_wg + 1), dim3(256), &a, sizeof a);
        ok = dsg_aql::finish(h->aql) && ok;
        h->aql.nofence = true;
        ok = ok && dsg_aql::run(h->aql, iters, 10.0);
        h->aql.nofence = false;
    }
    h->aql.trace.armed = trace_was_armed;
    unsigned res[2] = {1u, 0u};
    if (ok) ok = hipMemcpy(res, a.err, sizeof res, hipMemcpyDeviceToHost) == hipSuccess;
    uc_pool_give(h->cfg.device, buf);      // (uncached memory goes back to the pool, never to hipFree: see UcPool)
    const bool good = ok && res[0] == 0u && res[1] == (unsigned)iters;      // no stale word seen, and the reader really ran `iters` times
    if (!good)
        fprintf(stderr, "libdsg_hip: WARNING: uncached-memory hand-off check failed on device %d (stale words: %u, iterations seen: %u of %d); "
                        "the step loop keeps its acquire / release fences\n", h->cfg.device, res[0], res[1], iters);
    g_uc_checked[dev].store(good ? 1 : 2, std::memory_order_release);
    return good;
}
#endif

#ifdef DSG_X_HOSTPROF
#include <chrono>
struct HostProf {
    double acc[16] = {0}; long n = 0; std::chrono::steady_clock::time_point t;
    void start() { t = std::chrono::steady_clock::now(); ++n; }
    void lap(int i) { auto u = std::chrono::steady_clock::now(); acc[i] += std::chrono::duration<double, std::micro>(u - t).count(); t = u; }
    ~HostProf() { if (n) { fprintf(stderr, "hostprof (us per call, %ld calls):", n); for (int i = 0; i < 16; ++i) if (acc[i] > 0) fprintf(stderr, " [%d] %.1f", i, acc[i] / n); fprintf(stderr, "\n"); } }
};
static HostProf g_hp;
#define HP_START() g_hp.start()
#define HP_LAP(i) g_hp.lap(i)
#else
#define HP_START() ((void)0)
#define HP_LAP(i) ((void)0)
#endif
static int sample_prepare(dsg_handle* h, const dsg_sample_args* a, int B, void* stream, SampleJob& job) {
    if (!h || !a) return fail(DSG_E_INVALID, "dsg_sample: null argument");
    if (!h->finalized || !h->cond_set) return fail(DSG_E_STATE, "dsg_sample before finalize / set_window_cond");
    int rows = 0;
    CHK(rows_for(h, B, &rows));
    if (a->mode != DSG_MODE_DDPM && a->mode != DSG_MODE_DDIM) return fail(DSG_E_INVALID, "mode");
    // (dump points are allowed with DDIM: ddim_sample_loop_progressive is built on them; the Python ddim_sample_loop itself
    // refuses dump_steps like the reference, gaussian_diffusion.py:913-916)
    if (a->mode == DSG_MODE_DDIM && a->const_noise)
        return fail(DSG_E_NOT_IMPLEMENTED, "ddim_sample_loop: const_noise (gaussian_diffusion.py:915-916)");
    HIPCHK(hipSetDevice(h->cfg.device));
    HP_START();
    CHK(order_after(h, stream));
    int n_run = 0;
    CHK(build_step_tables(h, a->mode, a->skip_timesteps, a->eta, &n_run));
    HP_LAP(0);
    // a chain may be run in pieces (first_step / max_steps: the lazy generator forms of the loops): steps [first, first + n_iter)
    const int first = a->first_step, n_total = n_run;
    if (first < 0 || first >= n_total || a->max_steps < 0) return fail(DSG_E_INVALID, "first_step / max_steps out of range");
    if (first > 0 && !a->init_noise) return fail(DSG_E_INVALID, "first_step > 0 resumes a chain: init_noise must hold x_t of that step");
    const int n_iter = a->max_steps > 0 ? std::min(a->max_steps, n_total - first) : n_total - first;
    const size_t n = (size_t)B * h->J * h->T;
    NoiseKey nk;
    nk.k0 = (unsigned)(a->seed & 0xffffffffu); nk.k1 = (unsigned)(a->seed >> 32);
    nk.s0 = (unsigned)(a->stream_id & 0xffffffffu); nk.s1 = (unsigned)(a->stream_id >> 32);

    // x_T (gaussian_diffusion.py:701-713)
    const float *noise_d = nullptr, *init_d = nullptr;
    CHK(to_dev(h, a->init_noise, h->io_tmp, n, &noise_d));
    CHK(to_dev(h, a->init_image, h->io_tmp2, n, &init_d));
    const int do_q = (first == 0 && (a->skip_timesteps > 0 || a->init_image)) ? 1 : 0;      // (a resumed chain starts from x_t as handed over)
    const int i0 = n_run - 1;
    KernelSel ksel;
    CHK(select_kernels(h, rows, ksel));      // (first: the layout of the state shadow belongs to the kernel set)
    CHK(ensure_set_buffers(h, ksel));
    CHK(launch_x_in(h, noise_d, init_d, do_q, (float)h->sched.sqrt_ac[i0], (float)h->sched.sqrt_1mac[i0],
                    noise_d ? 0 : 1, nk, a->draw_base, B, ksel));
    // replayed per-step noise
    const float* ext = nullptr;
    if (a->step_noise) {
        const size_t ne = (size_t)n_total * n;
        if (is_device_ptr(a->step_noise)) ext = a->step_noise;
        else {
            if (ne > h->ext_noise_cap) { CHK(dalloc(h, &h->ext_noise, ne, false)); h->ext_noise_cap = ne; }
            HIPCHK(hipMemcpyAsync(h->ext_noise, a->step_noise, ne * sizeof(float), hipMemcpyHostToDevice, h->stream));
            ext = h->ext_noise;
        }
    }
    hipLaunchKernelGGL(k_ctl_init, dim3(1), dim3(64), 0, h->stream, h->ctl, h->st_tmodel, first);
    HIPCHK(hipGetLastError());
    HP_LAP(1);
    {
        const unsigned dyn[5] = {nk.k0, nk.k1, nk.s0, nk.s1, a->draw_base + 1u};
        HIPCHK(hipMemcpyAsync(h->dyn, dyn, sizeof(dyn), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    HP_LAP(2);
    StepCtx& c = job.c;
    c.B = rows; c.out_mode = a->mode == DSG_MODE_DDPM ? OUT_DDPM : OUT_DDIM; c.use_ctr = true; c.ext_noise = ext;
    c.const_noise = a->const_noise; c.clip_x0 = a->clip_denoised ? 1 : 0;
    c.no_noise = (a->mode == DSG_MODE_DDIM && a->eta == 0.f && !ext) ? 1 : 0;
    c.ks = ksel;
    n_run = n_iter;                          // from here on: the steps of this call (the tables keep the whole chain: h->n_run)
    job.n_run = n_iter; job.B = B; job.done = 0; job.first = first;
    job.dumping = a->n_dump > 0 && a->dump_steps && a->dump_out;
    // steps_per_graph: 0 = default = no hipGraph.  Measured on MI355X / ROCm 7.2: hipGraph replay of the step is slower than
    // stream-ordered HIP launches (180 vs 157 us; 151-163 us with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0), and both lose to the
    // hand-written AQL submission (139 us) -- graphs stay opt-in (steps_per_graph > 0).
    job.spg = h->cfg.steps_per_graph == 0 ? -1 : h->cfg.steps_per_graph;
    if (job.dumping || ext) job.spg = -1;            // rare paths run eagerly (ext pointer / dump points are per call)
    h->aql_timing = false;
    h->last_path = 0; h->last_nofence = false;
    job.aql = false;
#ifndef DSG_EMU
    // The step loop as hand-written AQL packets (dsg_aql.h): one recording pass of run_step (no launch), argument
    // blocks to device memory, then n_run x the same packets on the handle's own HSA queue.  Any failure before the
    // first packet falls back to the HIP launches; a failure after submission is an error.
    const bool graph_wanted = job.spg > 0 && n_run >= job.spg;
    if (h->aql_mode == 1 && !job.dumping && !graph_wanted && n_run > 0) {
        bool planned = dsg_aql::init(h->aql, h->cfg.device, (const void*)&dsg_version);
        bool nofence = false;
        if (planned && h->uc_mode == 1) nofence = uc_selfcheck(h);
        HP_LAP(3);
        if (planned) {
            dsg_aql::begin(h->aql);
            const int rc = run_step_p(h, c);
            HP_LAP(4);
            planned = dsg_aql::finish(h->aql) && rc == 0;
            HP_LAP(5);
            h->aql.recording = false;
            h->aql.nofence = nofence;            // fence-free packets (see uc_mode; 2: uncached buffers behind the usual fences)
        }
        if (!planned) {
            if (!h->aql_warned) { fprintf(stderr, "libdsg_hip: AQL path unavailable (%s); using HIP launches\n", h->aql.err.c_str()); h->aql_warned = true; }
            h->aql_mode = 0;
        } else {
            HIPCHK(hipStreamSynchronize(h->stream));         // state / control block / conditioning are in place
            job.aql = true;
        }
    }
#endif
    HIPCHK(hipEventRecord(h->ev_t0, h->stream));
    HP_LAP(6);
    return 0;
}

// the step loop through the HIP runtime: hipGraph replays (opt-in) and / or stream-ordered launches
static int sample_run_hip(dsg_handle* h, const dsg_sample_args* a, SampleJob& job) {
    const int n_run = job.n_run, B = job.B, spg = job.spg;
    const StepCtx& c = job.c;
    const size_t n = (size_t)B * h->J * h->T;
    if (spg > 0 && n_run >= spg && job.done == 0) {
        // everything that varies between calls (step index, coefficients, noise key, conditioning) lives in device
        // memory, so one captured graph per (batch, sampler, mask batch, const_noise, steps, flags, kernel set) serves every window and clip
        dsg_handle::GKey key = {c.B, c.out_mode, h->mb, c.const_noise, n_run, h->n_run, (c.clip_x0 ? 1 : 0) | (h->cfgB ? 2 : 0) | (h->nomask ? 4 : 0) | (c.ks.attn_in_mid ? 8 : 0) | (c.no_noise ? 16 : 0),
                                c.ks.set};
        auto it = h->graphs.find(key);
        bool ok = true;
        if (it == h->graphs.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            hipError_t e = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rc = 0;
                for (int s = 0; s < spg && rc == 0; ++s) rc = run_step_p(h, c);
                e = hipStreamEndCapture(h->stream, &graph);
                if (rc != 0 || e != hipSuccess || !graph) ok = false;
                if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
            } else ok = false;
            if (!ok) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); }
            else {
                dsg_handle::GVal v = {exec, graph, spg};
                it = h->graphs.emplace(key, v).first;
            }
        }
        if (ok) {
            while (n_run - job.done >= spg) { HIPCHK(hipGraphLaunch(it->second.exec, h->stream)); job.done += spg; }
            h->last_path = 2;
            h->last_kset = c.ks.set;
        }
    }
    int di = 0;
    for (; job.done < n_run; ++job.done) {
        CHK(run_step_p(h, c));
        if (job.dumping) {
            while (di < a->n_dump && a->dump_steps[di] < job.first + job.done) ++di;
            if (di < a->n_dump && a->dump_steps[di] == job.first + job.done) {
                CHK(launch_x_out(h, h->fwd_out, B));
                CHK(from_dev(h, a->dump_out + (size_t)di * n, h->fwd_out, n));
                ++di;
            }
        }
    }
    return 0;
}

static int sample_finish(dsg_handle* h, float* out, void* stream, SampleJob& job) {
    const size_t n = (size_t)job.B * h->J * h->T;
    HP_LAP(7);                               // (the step loop itself)
    HIPCHK(hipEventRecord(h->ev_t1, h->stream));
    h->last_steps = job.n_run; h->timing_valid = true;
    h->last_kset = job.c.ks.set;
    CHK(launch_x_out(h, h->fwd_out, job.B));
    CHK(from_dev(h, out, h->fwd_out, n));
    CHK(order_before(h, stream));
    HP_LAP(8);
    return 0;
}

extern "C" int dsg_sample(dsg_handle* h, const dsg_sample_args* a, float* out, int B, void* stream) {
    if (!h || !a || !out) return fail(DSG_E_INVALID, "dsg_sample: null argument");
    SampleJob job;
    CHK(sample_prepare(h, a, B, stream, job));
#ifndef DSG_EMU
    if (job.aql) {
        if (!dsg_aql::run(h->aql, job.n_run, 60.0 + 0.01 * job.n_run)) return fail(DSG_E_RUNTIME, "AQL run: " + h->aql.err);
        job.done = job.n_run;
        h->aql_timing = true; h->aql_ms = h->aql.last_ms;
        h->last_path = 1; h->last_nofence = h->aql.nofence;
    }
#endif
    CHK(sample_run_hip(h, a, job));
    return sample_finish(h, out, stream, job);
}

// n lanes (handles of ONE device, normally a handle and its clones), one independent sampling call each, advanced
// concurrently.  With the AQL submission every lane has its own HSA queue and the host thread deals the steps round-robin (the
// queues' dependent packet chains overlap on the GPU); with HIP launches the lanes' streams are fed step by step.  Every lane
// runs the kernel set ITS handle selects (dsg_set_kernel_set / the batch): the call changes nothing about the arithmetic, so
// lane i's sample is bit-identical to dsg_sample(lanes[i], &args[i], ...) on its own.
extern "C" int dsg_sample_multi(dsg_handle** hs, int n, const dsg_sample_args* args, float** outs, int B, void* stream) {
    if (!hs || !args || !outs || n <= 0) return fail(DSG_E_INVALID, "dsg_sample_multi: bad argument");
    for (int i = 0; i < n; ++i) {
        if (!hs[i] || !outs[i]) return fail(DSG_E_INVALID, "dsg_sample_multi: null handle / output");
        for (int j = 0; j < i; ++j) if (hs[j] == hs[i]) return fail(DSG_E_INVALID, "dsg_sample_multi: a handle appears twice");
        if (hs[i]->cfg.device != hs[0]->cfg.device) return fail(DSG_E_INVALID, "dsg_sample_multi: lanes must live on one device");
    }
    if (n > 16) return fail(DSG_E_INVALID, "dsg_sample_multi: at most 16 lanes (4 overlap on the hardware; put further clips into the lanes' batches)");
    std::vector<SampleJob> jobs(n);
    struct LanesNow {          // how many lanes share the GPU during this call (select_kernels: block shape of k_ffn; never the arithmetic)
        dsg_handle** hs; int n;
        LanesNow(dsg_handle** hs_, int n_) : hs(hs_), n(n_) { for (int i = 0; i < n; ++i) hs[i]->lanes_now = n; }
        ~LanesNow() { for (int i = 0; i < n; ++i) hs[i]->lanes_now = 1; }
    } lanes_now(hs, n);
    for (int i = 0; i < n; ++i) CHK(sample_prepare(hs[i], &args[i], B, stream, jobs[i]));
    bool all_aql = true;
    for (int i = 0; i < n; ++i) all_aql = all_aql && jobs[i].aql;
#ifndef DSG_EMU
    if (all_aql) {
        std::vector<dsg_aql::Ctx*> ctxs(n);
        std::vector<int> steps(n);
        double tmax = 60.0;
        for (int i = 0; i < n; ++i) { ctxs[i] = &hs[i]->aql; steps[i] = jobs[i].n_run; tmax += 0.01 * jobs[i].n_run; }
        std::string err;
        if (!dsg_aql::run_multi(ctxs.data(), steps.data(), n, tmax, err)) return fail(DSG_E_RUNTIME, "AQL run: " + err);
        for (int i = 0; i < n; ++i) {
            jobs[i].done = jobs[i].n_run;
            hs[i]->aql_timing = true; hs[i]->aql_ms = hs[i]->aql.last_ms; hs[i]->last_path = 1; hs[i]->last_nofence = hs[i]->aql.nofence;
        }
    }
#endif
    if (!all_aql) {
        // HIP launches: an AQL plan that was recorded for some lanes is simply not used; step s of every lane, then s + 1
        bool plain = true;
        for (int i = 0; i < n; ++i) plain = plain && !jobs[i].dumping && !(jobs[i].spg > 0 && jobs[i].n_run >= jobs[i].spg);
        if (plain) {
            int more = 1;
            for (int s = 0; more; ++s) {
                more = 0;
                for (int i = 0; i < n; ++i)
                    if (s < jobs[i].n_run) { CHK(run_step_p(hs[i], jobs[i].c)); jobs[i].done = s + 1; more = 1; }
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        CHK(sample_run_hip(hs[i], &args[i], jobs[i]));      // whatever is left (graphs, dump points); nothing after AQL
        CHK(sample_finish(hs[i], outs[i], stream, jobs[i]));
    }
    return 0;
}

extern "C" int dsg_sync(dsg_handle* h) {
    if (!h) return fail(DSG_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int dsg_last_sample_ms(dsg_handle* h, float* ms, int* n_steps) {
    if (!h || !ms) return fail(DSG_E_INVALID, "null argument");
    if (!h->timing_valid) return fail(DSG_E_STATE, "no dsg_sample has run");
    HIPCHK(hipEventSynchronize(h->ev_t1));
    if (h->aql_timing) *ms = (float)h->aql_ms;      // AQL path: host clock from the first doorbell to the completion signal
    else HIPCHK(hipEventElapsedTime(ms, h->ev_t0, h->ev_t1));
    if (n_steps) *n_steps = h->last_steps;
    return 0;
}
// how the step loop of the last dsg_sample was submitted: 0 = HIP launches, 1 = hand-written AQL packets, 2 = hipGraph replay
extern "C" int dsg_last_sample_path(dsg_handle* h, int* path) {
    if (!h || !path) return fail(DSG_E_INVALID, "null argument");
    if (!h->timing_valid) return fail(DSG_E_STATE, "no dsg_sample has run");
    *path = h->last_path;
    return 0;
}

// 1 when the packets of the last dsg_sample's loop carried no acquire / release fences (AQL paths only)
extern "C" int dsg_last_sample_fence_free(dsg_handle* h, int* fence_free) {
    if (!h || !fence_free) return fail(DSG_E_INVALID, "null argument");
    if (!h->timing_valid) return fail(DSG_E_STATE, "no dsg_sample has run");
    *fence_free = h->last_nofence ? 1 : 0;
    return 0;
}

// the framework's noise stream as a tensor (what the fused sampler consumes for draw index `draw`): out [B, J, 1, T], device
// memory (written on `stream`) or host memory (generated on the device, copied back, synchronous)
extern "C" int dsg_noise(float* out, int B, int J, int T, uint64_t seed, uint64_t stream_id, uint32_t draw, void* stream) {
    if (!out || B <= 0 || J <= 0 || T <= 0) return fail(DSG_E_INVALID, "dsg_noise: bad argument");
    NoiseKey nk;
    nk.k0 = (unsigned)(seed & 0xffffffffu); nk.k1 = (unsigned)(seed >> 32);
    nk.s0 = (unsigned)(stream_id & 0xffffffffu); nk.s1 = (unsigned)(stream_id >> 32);
    const int Jq = rup(J, 4);
    const size_t n = (size_t)B * T * (Jq / 4), bytes = (size_t)B * J * T * sizeof(float);
    const bool dev = is_device_ptr(out);
    float* dst = out;
    if (!dev) HIPCHK(hipMalloc((void**)&dst, bytes));
    hipLaunchKernelGGL(k_noise, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, dst, B, J, Jq, T, nk, draw);
    hipError_t e = hipGetLastError();
    if (!dev) {          // host output: the temporary is released on every path
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e == hipSuccess) e = hipMemcpy(out, dst, bytes, hipMemcpyDeviceToHost);
        (void)hipFree(dst);
    }
    if (e != hipSuccess) return fail(DSG_E_RUNTIME, std::string("dsg_noise: ") + hipGetErrorString(e));
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// stand-alone fused sampler arithmetic on caller tensors
// ---------------------------------------------------------------------------------------------------------
static int elementwise(float* out, const float* p, const float* q, const float* z, const float* a, const float* c,
                       const float* s, int B, int64_t per, void* stream) {
    if (!out || !p || !a || B <= 0 || per <= 0) return fail(DSG_E_INVALID, "elementwise: bad argument");
    if (!is_device_ptr(out) || !is_device_ptr(p) || (q && !is_device_ptr(q)) || (z && !is_device_ptr(z)))
        return fail(DSG_E_INVALID, "elementwise kernels take device tensors");
    float* coef = nullptr;
    HIPCHK(hipMalloc((void**)&coef, 3 * B * sizeof(float)));
    std::vector<float> hc(3 * B, 0.f);
    for (int b = 0; b < B; ++b) { hc[b] = a[b]; hc[B + b] = c ? c[b] : 0.f; hc[2 * B + b] = s ? s[b] : 0.f; }
    HIPCHK(hipMemcpy(coef, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t n = (size_t)B * per;
    hipLaunchKernelGGL(k_axpbypcz, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       out, p, q, z, coef, coef + B, coef + 2 * B, B, (size_t)per);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipFree(coef));
    return 0;
}
extern "C" int dsg_q_sample(float* out, const float* x_start, const float* noise, const float* sqrt_ac,
                            const float* sqrt_1mac, int B, int64_t per, void* stream) {
    return elementwise(out, x_start, noise, nullptr, sqrt_ac, sqrt_1mac, nullptr, B, per, stream);
}
extern "C" int dsg_predict_xstart_from_eps(float* out, const float* x_t, const float* eps, const float* sqrt_recip,
                                           const float* sqrt_recipm1, int B, int64_t per, void* stream) {
    std::vector<float> neg(B);
    for (int b = 0; b < B; ++b) neg[b] = -sqrt_recipm1[b];
    return elementwise(out, x_t, eps, nullptr, sqrt_recip, neg.data(), nullptr, B, per, stream);
}
extern "C" int dsg_posterior_step(float* out, const float* x_start, const float* x_t, const float* noise,
                                  const float* coef1, const float* coef2, const float* sigma_nz, int B, int64_t per,
                                  void* stream) {
    return elementwise(out, x_start, x_t, noise, coef1, coef2, sigma_nz, B, per, stream);
}
extern "C" int dsg_ddim_step(float* out, const float* x_start, const float* x_t, const float* noise, const float* coef,
                             int B, int64_t per, void* stream) {
    if (!out || !x_start || !x_t || !coef || B <= 0 || per <= 0) return fail(DSG_E_INVALID, "ddim_step: bad argument");
    if (!is_device_ptr(out) || !is_device_ptr(x_start) || !is_device_ptr(x_t) || (noise && !is_device_ptr(noise)))
        return fail(DSG_E_INVALID, "elementwise kernels take device tensors");
    float* dc = nullptr;
    HIPCHK(hipMalloc((void**)&dc, 5 * B * sizeof(float)));
    HIPCHK(hipMemcpy(dc, coef, 5 * B * sizeof(float), hipMemcpyHostToDevice));
    const size_t n = (size_t)B * per;
    hipLaunchKernelGGL(k_ddim_step, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       out, x_start, x_t, noise, dc, B, (size_t)per);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(hipFree(dc));
    return 0;
}
