// TEST INFRASTRUCTURE ONLY -- SIMT emulation runtime for tests/emu/shim/hip/hip_runtime.h.
// Each workgroup runs on one OS worker thread as `blockDim` cooperative fibers (ucontext); __syncthreads, wave64
// shuffles and MFMA are rendezvous points.  Workgroups of a launch are spread over a small pool of OS threads.
#include <hip/hip_runtime.h>

#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {

// ---- "device" allocations --------------------------------------------------------------------------------
static std::mutex g_mu;
static std::map<uintptr_t, size_t> g_allocs;
void* dmalloc(size_t n) {
    void* p = nullptr;
    if (posix_memalign(&p, 256, n ? n : 16)) return nullptr;
    std::lock_guard<std::mutex> l(g_mu);
    g_allocs[(uintptr_t)p] = n;
    return p;
}
void dfree(void* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> l(g_mu); g_allocs.erase((uintptr_t)p); }
    free(p);
}
bool is_device(const void* p) {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return false;
    --it;
    return (uintptr_t)p < it->first + it->second;
}

// ---- fibers ----------------------------------------------------------------------------------------------
constexpr int MAXT = 1024;
constexpr size_t STACK = 96 * 1024;

struct Wave {
    int arrived = 0; unsigned gen = 0; int n = 0;
    unsigned xu[2][64];
    unsigned short a16[2][64][8], b16[2][64][8];
    float af[2][64], bfv[2][64];
};
struct Fiber { ucontext_t uc; Ctx ctx; bool done = false; char* stack = nullptr; int wave = 0, lane = 0; };
struct Worker {
    ucontext_t main_uc;
    std::vector<Fiber> fibers;
    Wave waves[MAXT / 64];
    int n = 0, arrived = 0; unsigned gen = 0;
    int cur = -1;
    const std::function<void()>* body = nullptr;
};
static thread_local Worker* tw = nullptr;

Ctx* cur() { return &tw->fibers[tw->cur].ctx; }

static void yield_to_main() {
    Worker* w = tw;
    swapcontext(&w->fibers[w->cur].uc, &w->main_uc);
}
static void fiber_entry() {
    Worker* w = tw;
    (*w->body)();
    w->fibers[w->cur].done = true;
    yield_to_main();
}
void syncthreads() {
    Worker* w = tw;
    const unsigned g = w->gen;
    if (++w->arrived == w->n) { w->arrived = 0; w->gen++; return; }
    while (w->gen == g) yield_to_main();
}
static void wave_sync(Wave& wv) {
    const unsigned g = wv.gen;
    if (++wv.arrived == wv.n) { wv.arrived = 0; wv.gen++; return; }
    while (wv.gen == g) yield_to_main();
}
// double-buffered exchange slots: the buffer index is the parity of the wave's rendezvous generation, read BEFORE the
// rendezvous, so a lane that is already in collective k+1 never overwrites what a slower lane still reads for k
static inline int wave_parity(Wave& wv) { return (int)(wv.gen & 1u); }
unsigned shfl_xor_u32(unsigned v, int mask) {
    Worker* w = tw; Fiber& f = w->fibers[w->cur]; Wave& wv = w->waves[f.wave];
    const int p = wave_parity(wv);
    wv.xu[p][f.lane] = v;
    wave_sync(wv);
    return wv.xu[p][(f.lane ^ mask) & 63];
}

f32x4_t mfma_16x16x32_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    Worker* w = tw; Fiber& f = w->fibers[w->cur]; Wave& wv = w->waves[f.wave];
    const int p = wave_parity(wv);
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const u16x8 au = __builtin_bit_cast(u16x8, a), bu = __builtin_bit_cast(u16x8, b);
    for (int e = 0; e < 8; ++e) { wv.a16[p][f.lane][e] = au[e]; wv.b16[p][f.lane][e] = bu[e]; }
    wave_sync(wv);
    const int j = f.lane & 15, g4 = f.lane >> 4;
    f32x4_t d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 8; ++e) {
                const float x = __builtin_bit_cast(float, (unsigned)wv.a16[p][i + 16 * g][e] << 16);
                const float y = __builtin_bit_cast(float, (unsigned)wv.b16[p][j + 16 * g][e] << 16);
                acc += x * y;
            }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l holds row / column (l & 31) and the 8 k-values of group (l >> 5) of A / B;
// D: column j = l & 31, row i = (r & 3) + 8 (r >> 2) + 4 (l >> 5) for register r (cdna_hip_programming.md s3)
f32x16_t mfma_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    Worker* w = tw; Fiber& f = w->fibers[w->cur]; Wave& wv = w->waves[f.wave];
    const int p = wave_parity(wv);
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const u16x8 au = __builtin_bit_cast(u16x8, a), bu = __builtin_bit_cast(u16x8, b);
    for (int e = 0; e < 8; ++e) { wv.a16[p][f.lane][e] = au[e]; wv.b16[p][f.lane][e] = bu[e]; }
    wave_sync(wv);
    const int j = f.lane & 31, hi = f.lane >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int g = 0; g < 2; ++g)
            for (int e = 0; e < 8; ++e) {
                const float x = __builtin_bit_cast(float, (unsigned)wv.a16[p][i + 32 * g][e] << 16);
                const float y = __builtin_bit_cast(float, (unsigned)wv.b16[p][j + 32 * g][e] << 16);
                acc += x * y;
            }
        d[r] = acc;
    }
    return d;
}
f32x4_t mfma_16x16x4_f32(float a, float b, f32x4_t c) {
    Worker* w = tw; Fiber& f = w->fibers[w->cur]; Wave& wv = w->waves[f.wave];
    const int p = wave_parity(wv);
    wv.af[p][f.lane] = a; wv.bfv[p][f.lane] = b;
    wave_sync(wv);
    const int j = f.lane & 15, g4 = f.lane >> 4;
    f32x4_t d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) acc = std::fmaf(wv.af[p][i + 16 * g], wv.bfv[p][j + 16 * g], acc);
        d[r] = acc;
    }
    return d;
}

static void run_block(Worker* w, dim3 grid, dim3 block, unsigned bid, const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAXT) { fprintf(stderr, "emu: block too large\n"); abort(); }
    if ((int)w->fibers.size() < n) {
        const size_t old = w->fibers.size();
        w->fibers.resize(n);
        for (size_t i = old; i < (size_t)n; ++i) w->fibers[i].stack = (char*)malloc(STACK);
    }
    w->n = n; w->arrived = 0; w->gen = 0; w->body = &body;
    const int nw = (n + 63) / 64;
    for (int i = 0; i < nw; ++i) { w->waves[i].arrived = 0; w->waves[i].gen = 0; w->waves[i].n = std::min(64, n - 64 * i); }
    for (int t = 0; t < n; ++t) {
        Fiber& f = w->fibers[t];
        f.done = false; f.wave = t >> 6; f.lane = t & 63;
        f.ctx.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.ctx.bid = dim3(bid % grid.x, (bid / grid.x) % grid.y, bid / (grid.x * grid.y));
        f.ctx.bdim = block; f.ctx.gdim = grid;
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack; f.uc.uc_stack.ss_size = STACK; f.uc.uc_link = &w->main_uc;
        makecontext(&f.uc, (void (*)())fiber_entry, 0);
    }
    int remaining = n;
    while (remaining > 0) {
        for (int t = 0; t < n; ++t) {
            Fiber& f = w->fibers[t];
            if (f.done) continue;
            w->cur = t;
            swapcontext(&w->main_uc, &f.uc);
            if (f.done) --remaining;
        }
    }
    w->cur = -1;
}

}  // namespace emu

namespace emu {
static std::vector<Worker*> g_pool;
void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const unsigned nb = grid.x * grid.y * grid.z;
    static const unsigned nthreads = [] {
        const char* e = getenv("DSG_EMU_THREADS");
        unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return n < 1 ? 1u : (n > 16 ? 16u : n);
    }();
    if (g_pool.size() < nthreads) { const size_t o = g_pool.size(); g_pool.resize(nthreads); for (size_t i = o; i < nthreads; ++i) g_pool[i] = new Worker(); }
    std::atomic<unsigned> next{0};
    auto work = [&](unsigned wid) {
        tw = g_pool[wid];
        for (;;) {
            const unsigned b = next.fetch_add(1);
            if (b >= nb) break;
            run_block(tw, grid, block, b, body);
        }
        tw = nullptr;
    };
    const unsigned nt = std::min(nthreads, nb);
    if (nt <= 1) { work(0); return; }
    std::vector<std::thread> th;
    for (unsigned i = 1; i < nt; ++i) th.emplace_back(work, i);
    work(0);
    for (auto& t : th) t.join();
}
}  // namespace emu
