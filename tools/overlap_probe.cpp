// overlap_probe.cpp -- round-3 verdict item 4: "overlapped dependent packets" on the fence-free path.  A chain of N dependent kernels
// shaped like a batch-1 GEMM of the step (96 / 48 / 192 workgroups; 32 KB of weights + 8 KB of the predecessor's output in, 8 KB out;
// all hand-off buffers in UNCACHED device memory, packets without acquire / release fences -- exactly the product's loop), submitted
//   A) as the product does: every packet carries the barrier bit (the command processor starts packet i + 1 when packet i has retired);
//   B) overlapped: NO barrier bit; every workgroup requests its weights first, lane 0 then polls the predecessor's arrival counter
//      (uncached memory, sc1 loads, s_sleep between polls), the workgroup loads the activations, computes, drains its stores
//      (s_waitcnt vmcnt(0): an uncached store that is acknowledged is visible), and arrives with one atomic add -- one counter per
//      packet, or 8 per packet sharded by the arriving workgroup's XCD; optionally a barrier bit every M-th packet to bound the look-ahead.
// The chain's values are checked (out == in + 1 per link), a workgroup that gives up waiting counts as an error, every host wait is bounded.
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/overlap_kernels.hip -o tools/_build/overlap_kernels.hsaco
//   g++ -O2 -std=c++17 -I/opt/rocm/include tools/overlap_probe.cpp -L/opt/rocm/lib -lhsa-runtime64 -o tools/_build/overlap_probe
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); \
    printf("ERR %s @%d: %s\n", #x, __LINE__, m_ ? m_ : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_dev_pool;
static bool g_have_gpu = false, g_have_cpu = false, g_have_dev = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t dev_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; uint32_t flags = 0; bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) { g_dev_pool = p; g_have_dev = true; }
    return HSA_STATUS_SUCCESS;
}
static void* dev_alloc(size_t n, bool uncached) {
    void* p = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_dev_pool, n, uncached ? HSA_AMD_MEMORY_POOL_UNCACHED_FLAG : 0, &p));
    hsa_agent_t both[2] = {g_gpu, g_cpu};
    hsa_amd_agents_allow_access(2, both, nullptr, p);
    return p;
}

struct OvArgs {      // = tools/overlap_kernels.hip
    const void* act; const void* w; void* out; unsigned* wait_ctr; unsigned* arrive_ctr; unsigned* err;
    int nb, wait_target, shards, sleep, work;
    unsigned* l2_ctr;
};

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tools/_build/overlap_kernels.hsaco";
    CK(hsa_init());
    CK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { printf("no gpu/cpu agent\n"); return 1; }
    CK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
    if (!g_have_dev) { printf("no device pool\n"); return 1; }
    std::ifstream f(path, std::ios::binary);
    std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (img.empty()) { printf("cannot read %s\n", path); return 1; }
    hsa_code_object_reader_t rd; hsa_executable_t ex;
    CK(hsa_code_object_reader_create_from_memory(img.data(), img.size(), &rd));
    CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    CK(hsa_executable_freeze(ex, nullptr));
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(ex, "k_link.kd", &g_gpu, &sym));
    uint64_t kobj; uint32_t kgroup, kpriv, kasz;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &kgroup));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &kpriv));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kasz));
    hsa_queue_t* q; hsa_signal_t done;
    CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    CK(hsa_signal_create(1, 0, nullptr, &done));

    const int N = 2000, NBMAX = 192;
    const size_t KS = 128;                                            // kernarg stride
    float* a0 = (float*)dev_alloc(NBMAX * 8192, true); float* a1 = (float*)dev_alloc(NBMAX * 8192, true);
    float* w = (float*)dev_alloc((size_t)NBMAX * 32768, false);
    unsigned* ctr = (unsigned*)dev_alloc((size_t)(N + 1) * 8 * 64, true);      // per packet: 8 shards x 64 B
    unsigned* err = (unsigned*)dev_alloc(256, true);
    unsigned* l2c = (unsigned*)dev_alloc((size_t)(N + 1) * 16 * 64, false);      // per packet: 16 cached counters (8 XCD-local + 8 displaced) x 64 B
    char* ka_dev = (char*)dev_alloc(N * KS, false);
    CK(hsa_amd_memory_fill(w, 0, (size_t)NBMAX * 32768 / 4));
    printf("kernarg %u B, group %u, private %u; hand-off buffers + counters: uncached device memory; packets: no acquire / release fences\n", kasz, kgroup, kpriv);

    struct V { const char* nm; int nb; bool overlap; int shards, sleep, bar_every, work; };
    const V vs[] = {
        {"A  barrier bit on every packet (the product's loop)        96 WG", 96, false, 1, 0, 1, 0},
        {"B  overlapped, one counter, s_sleep 1                      96 WG", 96, true, 1, 1, 0, 0},
        {"B  overlapped, one counter, s_sleep 4                      96 WG", 96, true, 1, 4, 0, 0},
        {"B  overlapped, 8 counters by XCD, s_sleep 1                96 WG", 96, true, 8, 1, 0, 0},
        {"B  overlapped, one counter, s_sleep 1, barrier every 4th   96 WG", 96, true, 1, 1, 4, 0},
        {"B  overlapped, one counter, s_sleep 1, barrier every 2nd   96 WG", 96, true, 1, 1, 2, 0},
        {"C  overlapped, two-level arrival (L2 counter per XCD -> 8 uncached adds)  96 WG", 96, true, 0, 1, 0, 0},
        {"A  barrier bit                                             24 WG", 24, false, 1, 0, 1, 0},
        {"B  overlapped, one counter, s_sleep 1                      24 WG", 24, true, 1, 1, 0, 0},
        {"A  barrier bit                                             48 WG", 48, false, 1, 0, 1, 0},
        {"B  overlapped, one counter, s_sleep 1                      48 WG", 48, true, 1, 1, 0, 0},
        {"A  barrier bit                                            192 WG", 192, false, 1, 0, 1, 0},
        {"B  overlapped, one counter, s_sleep 1                     192 WG", 192, true, 1, 1, 0, 0},
        {"B  overlapped, 8 counters by XCD, s_sleep 1               192 WG", 192, true, 8, 1, 0, 0},
        {"C  overlapped, two-level arrival                          192 WG", 192, true, 0, 1, 0, 0},
        {"A  barrier bit, + a dependent chain of 120 FMAs per link   96 WG", 96, false, 1, 0, 1, 120},
        {"B  overlapped, one counter, + 120 FMAs                     96 WG", 96, true, 1, 1, 0, 120},
        {"C  overlapped, two-level arrival, + 120 FMAs               96 WG", 96, true, 0, 1, 0, 120},
    };
    for (const V& v : vs) {
        double best = 1e9; unsigned errs = 0; size_t bad = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hsa_amd_memory_fill(a0, 0, NBMAX * 8192 / 4)); CK(hsa_amd_memory_fill(a1, 0, NBMAX * 8192 / 4));
            CK(hsa_amd_memory_fill(ctr, 0, (size_t)(N + 1) * 8 * 64 / 4)); CK(hsa_amd_memory_fill(err, 0, 64));
            CK(hsa_amd_memory_fill(l2c, 0, (size_t)(N + 1) * 16 * 64 / 4));
            std::vector<char> stage(N * KS, 0);
            for (int i = 0; i < N; ++i) {
                OvArgs a;
                a.act = (i & 1) ? a1 : a0; a.w = w; a.out = (i & 1) ? a0 : a1;
                a.wait_ctr = v.overlap ? ctr + (size_t)i * 128 : nullptr;             // counters of packet i - 1 live at slot i
                a.arrive_ctr = v.overlap ? ctr + (size_t)(i + 1) * 128 : nullptr;
                a.l2_ctr = l2c + (size_t)(i + 1) * 256;
                a.err = err; a.nb = v.nb; a.wait_target = i == 0 ? 0 : v.nb; a.shards = v.shards; a.sleep = v.sleep; a.work = v.work;
                memcpy(stage.data() + i * KS, &a, sizeof a);
            }
            CK(hsa_memory_copy(ka_dev, stage.data(), stage.size()));
            hsa_signal_store_relaxed(done, 1);
            const uint64_t first = hsa_queue_add_write_index_relaxed(q, N);
            const uint32_t mask = q->size - 1;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((first + i) & mask);
                p->setup = 1;
                p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
                p->grid_size_x = (uint32_t)v.nb * 256; p->grid_size_y = 1; p->grid_size_z = 1;
                p->private_segment_size = kpriv; p->group_segment_size = kgroup;
                p->kernel_object = kobj; p->kernarg_address = ka_dev + (size_t)i * KS; p->reserved2 = 0;
                p->completion_signal.handle = (i == N - 1) ? done.handle : 0;
                const bool bar = !v.overlap || (v.bar_every > 0 && i % v.bar_every == 0) || i == N - 1;
                const int acq = i == 0 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE, rel = i == N - 1 ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_NONE;
                const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((bar ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                        (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
            }
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(first + N - 1));
            hsa_signal_value_t sv = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 20ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE);
            auto t1 = std::chrono::steady_clock::now();
            if (sv >= 1) { printf("%-70s : TIMEOUT\n", v.nm); return 2; }
            best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
            unsigned e2[2] = {0, 0}; CK(hsa_memory_copy(e2, err, 8)); errs += e2[0];
            std::vector<float> h((size_t)v.nb * 2048);
            CK(hsa_memory_copy(h.data(), (N & 1) ? a1 : a0, h.size() * 4));       // link N - 1 (odd) wrote a0 when N is even
            for (float x : h) if (x != (float)N) ++bad;
        }
        printf("%-70s : %6.2f us/link   %s%s\n", v.nm, best, bad == 0 ? "chain values OK" : "CHAIN VALUES WRONG", errs ? "  (workgroups gave up waiting!)" : "");
    }
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
