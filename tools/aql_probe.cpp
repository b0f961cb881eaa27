// aql_probe.cpp -- what does a dependent kernel launch cost when the AQL packets are written by hand (HSA user-mode
// queue, barrier bit, chosen fence scopes, kernargs in device memory), i.e. without the HIP runtime in the launch path?
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/aql_kernels.hip -o tools/_build/aql_kernels.hsaco
//   g++ -O2 -std=c++17 -I/opt/rocm/include tools/aql_probe.cpp -L/opt/rocm/lib -lhsa-runtime64 -o tools/_build/aql_probe
//   tools/_build/aql_probe tools/_build/aql_kernels.hsaco
// Every wait is bounded (the probe gives up instead of hanging the box).
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); \
    printf("ERR %s @%d: %s\n", #x, __LINE__, m_ ? m_ : "?"); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_dev_pool, g_kernarg_pool;
static bool g_have_gpu = false, g_have_cpu = false, g_have_dev = false, g_have_ka = false;

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
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; uint32_t flags = 0; bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_ka) { g_kernarg_pool = p; g_have_ka = true; }
    return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg_size, group, priv; };
static Kernel get_kernel(hsa_executable_t ex, const char* name) {
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(ex, name, &g_gpu, &sym));
    Kernel k;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

static void* dev_alloc(size_t n) {
    void* p = nullptr;
    CK(hsa_amd_memory_pool_allocate(g_dev_pool, n, 0, &p));
    hsa_agent_t both[2] = {g_gpu, g_cpu};
    hsa_amd_agents_allow_access(2, both, nullptr, p);      // large-BAR boxes: the host may write kernargs into device memory
    return p;
}

struct Chain {
    hsa_queue_t* q; hsa_signal_t done;
    // submit n dispatches of kernel k (grid nb x 256); kernargs[i] at ka + i*stride; returns us per packet, <0 on timeout
    double run(const Kernel& k, int n, int nb, char* ka, size_t stride, int acq, int rel, bool barrier_bit) {
        hsa_signal_store_relaxed(done, 1);
        const uint64_t first = hsa_queue_add_write_index_relaxed(q, n);
        const uint32_t mask = q->size - 1;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((first + i) & mask);
            p->setup = 1;                         // 1 dimension
            p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = (uint32_t)nb * 256; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = k.priv; p->group_segment_size = k.group;
            p->kernel_object = k.object;
            p->kernarg_address = ka + (size_t)i * stride;
            p->reserved2 = 0;
            p->completion_signal.handle = (i == n - 1) ? done.handle : 0;
            uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier_bit ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                              (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(first + n - 1));
        // bounded wait: 5 s
        hsa_signal_value_t v = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 5ull * 1000 * 1000 * 1000, HSA_WAIT_STATE_ACTIVE);
        auto t1 = std::chrono::steady_clock::now();
        if (v >= 1) return -1.0;
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
    }
};

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tools/_build/aql_kernels.hsaco";
    CK(hsa_init());
    CK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have_gpu || !g_have_cpu) { printf("no gpu/cpu agent\n"); return 1; }
    CK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
    if (!g_have_dev || !g_have_ka) { printf("pools missing dev=%d kernarg=%d\n", g_have_dev, g_have_ka); return 1; }
    char name[64]; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name); printf("agent %s\n", name);

    std::ifstream f(path, std::ios::binary);
    std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (img.empty()) { printf("cannot read %s\n", path); return 1; }
    hsa_code_object_reader_t rd; hsa_executable_t ex;
    CK(hsa_code_object_reader_create_from_memory(img.data(), img.size(), &rd));
    CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    CK(hsa_executable_freeze(ex, nullptr));
    Kernel knull = get_kernel(ex, "k_null.kd"), kdep = get_kernel(ex, "k_dep.kd"), kwt = get_kernel(ex, "k_dep_wt.kd");
    printf("k_null kernarg %u B, k_dep kernarg %u B group %u priv %u\n", knull.kernarg_size, kdep.kernarg_size, kdep.group, kdep.priv);

    Chain c;
    CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &c.q));
    CK(hsa_signal_create(1, 0, nullptr, &c.done));

    const int N = 2048, NB = 96;
    const size_t KS = 64;                                           // kernarg stride
    // buffers
    float* a0 = (float*)dev_alloc(NB * 8192); float* a1 = (float*)dev_alloc(NB * 8192); float* w = (float*)dev_alloc((size_t)NB * 32768);
    CK(hsa_amd_memory_fill(a0, 0, NB * 8192 / 4)); CK(hsa_amd_memory_fill(a1, 0, NB * 8192 / 4)); CK(hsa_amd_memory_fill(w, 0, (size_t)NB * 32768 / 4));
    char* ka_host = nullptr; CK(hsa_amd_memory_pool_allocate(g_kernarg_pool, N * KS, 0, (void**)&ka_host));
    hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka_host);
    char* ka_dev = (char*)dev_alloc(N * KS);
    std::vector<char> stage(N * KS, 0);
    for (int i = 0; i < N; ++i) {
        void* args[3] = {(i & 1) ? (void*)a1 : (void*)a0, (void*)w, (i & 1) ? (void*)a0 : (void*)a1};
        memcpy(stage.data() + i * KS, args, sizeof(args));
        const int nbv = NB; memcpy(stage.data() + i * KS + 24, &nbv, 4);
    }
    memcpy(ka_host, stage.data(), stage.size());
    CK(hsa_memory_copy(ka_dev, stage.data(), stage.size()));
    std::vector<char> znull(N * KS, 0);
    char* ka_null = (char*)dev_alloc(N * KS); CK(hsa_memory_copy(ka_null, znull.data(), znull.size()));

    const int A = HSA_FENCE_SCOPE_AGENT, S = HSA_FENCE_SCOPE_SYSTEM, Z = HSA_FENCE_SCOPE_NONE;
    struct V { const char* nm; const Kernel* k; char* ka; int acq, rel; bool bar; int nb; };
    V vs[] = {
        {"null  agent/agent  barrier", &knull, ka_null, A, A, true, NB},
        {"null  none/none    barrier", &knull, ka_null, Z, Z, true, NB},
        {"null  system/system barrier", &knull, ka_null, S, S, true, NB},
        {"null  agent/agent  NO barrier bit", &knull, ka_null, A, A, false, NB},
        {"dep   agent/agent  barrier, kernarg device", &kdep, ka_dev, A, A, true, NB},
        {"dep   agent/agent  barrier, kernarg host pool", &kdep, ka_host, A, A, true, NB},
        {"dep   none/none    barrier, kernarg device", &kdep, ka_dev, Z, Z, true, NB},
        {"dep   agent/none   barrier, kernarg device", &kdep, ka_dev, A, Z, true, NB},
        {"dep   none/agent   barrier, kernarg device", &kdep, ka_dev, Z, A, true, NB},
        {"dep   system/system barrier, kernarg device", &kdep, ka_dev, S, S, true, NB},
        {"dep   agent/agent  barrier, kernarg device, 24 workgroups", &kdep, ka_dev, A, A, true, 24},
        {"dep write-through stores  agent/agent", &kwt, ka_dev, A, A, true, NB},
        {"dep write-through stores  agent/none (no release fence)", &kwt, ka_dev, A, Z, true, NB},
        {"dep write-through stores  none/none", &kwt, ka_dev, Z, Z, true, NB},
    };
    for (const V& v : vs) {
        double best = 1e9;
        for (int r = 0; r < 4; ++r) {
            if (v.k != &knull) { CK(hsa_amd_memory_fill(a0, 0, NB * 8192 / 4)); CK(hsa_amd_memory_fill(a1, 0, NB * 8192 / 4)); }
            const double us = c.run(*v.k, N, v.nb, v.ka, KS, v.acq, v.rel, v.bar);
            if (us < 0) { printf("%-58s : TIMEOUT\n", v.nm); return 2; }
            if (us < best) best = us;
        }
        // correctness of the dependent chain: after N launches every element written by the last launch == N
        double bad = -1;
        if (v.k != &knull && v.nb == NB) {
            std::vector<float> h(NB * 2048);
            CK(hsa_memory_copy(h.data(), (N & 1) ? a1 : a0, h.size() * 4));      // launch N-1 (odd index) wrote a0 when N even
            size_t nb = 0; for (float x : h) if (x != (float)N) ++nb;
            bad = (double)nb;
        }
        printf("%-58s : %6.2f us/packet%s\n", v.nm, best, bad < 0 ? "" : (bad == 0 ? "   chain values OK" : "   CHAIN VALUES WRONG"));
        if (bad > 0) printf("   (%g wrong elements)\n", bad);
    }
    hsa_queue_destroy(c.q);
    hsa_shut_down();
    return 0;
}
