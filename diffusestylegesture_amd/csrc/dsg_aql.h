// dsg_aql.h -- the denoising-step loop submitted as hand-written AQL packets on an own HSA user-mode queue.
//
// Why: at batch 1 a denoising step is 34 dependent launches and the HIP runtime's per-launch floor dominates it
// (tools/aql_probe.cpp, profiles/r01_i_aql_probe.log, MI355X: a dependent cross-XCD kernel chain costs 2.11 us per
// packet when the packets are written by hand -- barrier bit, agent-scope acquire/release, kernel arguments resident
// in device memory -- against 2.9 us through hipLaunchKernelGGL; a null kernel 1.53 vs 2.45 us).  Everything a step
// needs that changes from step to step (timestep, coefficients, noise counter) already lives in device memory
// (StepCtl), so the 34 argument blocks are written ONCE per dsg_sample call and every step re-submits the same 34
// packets: the host's work per step is 34 x 64-byte stores and one doorbell write.
//
// What the probe also established and this file relies on: the release fence must be agent scope (with release NONE a
// consumer on another XCD reads stale data: the per-XCD L2s are not coherent with each other without the write-back),
// system scope costs +1.8 us per packet, and kernel arguments in host memory cost 14 us per packet.
//
// The kernels are the SAME device code: `make` emits the device side of dsg_hip.cpp as a bare code object
// (csrc/dsg_kernels.hsaco) next to the library; it is loaded here through the HSA loader, and a host kernel pointer is
// mapped to its kernel descriptor through the mangled name HIP reports for it (hipKernelNameRefByPtr).  HIP appends
// hidden arguments (grid size in blocks, ...) behind the explicit ones; hand-written packets must provide them.
// Every wait is bounded; any failure disables the path for the handle and the HIP launch path takes over.
#pragma once
#ifndef DSG_EMU
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

namespace dsg_aql {

struct Kernel { uint64_t object = 0; uint32_t kernarg_size = 0, group = 0, priv = 0; };
struct Launch { Kernel k; unsigned gx, gy, gz, bx, by, bz; size_t ka_off, hidden_off; std::string name; int fence; };

// Per-packet timeline of the step loop AS IT IS TIMED (diagnostics: dsg_debug_trace_arm / tools/aql_timeline.py).  rocprofv3
// cannot see hand-written packets, so the profile of this path comes from the path itself, in one of two ways:
//   product library: the queue is put in profiling mode and every packet of the traced steps carries its own completion signal;
//       hsa_amd_profiling_get_dispatch_time returns the command processor's start / end timestamps of each dispatch.  Costs
//       ~0.8 us per traced packet (the traced steps run ~20 % slower), and "start" is when the packet is taken up, not when its
//       first wave runs: busy includes the dispatch overhead, the gaps read 0.
//   timeline build (-DDSG_STAMPS, `make stamps`): every WAVE of a traced kernel stores the 100 MHz steady counter at its start and
//       its end into its own entry of a ring (dsg_kernels.h: TlScope); slot index and ring pointer travel in reserved dwords of
//       the packet's implicit-argument block -- the traced steps get their own copies of the argument blocks with the slots filled
//       in, every other packet carries slot -1.  first-wave start = min, last-wave end = max over the entries of a slot.
//       (A first version with one atomic min / max per wave on a shared slot made the traced steps 2.6x slower: ~12 ns per
//       contended atomic x 400-1200 waves per kernel.)
// The un-traced steps of the same run keep the plain packets.
struct Trace {
    bool armed = false;
    int first = 0, n = 0;                    // traced steps [first, first + n)
    std::vector<hsa_signal_t> sig;           // n x packets-per-step (product library)
    char* ka_trace = nullptr; size_t ka_trace_cap = 0;      // timeline build: n copies of the argument blocks with slot indices
    char* ring = nullptr; size_t ring_cap = 0;              // ... and the per-wave stamp ring
    std::vector<double> us;                  // result: [n][packets][2] = start, end in us relative to the first traced start
    std::vector<double> marks;               // marks build: [n][packets][12][3] = waves that passed mark k, mean, last wave (us after the packet's first wave start)
    std::vector<std::string> names;          // kernel of each packet of a step
    double traced_span_us = 0.0;             // first traced start -> last traced end
};

struct Ctx {
    bool tried = false, ready = false, recording = false;
    std::string err;
    hsa_agent_t gpu{};
    hsa_executable_t ex{};
    hsa_queue_t* q = nullptr;
    hsa_signal_t done{};
    std::map<const void*, Kernel> cache;
    std::vector<char> image;
    char* ka_dev = nullptr; size_t ka_cap = 0;
    std::vector<char> ka_host;
    std::vector<Launch> plan;
    double last_ms = 0.0;
    bool nofence = false;   // no acquire / release between the packets of the loop: everything the loop writes is coherent without
                            // cache maintenance (uncached buffers, dsg_hip.cpp uc_mode)
    char bdf[32] = {0};     // PCI address of the agent the queue lives on (== the HIP device's: checked in init)
    Trace trace;
};

inline bool hsa_ok(Ctx& c, hsa_status_t s, const char* what) {
    if (s == HSA_STATUS_SUCCESS) return true;
    const char* m = nullptr; hsa_status_string(s, &m);
    c.err = std::string(what) + ": " + (m ? m : "?");
    return false;
}
#define DSG_AQL_CK(c, x) do { if (!hsa_ok(c, (x), #x)) return false; } while (0)

struct AgentPick { std::vector<hsa_agent_t> gpus; };
inline hsa_status_t agent_cb(hsa_agent_t a, void* d) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU) ((AgentPick*)d)->gpus.push_back(a);
    return HSA_STATUS_SUCCESS;
}

// one-time set-up for a handle: agent of the HIP device (matched by PCI bus/device/function), code object, queue
inline bool init(Ctx& c, int hip_device, const void* addr_in_library) {
    if (c.tried) return c.ready;
    c.tried = true;
    DSG_AQL_CK(c, hsa_init());
    AgentPick pick;
    DSG_AQL_CK(c, hsa_iterate_agents(agent_cb, &pick));
    if (pick.gpus.empty()) { c.err = "no HSA GPU agent"; return false; }
    {   // The HSA agent must be the HIP device this handle was created on (one rank per GPU: torch.cuda.set_device(LOCAL_RANK) ->
        // dsg_config.device): matched by PCI domain:bus:device.function, ALSO when only one agent is visible -- a queue on the
        // wrong GPU would run the packets against memory of another device.
        char bus[64] = {0};
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (hipDeviceGetPCIBusId(bus, sizeof bus, hip_device) != hipSuccess || sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) {
            c.err = "cannot read the PCI id of the HIP device"; return false;
        }
        bool found = false;
        for (hsa_agent_t a : pick.gpus) {
            uint32_t bdf = 0, domain = 0;
            hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
            hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
            if (bdf == ((b << 8) | (d << 3) | f) && domain == dom) { c.gpu = a; found = true; break; }
        }
        if (!found) { c.err = std::string("no HSA agent matches the HIP device's PCI id ") + bus; return false; }
        snprintf(c.bdf, sizeof c.bdf, "%04x:%02x:%02x.%x", dom, b, d, f);
        static std::atomic<bool> said[64];   // once per device and process (= once per rank)
        if (getenv("LOCAL_RANK") && !said[hip_device & 63].exchange(true)) {
            fprintf(stderr, "libdsg_hip: rank-local HIP device %d <-> HSA agent %s: AQL queue created there (%zu GPU agent(s) visible)\n",
                    hip_device, c.bdf, pick.gpus.size());
        }
    }
    Dl_info info;
    if (!dladdr(addr_in_library, &info) || !info.dli_fname) { c.err = "dladdr failed"; return false; }
    std::string path(info.dli_fname);
#if defined(DSG_HSACO_NAME)         // (development builds, Makefile: `make dev` -- bf16 only, own file names)
    path = path.substr(0, path.find_last_of('/') + 1) + DSG_HSACO_NAME;
#elif defined(DSG_STAMPS)
    path = path.substr(0, path.find_last_of('/') + 1) + (DSG_STAMPS >= 2 ? "dsg_kernels_marks.hsaco" : "dsg_kernels_stamps.hsaco");
#else
    path = path.substr(0, path.find_last_of('/') + 1) + "dsg_kernels.hsaco";
#endif
    std::ifstream f(path, std::ios::binary);
    c.image.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (c.image.empty()) { c.err = "cannot read " + path + " (run make)"; return false; }
    hsa_code_object_reader_t rd;
    DSG_AQL_CK(c, hsa_code_object_reader_create_from_memory(c.image.data(), c.image.size(), &rd));
    DSG_AQL_CK(c, hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &c.ex));
    DSG_AQL_CK(c, hsa_executable_load_agent_code_object(c.ex, c.gpu, rd, nullptr, nullptr));
    DSG_AQL_CK(c, hsa_executable_freeze(c.ex, nullptr));
    {   // the code object must come from the same sources as this library (same argument structs, same kernels)
        hsa_executable_symbol_t sym;
        uint64_t addr = 0;
        unsigned tag = ~0u;
        DSG_AQL_CK(c, hsa_executable_get_symbol_by_name(c.ex, "dsg_device_build_tag", &c.gpu, &sym));
        DSG_AQL_CK(c, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr));
        if (hsa_memory_copy(&tag, (const void*)addr, sizeof tag) != HSA_STATUS_SUCCESS || tag != (unsigned)DSG_BUILD_TAG) {
            c.err = path + " was built from different sources than the library (run make)";
            return false;
        }
    }
    DSG_AQL_CK(c, hsa_queue_create(c.gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &c.q));
    if (const char* e = getenv("DSG_CU_MASK")) {      // measurement only (tools/multiproc.py): comma-separated 32-bit hex words, CU bit 0 first
        std::vector<uint32_t> words;
        for (const char* p = e; *p;) {
            char* end = nullptr;
            words.push_back((uint32_t)strtoul(p, &end, 16));
            if (end == p) break;
            p = *end == ',' ? end + 1 : end;
        }
        if (!words.empty() && hsa_amd_queue_cu_set_mask(c.q, (uint32_t)words.size() * 32, words.data()) != HSA_STATUS_SUCCESS)
            fprintf(stderr, "libdsg_hip: DSG_CU_MASK: hsa_amd_queue_cu_set_mask failed (mask ignored)\n");
    }
    DSG_AQL_CK(c, hsa_signal_create(1, 0, nullptr, &c.done));
    c.ready = true;
    return true;
}

inline bool lookup(Ctx& c, const void* host_fn, hipStream_t stream, Kernel& out) {
    auto it = c.cache.find(host_fn);
    if (it != c.cache.end()) { out = it->second; return true; }
    const char* name = hipKernelNameRefByPtr(host_fn, stream);
    if (!name) { c.err = "hipKernelNameRefByPtr returned null"; return false; }
    const std::string kd = std::string(name) + ".kd";
    hsa_executable_symbol_t sym;
    DSG_AQL_CK(c, hsa_executable_get_symbol_by_name(c.ex, kd.c_str(), &c.gpu, &sym));
    Kernel k;
    DSG_AQL_CK(c, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    DSG_AQL_CK(c, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
    DSG_AQL_CK(c, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    DSG_AQL_CK(c, hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    if (k.priv != 0) { c.err = std::string(name) + " needs scratch memory, which this queue does not provide"; return false; }
    c.cache[host_fn] = k;
    out = k;
    return true;
}

// one launch of the step: explicit argument struct + the hidden arguments of code object v5 behind it
// (llvm AMDGPUUsage "Code Object V5 Kernel Argument": block counts u32 x3 at +0, group sizes u16 x3 at +12, remainders
// u16 x3 at +18, global offsets u64 x3 at +40, grid dims u16 at +64)
// fence (fence-free plans only): 1 = this packet acquires at agent scope, 2 = it releases -- the two packets of a step that touch
// the cached sampler state (dsg_hip.cpp: state_fences)
inline bool record(Ctx& c, const void* host_fn, hipStream_t stream, dim3 grid, dim3 block, const void* args, size_t size, int fence = 0) {
    Kernel k;
    if (!lookup(c, host_fn, stream, k)) return false;
    const size_t hidden = (size + 7) & ~(size_t)7;
    const size_t need = std::max<size_t>(k.kernarg_size, hidden + 144);
    const size_t off = (c.ka_host.size() + 255) & ~(size_t)255;
    c.ka_host.resize(off + ((need + 255) & ~(size_t)255), 0);
    char* p = c.ka_host.data() + off;
    std::memcpy(p, args, size);
    const uint32_t bc[3] = {grid.x, grid.y, grid.z};
    const uint16_t gs[3] = {(uint16_t)block.x, (uint16_t)block.y, (uint16_t)block.z};
    std::memcpy(p + hidden + 0, bc, sizeof bc);
    std::memcpy(p + hidden + 12, gs, sizeof gs);
    const uint16_t dims = 3;
    std::memcpy(p + hidden + 64, &dims, 2);
    const int no_slot = -1;                  // timeline build: "not traced" (dsg_kernels.h: DSG_TL_ARG_OFF); unused otherwise
    std::memcpy(p + hidden + 128, &no_slot, 4);
    const char* nm = hipKernelNameRefByPtr(host_fn, stream);
    c.plan.push_back(Launch{k, grid.x, grid.y, grid.z, block.x, block.y, block.z, off, hidden, nm ? nm : "?", fence});
    return true;
}

inline void begin(Ctx& c) { c.plan.clear(); c.ka_host.clear(); c.recording = true; c.nofence = false; }

// argument blocks -> device memory (once per dsg_sample call)
inline bool finish(Ctx& c) {
    c.recording = false;
    if (c.plan.empty()) { c.err = "empty plan"; return false; }
    if (c.ka_host.size() > c.ka_cap) {
        if (c.ka_dev) (void)hipFree(c.ka_dev);
        c.ka_cap = c.ka_host.size() * 2;
        if (hipMalloc((void**)&c.ka_dev, c.ka_cap) != hipSuccess) { c.ka_dev = nullptr; c.ka_cap = 0; c.err = "hipMalloc(kernarg)"; return false; }
    }
    if (hipMemcpy(c.ka_dev, c.ka_host.data(), c.ka_host.size(), hipMemcpyHostToDevice) != hipSuccess) { c.err = "hipMemcpy(kernarg)"; return false; }
    return hipDeviceSynchronize() == hipSuccess;
}

// packets of one step of `c` into its queue + doorbell; `last`: the final packet releases to system scope and carries the
// completion signal
inline void submit_step(Ctx& c, bool first_step, bool last_step, int step_index = -1) {
    const bool traced = c.trace.armed && step_index >= c.trace.first && step_index < c.trace.first + c.trace.n && !last_step;
    const uint32_t mask = c.q->size - 1;
    const size_t L = c.plan.size();
    const uint64_t first = hsa_queue_add_write_index_relaxed(c.q, L);
    for (size_t i = 0; i < L; ++i) {
        const Launch& l = c.plan[i];
        const bool last = last_step && (i == L - 1);
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)c.q->base_address + ((first + i) & mask);
        p->setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = (uint16_t)l.bx; p->workgroup_size_y = (uint16_t)l.by; p->workgroup_size_z = (uint16_t)l.bz;
        p->reserved0 = 0;
        p->grid_size_x = l.gx * l.bx; p->grid_size_y = l.gy * l.by; p->grid_size_z = l.gz * l.bz;
        p->private_segment_size = 0; p->group_segment_size = l.k.group;
        p->kernel_object = l.k.object;
        p->kernarg_address = c.ka_dev + l.ka_off;
#ifdef DSG_STAMPS
        if (traced) p->kernarg_address = c.trace.ka_trace + (size_t)(step_index - c.trace.first) * c.ka_host.size() + l.ka_off;
#endif
        p->reserved2 = 0;
#ifdef DSG_STAMPS
        p->completion_signal.handle = last ? c.done.handle : 0;
#else
        p->completion_signal.handle = last ? c.done.handle : (traced ? c.trace.sig[(size_t)(step_index - c.trace.first) * L + i].handle : 0);
#endif
        // fence-free plan: only the first packet acquires (everything the set-up kernels wrote) and only the last one releases
        // (the samples); a kernel's stores have been acknowledged when its waves end (s_endpgm waits for them) and the barrier
        // bit orders the packets
        const int mid_scope = c.nofence ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT;
        const int rel = last ? HSA_FENCE_SCOPE_SYSTEM : (l.fence == 2 ? HSA_FENCE_SCOPE_AGENT : mid_scope);
        const int acq = first_step && i == 0 ? HSA_FENCE_SCOPE_SYSTEM : (l.fence == 1 ? HSA_FENCE_SCOPE_AGENT : mid_scope);
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                           (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                           (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
    }
    hsa_signal_store_screlease(c.q->doorbell_signal, (hsa_signal_value_t)(first + L - 1));
}
// room for one more step in the queue?  (never overwrite packets the command processor has not consumed yet)
inline bool has_room(Ctx& c) {
    return hsa_queue_load_write_index_relaxed(c.q) + c.plan.size() - hsa_queue_load_read_index_scacquire(c.q) <= c.q->size;
}

// n_steps x plan, in order (barrier bit), agent-scope fences.  Returns false on time-out (seconds) -- the caller must treat
// the handle's state as lost.
inline bool run(Ctx& c, int n_steps, double timeout_s) {
    hsa_signal_store_relaxed(c.done, 1);
    Trace& tr = c.trace;
    const size_t L = c.plan.size();
#ifdef DSG_STAMPS
    typedef dsg::TlEntry Entry;
    constexpr size_t TLW = dsg::DSG_TL_WAVES;
#endif
    if (tr.armed) {
        tr.n = std::max(0, std::min(tr.n, n_steps - 1 - tr.first));      // never the last step (its last packet carries the completion signal)
#ifdef DSG_STAMPS
        if ((size_t)tr.n * L > 1024) tr.n = (int)(1024 / L);
        const size_t ring_bytes = (size_t)tr.n * L * TLW * sizeof(Entry);
        if (ring_bytes > tr.ring_cap) {
            if (tr.ring) (void)hipFree(tr.ring);
            tr.ring_cap = ring_bytes;
            if (hipMalloc((void**)&tr.ring, tr.ring_cap) != hipSuccess) { tr.ring = nullptr; tr.ring_cap = 0; c.err = "trace: hipMalloc(ring)"; return false; }
        }
        std::vector<char> blocks((size_t)tr.n * c.ka_host.size());
        for (int s = 0; s < tr.n; ++s) {
            std::memcpy(blocks.data() + (size_t)s * c.ka_host.size(), c.ka_host.data(), c.ka_host.size());
            for (size_t i = 0; i < L; ++i) {
                const int slot = (int)(s * L + i);
                char* hp = blocks.data() + (size_t)s * c.ka_host.size() + c.plan[i].ka_off + c.plan[i].hidden_off;
                std::memcpy(hp + 128, &slot, 4);
                std::memcpy(hp + 136, &tr.ring, 8);
            }
        }
        if (blocks.size() > tr.ka_trace_cap) {
            if (tr.ka_trace) (void)hipFree(tr.ka_trace);
            tr.ka_trace_cap = blocks.size();
            if (hipMalloc((void**)&tr.ka_trace, tr.ka_trace_cap) != hipSuccess) { tr.ka_trace = nullptr; tr.ka_trace_cap = 0; c.err = "trace: hipMalloc"; return false; }
        }
        if ((!blocks.empty() && hipMemcpy(tr.ka_trace, blocks.data(), blocks.size(), hipMemcpyHostToDevice) != hipSuccess) ||
            (ring_bytes && hipMemset(tr.ring, 0, ring_bytes) != hipSuccess) || hipDeviceSynchronize() != hipSuccess) {
            c.err = "trace: set-up copy failed"; return false;
        }
#else
        tr.sig.resize((size_t)tr.n * L);
        bool ok = hsa_amd_profiling_set_profiler_enabled(c.q, 1) == HSA_STATUS_SUCCESS;
        for (auto& sgn : tr.sig) ok = ok && hsa_signal_create(1, 0, nullptr, &sgn) == HSA_STATUS_SUCCESS;
        if (!ok) { c.err = "trace: cannot enable queue profiling"; return false; }
#endif
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < n_steps; ++s) {
        while (!has_room(c)) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { c.err = "queue stalled"; return false; }
        }
        submit_step(c, s == 0, s == n_steps - 1, s);
    }
    const uint64_t budget_ns = (uint64_t)(timeout_s * 1e9);
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(c.done, HSA_SIGNAL_CONDITION_LT, 1, budget_ns, HSA_WAIT_STATE_ACTIVE);
    c.last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (v >= 1) { c.err = "completion signal timed out"; return false; }
#ifdef DSG_STAMPS
    if (tr.armed) {
        int khz = 100000;
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
        std::vector<Entry> got((size_t)tr.n * L * TLW);
        if (!got.empty() && hipMemcpy(got.data(), tr.ring, got.size() * sizeof(Entry), hipMemcpyDeviceToHost) != hipSuccess) { c.err = "trace: read-back failed"; return false; }
        const size_t NS = (size_t)tr.n * L;
        std::vector<unsigned long long> t0(NS, ~0ull), t1(NS, 0ull);
        for (size_t sl = 0; sl < NS; ++sl)
            for (size_t w = 0; w < TLW; ++w) {
                const Entry& en = got[sl * TLW + w];
                if (en.t0 == 0) continue;                        // no wave wrote this entry
                t0[sl] = std::min(t0[sl], en.t0); t1[sl] = std::max(t1[sl], en.t1);
            }
        tr.us.assign(NS * 2, 0.0);
        tr.names.clear();
        for (const Launch& l : c.plan) tr.names.push_back(l.name);
        const unsigned long long base = NS ? t0[0] : 0ull;
        unsigned long long last_end = base;
        for (size_t i = 0; i < NS; ++i) {
            tr.us[2 * i] = (double)(long long)(t0[i] - base) * 1000.0 / khz;
            tr.us[2 * i + 1] = (double)(long long)(t1[i] - base) * 1000.0 / khz;
            last_end = std::max(last_end, t1[i]);
        }
        tr.traced_span_us = (double)(last_end - base) * 1000.0 / khz;
#if DSG_STAMPS >= 2
        constexpr int NM = dsg::DSG_TL_NMARK;
        tr.marks.assign(NS * NM * 3, 0.0);
        for (size_t sl = 0; sl < NS; ++sl)
            for (int k = 0; k < NM; ++k) {
                double cnt = 0.0, sum = 0.0, mx = 0.0;
                for (size_t w = 0; w < TLW; ++w) {
                    const Entry& en = got[sl * TLW + w];
                    if (en.t0 == 0 || en.m[k] == 0) continue;
                    const double d = (double)(long long)(en.m[k] - t0[sl]) * 1000.0 / khz;
                    cnt += 1.0; sum += d; mx = std::max(mx, d);
                }
                tr.marks[(sl * NM + k) * 3] = cnt; tr.marks[(sl * NM + k) * 3 + 1] = cnt > 0 ? sum / cnt : 0.0; tr.marks[(sl * NM + k) * 3 + 2] = mx;
            }
#endif
        tr.armed = false;
    }
#else
    if (tr.armed) {
        uint64_t freq = 0;
        hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
        const double to_us = freq ? 1e6 / (double)freq : 0.0;
        tr.us.assign(tr.sig.size() * 2, 0.0);
        tr.names.clear();
        for (const Launch& l : c.plan) tr.names.push_back(l.name);
        uint64_t base = 0, last_end = 0;
        for (size_t i = 0; i < tr.sig.size(); ++i) {
            hsa_amd_profiling_dispatch_time_t t{};
            if (hsa_amd_profiling_get_dispatch_time(c.gpu, tr.sig[i], &t) != HSA_STATUS_SUCCESS) { c.err = "trace: no dispatch time"; return false; }
            if (i == 0) base = t.start;
            tr.us[2 * i] = (double)(t.start - base) * to_us;
            tr.us[2 * i + 1] = (double)(t.end - base) * to_us;
            last_end = std::max(last_end, t.end);
        }
        tr.traced_span_us = (double)(last_end - base) * to_us;
        for (auto& sgn : tr.sig) hsa_signal_destroy(sgn);
        tr.sig.clear();
        hsa_amd_profiling_set_profiler_enabled(c.q, 0);
        tr.armed = false;
    }
#endif
    return true;
}

// Several lanes at once, one host thread: every lane's chain of dependent packets lives on its own queue, so the command
// processor overlaps them; the host deals one step at a time to every lane that has room (a full queue is skipped, never
// waited on, so one slow lane cannot starve the others).  last_ms of each lane = first doorbell of the call to ITS completion.
inline bool run_multi(Ctx** cs, const int* n_steps, int n, double timeout_s, std::string& err) {
    std::vector<int> next(n, 0);
    for (int i = 0; i < n; ++i) hsa_signal_store_relaxed(cs[i]->done, 1);
    const auto t0 = std::chrono::steady_clock::now();
    int open = 0;
    for (int i = 0; i < n; ++i) open += n_steps[i] > 0;
    unsigned idle = 0;
    while (open > 0) {
        bool any = false;
        for (int i = 0; i < n; ++i) {
            if (next[i] >= n_steps[i] || !has_room(*cs[i])) continue;
            submit_step(*cs[i], next[i] == 0, next[i] == n_steps[i] - 1, -1);
            if (++next[i] == n_steps[i]) --open;
            any = true;
        }
        if (!any && (++idle & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { err = "queues stalled"; return false; }
    }
    for (int i = 0; i < n; ++i) {
        if (n_steps[i] <= 0) { cs[i]->last_ms = 0.0; continue; }
        const double left = timeout_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(cs[i]->done, HSA_SIGNAL_CONDITION_LT, 1, (uint64_t)(std::max(left, 1.0) * 1e9), HSA_WAIT_STATE_ACTIVE);
        cs[i]->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (v >= 1) { err = cs[i]->err = "completion signal timed out"; return false; }
    }
    return true;
}

inline void destroy(Ctx& c) {
    if (c.q) hsa_queue_destroy(c.q);
    if (c.ready) { hsa_signal_destroy(c.done); hsa_executable_destroy(c.ex); }
    if (c.ka_dev) (void)hipFree(c.ka_dev);
    if (c.trace.ka_trace) (void)hipFree(c.trace.ka_trace);
    if (c.trace.ring) (void)hipFree(c.trace.ring);
    if (c.tried) hsa_shut_down();
    c = Ctx();
}

}  // namespace dsg_aql
#endif  // DSG_EMU
