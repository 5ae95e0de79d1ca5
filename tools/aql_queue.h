// User-mode AQL dispatch: a hardware queue of one's own, fed with pre-built kernel-dispatch packets (probe tool).
//
// Why it was tried (round 4): the software-pipelined launches (HISTORY.md section 4.3) need launch j+1 dispatched while launch j
// runs.  HIP offers that only across streams -- i.e. across hardware queues -- and refuses hipExtAnyOrderLaunch on gfx9.  An AQL
// packet carries the ordering itself: with the BARRIER bit clear the packet processor starts the next packet of the SAME queue
// as soon as the previous one's workgroups are placed, and its header says which cache actions surround the dispatch.  The
// packets of a step are identical for every position (the position lives in device memory), so they are built once; a replay is
// one memcpy of the packets into the ring plus one doorbell write (host cost: nothing measurable).
// What it measured (tools/pipelined_launch_probe.hip, profiles/r04_pipelined_launch_probe_v1_aql.log, the 1920 x 1920 phase):
//   barrier = 1, acquire / release at agent scope   8.3 us per phase  (HIP's own launch chain: 6.7 -- its packets fence less)
//   barrier = 0, acquire / release at agent scope   7.7
//   barrier = 0, no fences                          5.6  (valid for kernels that hand over with sc1 accesses only)
//   two HIP streams, barrier = 1 each                4.7  <- what the engine uses
// i.e. one queue without barriers pipelines less than two queues do (the packet processor still walks one queue's packets one
// after the other), and an agent-scope acquire costs 2 us per packet.  Not used by the library.
//
// The kernels stay ordinary HIP kernels: their descriptors are found in the executables HIP has loaded
// (hsa_ven_amd_loader_iterate_executables) under the name hipKernelNameRefByPtr gives, so there is one code object, one
// copy of every kernel, and the same kernel can still be launched through HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace jb_aql {

struct Kernel {
    uint64_t object = 0;                 // address of the kernel descriptor
    uint32_t kernarg_size = 0;           // explicit + implicit arguments
    uint32_t group_size = 0;             // static LDS
    uint32_t private_size = 0;           // scratch per work-item (the decode kernels: 0)
};

// Implicit arguments of code-object-v5 kernels that the launch owns (the block behind the explicit arguments, 8-byte
// aligned): block counts, workgroup sizes, remainders; the rest (printf / hostcall / heap / queue pointers) stays zero --
// the decode kernels use none of them.
struct ImplicitHead {
    uint32_t block_count[3];
    uint16_t group_size[3];
    uint16_t remainder[3];
    uint8_t reserved[16];
    uint64_t global_offset[3];
    uint16_t grid_dims;
};

inline uint16_t header(bool barrier, int acquire, int release) {
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

class Queue {
  public:
    std::string error;
    std::atomic<bool> dead{false};

    ~Queue() { destroy(); }

    // The hardware queue on the HSA agent behind HIP device `hip_device`; `packets`: ring size (power of two).
    bool create(int hip_device, uint32_t packets) {
        destroy();
        if (hsa_init() != HSA_STATUS_SUCCESS) return fail("hsa_init");
        inited_ = true;
        char bus[32] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), hip_device) != hipSuccess) return fail("hipDeviceGetPCIBusId");
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return fail(std::string("unparsable PCI bus id ") + bus);
        Find want{dom, (b << 8) | (d << 3) | f, {0}, false};
        hsa_iterate_agents(&Queue::find_agent, &want);
        if (!want.found) return fail(std::string("no HSA GPU agent at ") + bus);
        agent_ = want.agent;
        uint32_t max_size = 0;
        hsa_agent_get_info(agent_, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &max_size);
        if (packets > max_size) packets = max_size;
        if (hsa_queue_create(agent_, packets, HSA_QUEUE_TYPE_SINGLE, &Queue::queue_error, this, UINT32_MAX, UINT32_MAX, &q_) != HSA_STATUS_SUCCESS)
            return fail("hsa_queue_create");
        if (hsa_signal_create(1, 0, nullptr, &done_) != HSA_STATUS_SUCCESS) return fail("hsa_signal_create");
        if (hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(loader_), &loader_) != HSA_STATUS_SUCCESS ||
            !loader_.hsa_ven_amd_loader_iterate_executables)
            return fail("AMD loader extension 1.03 (iterate_executables) is not available");
        return true;
    }

    void destroy() {
        if (done_.handle) hsa_signal_destroy(done_);
        if (q_) hsa_queue_destroy(q_);
        done_.handle = 0; q_ = nullptr;
        if (inited_) hsa_shut_down();          // reference counted: HIP's own hsa_init keeps the runtime alive
        inited_ = false;
    }

    bool ok() const { return q_ != nullptr; }
    uint32_t size() const { return q_ ? q_->size : 0; }

    // Keep the queue off compute units (bit i of mask = compute unit i / 8 of XCD i % 8 on MI355X) / set its priority.
    bool set_cu_mask(const uint32_t* mask, uint32_t bits) {
        return hsa_amd_queue_cu_set_mask(q_, bits, mask) == HSA_STATUS_SUCCESS || fail("hsa_amd_queue_cu_set_mask");
    }
    bool set_priority(hsa_amd_queue_priority_t p) {
        return hsa_amd_queue_set_priority(q_, p) == HSA_STATUS_SUCCESS || fail("hsa_amd_queue_set_priority");
    }

    // Descriptor of a __global__ function of this process (HIP must have loaded its module: launch something first).
    bool lookup(const void* host_fn, Kernel* out) {
        const char* name = hipKernelNameRefByPtr(host_fn, nullptr);
        if (!name) return fail("hipKernelNameRefByPtr: not a kernel of this process");
        Sym s{this, std::string(name) + ".kd", out, false};
        loader_.hsa_ven_amd_loader_iterate_executables(&Queue::find_symbol, &s);
        if (!s.found) return fail("kernel descriptor " + s.name + " not found in the loaded executables (module not loaded yet?)");
        return true;
    }

    // A dispatch packet; `header` is kept aside (it is written last).
    static hsa_kernel_dispatch_packet_t packet(const Kernel& k, dim3 blocks, dim3 threads, uint32_t dynamic_lds, const void* kernarg) {
        hsa_kernel_dispatch_packet_t p;
        memset(&p, 0, sizeof(p));
        const int dims = blocks.z > 1 ? 3 : (blocks.y > 1 ? 2 : 1);
        p.setup = (uint16_t)(dims << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
        p.workgroup_size_x = (uint16_t)threads.x; p.workgroup_size_y = (uint16_t)threads.y; p.workgroup_size_z = (uint16_t)threads.z;
        p.grid_size_x = blocks.x * threads.x; p.grid_size_y = blocks.y * threads.y; p.grid_size_z = blocks.z * threads.z;
        p.private_segment_size = k.private_size;
        p.group_segment_size = k.group_size + dynamic_lds;
        p.kernel_object = k.object;
        p.kernarg_address = const_cast<void*>(kernarg);
        return p;
    }

    // Fill the launch-owned implicit arguments behind `explicit_bytes` of explicit ones in a host image of the kernarg
    // segment (`image` holds k.kernarg_size bytes, zeroed by the caller).
    static void fill_implicit(uint8_t* image, const Kernel& k, size_t explicit_bytes, dim3 blocks, dim3 threads) {
        const size_t off = (explicit_bytes + 7) & ~(size_t)7;
        if (off >= k.kernarg_size) return;                            // the kernel was compiled without implicit arguments
        ImplicitHead h;
        memset(&h, 0, sizeof(h));
        h.block_count[0] = blocks.x; h.block_count[1] = blocks.y; h.block_count[2] = blocks.z;
        h.group_size[0] = (uint16_t)threads.x; h.group_size[1] = (uint16_t)threads.y; h.group_size[2] = (uint16_t)threads.z;
        h.grid_dims = (uint16_t)(blocks.z > 1 ? 3 : (blocks.y > 1 ? 2 : 1));
        const size_t room = k.kernarg_size - off;                     // the compiler keeps the block only up to the last one used
        memcpy(image + off, &h, room < sizeof(h) ? room : sizeof(h));
    }

    // Append n packets (headers[i] goes with pk[i]); the last one signals `done` when `signal_last`.  Blocks while the
    // ring is full.  One producer thread per queue.
    void submit(const hsa_kernel_dispatch_packet_t* pk, const uint16_t* headers, int n, bool signal_last) {
        const uint64_t first = hsa_queue_add_write_index_relaxed(q_, (uint64_t)n);
        while (first + (uint64_t)n - hsa_queue_load_read_index_scacquire(q_) > q_->size) { /* ring full: the GPU drains it */ }
        auto* ring = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q_->base_address);
        const uint64_t mask = q_->size - 1;
        for (int i = 0; i < n; ++i) {
            hsa_kernel_dispatch_packet_t* slot = ring + ((first + (uint64_t)i) & mask);
            hsa_kernel_dispatch_packet_t p = pk[i];
            if (signal_last && i == n - 1) p.completion_signal = done_;
            // body first, the header (with the setup word: one 32-bit store) last and with release order
            memcpy(reinterpret_cast<uint8_t*>(slot) + 4, reinterpret_cast<const uint8_t*>(&p) + 4, sizeof(p) - 4);
            const uint32_t head = (uint32_t)headers[i] | ((uint32_t)p.setup << 16);
            __atomic_store_n(reinterpret_cast<uint32_t*>(slot), head, __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(q_->doorbell_signal, (hsa_signal_value_t)(first + (uint64_t)n - 1));
    }

    void arm() { hsa_signal_store_relaxed(done_, 1); }
    // Wait for the packet submitted with signal_last; false on timeout (nanoseconds of host time).
    bool wait(uint64_t timeout_ns = UINT64_MAX) {
        // timeout hint is in ticks of the system clock (HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY)
        uint64_t freq = 0;
        hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
        const uint64_t ticks = timeout_ns == UINT64_MAX ? UINT64_MAX : (uint64_t)((double)timeout_ns * 1e-9 * (double)freq);
        return hsa_signal_wait_scacquire(done_, HSA_SIGNAL_CONDITION_LT, 1, ticks, HSA_WAIT_STATE_BLOCKED) < 1;
    }

  private:
    struct Find { unsigned domain, bdf; hsa_agent_t agent; bool found; };
    struct Sym { Queue* self; std::string name; Kernel* out; bool found; };

    static hsa_status_t find_agent(hsa_agent_t a, void* data) {
        Find* w = (Find*)data;
        hsa_device_type_t type;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        uint32_t bdf = 0, dom = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
        if (bdf == w->bdf && dom == w->domain) { w->agent = a; w->found = true; return HSA_STATUS_INFO_BREAK; }
        return HSA_STATUS_SUCCESS;
    }
    static hsa_status_t find_symbol(hsa_executable_t exec, void* data) {
        Sym* s = (Sym*)data;
        hsa_executable_symbol_t sym;
        if (hsa_executable_get_symbol_by_name(exec, s->name.c_str(), &s->self->agent_, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &s->out->object);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &s->out->kernarg_size);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &s->out->group_size);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &s->out->private_size);
        s->found = true;
        return HSA_STATUS_INFO_BREAK;
    }
    // the packet processor rejected a packet / a wave faulted: the queue is dead from here on (reported by the next wait)
    static void queue_error(hsa_status_t status, hsa_queue_t*, void* data) {
        const char* msg = nullptr;
        hsa_status_string(status, &msg);
        Queue* self = (Queue*)data;
        self->error = std::string("queue error: ") + (msg ? msg : "unknown");
        self->dead.store(true);
        fprintf(stderr, "jb_aql: %s\n", self->error.c_str());
    }
    bool fail(const std::string& what) { error = what; return false; }

    bool inited_ = false;
    hsa_agent_t agent_{0};
    hsa_queue_t* q_ = nullptr;
    hsa_signal_t done_{0};
    hsa_ven_amd_loader_1_03_pfn_t loader_{};
};

}  // namespace jb_aql
