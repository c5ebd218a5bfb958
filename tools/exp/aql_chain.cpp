// Is the ~7 us per-replay hole of hipGraphLaunch a property of the device or of the HIP graph path?  The same dependent chain
// of 15 short kernels per "step" is submitted (a) as a captured hipGraph, replayed K times, (b) as eager hipModuleLaunchKernel
// calls, (c) as raw AQL packets (barrier bit, agent-scope fences) on a private HSA queue -- the host only writes 64-byte
// packets and rings the doorbell.  Prints microseconds per step and checks the chain's counter.
// build: hipcc --offload-arch=gfx950 --offload-device-only --no-gpu-bundle-output -O3 -o aql_chain_kernels.hsaco aql_chain_kernels.hip  (a raw ELF: HSA does not take bundles)
//        hipcc -O2 -o aql_chain.bin aql_chain.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t e = (x); if (e != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(e, &m); printf("%s: %s\n", #x, m); exit(1); } } while (0)

static hsa_agent_t g_gpu;
static bool g_have = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
    return HSA_STATUS_SUCCESS;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Args { unsigned long long* counter; unsigned* scratch; int ticks; };

int main(int argc, char** argv)
{
    const int links = 15, steps = argc > 1 ? atoi(argv[1]) : 2000, blocks = 256;
    const std::string dir = argc > 2 ? argv[2] : ".";
    std::vector<char> co;
    {
        FILE* f = fopen((dir + "/aql_chain_kernels.hsaco").c_str(), "rb");
        if (!f) { printf("no hsaco\n"); return 1; }
        fseek(f, 0, SEEK_END); co.resize(ftell(f)); fseek(f, 0, SEEK_SET);
        if (fread(co.data(), 1, co.size(), f) != co.size()) return 1;
        fclose(f);
    }
    CK(hipSetDevice(0));
    unsigned long long* counter; unsigned* scratch;
    CK(hipMalloc(&counter, 8)); CK(hipMalloc(&scratch, blocks * 64 * 4));
    CK(hipMemset(counter, 0, 8)); CK(hipMemset(scratch, 1, blocks * 64 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipModule_t mod; hipFunction_t fn;
    CK(hipModuleLoadData(&mod, co.data()));
    CK(hipModuleGetFunction(&fn, mod, "chain_link"));

    for (int ticks : {100, 200, 400}) {          // 1, 2, 4 us of work per link
        Args a{counter, scratch, ticks};
        size_t asz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        // (a) hipGraph
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < links; l++) CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
        CK(hipStreamEndCapture(st, &gr));
        CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemset(counter, 0, 8));
        double t0 = now_us();
        for (int i = 0; i < steps; i++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        double t_graph = (now_us() - t0) / steps;
        unsigned long long c_graph; CK(hipMemcpy(&c_graph, counter, 8, hipMemcpyDeviceToHost));
        // (b) eager
        CK(hipMemset(counter, 0, 8));
        t0 = now_us();
        for (int i = 0; i < steps; i++)
            for (int l = 0; l < links; l++) CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
        CK(hipStreamSynchronize(st));
        double t_eager = (now_us() - t0) / steps;
        unsigned long long c_eager; CK(hipMemcpy(&c_eager, counter, 8, hipMemcpyDeviceToHost));
        printf("work/link %.0f us: hipGraph %.2f us/step (counter %llu of %d), eager %.2f us/step (counter %llu)\n", ticks / 100.0, t_graph,
               c_graph, steps * links, t_eager, c_eager);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
    }

    // (c) raw AQL
    HK(hsa_init());
    HK(hsa_iterate_agents(agent_cb, nullptr));
    if (!g_have) { printf("no gpu agent\n"); return 1; }
    hsa_queue_t* q;
    HK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_code_object_reader_t rd; hsa_executable_t exe; hsa_executable_symbol_t sym;
    HK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &rd));
    HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    HK(hsa_executable_get_symbol_by_name(exe, "chain_link.kd", &g_gpu, &sym));
    uint64_t kobj; uint32_t kasz, gsz, psz;
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kasz));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gsz));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &psz));
    printf("AQL: kernel object %llx kernarg %u B group %u private %u, queue size %u\n", (unsigned long long)kobj, kasz, gsz, psz, q->size);
    hsa_signal_t done; HK(hsa_signal_create(1, 0, nullptr, &done));
    char* kargs; CK(hipMalloc(&kargs, 4096));          // kernarg block in device memory (zero-filled hidden arguments)
    for (int scope : {HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_SYSTEM})
    for (int ticks : {100, 200, 400}) {
        std::vector<char> host(kasz > 256 ? kasz : 256, 0);
        Args a{counter, scratch, ticks};
        memcpy(host.data(), &a, sizeof(a));
        // code-object-v5 hidden arguments right after the explicit ones (8-byte aligned): block counts (3 x u32), group sizes (3 x u16)
        size_t off = (sizeof(a) + 7) & ~7ull;
        uint32_t bc[3] = {(uint32_t)blocks, 1, 1}; uint16_t gs[3] = {256, 1, 1};
        memcpy(host.data() + off, bc, 12); memcpy(host.data() + off + 12, gs, 6);
        CK(hipMemcpy(kargs, host.data(), host.size(), hipMemcpyHostToDevice));
        CK(hipMemset(counter, 0, 8));
        CK(hipDeviceSynchronize());
        hsa_signal_store_relaxed(done, 1);
        const uint32_t mask = q->size - 1;
        hsa_kernel_dispatch_packet_t* base = (hsa_kernel_dispatch_packet_t*)q->base_address;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER)
                                | (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        const uint16_t header_last = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER)
                                     | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        const long total = (long)steps * links;
        double t0 = now_us();
        for (long i = 0; i < total; i++) {
            uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}      // ring full: wait for the packet processor
            hsa_kernel_dispatch_packet_t* p = base + (idx & mask);
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
            p->grid_size_x = blocks * 256; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = psz; p->group_segment_size = gsz;
            p->kernel_object = kobj; p->kernarg_address = kargs; p->reserved2 = 0;
            p->completion_signal.handle = (i == total - 1) ? done.handle : 0;
            __atomic_store_n(&p->header, (i == total - 1) ? header_last : header, __ATOMIC_RELEASE);
            if ((i % links) == links - 1 || i == total - 1) hsa_signal_store_screlease(q->doorbell_signal, idx);       // one doorbell per step
        }
        double t_sub = (now_us() - t0) / steps;
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
        double t_aql = (now_us() - t0) / steps;
        unsigned long long c; CK(hipMemcpy(&c, counter, 8, hipMemcpyDeviceToHost));
        printf("work/link %.0f us, fence scope %s: AQL %.2f us/step = %.2f us per link boundary (counter %llu of %ld)\n", ticks / 100.0,
               scope == HSA_FENCE_SCOPE_NONE ? "none" : scope == HSA_FENCE_SCOPE_AGENT ? "agent" : "system", t_aql, t_aql / links - ticks / 100.0, c, total);
    }
    hsa_queue_destroy(q);
    return 0;
}
