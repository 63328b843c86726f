// Device allocations of libquflow_hip.so, with optional guard zones (QUFLOW_HIP_DEBUG_GUARD=1).
//
// The HIP allocator hands out memory at a 2 MiB granularity (4 KiB for small requests): a kernel that stores a few rows past
// the end of its operand damages a neighbour or nothing at all, and faults only when it leaves the mapping -- which, on a
// shared node, can cost everybody their GPU.  Under the variable every allocation of the library is framed by two GUARD-byte
// zones holding a byte pattern; the zones are read back when the allocation is released and whenever
// qf_debug_guard_check() is called, so the suite and the size sweeps can be run once with every buffer fenced
// (tools/gpu/r6_guarded.sh; tests/conftest.py fails the test after which a zone is found damaged).  This file deliberately does not
// include qf_internal.h: that header routes `hipMalloc` / `hipFree` of every other translation unit to the two functions
// defined here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/quflow_hip.h"

void qf_set_error(const char *fmt, ...);

namespace {

// zone size: 64 KiB unless QUFLOW_HIP_DEBUG_GUARD_KB says otherwise (a multiple of 4; read once).  The user pointer sits one
// zone behind the allocator's (2 MiB-aligned) base: the size also decides how the library's buffers are aligned.
const size_t GUARD = [] {
    const char *e = getenv("QUFLOW_HIP_DEBUG_GUARD_KB");
    const long kb = e ? atol(e) : 64;
    return (size_t)(kb >= 4 && kb % 4 == 0 && kb <= (1 << 20) ? kb : 64) * 1024;
}();
constexpr unsigned char PATTERN = 0xA5;

struct guarded {
    unsigned char *base;
    size_t bytes;
    int device;
};

std::mutex g_mu;
std::map<void *, guarded> g_live;        // user pointer -> allocation
long long g_damaged = 0;                 // zones found damaged so far (live or since released)
long long g_allocations = 0;
char g_first[256] = "";

bool enabled()
{
    static const bool on = [] {
        const char *e = getenv("QUFLOW_HIP_DEBUG_GUARD");
        return e && e[0] && e[0] != '0';
    }();
    return on;
}

// The stream the pattern fills run on.  NOT the NULL stream, and not one stream: the runtime gives every stream that is used
// the next hardware queue, a queue's number mod 4 is the pipe it sits on, and two replicas whose queues share a pipe run at
// 11,000 instead of 18,000 timesteps/s together (N = 512, four replicas per GPU: profiles/r06_x4_hardware_queues.txt -- found
// when the first version of this file filled on the NULL stream, whose queue then appeared between the queues of an
// ensemble's members).  The guard therefore takes its queues in a block of FOUR at its first use on a device (three of them
// idle for ever): whatever streams the library creates before and after keep the pipes they would have had without it.
hipStream_t fill_stream(int device)
{
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t first = nullptr;
    unsigned char *scratch = nullptr;
    if (hipMalloc((void **)&scratch, 4096) != hipSuccess) return nullptr;
    for (int q = 0; q < 4; ++q) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        (void)hipMemsetAsync(scratch, 0, 4096, st);      // (a stream gets its hardware queue when it is first used)
        (void)hipStreamSynchronize(st);
        if (!first) first = st;
    }
    (void)hipFree(scratch);
    streams[device] = first;
    return first;
}

// reads one zone back and notes the first byte that no longer holds the pattern (g_mu is held)
void look(const guarded &g, const void *user, bool upper)
{
    static thread_local std::vector<unsigned char> host(GUARD);
    const unsigned char *zone = upper ? g.base + GUARD + g.bytes : g.base;
    // (the zone may start at any byte -- bytes is whatever the caller asked for; a copy has no alignment rule)
    hipStream_t fs = fill_stream(g.device);
    if (!fs || hipMemcpyAsync(host.data(), zone, GUARD, hipMemcpyDeviceToHost, fs) != hipSuccess || hipStreamSynchronize(fs) != hipSuccess) return;
    for (size_t i = 0; i < GUARD; ++i)
        if (host[i] != PATTERN) {
            ++g_damaged;
            const long long off = upper ? (long long)(g.bytes + i) : (long long)i - (long long)GUARD;
            if (!g_first[0])
                snprintf(g_first, sizeof g_first, "allocation of %zu bytes at %p on device %d: byte at offset %lld (%s the buffer) overwritten",
                         g.bytes, user, g.device, off, upper ? "past the end of" : "in front of");
            fprintf(stderr, "quflow_hip guard: allocation of %zu bytes at %p (device %d): offset %lld overwritten (0x%02x)\n", g.bytes, user,
                    g.device, off, host[i]);
            // restore the zone so that one stray store is reported once
            if (hipStream_t fs = fill_stream(g.device)) {
                (void)hipMemsetAsync((void *)zone, PATTERN, GUARD, fs);
                (void)hipStreamSynchronize(fs);
            }
            return;
        }
}

}  // namespace

hipError_t qf_guard_malloc(void **p, size_t bytes)
{
    if (!enabled()) return hipMalloc(p, bytes);
    {
        // QUFLOW_HIP_DEBUG_GUARD_LIMIT_MB: a device that is "full" at that many MiB of live library allocations -- the
        // out-of-memory path (error text, clean-up of a half-built context, the Python layer closing idle cached contexts
        // and trying again) exercised without exhausting 288 GB (tests/test_hip_envelope.py)
        static const long long limit = [] {
            const char *e = getenv("QUFLOW_HIP_DEBUG_GUARD_LIMIT_MB");
            return e ? atoll(e) * (1ll << 20) : -1ll;
        }();
        if (limit >= 0) {
            std::lock_guard<std::mutex> lk(g_mu);
            long long live = 0;
            for (auto &kv : g_live) live += (long long)kv.second.bytes;
            if (live + (long long)bytes > limit) return hipErrorOutOfMemory;
        }
    }
    unsigned char *base = nullptr;
    hipError_t e = hipMalloc((void **)&base, bytes + 2 * GUARD);
    if (e != hipSuccess) return e;
    int dev = -1;
    (void)hipGetDevice(&dev);
    hipStream_t fs = fill_stream(dev);
    if (!fs) {
        (void)hipFree(base);
        return hipErrorUnknown;
    }
    // (both fills complete before the call returns to a host that is about to use the buffer)
    if ((e = hipMemsetAsync(base, PATTERN, GUARD, fs)) != hipSuccess || (e = hipMemsetAsync(base + GUARD + bytes, PATTERN, GUARD, fs)) != hipSuccess ||
        (e = hipStreamSynchronize(fs)) != hipSuccess) {
        (void)hipFree(base);
        return e;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[base + GUARD] = guarded{base, bytes, dev};
    ++g_allocations;
    *p = base + GUARD;
    return hipSuccess;
}

hipError_t qf_guard_free(void *p)
{
    if (!enabled() || !p) return hipFree(p);
    guarded g{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) return hipFree(p);      // (allocated before the variable could matter: never, but harmless)
        g = it->second;
        (void)hipDeviceSynchronize();
        look(g, p, false);
        look(g, p, true);
        g_live.erase(it);
    }
    return hipFree(g.base);
}

extern "C" int qf_debug_guard_check(long long *allocations, long long *damaged, char *first, int n)
{
    if (allocations) *allocations = 0;
    if (damaged) *damaged = 0;
    if (first && n > 0) first[0] = 0;
    if (!enabled()) return QF_OK;
    {
        // QUFLOW_HIP_DEBUG_GUARD=selftest: the first check damages one byte on either side of a scratch allocation of its
        // own, so that a test can see the zones report (tests/test_hip_envelope.py::test_guard_zones_report_a_stray_store)
        static std::once_flag once;
        const char *e = getenv("QUFLOW_HIP_DEBUG_GUARD");
        if (e && e[0] == 's')
            std::call_once(once, [] {
                unsigned char *q = nullptr;
                if (qf_guard_malloc((void **)&q, 1000) != hipSuccess) return;
                int dev = 0;
                (void)hipGetDevice(&dev);
                if (hipStream_t fs = fill_stream(dev)) {
                    (void)hipMemsetAsync(q + 1000, 0, 1, fs);
                    (void)hipMemsetAsync(q - 1, 0, 1, fs);
                    (void)hipStreamSynchronize(fs);
                }
                (void)qf_guard_free(q);
            });
    }
    std::lock_guard<std::mutex> lk(g_mu);
    int dev0 = -1;
    (void)hipGetDevice(&dev0);
    for (auto &kv : g_live) {
        (void)hipSetDevice(kv.second.device);
        (void)hipDeviceSynchronize();
        look(kv.second, kv.first, false);
        look(kv.second, kv.first, true);
    }
    if (dev0 >= 0) (void)hipSetDevice(dev0);
    if (allocations) *allocations = g_allocations;
    if (damaged) *damaged = g_damaged;
    if (first && n > 0) snprintf(first, (size_t)n, "%s", g_first);
    return QF_OK;
}
