// Every enqueue of a step goes through a Launcher, so that the SAME host code can run in three ways:
//   normal      kernels / memsets / copies / event records are enqueued on the stream;
//   signature   (dry) nothing is enqueued; every operation and every argument byte is hashed.  Two steps with the same signature
//               enqueue bit-identical work: same kernels, grids, pointers, sizes, scalars;
//   capture     as normal, but the stream is being captured into a HIP graph (world.hip), which later steps with the same
//               signature replay with ONE hipGraphLaunch instead of ~35 launches — the small scenes are bound by the host's launch
//               rate (~5 us per launch against kernels of 2-5 us), not by the device.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>
#include <algorithm>

namespace mi {

// memset as a kernel of our own: a memset NODE of a captured graph raced with the kernels behind it under the HIP 7.0 runtime
// (history table cleared late -> different colours -> different poses, step 123 of the ragdoll scene); a kernel node is ordered like any other
__global__ __launch_bounds__(256) void k_fill(unsigned char* __restrict__ p, uint32_t value, size_t bytes) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const uint32_t b = value & 0xFFu, w = b | (b << 8) | (b << 16) | (b << 24);
    if (((uintptr_t)p & 15u) == 0u) {
        uint4* q = reinterpret_cast<uint4*>(p);
        const size_t n16 = bytes >> 4;
        for (size_t i = t; i < n16; i += nt) q[i] = make_uint4(w, w, w, w);
        for (size_t i = (n16 << 4) + t; i < bytes; i += nt) p[i] = (unsigned char)b;
    } else {
        for (size_t i = t; i < bytes; i += nt) p[i] = (unsigned char)b;
    }
}

struct Launcher;
// how an argument enters the signature: raw bytes by default; structs with padding bytes get an overload that mixes their fields (world.hip)
template <class T> inline void sigMix(Launcher& L, const T& v);

struct Launcher {
    bool dry = false;        // signature pass: hash only
    bool hashing = false;    // maintain the signature (off: plain launches, no hashing cost)
    bool trace = false;      // development: keep the per-operation hashes (MI_GRAPH_DEBUG)
    uint64_t h = 1469598103934665603ull;
    uint32_t ops = 0;
    std::vector<uint64_t> opHashes;
    hipError_t firstError = hipSuccess;

    void begin(bool dryRun, bool hash) { dry = dryRun; hashing = hash; h = 1469598103934665603ull; ops = 0; firstError = hipSuccess; if (trace) opHashes.clear(); }
    void bytes(const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        uint64_t x = h;
        for (size_t i = 0; i < n; ++i) { x ^= b[i]; x *= 1099511628211ull; }
        h = x;
    }
    template <class T> void pod(const T& v) { bytes(&v, sizeof(T)); }
    void opDone() { ++ops; if (trace) opHashes.push_back(h); }
    void note(hipError_t e) { if (e != hipSuccess && firstError == hipSuccess) firstError = e; }

    template <class... KArgs, class... Args>
    void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
        std::tuple<std::decay_t<KArgs>...> t(static_cast<std::decay_t<KArgs>>(std::forward<Args>(args))...);
        if (hashing) {
            const void* kp = reinterpret_cast<const void*>(kernel);
            pod(kp); pod(grid.x); pod(grid.y); pod(grid.z); pod(block.x); pod(block.y); pod(block.z); pod(lds);
            std::apply([&](const auto&... a) { (sigMix(*this, a), ...); }, t);
            opDone();
        }
        if (!dry) {
            if (traceEach) { std::fprintf(stderr, "[mi_physics]   launch %u: grid %u x %u, block %u, lds %zu\n", traceOrdinal++, grid.x, grid.y, block.x, lds); std::fflush(stderr); }
            std::apply([&](auto&... a) { hipLaunchKernelGGL(kernel, grid, block, lds, st, a...); }, t);
            note(hipGetLastError());
            if (traceEach) note(hipStreamSynchronize(st));   // (MI_DEBUG_SYNC: a memory fault ends the process inside this wait — the line above names the launch)
        }
    }
    bool traceEach = false; uint32_t traceOrdinal = 0;
    hipError_t memsetAsync(void* p, int v, size_t n, hipStream_t st) {
        if (!n) return hipSuccess;
        const uint32_t blocks = (uint32_t)std::min<size_t>(2048, (n / 16 + 255) / 256 + 1);
        launch(k_fill, dim3(blocks), dim3(256), 0, st, static_cast<unsigned char*>(p), (uint32_t)v, n);
        return dry ? hipSuccess : firstError;
    }
    hipError_t memcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st) {
        if (hashing) { const int tag = 0x4D43; pod(tag); pod(dst); pod(src); pod(n); pod(kind); opDone(); }
        if (dry) return hipSuccess;
        hipError_t e = hipMemcpyAsync(dst, src, n, kind, st); note(e); return e;
    }
    void eventRecord(hipEvent_t ev, hipStream_t st) {
        if (hashing) { const int tag = 0x4556; pod(tag); pod(ev); opDone(); }
        if (!dry) note(hipEventRecord(ev, st));
    }
};

template <class T> inline void sigMix(Launcher& L, const T& v) { L.bytes(&v, sizeof(T)); }

}  // namespace mi
