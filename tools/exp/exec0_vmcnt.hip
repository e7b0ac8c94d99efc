// Does a vector-memory instruction issued with EXEC = 0 still take part in vmcnt accounting on gfx950?
// One slow load (uncached memory), then N stores with EXEC = 0, then s_waitcnt vmcnt(N): if the masked stores count, the load must have
// landed when the wait returns; if they do not, the wait falls through at once and the destination register is still stale.
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef WAITN_STR
#define WAITN_STR "4"
#endif
__global__ void k(const float4* src, float4* sink, float4* out, int iters) {
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float v = -1.f, snap = 0.f;
        const float4* p = src + ((threadIdx.x + it * 977) & 0xFFFFF);
        unsigned long long save;
        asm volatile("global_load_dword %0, %3, off sc1\n\t"
                     "s_mov_b64 %2, exec\n\t"
                     "s_mov_b64 exec, 0\n\t"
                     "global_store_dword %4, %0, off\n\t"
                     "global_store_dword %4, %0, off\n\t"
                     "global_store_dword %4, %0, off\n\t"
                     "global_store_dword %4, %0, off\n\t"
                     "s_mov_b64 exec, %2\n\t"
                     "s_waitcnt vmcnt(" WAITN_STR ")\n\t"
                     "v_mov_b32 %1, %0"              // snapshot taken right behind the wait, inside the same asm block
                     : "+v"(v), "=&v"(snap), "=&s"(save) : "v"(p), "v"(sink + threadIdx.x) : "memory");
        if (snap != 3.f) ++bad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = make_float4((float)bad, 0, 0, 0);
}
int main() {
    float4 *src, *sink, *out; const size_t n = 1 << 20;
    hipExtMallocWithFlags((void**)&src, n * sizeof(float4), hipDeviceMallocUncached);
    hipMalloc(&sink, 4096 * sizeof(float4)); hipMalloc(&out, 64 * 256 * sizeof(float4));
    float4* h = (float4*)malloc(n * sizeof(float4)); for (size_t i = 0; i < n; ++i) h[i] = make_float4(3.f, 3.f, 3.f, 3.f);
    hipMemcpy(src, h, n * sizeof(float4), hipMemcpyHostToDevice);
    k<<<256, 64>>>(src, sink, out, 2000);
    float4* ho = (float4*)malloc(64 * 256 * sizeof(float4)); hipMemcpy(ho, out, 64 * 256 * sizeof(float4), hipMemcpyDeviceToHost);
    double bad = 0; for (int i = 0; i < 64 * 256; ++i) bad += ho[i].x;
    printf("stale reads after vmcnt(" WAITN_STR ") behind 4 EXEC=0 stores: %.0f of %d (0 = masked stores are counted)\n", bad, 64 * 256 * 2000);
    return 0;
}
