// atomic_probe.hip - how fast do device-scope atomicAdds complete when 8192 waves hammer K counters `stride` bytes apart?
// (k_traverse's work cursor: one counter completes one atomic per ~16 ns.)  One lane per wave issues `per_wave` dependent atomics,
// wave w on counter w % K.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/atomic_probe tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(1024) void k_hammer(uint32_t* base, uint32_t k, uint32_t stride_words, uint32_t per_wave, uint32_t* sink) {
    const uint32_t wave = blockIdx.x * 16u + (threadIdx.x >> 6);
    uint32_t* const c = base + (size_t)(wave % k) * stride_words;
    uint32_t acc = 0;
    if ((threadIdx.x & 63u) == 0u)
        for (uint32_t i = 0; i < per_wave; ++i) acc += atomicAdd(c, 1u);
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

int main() {
    uint32_t* buf = nullptr;
    const size_t bytes = 64u << 20;
    if (hipMalloc(&buf, bytes) != hipSuccess) return 2;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const uint32_t per_wave = 8;
    const uint32_t ks[] = {1, 2, 4, 8, 16};
    const uint32_t strides[] = {4, 64, 128, 256, 4096, 65536, 1u << 20};
    for (uint32_t k : ks)
        for (uint32_t stride : strides) {
            if (k == 1 && stride != 4) continue;
            if ((size_t)k * stride > bytes) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(buf, 0, bytes);
                hipDeviceSynchronize();
                hipEventRecord(a);
                k_hammer<<<512, 1024>>>(buf, k, stride / 4u, per_wave, buf + (bytes / 4 - 1));
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("%2u counters %8u bytes apart: %8.1f us for %u atomics = %6.2f ns per atomic\n", k, stride, best * 1e3f, 8192u * per_wave, best * 1e6f / (8192.0f * per_wave));
        }
    return 0;
}
