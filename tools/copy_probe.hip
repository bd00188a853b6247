// copy_probe.hip - what a plain streaming kernel reaches on this GPU: the yardstick for k_shade's 4.6 TB/s.
// float4 grid-stride read-only sum, write-only fill and copy over 8 GiB, with and without the non-temporal hint.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const f32x4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ src, float* out, size_t n) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ dst, size_t n) {
    const f32x4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <class F>
static float best_ms(F&& launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(a);
        launch();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    f32x4 *src = nullptr, *dst = nullptr;
    float* out = nullptr;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 2;
    (void)hipMemset(src, 1, bytes), (void)hipMemset(dst, 0, bytes);
    for (int grid : {2048, 4096, 16384}) {
        const float r0 = best_ms([&] { k_read<false><<<grid, 256>>>(src, out, n); }), r1 = best_ms([&] { k_read<true><<<grid, 256>>>(src, out, n); });
        const float w0 = best_ms([&] { k_fill<false><<<grid, 256>>>(dst, n); }), w1 = best_ms([&] { k_fill<true><<<grid, 256>>>(dst, n); });
        const float c0 = best_ms([&] { k_copy<false><<<grid, 256>>>(src, dst, n); }), c1 = best_ms([&] { k_copy<true><<<grid, 256>>>(src, dst, n); });
        const double gb = bytes / 1e9;
        printf("grid %5d: read %.2f / nt %.2f TB/s   write %.2f / nt %.2f TB/s   copy (read + write bytes) %.2f / nt %.2f TB/s\n", grid, gb / r0, gb / r1, gb / w0,
               gb / w1, 2 * gb / c0, 2 * gb / c1);
    }
    return 0;
}
