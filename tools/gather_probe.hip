// Micro-benchmark: what does one divergent vector load cost on gfx950, by width and by number of active lanes?
// Every lane gathers from pseudo-random 16-byte-aligned places of a table (3 MB: sits in one XCD's L2, as the
// device kd-tree does; or 16 KB: sits in the L1), 8 waves per SIMD, loads independent of each other, so the figure is
// the throughput of the address/data path (TA + TCP), not latency.  Output: wave-level loads per CU per microsecond
// and the cycles one CU spends per wave-level load.  Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int WORDS>
struct Vec;
template <> struct Vec<1> { typedef uint32_t T; };
template <> struct Vec<2> { typedef uint2 T; };
template <> struct Vec<3> { typedef uint3 T; };   // 12 bytes: global_load_dwordx3
template <> struct Vec<4> { typedef uint4 T; };

__device__ inline uint32_t fold(uint32_t v) { return v; }
__device__ inline uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ inline uint32_t fold(uint3 v) { return v.x ^ v.y ^ v.z; }
__device__ inline uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int WORDS>
__global__ __launch_bounds__(512) void k_gather(const uint4* table, uint32_t mask, uint32_t iters, uint32_t active, uint32_t* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = gid * 2654435761u + 12345u, acc = 0;
    if ((threadIdx.x & 63u) < active) {
        for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x = x * 1664525u + 1013904223u;
                const typename Vec<WORDS>::T v = *reinterpret_cast<const typename Vec<WORDS>::T*>(table + ((x >> 8) & mask));
                acc ^= fold(v);
            }
        }
    }
    if (acc == 0x12345678u) out[gid] = acc;  // never (keeps the loads)
}

template <int WORDS>
static double run(const uint4* table, uint32_t mask, uint32_t active, uint32_t* out, int n_cu) {
    const uint32_t iters = 256;
    const int blocks = n_cu * 4;  // 4 x 512 threads = 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    k_gather<WORDS><<<blocks, 512>>>(table, mask, 8, active, out);
    hipEventRecord(a);
    k_gather<WORDS><<<blocks, 512>>>(table, mask, iters, active, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_loads_per_cu = 32.0 * iters * 8;  // 32 waves per CU
    return ms * 1e-3 / wave_loads_per_cu;               // seconds of a CU per wave-level load
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 2;
    const int n_cu = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("# %s, %d CUs, %.2f GHz\n", p.name, n_cu, ghz);
    uint4* table;
    uint32_t* out;
    const size_t entries = 1u << 18;  // 4 MB allocated
    if (hipMalloc(&table, entries * 16) != hipSuccess || hipMalloc(&out, (size_t)n_cu * 4 * 512 * 4) != hipSuccess) return 2;
    hipMemset(table, 1, entries * 16);
    printf("%-8s %-7s %-7s %12s\n", "table", "bytes", "lanes", "cycles/load");
    for (uint32_t size_log2 : {10u, 17u}) {  // 16 KB, 2 MB
        const uint32_t mask = (1u << size_log2) - 1u;
        for (uint32_t active : {64u, 32u, 16u, 8u}) {
            const double c1 = run<1>(table, mask, active, out, n_cu) * ghz * 1e9;
            const double c2 = run<2>(table, mask, active, out, n_cu) * ghz * 1e9;
            const double c3 = run<3>(table, mask, active, out, n_cu) * ghz * 1e9;
            const double c4 = run<4>(table, mask, active, out, n_cu) * ghz * 1e9;
            const char* name = size_log2 == 10 ? "16KB" : "2MB";
            printf("%-8s %-7d %-7u %12.1f\n%-8s %-7d %-7u %12.1f\n%-8s %-7d %-7u %12.1f\n%-8s %-7d %-7u %12.1f\n", name, 4, active, c1, name, 8,
                   active, c2, name, 12, active, c3, name, 16, active, c4);
        }
    }
    return 0;
}
