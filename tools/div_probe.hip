// div_probe.hip - is the short division of k_traverse's node step (ptx_device.hip: split_quotient) the IEEE quotient?
//
//   r  = RN(1/d)         v_rcp_f32 (1 ulp) + one Newton step
//   q0 = RN(n * r)
//   t  = fma(fma(-d, q0, n), r, q0)
//
// against the correctly rounded n / d (v_div_scale / v_div_fmas / v_div_fixup) for
//   (1) EVERY pair of significands: n = 1.m_n, d = 1.m_d, 2^23 x 2^23 pairs (both binades of the quotient; rounding
//       does not depend on the exponents while nothing overflows or underflows, which the kernel's range guard sees to);
//   (2) 2^32 random pairs with random signs and exponents inside the guard's window (|n| in [2^-100, 2^61], |d| in
//       [2^-40, 2^10]: the bounds k_traverse derives from its per-ray check, short_div_ok), also n = 0.
// Usage: div_probe [log2 of the significands of n per launch, default 15]   prints the mismatch counts; exit status 1 if any.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o tools/bin/div_probe tools/div_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ float short_div(float n, float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    const float q0 = n * r;
    const float rem = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(rem, r, q0);
}

// one thread per significand of d; the launch covers significands [n_first, n_first + n_count) of n
__global__ __launch_bounds__(256) void k_pairs(uint32_t n_first, uint32_t n_count, unsigned long long* bad, uint32_t* first_bad) {
    const uint32_t md = blockIdx.x * 256u + threadIdx.x;  // < 2^23
    const float d = __uint_as_float(0x3F800000u | md);
    uint32_t wrong = 0;
    for (uint32_t k = 0; k < n_count; ++k) {
        const float n = __uint_as_float(0x3F800000u | (n_first + k));
        const float want = n / d;
        const float got = short_div(n, d);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (!wrong && atomicAdd(&first_bad[0], 1u) == 0u) first_bad[1] = __float_as_uint(n), first_bad[2] = __float_as_uint(d);
            ++wrong;
        }
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

__device__ inline uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_random(uint64_t seed, uint32_t per_thread, unsigned long long* bad, uint32_t* first_bad) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t wrong = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        const uint64_t a = mix(seed + tid * per_thread + k), b = mix(a);
        // exponent fields: n in [127 - 100, 127 + 61), d in [127 - 40, 127 + 10)
        const uint32_t en = 27u + (uint32_t)((a >> 23) % 161u), ed = 87u + (uint32_t)((b >> 23) % 50u);
        uint32_t nb = ((uint32_t)a & 0x807FFFFFu) | (en << 23);
        const uint32_t db = ((uint32_t)b & 0x807FFFFFu) | (ed << 23);
        if (((a >> 40) & 255u) == 0u) nb &= 0x80000000u;  // a signed zero now and then
        const float n = __uint_as_float(nb), d = __uint_as_float(db);
        const float want = n / d, got = short_div(n, d);
        // (the sign of a zero quotient is not compared: the node step only ever compares t)
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want == 0.0f && got == 0.0f);
        if (!same) {
            if (!wrong && atomicAdd(&first_bad[0], 1u) == 0u) first_bad[1] = nb, first_bad[2] = db;
            ++wrong;
        }
    }
    if (wrong) atomicAdd(bad, (unsigned long long)wrong);
}

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    const uint32_t log_chunk = argc > 1 ? (uint32_t)atoi(argv[1]) : 15u;
    if (log_chunk > 20u) {
        fprintf(stderr, "chunk too large\n");
        return 2;
    }
    unsigned long long* bad = nullptr;
    uint32_t* first_bad = nullptr;
    CHECK(hipMalloc(&bad, 8));
    CHECK(hipMalloc(&first_bad, 12));
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipMemset(first_bad, 0, 12));
    const uint32_t chunk = 1u << log_chunk, n_chunks = (1u << 23) / chunk;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        k_pairs<<<(1u << 23) / 256u, 256>>>(c * chunk, chunk, bad, first_bad);
        CHECK(hipGetLastError());
        if ((c & 15u) == 15u || c + 1u == n_chunks) {
            CHECK(hipDeviceSynchronize());
            unsigned long long b = 0;
            CHECK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
            printf("pairs: %u of %u launches done, mismatches so far %llu\n", c + 1u, n_chunks, b);
            fflush(stdout);
        }
    }
    unsigned long long bad_pairs = 0;
    CHECK(hipMemcpy(&bad_pairs, bad, 8, hipMemcpyDeviceToHost));
    uint32_t fb[3] = {0, 0, 0};
    CHECK(hipMemcpy(fb, first_bad, 12, hipMemcpyDeviceToHost));
    printf("significand pairs 2^46: %llu mismatches", bad_pairs);
    if (bad_pairs) printf(" (first: n = 0x%08x, d = 0x%08x)", fb[1], fb[2]);
    printf("\n");
    CHECK(hipMemset(bad, 0, 8));
    CHECK(hipMemset(first_bad, 0, 12));
    k_random<<<1u << 14, 256>>>(0x5EED0003ull, 1u << 10, bad, first_bad);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    unsigned long long bad_random = 0;
    CHECK(hipMemcpy(&bad_random, bad, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(fb, first_bad, 12, hipMemcpyDeviceToHost));
    printf("random pairs 2^32 inside the guard's window: %llu mismatches", bad_random);
    if (bad_random) printf(" (first: n = 0x%08x, d = 0x%08x)", fb[1], fb[2]);
    printf("\n");
    return bad_pairs || bad_random ? 1 : 0;
}
