// Micro-benchmark (analysis only): what a device-to-device copy of 1 GiB reaches on this chip -- the bound of k_replace_emit
// (search and replace writes the text again).  Variants: plain stores, non-temporal stores, 1 / 2 / 4 / 8 quads in flight per lane,
// grid of `waves_per_cu` x CUs workgroups striding the buffer; hipMemcpyAsync next to them.
// build: hipcc --offload-arch=gfx950 -O3 -o copy_rate copy_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int D, bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16) {
    const uint64_t stride = uint64_t(gridDim.x) * 256u * D;
    for (uint64_t i = uint64_t(blockIdx.x) * 256u * D + threadIdx.x; i < n16; i += stride) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = i + uint64_t(d) * 256u < n16 ? src[i + uint64_t(d) * 256u] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + uint64_t(d) * 256u < n16) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                if (NT) __builtin_nontemporal_store(u32x4{v[d].x, v[d].y, v[d].z, v[d].w}, reinterpret_cast<u32x4 *>(&dst[i + uint64_t(d) * 256u]));
                else dst[i + uint64_t(d) * 256u] = v[d];
            }
    }
}

// a wave copies ONE contiguous share of the buffer (k_replace_emit's shape: 1 KiB per wave-instruction, D in flight), the source
// `mis` bytes off the destination's alignment
template <int D>
__global__ __launch_bounds__(256) void k_copy_wave(const uint8_t *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16, uint32_t mis) {
    const uint64_t waves = uint64_t(gridDim.x) * 4u, w = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint64_t per = (n16 + waves - 1) / waves;
    const uint64_t lo = w * per, hi = lo + per < n16 ? lo + per : n16;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(1)));
    for (uint64_t i = lo + lane; i < hi; i += 64u * D) {
        uint4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + 64u * d < hi) { const u32x4 t = *reinterpret_cast<const u32x4 *>(src + mis + ((i + 64u * d) << 4)); v[d] = make_uint4(t.x, t.y, t.z, t.w); }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (i + 64u * d < hi) dst[i + 64u * d] = v[d];
    }
}

template <int D>
static void run_wave(const char *name, const uint4 *src, uint4 *dst, uint64_t bytes, int blocks, uint32_t mis) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t n16 = bytes / 16 - 16;   // (the misaligned source reads up to 64 + 16 bytes further)
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy_wave<D>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint8_t *>(src), dst, n16, mis);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_copy_wave<D>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint8_t *>(src), dst, n16, mis);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %6d mis %2u: %.3f ms per GiB-copy = %.0f GB/s read + written\n", name, blocks, mis, ms / reps, 2.0 * bytes / (ms / reps) / 1e6);
}

template <int D, bool NT>
static void run(const char *name, const uint4 *src, uint4 *dst, uint64_t bytes, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy<D, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_copy<D, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %6d: %.3f ms per GiB-copy = %.0f GB/s read + written\n", name, blocks, ms / reps, 2.0 * bytes / (ms / reps) / 1e6);
}

int main() {
    const uint64_t bytes = 1ull << 30;
    uint4 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 2, bytes);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int per : {4, 8, 16, 32}) {
        run<1, false>("1 quad, plain", src, dst, bytes, cus * per);
        run<4, false>("4 quads, plain", src, dst, bytes, cus * per);
        run<4, true>("4 quads, non-temporal", src, dst, bytes, cus * per);
        run<8, true>("8 quads, non-temporal", src, dst, bytes, cus * per);
    }
    for (int per : {4, 8, 16})
        for (uint32_t mis : {0u, 5u, 16u, 64u}) {
            run_wave<4>("wave shares, 4 quads", src, dst, bytes, cus * per, mis);
            run_wave<8>("wave shares, 8 quads", src, dst, bytes, cus * per, mis);
        }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s              : %.3f ms per GiB-copy = %.0f GB/s read + written\n", "hipMemcpyAsync D2D", ms / 10, 2.0 * bytes / (ms / 10) / 1e6);
    return 0;
}
