// Micro-benchmark (analysis only): what HBM read bandwidth do the scan kernels' ACCESS PATTERNS allow,
// with no compute at all?  1024-thread workgroups, `occ` workgroups' worth of LDS reserved to mimic the
// scan kernels' occupancy (1 block per CU).
//   mode 0: lane-per-haystack  -- lane streams its own 8 KiB region, 128 B (8 x dwordx4) per step, D lines in flight
//   mode 1: wave-per-haystack  -- lane l reads line l of the wave's current 8 KiB region (contiguous per wave)
//   mode 2: wave-per-haystack, coalesced -- lane l reads quads r*64 + l; a wave owns 64 CONSECUTIVE items (k_scan_anchor)
//   mode 3: as 2, but the wave's items are strided by the number of waves (all waves sweep one moving window)
//   mode 4: as 2, persistent: 256 workgroups, waves pull batches of 64 items from a counter
// build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int D>
__global__ __launch_bounds__(1024) void k(const uint4 *src, uint64_t n_items, uint32_t *sink) {
    extern __shared__ uint8_t smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    uint32_t acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    if (MODE == 0) {
        const uint64_t item = uint64_t(blockIdx.x) * 1024 + threadIdx.x;
        if (item >= n_items) return;
        const uint4 *p = src + item * 512;  // 8 KiB = 512 uint4
        uint4 buf[D][8];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int r = 0; r < 8; ++r) buf[d][r] = p[d * 8 + r];
        for (uint32_t line = 0; line < 64; line += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint4 v = buf[d][r];
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                    const uint32_t nl = line + D + d;
                    buf[d][r] = p[(nl < 64 ? nl : 63) * 8 + r];
                }
            }
        }
    } else if (MODE == 2 || MODE == 3) {
        const uint64_t wave_id = uint64_t(blockIdx.x) * 16 + (threadIdx.x >> 6);
        const uint64_t n_waves = uint64_t(gridDim.x) * 16;
        const uint64_t first = MODE == 2 ? wave_id * 64 : wave_id, step = MODE == 2 ? 1 : n_waves;
        if (first >= n_items) return;
        uint4 buf[D][8];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int r = 0; r < 8; ++r) buf[d][r] = src[(first + d * step) * 512 + r * 64 + lane];
        for (uint32_t it = 0; it < 64; it += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint4 v = buf[d][r];
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                    const uint32_t ni = it + D + d;
                    buf[d][r] = src[(first + (ni < 64 ? ni : 63) * step) * 512 + r * 64 + lane];
                }
            }
        }
    } else {
        const uint64_t wave_first = (uint64_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 64;
        if (wave_first >= n_items) return;
        uint4 buf[D][8];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int r = 0; r < 8; ++r) buf[d][r] = src[(wave_first + d) * 512 + lane * 8 + r];
        for (uint32_t it = 0; it < 64; it += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const uint4 v = buf[d][r];
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                    const uint32_t ni = it + D + d;
                    buf[d][r] = src[(wave_first + (ni < 64 ? ni : 63)) * 512 + lane * 8 + r];
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int D>
void run(const uint4 *d, uint64_t n_items, uint32_t *sink, int lds) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t blocks = uint32_t((n_items + 1023) / 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, D>), dim3(blocks), dim3(1024), lds, 0, d, n_items, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d depth %d lds %6d: %.3f ms per 4 GiB = %.0f GB/s\n", MODE, D, lds, ms / 5, n_items * 8192.0 / (ms / 5) / 1e6);
}

int main() {
    const uint64_t n_items = 1 << 19;
    uint4 *d; uint32_t *sink;
    hipMalloc(&d, n_items * 8192); hipMalloc(&sink, 64);
    hipMemset(d, 1, n_items * 8192);
    for (int lds : {160 * 1024 - 64, 64 * 1024, 1024}) {
        run<0, 1>(d, n_items, sink, lds);
        run<0, 2>(d, n_items, sink, lds);
        run<1, 1>(d, n_items, sink, lds);
        run<1, 2>(d, n_items, sink, lds);
        run<2, 1>(d, n_items, sink, lds);
        run<2, 2>(d, n_items, sink, lds);
        run<3, 1>(d, n_items, sink, lds);
        run<3, 2>(d, n_items, sink, lds);
    }
    return 0;
}
