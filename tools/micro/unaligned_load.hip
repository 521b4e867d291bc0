#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
__global__ void k(const uint8_t *src, uint32_t *out, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32x4 v = *reinterpret_cast<const u32x4_u *>(src + t * 7 + 1);
    out[t] = v.x ^ v.y ^ v.z ^ v.w;
}
int main() {
    const int n = 1 << 16;
    uint8_t *h = (uint8_t *)malloc(n * 7 + 64), *d; uint32_t *o, *ho = (uint32_t *)malloc(n * 4);
    for (int i = 0; i < n * 7 + 64; ++i) h[i] = uint8_t(i * 131 + (i >> 8));
    hipMalloc(&d, n * 7 + 64); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 7 + 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < n; ++t) { uint32_t w[4]; memcpy(w, h + t * 7 + 1, 16); if ((w[0] ^ w[1] ^ w[2] ^ w[3]) != ho[t]) ++bad; }
    printf("unaligned dwordx4: %s, %d mismatches\n", hipGetErrorString(e), bad);
    return bad != 0;
}
