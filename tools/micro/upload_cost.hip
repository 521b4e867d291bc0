// What an upload of the build-time trie costs: which part of "hipMalloc + pageable hipMemcpy" is slow the first time.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/upload_cost.hip -o /tmp/upload_cost && /tmp/upload_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double copy_ms(void *d, const void *s, size_t n) {
    const double t = now();
    (void)hipMemcpy(d, s, n, hipMemcpyHostToDevice);
    return now() - t;
}
int main() {
    double t0 = now();
    int n = 0;
    (void)hipGetDeviceCount(&n);
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    printf("runtime init %.1f ms\n", now() - t0);
    const size_t B = 9u << 20;
    void *dev_old = nullptr;
    (void)hipMalloc(&dev_old, B);
    std::vector<char> host_old(B, 1);
    printf("very first copy                       %.2f ms\n", copy_ms(dev_old, host_old.data(), B));
    printf("same buffers again                    %.2f ms\n", copy_ms(dev_old, host_old.data(), B));
    for (int rep = 0; rep < 2; ++rep) {
        { std::vector<char> h(B, 1); void *d; (void)hipMalloc(&d, B);
          printf("A fresh host, fresh device            %.2f ms\n", copy_ms(d, h.data(), B)); (void)hipFree(d); }
        { std::vector<char> h(B, 1);
          printf("B fresh host, used device             %.2f ms\n", copy_ms(dev_old, h.data(), B)); }
        { void *d; (void)hipMalloc(&d, B + 4096 * (rep + 1));
          printf("C used host, fresh device             %.2f ms\n", copy_ms(d, host_old.data(), B)); (void)hipFree(d); }
        { std::vector<char> h(B, 1); void *d; (void)hipMalloc(&d, B + 8192 * (rep + 1));
          double t = now(); (void)hipHostRegister(h.data(), B, hipHostRegisterDefault); double tr = now() - t;
          printf("D fresh host registered, fresh device %.2f ms (+ register %.2f ms)\n", copy_ms(d, h.data(), B), tr);
          (void)hipHostUnregister(h.data()); (void)hipFree(d); }
        { void *hp; (void)hipHostMalloc(&hp, B, hipHostMallocDefault); memset(hp, 1, B); void *d; (void)hipMalloc(&d, B + 12288 * (rep + 1));
          printf("E hipHostMalloc'd host, fresh device  %.2f ms\n", copy_ms(d, hp, B)); (void)hipFree(d); (void)hipHostFree(hp); }
        { std::vector<char> h(B, 1); void *d; (void)hipMalloc(&d, B + 16384 * (rep + 1)); (void)hipMemset(d, 0, B); (void)hipDeviceSynchronize();
          printf("F fresh host, device memset first     %.2f ms\n", copy_ms(d, h.data(), B)); (void)hipFree(d); }
    }
    return 0;
}
