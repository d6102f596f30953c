// Experiment: what is the fastest way to get N bytes from HBM into fresh host memory?  (tools/experiments, not built
// by default)  hipcc -O2 d2h_paths.cpp -o d2h_paths && ./d2h_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = (size_t)4 << 30;
  void* d; hipMalloc(&d, N); hipMemset(d, 1, N); hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now(); void* h; hipHostMalloc(&h, N, hipHostMallocDefault); double t1 = now();
    hipMemcpy(h, d, N, hipMemcpyDeviceToHost); double t2 = now();
    hipHostFree(h); double t3 = now();
    printf("pinned   : alloc %.3f s, copy %.3f s (%.1f GB/s), free %.3f s\n", t1 - t0, t2 - t1, N / (t2 - t1) / 1e9, t3 - t2);
    t0 = now(); h = malloc(N); t1 = now();
    hipMemcpy(h, d, N, hipMemcpyDeviceToHost); t2 = now();
    printf("pageable : alloc %.3f s, copy %.3f s (%.1f GB/s)\n", t1 - t0, t2 - t1, N / (t2 - t1) / 1e9);
    hipMemcpy(h, d, N, hipMemcpyDeviceToHost); t3 = now();
    printf("pageable (touched): copy %.3f s (%.1f GB/s)\n", t3 - t2, N / (t3 - t2) / 1e9);
    free(h);
    t0 = now(); h = mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); madvise(h, N, MADV_HUGEPAGE); t1 = now();
    hipMemcpy(h, d, N, hipMemcpyDeviceToHost); t2 = now();
    printf("populated: alloc %.3f s, copy %.3f s (%.1f GB/s)\n", t1 - t0, t2 - t1, N / (t2 - t1) / 1e9);
    t2 = now(); hipHostRegister(h, N, hipHostRegisterDefault); t3 = now();
    hipMemcpy(h, d, N, hipMemcpyDeviceToHost); double t4 = now();
    printf("registered: register %.3f s, copy %.3f s (%.1f GB/s)\n", t3 - t2, t4 - t3, N / (t4 - t3) / 1e9);
    hipHostUnregister(h); munmap(h, N);
  }
  return 0;
}
