// dev experiment: cycles per LDS access of one wave, aligned against unaligned addresses (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct __attribute__((packed, aligned(1))) u32p { uint32_t v; };
struct __attribute__((packed, aligned(1))) u64p { uint64_t v; };
template <int KIND>
__global__ void k(unsigned long long* out, int off, int stride) {
  __shared__ uint8_t buf[16384];
  const int lane = threadIdx.x;
  for (int i = lane; i < 16384; i += 64) buf[i] = (uint8_t)i;
  __syncthreads();
  uint32_t a = (uint32_t)(lane * stride + off) & 8191u;
  uint64_t acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < 512; it++) {
    if (KIND == 0) acc += ((const u32p*)(buf + a))->v;
    if (KIND == 1) acc += ((const u64p*)(buf + a))->v;
    if (KIND == 2) acc += buf[a];
    if (KIND == 3) { ((u32p*)(buf + a))->v = (uint32_t)acc; acc += it; }
    if (KIND == 4) { ((u64p*)(buf + a))->v = acc; acc += it; }
    if (KIND == 5) { buf[a] = (uint8_t)acc; acc += it; }
    a = (a + (uint32_t)(acc & 0u) + 64u) & 8191u;   // (dependent on the loaded value: the accesses are serialised)
    if (KIND <= 2) a = (a + (uint32_t)(acc & 0u)) & 8191u;
  }
  const unsigned long long t1 = clock64();
  if (lane == 0) { out[0] = t1 - t0; out[1] = acc; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  const char* nm[6] = {"read b32", "read b64", "read u8", "write b32", "write b64", "write u8"};
  for (int kind = 0; kind < 6; kind++)
    for (int stride : {8, 17, 41})
      for (int off = 0; off < 4; off++) {
        switch (kind) {
          case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, off, stride); break;
          case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, off, stride); break;
          case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, off, stride); break;
          case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, off, stride); break;
          case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, off, stride); break;
          case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, d, off, stride); break;
        }
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-9s stride %2d off %d: %6.1f cycles per access\n", nm[kind], stride, off, (double)h[0] / 512.0);
      }
  return 0;
}
