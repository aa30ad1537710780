#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef uint32_t u32a1 __attribute__((aligned(1)));
struct __attribute__((packed)) P2 { uint32_t a, b; };
struct __attribute__((packed)) P3 { uint32_t a, b, c; };
// each lane writes n = 1 + (lane & 3) tokens of 3 bytes at byte offset off (exclusive prefix sum), tokens from a pattern
__global__ void k(uint8_t* out, int mode) {
  __shared__ __align__(16) uint8_t row[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) row[i] = 0xFF;
  __syncthreads();
  int n = 1 + (lane & 3);
  int off = 0;
  for (int j = 0; j < lane; j++) off += 1 + (j & 3);
  uint8_t* dst = row + off * 3 + mode;  // mode shifts the phase too
  uint32_t loc = lane, f0 = 0x10 + lane, v0 = 0x50 + lane;
  uint32_t d0 = loc | (f0 << 8) | (v0 << 16) | (loc << 24);
  uint32_t d1 = (f0 + 1) | ((v0 + 1) << 8) | (loc << 16) | ((f0 + 2) << 24);
  uint32_t d2 = (v0 + 2) | (loc << 8) | ((f0 + 3) << 16) | ((v0 + 3) << 24);
  if (n == 1) {
    *(uint16_t __attribute__((aligned(1)))*)dst = (uint16_t)d0;
    dst[2] = (uint8_t)(d0 >> 16);
  } else if (n == 2) {
    *(u32a1*)dst = d0;
    *(uint16_t __attribute__((aligned(1)))*)(dst + 4) = (uint16_t)d1;
  } else if (n == 3) {
    P2 p{d0, d1};
    *(P2*)dst = p;
    dst[8] = (uint8_t)d2;
  } else {
    P3 p{d0, d1, d2};
    *(P3*)dst = p;
  }
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = row[i];
}
int main() {
  uint8_t* d; hipMalloc(&d, 1024);
  int bad = 0;
  for (int mode = 0; mode < 4; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    std::vector<uint8_t> h(1024);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    std::vector<uint8_t> ref(1024, 0xFF);
    int off = 0;
    for (int lane = 0; lane < 64; lane++) {
      int n = 1 + (lane & 3);
      for (int t = 0; t < n; t++) { ref[off * 3 + mode + t * 3] = lane; ref[off * 3 + mode + t * 3 + 1] = 0x10 + lane + t; ref[off * 3 + mode + t * 3 + 2] = 0x50 + lane + t; }
      off += n;
    }
    int nb = 0;
    for (int i = 0; i < 1024; i++) if (h[i] != ref[i]) { if (nb < 5) printf("mode %d byte %d: got %02x want %02x\n", mode, i, h[i], ref[i]); nb++; }
    printf("mode %d: %d mismatches\n", mode, nb);
    bad += nb;
  }
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
