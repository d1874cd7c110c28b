// Diagnostic (not part of the product): which XCD does block b of a launch run on, launch after launch?
// Prints, for a few consecutive launches of a 304-block grid, the XCC id of blocks 0..15 and whether the
// block -> XCD map of a launch equals that of the launch before it.  (If it does, a weight tile that a block index
// reads every launch stays in one XCD's L2 only if kernel boundaries do not invalidate L2.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x] = x & 0xF;
  }
}
int main() {
  const int nb = 304;
  unsigned* d;
  hipMalloc(&d, nb * 4 * 8);
  std::vector<unsigned> h(nb * 8);
  for (int l = 0; l < 8; ++l) hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d + l * nb);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), d, nb * 4 * 8, hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) {
    printf("launch %d:", l);
    for (int b = 0; b < 16; ++b) printf(" %u", h[l * nb + b]);
    int same = 0, rr = 0;
    for (int b = 0; b < nb; ++b) {
      same += l > 0 && h[l * nb + b] == h[(l - 1) * nb + b];
      rr += h[l * nb + b] == (h[l * nb] + b) % 8;
    }
    printf("  | same as previous launch: %d/%d, round-robin from block 0's XCD: %d/%d\n", same, nb, rr, nb);
  }
  return 0;
}
