// tools/probe/hwid_probe.hip -- which SIMD does wavefront w of a workgroup land on?  (diagnostic; see DESIGN.md, SAP kernel)
// hipcc --offload-arch=gfx950 -O2 -o tools/probe/hwid_probe tools/probe/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  __shared__ float lds[30000];   // ~120 KB: one workgroup per CU, as the SAP kernel
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  lds[threadIdx.x] = hw;
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;   // keep the workgroup resident for a while
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
  if (a == 12345.f) out[0] = (unsigned)lds[5];
}
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 512, nb = 1024, nw = nt / 64;
  unsigned* d; hipMalloc(&d, sizeof(unsigned) * nb * nw);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(nt), 0, 0, d, 20000);
  std::vector<unsigned> h(nb * nw);
  hipMemcpy(h.data(), d, sizeof(unsigned) * nb * nw, hipMemcpyDeviceToHost);
  int hist[16][4] = {{0}};
  for (int b = 0; b < nb; b++) for (int w = 0; w < nw; w++) hist[w][(h[b * nw + w] >> 4) & 3]++;
  for (int w = 0; w < nw; w++) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 6; b++) { printf("wg %d:", b); for (int w = 0; w < nw; w++) printf(" [simd %u wave %u cu %u se %u]", (h[b*nw+w]>>4)&3, h[b*nw+w]&15, (h[b*nw+w]>>8)&15, (h[b*nw+w]>>13)&7); printf("\n"); }
  return 0;
}
