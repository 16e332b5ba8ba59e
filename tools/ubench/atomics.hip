// microbenchmark: f32 global atomic add throughput on MI355X for the scatter
// patterns of the grid-gradient backward.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);}}while(0)

// pattern A: lane (q=l>>4, i=l&15): point i of a tile, dwords {4q+r, 16+4q+r}, 8 instrs per cell-corner
__global__ void patA(float* g, const int* cells, int npts, float v) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int q = lane >> 4, i = lane & 15;
  int pt = wave * 16 + i; if (pt >= npts) return;
  for (int c = 0; c < 8; ++c) {
    float* dst = g + (size_t)cells[pt * 8 + c] * 32 + 4 * q;
    for (int r = 0; r < 4; ++r) { atomicAdd(dst + r, v); atomicAdd(dst + 16 + r, v); }
  }
}
// pattern B: half-wave (32 lanes) covers one full 128-B cell; wave does 2 points per instr
__global__ void patB(float* g, const int* cells, int npts, float v) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int half = lane >> 5, ch = lane & 31;
  for (int j = 0; j < 8; ++j) {
    int pt = wave * 16 + 2 * j + half; if (pt >= npts) continue;
    for (int c = 0; c < 8; ++c)
      atomicAdd(g + (size_t)cells[pt * 8 + c] * 32 + ch, v);
  }
}
// pattern C: plain (non-atomic) read-modify-write with pattern B addressing (upper bound)
__global__ void patC(float* g, const int* cells, int npts, float v) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  int half = lane >> 5, ch = lane & 31;
  for (int j = 0; j < 8; ++j) {
    int pt = wave * 16 + 2 * j + half; if (pt >= npts) continue;
    for (int c = 0; c < 8; ++c) { float* p = g + (size_t)cells[pt * 8 + c] * 32 + ch; *p = *p + v; }
  }
}
// pattern D: 16 lanes x float4?? no vector atomics; D = pattern B with pk loads of cell ids hoisted
int main() {
  const int ncell = 63 * 75 * 71;  // fine grid
  const int npts = 48000 * 4;
  float* g; int* cells;
  CK(hipMalloc(&g, (size_t)ncell * 32 * 4)); CK(hipMemset(g, 0, (size_t)ncell * 32 * 4));
  std::vector<int> h(npts * 8);
  srand(1);
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: random cells; mode 1: ray-like (runs of neighbouring cells)
    for (int p = 0; p < npts; ++p) {
      int base = (mode == 0) ? rand() % (ncell - 6000) : ((p / 48) * 7919 % (ncell - 6000)) + (p % 48) / 3;
      int d[8] = {0, 1, 71, 72, 71 * 75, 71 * 75 + 1, 71 * 75 + 71, 71 * 75 + 72};
      for (int c = 0; c < 8; ++c) h[p * 8 + c] = base + d[c];
    }
    CK(hipMalloc(&cells, h.size() * 4)); CK(hipMemcpy(cells, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nw = (npts + 15) / 16, nb = (nw * 64 + 255) / 256;
    for (int k = 0; k < 3; ++k) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) {
          if (k == 0) hipLaunchKernelGGL(patA, dim3(nb), dim3(256), 0, 0, g, cells, npts, 1.0f);
          if (k == 1) hipLaunchKernelGGL(patB, dim3(nb), dim3(256), 0, 0, g, cells, npts, 1.0f);
          if (k == 2) hipLaunchKernelGGL(patC, dim3(nb), dim3(256), 0, 0, g, cells, npts, 1.0f);
        }
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        if (rep) printf("mode %d pattern %c: %.1f us for %d pts -> %.1f G dword-atomics/s, %.2f G cells/s\n", mode, 'A' + k, ms * 1e3, npts,
                        npts * 8.0 * 32 / ms / 1e6, npts * 8.0 / ms / 1e6);
      }
    }
    hipFree(cells);
  }
  return 0;
}
