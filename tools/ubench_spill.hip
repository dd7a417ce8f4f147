// Development aid: how fast can many independent waves stream a private region each, in the access shape of
// the align kernel's spill (one wave = one 64-thread block; 16 bytes per lane and access, 1 KB per
// wave-instruction, consecutive accesses 1 KB apart; `burst` accesses issued back to back)?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_spill.hip -o tools/ubench_spill && tools/ubench_spill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// mode 0: write, 1: read, 2: even blocks write while odd blocks read (other halves' regions)
template <int BURST>
__global__ __launch_bounds__(64) void stream(v4i *buf, long long per_wave16, int mode, int pace, v4i *sink) {
  const int lane = threadIdx.x;
  v4i *mine = buf + (long long)blockIdx.x * per_wave16;
  const bool wr = (mode == 0) || (mode == 2 && (blockIdx.x & 1) == 0);
  v4i acc = {0, 0, 0, 0};
  float f = (float)lane;
  for (long long p = 0; p < per_wave16; p += 64 * BURST) {
    if (wr) {
#pragma unroll
      for (int b = 0; b < BURST; b++) mine[p + b * 64 + lane] = v4i{(int)p, lane, b, 1};
    } else {
      v4i t[BURST];
#pragma unroll
      for (int b = 0; b < BURST; b++) t[b] = mine[p + b * 64 + lane];
#pragma unroll
      for (int b = 0; b < BURST; b++) acc += t[b];
    }
    for (int k = 0; k < pace * BURST; k++) f = __builtin_fmaf(f, 1.0001f, 0.5f);  // pacing: dependent FMAs
  }
  if (acc.x == 0x7fffffff || f == 1.2345f) sink[blockIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 4096;
  const long long bytes_per_wave = argc > 2 ? atoll(argv[2]) : (long long)2560 * 1024;
  const long long per_wave16 = bytes_per_wave / 16 / 256 * 256;
  v4i *buf, *sink;
  hipMalloc(&buf, (size_t)waves * per_wave16 * 16);
  hipMalloc(&sink, (size_t)waves * 16);
  hipMemset(buf, 0, (size_t)waves * per_wave16 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double gb = (double)waves * per_wave16 * 16 / 1e9;
  printf("%d waves x %.2f MB = %.1f GB per pass\n", waves, per_wave16 * 16 / 1e6, gb);
  for (int pace : {0, 64, 256}) {
    for (int burst : {1, 4}) {
      for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0);
          if (burst == 1) hipLaunchKernelGGL(stream<1>, dim3(waves), dim3(64), 0, 0, buf, per_wave16, mode, pace, sink);
          else hipLaunchKernelGGL(stream<4>, dim3(waves), dim3(64), 0, 0, buf, per_wave16, mode, pace, sink);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("pace %3d burst %d %-5s: %7.3f ms  %6.0f GB/s\n", pace, burst, mode == 0 ? "write" : mode == 1 ? "read" : "mixed", best, gb / best * 1e3);
      }
    }
  }
  return 0;
}
