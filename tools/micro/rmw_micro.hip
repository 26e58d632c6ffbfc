// rmw_micro.hip -- what can a read-modify-write walk over 4-KiB bricks reach on this chip?  (tools only; hipcc --offload-arch=gfx950)
// Sweeps the structure of the brick walk: bricks in flight per workgroup, grid size, temporal hints, workgroup size, scattered vs
// contiguous slots.  usage: rmw_micro [n_bricks] [total_bricks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// NT: 0 plain, 1 nt load, 2 nt store, 3 both.  SCAT: slot from the table, else the index itself.
template <int BR, int NT, bool SCAT, int THREADS>
__global__ void __launch_bounds__(THREADS) k_rmw(float* vol, const unsigned* __restrict__ table, unsigned n) {
  constexpr int PER = 1024 / (THREADS / 256) / 4;   // float4 per thread per brick: 256 threads -> 1, 512 -> would need half... (only 256 used for PER=1)
  for (unsigned q0 = blockIdx.x * BR; q0 < n; q0 += gridDim.x * BR) {
    v4* p[BR]; v4 q[BR];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const unsigned i = q0 + b < n ? q0 + b : q0;
      const unsigned slot = SCAT ? table[i] : i;
      p[b] = reinterpret_cast<v4*>(vol + (size_t)slot * 1024) + threadIdx.x;
      q[b] = (NT & 1) ? __builtin_nontemporal_load(p[b]) : *p[b];
    }
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      q[b].y += 1.0f;
      if (NT & 2) __builtin_nontemporal_store(q[b], p[b]); else *p[b] = q[b];
    }
  }
}
// one WAVE per brick quarter... variant: each wave walks its own bricks (wave = 64 lanes x 4 float4 = 4 KiB), no workgroup coupling
template <int BR, bool SCAT>
__global__ void __launch_bounds__(256) k_rmw_wave(float* vol, const unsigned* __restrict__ table, unsigned n) {
  const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = gridDim.x * 4;
  for (unsigned q0 = wave * BR; q0 < n; q0 += nw * BR) {
    v4* p[BR][4]; v4 q[BR][4];
#pragma unroll
    for (int b = 0; b < BR; ++b) {
      const unsigned i = q0 + b < n ? q0 + b : q0;
      const unsigned slot = SCAT ? table[i] : i;
#pragma unroll
      for (int k = 0; k < 4; ++k) { p[b][k] = reinterpret_cast<v4*>(vol + (size_t)slot * 1024) + k * 64 + lane; q[b][k] = *p[b][k]; }
    }
#pragma unroll
    for (int b = 0; b < BR; ++b)
#pragma unroll
      for (int k = 0; k < 4; ++k) { q[b][k].y += 1.0f; *p[b][k] = q[b][k]; }
  }
}

template <typename F> static float time_it(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 184086u, total = argc > 2 ? (unsigned)atoi(argv[2]) : 2097152u;
  float* vol; CK(hipMalloc((void**)&vol, (size_t)total * 4096)); CK(hipMemset(vol, 0, (size_t)total * 4096));
  // scattered table: runs of 4 x-adjacent bricks, runs spread pseudo-randomly but sorted (like a frustum walk in macro-cell order)
  std::vector<unsigned> t(n);
  { unsigned s = 12345u; std::vector<unsigned> starts(n / 4 + 1);
    for (auto& v : starts) { s = s * 1664525u + 1013904223u; v = (s >> 4) % (total / 4); }
    std::sort(starts.begin(), starts.end());
    for (unsigned i = 0; i < n; ++i) t[i] = starts[i / 4] * 4 + (i & 3); }
  unsigned* table; CK(hipMalloc((void**)&table, n * 4)); CK(hipMemcpy(table, t.data(), n * 4, hipMemcpyHostToDevice));
  const double gb = (double)n * 8192.0 / 1e9;
#define RUN(name, kern, grid, threads) do { float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, vol, table, n); }, 20); \
    printf("%-44s grid %6d: %.1f us  %.2f TB/s\n", name, (int)(grid), ms * 1000.f, gb / ms); } while (0)
  for (int grid : {1024, 2048, 4096, 8192, 16384, 32768}) {
    RUN("contig BR1", (k_rmw<1, 0, false, 256>), grid, 256);
    RUN("contig BR2", (k_rmw<2, 0, false, 256>), grid, 256);
    RUN("contig BR4", (k_rmw<4, 0, false, 256>), grid, 256);
    RUN("contig BR8", (k_rmw<8, 0, false, 256>), grid, 256);
    RUN("scatter BR4", (k_rmw<4, 0, true, 256>), grid, 256);
    RUN("scatter BR8", (k_rmw<8, 0, true, 256>), grid, 256);
  }
  for (int grid : {4096, 8192}) {
    RUN("scatter BR4 nt-load", (k_rmw<4, 1, true, 256>), grid, 256);
    RUN("scatter BR4 nt-store", (k_rmw<4, 2, true, 256>), grid, 256);
    RUN("scatter BR4 nt-both", (k_rmw<4, 3, true, 256>), grid, 256);
    RUN("scatter wave-per-brick BR1", (k_rmw_wave<1, true>), grid, 256);
    RUN("scatter wave-per-brick BR2", (k_rmw_wave<2, true>), grid, 256);
    RUN("contig wave-per-brick BR2", (k_rmw_wave<2, false>), grid, 256);
  }
  return 0;
}
