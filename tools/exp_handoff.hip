// exp_handoff.hip -- what does confining a Gauss-Newton step's exchange to ONE XCD buy?  (VERDICT r3 #3a; not part of the library)
//
// The persistent ICP loop's step: every publisher stores 27 tagged 64-bit words (value, tag), every folder polls all publishers' words until their tags are
// current, adds them up, and goes on.  This program times that exchange alone, 200 steps per launch, one 512-lane workgroup per CU (64 KiB of LDS each) as
// in k_icp_loop, for several placements / cache policies:
//   spread-all   : P publishers = workgroups 0..P-1 (round-robin over the 8 XCDs), ALL workgroups of the launch fold; sc1 stores, sc1 loads   (the product at the coarse levels)
//   spread-own   : the same publishers, only they fold
//   xcd-sc1      : P publishers = workgroups 0, 8, 16, ... (one XCD), only they fold; sc1 stores, sc1 loads
//   xcd-plain    : the same, PLAIN stores (the line stays in that XCD's L2) + sc1 loads (bypass L1, served by L2)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/scratch/exp_handoff tools/exp_handoff.hip ;  run: tools/scratch/exp_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define STEPS 200
#define MAX_WG 256

struct Args {
  unsigned long long* slots;      // STEPS x MAX_WG x 32 words
  unsigned long long* stamps;     // per workgroup: ticks (100 MHz) from its first to its last step
  unsigned* xcc;                  // per workgroup: XCC_ID
  int n_wg, n_pub, stride, all_fold, plain_store, work_sleep;
  unsigned tag_base;
};

__global__ void __launch_bounds__(512) k_exchange(Args a) {
  extern __shared__ float s_dyn[];                      // (only to keep one workgroup per CU)
  __shared__ float s_tot[32];
  const int wg = blockIdx.x, k = threadIdx.x & 31, part = threadIdx.x >> 5;
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    a.xcc[wg] = x & 0xF;
  }
  const bool publisher = (wg % a.stride) == 0 && (wg / a.stride) < a.n_pub;
  const bool folder = a.all_fold || publisher;
  if (!folder) return;
  const int my_pub = wg / a.stride;
  float acc = 0.f;
  unsigned long long t0 = 0, t1 = 0;
  for (int step = 0; step < STEPS; ++step) {
    if (step == 8 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned tag = a.tag_base + (unsigned)step;
    unsigned long long* row = a.slots + (size_t)step * MAX_WG * 32;
    // the "pixel phase": a fixed delay, then the workgroup's 27 sums are published by lanes 0..26
    for (int i = 0; i < a.work_sleep; ++i) __builtin_amdgcn_s_sleep(8);
    if (publisher && threadIdx.x < 27) {
      const unsigned long long w = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(1.0f + acc * 1e-9f);
      if (a.plain_store) { row[my_pub * 32 + threadIdx.x] = w; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else __hip_atomic_store(row + my_pub * 32 + threadIdx.x, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the fold: part p (32 lanes) takes publishers p, p + 16, ...; polls until the tag is current
    float s = 0.f;
    if (k < 27)
      for (int p = part; p < a.n_pub; p += 16) {
        unsigned long long u;
        do { u = __hip_atomic_load(row + p * 32 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(u >> 32) != tag);
        s += __uint_as_float((unsigned)u);
      }
    if (part == 0) s_tot[k] = 0.f;
    __syncthreads();
    if (k < 27 && s != 0.f) atomicAdd(&s_tot[k], s);
    __syncthreads();
    acc += s_tot[k & 15];
    __syncthreads();
  }
  if (threadIdx.x == 0) { t1 = __builtin_amdgcn_s_memrealtime(); a.stamps[wg] = t1 - t0; }
  if (acc == 12345.678f) a.stamps[wg] = 0;             // (keep acc alive)
}

int main() {
  Args a;
  hipMalloc(&a.slots, (size_t)STEPS * MAX_WG * 32 * 8);
  hipMalloc(&a.stamps, MAX_WG * 8);
  hipMalloc(&a.xcc, MAX_WG * 4);
  hipMemset(a.slots, 0, (size_t)STEPS * MAX_WG * 32 * 8);
  hipFuncSetAttribute((const void*)k_exchange, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  unsigned seq = 0;
  struct Mode { const char* name; int n_wg, n_pub, stride, all_fold, plain; };
  const Mode modes[] = {
    {"spread-all  P=13 (200 fold)", 200, 13, 1, 1, 0}, {"spread-own  P=13", 200, 13, 1, 0, 0}, {"xcd-sc1     P=13", 200, 13, 8, 0, 0}, {"xcd-plain   P=13", 200, 13, 8, 0, 1},
    {"spread-all  P=25 (200 fold)", 200, 25, 1, 1, 0}, {"spread-own  P=25", 200, 25, 1, 0, 0}, {"xcd-sc1     P=25", 200, 25, 8, 0, 0}, {"xcd-plain   P=25", 200, 25, 8, 0, 1},
    {"spread-all  P=50 (200 fold)", 200, 50, 1, 1, 0}, {"spread-own  P=50", 200, 50, 1, 0, 0},
    {"spread-all  P=200", 200, 200, 1, 1, 0},
  };
  for (int work = 0; work <= 2; work += 2)
    for (const Mode& m : modes) {
      std::vector<double> per_step;
      std::vector<unsigned> xcc(MAX_WG);
      for (int rep = 0; rep < 7; ++rep) {
        seq += 256;
        a.n_wg = m.n_wg; a.n_pub = m.n_pub; a.stride = m.stride; a.all_fold = m.all_fold; a.plain_store = m.plain; a.work_sleep = work; a.tag_base = seq;
        hipMemset(a.stamps, 0, MAX_WG * 8);
        hipLaunchKernelGGL(k_exchange, dim3(m.n_wg), dim3(512), 96 * 1024, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<unsigned long long> st(MAX_WG);
        hipMemcpy(st.data(), a.stamps, MAX_WG * 8, hipMemcpyDeviceToHost);
        hipMemcpy(xcc.data(), a.xcc, MAX_WG * 4, hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int i = 0; i < m.n_wg; ++i) mx = std::max(mx, st[i]);
        if (rep >= 2) per_step.push_back((double)mx * 0.01 / (STEPS - 8));   // us per step (100 MHz ticks)
      }
      std::sort(per_step.begin(), per_step.end());
      int on_xcd0 = 0;
      for (int p = 0; p < m.n_pub; ++p) on_xcd0 += xcc[p * m.stride] == xcc[0];
      printf("work %d  %-30s median %.2f us per step (min %.2f max %.2f); %d of %d publishers on workgroup 0's XCD\n", work, m.name, per_step[per_step.size() / 2],
             per_step.front(), per_step.back(), on_xcd0, m.n_pub);
    }
  return 0;
}
