#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SCOPE>
__global__ void k(unsigned* word, unsigned want, unsigned* wins_per_xcd, unsigned long long* t) {
  unsigned long long t0 = wall_clock64();
  unsigned old = __hip_atomic_exchange(word, want, __ATOMIC_RELAXED, SCOPE);
  if (threadIdx.x == 0 && old != want) atomicAdd(&wins_per_xcd[blockIdx.x & 7], 1u);
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int SCOPE>
__global__ void kcas(unsigned* word, unsigned seen, unsigned want, unsigned* wins_per_xcd, unsigned long long* t) {
  unsigned long long t0 = wall_clock64();
  unsigned old = seen;
  if (threadIdx.x == 0) { __hip_atomic_compare_exchange_strong(word, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE); if (old == seen) atomicAdd(&wins_per_xcd[blockIdx.x & 7], 1u); }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0 + (old & 0);
}
int main() {
  unsigned* word; unsigned* wins; unsigned long long* t;
  hipMalloc(&word, 4096); hipMalloc(&wins, 64); hipMalloc(&t, 8 * 4096);
  const int G = 336;
  for (int mode = 0; mode < 4; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(word, 0, 4096); hipMemset(wins, 0, 64);
      hipDeviceSynchronize();
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (mode == 0) k<__HIP_MEMORY_SCOPE_AGENT><<<G, 64>>>(word, 7u + rep, wins, t);
      if (mode == 1) k<__HIP_MEMORY_SCOPE_WORKGROUP><<<G, 64>>>(word, 7u + rep, wins, t);
      if (mode == 2) kcas<__HIP_MEMORY_SCOPE_AGENT><<<G, 64>>>(word, 0u, 7u + rep, wins, t);
      if (mode == 3) kcas<__HIP_MEMORY_SCOPE_WORKGROUP><<<G, 64>>>(word, 0u, 7u + rep, wins, t);
      hipEventRecord(b); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned hw[8]; unsigned long long ht[G];
      hipMemcpy(hw, wins, 32, hipMemcpyDeviceToHost); hipMemcpy(ht, t, 8 * G, hipMemcpyDeviceToHost);
      unsigned long long mx = 0, sum = 0; for (int i = 0; i < G; i++) { if (ht[i] > mx) mx = ht[i]; sum += ht[i]; }
      printf("mode %d rep %d: kernel %.1f us, per-wg atomic latency mean %.2f us max %.2f us, wins per xcd:", mode, rep, ms * 1e3, sum / (double)G / 100.0, mx / 100.0);
      for (int i = 0; i < 8; i++) printf(" %u", hw[i]);
      printf("\n");
    }
  }
  return 0;
}
