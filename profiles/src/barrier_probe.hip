// barrier_probe.hip - cost of a barrier over G resident 1024-thread workgroups on MI355X, with a realistic amount of dirty data
// between two barriers (every thread rewrites EL complex doubles of a vector that the other workgroups read afterwards).
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip ; run: ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VARIANT>
__device__ __forceinline__ void team_barrier(unsigned long long* bar, unsigned long long* xbar, unsigned long long& target, unsigned long long& xtarget,
                                             int G, int member) {
  if (VARIANT == 0) {  // every wave: agent-scope release before, acquire after
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    target += G;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  } else if (VARIANT == 1 || VARIANT == 2) {  // every wave waits for its own stores; one thread does the agent-scope release / acquire
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += G;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(VARIANT == 2 ? 8 : 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  } else {  // 3: two levels - members with the same blockIdx % 8 (one XCD under round-robin dispatch) meet at their own counter first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nx = G < 8 ? G : 8, per = G / nx, x = member % nx;
    target += nx;
    xtarget += per;
    if (threadIdx.x == 0) {
      unsigned long long* xb = xbar + 16 * x;
      const unsigned long long prev = __hip_atomic_fetch_add(xb, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1 == xtarget) __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);  // last of its group
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

template <int VARIANT>
__global__ void __launch_bounds__(1024) k_probe(unsigned long long* bar, double2* v, int n, int rounds, int G, double* out) {
  const int member = blockIdx.x, gtid = member * 1024 + threadIdx.x, gnt = G * 1024;
  unsigned long long target = 0, xtarget = 0;
  unsigned long long* xbar = bar + 16;
  double acc = 0.0;
  for (int r = 0; r < rounds; r++) {
    double2* src = v + (size_t)(r & 1) * n;
    double2* dst = v + (size_t)((r & 1) ^ 1) * n;
    for (int e = gtid; e < n; e += gnt) {
      const double2 a = src[(e + 1027) % n];  // written by another workgroup in the previous round
      dst[e] = make_double2(a.x + 1.0, a.y);
      acc += a.x;
    }
    team_barrier<VARIANT>(bar, xbar, target, xtarget, G, member);
  }
  if (gtid == 0) out[0] = acc;
  if (gtid == 0) out[1] = v[(size_t)(rounds & 1) * n + 5].x;  // = rounds if every round saw the previous one's data
}

template <int VARIANT>
int run(int G, int n, int rounds, unsigned long long* bar, double2* v, double* out) {
  CHECK(hipMemset(bar, 0, 4096));
  CHECK(hipMemset(v, 0, sizeof(double2) * 2 * n));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  void* args[] = {&bar, &v, &n, &rounds, &G, &out};
  CHECK(hipEventRecord(e0));
  CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_probe<VARIANT>), dim3(G), dim3(1024), args, 0, 0));
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double h[2];
  CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  printf("{\"variant\": %d, \"G\": %d, \"n\": %d, \"rounds\": %d, \"us_per_round\": %.3f, \"check\": %.0f}\n", VARIANT, G, n, rounds, 1e3 * ms / rounds, h[1]);
  return 0;
}

int main() {
  unsigned long long* bar;
  double2* v;
  double* out;
  const int nmax = 1 << 20;
  CHECK(hipMalloc(&bar, 4096));
  CHECK(hipMalloc(&v, sizeof(double2) * 2 * nmax));
  CHECK(hipMalloc(&out, 64));
  const int rounds = 2000;
  for (int n : {4096, 160000}) {
    for (int G : {1, 2, 8, 32, 64, 128, 256}) {
      if (run<0>(G, n, rounds, bar, v, out)) return 1;
      if (run<1>(G, n, rounds, bar, v, out)) return 1;
      if (run<2>(G, n, rounds, bar, v, out)) return 1;
      if (run<3>(G, n, rounds, bar, v, out)) return 1;
    }
  }
  return 0;
}
