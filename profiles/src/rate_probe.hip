// rate_probe.hip — issue rates of the vector instructions the fp32-mixed sweeps are built from, on one MI355X.
//   hipcc -O3 --offload-arch=gfx950 profiles/src/rate_probe.hip -o profiles/src/rate_probe && profiles/src/rate_probe
// Every kernel: 256 CUs x 4 groups of 256 threads (4 waves per SIMD), NACC independent accumulator chains per thread, ITERS passes.
// Reports wave-instruction cycles per SIMD (at the clock hipDeviceAttributeClockRate reports) and TFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int NACC = 16;

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float a, float b) {
  // MODE 0: v_fma_f32   1: v_pk_fma_f32 (plain)   2: v_fma_f64   3: v_pk_fma_f32 with op_sel swap + neg_hi on a broadcast low half
  // MODE 4: v_pk_fma_f32 with an SGPR-pair coefficient   5: v_pk_mul_f32 + v_pk_add_f32 pairs
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  if constexpr (MODE == 0) {
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (float)(tid + i) * 1e-9f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[tid] = s;
  } else if constexpr (MODE == 2) {
    double acc[NACC];
    const double ad = a, bd = b;
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (double)(tid + i) * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(ad), "v"(bd));
      }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[tid] = (float)s;
  } else {
    f2v acc[NACC];
    const f2v av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = f2v{(float)(tid + i) * 1e-9f, (float)(tid - i) * 1e-9f};
    float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
    const f2v sav = {sa, sa};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
          if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(av), "v"(bv));
          if constexpr (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "+v"(acc[i]) : "v"(av), "v"(bv));
          if constexpr (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "s"(sav), "v"(bv));
          if constexpr (MODE == 5) {
            if (i & 1) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(av));
            else asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(bv));
          }
        }
      }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y;
    out[tid] = s;
  }
}

// LDS read rate: every wave reads NRD values per pass from a 16 KB buffer at lane-dependent XOR addresses (the stencil's pattern)
template <int BYTES>
__global__ void __launch_bounds__(256) k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float buf[4096 + 64];
  for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = (float)i;
  __syncthreads();
  const unsigned base = (threadIdx.x * BYTES) & 16383u;
  float s = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const unsigned addr = base ^ (unsigned)(BYTES << (r & 7));
      if constexpr (BYTES == 8) {
        f2v v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        s += 0.f;  // (the value is never consumed: pure LDS issue)
        if (it < 0) out[0] = v.x;
      } else {
        float v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        if (it < 0) out[0] = v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}

template <typename K, typename... Args>
static double time_ms(K kern, int blocks, Args... args) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 5; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  int ncu = 0, khz = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
  const int blocks = ncu * 4, iters = 20000;
  float* d;
  CK(hipMalloc(&d, sizeof(float) * blocks * 256));
  const double ghz = khz * 1e-6;
  const double winst = (double)iters * 4 * NACC;  // wave-instructions per wave; 4 waves per SIMD
  struct Row { const char* name; double ms; double flop_per_lane_inst; };
  std::vector<Row> rows;
  rows.push_back({"v_fma_f32", time_ms(k_rate<0>, blocks, d, iters, 0.999f, 1e-9f), 2});
  rows.push_back({"v_pk_fma_f32", time_ms(k_rate<1>, blocks, d, iters, 0.999f, 1e-9f), 4});
  rows.push_back({"v_fma_f64", time_ms(k_rate<2>, blocks, d, iters, 0.999f, 1e-9f), 2});
  rows.push_back({"v_pk_fma_f32 op_sel+neg", time_ms(k_rate<3>, blocks, d, iters, 0.999f, 1e-9f), 4});
  rows.push_back({"v_pk_fma_f32 sgpr coeff", time_ms(k_rate<4>, blocks, d, iters, 0.999f, 1e-9f), 4});
  rows.push_back({"v_pk_mul/add_f32", time_ms(k_rate<5>, blocks, d, iters, 0.999f, 1e-9f), 2});
  printf("{\"device_cus\": %d, \"clock_ghz\": %.3f, \"rows\": [\n", ncu, ghz);
  for (auto& r : rows) {
    const double cyc = r.ms * 1e-3 * ghz * 1e9 / (winst * 4);  // per wave-instruction per SIMD (4 waves per SIMD)
    const double tf = winst * 64 * r.flop_per_lane_inst * (double)blocks * 4 / (r.ms * 1e-3) * 1e-12;
    printf("  {\"inst\": \"%s\", \"ms\": %.3f, \"cycles_per_wave_inst_per_simd\": %.3f, \"tflops\": %.1f},\n", r.name, r.ms, cyc, tf);
  }
  const int liters = 20000;
  const double l64 = time_ms(k_lds<8>, blocks, d, liters), l32 = time_ms(k_lds<4>, blocks, d, liters);
  const double nld = (double)liters * 16 * 16;  // wave-instructions per CU (16 waves)
  printf("  {\"inst\": \"ds_read_b64\", \"ms\": %.3f, \"cycles_per_wave_inst_per_cu\": %.3f},\n", l64, l64 * 1e-3 * ghz * 1e9 / nld);
  printf("  {\"inst\": \"ds_read_b32\", \"ms\": %.3f, \"cycles_per_wave_inst_per_cu\": %.3f}\n]}\n", l32, l32 * 1e-3 * ghz * 1e9 / nld);
  return 0;
}
