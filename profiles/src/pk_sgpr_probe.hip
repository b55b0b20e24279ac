// pk_sgpr_probe.hip — what v_pk_fma_f32 reads from an SGPR-pair source under op_sel / op_sel_hi (gfx950).
//   hipcc -O3 --offload-arch=gfx950 profiles/src/pk_sgpr_probe.hip -o profiles/src/pk_sgpr_probe && profiles/src/pk_sgpr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(float* out, float a, float b) {
  // SGPR pair (a, b); vector pair x = (1, 10), acc = 0
  v2f sp = {a, b}, x = {1.f, 10.f}, z = {0.f, 0.f}, r;
  v2f vp = {a, b};
  asm volatile("" : "+s"(sp));
  asm volatile("" : "+v"(vp), "+v"(x), "+v"(z));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(sp), "v"(x), "v"(z)); out[0] = r.x; out[1] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "s"(sp), "v"(x), "v"(z)); out[2] = r.x; out[3] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "s"(sp), "v"(x), "v"(z)); out[4] = r.x; out[5] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "s"(sp), "v"(x), "v"(z)); out[6] = r.x; out[7] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(vp), "v"(x), "v"(z)); out[8] = r.x; out[9] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(vp), "v"(x), "v"(z)); out[10] = r.x; out[11] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "s"(sp), "v"(x), "v"(z)); out[12] = r.x; out[13] = r.y;
}
int main() {
  float* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2.f, 3.f);
  float h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  const char* names[] = {"sgpr plain (expect 2,30)", "sgpr bcast lo (expect 2,20)", "sgpr bcast hi (expect 3,30)", "sgpr swap (expect 3,20)", "vgpr bcast lo (expect 2,20)", "vgpr bcast hi (expect 3,30)", "sgpr bcast lo neg (expect -2,-20)"};
  for (int i = 0; i < 7; i++) printf("%-36s -> %g, %g\n", names[i], h[2 * i], h[2 * i + 1]);
  return 0;
}
