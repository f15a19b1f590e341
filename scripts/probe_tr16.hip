// Hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (build+run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(short* out, int rowbytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  short* t = (short*)lds;
  const int cols = rowbytes / 2;
  for (int i = threadIdx.x; i < 16 * cols; i += 64) t[i] = (short)((i / cols) * 100 + (i % cols));  // value = row*100+col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int h = g >> 1, cb = g & 1;
  // block: rows 8h..8h+3 (first read) , columns cb*16 .. +15
  char* addr = lds + (8 * h + q) * rowbytes + (cb * 16 + 4 * p) * 2;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16 * 256, 0, d, 256);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int g = lane >> 4, li = lane & 15, hh = g >> 1, cb = g & 1;
    for (int e = 0; e < 4; ++e) {
      int want = (8 * hh + e) * 100 + cb * 16 + li;   // element e = row (8h+e), column cb*16+lane_in_group
      if (h[lane * 4 + e] != want) { if (bad < 12) printf("lane %d e %d got %d want %d\n", lane, e, h[lane*4+e], want); ++bad; }
    }
  }
  printf("tr16 probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "semantics as assumed", bad);
  printf("lane0: %d %d %d %d  lane1: %d %d %d %d lane17: %d %d %d %d lane33: %d %d %d %d\n", h[0],h[1],h[2],h[3],h[4],h[5],h[6],h[7],h[68],h[69],h[70],h[71],h[132],h[133],h[134],h[135]);
  return 0;
}
