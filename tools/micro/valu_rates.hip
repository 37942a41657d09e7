// Vector-instruction THROUGHPUT per SIMD with several wavefronts resident (gfx950): what an instruction costs a kernel that is bound
// by vector issue (the tile kernels at five waves per SIMD), as opposed to one wave alone (tools/micro/rate_f32.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Every wave runs N independent instructions of one kind (8 chains) between two s_memtime stamps; W waves share a SIMD (workgroups of
// 256 W threads, one per CU); printed: cycles per instruction PER SIMD = median wave's cycles / (W N).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, long long* cyc, float a0, double d0) {
  f2 x[8];
  double y[8];
  for (int i = 0; i < 8; ++i) { x[i] = f2{a0 + i + threadIdx.x, a0 - i}; y[i] = d0 + i + threadIdx.x; }
  const f2 m = f2{1.0001f, 0.9999f}, c = f2{a0, -a0};
  const double dm = 1.0000001, dc = d0;
  int iv = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const int j = u & 7;
      if (MODE == 0) x[j][0] = __builtin_fmaf(x[j][0], m[0], c[0]);                     // v_fma_f32
      if (MODE == 1) x[j] = __builtin_elementwise_fma(x[j], m, c);                      // v_pk_fma_f32
      if (MODE == 2) y[j] = __builtin_fma(y[j], dm, dc);                                // v_fma_f64
      if (MODE == 3) x[j][0] = __builtin_amdgcn_rsqf(x[j][0]);                          // v_rsq_f32
      if (MODE == 4) x[j][0] = (float)y[j], y[j] = y[j] + dc;                           // v_cvt_f32_f64 + v_add_f64 (2 instructions)
      if (MODE == 5) x[j][0] = x[j][0] > c[0] ? x[j][1] : m[0], x[j][1] += m[1];       // v_cmp + v_cndmask + v_add (3)
      if (MODE == 6) x[j] = x[j] * m;                                                   // v_pk_mul_f32
      if (MODE == 7) { asm volatile("v_mov_b32 %0, %1" : "=v"(iv) : "v"(iv)); }         // v_mov_b32
      if (MODE == 8) x[j][0] = x[j][0] * m[0];                                          // v_mul_f32
      if (MODE == 9) y[j] = y[j] * dm;                                                  // v_mul_f64
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)iv;
  for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1] + (float)y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per, float* out, long long* cyc) {
  for (int W : {1, 2, 4}) {
    const int threads = 256 * W, grid = 256, nw = grid * threads / 64;
    for (int r = 0; r < 3; ++r) { k<MODE><<<grid, threads>>>(out, cyc, 1.f, 1.0); hipDeviceSynchronize(); }
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s W = %d waves per SIMD: %6.2f cycles per instruction per SIMD (wave median %lld cycles)\n", name, W,
           (double)h[nw / 2] / (1024.0 * per * W), h[nw / 2]);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, sizeof(long long) * 256 * 16);
  run<0>("v_fma_f32", 1, out, cyc);
  run<1>("v_pk_fma_f32", 1, out, cyc);
  run<8>("v_mul_f32", 1, out, cyc);
  run<6>("v_pk_mul_f32", 1, out, cyc);
  run<2>("v_fma_f64", 1, out, cyc);
  run<9>("v_mul_f64", 1, out, cyc);
  run<3>("v_rsq_f32", 1, out, cyc);
  run<4>("v_cvt_f32_f64 + v_add_f64 (per pair)", 1, out, cyc);
  run<5>("v_cmp + v_cndmask + v_add (per triple)", 1, out, cyc);
  run<7>("v_mov_b32", 1, out, cyc);
  return 0;
}
