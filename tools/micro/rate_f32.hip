// Issue-rate microbenchmarks on one wavefront (gfx950): f32 MFMA chains, packed f32 FMA, ds_bpermute.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/rate_f32.hip -o /tmp/rate_f32 && /tmp/rate_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CH, int VALU>
__global__ void k_mfma(float* out, long long* cyc, float a0, float b0) {
  f4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0, t = a0;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      float av = a;
      if (VALU >= 2) { t = t - b; av = t * t; }
      if (VALU >= 4) { t = t * 1.0001f + b; t = t - av; }
      acc[u % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[u % CH], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = t;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
__global__ void k_valu(float* out, long long* cyc, float a0) {
  f2 x[8];
  for (int i = 0; i < 8; ++i) x[i] = f2{a0 + i, a0 - i};
  f2 m = f2{1.0001f, 0.9999f}, c = f2{a0, -a0};
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      if (MODE == 0) x[u & 7] = __builtin_elementwise_fma(x[u & 7], m, c);                 // v_pk_fma_f32, 8 independent chains
      if (MODE == 1) { x[u & 7][0] = __builtin_fmaf(x[u & 7][0], m[0], c[0]); }            // v_fma_f32
      if (MODE == 2) x[0] = __builtin_elementwise_fma(x[0], m, c);                 // dependent v_pk_fma_f32
      if (MODE == 3) x[u & 7][0] = __builtin_amdgcn_exp2f(x[u & 7][0]);                    // v_exp_f32
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int DEP>
__global__ void k_bperm(float* out, long long* cyc, float a0) {
  int v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 7 + i;
  const int adr = ((threadIdx.x + 16) & 63) * 4;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      if (DEP) v[0] = __builtin_amdgcn_ds_bpermute(adr, v[0]) + 1;
      else v[u & 7] = __builtin_amdgcn_ds_bpermute(adr, v[u & 7]);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[threadIdx.x] = (float)s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 64 * sizeof(float)); hipMalloc(&cyc, sizeof(long long));
  long long h;
#define RUN(name, kern, n) do { for (int r = 0; r < 3; ++r) { kern; hipDeviceSynchronize(); } hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-60s %8.1f cycles per instruction\n", name, (double)h / (n)); } while (0)
  RUN("mfma_f32_16x16x4f32, 1 chain (dependent)", (k_mfma<1, 0><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma_f32_16x16x4f32, 2 chains", (k_mfma<2, 0><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma_f32_16x16x4f32, 4 chains", (k_mfma<4, 0><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma_f32_16x16x4f32, 8 chains", (k_mfma<8, 0><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma 2 chains + sub, mul feeding A", (k_mfma<2, 2><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma 2 chains + 4 VALU", (k_mfma<2, 4><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("mfma 4 chains + sub, mul feeding A", (k_mfma<4, 2><<<1, 64>>>(out, cyc, 1.f, 2.f)), 512);
  RUN("v_pk_fma_f32, 8 independent chains", (k_valu<0><<<1, 64>>>(out, cyc, 1.f)), 512);
  RUN("v_fma_f32, 8 independent chains", (k_valu<1><<<1, 64>>>(out, cyc, 1.f)), 512);
  RUN("v_pk_fma_f32, dependent", (k_valu<2><<<1, 64>>>(out, cyc, 1.f)), 512);
  RUN("v_exp_f32, 8 independent", (k_valu<3><<<1, 64>>>(out, cyc, 1.f)), 512);
  RUN("ds_bpermute_b32, 8 independent", (k_bperm<0><<<1, 64>>>(out, cyc, 1.f)), 512);
  RUN("ds_bpermute_b32 + add, dependent (latency)", (k_bperm<1><<<1, 64>>>(out, cyc, 1.f)), 512);
  return 0;
}
