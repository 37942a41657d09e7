// Issue-rate microbenchmark behind DESIGN.md section 3.0's pipe model (build + run: tools/mfma_rate.sh on the GPU box).
//   How many cycles does one v_mfma_f32_16x16x4_f32 occupy a SIMD, how many a wave64 v_fma_f32 / v_pk_fma_f32, and do the
//   two overlap -- from one wavefront, and from several wavefronts of one SIMD?
// One workgroup; W waves per SIMD (blockDim = 256 * W: waves are dealt round-robin to the four SIMDs of the CU).  Every
// wave runs `iters` trips of a loop body of M independent MFMAs (8 accumulators) and V independent vector FMAs and
// stamps s_memtime / s_memrealtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int M, int V, bool PK, int MT = 0>
__global__ void __launch_bounds__(1024) rate_kernel(int iters, float seed, long long* out, float* sink) {
  float4_t acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = float4_t{seed, seed, seed, seed};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  float2_t pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = float2_t{seed + i, seed - i};
  const float2_t pm = float2_t{1.0001f, 0.9999f}, pa = float2_t{0.001f, 0.002f};
  const float4_t ha = float4_t{seed, a, b, seed}, hb = float4_t{b, seed, a, a};   // (bit patterns only: 8 halves each)
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < (M > V ? M : V); ++j) {
      if (j < M) {
        if constexpr (MT == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(a), "v"(b));   // (volatile: keeps the interleaving)
        else if constexpr (MT == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
        else if constexpr (MT == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(pm), "v"(pa));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
      }
      if (j < V) {
        if constexpr (PK) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j & 7]) : "v"(pm), "v"(pa));
        } else {
          if constexpr (MT == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(b));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 15]) : "v"(b), "v"(a));
        }
      }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + pv[i][0] + pv[i][1];
  for (int i = 0; i < 16; ++i) s += v[i];
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    out[(threadIdx.x >> 6) * 2] = t1 - t0;
    out[(threadIdx.x >> 6) * 2 + 1] = r1 - r0;
  }
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int M, int V, bool PK, int MT = 0>
static void run(const char* what, int waves_per_simd, long long* dout, float* sink) {
  const int iters = 2000, nw = 4 * waves_per_simd;
  rate_kernel<M, V, PK, MT><<<1, 64 * nw>>>(iters, 1.0f, dout, sink);
  rate_kernel<M, V, PK, MT><<<1, 64 * nw>>>(iters, 1.0f, dout, sink);
  std::vector<long long> h(2 * nw);
  if (hipMemcpy(h.data(), dout, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost) != hipSuccess) exit(1);
  long long tmax = 0, rmax = 0;
  for (int w = 0; w < nw; ++w) { if (h[2 * w] > tmax) tmax = h[2 * w]; if (h[2 * w + 1] > rmax) rmax = h[2 * w + 1]; }
  const double per_trip = (double)tmax / iters, us = rmax / 100.0;
  printf("%-44s waves/SIMD %d: %8.1f ticks per trip (%.2f ticks/us) -> per wave and instruction: %s %.2f\n", what,
         waves_per_simd, per_trip, tmax / us, M && V ? "mfma+valu pair" : (M ? "mfma" : "valu"),
         per_trip / waves_per_simd / (M > V ? M : V));
}

int main() {
  long long* dout; float* sink;
  if (hipMalloc(&dout, 4096) != hipSuccess || hipMalloc(&sink, 8192) != hipSuccess) return 1;
  for (int w = 1; w <= 4; w += (w == 1 ? 1 : 2) ) {
    run<8, 0, false>("8 independent mfma_16x16x4_f32", w, dout, sink);
    run<0, 16, false>("16 independent v_fma_f32", w, dout, sink);
    run<0, 8, true>("8 independent v_pk_fma_f32", w, dout, sink);
    run<8, 8, false>("8 mfma interleaved with 8 v_fma_f32", w, dout, sink);
    run<8, 16, false>("8 mfma interleaved with 16 v_fma_f32", w, dout, sink);
    run<8, 8, true>("8 mfma interleaved with 8 v_pk_fma_f32", w, dout, sink);
    run<8, 0, false, 1>("8 independent mfma_f32_16x16x32_f16", w, dout, sink);
    run<8, 0, false, 3>("8 independent mfma_f32_16x16x32_bf16", w, dout, sink);
    run<8, 0, false, 2>("8 independent mfma_f32_16x16x16_f16", w, dout, sink);
    run<8, 8, false, 1>("8 f16 mfma interleaved with 8 v_fma_f32", w, dout, sink);
    run<8, 16, false, 1>("8 f16 mfma interleaved with 16 v_fma_f32", w, dout, sink);
    run<8, 8, true, 1>("8 f16 mfma interleaved with 8 v_pk_fma_f32", w, dout, sink);
    run<0, 16, false, 4>("16 independent v_cvt_pk_f16_f32", w, dout, sink);
  }
  return 0;
}
