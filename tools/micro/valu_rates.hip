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
  const float d0f = (float)d0;
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
      if (MODE == 7) { asm volatile("v_mov_b32 %0, %1" : "=v"(iv) : "v"(iv)); }         // v_mov_b32, ONE dependent chain: a latency, not a price
      // (eight chains each, like the arithmetic above)
      if (MODE == 10) { asm volatile("v_mov_b32 %0, %1" : "=v"(x[j][0]) : "v"(x[j][1])); asm volatile("" : "+v"(x[j][1])); }
      if (MODE == 11) { unsigned r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x[j][0]), "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 12) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 13) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 14) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 15) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(x[j][1]) : ); x[j][0] = __uint_as_float(r); }
      if (MODE == 16) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[j][0]), "v"(x[j][1]) : "vcc"); }
      if (MODE == 18) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 19) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 20) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r) : "v"(x[j][1]) : ); x[j][0] = __uint_as_float(r); }
      if (MODE == 21) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[j][0]) : "v"(x[j][1])); }
      if (MODE == 22) { asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[j][0]) : "v"(x[j][1]), "v"(c[0])); }
      if (MODE == 23) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 24) { asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[j][0]) : "v"(x[j][1])); asm volatile("" : "+v"(x[j][1])); }
      if (MODE == 25) { asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[j][0]) : "v"(x[j][1]), "v"(c[0])); }
      if (MODE == 26) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 27) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j][0]) : "v"(x[j][1])); }
      if (MODE == 28) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j][0]) : "v"(x[j][1]), "v"(c[0])); }
      if (MODE == 29) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 30) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 31) { asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %1" : : "v"(x[j][0]), "v"(x[j][1]) : "s10", "s11"); }
      if (MODE == 32) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r)); x[j][0] = __uint_as_float(r); }
      if (MODE == 33) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 34) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_fma_mixhi_f16 %0, %0, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
      if (MODE == 35) { asm volatile("v_fma_mix_f32 %0, %0, 1.0, %1 op_sel_hi:[0,0,0]" : "+v"(x[j][0]) : "v"(x[j][1])); }
      if (MODE == 36) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j][0]) : "v"(x[j][1])); }
      if (MODE == 37) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[j][0])); }
      if (MODE == 38) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(x[j][1]), "v"(iv)); x[j][0] = __uint_as_float(r); }
      if (MODE == 40) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j][0]) : "v"(x[j][1]), "v"(c[0])); }        // three vector operands
      if (MODE == 41) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j][0]) : "v"(x[j][1]), "s"(a0)); }          // two + a scalar
      if (MODE == 42) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j][0]) : "s"(d0f), "s"(a0)); }              // one + two scalars
      if (MODE == 43) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(r) : "v"(x[j][1]) : ); x[j][0] = __uint_as_float(r); }
      if (MODE == 44) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j][0]) : "v"(x[j][1])); }
      if (MODE == 45) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j][0]) : "v"(x[(j + 1) & 7][1]), "v"(c[0]), "v"(m[1])); }   // no dependent chain at all
      if (MODE == 46) { unsigned r; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(x[(j + 1) & 7][1]), "v"(c[0])); x[j][0] = __uint_as_float(r); }
      if (MODE == 17) { unsigned r = __float_as_uint(x[j][0]); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r) : "v"(x[j][1])); x[j][0] = __uint_as_float(r); }
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
  run<7>("v_mov_b32 (one dependent chain)", 1, out, cyc);
  run<10>("v_mov_b32", 1, out, cyc);
  run<11>("v_cvt_pk_f16_f32", 1, out, cyc);
  run<12>("v_fma_mixlo_f16", 1, out, cyc);
  run<13>("v_max_u32", 1, out, cyc);
  run<14>("v_add_u32", 1, out, cyc);
  run<17>("v_lshl_add_u32", 1, out, cyc);
  run<15>("v_cndmask_b32 (vcc)", 1, out, cyc);
  run<16>("v_cmp_gt_f32 (vcc)", 1, out, cyc);
  run<31>("v_cmp_gt_f32_e64 (SGPR pair)", 1, out, cyc);
  run<20>("v_cndmask_b32_e64 (SGPR pair)", 1, out, cyc);
  run<18>("v_and_b32", 1, out, cyc);
  run<33>("v_and_or_b32", 1, out, cyc);
  run<19>("v_bfi_b32", 1, out, cyc);
  run<26>("v_perm_b32", 1, out, cyc);
  run<32>("v_lshlrev_b32", 1, out, cyc);
  run<38>("v_add3_u32", 1, out, cyc);
  run<29>("v_mad_u32_u24", 1, out, cyc);
  run<30>("v_mul_lo_u32", 1, out, cyc);
  run<23>("v_max3_u32", 1, out, cyc);
  run<21>("v_max_f32", 1, out, cyc);
  run<22>("v_med3_f32", 1, out, cyc);
  run<36>("v_add_f32", 1, out, cyc);
  run<27>("v_sub_f32", 1, out, cyc);
  run<28>("v_fmac_f32", 1, out, cyc);
  run<24>("v_cvt_f32_f16", 1, out, cyc);
  run<25>("v_dot2_f32_f16", 1, out, cyc);
  run<34>("v_fma_mixhi_f16", 1, out, cyc);
  run<35>("v_fma_mix_f32", 1, out, cyc);
  run<37>("v_exp_f32", 1, out, cyc);
  run<40>("v_fma_f32 v, v, v, v", 1, out, cyc);
  run<41>("v_fma_f32 v, v, v, s", 1, out, cyc);
  run<42>("v_fma_f32 v, v, s, s", 1, out, cyc);
  run<45>("v_fma_f32 v, v, v, v (independent)", 1, out, cyc);
  run<44>("v_mul_f32 v, v, v", 1, out, cyc);
  run<43>("v_cndmask_b32_e64 (vcc named)", 1, out, cyc);
  run<46>("v_cndmask_b32 (vcc, independent)", 1, out, cyc);
  return 0;
}
