// Fused LETKF analysis from tile lists and split records, TWO wavefronts per tile -- for unions of more than 32 slots
// (config 4: k = 80, ~63 local observations, 80 slots; 2-D meshes).  Same mathematics, formats and products as
// letkf_tile2.hip (reference: core/etkf.py:57-103 + interface/wrapper.py:86-98 + base.py:257-278), other distribution.
//
// One wavefront holding such a tile needs 456 registers (A fragments of the Gram matrix for five row blocks, five
// recurrence vectors of five blocks each): one wave per SIMD, the spill traffic through the accumulation registers a third
// of its instructions, and within one wave every recurrence step is a chain product -> update -> split -> product with
// nothing beside it (profiles/r03_c4_pmc.json: matrix pipe 26 %, vector unit 40 %).  Here a workgroup of two waves shares
// the tile's record image in LDS and splits the union's ROW BLOCKS: wave w keeps the Gram fragments G[t][.], the recurrence
// vectors and the accumulators of its own row blocks t only (three and two of five), so each fits 256 registers without
// spills and two workgroups' waves share a SIMD.  Per recurrence step a wave updates its rows, splits them into half pairs
// and publishes the 8-byte half fragments in LDS; after ONE workgroup barrier (the fragments are double-buffered) both
// waves read the complete right-hand side (16 columns x U slots) and multiply it with their own rows of G.  The spectral
// bound, the vectors' scale and x' w_mean are the only other things that cross between the waves (one barrier each).
// Sums run in the order of letkf_tile2.hip except x' w_mean (two partial sums): results agree to rounding (6e-8).
// Measured (tools/pair_ab.py, 1e5 points): config 4 0.186 -> 0.172 ms, the 316 x 316 mesh (k = 40, 64 slots) 0.092 -> 0.073 ms;
// a workgroup lives 49k cycles (tools/t2p_stamps.py: records 7.0k, Gram 6.3k, bound + degree 6.7k, 18 exchanges 22.6k, output
// 3.8k) against ~70k for the single wave -- two waves finish a tile 1.4 x sooner, not 2 x: every step is still the chain
// update -> split -> LDS -> barrier -> LDS -> 9 dependent MFMAs, now with a barrier in it.
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#ifdef MIA_T2P_STAMPS
namespace mia {
constexpr int kPStampN = 12, kPStampTiles = 8192;
__device__ long long g_t2p_stamps[kPStampTiles * kPStampN];
}
#define P_STAMP(i) do { if (tid == 0 && bid < kPStampTiles) g_t2p_stamps[bid * kPStampN + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define P_STAMP(i) do { } while (0)
#endif
#ifndef T2P_FRAG_BUFS
#define T2P_FRAG_BUFS 1
#endif
#ifndef T2P_WAVES
#define T2P_WAVES 2
#endif
namespace mia {

// x (4 values) -> hi = f16(x), lo = f16(x - hi): the half of split8_tied that one row block of a 32-slot fragment needs
__device__ __forceinline__ void split4_tied(const float (&x)[4], f2w& hi, f2w& lo) {
  typedef unsigned u2w __attribute__((ext_vector_type(2)));
  u2w hu, lu;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const f2w v = {x[2 * i], x[2 * i + 1]};
    const h2v a = __builtin_convertvector(v, h2v);
    hu[i] = __builtin_bit_cast(unsigned, a);
    lu[i] = hu[i];
  }
  asm("v_fma_mixlo_f16 %0, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %1, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %1, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "+v"(lu[0]), "+v"(lu[1])
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
  hi = __builtin_bit_cast(f2w, hu);
  lo = __builtin_bit_cast(f2w, lu);
}

// MROWS = false (the only instantiation launched): one state row per grid point, straight-line code -- 212 registers at UT = 5,
// KT = 5, two workgroups' waves per SIMD; the row loop of MROWS = true carries 382
// NW = wavefronts per tile (2 or 3): wave w owns row blocks [w OWN, (w + 1) OWN) and output member blocks [w J0, (w + 1) J0)
// Products behind wave-uniform branches (`o < n_own`): the compiler pads the wait states between a matrix instruction and the first
// vector read of its result on the fall-through side of a branch only (tools/check_mfma_hazards.py found reads six
// and seven states behind a skipped block).  Sixteen explicit wait states, tied to the accumulators on both sides so that neither
// the products nor the reads move across them.
template <int N>
__device__ __forceinline__ void mfma_settle(f4w* z) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(z[i]));
  asm volatile("s_nop 7\n\ts_nop 7");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(z[i]));
}

template <int UT, int KT, bool MROWS, int NW>
__global__ __launch_bounds__(64 * NW, MROWS ? 1 : (NW == 2 ? T2P_WAVES : NW))
void letkf_tile2p_kernel(Tile2Params P) {
  constexpr int UMAX = 16 * UT, NB = (KT + 1) / 2, NKB = (UT + 1) / 2;
  constexpr int OWN = (UT + NW - 1) / NW;           // row blocks of the union per wave (the last wave may hold fewer)
  constexpr int J0 = (KT + NW - 1) / NW;            // member blocks of the output per wave
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, h = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: what depends on it is scalar branches and selects)
  const int k = P.k, nc8 = P.nc8;
  const unsigned IMG = (unsigned)(UT * nc8) * 512u;
  unsigned char* zline = smem + IMG;                         // 512 zero bytes
  int* ukey = reinterpret_cast<int*>(smem + IMG + 512);      // [UMAX]
  float* wdl = reinterpret_cast<float*>(ukey + UMAX);        // [UMAX]
  float* El = wdl + UMAX;                                    // [UMAX]
  float* xch = El + UMAX;                                    // [NW waves][3][16]: per-column scalars that cross between the waves
  unsigned char* frag = reinterpret_cast<unsigned char*>(xch + NW * 3 * 16);      // [buffers][NKB][hi / lo][64 lanes] 16 bytes
  const int t_lo = wv * OWN, n_own = UT - t_lo < 0 ? 0 : (UT - t_lo < OWN ? UT - t_lo : OWN);
  const int j_lo = wv * J0, n_out = KT - j_lo < 0 ? 0 : (KT - j_lo < J0 ? KT - j_lo : J0);

  const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int64_t ntile = (P.ng + 15) >> 4;
  if (bid >= ntile) return;
  if (P.clr_counts) {
    const int ncl = *P.clr_n;
    for (int64_t i = bid * NT + tid; i < ncl; i += ntile * NT) P.clr_counts[i] = 0;
    if (bid == 0 && tid == 0) {
      const unsigned e = *P.clr_err;
      if (e) { atomicOr(P.err_out, (int)(e << 3)); *P.clr_err = 0u; }
    }
  }
  const int64_t q8 = ntile >> 3, r8 = ntile & 7, xcd = bid & 7;
  const int64_t tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t p0 = tile << 4;
  const int npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
  int64_t oc0 = P.o0 + p0;
  if (P.seg_len > 0) {
    const unsigned sgi = (unsigned)p0 / (unsigned)P.seg_len;
    oc0 = p0 - (int64_t)sgi * P.seg_len;
    P.Xa += (int64_t)sgi * P.seg_stride;
  }
  const unsigned ldxb = (unsigned)P.ldx * 4u, ldob = (unsigned)P.ldo * 4u;
  const int lrc = lr < npts ? lr : npts - 1;
  const bool colok = lr < npts;

  P_STAMP(0);
  if (tid == 0 && bid < 8192) {
#ifdef MIA_T2P_STAMPS
    g_t2p_stamps[bid * kPStampN + 10] = (long long)__builtin_amdgcn_s_memrealtime();
    g_t2p_stamps[bid * kPStampN + 9] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
  }
  // ---- first round trip: header, slot table, sqrt(rho) matrix (every wave all of it), the first state row
  const int4 hd = P.thdr[tile];
  for (int s = tid; s < UMAX; s += NT) ukey[s] = t2_ld<int32_t>(P.tidx + tile * UMAX, (unsigned)s * 4u);
  f4w dreg[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) dreg[t] = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
  const int sg = 2 * (h & 1) + (h >> 1);
  auto load_xs = [&](int mi, float (&xr)[NB][8]) {
    const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int m0 = 8 * (4 * b + sg);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mem = m0 + i;
        xr[b][i] = t2_ld<float>(xbase, (unsigned)(mem < k ? mem : k - 1) * ldxb + (unsigned)lrc * 4u);
      }
    }
  };
  float xsb[NB][8];
  load_xs(0, xsb);
  for (int i = tid; i < 32; i += NT) reinterpret_cast<f4w*>(zline)[i] = f4w{0.f, 0.f, 0.f, 0.f};
  // (the half fragments a missing row block would have written -- UT odd -- stay zero for the whole launch)
  for (int i = tid; i < T2P_FRAG_BUFS * NKB * 2 * 64; i += NT) reinterpret_cast<f4w*>(frag)[i] = f4w{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int U = __builtin_amdgcn_readfirstlane(hd.x);
  if (U < 0) {                     // the union of this tile did not fit its slots: loud failure, never a truncated analysis
    if (wv == 0) {
      if (colok && h == 0) P.flags[p0 + lr] = MIA_FLAG_OVERFLOW;
      const float nanv = __builtin_nanf("");
      if (colok)
        for (int it = h; it < P.m * k; it += 4) P.Xa[(int64_t)it * P.ldo + oc0 + lr] = nanv;
    }
    return;
  }
  P_STAMP(1);
  // ---- second round trip: the union's records straight into the LDS image (the waves take alternate kilobytes), tails.
  //      As in letkf_tile2_kernel.h (round 5): every lane's pieces belong to at most 2 UT records, whose 32-bit byte offsets are formed
  //      once; with the record's chunk count (2 KT - 1 or 2 KT) and the wave's index as compile-time constants behind wave-uniform
  //      branches the (row block, chunk) of every piece line is a constant too -- as run-time values every one of the thirteen loads
  //      of a wave cost 45 vector instructions, fifteen of them compares and selects (a sixth of the kernel's instructions).
  {
    const int g = lane >> 4, hl = g & 1, gh = g >> 1;
    unsigned roff[UT][2];
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int idx = ukey[16 * t + ((lr - 8 * par) & 15)];
        roff[t][par] = (unsigned)(idx < 0 ? (int)P.zero_rec : idx) * (unsigned)P.rb + 16u * (unsigned)hl;
      }
    auto gather = [&](auto nc8c, auto wvc) {
      constexpr int NC8 = decltype(nc8c)::value, WV = decltype(wvc)::value;
      constexpr int NL = (UT * NC8 + 1) / 2;
#pragma unroll
      for (int u = WV; u < NL; u += NW) {
        const int lA = 2 * u, lB = 2 * u + 1;
        const int tA = lA / NC8, cA = lA % NC8, tB = (lB / NC8) < UT ? lB / NC8 : UT - 1, cB = lB % NC8;
        const unsigned offA = roff[tA][cA & 1] + 32u * (unsigned)cA, offB = roff[tB][cB & 1] + 32u * (unsigned)cB;
        const unsigned off = gh ? offB : offA;
        const bool valid = 2 * u + gh < UT * NC8;
        if (valid)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(P.rec + off),
                                           (__attribute__((address_space(3))) void*)(smem + u * 1024), 16, 0, 0);
      }
    };
    auto gather_w = [&](auto nc8c) {
      if (wv == 0) gather(nc8c, std::integral_constant<int, 0>{});
      else if (NW < 3 || wv == 1) gather(nc8c, std::integral_constant<int, 1>{});
      else gather(nc8c, std::integral_constant<int, (NW > 2 ? 2 : 1)>{});
    };
    if (nc8 == 2 * KT - 1) gather_w(std::integral_constant<int, 2 * KT - 1>{});
    else gather_w(std::integral_constant<int, 2 * KT>{});
  }
  bool badrec = false;
  for (int s = tid; s < UMAX; s += NT) {
    const int idx = ukey[s];
    const int64_t j = idx < 0 ? P.zero_rec : (int64_t)idx;
    const f2w tl = *reinterpret_cast<const f2w*>(P.rec + j * P.rb + 32 * nc8);
    wdl[s] = tl[0];
    El[s] = tl[1];
    badrec = badrec || !(tl[1] == tl[1]);
  }
  auto frag_off = [&](int t, int b) -> unsigned {
    const int c = 4 * b + sg;
    const unsigned col = (unsigned)((lr + 8 * (c & 1)) & 15) * 16u;
    return c < nc8 ? (unsigned)(t * nc8 + c) * 512u + col : IMG + col;
  };
  auto split_x = [&](float (&xs_)[NB][8], float& xm, float& inv_sx, h8v (&xh)[NB], h8v (&xl)[NB]) {
    t2_split_x<NB, true>(xs_, colok, sg, k, P.inv_k, xm, inv_sx, xh, xl);      // (shared with letkf_tile2_kernel.h: the same bits)
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // a tile with a non-finite record: every point goes to the eigensolver kernel (see letkf_tile2.hip)
  const int anybad = __syncthreads_or(badrec ? 1 : 0);
  if (anybad) {
    if (wv == 0 && colok && h == 0) { P.flags[p0 + lr] = MIA_FLAG_RETRY; atomicAdd(P.retry_count, 1); }
    return;
  }

  P_STAMP(2);
  // ---- Gram matrix: this wave's COLUMN blocks G[t1][own t2] (by symmetry the A fragments of its own rows), own rows of Z
  h8v GAh[OWN][NKB], GAl[OWN][NKB];
  {
    f4w G[UT][OWN];
#pragma unroll
    for (int t1 = 0; t1 < UT; ++t1)
#pragma unroll
      for (int o = 0; o < OWN; ++o) G[t1][o] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      h8v ah[UT], al[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const unsigned off = frag_off(t, b);
        ah[t] = *reinterpret_cast<const h8v*>(smem + off);
        al[t] = *reinterpret_cast<const h8v*>(smem + off + 256);
      }
#pragma unroll
      for (int o = 0; o < OWN; ++o) {
        if (o < n_own) {
          const unsigned offo = frag_off(t_lo + o, b);       // fragment of this wave's row block (read again: no register select)
          const h8v bh_ = *reinterpret_cast<const h8v*>(smem + offo), bl_ = *reinterpret_cast<const h8v*>(smem + offo + 256);
#pragma unroll
          for (int t1 = 0; t1 < UT; ++t1) G[t1][o] = t2_mfma3(G[t1][o], ah[t1], al[t1], bh_, bl_);
        }
      }
    }
    mfma_settle<UT * OWN>(&G[0][0]);
    // D_hat = D E for every row block (the Gershgorin products and the right-hand sides need all of them)
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const f4w e4 = *reinterpret_cast<const f4w*>(El + 16 * t + 4 * h);
      dreg[t] *= e4;
    }
    // A fragments of this wave's rows: GA[o][kb] = (G[2 kb][t_lo + o], G[2 kb + 1][t_lo + o]) in the result layout
#pragma unroll
    for (int o = 0; o < OWN; ++o)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float gv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) gv[4 * tt + q] = 2 * kb + tt < UT ? G[2 * kb + tt < UT ? 2 * kb + tt : 0][o][q] * 0x1p-16f : 0.0f;
        split8_tied(gv, GAh[o][kb], GAl[o][kb]);
      }
  }
  P_STAMP(3);
  // this wave's own D_hat rows: loaded once more by address (cache-hot) -- picking them out of the register array by the wave's
  // number was a select per value and candidate row block (forty-eight at config 4, each the issue time of three multiply-adds)
  f4w down[OWN];
#pragma unroll
  for (int o = 0; o < OWN; ++o) {
    const int t = o < n_own ? t_lo + o : 0;
    const f4w d4 = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
    const f4w e4 = *reinterpret_cast<const f4w*>(El + 16 * t + 4 * h);
    down[o] = o < n_own ? d4 * e4 : f4w{0.f, 0.f, 0.f, 0.f};
  }
  // ---- Gershgorin bound over this wave's rows, largest |u_0| of its rows; the other wave's through LDS
  float alpha = 0.0f;
  int deg = 0, tab_idx = 0, degmax = 0, pflag = 0;
  bool decl = false;
  {
    f4w R[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) R[o] = f4w{0.f, 0.f, 0.f, 0.f};
    unsigned dmx = 0u;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const unsigned a = __float_as_uint(dreg[t][q]); dmx = a > dmx ? a : dmx; }
    dmx = t2_wave_max_u32(dmx);
    int esd;
    const float sd = pow2_scale(dmx, 0, &esd);
    const float inv_sd = __uint_as_float((unsigned)(127 - esd) << 23);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float dv[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[4 * tt + q] = 2 * kb + tt < UT ? dreg[2 * kb + tt < UT ? 2 * kb + tt : 0][q] * sd : 0.0f;
      const h8v dh = hi8(dv);
#pragma unroll
      for (int o = 0; o < OWN; ++o) {
        u4w ag = __builtin_bit_cast(u4w, GAh[o][kb]);
        ag &= 0x7fff7fffu;
        R[o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, ag), dh, R[o], 0, 0, 0);
      }
    }
    float L = 0.0f;
#pragma unroll
    for (int o = 0; o < OWN; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // (a plain maximum: a record that is not finite sent the tile to the eigensolver kernel above, nothing here is NaN)
        L = __builtin_fmaxf(L, down[o][q] * R[o][q]);
      }
    L = __uint_as_float(t2_max_h(__float_as_uint(L))) * inv_sd;
    if (h == 0) xch[(wv * 3 + 0) * 16 + lr] = L;
  }

  // ---- per state row
  const int tq_ = (lane & 15) >> 2, tp_ = lane & 3;
  auto frag_ptr = [&](int buf, int kb, int hl) -> unsigned char* { return frag + ((((buf * NKB + kb) * 2 + hl) * 64 + lane) << 4); };
  int fbuf = 0;
  // publish this wave's rows of a vector as half fragments, meet the other wave, read the complete right-hand side
  auto exchange = [&](const f4w (&tv)[OWN], h8v (&bh)[NKB], h8v (&bl)[NKB]) {
#pragma unroll
    for (int o = 0; o < OWN; ++o)
      if (o < n_own) {
        const int t = t_lo + o;
        float x4[4] = {tv[o][0], tv[o][1], tv[o][2], tv[o][3]};
        f2w hi, lo;
        split4_tied(x4, hi, lo);
        *reinterpret_cast<f2w*>(frag_ptr(fbuf, t >> 1, 0) + 8 * (t & 1)) = hi;
        *reinterpret_cast<f2w*>(frag_ptr(fbuf, t >> 1, 1) + 8 * (t & 1)) = lo;
      }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bh[kb] = *reinterpret_cast<const h8v*>(frag_ptr(fbuf, kb, 0));
      bl[kb] = *reinterpret_cast<const h8v*>(frag_ptr(fbuf, kb, 1));
    }
#if T2P_FRAG_BUFS == 2
    fbuf ^= 1;
#else
    __syncthreads();        // (one fragment buffer: nobody writes the next right-hand side before both waves have read this one)
#endif
  };
  for (int mi = 0; mi < (MROWS ? P.m : 1); ++mi) {
    float xm, inv_sx;
    f4w Z[OWN];
    {
      if (mi > 0) load_xs(mi, xsb);
      h8v xh[NB], xl[NB];
      split_x(xsb, xm, inv_sx, xh, xl);
#pragma unroll
      for (int o = 0; o < OWN; ++o) Z[o] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int o = 0; o < OWN; ++o)
          if (o < n_own) {
            const unsigned off = frag_off(t_lo + o, b);
            const h8v ah = *reinterpret_cast<const h8v*>(smem + off), al = *reinterpret_cast<const h8v*>(smem + off + 256);
            Z[o] = t2_mfma3(Z[o], ah, al, xh[b], xl[b]);
          }
      mfma_settle<OWN>(Z);
    }
    // u_0 = D^2 o Z of this wave's rows; its largest magnitude per column joins the spectral bound in the exchange
    f4w va[OWN], vb[OWN], aphi[OWN], apsi[OWN], ad2[OWN];
    {
      unsigned zmax = 0u;
#pragma unroll
      for (int o = 0; o < OWN; ++o) {
        const f4w d2 = down[o] * down[o];
        va[o] = Z[o] * d2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned a = __float_as_uint(va[o][q]) & 0x7fffffffu;
          zmax = a > zmax ? a : zmax;
        }
      }
      zmax = t2_max_h(zmax);
      if (h == 0) xch[(wv * 3 + 1) * 16 + lr] = __uint_as_float(zmax);
    }
    __syncthreads();
    float inv_s2;
    {
      if (mi == 0) {          // interval and degree of every point: shared by all rows
        float L = 0.0f;
        bool lnan = false;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const float lw = xch[(w * 3 + 0) * 16 + lr];
          lnan = lnan || lw != lw;
          L = fmaxf(L, lw);
        }
        if (lnan) L = __builtin_nanf("");
        L = fmaxf(L, 1e-37f) * 1.002f;
        if (!(L == L) || !(fabsf(L) < 1e30f)) { pflag |= MIA_FLAG_NONFINITE; L = 1.0f; }
        tab_idx = (int)ceilf(float(kTabPerOctave) * (__builtin_amdgcn_logf(L * P.inv_reg) + 16.0f)) + kTabIdx0;
        tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
        const int2 th = t2_ld<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
        deg = th.x;
        alpha = __builtin_ldexpf(__int_as_float(th.y) * P.inv_reg, 16);
        decl = colok && (deg > P.dmax || deg > kTabDeg - 1);
        if (decl && h == 0 && wv == 0) {
          P.flags[p0 + lr] = MIA_FLAG_RETRY;
          atomicAdd(P.retry_count, 1);
        }
        degmax = (int)wave_max_nonneg_dpp((colok && !decl) ? float(deg) : 0.0f);
      }
      unsigned zall = 0u;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const unsigned zw = __float_as_uint(xch[(w * 3 + 1) * 16 + lr]);
        zall = zw > zall ? zw : zall;
      }
      int es2;
      const float s2 = pow2_scale(zall, 8, &es2);
      inv_s2 = __uint_as_float((unsigned)(127 - es2) << 23);
#pragma unroll
      for (int o = 0; o < OWN; ++o) {
        va[o] *= s2;
        ad2[o] = alpha * (down[o] * down[o]);
      }
    }
    if (mi == 0) P_STAMP(4);
    const unsigned cbase = (unsigned)tab_idx * (unsigned)(kTabDeg * 8);
    auto coef = [&](int j) -> float2 { return t2_ld<float2>(P.tab_c, cbase + (unsigned)(j < kTabDeg ? j : kTabDeg - 1) * 8u); };
    const float2 c0 = coef(0), c1 = coef(1);
    float2 cn0 = coef(2), cn1 = coef(3);
    f4w y[OWN];
    auto product = [&](const f4w (&tv)[OWN]) {
      h8v bh[NKB], bl[NKB];
      exchange(tv, bh, bl);
#pragma unroll
      for (int o = 0; o < OWN; ++o) y[o] = t2_mfma3(f4w{0.f, 0.f, 0.f, 0.f}, GAh[o][0], GAl[o][0], bh[0], bl[0]);
#pragma unroll
      for (int kb = 1; kb < NKB; ++kb)
#pragma unroll
        for (int o = 0; o < OWN; ++o) y[o] = t2_mfma3(y[o], GAh[o][kb], GAl[o][kb], bh[kb], bl[kb]);
    };
    auto advance = [&](f4w (&vold)[OWN], const f4w (&vcur)[OWN], const float2 cj) {
      product(vcur);
#pragma unroll
      for (int o = 0; o < OWN; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tq = __builtin_fmaf(ad2[o][q], y[o][q], -vcur[o][q]);
          const float vn = __builtin_fmaf(2.0f, tq, -vold[o][q]);
          vold[o][q] = vn;
          aphi[o][q] = __builtin_fmaf(cj.x, vn, aphi[o][q]);
          apsi[o][q] = __builtin_fmaf(cj.y, vn, apsi[o][q]);
        }
    };
    product(va);
#pragma unroll
    for (int o = 0; o < OWN; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vq = __builtin_fmaf(ad2[o][q], y[o][q], -va[o][q]);
        vb[o][q] = vq;
        aphi[o][q] = __builtin_fmaf(c1.x, vq, c0.x * va[o][q]);
        apsi[o][q] = __builtin_fmaf(c1.y, vq, c0.y * va[o][q]);
      }
    int j = 2;
    for (; j + 1 <= degmax; j += 2) {
      const float2 cj = cn0, cj1 = cn1;
      cn0 = coef(j + 2); cn1 = coef(j + 3);
      advance(va, vb, cj);
      advance(vb, va, cj1);
    }
    if (j <= degmax) advance(va, vb, cn0);
    if (mi == 0) P_STAMP(5);
    // ---- output: x' w_mean (partial sums of the two waves), Xa' = Yw^T (D o Phi) for this wave's member blocks
    const float funs = inv_s2 * inv_sx;
    float zu = 0.0f;
#pragma unroll
    for (int o = 0; o < OWN; ++o)
      if (o < n_own) {
        const f4w w4 = *reinterpret_cast<const f4w*>(wdl + 16 * (t_lo + o) + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) zu = fmaf(w4[q], apsi[o][q], zu);
      }
    zu = t2_add_h(zu);
    if (h == 0) xch[(wv * 3 + 2) * 16 + lr] = zu;
    h8v ph_[NKB], pl_[NKB];
    exchange(aphi, ph_, pl_);          // (its barrier also publishes the partial sums)
    zu = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) zu += xch[(w * 3 + 2) * 16 + lr];
    zu *= P.cs_psi * funs;
    const float mterm = xm + zu;
    const float fo = P.cs_phi * funs;
    const float* xbase = P.X + (int64_t)mi * k * P.ldx + P.g0 + p0;
    float* obase = P.Xa + (int64_t)mi * k * P.ldo + oc0;
    const unsigned xo0 = (unsigned)(4 * h) * ldxb + (unsigned)lrc * 4u;
    const unsigned xolast = (unsigned)(k - 1) * ldxb + (unsigned)lrc * 4u;
    const unsigned olane = (unsigned)(4 * h) * ldob + (unsigned)lr * 4u;
    int pf = 0;
    unsigned amax = 0u;
#pragma unroll
    for (int jo = 0; jo < J0; ++jo) {
      if (jo < n_out) {
        const int tj = j_lo + jo;
        f4w xre;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned o = xo0 + (unsigned)(16 * tj + q) * ldxb;
          o = o < xolast ? o : xolast;
          xre[q] = t2_ld<float>(xbase, o);
        }
        f4w acc = f4w{0.f, 0.f, 0.f, 0.f};
        const int c = 2 * tj + (tp_ >> 1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          s4v a4[2][2];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const int tb = 2 * kb + tt < UT ? 2 * kb + tt : 0;
            const unsigned col = (unsigned)((4 * h + tq_ + 8 * (c & 1)) & 15) * 16u + 8u * (unsigned)(tp_ & 1);
            const unsigned o = (c < nc8 && 2 * kb + tt < UT) ? (unsigned)(tb * nc8 + c) * 512u + col : IMG + col;
            a4[tt][0] = t2_tr_read(smem + o);
            a4[tt][1] = t2_tr_read(smem + o + 256);
          }
          typedef short s8v __attribute__((__vector_size__(8 * sizeof(short))));
          const s8v ahs = __builtin_shufflevector(a4[0][0], a4[1][0], 0, 1, 2, 3, 4, 5, 6, 7);
          const s8v als = __builtin_shufflevector(a4[0][1], a4[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
          acc = t2_mfma3(acc, __builtin_bit_cast(h8v, ahs), __builtin_bit_cast(h8v, als), ph_[kb], pl_[kb]);
        }
        float vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vq[q] = acc[q] * fo + (mterm + P.f0 * (xre[q] - xm));
          // (largest |value| as a bit pattern -- NaN > inf > every finite value -- and ONE comparison at the end, letkf_tile2_kernel.h;
          //  rows beyond the ensemble hold the clamped member k - 1: the same magnitudes)
          const unsigned ab = __float_as_uint(vq[q]) & 0x7fffffffu;
          amax = ab > amax ? ab : amax;
          asm volatile("" : "+v"(vq[q]));        // (formed before the store's predicate: tools/check_mfma_hazards.py)
        }
        if (colok && !decl) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (16 * tj + 4 * h + q < k)
              *reinterpret_cast<float*>(reinterpret_cast<char*>(obase) + (olane + (unsigned)(16 * tj + q) * ldob)) = vq[q];
        }
      }
    }
    if (amax > 0x7149f2cau) pf = MIA_FLAG_NONFINITE;       // |value| > 1e30, infinite or NaN
    if (colok && !decl) pflag |= pf;
  }
  P_STAMP(6);
#ifdef MIA_T2P_STAMPS
  if (tid == 0 && bid < 8192) g_t2p_stamps[bid * kPStampN + 11] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  // flags: non-finite values met by either wave
  {
    const unsigned long long fb = __ballot(pflag != 0);
    const int anyf = ((fb >> lr) & 0x0001000100010001ull) != 0ull ? 1 : 0;
    if (h == 0) xch[(wv * 3 + 0) * 16 + lr] = (float)anyf;
    __syncthreads();
    if (wv == 0 && h == 0 && colok && !decl) {
      bool bad = false;
#pragma unroll
      for (int w = 0; w < NW; ++w) bad = bad || xch[(w * 3 + 0) * 16 + lr] != 0.0f;
      P.flags[p0 + lr] = (bad ? MIA_FLAG_NONFINITE : 0) | (deg << 8);
    }
  }
}

#ifdef MIA_T2P_STAMPS
extern "C" int mia_debug_t2p_stamps(long long* host, int n_tiles) {
  if (n_tiles > kPStampTiles) n_tiles = kPStampTiles;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_t2p_stamps), sizeof(long long) * kPStampN * (size_t)n_tiles);
}
#endif

static size_t tile2p_lds_bytes(int ut, int k, int nw) {
  const int nkb = (ut + 1) / 2;
  return (size_t)ut * split_nc8(k) * 512 + 512 + (size_t)16 * ut * 12 + (size_t)nw * 3 * 16 * 4 + (size_t)T2P_FRAG_BUFS * nkb * 2 * 64 * 16;
}

template <int UT, int KT, bool MROWS, int NW>
static int tile2p_launch_m(const Tile2Params& tp, hipStream_t stream) {
  const size_t lds = tile2p_lds_bytes(UT, tp.k, NW);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  auto kern = letkf_tile2p_kernel<UT, KT, MROWS, NW>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntile = (tp.ng + 15) >> 4;
  const int64_t gx = ntile < 65536 ? ntile : 65536;
  const int64_t gy = (ntile + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
#ifdef MIA_EXPERIMENTS
  if (MIA_EXP_FLAG("MIA_T2P_OCC")) {
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 64 * NW, lds);
    fprintf(stderr, "letkf_tile2p_kernel<%d,%d,%d,%d>: %d workgroups per CU at %zu bytes of LDS\n", UT, KT, (int)MROWS, NW, nb, lds);
  }
#endif
  hipEvent_t& stop = launch_stop_event();
  if (stop) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64 * NW), (unsigned)lds, stream, launch_start_event(), stop, 0, tp);
    stop = nullptr;
    launch_start_event() = nullptr;
  } else {
    kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64 * NW), lds, stream>>>(tp);
  }
  ++tile_launch_count();
  note_analysis_kernel("letkf_tile2p_kernel<%d, %d, %s, %d>", UT, KT, MROWS ? "true" : "false", NW);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// one state row per grid point only: with more rows the loop's carried values cost the second wave per SIMD and the kernel loses
// to one wave per tile (k = 80, m = 4: 0.75 against 0.51 ms per 1e5 points, tools/pair_ab.py)
template <int UT, int KT>
static int tile2p_launch(const Tile2Params& tp, hipStream_t stream) {
  if (tp.m != 1) return MIA_ERR_UNSUPPORTED;
#ifdef MIA_EXPERIMENTS        // (three waves per tile: 168 registers, three per SIMD -- config 4 0.175 against 0.177 ms with two, the mesh's four row
  {                           //  blocks leave the third wave idle: 0.099 against 0.074; not instantiated in the shipped library)
    int nw = 2;
    MIA_EXP_SET(nw, "MIA_T2P_NW", atoi);
    if (nw == 3) return tile2p_launch_m<UT, KT, false, 3>(tp, stream);
  }
#endif
  return tile2p_launch_m<UT, KT, false, 2>(tp, stream);
}

// unions of 33 .. 96 slots (UT = 3 .. 6): MIA_ERR_UNSUPPORTED for every other shape (the caller launches letkf_tile2_kernel)
int tile2p_launch_any(const Tile2Params& tp, int ut, int kt, hipStream_t stream) {
#define MIA_T2P_CASE(UU, KK) if (ut == UU && kt == KK) return tile2p_launch<UU, KK>(tp, stream);
#ifdef MIA_T2P_SINGLE          // (development builds: one instantiation)
  MIA_T2P_CASE(5, 5)
#else
  MIA_T2P_CASE(3, 2) MIA_T2P_CASE(3, 3) MIA_T2P_CASE(3, 4) MIA_T2P_CASE(3, 5) MIA_T2P_CASE(3, 6)
  MIA_T2P_CASE(4, 3) MIA_T2P_CASE(4, 4) MIA_T2P_CASE(4, 5) MIA_T2P_CASE(4, 6)
  MIA_T2P_CASE(5, 4) MIA_T2P_CASE(5, 5) MIA_T2P_CASE(5, 6)
  MIA_T2P_CASE(6, 5) MIA_T2P_CASE(6, 6)
#endif
#undef MIA_T2P_CASE
  return MIA_ERR_UNSUPPORTED;
}

}  // namespace mia
