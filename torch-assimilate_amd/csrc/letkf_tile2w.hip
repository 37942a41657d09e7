// The WEIGHTS of the LETKF (what LETKF.estimate_weights returns: interface/letkf.py:127-146, core/etkf.py:57-103 --
// W[g][i][j] = w_mean_i + W_pert_ij) on the tile route: same tile lists, split records, Gram matrix, interval / degree
// and coefficient table as letkf_tile2.hip, but the Chebyshev recurrence runs on a MATRIX block per grid point.
//
// With D_hat = D E (sqrt(rho) times the records' scales), Ghat the Gram matrix of the union's normalised records and
// X_g = alpha D_hat^2 Ghat - I (the affine image of the local matrix S_g = D_hat Ghat D_hat, similar to it):
//
//   W_pert = f0 I + cs_phi Yhat M_g Yhat^T,      M_g = Phi(X_g) D_hat^2 = D_hat phi(S_g) D_hat   (symmetric, U x U)
//   w_mean = cs_psi Yhat Psi(X_g) (D_hat^2 o wdl)                                               (wdl: innovation in the record's scale)
//
// A workgroup of FOUR wavefronts serves a tile, each wavefront four of its points.  What belongs to the tile -- record image,
// Gram matrix and its fragments, interval / degree of the sixteen points, phase A -- is made ONCE, by wave 0, and handed
// over in LDS (round 4: every quarter made it again, a quarter of the kernel's vector instructions; the kernel is bound by
// vector issue, so three wavefronts waiting at a barrier cost nothing that counts):
//   phase A  the vector recurrence of letkf_tile2.hip on the sixteen columns u_0 = D_hat^2 o wdl (one per point) -> w_mean
//            of all sixteen points from one output product (LDS)
//   phase B  per point: V_0 = D_hat^2 (diagonal), V_{j+1} = 2 (alpha D_hat^2 o (Ghat V_j) - V_j) - V_{j-1} on two column
//            blocks of sixteen -- 12 MFMAs per step at U <= 32 -- accumulating c_j V_j = M_g in the result layout, which by
//            symmetry IS the A-operand layout of the next product: P = M_g Yhat^T (transposed LDS reads of the records as
//            B fragments), W' = Yhat P (the same fragments as A operands); rows of 16 consecutive floats are stored.
// Every product is the split-precision MFMA triple of the analysis kernel; entries stay inside the half-precision range by
// powers of two folded into the coefficients (2^-10) and the split of P (2^-5).  Declined points (MIA_FLAG_RETRY from the
// analysis launch that precedes this one -- same decision, same code) are left untouched for mia_letkf_weights_retry_f32.
#include "mia_common.h"
#include <hip/hip_ext.h>
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_tiles.h"
#include <type_traits>

namespace mia {

struct Tile2wParams {
  int k; int64_t ng;
  const unsigned char* rec; int rb, nc8; int64_t zero_rec;
  const int4* thdr; const int32_t* tidx; const f4w* tD;
  float inv_reg, f0, cs_phi, cs_psi;
  float* W; int32_t* flags; int32_t* retry_count;
  int dmax;
  const int2* tab_hdr; const float2* tab_c;
};

constexpr int kWPts = 4;        // points of a tile per wavefront
// The matrix recurrence accumulates rounding over 32 columns and `degree` steps: beyond this degree a point is handed to the
// float64 eigensolver as well (MIA_FLAG_RETRY is added to the flag the analysis launch wrote, the point is counted): the random
// sweep of tools/stress_tile.py peaked at 7.4e-6 for degrees in the fifties, 3e-6 below the cap
constexpr int kWDegCap = 36;
#ifndef MIA_W_TRIM
#define MIA_W_TRIM 2
#endif

// K4: the ensemble size is a multiple of four (rows of W are stored sixteen bytes at a time; the predicates and addresses of the
// element-wise stores are the other instantiation's)
template <int UT, int KT, bool K4>
__global__ __launch_bounds__(64 * (16 / kWPts), KT <= 3 ? 3 : 2)
void letkf_tile2w_kernel(Tile2wParams P) {
  constexpr int UMAX = 16 * UT, NB = (KT + 1) / 2, NKB = (UT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, lr = lane & 15, h = lane >> 4;
  const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // this wave's quarter of the tile
  const int k = P.k, nc8 = P.nc8;
  const unsigned IMG = (unsigned)(UT * nc8) * 512u;
  unsigned char* zline = smem + IMG;
  int* ukey = reinterpret_cast<int*>(smem + IMG + 512);      // [UMAX]
  float* wdl = reinterpret_cast<float*>(ukey + UMAX);        // [UMAX]
  float* El = wdl + UMAX;                                    // [UMAX]
  float* Dl = El + UMAX;                                     // [16][UMAX] D_hat of the tile's points
  float* wbl = Dl + 16 * UMAX;                               // [16][16 KT] w_mean of the tile's points
  int4* ptl = reinterpret_cast<int4*>(wbl + 16 * 16 * KT);   // [16] per point: degree, table row, alpha, declined
  f4w* gal = reinterpret_cast<f4w*>(ptl + 16);               // [UT][NKB][2][64] 2^-16 Ghat in the A-fragment order of the lanes (f32)
  int* stl = reinterpret_cast<int*>(gal + UT * NKB * 2 * 64);   // [4] tile without weights (a record that is not finite)

  const int64_t tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int64_t ntile = (P.ng + 15) >> 4;
  if (tile >= ntile) return;
  const int64_t p0 = tile << 4;
  const int npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
  const bool colok = lr < npts;
  const int64_t kk = (int64_t)k * k;

  const int4 hd = P.thdr[tile];
  const int U = __builtin_amdgcn_readfirstlane(hd.x);
  if (U < 0) {                     // union overflow (flagged by the analysis launch): no weights either
    const float nanv = __builtin_nanf("");
    for (int i = 0; i < kWPts; ++i) {
      const int g = kWPts * sub + i;
      if (g < npts)
        for (int64_t e = lane; e < kk; e += 64) P.W[(p0 + g) * kk + e] = nanv;
    }
    return;
  }
  h8v GAh[UT][NKB], GAl[UT][NKB];
  const int sg = 2 * (h & 1) + (h >> 1);
  auto frag_off = [&](int t, int b) -> unsigned {
    const int c = 4 * b + sg;
    const unsigned col = (unsigned)((lr + 8 * (c & 1)) & 15) * 16u;
    return c < nc8 ? (unsigned)(t * nc8 + c) * 512u + col : IMG + col;
  };
  // (split8_tied outside the per-point recurrence: see mia_tiles.h)
  auto rhs_split = [&](const f4w (&tv)[UT], int kb, const float sc, h8v& bh, h8v& bl) {
    float bv[8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tk = 2 * kb + tt < UT ? 2 * kb + tt : 0;
        bv[4 * tt + q] = 2 * kb + tt < UT ? tv[tk][q] * sc : 0.0f;
      }
    split8_tied(bv, bh, bl);
  };
  auto rhs_split1 = [&](const f4w (&tv)[UT], int kb, h8v& bh, h8v& bl, auto tied) {
    float bv[8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tk = 2 * kb + tt < UT ? 2 * kb + tt : 0;
        bv[4 * tt + q] = 2 * kb + tt < UT ? tv[tk][q] : 0.0f;
      }
    if constexpr (decltype(tied)::value) split8_tied(bv, bh, bl); else split8(bv, bh, bl);
  };
  using tied_t = std::integral_constant<bool, true>;
  using fresh_t = std::integral_constant<bool, false>;
  // y = Ghat tv (one column block)
  auto product = [&](const f4w (&tv)[UT], f4w (&y)[UT], auto tied) {
    {
      h8v bh, bl;
      rhs_split1(tv, 0, bh, bl, tied);
#pragma unroll
      for (int t = 0; t < UT; ++t) y[t] = t2_mfma3(f4w{0.f, 0.f, 0.f, 0.f}, GAh[t][0], GAl[t][0], bh, bl);
    }
#pragma unroll
    for (int kb = 1; kb < NKB; ++kb)
      if (32 * kb < U) {
        h8v bh, bl;
        rhs_split1(tv, kb, bh, bl, tied);
#pragma unroll
        for (int t = 0; t < UT; ++t) y[t] = t2_mfma3(y[t], GAh[t][kb], GAl[t][kb], bh, bl);
      }
  };
  // fragments of the records read transposed: slots 16 (2 kb + tt) + 4 h + q for member 16 tj + lr -- the A operand of
  // Yhat (.) and the B operand of (.) Yhat^T alike
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  auto yfrag = [&](int tj, int kb, h8v& yh, h8v& yl) {
    const int c = 2 * tj + (tp >> 1);
    s4v a4[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int tb = 2 * kb + tt < UT ? 2 * kb + tt : 0;
      const unsigned col = (unsigned)((4 * h + tq + 8 * (c & 1)) & 15) * 16u + 8u * (unsigned)(tp & 1);
      const unsigned o = (c < nc8 && 2 * kb + tt < UT) ? (unsigned)(tb * nc8 + c) * 512u + col : IMG + col;
      a4[tt][0] = t2_tr_read(smem + o);
      a4[tt][1] = t2_tr_read(smem + o + 256);
    }
    typedef short s8v __attribute__((__vector_size__(8 * sizeof(short))));
    const s8v ahs = __builtin_shufflevector(a4[0][0], a4[1][0], 0, 1, 2, 3, 4, 5, 6, 7);
    const s8v als = __builtin_shufflevector(a4[0][1], a4[1][1], 0, 1, 2, 3, 4, 5, 6, 7);
    yh = __builtin_bit_cast(h8v, ahs);
    yl = __builtin_bit_cast(h8v, als);
  };

  // (which wave: rotating with the tile -- wave i of every workgroup sits on SIMD i, and with the same wave taking the tile's part
  //  everywhere that SIMD sets the pace while the other three idle a quarter of the time)
  const int leader = (int)(((tile >> 3) + (tile >> 8)) & 3);
  if (sub == leader) {              // ======== the tile's part: one wave
  int myidx[(UMAX + 63) / 64];
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int s = lane + 64 * r;
    myidx[r] = s < UMAX ? t2_ld<int32_t>(P.tidx + tile * UMAX, (unsigned)s * 4u) : -1;
  }
  f4w dreg[UT];
#pragma unroll
  for (int t = 0; t < UT; ++t) dreg[t] = t2_ld<f4w>(P.tD + (tile * UT + t) * 64, (unsigned)lane * 16u);
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int s = lane + 64 * r;
    if (s < UMAX) ukey[s] = myidx[r];
  }
  for (int i = lane; i < 32; i += 64) reinterpret_cast<f4w*>(zline)[i] = f4w{0.f, 0.f, 0.f, 0.f};
  MIA_T2_SYNC();
  // the union's records -> LDS image (letkf_tile2.hip), tails
  {
    const int g = lane >> 4, hl = g & 1;
    int tc = g >> 1;
    constexpr int NLmax = (UT * 2 * KT + 1) / 2;
#pragma unroll
    for (int u = 0; u < NLmax; ++u) {
      if (2 * u < UT * nc8) {
        int t = 0, c = tc;
#pragma unroll
        for (int i = 1; i < UT; ++i)
          if (c >= nc8) { c -= nc8; ++t; }
        const bool valid = tc < UT * nc8;
        const int r = 16 * t + ((lr - 8 * (c & 1)) & 15);
        const int idx = valid ? ukey[r] : -1;
        const int64_t j = idx < 0 ? P.zero_rec : (int64_t)idx;
        const unsigned char* src = P.rec + j * P.rb + (32 * c + 16 * hl);
        if (valid)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src),
                                           (__attribute__((address_space(3))) void*)(smem + u * 1024), 16, 0, 0);
      }
      tc += 2;
    }
  }
  f2w tails[(UMAX + 63) / 64];
#pragma unroll
  for (int r = 0; r < (UMAX + 63) / 64; ++r) {
    const int64_t j = myidx[r] < 0 ? P.zero_rec : (int64_t)myidx[r];
    tails[r] = *reinterpret_cast<const f2w*>(P.rec + j * P.rb + 32 * nc8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();

  // ---- Gram matrix, D_hat, A fragments of Ghat, interval and degree of every point: as in letkf_tile2.hip
  float alpha = 0.0f;
  int deg = 0, tab_idx = 0, degmax = 0;
  bool decl = false, nowt = false;
  {
    f4w G[UT][UT];
#pragma unroll
    for (int t1 = 0; t1 < UT; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < UT; ++t2) G[t1][t2] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      h8v ah[UT], al[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const unsigned o = frag_off(t, b);
        ah[t] = *reinterpret_cast<const h8v*>(smem + o);
        al[t] = *reinterpret_cast<const h8v*>(smem + o + 256);
      }
#pragma unroll
      for (int t2 = 0; t2 < UT; ++t2)
#pragma unroll
        for (int t1 = 0; t1 < UT; ++t1) G[t1][t2] = t2_mfma3(G[t1][t2], ah[t1], al[t1], ah[t2], al[t2]);
    }
    bool badrec = false;
#pragma unroll
    for (int r = 0; r < (UMAX + 63) / 64; ++r) {
      const int s = lane + 64 * r;
      if (s < UMAX) {
        wdl[s] = tails[r][0];
        El[s] = tails[r][1];
        badrec = badrec || !(tails[r][1] == tails[r][1]);
      }
    }
    nowt = __any(badrec);                 // (every point of the tile was handed to the eigensolver by the analysis launch)
    MIA_T2_SYNC();
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const f4w e4 = *reinterpret_cast<const f4w*>(El + 16 * t + 4 * h);
      dreg[t] *= e4;
    }
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float gv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) gv[4 * tt + q] = 2 * kb + tt < UT ? G[2 * kb + tt < UT ? 2 * kb + tt : 0][t][q] * 0x1p-16f : 0.0f;
        split8(gv, GAh[t][kb], GAl[t][kb]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) gal[((t * NKB + kb) * 2 + tt) * 64 + lane] = f4w{gv[4 * tt], gv[4 * tt + 1], gv[4 * tt + 2], gv[4 * tt + 3]};
      }
    f4w R[UT];
#pragma unroll
    for (int t = 0; t < UT; ++t) R[t] = f4w{0.f, 0.f, 0.f, 0.f};
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
    for (int kb = 0; kb < NKB; ++kb)
      if (kb == 0 || 32 * kb < U) {     // (kb = 0 unconditionally: no branch between this product and the use of its result)
        float dv[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q) dv[4 * tt + q] = 2 * kb + tt < UT ? dreg[2 * kb + tt < UT ? 2 * kb + tt : 0][q] * sd : 0.0f;
        const h8v dh = hi8(dv);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          u4w ag = __builtin_bit_cast(u4w, GAh[t][kb]);
          ag &= 0x7fff7fffu;
          R[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, ag), dh, R[t], 0, 0, 0);
        }
      }
    float L = 0.0f;
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = dreg[t][q] * R[t][q];
        L = (v > L || v != v) ? v : L;
      }
    L = __uint_as_float(t2_max_h(__float_as_uint(L))) * inv_sd;
    L = fmaxf(L, 1e-37f) * 1.002f;
    if (!(L == L) || !(fabsf(L) < 1e30f)) L = 1.0f;
    tab_idx = (int)ceilf(float(kTabPerOctave) * (__builtin_amdgcn_logf(L * P.inv_reg) + 16.0f)) + kTabIdx0;
    tab_idx = tab_idx < 0 ? 0 : (tab_idx > kTabN - 1 ? kTabN - 1 : tab_idx);
    const int2 th = t2_ld<int2>(P.tab_hdr, (unsigned)tab_idx * 8u);
    deg = th.x;
    alpha = __builtin_ldexpf(__int_as_float(th.y) * P.inv_reg, 16);
    decl = colok && (deg > P.dmax || deg > kTabDeg - 1);
    degmax = (int)wave_max_nonneg_dpp((colok && !decl) ? float(deg) : 0.0f);
  }
  if (lane == 0) stl[0] = nowt ? 1 : 0;
  if (!nowt) {
  // D_hat of the tile's points, by slot (phase B reads them as row factors); the fragments of Ghat and what the table said
#pragma unroll
  for (int t = 0; t < UT; ++t) *reinterpret_cast<f4w*>(Dl + lr * UMAX + 16 * t + 4 * h) = dreg[t];
  if (h == 0) ptl[lr] = int4{deg, tab_idx, __float_as_int(alpha), decl ? 1 : 0};

  // ---- phase A: w_mean of the sixteen points (columns), Psi(X) (D_hat^2 o wdl)
  {
    const unsigned cbase = (unsigned)tab_idx * (unsigned)(kTabDeg * 8);
    auto coef = [&](int j) -> float { return t2_ld<float2>(P.tab_c, cbase + (unsigned)(j < kTabDeg ? j : kTabDeg - 1) * 8u).y; };
    const float c0 = coef(0), c1 = coef(1);
    float cn0 = coef(2), cn1 = coef(3);
    f4w va[UT], vb[UT], apsi[UT], ad2[UT], y[UT];
    float inv_s2;
    {
      unsigned zmax = 0u;
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        const f4w d2 = dreg[t] * dreg[t];
        const f4w w4 = *reinterpret_cast<const f4w*>(wdl + 16 * t + 4 * h);
        ad2[t] = alpha * d2;
        va[t] = w4 * d2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned a = __float_as_uint(va[t][q]) & 0x7fffffffu;
          zmax = a > zmax ? a : zmax;
        }
      }
      zmax = t2_max_h(zmax);
      int es2;
      const float s2 = pow2_scale(zmax, 8, &es2);
      inv_s2 = __uint_as_float((unsigned)(127 - es2) << 23);
#pragma unroll
      for (int t = 0; t < UT; ++t) va[t] *= s2;
    }
    auto advance = [&](f4w (&vold)[UT], const f4w (&vcur)[UT], const float cj) {
      product(vcur, y, tied_t{});
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float tq_ = __builtin_fmaf(ad2[t][q], y[t][q], -vcur[t][q]);
          const float vn = __builtin_fmaf(2.0f, tq_, -vold[t][q]);
          vold[t][q] = vn;
          apsi[t][q] = __builtin_fmaf(cj, vn, apsi[t][q]);
        }
    };
    product(va, y, tied_t{});
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vq = __builtin_fmaf(ad2[t][q], y[t][q], -va[t][q]);
        vb[t][q] = vq;
        apsi[t][q] = __builtin_fmaf(c1, vq, c0 * va[t][q]);
      }
    int j = 2;
    for (; j + 1 <= degmax; j += 2) {
      const float cj = cn0, cj1 = cn1;
      cn0 = coef(j + 2); cn1 = coef(j + 3);
      advance(va, vb, cj);
      advance(vb, va, cj1);
    }
    if (j <= degmax) advance(va, vb, cn0);
    h8v ph_[NKB], pl_[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) rhs_split1(apsi, kb, ph_[kb], pl_[kb], tied_t{});
    const float fo = P.cs_psi * inv_s2;
#pragma unroll
    for (int tj = 0; tj < KT; ++tj) {
      f4w acc = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (kb == 0 || 32 * kb < U) {
          h8v yh, yl;
          yfrag(tj, kb, yh, yl);
          acc = t2_mfma3(acc, yh, yl, ph_[kb], pl_[kb]);
        }
      acc *= fo;
      *reinterpret_cast<f4w*>(wbl + lr * (16 * KT) + 16 * tj + 4 * h) = acc;
    }
  }
  }   // (a tile with weights)
  }   // ======== the tile's part
  __syncthreads();                   // (the only one: nobody leaves before it)
  if (stl[0] != 0 || kWPts * sub >= npts) return;
  // (store addresses: row 16 tj + lr, columns 16 ti + 4 h .. -- one lane offset, the blocks' strides are the wave's or immediates;
  //  only the LAST row / column block can reach beyond k, by the definition of KT)
  const unsigned offb = (unsigned)(lr * k + 4 * h) * 4u;
  const bool rowok_last = 16 * (KT - 1) + lr < k, colok_last = 16 * (KT - 1) + 4 * h < k;

  // ---- phase B: the points of this wave, one after the other
  for (int i = 0; i < kWPts; ++i) {
    const int g = kWPts * sub + i;                 // (wave-uniform)
    if (g >= npts) break;
    const int4 pg = ptl[g];                        // (one address for the wave)
    if (__builtin_amdgcn_readfirstlane(pg.w)) continue;
    const int deg_t = __builtin_amdgcn_readfirstlane(pg.x);
    if (deg_t > kWDegCap) {
      if (lane == 0) { atomicOr(P.flags + p0 + g, MIA_FLAG_RETRY); atomicAdd(P.retry_count, 1); }
      continue;
    }
    // the table's degree carries two steps of margin over the a-priori count (cheb_table_kernel): the MATRIX recurrence, a
    // thirteenth of this kernel's instructions per step, runs without them (tools/stress_tile.py --weights: unchanged worst case)
    const int deg_g = deg_t - MIA_W_TRIM > 3 ? deg_t - MIA_W_TRIM : (deg_t < 3 ? deg_t : 3);
    const float alpha_g = __int_as_float(__builtin_amdgcn_readfirstlane(pg.z));
    const unsigned cbase = (unsigned)__builtin_amdgcn_readfirstlane(pg.y) * (unsigned)(kTabDeg * 8);
    // (one address for the wave; the value is used RAW -- the factor 2^-10 that keeps M inside the half range for P rides on the
    //  sign below -- so that the loads issued two steps ahead are waited for where they are used, not where they are issued)
    auto coef = [&](int j) -> float {
      return t2_ld<float2>(P.tab_c, cbase + (unsigned)(j < kTabDeg ? j : kTabDeg - 1) * 8u).x;
    };
    // The recurrence as ONE matrix per point: V_{j+1} = A' V_j - V_{j-1} with A' = 2 alpha_g D_hat^2 Ghat - 2 I, whose A fragments are
    // formed once per point (rows scaled by this lane's row factor, the diagonal in place) -- then the matrix instruction does the
    // whole step: its accumulator input IS V_{j-1}, carried with signs W_j = sigma_j V_j, sigma = + + - - + + ..., so that the
    // subtraction becomes an addition and the sign that leaves on A' rides in the split of the right-hand side (split8n: source
    // modifiers, no instruction).  Per column block and step 12 (split) + 8 (accumulate c_j V_j) vector instructions where the
    // elementwise form had 44: the kernel was bound by vector issue (82 % busy, profiles/r04_w_pmc.json).
    f4w d2[UT];
    unsigned dmx = 0u;
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const f4w d4 = *reinterpret_cast<const f4w*>(Dl + g * UMAX + 16 * t + 4 * h);
      d2[t] = d4 * d4;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const unsigned a = __float_as_uint(d2[t][q]); dmx = a > dmx ? a : dmx; }
    }
    dmx = t2_wave_max_u32(dmx);
    int es;
    const float s = pow2_scale(dmx, 8, &es);
    const float inv_s = __uint_as_float((unsigned)(127 - es) << 23);
    h8v APh[UT][NKB], APl[UT][NKB];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      const float dr = Dl[g * UMAX + 16 * t + lr];                 // D_hat of row 16 t + lr (A layout: lane = row)
      const float r2 = 2.0f * alpha_g * dr * dr;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        float av[8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const f4w g4 = gal[((t * NKB + kb) * 2 + tt) * 64 + lane];      // 2^-16 Ghat (alpha carries 2^16)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool dg = 2 * kb + tt == t && 4 * h + q == lr;
            av[4 * tt + q] = __builtin_fmaf(r2, g4[q], dg ? -2.0f : 0.0f);
          }
        }
        split8_tied(av, APh[t][kb], APl[t][kb]);
      }
    }
    // V[cb][t][q] = V[slot 16 t + 4 h + q][slot 16 cb + lr]
    f4w va[UT][UT], vb[UT][UT], am[UT][UT];
    const float c0 = coef(0) * 0x1p-10f, c1 = coef(1) * 0x1p-10f;
    float cn0 = coef(2), cn1 = coef(3);
#pragma unroll
    for (int cb = 0; cb < UT; ++cb)
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) va[cb][t][q] = (t == cb && 4 * h + q == lr) ? d2[t][q] * s : 0.0f;
    // acc += (+-) A' tv (one column block); neg: the right-hand side enters negated
    auto product_acc = [&](const f4w (&tv)[UT], f4w (&acc)[UT], auto neg, auto tied) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (kb == 0 || 32 * kb < U) {
          float bv[8];
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int tk = 2 * kb + tt < UT ? 2 * kb + tt : 0;
              bv[4 * tt + q] = 2 * kb + tt < UT ? tv[tk][q] : 0.0f;
            }
          h8v bh, bl;
          if constexpr (decltype(neg)::value) {
            split8n(bv, bh, bl);
          } else {
            if constexpr (decltype(tied)::value) split8_tied(bv, bh, bl); else split8(bv, bh, bl);
          }
#pragma unroll
          for (int t = 0; t < UT; ++t) acc[t] = t2_mfma3(acc[t], APh[t][kb], APl[t][kb], bh, bl);
        }
    };
    // V_1 = X V_0 = A' V_0 / 2 (sigma_0 = sigma_1 = +)
    auto first_block = [&](int cb, auto tied) {
#pragma unroll
      for (int t = 0; t < UT; ++t) vb[cb][t] = f4w{0.f, 0.f, 0.f, 0.f};
      product_acc(va[cb], vb[cb], fresh_t{}, tied);
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float vq = 0.5f * vb[cb][t][q];
          vb[cb][t][q] = vq;
          am[cb][t][q] = __builtin_fmaf(c1, vq, c0 * va[cb][t][q]);
        }
    };
    first_block(0, fresh_t{});
    if constexpr (UT > 1) first_block(1, tied_t{});
    // W_new = (+-) A' W_cur + W_old, in place over W_old; am += (sigma c_j) W_new
    auto advance = [&](f4w (&vold)[UT][UT], const f4w (&vcur)[UT][UT], const float cjs, auto neg) {
#pragma unroll
      for (int cb = 0; cb < UT; ++cb) {
        if (cb == 0) product_acc(vcur[cb], vold[cb], neg, fresh_t{}); else product_acc(vcur[cb], vold[cb], neg, tied_t{});
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) am[cb][t][q] = __builtin_fmaf(cjs, vold[cb][t][q], am[cb][t][q]);
      }
    };
    int j = 2;
    for (; j + 1 <= deg_g; j += 2) {
      const float sg_ = (j & 2) ? -0x1p-10f : 0x1p-10f;    // sigma_j = sigma_{j+1} (j even), times 2^-10
      const float cj = cn0 * sg_, cj1 = cn1 * sg_;
      cn0 = coef(j + 2); cn1 = coef(j + 3);
      advance(va, vb, cj, tied_t{});        // W_j     = -A' W_{j-1} + W_{j-2}   (tied_t = "true": negated right-hand side)
      advance(vb, va, cj1, fresh_t{});      // W_{j+1} =  A' W_j     + W_{j-1}
    }
    if (j <= deg_g) advance(va, vb, cn0 * ((j & 2) ? -0x1p-10f : 0x1p-10f), tied_t{});
    // (kb = 0 runs unconditionally in the products below -- a tile without any observation multiplies zero fragments -- so that the
    //  accumulators start from the matrix instruction's zero operand: four register moves and a branch less per product)
    // ---- P = M Yhat^T: A fragments of M (row block rb) = the registers of its column block rb (M is symmetric)
    h8v mh[UT][NKB], ml[UT][NKB];
#pragma unroll
    for (int rb = 0; rb < UT; ++rb)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) rhs_split1(am[rb], kb, mh[rb][kb], ml[rb][kb], tied_t{});
    const float wsc = P.cs_phi * inv_s * 0x1p15f;
    float* Wg = P.W + (p0 + g) * kk;
    float chk = 0.0f;
#pragma unroll
    for (int tj = 0; tj < KT; ++tj) {
      f4w p1[UT];
#pragma unroll
      for (int rb = 0; rb < UT; ++rb) p1[rb] = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
        if (kb == 0 || 32 * kb < U) {
          h8v yh, yl;
          yfrag(tj, kb, yh, yl);
#pragma unroll
          for (int rb = 0; rb < UT; ++rb) p1[rb] = t2_mfma3(p1[rb], mh[rb][kb], ml[rb][kb], yh, yl);
        }
      h8v ph_[NKB], pl_[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) rhs_split(p1, kb, 0x1p-5f, ph_[kb], pl_[kb]);
      const int col = 16 * tj + lr;
#pragma unroll
      for (int ti = 0; ti < KT; ++ti) {
        f4w acc = f4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
          if (kb == 0 || 32 * kb < U) {
            h8v yh, yl;
            yfrag(ti, kb, yh, yl);
            acc = t2_mfma3(acc, yh, yl, ph_[kb], pl_[kb]);
          }
        // W_pert is symmetric: this lane's four values W_pert[16 ti + 4 h + q][16 tj + lr] are stored as W[16 tj + lr][16 ti + 4 h + q],
        // q = 0..3 -- FOUR CONSECUTIVE floats of one row, one 16-byte store per lane (round 4: four 4-byte stores per lane, 36 store
        // instructions per point instead of 9); w_mean is the ROW's: one value per lane
        const float wm = wbl[g * (16 * KT) + 16 * tj + lr];
        // (the values are formed outside the store's branch: the compiler pads the wait states between a matrix instruction
        //  and the first vector read of its result on the fall-through side of a branch only -- tools/check_mfma_hazards.py)
        f4w vq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          vq[q] = __builtin_fmaf(acc[q], wsc, wm);
          if (ti == tj) vq[q] += (4 * h + q == lr ? P.f0 : 0.0f);      // (the diagonal lives in the diagonal blocks)
          chk = __builtin_fmaf(vq[q], 0.0f, chk);                      // NaN once any value is NaN or infinite: one test per point
        }
        asm volatile("" : "+v"(vq));
        const int row = col, c0_ = 16 * ti + 4 * h;                    // (col = 16 tj + lr: the stored ROW)
        if constexpr (K4) {
          if ((tj < KT - 1 || rowok_last) && (ti < KT - 1 || colok_last))
            *reinterpret_cast<f4w*>(reinterpret_cast<char*>(Wg) + (offb + (unsigned)(tj * 64 * k)) + 64 * ti) = vq;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (row < k && c0_ + q < k) Wg[(unsigned)(row * k + c0_ + q)] = vq[q];
        }
      }
    }
    if (__any(chk != chk) && lane == 0) atomicOr(P.flags + p0 + g, MIA_FLAG_NONFINITE);
  }
}

static size_t tile2w_lds_bytes(int ut, int kt, int k) {
  return (size_t)ut * split_nc8(k) * 512 + 512 + (size_t)16 * ut * 12 + (size_t)16 * 16 * ut * 4 + (size_t)16 * 16 * kt * 4 + 256 +
         (size_t)ut * ((ut + 1) / 2) * 2 * 64 * 16 + 16;
}

template <int UT, int KT>
static int tile2w_launch(const Tile2wParams& tp, hipStream_t stream) {
  const size_t lds = tile2w_lds_bytes(UT, KT, tp.k);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  auto kern = (tp.k & 3) == 0 ? letkf_tile2w_kernel<UT, KT, true> : letkf_tile2w_kernel<UT, KT, false>;
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t nwork = (tp.ng + 15) >> 4;
  const int64_t gx = nwork < 65536 ? nwork : 65536;
  const int64_t gy = (nwork + gx - 1) / gx;
  if (gy > 65535) return MIA_ERR_UNSUPPORTED;
  kern<<<dim3((unsigned)gx, (unsigned)gy), dim3(64 * (16 / kWPts)), lds, stream>>>(tp);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <int UT>
static int tile2w_launch_u(const Tile2wParams& tp, int kt, hipStream_t stream) {
  switch (kt) {
    case 1: return tile2w_launch<UT, 1>(tp, stream);
    case 2: return tile2w_launch<UT, 2>(tp, stream);
    case 3: return tile2w_launch<UT, 3>(tp, stream);
    case 4: return tile2w_launch<UT, 4>(tp, stream);
    case 5: return tile2w_launch<UT, 5>(tp, stream);
    case 6: return tile2w_launch<UT, 6>(tp, stream);
  }
  return MIA_ERR_UNSUPPORTED;
}

// dual route with a union of at most 32 slots (UT <= 2), k <= 96, W addressed with 32-bit element offsets inside a point
bool tile2w_covers(int k, int p_max, int extra_blocks) {
  const int ut = tile_ut_for(p_max) + extra_blocks, kt = (k + 15) >> 4;
  return k >= 2 && k <= 96 && p_max <= k && ut >= 1 && ut <= 2 && ut <= kt + 1;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_letkf_weights_tiles_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                           const void* split_rec, int64_t P, const void* tile_lists, int p_max,
                                           int extra_blocks, float inf_factor, float* Xa, int64_t ldo, int64_t o0, float* W,
                                           int32_t* flags, int32_t* retry_count, void* stream_) {
  (void)hipGetLastError();
  hipStream_t stream = (hipStream_t)stream_;
  if (m < 1 || k < 2 || g1 < g0 || g0 < 0 || P < 0 || p_max < 0 || extra_blocks < 0 || !(inf_factor > 0.0f)) return MIA_ERR_SIZE;
  if (g1 == g0) return MIA_OK;
  if (!X || !Xa || !W || !split_rec || !tile_lists || !flags || !retry_count) return MIA_ERR_NULL;
  const int64_t ng = g1 - g0;
  if (!tile2_covers(m, k, p_max, extra_blocks, ldx, ldo, ng) || !tile2w_covers(k, p_max, extra_blocks)) return MIA_ERR_UNSUPPORTED;
  const int2* th = nullptr;
  const float2* tc = nullptr;
  if (!cheb_dual_table(stream, &th, &tc)) return MIA_ERR_UNSUPPORTED;
  const int ut = tile_ut_for(p_max) + extra_blocks, kt = (k + 15) >> 4;
  const int dmax = option(MIA_OPT_CHEB_DMAX);
  // the analysis (Xa, flags, declined points) first: the weights kernel repeats its decisions and ORs into its flags
  int rc = tile2_analysis_launch(X, ldx, m, k, g0, ng, split_rec, P, tile_lists, ut, inf_factor, Xa, ldo, o0, flags, retry_count, dmax,
                                 th, tc, stream, 0, 0, nullptr);
  if (rc != MIA_OK) return rc;
  const TileListLayout L = tile_list_layout(ng, ut);
  const char* base = (const char*)tile_lists;
  Tile2wParams tp;
  tp.k = k; tp.ng = ng;
  tp.rec = (const unsigned char*)split_rec; tp.rb = split_rec_bytes(k); tp.nc8 = split_nc8(k); tp.zero_rec = P;
  tp.thdr = (const int4*)(base + L.hdr); tp.tidx = (const int32_t*)(base + L.idx); tp.tD = (const f4w*)(base + L.D);
  const double rg = (double)(k - 1) / (double)inf_factor, km = (double)(k - 1);
  tp.inv_reg = (float)(1.0 / rg);
  tp.f0 = (float)sqrt(km / rg);
  tp.cs_phi = (float)(sqrt(km) / (rg * sqrt(rg)));
  tp.cs_psi = (float)(1.0 / rg);
  tp.W = W; tp.flags = flags; tp.retry_count = retry_count; tp.dmax = dmax; tp.tab_hdr = th; tp.tab_c = tc;
  return ut == 1 ? tile2w_launch_u<1>(tp, kt, stream) : tile2w_launch_u<2>(tp, kt, stream);
}
