// Ensemble transform with PER-GRID-POINT weights on tiles of sixteen grid points: _apply_weights with weights of dims
// (grid, ensemble, ensemble_new) -- pytassim/interface/base.py:257-278, what update_state does with the result of
// estimate_weights (assimilation/filter/filter.py:157-164) for every variable and time of the state:
//     xa[v][j][g] = mean_vg + sum_i (x[v][i][g] - mean_vg) W[g][i][j]          mean_vg = (1 / k) sum_i x[v][i][g]
//
// HBM-bound work: 4 k^2 bytes of W per point once + 8 k bytes per state row and point.  The round-1 kernel (ienks.hip, kept for
// float64 and ensembles beyond 96 members) gave a wavefront one grid point: its loads of x and stores of xa are one float per
// 64-byte sector (lanes walk the member axis, ldx floats apart) and W is read again for every state row -- 0.11 ms per row and
// 1e5 points, five times the analysis kernel's own row loop.  Here a workgroup of four wavefronts owns a TILE of sixteen
// consecutive grid points:
//   * the tile's state rows arrive sixteen rows at a time as whole 64-byte segments (row v, member i, the sixteen points) in an
//     LDS image [v][i][g] whose row pitch is odd in words: the fragment reads below touch every bank once;
//   * wavefront w transforms points 4 w .. 4 w + 3, one after the other, as xa_g^T (k x 16 rows) = W_g^T (k x k) x_g'^T (k x 16 rows)
//     with v_mfma_f32_16x16x4_f32: the A operand IS W_g read straight from memory (lane = column j: 64-byte runs of a row of W,
//     no staging, 4 k^2 / 64 registers hold the whole matrix for k <= 48), the B operand the point's column of the image (mean
//     removed in registers: a lane's values all belong to one state row), the result goes back into the image in place -- column
//     g belongs to this wavefront alone;
//   * the image leaves as whole segments again.
// float32 matrix instructions: the transform needs the f32 accuracy of its inputs, its flops (2 m k^2 per point) stay below the
// memory time up to m ~ 200 rows, and no operand needs splitting.
#include "mia_common.h"
#include "mia_kernels.h"
#include "mia_options.h"

namespace mia {

using f4a = __attribute__((ext_vector_type(4))) float;

struct ApplyTileParams {
  const float* X; int64_t ldx; int m, k; int64_t g0, ng; const float* W; float* Xa; int64_t ldo, o0;
  int kp;        // members rounded up to a multiple of 4 (depth of one matrix instruction)
  int pitch;     // words per state row of the LDS image: 16 kp + 1
  int exp_skip;  // (experiment builds) 1: no products, 2: no loads of x, 4: no stores of xa, 8: no loads of W
};

// KT = member blocks of sixteen (k <= 16 KT)
template <int KT>
__global__ __launch_bounds__(256, KT <= 4 ? 3 : 1) void apply_local_tile_kernel(ApplyTileParams P) {
  extern __shared__ __attribute__((aligned(16))) float img[];          // [16 rows][kp members][16 points], row pitch P.pitch
  constexpr int KS = 4 * KT;                                           // depth steps of four members
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, h = lane >> 4;
  const int k = P.k, kp = P.kp, pitch = P.pitch;
  const int64_t p0 = (int64_t)blockIdx.x << 4;
  const int npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
  const int nks = kp >> 2;                                             // depth steps that hold members
  const int sg = tid & 15, s0 = tid >> 4;                              // staging: segment s0 + 16 it, point sg of it
  const float inv_k = 1.0f / (float)k;
  for (int r0 = 0; r0 < P.m; r0 += 16) {
    const int nrows = P.m - r0 < 16 ? P.m - r0 : 16;
    // ---- the tile's next sixteen state rows into the image (rows / points / members that do not exist: zeros): thread (s0, sg)
    //      takes point sg of the segments (row v, member s0 + 16 ib), eight rows at a time.  (Measured: sixteen unconditional
    //      loads in flight per thread -- clamped addresses, no branches -- run at HALF this speed: every 64-byte segment of a tile
    //      lies in another page, 400 KB and 16 MB apart, and more of them in flight is more translation misses, not more bandwidth.)
    {
      const char* xb = reinterpret_cast<const char*>(P.X + (int64_t)r0 * k * P.ldx + P.g0 + p0);
      const unsigned xrow32 = (unsigned)k * (unsigned)P.ldx * 4u;
      for (int i = s0; i < kp; i += 16) {
        const unsigned off = ((unsigned)i * (unsigned)P.ldx + (unsigned)sg) * 4u;
        const bool keep = i < k && sg < npts && !(P.exp_skip & 2);
#pragma unroll
        for (int vb = 0; vb < 16; vb += 8) {
          float val[8];
          const char* xv = xb + (int64_t)vb * xrow32;                  // (uniform; the 32-bit lane offsets span eight rows)
#pragma unroll
          for (int u = 0; u < 8; ++u)
            val[u] = (keep && vb + u < nrows) ? *reinterpret_cast<const float*>(xv + (off + (unsigned)u * xrow32)) : 0.0f;
#pragma unroll
          for (int u = 0; u < 8; ++u) img[(vb + u) * pitch + i * 16 + sg] = val[u];
        }
      }
    }
    __syncthreads();
    // ---- this wavefront's four points, one block row (sixteen members j) of one point at a time: its sixteen-by-k slice of W_g is
    //      the A operand, requested one unit ahead of its products (two register sets of KS values)
    auto load_w = [&](int g, int jb, float (&a)[KS]) {                // a[ks] = W_g[4 ks + h][16 jb + lr]
      const float* wg = P.W + (p0 + g) * (int64_t)k * k;
      const int j = 16 * jb + lr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int i = 4 * ks + h;
        a[ks] = (g < npts && ks < nks && i < k && j < k && !(P.exp_skip & 8)) ? wg[i * k + j] : 0.0f;
      }
    };
    float a[KS], an[KS], b[KS];
    float mean = 0.0f;
    load_w(4 * wave, 0, an);
#pragma unroll 1
    for (int u = 0; u < 4 * KT; ++u) {
      const int pp = u / KT, jb = u - pp * KT, g = 4 * wave + pp;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = an[ks];
      if (u + 1 < 4 * KT) {
        const int un = u + 1, ppn = un / KT;
        load_w(4 * wave + ppn, un - ppn * KT, an);
      }
      if (g >= npts || 16 * jb >= k || (P.exp_skip & 1)) continue;      // (wave-uniform)
      if (jb == 0) {
        // B fragments: b[ks] = x[row lr][member 4 ks + h] of point g; the row's mean over the members (its four lanes hold
        // disjoint quarters of them), removed before the products
        float part = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          b[ks] = ks < nks ? img[lr * pitch + (4 * ks + h) * 16 + g] : 0.0f;
          part += b[ks];
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        mean = part * inv_k;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] -= mean;                 // (padding members become -mean against zero rows of W: nothing)
      }
      // the block row of the result: rows j = 16 jb + 4 h + q of state row lr, back into column g of the image (all KS depth
      // steps run: beyond the members a is zero)
      f4a acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], b[ks], acc, 0, 0, 0);
      float out[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q] = acc[q] + mean;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 16 * jb + 4 * h + q;
        if (j < k) img[lr * pitch + j * 16 + g] = out[q];
      }
    }
    __syncthreads();
    // ---- the image leaves as whole segments
    {
      char* ob = reinterpret_cast<char*>(P.Xa + (int64_t)r0 * k * P.ldo + P.o0 + p0);
      const unsigned orow32 = (unsigned)k * (unsigned)P.ldo * 4u;
      for (int i = s0; i < k; i += 16) {
        const unsigned off = ((unsigned)i * (unsigned)P.ldo + (unsigned)sg) * 4u;
        if (sg < npts && !(P.exp_skip & 4)) {
#pragma unroll 4
          for (int v = 0; v < nrows; ++v) *reinterpret_cast<float*>(ob + (int64_t)v * orow32 + off) = img[v * pitch + i * 16 + sg];
        }
      }
    }
    __syncthreads();
  }
}

// ONE weight matrix for all grid points (the global ETKF: _apply_weights with weights of dims (ensemble, ensemble_new),
// interface/etkf.py:99-120 + base.py:257-278): xa_v (k x G) = W^T x_v' + mean is a plain product with the GRID POINTS as the
// columns -- sixteen consecutive points are the N dimension of the matrix instruction, so the B operand (64-byte runs of a
// member's row) and the result (64-byte runs of a new member's row) go straight between memory and registers: no LDS, no
// transposition.  One wavefront per tile of sixteen points, all state rows; W^T stays in registers.
template <int KT>
__global__ __launch_bounds__(256) void apply_global_tile_kernel(ApplyTileParams P) {
  constexpr int KS = 4 * KT;
  constexpr bool kWholeW = KT <= 3;
  const int lane = threadIdx.x & 63, lr = lane & 15, h = lane >> 4;
  const int k = P.k;
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t p0 = tile << 4;
  if (p0 >= P.ng) return;
  const int npts = P.ng - p0 < 16 ? (int)(P.ng - p0) : 16;
  const float inv_k = 1.0f / (float)k;
  auto load_w = [&](int jb, float (&a)[KS]) {                          // a[ks] = W[4 ks + h][16 jb + lr]
    const int j = 16 * jb + lr;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int i = 4 * ks + h;
      a[ks] = (i < k && j < k) ? P.W[i * k + j] : 0.0f;
    }
  };
  float aw[kWholeW ? KT : 1][KS];
  if constexpr (kWholeW) {
#pragma unroll
    for (int jb = 0; jb < KT; ++jb) load_w(jb, aw[jb]);
  }
  const bool col = lr < npts;
  const unsigned xoff = (unsigned)lr * 4u, xstep = (unsigned)P.ldx * 4u, ostep = (unsigned)P.ldo * 4u;
  for (int v = 0; v < P.m; ++v) {
    const char* xb = reinterpret_cast<const char*>(P.X + (int64_t)v * k * P.ldx + P.g0 + p0);
    char* ob = reinterpret_cast<char*>(P.Xa + (int64_t)v * k * P.ldo + P.o0 + p0);
    float b[KS];
    float part = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int i = 4 * ks + h;
      b[ks] = (col && i < k) ? *reinterpret_cast<const float*>(xb + (xoff + (unsigned)i * xstep)) : 0.0f;
      part += b[ks];
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    const float mean = part * inv_k;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) b[ks] -= mean;                     // (members past k meet zero rows of W)
#pragma unroll
    for (int jb = 0; jb < KT; ++jb) {
      if (16 * jb < k) {                                               // (uniform)
        f4a acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (kWholeW) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[jb][ks], b[ks], acc, 0, 0, 0);
        } else {
          float one[KS];
          load_w(jb, one);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(one[ks], b[ks], acc, 0, 0, 0);
        }
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = acc[q] + mean;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 16 * jb + 4 * h + q;
          if (col && j < k) *reinterpret_cast<float*>(ob + (xoff + (unsigned)j * ostep)) = out[q];
        }
      }
    }
  }
}

int apply_global_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* W, float* Xa,
                             int64_t ldo, int64_t o0, hipStream_t stream) {
  if (k < 2 || k > 96 || m < 1 || ng < 1) return MIA_ERR_UNSUPPORTED;
  if ((int64_t)(k + 1) * ldx * 4 >= ((int64_t)1 << 32) || (int64_t)(k + 1) * ldo * 4 >= ((int64_t)1 << 32)) return MIA_ERR_UNSUPPORTED;
  const int64_t nb = (((ng + 15) >> 4) + 3) >> 2;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  ApplyTileParams p{X, ldx, m, k, g0, ng, W, Xa, ldo, o0, (k + 3) & ~3, 0, 0};
  const int kt = (k + 15) >> 4;
  void (*kern)(ApplyTileParams) = nullptr;
  switch (kt) {
    case 1: kern = apply_global_tile_kernel<1>; break;
    case 2: kern = apply_global_tile_kernel<2>; break;
    case 3: kern = apply_global_tile_kernel<3>; break;
    case 4: kern = apply_global_tile_kernel<4>; break;
    case 5: kern = apply_global_tile_kernel<5>; break;
    case 6: kern = apply_global_tile_kernel<6>; break;
    default: return MIA_ERR_UNSUPPORTED;
  }
  kern<<<dim3((unsigned)nb), dim3(256), 0, stream>>>(p);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

// float32, 2 <= k <= 96; MIA_ERR_UNSUPPORTED otherwise (the caller keeps the one-point-per-wavefront kernel of ienks.hip)
int apply_local_tile_launch(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t ng, const float* W, float* Xa,
                            int64_t ldo, int64_t o0, hipStream_t stream) {
  if (k < 2 || k > 96 || m < 1 || ng < 1) return MIA_ERR_UNSUPPORTED;
  // (32-bit lane offsets: eight state rows of x, one member row of xa)
  if ((int64_t)9 * k * ldx * 4 >= ((int64_t)1 << 32) || (int64_t)(k + 1) * ldo * 4 >= ((int64_t)1 << 32)) return MIA_ERR_UNSUPPORTED;
  const int64_t ntile = (ng + 15) >> 4;
  if (ntile > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  ApplyTileParams p{X, ldx, m, k, g0, ng, W, Xa, ldo, o0, (k + 3) & ~3, 0, 0};
  MIA_EXP_SET(p.exp_skip, "MIA_APPLY_SKIP", atoi);
  p.pitch = 16 * p.kp + 1;
  const size_t lds = (size_t)16 * p.pitch * sizeof(float);
  if (lds > kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  const int kt = (k + 15) >> 4;
  void (*kern)(ApplyTileParams) = nullptr;
  switch (kt) {
    case 1: kern = apply_local_tile_kernel<1>; break;
    case 2: kern = apply_local_tile_kernel<2>; break;
    case 3: kern = apply_local_tile_kernel<3>; break;
    case 4: kern = apply_local_tile_kernel<4>; break;
    case 5: kern = apply_local_tile_kernel<5>; break;
    case 6: kern = apply_local_tile_kernel<6>; break;
    default: return MIA_ERR_UNSUPPORTED;
  }
  if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<dim3((unsigned)ntile), dim3(256), lds, stream>>>(p);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia
