// Global (unlocalised) ETKF on gfx950: ETKF.estimate_weights (pytassim/interface/etkf.py:99-120)
// = one ETKFModule.forward on the full (k, P) block (core/etkf.py:79-103), and the global
// ensemble transform _apply_weights with 2-D weights (interface/base.py:257-278).
//
//  1. gram_partial_kernel : C = Yb Yb^T and b = Yb d^T accumulated per observation chunk into
//     slabs (no float atomics: the slab sum below runs in a fixed order, so results are
//     bitwise reproducible);
//  2. etkf_solve_kernel   : one workgroup sums the slabs, runs the LDS Jacobi eigensolver,
//     forms weights = w_mean 1^T + V diag(sqrt((k-1)/l)) V^T;
//  3. apply_weights_kernel: xa = mean + X' weights for every grid point (one thread per point,
//     weights broadcast from LDS).
#include "mia_common.h"
#include "mia_jacobi.h"
#include "mia_kernel_prog.h"
#include "mia_kernels.h"

namespace mia {

template <typename T>
__global__ __launch_bounds__(256) void gram_partial_kernel(const T* Yb, const T* d, int k, int64_t P, int ch,
                                                           T* slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);   // [k + 1][ch + 1], row k holds d
  const int ld = ch + 1;
  const int64_t j0 = (int64_t)blockIdx.x * ch;
  int nj = (int)((P - j0) < ch ? (P - j0) : ch);
  for (int it = threadIdx.x; it < (k + 1) * ch; it += 256) {
    const int i = it / ch, j = it - i * ch;
    T v = T(0);
    if (j < nj) v = (i < k) ? Yb[(int64_t)i * P + j0 + j] : d[j0 + j];
    tile[i * ld + j] = v;
  }
  __syncthreads();
  T* slab = slabs + (size_t)blockIdx.x * ((size_t)k * k + k);
  for (int it = threadIdx.x; it < k * (k + 1); it += 256) {
    const int a = it / (k + 1), b = it - a * (k + 1);   // b == k -> right-hand side
    if (b < k && b < a) continue;                        // lower triangle mirrored below
    T acc = T(0);
    for (int j = 0; j < nj; ++j) acc += tile[a * ld + j] * tile[b * ld + j];
    if (b == k) slab[(size_t)k * k + a] = acc;
    else { slab[a * k + b] = acc; slab[b * k + a] = acc; }
  }
}

// Global KERNELISED ETKF (KETKF.estimate_weights, interface/ketkf.py:34-123 -> KETKFModule, core/ketkf.py:65-94) for
// any number of observations: the three pair statistics every reference kernel is a function of -- x.y, |x-y|_2^2,
// |x-y|_1 over the P observations, for all member pairs and for every member against the observation vector -- are
// accumulated per observation chunk into slabs [dot | sq | l1], each (k*k + k) values, summed in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void pairstat_partial_kernel(const T* Yb, const T* d, int k, int64_t P, int ch, T* slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);   // [k + 1][ch + 1], row k holds d
  const int ld = ch + 1;
  const int64_t j0 = (int64_t)blockIdx.x * ch;
  const int nj = (int)((P - j0) < ch ? (P - j0) : ch);
  for (int it = threadIdx.x; it < (k + 1) * ch; it += 256) {
    const int i = it / ch, j = it - i * ch;
    T v = T(0);
    if (j < nj) v = (i < k) ? Yb[(int64_t)i * P + j0 + j] : d[j0 + j];
    tile[i * ld + j] = v;
  }
  __syncthreads();
  const size_t one = (size_t)k * k + k;
  T* slab = slabs + (size_t)blockIdx.x * 3 * one;
  for (int it = threadIdx.x; it < k * (k + 1); it += 256) {
    const int a = it / (k + 1), b = it - a * (k + 1);   // b == k -> against the observations
    if (b < k && b < a) continue;                        // lower triangle mirrored below
    T dt = T(0), sq = T(0), l1 = T(0);
    for (int j = 0; j < nj; ++j) {
      const T xa = tile[a * ld + j], xb = tile[b * ld + j];
      const T df = xa - xb;
      dt += xa * xb; sq += df * df; l1 += t_abs(df);
    }
    if (b == k) { slab[(size_t)k * k + a] = dt; slab[one + (size_t)k * k + a] = sq; slab[2 * one + (size_t)k * k + a] = l1; }
    else {
      slab[a * k + b] = dt; slab[b * k + a] = dt;
      slab[one + a * k + b] = sq; slab[one + b * k + a] = sq;
      slab[2 * one + a * k + b] = l1; slab[2 * one + b * k + a] = l1;
    }
  }
}

template <typename T, bool KERN>
__global__ __launch_bounds__(256) void etkf_solve_kernel(const T* slabs, int nslab, int k, T reg, T tol,
                                                         int max_sweeps, T* W, int32_t* flags, KernelProgram<T> prog,
                                                         T prior_diag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x;
  const int n = (k + 1) & ~1, lda = n | 1;
  T* A = sm;                     // [n][lda]
  T* V = A + (size_t)n * lda;    // [n][lda]
  T* cs = V + (size_t)n * lda;   // [n]
  T* rhs = cs + n;               // [n]
  T* gW = rhs + n;               // [n]
  T* av = gW + n;                // [n]
  T* wbar = av + n;              // [n]
  int* iflag = reinterpret_cast<int*>(wbar + n);
  if (KERN && nslab == 0) {      // no observation at all: the inflated prior (core/etkf.py:91-95)
    for (int it = tid; it < k * k; it += 256) W[it] = (it / k == it % k) ? prior_diag : T(0);
    if (flags && tid == 0) flags[0] = 0;
    return;
  }
  const size_t one = (size_t)k * k + k;
  const size_t slab_sz = KERN ? 3 * one : one;
  for (int it = tid; it < n * n; it += 256) {
    const int a = it / n, b = it - a * n;
    T acc = T(0);
    if (a < k && b < k) {
      if (KERN) {
        T sq = T(0), l1 = T(0);
        for (int s = 0; s < nslab; ++s) {
          acc += slabs[s * slab_sz + a * k + b]; sq += slabs[s * slab_sz + one + a * k + b]; l1 += slabs[s * slab_sz + 2 * one + a * k + b];
        }
        acc = kprog_eval(prog, acc, sq, l1, a == b);
      } else {
        for (int s = 0; s < nslab; ++s) acc += slabs[s * slab_sz + a * k + b];
      }
    }
    A[a * lda + b] = acc;
    V[a * lda + b] = (a == b) ? T(1) : T(0);
  }
  for (int a = tid; a < n; a += 256) {
    T acc = T(0);
    if (a < k) {
      if (KERN) {
        T sq = T(0), l1 = T(0);
        for (int s = 0; s < nslab; ++s) {
          acc += slabs[s * slab_sz + (size_t)k * k + a]; sq += slabs[s * slab_sz + one + (size_t)k * k + a];
          l1 += slabs[s * slab_sz + 2 * one + (size_t)k * k + a];
        }
        acc = kprog_eval(prog, acc, sq, l1, false);     // k(Yb, d): never the same sample (diag.py:65-66)
      } else {
        for (int s = 0; s < nslab; ++s) acc += slabs[s * slab_sz + (size_t)k * k + a];
      }
    }
    rhs[a] = acc;
  }
  __syncthreads();
  if (KERN) {      // double centring of K and centring of k(Yb, d)   (core/ketkf.py:77-89); av, gW as scratch
    for (int a = tid; a < k; a += 256) {
      T acc = T(0);
      for (int b = 0; b < k; ++b) acc += A[a * lda + b];
      av[a] = acc / T(k);                                  // row (= column) means
    }
    __syncthreads();
    if (tid == 0) {
      T gm = T(0), om = T(0);
      for (int a = 0; a < k; ++a) { gm += av[a]; om += rhs[a]; }
      gW[0] = gm / T(k); gW[1] = om / T(k);
    }
    __syncthreads();
    const T gm = gW[0], om = gW[1];
    for (int it = tid; it < k * k; it += 256) {
      const int a = it / k, b = it - a * k;
      A[a * lda + b] = A[a * lda + b] - av[b] - (av[a] - gm);
    }
    for (int a = tid; a < k; a += 256) rhs[a] = rhs[a] - om - (av[a] - gm);
    __syncthreads();
  }
  const bool conv = jacobi_lds<T, 256>(A, V, cs, iflag, n, n, lda, reg, tol, max_sweeps);
  const T km1 = T(k - 1);
  for (int r = tid; r < n; r += 256) {
    T lam = A[r * lda + r];
    lam = lam > T(0) ? lam : T(0);       // clamp(min=0), core/utils.py:58
    const T le = lam + reg;              // + reg_value, :59
    T acc = T(0);
    for (int b = 0; b < k; ++b) acc += V[b * lda + r] * rhs[b];
    gW[r] = r < k ? t_sqrt(km1 / le) : T(0);
    av[r] = r < k ? acc / le : T(0);
  }
  __syncthreads();
  for (int i = tid; i < k; i += 256) {
    T acc = T(0);
    for (int r = 0; r < n; ++r) acc += V[i * lda + r] * av[r];
    wbar[i] = acc;
  }
  __syncthreads();
  for (int it = tid; it < k * k; it += 256) {
    const int i = it / k, j = it - i * k;
    T acc = wbar[i];
    for (int r = 0; r < n; ++r) acc += gW[r] * V[i * lda + r] * V[j * lda + r];
    W[it] = acc;
  }
  if (flags && tid == 0) flags[0] = conv ? 0 : MIA_FLAG_NOCONV;
}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void apply_weights_kernel(const T* X, int64_t ldx, int m, int k, int64_t g0,
                                                           int64_t ng, const T* W, T* Xa, int64_t ldo, int64_t o0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ws = reinterpret_cast<T*>(smem_raw);   // [k][k]
  T* xs = Ws + (size_t)k * k;               // [k][NT], one column per thread
  const int tid = threadIdx.x;
  for (int it = tid; it < k * k; it += NT) Ws[it] = W[it];
  __syncthreads();
  const int64_t pt = (int64_t)blockIdx.x * NT + tid;
  const int mi = blockIdx.y;
  if (pt >= ng) return;
  const T* xrow = X + (int64_t)mi * k * ldx + g0 + pt;
  T mean = T(0);
  for (int i = 0; i < k; ++i) { const T v = xrow[(int64_t)i * ldx]; xs[i * NT + tid] = v; mean += v; }
  mean /= T(k);
  T* orow = Xa + (int64_t)mi * k * ldo + o0 + pt;
  for (int j = 0; j < k; ++j) {
    T acc = T(0);
    for (int i = 0; i < k; ++i) acc += (xs[i * NT + tid] - mean) * Ws[i * k + j];
    orow[(int64_t)j * ldo] = mean + acc;
  }
}

static inline float t_sqrt_host(float x) { return sqrtf(x); }
static inline double t_sqrt_host(double x) { return sqrt(x); }

static int gram_chunk(int k, int eb) {
  int ch = (int)(65536 / ((size_t)(k + 1) * eb)) - 1;
  ch = ch / 32 * 32;
  if (ch > 256) ch = 256;
  if (ch < 32) ch = 32;
  return ch;
}

template <typename T>
static int etkf_weights_impl(const T* Yb, const T* d, int k, int64_t P, T inf_factor, T* W, int32_t* flags,
                             void* ws, size_t ws_bytes, hipStream_t stream) {
  if (k < 2 || P < 0 || !(inf_factor > T(0))) return MIA_ERR_SIZE;
  if (!W) return MIA_ERR_NULL;
  if (P > 0 && (!Yb || !d)) return MIA_ERR_NULL;
  size_t need = 0;
  int rc = mia_etkf_workspace_bytes(k, P, (int)sizeof(T), &need);
  if (rc != MIA_OK) return rc;
  if (ws_bytes < need || (P > 0 && !ws)) return MIA_ERR_WORKSPACE;
  const int ch = gram_chunk(k, (int)sizeof(T));
  const int64_t nslab = (P + ch - 1) / ch;
  if (nslab > 1 << 20) return MIA_ERR_UNSUPPORTED;
  T* slabs = (T*)ws;
  if (nslab > 0) {
    const size_t lds = (size_t)(k + 1) * (ch + 1) * sizeof(T);
    auto kern = gram_partial_kernel<T>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nslab), dim3(256), lds, stream>>>(Yb, d, k, P, ch, slabs);
    MIA_LAUNCH_CHECK();
  }
  const int n = (k + 1) & ~1, lda = n | 1;
  const size_t lds2 = ((size_t)2 * n * lda + 5 * (size_t)n) * sizeof(T) + 16;
  if (lds2 > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  auto kern2 = etkf_solve_kernel<T, false>;
  if (lds2 > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  KernelProgram<T> none;
  none.n = 0;
  kern2<<<dim3(1), dim3(256), lds2, stream>>>(slabs, (int)nslab, k, T(k - 1) / inf_factor,
                                               sizeof(T) == 4 ? T(2.4e-7) : T(9e-16), sizeof(T) == 4 ? 16 : 24, W, flags,
                                               none, T(0));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <typename T>
static int ketkf_weights_impl(const T* Yb, const T* d, int k, int64_t P, T inf_factor, const mia_kernel_op_t* prog,
                              int n_ops, T* W, int32_t* flags, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (k < 2 || P < 0 || !(inf_factor > T(0))) return MIA_ERR_SIZE;
  int rc = kernel_program_check(prog, n_ops);
  if (rc != MIA_OK) return rc;
  if (!W) return MIA_ERR_NULL;
  if (P > 0 && (!Yb || !d)) return MIA_ERR_NULL;
  size_t need = 0;
  rc = mia_ketkf_workspace_bytes(k, P, (int)sizeof(T), &need);
  if (rc != MIA_OK) return rc;
  if (ws_bytes < need || (P > 0 && !ws)) return MIA_ERR_WORKSPACE;
  const int ch = gram_chunk(k, (int)sizeof(T));
  const int64_t nslab = (P + ch - 1) / ch;
  if (nslab > 1 << 20) return MIA_ERR_UNSUPPORTED;
  T* slabs = (T*)ws;
  if (nslab > 0) {
    const size_t lds = (size_t)(k + 1) * (ch + 1) * sizeof(T);
    auto kern = pairstat_partial_kernel<T>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nslab), dim3(256), lds, stream>>>(Yb, d, k, P, ch, slabs);
    MIA_LAUNCH_CHECK();
  }
  const int n = (k + 1) & ~1, lda = n | 1;
  const size_t lds2 = ((size_t)2 * n * lda + 5 * (size_t)n) * sizeof(T) + 16;
  if (lds2 > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  KernelProgram<T> kp;
  kp.n = n_ops;
  for (int i = 0; i < n_ops; ++i) { kp.op[i] = (unsigned char)prog[i].op; kp.val[i] = T(prog[i].value); }
  auto kern2 = etkf_solve_kernel<T, true>;
  if (lds2 > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  kern2<<<dim3(1), dim3(256), lds2, stream>>>(slabs, (int)nslab, k, T(k - 1) / inf_factor,
                                               sizeof(T) == 4 ? T(2.4e-7) : T(9e-16), sizeof(T) == 4 ? 16 : 24, W, flags,
                                               kp, t_sqrt_host(inf_factor));
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

template <typename T>
static int apply_weights_impl(const T* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1, const T* W, T* Xa,
                              int64_t ldo, int64_t o0, hipStream_t stream) {
  if (g1 < g0 || g0 < 0 || m < 1 || k < 2) return MIA_ERR_SIZE;
  const int64_t ng = g1 - g0;
  if (ng == 0) return MIA_OK;
  if (!X || !W || !Xa) return MIA_ERR_NULL;
  if (ldx < g1 || ldo < o0 + ng) return MIA_ERR_SIZE;
  if (m > 65535) return MIA_ERR_UNSUPPORTED;
  const bool small = (size_t)k * 256 * sizeof(T) <= 64 * 1024;
  const int NT = small ? 256 : 64;
  const size_t lds = ((size_t)k * k + (size_t)k * NT) * sizeof(T);
  if (lds > (long long)kMaxDynamicLds) return MIA_ERR_UNSUPPORTED;
  const int64_t nb = (ng + NT - 1) / NT;
  if (nb > 2147483647LL) return MIA_ERR_UNSUPPORTED;
  if (small) {
    auto kern = apply_weights_kernel<T, 256>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nb, (unsigned)m), dim3(256), lds, stream>>>(X, ldx, m, k, g0, ng, W, Xa, ldo, o0);
  } else {
    auto kern = apply_weights_kernel<T, 64>;
    if (lds > 48 * 1024) MIA_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<dim3((unsigned)nb, (unsigned)m), dim3(64), lds, stream>>>(X, ldx, m, k, g0, ng, W, Xa, ldo, o0);
  }
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

}  // namespace mia

using namespace mia;

extern "C" int mia_etkf_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  if (k < 2 || P < 0 || (elem_bytes != 4 && elem_bytes != 8)) return MIA_ERR_SIZE;
  const int ch = gram_chunk(k, elem_bytes);
  const size_t nslab = (size_t)((P + ch - 1) / ch);
  *bytes = align_up(nslab * ((size_t)k * k + k) * elem_bytes + 256, 256);
  return MIA_OK;
}

extern "C" int mia_ketkf_workspace_bytes(int k, int64_t P, int elem_bytes, size_t* bytes) {
  size_t one = 0;
  const int rc = mia_etkf_workspace_bytes(k, P, elem_bytes, &one);
  if (rc != MIA_OK) return rc;
  *bytes = align_up(3 * one, 256);
  return MIA_OK;
}
extern "C" int mia_ketkf_weights_f32(const float* Yb, const float* d, int k, int64_t P, float inf_factor,
                                     const mia_kernel_op_t* prog, int n_ops, float* W, int32_t* flags_opt, void* ws,
                                     size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return ketkf_weights_impl<float>(Yb, d, k, P, inf_factor, prog, n_ops, W, flags_opt, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mia_ketkf_weights_f64(const double* Yb, const double* d, int k, int64_t P, double inf_factor,
                                     const mia_kernel_op_t* prog, int n_ops, double* W, int32_t* flags_opt, void* ws,
                                     size_t ws_bytes, void* stream) {
  (void)hipGetLastError();
  return ketkf_weights_impl<double>(Yb, d, k, P, inf_factor, prog, n_ops, W, flags_opt, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mia_etkf_weights_f32(const float* Yb, const float* d, int k, int64_t P, float inf_factor, float* W,
                                    int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return etkf_weights_impl<float>(Yb, d, k, P, inf_factor, W, flags_opt, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mia_etkf_weights_f64(const double* Yb, const double* d, int k, int64_t P, double inf_factor,
                                    double* W, int32_t* flags_opt, void* ws, size_t ws_bytes, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return etkf_weights_impl<double>(Yb, d, k, P, inf_factor, W, flags_opt, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mia_apply_weights_f32(const float* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                     const float* W, float* Xa, int64_t ldo, int64_t o0, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  // grid points as the columns of a matrix-instruction product (apply_local.hip) where the shape allows: the same checks first
  if (g1 > g0 && g0 >= 0 && m >= 1 && k >= 2 && X && W && Xa && ldx >= g1 && ldo >= o0 + (g1 - g0) && m <= 65535) {
    const int rc = apply_global_tile_launch(X, ldx, m, k, g0, g1 - g0, W, Xa, ldo, o0, (hipStream_t)stream);
    if (rc != MIA_ERR_UNSUPPORTED) return rc;
  }
  return apply_weights_impl<float>(X, ldx, m, k, g0, g1, W, Xa, ldo, o0, (hipStream_t)stream);
}
extern "C" int mia_apply_weights_f64(const double* X, int64_t ldx, int m, int k, int64_t g0, int64_t g1,
                                     const double* W, double* Xa, int64_t ldo, int64_t o0, void* stream) {
  (void)hipGetLastError();  // drop stale per-thread error state left by other users of the runtime
  return apply_weights_impl<double>(X, ldx, m, k, g0, g1, W, Xa, ldo, o0, (hipStream_t)stream);
}
