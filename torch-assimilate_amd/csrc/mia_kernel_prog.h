// Device-side evaluation of a kernel expression (mia_kernel_op_t program, include/mia_letkf.h), shared by the localised
// analysis kernel (letkf_wave.hip) and the global kernelised ETKF (etkf_global.hip).
#pragma once
#include "mia_common.h"
#include "mia_jacobi.h"

namespace mia {

// Kernel expression in reverse Polish form (MIA_KOP_*), evaluated per matrix element from the three pair
// statistics every reference kernel is a function of: x.y (kernels/utils.py:38-58 dot_product), |x-y|_2^2
// (utils.py:93-110 euclidean_dist) and |x-y|_1 (utils.py:61-90 distance_matrix, norm 1).  The operand stack is six
// named registers shifted on push/pop (no indexed private array, so no scratch memory).
template <typename T>
__device__ inline T kprog_eval(const KernelProgram<T>& kp, T dot, T sq, T l1, bool same) {
  T s0 = T(0), s1 = T(0), s2 = T(0), s3 = T(0), s4 = T(0), s5 = T(0);
  for (int i = 0; i < kp.n; ++i) {
    const int op = kp.op[i];
    if (op <= MIA_KOP_DIAG) {      // push
      T v;
      switch (op) {
        case MIA_KOP_DOT: v = dot; break;
        case MIA_KOP_SQDIST: v = sq; break;
        case MIA_KOP_L1DIST: v = l1; break;
        case MIA_KOP_CONST: v = kp.val[i]; break;
        default: v = same ? kp.val[i] : T(0); break;   // MIA_KOP_DIAG
      }
      s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
    } else if (op <= MIA_KOP_POW) {   // binary: (s1 op s0)
      T v;
      if (op == MIA_KOP_ADD) v = s1 + s0;
      else if (op == MIA_KOP_MUL) v = s1 * s0;
      else v = t_pow(s1, s0);
      s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5;
    } else {                         // unary
      if (op == MIA_KOP_EXP) s0 = t_exp(s0);
      else if (op == MIA_KOP_TANH) s0 = t_tanh(s0);
      else s0 = t_sin(s0);
    }
  }
  return s0;
}

}  // namespace mia
