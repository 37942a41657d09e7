// Version / status strings of the C ABI (include/mia_letkf.h).
#include "mia_letkf.h"

extern "C" int mia_version(void) { return MIA_VERSION; }

extern "C" const char* mia_status_string(int status) {
  switch (status) {
    case MIA_OK: return "ok";
    case MIA_ERR_NULL: return "required pointer is NULL";
    case MIA_ERR_SIZE: return "invalid or inconsistent size argument";
    case MIA_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels (LDS capacity / index range)";
    case MIA_ERR_WORKSPACE: return "workspace too small (use the *_workspace_bytes query)";
    case MIA_ERR_ALIGN: return "workspace pointer must be 256-byte aligned";
    case MIA_ERR_COMM: return "RCCL / communicator failure (mia_comm_last_error has the detail)";
    default: return status > 0 ? "HIP runtime error (value is the hipError_t)" : "unknown status";
  }
}
