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

// ---- route options (include/mia_letkf.h, mia_set_option): explicit, process-wide switches for the routes a caller or a test
//      may legitimately want to steer.  They replace the MIA_* environment variables round 1 read inside launch code;
//      what remains environment-driven is compiled in only with -DMIA_EXPERIMENTS (tools/ builds).
#include <atomic>
#include <cstring>
#include "mia_options.h"

namespace mia {
static std::atomic<int> g_opt[MIA_OPT_COUNT_] = {62, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0};
// A thread may run under a SNAPSHOT of the options (the step driver's launch threads: a job is enqueued with the routes that
// were in force when the caller submitted it, whatever mia_set_option does in the meantime)
static thread_local const int* t_override = nullptr;
int option(int id) {
  if (id < 0 || id >= MIA_OPT_COUNT_) return 0;
  return t_override ? t_override[id] : g_opt[id].load(std::memory_order_relaxed);
}
void option_snapshot(int* out) { for (int i = 0; i < MIA_OPT_COUNT_; ++i) out[i] = g_opt[i].load(std::memory_order_relaxed); }
void option_override(const int* snapshot) { t_override = snapshot; }
}  // namespace mia

// ---- the analysis kernel launched last (any thread), as rocprofv3 names it
#include <cstdarg>
#include <cstdio>
#include <mutex>
namespace mia {
static std::mutex g_last_mu;
static char g_last_kernel[160] = "";
void note_analysis_kernel(const char* fmt, ...) {
  char buf[sizeof(g_last_kernel)];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lk(g_last_mu);
  memcpy(g_last_kernel, buf, sizeof(buf));
}
}  // namespace mia

extern "C" int mia_last_analysis_kernel(char* buf, int n) {
  if (!buf || n < 1) return MIA_ERR_NULL;
  std::lock_guard<std::mutex> lk(mia::g_last_mu);
  snprintf(buf, (size_t)n, "%s", mia::g_last_kernel);
  return MIA_OK;
}

static const char* const kOptNames[MIA_OPT_COUNT_] = {"cheb_dmax", "cheb_table", "cheb_rowbatch", "cheb_big", "tile",
                                                      "tile_split", "localize_quad", "step_hostwait", "step_lazy_sort",
                                                      "segment_signal", "tile_lists", "bucket_index", "tile_pair", "tile_fused",
                                                      "step_coalesce"};
static const int kOptDefault[MIA_OPT_COUNT_] = {62, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0};

extern "C" int mia_set_option(const char* name, int value) {
  if (!name) return MIA_ERR_NULL;
  for (int i = 0; i < MIA_OPT_COUNT_; ++i)
    if (!strcmp(name, kOptNames[i])) {
      if (i == MIA_OPT_CHEB_DMAX) { if (value < 0) value = kOptDefault[i]; if (value < 3 || value > 62) return MIA_ERR_SIZE; }
      else if (i == MIA_OPT_STEP_COALESCE) { if (value < 0) value = kOptDefault[i]; if (value > 4) return MIA_ERR_SIZE; }
      else value = value < 0 ? kOptDefault[i] : (value != 0);
      mia::g_opt[i].store(value, std::memory_order_relaxed);
      return MIA_OK;
    }
  return MIA_ERR_UNSUPPORTED;
}

extern "C" int mia_get_option(const char* name, int* value) {
  if (!name || !value) return MIA_ERR_NULL;
  for (int i = 0; i < MIA_OPT_COUNT_; ++i)
    if (!strcmp(name, kOptNames[i])) { *value = mia::option(i); return MIA_OK; }
  return MIA_ERR_UNSUPPORTED;
}
