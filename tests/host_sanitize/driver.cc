// Drives the host side of the step driver (csrc/sharded_step.hip and what it calls) under a sanitizer, against the stub HIP runtime
// of hip_stub.cc: several caller threads, each with its own rotating pipeline slots (workspace, counters, events), submit steps to
// the library's launch threads (mia_letkf_step_submit / _join), mix in synchronous steps, redo calls (phase 1), geometry epochs, the
// list route, a custom communicator with pieces, option changes and workspace releases -- the call pattern of
// torch-assimilate_amd/sharded.py and then some.  Kernels do nothing here: what is checked is memory safety and data races of the
// host logic (tests/test_host_sanitizers.py).  Exit status 0 = every call returned what it should.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "mia_letkf.h"

extern "C" long long mia_stub_launch_count();
extern "C" long long mia_stub_live_allocations();
extern "C" int hipMalloc(void**, size_t);
extern "C" int hipFree(void*);

static std::atomic<int> g_fail{0};
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "driver: %s failed at line %d\n", #cond, __LINE__); ++g_fail; } } while (0)

static void* dmalloc(size_t n) { void* p = nullptr; CHECK(hipMalloc(&p, n) == 0); return p; }

struct Problem {
  int64_t G, P; int m, k;
  float *X, *Yb, *d; double *grid, *obs;
  Problem(int64_t G_, int64_t P_, int m_, int k_) : G(G_), P(P_), m(m_), k(k_) {
    X = (float*)dmalloc(sizeof(float) * m * k * G); Yb = (float*)dmalloc(sizeof(float) * k * P); d = (float*)dmalloc(sizeof(float) * P);
    grid = (double*)dmalloc(sizeof(double) * G); obs = (double*)dmalloc(sizeof(double) * P);
    for (int64_t g = 0; g < G; ++g) grid[g] = (double)g;
    for (int64_t j = 0; j < P; ++j) obs[j] = 2.0 * (double)j;
  }
  ~Problem() { hipFree(X); hipFree(Yb); hipFree(d); hipFree(grid); hipFree(obs); }
};

struct Slot {
  void* ws = nullptr; size_t ws_bytes = 0; int32_t* counters; int32_t* flags; float* Xa; int32_t* host8; void* event = nullptr; void* job = nullptr;
};

static int allgather_cb(void* ctx, const void* send, void* recv, size_t bytes, void*) {
  ++*(std::atomic<int>*)ctx;
  memcpy(recv, send, bytes);      // world 1
  return 0;
}
static int allreduce_cb(void* ctx, int32_t*, int, void*) { ++*(std::atomic<int>*)ctx; return 0; }

static void caller(int id, int n_steps) {
  Problem pr(4000 + 160 * id, 2000 + 80 * id, 1 + (id & 1), 40);
  const int32_t cg[1] = {0};
  const double rc[1] = {10.0};
  const int n_slots = 3;
  std::vector<Slot> slots(n_slots);
  std::atomic<int> cb_calls{0};
  mia_comm_t* comm = nullptr;
  const int chunks = (id == 2) ? 2 : 1;
  if (id == 2) CHECK(mia_comm_create_custom(0, 1, allgather_cb, allreduce_cb, &cb_calls, &comm) == MIA_OK);
  for (auto& s : slots) {
    CHECK(mia_letkf_sharded_step_workspace_bytes(pr.G, pr.m, pr.k, pr.P, 1, 1, chunks, 20, &s.ws_bytes) == MIA_OK);
    s.ws = dmalloc(s.ws_bytes);
    s.counters = (int32_t*)dmalloc(32); s.flags = (int32_t*)dmalloc(sizeof(int32_t) * pr.G);
    s.Xa = (float*)dmalloc(sizeof(float) * pr.m * pr.k * pr.G); s.host8 = (int32_t*)calloc(8, sizeof(int32_t));
  }
  void* streams[4] = {(void*)(uintptr_t)(0x1000 + 16 * id), (void*)(uintptr_t)(0x1004 + 16 * id), (void*)(uintptr_t)(0x1008 + 16 * id),
                      (void*)(uintptr_t)(0x100c + 16 * id)};
  for (int it = 0; it < n_steps; ++it) {
    Slot& s = slots[it % n_slots];
    if (s.job) { CHECK(mia_letkf_step_join(s.job) == MIA_OK); s.job = nullptr; CHECK(mia_event_synchronize(s.event) == MIA_OK); }
    int flags = MIA_STEP_NO_JOIN | (it >= n_slots ? MIA_STEP_WS_CLEAN : 0);
    if (id == 1 && it % 7 == 3) flags |= MIA_STEP_NO_TILE_LISTS;            // the per-point list route now and then
    if (id == 3) flags |= (it >= n_slots ? MIA_STEP_REUSE_LISTS : 0) | MIA_STEP_KEEP_LISTS;      // a geometry epoch
    if (id == 0 && it % 11 == 5) flags |= MIA_STEP_SCAN_INDEX;
    if (it % 5 == 4) {
      // a synchronous step on this slot (what ShardedLetkf.assimilate does): the queued ones first
      CHECK(mia_letkf_step_drain() == MIA_OK);
      int rc1 = mia_letkf_sharded_step_streams_f32(pr.X, pr.G, pr.m, pr.k, pr.Yb, pr.d, pr.P, pr.grid, pr.obs, 1, cg, rc, 1, 1e-5, 1.1f, 0.0f, 0, 20,
                                                    comm, chunks, 0, s.Xa, s.flags, s.counters, s.ws, s.ws_bytes, streams[0], streams[1], nullptr,
                                                    flags & ~MIA_STEP_NO_JOIN);
      CHECK(rc1 == MIA_OK);
      CHECK(mia_letkf_step_readback(s.counters, s.host8, streams[0], streams[0], &s.event) == MIA_OK);
      if (it % 10 == 9) {      // ... and a redo of declined points (phase 1)
        rc1 = mia_letkf_sharded_step_streams_f32(pr.X, pr.G, pr.m, pr.k, pr.Yb, pr.d, pr.P, pr.grid, pr.obs, 1, cg, rc, 1, 1e-5, 1.1f, 0.0f, 0, 20,
                                                  comm, chunks, 1, s.Xa, s.flags, s.counters, s.ws, s.ws_bytes, streams[0], streams[1], nullptr,
                                                  flags & ~MIA_STEP_NO_JOIN);
        CHECK(rc1 == MIA_OK);
      }
      continue;
    }
    int rc2 = mia_letkf_step_submit(pr.X, pr.G, pr.m, pr.k, pr.Yb, pr.d, pr.P, pr.grid, pr.obs, 1, cg, rc, 1, 1e-5, 1.1f, id == 4 ? 0.5f : 0.0f, 0, 20,
                                    comm, chunks, 0, s.Xa, s.flags, s.counters, s.ws, s.ws_bytes, streams[0], streams[1], streams[2 + (it & 1)], flags,
                                    s.host8, comm ? streams[1] : streams[0], streams[1], &s.event, nullptr, nullptr, &s.job);
    CHECK(rc2 == MIA_OK);
    if (it % 13 == 6) {          // the caller gives a workspace up between steps (a new geometry): release, then reuse the address
      CHECK(mia_letkf_step_join(s.job) == MIA_OK); s.job = nullptr;
      CHECK(mia_letkf_step_workspace_release(s.ws) == MIA_OK);
    }
    if (id == 0 && it % 9 == 2) { CHECK(mia_set_option("tile_fused", it & 1) == MIA_OK); }      // route switches while steps are in flight
  }
  for (auto& s : slots) {
    if (s.job) CHECK(mia_letkf_step_join(s.job) == MIA_OK);
    if (s.event) CHECK(mia_event_synchronize(s.event) == MIA_OK);
    CHECK(mia_letkf_step_workspace_release(s.ws) == MIA_OK);
    if (s.event) mia_event_destroy(s.event);
    hipFree(s.ws); hipFree(s.counters); hipFree(s.flags); hipFree(s.Xa); free(s.host8);
  }
  if (comm) { CHECK(cb_calls.load() > 0); CHECK(mia_comm_destroy(comm) == MIA_OK); }
}

int main(int argc, char** argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 5, n_steps = argc > 2 ? atoi(argv[2]) : 60;
  CHECK(mia_version() == 100);
  char name[160];
  std::vector<std::thread> th;
  for (int i = 0; i < n_threads; ++i) th.emplace_back(caller, i, n_steps);
  for (auto& t : th) t.join();
  CHECK(mia_letkf_step_drain() == MIA_OK);
  CHECK(mia_set_option("tile_fused", -1) == MIA_OK);
  CHECK(mia_last_analysis_kernel(name, (int)sizeof name) == MIA_OK);
  double a = 0, b = 0; long long n = 0;
  CHECK(mia_letkf_step_launch_stats(&a, &b, &n) == MIA_OK);
  printf("driver: %d caller threads x %d steps, %lld steps through the launch threads, %lld kernel launches into the stub, last analysis kernel '%s', "
         "%d failed checks\n", n_threads, n_steps, n, mia_stub_launch_count(), name, g_fail.load());
  return g_fail.load() ? 1 : 0;
}
