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
  void* ws = nullptr; size_t ws_bytes = 0; int32_t* counters; int32_t* flags; float* Xa; int32_t* host8; void* event = nullptr; void* job = nullptr; void* in_event = nullptr;
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
    if (it % 5 == 4 && id == 5) {
      // the same through the argument block, in one call (ShardedLetkf._run_fast)
      mia_step_args_t a;
      memset(&a, 0, sizeof(a));
      a.X = pr.X; a.G = pr.G; a.m = pr.m; a.k = pr.k; a.Yb = pr.Yb; a.d = pr.d; a.P = pr.P; a.grid_xyz = pr.grid; a.obs_xyz = pr.obs;
      a.n_coord = 1; a.coord_group[0] = 0; a.gc_c[0] = 10.0; a.n_r = 1; a.gc_eps = 1e-5; a.inf_factor = 1.1f; a.method = 0;
      a.p_max_assumed = 20; a.n_chunks = 1; a.Xa = s.Xa; a.flags = s.flags; a.counters = s.counters; a.ws = s.ws; a.ws_bytes = s.ws_bytes;
      a.stream = a.after_stream = a.on_stream = streams[0]; a.comm_stream = streams[1]; a.step_flags = flags; a.host8 = s.host8;
      a.done_event = &s.event;
      int32_t out8[8];
      CHECK(mia_letkf_step_run_args(&a, out8) == MIA_OK);
      continue;
    }
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
    int rc2;
    if (id == 0 || id == 5) {
      // the argument-block submission + one-call collection of ShardedLetkf's steady-state path, every third step timed
      mia_step_args_t a;
      memset(&a, 0, sizeof(a));
      a.X = pr.X; a.G = pr.G; a.m = pr.m; a.k = pr.k; a.Yb = pr.Yb; a.d = pr.d; a.P = pr.P; a.grid_xyz = pr.grid; a.obs_xyz = pr.obs;
      a.n_coord = 1; a.coord_group[0] = 0; a.gc_c[0] = 10.0; a.n_r = 1; a.gc_eps = 1e-5; a.inf_factor = 1.1f; a.gamma = 0.0f; a.method = 0;
      a.p_max_assumed = 20; a.comm = nullptr; a.n_chunks = 1; a.phase = 0; a.Xa = s.Xa; a.flags = s.flags; a.counters = s.counters;
      a.ws = s.ws; a.ws_bytes = s.ws_bytes; a.stream = streams[it % 3 == 0 ? 0 : 3]; a.comm_stream = streams[1]; a.prep_stream = streams[2];
      a.step_flags = flags; a.host8 = s.host8; a.after_stream = a.stream; a.on_stream = streams[1]; a.done_event = &s.event;
      a.caller_stream = nullptr; a.in_event = &s.in_event;
      void *t0 = nullptr, *t1 = nullptr;
      if (it % 3 == 1) { CHECK(mia_timing_event_acquire(&t0) == MIA_OK); CHECK(mia_timing_event_acquire(&t1) == MIA_OK); }
      a.time_start_event = t0; a.time_stop_event = t1;
      rc2 = mia_letkf_step_submit_args(&a, &s.job);
      CHECK(rc2 == MIA_OK);
      if (it % 2 == 0) {          // collected at once, in one call
        int32_t out8[8]; int bn = 0;
        CHECK(mia_letkf_step_collect(s.job, &s.event, s.host8, streams[0], 1, out8, &bn) == MIA_OK); s.job = nullptr;
        CHECK(bn >= 1 && bn <= 4);
      }
      if (t0) { float ms = 0; (void)mia_timing_event_elapsed_ms(t0, t1, &ms); CHECK(mia_timing_event_release(t0) == MIA_OK); CHECK(mia_timing_event_release(t1) == MIA_OK); }
    } else {
      rc2 = mia_letkf_step_submit(pr.X, pr.G, pr.m, pr.k, pr.Yb, pr.d, pr.P, pr.grid, pr.obs, 1, cg, rc, 1, 1e-5, 1.1f, id == 4 ? 0.5f : 0.0f, 0, 20,
                                  comm, chunks, 0, s.Xa, s.flags, s.counters, s.ws, s.ws_bytes, streams[0], streams[1], streams[2 + (it & 1)], flags,
                                  s.host8, comm ? streams[1] : streams[0], streams[1], &s.event, nullptr, nullptr, &s.job);
      CHECK(rc2 == MIA_OK);
    }
    if (id == 1 && it % 6 == 1) { CHECK(mia_set_option("step_coalesce", (it / 6) % 3) == MIA_OK); }      // launch coalescing on / off while steps are in flight
    if (it % 13 == 6) {          // the caller gives a workspace up between steps (a new geometry): release, then reuse the address
      if (s.job) { CHECK(mia_letkf_step_join(s.job) == MIA_OK); } s.job = nullptr;
      CHECK(mia_letkf_step_workspace_release(s.ws) == MIA_OK);
    }
    if (id == 0 && it % 9 == 2) { CHECK(mia_set_option("tile_fused", it & 1) == MIA_OK); }      // route switches while steps are in flight
  }
  for (auto& s : slots) {
    if (s.job) CHECK(mia_letkf_step_join(s.job) == MIA_OK);
    if (s.event) CHECK(mia_event_synchronize(s.event) == MIA_OK);
    CHECK(mia_letkf_step_workspace_release(s.ws) == MIA_OK);
    if (s.event) mia_event_destroy(s.event);
    if (s.in_event) mia_event_destroy(s.in_event);
    hipFree(s.ws); hipFree(s.counters); hipFree(s.flags); hipFree(s.Xa); free(s.host8);
  }
  if (comm) { CHECK(cb_calls.load() > 0); CHECK(mia_comm_destroy(comm) == MIA_OK); }
}

int main(int argc, char** argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 6, n_steps = argc > 2 ? atoi(argv[2]) : 60;
  CHECK(mia_version() == 100);
  char name[160];
  std::vector<std::thread> th;
  for (int i = 0; i < n_threads; ++i) th.emplace_back(caller, i, n_steps);
  for (auto& t : th) t.join();
  CHECK(mia_letkf_step_drain() == MIA_OK);
  CHECK(mia_set_option("tile_fused", -1) == MIA_OK);
  CHECK(mia_set_option("step_coalesce", -1) == MIA_OK);
  long long co_l = 0, co_s = 0, trace[8 * 16];
  CHECK(mia_letkf_step_coalesce_stats(&co_l, &co_s) == MIA_OK && co_s >= co_l);
  CHECK(mia_debug_step_trace(trace, 16) == 16);
  CHECK(mia_last_analysis_kernel(name, (int)sizeof name) == MIA_OK);
  double a = 0, b = 0; long long n = 0;
  CHECK(mia_letkf_step_launch_stats(&a, &b, &n) == MIA_OK);
  printf("driver: %d caller threads x %d steps, %lld steps through the launch threads, %lld kernel launches into the stub, last analysis kernel '%s', "
         "%d failed checks\n", n_threads, n_steps, n, mia_stub_launch_count(), name, g_fail.load());
  return g_fail.load() ? 1 : 0;
}
