// Native driver of one assimilation step of one rank's block of grid points (single C call per step):
//   pack records -> observation cell index + Gaspari-Cohn neighbour lists -> local analysis, and for world > 1
//   the block is analysed in chunks whose RCCL all-gather (+ the copy into the (m, k, G) result) runs on a
//   second HIP stream while the next chunk is analysed.
// Replaces the per-step Python orchestration (≈15 launches through ctypes / torch.distributed cost more host
// time than the ≈0.35 ms of GPU work they enqueue).  The step it drives is the reference's
// DaskLocalization -> localized_etkf -> apply_weights path (pytassim/interface/letkf.py:102-137,
// etkf.py:169-207); the reference has no multi-device path, its unit of distribution is the
// dask chunk of grid points (letkf.py:118-131), which is the block / chunk here.
//
// RCCL is bound at run time (dlopen of the library the process already uses, normally torch's bundled
// librccl.so) so that this library keeps loading on machines without RCCL and never pulls in a second
// HIP runtime.
#include <dlfcn.h>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include "mia_common.h"
#include "mia_kernels.h"
#include "mia_options.h"
#include "mia_pack_dev.h"
#include "mia_tiles.h"

// ---- the few RCCL declarations needed (ABI of rccl.h 2.x: opaque comm, 128-byte id, C enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclInt32 = 2, kNcclFloat32 = 7, kNcclMax = 2, kNcclUint8 = 1 };
typedef int (*pfn_ncclGetUniqueId)(ncclUniqueId*);
typedef int (*pfn_ncclCommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*pfn_ncclCommDestroy)(ncclComm_t);
typedef int (*pfn_ncclAllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
typedef int (*pfn_ncclAllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef const char* (*pfn_ncclGetErrorString)(int);

namespace {

struct RcclApi {
  void* handle = nullptr;
  pfn_ncclGetUniqueId GetUniqueId = nullptr;
  pfn_ncclCommInitRank CommInitRank = nullptr;
  pfn_ncclCommDestroy CommDestroy = nullptr;
  pfn_ncclAllGather AllGather = nullptr;
  pfn_ncclAllReduce AllReduce = nullptr;
  pfn_ncclGetErrorString GetErrorString = nullptr;
} g_rccl;

char g_comm_error[512] = "";

void set_error(const char* what, int code) {
  const char* msg = (g_rccl.GetErrorString && code > 0) ? g_rccl.GetErrorString(code) : "";
  snprintf(g_comm_error, sizeof(g_comm_error), "%s (code %d) %s", what, code, msg);
}

constexpr int kMaxChunks = 16;   // pieces per block; events: [c] piece c, [kMaxChunks-1] records packed (pieces <= 15)
constexpr int kMaxRanks = 16, kMaxSlots = 8;     // direct exchange: ranks of one node, steps in flight
// synchronisation area of one rank, in uint32 words: per slot [kMaxRanks] ready, [kMaxRanks] free, [kMaxRanks][4] counters
constexpr int kSyncSlotWords = kMaxRanks * 6;
constexpr size_t kSyncBytes = (size_t)kMaxSlots * kSyncSlotWords * sizeof(uint32_t);

}  // namespace

struct mia_comm {
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  mia_allgather_fn ag = nullptr;
  mia_allreduce_max_i32_fn ar = nullptr;
  void* ctx = nullptr;
  hipEvent_t ev[kMaxChunks + 2] = {};
  hipEvent_t evp[kMaxChunks] = {};      // piece c gathered (exchange stream -> placement stream)
  hipStream_t place_stream = nullptr;   // optional: mia_comm_set_place_stream
  int n_ev = 0;
  // ---- direct (peer-mapped) exchange, see "Direct exchange" below
  int peer_slots = 0;                    // result buffers this rank owns (one per step in flight)
  size_t peer_bytes = 0;                 // bytes of one result buffer
  float* peer_buf[kMaxRanks][kMaxSlots] = {};   // [rank][slot]: rank's result buffers as mapped into this process
  uint32_t* peer_sync[kMaxRanks] = {};   // [rank]: its synchronisation area (fine-grained device memory)
  bool peer_owned = false;               // buffers of `rank` were allocated by mia_comm_peer_alloc (freed on destroy)
  bool peer_opened[kMaxRanks] = {};      // mapped through hipIpcOpenMemHandle (closed on destroy)
  int peer_ready = 0;                    // every rank attached
  uint32_t peer_seq[kMaxSlots] = {};     // exchanges done per slot (the sequence number the flags carry)
  // bound of the device-side waits for the peers' flags, in polls of ~1-2 us (mia_comm_peer_wait_bound).  Ranks of a real run
  // drift apart by seconds (I/O of one rank between steps, a first-step table build, a debugger): the default is ~1 minute --
  // an RCCL collective would simply wait; a waiter that gives up raises error bit 2, it never hangs the grid
  int peer_wait_polls = 1 << 25;
};

namespace {

int comm_events(mia_comm* c) {
  if (c->n_ev) return MIA_OK;
  for (int i = 0; i < kMaxChunks + 2; ++i) MIA_HIP_TRY(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
  for (int i = 0; i < kMaxChunks; ++i) MIA_HIP_TRY(hipEventCreateWithFlags(&c->evp[i], hipEventDisableTiming));
  c->n_ev = kMaxChunks + 2;
  return MIA_OK;
}

int comm_allgather(mia_comm* c, const void* send, void* recv, size_t bytes, hipStream_t s) {
  if (c->ag) return c->ag(c->ctx, send, recv, bytes, (void*)s) == 0 ? MIA_OK : MIA_ERR_COMM;
  int rc = g_rccl.AllGather(send, recv, bytes, kNcclUint8, c->nccl, s);
  if (rc != kNcclSuccess) { set_error("ncclAllGather failed", rc); return MIA_ERR_COMM; }
  return MIA_OK;
}

int comm_allreduce_max(mia_comm* c, int32_t* buf, int n, hipStream_t s) {
  if (c->ar) return c->ar(c->ctx, buf, n, (void*)s) == 0 ? MIA_OK : MIA_ERR_COMM;
  int rc = g_rccl.AllReduce(buf, buf, (size_t)n, kNcclInt32, kNcclMax, c->nccl, s);
  if (rc != kNcclSuccess) { set_error("ncclAllReduce failed", rc); return MIA_ERR_COMM; }
  return MIA_OK;
}

// gathered chunk [world][rows][nc]  ->  result rows [rows][G] at columns r * n + off + i
// (i < nc, off + i < n, column < G).  x: column (4 per thread when everything is 4-aligned), y: row, z: rank.
// Every rank's piece carries a 16-byte trailer {longest list, truncated lists, declined points, error bits};
// with ctr_out the first thread also folds the trailers: ctr_out[0..3] = this rank's, [4..7] = max over ranks
// (the all-reduce of the redo decision rides on the last piece's all-gather instead of being a collective).
// Single-wave workgroups, four column groups per lane: with steps in flight this kernel runs beside a later step's
// analysis kernel, which fills every SIMD's register file -- a lone wave takes the slot of the next analysis wave that
// retires, a 4-wave workgroup waits for one to retire on every SIMD of a CU at once (see localize.hip).
constexpr int kPlaceThreads = 64, kPlaceUnroll = 4;
template <int VEC>
__global__ void __launch_bounds__(kPlaceThreads) place_chunk_kernel(const float* __restrict__ gath, float* __restrict__ out,
                                                          int64_t G, int64_t n, int64_t off, int nc, int rows,
                                                          size_t rank_stride /* floats */, int32_t* ctr_out, int rank) {
  const int r = blockIdx.z, row = blockIdx.y;
  if (ctr_out && blockIdx.x == 0 && row == 0 && r == 0 && threadIdx.x < 4) {
    int mx = 0, own = 0;
    for (int q = 0; q < (int)gridDim.z; ++q) {
      const int v = reinterpret_cast<const int32_t*>(gath + (size_t)q * rank_stride + (size_t)rows * nc)[threadIdx.x];
      mx = q == 0 ? v : (threadIdx.x == 3 ? (mx | v) : (v > mx ? v : mx));
      if (q == rank) own = v;
    }
    ctr_out[threadIdx.x] = own;
    ctr_out[4 + threadIdx.x] = mx;
  }
#pragma unroll
  for (int u = 0; u < kPlaceUnroll; ++u) {
  const int i = ((blockIdx.x * kPlaceUnroll + u) * kPlaceThreads + threadIdx.x) * VEC;
  if (i >= nc) return;
  const int64_t in_block = off + i;
  const int64_t col = (int64_t)r * n + in_block;
  const float* src = gath + (size_t)r * rank_stride + (size_t)row * (size_t)nc + i;
  float* dst = out + (size_t)row * (size_t)G + col;
  if (VEC == 4) {
    if (in_block + 3 < n && col + 3 < G) {
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
      continue;
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    if (i + v < nc && in_block + v < n && col + v < G) dst[v] = src[v];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct exchange.  The all-gather of the analysis ensemble moves world x (m k n) floats into every rank; RCCL's ring
// forwards each block hop by hop (per-link bound, (world - 1) latencies) into a staging buffer that a placement kernel then
// copies into the (m, k, G) result.  xGMI is point to point, so every rank can instead WRITE ITS BLOCK STRAIGHT INTO THE
// RESULT BUFFER OF ALL PEERS, over its world - 1 links at once: the result buffers are library-owned, exported with
// hipIpcGetMemHandle and mapped by every rank of the node.  Per step and slot, with sequence number q:
//   submit       free[slot][me] = q in every peer's sync area: "my buffer `slot` may be overwritten for step q" (its previous
//                result was collected, or the caller would not reuse the slot)
//   analysis     this rank's block, written into its own result buffer (no staging)
//   exchange stream:  wait  free[slot][p] >= q for all peers
//                     push  block (16-byte accesses) + this rank's four redo counters -> every peer
//                     signal ready[slot][me] = q in every peer's sync area (a kernel of its own: the push kernel's end is the
//                            system-scope release of its stores)
//                     wait  ready[slot][p] >= q for all peers; fold the counters (max over ranks)
// Flags live in fine-grained device memory and are accessed with system-scope atomics; waits are bounded (error bit 1 of
// counters[3] / [7], never a hung grid).  No collective, no staging copy, no placement kernel: 2 x block bytes of local HBM
// traffic instead of 2 x world x block.  RCCL stays the fallback (and the route of the first, exact-list step).
struct PeerPtrs { float* buf[kMaxRanks]; uint32_t* sync[kMaxRanks]; };

__global__ void __launch_bounds__(64) peer_flag_kernel(PeerPtrs pp, int world, int word, uint32_t value) {
  const int p = threadIdx.x;      // one lane per rank (own area included: keeps the arithmetic uniform)
  if (p < world) __hip_atomic_store(pp.sync[p] + word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) peer_wait_kernel(const uint32_t* flags /* [kMaxRanks] of this rank's area */, int world,
                                                       int rank, uint32_t seq, int32_t* err, int max_polls,
                                                       const int32_t* ctr_all /* [kMaxRanks][4] or null */, int32_t* counters) {
  const int p = threadIdx.x;
  bool ok = false;
  for (int poll = 0; poll < max_polls; ++poll) {
    const uint32_t v = (p < world && p != rank) ? __hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) : seq;
    ok = (int32_t)(v - seq) >= 0;
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(32);
  }
  if (!__all(ok) && p == 0) atomicOr(err, 2);      // exit condition every wave reaches: ~seconds, then report
  if (ctr_all && p < 4) {                             // redo decision: max over the ranks' counters (or of the error bits)
    int mx = 0;
    for (int q = 0; q < world; ++q) {
      const int v = __hip_atomic_load(ctr_all + q * 4 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      mx = q == 0 ? v : (p == 3 ? (mx | v) : (v > mx ? v : mx));
    }
    counters[4 + p] = p == 3 ? (mx | counters[3]) : mx;
  }
}

// block [rows][n] at column b0 of the (rows, G) result -> the same place in every peer's buffer; blockIdx.z = peer
__global__ void __launch_bounds__(256) peer_push_kernel(PeerPtrs pp, int world, int rank, int64_t G, int64_t b0, int64_t n,
                                                        int rows, int slot_word0, const int32_t* own_counters) {
  int peer = blockIdx.z;
  if (peer >= rank) ++peer;                           // (world - 1 peers)
  const float* src = pp.buf[rank] + (size_t)blockIdx.y * G + b0;
  float* dst = pp.buf[peer] + (size_t)blockIdx.y * G + b0;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4) {      // this rank's counters: to the peer, and (once) to itself
    const int32_t v = own_counters[threadIdx.x];
    __hip_atomic_store(reinterpret_cast<int32_t*>(pp.sync[peer]) + slot_word0 + 2 * kMaxRanks + 4 * rank + threadIdx.x, v,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (blockIdx.z == 0)
      __hip_atomic_store(reinterpret_cast<int32_t*>(pp.sync[rank]) + slot_word0 + 2 * kMaxRanks + 4 * rank + threadIdx.x, v,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 16-byte accesses where source and destination rows are aligned alike (b0, G multiples of 4), scalars otherwise
  const bool vec = ((G | b0) & 3) == 0 && ((reinterpret_cast<uintptr_t>(pp.buf[rank]) | reinterpret_cast<uintptr_t>(pp.buf[peer])) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
  }
}

int peer_slot_of(const mia_comm* c, const float* Xa) {
  if (!c || !c->peer_ready) return -1;
  for (int s = 0; s < c->peer_slots; ++s)
    if (c->peer_buf[c->rank][s] == Xa) return s;
  return -1;
}

PeerPtrs peer_ptrs(const mia_comm* c, int slot) {
  PeerPtrs pp;
  for (int r = 0; r < kMaxRanks; ++r) {
    pp.buf[r] = r < c->world ? c->peer_buf[r][slot] : nullptr;
    pp.sync[r] = r < c->world ? c->peer_sync[r] : nullptr;
  }
  return pp;
}

// first half of an exchange: new sequence number, "my buffer of this slot may be overwritten" to every peer (stream ps)
int peer_begin(mia_comm* c, int slot, hipStream_t ps, uint32_t* seq_out) {
  const uint32_t seq = ++c->peer_seq[slot];
  peer_flag_kernel<<<1, 64, 0, ps>>>(peer_ptrs(c, slot), c->world, slot * kSyncSlotWords + kMaxRanks + c->rank, seq);
  MIA_LAUNCH_CHECK();
  *seq_out = seq;
  return MIA_OK;
}

// second half, on the exchange stream cs (the caller has ordered it behind the block's producer): wait for the peers'
// buffers, push block [rows][b0, b1) and the four counters, signal, wait for the peers' blocks, fold the counters
int peer_finish(mia_comm* c, int slot, uint32_t seq, int64_t G, int64_t b0, int64_t b1, int rows, int32_t* counters,
                hipStream_t cs) {
  const int world = c->world, rank = c->rank, sw0 = slot * kSyncSlotWords;
  const PeerPtrs pp = peer_ptrs(c, slot);
  const uint32_t* my = c->peer_sync[rank] + sw0;
  peer_wait_kernel<<<1, 64, 0, cs>>>(my + kMaxRanks, world, rank, seq, counters + 3, c->peer_wait_polls, nullptr, nullptr);
  MIA_LAUNCH_CHECK();
  const int64_t nb = b1 > b0 ? b1 - b0 : 0;
  unsigned gx = (unsigned)((nb / 4 + 255) / 256);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  peer_push_kernel<<<dim3(gx, (unsigned)(nb ? rows : 1), (unsigned)(world - 1)), 256, 0, cs>>>(pp, world, rank, G, nb ? b0 : 0, nb,
                                                                                          rows, sw0, counters);
  MIA_LAUNCH_CHECK();
  peer_flag_kernel<<<1, 64, 0, cs>>>(pp, world, sw0 + rank, seq);
  MIA_LAUNCH_CHECK();
  peer_wait_kernel<<<1, 64, 0, cs>>>(my, world, rank, seq, counters + 3, c->peer_wait_polls,
                                     reinterpret_cast<const int32_t*>(my + 2 * kMaxRanks), counters);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

struct StepLayout {
  size_t rec, loc, cnt, idx, w, done, bufs, gath, total;
  size_t hrec, tl, scratch;      // tile route: split records (P + 1), tile lists of the block, 256 bytes of scratch counters
  size_t loc_bytes, send_bytes, chunk_bytes;
  int cap, ut;
  int64_t n, nc;
};

int step_layout(int64_t G, int m, int k, int64_t P, int n_coord, int world, int n_chunks, int p_max_assumed,
                StepLayout* L, int tile_extra = 0) {
  if (G < 0 || m <= 0 || k <= 0 || P < 0 || n_coord <= 0 || world <= 0 || n_chunks <= 0 || n_chunks > kMaxChunks - 1 ||
      p_max_assumed < 0)
    return MIA_ERR_SIZE;
  const int kp = (k + 1 + 3) / 4 * 4;
  L->n = (G + world - 1) / world;
  L->nc = ((L->n + n_chunks - 1) / n_chunks + 15) / 16 * 16;   // (whole tiles of sixteen points per piece)
  L->cap = p_max_assumed < 8 ? 8 : (p_max_assumed + 7) / 8 * 8;
  size_t o = 0;
  L->rec = o; o = mia::align_up(o + (size_t)(P > 0 ? P : 1) * kp * sizeof(float), 256);
  int rc = mia_letkf_localize_workspace_bytes(P, n_coord, &L->loc_bytes);
  if (rc != MIA_OK) return rc;
  L->loc = o; o = mia::align_up(o + L->loc_bytes, 256);
  L->cnt = o; o = mia::align_up(o + (size_t)L->n * sizeof(int32_t), 256);
  L->idx = o; o = mia::align_up(o + (size_t)L->n * L->cap * sizeof(int32_t), 256);
  L->w = o; o = mia::align_up(o + (size_t)L->n * L->cap * sizeof(double), 256);
  L->done = o; o = mia::align_up(o + (size_t)kMaxChunks * 64 * mia::kSlotStride * sizeof(int32_t), 256);
  // (the workspace is sized for the largest tile lists the ensemble size allows, so that a caller may add slots -- MIA_STEP_TILE_EXTRA
  //  -- without a new workspace query)
  const int ut0 = mia::tile_ut_for(p_max_assumed < L->cap ? p_max_assumed : L->cap), kt = (k + 15) >> 4;
  const int ut_most = kt + 1 < 6 ? kt + 1 : 6;
  L->ut = ut0 + tile_extra;
  L->hrec = L->tl = o;
  if (ut0 <= ut_most) {
    L->hrec = o; o = mia::align_up(o + (size_t)(P + 1) * mia::split_rec_bytes(k), 256);
    L->tl = o; o = mia::align_up(o + mia::tile_list_layout(L->n, ut_most).bytes, 256);
  }
  if (L->ut > ut_most) L->ut = 7;      // (no tile route)
  L->scratch = o; o += 256;
  L->bufs = L->gath = o;
  L->send_bytes = (size_t)m * k * L->nc * sizeof(float) + 16;   // piece + counter trailer
  L->chunk_bytes = mia::align_up(L->send_bytes, 256);
  if (world > 1 || n_chunks > 1) {
    L->bufs = o; o += L->chunk_bytes * n_chunks;
    L->gath = o; o += mia::align_up(L->send_bytes * world, 256) * n_chunks;
  }
  L->total = o;
  return MIA_OK;
}

}  // namespace

static int prep_event_fwd(hipEvent_t* ev);      // (ring of ordering events, defined with the step driver below)

extern "C" const char* mia_comm_last_error(void) { return g_comm_error; }

extern "C" int mia_comm_load(const char* rccl_path) {
  if (g_rccl.handle) return MIA_OK;
  const char* path = (rccl_path && rccl_path[0]) ? rccl_path : "librccl.so";
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf(g_comm_error, sizeof(g_comm_error), "dlopen(%s) failed: %s", path, dlerror());
    return MIA_ERR_COMM;
  }
  RcclApi api;
  api.handle = h;
  api.GetUniqueId = (pfn_ncclGetUniqueId)dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (pfn_ncclCommInitRank)dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (pfn_ncclCommDestroy)dlsym(h, "ncclCommDestroy");
  api.AllGather = (pfn_ncclAllGather)dlsym(h, "ncclAllGather");
  api.AllReduce = (pfn_ncclAllReduce)dlsym(h, "ncclAllReduce");
  api.GetErrorString = (pfn_ncclGetErrorString)dlsym(h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce) {
    snprintf(g_comm_error, sizeof(g_comm_error), "%s does not export the RCCL collectives", path);
    return MIA_ERR_COMM;
  }
  g_rccl = api;
  return MIA_OK;
}

extern "C" int mia_comm_unique_id(void* id128) {
  if (!id128) return MIA_ERR_NULL;
  if (!g_rccl.handle) { set_error("mia_comm_load was not called", 0); return MIA_ERR_COMM; }
  ncclUniqueId id;
  int rc = g_rccl.GetUniqueId(&id);
  if (rc != kNcclSuccess) { set_error("ncclGetUniqueId failed", rc); return MIA_ERR_COMM; }
  memcpy(id128, id.internal, 128);
  return MIA_OK;
}

extern "C" int mia_comm_create(const void* id128, int rank, int world, mia_comm_t** out) {
  if (!id128 || !out) return MIA_ERR_NULL;
  if (world <= 0 || rank < 0 || rank >= world) return MIA_ERR_SIZE;
  if (!g_rccl.handle) { set_error("mia_comm_load was not called", 0); return MIA_ERR_COMM; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  mia_comm* c = new mia_comm();
  c->rank = rank;
  c->world = world;
  int rc = g_rccl.CommInitRank(&c->nccl, world, id, rank);
  if (rc != kNcclSuccess) { set_error("ncclCommInitRank failed", rc); delete c; return MIA_ERR_COMM; }
  *out = c;
  return MIA_OK;
}

extern "C" int mia_comm_create_custom(int rank, int world, mia_allgather_fn allgather,
                                      mia_allreduce_max_i32_fn allreduce_max, void* ctx, mia_comm_t** out) {
  if (!allgather || !allreduce_max || !out) return MIA_ERR_NULL;
  if (world <= 0 || rank < 0 || rank >= world) return MIA_ERR_SIZE;
  mia_comm* c = new mia_comm();
  c->rank = rank;
  c->world = world;
  c->ag = allgather;
  c->ar = allreduce_max;
  c->ctx = ctx;
  *out = c;
  return MIA_OK;
}

// a communicator that only carries the block partition (rank, world): for steps whose analysis STAYS block-sharded
// (MIA_STEP_NO_GATHER -- what the reference's dask chunks along `grid` do, interface/letkf.py:118-131); no exchange can run on it
extern "C" int mia_comm_create_partition(int rank, int world, mia_comm_t** out) {
  if (!out) return MIA_ERR_NULL;
  if (world <= 0 || rank < 0 || rank >= world) return MIA_ERR_SIZE;
  mia_comm* c = new mia_comm();
  c->rank = rank;
  c->world = world;
  *out = c;
  return MIA_OK;
}

extern "C" int mia_comm_set_place_stream(mia_comm_t* c, void* stream) {
  if (!c) return MIA_ERR_NULL;
  c->place_stream = (hipStream_t)stream;
  return MIA_OK;
}

// ---- direct exchange: buffers, handles, attachment (protocol: see "Direct exchange" above)
extern "C" int mia_comm_peer_alloc(mia_comm_t* c, size_t result_bytes, int n_slots, void* ipc_handles_out) {
  if (!c) return MIA_ERR_NULL;
  if (n_slots < 1 || n_slots > kMaxSlots || result_bytes == 0 || c->world > kMaxRanks) return MIA_ERR_SIZE;
  if (c->peer_slots) return MIA_ERR_UNSUPPORTED;          // one allocation per communicator
  (void)hipGetLastError();
  hipIpcMemHandle_t* hs = reinterpret_cast<hipIpcMemHandle_t*>(ipc_handles_out);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle table of mia_comm_peer_alloc / _open is 64 bytes per entry");
  for (int s = 0; s < n_slots; ++s) {
    void* b = nullptr;
    if (hipMalloc(&b, mia::align_up(result_bytes, 256)) != hipSuccess) { set_error("hipMalloc of a result buffer failed", 0); (void)hipGetLastError(); return MIA_ERR_COMM; }
    c->peer_buf[c->rank][s] = (float*)b;
    c->peer_slots = s + 1;
    c->peer_owned = true;
    if (hs && hipIpcGetMemHandle(&hs[s], b) != hipSuccess) { set_error("hipIpcGetMemHandle(result buffer) failed", 0); (void)hipGetLastError(); return MIA_ERR_COMM; }
  }
  void* sy = nullptr;
  if (hipExtMallocWithFlags(&sy, kSyncBytes, hipDeviceMallocFinegrained) != hipSuccess) { set_error("fine-grained allocation of the sync area failed", 0); (void)hipGetLastError(); return MIA_ERR_COMM; }
  c->peer_sync[c->rank] = (uint32_t*)sy;
  if (hipMemset(sy, 0, kSyncBytes) != hipSuccess) { (void)hipGetLastError(); return MIA_ERR_COMM; }
  if (hs && hipIpcGetMemHandle(&hs[n_slots], sy) != hipSuccess) { set_error("hipIpcGetMemHandle(sync area) failed", 0); (void)hipGetLastError(); return MIA_ERR_COMM; }
  c->peer_bytes = result_bytes;
  if (c->world == 1) c->peer_ready = 1;
  return MIA_OK;
}

static void peer_check_ready(mia_comm* c) {
  int ok = c->peer_slots > 0;
  for (int r = 0; r < c->world && ok; ++r) {
    ok = c->peer_sync[r] != nullptr;
    for (int s = 0; s < c->peer_slots && ok; ++s) ok = c->peer_buf[r][s] != nullptr;
  }
  c->peer_ready = ok;
}

// all_handles: [world][n_slots + 1] handles as every rank's mia_comm_peer_alloc filled them (any all-gather of the host's)
extern "C" int mia_comm_peer_open(mia_comm_t* c, const void* all_handles) {
  if (!c || !all_handles) return MIA_ERR_NULL;
  if (!c->peer_slots) return MIA_ERR_SIZE;
  (void)hipGetLastError();
  const hipIpcMemHandle_t* hs = reinterpret_cast<const hipIpcMemHandle_t*>(all_handles);
  const int per = c->peer_slots + 1;
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    for (int s = 0; s < per; ++s) {
      void* ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, hs[(size_t)r * per + s], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        set_error("hipIpcOpenMemHandle failed", r);
        (void)hipGetLastError();
        return MIA_ERR_COMM;
      }
      if (s < c->peer_slots) c->peer_buf[r][s] = (float*)ptr; else c->peer_sync[r] = (uint32_t*)ptr;
    }
    c->peer_opened[r] = true;
  }
  peer_check_ready(c);
  return c->peer_ready ? MIA_OK : MIA_ERR_COMM;
}

// in-process attachment of a peer's buffers (ranks that share an address space: tests, one process driving several GPUs)
extern "C" int mia_comm_peer_attach(mia_comm_t* c, int peer, void* const* result_bufs, void* sync_area) {
  if (!c || !result_bufs || !sync_area) return MIA_ERR_NULL;
  if (peer < 0 || peer >= c->world || peer == c->rank || !c->peer_slots) return MIA_ERR_SIZE;
  for (int s = 0; s < c->peer_slots; ++s) c->peer_buf[peer][s] = (float*)result_bufs[s];
  c->peer_sync[peer] = (uint32_t*)sync_area;
  peer_check_ready(c);
  return MIA_OK;
}

extern "C" int mia_comm_peer_wait_bound(mia_comm_t* c, int log2_polls) {
  if (!c) return MIA_ERR_NULL;
  if (log2_polls < 10 || log2_polls > 30) return MIA_ERR_SIZE;
  c->peer_wait_polls = 1 << log2_polls;
  return MIA_OK;
}

extern "C" void* mia_comm_peer_buffer(mia_comm_t* c, int slot) {
  return (c && slot >= 0 && slot < c->peer_slots) ? (void*)c->peer_buf[c->rank][slot] : nullptr;
}
extern "C" void* mia_comm_peer_sync_area(mia_comm_t* c) { return c ? (void*)c->peer_sync[c->rank] : nullptr; }

// The exchange alone: block [rows][b0, b1) of result buffer `slot` (already written by work enqueued on `stream`) goes to
// every peer; when `stream` has passed this call, the peers' blocks have landed in this rank's buffer and counters[4..7]
// hold the maximum over the ranks of everybody's counters[0..3] (device int32[8]).  All ranks call it in the same order.
extern "C" int mia_comm_peer_exchange(mia_comm_t* c, int slot, int rows, int64_t G, int64_t b0, int64_t b1, int32_t* counters,
                                      void* stream) {
  if (!c || !counters) return MIA_ERR_NULL;
  if (!c->peer_ready || slot < 0 || slot >= c->peer_slots || rows < 1 || G < 1 || b0 < 0 || b1 > G) return MIA_ERR_SIZE;
  if ((size_t)rows * G * sizeof(float) > c->peer_bytes) return MIA_ERR_SIZE;
  if (c->world == 1) return MIA_OK;
  (void)hipGetLastError();
  uint32_t seq = 0;
  int rc = peer_begin(c, slot, (hipStream_t)stream, &seq);
  if (rc != MIA_OK) return rc;
  return peer_finish(c, slot, seq, G, b0, b1, rows, counters, (hipStream_t)stream);
}

// A waiter of the last exchange on `slot` gave up (error bit 2 of counters[3] / [7]): wait AGAIN for the peers' ready flags of that
// exchange and fold the counters once more -- a peer that has not raised its flag within the bound is late (a first-step table
// build, I/O between two steps, a debugger), and its push does not depend on anything this rank does.  Clears error bit 2 first; it
// is set again if this wait gives up too.  The caller decides how often to come back before it calls the peer dead.
__global__ void __launch_bounds__(64) peer_clear_timeout_kernel(int32_t* counters) {
  if (threadIdx.x == 0) { counters[3] &= ~2; counters[7] &= ~2; }
}
extern "C" int mia_comm_peer_rewait(mia_comm_t* c, int slot, int32_t* counters, void* stream) {
  if (!c || !counters) return MIA_ERR_NULL;
  if (!c->peer_ready || slot < 0 || slot >= c->peer_slots) return MIA_ERR_SIZE;
  if (c->world == 1) return MIA_OK;
  (void)hipGetLastError();
  hipStream_t cs = (hipStream_t)stream;
  const uint32_t* my = c->peer_sync[c->rank] + slot * kSyncSlotWords;
  peer_clear_timeout_kernel<<<1, 64, 0, cs>>>(counters);
  MIA_LAUNCH_CHECK();
  peer_wait_kernel<<<1, 64, 0, cs>>>(my, c->world, c->rank, c->peer_seq[slot], counters + 3, c->peer_wait_polls,
                                     reinterpret_cast<const int32_t*>(my + 2 * kMaxRanks), counters);
  MIA_LAUNCH_CHECK();
  return MIA_OK;
}

extern "C" int mia_comm_destroy(mia_comm_t* c) {
  if (!c) return MIA_OK;
  for (int r = 0; r < c->world && r < kMaxRanks; ++r) {
    if (r == c->rank || !c->peer_opened[r]) continue;
    for (int s = 0; s < c->peer_slots; ++s) if (c->peer_buf[r][s]) (void)hipIpcCloseMemHandle(c->peer_buf[r][s]);
    if (c->peer_sync[r]) (void)hipIpcCloseMemHandle(c->peer_sync[r]);
  }
  if (c->peer_owned) {
    for (int s = 0; s < c->peer_slots; ++s) if (c->peer_buf[c->rank][s]) (void)hipFree(c->peer_buf[c->rank][s]);
    if (c->peer_sync[c->rank]) (void)hipFree(c->peer_sync[c->rank]);
  }
  (void)hipGetLastError();
  for (int i = 0; i < c->n_ev; ++i) (void)hipEventDestroy(c->ev[i]);
  if (c->n_ev)
    for (int i = 0; i < kMaxChunks; ++i) (void)hipEventDestroy(c->evp[i]);
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
  delete c;
  return MIA_OK;
}

// Host-overhead helpers of the pipelined step loop (one ctypes call each instead of five torch calls): the eight counters
// of a step are copied to pinned host memory on `on_stream` once `after_stream` has passed its current point, and an event
// the library owns (created on first use, reused by the caller for the same slot) marks the copy's completion.
extern "C" int mia_letkf_step_readback(const int32_t* counters, int32_t* host8, void* after_stream, void* on_stream,
                                       void** done_event) {
  if (!counters || !host8 || !done_event) return MIA_ERR_NULL;
  (void)hipGetLastError();
  hipStream_t a = (hipStream_t)after_stream, o = (hipStream_t)on_stream;
  hipEvent_t ev = (hipEvent_t)*done_event;
  if (!ev) {
    MIA_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *done_event = (void*)ev;
  }
  if (a != o) {
    hipEvent_t pe;
    int rc = prep_event_fwd(&pe);
    if (rc != MIA_OK) return rc;
    MIA_HIP_TRY(hipEventRecord(pe, a));
    MIA_HIP_TRY(hipStreamWaitEvent(o, pe, 0));
  }
  MIA_HIP_TRY(hipMemcpyAsync(host8, counters, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, o));
  MIA_HIP_TRY(hipEventRecord(ev, o));
  return MIA_OK;
}
static int readback_after_event(const int32_t* counters, int32_t* host8, hipEvent_t after, void* on_stream, void** done_event) {
  if (!counters || !host8 || !done_event || !after) return MIA_ERR_NULL;
  (void)hipGetLastError();
  hipStream_t o = (hipStream_t)on_stream;
  hipEvent_t ev = (hipEvent_t)*done_event;
  if (!ev) {
    MIA_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    *done_event = (void*)ev;
  }
  MIA_HIP_TRY(hipStreamWaitEvent(o, after, 0));
  MIA_HIP_TRY(hipMemcpyAsync(host8, counters, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, o));
  MIA_HIP_TRY(hipEventRecord(ev, o));
  return MIA_OK;
}
extern "C" int mia_event_synchronize(void* event) {
  if (!event) return MIA_ERR_NULL;
  MIA_HIP_TRY(hipEventSynchronize((hipEvent_t)event));
  return MIA_OK;
}
// `dst` waits for everything enqueued on `src` so far (an event of the caller's, created on first use): the two runtime calls
// of torch's Stream.wait_stream without its per-call Python objects (~8 us of a ~25 us submit)
extern "C" int mia_stream_wait_stream(void* dst_stream, void* src_stream, void** event_io) {
  if (!event_io) return MIA_ERR_NULL;
  (void)hipGetLastError();
  if (!*event_io) {
    hipEvent_t e = nullptr;
    MIA_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event_io = (void*)e;
  }
  MIA_HIP_TRY(hipEventRecord((hipEvent_t)*event_io, (hipStream_t)src_stream));
  MIA_HIP_TRY(hipStreamWaitEvent((hipStream_t)dst_stream, (hipEvent_t)*event_io, 0));
  return MIA_OK;
}

extern "C" int mia_stream_wait_event(void* stream, void* event) {
  if (!event) return MIA_ERR_NULL;
  MIA_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return MIA_OK;
}
extern "C" int mia_event_destroy(void* event) {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
  return MIA_OK;
}

extern "C" int mia_letkf_sharded_step_workspace_bytes(int64_t G, int m, int k, int64_t P, int n_coord, int world,
                                                      int n_chunks, int p_max_assumed, size_t* bytes) {
  if (!bytes) return MIA_ERR_NULL;
  StepLayout L;
  int rc = step_layout(G, m, k, P, n_coord, world, n_chunks, p_max_assumed, &L);
  if (rc != MIA_OK) return rc;
  *bytes = L.total;
  return MIA_OK;
}

extern "C" int mia_letkf_sharded_step_f32(const float* X, int64_t G, int m, int k,
                                          const float* Yb, const float* d, int64_t P,
                                          const double* grid_xyz, const double* obs_xyz, int n_coord,
                                          const int32_t* coord_group, const double* gc_c, int n_r, double gc_eps,
                                          float inf_factor, float gamma, int method, int p_max_assumed,
                                          mia_comm_t* comm, int n_chunks, int phase,
                                          float* Xa, int32_t* flags, int32_t* counters,
                                          void* ws, size_t ws_bytes, void* stream, void* comm_stream) {
  return mia_letkf_sharded_step_streams_f32(X, G, m, k, Yb, d, P, grid_xyz, obs_xyz, n_coord, coord_group, gc_c, n_r,
                                            gc_eps, inf_factor, gamma, method, p_max_assumed, comm, n_chunks, phase, Xa,
                                            flags, counters, ws, ws_bytes, stream, comm_stream, nullptr, 0);
}

namespace {
// one-shot profiling hook (mia_letkf_step_timing_events)
thread_local hipEvent_t t_time_start = nullptr, t_time_stop = nullptr;

// events that order the preparation stream before the analysis stream (no communicator, hence no event storage of
// its own, on the single-rank route): a small ring, created on first use
// (one ring per device: an event belongs to the device that was current when it was created, and the launch threads serve
//  jobs of several devices)
constexpr int kPrepDevices = 16;
hipEvent_t g_prep_ev[kPrepDevices][64];
std::atomic<unsigned> g_prep_next[kPrepDevices];
std::mutex g_prep_mutex;
bool g_prep_init[kPrepDevices] = {};
int prep_event(hipEvent_t* ev) {
  int dev = 0;
  MIA_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kPrepDevices) return MIA_ERR_UNSUPPORTED;
  {
    std::lock_guard<std::mutex> lock(g_prep_mutex);
    if (!g_prep_init[dev]) {
      for (int i = 0; i < 64; ++i) MIA_HIP_TRY(hipEventCreateWithFlags(&g_prep_ev[dev][i], hipEventDisableTiming));
      g_prep_init[dev] = true;
    }
  }
  *ev = g_prep_ev[dev][g_prep_next[dev].fetch_add(1) & 63];
  return MIA_OK;
}
}  // namespace

static int prep_event_fwd(hipEvent_t* ev) { return prep_event(ev); }

extern "C" int mia_letkf_step_timing_events(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) return MIA_ERR_NULL;
  t_time_start = (hipEvent_t)start_event;
  t_time_stop = (hipEvent_t)stop_event;
  return MIA_OK;
}

// stage 0: the whole step.  The launch threads split it: stage 1 = what goes to the preparation stream (free flags of the
// direct exchange, records, index, lists) up to the event that orders the analysis behind it (*pe_io, *seq_io out);
// stage 2 = everything from that wait on (analysis, exchange), with *pe_io / *seq_io as stage 1 left them.
constexpr int kStepPrepDone = 0x100;      // internal step flag: the launch thread has waited for the preparation on the host

static inline char* base_of(void* ws) { return (char*)ws; }

// what the tile lists held by a step workspace were built for (geometry epochs, MIA_STEP_REUSE_LISTS)
struct GeomStamp {
  int route, ut, bucket, n_coord, n_r, rank, world, k;
  int64_t G, P;
  double eps, rc[MIA_MAX_RADII];
  int cg[MIA_MAX_COORD];
};
static std::mutex g_stamp_mu;
// ... and which of the index layout's two per-cell count arrays (cursor / start: the bucket index needs no starts) the workspace's
// NEXT bucket build bins into.  Every bucket step uses one and has its analysis launch put the OTHER back to zero -- the one the
// previous bucket step on this workspace left dirty -- because with the fused kernel (letkf_tile2f.hip) the wavefronts that read
// the counts and the ones that would clear them are the same launch.  cnt_use / fused: the decision taken for the step in flight.
// unknown: the entry was made for a workspace this table knows nothing about (its first step -- or one whose entry was pushed out of
// the table since): the state of its count arrays is not known either, so its next index build clears them whatever the caller
// says about the workspace (MIA_STEP_WS_CLEAN).
struct GeomEntry { void* ws; GeomStamp st; bool valid, reuse; int cnt_cur, cnt_use; bool fused; bool unknown; };
static std::deque<GeomEntry> g_stamps;        // (a handful of pipeline slots per process)
// decide = true (a step's preparation): reuse is granted when asked for AND the workspace's stamp equals `now`; otherwise the
// stamp becomes `now` (lists are rebuilt; route 0 = no tile lists).  decide = false: the decision taken for this workspace.
static bool geom_reuse_decision(void* ws, const GeomStamp& now, bool asked, bool decide) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  GeomEntry* e = nullptr;
  for (auto& x : g_stamps)
    if (x.ws == ws) { e = &x; break; }
  if (!decide) return e ? e->reuse : false;
  if (!e) {
    if (g_stamps.size() >= 64) g_stamps.pop_front();
    g_stamps.push_back(GeomEntry{ws, now, false, false, 0, 0, false, true});
    e = &g_stamps.back();
  }
  e->reuse = asked && e->valid && memcmp(&e->st, &now, sizeof now) == 0;
  if (!e->reuse) { memcpy(&e->st, &now, sizeof now); e->valid = now.route != 0; }      // (bytes, padding included: compared as bytes)
  return e->reuse;
}

// The count array of a step's bucket build and whether its analysis localises in the kernel.  decide = true (the step's
// preparation; after geom_reuse_decision, which creates the entry): `bucket_step` takes the workspace's current array and flips it
// for the next one; decide = false: what was decided.  scan_build: a scan-based index is about to be built on the workspace -- it
// needs array 0 (cursor) clean, returns false when a full clear must come first, and leaves array 0 the current one.
// *must_clear: the workspace's count arrays are in an unknown state (see GeomEntry): the build about to run clears them first.
static void count_array_decision(void* ws, bool decide, bool bucket_step, bool fused, int* use, bool* fused_out, bool* must_clear) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  GeomEntry* e = nullptr;
  for (auto& x : g_stamps)
    if (x.ws == ws) { e = &x; break; }
  *must_clear = false;
  if (!e) { *use = 0; *fused_out = false; *must_clear = true; return; }
  if (decide) {
    if (bucket_step && e->unknown) { *must_clear = true; e->unknown = false; e->cnt_cur = 0; }
    e->fused = bucket_step && fused;
    e->cnt_use = e->cnt_cur;
    if (bucket_step) e->cnt_cur ^= 1;
  }
  *use = e->cnt_use;
  *fused_out = e->fused;
}
// What a step decided where its preparation was enqueued (stage 1 / phase 0), CARRIED by the step -- the job of a step in flight, a
// local of the one-call form -- to where its analysis is enqueued: the table above is only the workspace's memory BETWEEN steps (its
// stamp, which count array comes next).  An entry pushed out of the table while a step is in flight costs that workspace's next
// step a full rebuild (`unknown`), never the step in flight its decisions (round 4 looked them up again by pointer in stage 2).
struct StepDecision { bool set = false, reuse = false, fused = false, must_clear = false; int cnt_use = 0; };

// forget what the table holds about a workspace (its memory is about to be freed or handed to another runner: the next step on that
// address starts from `unknown`)
extern "C" int mia_letkf_step_workspace_release(void* ws) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  for (auto it = g_stamps.begin(); it != g_stamps.end(); ++it)
    if (it->ws == ws) { g_stamps.erase(it); break; }
  return MIA_OK;
}

static bool count_arrays_clean_for_scan(void* ws) {
  std::lock_guard<std::mutex> lock(g_stamp_mu);
  for (auto& x : g_stamps)
    if (x.ws == ws) {
      const bool clean = x.cnt_cur == 0 && !x.unknown;
      x.cnt_cur = 0;
      x.unknown = false;      // (the caller clears everything when told "not clean")
      return clean;
    }
  return false;
}

static int step_impl(const float* X, int64_t G, int m, int k, const float* Yb, const float* d, int64_t P,
                     const double* grid_xyz, const double* obs_xyz, int n_coord, const int32_t* coord_group,
                     const double* gc_c, int n_r, double gc_eps, float inf_factor, float gamma, int method, int p_max_assumed,
                     mia_comm_t* comm, int n_chunks, int phase, float* Xa, int32_t* flags, int32_t* counters, void* ws,
                     size_t ws_bytes, void* stream, void* comm_stream, void* prep_stream, int step_flags, int stage,
                     hipEvent_t* pe_io, uint32_t* seq_io, hipEvent_t t_start, hipEvent_t t_stop, hipEvent_t* kdone_out,
                     StepDecision* dec_io) {
  const bool do1 = stage != 2, do2 = stage != 1;
  if (kdone_out) *kdone_out = nullptr;
  if (!X || !Xa || !flags || !counters || !ws || !grid_xyz || !coord_group || !gc_c) return MIA_ERR_NULL;
  if (P > 0 && (!Yb || !d || !obs_xyz)) return MIA_ERR_NULL;
  if (method < 0 || method > 2 || (phase != 0 && phase != 1)) return MIA_ERR_SIZE;
  if ((uintptr_t)ws % 256) return MIA_ERR_ALIGN;
  const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
  if (!comm) n_chunks = 1;
  // exchange route: any real multi-rank world; a one-rank communicator takes it only when chunking is asked
  // for (lets a single-GPU box drive the RCCL calls and the chunk pipeline)
  // direct exchange: Xa is one of the communicator's peer-mapped result buffers (every rank passes the same slot)
  // MIA_STEP_NO_GATHER: this rank analyses its block of the partition and keeps it -- Xa is the block, (m k, block length),
  // nothing is exchanged and no counter is reduced over the ranks (every rank validates its own step)
  const bool no_gather = comm && (step_flags & MIA_STEP_NO_GATHER) != 0;
  if (no_gather) n_chunks = 1;
  if (comm && !no_gather && world > 1 && !comm->nccl && !comm->ag) return MIA_ERR_COMM;      // (a partition-only communicator)
  const int peer_slot = (comm && world > 1 && !no_gather) ? peer_slot_of(comm, Xa) : -1;
  const bool peer = peer_slot >= 0;
  if (peer) n_chunks = 1;
  const bool exch = comm && !peer && !no_gather && (world > 1 || n_chunks > 1);
  StepLayout L;
  int rc = step_layout(G, m, k, P, n_coord, world, n_chunks, p_max_assumed, &L, (step_flags >> 4) & 7);
  if (rc != MIA_OK) return rc;
  if (ws_bytes < L.total) return MIA_ERR_WORKSPACE;
  if ((exch || peer) && !comm_stream) return MIA_ERR_NULL;
  if (peer && (size_t)m * k * G * sizeof(float) > comm->peer_bytes) return MIA_ERR_SIZE;
  hipStream_t s = (hipStream_t)stream, cs = (hipStream_t)comm_stream;
  hipStream_t ps = prep_stream ? (hipStream_t)prep_stream : s;      // records, index, lists
  // MIA_SEGMENT_SIGNAL=0: one launch + one event per piece instead of the segmented launch (fallback / A-B runs)
  const bool signal_mode = mia::option(MIA_OPT_SEGMENT_SIGNAL) != 0;
  // Lazy sort: when the block's analysis is ONE plain launch that the sixteen-points-per-wavefront kernel will take -- it
  // ranks every tile's union by observation index itself, so the order inside a neighbour list means nothing to it -- the
  // observation index is built WITHOUT its per-cell sort (one kernel and one launch gap less in the preparation chain), and
  // only a redo of declined points (phase 1: the eigensolver kernel, which sums in list order) first puts the lists of
  // exactly those points into the order a sorted index gives.  The rule is evaluated from the call's arguments, so phase 0
  // and phase 1 of a step agree; the analysis call below is checked against it.
  const int64_t blk = (int64_t)rank * L.n < G ? (((int64_t)rank * L.n + L.n < G ? (int64_t)rank * L.n + L.n : G) - (int64_t)rank * L.n) : 0;
  const bool lazy = mia::option(MIA_OPT_STEP_LAZY_SORT) != 0 && !exch && n_chunks == 1 && method != 1 && blk > 0 && P > 0 &&
                    L.cap <= 128 &&
                    mia::cheb_tile_will_serve(m, k, p_max_assumed < L.cap ? p_max_assumed : L.cap, L.cap, gamma, G, G, blk,
                                              (hipStream_t)(prep_stream ? prep_stream : stream));
  // Tile route (round 3): the localisation kernel emits tile-shaped lists (union + sqrt(rho) matrix per sixteen points), the
  // records are packed as scaled half pairs, and letkf_tile2_kernel analyses from both -- no per-point lists are written or
  // read.  Taken when the block's analysis is one plain launch of a shape the kernel covers; a tile whose union does not fit
  // its slots is counted in counters[1] and the caller repeats the step with MIA_STEP_NO_TILE_LISTS (scattered grids).
  // Declined points (phase 1) are redone from per-point lists built then, over the index this step left in its workspace.
  const int pm_tl = p_max_assumed < L.cap ? p_max_assumed : L.cap;
  const int2* tl_th = nullptr;
  const float2* tl_tc = nullptr;
  // (gamma > 0: the RBF-kernelised filter on the same tile lists -- lketkf_tile_kernel reads Yb and d themselves, no records)
  const bool tl_rbf = gamma > 0.0f;
  const bool tl_route = mia::option(MIA_OPT_TILE_LISTS) != 0 && !(step_flags & MIA_STEP_NO_TILE_LISTS) && (n_chunks == 1 || exch) &&
                        method != 1 && blk > 0 && P > 0 && L.ut <= 6 && mia::option(MIA_OPT_TILE) != 0 &&
                        (tl_rbf ? mia::lketkf_tile_covers(m, k, pm_tl, (step_flags >> 4) & 7, G, exch ? L.nc : (no_gather ? blk : G), blk, P) &&
                                      mia::cheb_primal_table((hipStream_t)(prep_stream ? prep_stream : stream), &tl_th, &tl_tc)
                                : mia::option(MIA_OPT_TILE_SPLIT) != 0 &&
                                      mia::tile2_covers(m, k, pm_tl, (step_flags >> 4) & 7, G, exch ? L.nc : (no_gather ? blk : G), blk) &&
                                      mia::tile2_records_addressable(k, P) &&
                                      mia::cheb_dual_table((hipStream_t)(prep_stream ? prep_stream : stream), &tl_th, &tl_tc));
  const bool tl_bucket = tl_route && mia::option(MIA_OPT_BUCKET_INDEX) != 0 && !(step_flags & MIA_STEP_SCAN_INDEX);
  // geometry epoch: the tile lists this workspace holds are used again (the caller vouches for unchanged coordinates, radii,
  // eps and block); only the split records are rebuilt.  Nothing to clear after the analysis: no index was built
  // ... and the library checks what it can: a stamp of what this workspace's lists were built for (route, format, block, radii,
  // eps, coordinate groups, sizes), kept on the host per workspace.  A step that asks for reuse with anything else -- an option or
  // attribute changed in between, another kernel family -- silently rebuilds instead of analysing from stale memory.
  // Fused localisation (letkf_tile2f.hip): the analysis wavefronts build their tiles' lists themselves over the bucket index -- no
  // list kernel, no lists in memory (the workspace's stamp says so: route 0).  Not when the caller declares a geometry epoch: the
  // lists are what an epoch keeps.
  const bool want_fused = tl_route && tl_bucket && !tl_rbf && !(step_flags & (MIA_STEP_REUSE_LISTS | MIA_STEP_KEEP_LISTS)) &&
                          mia::option(MIA_OPT_TILE_FUSED) != 0 && mia::tile2f_covers(m, k, L.ut, n_coord);
  GeomStamp stamp_now;
  memset(&stamp_now, 0, sizeof stamp_now);
  stamp_now.route = (tl_route && !want_fused) ? (tl_rbf ? 2 : 1) : 0;
  stamp_now.ut = L.ut; stamp_now.bucket = tl_bucket ? 1 : 0; stamp_now.n_coord = n_coord; stamp_now.n_r = n_r;
  stamp_now.G = G; stamp_now.P = P; stamp_now.rank = rank; stamp_now.world = world; stamp_now.k = k; stamp_now.eps = gc_eps;
  for (int i = 0; i < n_r && i < MIA_MAX_RADII; ++i) stamp_now.rc[i] = gc_c[i];
  for (int i = 0; i < n_coord && i < MIA_MAX_COORD; ++i) stamp_now.cg[i] = coord_group[i];
  // (decided once per step -- where its preparation is enqueued; the analysis stage and a redo of declined points read the decision)
  StepDecision dec_local;
  StepDecision& D = dec_io ? *dec_io : dec_local;
  if (phase == 0 && do1) {
    D.reuse = geom_reuse_decision(ws, stamp_now, tl_route && (step_flags & MIA_STEP_REUSE_LISTS) != 0, true);
    count_array_decision(ws, true, tl_bucket && !D.reuse, want_fused && !D.reuse, &D.cnt_use, &D.fused, &D.must_clear);
    D.set = true;
  } else if (!D.set) {      // (a redo of declined points, phase 1: a call of its own -- what the table still knows)
    D.reuse = geom_reuse_decision(ws, stamp_now, false, false);
    count_array_decision(ws, false, false, false, &D.cnt_use, &D.fused, &D.must_clear);
  }
  const bool tl_reuse = D.reuse, tl_fused = D.fused, cnt_must_clear = D.must_clear;
  const int cnt_use = D.cnt_use;
  // (the analysis launch puts the OTHER per-cell count array and the build's error word back to zero, see Tile2Params / GeomEntry)
  mia::Tile2Housekeeping tl_hk{nullptr, nullptr, nullptr, nullptr};
  int* tl_counts = nullptr;
  if (tl_bucket && !tl_reuse) {
    const mia::IndexLayout IL = mia::index_layout(base_of(ws) + L.loc, P, n_coord);
    tl_counts = cnt_use ? IL.start : IL.cursor;
    tl_hk = mia::Tile2Housekeeping{cnt_use ? IL.cursor : IL.start, &IL.hdr->ncell, &IL.hdr->err, nullptr};      // (err_out: below, once ctr is known)
  }
  // a step in flight whose analysis is ONE plain launch (stage 2 after the host-side wait): the launch carries its completion
  // (and timing) events in its own dispatch packet
  const bool carried = kdone_out && phase == 0 && !exch && !peer && n_chunks == 1 && method != 1 && (step_flags & kStepPrepDone) &&
                       (!t_start == !t_stop);
  char* base = (char*)ws;
  float* rec = (float*)(base + L.rec);
  int32_t* cnt = (int32_t*)(base + L.cnt);
  int32_t* idx = (int32_t*)(base + L.idx);
  double* w = (double*)(base + L.w);
  const int64_t b0 = (int64_t)rank * L.n < G ? (int64_t)rank * L.n : G;
  const int64_t b1 = b0 + L.n < G ? b0 + L.n : G;
  const bool eig_only = method == 1;   // auto = matfun at every m (it wins at every m measured, tools/time_rows.py)
  const int rows = m * k;
  const size_t gath_stride = mia::align_up(L.send_bytes * world, 256);
  int32_t* done = (int32_t*)(base + L.done);
  // exchange route: the redo counters live in the trailer of the last piece and travel with its all-gather
  int32_t* ctr = exch ? (int32_t*)(base + L.bufs + L.chunk_bytes * (n_chunks - 1) + (size_t)rows * L.nc * sizeof(float))
                      : counters;
  tl_hk.err_out = ctr + 3;
  mia::Tile2Loc tl_loc;
  if (tl_fused) {
    rc = mia::make_scan_params(&tl_loc.scan, grid_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, base + L.loc, MIA_TAPER_GC, true);
    if (rc != MIA_OK) return rc;
    tl_loc.scan.start = tl_counts;
    tl_loc.stats = ctr;
    tl_loc.longest_bound = pm_tl;
  }
  (void)hipGetLastError();
  if (exch || peer) {
    rc = comm_events(comm);
    if (rc != MIA_OK) return rc;
  }
  uint32_t seq = *seq_io;
  if (peer && do1) {      // "my buffer of this slot may be overwritten": told to every peer before anything else of the step
    rc = peer_begin(comm, peer_slot, ps, &seq);
    if (rc != MIA_OK) return rc;
    *seq_io = seq;
  }

  bool segmented = false, tl_block = false;
  if (phase == 0) {
    // counters[0..3] = {longest list, truncated lists, declined points, error bits} of this rank; [4..7] = max over ranks.
    // They, the trailer copy and the segment slots are cleared by the first index kernel when it runs
    // (every fill launch of its own costs ~5-8 us of the ~100 us this phase takes)
    const bool zero_in_kernel = P > 0 && b1 > b0 && !tl_reuse;
    const size_t done_ints = (size_t)n_chunks * 64 * mia::kSlotStride;
    if (do1) {
    if (!zero_in_kernel) {
      MIA_HIP_TRY(hipMemsetAsync(counters, 0, 8 * sizeof(int32_t), ps));
      if (exch) {
        MIA_HIP_TRY(hipMemsetAsync(ctr, 0, 4 * sizeof(int32_t), ps));
        MIA_HIP_TRY(hipMemsetAsync(done, 0, done_ints * sizeof(int32_t), ps));
      }
    }
    if (b1 > b0 && tl_reuse) {
      if (!tl_rbf) {
        rc = mia::split_pack_launch(Yb, d, k, P, base + L.hrec, ps);
        if (rc != MIA_OK) return rc;
      }
    } else if (b1 > b0 && tl_route) {
      const mia::ZeroJob zj{{counters, exch ? ctr : nullptr, exch ? done : nullptr},
                            {8, exch ? 4 : 0, exch ? (int64_t)done_ints : 0}};
      // bucket index: one kernel over the cell grid this workspace already holds (validated per observation); the first step on a
      // workspace, or one sent back by error bit 8, runs the bounding-box kernel first
      // (the split records ride in the bucket kernel -- small, and no LDS of its own -- rather than in the tile-list kernel, whose
      //  occupancy the packing's 10 KB of LDS per workgroup would cap)
      const mia::SplitPackJob sj{Yb, d, (unsigned char*)(base + L.hrec), k};
      if (tl_bucket)
        rc = mia::index_bucket_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, base + L.loc, L.loc_bytes, ps,
                                          zero_in_kernel ? &zj : nullptr,
                                          !(step_flags & MIA_STEP_WS_CLEAN) || (step_flags & MIA_STEP_FRESH_BOX) || cnt_must_clear,
                                          tl_rbf ? nullptr : &sj,
                                          tl_counts);
      else {
        const bool arrays_clean = count_arrays_clean_for_scan(ws);      // (a scan-based build counts in array 0)
        rc = mia::index_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, base + L.loc, L.loc_bytes, ps, nullptr,
                                   zero_in_kernel ? &zj : nullptr, (step_flags & MIA_STEP_WS_CLEAN) != 0 && arrays_clean, false);
      }
      if (rc != MIA_OK) return rc;
      if (!tl_fused) {
        rc = mia::tile_lists_launch(grid_xyz, b0, b1 - b0, P, n_coord, coord_group, gc_c, n_r, gc_eps, MIA_TAPER_GC, L.ut,
                                    base + L.tl, ctr, base + L.loc, ps, (tl_bucket || tl_rbf) ? nullptr : &sj, tl_bucket, tl_counts);
        if (rc != MIA_OK) return rc;
      }
    } else if (b1 > b0) {
      // the record packing rides inside the first index kernel too (independent work, no launch of its own)
      const mia::PackJob job{Yb, d, rec, k, (k + 1 + 3) / 4 * 4};
      const mia::ZeroJob zj{{counters, exch ? ctr : nullptr, exch ? done : nullptr},
                            {8, exch ? 4 : 0, exch ? (int64_t)done_ints : 0}};
      rc = mia::localize_impl(grid_xyz, b0, b1, obs_xyz, P, n_coord, coord_group, gc_c, n_r, gc_eps, L.cap,
                              cnt, idx, w, ctr, base + L.loc, L.loc_bytes, ps, P > 0 ? &job : nullptr, true,
                              zero_in_kernel ? &zj : nullptr, MIA_TAPER_GC,
                              (step_flags & MIA_STEP_WS_CLEAN) != 0 && count_arrays_clean_for_scan(ws), !lazy);
      if (rc != MIA_OK) return rc;
    }
    if (ps != s) {   // the analysis stream starts once the preparation stream has produced records and lists
      rc = prep_event(pe_io);
      if (rc != MIA_OK) return rc;
      MIA_HIP_TRY(hipEventRecord(*pe_io, ps));
    }
    if (exch) MIA_HIP_TRY(hipEventRecord(comm->ev[kMaxChunks], ps));
    }   // do1
    if (!do2) return MIA_OK;
    if (ps != s && !(step_flags & kStepPrepDone)) MIA_HIP_TRY(hipStreamWaitEvent(s, *pe_io, 0));
    // the side stream starts once the lists exist (and the slots it polls have been cleared)
    if (exch) MIA_HIP_TRY(hipStreamWaitEvent(cs, comm->ev[kMaxChunks], 0));
    if (t_start && !carried) MIA_HIP_TRY(hipEventRecord(t_start, s));   // (after the wait for the lists: kernel time only)
    // tile route with several pieces: ONE launch over the block, every tile writes into its piece's buffer; the pieces are
    // exchanged once it has finished (the kernel is a fraction of one piece's all-gather: nothing to overlap inside it)
    if (tl_route && n_chunks > 1 && b1 > b0) {
      rc = tl_rbf ? mia::lketkf_tile_launch(X, G, m, k, b0, b1 - b0, Yb, d, P, base + L.tl, L.ut, inf_factor, gamma,
                                            (float*)(base + L.bufs), L.nc, 0, flags, ctr + 2, mia::option(MIA_OPT_CHEB_DMAX), tl_th, tl_tc, s,
                                            (int)L.nc, (int64_t)(L.chunk_bytes / sizeof(float)), (tl_bucket && !tl_reuse) ? &tl_hk : nullptr)
                  : mia::tile2_analysis_launch(X, G, m, k, b0, b1 - b0, base + L.hrec, P, base + L.tl, L.ut, inf_factor,
                                      (float*)(base + L.bufs), L.nc, 0, flags, ctr + 2, mia::option(MIA_OPT_CHEB_DMAX), tl_th, tl_tc, s,
                                      (int)L.nc, (int64_t)(L.chunk_bytes / sizeof(float)), (tl_bucket && !tl_reuse) ? &tl_hk : nullptr,
                                      tl_fused ? &tl_loc : nullptr);
      if (rc != MIA_OK) return rc;
      tl_block = true;
    }
    // one launch over the whole block whose segments are exchanged as they complete (no kernel boundary, no
    // event between the pieces: a 1e5-point block in 4 launches costs 292 us instead of 245 us on MI355X)
    if (!tl_block && exch && n_chunks > 1 && !eig_only && b1 > b0 && signal_mode) {
      rc = mia::cheb_analysis_launch(X, G, m, k, b0, b1 - b0, rec, cnt, idx, w, L.cap, p_max_assumed < L.cap ? p_max_assumed : L.cap,
                                     inf_factor, gamma > 0.0f ? 1 : 0, gamma, (float*)(base + L.bufs), L.nc, 0, flags,
                                     ctr + 2, nullptr, nullptr, s, (int)L.nc, (int64_t)(L.chunk_bytes / sizeof(float)), done);
      if (rc == MIA_OK) segmented = true;
      else if (rc != MIA_ERR_UNSUPPORTED) return rc;
    }
  }
  if (!do2) return MIA_OK;

  for (int c = 0; c < n_chunks; ++c) {
    const int64_t c0 = b0 + c * L.nc < b1 ? b0 + c * L.nc : b1;
    const int64_t c1 = c0 + L.nc < b1 ? c0 + L.nc : b1;
    float* dst = exch ? (float*)(base + L.bufs + L.chunk_bytes * c) : Xa;
    const int64_t ldo = exch ? L.nc : (no_gather ? b1 - b0 : G);
    const int64_t o0 = exch ? 0 : (no_gather ? c0 - b0 : c0);
    if (c1 > c0 && !segmented && !tl_block) {
      const int32_t* ccnt = cnt + (c0 - b0);
      const int32_t* cidx = idx + (size_t)(c0 - b0) * L.cap;
      const double* cw = w + (size_t)(c0 - b0) * L.cap;
      int32_t* cfl = flags + (c0 - b0);
      if (phase == 1 && tl_route) {
        // declined points of the tile route: float32 records and per-point lists are built now (the step's index is still in
        // its workspace, unsorted: the flagged points' lists are put into sorted-index order as on the lazy route)
        if (c == 0) {
          rc = mia_letkf_pack_obs_f32(Yb, d, k, P, rec, stream);
          if (rc != MIA_OK) return rc;
          if (tl_bucket) {      // (the buckets are no scan-based index: build one, unsorted like the lazy route's)
            (void)count_arrays_clean_for_scan(ws);      // (cleared whole below; the scan counts in array 0)
            rc = mia::index_build_impl(obs_xyz, P, n_coord, coord_group, gc_c, n_r, base + L.loc, L.loc_bytes, (hipStream_t)stream,
                                       nullptr, nullptr, false, false);
            if (rc != MIA_OK) return rc;
          }
        }
        rc = mia::localize_lists_impl(grid_xyz, c0, c1, P, n_coord, coord_group, gc_c, n_r, gc_eps, L.cap, const_cast<int32_t*>(ccnt),
                                      const_cast<int32_t*>(cidx), const_cast<double*>(cw), (int32_t*)(base + L.scratch),
                                      base + L.loc, (hipStream_t)stream, nullptr, MIA_TAPER_GC);
        if (rc != MIA_OK) return rc;
        rc = mia::sort_flagged_lists(cfl, ccnt, const_cast<int32_t*>(cidx), const_cast<double*>(cw), c1 - c0, (int)L.cap,
                                     base + L.loc, P, n_coord, (hipStream_t)stream);
        if (rc != MIA_OK) return rc;
        rc = mia_letkf_analysis_retry_f32(X, G, m, k, c0, c1, rec, P, ccnt, cidx, cw, L.cap, p_max_assumed, inf_factor,
                                          gamma, dst, ldo, o0, cfl, stream);
        if (rc != MIA_OK) return rc;
      } else if (phase == 1) {
        if (lazy) {      // (the lists of the declined points into sorted-index order, see above)
          rc = mia::sort_flagged_lists(cfl, ccnt, const_cast<int32_t*>(cidx), const_cast<double*>(cw), c1 - c0, (int)L.cap,
                                       base + L.loc, P, n_coord, (hipStream_t)stream);
          if (rc != MIA_OK) return rc;
        }
        rc = mia_letkf_analysis_retry_f32(X, G, m, k, c0, c1, rec, P, ccnt, cidx, cw, L.cap, p_max_assumed, inf_factor,
                                          gamma, dst, ldo, o0, cfl, stream);
        if (rc != MIA_OK) return rc;
      } else {
        rc = MIA_ERR_UNSUPPORTED;
        if (!eig_only) {
          // a step in flight whose analysis is one plain launch: the launch carries its completion event itself
          hipEvent_t kstop = nullptr;
          if (carried) {          // (a timed step: the dispatch's own start / stop times, no marker packets either)
            kstop = t_stop;
            if (!kstop) {
              rc = prep_event(&kstop);
              if (rc != MIA_OK) return rc;
            }
            mia::launch_stop_event() = kstop;
            mia::launch_start_event() = t_start;
          }
          const unsigned long long tiles_before = mia::tile_launch_count();
          if (tl_route && tl_rbf)
            rc = mia::lketkf_tile_launch(X, G, m, k, c0, c1 - c0, Yb, d, P, base + L.tl, L.ut, inf_factor, gamma, dst, ldo, o0,
                                         cfl, ctr + 2, mia::option(MIA_OPT_CHEB_DMAX), tl_th, tl_tc, (hipStream_t)stream, 0, 0,
                                         (tl_bucket && !tl_reuse) ? &tl_hk : nullptr);
          else if (tl_route)
            rc = mia::tile2_analysis_launch(X, G, m, k, c0, c1 - c0, base + L.hrec, P, base + L.tl, L.ut, inf_factor, dst, ldo, o0,
                                            cfl, ctr + 2, mia::option(MIA_OPT_CHEB_DMAX), tl_th, tl_tc, (hipStream_t)stream, 0, 0,
                                            (tl_bucket && !tl_reuse) ? &tl_hk : nullptr, tl_fused ? &tl_loc : nullptr);
          else
            rc = mia_letkf_analysis_matfun_f32(X, G, m, k, c0, c1, rec, P, ccnt, cidx, cw, L.cap, p_max_assumed,
                                               inf_factor, gamma, dst, ldo, o0, cfl, ctr + 2, stream);
          // (an unsorted index is only right for the kernel the rule above predicted; the tile route has no other kernel)
          if ((lazy || tl_route) && (rc != MIA_OK || mia::tile_launch_count() == tiles_before)) {
            mia::launch_stop_event() = nullptr;
            mia::launch_start_event() = nullptr;
            return rc != MIA_OK ? rc : MIA_ERR_UNSUPPORTED;
          }
          if (kstop) {
            if (mia::launch_stop_event() == nullptr) {
              *kdone_out = kstop;      // (taken by the tile kernel's launch)
            } else if (t_start) {      // another kernel served the shape: ordinary markers around it (late start: after the fact)
              MIA_HIP_TRY(hipEventRecord(t_start, s));
              MIA_HIP_TRY(hipEventRecord(t_stop, s));
            }
            mia::launch_stop_event() = nullptr;
            mia::launch_start_event() = nullptr;
          }
        }
        if (rc == MIA_ERR_UNSUPPORTED)
          rc = mia_letkf_analysis_packed_f32(X, G, m, k, c0, c1, rec, P, ccnt, cidx, cw, L.cap, p_max_assumed,
                                             inf_factor, gamma, dst, ldo, o0, nullptr, cfl, stream);
        if (rc != MIA_OK) return rc;
      }
    }
    if (exch) {
      float* gath = (float*)(base + L.gath + gath_stride * c);
      if (segmented) {
        if (c1 > c0) {
          rc = mia::segment_wait_launch(done + (size_t)c * 64 * mia::kSlotStride, (int)(c1 - c0), ctr + 3, cs);
          if (rc != MIA_OK) return rc;
        }
      } else if (!tl_block || c == 0) {      // (the tile route's one launch: the first piece's wait covers all of them)
        MIA_HIP_TRY(hipEventRecord(comm->ev[c], s));
        MIA_HIP_TRY(hipStreamWaitEvent(cs, comm->ev[c], 0));
      }
      rc = comm_allgather(comm, dst, gath, L.send_bytes, cs);
      if (rc != MIA_OK) return rc;
      // With steps in flight (MIA_STEP_NO_JOIN) and a placement stream, the copy of the gathered piece into the result
      // leaves the exchange stream: the next step's all-gather need not wait for 2 x world x piece bytes of HBM traffic
      hipStream_t xs = cs;
      if (comm->place_stream && (step_flags & MIA_STEP_NO_JOIN)) {
        xs = comm->place_stream;
        MIA_HIP_TRY(hipEventRecord(comm->evp[c], cs));
        MIA_HIP_TRY(hipStreamWaitEvent(xs, comm->evp[c], 0));
      }
      const int64_t off = (int64_t)c * L.nc;
      int32_t* ctr_out = (phase == 0 && c == n_chunks - 1) ? counters : nullptr;
      const bool vec = (L.nc % 4 == 0) && (G % 4 == 0) && (L.n % 4 == 0) && ((uintptr_t)Xa % 16 == 0);
      if (vec) {
        const int per = kPlaceThreads * kPlaceUnroll;
        dim3 grid((unsigned)((L.nc / 4 + per - 1) / per), (unsigned)rows, (unsigned)world);
        place_chunk_kernel<4><<<grid, kPlaceThreads, 0, xs>>>(gath, Xa, G, L.n, off, (int)L.nc, rows, L.send_bytes / sizeof(float),
                                                    ctr_out, rank);
      } else {
        const int per = kPlaceThreads * kPlaceUnroll;
        dim3 grid((unsigned)((L.nc + per - 1) / per), (unsigned)rows, (unsigned)world);
        place_chunk_kernel<1><<<grid, kPlaceThreads, 0, xs>>>(gath, Xa, G, L.n, off, (int)L.nc, rows, L.send_bytes / sizeof(float),
                                                    ctr_out, rank);
      }
      MIA_LAUNCH_CHECK();
    }
  }

  if (peer) {      // block analysed (stream s) -> exchange stream: wait, push, signal, wait (see "Direct exchange")
    MIA_HIP_TRY(hipEventRecord(comm->ev[0], s));
    MIA_HIP_TRY(hipStreamWaitEvent(cs, comm->ev[0], 0));
    rc = peer_finish(comm, peer_slot, seq, G, b0, b1, rows, counters, cs);
    if (rc != MIA_OK) return rc;
  }
  if (phase == 0 && t_stop && !carried) MIA_HIP_TRY(hipEventRecord(t_stop, s));
  // (without the exchange route counters[4..7] stay zero: the rank's own [0..3] are the whole story)
  if ((exch || peer) && !(step_flags & MIA_STEP_NO_JOIN)) {   // the caller's stream continues after the exchange
    MIA_HIP_TRY(hipEventRecord(comm->ev[kMaxChunks + 1], cs));
    MIA_HIP_TRY(hipStreamWaitEvent(s, comm->ev[kMaxChunks + 1], 0));
  }
  return MIA_OK;
}

extern "C" int mia_letkf_sharded_step_streams_f32(const float* X, int64_t G, int m, int k,
                                                  const float* Yb, const float* d, int64_t P,
                                                  const double* grid_xyz, const double* obs_xyz, int n_coord,
                                                  const int32_t* coord_group, const double* gc_c, int n_r, double gc_eps,
                                                  float inf_factor, float gamma, int method, int p_max_assumed,
                                                  mia_comm_t* comm, int n_chunks, int phase,
                                                  float* Xa, int32_t* flags, int32_t* counters,
                                                  void* ws, size_t ws_bytes, void* stream, void* comm_stream,
                                                  void* prep_stream, int step_flags) {
  hipEvent_t pe = nullptr;
  uint32_t seq = 0;
  const hipEvent_t t0 = t_time_start, t1 = t_time_stop;
  t_time_start = t_time_stop = nullptr;
  return step_impl(X, G, m, k, Yb, d, P, grid_xyz, obs_xyz, n_coord, coord_group, gc_c, n_r, gc_eps, inf_factor, gamma, method,
                   p_max_assumed, comm, n_chunks, phase, Xa, flags, counters, ws, ws_bytes, stream, comm_stream, prep_stream,
                   step_flags, 0, &pe, &seq, t0, t1, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------------
// Launch threads.  One step is ~9 kernel launches, a copy and half a dozen event operations: ~100 us of HIP runtime calls
// on the calling thread -- more than a step's GPU time since the sixteen-point kernel (host-bound pipeline: 0.104 ms per
// step with the GPU ~75 % busy, whichever thread made the calls).  mia_letkf_step_submit hands the step to TWO worker threads
// of the library (no throughput gain measured at N = 1, where the analysis stream is the bound; the caller's submit() drops
// from 80 to 24 us): thread A enqueues what goes to the preparation stream (stage 1 of step_impl), thread B what follows
// (analysis, exchange, read-back), each in submission order -- every stream sees its work in step order, every rank enqueues
// its exchanges in the same order -- so that the host time per step is the larger half, not the sum, and overlaps the
// caller's own per-step work.  mia_letkf_step_join waits until the job's launches are enqueued (not until the GPU has run
// them: that is what the read-back event is for).
namespace {
struct StepJob {
  const float* X; int64_t G; int m, k; const float* Yb; const float* d; int64_t P; const double* grid; const double* obs;
  int n_coord; int32_t cg[MIA_MAX_COORD]; double rc_[MIA_MAX_RADII]; int n_r; double eps; float inf, gamma; int method, hint;
  mia_comm_t* comm; int n_chunks, phase; float* Xa; int32_t* flags; int32_t* counters; void* ws; size_t ws_bytes;
  void *stream, *comm_stream, *prep_stream; int step_flags;
  int32_t* host8; void *after, *on; void** done_event; void *t0, *t1;
  hipEvent_t pe = nullptr; uint32_t seq = 0;
  hipEvent_t kdone = nullptr;      // completion event carried by the analysis launch itself (stage 2), if any
  StepDecision dec;                // what stage 1 decided about the workspace's lists and count arrays, for stage 2
  int opts[MIA_OPT_COUNT_];        // the route options as they stood when the caller submitted the step
  int batch_n = 1;                 // steps in the analysis launch this step was part of (launch coalescing)
  long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // host time stamps (ns): submitted, A begins, A done, B has it, its preparation seen done,
                                                   // analysis enqueued, read-back enqueued (mia_debug_step_trace)
  int device = 0;
  int rc = 0;
  bool done = false;
  int run(int stage) {
    struct Scope { Scope(const int* o) { mia::option_override(o); } ~Scope() { mia::option_override(nullptr); } } scope(opts);
    return step_impl(X, G, m, k, Yb, d, P, grid, obs, n_coord, cg, rc_, n_r, eps, inf, gamma, method, hint, comm, n_chunks, phase,
                     Xa, flags, counters, ws, ws_bytes, stream, comm_stream, prep_stream, step_flags, stage, &pe, &seq,
                     (hipEvent_t)t0, (hipEvent_t)t1, stage == 2 ? &kdone : nullptr, &dec);
  }
};
struct LaunchThreads {
  std::thread ta, tb;
  std::mutex mu;
  std::condition_variable cv_a, cv_b, cv_done;
  std::deque<StepJob*> qa, qb;
  int device = 0;
  bool stop = false, started = false;
  int busy = 0;          // jobs handed in and not yet finished by thread B
  std::atomic<long long> ns_a{0}, ns_b{0}, n_jobs{0};     // host time spent enqueueing (mia_letkf_step_launch_stats)
  // entries of the two queues, readable without the lock: a thread whose queue has run dry polls its counter for kSpinUs before
  // it sleeps on the condition variable.  Waking a sleeping thread costs 30-60 us on this host (a step's whole GPU time): the
  // first analysis launch of a burst of steps came 80 us after the first preparation kernel had finished
  // (profiles/r05_timeline_steps20.txt); a caller that submits its next step within kSpinUs finds both threads awake
  std::atomic<int> na{0}, nb_q{0};
  static constexpr int kTraceN = 256;
  std::array<long long, 8> trace[kTraceN];      // the stamps of the last kTraceN steps (mia_debug_step_trace)
  unsigned long long trace_n = 0;
  static long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  static constexpr long long kSpinUs = 400;
  void spin_for(const std::atomic<int>& n) {
    if (n.load(std::memory_order_acquire) > 0) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int i = 0; i < 64; ++i) {
        if (n.load(std::memory_order_acquire) > 0) return;
        __builtin_ia32_pause();
      }
      if (stop_flag.load(std::memory_order_relaxed)) return;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinUs) return;
    }
  }
  std::atomic<bool> stop_flag{false};
  static void relax(int n = 40) {
    for (int i = 0; i < n; ++i) __builtin_ia32_pause();      // 40: ~1 us
  }
  std::atomic<long long> n_launches{0}, n_launch_steps{0};      // analysis launches made with the collector on, and the steps in them
  static int i_rc_first(int a, int b) { return a != MIA_OK ? a : b; }
  void run_a() {
    int cur_a = device;
    (void)hipSetDevice(device);
    for (;;) {
      StepJob* j;
      spin_for(na);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_a.wait(lk, [&] { return stop || !qa.empty(); });
        if (qa.empty()) return;
        j = qa.front();
        qa.pop_front();
        --na;
      }
      if (j->device != cur_a) { (void)hipSetDevice(j->device); cur_a = j->device; }
      const auto ta0 = std::chrono::steady_clock::now();
      j->ts[1] = now_ns();
      const int rc = j->run(1);
      j->ts[2] = now_ns();
      ns_a += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - ta0).count();
      ++n_jobs;
      {
        std::lock_guard<std::mutex> lk(mu);
        j->rc = rc;
        qb.push_back(j);
        ++nb_q;
      }
      cv_b.notify_one();
    }
  }
  void run_b() {
    int cur_b = device;
    (void)hipSetDevice(device);
    std::deque<hipEvent_t> running;      // completion events of this thread's coalesced launches that may still be running
    void* last_stream = nullptr;
    hipEvent_t own_ev[8] = {};
    bool own_init = false;
    unsigned own_next = 0;
    struct OwnEvents {      // (destroyed when the thread ends)
      hipEvent_t (&ev)[8]; bool& init;
      ~OwnEvents() { if (init) for (auto e : ev) if (e) (void)hipEventDestroy(e); }
    } own_guard{own_ev, own_init};
    for (;;) {
      StepJob* j;
      spin_for(nb_q);
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_b.wait(lk, [&] { return stop || !qb.empty(); });
        if (qb.empty()) return;
        j = qb.front();
        qb.pop_front();
        --nb_q;
      }
      if (j->device != cur_b) { (void)hipSetDevice(j->device); cur_b = j->device; running.clear(); if (own_init) { for (auto& e : own_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } } own_init = false; }
      j->ts[3] = now_ns();
      int rc = j->rc;
      // Steps in flight: wait for the step's preparation HERE, on the host, and enqueue the analysis kernel with nothing in
      // front of it.  A stream-wait in the analysis queue is a barrier packet between every two analysis kernels (11-16 us
      // from the end of one to the start of the next, 6-7 without): 0.093 -> 0.088 ms per step once the chip has room for
      // the preparation beside the analysis kernel (it made no difference while the three-wave kernel filled it).  The
      // preparation runs two steps ahead, so the wait is short; query + yield rather than a spinning synchronise.
      if (rc == MIA_OK && j->pe && (j->step_flags & MIA_STEP_NO_JOIN) && mia::option(MIA_OPT_STEP_HOSTWAIT) != 0) {
        hipError_t q;
        while ((q = hipEventQuery(j->pe)) == hipErrorNotReady) relax();      // (a microsecond between two queries: the runtime's locks are
                                                                             //  the caller's and the other launch thread's too)
        if (q == hipSuccess) j->step_flags |= kStepPrepDone;
        else (void)hipGetLastError();         // (leave the ordering to the stream wait)
      }
      auto tb0 = std::chrono::steady_clock::now();
      j->ts[4] = now_ns();
      // Launch coalescing: a launch of one step's 6250 tiles spends a quarter of its time filling and draining the chip, and kernels
      // of different streams overlap badly (letkf_tile2f.hip).  Steps in flight are independent: while this step's analysis is being
      // put together, the FOLLOWING steps of the queue whose preparation has already finished are put together too, and their tiles
      // go to the GPU as one grid (up to kT2fBatchMax steps, the fused kernel only; a timed step always opens a launch).  Every
      // step keeps its own workspace, counters, flags, result, read-back and completion event.
      // A launch thread that enqueues every step the moment it is ready never finds a second one waiting: the queue in front of the
      // GPU would be the hardware's, where it cannot be merged.  So with the option on the thread holds a step back while
      // `step_coalesce` launches of its own are still running, and the steps that become ready meanwhile join it.
      // MEASURED (profiles/r05_coalesce.txt) and therefore OFF by default: the merged launch is cheaper per step as predicted (90 us
      // for 2.9 steps = 31 us per step against 47-55 us for a launch of one step beside its neighbours), but holding steps back
      // lengthens the loop step -> result -> next submission of a pipeline with eight slots by more than the launch saves:
      // 0.052 ms per step with one launch running at a time, 0.048-0.050 with two, against 0.047 without.
      StepJob* batch[mia::kT2fBatchMax] = {j, nullptr, nullptr, nullptr};
      int nb = 1;
      const bool may = rc == MIA_OK && (j->step_flags & kStepPrepDone) && j->phase == 0 && !j->comm && j->n_chunks == 1 && j->method != 1 &&
                       j->opts[MIA_OPT_STEP_COALESCE] != 0 && j->host8;
      if (may) {
        tb0 = std::chrono::steady_clock::now();
        mia::tile2f_collect_begin();
      }
      if (rc == MIA_OK) rc = j->run(2);
      if (may) {
        const size_t w = (size_t)j->opts[MIA_OPT_STEP_COALESCE];
        bool open = mia::tile2f_collecting() && mia::tile2f_collected() == 1;
        long long waited = 0;
        // the step is put together; it goes to the GPU when fewer than w launches of this thread are still running, and the steps
        // that become ready until then join it
        for (;;) {
          // (the queue's atomic count first: a spinning thread that takes the queue's mutex on every turn starves the caller's
          //  submit() of it -- submissions of 60-140 us were measured, profiles/r05_step_trace_coalesce.txt)
          while (open && nb < mia::kT2fBatchMax && nb_q.load(std::memory_order_acquire) > 0) {
            StepJob* c = nullptr;
            {
              std::lock_guard<std::mutex> lk(mu);
              if (!qb.empty()) c = qb.front();
              const bool ok = c && c->rc == MIA_OK && c->device == j->device && c->pe && (c->step_flags & MIA_STEP_NO_JOIN) && c->phase == 0 &&
                              !c->comm && c->n_chunks == 1 && c->method != 1 && c->host8 && !c->t0 && !c->t1 &&
                              c->opts[MIA_OPT_STEP_COALESCE] != 0 && c->opts[MIA_OPT_STEP_HOSTWAIT] != 0;
              if (ok && hipEventQuery(c->pe) == hipSuccess) { qb.pop_front(); --nb_q; }      // (its preparation has finished: no waiting for it)
              else { (void)hipGetLastError(); c = nullptr; }
            }
            if (!c) break;
            c->step_flags |= kStepPrepDone;
            const int had = mia::tile2f_collected();
            c->rc = c->run(2);
            batch[nb++] = c;
            if (!mia::tile2f_collecting() || mia::tile2f_collected() == had) open = false;      // (went another way: launched by itself)
          }
          while (!running.empty() && hipEventQuery(running.front()) != hipErrorNotReady) { (void)hipGetLastError(); running.pop_front(); }
          if (running.size() < w) break;
          const auto tw = std::chrono::steady_clock::now();
          relax(240);
          waited += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw).count();
        }
        ns_b -= waited;      // (the host-time statistic: enqueueing, not waiting)
      }
      if (may) {
        const int n_in = mia::tile2f_collected();
        // on a stream other than the previous launch's where the steps offer one: the head of this launch overlaps the tail of that
        // (one stream for all these launches measured the same: profiles/r05_coalesce.txt)
        void* ls = j->stream;
        for (int i = 0; i < n_in; ++i)
          if (batch[i]->stream != last_stream) { ls = batch[i]->stream; break; }
        int lrc = mia::tile2f_collect_launch((hipStream_t)ls);
        if (lrc == MIA_OK && n_in > 0) {
          last_stream = ls;
          hipEvent_t done = n_in > 1 ? batch[n_in - 1]->kdone : (j->t1 ? nullptr : j->kdone);
          if (!done) {        // (a timed step alone: its stop event is the caller's, which may not outlive the step)
            if (!own_init) {
              for (auto& e : own_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
              own_init = true;
            }
            done = own_ev[own_next++ & 7];
            if (done && hipEventRecord(done, (hipStream_t)ls) != hipSuccess) { (void)hipGetLastError(); done = nullptr; }
          }
          if (done) running.push_back(done);
        }
        if (lrc != MIA_OK)
          for (int i = 0; i < nb; ++i) if (i < n_in && batch[i]->rc == MIA_OK) batch[i]->rc = lrc;
        rc = i_rc_first(rc, batch[0]->rc);
        for (int i = 0; i < nb; ++i) batch[i]->batch_n = i < n_in ? n_in : 1;
        n_launches += 1;
        n_launch_steps += n_in > 0 ? n_in : 1;
      }
      j->ts[5] = now_ns();
      for (int i = 0; i < nb; ++i) {
        StepJob* b = batch[i];
        int brc = i == 0 ? rc : b->rc;
        if (brc == MIA_OK && b->host8) {
          // (the read-back waits for the kernel's own completion event when the launch carried one: no marker on the stream)
          if (b->kdone && b->after == b->stream) brc = readback_after_event(b->counters, b->host8, b->kdone, b->on, b->done_event);
          else brc = mia_letkf_step_readback(b->counters, b->host8, b->after, b->on, b->done_event);
        }
        b->rc = brc;
      }
      ns_b += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tb0).count();
      j->ts[6] = now_ns();
      {
        std::lock_guard<std::mutex> lk(mu);
        for (int i = 0; i < nb; ++i) {
          trace[trace_n++ % kTraceN] = *reinterpret_cast<std::array<long long, 8>*>(batch[i]->ts);
          batch[i]->done = true; --busy;
        }
      }
      cv_done.notify_all();
    }
  }
  ~LaunchThreads() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; stop_flag = true; }
    cv_a.notify_all();
    cv_b.notify_all();
    if (ta.joinable()) ta.join();
    if (tb.joinable()) tb.join();
  }
};
LaunchThreads g_launcher;
}  // namespace

static thread_local long long t_submit_entry = 0;      // (diagnostics: when the argument-block submission was entered, mia_debug_step_trace)
extern "C" int mia_letkf_step_submit(const float* X, int64_t G, int m, int k, const float* Yb, const float* d, int64_t P,
                                     const double* grid_xyz, const double* obs_xyz, int n_coord, const int32_t* coord_group,
                                     const double* gc_c, int n_r, double gc_eps, float inf_factor, float gamma, int method,
                                     int p_max_assumed, mia_comm_t* comm, int n_chunks, int phase, float* Xa, int32_t* flags,
                                     int32_t* counters, void* ws, size_t ws_bytes, void* stream, void* comm_stream,
                                     void* prep_stream, int step_flags, int32_t* host8, void* after_stream, void* on_stream,
                                     void** done_event, void* time_start_event, void* time_stop_event, void** job_out) {
  if (!job_out || !coord_group || !gc_c) return MIA_ERR_NULL;
  if (n_coord < 1 || n_coord > MIA_MAX_COORD || n_r < 1 || n_r > MIA_MAX_RADII) return MIA_ERR_SIZE;
  StepJob* j = new StepJob();
  j->X = X; j->G = G; j->m = m; j->k = k; j->Yb = Yb; j->d = d; j->P = P; j->grid = grid_xyz; j->obs = obs_xyz;
  j->n_coord = n_coord; j->n_r = n_r; j->eps = gc_eps; j->inf = inf_factor; j->gamma = gamma; j->method = method;
  for (int c = 0; c < n_coord; ++c) j->cg[c] = coord_group[c];
  for (int r = 0; r < n_r; ++r) j->rc_[r] = gc_c[r];
  j->hint = p_max_assumed; j->comm = comm; j->n_chunks = n_chunks; j->phase = phase; j->Xa = Xa; j->flags = flags;
  j->counters = counters; j->ws = ws; j->ws_bytes = ws_bytes; j->stream = stream; j->comm_stream = comm_stream;
  j->prep_stream = prep_stream; j->step_flags = step_flags; j->host8 = host8; j->after = after_stream; j->on = on_stream;
  j->done_event = done_event; j->t0 = time_start_event; j->t1 = time_stop_event;
  mia::option_snapshot(j->opts);
  j->ts[0] = LaunchThreads::now_ns();
  j->ts[7] = t_submit_entry;            // (diagnostics: entry of the argument-block submission; 0 through the plain entry)
  t_submit_entry = 0;
  if (hipGetDevice(&j->device) != hipSuccess) { (void)hipGetLastError(); delete j; return MIA_ERR_UNSUPPORTED; }
  {
    std::lock_guard<std::mutex> lk(g_launcher.mu);
    if (!g_launcher.started) {
      g_launcher.device = j->device;
      g_launcher.started = true;
      g_launcher.ta = std::thread([] { g_launcher.run_a(); });
      g_launcher.tb = std::thread([] { g_launcher.run_b(); });
    }
    g_launcher.qa.push_back(j);
    ++g_launcher.na;
    ++g_launcher.busy;
  }
  g_launcher.cv_a.notify_one();
  *job_out = j;
  return MIA_OK;
}

extern "C" int mia_letkf_step_submit_args(const mia_step_args_t* a, void** job_out) {
  if (!a) return MIA_ERR_NULL;
  t_submit_entry = LaunchThreads::now_ns();
  if (a->in_event) {
    // (an idle caller stream has produced everything it ever will before this call: nothing to wait for -- no event, and no
    //  barrier packet in front of the preparation kernel)
    const hipError_t q = hipStreamQuery((hipStream_t)a->caller_stream);
    if (q != hipSuccess) {
      (void)hipGetLastError();
      const int rc = mia_stream_wait_stream(a->prep_stream, a->caller_stream, a->in_event);
      if (rc != MIA_OK) return rc;
    }
  }
  return mia_letkf_step_submit(a->X, a->G, a->m, a->k, a->Yb, a->d, a->P, a->grid_xyz, a->obs_xyz, a->n_coord, a->coord_group, a->gc_c,
                               a->n_r, a->gc_eps, a->inf_factor, a->gamma, a->method, a->p_max_assumed, a->comm, a->n_chunks, a->phase,
                               a->Xa, a->flags, a->counters, a->ws, a->ws_bytes, a->stream, a->comm_stream, a->prep_stream, a->step_flags,
                               a->host8, a->after_stream, a->on_stream, a->done_event, a->time_start_event, a->time_stop_event, job_out);
}

// One step taken at once on the caller's thread through the argument block: what mia_letkf_step_drain + the step call +
// mia_letkf_step_readback (+ mia_event_synchronize and a copy of the counters, when out8 is given) do, in one call.
extern "C" int mia_letkf_step_run_args(const mia_step_args_t* a, int32_t* out8) {
  if (!a || !a->done_event || !a->host8) return MIA_ERR_NULL;
  int rc = mia_letkf_step_drain();       // (a synchronous step must not overtake queued ones)
  if (rc != MIA_OK) return rc;
  if (a->in_event) {
    rc = mia_stream_wait_stream(a->prep_stream ? a->prep_stream : a->stream, a->caller_stream, a->in_event);
    if (rc != MIA_OK) return rc;
  }
  if ((a->time_start_event == nullptr) != (a->time_stop_event == nullptr)) return MIA_ERR_NULL;
  hipEvent_t pe = nullptr;
  uint32_t seq = 0;
  rc = step_impl(a->X, a->G, a->m, a->k, a->Yb, a->d, a->P, a->grid_xyz, a->obs_xyz, a->n_coord, a->coord_group, a->gc_c, a->n_r, a->gc_eps,
                 a->inf_factor, a->gamma, a->method, a->p_max_assumed, a->comm, a->n_chunks, a->phase, a->Xa, a->flags, a->counters, a->ws,
                 a->ws_bytes, a->stream, a->comm_stream, a->prep_stream, a->step_flags & ~MIA_STEP_NO_JOIN, 0, &pe, &seq,
                 (hipEvent_t)a->time_start_event, (hipEvent_t)a->time_stop_event, nullptr, nullptr);
  if (rc != MIA_OK) return rc;
  rc = mia_letkf_step_readback(a->counters, a->host8, a->after_stream, a->on_stream, a->done_event);
  if (rc != MIA_OK || !out8) return rc;
  MIA_HIP_TRY(hipEventSynchronize((hipEvent_t)*a->done_event));
  for (int i = 0; i < 8; ++i) out8[i] = a->host8[i];
  return MIA_OK;
}

extern "C" int mia_letkf_step_collect(void* job, void** done_event, const int32_t* host8, void* consumer_stream, int consumer_stream_valid,
                                      int32_t* out8, int* batch_n) {
  if (!job || !done_event || !host8 || !out8) return MIA_ERR_NULL;
  const int rc = mia_letkf_step_join_info(job, batch_n);
  if (rc != MIA_OK) return rc;
  const hipEvent_t ev = (hipEvent_t)*done_event;      // (made by the step's read-back on the launch thread: read after the join)
  if (!ev) return MIA_ERR_NULL;
  MIA_HIP_TRY(hipEventSynchronize(ev));
  for (int i = 0; i < 8; ++i) out8[i] = host8[i];
  if (consumer_stream_valid) MIA_HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, ev, 0));
  return MIA_OK;
}

namespace {
std::mutex g_tev_mutex;
std::vector<hipEvent_t> g_tev_free;
}
extern "C" int mia_timing_event_acquire(void** event) {
  if (!event) return MIA_ERR_NULL;
  {
    std::lock_guard<std::mutex> lk(g_tev_mutex);
    if (!g_tev_free.empty()) { *event = (void*)g_tev_free.back(); g_tev_free.pop_back(); return MIA_OK; }
  }
  hipEvent_t e = nullptr;
  MIA_HIP_TRY(hipEventCreate(&e));
  *event = (void*)e;
  return MIA_OK;
}
extern "C" int mia_timing_event_release(void* event) {
  if (!event) return MIA_ERR_NULL;
  std::lock_guard<std::mutex> lk(g_tev_mutex);
  g_tev_free.push_back((hipEvent_t)event);
  return MIA_OK;
}
extern "C" int mia_timing_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
  if (!start_event || !stop_event || !ms) return MIA_ERR_NULL;
  MIA_HIP_TRY(hipEventSynchronize((hipEvent_t)stop_event));
  MIA_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event));
  return MIA_OK;
}

// the host time stamps of the last steps handed to the launch threads, oldest first: 8 values per step (ns of the steady clock:
// submitted, thread A begins / has enqueued the preparation, thread B takes the step / has seen its preparation finished / has
// enqueued the analysis / the read-back, 0).  Returns the number of steps written (tools/step_trace.py)
extern "C" int mia_debug_step_trace(long long* out, int max_steps) {
  if (!out || max_steps <= 0) return 0;
  std::lock_guard<std::mutex> lk(g_launcher.mu);
  const unsigned long long n = g_launcher.trace_n;
  const int have = (int)(n < (unsigned long long)LaunchThreads::kTraceN ? n : LaunchThreads::kTraceN);
  const int take = have < max_steps ? have : max_steps;
  for (int i = 0; i < take; ++i) {
    const auto& t = g_launcher.trace[(n - take + i) % LaunchThreads::kTraceN];
    for (int q = 0; q < 8; ++q) out[8 * i + q] = t[q];
  }
  return take;
}

// host time the two launch threads have spent enqueueing so far (microseconds: preparation stage, analysis / exchange /
// read-back stage) and the number of steps: tells a pipeline that waits for its launches from one that waits for the GPU
extern "C" int mia_letkf_step_launch_stats(double* prep_us, double* rest_us, long long* steps) {
  if (!prep_us || !rest_us || !steps) return MIA_ERR_NULL;
  *prep_us = g_launcher.ns_a.load() * 1e-3;
  *rest_us = g_launcher.ns_b.load() * 1e-3;
  *steps = g_launcher.n_jobs.load();
  return MIA_OK;
}

// waits until the job's launches are enqueued; returns the step call's status and frees the job
extern "C" int mia_letkf_step_join(void* job) {
  if (!job) return MIA_ERR_NULL;
  StepJob* j = (StepJob*)job;
  std::unique_lock<std::mutex> lk(g_launcher.mu);
  g_launcher.cv_done.wait(lk, [&] { return j->done; });
  const int rc = j->rc;
  lk.unlock();
  delete j;
  return rc;
}

// ... and how many steps shared this step's analysis launch (1: its own launch)
extern "C" int mia_letkf_step_join_info(void* job, int* batch_n) {
  if (!job) return MIA_ERR_NULL;
  StepJob* j = (StepJob*)job;
  std::unique_lock<std::mutex> lk(g_launcher.mu);
  g_launcher.cv_done.wait(lk, [&] { return j->done; });
  const int rc = j->rc;
  if (batch_n) *batch_n = j->batch_n;
  lk.unlock();
  delete j;
  return rc;
}
// analysis launches of steps in flight so far and the steps they carried (launch coalescing: steps / launches > 1)
extern "C" int mia_letkf_step_coalesce_stats(long long* launches, long long* steps) {
  if (!launches || !steps) return MIA_ERR_NULL;
  *launches = g_launcher.n_launches.load();
  *steps = g_launcher.n_launch_steps.load();
  return MIA_OK;
}

// waits until nothing is queued or running on the launch thread (before a synchronous call that must not overtake it)
extern "C" int mia_letkf_step_drain(void) {
  std::unique_lock<std::mutex> lk(g_launcher.mu);
  g_launcher.cv_done.wait(lk, [&] { return g_launcher.busy == 0; });
  return MIA_OK;
}
