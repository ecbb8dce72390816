// dp_rccl.hip — the data-parallel exchange behind the C ABI (SURVEY.md §8b `dp_allreduce_{begin,wait}`, §8e).
//
// One process per GPU; the only exchange of the training step is the average of a net's flat fp32 gradient bucket across the
// replicas.  This file owns the RCCL communicator, a side HIP stream and the events that order it against the caller's streams:
//   pcg_dp_allreduce        average in stream order on the caller's stream (D's bucket: Adam(D) needs it immediately)
//   pcg_dp_allreduce_begin  average on the library's side stream, ordered after everything already queued on the producer stream
//   pcg_dp_side_stream      ... the caller may queue more work behind it there (Adam(G)), then pcg_dp_record(slot)
//   pcg_dp_allreduce_wait   make a consumer stream wait (on the GPU) for slot's reduction (+ recorded follow-up work)
//   pcg_dp_allreduce_sum_f64 / pcg_dp_broadcast   small in-stream collectives: exact-BatchNorm statistic sums, initial weights
// RCCL is bound at run time (dlopen of librccl.so.1: the copy PyTorch-ROCm has already mapped when this library lives in a
// PyTorch process, the ROCm one otherwise), so libpcgan_hip.so has no link-time dependency on it and single-GPU use never loads it.
// xGMI is point-to-point; bucket sizes (DCGAN: 11 MB + 14 MB per step, one call each) are chosen by the host layer.
#include "pcg_common.h"
#include <dlfcn.h>
#include <string.h>

namespace pcg {
namespace {

// the handful of RCCL declarations used (rccl.h: ncclResult_t = int, ncclSuccess = 0, ncclFloat = 7, ncclDouble = 8, ncclSum = 0, ncclAvg = 4)
typedef struct ncclComm* ncclComm_t;
struct UniqueId { char internal[PCG_DP_UNIQUE_ID_BYTES]; };
constexpr int kFloat = 7, kDouble = 8, kSum = 0, kAvg = 4;
using fn_get_id = int (*)(UniqueId*);
using fn_init = int (*)(ncclComm_t*, int, UniqueId, int);
using fn_destroy = int (*)(ncclComm_t);
using fn_allreduce = int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
using fn_bcast = int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
using fn_errstr = const char* (*)(int);
using fn_version = int (*)(int*);

constexpr int kSlots = 8;
struct Dp {
  void* lib = nullptr;
  fn_get_id get_id = nullptr; fn_init init = nullptr; fn_destroy destroy = nullptr; fn_allreduce allreduce = nullptr;
  fn_bcast bcast = nullptr; fn_errstr errstr = nullptr; fn_version version = nullptr;
  float* token = nullptr;        // 64 bytes of device memory: the operand of pcg_dp_barrier's 4-byte all-reduce
  ncclComm_t comm = nullptr;
  int rank = -1, world = 0, device = -1;
  hipStream_t side = nullptr;
  hipEvent_t ready[kSlots] = {}, done[kSlots] = {};
  bool pending[kSlots] = {};
  bool sync_bn = false;
} g;

int bind() {
  if (g.lib) return PCG_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);       // already mapped (PyTorch-ROCm): share it
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { set_error("pcg_dp: cannot load RCCL (librccl.so.1): %s", dlerror()); return PCG_ERR_INVALID; }
  g.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  g.init = (fn_init)dlsym(h, "ncclCommInitRank");
  g.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  g.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  g.bcast = (fn_bcast)dlsym(h, "ncclBroadcast");
  g.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  g.version = (fn_version)dlsym(h, "ncclGetVersion");      // optional
  if (!g.get_id || !g.init || !g.destroy || !g.allreduce || !g.bcast || !g.errstr) {
    set_error("pcg_dp: librccl lacks an expected symbol");
    return PCG_ERR_INVALID;
  }
  g.lib = h;
  return PCG_OK;
}

int nccl_ok(int rc, const char* what) {
  if (rc == 0) return PCG_OK;
  set_error("%s: RCCL error %d: %s", what, rc, g.errstr ? g.errstr(rc) : "?");
  return PCG_ERR_LAUNCH;
}
int hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return PCG_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return PCG_ERR_LAUNCH;
}
}  // namespace

// internal hooks for batchnorm.hip (exact global-batch BatchNorm)
bool dp_sync_bn() { return g.comm != nullptr && g.sync_bn; }
int dp_world() { return g.comm ? g.world : 1; }
int dp_allreduce_f64(double* buf, int64_t n, hipStream_t s) {
  return nccl_ok(g.allreduce(buf, buf, (size_t)n, kDouble, kSum, g.comm, s), "ncclAllReduce(BatchNorm sums)");
}

namespace {
#define DP_READY(who) PCG_REQUIRE(g.comm != nullptr, "%s: pcg_dp_init has not been called", who)

// pcg_dp_init failed after ncclCommInitRank: destroy what exists (null handles are skipped), leave the state "not initialised".
// The error text of the failure that brought us here is kept (the destroy calls do not touch it).
void release_partial() {
  for (int i = 0; i < kSlots; ++i) {
    if (g.ready[i]) (void)hipEventDestroy(g.ready[i]);
    if (g.done[i]) (void)hipEventDestroy(g.done[i]);
    g.ready[i] = nullptr; g.done[i] = nullptr; g.pending[i] = false;
  }
  if (g.side) (void)hipStreamDestroy(g.side);
  g.side = nullptr;
  if (g.token) (void)hipFree(g.token);
  g.token = nullptr;
  if (g.comm) (void)g.destroy(g.comm);
  g.comm = nullptr;
  g.rank = -1; g.world = 0;
}

}  // namespace
}  // namespace pcg

using namespace pcg;

extern "C" int pcg_dp_unique_id(void* id_out) {
  PCG_REQUIRE(id_out != nullptr, "pcg_dp_unique_id: null output");
  if (int e = bind()) return e;
  UniqueId id;
  if (int e = nccl_ok(g.get_id(&id), "ncclGetUniqueId")) return e;
  memcpy(id_out, &id, sizeof(id));
  return PCG_OK;
}

extern "C" int pcg_dp_init(const void* id_in, int32_t rank, int32_t world) {
  PCG_REQUIRE(id_in && world >= 1 && rank >= 0 && rank < world, "pcg_dp_init: bad arguments (rank %d of %d)", rank, world);
  PCG_REQUIRE(g.comm == nullptr, "pcg_dp_init: already initialised (rank %d of %d); call pcg_dp_shutdown first", g.rank, g.world);
  if (int e = bind()) return e;
  if (int e = hip_ok(hipGetDevice(&g.device), "hipGetDevice")) return e;
  UniqueId id;
  memcpy(&id, id_in, sizeof(id));
  if (int e = nccl_ok(g.init(&g.comm, world, id, rank), "ncclCommInitRank")) { g.comm = nullptr; return e; }
  // From here on a failure must give everything back (communicator, stream, the events made so far): the caller sees "not
  // initialised" and may call pcg_dp_init again; pcg_last_error() keeps the text of the FIRST failure.
  g.side = nullptr;
  for (int i = 0; i < kSlots; ++i) { g.ready[i] = nullptr; g.done[i] = nullptr; g.pending[i] = false; }
  int err = hip_ok(hipStreamCreateWithFlags(&g.side, hipStreamNonBlocking), "hipStreamCreate");
  if (err == PCG_OK) err = hip_ok(hipMalloc((void**)&g.token, 64), "hipMalloc(barrier token)");
  if (err == PCG_OK) err = hip_ok(hipMemset(g.token, 0, 64), "hipMemset(barrier token)");
  for (int i = 0; i < kSlots && err == PCG_OK; ++i) {
    err = hip_ok(hipEventCreateWithFlags(&g.ready[i], hipEventDisableTiming), "hipEventCreate");
    if (err == PCG_OK) err = hip_ok(hipEventCreateWithFlags(&g.done[i], hipEventDisableTiming), "hipEventCreate");
  }
  if (err != PCG_OK) {
    release_partial();
    return err;
  }
  g.rank = rank; g.world = world;
  return PCG_OK;
}

extern "C" int32_t pcg_dp_world(void) { return g.comm ? g.world : 0; }
extern "C" int32_t pcg_dp_rank(void) { return g.comm ? g.rank : -1; }
extern "C" pcg_stream_t pcg_dp_side_stream(void) { return g.comm ? (pcg_stream_t)g.side : nullptr; }

extern "C" int pcg_dp_allreduce(float* buf, int64_t n, pcg_stream_t stream) {
  DP_READY("pcg_dp_allreduce");
  PCG_REQUIRE(buf && n > 0, "pcg_dp_allreduce: bad arguments");
  return nccl_ok(g.allreduce(buf, buf, (size_t)n, kFloat, kAvg, g.comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int pcg_dp_allreduce_begin(float* buf, int64_t n, int32_t slot, pcg_stream_t producer_stream) {
  DP_READY("pcg_dp_allreduce_begin");
  PCG_REQUIRE(buf && n > 0 && slot >= 0 && slot < kSlots, "pcg_dp_allreduce_begin: bad arguments (slot %d of %d)", slot, kSlots);
  if (int e = hip_ok(hipEventRecord(g.ready[slot], (hipStream_t)producer_stream), "hipEventRecord")) return e;
  if (int e = hip_ok(hipStreamWaitEvent(g.side, g.ready[slot], 0), "hipStreamWaitEvent")) return e;
  if (int e = nccl_ok(g.allreduce(buf, buf, (size_t)n, kFloat, kAvg, g.comm, g.side), "ncclAllReduce")) return e;
  g.pending[slot] = true;
  return hip_ok(hipEventRecord(g.done[slot], g.side), "hipEventRecord");
}

extern "C" int pcg_dp_record(int32_t slot) {
  DP_READY("pcg_dp_record");
  PCG_REQUIRE(slot >= 0 && slot < kSlots && g.pending[slot], "pcg_dp_record: slot %d has no reduction in flight", slot);
  return hip_ok(hipEventRecord(g.done[slot], g.side), "hipEventRecord");
}

extern "C" int pcg_dp_allreduce_wait(int32_t slot, pcg_stream_t consumer_stream) {
  DP_READY("pcg_dp_allreduce_wait");
  PCG_REQUIRE(slot >= 0 && slot < kSlots, "pcg_dp_allreduce_wait: bad slot %d", slot);
  if (!g.pending[slot]) return PCG_OK;
  g.pending[slot] = false;
  return hip_ok(hipStreamWaitEvent((hipStream_t)consumer_stream, g.done[slot], 0), "hipStreamWaitEvent");
}

extern "C" int pcg_dp_allreduce_sum_f64(double* buf, int64_t n, pcg_stream_t stream) {
  DP_READY("pcg_dp_allreduce_sum_f64");
  PCG_REQUIRE(buf && n > 0, "pcg_dp_allreduce_sum_f64: bad arguments");
  return nccl_ok(g.allreduce(buf, buf, (size_t)n, kDouble, kSum, g.comm, (hipStream_t)stream), "ncclAllReduce(f64)");
}

extern "C" int pcg_dp_broadcast(void* buf, int64_t nbytes, int32_t root, pcg_stream_t stream) {
  DP_READY("pcg_dp_broadcast");
  PCG_REQUIRE(buf && nbytes > 0 && root >= 0 && root < g.world, "pcg_dp_broadcast: bad arguments");
  return nccl_ok(g.bcast(buf, buf, (size_t)nbytes, /*ncclInt8*/ 0, root, g.comm, (hipStream_t)stream), "ncclBroadcast");
}

// A barrier of the ranks ON THE LIBRARY'S COMMUNICATOR: a 4-byte all-reduce in stream order (the caller synchronises the stream).
// bench.py brackets its timed region with it, so the region runs on ONE communicator — the one that carries the exchange.
extern "C" int pcg_dp_barrier(pcg_stream_t stream) {
  DP_READY("pcg_dp_barrier");
  return nccl_ok(g.allreduce(g.token, g.token, 1, kFloat, kSum, g.comm, (hipStream_t)stream), "ncclAllReduce(barrier)");
}
// ncclGetVersion of the RCCL this library bound (e.g. 22105 = 2.21.5), 0 if RCCL is not loaded or too old to say
extern "C" int32_t pcg_dp_rccl_version(void) {
  if (bind() != PCG_OK || !g.version) return 0;
  int v = 0;
  return g.version(&v) == 0 ? v : 0;
}

extern "C" int pcg_dp_sync_batchnorm(int32_t enable) {
  PCG_REQUIRE(!enable || g.comm != nullptr, "pcg_dp_sync_batchnorm: pcg_dp_init has not been called");
  g.sync_bn = enable != 0;
  return PCG_OK;
}

extern "C" int pcg_dp_shutdown(void) {
  g.sync_bn = false;
  if (!g.comm) return PCG_OK;
  (void)hipStreamSynchronize(g.side);
  const int rc = g.destroy(g.comm);
  g.comm = nullptr;
  for (int i = 0; i < kSlots; ++i) { (void)hipEventDestroy(g.ready[i]); (void)hipEventDestroy(g.done[i]); g.pending[i] = false; }
  (void)hipStreamDestroy(g.side);
  if (g.token) (void)hipFree(g.token);
  g.token = nullptr;
  g.side = nullptr; g.rank = -1; g.world = 0;
  return nccl_ok(rc, "ncclCommDestroy");
}
