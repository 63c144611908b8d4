// olap_sharded.hip — the multi-GPU part of libolapgpu's C ABI (include/olap_hip.h, "Multi-GPU"):
// communicators, the sharded drillUp of dimension 0 (local partial + ONE collective + finish) and
// the sharded store handle the Node.js host binds.
//
// Reference call site this sits behind: the per-measure store call of Cube.drillUp,
// /root/reference/src/cube.js:1012-1020 (newCube.storedMeasures[m] = store.drillUp(old, new, rule)),
// store method /root/reference/src/store/in-memory.js:265-334.  The reference itself has no
// distributed code; the partitioning follows BASELINE.json's north_star and SURVEY.md §8(e).
//
// RCCL is bound at run time (dlopen of librccl.so.1): hosts that never shard do not pay for loading
// it, and a process that has torch loaded shares torch's copy (same soname).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "olap_device.hpp"
#include "olap_internal.hpp"

using namespace olap;

// ------------------------------------------------------------------ RCCL, bound lazily
namespace {
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl *rccl() {
  static std::mutex mu;
  static Rccl *r = nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (r) return r;
  r = new Rccl();
  // OLAP_RCCL_LIB names the library to bind instead of the default sonames (a site with its own build; the test of
  // the "RCCL cannot be loaded" path points it at a file that does not exist)
  const char *forced = getenv("OLAP_RCCL_LIB");
  const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  const char *msg = nullptr;
  if (forced && *forced) {
    r->handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    if (!r->handle) msg = dlerror();  // (read once: dlerror() clears the message it returns)
  } else {
    for (const char *n : names) {
      r->handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r->handle) break;
      if (!msg) msg = dlerror();
    }
  }
  if (!r->handle) {
    r->error = std::string("cannot load RCCL: ") + (msg ? msg : "librccl.so.1 not found");
    return r;
  }
#define OLAP_BIND(name)                                                              \
  r->name = (decltype(r->name))dlsym(r->handle, "nccl" #name);                        \
  if (!r->name && r->error.empty()) r->error = "RCCL lacks the symbol nccl" #name;
  OLAP_BIND(GetUniqueId)
  OLAP_BIND(CommInitRank)
  OLAP_BIND(CommInitAll)
  OLAP_BIND(CommDestroy)
  OLAP_BIND(Reduce)
  OLAP_BIND(Broadcast)
  OLAP_BIND(AllReduce)
  OLAP_BIND(ReduceScatter)
  OLAP_BIND(AllGather)
  OLAP_BIND(GroupStart)
  OLAP_BIND(GroupEnd)
  OLAP_BIND(GetErrorString)
#undef OLAP_BIND
  return r;
}

int rccl_fail(ncclResult_t res, const char *what) {
  Rccl *r = rccl();
  return fail(OLAP_ERR_HIP, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(res) : "RCCL error");
}

#define RCCL_TRY(expr)                                          \
  do {                                                          \
    ncclResult_t r__ = (expr);                                  \
    if (r__ != ncclSuccess) return rccl_fail(r__, #expr);       \
  } while (0)

ncclDataType_t nccl_type(int dtype) {
  switch (dtype) {
    case OLAP_INT32: return ncclInt32;
    case OLAP_UINT32: return ncclUint32;
    case OLAP_FLOAT32: return ncclFloat32;
    default: return ncclFloat64;
  }
}
// Integer partial sums are added modulo 2^32 (the typed result of the one-device kernel is the exact float64 sum
// modulo 2^32 as well, Cell<int32_t>::from_f64): as UNSIGNED words, whose wrap-around is defined.
ncclDataType_t nccl_sum_type(int dtype) { return dtype == OLAP_INT32 ? ncclUint32 : nccl_type(dtype); }

enum Transport { TRANSPORT_RCCL = 0, TRANSPORT_DIRECT = 1, TRANSPORT_DETACHED = 2 };
}  // namespace

struct ShardWorkers;  // one issuing thread per local rank (below)

struct olap_comm {
  struct Local {
    int rank = 0;
    int device = 0;
    ncclComm_t nccl = nullptr;
    hipStream_t xstream = nullptr;  // exchange + finishing kernels of this rank
  };
  int world = 1;
  int transport = TRANSPORT_DETACHED;
  std::vector<Local> local;
  ShardWorkers *workers = nullptr;  // one process, several devices: who issues each rank's part of a step
};
static void comm_start_workers(olap_comm *c);
static void comm_stop_workers(olap_comm *c);

static int comm_make_streams(olap_comm *c) {
  DeviceGuard guard;
  for (auto &l : c->local) {
    HIP_TRY(hipSetDevice(l.device));
    HIP_TRY(hipStreamCreateWithFlags(&l.xstream, hipStreamNonBlocking));
  }
  return OLAP_OK;
}

static int check_device_index(int device) {
  int rc = require_device();
  if (rc) return rc;
  const int n = olap_device_count();
  if (device < 0 || device >= n) return fail(OLAP_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device, n);
  return OLAP_OK;
}

extern "C" int olap_comm_unique_id(char id[OLAP_UNIQUE_ID_BYTES]) {
  if (!id) return fail(OLAP_ERR_INVALID_ARGUMENT, "id is NULL");
  static_assert(sizeof(ncclUniqueId) == OLAP_UNIQUE_ID_BYTES, "unique id size");
  Rccl *r = rccl();
  if (!r->error.empty()) return fail(OLAP_ERR_NO_DEVICE, "%s", r->error.c_str());
  ncclUniqueId u;
  RCCL_TRY(r->GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return OLAP_OK;
}

extern "C" int olap_comm_init_rank(olap_comm **comm, const char id[OLAP_UNIQUE_ID_BYTES], int world, int rank, int device) {
  if (!comm || !id) return fail(OLAP_ERR_INVALID_ARGUMENT, "comm/id is NULL");
  *comm = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(OLAP_ERR_INVALID_ARGUMENT, "rank %d outside a world of %d", rank, world);
  int rc = check_device_index(device);
  if (rc) return rc;
  Rccl *r = rccl();
  if (!r->error.empty()) return fail(OLAP_ERR_NO_DEVICE, "%s", r->error.c_str());
  olap_comm *c = new (std::nothrow) olap_comm();
  if (!c) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  c->world = world;
  c->transport = TRANSPORT_RCCL;
  c->local.resize(1);
  c->local[0].rank = rank;
  c->local[0].device = device;
  DeviceGuard guard;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipSetDevice");
  }
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclResult_t res = r->CommInitRank(&c->local[0].nccl, world, u, rank);
  if (res != ncclSuccess) {
    delete c;
    return rccl_fail(res, "ncclCommInitRank");
  }
  if ((rc = comm_make_streams(c))) {
    olap_comm_destroy(c);
    return rc;
  }
  *comm = c;
  return OLAP_OK;
}

extern "C" int olap_comm_init_all(olap_comm **comm, const int *devices, int n) {
  if (!comm || !devices) return fail(OLAP_ERR_INVALID_ARGUMENT, "comm/devices is NULL");
  *comm = nullptr;
  if (n < 1 || n > 1024) return fail(OLAP_ERR_INVALID_ARGUMENT, "device list of %d entries", n);
  int rc;
  for (int i = 0; i < n; ++i)
    if ((rc = check_device_index(devices[i]))) return rc;
  bool all_same = true, all_distinct = true;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      if (devices[i] == devices[j]) all_distinct = false;
      else all_same = false;
    }
  if (!all_distinct && !all_same)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "device list must name distinct devices (RCCL over xGMI) or one device repeated (direct exchange)");
  olap_comm *c = new (std::nothrow) olap_comm();
  if (!c) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  c->world = n;
  c->local.resize(n);
  for (int i = 0; i < n; ++i) {
    c->local[i].rank = i;
    c->local[i].device = devices[i];
  }
  if (all_distinct) {
    Rccl *r = rccl();
    if (!r->error.empty()) {
      delete c;
      return fail(OLAP_ERR_NO_DEVICE, "%s", r->error.c_str());
    }
    c->transport = TRANSPORT_RCCL;
    std::vector<ncclComm_t> comms(n);
    DeviceGuard guard;
    ncclResult_t res = r->CommInitAll(comms.data(), n, devices);
    if (res != ncclSuccess) {
      delete c;
      return rccl_fail(res, "ncclCommInitAll");
    }
    for (int i = 0; i < n; ++i) c->local[i].nccl = comms[i];
  } else {
    c->transport = TRANSPORT_DIRECT;  // every rank on one device: peers' buffers are read directly
  }
  if ((rc = comm_make_streams(c))) {
    olap_comm_destroy(c);
    return rc;
  }
  comm_start_workers(c);
  *comm = c;
  return OLAP_OK;
}

extern "C" int olap_comm_init_detached(olap_comm **comm, int world, int rank, int device) {
  if (!comm) return fail(OLAP_ERR_INVALID_ARGUMENT, "comm is NULL");
  *comm = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(OLAP_ERR_INVALID_ARGUMENT, "rank %d outside a world of %d", rank, world);
  int rc = check_device_index(device);
  if (rc) return rc;
  olap_comm *c = new (std::nothrow) olap_comm();
  if (!c) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  c->world = world;
  c->transport = TRANSPORT_DETACHED;
  c->local.resize(1);
  c->local[0].rank = rank;
  c->local[0].device = device;
  if ((rc = comm_make_streams(c))) {
    olap_comm_destroy(c);
    return rc;
  }
  *comm = c;
  return OLAP_OK;
}

static void drop_cached_steps(const olap_comm *comm);

extern "C" void olap_comm_destroy(olap_comm *c) {
  if (!c) return;
  drop_cached_steps(c);  // steps the store handles keep for this communicator
  comm_stop_workers(c);
  DeviceGuard guard;
  Rccl *r = c->transport == TRANSPORT_RCCL ? rccl() : nullptr;
  for (auto &l : c->local) {
    (void)hipSetDevice(l.device);
    if (l.xstream) {
      (void)hipStreamSynchronize(l.xstream);
      (void)hipStreamDestroy(l.xstream);
    }
    if (l.nccl && r && r->CommDestroy) (void)r->CommDestroy(l.nccl);
  }
  delete c;
}

extern "C" int olap_comm_world(const olap_comm *c) { return c ? c->world : 0; }
extern "C" int olap_comm_local_count(const olap_comm *c) { return c ? (int)c->local.size() : 0; }
extern "C" int olap_comm_local_rank(const olap_comm *c, int local) {
  return (c && local >= 0 && local < (int)c->local.size()) ? c->local[local].rank : -1;
}
extern "C" int olap_comm_local_device(const olap_comm *c, int local) {
  return (c && local >= 0 && local < (int)c->local.size()) ? c->local[local].device : -1;
}
extern "C" const char *olap_comm_transport(const olap_comm *c) {
  if (!c) return "";
  return c->transport == TRANSPORT_RCCL ? "rccl" : c->transport == TRANSPORT_DIRECT ? "direct" : "detached";
}

// ------------------------------------------------------------------ host-only partition arithmetic
extern "C" int olap_shard_bounds(uint32_t n_rows, int world, uint32_t *bounds) {
  if (!bounds) return fail(OLAP_ERR_INVALID_ARGUMENT, "bounds is NULL");
  if (world < 1) return fail(OLAP_ERR_INVALID_ARGUMENT, "world must be >= 1");
  const uint32_t base = n_rows / (uint32_t)world, extra = n_rows % (uint32_t)world;
  bounds[0] = 0;
  for (int r = 0; r < world; ++r) bounds[r + 1] = bounds[r] + base + ((uint32_t)r < extra ? 1u : 0u);
  return OLAP_OK;
}

extern "C" int olap_shard_dice_bounds(const uint32_t *bounds, int world, const int32_t *rows, uint32_t n_sel, uint32_t *new_bounds) {
  if (!bounds || !new_bounds || (n_sel && !rows)) return fail(OLAP_ERR_INVALID_ARGUMENT, "bounds/rows is NULL");
  if (world < 1) return fail(OLAP_ERR_INVALID_ARGUMENT, "world must be >= 1");
  const uint32_t n_rows = bounds[world];
  for (uint32_t i = 0; i < n_sel; ++i) {
    if (rows[i] < 0 || (uint32_t)rows[i] >= n_rows || (i > 0 && rows[i] <= rows[i - 1]))
      return fail(OLAP_ERR_INVALID_ARGUMENT,
                  "sharded: a dice of the sharded dimension takes strictly ascending existing rows (entry %u = %d); gather first", i, rows[i]);
  }
  for (int r = 0; r <= world; ++r)
    new_bounds[r] = (uint32_t)(std::lower_bound(rows, rows + n_sel, (int32_t)std::min<uint32_t>(bounds[r], 0x7FFFFFFFu)) - rows);
  return OLAP_OK;
}

static bool is_float_dtype(int dtype) { return dtype == OLAP_FLOAT32 || dtype == OLAP_FLOAT64; }

extern "C" int olap_shard_recipe_get(int dtype, int default_kind, int method, olap_shard_recipe *rc) {
  if (!rc) return fail(OLAP_ERR_INVALID_ARGUMENT, "recipe is NULL");
  int e;
  if ((e = check_dtype(dtype)) || (e = check_default(default_kind))) return e;
  if (method < OLAP_SUM || method > OLAP_PRODUCT) return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", method);
  memset(rc, 0, sizeof *rc);
  const bool def_nan = default_kind == OLAP_DEFAULT_NAN;
  const bool primary = def_nan && !is_float_dtype(dtype);  // the mask carries information the values cannot
  rc->payload_dtype[0] = dtype;
  rc->payload_dtype[1] = OLAP_INT32;
  if (method == OLAP_SUM && is_float_dtype(dtype)) {
    // The reference adds every contribution of an output cell in float64 and never rounds in between
    // (in-memory.js:282-290, :311-318); so do the one-device kernels.  A partial rounded to Float32 before the ranks
    // are added breaks that under cancellation ([2^24, 1 | -2^24] -> 0 and unset instead of 1 and set), so each rank
    // ships its float64 ACCUMULATOR, the ranks are added in float64 and the sum is rounded to the cell type ONCE.
    rc->local_method = OLAP_PARTIAL_AVERAGE;  // (float64 sum of the set cells, contribution count)
    rc->payload_dtype[0] = OLAP_FLOAT64;
    rc->payload_op[0] = OLAP_XCHG_SUM;
    rc->payload_op[1] = OLAP_XCHG_SUM;
    rc->n_payloads = def_nan ? 2 : 1;  // 0 default: set <=> the rounded sum != 0; NaN default: somebody contributed
    rc->finish = OLAP_FINISH_ROUND;
  } else if (method == OLAP_SUM) {
    // integer cells are added modulo 2^32 on every path (exact): the typed partials travel
    rc->local_method = OLAP_SUM;
    rc->payload_op[0] = OLAP_XCHG_SUM;
    if (!def_nan) {  // set <=> value != 0: a rank without contributions ships 0, the neutral element
      rc->n_payloads = 1;
      rc->finish = OLAP_FINISH_NONE;
    } else {  // the mask is primary (no integer equals NaN): the masks are OR-ed (MAX of 0 / 0x2), never added
      rc->n_payloads = 2;
      rc->payload_op[1] = OLAP_XCHG_MAX;
      rc->finish = OLAP_FINISH_RESTORE;
    }
  } else if (method == OLAP_AVERAGE) {
    rc->local_method = OLAP_PARTIAL_AVERAGE;  // (float64 sum, contribution count) in one local pass; unset cells ship 0
    rc->payload_dtype[0] = OLAP_FLOAT64;      // every cell type: an Int32 sum may pass 2^31 before it is divided
    rc->n_payloads = 2;
    rc->payload_op[0] = OLAP_XCHG_SUM;
    rc->payload_op[1] = OLAP_XCHG_SUM;
    rc->finish = OLAP_FINISH_AVERAGE;
  } else {
    // highest / lowest / first / last / product are associative in rank order (ranks own ascending
    // row ranges): gather the partials, run the same drillUp over the rank axis
    rc->local_method = method;
    rc->n_payloads = primary ? 2 : 1;  // elsewhere set <=> value != default
    rc->payload_op[0] = OLAP_XCHG_GATHER;
    rc->payload_op[1] = OLAP_XCHG_GATHER;
    rc->finish = OLAP_FINISH_COMBINE;
  }
  return OLAP_OK;
}

// ------------------------------------------------------------------ small kernels of the exchange
namespace {
// after a SUM of partials + MAX of masks (integer cells over a NaN default): cells nobody contributed to get the
// default back.  flags == nullptr: the mask is a function of the values (set <=> value != default).  In place
// (src == dst, flags == dst_status) or into a destination of its own.
template <typename T>
__global__ __launch_bounds__(kBlock) void restore_default_kernel(const T *src, const int32_t *flags, T *dst, int32_t *dst_status, uint64_t n,
                                                                 int def_nan) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const T v = src[i];
    const bool set = (!flags || (flags[i] & OLAP_STATUS_SET) != 0) && !Cell<T>::is_default(v, def_nan != 0);
    dst[i] = set ? v : Cell<T>::default_value(def_nan != 0);
    if (dst_status) dst_status[i] = set ? OLAP_STATUS_SET : 0;
  }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void partial_round_kernel(const double *sums, const int32_t *counts, T *values, int32_t *status,
                                                               uint64_t n, int def_nan_i, int finish) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    T ov;
    int32_t os;
    finish_cell<T, double>(finish, sums[i], counts ? (uint32_t)counts[i] : 0u, counts != nullptr, def_nan_i != 0, ov, os);
    values[i] = ov;
    if (status) status[i] = os;
  }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void fill_default_kernel(T *values, uint64_t n, int def_nan) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    values[i] = Cell<T>::default_value(def_nan != 0);
}
// Integer partial sums are added modulo 2^32, as unsigned words (see nccl_sum_type)
template <typename P> struct SumWord { typedef P type; };
template <> struct SumWord<int32_t> { typedef uint32_t type; };

// direct transport: dst[i] = op over the ranks q of src[q][first + i]
template <typename T, int OP>
__global__ __launch_bounds__(kBlock) void direct_combine_kernel(const T *const *src, int n_src, T *dst, uint64_t first, uint64_t n) {
  typedef typename SumWord<T>::type W;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    T acc = src[0][first + i];
    for (int q = 1; q < n_src; ++q) {
      const T v = src[q][first + i];
      if constexpr (OP == OLAP_XCHG_SUM) acc = (T)((W)acc + (W)v);
      else acc = v > acc ? v : acc;
    }
    dst[i] = acc;
  }
}

// Direct transport, every rank on ONE device: the exchange and the finishing kernel of ALL ranks as one launch.  A
// cell's payloads are added over the ranks in rank order (float64 partials: in float64), finished once, and written to
// every destination whose block holds it (SCATTER: the owner's; ALL: everybody's; ROOT: rank 0's).
constexpr int kFusedMaxRanks = 16;
constexpr int kMaxBatchRanks = 8;  // pairs olap_plan_run_batch puts into one launch
struct FusedDests {
  int n;
  void *values[kFusedMaxRanks];
  int32_t *status[kFusedMaxRanks];
  uint64_t first[kFusedMaxRanks], count[kFusedMaxRanks];
};
template <typename T, typename P>
__global__ __launch_bounds__(kBlock) void direct_fused_kernel(const P *const *src0, const int32_t *const *src1, int world, int op1, uint64_t n_out,
                                                              int finish, int def_nan, const FusedDests d) {
  typedef typename SumWord<P>::type W;
  for (uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x; c < n_out; c += (uint64_t)gridDim.x * kBlock) {
    W a = (W)src0[0][c];
    for (int q = 1; q < world; ++q) a = a + (W)src0[q][c];
    uint32_t b = 0;
    if (src1) {
      int32_t m = src1[0][c];
      for (int q = 1; q < world; ++q) {
        const int32_t v = src1[q][c];
        m = op1 == OLAP_XCHG_SUM ? (int32_t)((uint32_t)m + (uint32_t)v) : (v > m ? v : m);
      }
      b = (uint32_t)m;
    }
    T ov;
    int32_t os;
    finish_cell<T, P>(finish, (P)a, b, src1 != nullptr, def_nan != 0, ov, os);
    for (int j = 0; j < d.n; ++j) {
      const uint64_t rel = c - d.first[j];
      if (c >= d.first[j] && rel < d.count[j]) {
        ((T *)d.values[j])[rel] = ov;
        if (d.status[j]) d.status[j][rel] = os;
      }
    }
  }
}

unsigned grid_for_n(uint64_t n) {
  const uint64_t want = (n + kBlock - 1) / kBlock;
  return (unsigned)(want < 1 ? 1 : (want < 2048 ? want : 2048));
}

#define SHARD_DISPATCH(dtype, CALL)                        \
  switch (dtype) {                                         \
    case OLAP_INT32: { using T = int32_t; CALL; break; }   \
    case OLAP_UINT32: { using T = uint32_t; CALL; break; } \
    case OLAP_FLOAT32: { using T = float; CALL; break; }   \
    default: { using T = double; CALL; break; }            \
  }

int launch_check(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return OLAP_OK;
}
}  // namespace

// ------------------------------------------------------------------ one issuing thread per device
// One process that drives n devices (the Node.js host: olap_comm_init_all) would otherwise issue every device's part
// of a step from ONE thread — device switch, launch, collective, launch, n times over: at 8 devices several hundred
// microseconds of host time against an 80 us slab.  Each local rank gets a worker thread that
// made its device current once and issues its own rank's sequence; the caller's thread only hands the step over and
// waits until every worker has ISSUED (not finished) its part.  Workers spin briefly after a step before they
// sleep, so a burst of queries does not pay a wake-up per step.  OLAP_SHARD_THREADS=0 turns them off (everything is
// issued by the calling thread), =1 also gives the ranks of the direct transport workers (tests: the hand-over, the
// barrier and the error relay run on a one-GPU box).
namespace {
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
}  // namespace

struct ShardWorkers {
  struct Slot {
    int rc = 0;
    std::string err;
  };
  int n = 0;
  std::vector<std::thread> threads;
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> done{0};
  std::atomic<int> arrived{0};
  std::atomic<uint64_t> barrier_gen{0};
  std::atomic<bool> stop{false};
  const std::function<int(int)> *job = nullptr;
  static constexpr int64_t kSpinNs = 200 * 1000;  // how long a worker (and the caller) spins before it sleeps

  static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  void start(const std::vector<int> &devices) {
    n = (int)devices.size();
    slots.resize(n);
    for (int i = 0; i < n; ++i) threads.emplace_back([this, i, dev = devices[i]] { loop(i, dev); });
  }
  ~ShardWorkers() {
    {
      std::lock_guard<std::mutex> lock(mu);
      stop.store(true);
      gen.fetch_add(1);
    }
    cv_go.notify_all();
    for (auto &t : threads) t.join();
  }
  void loop(int i, int device) {
    (void)hipSetDevice(device);
    uint64_t seen = 0;
    for (;;) {
      const int64_t t0 = now_ns();
      int polls = 0;
      while (gen.load(std::memory_order_acquire) == seen) {
        cpu_relax();
        if ((++polls & 63) == 0 && now_ns() - t0 > kSpinNs) {
          std::unique_lock<std::mutex> lock(mu);
          cv_go.wait(lock, [&] { return gen.load(std::memory_order_acquire) != seen; });
        }
      }
      if (stop.load()) return;
      seen = gen.load(std::memory_order_acquire);
      const int rc = (*job)(i);
      slots[i].rc = rc;
      if (rc) slots[i].err = olap_last_error();
      if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
        std::lock_guard<std::mutex> lock(mu);
        cv_done.notify_one();
      }
    }
  }
  // runs fn(local rank) on every worker; returns the first failure (its message becomes the caller's olap_last_error())
  int run(const std::function<int(int)> &fn) {
    job = &fn;
    done.store(0, std::memory_order_relaxed);
    for (auto &s : slots) s.rc = 0;
    {
      std::lock_guard<std::mutex> lock(mu);
      gen.fetch_add(1, std::memory_order_release);
    }
    cv_go.notify_all();
    const int64_t t0 = now_ns();
    int polls = 0;
    while (done.load(std::memory_order_acquire) != n) {
      cpu_relax();
      if ((++polls & 63) == 0 && now_ns() - t0 > kSpinNs) {
        std::unique_lock<std::mutex> lock(mu);
        cv_done.wait(lock, [&] { return done.load(std::memory_order_acquire) == n; });
      }
    }
    job = nullptr;
    for (auto &s : slots)
      if (s.rc) return fail(s.rc, "%s", s.err.c_str());
    return OLAP_OK;
  }
  // all workers of the current step meet here (they are all running: a spin barrier)
  void barrier() {
    const uint64_t g = barrier_gen.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
      arrived.store(0, std::memory_order_relaxed);
      barrier_gen.fetch_add(1, std::memory_order_release);
    } else {
      while (barrier_gen.load(std::memory_order_acquire) == g) cpu_relax();
    }
  }
};

static void comm_start_workers(olap_comm *c) {
  if (c->local.size() < 2) return;
  const char *e = getenv("OLAP_SHARD_THREADS");
  const int forced = e ? atoi(e) : -1;
  if (forced == 0) return;
  if (c->transport != TRANSPORT_RCCL && forced != 1) return;  // ranks on one device: one fused launch instead (step_direct_fused)
  std::vector<int> devices;
  for (auto &l : c->local) devices.push_back(l.device);
  c->workers = new ShardWorkers();
  c->workers->start(devices);
}
static void comm_stop_workers(olap_comm *c) {
  delete c->workers;
  c->workers = nullptr;
}

// ------------------------------------------------------------------ sharded drillUp of dimension 0
namespace {
struct BufSet {
  void *send[2] = {nullptr, nullptr};      // partial values / partial mask-or-counts (direct transport: slices of send_base)
  void *send_base[2] = {nullptr, nullptr}; // direct transport, local rank 0 only: every rank's partial, rank after rank
  void *recv[2] = {nullptr, nullptr};      // combined (or gathered) payloads
  void *result = nullptr;                  // FINISH_COMBINE / _ROUND / _AVERAGE (else the result is recv[0] in place)
  int32_t *result_status = nullptr;
  const void **peers[2] = {nullptr, nullptr};  // direct transport: device table of every rank's send[p]
  hipEvent_t local_done = nullptr, xchg_done = nullptr;
  bool in_flight = false;
};
struct RankState {
  olap_plan *local_plan = nullptr;
  olap_plan *combine_plan = nullptr;
  uint64_t local_cells = 0;
  BufSet set[2];
};
// where a rank's finished cells go when the caller provides the buffers (the store handles: straight into the
// result store, no copy): global output cells [first, first + count), inside the rank's block
struct StepDest {
  void *values = nullptr;
  int32_t *status = nullptr;
  uint64_t first = 0, count = 0;
};
}  // namespace

struct olap_shard_drillup {
  olap_comm *comm = nullptr;
  olap_shard_recipe recipe{};
  int dtype = 0, default_kind = 0, method = 0, placement = 0, depth = 1;
  uint64_t n_out = 0, per = 0, n_send = 0;
  bool mask_primary = false;
  bool fused = false;  // direct transport with few ranks: exchange + finish of all ranks is one launch
  bool same_plans = false;  // ... and every rank's slab has the same rows-to-groups map: the local passes are one launch too
  int cur = 0;       // buffer set of the last step
  std::vector<RankState> ranks;
};

// the typed result lives in a buffer of its own (gathered partials; float64 partials rounded once at the end)
static bool separate_result(int finish) { return finish == OLAP_FINISH_COMBINE || finish == OLAP_FINISH_ROUND || finish == OLAP_FINISH_AVERAGE; }

static bool is_scatter(int placement) { return placement == OLAP_PLACE_SCATTER || placement == OLAP_PLACE_SCATTER_ROWS; }

static size_t payload_size(const olap_shard_drillup *op, int p) { return olap_dtype_size(op->recipe.payload_dtype[p]); }

// cells of recv[p] on a rank
static uint64_t recv_cells(const olap_shard_drillup *op, int p) {
  if (op->recipe.payload_op[p] == OLAP_XCHG_GATHER) return op->n_out * (uint64_t)op->comm->world;
  return is_scatter(op->placement) ? op->per : op->n_out;
}

// global output cells [first, first + count) that `rank` holds after a step
static void rank_block(const olap_shard_drillup *op, int rank, uint64_t *first, uint64_t *count) {
  uint64_t f = 0, n = op->n_out;
  if (is_scatter(op->placement)) {
    f = std::min<uint64_t>((uint64_t)rank * op->per, op->n_out);
    n = std::min<uint64_t>(op->per, op->n_out - f);
  } else if (op->placement == OLAP_PLACE_ROOT && rank != 0) {
    n = 0;
  }
  *first = f;
  *count = n;
}

extern "C" void olap_shard_drillup_destroy(olap_shard_drillup *op) {
  if (!op) return;
  DeviceGuard guard;
  const bool direct = op->comm->transport == TRANSPORT_DIRECT;
  for (size_t i = 0; i < op->ranks.size(); ++i) {
    const auto &l = op->comm->local[i];
    (void)hipSetDevice(l.device);
    (void)hipStreamSynchronize(l.xstream);
    (void)hipStreamSynchronize(nullptr);
    RankState &rs = op->ranks[i];
    if (rs.local_plan) olap_plan_destroy(rs.local_plan);
    if (rs.combine_plan) olap_plan_destroy(rs.combine_plan);
    for (BufSet &b : rs.set) {
      for (int p = 0; p < 2; ++p) {
        if (b.send[p] && !direct) dev_free(b.send[p]);
        if (b.send_base[p]) dev_free(b.send_base[p]);
        if (b.recv[p]) dev_free(b.recv[p]);
        if (b.peers[p]) dev_free((void *)b.peers[p]);
      }
      if (b.result) dev_free(b.result);
      if (b.result_status) dev_free(b.result_status);
      if (b.local_done) (void)hipEventDestroy(b.local_done);
      if (b.xchg_done) (void)hipEventDestroy(b.xchg_done);
    }
  }
  delete op;
}

extern "C" int olap_shard_drillup_create(olap_shard_drillup **out, olap_comm *comm, int dtype, int default_kind, int method,
                                         int ndim, const uint32_t *lens, const uint32_t *new_len, const uint32_t *bounds,
                                         const uint32_t *const *maps, int placement, int depth) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "op out-pointer is NULL");
  *out = nullptr;
  if (!comm) return fail(OLAP_ERR_INVALID_ARGUMENT, "comm is NULL");
  if (ndim < 1 || ndim > OLAP_MAX_DIMS || !lens || !new_len || !bounds || !maps)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "a sharded drillUp needs at least one dimension, its lengths, bounds and maps");
  if (placement < OLAP_PLACE_SCATTER || placement > OLAP_PLACE_SCATTER_ROWS) return fail(OLAP_ERR_INVALID_ARGUMENT, "placement %d", placement);
  if (depth != 1 && depth != 2) return fail(OLAP_ERR_INVALID_ARGUMENT, "depth must be 1 or 2");
  olap_shard_recipe recipe;
  int rc = olap_shard_recipe_get(dtype, default_kind, method, &recipe);
  if (rc) return rc;
  const int world = comm->world;
  if (bounds[0] != 0 || bounds[world] != lens[0]) return fail(OLAP_ERR_INVALID_ARGUMENT, "bounds must run from 0 to the extent of dimension 0");
  for (int r = 0; r < world; ++r)
    if (bounds[r + 1] < bounds[r]) return fail(OLAP_ERR_INVALID_ARGUMENT, "bounds must ascend");
  for (int d = 0; d < ndim; ++d) {
    if (lens[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
    for (uint32_t k = 0; k < lens[d]; ++k)
      if (maps[d][k] >= new_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillUp map of dimension %d: entry %u = %u is outside the new dimension (%u items)", d, k, maps[d][k], new_len[d]);
  }
  if ((rc = require_device())) return rc;

  olap_shard_drillup *op = new (std::nothrow) olap_shard_drillup();
  if (!op) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  op->comm = comm;
  op->recipe = recipe;
  op->dtype = dtype;
  op->default_kind = default_kind;
  op->method = method;
  op->depth = depth;
  op->mask_primary = default_kind == OLAP_DEFAULT_NAN && !is_float_dtype(dtype);
  // gathered partials are combined on every rank, so their result is whole everywhere
  op->placement = recipe.finish == OLAP_FINISH_COMBINE ? OLAP_PLACE_ALL : placement;
  uint64_t n_out = 1;
  for (int d = 0; d < ndim; ++d) n_out *= new_len[d];
  op->n_out = n_out;
  op->per = (n_out + (uint64_t)world - 1) / (uint64_t)world;
  if (op->placement == OLAP_PLACE_SCATTER_ROWS) {  // blocks of whole rows of the new leading dimension
    const uint64_t row = new_len[0] ? n_out / new_len[0] : 0;
    op->per = ((uint64_t)new_len[0] + (uint64_t)world - 1) / (uint64_t)world * row;
  }
  op->n_send = is_scatter(op->placement) ? op->per * (uint64_t)world : n_out;  // padded so that it divides
  op->ranks.resize(comm->local.size());
  const bool direct = comm->transport == TRANSPORT_DIRECT;
  op->fused = direct && world <= kFusedMaxRanks && !comm->workers && !getenv("OLAP_SHARD_NO_FUSED");

  DeviceGuard guard;
  const bool def_nan = default_kind == OLAP_DEFAULT_NAN;
  std::vector<uint32_t> local_len(lens, lens + ndim);
  std::vector<const uint32_t *> local_maps(maps, maps + ndim);
  for (size_t i = 0; i < op->ranks.size() && !rc; ++i) {
    const auto &l = comm->local[i];
    RankState &rs = op->ranks[i];
    hipError_t e = hipSetDevice(l.device);
    if (e != hipSuccess) {
      rc = hip_fail(e, "hipSetDevice");
      break;
    }
    const uint32_t lo = bounds[l.rank], hi = bounds[l.rank + 1];
    local_len[0] = hi - lo;
    local_maps[0] = maps[0] + lo;
    rs.local_cells = 1;
    for (int d = 0; d < ndim; ++d) rs.local_cells *= local_len[d];
    if (rs.local_cells > 0)
      rc = olap_drillup_plan(&rs.local_plan, dtype, default_kind, recipe.local_method, ndim, local_len.data(), new_len, local_maps.data());
    if (rc) break;
    if (recipe.finish == OLAP_FINISH_COMBINE) {
      // the same drillUp over the rank axis: [world, new_len...] -> [1, new_len...]
      if (ndim + 1 > OLAP_MAX_DIMS) {
        rc = fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: too many dimensions for a gathered combine");
        break;
      }
      std::vector<uint32_t> cl_old{(uint32_t)world}, cl_new{1u};
      std::vector<std::vector<uint32_t>> tabs;
      tabs.emplace_back((size_t)world, 0u);
      for (int d = 0; d < ndim; ++d) {
        cl_old.push_back(new_len[d]);
        cl_new.push_back(new_len[d]);
        tabs.emplace_back((size_t)new_len[d]);
        for (uint32_t k = 0; k < new_len[d]; ++k) tabs.back()[k] = k;
      }
      std::vector<const uint32_t *> cm;
      for (auto &t : tabs) cm.push_back(t.data());
      rc = olap_drillup_plan(&rs.combine_plan, dtype, default_kind, method, ndim + 1, cl_old.data(), cl_new.data(), cm.data());
      if (rc) break;
    }
    for (int k = 0; k < depth && !rc; ++k) {
      BufSet &b = rs.set[k];
      for (int p = 0; p < recipe.n_payloads && !rc; ++p) {
        const size_t es = payload_size(op, p);
        const uint64_t cells = std::max<uint64_t>(op->n_send, 1);
        if (direct) {
          // every rank's partial in ONE block, rank after rank (the ranks share the device): gathered partials need no
          // copy at all, and one table names them for the fused kernel
          if (i == 0) {
            e = dev_alloc(&b.send_base[p], cells * es * (size_t)world);
            if (e == hipSuccess) e = hipMemsetAsync(b.send_base[p], 0, cells * es * (size_t)world, nullptr);
          }
          b.send[p] = (char *)op->ranks[0].set[k].send_base[p] + (size_t)l.rank * cells * es;
        } else {
          e = dev_alloc(&b.send[p], cells * es);
          if (e == hipSuccess) e = hipMemsetAsync(b.send[p], 0, cells * es, nullptr);
        }
        // (fused direct steps read the partials where they lie; gathered partials of the direct transport too)
        const bool needs_recv = !(op->placement == OLAP_PLACE_ROOT && l.rank != 0 && recipe.payload_op[p] != OLAP_XCHG_GATHER) &&
                                !(direct && (op->fused || recipe.payload_op[p] == OLAP_XCHG_GATHER));
        if (e == hipSuccess && needs_recv) e = dev_alloc(&b.recv[p], std::max<uint64_t>(recv_cells(op, p), 1) * es);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(sharded drillUp buffers)");
      }
      if (!rc && rs.local_cells == 0 && def_nan && is_float_dtype(dtype) && recipe.payload_op[0] == OLAP_XCHG_GATHER) {
        // a rank without rows ships "unset everywhere": the canonical default
        SHARD_DISPATCH(dtype, hipLaunchKernelGGL((fill_default_kernel<T>), grid_for_n(n_out), kBlock, 0, nullptr, (T *)b.send[0], n_out, 1));
        rc = launch_check("fill_default_kernel");
      }
      if (!rc && (separate_result(recipe.finish) || op->fused) && !(op->placement == OLAP_PLACE_ROOT && l.rank != 0)) {
        const uint64_t cells = recipe.finish == OLAP_FINISH_COMBINE ? n_out : recv_cells(op, 0);
        e = dev_alloc(&b.result, std::max<uint64_t>(cells, 1) * olap_dtype_size(dtype));
        if (e == hipSuccess) e = dev_alloc((void **)&b.result_status, std::max<uint64_t>(cells, 1) * sizeof(int32_t));
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(sharded drillUp result)");
      }
      if (!rc) {
        e = hipEventCreateWithFlags(&b.local_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b.xchg_done, hipEventDisableTiming);
        if (e != hipSuccess) rc = hip_fail(e, "hipEventCreate");
      }
    }
    if (!rc) {
      e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
    }
  }
  if (!rc && op->fused && world <= kMaxBatchRanks) {
    // equal slabs with equal row maps (the '-> all' roll-ups of a balanced partition): one plan serves every rank
    bool same = true;
    const uint32_t rows = bounds[1] - bounds[0];
    for (int r = 1; r < world && same; ++r)
      same = bounds[r + 1] - bounds[r] == rows && std::equal(maps[0] + bounds[r], maps[0] + bounds[r + 1], maps[0] + bounds[0]);
    op->same_plans = same && rows > 0;
  }
  // direct transport: every rank's table of its peers' send buffers
  if (!rc && direct) {
    for (size_t i = 0; i < op->ranks.size() && !rc; ++i) {
      if (op->fused && i != 0) break;  // one table serves the fused launch
      (void)hipSetDevice(comm->local[i].device);
      for (int k = 0; k < depth && !rc; ++k)
        for (int p = 0; p < recipe.n_payloads && !rc; ++p) {
          std::vector<const void *> tab(world);
          for (int q = 0; q < world; ++q) tab[q] = op->ranks[q].set[k].send[p];  // all ranks are local, local index == rank
          void *dev = nullptr;
          hipError_t e = dev_alloc(&dev, tab.size() * sizeof(void *));
          if (e == hipSuccess) e = hipMemcpy(dev, tab.data(), tab.size() * sizeof(void *), hipMemcpyHostToDevice);
          if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(peer table)");
          op->ranks[i].set[k].peers[p] = (const void **)dev;
        }
    }
  }
  if (rc) {
    olap_shard_drillup_destroy(op);
    return rc;
  }
  *out = op;
  return OLAP_OK;
}

extern "C" uint64_t olap_shard_drillup_out_cells(const olap_shard_drillup *op) { return op ? op->n_out : 0; }
extern "C" uint64_t olap_shard_drillup_local_cells(const olap_shard_drillup *op, int local) {
  return (op && local >= 0 && local < (int)op->ranks.size()) ? op->ranks[local].local_cells : 0;
}
extern "C" const char *olap_shard_drillup_kernel_name(const olap_shard_drillup *op, int local) {
  if (!op || local < 0 || local >= (int)op->ranks.size() || !op->ranks[local].local_plan) return "";
  return olap_plan_kernel_name(op->ranks[local].local_plan);
}

static int check_local(const olap_shard_drillup *op, int local) {
  if (!op) return fail(OLAP_ERR_INVALID_ARGUMENT, "op is NULL");
  if (local < 0 || local >= (int)op->ranks.size()) return fail(OLAP_ERR_INVALID_ARGUMENT, "local rank %d outside [0, %d)", local, (int)op->ranks.size());
  return OLAP_OK;
}

// ---- the three phases of one rank (the rank's device is current) ------------------------------------------------------
// local phase into buffer set k
static int shard_local(olap_shard_drillup *op, int local, int k, const void *in_values, const int32_t *in_status, hipStream_t stream) {
  RankState &rs = op->ranks[local];
  BufSet &b = rs.set[k];
  if (!rs.local_plan) return OLAP_OK;  // no rows: the buffers hold "unset" since creation
  if (op->mask_primary && !in_status)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "integer cells over a NaN default: the status mask is required");
  return olap_plan_run(rs.local_plan, in_values, in_status, b.send[0], op->recipe.n_payloads > 1 ? (int32_t *)b.send[1] : nullptr, stream);
}

extern "C" int olap_shard_drillup_local(olap_shard_drillup *op, int local, const void *in_values, const int32_t *in_status, void *stream) {
  int rc = check_local(op, local);
  if (rc) return rc;
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(op->comm->local[local].device));
  return shard_local(op, local, op->cur, in_values, in_status, (hipStream_t)stream);
}

// finishing kernels of buffer set k on `stream`; dest (optional): the caller's buffers instead of the op's own
static int shard_finish(olap_shard_drillup *op, int local, int k, hipStream_t stream, const StepDest *dest) {
  RankState &rs = op->ranks[local];
  BufSet &b = rs.set[k];
  const auto &l = op->comm->local[local];
  uint64_t first = 0, n = 0;
  rank_block(op, l.rank, &first, &n);
  if (n == 0) return OLAP_OK;  // nothing arrived here
  if (dest && !dest->values) return OLAP_OK;  // the caller does not want this rank's copy
  const bool def_nan = op->default_kind == OLAP_DEFAULT_NAN;
  const size_t es = olap_dtype_size(op->dtype);
  // where the finished cells go, and which of the rank's cells
  uint64_t off = 0, cnt = is_scatter(op->placement) ? op->per : op->n_out;  // (in place: the whole padded block)
  void *ov = nullptr;
  int32_t *os = nullptr;
  if (dest) {
    if (dest->first < first || dest->first + dest->count > first + n)
      return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the result block of a rank is not where it was expected");
    off = dest->first - first;
    cnt = dest->count;
    ov = dest->values;
    os = dest->status;
    if (cnt == 0) return OLAP_OK;
  }
  int rc = OLAP_OK;
  switch (op->recipe.finish) {
    case OLAP_FINISH_NONE:
      if (!dest) break;  // the reduced values are the result where they lie
      SHARD_DISPATCH(op->dtype, hipLaunchKernelGGL((restore_default_kernel<T>), grid_for_n(cnt), kBlock, 0, stream, (const T *)b.recv[0] + off,
                                                   (const int32_t *)nullptr, (T *)ov, os, cnt, def_nan));
      rc = launch_check("restore_default_kernel");
      break;
    case OLAP_FINISH_RESTORE:
      SHARD_DISPATCH(op->dtype, hipLaunchKernelGGL((restore_default_kernel<T>), grid_for_n(cnt), kBlock, 0, stream, (const T *)b.recv[0] + off,
                                                   (const int32_t *)b.recv[1] + off, dest ? (T *)ov : (T *)b.recv[0],
                                                   dest ? os : (int32_t *)b.recv[1], cnt, def_nan));
      rc = launch_check("restore_default_kernel");
      break;
    case OLAP_FINISH_ROUND:
    case OLAP_FINISH_AVERAGE:
      SHARD_DISPATCH(op->dtype, hipLaunchKernelGGL((partial_round_kernel<T>), grid_for_n(cnt), kBlock, 0, stream, (const double *)b.recv[0] + off,
                                                   op->recipe.n_payloads > 1 ? (const int32_t *)b.recv[1] + off : nullptr,
                                                   dest ? (T *)ov : (T *)b.result, dest ? os : b.result_status, cnt, def_nan, op->recipe.finish));
      rc = launch_check("partial_round_kernel");
      break;
    case OLAP_FINISH_COMBINE: {
      // (the gathered partials of the direct transport lie rank after rank in the shared block: no copy)
      const bool direct = op->comm->transport == TRANSPORT_DIRECT;
      const void *gv = direct ? op->ranks[0].set[k].send_base[0] : b.recv[0];
      const int32_t *gs = op->recipe.n_payloads > 1 ? (const int32_t *)(direct ? op->ranks[0].set[k].send_base[1] : b.recv[1]) : nullptr;
      const bool whole = dest && off == 0 && cnt == op->n_out;
      rc = olap_plan_run(rs.combine_plan, gv, gs, whole ? ov : b.result, whole ? os : b.result_status, stream);
      if (!rc && dest && !whole) {  // a result that stays sharded: this rank keeps its rows of the combined cube
        hipError_t e = hipMemcpyAsync(ov, (const char *)b.result + off * es, cnt * es, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess && os) e = hipMemcpyAsync(os, b.result_status + off, cnt * sizeof(int32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) rc = hip_fail(e, "sharded drillUp result");
      }
      break;
    }
  }
  return rc;
}

extern "C" int olap_shard_drillup_finish(olap_shard_drillup *op, int local, void *stream) {
  int rc = check_local(op, local);
  if (rc) return rc;
  if (op->fused) return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the ranks of this communicator share a device and finish in one launch (olap_shard_drillup_step)");
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(op->comm->local[local].device));
  return shard_finish(op, local, op->cur, (hipStream_t)stream, nullptr);
}

// RCCL collectives of local rank i for buffer set k on stream x (the caller opens / closes the group)
static ncclResult_t rank_collectives(olap_shard_drillup *op, size_t i, int k, hipStream_t x) {
  Rccl *r = rccl();
  const olap_shard_recipe &rp = op->recipe;
  BufSet &b = op->ranks[i].set[k];
  const auto &l = op->comm->local[i];
  ncclResult_t res = ncclSuccess;
  for (int p = 0; p < rp.n_payloads && res == ncclSuccess; ++p) {
    const ncclDataType_t ty = rp.payload_op[p] == OLAP_XCHG_SUM ? nccl_sum_type(rp.payload_dtype[p]) : nccl_type(rp.payload_dtype[p]);
    if (rp.payload_op[p] == OLAP_XCHG_GATHER) {
      res = r->AllGather(b.send[p], b.recv[p], op->n_out, ty, l.nccl, x);
      continue;
    }
    const ncclRedOp_t red = rp.payload_op[p] == OLAP_XCHG_SUM ? ncclSum : ncclMax;
    if (is_scatter(op->placement)) res = r->ReduceScatter(b.send[p], b.recv[p], op->per, ty, red, l.nccl, x);
    else if (op->placement == OLAP_PLACE_ALL) res = r->AllReduce(b.send[p], b.recv[p], op->n_out, ty, red, l.nccl, x);
    else res = r->Reduce(b.send[p], b.recv[p] ? b.recv[p] : b.send[p], op->n_out, ty, red, 0, l.nccl, x);
  }
  return res;
}

// direct transport, rank i: its block of every payload combined from the peers' partials (the rank's device is current)
static int rank_direct_combine(olap_shard_drillup *op, size_t i, int k, hipStream_t x) {
  olap_comm *c = op->comm;
  const olap_shard_recipe &rp = op->recipe;
  BufSet &b = op->ranks[i].set[k];
  const auto &l = c->local[i];
  for (int p = 0; p < rp.n_payloads; ++p) {
    if (rp.payload_op[p] == OLAP_XCHG_GATHER) continue;  // read in place by the combine plan (shared block)
    if (op->placement == OLAP_PLACE_ROOT && l.rank != 0) continue;
    const uint64_t first = is_scatter(op->placement) ? (uint64_t)l.rank * op->per : 0;
    const uint64_t n = is_scatter(op->placement) ? op->per : op->n_out;
    const bool sum = rp.payload_op[p] == OLAP_XCHG_SUM;
#define DIRECT_LAUNCH(T)                                                                                                     \
  do {                                                                                                                       \
    if (sum) hipLaunchKernelGGL((direct_combine_kernel<T, OLAP_XCHG_SUM>), grid_for_n(n), kBlock, 0, x,                       \
                                (const T *const *)b.peers[p], c->world, (T *)b.recv[p], first, n);                           \
    else hipLaunchKernelGGL((direct_combine_kernel<T, OLAP_XCHG_MAX>), grid_for_n(n), kBlock, 0, x,                           \
                            (const T *const *)b.peers[p], c->world, (T *)b.recv[p], first, n);                               \
  } while (0)
    switch (rp.payload_dtype[p]) {
      case OLAP_INT32: DIRECT_LAUNCH(int32_t); break;
      case OLAP_UINT32: DIRECT_LAUNCH(uint32_t); break;
      case OLAP_FLOAT32: DIRECT_LAUNCH(float); break;
      default: DIRECT_LAUNCH(double); break;
    }
#undef DIRECT_LAUNCH
    int rc = launch_check("direct_combine_kernel");
    if (rc) return rc;
  }
  return OLAP_OK;
}

// exchange of buffer set k for every local rank from ONE thread; xs[i] = the stream rank i's collective is enqueued on
static int shard_exchange(olap_shard_drillup *op, int k, const std::vector<hipStream_t> &xs) {
  olap_comm *c = op->comm;
  if (c->transport == TRANSPORT_DETACHED)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: a detached communicator has no transport; move the payloads yourself (olap_shard_drillup_payload)");
  if (c->transport == TRANSPORT_RCCL) {
    Rccl *r = rccl();
    RCCL_TRY(r->GroupStart());
    ncclResult_t res = ncclSuccess;
    for (size_t i = 0; i < op->ranks.size() && res == ncclSuccess; ++i) res = rank_collectives(op, i, k, xs[i]);
    ncclResult_t end = r->GroupEnd();
    if (res != ncclSuccess) return rccl_fail(res, "RCCL collective");
    if (end != ncclSuccess) return rccl_fail(end, "ncclGroupEnd");
    return OLAP_OK;
  }
  // direct: all ranks live on one device; each destination reads its peers' partials
  for (size_t i = 0; i < op->ranks.size(); ++i) {
    HIP_TRY(hipSetDevice(c->local[i].device));
    int rc = rank_direct_combine(op, i, k, xs[i]);
    if (rc) return rc;
  }
  return OLAP_OK;
}

extern "C" int olap_shard_drillup_exchange(olap_shard_drillup *op, void *const *streams) {
  if (!op) return fail(OLAP_ERR_INVALID_ARGUMENT, "op is NULL");
  if (op->fused) return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the ranks of this communicator share a device and exchange in one launch (olap_shard_drillup_step)");
  DeviceGuard guard;
  std::vector<hipStream_t> xs(op->ranks.size());
  for (size_t i = 0; i < xs.size(); ++i) xs[i] = streams ? (hipStream_t)streams[i] : nullptr;
  if (op->comm->transport == TRANSPORT_DIRECT && xs.size() > 1) {
    // a destination reads every peer's partial: when the callers' streams differ they must have been joined
    for (size_t i = 1; i < xs.size(); ++i)
      if (xs[i] != xs[0]) return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the direct exchange phase needs one common stream (or use olap_shard_drillup_step)");
  }
  return shard_exchange(op, op->cur, xs);
}

// ---- one step, three ways -----------------------------------------------------------------------------------------------
// (a) ranks share one device (direct transport, few ranks): n local launches and ONE launch that adds the partials,
//     finishes and places every rank's block; no events when the callers' streams are one stream.
static int step_direct_fused(olap_shard_drillup *op, int k, const void *const *in_values, const int32_t *const *in_status,
                             void *const *streams, const StepDest *dest) {
  olap_comm *c = op->comm;
  const size_t nl = op->ranks.size();
  const olap_shard_recipe &rp = op->recipe;
  BufSet &b0 = op->ranks[0].set[k];
  HIP_TRY(hipSetDevice(c->local[0].device));
  bool one_stream = true;
  for (size_t i = 1; i < nl; ++i) one_stream = one_stream && (!streams || streams[i] == streams[0]);
  hipStream_t s0 = streams ? (hipStream_t)streams[0] : nullptr;
  hipStream_t x = one_stream ? s0 : c->local[0].xstream;
  int rc;
  bool locals_done = false;
  if (one_stream && op->same_plans) {
    // every rank's local pass in ONE launch (the same plan over n slabs: blockIdx.y picks the rank)
    if (b0.in_flight) HIP_TRY(hipStreamWaitEvent(s0, b0.xchg_done, 0));
    if (op->mask_primary && !in_status) return fail(OLAP_ERR_INVALID_ARGUMENT, "integer cells over a NaN default: the status mask is required");
    void *outs[kMaxBatchRanks];
    int32_t *couts[kMaxBatchRanks];
    for (size_t i = 0; i < nl; ++i) {
      outs[i] = op->ranks[i].set[k].send[0];
      couts[i] = rp.n_payloads > 1 ? (int32_t *)op->ranks[i].set[k].send[1] : nullptr;
    }
    if ((rc = olap_plan_run_batch(op->ranks[0].local_plan, (int)nl, in_values, in_status, outs, rp.n_payloads > 1 ? couts : nullptr, s0))) return rc;
    locals_done = true;
  }
  for (size_t i = 0; i < nl && !locals_done; ++i) {
    hipStream_t s = streams ? (hipStream_t)streams[i] : nullptr;
    if (b0.in_flight) HIP_TRY(hipStreamWaitEvent(s, b0.xchg_done, 0));  // the set's previous exchange (on a stream of its own) read these partials
    if ((rc = shard_local(op, (int)i, k, in_values[i], in_status ? in_status[i] : nullptr, s))) return rc;
    if (!one_stream) {
      HIP_TRY(hipEventRecord(op->ranks[i].set[k].local_done, s));
      HIP_TRY(hipStreamWaitEvent(x, op->ranks[i].set[k].local_done, 0));
    }
  }
  op->cur = k;
  if (rp.finish == OLAP_FINISH_COMBINE) {
    // the partials lie rank after rank in one block: the same drillUp over the rank axis reads them in place
    for (size_t i = 0; i < nl; ++i) {
      if (dest && !dest[i].values) continue;
      if ((rc = shard_finish(op, (int)i, k, x, dest ? &dest[i] : nullptr))) return rc;
    }
  } else {
    FusedDests d{};
    for (size_t i = 0; i < nl; ++i) {
      uint64_t first = 0, n = 0;
      rank_block(op, c->local[i].rank, &first, &n);
      if (n == 0) continue;
      StepDest want;
      if (dest) {
        want = dest[i];
        if (!want.values || want.count == 0) continue;
        if (want.first < first || want.first + want.count > first + n)
          return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the result block of a rank is not where it was expected");
      } else {
        want.values = op->ranks[i].set[k].result;
        want.status = op->ranks[i].set[k].result_status;
        want.first = first;
        want.count = n;
      }
      d.values[d.n] = want.values;
      d.status[d.n] = want.status;
      d.first[d.n] = want.first;
      d.count[d.n] = want.count;
      ++d.n;
    }
    if (d.n > 0) {
      const int32_t *const *src1 = rp.n_payloads > 1 ? (const int32_t *const *)b0.peers[1] : nullptr;
      const int dn = op->default_kind == OLAP_DEFAULT_NAN;
      const unsigned grid = grid_for_n(op->n_out);
      if (rp.payload_dtype[0] == OLAP_FLOAT64 && op->dtype != OLAP_FLOAT64) {
        SHARD_DISPATCH(op->dtype, hipLaunchKernelGGL((direct_fused_kernel<T, double>), grid, kBlock, 0, x, (const double *const *)b0.peers[0], src1,
                                                     c->world, rp.payload_op[1], op->n_out, rp.finish, dn, d));
      } else {
        SHARD_DISPATCH(op->dtype, hipLaunchKernelGGL((direct_fused_kernel<T, T>), grid, kBlock, 0, x, (const T *const *)b0.peers[0], src1, c->world,
                                                     rp.payload_op[1], op->n_out, rp.finish, dn, d));
      }
      if ((rc = launch_check("direct_fused_kernel"))) return rc;
    }
  }
  if (!one_stream) {
    HIP_TRY(hipEventRecord(b0.xchg_done, x));
    b0.in_flight = true;
    for (size_t i = 0; i < nl; ++i) {
      op->ranks[i].set[k].in_flight = true;
      if (op->depth == 1) HIP_TRY(hipStreamWaitEvent(streams ? (hipStream_t)streams[i] : nullptr, b0.xchg_done, 0));
    }
  }
  return OLAP_OK;
}

// (b) one rank's whole sequence, issued by the thread that owns the rank's device (a worker, or the caller when the
//     process drives a single rank): no device switch; at depth 1 everything rides the caller's stream — launch,
//     collective, launch, no event at all.
static int step_rank_sequence(olap_shard_drillup *op, size_t i, int k, const void *in_values, const int32_t *in_status, hipStream_t s,
                              const StepDest *dest, ShardWorkers *w) {
  olap_comm *c = op->comm;
  BufSet &b = op->ranks[i].set[k];
  const auto &l = c->local[i];
  const bool direct = c->transport == TRANSPORT_DIRECT;
  const bool pipelined = op->depth == 2 || direct;  // a stream of its own for the exchange
  hipStream_t x = pipelined ? l.xstream : s;
  int rc = OLAP_OK;
  hipError_t e = hipSuccess;
  if (b.in_flight && pipelined) e = hipStreamWaitEvent(s, b.xchg_done, 0);
  if (e == hipSuccess) rc = shard_local(op, (int)i, k, in_values, in_status, s);
  if (!rc && e == hipSuccess && pipelined) {
    e = hipEventRecord(b.local_done, s);
    if (e == hipSuccess && !direct) e = hipStreamWaitEvent(x, b.local_done, 0);
  }
  if (direct && w) {
    // a rank of the direct transport reads EVERY peer's partial: all of them must have been recorded before anybody waits
    w->barrier();
    for (size_t j = 0; j < op->ranks.size() && e == hipSuccess; ++j) e = hipStreamWaitEvent(x, op->ranks[j].set[k].local_done, 0);
  }
  if (e != hipSuccess) rc = hip_fail(e, "sharded step (events)");
  if (!rc) {
    if (direct) {
      rc = rank_direct_combine(op, i, k, x);
    } else {
      Rccl *r = rccl();
      ncclResult_t res = r->GroupStart();
      if (res == ncclSuccess) res = rank_collectives(op, i, k, x);
      const ncclResult_t end = r->GroupEnd();
      if (res != ncclSuccess) rc = rccl_fail(res, "RCCL collective");
      else if (end != ncclSuccess) rc = rccl_fail(end, "ncclGroupEnd");
    }
  }
  if (!rc) rc = shard_finish(op, (int)i, k, x, dest);
  if (!rc && pipelined) {
    e = hipEventRecord(b.xchg_done, x);
    b.in_flight = true;
    if (e != hipSuccess) rc = hip_fail(e, "hipEventRecord");
  }
  if (direct && w) {
    // a peer's next local pass overwrites a partial that THIS rank's exchange reads: every caller stream waits for every exchange
    w->barrier();
    for (size_t j = 0; j < op->ranks.size() && !rc; ++j) {
      e = hipStreamWaitEvent(s, op->ranks[j].set[k].xchg_done, 0);
      if (e != hipSuccess) rc = hip_fail(e, "hipStreamWaitEvent");
    }
  } else if (!rc && pipelined && op->depth == 1) {
    e = hipStreamWaitEvent(s, b.xchg_done, 0);
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamWaitEvent");
  }
  return rc;
}

// (c) every local rank from the calling thread: all local passes, ONE group of collectives, all finishes
static int step_inline(olap_shard_drillup *op, int k, const void *const *in_values, const int32_t *const *in_status, void *const *streams,
                       const StepDest *dest) {
  olap_comm *c = op->comm;
  const size_t nl = op->ranks.size();
  std::vector<hipStream_t> xs(nl);
  int rc;
  // 1. local reductions, each on its caller's stream, behind the previous use of this buffer set
  for (size_t i = 0; i < nl; ++i) {
    const auto &l = c->local[i];
    BufSet &b = op->ranks[i].set[k];
    hipStream_t s = streams ? (hipStream_t)streams[i] : nullptr;
    xs[i] = l.xstream;
    HIP_TRY(hipSetDevice(l.device));
    if (b.in_flight) HIP_TRY(hipStreamWaitEvent(s, b.xchg_done, 0));
    if ((rc = shard_local(op, (int)i, k, in_values[i], in_status ? in_status[i] : nullptr, s))) return rc;
    HIP_TRY(hipEventRecord(b.local_done, s));
  }
  // 2. the exchange streams wait for the partials they read (with the direct transport: everybody's)
  for (size_t i = 0; i < nl; ++i) {
    HIP_TRY(hipSetDevice(c->local[i].device));
    if (c->transport == TRANSPORT_DIRECT) {
      for (size_t j = 0; j < nl; ++j) HIP_TRY(hipStreamWaitEvent(xs[i], op->ranks[j].set[k].local_done, 0));
    } else {
      HIP_TRY(hipStreamWaitEvent(xs[i], op->ranks[i].set[k].local_done, 0));
    }
  }
  op->cur = k;
  if ((rc = shard_exchange(op, k, xs))) return rc;
  // 3. finish behind the collective; depth 1: the caller's stream is ordered behind the whole step
  for (size_t i = 0; i < nl; ++i) {
    const auto &l = c->local[i];
    BufSet &b = op->ranks[i].set[k];
    HIP_TRY(hipSetDevice(l.device));
    if ((rc = shard_finish(op, (int)i, k, xs[i], dest ? &dest[i] : nullptr))) return rc;
    HIP_TRY(hipEventRecord(b.xchg_done, xs[i]));
    b.in_flight = true;
  }
  if (c->transport == TRANSPORT_DIRECT) {
    // a peer's next local pass overwrites a partial that THIS rank's exchange reads: every caller stream waits for
    // every exchange of this set before it may reuse it
    for (size_t i = 0; i < nl; ++i) {
      hipStream_t s = streams ? (hipStream_t)streams[i] : nullptr;
      for (size_t j = 0; j < nl; ++j) HIP_TRY(hipStreamWaitEvent(s, op->ranks[j].set[k].xchg_done, 0));
    }
  } else if (op->depth == 1) {
    for (size_t i = 0; i < nl; ++i) {
      HIP_TRY(hipSetDevice(c->local[i].device));
      HIP_TRY(hipStreamWaitEvent(streams ? (hipStream_t)streams[i] : nullptr, op->ranks[i].set[k].xchg_done, 0));
    }
  }
  return OLAP_OK;
}

static int shard_step(olap_shard_drillup *op, const void *const *in_values, const int32_t *const *in_status, void *const *streams,
                      const StepDest *dest) {
  if (!op || !in_values) return fail(OLAP_ERR_INVALID_ARGUMENT, "op/in_values is NULL");
  olap_comm *c = op->comm;
  const int k = op->depth == 2 ? (op->cur ^ 1) : 0;
  DeviceGuard guard;
  if (op->fused) return step_direct_fused(op, k, in_values, in_status, streams, dest);
  if (c->transport == TRANSPORT_DETACHED)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: a detached communicator has no transport; move the payloads yourself (olap_shard_drillup_payload)");
  if (c->workers) {
    op->cur = k;
    const std::function<int(int)> job = [&](int i) -> int {
      return step_rank_sequence(op, (size_t)i, k, in_values[i], in_status ? in_status[i] : nullptr, streams ? (hipStream_t)streams[i] : nullptr,
                                dest ? &dest[i] : nullptr, c->workers);
    };
    return c->workers->run(job);
  }
  if (op->ranks.size() == 1 && c->transport == TRANSPORT_RCCL) {  // one process per GPU: the caller's thread owns the device
    HIP_TRY(hipSetDevice(c->local[0].device));
    op->cur = k;
    return step_rank_sequence(op, 0, k, in_values[0], in_status ? in_status[0] : nullptr, streams ? (hipStream_t)streams[0] : nullptr,
                              dest ? &dest[0] : nullptr, nullptr);
  }
  return step_inline(op, k, in_values, in_status, streams, dest);
}

extern "C" int olap_shard_drillup_step(olap_shard_drillup *op, const void *const *in_values, const int32_t *const *in_status,
                                       void *const *streams) {
  return shard_step(op, in_values, in_status, streams, nullptr);
}

extern "C" int olap_shard_drillup_wait(olap_shard_drillup *op, void *const *streams) {
  if (!op) return fail(OLAP_ERR_INVALID_ARGUMENT, "op is NULL");
  DeviceGuard guard;
  for (size_t i = 0; i < op->ranks.size(); ++i) {
    HIP_TRY(hipSetDevice(op->comm->local[i].device));
    for (int k = 0; k < op->depth; ++k) {
      const BufSet &b = op->fused ? op->ranks[0].set[k] : op->ranks[i].set[k];  // (a fused step records one event for all ranks)
      if (b.in_flight) HIP_TRY(hipStreamWaitEvent(streams ? (hipStream_t)streams[i] : nullptr, b.xchg_done, 0));
    }
  }
  return OLAP_OK;
}

extern "C" int olap_shard_drillup_payload(const olap_shard_drillup *op, int local, int p, void **send, void **recv,
                                          uint64_t *count, int *dtype, int *xchg_op) {
  int rc = check_local(op, local);
  if (rc) return rc;
  if (p < 0 || p >= op->recipe.n_payloads) return fail(OLAP_ERR_INVALID_ARGUMENT, "payload %d outside [0, %d)", p, op->recipe.n_payloads);
  const BufSet &b = op->ranks[local].set[op->cur];
  if (send) *send = b.send[p];
  if (recv) *recv = b.recv[p];
  if (count) *count = op->recipe.payload_op[p] == OLAP_XCHG_GATHER ? op->n_out : op->n_send;
  if (dtype) *dtype = op->recipe.payload_dtype[p];
  if (xchg_op) *xchg_op = op->recipe.payload_op[p];
  return OLAP_OK;
}

extern "C" int olap_shard_drillup_result(const olap_shard_drillup *op, int local, void **values, int32_t **status,
                                         uint64_t *first, uint64_t *count) {
  int rc = check_local(op, local);
  if (rc) return rc;
  const BufSet &b = op->ranks[local].set[op->cur];
  uint64_t f = 0, n = 0;
  rank_block(op, op->comm->local[local].rank, &f, &n);
  void *v = nullptr;
  int32_t *s = nullptr;
  if (separate_result(op->recipe.finish) || op->fused) {
    v = b.result;
    s = b.result_status;
  } else {
    v = b.recv[0];
    s = op->recipe.n_payloads > 1 ? (int32_t *)b.recv[1] : nullptr;  // restored / finished in place
  }
  if (n == 0) v = nullptr, s = nullptr;
  if (values) *values = v;
  if (status) *status = s;
  if (first) *first = f;
  if (count) *count = n;
  return OLAP_OK;
}

// ------------------------------------------------------------------ cache of drillUp steps for the store handles
// Building a step costs plans, table uploads, buffers and events on every device (320 us for two shards of one GPU,
// against a 25 us query); a dashboard repeats the same few roll-ups.  The handle layer therefore keeps its steps in
// an LRU keyed by everything that defines them; olap_comm_destroy drops the entries of its communicator.  Like the
// handles themselves this is for one thread at a time.
namespace {
struct OpCache {
  struct Entry {
    olap_shard_drillup *op;
    uint64_t tick;
    size_t bytes;
  };
  std::mutex mu;
  std::unordered_map<std::string, Entry> map;
  uint64_t tick = 0;
  size_t bytes = 0;
  static constexpr size_t kMaxEntries = 32;
  static constexpr size_t kMaxBytes = (size_t)2 << 30;

  olap_shard_drillup *find(const std::string &key) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = map.find(key);
    if (it == map.end()) return nullptr;
    it->second.tick = ++tick;
    return it->second.op;
  }
  void insert(const std::string &key, olap_shard_drillup *op, size_t op_bytes) {
    std::vector<olap_shard_drillup *> evicted;
    {
      std::lock_guard<std::mutex> lock(mu);
      while (!map.empty() && (map.size() >= kMaxEntries || bytes + op_bytes > kMaxBytes)) {
        auto oldest = map.begin();
        for (auto it = map.begin(); it != map.end(); ++it)
          if (it->second.tick < oldest->second.tick) oldest = it;
        evicted.push_back(oldest->second.op);
        bytes -= oldest->second.bytes;
        map.erase(oldest);
      }
      map[key] = Entry{op, ++tick, op_bytes};
      bytes += op_bytes;
    }
    for (olap_shard_drillup *o : evicted) olap_shard_drillup_destroy(o);
  }
  void drop_comm(const olap_comm *comm) {
    std::vector<olap_shard_drillup *> gone;
    {
      std::lock_guard<std::mutex> lock(mu);
      for (auto it = map.begin(); it != map.end();) {
        if (it->second.op->comm == comm) {
          gone.push_back(it->second.op);
          bytes -= it->second.bytes;
          it = map.erase(it);
        } else {
          ++it;
        }
      }
    }
    for (olap_shard_drillup *o : gone) olap_shard_drillup_destroy(o);
  }
};
OpCache &op_cache() {
  static OpCache *c = new OpCache();  // leaked on purpose, like the pool
  return *c;
}

size_t op_bytes(const olap_shard_drillup *op) {
  size_t total = 0;
  for (int p = 0; p < op->recipe.n_payloads; ++p) total += (size_t)(op->n_send + recv_cells(op, p)) * payload_size(op, p);
  if (separate_result(op->recipe.finish)) total += (size_t)op->n_out * (olap_dtype_size(op->dtype) + 4);
  return total * op->depth * op->ranks.size();
}
}  // namespace

static void drop_cached_steps(const olap_comm *comm) { op_cache().drop_comm(comm); }

// ------------------------------------------------------------------ sharded store handle
struct olap_sharded_store {
  olap_comm *comm = nullptr;
  std::vector<uint32_t> lens, bounds;
  int dtype = 0, default_kind = 0;
  uint64_t inner0 = 1, size = 0;
  std::vector<olap_store *> shard;  // one per local rank
};

static uint64_t slab_cells(const olap_sharded_store *s, int local) {
  const int r = s->comm->local[local].rank;
  return (uint64_t)(s->bounds[r + 1] - s->bounds[r]) * s->inner0;
}
static uint64_t slab_first(const olap_sharded_store *s, int local) {
  return (uint64_t)s->bounds[s->comm->local[local].rank] * s->inner0;
}

// the frame of a sharded store (no shards yet)
static int sharded_frame(olap_sharded_store **out, olap_comm *comm, int ndim, const uint32_t *lens, int dtype, int default_kind,
                         const uint32_t *bounds) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store out-pointer is NULL");
  *out = nullptr;
  if (!comm) return fail(OLAP_ERR_INVALID_ARGUMENT, "comm is NULL");
  int rc;
  if ((rc = check_default(default_kind)) || (rc = check_dtype(dtype))) return rc;
  if (ndim < 1 || ndim > OLAP_MAX_DIMS || !lens) return fail(OLAP_ERR_INVALID_ARGUMENT, "a sharded store needs 1..%d dimensions", OLAP_MAX_DIMS);
  olap_sharded_store *s = new (std::nothrow) olap_sharded_store();
  if (!s) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  s->comm = comm;
  s->lens.assign(lens, lens + ndim);
  s->dtype = dtype;
  s->default_kind = default_kind;
  long double cells = 1;
  for (int d = 0; d < ndim; ++d) cells *= lens[d];
  if (cells > 1.0e12L) {
    delete s;
    return fail(OLAP_ERR_INVALID_ARGUMENT, "cube too large");
  }
  s->inner0 = 1;
  for (int d = 1; d < ndim; ++d) s->inner0 *= lens[d];
  s->size = s->inner0 * lens[0];
  s->bounds.resize(comm->world + 1);
  if (bounds) {
    s->bounds.assign(bounds, bounds + comm->world + 1);
    bool ok = s->bounds[0] == 0 && s->bounds[comm->world] == lens[0];
    for (int r = 0; r < comm->world; ++r) ok = ok && s->bounds[r + 1] >= s->bounds[r];
    if (!ok) {
      delete s;
      return fail(OLAP_ERR_INVALID_ARGUMENT, "bounds must ascend from 0 to the extent of dimension 0");
    }
  } else {
    olap_shard_bounds(lens[0], comm->world, s->bounds.data());
  }
  s->shard.assign(comm->local.size(), nullptr);
  *out = s;
  return OLAP_OK;
}

extern "C" void olap_sharded_store_destroy(olap_sharded_store *s) {
  if (!s) return;
  for (olap_store *sh : s->shard) olap_store_destroy(sh);
  delete s;
}

extern "C" int olap_sharded_store_create(olap_sharded_store **out, olap_comm *comm, int ndim, const uint32_t *lens, int dtype,
                                         int default_kind, const uint32_t *bounds) {
  olap_sharded_store *s = nullptr;
  int rc = sharded_frame(&s, comm, ndim, lens, dtype, default_kind, bounds);
  if (rc) return rc;
  if ((rc = require_device())) {
    delete s;
    return rc;
  }
  DeviceGuard guard;
  for (size_t i = 0; i < s->shard.size() && !rc; ++i) {
    hipError_t e = hipSetDevice(comm->local[i].device);
    if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice");
    else rc = olap_store_create(&s->shard[i], slab_cells(s, (int)i), dtype, default_kind);
  }
  if (rc) {
    olap_sharded_store_destroy(s);
    return rc;
  }
  *out = s;
  return OLAP_OK;
}

extern "C" uint64_t olap_sharded_store_size(const olap_sharded_store *s) { return s ? s->size : 0; }
extern "C" int olap_sharded_store_ndim(const olap_sharded_store *s) { return s ? (int)s->lens.size() : 0; }
extern "C" const uint32_t *olap_sharded_store_lens(const olap_sharded_store *s) { return s ? s->lens.data() : nullptr; }
extern "C" const uint32_t *olap_sharded_store_bounds(const olap_sharded_store *s) { return s ? s->bounds.data() : nullptr; }
extern "C" int olap_sharded_store_reshape(olap_sharded_store *s, int ndim, const uint32_t *lens) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (ndim < 1 || ndim > OLAP_MAX_DIMS || !lens) return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: a sharded store keeps at least its sharded dimension");
  uint64_t inner = 1;
  for (int d = 1; d < ndim; ++d) inner *= lens[d];
  if (lens[0] != s->lens[0] || inner != s->inner0)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the dimensions [%u, ...] do not keep the sharded extent %u in front; gather first", lens[0], s->lens[0]);
  s->lens.assign(lens, lens + ndim);
  return OLAP_OK;
}
extern "C" olap_comm *olap_sharded_store_comm(const olap_sharded_store *s) { return s ? s->comm : nullptr; }
extern "C" olap_store *olap_sharded_store_shard(const olap_sharded_store *s, int local) {
  return (s && local >= 0 && local < (int)s->shard.size()) ? s->shard[local] : nullptr;
}

// runs fn(local index, shard) with the shard's device current.  With an issuing thread per rank (one process, several
// devices) every shard's call goes out from ITS thread at once — a bulk operation that leaves dimension 0 alone is one
// store call per shard, ~10 us of host time each: from one thread eight devices would start up to 70 us apart; fn must
// then only touch its own shard's slots.  in_order: the calls must run one after the other (fn accumulates).
template <typename F>
static int for_each_shard(const olap_sharded_store *s, F fn, bool in_order = false) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->comm->workers && !in_order && s->shard.size() > 1) {
    const std::function<int(int)> job = [&](int i) -> int { return fn(i, s->shard[i]); };
    return s->comm->workers->run(job);
  }
  DeviceGuard guard;
  for (size_t i = 0; i < s->shard.size(); ++i) {
    HIP_TRY(hipSetDevice(s->comm->local[i].device));
    int rc = fn((int)i, s->shard[i]);
    if (rc) return rc;
  }
  return OLAP_OK;
}

extern "C" int olap_sharded_store_fill_seeded(olap_sharded_store *s, uint32_t seed, double frac) {
  return for_each_shard(s, [&](int i, olap_store *sh) -> int {
    const uint64_t n = sh->size;
    if (!n) return OLAP_OK;
    drop_lazy_status(sh);
    const bool fnan = sh->default_kind == OLAP_DEFAULT_NAN && is_float_dtype(sh->dtype);
    int32_t *st = sh->status;
    int32_t *tmp = nullptr;
    if (!st && fnan) {
      HIP_TRY(dev_alloc((void **)&tmp, n * sizeof(int32_t)));
      st = tmp;
    }
    int rc = olap_fill_seeded(sh->values, st, n, slab_first(s, i), sh->dtype, seed, frac, nullptr);
    if (!rc && fnan) {  // the generator leaves 0 in dropped cells; under a NaN default they hold NaN
      SHARD_DISPATCH(sh->dtype, hipLaunchKernelGGL((restore_default_kernel<T>), grid_for_n(n), kBlock, 0, nullptr, (const T *)sh->values, (const int32_t *)st,
                                                   (T *)sh->values, st, n, 1));
      rc = launch_check("restore_default_kernel");
    }
    hipError_t e = hipStreamSynchronize(nullptr);
    if (tmp) dev_free(tmp);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "fill_seeded");
    return rc;
  });
}

extern "C" int olap_sharded_store_set_data_f64(olap_sharded_store *s, const double *host, uint64_t n) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (n != s->size)
    return fail(OLAP_ERR_LENGTH_MISMATCH, "value length is invalid: %llu !== %llu", (unsigned long long)s->size, (unsigned long long)n);
  if (n && !host) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  return for_each_shard(s, [&](int i, olap_store *sh) { return olap_store_set_data_f64(sh, host + slab_first(s, i), sh->size); });
}
extern "C" int olap_sharded_store_get_data_f64(const olap_sharded_store *s, double *host) {
  if (s && s->size && !host) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  return for_each_shard(s, [&](int i, olap_store *sh) { return olap_store_get_data_f64(sh, host + slab_first(s, i)); });
}
extern "C" int olap_sharded_store_get_status(const olap_sharded_store *s, int32_t *host) {
  if (s && s->size && !host) return fail(OLAP_ERR_INVALID_ARGUMENT, "status is NULL");
  return for_each_shard(s, [&](int i, olap_store *sh) { return olap_store_get_status(sh, host + slab_first(s, i)); });
}

// local index of the rank that owns flat cell `index`, or -1
static int owner_of(const olap_sharded_store *s, uint64_t index, uint64_t *local_index) {
  const uint64_t row = s->inner0 ? index / s->inner0 : 0;
  for (size_t i = 0; i < s->shard.size(); ++i) {
    const int r = s->comm->local[i].rank;
    if (row >= s->bounds[r] && row < s->bounds[r + 1]) {
      *local_index = index - (uint64_t)s->bounds[r] * s->inner0;
      return (int)i;
    }
  }
  return -1;
}

extern "C" int olap_sharded_store_get_value(const olap_sharded_store *s, uint64_t index, double *value, int *is_set) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (index >= s->size) {
    if (value) *value = s->default_kind == OLAP_DEFAULT_NAN ? NAN : 0.0;
    if (is_set) *is_set = 0;
    return OLAP_OK;
  }
  uint64_t li = 0;
  const int i = owner_of(s, index, &li);
  if (i < 0) return fail(OLAP_ERR_INDEX_RANGE, "sharded: cell %llu belongs to a rank this process does not drive", (unsigned long long)index);
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->comm->local[i].device));
  return olap_store_get_value(s->shard[i], li, value, is_set);
}
extern "C" int olap_sharded_store_set_value(olap_sharded_store *s, uint64_t index, double value, int is_null) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (index >= s->size) return fail(OLAP_ERR_INDEX_RANGE, "cell index %llu out of bounds [0, %llu[", (unsigned long long)index, (unsigned long long)s->size);
  uint64_t li = 0;
  const int i = owner_of(s, index, &li);
  if (i < 0) return fail(OLAP_ERR_INDEX_RANGE, "sharded: cell %llu belongs to a rank this process does not drive", (unsigned long long)index);
  DeviceGuard guard;
  HIP_TRY(hipSetDevice(s->comm->local[i].device));
  return olap_store_set_value(s->shard[i], li, value, is_null);
}
extern "C" int olap_sharded_store_fill(olap_sharded_store *s, double value) {
  return for_each_shard(s, [&](int, olap_store *sh) { return olap_store_fill(sh, value); });
}
extern "C" int olap_sharded_store_total(const olap_sharded_store *s, double *total) {
  if (!total) return fail(OLAP_ERR_INVALID_ARGUMENT, "total is NULL");
  double acc = 0.0;
  int rc = for_each_shard(s, [&](int, olap_store *sh) {
    double t = 0.0;
    int e = olap_store_total(sh, &t);
    acc += t;  // ranks own ascending index ranges: the reference's order of addition between slabs
    return e;
  }, /*in_order=*/true);
  if (rc) return rc;
  olap_comm *c = s->comm;
  if ((int)c->local.size() != c->world && c->transport == TRANSPORT_RCCL) {
    // one process per GPU: the other ranks' slabs are added over RCCL (every rank calls this, like any collective)
    const auto &l = c->local[0];
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(l.device));
    double *dev = nullptr;
    HIP_TRY(dev_alloc((void **)&dev, sizeof(double)));
    hipError_t e = hipMemcpyAsync(dev, &acc, sizeof(double), hipMemcpyHostToDevice, l.xstream);
    ncclResult_t res = ncclSuccess;
    if (e == hipSuccess) res = rccl()->AllReduce(dev, dev, 1, ncclFloat64, ncclSum, l.nccl, l.xstream);
    if (e == hipSuccess && res == ncclSuccess) e = hipMemcpyAsync(&acc, dev, sizeof(double), hipMemcpyDeviceToHost, l.xstream);
    if (e == hipSuccess && res == ncclSuccess) e = hipStreamSynchronize(l.xstream);
    dev_free(dev);
    if (res != ncclSuccess) return rccl_fail(res, "ncclAllReduce(total)");
    if (e != hipSuccess) return hip_fail(e, "sharded total");
  }
  *total = acc;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_clone(const olap_sharded_store *s, olap_sharded_store **out) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  olap_sharded_store *c = nullptr;
  int rc = sharded_frame(&c, s->comm, (int)s->lens.size(), s->lens.data(), s->dtype, s->default_kind, s->bounds.data());
  if (rc) return rc;
  rc = for_each_shard(s, [&](int i, olap_store *sh) { return olap_store_clone(sh, &c->shard[i]); });
  if (rc) {
    olap_sharded_store_destroy(c);
    return rc;
  }
  *out = c;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_gather(const olap_sharded_store *s, olap_store **out) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  olap_comm *c = s->comm;
  const bool one_process = (int)c->local.size() == c->world;
  if (!one_process && c->transport != TRANSPORT_RCCL)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: gathering across processes needs the RCCL transport");
  DeviceGuard guard;
  // the slabs must be complete before another device (or RCCL's stream) reads them
  for (size_t i = 0; i < s->shard.size(); ++i) {
    HIP_TRY(hipSetDevice(c->local[i].device));
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  HIP_TRY(hipSetDevice(c->local[0].device));
  olap_store *w = nullptr;
  int rc = store_alloc(&w, s->size, s->dtype, s->default_kind);
  if (rc) return rc;
  const size_t es = olap_dtype_size(s->dtype);
  hipError_t e = hipSuccess;
  if (one_process) {
    for (size_t i = 0; i < s->shard.size() && e == hipSuccess; ++i) {
      const olap_store *sh = s->shard[i];
      if (!sh->size) continue;
      e = hipMemcpyAsync((char *)w->values + slab_first(s, (int)i) * es, sh->values, sh->size * es, hipMemcpyDefault, nullptr);
      if (e == hipSuccess && w->status)
        e = hipMemcpyAsync(w->status + slab_first(s, (int)i), sh->status, sh->size * sizeof(int32_t), hipMemcpyDefault, nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  } else {
    Rccl *r = rccl();
    const auto &l = c->local[0];
    ncclResult_t res = r->GroupStart();
    for (int q = 0; q < c->world && res == ncclSuccess; ++q) {
      const uint64_t first = (uint64_t)s->bounds[q] * s->inner0, n = (uint64_t)(s->bounds[q + 1] - s->bounds[q]) * s->inner0;
      if (!n) continue;
      const void *src = q == l.rank ? s->shard[0]->values : nullptr;
      res = r->Broadcast(src ? src : (char *)w->values + first * es, (char *)w->values + first * es, n, nccl_type(s->dtype), q, l.nccl, l.xstream);
      if (res == ncclSuccess && w->status) {
        const void *ssrc = q == l.rank ? (const void *)s->shard[0]->status : nullptr;
        res = r->Broadcast(ssrc ? ssrc : (const void *)(w->status + first), w->status + first, n, ncclInt32, q, l.nccl, l.xstream);
      }
    }
    ncclResult_t end = r->GroupEnd();
    if (res != ncclSuccess || end != ncclSuccess) {
      olap_store_destroy(w);
      return rccl_fail(res != ncclSuccess ? res : end, "ncclBroadcast");
    }
    e = hipStreamSynchronize(l.xstream);
  }
  if (e != hipSuccess) {
    olap_store_destroy(w);
    return hip_fail(e, "sharded gather");
  }
  *out = w;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_scatter(olap_sharded_store **out, olap_comm *comm, const olap_store *whole, int ndim, const uint32_t *lens) {
  if (!whole) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (whole->track_order)  // (first / last of the shards combine in ROW order: the Map's insertion order would be lost silently)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "ordered: this store tracks its insertion order, which a row partition cannot keep; it stays on one device (gather first / do not scatter)");
  olap_sharded_store *s = nullptr;
  int rc = sharded_frame(&s, comm, ndim, lens, whole->dtype, whole->default_kind, nullptr);
  if (rc) return rc;
  if (s->size != whole->size) {
    delete s;
    return fail(OLAP_ERR_LENGTH_MISMATCH, "store holds %llu cells but the dimensions describe %llu", (unsigned long long)whole->size, (unsigned long long)s->size);
  }
  const size_t es = olap_dtype_size(s->dtype);
  {
    DeviceGuard guard;
    hipError_t e = hipSetDevice(whole->device);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // the source must be complete before peers read it
    if (e != hipSuccess) {
      delete s;
      return hip_fail(e, "sharded scatter");
    }
  }
  rc = for_each_shard(s, [&](int i, olap_store *) -> int {
    olap_store *sh = nullptr;
    int e2 = store_alloc(&sh, slab_cells(s, i), s->dtype, s->default_kind);
    if (e2) return e2;
    s->shard[i] = sh;
    if (!sh->size) return OLAP_OK;
    HIP_TRY(hipMemcpyAsync(sh->values, (const char *)whole->values + slab_first(s, i) * es, sh->size * es, hipMemcpyDefault, nullptr));
    if (sh->status) HIP_TRY(hipMemcpyAsync(sh->status, whole->status + slab_first(s, i), sh->size * sizeof(int32_t), hipMemcpyDefault, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return OLAP_OK;
  });
  if (rc) {
    olap_sharded_store_destroy(s);
    return rc;
  }
  *out = s;
  return OLAP_OK;
}

// computed measures over sharded inputs: the element-wise interpreter runs per shard, each into its slab of host_out
extern "C" int olap_sharded_store_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                                               const olap_sharded_store *const *inputs, const double *scalars, int n_scalars,
                                               double *host_out) {
  if (n_inputs <= 0 || !inputs || !inputs[0]) return fail(OLAP_ERR_INVALID_ARGUMENT, "a formula needs at least one stored measure to size its result");
  if (n_inputs > OLAP_FORMULA_MAX_INPUTS) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula: too many inputs");
  const olap_sharded_store *s0 = inputs[0];
  for (int k = 1; k < n_inputs; ++k) {
    if (!inputs[k]) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula input %d is NULL", k);
    if (inputs[k]->comm != s0->comm || inputs[k]->size != s0->size || inputs[k]->bounds != s0->bounds || inputs[k]->inner0 != s0->inner0)
      return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: the formula's inputs are not partitioned alike; gather first");
  }
  if (s0->size && !host_out) return fail(OLAP_ERR_INVALID_ARGUMENT, "out is NULL");
  return for_each_shard(s0, [&](int i, olap_store *sh) -> int {
    if (!sh->size) return OLAP_OK;
    const olap_store *shards[OLAP_FORMULA_MAX_INPUTS];
    for (int k = 0; k < n_inputs; ++k) shards[k] = inputs[k]->shard[i];
    return olap_store_eval_formula(code, n_code, consts, n_consts, n_inputs, shards, scalars, n_scalars, host_out + slab_first(s0, i));
  });
}

static bool identity_map(const uint32_t *m, uint32_t old_len, uint32_t new_len) {
  if (old_len != new_len) return false;
  for (uint32_t k = 0; k < old_len; ++k)
    if (m[k] != k) return false;
  return true;
}

// lens of local rank i's slab, with dimension 0 replaced by its row count
static std::vector<uint32_t> local_lens(const olap_sharded_store *s, int i, const uint32_t *lens) {
  std::vector<uint32_t> v(lens, lens + s->lens.size());
  const int r = s->comm->local[i].rank;
  v[0] = s->bounds[r + 1] - s->bounds[r];
  return v;
}

extern "C" int olap_sharded_store_drillup(const olap_sharded_store *s, olap_sharded_store **out_sharded, olap_store **out_whole,
                                          const uint32_t *new_len, const uint32_t *const *maps, int method) {
  if (!s || !out_sharded || !out_whole) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/out is NULL");
  *out_sharded = nullptr;
  *out_whole = nullptr;
  const int ndim = (int)s->lens.size();
  if (!new_len || !maps) return fail(OLAP_ERR_INVALID_ARGUMENT, "new_len/maps is NULL");
  for (int d = 0; d < ndim; ++d)
    if (s->lens[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
  if (method < OLAP_SUM || method > OLAP_PRODUCT) return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", method);
  if (identity_map(maps[0], s->lens[0], new_len[0])) {
    // the sharded axis is untouched: per shard, no communication
    olap_sharded_store *o = nullptr;
    int rc = sharded_frame(&o, s->comm, ndim, new_len, s->dtype, s->default_kind, s->bounds.data());
    if (rc) return rc;
    rc = for_each_shard(s, [&](int i, olap_store *sh) {
      std::vector<uint32_t> ol = local_lens(s, i, s->lens.data()), nl = local_lens(s, i, new_len);
      std::vector<const uint32_t *> lm(maps, maps + ndim);
      return olap_store_drillup(sh, &o->shard[i], ndim, ol.data(), nl.data(), lm.data(), method);
    });
    if (rc) {
      olap_sharded_store_destroy(o);
      return rc;
    }
    *out_sharded = o;
    return OLAP_OK;
  }
  // the sharded axis is rolled up: partial + one collective.  With at least one new row per rank the result STAYS
  // sharded along the new leading dimension (reduce-scatter of whole rows: rank r keeps rows [r*p, (r+1)*p), p =
  // ceil(G0 / world)); otherwise it is K0/G0 times smaller and arrives whole on the device of local rank 0.
  olap_comm *c = s->comm;
  const bool one_process = (int)c->local.size() == c->world;
  const bool keep_sharded = new_len[0] >= (uint32_t)c->world;
  const int placement = keep_sharded ? OLAP_PLACE_SCATTER_ROWS : one_process ? OLAP_PLACE_ROOT : OLAP_PLACE_ALL;
  std::string key;
  {
    auto raw = [&](const void *ptr, size_t n) { key.append((const char *)ptr, n); };
    const int head[5] = {s->dtype, s->default_kind, method, ndim, placement};
    raw(&c, sizeof c);
    raw(head, sizeof head);
    raw(s->lens.data(), s->lens.size() * sizeof(uint32_t));
    raw(new_len, (size_t)ndim * sizeof(uint32_t));
    raw(s->bounds.data(), s->bounds.size() * sizeof(uint32_t));
    for (int d = 0; d < ndim; ++d) raw(maps[d], (size_t)s->lens[d] * sizeof(uint32_t));
  }
  olap_shard_drillup *op = op_cache().find(key);
  int rc = OLAP_OK;
  if (!op) {
    rc = olap_shard_drillup_create(&op, c, s->dtype, s->default_kind, method, ndim, s->lens.data(), new_len, s->bounds.data(), maps, placement, 1);
    if (rc) return rc;
    op_cache().insert(key, op, op_bytes(op));
  }
  const size_t nl = s->shard.size();
  std::vector<const void *> vals(nl);
  std::vector<const int32_t *> stat(nl);
  for (size_t i = 0; i < nl; ++i) {
    vals[i] = s->shard[i]->values;
    stat[i] = mask_needed(s->shard[i]);
  }
  // The result stores come first: the finishing kernels write straight into them (no copy out of the step's buffers,
  // and the cached step can run again at once).  Pool memory is recycled in null-stream order and every path of the
  // step orders its finishing kernel behind the rank's null stream.
  std::vector<StepDest> dest(nl);
  auto result_store = [&](int i, uint64_t from, uint64_t count, olap_store **dst) -> int {
    DeviceGuard guard;
    HIP_TRY(hipSetDevice(c->local[i].device));
    olap_store *w = nullptr;
    int r2 = store_alloc(&w, count, s->dtype, s->default_kind);
    if (r2) return r2;
    dest[i].values = w->values;
    dest[i].status = w->status;  // (only where the mask is primary: integer cells over a NaN default)
    dest[i].first = from;
    dest[i].count = count;
    *dst = w;
    return OLAP_OK;
  };
  if (!keep_sharded) {
    olap_store *w = nullptr;
    if ((rc = result_store(0, 0, op->n_out, &w))) return rc;
    if ((rc = shard_step(op, vals.data(), stat.data(), nullptr, dest.data()))) {
      olap_store_destroy(w);
      return rc;
    }
    *out_whole = w;
    return OLAP_OK;
  }
  const uint64_t row = op->n_out / new_len[0], per_rows = ((uint64_t)new_len[0] + (uint64_t)c->world - 1) / (uint64_t)c->world;
  std::vector<uint32_t> nb(c->world + 1);
  for (int r = 0; r <= c->world; ++r) nb[r] = (uint32_t)std::min<uint64_t>((uint64_t)r * per_rows, new_len[0]);
  olap_sharded_store *o = nullptr;
  if ((rc = sharded_frame(&o, c, ndim, new_len, s->dtype, s->default_kind, nb.data()))) return rc;
  for (size_t i = 0; i < nl && !rc; ++i) {
    const int r = c->local[i].rank;
    rc = result_store((int)i, (uint64_t)nb[r] * row, (uint64_t)(nb[r + 1] - nb[r]) * row, &o->shard[i]);
  }
  if (!rc) rc = shard_step(op, vals.data(), stat.data(), nullptr, dest.data());
  if (rc) {
    olap_sharded_store_destroy(o);
    return rc;
  }
  *out_sharded = o;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_dice(const olap_sharded_store *s, olap_sharded_store **out, const uint32_t *new_len, const int32_t *const *sel) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/out is NULL");
  *out = nullptr;
  const int ndim = (int)s->lens.size();
  if (!new_len || !sel) return fail(OLAP_ERR_INVALID_ARGUMENT, "new_len/sel is NULL");
  for (int d = 0; d < ndim; ++d)
    if (new_len[d] && !sel[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "sel[%d] is NULL", d);
  std::vector<uint32_t> nb(s->comm->world + 1);
  int rc = olap_shard_dice_bounds(s->bounds.data(), s->comm->world, sel[0], new_len[0], nb.data());
  if (rc) return rc;
  olap_sharded_store *o = nullptr;
  if ((rc = sharded_frame(&o, s->comm, ndim, new_len, s->dtype, s->default_kind, nb.data()))) return rc;
  rc = for_each_shard(s, [&](int i, olap_store *sh) {
    const int r = s->comm->local[i].rank;
    std::vector<uint32_t> ol = local_lens(s, i, s->lens.data()), nl(new_len, new_len + ndim);
    nl[0] = nb[r + 1] - nb[r];
    std::vector<int32_t> rows(sel[0] + nb[r], sel[0] + nb[r + 1]);
    for (auto &x : rows) x -= (int32_t)s->bounds[r];
    std::vector<const int32_t *> ls(sel, sel + ndim);
    ls[0] = rows.data();
    return olap_store_dice(sh, &o->shard[i], ndim, ol.data(), nl.data(), ls.data());
  });
  if (rc) {
    olap_sharded_store_destroy(o);
    return rc;
  }
  *out = o;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_drilldown(const olap_sharded_store *s, olap_sharded_store **out, const uint32_t *new_len,
                                            const uint32_t *const *maps, int method, const double *distributions, uint64_t n_dist) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/out is NULL");
  *out = nullptr;
  const int ndim = (int)s->lens.size();
  if (!new_len || !maps) return fail(OLAP_ERR_INVALID_ARGUMENT, "new_len/maps is NULL");
  for (int d = 0; d < ndim; ++d)
    if (new_len[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
  if (!identity_map(maps[0], new_len[0], s->lens[0]))
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: refining the sharded dimension changes the partition; gather first");
  if (distributions)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: drillDown with distributions indexes weights by the global cell index; gather first");
  (void)n_dist;
  olap_sharded_store *o = nullptr;
  int rc = sharded_frame(&o, s->comm, ndim, new_len, s->dtype, s->default_kind, s->bounds.data());
  if (rc) return rc;
  rc = for_each_shard(s, [&](int i, olap_store *sh) {
    std::vector<uint32_t> ol = local_lens(s, i, s->lens.data()), nl = local_lens(s, i, new_len);
    std::vector<uint32_t> rows(nl[0]);
    for (uint32_t k = 0; k < nl[0]; ++k) rows[k] = k;
    std::vector<const uint32_t *> lm(maps, maps + ndim);
    lm[0] = rows.data();
    return olap_store_drilldown(sh, &o->shard[i], ndim, ol.data(), nl.data(), lm.data(), method, nullptr, 0);
  });
  if (rc) {
    olap_sharded_store_destroy(o);
    return rc;
  }
  *out = o;
  return OLAP_OK;
}

extern "C" int olap_sharded_store_reorder(const olap_sharded_store *s, olap_sharded_store **out, const int32_t *perm) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/out is NULL");
  *out = nullptr;
  const int ndim = (int)s->lens.size();
  if (!perm) return fail(OLAP_ERR_INVALID_ARGUMENT, "perm is NULL");
  if (perm[0] != 0) return fail(OLAP_ERR_INVALID_ARGUMENT, "sharded: moving the sharded dimension needs an all-to-all; gather first");
  std::vector<uint32_t> nl(ndim);
  for (int d = 0; d < ndim; ++d) {
    if (perm[d] < 0 || perm[d] >= ndim) return fail(OLAP_ERR_INVALID_ARGUMENT, "perm[%d] = %d out of range", d, perm[d]);
    nl[d] = s->lens[perm[d]];
  }
  olap_sharded_store *o = nullptr;
  int rc = sharded_frame(&o, s->comm, ndim, nl.data(), s->dtype, s->default_kind, s->bounds.data());
  if (rc) return rc;
  rc = for_each_shard(s, [&](int i, olap_store *sh) {
    std::vector<uint32_t> ol = local_lens(s, i, s->lens.data());
    return olap_store_reorder(sh, &o->shard[i], ndim, ol.data(), perm);
  });
  if (rc) {
    olap_sharded_store_destroy(o);
    return rc;
  }
  *out = o;
  return OLAP_OK;
}
