// olap_capi.hip — host side of libolapgpu: argument validation, launch planning (folding the
// per-dimension index maps of the reference's store methods into kernel views), the C ABI of
// include/olap_hip.h and the store handles used by the Node.js addon.
//
// Reference interface being replaced: InMemoryStore, /root/reference/src/store/in-memory.js
// (drillUp :265-334, drillDown :336-430, dice :213-263, reorder :178-211, load :139-176,
// accessors :8-137).  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "olap_internal.hpp"
#include "olap_kernels.hpp"

using namespace olap;

// ------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  if (e == hipErrorOutOfMemory) return fail(OLAP_ERR_OUT_OF_MEMORY, "%s: %s", what, hipGetErrorString(e));
  return fail(OLAP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// ------------------------------------------------------------------ device memory pool
// hipMalloc / hipFree cost milliseconds for buffers of tens of MB (and hipFree synchronises the
// device), which would dwarf a 70 us kernel: every query allocates its result store.  Freed blocks
// are therefore kept in per-size free lists and handed out again; sizes are rounded up to 1/8-octave
// steps (<= 12.5 % slack).  OLAP_POOL_BYTES caps what may sit idle in the lists (default 32 GiB);
// beyond it blocks go back to the driver.
#include <map>
#include <mutex>
#include <unordered_map>

// OLAP_PLAN_DRY=1: plans are validated and BUILT without a device — their index tables land in host memory
// and olap_plan_run refuses to launch.  A diagnostic (which kernel would a query take?) and the way the
// planning code (CSR, tile cuts, remap and brick tables) runs under AddressSanitizer / UBSan on a machine
// without a GPU (tests/test_capi_nogpu.py, tools/asan_host.sh).  Never a compute path.
static bool plan_dry() {
  static const bool dry = getenv("OLAP_PLAN_DRY") != nullptr;
  return dry;
}

namespace {
struct DevicePool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void *> idle;          // (device, rounded bytes) -> block
  std::unordered_map<void *, std::pair<int, size_t>> live;     // block -> (device, rounded bytes)
  size_t idle_bytes = 0;
  size_t cap = 32ull << 30;
  DevicePool() {
    if (const char *e = getenv("OLAP_POOL_BYTES")) cap = (size_t)strtoull(e, nullptr, 10);
  }
  static size_t round_up(size_t bytes) {
    if (bytes < 256) return 256;
    size_t step = 256;
    while ((step << 4) <= bytes) step <<= 1;  // step = 1/8 .. 1/16 of the size's octave
    return (bytes + step - 1) / step * step;
  }
  hipError_t alloc(void **out, size_t bytes) {
    if (plan_dry()) {  // host memory standing in for a plan's tables (see plan_dry)
      if (posix_memalign(out, 256, round_up(bytes)) != 0) return hipErrorOutOfMemory;
      std::lock_guard<std::mutex> lock(mu);
      live[*out] = {-1, round_up(bytes)};
      return hipSuccess;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t r = round_up(bytes);
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = idle.find({dev, r});
      if (it != idle.end()) {
        *out = it->second;
        idle.erase(it);
        idle_bytes -= r;
        live[*out] = {dev, r};
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(out, r);
    if (e != hipSuccess) {  // give the driver back what we hoard, then retry once
      (void)hipGetLastError();
      trim(0);
      e = hipMalloc(out, r);
    }
    if (e == hipSuccess) {
      std::lock_guard<std::mutex> lock(mu);
      live[*out] = {dev, r};
    }
    return e;
  }
  void release(void *p) {
    if (!p) return;
    std::pair<int, size_t> key;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = live.find(p);
      if (it == live.end()) {  // not ours
        (void)hipFree(p);
        return;
      }
      key = it->second;
      live.erase(it);
      if (key.first < 0) {  // dry-plan host block
        free(p);
        return;
      }
      if (idle_bytes + key.second <= cap) {
        idle.insert({key, p});
        idle_bytes += key.second;
        return;
      }
    }
    (void)hipFree(p);
  }
  void trim(size_t keep) {
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> lock(mu);
      for (auto it = idle.begin(); it != idle.end() && idle_bytes > keep;) {
        drop.push_back(it->second);
        idle_bytes -= it->first.second;
        it = idle.erase(it);
      }
    }
    for (void *q : drop) (void)hipFree(q);
  }
};
DevicePool &pool() {
  static DevicePool *p = new DevicePool();  // leaked on purpose: outlives every handle finalizer
  return *p;
}
}  // namespace

hipError_t dev_alloc(void **out, size_t bytes) { return pool().alloc(out, bytes); }
void dev_free(void *p) { pool().release(p); }

extern "C" const char *olap_last_error(void) { return g_last_error.c_str(); }
extern "C" int olap_abi_version(void) { return OLAP_ABI_VERSION; }

extern "C" int olap_method_from_name(const char *name) {
  if (!name) return OLAP_SUM;  // default parameter, in-memory.js:265
  static const char *names[] = {"sum", "average", "highest", "lowest", "first", "last", "product"};
  for (int i = 0; i < 7; ++i)
    if (!strcmp(name, names[i])) return i;
  return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %s", name);
}

extern "C" int olap_dtype_from_name(const char *name) {
  static const char *names[] = {"int32", "uint32", "float32", "float64"};
  if (name)
    for (int i = 0; i < 4; ++i)
      if (!strcmp(name, names[i])) return i;
  return fail(OLAP_ERR_INVALID_TYPE, "Invalid type");
}

extern "C" size_t olap_dtype_size(int dtype) {
  switch (dtype) {
    case OLAP_INT32:
    case OLAP_UINT32:
    case OLAP_FLOAT32: return 4;
    case OLAP_FLOAT64: return 8;
    default: return 0;
  }
}

int check_dtype(int dtype) {
  if (dtype < OLAP_INT32 || dtype > OLAP_FLOAT64) return fail(OLAP_ERR_INVALID_TYPE, "Invalid type");
  return OLAP_OK;
}
int check_default(int kind) {
  if (kind != OLAP_DEFAULT_ZERO && kind != OLAP_DEFAULT_NAN)
    return fail(OLAP_ERR_INVALID_DEFAULT, "Invalid default value, only NaN and 0 are supported");
  return OLAP_OK;
}

// ------------------------------------------------------------------ device
extern "C" int olap_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int require_device() {
  static thread_local int ok = -1;
  if (ok == 1 || plan_dry()) return OLAP_OK;
  if (olap_device_count() <= 0)
    return fail(OLAP_ERR_NO_DEVICE, "no HIP device available: libolapgpu has no CPU fallback");
  ok = 1;
  return OLAP_OK;
}

extern "C" int olap_set_device(int device) {
  int n = olap_device_count();
  if (n <= 0) return fail(OLAP_ERR_NO_DEVICE, "no HIP device available: libolapgpu has no CPU fallback");
  if (device < 0 || device >= n) return fail(OLAP_ERR_INVALID_ARGUMENT, "device %d out of range [0, %d)", device, n);
  HIP_TRY(hipSetDevice(device));
  return OLAP_OK;
}

extern "C" int olap_device_synchronize(void) {
  int rc = require_device();
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  return OLAP_OK;
}

// ------------------------------------------------------------------ plans
enum PlanKind { PLAN_DRILLUP_AXIS, PLAN_DRILLUP_GENERIC, PLAN_GATHER, PLAN_LOAD, PLAN_DRILLDOWN, PLAN_BRICK, PLAN_GATHER_REDUCE };

struct olap_plan {
  PlanKind kind;
  int dtype = 0, def_nan = 0, his_def_nan = 0, method = 0;
  uint64_t in_cells = 0, out_cells = 0;
  int vec = 1;
  DrillUpAxis axis{};
  DrillUpGeneric gen{};
  Remap remap{};
  DrillDown dd{};
  DrillDownScale dds{};
  Brick brick{};
  uint64_t n_bricks = 0;
  GatherReduce gr{};
  bool xy_ok = false;                      // reorder without a mask: two-axis LDS transpose (olap_transpose.hip; 4- and 8-byte cells)
  TransposeXY xy{};
  DrillUpReduce reduce{};                  // S > 0: reduce regime of the one-axis drillUp
  LoadPermute lperm{};                     // lperm.perm != nullptr: load whose innermost items are permuted, nothing else (load_permute_rows_kernel)
  SegmentedRows seg{};                     // S_tot > 0: wide rows of that regime run as the row kernel over segments + a fold (all rules but product)
  bool dd_two_pass = false;                // float cells, no distributions: scale + broadcast
  bool dice_direct = false;                // dice of one dimension, rows not whole 16-byte groups: dice_direct_kernel
  DiceRows dice_rows{};
  bool dd_rows = false;                    // one refined axis, wide rows: drilldown_rows_kernel (uses `axis`)
  uint32_t dd_longest = 0;                 // children of the largest parent
  void *dev_tab2 = nullptr;                // second table set (two-pass drillDown)
  void *dev_tmp = nullptr;                 // quotients of the two-pass drillDown (old cells)
  std::vector<void *> owned;               // further device allocations freed with the plan
  void *dev_tab = nullptr;                 // index tables
  double *dev_dist = nullptr;              // drillDown distributions
  unsigned long long *dev_err = nullptr;   // drillDown deferred error word
  std::string kernel_name;
  hipStream_t last_stream = nullptr;
  bool ran = false;
  int device = 0;                          // the device the tables and scratch live on (current at creation)
  olap_plan() { (void)hipGetDevice(&device); }
  int cache_refs = 0;                      // handle-layer LRU: users between find() and release() (guarded by the cache lock)
  bool cache_evicted = false;              // evicted while in use: the last release() destroys it
};

static uint64_t product(const uint32_t *v, int n) {
  uint64_t p = 1;
  for (int i = 0; i < n; ++i) p *= v[i];
  return p;
}

static int check_dims(int ndim, const uint32_t *a, const uint32_t *b) {
  if (ndim < 0 || ndim > OLAP_MAX_DIMS) return fail(OLAP_ERR_INVALID_ARGUMENT, "ndim %d out of range [0, %d]", ndim, OLAP_MAX_DIMS);
  if (ndim > 0 && (!a || !b)) return fail(OLAP_ERR_INVALID_ARGUMENT, "dimension length vectors must not be NULL");
  // 2^40 cells (4 TiB of float32) is far beyond one device; it also keeps index math in range
  long double pa = 1, pb = 1;
  for (int d = 0; d < ndim; ++d) {
    pa *= a[d];
    pb *= b[d];
  }
  if (pa > 1.0e12L || pb > 1.0e12L) return fail(OLAP_ERR_INVALID_ARGUMENT, "cube too large");
  return OLAP_OK;
}

static int upload(void **dev, const void *host, size_t bytes) {
  *dev = nullptr;
  if (bytes == 0) bytes = 16;
  HIP_TRY(dev_alloc(dev, bytes));
  if (host && plan_dry()) memcpy(*dev, host, bytes);
  else if (host) HIP_TRY(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
  return OLAP_OK;
}

static int vec_for(int dtype, uint64_t contiguous) {
  const int maxv = dtype == OLAP_FLOAT64 ? 2 : 4;  // 16 B per lane
  for (int v = maxv; v > 1; v >>= 1)
    if (contiguous % (uint64_t)v == 0) return v;
  return 1;
}

extern "C" void olap_plan_destroy(olap_plan *p) {
  if (!p) return;
  // the tables go back to the pool and may be handed out again at once: wait for the launches
  // that still read them (hipFree used to imply this)
  // (on the plan's OWN device: the plan cache may evict a plan of device A while device B is current, and a null
  // last_stream would then name B's null stream)
  if (p->ran && !plan_dry()) {
    DeviceGuard guard;
    if (p->device >= 0) (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->last_stream);
  }
  for (void *q : p->owned) dev_free(q);
  if (p->dev_tab) dev_free(p->dev_tab);
  if (p->dev_tab2) dev_free(p->dev_tab2);
  if (p->dev_tmp) dev_free(p->dev_tmp);
  if (p->dev_dist) dev_free(p->dev_dist);
  if (p->dev_err) dev_free(p->dev_err);
  delete p;
}

extern "C" uint64_t olap_plan_in_cells(const olap_plan *p) { return p ? p->in_cells : 0; }
extern "C" uint64_t olap_plan_out_cells(const olap_plan *p) { return p ? p->out_cells : 0; }
extern "C" const char *olap_plan_kernel_name(const olap_plan *p) { return p ? p->kernel_name.c_str() : ""; }

static bool is_identity_u32(const uint32_t *m, uint32_t old_len, uint32_t new_len) {
  if (old_len != new_len) return false;
  for (uint32_t k = 0; k < old_len; ++k)
    if (m[k] != k) return false;
  return true;
}

// ---- drillUp ------------------------------------------------------------------------------
extern "C" int olap_drillup_plan(olap_plan **out, int dtype, int default_kind, int method, int ndim,
                                 const uint32_t *old_len, const uint32_t *new_len,
                                 const uint32_t *const *maps) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (method < OLAP_SUM || method > OLAP_PARTIAL_AVERAGE)
    return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", method);
  if ((rc = check_dims(ndim, old_len, new_len))) return rc;
  if (ndim > 0 && !maps) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps is NULL");
  std::vector<bool> ident(ndim);
  int n_changed = 0, changed = -1;
  for (int d = 0; d < ndim; ++d) {
    if (old_len[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
    for (uint32_t k = 0; k < old_len[d]; ++k)
      if (maps[d][k] >= new_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillUp map of dimension %d: entry %u = %u is outside the new dimension (%u items)", d, k, maps[d][k], new_len[d]);
    ident[d] = is_identity_u32(maps[d], old_len[d], new_len[d]);
    if (!ident[d]) {
      ++n_changed;
      changed = d;
    }
  }
  if ((rc = require_device())) return rc;

  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->dtype = dtype;
  p->def_nan = default_kind == OLAP_DEFAULT_NAN;
  p->method = method;
  p->in_cells = product(old_len, ndim);
  p->out_cells = product(new_len, ndim);

  // CSR of a K->G map: members of each group in ascending order (counting sort, stable)
  auto build_csr = [](const uint32_t *m, uint32_t K, uint32_t G, std::vector<uint32_t> &gstart, std::vector<uint32_t> &order) {
    gstart.assign((size_t)G + 1, 0);
    for (uint32_t k = 0; k < K; ++k) gstart[m[k] + 1]++;
    for (uint32_t g = 0; g < G; ++g) gstart[g + 1] += gstart[g];
    order.resize(K);
    std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
    for (uint32_t k = 0; k < K; ++k) order[cur[m[k]]++] = k;
  };

  if (n_changed <= 1) {
    // One changed axis (or none: then the whole cube is a single untouched "axis" of length 1).
    p->kind = PLAN_DRILLUP_AXIS;
    DrillUpAxis &a = p->axis;
    std::vector<uint32_t> gstart, order;
    if (n_changed == 1) {
      a.outer = product(old_len, changed);
      a.K = old_len[changed];
      a.G = new_len[changed];
      a.inner = product(old_len + changed + 1, ndim - changed - 1);
      build_csr(maps[changed], old_len[changed], new_len[changed], gstart, order);
    } else {
      a.outer = 1;
      a.K = 1;
      a.G = 1;
      a.inner = p->in_cells;
      gstart = {0, 1};
      order = {0};
    }
    bool contiguous = true;
    for (size_t j = 0; j < order.size(); ++j)
      if (order[j] != j) contiguous = false;
    p->vec = vec_for(dtype, a.inner);
    a.def_nan = p->def_nan;
    a.min_group = 0;
    a.depth = 0;
    std::vector<uint32_t> tab(gstart);
    const size_t order_off = tab.size();
    if (!contiguous) tab.insert(tab.end(), order.begin(), order.end());
    if ((rc = upload(&p->dev_tab, tab.data(), tab.size() * sizeof(uint32_t)))) {
      olap_plan_destroy(p);
      return rc;
    }
    a.gstart = (const uint32_t *)p->dev_tab;
    a.order = contiguous ? nullptr : (const uint32_t *)p->dev_tab + order_off;
    // row-tile regime with interleaved groups (drillup_tile_kernel MODE 3): where every cell of a row goes in
    // the permuted tile
    a.perm_cell = a.perm_grp = nullptr;
    a.perm_pitch = 0;
    {
      const uint64_t budget = kTileBytes / olap_dtype_size(dtype);
      if (!contiguous && a.inner > 0 && a.inner < 4096 && a.K * a.inner <= budget) {
        TilePerm tp;
        tile_perm_build(gstart.data(), order.data(), (uint32_t)a.K, (uint32_t)a.G, (uint32_t)a.inner, budget, 16 / olap_dtype_size(dtype), &tp);
        std::vector<uint32_t> both(tp.cell);
        both.insert(both.end(), tp.grp.begin(), tp.grp.end());
        void *dev_perm = nullptr;
        if ((rc = upload(&dev_perm, both.data(), both.size() * sizeof(uint32_t)))) {
          olap_plan_destroy(p);
          return rc;
        }
        p->owned.push_back(dev_perm);
        a.perm_cell = (const uint32_t *)dev_perm;
        a.perm_grp = (const uint32_t *)dev_perm + tp.cell.size();
        a.perm_pitch = tp.pitch;
      }
    }
    // group-tile regime (drillup_gtile_kernel): contiguous groups, short row pieces, rows too long for
    // the row-tile regime; the group list is cut into tiles of whole groups that fit kTileBytes
    bool gtile_ok = false;
    a.gtile = nullptr;
    a.n_gtile = 0;
    {
      const uint64_t budget = kTileBytes / olap_dtype_size(dtype);
      const uint64_t vcells = 16 / olap_dtype_size(dtype);
      uint64_t gtile_max_slots = 128;
      if (const char *e = getenv("OLAP_GTILE_MAX_SLOTS")) gtile_max_slots = (uint64_t)atoll(e);  // developer knob
      if (contiguous && a.G > 1 && a.inner > 0 && a.inner / (uint64_t)p->vec < gtile_max_slots && a.K * a.inner > budget && !getenv("OLAP_NO_GTILE")) {
        std::vector<uint32_t> cut{0};
        uint64_t cells = 0;
        uint32_t groups = 0;
        bool fits = true;
        for (uint32_t g = 0; g < a.G && fits; ++g) {
          const uint64_t c = (uint64_t)(gstart[g + 1] - gstart[g]) * a.inner;
          if (c > budget - vcells) fits = false;
          if (cells + c > budget - vcells || groups == kGroupTileMaxGroups) {
            cut.push_back(g);
            cells = 0;
            groups = 0;
          }
          cells += c;
          ++groups;
        }
        cut.push_back((uint32_t)a.G);
        // a tile reduces groups-in-tile x inner output cells, one lane each: with a long group or two
        // per tile most of the workgroup idles and the flat form is faster ([3001,3333,10] -> 11 groups
        // of 303 members: 145 us here, 89 us flat) — unless every group is long enough for several lanes to share an
        // output cell (>= 256 members: the kernel's cooperative form)
        uint32_t shortest = ~0u;
        for (uint32_t g = 0; g < a.G; ++g) shortest = std::min(shortest, gstart[g + 1] - gstart[g]);
        const bool coop = shortest >= 256 && !getenv("OLAP_GTILE_NO_COOP");  // (whatever the rule: a plan's tables do not depend on it)
        // (short groups too: a lane's work is its group's members, so a tile with few output cells is cheap when those are
        // few — day -> month over 8-byte cells with the day innermost: 67 months per tile, 31 LDS reads each; 320 us flat)
        uint32_t longest_group = 0;
        for (uint32_t g = 0; g < a.G; ++g) longest_group = std::max(longest_group, gstart[g + 1] - gstart[g]);
        const bool enough_lanes = a.G * a.inner >= 64 * (cut.size() - 1) || coop || longest_group <= 64;
        if (fits && enough_lanes && a.outer * (cut.size() - 1) < 0x7FFFFFFFull) {
          void *dev_cut = nullptr;
          if ((rc = upload(&dev_cut, cut.data(), cut.size() * sizeof(uint32_t)))) {
            olap_plan_destroy(p);
            return rc;
          }
          p->owned.push_back(dev_cut);
          a.gtile = (const uint32_t *)dev_cut;
          a.n_gtile = (uint32_t)(cut.size() - 1);
          a.min_group = coop ? shortest : 0;
          gtile_ok = true;
        }
      }
    }
    // few output cells and long groups: cooperative reduction of S segments per group (workspace
    // in the plan), see DrillUpReduce
    {
      uint32_t longest = 0;
      for (size_t gi = 0; gi + 1 < gstart.size(); ++gi) longest = std::max(longest, gstart[gi + 1] - gstart[gi]);
      const uint64_t cells = a.outer * a.G * a.inner;
      // (a wavefront per 271-cell row was tried for more outputs than this: 260 us against the tile's 103)
      uint64_t reduce_max_cells = 131072;
      if (const char *e = getenv("OLAP_REDUCE_MAX_CELLS")) reduce_max_cells = (uint64_t)atoll(e);  // developer / test knob: who takes small outputs
      if (cells > 0 && cells < reduce_max_cells && longest >= 256) {
        DrillUpReduce &rd = p->reduce;
        uint64_t S;
        // (rows of up to 1 024 four-byte cells too when the 16-byte form below applies — one contiguous '-> all' group
        // whose steps are whole 16-byte groups: a unit then streams its segment as ONE contiguous range, 1-7 rows per
        // step, where the lane-per-cell split form reads 4 KB pieces a row apart: [4e5,250] -> [1,250] 88 us -> see DESIGN K1r)
        // (V = cells per 16 bytes; the smallest whole number of rows that is whole 16-byte groups must fit a unit's lanes)
        const uint64_t V16 = 16 / olap_dtype_size(dtype);
        uint64_t rows_min = 1;
        while ((rows_min * a.inner) % V16 != 0) rows_min *= 2;
        const bool wide16 = a.inner > 128 && rows_min * a.inner <= (uint64_t)kBlock * V16 && contiguous && a.G == 1 && (a.K * a.inner) % V16 == 0 &&
                            !getenv("OLAP_REDUCE_NO_WIDE");
        if (a.inner <= 128 || wide16) {
          const uint64_t groups = a.outer * a.G;
          // Segments per group.  More, shorter segments lose to the per-unit epilogue and to the partials they write
          // and the merge reads back ([1e6,100] -> [1,100] with 4 096 units: 81 us; with 512: 65 us); fewer leave CUs
          // idle.  Workgroup-sized units are few enough to be resident all at once, so what matters is that every CU gets
          // the same number: k per CU for the smallest k in 2..8 that the groups fill to >= 94 % (1 per CU is slower:
          // 84-100 us); with more groups than that the dispatcher balances ~4 K units by itself (tools/reduce_units.py).
          uint64_t by_grid = (4096 + groups - 1) / groups;
          if (const char *e = getenv("OLAP_REDUCE_UNITS")) {
            by_grid = std::max<uint64_t>(1, (uint64_t)atoll(e) / groups);
          } else {
            int cus = 256;
            if (!plan_dry()) {
              int dev = 0, n = 0;
              if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
            }
            for (uint64_t k = 2; k <= 8; ++k) {
              const uint64_t total = (uint64_t)cus * k, s_k = total / groups;
              if (s_k >= 1 && groups * s_k * 100 >= total * 94) {
                by_grid = s_k;
                break;
              }
            }
          }
          const uint64_t by_work = std::max<uint64_t>(1, (uint64_t)longest * a.inner / 8192);  // >= 8 K cells each
          S = std::max<uint64_t>(1, std::min(by_grid, by_work));
          // short segments: a wavefront per segment; long ones: the whole workgroup
          const uint64_t seg_cells = ((uint64_t)longest + S - 1) / S * a.inner;
          rd.unit = (seg_cells <= 16384 && a.inner <= 64) ? 64 : kBlock;
          uint32_t rows = 1;
          while (rows * 2 * a.inner <= (uint64_t)rd.unit) rows *= 2;
          rd.rows = rows;
          // 16 B form: one contiguous '-> all' group, whole rows and segments in multiples of 4 cells
          uint64_t seg_len = ((uint64_t)longest + S - 1) / S;
          seg_len = (seg_len + 3) & ~3ull;
          // rows per step: a power of two whose cells are whole 16-byte groups (4-byte cells: 1 for rows of a multiple of 4
          // cells, 2 for even rows, 4 otherwise; 8-byte cells: 1 for even rows, 2 otherwise), doubled while the unit's lanes hold them
          uint32_t rows4 = (uint32_t)rows_min;
          while ((uint64_t)rows4 * 2 * a.inner <= (uint64_t)rd.unit * V16) rows4 *= 2;
          if (rows4 < V16 && a.inner <= 128) rows4 = (uint32_t)V16;  // (narrow rows: as before)
          if (contiguous && a.G == 1 && (a.K * a.inner) % V16 == 0 && (uint64_t)rows4 * a.inner <= (uint64_t)rd.unit * V16) {
            // as many rows per step as the unit's lanes hold (whole 16-byte groups): 10 rows of 100 cells fill 250 of 256
            // lanes where the power of two below fills 200; the rows are then merged through LDS instead of lane to lane
            uint64_t rows_any = (uint64_t)rd.unit * V16 / a.inner;
            while (rows_any > rows4 && (rows_any * a.inner) % V16 != 0) --rows_any;
            if (a.inner * 2 > V16 && rows_any * 10 >= (uint64_t)rows4 * 11 && !getenv("OLAP_REDUCE_POW2_ROWS")) rows4 = (uint32_t)rows_any;
            rd.vec4 = 1;
            rd.rows = rows4;
            rd.seg_len = (uint32_t)seg_len;
          } else if (contiguous && a.G == 1 && a.inner == 1) {
            // rows at any cell offset: aligned 16-byte groups with masked ends (every lane of the unit busy)
            rd.vec4 = 1;
            rd.edge = 1;
            rd.rows = rd.unit * (uint32_t)V16;
            rd.seg_len = (uint32_t)seg_len;
          }
        } else {
          rd.rows = 0;
          rd.unit = kBlock;
          S = std::max<uint64_t>(1, std::min<uint64_t>((524288 + cells - 1) / cells, longest / 32));
          // 16-byte lanes (drillup_split4_kernel): four output cells per lane; as many segments as give every CU the same
          // number of workgroups (k per CU, smallest k in 2..8 filled to >= 94 %), each of at least 32 members
          if (olap_dtype_size(dtype) == 4 && a.inner % 4 == 0 && !getenv("OLAP_NO_SPLIT4")) {
            rd.vec4 = 1;
            int cus = 256;
            if (!plan_dry()) {
              int dev = 0, n = 0;
              if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
            }
            const uint64_t lanes_per_seg = cells / 4;
            uint64_t s_bal = 0;
            for (uint64_t k = 2; k <= 8 && !s_bal; ++k) {
              const uint64_t total = (uint64_t)cus * k * kBlock, s_k = total / lanes_per_seg;
              if (s_k >= 1 && lanes_per_seg * s_k * 100 >= total * 94) s_bal = s_k;
            }
            if (!s_bal) s_bal = std::max<uint64_t>(1, (uint64_t)cus * 8 * kBlock / lanes_per_seg);
            if (const char *e = getenv("OLAP_SPLIT_K")) s_bal = std::max<uint64_t>(1, (uint64_t)cus * (uint64_t)atoll(e) * kBlock / lanes_per_seg);  // developer knob: workgroups per CU
            S = std::max<uint64_t>(1, std::min<uint64_t>(s_bal, longest / 32));
          }
        }
        if (rd.rows == 0 && !plan_dry() && !getenv("OLAP_NO_SEGMENTED_ROWS") && a.inner / (uint64_t)p->vec >= 128 && a.inner * olap_dtype_size(dtype) >= 2048) {
          // Wide rows (beyond the cooperative forms): the row kernel over SEGMENTS of the groups + a fold (see
          // SegmentedRows) — as many segments as give the chip ~8 workgroups per CU, at least 16 members each.
          int cus = 256;
          {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
          }
          const uint64_t n_vec = a.inner / (uint64_t)p->vec, bpr = (n_vec + kBlock - 1) / kBlock;
          // (two workgroups per CU with four rows in flight per lane: [1e4,1e4] -> [1,1e4] 62.9 us; eight per CU with one
          // row in flight, as the plain row regime runs: 75-77 us — four times the partials to write and to fold)
          uint64_t per_cu = 2;
          if (const char *e = getenv("OLAP_SEG_WG_PER_CU")) per_cu = std::max<uint64_t>(1, (uint64_t)atoll(e));  // developer knob
          const uint64_t want = std::max<uint64_t>(1, (uint64_t)cus * per_cu / std::max<uint64_t>(1, a.outer * bpr));
          const uint64_t seg_len = std::max<uint64_t>(16, (a.K + want - 1) / want);
          std::vector<uint32_t> sg_gstart{0}, sg_first{0};
          for (uint64_t gi = 0; gi < a.G; ++gi) {
            for (uint64_t j = gstart[gi]; j < gstart[gi + 1]; j += seg_len) sg_gstart.push_back((uint32_t)std::min<uint64_t>(j + seg_len, gstart[gi + 1]));
            sg_first.push_back((uint32_t)sg_gstart.size() - 1);
          }
          const uint64_t s_tot = sg_gstart.size() - 1;
          void *d_g = nullptr, *d_f = nullptr, *d_p = nullptr, *d_a = nullptr;
          if (s_tot > a.G && a.outer * s_tot * bpr < 0x7FFFFFFFull && !(rc = upload(&d_g, sg_gstart.data(), sg_gstart.size() * 4)) &&
              !(rc = upload(&d_f, sg_first.data(), sg_first.size() * 4))) {
            hipError_t e1 = dev_alloc(&d_p, a.outer * s_tot * a.inner * sizeof(double));
            if (e1 == hipSuccess) e1 = dev_alloc(&d_a, a.outer * s_tot * a.inner * sizeof(int32_t));
            if (e1 == hipSuccess) {
              p->seg.S_tot = (uint32_t)s_tot;
              p->seg.gstart = (const uint32_t *)d_g;
              p->seg.seg_start = (const uint32_t *)d_f;
              p->seg.partial = d_p;
              p->seg.aux = (int32_t *)d_a;
            } else {
              (void)hipGetLastError();  // no room for the partials: the split forms below need less
            }
          }
          for (void *q : {d_g, d_f, d_p, d_a})
            if (q) p->owned.push_back(q);
          if (rc) {
            olap_plan_destroy(p);
            return rc;
          }
        }
        rd.S = (uint32_t)S;
        if (!rd.vec4 || rd.rows == 0) rd.seg_len = (uint32_t)((longest + S - 1) / S);
        hipError_t e = dev_alloc(&p->dev_tmp, cells * S * sizeof(Partial));
        if (e != hipSuccess) {
          olap_plan_destroy(p);
          return hip_fail(e, "hipMalloc(drillUp reduce workspace)");
        }
        rd.part = (Partial *)p->dev_tmp;
      }
    }
    if (p->seg.S_tot > 0 && method != OLAP_PRODUCT && method != OLAP_PARTIAL_AVERAGE)
      p->kernel_name = "drillup_rows_kernel (segments)+segments_combine_kernel";
    else if (p->reduce.S > 0)
      p->kernel_name = p->reduce.rows == 0 ? (p->reduce.vec4 ? "drillup_split4_kernel+drillup_merge_kernel" : "drillup_split_kernel+drillup_merge_kernel")
                       : p->reduce.vec4    ? (p->reduce.S == 1 ? "drillup_reduce4_kernel" : "drillup_reduce4_kernel+drillup_merge_kernel")
                                           : "drillup_reduce_kernel+drillup_merge_kernel";
    else if (gtile_ok) p->kernel_name = "drillup_gtile_kernel";
    else if (a.inner / (uint64_t)p->vec >= 128) p->kernel_name = "drillup_rows_kernel";
    else if (a.inner < 128 && a.K * a.inner <= (16 * 1024) / olap_dtype_size(dtype) && a.K * a.inner > 0 &&
             (a.perm_cell && !getenv("OLAP_TILE_NO_PERMUTE") ? 2 * a.G : a.G + 1 + (contiguous ? 0 : a.K)) * 4 <= 16 * 1024)
      p->kernel_name = "drillup_tile_kernel";  // (the launcher re-checks alignment; may still pick the flat form)
    else p->kernel_name = "drillup_flat_kernel";
  } else {
    p->kind = PLAN_DRILLUP_GENERIC;
    DrillUpGeneric &g = p->gen;
    std::vector<uint32_t> tab;
    std::vector<uint64_t> old_stride(ndim);
    {
      uint64_t s = 1;
      for (int d = ndim - 1; d >= 0; --d) {
        old_stride[d] = s;
        s *= old_len[d];
      }
    }
    int nd = 0;
    for (int d = 0; d < ndim; ++d) {
      if (ident[d] && nd > 0 && g.csr[nd - 1] < 0 && ident[d - 1]) {
        // merge with the previous identity dim
        g.new_len[nd - 1] *= new_len[d];
        g.old_stride[nd - 1] = old_stride[d];
        continue;
      }
      if (nd == kMaxDims) {
        olap_plan_destroy(p);
        return fail(OLAP_ERR_INVALID_ARGUMENT, "drillUp: more than %d non-mergeable dimensions", kMaxDims);
      }
      g.new_len[nd] = new_len[d];
      g.old_stride[nd] = old_stride[d];
      if (ident[d]) {
        g.csr[nd] = -1;
        g.ord[nd] = -1;
      } else {
        std::vector<uint32_t> gstart, order;
        build_csr(maps[d], old_len[d], new_len[d], gstart, order);
        const uint32_t ord_base = (uint32_t)(tab.size() + gstart.size());
        g.csr[nd] = (int32_t)tab.size();
        g.ord[nd] = 0;  // gstart entries index straight into `tab` (absolute)
        for (auto &x : gstart) x += ord_base;
        tab.insert(tab.end(), gstart.begin(), gstart.end());
        tab.insert(tab.end(), order.begin(), order.end());
      }
      ++nd;
    }
    g.nd = nd;
    g.total = p->out_cells;
    g.def_nan = p->def_nan;
    if ((rc = upload(&p->dev_tab, tab.data(), tab.size() * sizeof(uint32_t)))) {
      olap_plan_destroy(p);
      return rc;
    }
    g.tab = (const uint32_t *)p->dev_tab;
    p->kernel_name = "drillup_generic_kernel";
  }
  *out = p;
  return OLAP_OK;
}

// ---- shared: build a Remap over destination dims -------------------------------------------
struct RemapDim {
  uint32_t len;
  bool arithmetic;
  uint64_t stride;              // arithmetic dims
  std::vector<int64_t> table;   // table dims: offset or -1
};

static int finish_remap(olap_plan *p, std::vector<RemapDim> &dims, uint64_t iter_cells, bool allow_vec) {
  // merge adjacent arithmetic dims whose strides nest (outer.stride == inner.stride * inner.len)
  std::vector<RemapDim> m;
  for (auto &d : dims) {
    if (!m.empty() && m.back().arithmetic && d.arithmetic && m.back().stride == d.stride * d.len &&
        (uint64_t)m.back().len * d.len <= 0xFFFFFFFFull) {
      m.back().len *= d.len;
      m.back().stride = d.stride;
    } else {
      m.push_back(d);
    }
  }
  if ((int)m.size() > kMaxDims) return fail(OLAP_ERR_INVALID_ARGUMENT, "more than %d non-mergeable dimensions", kMaxDims);
  Remap &r = p->remap;
  std::vector<int64_t> tab;
  r.nd = (int)m.size();
  for (int d = 0; d < r.nd; ++d) {
    r.len[d] = m[d].len;
    r.stride[d] = m[d].stride;
    if (m[d].arithmetic) {
      r.tab_off[d] = -1;
    } else {
      r.tab_off[d] = (int32_t)tab.size();
      tab.insert(tab.end(), m[d].table.begin(), m[d].table.end());
    }
  }
  p->vec = 1;
  if (allow_vec && r.nd > 0 && m.back().arithmetic && m.back().stride == 1) p->vec = vec_for(p->dtype, m.back().len);
  if (r.nd == 0) p->vec = 1;
  r.total = iter_cells / (uint64_t)p->vec;
  r.def_nan = p->def_nan;
  r.src_def_nan = p->his_def_nan;
  int rc = upload(&p->dev_tab, tab.data(), tab.size() * sizeof(int64_t));
  if (rc) return rc;
  r.tab = (const int64_t *)p->dev_tab;
  return OLAP_OK;
}

// ---- dice -----------------------------------------------------------------------------------
extern "C" int olap_dice_plan(olap_plan **out, int dtype, int default_kind, int ndim,
                              const uint32_t *old_len, const uint32_t *new_len,
                              const int32_t *const *sel) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if ((rc = check_dims(ndim, old_len, new_len))) return rc;
  if (ndim > 0 && !sel) return fail(OLAP_ERR_INVALID_ARGUMENT, "sel is NULL");
  for (int d = 0; d < ndim; ++d) {
    if (new_len[d] && !sel[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "sel[%d] is NULL", d);
    for (uint32_t j = 0; j < new_len[d]; ++j)
      if (sel[d][j] >= 0 && (uint32_t)sel[d][j] >= old_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "dice selection of dimension %d: entry %u = %d is outside the old dimension (%u items)", d, j, sel[d][j], old_len[d]);
  }
  if ((rc = require_device())) return rc;
  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->kind = PLAN_GATHER;
  p->dtype = dtype;
  p->def_nan = default_kind == OLAP_DEFAULT_NAN;
  p->in_cells = product(old_len, ndim);
  p->out_cells = product(new_len, ndim);
  std::vector<RemapDim> dims(ndim);
  std::vector<char> untouched(ndim > 0 ? ndim : 1, 1);
  uint64_t stride = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    RemapDim &rd = dims[d];
    rd.len = new_len[d];
    rd.stride = stride;
    bool ident = old_len[d] == new_len[d];
    for (uint32_t j = 0; ident && j < new_len[d]; ++j) ident = sel[d][j] == (int32_t)j;
    rd.arithmetic = ident;
    untouched[d] = ident;
    if (!ident) {
      // Map(oldIdx -> newIdx) built left to right (in-memory.js:219-224): of two new items naming
      // the same old item only the LAST receives the cells
      rd.table.assign(new_len[d], -1);
      std::vector<int64_t> last(old_len[d] ? old_len[d] : 1, -1);
      for (uint32_t j = 0; j < new_len[d]; ++j)
        if (sel[d][j] >= 0) last[sel[d][j]] = j;
      for (uint32_t j = 0; j < new_len[d]; ++j)
        if (sel[d][j] >= 0 && last[sel[d][j]] == (int64_t)j) rd.table[j] = (int64_t)sel[d][j] * (int64_t)stride;
    }
    stride *= old_len[d];
  }
  if ((rc = finish_remap(p, dims, p->out_cells, true))) {
    olap_plan_destroy(p);
    return rc;
  }
  p->kernel_name = "gather(dice)";
  // One diced dimension over rows that are not whole 16-byte groups (cubes with odd extents): the gather would move
  // 4 bytes per lane; dice_direct_kernel writes aligned 16-byte groups of the (contiguous) result and fetches each
  // with one cell-aligned 16-byte load.  An unknown item, or an old item that a later new item names again, fills
  // its row with the default (the effective selection below).
  {
    int axis = -1, touched = 0;
    for (int d = 0; d < ndim; ++d)
      if (!untouched[d]) {
        axis = d;
        ++touched;
      }
    const uint64_t vf = 16 / olap_dtype_size(dtype);
    if (touched == 1 && p->out_cells > 0 && !getenv("OLAP_DICE_NO_DIRECT")) {
      uint64_t outer = 1, inner = 1;
      for (int d = 0; d < axis; ++d) outer *= old_len[d];
      for (int d = axis + 1; d < ndim; ++d) inner *= old_len[d];
      const uint32_t G = old_len[axis], K = new_len[axis];
      DiceRows rows{};
      rows.outer = outer;
      rows.k_old = G;
      rows.k_new = K;
      rows.inner = inner;
      rows.def_nan = p->def_nan;
      const bool fits = olap_dtype_size(dtype) == 8 ? dice_direct_fits<double>(rows, nullptr) : dice_direct_fits<float>(rows, nullptr);
      if (inner % vf != 0 && K > 0 && fits) {
        std::vector<int32_t> eff(K, -1);
        std::vector<int64_t> last(G ? G : 1, -1);
        for (uint32_t j = 0; j < K; ++j)
          if (sel[axis][j] >= 0) last[sel[axis][j]] = j;
        for (uint32_t j = 0; j < K; ++j)
          if (sel[axis][j] >= 0 && last[sel[axis][j]] == (int64_t)j) eff[j] = sel[axis][j];
        void *dev = nullptr;
        if ((rc = upload(&dev, eff.data(), eff.size() * sizeof(int32_t)))) {
          olap_plan_destroy(p);
          return rc;
        }
        p->owned.push_back(dev);
        rows.sel = (const int32_t *)dev;
        p->dice_rows = rows;
        p->dice_direct = true;
        p->kernel_name = "dice_direct_kernel";
      }
    }
  }
  *out = p;
  return OLAP_OK;
}

// ---- fused dice -> drillUp --------------------------------------------------------------------
extern "C" int olap_dice_drillup_plan(olap_plan **out, int dtype, int default_kind, int method, int ndim,
                                      const uint32_t *old_len, const uint32_t *mid_len, const uint32_t *new_len,
                                      const int32_t *const *sel, const uint32_t *const *maps) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (method < OLAP_SUM || method > OLAP_PARTIAL_AVERAGE)
    return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", method);
  if ((rc = check_dims(ndim, old_len, mid_len)) || (rc = check_dims(ndim, mid_len, new_len))) return rc;
  if (ndim > 0 && (!sel || !maps)) return fail(OLAP_ERR_INVALID_ARGUMENT, "sel/maps is NULL");
  int axis = -1, n_changed = 0;
  for (int d = 0; d < ndim; ++d) {
    if (mid_len[d] && (!sel[d] || !maps[d])) return fail(OLAP_ERR_INVALID_ARGUMENT, "sel[%d]/maps[%d] is NULL", d, d);
    for (uint32_t j = 0; j < mid_len[d]; ++j) {
      if (sel[d][j] >= 0 && (uint32_t)sel[d][j] >= old_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "dice selection of dimension %d: entry %u = %d is outside the old dimension (%u items)", d, j, sel[d][j], old_len[d]);
      if (maps[d][j] >= new_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillUp map of dimension %d: entry %u = %u is outside the new dimension (%u items)", d, j, maps[d][j], new_len[d]);
    }
    if (!is_identity_u32(maps[d], mid_len[d], new_len[d])) {
      ++n_changed;
      axis = d;
    }
  }
  if (n_changed > 1)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "fused dice+drillUp takes one rolled-up dimension (%d given): run the two plans", n_changed);
  if (n_changed == 0) {
    if (method != OLAP_AVERAGE && method != OLAP_PARTIAL_AVERAGE)  // a plain dice (average of one cell = the cell too,
      return olap_dice_plan(out, dtype, default_kind, ndim, old_len, mid_len, sel);  // but keep its count semantics below)
    axis = ndim - 1;
    if (ndim == 0) return olap_dice_plan(out, dtype, default_kind, ndim, old_len, mid_len, sel);
  }
  if ((rc = require_device())) return rc;
  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->kind = PLAN_GATHER_REDUCE;
  p->dtype = dtype;
  p->def_nan = default_kind == OLAP_DEFAULT_NAN;
  p->method = method;
  p->in_cells = product(old_len, ndim);
  p->out_cells = product(new_len, ndim);
  std::vector<uint64_t> old_stride(ndim);
  {
    uint64_t s = 1;
    for (int d = ndim - 1; d >= 0; --d) {
      old_stride[d] = s;
      s *= old_len[d];
    }
  }
  // duplicates in a selection: only the LAST new item naming an old item receives its cells (:219-224)
  auto effective_sel = [&](int d) {
    std::vector<int64_t> e(mid_len[d], -1);
    std::vector<int64_t> last(old_len[d] ? old_len[d] : 1, -1);
    for (uint32_t j = 0; j < mid_len[d]; ++j)
      if (sel[d][j] >= 0) last[sel[d][j]] = j;
    for (uint32_t j = 0; j < mid_len[d]; ++j)
      if (sel[d][j] >= 0 && last[sel[d][j]] == (int64_t)j) e[j] = sel[d][j];
    return e;
  };
  // output dims (collapsed): untouched neighbours merge; the rolled-up axis stays alone
  std::vector<RemapDim> dims;
  std::vector<int> dim_of;  // original dim of each collapsed dim's LAST member
  int axis_c = -1;
  for (int d = 0; d < ndim; ++d) {
    RemapDim rd;
    rd.len = new_len[d];
    rd.stride = old_stride[d];
    if (d == axis) {
      rd.arithmetic = false;  // placeholder: handled through gstart/member_off
      axis_c = (int)dims.size();
      dims.push_back(rd);
      continue;
    }
    const std::vector<int64_t> e = effective_sel(d);
    bool ident = old_len[d] == mid_len[d];
    for (uint32_t j = 0; ident && j < mid_len[d]; ++j) ident = e[j] == (int64_t)j;
    rd.arithmetic = ident;
    if (!ident) {
      rd.table.resize(mid_len[d]);
      for (uint32_t j = 0; j < mid_len[d]; ++j) rd.table[j] = e[j] < 0 ? -1 : e[j] * (int64_t)old_stride[d];
    }
    const bool prev_mergeable = !dims.empty() && (int)dims.size() - 1 != axis_c && dims.back().arithmetic && rd.arithmetic &&
                                dims.back().stride == rd.stride * rd.len && (uint64_t)dims.back().len * rd.len <= 0xFFFFFFFFull;
    if (prev_mergeable) {
      dims.back().len *= rd.len;
      dims.back().stride = rd.stride;
    } else {
      dims.push_back(rd);
    }
  }
  if ((int)dims.size() > kMaxDims) {
    olap_plan_destroy(p);
    return fail(OLAP_ERR_INVALID_ARGUMENT, "more than %d non-mergeable dimensions", kMaxDims);
  }
  GatherReduce &g = p->gr;
  Remap &r = g.r;
  std::vector<int64_t> tab;
  r.nd = (int)dims.size();
  for (int d = 0; d < r.nd; ++d) {
    r.len[d] = dims[d].len;
    r.stride[d] = dims[d].stride;
    if (d == axis_c || dims[d].arithmetic) {
      r.tab_off[d] = -1;
    } else {
      r.tab_off[d] = (int32_t)tab.size();
      tab.insert(tab.end(), dims[d].table.begin(), dims[d].table.end());
    }
  }
  // members of each group: mid indices ascending, mapped through the selection to source offsets;
  // rows the selection cannot find contribute nothing
  const uint32_t G = new_len[axis], K = mid_len[axis];
  const std::vector<int64_t> e_axis = effective_sel(axis);
  std::vector<uint32_t> gstart((size_t)G + 1, 0);
  for (uint32_t k = 0; k < K; ++k)
    if (e_axis[k] >= 0) gstart[maps[axis][k] + 1]++;
  for (uint32_t gg = 0; gg < G; ++gg) gstart[gg + 1] += gstart[gg];
  std::vector<int64_t> member(gstart[G] ? gstart[G] : 1, 0);
  {
    std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
    for (uint32_t k = 0; k < K; ++k)
      if (e_axis[k] >= 0) member[cur[maps[axis][k]]++] = e_axis[k] * (int64_t)old_stride[axis];
  }
  const size_t member_at = tab.size();
  tab.insert(tab.end(), member.begin(), member.end());
  g.axis = axis_c;
  p->vec = 1;
  if (r.nd > 0 && axis_c != r.nd - 1 && dims.back().arithmetic && dims.back().stride == 1) p->vec = vec_for(dtype, dims.back().len);
  r.total = p->out_cells / (uint64_t)p->vec;
  r.def_nan = p->def_nan;
  r.src_def_nan = p->def_nan;
  if ((rc = upload(&p->dev_tab, tab.data(), tab.size() * sizeof(int64_t)))) {
    olap_plan_destroy(p);
    return rc;
  }
  r.tab = (const int64_t *)p->dev_tab;
  g.member_off = r.tab + member_at;
  if ((rc = upload(&p->dev_tab2, gstart.data(), gstart.size() * sizeof(uint32_t)))) {
    olap_plan_destroy(p);
    return rc;
  }
  g.gstart = (const uint32_t *)p->dev_tab2;
  p->kernel_name = "gather_reduce_kernel";
  *out = p;
  return OLAP_OK;
}

// ---- reorder --------------------------------------------------------------------------------
extern "C" int olap_reorder_plan(olap_plan **out, int dtype, int default_kind, int ndim,
                                 const uint32_t *old_len, const int32_t *perm) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if ((rc = check_dims(ndim, old_len, old_len))) return rc;
  if (ndim > 0 && !perm) return fail(OLAP_ERR_INVALID_ARGUMENT, "perm is NULL");
  std::vector<bool> seen(ndim, false);
  for (int d = 0; d < ndim; ++d) {
    if (perm[d] < 0 || perm[d] >= ndim || seen[perm[d]])
      return fail(OLAP_ERR_INVALID_ARGUMENT, "reorder: perm is not a permutation of the dimensions");
    seen[perm[d]] = true;
  }
  if ((rc = require_device())) return rc;
  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->kind = PLAN_GATHER;
  p->dtype = dtype;
  p->def_nan = default_kind == OLAP_DEFAULT_NAN;
  p->in_cells = p->out_cells = product(old_len, ndim);
  std::vector<uint64_t> old_stride(ndim);
  uint64_t s = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    old_stride[d] = s;
    s *= old_len[d];
  }
  std::vector<RemapDim> dims(ndim);
  for (int d = 0; d < ndim; ++d) {
    dims[d].len = old_len[perm[d]];
    dims[d].arithmetic = true;
    dims[d].stride = old_stride[perm[d]];
  }
  // merge new dims that are adjacent in the source too, and drop extent-1 dims
  std::vector<RemapDim> m;
  for (auto &d : dims) {
    if (d.len == 1 && ndim > 1) continue;
    if (!m.empty() && m.back().stride == d.stride * d.len && (uint64_t)m.back().len * d.len <= 0xFFFFFFFFull) {
      m.back().len *= d.len;
      m.back().stride = d.stride;
    } else {
      m.push_back(d);
    }
  }
  const bool contiguous_tail = m.empty() || m.back().stride == 1;
  if (!contiguous_tail && p->out_cells > 0 && !getenv("OLAP_NO_XY")) {
    // two-axis transpose: X = the source's fastest dimensions, Y = the destination's fastest ones (olap_transpose.hip)
    const int n = (int)m.size();
    std::vector<uint64_t> ostr(n);
    uint64_t st = 1;
    for (int d = n - 1; d >= 0; --d) {
      ostr[d] = st;
      st *= m[d].len;
    }
    std::vector<int> by_in(n), by_out(n);
    for (int d = 0; d < n; ++d) by_in[d] = by_out[d] = d;
    std::sort(by_in.begin(), by_in.end(), [&](int x, int y) { return m[x].stride < m[y].stride; });
    std::sort(by_out.begin(), by_out.end(), [&](int x, int y) { return ostr[x] < ostr[y]; });
    TransposeXY &t = p->xy;
    std::vector<char> taken(n, 0);
    t.lx = t.ly = 1;
    for (int d : by_in) {
      if (t.lx >= 128 || t.nx == kTransposeMaxAxis || d == by_out[0]) break;  // the destination's fastest dimension is Y's
      t.len_x[t.nx] = m[d].len;
      t.out_stride_x[t.nx] = ostr[d];
      ++t.nx;
      t.lx *= m[d].len;
      taken[d] = 1;
    }
    for (int d : by_out) {
      if (t.ly >= 128 || t.ny == kTransposeMaxAxis || taken[d]) break;
      t.len_y[t.ny] = m[d].len;
      t.in_stride_y[t.ny] = m[d].stride;
      ++t.ny;
      t.ly *= m[d].len;
      taken[d] = 2;
    }
    bool ok = t.lx >= 16 && t.ly >= 16 && t.lx < 0x7FFFFFFFull && t.ly < 0x7FFFFFFFull;
    t.batch = 1;
    bool in4 = true, out4 = true;  // every batch / cross stride a multiple of 4 cells: rows start 16-byte aligned
    for (int d = 0; d < n && ok; ++d) {
      if (taken[d]) continue;
      if (t.nb == kTransposeMaxBatch) {
        ok = false;
        break;
      }
      t.len_b[t.nb] = m[d].len;
      t.in_stride_b[t.nb] = m[d].stride;
      t.out_stride_b[t.nb] = ostr[d];
      ++t.nb;
      t.batch *= m[d].len;
      in4 = in4 && m[d].stride % 4 == 0;
      out4 = out4 && ostr[d] % 4 == 0;
    }
    for (int k = 0; k < t.ny; ++k) in4 = in4 && t.in_stride_y[k] % 4 == 0;
    for (int k = 0; k < t.nx; ++k) out4 = out4 && t.out_stride_x[k] % 4 == 0;
    if (ok) {
      // measured (tools/pmc_probe.py): long runs matter more on the write side
      const bool wide_cells = olap_dtype_size(dtype) == 8;  // (64 x 64 tiles of 8-byte cells: the same 512-byte runs, 33 KB of LDS)
      t.ty = t.ly >= 96 && !wide_cells ? 128 : 64;
      t.tx = 64;
      if (const char *e = getenv("OLAP_XY_TILE")) {  // developer knob: "64x64" | "128x64" | "64x128"
        int a = 0, b2 = 0;
        if (sscanf(e, "%dx%d", &a, &b2) == 2 && (a == 64 || a == 128) && (b2 == 64 || b2 == 128) && !(wide_cells && a == 128 && b2 == 128)) {
          t.tx = a;
          t.ty = b2;
        }
      }
      // workgroups that run at the same time cover 4 x 4 blocks of tiles, Y-fastest — pieces of the same rows on both
      // sides: 5-16 % on every 2-D shape of tools/xy_order.sh ([3652,27400] 211 -> 179 us), profiles/transpose_order_r02.txt;
      // 16 x 16 and more lose to the padding of the grid.  (Streaming or cached stores: no difference.)
      t.super = 4;
      if (const char *e = getenv("OLAP_XY_SUPER")) t.super = std::max(1, atoi(e));
      t.y_first = 1;
      if (const char *e = getenv("OLAP_XY_ORDER")) t.y_first = e[0] == 'y';
      t.cached_stores = getenv("OLAP_XY_CACHED_STORES") != nullptr;
      t.tiles_x = (t.lx + t.tx - 1) / t.tx;
      t.tiles_y = (t.ly + t.ty - 1) / t.ty;
      {
        // the grid is padded to whole blocks: with 5 tiles across, 4 x 4 blocks launch 8 (three of them empty) and the
        // Y-fastest walk loses its point — [3652,100,274] -> (sku, day, location) 176 -> 224 us; such shapes keep the
        // plain X-fastest walk
        const uint64_t k = (uint64_t)t.super;
        const uint64_t padded = (t.tiles_x + k - 1) / k * k * ((t.tiles_y + k - 1) / k * k);
        if (padded * 4 > t.tiles_x * t.tiles_y * 5 && !getenv("OLAP_XY_SUPER")) {  // (29 tiles padded to 32 still gain 15 %)
          t.super = 1;
          if (!getenv("OLAP_XY_ORDER")) t.y_first = 0;
        }
      }
      t.vec_in = !wide_cells && in4 && t.lx % 4 == 0;
      t.vec_out = !wide_cells && out4 && t.ly % 4 == 0;
      const bool is_float = dtype == OLAP_FLOAT32 || dtype == OLAP_FLOAT64;
      t.default_test = is_float ? (p->def_nan ? 2 : 1) : (p->def_nan ? 3 : 0);
      ok = (t.tiles_x + t.super - 1) / t.super * ((t.tiles_y + t.super - 1) / t.super) * t.super * t.super * t.batch < 0x7FFFFFFFull;  // the grid is padded to whole super-tiles
    }
    p->xy_ok = ok;
  }
  if (!contiguous_tail && (int)m.size() <= kMaxDims && p->out_cells > 0) {
    // brick transpose: pick chunk extents so that a brick is long in both orders
    const int n = (int)m.size();
    std::vector<uint64_t> out_stride(n);
    uint64_t st = 1;
    for (int d = n - 1; d >= 0; --d) {
      out_stride[d] = st;
      st *= m[d].len;
    }
    std::vector<int> by_in(n), by_out(n);
    for (int d = 0; d < n; ++d) by_in[d] = by_out[d] = d;
    std::sort(by_in.begin(), by_in.end(), [&](int x, int y) { return m[x].stride < m[y].stride; });
    std::sort(by_out.begin(), by_out.end(), [&](int x, int y) { return out_stride[x] < out_stride[y]; });
    // First choice, the 16-byte form (reorder_brick4_kernel): on each side take whole dimensions from
    // the fastest one until the contiguous run is >= 64 cells and a multiple of 4 (the last one taken
    // may be a divisor of its dimension), so no brick is ragged and every run is whole 16 B groups.
    std::vector<uint32_t> chunk(n, 1);
    bool quad_chunks = olap_dtype_size(dtype) == 4 && !getenv("OLAP_BRICK_NO_QUAD");
    if (quad_chunks) {
      auto grow_quad = [&](const std::vector<int> &order) {
        uint64_t prod = 1;
        for (int d : order) {
          if (prod >= 64 && prod % 4 == 0) break;
          uint32_t pick = m[d].len;
          for (uint32_t c = 1; c <= m[d].len && c <= 4096; ++c)
            if (m[d].len % c == 0 && prod * c >= 64 && (prod * c) % 4 == 0) {
              pick = c;
              break;
            }
          chunk[d] = std::max(chunk[d], pick);
          prod *= chunk[d];
          if (chunk[d] != m[d].len) break;  // a partial dimension ends the contiguous run
        }
        return prod >= 16 && prod % 4 == 0;
      };
      quad_chunks = grow_quad(by_in) && grow_quad(by_out);
      uint64_t e = 1;
      int active = 0;
      for (int d = 0; d < n; ++d) {
        e *= chunk[d];
        active += chunk[d] > 1;
        if (m[d].len % chunk[d] != 0) quad_chunks = false;
      }
      if (e > 12288 || active > 4) quad_chunks = false;  // 48 KiB of cells per brick (54 KiB padded)
    }
    // tuning knobs (developer use): OLAP_BRICK_TARGET = run length aimed at in both orders,
    // OLAP_BRICK_CAP = cells per brick
    uint64_t cap = (32 * 1024) / olap_dtype_size(dtype) / 2;  // 16 KiB of float32
    uint32_t first_target = 64;
    if (const char *e = getenv("OLAP_BRICK_TARGET")) first_target = (uint32_t)std::max(4, atoi(e));
    if (const char *e = getenv("OLAP_BRICK_CAP")) cap = (uint64_t)std::max(64, atoi(e));
    if (quad_chunks) cap = 12288;
    for (uint32_t target = first_target; target >= 4 && !quad_chunks; target /= 2) {
      std::fill(chunk.begin(), chunk.end(), 1u);
      auto grow = [&](const std::vector<int> &order) {
        uint64_t prod = 1;
        for (int d : order) {
          if (prod >= target) break;
          uint64_t want = (target + prod - 1) / prod;
          // prefer an extent that divides the dimension (no ragged bricks) when one is close
          for (uint64_t c2 = want; c2 <= 2 * want && c2 <= m[d].len; ++c2)
            if (m[d].len % c2 == 0) {
              want = c2;
              break;
            }
          const uint32_t c = (uint32_t)std::min<uint64_t>(m[d].len, want);
          chunk[d] = std::max(chunk[d], c);
          prod *= c;
        }
      };
      grow(by_in);
      grow(by_out);
      uint64_t e = 1;
      for (int d = 0; d < n; ++d) e *= chunk[d];
      if (e <= cap) break;
    }
    Brick &b = p->brick;
    b.nd = n;
    uint64_t elems = 1, bricks = 1;
    bool ragged = false, small_chunks = true;
    for (int d = 0; d < n; ++d) {
      b.len[d] = m[d].len;
      b.chunk[d] = chunk[d];
      b.nblk[d] = (m[d].len + chunk[d] - 1) / chunk[d];
      b.in_stride[d] = m[d].stride;
      b.out_stride[d] = out_stride[d];
      elems *= chunk[d];
      bricks *= b.nblk[d];
      ragged = ragged || (m[d].len % chunk[d]) != 0;
      small_chunks = small_chunks && chunk[d] <= 127;
    }
    std::vector<int> act_rd, act_wr;  // active dims, fastest first in each order
    for (int d : by_in)
      if (chunk[d] > 1) act_rd.push_back(d);
    for (int d : by_out)
      if (chunk[d] > 1) act_wr.push_back(d);
    // offsets inside a brick must fit 32 bits, digits one byte each
    uint64_t span_in = 0, span_out = 0;
    for (int d : act_rd) {
      span_in += (uint64_t)(chunk[d] - 1) * m[d].stride;
      span_out += (uint64_t)(chunk[d] - 1) * out_stride[d];
    }
    if (bricks < 0x7FFFFFFFull && elems <= cap && act_rd.size() <= 4 && (small_chunks || !ragged) && span_in < 0xFFFFFFFFull &&
        span_out < 0xFFFFFFFFull) {
      b.n_act = (int)act_rd.size();
      for (int k = 0; k < 4; ++k) b.act_dim[k] = k < b.n_act ? act_rd[k] : -1;
      b.elems = (uint32_t)elems;
      b.ragged = ragged;
      b.def_nan = p->def_nan;
      // tables: 5 arrays of `elems` uint32
      std::vector<uint32_t> tabs(5 * elems);
      uint32_t *rd_off = tabs.data(), *wr_off = rd_off + elems, *wr_lds = wr_off + elems, *rd_dig = wr_lds + elems,
               *wr_dig = rd_dig + elems;
      auto slot = [&](int d) {  // byte position of dim d in the packed digits
        for (int k = 0; k < b.n_act; ++k)
          if (act_rd[k] == d) return k;
        return 0;
      };
      std::vector<uint32_t> lds_stride(n, 0);
      {
        uint32_t ls = 1;
        for (int d : act_rd) {
          lds_stride[d] = ls;
          ls *= chunk[d];
        }
      }
      for (uint64_t e = 0; e < elems; ++e) {
        uint64_t c = e, off = 0;
        uint32_t dig = 0;
        for (int d : act_rd) {
          const uint32_t g = (uint32_t)(c % chunk[d]);
          c /= chunk[d];
          off += (uint64_t)g * m[d].stride;
          dig |= g << (8 * slot(d));
        }
        rd_off[e] = (uint32_t)off;
        rd_dig[e] = dig;
      }
      for (uint64_t f = 0; f < elems; ++f) {
        uint64_t c = f, off = 0;
        uint32_t dig = 0, pos = 0;
        for (int d : act_wr) {
          const uint32_t g = (uint32_t)(c % chunk[d]);
          c /= chunk[d];
          off += (uint64_t)g * out_stride[d];
          dig |= g << (8 * slot(d));
          pos += g * lds_stride[d];
        }
        wr_off[f] = (uint32_t)off;
        wr_dig[f] = dig;
        wr_lds[f] = pos;
      }
      // 16-byte form: groups of 4 cells adjacent and aligned on both sides, for every brick
      bool quad = quad_chunks && !ragged && elems % 4 == 0 && lds_stride.size() > 0;
      for (int d = 0; d < n && quad; ++d)
        if (b.nblk[d] > 1 && (((uint64_t)chunk[d] * m[d].stride) % 4 != 0 || ((uint64_t)chunk[d] * out_stride[d]) % 4 != 0)) quad = false;
      for (uint64_t q = 0; q < elems / 4 && quad; ++q)
        for (uint32_t j = 0; j < 4; ++j)
          if (rd_off[4 * q] % 4 != 0 || rd_off[4 * q + j] != rd_off[4 * q] + j || wr_off[4 * q] % 4 != 0 || wr_off[4 * q + j] != wr_off[4 * q] + j) quad = false;
      const size_t scalar_words = tabs.size();
      if (quad) {
        auto pad4 = [](uint32_t e) { return e + ((e >> 5) << 2); };
        if (pad4((uint32_t)elems) > 0xFFFFu) quad = false;
        else {
          tabs.resize(scalar_words + (elems / 4) * 4);  // rd_off4, wr_off4, wr_pos4 (2 words each)
          uint32_t *base = tabs.data();
          uint32_t *rd4 = base + scalar_words, *wr4 = rd4 + elems / 4, *pos4 = wr4 + elems / 4;
          for (uint64_t q = 0; q < elems / 4; ++q) {
            rd4[q] = base[4 * q];                // rd_off
            wr4[q] = base[elems + 4 * q];        // wr_off
            const uint32_t *lp = base + 2 * elems + 4 * q;  // wr_lds
            pos4[2 * q] = pad4(lp[0]) | (pad4(lp[1]) << 16);
            pos4[2 * q + 1] = pad4(lp[2]) | (pad4(lp[3]) << 16);
          }
        }
      }
      if ((rc = upload(&p->dev_tab, tabs.data(), tabs.size() * sizeof(uint32_t)))) {
        olap_plan_destroy(p);
        return rc;
      }
      const uint32_t *dev = (const uint32_t *)p->dev_tab;
      b.rd_off = dev;
      b.wr_off = dev + elems;
      b.wr_lds = dev + 2 * elems;
      b.rd_dig = dev + 3 * elems;
      b.wr_dig = dev + 4 * elems;
      b.quad = quad ? 1 : 0;
      b.rd_off4 = quad ? dev + scalar_words : nullptr;
      b.wr_off4 = quad ? dev + scalar_words + elems / 4 : nullptr;
      b.wr_pos4 = quad ? (const uint2 *)(dev + scalar_words + 2 * (elems / 4)) : nullptr;
      p->kind = PLAN_BRICK;
      p->n_bricks = bricks;
      p->kernel_name = p->xy_ok ? "transpose_xy_kernel" : quad ? "reorder_brick4_kernel" : "reorder_brick_kernel";
      *out = p;
      return OLAP_OK;
    }
  }
  if ((rc = finish_remap(p, dims, p->out_cells, true))) {
    olap_plan_destroy(p);
    return rc;
  }
  p->kernel_name = p->xy_ok ? "transpose_xy_kernel" : "gather(reorder)";
  *out = p;
  return OLAP_OK;
}

// ---- load -----------------------------------------------------------------------------------
extern "C" int olap_load_plan(olap_plan **out, int dtype, int my_default_kind, int his_default_kind,
                              int ndim, const uint32_t *my_len, const uint32_t *his_len,
                              const int32_t *const *his_to_mine) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(my_default_kind)) || (rc = check_default(his_default_kind))) return rc;
  if ((rc = check_dims(ndim, my_len, his_len))) return rc;
  if (ndim > 0 && !his_to_mine) return fail(OLAP_ERR_INVALID_ARGUMENT, "his_to_mine is NULL");
  for (int d = 0; d < ndim; ++d) {
    if (his_len[d] && !his_to_mine[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "his_to_mine[%d] is NULL", d);
    for (uint32_t j = 0; j < his_len[d]; ++j)
      if (his_to_mine[d][j] >= 0 && (uint32_t)his_to_mine[d][j] >= my_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "load map of dimension %d: entry %u = %d is outside this store's dimension (%u items)", d, j, his_to_mine[d][j], my_len[d]);
  }
  if ((rc = require_device())) return rc;
  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->kind = PLAN_LOAD;
  p->dtype = dtype;
  p->def_nan = my_default_kind == OLAP_DEFAULT_NAN;
  p->his_def_nan = his_default_kind == OLAP_DEFAULT_NAN;
  p->in_cells = product(his_len, ndim);
  p->out_cells = product(my_len, ndim);
  std::vector<RemapDim> dims(ndim);
  uint64_t stride = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    RemapDim &rd = dims[d];
    rd.len = his_len[d];
    rd.stride = stride;
    bool ident = his_len[d] == my_len[d];
    for (uint32_t j = 0; ident && j < his_len[d]; ++j) ident = his_to_mine[d][j] == (int32_t)j;
    rd.arithmetic = ident;
    if (!ident) {
      rd.table.resize(his_len[d]);
      for (uint32_t j = 0; j < his_len[d]; ++j)
        rd.table[j] = his_to_mine[d][j] < 0 ? -1 : (int64_t)his_to_mine[d][j] * (int64_t)stride;
    }
    stride *= my_len[d];
  }
  // the innermost dimension's items in another order and nothing else: rows rearranged through LDS
  bool permuted_rows = false;
  if (ndim >= 1 && !getenv("OLAP_LOAD_NO_PERMUTE")) {
    const int last = ndim - 1;
    const uint64_t row_cap = kTileBytes / olap_dtype_size(dtype);
    bool ok = !dims[last].arithmetic && his_len[last] == my_len[last] && his_len[last] >= 2 && his_len[last] <= row_cap;
    for (int d = 0; d < last && ok; ++d) ok = dims[d].arithmetic;
    std::vector<uint32_t> perm;
    if (ok) {
      std::vector<char> seen(my_len[last], 0);
      perm.resize(his_len[last]);
      for (uint32_t j = 0; j < his_len[last] && ok; ++j) {
        const int32_t m = his_to_mine[last][j];
        if (m < 0 || seen[m]) ok = false;
        else {
          seen[m] = 1;
          perm[j] = (uint32_t)m;
        }
      }
    }
    LoadPermute lp{};
    if (ok && small_div_for(his_len[last], row_cap, &lp.by_len)) {
      void *dev = nullptr;
      if ((rc = upload(&dev, perm.data(), perm.size() * sizeof(uint32_t)))) {
        olap_plan_destroy(p);
        return rc;
      }
      p->owned.push_back(dev);
      lp.perm = (const uint32_t *)dev;
      lp.len = his_len[last];
      lp.n_rows = p->in_cells / his_len[last];
      lp.rows_per_tile = (uint32_t)(row_cap / his_len[last]);
      lp.def_nan = p->def_nan;
      lp.src_def_nan = p->his_def_nan;
      p->lperm = lp;
      permuted_rows = true;
    }
  }
  // (16-byte lanes when the innermost merged run is contiguous in both stores: finish_remap checks it)
  if ((rc = finish_remap(p, dims, p->in_cells, true))) {
    olap_plan_destroy(p);
    return rc;
  }
  p->kernel_name = permuted_rows ? "load_permute_rows_kernel"
                   : p->vec > 1  ? "load_scatter (16-byte lanes)"
                                 : "load_scatter (16-byte runs of the other store)";
  *out = p;
  return OLAP_OK;
}

// ---- drillDown ------------------------------------------------------------------------------
extern "C" int olap_drilldown_plan(olap_plan **out, int dtype, int default_kind, int method, int ndim,
                                   const uint32_t *old_len, const uint32_t *new_len,
                                   const uint32_t *const *maps, const double *distributions,
                                   uint64_t n_dist) {
  if (!out) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan out-pointer is NULL");
  *out = nullptr;
  int rc;
  // the measure's declared type is int32 / uint32 whatever the cells are (olap_hip.h): in-memory.js:343
  const bool integer_measure = (method & OLAP_DRILLDOWN_INTEGER_MEASURE) != 0 || dtype == OLAP_INT32 || dtype == OLAP_UINT32;
  method &= ~OLAP_DRILLDOWN_INTEGER_MEASURE;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if ((rc = check_dims(ndim, old_len, new_len))) return rc;
  if (ndim > 0 && !maps) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps is NULL");
  for (int d = 0; d < ndim; ++d) {
    if (new_len[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
    for (uint32_t j = 0; j < new_len[d]; ++j)
      if (maps[d][j] >= old_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillDown map of dimension %d: entry %u = %u is outside the old dimension (%u items)", d, j, maps[d][j], old_len[d]);
  }
  if ((rc = require_device())) return rc;
  olap_plan *p = new (std::nothrow) olap_plan();
  if (!p) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  p->kind = PLAN_DRILLDOWN;
  p->dtype = dtype;
  p->def_nan = default_kind == OLAP_DEFAULT_NAN;
  p->method = method;
  p->in_cells = product(old_len, ndim);
  p->out_cells = product(new_len, ndim);
  DrillDown &a = p->dd;
  std::vector<uint32_t> tab;
  std::vector<uint64_t> old_stride(ndim);
  {
    uint64_t s = 1;
    for (int d = ndim - 1; d >= 0; --d) {
      old_stride[d] = s;
      s *= old_len[d];
    }
  }
  int nd = 0;
  bool prev_ident = false;
  DrillDownScale &sc = p->dds;
  std::vector<uint32_t> tab2;            // child counts per old index (two-pass form)
  std::vector<RemapDim> bcast;           // broadcast of the quotients (two-pass form)
  for (int d = 0; d < ndim; ++d) {
    const bool ident = is_identity_u32(maps[d], new_len[d], old_len[d]);
    if (ident && prev_ident && (uint64_t)a.new_len[nd - 1] * new_len[d] <= 0xFFFFFFFFull) {
      a.new_len[nd - 1] *= new_len[d];
      a.old_stride[nd - 1] = old_stride[d];
      sc.old_len[nd - 1] *= old_len[d];
      bcast.back().len *= new_len[d];
      bcast.back().stride = old_stride[d];
      continue;
    }
    if (nd == kMaxDims) {
      olap_plan_destroy(p);
      return fail(OLAP_ERR_INVALID_ARGUMENT, "drillDown: more than %d non-mergeable dimensions", kMaxDims);
    }
    a.new_len[nd] = new_len[d];
    a.old_stride[nd] = old_stride[d];
    sc.old_len[nd] = old_len[d];
    RemapDim rd;
    rd.len = new_len[d];
    rd.stride = old_stride[d];
    rd.arithmetic = ident;
    if (ident) {
      a.tab_off[nd] = -1;
      sc.tab_off[nd] = -1;
    } else {
      const uint32_t L = new_len[d];
      a.tab_off[nd] = (int32_t)tab.size();
      std::vector<uint32_t> count(old_len[d] ? old_len[d] : 1, 0), rank(L);
      for (uint32_t j = 0; j < L; ++j) rank[j] = count[maps[d][j]]++;
      for (uint32_t j = 0; j < L; ++j) tab.push_back(maps[d][j]);
      for (uint32_t j = 0; j < L; ++j) tab.push_back(count[maps[d][j]]);
      for (uint32_t j = 0; j < L; ++j) tab.push_back(rank[j]);
      sc.tab_off[nd] = (int32_t)tab2.size();
      tab2.insert(tab2.end(), count.begin(), count.begin() + old_len[d]);
      rd.table.resize(L);
      for (uint32_t j = 0; j < L; ++j) rd.table[j] = (int64_t)maps[d][j] * (int64_t)old_stride[d];
    }
    bcast.push_back(rd);
    prev_ident = ident;
    ++nd;
  }
  a.nd = nd;
  sc.nd = nd;
  sc.total = p->in_cells;
  sc.def_nan = p->def_nan;
  sc.divide = method == OLAP_SUM;
  // One refined axis and rows wide enough to fill workgroups: the row form (children streamed
  // from one read of the parent row).
  {
    int n_changed = 0, changed = -1;
    for (int d = 0; d < ndim; ++d)
      if (!is_identity_u32(maps[d], new_len[d], old_len[d])) {
        ++n_changed;
        changed = d;
      }
    if (!distributions && n_changed == 1) {
      DrillUpAxis &ax = p->axis;
      ax.outer = product(new_len, changed);
      ax.K = new_len[changed];   // children side
      ax.G = old_len[changed];   // parents side
      ax.inner = product(new_len + changed + 1, ndim - changed - 1);
      const int vec = vec_for(dtype, ax.inner);
      if (ax.inner / (uint64_t)vec >= 128 && ax.outer * (uint64_t)ax.K * ((ax.inner / vec + kBlock - 1) / kBlock) < 0x3FFFFFFFull) {
        // CSR of the children of every parent, ascending
        const uint32_t K = new_len[changed], G = old_len[changed];
        std::vector<uint32_t> gstart((size_t)G + 1, 0), order(K);
        for (uint32_t k = 0; k < K; ++k) gstart[maps[changed][k] + 1]++;
        for (uint32_t gg = 0; gg < G; ++gg) gstart[gg + 1] += gstart[gg];
        {
          std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
          for (uint32_t k = 0; k < K; ++k) order[cur[maps[changed][k]]++] = k;
        }
        bool contiguous = true;
        for (uint32_t k = 0; k < K; ++k) contiguous = contiguous && order[k] == k;
        for (uint32_t gg = 0; gg < G; ++gg) p->dd_longest = std::max(p->dd_longest, gstart[gg + 1] - gstart[gg]);
        std::vector<uint32_t> csr(gstart);
        const size_t order_at = csr.size();
        if (!contiguous) csr.insert(csr.end(), order.begin(), order.end());
        void *dev = nullptr;
        if ((rc = upload(&dev, csr.data(), csr.size() * sizeof(uint32_t)))) {
          olap_plan_destroy(p);
          return rc;
        }
        p->owned.push_back(dev);
        ax.gstart = (const uint32_t *)dev;
        ax.order = contiguous ? nullptr : (const uint32_t *)dev + order_at;
        ax.def_nan = p->def_nan;
        p->vec = vec;
        p->dd_rows = true;
      }
    }
  }
  p->dd_two_pass = !p->dd_rows && !distributions && !integer_measure;
  if (p->dd_two_pass) {
    if ((rc = finish_remap(p, bcast, p->out_cells, true))) {  // uploads p->dev_tab (int64 offsets)
      olap_plan_destroy(p);
      return rc;
    }
    p->dev_tab2 = p->dev_tab;  // keep: the per-cell tables below go to dev_tab
    p->dev_tab = nullptr;
    void *cnt = nullptr;
    if ((rc = upload(&cnt, tab2.data(), tab2.size() * sizeof(uint32_t)))) {
      olap_plan_destroy(p);
      return rc;
    }
    sc.tab = (const uint32_t *)cnt;
    p->dev_tmp = nullptr;
    hipError_t e = dev_alloc(&p->dev_tmp, (p->in_cells ? p->in_cells : 1) * olap_dtype_size(dtype) + (size_t)16);
    if (e != hipSuccess) {
      dev_free(cnt);
      olap_plan_destroy(p);
      return hip_fail(e, "hipMalloc(drillDown quotients)");
    }
    p->owned.push_back(cnt);
  }
  a.total = p->out_cells;
  a.def_nan = p->def_nan;
  a.method = method;
  a.use_rounding = integer_measure;  // in-memory.js:343
  a.dist = nullptr;
  a.n_dist = 0;
  if ((rc = upload(&p->dev_tab, tab.data(), tab.size() * sizeof(uint32_t)))) {
    olap_plan_destroy(p);
    return rc;
  }
  a.tab = (const uint32_t *)p->dev_tab;
  if (distributions) {
    void *dv = nullptr;
    if ((rc = upload(&dv, distributions, n_dist * sizeof(double)))) {
      olap_plan_destroy(p);
      return rc;
    }
    p->dev_dist = (double *)dv;
    a.dist = p->dev_dist;
    a.n_dist = n_dist;
    a.added_len = (double)p->out_cells / (double)p->in_cells;  // :392
    const double shared = (double)n_dist / a.added_len;        // :393
    a.chunk = (double)p->out_cells / shared;                   // :395
  }
  {
    void *ev = nullptr;
    const unsigned long long init = ~0ull;
    if ((rc = upload(&ev, &init, sizeof(init)))) {
      olap_plan_destroy(p);
      return rc;
    }
    p->dev_err = (unsigned long long *)ev;
    a.err = p->dev_err;
  }
  const bool dd_lines = p->dd_rows && !(method == OLAP_SUM && p->dd.use_rounding) && (p->axis.inner * olap_dtype_size(dtype)) % 128 != 0 &&
                        ((size_t)p->vec * olap_dtype_size(dtype) == 16 || p->axis.inner * olap_dtype_size(dtype) >= 2048);
  p->kernel_name = p->dd_rows ? (dd_lines ? "drilldown_rows_lines_kernel" : "drilldown_rows_kernel") : (p->dd_two_pass ? "drilldown_scale_kernel+gather_kernel" : "drilldown_kernel");
  *out = p;
  return OLAP_OK;
}

// ---- run ------------------------------------------------------------------------------------
static bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// `rule` >= 0: the drillUp rule to run with instead of the one the plan was built for (a drillUp plan's tables do not
// depend on it; nothing in the plan is written, so a cached plan stays shareable)
template <typename T>
static int run_typed(olap_plan *p, const void *in_v, const int32_t *in_s, void *out_v, int32_t *out_s,
                     hipStream_t stream, int rule = -1) {
  const int drillup_method = rule >= 0 ? rule : p->method;
  const T *in = (const T *)in_v;
  T *out = (T *)out_v;
  const bool hs = in_s != nullptr;
  hipError_t e = hipSuccess;
  if (p->xy_ok && !hs) {  // reorder without a mask to honour: pure permutation of the cells
    e = launch_transpose_xy(p->xy, (int)sizeof(T), in_v, out_v, out_s, aligned16(in_v) && aligned16(out_v) && (!out_s || aligned16(out_s)), stream);
    if (e != hipSuccess) return hip_fail(e, p->kernel_name.c_str());
    return OLAP_OK;
  }
  switch (p->kind) {
    case PLAN_DRILLUP_AXIS: {
      DrillUpAxis a = p->axis;
      int vec = p->vec;
      a.aligned16 = aligned16(in) && aligned16(out) && (!in_s || aligned16(in_s)) && (!out_s || aligned16(out_s));
      if (!a.aligned16) vec = 1;
      a.n_vec = a.inner / (uint64_t)vec;
      a.total = a.outer * a.G * a.n_vec;
      a.blocks_per_row = (a.n_vec + kBlock - 1) / kBlock;
      { const char *x = getenv("OLAP_XCD_ORDER"); a.xcd_order = x ? atoi(x) : 1; }  // 0: A/B against the dispatch order
      if (p->seg.S_tot > 0 && drillup_method != OLAP_PRODUCT && drillup_method != OLAP_PARTIAL_AVERAGE) {
        e = Launch<T>::drillup_segmented(drillup_method, hs, vec, in, in_s, out, out_s, a, p->seg, stream);
        break;
      }
      if (p->reduce.S > 0) {
        e = Launch<T>::drillup_reduce(drillup_method, hs, in, in_s, out, out_s, a, p->reduce, stream);
        break;
      }
      e = Launch<T>::drillup_axis(drillup_method, hs, vec, in, in_s, out, out_s, a, stream);
      break;
    }
    case PLAN_DRILLUP_GENERIC:
      e = Launch<T>::drillup_generic(drillup_method, hs, in, in_s, out, out_s, p->gen, stream);
      break;
    case PLAN_GATHER: {
      if (p->dice_direct && aligned16(out) && (!out_s || aligned16(out_s))) {
        e = Launch<T>::dice_direct(hs, in, in_s, out, out_s, p->dice_rows, stream);
        break;
      }
      Remap r = p->remap;
      int vec = p->vec;
      if (vec > 1 && !(aligned16(in) && aligned16(out) && (!in_s || aligned16(in_s)) && (!out_s || aligned16(out_s)))) {
        r.total *= (uint64_t)vec;
        vec = 1;
      }
      e = Launch<T>::gather(hs, vec, in, in_s, out, out_s, r, stream);
      break;
    }
    case PLAN_LOAD: {
      if (p->lperm.perm) {
        e = Launch<T>::load_permute(hs, in, in_s, out, out_s, p->lperm, stream);
        break;
      }
      Remap r = p->remap;
      int vec = p->vec;
      if (vec > 1 && !(aligned16(in) && aligned16(out) && (!in_s || aligned16(in_s)) && (!out_s || aligned16(out_s)))) {
        r.total *= (uint64_t)vec;
        vec = 1;
      }
      e = Launch<T>::load_scatter(hs, vec, in, in_s, out, out_s, r, stream);
      break;
    }
    case PLAN_BRICK:
      e = Launch<T>::reorder_brick(hs, in, in_s, out, out_s, p->brick, p->n_bricks, stream);
      break;
    case PLAN_GATHER_REDUCE: {
      GatherReduce g = p->gr;
      int vec = p->vec;
      if (vec > 1 && !(aligned16(in) && aligned16(out) && (!in_s || aligned16(in_s)) && (!out_s || aligned16(out_s)))) {
        g.r.total *= (uint64_t)vec;
        vec = 1;
      }
      e = Launch<T>::gather_reduce(p->method, hs, vec, in, in_s, out, out_s, g, stream);
      break;
    }
    case PLAN_DRILLDOWN: {
      if (p->dd_rows) {
        DrillUpAxis a = p->axis;
        int vec = p->vec;
        a.aligned16 = aligned16(in) && aligned16(out) && (!in_s || aligned16(in_s)) && (!out_s || aligned16(out_s));
        if (!a.aligned16) vec = 1;
        a.n_vec = a.inner / (uint64_t)vec;
        a.blocks_per_row = (a.n_vec + kBlock - 1) / kBlock;
        const bool divide = p->method == OLAP_SUM;
        const bool spread = divide && p->dd.use_rounding;
        // rows off the 128-byte grid: store line-aligned windows from LDS (drilldown_rows_lines_kernel)
        const bool whole_groups = vec * sizeof(T) == 16;  // rows are whole 16-byte groups
        if (!spread && a.aligned16 && (a.inner * sizeof(T)) % 128 != 0 && (whole_groups || a.inner * sizeof(T) >= 2048) &&
            !getenv("OLAP_DD_NO_LINES"))
          e = Launch<T>::drilldown_rows_lines(hs, !whole_groups, in, in_s, out, out_s, a, divide, p->dd_longest, stream);
        else
          e = Launch<T>::drilldown_rows(hs, vec, in, in_s, out, out_s, a, divide, p->dd.use_rounding, p->dd_longest, stream);
        break;
      }
      if (p->dd_two_pass) {
        T *q = (T *)p->dev_tmp;
        e = Launch<T>::drilldown_scale(hs, in, in_s, q, p->dds, stream);
        if (e != hipSuccess) break;
        Remap r = p->remap;
        int vec = p->vec;
        if (vec > 1 && !(aligned16(out) && (!out_s || aligned16(out_s)))) {
          r.total *= (uint64_t)vec;
          vec = 1;
        }
        e = Launch<T>::gather(false, vec, q, nullptr, out, out_s, r, stream);
        break;
      }
      const unsigned long long init = ~0ull;
      e = hipMemcpyAsync(p->dev_err, &init, sizeof(init), hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = Launch<T>::drilldown(hs, in, in_s, out, out_s, p->dd, stream);
      break;
    }
  }
  if (e != hipSuccess) return hip_fail(e, p->kernel_name.c_str());
  return OLAP_OK;
}

static int plan_run_rule(olap_plan *p, const void *in_values, const int32_t *in_status, void *out_values, int32_t *out_status, void *stream,
                         int rule) {
  if (!p) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan is NULL");
  if (plan_dry()) return fail(OLAP_ERR_NO_DEVICE, "OLAP_PLAN_DRY is set: plans are built for inspection only; libolapgpu has no CPU fallback");
  if ((p->in_cells && !in_values) || (p->out_cells && !out_values))
    return fail(OLAP_ERR_INVALID_ARGUMENT, "values pointers must not be NULL");
  {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != p->device)
      return fail(OLAP_ERR_INVALID_ARGUMENT, "plan was built on device %d but the current device is %d", p->device, cur);
  }
  hipStream_t s = (hipStream_t)stream;
  p->last_stream = s;
  p->ran = true;
  switch (p->dtype) {
    case OLAP_INT32: return run_typed<int32_t>(p, in_values, in_status, out_values, out_status, s, rule);
    case OLAP_UINT32: return run_typed<uint32_t>(p, in_values, in_status, out_values, out_status, s, rule);
    case OLAP_FLOAT32: return run_typed<float>(p, in_values, in_status, out_values, out_status, s, rule);
    default: return run_typed<double>(p, in_values, in_status, out_values, out_status, s, rule);
  }
}

extern "C" int olap_plan_run(olap_plan *p, const void *in_values, const int32_t *in_status,
                             void *out_values, int32_t *out_status, void *stream) {
  return plan_run_rule(p, in_values, in_status, out_values, out_status, stream, -1);
}

// One launch for several measures (Batch<T>): drillUp plans of one axis outside the cooperative reduce regime (whose
// workspace belongs to the plan); everything else runs pair by pair — still one call for the host.
template <typename T>
static int run_batch_typed(olap_plan *p, int n, const void *const *in_v, const int32_t *const *in_s, void *const *out_v,
                           int32_t *const *out_s, hipStream_t stream) {
  const bool hs = in_s && in_s[0];
  for (int first = 0; first < n; first += kMaxBatch) {
    const int nb = std::min(n - first, (int)kMaxBatch);
    Batch<T> b{};
    bool al = true;
    for (int i = 0; i < nb; ++i) {
      b.in[i] = (const T *)in_v[first + i];
      b.st_in[i] = hs ? in_s[first + i] : nullptr;
      b.out[i] = (T *)out_v[first + i];
      b.st_out[i] = out_s ? out_s[first + i] : nullptr;
      al = al && aligned16(b.in[i]) && aligned16(b.out[i]) && (!b.st_in[i] || aligned16(b.st_in[i])) && (!b.st_out[i] || aligned16(b.st_out[i]));
    }
    DrillUpAxis a = p->axis;
    int vec = p->vec;
    a.aligned16 = al;
    if (!al) vec = 1;
    a.n_vec = a.inner / (uint64_t)vec;
    a.total = a.outer * a.G * a.n_vec;
    a.blocks_per_row = (a.n_vec + kBlock - 1) / kBlock;
    { const char *x = getenv("OLAP_XCD_ORDER"); a.xcd_order = x ? atoi(x) : 1; }
    hipError_t e = Launch<T>::drillup_axis_batch(p->method, hs, vec, b, (unsigned)nb, a, stream);
    if (e != hipSuccess) return hip_fail(e, p->kernel_name.c_str());
  }
  return OLAP_OK;
}

// n pairs whose rules differ (methods[i]), same plan otherwise.  OLAP_OK when they went out as mixed-rule launches;
// OLAP_ERR_UNSUPPORTED (nothing launched) when this plan / these buffers need rule-by-rule launches.
constexpr int OLAP_MIXED_NOT_APPLICABLE = -1000;
template <typename T>
static int run_mixed_typed(olap_plan *p, int n, const int *methods, const void *const *in_v, const int32_t *const *in_s, void *const *out_v,
                           int32_t *const *out_s, hipStream_t stream) {
  const bool hs = in_s && in_s[0];
  bool deep = false;
  for (int i = 0; i < n; ++i) deep = deep || methods[i] == OLAP_PRODUCT;
  for (int first = 0; first < n; first += kMaxBatch) {
    const int nb = std::min(n - first, (int)kMaxBatch);
    Batch<T> b{};
    bool al = true;
    for (int i = 0; i < nb; ++i) {
      b.in[i] = (const T *)in_v[first + i];
      b.st_in[i] = hs ? in_s[first + i] : nullptr;
      b.out[i] = (T *)out_v[first + i];
      b.st_out[i] = out_s ? out_s[first + i] : nullptr;
      b.method[i] = methods[first + i];
      al = al && aligned16(b.in[i]) && aligned16(b.out[i]) && (!b.st_in[i] || aligned16(b.st_in[i])) && (!b.st_out[i] || aligned16(b.st_out[i]));
    }
    DrillUpAxis a = p->axis;
    a.aligned16 = al;
    a.n_vec = a.inner / (uint64_t)p->vec;
    a.total = a.outer * a.G * a.n_vec;
    a.blocks_per_row = (a.n_vec + kBlock - 1) / kBlock;
    { const char *x = getenv("OLAP_XCD_ORDER"); a.xcd_order = x ? atoi(x) : 1; }
    hipError_t e = Launch<T>::drillup_rows_mixed(hs, p->vec, b, (unsigned)nb, a, deep, stream);
    if (e == hipErrorNotSupported && first == 0) return OLAP_MIXED_NOT_APPLICABLE;
    if (e != hipSuccess) return hip_fail(e, "drillup_rows_mixed_kernel");
  }
  return OLAP_OK;
}

// (internal) the pairs of one plan with a rule each; masks on all pairs or on none
static int plan_run_mixed(olap_plan *p, int n, const int *methods, const void *const *in_values, const int32_t *const *in_status,
                          void *const *out_values, int32_t *const *out_status, hipStream_t s) {
  if (p->kind != PLAN_DRILLUP_AXIS || p->reduce.S != 0 || plan_dry() || getenv("OLAP_NO_MIXED_RULES")) return OLAP_MIXED_NOT_APPLICABLE;
  for (int i = 0; i < n; ++i)
    if (methods[i] < OLAP_SUM || methods[i] > OLAP_PRODUCT) return OLAP_MIXED_NOT_APPLICABLE;
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != p->device) return OLAP_MIXED_NOT_APPLICABLE;
  p->last_stream = s;
  p->ran = true;
  switch (p->dtype) {
    case OLAP_INT32: return run_mixed_typed<int32_t>(p, n, methods, in_values, in_status, out_values, out_status, s);
    case OLAP_UINT32: return run_mixed_typed<uint32_t>(p, n, methods, in_values, in_status, out_values, out_status, s);
    case OLAP_FLOAT32: return run_mixed_typed<float>(p, n, methods, in_values, in_status, out_values, out_status, s);
    default: return run_mixed_typed<double>(p, n, methods, in_values, in_status, out_values, out_status, s);
  }
}

extern "C" int olap_plan_run_batch_rules(olap_plan *p, int n, const int *methods, const void *const *in_values,
                                         const int32_t *const *in_status, void *const *out_values, int32_t *const *out_status, void *stream) {
  if (!p) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan is NULL");
  if (n < 0 || (n > 0 && (!methods || !in_values || !out_values))) return fail(OLAP_ERR_INVALID_ARGUMENT, "batch of %d: method and values lists must not be NULL", n);
  if (p->kind != PLAN_DRILLUP_AXIS && p->kind != PLAN_DRILLUP_GENERIC) return fail(OLAP_ERR_INVALID_ARGUMENT, "rules per pair need a drillUp plan");
  if (plan_dry()) return fail(OLAP_ERR_NO_DEVICE, "OLAP_PLAN_DRY is set: plans are built for inspection only; libolapgpu has no CPU fallback");
  bool masks_in = false, masks_out = false, mixed_masks = false;
  for (int i = 0; i < n; ++i) {
    if (methods[i] < OLAP_SUM || methods[i] > OLAP_PRODUCT) return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", methods[i]);
    if ((p->in_cells && !in_values[i]) || (p->out_cells && !out_values[i]))
      return fail(OLAP_ERR_INVALID_ARGUMENT, "values pointers must not be NULL (pair %d of the batch)", i);
    const bool mi = in_status && in_status[i], mo = out_status && out_status[i];
    if (i == 0) masks_in = mi, masks_out = mo;
    else if (mi != masks_in || mo != masks_out) mixed_masks = true;
  }
  if (n > 1 && !mixed_masks) {
    const int rc = plan_run_mixed(p, n, methods, in_values, in_status, out_values, out_status, (hipStream_t)stream);
    if (rc != OLAP_MIXED_NOT_APPLICABLE) return rc;
  }
  // pair by pair: a drillUp plan's tables do not depend on the rule
  int rc = OLAP_OK;
  for (int i = 0; i < n && !rc; ++i)
    rc = plan_run_rule(p, in_values[i], in_status ? in_status[i] : nullptr, out_values[i], out_status ? out_status[i] : nullptr, stream, methods[i]);
  return rc;
}

extern "C" int olap_plan_run_batch(olap_plan *p, int n, const void *const *in_values, const int32_t *const *in_status,
                                   void *const *out_values, int32_t *const *out_status, void *stream) {
  if (!p) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan is NULL");
  if (n < 0 || (n > 0 && (!in_values || !out_values))) return fail(OLAP_ERR_INVALID_ARGUMENT, "batch of %d: values pointer lists must not be NULL", n);
  if (plan_dry()) return fail(OLAP_ERR_NO_DEVICE, "OLAP_PLAN_DRY is set: plans are built for inspection only; libolapgpu has no CPU fallback");
  bool masks_in = false, masks_out = false, mixed = false;
  for (int i = 0; i < n; ++i) {
    if ((p->in_cells && !in_values[i]) || (p->out_cells && !out_values[i]))
      return fail(OLAP_ERR_INVALID_ARGUMENT, "values pointers must not be NULL (pair %d of the batch)", i);
    const bool mi = in_status && in_status[i], mo = out_status && out_status[i];
    if (i == 0) masks_in = mi, masks_out = mo;
    else if (mi != masks_in || mo != masks_out) mixed = true;
  }
  const bool one_launch = p->kind == PLAN_DRILLUP_AXIS && p->reduce.S == 0 && !mixed && n > 1;
  if (!one_launch) {
    for (int i = 0; i < n; ++i) {
      const int rc = olap_plan_run(p, in_values[i], in_status ? in_status[i] : nullptr, out_values[i], out_status ? out_status[i] : nullptr, stream);
      if (rc) return rc;
    }
    return OLAP_OK;
  }
  {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != p->device)
      return fail(OLAP_ERR_INVALID_ARGUMENT, "plan was built on device %d but the current device is %d", p->device, cur);
  }
  hipStream_t s = (hipStream_t)stream;
  p->last_stream = s;
  p->ran = true;
  switch (p->dtype) {
    case OLAP_INT32: return run_batch_typed<int32_t>(p, n, in_values, in_status, out_values, out_status, s);
    case OLAP_UINT32: return run_batch_typed<uint32_t>(p, n, in_values, in_status, out_values, out_status, s);
    case OLAP_FLOAT32: return run_batch_typed<float>(p, n, in_values, in_status, out_values, out_status, s);
    default: return run_batch_typed<double>(p, n, in_values, in_status, out_values, out_status, s);
  }
}

extern "C" int olap_plan_status(olap_plan *p) {
  if (!p) return fail(OLAP_ERR_INVALID_ARGUMENT, "plan is NULL");
  if (p->kind != PLAN_DRILLDOWN || !p->dd.dist || !p->ran) return OLAP_OK;
  unsigned long long bad = ~0ull;
  HIP_TRY(hipMemcpy(&bad, p->dev_err, sizeof(bad), hipMemcpyDeviceToHost));
  if (bad == ~0ull) return OLAP_OK;
  const DrillDown &a = p->dd;
  const double di = std::floor((double)bad / a.chunk) * a.added_len + std::fmod((double)bad, a.added_len);
  // JS prints an integral double without a fraction
  if (di == std::floor(di) && std::fabs(di) < 1e15) return fail(OLAP_ERR_DISTRIBUTION_MISSING, "distribution missing for index %lld", (long long)di);
  return fail(OLAP_ERR_DISTRIBUTION_MISSING, "distribution missing for index %.17g", di);
}

// ------------------------------------------------------------------ element-wise helpers
#define DISPATCH_DTYPE(dtype, CALL)                      \
  switch (dtype) {                                       \
    case OLAP_INT32: { using T = int32_t; CALL; break; } \
    case OLAP_UINT32: { using T = uint32_t; CALL; break; } \
    case OLAP_FLOAT32: { using T = float; CALL; break; }  \
    default: { using T = double; CALL; break; }           \
  }

extern "C" int olap_canonicalize(void *values, int32_t *status, uint64_t n, int dtype, int default_kind,
                                 int use_existing_status, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (n && !values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if (use_existing_status && !status) return fail(OLAP_ERR_INVALID_ARGUMENT, "use_existing_status needs a status buffer");
  if ((rc = require_device())) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(dtype, e = Launch<T>::canonicalize((T *)values, status, n, default_kind == OLAP_DEFAULT_NAN, use_existing_status, (hipStream_t)stream));
  if (e != hipSuccess) return hip_fail(e, "canonicalize");
  return OLAP_OK;
}

extern "C" int olap_convert_from_f64(const double *src, void *values, int32_t *status, uint64_t n, int dtype,
                                     int default_kind, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (n && (!values || !src)) return fail(OLAP_ERR_INVALID_ARGUMENT, "src/values is NULL");
  if ((rc = require_device())) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(dtype, e = Launch<T>::from_f64(src, (T *)values, status, n, default_kind == OLAP_DEFAULT_NAN, (hipStream_t)stream));
  if (e != hipSuccess) return hip_fail(e, "convert_from_f64");
  return OLAP_OK;
}

extern "C" int olap_convert_to_f64(const void *values, double *dst, uint64_t n, int dtype, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype))) return rc;
  if (n && (!values || !dst)) return fail(OLAP_ERR_INVALID_ARGUMENT, "values/dst is NULL");
  if ((rc = require_device())) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(dtype, e = Launch<T>::to_f64((const T *)values, dst, n, (hipStream_t)stream));
  if (e != hipSuccess) return hip_fail(e, "convert_to_f64");
  return OLAP_OK;
}

extern "C" int olap_fill_seeded(void *values, int32_t *status, uint64_t n, uint64_t first_cell, int dtype,
                                uint32_t seed, double frac, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype))) return rc;
  if (n && !values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if ((rc = require_device())) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(dtype, e = Launch<T>::fill_seeded((T *)values, status, n, first_cell, seed, frac, (hipStream_t)stream));
  if (e != hipSuccess) return hip_fail(e, "fill_seeded");
  return OLAP_OK;
}

extern "C" int olap_average_finish(void *values, const int32_t *counts, int32_t *out_status, uint64_t n, int dtype,
                                   int default_kind, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (n && (!values || !counts)) return fail(OLAP_ERR_INVALID_ARGUMENT, "values/counts is NULL");
  if ((rc = require_device())) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(dtype, e = Launch<T>::average_finish((T *)values, counts, out_status, n, default_kind == OLAP_DEFAULT_NAN, (hipStream_t)stream));
  if (e != hipSuccess) return hip_fail(e, "average_finish");
  return OLAP_OK;
}

// the copy runs on the null stream of the device that owns the buffer, so it is ordered behind the
// work enqueued there
static void enter_device_of(const void *device_ptr) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, device_ptr) == hipSuccess) (void)hipSetDevice(attr.device);
  else (void)hipGetLastError();
}
extern "C" int olap_memcpy_to_host(void *host, const void *device, uint64_t bytes) {
  if (bytes && (!host || !device)) return fail(OLAP_ERR_INVALID_ARGUMENT, "host/device pointer is NULL");
  int rc = require_device();
  if (rc) return rc;
  if (!bytes) return OLAP_OK;
  DeviceGuard guard;
  enter_device_of(device);
  HIP_TRY(hipMemcpy(host, device, bytes, hipMemcpyDeviceToHost));
  return OLAP_OK;
}
extern "C" int olap_memcpy_to_device(void *device, const void *host, uint64_t bytes) {
  if (bytes && (!host || !device)) return fail(OLAP_ERR_INVALID_ARGUMENT, "host/device pointer is NULL");
  int rc = require_device();
  if (rc) return rc;
  if (!bytes) return OLAP_OK;
  DeviceGuard guard;
  enter_device_of(device);
  HIP_TRY(hipMemcpy(device, host, bytes, hipMemcpyHostToDevice));
  return OLAP_OK;
}

// the box's achievable read ceiling: the best plain streaming read of tools/ceilings.hip's sweep — one tile of
// 2 x 256 sixteen-byte groups per workgroup, both loads in flight, non-temporal (profiles/ceilings_r01.txt)
__global__ __launch_bounds__(kBlock) void diag_read_kernel(const float *__restrict__ src, uint64_t n_vec, float *scratch) {
  const uint64_t base = (uint64_t)blockIdx.x * (2 * kBlock) + threadIdx.x;
  float acc = 0.f;
  Vec<float, 4> a{}, b{};
  if (base < n_vec) a = load_stream<float, 4>(src + base * 4);
  if (base + kBlock < n_vec) b = load_stream<float, 4>(src + (base + kBlock) * 4);
  acc = a.v[0] + a.v[1] + a.v[2] + a.v[3] + b.v[0] + b.v[1] + b.v[2] + b.v[3];
  if (acc == 123456.789f) scratch[blockIdx.x & 2047] = acc;  // keeps the loads alive without a store per lane
}

__global__ __launch_bounds__(kBlock) void diag_write_kernel(float *__restrict__ dst, uint64_t n_vec) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_vec) return;
  Vec<float, 4> x;
  x.v[0] = x.v[1] = x.v[2] = x.v[3] = (float)threadIdx.x;
  store_stream<float, 4>(dst + i * 4, x);
}

extern "C" int olap_diag_write_ceiling(void *device, uint64_t bytes, void *stream) {
  if (!device) return fail(OLAP_ERR_INVALID_ARGUMENT, "write ceiling: NULL buffer");
  int rc;
  if ((rc = require_device())) return rc;
  const uint64_t n_vec = bytes / 16;
  if (n_vec == 0) return OLAP_OK;
  const uint64_t grid = (n_vec + kBlock - 1) / kBlock;
  if (grid >= 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "write ceiling: buffer too large");
  hipLaunchKernelGGL(diag_write_kernel, (unsigned)grid, kBlock, 0, (hipStream_t)stream, (float *)device, n_vec);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "diag_write_kernel");
  return OLAP_OK;
}

extern "C" int olap_diag_tile_placement(int dtype, uint32_t K, uint32_t G, uint32_t inner, const uint32_t *map, uint32_t *cell_pos,
                                        uint32_t *group_bounds, uint32_t *pitch) {
  int rc;
  if ((rc = check_dtype(dtype))) return rc;
  if (!map || !cell_pos || !group_bounds || !pitch || K == 0 || G == 0 || inner == 0)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "tile placement: NULL argument or empty extent");
  const uint64_t budget = kTileBytes / olap_dtype_size(dtype);
  if ((uint64_t)K * inner > budget) return fail(OLAP_ERR_INVALID_ARGUMENT, "tile placement: a row of %llu cells does not fit a tile of %llu", (unsigned long long)K * inner, (unsigned long long)budget);
  std::vector<uint32_t> gstart((size_t)G + 1, 0), order(K);
  for (uint32_t k = 0; k < K; ++k) {
    if (map[k] >= G) return fail(OLAP_ERR_INDEX_RANGE, "tile placement: map entry %u = %u is outside the new dimension (%u items)", k, map[k], G);
    gstart[map[k] + 1]++;
  }
  for (uint32_t g = 0; g < G; ++g) gstart[g + 1] += gstart[g];
  std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
  for (uint32_t k = 0; k < K; ++k) order[cur[map[k]]++] = k;
  TilePerm tp;
  tile_perm_build(gstart.data(), order.data(), K, G, inner, budget, 16 / olap_dtype_size(dtype), &tp);
  memcpy(cell_pos, tp.cell.data(), (size_t)K * inner * sizeof(uint32_t));
  memcpy(group_bounds, tp.grp.data(), tp.grp.size() * sizeof(uint32_t));
  *pitch = tp.pitch;
  return OLAP_OK;
}

extern "C" int olap_diag_read_ceiling(const void *device, uint64_t bytes, void *scratch, void *stream) {
  if (!device || !scratch) return fail(OLAP_ERR_INVALID_ARGUMENT, "device/scratch is NULL");
  if (((uintptr_t)device & 15u) != 0) return fail(OLAP_ERR_INVALID_ARGUMENT, "buffer must be 16-byte aligned");
  int rc = require_device();
  if (rc) return rc;
  const uint64_t n_vec = bytes / 16;
  const uint64_t grid = (n_vec + 2 * kBlock - 1) / (2 * kBlock);
  if (grid == 0 || grid > 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "buffer size out of range for the read diagnostic");
  hipLaunchKernelGGL(diag_read_kernel, (unsigned)grid, kBlock, 0, (hipStream_t)stream, (const float *)device, n_vec, (float *)scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "diag_read_kernel");
  return OLAP_OK;
}

// ---- computed measures --------------------------------------------------------------------------
// the kernel is not templated: it lives in this translation unit
static int check_formula(const int32_t *code, int n_code, int n_consts, int n_inputs, int n_scalars) {
  if (!code || n_code <= 0 || n_code > OLAP_FORMULA_MAX_CODE) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program has %d words (1..%d allowed)", n_code, OLAP_FORMULA_MAX_CODE);
  if (n_consts < 0 || n_consts > OLAP_FORMULA_MAX_CONSTS || n_inputs < 0 || n_inputs > OLAP_FORMULA_MAX_INPUTS || n_scalars < 0 ||
      n_scalars > OLAP_FORMULA_MAX_INPUTS)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "formula uses too many constants / measures / totals");
  int depth = 0;
  for (int pc = 0; pc < n_code; ++pc) {
    const int op = code[pc];
    if (op == F_CONST || op == F_INPUT || op == F_SCALAR) {
      if (pc + 1 >= n_code) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program truncated");
      const int k = code[++pc];
      const int lim = op == F_CONST ? n_consts : (op == F_INPUT ? n_inputs : n_scalars);
      if (k < 0 || k >= lim) return fail(OLAP_ERR_INDEX_RANGE, "formula operand %d out of range", k);
      ++depth;
    } else if (op == F_SELECT) {
      if (depth < 3) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program underflows its stack");
      depth -= 2;
    } else if (op == F_NEG || op == F_ISNAN || (op >= F_ABS && op <= F_NOT)) {
      if (depth < 1) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program underflows its stack");
    } else if (op >= F_ADD && op <= F_ROUNDTO) {
      if (depth < 2) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program underflows its stack");
      --depth;
    } else {
      return fail(OLAP_ERR_INVALID_ARGUMENT, "unknown formula opcode %d", op);
    }
    if (depth > OLAP_FORMULA_MAX_STACK) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula needs a stack deeper than %d", OLAP_FORMULA_MAX_STACK);
  }
  if (depth != 1) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula program leaves %d values on its stack", depth);
  return OLAP_OK;
}

extern "C" int olap_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                                 const void *const *in_values, const int32_t *const *in_status, const int *in_dtypes,
                                 const int *in_defaults, const double *scalars, int n_scalars, double *out_f64,
                                 uint64_t n, void *stream) {
  int rc = check_formula(code, n_code, n_consts, n_inputs, n_scalars);
  if (rc) return rc;
  if ((n_consts && !consts) || (n_scalars && !scalars) || (n_inputs && (!in_values || !in_dtypes || !in_defaults)))
    return fail(OLAP_ERR_INVALID_ARGUMENT, "formula argument arrays must not be NULL");
  for (int k = 0; k < n_inputs; ++k) {
    if ((rc = check_dtype(in_dtypes[k])) || (rc = check_default(in_defaults[k]))) return rc;
    if (n && !in_values[k]) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula input %d is NULL", k);
  }
  if (n && !out_f64) return fail(OLAP_ERR_INVALID_ARGUMENT, "out is NULL");
  if ((rc = require_device())) return rc;
  if (n == 0) return OLAP_OK;
  FormulaProgram p{};
  p.n_code = n_code;
  memcpy(p.code, code, n_code * sizeof(int32_t));
  if (n_consts) memcpy(p.consts, consts, n_consts * sizeof(double));
  p.n_inputs = n_inputs;
  for (int k = 0; k < n_inputs; ++k) {
    p.in_values[k] = in_values[k];
    p.in_status[k] = in_status ? in_status[k] : nullptr;
    p.in_dtype[k] = in_dtypes[k];
    p.in_def_nan[k] = in_defaults[k] == OLAP_DEFAULT_NAN;
  }
  for (int k = 0; k < n_scalars; ++k) p.scalars[k] = scalars[k];
  hipLaunchKernelGGL(eval_formula_kernel<OLAP_FORMULA_MAX_STACK>, grid_stride_for(n), kBlock, 0, (hipStream_t)stream, p, out_f64, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "eval_formula");
  return OLAP_OK;
}

extern "C" int olap_total(const void *values, const int32_t *status, uint64_t n, int dtype, int default_kind,
                          double *total, uint64_t *n_set, void *stream) {
  int rc;
  if ((rc = check_dtype(dtype)) || (rc = check_default(default_kind))) return rc;
  if (n && !values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if ((rc = require_device())) return rc;
  struct Acc {
    double total;
    unsigned long long count;
  } host = {0.0, 0};
  if (n == 0) {
    if (total) *total = 0.0;
    if (n_set) *n_set = 0;
    return OLAP_OK;
  }
  // [0]: the result pair, [1..]: one partial slot per workgroup (deterministic two-stage reduction)
  Acc *dev = nullptr;
  HIP_TRY(dev_alloc((void **)&dev, sizeof(Acc) * (1 + (size_t)kTotalBlocks)));
  hipError_t e = hipSuccess;
  // the result pair lands in pinned host memory the finishing kernel writes directly (one per host thread, kept): the
  // call then costs one stream synchronisation instead of that plus a blocking 16-byte copy
  static thread_local Acc *pinned = nullptr;
  static thread_local bool pinned_tried = false;
  if (!pinned_tried) {
    pinned_tried = true;
    void *q = nullptr;
    if (hipHostMalloc(&q, sizeof(Acc), hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) pinned = (Acc *)q;
    else (void)hipGetLastError();
  }
  Acc *result = pinned ? pinned : dev;
  DISPATCH_DTYPE(dtype, e = Launch<T>::total((const T *)values, status, n, default_kind == OLAP_DEFAULT_NAN, dev + 1, &result->total, &result->count, (hipStream_t)stream));
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) {
    if (pinned) host = *pinned;
    else e = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost);
  }
  dev_free(dev);
  if (e != hipSuccess) return hip_fail(e, "total");
  if (total) *total = host.total;
  if (n_set) *n_set = host.count;
  return OLAP_OK;
}

// ------------------------------------------------------------------ store handles
// values are always resident.  The Int32 mask is materialised lazily: for float cells and for a
// zero default it is a pure function of the values (set <=> value != default, in-memory.js:122-133),
// so bulk operations neither read nor write it and it is built on first request
// (olap_store_get_status / _status_ptr / _get_keys).  Integer cells under a NaN default are the one
// case where the mask carries information of its own; such stores always hold it.
// (struct olap_store: olap_internal.hpp)

bool mask_is_primary(const olap_store *s) {
  return s->default_kind == OLAP_DEFAULT_NAN && (s->dtype == OLAP_INT32 || s->dtype == OLAP_UINT32);
}

int store_alloc(olap_store **out, uint64_t size, int dtype, int default_kind) {
  olap_store *s = new (std::nothrow) olap_store();
  if (!s) return fail(OLAP_ERR_OUT_OF_MEMORY, "out of host memory");
  s->size = size;
  s->dtype = dtype;
  s->default_kind = default_kind;
  s->values = nullptr;
  s->status = nullptr;
  s->device = 0;
  (void)hipGetDevice(&s->device);
  s->maybe_nonempty = true;  // results of operations; olap_store_create clears it
  s->hi_index = size ? size - 1 : 0;
  const size_t vb = (size ? size : 1) * olap_dtype_size(dtype), sb = (size ? size : 1) * sizeof(int32_t);
  hipError_t e = dev_alloc(&s->values, vb);
  if (e == hipSuccess && mask_is_primary(s)) e = dev_alloc((void **)&s->status, sb);
  if (e != hipSuccess) {
    if (s->values) dev_free(s->values);
    delete s;
    return hip_fail(e, "hipMalloc(store)");
  }
  *out = s;
  return OLAP_OK;
}

// builds the mask from the values when it has not been materialised yet
int ensure_status(const olap_store *s) {
  if (s->status) return OLAP_OK;
  int32_t *st = nullptr;
  HIP_TRY(dev_alloc((void **)&st, (s->size ? s->size : 1) * sizeof(int32_t)));
  s->status = st;
  int rc = olap_canonicalize(s->values, s->status, s->size, s->dtype, s->default_kind, 0, nullptr);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return OLAP_OK;
}

// every cell unset: values = default (and an all-zero mask where the mask is primary)
static int store_clear(olap_store *s) {
  const size_t n1 = s->size ? s->size : 1;
  if (s->status) HIP_TRY(hipMemsetAsync(s->status, 0, n1 * sizeof(int32_t), nullptr));
  const bool fnan = s->default_kind == OLAP_DEFAULT_NAN && (s->dtype == OLAP_FLOAT32 || s->dtype == OLAP_FLOAT64);
  if (!fnan) {
    HIP_TRY(hipMemsetAsync(s->values, 0, n1 * olap_dtype_size(s->dtype), nullptr));
  } else if (s->size) {
    // one NaN cell, then doubling device-to-device copies
    int rc = olap_store_fill(s, NAN);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
  return OLAP_OK;
}

extern "C" int olap_store_create(olap_store **store, uint64_t size, int dtype, int default_kind) {
  if (!store) return fail(OLAP_ERR_INVALID_ARGUMENT, "store out-pointer is NULL");
  *store = nullptr;
  int rc;
  if ((rc = check_default(default_kind)) || (rc = check_dtype(dtype))) return rc;  // same order as in-memory.js:56-60
  if (size > 1000000000000ull) return fail(OLAP_ERR_INVALID_ARGUMENT, "cube too large");
  if ((rc = require_device())) return rc;
  olap_store *s = nullptr;
  if ((rc = store_alloc(&s, size, dtype, default_kind))) return rc;
  if ((rc = store_clear(s))) {
    olap_store_destroy(s);
    return rc;
  }
  s->maybe_nonempty = false;  // nothing written yet: cells that arrive in ascending order stay in ascending order
  s->hi_index = 0;
  *store = s;
  return OLAP_OK;
}

/* Insertion order of the reference's Map (see olap_order.hip). */
extern "C" int olap_store_track_order(olap_store *s, int on) {
  OnStoreDevice on_device__(s);
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (on && s->size >= 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "insertion-order tracking supports stores below 2^31 cells");
  s->track_order = on != 0;
  if (!on) order_free(s);
  return OLAP_OK;
}
extern "C" int olap_store_order_tracked(const olap_store *s) { return s && s->track_order ? (s->seq ? 2 : 1) : 0; }

extern "C" void olap_store_destroy(olap_store *s) {
  OnStoreDevice on_device__(s);
  if (!s) return;
  if (s->values) dev_free(s->values);
  if (s->status) dev_free(s->status);
  order_free(s);
  delete s;
}

extern "C" int olap_store_clone(const olap_store *s, olap_store **out) {
  OnStoreDevice on_device__(s);
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  olap_store *c = nullptr;
  int rc = store_alloc(&c, s->size, s->dtype, s->default_kind);
  if (rc) return rc;
  hipError_t e = hipMemcpy(c->values, s->values, s->size * olap_dtype_size(s->dtype), hipMemcpyDeviceToDevice);
  if (e == hipSuccess && c->status) e = hipMemcpy(c->status, s->status, s->size * sizeof(int32_t), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) {
    olap_store_destroy(c);
    return hip_fail(e, "store_clone");
  }
  if ((rc = order_clone(s, c))) {
    olap_store_destroy(c);
    return rc;
  }
  *out = c;
  return OLAP_OK;
}

extern "C" uint64_t olap_store_size(const olap_store *s) { return s ? s->size : 0; }
extern "C" int olap_store_dtype(const olap_store *s) { return s ? s->dtype : -1; }
extern "C" int olap_store_default(const olap_store *s) { return s ? s->default_kind : -1; }
extern "C" uint64_t olap_store_byte_length(const olap_store *s) { return s ? s->size * olap_dtype_size(s->dtype) : 0; }
extern "C" void *olap_store_values_ptr(const olap_store *s) { return s ? s->values : nullptr; }
extern "C" int32_t *olap_store_status_ptr(const olap_store *s) {
  OnStoreDevice on_device__(s);
  if (!s || ensure_status(s)) return nullptr;
  return s->status;
}

static int check_length(const olap_store *s, uint64_t n) {
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->size != n)
    return fail(OLAP_ERR_LENGTH_MISMATCH, "value length is invalid: %llu !== %llu", (unsigned long long)s->size, (unsigned long long)n);
  return OLAP_OK;
}

// a bulk write replaces every cell: a lazily built mask is dropped, a primary one is rewritten
void drop_lazy_status(olap_store *s) {
  if (s->status && !mask_is_primary(s)) {
    dev_free(s->status);
    s->status = nullptr;
  }
}

extern "C" int olap_store_set_data(olap_store *s, const void *host_values, uint64_t n) {
  OnStoreDevice on_device__(s);
  int rc = check_length(s, n);
  if (rc) return rc;
  if (n && !host_values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if (n == 0) return OLAP_OK;
  if ((rc = order_before_bulk_write(s))) return rc;
  drop_lazy_status(s);
  HIP_TRY(hipMemcpy(s->values, host_values, n * olap_dtype_size(s->dtype), hipMemcpyHostToDevice));
  if ((rc = olap_canonicalize(s->values, s->status, n, s->dtype, s->default_kind, 0, nullptr))) return rc;
  if ((rc = order_after_bulk_write(s))) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return OLAP_OK;
}

extern "C" int olap_store_set_data_f64(olap_store *s, const double *host_values, uint64_t n) {
  OnStoreDevice on_device__(s);
  int rc = check_length(s, n);
  if (rc) return rc;
  if (n && !host_values) return fail(OLAP_ERR_INVALID_ARGUMENT, "values is NULL");
  if (n == 0) return OLAP_OK;
  if ((rc = order_before_bulk_write(s))) return rc;
  drop_lazy_status(s);
  double *tmp = nullptr;
  HIP_TRY(dev_alloc((void **)&tmp, n * sizeof(double)));
  hipError_t e = hipMemcpy(tmp, host_values, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = olap_convert_from_f64(tmp, s->values, s->status, n, s->dtype, s->default_kind, nullptr);
    if (!rc) rc = order_after_bulk_write(s);
    if (!rc) e = hipStreamSynchronize(nullptr);
  }
  dev_free(tmp);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "set_data_f64");
  return OLAP_OK;
}

extern "C" int olap_store_get_data(const olap_store *s, void *host_values) {
  OnStoreDevice on_device__(s);
  if (!s || (s->size && !host_values)) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/values is NULL");
  if (s->size) HIP_TRY(hipMemcpy(host_values, s->values, s->size * olap_dtype_size(s->dtype), hipMemcpyDeviceToHost));
  return OLAP_OK;
}

extern "C" int olap_store_get_data_f64(const olap_store *s, double *host_values) {
  OnStoreDevice on_device__(s);
  if (!s || (s->size && !host_values)) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/values is NULL");
  if (!s->size) return OLAP_OK;
  if (s->dtype == OLAP_FLOAT64) return olap_store_get_data(s, host_values);
  double *tmp = nullptr;
  HIP_TRY(dev_alloc((void **)&tmp, s->size * sizeof(double)));
  int rc = olap_convert_to_f64(s->values, tmp, s->size, s->dtype, nullptr);
  hipError_t e = hipSuccess;
  if (!rc) e = hipMemcpy(host_values, tmp, s->size * sizeof(double), hipMemcpyDeviceToHost);
  dev_free(tmp);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "get_data_f64");
  // integer cells under a NaN default read back as NaN where unset, like getValue (:118-120)
  if (s->default_kind == OLAP_DEFAULT_NAN && (s->dtype == OLAP_INT32 || s->dtype == OLAP_UINT32)) {
    std::vector<int32_t> st(s->size);
    HIP_TRY(hipMemcpy(st.data(), s->status, s->size * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < s->size; ++i)
      if (!(st[i] & OLAP_STATUS_SET)) host_values[i] = NAN;
  }
  return OLAP_OK;
}

extern "C" int olap_store_get_status(const olap_store *s, int32_t *host_status) {
  OnStoreDevice on_device__(s);
  if (!s || (s->size && !host_status)) return fail(OLAP_ERR_INVALID_ARGUMENT, "store/status is NULL");
  int rc = ensure_status(s);
  if (rc) return rc;
  if (s->size) HIP_TRY(hipMemcpy(host_status, s->status, s->size * sizeof(int32_t), hipMemcpyDeviceToHost));
  return OLAP_OK;
}

extern "C" int olap_store_count_set(const olap_store *s, uint64_t *n_set) {
  OnStoreDevice on_device__(s);
  if (!s || !n_set) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  return olap_total(s->values, s->status, s->size, s->dtype, s->default_kind, nullptr, n_set, nullptr);
}

extern "C" int olap_store_get_keys(const olap_store *s, uint64_t *host_keys, uint64_t cap, uint64_t *n_keys) {
  OnStoreDevice on_device__(s);
  if (!s || !n_keys) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->seq) {  // a tracked store whose order is not the flat index: the reference's _dataMap.keys()
    std::vector<uint64_t> keys;
    int rco = order_sorted_keys(s, keys);
    if (rco) return rco;
    for (uint64_t j = 0; j < keys.size() && host_keys && j < cap; ++j) host_keys[j] = keys[j];
    *n_keys = keys.size();
    return OLAP_OK;
  }
  int rc = ensure_status(s);
  if (rc) return rc;
  std::vector<int32_t> st(s->size ? s->size : 1);
  if (s->size) HIP_TRY(hipMemcpy(st.data(), s->status, s->size * sizeof(int32_t), hipMemcpyDeviceToHost));
  uint64_t n = 0;
  for (uint64_t i = 0; i < s->size; ++i)
    if (st[i] & OLAP_STATUS_SET) {
      if (host_keys && n < cap) host_keys[n] = i;
      ++n;
    }
  *n_keys = n;
  return OLAP_OK;
}

extern "C" int olap_store_get_value(const olap_store *s, uint64_t index, double *value, int *is_set) {
  OnStoreDevice on_device__(s);
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (index >= s->size) {  // Map.get of an absent key: the default (:118-120)
    if (value) *value = s->default_kind == OLAP_DEFAULT_NAN ? NAN : 0.0;
    if (is_set) *is_set = 0;
    return OLAP_OK;
  }
  int32_t st = OLAP_STATUS_SET;
  double v = 0;
  // one lane writes (value, status) into pinned host memory (one slot per host thread, kept) and the call waits for the
  // stream once — instead of one or two blocking copies; the copies remain as the fallback
  static thread_local CellOut *pinned = nullptr;
  static thread_local bool pinned_tried = false;
  if (!pinned_tried) {
    pinned_tried = true;
    void *q = nullptr;
    if (hipHostMalloc(&q, sizeof(CellOut), hipHostMallocPortable | hipHostMallocMapped) == hipSuccess) pinned = (CellOut *)q;
    else (void)hipGetLastError();
  }
  if (pinned) {
    hipError_t e = hipSuccess;
    DISPATCH_DTYPE(s->dtype, e = Launch<T>::get_cell((const T *)s->values, s->status, index, pinned, nullptr));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return hip_fail(e, "getValue");
    v = pinned->value;
    st = pinned->status;
  } else {
    if (s->status) HIP_TRY(hipMemcpy(&st, s->status + index, sizeof(st), hipMemcpyDeviceToHost));
    const size_t es = olap_dtype_size(s->dtype);
    unsigned char raw[8];
    HIP_TRY(hipMemcpy(raw, (const char *)s->values + index * es, es, hipMemcpyDeviceToHost));
    switch (s->dtype) {
      case OLAP_INT32: { int32_t x; memcpy(&x, raw, 4); v = x; break; }
      case OLAP_UINT32: { uint32_t x; memcpy(&x, raw, 4); v = x; break; }
      case OLAP_FLOAT32: { float x; memcpy(&x, raw, 4); v = x; break; }
      default: memcpy(&v, raw, 8);
    }
  }
  bool set = (st & OLAP_STATUS_SET) != 0;
  if (!mask_is_primary(s)) {  // set <=> value != default
    const bool is_float = s->dtype == OLAP_FLOAT32 || s->dtype == OLAP_FLOAT64;
    set = set && (s->default_kind == OLAP_DEFAULT_NAN ? !(is_float && v != v) : v != 0.0);
  }
  if (value) *value = set ? v : (s->default_kind == OLAP_DEFAULT_NAN ? NAN : 0.0);
  if (is_set) *is_set = set;
  return OLAP_OK;
}

extern "C" int olap_store_set_value(olap_store *s, uint64_t index, double value, int is_null) {
  OnStoreDevice on_device__(s);
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (index >= s->size) return fail(OLAP_ERR_INDEX_RANGE, "cell index %llu out of bounds [0, %llu[", (unsigned long long)index, (unsigned long long)s->size);
  int rc = order_before_set_value(s, index);
  if (rc) return rc;
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(s->dtype, e = Launch<T>::set_cell((T *)s->values, s->status, index, value, is_null, s->default_kind == OLAP_DEFAULT_NAN, nullptr));
  if (e == hipSuccess && (rc = order_after_set_value(s, index))) return rc;
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) return hip_fail(e, "set_value");
  return OLAP_OK;
}

extern "C" int olap_store_fill(olap_store *s, double value) {
  OnStoreDevice on_device__(s);
  if (!s) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (!s->size) return OLAP_OK;
  // fill = setValue(i, value) for every i (:135-137): one converted cell, broadcast
  int rc0 = order_before_bulk_write(s);
  if (rc0) return rc0;
  drop_lazy_status(s);
  std::vector<double> one(1, value);
  double *tmp = nullptr;
  HIP_TRY(dev_alloc((void **)&tmp, sizeof(double)));
  hipError_t e = hipMemcpy(tmp, one.data(), sizeof(double), hipMemcpyHostToDevice);
  int rc = OLAP_OK;
  if (e == hipSuccess) rc = olap_convert_from_f64(tmp, s->values, s->status, 1, s->dtype, s->default_kind, nullptr);
  if (e == hipSuccess && !rc) e = hipStreamSynchronize(nullptr);
  dev_free(tmp);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "fill");
  // replicate cell 0 (value + status) by doubling copies
  const size_t es = olap_dtype_size(s->dtype);
  uint64_t done = 1;
  while (done < s->size) {
    const uint64_t n = std::min(done, s->size - done);
    HIP_TRY(hipMemcpyAsync((char *)s->values + done * es, s->values, n * es, hipMemcpyDeviceToDevice, nullptr));
    if (s->status) HIP_TRY(hipMemcpyAsync(s->status + done, s->status, n * sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
    done += n;
  }
  if ((rc = order_after_bulk_write(s))) return rc;
  HIP_TRY(hipStreamSynchronize(nullptr));
  return OLAP_OK;
}

extern "C" int olap_store_total(const olap_store *s, double *total) {
  OnStoreDevice on_device__(s);
  if (!s || !total) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  return olap_total(s->values, s->status, s->size, s->dtype, s->default_kind, total, nullptr, nullptr);
}

extern "C" int olap_store_eval_formula(const int32_t *code, int n_code, const double *consts, int n_consts, int n_inputs,
                                       const olap_store *const *inputs, const double *scalars, int n_scalars,
                                       double *host_out) {
  int rc = check_formula(code, n_code, n_consts, n_inputs, n_scalars);
  if (rc) return rc;
  if (n_inputs <= 0 || !inputs) return fail(OLAP_ERR_INVALID_ARGUMENT, "a formula needs at least one stored measure to size its result");
  const void *vals[OLAP_FORMULA_MAX_INPUTS];
  const int32_t *stat[OLAP_FORMULA_MAX_INPUTS];
  int dtypes[OLAP_FORMULA_MAX_INPUTS], defs[OLAP_FORMULA_MAX_INPUTS];
  const uint64_t n = inputs[0] ? inputs[0]->size : 0;
  for (int k = 0; k < n_inputs; ++k) {
    if (!inputs[k]) return fail(OLAP_ERR_INVALID_ARGUMENT, "formula input %d is NULL", k);
    if (inputs[k]->size != n) return fail(OLAP_ERR_LENGTH_MISMATCH, "formula inputs have different sizes");
    vals[k] = inputs[k]->values;
    stat[k] = mask_is_primary(inputs[k]) ? inputs[k]->status : nullptr;
    dtypes[k] = inputs[k]->dtype;
    defs[k] = inputs[k]->default_kind;
  }
  if (n && !host_out) return fail(OLAP_ERR_INVALID_ARGUMENT, "out is NULL");
  if (n == 0) return OLAP_OK;
  double *dev = nullptr;
  HIP_TRY(dev_alloc((void **)&dev, n * sizeof(double)));
  rc = olap_eval_formula(code, n_code, consts, n_consts, n_inputs, vals, stat, dtypes, defs, scalars, n_scalars, dev, n, nullptr);
  hipError_t e = hipSuccess;
  if (!rc) e = hipMemcpy(host_out, dev, n * sizeof(double), hipMemcpyDeviceToHost);
  dev_free(dev);
  if (rc) return rc;
  if (e != hipSuccess) return hip_fail(e, "store_eval_formula");
  return OLAP_OK;
}

// ---- sparse form (the reference's serialised layout, in-memory.js:75-116) ------------------------
extern "C" int olap_store_to_sparse(const olap_store *s, uint32_t *host_indexes, void *host_values, uint64_t cap,
                                    uint64_t *n_set) {
  OnStoreDevice on_device__(s);
  if (!s || !n_set) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->size > 0xFFFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "sparse form uses 32-bit indexes: store too large");
  *n_set = 0;
  if (s->size == 0) return OLAP_OK;
  const unsigned n_chunks = (unsigned)std::min<uint64_t>(2048, (s->size + kBlock - 1) / kBlock);
  const uint64_t chunk = (s->size + n_chunks - 1) / n_chunks;
  const int32_t *mask = mask_is_primary(s) ? s->status : nullptr;
  unsigned long long *dev_counts = nullptr;
  HIP_TRY(dev_alloc((void **)&dev_counts, n_chunks * sizeof(unsigned long long)));
  std::vector<unsigned long long> counts(n_chunks), offsets(n_chunks);
  hipError_t e = hipSuccess;
  DISPATCH_DTYPE(s->dtype, e = Launch<T>::compact_count((const T *)s->values, mask, s->size, chunk, n_chunks, s->default_kind == OLAP_DEFAULT_NAN, dev_counts, nullptr));
  if (e == hipSuccess) e = hipMemcpy(counts.data(), dev_counts, n_chunks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  unsigned long long total = 0;
  for (unsigned b = 0; b < n_chunks; ++b) {
    offsets[b] = total;
    total += counts[b];
  }
  *n_set = total;
  if (e == hipSuccess && host_indexes && host_values && total > 0) {
    if (cap < total) {
      dev_free(dev_counts);
      return fail(OLAP_ERR_LENGTH_MISMATCH, "sparse form needs room for %llu cells, %llu given", total, (unsigned long long)cap);
    }
    uint32_t *dev_idx = nullptr;
    void *dev_val = nullptr;
    const size_t es = olap_dtype_size(s->dtype);
    e = dev_alloc((void **)&dev_idx, total * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc(&dev_val, total * es);
    if (e == hipSuccess) e = hipMemcpy(dev_counts, offsets.data(), n_chunks * sizeof(unsigned long long), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      DISPATCH_DTYPE(s->dtype, e = Launch<T>::compact_write((const T *)s->values, mask, s->size, chunk, n_chunks, s->default_kind == OLAP_DEFAULT_NAN, dev_counts, dev_idx, (T *)dev_val, nullptr));
    }
    if (e == hipSuccess) e = hipMemcpy(host_indexes, dev_idx, total * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(host_values, dev_val, total * es, hipMemcpyDeviceToHost);
    if (dev_idx) dev_free(dev_idx);
    if (dev_val) dev_free(dev_val);
    if (e == hipSuccess && s->seq) {
      // serialize() writes the Map's entries in insertion order (in-memory.js:94-100): permute the ascending lists
      std::vector<uint64_t> keys;
      int rco = order_sorted_keys(s, keys);
      if (rco) {
        dev_free(dev_counts);
        return rco;
      }
      if (keys.size() == total) {
        std::vector<uint32_t> pos(s->size, 0u);  // ascending rank of every set cell
        for (uint64_t j = 0; j < total; ++j) pos[host_indexes[j]] = (uint32_t)j;
        std::vector<unsigned char> vals((size_t)total * es);
        memcpy(vals.data(), host_values, vals.size());
        for (uint64_t j = 0; j < total; ++j) {
          host_indexes[j] = (uint32_t)keys[j];
          memcpy((char *)host_values + j * es, vals.data() + (size_t)pos[keys[j]] * es, es);
        }
      }
    }
  }
  dev_free(dev_counts);
  if (e != hipSuccess) return hip_fail(e, "to_sparse");
  return OLAP_OK;
}

extern "C" int olap_store_from_sparse(olap_store **store, uint64_t size, int dtype, int default_kind,
                                      const uint32_t *host_indexes, const void *host_values, uint64_t n) {
  if (!store) return fail(OLAP_ERR_INVALID_ARGUMENT, "store out-pointer is NULL");
  *store = nullptr;
  if (n && (!host_indexes || !host_values)) return fail(OLAP_ERR_INVALID_ARGUMENT, "indexes/values is NULL");
  for (uint64_t i = 0; i < n; ++i)
    if (host_indexes[i] >= size) return fail(OLAP_ERR_INDEX_RANGE, "cell index %u out of bounds [0, %llu[", host_indexes[i], (unsigned long long)size);
  olap_store *s = nullptr;
  int rc = olap_store_create(&s, size, dtype, default_kind);
  if (rc) return rc;
  if (n) {
    uint32_t *dev_idx = nullptr;
    void *dev_val = nullptr;
    const size_t es = olap_dtype_size(dtype);
    hipError_t e = dev_alloc((void **)&dev_idx, n * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc(&dev_val, n * es);
    if (e == hipSuccess) e = hipMemcpy(dev_idx, host_indexes, n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dev_val, host_values, n * es, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      DISPATCH_DTYPE(dtype, e = Launch<T>::scatter_sparse((T *)s->values, dev_idx, (const T *)dev_val, n, size, nullptr));
    }
    if (e == hipSuccess && s->status) {
      // integer cells under a NaN default: the listed cells are exactly the set ones
      std::vector<int32_t> two(n, OLAP_STATUS_SET);
      int32_t *dev_two = nullptr;
      e = dev_alloc((void **)&dev_two, n * sizeof(int32_t));
      if (e == hipSuccess) e = hipMemcpy(dev_two, two.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = Launch<int32_t>::scatter_sparse(s->status, dev_idx, dev_two, n, size, nullptr);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
      if (dev_two) dev_free(dev_two);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (dev_idx) dev_free(dev_idx);
    if (dev_val) dev_free(dev_val);
    if (e != hipSuccess) {
      olap_store_destroy(s);
      return hip_fail(e, "from_sparse");
    }
    // setValue semantics: a listed default value leaves the cell unset (float cells: implied by the value)
  }
  if ((rc = order_after_from_sparse(s, host_indexes, n))) {
    olap_store_destroy(s);
    return rc;
  }
  *store = s;
  return OLAP_OK;
}

// Integer cells under a NaN default are the one case where the mask carries information the
// values cannot; everywhere else the kernels derive "set" from the value and skip the mask read.
const int32_t *mask_needed(const olap_store *s) { return mask_is_primary(s) ? s->status : nullptr; }

// ---- plan cache of the handle layer ------------------------------------------------------------
// A dashboard issues the same few queries over and over; building a plan costs two or three small
// blocking uploads (~25 us), as much as a small kernel.  Plans made for store handles are kept in
// an LRU keyed by everything that defines them (kind, cell type, default, method, lengths, tables).
namespace {
struct PlanKey {
  std::string bytes;
  PlanKey() {  // a plan's tables and scratch live on the device that was current when it was built
    int dev = 0;
    (void)hipGetDevice(&dev);
    raw(&dev, sizeof dev);
  }
  bool empty() const { return bytes.size() <= sizeof(int); }
  void raw(const void *p, size_t n) { bytes.append((const char *)p, n); }
  void i32(int32_t v) { raw(&v, sizeof v); }
  void u64(uint64_t v) { raw(&v, sizeof v); }
  void u32s(const uint32_t *p, size_t n) {
    u64(n);
    if (n) raw(p, n * sizeof(uint32_t));
  }
  void tables(const uint32_t *const *t, const uint32_t *lens, int ndim) {
    for (int d = 0; d < ndim; ++d) u32s(t[d], lens[d]);
  }
};
struct PlanCache {
  std::mutex mu;
  std::unordered_map<std::string, std::pair<olap_plan *, uint64_t>> map;  // key -> (plan, last use)
  uint64_t tick = 0;
  static constexpr size_t kMax = 128;
  // find() pins the plan: an eviction by another thread between find() and release() only marks it,
  // and the last release() destroys it
  olap_plan *find(const std::string &key) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = map.find(key);
    if (it == map.end()) return nullptr;
    it->second.second = ++tick;
    it->second.first->cache_refs++;
    return it->second.first;
  }
  // inserts a freshly built plan, pinned for its builder
  void insert(const std::string &key, olap_plan *plan) {
    olap_plan *evicted = nullptr;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto dup = map.find(key);
      if (dup != map.end()) {  // another thread built the same plan meanwhile: ours stays private
        plan->cache_refs = 1;
        plan->cache_evicted = true;
        return;
      }
      if (map.size() >= kMax) {
        auto oldest = map.begin();
        for (auto it = map.begin(); it != map.end(); ++it)
          if (it->second.second < oldest->second.second) oldest = it;
        olap_plan *o = oldest->second.first;
        map.erase(oldest);
        if (o->cache_refs > 0) o->cache_evicted = true;
        else evicted = o;
      }
      plan->cache_refs = 1;
      map[key] = {plan, ++tick};
    }
    if (evicted) olap_plan_destroy(evicted);
  }
  void release(olap_plan *plan) {
    bool destroy = false;
    {
      std::lock_guard<std::mutex> lock(mu);
      destroy = --plan->cache_refs == 0 && plan->cache_evicted;
    }
    if (destroy) olap_plan_destroy(plan);
  }
};
PlanCache &plan_cache() {
  static PlanCache *c = new PlanCache();  // leaked on purpose, like the pool
  return *c;
}
}  // namespace

// Runs a (cached) plan into a freshly allocated store.  Everything is enqueued on the null stream
// and NOT waited for: later operations are ordered behind it by the stream and every host read is a
// blocking copy on that stream.  Only a drillDown with distributions is waited for, because its
// data-dependent error (in-memory.js:397-398) must surface from this call.
static int run_to_new_store(olap_plan *plan, const olap_store *in, olap_store **out) {
  olap_store *o = nullptr;
  int rc = store_alloc(&o, olap_plan_out_cells(plan), in->dtype, in->default_kind);
  if (!rc) rc = olap_plan_run(plan, in->values, mask_needed(in), o->values, o->status, nullptr);
  if (!rc && plan->kind == PLAN_DRILLDOWN && plan->dd.dist) {
    hipError_t e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) rc = hip_fail(e, "plan_run");
    if (!rc) rc = olap_plan_status(plan);
  }
  if (rc) {
    olap_store_destroy(o);
    return rc;
  }
  *out = o;
  return OLAP_OK;
}

static int check_store_cells(const olap_store *s, const olap_plan *plan) {
  if (s->size != olap_plan_in_cells(plan))
    return fail(OLAP_ERR_LENGTH_MISMATCH, "store holds %llu cells but the dimensions describe %llu", (unsigned long long)s->size, (unsigned long long)olap_plan_in_cells(plan));
  return OLAP_OK;
}

static bool bad_dims(int ndim, const void *a, const void *b) { return ndim < 0 || ndim > OLAP_MAX_DIMS || (ndim > 0 && (!a || !b)); }

extern "C" int olap_store_drillup(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                                  const uint32_t *new_len, const uint32_t *const *maps, int method) {
  OnStoreDevice on_device__(s);
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->track_order && !bad_dims(ndim, old_len, new_len) && (ndim == 0 || maps)) return order_drillup(s, out, ndim, old_len, new_len, maps, method);
  return store_drillup_plain(s, out, ndim, old_len, new_len, maps, method);
}

int store_drillup_plain(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                        const uint32_t *const *maps, int method) {
  *out = nullptr;
  olap_plan *plan = nullptr;
  PlanKey key;
  const bool keyable = !bad_dims(ndim, old_len, new_len) && (ndim == 0 || maps);
  if (keyable) {
    bool ok = true;
    for (int d = 0; d < ndim; ++d) ok = ok && (old_len[d] == 0 || maps[d]);
    if (ok) {
      key.i32('U');
      key.i32(s->dtype), key.i32(s->default_kind), key.i32(method), key.i32(ndim);
      key.u32s(old_len, ndim), key.u32s(new_len, ndim);
      key.tables(maps, old_len, ndim);
      plan = plan_cache().find(key.bytes);
    }
  }
  if (!plan) {
    int rc = olap_drillup_plan(&plan, s->dtype, s->default_kind, method, ndim, old_len, new_len, maps);
    if (rc) return rc;
    if (!key.empty()) plan_cache().insert(key.bytes, plan);
    else {
      rc = check_store_cells(s, plan);
      if (!rc) rc = run_to_new_store(plan, s, out);
      olap_plan_destroy(plan);
      return rc;
    }
  }
  int rc = check_store_cells(s, plan);
  if (!rc) rc = run_to_new_store(plan, s, out);
  plan_cache().release(plan);
  return rc;
}

// Cube.drillUp over several stored measures that share cell type, default and rule (src/cube.js:1012-1020 calls the
// store once per measure): one plan, one launch (olap_plan_run_batch).  Stores that differ in type or default, or that
// track their insertion order, take the single-store path one by one — same results, n launches.
extern "C" int olap_store_drillup_batch(int n, const olap_store *const *stores, olap_store **out, int ndim, const uint32_t *old_len,
                                        const uint32_t *new_len, const uint32_t *const *maps, int method) {
  if (n < 0 || (n > 0 && (!stores || !out))) return fail(OLAP_ERR_INVALID_ARGUMENT, "store list is NULL");
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  for (int i = 0; i < n; ++i)
    if (!stores[i]) return fail(OLAP_ERR_INVALID_ARGUMENT, "store %d of the batch is NULL", i);
  bool same = n > 1;
  for (int i = 0; i < n && same; ++i)
    same = stores[i]->dtype == stores[0]->dtype && stores[i]->default_kind == stores[0]->default_kind && stores[i]->size == stores[0]->size &&
           stores[i]->device == stores[0]->device && !stores[i]->track_order;
  auto undo = [&](int rc) {
    for (int i = 0; i < n; ++i) {
      if (out[i]) olap_store_destroy(out[i]);
      out[i] = nullptr;
    }
    return rc;
  };
  if (!same || bad_dims(ndim, old_len, new_len) || (ndim > 0 && !maps)) {
    for (int i = 0; i < n; ++i) {
      const int rc = olap_store_drillup(stores[i], &out[i], ndim, old_len, new_len, maps, method);
      if (rc) return undo(rc);
    }
    return OLAP_OK;
  }
  const olap_store *s0 = stores[0];
  OnStoreDevice on_device__(s0);
  olap_plan *plan = nullptr;
  PlanKey key;
  bool keyable = true;
  for (int d = 0; d < ndim; ++d) keyable = keyable && (old_len[d] == 0 || maps[d]);
  if (keyable) {
    key.i32('U');
    key.i32(s0->dtype), key.i32(s0->default_kind), key.i32(method), key.i32(ndim);
    key.u32s(old_len, ndim), key.u32s(new_len, ndim);
    key.tables(maps, old_len, ndim);
    plan = plan_cache().find(key.bytes);
  }
  bool cached = plan != nullptr;
  if (!plan) {
    int rc = olap_drillup_plan(&plan, s0->dtype, s0->default_kind, method, ndim, old_len, new_len, maps);
    if (rc) return rc;
    if (keyable) {
      plan_cache().insert(key.bytes, plan);
      cached = true;
    }
  }
  int rc = check_store_cells(s0, plan);
  std::vector<const void *> in_v(n);
  std::vector<const int32_t *> in_s(n);
  std::vector<void *> out_v(n);
  std::vector<int32_t *> out_s(n);
  for (int i = 0; i < n && !rc; ++i) {
    rc = store_alloc(&out[i], olap_plan_out_cells(plan), s0->dtype, s0->default_kind);
    if (rc) break;
    in_v[i] = stores[i]->values;
    in_s[i] = mask_needed(stores[i]);
    out_v[i] = out[i]->values;
    out_s[i] = out[i]->status;
  }
  if (!rc) rc = olap_plan_run_batch(plan, n, in_v.data(), in_s.data(), out_v.data(), out_s.data(), nullptr);
  if (cached) plan_cache().release(plan);
  else olap_plan_destroy(plan);
  return rc ? undo(rc) : OLAP_OK;
}

// Cube.drillUp over ALL stored measures of a cube, each with its own rule for the rolled-up dimension (methods[i]):
// measures that share cell type, default and size go out together — one mixed-rule launch when the roll-up runs in
// the row regime (drillup_rows_mixed_kernel), one launch per rule otherwise — the rest one by one.
extern "C" int olap_store_drillup_multi(int n, const olap_store *const *stores, const int *methods, olap_store **out, int ndim,
                                        const uint32_t *old_len, const uint32_t *new_len, const uint32_t *const *maps) {
  if (n < 0 || (n > 0 && (!stores || !out || !methods))) return fail(OLAP_ERR_INVALID_ARGUMENT, "store / method list is NULL");
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  for (int i = 0; i < n; ++i) {
    if (!stores[i]) return fail(OLAP_ERR_INVALID_ARGUMENT, "store %d of the batch is NULL", i);
    if (methods[i] < OLAP_SUM || methods[i] > OLAP_PRODUCT) return fail(OLAP_ERR_UNSUPPORTED_METHOD, "Unsupported aggregation method: %d", methods[i]);
  }
  auto undo = [&](int rc) {
    for (int i = 0; i < n; ++i) {
      if (out[i]) olap_store_destroy(out[i]);
      out[i] = nullptr;
    }
    return rc;
  };
  bool same = n > 1, one_rule = true;
  for (int i = 0; i < n; ++i) {
    same = same && stores[i]->dtype == stores[0]->dtype && stores[i]->default_kind == stores[0]->default_kind && stores[i]->size == stores[0]->size &&
           stores[i]->device == stores[0]->device && !stores[i]->track_order;
    one_rule = one_rule && methods[i] == methods[0];
  }
  bool keyable = !bad_dims(ndim, old_len, new_len) && (ndim == 0 || maps);
  for (int d = 0; keyable && d < ndim; ++d) keyable = old_len[d] == 0 || maps[d];
  if (same && !one_rule && keyable) {
    const olap_store *s0 = stores[0];
    OnStoreDevice on_device__(s0);
    PlanKey key;
    key.i32('U');
    key.i32(s0->dtype), key.i32(s0->default_kind), key.i32(methods[0]), key.i32(ndim);
    key.u32s(old_len, ndim), key.u32s(new_len, ndim);
    key.tables(maps, old_len, ndim);
    olap_plan *plan = plan_cache().find(key.bytes);
    if (!plan) {
      int rc = olap_drillup_plan(&plan, s0->dtype, s0->default_kind, methods[0], ndim, old_len, new_len, maps);
      if (rc) return rc;
      plan_cache().insert(key.bytes, plan);
    }
    int rc = check_store_cells(s0, plan);
    std::vector<const void *> in_v(n);
    std::vector<const int32_t *> in_s(n);
    std::vector<void *> out_v(n);
    std::vector<int32_t *> out_s(n);
    for (int i = 0; i < n && !rc; ++i) {
      rc = store_alloc(&out[i], olap_plan_out_cells(plan), s0->dtype, s0->default_kind);
      if (rc) break;
      in_v[i] = stores[i]->values;
      in_s[i] = mask_needed(stores[i]);
      out_v[i] = out[i]->values;
      out_s[i] = out[i]->status;
    }
    if (!rc) rc = plan_run_mixed(plan, n, methods, in_v.data(), in_s.data(), out_v.data(), out_s.data(), nullptr);
    plan_cache().release(plan);
    if (rc == OLAP_OK) return OLAP_OK;
    undo(rc);
    if (rc != OLAP_MIXED_NOT_APPLICABLE) return rc;
  }
  // rule by rule: the measures of one rule together (olap_store_drillup_batch groups further by cell type)
  std::vector<char> done(n, 0);
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    std::vector<int> members;
    for (int j = i; j < n; ++j)
      if (!done[j] && methods[j] == methods[i] && stores[j]->dtype == stores[i]->dtype && stores[j]->default_kind == stores[i]->default_kind &&
          stores[j]->size == stores[i]->size) {
        members.push_back(j);
        done[j] = 1;
      }
    std::vector<const olap_store *> in(members.size());
    std::vector<olap_store *> res(members.size(), nullptr);
    for (size_t k = 0; k < members.size(); ++k) in[k] = stores[members[k]];
    const int rc = olap_store_drillup_batch((int)members.size(), in.data(), res.data(), ndim, old_len, new_len, maps, methods[i]);
    if (rc) return undo(rc);
    for (size_t k = 0; k < members.size(); ++k) out[members[k]] = res[k];
  }
  return OLAP_OK;
}

static int store_drilldown_plain(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                                 const uint32_t *new_len, const uint32_t *const *maps, int method,
                                 const double *distributions, uint64_t n_dist) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  olap_plan *plan = nullptr;
  if (!distributions && !bad_dims(ndim, old_len, new_len) && (ndim == 0 || maps)) {
    // cached like the other operations (a plan costs a table upload and a few allocations: ~100 us against a 70 us kernel);
    // plans with distributions carry a per-run error word and their weights: built per call
    bool ok = true;
    for (int d = 0; d < ndim; ++d) ok = ok && (new_len[d] == 0 || maps[d]);
    if (ok) {
      PlanKey key;
      key.i32('W');
      key.i32(s->dtype), key.i32(s->default_kind), key.i32(method), key.i32(ndim);
      key.u32s(old_len, ndim), key.u32s(new_len, ndim);
      key.tables(maps, new_len, ndim);
      plan = plan_cache().find(key.bytes);
      bool cached = true;
      if (!plan) {
        int rc = olap_drilldown_plan(&plan, s->dtype, s->default_kind, method, ndim, old_len, new_len, maps, nullptr, 0);
        if (rc) return rc;
        cached = plan->dev_tmp == nullptr;  // (the two-pass form keeps a buffer of quotients as large as the parents: not held on to)
        if (cached) plan_cache().insert(key.bytes, plan);
      }
      int rc = check_store_cells(s, plan);
      if (!rc) rc = run_to_new_store(plan, s, out);
      if (cached) plan_cache().release(plan);
      else olap_plan_destroy(plan);
      return rc;
    }
  }
  int rc = olap_drilldown_plan(&plan, s->dtype, s->default_kind, method, ndim, old_len, new_len, maps, distributions, n_dist);
  if (rc) return rc;
  rc = check_store_cells(s, plan);
  if (!rc) rc = run_to_new_store(plan, s, out);
  olap_plan_destroy(plan);
  return rc;
}

extern "C" int olap_store_drilldown(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                                    const uint32_t *new_len, const uint32_t *const *maps, int method,
                                    const double *distributions, uint64_t n_dist) {
  OnStoreDevice on_device__(s);
  int rc = store_drilldown_plain(s, out, ndim, old_len, new_len, maps, method, distributions, n_dist);
  if (!rc && s->track_order) rc = order_after_drilldown(s, *out);
  return rc;
}

static int store_dice_plain(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                            const uint32_t *new_len, const int32_t *const *sel) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  olap_plan *plan = nullptr;
  PlanKey key;
  if (!bad_dims(ndim, old_len, new_len) && (ndim == 0 || sel)) {
    bool ok = true;
    for (int d = 0; d < ndim; ++d) ok = ok && (new_len[d] == 0 || sel[d]);
    if (ok) {
      key.i32('D');
      key.i32(s->dtype), key.i32(s->default_kind), key.i32(ndim);
      key.u32s(old_len, ndim), key.u32s(new_len, ndim);
      key.tables((const uint32_t *const *)sel, new_len, ndim);
      plan = plan_cache().find(key.bytes);
    }
  }
  if (!plan) {
    int rc = olap_dice_plan(&plan, s->dtype, s->default_kind, ndim, old_len, new_len, sel);
    if (rc) return rc;
    if (!key.empty()) plan_cache().insert(key.bytes, plan);
    else {
      rc = check_store_cells(s, plan);
      if (!rc) rc = run_to_new_store(plan, s, out);
      olap_plan_destroy(plan);
      return rc;
    }
  }
  int rc = check_store_cells(s, plan);
  if (!rc) rc = run_to_new_store(plan, s, out);
  plan_cache().release(plan);
  return rc;
}

// an operation on a tracked store failed after its result was allocated
static int drop_result(olap_store **out, int rc) {
  if (rc && out && *out) {
    olap_store_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

extern "C" int olap_store_dice(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                               const uint32_t *new_len, const int32_t *const *sel) {
  OnStoreDevice on_device__(s);
  int rc = store_dice_plain(s, out, ndim, old_len, new_len, sel);
  if (!rc && s->track_order) rc = drop_result(out, order_after_dice(s, *out, ndim, old_len, new_len, sel));
  return rc;
}

extern "C" int olap_store_dice_drillup(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                                       const uint32_t *mid_len, const uint32_t *new_len, const int32_t *const *sel,
                                       const uint32_t *const *maps, int method) {
  OnStoreDevice on_device__(s);
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  if (s->track_order) {  // the order of the diced intermediate matters: the two operations, one after the other
    olap_store *mid = nullptr;
    int rc = olap_store_dice(s, &mid, ndim, old_len, mid_len, sel);
    if (rc) return rc;
    rc = olap_store_drillup(mid, out, ndim, mid_len, new_len, maps, method);
    olap_store_destroy(mid);
    return rc;
  }
  olap_plan *plan = nullptr;
  PlanKey key;
  if (!bad_dims(ndim, old_len, mid_len) && !bad_dims(ndim, mid_len, new_len) && (ndim == 0 || (sel && maps))) {
    bool ok = true;
    for (int d = 0; d < ndim; ++d) ok = ok && (mid_len[d] == 0 || (sel[d] && maps[d]));
    if (ok) {
      key.i32('F');
      key.i32(s->dtype), key.i32(s->default_kind), key.i32(method), key.i32(ndim);
      key.u32s(old_len, ndim), key.u32s(mid_len, ndim), key.u32s(new_len, ndim);
      key.tables((const uint32_t *const *)sel, mid_len, ndim);
      key.tables(maps, mid_len, ndim);
      plan = plan_cache().find(key.bytes);
    }
  }
  if (!plan) {
    int rc = olap_dice_drillup_plan(&plan, s->dtype, s->default_kind, method, ndim, old_len, mid_len, new_len, sel, maps);
    if (rc) return rc;
    if (!key.empty()) plan_cache().insert(key.bytes, plan);
    else {
      rc = check_store_cells(s, plan);
      if (!rc) rc = run_to_new_store(plan, s, out);
      olap_plan_destroy(plan);
      return rc;
    }
  }
  int rc = check_store_cells(s, plan);
  if (!rc) rc = run_to_new_store(plan, s, out);
  plan_cache().release(plan);
  return rc;
}

static int store_reorder_plain(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const int32_t *perm) {
  if (!s || !out) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  *out = nullptr;
  olap_plan *plan = nullptr;
  PlanKey key;
  if (!bad_dims(ndim, old_len, perm)) {
    key.i32('R');
    key.i32(s->dtype), key.i32(s->default_kind), key.i32(ndim);
    key.u32s(old_len, ndim), key.u32s((const uint32_t *)perm, ndim);
    plan = plan_cache().find(key.bytes);
  }
  if (!plan) {
    int rc = olap_reorder_plan(&plan, s->dtype, s->default_kind, ndim, old_len, perm);
    if (rc) return rc;
    if (!key.empty()) plan_cache().insert(key.bytes, plan);
    else {
      rc = check_store_cells(s, plan);
      if (!rc) rc = run_to_new_store(plan, s, out);
      olap_plan_destroy(plan);
      return rc;
    }
  }
  int rc = check_store_cells(s, plan);
  if (!rc) rc = run_to_new_store(plan, s, out);
  plan_cache().release(plan);
  return rc;
}

extern "C" int olap_store_reorder(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len,
                                  const int32_t *perm) {
  OnStoreDevice on_device__(s);
  int rc = store_reorder_plain(s, out, ndim, old_len, perm);
  if (!rc && s->track_order) rc = drop_result(out, order_after_reorder(s, *out, ndim, old_len, perm));
  return rc;
}

static int store_load_plain(olap_store *s, const olap_store *other, int ndim, const uint32_t *my_len, const uint32_t *his_len,
                            const int32_t *const *his_to_mine);

extern "C" int olap_store_load(olap_store *s, const olap_store *other, int ndim, const uint32_t *my_len,
                               const uint32_t *his_len, const int32_t *const *his_to_mine) {
  OnStoreDevice on_device__(s);
  if (!s || !other) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->device != other->device)
    return fail(OLAP_ERR_INVALID_ARGUMENT, "load: the stores live on different devices (%d and %d)", s->device, other->device);
  int rc = order_before_load(s);
  if (!rc) rc = store_load_plain(s, other, ndim, my_len, his_len, his_to_mine);
  if (!rc) rc = order_after_load(s, other, ndim, my_len, his_len, his_to_mine);
  return rc;
}

static int store_load_plain(olap_store *s, const olap_store *other, int ndim, const uint32_t *my_len, const uint32_t *his_len,
                            const int32_t *const *his_to_mine) {
  if (!s || !other) return fail(OLAP_ERR_INVALID_ARGUMENT, "store is NULL");
  if (s->dtype != other->dtype) return fail(OLAP_ERR_INVALID_TYPE, "load: stores have different cell types");
  olap_plan *plan = nullptr;
  int rc = olap_load_plan(&plan, s->dtype, s->default_kind, other->default_kind, ndim, my_len, his_len, his_to_mine);
  if (rc) return rc;
  if (other->size != olap_plan_in_cells(plan) || s->size != olap_plan_out_cells(plan)) {
    olap_plan_destroy(plan);
    return fail(OLAP_ERR_LENGTH_MISMATCH, "load: store sizes do not match the dimensions");
  }
  drop_lazy_status(s);
  rc = olap_plan_run(plan, other->values, mask_needed(other), s->values, s->status, nullptr);
  olap_plan_destroy(plan);  // waits for the launch (its tables go back to the pool)
  return rc;
}
