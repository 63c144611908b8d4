// olap_napi.cc — thin N-API binding of include/olap_hip.h for the Node.js host.
//
// Exposes one class, `Store`, wrapping an `olap_store*` (a device-resident measure, the
// replacement of the reference's InMemoryStore, /root/reference/src/store/in-memory.js), plus a few
// module functions.  No arithmetic happens here: every method forwards to the C ABI and turns a
// non-zero return into a JS Error carrying olap_last_error() (the reference's own messages).
// Handles are freed by the GC finalizer (the reference API has no explicit free).
#include <node_api.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "olap_hip.h"

#define NAPI_OK(call)                                            \
  do {                                                           \
    if ((call) != napi_ok) {                                     \
      napi_throw_error(env, nullptr, "N-API call failed: " #call); \
      return nullptr;                                            \
    }                                                            \
  } while (0)

static napi_ref g_store_ctor = nullptr;

static napi_value throw_olap(napi_env env, int code) {
  char codebuf[16];
  snprintf(codebuf, sizeof codebuf, "OLAP%d", code);
  napi_throw_error(env, codebuf, olap_last_error());
  return nullptr;
}

// Device memory and the JS garbage collector.  Telling V8 about every result store
// (napi_adjust_external_memory) makes it run incremental mark-sweep steps for each 40 MB result: a
// drillUp then costs ~9 ms of GC around a 70 us kernel.  Telling it nothing would let dead stores
// pile up until the JS heap itself needs collecting.  So V8 is left alone while the stores held by
// live wrappers stay under a budget (OLAP_NAPI_GC_BYTES, default 16 GiB of the 288 GB; keep it
// below the pool's OLAP_POOL_BYTES so that what a collection frees is recycled, not hipFree'd).  When the
// budget is crossed, ONE large external-memory report makes V8 do a full collection right there
// (its hard limit for external memory is far below the amount reported), the report is taken back,
// and the wrappers found dead are swept (below).  The next such collection is not asked for before
// another quarter of the budget has been allocated, so live data above the budget does not thrash.
static int64_t g_held = 0;
static int64_t g_next_gc_at = 0;
static int64_t gc_budget() {
  static int64_t b = -1;
  if (b < 0) {
    const char *e = getenv("OLAP_NAPI_GC_BYTES");
    b = e ? (int64_t)strtoll(e, nullptr, 10) : ((int64_t)16 << 30);
  }
  return b;
}
static void sweep_boxes(napi_env env);
static void account(napi_env env, int64_t delta) {
  g_held += delta;
  if (delta > 0 && g_held > gc_budget() && g_held >= g_next_gc_at) {
    int64_t total;
    const int64_t nudge = (int64_t)8 << 30;
    napi_adjust_external_memory(env, nudge, &total);   // V8 collects synchronously in here
    napi_adjust_external_memory(env, -nudge, &total);
    sweep_boxes(env);
    g_next_gc_at = g_held + gc_budget() / 4;
  }
}

// Node runs N-API finalizers on a later event-loop turn, never inside a synchronous loop (even an
// explicit gc() does not run them), so a loop of 3 000 queries would hold 3 000 result stores.  Each
// wrapper therefore keeps its store in a small box next to a weak reference: once the budget is
// exceeded, creating a store first sweeps the boxes and frees the device memory of every wrapper the
// collector has already found dead (the weak reference is cleared during the GC itself); the late
// finalizer then finds an empty box.
struct StoreBox {
  olap_store *store;
  napi_ref ref;
  bool finalized;
};
static std::vector<StoreBox *> g_boxes;

static void release_box(napi_env env, StoreBox *box) {
  if (!box->store) return;
  account(env, -(int64_t)olap_store_byte_length(box->store));
  olap_store_destroy(box->store);
  box->store = nullptr;
}

static void finalize_store(napi_env env, void *data, void *) {
  StoreBox *box = (StoreBox *)data;
  release_box(env, box);
  if (box->ref) napi_delete_reference(env, box->ref);
  box->ref = nullptr;
  box->finalized = true;  // the box itself is deleted by the next sweep
}

static void sweep_boxes(napi_env env) {
  size_t keep = 0;
  for (size_t i = 0; i < g_boxes.size(); ++i) {
    StoreBox *box = g_boxes[i];
    if (box->finalized) {
      delete box;
      continue;
    }
    if (box->store && box->ref) {
      napi_value alive = nullptr;
      if (napi_get_reference_value(env, box->ref, &alive) == napi_ok && alive == nullptr) release_box(env, box);
    }
    g_boxes[keep++] = box;
  }
  g_boxes.resize(keep);
}

static olap_store *unwrap(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_unwrap(env, v, &p) != napi_ok || !p || !((StoreBox *)p)->store) {
    napi_throw_type_error(env, nullptr, "not a Store");
    return nullptr;
  }
  return ((StoreBox *)p)->store;
}

static napi_value wrap_new_store(napi_env env, olap_store *s) {
  napi_value ctor, obj, ext;
  NAPI_OK(napi_get_reference_value(env, g_store_ctor, &ctor));
  NAPI_OK(napi_create_external(env, s, nullptr, nullptr, &ext));
  if (napi_new_instance(env, ctor, 1, &ext, &obj) != napi_ok) {
    olap_store_destroy(s);
    return nullptr;
  }
  return obj;
}

// new Store(size, dtypeCode, defaultKind)  |  new Store(external) [internal]
static napi_value StoreNew(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3], self;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
  napi_valuetype t;
  NAPI_OK(napi_typeof(env, argv[0], &t));
  olap_store *s = nullptr;
  bool adopted = false;  // the handle came from wrap_new_store(), which destroys it itself when this constructor fails
  if (argc == 1 && t == napi_external) {
    void *p;
    NAPI_OK(napi_get_value_external(env, argv[0], &p));
    s = (olap_store *)p;
    adopted = true;
  } else {
    double size = 0;
    int32_t dtype = 0, def = 0;
    NAPI_OK(napi_get_value_double(env, argv[0], &size));
    NAPI_OK(napi_get_value_int32(env, argv[1], &dtype));
    NAPI_OK(napi_get_value_int32(env, argv[2], &def));
    int rc = olap_store_create(&s, (uint64_t)size, dtype, def);
    if (rc) return throw_olap(env, rc);
  }
  static unsigned since_sweep = 0;
  if (++since_sweep >= 1024) {  // housekeeping: drop the boxes of finalized wrappers
    sweep_boxes(env);
    since_sweep = 0;
  }
  StoreBox *box = new StoreBox{s, nullptr, false};
  if (napi_wrap(env, self, box, finalize_store, nullptr, &box->ref) != napi_ok) {
    if (!adopted) olap_store_destroy(s);
    delete box;
    napi_throw_error(env, nullptr, "napi_wrap failed");
    return nullptr;
  }
  g_boxes.push_back(box);
  account(env, (int64_t)olap_store_byte_length(s));
  return self;
}

#define STORE_METHOD_PROLOGUE(MAXARGS)                                     \
  size_t argc = MAXARGS;                                                   \
  napi_value argv[MAXARGS > 0 ? MAXARGS : 1], self;                        \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));       \
  olap_store *s = unwrap(env, self);                                       \
  if (!s) return nullptr;

static napi_value num(napi_env env, double v) {
  napi_value r;
  napi_create_double(env, v, &r);
  return r;
}

static napi_value StoreSize(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  return num(env, (double)olap_store_size(s));
}
static napi_value StoreByteLength(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  return num(env, (double)olap_store_byte_length(s));
}
static napi_value StoreDtype(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  return num(env, olap_store_dtype(s));
}
static napi_value StoreDefault(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  return num(env, olap_store_default(s));
}

static napi_typedarray_type ta_of(int dtype) {
  switch (dtype) {
    case OLAP_INT32: return napi_int32_array;
    case OLAP_UINT32: return napi_uint32_array;
    case OLAP_FLOAT32: return napi_float32_array;
    default: return napi_float64_array;
  }
}

// setData(typedArray): Float64Array = JS numbers (converted on the device like a TypedArray
// store would); otherwise a typed array of the store's own element type
static napi_value StoreSetData(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(1)
  bool is_ta = false;
  NAPI_OK(napi_is_typedarray(env, argv[0], &is_ta));
  if (!is_ta) {
    napi_throw_type_error(env, nullptr, "setData expects a TypedArray");
    return nullptr;
  }
  napi_typedarray_type type;
  size_t len;
  void *data;
  NAPI_OK(napi_get_typedarray_info(env, argv[0], &type, &len, &data, nullptr, nullptr));
  int rc;
  if (type == napi_float64_array && olap_store_dtype(s) != OLAP_FLOAT64) rc = olap_store_set_data_f64(s, (const double *)data, len);
  else if (type == ta_of(olap_store_dtype(s))) rc = olap_store_set_data(s, data, len);
  else if (olap_store_dtype(s) == OLAP_FLOAT64 && (type == napi_int32_array || type == napi_uint32_array || type == napi_float32_array)) {
    // an integer (or Float32) typed array into Float64 cells — what the Node host keeps integer measures in: widened
    // here, in one pass, instead of element by element in JavaScript
    std::vector<double> wide(len);
    if (type == napi_int32_array) for (size_t i = 0; i < len; ++i) wide[i] = (double)((const int32_t *)data)[i];
    else if (type == napi_uint32_array) for (size_t i = 0; i < len; ++i) wide[i] = (double)((const uint32_t *)data)[i];
    else for (size_t i = 0; i < len; ++i) wide[i] = (double)((const float *)data)[i];
    rc = olap_store_set_data(s, wide.data(), len);
  } else {
    napi_throw_type_error(env, nullptr, "setData: typed array does not match the store's element type");
    return nullptr;
  }
  if (rc) return throw_olap(env, rc);
  return nullptr;
}

static napi_value make_ta(napi_env env, napi_typedarray_type type, size_t elem, size_t n, void **data) {
  napi_value ab, ta;
  NAPI_OK(napi_create_arraybuffer(env, n * elem, data, &ab));
  NAPI_OK(napi_create_typedarray(env, type, n, ab, 0, &ta));
  return ta;
}

static napi_value StoreGetData(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  void *data;
  const int dt = olap_store_dtype(s);
  napi_value ta = make_ta(env, ta_of(dt), olap_dtype_size(dt), olap_store_size(s), &data);
  if (!ta) return nullptr;
  int rc = olap_store_get_data(s, data);
  if (rc) return throw_olap(env, rc);
  return ta;
}

static napi_value StoreGetDataF64(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  void *data;
  napi_value ta = make_ta(env, napi_float64_array, 8, olap_store_size(s), &data);
  if (!ta) return nullptr;
  int rc = olap_store_get_data_f64(s, (double *)data);
  if (rc) return throw_olap(env, rc);
  return ta;
}

static napi_value StoreGetStatus(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  void *data;
  napi_value ta = make_ta(env, napi_int32_array, 4, olap_store_size(s), &data);
  if (!ta) return nullptr;
  int rc = olap_store_get_status(s, (int32_t *)data);
  if (rc) return throw_olap(env, rc);
  return ta;
}

// ascending indices of the set cells, as a Float64Array (indices < 2^53)
static napi_value StoreGetKeys(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  uint64_t n = 0;
  int rc = olap_store_get_keys(s, nullptr, 0, &n);
  if (rc) return throw_olap(env, rc);
  std::vector<uint64_t> keys(n ? n : 1);
  rc = olap_store_get_keys(s, keys.data(), n, &n);
  if (rc) return throw_olap(env, rc);
  void *data;
  napi_value ta = make_ta(env, napi_float64_array, 8, n, &data);
  if (!ta) return nullptr;
  for (uint64_t i = 0; i < n; ++i) ((double *)data)[i] = (double)keys[i];
  return ta;
}

static napi_value StoreCountSet(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  uint64_t n = 0;
  int rc = olap_store_count_set(s, &n);
  if (rc) return throw_olap(env, rc);
  return num(env, (double)n);
}

// getValue(i) -> number, or undefined when the cell is unset (Map.get of an absent key)
static napi_value StoreGetValue(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(1)
  double idx = 0;
  NAPI_OK(napi_get_value_double(env, argv[0], &idx));
  napi_value undef;
  napi_get_undefined(env, &undef);
  if (!(idx >= 0) || idx != std::floor(idx)) return undef;
  double v = 0;
  int is_set = 0;
  int rc = olap_store_get_value(s, (uint64_t)idx, &v, &is_set);
  if (rc) return throw_olap(env, rc);
  return is_set ? num(env, v) : undef;
}

// setValue(i, v): undefined / null unset the cell (in-memory.js:122-133)
static napi_value StoreSetValue(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(2)
  double idx = 0, v = 0;
  NAPI_OK(napi_get_value_double(env, argv[0], &idx));
  napi_valuetype t = napi_undefined;
  if (argc > 1) NAPI_OK(napi_typeof(env, argv[1], &t));
  int is_null = (t == napi_undefined || t == napi_null);
  if (!is_null) {
    napi_value coerced;
    NAPI_OK(napi_coerce_to_number(env, argv[1], &coerced));
    NAPI_OK(napi_get_value_double(env, coerced, &v));
  }
  int rc = olap_store_set_value(s, (uint64_t)idx, v, is_null);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}

static napi_value StoreFill(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(1)
  double v = 0;
  napi_value coerced;
  NAPI_OK(napi_coerce_to_number(env, argv[0], &coerced));
  NAPI_OK(napi_get_value_double(env, coerced, &v));
  int rc = olap_store_fill(s, v);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}

static napi_value StoreTotal(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  double t = 0;
  int rc = olap_store_total(s, &t);
  if (rc) return throw_olap(env, rc);
  return num(env, t);
}

// trackOrder(on = true): keep the reference Map's insertion order (olap_store_track_order)
static napi_value StoreTrackOrder(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(1)
  bool on = true;
  if (argc > 0) {
    napi_valuetype t;
    NAPI_OK(napi_typeof(env, argv[0], &t));
    if (t == napi_boolean) NAPI_OK(napi_get_value_bool(env, argv[0], &on));
  }
  int rc = olap_store_track_order(s, on ? 1 : 0);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}
// orderTracked: 0 = not tracked, 1 = tracked and still ascending, 2 = tracked with an explicit order
static napi_value StoreOrderTracked(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  return num(env, olap_store_order_tracked(s));
}

static napi_value StoreClone(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  olap_store *c = nullptr;
  int rc = olap_store_clone(s, &c);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, c);
}

// totals(lens: Uint32Array, methods: Int32Array) -> Float64Array of the extended cube (olap_store_totals)
static bool get_u32_vec(napi_env env, napi_value v, std::vector<uint32_t> &out);
static napi_value StoreTotals(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(2)
  std::vector<uint32_t> lens, methods;
  if (argc < 2 || !get_u32_vec(env, argv[0], lens) || !get_u32_vec(env, argv[1], methods) || lens.size() != methods.size()) {
    napi_throw_type_error(env, nullptr, "totals(lens: Uint32Array, methods: Int32Array)");
    return nullptr;
  }
  double n = 1;
  for (uint32_t l : lens) n *= (double)l + 1;
  if (n > 4.0e9) {
    napi_throw_range_error(env, nullptr, "totals: extended cube too large");
    return nullptr;
  }
  void *data;
  napi_value ta = make_ta(env, napi_float64_array, 8, (size_t)n, &data);
  if (!ta) return nullptr;
  static const uint32_t none = 0;
  int rc = olap_store_totals(s, (int)lens.size(), lens.empty() ? &none : lens.data(), lens.empty() ? (const int *)&none : (const int *)methods.data(),
                             (double *)data, nullptr, nullptr, nullptr);
  if (rc) return throw_olap(env, rc);
  return ta;
}

// toSparse() -> { indexes: Uint32Array, values: TypedArray } of the set cells (ascending)
static napi_value StoreToSparse(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(0)
  uint64_t n = 0;
  int rc = olap_store_to_sparse(s, nullptr, nullptr, 0, &n);
  if (rc) return throw_olap(env, rc);
  void *idx, *vals;
  const int dt = olap_store_dtype(s);
  napi_value ta_idx = make_ta(env, napi_uint32_array, 4, n, &idx);
  napi_value ta_val = make_ta(env, ta_of(dt), olap_dtype_size(dt), n, &vals);
  if (!ta_idx || !ta_val) return nullptr;
  if (n) {
    rc = olap_store_to_sparse(s, (uint32_t *)idx, vals, n, &n);
    if (rc) return throw_olap(env, rc);
  }
  napi_value out;
  NAPI_OK(napi_create_object(env, &out));
  NAPI_OK(napi_set_named_property(env, out, "indexes", ta_idx));
  NAPI_OK(napi_set_named_property(env, out, "values", ta_val));
  return out;
}

// Store.fromSparse(size, dtypeCode, defaultKind, indexes: Uint32Array, values: TypedArray)
static napi_value StoreFromSparse(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (argc < 5) return nullptr;
  double size = 0;
  int32_t dtype = 0, def = 0;
  NAPI_OK(napi_get_value_double(env, argv[0], &size));
  NAPI_OK(napi_get_value_int32(env, argv[1], &dtype));
  NAPI_OK(napi_get_value_int32(env, argv[2], &def));
  napi_typedarray_type t_idx, t_val;
  size_t n_idx = 0, n_val = 0;
  void *idx = nullptr, *vals = nullptr;
  if (napi_get_typedarray_info(env, argv[3], &t_idx, &n_idx, &idx, nullptr, nullptr) != napi_ok || t_idx != napi_uint32_array ||
      napi_get_typedarray_info(env, argv[4], &t_val, &n_val, &vals, nullptr, nullptr) != napi_ok || n_idx != n_val ||
      t_val != ta_of(dtype)) {
    napi_throw_type_error(env, nullptr, "fromSparse(size, dtype, default, indexes: Uint32Array, values: TypedArray of the store type)");
    return nullptr;
  }
  olap_store *s = nullptr;
  static const uint32_t none = 0;
  int rc = olap_store_from_sparse(&s, (uint64_t)size, dtype, def, idx ? (const uint32_t *)idx : &none, vals ? vals : (const void *)&none, n_idx);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, s);
}

// ---- argument decoding for the bulk operations -----------------------------------------------
static bool get_u32_vec(napi_env env, napi_value v, std::vector<uint32_t> &out) {
  bool is_ta = false;
  if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return false;
  napi_typedarray_type type;
  size_t len;
  void *data;
  if (napi_get_typedarray_info(env, v, &type, &len, &data, nullptr, nullptr) != napi_ok) return false;
  if (type != napi_uint32_array && type != napi_int32_array) return false;
  out.assign((uint32_t *)data, (uint32_t *)data + len);
  return true;
}

// array of Uint32Array / Int32Array, one per dimension
static bool get_tables(napi_env env, napi_value v, std::vector<std::vector<uint32_t>> &out) {
  bool is_arr = false;
  if (napi_is_array(env, v, &is_arr) != napi_ok || !is_arr) return false;
  uint32_t n = 0;
  napi_get_array_length(env, v, &n);
  out.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    napi_value e;
    if (napi_get_element(env, v, i, &e) != napi_ok || !get_u32_vec(env, e, out[i])) return false;
  }
  return true;
}

struct OpArgs {
  std::vector<uint32_t> a_len, b_len;
  std::vector<std::vector<uint32_t>> tables;
  std::vector<const uint32_t *> ptrs;
  static uint32_t dummy;
  bool decode(napi_env env, napi_value la, napi_value lb, napi_value tb) {
    if (!get_u32_vec(env, la, a_len) || !get_u32_vec(env, lb, b_len) || !get_tables(env, tb, tables)) return false;
    if (a_len.size() != b_len.size() || tables.size() != a_len.size()) return false;
    for (auto &t : tables) ptrs.push_back(t.empty() ? &dummy : t.data());
    return true;
  }
};
uint32_t OpArgs::dummy = 0;

static napi_value bad_args(napi_env env, const char *what) {
  napi_throw_type_error(env, nullptr, what);
  return nullptr;
}

// the caller's current view of the dimensions (oldLen) is applied to the handle first: Cube adds and drops
// one-item dimensions around store calls without touching the cells
static bool sharded_view(napi_env env, olap_sharded_store *s, napi_value old_len) {
  std::vector<uint32_t> lens;
  if (!get_u32_vec(env, old_len, lens)) {
    napi_throw_type_error(env, nullptr, "oldLen must be a Uint32Array");
    return false;
  }
  int rc = olap_sharded_store_reshape(s, (int)lens.size(), lens.data());
  if (rc) {
    throw_olap(env, rc);
    return false;
  }
  return true;
}

static bool decode_lens_tables(napi_env env, napi_value lens, napi_value tables, size_t ndim, std::vector<uint32_t> &len_out,
                               std::vector<std::vector<uint32_t>> &tab_out, std::vector<const uint32_t *> &ptrs) {
  if (!get_u32_vec(env, lens, len_out) || !get_tables(env, tables, tab_out)) return false;
  if (len_out.size() != ndim || tab_out.size() != ndim) return false;
  for (auto &t : tab_out) ptrs.push_back(t.empty() ? &OpArgs::dummy : t.data());
  return true;
}

// drillUp(oldLen, newLen, maps, methodCode)
static napi_value StoreDrillUp(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(4)
  OpArgs a;
  int32_t method = 0;
  if (argc < 4 || !a.decode(env, argv[0], argv[1], argv[2]) || napi_get_value_int32(env, argv[3], &method) != napi_ok)
    return bad_args(env, "drillUp(oldLen: Uint32Array, newLen: Uint32Array, maps: Uint32Array[], method: number)");
  olap_store *out = nullptr;
  int rc = olap_store_drillup(s, &out, (int)a.a_len.size(), a.a_len.data(), a.b_len.data(), a.ptrs.data(), method);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, out);
}

// drillDown(oldLen, newLen, maps, methodCode, distributions: Float64Array | null)
static napi_value StoreDrillDown(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(5)
  OpArgs a;
  int32_t method = 0;
  if (argc < 4 || !a.decode(env, argv[0], argv[1], argv[2]) || napi_get_value_int32(env, argv[3], &method) != napi_ok)
    return bad_args(env, "drillDown(oldLen, newLen, maps, method, distributions)");
  const double *dist = nullptr;
  size_t n_dist = 0;
  if (argc > 4) {
    bool is_ta = false;
    napi_is_typedarray(env, argv[4], &is_ta);
    if (is_ta) {
      napi_typedarray_type type;
      void *data;
      NAPI_OK(napi_get_typedarray_info(env, argv[4], &type, &n_dist, &data, nullptr, nullptr));
      if (type != napi_float64_array) return bad_args(env, "distributions must be a Float64Array");
      dist = (const double *)data;
      static const double none = 0;
      if (!dist) dist = &none;
    }
  }
  olap_store *out = nullptr;
  int rc = olap_store_drilldown(s, &out, (int)a.a_len.size(), a.a_len.data(), a.b_len.data(), a.ptrs.data(), method, dist, n_dist);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, out);
}

// dice(oldLen, newLen, sel: Int32Array[])
static napi_value StoreDice(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(3)
  OpArgs a;
  if (argc < 3 || !a.decode(env, argv[0], argv[1], argv[2])) return bad_args(env, "dice(oldLen, newLen, sel: Int32Array[])");
  olap_store *out = nullptr;
  int rc = olap_store_dice(s, &out, (int)a.a_len.size(), a.a_len.data(), a.b_len.data(), (const int32_t *const *)a.ptrs.data());
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, out);
}

// diceDrillUp(oldLen, midLen, newLen, sel: Int32Array[], maps: Uint32Array[], methodCode)
static napi_value StoreDiceDrillUp(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(6)
  OpArgs a;
  std::vector<uint32_t> new_len;
  std::vector<std::vector<uint32_t>> maps;
  int32_t method = 0;
  if (argc < 6 || !a.decode(env, argv[0], argv[1], argv[3]) || !get_u32_vec(env, argv[2], new_len) || !get_tables(env, argv[4], maps) ||
      new_len.size() != a.a_len.size() || maps.size() != a.a_len.size() || napi_get_value_int32(env, argv[5], &method) != napi_ok)
    return bad_args(env, "diceDrillUp(oldLen, midLen, newLen, sel: Int32Array[], maps: Uint32Array[], method)");
  std::vector<const uint32_t *> map_ptrs;
  for (auto &m : maps) map_ptrs.push_back(m.empty() ? &OpArgs::dummy : m.data());
  olap_store *out = nullptr;
  int rc = olap_store_dice_drillup(s, &out, (int)a.a_len.size(), a.a_len.data(), a.b_len.data(), new_len.data(),
                                   (const int32_t *const *)a.ptrs.data(), map_ptrs.data(), method);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, out);
}

// reorder(oldLen, perm: Int32Array)
static napi_value StoreReorder(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(2)
  std::vector<uint32_t> len, perm;
  if (argc < 2 || !get_u32_vec(env, argv[0], len) || !get_u32_vec(env, argv[1], perm) || len.size() != perm.size())
    return bad_args(env, "reorder(oldLen: Uint32Array, perm: Int32Array)");
  olap_store *out = nullptr;
  int rc = olap_store_reorder(s, &out, (int)len.size(), len.data(), (const int32_t *)perm.data());
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, out);
}

// load(other: Store, myLen, hisLen, hisToMine: Int32Array[])
static napi_value StoreLoad(napi_env env, napi_callback_info info) {
  STORE_METHOD_PROLOGUE(4)
  if (argc < 4) return bad_args(env, "load(other, myLen, hisLen, hisToMine)");
  olap_store *other = unwrap(env, argv[0]);
  if (!other) return nullptr;
  OpArgs a;
  if (!a.decode(env, argv[1], argv[2], argv[3])) {
    // tables are indexed by HIS dimensions: decode() only checks equal counts
    return bad_args(env, "load(other: Store, myLen, hisLen, hisToMine: Int32Array[])");
  }
  int rc = olap_store_load(s, other, (int)a.a_len.size(), a.a_len.data(), a.b_len.data(), (const int32_t *const *)a.ptrs.data());
  if (rc) return throw_olap(env, rc);
  return nullptr;
}

// evalFormula(code: Int32Array, consts: Float64Array, stores: Store[], scalars: Float64Array) -> Float64Array
static napi_value EvalFormula(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (argc < 4) return bad_args(env, "evalFormula(code, consts, stores, scalars)");
  napi_typedarray_type t;
  size_t n_code = 0, n_consts = 0, n_scalars = 0;
  void *code = nullptr, *consts = nullptr, *scalars = nullptr;
  if (napi_get_typedarray_info(env, argv[0], &t, &n_code, &code, nullptr, nullptr) != napi_ok || t != napi_int32_array ||
      napi_get_typedarray_info(env, argv[1], &t, &n_consts, &consts, nullptr, nullptr) != napi_ok || t != napi_float64_array ||
      napi_get_typedarray_info(env, argv[3], &t, &n_scalars, &scalars, nullptr, nullptr) != napi_ok || t != napi_float64_array)
    return bad_args(env, "evalFormula(code: Int32Array, consts: Float64Array, stores: Store[], scalars: Float64Array)");
  bool is_arr = false;
  napi_is_array(env, argv[2], &is_arr);
  if (!is_arr) return bad_args(env, "evalFormula: stores must be an array of Store");
  uint32_t n_inputs = 0;
  napi_get_array_length(env, argv[2], &n_inputs);
  std::vector<const olap_store *> stores(n_inputs);
  for (uint32_t i = 0; i < n_inputs; ++i) {
    napi_value e;
    NAPI_OK(napi_get_element(env, argv[2], i, &e));
    stores[i] = unwrap(env, e);
    if (!stores[i]) return nullptr;
  }
  const uint64_t n = n_inputs ? olap_store_size(stores[0]) : 0;
  void *out;
  napi_value ta = make_ta(env, napi_float64_array, 8, n, &out);
  if (!ta) return nullptr;
  static const double zero = 0;
  int rc = olap_store_eval_formula((const int32_t *)code, (int)n_code, consts ? (const double *)consts : &zero, (int)n_consts, (int)n_inputs,
                                   stores.data(), scalars ? (const double *)scalars : &zero, (int)n_scalars, (double *)out);
  if (rc) return throw_olap(env, rc);
  return ta;
}

// drillUpMulti(stores: Store[], methods: Int32Array, oldLen, newLen, maps) -> Store[]
// Every stored measure of a cube with its own rule for the rolled-up dimension (olap_store_drillup_multi): one
// mixed-rule launch where the roll-up allows it, one launch per rule otherwise.
static napi_value DrillUpMulti(napi_env env, napi_callback_info info) {
  size_t argc = 5;
  napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  OpArgs a;
  bool is_arr = false;
  if (argc >= 1) napi_is_array(env, argv[0], &is_arr);
  napi_typedarray_type mt;
  size_t n_methods = 0;
  void *mdata = nullptr;
  if (argc < 5 || !is_arr || napi_get_typedarray_info(env, argv[1], &mt, &n_methods, &mdata, nullptr, nullptr) != napi_ok || mt != napi_int32_array ||
      !a.decode(env, argv[2], argv[3], argv[4]))
    return bad_args(env, "drillUpMulti(stores: Store[], methods: Int32Array, oldLen: Uint32Array, newLen: Uint32Array, maps: Uint32Array[])");
  uint32_t n = 0;
  napi_get_array_length(env, argv[0], &n);
  if (n_methods != n) return bad_args(env, "drillUpMulti: one method per store");
  std::vector<const olap_store *> stores(n);
  for (uint32_t i = 0; i < n; ++i) {
    napi_value e;
    NAPI_OK(napi_get_element(env, argv[0], i, &e));
    stores[i] = unwrap(env, e);
    if (!stores[i]) return nullptr;
  }
  std::vector<olap_store *> outs(n, nullptr);
  static const int none = 0;
  int rc = olap_store_drillup_multi((int)n, stores.data(), n ? (const int *)mdata : &none, outs.data(), (int)a.a_len.size(), a.a_len.data(), a.b_len.data(),
                                    a.ptrs.data());
  if (rc) return throw_olap(env, rc);
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, n, &arr));
  for (uint32_t i = 0; i < n; ++i) {
    napi_value w = wrap_new_store(env, outs[i]);
    if (!w) {
      for (uint32_t j = i + 1; j < n; ++j) olap_store_destroy(outs[j]);
      return nullptr;
    }
    napi_set_element(env, arr, i, w);
  }
  return arr;
}

// ---- ShardedStore: a measure split along dimension 0 over the devices of setDevices() -----------
// Wraps olap_sharded_store* (include/olap_hip.h, "Multi-GPU").  Method names and argument shapes are
// those of Store, so the JS HipStore drives either; what a sharded store cannot do in place throws an
// Error whose message starts with "sharded:" and the JS side gathers first.
struct CommRef {
  olap_comm *comm;
  int refs;      // live ShardedStore wrappers + 1 while it is the current communicator
};
static CommRef *g_comm = nullptr;
static napi_ref g_sharded_ctor = nullptr;

static void comm_release(CommRef *c) {
  if (c && --c->refs == 0) {
    olap_comm_destroy(c->comm);
    delete c;
  }
}

struct ShardedBox {
  olap_sharded_store *store;
  CommRef *comm;
  int64_t bytes;
};

static void finalize_sharded(napi_env env, void *data, void *) {
  ShardedBox *box = (ShardedBox *)data;
  if (box->store) {
    account(env, -box->bytes);
    olap_sharded_store_destroy(box->store);
  }
  comm_release(box->comm);
  delete box;
}

static ShardedBox *unwrap_sharded(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_unwrap(env, v, &p) != napi_ok || !p || !((ShardedBox *)p)->store) {
    napi_throw_type_error(env, nullptr, "not a ShardedStore");
    return nullptr;
  }
  return (ShardedBox *)p;
}

struct ShardedInit {
  olap_sharded_store *store;
  CommRef *comm;
};

static int64_t sharded_bytes(const olap_sharded_store *s) {
  const olap_store *first = olap_sharded_store_shard(s, 0);
  return (int64_t)(olap_sharded_store_size(s) * olap_dtype_size(first ? olap_store_dtype(first) : OLAP_FLOAT32));
}

static napi_value wrap_new_sharded(napi_env env, olap_sharded_store *s, CommRef *comm) {
  napi_value ctor, obj, ext;
  ShardedInit init{s, comm};
  NAPI_OK(napi_get_reference_value(env, g_sharded_ctor, &ctor));
  NAPI_OK(napi_create_external(env, &init, nullptr, nullptr, &ext));
  if (napi_new_instance(env, ctor, 1, &ext, &obj) != napi_ok) {
    olap_sharded_store_destroy(s);
    return nullptr;
  }
  return obj;
}

// new ShardedStore(lens: Uint32Array, dtypeCode, defaultKind)  |  new ShardedStore(external) [internal]
static napi_value ShardedNew(napi_env env, napi_callback_info info) {
  size_t argc = 3;
  napi_value argv[3], self;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
  napi_valuetype t;
  NAPI_OK(napi_typeof(env, argv[0], &t));
  olap_sharded_store *s = nullptr;
  CommRef *comm = nullptr;
  bool adopted = false;  // the handle came from wrap_new_sharded(), which destroys it itself when this constructor fails
  if (argc == 1 && t == napi_external) {
    void *p;
    NAPI_OK(napi_get_value_external(env, argv[0], &p));
    s = ((ShardedInit *)p)->store;
    comm = ((ShardedInit *)p)->comm;
    adopted = true;
  } else {
    if (!g_comm) {
      napi_throw_error(env, nullptr, "sharded: no device list; call setDevices([...]) first");
      return nullptr;
    }
    std::vector<uint32_t> lens;
    int32_t dtype = 0, def = 0;
    if (argc < 3 || !get_u32_vec(env, argv[0], lens) || napi_get_value_int32(env, argv[1], &dtype) != napi_ok ||
        napi_get_value_int32(env, argv[2], &def) != napi_ok) {
      napi_throw_type_error(env, nullptr, "new ShardedStore(lens: Uint32Array, dtype: number, default: number)");
      return nullptr;
    }
    comm = g_comm;
    int rc = olap_sharded_store_create(&s, comm->comm, (int)lens.size(), lens.data(), dtype, def, nullptr);
    if (rc) return throw_olap(env, rc);
  }
  ShardedBox *box = new ShardedBox{s, comm, sharded_bytes(s)};
  comm->refs++;
  if (napi_wrap(env, self, box, finalize_sharded, nullptr, nullptr) != napi_ok) {
    if (!adopted) olap_sharded_store_destroy(s);
    comm_release(comm);
    delete box;
    napi_throw_error(env, nullptr, "napi_wrap failed");
    return nullptr;
  }
  account(env, box->bytes);
  return self;
}

#define SHARDED_PROLOGUE(MAXARGS)                                          \
  size_t argc = MAXARGS;                                                   \
  napi_value argv[MAXARGS > 0 ? MAXARGS : 1], self;                        \
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));       \
  ShardedBox *box = unwrap_sharded(env, self);                             \
  if (!box) return nullptr;                                                \
  olap_sharded_store *s = box->store;

static napi_value ShardedSize(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  return num(env, (double)olap_sharded_store_size(s));
}
static napi_value ShardedByteLength(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  return num(env, (double)box->bytes);
}
static napi_value ShardedDtype(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  return num(env, olap_store_dtype(olap_sharded_store_shard(s, 0)));
}
static napi_value ShardedDefault(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  return num(env, olap_store_default(olap_sharded_store_shard(s, 0)));
}
static napi_value ShardedIsSharded(napi_env env, napi_callback_info) {
  napi_value t;
  napi_get_boolean(env, true, &t);
  return t;
}
// rows of dimension 0 owned by each rank: bounds[r] .. bounds[r+1]
static napi_value ShardedBounds(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  const int world = olap_comm_world(olap_sharded_store_comm(s));
  void *data;
  napi_value ta = make_ta(env, napi_uint32_array, 4, (size_t)world + 1, &data);
  if (!ta) return nullptr;
  memcpy(data, olap_sharded_store_bounds(s), ((size_t)world + 1) * 4);
  return ta;
}

// setData(Float64Array): JS numbers, converted on the devices like a TypedArray store would
static napi_value ShardedSetData(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(1)
  napi_typedarray_type type;
  size_t len;
  void *data;
  bool is_ta = false;
  NAPI_OK(napi_is_typedarray(env, argv[0], &is_ta));
  if (!is_ta || napi_get_typedarray_info(env, argv[0], &type, &len, &data, nullptr, nullptr) != napi_ok || type != napi_float64_array) {
    napi_throw_type_error(env, nullptr, "ShardedStore.setData expects a Float64Array");
    return nullptr;
  }
  int rc = olap_sharded_store_set_data_f64(s, (const double *)data, len);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}
static napi_value ShardedGetDataF64(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  void *data;
  napi_value ta = make_ta(env, napi_float64_array, 8, olap_sharded_store_size(s), &data);
  if (!ta) return nullptr;
  int rc = olap_sharded_store_get_data_f64(s, (double *)data);
  if (rc) return throw_olap(env, rc);
  return ta;
}
static napi_value ShardedGetStatus(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  void *data;
  napi_value ta = make_ta(env, napi_int32_array, 4, olap_sharded_store_size(s), &data);
  if (!ta) return nullptr;
  int rc = olap_sharded_store_get_status(s, (int32_t *)data);
  if (rc) return throw_olap(env, rc);
  return ta;
}
static napi_value ShardedGetValue(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(1)
  double idx = 0;
  NAPI_OK(napi_get_value_double(env, argv[0], &idx));
  napi_value undef;
  napi_get_undefined(env, &undef);
  if (!(idx >= 0) || idx != std::floor(idx)) return undef;
  double v = 0;
  int is_set = 0;
  int rc = olap_sharded_store_get_value(s, (uint64_t)idx, &v, &is_set);
  if (rc) return throw_olap(env, rc);
  return is_set ? num(env, v) : undef;
}
static napi_value ShardedSetValue(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(2)
  double idx = 0, v = 0;
  NAPI_OK(napi_get_value_double(env, argv[0], &idx));
  napi_valuetype t = napi_undefined;
  if (argc > 1) NAPI_OK(napi_typeof(env, argv[1], &t));
  int is_null = (t == napi_undefined || t == napi_null);
  if (!is_null) {
    napi_value coerced;
    NAPI_OK(napi_coerce_to_number(env, argv[1], &coerced));
    NAPI_OK(napi_get_value_double(env, coerced, &v));
  }
  int rc = olap_sharded_store_set_value(s, (uint64_t)idx, v, is_null);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}
static napi_value ShardedFill(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(1)
  double v = 0;
  napi_value coerced;
  NAPI_OK(napi_coerce_to_number(env, argv[0], &coerced));
  NAPI_OK(napi_get_value_double(env, coerced, &v));
  int rc = olap_sharded_store_fill(s, v);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}
static napi_value ShardedTotal(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  double t = 0;
  int rc = olap_sharded_store_total(s, &t);
  if (rc) return throw_olap(env, rc);
  return num(env, t);
}
static napi_value ShardedClone(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  olap_sharded_store *c = nullptr;
  int rc = olap_sharded_store_clone(s, &c);
  if (rc) return throw_olap(env, rc);
  return wrap_new_sharded(env, c, box->comm);
}
// gather() -> Store: the whole measure on the first device
static napi_value ShardedGather(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(0)
  olap_store *w = nullptr;
  int rc = olap_sharded_store_gather(s, &w);
  if (rc) return throw_olap(env, rc);
  return wrap_new_store(env, w);
}

// drillUp(oldLen, newLen, maps, method) -> ShardedStore (dimension 0 kept) | Store (dimension 0 rolled up: one collective)
static napi_value ShardedDrillUp(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(4)
  if (argc < 1 || !sharded_view(env, s, argv[0])) return nullptr;
  std::vector<uint32_t> new_len;
  std::vector<std::vector<uint32_t>> tables;
  std::vector<const uint32_t *> ptrs;
  int32_t method = 0;
  if (argc < 4 || !decode_lens_tables(env, argv[1], argv[2], (size_t)olap_sharded_store_ndim(s), new_len, tables, ptrs) ||
      napi_get_value_int32(env, argv[3], &method) != napi_ok)
    return bad_args(env, "drillUp(oldLen, newLen, maps, method)");
  olap_sharded_store *os = nullptr;
  olap_store *ow = nullptr;
  int rc = olap_sharded_store_drillup(s, &os, &ow, new_len.data(), ptrs.data(), method);
  if (rc) return throw_olap(env, rc);
  return os ? wrap_new_sharded(env, os, box->comm) : wrap_new_store(env, ow);
}
// dice(oldLen, newLen, sel)
static napi_value ShardedDice(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(3)
  if (argc < 1 || !sharded_view(env, s, argv[0])) return nullptr;
  std::vector<uint32_t> new_len;
  std::vector<std::vector<uint32_t>> tables;
  std::vector<const uint32_t *> ptrs;
  if (argc < 3 || !decode_lens_tables(env, argv[1], argv[2], (size_t)olap_sharded_store_ndim(s), new_len, tables, ptrs))
    return bad_args(env, "dice(oldLen, newLen, sel)");
  olap_sharded_store *os = nullptr;
  int rc = olap_sharded_store_dice(s, &os, new_len.data(), (const int32_t *const *)ptrs.data());
  if (rc) return throw_olap(env, rc);
  return wrap_new_sharded(env, os, box->comm);
}
// drillDown(oldLen, newLen, maps, method, distributions)
static napi_value ShardedDrillDown(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(5)
  if (argc < 1 || !sharded_view(env, s, argv[0])) return nullptr;
  std::vector<uint32_t> new_len;
  std::vector<std::vector<uint32_t>> tables;
  std::vector<const uint32_t *> ptrs;
  int32_t method = 0;
  if (argc < 4 || !decode_lens_tables(env, argv[1], argv[2], (size_t)olap_sharded_store_ndim(s), new_len, tables, ptrs) ||
      napi_get_value_int32(env, argv[3], &method) != napi_ok)
    return bad_args(env, "drillDown(oldLen, newLen, maps, method, distributions)");
  const double *dist = nullptr;
  size_t n_dist = 0;
  if (argc > 4) {
    bool is_ta = false;
    napi_is_typedarray(env, argv[4], &is_ta);
    if (is_ta) {
      napi_typedarray_type type;
      void *data;
      NAPI_OK(napi_get_typedarray_info(env, argv[4], &type, &n_dist, &data, nullptr, nullptr));
      static const double none = 0;
      dist = data ? (const double *)data : &none;
    }
  }
  olap_sharded_store *os = nullptr;
  int rc = olap_sharded_store_drilldown(s, &os, new_len.data(), ptrs.data(), method, dist, n_dist);
  if (rc) return throw_olap(env, rc);
  return wrap_new_sharded(env, os, box->comm);
}
// reorder(oldLen, perm)
static napi_value ShardedReorder(napi_env env, napi_callback_info info) {
  SHARDED_PROLOGUE(2)
  if (argc < 1 || !sharded_view(env, s, argv[0])) return nullptr;
  std::vector<uint32_t> perm;
  if (argc < 2 || !get_u32_vec(env, argv[1], perm) || perm.size() != (size_t)olap_sharded_store_ndim(s))
    return bad_args(env, "reorder(oldLen, perm)");
  olap_sharded_store *os = nullptr;
  int rc = olap_sharded_store_reorder(s, &os, (const int32_t *)perm.data());
  if (rc) return throw_olap(env, rc);
  return wrap_new_sharded(env, os, box->comm);
}

// shardStore(store: Store, lens: Uint32Array) -> ShardedStore (olap_sharded_store_scatter)
static napi_value ShardStore(napi_env env, napi_callback_info info) {
  size_t argc = 2;
  napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (!g_comm) {
    napi_throw_error(env, nullptr, "sharded: no device list; call setDevices([...]) first");
    return nullptr;
  }
  std::vector<uint32_t> lens;
  olap_store *whole = argc > 0 ? unwrap(env, argv[0]) : nullptr;
  if (!whole) return nullptr;
  if (argc < 2 || !get_u32_vec(env, argv[1], lens)) return bad_args(env, "shardStore(store, lens: Uint32Array)");
  olap_sharded_store *s = nullptr;
  int rc = olap_sharded_store_scatter(&s, g_comm->comm, whole, (int)lens.size(), lens.data());
  if (rc) return throw_olap(env, rc);
  return wrap_new_sharded(env, s, g_comm);
}

// evalFormulaSharded(code, consts, stores: ShardedStore[], scalars) -> Float64Array: per shard, no gather
static napi_value EvalFormulaSharded(napi_env env, napi_callback_info info) {
  size_t argc = 4;
  napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  if (argc < 4) return bad_args(env, "evalFormulaSharded(code, consts, stores, scalars)");
  napi_typedarray_type t;
  size_t n_code = 0, n_consts = 0, n_scalars = 0;
  void *code = nullptr, *consts = nullptr, *scalars = nullptr;
  if (napi_get_typedarray_info(env, argv[0], &t, &n_code, &code, nullptr, nullptr) != napi_ok || t != napi_int32_array ||
      napi_get_typedarray_info(env, argv[1], &t, &n_consts, &consts, nullptr, nullptr) != napi_ok || t != napi_float64_array ||
      napi_get_typedarray_info(env, argv[3], &t, &n_scalars, &scalars, nullptr, nullptr) != napi_ok || t != napi_float64_array)
    return bad_args(env, "evalFormulaSharded(code: Int32Array, consts: Float64Array, stores: ShardedStore[], scalars: Float64Array)");
  bool is_arr = false;
  napi_is_array(env, argv[2], &is_arr);
  uint32_t n_inputs = 0;
  if (is_arr) napi_get_array_length(env, argv[2], &n_inputs);
  if (!is_arr || n_inputs == 0) return bad_args(env, "evalFormulaSharded: stores must be a non-empty array of ShardedStore");
  std::vector<const olap_sharded_store *> stores(n_inputs);
  for (uint32_t i = 0; i < n_inputs; ++i) {
    napi_value e;
    NAPI_OK(napi_get_element(env, argv[2], i, &e));
    ShardedBox *box = unwrap_sharded(env, e);
    if (!box) return nullptr;
    stores[i] = box->store;
  }
  void *out;
  napi_value ta = make_ta(env, napi_float64_array, 8, olap_sharded_store_size(stores[0]), &out);
  if (!ta) return nullptr;
  static const double zero = 0;
  int rc = olap_sharded_store_eval_formula((const int32_t *)code, (int)n_code, consts ? (const double *)consts : &zero, (int)n_consts, (int)n_inputs,
                                           stores.data(), scalars ? (const double *)scalars : &zero, (int)n_scalars, (double *)out);
  if (rc) return throw_olap(env, rc);
  return ta;
}

// setDevices(devices: number[] | null): the devices new sharded measures are split over
// (olap_comm_init_all: distinct devices talk over RCCL / xGMI; one device repeated exchanges directly)
static napi_value SetDevices(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  bool is_arr = false;
  if (argc) napi_is_array(env, argv[0], &is_arr);
  std::vector<int> devices;
  if (is_arr) {
    uint32_t n = 0;
    napi_get_array_length(env, argv[0], &n);
    for (uint32_t i = 0; i < n; ++i) {
      napi_value e;
      int32_t d = 0;
      NAPI_OK(napi_get_element(env, argv[0], i, &e));
      NAPI_OK(napi_get_value_int32(env, e, &d));
      devices.push_back(d);
    }
  }
  CommRef *next = nullptr;
  if (devices.size() > 1) {
    olap_comm *c = nullptr;
    int rc = olap_comm_init_all(&c, devices.data(), (int)devices.size());
    if (rc) return throw_olap(env, rc);
    next = new CommRef{c, 1};
  }
  comm_release(g_comm);
  g_comm = next;
  return num(env, next ? olap_comm_world(next->comm) : 0);
}
static napi_value ShardWorld(napi_env env, napi_callback_info) { return num(env, g_comm ? olap_comm_world(g_comm->comm) : 0); }
static napi_value ShardTransport(napi_env env, napi_callback_info) {
  napi_value v;
  napi_create_string_utf8(env, g_comm ? olap_comm_transport(g_comm->comm) : "none", NAPI_AUTO_LENGTH, &v);
  return v;
}

// ---- module functions -------------------------------------------------------------------------
static napi_value MethodFromName(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  napi_valuetype t = napi_undefined;
  if (argc) napi_typeof(env, argv[0], &t);
  int code;
  if (t == napi_undefined) {
    code = olap_method_from_name(nullptr);
  } else {
    napi_value str;
    NAPI_OK(napi_coerce_to_string(env, argv[0], &str));
    char buf[128];
    size_t n = 0;
    NAPI_OK(napi_get_value_string_utf8(env, str, buf, sizeof buf, &n));
    code = olap_method_from_name(buf);
  }
  if (code < 0) return throw_olap(env, code);
  return num(env, code);
}

static napi_value DeviceCount(napi_env env, napi_callback_info) { return num(env, olap_device_count()); }
// bytes of device memory currently owned by live Store objects (diagnostics)
static napi_value HeldBytes(napi_env env, napi_callback_info) { return num(env, (double)g_held); }
static napi_value AbiVersion(napi_env env, napi_callback_info) { return num(env, olap_abi_version()); }

static napi_value SetDevice(napi_env env, napi_callback_info info) {
  size_t argc = 1;
  napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
  int32_t d = 0;
  NAPI_OK(napi_get_value_int32(env, argv[0], &d));
  int rc = olap_set_device(d);
  if (rc) return throw_olap(env, rc);
  return nullptr;
}

static napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
      {"size", nullptr, nullptr, StoreSize, nullptr, nullptr, napi_default, nullptr},
      {"byteLength", nullptr, nullptr, StoreByteLength, nullptr, nullptr, napi_default, nullptr},
      {"dtype", nullptr, nullptr, StoreDtype, nullptr, nullptr, napi_default, nullptr},
      {"defaultKind", nullptr, nullptr, StoreDefault, nullptr, nullptr, napi_default, nullptr},
      {"setData", nullptr, StoreSetData, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getData", nullptr, StoreGetData, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getDataF64", nullptr, StoreGetDataF64, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getStatus", nullptr, StoreGetStatus, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getKeys", nullptr, StoreGetKeys, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"countSet", nullptr, StoreCountSet, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getValue", nullptr, StoreGetValue, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"setValue", nullptr, StoreSetValue, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"fill", nullptr, StoreFill, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"total", nullptr, StoreTotal, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"toSparse", nullptr, StoreToSparse, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"totals", nullptr, StoreTotals, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"trackOrder", nullptr, StoreTrackOrder, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"orderTracked", nullptr, nullptr, StoreOrderTracked, nullptr, nullptr, napi_default, nullptr},
      {"clone", nullptr, StoreClone, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"drillUp", nullptr, StoreDrillUp, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"drillDown", nullptr, StoreDrillDown, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"dice", nullptr, StoreDice, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"diceDrillUp", nullptr, StoreDiceDrillUp, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"reorder", nullptr, StoreReorder, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"load", nullptr, StoreLoad, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_value ctor;
  if (napi_define_class(env, "Store", NAPI_AUTO_LENGTH, StoreNew, nullptr, sizeof(props) / sizeof(props[0]), props, &ctor) != napi_ok) return nullptr;
  napi_create_reference(env, ctor, 1, &g_store_ctor);
  napi_set_named_property(env, exports, "Store", ctor);
  napi_property_descriptor sprops[] = {
      {"size", nullptr, nullptr, ShardedSize, nullptr, nullptr, napi_default, nullptr},
      {"byteLength", nullptr, nullptr, ShardedByteLength, nullptr, nullptr, napi_default, nullptr},
      {"dtype", nullptr, nullptr, ShardedDtype, nullptr, nullptr, napi_default, nullptr},
      {"defaultKind", nullptr, nullptr, ShardedDefault, nullptr, nullptr, napi_default, nullptr},
      {"isSharded", nullptr, nullptr, ShardedIsSharded, nullptr, nullptr, napi_default, nullptr},
      {"bounds", nullptr, nullptr, ShardedBounds, nullptr, nullptr, napi_default, nullptr},
      {"setData", nullptr, ShardedSetData, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getDataF64", nullptr, ShardedGetDataF64, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getStatus", nullptr, ShardedGetStatus, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getValue", nullptr, ShardedGetValue, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"setValue", nullptr, ShardedSetValue, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"fill", nullptr, ShardedFill, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"total", nullptr, ShardedTotal, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"clone", nullptr, ShardedClone, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"gather", nullptr, ShardedGather, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"drillUp", nullptr, ShardedDrillUp, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"drillDown", nullptr, ShardedDrillDown, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"dice", nullptr, ShardedDice, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"reorder", nullptr, ShardedReorder, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_value sctor;
  if (napi_define_class(env, "ShardedStore", NAPI_AUTO_LENGTH, ShardedNew, nullptr, sizeof(sprops) / sizeof(sprops[0]), sprops, &sctor) != napi_ok) return nullptr;
  napi_create_reference(env, sctor, 1, &g_sharded_ctor);
  napi_set_named_property(env, exports, "ShardedStore", sctor);
  napi_property_descriptor fns[] = {
      {"setDevices", nullptr, SetDevices, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"shardWorld", nullptr, ShardWorld, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"shardTransport", nullptr, ShardTransport, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"shardStore", nullptr, ShardStore, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"evalFormulaSharded", nullptr, EvalFormulaSharded, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"evalFormula", nullptr, EvalFormula, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"drillUpMulti", nullptr, DrillUpMulti, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"storeFromSparse", nullptr, StoreFromSparse, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"methodFromName", nullptr, MethodFromName, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"heldBytes", nullptr, HeldBytes, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"abiVersion", nullptr, AbiVersion, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"setDevice", nullptr, SetDevice, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(fns) / sizeof(fns[0]), fns);
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
