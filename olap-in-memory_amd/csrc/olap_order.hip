// olap_order.hip — the insertion order of the reference's Map, for stores that ask for it
// (olap_store_track_order).
//
// The reference keeps a measure's cells in a Map and iterates it in INSERTION order
// (/root/reference/src/store/in-memory.js:298 drillUp, :189 reorder, :240 dice; keys() / serialize()
// expose it).  A dense buffer has no such order; for a store filled in ascending order (`data=`, `fill`)
// and rolled up while dense the two coincide, which is why the bulk kernels simply use the flat index.
// They differ after an out-of-order setValue, after `data=` over a partly filled store, after a roll-up
// of a SPARSE store along an outer dimension (its result Map is ordered by first hit), after a dice
// that permutes items and after any reorder — and then `first` / `last` answer differently.
//
// A tracked store therefore carries `seq`: one uint32 per cell, 0 for an unset cell, and set cells compare
// by seq exactly as the reference's Map entries compare by age.  seq == nullptr means "ascending flat
// index" and costs nothing.  Every store operation maps to its effect on that order:
//   setValue        a cell that becomes set is appended (seq = next), one that stays set keeps its place
//   data= / fill    cells that stay set keep their place, newly set cells are appended in index order
//   drillUp         an output cell sits where its FIRST contributing input cell sat: min of the members' seq
//                   (`first` / `last` pick the member with the smallest / largest seq: drillup_byseq_kernel)
//   dice, reorder   a cell keeps the seq of the cell it was copied from (the same plan run over seq)
//   drillDown       visits the new cells in ascending index order: ascending again
//   load            visits HIS cells in ascending index order (:159): cells it creates are appended in that order
// Values of order-independent methods come from the ordinary kernels; only seq is computed here.
// sum / average / product DO depend on the order — float64 addition is not associative, and a running value that
// hits the default drops the key, which then re-enters the Map at the END (in-memory.js:311-318, :126-131).  Over a
// store whose order is ascending the ordinary kernels already add in that order; over any other order the
// contributions are SORTED by (output cell, seq) and every output cell is replayed contribution by contribution
// (order_drillup_replay): the reference's values bit for bit and its key order exactly, at the price of a radix sort.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "olap_device.hpp"
#include "olap_internal.hpp"

using namespace olap;

namespace {

#define ORDER_DISPATCH(dtype, CALL)                        \
  switch (dtype) {                                         \
    case OLAP_INT32: { using T = int32_t; CALL; break; }   \
    case OLAP_UINT32: { using T = uint32_t; CALL; break; } \
    case OLAP_FLOAT32: { using T = float; CALL; break; }   \
    default: { using T = double; CALL; break; }            \
  }

unsigned grid_for_n(uint64_t n) {
  const uint64_t want = (n + kBlock - 1) / kBlock;
  return (unsigned)(want < 1 ? 1 : (want < 4096 ? want : 4096));
}

int launched(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, what);
  return OLAP_OK;
}

// seq[i] = set ? i + 1 : 0
template <typename T>
__global__ __launch_bounds__(kBlock) void seq_iota_kernel(const T *values, const int32_t *status, uint32_t *seq, uint64_t n, int def_nan) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
    seq[i] = cell_is_set<T>(values[i], status ? status[i] : OLAP_STATUS_SET, status != nullptr, def_nan != 0) ? (uint32_t)(i + 1) : 0u;
}
// after a bulk write: cells that stay set keep their seq, newly set ones are appended in index order, unset ones lose it
template <typename T>
__global__ __launch_bounds__(kBlock) void seq_after_bulk_kernel(const T *values, const int32_t *status, uint32_t *seq, uint64_t n, int def_nan,
                                                                uint32_t base) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const bool set = cell_is_set<T>(values[i], status ? status[i] : OLAP_STATUS_SET, status != nullptr, def_nan != 0);
    const uint32_t old = seq[i];
    seq[i] = set ? (old ? old : base + (uint32_t)i + 1u) : 0u;
  }
}
// seq[i] = set ? (seq[i] ? seq[i] : fallback[i]) : 0   (fallback may be nullptr)
template <typename T>
__global__ __launch_bounds__(kBlock) void seq_mask_kernel(const T *values, const int32_t *status, uint32_t *seq, const uint32_t *fallback, uint64_t n,
                                                          int def_nan) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const bool set = cell_is_set<T>(values[i], status ? status[i] : OLAP_STATUS_SET, status != nullptr, def_nan != 0);
    uint32_t v = seq[i];
    if (!v && fallback) v = fallback[i];
    seq[i] = set ? v : 0u;
  }
}
template <typename T>
__global__ void seq_set_cell_kernel(const T *values, const int32_t *status, uint32_t *seq, uint64_t index, uint32_t next, int def_nan) {
  const bool set = cell_is_set<T>(values[index], status ? status[index] : OLAP_STATUS_SET, status != nullptr, def_nan != 0);
  seq[index] = set ? (seq[index] ? seq[index] : next) : 0u;
}
__global__ __launch_bounds__(kBlock) void seq_ramp_kernel(uint32_t *dst, uint64_t n, uint32_t base) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) dst[i] = base + (uint32_t)i + 1u;
}

// `first` / `last` of a store whose order is not the flat index: view [outer, K, inner] -> [outer, G, inner], one lane
// per output cell (lanes along `inner`: coalesced), members of group g = order[gstart[g] .. gstart[g+1])
template <typename T>
__global__ __launch_bounds__(kBlock) void drillup_byseq_kernel(const T *__restrict__ in, const int32_t *__restrict__ st_in,
                                                               const uint32_t *__restrict__ in_seq, T *__restrict__ out,
                                                               int32_t *__restrict__ st_out, uint32_t *__restrict__ out_seq, uint64_t outer,
                                                               uint64_t K, uint64_t G, uint64_t inner, const uint32_t *__restrict__ gstart,
                                                               const uint32_t *__restrict__ order, int last, int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  const uint64_t total = outer * G * inner;
  for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = t % inner, og = t / inner, g = og % G, o = og / G;
    const uint64_t base = o * K * inner + i;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    T v_lo = Cell<T>::default_value(def_nan), v_hi = v_lo;
    for (uint32_t j = gstart[g]; j < gstart[g + 1]; ++j) {
      const uint64_t at = base + (uint64_t)order[j] * inner;
      const uint32_t s = in_seq[at];
      const T x = in[at];
      if (s == 0u || !cell_is_set<T>(x, st_in ? st_in[at] : OLAP_STATUS_SET, st_in != nullptr, def_nan)) continue;
      if (s < lo) {
        lo = s;
        v_lo = x;
      }
      if (s >= hi) {
        hi = s;
        v_hi = x;
      }
    }
    const bool has = hi != 0u;
    out[t] = has ? (last ? v_hi : v_lo) : Cell<T>::default_value(def_nan);
    if (st_out) st_out[t] = has ? OLAP_STATUS_SET : 0;
    out_seq[t] = has ? lo : 0u;  // the output cell was inserted when its first member was visited
  }
}

// ---- drillUp of a store whose order is NOT the flat index, for the rules that depend on it ------------------------
struct ReplayDims {
  int nd;
  uint32_t old_len[OLAP_MAX_DIMS];
  uint64_t new_stride[OLAP_MAX_DIMS];
  uint32_t tab_off[OLAP_MAX_DIMS];  // start of dimension d's map (old item -> new item) in `tab`
  const uint32_t *tab;               // device
};

// key = (output cell << 32) | seq for a set cell, all ones for an unset one (sorted to the end); value = the cell's index
__global__ __launch_bounds__(kBlock) void replay_keys_kernel(const uint32_t *__restrict__ seq, uint64_t n, const ReplayDims d, uint64_t *__restrict__ keys,
                                                             uint32_t *__restrict__ vals) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    const uint32_t q = seq[i];
    uint64_t key = ~0ull;
    if (q) {
      uint64_t c = i, o = 0;
      for (int k = d.nd - 1; k >= 0; --k) {
        const uint32_t digit = (uint32_t)(c % d.old_len[k]);
        c /= d.old_len[k];
        o += (uint64_t)d.tab[d.tab_off[k] + digit] * d.new_stride[k];
      }
      key = (o << 32) | (uint64_t)q;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
}

// One lane per output cell that has contributions: the lane whose sorted position starts the cell's run replays the
// run in seq order through the reference's state machine (Agg: first contribution stores, later ones aggregate, a
// running value equal to the default drops the key — which re-enters at the position of the contribution that
// brings it back, in-memory.js:311-318).
template <typename T, int METHOD>
__global__ __launch_bounds__(kBlock) void replay_kernel(const T *__restrict__ in, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                        uint64_t n, T *__restrict__ out, int32_t *__restrict__ st_out, uint32_t *__restrict__ out_seq,
                                                        int def_nan_i) {
  const bool def_nan = def_nan_i != 0;
  for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (uint64_t)gridDim.x * kBlock) {
    const uint64_t key = keys[p];
    if (key == ~0ull) continue;
    const uint64_t cell = key >> 32;
    if (p > 0 && (keys[p - 1] >> 32) == cell) continue;  // not the head of its run
    Agg<METHOD> agg;
    agg.init();
    uint32_t inserted = 0;
    for (uint64_t q = p; q < n; ++q) {
      const uint64_t kq = keys[q];
      if ((kq >> 32) != cell) break;  // (the all-ones keys of unset cells end the last run too)
      const bool had = agg.has;
      agg.add(Cell<T>::to_f64(in[vals[q]]), def_nan);
      if (!had && agg.has) inserted = (uint32_t)kq;
    }
    agg.finish(def_nan);
    T ov;
    int32_t os;
    emit_cell<T>(agg.acc, agg.has, def_nan, ov, os);
    out[cell] = ov;
    if (st_out) st_out[cell] = os;
    out_seq[cell] = os ? inserted : 0u;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void replay_clear_kernel(T *__restrict__ out, int32_t *__restrict__ st_out, uint32_t *__restrict__ out_seq, uint64_t n,
                                                              int def_nan) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
    out[i] = Cell<T>::default_value(def_nan != 0);
    if (st_out) st_out[i] = 0;
    out_seq[i] = 0u;
  }
}

template <typename T>
void launch_replay(int method, const T *in, const uint64_t *keys, const uint32_t *vals, uint64_t n, T *out, int32_t *st_out, uint32_t *out_seq,
                   int def_nan) {
  const unsigned grid = (unsigned)std::min<uint64_t>(65536, (n + kBlock - 1) / kBlock);
#define OLAP_REPLAY(M) hipLaunchKernelGGL((replay_kernel<T, M>), grid, kBlock, 0, nullptr, in, keys, vals, n, out, st_out, out_seq, def_nan)
  switch (method) {
    case OLAP_SUM: OLAP_REPLAY(OLAP_SUM); break;
    case OLAP_AVERAGE: OLAP_REPLAY(OLAP_AVERAGE); break;
    case OLAP_HIGHEST: OLAP_REPLAY(OLAP_HIGHEST); break;
    case OLAP_LOWEST: OLAP_REPLAY(OLAP_LOWEST); break;
    case OLAP_FIRST: OLAP_REPLAY(OLAP_FIRST); break;
    case OLAP_LAST: OLAP_REPLAY(OLAP_LAST); break;
    default: OLAP_REPLAY(OLAP_PRODUCT); break;
  }
#undef OLAP_REPLAY
}

// per workgroup: are the non-zero seq of its contiguous share ascending, and their smallest / largest
struct AscendingPart {
  uint32_t lo, hi;  // smallest / largest non-zero seq of the share (0xFFFFFFFF / 0: none)
  uint32_t ok;
};
__global__ __launch_bounds__(kBlock) void seq_ascending_kernel(const uint32_t *__restrict__ seq, uint64_t n, uint64_t share, AscendingPart *__restrict__ parts) {
  __shared__ uint32_t wave_max[kBlock / 64];
  __shared__ uint32_t s_ok, s_lo;
  const uint64_t beg = (uint64_t)blockIdx.x * share, end = beg + share < n ? beg + share : n;
  if (threadIdx.x == 0) {
    s_ok = 1u;
    s_lo = 0xFFFFFFFFu;
  }
  __syncthreads();
  uint32_t carry = 0;  // largest seq seen before this step (uniform)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t base = beg; base < end; base += kBlock) {
    const uint64_t i = base + threadIdx.x;
    const uint32_t v = i < end ? seq[i] : 0u;
    // exclusive prefix maximum over the step's 256 cells
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d, 64);
      if (lane >= d) inc = inc > up ? inc : up;
    }
    if (lane == 63) wave_max[wave] = inc;
    __syncthreads();
    uint32_t before = carry;
    for (int w = 0; w < wave; ++w) before = before > wave_max[w] ? before : wave_max[w];
    uint32_t exc = __shfl_up(inc, 1, 64);
    exc = lane == 0 ? 0u : exc;
    exc = exc > before ? exc : before;
    if (v && v <= exc) atomicAnd(&s_ok, 0u);
    if (v) atomicMin(&s_lo, v);
    uint32_t step_max = carry;
    for (int w = 0; w < kBlock / 64; ++w) step_max = step_max > wave_max[w] ? step_max : wave_max[w];
    carry = step_max;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    parts[blockIdx.x].lo = s_lo;
    parts[blockIdx.x].hi = carry;
    parts[blockIdx.x].ok = s_ok;
  }
}

// whether the set cells' seq ascend with the flat index (then the ordinary kernels already add in insertion order)
int seq_is_ascending(const olap_store *s, bool *ascending) {
  *ascending = true;
  if (!s->seq || s->size == 0) return OLAP_OK;
  const unsigned blocks = (unsigned)std::min<uint64_t>(1024, (s->size + kBlock - 1) / kBlock);
  const uint64_t share = ((s->size + blocks - 1) / blocks + kBlock - 1) / kBlock * kBlock;
  AscendingPart *dev = nullptr;
  HIP_TRY(dev_alloc((void **)&dev, blocks * sizeof(AscendingPart)));
  hipLaunchKernelGGL(seq_ascending_kernel, blocks, kBlock, 0, nullptr, s->seq, s->size, share, dev);
  std::vector<AscendingPart> host(blocks);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(host.data(), dev, blocks * sizeof(AscendingPart), hipMemcpyDeviceToHost);
  dev_free(dev);
  if (e != hipSuccess) return hip_fail(e, "seq_ascending_kernel");
  uint32_t seen = 0;
  for (const AscendingPart &p : host) {
    if (!p.ok || (p.hi && p.lo <= seen)) *ascending = false;
    seen = std::max(seen, p.hi);
  }
  return OLAP_OK;
}

int seq_alloc(const olap_store *s) {
  if (s->seq) return OLAP_OK;
  uint32_t *q = nullptr;
  HIP_TRY(dev_alloc((void **)&q, std::max<uint64_t>(s->size, 1) * sizeof(uint32_t)));
  s->seq = q;
  return OLAP_OK;
}

// makes the implicit order (ascending flat index) explicit
int seq_materialise(const olap_store *s) {
  if (s->seq) return OLAP_OK;
  if (s->size >= 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "insertion-order tracking supports stores below 2^31 cells");
  int rc = seq_alloc(s);
  if (rc) return rc;
  ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((seq_iota_kernel<T>), grid_for_n(s->size), kBlock, 0, nullptr, (const T *)s->values, mask_needed(s),
                                              s->seq, s->size, s->default_kind == OLAP_DEFAULT_NAN));
  s->next_seq = s->size + 1;
  return launched("seq_iota_kernel");
}

int seq_renumber_if_needed(const olap_store *s, uint64_t wanted) {
  // seq values are handed out upwards and never reused; long before 2^32 they are compacted to ranks on the host
  if (s->next_seq + wanted < 0xF0000000ull) return OLAP_OK;
  std::vector<uint32_t> host(s->size);
  HIP_TRY(hipMemcpy(host.data(), s->seq, s->size * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> idx;
  for (uint64_t i = 0; i < s->size; ++i)
    if (host[i]) idx.push_back((uint32_t)i);
  std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return host[a] < host[b]; });
  for (size_t r = 0; r < idx.size(); ++r) host[idx[r]] = (uint32_t)r + 1;
  HIP_TRY(hipMemcpy(s->seq, host.data(), s->size * sizeof(uint32_t), hipMemcpyHostToDevice));
  s->next_seq = idx.size() + 1;
  if (s->next_seq + wanted >= 0xF0000000ull) return fail(OLAP_ERR_INVALID_ARGUMENT, "insertion-order log full");
  return OLAP_OK;
}

int seq_mask(olap_store *s, const uint32_t *fallback) {
  ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((seq_mask_kernel<T>), grid_for_n(s->size), kBlock, 0, nullptr, (const T *)s->values, mask_needed(s), s->seq,
                                              fallback, s->size, s->default_kind == OLAP_DEFAULT_NAN));
  return launched("seq_mask_kernel");
}

// result store of an operation on a tracked store
void inherit(const olap_store *from, olap_store *to) {
  to->track_order = true;
  to->maybe_nonempty = true;
  to->hi_index = to->size ? to->size - 1 : 0;
  (void)from;
}

bool strictly_increasing(const int32_t *sel, uint32_t n) {
  int64_t prev = -1;
  for (uint32_t j = 0; j < n; ++j) {
    if (sel[j] < 0) continue;
    if (sel[j] <= prev) return false;
    prev = sel[j];
  }
  return true;
}

}  // namespace

void order_free(olap_store *s) {
  if (s->seq) dev_free(s->seq);
  s->seq = nullptr;
}

int order_clone(const olap_store *from, olap_store *to) {
  to->track_order = from->track_order;
  to->maybe_nonempty = from->maybe_nonempty;
  to->hi_index = from->hi_index;
  to->next_seq = from->next_seq;
  if (from->seq) {
    int rc = seq_alloc(to);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(to->seq, from->seq, from->size * sizeof(uint32_t), hipMemcpyDeviceToDevice));
  }
  return OLAP_OK;
}

// data= / fill over a store that may hold cells: their places must survive the overwrite
int order_before_bulk_write(olap_store *s) {
  if (!s->track_order || !s->maybe_nonempty) return OLAP_OK;  // nothing was ever written: the new cells arrive in ascending order
  return seq_materialise(s);
}

int order_after_bulk_write(olap_store *s) {
  s->hi_index = s->size ? s->size - 1 : 0;
  const bool had = s->maybe_nonempty;
  s->maybe_nonempty = true;
  if (!s->track_order || !s->seq) return OLAP_OK;
  (void)had;
  int rc = seq_renumber_if_needed(s, s->size + 1);
  if (rc) return rc;
  ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((seq_after_bulk_kernel<T>), grid_for_n(s->size), kBlock, 0, nullptr, (const T *)s->values, mask_needed(s),
                                              s->seq, s->size, s->default_kind == OLAP_DEFAULT_NAN, (uint32_t)s->next_seq));
  s->next_seq += s->size + 1;
  return launched("seq_after_bulk_kernel");
}

// olap_store_set_value is about to write cell `index`: while the order is still implicit (ascending) it stays so
// only if the cell lies above every cell that may be set — appending at the end is what ascending insertion does.
// Otherwise the order becomes explicit BEFORE the write, so that a cell which is already a key keeps its place
// and one which is not is appended (Map.set, in-memory.js:132).
int order_before_set_value(olap_store *s, uint64_t index) {
  if (!s->track_order || s->seq) return OLAP_OK;
  if (!s->maybe_nonempty || index > s->hi_index) return OLAP_OK;
  return seq_materialise(s);
}

int order_after_set_value(olap_store *s, uint64_t index) {
  s->maybe_nonempty = true;
  if (!s->track_order || !s->seq) {
    s->hi_index = std::max(s->hi_index, index);
    return OLAP_OK;
  }
  int rc = seq_renumber_if_needed(s, 2);
  if (rc) return rc;
  ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((seq_set_cell_kernel<T>), 1, 1, 0, nullptr, (const T *)s->values, mask_needed(s), s->seq, index,
                                              (uint32_t)s->next_seq, s->default_kind == OLAP_DEFAULT_NAN));
  s->next_seq += 1;
  return launched("seq_set_cell_kernel");
}

// deserialize: the blob lists the cells in Map order (in-memory.js:94-100, :103-116)
int order_after_from_sparse(olap_store *s, const uint32_t *idx, uint64_t n) {
  s->maybe_nonempty = n > 0;
  bool ascending = true;
  for (uint64_t j = 1; j < n && ascending; ++j) ascending = idx[j] > idx[j - 1];
  s->hi_index = n ? *std::max_element(idx, idx + n) : 0;
  if (ascending) return OLAP_OK;
  // a blob written in another order carries that order: keep it (turns tracking on for this store)
  if (s->size >= 0x7FFFFFFFull) return OLAP_OK;
  s->track_order = true;
  std::vector<uint32_t> host(s->size, 0u);
  for (uint64_t j = n; j-- > 0;)
    if (idx[j] < s->size) host[idx[j]] = (uint32_t)j + 1;  // first occurrence wins, as Map.set keeps an existing key in place
  int rc = seq_alloc(s);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(s->seq, host.data(), s->size * sizeof(uint32_t), hipMemcpyHostToDevice));
  s->next_seq = n + 1;
  return seq_mask(s, nullptr);  // cells whose value is the default were never set (setValue deletes)
}

int order_sorted_keys(const olap_store *s, std::vector<uint64_t> &keys) {
  std::vector<uint32_t> host(std::max<uint64_t>(s->size, 1));
  HIP_TRY(hipMemcpy(host.data(), s->seq, s->size * sizeof(uint32_t), hipMemcpyDeviceToHost));
  keys.clear();
  for (uint64_t i = 0; i < s->size; ++i)
    if (host[i]) keys.push_back(i);
  std::sort(keys.begin(), keys.end(), [&](uint64_t a, uint64_t b) { return host[a] < host[b]; });
  return OLAP_OK;
}

// drillUp of a tracked store by sorting its set cells by (output cell, seq) and replaying every output cell's run: any
// rule, any maps, the reference's values and key order exactly (s->seq is materialised)
static int order_drillup_replay(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                                const uint32_t *const *maps, int method) {
  uint64_t n_in = 1, n_out = 1;
  for (int d = 0; d < ndim; ++d) {
    if (old_len[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
    for (uint32_t k = 0; k < old_len[d]; ++k)
      if (maps[d][k] >= new_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillUp map of dimension %d: entry %u = %u is outside the new dimension (%u items)", d, k, maps[d][k], new_len[d]);
    n_in *= old_len[d];
    n_out *= new_len[d];
  }
  if (n_in != s->size) return fail(OLAP_ERR_LENGTH_MISMATCH, "store holds %llu cells but the dimensions describe %llu", (unsigned long long)s->size, (unsigned long long)n_in);
  if (n_in >= 0x7FFFFFFFull || n_out >= 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "insertion-order tracking supports stores below 2^31 cells");
  olap_store *o = nullptr;
  int rc = store_alloc(&o, n_out, s->dtype, s->default_kind);
  if (rc) return rc;
  inherit(s, o);
  o->next_seq = s->next_seq;
  if ((rc = seq_alloc(o))) {
    olap_store_destroy(o);
    return rc;
  }
  const int def_nan = s->default_kind == OLAP_DEFAULT_NAN;
  ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((replay_clear_kernel<T>), grid_for_n(n_out), kBlock, 0, nullptr, (T *)o->values, o->status, o->seq, n_out, def_nan));
  if (n_in == 0 || n_out == 0) {
    rc = launched("replay_clear_kernel");
    if (!rc) {
      hipError_t e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) rc = hip_fail(e, "replay_clear_kernel");
    }
    if (rc) olap_store_destroy(o);
    else *out = o;
    return rc;
  }
  ReplayDims dims{};
  dims.nd = ndim;
  std::vector<uint32_t> tab;
  {
    uint64_t stride = 1;
    for (int d = ndim - 1; d >= 0; --d) {
      dims.old_len[d] = old_len[d];
      dims.new_stride[d] = stride;
      stride *= new_len[d];
    }
    for (int d = 0; d < ndim; ++d) {
      dims.tab_off[d] = (uint32_t)tab.size();
      tab.insert(tab.end(), maps[d], maps[d] + old_len[d]);
    }
    if (tab.empty()) tab.push_back(0);
  }
  void *dev_tab = nullptr, *keys_a = nullptr, *keys_b = nullptr, *vals_a = nullptr, *vals_b = nullptr, *tmp = nullptr;
  size_t tmp_bytes = 0;
  // only the bits that differ need sorting: the seq (32) and the output cell's
  int cell_bits = 1;
  while ((n_out - 1) >> cell_bits) ++cell_bits;
  const int end_bit = 32 + cell_bits + 1;  // (+1: the all-ones key of an unset cell must sort last)
  hipError_t e = dev_alloc(&dev_tab, tab.size() * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemcpy(dev_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = dev_alloc(&keys_a, n_in * sizeof(uint64_t));
  if (e == hipSuccess) e = dev_alloc(&keys_b, n_in * sizeof(uint64_t));
  if (e == hipSuccess) e = dev_alloc(&vals_a, n_in * sizeof(uint32_t));
  if (e == hipSuccess) e = dev_alloc(&vals_b, n_in * sizeof(uint32_t));
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const uint64_t *)keys_a, (uint64_t *)keys_b, (const uint32_t *)vals_a, (uint32_t *)vals_b,
                                           (unsigned int)n_in, 0, end_bit > 64 ? 64 : end_bit, (hipStream_t) nullptr);
  if (e == hipSuccess) e = dev_alloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (e == hipSuccess) {
    dims.tab = (const uint32_t *)dev_tab;
    hipLaunchKernelGGL(replay_keys_kernel, grid_for_n(n_in), kBlock, 0, nullptr, (const uint32_t *)s->seq, n_in, dims, (uint64_t *)keys_a, (uint32_t *)vals_a);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const uint64_t *)keys_a, (uint64_t *)keys_b, (const uint32_t *)vals_a, (uint32_t *)vals_b,
                                           (unsigned int)n_in, 0, end_bit > 64 ? 64 : end_bit, (hipStream_t) nullptr);
  if (e == hipSuccess) {
    ORDER_DISPATCH(s->dtype, launch_replay<T>(method, (const T *)s->values, (const uint64_t *)keys_b, (const uint32_t *)vals_b, n_in, (T *)o->values, o->status,
                                               o->seq, def_nan));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // the scratch goes back to the pool
  for (void *q : {dev_tab, keys_a, keys_b, vals_a, vals_b, tmp})
    if (q) dev_free(q);
  if (e != hipSuccess) {
    olap_store_destroy(o);
    return hip_fail(e, "drillUp in insertion order (sort + replay)");
  }
  *out = o;
  return OLAP_OK;
}

int order_drillup(const olap_store *s, olap_store **out, int ndim, const uint32_t *old_len, const uint32_t *new_len,
                  const uint32_t *const *maps, int method) {
  *out = nullptr;
  const bool pick_by_order = (method == OLAP_FIRST || method == OLAP_LAST) && s->seq != nullptr;
  int rc;
  if (s->seq != nullptr && (method == OLAP_SUM || method == OLAP_AVERAGE || method == OLAP_PRODUCT)) {
    // these rules depend on the order of their contributions: replay them in it unless it is the flat index after all
    bool ascending = true;
    if ((rc = seq_is_ascending(s, &ascending))) return rc;
    if (!ascending) return order_drillup_replay(s, out, ndim, old_len, new_len, maps, method);
  }
  if (!pick_by_order) {
    // values: the ordinary kernels (first / last over an ascending store ARE by flat index)
    if ((rc = store_drillup_plain(s, out, ndim, old_len, new_len, maps, method))) return rc;
    olap_store *o = *out;
    inherit(s, o);
    // order of the result: each output cell sits where its first contributing cell sat
    if ((rc = seq_materialise(s)) || (rc = seq_alloc(o))) {
      olap_store_destroy(o);
      *out = nullptr;
      return rc;
    }
    olap_plan *plan = nullptr;
    rc = olap_drillup_plan(&plan, OLAP_UINT32, OLAP_DEFAULT_ZERO, OLAP_LOWEST, ndim, old_len, new_len, maps);
    if (!rc) rc = olap_plan_run(plan, s->seq, nullptr, o->seq, nullptr, nullptr);
    if (plan) olap_plan_destroy(plan);
    if (!rc) rc = seq_mask(o, nullptr);  // a result that equals the default is not a key (sum hitting 0, ...)
    o->next_seq = s->next_seq;
    if (rc) {
      olap_store_destroy(o);
      *out = nullptr;
    }
    return rc;
  }
  // first / last by insertion order: one rolled-up dimension (all Cube.drillUp ever asks for, src/cube.js:999-1000)
  int changed = -1;
  for (int d = 0; d < ndim; ++d) {
    if (old_len[d] && !maps[d]) return fail(OLAP_ERR_INVALID_ARGUMENT, "maps[%d] is NULL", d);
    bool ident = old_len[d] == new_len[d];
    for (uint32_t k = 0; k < old_len[d]; ++k) {
      if (maps[d][k] >= new_len[d])
        return fail(OLAP_ERR_INDEX_RANGE, "drillUp map of dimension %d: entry %u = %u is outside the new dimension (%u items)", d, k, maps[d][k], new_len[d]);
      ident = ident && maps[d][k] == k;
    }
    if (!ident) {
      if (changed >= 0) return order_drillup_replay(s, out, ndim, old_len, new_len, maps, method);  // several rolled-up dimensions at once
      changed = d;
    }
  }
  uint64_t outer = 1, inner = 1, K = 1, G = 1, n_in = 1, n_out = 1;
  for (int d = 0; d < ndim; ++d) {
    n_in *= old_len[d];
    n_out *= new_len[d];
  }
  if (n_in != s->size) return fail(OLAP_ERR_LENGTH_MISMATCH, "store holds %llu cells but the dimensions describe %llu", (unsigned long long)s->size, (unsigned long long)n_in);
  std::vector<uint32_t> tab;
  if (changed >= 0) {
    for (int d = 0; d < changed; ++d) outer *= old_len[d];
    for (int d = changed + 1; d < ndim; ++d) inner *= old_len[d];
    K = old_len[changed];
    G = new_len[changed];
    std::vector<uint32_t> gstart(G + 1, 0), order(K);
    for (uint32_t k = 0; k < K; ++k) gstart[maps[changed][k] + 1]++;
    for (uint64_t g = 0; g < G; ++g) gstart[g + 1] += gstart[g];
    std::vector<uint32_t> cur(gstart.begin(), gstart.end() - 1);
    for (uint32_t k = 0; k < K; ++k) order[cur[maps[changed][k]]++] = k;
    tab = gstart;
    tab.insert(tab.end(), order.begin(), order.end());
  } else {
    inner = n_in;
    tab = {0, 1, 0};
  }
  olap_store *o = nullptr;
  if ((rc = store_alloc(&o, n_out, s->dtype, s->default_kind))) return rc;
  inherit(s, o);
  uint32_t *dev_tab = nullptr;
  hipError_t e = dev_alloc((void **)&dev_tab, tab.size() * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemcpy(dev_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
  rc = e == hipSuccess ? seq_alloc(o) : hip_fail(e, "hipMalloc(drillUp by order)");
  if (!rc) {
    ORDER_DISPATCH(s->dtype, hipLaunchKernelGGL((drillup_byseq_kernel<T>), grid_for_n(n_out), kBlock, 0, nullptr, (const T *)s->values, mask_needed(s), s->seq,
                                                (T *)o->values, o->status, o->seq, outer, K, G, inner, dev_tab, dev_tab + G + 1, method == OLAP_LAST,
                                                s->default_kind == OLAP_DEFAULT_NAN));
    rc = launched("drillup_byseq_kernel");
  }
  if (!rc) {
    e = hipStreamSynchronize(nullptr);  // dev_tab goes back to the pool
    if (e != hipSuccess) rc = hip_fail(e, "drillup_byseq_kernel");
  }
  if (dev_tab) dev_free(dev_tab);
  o->next_seq = s->next_seq;
  if (rc) {
    olap_store_destroy(o);
    return rc;
  }
  *out = o;
  return OLAP_OK;
}

int order_after_dice(const olap_store *s, olap_store *o, int ndim, const uint32_t *old_len, const uint32_t *new_len, const int32_t *const *sel) {
  inherit(s, o);
  if (!s->seq) {
    bool monotone = true;  // new index increasing in old index: the surviving cells keep their (ascending) order
    for (int d = 0; d < ndim && monotone; ++d) monotone = strictly_increasing(sel[d], new_len[d]);
    if (monotone) return OLAP_OK;
  }
  int rc;
  if ((rc = seq_materialise(s)) || (rc = seq_alloc(o))) return rc;
  olap_plan *plan = nullptr;
  rc = olap_dice_plan(&plan, OLAP_UINT32, OLAP_DEFAULT_ZERO, ndim, old_len, new_len, sel);
  if (!rc) rc = olap_plan_run(plan, s->seq, nullptr, o->seq, nullptr, nullptr);
  if (plan) olap_plan_destroy(plan);
  o->next_seq = s->next_seq;
  return rc;
}

int order_after_reorder(const olap_store *s, olap_store *o, int ndim, const uint32_t *old_len, const int32_t *perm) {
  inherit(s, o);
  bool identity = true;
  for (int d = 0; d < ndim; ++d) identity = identity && perm[d] == d;
  if (identity && !s->seq) return OLAP_OK;
  int rc;
  if ((rc = seq_materialise(s)) || (rc = seq_alloc(o))) return rc;
  olap_plan *plan = nullptr;
  rc = olap_reorder_plan(&plan, OLAP_UINT32, OLAP_DEFAULT_ZERO, ndim, old_len, perm);
  if (!rc) rc = olap_plan_run(plan, s->seq, nullptr, o->seq, nullptr, nullptr);
  if (plan) olap_plan_destroy(plan);
  o->next_seq = s->next_seq;
  return rc;
}

int order_after_drilldown(const olap_store *s, olap_store *o) {
  inherit(s, o);  // the reference visits the new cells in ascending index order (:383): no seq
  return OLAP_OK;
}

int order_before_load(olap_store *mine) {
  if (!mine->track_order || !mine->maybe_nonempty) return OLAP_OK;
  return seq_materialise(mine);
}

int order_after_load(olap_store *mine, const olap_store *his, int ndim, const uint32_t *my_len, const uint32_t *his_len,
                     const int32_t *const *his_to_mine) {
  const bool was_fresh = !mine->maybe_nonempty;
  mine->maybe_nonempty = true;
  mine->hi_index = mine->size ? mine->size - 1 : 0;
  if (!mine->track_order) return OLAP_OK;
  if (was_fresh && !mine->seq) {
    // every key is new and arrives in HIS ascending index order (:159): ascending in MY index iff the remap is monotone
    bool monotone = true;
    for (int d = 0; d < ndim && monotone; ++d) monotone = strictly_increasing(his_to_mine[d], his_len[d]);
    if (monotone) return OLAP_OK;
  }
  if (his->size >= 0x7FFFFFFFull) return fail(OLAP_ERR_INVALID_ARGUMENT, "insertion-order tracking supports stores below 2^31 cells");
  int rc = seq_alloc(mine);
  if (rc) return rc;
  if (was_fresh) HIP_TRY(hipMemsetAsync(mine->seq, 0, std::max<uint64_t>(mine->size, 1) * sizeof(uint32_t), nullptr));
  if ((rc = seq_renumber_if_needed(mine, his->size + 1))) return rc;
  // where each of HIS cells lands, as the position it would be appended at: the load plan over a ramp
  uint32_t *ramp = nullptr, *landed = nullptr;
  HIP_TRY(dev_alloc((void **)&ramp, std::max<uint64_t>(his->size, 1) * sizeof(uint32_t)));
  hipError_t e = dev_alloc((void **)&landed, std::max<uint64_t>(mine->size, 1) * sizeof(uint32_t));
  if (e != hipSuccess) {
    dev_free(ramp);
    return hip_fail(e, "hipMalloc(load order)");
  }
  hipLaunchKernelGGL(seq_ramp_kernel, grid_for_n(his->size), kBlock, 0, nullptr, ramp, his->size, (uint32_t)mine->next_seq);
  e = hipMemsetAsync(landed, 0, std::max<uint64_t>(mine->size, 1) * sizeof(uint32_t), nullptr);
  olap_plan *plan = nullptr;
  rc = e == hipSuccess ? olap_load_plan(&plan, OLAP_UINT32, OLAP_DEFAULT_ZERO, OLAP_DEFAULT_ZERO, ndim, my_len, his_len, his_to_mine) : hip_fail(e, "hipMemset");
  if (!rc) rc = olap_plan_run(plan, ramp, nullptr, landed, nullptr, nullptr);
  if (plan) olap_plan_destroy(plan);
  // cells that were keys keep their place, cells the load created take the landing position, cells it unset lose theirs
  if (!rc) rc = seq_mask(mine, landed);
  mine->next_seq += his->size + 1;
  hipError_t e2 = hipStreamSynchronize(nullptr);
  dev_free(ramp);
  dev_free(landed);
  if (!rc && e2 != hipSuccess) rc = hip_fail(e2, "load order");
  return rc;
}
