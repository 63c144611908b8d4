/*
 * olap_oracle.c — CPU ORACLE (test infrastructure, not product code).  See olap_oracle.h.
 *
 * Restates /root/reference/src/store/in-memory.js function by function; every routine
 * cites the lines it follows.  Arithmetic is IEEE float64 throughout, as in JavaScript.
 * Parity status: PINNED by tests/test_oracle_golden.py against tests/golden/ (vectors
 * produced by running the reference itself, oracle/gen_golden.js).
 */
#include "olap_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TOMBSTONE UINT64_MAX

struct oracle_store {
  uint64_t size;
  int type;
  int default_is_nan;
  double *val;   /* value per flat index (meaningful only where pos >= 0) */
  int64_t *pos;  /* position of the key in `log`, or -1 when the key is absent */
  uint64_t *log; /* insertion log: keys in insertion order, TOMBSTONE = deleted */
  uint64_t log_len, log_cap;
  uint64_t n_keys;
};

/* ------------------------------------------------------------------ Map emulation */
static void map_set(oracle_store *s, uint64_t k, double v) {
  if (s->pos[k] >= 0) { /* Map.set on an existing key keeps its position */
    s->val[k] = v;
    return;
  }
  if (s->log_len == s->log_cap) {
    uint64_t cap = s->log_cap ? s->log_cap * 2 : 64;
    s->log = (uint64_t *)realloc(s->log, cap * sizeof(uint64_t));
    s->log_cap = cap;
  }
  s->pos[k] = (int64_t)s->log_len;
  s->log[s->log_len++] = k;
  s->val[k] = v;
  s->n_keys++;
}

static void map_delete(oracle_store *s, uint64_t k) {
  if (s->pos[k] < 0) return;
  s->log[s->pos[k]] = TOMBSTONE;
  s->pos[k] = -1;
  s->n_keys--;
}

static inline int map_has(const oracle_store *s, uint64_t k) { return s->pos[k] >= 0; }

/* ------------------------------------------------------------------ in-memory.js:48-64 */
oracle_store *oracle_store_new(uint64_t size, int type, int default_is_nan) {
  oracle_store *s = (oracle_store *)calloc(1, sizeof(*s));
  s->size = size;
  s->type = type;
  s->default_is_nan = default_is_nan;
  s->val = (double *)malloc((size ? size : 1) * sizeof(double));
  s->pos = (int64_t *)malloc((size ? size : 1) * sizeof(int64_t));
  for (uint64_t i = 0; i < size; ++i) s->pos[i] = -1;
  return s;
}

void oracle_store_free(oracle_store *s) {
  if (!s) return;
  free(s->val);
  free(s->pos);
  free(s->log);
  free(s);
}

/* :66-73 — new Map(oldMap) re-inserts the live entries in order */
oracle_store *oracle_store_clone(const oracle_store *s) {
  oracle_store *c = oracle_store_new(s->size, s->type, s->default_is_nan);
  for (uint64_t p = 0; p < s->log_len; ++p)
    if (s->log[p] != TOMBSTONE) map_set(c, s->log[p], s->val[s->log[p]]);
  return c;
}

uint64_t oracle_size(const oracle_store *s) { return s->size; }
int oracle_type(const oracle_store *s) { return s->type; }
int oracle_default_is_nan(const oracle_store *s) { return s->default_is_nan; }
uint64_t oracle_num_keys(const oracle_store *s) { return s->n_keys; }

void oracle_entries(const oracle_store *s, uint64_t *keys, double *values) {
  uint64_t n = 0;
  for (uint64_t p = 0; p < s->log_len; ++p) {
    uint64_t k = s->log[p];
    if (k == TOMBSTONE) continue;
    keys[n] = k;
    values[n] = s->val[k];
    ++n;
  }
}

static inline double default_value(const oracle_store *s) { return s->default_is_nan ? NAN : 0.0; }

/* :30-37 */
void oracle_dense(const oracle_store *s, double *values, uint8_t *present) {
  const double d = default_value(s);
  for (uint64_t i = 0; i < s->size; ++i) {
    int has = map_has(s, i);
    if (values) values[i] = has ? s->val[i] : d;
    if (present) present[i] = (uint8_t)has;
  }
}

/* :22-28 — sum over Map.values() in insertion order */
double oracle_total(const oracle_store *s) {
  double t = 0;
  for (uint64_t p = 0; p < s->log_len; ++p)
    if (s->log[p] != TOMBSTONE) t += s->val[s->log[p]];
  return t;
}

/* :118-120 */
double oracle_get_value(const oracle_store *s, uint64_t index) {
  return map_has(s, index) ? s->val[index] : default_value(s);
}

/* :122-133 — store unless value === default or (default is NaN and value is NaN) */
void oracle_set_value(oracle_store *s, uint64_t index, double value) {
  int is_default;
  if (s->default_is_nan)
    is_default = isnan(value);
  else
    is_default = (value == 0.0); /* `value !== 0` is false for both +0 and -0 */
  if (!is_default)
    map_set(s, index, value);
  else
    map_delete(s, index);
}

void oracle_unset_value(oracle_store *s, uint64_t index) { map_delete(s, index); }

/* :39-46 (length check is the caller's) */
void oracle_set_data(oracle_store *s, const double *values) {
  for (uint64_t i = 0; i < s->size; ++i) oracle_set_value(s, i, values[i]);
}

/* :135-137 */
void oracle_fill(oracle_store *s, double value) {
  for (uint64_t i = 0; i < s->size; ++i) oracle_set_value(s, i, value);
}

/* mulberry32, same stream as oracle/gen_golden.js */
static inline double mulberry32_next(uint32_t *state) {
  uint32_t a = (*state += 0x6D2B79F5u);
  uint32_t t = (a ^ (a >> 15)) * (1u | a);
  t = (t + ((t ^ (t >> 7)) * (61u | t))) ^ t;
  return (double)(t ^ (t >> 14)) / 4294967296.0;
}

void oracle_fill_seeded(oracle_store *s, uint32_t seed, double frac) {
  uint32_t st = seed;
  for (uint64_t i = 0; i < s->size; ++i) {
    double v = (double)(float)(0.5 + mulberry32_next(&st)); /* Math.fround */
    int keep = mulberry32_next(&st) < frac;
    if (keep) oracle_set_value(s, i, v);
  }
}

/* Math.max / Math.min: NaN-propagating, +0 > -0 */
static inline double js_max(double a, double b) {
  if (isnan(a) || isnan(b)) return NAN;
  if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
  return a > b ? a : b;
}
static inline double js_min(double a, double b) {
  if (isnan(a) || isnan(b)) return NAN;
  if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
  return a < b ? a : b;
}

static uint64_t product_u32(const uint32_t *v, int n) {
  uint64_t p = 1;
  for (int i = 0; i < n; ++i) p *= v[i];
  return p;
}

/* ------------------------------------------------------------------ in-memory.js:265-334 */
oracle_store *oracle_drillup(const oracle_store *s, int ndim, const uint32_t *old_len,
                             const uint32_t *new_len, const uint32_t *maps, int method) {
  if (method < ORACLE_SUM || method > ORACLE_PRODUCT) return NULL; /* :294-296 throws */
  const uint64_t new_size = product_u32(new_len, ndim);                                  /* :266 */
  oracle_store *out = oracle_store_new(new_size, s->type, s->default_is_nan);            /* :276 */
  uint16_t *contributions = (uint16_t *)calloc(new_size ? new_size : 1, sizeof(uint16_t)); /* :278 */
  uint32_t digit[64];
  const uint32_t *map_of[64];
  {
    const uint32_t *m = maps;
    for (int d = 0; d < ndim; ++d) {
      map_of[d] = m;
      m += old_len[d];
    }
  }

  for (uint64_t p = 0; p < s->log_len; ++p) { /* :298 entries() in insertion order */
    const uint64_t old_idx = s->log[p];
    if (old_idx == TOMBSTONE) continue;
    const double old_value = s->val[old_idx];
    uint64_t c = old_idx; /* :299-303 */
    for (int d = ndim - 1; d >= 0; --d) {
      digit[d] = (uint32_t)(c % old_len[d]);
      c /= old_len[d];
    }
    uint64_t new_idx = 0; /* :305-309 */
    for (int d = 0; d < ndim; ++d) new_idx = new_idx * new_len[d] + map_of[d][digit[d]];

    if (!map_has(out, new_idx)) { /* :311-318 */
      oracle_set_value(out, new_idx, old_value);
    } else {
      const double a = oracle_get_value(out, new_idx), b = old_value;
      double r;
      switch (method) { /* :282-290 */
        case ORACLE_SUM:
        case ORACLE_AVERAGE: r = a + b; break;
        case ORACLE_HIGHEST: r = js_max(a, b); break;
        case ORACLE_LOWEST: r = js_min(a, b); break;
        case ORACLE_FIRST: r = a; break;
        case ORACLE_LAST: r = b; break;
        default: r = a * b; break;
      }
      oracle_set_value(out, new_idx, r);
    }
    contributions[new_idx] += 1; /* :320 — Uint16, wraps at 65536 */
  }

  if (method == ORACLE_AVERAGE) { /* :323-331 */
    for (uint64_t i = 0; i < new_size; ++i)
      if (contributions[i]) oracle_set_value(out, i, oracle_get_value(out, i) / contributions[i]);
  }
  free(contributions);
  return out;
}

/* ------------------------------------------------------------------ in-memory.js:336-430 */
oracle_store *oracle_drilldown(const oracle_store *s, int ndim, const uint32_t *old_len,
                               const uint32_t *new_len, const uint32_t *maps, int method,
                               const double *distributions, uint64_t n_dist,
                               int64_t *missing_index) {
  const int use_rounding = (s->type == ORACLE_INT32 || s->type == ORACLE_UINT32); /* :343 */
  const uint64_t old_size = s->size;                                              /* :344 */
  const uint64_t new_size = product_u32(new_len, ndim);                           /* :345 */
  uint32_t *contrib_ids = (uint32_t *)calloc(old_size ? old_size : 1, sizeof(uint32_t));   /* :356 */
  uint32_t *contrib_total = (uint32_t *)calloc(old_size ? old_size : 1, sizeof(uint32_t)); /* :357 */
  uint64_t *idx_new_old = (uint64_t *)malloc((new_size ? new_size : 1) * sizeof(uint64_t)); /* :359 */
  uint32_t digit[64];
  const uint32_t *map_of[64];
  {
    const uint32_t *m = maps;
    for (int d = 0; d < ndim; ++d) {
      map_of[d] = m;
      m += new_len[d];
    }
  }
  if (missing_index) *missing_index = -1;

  for (uint64_t new_idx = 0; new_idx < new_size; ++new_idx) { /* :361-379 */
    uint64_t c = new_idx;
    for (int d = ndim - 1; d >= 0; --d) {
      digit[d] = (uint32_t)(c % new_len[d]);
      c /= new_len[d];
    }
    uint64_t old_idx = 0;
    for (int d = 0; d < ndim; ++d) old_idx = old_idx * old_len[d] + map_of[d][digit[d]];
    idx_new_old[new_idx] = old_idx;
    contrib_total[old_idx] += 1;
  }

  oracle_store *out = oracle_store_new(new_size, s->type, s->default_is_nan); /* :381 */
  for (uint64_t new_idx = 0; new_idx < new_size; ++new_idx) {                 /* :383-427 */
    const uint64_t old_idx = idx_new_old[new_idx];
    if (!map_has(s, old_idx)) continue; /* :386-387 `if (!oldValue) continue` … */
    const double old_value = s->val[old_idx];
    if (old_value == 0.0 || isnan(old_value)) continue; /* … also skips 0, -0 and NaN */
    const double n = (double)contrib_total[old_idx];

    if (distributions) { /* :391-400 */
      const double added_len = (double)new_size / (double)old_size;
      const double shared = (double)n_dist / added_len;
      const double di = floor((double)new_idx / ((double)new_size / shared)) * added_len +
                        fmod((double)new_idx, added_len);
      const int ok = (di >= 0 && di < (double)n_dist && di == floor(di) && !isnan(distributions[(uint64_t)di]));
      if (!ok) { /* `distributions[distIndex] == null` -> throw */
        if (missing_index) *missing_index = (int64_t)di;
        oracle_store_free(out);
        out = NULL;
        break;
      }
      oracle_set_value(out, new_idx, old_value * distributions[(uint64_t)di]);
    } else if (method == ORACLE_SUM) { /* :402 */
      if (use_rounding) {              /* :403-417 */
        const double value = floor(old_value / n);
        const double remainder = fmod(old_value, n);
        const double cid = (double)contrib_ids[old_idx];
        const double one_over = remainder / n;
        const int last_is_same = floor(cid * one_over) == floor((cid - 1) * one_over);
        const double nv = floor(value);
        oracle_set_value(out, new_idx, last_is_same ? nv : nv + 1);
      } else {
        oracle_set_value(out, new_idx, old_value / n); /* :419 */
      }
    } else {
      oracle_set_value(out, new_idx, old_value); /* :422 */
    }
    contrib_ids[old_idx]++; /* :426 */
  }
  free(contrib_ids);
  free(contrib_total);
  free(idx_new_old);
  return out;
}

/* ------------------------------------------------------------------ in-memory.js:213-263 */
oracle_store *oracle_dice(const oracle_store *s, int ndim, const uint32_t *old_len,
                          const uint32_t *new_len, const int32_t *sel) {
  const uint64_t new_size = product_u32(new_len, ndim); /* :214 */
  /* :219-224 Map(oldIdx -> newIdx); a later duplicate overrides an earlier one */
  int64_t *inv[64];
  {
    const int32_t *t = sel;
    for (int d = 0; d < ndim; ++d) {
      inv[d] = (int64_t *)malloc((old_len[d] ? old_len[d] : 1) * sizeof(int64_t));
      for (uint32_t j = 0; j < old_len[d]; ++j) inv[d][j] = -1;
      for (uint32_t j = 0; j < new_len[d]; ++j)
        if (t[j] >= 0 && (uint32_t)t[j] < old_len[d]) inv[d][t[j]] = j;
      t += new_len[d];
    }
  }
  oracle_store *out = oracle_store_new(new_size, s->type, s->default_is_nan); /* :226-230 */
  uint32_t nd[64];
  for (uint64_t p = 0; p < s->log_len; ++p) { /* :235 */
    const uint64_t old_idx = s->log[p];
    if (old_idx == TOMBSTONE) continue;
    uint64_t c = old_idx;
    int halt = 0;
    for (int d = ndim - 1; d >= 0; --d) { /* :238-251 */
      const uint32_t od = (uint32_t)(c % old_len[d]);
      const int64_t m = inv[d][od];
      if (m < 0) {
        halt = 1;
        break;
      }
      nd[d] = (uint32_t)m;
      c /= old_len[d];
    }
    if (halt) continue;
    uint64_t new_idx = 0; /* :254-257 */
    for (int d = 0; d < ndim; ++d) new_idx = new_idx * new_len[d] + nd[d];
    oracle_set_value(out, new_idx, s->val[old_idx]); /* :259 */
  }
  for (int d = 0; d < ndim; ++d) free(inv[d]);
  return out;
}

/* ------------------------------------------------------------------ in-memory.js:178-211 */
oracle_store *oracle_reorder(const oracle_store *s, int ndim, const uint32_t *old_len,
                             const int32_t *perm) {
  oracle_store *out = oracle_store_new(s->size, s->type, s->default_is_nan); /* :179-183 */
  uint32_t od[64];
  for (uint64_t p = 0; p < s->log_len; ++p) { /* :192 */
    const uint64_t old_idx = s->log[p];
    if (old_idx == TOMBSTONE) continue;
    uint64_t c = old_idx;
    for (int d = ndim - 1; d >= 0; --d) { /* :194-198 */
      od[d] = (uint32_t)(c % old_len[d]);
      c /= old_len[d];
    }
    uint64_t new_idx = 0; /* :201-205 */
    for (int d = 0; d < ndim; ++d) new_idx = new_idx * old_len[perm[d]] + od[perm[d]];
    oracle_set_value(out, new_idx, s->val[old_idx]); /* :207 */
  }
  return out;
}

/* ------------------------------------------------------------------ in-memory.js:139-176 */
void oracle_load(oracle_store *mine, const oracle_store *his, int ndim, const uint32_t *my_len,
                 const uint32_t *his_len, const int32_t *his_to_mine) {
  const int32_t *map_of[64];
  {
    const int32_t *m = his_to_mine;
    for (int d = 0; d < ndim; ++d) {
      map_of[d] = m;
      m += his_len[d];
    }
  }
  uint32_t hd[64];
  for (uint64_t other = 0; other < his->size; ++other) { /* :159 dense over his size */
    uint64_t c = other;
    for (int d = ndim - 1; d >= 0; --d) { /* :161-165 */
      hd[d] = (uint32_t)(c % his_len[d]);
      c /= his_len[d];
    }
    uint64_t my_idx = 0; /* :168-172 */
    int known = 1;
    for (int d = 0; d < ndim; ++d) {
      const int32_t off = map_of[d][hd[d]];
      if (off < 0) { /* reference would compute a NaN key here; reshape() dices first so it never happens */
        known = 0;
        break;
      }
      my_idx = my_idx * my_len[d] + (uint32_t)off;
    }
    if (!known) continue;
    oracle_set_value(mine, my_idx, oracle_get_value(his, other)); /* :174 */
  }
}
