"""Multi-GPU host: one process per GPU, the cube partitioned along its outermost dimension.

Row-major layout makes a dim0 partition a set of contiguous slabs (src/cube.js:709-728: last
dimension fastest): rank r owns rows [row_lo, row_hi) of dimension 0, i.e. the flat range
[row_lo*inner0, row_hi*inner0).  Every store operation that leaves dimension 0 alone runs per
shard with no communication.  drillUp ON dimension 0 (in-memory.js:265-334 with a non-identity
map on the sharded axis) reduces each rank's own rows into a partial [G, inner0] cube with the
same kernel and a row sub-map, then ONE collective combines the partials:

  sum      -> reduce-scatter (all-reduce when the output does not divide) with ncclSum over xGMI
  average  -> the same on (float sum, int32 contribution count) pairs produced in ONE local pass
              (OLAP_PARTIAL_AVERAGE), divided afterwards (olap_average_finish)
  other    -> all-gather of (value, status) partials, then the same drillUp kernel over the rank
              axis: highest / lowest / first / last / product are associative in rank order

torch.distributed provides the process group only (backend "nccl" is RCCL on ROCm; with "gloo"
device tensors are staged through host memory, for rehearsals).  All cell arithmetic goes through
an engine object; the package ships exactly one, HipEngine (libolapgpu).  CPU tests inject their
own engine to rehearse the partition/collective logic without a GPU.
"""
import numpy as np

from . import capi
from .hipstore import Plan


def partition_rows(n_rows, world):
    """Contiguous, balanced split of dimension 0: the first (n_rows % world) ranks get one more."""
    base, extra = divmod(int(n_rows), int(world))
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < extra else 0))
    return bounds


class _HipDrillUp:
    def __init__(self, engine, plan):
        self.engine, self.plan = engine, plan

    def run(self, values, status, out_values, out_status):
        self.plan.run(values.data_ptr(), status.data_ptr() if status is not None else None, out_values.data_ptr(),
                      out_status.data_ptr() if out_status is not None else None, self.engine.stream())


class HipEngine:
    """Cell arithmetic on the current HIP device; torch tensors are the device-memory plumbing."""

    name = "hip"

    def __init__(self, device):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        if capi.lib().olap_device_count() < 1:
            raise capi.OlapError(capi.ERR_NO_DEVICE, "no HIP device available: libolapgpu has no CPU fallback")
        torch.cuda.set_device(self.device)

    def empty(self, n, dtype):
        t = self.torch
        td = {"float32": t.float32, "float64": t.float64, "int32": t.int32, "uint32": t.int32}[dtype]
        return t.empty(max(int(n), 1), dtype=td, device=self.device)[: int(n)]

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def make_drillup(self, dtype, default, method, old_len, new_len, maps):
        return _HipDrillUp(self, Plan.drillup(dtype, default, method, old_len, new_len, maps))

    def make_dice(self, dtype, default, old_len, new_len, sel):
        return _HipDrillUp(self, Plan.dice(dtype, default, old_len, new_len, sel))

    def make_drilldown(self, dtype, default, method, old_len, new_len, maps):
        return _HipDrillUp(self, Plan.drilldown(dtype, default, method, old_len, new_len, maps))

    def fill_seeded(self, values, status, n, first_cell, dtype, seed, frac):
        capi.check(capi.lib().olap_fill_seeded(values.data_ptr(), status.data_ptr(), int(n), int(first_cell),
                                               capi.DTYPES[dtype], int(seed), float(frac), self.stream()))

    def average_finish(self, values, counts, status, dtype, default):
        kind = capi.DEFAULT_NAN if default != default else capi.DEFAULT_ZERO
        capi.check(capi.lib().olap_average_finish(values.data_ptr(), counts.data_ptr(),
                                                  status.data_ptr() if status is not None else None, values.numel(),
                                                  capi.DTYPES[dtype], kind, self.stream()))


class ShardedStore:
    """One measure of a cube whose dimension 0 is split across the ranks of a process group."""

    def __init__(self, lens, dtype="float32", default=0.0, rank=0, world=1, engine=None, group=None, bounds=None):
        if engine is None:
            raise ValueError("an engine is required (HipEngine on a GPU box)")
        self.lens = [int(x) for x in lens]
        self.dtype, self.default = dtype, default
        self.rank, self.world, self.group = int(rank), int(world), group
        self.engine = engine
        # `bounds`: an explicit (possibly uneven) row partition, e.g. what a dice of dimension 0 leaves
        self.bounds = partition_rows(self.lens[0], self.world) if bounds is None else [int(b) for b in bounds]
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or self.bounds[-1] != self.lens[0]:
            raise ValueError("bounds must list world+1 ascending row offsets from 0 to the extent of dimension 0")
        self.row_lo, self.row_hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self.inner0 = int(np.prod(self.lens[1:])) if len(self.lens) > 1 else 1
        self.local_cells = (self.row_hi - self.row_lo) * self.inner0
        self.values = engine.empty(self.local_cells, dtype)
        self.status = engine.empty(self.local_cells, "int32")

    @property
    def local_lens(self):
        return [self.row_hi - self.row_lo] + self.lens[1:]

    def fill_seeded(self, seed=20240807, frac=1.0):
        """SURVEY §8(d) synthetic measure; each rank generates its own slab of the global stream."""
        self.engine.fill_seeded(self.values, self.status, self.local_cells, self.row_lo * self.inner0, self.dtype,
                                seed, frac)
        return self

    def drillup_other_axis(self, axis, axis_map, n_groups, method="sum"):
        """drillUp on a non-sharded axis: per shard, no communication; the partition is kept."""
        if axis < 1:
            raise ValueError("use plan_drillup_dim0 for the sharded axis")
        old_len = self.local_lens
        new_len = list(old_len)
        new_len[axis] = int(n_groups)
        maps = [np.arange(l, dtype=np.uint32) for l in old_len]
        maps[axis] = np.asarray(axis_map, dtype=np.uint32)
        out = ShardedStore([self.lens[0]] + new_len[1:], self.dtype, self.default, self.rank, self.world, self.engine,
                           self.group)
        op = self.engine.make_drillup(self.dtype, self.default, method, old_len, new_len, maps)
        op.run(self.values, None, out.values, out.status)
        out._keepalive = op
        return out

    def _local(self, op, new_tail, bounds=None, new_rows=None):
        """Runs a per-shard plan (no communication) into a new store with the same partition."""
        rows = self.lens[0] if new_rows is None else new_rows
        out = ShardedStore([rows] + list(new_tail), self.dtype, self.default, self.rank, self.world, self.engine, self.group,
                           self.bounds if bounds is None else bounds)
        op.run(self.values, None, out.values, out.status)
        out._keepalive = op
        return out

    def dice_other_axes(self, selections):
        """dice on non-sharded dimensions (in-memory.js:213-263): selections[d] lists the OLD item index
        of every new item of dimension d (-1: unknown item), None keeps a dimension; selections[0]
        must be None.  Per shard, no communication, partition kept."""
        if selections[0] is not None:
            raise ValueError("dimension 0 is the sharded axis: use dice_dim0")
        old_len = self.local_lens
        sel = [np.arange(l, dtype=np.int32) if s is None else np.asarray(s, dtype=np.int32) for l, s in zip(old_len, selections)]
        new_len = [len(x) for x in sel]
        return self._local(self.engine.make_dice(self.dtype, self.default, old_len, new_len, sel), new_len[1:])

    def drilldown_other_axis(self, axis, child_to_parent, method="sum"):
        """drillDown of a non-sharded dimension (in-memory.js:336-430): child_to_parent[new item] = old
        item.  Per shard, no communication, partition kept."""
        if axis < 1:
            raise ValueError("refining the sharded axis changes the partition: not provided")
        old_len = self.local_lens
        child = np.asarray(child_to_parent, dtype=np.uint32)
        new_len = list(old_len)
        new_len[axis] = len(child)
        maps = [np.arange(l, dtype=np.uint32) for l in old_len]
        maps[axis] = child
        return self._local(self.engine.make_drilldown(self.dtype, self.default, method, old_len, new_len, maps), new_len[1:])

    def dice_dim0(self, rows):
        """dice of the sharded dimension by an ASCENDING list of global rows: every rank keeps those of
        its own rows that were selected — no data moves, the partition becomes uneven.  (A selection
        that reorders rows across ranks would need an all-to-all; the reference's dice keeps item
        order unless asked otherwise, src/cube.js:821-857.)"""
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size and (np.any(np.diff(rows) <= 0) or rows[0] < 0 or rows[-1] >= self.lens[0]):
            raise ValueError("dice_dim0 takes strictly ascending row indices inside dimension 0")
        bounds = [int(np.searchsorted(rows, b)) for b in self.bounds]
        mine = rows[bounds[self.rank]:bounds[self.rank + 1]] - self.row_lo
        old_len = self.local_lens
        sel = [mine.astype(np.int32)] + [np.arange(l, dtype=np.int32) for l in old_len[1:]]
        new_len = [len(mine)] + old_len[1:]
        return self._local(self.engine.make_dice(self.dtype, self.default, old_len, new_len, sel), self.lens[1:], bounds, len(rows))

    def plan_drillup_dim0(self, row_map, n_groups, method="sum", always_collective=False):
        """Prepares drillUp of the sharded axis: row_map[global row] -> group (< n_groups)."""
        return Dim0DrillUp(self, row_map, n_groups, method, always_collective)


class Dim0DrillUp:
    """Reusable step: local partial + one collective.  `step()` is what bench.py times at N > 1."""

    def __init__(self, store, row_map, n_groups, method="sum", always_collective=False):
        import torch.distributed as dist

        self.dist = dist
        self.s = s = store
        self.method = method
        # always_collective: run the collective even in a one-rank group (lets a single-GPU box
        # exercise the RCCL code path; with one rank every collective is the identity)
        self.collective = store.world > 1 or always_collective
        row_map = np.asarray(row_map, dtype=np.uint32)
        if row_map.size != s.lens[0]:
            raise ValueError("row_map must have one entry per row of dimension 0")
        n_groups = int(n_groups)
        old_len = s.local_lens
        new_len = [n_groups] + s.lens[1:]
        maps = [row_map[s.row_lo:s.row_hi]] + [np.arange(l, dtype=np.uint32) for l in s.lens[1:]]
        eng = s.engine
        w = s.world
        self.n_out = n_groups * s.inner0
        self.additive = method in ("sum", "average")
        local_method = capi.PARTIAL_AVERAGE if (method == "average" and self.collective) else method
        self.local = eng.make_drillup(s.dtype, s.default, local_method, old_len, new_len, maps)
        self.partial = eng.empty(self.n_out, s.dtype)
        # `sum` over a zero default: the mask is a function of the value (set <=> value != 0), so the
        # path neither writes nor ships it.  `average` ships contribution counts in its place.
        zero_default = not (s.default != s.default)
        self.partial_status = None if (method == "sum" and zero_default) else eng.empty(self.n_out, "int32")
        self.scatter = self.additive and self.collective and self.n_out % w == 0
        self.staged = self.collective and dist.get_backend(s.group) == "gloo" and getattr(self.partial, "is_cuda", False)
        if not self.collective:
            self.result, self.result_status = self.partial, self.partial_status
        elif self.additive:
            n_res = self.n_out // w if self.scatter else self.n_out
            self.result = eng.empty(n_res, s.dtype)
            self.result_status = None if self.partial_status is None else eng.empty(n_res, "int32")
            self.result_counts = eng.empty(n_res, "int32") if method == "average" else None
        else:
            self.gathered = eng.empty(self.n_out * w, s.dtype)
            self.gathered_status = eng.empty(self.n_out * w, "int32")
            self.result = eng.empty(self.n_out, s.dtype)
            self.result_status = eng.empty(self.n_out, "int32")
            self.combine = eng.make_drillup(s.dtype, s.default, method, [w, self.n_out], [1, self.n_out],
                                            [np.zeros(w, np.uint32), np.arange(self.n_out, dtype=np.uint32)])

    @property
    def result_range(self):
        """Flat range of the global output held in `result` on this rank."""
        if self.scatter:
            per = self.n_out // self.s.world
            return self.s.rank * per, (self.s.rank + 1) * per
        return 0, self.n_out

    def _sum_across_ranks(self, src, dst):
        """dst <- element-wise sum of every rank's src (scattered when the output divides)."""
        s, dist = self.s, self.dist
        a = src.cpu() if self.staged else src
        b = a.new_empty(dst.numel()) if self.staged else dst
        if self.scatter and dist.get_backend(s.group) != "gloo":
            dist.reduce_scatter_tensor(b, a, op=dist.ReduceOp.SUM, group=s.group)
        elif self.scatter:  # gloo has no reduce_scatter: all-reduce, keep this rank's slice
            full = a.clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=s.group)
            lo, hi = self.result_range
            b.copy_(full[lo:hi])
        else:
            b.copy_(a)
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=s.group)
        if self.staged:
            dst.copy_(b)

    # ---- pipelined form for streams of independent queries (bench.py at N > 1) ----------------
    def step_pipelined(self):
        """Same work as step() for `sum`, but the collective is asynchronous and ping-pongs between two
        (partial, result) buffer pairs: the reduce-scatter of query i runs on RCCL's stream while the
        local reduction of query i+1 runs on the compute stream.  Returns the result tensor of THIS
        query; it is complete after its work handle (or flush()) has been waited for."""
        s, dist = self.s, self.dist
        if not self.collective or self.method != "sum" or self.partial_status is not None or self.staged:
            return self.step()
        if not hasattr(self, "_pipe"):
            eng = s.engine
            self._pipe = {"i": 0, "partial": [self.partial, eng.empty(self.n_out, s.dtype)],
                          "result": [self.result, eng.empty(self.result.numel(), s.dtype)], "work": [None, None]}
        p = self._pipe
        k = p["i"] & 1
        p["i"] += 1
        if p["work"][k] is not None:
            p["work"][k].wait()  # the collective that last read partial[k] / wrote result[k]
        self.local.run(s.values, None, p["partial"][k], None)
        p["work"][k] = self._sum_across_ranks_async(p["partial"][k], p["result"][k])
        return p["result"][k]

    def _sum_across_ranks_async(self, src, dst):
        """Asynchronous form of _sum_across_ranks; returns an object with wait()."""
        s, dist = self.s, self.dist
        if self.scatter and dist.get_backend(s.group) != "gloo":
            return dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=s.group, async_op=True)
        if self.scatter:  # gloo rehearsal: all-reduce a copy, keep this rank's slice when it lands
            full = src.clone()
            work = dist.all_reduce(full, op=dist.ReduceOp.SUM, group=s.group, async_op=True)
            lo, hi = self.result_range

            class _Slice:
                def wait(self_inner):
                    work.wait()
                    dst.copy_(full[lo:hi])

            return _Slice()
        dst.copy_(src)
        return dist.all_reduce(dst, op=dist.ReduceOp.SUM, group=s.group, async_op=True)

    def flush(self):
        """Waits (on the current stream) for every outstanding pipelined collective."""
        if hasattr(self, "_pipe"):
            for k in (0, 1):
                if self._pipe["work"][k] is not None:
                    self._pipe["work"][k].wait()
                    self._pipe["work"][k] = None

    def step(self):
        s, dist = self.s, self.dist
        self.local.run(s.values, None, self.partial, self.partial_status)
        if not self.collective:
            return self.result
        if self.additive:
            self._sum_across_ranks(self.partial, self.result)
            if self.method == "average":
                self._sum_across_ranks(self.partial_status, self.result_counts)
                s.engine.average_finish(self.result, self.result_counts, self.result_status, s.dtype, s.default)
            elif self.partial_status is not None:  # sum over a NaN default: set where any rank was set
                self._sum_across_ranks(self.partial_status, self.result_status)
            return self.result
        if self.staged:
            g = self.partial.cpu().new_empty(self.n_out * s.world)
            g_st = self.partial_status.cpu().new_empty(self.n_out * s.world)
            dist.all_gather_into_tensor(g, self.partial.cpu(), group=s.group)
            dist.all_gather_into_tensor(g_st, self.partial_status.cpu(), group=s.group)
            self.gathered.copy_(g)
            self.gathered_status.copy_(g_st)
        else:
            dist.all_gather_into_tensor(self.gathered, self.partial, group=s.group)
            dist.all_gather_into_tensor(self.gathered_status, self.partial_status, group=s.group)
        self.combine.run(self.gathered, self.gathered_status, self.result, self.result_status)
        return self.result
