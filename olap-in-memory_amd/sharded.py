"""Multi-GPU host: a thin ctypes view of libolapgpu's sharded entry points (include/olap_hip.h,
"Multi-GPU"; implementation olap-in-memory_amd/csrc/olap_sharded.hip).

The cube is partitioned along its outermost dimension (row-major layout, src/cube.js:709-728, makes
the slabs contiguous).  Everything that matters lives behind the C ABI — the row partition, the row
sub-maps, what a sharded drillUp ships (olap_shard_recipe), the RCCL collectives and the finishing
kernels — so that the Node.js host reaches it too.  What is left here:

  Comm           olap_comm: init_rank (one process per GPU; the unique id travels over
                 torch.distributed's store), init_all (one process, a device list), detached
  ShardedStore   olap_sharded_store: one measure, dimension 0 split over the ranks
  ShardDrillUp   olap_shard_drillup: the reusable "local partial + one collective + finish" step
                 that bench.py times at N > 1
  exchange_over_process_group
                 rehearsal aid: moves a detached op's payloads through host memory and a
                 torch.distributed group (gloo), so N processes sharing ONE GPU can run the N > 1
                 code path; RCCL refuses two ranks on one device
  HipEngine      torch tensors as device-memory plumbing for bench.py and tools/
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import check
from .hipstore import HipStore, Plan, _default_kind, _method_code, _tables, _u32

NP_OF_DTYPE = {0: np.int32, 1: np.uint32, 2: np.float32, 3: np.float64}
NAME_OF_DTYPE = {0: "int32", 1: "uint32", 2: "float32", 3: "float64"}


def partition_rows(n_rows, world):
    """olap_shard_bounds: contiguous, balanced split of dimension 0 (the first n_rows % world ranks get one more)."""
    b = np.zeros(int(world) + 1, np.uint32)
    check(capi.lib().olap_shard_bounds(int(n_rows), int(world), b.ctypes.data_as(capi._pu32)))
    return [int(x) for x in b]


def dice_bounds(bounds, rows):
    """olap_shard_dice_bounds: the partition left by a dice of dimension 0 with ascending `rows`."""
    b = _u32(bounds)
    r = np.ascontiguousarray(np.asarray(rows, dtype=np.int32))
    out = np.zeros(b.size, np.uint32)
    check(capi.lib().olap_shard_dice_bounds(b.ctypes.data_as(capi._pu32), b.size - 1, r.ctypes.data_as(capi._pi32), r.size,
                                            out.ctypes.data_as(capi._pu32)))
    return [int(x) for x in out]


def recipe(dtype, default, method):
    """olap_shard_recipe_get as a dict (what one sharded drillUp of dimension 0 ships, and how it is combined)."""
    r = capi.ShardRecipe()
    check(capi.lib().olap_shard_recipe_get(capi.DTYPES[dtype], _default_kind(default), _method_code(method), C.byref(r)))
    return {"local_method": r.local_method, "zero_unset": bool(r.zero_unset), "n_payloads": r.n_payloads,
            "payload_dtype": list(r.payload_dtype), "payload_op": list(r.payload_op), "finish": r.finish}


class Comm:
    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
        check(capi.lib().olap_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, unique_id, world, rank, device):
        h = C.c_void_p()
        check(capi.lib().olap_comm_init_rank(C.byref(h), bytes(unique_id), int(world), int(rank), int(device)))
        return cls(h)

    @classmethod
    def init_all(cls, devices):
        d = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = C.c_void_p()
        check(capi.lib().olap_comm_init_all(C.byref(h), d, len(devices)))
        return cls(h)

    @classmethod
    def detached(cls, world, rank, device=0):
        h = C.c_void_p()
        check(capi.lib().olap_comm_init_detached(C.byref(h), int(world), int(rank), int(device)))
        return cls(h)

    @classmethod
    def from_process_group(cls, dist, device, group=None):
        """One process per GPU: rank 0 makes the RCCL unique id, torch.distributed carries it.  Either every rank
        returns a communicator or every rank raises (a failure on one rank is agreed on before anybody blocks in RCCL)."""
        import torch

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        on_gpu = dist.get_backend(group) == "nccl"
        where = torch.device("cuda", int(device)) if on_gpu else "cpu"
        # 1. can EVERY rank bind RCCL at all?  (ncclCommInitRank is collective: a rank that cannot even load the
        #    library must not leave the others waiting inside it)
        my_id, failure = None, None
        try:
            my_id = cls.unique_id()
        except capi.OlapError as err:
            failure = err
        ok = torch.tensor([0 if failure else 1], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise failure or capi.OlapError(capi.ERR_NO_DEVICE, "another rank could not load RCCL")
        # 2. rank 0's id reaches everybody
        box = [my_id if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        comm, failure = None, None
        try:
            comm = cls.init_rank(box[0], world, rank, device)
        except capi.OlapError as err:
            failure = err
        ok = torch.tensor([0 if failure else 1], dtype=torch.int32, device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.destroy()
            raise failure or capi.OlapError(capi.ERR_HIP, "another rank could not create its RCCL communicator")
        return comm

    @property
    def world(self):
        return capi.lib().olap_comm_world(self._h)

    @property
    def local_count(self):
        return capi.lib().olap_comm_local_count(self._h)

    def local_rank(self, i=0):
        return capi.lib().olap_comm_local_rank(self._h, i)

    def local_device(self, i=0):
        return capi.lib().olap_comm_local_device(self._h, i)

    @property
    def transport(self):
        return capi.lib().olap_comm_transport(self._h).decode()

    def destroy(self):
        if self._h:
            capi.lib().olap_comm_destroy(self._h)
            self._h = None


def _ptr_array(ptrs):
    arr = (C.c_void_p * max(len(ptrs), 1))()
    for i, p in enumerate(ptrs):
        arr[i] = p or None
    return arr


class ShardDrillUp:
    """olap_shard_drillup: drillUp of the sharded dimension as local partial + ONE collective + finish."""

    def __init__(self, comm, dtype, default, method, lens, new_len, bounds, maps, placement=capi.PLACE_SCATTER, depth=1):
        self.comm = comm
        self.dtype = dtype
        ol, nl, bd = _u32(lens), _u32(new_len), _u32(bounds)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        h = C.c_void_p()
        check(capi.lib().olap_shard_drillup_create(C.byref(h), comm._h, capi.DTYPES[dtype], _default_kind(default),
                                                   _method_code(method), len(ol), ol.ctypes.data_as(capi._pu32),
                                                   nl.ctypes.data_as(capi._pu32), bd.ctypes.data_as(capi._pu32), arr,
                                                   int(placement), int(depth)))
        self._h = h
        self.recipe = recipe(dtype, default, method)
        # gathered partials are combined on every rank: their result is whole everywhere
        self.placement = capi.PLACE_ALL if self.recipe["finish"] == capi.FINISH_COMBINE else int(placement)

    @property
    def out_cells(self):
        return int(capi.lib().olap_shard_drillup_out_cells(self._h))

    def local_cells(self, i=0):
        return int(capi.lib().olap_shard_drillup_local_cells(self._h, i))

    def kernel_name(self, i=0):
        return capi.lib().olap_shard_drillup_kernel_name(self._h, i).decode()

    def step(self, in_values, in_status=None, streams=None):
        """in_values / in_status / streams: one device address per local rank."""
        v = _ptr_array(in_values)
        s = _ptr_array(in_status) if in_status is not None else None
        st = _ptr_array(streams) if streams is not None else None
        check(capi.lib().olap_shard_drillup_step(self._h, v, s, st))

    def wait(self, streams=None):
        check(capi.lib().olap_shard_drillup_wait(self._h, _ptr_array(streams) if streams is not None else None))

    def local(self, i, in_values, in_status=None, stream=None):
        check(capi.lib().olap_shard_drillup_local(self._h, i, in_values or None, in_status or None, stream or None))

    def exchange(self, streams=None):
        check(capi.lib().olap_shard_drillup_exchange(self._h, _ptr_array(streams) if streams is not None else None))

    def finish(self, i, stream=None):
        check(capi.lib().olap_shard_drillup_finish(self._h, i, stream or None))

    def payload(self, i, p):
        send, recv, n, dt, op = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_int(), C.c_int()
        check(capi.lib().olap_shard_drillup_payload(self._h, i, p, C.byref(send), C.byref(recv), C.byref(n), C.byref(dt), C.byref(op)))
        return send.value, recv.value, n.value, dt.value, op.value

    def result(self, i=0):
        """-> (values address, status address or None, first flat cell, cell count) of the last step."""
        v, s, f, n = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(capi.lib().olap_shard_drillup_result(self._h, i, C.byref(v), C.byref(s), C.byref(f), C.byref(n)))
        return v.value, s.value, f.value, n.value

    def result_host(self, i=0):
        """(values, status or None, first) of the last step copied to host arrays (blocking)."""
        v, s, f, n = self.result(i)
        vals = np.zeros(max(n, 1), NP_OF_DTYPE[capi.DTYPES[self.dtype]])
        if n:
            check(capi.lib().olap_memcpy_to_host(vals.ctypes.data_as(C.c_void_p), v, n * vals.itemsize))
        stat = None
        if s and n:
            stat = np.zeros(n, np.int32)
            check(capi.lib().olap_memcpy_to_host(stat.ctypes.data_as(C.c_void_p), s, n * 4))
        return vals[:n], stat, f

    def destroy(self):
        if self._h:
            capi.lib().olap_shard_drillup_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def exchange_over_process_group(op, dist, group=None):
    """Rehearsal aid for a DETACHED communicator: plays olap_shard_drillup_exchange() through host memory and
    a torch.distributed group (gloo), honouring the op's placement by filling `recv` exactly as RCCL would."""
    import torch

    L = capi.lib()
    check(L.olap_device_synchronize())
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    for p in range(op.recipe["n_payloads"]):
        send, recv, n, dt, xop = op.payload(0, p)
        host = np.zeros(max(n, 1), NP_OF_DTYPE[dt])
        check(L.olap_memcpy_to_host(host.ctypes.data_as(C.c_void_p), send, n * host.itemsize))
        wire = host.view(np.int32) if dt == capi.DTYPES["uint32"] else host  # gloo has no uint32; sums wrap alike
        t = torch.from_numpy(wire)
        if xop == capi.XCHG_GATHER:
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t, group=group)
            full = np.concatenate([o.numpy() for o in outs])
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM if xop == capi.XCHG_SUM else dist.ReduceOp.MAX, group=group)
            full = t.numpy()
            if op.placement in (capi.PLACE_SCATTER, capi.PLACE_SCATTER_ROWS):  # this rank keeps its own block of the padded payload
                per = n // world
                full = full[rank * per:(rank + 1) * per]
        if recv:
            full = np.ascontiguousarray(full)
            check(L.olap_memcpy_to_device(recv, full.ctypes.data_as(C.c_void_p), full.size * full.itemsize))


class ShardedStore:
    """olap_sharded_store: one measure of a cube whose dimension 0 is split over the ranks of a Comm."""

    def __init__(self, comm, lens, dtype="float32", default=0.0, bounds=None, _handle=None):
        self.comm = comm
        self._lib = capi.lib()
        if _handle is not None:
            self._h = _handle
        else:
            ol = _u32(lens)
            bd = _u32(bounds) if bounds is not None else None
            h = C.c_void_p()
            check(self._lib.olap_sharded_store_create(C.byref(h), comm._h, len(ol), ol.ctypes.data_as(capi._pu32), capi.DTYPES[dtype],
                                                      _default_kind(default), bd.ctypes.data_as(capi._pu32) if bd is not None else None))
            self._h = h
        self.dtype = dtype
        self.default = default

    def _wrap(self, handle):
        return ShardedStore(self.comm, None, self.dtype, self.default, _handle=handle)

    def __del__(self):
        try:
            if self._h:
                self._lib.olap_sharded_store_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def size(self):
        return int(self._lib.olap_sharded_store_size(self._h))

    @property
    def lens(self):
        n = self._lib.olap_sharded_store_ndim(self._h)
        p = self._lib.olap_sharded_store_lens(self._h)
        return [int(p[i]) for i in range(n)]

    @property
    def bounds(self):
        p = self._lib.olap_sharded_store_bounds(self._h)
        return [int(p[i]) for i in range(self.comm.world + 1)]

    @property
    def inner0(self):
        return int(np.prod(self.lens[1:])) if len(self.lens) > 1 else 1

    def shard(self, i=0):
        """Borrowed HipStore view of local rank i's slab (do not destroy)."""
        h = self._lib.olap_sharded_store_shard(self._h, i)
        s = HipStore(0, _handle=C.c_void_p(h))
        s._borrowed = self  # keeps the owner alive
        s.__class__ = _BorrowedStore
        return s

    def local_range(self, i=0):
        b, r = self.bounds, self.comm.local_rank(i)
        return b[r] * self.inner0, b[r + 1] * self.inner0

    def fill_seeded(self, seed=20240807, frac=1.0):
        check(self._lib.olap_sharded_store_fill_seeded(self._h, int(seed), float(frac)))
        return self

    def set_data_f64(self, values):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
        check(self._lib.olap_sharded_store_set_data_f64(self._h, v.ctypes.data_as(capi._pdbl), v.size))
        return self

    def get_data_f64(self):
        """Full-size array; only the rows of this process' ranks are filled (the rest stay NaN)."""
        out = np.full(max(self.size, 1), np.nan)
        check(self._lib.olap_sharded_store_get_data_f64(self._h, out.ctypes.data_as(capi._pdbl)))
        return out[: self.size]

    def get_status(self):
        out = np.full(max(self.size, 1), -1, np.int32)
        check(self._lib.olap_sharded_store_get_status(self._h, out.ctypes.data_as(capi._pi32)))
        return out[: self.size]

    def get_value(self, index):
        v, s = C.c_double(), C.c_int()
        check(self._lib.olap_sharded_store_get_value(self._h, int(index), C.byref(v), C.byref(s)))
        return v.value, bool(s.value)

    def set_value(self, index, value):
        if value is None:
            check(self._lib.olap_sharded_store_set_value(self._h, int(index), 0.0, 1))
        else:
            check(self._lib.olap_sharded_store_set_value(self._h, int(index), float(value), 0))

    def fill(self, value):
        check(self._lib.olap_sharded_store_fill(self._h, float(value)))

    @property
    def total(self):
        t = C.c_double()
        check(self._lib.olap_sharded_store_total(self._h, C.byref(t)))
        return t.value

    def clone(self):
        h = C.c_void_p()
        check(self._lib.olap_sharded_store_clone(self._h, C.byref(h)))
        return self._wrap(h)

    def gather(self):
        h = C.c_void_p()
        check(self._lib.olap_sharded_store_gather(self._h, C.byref(h)))
        return HipStore(0, _handle=h)

    @classmethod
    def scatter(cls, comm, whole, lens):
        ol = _u32(lens)
        h = C.c_void_p()
        check(capi.lib().olap_sharded_store_scatter(C.byref(h), comm._h, whole._h, len(ol), ol.ctypes.data_as(capi._pu32)))
        return cls(comm, None, whole.type, float("nan") if whole.default_is_nan else 0.0, _handle=h)

    # ---- bulk operations (names follow in-memory.js)
    def drill_up(self, new_len, maps, method="sum"):
        """-> ShardedStore (dimension 0 untouched, or rolled up to at least one row per rank: the result stays sharded
        along the new leading dimension) or HipStore (dimension 0 rolled up to fewer rows than ranks, e.g. 'all')."""
        nl = _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        hs, hw = C.c_void_p(), C.c_void_p()
        check(self._lib.olap_sharded_store_drillup(self._h, C.byref(hs), C.byref(hw), nl.ctypes.data_as(capi._pu32), arr, _method_code(method)))
        return self._wrap(hs) if hs.value else HipStore(0, _handle=hw)

    def dice(self, new_len, sel):
        nl = _u32(new_len)
        keep, arr = _tables(sel, np.int32, C.c_int32)
        h = C.c_void_p()
        check(self._lib.olap_sharded_store_dice(self._h, C.byref(h), nl.ctypes.data_as(capi._pu32), arr))
        return self._wrap(h)

    def drill_down(self, new_len, maps, method="sum", distributions=None, integer_measure=False):
        nl = _u32(new_len)
        keep, arr = _tables(maps, np.uint32, C.c_uint32)
        if distributions is not None:
            d = np.ascontiguousarray(np.asarray(distributions, dtype=np.float64))
            dp, dn = d.ctypes.data_as(capi._pdbl), d.size
        else:
            dp, dn = None, 0
        h = C.c_void_p()
        check(self._lib.olap_sharded_store_drilldown(self._h, C.byref(h), nl.ctypes.data_as(capi._pu32), arr,
                                                     _method_code(method) | (capi.DRILLDOWN_INTEGER_MEASURE if integer_measure else 0), dp, dn))
        return self._wrap(h)

    def reorder(self, perm):
        p = np.ascontiguousarray(np.asarray(perm, dtype=np.int32))
        h = C.c_void_p()
        check(self._lib.olap_sharded_store_reorder(self._h, C.byref(h), p.ctypes.data_as(capi._pi32)))
        return self._wrap(h)

    def plan_drillup_dim0(self, row_map, n_groups, method="sum", placement=capi.PLACE_SCATTER, depth=1):
        """The reusable step bench.py times: drillUp of dimension 0 with `row_map`, other dimensions kept."""
        lens = self.lens
        maps = [np.asarray(row_map, dtype=np.uint32)] + [np.arange(l, dtype=np.uint32) for l in lens[1:]]
        return ShardDrillUp(self.comm, self.dtype, self.default, method, lens, [int(n_groups)] + lens[1:], self.bounds, maps, placement, depth)

    def step_inputs(self):
        """(values addresses, status addresses) of the local slabs, as ShardDrillUp.step() takes them."""
        n = self.comm.local_count
        primary = self.default != self.default and self.dtype in ("int32", "uint32")
        vals = [self._lib.olap_store_values_ptr(self._lib.olap_sharded_store_shard(self._h, i)) for i in range(n)]
        stat = [self._lib.olap_store_status_ptr(self._lib.olap_sharded_store_shard(self._h, i)) for i in range(n)] if primary else None
        return vals, stat


class _BorrowedStore(HipStore):
    def __del__(self):  # the slab belongs to its ShardedStore
        self._h = None


class _HipDrillUp:
    def __init__(self, engine, plan):
        self.engine, self.plan = engine, plan

    def run(self, values, status, out_values, out_status):
        self.plan.run(values.data_ptr(), status.data_ptr() if status is not None else None, out_values.data_ptr(),
                      out_status.data_ptr() if out_status is not None else None, self.engine.stream())


class HipEngine:
    """torch tensors as device-memory plumbing for raw-pointer plans (bench.py, tools/)."""

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
        capi.check(capi.lib().olap_fill_seeded(values.data_ptr(), status.data_ptr() if status is not None else None, int(n),
                                               int(first_cell), capi.DTYPES[dtype], int(seed), float(frac), self.stream()))
