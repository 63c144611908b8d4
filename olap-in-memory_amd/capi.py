"""ctypes declarations for include/olap_hip.h (every exported symbol is declared here)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))

DTYPES = {"int32": 0, "uint32": 1, "float32": 2, "float64": 3}
DTYPE_NAMES = {v: k for k, v in DTYPES.items()}
DTYPE_SIZE = {0: 4, 1: 4, 2: 4, 3: 8}
METHODS = {"sum": 0, "average": 1, "highest": 2, "lowest": 3, "first": 4, "last": 5, "product": 6}
PARTIAL_AVERAGE = 7  # shard-local half of `average` (include/olap_hip.h)
DEFAULT_ZERO, DEFAULT_NAN = 0, 1
STATUS_SET = 0x2

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_INVALID_TYPE = -2
ERR_INVALID_DEFAULT = -3
ERR_UNSUPPORTED_METHOD = -4
ERR_LENGTH_MISMATCH = -5
ERR_DISTRIBUTION_MISSING = -6
ERR_NO_DEVICE = -7
ERR_HIP = -8
ERR_OUT_OF_MEMORY = -9
ERR_INDEX_RANGE = -10


class OlapError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def lib_path():
    # OLAP_LIBOLAPGPU: another build of the same library (the sanitizer build of tests/_plan_dry_worker.py)
    return os.environ.get("OLAP_LIBOLAPGPU") or os.path.join(HERE, "lib", "libolapgpu.so")


_vp, _u64, _i32, _dbl, _sz = C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_size_t
_pu32, _pi32, _pdbl, _pu64 = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint64)
_ppu32, _ppi32 = C.POINTER(_pu32), C.POINTER(_pi32)
_pvp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list exactly the functions include/olap_hip.h declares
SIGNATURES = {
    "olap_last_error": (C.c_char_p, []),
    "olap_abi_version": (_i32, []),
    "olap_method_from_name": (_i32, [C.c_char_p]),
    "olap_dtype_from_name": (_i32, [C.c_char_p]),
    "olap_dtype_size": (_sz, [_i32]),
    "olap_device_count": (_i32, []),
    "olap_set_device": (_i32, [_i32]),
    "olap_device_synchronize": (_i32, []),
    "olap_drillup_plan": (_i32, [_pvp, _i32, _i32, _i32, _i32, _pu32, _pu32, _ppu32]),
    "olap_drilldown_plan": (_i32, [_pvp, _i32, _i32, _i32, _i32, _pu32, _pu32, _ppu32, _pdbl, _u64]),
    "olap_dice_plan": (_i32, [_pvp, _i32, _i32, _i32, _pu32, _pu32, _ppi32]),
    "olap_dice_drillup_plan": (_i32, [_pvp, _i32, _i32, _i32, _i32, _pu32, _pu32, _pu32, _ppi32, _ppu32]),
    "olap_reorder_plan": (_i32, [_pvp, _i32, _i32, _i32, _pu32, _pi32]),
    "olap_load_plan": (_i32, [_pvp, _i32, _i32, _i32, _i32, _pu32, _pu32, _ppi32]),
    "olap_plan_in_cells": (_u64, [_vp]),
    "olap_plan_out_cells": (_u64, [_vp]),
    "olap_plan_kernel_name": (C.c_char_p, [_vp]),
    "olap_plan_run": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "olap_diag_tile_placement": (_i32, [_i32, C.c_uint32, C.c_uint32, C.c_uint32, _pu32, _pu32, _pu32, _pu32]),
    "olap_plan_run_batch": (_i32, [_vp, _i32, _pvp, _pvp, _pvp, _pvp, _vp]),
    "olap_plan_run_batch_rules": (_i32, [_vp, _i32, C.POINTER(C.c_int), _pvp, _pvp, _pvp, _pvp, _vp]),
    "olap_plan_status": (_i32, [_vp]),
    "olap_plan_destroy": (None, [_vp]),
    "olap_canonicalize": (_i32, [_vp, _vp, _u64, _i32, _i32, _i32, _vp]),
    "olap_convert_from_f64": (_i32, [_vp, _vp, _vp, _u64, _i32, _i32, _vp]),
    "olap_convert_to_f64": (_i32, [_vp, _vp, _u64, _i32, _vp]),
    "olap_fill_seeded": (_i32, [_vp, _vp, _u64, _u64, _i32, C.c_uint32, _dbl, _vp]),
    "olap_average_finish": (_i32, [_vp, _vp, _vp, _u64, _i32, _i32, _vp]),
    "olap_eval_formula": (_i32, [_pi32, _i32, _pdbl, _i32, _i32, _pvp, _pvp, C.POINTER(C.c_int), C.POINTER(C.c_int), _pdbl, _i32, _vp, _u64, _vp]),
    "olap_store_eval_formula": (_i32, [_pi32, _i32, _pdbl, _i32, _i32, _pvp, _pdbl, _i32, _pdbl]),
    "olap_total": (_i32, [_vp, _vp, _u64, _i32, _i32, _pdbl, _pu64, _vp]),
    "olap_store_create": (_i32, [_pvp, _u64, _i32, _i32]),
    "olap_store_destroy": (None, [_vp]),
    "olap_store_clone": (_i32, [_vp, _pvp]),
    "olap_store_size": (_u64, [_vp]),
    "olap_store_dtype": (_i32, [_vp]),
    "olap_store_default": (_i32, [_vp]),
    "olap_store_byte_length": (_u64, [_vp]),
    "olap_store_values_ptr": (_vp, [_vp]),
    "olap_store_status_ptr": (_vp, [_vp]),
    "olap_store_track_order": (_i32, [_vp, _i32]),
    "olap_store_order_tracked": (_i32, [_vp]),
    "olap_store_set_data": (_i32, [_vp, _vp, _u64]),
    "olap_store_set_data_f64": (_i32, [_vp, _pdbl, _u64]),
    "olap_store_get_data": (_i32, [_vp, _vp]),
    "olap_store_get_data_f64": (_i32, [_vp, _pdbl]),
    "olap_store_get_status": (_i32, [_vp, _pi32]),
    "olap_store_count_set": (_i32, [_vp, _pu64]),
    "olap_store_get_keys": (_i32, [_vp, _pu64, _u64, _pu64]),
    "olap_store_get_value": (_i32, [_vp, _u64, _pdbl, C.POINTER(C.c_int)]),
    "olap_store_set_value": (_i32, [_vp, _u64, _dbl, _i32]),
    "olap_store_fill": (_i32, [_vp, _dbl]),
    "olap_store_total": (_i32, [_vp, _pdbl]),
    "olap_store_to_sparse": (_i32, [_vp, _pu32, _vp, _u64, _pu64]),
    "olap_store_from_sparse": (_i32, [_pvp, _u64, _i32, _i32, _pu32, _vp, _u64]),
    "olap_store_totals": (_i32, [_vp, _i32, _pu32, C.POINTER(C.c_int), _pdbl, _pi32, C.POINTER(C.c_int), _pu64]),
    "olap_store_drillup": (_i32, [_vp, _pvp, _i32, _pu32, _pu32, _ppu32, _i32]),
    "olap_store_drillup_batch": (_i32, [_i32, _pvp, _pvp, _i32, _pu32, _pu32, _ppu32, _i32]),
    "olap_store_drillup_multi": (_i32, [_i32, _pvp, C.POINTER(C.c_int), _pvp, _i32, _pu32, _pu32, _ppu32]),
    "olap_store_drilldown": (_i32, [_vp, _pvp, _i32, _pu32, _pu32, _ppu32, _i32, _pdbl, _u64]),
    "olap_store_dice": (_i32, [_vp, _pvp, _i32, _pu32, _pu32, _ppi32]),
    "olap_store_dice_drillup": (_i32, [_vp, _pvp, _i32, _pu32, _pu32, _pu32, _ppi32, _ppu32, _i32]),
    "olap_store_reorder": (_i32, [_vp, _pvp, _i32, _pu32, _pi32]),
    "olap_store_load": (_i32, [_vp, _vp, _i32, _pu32, _pu32, _ppi32]),
    "olap_memcpy_to_host": (_i32, [_vp, _vp, _u64]),
    "olap_memcpy_to_device": (_i32, [_vp, _vp, _u64]),
    "olap_diag_read_ceiling": (_i32, [_vp, _u64, _vp, _vp]),
    "olap_diag_write_ceiling": (_i32, [_vp, _u64, _vp]),
    # multi-GPU (include/olap_hip.h, "Multi-GPU")
    "olap_comm_unique_id": (_i32, [C.c_char_p]),
    "olap_comm_init_rank": (_i32, [_pvp, C.c_char_p, _i32, _i32, _i32]),
    "olap_comm_init_all": (_i32, [_pvp, C.POINTER(C.c_int), _i32]),
    "olap_comm_init_detached": (_i32, [_pvp, _i32, _i32, _i32]),
    "olap_comm_destroy": (None, [_vp]),
    "olap_comm_world": (_i32, [_vp]),
    "olap_comm_local_count": (_i32, [_vp]),
    "olap_comm_local_rank": (_i32, [_vp, _i32]),
    "olap_comm_local_device": (_i32, [_vp, _i32]),
    "olap_comm_transport": (C.c_char_p, [_vp]),
    "olap_shard_bounds": (_i32, [C.c_uint32, _i32, _pu32]),
    "olap_shard_dice_bounds": (_i32, [_pu32, _i32, _pi32, C.c_uint32, _pu32]),
    "olap_shard_recipe_get": (_i32, [_i32, _i32, _i32, _vp]),
    "olap_shard_drillup_create": (_i32, [_pvp, _vp, _i32, _i32, _i32, _i32, _pu32, _pu32, _pu32, _ppu32, _i32, _i32]),
    "olap_shard_drillup_destroy": (None, [_vp]),
    "olap_shard_drillup_out_cells": (_u64, [_vp]),
    "olap_shard_drillup_local_cells": (_u64, [_vp, _i32]),
    "olap_shard_drillup_kernel_name": (C.c_char_p, [_vp, _i32]),
    "olap_shard_drillup_step": (_i32, [_vp, _pvp, _pvp, _pvp]),
    "olap_shard_drillup_wait": (_i32, [_vp, _pvp]),
    "olap_shard_drillup_local": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "olap_shard_drillup_exchange": (_i32, [_vp, _pvp]),
    "olap_shard_drillup_finish": (_i32, [_vp, _i32, _vp]),
    "olap_shard_drillup_payload": (_i32, [_vp, _i32, _i32, _pvp, _pvp, _pu64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "olap_shard_drillup_result": (_i32, [_vp, _i32, _pvp, _pvp, _pu64, _pu64]),
    "olap_sharded_store_create": (_i32, [_pvp, _vp, _i32, _pu32, _i32, _i32, _pu32]),
    "olap_sharded_store_destroy": (None, [_vp]),
    "olap_sharded_store_size": (_u64, [_vp]),
    "olap_sharded_store_ndim": (_i32, [_vp]),
    "olap_sharded_store_lens": (_pu32, [_vp]),
    "olap_sharded_store_bounds": (_pu32, [_vp]),
    "olap_sharded_store_reshape": (_i32, [_vp, _i32, _pu32]),
    "olap_sharded_store_comm": (_vp, [_vp]),
    "olap_sharded_store_shard": (_vp, [_vp, _i32]),
    "olap_sharded_store_fill_seeded": (_i32, [_vp, C.c_uint32, _dbl]),
    "olap_sharded_store_set_data_f64": (_i32, [_vp, _pdbl, _u64]),
    "olap_sharded_store_get_data_f64": (_i32, [_vp, _pdbl]),
    "olap_sharded_store_get_status": (_i32, [_vp, _pi32]),
    "olap_sharded_store_get_value": (_i32, [_vp, _u64, _pdbl, C.POINTER(C.c_int)]),
    "olap_sharded_store_set_value": (_i32, [_vp, _u64, _dbl, _i32]),
    "olap_sharded_store_fill": (_i32, [_vp, _dbl]),
    "olap_sharded_store_total": (_i32, [_vp, _pdbl]),
    "olap_sharded_store_eval_formula": (_i32, [_pi32, _i32, _pdbl, _i32, _i32, _pvp, _pdbl, _i32, _pdbl]),
    "olap_sharded_store_clone": (_i32, [_vp, _pvp]),
    "olap_sharded_store_gather": (_i32, [_vp, _pvp]),
    "olap_sharded_store_scatter": (_i32, [_pvp, _vp, _vp, _i32, _pu32]),
    "olap_sharded_store_drillup": (_i32, [_vp, _pvp, _pvp, _pu32, _ppu32, _i32]),
    "olap_sharded_store_dice": (_i32, [_vp, _pvp, _pu32, _ppi32]),
    "olap_sharded_store_drilldown": (_i32, [_vp, _pvp, _pu32, _ppu32, _i32, _pdbl, _u64]),
    "olap_sharded_store_reorder": (_i32, [_vp, _pvp, _pi32]),
}


class ShardRecipe(C.Structure):
    """olap_shard_recipe"""
    _fields_ = [("local_method", C.c_int), ("zero_unset", C.c_int), ("n_payloads", C.c_int),
                ("payload_dtype", C.c_int * 2), ("payload_op", C.c_int * 2), ("finish", C.c_int)]


XCHG_SUM, XCHG_MAX, XCHG_GATHER = 0, 1, 2
FINISH_NONE, FINISH_RESTORE, FINISH_AVERAGE, FINISH_COMBINE, FINISH_ROUND = 0, 1, 2, 3, 4
DRILLDOWN_INTEGER_MEASURE = 0x100  # OR into drillDown's method: int32 / uint32 measure held in float64 cells
PLACE_SCATTER, PLACE_ALL, PLACE_ROOT, PLACE_SCATTER_ROWS = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128

_lib = None


def lib():
    """Loads lib/libolapgpu.so; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise OlapError(ERR_NO_DEVICE, "libolapgpu.so is not built (%s): run __graft_entry__.build()" % path)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def last_error():
    return lib().olap_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        raise OlapError(rc, last_error())
    return rc
