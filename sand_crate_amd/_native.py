"""ctypes binding of include/sandcrate_hip.h.

There is no CPU path: if libsandcrate_hip.so is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libsandcrate_hip.so"

MAX_NEIGHBORS = 20
MAX_SEGMENTS = 16
MAX_BODIES = 8
NUM_KERNELS = 12
NOISE_NONE, NOISE_HOST, NOISE_COUNTER = 0, 1, 2
ERR_ARG = -1
ERR_HIP = -2
ERR_CAPACITY = -3
ERR_STATE = -4
ERR_DOMAIN = -5


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsandcrate_hip: {message} (code {code})")
        self.code = code


class Params(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "dt", "particle_radius", "wall_collision_decay", "pressure_amplifier", "ignored_pressure",
        "collider_noise_level", "viscosity", "surface_smoothing", "target_pressure", "gravity_x", "gravity_y")]


class Body(C.Structure):
    _fields_ = [("position_x", C.c_double), ("position_y", C.c_double),
                ("center_velocity_x", C.c_double), ("center_velocity_y", C.c_double),
                ("angular_clockwise_velocity", C.c_double), ("n_segments", C.c_int32), ("reserved", C.c_int32)]


class TickInputs(C.Structure):
    _fields_ = [("params", Params), ("segments", C.POINTER(C.c_double)), ("padded", C.POINTER(C.c_double)),
                ("bodies", C.POINTER(Body)), ("n_segments", C.c_int32), ("n_bodies", C.c_int32)]


class Source(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("radius", "position_x", "position_y", "velocity_x", "velocity_y", "noise")] + [
        ("flow", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("particles", C.c_int64), ("neighbor_slots", C.c_int64), ("max_neighbors", C.c_int32),
                ("wall_particles", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes): every symbol include/sandcrate_hip.h declares
SIGNATURES = {
    "sc_last_error": (C.c_char_p, []),
    "sc_abi_version": (C.c_int, []),
    "sc_create": (C.c_int, [C.c_int, C.c_int64, C.POINTER(_P)]),
    "sc_destroy": (C.c_int, [_P]),
    "sc_set_stream": (C.c_int, [_P, _P]),
    "sc_use_own_stream": (C.c_int, [_P]),
    "sc_upload_state": (C.c_int, [_P, _D, _D, C.c_int64]),
    "sc_append_particles": (C.c_int, [_P, _D, _D, C.c_int64]),
    "sc_count": (C.c_int, [_P, _I64]),
    "sc_download_state": (C.c_int, [_P, _D, _D, _D, _I64, C.c_int64, _I64]),
    "sc_set_params": (C.c_int, [_P, C.POINTER(Params)]),
    "sc_set_segments": (C.c_int, [_P, _D, _D, C.c_int32, C.POINTER(Body), C.c_int32]),
    "sc_set_noise_mode": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "sc_set_next_inputs": (C.c_int, [_P, C.POINTER(Params), _D, C.c_int32, C.POINTER(Body), C.c_int32]),
    "sc_step_begin": (C.c_int, [_P]),
    "sc_step_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "sc_set_noise_host": (C.c_int, [_P, _D, C.c_int64]),
    "sc_step_finish": (C.c_int, [_P]),
    "sc_step": (C.c_int, [_P, C.c_int32]),
    "sc_tick": (C.c_int, [_P, C.POINTER(TickInputs), C.POINTER(TickInputs)]),
    "sc_synchronize": (C.c_int, [_P]),
    "sc_set_scan_patience": (C.c_int, [_P, C.c_int64]),
    "sc_download_sort": (C.c_int, [_P, _I64, _I64, C.c_int64, _I64]),
    "sc_download_neighbors": (C.c_int, [_P, _I64, _I32, _I64, _D, C.c_int64, _I64]),
    "sc_download_normals": (C.c_int, [_P, _D, C.c_int64, _I64]),
    "sc_neighbor_search": (C.c_int, [C.c_int, _D, C.c_int64, C.c_double, _I64, _I64, _I32, _I64]),
    "sc_points_to_segments": (C.c_int, [C.c_int, _D, C.c_int64, _D, C.c_int32, _D, _D]),
    "sc_pad_segments": (C.c_int, [_D, C.c_int32, C.c_double, _D]),
    "sc_enable_timing": (C.c_int, [_P, C.c_int]),
    "sc_reset_timing": (C.c_int, [_P]),
    "sc_get_timing": (C.c_int, [_P, _D, _I64]),
    "sc_kernel_name": (C.c_char_p, [C.c_int]),
    "sc_set_slab": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "sc_set_slab_axis": (C.c_int, [_P, C.c_int32]),
    "sc_set_band_flag": (C.c_int, [_P, C.c_int]),
    "sc_upload_state_ids": (C.c_int, [_P, _D, _D, _I64, C.c_int64]),
    "sc_append_particles_ids": (C.c_int, [_P, _D, _D, _I64, C.c_int64]),
    "sc_halo_pack": (C.c_int, [_P, _P, _P, C.c_int64]),
    "sc_halo_sizes": (C.c_int, [_P, C.c_int64, _I64, _I64, _I64, _I64]),
    "sc_halo_unpack": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "sc_column_histogram": (C.c_int, [_P, C.c_int64, C.c_int32, _I64]),
    "sc_comm_available": (C.c_int, [C.c_char_p]),
    "sc_comm_unique_id": (C.c_int, [C.c_char_p, _P]),
    "sc_comm_init": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int32]),
    "sc_comm_destroy": (C.c_int, [_P]),
    "sc_halo_exchange": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P, C.c_int64, _P, C.c_int64, C.c_int32]),
    "sc_owned_count": (C.c_int, [_P, _I64]),
    "sc_set_halo_overlap": (C.c_int, [_P, C.c_int]),
    "sc_side_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "sc_halo_overlap_begin": (C.c_int, [_P, _P]),
    "sc_halo_overlap_end": (C.c_int, [_P]),
    "sc_enable_force_monitor": (C.c_int, [_P, C.c_int]),
    "sc_get_force_monitor": (C.c_int, [_P, _D, _I64]),
    "sc_checkpoint_begin": (C.c_int, [_P]),
    "sc_checkpoint_finish": (C.c_int, [_P, _D, _D, _I64, C.c_int64, _I64, _I64, _I64, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_int32)]),
    "sc_restore_counters": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "sc_rng_set_state": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int32]),
    "sc_rng_get_state": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
    "sc_emit_particles": (C.c_int, [_P, C.POINTER(Source), C.c_int32, C.c_double, C.c_int64]),
}

_lib = None


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError(-2, f"{LIB_PATH} is missing; build it with `python -m sand_crate_amd.build` "
                                  "(there is no CPU fallback)")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError(rc, load().sc_last_error().decode("utf-8", "replace"))


def dptr(a: np.ndarray | None):
    # (ctypes.cast of the raw address: a quarter of `a.ctypes.data_as`'s time, and this runs several times per tick;
    # the caller keeps `a` alive)
    return None if a is None else C.cast(a.__array_interface__["data"][0], _D)


def i64ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_I64)


def i32ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_I32)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)
