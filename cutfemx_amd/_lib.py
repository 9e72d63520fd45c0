"""ctypes binding of libcutfemx_amd.so (the C ABI in include/cutfemx_amd.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is
present when a compute entry point is called, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

import numpy as np

_HERE = Path(__file__).resolve().parent
# CFX_LIB: another build of the same engine (timing-ablation / variant builds under build/, tools/*_variants.sh)
LIB_PATH = Path(os.environ["CFX_LIB"]).resolve() if os.environ.get("CFX_LIB") else _HERE / "libcutfemx_amd.so"

OK, ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_OUT_OF_RANGE, ERR_HIP, ERR_STEP_VOID = 0, -1, -2, -3, -4, -5


class StepVoid(RuntimeError):
    """Raised inside a cutfemx_amd.step whose capacities (taken from the previous step) did not fit: the results of
    the step are void; `cutfemx_amd.run_step` ends it and repeats the body."""

INSIDE, INTERSECTED, OUTSIDE = -1, 0, 1
CELL, EXTERIOR_FACET, INTERIOR_FACET = 0, 1, 2
K_MASS, K_STIFFNESS, K_NITSCHE, K_GHOST_GRADJUMP, K_ELASTICITY = 1, 2, 3, 4, 5
K_EXTENSION_L2 = 8
K_JUMP = 9
K_SIP = 10
K_DIV_TEST, K_DIV_TRIAL = 20, 21
L_SOURCE, L_NITSCHE_RHS = 101, 102
F_ONE, F_SINPROD, F_POISSON_RHS, F_COEFFICIENT = 0, 1, 2, 3


class CutOptions(C.Structure):
    _fields_ = [("cut_approximation_order", C.c_int32), ("max_refinement_iterations", C.c_int32),
                ("edge_max_depth", C.c_int32), ("reserved", C.c_int32)]


class RulesView(C.Structure):
    _fields_ = [("tdim", C.c_int32), ("gdim", C.c_int32), ("nq", C.c_int64), ("nr", C.c_int64),
                ("points", C.c_void_p), ("weights", C.c_void_p), ("offsets", C.c_void_p),
                ("parent_map", C.c_void_p), ("host_width", C.c_int32), ("reserved", C.c_int32),
                ("host_rows", C.c_void_p), ("host_verts", C.c_void_p)]


class Integral(C.Structure):
    _fields_ = [("type", C.c_int32), ("kernel", C.c_int32), ("qdegree", C.c_int32),
                ("point_stride", C.c_int32), ("entities", C.c_void_p), ("n_entities", C.c_int64),
                ("rules", C.c_void_p), ("point_data", C.c_void_p), ("params", C.c_double * 8),
                ("coefficient", C.c_void_p)]


class AggregationView(C.Structure):
    _fields_ = [("ncells", C.c_int64), ("root_cell", C.c_void_p), ("aggregate_id", C.c_void_p),
                ("propagation_depth", C.c_void_p), ("cut_volume_fraction", C.c_void_p),
                ("active_cells", C.c_void_p), ("cut_cells", C.c_void_p), ("interior_cells", C.c_void_p),
                ("well_posed_cells", C.c_void_p), ("ill_posed_cells", C.c_void_p), ("rootless_cells", C.c_void_p),
                ("n_active", C.c_int64), ("n_cut", C.c_int64), ("n_interior", C.c_int64),
                ("n_well_posed", C.c_int64), ("n_ill_posed", C.c_int64), ("n_rootless", C.c_int64),
                ("pairs", C.c_void_p), ("n_pairs", C.c_int64)]


class DistExchange(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("send_offset", C.c_int64), ("send_count", C.c_int64),
                ("send_index", C.c_void_p), ("recv_offset", C.c_int64), ("recv_count", C.c_int64),
                ("recv_index", C.c_void_p)]


class DistRowExchange(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("send_row_lo", C.c_int64), ("send_row_hi", C.c_int64),
                ("recv_row_lo", C.c_int64), ("recv_row_hi", C.c_int64)]


HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class PatternView(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("nnz", C.c_int64), ("indptr", C.c_void_p),
                ("indices", C.c_void_p), ("ncols", C.c_int64)]


# every symbol declared in include/cutfemx_amd.h
SYMBOLS = [
    "cfx_init", "cfx_last_error", "cfx_set_stream", "cfx_synchronize", "cfx_step_begin", "cfx_step_end", "cfx_step_resolve", "cfx_step_abort",
    "cfx_step_set_margin", "cfx_step_forget", "cfx_sync_count", "cfx_list_count", "cfx_integrand_register", "cfx_integrand_compile", "cfx_integrand_register_facet", "cfx_integrand_register_variant", "cfx_integrand_compile_bs", "cfx_overlap_begin", "cfx_overlap_side", "cfx_overlap_end", "cfx_copy",
    "cfx_device_alloc", "cfx_device_free", "cfx_device_memset", "cfx_device_cache_release", "cfx_device_memory_stats", "cfx_profile_enable", "cfx_profile_reset",
    "cfx_profile_count", "cfx_profile_get", "cfx_event_create", "cfx_event_record",
    "cfx_event_elapsed_ms", "cfx_event_destroy", "cfx_mesh_create", "cfx_mesh_create_box", "cfx_mesh_create_slab",
    "cfx_mesh_info", "cfx_mesh_destroy", "cfx_cut_options_default", "cfx_cut_create",
    "cfx_cut_restrict", "cfx_cut_create_facets", "cfx_exterior_facets", "cfx_full_facet_rules",
    "cfx_facet_rules_to_cells", "cfx_cut_update", "cfx_cut_info", "cfx_cut_domain", "cfx_locate_entities",
    "cfx_runtime_quadrature", "cfx_runtime_quadratures", "cfx_full_cell_rules", "cfx_rules_create", "cfx_rules_view_get",
    "cfx_rules_physical_points", "cfx_rules_destroy", "cfx_evaluate_normals",
    "cfx_evaluate_values", "cfx_ghost_penalty_facets", "cfx_interior_facets_for_cells", "cfx_cell_aggregation_create", "cfx_cell_aggregation_view_get",
    "cfx_cell_aggregation_destroy", "cfx_cut_destroy", "cfx_space_create",
    "cfx_space_static_bytes", "cfx_space_destroy", "cfx_form_create", "cfx_form_create2", "cfx_form_destroy", "cfx_form_prepare", "cfx_create_sparsity",
    "cfx_pattern_view_get", "cfx_pattern_reuse_stats", "cfx_pattern_destroy", "cfx_assemble_matrix", "cfx_assemble_matrix_zeroed", "cfx_assemble_vector",
    "cfx_apply_lifting", "cfx_set_bc", "cfx_zero_rows", "cfx_csr_block_merge", "cfx_csr_permute", "cfx_tabulate_entity", "cfx_active_domain", "cfx_active_view", "cfx_deactivate_outside",
    "cfx_active_destroy",
    "cfx_dist_unique_id", "cfx_dist_comm_create", "cfx_dist_comm_create_host", "cfx_dist_comm_create_device", "cfx_dist_comm_info", "cfx_dist_comm_destroy",
    "cfx_dist_scatter_forward", "cfx_dist_scatter_reverse_add", "cfx_dist_scatter_reverse_matrix", "cfx_dist_indicator_or",
    "cfx_dist_indicator_forward",
    "cfx_mesh_create_f32", "cfx_cut_create_f32", "cfx_cut_update_f32", "cfx_rules_create_f32", "cfx_rules_view_get_f32",
    "cfx_rules_physical_points_f32", "cfx_evaluate_normals_f32", "cfx_evaluate_values_f32", "cfx_widen_f32",
    "cfx_assemble_matrix_f32", "cfx_assemble_matrix_zeroed_f32", "cfx_assemble_vector_f32", "cfx_apply_lifting_f32",
    "cfx_set_bc_f32", "cfx_zero_rows_f32", "cfx_deactivate_outside_f32",
    "cfx_assemble_matrix_c128", "cfx_assemble_vector_c128", "cfx_apply_lifting_c128", "cfx_set_bc_c128",
    "cfx_deactivate_outside_c128",
    "cfx_assemble_matrix_c64", "cfx_assemble_vector_c64", "cfx_apply_lifting_c64", "cfx_set_bc_c64", "cfx_deactivate_outside_c64",
]

_lib = None
_initialised = False


def load():
    """dlopen the engine (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). cutfemx_amd has no CPU fallback.")
        # A PyTorch-ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.  Loaded after
        # ours it becomes a second runtime in the process and torch then finds "No HIP GPUs";
        # loaded before, the dynamic loader hands the same copy to both.  So when torch is
        # installed it is imported first (dist.py and bench.py need it anyway).
        import importlib.util
        import sys
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
        _lib = C.CDLL(str(LIB_PATH))
        _lib.cfx_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            getattr(_lib, name)  # AttributeError if the ABI and the header diverge
    return _lib


def lib():
    """Engine with an initialised HIP device (raises without one)."""
    global _initialised
    l = load()
    if not _initialised:
        check(l.cfx_init(_default_device()))
        _initialised = True
    return l


def _default_device() -> int:
    import os
    return int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("CFX_DEVICE") is None \
        else int(os.environ["CFX_DEVICE"])


def check(code: int):
    if code == OK:
        return
    msg = (load().cfx_last_error() or b"").decode()
    if code == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if code == ERR_OUT_OF_RANGE:
        raise IndexError(msg)
    if code == ERR_STEP_VOID:
        raise StepVoid(msg)
    raise RuntimeError(msg)


class DeviceBuffer:
    """Raw HBM allocation owned by Python (cfx_device_alloc / cfx_device_free)."""

    def __init__(self, count: int, dtype, shape=None):
        self.dtype = np.dtype(dtype)
        self.size = int(count)
        self.shape = shape if shape is not None else (self.size,)
        self.ndim = len(self.shape)
        p = C.c_void_p()
        check(lib().cfx_device_alloc(C.byref(p), C.c_size_t(max(self.size, 1) * self.dtype.itemsize)))
        self.ptr = p.value

    def numpy(self) -> np.ndarray:
        return download(self.ptr, self.size, self.dtype).reshape(self.shape)

    def fill_from(self, host: np.ndarray):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        check(lib().cfx_copy(C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes)))

    def __del__(self):
        try:
            if self.ptr:
                load().cfx_device_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def is_device(a) -> bool:
    return isinstance(a, DeviceBuffer) or (is_torch(a) and a.is_cuda)


def as_ptr(a, dtype, keep: list):
    """Pointer to a contiguous array of `dtype` (numpy host array or torch tensor)."""
    if a is None:
        return None
    if isinstance(a, DeviceBuffer):
        if a.dtype != np.dtype(dtype):
            raise TypeError(f"device buffer has dtype {a.dtype}, expected {np.dtype(dtype)}")
        keep.append(a)
        return C.c_void_p(a.ptr)
    if is_torch(a):
        import torch
        want = {np.float64: torch.float64, np.float32: torch.float32, np.int32: torch.int32, np.int8: torch.int8}[dtype]
        if a.dtype != want or not a.is_contiguous():
            a = a.to(want).contiguous()
        keep.append(a)
        return C.c_void_p(a.data_ptr())
    arr = np.ascontiguousarray(a, dtype=dtype)
    keep.append(arr)
    return arr.ctypes.data_as(C.c_void_p)


def scalar_dtype(a):
    """np.float32 for a float32 numpy array / torch tensor / DeviceBuffer, else np.float64: which instantiation
    of the boundary (the reference's T of MatrixCSR<T> / Vector<T> / Function<T>) a container selects."""
    if a is None:
        return np.float64
    if getattr(a, "dtype", None) == np.dtype(np.complex128):
        return np.complex128                       # host numpy vectors of the complex128 instantiation (cfx_*_c128)
    if getattr(a, "dtype", None) == np.dtype(np.complex64):
        return np.complex64                        # ... and of the complex64 one (cfx_*_c64)
    if isinstance(a, DeviceBuffer):
        return np.float32 if a.dtype == np.dtype(np.float32) else np.float64
    if is_torch(a):
        import torch
        return np.float32 if a.dtype == torch.float32 else np.float64
    return np.float32 if getattr(a, "dtype", None) == np.dtype(np.float32) else np.float64


def release_cache():
    """Give the engine's cached HBM blocks back to the driver (they are reused between steps otherwise)."""
    check(lib().cfx_device_cache_release())


def memory_stats(reset_peak: bool = False) -> dict:
    """HBM held by the engine's block cache (bytes): handed out, cached, high-water mark of their sum."""
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(lib().cfx_device_memory_stats(C.byref(a), C.byref(b), C.byref(c), 1 if reset_peak else 0))
    return dict(in_use=a.value, cached=b.value, peak=c.value)


_step_open = False   # a cutfemx_amd.step is open in this process: counts of library lists may still be in HBM


def resolve_counts():
    """Inside a sync-free step: fetch the counts published so far (cfx_step_resolve), so that sizes read from the
    library right after are exact -- what a copy of a list to the host needs.  No-op outside a step."""
    if _step_open:
        check(lib().cfx_step_resolve())


def sync_count() -> int:
    """Host round trips (size / error read-backs) the engine has made so far."""
    n = C.c_int64()
    check(lib().cfx_sync_count(C.byref(n)))
    return n.value


def download(ptr, n: int, dtype) -> np.ndarray:
    out = np.empty(int(n), dtype=dtype)
    if n > 0:
        check(lib().cfx_copy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes)))
    return out
