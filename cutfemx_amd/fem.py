"""form / create_matrix / assemble_matrix / assemble_vector / active_domain /
deactivate_outside: the Python surface of python/cutfemx/fem.py backed by the
HIP engine.

The reference compiles UFL forms to JIT kernels (runintgen/FFCx); a GPU engine
cannot call a CPU function pointer per entity, so a form here is a list of
`Integral` descriptors -- an integrand id + parameters + the same integration
domain data the reference puts in `ufl.Measure(subdomain_data=...)`:

    dx(subdomain_data=[inside_cells, rules]) -> Integral(..., cells=inside_cells, rules=rules)
    dx(subdomain_data=interface_rules)       -> Integral(..., rules=interface_rules, point_data=normals)
    dS(subdomain_data=ghost_facets)          -> Integral(..., facets=ghost_facets)
    dS(subdomain_data=[facets, facet_rules]) -> Integral(..., facets=facets, rules=facet_rules)   (rules hosted by interior facets)
    ds(subdomain_data=[facets, facet_rules]) -> Integral(..., rules=full_facet_rules(...).to_cells()), Integral(..., rules=facet_rules.to_cells())
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

_C128, _C64 = np.dtype(np.complex128), np.dtype(np.complex64)


def _is_complex(dt) -> bool:
    return dt is not None and np.dtype(dt) in (_C128, _C64)


def _cfn(name: str, dt):
    """the complex entry point of `name` for the scalar type dt: cfx_<name>_c128 or cfx_<name>_c64"""
    return getattr(_lib.lib(), f"cfx_{name}_{'c64' if np.dtype(dt) == _C64 else 'c128'}")

from .cut import FacetRows, RuntimeQuadratureRules
from .mesh import FunctionSpace

MASS, STIFFNESS, NITSCHE, GHOST_GRADJUMP, ELASTICITY = (_lib.K_MASS, _lib.K_STIFFNESS, _lib.K_NITSCHE,
                                                        _lib.K_GHOST_GRADJUMP, _lib.K_ELASTICITY)
EXTENSION_L2 = _lib.K_EXTENSION_L2
JUMP = _lib.K_JUMP
SIP = _lib.K_SIP
SOURCE, NITSCHE_RHS = _lib.L_SOURCE, _lib.L_NITSCHE_RHS
DIV_TEST, DIV_TRIAL = _lib.K_DIV_TEST, _lib.K_DIV_TRIAL   # rectangular blocks: scale div(v) p / scale q div(u)
F_ONE, F_SINPROD, F_POISSON_RHS, F_COEFFICIENT = _lib.F_ONE, _lib.F_SINPROD, _lib.F_POISSON_RHS, _lib.F_COEFFICIENT


@dataclass
class Integral:
    """One integral of a form (the per-integral tuple of python/cutfemx/fem.py:346-351)."""
    kernel: int
    cells: object = None                      # standard (uncut) cells: ids, numpy/torch/(ptr, n)
    rules: RuntimeQuadratureRules | None = None  # runtime quadrature on the cut cells
    facets: FacetRows | np.ndarray | None = None  # interior facets (c0,lf0,c1,lf1)
    point_data: object = None                 # per-point coefficients aligned with `rules`
    params: tuple = ()
    qdegree: int = 2                          # quadrature degree on the standard entities
    coefficient: object = None                # Function / dof values of the form's space (field id F_COEFFICIENT)
    scale: complex = 1.0                      # constant that multiplies the integrand (fem.Constant kappa in `kappa * ... * dx`,
                                              # test_complex_assembly.py:55-91): a complex value makes the form complex128
    _keep: list = field(default_factory=list, repr=False)

    def _coefficient_values(self):
        return None if self.coefficient is None else getattr(self.coefficient, "values", self.coefficient)

    def _complex_coefficient(self) -> bool:
        v = self._coefficient_values()
        return v is not None and np.iscomplexobj(v)

    def _cstruct(self, part: str = "re") -> _lib.Integral:
        keep = []
        if part == "re":
            self._keep = keep          # what the form of the real parts aliases lives as long as this descriptor ...
        else:
            self._keep_im = keep       # ... and so does what the form of the imaginary parts aliases
        itype = _lib.CELL
        ent_ptr, n_ent = None, 0
        if self.facets is None and self.rules is not None and self.rules.host_width == 4 and self.kernel in (
                _lib.K_GHOST_GRADJUMP, _lib.K_JUMP, _lib.K_SIP):
            itype = _lib.INTERIOR_FACET       # dS over runtime rules only
        if self.facets is not None:
            itype = _lib.INTERIOR_FACET
            if isinstance(self.facets, FacetRows):
                ent_ptr, n_ent = C.c_void_p(self.facets.ptr), self.facets.size
                keep.append(self.facets)
            elif _lib.is_torch(self.facets):  # (nf, 4) int32 tensor, host or device
                ent_ptr, n_ent = _lib.as_ptr(self.facets, np.int32, keep), int(self.facets.numel() // 4)
            else:
                rows = np.ascontiguousarray(self.facets, dtype=np.int32).reshape(-1, 4)
                ent_ptr, n_ent = _lib.as_ptr(rows, np.int32, keep), rows.shape[0]
        elif self.cells is not None:
            if isinstance(self.cells, tuple):  # (device pointer, count)
                ent_ptr, n_ent = C.c_void_p(self.cells[0]), int(self.cells[1])
                keep.append(self.cells)       # a DeviceEntities tuple pins the CutData that owns the list
            else:
                ent_ptr = _lib.as_ptr(self.cells, np.int32, keep)
                n_ent = int(keep[-1].numel() if _lib.is_torch(keep[-1]) else keep[-1].size)
        pd, stride = None, 0
        if self.point_data is not None:
            pdata = self.point_data
            stride = 1 if pdata.ndim == 1 else int(pdata.shape[1])
            pd = _f64_ptr(pdata, keep)
        params = (C.c_double * 8)(*([float(v) for v in self.params] + [0.0] * (8 - len(self.params))))
        coeff = None
        if self.coefficient is not None:
            values = self._coefficient_values()   # a Function or its dof array
            if np.iscomplexobj(values):           # complex Function: its real / imaginary parts feed two real forms
                values = np.ascontiguousarray(np.real(values) if part == "re" else np.imag(values), dtype=np.float64)
            coeff = _f64_ptr(values, keep)
        return _lib.Integral(itype, self.kernel, int(self.qdegree), stride, ent_ptr, n_ent,
                             self.rules._h if self.rules is not None else None, pd, params, coeff)


class _Widened:
    """fp64 HBM copy of a float32 device array (cfx_widen_f32), released with the integral that holds it."""

    def __init__(self, ptr, n):
        p = C.c_void_p()
        _lib.check(_lib.lib().cfx_widen_f32(C.c_void_p(ptr), C.c_int64(n), C.byref(p)))
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().cfx_device_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def _f64_ptr(a, keep: list):
    """Per-point data / coefficient values for a cfx_integral: float32 device arrays are widened in HBM."""
    if _lib.scalar_dtype(a) == np.float32 and _lib.is_device(a):
        n = a.size if isinstance(a, _lib.DeviceBuffer) else int(a.numel())
        w = _Widened(a.ptr if isinstance(a, _lib.DeviceBuffer) else a.contiguous().data_ptr(), n)
        keep.append(w)
        return C.c_void_p(w.ptr)
    return _lib.as_ptr(a, np.float64, keep)


class CutForm:
    """Compiled form handle (python/cutfemx/fem.py CutForm)."""

    def __init__(self, V: FunctionSpace, integrals, rank: int, trial_space: FunctionSpace | None = None, dtype=None):
        self.function_space, self.rank = V, rank
        # Form::function_spaces() = [test, trial] (Form.h:119-178): the same space twice unless `trial_space` is given
        self.trial_space = V if trial_space is None else trial_space
        self.integrals = list(integrals)
        # scalar type T of the form (cutfemx.fem.form(..., dtype=), wrappers/fem.cpp:490-500): complex128 when asked for,
        # or when a constant / coefficient of an integral is complex; real forms with unit constants stay on the
        # float64 entry points
        cplx = any(isinstance(i.scale, complex) and i.scale.imag != 0.0 for i in self.integrals) \
            or any(i._complex_coefficient() for i in self.integrals)
        if dtype is not None and np.dtype(dtype) not in (np.dtype(np.float64), np.dtype(np.float32), _C128, _C64):
            raise TypeError("forms are float64 (float32 containers), complex128 or complex64")
        if cplx and dtype is not None and not _is_complex(dtype):
            raise TypeError("a complex constant or coefficient needs dtype=complex128 (or complex64)")
        # (complex64: the containers the form assembles into are interleaved float32; the form itself is the same)
        self.dtype = np.dtype(dtype) if _is_complex(dtype) else (_C128 if cplx else np.dtype(np.float64))
        self._scaled = any(complex(i.scale) != 1.0 for i in self.integrals)
        if self._scaled and not _is_complex(self.dtype):
            raise TypeError("Integral.scale belongs to complex forms (fold a real constant into `params`)")
        self._h_im = None

        def create(part):
            # (the arrays an Integral hands over live until its next _cstruct call: each form is created -- and its host
            # arrays uploaded -- before the next one is described)
            arr = (_lib.Integral * len(self.integrals))(*[i._cstruct(part) for i in self.integrals])
            h = C.c_void_p()
            if self.trial_space is not V:
                if rank != 2:
                    raise ValueError("a trial space goes with a bilinear form")
                _lib.check(_lib.lib().cfx_form_create2(V._h, self.trial_space._h, len(self.integrals), arr, C.byref(h)))
            else:
                _lib.check(_lib.lib().cfx_form_create(V._h, rank, len(self.integrals), arr, C.byref(h)))
            return h
        self._h = None
        self._h = create("re")
        if any(i._complex_coefficient() for i in self.integrals):
            # the imaginary parts of the complex coefficients: a second real form, assembled with the constants i s_k
            # for the integrals that carry one and 0 for the others
            self._h_im = create("im")

    def _parts(self):
        """(form handle, interleaved complex constants) of the real forms that make up this complex128 form."""
        n = len(self.integrals)
        s = np.array([complex(i.scale) for i in self.integrals])
        out = [(self._h, (C.c_double * (2 * n))(*np.column_stack([s.real, s.imag]).ravel()))]
        if self._h_im is not None:
            t = np.array([1j * complex(i.scale) if i._complex_coefficient() else 0.0 for i in self.integrals])
            out.append((self._h_im, (C.c_double * (2 * n))(*np.column_stack([t.real, t.imag]).ravel())))
        return out

    def prepare(self):
        """Build the form's derived tables now (cfx_form_prepare) instead of inside the first assembly call."""
        _lib.check(_lib.lib().cfx_form_prepare(self._h))
        return self

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_form_destroy(self._h)
                self._h = None
            if getattr(self, "_h_im", None):
                _lib.load().cfx_form_destroy(self._h_im)
                self._h_im = None
        except Exception:
            pass


class overlap:
    """Two independent pieces of a step on two HIP streams (cfx_overlap_begin / _side / _end):

        L.prepare(); a.prepare()                 # shared tables are built before the lanes part
        with fem.overlap() as lanes:
            lanes.side(lambda: fem.assemble_vector(L, b))          # queued on the second stream
            A = fem.create_matrix(a); fem.assemble_matrix(a, A=A)  # meanwhile on the main stream
        # joined here

    The reference assembles a and L one after the other (python/demo/demo_poisson.py:40-60); they share inputs only."""

    def __enter__(self):
        _lib.check(_lib.lib().cfx_overlap_begin())
        return self

    def side(self, fn):
        _lib.check(_lib.lib().cfx_overlap_side(1))
        try:
            return fn()
        finally:
            _lib.check(_lib.lib().cfx_overlap_side(0))

    def __exit__(self, *exc):
        _lib.check(_lib.lib().cfx_overlap_end())
        return False


_user_integrand_rank: dict[int, int] = {}


def register_integrand(name: str, source: str, rank: int = 2, facet: bool = False, variant=None) -> int:
    """Register the HIP C++ source of an integrand; returns the id to put into `Integral.kernel`.

    The reference generates a tabulate_tensor kernel per form at run time (runintgen / FFCx,
    python/cutfemx/_runintgen_adapter.py:181-217) and calls it per entity on the CPU; here the SOURCE of a device function

        __device__ void name(double* A, const double* w, const double* c, const double* coordinate_dofs,
                             int nq, const double* points, const double* weights, const double* point_data)

    (local tensor, packed coefficient, constants = Integral.params, vertex coordinates [(tdim+1)][3], the entity's
    rule: reference points, physical-measure weights, per-point data) is compiled for gfx950 with hipRTC -- see
    include/cutfemx_amd.h (cfx_integrand_register) for the contract and the helpers in scope (cfx_tabulate, ...).
    Cell integrals of Lagrange spaces of degree 1 or 2 (scalar or vector-valued: CFX_BS, CFX_NDB), over standard entities
    and / or runtime rules.  `facet=True`: an interior-facet integrand

        __device__ void name(double* A, const double* w, const double* c, const double* coordinate_dofs,
                             const int* entity_local_index, int nq, const double* points0, const double* points1,
                             const double* weights)

    over (c0, lf0, c1, lf1) rows -- macro tensor [[00, 01], [10, 11]], both cells' coordinate_dofs, {lf0, lf1}
    (assemble_matrix_impl.h:528-542).  `variant=(tdim, dofs per cell[, bs])`: the variant the source is validated
    against at registration (default (3, 4, 1))."""
    kid = C.c_int()
    if variant is None and not facet:
        _lib.check(_lib.load().cfx_integrand_register(name.encode(), source.encode(), int(rank), C.byref(kid)))
    elif variant is None:
        _lib.check(_lib.load().cfx_integrand_register_facet(name.encode(), source.encode(), C.byref(kid)))
    else:
        tdim, nd, bs = (list(variant) + [1])[:3]
        _lib.check(_lib.load().cfx_integrand_register_variant(name.encode(), source.encode(), int(rank), int(bool(facet)),
                                                              int(tdim), int(nd), int(bs), C.byref(kid)))
    _user_integrand_rank[kid.value] = int(rank)
    return kid.value


def compile_integrand(kernel_id: int, tdim: int, ndofs_cell: int, bs: int = 1) -> None:
    """Compile the (tdim, dofs per cell, block size) variant of a registered integrand now instead of at its first use."""
    _lib.check(_lib.load().cfx_integrand_compile_bs(int(kernel_id), int(tdim), int(ndofs_cell), int(bs)))


def form(integrals, V: FunctionSpace, rank: int | None = None, trial_space: FunctionSpace | None = None, dtype=None) -> CutForm:
    """Create a form from integral descriptors (stands in for cutfemx.fem.form).  `V` is the test space; a bilinear
    form whose trial space differs (the off-diagonal blocks of a Stokes system, test_assembly_stokes.py:34-95) names it
    with `trial_space`."""
    integrals = list(integrals)
    if rank is None:
        ranks = {_user_integrand_rank[i.kernel] if i.kernel in _user_integrand_rank else (2 if i.kernel < 100 else 1)
                 for i in integrals}
        if len(ranks) != 1:
            raise ValueError("all integrals of a form must have the same rank")
        rank = ranks.pop()
    return CutForm(V, integrals, rank, trial_space, dtype)


class MatrixCSR:
    """dolfinx.la.MatrixCSR stand-in: indptr (int64), indices (int32), data in HBM."""

    def __init__(self, pattern_handle, V: FunctionSpace, values=None, dtype=None):
        self._p = pattern_handle
        self.function_space = V
        # scalar type T of la::MatrixCSR<T> (wrappers/fem.cpp:490-500): float64, or float32 through the *_f32 entry points
        self.dtype = np.dtype(_lib.scalar_dtype(values) if dtype is None else dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64), _C128, _C64):
            raise TypeError("MatrixCSR holds float64, float32, complex128 or complex64 values")
        v = _lib.PatternView()
        _lib.check(_lib.lib().cfx_pattern_view_get(self._p, C.byref(v)))
        self._view = v
        self.nrows, self.ncols = int(v.nrows), int(v.ncols)
        self._nnz_pending = _lib._step_open   # made inside a cutfemx_amd.step: nnz may still be in HBM
        self._nnz = int(v.nnz)
        self._owns_values = values is None
        self._zero_pending = False
        if values is None:
            p = C.c_void_p()
            _lib.check(_lib.lib().cfx_device_alloc(C.byref(p), C.c_size_t(self.dtype.itemsize * max(self.nnz, 1))))
            self._vptr = p.value
            self.set_value(0.0)
        else:  # caller-owned HBM buffer: set_value / cfx_assemble_matrix write 8*nnz bytes through the raw pointer
            if isinstance(values, _lib.DeviceBuffer):
                ok_type, count, ptr = values.dtype == self.dtype, values.size, values.ptr
            elif _lib.is_torch(values):
                import torch
                want = torch.float32 if self.dtype == np.dtype(np.float32) else torch.float64
                ok_type = values.dtype == want and values.is_cuda and values.is_contiguous()
                count, ptr = int(values.numel()), values.data_ptr()
            else:
                raise TypeError("values must be a device torch tensor or a DeviceBuffer")
            if not ok_type:
                raise TypeError(f"values must be a contiguous {self.dtype} buffer in HBM")
            if count < self.nnz:
                _lib.lib().cfx_pattern_destroy(self._p)
                self._p = None
                raise ValueError(f"values holds {count} entries but the sparsity pattern needs nnz = {self.nnz}")
            self._values_keep = values
            self._vptr = ptr

    @property
    def nnz(self) -> int:
        """Stored entries.  While the cutfemx_amd.step that made the matrix is open this is the capacity of the
        index / value arrays (the exact count is still in HBM); afterwards it is exact."""
        if self._nnz_pending:
            v = _lib.PatternView()
            _lib.check(_lib.lib().cfx_pattern_view_get(self._p, C.byref(v)))
            self._nnz = int(v.nnz)
            self._nnz_pending = _lib._step_open
        return self._nnz

    @property
    def reuse_stats(self) -> tuple[int, int]:
        """(rows of the pattern that needed a hash set, those of them copied from the space's previous pattern)."""
        h, r = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cfx_pattern_reuse_stats(self._p, C.byref(h), C.byref(r)))
        return h.value, r.value

    @property
    def values_ptr(self):
        """HBM address of the value array; a pending set_value(0) is carried out first."""
        if self._zero_pending:
            self._zero_pending = False
            _lib.check(_lib.lib().cfx_device_memset(C.c_void_p(self._vptr), 0, C.c_size_t(self.dtype.itemsize * self.nnz)))
        return self._vptr

    @values_ptr.setter
    def values_ptr(self, p):
        self._vptr = p

    def set_value(self, v: float):
        """la::MatrixCSR::set_value.  Zeroing is deferred to the next use of the values: assemble_matrix then
        clears and assembles in one library call (cfx_assemble_matrix_zeroed)."""
        if complex(v) == 0.0:
            self._zero_pending = True
            return
        self._zero_pending = False
        z = np.full(self.nnz, v if _is_complex(self.dtype) else float(v), dtype=self.dtype)
        _lib.check(_lib.lib().cfx_copy(C.c_void_p(self._vptr), z.ctypes.data_as(C.c_void_p),
                                       C.c_size_t(z.nbytes)))

    def row_block(self, lo: int, hi: int):
        """(indptr - indptr[lo], indices, data) of rows lo..hi-1, downloading only those rows."""
        ip = _lib.download(self._view.indptr + 8 * lo, hi - lo + 1, np.int64)
        e0, e1 = int(ip[0]), int(ip[-1])
        ix = _lib.download(self._view.indices + 4 * e0, e1 - e0, np.int32)
        va = _lib.download(self.values_ptr + self.dtype.itemsize * e0, e1 - e0, self.dtype)
        return ip - e0, ix, va

    def torch_views(self, device):
        """Zero-copy torch tensors (indptr int64, indices int32, values f64) over the HBM arrays."""
        from .dist import as_torch
        return (as_torch(self._view.indptr, self.nrows + 1, "int64", device),
                as_torch(self._view.indices, self.nnz, "int32", device),
                as_torch(self.values_ptr, self.nnz, self.dtype.name, device))

    @property
    def indptr(self):
        return _lib.download(self._view.indptr, self.nrows + 1, np.int64)

    @property
    def indices(self):
        _lib.resolve_counts()
        return _lib.download(self._view.indices, self.nnz, np.int32)

    @property
    def data(self):
        _lib.resolve_counts()
        return _lib.download(self.values_ptr, self.nnz, self.dtype)

    def scatter_reverse(self):
        """Serial: nothing to reduce (multi-GPU reduction lives in cutfemx_amd.dist)."""

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.data, self.indices, self.indptr), shape=(self.nrows, self.ncols))

    def to_dense(self):
        return self.to_scipy().toarray()

    def __del__(self):
        try:
            l = _lib.load()
            if getattr(self, "_owns_values", False) and getattr(self, "_vptr", None):
                l.cfx_device_free(C.c_void_p(self._vptr))
                self._vptr = None
            if self._p:
                l.cfx_pattern_destroy(self._p)
                self._p = None
        except Exception:
            pass


def create_matrix(a: CutForm, values=None, dtype=None) -> MatrixCSR:
    """Sparsity of a bilinear form incl. the all-rows diagonal
    (python/cutfemx/fem.py:810-848 -> assembler.h:567-592).  dtype (or a float32 `values` buffer) selects the
    scalar type of the matrix: float64 (default) or float32."""
    p = C.c_void_p()
    _lib.check(_lib.lib().cfx_create_sparsity(a._h, C.byref(p)))
    if dtype is None and values is None and _is_complex(a.dtype):
        dtype = a.dtype
    return MatrixCSR(p, a.function_space, values, dtype)


def assemble_matrix(a: CutForm, bcs=None, A: MatrixCSR | None = None) -> MatrixCSR:
    """Assemble a bilinear form into CSR (python/cutfemx/fem.py:886-942).
    `bcs` is an int8 dof marker applied to rows and columns, or (bc0, bc1)."""
    if A is None:
        A = create_matrix(a)
    keep: list = []
    bc0 = bc1 = None
    if bcs is not None:
        b0, b1 = bcs if isinstance(bcs, tuple) else (bcs, bcs)
        bc0, bc1 = _lib.as_ptr(b0, np.int8, keep), _lib.as_ptr(b1, np.int8, keep)
    l = _lib.lib()
    if _is_complex(A.dtype):
        zero = 1 if A._zero_pending else 0
        A._zero_pending = False
        for h, scales in a._parts():
            _lib.check(_cfn("assemble_matrix", A.dtype)(h, A._p, bc0, bc1, scales, zero, C.c_void_p(A._vptr)))
            zero = 0
        return A
    if _is_complex(a.dtype):
        raise TypeError("a complex form assembles into a complex matrix (create_matrix(a) makes one)")
    f32 = A.dtype == np.dtype(np.float32)
    if A._zero_pending:     # A.set_value(0) + assemble_matrix(A, a, bcs) as one call
        A._zero_pending = False
        fn = l.cfx_assemble_matrix_zeroed_f32 if f32 else l.cfx_assemble_matrix_zeroed
    else:
        fn = l.cfx_assemble_matrix_f32 if f32 else l.cfx_assemble_matrix
    _lib.check(fn(a._h, A._p, bc0, bc1, C.c_void_p(A._vptr)))
    return A


def assemble_vector(L: CutForm, b=None, dtype=np.float64):
    """Assemble a linear form (python/cutfemx/fem.py:851-883).  Returns a numpy
    vector, or accumulates into `b` (numpy array or device torch tensor; float64 or float32)."""
    V = L.function_space
    if b is None:
        b = np.zeros(V.ndofs * V.bs, dtype=L.dtype if _is_complex(L.dtype) else dtype)
    if _is_complex(_lib.scalar_dtype(b)):
        for h, scales in L._parts():
            _lib.check(_cfn("assemble_vector", _lib.scalar_dtype(b))(h, scales, _vec_ptr(b)))
        return b
    if _is_complex(L.dtype):
        raise TypeError("a complex form assembles into a complex vector")
    fn = _lib.lib().cfx_assemble_vector_f32 if _lib.scalar_dtype(b) == np.float32 else _lib.lib().cfx_assemble_vector
    _lib.check(fn(L._h, _vec_ptr(b)))
    return b


def _vec_ptr(b):
    return C.c_void_p(b.data_ptr()) if _lib.is_torch(b) else b.ctypes.data_as(C.c_void_p)


def apply_lifting(b, a: CutForm, bc_markers, bc_values, x0=None, alpha: float = 1.0):
    """b <- b - alpha A (g - x0) over the Dirichlet columns, entity by entity
    (python/cutfemx/fem.py:604-632 -> lift_bc_impl, assemble_vector_impl.h:383-436).
    `bc_markers` (int8) / `bc_values` / `x0` hold one entry per dof; one bilinear form,
    i.e. one block of the reference's list-of-forms signature."""
    keep: list = []
    if _is_complex(a.dtype) and not _is_complex(_lib.scalar_dtype(b)):
        # (as assemble_matrix / assemble_vector: the real entry points would drop the constants and the imaginary part)
        raise TypeError("a complex form lifts into a complex vector")
    if _is_complex(_lib.scalar_dtype(b)):
        cdt = np.dtype(_lib.scalar_dtype(b))
        g = np.ascontiguousarray(bc_values, dtype=cdt)
        x = None if x0 is None else np.ascontiguousarray(x0, dtype=cdt)
        al = complex(alpha)
        for h, scales in a._parts():
            _lib.check(_cfn("apply_lifting", cdt)(
                h, _lib.as_ptr(bc_markers, np.int8, keep), g.ctypes.data_as(C.c_void_p),
                None if x is None else x.ctypes.data_as(C.c_void_p), C.c_double(al.real), C.c_double(al.imag), scales, _vec_ptr(b)))
        return b
    if _lib.scalar_dtype(b) == np.float32:
        _lib.check(_lib.lib().cfx_apply_lifting_f32(
            a._h, _lib.as_ptr(bc_markers, np.int8, keep), _lib.as_ptr(bc_values, np.float32, keep),
            _lib.as_ptr(x0, np.float32, keep), C.c_float(alpha), _vec_ptr(b)))
        return b
    _lib.check(_lib.lib().cfx_apply_lifting(
        a._h, _lib.as_ptr(bc_markers, np.int8, keep), _lib.as_ptr(bc_values, np.float64, keep),
        _lib.as_ptr(x0, np.float64, keep), C.c_double(alpha), _vec_ptr(b)))
    return b


def set_bc(b, bc_markers, bc_values, x0=None, alpha: float = 1.0):
    """b[dofs] = alpha (g - x0) on the marked dofs (dolfinx.fem.set_bc / DirichletBC.set)."""
    keep: list = []
    n = b.numel() if _lib.is_torch(b) else b.size
    if _is_complex(_lib.scalar_dtype(b)):
        cdt = np.dtype(_lib.scalar_dtype(b))
        g = np.ascontiguousarray(bc_values, dtype=cdt)
        x = None if x0 is None else np.ascontiguousarray(x0, dtype=cdt)
        al = complex(alpha)
        _lib.check(_cfn("set_bc", cdt)(C.c_int64(n), _lib.as_ptr(bc_markers, np.int8, keep), g.ctypes.data_as(C.c_void_p),
                                             None if x is None else x.ctypes.data_as(C.c_void_p), C.c_double(al.real),
                                             C.c_double(al.imag), _vec_ptr(b)))
        return b
    if _lib.scalar_dtype(b) == np.float32:
        _lib.check(_lib.lib().cfx_set_bc_f32(
            C.c_int64(n), _lib.as_ptr(bc_markers, np.int8, keep), _lib.as_ptr(bc_values, np.float32, keep),
            _lib.as_ptr(x0, np.float32, keep), C.c_float(alpha), _vec_ptr(b)))
        return b
    _lib.check(_lib.lib().cfx_set_bc(
        C.c_int64(n), _lib.as_ptr(bc_markers, np.int8, keep), _lib.as_ptr(bc_values, np.float64, keep),
        _lib.as_ptr(x0, np.float64, keep), C.c_double(alpha), _vec_ptr(b)))
    return b


def assemble_scalar(M: CutForm) -> float:
    """Functional int f over the form's cells and runtime rules (python/cutfemx/fem.py:522-528).
    `M` is a linear form of SOURCE integrals: the Lagrange basis is a partition of unity, so the
    functional is the sum of the assembled vector (python/tests/test_cut_api.py:796-811 checks the
    reference the same way)."""
    if M.rank != 1 or any(i.kernel != SOURCE for i in M.integrals):
        raise ValueError("assemble_scalar takes a linear form made of SOURCE integrals (f dx)")
    b = assemble_vector(M)
    return complex(b.sum()) if _is_complex(M.dtype) else float(b.sum())


def zero_rows(A: MatrixCSR, *, tol: float = 0.0) -> np.ndarray:
    """Rows whose assembled entries are all <= tol in magnitude (python/cutfemx/fem.py:777-782)."""
    p, n = C.c_void_p(), C.c_int64()
    if A.dtype == np.dtype(np.float32):
        _lib.check(_lib.lib().cfx_zero_rows_f32(A._p, C.c_void_p(A.values_ptr), C.c_float(tol), C.byref(p), C.byref(n)))
    else:
        _lib.check(_lib.lib().cfx_zero_rows(A._p, C.c_void_p(A.values_ptr), C.c_double(tol), C.byref(p), C.byref(n)))
    out = _lib.download(p.value, n.value, np.int32)
    _lib.check(_lib.lib().cfx_device_free(p))
    return out


def zero_block_rows(A_blocks, *, tol: float = 0.0) -> list[np.ndarray]:
    """Zero rows of each block row of a MatrixCSR block system (python/cutfemx/fem.py:784-800,
    cpp/cutfemx/fem/deactivate.h:279-320): row r of block row i is listed when it is zero in EVERY block A[i][j]
    (None blocks count as zero).  Same checks and messages as the reference."""
    rows = [list(r) for r in A_blocks]
    if not rows:
        raise RuntimeError("Zero-row scan requires at least one block row")
    nb = len(rows)
    out = []
    for i, row in enumerate(rows):
        if len(row) != nb:
            raise RuntimeError("Zero-row scan requires a square block matrix")
        if row[i] is None:
            raise RuntimeError("Zero-row scan requires every diagonal matrix block")
        nrows = row[i].nrows
        if any(A is not None and A.nrows != nrows for A in row):
            raise RuntimeError("Zero-row scan found incompatible row maps in a block row")
        zero = None
        for A in row:
            if A is None:
                continue
            z = zero_rows(A, tol=tol)
            zero = z if zero is None else np.intersect1d(zero, z, assume_unique=True)
        out.append(np.asarray(zero, dtype=np.int32))
    return out


class MergedCSR:
    """The monolithic CSR matrix of a block system (merge_blocks): indptr (int64), indices (int32), float64 data in HBM."""

    def __init__(self, indptr, indices, values, nnz, nrows, ncols, row_offsets, col_offsets):
        self._indptr, self._indices, self._values = indptr, indices, values
        self.nnz, self.nrows, self.ncols = int(nnz), int(nrows), int(ncols)
        self.row_offsets, self.col_offsets = row_offsets, col_offsets   # first row / column of every block
        self.dtype = np.dtype(np.float64)

    @property
    def indptr(self):
        return _lib.download(self._indptr, self.nrows + 1, np.int64)

    @property
    def indices(self):
        return _lib.download(self._indices, self.nnz, np.int32)

    @property
    def data(self):
        return _lib.download(self._values, self.nnz, np.float64)

    def torch_views(self, device):
        """Zero-copy torch tensors (indptr int64, indices int32, values f64) over the HBM arrays."""
        from .dist import as_torch
        return (as_torch(self._indptr, self.nrows + 1, "int64", device), as_torch(self._indices, self.nnz, "int32", device),
                as_torch(self._values, self.nnz, "float64", device))

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.data, self.indices, self.indptr), shape=(self.nrows, self.ncols))

    def to_dense(self):
        return self.to_scipy().toarray()

    def __del__(self):
        try:
            l = _lib.load()
            for name in ("_indptr", "_indices", "_values"):
                if getattr(self, name, None):
                    l.cfx_device_free(C.c_void_p(getattr(self, name)))
                    setattr(self, name, None)
        except Exception:
            pass


def merge_blocks(A_blocks) -> MergedCSR:
    """One CSR matrix from a block system of float64 MatrixCSR blocks (None: an empty block): the monolithic matrix the
    reference assembles on a mixed element (python/tests/test_assembly_stokes.py:34-95), with block-ordered dofs -- all
    rows of block row 0, then of block row 1 ...; columns likewise.  Every block row / column needs one block that fixes
    its size; sizes must agree along rows and columns.  Runs on the GPU (cfx_csr_block_merge); the blocks are not changed."""
    rows = [list(r) for r in A_blocks]
    if not rows or not rows[0]:
        raise RuntimeError("merge_blocks requires at least one block")
    nbr, nbc = len(rows), len(rows[0])
    if any(len(r) != nbc for r in rows):
        raise RuntimeError("merge_blocks requires the same number of blocks in every block row")
    nrows, ncols = [None] * nbr, [None] * nbc
    for i, r in enumerate(rows):
        for j, A in enumerate(r):
            if A is None:
                continue
            if A.dtype != np.dtype(np.float64):
                raise TypeError("merge_blocks takes float64 blocks")
            if nrows[i] not in (None, A.nrows) or ncols[j] not in (None, A.ncols):
                raise RuntimeError("merge_blocks found incompatible block sizes")
            nrows[i], ncols[j] = A.nrows, A.ncols
    if any(n is None for n in nrows) or any(n is None for n in ncols):
        raise RuntimeError("merge_blocks requires a block in every block row and every block column")
    _lib.resolve_counts()
    n = nbr * nbc
    ip, ix, va = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    for i, r in enumerate(rows):
        for j, A in enumerate(r):
            if A is not None:
                k = i * nbc + j
                ip[k], ix[k], va[k] = A._view.indptr, A._view.indices, A.values_ptr
    nr, nc = (C.c_int64 * nbr)(*nrows), (C.c_int64 * nbc)(*ncols)
    o_ip, o_ix, o_va, o_nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    _lib.check(_lib.lib().cfx_csr_block_merge(nbr, nbc, ip, ix, va, nr, nc, C.byref(o_ip), C.byref(o_ix), C.byref(o_va),
                                              C.byref(o_nnz)))
    return MergedCSR(o_ip.value, o_ix.value, o_va.value, o_nnz.value, sum(nrows), sum(ncols),
                     np.concatenate([[0], np.cumsum(nrows)[:-1]]).astype(np.int64),
                     np.concatenate([[0], np.cumsum(ncols)[:-1]]).astype(np.int64))


def permute_csr(A, row_perm, col_perm=None) -> MergedCSR:
    """The matrix `A` (a MergedCSR or MatrixCSR of float64 values) with row r moved to row_perm[r] and column c to
    col_perm[c] (default: the same map), columns of every row ascending again -- on the GPU (cfx_csr_permute)."""
    col_perm = row_perm if col_perm is None else col_perm
    rp = np.ascontiguousarray(row_perm, dtype=np.int32)
    cp = np.ascontiguousarray(col_perm, dtype=np.int32)
    if rp.size != A.nrows or cp.size != A.ncols:
        raise ValueError("permute_csr: one target per row and per column")
    if isinstance(A, MergedCSR):
        ip, ix, va, nnz = A._indptr, A._indices, A._values, A.nnz
    else:
        _lib.resolve_counts()
        ip, ix, va, nnz = A._view.indptr, A._view.indices, A.values_ptr, A.nnz
    o_ip, o_ix, o_va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(_lib.lib().cfx_csr_permute(C.c_int64(A.nrows), C.c_int64(A.ncols), C.c_void_p(ip), C.c_void_p(ix), C.c_void_p(va),
                                          rp.ctypes.data_as(C.c_void_p), cp.ctypes.data_as(C.c_void_p), C.byref(o_ip),
                                          C.byref(o_ix), C.byref(o_va)))
    return MergedCSR(o_ip.value, o_ix.value, o_va.value, nnz, A.nrows, A.ncols, np.zeros(1, np.int64), np.zeros(1, np.int64))


class MixedSpace:
    """A space on a mixed element -- `mixed_element([P2 vector, P1])` of python/tests/test_assembly_stokes.py:34-95 -- as ONE
    object: the caller's mixed dofmap (DOLFINx's: the dofs of all sub-elements numbered together, cell by cell) and the
    sub-elements' (degree, block size).

        W = fem.MixedSpace(mesh, mixed_dofmap, ndofs, [(2, gdim), (1, 1)])
        a = {(0, 0): [Integral(...)], (0, 1): [...], (1, 0): [...], (1, 1): [...]}   # integrals by (test, trial) sub-element
        A = W.assemble_matrix(a)          # ONE CSR matrix in the numbering of `mixed_dofmap`

    Row `mixed_dofmap[c]` lists the cell's dofs sub-element by sub-element; sub-element s takes ndofs_cell(degree_s) x
    bs_s entries, (node i, component k) at i * bs_s + k (the layout of DOLFINx's flattened mixed dofmaps).  The engine
    assembles per (test, trial) pair on block spaces it derives from the columns of the mixed dofmap -- sub.space(s): nodes
    numbered in the order of their first component's mixed dof --, merges the blocks (merge_blocks: block-ordered dofs)
    and moves rows and columns to the mixed numbering (permute_csr with block_to_mixed)."""

    def __init__(self, mesh, mixed_dofmap, ndofs: int, subs):
        from .mesh import FunctionSpace
        M = np.ascontiguousarray(mixed_dofmap, dtype=np.int32)
        if M.ndim != 2 or M.shape[0] != mesh.num_cells:
            raise ValueError("MixedSpace: mixed_dofmap holds one row per cell")
        tdim = mesh.tdim
        self.mesh, self.ndofs, self.subs = mesh, int(ndofs), [(int(d), int(b)) for d, b in subs]
        self.spaces, self.block_to_mixed, self.offsets = [], [], []
        o = 0
        for degree, bs in self.subs:
            if degree not in (1, 2):
                raise ValueError("MixedSpace: sub-elements of degree 1 or 2")
            nd = tdim + 1 if degree == 1 else (tdim + 1) * (tdim + 2) // 2
            if o + nd * bs > M.shape[1]:
                raise ValueError("MixedSpace: the mixed dofmap has fewer entries per cell than its sub-elements need")
            cols = o + bs * np.arange(nd)
            comp0 = M[:, cols]
            nodes = np.unique(comp0)                       # ascending: node k of the block space <-> nodes[k]
            sub_map = np.searchsorted(nodes, comp0).astype(np.int32)
            perm = np.full(nodes.size * bs, -1, dtype=np.int64)
            for k in range(bs):
                perm[sub_map.ravel().astype(np.int64) * bs + k] = M[:, cols + k].ravel()
            if degree == 1 and nodes.size == mesh.num_nodes and np.array_equal(sub_map, mesh.conn):
                V = FunctionSpace(mesh, 1, bs=bs)           # the geometry dofmap itself: the stencil paths apply
            else:
                V = FunctionSpace(mesh, degree, dofmap=sub_map, ndofs=int(nodes.size), bs=bs)
            self.spaces.append(V)
            self.block_to_mixed.append(perm)
            self.offsets.append(o)
            o += nd * bs
        if o != M.shape[1]:
            raise ValueError("MixedSpace: the sub-elements do not account for every entry of a mixed dofmap row")
        P = np.concatenate(self.block_to_mixed)
        if P.size != self.ndofs or P.min() < 0 or np.unique(P).size != P.size or P.max() != self.ndofs - 1:
            raise ValueError("MixedSpace: the sub-elements' dofs do not number 0 .. ndofs - 1 exactly once")
        self.permutation = P.astype(np.int32)              # block-ordered dof -> mixed dof
        self.block_offsets = np.concatenate([[0], np.cumsum([p.size for p in self.block_to_mixed])]).astype(np.int64)

    def sub(self, s: int):
        """The block space of sub-element s (W.sub(s).collapse() of the reference)."""
        return self.spaces[s]

    def form(self, integrals, test: int, trial: int) -> CutForm:
        """The bilinear form of the integrals between sub-elements `test` and `trial`."""
        V0, V1 = self.spaces[test], self.spaces[trial]
        return form(list(integrals), V0, rank=2, trial_space=None if trial == test else V1)

    def assemble_matrix(self, blocks: dict, deactivate: dict | None = None) -> MergedCSR:
        """ONE matrix in the mixed numbering from integrals given by (test, trial) sub-element pair.  `deactivate`
        (optional): {s: ActiveDomain or True} -- the inactive rows of diagonal block (s, s) get a unit diagonal
        (deactivate_outside_blocks) before the merge; True takes the block's own active domain."""
        ns = len(self.spaces)
        forms = {k: (v if isinstance(v, CutForm) else self.form(v, *k)) for k, v in blocks.items()}
        A = [[assemble_matrix(forms[(i, j)]) if (i, j) in forms else None for j in range(ns)] for i in range(ns)]
        if deactivate:
            doms = []
            for s in range(ns):
                d = deactivate.get(s)
                doms.append(active_domain(forms[(s, s)]) if d is True else d)
            if all(d is not None for d in doms):
                deactivate_outside_blocks(A, doms)
            else:
                for s, d in enumerate(doms):
                    if d is not None:
                        deactivate_outside(A[s][s], None, d)
        # (a sub-element without any block still needs its rows: an empty pattern of the right size is not built here)
        merged = merge_blocks(A)
        return permute_csr(merged, self.permutation)

    def vector_to_mixed(self, b_blocks) -> np.ndarray:
        """Block vectors (one per sub-element, block numbering) -> one vector in the mixed numbering."""
        out = np.zeros(self.ndofs, dtype=np.float64)
        for perm, b in zip(self.block_to_mixed, b_blocks):
            out[perm] = np.asarray(b.cpu() if hasattr(b, "cpu") else b, dtype=np.float64)
        return out


def deactivate_outside_blocks(A_blocks, active_domains, b_blocks=None, *, diagonal: float = 1.0,
                              rhs_value: float = 0.0) -> list:
    """Deactivate block rows from per-row active-domain support (python/cutfemx/fem.py:739-775,
    cpp/cutfemx/fem/deactivate.h:420-457): the inactive rows of block row i come from active_domains[i]; only the
    diagonal block A_blocks[i][i] and the optional right-hand side b_blocks[i] are modified -- off-diagonal blocks are
    left alone on purpose (an inactive coupling entry is a form / domain consistency bug, not something to clean)."""
    rows = [list(r) for r in A_blocks]
    domains = list(active_domains)
    if not rows:
        raise RuntimeError("Block deactivation requires at least one block row")
    if len(rows) != len(domains):
        raise RuntimeError("Block deactivation requires one ActiveDomain per block row")
    nb = len(rows)
    for i, row in enumerate(rows):
        if len(row) != nb:
            raise RuntimeError("Block deactivation requires a square block matrix")
        if domains[i] is None:
            raise RuntimeError("Block deactivation received a null ActiveDomain")
        if row[i] is None:
            raise RuntimeError("Block deactivation requires every diagonal matrix block")
    if b_blocks is not None:
        b_blocks = list(b_blocks)
        if len(b_blocks) != nb:
            raise RuntimeError("Block deactivation requires one RHS vector per block row")
        if any(b is None for b in b_blocks):
            raise RuntimeError("Block deactivation received a null RHS vector")
    for i in range(nb):
        deactivate_outside(rows[i][i], None if b_blocks is None else b_blocks[i], domains[i], diagonal=diagonal,
                           rhs_value=rhs_value)
    return domains


def tabulate_entity(a: CutForm, integral: int, index: int, use_rule: bool) -> np.ndarray:
    """Local tensor of one entity (for local-entry parity checks)."""
    V = a.function_space
    I = a.integrals[integral]
    facet_type = I.facets is not None or (I.rules is not None and I.rules.host_width == 4)
    nloc = V.ndofs_cell * V.bs * (2 if facet_type else 1)
    V1 = getattr(a, "trial_space", V)
    nloc1 = nloc if V1 is V else V1.ndofs_cell * V1.bs * (2 if facet_type else 1)   # [(nd0 bs0) x (nd1 bs1)] row-major for rectangular forms (facets: both cells)
    Ae = np.zeros((nloc, nloc1) if a.rank == 2 else (nloc,))
    _lib.check(_lib.lib().cfx_tabulate_entity(a._h, integral, C.c_int64(index), int(use_rule),
                                              Ae.ctypes.data_as(C.c_void_p)))
    return Ae


class ActiveDomain:
    """cutfemx.fem.ActiveDomain (cpp/cutfemx/fem/deactivate.h:387-400)."""

    def __init__(self, handle, V, form=None):
        # (the lists of the domain are compacted on first request from the plan of `form`, whose facet rows alias the
        # form's own entity arrays: the form -- and with it the lists its integrals pin -- lives as long as this object)
        self._h, self.function_space, self._form = handle, V, form

    def _view(self):
        """(active cells, count, inactive dofs, count): the counts are capacities while the cutfemx_amd.step that made
        the domain is open, exact afterwards."""
        ac, na, idf, ni = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        _lib.check(_lib.lib().cfx_active_view(self._h, C.byref(ac), C.byref(na), C.byref(idf), C.byref(ni)))
        return ac.value, na.value, idf.value, ni.value

    @property
    def active_cells(self):
        _lib.resolve_counts()
        ac, na, _, _ = self._view()
        return _lib.download(ac, na, np.int32)

    @property
    def inactive_dofs(self):
        _lib.resolve_counts()
        _, _, idf, ni = self._view()
        return _lib.download(idf, ni, np.int32)

    @property
    def num_active_dofs(self):
        """Dofs touched by an entity of the form (no list is built for this: the row marks of the form's plan)."""
        V = self.function_space
        ni = C.c_int64()
        _lib.check(_lib.lib().cfx_active_view(self._h, None, None, None, C.byref(ni)))
        return V.ndofs * V.bs - ni.value

    def __del__(self):
        try:
            if self._h:
                _lib.load().cfx_active_destroy(self._h)
                self._h = None
        except Exception:
            pass


def active_domain(a: CutForm) -> ActiveDomain:
    h = C.c_void_p()
    _lib.check(_lib.lib().cfx_active_domain(a._h, C.byref(h)))
    return ActiveDomain(h, a.function_space, a)


def deactivate_outside(A: MatrixCSR | None, b, domain: ActiveDomain, diagonal: float = 1.0,
                       rhs_value: float = 0.0) -> ActiveDomain:
    """diag=1 / rhs=0 on dofs outside the active domain (deactivate.h:402-418)."""
    bp = None
    if b is not None:
        bp = C.c_void_p(b.data_ptr()) if _lib.is_torch(b) else b.ctypes.data_as(C.c_void_p)
    dts = {np.dtype(A.dtype)} if A is not None else set()
    if b is not None:
        dts.add(np.dtype(_lib.scalar_dtype(b)))
    if len(dts) > 1:
        raise TypeError("deactivate_outside: matrix and vector must have one scalar type")
    if dts in ({_C128}, {_C64}):
        d, r = complex(diagonal), complex(rhs_value)
        _lib.check(_cfn("deactivate_outside", next(iter(dts)))(domain._h, A._p if A is not None else None,
                                                          C.c_void_p(A.values_ptr) if A is not None else None, bp,
                                                          C.c_double(d.real), C.c_double(d.imag), C.c_double(r.real), C.c_double(r.imag)))
        return domain
    if dts == {np.dtype(np.float32)}:
        _lib.check(_lib.lib().cfx_deactivate_outside_f32(domain._h, A._p if A is not None else None,
                                                         C.c_void_p(A.values_ptr) if A is not None else None, bp,
                                                         C.c_float(diagonal), C.c_float(rhs_value)))
        return domain
    _lib.check(_lib.lib().cfx_deactivate_outside(domain._h, A._p if A is not None else None,
                                                 C.c_void_p(A.values_ptr) if A is not None else None, bp,
                                                 C.c_double(diagonal), C.c_double(rhs_value)))
    return domain
