"""ctypes binding of libcmfhip.so (C ABI in include/cmfhip.h).

There is deliberately no CPU fallback: if the shared library has not been built
or no MI355X is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcmfhip.so")

CMF_U, CMF_V, CMF_Z = 0, 1, 2
CMF_UPD_U, CMF_UPD_V, CMF_UPD_Z = 1, 2, 4   # update_mask bits (include/cmfhip.h)
CMF_NN_U, CMF_NN_V, CMF_NN_Z = 1, 2, 4      # nn_mask bits
LINKS = {"linear": 0, "logit": 1}
UPD_U, UPD_V, UPD_Z = 1, 2, 4
K_GEMM_NN, K_GEMM_TN, K_GEMM_NT, K_ELEMWISE, K_EIGEN = 0, 1, 2, 3, 4
KERNEL_CLASSES = {"gemm_nn": 0, "gemm_tn": 1, "gemm_nt": 2, "elementwise": 3, "eigen": 4, "gemm_small": 5, "spmm": 6, "rowhess": 7, "gemm_pair": 8}

_ERR = {1: ValueError, 2: RuntimeError, 3: MemoryError, 4: RuntimeError, 5: NotImplementedError}

_i64, _i32, _dbl, _vp = C.c_int64, C.c_int, C.c_double, C.c_void_p
_pd = C.POINTER(C.c_double)
_pf = C.POINTER(C.c_float)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)

# name -> (argtypes) ; every function returns int except cmf_last_error
PROTOTYPES = {
    "cmf_device_count": [C.POINTER(C.c_int)],
    "cmf_ctx_create": [C.POINTER(_vp), _i32, _vp],
    "cmf_ctx_destroy": [_vp],
    "cmf_sync": [_vp],
    "cmf_set_option": [_vp, C.c_char_p, _i64],
    "cmf_set_problem": [_vp, _i64, _i64, _i64, _i32],
    "cmf_set_data_f64": [_vp, _i32, _pd, _i64, _i64],
    "cmf_set_data_f32": [_vp, _i32, _pf, _i64, _i64],
    "cmf_set_data_csr": [_vp, _i32, _pi64, _pi32, _pd, _i64],
    "cmf_fill_data_synthetic": [_vp, _i32, C.c_uint64, _i64, _i64],
    "cmf_fill_factor_synthetic": [_vp, _i32, C.c_uint64, _i64, _dbl],
    "cmf_fill_data_synthetic_kind": [_vp, _i32, C.c_uint64, _i64, _i64, _i32, _dbl],
    "cmf_get_data_f32": [_vp, _i32, _pf, _i64, _i64],
    "cmf_data_layout": [_vp, _i32, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "cmf_sparse_layout": [_vp, _i32, _pi64],
    "cmf_get_data_block_f32": [_vp, _i32, _i64, _i64, _i64, _i64, _pf],
    "cmf_sample_lists": [_vp, _i32, C.c_uint64, _dbl, _i64, _i64, _pi32],
    "cmf_newton_clamp_stats": [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), _i32],
    "cmf_newton_clamp_routes": [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "cmf_data_matmul_f64": [_vp, _i32, _i32, _pd, _i64, _i32, _pd],
    "cmf_rsvd": [_vp, _i32, _i32, _i32, _i32, _i32, _pd, _pd, _pd, _pd],
    "cmf_data_sum": [_vp, _pd, _pd],
    "cmf_data_block_sums_f64": [_vp, _i32, _i32, _pd],
    "cmf_set_factor_f64": [_vp, _i32, _pd, _i64, _i64],
    "cmf_get_factor_f64": [_vp, _i32, _pd, _i64, _i64],
    "cmf_mu_step": [_vp, _dbl, _dbl, _i32],
    "cmf_mu_step_error": [_vp, _dbl, _dbl, _i32, _pd, _pd],
    "cmf_v_buf_elems": [_vp, _pi64],
    "cmf_mu_v_partials": [_vp, _vp],
    "cmf_mu_v_apply": [_vp, _vp, _dbl, _dbl],
    "cmf_mu_v_partials_rows": [_vp, _vp, _i64, _i64, _i32],
    "cmf_mu_uz_update": [_vp, _dbl, _dbl, _i32],
    "cmf_mu_blocked_layout": [_vp, _i32, _pi64, _pi64],
    "cmf_mu_v_partials_split": [_vp, _vp, _vp],
    "cmf_mu_v_apply_rows": [_vp, _vp, _vp, _i64, _i64, _dbl, _dbl],
    "cmf_mu_gram_v_rows": [_vp, _i64, _i64, _vp],
    "cmf_mu_uz_update_gram": [_vp, _vp, _dbl, _dbl, _i32],
    "cmf_newton_step": [_vp, _dbl, _dbl, _dbl, _i32, _i32, _i32, _i32, _dbl, _dbl,
                        _pi32, _pi32, _pi32, _pi32],
    "cmf_newton_step_device_sampled": [_vp, _dbl, _dbl, _dbl, _i32, _i32, _i32, _i32, _dbl, _dbl, C.c_uint64],
    "cmf_newton_uz_update": [_vp, _dbl, _dbl, _dbl, _i32, _i32, _dbl],
    "cmf_newton_v_partials": [_vp, _dbl, _vp],
    "cmf_newton_v_apply": [_vp, _vp, _dbl, _dbl, _i32, _dbl],
    "cmf_newton_v_gram": [_vp, _dbl, _vp],
    "cmf_newton_v_products": [_vp, _dbl, _dbl, _dbl, _vp, _vp],
    "cmf_newton_v_finish": [_vp, _vp, _dbl, _i32],
    "cmf_residual_sq": [_vp, _i32, _i32, _pd, _pd],
    "cmf_data_sq": [_vp, _pd, _pd],
    "cmf_safe_invert_batch": [_vp, _pd, _pd, _i32, _i32, _dbl],
    "cmf_safe_invert_f64": [_vp, _pd, _pd, _i32, _dbl],
    "cmf_safe_solve_batch": [_vp, _pd, _pd, _pd, _pd, _i32, _i32, _dbl, _i32],
    "cmf_debug_clock": [_vp, _pd, _pd],
    "cmf_kernel_timing": [_vp, _i32],
    "cmf_kernel_time": [_vp, _i32, _pd, _pi64, _pd],
    "cmf_kernel_timing_reset": [_vp],
    "cmf_rowhess_samples": [_vp, _pd, _pd],
    "cmf_marker": [_vp],
    "cmf_marker_times": [_vp, _pd, _i64, _pi64],
    "cmf_get_stream": [_vp, C.POINTER(_vp)],
    "cmf_get_geometry": [_vp, _pi64, _pi64, _pi64, C.POINTER(C.c_int)],
    "cmf_factor_dev_ptr": [_vp, _i32, C.POINTER(_pf)],
    "cmf_scratch_alloc": [_vp, _i64, C.POINTER(_vp)],
    "cmf_scratch_free": [_vp, _vp],
    "cmf_export_factor_rows": [_vp, _i32, _vp],
    "cmf_import_factor_rows": [_vp, _i32, _vp],
    "cmf_comm_unique_id": [C.c_char_p],
    "cmf_comm_init": [_vp, _i32, _i32, C.c_char_p],
    "cmf_comm_destroy": [_vp],
    "cmf_comm_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "cmf_comm_allreduce_f32": [_vp, _vp, _i64],
    "cmf_comm_allreduce_f64": [_vp, _vp, _i64],
    "cmf_comm_allreduce_f32_bg": [_vp, _vp, _i64],
    "cmf_comm_join": [_vp],
    "cmf_comm_exposed_ms": [_vp, _pd, _i32],
    "cmf_comm_allgather_f32": [_vp, _vp, _i64],
    "cmf_comm_reduce_scatter_f32": [_vp, _vp, _i64],
    "cmf_comm_group_start": [_vp],
    "cmf_comm_group_end": [_vp],
    "cmf_comm_launch_points": [_vp, _pi64],
    "cmf_scale_f32": [_vp, _vp, _i64, _dbl],
    "cmf_comm_count": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "cmf_comm_stats_kind": [_vp, _i32, _pi64, _pi64, _pd],
    "cmf_comm_allreduce_host_f64": [_vp, _pd, _i32, _i32],
    "cmf_comm_barrier": [_vp],
    "cmf_comm_timing": [_vp, _i32],
    "cmf_comm_stats": [_vp, _pi64, _pi64, _pd, _i32],
    "cmf_copy_to_host": [_vp, _vp, _vp, _i64],
    "cmf_copy_from_host": [_vp, _vp, _vp, _i64],
}



class RunParams(C.Structure):
    """cmf_run_params of include/cmfhip.h"""
    _fields_ = [("solver", C.c_int), ("l1", C.c_double), ("l2", C.c_double), ("alpha", C.c_double), ("alpha_err", C.c_double),
                ("x_link", C.c_int), ("y_link", C.c_int), ("nn_mask", C.c_int), ("update_mask", C.c_int),
                ("hessian_pertubation", C.c_double), ("sg_ratio", C.c_double), ("seed", C.c_uint64)]


PROTOTYPES["cmf_run"] = [_vp, C.POINTER(RunParams), _i32, _dbl, _i32, C.POINTER(C.c_int), _pd, _pd, _i32, C.POINTER(C.c_int)]

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and
    ask for it by its unversioned name, so if libcmfhip.so pulled in /opt/rocm's copy first, a
    later ``import torch`` would load a second runtime that finds no GPU.  When torch is
    installed (not necessarily imported) its bundled runtime is loaded first; libcmfhip's
    ``libamdhip64.so.7`` dependency then resolves to that same image."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libcmfhip.so (once) and attach prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pycmf_amd: %s is missing -- build it with `python -m pycmf_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.cmf_last_error.restype = C.c_char_p
    lib.cmf_last_error.argtypes = []
    # the library must have been compiled from the sources next to it (it is git-ignored and travels with the working tree)
    if os.environ.get("PYCMF_AMD_SKIP_HASH_CHECK") != "1" and os.path.isdir(os.path.join(_HERE, "csrc")):
        from . import build as _build
        try:
            lib.cmf_source_hash.restype = C.c_char_p
            lib.cmf_source_hash.argtypes = []
            have = lib.cmf_source_hash().decode()
        except AttributeError:
            have = "unstamped"
        want = _build.source_hash()
        if have != want:
            raise RuntimeError("pycmf_amd: %s was built from other sources than the ones in pycmf_amd/csrc and include/ "
                               "(library %s..., sources %s...): rebuild it with `python -m pycmf_amd.build`"
                               % (LIB_PATH, have[:12], want[:12]))
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().cmf_last_error().decode("utf-8", "replace")
        raise _ERR.get(rc, RuntimeError)("libcmfhip: " + msg)


def device_count():
    n = C.c_int(0)
    _touch()
    check(load().cmf_device_count(C.byref(n)))
    return n.value


def _strides(a):
    return a.strides[0] // a.itemsize, a.strides[1] // a.itemsize


class _Scratch:
    """Device scratch handed out by Context.scratch (freed with the context or by release())."""

    def __init__(self, ctx, ptr, nbytes):
        self._ctx, self._ptr, self.nbytes = ctx, ptr, nbytes

    def data_ptr(self):
        return self._ptr

    def release(self):
        if self._ptr and self._ctx._h:
            check(self._ctx._lib.cmf_scratch_free(self._ctx._h, _vp(self._ptr)))
        self._ptr = None


_gpu_touched = False     # set by the first call that initialises the HIP runtime in this process


def gpu_touched():
    """Has this process made a HIP call through pycmf_amd yet?  (multi_gpu forks its ranks off the caller's process only while
    the answer is no: a forked child must not inherit an initialised runtime.)"""
    return _gpu_touched


def _touch():
    global _gpu_touched
    _gpu_touched = True


class DeviceArray:
    """rows x cols float32 (or float64) matrix in context scratch, row-major: the staging / partial buffers of the sharded
    drivers when no other device allocator is around.  Duck-types what pycmf_amd/sharded.py uses of a torch tensor:
    ``data_ptr()``, ``shape``, ``numel()``, ``element_size()`` and row slicing (a view)."""

    def __init__(self, ctx, rows, cols=1, itemsize=4, _ptr=None, _owner=None):
        self.shape = (int(rows), int(cols))
        self.itemsize = itemsize
        if _ptr is None:
            self._owner = ctx.scratch(max(self.numel() * itemsize, 16))
            self._ptr = self._owner.data_ptr()
        else:
            self._owner, self._ptr = _owner, _ptr
        self._ctx = ctx

    def data_ptr(self):
        return self._ptr

    def numel(self):
        return self.shape[0] * self.shape[1]

    def element_size(self):
        return self.itemsize

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("DeviceArray supports contiguous row slices only")
        lo, hi, _ = sl.indices(self.shape[0])
        hi = max(hi, lo)
        return DeviceArray(self._ctx, hi - lo, self.shape[1], self.itemsize,
                           _ptr=self._ptr + lo * self.shape[1] * self.itemsize, _owner=self._owner)

    def release(self):
        if self._owner is not None and hasattr(self._owner, "release"):
            self._owner.release()


class Context:
    """Owns one cmf_ctx (one GPU).  Thin, typed wrapper over the C ABI."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        _touch()
        self._h = _vp()
        check(self._lib.cmf_ctx_create(C.byref(self._h), int(device), _vp(stream or 0)))
        self.shape = None
        self._keep = []

    def close(self):
        if self._h:
            self._lib.cmf_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- problem / data
    def set_problem(self, m, d, p, k):
        check(self._lib.cmf_set_problem(self._h, m, d, p, k))
        self.shape = (int(m), int(d), int(p), int(k))

    def set_data(self, which, A):
        """Upload X (which=0) or Y (which=1): ndarray (any strides) or scipy CSR/CSC."""
        import scipy.sparse as sp
        if sp.issparse(A):
            A = A.tocsr()  # a copy unless A already is CSR
            if not A.has_canonical_format:
                A = A.copy()           # never canonicalise the caller's matrix in place
                A.sum_duplicates()
            indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
            indices = np.ascontiguousarray(A.indices, dtype=np.int32)
            data = np.ascontiguousarray(A.data, dtype=np.float64)
            check(self._lib.cmf_set_data_csr(self._h, which, indptr.ctypes.data_as(_pi64),
                                             indices.ctypes.data_as(_pi32),
                                             data.ctypes.data_as(_pd), A.nnz))
            return
        A = np.asarray(A)
        if A.dtype == np.float32:
            rs, cs = _strides(A)
            check(self._lib.cmf_set_data_f32(self._h, which, A.ctypes.data_as(_pf), rs, cs))
        else:
            A = A if A.dtype == np.float64 else A.astype(np.float64)
            rs, cs = _strides(A)
            check(self._lib.cmf_set_data_f64(self._h, which, A.ctypes.data_as(_pd), rs, cs))

    def get_data(self, which):
        m, d, p, _ = self.shape
        out = np.empty((m, d) if which == 0 else (d, p), dtype=np.float32)
        check(self._lib.cmf_get_data_f32(self._h, which, out.ctypes.data_as(_pf), out.shape[1], 1))
        return out

    def data_matmul(self, which, trans, B):
        """op(X|Y) @ B on the device for a host float64 matrix B; returns a host float64 array."""
        B = np.ascontiguousarray(B, dtype=np.float64)
        m, d, p, _ = self.shape
        ar, ac = ((m, d), (d, p))[which]
        out = np.empty((ac if trans else ar, B.shape[1]))
        check(self._lib.cmf_data_matmul_f64(self._h, which, 1 if trans else 0, B.ctypes.data_as(_pd), B.shape[0], B.shape[1],
                                            out.ctypes.data_as(_pd)))
        return out

    def rsvd(self, which, transpose, k, size, n_iter, omega):
        """Randomized truncated SVD of X / Y on the device copy; returns (U, S, Vt) float64."""
        m, d, p, _ = self.shape
        rows, cols = ((m, d), (d, p))[which]
        omega = np.ascontiguousarray(omega, dtype=np.float64)
        assert omega.shape == ((rows if transpose else cols), size)
        U, S, Vt = np.empty((rows, k)), np.empty(k), np.empty((k, cols))
        check(self._lib.cmf_rsvd(self._h, which, 1 if transpose else 0, k, size, n_iter, omega.ctypes.data_as(_pd),
                                 U.ctypes.data_as(_pd), S.ctypes.data_as(_pd), Vt.ctypes.data_as(_pd)))
        return U, S, Vt

    def data_sum(self):
        sx, sy = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_data_sum(self._h, C.byref(sx), C.byref(sy)))
        return sx.value, sy.value

    def sparse_layout(self, which):
        """(groups, split rows, pieces, accumulator rows per group) of the blocked SpMM image of A and of A^T."""
        out = (C.c_int64 * 8)()
        check(self._lib.cmf_sparse_layout(self._h, which, out))
        v = [int(x) for x in out]
        return tuple(v[:4]), tuple(v[4:])

    def data_layout(self, which):
        """(dense image exists, native CSR pair resident) of X (0) / Y (1)."""
        a, b = C.c_int(0), C.c_int(0)
        check(self._lib.cmf_data_layout(self._h, which, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def get_data_block(self, which, row0, nrows, col0, ncols):
        """float32 block of the dense device image of X (0) / Y (1)."""
        out = np.empty((nrows, ncols), dtype=np.float32)
        check(self._lib.cmf_get_data_block_f32(self._h, which, row0, nrows, col0, ncols, out.ctypes.data_as(_pf)))
        return out

    def sample_lists(self, sweep, seed, ratio, row0, nrows):
        """Index lists of the device sampler for rows [row0, row0 + nrows) of sweep 0 (U) / 1 (Z) / 2 (V, X side) / 3 (V, Y side)."""
        m, d, p = self.shape[:3]
        n = d if sweep <= 1 else (m if sweep == 2 else p)
        per = int(n * ratio)
        out = np.empty((nrows, per), dtype=np.int32)
        check(self._lib.cmf_sample_lists(self._h, sweep, seed, ratio, row0, nrows, out.ctypes.data_as(_pi32)))
        return out

    def fill_data_synthetic(self, which, seed, row0=0, col0=0, kind=0, param=0.0):
        """kind 0: |N(0,1)|; 1: sigmoid(N(0,1)); 2: Bernoulli(param) in {0, 1}"""
        if kind == 0:
            check(self._lib.cmf_fill_data_synthetic(self._h, which, seed, row0, col0))
        else:
            check(self._lib.cmf_fill_data_synthetic_kind(self._h, which, seed, row0, col0, kind, param))

    def fill_factor_synthetic(self, which, seed, row0=0, scale=1.0):
        check(self._lib.cmf_fill_factor_synthetic(self._h, which, seed, row0, scale))

    def set_factor(self, which, F):
        F = np.asarray(F)
        if F.dtype != np.float64:
            F = F.astype(np.float64)
        rs, cs = _strides(F)
        check(self._lib.cmf_set_factor_f64(self._h, which, F.ctypes.data_as(_pd), rs, cs))

    def get_factor_into(self, which, F):
        """Write the device factor back into the caller's float64 array, in place."""
        assert F.dtype == np.float64
        rs, cs = _strides(F)
        check(self._lib.cmf_get_factor_f64(self._h, which, F.ctypes.data_as(_pd), rs, cs))

    def get_factor(self, which):
        m, d, p, k = self.shape
        out = np.empty(((m, d, p)[which], k))
        self.get_factor_into(which, out)
        return out

    # ---- MU
    def mu_step(self, l1, l2, mask=7):
        check(self._lib.cmf_mu_step(self._h, l1, l2, mask))

    def mu_step_error(self, l1, l2, mask=7):
        """One MU iteration and the squared residuals (ex2, ey2) of the factors it leaves, from the step's own products."""
        ex2, ey2 = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_mu_step_error(self._h, l1, l2, mask, C.byref(ex2), C.byref(ey2)))
        return ex2.value, ey2.value

    def v_buf_elems(self):
        n = C.c_int64(0)
        check(self._lib.cmf_v_buf_elems(self._h, C.byref(n)))
        return n.value

    def mu_v_partials(self, dev_ptr):
        check(self._lib.cmf_mu_v_partials(self._h, _vp(dev_ptr)))

    def mu_v_partials_rows(self, dev_ptr, row0, nrows, with_gram):
        check(self._lib.cmf_mu_v_partials_rows(self._h, _vp(dev_ptr), row0, nrows, 1 if with_gram else 0))

    def mu_v_apply(self, dev_ptr, l1, l2):
        check(self._lib.cmf_mu_v_apply(self._h, _vp(dev_ptr), l1, l2))

    def mu_uz_update(self, l1, l2, mask=7):
        check(self._lib.cmf_mu_uz_update(self._h, l1, l2, mask))

    # row-blocked V update (reduce-scatter + epilogue on the rank's block + all-gather): include/cmfhip.h
    def mu_blocked_layout(self, world):
        """(rows per block, floats of the partial buffer); grows the allocation behind V so that the all-gather runs in place."""
        b, n = C.c_int64(0), C.c_int64(0)
        check(self._lib.cmf_mu_blocked_layout(self._h, int(world), C.byref(b), C.byref(n)))
        return b.value, n.value

    def mu_v_partials_split(self, dev_p, dev_g):
        check(self._lib.cmf_mu_v_partials_split(self._h, _vp(dev_p), _vp(dev_g)))

    def mu_v_apply_rows(self, dev_p_rows, dev_g, row0, nrows, l1, l2):
        check(self._lib.cmf_mu_v_apply_rows(self._h, _vp(dev_p_rows), _vp(dev_g), row0, nrows, l1, l2))

    def mu_gram_v_rows(self, row0, nrows, dev_g2):
        check(self._lib.cmf_mu_gram_v_rows(self._h, row0, nrows, _vp(dev_g2)))

    def mu_uz_update_gram(self, dev_g2, l1, l2, mask=7):
        check(self._lib.cmf_mu_uz_update_gram(self._h, _vp(dev_g2), l1, l2, mask))

    def factor_dev_ptr(self, which):
        p = _pf()
        check(self._lib.cmf_factor_dev_ptr(self._h, which, C.byref(p)))
        return C.cast(p, _vp).value

    def run(self, solver, max_iter, tol, l1=0.0, l2=0.0, alpha=0.5, alpha_err=0.5, x_link="linear", y_link="linear", nn_mask=0,
            update_mask=7, pert=0.2, ratio=1.0, seed=0, check_every=10):
        """The whole outer loop in C (cmf_run): returns (n_iter, errors, seconds) -- errors[0] is the error at init, one more
        entry per convergence check, seconds the time since the call at those points."""
        prm = RunParams(0 if solver == "mu" else 1, l1, l2, alpha, alpha_err, LINKS[x_link], LINKS[y_link], nn_mask, update_mask,
                        pert, ratio, int(seed))
        cap = max_iter // max(check_every, 1) + 2
        errs, secs = (C.c_double * cap)(), (C.c_double * cap)()
        n_iter, n_trace = C.c_int(0), C.c_int(0)
        check(self._lib.cmf_run(self._h, C.byref(prm), int(max_iter), float(tol), int(check_every), C.byref(n_iter), errs, secs, cap,
                                C.byref(n_trace)))
        n = min(n_trace.value, cap)
        return n_iter.value, [errs[i] for i in range(n)], [secs[i] for i in range(n)]

    # ---- Newton
    def newton_step(self, alpha, l1, l2, x_link, y_link, nn_mask, upd_mask, pert, ratio,
                    u_idx=None, z_idx=None, vx_idx=None, vy_idx=None):
        def ptr(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.int32)
            self._keep.append(a)
            return a.ctypes.data_as(_pi32)
        self._keep = []
        check(self._lib.cmf_newton_step(self._h, alpha, l1, l2, LINKS[x_link], LINKS[y_link],
                                        nn_mask, upd_mask, pert, ratio,
                                        ptr(u_idx), ptr(z_idx), ptr(vx_idx), ptr(vy_idx)))
        self._keep = []

    def newton_step_device_sampled(self, alpha, l1, l2, x_link, y_link, nn_mask, upd_mask, pert, ratio, seed):
        check(self._lib.cmf_newton_step_device_sampled(self._h, alpha, l1, l2, LINKS[x_link], LINKS[y_link],
                                                       nn_mask, upd_mask, pert, ratio, seed))

    def newton_uz_update(self, alpha, l1, l2, nn_mask, upd_mask, pert):
        check(self._lib.cmf_newton_uz_update(self._h, alpha, l1, l2, nn_mask, upd_mask, pert))

    def newton_v_partials(self, alpha, dev_ptr):
        check(self._lib.cmf_newton_v_partials(self._h, alpha, _vp(dev_ptr)))

    def newton_v_apply(self, dev_ptr, l1, l2, nn_mask, pert):
        check(self._lib.cmf_newton_v_apply(self._h, _vp(dev_ptr), l1, l2, nn_mask, pert))

    def newton_v_gram(self, alpha, dev_gbuf):
        check(self._lib.cmf_newton_v_gram(self._h, alpha, _vp(dev_gbuf)))

    def newton_v_products(self, alpha, l2, pert, dev_gbuf, dev_pbuf):
        check(self._lib.cmf_newton_v_products(self._h, alpha, l2, pert, _vp(dev_gbuf), _vp(dev_pbuf)))

    def newton_v_finish(self, dev_pbuf, l1, nn_mask):
        check(self._lib.cmf_newton_v_finish(self._h, _vp(dev_pbuf), l1, nn_mask))

    # ---- metrics
    def newton_clamp_stats(self, reset=False, full=False):
        """(rows whose float32 Hessian went through the spectral clamp and stayed float32, largest ||H||_F / pert (or condition
        estimate of a plain solve) among the rows left in float32, rows redone in float64) since the last reset; ``full`` adds the
        largest condition estimate over all plain float32 solves."""
        rows, ratio, refined, plain = C.c_int64(0), C.c_double(0), C.c_int64(0), C.c_double(0)
        check(self._lib.cmf_newton_clamp_stats(self._h, C.byref(rows), C.byref(ratio), C.byref(refined), C.byref(plain), 1 if reset else 0))
        return (rows.value, ratio.value, refined.value, plain.value) if full else (rows.value, ratio.value, refined.value)

    def newton_clamp_routes(self):
        """(rows served by the tridiagonal eigen-solve, rows served by the rank-one shortcut) since the context was created."""
        e, r = C.c_int64(0), C.c_int64(0)
        check(self._lib.cmf_newton_clamp_routes(self._h, C.byref(e), C.byref(r)))
        return e.value, r.value

    def residual_sq(self, x_link="linear", y_link="linear"):
        ex, ey = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_residual_sq(self._h, LINKS[x_link], LINKS[y_link], C.byref(ex), C.byref(ey)))
        return ex.value, ey.value

    def data_sq(self):
        x2, y2 = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_data_sq(self._h, C.byref(x2), C.byref(y2)))
        return x2.value, y2.value

    def safe_invert_batch(self, H, pert):
        H = np.ascontiguousarray(H, dtype=np.float64)
        n, k, _ = H.shape
        out = np.empty_like(H)
        check(self._lib.cmf_safe_invert_batch(self._h, H.ctypes.data_as(_pd), out.ctypes.data_as(_pd), n, k, pert))
        return out

    def safe_solve_batch(self, H, g, pert, method=0, eigenvalues=False):
        """g_b * safe_inverse(H_b) for a batch (the step of a per-row sweep), float32 device path; method 1: all through the
        tridiagonal eigen-solve.  eigenvalues=True also returns the eigenvalues the QL iteration found (method 1)."""
        H = np.ascontiguousarray(H, dtype=np.float64)
        g = np.ascontiguousarray(g, dtype=np.float64)
        n, k, _ = H.shape
        out = np.empty((n, k), dtype=np.float64)
        lam = np.empty((n, k), dtype=np.float64) if eigenvalues else None
        check(self._lib.cmf_safe_solve_batch(self._h, H.ctypes.data_as(_pd), g.ctypes.data_as(_pd), out.ctypes.data_as(_pd),
                                             lam.ctypes.data_as(_pd) if eigenvalues else None, n, k, pert, method))
        return (out, lam) if eigenvalues else out

    def safe_invert_f64(self, H, pert):
        """float64 path of the shared Hessian: H is k x k with k = this context's n_components."""
        H = np.ascontiguousarray(H, dtype=np.float64)
        out = np.empty_like(H)
        check(self._lib.cmf_safe_invert_f64(self._h, H.ctypes.data_as(_pd), out.ctypes.data_as(_pd), H.shape[0], pert))
        return out

    def scratch(self, nbytes):
        """Zero-filled device buffer owned by the context; the returned object has ``data_ptr()`` like a torch tensor."""
        p = _vp()
        check(self._lib.cmf_scratch_alloc(self._h, int(nbytes), C.byref(p)))
        return _Scratch(self, p.value, int(nbytes))

    # ---- collectives (RCCL inside the C ABI: csrc/cmf_comm.hip.h)
    def comm_init(self, rank, world, unique_id):
        check(self._lib.cmf_comm_init(self._h, rank, world, bytes(unique_id)))

    def comm_destroy(self):
        check(self._lib.cmf_comm_destroy(self._h))

    def comm_allreduce(self, buf):
        """In-place sum over the ranks of a DeviceArray / scratch / tensor-like (float32, or float64 by element_size)."""
        n = buf.numel()
        if buf.element_size() == 8:
            check(self._lib.cmf_comm_allreduce_f64(self._h, _vp(buf.data_ptr()), n))
        else:
            check(self._lib.cmf_comm_allreduce_f32(self._h, _vp(buf.data_ptr()), n))

    def comm_allreduce_bg(self, buf):
        check(self._lib.cmf_comm_allreduce_f32_bg(self._h, _vp(buf.data_ptr()), buf.numel()))

    def comm_join(self):
        check(self._lib.cmf_comm_join(self._h))

    def comm_exposed_ms(self, reset=False):
        ms = C.c_double(0)
        check(self._lib.cmf_comm_exposed_ms(self._h, C.byref(ms), 1 if reset else 0))
        return ms.value

    def comm_allgather(self, full, elems_per_rank):
        check(self._lib.cmf_comm_allgather_f32(self._h, _vp(full.data_ptr()), elems_per_rank))

    def comm_reduce_scatter(self, full, elems_per_rank):
        check(self._lib.cmf_comm_reduce_scatter_f32(self._h, _vp(full.data_ptr()), elems_per_rank))

    def data_block_sums(self, which, axis):
        """float64 sums of the dense device image of X (0) / Y (1) over blocks of 256 rows (axis 0 -> [rows_pad / 256, cols_pad]) or
        256 columns (axis 1 -> [rows_pad, cols_pad / 256]); padded extents."""
        mp, dp, pp, _ = self.geometry()
        rp, cp = (mp, dp) if which == 0 else (dp, pp)
        out = np.empty((rp // 256, cp) if axis == 0 else (rp, cp // 256), dtype=np.float64)
        check(self._lib.cmf_data_block_sums_f64(self._h, int(which), int(axis), out.ctypes.data_as(_pd)))
        return out

    def comm_group_start(self):
        check(self._lib.cmf_comm_group_start(self._h))

    def comm_group_end(self):
        check(self._lib.cmf_comm_group_end(self._h))

    def comm_launch_points(self):
        n = C.c_int64(0)
        check(self._lib.cmf_comm_launch_points(self._h, C.byref(n)))
        return n.value

    def scale(self, buf, factor):
        """buf *= factor on the context's stream (device array)."""
        check(self._lib.cmf_scale_f32(self._h, _vp(buf.data_ptr()), buf.numel(), float(factor)))

    def comm_count(self):
        """(ranks, this rank) as RCCL reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(0), C.c_int(0)
        check(self._lib.cmf_comm_count(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_stats_kind(self, kind):
        """(calls, payload bytes, ms) of one kind of collective: 0 all-reduce f32, 1 all-reduce f64, 2 all-gather, 3 reduce-scatter, 4 group
        (calls = groups closed, ms = events around the whole group; the members are counted under their own kinds without ms)."""
        calls, nbytes, ms = C.c_int64(0), C.c_int64(0), C.c_double(0)
        check(self._lib.cmf_comm_stats_kind(self._h, int(kind), C.byref(calls), C.byref(nbytes), C.byref(ms)))
        return calls.value, nbytes.value, ms.value

    def comm_allreduce_host(self, values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(self._lib.cmf_comm_allreduce_host_f64(self._h, a.ctypes.data_as(_pd), a.size, 0 if op == "sum" else 1))
        return a

    def comm_barrier(self):
        check(self._lib.cmf_comm_barrier(self._h))

    def comm_timing(self, enable=True):
        check(self._lib.cmf_comm_timing(self._h, 1 if enable else 0))

    def comm_stats(self, reset=False):
        """(calls, payload bytes, ms on the stream) since the last reset; waits for the stream."""
        calls, nbytes, ms = C.c_int64(0), C.c_int64(0), C.c_double(0)
        check(self._lib.cmf_comm_stats(self._h, C.byref(calls), C.byref(nbytes), C.byref(ms), 1 if reset else 0))
        return calls.value, nbytes.value, ms.value

    def copy_to_host(self, buf, dtype=np.float32):
        """numpy copy of a DeviceArray / scratch buffer (waits for the stream)."""
        out = np.empty(buf.numel(), dtype=np.float64 if buf.element_size() == 8 else dtype)
        check(self._lib.cmf_copy_to_host(self._h, _vp(buf.data_ptr()), out.ctypes.data_as(_vp), out.nbytes))
        return out.reshape(buf.shape) if hasattr(buf, "shape") else out

    def copy_from_host(self, buf, a):
        a = np.ascontiguousarray(a)
        check(self._lib.cmf_copy_from_host(self._h, _vp(buf.data_ptr()), a.ctypes.data_as(_vp), a.nbytes))

    def export_factor_rows(self, which, dev_ptr):
        """Device-to-device copy of all valid rows (k_pad floats each) to `dev_ptr`, on the context's stream."""
        check(self._lib.cmf_export_factor_rows(self._h, which, _vp(dev_ptr)))

    def import_factor_rows(self, which, dev_ptr):
        check(self._lib.cmf_import_factor_rows(self._h, which, _vp(dev_ptr)))

    def set_option(self, name, value):
        check(self._lib.cmf_set_option(self._h, name.encode(), int(value)))

    def sync(self):
        check(self._lib.cmf_sync(self._h))

    def debug_clock(self):
        g, u = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_debug_clock(self._h, C.byref(g), C.byref(u)))
        return g.value, u.value

    def kernel_timing(self, enable):
        """False / 0 off, True / 1 every launch, 2 the data-pass classes only (see cmfhip.h)."""
        check(self._lib.cmf_kernel_timing(self._h, int(enable)))

    def kernel_timing_reset(self):
        check(self._lib.cmf_kernel_timing_reset(self._h))

    def kernel_time(self, cls):
        """(accumulated ms, launches, algorithmic flops) of one kernel class."""
        ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
        check(self._lib.cmf_kernel_time(self._h, KERNEL_CLASSES.get(cls, cls), C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    def rowhess_samples(self):
        """(sample rows credited by the algorithm, sample rows gathered by the outer-product kernel) since the last reset."""
        a, b = C.c_double(0), C.c_double(0)
        check(self._lib.cmf_rowhess_samples(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def marker(self):
        check(self._lib.cmf_marker(self._h))

    def marker_times(self, cap=4096):
        """Elapsed ms of every marker since the first one (waits for the stream; clears the markers)."""
        buf = (C.c_double * cap)()
        n = C.c_int64(0)
        check(self._lib.cmf_marker_times(self._h, buf, cap, C.byref(n)))
        return [buf[i] for i in range(min(n.value, cap))]

    def stream_handle(self):
        p = _vp()
        check(self._lib.cmf_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def geometry(self):
        a, b, c_, k = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(self._lib.cmf_get_geometry(self._h, C.byref(a), C.byref(b), C.byref(c_), C.byref(k)))
        return a.value, b.value, c_.value, k.value
