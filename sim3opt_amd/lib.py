"""ctypes binding of the C-ABI (include/sim3opt.h) -- used by tests/ and bench.py.

This is plumbing, not the product: every call lands in libsim3opt.so (HIP, gfx950).
There is no Python/CPU fallback; a missing library raises at import of the symbol table
and a missing GPU makes `Graph.initialize()` raise Sim3OptError(SIM3OPT_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SIM3OPT_LIB: another build of the same library, e.g. the host-sanitizer build of scripts/host_asan_suite.sh)
LIB_PATH = os.environ.get("SIM3OPT_LIB") or os.path.join(_HERE, "libsim3opt.so")
# (a tuning script that A/Bs two builds sets sim3opt_amd.lib.LIB_PATH before the first load(); no
# environment variable redirects the dlopen)

OK, ERR_ARG, ERR_STATE, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_COMM = 0, -1, -2, -3, -4, -5, -6
KERNEL_NONE, KERNEL_HUBER = 0, 1


class Options(C.Structure):
    _fields_ = [
        ("tau", C.c_double),
        ("user_lambda_init", C.c_double),
        ("good_step_lower", C.c_double),
        ("good_step_upper", C.c_double),
        ("max_trials", C.c_int32),
        ("fd_delta", C.c_double),
        ("exp_eps", C.c_double),
        ("small_rot_half", C.c_int32),
        ("fix_small_angle_b", C.c_int32),
        ("dof_mask", C.c_int32),
        ("pcg_max_iters", C.c_int32),
        ("pcg_rel_tol", C.c_double),
        ("pcg_check_every", C.c_int32),
        ("pcg_graph", C.c_int32),
        ("preconditioner", C.c_int32),
        ("chain_segment", C.c_int32),
        ("device", C.c_int32),
        ("verbose", C.c_int32),
        ("time_kernels", C.c_int32),
        ("linear_solver", C.c_int32),
        ("amg_cycle", C.c_int32 * 4),
        ("amg_passes", C.c_int32 * 3),
        ("amg_additive", C.c_int32),
        ("amg_fp32", C.c_int32),
        ("amg_pivot", C.c_int32),
        ("amg_coarsest", C.c_int32),
        ("adaptive_prec", C.c_int32),
        ("row_order", C.c_int32),
        ("halo_exchange", C.c_int32),
        ("span_grid", C.c_int32),
        ("force_collectives", C.c_int32),
        ("amg_shard_rows", C.c_int32),
        ("amg_virtual_ranks", C.c_int32),
        ("pcg_batch", C.c_int32),
        ("amg_omega", C.c_double),
        ("amg_over", C.c_double * 2),
        ("direct_max_pairs", C.c_int64),
        ("debug_full_arrays", C.c_int32),
    ]


class IterStats(C.Structure):
    _fields_ = [
        ("chi2_before", C.c_double),
        ("chi2_after", C.c_double),
        ("lambda_", C.c_double),
        ("rho", C.c_double),
        ("trials", C.c_int32),
        ("pcg_iters", C.c_int32),
        ("pcg_rel_res", C.c_double),
        ("ms_linearize", C.c_double),
        ("ms_solve", C.c_double),
        ("ms_update", C.c_double),
        ("pcg_capped", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class BaOptions(C.Structure):
    """sim3opt_ba_options (include/sim3opt.h), field for field."""
    _fields_ = [
        ("huber_delta", C.c_double),
        ("pixel_noise", C.c_double),
        ("tau", C.c_double),
        ("user_lambda_init", C.c_double),
        ("max_trials", C.c_int32),
        ("pcg_max_iters", C.c_int32),
        ("pcg_rel_tol", C.c_double),
        ("linear_solver", C.c_int32),
        ("device", C.c_int32),
        ("verbose", C.c_int32),
    ]


class KernelTimes(C.Structure):
    _fields_ = [
        ("ms_spmv", C.c_double), ("n_spmv", C.c_int64),
        ("ms_pcg_vec", C.c_double), ("n_pcg_vec", C.c_int64),
        ("ms_linearize", C.c_double), ("n_linearize", C.c_int64),
        ("ms_chi2", C.c_double), ("n_chi2", C.c_int64),
        ("ms_update", C.c_double), ("n_update", C.c_int64),
        ("ms_replicated_levels", C.c_double), ("n_replicated_visits", C.c_int64),
        ("n_batches", C.c_int64), ("n_batched_solves", C.c_int64),
    ]


class CommTimes(C.Structure):
    _fields_ = [
        ("ms_allreduce", C.c_double), ("n_allreduce", C.c_int64), ("bytes_allreduce", C.c_int64),
        ("ms_allgather", C.c_double), ("n_allgather", C.c_int64), ("bytes_allgather", C.c_int64),
        ("ms_exchange", C.c_double), ("n_exchange", C.c_int64), ("bytes_exchange", C.c_int64),
    ]


# every symbol include/sim3opt.h declares, with its signature
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint8)
_vp = C.c_void_p
SYMBOLS = {
    "sim3opt_version": (C.c_int, []),
    "sim3opt_options_default": (None, [C.POINTER(Options)]),
    "sim3opt_create": (_vp, []),
    "sim3opt_destroy": (None, [_vp]),
    "sim3opt_set_options": (C.c_int, [_vp, C.POINTER(Options)]),
    "sim3opt_get_options": (C.c_int, [_vp, C.POINTER(Options)]),
    "sim3opt_last_error": (C.c_char_p, [_vp]),
    "sim3opt_add_vertex": (C.c_int, [_vp, C.c_int32, _dp, C.c_int32]),
    "sim3opt_add_vertices": (C.c_int, [_vp, C.c_int32, _ip, _dp, _up]),
    "sim3opt_add_edge": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp, _dp, C.c_int32, C.c_double]),
    "sim3opt_add_edges": (C.c_int, [_vp, C.c_int32, _ip, _ip, _dp, _dp, C.c_int32, C.c_double]),
    "sim3opt_num_vertices": (C.c_int32, [_vp]),
    "sim3opt_num_edges": (C.c_int32, [_vp]),
    "sim3opt_get_edge": (C.c_int, [_vp, C.c_int32, _ip, _ip, _dp]),
    "sim3opt_initialize": (C.c_int, [_vp]),
    "sim3opt_optimize": (C.c_int, [_vp, C.c_int32]),
    "sim3opt_get_vertex": (C.c_int, [_vp, C.c_int32, _dp]),
    "sim3opt_set_vertex": (C.c_int, [_vp, C.c_int32, _dp]),
    "sim3opt_get_vertices": (C.c_int, [_vp, _dp]),
    "sim3opt_set_vertices": (C.c_int, [_vp, _dp]),
    "sim3opt_chi2": (C.c_int, [_vp, _dp]),
    "sim3opt_num_iterations": (C.c_int32, [_vp]),
    "sim3opt_get_stats": (C.c_int, [_vp, C.c_int32, C.POINTER(IterStats)]),
    "sim3opt_get_comm_times": (C.c_int, [_vp, C.POINTER(CommTimes)]),
    "sim3opt_get_kernel_times": (C.c_int, [_vp, C.POINTER(KernelTimes)]),
    "sim3opt_reset_kernel_times": (C.c_int, [_vp]),
    "sim3opt_edge_errors": (C.c_int, [_vp, _dp]),
    "sim3opt_linearize": (C.c_int, [_vp]),
    "sim3opt_system_dims": (C.c_int, [_vp, _ip, C.POINTER(C.c_int64)]),
    "sim3opt_system_pattern": (C.c_int, [_vp, _ip, C.POINTER(C.c_int64), _ip, _ip]),
    "sim3opt_get_system": (C.c_int, [_vp, _ip, _ip, _dp, _dp]),
    "sim3opt_solve": (C.c_int, [_vp, C.c_double, _dp, _ip, _dp]),
    "sim3opt_bench_spmv": (C.c_int, [_vp, C.c_int32, _dp]),
    "sim3opt_bench_stream": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp]),
    "sim3opt_preconditioner_in_use": (C.c_int, [_vp]),
    "sim3opt_amg_hierarchy": (C.c_int, [_vp, C.c_int32, _ip, _ip, _vp, _ip]),
    "sim3opt_linear_solver_in_use": (C.c_int, [_vp]),
    "sim3opt_direct_plan": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int64), _ip, _ip, _ip, _ip, _ip, _ip,
                                      _ip, _ip, _ip, _ip, _ip, _ip]),
    "sim3opt_comm_allgather_plan": (C.c_int, [C.c_int32, C.c_int32, _ip, C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int64)]),
    "sim3opt_comm_unique_id": (C.c_int, [_up]),
    "sim3opt_comm_init": (C.c_int, [_vp, C.c_int32, C.c_int32, _up]),
    "sim3opt_comm_init_callbacks": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "sim3opt_comm_set_alltoallv": (C.c_int, [_vp, _vp]),
    "sim3opt_amg_in_use": (C.c_int, [_vp, _ip, _ip, _ip]),
    "sim3opt_device_bytes": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "sim3opt_halo_plan": (C.c_int, [_vp, C.c_int32, C.c_int32, _ip, _ip, _ip, _ip, _ip, _ip]),
    "sim3opt_local_rows": (C.c_int, [_vp, _ip, _ip]),
    "sim3opt_partition_plan": (C.c_int, [_vp, C.c_int32, C.c_int32, _ip, _ip, _ip, C.POINTER(C.c_int64)]),
    "sim3opt_partition_rows": (C.c_int, [C.c_int32, _ip, C.c_int32, _ip]),
    "sim3opt_partition_rows_equal": (C.c_int, [C.c_int32, C.c_int32, _ip]),
    "sim3opt_load_kitti_direct": (C.c_int, [_vp, C.c_char_p, C.c_int32]),
    "sim3opt_load_kitti_gt_loops": (C.c_int, [_vp, C.c_char_p]),
    "sim3opt_release_device_cache": (None, []),
    "sim3opt_write_poses": (C.c_int, [_vp, C.c_char_p, _ip]),
    "sim3opt_stepwise_scale_init": (C.c_int, [_vp, _dp]),
    "sim3opt_read_keyframe_bin": (C.c_int, [C.c_char_p, _ip, _dp, _dp, _ip, C.POINTER(C.c_uint32), _dp,
                                            _dp, C.c_int32]),
    "sim3opt_reanchor_points": (C.c_int, [C.c_int32, _dp, _dp, C.c_int32, _dp, C.c_int32, _ip, _ip,
                                          C.c_int32]),
    "sim3opt_write_g2o": (C.c_int, [_vp, C.c_char_p]),
    "sim3opt_write_bal": (C.c_int, [C.c_char_p, C.c_int32, _dp, _dp, _dp, C.c_int32, _dp, C.c_int32, _ip, _ip, _dp]),
    "sim3opt_align_trajectory": (C.c_int, [C.c_int32, _dp, _dp, C.c_int32, _dp, _dp, _dp]),
    "sim3opt_ba_options_default": (None, [C.POINTER(BaOptions)]),
    "sim3opt_ba_create": (_vp, []),
    "sim3opt_ba_destroy": (None, [_vp]),
    "sim3opt_ba_last_error": (C.c_char_p, [_vp]),
    "sim3opt_ba_set_options": (C.c_int, [_vp, C.POINTER(BaOptions)]),
    "sim3opt_ba_set_problem": (C.c_int, [_vp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _ip, _ip, _dp,
                                         C.c_double, C.c_double, C.c_double]),
    "sim3opt_ba_set_fixed_cameras": (C.c_int, [_vp, _up]),
    "sim3opt_ba_read_bal": (C.c_int, [_vp, C.c_char_p, C.c_double, C.c_double, C.c_double]),
    "sim3opt_ba_dims": (C.c_int, [_vp, _ip, _ip, _ip]),
    "sim3opt_ba_chi2": (C.c_int, [_vp, _dp]),
    "sim3opt_ba_optimize": (C.c_int, [_vp, C.c_int32]),
    "sim3opt_ba_get_cameras": (C.c_int, [_vp, _dp]),
    "sim3opt_ba_get_points": (C.c_int, [_vp, _dp]),
    "sim3opt_ba_num_iterations": (C.c_int32, [_vp]),
    "sim3opt_ba_get_stats": (C.c_int, [_vp, C.c_int32, C.POINTER(IterStats)]),
    "sim3opt_ba_write_poses": (C.c_int, [_vp, C.c_char_p]),
}

_lib = None


OPTIONAL_SYMBOLS = {
    "sim3opt_bench_spmv_symmetric": (C.c_int, [_vp, C.c_int32, _dp]),
    "sim3opt_bench_spmv_rowlane": (C.c_int, [_vp, C.c_int32, C.c_int32, _dp]),
}


def hip_runtime_path():
    """Path of the libamdhip64 this process has mapped (the one libsim3opt.so is bound to), or None."""
    try:
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    return ln.split()[-1]
    except OSError:
        pass
    return None


def load():
    """Loads libsim3opt.so and binds every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not built: run `python -m sim3opt_amd.build` (hipcc, gfx950)")
        # One HIP runtime per process: libsim3opt.so names the system's libamdhip64, a PyTorch-ROCm wheel
        # carries its own.  Whichever is mapped first serves both (same soname) -- but if the library were
        # mapped before torch and torch then initialised its own copy, the library's copy would find the
        # device taken ("no usable HIP device").  So a process that will ALSO use torch (bench.py, the
        # tests, build() + smoke() in one interpreter) must map torch's first: it says so by importing
        # torch before this call, or by SIM3OPT_PRELOAD_TORCH=1.  Otherwise the library binds to the ROCm
        # it was compiled against and this module imports nothing.
        if "torch" not in sys.modules and os.environ.get("SIM3OPT_PRELOAD_TORCH", "0") not in ("", "0"):
            try:
                import torch  # noqa: F401
            except Exception:  # no torch in this interpreter: the system runtime is the only one
                pass
        L = C.CDLL(LIB_PATH)
        if os.environ.get("SIM3OPT_VERBOSE_LOAD"):
            print(f"sim3opt: {LIB_PATH} loaded; HIP runtime mapped: {hip_runtime_path()}", file=sys.stderr)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in OPTIONAL_SYMBOLS.items():  # a SIM3OPT_BENCH_HOOKS build only
            fn = getattr(L, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int32, C.c_int32)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.POINTER(C.c_int64), C.c_int32, C.c_int32)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.POINTER(C.c_int64), _dp, C.POINTER(C.c_int64), C.c_int32,
                           C.c_int32)


class Sim3OptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sim3opt error {code}: {msg}")
        self.code = code


def default_options(**kw):
    o = Options()
    load().sim3opt_options_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Graph:
    """Thin object wrapper over sim3opt_graph* (one per optimiser, like g2o::SparseOptimizer)."""

    def __init__(self, **options):
        self._L = load()
        self._g = self._L.sim3opt_create()
        if not self._g:
            raise MemoryError("sim3opt_create")
        if options:
            self.set_options(**options)

    def close(self):
        if self._g:
            self._L.sim3opt_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise Sim3OptError(rc, self._L.sim3opt_last_error(self._g).decode())
        return rc

    # ---- configuration ----
    def set_options(self, **kw):
        o = Options()
        self._chk(self._L.sim3opt_get_options(self._g, C.byref(o)))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            cur = getattr(o, k)
            if hasattr(cur, "__len__"):  # array fields (amg_cycle, amg_passes, amg_over): element by element
                for i, x in enumerate(v):
                    cur[i] = x
            else:
                setattr(o, k, v)
        self._chk(self._L.sim3opt_set_options(self._g, C.byref(o)))

    def options(self):
        o = Options()
        self._chk(self._L.sim3opt_get_options(self._g, C.byref(o)))
        return o

    # ---- graph construction ----
    def add_vertex(self, vid, state, fixed=False):
        s = _f64(state)
        self._chk(self._L.sim3opt_add_vertex(self._g, int(vid), _p(s, _dp), int(bool(fixed))))

    def add_vertices(self, states, fixed=None, ids=None):
        s = _f64(states).reshape(-1, 8)
        f = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8)
        i = None if ids is None else _i32(ids)
        self._chk(self._L.sim3opt_add_vertices(self._g, s.shape[0], _p(i, _ip), _p(s, _dp),
                                               _p(f, _up)))

    def add_edge(self, v0, v1, meas, info=None, kernel=KERNEL_NONE, kernel_delta=0.0):
        m = _f64(meas)
        inf = None if info is None else np.asfortranarray(info, dtype=np.float64).ravel(order="F")
        self._chk(self._L.sim3opt_add_edge(self._g, int(v0), int(v1), _p(m, _dp), _p(inf, _dp),
                                           int(kernel), float(kernel_delta)))

    def add_edges(self, v0, v1, meas, info=None, kernel=KERNEL_NONE, kernel_delta=0.0):
        a, b = _i32(v0), _i32(v1)
        m = _f64(meas).reshape(-1, 8)
        inf = None
        if info is not None:  # (m, 7, 7) [k, r, c] -> column-major blocks
            inf = _f64(np.asarray(info).reshape(-1, 7, 7).transpose(0, 2, 1)).reshape(-1, 49)
        self._chk(self._L.sim3opt_add_edges(self._g, a.shape[0], _p(a, _ip), _p(b, _ip),
                                            _p(m, _dp), _p(inf, _dp), int(kernel),
                                            float(kernel_delta)))

    @property
    def num_vertices(self):
        return self._L.sim3opt_num_vertices(self._g)

    @property
    def num_edges(self):
        return self._L.sim3opt_num_edges(self._g)

    def get_edge(self, k):
        a, b = C.c_int32(), C.c_int32()
        m = np.empty(8)
        self._chk(self._L.sim3opt_get_edge(self._g, int(k), C.byref(a), C.byref(b), _p(m, _dp)))
        return a.value, b.value, m

    # ---- multi-GPU ----
    def comm_init_rccl(self, rank, world, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.shape == (128,)
        self._chk(self._L.sim3opt_comm_init(self._g, int(rank), int(world), _p(uid, _up)))

    def comm_init_callbacks(self, rank, world, allreduce, allgatherv, alltoallv=None):
        """allreduce(np_array, op) and allgatherv(np_array, offsets, rank) operate IN PLACE on
        numpy views of the library's pinned host staging buffer; the optional
        alltoallv(send, send_offsets, recv, recv_offsets, rank) is the neighbour exchange
        (send[send_offsets[p]:send_offsets[p+1]] goes to rank p, recv[...] comes from it)."""
        def _ar(ctx, buf, n, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(n,)), int(op))
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def _ag(ctx, buf, offs, rk, world_):
            try:
                o = np.ctypeslib.as_array(offs, shape=(world_ + 1,))
                allgatherv(np.ctypeslib.as_array(buf, shape=(int(o[-1]),)), o, int(rk))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def _aa(ctx, sbuf, soffs, rbuf, roffs, rk, world_):
            try:
                so = np.ctypeslib.as_array(soffs, shape=(world_ + 1,))
                ro = np.ctypeslib.as_array(roffs, shape=(world_ + 1,))
                sv = np.ctypeslib.as_array(sbuf, shape=(max(int(so[-1]), 1),))
                rv = np.ctypeslib.as_array(rbuf, shape=(max(int(ro[-1]), 1),))
                alltoallv(sv, so, rv, ro, int(rk))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._cb_refs = (ALLREDUCE_FN(_ar), ALLGATHERV_FN(_ag), ALLTOALLV_FN(_aa))  # keep alive
        self._chk(self._L.sim3opt_comm_init_callbacks(
            self._g, int(rank), int(world), C.cast(self._cb_refs[0], C.c_void_p),
            C.cast(self._cb_refs[1], C.c_void_p), None))
        if alltoallv is not None:
            self._chk(self._L.sim3opt_comm_set_alltoallv(self._g, C.cast(self._cb_refs[2], C.c_void_p)))

    def local_rows(self):
        a, b = C.c_int32(), C.c_int32()
        self._chk(self._L.sim3opt_local_rows(self._g, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- optimisation ----
    def initialize(self):
        self._chk(self._L.sim3opt_initialize(self._g))

    def optimize(self, max_iters):
        """Returns iterations executed (g2o convention: 0 failure, -1 nothing to do)."""
        it = self._L.sim3opt_optimize(self._g, int(max_iters))
        if it == 0 and max_iters > 0:
            raise Sim3OptError(0, self._L.sim3opt_last_error(self._g).decode())
        return it

    def chi2(self):
        v = C.c_double()
        self._chk(self._L.sim3opt_chi2(self._g, C.byref(v)))
        return v.value

    def get_vertex(self, vid):
        s = np.empty(8)
        self._chk(self._L.sim3opt_get_vertex(self._g, int(vid), _p(s, _dp)))
        return s

    def set_vertex(self, vid, state):
        s = _f64(state)
        self._chk(self._L.sim3opt_set_vertex(self._g, int(vid), _p(s, _dp)))

    def get_vertices(self):
        s = np.empty((self.num_vertices, 8))
        self._chk(self._L.sim3opt_get_vertices(self._g, _p(s, _dp)))
        return s

    def set_vertices(self, states):
        s = _f64(states).reshape(-1, 8)
        assert s.shape[0] == self.num_vertices
        self._chk(self._L.sim3opt_set_vertices(self._g, _p(s, _dp)))

    def stats(self):
        out = []
        for i in range(self._L.sim3opt_num_iterations(self._g)):
            st = IterStats()
            self._chk(self._L.sim3opt_get_stats(self._g, i, C.byref(st)))
            out.append(st)
        return out

    def kernel_times(self, reset=False):
        kt = KernelTimes()
        self._chk(self._L.sim3opt_get_kernel_times(self._g, C.byref(kt)))
        if reset:
            self._chk(self._L.sim3opt_reset_kernel_times(self._g))
        return kt

    def comm_times(self):
        """Device time, count and payload of the collectives since initialize / the last reset
        (options.time_kernels): dict of the sim3opt_comm_times fields."""
        ct = CommTimes()
        self._chk(self._L.sim3opt_get_comm_times(self._g, C.byref(ct)))
        return {k: getattr(ct, k) for k, _ in CommTimes._fields_}

    # ---- kernel-level access ----
    def edge_errors(self):
        e = np.empty((self.num_edges, 7))
        self._chk(self._L.sim3opt_edge_errors(self._g, _p(e, _dp)))
        return e

    def linearize(self):
        self._chk(self._L.sim3opt_linearize(self._g))

    def system_dims(self):
        nb, nnzb = C.c_int32(), C.c_int64()
        self._chk(self._L.sim3opt_system_dims(self._g, C.byref(nb), C.byref(nnzb)))
        return nb.value, nnzb.value

    def get_system(self):
        """(rowptr, colidx, blocks[nnzb, 7, 7] indexed [k, r, c], b)."""
        nb, nnzb = self.system_dims()
        rowptr = np.empty(nb + 1, dtype=np.int32)
        colidx = np.empty(nnzb, dtype=np.int32)
        vals = np.empty((nnzb, 49))
        b = np.empty(7 * nb)
        self._chk(self._L.sim3opt_get_system(self._g, _p(rowptr, _ip), _p(colidx, _ip),
                                             _p(vals, _dp), _p(b, _dp)))
        return rowptr, colidx, vals.reshape(-1, 7, 7).transpose(0, 2, 1).copy(), b

    def dense_system(self):
        """Dense (H, b) assembled from the block-CSR copy (small graphs, tests only)."""
        rowptr, colidx, blocks, b = self.get_system()
        nb = rowptr.shape[0] - 1
        H = np.zeros((7 * nb, 7 * nb))
        for i in range(nb):
            for k in range(rowptr[i], rowptr[i + 1]):
                j = colidx[k]
                H[7 * i:7 * i + 7, 7 * j:7 * j + 7] += blocks[k]
        return H, b

    def solve(self, lam):
        nb, _ = self.system_dims()
        x = np.empty(7 * nb)
        it = C.c_int32()
        rr = C.c_double()
        self._chk(self._L.sim3opt_solve(self._g, float(lam), _p(x, _dp), C.byref(it),
                                        C.byref(rr)))
        return x, it.value, rr.value

    def preconditioner_in_use(self):
        rc = self._L.sim3opt_preconditioner_in_use(self._g)
        if rc < 0:
            self._chk(rc)
        return rc

    def amg_in_use(self):
        """dict(levels, partitioned_levels, cycle): what the automatic multigrid choices resolved to."""
        nl, ns = C.c_int32(), C.c_int32()
        v = np.zeros(4, dtype=np.int32)
        self._chk(self._L.sim3opt_amg_in_use(self._g, C.byref(nl), C.byref(ns), _p(v, _ip)))
        return dict(levels=nl.value, partitioned_levels=ns.value, cycle=[int(x) for x in v])

    def device_bytes(self):
        """(bytes of the block arrays as allocated on this rank, bytes one rank holding the whole graph allocates)."""
        b = (C.c_int64 * 2)()
        self._chk(self._L.sim3opt_device_bytes(self._g, b))
        return int(b[0]), int(b[1])

    def system_pattern(self):
        """(rowptr, colidx) of the block-CSR system; host only."""
        nb, nnzb = C.c_int32(), C.c_int64()
        self._chk(self._L.sim3opt_system_pattern(self._g, C.byref(nb), C.byref(nnzb), None, None))
        rowptr = np.zeros(nb.value + 1, dtype=np.int32)
        colidx = np.zeros(max(nnzb.value, 1), dtype=np.int32)
        self._chk(self._L.sim3opt_system_pattern(self._g, None, None, _p(rowptr, _ip), _p(colidx, _ip)))
        return rowptr, colidx[:nnzb.value]

    def linear_solver_in_use(self):
        rc = self._L.sim3opt_linear_solver_in_use(self._g)
        if rc < 0:
            self._chk(rc)
        return rc

    def direct_plan(self, max_pairs=0):
        """Plan of the exact sparse block Cholesky as a dict of numpy arrays; host only."""
        dims = np.zeros(8, dtype=np.int64)
        dp = dims.ctypes.data_as(C.POINTER(C.c_int64))
        null = [None] * 12
        self._chk(self._L.sim3opt_direct_plan(self._g, int(max_pairs), dp, *null))
        nb, nL, npairs, height, ngroups, nlev, nsrc, nrounds = (int(x) for x in dims[:8])
        arr = dict(perm=nb, colptr=nb + 1, lrow=nL, srcptr=nL + 1, src=nsrc, pairptr=nL + 1,
                   pa=npairs, pb=npairs, gptr=ngroups + 1, lcolp=nlev + 1, rptr=nlev + 1,
                   cells=18 * nrounds)
        out = {k: np.zeros(max(n, 1), dtype=np.int32) for k, n in arr.items()}
        self._chk(self._L.sim3opt_direct_plan(self._g, int(max_pairs), dp,
                                              *[_p(out[k], _ip) for k in arr]))
        out = {k: out[k][:n] for k, n in arr.items()}
        out.update(nb=nb, nL=nL, npairs=npairs, height=height, ngroups=ngroups, nlevels=nlev,
                   nrounds=nrounds)
        return out

    def amg_hierarchy(self):
        """(rows per level, blocks per level, level-1 row of every level-0 block row); host only."""
        nl = C.c_int32()
        rows = np.zeros(16, dtype=np.int32)
        blocks = np.zeros(16, dtype=np.int64)
        nfree = self.num_vertices
        agg = np.full(nfree, -1, dtype=np.int32)
        self._chk(self._L.sim3opt_amg_hierarchy(self._g, 16, C.byref(nl), _p(rows, _ip),
                                                blocks.ctypes.data_as(C.c_void_p), _p(agg, _ip)))
        return rows[:nl.value].copy(), blocks[:nl.value].copy(), agg

    def partition_plan(self, world, locality=True):
        """(vertex of every block row, row_begin, boundary rows per rank, cut edges); host only."""
        nfree = C.c_int32()
        nblk = C.c_int64()
        self._chk(self._L.sim3opt_system_pattern(self._g, C.byref(nfree), C.byref(nblk), None, None))
        v = np.empty(nfree.value, dtype=np.int32)
        rb = np.empty(world + 1, dtype=np.int32)
        bnd = np.empty(world, dtype=np.int32)
        cut = C.c_int64()
        self._chk(self._L.sim3opt_partition_plan(self._g, int(world), int(bool(locality)), _p(v, _ip), _p(rb, _ip),
                                                 _p(bnd, _ip), C.byref(cut)))
        return v, rb, bnd, cut.value

    def halo_plan(self, world, rank):
        """(send_rows, send_seg, recv_rows, recv_seg) of `rank` on level 0 of the `world`-rank partition; host only."""
        ns, nr = C.c_int32(), C.c_int32()
        self._chk(self._L.sim3opt_halo_plan(self._g, int(world), int(rank), C.byref(ns), C.byref(nr), None, None,
                                            None, None))
        sr, rr = np.zeros(max(ns.value, 1), np.int32), np.zeros(max(nr.value, 1), np.int32)
        ss, rs = np.zeros(world + 1, np.int32), np.zeros(world + 1, np.int32)
        self._chk(self._L.sim3opt_halo_plan(self._g, int(world), int(rank), None, None, _p(sr, _ip), _p(ss, _ip),
                                            _p(rr, _ip), _p(rs, _ip)))
        return sr[:ns.value], ss, rr[:nr.value], rs

    def bench_spmv(self, reps=20):
        ms = C.c_double()
        self._chk(self._L.sim3opt_bench_spmv(self._g, int(reps), C.byref(ms)))
        return ms.value

    def bench_spmv_symmetric(self, reps=20):
        """Prototype of the two-phase upper-triangle SpMV: (ms phase 1, ms phase 2, max rel difference to
        the product SpMV, bytes of its stream)."""
        out = np.zeros(4)
        self._chk(self._L.sim3opt_bench_spmv_symmetric(self._g, int(reps), _p(out, _dp)))
        return tuple(float(v) for v in out)

    def bench_spmv_rowlane(self, reps=20, rows_per_group=2):
        """Prototype of the row-per-lane FP32 passes (a SIM3OPT_BENCH_HOOKS build): 8 numbers, see sim3opt_bench.h."""
        out = np.zeros(8)
        self._chk(self._L.sim3opt_bench_spmv_rowlane(self._g, int(reps), int(rows_per_group), _p(out, _dp)))
        return tuple(float(v) for v in out)

    def bench_stream(self, mode, reps=20):
        ms = C.c_double()
        self._chk(self._L.sim3opt_bench_stream(self._g, int(mode), int(reps), C.byref(ms)))
        return ms.value

    # ---- reference-format I/O ----
    def load_kitti_direct(self, directory, use_one_constraint=True):
        self._chk(self._L.sim3opt_load_kitti_direct(self._g, os.fsencode(directory),
                                                    int(bool(use_one_constraint))))

    def load_kitti_gt_loops(self, directory):
        """Ground-truth poses + line 1 of every loop record: all residuals ~ 0 (a convention pin)."""
        self._chk(self._L.sim3opt_load_kitti_gt_loops(self._g, os.fsencode(directory)))

    def stepwise_scale_init(self):
        """Stage 1 of the stepwise pipeline; returns the sigma_min/sigma_max estimate."""
        r = C.c_double()
        self._chk(self._L.sim3opt_stepwise_scale_init(self._g, C.byref(r)))
        return r.value

    def write_g2o(self, path):
        self._chk(self._L.sim3opt_write_g2o(self._g, os.fsencode(path)))

    def write_poses(self, path, image_ids=None):
        ids = None if image_ids is None else _i32(image_ids)
        self._chk(self._L.sim3opt_write_poses(self._g, os.fsencode(path), _p(ids, _ip)))


def read_keyframe_bin(path):
    """dict(kf_id, Rw2c 3x3, twinc, point_ids, points_w (n,3), obs_uv (n,2))  -- LoadComboKeyFrame."""
    Lb = load()
    kf, n = C.c_int32(), C.c_int32()
    R, t = np.empty(9), np.empty(3)
    rc = Lb.sim3opt_read_keyframe_bin(os.fsencode(path), C.byref(kf), _p(R, _dp), _p(t, _dp),
                                      C.byref(n), None, None, None, 0)
    if rc != OK:
        raise Sim3OptError(rc, "read_keyframe_bin")
    ids = np.empty(max(n.value, 1), dtype=np.uint32)
    pts = np.empty((max(n.value, 1), 3))
    uv = np.empty((max(n.value, 1), 2))
    rc = Lb.sim3opt_read_keyframe_bin(os.fsencode(path), C.byref(kf), _p(R, _dp), _p(t, _dp),
                                      C.byref(n), ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      _p(pts, _dp), _p(uv, _dp), n.value)
    if rc != OK:
        raise Sim3OptError(rc, "read_keyframe_bin")
    k = n.value
    return dict(kf_id=kf.value, Rw2c=R.reshape(3, 3), twinc=t, point_ids=ids[:k], points_w=pts[:k],
                obs_uv=uv[:k])


def reanchor_points(old_Rt, new_states, points, obs_frame, obs_point, device=-1):
    """figureKITTIBA's point re-anchoring on the GPU; returns the corrected (n_points, 3) array."""
    rt = _f64(old_Rt).reshape(-1, 12)
    st = _f64(new_states).reshape(-1, 8)
    pts = _f64(points).reshape(-1, 3).copy()
    of, op = _i32(obs_frame), _i32(obs_point)
    rc = load().sim3opt_reanchor_points(rt.shape[0], _p(rt, _dp), _p(st, _dp), pts.shape[0],
                                        _p(pts, _dp), of.shape[0], _p(of, _ip), _p(op, _ip),
                                        int(device))
    if rc != OK:
        raise Sim3OptError(rc, "reanchor_points")
    return pts


def write_bal(path, Rw2c, tw2c, f_k1_k2, points, obs_cam, obs_point, obs_uv):
    """SaveBALFile (drawPTAMPoints.cpp:218-283): cameras (R_w2c row-major, t_w2c), points, observations."""
    R = _f64(Rw2c).reshape(-1, 9)
    t = _f64(tw2c).reshape(-1, 3)
    fk = _f64(f_k1_k2).reshape(3)
    pts = _f64(points).reshape(-1, 3)
    oc, op = _i32(obs_cam), _i32(obs_point)
    uv = _f64(obs_uv).reshape(-1, 2)
    rc = load().sim3opt_write_bal(os.fsencode(path), R.shape[0], _p(R, _dp), _p(t, _dp), _p(fk, _dp),
                                  pts.shape[0], _p(pts, _dp), oc.shape[0], _p(oc, _ip), _p(op, _ip),
                                  _p(uv, _dp))
    if rc != OK:
        raise Sim3OptError(rc, "write_bal")


# KITTI calibration the reference hard-codes for ba_demo (bal_example.cpp:90-91, kitti_surf.cpp:52-57)
KITTI_FOCAL, KITTI_CX, KITTI_CY = 718.856, 607.1928, 185.2157


class BundleAdjuster:
    """sim3opt_ba*: the reference's ba_demo (bal_example.cpp:44-243) -- SE(3) cameras + points, Huber,
    LM over the Schur complement -- on the GPU.  Mirrors the demo's flow: read a BAL file (or hand the
    arrays over), optimize(maxIterations), write the camera poses."""

    def __init__(self, **options):
        self._L = load()
        self._b = self._L.sim3opt_ba_create()
        if not self._b:
            raise MemoryError("sim3opt_ba_create")
        if options:
            self.set_options(**options)

    def close(self):
        if self._b:
            self._L.sim3opt_ba_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != OK:
            raise Sim3OptError(rc, f"{what}: {self._L.sim3opt_ba_last_error(self._b).decode()}")

    def set_options(self, **kw):
        o = BaOptions()
        self._L.sim3opt_ba_options_default(C.byref(o))
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        self._chk(self._L.sim3opt_ba_set_options(self._b, C.byref(o)), "ba_set_options")

    def set_problem(self, cams, points, obs_cam, obs_point, obs_uv, focal=KITTI_FOCAL, cx=KITTI_CX,
                    cy=KITTI_CY):
        cq, pts = _f64(cams).reshape(-1, 7), _f64(points).reshape(-1, 3)
        oc, op, uv = _i32(obs_cam), _i32(obs_point), _f64(obs_uv).reshape(-1, 2)
        if not (oc.shape[0] == op.shape[0] == uv.shape[0]):
            raise ValueError("observation arrays differ in length")
        self._chk(self._L.sim3opt_ba_set_problem(self._b, cq.shape[0], _p(cq, _dp), pts.shape[0],
                                                 _p(pts, _dp), oc.shape[0], _p(oc, _ip), _p(op, _ip),
                                                 _p(uv, _dp), focal, cx, cy), "ba_set_problem")

    def set_fixed_cameras(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape[0] != self.dims()[0]:
            raise ValueError("one flag per camera")
        self._chk(self._L.sim3opt_ba_set_fixed_cameras(self._b, _p(m, _up)), "ba_set_fixed_cameras")

    def read_bal(self, path, focal=KITTI_FOCAL, cx=KITTI_CX, cy=KITTI_CY):
        self._chk(self._L.sim3opt_ba_read_bal(self._b, os.fsencode(path), focal, cx, cy), "ba_read_bal")

    def dims(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._L.sim3opt_ba_dims(self._b, C.byref(a), C.byref(b), C.byref(c)), "ba_dims")
        return a.value, b.value, c.value

    def chi2(self):
        v = C.c_double()
        self._chk(self._L.sim3opt_ba_chi2(self._b, C.byref(v)), "ba_chi2")
        return v.value

    def optimize(self, max_iters=5):
        """LM iterations performed (0 = failure, as g2o); raises when the library reports an error."""
        n = self._L.sim3opt_ba_optimize(self._b, int(max_iters))
        if n <= 0:
            msg = self._L.sim3opt_ba_last_error(self._b).decode()
            if msg:
                raise Sim3OptError(n, f"ba_optimize: {msg}")
        return n

    def cameras(self):
        nc = self.dims()[0]
        out = np.empty((nc, 7))
        self._chk(self._L.sim3opt_ba_get_cameras(self._b, _p(out, _dp)), "ba_get_cameras")
        return out

    def points(self):
        n = self.dims()[1]
        out = np.empty((n, 3))
        self._chk(self._L.sim3opt_ba_get_points(self._b, _p(out, _dp)), "ba_get_points")
        return out

    def stats(self):
        out = []
        for i in range(self._L.sim3opt_ba_num_iterations(self._b)):
            st = IterStats()
            self._chk(self._L.sim3opt_ba_get_stats(self._b, i, C.byref(st)), "ba_get_stats")
            out.append({k: getattr(st, k) for k, _ in IterStats._fields_})
        return out

    def write_poses(self, path):
        self._chk(self._L.sim3opt_ba_write_poses(self._b, os.fsencode(path)), "ba_write_poses")


def align_trajectory(query_xyz, train_xyz, with_scale=True):
    """(S 4x4, rmse, max_dev) of the Umeyama alignment query -> train (kitti_surf.cpp:1091-1161)."""
    q, t = _f64(query_xyz).reshape(-1, 3), _f64(train_xyz).reshape(-1, 3)
    S = np.empty(16)
    rm, mx = C.c_double(), C.c_double()
    rc = load().sim3opt_align_trajectory(q.shape[0], _p(q, _dp), _p(t, _dp), int(with_scale),
                                         _p(S, _dp), C.byref(rm), C.byref(mx))
    if rc != OK:
        raise Sim3OptError(rc, "align_trajectory")
    return S.reshape(4, 4), rm.value, mx.value


def partition_rows_equal(n_block_rows, world):
    out = np.empty(world + 1, dtype=np.int32)
    rc = load().sim3opt_partition_rows_equal(int(n_block_rows), int(world), _p(out, _ip))
    if rc != OK:
        raise Sim3OptError(rc, "partition_rows_equal")
    return out


def partition_rows(rowptr, world):
    rp = _i32(rowptr)
    out = np.empty(world + 1, dtype=np.int32)
    rc = load().sim3opt_partition_rows(rp.shape[0] - 1, _p(rp, _ip), int(world), _p(out, _ip))
    if rc != OK:
        raise Sim3OptError(rc, "partition_rows")
    return out
