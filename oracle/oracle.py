"""ctypes binding of the CPU oracle (oracle/sim3_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (sim3opt_amd/) never imports this.
Parity unpinned: see the header of oracle/sim3_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_sim3.so")


class Options(C.Structure):
    _fields_ = [
        ("tau", C.c_double),
        ("user_lambda_init", C.c_double),
        ("good_step_lower", C.c_double),
        ("good_step_upper", C.c_double),
        ("max_trials", C.c_int),
        ("fd_delta", C.c_double),
        ("exp_eps", C.c_double),
        ("small_rot_half", C.c_int),
        ("fix_small_angle_b", C.c_int),
        ("dof_mask", C.c_int),
        ("threads", C.c_int),
    ]


class Iter(C.Structure):
    _fields_ = [
        ("chi2_before", C.c_double),
        ("chi2_after", C.c_double),
        ("lambda_", C.c_double),
        ("rho", C.c_double),
        ("trials", C.c_int),
        ("solve_ok", C.c_int),
        ("t_linearize", C.c_double),
        ("t_solve", C.c_double),
        ("t_update", C.c_double),
    ]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sim3_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_sim3.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        up = C.POINTER(C.c_ubyte)
        op = C.POINTER(Options)
        L.or_options_default.argtypes = [op]
        L.or_sim3_exp.argtypes = [dp, op, dp]
        L.or_sim3_log.argtypes = [dp, op, dp]
        L.or_sim3_mul.argtypes = [dp, dp, dp]
        L.or_sim3_inv.argtypes = [dp, dp]
        L.or_quat_from_R.argtypes = [dp, dp]
        L.or_R_from_quat.argtypes = [dp, dp]
        L.or_euler_rpy_to_R.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.or_edge_error.argtypes = [dp, dp, dp, op, dp]
        L.or_edge_jacobians.argtypes = [dp, dp, dp, op, dp, dp]
        L.or_all_errors.argtypes = [C.c_int, ip, ip, dp, dp, op, dp]
        L.or_all_jacobians.argtypes = [C.c_int, ip, ip, dp, dp, op, dp, dp]
        L.or_chi2.argtypes = [C.c_int, dp, C.c_int, ip, ip, dp, dp, C.c_int, C.c_double, op]
        L.or_chi2.restype = C.c_double
        L.or_build_dense.argtypes = [C.c_int, dp, up, C.c_int, ip, ip, dp, dp, C.c_int,
                                     C.c_double, op, dp, dp]
        L.or_build_dense.restype = C.c_int
        L.or_solve_once.argtypes = [C.c_int, dp, up, C.c_int, ip, ip, dp, dp, C.c_int,
                                    C.c_double, op, C.c_double, dp, dp]
        L.or_solve_once.restype = C.c_int
        L.or_optimize.argtypes = [C.c_int, dp, up, C.c_int, ip, ip, dp, dp, C.c_int,
                                  C.c_double, C.c_int, op, C.POINTER(Iter)]
        L.or_optimize.restype = C.c_int
        L.or_last_lnz.restype = C.c_long
        _lib = L
    return _lib


def default_options(**kw):
    o = Options()
    lib().or_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- group operations on 8-double states [qx qy qz qw tx ty tz s] ----
def sim3_exp(xi, opt=None):
    opt = opt or default_options()
    xi = _f64(xi)
    out = np.empty(8)
    lib().or_sim3_exp(_dp(xi), C.byref(opt), _dp(out))
    return out


def sim3_log(S, opt=None):
    opt = opt or default_options()
    S = _f64(S)
    out = np.empty(7)
    lib().or_sim3_log(_dp(S), C.byref(opt), _dp(out))
    return out


def sim3_mul(a, b):
    a, b = _f64(a), _f64(b)
    out = np.empty(8)
    lib().or_sim3_mul(_dp(a), _dp(b), _dp(out))
    return out


def sim3_inv(a):
    a = _f64(a)
    out = np.empty(8)
    lib().or_sim3_inv(_dp(a), _dp(out))
    return out


def quat_from_R(R):
    R = _f64(R).reshape(9)
    q = np.empty(4)
    lib().or_quat_from_R(_dp(R), _dp(q))
    return q


def R_from_quat(q):
    q = _f64(q)
    R = np.empty(9)
    lib().or_R_from_quat(_dp(q), _dp(R))
    return R.reshape(3, 3)


def euler_rpy_to_R(r, p, y):
    R = np.empty(9)
    lib().or_euler_rpy_to_R(r, p, y, _dp(R))
    return R.reshape(3, 3)


def edge_error(Cm, S0, S1, opt=None):
    opt = opt or default_options()
    Cm, S0, S1 = _f64(Cm), _f64(S0), _f64(S1)
    e = np.empty(7)
    lib().or_edge_error(_dp(Cm), _dp(S0), _dp(S1), C.byref(opt), _dp(e))
    return e


def edge_jacobians(Cm, S0, S1, opt=None):
    """Returns (A, B) as 7x7 numpy matrices (A[r, c] = d e_r / d delta0_c)."""
    opt = opt or default_options()
    Cm, S0, S1 = _f64(Cm), _f64(S0), _f64(S1)
    A = np.empty(49)
    B = np.empty(49)
    lib().or_edge_jacobians(_dp(Cm), _dp(S0), _dp(S1), C.byref(opt), _dp(A), _dp(B))
    return A.reshape(7, 7).T.copy(), B.reshape(7, 7).T.copy()


class Graph:
    """Dense-id pose graph in the oracle's array form."""

    def __init__(self, states, fixed, v0, v1, meas, info=None, kernel=0, kdelta=0.0):
        self.states = _f64(states).reshape(-1, 8).copy()
        self.fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        self.v0 = _i32(v0)
        self.v1 = _i32(v1)
        self.meas = _f64(meas).reshape(-1, 8)
        self.info = None if info is None else _f64(info).reshape(-1, 49)
        self.kernel = int(kernel)
        self.kdelta = float(kdelta)

    @property
    def nv(self):
        return self.states.shape[0]

    @property
    def ne(self):
        return self.v0.shape[0]

    @property
    def n_free(self):
        return int((self.fixed == 0).sum())

    def _args(self, states=None):
        st = self.states if states is None else states
        return (self.nv, _dp(st), )

    def errors(self, opt=None):
        opt = opt or default_options()
        e = np.empty((self.ne, 7))
        lib().or_all_errors(self.ne, _ip(self.v0), _ip(self.v1), _dp(self.meas),
                            _dp(self.states), C.byref(opt), _dp(e))
        return e

    def jacobians(self, opt=None):
        """(A, B) each (ne, 7, 7) with [k, r, c] indexing."""
        opt = opt or default_options()
        A = np.empty((self.ne, 49))
        B = np.empty((self.ne, 49))
        lib().or_all_jacobians(self.ne, _ip(self.v0), _ip(self.v1), _dp(self.meas),
                               _dp(self.states), C.byref(opt), _dp(A), _dp(B))
        return (A.reshape(-1, 7, 7).transpose(0, 2, 1).copy(),
                B.reshape(-1, 7, 7).transpose(0, 2, 1).copy())

    def chi2(self, opt=None):
        opt = opt or default_options()
        return lib().or_chi2(self.nv, _dp(self.states), self.ne, _ip(self.v0), _ip(self.v1),
                             _dp(self.meas), _dp(self.info), self.kernel, self.kdelta,
                             C.byref(opt))

    def build_dense(self, opt=None):
        """(H, b): H is (n, n) symmetric, b is (n,), n = 7 * n_free."""
        opt = opt or default_options()
        n = 7 * self.n_free
        H = np.empty(n * n)
        b = np.empty(n)
        lib().or_build_dense(self.nv, _dp(self.states), _up(self.fixed), self.ne,
                             _ip(self.v0), _ip(self.v1), _dp(self.meas), _dp(self.info),
                             self.kernel, self.kdelta, C.byref(opt), _dp(H), _dp(b))
        return H.reshape(n, n).T.copy(), b

    def solve_once(self, lam, opt=None):
        """Returns (ok, x, b) of (H + lam I) x = b via the oracle's sparse LDL^T."""
        opt = opt or default_options()
        n = 7 * self.n_free
        x = np.zeros(n)
        b = np.zeros(n)
        ok = lib().or_solve_once(self.nv, _dp(self.states), _up(self.fixed), self.ne,
                                 _ip(self.v0), _ip(self.v1), _dp(self.meas), _dp(self.info),
                                 self.kernel, self.kdelta, C.byref(opt), lam, _dp(x), _dp(b))
        return bool(ok), x, b

    def optimize(self, max_iters, opt=None):
        """Runs LM in place on self.states. Returns (iterations, list of Iter)."""
        opt = opt or default_options()
        trace = (Iter * max(max_iters, 1))()
        it = lib().or_optimize(self.nv, _dp(self.states), _up(self.fixed), self.ne,
                               _ip(self.v0), _ip(self.v1), _dp(self.meas), _dp(self.info),
                               self.kernel, self.kdelta, max_iters, C.byref(opt), trace)
        return it, [trace[i] for i in range(max(it, 0))]
