"""ctypes binding of the CPU oracle (oracle/mna_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and
by bench.py's cpu_baseline leg -- never by anything under circuitsimulator_amd/.
Pinning status: see oracle/mna_oracle.h ("parity unpinned" by the reference's
own tests; pinned against SURVEY.md-recorded reference outputs).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
        L.oracle_lu_decompose.restype = C.c_int
        L.oracle_lu_decompose.argtypes = [C.c_int, vp, vp, vp]
        L.oracle_solve_lu.restype = C.c_uint
        L.oracle_solve_lu.argtypes = [C.c_int, vp, vp, vp]
        L.oracle_stamp_dc.restype = None
        L.oracle_stamp_dc.argtypes = [vp, vp, i64, vp, dbl, dbl, vp, vp]
        L.oracle_stamp_tran.restype = None
        L.oracle_stamp_tran.argtypes = [vp, vp, i64, vp, vp, dbl, dbl, vp, vp]
        L.oracle_dc.restype = C.c_int
        L.oracle_dc.argtypes = [vp, vp, i64, vp, vp, vp]
        L.oracle_solve_gs.restype = C.c_int
        L.oracle_solve_gs.argtypes = [C.c_int, vp, vp, vp, C.c_int, dbl, vp]
        L.oracle_dc_gs.restype = C.c_int
        L.oracle_dc_gs.argtypes = [vp, vp, i64, vp, vp, vp]
        L.oracle_tran.restype = i64
        L.oracle_tran.argtypes = [vp, vp, i64, dbl, dbl, dbl, vp, vp, i64, vp, vp, vp, vp, vp]
        L.oracle_tran_batch_mt.restype = C.c_int
        L.oracle_tran_batch_mt.argtypes = [vp, vp, i64, C.c_int, C.c_int, dbl, dbl, C.c_int, vp]
        L.oracle_tran_num_steps.restype = i64
        L.oracle_tran_num_steps.argtypes = [dbl, dbl]
        L.oracle_write_csv_rows.restype = C.c_int
        L.oracle_write_csv_rows.argtypes = [C.c_char_p, C.c_char_p, vp, i64, C.c_int]
        L.oracle_pivot_log_reset.restype = None
        L.oracle_pivot_log_reset.argtypes = [C.c_int]
        L.oracle_pivot_log_distinct.restype = C.c_int
        L.oracle_pivot_log_get.restype = C.c_int
        L.oracle_pivot_log_get.argtypes = [C.c_int, vp, C.c_int]
        _lib = L
    return _lib


def _col(table, b):
    """(pointer, stride) of instance b in a slot-major [P][B] table, or of a flat [P] vector."""
    table = np.ascontiguousarray(table, dtype=np.float64)
    if table.ndim == 1:
        return table, table.ctypes.data, 1
    return table, table.ctypes.data + 8 * b, table.shape[1]


def lu_decompose(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    LU = np.zeros_like(A)
    perm = np.zeros(max(n, 1), dtype=np.int32)
    ok = lib().oracle_lu_decompose(n, A.ctypes.data, LU.ctypes.data, perm.ctypes.data)
    return bool(ok), LU, perm[:n]


def solve_lu(A, b):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(b.shape[0])
    flags = lib().oracle_solve_lu(b.shape[0], A.ctypes.data, b.ctypes.data, x.ctypes.data)
    return x, flags


def stamp_dc(ir, params, b, x, scale, gmin):
    keep, ptr, stride = _col(params, b)
    N = x.shape[0]
    G = np.zeros((N, N))
    I = np.zeros(N)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().oracle_stamp_dc(ir, ptr, stride, x.ctypes.data, scale, gmin, G.ctypes.data, I.ctypes.data)
    return G, I


def stamp_tran(ir, params, b, x, xprev, tnow, dt):
    keep, ptr, stride = _col(params, b)
    N = x.shape[0]
    G = np.zeros((N, N))
    I = np.zeros(N)
    x = np.ascontiguousarray(x, dtype=np.float64)
    xprev = np.ascontiguousarray(xprev, dtype=np.float64)
    lib().oracle_stamp_tran(ir, ptr, stride, x.ctypes.data, xprev.ctypes.data, tnow, dt, G.ctypes.data, I.ctypes.data)
    return G, I


def dc(ir, N, params, b=0):
    """-> (x[N], iters, status)"""
    keep, ptr, stride = _col(params, b)
    x = np.zeros(N)
    it = C.c_int32()
    st = C.c_uint32()
    rc = lib().oracle_dc(ir, ptr, stride, x.ctypes.data, C.byref(it), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle_dc failed: %d" % rc)
    return x, it.value, st.value


def solve_gs(A, b, x0=None, max_iters=1000, tol=1e-10):
    """Solver::solveLinearSystemGaussSeidel -> (x, sweeps)"""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x0a = np.ascontiguousarray(x0, dtype=np.float64) if x0 is not None else None
    x = np.zeros(b.shape[0])
    sweeps = lib().oracle_solve_gs(b.shape[0], A.ctypes.data, b.ctypes.data,
                                   x0a.ctypes.data if x0a is not None else None, max_iters, tol, x.ctypes.data)
    return x, sweeps


def dc_gs(ir, N, params, b=0):
    """dcSolveGaussSeidel -> (x[N], iters, status)"""
    keep, ptr, stride = _col(params, b)
    x = np.zeros(N)
    it = C.c_int32()
    st = C.c_uint32()
    rc = lib().oracle_dc_gs(ir, ptr, stride, x.ctypes.data, C.byref(it), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle_dc_gs failed: %d" % rc)
    return x, it.value, st.value


def tran(ir, N, params, b, tstep, tstop, tstart=0.0, x0=None, want_rows=True, want_step_iters=False):
    """-> dict(rows [n_rows][1+N] or None, x_final, iters, status, n_steps, step_iters)"""
    keep, ptr, stride = _col(params, b)
    ns = lib().oracle_tran_num_steps(tstep, tstop)
    if ns < 0:
        raise RuntimeError("oracle_tran: invalid .TRAN numbers (tstep and tstop must be > 0)")
    rows = np.zeros((ns + 1, N + 1)) if want_rows else None
    nrows = C.c_int64()
    its = C.c_int64()
    st = C.c_uint32()
    xf = np.zeros(N)
    si = np.zeros(max(ns, 1), dtype=np.int32) if want_step_iters else None
    x0a = np.ascontiguousarray(x0, dtype=np.float64) if x0 is not None else None
    rc = lib().oracle_tran(ir, ptr, stride, tstep, tstop, tstart,
                           x0a.ctypes.data if x0a is not None else None,
                           rows.ctypes.data if rows is not None else None, ns + 1, C.byref(nrows),
                           xf.ctypes.data, C.byref(its), si.ctypes.data if si is not None else None,
                           C.byref(st))
    if rc < 0:
        raise RuntimeError("oracle_tran failed: %d" % rc)
    return dict(rows=rows[:nrows.value] if rows is not None else None, x_final=xf, iters=its.value,
                status=st.value, n_steps=rc, step_iters=si[:ns] if si is not None else None)


def tran_batch_mt(ir, params, b_first, n_inst, tstep, tstop, n_threads):
    """n_inst instances of a slot-major [P][B] table on n_threads host threads -> (iters [n_inst], threads run)"""
    table = np.ascontiguousarray(params, dtype=np.float64)
    assert table.ndim == 2 and b_first + n_inst <= table.shape[1]
    its = np.zeros(max(n_inst, 1), dtype=np.int64)
    ran = lib().oracle_tran_batch_mt(ir, table.ctypes.data, table.shape[1], b_first, n_inst, tstep, tstop, n_threads,
                                     its.ctypes.data)
    if ran < 0:
        raise RuntimeError("oracle_tran_batch_mt failed")
    return its[:n_inst], ran


def pivot_log(enable=True):
    lib().oracle_pivot_log_reset(1 if enable else 0)


def pivot_sequences():
    L = lib()
    out = []
    for w in range(L.oracle_pivot_log_distinct()):
        buf = np.full(600, -1, dtype=np.int32)
        n = L.oracle_pivot_log_get(w, buf.ctypes.data, 600)
        out.append([(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n)])
    return out
