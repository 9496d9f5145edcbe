"""Host-side mirror of the reference's analysis interface over the C-ABI.

Netlist  ~ parseNetlist() + Circuit::assignEquationIndices()   (src/main.cpp:29,34)
Engine.dc    ~ computeDcOperatingPoint()                       (include/tanalisis.hpp:9)
Engine.tran  ~ runTransientAnalysisBackwardEuler()             (include/tanalisis.hpp:15-17)
lu_solve_batch ~ Solver::solveLinearSystemLU()                 (include/solver.hpp:83-131)

for a BATCH of instances.  torch is used for device memory and streams only;
all arithmetic happens in the HIP kernels behind include/csim.h.
"""
import ctypes as C

import numpy as np

from . import capi


class Netlist:
    """A parsed, indexed and flattened netlist (host only)."""

    def __init__(self, handle, source=None):
        self._h = handle
        self.source = source
        L = capi.lib()
        c = [C.c_int32() for _ in range(5)]
        capi.check(L.csim_netlist_counts(self._h, *[C.byref(v) for v in c]))
        self.n_nodes, self.n_elems, self.n_unknowns, self.n_node_eq, self.n_branch_eq = [v.value for v in c]
        self.eq_names = [L.csim_netlist_eq_name(self._h, i).decode() for i in range(self.n_unknowns)]
        en, ts, tp, t0 = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
        capi.check(L.csim_netlist_tran(self._h, C.byref(en), C.byref(ts), C.byref(tp), C.byref(t0)))
        self.tran_enabled, self.tstep, self.tstop, self.tstart = bool(en.value), ts.value, tp.value, t0.value
        self.probes = [L.csim_netlist_probe_eq(self._h, i) for i in range(L.csim_netlist_num_probes(self._h))]
        need = L.csim_netlist_csv_header(self._h, None, 0)
        buf = C.create_string_buffer(need + 1)
        L.csim_netlist_csv_header(self._h, buf, need + 1)
        self.csv_header = buf.value.decode()
        self.n_params = self._n_params()
        self.nominal_params = np.zeros(self.n_params, dtype=np.float64)
        capi.check(L.csim_netlist_nominal_params(self._h, self.nominal_params.ctypes.data))
        self.mc_kinds = np.zeros(self.n_params, dtype=np.int32)
        capi.check(L.csim_netlist_mc_kinds(self._h, self.mc_kinds.ctypes.data))

    @property
    def has_nonlinear(self):
        """any MOSFET -> Newton DC (csim_ir.has_nonlinear)"""
        ir = capi.lib().csim_netlist_ir(self._h)
        return bool(C.cast(ir, C.POINTER(C.c_int32))[5])

    def _n_params(self):
        # csim_ir: int32 n_unknowns, n_node_eq, n_branch_eq, n_elems, n_params, ...
        ir = capi.lib().csim_netlist_ir(self._h)
        return int(C.cast(ir, C.POINTER(C.c_int32))[4])

    @classmethod
    def from_file(cls, path):
        h = C.c_void_p()
        capi.check(capi.lib().csim_netlist_parse_file(str(path).encode(), C.byref(h)))
        return cls(h, source=str(path))

    @classmethod
    def from_text(cls, text):
        data = text.encode() if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        capi.check(capi.lib().csim_netlist_parse_text(data, len(data), C.byref(h)))
        return cls(h, source="<memory>")

    @property
    def handle(self):
        return self._h

    @property
    def ir_ptr(self):
        return C.c_void_p(capi.lib().csim_netlist_ir(self._h))

    def node_eq(self, name):
        return capi.lib().csim_netlist_node_eq(self._h, str(name).encode())

    def num_steps(self, tstep=None, tstop=None):
        return capi.lib().csim_tran_num_steps(self.tstep if tstep is None else tstep,
                                              self.tstop if tstop is None else tstop)

    def dc_sweeps(self):
        L = capi.lib()
        out = []
        for i in range(L.csim_netlist_num_dc_sweeps(self._h)):
            e, a, b, s = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
            capi.check(L.csim_netlist_dc_sweep(self._h, i, C.byref(e), C.byref(a), C.byref(b), C.byref(s)))
            out.append((e.value, a.value, b.value, s.value))
        return out

    def dc_sweep_table(self, i=0):
        """(.DC card i) -> (swept values [n], params [P][n]); every point is one DC instance."""
        L = capi.lib()
        n = L.csim_netlist_dc_sweep_points(self._h, i)
        params = np.zeros((self.n_params, n), dtype=np.float64)
        values = np.zeros(n, dtype=np.float64)
        if n > 0:
            capi.check(L.csim_netlist_dc_sweep_params(self._h, i, n, params.ctypes.data, values.ctypes.data))
        return values, params

    def mc_params_host(self, seed, sigma, b_first, B):
        """Host mirror of the device generator: numpy [P][B] (slot-major)."""
        out = np.zeros((self.n_params, B), dtype=np.float64)
        capi.check(capi.lib().csim_mc_params_host(self._h, seed, sigma, b_first, B, out.ctypes.data))
        return out

    def nominal_table(self, B):
        return np.repeat(self.nominal_params[:, None], B, axis=1).copy()

    def __del__(self):
        try:
            if self._h:
                capi.lib().csim_netlist_free(self._h)
                self._h = None
        except Exception:
            pass


def _torch():
    import torch
    return torch


class Engine:
    """One engine per (circuit, GPU).  Raises CsimError if no HIP device."""

    def __init__(self, netlist, device=0):
        self.netlist = netlist
        self.device = int(device)
        h = C.c_void_p()
        capi.check(capi.lib().csim_engine_create(netlist.handle, self.device, C.byref(h)))
        self._h = h
        self.N = netlist.n_unknowns
        self.P = netlist.n_params

    @property
    def tran_kernel(self):
        return capi.lib().csim_engine_tran_kernel(self._h).decode()

    @property
    def sched_info(self):
        """csim_engine_sched_info parsed: dict(text=..., ops=dict(fma, mul, addsub, recip, cmp)) or None"""
        text = capi.lib().csim_engine_sched_info(self._h).decode()
        if not text:
            return None
        ops = {}
        if "ops_per_solve:" in text:
            for item in text.split("ops_per_solve:")[1].split():
                k, _, v = item.partition("=")
                if v.isdigit():
                    ops[k] = int(v)
        return dict(text=text, ops=ops)

    def lanes_for_batch(self, B):
        """lanes per instance of the transient kernel for B instances: 1 / 16 (scheduled), 64 (general)"""
        n = capi.lib().csim_engine_lanes_for_batch(self._h, int(B))
        return n if n else 64

    def set_kernel(self, which):
        capi.check(capi.lib().csim_engine_set_kernel(self._h, {"auto": 0, "general": 1, "scheduled": 2, "faithful": 3}[which]))

    def set_option(self, key, value):
        """csim_engine_set_option: hybrid_rounds, hybrid_steps, lanes_per_instance, jit_dir, ... (include/csim.h)"""
        capi.check(capi.lib().csim_engine_set_option(self._h, str(key).encode(), str(value).encode()))

    def stat(self, key):
        """csim_engine_stat: "near_verified", "near_rolled_back" (include/csim.h)"""
        return int(capi.lib().csim_engine_stat(self._h, str(key).encode()))

    # -- device-pointer forms (torch tensors on cuda:<device>, slot-major) ----
    def _dev(self):
        return "cuda:%d" % self.device

    def _stream(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def upload_params(self, table):
        """numpy [P][B] -> device tensor"""
        torch = _torch()
        t = torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(self._dev())
        assert t.shape[0] == self.P
        return t

    def mc_params(self, seed, sigma, b_first, B):
        torch = _torch()
        out = torch.empty((self.P, B), dtype=torch.float64, device=self._dev())
        capi.check(capi.lib().csim_mc_params_dev(self._h, seed, sigma, b_first, B, out.data_ptr(), self._stream()))
        return out

    def dc(self, params):
        """params: device [P][B] -> (x [N][B], iters [B] int32, status [B] int32-bits)"""
        torch = _torch()
        B = params.shape[1]
        x = torch.empty((self.N, B), dtype=torch.float64, device=self._dev())
        it = torch.zeros(B, dtype=torch.int32, device=self._dev())
        st = torch.zeros(B, dtype=torch.int32, device=self._dev())
        capi.check(capi.lib().csim_dc_batch_dev(self._h, params.data_ptr(), B, x.data_ptr(), it.data_ptr(),
                                                st.data_ptr(), self._stream()))
        return x, it, st

    def dc_gs(self, params):
        """dcSolveGaussSeidel for a batch: params device [P][B] -> (x [N][B], iters, status)"""
        torch = _torch()
        B = params.shape[1]
        x = torch.empty((self.N, B), dtype=torch.float64, device=self._dev())
        it = torch.zeros(B, dtype=torch.int32, device=self._dev())
        st = torch.zeros(B, dtype=torch.int32, device=self._dev())
        capi.check(capi.lib().csim_dc_gs_batch_dev(self._h, params.data_ptr(), B, x.data_ptr(), it.data_ptr(),
                                                   st.data_ptr(), self._stream()))
        return x, it, st

    def dc_sweep(self, i=0):
        """Execute .DC card i of the netlist as one batch: -> (values, x [N][n], iters, status)."""
        values, table = self.netlist.dc_sweep_table(i)
        x, it, st = self.dc(self.upload_params(table))
        return values, x, it, st

    def tran(self, params, x, tstep, step_first, n_steps, iters, status, probes=None, out_stride=1,
             wave=None, step_iters=None):
        """Advance the batch by n_steps time steps in place (x, iters, status are updated)."""
        B = params.shape[1]
        n_probe = 0
        pe = None
        if wave is not None:
            pe = (C.c_int32 * len(probes))(*probes)
            n_probe = len(probes)
        capi.check(capi.lib().csim_tran_batch_dev(
            self._h, params.data_ptr(), B, float(tstep), int(step_first), int(n_steps), pe, n_probe,
            int(out_stride), wave.data_ptr() if wave is not None else None, x.data_ptr(), iters.data_ptr(),
            status.data_ptr(), step_iters.data_ptr() if step_iters is not None else None, self._stream()))

    # -- host-pointer forms (numpy, instance-major as in SURVEY.md 8b) --------
    def dc_host(self, params=None, B=1):
        if params is not None:
            params = np.ascontiguousarray(params, dtype=np.float64)
            B = params.shape[0]
        x = np.zeros((B, self.N))
        it = np.zeros(B, dtype=np.int32)
        st = np.zeros(B, dtype=np.uint32)
        capi.check(capi.lib().csim_dc_batch(self._h, params.ctypes.data if params is not None else None, B,
                                            x.ctypes.data, it.ctypes.data, st.ctypes.data))
        return x, it, st

    def tran_host(self, params=None, B=1, tstep=None, tstop=None, tstart=None, probes=None, out_stride=1):
        nl = self.netlist
        tstep = nl.tstep if tstep is None else tstep
        tstop = nl.tstop if tstop is None else tstop
        tstart = nl.tstart if tstart is None else tstart
        if params is not None:
            params = np.ascontiguousarray(params, dtype=np.float64)
            B = params.shape[0]
        wave = None
        pe = None
        n_probe = 0
        if probes is not None:
            n_probe = len(probes)
            pe = (C.c_int32 * n_probe)(*probes)
            rows = capi.lib().csim_tran_num_rows(tstep, tstop, tstart, out_stride)
            wave = np.zeros((B, rows, n_probe))
        xf = np.zeros((B, self.N))
        it = np.zeros(B, dtype=np.int64)
        st = np.zeros(B, dtype=np.uint32)
        capi.check(capi.lib().csim_tran_batch(
            self._h, params.ctypes.data if params is not None else None, B, tstep, tstop, tstart, pe, n_probe,
            out_stride, wave.ctypes.data if wave is not None else None, xf.ctypes.data, it.ctypes.data,
            st.ctypes.data))
        return wave, xf, it, st

    def write_csv(self, path, params=None, instance=0, tstep=None, tstop=None, tstart=None, probes=None):
        """csim_tran_write_csv: the transient of one instance of params ([B][P] numpy, instance-major; None = nominal)
        as the reference's CSV.  probes None: the netlist's .PLOTNV/.PRINT probes if any, else every unknown."""
        nl = self.netlist
        tstep = nl.tstep if tstep is None else tstep
        tstop = nl.tstop if tstop is None else tstop
        tstart = nl.tstart if tstart is None else tstart
        B = 1
        if params is not None:
            params = np.ascontiguousarray(params, dtype=np.float64)
            B = params.shape[0]
        pe, n_probe = None, 0
        if probes:
            n_probe = len(probes)
            pe = (C.c_int32 * n_probe)(*probes)
        capi.check(capi.lib().csim_tran_write_csv(
            self._h, params.ctypes.data if params is not None else None, B, int(instance), tstep, tstop, tstart,
            pe, n_probe, str(path).encode()))

    def jit_scheduled(self, params, tstep=None, plan_steps=200):
        """Plan + generate + hipcc + load the lane-per-instance kernel for this netlist (needs hipcc)."""
        tstep = self.netlist.tstep if tstep is None else tstep
        capi.check(capi.lib().csim_engine_jit_scheduled(self._h, params.data_ptr(), params.shape[1], float(tstep),
                                                        int(plan_steps)))

    @staticmethod
    def _positions(schedules, N):
        """["0:21,8:22", "-", ...] -> int32 [n][N] pivot row positions"""
        pos = np.tile(np.arange(N, dtype=np.int32), (len(schedules), 1))
        for a, text in enumerate(schedules):
            for item in text.replace("-", "").split(","):
                if item.strip():
                    k, p = item.split(":")
                    pos[a, int(k)] = int(p)
        return np.ascontiguousarray(pos)

    def jit_with_schedules(self, schedules, dc_schedules=()):
        """Generate + hipcc + load the kernels for explicit pivot schedules ("k:p,k:p" strings as
        returned by record_pivot_schedules / record_dc_pivot_schedules)."""
        pos = self._positions(list(schedules), self.N)
        dpos = self._positions(list(dc_schedules), self.N) if len(dc_schedules) else None
        capi.check(capi.lib().csim_engine_jit_with_schedules(
            self._h, pos.ctypes.data, pos.shape[0], dpos.ctypes.data if dpos is not None else None,
            dpos.shape[0] if dpos is not None else 0))

    @staticmethod
    def _schedules_in(info_text):
        """(transient, dc) schedule strings out of a csim_sched_info text"""
        if not info_text or "schedule=" not in info_text:
            return [], []
        body = info_text.split("schedule=", 1)[1].split(" lds_doubles", 1)[0]
        tran, dc = [], []
        for item in body.split(";"):
            item = item.strip()
            if not item:
                continue
            if item.startswith("dc "):
                dc.append(item[3:].strip())
            else:
                tran.append(item)
        return tran, dc

    def loaded_schedules(self):
        """(transient, dc) schedule strings of the generated library in use, in its order ([] / [] without one)."""
        info = self.sched_info
        return self._schedules_in(info["text"] if info else "")

    def refine_schedules(self, params, status, tstep=None, n_steps=300, max_instances=8, max_new=8):
        """Instances a run flagged CSIM_ST_SCHED_FALLBACK used pivot sequences the generated kernels do not carry:
        they then finish on the general kernel, a handful of waves alone on the chip (buffer.sp, 4 096 Monte-Carlo
        instances at sigma 5 %: 31 instances, 0.8 % of the work, most of the wall time).  This replays up to
        `max_instances` of them through the planner, appends the sequences it has not seen to the loaded ones and
        re-specialises (generate + hipcc + load; needs hipcc).  `status` = the status words of that run (tensor or
        array).  Returns the number of sequences added (0: nothing done).  Results never depend on the list --
        every factorisation verifies the sequence it uses -- only how many instances stay on the fast kernels."""
        st = status.cpu().numpy() if hasattr(status, "cpu") else np.asarray(status)
        flagged = np.nonzero((st & 0x20) != 0)[0]
        known, known_dc = self.loaded_schedules()
        if not len(flagged) or not known:
            return 0
        have = set(known)
        new = {}
        for b in flagged[:max_instances]:
            alts, _ = self.record_pivot_schedules(params, int(b), tstep, n_steps)
            for sched, n in alts:
                if sched not in have:
                    new[sched] = new.get(sched, 0) + n
        if not new:
            return 0
        room = min(max_new, 16 - len(known))          # csim_engine_jit_with_schedules takes up to 16 transient sequences
        added = [s for s, _ in sorted(new.items(), key=lambda kv: -kv[1])][:max(0, room)]
        if not added:
            return 0
        self.jit_with_schedules(known + added, known_dc)
        return len(added)

    def record_pivot_schedule(self, params, instance=0, tstep=None, n_steps=200):
        """Planner: pivot row position per column of the first transient factorisation of one
        instance (general kernel), as (schedule string, #factorisations, #with another sequence)."""
        tstep = self.netlist.tstep if tstep is None else tstep
        pos = np.zeros(self.N, dtype=np.int32)
        nlu, ndiff = C.c_int64(), C.c_int64()
        capi.check(capi.lib().csim_record_pivot_schedule(self._h, params.data_ptr(), params.shape[1], instance,
                                                         float(tstep), int(n_steps), pos.ctypes.data,
                                                         C.byref(nlu), C.byref(ndiff)))
        sched = ",".join("%d:%d" % (k, p) for k, p in enumerate(pos) if p != k)
        return sched, nlu.value, ndiff.value

    def record_pivot_schedules(self, params, instance=0, tstep=None, n_steps=200, max_alts=8):
        """Planner: every distinct pivot sequence of one instance's transient factorisations,
        most frequent first: ([(schedule string, count), ...], n_other)."""
        tstep = self.netlist.tstep if tstep is None else tstep
        pos = np.zeros((max_alts, self.N), dtype=np.int32)
        counts = np.zeros(max_alts, dtype=np.int64)
        n_alts, other = C.c_int32(), C.c_int64()
        capi.check(capi.lib().csim_record_pivot_schedules(self._h, params.data_ptr(), params.shape[1], instance,
                                                          float(tstep), int(n_steps), max_alts, pos.ctypes.data,
                                                          counts.ctypes.data, C.byref(n_alts), C.byref(other)))
        out = []
        for a in range(n_alts.value):
            sched = ",".join("%d:%d" % (k, p) for k, p in enumerate(pos[a]) if p != k) or "-"
            out.append((sched, int(counts[a])))
        return out, other.value

    def record_dc_pivot_schedules(self, params, instance=0, max_alts=8):
        """Planner on the DC operating point of one instance: ([(schedule string, count), ...], n_other)."""
        pos = np.zeros((max_alts, self.N), dtype=np.int32)
        counts = np.zeros(max_alts, dtype=np.int64)
        n_alts, other = C.c_int32(), C.c_int64()
        capi.check(capi.lib().csim_record_dc_pivot_schedules(self._h, params.data_ptr(), params.shape[1], instance,
                                                             max_alts, pos.ctypes.data, counts.ctypes.data,
                                                             C.byref(n_alts), C.byref(other)))
        out = []
        for a in range(n_alts.value):
            sched = ",".join("%d:%d" % (k, p) for k, p in enumerate(pos[a]) if p != k) or "-"
            out.append((sched, int(counts[a])))
        return out, other.value

    def close(self):
        if self._h:
            capi.lib().csim_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lu_solve_batch(A, b, device=0):
    """Batched Solver::solveLinearSystemLU on the GPU.  A [B][n][n], b [B][n] numpy."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    B, n = b.shape
    x = np.zeros((B, n))
    flags = np.zeros(B, dtype=np.uint32)
    capi.check(capi.lib().csim_lu_solve_batch(device, n, B, A.ctypes.data, b.ctypes.data, x.ctypes.data,
                                              flags.ctypes.data))
    return x, flags


def gs_solve_batch(A, b, x0=None, max_iters=1000, tol=1e-10, device=0):
    """Batched Solver::solveLinearSystemGaussSeidel on the GPU.  A [B][n][n], b/x0 [B][n] -> (x [B][n], sweeps [B])."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    B, n = b.shape
    x0a = np.ascontiguousarray(x0, dtype=np.float64) if x0 is not None else None
    x = np.zeros((B, n))
    sweeps = np.zeros(B, dtype=np.int32)
    capi.check(capi.lib().csim_gs_solve_batch(device, n, B, A.ctypes.data, b.ctypes.data,
                                              x0a.ctypes.data if x0a is not None else None, int(max_iters), float(tol),
                                              x.ctypes.data, sweeps.ctypes.data))
    return x, sweeps


def lu_decompose_batch(A, device=0):
    """Batched Solver::luDecompose on the GPU.  A [B][n][n] -> (LU [B][n][n], perm [B][n], flags [B])."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    B, n, _ = A.shape
    LU = np.zeros_like(A)
    perm = np.zeros((B, n), dtype=np.int32)
    flags = np.zeros(B, dtype=np.uint32)
    capi.check(capi.lib().csim_lu_decompose_batch(device, n, B, A.ctypes.data, LU.ctypes.data, perm.ctypes.data,
                                                  flags.ctypes.data))
    return LU, perm, flags
