"""Netlist front-end: dialect, number parsing, equation indexing (bit-exact with the
reference's src/circuit.cpp:42-61, per the eq maps recorded in SURVEY.md 8d/Appendix D)."""
import ctypes as C

import numpy as np
import pytest

from circuitsimulator_amd import Netlist, CsimError, capi


def test_buffer_summary_and_eq_map(buffer_nl, anchors):
    a = anchors["buffer"]
    assert buffer_nl.n_unknowns == a["n_unknowns"]
    assert buffer_nl.n_node_eq == a["n_node_eq"]
    assert buffer_nl.n_branch_eq == a["n_branch_eq"]
    assert buffer_nl.n_elems == a["n_elems"]
    assert buffer_nl.eq_names == a["eq_names"]
    assert buffer_nl.n_params == 36


def test_dbmixer_summary_and_eq_map(dbmixer_nl, anchors):
    a = anchors["dbmixer"]
    assert dbmixer_nl.n_unknowns == a["n_unknowns"]
    assert dbmixer_nl.n_node_eq == a["n_node_eq"]
    assert dbmixer_nl.n_branch_eq == a["n_branch_eq"]
    assert dbmixer_nl.n_elems == a["n_elems"]
    assert dbmixer_nl.eq_names == a["eq_names"]
    assert dbmixer_nl.n_params == 79


def test_tran_card_and_probes(buffer_nl, dbmixer_nl):
    assert buffer_nl.tran_enabled and buffer_nl.tstep == 1e-9 and buffer_nl.tstop == 300e-9
    assert buffer_nl.num_steps() == 300
    assert dbmixer_nl.tstep == 1e-13 and dbmixer_nl.tstop == 5e-9 and dbmixer_nl.num_steps() == 50000
    # .PLOTNV 101 / 118 (buffer), 102 / 103 (dbmixer); commented cards are ignored
    assert buffer_nl.probes == [buffer_nl.node_eq("101"), buffer_nl.node_eq("118")] == [1, 8]
    assert dbmixer_nl.probes == [1, 2]


def test_csv_header(buffer_nl):
    assert buffer_nl.csv_header == ("time,V(103),V(101),V(102),V(107),V(104),V(115),V(116),V(117),V(118),"
                                    "I(VDD),I(Vin),I(L1),I(L2)")


def test_nominal_params_bits(buffer_nl):
    p = buffer_nl.nominal_params
    # VDD: dc=3; Vin: SIN 1.5 2 10e6 0; Rin=10
    assert list(p[0:6]) == [3.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    assert list(p[6:12]) == [0.0, 1.5, 2.0, 10e6, 0.0, 0.0]
    assert p[12] == 10.0
    # M1: PMOS model 1: Vth=|VT|, K = MU*COX*(W/L) in the association of circuit.cpp:144
    assert p[13] == 0.75 and p[14] == 5e-2 * 0.3e-4 * (30e-6 / 0.35e-6) and p[15] == 0.05 and p[16] == 4.0e-14


DIALECT = """* title comment
V1 in 0 DC 2.5   $ inline comment
R1 in  mid 1k
+
r2 mid 0 2.2K    ; trailing semicolon is data on this line, not a comment
; full-line comment
C1 mid GND 1p
L1 mid out 3n
R3 out 0 10meg
I1 0 out 1m
M1 out mid 0 nch 2u
+ 1u
.model nch VT 0.5 MU 1e-2 COX 1e-3
+ LAMBDA 0.02 CJ0 1f
.tran 1n 10n 2n
.op
.dc V1 0 1 0.1
.print tran V(out) V(in,mid) I(V1)
.plotnv mid
.plotnc M1(d) R1
.weird card
"""


def test_dialect():
    nl = Netlist.from_text(DIALECT)
    # nodes in creation order: in, 0, mid, out  -> eq: in=0, mid=1, out=2; branches V1=3, L1=4
    assert nl.eq_names == ["in", "mid", "out", "V1", "L1"]
    assert nl.node_eq("0") == -1 and nl.node_eq("GND") == -1 and nl.node_eq("nope") == -2
    assert nl.n_elems == 8
    p = nl.nominal_params
    # V1 (6) R1 R2 C1 L1 R3 I1(6) M1(4)
    assert p[0] == 2.5
    assert p[6] == 1e3 and p[7] == 2.2 * 1e3
    assert p[8] == 1 * 1e-12 and p[9] == 3 * 1e-9 and p[10] == 10 * 1e6
    assert p[11] == 1 * 1e-3
    assert p[17] == 0.5 and p[18] == 1e-2 * 1e-3 * (2 * 1e-6 / (1 * 1e-6)) and p[19] == 0.02 and p[20] == 1 * 1e-15
    assert nl.tran_enabled and (nl.tstep, nl.tstop, nl.tstart) == (1e-9, 10e-9, 2e-9)
    assert nl.dc_sweeps() == [(0, 0.0, 1.0, 0.1)]
    assert nl.probes == [2, 1]          # V(out) from .print, then mid from .plotnv; V(in,mid) is a diff probe
    assert list(nl.mc_kinds) == [0] * 6 + [1, 1, 1, 1, 1] + [0] * 6 + [1, 2, 0, 0]


def test_model_after_use_and_pmos_sign():
    nl = Netlist.from_text("M1 d g s p 10u 1u 7\nVd d 0 1\nVg g 0 1\nVs s 0 1\n.MODEL 7 VT -0.6 MU 2e-2 COX 1e-3\n")
    ir = C.cast(capi.lib().csim_netlist_ir(nl.handle), C.POINTER(C.c_int32))
    assert nl.n_elems == 4
    assert nl.nominal_params[0] == 0.6          # |VT|
    assert nl.mc_kinds[0] == 1 and nl.mc_kinds[1] == 2
    assert ir[5] == 1                           # has_nonlinear


def test_unknown_model_and_bad_lines_are_skipped():
    nl = Netlist.from_text("R1 a 0 1k\nM1 a b 0 nomodel 1u 1u\nR2 a\nRbad a 0 xyz\nC1 a 0 1p\n")
    assert nl.n_elems == 2 and nl.n_unknowns == 1


def test_parse_file_missing():
    with pytest.raises(CsimError) as e:
        Netlist.from_file("/nonexistent/netlist.sp")
    assert e.value.code == capi.CSIM_ERR_IO


def test_empty_netlist():
    nl = Netlist.from_text("* nothing\n")
    assert nl.n_unknowns == 0 and nl.n_elems == 0 and not nl.tran_enabled


def test_num_steps_rule():
    L = capi.lib()
    # floor(tstop/dt + 1e-12), tanalisis.cpp:238
    assert L.csim_tran_num_steps(1e-9, 300e-9) == 300
    assert L.csim_tran_num_steps(3e-11, 300e-9) == 10000
    assert L.csim_tran_num_steps(1e-13, 5e-9) == 50000
    assert L.csim_tran_num_steps(0.0, 1.0) == -1
    assert L.csim_tran_num_rows(1e-9, 10e-9, 0.0, 1) == 11
    assert L.csim_tran_num_rows(1e-9, 10e-9, 2e-9, 1) == 9      # rows with t < tstart suppressed
    assert L.csim_tran_num_rows(1e-9, 10e-9, 0.0, 5) == 3


def test_dc_sweep_table():
    nl = Netlist.from_text("V1 in 0 DC 1\nR1 in out 1k\nR2 out 0 1k\nI1 0 out 1m\n.dc V1 0 2 0.5\n.dc I1 1m 0 -0.25m\n"
                           ".dc R1 0 1 1\n.dc V1 0 1 0\n.dc V1 0 1 -1\n")
    assert nl.dc_sweeps()[0] == (0, 0.0, 2.0, 0.5)
    v, t = nl.dc_sweep_table(0)
    assert list(v) == [0.0, 0.5, 1.0, 1.5, 2.0] and t.shape == (nl.n_params, 5)
    assert list(t[0]) == [0.0, 0.5, 1.0, 1.5, 2.0]              # V1's dcValue slot
    assert np.array_equal(t[1:], np.repeat(nl.nominal_params[1:, None], 5, axis=1))
    v, t = nl.dc_sweep_table(1)                                 # descending sweep of a current source
    assert len(v) == 5 and v[0] == 1e-3 and abs(v[-1]) < 1e-18
    assert nl.dc_sweep_table(2)[0].size == 0                    # not a source
    assert nl.dc_sweep_table(3)[0].size == 0                    # zero step
    assert nl.dc_sweep_table(4)[0].size == 0                    # step of the wrong sign


def test_schedule_strings_of_a_generated_library_round_trip():
    """Engine.refine_schedules rebuilds the kernels from the schedules the loaded library reports plus new ones:
    the parse of csim_sched_info and the schedule -> pivot-position table must round-trip (host logic, no GPU)."""
    from circuitsimulator_amd.engine import Engine
    info = ("buffer.sp N=13 schedule=0:9,1:10,5:11,7:12 ; 0:9,1:10,3:9,5:11,7:12,9:11,11:12 ; - ; dc 0:9,1:10,5:11,7:12,8:12 "
            "lds_doubles_per_lane=46/46 ops_per_solve: fma=48 mul=40 addsub=81 recip=7 cmp=38")
    tran, dc = Engine._schedules_in(info)
    assert tran == ["0:9,1:10,5:11,7:12", "0:9,1:10,3:9,5:11,7:12,9:11,11:12", "-"]
    assert dc == ["0:9,1:10,5:11,7:12,8:12"]
    assert Engine._schedules_in("") == ([], []) and Engine._schedules_in("general kernel") == ([], [])
    pos = Engine._positions(tran, 13)
    assert pos.shape == (3, 13) and pos.dtype == np.int32
    assert list(pos[0]) == [9, 10, 2, 3, 4, 11, 6, 12, 8, 9, 10, 11, 12]
    assert list(pos[2]) == list(range(13))                  # "-" = no swaps
    back = [",".join("%d:%d" % (k, p) for k, p in enumerate(row) if p != k) or "-" for row in pos]
    assert back == tran
