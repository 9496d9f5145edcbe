"""The C-ABI library loads on a box without a GPU, exports every symbol include/csim.h
declares, and refuses to compute without a device (there is no CPU path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from circuitsimulator_amd import capi, Netlist, CsimError
from conftest import ROOT, has_gpu


def _declared():
    text = open(os.path.join(ROOT, "include", "csim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csim_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), "libcsim.so does not export %s" % n
        assert n in capi.PROTOTYPES, "capi.py has no prototype for %s" % n
    for n in capi.PROTOTYPES:
        assert n in names, "capi.py binds %s which include/csim.h does not declare" % n


def test_version_string():
    assert b"gfx950" in capi.lib().csim_version()


def test_ir_struct_layout_matches_header(buffer_nl):
    # csim_ir starts with six int32 counters (include/csim_ir.h)
    ir = C.cast(capi.lib().csim_netlist_ir(buffer_nl.handle), C.POINTER(C.c_int32))
    assert [ir[i] for i in range(6)] == [13, 9, 4, 14, 36, 1]


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_engine_refuses_without_gpu(buffer_nl):
    from circuitsimulator_amd import Engine, lu_solve_batch
    with pytest.raises(CsimError) as e:
        Engine(buffer_nl, 0)
    assert e.value.code == capi.CSIM_ERR_NO_DEVICE
    with pytest.raises(CsimError) as e:
        lu_solve_batch(np.eye(2)[None], np.ones((1, 2)))
    assert e.value.code == capi.CSIM_ERR_NO_DEVICE


def test_null_arguments_are_errors_not_crashes():
    L = capi.lib()
    assert L.csim_netlist_parse_file(None, None) == capi.CSIM_ERR_ARG
    h = C.c_void_p()
    assert L.csim_engine_create(None, 0, C.byref(h)) == capi.CSIM_ERR_ARG
    assert L.csim_netlist_counts(None, None, None, None, None, None) == capi.CSIM_ERR_ARG
    assert L.csim_netlist_num_probes(None) == 0
    assert L.csim_last_error() is not None


def test_product_does_not_touch_the_oracle():
    """Nothing under circuitsimulator_amd/ may import, link or execute oracle/."""
    pkg = os.path.join(ROOT, "circuitsimulator_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "mna_oracle" not in src and "liboracle" not in src, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
