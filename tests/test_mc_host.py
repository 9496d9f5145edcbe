"""Monte-Carlo parameter generator, host mirror (the device kernel must equal it bitwise:
tests/test_gpu_parity.py).  Distribution frozen in csrc/engine/mc_draw.h / SURVEY.md 8d #3."""
import numpy as np
from scipy import stats


def test_instance_zero_is_nominal(dbmixer_nl):
    t = dbmixer_nl.mc_params_host(12345, 0.05, 0, 8)
    assert t.shape == (79, 8)
    assert np.array_equal(t[:, 0], dbmixer_nl.nominal_params)
    assert not np.array_equal(t[:, 1], dbmixer_nl.nominal_params)


def test_only_r_c_l_vt_mu_are_perturbed(dbmixer_nl):
    nl = dbmixer_nl
    t = nl.mc_params_host(12345, 0.05, 0, 64)
    fixed = nl.mc_kinds == 0
    assert np.array_equal(t[fixed], np.repeat(nl.nominal_params[fixed, None], 64, axis=1))
    scaled = nl.mc_kinds == 1
    z = (t[scaled] / nl.nominal_params[scaled, None] - 1.0) / 0.05
    assert np.abs(z).max() <= 3.0 + 1e-12 and np.abs(z[:, 1:]).min() > 0
    # MOS K follows MU: K'/K = 1 + sigma z with its own draw
    k = nl.mc_kinds == 2
    zk = (t[k] / nl.nominal_params[k, None] - 1.0) / 0.05
    assert np.abs(zk).max() <= 3.0 + 1e-9


def test_shards_regenerate_the_same_instances(dbmixer_nl):
    nl = dbmixer_nl
    whole = nl.mc_params_host(7, 0.05, 0, 100)
    a = nl.mc_params_host(7, 0.05, 0, 37)
    b = nl.mc_params_host(7, 0.05, 37, 63)
    assert np.array_equal(whole, np.concatenate([a, b], axis=1))
    assert not np.array_equal(whole, nl.mc_params_host(8, 0.05, 0, 100))


def test_distribution_is_clipped_standard_normal(buffer_nl):
    nl = buffer_nl
    t = nl.mc_params_host(2024, 0.05, 1, 20000)
    p = int(np.where(nl.mc_kinds == 1)[0][0])
    z = (t[p] / nl.nominal_params[p] - 1.0) / 0.05
    assert abs(z.mean()) < 0.03 and abs(z.std() - 0.9865) < 0.02     # std of N(0,1) clipped at 3
    assert z.max() <= 3.0 + 1e-12 and z.min() >= -3.0 - 1e-12
    # inverse-CDF accuracy of the rational approximation (Acklam, < 1.2e-9 relative)
    ks = stats.kstest(z[np.abs(z) < 2.999], "norm")
    assert ks.statistic < 0.02
