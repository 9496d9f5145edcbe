"""Synthetic workloads named by BASELINE.json / SURVEY.md 8(d)."""


def rc_ladder_netlist(n_nodes=256, r=100.0, c=1e-12, tstep=1e-9, tstop=100e-9):
    """BASELINE configs[3]: linear RC ladder, n_nodes node equations + 1 source branch
    (n_nodes = 256 -> 257 unknowns, 511 elements).  V1 n1 0 SIN 0 1 1e6 0; Rk nk n(k+1);
    Ck nk 0 for k >= 2."""
    lines = ["* synthetic RC ladder, %d nodes" % n_nodes, "V1 n1 0 SIN 0 1 1e6 0"]
    lines += ["R%d n%d n%d %.17g" % (k, k, k + 1, r) for k in range(1, n_nodes)]
    lines += ["C%d n%d 0 %.17g" % (k, k, c) for k in range(2, n_nodes + 1)]
    lines.append(".TRAN %.17g %.17g" % (tstep, tstop))
    return "\n".join(lines) + "\n"
