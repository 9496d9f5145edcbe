"""CPU oracle (test infrastructure only; see oracle/mna_oracle.h)."""
