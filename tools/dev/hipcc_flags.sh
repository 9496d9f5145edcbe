#!/bin/bash
# scratch: hipcc with extra flags from $CSIM_EXTRA_HIPCC_FLAGS (A/B of compiler options on JIT-generated kernels: CSIM_HIPCC=<this file>)
exec /opt/rocm/bin/hipcc $CSIM_EXTRA_HIPCC_FLAGS "$@"
