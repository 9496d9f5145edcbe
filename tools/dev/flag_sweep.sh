#!/bin/bash
# dev helper (GPU box): rebuild the generated libraries with extra hipcc flags and time the 16-lane kernel
OUT=gpurun_out/${1:-flags}; mkdir -p $OUT; shift
for flags in "$@"; do
  touch circuitsimulator_amd/csrc/schedules/MANIFEST
  make -s -C circuitsimulator_amd/csrc SCHED_HIPFLAGS="$flags" > $OUT/make.log 2>&1 || { echo "build failed: $flags"; tail -3 $OUT/make.log; continue; }
  echo "== flags: $flags"
  tools/dev/sweep.sh $(basename $OUT) 16:4096:6:2 1:65536:2:1
done
touch circuitsimulator_amd/csrc/schedules/MANIFEST; make -s -C circuitsimulator_amd/csrc > /dev/null 2>&1
