#!/bin/bash
# dev helper (GPU box): rebuild with different defaults of one GeneratorOptions member and time the 16-lane kernel
#   tools/dev/opt_sweep.sh <outdir> <member> <value>...      e.g.  opt_sweep.sh gw groupWavesPerEu 0 2
OUT=gpurun_out/${1:-opt}; mkdir -p $OUT; M=$2; shift 2
H=circuitsimulator_amd/csrc/engine/codegen.hpp
cp $H /tmp/codegen.hpp.orig
for v in "$@"; do
  sed "s/int $M = [-0-9]*;/int $M = $v;/" /tmp/codegen.hpp.orig > $H
  make -s -C circuitsimulator_amd/csrc > $OUT/make_$v.log 2>&1 || { echo "build failed: $v"; tail -3 $OUT/make_$v.log; continue; }
  echo "== $M $v"
  tools/dev/sweep.sh $(basename $OUT) 16:4096:6:2 16:8192:3:1
done
cp /tmp/codegen.hpp.orig $H
