#!/bin/bash
# dev helper (GPU box): rebuild with different defaults of GeneratorOptions::stageAhead and time the 16-lane kernel
OUT=gpurun_out/${1:-ahead}; mkdir -p $OUT; shift
H=circuitsimulator_amd/csrc/engine/codegen.hpp
cp $H /tmp/codegen.hpp.orig
for v in "$@"; do
  sed "s/int stageAhead = [-0-9]*;/int stageAhead = $v;/" /tmp/codegen.hpp.orig > $H
  make -s -C circuitsimulator_amd/csrc > $OUT/make_$v.log 2>&1 || { echo "build failed: $v"; tail -3 $OUT/make_$v.log; continue; }
  echo "== stage_ahead $v"
  tools/dev/sweep.sh $(basename $OUT) 16:4096:6:2
done
cp /tmp/codegen.hpp.orig $H
