#!/bin/bash
# Ablation builds of libelprep_hip.so into build_ab/ (git-ignored, travels with gpurun): one library per "<file>:<n>" argument, the
# file compiled with -DELP_ABL=<n>, everything else as built.  usage: build_abl.sh apply3:1 apply3:2 bqsr:4 ...   (run make first)
# The `#if ELP_ABL == n` blocks of a session are temporary: the ones of round 3 (apply3 without look-ups / loads / stores, the prologue
# without site walk / record stores / CIGAR loads; results in DESIGN.md 4.6b) were taken out again - put new ones in for a new ablation.
set -e
cd "$(dirname "$0")/../../elprep_amd/csrc"
mkdir -p ../../build_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off"
for v in "$@"; do
  f=${v%%:*}; n=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS -DELP_ABL=$n -c $f.hip -o ../../build_ab/${f}_abl$n.o
  objs=""; for o in *.o; do [ "$o" = "$f.o" ] && objs="$objs ../../build_ab/${f}_abl$n.o" || objs="$objs $o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/lib_${f}_abl$n.so $objs -ldl
  echo built build_ab/lib_${f}_abl$n.so
done
