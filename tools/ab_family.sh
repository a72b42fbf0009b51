#!/bin/bash
# A/B of prebuilt libraries on several input families (one box):  tools/ab_family.sh "<fam,fam>" build_ab/a.so build_ab/b.so ...
fams=$1; shift
for lib in "$@"; do
  echo "== $lib"
  MPC_AB_LIB=$lib python tools/realistic_probe.py --families $fams --steps 10 2>&1 | grep -v amdgpu | grep -v "^{" | cut -c1-200
done
