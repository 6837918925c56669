#!/bin/bash
# A/B of library builds on the GPU box: scripts/ab_lib.sh "<python script + args>" lib1.so lib2.so ...
CMD=$1; shift
for L in "$@"; do echo "=== $L"; VMPC_LIB_PATH=$L python $CMD 2>&1 | grep -v amdgpu.ids; done
