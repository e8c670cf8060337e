#!/bin/bash
# Same-box A / B of builds of libvita_hip.so that differ in the NT GEMM main loop (r06).  Usage: tools/ab_gemm_libs.sh out.txt lib1.so lib2.so ...
# Runs tools/ab_gemm.py for every library, three rounds interleaved (A B C A B C ...), so that box drift shows up as spread inside a library's rows.
out=$1; shift
: > "$out"
for round in 1 2 3; do
  for lib in "$@"; do
    VITA_HIP_LIB=$lib python tools/ab_gemm.py 2>&1 | tail -1 >> "$out"
  done
done
cat "$out"
