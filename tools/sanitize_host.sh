#!/bin/bash
# Sanitizer runs of everything that executes on the HOST (GPU AddressSanitizer is not available on the pool: CPU builds only).
#   1. the CPU oracle (test infrastructure) under ASan + UBSan: field, NTT, LDE, both Merkle backends, whole proofs, the verifier
#      on valid, tampered and truncated proofs;
#   2. the product library's host code under ASan (its device code is compiled as usual, -fno-gpu-sanitize): Cairo front-end,
#      verifier with both Merkle backends on valid / tampered / truncated / random inputs, host Poseidon, helpers.  No GPU needed.
# usage: tools/sanitize_host.sh   (from the repo root; builds into $SP_SANITIZE_DIR, default /tmp/sp_sanitize)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
D=${SP_SANITIZE_DIR:-/tmp/sp_sanitize}
export SP_SANITIZE_DIR=$D
mkdir -p $D/obj
g++ -O1 -g -march=x86-64-v3 -std=c++17 -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $D/liboracle_stark252.so $R/oracle/capi.cpp
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 $R/tools/sanitize_oracle.py
make -s -j8 -C $R/lambdaworks_cairo_prover_amd/csrc
cp $R/lambdaworks_cairo_prover_amd/csrc/*.o $D/obj/
for f in capi_host cairo_host cairo_air_host verifier; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c $R/lambdaworks_cairo_prover_amd/csrc/$f.cpp -o $D/obj/$f.o
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o $D/libstark252_hip.so $D/obj/*.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT python3 $R/tools/sanitize_product_host.py
