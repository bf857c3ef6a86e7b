#!/bin/bash
# CPU-only check: the device headers compiled for the host (tests/hostsim) under UndefinedBehaviorSanitizer,
# then the host-simulation test suites run against those builds.  (GPU sanitizers are not available on this pool.)
# usage: bash tools/ubsan_hostsim.sh        -- needs hipcc (host-only compile) and pytest
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
HS=$ROOT/tests/hostsim
OUT=$(mktemp -d /tmp/ubsan_hostsim.XXXX)
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-libgcc-file-name --rtlib=compiler-rt 2>/dev/null)")
FLAGS="-O1 -g -DVRF_GCOMB_BITS=8 --offload-host-only -fPIC -std=c++17 -Wno-unused-function -Wno-pass-failed -fsanitize=undefined -fno-sanitize-recover=undefined"
link() { /opt/rocm/bin/hipcc --offload-host-only -shared -fPIC -o "$1" "${@:2}" -Wl,--whole-archive "$RT/libclang_rt.ubsan_standalone-x86_64.a" "$RT/libclang_rt.ubsan_standalone_cxx-x86_64.a" -Wl,--no-whole-archive -lpthread -ldl; }
for f in hostsim_fe hostsim_verify hostsim_prove hostsim_bls hostsim_jj; do
  /opt/rocm/bin/hipcc $FLAGS -c "$HS/$f.hip" -o "$OUT/$f.o" 2>/dev/null
done
link "$OUT/libhostsim.so" "$OUT/hostsim_fe.o" "$OUT/hostsim_verify.o" "$OUT/hostsim_prove.o"
link "$OUT/libhostsim_bls.so" "$OUT/hostsim_bls.o"
link "$OUT/libhostsim_jj.so" "$OUT/hostsim_jj.o"
mkdir -p "$OUT/orig"
for l in libhostsim.so libhostsim_bls.so libhostsim_jj.so; do cp "$HS/$l" "$OUT/orig/$l"; cp "$OUT/$l" "$HS/$l"; done
restore() { for l in libhostsim.so libhostsim_bls.so libhostsim_jj.so; do cp "$OUT/orig/$l" "$HS/$l"; done; }
trap restore EXIT
cd "$ROOT"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_hostsim.py tests/test_bls_pairing.py tests/test_jubjub.py -x -q -m "not gpu"
