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
# eight long compilations (minutes each under the sanitizer): all at once
for f in hostsim_fe hostsim_verify hostsim_prove hostsim_bls hostsim_jj; do
  /opt/rocm/bin/hipcc $FLAGS -c "$HS/$f.hip" -o "$OUT/$f.o" 2>/dev/null &
done
# the builds of the other base fields: Ed25519 (f1), Baby-JubJub (f2), secp256r1 (P-256)
/opt/rocm/bin/hipcc $FLAGS -DVRF_FIELD=1 -c "$HS/hostsim_suite.hip" -o "$OUT/hostsim_suite_f1.o" 2>/dev/null &
/opt/rocm/bin/hipcc $FLAGS -DVRF_FIELD=2 -c "$HS/hostsim_suite.hip" -o "$OUT/hostsim_suite_f2.o" 2>/dev/null &
/opt/rocm/bin/hipcc $FLAGS -DVRF_FIELD=3 -c "$HS/hostsim_p256.hip" -o "$OUT/hostsim_p256.o" 2>/dev/null &
/opt/rocm/bin/hipcc $FLAGS -DVRF_FIELD=0 -c "$HS/hostsim_bsw.hip" -o "$OUT/hostsim_bsw.o" 2>/dev/null &      # bandersnatch_sw codec
wait
link "$OUT/libhostsim.so" "$OUT/hostsim_fe.o" "$OUT/hostsim_verify.o" "$OUT/hostsim_prove.o"
link "$OUT/libhostsim_bls.so" "$OUT/hostsim_bls.o"
link "$OUT/libhostsim_jj.so" "$OUT/hostsim_jj.o"
link "$OUT/libhostsim_f1.so" "$OUT/hostsim_suite_f1.o"
link "$OUT/libhostsim_f2.so" "$OUT/hostsim_suite_f2.o"
link "$OUT/libhostsim_p256.so" "$OUT/hostsim_p256.o"
link "$OUT/libhostsim_bsw.so" "$OUT/hostsim_bsw.o"
LIBS="libhostsim.so libhostsim_bls.so libhostsim_jj.so libhostsim_f1.so libhostsim_f2.so libhostsim_p256.so libhostsim_bsw.so"
# the test fixtures run `make` on the library they load: bring the regular build up to date first, so that make finds
# nothing to do and leaves the sanitizer builds in place
make -C "$HS" -j8 all > /dev/null
mkdir -p "$OUT/orig"
for l in $LIBS; do cp -p "$HS/$l" "$OUT/orig/$l"; cp "$OUT/$l" "$HS/$l"; done
restore() { for l in $LIBS; do cp -p "$OUT/orig/$l" "$HS/$l"; touch "$HS/$l"; done; }
trap restore EXIT
cd "$ROOT"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_hostsim.py tests/test_bls_pairing.py tests/test_jubjub.py tests/test_new_suites.py tests/test_secp256r1.py tests/test_bandersnatch_sw.py -x -q -m "not gpu"
