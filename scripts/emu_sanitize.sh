#!/bin/bash
# Sanitizer lane of the GPU-less harness (test infrastructure; nothing here ships): the UNMODIFIED kernels, compiled by
# tests/emu/build_emu.py with AddressSanitizer (or UBSan), under the emulator test suites.  A kernel that reads or writes past a
# caller's buffer, past its LDS arrays (the emulator's `static` stand-ins carry redzones) or past the ctx's scratch is reported
# with file:line of the access -- the tests' numpy / torch buffers come from the intercepted malloc.
# Third mode, `reverse` (or `alternate`): no instrumentation, but the emulator's scheduler visits the fibers of a block -- and the
# resident blocks of a cooperative grid -- in descending order between synchronisation points (SLK_EMU_ORDER, tests/emu/
# emu_runtime.cpp).  Every order is a legal execution of a correctly synchronised kernel: a missing __syncthreads() / SLK_WAVE_SYNC
# whose reader happens to run after its writer in ascending thread order fails here (checked by knocking out the barrier of
# k_rng_finalize: the key block comes back incomplete).
#   usage: scripts/emu_sanitize.sh [asan|ubsan|reverse|alternate] [pytest arguments; default: the emulator, host, sharded and bench-launch suites]
# The checker itself was put through the same: `make -C oracle -B _build/libslk_oracle.so CFLAGS="... -fsanitize=address,undefined"`
# under the oracle-using suites (392 tests, no report), then rebuilt plain -- not part of this script, which never touches oracle/_build.
# The first run compiles the instrumented library into its own tests/emu/_build_* directory (ASan: ~8 minutes).
set -u
MODE=${1:-asan}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
GCCLIB=$(dirname "$(g++ -print-file-name=libasan.so)")
if [ "$MODE" = asan ]; then
  export SLK_EMU_CXXFLAGS="-fsanitize=address -fno-omit-frame-pointer"
  # libstdc++ beside libasan: python does not link it, and ASan's __cxa_throw interceptor must find the real one before torch throws
  export LD_PRELOAD="$GCCLIB/libasan.so $(g++ -print-file-name=libstdc++.so.6)"
  export ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0}
elif [ "$MODE" = reverse ] || [ "$MODE" = alternate ]; then
  export SLK_EMU_ORDER=$MODE
else
  export SLK_EMU_CXXFLAGS="-fsanitize=undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
  export LD_PRELOAD="$GCCLIB/libubsan.so"
  export UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1}
fi
if [ $# -eq 0 ]; then
  set -- tests/test_emu_engine.py tests/test_emu_epoch.py tests/test_emu_seq.py tests/test_emu_sort.py tests/test_emu_bloom.py \
         tests/test_emu_explicit.py tests/test_emu_bench_parity.py tests/test_host_model.py tests/test_host_seq_model.py \
         tests/test_host_explicit_model.py tests/test_host_encoders.py tests/test_host_api.py tests/test_sharded.py tests/test_bench_cli.py
fi
LD_PRELOAD= python tests/emu/build_emu.py > /dev/null || exit 1   # (the compiler itself runs unsanitized)
exec python -m pytest "$@" -q -p no:cacheprovider -m "not gpu"
