#!/bin/bash
# The host-side readers (hic_reader.cpp with mst_inflate.h, text_reader.cpp) under AddressSanitizer + UBSan: the CPU reader tests
# -- round trips of every block encoding, the two independent readings, 300 random streams with flipped bits and truncations
# through the own inflate, the streamed read -- run against an instrumented build of libmustache_io.so.  No report = clean.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/libmustache_io_asan.so
cd "$REPO/mustache_amd/csrc"
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -o "$OUT" hic_reader.cpp text_reader.cpp -lz
cd "$REPO"
# libstdc++ has to be loaded before the sanitizer runtime resolves __cxa_throw (python itself does not link it)
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 \
    MUSTACHE_IO_LIB="$OUT" python -m pytest tests/test_hic_reader.py tests/test_hic_two_readings.py tests/test_text_reader.py \
    tests/test_readers_ref.py "tests/test_host_logic.py::test_fill_like_reference_equals_numpy_row_slices" -x -q -s -m "not gpu" 2>&1 | tee "${TMPDIR:-/tmp}/asan_reader_check.log" | tail -3
if grep -q "runtime error\|AddressSanitizer" "${TMPDIR:-/tmp}/asan_reader_check.log"; then
    echo "SANITIZER REPORTS:"; grep "runtime error\|AddressSanitizer" "${TMPDIR:-/tmp}/asan_reader_check.log" | sort | uniq -c
    exit 1
fi
echo "sanitizers: no report"
