#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the *real* reference hot path (Fortran) into
# oracle/_ref/libspeedy_ref_{t30,t63}.so with AMD flang, from the sources where they lie
# under /root/reference/source, plus our own bind(C) shim (oracle/ref_shim.f90).
#
#  * No reference source is copied into the repo.  Objects/.mod files and the one patched
#    file needed for T63 (params.f90 with trunc/ix/iy edited by sed -- the reference fixes
#    its resolution at compile time, params.f90:19-26) live in a mktemp scratch dir that is
#    deleted on exit.  Only the two .so files land in oracle/_ref/ (git-ignored).
#  * flang -O2, no -march, no fast-math: x86-64 baseline has no FMA and flang does not
#    reassociate, so results are the IEEE left-to-right evaluation of the source.  This is
#    the definition of "the reference's output" used by tests/golden.
#  * The flang runtime is linked statically, so the .so only needs libm/libc on the GPU box.
#
# A missing /root/reference (GPU box) or missing flang is not an error: prebuilt files are
# used if present, otherwise tests that need _ref skip.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${SPEEDY_REFERENCE:-/root/reference}/source"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
OUT="$HERE/_ref"

if [ ! -d "$REF" ] || [ ! -x "$FC" ]; then
    echo "build_ref: reference sources or flang not present; keeping prebuilt oracle/_ref (if any)"
    exit 0
fi
mkdir -p "$OUT"
MODS="types params physical_constants geometry fftpack fourier legendre spectral \
      dynamical_constants matrix_inversion horizontal_diffusion implicit"

build_one() {   # $1 = tag, $2 = sed program for params.f90 ('' = stock)
    local tag="$1" prog="$2" tmp srcs=() m
    tmp="$(mktemp -d /tmp/speedy_ref_${tag}.XXXXXX)"
    for m in $MODS; do
        if [ "$m" = params ] && [ -n "$prog" ]; then
            sed -e "$prog" "$REF/params.f90" > "$tmp/params.f90"
            srcs+=("$tmp/params.f90")
        else
            srcs+=("$REF/$m.f90")
        fi
    done
    ( cd "$tmp" && "$FC" -O2 -fPIC -shared -w -Wl,-Bsymbolic -o "$OUT/libspeedy_ref_${tag}.so" \
          "${srcs[@]}" "$HERE/ref_shim.f90" )
    rm -rf "$tmp"
    echo "build_ref: built $OUT/libspeedy_ref_${tag}.so"
}

newer() { [ -f "$1" ] && [ "$1" -nt "$HERE/ref_shim.f90" ] && [ "$1" -nt "$HERE/build_ref.sh" ]; }

newer "$OUT/libspeedy_ref_t30.so" || build_one t30 ''
newer "$OUT/libspeedy_ref_t63.so" || build_one t63 \
    's/trunc = 30 /trunc = 63 /; s/ix = 96 /ix = 192/; s/iy = 24 /iy = 48 /'
