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
      dynamical_constants matrix_inversion horizontal_diffusion implicit geopotential"

build_one() {   # $1 = tag, $2 = sed program for params.f90 ('' = stock), $3 = optimisation flags (default -O2),
                # $4 = sed program for geometry.f90 ('' = stock; see the kx = 5 / 7 builds below)
    local tag="$1" prog="$2" opt="${3:--O2}" gprog="${4:-}" tmp srcs=() m
    tmp="$(mktemp -d /tmp/speedy_ref_${tag}.XXXXXX)"
    for m in $MODS; do
        if [ "$m" = params ] && [ -n "$prog" ]; then
            sed -e "$prog" "$REF/params.f90" > "$tmp/params.f90"
            srcs+=("$tmp/params.f90")
        elif [ "$m" = geometry ] && [ -n "$gprog" ]; then
            sed -e "$gprog" "$REF/geometry.f90" > "$tmp/geometry.f90"
            srcs+=("$tmp/geometry.f90")
        else
            srcs+=("$REF/$m.f90")
        fi
    done
    # step_field_2d / step_field_3d (time_stepping.f90:126-167) depend only on `types`, `params` and spectral%trunct, but
    # their module also holds step(), which drags prognostics -> boundaries -> input_output -> NetCDF.  The two functions are
    # cut out of the reference file AS THEY ARE (sed by their own first line / the module's last line) into a scratch module
    # with the same two `use` lines the reference module has (time_stepping.f90:2-3); nothing is stubbed or rewritten.
    {   echo "module step_field_ref"; echo "    use types, only: p"; echo "    use params"; echo "    implicit none"; echo "contains"
        sed -n '/^    function step_field_3d/,/^end module/p' "$REF/time_stepping.f90" | sed '$d'
        echo "end module"; } > "$tmp/step_field_ref.f90"
    grep -q "function step_field_2d" "$tmp/step_field_ref.f90" || { echo "build_ref: step_field extraction failed"; exit 1; }
    # get_spectral_tendencies (tendencies.f90:241-293) the same way: its module also holds get_grid_point_tendencies, which
    # drags physics -> ... -> NetCDF, and it reads the prognostic arrays of module `prognostics`, whose initialisation routines
    # drag boundaries / input_output.  Both are cut out of the reference files AS THEY ARE: the subroutine by its own first and
    # last lines, and the DECLARATION PART of prognostics.f90 (everything before `contains`: prognostics.f90:4-24) minus the
    # one `public initialize_prognostics` line, whose procedure stays behind.  No statement is written or changed here.
    {   sed -n '/^module prognostics/,/^contains/p' "$REF/prognostics.f90" | sed -e '$d' -e '/public initialize_prognostics/d'
        echo "end module"; } > "$tmp/prognostics_decl.f90"
    {   echo "module spectral_tendencies_ref"; echo "    use types, only: p"; echo "    use params"; echo "    implicit none"; echo "contains"
        sed -n '/^    subroutine get_spectral_tendencies/,/^    end subroutine/p' "$REF/tendencies.f90"
        echo "end module"; } > "$tmp/spectral_tendencies_ref.f90"
    grep -q "complex(p) :: phis(mx,nx)" "$tmp/prognostics_decl.f90" && grep -q "laplacian(phi(:,:,k) + rgas\*tref(k)\*ps(:,:,j2))" "$tmp/spectral_tendencies_ref.f90" \
        || { echo "build_ref: get_spectral_tendencies extraction failed"; exit 1; }
    # The reference's whole time step, ADIABATIC: with the declaration part of prognostics.f90 in place, tendencies.f90 needs
    # nothing that is missing here except the column physics (physics -> ... -> NetCDF).  The scratch copy of tendencies.f90
    # has exactly the three physics lines deleted -- the `use physics` line and the two-line `call get_physical_tendencies`
    # (tendencies.f90:55, :205-206) -- i.e. it is the reference's get_tendencies with T_phy = 0; time_stepping.f90 (first_step,
    # step, and the diffusion block inside it) then compiles FROM THE REFERENCE FILE, unchanged.
    sed -e '/use physics, only: get_physical_tendencies/d' \
        -e '/call get_physical_tendencies(/,/utend, vtend, ttend, trtend)/d' "$REF/tendencies.f90" > "$tmp/tendencies.f90"
    if grep -q "get_physical_tendencies" "$tmp/tendencies.f90" || ! grep -q "phi = get_geopotential(t(:,:,:,j1), phis)" "$tmp/tendencies.f90" \
       || [ "$(wc -l < "$REF/tendencies.f90")" -ne "$(( $(wc -l < "$tmp/tendencies.f90") + 3 ))" ]; then
        echo "build_ref: adiabatic tendencies.f90 is not the reference minus its three physics lines"; exit 1
    fi
    # The gridded snapshot of input_output.f90 (subroutine output, :183-206: uvspec + five inverse transforms per level, then
    # the float32 conversions with q*1.0e-3, phi/grav, p0*exp(ps)).  The subroutine itself cannot compile here (every other
    # statement of it is a NetCDF call), so its COMPUTING lines are cut out of the reference file AS THEY ARE, in file order:
    # the two `use` lines it computes with, its declarations from `vor` to `integer :: k, ncid` (:101-116: the six spectral
    # arguments, the work arrays, the float32 arrays -- which become dummy arguments of the scratch subroutine simply by being
    # named in its header line), the transform loop (:183-192) and the conversion block (:199-205).  Written here: the module
    # frame (the reference module's own `use types` / `use params` / `implicit none` lines) and the subroutine header/end.
    {   echo "module output_fields_ref"; echo "    use types, only: p, sp"; echo "    use params"; echo "    implicit none"; echo "contains"
        echo "    subroutine output_fields(vor, div, t, ps, tr, phi, u_out, v_out, t_out, q_out, phi_out, ps_out)"
        sed -n '/^    subroutine output(/,/^    end subroutine/p' "$REF/input_output.f90" | sed -n \
            -e '/use physical_constants, only: p0, grav/p' \
            -e '/use spectral, only: spec_to_grid, uvspec/p' \
            -e '/complex(p), intent(in) :: vor(mx,nx,kx,2)/,/integer :: k, ncid/p' \
            -e '/! Convert prognostic fields from spectral space to grid point space/,/ps_grid = spec_to_grid(ps(:,:,1), 1)/p' \
            -e '/! Preprocess output variables/,/ps_out = real(p0\*exp(ps_grid), sp)/p'
        echo "    end subroutine"; echo "end module"; } > "$tmp/output_fields_ref.f90"
    if [ "$(grep -c . "$tmp/output_fields_ref.f90")" -ne 41 ] || ! grep -q "q_out = real(q_grid\*1.0e-3, sp)" "$tmp/output_fields_ref.f90" \
       || ! grep -q "phi_grid(:,:,k) = spec_to_grid(phi(:,:,k), 1)" "$tmp/output_fields_ref.f90" || grep -q "nf90" "$tmp/output_fields_ref.f90"; then
        echo "build_ref: output-field extraction of input_output.f90 failed"; exit 1
    fi
    ( cd "$tmp" && "$FC" $opt -fPIC -shared -w -Wl,-Bsymbolic -o "$OUT/libspeedy_ref_${tag}.so" \
          "${srcs[@]}" "$tmp/step_field_ref.f90" "$tmp/prognostics_decl.f90" "$tmp/spectral_tendencies_ref.f90" \
          "$tmp/tendencies.f90" "$REF/time_stepping.f90" "$tmp/output_fields_ref.f90" "$HERE/ref_shim.f90" )
    rm -rf "$tmp"
    echo "build_ref: built $OUT/libspeedy_ref_${tag}.so"
}

newer() { [ -f "$1" ] && [ "$1" -nt "$HERE/ref_shim.f90" ] && [ "$1" -nt "$HERE/build_ref.sh" ]; }

T63='s/trunc = 30 /trunc = 63 /; s/ix = 96 /ix = 192/; s/iy = 24 /iy = 48 /'
newer "$OUT/libspeedy_ref_t30.so" || build_one t30 ''
newer "$OUT/libspeedy_ref_t63.so" || build_one t63 "$T63"
# the reference's other level counts (geometry.f90:42-48) -- and 16 levels, for which it has no sigma set: the
# test supplies the half levels through the reference's public geometry variables (ref_shim.f90: ref_set_sigma)
# kx = 5 / 7: flang (unlike gfortran, which only warns) rejects geometry.f90 there, because the statically dead
# branches for the OTHER level counts assign hsg(:8) / hsg(:9) to an array of kx+1 elements (geometry.f90:45,47).
# The scratch copy has exactly those dead assignment lines deleted; the live branch is untouched.
newer "$OUT/libspeedy_ref_t30k5.so" || build_one t30k5 's/kx = 8 /kx = 5 /' -O2 '/hsg(:8) = /d; /hsg(:9) = /d'
newer "$OUT/libspeedy_ref_t30k7.so" || build_one t30k7 's/kx = 8 /kx = 7 /' -O2 '/hsg(:9) = /d'
newer "$OUT/libspeedy_ref_t63k16.so" || build_one t63k16 "$T63; s/kx = 8 /kx = 16/"
# CPU-baseline variant mirroring upstream's -Ofast (gfortran.makefile:18): value-unsafe optimisation allowed, AVX2+FMA
# (x86-64-v3 rather than -march=native: the .so is built here and timed on the GPU box's host CPU)
newer "$OUT/libspeedy_ref_t30fast.so" || build_one t30fast '' '-O3 -ffast-math -march=x86-64-v3'
newer "$OUT/libspeedy_ref_t63fast.so" || build_one t63fast "$T63" '-O3 -ffast-math -march=x86-64-v3'
