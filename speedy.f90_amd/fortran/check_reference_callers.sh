#!/usr/bin/env bash
# Build-container-only check (needs /root/reference + flang; not run on the GPU box, nothing is copied):
# the drop-in modules of this directory are compiled against the REFERENCE'S OWN types.f90 / params.f90 /
# physical_constants.f90 (read where they lie), and then
#   1. the reference callers that have no NetCDF dependency chain -- diagnostics.f90 (inverse_laplacian) and
#      sppt.f90 (el2, spec_to_grid) -- are compiled unchanged against the drop-in .mod files: names AND signatures;
#   2. for every other caller (tendencies, time_stepping, speedy, physics, input_output, prognostics, boundaries, forcing,
#      initialization: they cannot be compiled here because they pull in the netcdf module), each of their
#      `use <module>, only: ...` statements that names one of the replaced modules is extracted verbatim and compiled
#      in a probe unit: every name the reference imports must be public in the drop-in.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${SPEEDY_REFERENCE:-/root/reference}/source"
FC="${FC:-/opt/rocm/lib/llvm/bin/flang}"
if [ ! -d "$REF" ] || [ ! -x "$FC" ]; then echo "check_reference_callers: reference or flang absent -- skipped"; exit 0; fi
TMP="$(mktemp -d /tmp/spdy_callers.XXXXXX)"; trap 'rm -rf "$TMP"' EXIT
cd "$TMP"
"$FC" -c -w "$REF/types.f90" "$REF/params.f90" "$REF/physical_constants.f90" \
      "$HERE/spdy_c.f90" "$HERE/spectral.f90" "$HERE/horizontal_diffusion.f90" "$HERE/implicit.f90" "$HERE/geopotential.f90"
# time_stepping: the reference's dynamical_constants; its prognostics module pulls in the NetCDF chain, so the array
# declarations come from the stand-in (same names and shapes as prognostics.f90:16-24)
"$FC" -c -w -cpp "$REF/dynamical_constants.f90" "$HERE/support/host_prognostics.f90" "$HERE/time_stepping.f90"
echo "drop-ins compile against the reference's types/params"
"$FC" -c -w "$REF/diagnostics.f90" "$REF/sppt.f90"
echo "reference callers diagnostics.f90, sppt.f90 compile unchanged against the drop-in spectral.mod"
python3 - "$REF" > probes.f90 <<'PY'
import re, sys, glob, os
ref = sys.argv[1]
mods = ("spectral", "horizontal_diffusion", "implicit", "geopotential", "time_stepping")
n = 0
for path in sorted(glob.glob(os.path.join(ref, "*.f90"))):
    base = os.path.basename(path)
    if base in ("spectral.f90", "horizontal_diffusion.f90", "implicit.f90", "geopotential.f90", "time_stepping.f90"):
        continue
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"\s*use\s+(\w+)\s*,\s*only\s*:", lines[i], re.I)
        if m and m.group(1).lower() in mods:
            stmt, first = lines[i], i + 1
            while stmt.rstrip().endswith("&"):
                i += 1
                stmt = stmt.rstrip()[:-1] + " " + lines[i].strip().lstrip("&")
            n += 1
            print("subroutine probe_%d   ! %s:%d" % (n, base, first))
            print("    " + " ".join(stmt.split()))
            print("end subroutine")
        i += 1
sys.stderr.write("%d use-statements extracted\n" % n)
PY
"$FC" -c -w probes.f90
echo "every name the reference's callers import from spectral / horizontal_diffusion / implicit / geopotential / time_stepping is public in the drop-ins:"
grep -c "^subroutine" probes.f90
grep "use " probes.f90 | sort | uniq -c | sort -rn | head -30
