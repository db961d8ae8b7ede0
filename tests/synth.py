"""Synthetic inputs shared by tests, golden-vector generation, smoke() and bench.py.

Counter-based splitmix64 generator so the same fields can be reproduced in
any language: field b uses seed 20240229 + b.
  * band-limited spectra  s(m,n) = (xi1 + i*xi2)/(1+l),  xi ~ U(-1,1),  l = m'+n-1,
    zero outside the active triangle l <= trunc+1, Im(m'=0) = 0   (model-shaped input)
  * white grids           g ~ U(-0.5, 0.5), seed 42 + b           (not band-limited)
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n uniform doubles in [0,1) from splitmix64 stream `seed` (vectorised, counter-based)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def spectra(nb, trunc, first=0, full_rows=False):
    """[nb, nx, mx] complex128 band-limited spectra: l <= trunc populated (a `trunct`-ed
    prognostic field); full_rows=True also populates l = trunc+1 (incl. row n=nx at m'=0),
    which uvspec output does and the inverse transform reads."""
    mx, nx = trunc + 1, trunc + 2
    m = np.arange(mx)[None, :]
    n = np.arange(nx)[:, None]
    l = m + n
    active = (l <= trunc + 1) if full_rows else (l <= trunc)
    out = np.zeros((nb, nx, mx), np.complex128)
    for b in range(nb):
        u = splitmix64(20240229 + first + b, 2 * nx * mx).reshape(nx, mx, 2) * 2.0 - 1.0
        s = (u[..., 0] + 1j * u[..., 1]) / (1.0 + l)
        s = np.where(active, s, 0.0)
        s[:, 0] = s[:, 0].real
        out[b] = s
    return out


def grids(nb, ix, il, first=0):
    """[nb, il, ix] float64 white-noise grids U(-0.5,0.5)."""
    out = np.empty((nb, il, ix))
    for b in range(nb):
        out[b] = splitmix64(42 + first + b, il * ix).reshape(il, ix) - 0.5
    return out


def relerr(x, ref):
    """max|x-ref| / max|ref|  -- the tolerance norm of SURVEY.md s8c (1e-12 bar)."""
    x = np.asarray(x); ref = np.asarray(ref)
    d = np.max(np.abs(x - ref)) if x.size else 0.0
    s = np.max(np.abs(ref)) if ref.size else 0.0
    return float(d / s) if s > 0 else float(d)


# A 16-level half-sigma set for the T63 L16 configuration (BASELINE.json config 5).  The reference defines sigma
# levels for kx = 5, 7, 8 only (geometry.f90:42-48); this is its 8-level set with every layer split in two, written
# like the reference's literals (float32 values widened to double).
SIGMA_L16 = np.array([0.000, 0.025, 0.050, 0.095, 0.140, 0.200, 0.260, 0.340, 0.420, 0.510, 0.600, 0.685, 0.770, 0.835,
                      0.900, 0.950, 1.000], np.float32).astype(np.float64)


# 20 half levels for the kx > 16 code paths (level rows looped inside a block; the one-launch spectral step falls back to
# its five kernels): no reference build exists for it -- the oracle is the restatement that is pinned at 5, 7, 8 and 16 levels.
SIGMA_L20 = np.concatenate([[0.0], np.cumsum(np.array([0.02, 0.025, 0.03, 0.04, 0.045, 0.05, 0.055, 0.06, 0.065, 0.07, 0.07, 0.07,
                                                       0.065, 0.06, 0.055, 0.05, 0.045, 0.04, 0.035, 0.05]))]).astype(np.float32).astype(np.float64)
SIGMA_L20[-1] = 1.0
SIGMA_SETS = {"t63k16": SIGMA_L16, "t30k20": SIGMA_L20}


def tail_inputs(kx, nx, mx, seed=99):
    """Seeded (divdt, tdt, psdt)-shaped complex inputs: [kx,nx,mx] x 1e-6, [kx,nx,mx] x 1e-3, [nx,mx] x 1e-5."""
    u = splitmix64(seed, 2 * (2 * kx + 1) * nx * mx).reshape(2 * kx + 1, nx, mx, 2) * 2 - 1
    z = u[..., 0] + 1j * u[..., 1]
    return z[:kx] * 1e-6, z[kx:2 * kx] * 1e-3, z[2 * kx] * 1e-5


def cfield(shape, seed, scale=1.0):
    """Seeded complex array of the given shape, U(-1,1) + i U(-1,1), times scale."""
    n = int(np.prod(shape))
    u = splitmix64(seed, 2 * n).reshape(tuple(shape) + (2,)) * 2 - 1
    return (u[..., 0] + 1j * u[..., 1]) * scale
