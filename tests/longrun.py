"""A 2-day (72-step) adiabatic run from the reference's own rest state: the stand-in for BASELINE.json config 1 (the stock
T30 L8 2-day run of run.sh, which this image cannot build: NetCDF, gfortran and a boundary file are missing -- SURVEY.md s8c).

What the reference's main program does between its I/O (speedy.f90:24-54, time_stepping.f90:12-33) is: first_step -- a forward
half step, a first leapfrog step -- and then `call step(2, 2, 2*delt)` nsteps times.  With the reference's time_stepping.f90
compiled unchanged on tendencies.f90 minus its three physics lines (oracle/build_ref.sh) exactly that sequence runs here, from
the state initialize_from_rest_state builds (prognostics.f90:34-125, restated below): vor = div = 0; T = 216 K on the two
stratospheric levels and (288 K - gamma z_sfc) sigma^(R gamma) below; log surface pressure consistent with it over the
orography; a tropospheric humidity profile.  The only input the stock run reads from a file for this is the surface
geopotential phis0: a seeded smooth orography (peaks around 2 km) stands in for it.  tcorh is the reference's lapse-rate
correction of that orography (forcing.f90:73-81); qcorh needs the surface models and is a small multiple of it.
Two cases: "rest" -- exactly that state: the sigma-coordinate pressure-gradient error over the mountains spins up winds of
1-3 m/s and a gravity-wave adjustment (what the stock run does in its first days, minus the physics); and "wind" -- the same
with a seeded smooth vorticity field on every level (winds of 20-30 m/s, the magnitude of real jets), so that the advection
terms carry the evolution.  Both stay bounded for the 2 days (nothing but the horizontal diffusion damps them: no physics, no
surface drag); at 2.5 times the "wind" amplitude the adiabatic core blows up within a day.

Shared by the golden-vector generator (tests/golden/make_golden.py: ref_run72.npz from the flang-built reference), the CPU
test that chains the C oracle's call-by-call step against it and the GPU test that replays the captured device step."""
import numpy as np

import synth

DELT = 2400.0                   # params.f90:30 (nsteps = 36 per day)
NSTEPS = 72                     # 2 days
CHECKPOINTS = (1, 2, 4, 9, 18, 36, 72)    # leapfrog steps after first_step at which the prognostics are recorded
F32 = lambda x: float(np.float32(x))       # the reference's unsuffixed literals are float32
GRAV, RGAS = F32(9.81), F32(287.0)         # physical_constants.f90:19-22
GAMMA, HSCALE, HSHUM, REFRH1 = F32(6.0), F32(7.5), F32(2.5), F32(0.7)   # dynamical_constants.f90:12-15


CASES = {"rest": 0.0, "wind": 1.0e-5}       # amplitude of the seeded vorticity field (1/s)


def rest_state(ex, seed=4242, height=2000.0, wind=0.0):
    """ex: an executor with spec_to_grid, grid_to_spec, trunct, table("fsg") (the oracle).  Returns
    the dict tests/dynstep.py's steps take: vor, div, t, tr [2, kx, nx, mx]; ps [2, nx, mx]; phis, tcorh, qcorh [nx, mx]
    (complex128), both time levels equal (first_step overwrites the second)."""
    kx, nx, mx, trunc = ex.kx, ex.nx, ex.mx, ex.trunc
    l = np.arange(mx)[None, :] + np.arange(nx)[:, None]
    oro = synth.spectra(1, trunc, first=seed)[0] * (1.0 / (1.0 + l)) ** 0.5          # smooth: ~ (1 + l)^-1.5 spectrum
    oro[0, 0] = 0.0
    g = ex.spec_to_grid(oro, 1)
    phis0 = np.maximum(g, 0.0) * (GRAV * height / g.max())                            # land above a flat "sea"
    phis = ex.grid_to_spec(phis0)                                                     # prognostics.f90:52
    fsg = ex.table("fsg")
    gam1 = GAMMA / (F32(1000.0) * GRAV)
    tref, ttop = F32(288.0), F32(216.0)
    gam2, rgam = gam1 / tref, RGAS * gam1
    t = np.zeros((kx, nx, mx), np.complex128)
    surfs = -gam1 * phis
    t[0, 0, 0] = t[1, 0, 0] = np.sqrt(np.float32(2.0)) * ttop                         # (:77-78; sqrt(2.0) is float32)
    surfs[0, 0] = float(np.sqrt(np.float32(2.0))) * tref - gam1 * phis[0, 0]
    for k in range(2, kx):
        t[k] = surfs * fsg[k] ** rgam
    surfg = float(np.log(np.float32(1.013))) + (1.0 / rgam) * np.log(1.0 - gam2 * phis0)    # (:88-94)
    ps = ex.trunct(ex.grid_to_spec(surfg))
    qref, qexp = REFRH1 * F32(0.622) * F32(17.0), HSCALE / HSHUM
    surfq = ex.trunct(ex.grid_to_spec(qref * np.exp(qexp * surfg)))
    q = np.zeros((kx, nx, mx), np.complex128)
    for k in range(2, kx):
        q[k] = surfq * fsg[k] ** qexp
    z = np.zeros((2, kx, nx, mx), np.complex128)
    vor = z.copy()
    if wind:
        v = synth.spectra(kx, trunc, first=seed + 50) * (1.0 / (1.0 + l)) ** 0.5 * wind
        v[:, 0, 0] = 0.0
        vor = np.stack([v, v])
    tcorh = ex.grid_to_spec(gam1 * phis0)                                             # forcing.f90:73-81, gamlat = gamma/(1000 grav)
    return {"vor": vor, "div": z.copy(), "t": np.stack([t, t]), "tr": np.stack([q, q]), "ps": np.stack([ps, ps]),
            "phis": phis, "tcorh": tcorh, "qcorh": tcorh * (-2.0e-3)}


def run(step, tail_init, st, nsteps=NSTEPS, checkpoints=CHECKPOINTS, rob=None):
    """time_stepping.f90:12-33 + speedy.f90:33-38 with `step(j1, j2, dt, state) -> state` and `tail_init(dt)` supplied by the
    executor (the flang-built reference, the C oracle's call-by-call sequence, or a device step).  Returns {n: state after n
    leapfrog steps} for n in checkpoints."""
    tail_init(0.5 * DELT)
    st = step(1, 1, 0.5 * DELT, st)
    tail_init(DELT)
    st = step(1, 2, DELT, st)
    tail_init(2.0 * DELT)
    out = {}
    for n in range(1, nsteps + 1):
        st = step(2, 2, 2.0 * DELT, st)
        if n in checkpoints:
            out[n] = {k: np.array(st[k], copy=True) for k in ("vor", "div", "t", "tr", "ps")}
    return out


# what ref_run72.npz keeps of a checkpoint: ps whole, every second level of the 3-D fields at every second n and m
SUB = (slice(None), slice(None, None, 2), slice(None, None, 2), slice(None, None, 2))


def cut(name, a):
    return a if name == "ps" else a[SUB]
