"""The oracle's adiabatic time step -- the reference's own call sequence restated with the C oracle's pieces -- and the seeded
model state it runs on.  Shared by the GPU step tests, the Fortran drop-in test and the golden-vector generator.  Pinned bit for
bit to the flang-built reference (time_stepping.f90 unchanged on tendencies.f90 minus its three physics lines):
tests/test_oracle_golden.py::test_dynamics_step_pinned."""
import numpy as np

import synth

ROB, WIL = float(np.float32(0.05)), float(np.float32(0.53))          # params.f90:32-33 (float32 literals widened)
SDRAG = 1.0 / (float(np.float32(24.0 * 30.0)) * 3600.0)              # time_stepping.f90:77, dynamical_constants.f90:22



def state(sp, seed):
    """Band-limited prognostics [2, kx, nx, mx] (both time levels) and a few (nx, mx) fields."""
    kx, nx, mx = sp.kx, sp.nx, sp.mx

    def prog(first, scale):
        return (synth.spectra(2 * kx, sp.trunc, first=first) * scale).reshape(2, kx, nx, mx)
    st = {"vor": prog(seed, 1e-4), "div": prog(seed + 100, 1e-5), "t": prog(seed + 200, 30.0), "tr": prog(seed + 300, 1e-2)}
    st["t"][:, :, 0, 0] += 250.0 * np.sqrt(2.0)
    st["ps"] = (synth.spectra(2, sp.trunc, first=seed + 400) * 0.05).reshape(2, nx, mx)
    st["phis"] = synth.spectra(1, sp.trunc, first=seed + 500)[0] * 2000.0
    st["tcorh"] = synth.spectra(1, sp.trunc, first=seed + 600)[0] * 5.0
    st["qcorh"] = synth.spectra(1, sp.trunc, first=seed + 700)[0] * 1e-3
    return st


def oracle_dynamics_step(o, st, j1, dt, eps, j2=2, physics=None, before_diffusion=False):
    """One adiabatic time step of the dynamical core on the host, the reference's own call sequence (tendencies.f90:11-41,
    time_stepping.f90:35-118 without get_physical_tendencies): inverse transforms of time level j2, grid-space
    tendencies, direct transforms, spectral tendencies, implicit correction, diffusion, leapfrog/RAW."""
    kx, j2 = o.kx, j2 - 1
    ug, vg = [], []
    for k in range(kx):
        u, v = o.uvspec(st["vor"][j2, k], st["div"][j2, k])
        ug.append(o.spec_to_grid(u, 2)); vg.append(o.spec_to_grid(v, 2))
    ug, vg = np.stack(ug), np.stack(vg)
    vorg, divg, tg, trg = (np.stack([o.spec_to_grid(st[n][j2, k], 1) for k in range(kx)]) for n in ("vor", "div", "t", "tr"))
    dx, dy = o.grad(st["ps"][j2])
    px, py = o.spec_to_grid(dx, 2), o.spec_to_grid(dy, 2)
    U, V, PL = o.grid_tendencies(ug, vg, tg, vorg, divg, trg, px, py)
    P = 3 * kx
    if physics is not None:          # tendencies.f90:203-206: the physics adds to utend, vtend, ttend, trtend in grid space
        U, V, PL = np.array(U, copy=True), np.array(V, copy=True), np.array(PL, copy=True)
        physics(o, st, U[:kx], V[:kx], PL[kx:2 * kx], PL[2 * kx:3 * kx])
    vd = [o.vdspec(U[i], V[i], 2) for i in range(P)]
    pvor, pdiv = np.stack([x[0] for x in vd]), np.stack([x[1] for x in vd])
    pspec = np.stack([o.grid_to_spec(PL[i]) for i in range(P + 1)])
    pdiv, pspec = o.tendency_combine(pdiv, pspec)
    vordt, divdt, tdt, trdt, psdt = pvor[:kx], pdiv[:kx], pdiv[kx:2 * kx], pdiv[2 * kx:], pspec[P]
    divdt, tdt, psdt, phi = o.spectral_tendencies(st["div"][0], st["t"][0], st["ps"][0], st["phis"], divdt, tdt, psdt)
    divdt, tdt, psdt = o.implicit_terms(divdt, tdt, psdt)
    pre = {}
    if before_diffusion:       # what get_tendencies returns (tendencies.f90:11-41)
        pre = {"pre_vordt": np.array(vordt), "pre_divdt": np.array(divdt), "pre_tdt": np.array(tdt), "pre_psdt": np.array(psdt),
               "pre_trdt": np.array(trdt)}
    vordt, divdt, tdt, trdt = o.hdiff_step(st["vor"][0], st["div"][0], st["t"][0], st["tr"][0], st["tcorh"], st["qcorh"], SDRAG,
                                           vordt, divdt, tdt, trdt)
    new, fin = dict(st), {}
    # (step_field_* truncates its tendency argument in place, time_stepping.f90:146: what the step leaves behind is trunct(fdt))
    new["ps"], fin["psdt"] = o.step_field(j1, dt, eps, WIL, st["ps"], psdt)
    for n, d in (("vor", vordt), ("div", divdt), ("t", tdt), ("tr", trdt)):
        new[n], fin[n + "dt"] = o.step_field(j1, dt, eps, WIL, st[n], d)
    return new, dict({"U": U, "V": V, "PL": PL, "phi": phi}, **fin, **pre)


def wave_relerr(x, ref):
    """max|x - ref| / max|ref| with the global mean -- coefficient (n, m) = (0, 0) of every level -- removed from both:
    for t the mean is 250*sqrt(2) against waves of O(30/(1+l)), so the plain norm is carried by the mean."""
    x, ref = np.array(x, copy=True), np.array(ref, copy=True)
    x[..., 0, 0] = 0.0
    ref[..., 0, 0] = 0.0
    return synth.relerr(x, ref)
