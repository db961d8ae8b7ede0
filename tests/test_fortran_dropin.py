"""The Fortran host side: the drop-in modules `spectral`, `horizontal_diffusion`, `implicit`, `geopotential`
(speedy.f90_amd/fortran/) that keep the reference's module names, public names and signatures over the C ABI.
A flang-built driver calls them exactly as the model would; results are compared with the oracle at 1e-12."""
import os
import subprocess

import numpy as np
import pytest

import synth
from conftest import ROOT, TOL
from synth import tail_inputs

FDIR = os.path.join(ROOT, "speedy.f90_amd", "fortran")


def driver(tag):
    return os.path.join(FDIR, "build", tag, "dropin_driver")


def test_fortran_sources_keep_reference_api():
    """Same module name and public names as the reference's spectral module (spectral.f90:1-11)."""
    src = open(os.path.join(FDIR, "spectral.f90")).read().lower()
    assert "module spectral" in src
    for name in ("el2", "initialize_spectral", "laplacian", "inverse_laplacian", "spec_to_grid", "grid_to_spec",
                 "grad", "vds", "uvspec", "vdspec", "trunct"):
        assert name in src.split("contains")[0], name
    assert "function spec_to_grid(vorm, kcos) result(vorg)" in src
    assert "function grid_to_spec(vorg) result(vorm)" in src
    assert "bind(c" in open(os.path.join(FDIR, "spdy_c.f90")).read().lower()
    # the tail modules: reference module names and full public sets (horizontal_diffusion.f90:10-11, implicit.f90:10-11,
    # geopotential.f90:11)
    for mod, names in (("horizontal_diffusion", ("initialize_horizontal_diffusion", "do_horizontal_diffusion", "dmp", "dmpd", "dmps",
                                                 "dmp1", "dmp1d", "dmp1s", "tcorv", "qcorv", "tcorh", "qcorh")),
                       ("implicit", ("initialize_implicit", "implicit_terms", "tref", "tref2", "tref3")),
                       ("geopotential", ("initialize_geopotential", "get_geopotential"))):
        src = open(os.path.join(FDIR, mod + ".f90")).read().lower()
        assert "module " + mod in src
        head = src.split("contains")[0]
        pub = " ".join(l for l in head.split("\n") if l.strip().startswith("public"))
        for name in names:
            assert name in pub, (mod, name)


def test_reference_callers_resolve_against_dropins():
    """Build container only: compile the drop-ins against the reference's own types/params and check its callers'
    `use ..., only:` lists (and two whole caller files) against them (fortran/check_reference_callers.sh)."""
    if not os.path.isdir("/root/reference/source") or not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        pytest.skip("needs /root/reference and flang (build container)")
    r = subprocess.run([os.path.join(FDIR, "check_reference_callers.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compile unchanged against the drop-in" in r.stdout


@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_driver_fails_loudly_without_gpu(tag):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    if not os.path.exists(driver(tag)):
        pytest.skip("Fortran driver not built (no flang)")
    r = subprocess.run([driver(tag), "/dev/null", "/dev/null"], capture_output=True, text=True)
    assert r.returncode != 0                          # no device -> error stop, never a silent CPU path


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_dropin_module_vs_oracle(tag, tmp_path, oracle_factory):
    if not os.path.exists(driver(tag)):
        pytest.skip("Fortran driver not built (no flang on this box and no prebuilt binary)")
    o = oracle_factory(tag)
    nx, mx, il, ix, kx = o.nx, o.mx, o.il, o.ix, o.kx
    S = synth.spectra(2, o.trunc, first=40, full_rows=True)
    G = synth.grids(2, ix, il, first=40)
    sk, tk, ps = tail_inputs(kx, nx, mx)
    o.tail_init(4800.0)
    dmp, dmp1 = o.table("dmp").reshape(nx, mx), o.table("dmp1").reshape(nx, mx)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for a in (S, G, sk, tk, ps, dmp, dmp1):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([driver(tag), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, np.float64)
    pos = [0]

    def take(shape, cplx):
        n = int(np.prod(shape)) * (2 if cplx else 1)
        a = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return a.view(np.complex128).reshape(shape) if cplx else a.reshape(shape)

    def ok(x, ref):
        assert synth.relerr(x, ref) <= TOL

    sp, gr = (nx, mx), (il, ix)
    assert np.array_equal(take(sp, False), o.table("el2").reshape(sp))
    ok(take(gr, False), o.spec_to_grid(S[0], 1))
    ok(take(gr, False), o.spec_to_grid(S[1], 2))
    ok(take(sp, True), o.grid_to_spec(G[0]))
    ok(take(sp, True), o.laplacian(S[0]))
    ok(take(sp, True), o.inverse_laplacian(S[0]))
    ok(take(sp, True), o.trunct(S[0]))
    rdx, rdy = o.grad(S[0]); ok(take(sp, True), rdx); ok(take(sp, True), rdy)
    ru, rv = o.uvspec(S[0], S[1]); ok(take(sp, True), ru); ok(take(sp, True), rv)
    a, b = o.vds(S[0], S[1]); ok(take(sp, True), a); ok(take(sp, True), b)
    a, b = o.vdspec(G[0], G[1], 2); ok(take(sp, True), a); ok(take(sp, True), b)
    kc = [1 + (k % 2) for k in range(1, kx + 1)]
    glev = take((kx,) + gr, False)
    for k in range(kx):
        ok(glev[k], o.spec_to_grid(sk[k], kc[k]))
    slev = take((kx,) + sp, True)
    for k in range(kx):
        ok(slev[k], o.grid_to_spec(glev[k]))
    # module tables filled by initialize_horizontal_diffusion / initialize_implicit (public in the reference)
    for name in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s"):
        assert np.array_equal(take(sp, False), o.table(name).reshape(sp)), name
    for name in ("tcorv", "qcorv", "tref", "tref2", "tref3"):
        assert np.array_equal(take((kx,), False), o.table(name)), name
    ok(take((kx,) + sp, True), o.hdiff(tk, sk, dmp, dmp1))
    ok(take(sp, True), o.hdiff(ps, sk[0], o.table("dmps").reshape(sp), o.table("dmp1s").reshape(sp)))
    ok(take((kx,) + sp, True), o.geopotential(tk, ps))
    rd, rt, rp = o.implicit_terms(sk, tk, ps)
    ok(take((kx,) + sp, True), rd); ok(take((kx,) + sp, True), rt); ok(take(sp, True), rp)
    # level-stack sequences in one call each; inputs: the *updated* sk, tk written back by implicit_terms above
    ug, vg = take((kx,) + gr, False), take((kx,) + gr, False)
    for k in range(kx):
        ru, rv = o.uvspec(rd[k], rt[k])
        ok(ug[k], o.spec_to_grid(ru, 2)); ok(vg[k], o.spec_to_grid(rv, 2))
    gx, gy = take((kx,) + gr, False), take((kx,) + gr, False)
    for k in range(kx):
        rdx, rdy = o.grad(rd[k])
        ok(gx[k], o.spec_to_grid(rdx, 2)); ok(gy[k], o.spec_to_grid(rdy, 2))
    vorl, divl = take((kx,) + sp, True), take((kx,) + sp, True)
    for k in range(kx):
        a, b = o.vdspec(gx[k], gy[k], 2)
        ok(vorl[k], a); ok(divl[k], b)
    assert pos[0] == raw.size
