"""CPU: pin the C restatement (oracle/speedy_oracle.c) to the reference.

1. against the committed golden vectors (reference outputs produced by the flang build of
   /root/reference/source, tests/golden/make_golden.py) -- runs everywhere;
2. against the live reference library oracle/_ref (when it has been built) on extra seeds.
The restatement is expected to be bit-exact; the assertion bar is 1e-15 relative.
"""
import numpy as np
import pytest

import synth
from golden.make_golden import tail_inputs

EXACT = 1e-15
TAGS = ("t30", "t63")


def close(x, ref, tol=EXACT):
    assert x.shape == ref.shape
    assert synth.relerr(x, ref) <= tol


@pytest.mark.parametrize("tag", TAGS)
def test_tables(tag, golden, oracle_factory):
    g, o = golden(tag), oracle_factory(tag)
    assert list(g["dims"]) == [o.trunc, o.ix, o.iy, o.il, o.kx, o.nx, o.mx]
    for name in ("sia_half", "cosgr", "cosgr2", "hsg", "dhs", "fsg", "dhsr", "fsgr", "work"):
        assert np.array_equal(o.table(name), g["tab_" + name]), name
    assert np.array_equal(o.table("coa_half")[: o.iy], g["tab_coa_half"])
    assert np.array_equal(o.table("ifac")[:6].astype(int), g["tab_ifac"][:6])
    assert np.array_equal(o.table("epsi"), g["tab_epsi"].ravel())
    assert np.array_equal(o.table("el2"), g["tab_el2"].ravel())


def test_known_answers_appendix_f(oracle_factory):
    """SURVEY.md Appendix F known-answer values (flang -O2 oracle run)."""
    o = oracle_factory("t30")
    sia = o.table("sia_half")
    bits = sia.astype(np.float32).view(np.uint32)
    assert np.array_equal(sia, sia.astype(np.float32).astype(np.float64))      # exactly float32 values
    assert bits[0] == 0x3F7FB2AE and bits[11] == 0x3F395CD2 and bits[23] == 0x3D04A2C4
    assert o.table("work")[0] == 0.997858923119484320 and o.table("work")[1] == 0.0654031310475514244
    assert list(o.table("ifac")[:6].astype(int)) == [96, 4, 2, 4, 4, 3]
    assert abs(o.table("wt").sum() - 1.0) < 2e-15
    s = o.grid_to_spec(np.ones((o.il, o.ix)))
    assert s[0, 0].real == 1.41421358031670752
    o63 = oracle_factory("t63")
    assert list(o63.table("ifac")[:6].astype(int)) == [192, 4, 4, 4, 4, 3]
    assert o63.table("sia_half").astype(np.float32).view(np.uint32)[47] == 0x3C855767
    nsh2 = o.table("nsh2").astype(int)
    assert nsh2[0] == 62 and nsh2[-1] == 2 and nsh2.sum() == 1054


@pytest.mark.parametrize("tag", TAGS)
def test_transform_stages(tag, golden, oracle_factory):
    g, o = golden(tag), oracle_factory(tag)
    S, G = g["S"], g["G"]
    for b in range(g["leginv"].shape[0]):
        close(o.legendre_inv(S[b]), g["leginv"][b])
        close(o.fourier_inv(g["leginv"][b], 1), g["finv1"][b])
        close(o.fourier_inv(g["leginv"][b], 2), g["finv2"][b])
        close(o.fourier_dir(G[b]), g["fdir"][b])
        close(o.legendre_dir(g["fdir"][b]), g["legdir"][b])
        close(o.spec_to_grid(S[b], 1), g["s2g1"][b])
        close(o.spec_to_grid(S[b], 2), g["s2g2"][b])
        close(o.grid_to_spec(G[b]), g["g2s"][b])
        # structural facts the reference exhibits
        assert np.all(g["g2s"][b][-1] == 0)                      # row nx stays 0
        assert np.array_equal(g["finv2"][b], g["finv1"][b] * o.table("cosgr")[:, None])
    close(o.grid_to_spec(np.ones((o.il, o.ix))), g["ones_g2s"])


@pytest.mark.parametrize("tag", TAGS)
def test_spectral_operators(tag, golden, oracle_factory):
    g, o = golden(tag), oracle_factory(tag)
    S, G = g["S"], g["G"]
    close(o.laplacian(S[0]), g["lap"])
    close(o.inverse_laplacian(S[0]), g["invlap"])
    close(o.trunct(S[0]), g["trunct"])
    dx, dy = o.grad(S[0]); close(dx, g["grad_dx"]); close(dy, g["grad_dy"])
    a, b = o.vds(S[0], S[1]); close(a, g["vds_vor"]); close(b, g["vds_div"])
    a, b = o.uvspec(S[0], S[1]); close(a, g["uv_u"]); close(b, g["uv_v"])
    a, b = o.vdspec(G[0], G[1], 2); close(a, g["vdspec2_vor"]); close(b, g["vdspec2_div"])
    a, b = o.vdspec(G[0], G[1], 1); close(a, g["vdspec1_vor"]); close(b, g["vdspec1_div"])


@pytest.mark.parametrize("tag", TAGS)
def test_fftpack_vectors(tag, golden, oracle_factory):
    import ctypes
    g, o = golden(tag), oracle_factory(tag)
    work = np.ascontiguousarray(g["tab_work"]); ifac = np.ascontiguousarray(g["tab_ifac"].astype(np.int32))
    for key, fn in (("fft_b", o.lib.orc_rfftb1), ("fft_f", o.lib.orc_rfftf1)):
        c = g["fft_in"].copy(); ch = np.zeros_like(c)
        rc = fn(ctypes.c_int(c.size), c.ctypes.data_as(ctypes.c_void_p), ch.ctypes.data_as(ctypes.c_void_p),
                work.ctypes.data_as(ctypes.c_void_p), ifac.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        close(c, g[key])


@pytest.mark.parametrize("tag", TAGS)
def test_tail(tag, golden, oracle_factory):
    g, o = golden(tag), oracle_factory(tag)
    if "imp_div" in g.files:
        div, t, ps = g["imp_div"], g["imp_t"], g["imp_ps"]
        assert np.array_equal(div, tail_inputs(o.kx, o.nx, o.mx)[0])
    else:
        div, t, ps = tail_inputs(o.kx, o.nx, o.mx)
    for dt in g["dts"]:
        key = "dt%d_" % int(dt)
        o.tail_init(float(dt))
        for name in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s", "tref", "tref2", "tref3"):
            close(o.table(name), g[key + name].ravel())
        if dt in g["imp_dts"]:
            a, b, c = o.implicit_terms(div, t, ps)
            close(a, g[key + "imp_div_out"]); close(b, g[key + "imp_t_out"]); close(c, g[key + "imp_ps_out"])
        if key + "hdiff2d" in g.files:
            close(o.hdiff(ps, 2 * ps, g[key + "dmps"], g[key + "dmp1s"]), g[key + "hdiff2d"])
        if key + "hdiff3d" in g.files:
            close(o.hdiff(t, div, g[key + "dmp"], g[key + "dmp1"]), g[key + "hdiff3d"])


@pytest.mark.parametrize("tag", TAGS)
def test_against_live_reference(tag, oracle_factory):
    """Extra seeds against oracle/_ref when it exists (build container; prebuilt on the GPU box)."""
    from oracle.pyoracle import Reference
    if not Reference.available(tag):
        pytest.skip("oracle/_ref not built")
    r, o = Reference(tag), oracle_factory(tag)
    S = synth.spectra(3, o.trunc, first=100, full_rows=True)
    G = synth.grids(3, o.ix, o.il, first=100)
    for b in range(3):
        for kcos in (1, 2):
            close(o.spec_to_grid(S[b], kcos), r.spec_to_grid(S[b], kcos))
        close(o.grid_to_spec(G[b]), r.grid_to_spec(G[b]))
    # round-trip accuracy of the reference itself is only ~5e-5 (first-guess Gaussian latitudes)
    s2 = o.grid_to_spec(o.spec_to_grid(synth.spectra(1, o.trunc)[0], 1))
    err = np.max(np.abs(s2 - synth.spectra(1, o.trunc)[0])[:-1])
    assert 1e-6 < err < 5e-3
    out = o.roundtrip_loop(G[:2]); ref = r.roundtrip_loop(G[:2])
    close(out, ref)
