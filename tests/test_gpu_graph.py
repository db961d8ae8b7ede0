"""HIP-graph replay of device-resident calls (include/spdy.h: spdy_graph_*): the replayed sequence must
give exactly what the same calls give when issued one by one, and a capture must refuse what it cannot hold."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res", ["t30", "t63"])
def test_graph_replay_equals_eager(res):
    import torch
    import speedy_f90_amd as s
    nb = 91 if res == "t30" else 20                      # the per-step inverse batch of the T30 L8 model
    sp = s.Spectral(res, kx=8, max_batch=nb, device=0)   # keeps its own (non-default) stream
    dev = torch.device("cuda", 0)
    S = torch.from_numpy(synth.spectra(nb, sp.trunc, first=0)).to(dev)
    grid_e, spec_e = torch.zeros((nb, sp.il, sp.ix), dtype=torch.float64, device=dev), torch.zeros_like(S)
    grid_g, spec_g = torch.zeros_like(grid_e), torch.zeros_like(S)
    u_e, v_e, u_g, v_g = (torch.zeros_like(S[:8]) for _ in range(4))
    torch.cuda.synchronize()
    # eager
    sp.spec_to_grid_dev(S, grid_e, kcos=1)
    sp.grid_to_spec_dev(grid_e, spec_e)
    sp.uvspec_dev(spec_e[:8], spec_e[8:16], u_e, v_e)
    sp.synchronize()
    # the same three calls, captured once and replayed twice
    with sp.graph_capture() as g:
        sp.spec_to_grid_dev(S, grid_g, kcos=1)
        sp.grid_to_spec_dev(grid_g, spec_g)
        sp.uvspec_dev(spec_g[:8], spec_g[8:16], u_g, v_g)
    assert float(grid_g.abs().max()) == 0.0              # nothing ran during the capture
    for _ in range(2):
        g.launch()
    sp.synchronize()
    for a, b in ((grid_e, grid_g), (spec_e, spec_g), (u_e, u_g), (v_e, v_g)):
        assert torch.equal(a, b)
    g.close()
    sp.close()


def test_capture_refuses_host_paths_and_recovers():
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral("t30", kx=8, max_batch=8, device=0)
    S = synth.spectra(1, sp.trunc, first=3)[0]
    want = sp.spec_to_grid(S, 1)
    with pytest.raises(s.SpdyError):
        with sp.graph_capture():
            sp.spec_to_grid(S, 1)                          # host pointers: H2D/D2H copies and a sync
    got = sp.spec_to_grid(S, 1)                            # the plan is usable again
    assert np.array_equal(want, got)
    with pytest.raises(s.SpdyError):                       # end without begin
        h = __import__("ctypes").c_void_p()
        s.check(sp.lib.spdy_graph_end(sp.h, __import__("ctypes").byref(h)))
    sp.close()
