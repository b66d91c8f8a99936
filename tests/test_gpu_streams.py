"""Device-memory calls on a caller's (non-blocking) stream: everything a call enqueues - fills of internal buffers,
the side stream of a fused stress launch, slot reductions - has to be ordered against THAT stream, not against the
null stream.  The inputs of every call are produced late on the caller's stream (behind a spin kernel; before that
the buffers hold NaN), the FIRST call on a fresh handle included (it is the one that allocates and fills the internal
buffers); the result has to equal the host-memory call of a second handle."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    return [("se2", dict(k=2, nrhs=1)), ("se2_slots", dict(k=2, nrhs=1, scatter=0)),
            ("se2_r3", dict(k=2, nrhs=3)), ("se3", dict(k=3, nrhs=1)), ("se4", dict(k=4, nrhs=1)),
            ("stress2", dict(k=2, nrhs=2, stress=True)), ("stress2_slots", dict(k=2, nrhs=2, stress=True, scatter=0)),
            ("stress3", dict(k=3, nrhs=2, stress=True)), ("ev2", dict(k=2, nrhs=1, ev=True)),
            ("ev3", dict(k=3, nrhs=1, ev=True))]


@pytest.mark.parametrize("name,cfg", _cases(), ids=[c[0] for c in _cases()])
def test_calls_on_a_user_stream_are_ordered(name, cfg):
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.mesh import create_unit_square
    from synthetic import facet_types, make_compatible_data, make_compatible_stress_data
    k, R = cfg["k"], cfg["nrhs"]
    mesh = create_unit_square(24, shuffle_seed=2, perturb=0.15)
    ft1 = facet_types(mesh, lambda x: x[:, 0] < 0.3)
    if cfg.get("stress"):
        ft = np.repeat(facet_types(mesh, None), 2, axis=0)
        G, f = make_compatible_stress_data(mesh, k, ft)
    else:
        ft = np.repeat(ft1, R, axis=0)
        Gs, fs = zip(*[make_compatible_data(mesh, k, ft1, seed=10 + r) for r in range(R)])
        G, f = np.stack(Gs), np.stack(fs)
    dm = cpp.DeviceMesh(mesh)

    def handle():
        if cfg.get("ev"):
            h = cpp.ConstrainedMinEquilibrator(dm, k, R)
        else:
            h = cpp.SemiExplicitEquilibrator(dm, k, R, reconstruct_stress=bool(cfg.get("stress")))
        if "scatter" in cfg:
            h.set_option("scatter", cfg["scatter"])
        h.set_boundary(ft)
        return h

    ref = handle().equilibrate_host(G, f)
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(device=dev)
    g_src = torch.from_numpy(G).to(dev)
    f_src = torch.from_numpy(f).to(dev)
    g_dev = torch.full_like(g_src, float("nan"))
    f_dev = torch.full_like(f_src, float("nan"))
    x_dev = torch.full((R, ref.shape[1]), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    h = handle()
    for call in range(2):  # the first call allocates and fills the internal buffers
        with torch.cuda.stream(s):
            torch.cuda._sleep(40_000_000)  # ~ 15-20 ms: the inputs below exist only after it
            g_dev.copy_(g_src)
            f_dev.copy_(f_src)
            x_dev.zero_()
            h.equilibrate_device(g_dev.data_ptr(), f_dev.data_ptr(), x_dev.data_ptr(), stream=s.cuda_stream)
            out = x_dev.clone()
            g_dev.fill_(float("nan"))  # and are gone right behind the call
            f_dev.fill_(float("nan"))
            x_dev.fill_(float("nan"))
        h.check_status(s.cuda_stream)
        s.synchronize()
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), (name, call)
        # (same kernels in the same order; the generic weak-symmetry kernel assembles with LDS atomics, so not bit-equal)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (name, call, np.abs(got - ref).max())
