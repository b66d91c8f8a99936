"""Multi-GPU decomposition emulated on ONE device: the strips of a 3-rank run are swept one after
the other by the HIP equilibrator (node_mask = owned nodes), their ghost rows travel through the
halo kernels of the C ABI (eqlb_halo_pack / eqlb_halo_unpack_add) instead of RCCL, and the owned
rows must equal the single-domain result.  (The RCCL transport itself is covered by the gloo
tests on CPU tensors and by the driver's multi-GPU bench.)"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 2, 3])
def test_three_strips_on_one_gpu(oracle_mod, k):
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd import distributed as dd
    from dolfinx_eqlb_amd.mesh import create_rectangle
    from synthetic import facet_types, make_compatible_data
    world, n = 3, 6
    nrt = k * (k + 2)
    dev = torch.device("cuda", 0)
    gmesh = create_rectangle(world * n, n, 0.0, float(world))
    gft = facet_types(gmesh)
    gG, gf = make_compatible_data(gmesh, k, gft, seed=5)
    gref = oracle_mod.se_reconstruct(gmesh, k, gft, gG[None], gf[None])[0].reshape(gmesh.ncells, nrt)

    parts, xs, gcells = [], [], []
    for rank in range(world):
        part = dd.StripPartition(n, rank, world)
        gi, gj, gt = part.grid_ids
        gcell = (gj * (world * n) + gi + rank * n) * 4 + gt
        m = part.mesh
        G = gG.reshape(gmesh.ncells, -1)[gcell].ravel()
        f = gf.reshape(gmesh.ncells, -1)[gcell].ravel()
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(m), k, 1)
        eq.set_boundary(part.facet_types(), node_mask=part.node_mask)
        x = torch.zeros(m.ncells * nrt, dtype=torch.float64, device=dev)
        dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)
        for _ in range(2):  # two accumulating sweeps
            eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        parts.append(part)
        xs.append(x)
        gcells.append(gcell)
    # reverse halo: rank r sends its ghost rows to rank r + 1
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world - 1):
        src, dst = parts[r], parts[r + 1]
        sidx = torch.from_numpy(src.send_cells).to(dev)
        ridx = torch.from_numpy(dst.recv_cells).to(dev)
        assert sidx.numel() == ridx.numel() == 3 * n
        buf = torch.empty(sidx.numel() * nrt, dtype=torch.float64, device=dev)
        cpp.halo_pack(xs[r].data_ptr(), sidx.data_ptr(), buf.data_ptr(), 1, sidx.numel(), nrt,
                      src.mesh.ncells, True, stream)
        cpp.halo_unpack_add(xs[r + 1].data_ptr(), ridx.data_ptr(), buf.data_ptr(), 1, ridx.numel(),
                            nrt, dst.mesh.ncells, stream)
    torch.cuda.synchronize()
    for r in range(world):
        part = parts[r]
        x = xs[r].cpu().numpy().reshape(part.mesh.ncells, nrt)
        got = x[part.cell_owned]
        ref = 2.0 * gref[gcells[r][part.cell_owned]]
        assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()
        if part.send_cells.size:
            assert np.all(x[part.send_cells] == 0.0)  # ghost rows cleared after packing


@pytest.mark.parametrize("ev", [False, True])
def test_device_calls_report_bad_patches_through_check_status(ev):
    """Device-memory calls are asynchronous: a patch system that is not positive definite (the
    matrix depends on the geometry only - here a cell of zero area) raises the device flag,
    eqlb_*_check_status reports and clears it."""
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd.mesh import create_rectangle
    from synthetic import facet_types, make_compatible_data
    k = 2
    dev = torch.device("cuda", 0)
    mesh = create_rectangle(6, 6)
    ft = facet_types(mesh)
    G, f = make_compatible_data(mesh, k, ft, seed=3)
    stream = torch.cuda.current_stream().cuda_stream
    dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)

    def sweep(m):
        eq = cpp.ConstrainedMinEquilibrator(cpp.DeviceMesh(m), k, 1) if ev \
            else cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(m), k, 1)
        eq.set_boundary(ft)
        nout = eq.ndofs if ev else m.ncells * k * (k + 2)
        x = torch.zeros(nout, dtype=torch.float64, device=dev)
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
        return eq, x

    eq, x = sweep(mesh)
    eq.check_status(stream)  # regular mesh: no complaint
    assert bool(torch.isfinite(x).all())
    # collapse one cell: its third vertex is moved onto the line through the other two
    import copy
    bad = copy.deepcopy(mesh)
    c = bad.ncells // 2
    v = bad.cell_nodes[c]
    bad.x[v[2], :2] = 0.5 * (bad.x[v[0], :2] + bad.x[v[1], :2])
    eq, x = sweep(bad)
    with pytest.raises(RuntimeError):
        eq.check_status(stream)
    eq.check_status(stream)  # the flag was cleared


def test_two_phase_sweep_with_halo_between(oracle_mod):
    """The multi-GPU step of bench.py on ONE device: tiles owning ghost cells first (priority
    tiles), the ghost rows packed and 'sent' while the remaining tiles are swept, unpack at the end -
    the result must equal the single-domain reference, and the priority tiles must own every
    ghost cell."""
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd import distributed as dd
    from dolfinx_eqlb_amd.mesh import create_rectangle
    from synthetic import facet_types, make_compatible_data
    world, n, k = 3, 24, 2
    nrt = k * (k + 2)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    gmesh = create_rectangle(world * n, n, 0.0, float(world))
    gft = facet_types(gmesh)
    gG, gf = make_compatible_data(gmesh, k, gft, seed=5)
    gref = oracle_mod.se_reconstruct(gmesh, k, gft, gG[None], gf[None])[0].reshape(gmesh.ncells, nrt)
    parts, xs, gcells, bufs = [], [], [], []
    for rank in range(world):
        part = dd.StripPartition(n, rank, world)
        gi, gj, gt = part.grid_ids
        gcell = (gj * (world * n) + gi + rank * n) * 4 + gt
        m = part.mesh
        G = gG.reshape(gmesh.ncells, -1)[gcell].ravel()
        f = gf.reshape(gmesh.ncells, -1)[gcell].ravel()
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(m), k, 1)
        eq.set_priority_cells(part.send_cells)
        eq.set_boundary(part.facet_types(), node_mask=part.node_mask)
        ti = eq.tiling_info()
        nprio = eq.num_priority_tiles
        assert ti["ntiles"] >= 4
        if part.send_cells.size:
            assert 0 < nprio < ti["ntiles"]
        else:
            assert nprio == 0
        x = torch.zeros(m.ncells * nrt, dtype=torch.float64, device=dev)
        dG, df = torch.from_numpy(G).to(dev), torch.from_numpy(f).to(dev)
        sidx = torch.from_numpy(part.send_cells).to(dev)
        buf = torch.zeros(max(sidx.numel(), 1) * nrt, dtype=torch.float64, device=dev)
        # phase 1: priority tiles, then the ghost rows leave (they must be final here)
        eq.set_option("tile_first", 0)
        eq.set_option("tile_count", nprio)
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), x.data_ptr(), stream)
        if sidx.numel():
            cpp.halo_pack(x.data_ptr(), sidx.data_ptr(), buf.data_ptr(), 1, sidx.numel(), nrt,
                          m.ncells, True, stream)
        # phase 2: the rest of the tiles
        eq.set_option("tile_first", 0)
        eq.set_option("tile_count", -1)
        eq.equilibrate_device_tiles(dG.data_ptr(), df.data_ptr(), x.data_ptr(), nprio, -1, stream)
        torch.cuda.synchronize()
        parts.append(part)
        xs.append(x)
        gcells.append(gcell)
        bufs.append(buf)
    for r in range(1, world):
        ridx = torch.from_numpy(parts[r].recv_cells).to(dev)
        cpp.halo_unpack_add(xs[r].data_ptr(), ridx.data_ptr(), bufs[r - 1].data_ptr(), 1, ridx.numel(),
                            nrt, parts[r].mesh.ncells, stream)
    torch.cuda.synchronize()
    for r in range(world):
        own = parts[r].cell_owned
        got = xs[r].cpu().numpy().reshape(-1, nrt)[own]
        ref = gref[gcells[r][own]]
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(gref).max()


@pytest.mark.parametrize("stress", [False, True])
def test_halo_exchange_class_on_device_with_side_stream_transport(oracle_mod, monkeypatch, stress):
    """HaloExchange.start / finish themselves (not the bare halo kernels) on CUDA tensors: three strips
    on one device, two accumulating two-phase steps.  RCCL is replaced by a stand-in with the stream
    semantics of the NCCL backend: batch_isend_irecv enqueues the copies on a SIDE stream that waits
    for the caller's current stream, wait() makes the current stream wait for the side stream.  That
    exercises what the gloo tests cannot: the ordering of the ctypes-launched pack / unpack kernels
    (raw stream handle) against the transport stream, the re-use of the cached P2POp list and of the
    persistent send / receive buffers across steps, and the clearing of the ghost rows.
    stress: two stress rows + weak symmetry in the two-phase sweep - the patches the fused stress kernel
    does not take (boundary patches: generic kernels) touch ghost cells as well and have to be complete
    before the rows are packed, i.e. run with the FIRST range of tiles."""
    import torch
    import torch.distributed as dist
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd import distributed as dd
    from dolfinx_eqlb_amd.mesh import create_rectangle
    from synthetic import facet_types, make_compatible_data
    world, n, k, nsteps = 3, 24, 2, 3
    nrt = k * (k + 2)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    mailbox = {}  # (src, dst) -> tensor in flight

    class P2POp:  # descriptors only (the real class needs an initialised process group)
        def __init__(self, op, tensor, peer, *a, **kw):
            self.op, self.tensor, self.peer = op, tensor, peer

    class Work:
        def wait(self):
            torch.cuda.current_stream().wait_stream(side)
            return True

    current_rank = [0]

    def batch_isend_irecv(ops):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for op in ops:
                if op.op is dist.isend:
                    mailbox[(current_rank[0], op.peer)] = op.tensor.clone()
                else:
                    op.tensor.copy_(mailbox.pop((op.peer, current_rank[0])))
        return [Work() for _ in ops]

    monkeypatch.setattr(dist, "P2POp", P2POp)
    monkeypatch.setattr(dist, "batch_isend_irecv", batch_isend_irecv)

    gmesh = create_rectangle(world * n, n, 0.0, float(world))
    gft = facet_types(gmesh)
    R = 2 if stress else 1
    if stress:
        from synthetic import make_compatible_stress_data
        gft = np.repeat(gft, 2, axis=0)
        gG, gf = make_compatible_stress_data(gmesh, k, gft)
        gref = oracle_mod.se_reconstruct(gmesh, k, gft, gG, gf, stress=True).reshape(R, gmesh.ncells, nrt)
    else:
        gG, gf = make_compatible_data(gmesh, k, gft, seed=5)
        gG, gf = gG[None], gf[None]
        gref = oracle_mod.se_reconstruct(gmesh, k, gft, gG, gf).reshape(R, gmesh.ncells, nrt)
    ranks = []
    for rank in range(world):
        part = dd.StripPartition(n, rank, world)
        gi, gj, gt = part.grid_ids
        gcell = (gj * (world * n) + gi + rank * n) * 4 + gt
        eq = cpp.SemiExplicitEquilibrator(cpp.DeviceMesh(part.mesh), k, R, reconstruct_stress=stress)
        eq.set_priority_cells(part.send_cells)
        eq.set_boundary(part.facet_types(R), node_mask=part.node_mask)
        ranks.append(dict(part=part, gcell=gcell, eq=eq, halo=dd.HaloExchange(part, nrt, dev, R),
                          G=torch.from_numpy(np.ascontiguousarray(gG.reshape(R, gmesh.ncells, -1)[:, gcell]).ravel()).to(dev),
                          f=torch.from_numpy(np.ascontiguousarray(gf.reshape(R, gmesh.ncells, -1)[:, gcell]).ravel()).to(dev),
                          x=torch.zeros(R * part.mesh.ncells * nrt, dtype=torch.float64, device=dev)))
    stream = torch.cuda.current_stream().cuda_stream
    for step in range(nsteps):
        for rank, r in enumerate(ranks):  # ascending: the send of rank - 1 is in flight when rank receives
            current_rank[0] = rank
            eq, x, nprio = r["eq"], r["x"], r["eq"].num_priority_tiles
            eq.equilibrate_device_tiles(r["G"].data_ptr(), r["f"].data_ptr(), x.data_ptr(), 0, nprio, stream)
            reqs = r["halo"].start(x)
            eq.equilibrate_device_tiles(r["G"].data_ptr(), r["f"].data_ptr(), x.data_ptr(), nprio, -1, stream)
            r["halo"].finish(x, reqs)
    torch.cuda.synchronize()
    assert not mailbox
    for r in ranks:
        part = r["part"]
        x = r["x"].cpu().numpy().reshape(R, part.mesh.ncells, nrt)
        ref = nsteps * gref[:, r["gcell"][part.cell_owned]]
        assert np.abs(x[:, part.cell_owned] - ref).max() <= 1e-10 * nsteps * np.abs(gref).max()
        if part.send_cells.size:
            assert np.all(x[:, part.send_cells] == 0.0)  # ghost rows cleared after packing


def test_rccl_transport_self_send_on_one_device():
    """The library's own transport (eqlb_halo_exchange / eqlb_halo_reduce: ncclGroupStart, ncclSend, ncclRecv,
    ncclGroupEnd on the caller's communicator and stream) with REAL RCCL calls on one device: a one-rank
    communicator made through eqlb_rccl_get_unique_id / eqlb_rccl_comm_create, the rank is its own peer - rows A
    are packed, cleared, sent to rank 0 and added onto rows B.  What it can check here: symbol resolution next to
    torch's bundled RCCL, argument marshalling of the pointer / count arrays, the 8-byte data type, stream order
    of pack -> send/recv -> unpack, re-use of the plan across steps.  What it cannot: more than one rank
    (RCCL refuses two ranks on one device; the multi-rank run is the driver's)."""
    import torch
    from dolfinx_eqlb_amd import cpp
    from dolfinx_eqlb_amd import distributed as dd
    dev = torch.device("cuda", 0)
    nrhs, nent, nrt = 2, 1000, 8
    rng = np.random.default_rng(4)
    perm = rng.permutation(nent)
    A, B = np.sort(perm[:137]), np.sort(perm[137:274])
    comm = cpp.RcclComm(cpp.RcclComm.unique_id(), 1, 0)
    try:
        import types
        part = types.SimpleNamespace(rank=0, send={0: A}, recv={0: B}, mesh=types.SimpleNamespace(ncells=nent))
        halo = dd.HaloExchange(part, nrt, dev, nrhs, comm=comm)
        x0 = rng.standard_normal((nrhs, nent, nrt))
        expect = x0.copy()
        for mode in ("reduce_rccl", "start_finish", "reduce_rccl"):
            x = torch.from_numpy(expect.ravel().copy()).to(dev)
            if mode == "reduce_rccl":
                halo.reduce_rccl(x)       # one library call
            else:
                halo.finish(x, halo.start(x))  # pack + grouped exchange, then unpack (two-phase sweep in between)
            torch.cuda.synchronize()
            nxt = expect.copy()
            nxt[:, B] += expect[:, A]
            nxt[:, A] = 0.0
            got = x.cpu().numpy().reshape(nrhs, nent, nrt)
            assert np.array_equal(got, nxt), mode
            expect = nxt
            expect[:, A] = rng.standard_normal((nrhs, A.size, nrt))  # new ghost rows for the next step
    finally:
        comm.destroy()


def test_rccl_transport_argument_checks():
    from dolfinx_eqlb_amd import cpp
    import ctypes as C
    with pytest.raises(RuntimeError, match="128 bytes"):
        cpp.RcclComm(b"short", 1, 0)
    rc = cpp.lib().eqlb_halo_exchange(None, C.c_int32(0), None, None, None, None, None, None)
    assert rc != 0 and b"eqlb_halo_exchange" in cpp.lib().eqlb_last_error()


def test_pybind_halo_exchange_self_send():
    """The same self-send through the pybind11 module (the C++ host's classes RcclComm / HaloExchange over
    eqlb_halo_create / eqlb_halo_reduce_plan): index lists as host arrays, buffers owned by the library."""
    import torch
    from dolfinx_eqlb_amd import _cpp
    dev = torch.device("cuda", 0)
    nrhs, nent, nrt = 2, 600, 15
    rng = np.random.default_rng(7)
    perm = rng.permutation(nent)
    A, B = np.sort(perm[:90]).astype(np.int64), np.sort(perm[90:180]).astype(np.int64)
    comm = _cpp.RcclComm(_cpp.rccl_unique_id(), 1, 0)
    halo = _cpp.HaloExchange(nrhs, nrt, nent, {0: A}, {0: B})
    assert halo.bytes() == (8 * nrhs * nrt * A.size, 8 * nrhs * nrt * B.size)
    x0 = rng.standard_normal((nrhs, nent, nrt))
    x = torch.from_numpy(x0.ravel().copy()).to(dev)
    _cpp.set_stream(torch.cuda.current_stream().cuda_stream)
    halo.reduce_ptr(comm, x.data_ptr())
    torch.cuda.synchronize()
    expect = x0.copy()
    expect[:, B] += x0[:, A]
    expect[:, A] = 0.0
    assert np.array_equal(x.cpu().numpy().reshape(nrhs, nent, nrt), expect)
    with pytest.raises(RuntimeError, match="out of range"):
        _cpp.HaloExchange(1, 3, 10, {0: np.array([11], dtype=np.int64)}, {})
