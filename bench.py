#!/usr/bin/env python3
"""Headline benchmark: equilibrated patches/s (fp64) of the semi-explicit flux equilibration
(FluxEqlbSE, RT_2) on a 1M-triangle Poisson case per GPU (BASELINE.json configs[1]).

A step = one eqlb_se_equilibrate call = the reference's timed region, one
`equilibrate_fluxes()` (python/test/performance/perftest.py:145-147), over all patches of the
mesh, with the inputs (projected flux, projected RHS) and the output resident in HBM and the
patch SoA / reference tensors cached in the handle ("warm").  The "cold" figure
(handle creation + patch construction + one call on host arrays, what the reference pays inside
its timed region, se/reconstruction.hpp:275-313) is reported beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n 500] [--k 2] [--stress | --ev]

--gpus N > 1 without a torch.distributed environment: this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything
touches the GPU) and relays its output and exit code.  Under torch.distributed.run (one rank per
GPU): the unit-square strips of the ranks form one [0,N]x[0,1] domain; patches are partitioned by
node ownership and the partial sums of the ghost-cell RT DOFs are exchanged with the strip
neighbours (RCCL send/recv) inside the timed step - weak scaling, 1M triangles per GPU.

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` (dominant kernel,
HIP-event time measured live over the timed region) and `cpu_baseline` (the CPU restatement of
the reference algorithm, oracle/, timed on this box's host cores).
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]  # tools/synthetic.py: the synthetic workload

import numpy as np  # noqa: E402

VALU_NS_PER_INSTRUCTION = 2.0  # measured, see roofline.valu_floor_ms below
HEADLINE_METRIC = "equilibrated patches/s (fp64) on 1M-tri Poisson k=2; L2 flux-divergence residual"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=500, help="squares per side and GPU (500 -> 1M triangles)")
    ap.add_argument("--k", type=int, default=2, help="RT degree")
    ap.add_argument("--solver", type=int, default=None)
    ap.add_argument("--scatter", type=int, default=None,
                    help="0 slots + reduction, 1 atomics, 2 tiled (default where available; for --stress the "
                         "fused launch of both rows and the weak-symmetry step)")
    ap.add_argument("--fused", type=int, default=1, help="all patch-size bins in one launch")
    ap.add_argument("--accumulate", type=int, default=1,
                    help="1 (default, reference semantics): flux_hdiv += result; 0: store")
    ap.add_argument("--stress", action="store_true",
                    help="two rows + weak symmetry (BASELINE configs[3]); not the headline")
    ap.add_argument("--ev", action="store_true",
                    help="constrained-minimisation equilibrator (FluxEqlbEV, configs[2]); not the headline")
    ap.add_argument("--tile-cells", type=int, default=0, help="cells per tile of the tiled launch (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shuffle", type=int, default=None, help="seed for random local vertex order")
    ap.add_argument("--halo", choices=["rccl", "torch"], default="rccl",
                    help="N > 1: transport of the reverse halo - rccl: the library's own grouped ncclSend / ncclRecv "
                         "(eqlb_halo_exchange) on a communicator made through the C ABI; torch: torch.distributed "
                         "isend / irecv (nccl backend = RCCL).  rccl falls back to torch if the communicator "
                         "cannot be made on every rank")
    ap.add_argument("--nrhs", type=int, default=1,
                    help="right-hand sides equilibrated by one call (plain fluxes; the reference's multi-RHS test "
                         "uses 4, test_fluxeqlb_multirhs.py:24-186); not the headline")
    ap.add_argument("--settle", type=int, default=1,
                    help="1: untimed probes of K steps after the W warmup steps until the step time has settled "
                         "(clock ramp after the idle set-up phase); 0: none")
    ap.add_argument("--multi-rhs", type=int, default=1,
                    help="1 (library default): all right-hand sides of a tiled call in one launch; 0: one launch each")
    ap.add_argument("--windows", type=int, default=5,
                    help="the K-step window is repeated this many times; value / ms_per_step are those of the FIRST "
                         "window (the contract's timed region), ms_per_step_windows reports min / median / max")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N outside torch.distributed.run: start the N ranks as a child process tree.  Nothing in
    this process has touched the GPU yet (no torch import, no HIP call)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def compulsory_bytes_per_cell(k, nrhs, ev=False):
    """8 R [k(k+1) + k(k+1)/2 + k(k+2)] + 24 bytes per cell (160 at k = 2, R = 1; 296 for
    the two stress rows; 288 at k = 3); EV writes the conforming space instead: 1.5 k facet + k^2-k
    interior DOFs per cell (136 B/cell at k=2)."""
    nout = (1.5 * k + k * k - k) if ev else k * (k + 2)
    return 8 * nrhs * (k * (k + 1) + k * (k + 1) // 2 + nout) + 24


def metric_name(args):
    """BASELINE.json's metric for the headline configuration; the other configurations of
    BASELINE.json name themselves (same unit, same residual)."""
    tri = f"{4 * args.n * args.n / 1e6:g}M-tri"
    if args.stress:
        return (f"equilibrated patches/s (fp64) on {tri} linear elasticity k={args.k}, weakly symmetric "
                f"stress (two rows + symmetry step); L2 flux-divergence and weak-symmetry residuals")
    if args.ev:
        return (f"equilibrated patches/s (fp64) on {tri} Poisson k={args.k}, FluxEqlbEV "
                f"constrained minimisation; L2 flux-divergence residual")
    if args.k == 2 and args.n == 500:
        return HEADLINE_METRIC
    return f"equilibrated patches/s (fp64) on {tri} Poisson k={args.k}; L2 flux-divergence residual"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with "
                 f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` "
                 f"(or run `python bench.py --gpus {args.gpus}` outside torch.distributed.run)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    k, n = args.k, args.n
    nrt, nd = k * (k + 2), k * (k + 1) // 2

    # ---- problem setup (not timed, host only): mesh strip of this rank, compatible synthetic data
    from dolfinx_eqlb_amd import distributed as dd
    from dolfinx_eqlb_amd.eqlb import check_eqlb_conditions as chk
    from synthetic import make_compatible_data, make_compatible_stress_data
    part = dd.StripPartition(n, rank, world, shuffle_seed=args.shuffle)
    mesh = part.mesh
    ft = part.facet_types()
    nrhs = 2 if args.stress else max(1, args.nrhs)
    if args.nrhs > 1 and (args.stress or args.ev or world > 1):
        sys.exit("bench.py: --nrhs > 1 is a one-GPU line of the plain semi-explicit equilibration")

    def strip_data(seed):
        """N > 1: ONE global data set - every strip carries the same rows, made compatible on the
        x-periodic unit strip, so the hat-function orthogonality holds on the union [0, N] x [0, 1] and
        the ghost cells of a rank (first column of the next strip) see what their owner sees.  The
        divergence residual of the owned cells then checks the halo exchange across the ranks."""
        from dolfinx_eqlb_amd.mesh import create_rectangle
        unit = create_rectangle(n, n)
        pm = np.arange(unit.nnodes)
        pm[np.arange(n + 1) * (n + 1) + n] = np.arange(n + 1) * (n + 1)  # x = 1 -> x = 0
        uft = np.zeros((1, unit.nfacets), dtype=np.int8)
        bf = unit.boundary_facets()
        ym = unit.facet_midpoints()[bf][:, 1]
        uft[0, bf[(np.abs(ym) < 1e-12) | (np.abs(ym - 1) < 1e-12)]] = 1
        Gu, fu = make_compatible_data(unit, k, uft, seed=seed, node_map=pm)
        gi, gj, gt = part.grid_ids
        src = (gj * n + (gi % n)) * 4 + gt
        return (np.ascontiguousarray(Gu.reshape(unit.ncells, -1)[src]).ravel(),
                np.ascontiguousarray(fu.reshape(unit.ncells, -1)[src]).ravel())

    if args.stress:
        ft = np.repeat(ft, 2, axis=0)
        if world > 1:
            rows = [strip_data(20241003 + 17 * r) for r in range(2)]  # force balance only (throughput run)
            G = np.concatenate([r_[0] for r_ in rows])
            f = np.concatenate([r_[1] for r_ in rows])
        else:
            # force- AND moment-balanced rows (what a P_k Galerkin elasticity solution provides): without
            # the moment balance the weak-symmetry patch problems are solvable but not symmetric
            G2, f2 = make_compatible_stress_data(mesh, k, ft, seed=20241003)
            G, f = G2.ravel(), f2.ravel()
    elif world > 1:
        G, f = strip_data(20241003)
    elif nrhs > 1:
        ft = np.repeat(ft, nrhs, axis=0)
        rows = [make_compatible_data(mesh, k, ft[:1], seed=20241003 + 17 * r) for r in range(nrhs)]
        G = np.concatenate([r_[0] for r_ in rows])
        f = np.concatenate([r_[1] for r_ in rows])
    else:
        G, f = make_compatible_data(mesh, k, ft, seed=20241003)

    # the all-cores CPU figure forks worker processes: do it BEFORE this process touches the GPU
    cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stress and not args.ev and nrhs == 1:
        cpu_all = cpu_baseline_all_cores(mesh, k, ft, G, f)

    import torch
    import torch.distributed as dist

    from dolfinx_eqlb_amd import cpp

    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.zeros(1, device=dev)  # HIP context up before the cold timing starts
    torch.cuda.synchronize()

    # ---- cold path (SURVEY 8d "cold alongside"): mesh upload, handle + patch construction, one call on
    # host arrays (H2D of G, f, sweep, D2H of the flux) - every step the reference's timed region has
    t_c0 = time.perf_counter()
    dmesh = cpp.DeviceMesh(mesh)
    t_c1 = time.perf_counter()
    if args.ev:
        if args.stress:
            raise SystemExit("--ev runs without --stress")
        eq = cpp.ConstrainedMinEquilibrator(dmesh, k, nrhs)
        fused = True
        eq.set_option("accumulate", args.accumulate)
        if args.tile_cells:
            eq.set_option("tile_cells", args.tile_cells)
        eq.set_boundary(ft, node_mask=part.node_mask)
        nout = eq.ndofs
    else:
        eq = cpp.SemiExplicitEquilibrator(dmesh, k, nrhs, reconstruct_stress=args.stress)
        if args.solver is not None:
            eq.set_option("solver", args.solver)
        if args.scatter is None:  # the library default (AUTO): tiled launches for k <= 3, also for the RT_2 stress
            args.scatter = 2 if (k <= 3 and args.solver in (None, 1) and not (args.stress and k != 2)) else 0
        eq.set_option("scatter", args.scatter)
        # (k = 4: register solver, one launch per lanes-per-patch bin + slot reduction)
        fused = (bool(args.fused) and args.solver in (None, 1) and k <= 3) or args.scatter == 2
        eq.set_option("fused", int(fused))
        eq.set_option("accumulate", args.accumulate)
        eq.set_option("multi_rhs", args.multi_rhs)
        if args.tile_cells:
            eq.set_option("tile_cells", args.tile_cells)
        if world > 1 and args.scatter == 2:
            eq.set_priority_cells(part.send_cells)  # their tiles run first: halo exchange behind the rest
        eq.set_boundary(ft, node_mask=part.node_mask)
        nout = mesh.ncells * nrt
    t_sb = time.perf_counter()  # end of handle creation + patch construction
    # (the output vector exists before the call, as the reference's flux Functions do, FluxEqlbSE.py:104-108: a
    # freshly calloc'ed array would add its page faults to the transfer)
    x_cold = np.zeros((nrhs, nout))
    x_cold.fill(0.0)
    t_c2 = time.perf_counter()
    eq.equilibrate_host(G.reshape(nrhs, -1), f.reshape(nrhs, -1), x_cold)
    t_c3 = time.perf_counter()
    npatch_local = eq.num_patches
    cold = {"mesh_upload_ms": (t_c1 - t_c0) * 1e3, "create_set_boundary_ms": (t_sb - t_c1) * 1e3,
            "host_call_ms": (t_c3 - t_c2) * 1e3, "cold_ms": ((t_sb - t_c1) + (t_c3 - t_c2)) * 1e3,
            "patches_per_s": npatch_local / ((t_sb - t_c1) + (t_c3 - t_c2)),
            "note": "cold_ms = handle creation + patch construction (set_boundary) + one call on pageable host "
                    "arrays; the mesh upload is once per mesh"}
    tiling = eq.tiling_info() if (not args.ev and args.scatter == 2) else None

    d_G = torch.from_numpy(G).to(dev)
    d_f = torch.from_numpy(f).to(dev)
    d_x = torch.zeros(nrhs * nout, dtype=torch.float64, device=dev)
    comm = None
    if world > 1 and args.halo == "rccl":
        # a communicator of the library's own (the C++ host's transport, include/eqlb.h: eqlb_halo_exchange): the
        # unique id travels through the torch.distributed group that the launcher set up
        idt = torch.zeros(129, dtype=torch.uint8, device=dev)  # id + "rank 0 has one" (all ranks must agree
        if rank == 0:                                            # before the collective ncclCommInitRank)
            try:
                idt[:128] = torch.frombuffer(bytearray(cpp.RcclComm.unique_id()), dtype=torch.uint8).to(dev)
                idt[128] = 1
            except RuntimeError as e:
                print(f"[bench] no RCCL unique id through the C ABI ({e}); torch transport", file=sys.stderr, flush=True)
        dist.broadcast(idt, 0)
        idh = idt.cpu().numpy()
        ok = float(idh[128])
        if ok:
            try:
                comm = cpp.RcclComm(idh[:128].tobytes(), world, rank)
            except RuntimeError as e:
                print(f"[bench rank {rank}] RCCL communicator through the C ABI failed ({e}); torch transport",
                      file=sys.stderr, flush=True)
                ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0 and comm is not None:
            comm.destroy()
            comm = None
    if world > 1 and args.ev:
        # conforming DOFs of the ghost cells (facet DOFs travel with them) go to their owner
        halo = dd.HaloExchange(part, 1, dev, nrhs, lists=part.conforming_halo(k), nentries=nout, comm=comm)
    else:
        halo = dd.HaloExchange(part, nrt, dev, nrhs, comm=comm) if world > 1 else None
    stream = torch.cuda.current_stream().cuda_stream

    pG, pf, px = d_G.data_ptr(), d_f.data_ptr(), d_x.data_ptr()
    two_phase = halo is not None and not args.ev and args.scatter == 2
    nprio = eq.num_priority_tiles if two_phase else 0

    def step():
        if two_phase:
            # tiles owning ghost cells, then the reverse halo of their rows in flight behind the rest
            eq.equilibrate_device_tiles(pG, pf, px, 0, nprio, stream)
            reqs = halo.start(d_x)
            eq.equilibrate_device_tiles(pG, pf, px, nprio, -1, stream)
            halo.finish(d_x, reqs)
            return
        eq.equilibrate_device(pG, pf, px, stream)
        if halo is not None:
            halo.reduce(d_x)

    def barrier():
        if world > 1:
            dist.barrier()

    # correctness of this very configuration (single sweep into a zeroed vector, device-memory path)
    step()
    torch.cuda.synchronize()
    eq.check_status(stream)
    x_host = d_x.cpu().numpy().copy()
    checks = {}
    if world == 1:
        # the host-memory call of the cold path and the device-memory call agree bitwise
        checks["host_call_equals_device_call"] = bool(np.array_equal(x_host, x_cold.ravel()))
    del x_cold
    res = nrm = None
    if world == 1 and args.ev:
        from dolfinx_eqlb_amd.eqlb.conforming import conforming_to_broken
        res, nrm = chk.divergence_residual(mesh, k, conforming_to_broken(mesh, k, x_host),
                                           np.zeros_like(G), f)
    elif world == 1 and args.stress:
        xr = x_host.reshape(2, -1)
        rr = [chk.divergence_residual(mesh, k, xr[r], G.reshape(2, -1)[r], f.reshape(2, -1)[r]) for r in range(2)]
        res = float(np.sqrt(rr[0][0] ** 2 + rr[1][0] ** 2))
        nrm = float(np.sqrt(rr[0][1] ** 2 + rr[1][1] ** 2))
        checks["weak_symmetry_residual_max"] = chk.weak_symmetry_residual(mesh, k, xr)[0]
    elif world == 1:
        rr = [chk.divergence_residual(mesh, k, x_host.reshape(nrhs, -1)[r], G.reshape(nrhs, -1)[r],
                                      f.reshape(nrhs, -1)[r]) for r in range(nrhs)]
        res = float(np.sqrt(sum(a_ ** 2 for a_, _ in rr)))
        nrm = float(np.sqrt(sum(b_ ** 2 for _, b_ in rr)))
    elif args.ev:
        # across the ranks, conforming output: the reverse halo completes the DOFs on their OWNER; checked are the
        # owned cells whose DOFs all stay here (the cells along the interface to the next rank gave their facet
        # DOFs away: n of 4 n^2 cells)
        import types
        from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap, conforming_to_broken
        cd, _ = conforming_dofmap(mesh, k)
        sent = np.zeros(nout, dtype=bool)
        for d_ in part.conforming_halo(k)[0].values():
            sent[d_] = True
        own = np.nonzero(part.cell_owned & ~sent[cd].any(axis=1))[0]
        xb = conforming_to_broken(mesh, k, x_host.reshape(nrhs, -1)).reshape(nrhs, mesh.ncells, nrt)
        sub = types.SimpleNamespace(x=mesh.x, cell_nodes=mesh.cell_nodes[own], ncells=own.size)
        r2 = n2 = 0.0
        for r in range(nrhs):
            a_, b_ = chk.divergence_residual(sub, k, xb[r][own].ravel(), np.zeros(own.size * nd * 2),
                                             f.reshape(nrhs, -1, nd)[r][own].ravel())
            r2, n2 = r2 + a_ ** 2, n2 + b_ ** 2
        t = torch.tensor([r2, n2], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        res, nrm = float(np.sqrt(t[0].item())), float(np.sqrt(t[1].item()))
    elif not args.ev:
        # across the ranks: the owned cells hold their own rows + the rows the neighbour computed for them
        import types
        own = np.nonzero(part.cell_owned)[0]
        sub = types.SimpleNamespace(x=mesh.x, cell_nodes=mesh.cell_nodes[own], ncells=own.size)
        r2 = n2 = 0.0
        for r in range(nrhs):
            a_, b_ = chk.divergence_residual(sub, k, x_host.reshape(nrhs, -1, nrt)[r][own].ravel(),
                                             G.reshape(nrhs, -1, nd * 2)[r][own].ravel(),
                                             f.reshape(nrhs, -1, nd)[r][own].ravel())
            r2, n2 = r2 + a_ ** 2, n2 + b_ ** 2
        t = torch.tensor([r2, n2], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        res, nrm = float(np.sqrt(t[0].item())), float(np.sqrt(t[1].item()))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # Clock settle (untimed, reported as `settle_probes`): a GPU that idled through the host-side set-up above
    # needs tens of milliseconds of work to reach the clocks it holds under load - with W = 3 warmup steps
    # (0.3 ms) the first timed window of round 2 ran 5 - 15 % slower than the steady state.  Probes of K steps are
    # repeated for at least 40 ms and until two consecutive ones agree within 1 % (at most 24); the timed region
    # below is unchanged: exactly K steps, bracketed by barrier + synchronize.
    settle = []
    if args.settle:
        # (measured: the step time of the 1M-triangle RT_2 sweep keeps falling for ~30 ms of back-to-back
        # launches, 0.096 -> 0.081 ms: profiles/r03_settle_probes.txt)
        t_settle0 = time.perf_counter()
        while len(settle) < 24:
            tp = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            settle.append((time.perf_counter() - tp) / args.steps * 1e3)
            busy_ms = (time.perf_counter() - t_settle0) * 1e3
            if len(settle) >= 2 and busy_ms >= 40.0 and abs(settle[-1] - settle[-2]) <= 0.01 * settle[-2]:
                break
    barrier()
    # Timed region: K steps back to back.  When a step is ONE kernel launch (tiled / fused launch
    # on one GPU) the kernel's average duration is taken from two HIP events that bracket the whole
    # timed region on the launch stream (torch's current stream is the stream handed to the
    # library); per-launch event pairs would put ~5 us of barrier packets between the launches.
    single_kernel = world == 1 and fused and (args.ev and k <= 3 or (not args.ev and args.scatter == 2))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eq.set_option("timing", 0)  # no per-launch events inside the timed region
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([npatch_local], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        npatch_total = int(cnt.item())
    else:
        npatch_total = npatch_local
    step_dev_ms = ev0.elapsed_time(ev1) / args.steps  # device time of one step on the launch stream
    # the same window again (bench hygiene: one 2 ms window sits inside the box-to-box noise): wall clock per
    # window, bracketed like the first one
    win_ms = [elapsed / args.steps * 1e3]
    for _ in range(max(0, args.windows - 1)):
        torch.cuda.synchronize()
        barrier()
        tw0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        tw = time.perf_counter() - tw0
        if world > 1:
            t = torch.tensor([tw], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tw = float(t.item())
        win_ms.append(tw / args.steps * 1e3)

    # ---- the halo exchange alone (pack, transfer, unpack-add; the rows it moves are whatever the last step left)
    halo_info = None
    if halo is not None:
        scratch = torch.zeros_like(d_x)
        for _ in range(2):
            halo.reduce(scratch)
        torch.cuda.synchronize()
        barrier()
        th0 = time.perf_counter()
        for _ in range(args.steps):
            halo.reduce(scratch)
        torch.cuda.synchronize()
        barrier()
        th = torch.tensor([(time.perf_counter() - th0) / args.steps * 1e3], dtype=torch.float64, device=dev)
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        if args.ev:
            hs, hr = part.conforming_halo(k)
            b_s, b_r = 8 * nrhs * sum(len(v) for v in hs.values()), 8 * nrhs * sum(len(v) for v in hr.values())
        else:
            b_s, b_r = part.halo_bytes(nrt, nrhs)
        hb = torch.tensor([float(b_s), float(b_r)], dtype=torch.float64, device=dev)
        hmax = hb.clone()
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        halo_info = {"transport": "eqlb_halo_exchange (RCCL ncclSend/ncclRecv through the C ABI)" if comm is not None
                     else "torch.distributed isend/irecv (nccl backend = RCCL)",
                     "rank0_bytes_sent": int(b_s), "rank0_bytes_received": int(b_r),
                     "max_bytes_sent": int(hmax[0].item()), "max_bytes_received": int(hmax[1].item()),
                     "halo_only_ms": float(th.item()),
                     "note": "halo_only_ms = pack + transfer + unpack-add alone, max over ranks, mean of K calls; "
                             "inside a step it overlaps the sweep of the interior tiles"}
        del scratch

    # ---- per-kernel device times.  A step that is ONE kernel launch: the two HIP events that bracket the
    # timed region.  A step of several kernels: the same K steps once more, now with the library's HIP
    # event pairs around each kernel group on the launch stream (they would put barrier packets between
    # the launches of the timed region itself).
    bytes_sweep = float(compulsory_bytes_per_cell(k, nrhs, args.ev) * part.ncells_owned)
    if args.ev:  # library default: tiled launch for k <= 3
        patch_kernel = f"k_se_patch_tiled<K={k},EV>" if k <= 3 else f"k_ev_patch_fused<K={k}>"
    elif args.scatter == 2 and args.stress:
        patch_kernel = "k_se_stress_tiled"
    elif args.scatter == 2:
        patch_kernel = f"k_se_patch_tiled<K={k}>"
    elif fused:
        patch_kernel = f"k_se_patch_fused<K={k}>"
    else:
        patch_kernel = None
    bins_ms = None
    if single_kernel:
        if nrhs > 1 and not args.stress:
            patch_kernel = (f"k_se_patch_tiled_multi<K={k}> ({nrhs} right-hand sides)" if args.multi_rhs
                            else patch_kernel + f" x{nrhs} launches")
        kernels_ms = {patch_kernel: step_dev_ms}
        timing_method = "two HIP events around the timed region / steps"
    elif two_phase:
        # per-launch event pairs would serialise the two launches of a step against the halo exchange
        kernels_ms = {"step (two tile-range launches + halo pack / RCCL send-recv / unpack)": step_dev_ms}
        timing_method = "two HIP events around the timed region / steps (whole step incl. halo exchange)"
    else:
        eq.set_option("timing", 1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        bins_ms = [eq.last_kernel_ms(b) for b in range(5)]
        kernels_ms = {}
        if fused:
            kernels_ms[patch_kernel + (f" x{nrhs} rows" if nrhs > 1 else "")] = bins_ms[0]
        else:
            kernels_ms.update({f"k_se_patch<K={k},P={4 << b}>": bins_ms[b] for b in range(5) if bins_ms[b] > 0})
        if args.stress:
            lean = k == 2  # no flux BCs on the benchmark's stress rows
            kernels_ms["k_se_weaksym_lean" if lean else f"k_se_weaksym<K={k}>"] = eq.last_kernel_ms(6)
        red = eq.last_kernel_ms(5)
        if red > 0:
            kernels_ms["k_ev_reduce" if args.ev else "k_reduce_slots"] = red
        if halo is not None:
            kernels_ms["step incl. halo exchange"] = step_dev_ms
        timing_method = ("HIP event pair per kernel group, measured over the same K steps repeated after the "
                         "timed region (mean)")
    eq.set_option("timing", 0)
    kname = max(kernels_ms, key=kernels_ms.get)
    if single_kernel or two_phase:
        alg_bytes, t_roof = bytes_sweep, step_dev_ms
        roof_note = None
    elif fused or args.stress:
        # several kernels share the sweep's compulsory bytes (inputs read once, output written once):
        # the fraction is the whole step's, the kernel named is the longest one
        alg_bytes = bytes_sweep
        t_roof = step_dev_ms
        roof_note = ("multi-kernel step: achieved = compulsory bytes of the whole step / device time of a step "
                     "(two HIP events around the timed region / steps); `kernel` is the longest one (all_kernels_ms)")
    else:      # one launch per bin: share of the sweep done by the dominant bin's launch
        ncells_bin = part.patch_cells_per_bin()
        dom = int(np.argmax(bins_ms))
        alg_bytes = bytes_sweep * ncells_bin[dom] / float(sum(ncells_bin))
        t_roof = bins_ms[dom]
        roof_note = None
    achieved = alg_bytes / (t_roof * 1e-3) / 1e9 if t_roof > 0 else 0.0
    peak = 8000.0
    pmc = measured_counters(kname) if n == 500 and world == 1 else {}

    out = {
        "metric": metric_name(args),
        "value": npatch_total * args.steps / elapsed,
        "unit": "patches/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "settle_probes": {"n": len(settle), "ms_per_step": [float(v) for v in settle],
                          "note": "untimed probes of K steps between the W warmup steps and the timed region, "
                                  "repeated for at least 40 ms and until two agree within 1 % (GPU clocks after the idle set-up)"},
        "ms_per_step_windows": {"n": len(win_ms), "min": float(np.min(win_ms)), "median": float(np.median(win_ms)),
                                "max": float(np.max(win_ms)), "all": [float(w) for w in win_ms],
                                "note": "the K-step window repeated; value / ms_per_step are the first window's"},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"{'Linear elasticity' if args.stress else 'Poisson'} {4 * n * n} triangles per GPU "
                         f"(crossed unit square {n}x{n}), P{k} primal, "
                         f"{'FluxEqlbEV' if args.ev else 'FluxEqlbSE'} RT{k}"
                         f"{', two stress rows + weak symmetry' if args.stress else ''}, "
                         f"homogeneous Dirichlet, fp64"),
            "headline": metric_name(args) == HEADLINE_METRIC,
            "patches_per_gpu": npatch_local, "cells_per_gpu": int(part.ncells_owned),
            "nrhs": nrhs, "weak_symmetry": bool(args.stress),
            "partition": ("node-ownership strips, halo exchange behind the interior tiles" if two_phase
                          else "node-ownership strips") if world > 1 else "none",
            "solver": eq_solver_name(None if args.ev else args.solver),
            "scatter": ("tiled" if k <= 3 else "slots") if args.ev else eq_scatter_name(args.scatter),
            "accumulate": bool(args.accumulate),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kname,
            "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
            "traffic": pmc.get("traffic"),
            "traffic_stale": bool(pmc.get("stale", False)),
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernel_ms": kernels_ms[kname],
            "kernel_ms_method": timing_method,
            "all_kernels_ms": kernels_ms,
        },
        "cold": cold,
    }
    if roof_note:
        out["roofline"]["note"] = roof_note
    if pmc.get("insts_valu"):
        # the binding resource of this kernel is VALU issue + latency, not HBM:
        # issue floor = wave-level VALU instructions / 1024 SIMDs x 2.0 ns, the MEASURED issue time of a
        # wave64 instruction on one SIMD of this part at 4 waves/SIMD (fp64 FMA 2.10 ns, fp64 MUL / ADD and
        # 32-bit DPP moves 1.91 - 1.98 ns: tools/microbench/valu_rate.hip, profiles/r02_valu_rate.txt) -
        # i.e. 4 cycles at the ~2.0 GHz the chip holds under fp64 load, not at the 2.4 GHz of the data sheet
        floor_ms = pmc["insts_valu"] / 1024.0 * VALU_NS_PER_INSTRUCTION * 1e-6
        out["roofline"]["valu_floor_ms"] = floor_ms
        out["roofline"]["valu_frac"] = floor_ms / kernels_ms[kname] if kernels_ms[kname] > 0 else None
        out["roofline"]["valu_source"] = pmc.get("source")
    if halo_info is not None:
        out["halo"] = halo_info
    if tiling is not None:
        out["config"]["tiling"] = tiling
    if res is not None:
        out["div_residual_L2"] = res
        out["rhs_norm_L2"] = nrm
        out["div_residual_rel"] = res / nrm
    if checks:
        out["checks"] = checks

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.ev:
        out["cpu_baseline"] = cpu_baseline_ev(mesh, k, ft, G, f)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline and args.stress:
        out["cpu_baseline"] = cpu_baseline_stress(mesh, k, ft, G.reshape(2, -1), f.reshape(2, -1))
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (R right-hand sides: the CPU figure is the one-RHS sweep, R times the work per patch on the device side)
        out["cpu_baseline"] = cpu_baseline(mesh, k, ft[:1], G.reshape(nrhs, -1)[0], f.reshape(nrhs, -1)[0], npatch_local)
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.destroy()
        dist.destroy_process_group()


def measured_counters(kernel):
    """Per-launch PMC figures of a kernel from the committed rocprofv3 passes (profiles/traffic.json:
    HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE, SQ_INSTS_VALU); PMC counters cannot be read from inside
    the process.  Entries are either a byte count or {"traffic", "insts_valu", "source"}."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            v = json.load(fh).get(kernel)
    except OSError:
        return {}
    if v is None:
        return {}
    if not isinstance(v, dict):
        return {"traffic": v}
    # counters of an older build of the kernel are not this build's: instruction counts are dropped, the traffic
    # figure is kept but marked
    sha = kernel_source_sha(kernel)
    if v.get("kernel_sha") and sha and v["kernel_sha"] != sha:
        return {"traffic": v.get("traffic"), "stale": True,
                "source": f"{v.get('source')} (taken on an older build of the kernel)"}
    return v


def kernel_source_sha(kernel):
    """First 16 hex digits of the sha256 of the source file that holds `kernel` (profiles/traffic.json)."""
    import hashlib
    fname = "eqlb_stress_tiled.hip" if "stress_tiled" in kernel else "eqlb_se_kernels.hip"
    try:
        with open(os.path.join(ROOT, "dolfinx_eqlb_amd", "csrc", fname), "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()[:16]
    except OSError:
        return None


def eq_solver_name(v):
    return {None: "shuffle", 0: "lds_cholesky", 1: "shuffle", 9: "none(timing only)"}[v]


def eq_scatter_name(v):
    return {None: "slots", 0: "slots", 1: "atomic", 2: "tiled"}[v]


def cpu_baseline(mesh, k, ft, G, f, npatch):
    """Single-thread CPU restatement of the reference algorithm (oracle/eqlb_oracle.c) on the
    same mesh and data.  The whole sweep takes only a few seconds on one core, so the sample is
    the complete workload (all patches), best of 3."""
    from oracle import oracle
    x = np.zeros((1, mesh.ncells * k * (k + 2)))
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        oracle.se_reconstruct(mesh, k, ft, G[None], f[None], flux_hdiv=x)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": mesh.nnodes / best, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"all {npatch} patches of the workload (one full sweep), best of 3, "
                      f"{best:.2f} s per sweep",
            "note": "CPU restatement of the reference algorithm (not the dolfinx_eqlb binary)"}


def cpu_baseline_stress(mesh, k, ft, G, f):
    """The same restatement with the weak-symmetry step (se/solve_patch_weaksym.hpp) on a bounded
    sample: a contiguous range of nodes of the workload, extended to ~10 s of CPU work."""
    from oracle import oracle
    x = np.zeros((2, mesh.ncells * k * (k + 2)))
    nn = min(mesh.nnodes, 20000)
    t0 = time.perf_counter()
    oracle.se_reconstruct(mesh, k, ft, G, f, flux_hdiv=x, node_range=(0, nn), stress=True)
    dt = time.perf_counter() - t0
    nn2 = int(min(mesh.nnodes, max(nn, nn * 10.0 / max(dt, 1e-3))))
    if nn2 > nn:
        t0 = time.perf_counter()
        oracle.se_reconstruct(mesh, k, ft, G, f, flux_hdiv=x, node_range=(0, nn2), stress=True)
        dt = time.perf_counter() - t0
        nn = nn2
    return {"value": nn / dt, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"patches of nodes 0..{nn} of the workload (one pass, two rows + weak symmetry), {dt:.2f} s",
            "note": "CPU restatement of the reference algorithm (not the dolfinx_eqlb binary)"}


_FORK_JOB = None  # (mesh, k, ft, G, f) inherited by the forked workers of cpu_baseline_all_cores


def _oracle_range(rng):
    from oracle import oracle
    mesh, k, ft, G, f = _FORK_JOB
    x = np.zeros((1, mesh.ncells * k * (k + 2)))
    oracle.se_reconstruct(mesh, k, ft, G[None], f[None], flux_hdiv=x, node_range=rng)
    return rng[1] - rng[0]


def cpu_baseline_all_cores(mesh, k, ft, G, f):
    """Context figure: the same single-thread restatement run by one process per host
    core on disjoint node ranges (patches are independent; every worker accumulates into its own
    vector, the final reduction is not included).  Forked workers: call before the GPU is touched."""
    global _FORK_JOB
    import multiprocessing as mp
    from oracle import oracle
    oracle.build()
    ncores = max(1, min(16, os.cpu_count() or 1))
    _FORK_JOB = (mesh, k, ft, G, f)
    bounds = np.linspace(0, mesh.nnodes, 4 * ncores + 1).astype(int)
    ranges = [(int(a), int(b)) for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    best = None
    with mp.get_context("fork").Pool(ncores) as pool:
        pool.map(_oracle_range, ranges[:ncores])  # warm up the workers (library load)
        for rep in range(2):
            t0 = time.perf_counter()
            done = sum(pool.map(_oracle_range, ranges, chunksize=1))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    _FORK_JOB = None
    return {"value": done / best, "unit": "patches/s", "cores": ncores, "kind": "port",
            "sample": f"all {done} patches of the workload on {ncores} worker processes, best of 2, "
                      f"{best:.2f} s per sweep",
            "note": "context only: the reference is single-threaded (se/reconstruction.hpp:286)"}


def cpu_baseline_ev(mesh, k, ft, G, f):
    """CPU restatement of the reference's EV algorithm (saddle-point LU per patch,
    oracle/eqlb_oracle_ev.c) on a bounded sample: a contiguous range of nodes of the same mesh."""
    from dolfinx_eqlb_amd.eqlb.conforming import conforming_dofmap
    from oracle import oracle
    cd, nd = conforming_dofmap(mesh, k)
    x = np.zeros((1, nd))
    nn = min(mesh.nnodes, 20000)
    t0 = time.perf_counter()
    oracle.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd, flux_hdiv=x, node_range=(0, nn))
    dt = time.perf_counter() - t0
    # extend the sample to ~10 s of CPU work
    nn2 = int(min(mesh.nnodes, max(nn, nn * 10.0 / max(dt, 1e-3))))
    if nn2 > nn:
        t0 = time.perf_counter()
        oracle.ev_reconstruct(mesh, k, ft, G[None], f[None], cd, nd, flux_hdiv=x,
                              node_range=(0, nn2))
        dt = time.perf_counter() - t0
        nn = nn2
    return {"value": nn / dt, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"patches of nodes 0..{nn} of the workload (one pass), {dt:.2f} s",
            "note": "CPU restatement of the reference's EV algorithm (dense LU of the "
                    "saddle-point system per patch); not the dolfinx_eqlb binary"}


if __name__ == "__main__":
    main()
