"""Multi-GPU decomposition of the patch loop: node-ownership strips + reverse halo reduction.

The reference has no distributed equilibration (its node loop runs over
index_map(0)->size_local() owned nodes, se/reconstruction.hpp:90,286, and the flux of ghost
cells is never reduced, FluxEqlbSE.py:164 "TODO").  Design here: patches are
independent and partitioned by node ownership; every cell's RT DOFs receive exactly one
contribution per vertex patch (se/solve_patch_semiexplt.hpp:1157-1160), so the only exchange
is the ADDITION of the partial sums a rank computed for cells owned by a neighbour.  With
strips along x each rank has at most two neighbours: point-to-point send/recv over xGMI
(torch.distributed NCCL backend == RCCL), no collective on the data path.

Strip r owns the squares with column index in [r n, (r+1) n) of a (world n) x n crossed grid on
[0, world] x [0, 1] and the nodes with x in (r, r+1] (x = 0 included for r = 0).  Its local
mesh also holds the 3 n ghost triangles of strip r+1 that touch its right interface nodes;
after the local sweep their rows are sent to rank r+1 and added there.
"""

import numpy as np

from .mesh import create_rectangle


class StripPartition:
    def __init__(self, n: int, rank: int = 0, world: int = 1, shuffle_seed=None):
        self.n, self.rank, self.world = n, rank, world
        ghost = 1 if rank < world - 1 else 0
        nx = n + ghost
        self.nx = nx

        def keep(i, j, t):
            return ~((i == n) & (t == 1))  # ghost column: drop the triangle not touching x = r+1

        seed = shuffle_seed if world == 1 else None  # shared cells need one local vertex order
        self.mesh, (gi, gj, gt) = create_rectangle(
            nx, n, x0=float(rank), x1=float(rank) + nx / n, keep_cell=keep if ghost else None,
            shuffle_seed=seed, return_grid_ids=True)
        self.grid_ids = (gi, gj, gt)
        m = self.mesh
        self.cell_owned = gi < n
        self.ncells_owned = int(self.cell_owned.sum())

        # node ownership: corner nodes (nx+1) x (n+1) row-major, then centres nx x n
        ncorner = (nx + 1) * (n + 1)
        col_corner = np.arange(ncorner) % (nx + 1)
        col_centre = np.arange(nx * n) % nx
        lo = 0 if rank == 0 else 1
        owned_corner = (col_corner >= lo) & (col_corner <= n)
        owned_centre = col_centre < n
        mask = np.concatenate([owned_corner, owned_centre]).astype(np.uint8)
        assert mask.size == m.nnodes
        self.node_mask = None if world == 1 else mask

        # halo lists, both sides ordered by (row j, triangle t) of the shared column
        order = np.lexsort((gt, gj))
        self.send_cells = np.zeros(0, dtype=np.int64)  # ghost cells -> rank + 1
        self.recv_cells = np.zeros(0, dtype=np.int64)  # own first column <- rank - 1
        if ghost:
            sel = order[(gi == n)[order]]
            self.send_cells = sel.astype(np.int64)
        if rank > 0:
            sel = order[((gi == 0) & (gt != 1))[order]]
            self.recv_cells = sel.astype(np.int64)

    def facet_types(self, nrhs: int = 1):
        """Homogeneous Dirichlet on every boundary facet of the local mesh (the artificial
        ones only touch nodes this rank does not own)."""
        ft = np.zeros((nrhs, self.mesh.nfacets), dtype=np.int8)
        ft[:, self.mesh.boundary_facets()] = 1
        return ft

    def patch_cells_per_bin(self):
        """Number of (patch, cell) pairs handled by the kernel launch of each lane-group bin
        (P = 4, 8, 16, 32, 64 >= number of patch facets)."""
        m = self.mesh
        nc = np.diff(m.node_cells_offsets)
        nf = np.diff(m.node_facets_offsets)
        sel = np.ones(m.nnodes, dtype=bool) if self.node_mask is None else self.node_mask.astype(bool)
        out = []
        lo = 0
        for P in (4, 8, 16, 32, 64):
            inbin = sel & (nf > lo) & (nf <= P)
            out.append(int(nc[inbin].sum()))
            lo = P
        return out


class HaloExchange:
    """Reverse halo reduction of the ghost-cell rows of the RT coefficient vector."""

    def __init__(self, part: StripPartition, nrt: int, device, nrhs: int = 1):
        import torch
        self.part, self.nrt, self.nrhs = part, nrt, nrhs
        self.send_idx = torch.from_numpy(part.send_cells).to(device)
        self.recv_idx = torch.from_numpy(part.recv_cells).to(device)
        self.send_buf = torch.zeros((nrhs, part.send_cells.size, nrt), dtype=torch.float64, device=device)
        self.recv_buf = torch.zeros((nrhs, part.recv_cells.size, nrt), dtype=torch.float64, device=device)
        self._ops = None

    def reduce(self, x):
        """x: [nrhs * ncells * nrt] tensor; adds the neighbour's partial sums to the owned rows
        and clears the ghost rows (their content now lives on the owner).  Device tensors use the
        two halo kernels of libeqlb_amd.so (one launch each) around the RCCL send/recv; CPU tensors
        (gloo tests) use plain indexing."""
        return self.finish(x, self.start(x))

    def start(self, x):
        """First half: pack the ghost rows (final once the tiles that own them have run) and post
        the send / receive.  Work enqueued on the current stream after this call overlaps the
        transfer; finish() orders the unpack behind it."""
        import torch
        import torch.distributed as dist
        part = self.part
        xv = x.view(self.nrhs, part.mesh.ncells, self.nrt)
        on_gpu = x.is_cuda
        ns, nr = int(part.send_cells.size), int(part.recv_cells.size)
        stream = torch.cuda.current_stream().cuda_stream if on_gpu else 0
        ops = []
        if ns:
            if on_gpu:
                from . import cpp
                cpp.halo_pack(x.data_ptr(), self.send_idx.data_ptr(), self.send_buf.data_ptr(),
                              self.nrhs, ns, self.nrt, part.mesh.ncells, True, stream)
            else:
                self.send_buf.copy_(xv[:, self.send_idx, :])
                xv[:, self.send_idx, :] = 0.0
        if self._ops is None:  # the descriptors are reused: same buffers, same peers every step
            if ns:
                ops.append(dist.P2POp(dist.isend, self.send_buf, part.rank + 1))
            if nr:
                ops.append(dist.P2POp(dist.irecv, self.recv_buf, part.rank - 1))
            self._ops = ops
        return dist.batch_isend_irecv(self._ops) if self._ops else []

    def finish(self, x, reqs):
        """Second half: wait for the transfer (the current stream waits, not the host) and add the
        received partial sums to the owned rows."""
        import torch
        part = self.part
        on_gpu = x.is_cuda
        nr = int(part.recv_cells.size)
        for req in reqs:
            req.wait()
        if nr:
            if on_gpu:
                from . import cpp
                stream = torch.cuda.current_stream().cuda_stream
                cpp.halo_unpack_add(x.data_ptr(), self.recv_idx.data_ptr(), self.recv_buf.data_ptr(),
                                    self.nrhs, nr, self.nrt, part.mesh.ncells, stream)
            else:
                xv = x.view(self.nrhs, part.mesh.ncells, self.nrt)
                xv[:, self.recv_idx, :] += self.recv_buf
        return x
