"""Multi-GPU decomposition of the patch loop: node ownership + reverse halo reduction.

The reference has no distributed equilibration (its node loop runs over
index_map(0)->size_local() owned nodes, se/reconstruction.hpp:90,286, and the flux of ghost
cells is never reduced, FluxEqlbSE.py:164 "TODO").  Design here: patches are
independent and partitioned by node ownership; every cell's RT DOFs receive exactly one
contribution per vertex patch (se/solve_patch_semiexplt.hpp:1157-1160), so the only exchange
is the ADDITION of the partial sums a rank computed for cells (or conforming DOFs) owned by a
neighbour: point-to-point send/recv over xGMI (torch.distributed NCCL backend == RCCL), no
collective on the data path.

Two producers of a decomposition share one interface (mesh, node_mask, cell_owned, send / recv
lists per peer):

  Partition       any mesh, any (node owner, cell owner) pair - the ownership-driven loop of the
                  reference for an arbitrary DOLFINx partition; built from the GLOBAL mesh.
  StripPartition  the benchmark's strips of a crossed rectangle, built WITHOUT the global mesh
                  (8M triangles per node would not be replicated on every rank).

A rank's local mesh holds every cell with a vertex it owns: its own cells plus the ghost cells
(owned by the rank that owns their highest-ranked vertex) its boundary patches reach into.
"""

import numpy as np

from .mesh import create_mesh, create_rectangle


def _patch_cells_per_bin(mesh, node_mask):
    """Number of (patch, cell) pairs handled by the kernel launch of each lane-group bin
    (P = 4, 8, 16, 32, 64 >= number of patch facets)."""
    nc = np.diff(mesh.node_cells_offsets)
    nf = np.diff(mesh.node_facets_offsets)
    sel = np.ones(mesh.nnodes, dtype=bool) if node_mask is None else node_mask.astype(bool)
    out = []
    lo = 0
    for P in (4, 8, 16, 32, 64):
        inbin = sel & (nf > lo) & (nf <= P)
        out.append(int(nc[inbin].sum()))
        lo = P
    return out


class Partition:
    """Decomposition of an arbitrary mesh by node ownership.

    node_owner [nnodes]: rank that equilibrates the patch of each node; cell_owner [ncells]
    (default: the highest owner among the cell's vertices): rank that holds the final RT DOFs of
    the cell.  Local cells keep the local vertex order of the global mesh, so shared cells agree
    on every rank.  send[q] / recv[q]: local cell ids whose rows go to / arrive from rank q, both
    ordered by global cell id."""

    def __init__(self, gmesh, node_owner, rank: int, world: int, cell_owner=None):
        node_owner = np.asarray(node_owner)
        self.rank, self.world = rank, world
        vo = node_owner[gmesh.cell_nodes]                      # [ncells, 3]
        if cell_owner is None:
            cell_owner = vo.max(axis=1)
        cell_owner = np.asarray(cell_owner)
        if np.any((cell_owner[:, None] != vo).all(axis=1)):
            raise RuntimeError("Partition: a cell must be owned by the owner of one of its vertices")
        local = np.nonzero((vo == rank).any(axis=1))[0]        # cells reached by this rank's patches
        self.cell_global = local
        gnodes = np.unique(gmesh.cell_nodes[local])
        self.node_global = gnodes
        g2l = -np.ones(gmesh.nnodes, dtype=np.int64)
        g2l[gnodes] = np.arange(gnodes.size)
        self.mesh = create_mesh(gmesh.x[gnodes, :2], g2l[gmesh.cell_nodes[local]].astype(np.int32))
        self.cell_owned = cell_owner[local] == rank
        self.ncells_owned = int(self.cell_owned.sum())
        mask = (node_owner[gnodes] == rank).astype(np.uint8)
        self.node_mask = None if world == 1 else mask
        # halo lists: my ghost cells go to their owner; the owner finds them among ITS cells that have
        # a vertex of mine (same rule evaluated from the other side), both sides sorted by global id
        self.send, self.recv = {}, {}
        for q in range(world):
            if q == rank:
                continue
            s = np.nonzero(cell_owner[local] == q)[0]
            if s.size:
                self.send[q] = s.astype(np.int64)
            r = np.nonzero((cell_owner[local] == rank) & (vo[local] == q).any(axis=1))[0]
            if r.size:
                self.recv[q] = r.astype(np.int64)
        # facets of the global mesh -> local facets (for boundary data given on the global mesh)
        self._gmesh_nfacets = gmesh.nfacets
        key_g = gmesh.facet_nodes[:, 0].astype(np.int64) * gmesh.nnodes + gmesh.facet_nodes[:, 1]
        lf = gnodes[self.mesh.facet_nodes]                      # global node ids of the local facets
        key_l = np.minimum(lf[:, 0], lf[:, 1]).astype(np.int64) * gmesh.nnodes + np.maximum(lf[:, 0], lf[:, 1])
        order = np.argsort(key_g)
        self.facet_global = order[np.searchsorted(key_g[order], key_l)]
        # conforming DOFs (EV): a facet is owned by the highest owner of its cells and HELD by every rank
        # that owns a vertex of one of its cells (those ranks have it in their local mesh)
        off = gmesh.facet_cells_offsets
        c0 = gmesh.facet_cells[off[:-1]]
        c1 = gmesh.facet_cells[np.minimum(off[:-1] + 1, off[1:] - 1)]  # == c0 on boundary facets
        self._facet_owner = np.maximum(cell_owner[c0], cell_owner[c1])[self.facet_global]
        self._facet_holders = np.concatenate([vo[c0], vo[c1]], axis=1)[self.facet_global]  # [nf_local, 6]
        self._cell_owner_l = cell_owner[local]
        self._cell_holders = vo[local]

    @classmethod
    def from_local(cls, x_local, cell_nodes_local, node_owner_local, cell_owner_local, cell_global, rank: int,
                   world: int, node_global=None):
        """The same decomposition from what ONE rank holds - no global mesh anywhere (an 8M-triangle mesh is
        not replicated on the 8 ranks of a node; a DOLFINx mesh is distributed to begin with):

          x_local [nn, 2], cell_nodes_local [nc, 3]   the rank's cells = every cell with a vertex it owns
                                                      (own cells + the ghost layer), local node numbering
          node_owner_local [nn], cell_owner_local [nc] owning rank of every local node / cell
          cell_global [nc]                             a global cell id: orders the halo lists the same way on
                                                      both sides of an interface
          node_global [nn] (optional)                  global node ids (kept for the caller)

        A cell must be owned by the owner of one of its vertices, shared cells must carry the same local vertex
        order on every rank that holds them (both as in __init__).  The conforming (EV) halo needs facet
        ownership across ranks and is available from the global constructor only."""
        self = cls.__new__(cls)
        node_owner_local = np.asarray(node_owner_local)
        cell_owner_local = np.asarray(cell_owner_local)
        cell_global = np.asarray(cell_global, dtype=np.int64)
        cn = np.asarray(cell_nodes_local)
        self.rank, self.world = rank, world
        vo = node_owner_local[cn]
        if np.any((cell_owner_local[:, None] != vo).all(axis=1)):
            raise RuntimeError("Partition: a cell must be owned by the owner of one of its vertices")
        if not (vo == rank).any(axis=1).all():
            raise RuntimeError("Partition.from_local: every local cell must have a vertex owned by this rank")
        self.cell_global = cell_global
        self.node_global = None if node_global is None else np.asarray(node_global)
        self.mesh = create_mesh(np.asarray(x_local, dtype=np.float64)[:, :2], cn.astype(np.int32))
        self.cell_owned = cell_owner_local == rank
        self.ncells_owned = int(self.cell_owned.sum())
        self.node_mask = None if world == 1 else (node_owner_local == rank).astype(np.uint8)
        order = np.argsort(cell_global, kind="stable")
        self.send, self.recv = {}, {}
        for q in range(world):
            if q == rank:
                continue
            s = order[(cell_owner_local == q)[order]]
            if s.size:
                self.send[q] = s.astype(np.int64)
            r = order[((cell_owner_local == rank) & (vo == q).any(axis=1))[order]]
            if r.size:
                self.recv[q] = r.astype(np.int64)
        self._gmesh_nfacets = None
        return self

    def halo_bytes(self, width, nrhs=1):
        """(bytes sent, bytes received) by this rank in one reverse halo of rows of `width` doubles."""
        ns = sum(len(v) for v in self.send.values())
        nr = sum(len(v) for v in self.recv.values())
        return 8 * width * nrhs * ns, 8 * width * nrhs * nr

    @property
    def send_cells(self):
        """All ghost cells (priority cells of the two-phase sweep)."""
        return np.concatenate([v for _, v in sorted(self.send.items())]) if self.send else np.zeros(0, np.int64)

    def facet_types(self, global_facet_type):
        """facet_type table of the local mesh from the one of the global mesh [nrhs, nfacets]; the
        artificial boundary facets of the ghost layer (interior facets of the global mesh) only touch
        nodes this rank does not own and are marked as primal-Dirichlet facets."""
        gft = np.asarray(global_facet_type).reshape(-1, self._gmesh_nfacets)
        ft = gft[:, self.facet_global].astype(np.int8)
        artificial = np.zeros(self.mesh.nfacets, dtype=bool)
        artificial[self.mesh.boundary_facets()] = True
        artificial &= gft[0, self.facet_global] == 0
        ft[:, artificial] = 1
        return ft

    def patch_cells_per_bin(self):
        return _patch_cells_per_bin(self.mesh, self.node_mask)

    def conforming_halo(self, k):
        """Halo lists of the conforming RT_k DOFs (EV equilibrator, default numbering of
        eqlb/conforming.py): (send, recv) dicts of local DOF indices per peer, matching order (global
        facet id, then global cell id).  Every rank that holds a facet computes a partial value for its
        DOFs; the partial values are sent to the facet's owner (highest owner among its cells) and
        added there; interior DOFs travel with their cell."""
        m, r = self.mesh, self.rank
        ni = k * k - k
        jf = np.arange(k, dtype=np.int64)
        ji = np.arange(ni, dtype=np.int64)
        f_order = np.argsort(self.facet_global)
        c_order = np.argsort(self.cell_global)
        send, recv = {}, {}
        for q in range(self.world):
            if q == r:
                continue
            fs = f_order[(self._facet_owner == q)[f_order]]
            cs = c_order[(self._cell_owner_l == q)[c_order]]
            d = np.concatenate([(fs[:, None] * k + jf).ravel(),
                                (m.nfacets * k + cs[:, None] * ni + ji).ravel()]).astype(np.int64)
            if d.size:
                send[q] = d
            fr = f_order[((self._facet_owner == r) & (self._facet_holders == q).any(axis=1))[f_order]]
            cr = c_order[((self._cell_owner_l == r) & (self._cell_holders == q).any(axis=1))[c_order]]
            d = np.concatenate([(fr[:, None] * k + jf).ravel(),
                                (m.nfacets * k + cr[:, None] * ni + ji).ravel()]).astype(np.int64)
            if d.size:
                recv[q] = d
        return send, recv


def _conforming_halo(mesh, k, send, recv):
    from .eqlb.conforming import conforming_dofmap
    cd, _ = conforming_dofmap(mesh, k)

    def dofs(cells):
        # per cell the same local DOF order on both sides; a facet shared by two cells of the list appears
        # twice - at the same positions on both sides - and is kept once (first occurrence)
        d = cd[cells].reshape(-1).astype(np.int64)
        first = np.sort(np.unique(d, return_index=True)[1])
        return d[first]
    return {q: dofs(c) for q, c in send.items()}, {q: dofs(c) for q, c in recv.items()}


class StripPartition:
    """Strip r owns the squares with column index in [r n, (r+1) n) of a (world n) x n crossed grid
    on [0, world] x [0, 1] and the nodes with x in (r, r+1] (x = 0 included for r = 0).  Its local
    mesh also holds the 3 n ghost triangles of strip r+1 that touch its right interface nodes;
    after the local sweep their rows are sent to rank r+1 and added there."""

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
        self.send, self.recv = {}, {}
        if ghost:
            sel = order[(gi == n)[order]]
            self.send_cells = sel.astype(np.int64)
            self.send[rank + 1] = self.send_cells
        if rank > 0:
            sel = order[((gi == 0) & (gt != 1))[order]]
            self.recv_cells = sel.astype(np.int64)
            self.recv[rank - 1] = self.recv_cells

    def facet_types(self, nrhs: int = 1):
        """Homogeneous Dirichlet on every boundary facet of the local mesh (the artificial
        ones only touch nodes this rank does not own)."""
        ft = np.zeros((nrhs, self.mesh.nfacets), dtype=np.int8)
        ft[:, self.mesh.boundary_facets()] = 1
        return ft

    def patch_cells_per_bin(self):
        return _patch_cells_per_bin(self.mesh, self.node_mask)

    def halo_bytes(self, width, nrhs=1):
        """(bytes sent, bytes received) by this rank in one reverse halo of rows of `width` doubles."""
        return 8 * width * nrhs * len(self.send_cells), 8 * width * nrhs * len(self.recv_cells)

    def conforming_halo(self, k):
        return _conforming_halo(self.mesh, k, self.send, self.recv)


class _StreamJoin:
    """Request object of the RCCL transport: wait() orders the current stream behind the transfer stream."""

    def __init__(self, side):
        self.side = side

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_stream(self.side)
        return True


class HaloExchange:
    """Reverse halo reduction: rows (RT DOFs of ghost cells, `width` values per entry) or scalar
    conforming DOFs (width = 1) are gathered, sent to their owner, added there and cleared here.

    part: Partition / StripPartition (rows of the ghost cells), or explicit lists through `lists` =
    (send, recv) dicts of index arrays per peer with `nentries` = number of rows of the vector
    (EV: the conforming DOF lists of `part.conforming_halo(k)`)."""

    def __init__(self, part, width: int, device, nrhs: int = 1, lists=None, nentries=None, comm=None):
        """comm: None - transport = torch.distributed point-to-point (isend / irecv; nccl backend = RCCL, or
        gloo on CPU tensors); a cpp.RcclComm (or raw ncclComm_t) - transport = the library's own grouped
        ncclSend / ncclRecv (eqlb_halo_reduce / eqlb_halo_exchange), device tensors only."""
        import torch
        self.part, self.nrt, self.nrhs = part, width, nrhs
        self.comm = comm
        self._plan = None
        self._side = None
        self.rank = part.rank
        send, recv = (part.send, part.recv) if lists is None else lists
        self.nentries = part.mesh.ncells if nentries is None else int(nentries)
        self.peers = sorted(set(send) | set(recv))
        self.send_idx, self.recv_idx, self.send_buf, self.recv_buf = {}, {}, {}, {}
        for q in self.peers:
            if q in send:
                s = send[q]
                self.send_idx[q] = torch.from_numpy(np.ascontiguousarray(s, dtype=np.int64)).to(device)
                self.send_buf[q] = torch.zeros((nrhs, len(s), width), dtype=torch.float64, device=device)
            if q in recv:
                r = recv[q]
                self.recv_idx[q] = torch.from_numpy(np.ascontiguousarray(r, dtype=np.int64)).to(device)
                self.recv_buf[q] = torch.zeros((nrhs, len(r), width), dtype=torch.float64, device=device)
        self._ops = None

    def reduce(self, x):
        """x: [nrhs * nentries * width] tensor; adds the neighbours' partial sums to the owned rows
        and clears the ghost rows (their content now lives on the owner).  Device tensors use the
        two halo kernels of libeqlb_amd.so (one launch per peer and direction) around the RCCL
        send/recv; CPU tensors (gloo tests) use plain indexing."""
        return self.finish(x, self.start(x))

    def start(self, x):
        """First half: pack the ghost rows (final once the tiles that own them have run) and post
        the sends / receives.  Work enqueued on the current stream after this call overlaps the
        transfer; finish() orders the unpack behind it."""
        import torch
        import torch.distributed as dist
        xv = x.view(self.nrhs, self.nentries, self.nrt)
        on_gpu = x.is_cuda
        stream = torch.cuda.current_stream().cuda_stream if on_gpu else 0
        if self.comm is not None:
            # the library's transport (eqlb_halo_exchange): pack kernels and the grouped ncclSend / ncclRecv go to
            # a SIDE stream that waits for what the current stream holds now (the sweep of the tiles that own the
            # ghost rows); launches enqueued on the current stream after this call overlap the transfer,
            # finish() makes the current stream wait for the side stream before it unpacks
            from . import cpp
            if not on_gpu:
                raise RuntimeError("HaloExchange: the RCCL transport needs device tensors")
            if self._side is None:
                self._side = torch.cuda.Stream(device=x.device)
            self._side.wait_stream(torch.cuda.current_stream())
            side = self._side.cuda_stream
            for q, idx in self.send_idx.items():
                cpp.halo_pack(x.data_ptr(), idx.data_ptr(), self.send_buf[q].data_ptr(), self.nrhs, idx.numel(),
                              self.nrt, self.nentries, True, side)
            cpp.halo_exchange(self.comm, self._rccl_plan(), self.nrhs, self.nrt, side)
            return [_StreamJoin(self._side)]
        for q, idx in self.send_idx.items():
            buf = self.send_buf[q]
            if on_gpu:
                from . import cpp
                cpp.halo_pack(x.data_ptr(), idx.data_ptr(), buf.data_ptr(), self.nrhs, idx.numel(), self.nrt,
                              self.nentries, True, stream)
            else:
                buf.copy_(xv[:, idx, :])
                xv[:, idx, :] = 0.0
        if self._ops is None:  # the descriptors are reused: same buffers, same peers every step
            ops = []
            for q in self.peers:
                if q in self.send_buf:
                    ops.append(dist.P2POp(dist.isend, self.send_buf[q], q))
                if q in self.recv_buf:
                    ops.append(dist.P2POp(dist.irecv, self.recv_buf[q], q))
            self._ops = ops
        return dist.batch_isend_irecv(self._ops) if self._ops else []

    def _rccl_plan(self):
        if self._plan is None:
            from . import cpp
            z = lambda d, q: d[q].data_ptr() if q in d else 0  # noqa: E731
            n = lambda d, q: d[q].numel() if q in d else 0  # noqa: E731
            self._plan = cpp.HaloPlan(self.peers, [z(self.send_idx, q) for q in self.peers],
                                      [n(self.send_idx, q) for q in self.peers],
                                      [z(self.send_buf, q) for q in self.peers],
                                      [z(self.recv_idx, q) for q in self.peers],
                                      [n(self.recv_idx, q) for q in self.peers],
                                      [z(self.recv_buf, q) for q in self.peers])
        return self._plan

    def reduce_rccl(self, x):
        """The whole reduction in ONE library call (eqlb_halo_reduce) on the current stream."""
        import torch
        from . import cpp
        cpp.halo_reduce(self.comm, self._rccl_plan(), x.data_ptr(), self.nrhs, self.nrt, self.nentries,
                        torch.cuda.current_stream().cuda_stream)
        return x

    def finish(self, x, reqs):
        """Second half: wait for the transfer (the current stream waits, not the host) and add the
        received partial sums to the owned rows."""
        import torch
        on_gpu = x.is_cuda
        for req in reqs:
            req.wait()
        for q, idx in self.recv_idx.items():
            buf = self.recv_buf[q]
            if on_gpu:
                from . import cpp
                stream = torch.cuda.current_stream().cuda_stream
                cpp.halo_unpack_add(x.data_ptr(), idx.data_ptr(), buf.data_ptr(), self.nrhs, idx.numel(),
                                    self.nrt, self.nentries, stream)
            else:
                xv = x.view(self.nrhs, self.nentries, self.nrt)
                xv[:, idx, :] += buf
        return x
