"""Flat triangle-mesh container (the arrays the C ABI takes) and unit-square generators.

The reference obtains these from DOLFINx (`FluxEquilibrator.initialise_mesh_info`,
python/dolfinx_eqlb/eqlb/FluxEquilibrator.py:52-67: entity permutations and the
connectivities 0<->1, 0<->2, 1<->2).  Here they are plain int32 arrays:

  x            [nnodes, 3]  f64   node coordinates (z padded)
  cell_nodes   [ncells, 3]  i32   local vertex order is arbitrary (may be reflected)
  cell_facets  [ncells, 3]  i32   local facet f is opposite local vertex f
  facet_nodes  [nfacets, 2] i32   low global node first
  facet_cells  CSR                ascending cell index (1 entry on the boundary)
  node_cells   CSR                ascending
  node_facets  CSR                ascending
  facet_perm   [ncells, 3]  u8    1 iff the cell traverses the facet (low local vertex ->
                                  high local vertex) against its global low->high direction;
                                  plays the role of DOLFINx' facet permutation info
                                  (se/solve_patch_semiexplt.hpp:334-388)
"""

from dataclasses import dataclass

import numpy as np

_FACET_VERTS = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32)


@dataclass
class Mesh:
    x: np.ndarray
    cell_nodes: np.ndarray
    cell_facets: np.ndarray
    facet_nodes: np.ndarray
    facet_cells_offsets: np.ndarray
    facet_cells: np.ndarray
    node_cells_offsets: np.ndarray
    node_cells: np.ndarray
    node_facets_offsets: np.ndarray
    node_facets: np.ndarray
    facet_perm: np.ndarray

    @property
    def nnodes(self):
        return self.x.shape[0]

    @property
    def ncells(self):
        return self.cell_nodes.shape[0]

    @property
    def nfacets(self):
        return self.facet_nodes.shape[0]

    def boundary_facets(self):
        cnt = np.diff(self.facet_cells_offsets)
        return np.nonzero(cnt == 1)[0].astype(np.int32)

    def facet_midpoints(self):
        return 0.5 * (self.x[self.facet_nodes[:, 0], :2] + self.x[self.facet_nodes[:, 1], :2])


def _csr_from_pairs(rows, cols, nrows):
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    offsets = np.zeros(nrows + 1, dtype=np.int32)
    np.add.at(offsets, rows + 1, 1)
    return np.cumsum(offsets, dtype=np.int32), cols.astype(np.int32)


def create_mesh(x2d: np.ndarray, cell_nodes: np.ndarray) -> Mesh:
    """Build all connectivities from coordinates and cell->node (any local order)."""
    cell_nodes = np.ascontiguousarray(cell_nodes, dtype=np.int32)
    ncells = cell_nodes.shape[0]
    nnodes = x2d.shape[0]
    x = np.zeros((nnodes, 3))
    x[:, :2] = x2d[:, :2]

    # facets = unique edges; local facet f is opposite local vertex f
    ea = cell_nodes[:, _FACET_VERTS[:, 0]]  # [ncells, 3] first local vertex of facet f
    eb = cell_nodes[:, _FACET_VERTS[:, 1]]
    lo = np.minimum(ea, eb).astype(np.int64)
    hi = np.maximum(ea, eb).astype(np.int64)
    keys = (lo * nnodes + hi).ravel()
    ukeys, inv = np.unique(keys, return_inverse=True)
    nfacets = ukeys.size
    cell_facets = inv.reshape(ncells, 3).astype(np.int32)
    facet_nodes = np.stack([ukeys // nnodes, ukeys % nnodes], axis=1).astype(np.int32)
    facet_perm = (ea > eb).astype(np.uint8)

    cells_rep = np.repeat(np.arange(ncells, dtype=np.int64), 3)
    fc_off, fc = _csr_from_pairs(cell_facets.ravel().astype(np.int64), cells_rep, nfacets)
    nc_off, nc = _csr_from_pairs(cell_nodes.ravel().astype(np.int64), cells_rep, nnodes)
    fr = np.repeat(np.arange(nfacets, dtype=np.int64), 2)
    nf_off, nf = _csr_from_pairs(facet_nodes.ravel().astype(np.int64), fr, nnodes)

    return Mesh(x, cell_nodes, cell_facets, facet_nodes, fc_off, fc, nc_off, nc, nf_off, nf,
                facet_perm)


def _shuffle_local_order(cell_nodes, seed):
    """Per-cell random permutation of the local vertex order (rotations AND reflections), so
    that about half of the facets are 'reversed' and about half of the cells have detJ < 0."""
    rng = np.random.default_rng(seed)
    perms = np.array([[0, 1, 2], [1, 2, 0], [2, 0, 1], [0, 2, 1], [2, 1, 0], [1, 0, 2]])
    pick = rng.integers(0, 6, size=cell_nodes.shape[0])
    return np.take_along_axis(cell_nodes, perms[pick], axis=1)


def create_rectangle(nx: int, ny: int, x0: float = 0.0, x1: float = 1.0, y0: float = 0.0,
                     y1: float = 1.0, diagonal: str = "crossed", shuffle_seed=None,
                     perturb: float = 0.0, perturb_seed: int = 7, keep_cell=None,
                     return_grid_ids: bool = False):
    """Structured triangulation of [x0,x1] x [y0,y1] with nx x ny squares.

    crossed: 4 triangles per square in the order (v00,v10,c), (v10,v11,c), (v11,v01,c),
    (v01,v00,c) (bottom, right, top, left), cell id (j*nx + i)*4 + t; right: 2 per square.
    keep_cell: optional callable(i, j, t) -> bool mask to drop cells (used for ghost layers).
    """
    ii, jj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="xy")
    corners = np.stack([x0 + (x1 - x0) * ii.ravel() / nx, y0 + (y1 - y0) * jj.ravel() / ny], axis=1)
    ci, cj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    ci, cj = ci.ravel(), cj.ravel()
    v00 = cj * (nx + 1) + ci
    v10 = v00 + 1
    v01 = v00 + (nx + 1)
    v11 = v01 + 1
    if diagonal == "crossed":
        centres = np.stack([x0 + (x1 - x0) * (ci + 0.5) / nx, y0 + (y1 - y0) * (cj + 0.5) / ny], axis=1)
        c = (nx + 1) * (ny + 1) + cj * nx + ci
        x2d = np.concatenate([corners, centres])
        cells = np.stack([np.stack([v00, v10, c], 1), np.stack([v10, v11, c], 1),
                          np.stack([v11, v01, c], 1), np.stack([v01, v00, c], 1)], axis=1)
        nt = 4
    elif diagonal == "right":
        x2d = corners
        cells = np.stack([np.stack([v00, v10, v11], 1), np.stack([v00, v11, v01], 1)], axis=1)
        nt = 2
    else:
        raise ValueError("diagonal must be 'crossed' or 'right'")
    cells = cells.reshape(-1, 3)
    gi = np.repeat(ci, nt)
    gj = np.repeat(cj, nt)
    gt = np.tile(np.arange(nt), ci.size)
    if keep_cell is not None:
        keep = keep_cell(gi, gj, gt)
        cells, gi, gj, gt = cells[keep], gi[keep], gj[keep], gt[keep]

    if perturb > 0.0:
        rng = np.random.default_rng(perturb_seed)
        eps = 1e-12
        interior = (x2d[:, 0] > x0 + eps) & (x2d[:, 0] < x1 - eps) \
            & (x2d[:, 1] > y0 + eps) & (x2d[:, 1] < y1 - eps)
        x2d = x2d.copy()
        h = min((x1 - x0) / nx, (y1 - y0) / ny)
        x2d[interior] += perturb * h * (rng.random((interior.sum(), 2)) - 0.5)

    if shuffle_seed is not None:
        cells = _shuffle_local_order(cells, shuffle_seed)
    mesh = create_mesh(x2d, cells.astype(np.int32))
    if return_grid_ids:
        return mesh, (gi, gj, gt)
    return mesh


def create_unit_square(n: int, diagonal: str = "crossed", shuffle_seed=None,
                       perturb: float = 0.0, perturb_seed: int = 7) -> Mesh:
    """Unit square, n x n squares, `crossed` (4 triangles per square, as
    python/test/performance/perftest.py:62-73) or `right` diagonals.

    shuffle_seed: None -> canonical counter-clockwise cells; int -> random local vertex order.
    perturb:      relative random displacement of interior nodes (irregular geometry).
    """
    return create_rectangle(n, n, diagonal=diagonal, shuffle_seed=shuffle_seed, perturb=perturb,
                            perturb_seed=perturb_seed)


def create_disk(nsectors: int, nrings: int, shuffle_seed=None, radius: float = 1.0) -> Mesh:
    """Polar triangulation of a disk: a centre node of valence `nsectors` and `nrings` rings of
    `nsectors` nodes each (ring-to-ring quads split into two triangles).  Irregular-valence test
    mesh (adaptive / gmsh meshes of the reference's demos have such nodes); the boundary nodes
    have 3-cell patches."""
    ang = 2.0 * np.pi * np.arange(nsectors) / nsectors
    pts = [np.zeros((1, 2))]
    for r in range(1, nrings + 1):
        rad = radius * r / nrings
        # alternate rings are rotated by half a sector for better-shaped triangles
        off = 0.5 * (2.0 * np.pi / nsectors) * ((r - 1) % 2)
        pts.append(rad * np.stack([np.cos(ang + off), np.sin(ang + off)], axis=1))
    x = np.concatenate(pts)

    def nid(r, i):
        return 1 + (r - 1) * nsectors + (i % nsectors)

    cells = []
    for i in range(nsectors):
        cells.append([0, nid(1, i), nid(1, i + 1)])
    for r in range(1, nrings):
        for i in range(nsectors):
            a, b = nid(r, i), nid(r, i + 1)
            if (r - 1) % 2 == 0:   # outer ring rotated forward
                c, d = nid(r + 1, i), nid(r + 1, i + 1)
                cells.append([a, c, b])
                cells.append([b, c, d])
            else:
                c, d = nid(r + 1, i), nid(r + 1, i + 1)
                cells.append([a, d, b])
                cells.append([a, c, d])
    cells = np.array(cells, dtype=np.int32)
    if shuffle_seed is not None:
        cells = _shuffle_local_order(cells, shuffle_seed)
    return create_mesh(x, cells)
