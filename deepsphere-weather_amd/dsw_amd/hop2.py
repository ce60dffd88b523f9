"""Tile plans for the fused two-hop SpMM kernel (``csrc/dsw_spmm2.hip``).

Two consecutive applications of the same sparse operator (``T_1 = L x`` then ``T_2 = 2 L T_1 - x``,
or two steps of the adjoint recurrence) are fused into one launch so that the intermediate never
makes a round trip through HBM.  A workgroup owns a tile of ``R`` consecutive rows of one sample
and needs

* the first-hop result on ``S1`` = tile rows + every column they reference (the 1-ring), and hence
* the input on ``S2`` = ``S1`` + every column the rows of ``S1`` reference (the 2-ring).

This module computes, once per operator (host side, numpy), for every tile: the gather list of
``S2`` (global row ids, ``S1`` first, tile rows first of all) and the CSR of the ``S1`` rows with
column indices rewritten as *positions in that list* (uint16).  Rows ``[0, R_t)`` of that local CSR
are the tile rows, and all their columns lie in ``S1``, so the same local CSR drives both hops.

In HEALPix nested order a 128-row tile is an 8x16 pixel patch: |S1| ~ 180, |S2| ~ 240 for the
8-neighbour stencil.  Nothing here is specific to HEALPix: any square CSR operator works, the
halo sizes just grow with worse locality (the builder reports them and the caller falls back to
two single-hop launches when they do not fit LDS).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch


class Hop2PlanStruct(ctypes.Structure):
    """Mirror of ``dsw_hop2_plan`` in ``include/dsw_hip.h`` (device pointers + sizes)."""

    _fields_ = [
        ("n_tiles", ctypes.c_int32),
        ("tile_rows", ctypes.c_int32),
        ("max_n1", ctypes.c_int32),
        ("max_n2", ctypes.c_int32),
        ("max_nnz", ctypes.c_int32),
        ("reserved", ctypes.c_int32),      # longest local CSR row (ELL width before rounding to 4)
        ("tile_meta", ctypes.c_void_p),   # int32 [n_tiles][6]: s2_off, n1, n2, nnz_off, rp_off, 0
        ("s2_rows", ctypes.c_void_p),     # int32, concatenated gather lists
        ("lrowptr", ctypes.c_void_p),     # int32, concatenated, (n1 + 1) per tile, tile-relative
        ("lcol", ctypes.c_void_p),        # uint16, concatenated
        ("lval", ctypes.c_void_p),        # float32, concatenated
        ("explicit_tiles", ctypes.c_int32),   # 1: tile rows = the first tile_meta[t][5] entries of the gather list
        ("hops", ctypes.c_int32),             # 2: fused two-hop kernel; 1: staged one-hop kernel (csrc/dsw_spmm1s.hip)
        ("ell_w", ctypes.c_int32),            # > 0: the tile rows' stencils also as padded ELL (64 rows x ell_w per tile)
        ("ell_pos", ctypes.c_void_p),         # uint16 [n_tiles][64][ell_w]: list positions (padding: the row itself)
        ("ell_val", ctypes.c_void_p),         # float32 [n_tiles][64][ell_w] (padding: 0)
        ("ell2", ctypes.c_void_p),            # two-hop plans: per tile [max_n1][W] fp32 values + [max_n1][W] u8 positions (LDS image)
        ("ell2_stride", ctypes.c_int64),      # bytes between the images of consecutive tiles
        ("struct_bytes", ctypes.c_int64),     # sizeof(dsw_hop2_plan) as this binding declares it (checked by every entry point)
    ]


class Hop2Plan:
    def __init__(self, tile_rows, tile_meta, s2_rows, lrowptr, lcol, lval, max_n1, max_n2, max_nnz, n_rows,
                 max_row_len=0, explicit_tiles=False, hops=2):
        self.hops = int(hops)
        self.ell_w = 0
        self.ell_pos = self.ell_val = None
        self.ell2 = None
        self.ell2_stride = 0
        self.max_row_len = int(max_row_len)
        self.explicit_tiles = bool(explicit_tiles)
        self.tile_rows = int(tile_rows)
        self.tile_meta = tile_meta
        self.s2_rows = s2_rows
        self.lrowptr = lrowptr
        self.lcol = lcol
        self.lval = lval
        self.max_n1, self.max_n2, self.max_nnz = int(max_n1), int(max_n2), int(max_nnz)
        self.n_tiles = int(tile_meta.shape[0])
        self.n_rows = int(n_rows)
        self._struct = None
        self._dev = None

    def lds_bytes(self, row_bytes: int, single_buf: bool = False) -> int:
        """LDS the kernel carves (must match hop2_lds_bytes in csrc/dsw_spmm2.hip): the input rows on
        S2 (two buffers unless ``single_buf``), the first-hop rows on S1, {col, val} pairs, the gather list."""
        ell_w = (self.max_row_len + 3) & ~3
        if self.hops == 1:      # hop1_lds_bytes / hop1_dma_lds_bytes in csrc/dsw_spmm1s.hip
            scratch = self.max_n1 * ell_w * 6 + ((self.max_n2 + 3) & ~3) * 4 + 16 + (self.max_n1 + 1) * 4
            buf = self.max_n2 * row_bytes
            dma = self.ell_w > 0 and self.tile_rows * row_bytes <= 512 * 16 and (row_bytes // 16) & (row_bytes // 16 - 1) == 0 \
                and self.max_n2 * row_bytes <= 3 * 512 * 16
            # LDS-DMA form: ring of three buffers of staged rows + two 8 KiB buffers of epilogue rows (stencils from the ELL image);
            # generic form: one buffer + the scratch
            return ((3 * buf + 2 * 512 * 16 if dma else buf + scratch) + 15) & ~15
        s = (self.max_n1 + (1 if single_buf else 2) * self.max_n2) * row_bytes   # bufT + input rows
        s += self.max_n1 * ell_w * 6                      # ELL: fp32 values + u16 list positions
        s += ((self.max_n2 + 3) & ~3) * 4 + 16            # gather list + the tile's loop length
        return (s + 15) & ~15

    ELL_ROWS, ELL_WS = 64, (24, 32)   # shapes the LDS-DMA kernel keeps in registers: one row per lane group, 24 / 32 entries

    def add_ell(self):
        """Padded ELL image of the tile rows' stencils (hops = 1 plans with <= 64 rows per tile and <= 24 entries per
        row - 32 when a few rows are longer): what the LDS-DMA kernel loads straight into registers - 9 independent 16-byte loads per lane instead of a
        CSR -> ELL expansion through LDS with two workgroup barriers in every workgroup's prologue.  tile_meta[t][5]
        becomes the tile's longest row (the kernel's gather length; explicit tiles keep their row count in [1])."""
        if self.hops != 1 or self.tile_rows > self.ELL_ROWS or self.max_row_len > self.ELL_WS[-1] or self.max_n1 > self.ELL_ROWS:
            return self
        R, W = self.ELL_ROWS, min(w for w in self.ELL_WS if w >= self.max_row_len)
        pos = np.zeros((self.n_tiles, R, W), dtype=np.uint16)
        pos[:] = np.arange(R, dtype=np.uint16)[None, :, None]          # padding: the row itself, weight 0
        val = np.zeros((self.n_tiles, R, W), dtype=np.float32)
        meta = self.tile_meta.copy()
        for t in range(self.n_tiles):
            s2_off, rt, n2, nnz_off, rp_off, _ = (int(v) for v in self.tile_meta[t])
            lrp = self.lrowptr[rp_off:rp_off + rt + 1].astype(np.int64)
            lens = np.diff(lrp)
            rid = np.repeat(np.arange(rt), lens)
            col = np.arange(int(lrp[-1])) - np.repeat(lrp[:-1], lens)
            pos[t, rid, col] = self.lcol[nnz_off:nnz_off + lrp[-1]]
            val[t, rid, col] = self.lval[nnz_off:nnz_off + lrp[-1]]
            meta[t, 5] = int(lens.max()) if lens.size else 0
        self.tile_meta, self.ell_pos, self.ell_val, self.ell_w = meta, pos.reshape(-1), val.reshape(-1), W
        return self

    def add_ell2(self):
        """Two-hop plans on consecutive tiles: the ELL image of every tile exactly as the one-launch kernels (csrc/dsw_fwd3.hip,
        csrc/dsw_bwd3d.hip) hold it in LDS - [max_n1][W] fp32 values, then [max_n1][W] u8 list positions, padding {own row, 0} -
        so that their prologue copies it in one round of loads; tile_meta[t][5] becomes the tile's longest row (at least 2,
        as the kernels' own expansion computes it)."""
        if self.hops == 1 or self.explicit_tiles or self.max_n2 > 255 or self.max_row_len <= 0:
            return self
        W = (self.max_row_len + 3) & ~3
        vbytes = self.max_n1 * W * 4
        stride = (vbytes + self.max_n1 * W + 15) & ~15
        img = np.zeros((self.n_tiles, stride), dtype=np.uint8)
        meta = self.tile_meta.copy()
        for t in range(self.n_tiles):
            s2_off, n1, n2, nnz_off, rp_off, _ = (int(v) for v in self.tile_meta[t])
            lrp = self.lrowptr[rp_off:rp_off + n1 + 1].astype(np.int64)
            lens = np.diff(lrp)
            val = np.zeros((self.max_n1, W), dtype=np.float32)
            pos = np.zeros((self.max_n1, W), dtype=np.uint8)
            pos[:] = np.arange(self.max_n1, dtype=np.int64)[:, None].astype(np.uint8)      # padding: the row itself, weight 0
            rid = np.repeat(np.arange(n1), lens)
            col = np.arange(int(lrp[-1])) - np.repeat(lrp[:-1], lens)
            pos[rid, col] = self.lcol[nnz_off:nnz_off + lrp[-1]].astype(np.uint8)
            val[rid, col] = self.lval[nnz_off:nnz_off + lrp[-1]]
            img[t, :vbytes] = val.reshape(-1).view(np.uint8)
            img[t, vbytes:vbytes + self.max_n1 * W] = pos.reshape(-1)
            meta[t, 5] = max(2, int(lens.max()) if lens.size else 0)
        self.tile_meta, self.ell2, self.ell2_stride = meta, img.reshape(-1), int(stride)
        return self

    def gather_passes_per_row(self, slots: int = 64) -> float:
        """Passes of ``slots`` row slots the kernel spends per output row (first hop on tile + 1-ring, second hop on the
        tile) - the cost figure tile heights are compared by."""
        n1 = self.tile_meta[:, 1].astype(np.int64)
        rt = self.tile_meta[:, 1].astype(np.int64) if self.hops == 1 else self.tile_meta[:, 5].astype(np.int64) if self.explicit_tiles else \
            np.minimum(self.tile_rows, self.n_rows - np.arange(self.n_tiles) * self.tile_rows)
        if self.hops == 1:
            return float((-(-rt // slots)).sum()) / float(rt.sum())
        return float((-(-n1 // slots) + -(-rt // slots)).sum()) / float(rt.sum())

    def to(self, device):
        """Device-resident copy (cached) + the ctypes struct handed to the C ABI."""
        if self._dev is not None and self._dev[0] == device:
            return self
        arrs = {}
        for name in ("tile_meta", "s2_rows", "lrowptr", "lcol", "lval") + (("ell_pos", "ell_val") if self.ell_w else ()) + \
                (("ell2",) if self.ell2 is not None else ()):
            a = getattr(self, name)
            t = torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a)
            arrs[name] = t.to(device)
        st = Hop2PlanStruct(
            self.n_tiles, self.tile_rows, self.max_n1, self.max_n2, self.max_nnz, self.max_row_len,
            arrs["tile_meta"].data_ptr(), arrs["s2_rows"].data_ptr(), arrs["lrowptr"].data_ptr(),
            arrs["lcol"].data_ptr(), arrs["lval"].data_ptr(), 1 if self.explicit_tiles else 0, self.hops,
            self.ell_w, arrs["ell_pos"].data_ptr() if self.ell_w else None, arrs["ell_val"].data_ptr() if self.ell_w else None,
            arrs["ell2"].data_ptr() if self.ell2 is not None else None, self.ell2_stride,
            ctypes.sizeof(Hop2PlanStruct),
        )
        self._dev = (device, arrs)   # keeps the device tensors alive
        self._struct = st
        return self

    @property
    def struct_ptr(self):
        return ctypes.byref(self._struct)


def _ring_sizes(rowptr, colind, tile, mark):
    """(|S1|, |S2|) of a row set: the set plus its 1-ring, plus its 2-ring (``mark``: all-False scratch, restored)."""
    def nbrs(rows):
        starts, lens = rowptr[rows], rowptr[rows + 1] - rowptr[rows]
        idx = np.repeat(starts - np.cumsum(np.concatenate([[0], lens[:-1]])), lens) + np.arange(int(lens.sum()))
        return np.unique(colind[idx])
    mark[tile] = True
    c1 = nbrs(tile)
    h1 = c1[~mark[c1]]
    mark[h1] = True
    s1 = np.concatenate([tile, h1])
    c2 = nbrs(s1)
    n2 = s1.size + int((~mark[c2]).sum())
    mark[s1] = False
    return s1.size, n2


def cluster_tiles(rowptr: np.ndarray, colind: np.ndarray, tile_rows: int, max_n1: int = 0, max_n2: int = 0):
    """Partition the rows of a square operator into compact tiles of <= ``tile_rows`` rows by greedy breadth-first growth
    over its graph (no coordinates needed): the seed is the lowest unassigned row, every level adds the unassigned
    neighbours of the previous one; the search front also walks through rows that are already taken, so that the slivers
    left between earlier tiles are swept up instead of becoming one-row tiles.  Seeds advance in row order, i.e.
    consecutive tiles stay neighbours (what the XCD-aware launch order wants).  Equiangular 200 x 400, k = 20: 1260 tiles
    for 80 000 rows (ideal 1250), 1-/2-ring sizes 152 / 267 on average - a row-major strip of 64 rows has 340 / 648.
    ``max_n1`` / ``max_n2`` (> 0): a tile whose tile + 1-ring / + 2-ring exceeds them keeps only the longest prefix of its
    growth order that fits (the inner, compact part; found by bisection) and returns the rest to the pool - the kernel
    sizes its LDS by the LARGEST neighbourhood of the plan, and a few ragged tiles (poles, swept-up slivers) would
    otherwise cost every workgroup its second CU slot.  (Halving such tiles instead left 2205 tiles of 36 rows on the
    equiangular graph at height 64; trimming leaves 1361 of 59.)"""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    n = rowptr.shape[0] - 1
    assigned = np.zeros(n, dtype=bool)
    seen = np.zeros(n, dtype=np.int64)
    mark = np.zeros(n, dtype=bool)
    tiles = []
    capped = max_n1 > 0 or max_n2 > 0

    def fits(tile):
        n1, n2 = _ring_sizes(rowptr, colind, tile, mark)
        return (max_n1 <= 0 or n1 <= max_n1) and (max_n2 <= 0 or n2 <= max_n2)

    seed = 0
    stamp = 0
    while True:
        while seed < n and assigned[seed]:
            seed += 1
        if seed >= n:
            break
        stamp += 1
        parts = [np.array([seed], dtype=np.int64)]
        count = 1
        assigned[seed] = True
        seen[seed] = stamp
        frontier = parts[0]
        levels = 0
        while count < tile_rows and frontier.size and levels < 24:
            starts, lens = rowptr[frontier], rowptr[frontier + 1] - rowptr[frontier]
            idx = np.repeat(starts - np.cumsum(np.concatenate([[0], lens[:-1]])), lens) + np.arange(int(lens.sum()))
            nb = np.unique(colind[idx])
            nb = nb[seen[nb] != stamp]
            seen[nb] = stamp
            take = nb[~assigned[nb]][:tile_rows - count]
            if take.size:
                assigned[take] = True
                parts.append(take)
                count += take.size
            frontier = nb
            levels += 1
        tile = np.concatenate(parts)
        if capped and tile.size > 8 and not fits(tile):
            lo, hi = 8, tile.size - 1
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if fits(tile[:mid]):
                    lo = mid
                else:
                    hi = mid - 1
            back = tile[lo:]
            assigned[back] = False
            seed = min(seed, int(back.min()))
            tile = tile[:lo]
        tiles.append(tile)
    return tiles


def build_hop2_plan(rowptr: np.ndarray, colind: np.ndarray, values: np.ndarray, tile_rows: int, tiles=None,
                    hops: int = 2) -> Hop2Plan:
    """Plan for a square CSR operator (int32 rowptr/colind, fp32 values) and a tile size.  ``tiles``: explicit row sets
    (a partition of the rows, each of <= ``tile_rows`` rows, e.g. from ``cluster_tiles``); default: consecutive rows.
    ``hops = 1``: plan of the staged ONE-hop kernel - the local CSR covers the tile rows only (``n1`` = tile rows) and
    the gather list ends with their 1-ring."""
    if hops not in (1, 2):
        raise ValueError("hops must be 1 or 2")
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colind = np.asarray(colind, dtype=np.int64)
    values = np.asarray(values, dtype=np.float32)
    n = rowptr.shape[0] - 1
    explicit = tiles is not None
    if explicit:
        sizes = np.array([len(t) for t in tiles], dtype=np.int64)
        cover = np.concatenate([np.asarray(t, dtype=np.int64) for t in tiles]) if len(tiles) else np.zeros(0, np.int64)
        if sizes.size == 0 or sizes.max() > tile_rows or sizes.min() < 1 or cover.size != n or \
                not np.array_equal(np.sort(cover), np.arange(n)):
            raise ValueError("tiles must partition the rows into sets of 1..tile_rows rows")
    n_tiles = len(tiles) if explicit else (n + tile_rows - 1) // tile_rows
    meta = np.zeros((n_tiles, 6), dtype=np.int32)
    s2_chunks, rp_chunks, col_chunks, val_chunks = [], [], [], []
    s2_off = nnz_off = rp_off = 0
    max_n1 = max_n2 = max_nnz = max_len = 0
    pos = np.full(n, -1, dtype=np.int64)   # scratch: global row -> position in the current gather list
    for t in range(n_tiles):
        if explicit:
            tile = np.asarray(tiles[t], dtype=np.int64)
            starts_t, lens_t = rowptr[tile], rowptr[tile + 1] - rowptr[tile]
            idx_t = np.repeat(starts_t - np.cumsum(np.concatenate([[0], lens_t[:-1]])), lens_t) + np.arange(int(lens_t.sum()))
            c1 = np.unique(colind[idx_t])
            pos[tile] = 0
            halo1 = c1[pos[c1] < 0]
            pos[tile] = -1
        else:
            r0, r1 = t * tile_rows, min(n, (t + 1) * tile_rows)
            tile = np.arange(r0, r1)
            c1 = np.unique(colind[rowptr[r0]:rowptr[r1]])
            halo1 = c1[(c1 < r0) | (c1 >= r1)]
        s1 = tile if hops == 1 else np.concatenate([tile, halo1])
        # columns referenced by the S1 rows
        starts, ends = rowptr[s1], rowptr[s1 + 1]
        lens = ends - starts
        idx = np.repeat(starts - np.cumsum(np.concatenate([[0], lens[:-1]])), lens) + np.arange(lens.sum())
        cols = colind[idx]
        pos[s1] = np.arange(s1.size)
        new = np.unique(cols[pos[cols] < 0])
        s2 = np.concatenate([s1, new])
        pos[new] = s1.size + np.arange(new.size)
        if s2.size > 65535:
            raise ValueError("tile neighbourhood exceeds 65535 rows; operator too dense for the fused path")
        lcol = pos[cols].astype(np.uint16)
        lrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        order = _bank_friendly_order(lcol, lens)
        lcol, idx = lcol[order], idx[order]
        pos[s2] = -1
        meta[t] = (s2_off, s1.size, s2.size, nnz_off, rp_off, tile.size if explicit else 0)
        s2_chunks.append(s2.astype(np.int32))
        rp_chunks.append(lrp)
        col_chunks.append(lcol)
        val_chunks.append(values[idx])
        s2_off += s2.size
        rp_off += s1.size + 1
        nnz_off += int(lens.sum())
        max_n1, max_n2, max_nnz = max(max_n1, s1.size), max(max_n2, s2.size), max(max_nnz, int(lens.sum()))
        max_len = max(max_len, int(lens.max()) if lens.size else 0)
    return Hop2Plan(
        tile_rows, meta, np.concatenate(s2_chunks), np.concatenate(rp_chunks), np.concatenate(col_chunks),
        np.concatenate(val_chunks), max_n1, max_n2, max_nnz, n, max_len, explicit_tiles=explicit, hops=hops,
    ).add_ell().add_ell2()


def _bank_friendly_order(lcol: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Permutation of one tile's local CSR entries (rows keep their extents) that removes most LDS bank conflicts
    of the gather: a staged row is 128 bytes = one HALF of the 256-byte bank row, picked by the parity of its list
    position, and ``ds_read_b128`` serves, in one cycle, the same 64-byte half of the entries that list rows
    ``8m + {0, 3}`` (and ``{1, 2}``, ``{4, 7}``, ``{5, 6}``) gather at the same step - conflict-free exactly when the
    two positions differ in parity.  So the entries of list row ``i`` alternate even / odd positions, starting with
    parity ``(i >> 1) & 1`` (opposite for the two rows of every pair); the in-kernel padding (own position) has the
    parity of ``i`` and is opposite within a pair as well.  Only the summation order inside a row changes."""
    n1 = lens.shape[0]
    rid = np.repeat(np.arange(n1), lens)
    par = (lcol & 1).astype(np.int64)
    first = par == ((rid >> 1) & 1)
    key = rid * 2 + par
    o = np.argsort(key, kind="stable")
    ks = key[o]
    rank = np.empty_like(rid)
    rank[o] = np.arange(ks.size) - np.searchsorted(ks, ks, side="left")
    slot = 2 * rank + (~first)
    return np.lexsort((slot, rid))


def emulate_hop1(plan: Hop2Plan, U, Z, Z2, a, b, c):
    """numpy restatement of one staged single-hop launch (``hops = 1`` plans): ``Y = a * L U + b * Z + c * Z2``."""
    assert plan.hops == 1
    V, C = U.shape
    y = np.zeros((V, C))
    for t in range(plan.n_tiles):
        s2_off, rt, n2, nnz_off, rp_off, _ = (int(v) for v in plan.tile_meta[t])
        rows = plan.s2_rows[s2_off:s2_off + n2].astype(np.int64)
        lrp = plan.lrowptr[rp_off:rp_off + rt + 1].astype(np.int64)
        lcol = plan.lcol[nnz_off:nnz_off + lrp[-1]].astype(np.int64)
        lval = plan.lval[nnz_off:nnz_off + lrp[-1]].astype(np.float64)
        bufx = U[rows]
        for i in range(rt):
            sl = slice(lrp[i], lrp[i + 1])
            acc = a * (lval[sl, None] * bufx[lcol[sl]]).sum(0)
            if Z is not None:
                acc = acc + b * Z[rows[i]]
            if Z2 is not None:
                acc = acc + c * Z2[rows[i]]
            y[rows[i]] = acc
    return y


def emulate_hop2(plan: Hop2Plan, U, Z1, Z1b, Z2, a1, b1, d1, a2, b2, c2):
    """numpy restatement of what one fused launch computes (used by the CPU tests of the plan):

        Y1 = a1 * L U + b1 * Z1 + d1 * Z1b      (on S1 of every tile; returned on tile rows)
        Y2 = a2 * L Y1 + b2 * U + c2 * Z2       (on tile rows)

    ``U``/``Z*``: ``[V, C]`` float64 arrays or None.  Returns ``(Y1, Y2)`` as ``[V, C]``.
    """
    V, C = U.shape
    y1 = np.zeros((V, C))
    y2 = np.zeros((V, C))
    for t in range(plan.n_tiles):
        s2_off, n1, n2, nnz_off, rp_off, _ = (int(v) for v in plan.tile_meta[t])
        rows = plan.s2_rows[s2_off:s2_off + n2].astype(np.int64)
        lrp = plan.lrowptr[rp_off:rp_off + n1 + 1].astype(np.int64)
        lcol = plan.lcol[nnz_off:nnz_off + lrp[-1]].astype(np.int64)
        lval = plan.lval[nnz_off:nnz_off + lrp[-1]].astype(np.float64)
        bufx = U[rows]
        buft = np.zeros((n1, C))
        for i in range(n1):
            sl = slice(lrp[i], lrp[i + 1])
            buft[i] = a1 * (lval[sl, None] * bufx[lcol[sl]]).sum(0)
        if Z1 is not None:
            buft += b1 * Z1[rows[:n1]]
        if Z1b is not None:
            buft += d1 * Z1b[rows[:n1]]
        rt = int(plan.tile_meta[t][5]) if plan.explicit_tiles else min(plan.tile_rows, V - t * plan.tile_rows)
        for i in range(rt):
            sl = slice(lrp[i], lrp[i + 1])
            acc = a2 * (lval[sl, None] * buft[lcol[sl]]).sum(0) + b2 * bufx[i]
            if Z2 is not None:
                acc = acc + c2 * Z2[rows[i]]
            y2[rows[i]] = acc
        y1[rows[:rt]] = buft[:rt]
    return y1, y2
