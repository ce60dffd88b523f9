"""Host side of the hot path: operator caches and ``torch.autograd.Function`` wrappers that call the
C ABI of ``libdsw_hip.so`` on the current HIP stream.

PyTorch is plumbing here (device memory, streams, autograd graph); the convolutions, poolings, residual
epilogues and their backward passes run in the hand-written gfx950 kernels under ``csrc/``.  What stock
torch still launches inside a model step is glue, not the path: autograd's ``add`` where a tensor has two
consumers and no fused route (a ResBlock's input: conv stack + residual map - the fused alternative,
``dense_mix_fork`` / ``dX_add``, exists and measured slower), the zero-pad of an unaligned input width
(``_padded_width``), the loss.  CPU tensors are rejected: there is no fallback implementation in the product.  (Tests may inject a checker backend with
``set_test_backend`` to exercise the host logic on CPU; nothing in the package does.)
"""
from __future__ import annotations

import weakref

import torch

from . import _native

__all__ = [
    "CsrOperator",
    "get_operator",
    "cheb_conv",
    "sparse_remap",
    "sparse_remap_add",
    "cheb_basis",
    "maxval_pool",
    "maxval_unpool",
    "set_test_backend",
]

_DTYPES = {torch.float32: _native.DSW_F32, torch.bfloat16: _native.DSW_BF16}


# ----------------------------------------------------------------------------------------------
# Operator: device CSR (+ lazily, CSR of the transpose) derived from a torch sparse COO buffer
# ----------------------------------------------------------------------------------------------
class CsrOperator:
    """int32 CSR + fp32 values of a sparse operator, resident on the operator's device.

    Derived *cache* of the module's sparse-COO buffer (which stays the state_dict citizen, as in
    the reference ``modules/layers.py:241,954``).  Values are widened to fp32 whatever the
    buffer dtype is: the kernels keep operator and accumulators in fp32.
    """

    def __init__(self, rowptr, colind, values, shape):
        self.rowptr = rowptr
        self.colind = colind
        self.values = values
        self.shape = (int(shape[0]), int(shape[1]))
        self.nnz = int(colind.numel())
        self._t = None
        self._remap = None        # remap plan (interpolation pooling products), built on first use
        self._plans = {}          # row_bytes -> device plan (or None)
        self._plans_by_rows = {}  # tile_rows -> host plan (or None): built once per tile size, shared by every row size

    @classmethod
    def from_sparse_coo(cls, mat: torch.Tensor) -> "CsrOperator":
        if mat.layout != torch.sparse_coo:
            raise TypeError("expected a torch sparse COO tensor")
        if mat.dim() != 2:
            raise ValueError("operator must be 2-D")
        if not mat.is_coalesced():
            mat = mat.coalesce()
        idx = mat.indices()
        nrow, ncol = mat.shape
        if nrow >= 2**31 or ncol >= 2**31 or idx.shape[1] >= 2**31:
            raise ValueError("operator too large for int32 CSR")
        # coalesced COO is sorted row-major (checked by the golden fixture G4), i.e. CSR-ready
        rowptr = torch._convert_indices_from_coo_to_csr(idx[0], nrow, out_int32=True)
        colind = idx[1].to(torch.int32).contiguous()
        values = mat.values().to(torch.float32).contiguous()
        return cls(rowptr, colind, values, (nrow, ncol))

    @property
    def device(self):
        return self.values.device

    def hop2_plan(self, row_bytes: int):
        """Tile plan of the LDS-staged recurrence kernels for rows of ``row_bytes`` bytes, or ``None`` when the operator
        is not square / the tile neighbourhoods do not fit LDS (built once, cached).  Sparse stencils (HEALPix k = 8: 9
        entries per row) get the plan of the fused TWO-hop kernel; dense ones (the reference's default k = 20 graph,
        equiangular k = 20: 21+ entries) the plan of the staged ONE-hop kernel - there the first-hop redundancy of a fused
        pair costs more than the round trip of the intermediate plane (``HOP_MODE`` overrides the choice for A/B runs)."""
        if self.shape[0] != self.shape[1] or row_bytes % 16 != 0 or not self.values.is_cuda:
            return None
        key = (row_bytes, HOP_MODE, STAGED_TILE_ROWS)
        if key not in self._plans:
            from . import hop2

            plan = None
            host_csr = []

            def host(i):
                if not host_csr:
                    host_csr.extend(t.cpu().numpy() for t in (self.rowptr, self.colind, self.values))
                return host_csr[i]

            def host_plan(rows, clustered=False, hops=2):
                key = (rows, clustered, hops)
                if key not in self._plans_by_rows:
                    try:
                        tiles = None
                        lens = host(0)[1:] - host(0)[:-1]
                        w = (int(lens.max()) + 3) & ~3 if lens.size else 4
                        if clustered and hops == 2:
                            # neighbourhood caps = what two workgroups per CU can stage of 128-byte rows (one input
                            # buffer, ELL of the longest row): n1 * (128 + 6 w) + n2 * 128 <= 80 KiB at n2 ~ 1.8 n1
                            cap1 = int((80 * 1024 - 2048) / (128 + 6 * w + 1.8 * 128))
                            tiles = hop2.cluster_tiles(host(0), host(1), rows, max_n1=cap1, max_n2=int(1.8 * cap1))
                        elif clustered:
                            # one hop: tile + 1-ring staged; LDS-DMA form (stencils of <= 32 entries): a ring of three
                            # buffers of 128-byte rows + 16 KiB of epilogue rows; generic form: one buffer + ELL of the tile
                            cap1 = (80 * 1024 - 16 * 1024) // (3 * 128) if w <= 32 and rows <= 64 else \
                                int((80 * 1024 - 2560 - rows * (6 * w + 4)) / 132)
                            tiles = hop2.cluster_tiles(host(0), host(1), rows, max_n1=cap1)
                        self._plans_by_rows[key] = hop2.build_hop2_plan(host(0), host(1), host(2), rows, tiles=tiles, hops=hops)
                    except ValueError:
                        self._plans_by_rows[key] = None
                return self._plans_by_rows[key]

            dense = self.nnz >= STAGED_MIN_ROW_LEN * self.shape[0]
            if HOP_MODE == "staged" or (HOP_MODE == "auto" and dense):
                # staged one-hop plan (two workgroups per CU): tiles of consecutive rows when those are compact (HEALPix
                # nested order: tile + 1-ring = 2.2-2.4x the tile at k = 20), else tiles clustered from the graph (ring
                # order, equiangular row-major: a strip of 64 consecutive rows drags in 5x its own rows)
                def fits(cand):
                    return cand is not None and cand.lds_bytes(row_bytes) <= 80 * 1024 and cand.max_n2 * row_bytes <= 65535

                def staged_rows(cand):    # rows staged per output row
                    return float(cand.tile_meta[:, 2].sum()) / float(cand.tile_meta[:, 1].sum())

                for rows in STAGED_TILE_ROWS:
                    if rows > self.shape[0]:
                        continue
                    cand = host_plan(rows, False, 1)
                    if fits(cand):
                        plan = cand
                        break
                if (plan is None or staged_rows(plan) > 3.0):
                    for rows in STAGED_TILE_ROWS:
                        if self.shape[0] < MIN_CLUSTERED_TILES * rows:
                            continue
                        cand = host_plan(rows, True, 1)
                        if fits(cand) and (plan is None or staged_rows(cand) < staged_rows(plan)):
                            plan = cand
                            break

            # largest tile whose workgroup still leaves room for >= 2 workgroups per CU (<= 80 KiB of
            # the 160 KiB LDS); a single resident workgroup (<= 156 KiB) is the last resort
            # (the kernel drops to ONE input-row buffer when that is what lets a second workgroup share the CU).
            # Tiles of CONSECUTIVE rows first (HEALPix nested order: a tile is a square patch); when their
            # neighbourhoods are too large - row order that is not 2-D local: equiangular row-major, HEALPix ring
            # order - tiles clustered from the operator's graph.
            for budget, single in ((80 * 1024, False), (80 * 1024, True), (156 * 1024, False), (156 * 1024, True)):
                if plan is not None:
                    break
                for rows in (256, 128, 64):
                    if rows > self.shape[0]:
                        continue
                    cand = host_plan(rows)
                    if cand is not None and cand.lds_bytes(row_bytes, single) <= budget:
                        plan = cand
                        break
                if plan is None:
                    # clustered tiles: of the tile heights that fit, the one with the fewest 64-slot gather passes per
                    # output row (a pass costs the same whether its slots are full or not; a height whose tiles had to
                    # be halved to respect the neighbourhood caps loses to a smaller one)
                    best = None
                    for rows in (64, 56, 48):
                        if self.shape[0] < MIN_CLUSTERED_TILES * rows:
                            continue    # a graph of a few tiles gains nothing from the fused path (a launch of < 8
                                        # workgroups per batch chunk; measured slower than one launch per hop)
                        cand = host_plan(rows, True)
                        if cand is None or cand.lds_bytes(row_bytes, single) > budget:
                            continue
                        cost = cand.gather_passes_per_row()
                        if best is None or cost < best:
                            best, plan = cost, cand
            self._plans[key] = None if plan is None else plan.to(self.device)
        return self._plans[key]

    def remap_plan(self):
        """Plan of this matrix as a REMAP (pooling) operator for ``dsw_remap_csr`` (include/dsw_hip.h: dsw_remap_plan), built
        once: the structure of a regular hierarchy (m entries per row in columns m r .. m r + m - 1, or one entry per row in
        column r / m - the HEALPix nested pooling / unpooling and their transposes) when the matrix has it, else the list of
        its long rows (polar cells of a cross-sampling matrix)."""
        if self._remap is None:
            import ctypes

            class _Plan(ctypes.Structure):
                _fields_ = [("kind", ctypes.c_int32), ("m", ctypes.c_int32), ("long_thr", ctypes.c_int32),
                            ("n_long", ctypes.c_int32), ("long_rows", ctypes.c_void_p), ("parts", ctypes.c_int32),
                            ("reserved", ctypes.c_int32)]

            nrow, ncol = self.shape
            kind, m, thr, long_rows, parts = 0, 0, 0, None, 1
            if self.values.is_cuda and nrow > 0 and self.nnz > 0:
                dev = self.device
                if self.nnz == ncol and ncol % nrow == 0:          # GROUPS: m = ncol / nrow entries per row, identity columns
                    mm = ncol // nrow
                    if torch.equal(self.rowptr, torch.arange(0, self.nnz + 1, mm, device=dev, dtype=torch.int32)) and \
                            torch.equal(self.colind, torch.arange(self.nnz, device=dev, dtype=torch.int32)):
                        kind, m = 1, mm
                if kind == 0 and self.nnz == nrow and nrow % ncol == 0:    # BROADCAST: one entry per row in column r / m
                    mm = nrow // ncol
                    if torch.equal(self.rowptr, torch.arange(nrow + 1, device=dev, dtype=torch.int32)) and \
                            torch.equal(self.colind, torch.arange(nrow, device=dev, dtype=torch.int32) // mm):
                        kind, m = 2, mm
                if kind == 0:
                    lens = self.rowptr[1:] - self.rowptr[:-1]
                    # rows beyond twice the mean length (at least 16 entries) get whole waves: the lane-group-per-row path
                    # then never walks a chain much longer than its average one
                    thr = max(16, int(2.0 * self.nnz / nrow + 0.5))
                    long_rows = torch.nonzero(lens > thr).flatten().to(torch.int32).contiguous()
                    if long_rows.numel() == 0:
                        long_rows = None
                    mean = self.nnz / nrow
                    parts = 4 if mean >= 10 else 2 if mean >= 6 else 1
                    if parts > 1 and long_rows is None and int(lens.max()) > thr:
                        parts = 1
            plan = _Plan(kind, m, thr if long_rows is not None else 0, 0 if long_rows is None else int(long_rows.numel()),
                         None if long_rows is None else long_rows.data_ptr(), parts, 0)
            self._remap = (plan, long_rows)      # the list must outlive the plan
        return self._remap[0]

    def transpose(self) -> "CsrOperator":
        """CSR of the transposed operator (built once, on first backward)."""
        if self._t is None:
            nrow, ncol = self.shape
            counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            rows = torch.repeat_interleave(
                torch.arange(nrow, device=self.device, dtype=torch.int64), counts
            )
            cols = self.colind.to(torch.int64)
            order = torch.argsort(cols * nrow + rows)
            t_rows = cols[order]
            t_rowptr = torch._convert_indices_from_coo_to_csr(t_rows, ncol, out_int32=True)
            t = CsrOperator(
                t_rowptr, rows[order].to(torch.int32).contiguous(), self.values[order].contiguous(),
                (ncol, nrow),
            )
            # an exactly symmetric operator (content-equal to a live CSR, usually `self`) shares that object and
            # with it the tile plans; anything else keeps its own transpose
            shared = _dedup_by_content(t)
            if shared is t:
                t._t = self
            self._t = shared
        return self._t


HOP_MODE = "auto"            # "auto" | "fused" | "staged": which staged recurrence kernel dense stencils take (A/B runs, tests)
STAGED_TILE_ROWS = (64, 128) # tile heights the staged one-hop plan tries, in this order (64: one row per lane group)
STAGED_MIN_ROW_LEN = 14.0    # average entries per row from which "auto" picks the staged one-hop kernel
MIN_CLUSTERED_TILES = 8   # performance choice only (tests lower it to run the fused path on tiny graphs as well)

_op_cache: dict = {}
_content_cache: dict = {}   # (device, shape, nnz) -> [weakref(CsrOperator)]: operators with equal content share ONE object


def _dedup_by_content(op: CsrOperator) -> CsrOperator:
    """``model.to(device)`` gives every ConvCheb of a U-Net level its own copy of the same Laplacian (11 buffers,
    3 distinct operators).  The derived CSR, its transpose and the host-built two-hop tile plans are keyed on the
    CONTENT: a freshly derived CSR that equals a live one (exact comparison on the device, a few MB) is dropped in
    favour of it, so transposes and plans are built once per distinct operator, not once per layer."""
    key = (str(op.device), op.shape, op.nnz)
    alive = []
    hit = None
    for ref in _content_cache.get(key, []):
        cand = ref()
        if cand is None:
            continue
        alive.append(ref)
        if hit is None and torch.equal(cand.colind, op.colind) and torch.equal(cand.rowptr, op.rowptr) \
                and torch.equal(cand.values, op.values):
            hit = cand
    if hit is None:
        alive.append(weakref.ref(op))
        hit = op
    _content_cache[key] = alive
    return hit


def invalidate_operator_caches():
    """Forget every derived CSR / transpose / tile plan (after an in-place change of operator buffers that the
    identity / version key cannot see, e.g. a broadcast into a sparse tensor's values)."""
    _op_cache.clear()
    _content_cache.clear()


def get_operator(mat: torch.Tensor) -> CsrOperator:
    """CSR cache keyed on the sparse buffer's identity / version / device / dtype, de-duplicated by content.

    ``model.to(device)`` and dtype casts create new tensors (new id); ``load_state_dict`` copies
    in place and bumps ``_version`` - both invalidate the entry.
    """
    key = id(mat)
    sig = (mat._version, mat.device, mat.dtype, tuple(mat.shape))
    hit = _op_cache.get(key)
    if hit is not None and hit[0]() is mat and hit[1] == sig:
        return hit[2]
    op = _dedup_by_content(CsrOperator.from_sparse_coo(mat))

    def _evict(_ref, key=key):
        _op_cache.pop(key, None)

    _op_cache[key] = (weakref.ref(mat, _evict), sig, op)
    return op


# ----------------------------------------------------------------------------------------------
# Backends: the HIP library (product) and an injectable checker (tests only)
# ----------------------------------------------------------------------------------------------
def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _plan_ptr(op, x, channels=None):
    """Address of the dsw_hop2_plan struct for this operator / row size (``channels`` of x's dtype, default the
    last axis of x), or None."""
    if _plan_ptr.disabled or op is None:
        return None, None
    row_bytes = (x.shape[-1] if channels is None else channels) * x.element_size()
    if row_bytes > 128 and row_bytes % 128 == 0:
        row_bytes = 128          # wide rows are staged one 128-byte channel chunk at a time (dsw_spmm2.hip)
    plan = op.hop2_plan(row_bytes)
    if plan is None:
        return None, None
    import ctypes

    return ctypes.addressof(plan._struct), plan   # the plan object must outlive the call


_plan_ptr.disabled = False
_FWD_FUSED = True   # forward hop pairs run fused whenever a plan exists


def _aligned16(t):
    """A backward without basis planes (``T`` = None) exists only as the one-launch dual form, whose 16-byte vector loads need
    16-byte aligned operands.  Tensors torch allocates are (256-byte) aligned; a contiguous VIEW into a flat buffer at an odd
    element offset (flattened parameter / gradient buckets of other frameworks) is not: it is copied once - rare, and cheaper
    than keeping the planes for everybody (the shape check of the forward cannot see the backward's pointers)."""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


class _HipBackend:
    """Thin tensor-level wrapper over the C ABI (include/dsw_hip.h)."""

    name = "hip"

    def spmm(self, op, x, alpha=1.0, z=None, beta=0.0, z2=None, gamma=0.0, out=None):
        """``x`` / ``out`` / ``z``: dense ``[B, V, C]`` or row-strided channel slices of wider tensors (see ``row_stride``)."""
        lib = _native.load()
        B, v_in, C = x.shape
        assert v_in == op.shape[1]
        y = out if out is not None else torch.empty((B, op.shape[0], C), dtype=x.dtype, device=x.device)
        ldx, ldy = row_stride(x), row_stride(y)
        ldz = C if z is None else row_stride(z)
        assert ldx is not None and ldy is not None and ldz is not None and y.shape == (B, op.shape[0], C)
        assert z is None or z.shape == y.shape
        with torch.cuda.device(x.device):
            if ldx == C and ldy == C and ldz == C:
                rc = lib.dsw_spmm_csr(
                    op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), op.shape[0], op.shape[1],
                    op.nnz, x.data_ptr(), y.data_ptr(), B, C, alpha, _ptr(z), beta, _ptr(z2), gamma,
                    _DTYPES[x.dtype], _stream(x),
                )
            else:
                rc = lib.dsw_spmm_csr_ld(
                    op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), op.shape[0], op.shape[1],
                    op.nnz, x.data_ptr(), ldx, y.data_ptr(), ldy, B, C, alpha, _ptr(z), ldz, beta, _ptr(z2), gamma,
                    _DTYPES[x.dtype], _stream(x),
                )
        _native.check(rc, "dsw_spmm_csr")
        return y

    def remap(self, op, x, z=None, beta=0.0, out=None):
        """Interpolation-pooling product ``op @ x (+ beta z)`` through ``dsw_remap_csr`` with the operator's remap plan."""
        import ctypes

        lib = _native.load()
        B, v_in, C = x.shape
        assert v_in == op.shape[1]
        y = out if out is not None else torch.empty((B, op.shape[0], C), dtype=x.dtype, device=x.device)
        ldx, ldy = row_stride(x), row_stride(y)
        ldz = C if z is None else row_stride(z)
        assert ldx is not None and ldy is not None and ldz is not None and y.shape == (B, op.shape[0], C)
        assert z is None or z.shape == y.shape
        plan = op.remap_plan()
        with torch.cuda.device(x.device):
            rc = lib.dsw_remap_csr(ctypes.addressof(plan), op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(),
                                   op.shape[0], op.shape[1], op.nnz, x.data_ptr(), ldx, y.data_ptr(), ldy, B, C, _ptr(z), ldz,
                                   beta, _DTYPES[x.dtype], _stream(x))
        _native.check(rc, "dsw_remap_csr")
        return y

    def cheb_basis(self, op, x, K):
        lib = _native.load()
        B, V, C = x.shape
        T = torch.empty((max(K - 1, 0), B, V, C), dtype=x.dtype, device=x.device)
        if K > 1:
            pp, _keep = _plan_ptr(op, x) if (K > 1 and _FWD_FUSED) else (None, None)
            with torch.cuda.device(x.device):
                rc = lib.dsw_cheb_basis_fwd(
                    op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz,
                    x.data_ptr(), T.data_ptr(), B, C, K, _DTYPES[x.dtype], _stream(x), pp,
                )
            _native.check(rc, "dsw_cheb_basis_fwd")
        return T

    def cheb_fwd(self, op, x, w, bias, relu=False, keep_basis=True, no_backward=False):
        """``keep_basis=False``: the caller's backward is plain ``cheb_bwd`` / ``cheb_bwd_res`` without scale / dx_add - where
        that backward runs in the dual form (``dsw_cheb_bwd_needs_basis`` == 0: X and dY only) AND the forward is the
        one-launch kernel, the basis planes are neither allocated nor stored and ``T`` comes back as None.
        ``no_backward=True`` (inference, ``torch.no_grad()``): no backward will ever run, so the planes are dropped wherever
        the forward can do without them, without building the plan of the transposed operator."""
        lib = _native.load()
        B, V, Fin = x.shape
        _, K, Fout = w.shape
        y = torch.empty((B, V, Fout), dtype=x.dtype, device=x.device)
        # mix-first layers (channel-shrinking, see dsw_cheb_mix_first) run their hops on Fout channels, use T as
        # scratch only and need nothing but x for backward
        mix_first = bool(lib.dsw_cheb_mix_first(Fin, Fout, K))
        pp, _keep = (_plan_ptr(op, x, Fout if mix_first else Fin)
                     if (K > 1 and (_FWD_FUSED or mix_first)) else (None, None))
        drop = False
        if not keep_basis and K > 1 and not mix_first and pp is not None and op is not None \
                and lib.dsw_cheb_fwd_path(pp, Fin, Fout, K, _DTYPES[x.dtype]) == 3:      # DSW_FWD_ONE_LAUNCH takes T = NULL
            if no_backward:
                drop = True        # inference / no_grad: nobody will read the planes - and the plan of L^T is not even built
            else:
                ppt, _keep_t = _plan_ptr(op.transpose(), x, Fin)
                drop = ppt is not None and lib.dsw_cheb_bwd_needs_basis(ppt, V, Fin, Fout, K, _DTYPES[x.dtype]) == 0
        T = torch.empty((K - 1, B, V, Fin), dtype=x.dtype, device=x.device) if (K > 1 and not drop) else None
        csr = (None, None, None, V, 0) if op is None else (
            op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz)
        ws, nws = self._fwd_workspace(lib, x, Fin, Fout, K)
        with torch.cuda.device(x.device):
            rc = lib.dsw_cheb_fwd_ws(
                *csr,
                x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), Fout, _ptr(T), B, Fin, Fout, K,
                _DTYPES[x.dtype], _stream(x), pp, 1 if relu else 0, None, None, 0, _ptr(ws), nws,
            )
        _native.check(rc, "dsw_cheb_fwd_ws")
        return y, (None if mix_first else T)

    @staticmethod
    def _fwd_workspace(lib, x, Fin, Fout, K):
        """Scratch for the per-call weight image of the forward (dsw_cheb_fwd_workspace_bytes): only wide fp32 layers (the
        streaming-W GEMM) and mix-first layers use it; the 32-channel layers of the one-launch forward get none."""
        if x.dtype != torch.float32 or K * Fin * Fout < 3 * 64 * 128:
            return None, 0
        n = int(lib.dsw_cheb_fwd_workspace_bytes(x.shape[0], x.shape[1], Fin, Fout, K, _DTYPES[x.dtype]))
        if n < 0:
            _native.check(n, "dsw_cheb_fwd_workspace_bytes")
        return torch.empty((n,), dtype=torch.uint8, device=x.device), n

    @staticmethod
    def _bwd_workspace(lib, x, Fin, Fout, K, dt):
        """Scratch of one backward call.  The size depends on the CU count of the CURRENT device (slabs of the one-launch dual
        backward): queried under the guard of the tensor's device, where the launch will run."""
        with torch.cuda.device(x.device):
            nbytes = int(lib.dsw_cheb_bwd_workspace_bytes(x.shape[0], x.shape[1], Fin, Fout, K, dt))
        if nbytes < 0:
            _native.check(nbytes, "dsw_cheb_bwd_workspace_bytes")
        return torch.empty((nbytes,), dtype=torch.uint8, device=x.device), nbytes

    def cheb_bwd(self, op, x, T, w, dy, need_dx, need_dw, need_db):
        lib = _native.load()
        B, V, Fin = x.shape
        _, K, Fout = w.shape
        dt = _DTYPES[x.dtype]
        dx = torch.empty_like(x) if need_dx else None
        want_w = need_dw or need_db
        dw = torch.empty_like(w) if want_w else None
        db = torch.empty((Fout,), dtype=w.dtype, device=w.device) if want_w else None
        ws, nbytes = self._bwd_workspace(lib, x, Fin, Fout, K, dt)
        mix_first = bool(lib.dsw_cheb_mix_first(Fin, Fout, K))
        if T is None and K > 1 and not mix_first:
            x, w, dy = _aligned16(x), _aligned16(w), _aligned16(dy)
        need_hops = K > 1 and (need_dx or (mix_first and want_w) or T is None)    # (T is None: the dual form runs its hops on dY)
        opt = op.transpose() if need_hops else op
        pp, _keep = _plan_ptr(opt, x, Fout if mix_first else Fin) if (need_hops and K > 1) else (None, None)
        csr = (None, None, None, V, 0) if opt is None else (
            opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz)
        with torch.cuda.device(x.device):
            rc = lib.dsw_cheb_bwd(
                *csr,
                x.data_ptr(), _ptr(T), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(dw), _ptr(db),
                ws.data_ptr(), nbytes, B, Fin, Fout, K, dt, _stream(x), pp,
            )
        _native.check(rc, "dsw_cheb_bwd")
        return dx, (dw if need_dw else None), (db if need_db else None)


    def cheb_fwd_res(self, op, x, w, bias, scale, res, out=None):
        """``y = scale * conv(x) + res`` in the epilogue of the channel-mix GEMM (dsw_cheb_fwd_res); ``out``: optional
        row-strided destination (basis-first layers).  Returns (y, T or None)."""
        lib = _native.load()
        B, V, Fin = x.shape
        _, K, Fout = w.shape
        y = out if out is not None else torch.empty((B, V, Fout), dtype=x.dtype, device=x.device)
        T = torch.empty((K - 1, B, V, Fin), dtype=x.dtype, device=x.device) if K > 1 else None
        mix_first = bool(lib.dsw_cheb_mix_first(Fin, Fout, K))
        pp, _keep = (_plan_ptr(op, x, Fout if mix_first else Fin)
                     if (K > 1 and (_FWD_FUSED or mix_first)) else (None, None))
        csr = (None, None, None, V, 0) if op is None else (
            op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), V, op.nnz)
        ws, nws = self._fwd_workspace(lib, x, Fin, Fout, K)
        with torch.cuda.device(x.device):
            rc = lib.dsw_cheb_fwd_ws(
                *csr, x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), row_stride(y), _ptr(T), B, Fin, Fout, K,
                _DTYPES[x.dtype], _stream(x), pp, 0, _ptr(scale), _ptr(res), 0 if res is None else row_stride(res),
                _ptr(ws), nws,
            )
        _native.check(rc, "dsw_cheb_fwd_ws")
        return y, (None if mix_first else T)

    def cheb_bwd_res(self, op, x, T, w, dy, need_dx, need_dw, scale=None, dx_add=None, acc_w=None, acc_b=None):
        """Backward of ``scale * conv(x)`` taking ``dy`` as it arrives: ``dx = scale * (...) + dx_add``; ``dw_raw`` /
        ``db_raw`` are NOT scaled (see ``rezero_param_grads``).  ``acc_w`` (and ``acc_b``): gradient buffers of the
        parameters to ADD the weight gradients to (``dsw_cheb_bwd_res`` with ``accumulate_dw``); they are returned."""
        lib = _native.load()
        B, V, Fin = x.shape
        _, K, Fout = w.shape
        dt = _DTYPES[x.dtype]
        dx = torch.empty_like(x) if need_dx else None
        accumulate = acc_w is not None
        if accumulate:
            dw, db = acc_w, acc_b
        else:
            dw = torch.empty_like(w) if need_dw else None
            db = torch.empty((Fout,), dtype=w.dtype, device=w.device) if need_dw else None
        ws, nbytes = self._bwd_workspace(lib, x, Fin, Fout, K, dt)
        mix_first = bool(lib.dsw_cheb_mix_first(Fin, Fout, K))
        if T is None and K > 1 and not mix_first:
            x, w, dy = _aligned16(x), _aligned16(w), _aligned16(dy)
        need_hops = K > 1 and (need_dx or (mix_first and need_dw) or T is None)
        opt = op.transpose() if need_hops else op
        pp, _keep = _plan_ptr(opt, x, Fout if mix_first else Fin) if (need_hops and K > 1) else (None, None)
        csr = (None, None, None, V, 0) if opt is None else (
            opt.rowptr.data_ptr(), opt.colind.data_ptr(), opt.values.data_ptr(), V, opt.nnz)
        with torch.cuda.device(x.device):
            rc = lib.dsw_cheb_bwd_res(
                *csr, x.data_ptr(), _ptr(T), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(dw), _ptr(db),
                ws.data_ptr(), nbytes, B, Fin, Fout, K, dt, _stream(x), pp, _ptr(scale),
                _ptr(dx_add) if need_dx else None, 0 if dx_add is None else row_stride(dx_add), 1 if accumulate else 0,
            )
        _native.check(rc, "dsw_cheb_bwd_res")
        return dx, dw, db

    # (device, stream handle) -> the (zero-initialised, self-re-arming) ticket + partials workspace of dsw_rezero_param_grads.
    # One per STREAM: calls sharing a workspace must be stream-ordered (include/dsw_hip.h), and backward branches may run on
    # side streams (GradBucket's overlap mode, forked residual branches, the capture stream of a HIP graph).  A recycled
    # handle inherits the previous stream's workspace, which is sound: the kernel re-arms the ticket at the end of every call,
    # so a workspace is only ever left half-armed by an aborted kernel (a device fault ends the process).  The cache is
    # BOUNDED (least recently used entry dropped beyond _RPG_WS_MAX; torch's stream pool has 32 streams per device), so
    # short-lived streams cannot leak workspaces without limit.
    _rpg_ws = {}
    _RPG_WS_MAX = 64

    def rezero_param_grads(self, w, bias, dw_raw, db_raw, scale):
        lib = _native.load()
        ds = torch.empty_like(scale)
        nb = 0 if bias is None else bias.numel()
        key = (w.device, _stream(w))
        ws = self._rpg_ws.pop(key, None)
        if ws is None:
            ws = torch.zeros(int(lib.dsw_rezero_param_grads_workspace_bytes()), dtype=torch.uint8, device=w.device)
            while len(self._rpg_ws) >= self._RPG_WS_MAX:
                self._rpg_ws.pop(next(iter(self._rpg_ws)))
        self._rpg_ws[key] = ws          # (re-)inserted last: dict order = recency
        with torch.cuda.device(w.device):
            rc = lib.dsw_rezero_param_grads(w.data_ptr(), _ptr(bias), dw_raw.data_ptr(), _ptr(db_raw) if nb else None,
                                            scale.data_ptr(), dw_raw.data_ptr(), _ptr(db_raw) if nb else None, ds.data_ptr(),
                                            w.numel(), nb, ws.data_ptr(), ws.numel(), _DTYPES[w.dtype], _stream(w))
        _native.check(rc, "dsw_rezero_param_grads")
        return dw_raw, (db_raw if nb else None), ds

    def relu_bwd(self, dy, y):
        lib = _native.load()
        out = torch.empty_like(dy)
        with torch.cuda.device(dy.device):
            rc = lib.dsw_relu_bwd(dy.data_ptr(), y.data_ptr(), out.data_ptr(), dy.numel(), _DTYPES[dy.dtype], _stream(dy))
        _native.check(rc, "dsw_relu_bwd")
        return out

    def rezero_fwd(self, c, r, w, out=None):
        lib = _native.load()
        if out is not None:    # a row-strided destination: the block's half of a concatenation buffer
            C = c.shape[-1]
            with torch.cuda.device(c.device):
                rc = lib.dsw_rezero_residual_fwd_ld(c.data_ptr(), r.data_ptr(), w.data_ptr(), out.data_ptr(),
                                                    c.numel() // C, C, row_stride(out), _DTYPES[c.dtype], _stream(c))
            _native.check(rc, "dsw_rezero_residual_fwd_ld")
            return out
        y = torch.empty_like(c)
        with torch.cuda.device(c.device):
            rc = lib.dsw_rezero_residual_fwd(c.data_ptr(), r.data_ptr(), w.data_ptr(), y.data_ptr(), c.numel(),
                                             _DTYPES[c.dtype], _stream(c))
        _native.check(rc, "dsw_rezero_residual_fwd")
        return y

    def rezero_bwd(self, g, c, w, need_c):
        lib = _native.load()
        gc = torch.empty_like(g) if need_c else None
        gw = torch.empty_like(w)
        nb = int(lib.dsw_rezero_residual_workspace_bytes())
        ws = torch.empty((nb,), dtype=torch.uint8, device=g.device)
        with torch.cuda.device(g.device):
            rc = lib.dsw_rezero_residual_bwd(g.data_ptr(), c.data_ptr(), w.data_ptr(), _ptr(gc), gw.data_ptr(),
                                             ws.data_ptr(), nb, g.numel(), _DTYPES[g.dtype], _stream(g))
        _native.check(rc, "dsw_rezero_residual_bwd")
        return gc, gw


    def maxval_pool_fwd(self, op, x):
        lib = _native.load()
        B, v_in, C = x.shape
        y = torch.empty((B, op.shape[0], C), dtype=x.dtype, device=x.device)
        sel = torch.empty((B, op.shape[0], C), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dsw_maxval_pool_fwd(op.rowptr.data_ptr(), op.colind.data_ptr(), op.values.data_ptr(), op.shape[0],
                                         op.shape[1], x.data_ptr(), y.data_ptr(), sel.data_ptr(), B, C, _DTYPES[x.dtype],
                                         _stream(x))
        _native.check(rc, "dsw_maxval_pool_fwd")
        return y, sel

    def maxval_pool_bwd(self, op, dy, sel):
        lib = _native.load()
        B, v_out, C = dy.shape
        opt = op.transpose()
        dx = torch.empty((B, op.shape[1], C), dtype=dy.dtype, device=dy.device)
        with torch.cuda.device(dy.device):
            rc = lib.dsw_maxval_pool_bwd(opt.rowptr.data_ptr(), opt.colind.data_ptr(), op.shape[1], op.shape[0],
                                         dy.data_ptr(), sel.data_ptr(), dx.data_ptr(), B, C, _DTYPES[dy.dtype], _stream(dy))
        _native.check(rc, "dsw_maxval_pool_bwd")
        return dx

    def maxval_unpool_fwd(self, x, sel, v_fine):
        lib = _native.load()
        B, v_coarse, C = x.shape
        y = torch.empty((B, v_fine, C), dtype=x.dtype, device=x.device)
        nb = int(lib.dsw_maxval_unpool_workspace_bytes(B, v_fine, C))
        ws = torch.empty((max(nb, 4),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dsw_maxval_unpool_fwd(sel.data_ptr(), x.data_ptr(), y.data_ptr(), ws.data_ptr(), nb, B, v_coarse,
                                           v_fine, C, _DTYPES[x.dtype], _stream(x))
        _native.check(rc, "dsw_maxval_unpool_fwd")
        return y

    def maxval_unpool_bwd(self, dy, sel):
        lib = _native.load()
        B, v_fine, C = dy.shape
        v_coarse = sel.shape[1]
        dx = torch.empty((B, v_coarse, C), dtype=dy.dtype, device=dy.device)
        with torch.cuda.device(dy.device):
            rc = lib.dsw_maxval_unpool_bwd(sel.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, v_coarse, v_fine, C,
                                           _DTYPES[dy.dtype], _stream(dy))
        _native.check(rc, "dsw_maxval_unpool_bwd")
        return dx


_HIP = _HipBackend()
_test_backend = None


def set_test_backend(backend):
    """TEST-ONLY hook: backend used for non-CUDA tensors (``None`` restores the strict default).

    The product never calls this.  It exists so that the host logic (autograd wiring, module
    plumbing, DDP sharding) can be exercised on CPU with the oracle standing in as the checker.
    """
    global _test_backend
    _test_backend = backend


def _backend_for(t: torch.Tensor):
    if t.is_cuda:
        return _HIP
    if _test_backend is not None:
        return _test_backend
    raise RuntimeError(
        "dsw: the ConvCheb / RemapBlock hot path runs only on a ROCm device (got a "
        f"{t.device.type} tensor). There is no CPU fallback; move the model and inputs to 'cuda'."
    )


def _check_dtype(*tensors):
    dt = tensors[0].dtype
    if dt not in _DTYPES:
        raise TypeError(f"dsw: unsupported dtype {dt}; the HIP path implements float32 and bfloat16")
    for t in tensors[1:]:
        if t is not None and t.dtype != dt:
            raise TypeError(f"dsw: mixed dtypes {dt} vs {t.dtype}")


# ----------------------------------------------------------------------------------------------
# autograd Functions
# ----------------------------------------------------------------------------------------------
PAD_INPUT_CHANNELS = True   # see _padded_width


def _padded_width(fin: int, dtype) -> int:
    """Input width the kernels should see.  The fast paths (whole-forward kernel, two-hop pairs, bf16-pipe GEMMs, fused
    backward pass) want rows of whole 32-channel chunks; a layer with, say, 18 input channels (the U-Net's first:
    3 time steps x 6 fields) otherwise takes the exact-fp32 kernels with predicated scalar loads - 4 launches, 150 us =
    4 % of the U-Net step (round 2).  Zero channels contribute exact zeros, so such a layer is evaluated as the next
    multiple of 32 (x and the weight rows padded with zeros, the gradients sliced back) whenever that is at most twice
    the real width: 16..31 -> 32, 33..63 -> 64, ...  fp32 only (the bf16 paths have other alignment rules)."""
    if not PAD_INPUT_CHANNELS or dtype != torch.float32 or fin % 32 == 0:
        return fin
    padded = (fin + 31) // 32 * 32
    return padded if padded <= 2 * fin else fin


def grad_accumulators(weight, bias):
    """The gradient buffers a ``GradBucket`` in direct mode attached to these parameters (``None`` otherwise / when the
    bias has none while the weight does: the kernels produce both or neither)."""
    aw = getattr(weight, "_dsw_grad_acc", None)
    if aw is None:
        return None
    ab = None if bias is None else getattr(bias, "_dsw_grad_acc", None)
    if bias is not None and ab is None:
        return None
    # The pointer is only as good as the bucket that set it: it must still BE the parameter's gradient (a second bucket,
    # `zero_grad(set_to_none=True)`, `.to()` / `.half()` after attach all replace `p.grad`) and have the parameter's layout
    # (the kernels write the parameter's element order).  Anything else falls back to returning dW / db to autograd.
    if weight.grad is not aw or aw.shape != weight.shape or aw.stride() != weight.stride() or aw.dtype != weight.dtype \
            or aw.device != weight.device:
        return None
    if bias is not None and (bias.grad is not ab or ab.shape != bias.shape or ab.dtype != bias.dtype):
        return None
    return aw, ab


class _ChebConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, op, relu=False, acc=None):
        be = _backend_for(x)
        ctx.acc = acc if (acc is not None and x.is_cuda) else None
        fin = x.shape[-1]
        fpad = _padded_width(fin, x.dtype) if x.is_cuda else fin
        ctx.fin = fin
        if fpad != fin:
            xc = torch.nn.functional.pad(x, (0, fpad - fin))                       # [B, V, fpad], zeros behind the real channels
            wc = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, fpad - fin))      # [fpad, K, Fout]
        else:
            xc = x.contiguous()
            wc = weight.contiguous()
        bc = None if bias is None else bias.contiguous()
        if getattr(be, "name", "") == "hip":
            # this Function's backward is the plain closed form: the basis planes are dropped where it runs in the dual form
            no_bwd = not any(ctx.needs_input_grad[:3])    # (apply() under no_grad / on constants: all False)
            y, T = be.cheb_fwd(op, xc, wc, bc, relu, keep_basis=False, no_backward=no_bwd)
        else:
            y, T = be.cheb_fwd(op, xc, wc, bc, relu) if relu else be.cheb_fwd(op, xc, wc, bc)
        # the plain output is NOT saved: callers modify it in place (layers.py:375, my_models_graph.py:213).  With the
        # fused activation the output IS the ReLU result, whose sign pattern backward needs (as F.relu saves its result);
        # an in-place edit by the caller then trips autograd's version check instead of corrupting gradients.
        ctx.save_for_backward(xc, wc, T, y if relu else None)
        ctx.op = op
        ctx.has_bias = bias is not None
        ctx.be = be
        ctx.relu = relu
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        xc, wc, T, y = ctx.saved_tensors
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], (
            ctx.has_bias and ctx.needs_input_grad[2]
        )
        dy = dy.contiguous()
        if ctx.relu:
            dy = ctx.be.relu_bwd(dy, y)
        if ctx.acc is not None and need_dw and ctx.fin == xc.shape[-1]:
            # the parameters' gradient buffers take the contribution directly (dW +=, db += in the reduce stage of the
            # weight-gradient kernels): nothing is returned for them, autograd launches no `add`
            dx, _w, _b = ctx.be.cheb_bwd_res(ctx.op, xc, T, wc, dy, need_dx, True, acc_w=ctx.acc[0], acc_b=ctx.acc[1])
            return dx, None, None, None, None, None
        dx, dw, db = ctx.be.cheb_bwd(ctx.op, xc, T, wc, dy, need_dx, need_dw, need_db)
        if ctx.fin != xc.shape[-1]:       # zero-padded input channels: the real ones are the leading slice
            dx = None if dx is None else dx[..., :ctx.fin].contiguous()
            dw = None if dw is None else dw[:ctx.fin].contiguous()
        return dx, dw, db, None, None, None


class _ChebConvResFn(torch.autograd.Function):
    """``scale * conv(x) + res`` - the last convolution of a residual block with the block's ReZero scale and residual add
    in its epilogue (reference my_models_graph.py:205-216).  Backward never forms ``scale * dY``: the dgrad kernels scale
    their output, the weight gradients are rescaled by one tiny launch that also yields d scale = <W, dW_raw> + <b, db_raw>
    (= sum dY * conv(x), without conv(x) ever having been stored)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, res, op, out, epilogue=True):
        be = _backend_for(x)
        xc, wc = x.contiguous(), weight.contiguous()
        bc = None if bias is None else bias.contiguous()
        if epilogue:
            rc = res if _rows_ok(res) else res.contiguous()
            y, T = be.cheb_fwd_res(op, xc, wc, bc, scale, rc, out=out.t if out is not None else None)
        else:
            # forward as two launches (convolution, then scale + residual in one pass over its output - the unscaled
            # convolution is NOT kept); the backward below is the same
            c, T = be.cheb_fwd(op, xc, wc, bc)
            y = be.rezero_fwd(c, res.contiguous(), scale, out=out.t if out is not None else None)
        ctx.save_for_backward(xc, wc, bc, T, scale)
        ctx.op, ctx.be = op, be
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xc, wc, bc, T, scale = ctx.saved_tensors
        need_dx = ctx.needs_input_grad[0]
        need_par = ctx.needs_input_grad[1] or (bc is not None and ctx.needs_input_grad[2]) or ctx.needs_input_grad[3]
        g = g.contiguous()
        dx, dw, db = ctx.be.cheb_bwd_res(ctx.op, xc, T, wc, g, need_dx, need_par, scale=scale)
        ds = None
        if need_par:
            dw, db, ds = ctx.be.rezero_param_grads(wc, bc, dw, db, scale)
        return (dx, dw if ctx.needs_input_grad[1] else None, db if (bc is not None and ctx.needs_input_grad[2]) else None,
                ds if ctx.needs_input_grad[3] else None, g if ctx.needs_input_grad[4] else None, None, None, None)


class _DenseForkFn(torch.autograd.Function):
    """``(x_again, x @ weight + bias)`` for an input with a second consumer (the residual branch of a ResBlock reads the
    block's input next to the convolution stack): the gradient the other consumer sends back through ``x_again`` is added
    in the epilogue of this map's dgrad GEMM instead of by an autograd ``add`` pass over the tensor."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        be = _backend_for(x)
        lead = x.shape[:-1]
        x2 = x.contiguous().reshape(1, -1, x.shape[-1])
        wc = weight.unsqueeze(1).contiguous()
        bc = None if bias is None else bias.contiguous()
        y, _T = be.cheb_fwd(None, x2, wc, bc)
        ctx.save_for_backward(x2, wc)
        ctx.be, ctx.has_bias, ctx.wshape = be, bias is not None, weight.shape
        return x.view_as(x), y.reshape(*lead, weight.shape[1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_x, g_r):
        x2, wc = ctx.saved_tensors
        if g_r is None:
            return g_x, None, None
        need_dx = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        g2 = g_r.contiguous().reshape(1, -1, g_r.shape[-1])
        add = None
        if g_x is not None and need_dx:
            add = g_x.reshape(1, -1, g_x.shape[-1]) if g_x.is_contiguous() else g_x.contiguous().reshape(1, -1, g_x.shape[-1])
        dx, dw, db = ctx.be.cheb_bwd_res(None, x2, None, wc, g2, need_dx, need_w, dx_add=add)
        if dx is not None:
            dx = dx.reshape(g_r.shape[:-1] + (x2.shape[-1],))
        return (dx, dw.reshape(ctx.wshape) if (dw is not None and ctx.needs_input_grad[1]) else None,
                db if (ctx.has_bias and ctx.needs_input_grad[2]) else None)


class _Out:
    """Carries a preallocated destination tensor into an autograd Function WITHOUT making it an input of the node."""

    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def row_stride(t: torch.Tensor):
    """Elements between consecutive node rows of a ``[B, V, C]`` tensor whose rows are contiguous and whose samples follow
    each other without a gap (a dense tensor, or a channel slice ``buf[..., a:b]`` of a dense wider one); else None."""
    if t.dim() != 3:
        return None
    B, V, C = t.shape
    sb, sv, sc = t.stride()
    if C == 0 or V == 0 or B == 0:
        return C
    if sc != 1 and C > 1:
        return None
    ld = sv if V > 1 else max(C, sb if B > 1 else C)
    if ld < C or (B > 1 and sb != V * ld):
        return None
    return ld


def _rows_ok(t: torch.Tensor) -> bool:
    """Row-strided and 16-byte aligned (what the strided kernel entry points take)."""
    ld = row_stride(t)
    es = t.element_size()
    return ld is not None and (ld * es) % 16 == 0 and (t.shape[-1] * es) % 16 == 0 and t.data_ptr() % 16 == 0


class _RezeroResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, r, w, out):
        be = _backend_for(c)
        cc, rc = c.contiguous(), r.contiguous()
        ctx.save_for_backward(cc, w)
        ctx.be = be
        return be.rezero_fwd(cc, rc, w, out=out.t if out is not None else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        c, w = ctx.saved_tensors
        g = g.contiguous()
        gc, gw = ctx.be.rezero_bwd(g, c, w, ctx.needs_input_grad[0])
        return gc, (g if ctx.needs_input_grad[1] else None), (gw if ctx.needs_input_grad[2] else None), None


def _remap(be, op, x, z=None, beta=0.0, out=None):
    """The remap product on the backend: the planned entry point of the HIP library, the plain product of a test backend."""
    if hasattr(be, "remap"):
        return be.remap(op, x, z=z, beta=beta, out=out)
    return be.spmm(op, x, z=z, beta=beta, out=out)


class _RemapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, out):
        be = _backend_for(x)
        ctx.op = op
        ctx.be = be
        return _remap(be, op, x if _rows_ok(x) else x.contiguous(), out=out.t if out is not None else None)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return _remap(ctx.be, ctx.op.transpose(), dy if _rows_ok(dy) else dy.contiguous()), None, None


class _RemapForkFn(torch.autograd.Function):
    """``(x, M x)`` for a tensor with a second consumer (the U-Net's skip tensors feed the pooling AND the decoder's
    concatenation): autograd delivers the second consumer's gradient here, and the transposed product adds it in its
    epilogue (``dX = g_other + M^T dY``, one pass) instead of a separate read-read-write ``add`` over the tensor."""

    @staticmethod
    def forward(ctx, x, op):
        be = _backend_for(x)
        ctx.op = op
        ctx.be = be
        return x.view_as(x), _remap(be, op, x if _rows_ok(x) else x.contiguous())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_x, dy):
        if dy is None:
            return g_x, None
        dy = dy if _rows_ok(dy) else dy.contiguous()
        if g_x is None:
            return _remap(ctx.be, ctx.op.transpose(), dy), None
        g_x = g_x if _rows_ok(g_x) else g_x.contiguous()
        return _remap(ctx.be, ctx.op.transpose(), dy, z=g_x, beta=1.0), None


class _RemapAddFn(torch.autograd.Function):
    """``M x + addend`` in one product (the addend is the epilogue operand of the remap kernels): an unpooling whose result
    is added to a skip tensor of the fine level, without the separate read-read-write ``add`` pass over the fine tensor."""

    @staticmethod
    def forward(ctx, x, addend, op):
        be = _backend_for(x)
        ctx.op, ctx.be = op, be
        return _remap(be, op, x if _rows_ok(x) else x.contiguous(), z=addend if _rows_ok(addend) else addend.contiguous(),
                      beta=1.0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gx = None
        if ctx.needs_input_grad[0]:
            gx = _remap(ctx.be, ctx.op.transpose(), g if _rows_ok(g) else g.contiguous())
        return gx, (g if ctx.needs_input_grad[1] else None), None


class _ConcatInPlaceFn(torch.autograd.Function):
    """``torch.cat((left, right), dim=2)`` when ``left`` and ``right`` already ARE the two channel slices of ``buf``:
    no data moves forward, backward hands out the two slices of the gradient."""

    @staticmethod
    def forward(ctx, left, right, buf):
        ctx.split = left.shape[-1]
        return buf.t

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return g[..., :ctx.split], g[..., ctx.split:], None


class _MaxValPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        be = _backend_for(x)
        y, sel = be.maxval_pool_fwd(op, x.contiguous())
        ctx.op, ctx.be = op, be
        ctx.save_for_backward(sel)
        ctx.mark_non_differentiable(sel)
        return y, sel

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, _dsel):
        (sel,) = ctx.saved_tensors
        return ctx.be.maxval_pool_bwd(ctx.op, dy.contiguous(), sel), None


class _MaxValUnpoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sel, v_fine):
        be = _backend_for(x)
        ctx.be = be
        ctx.save_for_backward(sel)
        return be.maxval_unpool_fwd(x.contiguous(), sel, v_fine)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (sel,) = ctx.saved_tensors
        return ctx.be.maxval_unpool_bwd(dy.contiguous(), sel), None, None


def maxval_pool(op: CsrOperator, x: torch.Tensor):
    """Max-value pooling (reference layers.py:1040-1079): per coarse cell the (unweighted) value of the fine cell whose
    WEIGHTED value is largest.  Returns ``(y [B, Vd, F], sel int32 [B, Vd, F])`` with ``sel`` the chosen fine cell."""
    if x.dim() != 3:
        raise ValueError("expected input [B, V, F]")
    assert x.shape[1] == op.shape[1], "remap_matrix.shape[1] != x.shape[1]"      # the reference's assert
    _check_dtype(x)
    return _MaxValPoolFn.apply(x, op)


def maxval_unpool(x: torch.Tensor, sel: torch.Tensor, v_fine: int) -> torch.Tensor:
    """Max-value unpooling (reference layers.py:1082-1103): every coarse value returns to the fine cell it was pooled
    from, all other fine cells are zero.  ``sel``: int32 ``[B, Vc, F]`` as returned by :func:`maxval_pool`."""
    if x.dim() != 3 or sel.shape != x.shape or sel.dtype != torch.int32:
        raise ValueError("expected x [B, Vc, F] and an int32 selection of the same shape")
    _check_dtype(x)
    return _MaxValUnpoolFn.apply(x, sel.contiguous(), int(v_fine))


def maxval_reference_index(sel: torch.Tensor) -> torch.Tensor:
    """Compact selection ``[B, Vd, F]`` -> the reference's index tensor ``[2, F*B*Vd]`` int64 (layers.py:1069-1075):
    row = chosen fine cell, column = f*B + b of the reference's internal ``[V, F*B]`` layout, listed column-major."""
    B, D, F = sel.shape
    row = sel.permute(2, 0, 1).reshape(-1).to(torch.int64)
    col = torch.arange(F * B, device=sel.device, dtype=torch.int64).repeat_interleave(D)
    return torch.stack([row, col])


def maxval_compact_index(index: torch.Tensor, B: int, D: int, F: int) -> torch.Tensor:
    """Inverse of :func:`maxval_reference_index`."""
    return index[0].reshape(F, B, D).permute(1, 2, 0).to(torch.int32).contiguous()


def cheb_conv(op: CsrOperator, x: torch.Tensor, weight: torch.Tensor, bias=None, activation=None) -> torch.Tensor:
    """``Y = act(sum_k T_k(L) x W_k (+ bias))`` for node-major ``x [B, V, Fin]``, ``weight [Fin, K, Fout]``;
    ``activation``: None or "relu" (fused into the epilogue of the channel-mix GEMM)."""
    if activation not in (None, "relu"):
        raise ValueError("fused activation: None or 'relu'")
    if x.dim() != 3 or weight.dim() != 3:
        raise ValueError("expected inputs [B, V, Fin] and weight [Fin, K, Fout]")
    if x.shape[1] != op.shape[1] or op.shape[0] != op.shape[1]:
        raise ValueError(
            f"operator shape {op.shape} does not match the {x.shape[1]} nodes of the input"
        )
    _check_dtype(x, weight, bias)
    # (direct accumulation: the kernels write dW in the contiguous [Fin, K, Fout] order)
    acc = grad_accumulators(weight, bias) if weight.is_contiguous() else None
    return _ChebConvFn.apply(x, weight, bias, op, activation == "relu", acc)


def dense_mix(x: torch.Tensor, weight: torch.Tensor, bias=None, acc=None) -> torch.Tensor:
    """``y = x @ weight (+ bias)`` over the last axis, ``weight [Fin, Fout]``: the K = 1 channel mix (no operator),
    i.e. the per-node linear map of the residual branch (my_models_graph.py:177-180) on the same MFMA GEMM
    kernels as ConvCheb - forward, dgrad and wgrad."""
    if weight.dim() != 2 or x.shape[-1] != weight.shape[0]:
        raise ValueError("expected x [..., Fin] and weight [Fin, Fout]")
    _check_dtype(x, weight, bias)
    lead = x.shape[:-1]
    y = _ChebConvFn.apply(x.reshape(1, -1, x.shape[-1]), weight.unsqueeze(1), bias, None, False, acc)
    return y.reshape(*lead, weight.shape[1])


def cheb_conv_res(op: CsrOperator, x: torch.Tensor, weight: torch.Tensor, bias, scale: torch.Tensor, res: torch.Tensor,
                  out: torch.Tensor = None, epilogue: bool = True) -> torch.Tensor:
    """``scale * (sum_k T_k(L) x W_k + bias) + res`` with ``scale`` a one-element tensor (the ReZero parameter) and
    ``res`` of the output's shape: the tail of a residual block in the epilogue of its last convolution.  ``out``: optional
    preallocated channel slice to write into (see ``skip_slot``; basis-first layers)."""
    if x.dim() != 3 or weight.dim() != 3 or scale.numel() != 1:
        raise ValueError("expected inputs [B, V, Fin], weight [Fin, K, Fout] and a one-element scale")
    if x.shape[1] != op.shape[1] or op.shape[0] != op.shape[1]:
        raise ValueError(f"operator shape {op.shape} does not match the {x.shape[1]} nodes of the input")
    if tuple(res.shape) != (x.shape[0], x.shape[1], weight.shape[2]):
        raise ValueError("`res` must have the shape of the layer's output")
    _check_dtype(x, weight, bias, scale, res)
    return _ChebConvResFn.apply(x, weight, bias, scale, res, op, _check_out(out, res.shape, x), epilogue)


def cheb_conv_res_takes_out(fin: int, fout: int, K: int) -> bool:
    """Whether ``cheb_conv_res(..., out=slice)`` can write into a row-strided destination for this layer shape (the
    mix-first evaluation order ends in an SpMM on dense planes)."""
    return not bool(_native.load().dsw_cheb_mix_first(fin, fout, K))


def dense_mix_fork(x: torch.Tensor, weight: torch.Tensor, bias=None):
    """``(x_again, dense_mix(x, weight, bias))``: give ``x_again`` to the OTHER consumer of ``x`` and its gradient is added
    inside this map's backward GEMM (residual branch of a ResBlock, my_models_graph.py:213)."""
    if weight.dim() != 2 or x.shape[-1] != weight.shape[0] or x.dim() != 3:
        raise ValueError("expected x [B, V, Fin] and weight [Fin, Fout]")
    _check_dtype(x, weight, bias)
    return _DenseForkFn.apply(x, weight, bias)


def _check_out(out, shape, like):
    if out is None:
        return None
    if tuple(out.shape) != tuple(shape) or out.dtype != like.dtype or out.device != like.device or not _rows_ok(out):
        raise ValueError("`out` must be a 16-byte aligned [B, V, C] channel slice (or dense tensor) of the result's shape, "
                         "dtype and device")
    return _Out(out)


def rezero_residual(c: torch.Tensor, r: torch.Tensor, w: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """``w * c + r`` with ``w`` a one-element tensor (the ReZero parameter): the epilogue of the residual block
    (my_models_graph.py:211-215) in one pass forward and one pass + a tiny reduction backward.  ``out``: an optional
    preallocated destination (e.g. the block's channel slice of a concatenation buffer, see ``skip_slot``)."""
    if c.shape != r.shape or w.numel() != 1:
        raise ValueError("expected c and r of one shape and a one-element w")
    _check_dtype(c, r, w)
    return _RezeroResidualFn.apply(c, r, w, _check_out(out, c.shape, c))


def sparse_remap(op: CsrOperator, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """``Y[b, d, f] = sum_v M[d, v] X[b, v, f]`` (pooling / unpooling), output contiguous [B, Vd, F] - or written into
    ``out``, a preallocated channel slice of a wider tensor.  ``x`` may itself be such a slice (no copy is made)."""
    if x.dim() != 3:
        raise ValueError("expected input [B, V, F]")
    if x.shape[1] != op.shape[1]:
        raise ValueError(f"remap matrix has {op.shape[1]} source nodes, input has {x.shape[1]}")
    _check_dtype(x)
    return _RemapFn.apply(x, op, _check_out(out, (x.shape[0], op.shape[0], x.shape[2]), x))


def sparse_remap_add(op: CsrOperator, x: torch.Tensor, addend: torch.Tensor) -> torch.Tensor:
    """``sparse_remap(op, x) + addend`` in one launch (``addend``: ``[B, Vd, F]``, e.g. the fine-level tensor an unpooled
    coarse result is added to)."""
    if x.dim() != 3 or addend.dim() != 3:
        raise ValueError("expected input [B, V, F] and addend [B, Vd, F]")
    if x.shape[1] != op.shape[1]:
        raise ValueError(f"remap matrix has {op.shape[1]} source nodes, input has {x.shape[1]}")
    if tuple(addend.shape) != (x.shape[0], op.shape[0], x.shape[2]):
        raise ValueError("`addend` must have the shape of the product")
    _check_dtype(x, addend)
    return _RemapAddFn.apply(x, addend, op)


def sparse_remap_fork(op: CsrOperator, x: torch.Tensor):
    """``(x_again, sparse_remap(op, x))``.  Give ``x_again`` to every OTHER consumer of ``x``: same values and storage,
    and the gradient those consumers send back is added inside the backward product of the remap instead of by an
    ``add`` of autograd's gradient accumulation (the skip tensors of the U-Net: my_models_graph.py:504-545)."""
    if x.dim() != 3:
        raise ValueError("expected input [B, V, F]")
    if x.shape[1] != op.shape[1]:
        raise ValueError(f"remap matrix has {op.shape[1]} source nodes, input has {x.shape[1]}")
    _check_dtype(x)
    return _RemapForkFn.apply(x, op)


def _alias(t: torch.Tensor, offset: int, size, stride) -> torch.Tensor:
    """A tensor over the same storage as ``t`` that autograd does NOT treat as a view of it (no shared version counter,
    no base): the slices of a concatenation buffer are written by different graph nodes, which the view + in-place
    checks of autograd would otherwise refuse."""
    return torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), offset, size, stride)


_concat_buffers = weakref.WeakKeyDictionary()   # storage of a buffer made by skip_slot -> [B, V, width_left, width_skip, completed]


def skip_slot(like: torch.Tensor, n_nodes: int, width_left: int, width_skip: int):
    """Channel slice ``[..., width_left:]`` of a fresh ``[B, n_nodes, width_left + width_skip]`` buffer: the place an
    encoder block writes its output to (``ResBlock(x, out=slot)``) so that the decoder's
    ``torch.cat((unpooled, skip), dim=2)`` (my_models_graph.py:528-545) needs no copy - the unpooling later fills
    ``[..., :width_left]`` of the same buffer (``concat_in_place``).  None when the slices would not be 16-byte aligned.
    The buffer's storage is REGISTERED as owned by this mechanism: ``skip_buffer`` only ever completes storages made
    here, never a caller's tensor that merely has the same strides."""
    es = like.element_size()
    if (width_left * es) % 16 or (width_skip * es) % 16:
        return None
    width = width_left + width_skip
    buf = like.new_empty((like.shape[0], n_nodes, width))
    _concat_buffers[buf.untyped_storage()] = [like.shape[0], n_nodes, width_left, width_skip, False]
    return _alias(buf, width_left, (like.shape[0], n_nodes, width_skip), (n_nodes * width, width, 1))


def skip_buffer(skip: torch.Tensor, width_left: int):
    """The ``[B, V, width_left + C]`` buffer whose right-hand channel slice ``skip`` is, or None when ``skip`` is an
    ordinary tensor - i.e. its storage was not allocated by ``skip_slot`` with exactly this geometry, whatever its
    strides look like - or when the buffer was already completed once (a second concatenation with the same skip tensor
    must not overwrite the first one's left half: it takes the copying path)."""
    if skip.dim() != 3:
        return None
    entry = _concat_buffers.get(skip.untyped_storage())
    B, V, C = skip.shape
    width = width_left + C
    if entry is None or entry[4] or entry[:4] != [B, V, width_left, C]:
        return None
    if skip.stride() != (V * width, width, 1) or skip.storage_offset() != width_left:
        return None
    return _alias(skip, 0, (B, V, width), (V * width, width, 1))


def left_slot(buf: torch.Tensor, width_left: int) -> torch.Tensor:
    """Channel slice ``[..., :width_left]`` of a concatenation buffer, as a destination for the unpooling."""
    B, V, width = buf.shape
    return _alias(buf, 0, (B, V, width_left), (V * width, width, 1))


def concat_in_place(left: torch.Tensor, skip: torch.Tensor, buf: torch.Tensor) -> torch.Tensor:
    """``torch.cat((left, skip), dim=2)`` for ``left = buf[..., :w]`` (just written by the unpooling) and
    ``skip = buf[..., w:]``: returns ``buf`` with the autograd edges of a concatenation."""
    w = left.shape[-1]
    es = buf.element_size()
    if (left.data_ptr() != buf.data_ptr() or skip.data_ptr() != buf.data_ptr() + w * es
            or left.stride() != buf.stride() or skip.stride() != buf.stride()):
        raise ValueError("left / skip are not the two channel slices of buf")
    entry = _concat_buffers.get(buf.untyped_storage())
    if entry is not None:
        entry[4] = True        # completed: a second decode of the same encodings takes the copying path
    return _ConcatInPlaceFn.apply(left, skip, _Out(buf))


def cheb_basis(op: CsrOperator, x: torch.Tensor, K: int) -> torch.Tensor:
    """``T_1 .. T_{K-1}`` as ``[K-1, B, V, C]`` (no autograd; used by benchmarks and tests)."""
    _check_dtype(x)
    return _backend_for(x).cheb_basis(op, x.contiguous(), K)
