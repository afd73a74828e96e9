"""Host side of the MI355X message-passing backend: plans, plan caches and autograd glue over the
C ABI (include/ggl_mpops.h).

This is the part of the reference that lives in ``gammagl/mpops/torch_ext/src/*.cpp`` — the seven
``torch::autograd::Function``s and their device dispatch (src/segment_sum.cpp:35-54,
src/segment_mean.cpp:36-63, src/segment_max.cpp:37-61, src/gspmm.cpp:26-260) — restated for a
backend whose kernels work on a destination-sorted plan instead of atomics:

* ``SegPlan``   : perm / rowptr / long-row lists for one id vector (built once, cached by the
                  identity + version counter of the id tensor's storage; no per-call host sync).
* ``GraphPlan`` : the pair of SegPlans (by destination, by source) for one ``edge_index`` plus the
                  int32 column arrays — CSR for the forward SpMM, CSC for its backward.
* ``Engine``    : the ops.  PyTorch is plumbing here (device memory, streams, autograd graph).

``Engine`` takes the ctypes library as a constructor argument; the product singleton
(``gammagl_amd.engine()``) is always built on the HIP library and refuses non-GPU tensors.
"""
import ctypes
import math
import os
from collections import OrderedDict

import torch
from torch.multiprocessing.reductions import StorageWeakRef

from . import _lib
from ._lib import SegPlanC

_DTYPE_CODE = {
    torch.uint8: 0, torch.int8: 1, torch.int16: 2, torch.int32: 3, torch.int64: 4,
    torch.float16: 5, torch.bfloat16: 6, torch.float32: 7, torch.float64: 8,
}
_FLOAT_DTYPES = (torch.float16, torch.bfloat16, torch.float32, torch.float64)

DEFAULT_CHUNK = int(os.environ.get("GGL_LONG_ROW", "0"))  # 0 = automatic, see Engine.auto_chunk


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class SegPlan:
    """Destination-sorted view of one id vector (struct ggl_segplan + the tensors that own it)."""

    __slots__ = ("N", "E", "rowptr", "perm", "is_sorted", "max_len", "chunk", "long_rows",
                 "chunk_ptr", "n_long", "n_chunks", "device", "row_order", "uid", "xcd_run", "order_fn", "uses", "wperm", "long_order", "hub_first")

    def c_struct(self, partial=None, perm_override=None, unsplit=False, skip_long=False):
        """`unsplit`: present the plan without its long-row table, every row walked in one piece.
        `skip_long`: withhold the long-row table but keep the threshold — rows longer than `chunk` are left out
        of the launch (ggl_segment_hub16 fills them in)."""
        perm = self.perm if perm_override is None else perm_override
        n_long = 0 if (unsplit or skip_long) else self.n_long
        lo = getattr(self, "long_order", None)
        fn = getattr(self, "order_fn", None)
        if fn is not None:
            # the row hand-out order is a scheduling aid worth ~100 us of sorting: a plan that is used ONCE (a fresh
            # edge list per mini-batch) never pays for it, a plan that comes back gets it on its second launch
            self.uses = getattr(self, "uses", 0) + 1
            # never while a hipGraph is being recorded: the argsort would be allocated in the capture pool and only
            # FILLED on replay, and an eager launch of this plan before the first replay would read garbage row ids
            if self.uses >= 2 and not (self.rowptr.is_cuda and torch.cuda.is_current_stream_capturing()):
                self.order_fn = None
                self.row_order = fn(self.counts())
        return SegPlanC(
            rowptr=self.rowptr.data_ptr(), perm=(perm.data_ptr() if perm is not None else None),
            long_rows=(self.long_rows.data_ptr() if n_long else None),
            chunk_ptr=(self.chunk_ptr.data_ptr() if n_long else None),
            n_long=n_long, n_chunks=(self.n_chunks if n_long else 0),
            chunk=((1 << 62) if unsplit else self.chunk),
            partial=(partial.data_ptr() if partial is not None else None), N=self.N, E=self.E,
            row_order=(self.row_order.data_ptr() if self.row_order is not None else None),
            # > 0: XCD runs (a node order with locality); -1: no runs, but the long rows LEAD the id range (a degree-sorted
            # order): the hub walk of a column-blocked aggregate then runs once over the full width (include/ggl_mpops.h)
            xcd_run_rows=(int(getattr(self, "xcd_run", 0) or 0) or (-1 if (n_long and getattr(self, "hub_first", False)) else 0)),
            long_order=(lo.data_ptr() if (n_long and lo is not None) else None),
            max_len=int(getattr(self, "max_len", 0) or 0))

    def counts(self):
        return self.rowptr[1:] - self.rowptr[:-1]


class GraphPlan:
    """CSR (rows = destination) and, lazily, CSC (rows = source) plans of one edge_index."""

    __slots__ = ("engine", "index", "N_dst", "N_src", "E", "fwd", "col", "_bwd", "_colT", "_posT", "_tpos", "_rowidx", "aux")

    def __init__(self, engine, index, n_dst, n_src):
        self.engine = engine
        self.index = index
        self.N_dst, self.N_src = int(n_dst), int(n_src)
        self.E = int(index.shape[1])
        # shared with the segment-op cache: degree(dst) / unsorted_segment_*(.., edge_index[1], N)
        # on the same edge list reuse this very plan (and vice versa)
        self.fwd = engine.seg_plan(index[1], self.N_dst)
        engine._check_range(index[0], self.N_src)
        self.col = engine.gather_i32(index[0], self.fwd.perm)
        self._bwd = self._colT = self._posT = self._tpos = self._rowidx = None
        self.aux = {}  # graph-constant tensors callers derive from this edge list (e.g. GCN edge norms)
        self._schedule()

    @classmethod
    def from_csr(cls, engine, row_ptr, col_ind, col_ptr, row_ind, permute, n_rows, n_cols):
        """A GraphPlan from structures the caller already holds, no sort: the CSR of the aggregating rows
        (`row_ptr` [n_rows + 1], `col_ind` [E]: the nodes each row gathers from), its transpose (`col_ptr` [n_cols + 1],
        `row_ind` [E]) and `permute` [E] = the CSR position of every CSC entry — the five tensors FusedGATConv takes as
        keyword arguments (fusedgat_conv.py:95-100) and otherwise rebuilds on the host in every forward (:102-117)."""
        gp = cls.__new__(cls)
        gp.engine, gp.index = engine, None
        gp.N_dst, gp.N_src, gp.E = int(n_rows), int(n_cols), int(col_ind.shape[0])
        for nm, t, n in (("row_ptr", row_ptr, gp.N_dst + 1), ("col_ptr", col_ptr, gp.N_src + 1), ("col_ind", col_ind, gp.E),
                         ("row_ind", row_ind, gp.E), ("permute", permute, gp.E)):
            if t.dim() != 1 or int(t.shape[0]) != n or t.dtype not in (torch.int32, torch.int64):
                raise RuntimeError(f"{nm} must be a 1-D int32 / int64 tensor of {n} elements, got {tuple(t.shape)} {t.dtype}")
        engine._dev(row_ptr, col_ind, col_ptr, row_ind, permute)
        engine._check_range(col_ind, gp.N_src)
        engine._check_range(row_ind, gp.N_dst)
        engine._check_range(permute, max(gp.E, 1))
        for nm, ptr in (("row_ptr", row_ptr), ("col_ptr", col_ptr)):   # one-off (per plan) host reads
            if int(ptr[0]) != 0 or int(ptr[-1]) != gp.E or (ptr.numel() > 1 and bool((ptr[1:] < ptr[:-1]).any())):
                raise RuntimeError(f"{nm} must rise from 0 to the number of edges ({gp.E})")
        def own_i32(t):   # the plan keeps ITS OWN int32 copy: a later in-place edit of the caller's tensor cannot reach it
            return t.to(torch.int32).contiguous() if t.dtype != torch.int32 else t.clone().contiguous()

        gp.fwd = engine.plan_from_rowptr(row_ptr.clone() if row_ptr.dtype == torch.int64 else row_ptr, gp.E)
        gp.col = own_i32(col_ind)
        gp._bwd = engine.plan_from_rowptr(col_ptr.clone() if col_ptr.dtype == torch.int64 else col_ptr, gp.E)
        gp._colT = own_i32(row_ind)
        gp._posT = own_i32(permute)
        # edge weights arrive in CSR order: the transposed walk reads them through `permute` (a COO-built plan's CSC side
        # carries the original edge id of every position in its own `perm` instead)
        gp._bwd.wperm = gp._posT
        gp._rowidx = gp._tpos = None
        gp.aux = {}
        gp._schedule()
        return gp

    def locality(self, samples=1 << 16):
        """Share of the edges (a strided sample) whose two endpoints lie within N / 64 ids of each other: ~3 % for
        randomly labelled nodes, 30-40 % for degree-sorted power-law graphs (hub-to-hub edges), 70 %+ when the order
        comes from a clustering (partition.cluster_order) or from the data itself.  One host read."""
        E, N = self.E, max(self.N_dst, self.N_src)
        if E == 0 or self.N_dst != self.N_src:
            return 0.0
        S = min(int(samples), E)
        pos = torch.arange(S, device=self.col.device, dtype=torch.int64) * (E // S)
        rows = torch.searchsorted(self.fwd.rowptr, pos, right=True) - 1
        near = (self.col[pos].long() - rows).abs() < max(N // 64, 4096)
        return float(near.float().mean())

    def _schedule(self):
        """Scheduling hint for the row kernels (results identical either way): on a graph whose node order carries
        locality every XCD gets RUNS of consecutive row slots (SegPlan.xcd_run -> ggl_segplan.xcd_run_rows), so that a
        neighbourhood's source rows are fetched into ONE private L2 instead of all eight — measured on the
        products-sized planted-community graph in cluster order: K = 256 aggregate 13.4 -> 10.9 ms, K = 64
        3.31 -> 2.62 ms; on randomly labelled or degree-sorted R-MAT the same mapping LOSES 7-8 % (nothing to keep,
        and runs of heavy rows unbalance the XCDs), hence the test (profiles/r3_xcd_run_swizzle.txt)."""
        eng = self.engine
        run = int(eng.xcd_run_rows)
        if run < 0:       # automatic: the library's rule on this plan's locality (locality() is one host read)
            need = int(eng.lib.ggl_policy_xcd_run_rows(self.E, 1.0)) > 0 or int(eng.lib.ggl_policy_xcd_run_rows(self.E, 0.0)) > 0
            run = int(eng.lib.ggl_policy_xcd_run_rows(self.E, self.locality() if need else 0.0))
        self.fwd.xcd_run = run
        if self._bwd is not None:
            self._bwd.xcd_run = run

    @property
    def rowidx(self):
        """destination node of every sorted position (int32 [E]): the row each element of the forward walk belongs
        to, for walks that run flat over positions instead of row by row (ggl_bspmm_grad_w_sorted)."""
        if self._rowidx is None:
            if self.index is not None:
                self._rowidx = self.engine.gather_i32(self.index[1], self.fwd.perm)
            else:
                self._rowidx = torch.repeat_interleave(
                    torch.arange(self.N_dst, device=self.fwd.rowptr.device, dtype=torch.int32), self.fwd.counts())
        return self._rowidx

    @property
    def bwd(self):
        if self._bwd is None:
            self._bwd = self.engine.seg_plan(self.index[0], self.N_src)
            self._colT = self.engine.gather_i32(self.index[1], self._bwd.perm)
            self._bwd.xcd_run = int(getattr(self.fwd, "xcd_run", 0) or 0)
        return self._bwd

    @property
    def colT(self):
        self.bwd  # noqa: B018
        return self._colT

    @property
    def posT(self):
        """transposed sorted position -> forward sorted position (int32 [E])."""
        if self._posT is None:
            E, dev = self.E, self.index.device
            ar = torch.arange(E, device=dev, dtype=torch.int32)
            pf = self.fwd.perm if self.fwd.perm is not None else ar
            pt = self.bwd.perm if self.bwd.perm is not None else ar
            inv = torch.empty(E, device=dev, dtype=torch.int32)
            inv[pf.long()] = ar
            self._posT = inv[pt.long()].contiguous()
        return self._posT

    @property
    def tpos(self):
        """forward sorted position -> transposed sorted position (int32 [E]): the inverse of `posT` — where
        ggl_spmm_max_mask scatters an edge's winner bits for the transposed walk of the max backward."""
        if self._tpos is None:
            posT = self.posT
            self._tpos = torch.empty_like(posT)
            eng = self.engine
            eng._check(eng.lib.ggl_invert_perm(_ptr(posT), self.E, _ptr(self._tpos), eng._stream(posT.device)))
        return self._tpos


class _PlanCache:
    """LRU keyed on the identity of the id tensor's storage + its version counter, bounded by entry count AND
    by the bytes its plans hold (a products-sized GraphPlan is ~2.5 GB of HBM: a caller that builds a fresh
    edge_index every epoch must not accumulate 16 of them).  An entry dies with its id tensor's storage.

    Identity + version cannot see a mutation made behind autograd's back (`.data`, numpy-shared memory): set
    GGL_VERIFY_PLANS=1 to keep a (first, last, sum) checksum of the ids with every plan and verify it on each
    hit (one device reduction + a host read per call: a debugging aid, off by default)."""

    def __init__(self, cap=16, max_bytes=None):
        self.cap = cap
        self.max_bytes = int(float(os.environ.get("GGL_PLAN_CACHE_GB", "48")) * 2**30) if max_bytes is None else max_bytes
        self.d = OrderedDict()
        self.bytes = 0
        self.verify = os.environ.get("GGL_VERIFY_PLANS", "0") == "1"

    @staticmethod
    def key(t, extra):
        st = t.untyped_storage()
        return (st._cdata, t.storage_offset(), tuple(t.shape), tuple(t.stride()), t._version,
                t.dtype, str(t.device)) + tuple(extra)

    @staticmethod
    def _checksum(t):
        if t.numel() == 0:
            return (0, 0, 0)
        f = t.reshape(-1)
        return (int(f[0]), int(f[-1]), int(f.sum()))

    @staticmethod
    def _nbytes(val):
        """HBM held by a cached value (SegPlan / GraphPlan / tensor), for the byte bound."""
        if isinstance(val, SegPlan):   # the common miss (a fresh id tensor per mini-batch): no generic walk
            own = sum(t.untyped_storage().nbytes() for t in (val.rowptr, val.perm, val.long_rows, val.chunk_ptr)
                      if t is not None)
            return own + 4 * val.N        # (+ the row order it gets if it is launched again)
        seen, total = set(), 0

        def visit(o, depth=0):
            nonlocal total
            if isinstance(o, torch.Tensor):
                k = o.untyped_storage()._cdata
                if k not in seen:
                    seen.add(k)
                    total += o.untyped_storage().nbytes()
            elif depth < 3 and hasattr(o, "__slots__"):
                for a in o.__slots__:
                    if a not in ("engine", "index"):      # (the caller's own edge_index is not ours to count)
                        visit(getattr(o, a, None), depth + 1)
            elif depth < 3 and isinstance(o, dict):
                for v in o.values():
                    visit(v, depth + 1)

        visit(val)
        return total

    def get(self, t, extra):
        k = self.key(t, extra)
        hit = self.d.get(k)
        if hit is not None:
            ref, val, nb, chk = hit
            if not ref.expired():
                if chk is not None and chk != self._checksum(t):
                    raise RuntimeError("gammagl_amd: an id tensor was modified in place behind its version counter "
                                       "(.data / shared memory) after its plan was cached; call "
                                       "Engine.clear_caches() after such edits")
                self.d.move_to_end(k)
                return val
            self.bytes -= nb
            del self.d[k]
        return None

    def put(self, t, extra, val):
        k = self.key(t, extra)
        old = self.d.pop(k, None)
        if old is not None:
            self.bytes -= old[2]
        nb = self._nbytes(val)
        self.d[k] = (StorageWeakRef(t.untyped_storage()), val, nb, self._checksum(t) if self.verify else None)
        self.bytes += nb
        while len(self.d) > 1 and (len(self.d) > self.cap or self.bytes > self.max_bytes):
            _, (_, _, onb, _) = self.d.popitem(last=False)
            self.bytes -= onb

    def clear(self):
        self.d.clear()
        self.bytes = 0


class Engine:
    def __init__(self, lib, require_cuda=True, cpu_only=False):
        self.lib = lib
        self.require_cuda = require_cuda
        self.is_product = False       # set by gammagl_amd.engine() / host_engine(): bound to the shipped library (dist._default_route)
        self.cpu_only = cpu_only      # the host build: its kernels dereference host pointers
        self.seg_cache = _PlanCache()
        self.graph_cache = _PlanCache()
        self.w_cache = _PlanCache(cap=8)
        self._rng = {}
        self.stats = {"plans_built": 0, "plan_hits": 0}
        self.chunk = DEFAULT_CHUNK  # long-row threshold == elements per chunk; 0 = auto_chunk(E)
        self.gat_fast = True        # fused GAT: the low-VALU kernels where the head shape allows (GPU build only)
        self.hub16 = True           # f16 / bf16 sums: LDS-pipelined hub rows (GPU build only; A/B switch)
        self.hub16_overlap = True   # ... launched on a side stream beside the walk over the other rows (A/B switch)
        self._side = {}
        self.mean_bwd_prescale = True  # spmm mean backward = rows pre-divided by their count + plain SpMM-sum (A/B switch)
        # the launch policy's constants come from the kernel library (ggl_policy_*, include/ggl_mpops.h): ONE copy for this
        # host and the C++ one (csrc/torch/ggl_torch.cpp); the attributes below are A/B overrides for tests and probes
        win, heavy = ctypes.c_int64(0), ctypes.c_int64(0)
        lib.ggl_policy_row_order(ctypes.byref(win), ctypes.byref(heavy))
        self.row_order_window, self.row_order_heavy = int(win.value), int(heavy.value)   # see _row_order (0 = global sort)
        self.xcd_run_rows = -1   # -1 = per graph (ggl_policy_xcd_run_rows on the plan's locality), >= 0 forces it
        self.gradw_sorted = True    # bspmm weight gradient along the sorted plan with LDS-staged strips (A/B switch)
        self._make_functions()

    def clear_caches(self):
        """Drop every cached plan, sorted-weight copy and graph-constant (e.g. GCN norms).  Plans are keyed on
        the identity + version counter of the id tensor: call this after editing an edge list through `.data`
        or memory shared with numpy, or to hand the HBM back."""
        self.seg_cache.clear()
        self.graph_cache.clear()
        self.w_cache.clear()

    # ---- plumbing ----------------------------------------------------------------------------
    def _check(self, rc):
        if rc == _lib.GGL_OK:
            return
        msg = (self.lib.ggl_last_error() or b"").decode("utf-8", "replace")
        if rc == _lib.GGL_EINDEX:
            raise IndexError(msg)
        raise RuntimeError(f"ggl_mpops error {rc}: {msg}")

    def _dev(self, *tensors):
        dev = None
        for t in tensors:
            if t is None or isinstance(t, GraphPlan):
                continue
            if self.require_cuda and not t.is_cuda:
                raise RuntimeError(
                    "this is the MI355X engine (libggl_mpops_hip.so): got a tensor on %s; CPU tensors go through "
                    "gammagl_amd.engine(tensor) / gammagl_amd.mpops, which route them to the host build" % t.device)
            if self.cpu_only and t.is_cuda:
                raise RuntimeError("this is the host build of the kernels (CPU tensors only): got a tensor on %s"
                                   % t.device)
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RuntimeError("Tensor device inconsistent error.")  # segment_sum.cpp:31
        return dev

    def _side_stream(self, dev):
        s = self._side.get(str(dev))
        if s is None:
            s = torch.cuda.Stream(device=dev)
            self._side[str(dev)] = s
        return s

    @staticmethod
    def _stream(dev):
        if dev is not None and dev.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        return None

    @staticmethod
    def _code(t):
        try:
            return _DTYPE_CODE[t.dtype]
        except KeyError:
            raise RuntimeError(f"unsupported dtype {t.dtype}") from None

    # ---- plans -------------------------------------------------------------------------------
    def _check_range(self, ids, n):
        if ids.numel() == 0:
            return
        # one-off (per plan) validation of the gathered side of an edge list (one host read)
        lo, hi = torch.stack(torch.aminmax(ids)).tolist()
        if lo < 0 or hi >= n:
            raise IndexError(f"node id out of range [0, {n})")

    def auto_chunk(self, E):
        """Long-row threshold for a plan of E elements: `ggl_policy_chunk` (include/ggl_mpops.h) — the ONE copy of the
        rule both hosts use (the largest power of two <= E / resident wavefronts, clamped to [256, 4096]:
        profiles/r1_arxiv_chunk_sweep.txt)."""
        return int(self.lib.ggl_policy_chunk(int(E)))

    def build_plan(self, ids, N, chunk=None):
        """Sort `ids` (int64 [E]) into a SegPlan.  Synchronous; run once per edge list."""
        dev = self._dev(ids)
        ids = ids.contiguous()
        if ids.dtype != torch.int64:
            ids = ids.to(torch.int64)
        E, N = int(ids.shape[0]), int(N)
        chunk = int(chunk or self.chunk or self.auto_chunk(E))
        st = self._stream(dev)
        p = SegPlan()
        p.N, p.E, p.chunk, p.device = N, E, chunk, dev
        p.rowptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
        perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        wsb = self.lib.ggl_plan_workspace_bytes(E, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        is_sorted, max_len = ctypes.c_int32(0), ctypes.c_int64(0)
        self._check(self.lib.ggl_plan_build(_ptr(ids), E, N, _ptr(perm), _ptr(p.rowptr), _ptr(ws),
                                            wsb, st, ctypes.byref(is_sorted), ctypes.byref(max_len)))
        p.is_sorted, p.max_len = bool(is_sorted.value), int(max_len.value)
        p.perm = None if p.is_sorted else perm[:E]
        p.n_long = p.n_chunks = 0
        p.long_rows = p.chunk_ptr = p.long_order = None
        if p.max_len > chunk:
            lwb = self.lib.ggl_plan_long_workspace_bytes(N)
            lws = torch.empty(lwb, dtype=torch.uint8, device=dev)
            nl, nc = ctypes.c_int64(0), ctypes.c_int64(0)
            self._check(self.lib.ggl_plan_long_count(_ptr(p.rowptr), N, chunk, _ptr(lws), lwb, st,
                                                     ctypes.byref(nl), ctypes.byref(nc)))
            p.n_long, p.n_chunks = int(nl.value), int(nc.value)
            p.long_rows = torch.empty(p.n_long, dtype=torch.int32, device=dev)
            p.chunk_ptr = torch.empty(p.n_long + 1, dtype=torch.int64, device=dev)
            self._check(self.lib.ggl_plan_long_fill(_ptr(p.rowptr), N, chunk, p.n_long,
                                                    _ptr(p.long_rows), _ptr(p.chunk_ptr), _ptr(lws),
                                                    lwb, st))
            p.long_order = self._long_order(p)
            # do the long rows lead the id range (degree-sorted node order)?  One more host read, at plan build only.
            p.hub_first = bool(p.n_long >= 8 and int(p.long_rows[-1]) < 4 * p.n_long)
        # scheduling aid: rows by descending length inside id windows, see _row_order / ggl_segplan.row_order —
        # computed when the plan is launched a second time (SegPlan.c_struct)
        p.row_order, p.uses, p.order_fn = None, 0, (self._row_order if N > 1 else None)
        self.stats["plans_built"] += 1
        p.uid = self.stats["plans_built"]
        return p

    @staticmethod
    def _long_order(p):
        """positions in `long_rows` by descending row length (ggl_segplan.long_order): the serial hub walk starts its
        longest add chains first.  No host read."""
        if not p.n_long:
            return None
        lens = (p.rowptr[1:] - p.rowptr[:-1])[p.long_rows.long()]
        return torch.argsort(lens, descending=True, stable=True).to(torch.int32)

    def _row_order(self, counts):
        """The order rows are handed to lane groups in (kernels where several rows share a wavefront): rows of similar
        length next to each other, so the lanes of a wave finish together (measured K = 64: 7.7 -> 5.9 ms).  A GLOBAL
        sort by length does that but scatters the row ids a workgroup — and the whole chip at any moment — works on
        over the entire graph: on a graph whose node order carries locality (partition.cluster_order, real
        co-purchase / citation graphs) the sources gathered at one time then span every community and L2 keeps
        nothing.  So: rows of `row_order_heavy` elements or more first, longest first (they bound the launch's tail);
        everything else sorted by length inside windows of `row_order_window` consecutive ids, windows in id
        order.  No host read."""
        N = int(counts.shape[0])
        W = int(self.row_order_window)
        if W <= 0 or N <= W:
            return torch.argsort(counts, descending=True, stable=True).to(torch.int32)
        ar = torch.arange(N, device=counts.device, dtype=torch.int64)
        group = torch.where(counts >= int(self.row_order_heavy), torch.zeros_like(ar), 1 + ar // W)
        key = (group << 32) | ((1 << 31) - counts.clamp(max=(1 << 31) - 1))
        return torch.argsort(key, stable=True).to(torch.int32)

    def plan_from_rowptr(self, rowptr, E, max_len=None):
        """SegPlan for elements that are ALREADY grouped by segment (CSR: a sampler's block, a sorted edge
        list): no sort, and no host sync when the caller knows ``max_len`` (e.g. the fan-out)."""
        dev = self._dev(rowptr)
        p = SegPlan()
        p.N, p.E, p.chunk, p.device = int(rowptr.shape[0]) - 1, int(E), int(self.chunk or self.auto_chunk(E)), dev
        p.rowptr = rowptr.contiguous().to(torch.int64)
        p.perm, p.is_sorted = None, True
        p.max_len = int(max_len) if max_len is not None else (int(p.counts().max()) if p.N > 0 else 0)
        p.n_long = p.n_chunks = 0
        p.long_rows = p.chunk_ptr = p.long_order = None
        if p.max_len > p.chunk:
            st = self._stream(dev)
            lwb = self.lib.ggl_plan_long_workspace_bytes(p.N)
            lws = torch.empty(lwb, dtype=torch.uint8, device=dev)
            nl, nc = ctypes.c_int64(0), ctypes.c_int64(0)
            self._check(self.lib.ggl_plan_long_count(_ptr(p.rowptr), p.N, p.chunk, _ptr(lws), lwb, st,
                                                     ctypes.byref(nl), ctypes.byref(nc)))
            p.n_long, p.n_chunks = int(nl.value), int(nc.value)
            p.long_rows = torch.empty(p.n_long, dtype=torch.int32, device=dev)
            p.chunk_ptr = torch.empty(p.n_long + 1, dtype=torch.int64, device=dev)
            self._check(self.lib.ggl_plan_long_fill(_ptr(p.rowptr), p.N, p.chunk, p.n_long,
                                                    _ptr(p.long_rows), _ptr(p.chunk_ptr), _ptr(lws), lwb, st))
            p.long_order = self._long_order(p)
        p.row_order, p.uses, p.order_fn = None, 0, (self._row_order if p.N > 1 else None)
        self.stats["plans_built"] += 1
        p.uid = self.stats["plans_built"]
        return p

    def adopt_plan(self, ids, N, plan):
        """Register a plan built elsewhere (e.g. straight from a sampler's CSR block) for the id tensor
        the layers will pass to unsorted_segment_*: the call then hits the cache — no sort, no sync."""
        self.seg_cache.put(ids, (int(N), self.chunk), plan)

    def segment_reduce(self, x, plan, op="sum"):
        """sum / mean / max of x[E, ...] over an explicit plan (no autograd): the aggregate of a sampled
        block whose edges are already grouped by destination."""
        self._dev(x)
        out, arg = self._segment_fwd(op, x.contiguous(), plan)
        return out if op != "max" else (out, arg)

    def seg_plan(self, ids, N):
        plan = self.seg_cache.get(ids, (int(N), self.chunk))
        if plan is None:
            plan = self.build_plan(ids, N)
            self.seg_cache.put(ids, (int(N), self.chunk), plan)
        else:
            self.stats["plan_hits"] += 1
        return plan

    def graph_plan(self, index, n_dst, n_src=None):
        n_src = n_dst if n_src is None else n_src
        gp = self.graph_cache.get(index, (int(n_dst), int(n_src), self.chunk))
        if gp is None:
            self._dev(index)
            if index.dim() != 2 or index.shape[0] != 2:
                raise RuntimeError("index must have shape [2, num_edges]")
            idx = index if index.dtype == torch.int64 else index.to(torch.int64)
            gp = GraphPlan(self, idx.contiguous(), n_dst, n_src)
            self.graph_cache.put(index, (int(n_dst), int(n_src), self.chunk), gp)
        else:
            self.stats["plan_hits"] += 1
        return gp

    def graph_plan_from_csr(self, row_ptr, col_ind, col_ptr, row_ind, permute, n_rows=None, n_cols=None):
        """GraphPlan of prebuilt CSR + CSC + permutation (FusedGATConv's keyword arguments), cached on `row_ptr`."""
        n_rows = int(row_ptr.shape[0]) - 1 if n_rows is None else int(n_rows)
        n_cols = int(col_ptr.shape[0]) - 1 if n_cols is None else int(n_cols)
        extra = ("csr", n_rows, n_cols, self.chunk) + tuple(_PlanCache.key(t, ()) for t in (col_ind, col_ptr, row_ind, permute))
        gp = self.graph_cache.get(row_ptr, extra)
        if gp is not None and any(r.expired() for r in gp.aux.get("_csr_refs", ())):
            # one of the four other tensors died: its storage address (part of the key) may since have been handed to
            # a different tensor of the same shape — identity + version cannot tell, so this is a miss
            gp = None
        if gp is None:
            gp = GraphPlan.from_csr(self, row_ptr, col_ind, col_ptr, row_ind, permute, n_rows, n_cols)
            gp.aux["_csr_refs"] = tuple(StorageWeakRef(t.untyped_storage()) for t in (col_ind, col_ptr, row_ind, permute))
            self.graph_cache.put(row_ptr, extra, gp)
        else:
            self.stats["plan_hits"] += 1
        return gp

    def gather_i32(self, src_i64, perm):
        dev = src_i64.device
        src_i64 = src_i64.contiguous()
        E = int(src_i64.shape[0])
        out = torch.empty(E, dtype=torch.int32, device=dev)
        self._check(self.lib.ggl_gather_i64_to_i32(_ptr(src_i64), _ptr(perm), E, _ptr(out),
                                                   self._stream(dev)))
        return out

    def _partial(self, plan, dtype, K, with_arg, dev):
        if plan.n_long == 0:
            return None
        nb = self.lib.ggl_partial_bytes(_DTYPE_CODE[dtype], plan.n_chunks, K, 1 if with_arg else 0)
        return torch.empty(nb + 16, dtype=torch.uint8, device=dev)

    # ---- raw (non-autograd) forward launches -------------------------------------------------
    def _segment_fwd(self, op, x, plan):
        dev = x.device
        E = int(x.shape[0])
        if E != plan.E:
            raise IndexError("fisrt dimension of x and index should be same")  # segment_sum_cpu.cpp:17-19
        K = x.numel() // E if E > 0 else int(math.prod(x.shape[1:]))
        out = torch.empty((plan.N,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        st = self._stream(dev)
        code = self._code(x)
        # f16 / bf16 sums accumulate in the storage type (segment_sum_cpu.cpp:47-56): the result depends on
        # the serial order well beyond rounding (a running f16 sum of ones sticks at 2048), so combining
        # chunk partials would not reproduce the reference: those rows are always walked in one piece
        unsplit = op != "max" and x.dtype in (torch.float16, torch.bfloat16)
        # ... but only their ADD chain is serial, not the loads: where the library has the LDS-pipelined hub kernel
        # (GPU build, hub16.hip) the hub rows get a launch of their own, a workgroup per (row, 64-column slab)
        hubs = bool(unsplit and plan.n_long > 0 and self.hub16 and
                    self.lib.ggl_segment_hub16_supported(code, K, _ptr(x), _ptr(out)))
        part = None if unsplit else self._partial(plan, x.dtype, K, op == "max", dev)
        cs = plan.c_struct(part, unsplit=unsplit and not hubs, skip_long=hubs)
        if op in ("sum", "mean"):
            fn = self.lib.ggl_segment_sum if op == "sum" else self.lib.ggl_segment_mean
            side = None
            if hubs:
                # the hub rows' serial add chains (16-24 ns per element: 2.4-3.5 ms for a 147 000-element hub) run BESIDE
                # the launch over all other rows, on a side stream — they write disjoint rows of `out`
                full = plan.c_struct(None)
                if dev.type == "cuda" and self.hub16_overlap:
                    cur = torch.cuda.current_stream(dev)
                    side = self._side_stream(dev)
                    side.wait_stream(cur)
                    self._check(self.lib.ggl_segment_hub16(code, 0 if op == "sum" else 1, _ptr(x), ctypes.byref(full), K,
                                                           _ptr(out), ctypes.c_void_p(side.cuda_stream)))
            self._check(fn(code, _ptr(x), ctypes.byref(cs), K, _ptr(out), st))
            if hubs and side is None:
                self._check(self.lib.ggl_segment_hub16(code, 0 if op == "sum" else 1, _ptr(x), ctypes.byref(full), K,
                                                       _ptr(out), st))
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
            return out, None
        arg = torch.empty((plan.N,) + tuple(x.shape[1:]), dtype=torch.int64, device=dev)
        self._check(self.lib.ggl_segment_max(code, _ptr(x), ctypes.byref(cs), K, _ptr(out), _ptr(arg),
                                             E, st))
        return out, arg

    @staticmethod
    def _row_stride(t, what):
        if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32:
            raise RuntimeError(f"{what} must be a 2-D f32 matrix with unit column stride (a column block is fine)")
        return int(t.stride(0))

    def spmm_sum_into(self, plan, col, w, x, out, accumulate=False):
        """out (+)= A x for 2-D f32 x / out that may be column blocks of wider matrices (row strides are
        passed down: no contiguous copy, no temporary, no separate add) — ggl_spmm_sum_ex."""
        dev = x.device
        K = int(x.shape[1])
        if out.shape[1] != K or out.shape[0] != plan.N:
            raise RuntimeError("out must be [plan rows, x columns]")
        part = self._partial(plan, torch.float32, K, False, dev)
        w, w_by_pos, wp = self._weights(plan, w)
        cs = plan.c_struct(part, wp)
        self._check(self.lib.ggl_spmm_sum_ex(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x),
                                             self._row_stride(x, "x"), K, _ptr(out), self._row_stride(out, "out"),
                                             int(bool(accumulate)), self._stream(dev)))
        return out

    def gather_rows_into(self, src, idx, out):
        """out[i, :] = src[idx[i], :] for 2-D f32 src / out that may be column blocks of wider matrices (one
        kernel, no strided view + index_select + contiguous copy) — ggl_gather_rows_f32_ex."""
        dev = src.device
        n, K = int(idx.shape[0]), int(src.shape[1])
        if out.shape[0] != n or out.shape[1] != K or idx.dtype != torch.int64:
            raise RuntimeError("gather_rows_into: out must be [len(idx), columns of src], idx int64")
        self._check(self.lib.ggl_gather_rows_f32_ex(_ptr(src), self._row_stride(src, "src"), _ptr(idx), n, K,
                                                    _ptr(out), self._row_stride(out, "out"), self._stream(dev)))
        return out

    def spmm_epi_into(self, plan, col, w, x, out, accumulate=False, mean=False, add=None, bias=None, relu=False,
                      p_drop=0.0, rng=None, epi_K=0, col0=0, advance_rng=True):
        """out = dropout(relu(reduce(A x) (+ out) + add + bias)) on column blocks, no autograd — ggl_spmm_epi_ex.
        `bias` [>= col0 + K] is the full-width bias; `add` a [N, K] block view."""
        dev = x.device
        K = int(x.shape[1])
        if out.shape[1] != K or out.shape[0] != plan.N:
            raise RuntimeError("out must be [plan rows, x columns]")
        part = self._partial(plan, torch.float32, K, False, dev)
        w, w_by_pos, wp = self._weights(plan, w)
        cs = plan.c_struct(part, wp)
        b = None
        if bias is not None:
            b = ctypes.c_void_p(bias.data_ptr() + 4 * int(col0))
        self._check(self.lib.ggl_spmm_epi_ex(
            ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), self._row_stride(x, "x"), K, _ptr(out),
            self._row_stride(out, "out"), int(bool(accumulate)), int(bool(mean)), _ptr(add),
            (self._row_stride(add, "add") if add is not None else 0), b, int(bool(relu)), float(p_drop),
            _ptr(rng), int(epi_K), int(col0), int(bool(advance_rng)), self._stream(dev)))
        return out

    def segment_sum_into(self, x, plan, out, accumulate=False):
        """out (+)= segment_sum(x) over `plan`, same strided / in-place conventions — ggl_segment_sum_ex."""
        dev = x.device
        K = int(x.shape[1])
        if int(x.shape[0]) != plan.E or out.shape[1] != K or out.shape[0] != plan.N:
            raise IndexError("segment_sum_into: shapes do not match the plan")
        part = self._partial(plan, torch.float32, K, False, dev)
        cs = plan.c_struct(part)
        self._check(self.lib.ggl_segment_sum_ex(self._code(x), _ptr(x), self._row_stride(x, "x"), ctypes.byref(cs), K,
                                                _ptr(out), self._row_stride(out, "out"), int(bool(accumulate)),
                                                self._stream(dev)))
        return out

    def _spmm_fwd(self, op, plan, col, w, x, n_out, perm_override=None, aux=None, gp=None):
        """op in sum/mean/max/mean_bwd/max_bwd.  x [N_in, *]; returns out [n_out, *] (+argsrc)."""
        dev = x.device
        K = int(math.prod(x.shape[1:]))
        if op in ("sum", "mean", "max") and x.dim() == 2:
            # ggl_policy_spmm_width (include/ggl_mpops.h): rows wider than 256 columns that are not whole cache lines
            # (602 Reddit features) -> a multiple of 64 for the 64-column blocks (products-sized K = 602: 55 -> 42 ms);
            # class-count widths (47, 41, 7 ...) -> a multiple of 4 for the float4 kernels (K = 47: 9.3 -> 5.8 ms); the
            # max walk of wide rows that are not 16-byte pieces -> a multiple of 4 (K = 602: 69 -> 60 ms).  One
            # zero-padded copy of x; the pad columns (values and argmax) are dropped.
            Kp = int(self.lib.ggl_policy_spmm_width(1 if op == "max" else 0, K, plan.E, int(x.shape[0])))
            if Kp != K:
                xp = torch.nn.functional.pad(x, (0, Kp - K))
                out, arg = self._spmm_fwd(op, plan, col, w, xp, n_out, perm_override, aux)
                return out[:, :K].contiguous(), (arg[:, :K].contiguous() if arg is not None else None)
        out = torch.empty((n_out,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
        st = self._stream(dev)
        part = self._partial(plan, torch.float32, K, op == "max", dev)
        w, w_by_pos, wp = self._weights(plan, w, perm_override)
        cs = plan.c_struct(part, wp)
        L = self.lib
        if op == "sum":
            self._check(L.ggl_spmm_sum(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), K,
                                       _ptr(out), st))
        elif op == "mean":
            self._check(L.ggl_spmm_mean(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), K,
                                        _ptr(out), st))
        elif op == "max":
            arg = torch.empty(out.shape, dtype=torch.int64, device=dev)
            self._check(L.ggl_spmm_max(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), K,
                                       _ptr(out), _ptr(arg), st))
            return out, arg
        elif op == "mean_bwd":
            if self.mean_bwd_prescale and x.dim() == 2 and self.lib.ggl_policy_mean_bwd_prescale(plan.E, int(x.shape[0])):
                # gx[src] += (g[dst] / count[dst]) * w (spmm_mean_cpu.cpp:90-100): the division depends on the
                # destination row only, so it is done ONCE per row (the same rounded divide on the same operands) and
                # the walk is the plain transposed SpMM-sum — no per-edge degree lookup (a random 16-byte read, i.e. one
                # more line per edge), and the 64-column blocks apply: products-sized K = 256 23.0 -> 15.5 ms, same bits
                cnt = (aux[1:] - aux[:-1]).clamp(min=1).to(torch.float32).unsqueeze(1)
                xs = x / cnt
                self._check(L.ggl_spmm_sum(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(xs), K,
                                           _ptr(out), st))
            else:
                self._check(L.ggl_spmm_mean_bwd(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x),
                                                _ptr(aux), K, _ptr(out), st))
        elif op == "max_bwd":
            # the form is the library's decision (ggl_policy_maxbwd_form): the winner mask is an E x K/8-byte transient, taken only
            # where it is smaller than a multiple of the witness matrix it replaces and that matrix is not cache-resident
            form = int(L.ggl_policy_maxbwd_form(plan.E, int(aux.shape[0]), K)) if gp is not None else (
                1 if int(L.ggl_get_option(b"maxbwd_arg32")) else 0)
            mask = None
            if form == 2:
                try:
                    mask = torch.empty(int(L.ggl_spmm_max_mask_bytes(plan.E, K)) // 4 + 4, dtype=torch.int32, device=dev)
                except torch.OutOfMemoryError:
                    form = 1                    # no room for the transient: the int32 witness copy ([N, K] x 4 bytes)
            if form == 2:
                # a 1-bit winner mask built in DESTINATION order (the witness row is wave-uniform there), read in the
                # transposed walk's own order: K / 8 bytes per edge instead of 8K (include/ggl_mpops.h)
                fs = gp.fwd.c_struct(None)
                # records in forward order (coalesced writes; the walk reads record posT[t]) unless the A/B knob asks for
                # the scatter to transposed positions (maxbwd_mask_scatter: streamed reads, slower writes)
                scatter = int(L.ggl_get_option(b"maxbwd_mask_scatter")) != 0
                self._check(L.ggl_spmm_max_mask(ctypes.byref(fs), _ptr(gp.col), _ptr(gp.tpos) if scatter else None, _ptr(aux), K,
                                                _ptr(mask), st))
                self._check(L.ggl_spmm_max_bwd_mask(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), _ptr(mask),
                                                    None if scatter else _ptr(gp.posT), K, _ptr(out), st))
            elif form == 1:   # witnesses from a compact int32 copy (one [N, K] pass)
                aux32 = aux.to(torch.int32)
                self._check(L.ggl_spmm_max_bwd32(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x),
                                                 _ptr(aux32), K, _ptr(out), st))
            else:
                self._check(L.ggl_spmm_max_bwd(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x),
                                               _ptr(aux), K, _ptr(out), st))
        else:
            raise ValueError(op)
        return out, None

    def _weights(self, plan, w, perm_override=None):
        """(weights, w_by_pos, perm for the launch struct) of one walk over `plan`: weights that came back a second time
        are streamed in sorted order (`_sorted_weights`); otherwise the kernel reads w[perm[p]] — through the plan's own
        permutation, or, on the CSC side of a CSR-built plan, through `wperm` (GraphPlan.from_csr)."""
        if perm_override is not None or w is None:
            return w, 0, perm_override
        wp = plan.perm if plan.perm is not None else getattr(plan, "wperm", None)
        if wp is None:
            return w, 0, None
        w, by_pos = self._sorted_weights(plan, w, wp)
        return w, by_pos, (wp if plan.perm is None else None)

    def _sorted_weights(self, plan, w, perm=None):
        """Edge weights in the plan's sorted order, for weights that are REUSED.

        The kernels can read w[perm[p]] themselves (one random 4-byte read per edge).  A weight tensor
        seen for the second time on the same plan (GCN: the same normalisation weights feed every layer,
        forward and backward, every step) is gathered once into sorted order and cached by the
        identity + version of its storage, after which every launch streams it (w_by_pos = 1)."""
        key_extra = (plan.uid,)
        hit = self.w_cache.get(w, key_extra)
        if hit is None:
            self.w_cache.put(w, key_extra, False)  # first sight: remember it, gather in-kernel
            return w, 0
        if hit is False:
            E = int(plan.E)
            H = w.numel() // E if E > 0 else 1
            ws = torch.empty_like(w)
            self._check(self.lib.ggl_gather_rows_f32(_ptr(w), _ptr(plan.perm if perm is None else perm), E, max(H, 1), _ptr(ws),
                                                     self._stream(w.device)))
            self.w_cache.put(w, key_extra, ws)
            hit = ws
        return hit, 1

    def _bspmm_fwd(self, plan, col, w, x, n_out, perm_override=None):
        dev = x.device
        H, C = int(x.shape[1]), int(x.shape[2])
        out = torch.empty((n_out, H, C), dtype=torch.float32, device=dev)
        part = self._partial(plan, torch.float32, H * C, False, dev)
        w, w_by_pos, wp = self._weights(plan, w, perm_override)
        cs = plan.c_struct(part, wp)
        self._check(self.lib.ggl_bspmm_sum(ctypes.byref(cs), _ptr(col), _ptr(w), w_by_pos, _ptr(x), H, C,
                                           _ptr(out), self._stream(dev)))
        return out

    @staticmethod
    def _check_f32(name, t):
        if t.dtype != torch.float32:
            # spmm_sum_cpu.cpp:22 data_ptr<float>() on a non-float tensor
            raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")

    # ---- autograd Functions (closures over this engine) ---------------------------------------
    def _make_functions(self):
        eng = self

        class SegmentSum(torch.autograd.Function):  # src/segment_sum.cpp:35-54
            @staticmethod
            def forward(ctx, x, ids, N):
                plan = eng.seg_plan(ids, N)
                out, _ = eng._segment_fwd("sum", x, plan)
                ctx.save_for_backward(ids)
                ctx.x_shape = x.shape
                return out

            @staticmethod
            def backward(ctx, g):
                (ids,) = ctx.saved_tensors
                g = g.contiguous()
                E = ctx.x_shape[0]
                K = int(math.prod(ctx.x_shape[1:]))
                gin = torch.empty(ctx.x_shape, dtype=g.dtype, device=g.device)
                eng._check(eng.lib.ggl_segment_sum_bwd(eng._code(g), _ptr(g), _ptr(ids), E, K,
                                                       _ptr(gin), eng._stream(g.device)))
                return gin, None, None

        class SegmentMean(torch.autograd.Function):  # src/segment_mean.cpp:36-63
            @staticmethod
            def forward(ctx, x, ids, N):
                plan = eng.seg_plan(ids, N)
                out, _ = eng._segment_fwd("mean", x, plan)
                ctx.save_for_backward(ids, plan.rowptr)
                ctx.x_shape = x.shape
                return out

            @staticmethod
            def backward(ctx, g):
                ids, rowptr = ctx.saved_tensors
                g = g.contiguous()
                if g.dtype not in _FLOAT_DTYPES:
                    raise RuntimeError("segment_mean backward needs a floating dtype")
                E = ctx.x_shape[0]
                K = int(math.prod(ctx.x_shape[1:]))
                gin = torch.empty(ctx.x_shape, dtype=g.dtype, device=g.device)
                eng._check(eng.lib.ggl_segment_mean_bwd(eng._code(g), _ptr(g), _ptr(ids), _ptr(rowptr),
                                                        E, K, _ptr(gin), eng._stream(g.device)))
                return gin, None, None

        class SegmentMax(torch.autograd.Function):  # src/segment_max.cpp:37-61
            @staticmethod
            def forward(ctx, x, ids, N):
                plan = eng.seg_plan(ids, N)
                out, arg = eng._segment_fwd("max", x, plan)
                ctx.save_for_backward(arg)
                ctx.x_shape = x.shape
                ctx.mark_non_differentiable(arg)
                return out, arg

            @staticmethod
            def backward(ctx, g, _garg):
                (arg,) = ctx.saved_tensors
                g = g.contiguous()
                E = ctx.x_shape[0]
                K = int(math.prod(ctx.x_shape[1:]))
                gin = torch.empty(ctx.x_shape, dtype=g.dtype, device=g.device)
                eng._check(eng.lib.ggl_segment_max_bwd(eng._code(g), _ptr(g), _ptr(arg), E,
                                                       int(arg.shape[0]), K, _ptr(gin),
                                                       eng._stream(g.device)))
                return gin, None, None

        class SpMMSum(torch.autograd.Function):  # src/gspmm.cpp:26-80
            @staticmethod
            def forward(ctx, gp, w, x):
                out, _ = eng._spmm_fwd("sum", gp.fwd, gp.col, w, x, gp.N_dst)
                ctx.gp, ctx.w = gp, w
                return out

            @staticmethod
            def backward(ctx, g):
                gp = ctx.gp
                gx, _ = eng._spmm_fwd("sum", gp.bwd, gp.colT, ctx.w, g.contiguous(), gp.N_src)
                return None, None, gx  # weight is non-differentiable in the reference (gspmm.cpp:30)

        class SpMMMean(torch.autograd.Function):  # src/gspmm.cpp:82-141
            @staticmethod
            def forward(ctx, gp, w, x):
                out, _ = eng._spmm_fwd("mean", gp.fwd, gp.col, w, x, gp.N_dst)
                ctx.gp, ctx.w = gp, w
                return out

            @staticmethod
            def backward(ctx, g):
                gp = ctx.gp
                gx, _ = eng._spmm_fwd("mean_bwd", gp.bwd, gp.colT, ctx.w, g.contiguous(), gp.N_src,
                                      aux=gp.fwd.rowptr)
                return None, None, gx

        class SpMMMax(torch.autograd.Function):  # src/gspmm.cpp:143-202
            @staticmethod
            def forward(ctx, gp, w, x):
                out, arg = eng._spmm_fwd("max", gp.fwd, gp.col, w, x, gp.N_dst)
                ctx.gp, ctx.w = gp, w
                ctx.save_for_backward(arg)
                return out

            @staticmethod
            def backward(ctx, g):
                gp = ctx.gp
                (arg,) = ctx.saved_tensors
                gx, _ = eng._spmm_fwd("max_bwd", gp.bwd, gp.colT, ctx.w, g.contiguous(), gp.N_src,
                                      aux=arg, gp=gp)
                return None, None, gx

        class BSpMMSum(torch.autograd.Function):  # src/gspmm.cpp:204-260
            @staticmethod
            def forward(ctx, gp, w, x):
                out = eng._bspmm_fwd(gp.fwd, gp.col, w, x, gp.N_dst)
                ctx.gp = gp
                ctx.save_for_backward(w, x)
                return out

            @staticmethod
            def backward(ctx, g):
                gp = ctx.gp
                w, x = ctx.saved_tensors
                g = g.contiguous()
                gx = eng._bspmm_fwd(gp.bwd, gp.colT, w, g, gp.N_src)
                H, C = int(x.shape[1]), int(x.shape[2])
                gw = torch.empty_like(w)
                # a plan built from the caller's CSR has no COO edge list to walk (gp.index is None): it always takes the
                # sorted route, whose plain kernel covers any channel count
                if gp.index is None or (eng.gradw_sorted and eng.lib.ggl_policy_gradw_sorted(H, C)):
                    # along the destination-sorted forward plan, strips staged through LDS (edgedot.hip): the g rows
                    # of a batch are a handful of rows, only x[src] is a random gather — and a coalesced one
                    sb = eng.lib.ggl_bspmm_grad_w_sorted_scratch_bytes(gp.E, gp.N_dst, H, C)
                    scratch = torch.empty(sb // 4, dtype=torch.float32, device=g.device) if sb else None
                    cs = gp.fwd.c_struct(None)
                    eng._check(eng.lib.ggl_bspmm_grad_w_sorted(ctypes.byref(cs), _ptr(gp.col), _ptr(gp.rowidx), _ptr(x),
                                                               _ptr(g), H, C, _ptr(gw), _ptr(scratch),
                                                               eng._stream(g.device)))
                else:
                    eng._check(eng.lib.ggl_bspmm_grad_w(_ptr(gp.index), _ptr(x), _ptr(g), gp.E, H, C,
                                                        _ptr(gw), eng._stream(g.device)))
                # the reference returns grad_weight although it marked weight non-differentiable
                # (gspmm.cpp:208,259): w.grad is populated there, and here.
                return None, gw, gx

        class GATFused(torch.autograd.Function):
            """edge-softmax + aggregate in one kernel (gat_conv.py:103-112 + softmax.py:29-35)."""

            @staticmethod
            def forward(ctx, gp, el, er, x, slope, p_drop=0.0):
                dev = x.device
                N, H, C = gp.N_dst, int(x.shape[1]), int(x.shape[2])
                out = torch.empty((N, H, C), dtype=torch.float32, device=dev)
                rmax = torch.empty((N, H), dtype=torch.float32, device=dev)
                rden = torch.empty((N, H), dtype=torch.float32, device=dev)
                part = None
                if gp.fwd.n_long > 0:
                    part = torch.empty(eng.lib.ggl_gat_partial_bytes(gp.fwd.n_chunks, H, C) + 16,
                                       dtype=torch.uint8, device=dev)
                cs = gp.fwd.c_struct(part)
                rng = rng_used = None
                if p_drop > 0:
                    rng = eng._rng_state(dev)
                    rng_used = rng.clone()  # the {seed, offset} this launch reads; the backward redraws the mask
                fast = bool(eng.gat_fast and eng.lib.ggl_gat_fast_supported(H, C))
                if fast:
                    eng._check(eng.lib.ggl_gat_fast_fwd(ctypes.byref(cs), _ptr(gp.col), _ptr(el), _ptr(er), _ptr(x),
                                                        int(x.shape[0]), float(slope), H, C, float(p_drop), _ptr(rng),
                                                        _ptr(out), _ptr(rmax), _ptr(rden), eng._stream(dev)))
                else:
                    eng._check(eng.lib.ggl_gat_fused_fwd(ctypes.byref(cs), _ptr(gp.col), _ptr(el), _ptr(er),
                                                         _ptr(x), float(slope), H, C, float(p_drop), _ptr(rng),
                                                         _ptr(out), _ptr(rmax), _ptr(rden), eng._stream(dev)))
                ctx.fast = fast
                ctx.gp, ctx.slope, ctx.p_drop, ctx.rng_used = gp, float(slope), float(p_drop), rng_used
                ctx.save_for_backward(el, er, x, out, rmax, rden)
                return out

            @staticmethod
            def backward(ctx, g):
                gp = ctx.gp
                el, er, x, out, rmax, rden = ctx.saved_tensors
                g = g.contiguous()
                dev = g.device
                H, C = int(x.shape[1]), int(x.shape[2])
                st = eng._stream(dev)
                if ctx.fast:  # both walks recompute alpha / de from per-row constants: no [E, H, 2] buffer
                    bwd = gp.bwd
                    stats = torch.empty((gp.N_dst, H, 4), dtype=torch.float32, device=dev)
                    ger = torch.empty_like(er)
                    gx = torch.empty((gp.N_src, H, C), dtype=torch.float32, device=dev)
                    gel = torch.empty((gp.N_src, H), dtype=torch.float32, device=dev)
                    part_f = eng._partial(gp.fwd, torch.float32, 8 * H, False, dev)   # four double sums per chunk and head
                    part_t = eng._partial(bwd, torch.float32, H * C + H, False, dev)
                    cs, csT = gp.fwd.c_struct(part_f), bwd.c_struct(part_t)
                    posT = gp.posT if ctx.p_drop > 0 else None
                    eng._check(eng.lib.ggl_gat_fast_bwd(
                        ctypes.byref(cs), _ptr(gp.col), ctypes.byref(csT), _ptr(gp.colT), _ptr(posT), _ptr(el),
                        _ptr(er), _ptr(x), _ptr(g), _ptr(out), _ptr(rmax), _ptr(rden), ctx.slope, H, C, ctx.p_drop,
                        _ptr(ctx.rng_used), _ptr(stats), _ptr(gx), _ptr(gel), _ptr(ger), st))
                    return None, gel, ger, gx, None, None
                # alpha and de interleaved [E, H, 2]: the source-side walk fetches both with one 64-byte line
                ad = torch.empty((max(gp.E, 1), H, 2), dtype=torch.float32, device=dev)
                alpha, de = ad.data_ptr(), ad.data_ptr() + 4
                ger = torch.empty_like(er)
                part_f = eng._partial(gp.fwd, torch.float32, H, False, dev)  # must outlive the launch
                cs = gp.fwd.c_struct(part_f)
                eng._check(eng.lib.ggl_gat_fused_bwd_dst(
                    ctypes.byref(cs), _ptr(gp.col), None, _ptr(el), _ptr(er), _ptr(x),
                    _ptr(g), _ptr(out), _ptr(rmax), _ptr(rden), ctx.slope, H, C, ctx.p_drop,
                    _ptr(ctx.rng_used), alpha, de, _ptr(ger), None, st))
                bwd = gp.bwd
                gx = torch.empty((gp.N_src, H, C), dtype=torch.float32, device=dev)
                gel = torch.empty((gp.N_src, H), dtype=torch.float32, device=dev)
                part = eng._partial(bwd, torch.float32, H * C + H, False, dev)  # gx and gel partials of long rows
                csT = bwd.c_struct(part)
                eng._check(eng.lib.ggl_gat_fused_bwd_src(ctypes.byref(csT), _ptr(gp.colT), _ptr(gp.posT),
                                                         alpha, de, _ptr(g), H, C,
                                                         _ptr(gx), _ptr(gel), st))
                return None, gel, ger, gx, None, None

        class BiasAdd(torch.autograd.Function):
            """out = x + bias (bias broadcast over rows); d bias = column sums of the gradient."""

            @staticmethod
            def forward(ctx, x, bias):
                ctx.bias_shape = bias.shape
                return x + bias

            @staticmethod
            def backward(ctx, g):
                gb = eng.colsum(g.reshape(g.shape[0], -1)).reshape(ctx.bias_shape)
                return g, gb

        class BiasAct(torch.autograd.Function):
            """y = dropout(relu(a + bias)) in one kernel; backward rebuilds the mask from y and reduces
            the bias gradient in the same pass (csrc/epilogue.hip)."""

            @staticmethod
            def forward(ctx, a, bias, relu, p_drop):
                a = a.contiguous()
                dev = a.device
                N = int(a.shape[0])
                K = a.numel() // N if N > 0 else int(math.prod(a.shape[1:]))
                y = torch.empty_like(a)
                rng = eng._rng_state(dev) if p_drop > 0 else None
                rng_used = rng.clone() if rng is not None else None  # the {seed, offset} this launch reads
                b = bias.contiguous().reshape(-1) if bias is not None else None
                eng._check(eng.lib.ggl_bias_act_fwd(_ptr(a), _ptr(b), N, K, int(relu), float(p_drop),
                                                    _ptr(rng), _ptr(y), eng._stream(dev)))
                ctx.cfg = (N, K, int(relu), float(p_drop), None if bias is None else bias.shape)
                ctx.rng_used = rng_used
                ctx.save_for_backward(y)
                return y

            @staticmethod
            def backward(ctx, g):
                (y,) = ctx.saved_tensors
                N, K, relu, p_drop, bshape = ctx.cfg
                g = g.contiguous()
                dev = g.device
                ga = torch.empty_like(g)
                gb = torch.empty(K, dtype=torch.float32, device=dev) if bshape is not None else None
                wsb = eng.lib.ggl_bias_act_bwd_workspace_bytes(N, K)
                ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
                eng._check(eng.lib.ggl_bias_act_bwd(_ptr(g), _ptr(y), N, K, relu, p_drop, _ptr(ctx.rng_used),
                                                    _ptr(ga), _ptr(gb), _ptr(ws), wsb, eng._stream(dev)))
                return ga, (gb.reshape(bshape) if gb is not None else None), None, None

        class SpMMSumBiasAct(torch.autograd.Function):
            """y = dropout(relu(A x + bias)) in ONE kernel: the epilogue is applied to each finished row of
            the SpMM in registers (reduce.hip MODE_SPMM_EPI); backward = bias_act_bwd, transposed SpMM."""

            @staticmethod
            def forward(ctx, gp, w, x, bias, relu, p_drop):
                dev = x.device
                K = int(x.shape[1])
                plan = gp.fwd
                y = torch.empty((gp.N_dst, K), dtype=torch.float32, device=dev)
                part = eng._partial(plan, torch.float32, K, False, dev)
                ww, w_by_pos, wp = eng._weights(plan, w)
                cs = plan.c_struct(part, wp)
                rng = eng._rng_state(dev) if p_drop > 0 else None
                ctx.rng_used = rng.clone() if rng is not None else None
                b = bias.contiguous().reshape(-1) if bias is not None else None
                eng._check(eng.lib.ggl_spmm_sum_bias_act(ctypes.byref(cs), _ptr(gp.col), _ptr(ww), w_by_pos,
                                                         _ptr(x), K, _ptr(b), int(relu), float(p_drop),
                                                         _ptr(rng), _ptr(y), eng._stream(dev)))
                ctx.gp, ctx.w = gp, w
                ctx.cfg = (int(gp.N_dst), K, int(relu), float(p_drop), None if bias is None else bias.shape)
                ctx.save_for_backward(y)
                return y

            @staticmethod
            def backward(ctx, g):
                (y,) = ctx.saved_tensors
                N, K, relu, p_drop, bshape = ctx.cfg
                g = g.contiguous()
                dev = g.device
                ga = torch.empty_like(g)
                gb = torch.empty(K, dtype=torch.float32, device=dev) if bshape is not None else None
                wsb = eng.lib.ggl_bias_act_bwd_workspace_bytes(N, K)
                ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
                eng._check(eng.lib.ggl_bias_act_bwd(_ptr(g), _ptr(y), N, K, relu, p_drop, _ptr(ctx.rng_used),
                                                    _ptr(ga), _ptr(gb), _ptr(ws), wsb, eng._stream(dev)))
                gp = ctx.gp
                gx = None
                if ctx.needs_input_grad[2]:
                    gx, _ = eng._spmm_fwd("sum", gp.bwd, gp.colT, ctx.w, ga, gp.N_src)
                return None, None, gx, (gb.reshape(bshape) if gb is not None else None), None, None

        def _epi_backward(ctx, g, y):
            """Through dropout / ReLU / + bias / + add in one pass: returns (ga, gbias)."""
            N, K, relu, p_drop, bshape = ctx.cfg
            g = g.contiguous()
            dev = g.device
            if not (relu or p_drop > 0 or bshape is not None):
                return g, None
            ga = torch.empty_like(g)
            gb = torch.empty(K, dtype=torch.float32, device=dev) if bshape is not None else None
            wsb = eng.lib.ggl_bias_act_bwd_workspace_bytes(N, K)
            ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
            eng._check(eng.lib.ggl_bias_act_bwd(_ptr(g), _ptr(y), N, K, relu, p_drop, _ptr(ctx.rng_used),
                                                _ptr(ga), _ptr(gb), _ptr(ws), wsb, eng._stream(dev)))
            return ga, (gb.reshape(bshape) if gb is not None else None)

        class SpMMEpi(torch.autograd.Function):
            """y = dropout(relu(reduce(A x) + add + bias)), reduce = sum | mean, in ONE kernel (ggl_spmm_epi_ex):
            GCNConv's "+ bias" (gcn_conv.py:105-106) and SAGEConv's "mean + fc_self(x_dst) + bias -> act"
            (sage_conv.py:100-108) applied to each finished row in registers."""

            @staticmethod
            def forward(ctx, gp, w, x, mean, add, bias, relu, p_drop):
                dev = x.device
                K = int(x.shape[1])
                y = torch.empty((gp.N_dst, K), dtype=torch.float32, device=dev)
                rng = eng._rng_state(dev) if p_drop > 0 else None
                ctx.rng_used = rng.clone() if rng is not None else None
                b = bias.contiguous().reshape(-1) if bias is not None else None
                a = add.contiguous() if add is not None else None
                eng.spmm_epi_into(gp.fwd, gp.col, w, x, y, mean=mean, add=a, bias=b, relu=relu, p_drop=p_drop, rng=rng)
                ctx.gp, ctx.w, ctx.mean, ctx.has_add = gp, w, bool(mean), add is not None
                ctx.cfg = (int(gp.N_dst), K, int(relu), float(p_drop), None if bias is None else bias.shape)
                ctx.save_for_backward(y)
                return y

            @staticmethod
            def backward(ctx, g):
                (y,) = ctx.saved_tensors
                ga, gb = _epi_backward(ctx, g, y)
                gp = ctx.gp
                gx = None
                if ctx.needs_input_grad[2]:
                    if ctx.mean:
                        gx, _ = eng._spmm_fwd("mean_bwd", gp.bwd, gp.colT, ctx.w, ga, gp.N_src, aux=gp.fwd.rowptr)
                    else:
                        gx, _ = eng._spmm_fwd("sum", gp.bwd, gp.colT, ctx.w, ga, gp.N_src)
                return None, None, gx, None, (ga if ctx.has_add else None), gb, None, None

        class SegmentEpi(torch.autograd.Function):
            """The same epilogue on segment_sum / segment_mean of f32 messages x[E, K] (ggl_segment_epi): the
            message() + aggregate() route of a sampled SAGEConv block."""

            @staticmethod
            def forward(ctx, x, ids, N, mean, add, bias, relu):
                dev = x.device
                plan = eng.seg_plan(ids, N)
                K = int(x.shape[1])
                if int(x.shape[0]) != plan.E:
                    raise IndexError("fisrt dimension of x and index should be same")
                y = torch.empty((plan.N, K), dtype=torch.float32, device=dev)
                part = eng._partial(plan, torch.float32, K, False, dev)
                cs = plan.c_struct(part)
                b = bias.contiguous().reshape(-1) if bias is not None else None
                a = add.contiguous() if add is not None else None
                eng._check(eng.lib.ggl_segment_epi(_ptr(x), ctypes.byref(cs), K, int(bool(mean)), _ptr(a), 0, _ptr(b),
                                                   int(bool(relu)), 0.0, None, _ptr(y), eng._stream(dev)))
                ctx.mean, ctx.has_add, ctx.x_shape, ctx.rng_used = bool(mean), add is not None, x.shape, None
                ctx.cfg = (int(plan.N), K, int(relu), 0.0, None if bias is None else bias.shape)
                ctx.save_for_backward(y, ids, plan.rowptr)
                return y

            @staticmethod
            def backward(ctx, g):
                y, ids, rowptr = ctx.saved_tensors
                ga, gb = _epi_backward(ctx, g, y)
                gx = None
                if ctx.needs_input_grad[0]:
                    E, K = int(ctx.x_shape[0]), int(ctx.x_shape[1])
                    gx = torch.empty(ctx.x_shape, dtype=ga.dtype, device=ga.device)
                    if ctx.mean:
                        eng._check(eng.lib.ggl_segment_mean_bwd(eng._code(ga), _ptr(ga), _ptr(ids), _ptr(rowptr), E, K,
                                                                _ptr(gx), eng._stream(ga.device)))
                    else:
                        eng._check(eng.lib.ggl_segment_sum_bwd(eng._code(ga), _ptr(ga), _ptr(ids), E, K, _ptr(gx),
                                                               eng._stream(ga.device)))
                return gx, None, None, None, (ga if ctx.has_add else None), gb, None

        class GATHeadMean(torch.autograd.Function):
            """y_i = 1/H sum_h sum_j alpha_ijh (x_j W_h) for a head-averaging GAT layer (gat_conv.py:98-122 with
            concat=False), aggregated BEFORE it is transformed: y_i = 1/H (sum_j alpha_ijh x_j) W_h, logits from
            el = x (W a_src), er = x (W a_dst).  The three walks gather the F-float input row / the C-float output
            gradient instead of the H x C transformed row (gat.hip, ggl_gat_sh_*); everything dense runs as GEMMs."""

            @staticmethod
            def forward(ctx, gp, x, W, att, slope, p_drop):
                dev = x.device
                N, F = int(x.shape[0]), int(x.shape[1])
                H = 8
                C = int(W.shape[1]) // H
                Wr = W.view(F, H, C)
                a_src, a_dst = att[0, :, :C], att[0, :, C:]
                U, V = (Wr * a_src).sum(-1), (Wr * a_dst).sum(-1)           # [F, H]
                el, er = (x @ U).contiguous(), (x @ V).contiguous()         # [N, H]
                rowmax = torch.empty((N, H), dtype=torch.float32, device=dev)
                den = torch.empty((N, H), dtype=torch.float32, device=dev)
                A = torch.empty((N, H, F), dtype=torch.float32, device=dev)
                part = None
                if gp.fwd.n_long > 0:
                    part = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(gp.fwd.n_chunks, F) + 16, dtype=torch.uint8,
                                       device=dev)
                cs = gp.fwd.c_struct(part)
                rng = rng_used = None
                if p_drop > 0:
                    rng = eng._rng_state(dev)
                    rng_used = rng.clone()
                eng._check(eng.lib.ggl_gat_sh_fwd(ctypes.byref(cs), _ptr(gp.col), _ptr(el), _ptr(er), _ptr(x), F,
                                                  float(slope), float(p_drop), _ptr(rng), _ptr(rowmax), _ptr(A),
                                                  _ptr(den), eng._stream(dev)))
                Wst = Wr.permute(1, 0, 2).reshape(H * F, C)
                y = (A.view(N, H * F) @ Wst) / H
                ctx.gp, ctx.slope, ctx.p_drop, ctx.rng_used = gp, float(slope), float(p_drop), rng_used
                ctx.save_for_backward(x, W, att, el, er, rowmax, den, A)
                return y

            @staticmethod
            def backward(ctx, gy):
                gp = ctx.gp
                x, W, att, el, er, rowmax, den, A = ctx.saved_tensors
                dev = gy.device
                N, F = int(x.shape[0]), int(x.shape[1])
                H = 8
                C = int(W.shape[1]) // H
                Cp = C + (-C) % 4
                Wr = W.view(F, H, C)
                a_src, a_dst = att[0, :, :C], att[0, :, C:]
                U, V = (Wr * a_src).sum(-1), (Wr * a_dst).sum(-1)
                Wst = Wr.permute(1, 0, 2).reshape(H * F, C)
                gyh = gy.contiguous() / H
                gyp = torch.nn.functional.pad(gyh, (0, Cp - C)).contiguous()
                G = (gyh @ Wst.t()).view(N, H, F).contiguous()               # dL/dA
                # {er, m, 1 / (den + 1e-16), <G, A>} per (row, head) in one pass (was: product + reduce + reciprocal + stack)
                stats = torch.empty((N, H, 4), dtype=torch.float32, device=dev)
                eng._check(eng.lib.ggl_gat_sh_stats(_ptr(er), _ptr(rowmax), _ptr(den), _ptr(G), _ptr(A), N, F, _ptr(stats),
                                                    eng._stream(dev)))
                z = torch.nn.functional.pad((x @ W).view(N, H, C), (0, Cp - C)).contiguous()
                ger = torch.empty((N, H), dtype=torch.float32, device=dev)
                gel = torch.empty((N, H), dtype=torch.float32, device=dev)
                T = torch.empty((N, H, Cp), dtype=torch.float32, device=dev)
                bwd = gp.bwd
                # forward plan's partial: four doubles per hub chunk and head (the destination walk's row sums, round 6)
                part_f = None
                if gp.fwd.n_long > 0:
                    part_f = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(gp.fwd.n_chunks, 8) + 16, dtype=torch.uint8, device=dev)
                part_t = None
                if bwd.n_long > 0:
                    part_t = torch.empty(eng.lib.ggl_gat_sh_partial_bytes(bwd.n_chunks, Cp) + 16, dtype=torch.uint8,
                                         device=dev)
                cs, csT = gp.fwd.c_struct(part_f), bwd.c_struct(part_t)
                posT = gp.posT if ctx.p_drop > 0 else None
                eng._check(eng.lib.ggl_gat_sh_bwd(ctypes.byref(cs), _ptr(gp.col), ctypes.byref(csT), _ptr(gp.colT),
                                                  _ptr(posT), _ptr(el), _ptr(x), F, _ptr(G), _ptr(stats), _ptr(z),
                                                  _ptr(gyp), Cp, ctx.slope, ctx.p_drop, _ptr(ctx.rng_used), _ptr(ger),
                                                  _ptr(T), _ptr(gel), eng._stream(dev)))
                gx = torch.einsum("nhc,fhc->nf", T[:, :, :C], Wr) + gel @ U.t() + ger @ V.t()
                # reductions over the N nodes: slab-split two-level sums (dense.wgrad), not one GEMM with an N-long accumulation per
                # element (round 6: a tuned kernel choice for the latter left these 1e-3 from a float64 evaluation)
                from .dense import wgrad

                gU, gV = wgrad(x, gel), wgrad(x, ger)                       # x^T gel, x^T ger: [F, H]
                gW = wgrad(A.view(N, H * F), gyh).view(H, F, C).permute(1, 0, 2) \
                    + gU.unsqueeze(-1) * a_src + gV.unsqueeze(-1) * a_dst
                gatt = torch.cat([torch.einsum("fh,fhc->hc", gU, Wr), torch.einsum("fh,fhc->hc", gV, Wr)], dim=-1)
                return None, gx, gW.reshape(F, H * C), gatt.unsqueeze(0), None, None

        self.GATHeadMean = GATHeadMean
        class BlockMeanEpi(torch.autograd.Function):
            """relu(mean_{j in block row i} x[j] + add_i + bias) over a sampler Block (static capacities,
            device-side sizes): forward = the rectangular SpMM-mean with the epilogue in its store; backward
            = the MEANBWD walk of the block's CSC, which is built on the device without a host read."""

            @staticmethod
            def forward(ctx, x, blk, add, bias, relu):
                dev = x.device
                K = int(x.shape[1])
                if int(x.shape[0]) != blk.n_src_cap:
                    raise RuntimeError(f"block expects {blk.n_src_cap} source rows, got {x.shape[0]}")
                y = torch.empty((blk.n_dst_cap, K), dtype=torch.float32, device=dev)
                b = bias.contiguous().reshape(-1) if bias is not None else None
                a = add.contiguous() if add is not None else None
                eng.spmm_epi_into(blk.plan, blk.col, None, x, y, mean=True, add=a, bias=b, relu=relu)
                ctx.blk, ctx.has_add, ctx.rng_used = blk, add is not None, None
                ctx.cfg = (blk.n_dst_cap, K, int(relu), 0.0, None if bias is None else bias.shape)
                ctx.save_for_backward(y)
                return y

            @staticmethod
            def backward(ctx, g):
                (y,) = ctx.saved_tensors
                ga, gb = _epi_backward(ctx, g, y)
                gx = None
                if ctx.needs_input_grad[0]:
                    blk = ctx.blk
                    planT, dstT = blk.transposed()
                    gx, _ = eng._spmm_fwd("mean_bwd", planT, dstT, None, ga, blk.n_src_cap, aux=blk.rowptr)
                return gx, None, (ga if ctx.has_add else None), gb, None

        self.BlockMeanEpi = BlockMeanEpi
        self.SpMMEpi, self.SegmentEpi = SpMMEpi, SegmentEpi
        self.SpMMSumBiasAct = SpMMSumBiasAct
        self.BiasAct = BiasAct
        self.BiasAdd = BiasAdd
        self.SegmentSum, self.SegmentMean, self.SegmentMax = SegmentSum, SegmentMean, SegmentMax
        self.SpMMSum, self.SpMMMean, self.SpMMMax = SpMMSum, SpMMMean, SpMMMax
        self.BSpMMSum, self.GATFused = BSpMMSum, GATFused

    # ---- the seven reference entry points (src/operators.cpp:51-59) + the fused GAT op ---------
    def _seg_args(self, x, index, N):
        self._dev(x, index)
        if index.dim() != 1:
            raise IndexError(f"index dimension should be 1, but got {index.dim()}")
        if x.shape[0] != index.shape[0]:
            raise IndexError("fisrt dimension of x and index should be same")
        if index.dtype != torch.int64:
            raise RuntimeError(f"expected scalar type Long but found {index.dtype}")
        return x.contiguous(), index, int(N)

    def c_segment_sum(self, x, index, N):
        return self.SegmentSum.apply(*self._seg_args(x, index, N))

    def c_segment_mean(self, x, index, N):
        return self.SegmentMean.apply(*self._seg_args(x, index, N))

    def c_segment_max(self, x, index, N):
        return self.SegmentMax.apply(*self._seg_args(x, index, N))[0]

    def segment_max_with_arg(self, x, index, N):
        return self.SegmentMax.apply(*self._seg_args(x, index, N))

    def _spmm_args(self, index, weight, x):
        self._dev(index, weight, x)
        self._check_f32("x", x)
        if weight is not None:
            self._check_f32("weight", weight)
            weight = weight.contiguous()
        gp = self.graph_plan(index, x.shape[0])
        return gp, weight, x.contiguous()

    def c_spmm_sum(self, index, weight, x):
        return self.SpMMSum.apply(*self._spmm_args(index, weight, x))

    def c_spmm_mean(self, index, weight, x):
        return self.SpMMMean.apply(*self._spmm_args(index, weight, x))

    def c_spmm_max(self, index, weight, x):
        return self.SpMMMax.apply(*self._spmm_args(index, weight, x))

    def c_bspmm_sum(self, index, weight, x):
        if x.dim() != 3:
            raise RuntimeError("bspmm expects x of shape [num_nodes, heads, channels]")
        gp, weight, x = self._spmm_args(index, weight, x)
        if weight is None or weight.dim() != 2 or weight.shape[1] != x.shape[1]:
            raise RuntimeError("bspmm expects weight of shape [num_edges, heads]")
        C = int(x.shape[2])
        Cp = int(self.lib.ggl_policy_head_channels(C, gp.E, int(x.shape[0])))
        if Cp != C:
            # odd channel counts (41 classes per head ...): one zero-padded copy of x keeps the row walks and the
            # per-edge weight-gradient dots on 16-byte slices; the pad channels sum to zero and are dropped
            xp = torch.nn.functional.pad(x, (0, Cp - C))
            return self.BSpMMSum.apply(gp, weight, xp.contiguous())[:, :, :C]
        return self.BSpMMSum.apply(gp, weight, x)

    def gat_fused(self, index, el, er, x, negative_slope=0.2, num_nodes=None, dropout_rate=0.0, training=True):
        """out[i,h,:] = sum_{j->i} dropout(softmax_i(LeakyReLU(el[j,h] + er[i,h]))) * x[j,h,:]
        (dropout on the attention coefficients as gat_conv.py:104 / GATConvFuse(..., dropout_rate))."""
        self._dev(index, el, er, x)
        for n, t in (("el", el), ("er", er), ("x", x)):
            self._check_f32(n, t)
        n = x.shape[0] if num_nodes is None else num_nodes
        gp = index if isinstance(index, GraphPlan) else self.graph_plan(index, n, x.shape[0])
        p = float(dropout_rate) if training else 0.0
        if not 0.0 <= p < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        C = int(x.shape[2])
        Cp = int(self.lib.ggl_policy_head_channels(C, gp.E, int(x.shape[0])))
        if Cp != C:
            # e.g. 41 classes per head: one zero-padded copy of x ([N,H,44]) keeps every walk on 16-byte slices
            # (Reddit-sized, 8 x 41: forward 50 -> 20 ms); the pad channels aggregate to zero and are dropped
            xp = torch.nn.functional.pad(x, (0, Cp - C))
            out = self.GATFused.apply(gp, el.contiguous(), er.contiguous(), xp.contiguous(), negative_slope, p)
            return out[:, :, :C]
        return self.GATFused.apply(gp, el.contiguous(), er.contiguous(), x.contiguous(),
                                   negative_slope, p)

    def _check_weight(self, weight, gp):
        """An edge-weight vector handed to a kernel as a raw pointer: f32 (spmm_sum_cpu.cpp:22 would raise
        "expected scalar type Float"), one value per edge."""
        if weight is None:
            return None
        self._check_f32("weight", weight)
        if weight.numel() != gp.E:
            raise RuntimeError(f"edge weight must hold one value per edge: got {tuple(weight.shape)} for "
                               f"{gp.E} edges")
        return weight.reshape(-1).contiguous()

    # rectangular / explicit-plan variants used by the harness and the multi-GPU layer
    def spmm(self, gp, weight, x, reduce="sum"):
        self._dev(x, weight)
        self._check_f32("x", x)
        weight = self._check_weight(weight, gp)
        fn = {"sum": self.SpMMSum, "mean": self.SpMMMean, "max": self.SpMMMax}[reduce]
        return fn.apply(gp, weight, x.contiguous())

    def colsum(self, g):
        """out[k] = sum_r g[r, k] for a row-major f32 [N, K] matrix (deterministic two-stage kernel)."""
        dev = self._dev(g)
        self._check_f32("g", g)
        g = g.contiguous()
        N = int(g.shape[0])
        K = g.numel() // N if N > 0 else int(math.prod(g.shape[1:]))
        out = torch.empty(tuple(g.shape[1:]), dtype=torch.float32, device=dev)
        wsb = self.lib.ggl_colsum_workspace_bytes(N, K)
        ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
        self._check(self.lib.ggl_colsum_f32(_ptr(g), N, K, _ptr(out), _ptr(ws), wsb, self._stream(dev)))
        return out

    def bias_add(self, x, bias):
        """x + bias with the bias gradient computed by ggl_colsum_f32 (gcn_conv.py:105-106)."""
        return self.BiasAdd.apply(x, bias)

    def _rng_state(self, dev):
        """Device-resident Philox state {seed, offset} for the fused dropout, seeded from torch's RNG."""
        st = self._rng.get(str(dev))
        if st is None:
            seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())  # follows torch.manual_seed
            st = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
            self._rng[str(dev)] = st
        return st

    def reseed(self, seed=None):
        """Forget the fused-dropout RNG state (the next use draws a new seed from torch's generator) — of this engine AND, for
        a product engine, of the C++ operator library (torch.ops.ggl keeps its own counter stream; since round 6 a one-rank
        training step takes that route, dist._default_route, so a reseed that skipped it was no reseed)."""
        self._rng.clear()
        if getattr(self, "is_product", False):
            from . import cpp_ops

            if cpp_ops._loaded:
                torch.ops.ggl.reseed()
        if seed is not None:
            torch.manual_seed(seed)

    def bias_act(self, a, bias=None, relu=False, p_drop=0.0, training=True):
        """dropout(relu(a + bias)) — the step after every aggregate (gcn_conv.py:105-106, models/gcn.py:55-59)."""
        self._dev(a, bias)
        self._check_f32("a", a)
        p = float(p_drop) if training else 0.0
        return self.BiasAct.apply(a, bias, bool(relu), p)

    def spmm_bias_act(self, gp, weight, x, bias=None, relu=False, p_drop=0.0, training=True):
        """dropout(relu(A x + bias)) for a GraphPlan `gp` (what GCNConv + the model's ReLU/dropout compute,
        gcn_conv.py:78-108, models/gcn.py:55-59).  One kernel when the feature width is a multiple of 4
        (16-byte rows); otherwise the SpMM and the epilogue kernel run back to back — same values."""
        self._dev(x, weight, bias)
        self._check_f32("x", x)
        weight = self._check_weight(weight, gp)
        if bias is not None:
            self._check_f32("bias", bias)
        p = float(p_drop) if training else 0.0
        if x.dim() == 2 and x.shape[1] % 4 == 0:
            return self.SpMMSumBiasAct.apply(gp, weight, x.contiguous(), bias, bool(relu), p)
        return self.BiasAct.apply(self.SpMMSum.apply(gp, weight, x.contiguous()), bias, bool(relu), p)

    def spmm_epi(self, gp, weight, x, reduce="sum", add=None, bias=None, relu=False, p_drop=0.0, training=True):
        """dropout(relu(reduce_{j->i} w x_j + add_i + bias)), reduce in {'sum', 'mean'}: one kernel for 16-byte
        rows (feature width a multiple of 4), the reduce op followed by the adds and the epilogue pass otherwise."""
        self._dev(x, weight, bias, add)
        self._check_f32("x", x)
        weight = self._check_weight(weight, gp)
        for n, t in (("bias", bias), ("add", add)):
            if t is not None:
                self._check_f32(n, t)
        if add is not None and tuple(add.shape) != (gp.N_dst, x.shape[1]):
            raise RuntimeError("add must be [destination rows, feature width]")
        p = float(p_drop) if training else 0.0
        if x.dim() == 2 and x.shape[1] % 4 == 0:
            return self.SpMMEpi.apply(gp, weight, x.contiguous(), reduce == "mean", add, bias, bool(relu), p)
        out = self.spmm(gp, weight, x, reduce)
        if add is not None:
            out = out + add
        return self.BiasAct.apply(out, bias, bool(relu), p)

    def segment_epi(self, msg, ids, N, reduce="mean", add=None, bias=None, relu=False):
        """relu(segment_{sum,mean}(msg, ids, N) + add + bias) for f32 messages [E, K] — one kernel."""
        self._dev(msg, ids, bias, add)
        self._check_f32("msg", msg)
        msg, ids, N = self._seg_args(msg, ids, N)
        if msg.dim() != 2:
            raise RuntimeError("segment_epi expects [E, K] messages")
        for n, t in (("bias", bias), ("add", add)):
            if t is not None:
                self._check_f32(n, t)
        if add is not None and tuple(add.shape) != (N, msg.shape[1]):
            raise RuntimeError("add must be [num_segments, feature width]")
        return self.SegmentEpi.apply(msg, ids, N, reduce == "mean", add, bias, bool(relu))

    def gat_headmean_supported(self, heads, in_channels, out_channels):
        return bool(self.gat_fast and self.lib.ggl_gat_sh_supported(int(heads), int(in_channels), int(out_channels)))

    def gat_headmean(self, index, x, W, att, negative_slope=0.2, num_nodes=None, dropout_rate=0.0, training=True):
        """mean over the 8 heads of a GAT layer's output (before the bias): [N, C] from x [N, F], W [F, 8 C],
        att [1, 8, 2 C] — the layer aggregated before it is transformed (see GATHeadMean)."""
        self._dev(index, x, W, att)
        for n, t in (("x", x), ("W", W), ("att", att)):
            self._check_f32(n, t)
        n = x.shape[0] if num_nodes is None else int(num_nodes)
        if n != x.shape[0]:
            raise RuntimeError("gat_headmean runs on square graphs (every destination is also a source row)")
        gp = index if isinstance(index, GraphPlan) else self.graph_plan(index, n, n)
        p = float(dropout_rate) if training else 0.0
        return self.GATHeadMean.apply(gp, x.contiguous(), W.contiguous(), att.contiguous(), negative_slope, p)

    def block_mean_epi(self, x, blk, add=None, bias=None, relu=False):
        """SAGEConv(mean) over a sampler Block: relu(mean of the sampled neighbours + add + bias), one kernel."""
        self._dev(x, add, bias)
        self._check_f32("x", x)
        return self.BlockMeanEpi.apply(x.contiguous(), blk, add, bias, bool(relu))

    def set_option(self, name, value):
        self._check(self.lib.ggl_set_option(name.encode(), int(value)))

    def time_spmm_sum(self, gp, weight, x, reps=10):
        """Average ms per launch of the dominant SpMM-sum kernel (hipEvents on the current stream)."""
        dev = x.device
        K = int(math.prod(x.shape[1:]))
        out = torch.empty((gp.N_dst,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
        part = self._partial(gp.fwd, torch.float32, K, False, dev)
        cs = gp.fwd.c_struct(part)
        if getattr(gp.fwd, "order_fn", None) is not None:
            # a plan's row hand-out order is computed on its SECOND launch (SegPlan.c_struct); the kernel is timed the way a training
            # step runs it — on a plan that comes back.  (Round 6: with the step on torch.ops.ggl this engine's plan can arrive
            # here unused; without the order the products aggregate measured 23.7 instead of 13.5 ms.)
            cs = gp.fwd.c_struct(part)
        ms = ctypes.c_float(0.0)
        w_by_pos = 0
        if weight is not None and gp.fwd.perm is not None:
            self._sorted_weights(gp.fwd, weight)  # a reused weight vector: timed the way the step runs it
            weight, w_by_pos = self._sorted_weights(gp.fwd, weight)
        self._check(self.lib.ggl_time_spmm_sum(ctypes.byref(cs), _ptr(gp.col), _ptr(weight), w_by_pos, _ptr(x),
                                               K, _ptr(out), self._stream(dev), int(reps),
                                               ctypes.byref(ms)))
        return float(ms.value)
