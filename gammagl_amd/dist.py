"""Multi-GPU full-graph aggregation: 1-D node partition + 1-hop halo exchange.

The reference has no distributed code at all; this is a new design for one
node of 8 MI355X, one process per GPU over ``torch.distributed`` (backend "nccl" = RCCL over xGMI):

* rank p owns a contiguous range of destination rows (ranges balanced by in-edge count), the
  activations of those rows and the CSR rows (in-edges) that produce them;
* per layer, forward:  (1) gather the local rows other ranks need (precomputed send lists),
  (2) ONE all-to-all-v of K*4-byte rows — on the fully connected xGMI mesh every peer pair has its
  own link, so an all-to-all is link-parallel rather than ring-bound — issued asynchronously (RCCL
  runs it on its own stream), (3) meanwhile the SpMM over the edges whose source is local,
  (4) after the wait, the SpMM over the edges whose source arrived in the halo buffer, added in.
  backward: transposed halo SpMM -> reverse all-to-all-v, overlapped with the transposed local SpMM,
  then a deterministic segment-sum of the returned rows into the local gradient (no atomics);
  the exchange is cut into feature-column chunks so that chunk c+1 travels while chunk c is multiplied;
* weight gradients: one flat all-reduce per step (3 small matrices);
* rows that never change — the INPUT features — are exchanged once (`PartitionedGraph.with_halo`): the first
  layer then aggregates over `[x_local ; x_halo]` (or transforms those rows itself, A (X W) = A_loc (X_loc W) +
  A_halo (X_halo W)) and needs no exchange in either direction, forward or backward: the gradient of a halo row of
  X W is consumed where it is produced, by the rank's own share of dW = X^T dH, which is all-reduced anyway;
* send / receive buffers are persistent per (direction, chunk, shape) (`PartitionedGraph._buf`), and
  `PartitionedGraph.profile` (a dict, off when None) collects per-step exchange volumes and the time the compute
  stream spent waiting for each all-to-all (events around `work.wait()`), which bench.py reports for N > 1.

Every output row is still reduced on exactly one GPU by the same kernels as the single-GPU path;
only the association (local edges first, then halo edges) differs, so results agree with the
single-GPU run to float rounding (<= 1e-5 relative), not bit for bit.
"""
import os
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import engine as _default_engine
from .dense import wgrad
from .ops import _ptr

HALO_CHUNKS = int(os.environ.get("GGL_HALO_CHUNKS", "0"))  # 0 = automatic (4 at K >= 256, 2 at K >= 128)
DIST_EXACT = os.environ.get("GGL_DIST_EXACT", "1") == "1"   # 0: partitioned aggregates walk their hub rows chunk by chunk (A/B)
A2A_MODE = os.environ.get("GGL_HALO_A2A", "a2a")            # "a2a": all_to_all_single | "p2p": grouped isend / irecv per peer


def balanced_bounds(dst, num_nodes, world):
    """Contiguous node ranges with ~equal numbers of in-edges (identical on every rank)."""
    deg = torch.bincount(dst, minlength=num_nodes)
    cum = torch.cumsum(deg, 0)
    total = int(cum[-1]) if num_nodes > 0 else 0
    targets = torch.arange(1, world, device=dst.device, dtype=torch.float64) * (total / world)
    cuts = torch.searchsorted(cum.double(), targets).clamp(max=num_nodes)
    return [0] + [int(c) for c in cuts.tolist()] + [int(num_nodes)]


class _Done:
    """Stand-in for a collective's work handle where nothing travels (dry runs)."""

    def wait(self):
        return True


class _Works:
    """The work handles of one grouped point-to-point exchange behind the single wait() an all-to-all's handle has."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def _cpp_ops():
    from . import cpp_ops

    return cpp_ops.load()


def _default_route(eng):
    want = os.environ.get("GGL_ROUTE", "auto")
    if want not in ("auto", "cpp", "ctypes"):
        raise ValueError(f"GGL_ROUTE={want!r}: expected auto, cpp or ctypes")
    if want == "ctypes":
        return "ctypes"
    from . import cpp_ops

    # an engine injected on another build of the kernel library (tests: -O1 / ASan host builds) is only reachable through ctypes
    product = getattr(eng, "is_product", False)
    if want == "cpp":
        if not cpp_ops.available():
            raise ImportError(f"GGL_ROUTE=cpp but {cpp_ops.LIB_PATH} is not built")
        return "cpp"
    return "cpp" if (product and cpp_ops.enabled()) else "ctypes"


class PartitionedGraph:
    """This rank's share of a weighted graph, plus everything the halo exchange needs.

    Two ways in: from the global edge list (every rank filters it: small graphs and tests), or —
    `from_local` — from the edges this rank already owns (`synth.rmat_partitioned`: no rank ever sees the
    whole list; the only route to a papers100M-sized graph)."""

    def __init__(self, edge_index, edge_weight, num_nodes, rank=0, world=1, group=None, eng=None,
                 bounds=None, self_halo_from=None):
        """`self_halo_from` (tests only, world == 1 with an initialised process group): treat the
        local rows >= that global id as if they lived on a remote rank, so the complete exchange path
        (send lists, all-to-all-v with itself, halo SpMM, reverse exchange) runs through RCCL on a
        single GPU."""
        src, dst = edge_index[0], edge_index[1]
        bounds = bounds or balanced_bounds(dst, num_nodes, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        mine = (dst >= lo) & (dst < hi)
        self._setup(src[mine], dst[mine] - lo, edge_weight[mine], bounds, num_nodes, int(edge_index.shape[1]),
                    rank, world, group, eng, self_halo_from=self_halo_from)

    @classmethod
    def from_local(cls, src, dst_local, w, bounds, num_nodes, e_global, rank=0, world=1, group=None, eng=None,
                   send_rows=None):
        """`src` global source ids / `dst_local` local destination ids / `w` weights of the edges whose
        destination this rank owns.  `send_rows` (a list with one tensor of local row ids per peer) makes it a
        DRY partition: the process plays rank `rank` of a len(send_rows)-way partition on its own — send lists
        as given, buffers and kernels exactly as in a real run, nothing on the wire (one GPU measuring one
        rank's share of a graph that needs eight)."""
        pg = cls.__new__(cls)
        pg._setup(src, dst_local, w, bounds, num_nodes, e_global, rank, world, group, eng, send_rows=send_rows)
        return pg

    def _setup(self, s, d, w, bounds, num_nodes, e_global, rank, world, group, eng, self_halo_from=None,
               send_rows=None):
        self.eng = eng if eng is not None else _default_engine()
        self.rank, self.world, self.group = rank, world, group
        # which host implementation serves a rank WITHOUT a halo: "cpp" = torch.ops.ggl (the route compat/_torch_ext.py binds),
        # "ctypes" = the Python engine.  Partitioned runs use the engine's in-place / accumulating forms, which the
        # operator surface does not have.  GGL_ROUTE overrides; "cpp" needs libggl_torch.so (built by `make -C csrc torch`).
        self.route = _default_route(self.eng)
        self.dry = send_rows is not None
        self.comm = world > 1 or self_halo_from is not None or self.dry
        _apply_dist_exact(self)
        dev = s.device
        self.bounds = list(bounds)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_local, self.n_global = lo, hi, hi - lo, int(num_nodes)
        self.e_global = int(e_global)
        self.e_local = int(s.shape[0])
        is_loc = (s >= lo) & (s < (hi if self_halo_from is None else int(self_halo_from)))
        # edges whose source row is local
        self.ei_loc = torch.stack([s[is_loc] - lo, d[is_loc]]).contiguous()
        self.w_loc = w[is_loc].contiguous()
        self.gp_loc = self.eng.graph_plan(self.ei_loc, self.n_local, self.n_local)
        # edges whose source row lives elsewhere: halo buffer ordered by global id (= by owner)
        rem = ~is_loc
        rs, rd = s[rem], d[rem]
        self.halo_ids = torch.unique(rs)  # sorted
        self.n_halo = int(self.halo_ids.shape[0])
        bt = torch.tensor(self.bounds, device=dev, dtype=torch.int64)
        owner_start = torch.searchsorted(self.halo_ids, bt)  # halo rows owned by q: [start[q], start[q+1])
        self.recv_splits = (owner_start[1:] - owner_start[:-1]).tolist()
        if self.n_halo > 0:
            self.ei_halo = torch.stack([torch.searchsorted(self.halo_ids, rs), rd]).contiguous()
            self.w_halo = w[rem].contiguous()
            self.gp_halo = self.eng.graph_plan(self.ei_halo, self.n_local, self.n_halo)
        else:
            self.ei_halo = self.w_halo = self.gp_halo = None
        del rs, rd, rem, is_loc
        # tell every owner which of its rows we need (only halo ids travel, never an edge list)
        if self.dry:
            self.send_splits = [int(t.numel()) for t in send_rows]
            self.send_idx = torch.cat([t.to(torch.int64) for t in send_rows]).contiguous()
        elif self.comm:
            rc = torch.tensor(self.recv_splits, device=dev, dtype=torch.int64)
            sc = torch.empty_like(rc)
            dist.all_to_all_single(sc, rc, group=group)
            self.send_splits = sc.tolist()
            req = torch.empty(int(sc.sum()), device=dev, dtype=torch.int64)
            dist.all_to_all_single(req, self.halo_ids.contiguous(), self.send_splits, self.recv_splits,
                                   group=group)
            self.send_idx = (req - lo).contiguous()
        else:
            self.send_splits = [0]
            self.send_idx = torch.empty(0, device=dev, dtype=torch.int64)
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.n_local)
        self.n_send = int(self.send_idx.shape[0])
        self.send_plan = self.eng.seg_plan(self.send_idx, self.n_local) if self.n_send > 0 else None
        self._bufs = {}        # persistent exchange buffers, see _buf
        self.halo_chunks = {}  # column chunks of the exchange measured on this machine: {"exchange": n, "const": n}
        self._const = {}       # [x_local ; x_halo] of tensors that never change, see with_halo
        self.profile = None    # set to {} to collect exchange statistics (bench.py, N > 1)

    # ---- persistent buffers / constants / instrumentation ----------------------------------------
    def _buf(self, tag, rows, cols, dtype, device):
        """One flat buffer per `tag`, grown to the largest request and handed out as a [rows, cols] view: allocated on
        first use and reused by every later step, so the exchange runs on fixed addresses (no allocator traffic per
        chunk per call; a precondition for recording the step into a hipGraph).  Tags are shared where lifetimes
        cannot overlap — ("halo", chunk) serves the forward receive buffer and the backward's outgoing halo
        gradients, ("send", chunk) the forward send rows and the returned gradients, layers of every width — which
        keeps the pool at 2 x (halo + send rows) x the widest layer (a papers100M-sized share: 2 x 24 GB).
        Reuse is safe in stream order: a buffer's consumer (the halo SpMM / the segment-sum of returned rows) is
        enqueued before the next collective that overwrites it is issued, and the process group makes its stream
        wait for the issuing stream."""
        need = int(rows) * int(cols)
        b = self._bufs.get(tag)
        if b is None or b.device != device or b.dtype != dtype or b.numel() < need:
            b = torch.empty(max(need, 1), dtype=dtype, device=device)
            self._bufs[tag] = b
        return b[:need].view(int(rows), int(cols))

    def with_halo(self, x):
        """[x_local ; x_halo] for a tensor of local rows that NEVER changes (the input features): the halo rows are
        fetched from their owners once and kept, keyed on the identity + version of `x`.  An aggregate over the
        result (`aggregate(..., halo_included=True)`) exchanges nothing, in either direction."""
        if not self.comm:
            return x   # (a rank WITHOUT halo rows still takes part below: its peers may need its rows, and a collective
            #            that some ranks skip is a deadlock)
        key = (x.untyped_storage()._cdata, x.storage_offset(), tuple(x.shape), x._version, str(x.device))
        hit = self._const.get(key)
        if hit is None:
            with torch.no_grad():
                xc = x.detach().contiguous()
                send = xc.index_select(0, self.send_idx) if self.n_send > 0 else xc.new_empty((0,) + tuple(xc.shape[1:]))
                recv = torch.empty((self.n_halo,) + tuple(xc.shape[1:]), dtype=xc.dtype, device=xc.device)
                if self.dry:
                    recv.zero_()
                else:
                    dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.group)
                hit = torch.cat([xc, recv], 0)
            self._const.clear()      # one constant at a time: a new input tensor replaces the old copy
            self._const[key] = hit
            self._const_ref = x      # keeps the storage alive so its identity cannot be recycled
        return hit

    def _note(self, **kw):
        if self.profile is not None:
            for k, v in kw.items():
                self.profile[k] = self.profile.get(k, 0) + v

    def _timed_wait(self, work, dev):
        """work.wait() with the stall it causes recorded when profiling: on the GPU the compute stream's wait is
        bracketed by two events (read back by `profile_summary`), on the host (gloo) by the wall clock."""
        if self.profile is None:
            work.wait()
            return
        if dev.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            work.wait()
            e1.record()
            self.profile.setdefault("_events", []).append((e0, e1))
        else:
            t0 = time.perf_counter()
            work.wait()
            self.profile["exposed_ms"] = self.profile.get("exposed_ms", 0.0) + (time.perf_counter() - t0) * 1e3

    def profile_summary(self, steps):
        """Per-step averages of what `profile` collected over `steps` steps (call after a device sync)."""
        p = dict(self.profile or {})
        ev = p.pop("_events", [])
        ms = p.pop("exposed_ms", 0.0) + sum(a.elapsed_time(b) for a, b in ev)
        s = max(int(steps), 1)
        return {"halo_exposed_ms": ms / s, "a2a_calls": p.get("a2a_calls", 0) / s,
                "a2a_GB_in": p.get("bytes_in", 0) / s / 1e9, "a2a_GB_out": p.get("bytes_out", 0) / s / 1e9}

    # ---- raw building blocks -------------------------------------------------------------------
    def _a2a(self, out_rows, inp, out_splits, in_splits, tag="a2a"):
        out = self._buf(tag, out_rows, inp.shape[1], inp.dtype, inp.device)
        self._note(a2a_calls=1, bytes_in=out.numel() * out.element_size(), bytes_out=inp.numel() * inp.element_size())
        if self.dry:  # nothing travels: the receive buffer keeps whatever it holds (finite values for timing)
            out.zero_()
            return out, _Done()
        if A2A_MODE == "p2p":
            # the same exchange as explicit grouped point-to-point transfers (ncclGroupStart; ncclSend / ncclRecv per
            # peer; ncclGroupEnd — what SURVEY.md §8e sketches): an A/B switch for the first real multi-GPU run, where
            # all_to_all_single's uneven-split path over xGMI is untested (GGL_HALO_A2A=p2p)
            ops, o0, i0 = [], 0, 0
            me = dist.get_rank(self.group)
            for peer, (no, ni) in enumerate(zip(out_splits, in_splits)):
                if peer == me:
                    if no:
                        out[o0:o0 + no].copy_(inp[i0:i0 + ni])
                elif no or ni:
                    gpeer = dist.get_global_rank(self.group, peer) if self.group is not None else peer
                    if ni:
                        ops.append(dist.P2POp(dist.isend, inp[i0:i0 + ni], gpeer, group=self.group))
                    if no:
                        ops.append(dist.P2POp(dist.irecv, out[o0:o0 + no], gpeer, group=self.group))
                o0, i0 = o0 + no, i0 + ni
            return out, (_Works(dist.batch_isend_irecv(ops)) if ops else _Done())
        work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)
        return out, work

    def release(self):
        """Undo what constructing this graph changed process-wide (the GGL_DIST_EXACT=0 switch of `exact_long_rows`)."""
        _restore_dist_exact(self)

    def aggregate(self, h, bias=None, relu=False, p_drop=0.0, training=True, halo_included=False):
        """out[i] = dropout(relu(sum_{j->i} w_ij h[j] + bias)) for the local rows i (autograd-aware).  The
        epilogue (gcn_conv.py:105-106, models/gcn.py:55-59) rides on the store of the LAST edge block added:
        the local SpMM when this rank has no halo, the halo-source SpMM otherwise.  `halo_included`: `h` holds
        n_local + n_halo rows, the halo rows already in place behind the local ones (`with_halo`, or rows this rank
        computed from them): nothing is exchanged, and the gradient comes back with the same n_local + n_halo rows."""
        p = float(p_drop) if training else 0.0
        if self.route == "cpp" and not self.comm and not halo_included and h.shape[1] % 4 == 0:
            # ONE GPU, no halo: the aggregate through the operator the zero-edit drop-in binds — torch.ops.ggl.spmm_epi, i.e.
            # dispatcher -> libggl_torch.so (plan cache + autograd in C++) -> C ABI -> libggl_mpops_hip.so (round-5 verdict: the
            # benchmarked route was the ctypes engine, which a GammaGL user of compat/_torch_ext.py never reaches)
            return _cpp_ops().spmm_epi(self.ei_loc, self.w_loc, h, False, None, bias, bool(relu), p)
        return _HaloAggregate.apply(h, self, bias, bool(relu), p, bool(halo_included))


def _restore_dist_exact(pg):
    before = getattr(pg.eng, "_exact_before_dist", None)
    if before is not None:
        pg.eng.lib.ggl_set_option(b"exact_long_rows", int(before))
        del pg.eng._exact_before_dist


def _apply_dist_exact(pg):
    """A/B switch (GGL_DIST_EXACT=0): partitioned aggregates walk their f32 hub rows CHUNKED.  A row of a partitioned graph is
    the sum of two launches — its local-source edges, then its halo-source edges added on top — so the reference's serial
    order is out of reach there whatever a single launch does; the serial hub walk (hubf32.hip) is kept anyway because since
    it starts its longest rows first it is the FASTER walk: products-sized dry shares 17.05 -> 15.94 ms per step at P = 8,
    29.5 -> 27.1 at P = 4 (profiles/r4_dry_share_knobs.txt).  The switch sets the library option `exact_long_rows` ONCE, when
    a partitioned graph with a communicator is constructed — a rank is a process, and the option is process-wide (round 4
    flipped it around every forward / backward, which other threads' launches could observe half-way)."""
    if pg.comm and not DIST_EXACT:
        # process-wide: remembered so that release() can put it back (round-5 advisor: every later non-partitioned aggregate
        # of the process, and bench.py's parity leg, silently lost the reference-order hub walk)
        if not hasattr(pg.eng, "_exact_before_dist"):
            pg.eng._exact_before_dist = int(pg.eng.lib.ggl_get_option(b"exact_long_rows"))
        pg.eng.lib.ggl_set_option(b"exact_long_rows", 0)
        import warnings

        warnings.warn("GGL_DIST_EXACT=0: library option exact_long_rows set to 0 for this process until "
                      "PartitionedGraph.release() (partitioned hub rows walk chunked)", stacklevel=3)


class _HaloAggregate(torch.autograd.Function):
    @staticmethod
    def _pad4(t):
        """Feature width to a multiple of 4 so the float4 kernels apply (K = 47 classes -> 48: measured
        9.3 -> 5.8 ms forward, 7.3 -> 3.9 ms backward on the products-sized graph, profiles/)."""
        k = t.shape[1]
        return t if k % 4 == 0 else torch.nn.functional.pad(t, (0, (-k) % 4))

    @staticmethod
    def _chunks(K, pg=None, kind="exchange"):
        """Feature-column chunks of the exchange: the all-to-all-v of chunk c+1 runs while the halo SpMM of
        chunk c computes (RCCL executes the queued collectives in order on its own stream; the compute
        stream only waits for the chunk it is about to use).  The chunks cost compute (four 64-wide halo walks run
        at 60 % of one 256-wide walk: profiles/r4_dry_share_knobs.txt), so how many pay depends on the links: the
        count is GGL_HALO_CHUNKS if set, else what `DistGCNTrainer.tune_halo_chunks` measured on this machine
        (`pg.halo_chunks[kind]`, layers of 128 columns and more; `kind` "const" = the first layer, whose halo buffer
        is filled by a GEMM and never travels), else 4 at K >= 256, 2 at K >= 128."""
        tuned = pg.halo_chunks.get(kind) if (pg is not None and K >= 128) else None
        n = HALO_CHUNKS if HALO_CHUNKS > 0 else tuned if tuned else (4 if K >= 256 else 2 if K >= 128 else 1)
        while n > 1 and K % (4 * n) != 0:
            n -= 1
        w = K // n
        return [(i * w, (i + 1) * w) for i in range(n)]

    @staticmethod
    def forward(ctx, h, pg, bias, relu, p_drop, pre=False):
        eng = pg.eng
        pre = bool(pre and pg.comm)      # halo rows already in place: NO collective, whether or not this rank has a halo
        if h.shape[0] != pg.n_local + (pg.n_halo if pre else 0):
            raise RuntimeError(f"aggregate: expected {pg.n_local + (pg.n_halo if pre else 0)} rows, got {h.shape[0]}")
        ctx.k_orig = h.shape[1]
        epi = bias is not None or relu or p_drop > 0
        fused = epi and h.shape[1] % 4 == 0   # (odd widths: aggregate on padded rows, epilogue as its own pass)
        h = _HaloAggregate._pad4(h.contiguous())
        K = h.shape[1]
        dev = h.device
        rng = eng._rng_state(dev) if (fused and p_drop > 0) else None
        ctx.rng_used = rng.clone() if rng is not None else None
        b = bias.contiguous().reshape(-1) if (fused and bias is not None) else None
        works = []
        if pre:      # the halo rows sit behind the local ones already: one "chunk", nothing on the wire
            if pg.n_halo > 0:
                works.append((0, K, h[pg.n_local:], _Done()))
        elif pg.comm:
            for ci, (c0, c1) in enumerate(_HaloAggregate._chunks(K, pg)):
                send = pg._buf(("send", ci), pg.n_send, c1 - c0, h.dtype, dev)
                if pg.n_send > 0:  # one kernel: rows of the column block straight into the send buffer
                    eng.gather_rows_into(h[:, c0:c1], pg.send_idx, send)
                recv, work = pg._a2a(pg.n_halo, send, pg.recv_splits, pg.send_splits, tag=("halo", ci))
                works.append((c0, c1, recv, work))
        out = torch.empty((pg.n_local, K), dtype=torch.float32, device=dev)
        last_is_local = fused and (not pg.comm or pg.n_halo == 0)
        # local-source edges: overlaps the exchange
        if last_is_local:
            eng.spmm_epi_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, out, bias=b, relu=relu, p_drop=p_drop,
                              rng=rng, epi_K=K)
        else:
            eng.spmm_sum_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, out)
        for i, (c0, c1, recv, work) in enumerate(works):
            pg._timed_wait(work, dev)
            if pg.n_halo > 0:  # halo-source edges added onto the column block in place
                if fused:
                    eng.spmm_epi_into(pg.gp_halo.fwd, pg.gp_halo.col, pg.w_halo, recv, out[:, c0:c1], accumulate=True,
                                      bias=b, relu=relu, p_drop=p_drop, rng=rng, epi_K=K, col0=c0,
                                      advance_rng=(i == len(works) - 1))
                else:
                    eng.spmm_sum_into(pg.gp_halo.fwd, pg.gp_halo.col, pg.w_halo, recv, out[:, c0:c1], accumulate=True)
        if out.shape[1] != ctx.k_orig:
            out = out[:, :ctx.k_orig].contiguous()
        if epi and not fused:
            rng = eng._rng_state(dev) if p_drop > 0 else None
            ctx.rng_used = rng.clone() if rng is not None else None
            y = torch.empty_like(out)
            bb = bias.contiguous().reshape(-1) if bias is not None else None
            eng._check(eng.lib.ggl_bias_act_fwd(_ptr(out), _ptr(bb), out.shape[0], out.shape[1], int(relu),
                                                float(p_drop), _ptr(rng), _ptr(y), eng._stream(out.device)))
            out = y
        ctx.pg, ctx.pre, ctx.epi = pg, pre, (epi, relu, p_drop, None if bias is None else bias.shape)
        if epi:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        pg = ctx.pg
        eng = pg.eng
        g = g.contiguous()
        dev = g.device
        epi, relu, p_drop, bshape = ctx.epi
        gb = None
        if epi:  # through dropout / ReLU / + bias in one pass (mask redrawn from the saved rng state)
            (y,) = ctx.saved_tensors
            N, K0 = int(g.shape[0]), int(g.shape[1])
            ga = torch.empty_like(g)
            gb = torch.empty(K0, dtype=torch.float32, device=dev) if bshape is not None else None
            wsb = eng.lib.ggl_bias_act_bwd_workspace_bytes(N, K0)
            ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
            eng._check(eng.lib.ggl_bias_act_bwd(_ptr(g), _ptr(y), N, K0, int(relu), float(p_drop),
                                                _ptr(ctx.rng_used), _ptr(ga), _ptr(gb), _ptr(ws), wsb,
                                                eng._stream(dev)))
            g = ga
            if gb is not None:
                gb = gb.reshape(bshape)
        if not ctx.needs_input_grad[0]:
            return None, None, gb, None, None, None
        g = _HaloAggregate._pad4(g)
        K = g.shape[1]
        if ctx.pre:
            # the gradient of a halo row stays here (its consumer is this rank's share of a weight gradient): both
            # transposed walks write into one [n_local + n_halo, K] result, nothing travels back
            gh = torch.empty((pg.n_local + pg.n_halo, K), dtype=g.dtype, device=dev)
            if pg.n_halo > 0:
                eng.spmm_sum_into(pg.gp_halo.bwd, pg.gp_halo.colT, pg.w_halo, g, gh[pg.n_local:])
            eng.spmm_sum_into(pg.gp_loc.bwd, pg.gp_loc.colT, pg.w_loc, g, gh[:pg.n_local])
            return (gh if K == ctx.k_orig else gh[:, :ctx.k_orig].contiguous()), None, gb, None, None, None
        works = []
        if pg.comm:
            for ci, (c0, c1) in enumerate(_HaloAggregate._chunks(K, pg)):
                ghalo = pg._buf(("halo", ci), pg.n_halo, c1 - c0, g.dtype, dev)
                if pg.n_halo > 0:  # reads the column block of g in place (row stride passed down)
                    eng.spmm_sum_into(pg.gp_halo.bwd, pg.gp_halo.colT, pg.w_halo, g[:, c0:c1], ghalo)
                gsend, work = pg._a2a(pg.n_send, ghalo, pg.send_splits, pg.recv_splits, tag=("send", ci))  # chunk c travels while c+1 computes
                works.append((c0, c1, gsend, work))
        gh, _ = eng._spmm_fwd("sum", pg.gp_loc.bwd, pg.gp_loc.colT, pg.w_loc, g, pg.n_local)  # overlaps
        for (c0, c1, gsend, work) in works:
            pg._timed_wait(work, dev)
            if pg.n_send > 0:  # deterministic scatter-add of the returned rows, onto the column block in place
                eng.segment_sum_into(gsend, pg.send_plan, gh[:, c0:c1], accumulate=True)
        return (gh if gh.shape[1] == ctx.k_orig else gh[:, :ctx.k_orig].contiguous()), None, gb, None, None, None


class _ConstInputLayer(torch.autograd.Function):
    """The FIRST GCN layer in GammaGL's association, A (X W) (gcn_conv.py:79), on input features that never change
    and whose halo rows this rank holds (`PartitionedGraph.with_halo`): A (X W) = A_loc (X_loc W) + A_halo (X_halo W).
    The halo buffer the other layers receive over the wire is here FILLED BY A GEMM, one column chunk at a time, into
    the very same persistent buffers — no exchange in either direction, no [n_halo, K] tensor of its own:

      forward   h_loc = X_loc W^T; out = A_loc h_loc; per chunk c: buf_c = X_halo W_c^T, out[:, c] += A_halo buf_c
                (bias / ReLU / dropout ride on the store of the last block added, as everywhere);
      backward  ga = epilogue'(g); dW = (A_loc^T ga)^T X_loc; per chunk c: buf_c = A_halo^T ga[:, c],
                dW[c] += buf_c^T X_halo.  The gradient of a halo row of X W is consumed where it is produced — this
                rank's share of dW, which the step all-reduces anyway.  X carries no gradient.

    Costs (n_halo x f_in x K) extra multiply-adds each way instead of 2 x n_halo x K floats on the links
    (products-sized graph, 8 ranks: 0.6 ms of GEMM instead of 1.7 GB over xGMI per step)."""

    @staticmethod
    def forward(ctx, x_cat, w, pg, bias, relu, p_drop, pad_out):
        eng, nl = pg.eng, pg.n_local
        dev = x_cat.device
        wp = w if not pad_out else F.pad(w, (0, 0, 0, int(pad_out)))       # [K, f_in]
        K = int(wp.shape[0])
        x_loc, x_halo = x_cat[:nl], x_cat[nl:]
        h = x_loc @ wp.t()
        out = torch.empty((nl, K), dtype=torch.float32, device=dev)
        epi = bias is not None or relu or p_drop > 0
        rng = eng._rng_state(dev) if p_drop > 0 else None
        ctx.rng_used = rng.clone() if rng is not None else None
        b = bias.contiguous().reshape(-1) if bias is not None else None
        if pg.n_halo == 0:   # a rank without halo rows: the epilogue rides on the local walk
            eng.spmm_epi_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, out, bias=b, relu=relu, p_drop=p_drop, rng=rng,
                              epi_K=K)
        else:
            eng.spmm_sum_into(pg.gp_loc.fwd, pg.gp_loc.col, pg.w_loc, h, out)
        chunks = _HaloAggregate._chunks(K, pg, "const") if pg.n_halo > 0 else []
        for i, (c0, c1) in enumerate(chunks):
            hh = pg._buf(("halo", i), pg.n_halo, c1 - c0, torch.float32, dev)
            torch.mm(x_halo, wp[c0:c1].t(), out=hh)
            eng.spmm_epi_into(pg.gp_halo.fwd, pg.gp_halo.col, pg.w_halo, hh, out[:, c0:c1], accumulate=True, bias=b,
                              relu=relu, p_drop=p_drop, rng=rng, epi_K=K, col0=c0, advance_rng=(i == len(chunks) - 1))
        ctx.pg, ctx.n_out = pg, int(w.shape[0])
        ctx.epi = (epi, bool(relu), float(p_drop), None if bias is None else bias.shape)
        ctx.save_for_backward(x_cat, out if epi else None)
        return out

    @staticmethod
    def backward(ctx, g):
        pg = ctx.pg
        eng, nl = pg.eng, pg.n_local
        x_cat, y = ctx.saved_tensors
        g = g.contiguous()
        dev = g.device
        epi, relu, p_drop, bshape = ctx.epi
        gb = None
        if epi:
            N, K0 = int(g.shape[0]), int(g.shape[1])
            ga = torch.empty_like(g)
            gb = torch.empty(K0, dtype=torch.float32, device=dev) if bshape is not None else None
            wsb = eng.lib.ggl_bias_act_bwd_workspace_bytes(N, K0)
            ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
            eng._check(eng.lib.ggl_bias_act_bwd(_ptr(g), _ptr(y), N, K0, int(relu), float(p_drop), _ptr(ctx.rng_used),
                                                _ptr(ga), _ptr(gb), _ptr(ws), wsb, eng._stream(dev)))
            g = ga
            if gb is not None:
                gb = gb.reshape(bshape)
        gw = None
        if ctx.needs_input_grad[1]:
            K = int(g.shape[1])
            x_loc, x_halo = x_cat[:nl], x_cat[nl:]
            gl, _ = eng._spmm_fwd("sum", pg.gp_loc.bwd, pg.gp_loc.colT, pg.w_loc, g, nl)
            gw = wgrad(gl, x_loc)                                            # [K, f_in]
            del gl
            for i, (c0, c1) in enumerate(_HaloAggregate._chunks(K, pg, "const") if pg.n_halo > 0 else []):
                gh = pg._buf(("halo", i), pg.n_halo, c1 - c0, torch.float32, dev)
                eng.spmm_sum_into(pg.gp_halo.bwd, pg.gp_halo.colT, pg.w_halo, g[:, c0:c1], gh)
                gw[c0:c1] += wgrad(gh, x_halo)
            gw = gw[:ctx.n_out]
        return None, gw, None, gb, None, None, None


class _LinearSideWgrad(torch.autograd.Function):
    """y = x @ W^T whose WEIGHT gradient (a compute-bound [out, N] x [N, in] GEMM that nothing needs before
    the optimizer step) is issued on a side HIP stream, so it runs under the next layer's HBM-bound
    transposed SpMM instead of in front of it.  The input gradient stays on the main stream.

    The side-stream result is NOT returned through autograd: AccumulateGrad would read it on the main
    stream right away (``grad += gw`` when a .grad already exists) without waiting for the side stream.
    It is parked in ``sink`` and installed by ``DistGCN.join()`` after the streams are joined."""

    @staticmethod
    def forward(ctx, x, w, side, sink, pad_out=0):
        """`pad_out` extra all-zero output columns (class counts like 47 -> 48: 16-byte rows for the float4
        aggregate, produced here for free instead of by a padded copy of the [N, 47] result)."""
        ctx.save_for_backward(x, w)
        ctx.side, ctx.sink, ctx.pad_out = side, sink, int(pad_out)
        wt = w.t() if not pad_out else F.pad(w, (0, 0, 0, int(pad_out))).t()
        return x @ wt

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        side, n_out = ctx.side, w.shape[0]
        gx = None
        if ctx.needs_input_grad[0]:
            # first, and the side stream waits for it: issuing the weight-gradient GEMM beside this one (both read g)
            # was measured — two compute-bound GEMMs sharing the CUs stretch each other (2.6 -> 4.5 ms and
            # 3.6 -> 7.9 ms) and the step lost 1.3 ms; the side stream earns its keep under the HBM-bound SpMM only
            gx = g @ (w if not ctx.pad_out else F.pad(w, (0, 0, 0, ctx.pad_out)))
        gw = None
        if ctx.needs_input_grad[1]:
            g = g.contiguous()
            if side is None or not g.is_cuda:
                gw = wgrad(g, x)[:n_out]
            else:
                cur = torch.cuda.current_stream(g.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    gws = wgrad(g, x)[:n_out]
                gws.record_stream(cur)  # consumed on the main stream after join()
                # g and x are read by the side stream: they are kept alive in the sink until join() has made the
                # main stream wait for it (record_stream on these 2.5 GB tensors left the caching allocator unable
                # to reuse their blocks promptly: reserved memory crept from 64 to 117 GB over 60 steps)
                ctx.sink.append((w, gws, g, x))
        return gx, gw, None, None, None


class DistGCN(torch.nn.Module):
    """GCNModel(norm='none') on precomputed symmetric-normalised edge weights (models/gcn.py:30-64,
    gcn_conv.py:78-108 with norm='none'), each GCNConv's propagate replaced by the halo aggregate."""

    def __init__(self, feature_dim, hidden_dim, num_class, num_layers=3, drop_rate=0.5, overlap_wgrad=True,
                 aggregate_first=False, const_input_halo=True):
        """Default: every layer computes A (X W), the association GammaGL's GCNConv writes (gcn_conv.py:79).
        `aggregate_first` (an option, NOT the reference's association: results then agree with it to rounding
        only): a layer whose input is NARROWER than its output computes (A X) W instead — the same product,
        associated the cheap way round (the rule DGL's GraphConv applies).  For the first layer of the products model (100 -> 256)
        the aggregate then moves 400-byte rows instead of 1 KiB ones, and its BACKWARD needs no aggregation at all:
        the input features carry no gradient, and dW = (A X)^T dH is a GEMM on the saved aggregate."""
        super().__init__()
        self.overlap_wgrad, self.aggregate_first = overlap_wgrad, aggregate_first
        self.const_input_halo = const_input_halo   # exchange the halo rows of the (constant) input features once
        self.agg_per_step = 2 * num_layers   # aggregations one training step executes (set by forward)
        dims = [feature_dim] + [hidden_dim] * (num_layers - 1) + [num_class]
        self.lin = torch.nn.ModuleList([torch.nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])
        self.bias = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1, b)) for b in dims[1:]])
        for lin in self.lin:
            torch.nn.init.xavier_uniform_(lin.weight)
        self.dropout = torch.nn.Dropout(drop_rate)
        self.side = None  # side stream for the weight-gradient GEMMs (created on first CUDA use)
        self._sink = []   # (weight, side-stream gradient) pairs waiting for join()

    def join(self):
        """Join the side stream and install the weight gradients it produced (call after backward, before
        anything reads the .grad of a Linear weight)."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        for w, gw, *_held in self._sink:
            if w.grad is None:
                w.grad = gw
            else:
                w.grad.add_(gw)
        self._sink.clear()

    def forward(self, x, pg):
        n = len(self.lin)
        if x.is_cuda and self.side is None and self.overlap_wgrad:
            self.side = torch.cuda.Stream(device=x.device)
        n_agg = 0
        # input features never change: their halo rows are fetched once (PartitionedGraph.with_halo) and the first
        # layer runs over [x_local ; x_halo] without any exchange, forward or backward
        pre = bool(self.const_input_halo and pg.comm and not x.requires_grad)   # (the same on every rank: it decides
        #                                                                          whether layer 1 issues collectives)
        if pre:
            x = pg.with_halo(x)
        for i in range(n):
            hidden = i < n - 1
            n_out = self.lin[i].weight.shape[0]
            pad = (-n_out) % 4 if n_out >= 8 else 0   # e.g. 47 classes -> 48 columns, dropped at the end
            bias = self.bias[i] if not pad else F.pad(self.bias[i], (0, pad))
            p = self.dropout.p if hidden else 0.0
            if self.aggregate_first and x.shape[1] < n_out and x.shape[1] % 4 == 0:
                # (A X) W: aggregate the narrower side; the epilogue follows the GEMM as its own pass
                n_agg += 2 if x.requires_grad else 1
                z = pg.aggregate(x, halo_included=pre)
                h = _LinearSideWgrad.apply(z, self.lin[i].weight, self.side, self._sink, pad)
                x = pg.eng.bias_act(h, bias, relu=hidden, p_drop=p, training=self.training)
            else:
                # A (X W), GammaGL's association (gcn_conv.py:79).  With the halo rows of a constant input in place
                # the transform runs on them too: A (X W) = A_loc (X_loc W) + A_halo (X_halo W)
                n_agg += 2
                if pre and (n_out + pad) % 4 == 0:
                    # constant input, halo rows held here: the halo buffer is filled by a GEMM instead of the wire
                    x = _ConstInputLayer.apply(x, self.lin[i].weight, pg, bias, hidden,
                                               p if self.training else 0.0, pad)
                else:
                    h = _LinearSideWgrad.apply(x, self.lin[i].weight, self.side, self._sink, pad)
                    # + bias, ReLU and dropout ride on the store of the last edge block added to a row: the local SpMM
                    # on one GPU, the halo-source SpMM behind the exchange otherwise (reduce.hip MODE_SPMM_EPI)
                    x = pg.aggregate(h, bias, relu=hidden, p_drop=p, training=self.training, halo_included=pre)
            pre = False
            if pad:
                x = x[:, :n_out]
        self.agg_per_step = n_agg
        return x


class DistGCNTrainer:
    def __init__(self, pg, feature_dim, hidden_dim, num_class, num_layers=3, drop_rate=0.5, lr=0.01,
                 l2_coef=5e-4, seed=0, device="cuda", aggregate_first=False, capturable=False, const_input_halo=True):
        """`capturable`: Adam keeps its step counter on the device so that `capture()` can record the whole step
        into one hipGraph (launch-bound graphs: an arxiv-sized step is ~90 launches of 5-300 us)."""
        self.pg = pg
        torch.manual_seed(seed)  # identical initial weights on every rank
        self.net = DistGCN(feature_dim, hidden_dim, num_class, num_layers, drop_rate,
                           aggregate_first=aggregate_first, const_input_halo=const_input_halo).to(device)
        on_gpu = torch.device(device).type == "cuda"
        try:    # one fused optimizer kernel for all parameters on the GPU
            self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, weight_decay=l2_coef, fused=on_gpu,
                                        capturable=bool(capturable) and on_gpu)
        except (RuntimeError, TypeError):
            self.opt = torch.optim.Adam(self.net.parameters(), lr=lr, weight_decay=l2_coef)
        self.graph = None
        torch.manual_seed(seed + 1000 * (pg.rank + 1))  # independent dropout masks per rank

    def step(self, x_local, y_local, train_local, n_train_global):
        pg = self.pg
        self.net.train()
        self.opt.zero_grad(set_to_none=True)
        logits = self.net(x_local, pg)
        loss = F.cross_entropy(logits[train_local], y_local[train_local], reduction="sum") / n_train_global
        loss.backward()
        self.net.join()
        if pg.world > 1:  # one flat bucket: 3 small weight matrices + biases
            params = [p for p in self.net.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat, group=pg.group)
            o = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.opt.step()
        return loss.detach()

    def tune_halo_chunks(self, x_local, y_local, train_local, n_train_global, candidates=(1, 2, 4), iters=2):
        """MEASURE how many column chunks the halo exchange should run in, on this machine and this partition, instead
        of assuming: each candidate runs `iters` forward + backward passes of the model (no optimizer step, gradients
        dropped, the dropout stream put back afterwards: the training trajectory is unchanged — chunking never
        changes a value, only when columns are computed), the slowest rank's time decides (a MAX all-reduce, so every
        rank picks the same count — they must: the count is the number of collectives issued), first for the layers
        that exchange, then for the constant-input first layer.  Call it on every rank, before `capture()`.
        Returns {"exchange": n, "const": n, "ms": {...}}.  GGL_HALO_CHUNKS set: nothing is measured."""
        pg = self.pg
        out = {"exchange": None, "const": None, "ms": {}}
        if HALO_CHUNKS > 0 or not pg.comm:
            return out
        dev = x_local.device
        on_gpu = dev.type == "cuda"
        real = pg.world > 1 and not pg.dry
        rng = pg.eng._rng_state(dev)
        rng_saved = rng.clone() if rng is not None else None

        def fb():
            self.net.train()
            self.opt.zero_grad(set_to_none=True)
            logits = self.net(x_local, pg)
            (F.cross_entropy(logits[train_local], y_local[train_local], reduction="sum") / n_train_global).backward()
            self.net.join()

        def timed():
            fb()                                                  # buffers of this layout allocated, caches warm
            if on_gpu:
                torch.cuda.synchronize(dev)
            if real:
                dist.barrier(group=pg.group)
            t0 = time.perf_counter()
            for _ in range(iters):
                fb()
            if on_gpu:
                torch.cuda.synchronize(dev)
            t = torch.tensor([(time.perf_counter() - t0) * 1e3 / iters], dtype=torch.float64, device=dev)
            if real:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=pg.group)
            return float(t.item())

        if on_gpu:
            torch.cuda.synchronize(dev)   # nothing of an earlier step may still read the buffers dropped below
        for kind in ("exchange", "const"):
            best = None
            for n in candidates:
                pg.halo_chunks[kind] = int(n)
                pg._bufs.clear()      # one layout's exchange buffers alive at a time (a candidate's are up to 2 x n_halo x K)
                ms = timed()
                out["ms"][f"{kind}={n}"] = round(ms, 3)
                if best is None or ms < best[0]:
                    best = (ms, int(n))
            pg.halo_chunks[kind] = out[kind] = best[1]
        pg._bufs.clear()          # only the layout that won stays allocated
        self.opt.zero_grad(set_to_none=True)
        if rng_saved is not None:
            rng.copy_(rng_saved)
        return out

    def capture(self, x_local, y_local, train_local, n_train_global, warmup=3, collectives=False):
        """Record step(...) on these very tensors into one hipGraph (`capturable=True`); afterwards `replay()` runs a
        training step — the kernels never sync or allocate outside the graph's pool, the side stream's weight-gradient
        GEMMs and the dropout draws (state advanced on the device) are part of it.  Single-rank steps (and dry
        partitions) by default.  A step that exchanges halos contains RCCL collectives: recording those crashed in round 3
        (hipStreamEndCapture segfault, profiles/r3_rccl_capture_attempt.txt) because the communicator met its first
        collective INSIDE the capture; with the communicator warmed by eager steps first — which `warmup` does — an
        all-reduce and an all-to-all-v with split sizes record and replay correctly on this stack in "thread_local"
        capture mode (torch 2.10 + ROCm 7.0 + RCCL, world-size-1 group on one MI355X: profiles/r4_rccl_capture_retry.txt).
        That is one GPU's evidence, so the N > 1 capture stays opt-in: `collectives=True` or GGL_CAPTURE_COLLECTIVES=1;
        the eager N > 1 step is what bench.py times."""
        want = collectives or os.environ.get("GGL_CAPTURE_COLLECTIVES") == "1"
        if self.pg.comm and not self.pg.dry and not want:
            raise RuntimeError("this step exchanges halos through RCCL; recording collectives into a hipGraph is opt-in "
                               "(collectives=True / GGL_CAPTURE_COLLECTIVES=1, see the docstring)")
        from .trainer import GraphedStep

        self.graph = GraphedStep(lambda: self.step(x_local, y_local, train_local, n_train_global), warmup=warmup,
                                 capture_error_mode="thread_local" if want else "global")
        return self.graph

    def replay(self):
        return self.graph()


def build_partition(n_nodes, n_edges, seed, rank, world, group, dev, eng, relabel="random", order="src",
                    parts=None, stats=None, kind="rmat"):
    """This rank's PartitionedGraph of the synthetic benchmark graph, built WITHOUT a global edge list:
    `synth.rmat_partitioned` hands every rank only the edges whose destination it owns (same graph for
    every world size), the halo bookkeeping exchanges node ids only.  `parts` > world == 1: a dry
    partition — this process plays rank `rank` of `parts` (one GPU measuring one rank's share).
    `kind` = "planted" (a graph with community structure) needs the whole graph in one place: every rank builds it
    itself (`synth.full_graph_partitioned` — graphs that fit one GPU, i.e. everything up to the products size), and so
    does `relabel` = "cluster" (the locality-aware order of `partition.cluster_order`) in a single process.  With
    world > 1 the R-MAT shares are relabelled IN PLACE: `partition.cluster_order_distributed` sweeps over the shares
    (halo labels by all-to-all), `synth.repartition` routes the renamed edges to their new owners — same result as
    the single-process order, at any graph size."""
    from .synth import _Comm, full_graph_partitioned, repartition, rmat_partitioned

    if kind == "rmat" and relabel == "cluster" and world > 1 and (parts or world) == world:
        from .partition import cluster_order_distributed

        comm = _Comm(rank, world, group)
        g = rmat_partitioned(n_nodes, n_edges, seed=seed, rank=rank, world=world, group=group, device=dev,
                             relabel="random", order=order, stats=stats)
        clusters = max(8, min(8192, int(n_nodes) // 600))
        new_id, lab = cluster_order_distributed(g, comm, clusters=clusters, sweeps=30, seed=seed)
        g = repartition(g, new_id, comm, order=order)
        if stats is not None:
            stats.update(clusters=clusters, local_edges=int(g["src"].numel()))
    elif kind != "rmat" or relabel == "cluster":
        g = full_graph_partitioned(kind, n_nodes, n_edges, seed=seed, rank=rank, world=world, device=dev,
                                   relabel=relabel, order=order, parts=parts, stats=stats, eng=eng)
    else:
        g = rmat_partitioned(n_nodes, n_edges, seed=seed, rank=rank, world=world, group=group, device=dev,
                             relabel=relabel, order=order, parts=parts, stats=stats)
    dry = world == 1 and (parts or 1) > 1
    pg = PartitionedGraph.from_local(g["src"], g["dst"], g["w"], g["bounds"], n_nodes, g["e_global"],
                                     rank=g["rank"], world=world, group=group, eng=eng,
                                     send_rows=g.get("send_rows") if dry else None)
    return pg
