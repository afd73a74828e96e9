"""Harness-side callers of the hot path: torch restatements of the
reference layers that sit directly on ``gammagl.mpops`` — just enough of them to drive the ops the
way GammaGL does.  The reference classes are TensorLayerX modules (``tlx`` is not installable
here); every tlx call they make on this path maps 1:1 onto torch (``tlx.gather`` = index_select,
``tlx.pow``, ``tlx.reshape`` ...), so these are line-for-line the same computations:

* ``MessagePassing``  — layers/conv/message_passing.py:35-156 (message / aggregate /
  message_aggregate / propagate, including the "fused route only if the class defines
  message_aggregate itself" rule at :144)
* ``GCNConv``         — layers/conv/gcn_conv.py:78-115, with the reference's own (commented-out)
  ``message_aggregate`` -> ``gspmm`` override at :110-115 enabled, which is what makes
  ogbn-products-sized graphs fit one GPU (the unfused route materialises an [E,K] = 129 GB message)
* ``SAGEConv``        — layers/conv/sage_conv.py:56-108 ('mean', 'gcn' and 'pool' aggregators)
* ``GATConv``         — layers/conv/gat_conv.py:98-122 (unfused: segment_softmax + propagate)
* ``FusedGATConv``    — layers/conv/fusedgat_conv.py:89-130 with our single-kernel op in place of
  the external dgNN GATConvFuse
* ``GCNModel``        — models/gcn.py:30-64
"""
import torch
from torch import nn

from . import engine as _engine
from .dense import Linear, _LinearFn, node_matmul
from .sampler import Block
from .mpops import (bspmm, gspmm, unsorted_segment_max, unsorted_segment_mean,  # noqa: F401
                    unsorted_segment_sum, use_ext)


def degree(index, num_nodes, dtype=torch.float32):
    """utils/degree.py:10-40: unsorted_segment_sum(ones[E], index, N) (K = 1).

    The sum of ones per segment is the segment's element count, which the cached plan of ``index`` already
    holds as rowptr differences — no kernel, no pass over E.  Float dtypes clamp at
    the value where the reference's ``+= 1`` accumulation in that dtype stops growing (2^24 for f32)."""
    if not index.is_cuda or index.dim() != 1 or index.dtype != torch.int64:
        one = torch.ones((index.shape[0],), dtype=dtype, device=index.device)
        return unsorted_segment_sum(one, index, num_nodes)
    cnt = _engine(index).seg_plan(index, num_nodes).counts()
    sat = {torch.float32: 1 << 24, torch.float16: 2048, torch.bfloat16: 256}.get(dtype)
    if sat is not None:
        cnt = cnt.clamp(max=sat)
    return cnt.to(dtype)


def calc_gcn_norm(edge_index, num_nodes, edge_weight=None):
    """utils/norm.py:5-30."""
    src, dst = edge_index[0], edge_index[1]
    if edge_weight is None:
        edge_weight = torch.ones((edge_index.shape[1], 1), device=edge_index.device)
    deg = unsorted_segment_sum(edge_weight, src, num_segments=num_nodes).reshape(-1)
    deg_inv_sqrt = deg.pow(-0.5)
    return deg_inv_sqrt[src] * edge_weight.reshape(-1) * deg_inv_sqrt[dst]


def segment_softmax(data, segment_ids, num_segments):
    """utils/softmax.py:10-36."""
    max_values = unsorted_segment_max(data, segment_ids, num_segments=num_segments)
    exp = torch.exp(data - max_values[segment_ids])
    denominator = unsorted_segment_sum(exp, segment_ids, num_segments=num_segments)
    return exp / (denominator[segment_ids] + 1e-16)


def add_self_loops(edge_index, num_nodes):
    """utils/loop.py:57-142 with n_loops=1, no edge attributes: loops are appended at the end."""
    loops = torch.arange(num_nodes, device=edge_index.device, dtype=edge_index.dtype)
    return torch.cat([edge_index, torch.stack([loops, loops])], dim=1)


FUSED_MIN_EDGES = 2_000_000  # default message()+aggregate() pairs take the fused SpMM from this many edges up
SAGE_FUSE_EPILOGUE = True    # SAGEConv(mean): "+ fc_self(x_dst) + bias -> act" in the aggregate's store (A/B switch)


class MessagePassing(nn.Module):
    def message(self, x, edge_index, edge_weight=None):
        msg = x.index_select(0, edge_index[0, :])
        if edge_weight is not None:
            return msg * edge_weight.unsqueeze(-1)
        return msg

    _REDUCERS = {'sum': unsorted_segment_sum, 'mean': unsorted_segment_mean, 'max': unsorted_segment_max}

    def aggregate(self, msg, edge_index, num_nodes=None, aggr='sum'):
        """Reduce the messages onto their destination nodes (message_passing.py:63-92: same signature, same three
        reducers over edge_index[1])."""
        reduce = self._REDUCERS.get(aggr)
        if reduce is None:
            raise NotImplementedError(f'aggr={aggr!r}: this layer aggregates with sum, mean or max')
        return reduce(msg, edge_index[1, :], num_nodes)

    def message_aggregate(self, x, edge_index, edge_weight=None, aggr='sum'):
        if use_ext:
            if edge_weight is None:
                edge_weight = torch.ones(edge_index.shape[1], device=x.device, dtype=x.dtype)
            return gspmm(edge_index, edge_weight, x, aggr)
        msg = self.message(x, edge_index, edge_weight)
        return self.aggregate(msg, edge_index)

    def update(self, x):
        return x

    def propagate(self, x, edge_index, aggr='sum', **kwargs):
        if kwargs.get('num_nodes') is None:
            kwargs['num_nodes'] = x.shape[0]
        if 'message_aggregate' in self.__class__.__dict__:  # message_passing.py:144
            x = self.message_aggregate(x, edge_index, edge_weight=kwargs.get('edge_weight'), aggr=aggr)
        elif (aggr in ('sum', 'mean') and edge_index.shape[1] >= FUSED_MIN_EDGES and x.dim() == 2
              and x.dtype == torch.float32 and type(self).message is MessagePassing.message
              and type(self).aggregate is MessagePassing.aggregate
              # gspmm treats the edge weight as a constant (gspmm.cpp:30); a weight that needs a gradient
              # must stay on the message() route, which differentiates through the multiply — and so must a
              # weight that is not one f32 value per edge (the message() route promotes / broadcasts it)
              and not (kwargs.get('edge_weight') is not None and
                       (kwargs['edge_weight'].requires_grad or kwargs['edge_weight'].dtype != torch.float32
                        or kwargs['edge_weight'].numel() != edge_index.shape[1]))):
            # The default message() (gather * weight) + aggregate() pair IS an SpMM.  A big (full-graph) edge
            # list takes the fused rectangular kernel: no [E, K] message tensor (Reddit-sized SAGEConv layer:
            # 155 -> 14.7 ms forward+backward, 59 GB less HBM), same sums in the same order.  Sampled blocks stay
            # below the threshold on purpose: they are NEW edge lists every batch and the fused backward would
            # need a transposed plan (a sort + host syncs) each time — measured 4.5 vs 3.1 ms per batch.
            eng = _engine(x)
            gp = eng.graph_plan(edge_index, int(kwargs['num_nodes']), int(x.shape[0]))
            ew = kwargs.get('edge_weight')
            x = eng.spmm(gp, None if ew is None else ew.reshape(-1).contiguous(), x, aggr)
        elif (aggr == 'sum' and edge_index.shape[1] >= FUSED_MIN_EDGES and x.dim() == 3 and x.dtype == torch.float32
              and kwargs.get('edge_weight') is not None and kwargs['edge_weight'].dim() == 2
              and kwargs['edge_weight'].dtype == torch.float32
              and kwargs['edge_weight'].shape == (edge_index.shape[1], x.shape[1])
              and int(kwargs['num_nodes']) == x.shape[0] and type(self).message is MessagePassing.message
              and type(self).aggregate is MessagePassing.aggregate):
            # multi-head messages x[src,h,:] * w[e,h] summed per destination == bspmm (what GATConv's own
            # commented-out message_aggregate would call, gat_conv.py:124-129): no [E, H, C] message tensor
            x = bspmm(edge_index, kwargs['edge_weight'], x, 'sum')
        else:
            msg = self.message(x, edge_index, edge_weight=kwargs.get('edge_weight'))
            x = self.aggregate(msg, edge_index, num_nodes=kwargs['num_nodes'], aggr=aggr)
        return self.update(x)


CACHE_GCN_NORM = True    # False: every GCNConv.forward recomputes its normalisation, as the reference does (bench.py's side figure)


class GCNConv(MessagePassing):
    def __init__(self, in_channels, out_channels, norm='both', add_bias=True):
        super().__init__()
        if norm not in ['left', 'right', 'none', 'both']:
            raise ValueError('Invalid norm value. Must be either "none", "both", "right" or "left".'
                             ' But got "{}".'.format(norm))
        self._norm = norm
        self.linear = Linear(in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.linear.weight)
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if add_bias else None

    def _norm_weights(self, edge_index, edge_weight, num_nodes, device):
        """gcn_conv.py:84-102: deg^-1/2 (or 1/deg) of the source and/or destination end of every edge, degrees
        counted on the edge list as given.  The reference recomputes this in every layer of every step; with
        edge_weight=None it depends on the graph alone, so it is computed once per edge_index and kept on the
        cached GraphPlan (the same tensor every call also lets the SpMM stream its sorted copy)."""
        gp = None
        if edge_weight is None and CACHE_GCN_NORM:
            gp = _engine(edge_index).graph_plan(edge_index, num_nodes)
            hit = gp.aux.get(("gcn_norm", self._norm))
            if hit is not None:
                return hit
        src, dst = edge_index[0], edge_index[1]
        ew = torch.ones((edge_index.shape[1],), device=device) if edge_weight is None else edge_weight.reshape(-1)
        weights = ew
        if self._norm in ['left', 'both']:
            deg = degree(src, num_nodes=num_nodes, dtype=torch.float32)
            norm = deg.pow(-0.5) if self._norm == 'both' else 1.0 / deg
            weights = norm[src] * ew
        if self._norm in ['right', 'both']:
            deg = degree(dst, num_nodes=num_nodes, dtype=torch.float32)
            norm = deg.pow(-0.5) if self._norm == 'both' else 1.0 / deg
            weights = weights * norm[dst]
        if gp is not None:
            gp.aux[("gcn_norm", self._norm)] = weights
        return weights

    def forward(self, x, edge_index, edge_weight=None, num_nodes=None, _epilogue=None):
        """`_epilogue=(relu, p_drop, training)` (used by GCNModel): the ReLU and dropout the model applies right
        after this layer, fused with the bias into the aggregate's store (or one pass after it)."""
        n_out = self.linear.weight.shape[0]
        pad = (-n_out) % 4 if n_out >= 8 else 0
        if pad and x.dim() == 2:  # class-count widths: 47 -> 48 columns inside the GEMM, dropped at the end
            x = _LinearFn.apply(x.contiguous(), torch.nn.functional.pad(self.linear.weight, (0, 0, 0, pad)))
        else:
            x = self.linear(x)
        num_nodes = x.shape[0]
        weights = self._norm_weights(edge_index, edge_weight, num_nodes, x.device)
        relu, p_drop, training = _epilogue if _epilogue is not None else (False, 0.0, False)
        bias = self.bias
        if pad and bias is not None:
            bias = torch.nn.functional.pad(bias, (0, pad))
        if (weights.requires_grad or weights.dtype != torch.float32 or weights.dim() != 1
                or weights.numel() != edge_index.shape[1]):
            # a learnable edge weight: gspmm treats weights as constants (gspmm.cpp:30), so stay on the
            # message() * weight -> unsorted_segment_sum route, which differentiates through the multiply; the
            # same route takes weights that are not one f32 value per edge (a float64 edge_weight promotes the
            # messages exactly as the reference's message() does, a mis-shaped one raises there as it does here)
            out = self.aggregate(self.message(x, edge_index, weights), edge_index, num_nodes, 'sum')
            if out.dtype != torch.float32:   # promoted messages: the epilogue in torch, as the reference runs it
                out = out + bias if bias is not None else out
                out = torch.relu(out) if relu else out
                out = torch.nn.functional.dropout(out, p_drop, training) if p_drop > 0 else out
            elif bias is not None or relu or p_drop > 0:
                out = _engine(out).bias_act(out, bias, relu=relu, p_drop=p_drop, training=training)
        elif (x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0
              and weights.dtype == torch.float32 and weights.numel() == edge_index.shape[1]):
            eng = _engine(x)
            out = eng.spmm_bias_act(eng.graph_plan(edge_index, num_nodes), weights, x, bias, relu=relu,
                                    p_drop=p_drop, training=training)
        else:
            out = self.propagate(x, edge_index, edge_weight=weights, num_nodes=num_nodes)
            if bias is not None or relu or p_drop > 0:
                out = _engine(out).bias_act(out, bias, relu=relu, p_drop=p_drop, training=training)
        return out[:, :n_out] if pad else out

    def message_aggregate(self, x, edge_index, edge_weight=None, aggr="sum"):  # gcn_conv.py:110-115
        if edge_weight is None:
            edge_weight = torch.ones((edge_index.shape[1],), dtype=torch.float32, device=x.device)
        return gspmm(edge_index, edge_weight, x, aggr)


class SAGEConv(MessagePassing):
    """layers/conv/sage_conv.py.  ASSOCIATION: the reference always transforms first, mean(fc_neigh(x_src))
    (sage_conv.py:100).  With aggr='mean', f32 features and a layer whose input is NARROWER than its output this
    class computes fc_neigh(mean(x_src)) instead — the same product associated the cheap way round (the rule DGL's
    SAGEConv applies): results then agree with the reference to f32 rounding (1e-5 relative), not bit for bit.
    Set `gammagl_amd.layers.SAGE_FUSE_EPILOGUE = False` for the reference's order of operations throughout."""

    def __init__(self, in_channels, out_channels, activation=None, aggr="mean", add_bias=True):
        super().__init__()
        self.aggr = aggr
        self.act = activation
        self.fc_neigh = Linear(in_channels, out_channels, bias=False)
        if aggr != 'gcn':
            self.fc_self = Linear(in_channels, out_channels, bias=False)
        if aggr == 'pool':
            self.pool = Linear(in_channels, in_channels, bias=False)
        if aggr == 'lstm':   # sage_conv.py:46-47
            self.in_feat = in_channels
            self.lstm = nn.LSTM(input_size=in_channels, hidden_size=in_channels, batch_first=True)
        self.bias = nn.Parameter(torch.zeros(1, out_channels)) if add_bias else None

    def forward(self, feat, edge):
        src_feat, dst_feat = feat if isinstance(feat, tuple) else (feat, feat)
        num_nodes = int(dst_feat.shape[0])
        if isinstance(edge, Block):   # a static-shape block of the BlockSampler (graph-capturable step)
            if self.aggr != 'mean':
                raise NotImplementedError("sampler Blocks carry the mean aggregator")
            fused_act = self.act is None or self.act is torch.relu or self.act is torch.nn.functional.relu
            w_n, w_s = self.fc_neigh.weight, self.fc_self.weight
            if w_n.shape[1] < w_n.shape[0] and src_feat.shape[1] % 4 == 0:
                # input narrower than output: aggregate first (the rule DGL's SAGEConv applies; sage_conv.py:100 always
                # transforms first).  In a sampled block this also shrinks the GEMM from the block's SOURCE rows to its
                # destination rows (100 608 -> 18 432 on the products-sized graph) and, for the first layer, whose
                # input rows carry no gradient, removes the backward aggregate together with the block's CSC build.
                agg = edge.eng.block_mean_epi(src_feat, edge)
                out = torch.nn.functional.linear(torch.cat([agg, dst_feat], dim=1), torch.cat([w_n, w_s], dim=1))
                if self.bias is not None or self.act is not None:
                    out = edge.eng.bias_act(out, self.bias, relu=fused_act and self.act is not None)
                return out if fused_act else self.act(out)
            out = edge.eng.block_mean_epi(self.fc_neigh(src_feat), edge, add=self.fc_self(dst_feat), bias=self.bias,
                                           relu=fused_act and self.act is not None)
            return out if fused_act else self.act(out)
        if self.aggr == 'mean':
            fused_act = self.act is None or self.act is torch.relu or self.act is torch.nn.functional.relu
            w_n = self.fc_neigh.weight
            if (SAGE_FUSE_EPILOGUE and w_n.shape[1] < w_n.shape[0] and src_feat.dim() == 2 and src_feat.shape[1] % 4 == 0
                    and src_feat.dtype == torch.float32 and fused_act):
                # input narrower than output: aggregate first, transform the (fewer, in a sampled block) destination
                # rows afterwards — see the Block branch above
                eng = _engine(src_feat)
                if edge.shape[1] >= FUSED_MIN_EDGES:
                    gp = eng.graph_plan(edge, num_nodes, int(src_feat.shape[0]))
                    agg = eng.spmm(gp, None, src_feat, "mean")
                else:
                    agg = unsorted_segment_mean(src_feat.index_select(0, edge[0]), edge[1], num_nodes)
                out = torch.nn.functional.linear(torch.cat([agg, dst_feat], dim=1), torch.cat([w_n, self.fc_self.weight], dim=1))
                if self.bias is not None or self.act is not None:
                    out = eng.bias_act(out, self.bias, relu=self.act is not None)
                return out
            src_feat = self.fc_neigh(src_feat)
            big = edge.shape[1] >= FUSED_MIN_EDGES
            # (a small block whose width is not a multiple of 4 — e.g. the 47-class output layer — has no fused form:
            # it goes straight to propagate() below without computing fc_self / the gather here first and again there)
            if (SAGE_FUSE_EPILOGUE and src_feat.dim() == 2 and src_feat.dtype == torch.float32 and fused_act
                    and (big or src_feat.shape[1] % 4 == 0)):
                # "mean + fc_self(x_dst) + bias -> act" (sage_conv.py:100-108) rides on the aggregate's store:
                # the fused rectangular SpMM-mean for big edge lists, the segment route for sampled blocks
                eng = _engine(src_feat)
                self_term = self.fc_self(dst_feat)
                if big:
                    gp = eng.graph_plan(edge, num_nodes, int(src_feat.shape[0]))
                    return eng.spmm_epi(gp, None, src_feat, "mean", add=self_term, bias=self.bias,
                                        relu=self.act is not None)
                return eng.segment_epi(src_feat.index_select(0, edge[0]), edge[1], num_nodes, "mean", add=self_term,
                                       bias=self.bias, relu=self.act is not None)
            # (propagate picks the fused SpMM-mean for big edge lists, the segment route for sampled blocks)
            out = self.propagate(src_feat, edge, edge_weight=None, num_nodes=num_nodes, aggr='mean')
        elif self.aggr == 'gcn':
            src_feat = self.fc_neigh(src_feat)
            n = int(1 + edge[0].max())
            edge = add_self_loops(edge, n)
            weight = calc_gcn_norm(edge, n)
            out = self.propagate(src_feat, edge, edge_weight=weight, num_nodes=n, aggr='sum')
            out = out[:num_nodes]
        elif self.aggr == 'pool':
            src_feat = torch.relu(self.pool(src_feat))
            out = self.propagate(src_feat, edge, edge_weight=None, num_nodes=num_nodes, aggr='max')
            out = self.fc_neigh(out)
        elif self.aggr == 'lstm':
            # sage_conv.py:93-98: no graph op at all — the source rows are taken as [N_dst, fan-out, D] sequences in
            # the order they arrive (the edge list is not consulted) and the LSTM's last hidden state is the aggregate
            size = dst_feat.shape[0]
            seq = src_feat.reshape(size, -1, src_feat.shape[1])
            h0 = (torch.zeros((1, size, self.in_feat), dtype=seq.dtype, device=seq.device),
                  torch.zeros((1, size, self.in_feat), dtype=seq.dtype, device=seq.device))
            _, (rst, _) = self.lstm(seq, h0)
            out = self.fc_neigh(rst[0])
        else:
            raise NotImplementedError(self.aggr)
        if self.aggr != 'gcn':
            out = out + self.fc_self(dst_feat)
        if self.bias is not None:
            out = out + self.bias
        return self.act(out) if self.act is not None else out


class GATConv(MessagePassing):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout_rate=0.,
                 add_bias=True):
        super().__init__()
        self.heads, self.out_channels, self.concat = heads, out_channels, concat
        self.negative_slope = negative_slope
        self.dropout_rate = dropout_rate
        self.dropout = nn.Dropout(dropout_rate)   # on the attention coefficients (gat_conv.py:90,104)
        self.w = nn.Parameter(torch.empty(in_channels, out_channels * heads))
        self.att = nn.Parameter(torch.empty(1, heads, out_channels * 2))
        nn.init.trunc_normal_(self.w, std=0.05)
        nn.init.trunc_normal_(self.att, std=0.05)
        nb = heads * out_channels if concat else out_channels
        self.bias = nn.Parameter(torch.zeros(nb)) if add_bias else None

    def _finish(self, x):
        if self.concat:
            x = x.reshape(-1, self.heads * self.out_channels)
        else:
            x = x.mean(dim=1)
        return x + self.bias if self.bias is not None else x

    def forward(self, x, edge_index, num_nodes=None):
        x = node_matmul(x, self.w).reshape(-1, self.heads, self.out_channels)
        node_src, node_dst = edge_index[0, :], edge_index[1, :]
        feat = torch.cat((x[node_src], x[node_dst]), dim=-1)
        e = (feat * self.att).sum(dim=-1)
        e = torch.nn.functional.leaky_relu(e, self.negative_slope)
        alpha = self.dropout(segment_softmax(e, node_dst, num_nodes))
        x = self.propagate(x, edge_index, num_nodes=num_nodes, edge_weight=alpha)
        return self._finish(x)


class FusedGATConv(GATConv):
    """Same parameters and math as GATConv; logits, softmax, attention dropout and aggregate run in one HIP
    kernel per direction.  The reference layer (fusedgat_conv.py:89-130) optionally takes a prebuilt
    CSR/CSC (`row_ptr`, `col_ind`, `col_ptr`, `row_ind`, `permute`) to skip its numpy preprocessing; here
    the destination-sorted plan is built on the device once per edge_index and cached, so those keywords are
    accepted and not needed.  Aggregates into edge_index[1] like GATConv (the reference's fused layer builds
    its CSR on edge_index[0], which only coincides on symmetric graphs — SURVEY.md §8a row G)."""

    def forward(self, x, edge_index, num_nodes=None, **kwargs):
        H, C = self.heads, self.out_channels
        eng = _engine(x)
        if 'row_ptr' in kwargs:
            # fusedgat_conv.py:95-100: the caller's own CSR (rows = the aggregating nodes, col_ind = the nodes they
            # gather from), its transpose and the CSC -> CSR position map: taken as they are, no sort, no host trip
            edge_index = eng.graph_plan_from_csr(kwargs['row_ptr'], kwargs['col_ind'], kwargs['col_ptr'],
                                                 kwargs['row_ind'], kwargs['permute'])
        if (not self.concat and x.dim() == 2 and x.dtype == torch.float32 and (num_nodes is None or num_nodes == x.shape[0])
                and eng.gat_headmean_supported(H, x.shape[1], C)):
            # a head-averaging layer whose input row (F floats) is narrower than its H x C transformed row: aggregate
            # the input per head, transform afterwards (same math, 1408 B -> 256 B gathered per edge on the Reddit GAT)
            y = eng.gat_headmean(edge_index, x, self.w, self.att, self.negative_slope, num_nodes=x.shape[0],
                                 dropout_rate=self.dropout_rate, training=self.training)
            # (bias_add: the bias gradient as the library's two-stage column sum instead of a torch reduce over [N, C])
            return eng.bias_add(y, self.bias) if self.bias is not None else y
        w = self.w
        pad = (-C) % 4 if C >= 8 else 0
        if pad:  # e.g. 41 classes per head: 44 channels inside the GEMM keep the kernels on 16-byte slices
            w = torch.nn.functional.pad(w.reshape(-1, H, C), (0, pad)).reshape(-1, H * (C + pad))
        x = node_matmul(x, w).reshape(-1, H, C + pad)
        el = (x[:, :, :C] * self.att[:, :, :C]).sum(dim=-1)   # source term  a_src . x_j
        er = (x[:, :, :C] * self.att[:, :, C:]).sum(dim=-1)   # destination term a_dst . x_i
        x = _engine(x).gat_fused(edge_index, el, er, x, self.negative_slope, num_nodes=num_nodes,
                                dropout_rate=self.dropout_rate, training=self.training)
        return self._finish(x[:, :, :C] if pad else x)


class GCNModel(nn.Module):
    """models/gcn.py:30-64."""

    def __init__(self, feature_dim, hidden_dim, num_class, drop_rate=0.2, num_layers=2, norm='both'):
        super().__init__()
        self.num_layers = num_layers
        if num_layers == 1:
            self.conv = nn.ModuleList([GCNConv(feature_dim, num_class, norm=norm)])
        else:
            self.conv = nn.ModuleList([GCNConv(feature_dim, hidden_dim, norm=norm)])
            for _ in range(1, num_layers - 1):
                self.conv.append(GCNConv(hidden_dim, hidden_dim, norm=norm))
            self.conv.append(GCNConv(hidden_dim, num_class, norm=norm))
        self.dropout = nn.Dropout(drop_rate)

    def forward(self, x, edge_index, edge_weight, num_nodes):
        if self.num_layers == 1:
            return self.conv[0](x, edge_index, edge_weight, num_nodes)
        for i in range(self.num_layers - 1):
            # relu(conv(x)) then dropout (models/gcn.py:55-59), handed to the layer so that bias, ReLU and
            # dropout ride on the aggregate's store instead of three more passes over [N, hidden]
            x = self.conv[i](x, edge_index, edge_weight, num_nodes,
                             _epilogue=(True, self.dropout.p, self.training))
        return self.conv[-1](x, edge_index, edge_weight, num_nodes)


class GATModel(nn.Module):
    """models/gat.py:4-73: dropout -> GATConv (-> ELU) per layer, attention dropout inside the conv, the
    last layer averages its heads.  `fused=True` uses FusedGATConv (one kernel per direction)."""

    def __init__(self, feature_dim, hidden_dim, num_class, heads, drop_rate, num_layers, fused=True):
        super().__init__()
        conv = FusedGATConv if fused else GATConv
        if num_layers == 1:
            hidden_dim = num_class
        self.gat_list = nn.ModuleList()
        for i in range(num_layers):
            if i == 0:
                self.gat_list.append(conv(feature_dim, hidden_dim, heads=heads, dropout_rate=drop_rate, concat=True))
            elif i == num_layers - 1:
                self.gat_list.append(conv(hidden_dim * heads, num_class, heads=heads, dropout_rate=drop_rate,
                                          concat=False))
            else:
                self.gat_list.append(conv(hidden_dim * heads, hidden_dim, heads=heads, dropout_rate=drop_rate,
                                          concat=True))
        self.dropout = nn.Dropout(drop_rate)

    def forward(self, x, edge_index, num_nodes):
        for i, gat in enumerate(self.gat_list):
            x = gat(self.dropout(x), edge_index, num_nodes)
            if i < len(self.gat_list) - 1:
                x = torch.nn.functional.elu(x)
        return x


class GraphSAGEFullModel(nn.Module):
    """models/graphsage.py:7-32 (GraphSAGE_Full_Model): full-graph GraphSAGE — dropout on the input, n_layers
    hidden SAGEConv layers with `activation`, an output SAGEConv without, dropout between layers.  On a big edge
    list every mean / gcn aggregate takes the fused SpMM route (MessagePassing.propagate)."""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout, aggregator_type):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        convs = [SAGEConv(in_feats, n_hidden, activation, aggregator_type)]
        convs += [SAGEConv(n_hidden, n_hidden, activation, aggregator_type) for _ in range(n_layers - 1)]
        convs += [SAGEConv(n_hidden, n_classes, None, aggregator_type)]
        self.convs = nn.ModuleList(convs)

    def forward(self, feat, edge):
        h = self.dropout(feat)
        for i, layer in enumerate(self.convs):
            h = layer(h, edge)
            if i != len(self.convs) - 1:
                h = self.dropout(h)
        return h


class GraphSAGESampleModel(nn.Module):
    """models/graphsage.py:35-83 (GraphSAGE_Sample_Model): SAGEConv(mean) per sampled hop; the target
    nodes of a block are the first size[1] rows of its input ("target nodes are always placed first")."""

    def __init__(self, in_feat, hid_feat, out_feat, drop_rate=0.0, num_layers=2):
        super().__init__()
        self.dropout = nn.Dropout(drop_rate)
        if num_layers == 1:
            self.convs = nn.ModuleList([SAGEConv(in_feat, out_feat)])
        else:
            convs = [SAGEConv(in_feat, hid_feat, torch.relu)]
            convs += [SAGEConv(hid_feat, hid_feat, torch.relu) for _ in range(num_layers - 2)]
            convs += [SAGEConv(hid_feat, out_feat)]
            self.convs = nn.ModuleList(convs)

    def forward(self, x, adjs):
        adjs = adjs if isinstance(adjs, (list, tuple)) else [adjs]
        for i, (conv, adj) in enumerate(zip(self.convs, adjs)):
            x = conv((x, x[: adj.size[1]]), adj if isinstance(adj, Block) else adj.edge_index)
            if i != len(self.convs) - 1:
                x = self.dropout(x)
        return x

    @torch.no_grad()
    def inference(self, feat, sampler, batch_size=4096):
        """models/graphsage.py:85-103: layer by layer over ALL nodes with one full-neighbourhood hop per batch
        (a NeighborSampler built with sample_lists=[-1]); returns the logits of every node."""
        n = feat.shape[0]
        for i, layer in enumerate(self.convs):
            out = None
            for b in range(0, n, batch_size):
                dst = torch.arange(b, min(n, b + batch_size), device=feat.device)
                _, n_id, adj = sampler.sample(dst)
                adj = adj[0] if isinstance(adj, (list, tuple)) else adj
                h = feat[n_id]
                h = layer((h, h[: adj.size[1]]), adj.edge_index)
                if out is None:
                    out = torch.empty((n, h.shape[1]), dtype=h.dtype, device=h.device)
                out[dst] = h
            feat = out
        return feat


__all__ = ["MessagePassing", "GCNConv", "SAGEConv", "GATConv", "FusedGATConv", "GCNModel", "GATModel", "GraphSAGEFullModel", "GraphSAGESampleModel", "degree",
           "calc_gcn_norm", "segment_softmax", "add_self_loops"]
