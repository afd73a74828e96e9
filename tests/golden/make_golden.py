#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (this container only).

What runs: the reference's Python op surface ``/root/reference/gammagl/mpops/torch.py`` imported
by file path into a throw-away stub package under /tmp, on top of the reference's own CPU
extension built by ``oracle/Makefile`` from the reference sources (``oracle/_ref/_torch_ext.so``).
Nothing of the reference is written into this repo: the fixtures are inputs + the outputs (and
gradients) the reference produced for them.

The known answers of the reference's own tests for this path are embedded as DATA and asserted
against the reference's outputs before anything is saved:
  * tests/mpops/torch_ops.py:27-72         (63 cases: 3 ops x 7 dtypes x dims 1..3)
  * tests/layers/conv/test_message_passing.py:12-24
  * tests/utils/test_degree.py:5-9, tests/utils/test_softmax.py:13-21, tests/utils/test_norm.py:7-27
  * docstring examples mpops/torch.py:64-71,120-127,229-235,260-266,291-297,322-330

Usage:  python tests/golden/make_golden.py   (needs /root/reference; run `make -C oracle ref` first)
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GGL_REFERENCE", "/root/reference")


def load_reference_mpops():
    so = os.path.join(REPO, "oracle", "_ref", "_torch_ext.so")
    assert os.path.exists(so), "run `make -C oracle ref` first"
    root = tempfile.mkdtemp(prefix="ggl_refpkg_")
    pkg = os.path.join(root, "refmpops")
    os.makedirs(os.path.join(pkg, "torch_ext"))
    open(os.path.join(pkg, "__init__.py"), "w").close()
    open(os.path.join(pkg, "torch_ext", "__init__.py"), "w").close()
    os.symlink(os.path.join(REF, "gammagl", "mpops", "torch.py"), os.path.join(pkg, "ops.py"))
    os.symlink(so, os.path.join(pkg, "torch_ext", "_torch_ext.so"))
    sys.path.insert(0, root)
    mod = importlib.import_module("refmpops.ops")
    assert mod.use_ext is True, "reference extension did not load"
    return mod


R = load_reference_mpops()
T = torch.tensor


def npy(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.detach().numpy()


# ------------------------------------------------------------------------------------------------
# 1. the reference's own known-answer tests
# ------------------------------------------------------------------------------------------------
def kat():
    out = {}
    dtypes = [torch.int8, torch.int16, torch.int32, torch.int64, torch.float16, torch.float32,
              torch.float64]
    gen = {1: [1, 2, 3], 2: [[1, 2], [3, 4], [5, 6]],
           3: [[[1, 2], [3, 4]], [[5, 6], [7, 8]], [[9, 10], [11, 12]]]}
    exp = {
        "max": {1: [2, 3], 2: [[3, 4], [5, 6]], 3: [[[5, 6], [7, 8]], [[9, 10], [11, 12]]]},
        "sum": {1: [3, 3], 2: [[4, 6], [5, 6]], 3: [[[6, 8], [10, 12]], [[9, 10], [11, 12]]]},
        "mean": {1: [1.5, 3], 2: [[2, 3], [5, 6]], 3: [[[3, 4], [5, 6]], [[9, 10], [11, 12]]]},
    }
    ops = {"max": R.segment_max, "sum": R.segment_sum, "mean": R.segment_mean}
    idx = T([0, 0, 1], dtype=torch.int64)
    n = 0
    for name, op in ops.items():
        for dt in dtypes:
            for dim in (1, 2, 3):
                x = T(gen[dim], dtype=dt)
                y = op(x, idx, 2)
                e = T(exp[name][dim], dtype=dt)  # same cast the reference test applies
                assert y.dtype == dt
                assert np.allclose(npy(y).astype(np.float64), npy(e).astype(np.float64), atol=1e-5), (name, dt, dim)
                key = f"{name}_{str(dt).split('.')[1]}_d{dim}"
                out[key + "_x"] = npy(x)
                out[key + "_y"] = npy(y)
                n += 1
    assert n == 63
    out["idx"] = idx.numpy()

    # test_message_passing.py: gather(x, src) then aggregate over dst
    x = torch.arange(0, 8, dtype=torch.float32).reshape(4, 2)
    ei = T([[0, 1, 2, 2, 3], [1, 2, 1, 0, 3]])
    msg = x[ei[0]]
    s, m, mx = (R.unsorted_segment_sum(msg, ei[1], 4), R.unsorted_segment_mean(msg, ei[1], 4),
                R.unsorted_segment_max(msg, ei[1], 4))
    assert s.tolist() == [[4.0, 5.0], [4.0, 6.0], [2.0, 3.0], [6.0, 7.0]]
    assert m.tolist() == [[4.0, 5.0], [2.0, 3.0], [2.0, 3.0], [6.0, 7.0]]
    assert mx.tolist() == [[4.0, 5.0], [4.0, 5.0], [2.0, 3.0], [6.0, 7.0]]
    out.update(mp_x=npy(x), mp_ei=npy(ei), mp_sum=npy(s), mp_mean=npy(m), mp_max=npy(mx))

    # test_degree.py: int64 ones, K=1
    row = T([0, 1, 0, 2, 0])
    deg = R.unsorted_segment_sum(torch.ones(5, dtype=torch.int64), row, 3)
    assert deg.dtype == torch.int64 and deg.tolist() == [3, 1, 1]
    out.update(deg_row=npy(row), deg_out=npy(deg))

    # test_softmax.py: segment_softmax (utils/softmax.py:29-35) composed from the reference ops
    xs = T([[1, 1], [1, 1], [2, 4], [2, 4]], dtype=torch.float32)
    x_e = xs[ei[0]]
    mxv = R.unsorted_segment_max(x_e, ei[1], 4)
    ex = torch.exp(x_e - mxv[ei[1]])
    den = R.unsorted_segment_sum(ex, ei[1], 4)
    score = ex / (den[ei[1]] + 1e-16)
    assert abs(float((score[2] - T([0.7311, 0.9526])).sum())) < 1e-4
    out.update(sm_x=npy(x_e), sm_score=npy(score))

    # test_norm.py: calc_gcn_norm (utils/norm.py:24-30)
    e2 = T([[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]])
    ew = torch.ones(6, 1)
    dg = R.unsorted_segment_sum(ew, e2[0], 4).reshape(-1)
    dis = dg.pow(-0.5)
    wts = dis[e2[0]] * ew.reshape(-1) * dis[e2[1]]
    d = np.array([1, 2, 2, 1]) ** -0.5
    assert np.allclose(wts.numpy(), [d[0] * d[1], d[1] * d[0], d[1] * d[2], d[2] * d[1], d[2] * d[3], d[3] * d[2]])
    out.update(norm_ei=npy(e2), norm_w=npy(wts))

    # docstring examples
    xd = T([[1., 2., 3., 4.], [4., 3., 2., 1.], [5., 6., 7., 8.]])
    ids = T([0, 2, 0])
    assert R.unsorted_segment_sum(xd, ids, 3).tolist() == [[6, 8, 10, 12], [0, 0, 0, 0], [4, 3, 2, 1]]
    assert R.unsorted_segment_mean(xd, ids, 3).tolist() == [[3, 4, 5, 6], [0, 0, 0, 0], [4, 3, 2, 1]]
    dm = R.unsorted_segment_max(xd, ids, 3)  # docstring claims zeros for the empty row: C++ gives lowest()
    assert dm[0].tolist() == [5, 6, 7, 8] and dm[2].tolist() == [4, 3, 2, 1]
    assert float(dm[1, 0]) == float(np.finfo(np.float32).min)
    gi = T([[0, 1, 1, 1, 2, 3, 3, 4], [1, 0, 2, 3, 1, 1, 4, 3]])
    gy = R.gspmm(gi, 2 * torch.ones(8), 2 * torch.ones(5, 8))
    assert gy[:, 0].tolist() == [4, 12, 4, 8, 4]
    out.update(doc_x=npy(xd), doc_ids=npy(ids), doc_max=npy(dm), doc_gi=npy(gi), doc_gspmm=npy(gy))
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **out)
    print("kat.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------
# 2. seeded random cases for the segment ops (forward + backward), all dtypes
# ------------------------------------------------------------------------------------------------
def rand_x(rng, shape, dt):
    if dt.is_floating_point:
        # small-integer-valued floats + a few exact halves: many ties, exact in f16/bf16
        v = rng.integers(-6, 7, size=shape).astype(np.float64) * 0.5
        return torch.tensor(v, dtype=dt)
    lo, hi = (0, 50) if dt == torch.uint8 else (-40, 41)
    return torch.tensor(rng.integers(lo, hi, size=shape), dtype=dt)


def segment_cases():
    rng = np.random.default_rng(20260926)
    out = {}
    dts = [torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64, torch.float16,
           torch.bfloat16, torch.float32, torch.float64]
    shapes = [  # (E, trailing shape, N)
        (37, (), 9), (64, (1,), 64), (200, (7,), 23), (129, (8,), 40), (300, (16,), 300),
        (90, (2, 3), 17), (513, (64,), 31), (70, (256,), 11), (50, (3, 2, 2), 60), (1, (5,), 3),
    ]
    ci = 0
    for (E, tail, N) in shapes:
        hi = min(N, E)  # keep ids < E: the reference's defined domain for mean
        ids = rng.integers(0, hi, size=E)
        if E > 8:
            ids[rng.integers(0, E, size=E // 4)] = ids[0]  # one heavy segment
        ids_t = torch.tensor(ids, dtype=torch.int64)
        for dt in dts:
            x = rand_x(rng, (E,) + tail, dt)
            key = f"c{ci}_{str(dt).split('.')[1]}"
            out[key + "_x"] = npy(x)
            out[key + "_sum"] = npy(R.unsorted_segment_sum(x, ids_t, N))
            out[key + "_mean"] = npy(R.unsorted_segment_mean(x, ids_t, N))
            out[key + "_max"] = npy(R.unsorted_segment_max(x, ids_t, N))
        out[f"c{ci}_ids"] = ids
        out[f"c{ci}_N"] = np.int64(N)
        ci += 1
    out["ncases"] = np.int64(ci)

    # generic float values (not tie-prone) in f32/f64 incl. backward
    bi = 0
    for (E, tail, N, full) in [(400, (16,), 50, True), (257, (3,), 257, False), (1000, (64,), 120, True),
                               (333, (2, 4), 333, False), (60, (256,), 7, True), (45, (), 45, False)]:
        # full=True: every segment non-empty (max backward well defined in the reference for N != E)
        ids = rng.integers(0, N, size=E)
        if full:
            ids[:N] = rng.permutation(N)
        ids_t = torch.tensor(ids, dtype=torch.int64)
        for dt in (torch.float32, torch.float64):
            key = f"b{bi}_{str(dt).split('.')[1]}"
            xv = torch.tensor(rng.standard_normal((E,) + tail), dtype=dt)
            # plant exact ties for max
            if E > 10:
                xv[5] = xv[2]
                ids_t[5] = ids_t[2]
            g = torch.tensor(rng.standard_normal((N,) + tail), dtype=dt)
            out[key + "_x"], out[key + "_g"] = npy(xv), npy(g)
            for name, op in (("sum", R.unsorted_segment_sum), ("mean", R.unsorted_segment_mean),
                             ("max", R.unsorted_segment_max)):
                xr = xv.clone().requires_grad_(True)
                y = op(xr, ids_t, N)
                y.backward(g)
                out[f"{key}_{name}"] = npy(y)
                out[f"{key}_{name}_gx"] = npy(xr.grad)
        out[f"b{bi}_ids"] = ids_t.numpy()
        out[f"b{bi}_N"] = np.int64(N)
        bi += 1
    out["nbwd"] = np.int64(bi)

    # NaN / inf / signed-zero behaviour of max (strict <: NaN never wins, -0.0 == +0.0 -> first wins)
    xs = torch.tensor([[np.nan, -0.0, np.inf, -np.inf], [1.0, 0.0, np.inf, -np.inf],
                       [np.nan, 0.0, 1.0, -np.inf], [2.0, np.nan, -np.inf, np.nan]], dtype=torch.float32)
    ids_t = T([1, 1, 0, 1])
    xr = xs.clone().requires_grad_(True)
    y = R.unsorted_segment_max(xr, ids_t, 4)  # N == E so the reference's sentinel is harmless
    y.backward(torch.arange(16, dtype=torch.float32).reshape(4, 4) + 1)
    out.update(sp_x=npy(xs), sp_ids=npy(ids_t), sp_max=npy(y), sp_gx=npy(xr.grad))
    ysum = R.unsorted_segment_sum(xs, ids_t, 4)
    out["sp_sum"] = npy(ysum)

    # f16 / bf16 accumulate-in-storage-dtype and count saturation (f16 count stops at 2048, bf16 at 256)
    E = 3000
    ids_t = torch.zeros(E, dtype=torch.int64)
    ids_t[-100:] = 1
    for dt, nm in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        x = torch.tensor(rng.integers(1, 4, size=(E, 2)).astype(np.float32) * 0.25, dtype=dt)
        out[f"sat_{nm}_x"] = npy(x)
        out[f"sat_{nm}_sum"] = npy(R.unsorted_segment_sum(x, ids_t, 3))
        out[f"sat_{nm}_mean"] = npy(R.unsorted_segment_mean(x, ids_t, 3))
    out["sat_ids"] = ids_t.numpy()
    np.savez_compressed(os.path.join(HERE, "segment.npz"), **out)
    print("segment.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------
# 3. gspmm / bspmm forward + backward (nothing in the reference's tests pins these)
# ------------------------------------------------------------------------------------------------
def spmm_cases():
    rng = np.random.default_rng(7)
    out = {}
    ci = 0
    for (N, E, K) in [(5, 8, 8), (40, 300, 1), (64, 400, 16), (33, 500, 7), (120, 2000, 64), (17, 90, 256),
                      (50, 0, 4)]:
        src = rng.integers(0, N, size=E)
        dst = rng.integers(0, max(N - 3, 1), size=E)  # last rows isolated
        if E > 20:
            src[10:14] = src[9]
            dst[10:14] = dst[9]  # duplicate (src,dst) pairs: spmm_max backward feeds all of them
        index = torch.tensor(np.stack([src, dst]), dtype=torch.int64)
        w = torch.tensor(rng.standard_normal(E), dtype=torch.float32)
        xv = torch.tensor(rng.standard_normal((N, K)), dtype=torch.float32)
        g = torch.tensor(rng.standard_normal((N, K)), dtype=torch.float32)
        key = f"s{ci}"
        out[key + "_index"], out[key + "_w"], out[key + "_x"], out[key + "_g"] = npy(index), npy(w), npy(xv), npy(g)
        for red in ("sum", "mean", "max"):
            xr = xv.clone().requires_grad_(True)
            y = R.gspmm(index, w, xr, red)
            y.backward(g)
            out[f"{key}_{red}"] = npy(y)
            out[f"{key}_{red}_gx"] = npy(xr.grad)
        ci += 1
    out["nspmm"] = np.int64(ci)
    bi = 0
    for (N, E, H, C) in [(30, 200, 8, 8), (50, 400, 4, 16), (12, 60, 1, 5), (25, 300, 8, 64)]:
        src = rng.integers(0, N, size=E)
        dst = rng.integers(0, N, size=E)
        index = torch.tensor(np.stack([src, dst]), dtype=torch.int64)
        w = torch.tensor(rng.standard_normal((E, H)), dtype=torch.float32, requires_grad=True)
        xv = torch.tensor(rng.standard_normal((N, H, C)), dtype=torch.float32, requires_grad=True)
        g = torch.tensor(rng.standard_normal((N, H, C)), dtype=torch.float32)
        y = R.bspmm(index, w, xv, "sum")
        y.backward(g)
        key = f"bs{bi}"
        out.update({key + "_index": npy(index), key + "_w": npy(w), key + "_x": npy(xv), key + "_g": npy(g),
                    key + "_y": npy(y), key + "_gx": npy(xv.grad), key + "_gw": npy(w.grad)})
        bi += 1
    out["nbspmm"] = np.int64(bi)
    np.savez_compressed(os.path.join(HERE, "spmm.npz"), **out)
    print("spmm.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------
# 4. layer-level fixtures: one GCNConv.forward-equivalent and one GATConv-equivalent computed with
#    the reference ops + torch for the dense parts
# ------------------------------------------------------------------------------------------------
def layer_cases():
    rng = np.random.default_rng(11)
    out = {}
    N, E, Fin, K = 64, 400, 12, 16
    src = rng.integers(0, N, size=E)
    dst = rng.integers(0, N, size=E)
    loops = np.arange(N)
    ei = torch.tensor(np.stack([np.concatenate([src, loops]), np.concatenate([dst, loops])]), dtype=torch.int64)
    x = torch.tensor(rng.standard_normal((N, Fin)), dtype=torch.float32)
    W = torch.tensor(rng.standard_normal((Fin, K)) * 0.3, dtype=torch.float32, requires_grad=True)
    b = torch.tensor(rng.standard_normal((1, K)) * 0.1, dtype=torch.float32)
    # gcn_conv.py:78-108 with norm='both'
    h = x @ W
    s, d = ei[0], ei[1]
    ones = torch.ones(ei.shape[1])
    deg_s = R.unsorted_segment_sum(ones, s, N)
    wts = deg_s.pow(-0.5)[s] * ones
    deg_d = R.unsorted_segment_sum(ones, d, N)
    wts = wts * deg_d.pow(-0.5)[d]
    msg = h[s] * wts.unsqueeze(-1)
    y = R.unsorted_segment_sum(msg, d, N) + b
    g = torch.tensor(rng.standard_normal((N, K)), dtype=torch.float32)
    y.backward(g)
    out.update(gcn_ei=npy(ei), gcn_x=npy(x), gcn_W=npy(W), gcn_b=npy(b), gcn_w=npy(wts), gcn_y=npy(y),
               gcn_g=npy(g), gcn_gW=npy(W.grad))
    # same through the reference gspmm (fused form): must agree with the unfused one
    y2 = R.gspmm(ei, wts.detach(), (x @ W).detach(), "sum") + b
    assert torch.allclose(y2, y.detach(), rtol=1e-5, atol=1e-6)

    # gat_conv.py:98-112 (dropout 0), heads=4, C=8
    H, C = 4, 8
    xg = torch.tensor(rng.standard_normal((N, H, C)), dtype=torch.float32, requires_grad=True)
    att = torch.tensor(rng.standard_normal((1, H, 2 * C)) * 0.5, dtype=torch.float32)
    feat = torch.cat((xg[s], xg[d]), dim=-1)
    e = (feat * att).sum(-1)
    e = torch.nn.functional.leaky_relu(e, 0.2)
    mx = R.unsorted_segment_max(e, d, N)
    ex = torch.exp(e - mx[d])
    den = R.unsorted_segment_sum(ex, d, N)
    alpha = ex / (den[d] + 1e-16)
    yg = R.unsorted_segment_sum(xg[s] * alpha.unsqueeze(-1), d, N)
    gg = torch.tensor(rng.standard_normal((N, H, C)), dtype=torch.float32)
    yg.backward(gg)
    el = (xg.detach() * att[:, :, :C]).sum(-1)
    er = (xg.detach() * att[:, :, C:]).sum(-1)
    out.update(gat_ei=npy(ei), gat_x=npy(xg), gat_att=npy(att), gat_el=npy(el), gat_er=npy(er),
               gat_alpha=npy(alpha), gat_y=npy(yg), gat_g=npy(gg), gat_gx=npy(xg.grad))
    # the head-averaging output layer (gat_conv.py:98-122 with concat=False: reduce_mean over heads :115-118, + bias) and the
    # two-layer GATModel (models/gat.py:36-72, eval mode) — composed by oracle/parity.py from the REFERENCE's segment ops
    # (round 6: the path FusedGATConv sends to ggl_gat_sh_* had no reference-made vector)
    sys.path.insert(0, REPO)
    from oracle import parity

    seg = (R.unsorted_segment_max, R.unsorted_segment_sum)
    H, Fin, C = 8, 16, 5
    xm = torch.tensor(rng.standard_normal((N, Fin)), dtype=torch.float32, requires_grad=True)
    Wm = torch.tensor(rng.standard_normal((Fin, H * C)) * 0.4, dtype=torch.float32, requires_grad=True)
    am = torch.tensor(rng.standard_normal((1, H, 2 * C)) * 0.5, dtype=torch.float32, requires_grad=True)
    bm = torch.tensor(rng.standard_normal((C,)) * 0.1, dtype=torch.float32, requires_grad=True)
    ym = parity.gat_conv_composed(xm, Wm, am, bm, ei, N, H, C, concat=False, slope=0.2, seg=seg)
    gm = torch.tensor(rng.standard_normal((N, C)), dtype=torch.float32)
    ym.backward(gm)
    out.update(gatm_x=npy(xm), gatm_W=npy(Wm), gatm_att=npy(am), gatm_b=npy(bm), gatm_y=npy(ym), gatm_g=npy(gm),
               gatm_gx=npy(xm.grad), gatm_gW=npy(Wm.grad), gatm_gatt=npy(am.grad), gatm_gb=npy(bm.grad))
    Fin2, Hd, NC = 12, 4, 7          # GATModel(12, 4, 7, heads=8, num_layers=2): 12 -> 8 x 4 (concat, ELU) -> mean of 8 x 7
    x2 = torch.tensor(rng.standard_normal((N, Fin2)), dtype=torch.float32)
    params = [(torch.tensor(rng.standard_normal((Fin2, H * Hd)) * 0.4, dtype=torch.float32, requires_grad=True),
               torch.tensor(rng.standard_normal((1, H, 2 * Hd)) * 0.5, dtype=torch.float32, requires_grad=True),
               torch.tensor(rng.standard_normal((H * Hd,)) * 0.1, dtype=torch.float32, requires_grad=True)),
              (torch.tensor(rng.standard_normal((H * Hd, H * NC)) * 0.3, dtype=torch.float32, requires_grad=True),
               torch.tensor(rng.standard_normal((1, H, 2 * NC)) * 0.5, dtype=torch.float32, requires_grad=True),
               torch.tensor(rng.standard_normal((NC,)) * 0.1, dtype=torch.float32, requires_grad=True))]
    y2m = parity.gat_model_composed(x2, params, ei, N, H, slope=0.2, seg=seg)
    g2m = torch.tensor(rng.standard_normal((N, NC)), dtype=torch.float32)
    y2m.backward(g2m)
    out.update(gatmodel_x=npy(x2), gatmodel_y=npy(y2m), gatmodel_g=npy(g2m))
    for li, (Wl, al, bl) in enumerate(params):
        out.update({f"gatmodel_W{li}": npy(Wl), f"gatmodel_att{li}": npy(al), f"gatmodel_b{li}": npy(bl),
                    f"gatmodel_gW{li}": npy(Wl.grad), f"gatmodel_gatt{li}": npy(al.grad), f"gatmodel_gb{li}": npy(bl.grad)})
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)
    print("layers.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------------------
# 5. neighbour sampler: the reference's own c_sample_adj (oracle/_ref/_sample.so, built from
#    ops/sparse/cpu/sample.cpp as it lies in the reference tree) on its DETERMINISTIC branches:
#      * num_neighbors < 0 (sample.cpp:39-55): no sampling;
#      * replace=False with num_neighbors >= every row's length (:77-80): every neighbour taken — the new
#        nodes are numbered in the iteration order of a std::unordered_set, so tests compare this one up
#        to that numbering (same seeds first, same (seed, neighbour, e_id) triples).
#    The sampled branches draw from srand(time(0)); rand() (sparse_utils.cpp:31-39) and cannot be pinned
#    by vectors: tests/ hold their invariants instead.
# ------------------------------------------------------------------------------------------------
def sampler_cases():
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    import _sample  # noqa: E402  (PYBIND11_MODULE(_sample) of the reference's sample.cpp)

    rng = np.random.default_rng(2024)
    out = {}
    ci = 0

    def csr(N, E, hub=None, dup=0):
        ei = rng.integers(0, N, size=(2, E)).astype(np.int64)
        if hub is not None:
            ei[1, : E // 5] = hub
        if dup:
            ei[:, :dup] = ei[:, dup:2 * dup]          # repeated edges (multi-graph rows)
        order = np.argsort(ei[1], kind="stable")
        rowptr = np.concatenate(([0], np.cumsum(np.bincount(ei[1], minlength=N)))).astype(np.int64)
        return rowptr, ei[0][order].copy()

    def emit(rowptr, col, seeds, fanout):
        nonlocal ci
        res = _sample.c_sample_adj(rowptr, col, seeds, int(fanout), False)
        k = f"c{ci}"
        out[k + "_rowptr"], out[k + "_col"], out[k + "_seeds"] = rowptr, col, seeds
        out[k + "_fanout"] = np.int64(fanout)
        for nm, a in zip(("orp", "ocol", "nid", "eid"), res):
            out[f"{k}_{nm}"] = np.asarray(a, dtype=np.int64)
        ci += 1

    # the tiny example worked by hand in tests (4 nodes)
    rp0 = np.array([0, 2, 5, 5, 6], np.int64)
    c0 = np.array([1, 3, 0, 2, 2, 1], np.int64)
    emit(rp0, c0, np.array([3, 0], np.int64), -1)
    emit(rp0, c0, np.array([1, 0, 2], np.int64), -1)
    emit(rp0, c0, np.array([1, 0], np.int64), 5)
    emit(rp0, c0, np.array([], np.int64), -1)
    for (N, E, hub, dup, B) in ((30, 200, 4, 0, 8), (300, 4000, 7, 0, 64), (300, 4000, 7, 150, 64),
                                (50, 60, None, 0, 50), (1000, 30000, 11, 500, 200)):
        rowptr, col = csr(N, E, hub, dup)
        perm = rng.permutation(N)
        seeds = perm[:B].astype(np.int64)
        if hub is not None and hub not in seeds:
            seeds[0] = hub
        emit(rowptr, col, seeds, -1)
        emit(rowptr, col, seeds, int(np.diff(rowptr).max()) + 3)   # take-all branch of the sampled code path
        dups = seeds.copy()
        dups[1::3] = dups[0]                                        # a seed listed several times (:24-29)
        emit(rowptr, col, dups, -1)
    out["ncases"] = np.int64(ci)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler.npz:", ci, "cases")


# ------------------------------------------------------------------------------------------------
# 6. COO <-> CSR pointers: the reference's c_ind2ptr / c_ptr2ind (oracle/_ref/_convert.so, built from
#    ops/sparse/cpu/convert.cpp), the host side of FusedGATConv's preprocessing (fusedgat_conv.py:103-117)
# ------------------------------------------------------------------------------------------------
def convert_cases():
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    import _convert  # noqa: E402

    rng = np.random.default_rng(77)
    out = {}
    ci = 0
    for (M, E, lo, hi) in ((1, 0, 0, 1), (5, 1, 2, 3), (5, 7, 0, 5), (40, 300, 0, 40), (40, 300, 5, 30),
                           (1000, 20000, 0, 1000), (7, 5000, 0, 7), (64, 64, 63, 64), (64, 64, 0, 1)):
        ind = np.sort(rng.integers(lo, hi, size=E)).astype(np.int64)
        for workers in (1, 0):
            ptr = np.asarray(_convert.c_ind2ptr(ind, M, workers), np.int64)
            back = np.asarray(_convert.c_ptr2ind(ptr, E, workers), np.int64)
            assert (back == ind).all()      # the reference's own round trip
            if workers == 1:
                out[f"v{ci}_ind"], out[f"v{ci}_M"], out[f"v{ci}_ptr"] = ind, np.int64(M), ptr
            else:
                assert (ptr == out[f"v{ci}_ptr"]).all()
        ci += 1
    out["ncases"] = np.int64(ci)
    np.savez_compressed(os.path.join(HERE, "convert.npz"), **out)
    print("convert.npz:", ci, "cases")


if __name__ == "__main__":
    torch.manual_seed(0)
    sampler_cases()
    convert_cases()
    kat()
    segment_cases()
    spmm_cases()
    layer_cases()
