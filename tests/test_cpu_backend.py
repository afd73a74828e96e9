"""The CPU side of the boundary ("CPU + HIP impls"; BASELINE config 1 is `--gpu -1`), as shipped:

* `gammagl_amd.mpops` on CPU tensors -> the CPU dispatch key -> libggl_mpops_host.so (the host build of the kernel
  sources), checked against the oracle and the reference's own known answers;
* ZERO-EDIT binding: a stand-in package whose `mpops/torch.py` holds exactly GammaGL's import statement
  (gammagl/mpops/torch.py:3-7) binds `gammagl_amd/compat/_torch_ext.py` dropped at `mpops/torch_ext/_torch_ext.py`
  (the stand-in is written here by the test: no reference file is copied);
* examples/gcn_trainer_amd.py --gpu -1 trains end to end on the CPU."""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

# the one statement through which GammaGL binds its native ops (gammagl/mpops/torch.py:3-7)
BINDING = '''use_ext = False
try:
    from .torch_ext._torch_ext import c_segment_sum, c_segment_mean, c_segment_max, c_spmm_sum, c_spmm_mean, c_spmm_max, c_bspmm_sum
    use_ext = True
except:
    pass
'''


def _stub_package(tmp_path):
    root = tmp_path / "ggl_standin"
    (root / "mpops" / "torch_ext").mkdir(parents=True)
    (root / "__init__.py").write_text("")
    (root / "mpops" / "__init__.py").write_text("")
    (root / "mpops" / "torch.py").write_text("import torch\n" + BINDING)          # note: torch_ext has no __init__.py
    shutil.copy(os.path.join(REPO, "gammagl_amd", "compat", "_torch_ext.py"), root / "mpops" / "torch_ext" / "_torch_ext.py")
    sys.path.insert(0, str(tmp_path))
    try:
        return importlib.import_module("ggl_standin.mpops.torch")
    finally:
        sys.path.pop(0)


def test_zero_edit_binding_and_cpu_dispatch(tmp_path, golden, oracle):
    m = _stub_package(tmp_path)
    assert m.use_ext is True
    g = golden["kat"]
    idx = torch.from_numpy(g["idx"].copy())
    for op, fn in (("sum", m.c_segment_sum), ("mean", m.c_segment_mean), ("max", m.c_segment_max)):
        for dt in ("float32", "int64", "float64"):
            x = torch.from_numpy(g[f"{op}_{dt}_d2_x"].copy())
            np.testing.assert_array_equal(fn(x, idx, 2).numpy(), g[f"{op}_{dt}_d2_y"], err_msg=f"{op} {dt}")
    # the SpMMs with autograd, against the oracle, bit for bit (short rows)
    gen = torch.Generator().manual_seed(5)
    N, E = 40, 500
    ei = torch.randint(0, N, (2, E), generator=gen)
    w = torch.rand(E, generator=gen)
    x = torch.randn(N, 12, generator=gen, requires_grad=True)
    go = torch.randn(N, 12, generator=gen)
    y = m.c_spmm_sum(ei, w, x)
    y.backward(go)
    np.testing.assert_array_equal(y.detach().numpy(), oracle.spmm_sum_fwd(ei.numpy(), w.numpy(), x.detach().numpy()))
    np.testing.assert_array_equal(x.grad.numpy(), oracle.spmm_sum_bwd(ei.numpy(), w.numpy(), go.numpy()))
    ym, _ = oracle.spmm_mean_fwd(ei.numpy(), w.numpy(), x.detach().numpy())
    np.testing.assert_array_equal(m.c_spmm_mean(ei, w, x.detach()).numpy(), ym)
    yx, _ = oracle.spmm_max_fwd(ei.numpy(), w.numpy(), x.detach().numpy())
    np.testing.assert_array_equal(m.c_spmm_max(ei, w, x.detach()).numpy(), yx)
    xh = torch.randn(N, 4, 8, generator=gen, requires_grad=True)
    wh = torch.rand(E, 4, generator=gen, requires_grad=True)
    gh = torch.randn(N, 4, 8, generator=gen)
    yb = m.c_bspmm_sum(ei, wh, xh)
    yb.backward(gh)
    ogx, ogw = oracle.bspmm_sum_bwd(ei.numpy(), wh.detach().numpy(), xh.detach().numpy(), gh.numpy())
    np.testing.assert_array_equal(yb.detach().numpy(), oracle.bspmm_sum_fwd(ei.numpy(), wh.detach().numpy(), xh.detach().numpy()))
    np.testing.assert_array_equal(xh.grad.numpy(), ogx)
    np.testing.assert_array_equal(wh.grad.numpy(), ogw)
    # error behaviour of the pybind functions: int32 index -> "expected scalar type Long"; size mismatch -> IndexError
    try:
        m.c_segment_sum(torch.ones(3, 2), torch.tensor([0, 1, 1], dtype=torch.int32), 2)
        raise AssertionError("int32 index accepted")
    except RuntimeError as ex:
        assert "Long" in str(ex)
    try:
        m.c_segment_sum(torch.ones(4, 2), torch.tensor([0, 1, 1]), 2)
        raise AssertionError("size mismatch accepted")
    except IndexError:
        pass


def test_mpops_surface_on_cpu_tensors(oracle):
    """gammagl_amd.mpops — the drop-in for gammagl/mpops/torch.py — on CPU tensors: routed to the host build, never to
    the HIP engine, results = the oracle's."""
    import gammagl_amd
    from gammagl_amd import mpops

    gen = torch.Generator().manual_seed(9)
    ids = torch.randint(0, 17, (200,), generator=gen)
    x = torch.randn(200, 7, generator=gen)
    np.testing.assert_array_equal(mpops.unsorted_segment_sum(x, ids, 17).numpy(), oracle.segment_sum(x.numpy(), ids.numpy(), 17))
    np.testing.assert_array_equal(mpops.unsorted_segment_mean(x, ids, 17).numpy(), oracle.segment_mean(x.numpy(), ids.numpy(), 17))
    np.testing.assert_array_equal(mpops.unsorted_segment_max(x, ids, 17).numpy(), oracle.segment_max(x.numpy(), ids.numpy(), 17)[0])
    assert mpops.segment_sum(x, ids).shape[0] == int(ids.max()) + 1          # num_segments=None: max(ids) + 1
    ei = torch.randint(0, 17, (2, 150), generator=gen)
    xn = torch.randn(17, 5, generator=gen)
    np.testing.assert_array_equal(mpops.gspmm(ei, None, xn).numpy(),
                                  oracle.spmm_sum_fwd(ei.numpy(), np.ones(150, np.float32), xn.numpy()))
    assert gammagl_amd.engine(x) is gammagl_amd.host_engine() and gammagl_amd.host_engine().cpu_only
    assert mpops.use_ext is True


def test_gcn_trainer_example_runs_on_the_cpu():
    """BASELINE config 1: the GCN trainer with --gpu -1 (TL_BACKEND=torch, CPU): loss decreases, accuracy beats chance."""
    env = {k: v for k, v in os.environ.items() if k != "GGL_BENCH_EMUL"}
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "gcn_trainer_amd.py"), "--gpu", "-1", "--n_epoch", "12",
                        "--hidden_dim", "16"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "loss" in ln.lower()]
    assert len(lines) >= 2, r.stdout[-1000:]


def test_gat_and_sage_trainer_examples_run_on_the_cpu():
    """The other two reference trainers with their own `--gpu -1`: fused and unfused GAT, and the neighbour-sampled
    GraphSAGE trainer with both samplers (the samplers pick the host build from the edge list's device)."""
    env = {k: v for k, v in os.environ.items() if k != "GGL_BENCH_EMUL"}
    runs = [["gat_trainer_amd.py", "--n_epoch", "2"], ["gat_trainer_amd.py", "--n_epoch", "1", "--unfused"],
            ["sage_trainer_amd.py", "--n_epoch", "1", "--nodes", "6000", "--batch_size", "512", "--hidden_dim", "16"],
            ["sage_trainer_amd.py", "--n_epoch", "1", "--nodes", "6000", "--batch_size", "512", "--hidden_dim", "16",
             "--sampler", "dynamic"]]
    for argv in runs:
        r = subprocess.run([sys.executable, os.path.join(REPO, "examples", argv[0]), "--gpu", "-1"] + argv[1:],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (argv, r.stderr[-2000:])
        assert "test acc" in r.stdout.lower(), (argv, r.stdout[-500:])


def test_zero_edit_binding_of_the_sparse_ops(tmp_path, golden):
    """`gammagl/ops/sparse/sparse.py:26-29` binds its GPU module in one import statement: a stand-in package holding
    exactly that statement binds `gammagl_amd/compat/_sparse_cuda.py` dropped at `ops/sparse/_sparse_cuda.py`; the
    conversions reproduce the reference's compiled c_ind2ptr / c_ptr2ind (golden/convert.npz), sample_adj the branches of
    c_sample_adj that draw nothing (golden/sampler.npz)."""
    root = tmp_path / "ggl_standin2"
    (root / "ops" / "sparse").mkdir(parents=True)
    for d in (root, root / "ops", root / "ops" / "sparse"):
        (d / "__init__.py").write_text("")
    (root / "ops" / "sparse" / "sparse.py").write_text(
        "from ._sparse_cuda import (cuda_torch_ind2ptr, cuda_torch_ptr2ind, cuda_torch_neighbor_sample, cuda_torch_sample_adj)\n")
    shutil.copy(os.path.join(REPO, "gammagl_amd", "compat", "_sparse_cuda.py"), root / "ops" / "sparse" / "_sparse_cuda.py")
    sys.path.insert(0, str(tmp_path))
    try:
        m = importlib.import_module("ggl_standin2.ops.sparse.sparse")
    finally:
        sys.path.pop(0)
    g = golden["convert"]
    for i in range(int(g["ncases"])):
        ind, M = torch.from_numpy(g[f"v{i}_ind"].copy()), int(g[f"v{i}_M"])
        ptr = m.cuda_torch_ind2ptr(ind, M)
        np.testing.assert_array_equal(ptr.numpy(), g[f"v{i}_ptr"], err_msg=f"case {i}")
        np.testing.assert_array_equal(m.cuda_torch_ptr2ind(ptr, ind.shape[0]).numpy(), np.sort(g[f"v{i}_ind"]), err_msg=f"case {i}")
    # one hop, against the blocks the reference's compiled c_sample_adj produced (the branches that draw nothing)
    import parity_cases as pc

    gs = golden["sampler"]
    for ci in range(int(gs["ncases"])):
        k = f"c{ci}"
        rowptr, col, seeds = (torch.from_numpy(gs[k + n].copy()) for n in ("_rowptr", "_col", "_seeds"))
        out = m.cuda_torch_sample_adj(rowptr, col, seeds, torch.tensor([int(gs[k + "_fanout"])]), False, False, 0)
        assert isinstance(out, list) and len(out) == 4
        pc.compare_block_with_reference([t.numpy() for t in out], gs, k, f"sampler case {ci}")
    # the multi-hop sampler (neighbor_sample.cu:744-778): node / edge order of the fan-out -1 branch vs its restatement,
    # the hop contract for positive fan-outs
    from oracle import oracle as orc

    pc.check_neighbor_sample(m.cuda_torch_neighbor_sample, torch.device("cpu"), orc)


def test_dgnn_dropin_for_the_fused_gat_layer(tmp_path, oracle):
    """gammagl_amd/compat/dgNN: `from dgNN.operators import GATConvFuse` (fusedgat_conv.py:70-71) binds the fused
    kernels with zero edits; called the way the layer calls it, CPU tensors -> the host build."""
    import parity_cases as pc

    pc.check_dgnn_dropin(torch.device("cpu"), oracle, tmp_path)
