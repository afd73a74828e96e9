"""ctypes front-end for oracle/libggl_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package ``gammagl_amd``.  See oracle/ggl_oracle.h for
the reference file:line each function restates and how its parity is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libggl_oracle.so")

U8, I8, I16, I32, I64, F16, BF16, F32, F64 = range(9)

_NP2CODE = {
    np.dtype(np.uint8): U8, np.dtype(np.int8): I8, np.dtype(np.int16): I16,
    np.dtype(np.int32): I32, np.dtype(np.int64): I64, np.dtype(np.float16): F16,
    np.dtype(np.float32): F32, np.dtype(np.float64): F64,
}


def build():
    """Compile the C restatement (gcc, ~1 s)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def _lib():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    return lib


_L = None


def lib():
    global _L
    if _L is None:
        _L = _lib()
    return _L


class OracleIndexError(IndexError):
    pass


def _check(rc):
    if rc == -1:
        raise OracleIndexError("segment id / node id out of range")
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _code(x, bf16):
    if bf16:
        assert x.dtype == np.uint16, "bf16 payloads travel as uint16 bit patterns"
        return BF16
    return _NP2CODE[x.dtype]


def _ek(x):
    E = x.shape[0]
    K = int(np.prod(x.shape[1:], dtype=np.int64)) if x.ndim > 1 else 1
    return E, K


def _seg(fn, x, idx, N, bf16=False):
    x = np.ascontiguousarray(x)
    idx = _i64(idx)
    E, K = _ek(x)
    out = np.empty((N,) + x.shape[1:], dtype=x.dtype)
    _check(fn(_code(x, bf16), _p(x), _p(idx), ctypes.c_int64(E), ctypes.c_int64(K),
              ctypes.c_int64(N), _p(out)))
    return out


def segment_sum(x, idx, N, bf16=False):
    return _seg(lib().ggl_oracle_segment_sum, x, idx, N, bf16)


def segment_mean(x, idx, N, bf16=False):
    return _seg(lib().ggl_oracle_segment_mean, x, idx, N, bf16)


def segment_max(x, idx, N, bf16=False, arg_fill=None):
    x = np.ascontiguousarray(x)
    idx = _i64(idx)
    E, K = _ek(x)
    out = np.empty((N,) + x.shape[1:], dtype=x.dtype)
    arg = np.empty((N,) + x.shape[1:], dtype=np.int64)
    fill = E if arg_fill is None else arg_fill
    _check(lib().ggl_oracle_segment_max(_code(x, bf16), _p(x), _p(idx), ctypes.c_int64(E),
                                        ctypes.c_int64(K), ctypes.c_int64(N), _p(out), _p(arg),
                                        ctypes.c_int64(fill)))
    return out, arg


def _seg_bwd(fn, g, second, E, N):
    g = np.ascontiguousarray(g)
    second = _i64(second)
    K = int(np.prod(g.shape[1:], dtype=np.int64)) if g.ndim > 1 else 1
    gin = np.empty((E,) + g.shape[1:], dtype=g.dtype)
    _check(fn(_NP2CODE[g.dtype], _p(g), _p(second), ctypes.c_int64(E), ctypes.c_int64(K),
              ctypes.c_int64(N), _p(gin)))
    return gin


def segment_sum_bwd(gout, idx, N):
    return _seg_bwd(lib().ggl_oracle_segment_sum_bwd, gout, idx, len(idx), N)


def segment_mean_bwd(gout, idx, N):
    return _seg_bwd(lib().ggl_oracle_segment_mean_bwd, gout, idx, len(idx), N)


def segment_max_bwd(gout, arg, E):
    return _seg_bwd(lib().ggl_oracle_segment_max_bwd, gout, arg, E, gout.shape[0])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _nk(x):
    N = x.shape[0]
    K = int(np.prod(x.shape[1:], dtype=np.int64)) if x.ndim > 1 else 1
    return N, K


def spmm_sum_fwd(index, w, x):
    index, w, x = _i64(index), _f32(w), _f32(x)
    E = index.shape[1]
    N, K = _nk(x)
    out = np.empty_like(x)
    _check(lib().ggl_oracle_spmm_sum_fwd(_p(index), _p(w), _p(x), ctypes.c_int64(E),
                                         ctypes.c_int64(N), ctypes.c_int64(K), _p(out)))
    return out


def spmm_sum_bwd(index, w, g):
    index, w, g = _i64(index), _f32(w), _f32(g)
    E = index.shape[1]
    N, K = _nk(g)
    gx = np.empty_like(g)
    _check(lib().ggl_oracle_spmm_sum_bwd(_p(index), _p(w), _p(g), ctypes.c_int64(E),
                                         ctypes.c_int64(N), ctypes.c_int64(K), _p(gx)))
    return gx


def spmm_mean_fwd(index, w, x):
    index, w, x = _i64(index), _f32(w), _f32(x)
    E = index.shape[1]
    N, K = _nk(x)
    out = np.empty_like(x)
    cnt = np.empty((N,), dtype=np.int64)
    _check(lib().ggl_oracle_spmm_mean_fwd(_p(index), _p(w), _p(x), ctypes.c_int64(E),
                                          ctypes.c_int64(N), ctypes.c_int64(K), _p(out), _p(cnt)))
    return out, cnt


def spmm_mean_bwd(index, w, g, cnt):
    index, w, g, cnt = _i64(index), _f32(w), _f32(g), _i64(cnt)
    E = index.shape[1]
    N, K = _nk(g)
    gx = np.empty_like(g)
    _check(lib().ggl_oracle_spmm_mean_bwd(_p(index), _p(w), _p(g), _p(cnt), ctypes.c_int64(E),
                                          ctypes.c_int64(N), ctypes.c_int64(K), _p(gx)))
    return gx


def spmm_max_fwd(index, w, x):
    index, w, x = _i64(index), _f32(w), _f32(x)
    E = index.shape[1]
    N, K = _nk(x)
    out = np.empty_like(x)
    arg = np.empty(x.shape, dtype=np.int64)
    _check(lib().ggl_oracle_spmm_max_fwd(_p(index), _p(w), _p(x), ctypes.c_int64(E),
                                         ctypes.c_int64(N), ctypes.c_int64(K), _p(out), _p(arg)))
    return out, arg


def spmm_max_bwd(index, w, g, arg):
    index, w, g, arg = _i64(index), _f32(w), _f32(g), _i64(arg)
    E = index.shape[1]
    N, K = _nk(g)
    gx = np.empty_like(g)
    _check(lib().ggl_oracle_spmm_max_bwd(_p(index), _p(w), _p(g), _p(arg), ctypes.c_int64(E),
                                         ctypes.c_int64(N), ctypes.c_int64(K), _p(gx)))
    return gx


def bspmm_sum_fwd(index, w, x):
    index, w, x = _i64(index), _f32(w), _f32(x)
    E = index.shape[1]
    N, H, C = x.shape
    out = np.empty_like(x)
    _check(lib().ggl_oracle_bspmm_sum_fwd(_p(index), _p(w), _p(x), ctypes.c_int64(E),
                                          ctypes.c_int64(N), ctypes.c_int64(H), ctypes.c_int64(C),
                                          _p(out)))
    return out


def bspmm_sum_bwd(index, w, x, g):
    index, w, x, g = _i64(index), _f32(w), _f32(x), _f32(g)
    E = index.shape[1]
    N, H, C = x.shape
    gx = np.empty_like(x)
    gw = np.empty_like(w)
    _check(lib().ggl_oracle_bspmm_sum_bwd(_p(index), _p(w), _p(x), _p(g), ctypes.c_int64(E),
                                          ctypes.c_int64(N), ctypes.c_int64(H), ctypes.c_int64(C),
                                          _p(gx), _p(gw)))
    return gx, gw


def gat_fwd(index, el, er, x, slope=0.2, return_alpha=False):
    index, el, er, x = _i64(index), _f32(el), _f32(er), _f32(x)
    E = index.shape[1]
    N, H, C = x.shape
    out = np.empty_like(x)
    alpha = np.empty((E, H), dtype=np.float32)
    _check(lib().ggl_oracle_gat_fwd(_p(index), _p(el), _p(er), _p(x), ctypes.c_float(slope),
                                    ctypes.c_int64(E), ctypes.c_int64(N), ctypes.c_int64(H),
                                    ctypes.c_int64(C), _p(out), _p(alpha)))
    return (out, alpha) if return_alpha else out


def gat_bwd(index, el, er, x, g, slope=0.2):
    index, el, er, x, g = _i64(index), _f32(el), _f32(er), _f32(x), _f32(g)
    E = index.shape[1]
    N, H, C = x.shape
    gel = np.empty_like(el)
    ger = np.empty_like(er)
    gx = np.empty_like(x)
    _check(lib().ggl_oracle_gat_bwd(_p(index), _p(el), _p(er), _p(x), _p(g), ctypes.c_float(slope),
                                    ctypes.c_int64(E), ctypes.c_int64(N), ctypes.c_int64(H),
                                    ctypes.c_int64(C), _p(gel), _p(ger), _p(gx)))
    return gel, ger, gx


def f32_to_bf16_bits(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    fn = lib().ggl_oracle_f32_to_bf16
    fn.restype = ctypes.c_uint16
    fn.argtypes = [ctypes.c_float]
    return np.array([fn(float(v)) for v in a.ravel()], dtype=np.uint16).reshape(a.shape)


def bf16_bits_to_f32(a):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return (a.astype(np.uint32) << 16).view(np.float32)


# ---- the reference's own compiled CPU extension (oracle/_ref), when present -------------------
def load_ref_ext():
    """Import oracle/_ref/_torch_ext.so (the reference's CPU extension built by oracle/Makefile
    from the reference sources).  Returns None when it has not been built."""
    path = os.path.join(_HERE, "_ref", "_torch_ext.so")
    if not os.path.exists(path):
        return None
    import importlib.util

    import torch  # noqa: F401  (the extension links libtorch)

    spec = importlib.util.spec_from_file_location("_torch_ext", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- neighbour sampling: the deterministic branch of sample_adj, restated in Python ----------------
def sample_adj_full(rowptr, col, idx):
    """ops/sparse/cpu/sample.cpp:10-135 with num_neighbors < 0 (no sampling, :39-55): n_id = seeds then
    new nodes in first-seen order (:24-29,:48-51); every row's (local col, e_id) pairs sorted by local
    id (:112-118).  Returns (out_rowptr, out_col, out_n_id, out_e_id) as int64 arrays.
    Pinned: tests/test_oracle_golden.py holds it to tests/golden/sampler.npz, the outputs of the reference's own
    c_sample_adj (oracle/_ref/_sample.so, compiled from sample.cpp by oracle/Makefile)."""
    rowptr, col, idx = np.asarray(rowptr), np.asarray(col), np.asarray(idx)
    n_ids = [int(i) for i in idx]
    n_id_map = {}
    for n, i in enumerate(n_ids):
        n_id_map[i] = n          # a seed listed twice maps to its LAST position (operator[] overwrite, :27)
    out_rowptr = [0]
    cols = []
    for n in idx:
        row = []
        for e in range(int(rowptr[n]), int(rowptr[n + 1])):
            c = int(col[e])
            if c not in n_id_map:
                n_id_map[c] = len(n_ids)
                n_ids.append(c)
            row.append((n_id_map[c], e))
        row.sort(key=lambda t: t[0])
        cols.append(row)
        out_rowptr.append(out_rowptr[-1] + len(row))
    flat = [t for r in cols for t in r]
    return (np.array(out_rowptr, np.int64), np.array([t[0] for t in flat], np.int64),
            np.array(n_ids, np.int64), np.array([t[1] for t in flat], np.int64))


# ---- the multi-hop GPU sampler's deterministic branch (fan-out -1), restated in Python ---------------------------
def neighbor_sample_full(colptr, row, input_nodes, num_hops):
    """ops/sparse/cuda/neighbor_sample.cu:634-741 (cu_neighbor_sample) with every fan-out = -1 (no draw:
    get_eids_neighbor_sampler :120-123 takes rnd = row_offset): hop l lists ALL in-edges of the nodes hop l - 1 added
    (positions colptr[v] .. colptr[v + 1] - 1 of `row`, frontier node by frontier node, :131); the far ends not yet in
    the node list are appended in ascending id order (get_new_input_nodes :436-532); no new node ends the loop (:665-668).
    Returns (sample_cols, sample_rows, sample_nodes, sample_edges) as int64 arrays (:704-733): per edge the position in
    the node list of its owner (kernal_get_col :362-378) and of its far end (kernal_get_row :380-396).
    PARITY UNPINNED: the CUDA file cannot be built here (no nvcc) and the CPU counterpart (neighbor_sample.cpp) needs the
    `parallel_hashmap` submodule, which is empty in the reference checkout — this restatement follows the .cu line by
    line and is the only checker of that function's node / edge ORDER; its per-hop building block (sample_adj) is pinned
    against the reference's compiled c_sample_adj (tests/golden/sampler.npz)."""
    colptr, row = np.asarray(colptr, np.int64), np.asarray(row, np.int64)
    nodes = [int(v) for v in np.asarray(input_nodes).reshape(-1)]
    have = set(nodes)
    frontier, f_off = list(nodes), 0
    cols, edges = [], []
    for _ in range(int(num_hops)):
        if not frontier:
            break
        new = set()
        for k, v in enumerate(frontier):
            for e in range(int(colptr[v]), int(colptr[v + 1])):
                edges.append(e)
                cols.append(f_off + k)
                u = int(row[e])
                if u not in have:
                    new.add(u)
        if not new:
            break
        f_off = len(nodes)
        frontier = sorted(new)
        nodes.extend(frontier)
        have.update(frontier)
    pos = {}
    for i, v in enumerate(nodes):
        pos.setdefault(v, i)
    rows = [pos[int(row[e])] for e in edges]
    return (np.array(cols, np.int64), np.array(rows, np.int64), np.array(nodes, np.int64), np.array(edges, np.int64))
