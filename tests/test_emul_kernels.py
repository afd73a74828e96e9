"""Kernel-logic checks WITHOUT a GPU: the HIP kernel sources compiled for the host (tests/emul/,
one thread at a time; the kernels use no LDS / shuffles / atomics so the arithmetic and its order are
identical) behind the same Engine class the product uses, compared with the reference's golden
vectors and the oracle.  This is test infrastructure: the product never loads libggl_emul.so."""
import os
import subprocess

import pytest
import torch

import parity_cases as pc

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = torch.device("cpu")


@pytest.fixture(scope="module")
def eng():
    subprocess.check_call([os.path.join(HERE, "emul", "build.sh")])
    from gammagl_amd import _lib
    from gammagl_amd.ops import Engine

    return Engine(_lib.bind(os.path.join(HERE, "emul", "libggl_emul.so")), require_cuda=False)


def test_reference_known_answers(eng, golden):
    pc.check_kat(eng, DEV, golden)


def test_segment_all_dtypes_bit_exact(eng, golden):
    pc.check_segment_all_dtypes(eng, DEV, golden)


def test_segment_forward_backward_bit_exact(eng, golden):
    pc.check_segment_fwd_bwd(eng, DEV, golden)


def test_special_values_and_half_saturation(eng, golden):
    pc.check_special_values(eng, DEV, golden)


def test_gspmm_bspmm_golden(eng, golden):
    pc.check_spmm_golden(eng, DEV, golden)


def test_gcn_and_gat_layer_golden(eng, golden):
    pc.check_layers_golden(eng, DEV, golden)


def test_folded_2d_grids(eng, oracle, golden):
    """A dispatch holds at most 2^32 work-items per grid dimension (67 M rows at 4 rows per 256-thread
    block): wider launches are folded into (x, y).  Force the fold on tiny problems (max_grid_x = 3) and
    rerun the parity cases: every kernel must index through block_id() and ignore padding blocks."""
    eng.set_option("max_grid_x", 3)
    try:
        pc.check_kat(eng, DEV, golden)
        pc.check_random_vs_oracle(eng, DEV, oracle)
        pc.check_long_rows(eng, DEV, oracle)
        pc.check_gat_random(eng, DEV, oracle)
        pc.check_gat_dropout(eng, DEV, oracle)
        pc.check_spmm_bias_act(eng, DEV)
        pc.check_strided_accumulate(eng, DEV, oracle)
        pc.check_colsum(eng, DEV)
        pc.check_bias_act(eng, DEV)
        pc.check_convert(eng, DEV)
    finally:
        eng.set_option("max_grid_x", 1 << 22)


def test_bspmm_wide_heads(eng):
    pc.check_bspmm_wide(eng, DEV)


def test_attention_dropout_words_are_a_sound_random_source():
    pc.check_drop_word_statistics()


def test_scheduling_knobs_never_change_a_bit(eng):
    pc.check_schedule_invariance(eng, DEV)


def test_half_precision_ragged_rows(eng, oracle):
    pc.check_half_ragged_rows(eng, DEV, oracle)


def test_bspmm_weight_gradient_on_the_sorted_plan(eng, oracle):
    pc.check_bspmm_gradw_sorted(eng, DEV, oracle)


def test_strided_and_accumulating_forms(eng, oracle):
    pc.check_strided_accumulate(eng, DEV, oracle)


def test_spmm_with_fused_epilogue(eng):
    pc.check_spmm_bias_act(eng, DEV)


def test_random_vs_oracle(eng, oracle):
    pc.check_random_vs_oracle(eng, DEV, oracle)


def test_long_row_chunking(eng, oracle):
    pc.check_long_rows(eng, DEV, oracle)


def test_long_rows_in_the_reference_order(eng, oracle):
    pc.check_exact_long_rows(eng, DEV, oracle)


def test_host_build_walks_every_summing_row_in_one_piece(eng, oracle):
    """CPU tensors (the reference dispatches on x.is_cpu() too): no row of any summing mode is chunked, so f64 sums —
    which the GPU's serial hub kernel does not cover — and the backward walks are the reference's bits as well."""
    import numpy as np
    old = eng.chunk
    eng.chunk = 64
    eng.clear_caches()
    try:
        rng = np.random.default_rng(5)
        N, E = 40, 4000
        ids = rng.integers(0, N, size=E).astype(np.int64)
        ids[:2500] = 9
        rng.shuffle(ids)
        for K in (1, 5, 64):
            x = rng.standard_normal((E, K)) * 3
            xt, it = pc.to_t(x, DEV), pc.to_t(ids, DEV)
            assert eng.seg_plan(it, N).n_long >= 1
            pc.assert_same(pc.to_np(eng.c_segment_sum(xt, it, N)), oracle.segment_sum(x, ids, N), f"f64 host sum K{K}")
            pc.assert_same(pc.to_np(eng.c_segment_mean(xt, it, N)), oracle.segment_mean(x, ids, N), f"f64 host mean K{K}")
        # ... and the backward walks of gspmm mean / max over hub SOURCES (the GPU keeps its chunks there)
        index = np.stack([rng.integers(0, N, size=E), rng.integers(0, N, size=E)]).astype(np.int64)
        index[0, :1800] = 3                       # a hub source: a long row of the transposed plan
        index[1, 1800:3000] = 5                   # a hub destination
        index = np.ascontiguousarray(index[:, rng.permutation(E)])
        w = rng.standard_normal(E).astype(np.float32)
        for K in (4, 33):
            xs = rng.standard_normal((N, K)).astype(np.float32)
            go = rng.standard_normal((N, K)).astype(np.float32)
            it, wt = pc.to_t(index, DEV), pc.to_t(w, DEV)
            assert eng.graph_plan(it, N).bwd.n_long >= 1
            xt = pc.to_t(xs, DEV).requires_grad_(True)
            eng.c_spmm_mean(it, wt, xt).backward(pc.to_t(go, DEV))
            _, cnt = oracle.spmm_mean_fwd(index, w, xs)
            pc.assert_same(pc.to_np(xt.grad), oracle.spmm_mean_bwd(index, w, go, cnt), f"host mean backward K{K}")
            xt = pc.to_t(xs, DEV).requires_grad_(True)
            eng.c_spmm_max(it, wt, xt).backward(pc.to_t(go, DEV))
            _, arg = oracle.spmm_max_fwd(index, w, xs)
            pc.assert_same(pc.to_np(xt.grad), oracle.spmm_max_bwd(index, w, go, arg), f"host max backward K{K}")
    finally:
        eng.chunk = old
        eng.clear_caches()


def test_max_backward_forms(eng, oracle):
    pc.check_max_backward_forms(eng, DEV, oracle)


def test_gat_fused_random(eng, oracle):
    pc.check_gat_random(eng, DEV, oracle)


def test_gat_attention_dropout(eng, oracle):
    pc.check_gat_dropout(eng, DEV, oracle)


def test_edge_cases_and_errors(eng, oracle):
    pc.check_edge_cases(eng, DEV, oracle)


def test_plan_cache(eng):
    pc.check_plan_cache(eng, DEV)


def test_colsum_bias_gradient(eng):
    pc.check_colsum(eng, DEV)


def test_format_conversion_ind2ptr_ptr2ind_sort_edge_index(eng, golden):
    pc.check_convert(eng, DEV, golden)


def test_fused_bias_relu_dropout(eng):
    pc.check_bias_act(eng, DEV)


def test_neighbor_sampler(eng, oracle, golden):
    pc.check_sampler(eng, DEV, oracle)
    pc.check_sampler_golden(eng, DEV, golden)


def test_dropout_without_relu_gradient(eng):
    pc.check_dropout_without_relu_gradient(eng, DEV)


def test_epilogue_forms_sage_and_column_blocks(eng):
    pc.check_epilogue_forms(eng, DEV)


def test_weight_dtype_guard(eng):
    pc.check_weight_dtype_guard(eng, DEV)


def test_static_shape_block_sampler(eng, oracle):
    pc.check_block_sampler(eng, DEV, oracle)


def test_int_vector_lanes_and_padded_max_walk(eng, oracle):
    pc.check_round4_paths(eng, DEV, oracle)
