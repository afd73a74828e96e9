"""Runs every shared parity case on the AddressSanitizer build of the host-emulated kernels
(launched by tests/test_emul_asan.py with LD_PRELOAD=libclang_rt.asan): out-of-bounds indexing or a
buffer that dies before its launch shows up here instead of as silent corruption on the GPU."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import parity_cases as pc  # noqa: E402
from gammagl_amd import _lib  # noqa: E402
from gammagl_amd.ops import Engine  # noqa: E402
from oracle import oracle as o  # noqa: E402

eng = Engine(_lib.bind(os.path.join(HERE, "libggl_emul_asan.so")), require_cuda=False)
dev = torch.device("cpu")
gd = os.path.join(REPO, "tests", "golden")
golden = {n: np.load(os.path.join(gd, n + ".npz")) for n in ("kat", "segment", "spmm", "layers")}
cases = [
    ("long_rows", lambda: pc.check_long_rows(eng, dev, o)), ("gat", lambda: pc.check_gat_random(eng, dev, o)),
    ("kat", lambda: pc.check_kat(eng, dev, golden)), ("dtypes", lambda: pc.check_segment_all_dtypes(eng, dev, golden)),
    ("fwd_bwd", lambda: pc.check_segment_fwd_bwd(eng, dev, golden)), ("special", lambda: pc.check_special_values(eng, dev, golden)),
    ("spmm", lambda: pc.check_spmm_golden(eng, dev, golden)), ("layers", lambda: pc.check_layers_golden(eng, dev, golden)),
    ("random", lambda: pc.check_random_vs_oracle(eng, dev, o, sizes=[(50, 400)])), ("edge", lambda: pc.check_edge_cases(eng, dev, o)),
    ("convert", lambda: pc.check_convert(eng, dev)), ("colsum", lambda: pc.check_colsum(eng, dev)), ("sampler", lambda: pc.check_sampler(eng, dev, o)), ("bias_act", lambda: pc.check_bias_act(eng, dev)), ("spmm_bias_act", lambda: pc.check_spmm_bias_act(eng, dev)), ("strided", lambda: pc.check_strided_accumulate(eng, dev, o)), ("gat_dropout", lambda: pc.check_gat_dropout(eng, dev, o)),
]
for name, fn in cases:
    fn()
    print(name, "ok", flush=True)
print("ASAN_CLEAN")
