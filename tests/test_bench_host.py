"""bench.py's host logic that needs no GPU: the compact headline line and the per-aggregate counter accounting."""
import csv
import importlib.util
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("ggl_bench", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_counter_totals_are_per_aggregate_then_per_launch(tmp_path):
    """The row walk runs once per 64-column block (4 dispatches per K = 256 aggregate), the hub walk beside it ONCE per
    aggregate (round 5): a counter's value per launch = (4 row-walk dispatches + 1 hub dispatch) / 4 — not row walk + hub
    (which counted the hub's bytes four times: the first round-5 run reported 1.25 of the HBM peak that way).  Dispatches
    of the same kernels from BEFORE the probe's own launches (graph construction) are left out."""
    b = _bench()
    rows = []
    did = 0

    def add(kernel, value, counter="FETCH_SIZE"):
        nonlocal did
        did += 1
        rows.append({"Dispatch_Id": did, "Kernel_Name": kernel, "Counter_Name": counter, "Counter_Value": value})

    from gammagl_amd.benchmarks import CALIB

    for which in ("stream_read", "stream_copy", "gather256"):
        for _ in range(3):
            add(CALIB[which]["kernel"], 7.0)
    add("void ggl::row_reduce_kernel<float, 4, 0, 1, 1, true, 8, false>(...)", 999.0)     # an earlier, unrelated launch
    add("void ggl::hub_rows_f32_kernel<3, true, false, 4, 4, false>(ggl::HubF32Args)", 555.0)
    aggs, launches = 5, 4
    for _ in range(aggs):
        add("void ggl::hub_rows_f32_kernel<3, true, false, 4, 4, false>(ggl::HubF32Args)", 40.0)
        for _ in range(launches):
            add("void ggl::row_reduce_kernel<float, 4, 0, 1, 1, true, 8, false>(...)", 100.0)
    path = tmp_path / "p_counter_collection.csv"
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    out = f"pmc-probe: E=1 launches/aggregate=4 aggregates={aggs} dispatches={aggs * launches},{aggs}"
    vals, cal = b.counters_per_launch(str(path), out, b.KERNEL_OF["gcn"], ("FETCH_SIZE",))
    assert abs(vals["FETCH_SIZE"] - (4 * 100.0 + 40.0) / 4) < 1e-9
    assert [v for _, v in cal["FETCH_SIZE"]["gather256"]] == [7.0, 7.0, 7.0]
    # one hub launch per column-block launch (hub_one_launch = 0): 4 + 4 dispatches per aggregate
    rows2 = [r for r in rows if "hub_rows" not in r["Kernel_Name"]]
    for i in range(aggs * launches):
        rows2.append({"Dispatch_Id": 1000 + i, "Kernel_Name": "void ggl::hub_rows_f32_kernel<3, true, false, 4, 4, false>(x)",
                      "Counter_Name": "FETCH_SIZE", "Counter_Value": 10.0})
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows2)
    out = f"pmc-probe: aggregates={aggs} dispatches={aggs * launches},{aggs * launches}"
    vals, _ = b.counters_per_launch(str(path), out, b.KERNEL_OF["gcn"], ("FETCH_SIZE",))
    assert abs(vals["FETCH_SIZE"] - 110.0) < 1e-9
    # a kernel that never ran
    vals, _ = b.counters_per_launch(str(path), out, ("no_such_kernel",), ("FETCH_SIZE",))
    assert vals["FETCH_SIZE"] is None


def test_compact_line_keeps_the_contract_fields_under_the_limit():
    """The round-4 record (25 KB with four nested secondary lines: the driver could not parse it) compacts to < 4 KB with
    every field of the contract, the roofline / cpu_baseline / parity objects and one summary row per secondary config."""
    b = _bench()
    d = json.load(open(os.path.join(REPO, "profiles", "r4_bench_default.json")))
    line = b.compact_line(d)
    assert len(line) < b.LINE_LIMIT <= 4000 and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "secondary"):
        assert k in c, k
    assert abs(c["value"] - d["value"]) / d["value"] < 1e-5 and abs(c["ms_per_step"] - d["ms_per_step"]) < 1e-3
    rf = c["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] <= 1 and rf["traffic"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert c["cpu_baseline"]["kind"] == "reference" and c["cpu_baseline"]["cores"] == 1 and c["cpu_baseline"]["sample"]
    assert c["parity"]["ok"] is True and c["parity"]["rows_bit_exact_frac"] == 1.0
    assert [s["workload"] for s in c["secondary"]] == ["arxiv", "reddit-gat", "sage-minibatch", "papers-share"]
    assert all(s["parity_ok"] is True and s["value"] > 0 for s in c["secondary"])
    # prose that would push the line over the limit is cut, never the fields
    d["roofline"]["traffic_source"] = "x" * 5000
    d["config"]["workload"] = "products: " + "y" * 5000
    assert len(b.compact_line(d)) < b.LINE_LIMIT
