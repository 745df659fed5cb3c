"""The bench line's contract (driver + judge read these keys), checked on the committed line of this round."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round4_bench_line_roofline_is_bounded_by_construction():
    """The round's own line (profiles/r04_bench_n1.json, the default command on one box).  Round-3 review: the headline `frac` was 1.10
    -- achieved algorithmic bytes over the HBM spec peak for an operator that sits in the Infinity Cache.  Now: `bound` says where the
    working set lives, the cache-resident object is priced against the streaming ceiling measured in the same run (`peak` =
    `peak_measured`, `frac` = `frac_of_measured` = time(ceiling) / time(kernel) <= 1), the HBM-resident objects keep the 8 TB/s spec
    peak for `frac` and carry `stored_frac` and `frac_of_measured` (both <= 1) next to it."""
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_n1.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "peak_measured", "frac_of_measured", "ceiling"):
        assert k in r, k
    assert r["bound"] == "infinity_cache" and r["cache_resident"] is True and r["working_set_bytes"] < 256 * 2 ** 20
    assert r["peak"] == r["peak_measured"] and r["frac"] == r["frac_of_measured"] and 0.5 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and abs(r["frac"] - r["ceiling"]["us_per_launch"] / r["us_per_launch"]) < 2e-3
    assert r["peak_hbm_spec"] == 8000.0 and abs(r["frac_of_hbm_spec_peak"] - r["achieved"] / 8000.0) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["us_per_launch"] * 1e-6) / 1e9) < 1.0
    assert r["traffic"] is not None and r["traffic_measured_in_run"] is False and r["traffic_kernel"] in r["kernel"]
    assert abs(r["ceiling"]["bytes_moved"] / r["ceiling"]["stored_bytes"] - 1) < 0.05      # the ceiling moves the bytes the form stores
    for key in ("spmv_hbm_resident", "spmv_irregular"):
        h = d[key]
        assert h["bound"] == "hbm" and h["cache_resident"] is False and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
        assert 0.2 < h["frac_of_measured"] <= 1.0 and 0.2 < h["stored_frac"] <= 1.0, (key, h["frac_of_measured"], h["stored_frac"])
        assert abs(h["achieved"] - h["algorithmic_bytes"] / (h["us_per_launch"] * 1e-6) / 1e9) < 1.0
    assert d["spmv_hbm_resident"]["check_max_rel_err"] <= 1e-13 and d["spmv_hbm_resident"]["traffic"] is not None
    i = d["spmv_irregular"]
    assert i["rows"] >= 1_000_000 and i["check_ok"] is True and i["row_lengths"]["coefficient_of_variation"] > 0.8 and i["row_lengths"]["max"] > 2048
    assert i["row_block_plan"]["long_rows"] >= 1 and 9.5 < i["stored_bytes_per_nnz"] < 12.5 and i["jacobi"]["us_per_sweep"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert abs(d["value"] - r["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    assert d["check"]["ok"] is True and d["check"]["max_rel_err"] <= 1e-13
    v = d["vcycle"]
    assert v["pcg_iterations"] == 9 and f"{v['final_residual']:.6e}" == "5.355578e-05" and f"{v['initial_residual']:.6e}" == "5.992963e+04"
    assert 0.5 < v["vcycle_frac_of_hbm_peak"] <= 1.0 and abs(v["vcycle_algorithmic_gbs"] - v["vcycle_algorithmic_bytes"] / (v["vcycle_ms"] * 1e-3) / 1e9) < 1.0
    # BASELINE configs[2]: the full V-cycle on Poisson 256^3, in the driver's own line since round 4
    w = d["vcycle_256"]
    pin = json.load(open(os.path.join(ROOT, "tests", "golden", "hierarchy_integers.json")))["poisson256"]
    assert w["levels"] == 10 and w["rows"] == pin["rows"] and w["nnz"] == pin["nnz"]
    assert w["pcg_iterations"] == 9 and f"{w['initial_residual']:.6e}" == "1.705086e+05" and f"{w['relative_residual']:.3e}" == "5.641e-09"
    assert 0.5 < w["vcycle_frac_of_hbm_peak"] <= 1.0 and w["vcycle_ms"] > 0


def test_committed_bench_line_has_the_contract_keys():
    """the round-2 line (kept: its fields are what the round-2 documents quote)"""
    line = open(os.path.join(ROOT, "profiles", "r02_bench_n1.json")).read().strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["us_per_launch"] * 1e-6) / 1e9) < 1.0
    assert r["traffic"] is None or 0.5 * r["algorithmic_bytes"] < r["traffic"] < 2 * r["algorithmic_bytes"]
    # the line states whether the timed working set is Infinity-Cache resident and carries the HBM-resident figure
    assert r["cache_resident"] is True and r["working_set_bytes"] < 256 * 2 ** 20 and r["traffic_measured_in_run"] is False
    h = d["spmv_hbm_resident"]
    assert h["cache_resident"] is False and h["working_set_bytes"] > 2 ** 30 and 0.5 < h["frac"] < 0.8
    assert abs(h["achieved"] - h["algorithmic_bytes"] / (h["us_per_launch"] * 1e-6) / 1e9) < 1.0 and h["check_max_rel_err"] <= 1e-13
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert abs(d["value"] - r["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"]
    assert d["check"]["ok"] is True and d["check"]["max_rel_err"] <= 1e-13
    v = d["vcycle"]
    assert v["pcg_iterations"] == 9 and f"{v['final_residual']:.6e}" == "5.355578e-05" and f"{v['initial_residual']:.6e}" == "5.992963e+04"


def test_committed_kernel_trace_agrees_with_the_bench_line():
    """rocprofv3 --kernel-trace of the same command (profiles/r02_bench_n1_kernel_stats_by_grid.csv, split by operator):
    the average duration of the bench kernel on the 128^3 operator agrees with the line's HIP-event figure, in the
    profiled run itself and in the committed unprofiled line"""
    import csv
    for rnd in ("r02", "r03", "r04"):
        rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_n1_kernel_stats_by_grid.csv"))))
        for name in (f"{rnd}_bench_n1_under_rocprof.json", f"{rnd}_bench_n1.json"):
            d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
            kernel = d["roofline"]["kernel"].split(",")[0].split("<")[0]           # "k_sell" | "k_csr_cc16" | "k_sellp" (round 3)
            rows_per_wg = 256                                                      # k_sell / k_sellp: 4 slices of 64 rows per workgroup
            cand = [r for r in rows if f"sk::{kernel}<0," in r["Name"] and int(r["Calls"]) >= 100]
            if kernel in ("k_sell", "k_sellp"):
                cand = [r for r in cand if int(r["Workgroups"]) == (2000376 + rows_per_wg - 1) // rows_per_wg]
            assert cand, (kernel, [r["Name"] for r in rows[:5]])
            avg_us = float(max(cand, key=lambda r: int(r["Calls"]))["AverageNs"]) / 1e3
            assert abs(avg_us - d["roofline"]["us_per_launch"]) <= 0.05 * avg_us, (name, avg_us, d["roofline"]["us_per_launch"])
        if rnd in ("r03", "r04"):                                                  # ... and the HBM-resident 256^3 figure with its kernel
            h = d["spmv_hbm_resident"]
            hk = h["kernel"].split(",")[0]
            cand = [r for r in rows if f"sk::{hk}<0," in r["Name"] and int(r["Workgroups"]) > 30000]
            assert cand, hk
            avg_us = float(max(cand, key=lambda r: int(r["Calls"]))["AverageNs"]) / 1e3
            assert abs(avg_us - h["us_per_launch"]) <= 0.05 * avg_us, (avg_us, h["us_per_launch"])
