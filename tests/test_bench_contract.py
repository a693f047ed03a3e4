"""The one JSON line bench.py prints (the driver's contract): checked on the line committed under profiles/ (the bench itself needs a GPU)
and on bench.py's argument defaults."""
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


import pytest


@pytest.mark.parametrize("name", ["r02_bench.json", "r03_bench.json", "r04_bench.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    j = json.loads(open(os.path.join(ROOT, "profiles", name)).read())
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert "registered points/sec" in base["metric"] and "registered points/sec" in j["metric"] and j["unit"] == "points/s"
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and not any(k in j["config"] for k in ("model", "seq_len", "global_batch"))
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"]
    # whole-job value and the per-step time are one measurement
    n_raw = j["config"]["n_raw"]
    assert abs(j["value"] - n_raw / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"]


def test_round3_line_carries_the_parity_of_the_timed_path_and_every_config():
    """what VERDICT r02 asked of the driver-run line: the batched call's own poses against the oracle, configs 3 / 4 as legs, the three
    fractions of the dominant kernel side by side"""
    j = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench.json")).read())
    b = j["cpu_baseline"]["port"]["batch_vs_oracle_pose"]
    assert b["max_dpos_m"] < 1e-9 and b["max_drot_rad"] < 1e-9 and b["scans_checked"] >= 8
    assert b["all_timed_results_bit_identical_to_the_checked_ones"] is True and b["timed_results"] >= 32 * 3
    g = j["cpu_baseline"]["gpu_vs_reference_pose"]
    assert g["max_dpos_m"] < 1e-4 and g["max_drot_rad"] < 1e-5  # north_star's bar, against the reference's own code
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["traffic"] and 0 < r["frac_hbm_traffic"] < r["frac_touched"] < r["frac"] < 1.0
    assert abs(r["frac_touched"] - r["touched_bytes_per_launch"] / r["algorithmic_bytes_per_launch"] * r["frac"]) < 1e-3
    c = j["configs"]
    assert set(c) >= {"config2_1e6_map", "config3_stream_to_1e7_points", "config3_stream_lru_1e5_300_sweeps", "config4_localize_5e7_map"}
    assert c["config3_stream_to_1e7_points"]["map_points_end"] >= 10_000_000 and c["config3_stream_to_1e7_points"]["voxels_evicted"] == 0
    assert c["config3_stream_lru_1e5_300_sweeps"]["voxels_evicted"] > 0
    assert c["config4_localize_5e7_map"]["resident_map"]["target_points"] == 50_000_000
    assert c["config4_localize_5e7_map"]["resident_map"]["converged"] == 200
    for case in ("resident_map", "local_200k_map"):  # the throughput form of config 4: same poses as the per-scan run, bit for bit
        for nb in ("32_scans_per_call", "64_scans_per_call"):
            t = c["config4_localize_5e7_map"][case]["batched"][nb]
            assert t["max_abs_difference_from_the_single_scan_results"] == 0.0 and t["converged"] == 200
            assert t["ms_per_scan"] < 0.5 * c["config4_localize_5e7_map"][case]["ms_per_scan"] and 0 < t["roofline"]["frac"] < 1
    m = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_merge.json")).read())
    assert m["n_gpus"] == 1 and "NOT measured" in m["collective"]["backend"] and m["collective"]["states_identical_on_all_ranks"] is True


def test_round4_line_is_measured_on_varied_inputs_and_says_what_bounds_the_kernel():
    """what VERDICT r03 asked of the driver-run line: a pool of >= 128 scans spread over the map with round 3's pool beside it, the dominant kernel
    labelled by what bounds it (VALU issue) with the PMC instruction count, timed_region, configs 2 and 5 with their own roofline + baseline,
    the kNN figures on the map config 3 grows, the pooled scans against the reference's pinned build"""
    j = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench.json")).read())
    assert j["config"]["scan_pool"] >= 128 and j["config"]["spread_m"] >= 90.0
    assert j["rccl_ranks"] == j["n_gpus"] == 1 and j["collective"] is None
    r = j["roofline"]
    assert r["bound"] == "valu" and abs(r["timed_region"] - j["timed_seconds"]) < 1e-9 and r["timed_region"] > 3.0
    v = r["valu"]
    assert v["source"].startswith("PMC") and 0.3 < v["frac_of_valu_issue_peak"] < 1.0
    assert abs(v["frac_of_valu_issue_peak"] - v["issue_bound_us"] / r["avg_launch_us"]) < 2e-3
    assert r["traffic"] and r["frac_hbm_traffic"] < r["frac_touched"] < r["frac"]
    assert r["whole_scan"]["terms"]["B_ins"] == 0
    b = j["batch_vs_oracle_pose"]
    assert b["scans_checked"] == 128 and b["parity_ok"] and b["all_timed_results_bit_identical_to_the_checked_ones"] and b["max_dpos_m"] < 1e-9
    g = j["cpu_baseline"]["gpu_vs_reference_pose"]
    pinned = g["pinned_build"]["neighbour_lists_in_canonical_order"]
    assert pinned["median_dpos_m"] < 1e-12  # the same bits as the reference's own code, except ...
    left = g["pinned_build"]["what_is_left_in_canonical_order"]
    # ... where the reference breaks an exact f32 tie at the fifth-nearest boundary its own (implementation-defined) way: reported, with the tie
    for m in left["their_members"]:
        assert m["only_in_oracle_d2"] == m["only_in_reference_d2"]
    assert pinned["scans_beyond_1e_4_m_or_1e_5_rad"] == len({left["scan"]}) == 1
    c = j["configs"]
    p8 = c["pool8_one_spot"]
    assert p8["roofline"]["bound"] == "valu" and p8["ms_per_scan"] > j["ms_per_step"] * 0.8
    for k in ("config2_1e6_map", "config5_merge_8_submaps_1_gpu"):
        assert c[k]["roofline"]["bound"] == "valu" and c[k]["roofline"]["frac"] > 0 and c[k]["cpu_baseline"]["kind"] == "reference", k
        assert c[k]["cpu_baseline"]["value"] > 0 and c[k]["cpu_baseline"]["unit"] == "points/s"
    for k in ("config3_stream_to_1e7_points", "config3_stream_lru_1e5_300_sweeps"):
        kk = c[k]["knn_on_this_map"]
        assert kk["candidates_per_query"] > 5 and kk["touched_per_query"] <= kk["candidates_per_query"] and kk["registered"] >= 16, k
        assert c[k]["cpu_baseline"]["kind"] == "reference" and c[k]["cpu_baseline"]["ms_per_scan"] > 10 * c[k]["ms_per_scan"]
    assert c["config3_stream_lru_1e5_300_sweeps"]["cpu_baseline"]["sweeps_with_the_reference_map_at_capacity"] > 50
    c4 = c["config4_localize_5e7_map"]
    assert c4["scan_pool"] >= 32 and c4["local_200k_map"]["ms_per_scan"] < 0.3


def test_round5_line_is_the_compact_one_and_its_roofline_is_a_utilisation():
    """the committed line of the driver's command is the compact one (<= 6 KB, one line), carries every contract field, a
    roofline.frac that is a utilisation (PMC traffic over the live kernel time: never above 1) with the algorithmic credit figure and a MEASURED VALU
    fraction beside it, the reference as cpu_baseline with the count of scans beyond the north_star tolerance, the upload-included leg, and one short
    record per BASELINE configuration"""
    raw = open(os.path.join(ROOT, "profiles", "r05_bench.json")).read()
    assert len(raw) <= 6144 and raw.count("\n") <= 1
    j = json.loads(raw)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "configs", "upload_included", "batch_vs_oracle_pose"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["unit"] == "points/s" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert abs(j["value"] - j["config"]["n_raw"] / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"] and j["ms_per_step"] < 0.021
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] and r["frac_basis"].startswith("pmc") and 0.2 < r["frac"] < r["frac_touched"] < 1.0 < r["frac_algorithmic"]
    assert abs(r["frac"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 1e9 / 8000.0) < 2e-3
    assert 0.3 < r["frac_valu"] < 1.0 and 5e11 < r["valu_peak_wave_insts_per_s"] < 1.3e12  # measured in the same process (tools/valu_peak)
    assert abs(r["frac_valu"] - r["valu_wave_insts_per_launch"] / r["valu_peak_wave_insts_per_s"] / (r["avg_launch_us"] * 1e-6)) < 2e-3
    c = j["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 8 and c["unit"] == "points/s" and c["ms_per_scan"] > 1000 * j["ms_per_step"]
    assert c["gpu_vs_reference_pose"]["scans"] == 128 and c["gpu_vs_reference_pose"]["beyond_1e-4m_or_1e-5rad"] <= 1
    b = j["batch_vs_oracle_pose"]
    assert b["parity_ok"] and b["scans_checked"] == 128 and b["max_dpos_m"] < 1e-9
    u = j["upload_included"]
    assert u["parity_ok"] and j["ms_per_step"] < u["ms_per_step"] < 0.08 and 20 < u["pcie_GBps"] < 64
    legs = j["configs"]
    assert set(legs) == {"pool8_one_spot", "config2_1e6_map", "config3_stream_to_1e7_points", "config3_stream_lru_1e5_300_sweeps", "config4_localize_5e7_map",
                         "config5_merge_8_submaps_1_gpu", "sequence_batch"}
    for k, v in legs.items():
        assert "error" not in v and v["ms_per_scan"] > 0 and 0 < v["roofline"]["frac"] < 1.0, k
    assert legs["config2_1e6_map"]["cpu_baseline"]["gpu_vs_reference_pose"]["beyond_1e-4m_or_1e-5rad"] <= 1
    c4 = legs["config4_localize_5e7_map"]
    assert c4["resident_map"]["not_converged_checked"] >= 1 and c4["local_200k_map"]["converged"] == 200
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_full.json")).read())
    assert full["value"] == pytest.approx(j["value"], rel=1e-6) and len(json.dumps(full)) > 20000
    nc = full["configs"]["config4_localize_5e7_map"]["resident_map"]["not_converged"]
    # the alignments that end at max_iterations: the reference's own NDT_CUDA ends there too on most of them, at the same poses
    assert nc["checked"] - nc["reference_converged"] >= nc["checked"] - 2
    assert all(abs(a["ours"]["pos_err_m"] - a["reference"]["pos_err_m"]) < 5e-3 for a in nc["alignments"])


def test_compact_stdout_line_fits_the_drivers_parser():
    """round 4's 30 KB line was not parsed.  bench.py prints bench_line.line(record): at most bench_line.LIMIT (< 8 KB) bytes with the
    contract fields, roofline and cpu_baseline numbers, and one short record per secondary leg -- checked on the largest record there is (round 4's)
    and on a record with every string blown up"""
    import sys

    sys.path.insert(0, ROOT)
    import bench_line

    assert bench_line.LIMIT <= 6144
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench.json")).read())
    assert len(json.dumps(full)) > 25000
    s = bench_line.line(full)
    assert len(s) <= bench_line.LIMIT and "\n" not in s
    j = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "configs"):
        assert k in j, k
    assert j["value"] == pytest.approx(full["value"], rel=1e-6) and j["ms_per_step"] == full["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in j["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in j["cpu_baseline"], k
    assert j["cpu_baseline"]["gpu_vs_reference_pose"]["beyond_1e-4m_or_1e-5rad"] == 1
    assert set(j["configs"]) == set(full["configs"])
    for name, leg in j["configs"].items():
        assert leg["ms_per_scan"] == pytest.approx(full["configs"][name]["ms_per_scan"]), name
    # a pathological record (every note ten times as long, twenty more legs) still fits: optional material is dropped, the contract fields stay
    def blow(x):
        if isinstance(x, dict):
            return {k: blow(v) for k, v in x.items()}
        return x * 10 if isinstance(x, str) and len(x) > 40 else x
    big = blow(full)
    big["configs"].update({f"extra{i}": dict(big["configs"]["config2_1e6_map"]) for i in range(20)})
    s2 = bench_line.line(big)
    j2 = json.loads(s2)
    assert len(s2) <= bench_line.LIMIT and j2["roofline"]["frac"] == j["roofline"]["frac"] and j2["cpu_baseline"]["kind"] == "reference"
    # the other --config forms go through the same function
    for name in ("r04_bench_sequences.json", "r03_bench_merge.json", "r03_bench_localize.json", "r03_bench_stream_to_1e7.json"):
        r = json.loads(open(os.path.join(ROOT, "profiles", name)).read())
        c = json.loads(bench_line.line(r))
        assert c["value"] == pytest.approx(r["value"], rel=1e-6) and "roofline" in c and len(json.dumps(c)) <= bench_line.LIMIT, name
    # every rank-0 record leaves through emit() (benchlegs/common.py); the two print(json.dumps(out)) in the legs are the refparity / rcclprobe CHILD
    # processes' hand-over lines
    import glob

    legs = {os.path.basename(f): open(f).read() for f in glob.glob(os.path.join(ROOT, "benchlegs", "*.py"))}
    src = "".join(v for k, v in legs.items() if k != "common.py")
    assert len(re.findall(r"^\s+emit\(", src, re.M)) >= 5 and src.count("print(json.dumps(out))") == 2
    assert legs["common.py"].count("print(json.dumps(out))") == 1  # (--full-line: a child of the metric leg hands its whole record over)


def test_bench_is_an_entry_point_and_one_module_per_leg():
    """bench.py was one 2 400-line file running five sub-benchmarks.  It parses the arguments, brings up the ranks and dispatches;
    every BASELINE configuration has its own module under benchlegs/, none of them reads /root/reference, and only the checker legs import oracle/."""
    n = len(open(os.path.join(ROOT, "bench.py")).read().splitlines())
    assert n <= 200, n
    for leg in ("metric", "stream", "localize", "merge", "sequences", "parity", "rccl", "common"):
        src = open(os.path.join(ROOT, "benchlegs", leg + ".py")).read()
        assert len(src.splitlines()) <= 800, leg
        assert not re.search(r"[\"']/root/reference", src), leg  # (comments name where the checker libraries were compiled from; no code opens the tree)
    for leg in ("rccl", "common"):  # nothing of the checker in the plumbing
        assert "oracle" not in open(os.path.join(ROOT, "benchlegs", leg + ".py")).read(), leg


def test_bench_defaults_are_the_drivers_assumptions():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    assert re.search(r'"--steps", type=int, default=\d+', src) and re.search(r'"--warmup", type=int, default=\d+', src)
    assert "oracle" not in src.split("def main")[0]  # nothing of oracle/ is imported at module level: only the cpu_baseline leg loads it


def test_recorded_sweep_loader(tmp_path):
    """--config stream --bin-dir: KITTI-layout *.bin sweeps (x, y, z, intensity f32; NCLT velodyne_sync converted to it) + optional imu.csv"""
    import importlib.util

    import numpy as np

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(0)
    clouds = [rng.normal(size=(n, 4)).astype(np.float32) for n in (100, 57, 3)]
    for k, c in enumerate(clouds):
        c.tofile(tmp_path / f"{k:06d}.bin")
    sweeps, (t, g, a) = bench.load_bin_dir(str(tmp_path))
    assert len(sweeps) == 3
    for (p, st), c in zip(sweeps, clouds):
        assert np.array_equal(p, c) and st.dtype == np.uint32 and len(st) == len(c) and st[0] == 0 and st[-1] < 100000 and np.all(np.diff(st.astype(np.int64)) >= 0)
    assert np.allclose(a, [0, 0, 9.81]) and np.all(g == 0) and t[-1] >= 0.3 + 0.2  # no imu.csv: a level sensor at rest
    np.savetxt(tmp_path / "imu.csv", np.array([[0.0, 1, 2, 3, 4, 5, 6], [0.01, 1, 2, 3, 4, 5, 6]]), delimiter=",")
    _, (t, g, a) = bench.load_bin_dir(str(tmp_path))
    assert t.tolist() == [0.0, 0.01] and g[1].tolist() == [1, 2, 3] and a[0].tolist() == [4, 5, 6]


def test_round6_record_carries_the_parity_figures_the_verdict_asked_for():
    """profiles/r06_bench_full.json (the record behind the committed line): the tie at the fifth place closed on the headline, config 2 held against
    the pinned build with the reference's own build-to-build difference beside it, config 3 with per-sweep figures from the reference's state,
    covariance and map, config 4 with the reference's own per-scan envelope; the roofline object with a calibrated utilisation, the unique-byte floor
    and the credit figure under its own name"""
    j = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")).read())
    g = j["cpu_baseline"]["gpu_vs_reference_pose"]
    assert g["scans"] == 128 and g["scans_beyond_1e_4_m_or_1e_5_rad"] == 0 and g["max_dpos_m"] < 1e-4
    pin = g["pinned_build"]
    assert pin["neighbour_lists_in_canonical_order"]["max_dpos_m"] < 1e-12 and pin["neighbour_lists_in_canonical_order"]["scans_beyond_1e_4_m_or_1e_5_rad"] == 0
    own = pin["the_references_release_build_against_its_pinned_build"]
    assert own["scans"] >= 32 and abs(own["max_dpos_m"] - g["max_dpos_m"]) < 2e-5  # what is left against the release build is the reference's own build-to-build step
    r = j["roofline"]
    assert r["traffic"] and 0 < r["frac_unique"] < r["frac"] < r["frac_touched"] < 1 < r["frac_algorithmic"] and 0.3 < r["frac_valu"] < 1
    assert "frac" not in r["whole_scan"] and r["whole_scan"]["credit_over_peak"] > 0
    c = j["configs"]
    c2 = c["config2_1e6_map"]["cpu_baseline"]["gpu_vs_reference_pose"]
    assert c2["pinned_build"]["neighbour_lists_as_nth_element_leaves_them"]["scans_beyond_1e_4_m_or_1e_5_rad"] == 0
    assert c2["pinned_build"]["the_references_release_build_against_its_pinned_build"]["scans_beyond_1e_4_m_or_1e_5_rad"] >= c2["scans_beyond_1e_4_m_or_1e_5_rad"]
    for leg in ("config3_stream_to_1e7_points", "config3_stream_lru_1e5_300_sweeps"):
        ps = c[leg]["cpu_baseline"]["gpu_vs_reference_per_sweep"]
        p = ps["against_the_pinned_build"]
        assert p["teacher_forced_state_and_map"]["sweeps"] >= 140 and p["teacher_forced_state_and_map"]["sweeps_beyond_1e_4_m_or_1e_5_rad"] == 0
        assert p["teacher_forced_state_and_map_tie_mode_2"]["max_dpos_m"] < 1e-12
        assert ps["teacher_forced_state_and_map"]["sweeps_beyond_1e_4_m_or_1e_5_rad"] <= 1  # (the timed build: vectorised Eigen, tied stamps)
    e = c["config4_localize_5e7_map"]["cpu_baseline"]["gpu_vs_reference_pose"]["per_scan_envelope"]
    assert e["scans"] == 64 and e["inside_in_translation"] >= 60 and e["reference_envelope_m"]["median"] > e["hip_to_nearest_reference_run_m"]["median"]
    assert json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read())["ms_per_step"] == j["ms_per_step"]
    assert len(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read()) <= 6000
