"""The one stdout line of bench.py, kept small.

bench.py gathers a large record (every leg's notes, samples, splits, drift curves: ~30 KB in round 4, which the driver's
parser did not take).  That record goes to bench_full*.json beside bench.py and to stderr; what goes to stdout is
compact(record): the contract fields, the numbers of `roofline` and `cpu_baseline`, and ONE short record per secondary leg --
at most LIMIT bytes, enforced here (optional keys are dropped in a fixed order until the line fits) and checked by
tests/test_bench_contract.py."""
import json

LIMIT = 6000  # bytes of the stdout line (the driver keeps an 8 KB tail)

_TOP = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "repeats", "timed_scans", "timed_seconds", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data")
_ROOF_NUM = ("bound", "limited_by", "achieved", "peak", "unit", "frac", "frac_basis", "traffic", "frac_unique", "unique_bytes_per_launch", "frac_algorithmic", "frac_touched", "frac_valu",
             "valu_peak_wave_insts_per_s", "valu_wave_insts_per_launch", "algorithmic_bytes_per_launch", "avg_launch_us",
             "candidates_per_query", "measured_copy_peak", "timed_region")
_CPU_NUM = ("value", "unit", "cores", "host_cpus", "kind", "ms_per_scan", "ms_per_sweep")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _num(v, sig=5):
    """floats to `sig` significant digits (a line of 22-digit doubles is mostly digits)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float(f"{v:.{sig}g}")


def _pick(d, keys, sig=5, text=110):
    return {k: (_short(d[k], text) if isinstance(d[k], str) else _num(d[k], sig)) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def roofline(r, text=90):
    if not isinstance(r, dict):
        return None
    o = _pick(r, _ROOF_NUM, 6, 40)
    if "kernel" in r:
        o["kernel"] = _short(r["kernel"], 48)
    ws = r.get("whole_scan")
    if isinstance(ws, dict):
        o["whole_scan"] = _pick(ws, ("algorithmic_bytes_per_scan", "credit_GBps", "credit_over_peak"))
    ok = r.get("other_kernels_us")
    if isinstance(ok, dict):
        o["other_kernels_us"] = {k: _num(v, 4) for k, v in ok.items() if not isinstance(v, (dict, list, str))}
    return o


def _own(g):
    r = g.get("reference_run_to_run") if isinstance(g, dict) else None
    return _num(r.get("max_dpos_m"), 3) if isinstance(r, dict) else None


def _beyond(g):
    """scans beyond the north_star tolerance (1e-4 m / 1e-5 rad) against the reference's build, wherever a leg recorded it"""
    if not isinstance(g, dict):
        return None
    for k in ("scans_beyond_1e_4_m_or_1e_5_rad", "scans_beyond_tolerance"):
        if k in g:
            o = {"scans": g.get("scans"), "beyond_1e-4m_or_1e-5rad": g[k], "max_dpos_m": _num(g.get("max_dpos_m"), 3), "max_drot_rad": _num(g.get("max_drot_rad"), 3)}
            if _own(g) is not None:
                o["reference_run_to_run_max_dpos_m"] = _own(g)
            pb = g.get("pinned_build")
            if isinstance(pb, dict):  # the build the path is pinned to: HIP in tie mode 2 against it as it is, the default mode against its lists in canonical order
                for src, dst in (("hip_in_tie_mode_2_against_the_pinned_build_as_it_is", "pinned_build_as_it_is_vs_tie_mode_2"), ("neighbour_lists_in_canonical_order", "pinned_build_canonical_lists")):
                    if isinstance(pb.get(src), dict):
                        o[dst] = {"scans": pb[src].get("scans"), "max_dpos_m": _num(pb[src].get("max_dpos_m"), 2)}
            pe = g.get("per_scan_envelope")
            if isinstance(pe, dict):
                o["inside_the_references_own_per_scan_envelope"] = "%s of %s" % (pe.get("hip_inside_the_references_own_envelope"), pe.get("scans"))
            return o
    return None


def cpu_baseline(c, text=90):
    if not isinstance(c, dict):
        return None
    o = _pick(c, _CPU_NUM)
    o["sample"] = _short(c.get("sample", ""), text)
    b = _beyond(c.get("gpu_vs_reference_pose"))
    if b:
        o["gpu_vs_reference_pose"] = b
    elif "gpu_vs_reference_pose_max_dpos_m" in c:
        o["gpu_vs_reference_pose"] = {"max_dpos_m": _num(c["gpu_vs_reference_pose_max_dpos_m"], 3)}
    for k in ("gpu_vs_oracle_pose_max_dpos_m", "pos_err_m_median"):
        if k in c:
            o[k] = _num(c[k], 3)
    p = c.get("port")
    if isinstance(p, dict):
        o["port"] = _pick(p, ("value", "cores", "ms_per_scan"))
    return o


def leg(c):
    """one short record per secondary leg: ms/scan, points/s, the roofline fraction, the same-run CPU baseline, the parity count"""
    if not isinstance(c, dict):
        return None
    if "error" in c and "ms_per_scan" not in c:
        return {"error": _short(c["error"], 160)}
    o = _pick(c, ("ms_per_scan", "points_per_s", "main_ms_median", "sessions", "sub_maps_per_gpu"))
    r = c.get("roofline")
    if isinstance(r, dict):
        o["roofline"] = _pick(r, ("frac", "avg_launch_us"), 4)
        basis = str(r.get("frac_basis", ""))
        o["roofline"]["basis"] = "pmc" if basis.startswith("pmc") else ("requested" if basis.startswith("bytes the kernel") else "algorithmic")
    b = c.get("cpu_baseline")
    if isinstance(b, dict):
        o["cpu_baseline"] = _pick(b, ("kind", "cores", "ms_per_scan", "ms_per_sweep"))
        g = _beyond(b.get("gpu_vs_reference_pose"))
        if g:
            o["cpu_baseline"]["gpu_vs_reference_pose"] = g
        gd = b.get("gpu_vs_reference_drive")
        if isinstance(gd, dict):  # config 3: a trajectory-level figure (two filters on the same drive)
            o["cpu_baseline"]["gpu_vs_reference_drive"] = {k: _num(v, 3) for k, v in gd.items() if k.startswith("dpos_m_after")}
    for name in ("resident_map", "local_200k_map"):  # config 4's two cases
        s = c.get(name)
        if isinstance(s, dict):
            o[name] = _pick(s, ("ms_per_scan", "converged"))
            nc = s.get("not_converged")
            if isinstance(nc, dict) and "checked" in nc:
                o[name]["not_converged_checked"] = nc["checked"]
                o[name]["reference_converged_on_them"] = nc.get("reference_converged")
    p = c.get("parity")
    if isinstance(p, dict):
        o["parity"] = _pick(p, ("sessions_checked", "bit_identical_to_the_per_session_engine", "max_abs_state_difference"), 3)
    co = c.get("collective")
    if isinstance(co, dict):
        o["collective"] = _pick(co, ("per_round_and_pass", "avg_us", "states_identical_on_all_ranks"))
    return o


def compact(out):
    o = {k: _num(out[k], 8) for k in _TOP if k in out}
    cfg = out.get("config") or {}
    o["config"] = {k: (_short(v, 230 if k == "workload" else 96) if isinstance(v, str) else _num(v)) for k, v in cfg.items() if not isinstance(v, dict)}
    o["roofline"] = roofline(out.get("roofline"))
    o["cpu_baseline"] = cpu_baseline(out.get("cpu_baseline"))
    b = out.get("batch_vs_oracle_pose")
    if isinstance(b, dict):
        o["batch_vs_oracle_pose"] = _pick(b, ("max_dpos_m", "max_drot_rad", "max_dstate", "scans_checked", "timed_results", "parity_ok",
                                              "all_timed_results_bit_identical_to_the_checked_ones"), 3)
    p = out.get("pose_error_vs_truth")
    if isinstance(p, dict):
        o["pose_error_vs_truth"] = _pick(p, ("max_dpos_m", "max_drot_rad", "median_m", "max_m"), 3)
    elif "pose_error_vs_truth_m" in out:
        o["pose_error_vs_truth_m"] = _num(out["pose_error_vs_truth_m"], 3)
    u = out.get("upload_included")
    if isinstance(u, dict):
        o["upload_included"] = _pick(u, ("ms_per_step", "value", "timed_scans", "host_bytes_per_scan", "pcie_GBps", "parity_ok"))
        if "error" in u:
            o["upload_included"]["error"] = _short(u["error"], 160)
    co = out.get("collective")
    if isinstance(co, dict):
        o["collective"] = _pick(co, ("rccl_ranks", "avg_us", "per_round_and_pass", "bytes", "iters", "states_identical_on_all_ranks"))
        if "error" in co:
            o["collective"]["error"] = _short(co["error"], 160)
        if "backend" in co:
            o["collective"]["backend"] = _short(co["backend"], 80)
    cs = out.get("configs")
    if isinstance(cs, dict):
        o["configs"] = {k: leg(v) for k, v in cs.items()}
    for k in ("resident_map", "local_200k_map", "parity", "latency"):  # the other --config forms carry these at top level
        if k not in o and isinstance(out.get(k), dict):
            o[k] = _pick(out[k], tuple(out[k].keys()), 4)
    if out.get("full_record"):
        o["full_record"] = out["full_record"]
    # the size guard: drop optional material in a fixed order until the line fits
    drops = [("configs", None, "roofline"), ("roofline", "other_kernels_us"), ("config", "engine"), ("configs", None, "cpu_baseline"), ("cpu_baseline", "sample"),
             ("configs",), ("config", "workload")]
    for d in drops:
        if len(json.dumps(o)) <= LIMIT:
            break
        if len(d) == 1:
            o.pop(d[0], None)
        elif len(d) == 2 and isinstance(o.get(d[0]), dict):
            if d == ("config", "workload"):
                o["config"]["workload"] = _short(o["config"].get("workload", ""), 120)
            else:
                o[d[0]].pop(d[1], None)
        elif len(d) == 3 and isinstance(o.get(d[0]), dict):
            for v in o[d[0]].values():
                if isinstance(v, dict):
                    v.pop(d[2], None)
    return o


def line(out):
    s = json.dumps(compact(out))
    assert len(s) <= LIMIT or "configs" not in json.loads(s), len(s)
    return s
