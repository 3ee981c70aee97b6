"""Pure (no torch, no GPU) assembly of bench.py's ``roofline`` object, so that every branch of it is unit-tested
on the CPU (tests/test_benchline.py).  Round 2 lost its driver-measured line to a ``%``-formatted note that was only
built on the headline workload, which no test ran: nothing here uses ``%`` formatting, and the GPU suite now runs
bench.py at its default workload too (tests/test_gpu_cli.py).

Prices: 32 B per box tested, 36 B per triangle tested.  Peaks from
/opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; XCD-L2 gather rate 16.8-18.8 TB/s chip-wide
("Indexed rows: gather into LDS", rows shared by every workgroup) -- 17 TB/s is used as the yardstick for a
kernel whose working set is L2-resident.
"""
import glob
import json
import os

HBM_PEAK_GBPS = 8000.0
L2_GATHER_PEAK_GBPS = 17000.0
BOX_BYTES_ALGORITHMIC = 32.0
BOX_BYTES_STORED = 16.0        # two boxes per 32-byte culling node (hrt_pack.h)
TRI_BYTES = 36.0


def newest_profile_figures(root, kname):
    """(traffic_bytes_per_launch | None, issue dict | None, profile dir | None) from the newest committed
    profiles/r*/traffic.json + pmc_per_kernel.json that describe `kname` (rocprofv3 --pmc passes, profiles/collect.sh)."""
    traffic = None
    prof = None
    for tf in sorted(glob.glob(os.path.join(root, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            tj = json.load(open(tf))
            if tj.get("kernel") == kname:
                traffic = round(float(tj["hbm_bytes_per_launch"]), 1)
                prof = os.path.dirname(tf)
                break
        except Exception:
            continue
    issue = None
    if traffic is not None:
        cands = [os.path.join(prof, "pmc_per_kernel.json")] + sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_per_kernel.json")), reverse=True)
        for pf in cands:
            try:
                issue = issue_from_pmc(json.load(open(pf)).get(kname, {}), os.path.relpath(os.path.dirname(pf), root))
                if issue:
                    break
            except Exception:
                continue
    return traffic, issue, (os.path.relpath(prof, root) if prof else None)


def issue_from_pmc(pk, profile_name):
    """Issue-side figures of one kernel from its per-frame PMC sums:
    VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
    sums the 8 XCDs); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); wait share = SQ_WAIT_ANY /
    SQ_WAVE_CYCLES."""
    def g(c):
        return float(pk[c]["sum_over_one_frame"])
    need = ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    if not all(c in pk for c in need):
        return None
    issue = {"profile": profile_name,
             "lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4),
             "wait_share": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)}
    if "GRBM_GUI_ACTIVE" in pk and g("GRBM_GUI_ACTIVE") > 0:
        issue["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * g("GRBM_GUI_ACTIVE") / 8.0), 4)
    if "TCC_HIT_sum" in pk and "TCC_MISS_sum" in pk and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        issue["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    return issue


def roofline_block(kname, k_bytes, k_ms, k_launches, frame_ms, frame_alg_bytes, box_per_ray, tri_per_ray,
                   trav_box_tests=None, trav_tri_tests=None, traffic=None, issue=None, note=None):
    """The ``roofline`` object of the bench line.

    achieved / frac: SURVEY 8(d)'s ALGORITHMIC bytes per launch of the dominant kernel / its mean launch time, against the HBM
    peak (the graded definition).  ``bound`` says what really limits the kernel: when the counters show that HBM moved less than
    half of those bytes (the BVH is L2-resident) it is "issue", with hbm_measured_* (true HBM figures), l2_frac (the bytes the
    kernel really fetches per box and triangle against the XCD-L2 gather rate) and issue{} beside it.
    """
    if k_ms <= 0 or k_launches <= 0:
        raise ValueError("kernel time and launch count must be positive")
    sec = k_ms * 1e-3
    achieved = k_bytes / sec / 1e9
    roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
            "algorithmic_bytes_per_launch": round(k_bytes, 1), "kernel_ms_per_launch": round(k_ms, 5),
            "launches_per_frame": k_launches, "kernel_ms_per_frame": round(k_ms * k_launches, 4),
            "frame_pipeline_ms": round(frame_ms, 4), "frame_algorithmic_bytes": frame_alg_bytes,
            "frame_algorithmic_gbps": round(frame_alg_bytes / (frame_ms * 1e-3) / 1e9, 2) if frame_ms > 0 else None,
            "box_tests_per_ray": round(box_per_ray, 3), "tri_tests_per_ray": round(tri_per_ray, 3),
            "definition": "achieved = algorithmic bytes per launch (32 B per box tested + 36 B per triangle tested, SURVEY 8d) / mean launch time"}
    if trav_box_tests is not None and trav_tri_tests is not None:
        b16 = (BOX_BYTES_STORED * trav_box_tests + TRI_BYTES * trav_tri_tests) / k_launches
        g16 = b16 / sec / 1e9
        roof["achieved_at_16B_per_box"] = round(g16, 2)
        roof["frac_at_16B_per_box"] = round(g16 / HBM_PEAK_GBPS, 5)
        # the yardstick that still discriminates for an L2-resident tree: bytes really fetched / XCD-L2 gather rate
        roof["l2_frac"] = round(g16 / L2_GATHER_PEAK_GBPS, 5)
        roof["l2_peak"] = L2_GATHER_PEAK_GBPS
    if traffic:
        hbm = traffic / sec / 1e9
        roof["hbm_measured_gbps"] = round(hbm, 2)
        roof["hbm_measured_frac"] = round(hbm / HBM_PEAK_GBPS, 5)
    if issue:
        roof["issue"] = issue
    if note is None and traffic and traffic < 0.5 * k_bytes:
        roof["bound"] = "issue"      # not HBM: see hbm_measured_frac, l2_frac and issue{}
        roof["bound_of_the_algorithmic_figure"] = "hbm"
        pct = 100.0 * traffic / k_bytes
        note = ("the BVH is L2-resident: measured HBM traffic is {:.0f} % of the algorithmic bytes, so 'frac' prices L2-served bytes "
                "against the HBM peak and can exceed 1; the kernel is bound by instruction issue and by the latency of its divergent "
                "node fetches (served by L1 / L2), not by HBM (DESIGN.md 4.1); l2_frac is the figure that discriminates").format(pct)
    if note:
        roof["note"] = note
    return roof


def last_json_line(text):
    """The last line of `text` that parses as a JSON object, or None (a relay's view of a child's stdout)."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except Exception:
                continue
    return None
