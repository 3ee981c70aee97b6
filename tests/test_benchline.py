"""bench.py's roofline object is assembled by a pure function (hobbyraytracer_amd/benchline.py): every branch of it runs here on
the CPU with synthetic numbers.  Round 2's bench line was lost to a '%'-formatted note in the branch only the headline workload
takes (traffic known and under half the algorithmic bytes), which no test reached (BENCH_r02.json: rc 1)."""
import json
import os
import subprocess
import sys

import pytest

from hobbyraytracer_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# round 2's headline figures (profiles/r02_v2): 4.069e9 algorithmic bytes per k_wf_ext launch, 0.543 ms, 0.32 GB of HBM traffic
R2 = dict(kname="k_wf_ext", k_bytes=4.069e9, k_ms=0.543, k_launches=50.0, frame_ms=51.4, frame_alg_bytes=2.05e11,
          box_per_ray=17.4, tri_per_ray=1.74, trav_box_tests=5.9e9, trav_tri_tests=4.1e8)
ISSUE = {"profile": "profiles/r02_v2", "lane_utilisation": 0.31, "wait_share": 0.52, "valu_busy": 0.75}


def test_headline_branch_traffic_and_issue_set():
    roof = benchline.roofline_block(**R2, traffic=3.19e8, issue=ISSUE)
    json.dumps(roof)
    assert roof["bound"] == "issue" and roof["bound_of_the_algorithmic_figure"] == "hbm"
    assert roof["frac"] == pytest.approx(4.069e9 / 0.543e-3 / 1e9 / 8000.0, rel=1e-4)
    assert roof["achieved"] == pytest.approx(roof["frac"] * roof["peak"], rel=1e-4)
    assert roof["traffic"] == 3.19e8
    assert roof["hbm_measured_frac"] == pytest.approx(3.19e8 / 0.543e-3 / 1e9 / 8000.0, rel=1e-3)
    b16 = (16 * 5.9e9 + 36 * 4.1e8) / 50.0
    assert roof["l2_frac"] == pytest.approx(b16 / 0.543e-3 / 1e9 / 17000.0, rel=1e-3)
    assert roof["frac_at_16B_per_box"] == pytest.approx(b16 / 0.543e-3 / 1e9 / 8000.0, rel=1e-3)
    assert roof["issue"] == ISSUE
    assert "8 % of the algorithmic bytes" in roof["note"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):      # the contract's keys
        assert k in roof


def test_traffic_close_to_the_algorithmic_bytes_stays_hbm_bound():
    roof = benchline.roofline_block(**R2, traffic=3.9e9, issue=ISSUE)
    assert roof["bound"] == "hbm" and "note" not in roof and roof["hbm_measured_frac"] > 0.8


def test_no_traffic_no_issue():
    roof = benchline.roofline_block(**R2)
    assert roof["traffic"] is None and roof["bound"] == "hbm"
    assert "issue" not in roof and "hbm_measured_frac" not in roof and "note" not in roof
    assert "l2_frac" in roof


def test_whole_frame_kernels_without_traversal_counts():
    roof = benchline.roofline_block("k_pathtrace", 2.0e11, 106.0, 1, 106.0, 2.0e11, 17.4, 1.7)
    assert "l2_frac" not in roof and roof["launches_per_frame"] == 1
    note = "no mesh in this scene: 100 % VALU bound, 0 % memory"      # a note with percent signs passes through untouched
    roof = benchline.roofline_block("k_wf_gen+k_wf_shade+k_wf_reduce (whole frame)", 1e9, 55.0, 1, 55.0, 1e9, 0.0, 0.0, traffic=1e6, note=note)
    assert roof["note"] == note and roof["bound"] == "hbm"


def test_bad_times_are_an_error_not_a_division_by_zero():
    with pytest.raises(ValueError):
        benchline.roofline_block(**{**R2, "k_ms": 0.0})
    with pytest.raises(ValueError):
        benchline.roofline_block(**{**R2, "k_launches": 0})


def test_committed_profiles_parse():
    """The newest committed profile is what the headline line quotes: it must parse, through the same code."""
    traffic, issue, prof = benchline.newest_profile_figures(ROOT, "k_wf_ext")
    assert traffic and traffic > 1e6 and prof.startswith("profiles/r")
    assert issue and 0 < issue["lane_utilisation"] < 1 and 0 < issue["wait_share"] < 1 and 0 < issue["valu_busy"] <= 1
    roof = benchline.roofline_block(**R2, traffic=traffic, issue=issue)
    json.dumps(roof)
    assert benchline.newest_profile_figures(ROOT, "no_such_kernel") == (None, None, None)


def test_issue_from_partial_counters():
    assert benchline.issue_from_pmc({}, "p") is None
    pk = {c: {"sum_over_one_frame": v} for c, v in (("SQ_THREAD_CYCLES_VALU", 640.0), ("SQ_ACTIVE_INST_VALU", 20.0), ("SQ_WAIT_ANY", 5.0), ("SQ_WAVE_CYCLES", 10.0))}
    iss = benchline.issue_from_pmc(pk, "p")
    assert iss == {"profile": "p", "lane_utilisation": 0.5, "wait_share": 0.5}
    pk.update({"TCC_HIT_sum": {"sum_over_one_frame": 9.0}, "TCC_MISS_sum": {"sum_over_one_frame": 1.0}, "GRBM_GUI_ACTIVE": {"sum_over_one_frame": 8.0}})
    iss = benchline.issue_from_pmc(pk, "p")
    assert iss["l2_hit_rate"] == 0.9 and iss["valu_busy"] == round(20.0 * 4 / 1024.0, 4)


def test_last_json_line():
    assert benchline.last_json_line("noise\n{\"a\": 1}\nW1005 torchrun chatter {not json}\n") == {"a": 1}
    assert benchline.last_json_line("nothing here\n{broken\n") is None


def test_bench_py_has_no_percent_formatting_of_notes():
    """The crash of round 2 was 'text with a bare % sign' % value.  bench.py and benchline.py build their strings with
    str.format / f-strings only."""
    import ast
    for f in ("bench.py", os.path.join("hobbyraytracer_amd", "benchline.py")):
        tree = ast.parse(open(os.path.join(ROOT, f)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mod):
                left = node.left
                assert not (isinstance(left, ast.Constant) and isinstance(left.value, str)) and not isinstance(left, ast.JoinedStr), \
                    f"{f}:{node.lineno}: '%' string formatting"


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks_and_relays_failure():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts torch.distributed.run as a child and exits with its
    code.  Without a GPU the ranks refuse ("needs a GPU"); what is checked here is the launch and the relay (the GPU suite checks
    the line, tests/test_gpu_cli.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""; env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--width", "32", "--height", "24", "--spp", "1",
                        "--steps", "1", "--warmup", "0", "--backend", "gloo", "--no-cpu-baseline", "--no-cli-wall-clock"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "bench.py needs a GPU" in p.stderr, p.stderr[-2000:]
    assert p.stdout.strip() == ""
