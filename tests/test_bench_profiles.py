"""bench.py quotes HBM traffic and the VALU issue rate from committed PMC passes -- only from a profile of THIS build, workload,
batch size and options (quote_profile).  Round 2's version picked "the newest" profile by lexicographic file name (r02_v10 sorted
before r02_v9) and matched the build id only; this pins the selection rules, and that the profiles committed with the tree
are the ones the shipped library is quoted from."""
import importlib.util
import json
import os

from tests.util import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write(d, tag, build, workload="ncov_r9_dna_q250", reads=100000, opts=(), fetch=100.0, write=200.0, valu=4.0e10, grbm=8.0e8, ms=50.0):
    json.dump({"build_id": build, "workload": workload, "reads": reads, "opts": list(opts)}, open(os.path.join(d, tag + "_meta.json"), "w"))
    with open(os.path.join(d, tag + "_pmc_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean_value_per_dispatch,mean_duration_ms_in_that_pass\n")
        k = '"void sfa::sdtw_fill_kernel<16, false, false, false, true, true>"'
        for c, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write), ("SQ_INSTS_VALU", valu), ("GRBM_GUI_ACTIVE", grbm)):
            f.write(f"{k},{c},3,{v:.1f},{ms:.3f}\n")
        f.write(f'"void sfa::sdtw_finalize_kernel",WRITE_SIZE,3,999999.0,0.010\n')  # other kernels are not the dominant one


def test_only_a_profile_of_this_build_workload_size_and_options_is_quoted(tmp_path):
    B = _bench()
    d = str(tmp_path)
    assert B.quote_profile(d, "ncov_r9_dna_q250", 100000, [], "aaa")[0] is None and "not profiled" in B.quote_profile(d, "ncov_r9_dna_q250", 100000, [], "aaa")[1]
    _write(d, "r03_v9", "old", fetch=1.0)
    _write(d, "r03_v10", "aaa", fetch=100.0, write=200.0)           # natural order: v10 is newer than v9
    _write(d, "r03_other", "aaa", workload="sequin_r9_rna_q250")
    _write(d, "r03_small", "aaa", reads=3000)
    _write(d, "r03_opt", "aaa", opts=["lds_ckpt=0"])
    t, src, issue = B.quote_profile(d, "ncov_r9_dna_q250", 100000, [], "aaa")
    assert t == round((2 * 100.0 + 200.0) * 1024) and "r03_v10" in src
    assert abs(issue["issue_rate"] - 4.0e10 / 1024 / (8.0e8 / 8)) < 1e-4 and abs(issue["sustained_clock_ghz"] - (8.0e8 / 8) / 50e-3 / 1e9) < 1e-3
    assert abs(issue["attainable_frac"] - issue["issue_rate"] / 0.455) < 1e-3
    # another build of the library: stale, nothing quoted -- and it names the newest matching profile, not the lexicographic one
    t, src, issue = B.quote_profile(d, "ncov_r9_dna_q250", 100000, [], "bbb")
    assert t is None and issue is None and src.startswith("stale: r03_v10_meta.json")
    # options and batch size are part of the identity
    assert "r03_opt" in B.quote_profile(d, "ncov_r9_dna_q250", 100000, ["lds_ckpt=0"], "aaa")[1]
    assert "r03_small" in B.quote_profile(d, "ncov_r9_dna_q250", 3000, [], "aaa")[1]
    assert B.quote_profile(d, "ncov_r9_dna_q250", 4000, [], "aaa")[0] is None


def test_every_baseline_configuration_has_a_committed_profile():
    """profiles/<newest round>_*: kernel statistics + PMC summary + meta for the workload shapes DESIGN.md section 4 tabulates, all
    taken on one build in one call (tools/profile_all.sh)."""
    import re
    B = _bench()
    want = {"ncov_r9_dna_q250": 100000, "sequin_r9_rna_q250": 100000, "rna004_fullref_dtwstd_q250": 100000, "r10_dna_1mb_q250": 125000,
            "ncov_r9_dna_q1000": 25000}
    names = [f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_.*_meta\.json$", f)]
    newest = max(int(re.match(r"r(\d+)_", f).group(1)) for f in names)
    metas = {}
    for f in names:
        if f.startswith("r%02d_" % newest):
            m = json.load(open(os.path.join(ROOT, "profiles", f)))
            metas[m["workload"]] = (f[:-len("_meta.json")], m)
    assert newest >= 4 and set(want) <= set(metas), (newest, set(want) - set(metas))
    builds = {m["build_id"] for _, m in metas.values()}
    assert len(builds) == 1  # one call, one build
    for wl, reads in want.items():
        tag, m = metas[wl]
        assert m["reads"] == reads and m["opts"] == []
        for suffix in ("_kernel_stats.csv", "_pmc_summary.csv"):
            assert os.path.getsize(os.path.join(ROOT, "profiles", tag + suffix)) > 100
        t, src, issue = B.quote_profile(os.path.join(ROOT, "profiles"), wl, reads, [], m["build_id"])
        assert t and t < 20e9 and issue and 0.40 < issue["issue_rate"] < 0.455 and 1.9 < issue["sustained_clock_ghz"] < 2.5, (wl, t, issue)
