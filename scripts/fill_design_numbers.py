"""Fills the @NAME@ placeholders of DESIGN.md / README.md from a bench line (the JSON bench.py printed).
usage: fill_design_numbers.py gpurun_out/<run>/bench.json [file ...]"""
import json, re, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v = d["variants"]
g = lambda k: v[k]
T = lambda x: "%.2f" % (x / 1000.0)            # TCUPS with two decimals
I = lambda x: "{:,.0f}".format(x).replace(",", " ")
vals = {
    "VALUE": I(d["value"]), "MS": "%.2f" % d["ms_per_step"], "SUS": I(d["sustained"]["gcups"]), "E2E": "%.2f" % d["e2e"]["ms"],
    "C3G": T(g("c3_with_paths_global_gcups")), "C3L": T(g("c3_with_paths_local_gcups")), "C3S": T(g("c3_with_paths_semiglobal_both_gcups")),
    "OHP": T(g("onehot_with_paths_gcups")), "FLP": T(g("float_profiles_with_paths_gcups")), "OHS": T(g("onehot_score_only_gcups")),
    "PPG": I(g("per_position_gaps_gcups")), "PPGP": I(g("per_position_gaps_with_paths_gcups")),
    "WIDE": I(g("wide_alphabet_gcups")), "WIDEP": I(g("wide_alphabet_with_paths_gcups")),
    "REF": I(g("reference_order_gcups")), "REFP": I(g("reference_order_with_paths_gcups")),
    "C4PLAN": "%.1f" % g("c4_rank_share_plan_ms"), "C4ALL": "%.0f" % g("c4_all_pairs_plan_ms"), "C3PLAN": "%.1f" % g("c3_path_plan_ms"),
    "C4F": I(g("c4_rank_share_float_gcups")), "C4ALLG": I(g("c4_all_pairs_one_gpu_gcups")), "C5ALL": I(g("c5_all_pairs_one_gpu_gcups")),
    "C5S": I(g("c5_shard_gcups")), "C4OG": I(g("c4_rank_share_onehot_global_gcups")), "C4OL": I(g("c4_rank_share_onehot_local_gcups")),
    "PPB": "%.0f" % g("c3_build_preprofiles_global_ms"), "PPBL": "%.0f" % g("c3_build_preprofiles_local_ms"),
    "C4E2E": "%.0f" % g("c4_rank_share_e2e_ms"),
}
for path in sys.argv[2:] or ["DESIGN.md"]:
    s = open(path).read()
    missing = set(re.findall(r"@([A-Z0-9]+)@", s)) - set(vals)
    assert not missing, missing
    for k, val in vals.items():
        s = s.replace("@%s@" % k, val)
    open(path, "w").write(s)
    print(path, "filled")
