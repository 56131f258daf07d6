"""cProfile of the guide tree + resident TreeMSA on 256 x ~400 aa: where the host time goes."""
import sys, os, time, cProfile, pstats, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "scripts", "exp_resident.py")).read().split("batch, serial =")[0])
batch = comp.BatchManager(idx)
for which in ("tree", "msa"):
    for rep in range(2):
        pr = cProfile.Profile()
        pr.enable()
        if which == "tree":
            tree = run(batch, comp.GuideTreeBuilder, sequences=seqs, track_id_sets=T, score_matrices=[blosum])['guide_tree']
        else:
            out = run(batch, comp.TreeMultipleSequenceAligner, sequences=seqs, guide_tree=tree, track_id_sets=T, score_matrices=[blosum])
        pr.disable()
    print("=====", which)
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
