"""cProfile of the whole recursion (product path) on a synthetic input; prints the top entries.

    python tools/recursion_profile.py [--taxa 5000] [--trees 100] [--top 45]
"""
import argparse, cProfile, pstats, sys, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic
from spectralclustersupertree_amd.treearrays import TreeArrays

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=5000)
ap.add_argument("--trees", type=int, default=100)
ap.add_argument("--strategy", default="branch")
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--native-arrays", action="store_true")
args = ap.parse_args()
warnings.simplefilter("ignore")
if args.native_arrays:
    arrays = synthetic.tree_arrays(1, args.taxa, args.trees, None)
else:
    trees = synthetic.tree_objects(1, args.taxa, args.trees, None)
    names = sorted(scs._all_tip_names(trees))
    arrays = TreeArrays.from_trees(trees, [1.0] * len(trees), names)
scs.default_device()
from spectralclustersupertree_amd import kmeans2
kmeans2.fast_path_active()
pr = cProfile.Profile()
pr.enable()
scs._construct(arrays, args.strategy, True, np.random.RandomState(0))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(args.top)
