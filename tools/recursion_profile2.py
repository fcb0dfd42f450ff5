"""cProfile of a whole construct_supertree recursion (run on the GPU box): where the host time
of the thousands of small nodes goes.  python tools/recursion_profile2.py [taxa] [trees]"""
import cProfile, pstats, sys, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd import scs, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200
arrays = synthetic.tree_arrays(1, n, m)
scs.default_device()
scs._construct(synthetic.tree_arrays(2, 300, 20), "branch", True, np.random.RandomState(0))  # warm-up
pr = cProfile.Profile()
pr.enable()
scs._construct(arrays, "branch", True, np.random.RandomState(0))
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(30)
print(out.getvalue())
