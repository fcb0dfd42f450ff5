"""Debug helper (GPU box): one fuzz_recursion case, the node where the labels differ -- the
product's embedding, scikit-learn's, and what the public k_means / kmeans2 make of each."""
import sys, warnings
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
warnings.simplefilter("ignore")
from test_gpu_recursion import recursion_input, trace_nodes, construct_supertree, TreeArrays, so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import kmeans2
from sklearn.cluster import k_means
np.set_printoptions(linewidth=200, precision=17)
case = eval(sys.argv[1])
trees, weights = recursion_input(case['seed'] % 100000, case['taxa'], case['trees'], case['leaves'], case['twins'], case['weighted'])
given = trees
if case['arrays']:
    names = sorted(so._all_tips(trees))
    given = TreeArrays.from_trees(trees, weights or [1.0] * len(trees), names)
rs = np.random.RandomState(case['seed'] % 9973)
with trace_nodes() as trace:
    construct_supertree(given, None if case['arrays'] else weights, case['strategy'], contract_edges=case['contract'], random_state=rs)
trace = list(trace)
otrace = []
def steer(entry, labels):
    k = len(otrace) - 1
    mine = trace[k]
    if np.array_equal(mine["labels"], labels):
        return labels
    matrix = entry["matrix"]
    st = np.random.RandomState(); st.set_state(entry["rng_state"])
    pts = so.spectral_maps(matrix, st)
    km = st.get_state()
    def pub(p):
        r = np.random.RandomState(); r.set_state(km); return k_means(p, 2, random_state=r, n_init=10)[1]
    def mineq(p):
        r = np.random.RandomState(); r.set_state(km); return kmeans2.labels(p, r)
    lam = np.sort(np.linalg.eigvalsh(to.normalized_operator(matrix)[0]))[::-1]
    print("call", k, "V", len(labels), "oracle labels", labels, "product labels", mine["labels"])
    print(" lambda", lam[:4])
    print(" sklearn maps col1", pts[:, 1]); print(" product maps col1", mine["maps"][:, 1])
    print(" col0 diff", np.abs(pts[:, 0] - mine["maps"][:, 0]).max())
    np.save(str(ROOT / "gpurun_out" / f"tie_case_{case['seed']}_{k}_maps.npy"), mine["maps"])
    np.save(str(ROOT / "gpurun_out" / f"tie_case_{case['seed']}_{k}_state.npy"), np.array(km, dtype=object), allow_pickle=True)
    print(" public(sklearn maps)", pub(pts), " public(product maps)", pub(mine["maps"]), " kmeans2(product maps)", mineq(mine["maps"]))
    return mine["labels"]
so.construct_supertree_oracle(trees, weights, case['strategy'], contract_edges=case['contract'],
                              random_state=np.random.RandomState(case['seed'] % 9973), trace=otrace, trace_matrices=True, steer=steer)
