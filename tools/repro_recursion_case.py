#!/usr/bin/env python3
"""Re-runs ONE case of tests/fuzz_recursion.py from the dictionary its FAIL line prints (on an MI355X):

    python tools/repro_recursion_case.py "{'seed': 676767675, 'taxa': 260, 'trees': 2, 'leaves': 260,
        'twins': 0, 'strategy': 'bootstrap', 'weighted': False, 'contract': True, 'arrays': False}"
"""
import ast
import sys
import warnings
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from test_gpu_recursion import compare_with_oracle, recursion_input  # noqa: E402

warnings.simplefilter("ignore")
what = ast.literal_eval(sys.argv[1])
trees, weights = recursion_input(what["seed"] % 100000, what["taxa"], what["trees"], what["leaves"], what["twins"],
                                 what["weighted"])
trace, ties = compare_with_oracle(trees, weights, what["strategy"], seed=what["seed"] % 9973,
                                  contract_edges=what["contract"], as_arrays=what["arrays"], ties_allowed=True)
print(f"ok: {len(trace)} spectral calls, {len(ties)} proven ties")
