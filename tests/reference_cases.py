"""Known-answer cases the reference's own tests hold for construct_supertree.

Data only: Newick inputs, options and the expected rooted topology of each case
in the reference's tests/test_spectral_cluster_supertree.py (cited per case),
plus the file fixtures of tests/test_data (committed under
tests/golden/reference_data).  Both the oracle test and the GPU end-to-end
test iterate over this list.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

DATA_DIR = Path(__file__).resolve().parent / "golden" / "reference_data"


@dataclass
class Case:
    name: str
    trees: list[str]
    expected: str
    weights: list[float] | None = None
    pcg_weighting: str = "one"
    contract_edges: bool = True
    cite: str = ""
    extra: dict = field(default_factory=dict)


INLINE_CASES: list[Case] = [
    # test_agreeable, tests/test_spectral_cluster_supertree.py:30-47
    Case("agreeable_1", ["((a,b),(c,d))", "((a,b),(c,(d,e)))"], "((a,b),(c,(d,e)))", cite=":30-40"),
    Case(
        "agreeable_2",
        ["(((a,b),(c,d)),(z,(x,y)))", "((a,((f,g),b)),(c,(d,e)))"],
        "(((a,(b,(f,g))),(c,(d,e))),((x,y),z))",
        cite=":42-47",
    ),
    # test_simple_inconsistency :64-77
    Case(
        "simple_inconsistency",
        ["(a,(b,c))", "(b,(c,d))", "(d,(a,b))"],
        "((a,b),(c,d))",
        cite=":64-77",
    ),
    # test_two_squares_inconsitency :80-106
    Case(
        "two_squares",
        [
            "((a,b),(c,d))",
            "((e,f),(g,h))",
            "(e,(a,c))",
            "(g,(b,d))",
            "(a,(e,g))",
            "(b,(f,h))",
            "(a,(b,e))",
            "(h,(d,g))",
        ],
        "(((a,b),(c,d)),((e,f),(g,h)))",
        cite=":80-106",
    ),
    # test_simple_contration :109-117
    Case(
        "simple_contraction",
        ["(((a,b),c),(d,e))", "((a,b),(c,d))"],
        "(((a,b),c),(d,e))",
        cite=":109-117",
    ),
    # test_size_two_trees :120-134
    Case("size_two_chain", ["(a,b)", "(b,c)", "(c,d)"], "(a,b,c,d)", cite=":124-130"),
    Case("size_two_single", ["(a,b)"], "(a,b)", cite=":132"),
    Case("size_two_twice", ["(a,b)", "(a,b)"], "(a,b)", cite=":133"),
    Case("size_two_swapped", ["(a,b)", "(b,a)"], "(a,b)", cite=":134"),
    # test_simple_weights :137-148
    Case("weights_2_1", ["(a,(b,c))", "(c,(a,b))"], "(a,(b,c))", weights=[2, 1], cite=":144"),
    Case("weights_1001_1", ["(a,(b,c))", "(c,(a,b))"], "(a,(b,c))", weights=[1.001, 1], cite=":145"),
    Case("weights_1_2", ["(a,(b,c))", "(c,(a,b))"], "(c,(a,b))", weights=[1, 2], cite=":147"),
    Case("weights_1_1001", ["(a,(b,c))", "(c,(a,b))"], "(c,(a,b))", weights=[1, 1.001], cite=":148"),
    # test_depth_weighting :151-179
    Case(
        "depth_one",
        ["(a,(b,(c,(d,e))))", "(d,(f,(a,b)))"],
        "((f,a),(b,(c,(d,e))))",
        pcg_weighting="one",
        contract_edges=False,
        cite=":162-167",
    ),
    Case(
        "depth_depth",
        ["(a,(b,(c,(d,e))))", "(d,(f,(a,b)))"],
        "((f,(a,b)),(c,(d,e)))",
        pcg_weighting="depth",
        contract_edges=False,
        cite=":168-173",
    ),
    Case(
        "depth_branch",
        ["(a,(b,(c,(d,e))))", "(d,(f,(a,b)))"],
        "((f,(a,b)),(c,(d,e)))",
        pcg_weighting="branch",
        contract_edges=False,
        cite=":174-179",
    ),
    # test_branch_weighting :182-209
    Case(
        "branch_one",
        ["(a:1,(b:1,(c:1,(d:1,e:1):1):1):1)", "(d:0.1,(f:0.1,(a:0.1,b:0.1):0.1):0.1)"],
        "((f,a),(b,(c,(d,e))))",
        pcg_weighting="one",
        contract_edges=False,
        cite=":192-197",
    ),
    Case(
        "branch_depth",
        ["(a:1,(b:1,(c:1,(d:1,e:1):1):1):1)", "(d:0.1,(f:0.1,(a:0.1,b:0.1):0.1):0.1)"],
        "((f,(a,b)),(c,(d,e)))",
        pcg_weighting="depth",
        contract_edges=False,
        cite=":198-203",
    ),
    Case(
        "branch_branch",
        ["(a:1,(b:1,(c:1,(d:1,e:1):1):1):1)", "(d:0.1,(f:0.1,(a:0.1,b:0.1):0.1):0.1)"],
        "((f,a),(b,(c,(d,e))))",
        pcg_weighting="branch",
        contract_edges=False,
        cite=":204-209",
    ),
    # test_bootstrap :212-241
    Case(
        "bootstrap_one",
        [
            "(a,(b,(c,(d,e)100)100)100)",
            "(a,(b,(d,(c,e)45)100)100)100",
            "(a,(b,(d,(c,e)50)100)100)100",
        ],
        "(a,(b,(d,(c,e))))",
        pcg_weighting="one",
        contract_edges=False,
        cite=":224-229",
    ),
    Case(
        "bootstrap_depth",
        [
            "(a,(b,(c,(d,e)100)100)100)",
            "(a,(b,(d,(c,e)45)100)100)100",
            "(a,(b,(d,(c,e)50)100)100)100",
        ],
        "(a,(b,(d,(c,e))))",
        pcg_weighting="depth",
        contract_edges=False,
        cite=":230-235",
    ),
    Case(
        "bootstrap_bootstrap",
        [
            "(a,(b,(c,(d,e)100)100)100)",
            "(a,(b,(d,(c,e)45)100)100)100",
            "(a,(b,(d,(c,e)50)100)100)100",
        ],
        "(a,(b,(c,(d,e))))",
        pcg_weighting="bootstrap",
        contract_edges=False,
        cite=":236-241",
    ),
]

# (name, source file, expected file, weighting) -- tests/test_spectral_cluster_supertree.py
FILE_CASES = [
    ("dcm_agreeable", "dcm_source_trees.tre", "dcm_model_tree.tre", "one"),  # :50-61
    ("dcm_iq", "dcm_iq_source.tre", "dcm_iq_expected.tre", "branch"),  # :244-248
    ("supertriplets", "supertriplets_source.tre", "supertriplets_expected.tre", "depth"),  # :251-255
]

# test_not_completed :258-274 -- third entry is a NotCompleted, weights [1, 2, 3]
NOT_COMPLETED_CASE = Case(
    "not_completed",
    ["((a,b),(c,d))", "((a,b),(c,(d,e)))"],
    "((a,b),(c,(d,e)))",
    weights=[1, 2, 3],
    cite=":258-274",
)
