"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol
include/scs_hip.h declares, and refuses to run without a device (no fallback)."""

import re
from pathlib import Path

import numpy as np
import pytest

from spectralclustersupertree_amd import _native as nv

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "scs_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(nv.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    lib = nv.load_library()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"libscs_hip.so lacks {name}"
    assert lib.scs_version() >= 100


def test_struct_layouts_match_header():
    # sizes the header implies (int32 fields then doubles, natural alignment)
    # (+ allgather_ms_total, allgather_bytes, n_allgather, n_apply32; + apply32_ms_total, apply32_bytes, lowp_renewals, reserved)
    assert nv.C.sizeof(nv.Stats) == 6 * 4 + 9 * 8 + 2 * 8 + 2 * 4 + 2 * 8 + 2 * 4 + 8  # (+ event_pair_ms)
    assert nv.C.sizeof(nv.BuildStats) == 8 * 4 + 8 * 8 + 2 * 4 + 8 + 2 * 4  # (+ tree_parallel_batches, spec_trees, spec_ms, listed_batches, reserved)


def test_struct_fields_match_header_in_name_type_and_order():
    # the ctypes mirrors against include/scs_hip.h, field by field
    import re

    text = (ROOT / "include" / "scs_hip.h").read_text()
    for c_name, mirror in (("scs_stats", nv.Stats), ("scs_build_stats", nv.BuildStats)):
        body = re.search(r"typedef struct " + c_name + r" \{(.*?)\} " + c_name + ";", text, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1)
            for name in names.split(","):
                name = name.strip()
                count = 1
                m = re.match(r"(\w+)\[(\d+)\]", name)
                if m:
                    name, count = m.group(1), int(m.group(2))
                fields.append((name, ctype, count))
        want = {"int32_t": nv.C.c_int32, "double": nv.C.c_double, "int64_t": nv.C.c_int64}
        got = list(mirror._fields_)
        assert len(got) == len(fields), (c_name, [f[0] for f in fields], [g[0] for g in got])
        for (name, ctype, count), (py_name, py_type) in zip(fields, got):
            assert py_name.rstrip("_") == name.rstrip("_"), (c_name, name, py_name)
            expect = want[ctype] * count if count > 1 else want[ctype]
            assert py_type is expect or (count > 1 and py_type._type_ is want[ctype] and py_type._length_ == count), (c_name, name)


def test_no_device_means_loud_failure():
    lib = nv.load_library()
    if lib.scs_device_count() > 0:
        pytest.skip("a HIP device is present")
    from spectralclustersupertree_amd.backend import Device

    with pytest.raises(nv.ScsError, match="no HIP device|no CPU path"):
        Device(0)


def test_construct_supertree_argument_errors_precede_native_calls():
    # reference: src/sc_supertree/scs.py:63-91 -- same exceptions, same texts
    from spectralclustersupertree_amd import construct_supertree
    from spectralclustersupertree_amd.tree import NotCompleted, make_tree

    t = make_tree("((a,b),(c,d))")
    with pytest.raises(ValueError, match="There must be at least one tree to make a supertree."):
        construct_supertree([])
    with pytest.raises(ValueError, match="Invalid weighting strategy selected: 'nope'"):
        construct_supertree([t, t], pcg_weighting="nope")
    with pytest.raises(ValueError, match=r"The number of trees \(2\) and tree weights \(1\) must match."):
        construct_supertree([t, t], weights=[1.0])
    with pytest.raises(ValueError, match="at least one tree"):
        construct_supertree([NotCompleted("ERROR", "x", "y")])


def test_trivial_paths_need_no_device():
    # single tree, <= 2 taxa and fully separable inputs never reach the eigen-solve
    # (reference: scs.py:96-106, 122-124)
    from spectralclustersupertree_amd import construct_supertree
    from spectralclustersupertree_amd.tree import make_tree

    one = construct_supertree([make_tree("((a,b)x,(c,d)y)z;")])
    assert one.sorted().same_shape(make_tree("((a,b),(c,d))").sorted())
    two = construct_supertree([make_tree("(a,b)"), make_tree("(b,a)")])
    assert two.sorted().same_shape(make_tree("(a,b)"))
    agree = construct_supertree([make_tree("((a,b),(c,d))"), make_tree("((a,b),(c,(d,e)))")])
    assert agree.sorted().same_shape(make_tree("((a,b),(c,(d,e)))").sorted())


def test_bootstrap_without_support_raises_type_error():
    # reference: scs.py:656 -- None * float
    from spectralclustersupertree_amd import flatten as fl
    from spectralclustersupertree_amd.tree import make_tree

    with pytest.raises(TypeError):
        fl.flatten_trees([make_tree("(a,(b,(c,d)))")], [1.0], "bootstrap")


def test_load_trees_reads_one_tree_per_line(tmp_path):
    from spectralclustersupertree_amd import load_trees

    p = tmp_path / "x.tre"
    p.write_text("((a,b),(c,d));\n(a:0.1,(b:0.2,c:0.3):0.4);\n")
    trees = load_trees(p)
    assert len(trees) == 2
    assert sorted(trees[1].get_tip_names()) == ["a", "b", "c"]


def test_tables_validate_rejects_bad_input():
    from spectralclustersupertree_amd import synthetic

    tb = synthetic.make_tables(1, 10, 2, "one")
    tb.validate()
    tb.leaf_taxon[0] = 99
    with pytest.raises(ValueError, match="out of range"):
        tb.validate()
    tb.leaf_taxon[0] = 0
    tb.adj_val = tb.adj_val.astype(np.float32)
    with pytest.raises(ValueError, match="float64"):
        tb.validate()
