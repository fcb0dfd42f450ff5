"""The ``scs`` command (reference: src/sc_supertree/cli.py:8-39; tests/test_cli.py of the
reference drives the same options through click's runner)."""

import pytest
from click.testing import CliRunner
from reference_cases import DATA_DIR, FILE_CASES

from spectralclustersupertree_amd.cli import scs
from spectralclustersupertree_amd.tree import load_tree


def test_cli_help_and_argument_errors():
    runner = CliRunner()
    res = runner.invoke(scs, [])
    assert res.exit_code in (0, 2) and "--in-file" in res.output and "--disable-contraction" in res.output
    res = runner.invoke(scs, ["-i", "x.tre"])
    assert res.exit_code == 2 and "--out-file" in res.output
    res = runner.invoke(scs, ["-i", "x.tre", "-o", "y.tre", "-p", "nonsense"])
    assert res.exit_code == 2
    res = runner.invoke(scs, ["--version"])
    assert res.exit_code == 0


def test_app_wrappers_validate_like_the_reference():
    from spectralclustersupertree_amd import _app

    with pytest.raises(TypeError, match="Invalid Path Type"):
        _app.load_trees(5)
    trees = _app.load_trees(DATA_DIR / FILE_CASES[0][1])
    assert len(trees) > 1
    with pytest.raises(ValueError, match="does not contain any tip names"):
        _app.outgroup_root(trees[0], priority_outgroups=["no_such_taxon"])


@pytest.mark.gpu
@pytest.mark.parametrize(("name", "src", "exp", "weighting"), FILE_CASES, ids=[c[0] for c in FILE_CASES])
def test_cli_reference_fixtures(tmp_path, name, src, exp, weighting):
    out = tmp_path / "out.tre"
    res = CliRunner().invoke(scs, ["-i", str(DATA_DIR / src), "-o", str(out), "-p", weighting])
    assert res.exit_code == 0, res.output
    assert load_tree(out).sorted().same_shape(load_tree(DATA_DIR / exp).sorted())


@pytest.mark.gpu
def test_cli_disable_contraction(tmp_path):
    src = tmp_path / "in.tre"
    src.write_text("((a,b),(c,(d,e)));\n((a,b),(c,d));\n(((a,b),c),(d,e));\n")
    out = tmp_path / "out.tre"
    res = CliRunner().invoke(scs, ["-i", str(src), "-o", str(out), "-p", "ONE", "--disable-contraction"])
    assert res.exit_code == 0, res.output
    assert sorted(load_tree(out).get_tip_names()) == list("abcde")


@pytest.mark.parametrize(("outgroups", "expected"), [
    (("b",), "(b,(a,((c,d),(e,f))))"), (["x", "y", "b", "c", "d"], "(b,(a,((c,d),(e,f))))"),
    (("c",), "(c,(d,((e,f),(a,b))))"), (("x", "y", "c", "b", "d"), "(c,(d,((e,f),(a,b))))"),
])
def test_outgroup_root_app(outgroups, expected):
    # reference: tests/test_app.py:301-326
    from spectralclustersupertree_amd import _app
    from spectralclustersupertree_amd.tree import make_tree

    got = _app.outgroup_root(make_tree("((a,b),((c,d),(e,f)))"), priority_outgroups=outgroups)
    assert got.sorted().same_shape(make_tree(expected).sorted()), got
