"""``scs`` command line front-end with the reference's options
(reference: src/sc_supertree/cli.py:8-39): default weighting ``branch`` (the library call
defaults to ``one``), no tree weights, no seed.  The source trees are read straight into flat
arrays by the C Newick loader -- no tree object is built for the input."""

from __future__ import annotations

import click

from spectralclustersupertree_amd import __version__


@click.command(no_args_is_help=True)
@click.version_option(__version__)
@click.option("-i", "--in-file", required=True, help="Line-separated Newick file with the source trees (parsed straight into flat arrays).")
@click.option("-o", "--out-file", required=True, help="Where the supertree is written (Newick).")
@click.option(
    "-p",
    "--pcg-weighting",
    help="How an edge of the proper cluster graph is weighted by the LCA of its two taxa.",
    default="branch",
    type=click.Choice(["one", "depth", "branch", "bootstrap"], case_sensitive=False),
)
@click.option(
    "--disable-contraction",
    help="Keep always-together taxa as separate vertices (slower, same result up to ties).",
    default=False,
    is_flag=True,
)
def scs(in_file: str, out_file: str, pcg_weighting: str, *, disable_contraction: bool) -> None:
    """Spectral Cluster Supertree of the source trees in IN_FILE, on the MI355X core."""
    from spectralclustersupertree_amd import construct_supertree
    from spectralclustersupertree_amd.load import load_tree_arrays

    supertree = construct_supertree(
        load_tree_arrays(in_file),
        pcg_weighting=pcg_weighting.lower(),
        contract_edges=not disable_contraction,
    )
    from spectralclustersupertree_amd.scs import default_team

    team = default_team()
    if team is None or team.rank == 0:  # a launched job: every rank holds the tree, one writes it
        supertree.write(out_file)


if __name__ == "__main__":
    scs()
