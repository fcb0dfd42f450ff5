"""A whole recursion walked by several in-process ranks (partition.LocalTeams: threads of this process, ONE GPU,
libscs_hip's in-process communicator in RCCL's place) against the single-device run -- run on the GPU box.

Three runs of the same synthetic input with the same seed:
  * one device (the product's default path: the level engine, look-ahead workers);
  * a team of --world ranks with the shared stream and the level engine on every rank: the larger nodes of every
    level dealt over the ranks, nodes of --shard-min vertices and more solved collectively (levels.Engine._process);
  * the same team with Team.level_engine = False: every rank walks every node by itself (rounds 2-5).
All three must give the same Newick string and leave the stream at the same draw.  On one GPU the ranks share the
card, so the times say what the team COSTS there (the replicated level splits, small batches and label
assignments; W ranks' worth of everything in the third run) -- not what N GPUs gain.  One JSON line.

    python tools/team_recursion_check.py [--taxa 20000] [--trees 1000] [--world 2] [--shard-min 4096]
"""
import argparse
import hashlib
import json
import sys
import time
import warnings
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from spectralclustersupertree_amd import levels, scs, synthetic
from spectralclustersupertree_amd.partition import LocalTeams

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=20000)
ap.add_argument("--trees", type=int, default=1000)
ap.add_argument("--strategy", default="branch")
ap.add_argument("--world", type=int, default=2)
ap.add_argument("--shard-min", type=int, default=4096)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--skip-replicated", action="store_true")
args = ap.parse_args()
warnings.simplefilter("ignore")


def make():
    return synthetic.tree_arrays(args.seed + 1, args.taxa, args.trees, random_weights=True)


def walk(arrays, team):
    rs = np.random.RandomState(args.seed)
    t0 = time.perf_counter()
    tree = scs._construct(arrays, args.strategy, True, rs, team=team)
    dt = time.perf_counter() - t0
    return hashlib.sha256(tree.get_newick().encode()).hexdigest(), int(rs.randint(1 << 30)), dt


KEYS = ("roots", "levels", "nodes", "mismatches", "n_large", "deferred", "team_dealt", "team_received", "team_collective",
        "t_small", "t_large", "t_split", "t_labels", "t_redo")


def picked():
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in levels.stats.items() if k in KEYS}


scs.default_device()  # (context creation outside the timings)
res = {"taxa": args.taxa, "trees": args.trees, "strategy": args.strategy, "world": args.world, "shard_min": args.shard_min}
digest, draw, dt = walk(make(), None)
res["single_device"] = {"seconds": round(dt, 3), "newick_sha256": digest, "next_draw": draw, "engine": picked()}


def team_run(level_engine: bool):
    copies = [make() for _ in range(args.world)]
    teams = LocalTeams(args.world, shard_min=args.shard_min)
    for t in teams.teams:
        t.level_engine = level_engine
    try:
        t0 = time.perf_counter()
        out = teams.run(lambda team: walk(copies[team.rank], team))
        wall = time.perf_counter() - t0
    finally:
        teams.close()
    return {"seconds": round(wall, 3), "per_rank_seconds": [round(o[2], 3) for o in out],
            "same_newick_on_every_rank": all(o[0] == digest for o in out),
            "same_next_draw_on_every_rank": all(o[1] == draw for o in out), "engine_all_ranks_summed": picked()}


res["team_level_engine"] = team_run(True)
if not args.skip_replicated:
    res["team_every_rank_every_node"] = team_run(False)
res["ok"] = all(res[k]["same_newick_on_every_rank"] and res[k]["same_next_draw_on_every_rank"]
                for k in ("team_level_engine", "team_every_rank_every_node") if k in res)
print(json.dumps(res))
sys.exit(0 if res["ok"] else 1)
