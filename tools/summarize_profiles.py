"""Turn gpurun_out/profiles_raw/ (tools/collect_profiles.sh) into the committed files under
profiles/: kernel-time table, PMC traffic per launch, SQ counters and phase stamps of the
accumulate kernel, bench lines, pmc_traffic.json.

    python tools/summarize_profiles.py r02
"""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = Path(__file__).resolve().parent.parent
raw = root / "gpurun_out" / "profiles_raw"
out = root / "profiles"


def newest(pattern, must_contain=None):
    files = sorted(glob.glob(str(raw / pattern)), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    for f in files:
        if must_contain is None or must_contain in Path(f).read_text(errors="ignore"):
            return f
    raise FileNotFoundError(pattern)


stats_csv = newest("kt/*/*kernel_stats.csv", "k_symm")
shutil.copy(stats_csv, out / f"{tag}_bench_cfg2_kernel_stats.csv")
shutil.copy(raw / "bench_under_rocprof.json", out / f"{tag}_bench_cfg2_under_rocprof.json")
shutil.copy(raw / "bench_default.json", out / f"{tag}_bench_default_run.json")
if (raw / "bench_cfg4.json").exists():
    shutil.copy(raw / "bench_cfg4.json", out / f"{tag}_bench_cfg4_single_gpu.json")
stamps = (raw / "stamps.txt").read_text()
(out / f"{tag}_accumulate_phase_stamps.txt").write_text(
    "In-kernel s_memtime phase shares of the tile kernels (SCS_ACC_STAMP=1 diagnostic variants, configs[2]: the first,\n"
    "short tree batch is k_accumulate_mono's, the two long ones k_accumulate_spec's); cycles are per wave per\n"
    "(workgroup, tree) step, three waves per SIMD.  k_accumulate_spec: every wave adds its own phases, so a producer\n"
    "phase is averaged over all twelve waves (x 3 for the producers' own figure), a consumer phase x 1.5.\n\n" + stamps)

rows = list(csv.DictReader(open(stats_csv)))
bench = json.load(open(raw / "bench_under_rocprof.json"))
default = json.load(open(raw / "bench_default.json"))


def pmc(kind, needle="k_accumulate"):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    f = newest(f"{kind}/*/*counter_collection.csv", needle)
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch, write = pmc("fetch"), pmc("write")
n, m, b = bench["config"]["n_taxa"], bench["config"]["n_trees"], bench["config"]["lobpcg_block"]


def key_of(acc, needle):
    return next(k for k in acc if needle in k)


def symm_key(acc, image):
    """the SYMM kernel that streams W (double) or its single-precision image (float) among the profiled names"""
    found = [k for k in acc
             if ("k_symm_tri<" in k or "k_symm_tri_tf<" in k or "k_symm<" in k) and (("float" in k) == image)]
    # (the loop's launches -- k_symm_tri_tf -- outnumber the few of the plain kernel: take the kind profiled most)
    return max(found, key=lambda k: acc[k]["FETCH_SIZE" if "FETCH_SIZE" in acc[k] else "WRITE_SIZE"][0]) if found else None


k_symm = symm_key(fetch, False) or key_of(fetch, "k_symm")
k_img = symm_key(fetch, True)
roofs = {r["kernel"].split(" ")[0].split("<")[0]: r for r in (bench["roofline"], bench["roofline_other"])}
for r in list(roofs.values()):  # the SYMM entry carries the launches of the other arithmetic inside it
    for nest in ("double_precision_launches", "image_launches"):
        if r.get(nest):
            roofs[r[nest]["kernel"].split(" ")[0].split("<")[0]] = r[nest]
# the tile kernel that walked the long tree batches (round 4: k_accumulate_spec up to 20 000 leaves)
ACC = next(k for k in roofs if k.startswith("k_accumulate"))
k_acc = key_of(fetch, ACC)
symm_alg = roofs["k_symm_tri" if "k_symm_tri" in roofs else next(k for k in roofs if k.startswith("k_symm"))]["bytes_per_launch"]
acc_alg = roofs[ACC]["bytes_per_launch"]
alg = {k_symm: (symm_alg, "bytes of W tiles streamed + block in/out"),
       **({k_img: (roofs["k_symm_tri_tf"]["bytes_per_launch"], "bytes of the single-precision image's tiles + block in/out")}
          if k_img and "k_symm_tri_tf" in roofs else {}),
       key_of(fetch, "k_degrees"): (8.0 * n * n, "8 V^2 read (+ 2 V^2 written when the pass leaves the single-precision image)"),
       k_acc: (acc_alg, "per tree batch: W tile sums written once + tables read once")}

lines = [f"# Round-{tag[1:]} profile: `python3 bench.py --steps 5 --no-cpu-baseline --no-extra --no-parity` "
         f"(BASELINE.json configs[2]: {n} taxa / {m} trees / {bench['config']['pcg_weighting']}), one MI355X", "",
         "Collected by `tools/collect_profiles.sh` with `rocprofv3 --kernel-trace --stats --output-format csv` (kernel times) and, in "
         "separate runs with `--kernel-trace` only, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and three SQ counter passes (bench with "
         "--steps 2 --warmup 0); summarised by `tools/summarize_profiles.py`.  Files: "
         f"`{tag}_bench_cfg2_kernel_stats.csv` (raw stats), `{tag}_bench_cfg2_under_rocprof.json` (bench line printed under the "
         f"profiler), `{tag}_bench_default_run.json` (un-profiled default `python bench.py` line incl. parity gates, cpu_baseline, "
         f"seeds, planted input and the configs[1]/configs[3] reference passes), `{tag}_accumulate_phase_stamps.txt`, "
         f"`{tag}_accumulate_sq_counters_*.txt` (SQ counters of the accumulate kernel: round-1 structure, with the hand-scheduled "
         f"cell loop, final structure), `{tag}_bench_cfg4_single_gpu.json` (configs[4] on ONE device).", "",
         "## Kernel time (11 passes of the hot path: 1 warm-up + 5 timed by the protocol + 5 on HBM-resident tables)", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:16]:
    lines.append(f"| `{r['Name'].split('(')[0]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | "
                 f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
lines += ["", "Live measurement inside bench.py in the same run (HIP events on the library's stream):"]
for name, r in roofs.items():
    if r.get("bound") == "lds":
        lines.append(f"* `{name}`: {r['avg_launch_ms']*1e3:.1f} us per launch x {r['launches_per_step']:.0f} launches per step; "
                     f"bound = LDS: {r['achieved']:.0f} GB/s of ds_read_b64 bytes (8 per cell-tree) of ~150 000 = frac {r['frac']}; "
                     f"algorithmic HBM bytes {r['hbm']['achieved']:.1f} GB/s (frac {r['hbm']['frac']}); "
                     f"{r['cell_trees_per_s']:.3e} cell-trees/s, fp64 VALU frac {r['frac_f64_valu']}.")
    else:
        lines.append(f"* `{name}`: {r['avg_launch_ms']*1e3:.1f} us per launch x {r['launches_per_step']:.0f} launches per step, "
                     f"{r['achieved']:.1f} GB/s of algorithmic bytes (frac {r['frac']}).")
lines += ["", "## HBM-side traffic per launch (PMC; FETCH_SIZE / WRITE_SIZE count KB: x 1024)", "",
          "gfx950 note (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half of the bytes of a wide coalesced "
          "streaming read, so the SYMM kernel and k_degrees are doubled before comparing with the algorithmic byte count; the "
          "accumulate kernel's fetches are random 8-byte gathers (uncalibrated width) and are quoted raw.", "",
          "| kernel | launches | FETCH_SIZE raw bytes | corrected | WRITE_SIZE bytes | algorithmic bytes per launch |",
          "|---|---|---|---|---|---|"]
traffic = {}
for k, (ab, note) in alg.items():
    fr = fetch[k]["FETCH_SIZE"][1] / fetch[k]["FETCH_SIZE"][0] * 1024
    wr = write[k]["WRITE_SIZE"][1] / write[k]["WRITE_SIZE"][0] * 1024
    corr = fr * 2 if ("k_symm" in k or "k_degrees" in k) else fr
    lines.append(f"| `{k}` | {fetch[k]['FETCH_SIZE'][0]} | {fr:,.0f} | {corr:,.0f} | {wr:,.0f} | {ab:,.0f} ({note}) |")
    traffic[k] = (fr, corr, wr)
fr, corr, wr = traffic[k_symm]
afr, acorr, awr = traffic[k_acc]
lines += ["", f"Reading: the SYMM kernel moves {(corr + wr)/1e6:.0f} MB per launch against {symm_alg/1e6:.0f} MB algorithmic "
          "(upper tiles of W + partial sums) -- no wasted re-reads.  The accumulate kernel fetches "
          f"{afr/1e9:.1f} GB per launch against {acc_alg/1e9:.2f} GB algorithmic: range-minimum gathers that miss the L2 "
          "(each 8-byte gather moves a whole line) and block records; it is bound by L2 line traffic, LDS and VALU, not by HBM "
          "(see the SQ counters and the phase stamps).", ""]

# SQ counters of the accumulate kernel from this collection
try:
    sq = defaultdict(lambda: [0, 0.0])
    for kind in ("sq1", "sq2", "sq3"):
        f = newest(f"{kind}/*/*counter_collection.csv", ACC)
        for r in csv.DictReader(open(f)):
            if ACC in r["Kernel_Name"]:
                a = sq[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    per = {k: v[1] / v[0] for k, v in sq.items()}
    cu_cycles = per["GRBM_GUI_ACTIVE"] / 8 * 256
    lines += [f"## SQ counters of `{ACC}` (per launch)", "",
              f"CU cycles (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs): {cu_cycles:.3e}; LDS busy (SQ_LDS_IDX_ACTIVE): "
              f"{per['SQ_LDS_IDX_ACTIVE'] / cu_cycles:.2f} of them, bank-conflict cycles (SQ_LDS_BANK_CONFLICT): "
              f"{per['SQ_LDS_BANK_CONFLICT'] / cu_cycles:.2f}; VALU busy (4 x SQ_ACTIVE_INST_VALU / (4 SIMDs x CU cycles)): "
              f"{per['SQ_ACTIVE_INST_VALU'] * 4 / (cu_cycles * 4):.2f}; VALU instructions per wave: "
              f"{per['SQ_INSTS_VALU'] / per['SQ_WAVES']:.0f}, LDS instructions per wave: {per['SQ_INSTS_LDS'] / per['SQ_WAVES']:.0f}, "
              f"waves: {per['SQ_WAVES']:.0f}.", ""]
    (out / f"{tag}_accumulate_sq_counters_final.txt").write_text(
        "\n".join(f"{k:28s} per launch {v:.6g}" for k, v in sorted(per.items())) + "\n")
except (FileNotFoundError, KeyError, ZeroDivisionError) as e:
    lines += [f"(SQ counter summary unavailable: {e})", ""]

st = default["stages"]
lines += ["## Default bench line (un-profiled)", "",
          f"configs[2]: {default['value']*1e3:.1f} ms per step by the SURVEY 8d protocol (tables upload {st.get('tables_upload_ms', 0):.1f} ms, "
          f"build {st['build_ms']:.1f} ms of which accumulate "
          f"{st['build_accumulate_ms']:.1f}, solve {st['fiedler_ms']:.1f} ms of which SYMM {st['fiedler_symm_ms']:.1f}, "
          f"{st['lobpcg_iterations']:.0f} LOBPCG iterations); on HBM-resident tables {default.get('value_tables_resident', 0)*1e3:.1f} ms; "
          f"dominant kernel {default['roofline']['kernel'].split(' ')[0]} (frac {default['roofline']['frac']}); path figure "
          f"{default['roofline_path']['achieved']:.0f} GB/s = {default['roofline_path']['frac']} of spec; cpu_baseline "
          f"{default['cpu_baseline']['value']:.1f} s ({default['cpu_baseline']['sample'][:60]}...); reference-style dict build sample: "
          f"{json.dumps(default['cpu_baseline'].get('reference_style_build', {}))}; kmeans2 fast path active: "
          f"{default.get('kmeans2_fast_path_active')}.", "",
          f"Parity gates at full size: {json.dumps(default['parity'])}", "",
          f"Seeds: {json.dumps(default.get('seeds', {}))}", "",
          f"Planted input: {default['planted'].get('value')} s, lambda2 {default['planted']['stages']['lambda2']:.4f}, "
          f"lambda3 {default['planted']['stages']['lambda3']:.4f}, {default['planted']['stages']['lobpcg_iterations']:.0f} iterations.", ""]
for k, v in default.get("other_workloads", {}).items():
    if "value" in v:
        lines.append(f"{k}: {v['value']:.4f} s per step (build {v['stages']['build_ms']:.1f} ms, solve {v['stages']['fiedler_ms']:.1f} ms, "
                     f"{v['stages']['lobpcg_iterations']:.0f} iterations), W rows mismatched: {v['parity']['w_cells_mismatched']}.")
if (raw / "bench_cfg4.json").exists():
    c4 = json.load(open(raw / "bench_cfg4.json"))
    lines.append(f"cfg4 (own run): {c4['value']:.3f} s per step (build {c4['stages']['build_ms']:.0f} ms, solve "
                 f"{c4['stages']['fiedler_ms']:.0f} ms, {c4['stages']['lobpcg_iterations']:.0f} iterations), W rows mismatched: "
                 f"{c4.get('parity', {}).get('w_cells_mismatched')}.")
(out / f"{tag}_bench_cfg2_summary.md").write_text("\n".join(lines) + "\n")

pj = {"_comment": "HBM-side bytes per launch from committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction "
                  "for wide coalesced reads where that applies, + WRITE_SIZE), keyed by workload name; bench.py quotes the "
                  "matching entry as roofline.traffic",
      "cfg2": [
          {"kernel": "k_symm_tri" if "k_symm_tri" in k_symm else k_symm.replace("void ", "").split("<")[0], "fetch_raw": round(fr), "fetch_corrected": round(corr),
           "write": round(wr), "traffic": round(corr + wr), "source": f"profiles/{tag}_bench_cfg2_summary.md"},
          *([{"kernel": "k_symm_tri_tf", "fetch_raw": round(traffic[k_img][0]), "fetch_corrected": round(traffic[k_img][1]),
              "write": round(traffic[k_img][2]), "traffic": round(traffic[k_img][1] + traffic[k_img][2]),
              "source": f"profiles/{tag}_bench_cfg2_summary.md (the launches that stream the single-precision image)"}]
            if k_img in traffic else []),
          {"kernel": ACC, "fetch_raw": round(afr), "fetch_corrected": round(acorr), "write": round(awr),
           "traffic": round(acorr + awr), "source": f"profiles/{tag}_bench_cfg2_summary.md (fetches are 8-byte gathers: "
                                                     "uncalibrated width, quoted raw)"}]}


def extra_rows(suffix, label, bench_file, kernels):
    """PMC rows + kernel stats of another workload's passes (kt_<suffix>, fetch_<suffix>, write_<suffix>)."""
    try:
        f2, w2 = pmc(f"fetch_{suffix}"), pmc(f"write_{suffix}")
        st2 = newest(f"kt_{suffix}/*/*kernel_stats.csv", "k_symm")
        shutil.copy(st2, out / f"{tag}_bench_{label}_kernel_stats.csv")
        if (raw / bench_file).exists():
            shutil.copy(raw / bench_file, out / f"{tag}_bench_{label}_under_rocprof.json")
        rows2 = []
        for needle in kernels:
            if needle == "k_symm_tri":
                k = symm_key(f2, False)
            elif needle == "k_symm_tri_tf":
                k = symm_key(f2, True)
            else:
                k = next((k for k in f2 if needle in k), None)
            if k is None:
                continue
            frr = f2[k]["FETCH_SIZE"][1] / f2[k]["FETCH_SIZE"][0] * 1024
            wrr = w2[k]["WRITE_SIZE"][1] / w2[k]["WRITE_SIZE"][0] * 1024
            cor = frr * 2 if "k_symm" in k else frr
            rows2.append({"kernel": needle if needle.startswith("k_symm_tri") else k.replace("void ", "").split("<")[0].split("(")[0], "launches_profiled": f2[k]["FETCH_SIZE"][0],
                          "fetch_raw": round(frr), "fetch_corrected": round(cor), "write": round(wrr),
                          "traffic": round(cor + wrr),
                          "source": f"profiles/{tag}_bench_{label}_kernel_stats.csv + PMC passes of tools/collect_profiles.sh"
                                    + ("" if "k_symm" in k else " (8-byte gathers: uncalibrated width, quoted raw)")})
        return rows2
    except (FileNotFoundError, KeyError, StopIteration, ZeroDivisionError) as e:
        print(f"(no {label} rows: {e})")
        return []


cfg3_rows = extra_rows("cfg3", "cfg3", "bench_cfg3_under_rocprof.json", ["k_symm_tri", "k_symm_tri_tf", "k_accumulate_spec", "k_accumulate_mono"])
if cfg3_rows:
    pj["cfg3"] = cfg3_rows
boot_rows = extra_rows("boot", "cfg2_bootstrap", "bench_boot_under_rocprof.json", ["k_accumulate_gen", "k_symm_tri"])
if boot_rows:
    pj["custom_10000_500_bootstrap"] = boot_rows
(out / "pmc_traffic.json").write_text(json.dumps(pj, indent=1))
with open(out / f"{tag}_bench_cfg2_summary.md", "a") as fh:
    fh.write("\n## HBM-side traffic of the other workloads (same PMC recipe; bench.py quotes them as roofline.traffic)\n\n")
    for key in ("cfg3", "custom_10000_500_bootstrap"):
        for r in pj.get(key, []):
            fh.write(f"* {key} `{r['kernel']}`: fetch {r['fetch_raw']:,} raw / {r['fetch_corrected']:,} corrected, write {r['write']:,} "
                     f"-> {r['traffic']:,} bytes per launch ({r['launches_profiled']} launches profiled)\n")
print("\n".join(lines[:60]))
