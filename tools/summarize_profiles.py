#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_round.sh) -> profiles/<tag>_*: bench line, kernel stats, HBM traffic table
(traffic.json) and the PMC tables.  python tools/summarize_profiles.py [tag]"""
import collections
import csv
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
head = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def counters(d):
    f = src / d / "p_counter_collection.csv"
    if not f.exists():
        f = next(iter((src / d).glob("*counter_collection.csv")), None)
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    if f is None or not f.exists():
        return by
    for r in csv.DictReader(open(f)):
        by[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


shutil.copy(src / "bench.json", dst / f"{tag}_bench.json")
shutil.copy(src / "stats" / "b_kernel_stats.csv", dst / f"{tag}_bench_kernel_stats.csv")
for name in ("rank_shape_bench", "gpass_bench", "lambdamart_bench"):
    if (src / f"{name}.log").exists():
        shutil.copy(src / f"{name}.log", dst / f"{tag}_{name}.log")

# ---- HBM traffic of the headline kernels: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)
fe, wr = counters("fetch"), counters("write")
traffic = {"_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of commit {head} ({tag}); static table, not measured in the bench run"}
rows = []
for (k, g), c in sorted(fe.items()):
    if "FETCH_SIZE" not in c:
        continue
    w = wr.get((k, g), {}).get("WRITE_SIZE", [0.0])
    f_kb, w_kb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(w) / len(w)
    rows.append((short(k), g, len(c["FETCH_SIZE"]), f_kb, w_kb, (2 * f_kb + w_kb) * 1024))
    if "inbatch_sweep_kernel<128, true, true" in k:
        traffic["inbatch_user_pass_n1"] = (2 * f_kb + w_kb) * 1024
    if "inbatch_gt_kernel<128" in k:
        traffic["inbatch_item_pass_n1"] = (2 * f_kb + w_kb) * 1024
with open(dst / f"{tag}_pmc_hbm_traffic.csv", "w") as f:
    f.write("kernel,grid,launches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_per_launch\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
(dst / "traffic.json").write_text(json.dumps(traffic, indent=1))


def table(legs, want, title, out):
    lines = [f"# {title}", "", f"commit {head}; `tools/profile_round.sh {tag}`; SQ_* in quad-cycles summed over the chip, "
             "MFMA_BUSY / LDS_* in cycles, FETCH/WRITE in KB (FETCH_SIZE x 2 = bytes read, MI355X_MICROARCH.md §HBM).", ""]
    for leg in legs:
        c1, c2, c3, c4 = counters(f"{leg}1"), counters(f"{leg}2"), counters(f"{leg}3"), counters(f"{leg}4")
        lines += [f"## {leg}", "",
                  "| kernel | grid | n | wave-cycles | wait % | issue-stall % | issue % | VALU inst | MFMA inst | MFMA busy % of GUI | "
                  "LDS inst | LDS active | LDS conflict | read MB | write MB |", "|" + "---|" * 15]
        for (k, g), c in sorted(c1.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
            if not any(w in k for w in want):
                continue
            m = lambda d, n: (sum(d.get((k, g), {}).get(n, [0])) / max(1, len(d.get((k, g), {}).get(n, [0]))))
            wc = m(c1, "SQ_WAVE_CYCLES")
            if wc <= 0:
                continue
            gui = m(c2, "GRBM_GUI_ACTIVE")
            busy = m(c2, "SQ_VALU_MFMA_BUSY_CYCLES")
            # GRBM_GUI_ACTIVE sums 8 XCDs; MFMA busy sums 1024 SIMDs: busy / (gui / 8 * 1024)
            pct = 100.0 * busy / (gui / 8 * 1024) if gui > 0 else 0.0
            lines.append(f"| `{short(k)}` | {g} | {len(c.get('SQ_WAVE_CYCLES', []))} | {wc:.3g} | {100 * m(c1, 'SQ_WAIT_ANY') / wc:.0f} | "
                         f"{100 * m(c1, 'SQ_WAIT_INST_ANY') / wc:.0f} | {100 * m(c1, 'SQ_ACTIVE_INST_ANY') / wc:.0f} | "
                         f"{m(c1, 'SQ_INSTS_VALU'):.3g} | {m(c2, 'SQ_INSTS_MFMA'):.3g} | {pct:.0f} | {m(c2, 'SQ_INSTS_LDS'):.3g} | "
                         f"{m(c2, 'SQ_LDS_IDX_ACTIVE'):.3g} | {m(c2, 'SQ_LDS_BANK_CONFLICT'):.3g} | {2 * m(c3, 'FETCH_SIZE') / 1024:.1f} | "
                         f"{m(c4, 'WRITE_SIZE') / 1024:.1f} |")
        lines.append("")
    (dst / out).write_text("\n".join(lines))


table(["head"], ["inbatch_sweep_kernel", "inbatch_gt_kernel", "tower_"], "Headline step: in-batch passes and towers", f"{tag}_pmc_headline.md")
table(["sampled_step"], ["tower_", "bpr_pair", "rows_", "adam_rows"], "Sampled-negative step (B = 65 536): towers and row-sparse optimiser",
      f"{tag}_pmc_towers.md")
table(["retrieval"], ["scan_bf16", "refine", "rerank", "finalize", "compact"], "Brute-force top-500 retrieval (4 096 queries x 1 M rows)",
      f"{tag}_pmc_retrieval.md")
table(["serve"], ["ivf_", "gbdt_", "finalize", "rank_features", "tower_fwd"], "cfg5 serve chain (batches of 256 + single requests)",
      f"{tag}_pmc_serve.md")
print("wrote", sorted(p.name for p in dst.glob(f"{tag}_*")))
