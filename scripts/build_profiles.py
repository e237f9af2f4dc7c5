"""Aggregate the rocprofv3 outputs of dbg/collect_profiles.sh into the three files kept under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha256, pool_sources_sha256  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
os.makedirs(dst, exist_ok=True)
stats = glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True)
shutil.copy(stats[0], dst + f"/{tag}_bench_kernel_stats.csv")
pool = glob.glob(src + "/pool/**/*kernel_stats.csv", recursive=True)
if pool:
    shutil.copy(pool[0], dst + f"/{tag}_pool_saturated_kernel_stats.csv")
def counters(sub):
    f = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":           # cycles of the graphics clock while the dispatch ran, against its duration
            agg[r["Kernel_Name"]]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg
fetch, write, mfma = counters("fetch"), counters("write"), counters("mfma")
rows = []
for k in fetch:
    fs = fetch[k].get("FETCH_SIZE", []); ws = write.get(k, {}).get("WRITE_SIZE", [])
    mb = mfma.get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", []); ga = mfma.get(k, {}).get("GRBM_GUI_ACTIVE", [])
    if not fs or not ws: continue
    fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
    mbm = sum(mb) / len(mb) if mb else 0.0
    gam = sum(ga) / len(ga) / 8 if ga else 0.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    frac = mbm / (1024 * gam) if gam else 0.0            # 1024 SIMDs
    rows.append((k, len(fs), fm, wm, (2 * fm + wm) * 1024, mbm, gam, frac))
rows.sort(key=lambda r: -r[1] * r[4])
with open(dst + f"/{tag}_bench_hbm_counters.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "fetch_size_kb_mean", "write_size_kb_mean", "hbm_bytes_per_dispatch_corrected", "mfma_busy_cycles_mean", "gui_active_cycles_per_xcd_mean", "mfma_busy_frac"])
    w.writerows(rows)
g = [r for r in rows if "gemm_256x256" in r[0]]
n = sum(r[1] for r in g)
js = {"kernel": "gemm_256x256 (all epilogue variants)", "kernel_sources_sha256": kernel_sources_sha256(),
      "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (three passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-points",
      "dispatches": n,
      "fetch_size_kb_mean": sum(r[1] * r[2] for r in g) / n, "write_size_kb_mean": sum(r[1] * r[3] for r in g) / n,
      "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE exact; both count requests leaving the L2, Infinity-Cache hits included",
      "traffic_bytes_per_launch": sum(r[1] * r[4] for r in g) / n,
      "mfma_busy_fraction_of_simd_cycles": sum(r[1] * r[5] for r in g) / sum(r[1] * 1024 * r[6] for r in g),
      # GRBM_GUI_ACTIVE / 8 XCDs over the dispatch's own duration: the graphics clock the kernel actually ran at
      "gfx_clock_ghz_in_kernel": sum(sum(mfma[r[0]]["GRBM_GUI_ACTIVE"]) for r in g) / 8 / sum(sum(mfma[r[0]]["_ns"]) for r in g)}
json.dump(js, open(dst + f"/{tag}_gemm_traffic.json", "w"), indent=1)
print(json.dumps(js, indent=1))
for r in rows[:8]: print(r[0][:70], r[1], f"{r[4]/1e6:.0f} MB", f"mfma {r[7]:.3f}")


# ---- the attention kernel's issue / wait split (counters of the kernel the bench step launches, by its own name)
def _file(sub):
    f = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)
    return f[0] if f else None


if _file("issue"):
    issue = counters("issue")
    with open(dst + f"/{tag}_attention_counters.txt", "w") as f:
        f.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY "
                "SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 (one pass): every dispatch of the attention kernel in the step\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the 1024 SIMDs;\n"
                "# GRBM_GUI_ACTIVE is summed over the 8 XCDs.\n")
        for k in issue:
            if "attn_d64" not in k:
                continue
            c = {n: sum(v) / len(v) for n, v in issue[k].items()}
            n = len(issue[k]["SQ_WAVE_CYCLES"])
            f.write(f"{k}\n")
            for name in sorted(c):
                if name != "_ns":
                    f.write(f"    {name:28s} n={n} mean={c[name]:.4g}\n")
            gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
            if gui and c.get("SQ_WAVE_CYCLES"):
                wc = c["SQ_WAVE_CYCLES"]
                f.write(f"    derived: duration {c['_ns'] / 1e3:.1f} us at {gui / c['_ns']:.2f} GHz; matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui):.3f} of SIMD cycles; "
                        f"of wave time: issuing {c['SQ_ACTIVE_INST_ANY'] / wc:.2f} (vector ALU {c['SQ_ACTIVE_INST_VALU'] / wc:.2f}), parked at waits / barriers {c['SQ_WAIT_ANY'] / wc:.2f}, "
                        f"stalled at issue {c['SQ_WAIT_INST_ANY'] / wc:.2f}\n")
    print(open(dst + f"/{tag}_attention_counters.txt").read())

# ---- memory traffic of the pooling kernel's chip-filling launch (16-bit edges)
if _file("poolfetch") and _file("poolwrite"):
    pf, pw = counters("poolfetch"), counters("poolwrite")
    for k in pf:
        if "tave_kernel" not in k or k not in pw:
            continue
        fm = sum(pf[k]["FETCH_SIZE"]) / len(pf[k]["FETCH_SIZE"])
        wm = sum(pw[k]["WRITE_SIZE"]) / len(pw[k]["WRITE_SIZE"])
        js = {"kernel": k, "units": 2048, "kernel_sources_sha256": pool_sources_sha256(), "dispatches": len(pf[k]["FETCH_SIZE"]),
              "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python3 scripts/pool_saturated.py",
              "fetch_size_kb_mean": fm, "write_size_kb_mean": wm,
              "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled; WRITE_SIZE exact; both count requests leaving the L2, Infinity-Cache hits included",
              "traffic_bytes_per_launch": (2 * fm + wm) * 1024, "algorithmic_bytes_per_launch": 2048 * 439296}
        js["traffic_over_algorithmic"] = js["traffic_bytes_per_launch"] / js["algorithmic_bytes_per_launch"]
        json.dump(js, open(dst + f"/{tag}_pool_traffic.json", "w"), indent=1)
        print(json.dumps(js, indent=1))
