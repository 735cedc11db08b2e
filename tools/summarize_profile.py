#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory into profiles/<name>/ (SUMMARY.md + the CSVs) and
profiles/traffic_latest.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in KiB by
rocprofv3; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads, so the read side is
doubled; WRITE_SIZE is exact.  Counters were collected in their own passes (one --pmc counter per run).
"""
import csv
import json
import os
import shutil
import sys


def main(src, name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles", name)
    os.makedirs(dst, exist_ok=True)
    for f in ("kernel_stats.csv", "jv_kernel_trace.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv", "bench_under_trace.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    bench = json.loads(open(os.path.join(src, "bench_under_trace.json")).read().strip().splitlines()[-1])
    steps, warm, B = bench["steps"], bench["warmup"], bench["config"]["queries_per_step"]
    grid = str(B * 64)   # one workgroup per query (jv_search_lds / pqf); the persistent pqp kernel launches a resident grid instead
    lines = [f"# rocprofv3 summary — {name}", "",
             f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --profile-mode "
             f"--workload {bench['config']['workload'].split(':')[0]} --steps {steps} --warmup {warm} --rerankk {bench['config']['rerankK']}`",
             "", f"bench line under the tracer: value = {bench['value']} {bench['unit']}, roofline = {json.dumps(bench['roofline'])}", "",
             "## kernel_stats.csv rows of this repo's kernels", "", "| kernel | calls | avg ms | total ms |", "|---|---|---|---|"]
    main_avg = None
    main_kernel = bench["roofline"].get("kernel", "jv_search_lds_kernel")
    for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
        if "jv_" in r["Name"] or "jvb_" in r["Name"]:
            avg = float(r["AverageNs"]) / 1e6
            lines.append(f"| `{r['Name'].split('(')[0]}` | {r['Calls']} | {avg:.4f} | {float(r['TotalDurationNs']) / 1e6:.2f} |")
            if main_kernel in r["Name"] and main_avg is None:
                main_avg = avg
    lines += ["", f"`{main_kernel}` is launched only by the {warm} warm-up + {steps} timed steps in --profile-mode "
                  "(index construction uses `jv_build_search_kernel`, escalation passes `jv_search_retry_kernel`), so its "
                  f"average ({main_avg:.4f} ms) is directly comparable with bench.py's `roofline.kernel_avg_ms` "
                  f"({bench['roofline']['kernel_avg_ms']} ms, HIP events around launch + escalation + big-path launches; rocprofv3 "
                  "attributes the main kernel's last ~2 ms — its slowest queries draining — to the dispatch that follows it, "
                  "which is why the no-op retry launch shows ~2 ms here while HIP events put the whole ladder at 0.04 ms, "
                  "tools/pqf_bailouts.py).", ""]
    # resource usage of the dispatched instances: from the CODE OBJECT's metadata (tools/kernel_resources.py: llvm-readelf --notes
    # on the unbundled gfx950 object) — rocprof's VGPR column is the allocation granule field, not the register count
    tpath = os.path.join(src, "jv_kernel_trace.csv")
    if os.path.exists(tpath):
        import subprocess
        seen = []
        for r in csv.DictReader(open(tpath)):
            nm = r.get("Kernel_Name", "").split("(")[0]
            if main_kernel in nm and nm not in seen:
                seen.append(nm)
        res = {}
        try:
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), "--filter", main_kernel],
                                 capture_output=True, text=True).stdout
            for line in out.splitlines():
                key = line[:90].strip()
                res[key] = line[90:].split()
        except Exception:
            res = {}
        lines += ["## resource usage of the dispatched instances (code object metadata)", "",
                  "| kernel | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane |", "|---|---|---|---|---|---|---|"]
        for nm in seen:
            f = res.get(nm)
            if f and len(f) >= 12:
                lines.append(f"| `{nm}` | {f[1]} | {f[3]} | {f[5]} | {f[7]} | {f[9]} | {f[11]} |")
            else:
                lines.append(f"| `{nm}` | ? | ? | ? | ? | ? | ? |")
        lines.append("")
    # SQ instruction / wait counters of the main kernel (tools/profile_bench.sh, one --pmc pass per counter group)
    sq = {}
    for f in sorted(os.listdir(src)):
        if f.startswith("pmc_sq") and f.endswith(".csv"):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
            for r in csv.DictReader(open(os.path.join(src, f))):
                if main_kernel in r["Kernel_Name"]:
                    sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if sq:
        tot = {k: sum(v) for k, v in sq.items()}
        lines += ["## SQ counters of the main kernel (sums over its dispatches; SQ_*_CYCLES / ACTIVE / WAIT count quad-cycles per wave)", "",
                  "| counter | value |", "|---|---|"]
        for k in sorted(tot):
            lines.append(f"| {k} | {tot[k]:.4g} |")
        wc = tot.get("SQ_WAVE_CYCLES")
        if wc:
            lines += ["", "Per wave-cycle: " + ", ".join(f"{k[3:]} {tot[k] / wc:.3f}" for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in tot) + "."]
            if "SQ_ACTIVE_INST_VALU" in tot and "SQ_WAVES" in tot:
                lines += [f"Waves per SIMD = 4 (16 per CU): the share of a SIMD's cycles in which SOME wave of it issues a vector instruction is at most "
                          f"4 x {tot['SQ_ACTIVE_INST_VALU'] / wc:.3f} = {4 * tot['SQ_ACTIVE_INST_VALU'] / wc:.2f} (a wave64 vector instruction holds its wave for one quad-cycle; "
                          "gfx950 SIMDs are 32 wide and retire it in two cycles: half of that in SIMD-busy terms)."]
        lines.append("")
    traffic = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(src, f"pmc_{cname}.csv")
        if not os.path.exists(p):
            continue
        by_name = {}
        for r in csv.DictReader(open(p)):
            if main_kernel in r["Kernel_Name"] and (r["Grid_Size"] == grid or "pqp" in main_kernel or "pqw" in main_kernel):
                by_name.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        # (the persistent kernel is launched twice per step: the main launch and the wider second launch over the few
        #  flagged queries, a different template instance — the main one is the instance that moves the bytes)
        vals = max(by_name.values(), key=lambda v: sum(v)) if by_name else []
        vals = vals[-steps:]
        if vals:
            traffic[cname] = sum(vals) / len(vals)
        # round 5: the batch's visited counts are taken by jv_visited_fast_kernel / jv_visited_kernel after the search launch (one
        # dispatch of each per step): their bytes belong to the step's traffic
        vis = {}
        for r in csv.DictReader(open(p)):
            if "jv_visited" in r["Kernel_Name"]:
                vis.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        extra = sum(sum(v[-steps:]) / len(v[-steps:]) for v in vis.values() if v)
        if vals and extra:
            traffic[cname + "_visited"] = extra
            traffic[cname] += extra
    if "FETCH_SIZE" in traffic:
        rd = traffic["FETCH_SIZE"] * 1024 * 2
        wr = traffic.get("WRITE_SIZE", 0.0) * 1024
        hbm = rd + wr
        lines += ["## HBM traffic of the main kernel (PMC, separate passes)", "",
                  f"* FETCH_SIZE mean over the {steps} timed launches: {traffic['FETCH_SIZE']:.1f} KiB -> x1024 x2 (gfx950 wide-load correction) = {rd / 1e9:.3f} GB",
                  f"* WRITE_SIZE mean: {traffic.get('WRITE_SIZE', 0.0):.1f} KiB -> {wr / 1e9:.4f} GB",
                  f"* of which the visited-count kernels behind the launch (jv_kernels_vis.hip): FETCH {traffic.get('FETCH_SIZE_visited', 0.0) * 2048 / 1e9:.3f} GB, WRITE {traffic.get('WRITE_SIZE_visited', 0.0) * 1024 / 1e9:.4f} GB",
                  f"* **HBM bytes per launch = {hbm / 1e9:.3f} GB** vs algorithmic {bench['roofline']['algorithmic_bytes_per_launch'] / 1e9:.3f} GB",
                  ""]
        tj = {"workload": bench["config"]["workload"].split(":")[0], "n": bench["config"]["docs_per_gpu"], "batch": B,
              "rerankK": bench["config"]["rerankK"], "dist": bench["config"].get("distribution", "aligned"), "hbm_bytes_per_launch": round(hbm, 1),
              "fetch_size_kib": traffic["FETCH_SIZE"], "write_size_kib": traffic.get("WRITE_SIZE"), "source": f"profiles/{name}",
              "date": __import__("datetime").date.today().isoformat()}
        path = os.path.join(root, "profiles", "traffic_latest.json")
        allj = {}
        if os.path.exists(path):
            try:
                allj = json.load(open(path))
            except Exception:
                allj = {}
        if "entries" not in allj:
            allj = {"entries": {}}
        allj["entries"][tj["workload"]] = tj
        json.dump(allj, open(path, "w"), indent=1)
    open(os.path.join(dst, "SUMMARY.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
