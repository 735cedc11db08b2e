#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in a built library, read from the gfx950 code object's metadata
(llvm-readelf --notes on the unbundled object) — what the hardware is told, not rocprof's allocation-granule column.
usage: tools/kernel_resources.py [lib.so | obj.o ...] [--filter regex] [--md]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    """yields paths of the gfx950 code objects inside `path` (a hipcc object file or shared library: the fat binary sits in the
    .hip_fatbin section, one clang offload bundle per translation unit, back to back)"""
    tmp = tempfile.mkdtemp(prefix="jvres_")
    fat = os.path.join(tmp, "fat.bin")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(tmp, "copy")], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for i, s in enumerate(starts):
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(data[s:(starts[i + 1] if i + 1 < len(starts) else len(data))])
        o = os.path.join(tmp, f"co{i}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={o}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(o) and os.path.getsize(o) > 0:
            yield o


def kernels(co):
    import yaml
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    i = txt.find("---")
    j = txt.rfind("...")
    if i < 0:
        return
    doc = yaml.safe_load(txt[i + 3:j if j > i else None])
    for k in (doc or {}).get("amdhsa.kernels", []):
        yield {kk.lstrip("."): vv for kk, vv in k.items()}


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True)
    return r.stdout.strip().split("(")[0] if r.returncode == 0 and r.stdout.strip() else n


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = re.compile(sys.argv[sys.argv.index("--filter") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--filter") + 1]]
    md = "--md" in sys.argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not args:
        args = [os.path.join(root, "opensearch-jvector_amd", "lib", "libjvgpu.so")]
    rows = []
    for path in args:
        for co in code_objects(path):
            for k in kernels(co):
                name = demangle(k.get("name", "?"))
                if flt and not flt.search(name):
                    continue
                rows.append((name, int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0)), int(k.get("sgpr_count", 0)),
                             int(k.get("vgpr_spill_count", 0)), int(k.get("sgpr_spill_count", 0)), int(k.get("private_segment_fixed_size", 0)),
                             int(k.get("group_segment_fixed_size", 0))))
    rows.sort()
    if md:
        print("| kernel | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane | static LDS B |\n|---|---|---|---|---|---|---|---|")
        for r in rows:
            print(f"| `{r[0]}` | " + " | ".join(str(x) for x in r[1:]) + " |")
    else:
        for r in rows:
            print(f"{r[0]:90s} vgpr {r[1]:3d} agpr {r[2]:3d} sgpr {r[3]:3d} vspill {r[4]:3d} sspill {r[5]:3d} scratch {r[6]:5d} lds {r[7]}")
    if rows:
        print(f"# {len(rows)} kernels; scratch per lane: max {max(r[6] for r in rows)} B, {sum(1 for r in rows if r[6] == 0)} without any", file=sys.stderr)


if __name__ == "__main__":
    main()
