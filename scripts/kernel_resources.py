"""Registers, LDS and instruction mix of the kernels of libsandcrate_hip.so (build container, no GPU needed).

    python scripts/kernel_resources.py [filter ...] [-- extra hipcc flags]

Compiles sandcrate_hip.hip with the library's flags plus -save-temps -Rpass-analysis=kernel-resource-usage into
/tmp/sc_isa and prints, per kernel whose demangled name contains a filter (default: pass_a, pass_b): SGPRs, VGPRs,
scratch, LDS, the occupancy the registers allow, and static counts of vector / scalar / LDS / memory instructions
of its ISA (a static count, not an execution count: loops count once).
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = Path("/tmp/sc_isa")


def demangle(names):
    res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return res.stdout.splitlines()


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    filters = args or ["pass_a", "pass_b"]
    OUT.mkdir(exist_ok=True)
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
           f"-I{ROOT / 'include'}", f"-I{ROOT / 'sand_crate_amd' / 'csrc'}", "-save-temps",
           "-Rpass-analysis=kernel-resource-usage", *extra, str(ROOT / "sand_crate_amd" / "csrc" / "sandcrate_hip.hip"),
           "-o", str(OUT / "lib.so"), "-ldl"]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=OUT)
    if res.returncode:
        print(res.stderr[-3000:])
        sys.exit(1)
    rows, cur = [], None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: [^ ]+\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    asm = next(OUT.glob("*gfx950*.s")).read_text()
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M):
        bodies.setdefault(m.group(1), m.group(2))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        if not any(f in n for f in filters):
            continue
        body = bodies.get(r["name"], "")
        ops = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
        kinds = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0}
        for o in ops:
            if o.startswith("v_"):
                kinds["valu"] += 1
            elif o.startswith("ds_"):
                kinds["lds"] += 1
            elif o.startswith(("global_", "buffer_", "flat_", "scratch_")):
                kinds["vmem"] += 1
            elif o.startswith("s_load") or o.startswith("s_buffer_load"):
                kinds["smem"] += 1
            elif o.startswith("s_"):
                kinds["salu"] += 1
        short = re.sub(r"^void ", "", n)
        short = short[:short.index("(")] if "(" in short else short
        print(f"{short:62s} sgpr {r.get('TotalSGPRs'):>3s} vgpr {r.get('VGPRs'):>3s} scratch {r.get('ScratchSize'):>3s} "
              f"lds {r.get('LDS Size'):>6s} occ {r.get('Occupancy'):>2s} | static {kinds}")


if __name__ == "__main__":
    main()
