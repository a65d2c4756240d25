"""Register / scratch table of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).

    python tools/kernel_resources.py eav_amd/csrc/gemm_sp.hip [-D... extra hipcc flags]

Prints one line per kernel: VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS bytes, demangled name.  No GPU needed."""
import os
import re
import subprocess
import sys


def resources(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null", "-I", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eav_amd", "csrc"), *extra]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["demangled"] = n.replace("(anonymous namespace)::", "")
    return rows


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    print(f"{'VGPR':>5} {'AGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
    for r in resources(src, extra):
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('ScratchSize', '?'):>8} "
              f"{r.get('Occupancy', '?'):>4} {r.get('LDS Size', '?'):>7}  {r['demangled'][:110]}")


if __name__ == "__main__":
    main()
