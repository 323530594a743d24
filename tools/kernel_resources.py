"""Kernel resource table (registers, spills, scratch, occupancy, LDS) of the product library, from
`hipcc -Rpass-analysis=kernel-resource-usage`.  usage: kernel_resources.py [remarks.txt] > profiles/rNN_kernel_resources.txt
Without an argument the kernel translation units are compiled here (cross-compiles without a GPU, ~3 min on 8 cores)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "manta_amd", "csrc")


def remarks():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    # one compile per kernel family (manta_amd/build.py: the translation units of the product library), in parallel
    sys.path.insert(0, ROOT)
    from concurrent.futures import ThreadPoolExecutor
    from manta_amd import build as b

    def one(tu):
        with tempfile.TemporaryDirectory() as d:
            cmd = [b.HIPCC] + b.FLAGS + ["-DMANTA_TU=%d" % tu, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, "kernels_tu.cpp"),
                   "-o", os.path.join(d, "k.o")]
            return subprocess.run(cmd, stderr=subprocess.PIPE, text=True, check=True).stderr
    with ThreadPoolExecutor(8) as ex:
        return "".join(ex.map(one, sorted(b.KERNEL_TUS)))


def main():
    txt = remarks()
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    keys = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), (r"TotalSGPRs", "sgpr"), (r"VGPRs Spill", "vgpr_spill"), (r"SGPRs Spill", "sgpr_spill"),
            (r"ScratchSize \[bytes/lane\]", "scratch_B_per_lane"), (r"Occupancy \[waves/SIMD\]", "waves_per_simd"),
            (r"LDS Size \[bytes/block\]", "static_lds_B")]
    print("kernel\t" + "\t".join(k[1] for k in keys))
    seen = set()
    for b in blocks:
        name = b.split("\n")[0].split(" [")[0].strip()
        if name in seen:
            continue
        seen.add(name)
        vals = []
        for pat, _ in keys:
            m = re.search(r"remark: [^\n]*\s" + pat + r": (\d+)", b)
            vals.append(m.group(1) if m else "?")
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE, text=True).stdout.strip() or name
        except OSError:
            pass
        name = re.sub(r"\(manta_dev::\w+\)$", "", name).replace("manta_dev::", "")
        print(name + "\t" + "\t".join(vals))


if __name__ == "__main__":
    main()
