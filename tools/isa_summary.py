#!/usr/bin/env python
"""Per-kernel ISA facts of every .hip file (no GPU needed): registers, spills, scratch, LDS and the instruction mix
(MFMA / VALU / SALU / LDS / global / atomics).  `python tools/isa_summary.py > profiles/rNN_isa_summary.txt`."""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "shine_mapping_amd", "csrc")
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        res = out.stdout.splitlines()
        return res if len(res) == len(names) else names
    except Exception:
        return names


def main():
    only = sys.argv[1:]
    print("# kernel | vgpr agpr sgpr | vgpr-spill sgpr-spill scratch-B lds-B | mfma valu salu ds global atomics")
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
            stem = os.path.basename(f)[:-4]
            if only and stem not in only:
                continue
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                            "-I" + SRC, "-c", f, "-save-temps=obj", "-o", os.path.join(tmp, stem + ".o")],
                           cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            asm = open(os.path.join(tmp, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            meta = {}
            for blk in re.split(r"\n  - \.agpr_count:", asm)[1:]:
                blk = ".agpr_count:" + blk
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in KEYS if re.search(r"\.%s:\s+(\d+)" % k, blk)}
            names = [n for n in meta if "rocprim" not in n.lower()]
            pretty = dict(zip(names, demangle(names)))
            print("## %s.hip%s" % (stem, "   (CHECK LIBRARY ONLY: libshine_check.so, tests / tools)" if stem == "shine_step_v0" else ""))
            for n in names:
                m = re.search(r"^%s:[^\n]*\n(.*?)s_endpgm" % re.escape(n), asm, flags=re.S | re.M)
                body = m.group(1) if m else ""
                ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
                cnt = lambda p: sum(1 for i in ins if re.match(p, i))
                k = meta[n]
                short = re.sub(r"\(.*", "", pretty[n]).replace("shine::", "")
                print("%-44s | %3d %3d %3d | %4d %4d %5d %6d | %3d %4d %4d %3d %3d %3d" % (
                    short[:44], k.get("vgpr_count", 0), k.get("agpr_count", 0), k.get("sgpr_count", 0), k.get("vgpr_spill_count", 0),
                    k.get("sgpr_spill_count", 0), k.get("private_segment_fixed_size", 0), k.get("group_segment_fixed_size", 0),
                    cnt(r"v_mfma"), cnt(r"v_(?!mfma)"), cnt(r"s_"), cnt(r"ds_"), cnt(r"(global|buffer|flat)_(?!atomic)"),
                    cnt(r"(global|buffer|flat)_atomic")))


if __name__ == "__main__":
    main()
