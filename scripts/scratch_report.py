"""Where the lane kernel's scratch is.  Compiles csrc/corr_lane.hip to gfx950 assembly (device only) and, for every instantiation
of corr_epl_lane_kernel, reports the scratch size and where the scratch loads / stores sit: inside the sample loops of the lean
path (innermost loops that load the record and hold the arms' FMAs), inside the per-sample float64 loops of the exact path (tied
samples), or in the straight-line code around them (flush, reseeding, epilogue).  usage: python scripts/scratch_report.py [out.md]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cu-sdr-collection_amd", "csrc", "corr_lane.hip")


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "corr_lane.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only",
                        "-S", SRC, "-o", asm], check=True, stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_ZN") and "corr_epl_lane_kernel" in l and ": " in l]
    dem = subprocess.run(["c++filt"] + [n for _, n in starts], capture_output=True, text=True).stdout.strip().split("\n")
    rows = []
    for k, ((i, _), d) in enumerate(zip(starts, dem)):
        m = re.search(r"corr_epl_lane_kernel<([^>]*)>", d)
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        for j in range(i, end):
            if lines[j].startswith(".Lfunc_end"):
                end = j
                break
        body = lines[i:end]
        meta = "\n".join(lines[end:end + 80])
        g = lambda pat: int((re.search(pat, meta) or [None, "-1"])[1])
        labels = {l.split(":")[0]: j for j, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
        loops = []
        for j, l in enumerate(body):
            mm = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < j:
                loops.append((labels[mm.group(1)], j))
        inner = [(a, b) for a, b in loops if not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in loops)]
        count = lambda a, b, pat: sum(1 for q in range(a, b + 1) if re.search(pat, body[q]))
        lean = [(a, b) for a, b in inner if count(a, b, r"global_load|buffer_load") > 0 and count(a, b, r"v_fma(c|_mix)?_f32|v_pk_fma") >= 30]
        exact = [(a, b) for a, b in inner if (a, b) not in lean and count(a, b, r"\bv_\w+_f64\b") >= 8]
        rows.append((m.group(1), g(r"; NumVgprs: (\d+)"), g(r"; ScratchSize: (\d+)"), count(0, len(body) - 1, r"scratch_"), len(lean),
                     sum(b - a for a, b in lean), sum(count(a, b, r"scratch_") for a, b in lean), sum(count(a, b, r"scratch_") for a, b in exact)))
    hdr = ("| ARMS, MODE, CL, TAB, DEVLOOP, DER | VGPRs | scratch B/lane | scratch loads+stores (static) | sample loops | their instructions | scratch ops inside them | inside the exact-path loops |\n"
           "|---|---|---|---|---|---|---|---|\n")
    txt = hdr + "".join("| %s | %d | %d | %d | %d | %d | %d | %d |\n" % r for r in rows)
    txt += ("\n%d instantiations, %d with scratch; %d of %d static scratch loads / stores sit inside sample loops (innermost loops that load the record and hold >= 30 f32 FMAs), "
            "%d inside the per-sample float64 loops of the exact path (tied samples), the rest in straight-line code around them.\n"
            % (len(rows), sum(1 for r in rows if r[2] > 0), sum(r[6] for r in rows), sum(r[3] for r in rows), sum(r[7] for r in rows)))
    if out:
        open(out, "w").write(txt)
    print(txt[-400:] if out else txt)


if __name__ == "__main__":
    main()
