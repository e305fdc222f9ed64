#!/usr/bin/env python3
"""Where does a kernel touch scratch (spilled VGPRs, private arrays)?  Compiles one translation unit of csrc/ to gfx950 assembly with line tables and, for the
chosen kernel, lists every LOOP of the code object (LLVM's "Loop Header / in Loop" block annotations) with its depth, the source lines it spans, its
instruction count and its scratch_load / scratch_store instructions -- the evidence VERDICT r5 asked for ("no spill traffic inside the hit / pair loops").
usage: tools/isa_scratch_report.py <unit.hip> <kernel name substring> [extra hipcc flags ...]   -> text on stdout (committed under profiles/)"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
src = os.path.join(ROOT, "stereo-visual-slam_amd", "csrc", unit)
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-gline-tables-only", "-S", "--cuda-device-only"]
    if unit in ("lm_kernels.hip", "ba_resident.hip", "geom_kernels.hip", "track_kernels.hip"):
        flags.append("-ffp-contract=fast")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + extra + [src, "-o", out])
    lines = open(out).read().splitlines()
# kernels: label line "name:" ... ".amdhsa_kernel name"
starts = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
meta = {}
cur = None
for l in lines:
    m = re.match(r"\s+- \.agpr_count:|\s+\.name:\s+(\S+)", l)
    if l.strip().startswith(".name:") and "_Z" in l:
        cur = l.split()[-1]; meta[cur] = {}
    for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
        if cur and l.strip().startswith(key + ":"):
            meta[cur][key] = l.split()[-1]
for si, (i0, name) in enumerate(starts):
    if pat not in name:
        continue
    i1 = next((j for j in range(i0, len(lines)) if lines[j].strip().startswith(".Lfunc_end")), len(lines))
    print("kernel %s" % name)
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    print("  = %s" % dem)
    print("  code object: " + "  ".join("%s %s" % (k, v) for k, v in meta.get(name, {}).items()))
    # blocks
    blocks = []   # (label, depth, header, [instr], [src lines])
    cur = {"label": "(entry)", "depth": 0, "hdr": None, "ins": [], "loc": []}
    file_names = {}
    for l in lines[i0:i1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\w+):\s*(;.*)?$", t)
        if m:
            blocks.append(cur)
            depth, hdr = 0, None
            c = m.group(2) or ""
            mm = re.search(r"Loop Header: Depth=(\d+)", c)
            if mm:
                depth, hdr = int(mm.group(1)), m.group(1)
            mm = re.search(r"in Loop: Header=(\w+) Depth=(\d+)", c)
            if mm:
                depth, hdr = int(mm.group(2)), ".L" + mm.group(1) if not mm.group(1).startswith(".L") else mm.group(1)
            cur = {"label": m.group(1), "depth": depth, "hdr": hdr, "ins": [], "loc": []}
            continue
        if t.startswith(";") and ("Loop" in t) and cur["ins"] == []:   # continuation comments of a block header (Parent Loop ... / Child Loop ...)
            mm = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", t)
            if mm:
                cur["depth"], cur["hdr"] = int(mm.group(1)), cur["label"]
            continue
        if t.startswith(".loc"):
            p = t.split()
            if len(p) >= 3 and p[2].isdigit() and int(p[2]) > 0:   # (line 0 = compiler-generated)
                cur["loc"].append(int(p[2]))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split()[0])
    blocks.append(cur)
    loops = collections.OrderedDict()
    for b in blocks:
        key = b["hdr"] if b["depth"] > 0 else "(straight-line code outside every loop)"
        L = loops.setdefault(key, {"depth": b["depth"], "n": 0, "ld": 0, "st": 0, "lo": 10 ** 9, "hi": 0, "valu": 0})
        L["depth"] = max(L["depth"], b["depth"]) if key.startswith("(") else b["depth"]
        L["n"] += len(b["ins"])
        L["ld"] += sum(1 for x in b["ins"] if x.startswith("scratch_load"))
        L["st"] += sum(1 for x in b["ins"] if x.startswith("scratch_store"))
        L["valu"] += sum(1 for x in b["ins"] if x.startswith("v_"))
        if b["loc"]:
            L["lo"] = min(L["lo"], min(b["loc"])); L["hi"] = max(L["hi"], max(b["loc"]))
    tot_ld = sum(L["ld"] for L in loops.values()); tot_st = sum(L["st"] for L in loops.values())
    print("  %d instructions, %d scratch_load + %d scratch_store in the whole kernel; per loop (blocks are attributed to their INNERMOST loop):" % (
        sum(L["n"] for L in loops.values()), tot_ld, tot_st))
    print("  %-14s %5s  %-13s %7s %7s %8s %8s" % ("loop header", "depth", "source lines", "instrs", "VALU", "scr.load", "scr.store"))
    for k, L in loops.items():
        print("  %-14s %5d  %-13s %7d %7d %8d %8d%s" % (k[:14] if not k.startswith("(") else "(no loop)", L["depth"], ("%d-%d" % (L["lo"], L["hi"])) if L["hi"] else "-", L["n"], L["valu"],
                                                  L["ld"], L["st"], "   <-- scratch inside a depth >= 2 loop" if L["depth"] >= 2 and (L["ld"] + L["st"]) else ""))
    deep = [(k, L) for k, L in loops.items() if L["depth"] >= 2]
    print("  loops of depth >= 2: %d, with scratch traffic: %d (%d loads, %d stores)" % (len(deep), sum(1 for _, L in deep if L["ld"] + L["st"]),
                                                                                       sum(L["ld"] for _, L in deep), sum(L["st"] for _, L in deep)))
    print()
