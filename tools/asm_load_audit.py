"""Dev tool (CPU): audit of the compiler-invisible loads of a kernel's ISA (cdna_hip_programming.md 5.7 item 1: "an asm load's VGPR
destination counts as written at ;;#ASMEND, so the compiler may read, copy, spill or reuse it before the data lands").

    hipcc --offload-arch=gfx950 -O3 -std=c++17 [flags] --cuda-device-only -S -o k.s -c clip-fsar_amd/csrc/gemm_vit.hip
    python tools/asm_load_audit.py k.s

For every `global_load_dword[x4] vN` INSIDE an asm statement (between ;;#ASMSTART / ;;#ASMEND) the listing is walked forward, along
fall-through order, until a `s_waitcnt vmcnt(0)` (or the function's end): every instruction that reads or writes the destination
register(s) before that wait is reported -- a compiler v_mov / spill / reuse there reads data that may not have landed (the last 16
lanes of a wave are written last).  `s_waitcnt vmcnt(N > 0)` is reported as "counted wait" with the number of VMEM instructions issued
since the load (the wait covers the load iff that many or more younger ones were issued)."""
import re, sys, collections

def regs_of(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()

def vregs(line):
    out = set()
    for t in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(t)
    return out

def main(path):
    lines = open(path).read().splitlines()
    fn = "?"
    in_asm = False
    findings = collections.Counter()
    detail = []
    n_loads = 0
    i = 0
    while i < len(lines):
        l = lines[i].strip()
        if l.endswith(":") and (l.startswith("_Z") or l.startswith("_ZN")) and not l.startswith(".L"):
            fn = l[:-1]
        if ";;#ASMSTART" in l:
            in_asm = True
        elif ";;#ASMEND" in l:
            in_asm = False
        elif in_asm and re.match(r"global_load_dword(x\d)? v", l):
            n_loads += 1
            dst = regs_of(l.split()[1].rstrip(","))
            vmem_since = 0
            j = i + 1
            asm2 = True
            covered = None
            while j < len(lines):
                t = lines[j].strip()
                if ";;#ASMSTART" in t:
                    asm2 = True
                elif ";;#ASMEND" in t:
                    asm2 = False
                elif t.startswith(".") or t.startswith(";") or not t or t.endswith(":"):
                    pass
                elif t.startswith("s_endpgm"):
                    covered = "end"
                    break
                else:
                    m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
                    if m:
                        n = int(m.group(1))
                        if n <= vmem_since:           # in-order returns: this load is among those waited for
                            covered = "vmcnt(%d) after %d younger VMEM" % (n, vmem_since)
                            break
                    if re.match(r"(global|buffer|scratch|flat)_(load|store)", t):
                        vmem_since += 1
                    touched = vregs(t) & dst
                    # benign: v_mad_u64_u32 used as a 32-bit multiply-add reads the HIGH half of its 64-bit addend as a don't-care
                    m64 = re.match(r"v_mad_[ui]64_[ui]32 v\[\d+:\d+\], \S+ \S+ \S+ v\[(\d+):(\d+)\]", t.replace(",", ", ").replace("  ", " "))
                    if m64 and touched == {int(m64.group(2))} and int(m64.group(2)) == int(m64.group(1)) + 1:
                        touched = set()
                    if touched and not re.match(r"global_load_dword(x\d)? v", t):
                        findings[fn] += 1
                        if len(detail) < 40:
                            detail.append((fn[-70:], i + 1, l, j + 1, t, "asm" if asm2 else "COMPILER"))
                j += 1
            if covered is None:
                findings[fn] += 1
                detail.append((fn[-70:], i + 1, l, -1, "no covering wait found", ""))
        i += 1
    main.last_count = n_loads
    print("%s: %d compiler-invisible VGPR loads audited" % (path, n_loads))
    if not findings:
        print("  no instruction touches a destination register between its load and the wait that covers it")
    for d in detail:
        print("  %s\n    line %d: %s\n    line %d: %s   [%s]" % d)
    return 1 if findings else 0

if __name__ == "__main__":
    sys.exit(max(main(p) for p in sys.argv[1:]))
