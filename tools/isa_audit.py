"""Static checks on gfx950 assembly of the tracker kernels (CPU only; no GPU run).

1. --hazard FILE.s     every SDWA instruction with dst_sel != DWORD (a partial register write) must not be followed directly by a
                       VALU instruction that reads the written register: on the gfx940 family the hardware does not interlock that
                       forwarding case and LLVM inserts the wait state only for SDWA instructions IT selected -- the shift in
                       lk.hip's shr_into_high_half is inline assembly, which the hazard recogniser cannot see into.  Used as a
                       build gate (visual-odometry_amd/build.py); exit status 1 when a site is found.
2. --spills FILE.s K   for kernel symbol K: every scratch (spill) store and reload with the stack of EXEC-narrowing regions
                       (s_and_saveexec ... s_or_b64 exec) it executes under, and a flag on every reload that runs under a region
                       stack that is NOT the store's stack or nested inside it (such a reload would hand lanes that were inactive
                       at the store whatever the scratch memory held).  Used for the study of the spill build (tools/diag_w4.sh,
                       DESIGN.md section 8).
"""
import re
import sys


def instructions(text):
    for no, ln in enumerate(text.split("\n"), 1):
        s = ln.strip()
        if not ln.startswith("\t") or not s or s[0] in ";." or s.startswith("//"):
            continue
        yield no, s.split(";")[0].strip()


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def hazard_sites(text):
    ins = list(instructions(text))
    sites = []
    for k, (no, s) in enumerate(ins[:-1]):
        if "_sdwa" not in s or "dst_sel:DWORD" in s or "dst_sel:" not in s:
            continue
        dst = regs_of(s.split(None, 1)[1].split(",")[0].strip())
        nno, nxt = ins[k + 1]
        if not nxt.startswith("v_") and not nxt.startswith("ds_") and not nxt.startswith("global_") and not nxt.startswith("scratch_"):
            continue                                   # SALU / waitcnt / branch in between: a wait state has passed
        ops = nxt.split(None, 1)[1] if " " in nxt else ""
        toks = [t.strip() for t in ops.split(",")]
        srcs = toks[1:]
        if re.match(r"v_(mac|fmac|dot2c|dot4c|dot8c)", nxt) or "_sdwa" in nxt or "dpp" in nxt:
            srcs = toks                                # the destination is read as well
        if any(dst & regs_of(t) for t in srcs):
            sites.append((no, s, nno, nxt))
    n_sdwa = sum(1 for _, s in ins if "_sdwa" in s and "dst_sel:" in s and "dst_sel:DWORD" not in s)
    return n_sdwa, sites


def kernel_text(text, symbol):
    start = text.index("\n%s:" % symbol)
    end = text.index(".end_amdhsa_kernel", start)
    return text[start:end]


def spill_report(text, symbol):
    body = kernel_text(text, symbol)
    stack, stores, out, bad = [], {}, [], 0
    for no, s in instructions(body):
        m = re.match(r"s_(and|or|andn2|xor)_saveexec_b64 (s\[\d+:\d+\])", s)
        if m:
            stack.append(m.group(2))
            continue
        m = re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\])", s)
        if m and m.group(1) in stack:
            while stack and stack.pop() != m.group(1):
                pass
            continue
        m = re.match(r"scratch_(store|load)_dword(x\d)? .*?off(?: offset:(\d+))?$", s)
        if m:
            off = int(m.group(3) or 0)
            n = {None: 1, "x2": 2, "x3": 3, "x4": 4}[m.group(2)]
            if m.group(1) == "store":
                for k in range(n):
                    stores.setdefault(off + 4 * k, []).append(tuple(stack))
                out.append("%5d  store  offset %3d  under %s" % (no, off, "/".join(stack) or "full EXEC"))
            else:
                st = [x for k in range(n) for x in stores.get(off + 4 * k, [])]
                ok = any(tuple(stack[:len(x)]) == x for x in st) if st else None
                flag = "" if ok else ("   <-- reload before any store (loop-carried or undefined on this path)" if ok is None
                                      else "   <-- reload under a WIDER region than every store seen so far")
                bad += ok is False
                out.append("%5d  reload offset %3d  under %s%s" % (no, off, "/".join(stack) or "full EXEC", flag))
    return out, bad


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--hazard":
        n, sites = hazard_sites(open(sys.argv[2]).read())
        print("%d partial-write SDWA instructions, %d followed directly by a reader of the written register" % (n, len(sites)))
        for no, s, nno, nxt in sites:
            print("  line %d: %s\n  line %d: %s" % (no, s, nno, nxt))
        sys.exit(1 if sites else 0)
    if len(sys.argv) >= 4 and sys.argv[1] == "--spills":
        out, bad = spill_report(open(sys.argv[2]).read(), sys.argv[3])
        print("\n".join(out))
        print("%d reloads under a wider EXEC region than their stores" % bad)
        sys.exit(0)
    print(__doc__)
    sys.exit(2)
