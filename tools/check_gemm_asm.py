#!/usr/bin/env python3
"""Static check of the hand-scheduled GEMM loops in a hipcc -S dump: between a kernel's first s_barrier and the
`s_nop 15` that closes the main loop, no instruction other than v_mfma may READ a register that an inline-asm
ds_read wrote (hipcc does not know that data lands late; a copy there would pick up stale data).
usage: check_gemm_asm.py gemm.s"""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def main(path):
    name, body, bad, nk = None, [], 0, 0
    def check(name, body):
        nonlocal bad, nk
        try:
            i0 = next(i for i, l in enumerate(body) if "s_barrier" in l)
            i1 = next(i for i, l in enumerate(body) if "s_nop 15" in l)
        except StopIteration:
            return
        nk += 1
        pending = set()   # registers written by ds_read since the last lgkmcnt wait
        for l in body[i0:i1]:
            t = l.strip()
            if t.startswith(".LBB"):  # basic-block boundary: control may arrive from a path with other live registers;
                pending = set()       # the hand-scheduled loops keep every asm load and its wait inside one block
                continue
            if not t or t.startswith(";") or t.startswith("."): continue
            op, _, rest = t.partition(" ")
            ops = [o.strip() for o in rest.split(",")]
            if op.startswith("ds_read"):
                pending |= regs(ops[0]); continue
            if op == "s_waitcnt" and "lgkmcnt(0)" in t:
                pending = set(); continue
            if op.startswith("v_mfma") or op.startswith("s_") or not ops: continue
            srcs = set()
            for o in ops[1:]: srcs |= regs(o.split(" ")[0])
            if op.startswith("global_load_lds"): srcs |= regs(ops[0])
            if srcs & pending:
                bad += 1
                print(f"{name}: `{t}` reads {sorted(srcs & pending)} before the LDS data has landed")
    for line in open(path):
        m = re.match(r"^(_ZN4klab\w*(gemm_glds_kernel|klab_lmhead_gemm|gemm_glds_fp8_kernel|gemm_glds_grouped_tn_kernel|gemm_glds_w8_kernel)\w*):", line)
        if m:
            if name: check(name, body)
            name, body = m.group(1), []
        elif name:
            body.append(line)
            if "s_endpgm" in line:
                check(name, body); name = None
    print(f"checked {nk} kernels, {bad} premature reads")
    bad2 = check_acc(path)
    return 1 if (bad or bad2) else 0


def check_acc(path):
    """second rule: an inline-asm MFMA's destination is not interlocked -- no other instruction may read it until the pipe has
    drained (the sources put `s_nop 15; s_nop 15` behind the last MFMA).  hipcc inserts register copies at control-flow joins; one
    placed between the last MFMA of a branch and the nops loses that MFMA's contribution (it happened: the last k-tile of every
    product, when the pipelined drain was first added as an if/else)."""
    src = open(path).read()
    bad = n = 0
    for m in re.finditer(r"^(_ZN4klab\w*(?:gemm_glds_kernel|klab_lmhead_gemm|gemm_glds_fp8_kernel|gemm_glds_grouped_tn_kernel|gemm_glds_w8_kernel)\w*):(.*?)s_endpgm", src, re.S | re.M):
        n += 1
        body = [l.strip() for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith((";", "."))]
        hot = {}   # register -> instructions since the MFMA that wrote it
        for l in body:
            op, _, rest = l.partition(" ")
            ops_ = [o.strip() for o in rest.split(",")]
            # age in (under-estimated) cycles: s_nop N = N + 1, an MFMA 16, anything else 4 (one wave64 VALU / LDS / scalar issue);
            # the sources put two `s_nop 15` (32 cycles) behind the last MFMA, which every parity test has validated -- flag reads younger than that
            if op.startswith("s_nop"):
                step = int(rest.strip() or 0) + 1
            elif op.startswith("v_mfma"):
                step = 16
            else:
                step = 4
            hot = {r: c + step for r, c in hot.items()}
            hot = {r: c for r, c in hot.items() if c < 32}
            if op.startswith("v_mfma"):
                for r in regs(ops_[0]):
                    hot[r] = 0
                continue
            if op.startswith("s_") or op.startswith("ds_read") or not ops_:
                continue
            srcs = set()
            for o in ops_[1:]:
                srcs |= regs(o.split(" ")[0])
            hit = srcs & set(hot)
            if hit:
                bad += 1
                print(f"{m.group(1)[:80]}: `{l}` reads MFMA results {sorted(hit)[:4]} ~{min(hot[r] for r in hit)} cycles after the MFMA")
                break
    print(f"checked {n} kernels for early accumulator reads, {bad} found")
    return bad

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
