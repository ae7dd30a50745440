#!/usr/bin/env python3
"""Static check of hand-counted `s_waitcnt vmcnt(N)` code (the asm-issued loads of qpg_gemm_dma_kernel and of the QW1_LEAD = 2 wide
GEMM): walks a kernel's gfx950 assembly along every path of bounded length, keeps the queue of outstanding VMEM operations (loads
return in order: `vmcnt(N)` retires all but the N most recent) and reports
  * an instruction that reads or writes a VGPR while a load into it is still in flight (the hardware does not interlock);
  * (with --lds) an s_barrier reached while an LDS DMA older than the allowed window may not have landed is NOT checked here -- the ring
    distance argument is in the kernel's comment; this tool checks registers only.
usage: check_isa_vmcnt.py file.s kernel_symbol_substring [max_steps]"""
import re
import sys

def parse(path, sym):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sym) + r"\S*:", l))
    body, labels = [], {}
    for l in lines[start + 1:]:
        t = l.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            labels[m.group(1)] = len(body)
            continue
        if t.startswith("."):
            continue
        body.append(t)
        if t.startswith("s_endpgm"):
            break
    return body, labels

def regs(tok):
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return []
    return [o.strip() for o in parts[1].split(",")]

def main():
    path, sym = sys.argv[1], sys.argv[2]
    max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
    body, labels = parse(path, sym)
    # depth-first over branch decisions, bounded: state = (pc, queue of (dest regs)), visited by (pc, queue signature, trips)
    bad, seen = [], set()
    stack = [(0, (), 0)]
    explored = 0
    while stack:
        pc, q, steps = stack.pop()
        while pc < len(body) and steps < max_steps:
            key = (pc, q)
            ins = body[pc]
            op = ins.split()[0]
            if op.startswith("s_cbranch") or op == "s_branch":
                if key in seen:
                    break
                seen.add(key)
                tgt = labels.get(operands(ins)[0])
                if op == "s_branch":
                    pc = tgt
                    continue
                if tgt is not None:
                    stack.append((tgt, q, steps))
                pc += 1
                continue
            ops = operands(ins)
            touched = set()
            for o in ops:
                touched |= regs(o.split()[0] if o else o)
            inflight = set().union(*[d for d in q]) if q else set()
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ins)
                if m:
                    n = int(m.group(1))
                    q = q[len(q) - n:] if n < len(q) else q
                    if n == 0:
                        q = ()
            elif op.startswith("global_load_lds") or op.startswith("buffer_load") and "lds" in ins:
                if touched & inflight:
                    bad.append((pc, ins, sorted(touched & inflight)))
                q = q + (frozenset(),)
            elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
                dest = regs(ops[0])
                addr = set().union(*[regs(o.split()[0]) for o in ops[1:]]) if len(ops) > 1 else set()
                if (addr | dest) & inflight:
                    bad.append((pc, ins, sorted((addr | dest) & inflight)))
                q = q + (frozenset(dest),)
            elif op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("global_atomic") or op.startswith("flat_store"):
                if touched & inflight:
                    bad.append((pc, ins, sorted(touched & inflight)))
                q = q + (frozenset(),)
            elif op == "s_endpgm":
                break
            else:
                if touched & inflight:
                    bad.append((pc, ins, sorted(touched & inflight)))
            pc += 1
            steps += 1
            explored += 1
    uniq = {}
    for pc, ins, r in bad:
        uniq.setdefault(pc, (ins, r))
    print(f"{sym}: {len(body)} instructions, {explored} steps explored, {len(uniq)} hazards")
    for pc in sorted(uniq)[:40]:
        print(f"  pc {pc}: {uniq[pc][0][:90]}   in flight: v{uniq[pc][1]}")
    return 1 if uniq else 0

if __name__ == "__main__":
    sys.exit(main())
