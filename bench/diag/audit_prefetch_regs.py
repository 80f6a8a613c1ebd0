"""Audit of the prefetching convolution kernels' machine code: the registers that the inline-assembly epilogue requests
write must not be read or written by any other instruction between the request and the wait that names them (the
compiler treats an assembly statement's outputs as written when the statement ends, so a register copy or a spill it
places there would move bytes that have not arrived).  Usage:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -I <csrc> -S --offload-device-only <csrc>/conv_dma.hip -o /tmp/conv_dma.s
    python bench/diag/audit_prefetch_regs.py /tmp/conv_dma.s

For every kernel whose last template argument (PF) is true it walks the text from each group of requests to the next
named wait (the requests and their wait are on a straight path by construction: see the tile loop) and reports every
instruction in between that mentions a destination register.  Exit status 1 if there is one.
"""
import re
import sys


def regs_of(tok: str):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def main(path: str) -> int:
    lines = open(path).read().split("\n")
    starts = [(i, ln.split(":")[0]) for i, ln in enumerate(lines) if re.match(r"^_Z\d+conv_dma_kernel\S*:", ln)]
    bad = 0
    checked = 0
    for i, name in starts:
        if not name.endswith("Lb1EEv13ConvDmaParams"):
            continue
        end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm"))
        body = lines[i:end]
        blocks, cur = [], None
        for k, ln in enumerate(body):
            t = ln.strip()
            if t.startswith(";;#ASMSTART"):
                cur = [k, []]
            elif t.startswith(";;#ASMEND"):
                blocks.append((cur[0], k, cur[1]))
                cur = None
            elif cur is not None:
                cur[1].append(t)
        k = 0
        while k < len(blocks):
            if not any(x.startswith("buffer_load") for x in blocks[k][2]):
                k += 1
                continue
            group = []
            while k < len(blocks) and any(x.startswith("buffer_load") for x in blocks[k][2]):
                group.append(blocks[k])
                k += 1
            dst = set()
            for b in group:
                for x in b[2]:
                    if x.startswith("buffer_load"):
                        dst |= regs_of(x.split()[1].rstrip(","))
            # the named wait: the next assembly block that is a lone s_waitcnt vmcnt -- in text order, or, when the
            # compiler laid the loop's last block out ahead of the requests, the last one before them; then the path is
            # (requests .. end of the kernel text) + (label that opens the wait's block .. wait)
            is_wait = lambda b: len(b[2]) == 1 and b[2][0].startswith("s_waitcnt vmcnt")
            wait = next((b for b in blocks[k:] if is_wait(b)), None)
            spans = None
            if wait is not None:
                spans = [(group[-1][1], wait[0])]
            else:
                before = [b for b in blocks if is_wait(b) and b[0] < group[0][0] and b[2][0] != "s_waitcnt vmcnt(0)"]
                if before:
                    wait = before[-1]
                    opener = max(q for q in range(wait[0]) if re.match(r"^\.LBB", body[q]))
                    spans = [(group[-1][1], len(body)), (opener, wait[0])]
            if wait is None:
                print(f"{name}: requests at line {group[0][0]} have no wait behind them")
                bad += 1
                continue
            checked += 1
            hits = []
            for a, b in spans:
                inasm = False
                for q in range(a, b):
                    t = body[q].strip()
                    if t.startswith(";;#ASMSTART"):
                        inasm = True
                    elif t.startswith(";;#ASMEND"):
                        inasm = False
                    elif not inasm:
                        t = t.split(";")[0]
                        if t and not t.startswith(".") and regs_of(t) & dst:
                            hits.append((q, t))
            labels = sum(1 for q in range(group[-1][1], wait[0]) if re.match(r"^\.LBB", body[q]))
            print(f"{name[18:62]}: requests at {group[0][0]}..{group[-1][1]}, wait '{wait[2][0]}' at {wait[0]}, "
                  f"{len(dst)} registers, {wait[0] - group[-1][1]} lines / {labels} labels between, {len(hits)} touched")
            for q, t in hits[:20]:
                print("     !!", q, t)
            bad += len(hits)
    print(f"{checked} request groups checked, {bad} problems")
    return 1 if bad or not checked else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
