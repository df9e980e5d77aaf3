"""Static check of the hand-scheduled kernels: no instruction may touch the destination of an LDS read before the
s_waitcnt that retires it.

The kernels issue their operand reads through inline asm so that the compiler's own waitcnt insertion does not see
them.  The price: the compiler believes those registers hold their new value as soon as the asm
statement has been issued, and when it decides to keep such a value somewhere else (v_accvgpr_write / v_mov of a split
live range) it copies a register whose read is still in flight -- the copy holds the old contents.  Whether that
happens depends on register allocation, i.e. on unrelated edits, so it is checked on the ISA that is shipped:

  * the control-flow graph of every kernel is rebuilt from the labels and branches of the compiler's .s output;
  * the in-order LDS return queue (ds_* instructions; s_waitcnt lgkmcnt(N) retires all but the N youngest) is
    propagated along every path until the set of (block, queue) states is closed;
  * any instruction that reads or writes a register still waiting for its ds_read is reported.

Scalar-memory loads share the counter but return out of order; the compiler only ever waits for them with
lgkmcnt(0), which empties the queue in this model as well, so they are modelled as queue slots without a register.

Usage: hipcc ... -S --cuda-device-only file.hip -o file.s ; python scripts/check_asm_hazard.py file.s [...]
Exit status 1 if a hazard is found.  scripts/scan_kernels.sh runs it over every kernel translation unit.
"""
import re
import sys

REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")
LABEL = re.compile(r"^([A-Za-z_.$][\w.$]*):")
MAX_QUEUE = 64


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out |= {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    return out


def functions(path):
    """Yields (name, [(line_no, text)]) for every function body of the file."""
    name, body = None, []
    for no, raw in enumerate(open(path).read().split("\n"), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";") else ""
        if not line.strip():
            continue
        m = LABEL.match(line)
        if m and not m.group(1).startswith(".L"):
            if name and body:
                yield name, body
            name, body = m.group(1), []
            continue
        s = line.strip()
        if s.startswith(".") and not LABEL.match(s):
            if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".text"):
                if name and body:
                    yield name, body
                name, body = None, []
            continue
        if name is not None:
            body.append((no, s))
    if name and body:
        yield name, body


def blocks_of(body):
    """Splits a function body into basic blocks; returns (blocks, successors)."""
    blocks, cur, label_at = [], [], {}
    for no, s in body:
        m = LABEL.match(s)
        if m:
            if cur:
                blocks.append(cur)
                cur = []
            label_at[m.group(1)] = len(blocks)
            continue
        cur.append((no, s))
        op = s.split()[0]
        if op.startswith("s_branch") or op.startswith("s_cbranch") or op == "s_endpgm":
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        no, s = b[-1]
        parts = s.split()
        op = parts[0]
        if op == "s_endpgm":
            succ.append([])
        elif op.startswith("s_branch"):
            succ.append([label_at[parts[1]]])
        elif op.startswith("s_cbranch"):
            succ.append([label_at[parts[-1]]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])
    return blocks, succ


def run_block(block, queue, report):
    queue = list(queue)
    for no, s in block:
        parts = s.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ""
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                del queue[: max(0, len(queue) - int(m.group(1)))]
            continue
        # (a ds_read over a pending destination is ordered by the in-order return; only its address is a use)
        touched = regs(rest.split(",", 1)[1] if op.startswith("ds_read") and "," in rest else rest)
        pend = set().union(*queue) if queue else set()
        if touched & pend:
            report(no, s)
        if op.startswith("ds_read"):
            queue.append(frozenset(regs(rest.split(",")[0])))
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append(frozenset())
        if len(queue) > MAX_QUEUE:                   # the hardware counter saturates long before this
            del queue[: len(queue) - MAX_QUEUE]
    return tuple(queue)


def scan(path):
    bad = {}
    for name, body in functions(path):
        blocks, succ = blocks_of(body)
        if not blocks:
            continue
        seen, work = set(), [(0, ())]
        while work:
            state = work.pop()
            if state in seen:
                continue
            seen.add(state)
            b, queue = state
            out = run_block(blocks[b], queue, lambda no, s: bad.setdefault((name, no), s))
            for nb in succ[b]:
                if (nb, out) not in seen:
                    work.append((nb, out))
    return [(k[0], k[1], s) for k, s in sorted(bad.items(), key=lambda kv: kv[0][1])]


if __name__ == "__main__":
    found = []
    for p in sys.argv[1:]:
        found += scan(p)
    for k, ln, l in found[:40]:
        print(f"HAZARD {k}:{ln}: {l}")
    print(f"{len(found)} hazardous accesses in {len(sys.argv) - 1} file(s)")
    sys.exit(1 if found else 0)
