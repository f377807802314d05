"""Happens-before race checker for the engines' launch schedules (test infrastructure).

Input: the text the library emits under `mi355_sched_trace(1)` (csrc/sched_trace.hip): one line per kernel launch with the byte regions it
reads and writes, one per event record, one per stream wait.  Model of the HIP semantics: launches of one stream execute in order;
`record(ev, s)` marks the position of stream s, `wait(s', ev)` makes everything enqueued on s' afterwards run after that position.  Check:
any two launches that touch overlapping bytes, at least one of them writing, must be ordered by the transitive closure of those edges.

A region is `count` blocks of `bytes`, `stride` apart (count 1: one interval): the q / k / V^T scatter epilogues write the image rows and
the text rows of one buffer from different streams, so overlap is decided on the exact block sets, not on bounding boxes.
"""
from __future__ import annotations

import bisect
from typing import Dict, List, Optional, Sequence, Tuple


class Region:
    __slots__ = ("ptr", "len", "stride", "count", "write", "_iv")

    def __init__(self, ptr: int, length: int, stride: int, count: int, write: bool):
        self.ptr, self.len, self.stride, self.count, self.write = ptr, length, stride if count > 1 else 0, max(count, 1), write
        self._iv = None

    @property
    def lo(self) -> int:
        return self.ptr

    @property
    def hi(self) -> int:
        return self.ptr + (self.count - 1) * self.stride + self.len

    def intervals(self) -> List[Tuple[int, int]]:
        if self._iv is None:
            if self.count == 1 or self.stride <= self.len:          # contiguous (or self-overlapping) blocks: one interval
                self._iv = [(self.lo, self.hi)]
            else:
                self._iv = [(self.ptr + i * self.stride, self.ptr + i * self.stride + self.len) for i in range(self.count)]
        return self._iv

    def overlaps(self, other: "Region") -> bool:
        if self.hi <= other.lo or other.hi <= self.lo:
            return False
        a, b = self.intervals(), other.intervals()
        if len(a) == 1 and len(b) == 1:
            return True
        if len(a) > len(b):
            a, b = b, a
        starts = [s for s, _ in b]
        for s, e in a:                                              # b is sorted by construction
            i = bisect.bisect_right(starts, s) - 1
            if i >= 0 and b[i][1] > s:
                return True
            if i + 1 < len(b) and b[i + 1][0] < e:
                return True
        return False


class Op:
    __slots__ = ("stream", "name", "regions")

    def __init__(self, stream: str, name: str, regions: List[Region]):
        self.stream, self.name, self.regions = stream, name, regions


class Schedule:
    def __init__(self):
        self.ops: List[Op] = []
        self.edges: List[Tuple[int, int]] = []
        self.last: Dict[str, int] = {}
        self.events: Dict[str, Optional[int]] = {}
        self.waits: List[Tuple[str, str]] = []           # (stream, event) of every wait, in order (for the mutation tests)

    def launch(self, stream: str, name: str, regions: Sequence[Region] = ()) -> int:
        i = len(self.ops)
        self.ops.append(Op(stream, name, list(regions)))
        if stream in self.last:
            self.edges.append((self.last[stream], i))
        self.last[stream] = i
        return i

    def record(self, ev: str, stream: str) -> None:
        self.events[ev] = self.last.get(stream)

    def wait(self, stream: str, ev: str) -> None:
        self.waits.append((stream, ev))
        src = self.events.get(ev)
        if src is None:
            return
        i = self.launch(stream, f"wait({ev})")
        self.edges.append((src, i))

    def streams(self) -> List[str]:
        return sorted({o.stream for o in self.ops if o.regions})

    def races(self, limit: int = 20) -> List[Tuple[str, str, str]]:
        n = len(self.ops)
        succ: List[List[int]] = [[] for _ in range(n)]
        for a, b in self.edges:
            succ[a].append(b)
        reach = [0] * n                                  # bitsets: ops are appended in a topological order (every edge goes forward)
        for i in reversed(range(n)):
            r = 0
            for j in succ[i]:
                r |= reach[j] | (1 << j)
            reach[i] = r
        # sweep over bounding boxes: candidate pairs only where two regions' boxes overlap
        items = []
        for i, op in enumerate(self.ops):
            for r in op.regions:
                items.append((r.lo, r.hi, i, r))
        items.sort(key=lambda t: t[0])
        out, seen = [], set()
        active: List[Tuple[int, int, Region]] = []       # (hi, op, region)
        for lo, hi, i, r in items:
            active = [a for a in active if a[0] > lo]
            for _, j, q in active:
                if i == j or not (r.write or q.write):
                    continue
                a, b = (i, j) if i < j else (j, i)
                if (a, b) in seen or (reach[a] >> b) & 1:
                    continue
                if r.overlaps(q):
                    seen.add((a, b))
                    out.append((self.ops[a].name + "@" + self.ops[a].stream, self.ops[b].name + "@" + self.ops[b].stream, hex(max(r.lo, q.lo))))
                    if len(out) >= limit:
                        return out
            active.append((hi, i, r))
        return out


def parse(text: str, drop_waits: Sequence[int] = ()) -> Schedule:
    """Trace text -> Schedule.  `drop_waits`: indices (in trace order) of stream waits to ignore -- the mutation that must make races appear."""
    s = Schedule()
    names: Dict[str, str] = {}
    n_wait = 0
    for line in text.splitlines():
        f = line.split()
        if not f:
            continue
        if f[0] == "L":
            st = names.setdefault(f[1], f"s{len(names)}")
            regs = []
            for tok in f[3:]:
                kind, ptr, ln, stride, count = tok.split(":")
                regs.append(Region(int(ptr, 16), int(ln), int(stride), int(count), kind == "W"))
            s.launch(st, f[2], regs)
        elif f[0] == "E":
            s.record(f[2], names.setdefault(f[1], f"s{len(names)}"))
        elif f[0] == "T":
            if n_wait not in drop_waits:
                s.wait(names.setdefault(f[1], f"s{len(names)}"), f[2])
            else:
                s.waits.append((names.setdefault(f[1], f"s{len(names)}"), f[2]))
            n_wait += 1
    return s


def n_waits(text: str) -> int:
    return sum(1 for line in text.splitlines() if line.startswith("T "))
