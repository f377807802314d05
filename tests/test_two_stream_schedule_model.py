"""Race check of the two-stream launch schedules (Qwen-Image blocks, FLUX.1 double blocks: csrc/qwen_engine.hip / flux_engine.hip
`forward_core`, tune keys 12 / 14) on a MODEL of the HIP stream semantics -- these schedules were written without a GPU at hand.

Every launch is (stream, buffers read, buffers written); launches of one stream are ordered; `record(ev, s)` / `wait(s, ev)` order
everything enqueued on the recording stream before the record ahead of everything enqueued on the waiting stream after the wait.  The
check: any two launches that touch the same buffer region, at least one of them writing, must be ordered by happens-before.  The launch
lists below transcribe the C++ (same order, same buffers; text and image rows of q / k / vT are separate regions); two consecutive forwards
are modelled so that the wrap-around (next forward's copy into `c`, next block 0 against this forward's tail) is covered, and the checker
itself is validated on schedules with a known missing edge.
"""
import itertools

import pytest


class Sched:
    def __init__(self):
        self.ops = []            # (stream, name, reads, writes)
        self.edges = []          # (op index a, op index b): a happens before b
        self.last = {}           # stream -> index of its last op
        self.events = {}         # event -> op index it was recorded after (None = nothing before it on that stream)

    def launch(self, stream, name, reads=(), writes=()):
        i = len(self.ops)
        self.ops.append((stream, name, frozenset(reads), frozenset(writes)))
        if stream in self.last:
            self.edges.append((self.last[stream], i))
        self.last[stream] = i
        return i

    def record(self, ev, stream):
        self.events[ev] = self.last.get(stream)

    def wait(self, stream, ev):
        src = self.events[ev]
        if src is None:
            return
        # a wait is a no-op launch on the waiting stream that depends on the recorded position
        i = self.launch(stream, f"wait({ev})")
        self.edges.append((src, i))

    def races(self):
        n = len(self.ops)
        reach = [set() for _ in range(n)]
        succ = [[] for _ in range(n)]
        for a, b in self.edges:
            succ[a].append(b)
        for i in reversed(range(n)):                      # ops are appended in a topological order (edges go forward)
            for j in succ[i]:
                reach[i].add(j)
                reach[i] |= reach[j]
        out = []
        for i, j in itertools.combinations(range(n), 2):
            si, ni, ri, wi = self.ops[i]
            sj, nj, rj, wj = self.ops[j]
            if (wi & (rj | wj)) or (wj & ri):
                if j not in reach[i] and i not in reach[j]:
                    out.append((ni, nj, sorted((wi & (rj | wj)) | (wj & ri))))
        return out


def qwen_forward(s, L, two, tag, skip=()):
    """csrc/qwen_engine.hip forward_core: image chain on 'st', text chain on 'ts' (= 'st' when single-stream)."""
    st = "st"
    ts = "side" if two else "st"
    qkb_c, big_c = ("qkbuf_c", "big_c") if two else ("qkbuf", "big")
    s.launch(st, f"{tag}:img_in", ["lat", "w"], ["x"])
    s.launch(st, f"{tag}:c<-c0", ["c0"], ["c"])
    if two and "fork_start" not in skip:
        s.record(f"{tag}:fork[L]", st)
        s.wait(ts, f"{tag}:fork[L]")
    for i in range(L):
        s.launch(ts, f"{tag}:{i}:ln_c", ["c", "mod"], ["cn"])
        s.launch(ts, f"{tag}:{i}:qk_c", ["cn", "w"], [qkb_c])
        s.launch(ts, f"{tag}:{i}:rope_c", [qkb_c], ["q.txt", "k.txt"])
        s.launch(ts, f"{tag}:{i}:vT_c", ["cn", "w"], ["vT.txt"])
        s.launch(st, f"{tag}:{i}:ln_x", ["x", "mod"], ["xn"])
        s.launch(st, f"{tag}:{i}:qk_x", ["xn", "w"], ["qkbuf"])
        s.launch(st, f"{tag}:{i}:rope_x", ["qkbuf"], ["q.img", "k.img"])
        s.launch(st, f"{tag}:{i}:vT_x", ["xn", "w"], ["vT.img"])
        if two and "join" not in skip:
            s.record(f"{tag}:join[{i}]", ts)
            s.wait(st, f"{tag}:join[{i}]")
        s.launch(st, f"{tag}:{i}:attn", ["q.img", "q.txt", "k.img", "k.txt", "vT.img", "vT.txt", "kvlen"], ["o_img", "o_ctx"])
        if two and "fork" not in skip:
            s.record(f"{tag}:fork[{i}]", st)
            s.wait(ts, f"{tag}:fork[{i}]")
        s.launch(st, f"{tag}:{i}:out_x", ["o_img", "w", "mod", "x"], ["x"])
        s.launch(ts, f"{tag}:{i}:out_c", ["o_ctx", "w", "mod", "c"], ["c"])
        s.launch(st, f"{tag}:{i}:ln2_x", ["x", "mod"], ["xn"])
        s.launch(st, f"{tag}:{i}:ff1_x", ["xn", "w"], ["big"])
        s.launch(st, f"{tag}:{i}:ff2_x", ["big", "w", "mod", "x"], ["x"])
        if i + 1 < L:
            s.launch(ts, f"{tag}:{i}:ln2_c", ["c", "mod"], ["cn"])
            s.launch(ts, f"{tag}:{i}:ff1_c", ["cn", "w"], [big_c])
            s.launch(ts, f"{tag}:{i}:ff2_c", [big_c, "w", "mod", "c"], ["c"])
    if two and "join_end" not in skip:
        s.record(f"{tag}:join[L]", ts)
        s.wait(st, f"{tag}:join[L]")
    s.launch(st, f"{tag}:ln_out", ["x", "mod"], ["xn"])
    s.launch(st, f"{tag}:proj_out", ["xn", "w"], ["v2"])
    # between forwards (rollout loop): CFG combine + scheduler step on `st`, then the next forward's conditioning rows are already in `mod`
    s.launch(st, f"{tag}:combine+sde", ["v2", "lat"], ["lat"])


def flux_forward(s, L, LS, two, tag, skip=()):
    """csrc/flux_engine.hip forward_core: double blocks (joint order [text | image]) then single blocks on the concatenated stream."""
    st = "st"
    ts = "side" if two else "st"
    qkb_c, big_c = ("qkbuf_c", "big_c") if two else ("qkbuf", "big")
    s.launch(st, f"{tag}:x_embed", ["lat", "w"], ["x"])
    s.launch(st, f"{tag}:c<-c0", ["c0"], ["c"])
    if two and "fork_start" not in skip:
        s.record(f"{tag}:fork[L]", st)
        s.wait(ts, f"{tag}:fork[L]")
    for i in range(L):
        s.launch(ts, f"{tag}:{i}:ln_c", ["c", "mod"], ["cn"])
        s.launch(ts, f"{tag}:{i}:qk_c", ["cn", "w"], [qkb_c])
        s.launch(ts, f"{tag}:{i}:rope_c", [qkb_c], ["q.txt", "k.txt"])
        s.launch(ts, f"{tag}:{i}:vT_c", ["cn", "w"], ["vT.txt"])
        s.launch(st, f"{tag}:{i}:ln_x", ["x", "mod"], ["xn"])
        s.launch(st, f"{tag}:{i}:qk_x", ["xn", "w"], ["qkbuf"])
        s.launch(st, f"{tag}:{i}:rope_x", ["qkbuf"], ["q.img", "k.img"])
        s.launch(st, f"{tag}:{i}:vT_x", ["xn", "w"], ["vT.img"])
        if two and "join" not in skip:
            s.record(f"{tag}:join[{i}]", ts)
            s.wait(st, f"{tag}:join[{i}]")
        s.launch(st, f"{tag}:{i}:attn", ["q.img", "q.txt", "k.img", "k.txt", "vT.img", "vT.txt"], ["o_img", "o_ctx"])
        if two and "fork" not in skip:
            s.record(f"{tag}:fork[{i}]", st)
            s.wait(ts, f"{tag}:fork[{i}]")
        s.launch(st, f"{tag}:{i}:out_x", ["o_img", "w", "mod", "x"], ["x"])
        s.launch(ts, f"{tag}:{i}:out_c", ["o_ctx", "w", "mod", "c"], ["c"])
        s.launch(st, f"{tag}:{i}:ln2_x", ["x", "mod"], ["xn"])
        s.launch(st, f"{tag}:{i}:ff1_x", ["xn", "w"], ["big"])
        s.launch(st, f"{tag}:{i}:ff2_x", ["big", "w", "mod", "x"], ["x"])
        s.launch(ts, f"{tag}:{i}:ln2_c", ["c", "mod"], ["cn"])
        s.launch(ts, f"{tag}:{i}:ff1_c", ["cn", "w"], [big_c])
        s.launch(ts, f"{tag}:{i}:ff2_c", [big_c, "w", "mod", "c"], ["c"])
    if two and "join_end" not in skip:
        s.record(f"{tag}:join[L]", ts)
        s.wait(st, f"{tag}:join[L]")
    s.launch(st, f"{tag}:y<-c", ["c"], ["y.txt"])
    s.launch(st, f"{tag}:y<-x", ["x"], ["y.img"])
    for i in range(LS):
        s.launch(st, f"{tag}:s{i}:ln", ["y.txt", "y.img", "mod"], ["yn"])
        s.launch(st, f"{tag}:s{i}:qk", ["yn", "w"], ["qkbuf"])
        s.launch(st, f"{tag}:s{i}:rope", ["qkbuf"], ["q.img", "q.txt", "k.img", "k.txt"])
        s.launch(st, f"{tag}:s{i}:vT", ["yn", "w"], ["vT.img", "vT.txt"])
        s.launch(st, f"{tag}:s{i}:mlp", ["yn", "w"], ["big"])
        s.launch(st, f"{tag}:s{i}:attn", ["q.img", "q.txt", "k.img", "k.txt", "vT.img", "vT.txt"], ["big"])
        s.launch(st, f"{tag}:s{i}:out", ["big", "w", "mod", "y.txt", "y.img"], ["y.txt", "y.img"])
    s.launch(st, f"{tag}:x<-y", ["y.img"], ["x"])
    s.launch(st, f"{tag}:ln_out", ["x", "mod"], ["xn"])
    s.launch(st, f"{tag}:proj_out", ["xn", "w"], ["v"])
    s.launch(st, f"{tag}:sde", ["v", "lat"], ["lat"])


def _prepare(s):
    # mi355_*_rollout before the loop: staged inputs, prompt preparation (c0, key lengths), the modulation table of all steps -- on `st`
    s.launch("st", "prepare", ["w"], ["lat", "c0", "kvlen", "mod"])


@pytest.mark.parametrize("two", [False, True])
def test_qwen_schedule_has_no_race(two):
    s = Sched()
    _prepare(s)
    for f in range(2):                       # two consecutive forwards of a rollout
        qwen_forward(s, L=3, two=two, tag=f"f{f}")
    assert s.races() == []


@pytest.mark.parametrize("two", [False, True])
def test_flux_schedule_has_no_race(two):
    s = Sched()
    _prepare(s)
    for f in range(2):
        flux_forward(s, L=3, LS=2, two=two, tag=f"f{f}")
    assert s.races() == []


@pytest.mark.parametrize("family", ["qwen", "flux"])
@pytest.mark.parametrize("missing", ["fork_start", "join", "fork", "join_end"])
def test_the_checker_sees_every_edge_that_is_needed(family, missing):
    """Each of the four kinds of edges is load-bearing: without it the model reports a race (so the clean result above means something)."""
    s = Sched()
    _prepare(s)
    for f in range(2):
        if family == "qwen":
            qwen_forward(s, L=3, two=True, tag=f"f{f}", skip=(missing,))
        else:
            flux_forward(s, L=3, LS=2, two=True, tag=f"f{f}", skip=(missing,))
    assert s.races() != []


def sd3_forward(s, L, dual, two, late, three, tag, skip=()):
    """csrc/engine.hip forward_core (the SHIPPED default is two = True; `late` above 16 384 image rows; `three` is opt-in): image chain on
    'st', text chain on 'ts', the image V^T / dual-attention projections on 'vs' in the three-stream variant.  `dual` = indices of the
    dual-attention blocks; the last block has no text out-projection / MLP (context_pre_only)."""
    st = "st"
    ts = "side" if two else "st"
    three = three and two
    vs = "side_v" if three else "st"
    s.launch(st, f"{tag}:patch_embed", ["lat", "w", "pe"], ["patches", "x"])
    s.launch(st, f"{tag}:c<-c0", ["c0"], ["c"])
    if two and "fork_start" not in skip:
        s.record(f"{tag}:fork[L]", st)
        s.wait(ts, f"{tag}:fork[L]")
    text_open = False
    for i in range(L):
        last, is_dual = i == L - 1, i in dual
        s.launch(st, f"{tag}:{i}:ln_x", ["x", "mod"], ["xn"] + (["xn2"] if is_dual else []))
        s.launch(ts, f"{tag}:{i}:ln_c", ["c", "mod"], ["cn"])
        if three:
            s.record(f"{tag}:vfork[{i}]", st)
            s.wait(vs, f"{tag}:vfork[{i}]")
        s.launch(st, f"{tag}:{i}:qk_x", ["xn", "w"], ["q.img", "k.img"])
        s.launch(vs, f"{tag}:{i}:vT_x", ["xn", "w"], ["vT.img"])
        s.launch(ts, f"{tag}:{i}:qk_c", ["cn", "w"], ["q.txt", "k.txt"])
        s.launch(ts, f"{tag}:{i}:vT_c", ["cn", "w"], ["vT.txt"])
        if three:
            s.record(f"{tag}:vjoin[{i}]", vs)
            s.wait(st, f"{tag}:vjoin[{i}]")
            if is_dual:
                s.launch(vs, f"{tag}:{i}:qk2", ["xn2", "w"], ["q2", "k2"])
                s.launch(vs, f"{tag}:{i}:vT2", ["xn2", "w"], ["vT2"])
                s.record(f"{tag}:djoin[{i}]", vs)
        if two and "join" not in skip:
            s.record(f"{tag}:join[{i}]", ts)
            s.wait(st, f"{tag}:join[{i}]")
            text_open = False
        s.launch(st, f"{tag}:{i}:attn", ["q.img", "q.txt", "k.img", "k.txt", "vT.img", "vT.txt"], ["o_img", "o_ctx"])
        fork_here = two and (not last or i + 1 < L)

        def fork():
            nonlocal text_open
            if "fork" not in skip:
                s.record(f"{tag}:fork[{i}]", st)
                s.wait(ts, f"{tag}:fork[{i}]")
            text_open = True
        is_late = is_dual and late
        if fork_here and not is_late:
            fork()
        s.launch(st, f"{tag}:{i}:out_x", ["o_img", "w", "mod", "x"], ["x"])
        if not two and not last:
            s.launch(st, f"{tag}:{i}:out_c", ["o_ctx", "w", "mod", "c"], ["c"])
        if is_dual:
            if three:
                s.wait(st, f"{tag}:djoin[{i}]")
            else:
                s.launch(st, f"{tag}:{i}:qk2", ["xn2", "w"], ["q2", "k2"])
                s.launch(st, f"{tag}:{i}:vT2", ["xn2", "w"], ["vT2"])
            s.launch(st, f"{tag}:{i}:attn2", ["q2", "k2", "vT2"], ["o_img"])          # S == n_img: never writes o_ctx
            if fork_here and is_late:
                fork()
            s.launch(st, f"{tag}:{i}:out2_x", ["o_img", "w", "mod", "x"], ["x"])
        if two and not last:
            s.launch(ts, f"{tag}:{i}:out_c", ["o_ctx", "w", "mod", "c"], ["c"])
        s.launch(st, f"{tag}:{i}:ln2_x", ["x", "mod"], ["xn"])
        s.launch(st, f"{tag}:{i}:ff1_x", ["xn", "w"], ["hid"])
        s.launch(st, f"{tag}:{i}:ff2_x", ["hid", "w", "mod", "x"], ["x"])
        if not last:
            s.launch(ts, f"{tag}:{i}:ln2_c", ["c", "mod"], ["cn"])
            s.launch(ts, f"{tag}:{i}:ff1_c", ["cn", "w"], ["chid"])
            s.launch(ts, f"{tag}:{i}:ff2_c", ["chid", "w", "mod", "c"], ["c"])
    if two and text_open and "join_end" not in skip:
        s.record(f"{tag}:join[L]", ts)
        s.wait(st, f"{tag}:join[L]")
    s.launch(st, f"{tag}:ln_out", ["x", "mod"], ["xn"])
    s.launch(st, f"{tag}:proj_out", ["xn", "w"], ["v"])
    s.launch(st, f"{tag}:cfg+sde", ["v", "lat"], ["lat"])


@pytest.mark.parametrize("two,late,three", [(False, False, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True)])
def test_sd3_schedule_has_no_race(two, late, three):
    """The SHIPPED SD3.5 schedule (bit-identity was measured on the GPU, which cannot prove the absence of a race): single stream, two
    streams with the early / late fork, and the opt-in three-stream variant -- 4 blocks, the first two with dual attention, two forwards."""
    s = Sched()
    _prepare(s)
    for f in range(2):
        sd3_forward(s, L=4, dual=(0, 1), two=two, late=late, three=three, tag=f"f{f}")
    assert s.races() == []


@pytest.mark.parametrize("missing", ["fork_start", "join", "fork"])
def test_sd3_checker_sensitivity(missing):
    s = Sched()
    _prepare(s)
    for f in range(2):
        sd3_forward(s, L=4, dual=(0, 1), two=True, late=True, three=False, tag=f"f{f}", skip=(missing,))
    assert s.races() != []
