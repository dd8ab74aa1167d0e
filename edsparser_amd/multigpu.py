"""Column-range partition of one alignment over several GPUs: the boundary stitch.

Every rank transforms its own column slab as if it were a whole alignment (all the heavy work,
no communication).  A run of equal column type (common / variant) that crosses a slab boundary
is then repaired with KB-sized exchanges only:

  common run  — joined textually: the right slab drops its leading "{" and its "{0}", the left slab
                drops its trailing "}".
  variant run — its distinct strings are defined over the whole run (reference
                msa_transforms.cpp:262-293), so the left-most slab ("owner") gets the raw columns of
                the run from the others, transforms that mini-alignment, and replaces its last
                segment with the result; the others drop their first segment.

The planning below is pure host logic on the gathered edge descriptors and is backend-agnostic:
in production the slab transform and the mini-alignment transform run on the GPU through the C
ABI; the CPU tests (gloo, world_size 2) plug the oracle in instead.
"""
from dataclasses import dataclass, field


@dataclass
class SlabEdges:
    n_segments: int
    cols: int                    # slab width in alignment columns
    eds_bytes: int
    seds_bytes: int
    first_is_variant: int
    first_cols: int
    first_eds_bytes: int
    first_seds_bytes: int
    last_is_variant: int
    last_cols: int
    last_eds_bytes: int
    last_seds_bytes: int


@dataclass
class SlabAction:
    """What one rank does to its slab text: drop bytes at both ends, append the merged segment."""
    front_eds: int = 0
    front_seds: int = 0
    back_eds: int = 0
    back_seds: int = 0
    owns: list = field(default_factory=list)       # chains (index into plan.chains) this rank recomputes
    extra_eds: bytes = b""
    extra_seds: bytes = b""


@dataclass
class Chain:
    first: int                   # left-most slab (owner of a variant chain)
    last: int
    variant: bool


@dataclass
class StitchPlan:
    chains: list
    actions: list


def plan_stitch(edges):
    """edges: list of SlabEdges in slab order -> StitchPlan (same on every rank)."""
    n = len(edges)
    join = [edges[s].last_is_variant == edges[s + 1].first_is_variant for s in range(n - 1)]
    chains = []
    s = 0
    while s < n - 1:
        if not join[s]:
            s += 1
            continue
        a, b = s, s + 1
        while b < n - 1 and join[b] and edges[b].n_segments == 1:
            b += 1
        chains.append(Chain(a, b, bool(edges[a].last_is_variant)))
        s = b
    actions = [SlabAction() for _ in range(n)]
    for ci, ch in enumerate(chains):
        if not ch.variant:
            for i in range(ch.first + 1, ch.last + 1):
                actions[i].front_eds += 1              # "{"
                actions[i].front_seds += 3             # "{0}"
            for i in range(ch.first, ch.last):
                actions[i].back_eds += 1               # "}"
        else:
            actions[ch.first].back_eds += edges[ch.first].last_eds_bytes
            actions[ch.first].back_seds += edges[ch.first].last_seds_bytes
            actions[ch.first].owns.append(ci)
            for i in range(ch.first + 1, ch.last + 1):
                actions[i].front_eds += edges[i].first_eds_bytes
                actions[i].front_seds += edges[i].first_seds_bytes
    return StitchPlan(chains, actions)


def chain_columns(chain, rank, edges):
    """Column range (col0, ncols) of slab `rank` that belongs to the variant chain, or None."""
    if rank < chain.first or rank > chain.last:
        return None
    e = edges[rank]
    if rank == chain.first:
        return e.cols - e.last_cols, e.last_cols
    return 0, e.first_cols                       # whole slab when it is a single run


def mini_alignment(column_blocks, n_rows):
    """Row-major blocks [(bytes, ncols), ...] of the same rows -> a one-line-per-row MSA image."""
    out = bytearray()
    for r in range(n_rows):
        out += b">r\n"
        for data, ncols in column_blocks:
            out += data[r * ncols:(r + 1) * ncols]
        out += b"\n"
    return bytes(out)


def piece_bounds(edges, action):
    """(eds_lo, eds_hi, seds_lo, seds_hi): the part of the slab's own text that survives."""
    e_lo, e_hi = action.front_eds, edges.eds_bytes - action.back_eds
    s_lo, s_hi = action.front_seds, edges.seds_bytes - action.back_seds
    if e_hi < e_lo:
        e_hi = e_lo
    if s_hi < s_lo:
        s_hi = s_lo
    return e_lo, e_hi, s_lo, s_hi


class RemoteRankError(RuntimeError):
    """Another rank's step failed; the message is that rank's own error text."""

    def __init__(self, rank, text):
        super().__init__(text)
        self.rank, self.message = rank, text


class TensorComm:
    """The exchanges of the partitions as tensor collectives (no pickled objects): all_gather_into_tensor of int64
    vectors of a fixed length, and of uint8 payloads padded to the largest contribution (their sizes travel first).
    `device` is where the collective's tensors live ("cuda" for nccl = RCCL, "cpu" for gloo)."""

    def __init__(self, dist, world, device=None):
        self.dist, self.world, self.device = dist, world, device

    def _dev(self):
        if self.device is None:
            self.device = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        return self.device

    def _gather(self, t):
        import torch
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous())
        return out.cpu()

    def ints(self, values):
        """Every rank contributes the same number of integers -> list (by rank) of lists."""
        import torch
        g = self._gather(torch.tensor([int(v) for v in values], dtype=torch.int64, device=self._dev()))
        return [[int(x) for x in g[r].tolist()] for r in range(self.world)]

    def payloads(self, data):
        """bytes of any length per rank -> list (by rank) of bytes."""
        import torch
        sizes = [v[0] for v in self.ints([len(data)])]
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8)
        if len(data):
            buf[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        g = self._gather(buf.to(self._dev()))
        return [g[r, :sizes[r]].numpy().tobytes() for r in range(self.world)]

    def raise_if_any_failed(self, err):
        """Collective: every rank passes its own exception (or None).  When any rank failed, the failing ranks' error
        texts travel as one payload exchange; a failing rank re-raises its own exception, every other rank raises
        RemoteRankError with the text of the lowest failing rank - so no rank is left waiting in a later collective
        and whichever rank prints has the reference-worded message."""
        flags = [v[0] for v in self.ints([0 if err is None else 1])]
        if not any(flags):
            return
        text = "" if err is None else str(getattr(err, "message", None) or err)
        texts = self.payloads(text.encode("utf-8", "replace"))
        if err is not None:
            raise err
        first = flags.index(1)
        raise RemoteRankError(first, texts[first].decode("utf-8", "replace"))


EDGE_FIELDS = ("n_segments", "cols", "eds_bytes", "seds_bytes", "first_is_variant", "first_cols", "first_eds_bytes",
               "first_seds_bytes", "last_is_variant", "last_cols", "last_eds_bytes", "last_seds_bytes")


class SlabStitcher:
    """Runs the stitch for one rank.  `dist` is torch.distributed (nccl = RCCL on the GPU box, gloo in the CPU
    tests).  Every exchange is a fixed-size tensor collective, the same calls on both backends:
      1. all_gather of the edge descriptors, 12 int64 per rank;
      2. only if a variant run crosses a boundary: all_gather of one uint8 buffer per rank with the raw columns of the
         (at most two) variant chains the rank takes part in - every rank derives all block sizes from step 1, so the
         buffers are padded to the largest contribution and need no size exchange;
      3. only then: all_gather of the piece sizes, 2 int64 per rank (without a recomputed segment every rank derives all
         sizes from step 1).
    `device` is where the collective's tensors live ("cuda" for nccl, "cpu" for gloo)."""

    def __init__(self, rank, world, n_rows, dist, transform, get_edges, get_columns, device=None):
        self.rank, self.world, self.n_rows, self.dist = rank, world, n_rows, dist
        self.transform = transform          # bytes (mini MSA) -> (eds, seds)
        self.get_edges = get_edges          # () -> SlabEdges of this rank's slab
        self.get_columns = get_columns      # (col0, ncols) -> row-major bytes of this rank's slab
        self.device = device
        self.last = None

    def _device(self):
        if self.device is None:
            self.device = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        return self.device

    def _all_gather(self, t):
        """t: 1-D tensor of the same length on every rank -> [world, len] (host copy)."""
        import torch
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out.view(-1), t.contiguous())
        return out.cpu()

    def stitch(self):
        import torch
        dev = self._device()
        my = self.get_edges()
        g = self._all_gather(torch.tensor([getattr(my, f) for f in EDGE_FIELDS], dtype=torch.int64, device=dev))
        gathered = [SlabEdges(*[int(v) for v in g[r].tolist()]) for r in range(self.world)]
        plan = plan_stitch(gathered)
        action = plan.actions[self.rank]
        variant = [ci for ci, ch in enumerate(plan.chains) if ch.variant]
        if variant:
            # raw columns of the variant chains: rank r's buffer = its blocks in chain order, n_rows x ncols bytes each
            def blocks_of(r):
                out = []
                for ci in variant:
                    rng = chain_columns(plan.chains[ci], r, gathered)
                    if rng is not None:
                        out.append((ci, rng[0], rng[1]))
                return out
            sizes_r = [sum(self.n_rows * nc for _, _, nc in blocks_of(r)) for r in range(self.world)]
            cap = max(max(sizes_r), 1)
            buf = torch.zeros(cap, dtype=torch.uint8)
            at = 0
            for _ci, c0, nc in blocks_of(self.rank):
                data = self.get_columns(c0, nc)
                buf[at:at + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
                at += len(data)
            allbuf = self._all_gather(buf.to(dev))
            where = {}                                         # (rank, chain) -> (offset, ncols)
            for r in range(self.world):
                at = 0
                for ci, _c0, nc in blocks_of(r):
                    where[(r, ci)] = (at, nc)
                    at += self.n_rows * nc
            for ci in action.owns:
                ch = plan.chains[ci]
                blocks = []
                for r in range(ch.first, ch.last + 1):
                    at, nc = where[(r, ci)]
                    blocks.append((bytes(allbuf[r, at:at + self.n_rows * nc].numpy().tobytes()), nc))
                e, s = self.transform(mini_alignment(blocks, self.n_rows))
                action.extra_eds += e
                action.extra_seds += s
        e_lo, e_hi, s_lo, s_hi = piece_bounds(my, action)
        if variant:
            sz = self._all_gather(torch.tensor([(e_hi - e_lo) + len(action.extra_eds), (s_hi - s_lo) + len(action.extra_seds)],
                                               dtype=torch.int64, device=dev))
            sizes = [(int(sz[r, 0]), int(sz[r, 1])) for r in range(self.world)]
        else:
            # no recomputed segment anywhere: every rank derives all piece sizes from the gathered edges
            sizes = []
            for r in range(self.world):
                b = piece_bounds(gathered[r], plan.actions[r])
                sizes.append((b[1] - b[0], b[3] - b[2]))
        self.last = {
            "action": action, "eds_range": (e_lo, e_hi), "seds_range": (s_lo, s_hi),
            "eds_offset": sum(x[0] for x in sizes[:self.rank]), "seds_offset": sum(x[1] for x in sizes[:self.rank]),
            "eds_total": sum(x[0] for x in sizes), "seds_total": sum(x[1] for x in sizes),
            "chains": len(plan.chains),
        }
        return self.last


def stitched_piece(eds, seds, result):
    """This rank's contribution to the whole .eds / .seds (bytes objects in, bytes out)."""
    e_lo, e_hi = result["eds_range"]
    s_lo, s_hi = result["seds_range"]
    a = result["action"]
    return eds[e_lo:e_hi] + a.extra_eds, seds[s_lo:s_hi] + a.extra_seds


def gpu_stitcher(ctx, mini_ctx, rank, world, n_rows, cols, dist, device=None):
    """SlabStitcher wired to the C ABI: edges and columns from the planned slab in `ctx`, mini
    alignments through a second context so that the slab's plan stays intact."""
    def get_edges():
        e = ctx.msa_edge_info()
        info = ctx.msa_info()
        return SlabEdges(n_segments=e["n_segments"], cols=cols, eds_bytes=get_edges.sizes[0], seds_bytes=get_edges.sizes[1],
                         first_is_variant=e["first_is_variant"], first_cols=e["first_cols"],
                         first_eds_bytes=e["first_eds_bytes"], first_seds_bytes=e["first_seds_bytes"],
                         last_is_variant=e["last_is_variant"], last_cols=e["last_cols"],
                         last_eds_bytes=e["last_eds_bytes"], last_seds_bytes=e["last_seds_bytes"]) if info else None
    get_edges.sizes = (0, 0)
    st = SlabStitcher(rank, world, n_rows, dist, lambda m: mini_ctx.msa_transform(m, 0), get_edges,
                      lambda c0, nc: ctx.msa_copy_columns(c0, nc, n_rows), device=device)
    st.set_sizes = lambda E, Q: setattr(get_edges, "sizes", (E, Q))
    return st


# ------------------------------------------------------------------------------------------------------
# msa2eds for a FILE over several GPUs: every rank cuts its column slab out of every row of the (memory-mapped)
# alignment - column c of row s lives at start[s] + c + c / line_width (msa_transforms.cpp:268-269) -, hands
# it to its GPU as a one-line-per-row image and the slabs are stitched as above.
# ------------------------------------------------------------------------------------------------------
def msa_layout(msa):
    """Row starts and geometry of a FASTA alignment image (bytes-like with find): (starts, raw_row_bytes,
    line_width or 0 for one-line rows, columns), or None when the file is not a plain uniform alignment (the
    unpartitioned transform then reports what is wrong with it, in the reference's words)."""
    n = len(msa)
    if n == 0 or msa[0:1] != b">":
        return None
    he = msa.find(b"\n")
    if he < 0:
        return None
    start0 = he + 1
    h2 = msa.find(b"\n>", start0)
    first_nl = msa.find(b"\n", start0)
    if h2 < 0 or first_nl < 0:
        return None
    lw, draw = first_nl - start0, h2 - start0
    if lw <= 0:
        return None
    if draw == lw:
        L, wrapped = lw, False
    else:
        nlines = (draw + 1 + lw) // (lw + 1)
        L, wrapped = draw + 1 - nlines, True
        if L <= 0 or (L - 1) // lw != nlines - 1:
            return None
    starts, pos = [start0], h2 + 1
    while pos < n:
        if msa[pos:pos + 1] != b">":
            if msa[pos:n].strip(b"\n"):
                return None                                    # something else than blank lines behind the last row
            break
        he = msa.find(b"\n", pos)
        if he < 0 or he + 1 + draw > n:
            return None
        st = he + 1
        if st + draw < n and msa[st + draw:st + draw + 1] != b"\n":
            return None
        starts.append(st)
        pos = st + draw + 1
    if len(starts) < 2:
        return None
    return starts, draw, (lw if wrapped else 0), L


def msa_slab_image(msa, layout, c0, c1):
    """Columns [c0, c1) of every row as a one-line-per-row alignment image."""
    starts, _draw, lw, _L = layout
    out = bytearray()
    for st in starts:
        if lw:
            piece = bytes(msa[st + c0 + c0 // lw:st + (c1 - 1) + (c1 - 1) // lw + 1]).replace(b"\n", b"")
        else:
            piece = bytes(msa[st + c0:st + c1])
        out += b">r\n"
        out += piece
        out += b"\n"
    return bytes(out)


ANCHOR_MAX_COLS = 1 << 16


def leds_anchor_fields(a, rank, world, ncols, eds_bytes, seds_bytes):
    """What a rank tells the others about its l-EDS slab (see MsaSharder): [ok, tail_col, tail_eds, tail_seds, head_end,
    head_eds, head_seds, ncols].  `a` is edsx_msa_anchor_info of the slab: its first and last common segment of at least
    l columns.  A middle slab needs two distinct anchors, the first slab only a last one, the last slab only a first one;
    an anchor further than ANCHOR_MAX_COLS from its end of the slab is not used (the boundary zone travels as raw
    columns)."""
    need_head, need_tail = rank > 0, rank + 1 < world
    ok = bool(a["found"])
    if ok and need_head and need_tail:
        ok = a["first_seg"] < a["last_seg"]
    if ok and need_head:
        ok = a["first_end"] <= ANCHOR_MAX_COLS
    if ok and need_tail:
        ok = ncols - a["last_col"] <= ANCHOR_MAX_COLS
    if not ok:
        return [0, 0, 0, 0, 0, 0, 0, ncols]
    return [1,
            a["last_col"] if need_tail else ncols, a["last_eds_bytes"] if need_tail else eds_bytes,
            a["last_seds_bytes"] if need_tail else seds_bytes,
            a["first_end"] if need_head else 0, a["first_eds_end"] if need_head else 0, a["first_seds_end"] if need_head else 0,
            ncols]


class MsaSharder:
    """Column-slab partition of an alignment FILE for one rank.  slab_fn(image, n_rows, ncols) ->
    (eds, seds, SlabEdges, get_columns) transforms a slab (the C ABI on the GPU, the oracle in the CPU tests);
    mini_fn(image) -> (eds, seds) transforms the few boundary columns of a variant run that crosses a cut.
    Context length l > 0 (leds_fn / mini_leds_fn given): an l-EDS joins variant runs with the common runs of fewer than
    l columns between them, and only a common run of at least l columns - or one at either end of the alignment - stands
    alone (msa_transforms.cpp:133-190).  A slab transformed on its own applies the "at either end" clause at its own
    ends, so its text is the alignment's text exactly between its first and its last standalone run of >= l columns (its
    anchors).  Every boundary is therefore recomputed from the last anchor of the left slab to the first anchor of the
    right one (both included), by the left rank: leds_fn(image, n_rows, ncols, l) -> (eds, seds, anchor_info,
    get_columns), mini_leds_fn(image, l) -> (eds, seds).  Exchanges: 8 int64 per rank, the raw boundary columns as one
    padded uint8 payload per rank, 2 int64 per rank.  A slab without usable anchors sends the file to rank 0."""

    def __init__(self, rank, world, dist, slab_fn, mini_fn, whole_fn, device=None, leds_fn=None, mini_leds_fn=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.slab_fn, self.mini_fn, self.whole_fn, self.device = slab_fn, mini_fn, whole_fn, device
        self.leds_fn, self.mini_leds_fn = leds_fn, mini_leds_fn
        self.comm = TensorComm(dist, world, device)

    def _run_leds(self, msa, layout, l):
        """-> the rank's piece, or None when some slab has no usable anchors (decided by every rank alike)"""
        rank, world = self.rank, self.world
        starts, _draw, _lw, L = layout
        n_rows = len(starts)
        c0, c1 = L * rank // world, L * (rank + 1) // world
        ncols = c1 - c0
        eds, seds, a, get_columns, err = b"", b"", None, None, None
        try:
            eds, seds, a, get_columns = self.leds_fn(msa_slab_image(msa, layout, c0, c1), n_rows, ncols, l)
        except Exception as ex:                                    # noqa: BLE001 - raised on every rank below
            err = ex
        self.comm.raise_if_any_failed(err)
        allv = self.comm.ints(leds_anchor_fields(a, rank, world, ncols, len(eds), len(seds)))
        if not all(v[0] for v in allv):
            return None
        _ok, tail_col, tail_eds, tail_seds, head_end, head_eds, head_seds, _n = allv[rank]
        tail_w = [v[7] - v[1] for v in allv]
        # boundary columns: [tail_col, ncols) and [0, head_end) of every slab, row-major blocks
        payload, err = b"", None
        try:
            if tail_w[rank]:
                payload += get_columns(tail_col, tail_w[rank])
            if head_end:
                payload += get_columns(0, head_end)
        except Exception as ex:                                    # noqa: BLE001
            err = ex
        self.comm.raise_if_any_failed(err)
        blocks = self.comm.payloads(payload)
        extra_e, extra_s, err = b"", b"", None
        if rank + 1 < world:
            try:
                wl, wr = tail_w[rank], allv[rank + 1][4]
                left = blocks[rank][:n_rows * wl]
                right = blocks[rank + 1][n_rows * tail_w[rank + 1]:n_rows * (tail_w[rank + 1] + wr)]
                mini = bytearray()
                for r in range(n_rows):
                    mini += b">r\n" + left[r * wl:(r + 1) * wl] + right[r * wr:(r + 1) * wr] + b"\n"
                extra_e, extra_s = self.mini_leds_fn(bytes(mini), l)
            except Exception as ex:                                # noqa: BLE001
                err = ex
        self.comm.raise_if_any_failed(err)
        e = eds[head_eds:tail_eds] + extra_e
        sd = seds[head_seds:tail_seds] + extra_s
        sizes = self.comm.ints([len(e), len(sd)])
        return {"eds": e, "seds": sd, "eds_offset": sum(v[0] for v in sizes[:rank]), "seds_offset": sum(v[1] for v in sizes[:rank]),
                "eds_total": sum(v[0] for v in sizes), "seds_total": sum(v[1] for v in sizes), "partitioned": True}

    def run(self, msa, context_len=0):
        rank, world = self.rank, self.world
        layout = msa_layout(msa)
        if context_len > 0 and layout is not None and world > 1:
            if self.leds_fn is not None and layout[3] >= 2 * world and layout[3] // world >= 4 * context_len:
                piece = self._run_leds(msa, layout, context_len)
                if piece is not None:
                    return piece
            layout = None                                          # no anchors, narrow slabs: rank 0 alone
        if layout is None or layout[3] < 2 * world or world == 1:
            # not partitioned (odd files; l-EDS slabs without standalone common runs near their ends): rank 0 alone
            eds, seds, err = b"", b"", None
            if rank == 0:
                try:
                    eds, seds = self.whole_fn(msa, context_len)
                except Exception as ex:                            # noqa: BLE001 — raised on every rank below
                    err = ex
            self.comm.raise_if_any_failed(err)
            import torch
            dev = self.device or ("cuda" if self.dist.get_backend() == "nccl" else "cpu")
            t = torch.tensor([len(eds), len(seds)], dtype=torch.int64, device=dev)
            self.dist.broadcast(t, 0)
            tot = [int(x) for x in t.cpu().tolist()]
            return {"eds": eds, "seds": seds, "eds_offset": 0, "seds_offset": 0, "eds_total": tot[0], "seds_total": tot[1],
                    "partitioned": False}
        starts, _draw, _lw, L = layout
        c0, c1 = L * rank // world, L * (rank + 1) // world
        eds, seds, edges, get_columns, err = b"", b"", None, None, None
        try:
            eds, seds, edges, get_columns = self.slab_fn(msa_slab_image(msa, layout, c0, c1), len(starts), c1 - c0)
        except Exception as ex:                                    # noqa: BLE001 — raised on every rank below
            err = ex
        self.comm.raise_if_any_failed(err)                         # before the stitch: no rank waits in its collectives
        st = SlabStitcher(rank, world, len(starts), self.dist, self.mini_fn, lambda: edges, get_columns, device=self.device)
        res = st.stitch()
        e, s = stitched_piece(eds, seds, res)
        return {"eds": e, "seds": s, "eds_offset": res["eds_offset"], "seds_offset": res["seds_offset"],
                "eds_total": res["eds_total"], "seds_total": res["seds_total"], "partitioned": True}


def gpu_msa_sharder(ctx, mini_ctx, rank, world, dist, device=None):
    """MsaSharder wired to the C ABI (host-buffer transform of the slab image; edges and boundary columns from the
    slab's plan, mini alignments through a second context)."""
    def slab_fn(image, n_rows, ncols):
        eds, seds = ctx.msa_transform(image, 0)
        e = ctx.msa_edge_info()
        edges = SlabEdges(n_segments=e["n_segments"], cols=ncols, eds_bytes=len(eds), seds_bytes=len(seds),
                          first_is_variant=e["first_is_variant"], first_cols=e["first_cols"],
                          first_eds_bytes=e["first_eds_bytes"], first_seds_bytes=e["first_seds_bytes"],
                          last_is_variant=e["last_is_variant"], last_cols=e["last_cols"],
                          last_eds_bytes=e["last_eds_bytes"], last_seds_bytes=e["last_seds_bytes"])
        return eds, seds, edges, lambda c0, nc: ctx.msa_copy_columns(c0, nc, n_rows)
    def leds_fn(image, n_rows, ncols, l):
        eds, seds = ctx.msa_transform(image, l)
        return eds, seds, ctx.msa_anchor_info(l), lambda c0, nc: ctx.msa_copy_columns(c0, nc, n_rows)
    return MsaSharder(rank, world, dist, slab_fn, lambda m: mini_ctx.msa_transform(m, 0),
                      lambda m, l: ctx.msa_transform(m, l), device=device, leds_fn=leds_fn,
                      mini_leds_fn=lambda m, l: mini_ctx.msa_transform(m, l))


# ======================================================================================================
# VCF -> EDS over several GPUs: partition by reference position (BASELINE configs[3])
# ======================================================================================================
#
# The reference groups records into maximal chains of overlapping records (vcf_transforms.cpp:482-534) and
# walks the groups once, flushing the reference text between them (:554-668).  A cut placed at the first
# record of a group is therefore free: the left range ends with exactly the common text the reference
# flushes in front of that group, the right range starts with the group, and the pieces concatenate to the
# reference's output byte for byte — no repair of text.  What is global is the *order* of the records: the
# reference sorts the whole array with an unstable std::sort (:715-718), so equal positions land in an order
# that depends on the whole array.  The exchange steps are therefore
#
#   1. every rank indexes its byte range of the VCF (POS, REF length, line spans; no genotypes),
#   2. all-gather of the (POS, REF length) arrays — 16 B per record,
#   3. every rank derives the same order (edsx_vcf_sort_order = that std::sort), group starts (running
#      maximum of record ends, :510) and cuts (first group start at or after k * n / world),
#   4. record lines whose position falls into another rank's range are exchanged as text (for a sorted
#      VCF: only the slivers next to the byte cuts; everything of a shuffled file),
#   5. every rank runs its range through edsx_vcf_transform_range on its own GPU,
#   6. all-gather of the piece sizes for the output offsets.  The output stays sharded.
#
# Files with wrapped positions (POS 0 or near 2^64, see vcf_device.hip) are not partitioned: rank 0 takes
# all records.  The FASTA is replicated (1 GB of 288 GB HBM for BASELINE configs[3]); only the walk over it
# is partitioned.

def vcf_byte_range(vcf, rank, world):
    """Byte range [lo, hi) of rank's share of the file, cut at line starts (same rule on every rank)."""
    n = len(vcf)

    def cut(k):
        if k <= 0:
            return 0
        if k >= world:
            return n
        g = n * k // world
        if g == 0:
            return 0
        nl = vcf.find(b"\n", g - 1)          # a line starts right behind the first newline at or after g-1
        return n if nl < 0 else nl + 1
    return cut(rank), cut(rank + 1)


def plan_vcf_cuts(pos, reflen, order, world):
    """Sorted-record index cuts c[0..world] (c[r] is a group start) + sorted starts; identical on every rank.

    pos / reflen: numpy uint64 in file order; order: permutation of the reference's sort."""
    import numpy as np
    n = len(pos)
    cuts = [0] * (world + 1)
    cuts[world] = n
    if n == 0:
        return cuts, np.zeros(0, dtype=np.uint64), False
    with np.errstate(over="ignore"):
        start = pos[order] - np.uint64(1)                     # wraps for POS 0 like the reference's size_t
        end = start + reflen[order]
    wraps = bool(np.any(start == np.uint64(0xFFFFFFFFFFFFFFFF)) or np.any(end < start))
    if wraps or world == 1:
        for r in range(1, world):
            cuts[r] = n
        return cuts, start, wraps
    prefmax = np.maximum.accumulate(end)
    flag = np.ones(n, dtype=bool)
    flag[1:] = start[1:] >= prefmax[:-1]                      # a record opens a group iff start >= every earlier end (:510)
    gstarts = np.flatnonzero(flag)
    for r in range(1, world):
        target = n * r // world
        k = int(np.searchsorted(gstarts, target, side="left"))
        cuts[r] = int(gstarts[k]) if k < len(gstarts) else n
        if cuts[r] < cuts[r - 1]:
            cuts[r] = cuts[r - 1]
    return cuts, start, wraps


def _line_runs(g_sel, k, base, off, length, vcf, lo):
    """Records g_sel (global file indices owned by this rank, ascending sorted index k) as maximal runs of lines that
    are adjacent both in the sorted order and in the file: -> (run_k int64[R], run_nbytes int64[R], blob) with the runs'
    text back to back in `blob` (inner newlines kept, none between runs).  Arrays, not per-line tuples: a shuffled
    file has as many runs as lines."""
    import numpy as np
    if len(g_sel) == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), b""
    loc = (g_sel - base).astype(np.int64)
    o, ln = off[loc].astype(np.int64), length[loc].astype(np.int64)
    brk = np.flatnonzero((np.diff(k) != 1) | (np.diff(loc) != 1) | (o[:-1] + ln[:-1] + 1 != o[1:])) + 1
    firsts = np.concatenate(([0], brk)).astype(np.int64)
    lasts = np.concatenate((brk, [len(loc)])).astype(np.int64) - 1
    b0, b1 = lo + o[firsts], lo + o[lasts] + ln[lasts]
    if len(firsts) == 1:
        blob = bytes(vcf[int(b0[0]):int(b1[0])])
    else:
        blob = b"".join([vcf[x:y] for x, y in zip(b0.tolist(), b1.tolist())])
    return k[firsts].astype(np.int64), (b1 - b0).astype(np.int64), blob


class VcfSharder:
    """Runs the position-range partition for one rank.  `dist` is torch.distributed (nccl = RCCL on the GPU
    box, gloo in the CPU tests); every exchange is a tensor collective (TensorComm).  The three callables are the C ABI on the
    GPU (gpu_vcf_sharder) and the oracle in the CPU tests."""

    def __init__(self, rank, world, dist, index_fn, sort_fn, range_fn, device=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.comm = TensorComm(dist, world, device)
        self.index_fn = index_fn            # bytes -> (pos, reflen, line_off, line_len, counters)
        self.sort_fn = sort_fn              # uint64[n] -> uint32[n]
        self.range_fn = range_fn            # (lines, fasta, cur0, next_start|None) -> (eds, seds, counters)
        self.last = None

    def run(self, vcf, fasta):
        import numpy as np
        rank, world = self.rank, self.world
        lo, hi = vcf_byte_range(vcf, rank, world)
        pos, reflen, off, length, counters = self.index_fn(vcf[lo:hi])
        # (POS, REF length) of every record, 16 bytes each, and the four counters: tensor collectives
        CK = ("total_variants", "processed_variants", "skipped_malformed", "skipped_unsupported_sv")
        head = self.comm.ints([len(pos)] + [counters[k] for k in CK])
        blobs = self.comm.payloads(np.concatenate((np.asarray(pos, dtype=np.uint64), np.asarray(reflen, dtype=np.uint64))).tobytes())
        gathered = []
        for r in range(world):
            arr = np.frombuffer(blobs[r], dtype=np.uint64)
            nr = head[r][0]
            gathered.append((arr[:nr], arr[nr:2 * nr], dict(zip(CK, head[r][1:]))))
        base = np.concatenate(([0], np.cumsum([len(g[0]) for g in gathered]))).astype(np.int64)
        gpos = np.concatenate([g[0] for g in gathered]).astype(np.uint64)
        greflen = np.concatenate([g[1] for g in gathered]).astype(np.uint64)
        if len(gpos) < 2 or bool(np.all(gpos[1:] > gpos[:-1])):
            order = np.arange(len(gpos), dtype=np.int64)           # strictly ascending: the sort cannot move anything
        else:
            order = np.argsort(gpos, kind="stable").astype(np.int64)
            sp = gpos[order]
            if bool(np.any(sp[1:] == sp[:-1])):                    # equal positions: the reference's std::sort decides
                order = self.sort_fn(gpos).astype(np.int64)
        cuts, start, wraps = plan_vcf_cuts(gpos, greflen, order, world)
        # lines of mine that another rank's range needs, and the runs I keep
        outgoing = {}
        mine = None
        for d in range(world):
            g = order[cuts[d]:cuts[d + 1]]
            sel = np.flatnonzero((g >= base[rank]) & (g < base[rank + 1]))
            if len(sel) == 0:
                continue
            runs = _line_runs(g[sel], cuts[d] + sel.astype(np.int64), base[rank], off, length, vcf, lo)
            if d == rank:
                mine = runs
            else:
                outgoing[d] = runs
        # record lines that change rank: per destination (run count, blob bytes) as integers, then one byte payload per
        # rank = for every destination in order its run_k (int64), run_nbytes (int64) and text
        hdr, parts = [], []
        for d in range(world):
            if d in outgoing:
                rk, rn, blob = outgoing[d]
                hdr += [len(rk), len(blob)]
                parts += [np.asarray(rk, dtype=np.int64).tobytes(), np.asarray(rn, dtype=np.int64).tobytes(), blob]
            else:
                hdr += [0, 0]
        heads = self.comm.ints(hdr)
        pay = self.comm.payloads(b"".join(parts))
        incoming = []
        for r in range(world):
            if r == rank:
                continue
            at = 0
            for d in range(world):
                nr, nb = heads[r][2 * d], heads[r][2 * d + 1]
                if d == rank and (nr or nb):
                    rk = np.frombuffer(pay[r], dtype=np.int64, count=nr, offset=at)
                    rn = np.frombuffer(pay[r], dtype=np.int64, count=nr, offset=at + 8 * nr)
                    incoming.append((rk, rn, pay[r][at + 16 * nr:at + 16 * nr + nb]))
                at += 16 * nr + nb
        sources = ([mine] if mine is not None else []) + incoming
        if not sources:
            lines = b""
        elif len(sources) == 1 and len(sources[0][0]) == 1:
            lines = sources[0][2]
        else:
            run_k = np.concatenate([x[0] for x in sources])
            run_n = np.concatenate([x[1] for x in sources])
            blob_base = np.concatenate(([0], np.cumsum([len(x[2]) for x in sources])))[:-1]
            run_off = np.concatenate([bb + np.concatenate(([0], np.cumsum(x[1])))[:-1] for bb, x in zip(blob_base, sources)])
            text = b"".join(x[2] for x in sources)
            by_k = np.argsort(run_k, kind="stable")
            s0, s1 = run_off[by_k], (run_off + run_n)[by_k]
            lines = b"\n".join([text[x:y] for x, y in zip(s0.tolist(), s1.tolist())])
        nonempty = [r for r in range(world) if cuts[r] < cuts[r + 1]]
        n_mine = cuts[rank + 1] - cuts[rank]
        eds = seds = b""
        groups = 0
        err = None
        runs_here = n_mine > 0 or (not nonempty and rank == 0)      # an empty VCF is rank 0's: the bare reference
        if runs_here:
            first = not nonempty or rank == nonempty[0]
            later = [r for r in nonempty if r > rank]
            cur0 = 0 if first else int(start[cuts[rank]])
            next_start = int(start[cuts[later[0]]]) if later else None
            try:
                eds, seds, st = self.range_fn(lines, fasta, cur0, next_start)
                groups = st["variant_groups"]
            except Exception as ex:                                # noqa: BLE001 — re-raised on every rank below
                err = ex
        self.comm.raise_if_any_failed(err)                         # the failing rank's own wording reaches every rank
        sizes = self.comm.ints([len(eds), len(seds), groups])
        stats = {k: sum(g[2][k] for g in gathered) for k in
                 ("total_variants", "processed_variants", "skipped_malformed", "skipped_unsupported_sv")}
        stats["variant_groups"] = sum(s[2] for s in sizes)
        self.last = {
            "eds": eds, "seds": seds, "stats": stats, "cuts": cuts, "records": n_mine, "wraps": wraps,
            "moved_lines_bytes": sum(len(x[2]) for x in outgoing.values()),
            "eds_offset": sum(s[0] for s in sizes[:rank]), "seds_offset": sum(s[1] for s in sizes[:rank]),
            "eds_total": sum(s[0] for s in sizes), "seds_total": sum(s[1] for s in sizes),
        }
        return self.last


def gpu_vcf_sharder(ctx, rank, world, dist, device=None):
    """VcfSharder wired to the C ABI of this rank's GPU context."""
    return VcfSharder(rank, world, dist, ctx.vcf_index, ctx.vcf_sort_order,
                      lambda lines, fasta, cur0, nxt: ctx.vcf_transform_range(lines, fasta, cur0, nxt), device=device)


# ======================================================================================================
# EDS -> l-EDS merge over several GPUs: partition by symbol range (merge row)
# ======================================================================================================
#
# A pair (i, i+1) is merged when one of the two is a short single string (shorter than l, not the first / last
# symbol) or both are degenerate (eds_transforms.cpp:75-97); inside a run of consecutive mergeable pairs the pairs at
# even offsets are taken (:63-66).  A *sentinel* — a single-string symbol of at least l characters whose two
# neighbours are degenerate — is in no mergeable pair, so no run crosses it and the two sides evolve exactly as
# inside the whole EDS, round for round.  Neighbouring ranks share the sentinel symbol (last symbol of the left
# range, first of the right range); the left one prints it.  The only way a sentinel can be drawn into a merge later
# is a LINEAR product collapsing its degenerate neighbour to one short string; every range reports whether its
# sentinels stayed untouched (edsx_leds_merge_range) and if one did not — or a range fails — rank 0 runs the
# unpartitioned merge, which also yields the reference's error text with whole-file positions.
#
# Exchange steps (all_gather_into_tensor of a few int64 each):
#   1. every rank scans its byte range of .eds / .seds: string starts, '{' of .seds, and the first sentinel at or after
#      the start of its range (with the number of strings in front of it inside the range),
#   2. the ranks that hold the sentinels' source sets in .seds locate them (set index = strings in front),
#   3. piece sizes + sentinel flags.
# Text with whitespace inside (other than at the end), commas outside braces or unbalanced braces is not
# partitioned (rank 0 takes it whole): the tools never write such files, and the tokeniser's exact error texts
# come from the whole-file path.

def _np_bytes(buf, lo, hi):
    import numpy as np
    return np.frombuffer(buf, dtype=np.uint8, count=hi - lo, offset=lo)


def _text_end(buf):
    """Length without trailing whitespace."""
    n = len(buf)
    while n and buf[n - 1:n].isspace():
        n -= 1
    return n


def eds_scan_range(eds, lo, hi, l, end, margin=1 << 16):
    """Scan bytes [lo, hi) of .eds text (hi <= end = text length without trailing whitespace).

    -> dict(ok, strings, cut) with cut = None or (sym_start, sym_end, strings_in_range_before_it): the first sentinel
    whose preceding '}' lies in [lo, hi)."""
    import numpy as np
    res = {"ok": True, "strings": 0, "cut": None}
    if hi <= lo:
        return res
    a = _np_bytes(eds, lo, hi)
    if np.any((a == 32) | ((a >= 9) & (a <= 13))):
        res["ok"] = False
        return res
    is_o, is_c, is_comma = a == 0x7B, a == 0x7D, a == 0x2C
    inside0 = 1 if eds.rfind(b"{", 0, lo) > eds.rfind(b"}", 0, lo) else 0
    delta = is_o.astype(np.int8) - is_c.astype(np.int8)
    depth_after = inside0 + np.cumsum(delta, dtype=np.int64)
    depth_before = depth_after - delta
    if depth_after.min(initial=0) < 0 or depth_after.max(initial=0) > 1 or np.any(is_comma & (depth_before == 0)) \
            or np.any(is_o & (depth_before == 1)):
        res["ok"] = False
        return res
    prev = np.empty_like(a)
    prev[1:] = a[:-1]
    prev[0] = eds[lo - 1] if lo > 0 else 0x7D
    starts = is_o | is_comma | ((depth_before == 0) & ~is_o & ~is_c & (prev == 0x7D))
    cs = np.concatenate(([0], np.cumsum(starts, dtype=np.int64)))
    res["strings"] = int(cs[-1])
    if lo == 0:
        return res                                               # rank 0's range starts the text: no cut of its own
    # ---- first sentinel: brace sequence of a window that reaches one group back and `margin` bytes ahead
    w0 = eds.rfind(b"{", 0, lo)
    w0 = lo if w0 < 0 else w0
    w1 = min(end, hi + margin + 4 * l)
    w = _np_bytes(eds, w0, w1)
    if np.any((w == 32) | ((w >= 9) & (w <= 13))):
        return res                                               # whitespace ahead: some rank reports it
    bmask = (w == 0x7B) | (w == 0x7D)
    bp = np.flatnonzero(bmask).astype(np.int64)
    bt = w[bp] == 0x7B                                           # True = '{'
    cc = np.concatenate(([0], np.cumsum(w == 0x2C, dtype=np.int64)))
    nb = len(bp)
    if nb < 4:
        return res

    def commas(x, y):                                            # commas strictly between window positions x < y
        return cc[y] - cc[x + 1]

    k = np.arange(1, nb - 1)                                     # bp[k] is the '}' in front of the sentinel
    base = (~bt[k]) & bt[k - 1] & (commas(bp[k - 1], bp[k]) > 0) & (bp[k] + w0 >= lo) & (bp[k] + w0 < hi) & bt[k + 1]
    need = max(int(l), 1)
    # COMPACT: } p {x,y}
    k2 = np.minimum(k + 2, nb - 1)
    compact = base & (bp[k + 1] - bp[k] - 1 >= need) & (commas(bp[k], bp[k + 1]) == 0) & (k + 2 < nb) & (~bt[k2]) & (commas(bp[k + 1], bp[k2]) > 0)
    # FULL: }{p}{x,y}
    k3, k4 = np.minimum(k + 3, nb - 1), np.minimum(k + 4, nb - 1)
    full = base & (bp[k + 1] == bp[k] + 1) & (k + 4 < nb) & (~bt[k2]) & (bp[k2] - bp[k + 1] - 1 >= need) & \
        (commas(bp[k + 1], bp[k2]) == 0) & bt[k3] & (bp[k3] == bp[k2] + 1) & (~bt[k4]) & (commas(bp[k3], bp[k4]) > 0)
    hit = np.flatnonzero(compact | full)
    if len(hit) == 0:
        return res
    j = int(hit[0])
    kk = int(k[j])
    if compact[j]:
        s, e = int(bp[kk]) + 1 + w0, int(bp[kk + 1]) + w0
    else:
        s, e = int(bp[kk + 1]) + w0, int(bp[kk + 2]) + 1 + w0
    res["cut"] = (s, e, int(cs[s - lo]))
    return res


def seds_scan_range(seds, lo, hi):
    """-> (ok, number of '{' in [lo, hi))"""
    import numpy as np
    if hi <= lo:
        return True, 0
    a = _np_bytes(seds, lo, hi)
    if np.any((a == 32) | ((a >= 9) & (a <= 13))):
        return False, 0
    return True, int(np.count_nonzero(a == 0x7B))


class MergeSharder:
    """Runs the symbol-range partition of the merge for one rank.  range_fn / whole_fn are the C ABI on the GPU
    (gpu_merge_sharder) and the oracle in the CPU tests."""

    def __init__(self, rank, world, dist, range_fn, whole_fn, device=None):
        self.rank, self.world, self.dist = rank, world, dist
        self.comm = TensorComm(dist, world, device)
        self.range_fn = range_fn    # (eds, seds|None, l, compact, head, tail) -> (leds, seds_out, head_intact, tail_intact)
        self.whole_fn = whole_fn    # (eds, seds|None, l, compact) -> (leds, seds_out)
        self.last = None

    def _whole_on_rank0(self, eds, seds, l, compact, why):
        out, sout, err = b"", b"", None
        if self.rank == 0:
            try:
                out, sout = self.whole_fn(eds, seds, l, compact)
            except Exception as ex:  # noqa: BLE001 — raised on every rank below
                err = ex
        self.comm.raise_if_any_failed(err)                         # rank 0 words the reference's error, on every rank
        sizes = self.comm.ints([len(out), len(sout)])
        self.last = {"leds": out, "seds": sout, "partitioned": False, "why": why, "ranges": 1,
                     "leds_offset": 0 if self.rank == 0 else sizes[0][0], "seds_offset": 0 if self.rank == 0 else sizes[0][1],
                     "leds_total": sizes[0][0], "seds_total": sizes[0][1]}
        return self.last

    def run(self, eds, seds, l, compact=True):
        import numpy as np
        rank, world = self.rank, self.world
        linear = seds is not None
        if world == 1 or l == 0:
            return self._whole_on_rank0(eds, seds, l, compact, "single rank")
        end = _text_end(eds)
        lo, hi = end * rank // world, end * (rank + 1) // world
        scan = eds_scan_range(eds, lo, hi, l, end)
        send = _text_end(seds) if linear else 0
        slo, shi = send * rank // world, send * (rank + 1) // world
        sok, sbraces = seds_scan_range(seds, slo, shi) if linear else (True, 0)
        cut = scan["cut"]
        g1 = [(bool(v[0]), v[1], (v[3], v[4], v[5]) if v[2] else None, v[6]) for v in
              self.comm.ints([1 if (scan["ok"] and sok) else 0, scan["strings"], 1 if cut is not None else 0] +
                             (list(cut) if cut is not None else [0, 0, 0]) + [sbraces])]
        if not all(g[0] for g in g1):
            return self._whole_on_rank0(eds, seds, l, compact, "text not partitionable")
        str_base = np.concatenate(([0], np.cumsum([g[1] for g in g1])))
        cut_ranks = [r for r in range(1, world) if g1[r][2] is not None]
        if not cut_ranks:
            return self._whole_on_rank0(eds, seds, l, compact, "no sentinel found")
        sent = {r: (g1[r][2][0], g1[r][2][1], int(str_base[r]) + g1[r][2][2]) for r in cut_ranks}   # start, end, string index
        # ---- .seds: the set of string M starts at the M-th '{'
        located = {}
        if linear:
            br_base = np.concatenate(([0], np.cumsum([g[3] for g in g1])))
            if br_base[-1] != str_base[-1]:
                return self._whole_on_rank0(eds, seds, l, compact, "source count differs from the cardinality")
            mine = [r for r in cut_ranks if br_base[rank] <= sent[r][2] < br_base[rank + 1]]
            if mine:
                pos = np.flatnonzero(_np_bytes(seds, slo, shi) == 0x7B)
                for r in mine:
                    p0 = int(pos[sent[r][2] - int(br_base[rank])]) + slo
                    located[r] = (p0, seds.find(b"}", p0) + 1)
            flat = []
            for r in range(world):                                  # (start, end) of sentinel r's source set, -1: not mine
                flat += list(located[r]) if r in located else [-1, -1]
            located = {}
            for g in self.comm.ints(flat):
                for r in range(world):
                    if g[2 * r] >= 0:
                        located[r] = (g[2 * r], g[2 * r + 1])
            if any(r not in located or located[r][1] <= 0 for r in cut_ranks):
                return self._whole_on_rank0(eds, seds, l, compact, "source set of a sentinel not found")
        # ---- my range
        owners = [0] + cut_ranks
        out = sout = b""
        head_ok = tail_ok = True
        err = None
        if rank in owners:
            i = owners.index(rank)
            nxt = owners[i + 1] if i + 1 < len(owners) else None
            e0 = 0 if rank == 0 else sent[rank][0]
            e1 = len(eds) if nxt is None else sent[nxt][1]
            s_slice = None
            if linear:
                s0 = 0 if rank == 0 else located[rank][0]
                s1 = len(seds) if nxt is None else located[nxt][1]
                s_slice = seds[s0:s1]
            try:
                out, sout, head_ok, tail_ok = self.range_fn(eds[e0:e1], s_slice, l, compact, rank != 0, nxt is not None)
            except Exception as ex:  # noqa: BLE001 — any failure sends the whole text to rank 0 (exact error text)
                err = ex
        g3 = self.comm.ints([len(out), len(sout), 1 if (head_ok and tail_ok and err is None) else 0])
        if not all(g[2] for g in g3):
            return self._whole_on_rank0(eds, seds, l, compact, "a sentinel was merged or a range failed")
        self.last = {"leds": out, "seds": sout, "partitioned": True, "why": "", "ranges": len(owners),
                     "leds_offset": sum(g[0] for g in g3[:rank]), "seds_offset": sum(g[1] for g in g3[:rank]),
                     "leds_total": sum(g[0] for g in g3), "seds_total": sum(g[1] for g in g3)}
        return self.last


def gpu_merge_sharder(ctx, rank, world, dist):
    """MergeSharder wired to the C ABI of this rank's GPU context."""
    return MergeSharder(rank, world, dist,
                        lambda e, s, l, c, h, t: ctx.leds_merge_range(e, s, l, c, h, t),
                        lambda e, s, l, c: ctx.leds_merge(e, s, l, c))
