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


class SlabStitcher:
    """Runs the stitch for one rank.  `dist` is torch.distributed (nccl = RCCL on the GPU box,
    gloo in the CPU tests); only all_gather_object on small Python objects is used."""

    def __init__(self, rank, world, n_rows, dist, transform, get_edges, get_columns):
        self.rank, self.world, self.n_rows, self.dist = rank, world, n_rows, dist
        self.transform = transform          # bytes (mini MSA) -> (eds, seds)
        self.get_edges = get_edges          # () -> SlabEdges of this rank's slab
        self.get_columns = get_columns      # (col0, ncols) -> row-major bytes of this rank's slab
        self.last = None

    def stitch(self):
        my = self.get_edges()
        gathered = [None] * self.world
        self.dist.all_gather_object(gathered, my)
        plan = plan_stitch(gathered)
        action = plan.actions[self.rank]
        if any(ch.variant for ch in plan.chains):
            # raw columns of every variant chain this rank takes part in (KBs)
            mine = {}
            for ci, ch in enumerate(plan.chains):
                if ch.variant:
                    rng = chain_columns(ch, self.rank, gathered)
                    if rng is not None:
                        mine[ci] = (self.get_columns(rng[0], rng[1]), rng[1])
            allcols = [None] * self.world
            self.dist.all_gather_object(allcols, mine)
            for ci in action.owns:
                ch = plan.chains[ci]
                blocks = [allcols[r][ci] for r in range(ch.first, ch.last + 1)]
                e, s = self.transform(mini_alignment(blocks, self.n_rows))
                action.extra_eds += e
                action.extra_seds += s
        e_lo, e_hi, s_lo, s_hi = piece_bounds(my, action)
        if any(ch.variant for ch in plan.chains):
            sizes = [None] * self.world
            self.dist.all_gather_object(sizes, ((e_hi - e_lo) + len(action.extra_eds), (s_hi - s_lo) + len(action.extra_seds)))
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


def gpu_stitcher(ctx, mini_ctx, rank, world, n_rows, cols, dist):
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
                      lambda c0, nc: ctx.msa_copy_columns(c0, nc, n_rows))
    st.set_sizes = lambda E, Q: setattr(get_edges, "sizes", (E, Q))
    return st


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


def _line_runs(gidx, k0, base, off, length, vcf, lo):
    """Records gidx (global file indices owned by this rank, in sorted order, first one at sorted index k0)
    -> [(sorted index of the first record, text)], maximal runs of lines adjacent in the file as one slice."""
    import numpy as np
    if len(gidx) == 0:
        return []
    loc = (gidx - base).astype(np.int64)
    o, ln = off[loc].astype(np.int64), length[loc].astype(np.int64)
    brk = np.flatnonzero((np.diff(loc) != 1) | (o[:-1] + ln[:-1] + 1 != o[1:])) + 1
    firsts = np.concatenate(([0], brk))
    lasts = np.concatenate((brk, [len(loc)])) - 1
    return [(k0 + int(a), bytes(vcf[lo + int(o[a]):lo + int(o[b] + ln[b])])) for a, b in zip(firsts, lasts)]


class VcfSharder:
    """Runs the position-range partition for one rank.  `dist` is torch.distributed (nccl = RCCL on the GPU
    box, gloo in the CPU tests); only all_gather_object is used.  The three callables are the C ABI on the
    GPU (gpu_vcf_sharder) and the oracle in the CPU tests."""

    def __init__(self, rank, world, dist, index_fn, sort_fn, range_fn):
        self.rank, self.world, self.dist = rank, world, dist
        self.index_fn = index_fn            # bytes -> (pos, reflen, line_off, line_len, counters)
        self.sort_fn = sort_fn              # uint64[n] -> uint32[n]
        self.range_fn = range_fn            # (lines, fasta, cur0, next_start|None) -> (eds, seds, counters)
        self.last = None

    def run(self, vcf, fasta):
        import numpy as np
        rank, world = self.rank, self.world
        lo, hi = vcf_byte_range(vcf, rank, world)
        pos, reflen, off, length, counters = self.index_fn(vcf[lo:hi])
        gathered = [None] * world
        self.dist.all_gather_object(gathered, (pos, reflen, counters))
        base = np.concatenate(([0], np.cumsum([len(g[0]) for g in gathered]))).astype(np.int64)
        gpos = np.concatenate([g[0] for g in gathered]).astype(np.uint64)
        greflen = np.concatenate([g[1] for g in gathered]).astype(np.uint64)
        order = self.sort_fn(gpos).astype(np.int64)
        cuts, start, wraps = plan_vcf_cuts(gpos, greflen, order, world)
        # lines of mine that another rank's range needs, and the runs I keep
        outgoing = {}
        pieces = []
        for d in range(world):
            g = order[cuts[d]:cuts[d + 1]]
            sel = np.flatnonzero((g >= base[rank]) & (g < base[rank + 1]))
            if len(sel) == 0:
                continue
            # runs must also be runs of the *sorted* order: split where the sorted index jumps
            jumps = np.flatnonzero(np.diff(sel) != 1) + 1
            runs = []
            for a, b in zip(np.concatenate(([0], jumps)), np.concatenate((jumps, [len(sel)]))):
                runs += _line_runs(g[sel[a:b]], cuts[d] + int(sel[a]), base[rank], off, length, vcf, lo)
            if d == rank:
                pieces += runs
            else:
                outgoing[d] = runs
        everything = [None] * world
        self.dist.all_gather_object(everything, outgoing)
        for r in range(world):
            if r != rank and rank in everything[r]:
                pieces += everything[r][rank]
        pieces.sort(key=lambda p: p[0])
        lines = b"\n".join(p[1] for p in pieces)
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
        sizes = [None] * world
        self.dist.all_gather_object(sizes, (len(eds), len(seds), groups, None if err is None else (type(err).__name__, str(err), getattr(err, "code", None))))
        failed = [s[3] for s in sizes if s[3] is not None]
        if failed:
            if err is not None:
                raise err
            raise RuntimeError("VCF range transform failed on another rank: %s: %s" % (failed[0][0], failed[0][1]))
        stats = {k: sum(g[2][k] for g in gathered) for k in
                 ("total_variants", "processed_variants", "skipped_malformed", "skipped_unsupported_sv")}
        stats["variant_groups"] = sum(s[2] for s in sizes)
        self.last = {
            "eds": eds, "seds": seds, "stats": stats, "cuts": cuts, "records": n_mine, "wraps": wraps,
            "moved_lines_bytes": sum(len(t) for runs in outgoing.values() for _, t in runs),
            "eds_offset": sum(s[0] for s in sizes[:rank]), "seds_offset": sum(s[1] for s in sizes[:rank]),
            "eds_total": sum(s[0] for s in sizes), "seds_total": sum(s[1] for s in sizes),
        }
        return self.last


def gpu_vcf_sharder(ctx, rank, world, dist):
    """VcfSharder wired to the C ABI of this rank's GPU context."""
    return VcfSharder(rank, world, dist, ctx.vcf_index, ctx.vcf_sort_order,
                      lambda lines, fasta, cur0, nxt: ctx.vcf_transform_range(lines, fasta, cur0, nxt))
