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
