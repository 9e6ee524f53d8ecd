"""The measured thresholds of the path in ONE place: each is a break-even between two forms that compute the same thing, tuned on the graph
named beside it (one MI355X; profiles/ holds the tables).  Read at call time (`tuning.T.<name>`), so a test or a host with another graph family
can set them on the instance (every user reads T at its call sites; a graph / partition / cover that was BUILT keeps the value it was built with); nothing here changes results beyond the order in which sums are associated."""
from dataclasses import dataclass


@dataclass
class Tuning:
    # ---- aggregation kernels (graph.py) ---------------------------------------------------------------------------------------------------
    # rows with more edges are reduced in chunks of this size by the hub kernels.  S-pl10M (Chung-Lu power law, 10^7 nodes / 10^8 edges):
    # 64 / 128 / 512 / 1024 measured equal or slower (profiles/r02_spmm_gather_policy.md, r04_hub_chunk_small_graphs.md)
    hub_threshold: int = 256
    # the hot source rows of an aggregation should fill the 256 MiB Infinity Cache: count = hot_bytes / row bytes.  S-pl10M, d = 256 fp32:
    # 262 144 rows is the measured optimum (profiles/r02_spmm_gather_policy.md)
    hot_bytes: int = 256 << 20

    # ---- row-sparse backward (trunk.py, graph.CSRGraph.grad_support_plan) ------------------------------------------------------------------
    # below this many nodes an eager step is launch-bound and the plan's extra launches cost more than the gather saves.  S-pubmed (19 717
    # nodes): 1.38 -> 1.46 ms/step with the plan; S-arxiv (169 343): 3.53 -> 3.40.  (graph.rowsparse_small_ok — set by enable_hip_graph —
    # lifts it: under replay launches are free, S-pubmed config 2: 0.787 -> 0.751 ms)
    rowsparse_min_nodes: int = 1 << 16
    # the plan is used while the loss rows are at most this share of the rows.  S-pl10M with 50 % / 70 % loss rows: 187.9 / 193.2 ms against
    # 195.4 / 196.3 dense
    rowsparse_s0_limit: float = 0.7
    # a level's output stays compact while its support is at most this share of the rows.  S-arxiv (support 61 %): 3.53 -> 3.40 ms/step against 0.6
    rowsparse_max_frac: float = 0.7
    # level j through its SOURCE rows' side (GEMM and weight gradient contracted over |S_j| rows) when the plan holds the orientation for it
    rowsparse_loss_side: bool = True
    # ... which grad_support_plan builds when S_{j+1} has at least this many more rows than S_j per edge between them.  Break-even measured on
    # S-products (108.5 - 109.0 ms alike); S-pl10M sits at 0.35
    fwd0_rows_per_edge: float = 0.25
    # ... and the level moves at least this many edges: below, its kernels are latency-bound and two more launches cost more than the rows
    # save.  S-arxiv (2.3 * 10^5 edges at level 0): 3.35 -> 3.41 ms/step with the form; S-pubmed under hipGraph 0.600 -> 0.627; S-pl1M (10^6): 16.15 -> 16.0
    fwd0_min_edges: int = 1 << 19
    # rows-only forward: the layer below the last one takes its sum first (aggregate on S_1, GEMM on |S_1| rows, store rows; the backward's level 1 through
    # the source rows' side) once S_1's rows are entered by at least this many edges — below, the Z-first form on S_1 (one kernel) wins on launches.
    # S-arxiv (1.7 * 10^6 edges into S_1): 2.94 -> 3.11 ms/step with the sum first; S-products: 90.1 -> 89.4; S-pl10M (7.4 * 10^7): 125.3 -> 120.9
    sum_first_below_min_edges: int = 1 << 24
    # rows-only forward: not below this many nodes, even where a replayed step (graph.rowsparse_small_ok) takes the row-sparse backward — the last layer
    # as aggregate + transform + store + head on the loss rows is four launches for one.  S-pubmed config 2 (19 717 nodes, bf16 rows, hipGraph replay):
    # 0.795 ms/step with it, 0.752 without (round 6, one box, alternating runs; the round-5 tree the same: 0.803 / 0.748); S-arxiv (169 343): 3.38 -> 2.94 with it
    rows_only_min_nodes: int = 1 << 16
    # rows-only forward: the rows of the output that the caller promised not to read come back as NaN (True) or as zeros (False).  Costs nothing (the
    # rows are written either way); makes a broken `rows_only` promise loud: whoever reads TeacherGNN.out / res.commonEmb outside the loss rows sees NaN
    rows_only_poison: bool = True
    # mixed-in gradients one cb_trunk_input_bwd_multi_f32 launch gathers (deeper 'Initial' trunks accumulate layer by layer): kernel limit
    mix_max: int = 7
    # gather mode keeps every layer's [N, d] gradient alive until the input stage; allowed while that is below this share of the free memory
    gather_mem_frac: float = 0.25

    # ---- node-sharded exchange (dist.py) --------------------------------------------------------------------------------------------------
    # one node's dense work (GEMMs, elementwise) ~ this many edges' aggregation work per step: the edge-balanced partition's node weight.
    # S-pl10M kernel profile (HISTORY.md section 6)
    node_weight: int = 12
    # a rank pair leaves the plain pull for the push / pull cover only for >= this share fewer rows (a pushed row costs its owner an
    # aggregation over the edges it sums).  S-pl10M: 25 - 32 % fewer rows (cover); the ogbn-products shape: < 6 % (pull)
    cover_min_gain: float = 0.10
    # time slices of the exchange pipeline once an aggregation moves this many bytes of halo rows per rank (below, a slice's kernels are
    # launch-sized).  S-pl10M P = 8 (profiles/r03_shard_probe_S-pl10M_*.txt)
    slice_min_bytes: int = 64 << 20
    # a level orientation of the sharded row-sparse backward is built while the level keeps at most this share of the edges
    support_max_edge_frac: float = 0.9


T = Tuning()
