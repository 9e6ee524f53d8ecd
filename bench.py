"""Benchmark of the TeacherGNN hot path (BASELINE.json metric: full-graph training steps/s +
aggregated edges/s on the synthetic power-law graph, 1/2/4/8 MI355X).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = forward + loss + backward + Adam step in train mode on the whole graph
(trainer_node_classification.run_trainSet without the head/tail metrics forward).  Inputs are
generated on the device before the timed region.  With N > 1 the SAME graph is node-sharded over
the ranks (strong scaling) with RCCL halo exchange.  Rank 0 prints one JSON line.

`roofline`: the dominant kernel (k_spmm_rows, the sum-aggregation) timed live with HIP events on its
launch stream inside the timed steps; achieved = algorithmic bytes (SURVEY.md §8d:
E*(d*4+4) + N*(d*4+4) [+4N row scale] [+d*4 bias]) / average launch duration.
`cpu_baseline`: the oracle (CPU restatement, kind "port") timed on the host cores on a bounded sample.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--dataset', default='S-pl10M', help='synthetic workload (S-pl10M = BASELINE headline config)')
    ap.add_argument('--cpu-baseline', type=int, default=1)
    ap.add_argument('--cpu-sample-nodes', type=int, default=200000)
    ap.add_argument('--hip-graph', type=int, default=0, help='replay the step as one hipGraph (pays off on launch-bound small graphs)')
    ap.add_argument('--agg-dtype', default='f32', choices=['f32', 'bf16'], help='bf16 = build-extension storage of the gathered rows')
    return ap.parse_args()


def make_args(dataset, extra=()):
    from gnn_tail_generalization_amd.base_options import BaseOptions
    argv = [f'--dataset={dataset}', '--train_which=TeacherGNN', '--num_layers=3', '--use_special_split=0', '--want_headtail=0',
            '--whetherHasSE=000', '--do_deg_analyze=0'] + list(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        return BaseOptions().get_arguments(argv)


def cpu_baseline(a, full_nodes):
    """Oracle training step (fwd + loss + bwd + Adam) on a node-subsampled instance of the same
    synthetic family, all host cores (torch intra-op + OpenMP aggregation)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import coldbrew_oracle as orc
    import oracle_c
    from gnn_tail_generalization_amd.data import synthetic_data
    cores = min(os.cpu_count() or 1, 32)   # more threads only add barrier/NUMA cost on these small per-op sizes
    torch.set_num_threads(cores)
    n = min(a.cpu_sample_nodes, full_nodes)
    data = synthetic_data(a.dataset, seed=0, device='cpu', n_override=n if n < full_nodes else None)
    args = make_args(a.dataset)
    cfg = orc.make_cfg(type_trick=args.type_trick, num_layers=args.num_layers, num_feats=args.num_feats, dim_hidden=args.dim_hidden,
                       num_classes=args.num_classes, dropout=0.0, res_alpha=args.res_alpha, se_reg=args.se_reg)
    torch.manual_seed(0)
    H, F_, C, L = args.dim_hidden, args.num_feats, args.num_classes, args.num_layers
    sd = {'model.model.layers_MLP.0.weight': torch.randn(H, F_) * 0.05, 'model.model.layers_MLP.0.bias': torch.zeros(H),
          'model.model.layers_MLP.1.weight': torch.randn(C, H) * 0.05, 'model.model.layers_MLP.1.bias': torch.zeros(C)}
    for i in range(L):
        sd[f'model.model.layers_GCN.{i}.weight'] = torch.randn(H, H) * 0.05
        sd[f'model.model.layers_GCN.{i}.bias'] = torch.zeros(H)
    csr = orc.build_csr(data.edge_index, n)
    orc.set_aggregate(oracle_c.aggregate_sum)
    try:
        orc.train_steps(cfg, sd, data.x, csr, data.y, data.train_mask, 1, lr=args.lr, weight_decay=args.weight_decay)   # warm-up
        t0 = time.time()
        k = 0
        while k < 2 or (time.time() - t0 < 10.0 and k < 20):
            orc.train_steps(cfg, sd, data.x, csr, data.y, data.train_mask, 1, lr=args.lr, weight_decay=args.weight_decay)
            k += 1
        dt = (time.time() - t0) / k
    finally:
        orc.set_aggregate(None)
    scale = n / full_nodes
    return {'value': (1.0 / dt) * scale, 'unit': 'steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{a.dataset} family sub-sampled to N={n} nodes / E={csr.E} edges ({k} timed steps, {dt:.3f} s/step); '
                      f'value = sample steps/s x {scale:.4g} (linear in nodes) to the full {full_nodes}-node workload; '
                      'dropout masks omitted (p=0) on the CPU leg'}


def main():
    a = parse()
    # Libraries (RCCL banner, rocm notices) write to the C stdout and flush it at exit, i.e. AFTER Python's
    # prints; keep a private copy of fd 1 for the JSON line and route everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False); the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device(f'cuda:{local_rank}')
    sharded = world > 1 or os.environ.get('COLDBREW_FORCE_SHARDED') == '1'   # the latter: exercise the sharded code on 1 GPU
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    if a.gpus != world and rank == 0:
        print(f'[bench] --gpus {a.gpus} but WORLD_SIZE={world}: using WORLD_SIZE', file=sys.stderr)

    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    args = make_args(a.dataset, [f'--manual_assign_GPU={local_rank}', f'--agg_dtype={a.agg_dtype}'])
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        if sharded:
            from gnn_tail_generalization_amd import dist as cbdist
            t = cbdist.ShardedTrainer(args, 0)
        else:
            t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    graph_obj = t.graph()
    n_nodes, n_edges = t.global_nodes(), t.global_edges()
    L = args.num_layers

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        t.train_step()
    sync()
    use_graph = bool(a.hip_graph) and not sharded
    if use_graph:
        # kernel events cannot be recorded inside a captured graph: time the aggregation in an eager pass of the
        # same K steps first, then capture and time the replays
        graph_obj.profile = []
        for _ in range(a.steps):
            t.train_step()
        sync()
        prof, graph_obj.profile = graph_obj.profile, None
        t.enable_hip_graph(warmup=1)
        t.train_step()
        sync()
    else:
        graph_obj.profile = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = t.train_step()
    sync()
    dt = time.perf_counter() - t0
    if not use_graph:
        prof, graph_obj.profile = graph_obj.profile, None
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt / a.steps * 1e3
    spmm_ms = [e0.elapsed_time(e1) for e0, e1, _, _ in prof]
    spmm_bytes = [b for _, _, b, _ in prof]                  # E(ds+4) + N(ds+4) [+4N] [+ds] per launch
    extra_bytes = [x for _, _, _, x in prof]                 # fused forward store: mixed-in X0 row + ReLU mask bits
    avg_ms = sum(spmm_ms) / max(len(spmm_ms), 1)
    avg_bytes = sum(spmm_bytes) / max(len(spmm_bytes), 1)
    avg_extra = sum(extra_bytes) / max(len(extra_bytes), 1)
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    achieved_incl = (avg_bytes + avg_extra) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    if sharded:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'spmm_pmc_traffic.json')
    if os.path.isfile(pmc):
        with contextlib.suppress(Exception):
            traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
    out = {
        'metric': 'teachergnn_fullgraph_train_steps_per_sec', 'value': a.steps / dt, 'unit': 'steps/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if a.agg_dtype == 'f32' else 'f32 (bf16-stored aggregation rows)',
        'data': 'synthetic',
        'aggregated_edges_per_sec': n_edges * 2 * L * a.steps / dt,
        'final_loss': float(loss),
        'config': {'workload': f'{a.dataset}: N={n_nodes} nodes, E={n_edges} edge_index columns (Chung-Lu power law gamma=2.3, '
                               f'symmetric + self-loops, ids permuted, seed 0), F={args.num_feats} H={args.dim_hidden} '
                               f'C={args.num_classes} L={L}, type_trick={args.type_trick} (residual mode), whetherHasSE=000, '
                               f'dropout={args.dropout}, Adam lr={args.lr}; step = fwd+loss+bwd+Adam, 2L={2 * L} aggregations',
                   'launch': 'one hipGraph replay per step' if use_graph else 'eager launches',
                   'gemm': ('fp32-input MFMA' if os.environ.get('CB_GEMM_PLAIN_F32') else
                            'fp32 operands as three exact bf16 limbs, 6 bf16 MFMA products, fp32 accumulate (error <= fp32 GEMM)'),
                   'parallelism': 'single GPU' if not sharded else f'node-sharded x{world} (RCCL all-gather exchange)'},
        'roofline': {'bound': 'hbm', 'kernel': f'k_spmm_rows (+hub kernels) d=256 {a.agg_dtype} source rows, f32 accumulate', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                     'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                     'launches_timed': len(spmm_ms), 'avg_launch_ms': avg_ms, 'algorithmic_bytes_per_launch': avg_bytes,
                     'fused_epilogue_bytes_per_launch': avg_extra, 'achieved_incl_fused_epilogue': achieved_incl},
    }
    if a.cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(a, n_nodes)
    os.write(json_fd, (json.dumps(out) + '\n').encode())


if __name__ == '__main__':
    main()
