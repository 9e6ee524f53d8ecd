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
import torch.distributed as dist

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
    ap.add_argument('--cpu-baseline-mode', default='auto', choices=['auto', 'full', 'sample'],
                    help='full = one measured full-size oracle step on the host cores; sample = sub-sampled + scaled (fallback)')
    ap.add_argument('--cpu-budget-s', type=float, default=20.0, help='keep timing full-size CPU steps until this many seconds (max 3 steps)')
    ap.add_argument('--pmc-traffic', type=int, default=1, help='collect roofline.traffic with two rocprofv3 --pmc passes of the aggregation (N=1 only)')
    ap.add_argument('--ref-epochs', type=int, default=2, help='epochs of the reference epoch (2 train fwd + 1 bwd + 1 eval fwd) to time; 0 = skip')
    ap.add_argument('--hip-graph', type=int, default=0, help='replay the step as one hipGraph (pays off on launch-bound small graphs)')
    ap.add_argument('--se', default='000', help="whetherHasSE of the reference (per-layer structural embedding tables `le`): '000' or '111'")
    ap.add_argument('--extra', default='', help='further reference CLI flags, space separated (e.g. "--force_set_to_best_config=0 --type_trick=Residual")')
    ap.add_argument('--layers', type=int, default=3, help='num_layers (BASELINE config 2 = Pubmed, 2 layers)')
    ap.add_argument('--agg-dtype', default='f32', choices=['f32', 'bf16'], help='bf16 = build-extension storage of the gathered rows')
    ap.add_argument('--dense-backward', type=int, default=1, help='also time K steps with the dense backward (CB_LOSS_ROWS=0) after the timed region (N=1, eager)')
    ap.add_argument('--check-n1', type=int, default=1, help='N > 1: rank 0 first computes the single-GPU training loss of the same model / seeds; '
                                                             'the sharded forward must reproduce it (sharding.loss_matches_n1)')
    return ap.parse_args()


def make_args(dataset, extra=(), se='000', layers=3):
    from gnn_tail_generalization_amd.base_options import BaseOptions
    argv = [f'--dataset={dataset}', '--train_which=TeacherGNN', f'--num_layers={layers}', '--use_special_split=0', '--want_headtail=0',
            f'--whetherHasSE={se}', '--do_deg_analyze=0'] + list(extra)
    with contextlib.redirect_stdout(io.StringIO()):
        return BaseOptions().get_arguments(argv)


def _host_info():
    model = ''
    with contextlib.suppress(Exception):
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    avail = 0
    with contextlib.suppress(Exception):
        for line in open('/proc/meminfo'):
            if line.startswith('MemAvailable'):
                avail = int(line.split()[1]) * 1024
    return model, avail


def _oracle_modules():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import coldbrew_oracle as orc
    import oracle_c
    return orc, oracle_c


def _cpu_threads():
    n = os.cpu_count() or 1
    return max(1, min(n // 2 if n >= 16 else n, 128))      # one thread per physical core (SMT siblings only add contention)


def cpu_baseline_sample(a, args, full_nodes):
    """Fallback: oracle training step on a node-subsampled instance of the same synthetic family, scaled linearly."""
    orc, oracle_c = _oracle_modules()
    from gnn_tail_generalization_amd.data import synthetic_data
    cores = min(_cpu_threads(), 32)   # more threads only add barrier/NUMA cost on these small per-op sizes
    torch.set_num_threads(cores)
    oracle_c.set_num_threads(cores)
    n = min(a.cpu_sample_nodes, full_nodes)
    data = synthetic_data(a.dataset, seed=0, device='cpu', n_override=n if n < full_nodes else None)
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
    return {'value': (1.0 / dt) * scale, 'unit': 'steps/s', 'cores': cores, 'kind': 'port', 'measured': False,
            'sample': f'EXTRAPOLATED: {a.dataset} family sub-sampled to N={n} nodes / E={csr.E} edges ({k} timed steps, {dt:.3f} s/step); '
                      f'value = sample steps/s x {scale:.4g} (linear in nodes) to the full {full_nodes}-node workload; '
                      'dropout masks omitted (p=0) on the CPU leg'}


def cpu_baseline_full(a, args, trainer, sd0, graph_obj, budget_s):
    """MEASURED: the oracle's training step (torch CPU ops + the C/OpenMP aggregation of oracle/coldbrew_oracle.c) on the
    SAME full-size inputs, initial weights and dropout rate as the GPU leg (keep-masks drawn by the product's Philox
    generator and injected, as in the parity tests).  One thread per physical core."""
    from types import SimpleNamespace
    import numpy as np
    orc, oracle_c = _oracle_modules()
    from gnn_tail_generalization_amd import ops
    cores = _cpu_threads()
    torch.set_num_threads(cores)
    oracle_c.set_num_threads(cores)
    d = trainer.data
    n, E = int(d.x.shape[0]), int(d.edge_index.shape[1])
    t_setup = time.time()
    x, y, mask = d.x.cpu(), d.y.cpu(), d.train_mask.cpu()
    src, dst = d.edge_index[0].cpu().numpy(), d.edge_index[1].cpu().numpy()
    rowptr, col = oracle_c.csr_from_coo(dst, src, n)                 # by-dst CSR (GCN.py:93-94)
    if graph_obj.symmetric:
        rowptr_t, col_t = rowptr, col
    else:
        rowptr_t, col_t = oracle_c.csr_from_coo(src, dst, n)
    deg_in, deg_out = np.diff(rowptr), np.diff(rowptr_t)
    csr = SimpleNamespace(N=n, E=E, rowptr=rowptr, col=col, rowptr_t=rowptr_t, col_t=col_t, in_deg=deg_in, out_deg=deg_out,
                          src=src, dst=dst)
    p = float(args.dropout)
    cfg = orc.make_cfg(type_trick=args.type_trick, num_layers=args.num_layers, num_feats=args.num_feats, dim_hidden=args.dim_hidden,
                       num_classes=args.num_classes, dropout=p, res_alpha=args.res_alpha, se_reg=args.se_reg)
    masks = None
    if p > 0:
        L, H = args.num_layers, args.dim_hidden
        shapes = [(n, args.num_feats)] + [(n, H)] * L + [(n, H)]
        masks = [ops.dropout_keep_mask(s, p, 1234 + i, d.x.device).cpu() for i, s in enumerate(shapes)]
    sd = {k: v.clone() for k, v in sd0.items()}
    t_setup = time.time() - t_setup
    orc.set_aggregate(oracle_c.aggregate_sum)
    try:
        t0 = time.time()
        k = 0
        while k < 1 or (time.time() - t0 < budget_s and k < 3):
            orc.train_steps(cfg, sd, x, csr, y, mask, 1, lr=args.lr, weight_decay=args.weight_decay,
                            dropout_masks_per_step=[masks] if masks is not None else None)
            k += 1
        dt = (time.time() - t0) / k
    finally:
        orc.set_aggregate(None)
    model, _ = _host_info()
    return {'value': 1.0 / dt, 'unit': 'steps/s', 'cores': cores, 'kind': 'port', 'measured': True, 'cpu_model': model,
            'host_logical_cpus': os.cpu_count(), 'torch_threads': torch.get_num_threads(), 'omp_threads': cores,
            'sample': f'MEASURED at full size: {k} timed step(s) of the whole {a.dataset} workload (N={n}, E={E}, '
                      f'{dt:.2f} s/step; no extrapolation), same inputs, initial weights and dropout p={p} (injected keep-masks) '
                      f'as the GPU leg; host set-up (copies, CSR build, masks) {t_setup:.1f} s not timed'}


def cpu_baseline(a, args, trainer, sd0, graph_obj, full_nodes):
    """`cpu_baseline` object of the bench line.  full = one real full-size step (needs host RAM for ~45 activation-sized
    tensors); sample = the sub-sampled, linearly scaled fallback (labelled EXTRAPOLATED)."""
    mode = a.cpu_baseline_mode
    if mode == 'auto':
        _, avail = _host_info()
        need = 45 * full_nodes * max(args.dim_hidden, args.num_feats) * 4 * 1.5
        mode = 'full' if (avail >= need and (os.cpu_count() or 1) >= 32) else 'sample'
    if mode == 'full':
        return cpu_baseline_full(a, args, trainer, sd0, graph_obj, a.cpu_budget_s)
    return cpu_baseline_sample(a, args, full_nodes)


def pmc_traffic(dataset, fused=False):
    """HBM-side bytes of one aggregation launch from the PMC counters, collected by THIS run: two separate `rocprofv3 --pmc` passes
    (FETCH_SIZE, WRITE_SIZE; kernel trace only) over tools/bench_spmm.py on the same graph, corrected as MI355X_MICROARCH.md's
    HBM section prescribes (KB units; FETCH_SIZE doubled on gfx950 for 16 B/lane reads).  Returns (bytes, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    if 'rocprof' in os.environ.get('LD_PRELOAD', '') or os.environ.get('ROCP_TOOL_LIBRARIES'):
        return None, 'this process is itself being profiled: nested counter collection skipped'
    tool = os.path.join(ROOT, 'tools', 'bench_agg_gemm.py' if fused else 'bench_spmm.py')      # the dominant kernel of this run
    main_kernel = 'k_agg_gemm2' if fused else 'k_spmm_rows'
    n_arg = [] if dataset == 'S-pl10M' else None
    if n_arg is None:
        return None, 'PMC pass only wired for the S-pl10M workload'
    vals = {}
    tmp = tempfile.mkdtemp(prefix='cb_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, ctr)
            subprocess.run(['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable, tool,
                            '--iters', '3'] + (['--parts', '0'] if fused else []), cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
            if not files:
                return None, f'no counter file for {ctr}'
            acc, disp = {}, {}
            for r in csv.DictReader(open(files[0])):
                name = r['Kernel_Name']
                key = next((k for k in (main_kernel, 'k_spmm_hub_chunks', 'k_spmm_hub_finish') if k in name), None)
                if key is None or r['Counter_Name'] != ctr:
                    continue
                acc[key] = acc.get(key, 0.0) + float(r['Counter_Value'])
                disp.setdefault(key, set()).add(r['Dispatch_Id'])
            if main_kernel not in acc:
                return None, f'{ctr}: aggregation kernel not found in the counter file'
            vals[ctr] = sum(acc[k] / len(disp[k]) for k in acc)          # KB per aggregation launch (one dispatch of each kernel)
        return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0, (
            f'measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over {os.path.basename(tool)} '
            f'--iters 3 on the same graph; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over {main_kernel} + hub kernels, per launch; '
            'L2 memory-side requests (Infinity-Cache hits included)')
    except Exception as e:  # noqa: BLE001
        return None, f'PMC pass failed: {type(e).__name__}: {e}'[:300]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def graph_desc(dataset):
    from gnn_tail_generalization_amd.data import ALIASES, SYNTHETIC
    _, _, _, _, loops, gamma = SYNTHETIC[ALIASES.get(dataset, dataset)]
    return (f'Chung-Lu power law gamma={gamma}, symmetric' + (' + self-loops' if loops else ', no self-loops, every degree >= 1')
            + ', ids permuted, seed 0')


def reference_epoch_rate(t, args, epochs, sync):
    """Epochs/s of the reference's own epoch (train_net: run_trainSet with the metrics-only second train-mode forward of
    want_headtail=1, trainer_node_classification.py:397-413, + run_testSet's eval forward, :453-495) = 2 train forwards +
    1 backward + Adam + 1 eval forward + the accuracy reductions with their host syncs."""
    from gnn_tail_generalization_amd.utils import save_graph_analyze
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        save_graph_analyze(t.global_nodes(), t.data, 0, verbose=False)       # head / tail node sets (utils.py:680-729)
    torch.cuda.synchronize()
    analyze_s = time.perf_counter() - t0
    for name in ('large_deg_idx', 'small_deg_idx'):
        setattr(t.data, name, torch.as_tensor(getattr(t.data, name), device=t.device))
    old = args.want_headtail
    args.want_headtail = 1
    try:
        t.train_net()
        sync()
        t0 = time.perf_counter()
        for _ in range(epochs):
            t.train_net()
        sync()
        dt = time.perf_counter() - t0
    finally:
        args.want_headtail = old
    return {'epochs_per_sec_reference_epoch': epochs / dt, 'reference_epoch_ms': dt / epochs * 1e3,
            'head_tail_split_s': analyze_s}


N1_SEEDS = list(range(424200, 424232))     # dropout seeds of the loss_matches_n1 forward (same on the single-GPU and the sharded side)


def n1_reference(a, local_rank, rank):
    """Rank 0: the single-GPU trainer on the whole graph -> (its initial state_dict + the training loss of one train-mode forward
    with fixed dropout seeds, the loaded data).  Other ranks: (None, None).  Any failure (e.g. the whole graph does not fit next to
    nothing else) turns the check off instead of ending the bench."""
    if rank != 0:
        return None, None
    from gnn_tail_generalization_amd import ops
    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    try:
        args1 = make_args(a.dataset, [f'--manual_assign_GPU={local_rank}', f'--agg_dtype={a.agg_dtype}'] + a.extra.split(), se=a.se, layers=a.layers)
        torch.manual_seed(0)
        t1 = tnc.trainer(args1, 0)
        t1.setup_teacherGNN()
        sd = {k: v.detach().cpu().clone() for k, v in t1.teacherGNN.state_dict().items()}
        t1.teacherGNN.train()
        ops._seed_override[:] = list(N1_SEEDS)
        with torch.no_grad():
            loss = float(t1.training_loss())
        ops._seed_override[:] = []
        data = t1.data
        t1.teacherGNN = t1.optimizer = None
        del t1
        torch.cuda.empty_cache()
        return {'sd': sd, 'loss': loss}, data
    except Exception as e:  # noqa: BLE001
        ops._seed_override[:] = []
        torch.cuda.empty_cache()
        print(f'[bench] single-GPU reference for loss_matches_n1 failed: {type(e).__name__}: {e}', file=sys.stderr)
        return None, None


def sharding_report(t, n1, a, dev):
    """Self-diagnosis of an N > 1 run, gathered from every rank: who took part, what each rank owns and exchanges, and whether the
    sharded forward reproduces the single-GPU loss (same weights, same dropout seeds)."""
    from gnn_tail_generalization_amd import dist as cbdist
    from gnn_tail_generalization_amd import norms_hip, ops
    world, rank = t.world, t.rank
    sg = t.sgraph
    plan = sg.f.plan
    mine = torch.tensor([rank, sg.N, sg.f.E, plan.n_halo if plan is not None else 0,
                         sg.f.interior.E if sg.f.interior is not None else sg.f.E,
                         max(plan.recv_counts_all) if plan is not None and world > 1 else 0,
                         max(plan.send_counts_all) if plan is not None and world > 1 else 0], dtype=torch.int64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        if dist.get_backend() == 'gloo':
            dist.all_gather(rows, mine)
        else:
            dr = [torch.zeros_like(mine, device=dev) for _ in range(world)]
            dist.all_gather(dr, mine.to(dev))
            rows = [r.cpu() for r in dr]
    else:
        rows = [mine]
    rep = {'ranks_seen': [int(r[0]) for r in rows], 'rows_per_rank': [int(r[1]) for r in rows], 'edges_per_rank': [int(r[2]) for r in rows],
           'halo_rows_per_rank': [int(r[3]) for r in rows], 'interior_edges_per_rank': [int(r[4]) for r in rows],
           'max_rows_on_one_link_recv_send': [[int(r[5]), int(r[6])] for r in rows],
           'partition': t.part.kind, 'exchange': sg.exchange_kind, 'overlap': bool(sg.overlap), 'wire': sg.wire,
           'slices': plan.n_slices if plan is not None else 1, 'symmetric': bool(sg.symmetric),
           'cover': bool(getattr(sg, 'cover', False)),      # push / pull vertex cover of the remote edges (dist.CoverPlan)
           'halo_rows_pull_only_rank0': int(getattr(plan, 'n_pull_only', plan.n_halo if plan is not None else 0)) if plan is not None else 0,
           'backend': dist.get_backend() if world > 1 or dist.is_initialized() else 'none'}
    # loss_matches_n1: every rank must walk the same collectives, so the decision is broadcast
    box = [None if n1 is None else {'loss': n1['loss'], 'sd': n1['sd']}]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        rep['loss_matches_n1'] = None
        return rep
    keep = {k: v.detach().clone() for k, v in t.teacherGNN.state_dict().items()}
    t.load_full_state_dict({k: v.to(dev) for k, v in box[0]['sd'].items()})
    t.teacherGNN.train()
    ops._seed_override[:] = list(N1_SEEDS)
    with torch.no_grad(), norms_hip.row_sharding(t.group, t.global_nodes()):
        loss = t.training_loss().detach().clone()
    ops._seed_override[:] = []
    cbdist._all_reduce(loss, group=t.group)
    t.teacherGNN.load_state_dict(keep)
    got, want = float(loss), box[0]['loss']
    rep['loss_n1'], rep['loss_sharded'] = want, got
    rep['loss_matches_n1'] = bool(abs(got - want) <= 1e-4 * max(1.0, abs(want)))
    return rep


def self_launch_argv(a, argv, env, n_devices):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): the command line that runs this file as N
    ranks under torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous), or None when nothing has to be re-launched.  A driver that
    launches the ranks itself (WORLD_SIZE set) is left alone.  Fewer visible devices than ranks is an error unless the gloo dry run of the
    N > 1 code path is asked for (COLDBREW_DIST_BACKEND=gloo: ranks share the devices there are)."""
    if a.gpus <= 1 or 'WORLD_SIZE' in env:
        return None
    if env.get('COLDBREW_DIST_BACKEND', 'nccl') != 'gloo' and n_devices < a.gpus:
        raise SystemExit(f'bench.py --gpus {a.gpus}: only {n_devices} device(s) visible (one rank per GPU over RCCL; '
                         'COLDBREW_DIST_BACKEND=gloo runs the N > 1 code path on fewer devices as a dry run)')
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def roofline_families(recs, steps):
    """Per-family table of the timed aggregation launches.  A family = launches of one kernel form on one CSR (same kind, edges, rows).
    `achieved` / `frac` are on SURVEY.md 8(d)'s bytes ONLY (E (d s + 4) + N (d s + 4) [+ 4 N] [+ d s]) / the average launch time / 8 TB/s;
    the streams a fused store (mixed-in row, mask words) and a dense tail (its output, addend, row scale) add are compulsory traffic of
    those kernels but not 8(d) bytes: they only enter `frac_incl_fused_streams`."""
    fams = {}
    for r in recs:
        fams.setdefault((r['kind'], r['edges'], r['rows']), []).append(r)
    out = []
    for (kind, edges, rows), rs in fams.items():
        n = len(rs)
        avg = sum(r['ms'] for r in rs) / n
        agg, store, tail = (sum(r[k] for r in rs) / n for k in ('agg', 'store', 'tail'))
        ach = agg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        ach_all = (agg + store + tail) / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        out.append({'kind': kind, 'edges_per_launch': edges, 'rows': rows, 'launches_timed': n, 'launches_per_step': n / max(steps, 1),
                    'avg_launch_ms': avg, 'total_ms_per_step': sum(r['ms'] for r in rs) / max(steps, 1), 'survey_8d_bytes_per_launch': agg,
                    'fused_store_bytes_per_launch': store, 'dense_tail_bytes_per_launch': tail, 'achieved': ach, 'frac': ach / HBM_PEAK_GBS,
                    'achieved_incl_fused_streams': ach_all, 'frac_incl_fused_streams': ach_all / HBM_PEAK_GBS})
    out.sort(key=lambda f: -f['total_ms_per_step'])
    return out


KERNEL_OF = {'plain': 'k_spmm_rows (+hub kernels)', 'colscale': 'k_spmm_rows<colscale> (+hub kernels)', 'fused_store': 'k_spmm_rows<fused store> (+hub kernels)',
             'agg_gemm': 'k_agg_gemm2<false> (+hub kernels): aggregation + the dX 256x256 contraction on the matrix cores in one kernel',
             'agg_gemm_fused': 'k_agg_gemm2<true> (+hub kernels): aggregation with the trunk store + the next 256x256 dense transform on the matrix cores in one kernel',
             'agg_gemm_fused_eval': 'k_agg_gemm2<true> evaluation form (+hub kernels)',
             'agg_gemm_head': 'k_agg_gemm2<true, narrow> (+hub kernels): last layer store + the output Linear (256 x C) in one kernel',
             'agg_gemm_head_eval': 'k_agg_gemm2<true, narrow> evaluation form (+hub kernels)',
             'agg_gemm_trunkbwd': 'k_agg_gemm2<TB> (+hub kernels)',
             'store_bwd': 'k_spmm_rows<store backward in the epilogue> (+hub kernels)'}


def main():
    a = parse()
    relaunch = self_launch_argv(a, sys.argv[1:], os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0)
    if relaunch is not None:
        sys.stdout.flush()
        os.execv(relaunch[0], relaunch)
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
    local_rank %= torch.cuda.device_count()            # >1 rank per GPU only in the gloo dry run of the N>1 code path
    torch.cuda.set_device(local_rank)
    dev = torch.device(f'cuda:{local_rank}')
    sharded = world > 1 or os.environ.get('COLDBREW_FORCE_SHARDED') == '1'   # the latter: exercise the sharded code on 1 GPU
    if sharded:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = os.environ.get('COLDBREW_DIST_BACKEND', 'nccl')      # 'gloo': dry run of this file's N>1 path on a 1-GPU box
        if backend == 'nccl':
            from gnn_tail_generalization_amd.dist import init_rccl
            init_rccl(rank, world, dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if a.gpus != world and rank == 0:      # (a launcher set WORLD_SIZE: its word counts; without one, --gpus N re-launches itself above)
        print(f'[bench] --gpus {a.gpus} but WORLD_SIZE={world}: using WORLD_SIZE', file=sys.stderr)

    from gnn_tail_generalization_amd import trainer_node_classification as tnc
    args = make_args(a.dataset, [f'--manual_assign_GPU={local_rank}', f'--agg_dtype={a.agg_dtype}'] + a.extra.split(), se=a.se, layers=a.layers)
    torch.manual_seed(0)
    n1 = None
    with contextlib.redirect_stdout(io.StringIO()):
        if sharded:
            from gnn_tail_generalization_amd import dist as cbdist
            data0 = None
            if a.check_n1 and a.se == '000':
                n1, data0 = n1_reference(a, local_rank, rank)          # rank 0: single-GPU loss on the whole graph; its data is reused below
            t = cbdist.ShardedTrainer(args, 0, data=data0)
            del data0
        else:
            t = tnc.trainer(args, 0)
        t.setup_teacherGNN()
    sharding = sharding_report(t, n1, a, dev) if sharded else None
    graph_obj = t.graph()
    n_nodes, n_edges = t.global_nodes(), t.global_edges()
    L = args.num_layers
    sd0 = {k: v.detach().cpu().clone() for k, v in t.teacherGNN.state_dict().items()} if not sharded else None
    torch.cuda.reset_peak_memory_stats(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    rng0 = torch.get_rng_state()           # (the dense-backward leg restarts from here: same weights, same dropout seeds)
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
    if sharded:
        graph_obj.exchange_log = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = t.train_step()
    sync()
    dt = time.perf_counter() - t0
    if not use_graph:
        prof, graph_obj.profile = graph_obj.profile, None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        cbdist._all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_step = dt / a.steps * 1e3
    peak_mem = torch.cuda.max_memory_allocated(dev)
    if world > 1:
        pm = torch.tensor([peak_mem], device=dev, dtype=torch.float64)
        cbdist._all_reduce(pm, op=dist.ReduceOp.MAX)
        peak_mem = float(pm.item())
    # every aggregation launch of the timed steps (graph.prof_rec): HIP events on the launch stream, SURVEY 8(d) bytes, the fused streams' bytes
    # apart, the edges the launch walked
    recs = [dict(r, ms=r['ev'][0].elapsed_time(r['ev'][1])) for r in prof]
    for r in recs:
        del r['ev']
    spmm_ms = [r['ms'] for r in recs]
    families = roofline_families(recs, a.steps)
    fam_main = families[0] if families else {'kind': 'plain', 'launches_timed': 0, 'avg_launch_ms': 0.0, 'survey_8d_bytes_per_launch': 0.0, 'achieved': 0.0,
                                             'frac': 0.0, 'achieved_incl_fused_streams': 0.0, 'frac_incl_fused_streams': 0.0, 'edges_per_launch': 0}
    edges_walked = float(sum(r['edges'] for r in recs))          # this rank's launches; node-sharded: summed over the ranks below
    if world > 1:
        ew = torch.tensor([edges_walked], device=dev, dtype=torch.float64)
        cbdist._all_reduce(ew)
        edges_walked = float(ew.item())
    row_sparse = (not sharded) and getattr(graph_obj, '_support_plan', None) is not None
    # the DENSE backward of the same run (CB_LOSS_ROWS=0), after the timed region: the trainer is put back to the initial weights, moments and
    # RNG state and runs warm-up + K steps again, so its final_loss is the bit-for-bit regression witness of rounds 2 - 4 (same steps, same
    # dropout seeds, every backward aggregation over all rows) and its ms_per_step the like-for-like figure beside `value`
    def restarted_leg(env, note, edges_per_step):
        """warm-up + K steps of the same trainer put back to the initial weights / moments / RNG state, under the switches `env`."""
        old_env = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            graph_obj.profile = None
            with torch.no_grad():
                t.teacherGNN.load_state_dict({k: v.to(dev) for k, v in sd0.items()})
            t.optimizer.state.clear()
            t.optimizer.zero_grad(set_to_none=True)
            torch.set_rng_state(rng0)
            for _ in range(a.warmup):
                t.train_step()
            sync()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                dloss = t.train_step()
            sync()
            dms = (time.perf_counter() - t1) / a.steps * 1e3
            leg = {'ms_per_step': dms, 'value': 1e3 / dms, 'unit': 'steps/s', 'steps': a.steps, 'warmup': a.warmup, 'final_loss': float(dloss), 'note': note}
            if edges_per_step:
                leg['aggregated_edges_per_sec'] = edges_per_step * 1e3 / dms
            return leg
        finally:
            for k, v in old_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    dense_bwd = all_rows_fwd = None
    # (the graph counts the training forwards that evaluated their last layer on the loss rows: trunk._last_layer_on_loss_rows, stack.py)
    rows_only_fwd = bool(row_sparse and getattr(graph_obj, 'rows_only_forwards', 0) > 0)
    if a.dense_backward and row_sparse and not use_graph and os.environ.get('CB_LOSS_ROWS', '1') != '0':
        if rows_only_fwd and os.environ.get('CB_ROWS_ONLY_FWD', '1') != '0':
            all_rows_fwd = restarted_leg({'CB_ROWS_ONLY_FWD': '0'},
                                         'the same trainer restarted from the initial weights / moments / RNG state with CB_ROWS_ONLY_FWD=0: every row of every '
                                         'layer is evaluated in the training forward, the backward stays row-sparse (the step of this bench before the rows-only '
                                         'forward), warm-up + K steps, outside the timed region', 0)
        dense_bwd = restarted_leg({'CB_LOSS_ROWS': '0'},
                                  'the same trainer restarted from the initial weights / moments / RNG state with CB_LOSS_ROWS=0 (every row of every layer in the '
                                  'forward, every backward aggregation over all rows: the reference\'s amount of work), warm-up + K steps, outside the timed '
                                  'region; final_loss is bit-comparable with the final_loss of rounds 2 - 4', n_edges * 2 * L)
    ref_epoch = None
    if a.ref_epochs > 0 and not sharded and not use_graph:
        ref_epoch = reference_epoch_rate(t, args, a.ref_epochs, sync)
    if sharded:
        xlog, graph_obj.exchange_log = graph_obj.exchange_log, None
        xm = [e0.elapsed_time(e1) for e0, e1 in xlog]
        stat = torch.tensor([sum(xm) / max(len(xm), 1), float(len(xm)) / max(a.steps, 1), sum(spmm_ms) / max(a.steps, 1)], device=dev, dtype=torch.float64)
        allst = [torch.zeros_like(stat) for _ in range(world)]
        if world > 1:
            if dist.get_backend() == 'gloo':
                hs = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(hs, stat.cpu())
                allst = hs
            else:
                dist.all_gather(allst, stat)
        else:
            allst = [stat]
        sharding['exchange_span_ms_per_rank'] = [round(float(v[0]), 3) for v in allst]     # first send issued -> last slice waited for, per aggregation
        sharding['exchanges_per_step'] = [round(float(v[1]), 2) for v in allst]
        sharding['local_aggregation_kernel_ms_per_step_per_rank'] = [round(float(v[2]), 3) for v in allst]
        # (rank 0's view) how the aggregations of the timed steps ran: merged first pass ([local | slice 0]) vs a separate interior pass, and the
        # levels of the row-sparse backward with the rows this rank reads / writes on each (None = all local rows)
        sg_ = t.sgraph
        sharding['first_pass_rank0'] = {'merged': int(sg_.merged_passes), 'interior_only': int(sg_.interior_passes)}
        cache_ = getattr(sg_, '_support_cache', None)
        sharding['row_sparse_levels_rank0'] = ([{'edges': int(lv.orient.E), 'rows_read': (lv.src.n if lv.src is not None else None),
                                                 'rows_written': (lv.dst.n if lv.dst is not None else None), 'halo_rows': int(lv.orient.plan.n_halo)}
                                                for lv in cache_[2]] if cache_ is not None else [])
        f0_ = getattr(sg_, '_fwd0_cache', None)
        sharding['rows_only_forward_rank0'] = ({'forwards': int(sg_.rows_only_forwards), 'edges': int(f0_[1].E), 'rows_written': int(f0_[0][0].src.n),
                                                'halo_rows': int(f0_[1].plan.n_halo), 'halo_rows_full_forward': int(sg_.f.plan.n_halo)}
                                               if f0_ is not None else {'forwards': int(getattr(sg_, 'rows_only_forwards', 0))})
        sharding['instage_folds_rank0'] = int(getattr(sg_, 'instage_folds', 0))      # backwards whose input stage ran inside the input Linear's weight gradient
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    # HBM-side bytes of the aggregation from the PMC counters: not collected by this process (counters need a rocprofv3
    # run); the committed profile's figure is carried under its own name, `traffic` stays null unless measured in-run
    traffic_profile = None
    pmc = os.path.join(ROOT, 'profiles', 'spmm_pmc_traffic.json')
    if os.path.isfile(pmc):
        with contextlib.suppress(Exception):
            traffic_profile = json.load(open(pmc)).get('hbm_bytes_per_launch')
    if sharded:
        par = (f'node-sharded x{world}, 1-D row partition ({t.part.kind}), RCCL {t.sgraph.exchange_kind} exchange'
               f'{" overlapped with the interior-column aggregation" if t.sgraph.overlap else ""}; roofline figures are per rank (rank 0 shard)')
    else:
        par = 'single GPU'
    out = {
        'metric': 'teachergnn_fullgraph_train_steps_per_sec', 'value': a.steps / dt, 'unit': 'steps/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if a.agg_dtype == 'f32' else 'f32 (bf16-stored aggregation rows)',
        'data': 'synthetic',
        'aggregated_edges_per_sec': edges_walked / dt,
        'nominal_edges_per_sec': n_edges * 2 * L * a.steps / dt,
        'edges_note': ('aggregated_edges_per_sec = the edges the timed aggregation launches actually walked (sum of the E of the CSR each launch ran on, '
                       f'{edges_walked / max(a.steps, 1) / max(n_edges, 1):.3f} x E per step) / time; nominal_edges_per_sec = 2L x E per step, the '
                       "reference's amount (every forward and backward aggregation over all edges)"),
        'final_loss': float(loss),
        'config': {'workload': f'{a.dataset}: N={n_nodes} nodes, E={n_edges} edge_index columns ({graph_desc(a.dataset)}), '
                               f'F={args.num_feats} H={args.dim_hidden} '
                               f'C={args.num_classes} L={L}, type_trick={args.type_trick} (residual mode), whetherHasSE={a.se}, '
                               f'dropout={args.dropout}, Adam lr={args.lr}; step = fwd+loss+bwd+Adam, 2L={2 * L} aggregations',
                   'launch': 'one hipGraph replay per step' if use_graph else 'eager launches',
                   'gemm': ('fp32-input MFMA' if os.environ.get('CB_GEMM_PLAIN_F32') else
                            'fp32 operands as three exact bf16 limbs, 6 bf16 MFMA products, fp32 accumulate (error <= fp32 GEMM)'),
                   'parallelism': par},
        'roofline': {'bound': 'hbm',
                     'kernel': f"{KERNEL_OF.get(fam_main['kind'], fam_main['kind'])}; d=256 {a.agg_dtype} source rows, f32 accumulate, "
                               f"{fam_main['edges_per_launch']} edges per launch",
                     'achieved': fam_main['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': fam_main['frac'], 'traffic': None,
                     'traffic_from_profile': traffic_profile,
                     'launches_timed': fam_main['launches_timed'], 'avg_launch_ms': fam_main['avg_launch_ms'],
                     'algorithmic_bytes_per_launch': fam_main['survey_8d_bytes_per_launch'],
                     'achieved_incl_fused_streams': fam_main['achieved_incl_fused_streams'], 'frac_incl_fused_streams': fam_main['frac_incl_fused_streams'],
                     'per_family': families,
                     'note': ('achieved / frac = SURVEY 8(d) bytes of the aggregation ONLY / average launch time (HIP events on the launch stream inside the '
                              'timed steps) / 8 TB/s, for the family with the largest time per step; per_family lists every kernel form x CSR of the step the '
                              "same way.  *_incl_fused_streams add the kernel's other compulsory streams (a fused store's mixed-in row + mask words, a dense "
                              "tail's output / addend / row scale) — traffic those kernels must move, not 8(d) bytes")},
    }
    if dense_bwd is not None:
        out['dense_backward'] = dense_bwd
    if all_rows_fwd is not None:
        out['all_rows_forward'] = all_rows_fwd
    if rows_only_fwd and os.environ.get('CB_ROWS_ONLY_FWD', '1') != '0':
        out['config']['forward'] = ('rows-only: the trainer reads the logits in the train rows only (trainer_node_classification.py:390-391) and says so '
                                    '(rows_only=True); the training forward then evaluates its last layer on those rows and the layer below on the rows that one '
                                    'reads (same loss and gradients; metrics / evaluation forwards always compute every row).  CB_ROWS_ONLY_FWD=0: every row, timed '
                                    'beside it as `all_rows_forward`; with the dense backward on top: `dense_backward`')
    if row_sparse:
        out['config']['backward'] = ('row-sparse: under the masked loss the gradient is exactly zero outside the rows the train rows reach after j hops; the levels '
                                     'of the backward whose support is <= 70 % of the rows (train rows, their neighbours) run on compact matrices and gather only '
                                     'those rows (the claim is verified on the device every step; CB_LOSS_ROWS=0: dense backward, timed beside it as `dense_backward`)')
        if getattr(graph_obj, 'instage_folds', 0) > 0:
            out['config']['backward'] += ('; input stage without a pass of its own: the level that writes all rows folds the mix gradients in its epilogue, the input '
                                          "Linear's weight gradient computes (X0 > 0) * (dropout_bwd(g) + fold) while it stages it (CB_INSTAGE_FOLD=0: the separate pass)")
    out['peak_mem_gb'] = peak_mem / 2 ** 30
    if sharding is not None:
        out['sharding'] = sharding
    if ref_epoch is not None:
        out.update(ref_epoch)
    if a.cpu_baseline and world == 1 and not sharded:
        out['cpu_baseline'] = cpu_baseline(a, args, t, sd0, graph_obj, n_nodes)
    if a.pmc_traffic and world == 1 and not sharded:
        del t, graph_obj                     # the PMC passes build their own copy of the graph in a child process
        torch.cuda.empty_cache()
        traffic, note = pmc_traffic(a.dataset, fused=fam_main['kind'].startswith('agg_gemm'))
        out['roofline']['traffic'] = traffic
        out['roofline']['traffic_note'] = note
    os.write(json_fd, (json.dumps(out) + '\n').encode())


if __name__ == '__main__':
    main()
