import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from conftest import load_golden
from gnn_tail_generalization_amd.graph import CSRGraph
DEV = 'cuda:0'
g = load_golden('case_graph_asym_multi')
n = g['cfg']['N_nodes']
G = CSRGraph(g['edge_index'].to(DEV), n)
for d in (40, 20):
    h = torch.randn(n, d, device=DEV)
    out = G.spmm(h)
    os.environ['X'] = '1'
    A = torch.zeros(n, n, device=DEV)
    A.index_put_((g['edge_index'][1].to(DEV), g['edge_index'][0].to(DEV)), torch.ones(g['edge_index'].shape[1], device=DEV), accumulate=True)
    ref = A @ h
    bad = ((out - ref).abs().max(1).values > 1e-4).nonzero().flatten().tolist()
    rp = G.rowptr.tolist()
    print('d', d, 'n', n, 'bad rows', bad)
    print('deg of bad', [rp[r + 1] - rp[r] for r in bad])
    print('rowptr[0:70]', rp[:70])
    for r in bad[:4]:
        print(r, 'got', out[r, :4].tolist(), 'ref', ref[r, :4].tolist(), 'got/next-ref', ref[min(r + 1, n - 1), :4].tolist(), ref[max(r - 1, 0), :4].tolist())
