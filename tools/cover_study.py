import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import scipy.sparse as sp
from scipy.sparse.csgraph import maximum_bipartite_matching
from gnn_tail_generalization_amd import dist as cbdist
from gnn_tail_generalization_amd.data import powerlaw_pairs, SYNTHETIC

name = sys.argv[1] if len(sys.argv) > 1 else 'S-pl1M'
n, f, c, pairs, loops, gamma = SYNTHETIC[name]
gen = torch.Generator().manual_seed(0)
t0 = time.time()
p = powerlaw_pairs(n, pairs, gamma, gen, 'cpu', cover=not loops)
src = torch.cat([p[0], p[1]]); dst = torch.cat([p[1], p[0]])
if loops:
    src = torch.cat([src, torch.arange(n)]); dst = torch.cat([dst, torch.arange(n)])
print(name, 'edges', src.numel(), 'gen s', time.time() - t0, flush=True)
in_deg = torch.bincount(dst, minlength=n)

def heuristic(u, v, rounds=3):
    # u, v: compact ids (0..nu), (0..nv) of the pair graph's edges
    nu, nv = int(u.max()) + 1, int(v.max()) + 1
    du = torch.bincount(u, minlength=nu); dv = torch.bincount(v, minlength=nv)
    pull = du[u] >= dv[v]
    res = []
    for it in range(rounds):
        S = torch.zeros(nu, dtype=torch.bool); S[u[pull]] = True
        res.append(int(S.sum()) + int(torch.unique(v[~pull]).numel()))
        pull = pull | S[u]                      # sources already sent absorb all their edges
        T = torch.zeros(nv, dtype=torch.bool); T[v[~pull]] = True
        pull = pull & ~T[v]                     # destinations already pushed absorb all their edges
        S = torch.zeros(nu, dtype=torch.bool); S[u[pull]] = True
        res.append(int(S.sum()) + int(T.sum()))
    return res

for P in (2, 4, 8):
    parts = [cbdist.Partition.balanced(in_deg, P, q) for q in range(P)]
    own_s = parts[0].owner(src); own_d = parts[0].owner(dst)
    rows = []
    pairs_to_do = [(1, 0)] if P == 2 else [(1, 0), (P - 1, 0), (0, P - 1)]
    for (q, r) in pairs_to_do:
        m = (own_s == q) & (own_d == r)
        us, vs = src[m], dst[m]
        uu, ui = torch.unique(us, return_inverse=True); vv, vi = torch.unique(vs, return_inverse=True)
        # distinct edges only (multigraph duplicates irrelevant for cover)
        t1 = time.time()
        A = sp.coo_matrix((np.ones(ui.numel(), dtype=np.int8), (ui.numpy(), vi.numpy())), shape=(uu.numel(), vv.numel())).tocsr()
        match = maximum_bipartite_matching(A, perm_type='column')
        mvc = int((match >= 0).sum())
        t2 = time.time()
        h = heuristic(ui, vi)
        print(f'P={P} pair {q}->{r}: edges {int(m.sum())} uniq_src(pull) {uu.numel()} uniq_dst(push) {vv.numel()} exact_cover {mvc} ({mvc/uu.numel():.3f} of pull) '
              f'heuristic {h} ({min(h)/uu.numel():.3f}) [match {t2-t1:.1f}s]', flush=True)
        # per-chunk (K=4) covers: sources split by owner's row chunk
        K = 4
        cb = cbdist.chunk_bounds(parts[q].lo(), parts[q].hi(), K)
        tot_mvc, tot_pull = 0, 0
        for k in range(K):
            mk = (us >= cb[k]) & (us < cb[k + 1])
            if int(mk.sum()) == 0: continue
            a, ai = torch.unique(us[mk], return_inverse=True); b, bi = torch.unique(vs[mk], return_inverse=True)
            A = sp.coo_matrix((np.ones(ai.numel(), dtype=np.int8), (ai.numpy(), bi.numpy())), shape=(a.numel(), b.numel())).tocsr()
            mt = maximum_bipartite_matching(A, perm_type='column')
            tot_mvc += int((mt >= 0).sum()); tot_pull += a.numel()
        print(f'      per-chunk K=4: pull {tot_pull} cover {tot_mvc} ({tot_mvc/tot_pull:.3f})', flush=True)
