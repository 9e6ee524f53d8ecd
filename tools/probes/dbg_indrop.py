import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnn_tail_generalization_amd import gemm, ops
DEV='cuda:0'
gen = torch.Generator(device=DEV).manual_seed(3)
M,K,N,row0 = 40000,128,256,0
x = torch.rand(M, K, device=DEV, generator=gen)
w = torch.randn(K, N, device=DEV, generator=gen) * 0.1
b = torch.randn(N, device=DEV, generator=gen)
p, s_in, s_out = 0.1, 0x1234ABCD5, 0x77
fused = gemm.mm_nn_indrop_drop2(x, w, p, s_in, s_out, row0, bias=b, relu=True)
xd = ops._dropout_raw(x, p, s_in, row0 * K)
y, yd = gemm.mm_nn_drop2(xd, w, p, s_out, row0, bias=b, relu=True)
y_nodrop = gemm.mm_nn(x, w, bias=b, relu=True)
print('fused vs dropped  max diff', float((fused[0]-y).abs().max()), 'frac rows equal', float((fused[0]==y).all(1).float().mean()))
print('fused vs undropped max diff', float((fused[0]-y_nodrop).abs().max()))
# which k-steps are wrong? use one-hot w
for kk in (0, 5, 16, 17, 100, 127):
    w1 = torch.zeros(K, N, device=DEV); w1[kk, :] = 1.0
    f = gemm.mm_nn_indrop_drop2(x, w1, p, s_in, s_out, row0, bias=None, relu=False)[0][:, 0]
    print(kk, 'match dropped', float((f == xd[:, kk]).float().mean()), 'match undropped', float((f == x[:, kk]).float().mean()))
import numpy as np
scale = np.float32(1.0) / (np.float32(1.0) - np.float32(0.1))
kk = 5
w1 = torch.zeros(K, N, device=DEV); w1[kk, :] = 1.0
f = gemm.mm_nn_indrop_drop2(x, w1, p, s_in, s_out, row0, bias=None, relu=False)[0][:, 0]
keep = xd[:, kk] != 0
t = (x[:, kk] * float(scale))
print('scale', float(scale), 'xd == x*scale (kept):', float((xd[:, kk][keep] == t[keep]).float().mean()), ' f == x*scale (kept):', float((f[keep] == t[keep]).float().mean()))
print('f zero where dropped:', float((f[~keep] == 0).float().mean()), 'kept frac', float(keep.float().mean()))
g_plain = gemm.mm_nn(xd, w1)[:, 0]
print('plain GEMM of xd reproduces xd:', float((g_plain == xd[:, kk]).float().mean()))
d = (f - xd[:, kk])[keep]
print('diff stats', float(d.abs().max()), float((d != 0).float().mean()), 'rel', float((d.abs() / xd[:, kk][keep]).max()))
