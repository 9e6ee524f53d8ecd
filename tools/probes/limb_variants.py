"""Builds measurement-only variants of libcoldbrew_hip.so (tools/probes/_bin/lib_<name>.so) in which one cost of the
three-limb GEMM is removed (results become wrong on purpose), to see what the kernels are sensitive to.
  nosplit : the limb split is replaced by a plain repack (VALU work of the split removed)
  mfma3   : only three of the six limb products are issued
  nostore : the NN epilogue's global stores are skipped
usage: python tools/probes/limb_variants.py build | python tools/probes/limb_variants.py run <name|base> [M]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'gnn-tail-generalization_amd', 'csrc')
BIN = os.path.join(ROOT, 'tools', 'probes', '_bin')

VARIANTS = {
    'nosplit': [('cb_limb_core.h', '''  hi = cvt_pk_bf16(a0, a1);
  const float r0 = a0 - __uint_as_float(hi << 16), r1 = a1 - __uint_as_float(hi & 0xffff0000u);       // exact
  mid = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);     // exact
  lo = cvt_pk_bf16(s0, s1);                                                                            // exact''', '''  hi = cvt_pk_bf16(a0, a1);
  mid = hi;
  lo = hi;''')],
    'mfma3': [('cb_gemm_limb.hip', '''    CB_MFMA4(a_mid, b_mid, H_, H_ * 6 + 2)                                                                                   \\
    CB_MFMA4(a_mid, b_hi, H_, H_ * 6 + 3)                                                                                    \\
    CB_MFMA4(a_hi, b_mid, H_, H_ * 6 + 4)                                                                                    \\
    CB_MFMA4(a_hi, b_hi, H_, H_ * 6 + 5)                                                                                     \\''', '''    CB_MFMA4(a_mid, b_mid, H_, H_ * 6 + 2)                                                                                   \\
    pr.template group<H_ * 6 + 3, NG>(); pr.template group<H_ * 6 + 4, NG>(); pr.template group<H_ * 6 + 5, NG>(); \\''')],
    'noload': [('cb_gemm_limb.hip', '''      oa.template load<PIECE>(fa, abase, lda, left2, nullptr, t);''', '''      if (left2 == -12345) oa.template load<PIECE>(fa, abase, lda, left2, nullptr, t);'''),
               ('cb_gemm_limb.hip', '''      ob.template load<J>(fb, bbase, ldb, left2, bscale, t);''', '''      if (left2 == -12345) ob.template load<J>(fb, bbase, ldb, left2, bscale, t);''')],
    'l2load': [('cb_gemm_limb.hip', '''      Producer<OA, OB> pr{oa, ob, fa[slot], fb[slot], nxt, nxt + OA::BYTES, a0 + st * astep, b0 + st * bstep,''', '''      Producer<OA, OB> pr{oa, ob, fa[slot], fb[slot], nxt, nxt + OA::BYTES, a0 + (st & 1) * astep, b0 + (st & 1) * bstep,''')],
    'l2blk': [('cb_gemm_limb.hip', '''  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + m0 * lda, KS, lda, B + n0,''', '''  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + (m0 & 0xfffff) * lda, KS, lda, B + n0,'''),
              ('cb_gemm_limb.hip', '''  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + r_begin * lda + i0, (int64_t)KS * lda, lda, G + r_begin * ldg + j0, (int64_t)KS * ldg, ldg,''', '''  limb_k_loop<WTN, PD, OA, OB>(oa, ob, smem, A + (r_begin & 0xfffff) * lda + i0, (int64_t)KS * lda, lda, G + (r_begin & 0xfffff) * ldg + j0, (int64_t)KS * ldg, ldg,''')],
    'rawstage': [('cb_gemm_limb.hip', '''    const bool live = ((rmask >> J) & 1) && (t & 3) * 4 < k_left;
    const float v[4] = {live ? f[J].x : 0.f, live ? f[J].y : 0.f, live ? f[J].z : 0.f, live ? f[J].w : 0.f};
    uint2 pl[3];
    split4(v, pl);''', '''    uint2 pl[3];
    pl[0] = make_uint2(__float_as_uint(f[J].x), __float_as_uint(f[J].y)); pl[1] = make_uint2(__float_as_uint(f[J].z), __float_as_uint(f[J].w)); pl[2] = pl[0];'''),
                 ('cb_gemm_limb.hip', '''    const bool live = cok && t / TPR + KPP * J < k_left;
    const float m = SCALED ? sc[J] : 1.f;
    const float v[4] = {live ? f[J].x * m : 0.f, live ? f[J].y * m : 0.f, live ? f[J].z * m : 0.f, live ? f[J].w * m : 0.f};
    uint2 pl[3];
    split4(v, pl);''', '''    uint2 pl[3];
    pl[0] = make_uint2(__float_as_uint(f[J].x), __float_as_uint(f[J].y)); pl[1] = make_uint2(__float_as_uint(f[J].z), __float_as_uint(f[J].w)); pl[2] = pl[0];''')],
    'nostage': [('cb_gemm_limb.hip', '''      if (do_stage) oa.template stage<PIECE>(fa, dstA, left1, t);''', '''      if (do_stage && left1 == -12345) oa.template stage<PIECE>(fa, dstA, left1, t);'''),
                ('cb_gemm_limb.hip', '''      if (do_stage) ob.template stage<J>(fb, dstB, left1, t);''', '''      if (do_stage && left1 == -12345) ob.template stage<J>(fb, dstB, left1, t);''')],
    'nobarrier': [('cb_gemm_limb.hip', '''      for (int j = 0; j < OB::NV; ++j) if (OB::HAS_SC) scs[slot][j] = ob.sc[j];
      __syncthreads();''', '''      for (int j = 0; j < OB::NV; ++j) if (OB::HAS_SC) scs[slot][j] = ob.sc[j];''')],
    'nofrag': [('cb_gemm_limb.hip', '''    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(S + plane * PLANE + addr));''', '''    uint4 z = make_uint4(addr, plane, addr, plane); return __builtin_bit_cast(bf16x8, z);'''),
               ('cb_gemm_limb.hip', '''    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * ROWB));''', '''    const s16x4 lo = {(short)addr, (short)plane, 1, 2};
    const s16x4 hi = {(short)plane, (short)addr, 3, 4}; (void)q;''')],
    'nostore': [('cb_gemm_core.h', '''      if (m < M && n < N) {
        const float4 v = *reinterpret_cast<const float4*>(Cs + row * LDB + c4);''', '''      if (m < M && n < N && ep.relu == 77) {
        const float4 v = *reinterpret_cast<const float4*>(Cs + row * LDB + c4);''')],
}


def build():
    os.makedirs(BIN, exist_ok=True)
    objs = [os.path.join(CSRC, '..', '_build', f) for f in os.listdir(os.path.join(CSRC, '..', '_build'))
            if f.endswith('.o') and f not in ('cb_gemm_limb.o', 'cb_gemm.o', 'cb_topk.o')]
    for name, edits in VARIANTS.items():
        tmp = os.path.join(BIN, 'src_' + name)
        os.makedirs(tmp, exist_ok=True)
        srcs = {f: open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith(('.h', '.hip'))}
        for _, old, new in edits:          # the pattern may live in the kernel file or in the shared core header
            hits = [f for f, txt in srcs.items() if old in txt]
            assert len(hits) == 1, (name, old[:60], hits)
            srcs[hits[0]] = srcs[hits[0]].replace(old, new)
        for f, txt in srcs.items():
            open(os.path.join(tmp, f), 'w').write(txt)
        outs = []
        for f in ('cb_gemm_limb.hip', 'cb_gemm.hip', 'cb_topk.hip'):
            o = os.path.join(tmp, f[:-4] + '.o')
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
                                   '-I' + os.path.join(ROOT, 'gnn-tail-generalization_amd', 'csrc'), '-c', os.path.join(tmp, f), '-o', o])
            outs.append(o)
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(BIN, f'lib_{name}.so')] + objs + outs)
        print('built', name)


def run(name, M):
    sys.path.insert(0, ROOT)
    import torch
    from gnn_tail_generalization_amd import _lib, gemm
    if name != 'base':
        _lib.LIB_PATH = os.path.join(BIN, f'lib_{name}.so')
    dev = 'cuda:0'
    a = torch.rand(M, 256, device=dev) - 0.5
    b = torch.rand(256, 256, device=dev) - 0.5
    g = torch.rand(M, 256, device=dev) - 0.5
    rs = torch.rand(M, device=dev)

    def timeit(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for i in range(iters):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return min(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    print(f'{name:8s} NN {timeit(lambda: gemm.mm_nn(a, b, rowscale=rs)):7.3f} ms   TN {timeit(lambda: gemm.mm_tn(a, g, rowscale=rs)):7.3f} ms', flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    else:
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000)
