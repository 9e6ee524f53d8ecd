"""CPU: the parts of bench.py's contract that need no GPU — the self-launch of `python bench.py --gpus N` and the
per-family roofline arithmetic (item 1a: frac on SURVEY.md 8(d) bytes only)."""
import argparse
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _a(gpus):
    return argparse.Namespace(gpus=gpus)


def test_one_gpu_or_a_launcher_means_no_relaunch():
    assert bench.self_launch_argv(_a(1), ['--gpus', '1'], {}, 8) is None
    # a driver that starts the ranks itself (torch.distributed.run sets WORLD_SIZE) is left alone
    assert bench.self_launch_argv(_a(8), ['--gpus', '8'], {'WORLD_SIZE': '8', 'RANK': '3'}, 8) is None


def test_gpus_n_without_a_launcher_relaunches_itself_as_n_ranks():
    argv = bench.self_launch_argv(_a(4), ['--gpus', '4', '--steps', '3'], {}, 8)
    assert argv[0] == sys.executable and argv[1:3] == ['-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in argv and '--nnodes=1' in argv
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1' and int(argv[argv.index('--master-port') + 1]) > 0
    i = argv.index(os.path.abspath(bench.__file__))
    assert argv[i + 1:] == ['--gpus', '4', '--steps', '3']


def test_fewer_devices_than_ranks_fails_loudly_except_in_the_gloo_dry_run():
    with pytest.raises(SystemExit, match='only 1 device'):
        bench.self_launch_argv(_a(8), ['--gpus', '8'], {}, 1)
    argv = bench.self_launch_argv(_a(2), ['--gpus', '2'], {'COLDBREW_DIST_BACKEND': 'gloo'}, 1)
    assert '--nproc-per-node=2' in argv


def test_roofline_families_use_survey_8d_bytes_only():
    recs = ([{'kind': 'agg_gemm_fused', 'edges': 100, 'rows': 10, 'agg': 8e9, 'store': 1e9, 'tail': 1e9, 'ms': 2.0}] * 4
            + [{'kind': 'plain', 'edges': 10, 'rows': 10, 'agg': 1e9, 'store': 0, 'tail': 0, 'ms': 0.25}] * 2)
    fams = bench.roofline_families(recs, steps=2)
    assert [f['kind'] for f in fams] == ['agg_gemm_fused', 'plain']            # by time per step
    f = fams[0]
    assert f['launches_timed'] == 4 and f['launches_per_step'] == 2 and f['total_ms_per_step'] == pytest.approx(4.0)
    assert f['achieved'] == pytest.approx(8e9 / 2e-3 / 1e9) and f['frac'] == pytest.approx(4000 / 8000)        # 8(d) bytes only
    assert f['frac_incl_fused_streams'] == pytest.approx(10e9 / 2e-3 / 1e9 / 8000)
    assert fams[1]['frac'] == pytest.approx(1e9 / 0.25e-3 / 1e9 / 8000) and fams[1]['frac_incl_fused_streams'] == fams[1]['frac']
