"""CPU (hipcc cross-compiles without a GPU): the compiler must not put a vmcnt wait in front of the transposing LDS reads of the DMA-ring kernels.

Round 5 found `s_waitcnt vmcnt(0)` inserted by hipcc in front of every tile's first ds_read_b64_tr_b16 (issued through the intrinsic) in the
attention forward / backward kernels and the TN weight-gradient GEMM: the rings were drained once per step (profiles/r5_attention_ab.txt §1).  The
reads are inline asm now (csrc/vf_common.h: vf_tr_frag2_wait); this test keeps it that way: it compiles the three sources to ISA and runs
tools/isa_vmcnt_audit.py over them — a compiler-inserted wait that names vmcnt directly in front of an LDS read inside a loop fails the test —
and checks that the audit still SEES the old form (-DVF_X_TRINTRIN)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, 'tools'))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CSRC = os.path.join(REPO, 'viewformer_amd', 'csrc')


def _isa(stem, tmp_path, extra=()):
    out = str(tmp_path / (stem + ('.x' if extra else '') + '.s'))
    subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-w', *extra, os.path.join(CSRC, stem + '.hip'), '-o', out],
                   check=True, capture_output=True, timeout=600)
    return out


def _lds_read_waits(res):
    return {k: [w for w in v if w[3].startswith('ds_read')] for k, v in res.items() if any(w[3].startswith('ds_read') for w in v)}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_no_compiler_vmcnt_wait_in_front_of_lds_reads_in_the_ring_kernels(tmp_path):
    from isa_vmcnt_audit import audit
    procs = {}
    for stem in ('attention_dma', 'attention_train_bf16', 'gemm_tn_bf16'):
        procs[stem] = _isa(stem, tmp_path)
    for stem, path in procs.items():
        bad = _lds_read_waits(audit(path))
        assert not bad, (stem, {k[:80]: v[:2] for k, v in bad.items()})
    # the detector itself: the intrinsic form of the same reads still shows the drain
    old = _lds_read_waits(audit(_isa('gemm_tn_bf16', tmp_path, ('-DVF_X_TRINTRIN',))))
    assert old and all(any('vmcnt(0)' in w[2] for w in v) for v in old.values()), old
