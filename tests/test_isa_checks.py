"""CPU: static proof on the gfx950 ISA that the split issue / wait LDS pipelines are hazard-free.

The GSE kernel and the packed GEMM issue `ds_read_b128` from inline asm and wait for the data one step later; in between the
compiler does not know the destination registers are still being filled.  scripts/check_inflight_regs.py propagates the "may be
in flight" register set over each kernel's control-flow graph and fails if any instruction touches such a register -- a violation
would be silent data corruption that a parity test can miss when the data happens to land in time.  hipcc cross-compiles here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'geotransformer_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _flags():
    """The Makefile's compile flags (so the checked ISA is the shipped ISA), minus -fPIC / -c."""
    for line in open(os.path.join(CSRC, 'Makefile')):
        if line.startswith('FLAGS'):
            flags = line.split(':=', 1)[1].split()
            return [f.replace('$(ARCH)', 'gfx950') for f in flags if f != '-fPIC']
    raise AssertionError('FLAGS not found in csrc/Makefile')


@pytest.mark.parametrize('source,prefix,min_kernels', [
    ('transformer.hip', '_ZN5geotr23gse_embed_bf16x3_kernel', 32),  # D in {32,64,128,256} x S in {2..5} x TERMS in {3,1}
    ('gemm.hip', '_ZN5geotr18gemm_packed_kernel', 6),               # three tilings x TERMS in {3,1}
])
def test_no_instruction_touches_an_in_flight_lds_fragment(tmp_path, source, prefix, min_kernels):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    asm = str(tmp_path / (source + '.s'))
    cmd = [HIPCC] + _flags() + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', os.path.join(CSRC, source), '-o', asm]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    assert res.returncode == 0, res.stderr[-2000:]
    chk = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'check_inflight_regs.py'), asm, prefix],
                         capture_output=True, text=True)
    lines = [ln for ln in chk.stdout.splitlines() if ln.startswith(prefix)]
    assert chk.returncode == 0, chk.stdout[-3000:]
    assert len(lines) >= min_kernels, chk.stdout[-2000:]
    # the check must have seen the asm reads at all (guards against the markers / mnemonics changing under it)
    assert all(int(ln.split(': ')[1].split()[0]) > 0 for ln in lines), chk.stdout[-2000:]


def test_matching_heads_are_compiled_without_packed_fp32_shuffles(tmp_path):
    """profiles/r03_concurrency_hazard.md: the one build of p2n_assign that misassigned points under multi-stream load was the
    SLP-vectorised one (ds_read2_b32 results consumed through v_pk_mov_b32 / op_sel'd v_pk_mul_f32).  matching.hip is therefore
    compiled with -fno-slp-vectorize; this pins the shipped ISA of the kernels that take discrete decisions: no v_pk_mov_b32 anywhere
    in the file, no packed fp32 arithmetic at all in the three consumers of the pyramid's point arrays."""
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    mk = open(os.path.join(CSRC, 'Makefile')).read()
    assert 'FLAGS_matching := -fno-slp-vectorize' in mk and '$(FLAGS_$*)' in mk
    asm = str(tmp_path / 'matching.s')
    cmd = [HIPCC] + _flags() + ['-fno-slp-vectorize', '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', os.path.join(CSRC, 'matching.hip'),
                                '-o', asm]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    assert res.returncode == 0, res.stderr[-2000:]
    name, packed = None, {}
    for line in open(asm):
        if line.startswith('_Z') and line.rstrip().endswith(':'):
            name = line.strip()[:-1]
        elif name and 'v_pk_' in line and not line.lstrip().startswith(';'):
            packed.setdefault(name, []).append(line.split()[0])
    assert not any(op == 'v_pk_mov_b32' for ops in packed.values() for op in ops), {k: v for k, v in packed.items() if 'v_pk_mov_b32' in v}
    for kernel in ('p2n_assign_kernel', 'p2n_knn_kernel', 'patch_gather_kernel'):
        hits = {k: v for k, v in packed.items() if kernel in k and any(op.startswith(('v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32')) for op in v)}
        assert not hits, hits
