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
