"""CPU: the stand-alone GPU programs under scripts/ (hazard reproducer, C++ ABI consumer) still compile for gfx950 against
the current sources, header and library -- they are launched through scripts/prepared_gpu_runs.sh, where a build error would cost a GPU session."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
COMMON = ['--offload-arch=gfx950', '-O1', '-std=c++17', '-ffp-contract=off', '-I', os.path.join(ROOT, 'include'),
          '-I', os.path.join(ROOT, 'geotransformer_amd', 'csrc'),
          '-I', os.path.join(ROOT, 'scripts')]
LINK = ['-L', os.path.join(ROOT, 'geotransformer_amd'), '-lgeotr_hip', '-Wl,-rpath,' + os.path.join(ROOT, 'geotransformer_amd')]
PROGRAMS = {'packed_fp32_mfma_hazard.hip': [], 'abi_bench.cpp': LINK}


def test_prepared_gpu_programs_compile(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')

    def build(item):
        name, extra = item
        exe = str(tmp_path / name.replace('.hip', '.bin'))
        res = subprocess.run([HIPCC] + COMMON + [os.path.join(ROOT, 'scripts', name)] + extra + ['-o', exe], capture_output=True, text=True)
        return name, res.returncode, res.stderr[-1500:], exe

    with ThreadPoolExecutor(max_workers=3) as pool:
        results = list(pool.map(build, PROGRAMS.items()))
    for name, rc, err, exe in results:
        assert rc == 0, (name, err)
        assert os.path.getsize(exe) > 10000, name
    # every program the launcher script names exists
    text = open(os.path.join(ROOT, 'scripts', 'prepared_gpu_runs.sh')).read()
    for name in list(PROGRAMS):
        assert 'scripts/' + name in text, name
