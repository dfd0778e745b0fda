"""CPU: static proof on the gfx950 ISA that the split issue / wait LDS pipelines are hazard-free.

The GSE kernel and the packed GEMM issue `ds_read_b128` from inline asm and wait for the data one step later; in between the
compiler does not know the destination registers are still being filled.  scripts/check_inflight_regs.py propagates the "may be
in flight" register set over each kernel's control-flow graph and fails if any instruction touches such a register -- a violation
would be silent data corruption that a parity test can miss when the data happens to land in time.  hipcc cross-compiles here."""
import os
import re
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
    ('gemm.hip', '_ZN5geotr18gemm_packed_kernel', 12),              # three tilings x TERMS in {3,1,0} on the two-slot ring + the 64-wide tile on the three-slot ring
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


# kernels whose packed fp32 code is WRITTEN as such (float2 / ext-vector arithmetic in the source, formed by the backend, not by the SLP
# vectoriser); everything else must come out of the compiler as scalar fp32 VALU code
EXPLICIT_PACKED = ('kpconv_gather_kernel', 'l2_normalize_kernel', 'lgr_score_kernel', 'nc_overlap_kernel', 'patch_sinkhorn_kernel',
                   'attn_softmax_kernel', 'attn_softmax_grouped_kernel', 'attn_pos_softmax_kernel', 'attn_pos_softmax_grouped_kernel',
                   'attn_softmax_grouped_pos_kernel', 'attn_softmax_extras_kernel', 'attn_pos_softmax_extras_kernel',  # (round 5: the same bodies)
                   'gse_embed_table_kernelILi128E')
LANE_HALF_SHUFFLES = ('kpconv_gather_kernel',)  # op_sel'd broadcasts of one neighbour weight over a channel pair, written by hand


def _packed_fp32_by_kernel(asm):
    name, packed = None, {}
    for line in open(asm):
        label = re.match(r'^(_Z\w+):', line)  # "<mangled name>:   ; @<mangled name>"
        if label:
            name = label.group(1)
        elif name and 'v_pk_' in line and not line.lstrip().startswith(';'):
            packed.setdefault(name, []).append(line.strip())
    return packed


_PACKED = re.compile(r'^(v_pk_(?:add|mul|fma)_f32)\s+v\[(\d+):(\d+)\],\s*(.*)$')


def _cross_half_reads_of_the_destination(op):
    """The ONE instruction form that was caught returning a wrong low half (profiles/r03_concurrency_hazard.md 4e):
    `v_pk_add_f32 v[12:13], v[26:27], v[12:13] op_sel:[0,1] ...` -- a packed fp32 instruction whose destination pair is also a source pair
    that is read ACROSS its halves (op_sel = 1: the low lane takes the high register; op_sel_hi = 0: the high lane takes the low one).
    Returns the indices of such sources."""
    m = _PACKED.match(op)
    if not m:
        return []
    dst = (int(m.group(2)), int(m.group(3)))
    rest = m.group(4)
    sel = re.search(r'op_sel:\[([\d,]+)\]', rest)
    sel_hi = re.search(r'op_sel_hi:\[([\d,]+)\]', rest)
    sel = [int(x) for x in sel.group(1).split(',')] if sel else [0, 0, 0]
    sel_hi = [int(x) for x in sel_hi.group(1).split(',')] if sel_hi else [1, 1, 1]
    hits = []
    for k, tok in enumerate(rest.split(',')[:3]):
        src = re.match(r'\s*v\[(\d+):(\d+)\]', tok)
        if not src:
            continue
        lo, hi = int(src.group(1)), int(src.group(2))
        overlaps = not (hi < dst[0] or lo > dst[1])
        crosses = (k < len(sel) and sel[k] == 1) or (k < len(sel_hi) and sel_hi[k] == 0)
        if overlaps and crosses:
            hits.append(k)
    return hits


def _low_lane_reads_the_high_register_of_source_1_or_2(op):
    """Round 4, the instruction-form matrix (profiles/r04_hazard_form_matrix.md; scripts/packed_fp32_mfma_hazard.hip VICTIM=instruction):
    next to kernels issuing double-rate bf16 MFMAs between loads, a packed fp32 instruction returns a wrong LOW half in lanes 48-63
    exactly when `op_sel` is set for source 1 or source 2 -- the low lane takes the HIGH register of that source pair -- whether or not
    the destination is one of the sources (forms 0, 1, 4, 6: 161-169 of 200 launches wrong).  `op_sel` on source 0 (form 5), any
    `op_sel_hi` pattern (form 2: what kpconv_gather's broadcasts use) and straight reads (form 3) never failed.  Returns the indices of
    the offending sources."""
    m = _PACKED.match(op)
    if not m:
        return []
    sel = re.search(r'op_sel:\[([\d,]+)\]', m.group(4))
    sel = [int(x) for x in sel.group(1).split(',')] if sel else []
    return [k for k in (1, 2) if k < len(sel) and sel[k] == 1]


def test_unsafe_form_detector_follows_the_measured_form_matrix():
    bad = _low_lane_reads_the_high_register_of_source_1_or_2
    assert bad('v_pk_add_f32 v[12:13], v[26:27], v[12:13] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]') == [1]       # form 0: failed
    assert bad('v_pk_add_f32 v[30:31], v[26:27], v[12:13] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]') == [1]       # form 1 (not in place): failed
    assert bad('v_pk_mul_f32 v[12:13], v[26:27], v[12:13] op_sel:[0,1]') == [1]                                  # form 4: failed
    assert bad('v_pk_fma_f32 v[12:13], v[26:27], v[26:27], v[12:13] op_sel:[0,0,1]') == [2]                      # form 6: failed
    assert bad('v_pk_add_f32 v[10:11], v[24:25], v[10:11] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]') == []     # form 2: never
    assert bad('v_pk_add_f32 v[10:11], v[24:25], v[10:11] neg_lo:[0,1] neg_hi:[0,1]') == []                      # form 3: never
    assert bad('v_pk_add_f32 v[10:11], v[10:11], v[24:25] op_sel:[1,0]') == []                                   # form 5 (source 0): never
    assert bad('v_pk_fma_f32 v[4:5], v[8:9], v[2:3], v[4:5] op_sel_hi:[1,0,1]') == []                            # kpconv_gather's broadcast


def test_cross_half_read_detector_knows_the_instruction_that_failed():
    assert _cross_half_reads_of_the_destination('v_pk_add_f32 v[12:13], v[26:27], v[12:13] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]') == [1]
    assert _cross_half_reads_of_the_destination('v_pk_fma_f32 v[14:15], v[14:15], v[38:39], v[24:25] op_sel:[1,0,0]') == [0]
    assert _cross_half_reads_of_the_destination('v_pk_add_f32 v[10:11], v[24:25], v[10:11] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]') == [1]
    assert _cross_half_reads_of_the_destination('v_pk_mul_f32 v[8:9], v[6:7], v[8:9]') == []                       # in place, straight
    assert _cross_half_reads_of_the_destination('v_pk_mul_f32 v[22:23], v[2:3], v[6:7] op_sel:[1,0]') == []        # crossed, not in place
    assert _cross_half_reads_of_the_destination('v_pk_fma_f32 v[4:5], v[8:9], s[2:3], v[4:5] op_sel_hi:[1,0,1]') == []


def test_no_kernel_carries_slp_vectorised_packed_fp32_code(tmp_path):
    """profiles/r03_concurrency_hazard.md: the two builds that returned wrong values under multi-stream load (p2n_assign's distance loop,
    the index computation of gse_embed_table) were the ones the SLP vectoriser had turned into packed fp32 sequences with lane-half
    shuffles (v_pk_mov_b32 ... op_sel, op_sel'd v_pk_mul_f32 / v_pk_add_f32); the same sources without the vectoriser never failed.
    Every file is therefore compiled with -fno-slp-vectorize, and this pins the shipped ISA of ALL of them: no v_pk_mov_b32 anywhere,
    packed fp32 arithmetic only in the kernels that spell it out in the source, op_sel'd packed arithmetic only in kpconv_gather, and
    NOWHERE the instruction form whose low half was caught wrong (destination pair = a source pair read across its halves)."""
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    mk = open(os.path.join(CSRC, 'Makefile')).read()
    assert '-fno-slp-vectorize' in _flags() and '$(FLAGS)' in mk
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    assert len(sources) >= 11

    def compile_one(src):
        asm = str(tmp_path / (src[:-4] + '.s'))
        cmd = [HIPCC] + _flags() + ['-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', asm]
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        assert res.returncode == 0, res.stderr[-2000:]
        return src, _packed_fp32_by_kernel(asm)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        per_file = dict(pool.map(compile_one, sources))
    seen_kernels = 0
    for src, packed in per_file.items():
        for kernel, ops in packed.items():
            seen_kernels += 1
            assert not any(op.startswith('v_pk_mov_b32') for op in ops), (src, kernel)
            arith = [op for op in ops if op.startswith(('v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32'))]
            if arith:
                assert any(tag in kernel for tag in EXPLICIT_PACKED), (src, kernel, arith[:4])
            crossed = [op for op in arith if _cross_half_reads_of_the_destination(op)]
            assert not crossed, (src, kernel, crossed[:4])  # nowhere, the hand-written kernels included
            # round 4: every form the instruction-level matrix caught, in place or not -- nowhere, EXPLICIT_PACKED kernels included
            unsafe = [op for op in arith if _low_lane_reads_the_high_register_of_source_1_or_2(op)]
            assert not unsafe, (src, kernel, unsafe[:4])
            if any('op_sel' in op for op in arith):
                assert any(tag in kernel for tag in LANE_HALF_SHUFFLES), (src, kernel, [op for op in arith if 'op_sel' in op][:4])
    assert seen_kernels >= 10  # the explicit kernels were found at all (guards against the mnemonics changing under the check)
    # the kernels that take discrete decisions on the pyramid's point arrays: no packed fp32 arithmetic at all
    for kernel in ('p2n_assign_kernel', 'p2n_knn_kernel', 'patch_gather_kernel'):
        hits = {k: v for k, v in per_file['matching.hip'].items() if kernel in k and
                any(op.startswith(('v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32')) for op in v)}
        assert not hits, hits


def test_the_in_flight_checker_sees_loop_back_edges_and_counted_vmcnt_waits(tmp_path):
    """The checker itself, on a hand-written kernel text: (1) a label followed by a comment is a block (round 4: such labels -- every loop
    header the compiler emits -- were dropped, so a register still in flight at the back edge went unseen); (2) an asm global load is in
    flight until a vmcnt wait that leaves FEWER younger operations outstanding than were issued after it."""
    checker = os.path.join(ROOT, 'scripts', 'check_inflight_regs.py')

    def run(body):
        asm = tmp_path / 'k.s'
        asm.write_text('_ZN5geotr4testEv:\n' + body + '\ts_endpgm\n')
        return subprocess.run([sys.executable, checker, str(asm), '_ZN5geotr4test'], capture_output=True, text=True)

    loop = ('\tv_mov_b32_e32 v4, 0\n'
            '.LBB0_1:                                ; %loop header with a comment\n'
            '\tv_add_f32_e32 v8, v4, v4\n'           # reads v4: clean on entry, in flight on the back edge (issued below, never waited for)
            '\t;;#ASMSTART\n\tds_read_b128 v[4:7], v1\n\t;;#ASMEND\n'
            '\ts_cbranch_scc1 .LBB0_1\n'
            '\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n')
    res = run(loop)
    assert res.returncode == 1 and 'v_add_f32_e32 v8, v4, v4' in res.stdout, res.stdout
    counted = ('\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v[2:3], off\n\t;;#ASMEND\n'
               '\tglobal_load_lds_dwordx4 v[10:11], off\n\tglobal_load_lds_dwordx4 v[12:13], off\n'
               '\t;;#ASMSTART\n\ts_waitcnt vmcnt({n})\n\t;;#ASMEND\n'
               '\tv_mfma_f32_32x32x2_f32 v[16:31], v4, v5, v[16:31]\n')
    assert run(counted.format(n=2)).returncode == 0          # the two younger DMA operations may stay outstanding
    bad = run(counted.format(n=3))
    assert bad.returncode == 1 and 'v_mfma' in bad.stdout, bad.stdout
