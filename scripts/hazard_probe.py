"""Round 3: what IS the multi-stream stale read of profiles/r02_concurrency_hazard.md?  (writes the evidence for profiles/r03_concurrency_hazard.md)

One process = one experiment (the switches are environment variables read at start-up):
  NEEDS the library built with the investigation tooling: make -C geotransformer_amd/csrc clean all FLAGS_matching=-DGEOTR_HAZARD_TOOLS
  (round 4: the shipped library carries the plain-load kernels only; without the flag the switches below are ignored and
  geotr_debug_probe_read is absent)
  GEOTR_P2N_MODE   0 agent-scope loads in p2n_assign (shipped) | 1 plain | 2 plain behind an agent acquire fence at kernel entry | 3 nt
  GEOTR_P2N_PROBE  1: a probe kernel in front of p2n_assign reads every word of the two point arrays plain / agent / plain and records
                   the words whose first read differs from the agent-scope read (csrc/matching.hip, geotr_debug_probe_read)
  GEOTR_ALLOC_LOG  1: every buffer handed to the two native calls is logged (thread, name, address range, time) -> the owners of a
                   stale word's address over time
  GEOTR_POISON_WS  1: workspaces / outputs filled with 0xFF before use
  anything else the runtime reads (GPU_MAX_HW_QUEUES, AMD_SERIALIZE_KERNEL, PYTORCH_NO_HIP_MEMORY_CACHING, ...)
Workload: `LANES` lanes (default 4), `LANES` rotated stacks of 8 x (20k + 20k) per submission, `REPS` submissions (default 24); every
output of every stack slot -- pyramid tables included -- is hashed per submission and compared with the majority over submissions.

usage: LABEL=name [env switches] python scripts/hazard_probe.py            (one JSON line on stdout, details on stderr)
"""
import ctypes
import hashlib
import json
import os
import sys
import time
from collections import Counter, defaultdict

import numpy as np
import torch

ROOT = os.environ.get('GEOTR_TREE') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # GEOTR_TREE: another checkout
sys.path.insert(0, ROOT)


def note(*a):
    print(*a, file=sys.stderr, flush=True)


class ProbeRecord(ctypes.Structure):
    _fields_ = [('addr', ctypes.c_uint64), ('clock', ctypes.c_uint64), ('plain', ctypes.c_uint32), ('agent', ctypes.c_uint32),
                ('plain_again', ctypes.c_uint32), ('kind_cloud', ctypes.c_uint32), ('elem', ctypes.c_uint32), ('hw_id', ctypes.c_uint32),
                ('xcc_id', ctypes.c_uint32), ('block', ctypes.c_uint32)]


_weights = {}


def digest(t):
    """Position-weighted 64-bit checksum computed on the device (the tables of one stack are hundreds of MB)."""
    t = t.detach().contiguous()
    flat = (t.view(torch.uint8) if t.element_size() == 1 else t.view(torch.int32)).reshape(-1).to(torch.int64)
    n = flat.numel()
    if n not in _weights:
        _weights[n] = torch.arange(n, device=t.device, dtype=torch.int64) % 1000003 + 1
    return int((flat * _weights[n]).sum().item())


def main():
    from geotransformer_amd import _lib, kernels, native
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import make_pair
    label = os.environ.get('LABEL', 'run')
    lanes, reps, stack = int(os.environ.get('LANES', '4')), int(os.environ.get('REPS', '24')), int(os.environ.get('STACK', '8'))
    if os.environ.get('GSE', 'table') == 'table':
        kernels.set_precision('bf16x3')
    else:
        kernels.set_precision('bf16x3', gse=os.environ['GSE'])  # GSE=mfma: the embedding on the round-1 MFMA kernel
    cfg = make_cfg('3dmatch')
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(i, '3dmatch', n_points=20000) for i in range(8)]
    pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
    try:
        runner = ConcurrentRegistration(pipe, lanes=lanes, stack=stack, return_pyramid=True)
        have_pyramid = True
    except TypeError:  # a checkout from before round 3
        runner = ConcurrentRegistration(pipe, lanes=lanes, stack=stack)
        have_pyramid = False
    n = stack * lanes
    batch = [pairs[(j + j // stack) % 8] for j in range(n)]  # rotated stacks: every lane holds a DIFFERENT stack at any time
    head_keys = ('ref_node_corr_knn_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks', 'src_node_corr_knn_masks',
                 'matching_scores', 'estimated_transform')
    feat_keys = ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f', 'ref_node_corr_indices', 'src_node_corr_indices')
    hashes = defaultdict(dict)   # (slot, key) -> {rep: digest}
    values = {}                  # float bits -> set of (slot, side, level, element) over rep 0 (for tracing stale values)
    lib = _lib.load()
    lib.geotr_debug_probe_read.restype = ctypes.c_int64
    lib.geotr_debug_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_uint32)]
    probe_on = os.environ.get('GEOTR_P2N_PROBE') == '1'
    records, stale_total, words_total = [], 0, 0
    t0 = time.perf_counter()
    depth = int(os.environ.get('DEPTH', '4'))  # submissions queued back to back before anything is drained: the lanes drift apart, so
    # that a stack's heads run while OTHER stacks are in their pyramid / backbone phases (a drained pipeline starts all lanes in step)
    fresh = os.environ.get('FRESH_PIPELINE') == '1'  # a new model + new lanes (threads, streams) for every round, as round 2's bisect tool did
    for rnd in range(0, reps, depth):
      if fresh and rnd:
        runner.close()
        torch.manual_seed(cfg.seed)
        np.random.seed(cfg.seed)
        pipe = RegistrationPipeline(cfg, device='cuda:0')
        runner = (ConcurrentRegistration(pipe, lanes=lanes, stack=stack, return_pyramid=True) if have_pyramid else
                  ConcurrentRegistration(pipe, lanes=lanes, stack=stack))
      allgot = {}
      for rep in range(rnd, min(reps, rnd + depth)):
        runner.submit(batch, lambda j, out, rep=rep: allgot.__setitem__((rep, j), out))
      runner.drain()
      torch.cuda.synchronize()
      for rep in range(rnd, min(reps, rnd + depth)):
        got = {j: allgot[(rep, j)] for j in range(n)}
        for j in range(n):
            o = got[j]
            for k in head_keys + feat_keys:
                hashes[(j, k)][rep] = digest(o[k])
            if j % stack == 0 and have_pyramid:  # the stack's pyramid, every table
                pyr = o['_stack_pyramid']
                for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
                    for i, t in enumerate(pyr[key]):
                        hashes[(j, f'pyramid.{key}{i}')][rep] = digest(t)
        if probe_on and rep == min(reps, rnd + depth) - 1:
            buf = (ProbeRecord * 65536)()
            cnt = (ctypes.c_uint32 * 2)()
            k = lib.geotr_debug_probe_read(buf, 65536, cnt)
            stale_total += int(cnt[0])
            words_total += int(cnt[1])
            for r in buf[:max(k, 0)]:
                records.append((rep, r.addr, r.clock, r.plain, r.agent, r.plain_again, r.kind_cloud, r.elem, r.hw_id, r.xcc_id, r.block))
            if not values:
                for j in range(n):  # every float of the two point arrays of every pair -> (slot, which array, element)
                    for level, key in ((1, 'ref_points_f'), (1, 'src_points_f'), (3, 'ref_points_c'), (3, 'src_points_c')):
                        bits = got[j][key].cpu().numpy().view(np.uint32).ravel()
                        for e, b in enumerate(bits.tolist()):
                            if len(values.setdefault(b, [])) < 4:
                                values[b].append((j, key, e))
        del got
      del allgot
    dt = time.perf_counter() - t0
    runner.close()

    # ---- determinism: digest of every (slot, key) per submission vs the majority over submissions ----
    bad_reps, bad_keys, bad_list = set(), Counter(), []
    for (j, k), per_rep in hashes.items():
        major, _ = Counter(per_rep.values()).most_common(1)[0]
        for rep, h in per_rep.items():
            if h != major:
                bad_reps.add(rep)
                bad_keys[k] += 1
                bad_list.append((rep, j, k))
    res = {'label': label, 'lanes': lanes, 'stack': stack, 'submissions': reps, 'submissions_with_any_difference': len(bad_reps),
           'differing_outputs_by_key': dict(bad_keys),
           'first_differences_rep_slot_key': sorted(bad_list)[:40], 'seconds': round(dt, 1),
           'env': {k: os.environ[k] for k in sorted(os.environ) if k.startswith(('GEOTR_', 'GPU_', 'AMD_', 'HSA_', 'PYTORCH_', 'HIP_', 'ROC'))}}

    # ---- the probe's stale words ----
    if probe_on:
        res['probe'] = {'words_compared': words_total, 'stale_words': stale_total, 'records': len(records)}
        if records:
            names = {0: 'assign pattern: superpoints', 1: 'assign pattern: points (dword)', 2: 'assign pattern: points (12-byte load)',
                     0x11: 'sweep before p2n_assign: points', 0x12: 'sweep before p2n_assign: superpoints', 0x13: 'sweep before p2n_knn: points',
                     0x14: 'sweep before p2n_knn: superpoints', 0x15: 'sweep before patch_gather: points',
                     0x16: 'sweep before patch_gather: per-superpoint index table (workspace)'}
            kinds = Counter(names.get(r[6] >> 16, hex(r[6] >> 16)) for r in records)
            lines = Counter((r[0], r[1] // 128) for r in records)
            per_line = Counter(lines.values())
            healed = sum(1 for r in records if r[5] == r[4])   # the second plain read returned the fresh value
            still = sum(1 for r in records if r[5] == r[3])
            res['probe'].update({
                'by_array': dict(kinds), 'distinct_128B_lines': len(lines), 'stale_words_per_line_histogram': dict(sorted(per_line.items())),
                'second_plain_read_fresh': healed, 'second_plain_read_still_stale': still,
                'by_xcc': dict(sorted(Counter(r[9] for r in records).items())),
                'distinct_compute_units': len({(r[9], r[8] & 0xffff00) for r in records}),
                'stale_value_is_poison_0xffffffff': sum(1 for r in records if r[3] == 0xFFFFFFFF),
                'stale_value_is_zero': sum(1 for r in records if r[3] == 0),
            })
            # where do the stale VALUES come from?  (a) the same element of another stack's array, (b) another element, (c) unknown
            src = Counter()
            for r in records[:4096]:
                hits = values.get(r[3], [])
                fresh = values.get(r[4], [])
                if not hits:
                    src['value not in any point array of this workload'] += 1
                else:
                    same_elem = [h for h in hits if fresh and h[1] == fresh[0][1] and h[2] == fresh[0][2] and h[0] != fresh[0][0]]
                    src['same element of ANOTHER pair\'s array' if same_elem else 'some other element of a point array'] += 1
            res['probe']['stale_value_origin'] = dict(src)
            # owners of the stale addresses over time (allocation log)
            log = getattr(native, 'ALLOC_LOG', None) or []
            if log:
                owners = Counter()
                for r in records[:512]:
                    hist = [(t, who, what) for who, what, ptr, nbytes, t in log if ptr <= r[1] < ptr + nbytes]
                    hist.sort()
                    owners[' -> '.join(f'{what}@{who[-1]}' for _, who, what in hist[-4:])] += 1
                res['probe']['address_owner_histories_last4'] = dict(owners.most_common(12))
            for r in records[:24]:
                note('  stale: rep %d addr %#x %s elem %d plain %#010x agent %#010x plain2 %#010x xcc %d hw %#x block %d' %
                     (r[0], r[1], hex(r[6] >> 16), r[7], r[3], r[4], r[5], r[9], r[8], r[10]))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
