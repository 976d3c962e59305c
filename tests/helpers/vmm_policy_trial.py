#!/usr/bin/env python3
"""The question of tools/vmm_reuse_repro.hip asked of THIS library: build libdswx_hip.so with each address-space
policy of proteus_amd/csrc/dswx_batch.hip (DSWX_VM_FREE_ADDRESSES: 0 retire the addresses of a dropped range for good --
the product --, 1 hipMemAddressFree them, 2 device-synchronize + free; round 4 also tried a free list of retired ranges
re-mapped for later requests: 9 of 160 cases wrong, profiles/r04_vmm_policy_trial.json, removed) and run the loop of
tests/test_gpu_parity.py::test_sliding_range_survives_repeated_placement against each: random small batches, allocation
churn, two sliding placements per batch, every layer of every tile against the C oracle.  Prints one JSON object:
per policy the number of (batch, placement) cases with a wrong layer and what the wrong layer looked like.

    python tests/helpers/vmm_policy_trial.py --build          # here (CPU container): compile the three variants in-tree
    python tests/helpers/vmm_policy_trial.py --cases 80       # on the GPU box: run them (one child process per policy)
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from proteus_amd import build as _build            # noqa: E402

TRIAL_DIR = os.path.join(_build.LIB_DIR, 'trial')
POLICIES = {1: 'release + hipMemAddressFree', 2: 'hipDeviceSynchronize + release + hipMemAddressFree', 0: 'retire the addresses, pool the chunks (product)'}


def lib_of(policy):
    return os.path.join(TRIAL_DIR, f'libdswx_hip_vm{policy}.so')


def build_variants():
    os.makedirs(TRIAL_DIR, exist_ok=True)
    hipcc = _build.find_hipcc()
    for pol in POLICIES:
        cmd = [hipcc] + _build.HIPCC_FLAGS + [f'-DDSWX_VM_FREE_ADDRESSES={pol}', '-I', _build.INCLUDE, '-I', _build.CSRC] + \
            _build.SOURCES + ['-o', lib_of(pol)]
        subprocess.run(cmd, check=True)
        print('built', os.path.relpath(lib_of(pol), ROOT), file=sys.stderr)


def run_policy(policy, cases, seed):
    """Child-process body: one policy, `cases` batches x 2 placements."""
    import numpy as np
    from oracle import c_oracle                     # the checker: this script lives under tests/ for that reason
    from proteus_amd import _capi
    from proteus_amd.synth import SEED, synth_tile
    ctx = _capi.Context(0, lib_path=lib_of(policy))
    rng = np.random.default_rng(seed)
    p = _capi.default_params()
    bad, total, examples = 0, 0, []
    for it in range(cases):
        n_tiles, h, w = int(rng.integers(1, 5)), int(rng.integers(50, 700)), int(rng.integers(50, 700))
        masks = bool(rng.integers(2))
        for junk in [ctx.malloc(int(rng.integers(1, 64)) << 20) for _ in range(int(rng.integers(0, 4)))]:
            junk.free()
        b = _capi.DeviceBatch(ctx, n_tiles, h, w, masks=masks, sliding_outputs=True)
        b.synth(SEED, tile0=100 + it)
        region = _capi.batch_layout(n_tiles, h, w, masks=masks, sliding_outputs=True)['write_span_bytes']
        for rep in range(2):
            b.place_slide(p, slack_bytes=int(region * rng.uniform(0.5, 3.0)),
                          step_bytes=int(rng.choice([1 << 20, 2 << 20, 5 << 19, 3 << 20])),
                          spread_gaps=int(rng.integers(0, 5)), refine_passes=int(rng.integers(0, 2)), launches=1)
            b.classify(p)
            ctx.synchronize()
            total += 1
            wrong = []
            for t in range(n_tiles):
                s_ = synth_tile(100 + it + t, h, w, with_masks=masks)
                kw = dict(land=s_['land'], shad=s_['shad'], ocean=s_['ocean']) if masks else {}
                exp = c_oracle.classify(p, s_['bands'], s_['fmask'], **kw)
                for key in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
                    got = b.read_tile(key, t)
                    if not np.array_equal(got, exp[key]):
                        wrong.append({'tile': t, 'layer': key, 'all_zero': bool((got == 0).all()),
                                      'wrong_px': int((got != exp[key]).sum()), 'px': int(got.size)})
            if wrong:
                bad += 1
                if len(examples) < 4:
                    examples.append({'case': it, 'placement': rep, 'wrong': wrong[:4]})
        b.free()
    acct = _capi.va_budget() if hasattr(ctx.lib, 'dswx_batch_va_budget') else {}
    ctx.close()
    print(json.dumps({'policy': policy, 'what': POLICIES[policy], 'cases': total, 'cases_with_a_wrong_layer': bad,
                      'examples': examples, 'address_space': acct}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--cases', type=int, default=80)
    ap.add_argument('--seed', type=int, default=2026)
    ap.add_argument('--policy', type=int, default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.build:
        build_variants()
        return 0
    if a.policy is not None:
        run_policy(a.policy, a.cases, a.seed)
        return 0
    out = []
    for pol in POLICIES:
        if not os.path.exists(lib_of(pol)):
            out.append({'policy': pol, 'error': 'variant not built (run with --build where hipcc is)'})
            continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--policy', str(pol), '--cases', str(a.cases),
                            '--seed', str(a.seed)], capture_output=True, text=True, timeout=3000)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"policy"')]
        out.append(json.loads(lines[-1]) if lines else {'policy': pol, 'error': (r.stderr or r.stdout)[-600:]})
    print(json.dumps({'trial': 'sliding placement x2 per batch, every layer vs the C oracle', 'results': out}, indent=1))
    return 0


if __name__ == '__main__':
    sys.exit(main())
